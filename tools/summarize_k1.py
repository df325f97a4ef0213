#!/usr/bin/env python3
"""Summary of tools/profile_k1.sh: per-kernel durations of the eval launches (kernel trace) and the HBM-side traffic per
evaluation sweep from the FETCH_SIZE / WRITE_SIZE passes (units and gfx950 correction as /opt/skills/guides/
MI355X_MICROARCH.md prescribes: FETCH_SIZE counts 32-byte... see the guide; tools/pmc_calib.hip measured the factors
used here in round 1: streaming 8 B/lane reads are under-reported by 2, scattered 8-byte reads are reported at 64 B per
touched line with no factor, so 2 x FETCH_SIZE is an upper bound and 1 x a lower bound)."""
import glob, json, os, sqlite3, sys
out = sys.argv[1]
res = {}
for name in ("plain", "kt", "fetch", "write"):
    try:
        res[name] = json.loads(open(os.path.join(out, name + ".json")).read().strip().splitlines()[-1])
    except Exception as ex:
        res[name] = {"error": str(ex)}
def db(sub):
    f = glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None
rows = []
d = db("kt")
if d:
    for r in d.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, max(end-start)/1e3 from kernels where name like '%k_eval%' group by name order by 3 desc"):
        rows.append(dict(kernel=r[0].split("(")[0], calls=r[1], total_us=r[2], avg_us=r[3], max_us=r[4]))
ctr = {}
for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    d = db(sub)
    if d:
        tot, n = 0.0, 0
        for name, val in d.execute("select kernel_name, value from counters_collection where counter_name = ?", (cname,)):
            if "k_eval" in name:
                tot += val
                n += 1
        ctr[cname] = dict(sum=tot, dispatches=n)
print(json.dumps(dict(runs=res, eval_kernels=rows, counters=ctr), indent=1))
