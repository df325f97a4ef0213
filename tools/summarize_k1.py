#!/usr/bin/env python3
"""Summary of tools/profile_k1.sh: per-kernel durations of the eval launches (kernel trace) and the HBM-side traffic per
evaluation sweep from the FETCH_SIZE / WRITE_SIZE passes (units and gfx950 correction as /opt/skills/guides/
MI355X_MICROARCH.md prescribes: FETCH_SIZE counts 32-byte... see the guide; tools/pmc_calib.hip measured the factors
used here in round 1: streaming 8 B/lane reads are under-reported by 2, scattered 8-byte reads are reported at 64 B per
touched line with no factor, so 2 x FETCH_SIZE is an upper bound and 1 x a lower bound)."""
import glob, json, os, sqlite3, sys


def markdown(files):
    """`summarize_k1.py --md a.json b.json`: the table kept as profiles/rNN_k1_summary.md."""
    print("# ESDF-gather kernel (K1 = one stage-2 cost + gradient evaluation of all 8192 candidates, `k_eval1..6`) on its own")
    print()
    print("Commands: `tools/profile_k1.sh` = `python3 tools/k1_gather.py {tables|hires}` plain, under `rocprofv3 --kernel-trace --stats`, "
          "and under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes).  Each run does one warm-up sweep and "
          "`repeats` timed sweeps inside one launch per class; counters are sums over both launches, divided by repeats + 1.")
    print()
    print("| workload | 3-D field | mean N | ms / sweep (plain) | ms / sweep (kernel trace) | algorithmic GB / sweep | of which ESDF gathers | "
          "achieved GB/s | of 8 TB/s | FETCH_SIZE GB / sweep (as reported) | WRITE_SIZE GB / sweep | HBM-side GB/s (FETCH + WRITE as reported) |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for f in files:
        d = json.load(open(f))
        r, k = d["runs"]["plain"], d["runs"]["kt"]
        n = r["repeats"] + 1
        fe = d["counters"]["FETCH_SIZE"]["sum"] * 1024 / n / 1e9
        wr = d["counters"]["WRITE_SIZE"]["sum"] * 1024 / n / 1e9
        print(f"| {r['workload']} | {r['map_bytes_3d'] / 1e6:.0f} MB | {r['mean_pieces']:.1f} | {r['ms_per_sweep']:.3f} | {k['ms_per_sweep']:.3f} | "
              f"{r['algorithmic_bytes_per_sweep'] / 1e9:.3f} | {r['esdf_gather_bytes_per_sweep'] / 1e9:.3f} | {r['achieved_GBps']:.0f} | "
              f"{r['frac_of_8TBps']:.3f} | {fe:.3f} | {wr:.3f} | {(fe + wr) / (r['ms_per_sweep'] * 1e-3):.0f} |")
    print()
    print("Per-kernel durations from the kernel trace (two launches per class: warm-up of 1 sweep, then `repeats` sweeps):")
    print()
    print("| workload | kernel | launches | total us | longest launch us |")
    print("|---|---|---:|---:|---:|")
    for f in files:
        d = json.load(open(f))
        for e in d["eval_kernels"]:
            print(f"| {d['runs']['plain']['workload']} | `{e['kernel']}` | {e['calls']} | {e['total_us']:.0f} | {e['max_us']:.0f} |")


if len(sys.argv) > 1 and sys.argv[1] == "--md":
    markdown(sys.argv[2:])
    sys.exit(0)
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
