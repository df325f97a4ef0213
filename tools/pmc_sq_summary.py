#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters collected by tools/pmc_sq.sh (rocpd sqlite databases p1..p3) + derived fractions.
usage: pmc_sq_summary.py gpurun_out/pmc_sq_<tag> [--md]"""
import collections, glob, os, sqlite3, sys
src = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for db in sorted(glob.glob(os.path.join(src, "p*", "pmc_results.db"))):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if view is None:
        print("no counters_collection view in", db, tabs[:20]); continue
    cols = [d[1] for d in c.execute(f"pragma table_info({view})")]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    for name, cn, v in c.execute(f"select {kn}, counter_name, sum(value) from {view} group by {kn}, counter_name"):
        k = name.split("(")[0]
        if "solve" in k or "eval" in k or "k_long" in k or "k_lat" in k:
            agg[k][cn] += v
tot = collections.defaultdict(float)
for k, d in agg.items():
    for a, b in d.items():
        tot[a] += b
agg["all solve kernels"] = tot
for k, d in sorted(agg.items()):
    w = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k}: " + ", ".join(f"{a}={b:.4g}" for a, b in sorted(d.items())))
    print(f"   issue any {d.get('SQ_ACTIVE_INST_ANY',0)/w:.3f}  VALU {d.get('SQ_ACTIVE_INST_VALU',0)/w:.3f}  LDS {d.get('SQ_ACTIVE_INST_LDS',0)/w:.3f}  "
          f"scalar {d.get('SQ_ACTIVE_INST_SCA',0)/w:.3f}  VMEM {d.get('SQ_ACTIVE_INST_VMEM',0)/w:.3f}  wait any {d.get('SQ_WAIT_ANY',0)/w:.3f}  "
          f"wait inst {d.get('SQ_WAIT_INST_ANY',0)/w:.3f}  busy/wave cycles {d.get('SQ_BUSY_CYCLES',0)/w:.3f}  "
          f"icache hit {d.get('SQC_ICACHE_HITS',0)/(d.get('SQC_ICACHE_REQ',0) or 1):.4f}  VALU insts {d.get('SQ_INSTS_VALU',0):.3g}")
