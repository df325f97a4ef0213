import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harness import workload as wl
from topay_amd import api
S = int(os.environ.get("S", "256"))
w2, lens2, paths2, scen2 = wl.cuboids_batch(S, 8)
res = {}
VARIANT = os.environ.get("TOPAY_VARIANT_LIB")   # optional second build (e.g. another optimisation level) to compare with
for name, lib in (("w2a", None), ("w2b", None)) + ((("w3", VARIANT),) if VARIANT else ()):
    gpu = api.MomaTrajOptBatch(device=0, lib_path=lib)
    gpu.set_map(w2.origin, w2.res, w2.dims, w2.min_b, w2.max_b, w2.esdf2d, w2.esdf3d)
    gpu.set_init_traj(lens2, paths2)
    ok = gpu.optimize()
    res[name] = (ok.copy(), gpu.traj_cost.copy(), gpu.stats().copy())
    print(name, "succ", ok.mean(), "evals", res[name][2][:, 2].sum() + res[name][2][:, 5].sum(), flush=True)
    gpu.close()
for a, b in (("w2a", "w2b"),) + ((("w2a", "w3"),) if VARIANT else ()):
    same = (res[a][1] == res[b][1]) | (np.isnan(res[a][1]) & np.isnan(res[b][1]))
    seq = (res[a][2] == res[b][2]).all(axis=1)
    print(a, "vs", b, ": identical cost", same.mean(), "identical stats", seq.mean(), "first mismatches", np.nonzero(~same)[0][:10])
    bad = np.nonzero(~same)[0]
    for i in bad[:5]:
        print("   traj", i, "N", None, "stats", res[a][2][i], "|", res[b][2][i], "cost", res[a][1][i], res[b][1][i])
if not VARIANT:
    sys.exit(0)
a, b = "w2a", "w3"
same = (res[a][1] == res[b][1]) | (np.isnan(res[a][1]) & np.isnan(res[b][1]))
seq = (res[a][2] == res[b][2]).all(axis=1)
idx = np.nonzero(same & ~seq)[0]
print("same cost, different stats:", len(idx))
for i in idx[:12]:
    print("   traj", i, res[a][2][i], "|", res[b][2][i], "cost", res[a][1][i])
