"""How many workgroups really share a compute unit / a SIMD during the bulk of a solve, and what an evaluation costs then.
usage: gpu_occupancy.py [scenarios] [keep only N <= nmax] [keep only N >= nmin]   (TOPAY_LIB selects the build)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topay_amd import api
from harness import workload as wl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nmax_only = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nmin_only = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tb = wl.TablesBatch(S, 8, base_seed=42, nthreads=0)
worlds = [tb.world(s) for s in tb.scenarios]
slot = {s: k for k, s in enumerate(tb.scenarios)}
map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)
o = api.MomaTrajOptBatch(device=0)
w0 = worlds[0]
o.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, np.stack([w.occ2d for w in worlds]), np.stack([w.occ3d for w in worlds]))
lens, paths = tb.lens, tb.paths
o.set_init_traj(lens, paths, map_ids=map_ids)
N = o.n_pieces()
if nmax_only or nmin_only:
    keep = np.where((N > 0) & (N <= (nmax_only or 999)) & (N >= nmin_only))[0]
    offs = np.concatenate([[0], np.cumsum(lens)])
    paths = np.concatenate([paths[offs[b]:offs[b + 1]] for b in keep])
    lens = lens[keep]; map_ids = map_ids[keep]
    o.set_init_traj(lens, paths, map_ids=map_ids)
    N = o.n_pieces()
for _ in range(2):
    o.reset(); o.optimize()
ms = o.last_kernel_ms()[0]
su, us, hw = o.start_us(), o.elapsed_us(), o.hw_ids()
m = N > 0
su, us, hw, Nm = su[m] - su[m].min(), us[m], hw[m], N[m]
T = (su + us).max()
ev = o.stats(); ev = (ev[:, 2] + ev[:, 5])[m]
print(f"B {m.sum()} kernel ms {ms:.1f} span ms {T/1e3:.1f} traj/s {m.sum()/ms*1e3:.0f}  sum device-s {us.sum()*1e-6:.1f}  mean resident {us.sum()/T:.0f}")
for frac in (0.2, 0.4, 0.6):
    t = frac * T
    live = (su <= t) & (su + us > t)
    cu = hw[live] >> 4
    simd = hw[live]
    _, ccnt = np.unique(cu, return_counts=True)
    _, scnt = np.unique(simd, return_counts=True)
    print(f" t={t/1e3:.0f} ms: resident {live.sum()}  CUs by workgroups {np.bincount(ccnt).tolist()}  SIMDs by workgroups {np.bincount(scnt).tolist()}")
for lo, hi in ((4, 6), (7, 8), (9, 10), (11, 13), (14, 15), (16, 21), (22, 32), (33, 64)):
    q = (Nm >= lo) & (Nm <= hi)
    if q.any(): print(f"  N {lo}-{hi}: n {q.sum()}, us per trajectory {us[q].mean():.0f}, per eval {us[q].sum() / ev[q].sum():.1f}, evals {ev[q].mean():.0f}")
print("  device-seconds by N (N: candidates, seconds):", " ".join(f"{n}:{(Nm == n).sum()},{us[Nm == n].sum() * 1e-6:.1f}" for n in range(16, 65) if (Nm == n).any()))

