"""Where are the idle SIMD slots during the bulk phase?  Per-XCD and per-CU residency from (start, elapsed, hw_id)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topay_amd import api
from harness import workload as wl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nmax_only = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # keep only candidates with N <= this (0 = all)
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
if nmax_only:
    keep = np.where((N > 0) & (N <= nmax_only))[0]
    offs = np.concatenate([[0], np.cumsum(lens)])
    paths = np.concatenate([paths[offs[b]:offs[b + 1]] for b in keep])
    lens = lens[keep]; map_ids = map_ids[keep]
    o.set_init_traj(lens, paths, map_ids=map_ids)
    N = o.n_pieces()
for _ in range(2):
    o.reset(); o.optimize()
su, us, hw = o.start_us(), o.elapsed_us(), o.hw_ids()
m = N > 0
su, us, hw = su[m] - su[m].min(), us[m], hw[m]
T = (su + us).max()
print("B", m.sum(), "span ms", T / 1e3, "N histogram", np.bincount(N[m]))
xcc = hw >> 16
for frac in (0.15, 0.3, 0.45, 0.6):
    t = frac * T
    live = (su <= t) & (su + us > t)
    per_x = np.bincount(xcc[live], minlength=8)
    cu = hw[live] >> 4
    per_cu = np.bincount(np.unique(cu, return_counts=True)[1], minlength=5)
    ncu = len(np.unique(hw >> 4))
    print(f"t={t/1e3:.0f} ms: resident {live.sum()}, per XCC {per_x.tolist()}, CUs with k waves (k=0..4): {[ncu - per_cu[1:].sum()] + per_cu[1:].tolist()}")
    # LDS class of the residents of CUs holding fewer than 4 waves
    cls = np.where(N[m][live] <= 10, 1, np.where(N[m][live] <= 21, 2, 3))
    cu_ids, inv = np.unique(cu, return_inverse=True)
    cnt = np.bincount(inv)
    short = cnt[inv] < 4
    print("    class mix on CUs with <4 waves:", np.bincount(cls[short], minlength=4)[1:].tolist(), " on full CUs:", np.bincount(cls[~short], minlength=4)[1:].tolist())
# calibration of the slot shares of the persistent launches: measured wave-seconds per class vs sum of N^p
Nm = N[m]
cls = np.where(Nm <= 10, 0, np.where(Nm <= 21, 1, 2))
work = np.array([us[cls == k].sum() for k in range(3)])
print("measured work share per class:", (work / work.sum()).round(4).tolist(), " last end per class ms:", [round(float((su + us)[cls == k].max()) / 1e3) for k in range(3)])
for p_ in (1.0, 1.5, 2.0, 2.5):
    w = np.array([(Nm[cls == k].astype(float) ** p_).sum() for k in range(3)])
    print(f"  N^{p_}: ", (w / w.sum()).round(4).tolist())
ev = o.stats(); ev = (ev[:, 2] + ev[:, 5])[m]
for lo, hi in ((4, 6), (7, 8), (9, 10), (11, 13), (14, 17), (18, 21), (22, 26), (27, 32)):
    q = (Nm >= lo) & (Nm <= hi)
    if q.any(): print(f"  N {lo}-{hi}: n {q.sum()}, us per trajectory {us[q].mean():.0f}, per eval {us[q].sum() / ev[q].sum():.1f}, evals {ev[q].mean():.0f}")
