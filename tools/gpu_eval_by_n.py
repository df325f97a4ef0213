"""Time of one stage-2 cost + gradient evaluation by number of pieces (eval hook, 2048 identical-length candidates per N):
what a short last sample pass costs.  13 N samples run in ceil(13 N / 64) passes; N = 5, 10, 15 put 1, 2, 3 samples into a
pass of their own, N = 11 fifteen.  Since round 5 such a pass divides the spheres of its samples over lane groups
(manipulator_block_split).  Prints us per evaluation (batch time / 2048 / repeats x resident share) for N = 4..16.

    python3 tools/gpu_eval_by_n.py            (TOPAY_LIB selects another build)
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import set_map
from topay_amd import api
from harness import workload as wl

w, lens0, paths0, scen0 = wl.cuboids_batch(3, 2)
lib = os.environ.get("TOPAY_LIB")
B, reps = 2048, 20
rows = []
def straight(L):
    k = max(2, int(np.ceil(L / 0.7)) + 1)
    t = np.linspace(0.0, 1.0, k)[:, None]
    a, g = np.array([-8.0, -8.5]), np.array([-8.0 + L * 0.7071, -8.5 + L * 0.7071])
    q0, q1 = np.linspace(0.2, -0.3, 7), np.linspace(-0.4, 0.5, 7)
    return np.concatenate([a + t * (g - a), np.full((k, 1), np.pi / 4), q0 + t * (q1 - q0)], axis=1)
seen = {}
for L in np.arange(2.0, 22.0, 0.25):
    o = api.MomaTrajOptBatch(device=0, lib_path=lib)
    set_map(o, w)
    p = straight(float(L))
    o.set_init_traj(np.array([len(p)], dtype=np.int32), p)
    N = int(o.n_pieces()[0])
    o.close()
    if N in seen or N < 4 or N > 16:
        continue
    seen[N] = float(L)
    o = api.MomaTrajOptBatch(device=0, lib_path=lib)
    set_map(o, w)
    o.set_init_traj(np.full(B, len(p), dtype=np.int32), np.concatenate([p] * B))
    o.eval_batch(2, 3)
    o.eval_batch(2, reps)
    ms, _ = o.last_kernel_ms()
    passes = int(np.ceil(13 * N / 64)); tail = 13 * N - 64 * (passes - 1)
    rows.append((N, passes, tail, ms * 1e3 / reps / B * min(B, 2048)))
    o.close()
rows.sort()
print("N passes tail  us per evaluation sweep of %d candidates / %d" % (B, B))
for N, ps, tl, us in rows:
    print("%2d %d %2d  %.1f" % (N, ps, tl, us))
