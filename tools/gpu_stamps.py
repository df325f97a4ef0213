import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harness import workload as wl
from topay_amd import api
gpu = api.MomaTrajOptBatch(device=0, lib_path=os.environ.get("TOPAY_LIB", "topay_amd/lib/libtopay_hip_stamps.so"))
names = ["fill", "LU", "subst", "jerk", "sweep1", "between", "rows+sweep2", "adjoint", "assemble", "lbfgs", "(twoloop)", "(state read back)", "(trial / accepted it. incl. twoloop)", "(ls prologue)", "(-)", "(poll, barrier, park)"]
for S in (int(sys.argv[1]) if len(sys.argv) > 1 else 128,):
    w2, lens2, paths2, scen2 = wl.cuboids_batch(S, 8)
    gpu.set_map(w2.origin, w2.res, w2.dims, w2.min_b, w2.max_b, w2.esdf2d, w2.esdf3d)
    gpu.set_init_traj(lens2, paths2)
    B = len(lens2)
    gpu.set_trace(64)
    t = time.time(); ok = gpu.optimize(); ms, nl = gpu.last_kernel_ms()
    st = gpu.stats()
    ev = (st[:, 2] + st[:, 5]).astype(float)
    tot = np.zeros(16)
    for b in range(B):
        raw = gpu.get_trace(b)
        tk = raw[8:8 + 16].view(np.int64)[:16].astype(float)
        tot += tk
    print("B", B, "kernel %.1f ms; evals %d" % (ms, ev.sum()))
    cyc_per_eval = tot / ev.sum()
    for n, c in zip(names, cyc_per_eval):
        print("   %-9s %10.0f cycles/eval" % (n, c))
    print("   total %.0f cycles/eval" % cyc_per_eval[:10].sum(), " N mean", gpu.n_pieces().mean(), "mean bound per iter", st[:, 7].sum() / max(1, st[:, 4].sum()))

import ctypes as C
try:
    arr = (C.c_longlong * 8)()
    gpu.L.topay_debug_mani_stamps(arr)
    v = np.array(list(arr), dtype=float)
    print("manipulator phases (share):", dict(zip(["sincos", "walk1", "pairs", "esdf", "walks2", "limits"], np.round(v[:6] / v[:6].sum(), 3))))
    if v[7] > 0:
        calls = v[7] / 12.0
        print("manipulator block: %.0f cycles per call, ESDF loop %.0f, of which waiting for the gathers of the sphere being finished %.0f (%.1f %% of the block)"
              % (v[:6].sum() / calls, v[3] / calls, v[6] / calls, 100.0 * v[6] / v[:6].sum()))
except Exception as e:
    print("no mani stamps", e)
