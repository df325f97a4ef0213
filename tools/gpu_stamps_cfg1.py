"""Where a planning call's time goes: per-phase shader-clock stamps of BASELINE configs[1] (one tables scenario x 64 candidates) on
the default kernels and on the helper-wave kernels (topay_set_latency_mode 1), for the candidate that sets the call's time.
Needs the diagnostics build: tools/ab_lib.sh stamps -DTOPAY_STAMPS; TOPAY_LIB=tools/libs/libtopay_stamps.so."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harness import workload as wl
from topay_amd import api
names = ["fill", "LU", "subst", "jerk", "sweep1", "between", "rows+sweep2", "adjoint", "assemble", "lbfgs", "(twoloop)", "(state read back)", "(trial / accepted it. incl. twoloop)", "(ls prologue)", "(-)", "(poll, barrier, park)"]
w1, _, _, lens1, paths1 = wl.tables_scenario(0, 64)
for mode in (0, 1):
    gpu = api.MomaTrajOptBatch(device=0, lib_path=os.environ.get("TOPAY_LIB", "tools/libs/libtopay_stamps.so"))
    gpu.set_map(w1.origin, w1.res, w1.dims, w1.min_b, w1.max_b, w1.esdf2d, w1.esdf3d)
    gpu.set_init_traj(lens1, paths1)
    gpu.set_latency_mode(mode)
    gpu.optimize(); gpu.reset()
    gpu.set_trace(64)
    gpu.optimize(); ms, nl = gpu.last_kernel_ms()
    st = gpu.stats(); N = gpu.n_pieces()
    ev = (st[:, 2] + st[:, 5]).astype(float)
    B = len(lens1)
    tk = np.zeros((B, 16))
    for b in range(B):
        raw = gpu.get_trace(b)
        tk[b] = raw[8:8 + 16].view(np.int64)[:16].astype(float)
    tot = tk[:, :10].sum(axis=1)
    order = np.argsort(-tot)
    print("== latency mode %d: kernel %.1f ms, %d launches (%d helper)" % (mode, ms, nl, gpu.last_helper_launches()))
    for b in order[:3]:
        print("  candidate %d: N %d, %d evaluations (%d + %d), %d iterations, stamped %.1f Mcycles = %.1f ms at 2.4 GHz"
              % (b, N[b], ev[b], st[b, 2], st[b, 5], st[b, 1] + st[b, 4], tot[b] / 1e6, tot[b] / 2.4e6))
    b = order[0]
    for n, c in zip(names, tk[b] / ev[b]):
        print("     %-18s %9.0f cycles per evaluation" % (n, c))
    print("     total %.0f cycles per evaluation (+ iteration)" % (tot[b] / ev[b]))
    gpu.close()
