import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from topay_amd.harness import workload as wl
from topay_amd import api
LIB=os.environ.get('TOPAY_LIB'); print('lib', LIB); gpu = api.MomaTrajOptBatch(device=0, lib_path=LIB)
for S in (32, 256, 1024):
    w2, lens2, paths2, scen2 = wl.cuboids_batch(S, 8)
    gpu.set_map(w2.origin, w2.res, w2.dims, w2.min_b, w2.max_b, w2.esdf2d, w2.esdf3d)
    gpu.set_init_traj(lens2, paths2)
    B = len(lens2)
    for stage in (1, 2):
        gpu.eval_batch(stage, 2)
        R = 20
        gpu.eval_batch(stage, R)
        ms, nl = gpu.last_kernel_ms()
        print("B", B, "stage", stage, "eval x%d: %.2f ms -> %.1f us per batch-eval, %.3f us/traj-eval" % (R, ms, ms * 1e3 / R, ms * 1e3 / R / B), flush=True)
    t = time.time(); ok = gpu.optimize(); ms, nl = gpu.last_kernel_ms()
    st = gpu.stats()
    ev = st[:, 2].sum() + st[:, 5].sum()
    print("B", B, "solve %.1f ms, total evals %d -> %.3f us per traj-eval overall; sum_bound %d; traj/s %.0f" % (ms, ev, ms * 1e3 / ev, st[:, 7].sum(), B / ms * 1e3), flush=True)
