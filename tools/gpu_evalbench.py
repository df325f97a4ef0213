import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harness import workload as wl
from topay_amd import api
S = int(os.environ.get("S", "256")); R = int(os.environ.get("R", "10")); stage = int(os.environ.get("STAGE", "2"))
gpu = api.MomaTrajOptBatch(device=0)
w2, lens2, paths2, scen2 = wl.cuboids_batch(S, 8)
gpu.set_map(w2.origin, w2.res, w2.dims, w2.min_b, w2.max_b, w2.esdf2d, w2.esdf3d)
gpu.set_init_traj(lens2, paths2)
gpu.eval_batch(stage, R)
ms, nl = gpu.last_kernel_ms()
N = gpu.n_pieces()
print("B", len(lens2), "stage", stage, "R", R, "ms", ms, "N mean", N.mean(), "evals", len(lens2) * R)
