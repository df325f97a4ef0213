"""Solve kernel time of the cuboids batch (S scenarios x 8 candidates on one map) for the library TOPAY_LIB selects: A/B timing of a
workload other than the bench's, best and mean of R solves."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harness import workload as wl
from topay_amd import api
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w2, lens2, paths2, scen2 = wl.cuboids_batch(S, 8)
gpu = api.MomaTrajOptBatch(device=0)
gpu.set_map(w2.origin, w2.res, w2.dims, w2.min_b, w2.max_b, w2.esdf2d, w2.esdf3d)
gpu.set_init_traj(lens2, paths2)
ms = []
for r in range(R + 1):
    gpu.reset(); gpu.optimize(); ms.append(gpu.last_kernel_ms()[0])
print("S %d: kernel ms %s  best %.1f mean %.1f (first run dropped)" % (S, " ".join("%.1f" % m for m in ms[1:]), min(ms[1:]), np.mean(ms[1:])))
