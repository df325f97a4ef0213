"""Batch size against solve time on the default and on the helper-wave kernels: where should topay_set_latency_mode(1) stop?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from topay_amd import api
from harness import workload as wl
for S in (8, 16, 32, 48, 64, 96, 128, 192):
    tb = wl.TablesBatch(S, 8, base_seed=42, nthreads=0)
    worlds = [tb.world(s) for s in tb.scenarios]
    slot = {s: k for k, s in enumerate(tb.scenarios)}
    map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)
    out = []
    for mode in (0, 2):
        o = api.MomaTrajOptBatch(device=0)
        w0 = worlds[0]
        o.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, np.stack([w.occ2d for w in worlds]), np.stack([w.occ3d for w in worlds]))
        o.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
        o.set_latency_mode(mode)
        ms = []
        for _ in range(3):
            o.reset(); o.optimize(); ms.append(o.last_kernel_ms()[0])
        out.append(min(ms))
        o.close()
    print("B %4d (N max %d): default %.1f ms, helper waves %.1f ms (%.2f)" % (len(tb.lens), max(1, 0), out[0], out[1], out[1] / out[0]))
