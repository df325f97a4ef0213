import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from topay_amd.harness import workload as wl
from topay_amd import api
S = int(os.environ.get("S", "512"))
tb = wl.TablesBatch(S, 8, base_seed=42, nthreads=0)
for mode in ("multi", "single"):
    gpu = api.MomaTrajOptBatch(device=0)
    slot = {}
    for k, s in enumerate(tb.scenarios):
        w = tb.world(s)
        gpu.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, map_id=k)
        slot[s] = k
    map_ids = np.array([slot[s] if mode == "multi" else 0 for s in tb.scen], dtype=np.int32)
    gpu.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
    B = len(tb.lens)
    for stage in (1, 2):
        gpu.eval_batch(stage, 2)
        gpu.eval_batch(stage, 10)
        ms, nl = gpu.last_kernel_ms()
        print(mode, "B", B, "stage", stage, "%.1f us per batch-eval, %.3f us/traj-eval" % (ms * 1e3 / 10, ms * 1e3 / 10 / B), "launches", nl, flush=True)
    t = time.time(); ok = gpu.optimize(); ms, nl = gpu.last_kernel_ms(); st = gpu.stats()
    print(mode, "solve %.1f ms" % ms, "evals", (st[:, 2] + st[:, 5]).sum(), "succ", ok.mean(), "N hist", np.bincount(gpu.n_pieces()), flush=True)
    gpu.close()
