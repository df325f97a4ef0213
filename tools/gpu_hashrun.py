import sys, hashlib, numpy as np
sys.path.insert(0,'.')
from topay_amd import api
from harness import workload as wl
tb = wl.TablesBatch(256, 8, base_seed=42, nthreads=0)
opt = api.MomaTrajOptBatch(device=0)   # TOPAY_LIB selects another build (codegen cross-check: the hash must not depend on the optimisation level)
slot = {}
for k, s in enumerate(tb.scenarios):
    w = tb.world(s); opt.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, map_id=k); slot[s] = k
opt.set_init_traj(tb.lens, tb.paths, map_ids=np.array([slot[s] for s in tb.scen], dtype=np.int32))
ok = opt.optimize()
h = hashlib.sha256(); h.update(ok.tobytes()); h.update(np.nan_to_num(opt.traj_cost).tobytes()); h.update(opt.stats().tobytes())
for b in (0, 100, 777): h.update(opt.get_x(b).tobytes())
print(h.hexdigest()[:16], ok.mean())
