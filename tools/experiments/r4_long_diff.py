"""Which long candidates (N >= 33) of the bench batch changed their solve between two builds (TOPAY_LIB_A / TOPAY_LIB_B)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from topay_amd import api
from harness import workload as wl
tb = wl.TablesBatch(1024, 8, base_seed=42, nthreads=0)
worlds = [tb.world(s) for s in tb.scenarios]
slot = {s: k for k, s in enumerate(tb.scenarios)}
map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)
res = {}
for tag in ("A", "B"):
    o = api.MomaTrajOptBatch(device=0, lib_path=os.environ["TOPAY_LIB_" + tag])
    w0 = worlds[0]
    o.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, np.stack([w.occ2d for w in worlds]), np.stack([w.occ3d for w in worlds]))
    o.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
    N = o.n_pieces()
    keep = np.where(N >= 33)[0]
    offs = np.concatenate([[0], np.cumsum(tb.lens)])
    paths = np.concatenate([tb.paths[offs[b]:offs[b + 1]] for b in keep])
    o.set_init_traj(tb.lens[keep], paths, map_ids=map_ids[keep])
    o.optimize()
    st = o.stats()
    res[tag] = (o.n_pieces(), st[:, 2] + st[:, 5], o.elapsed_us())
    o.close()
NA, eA, tA = res["A"]; NB, eB, tB = res["B"]
ch = np.nonzero(eA != eB)[0]
print("long candidates", len(NA), "changed", len(ch))
for b in ch:
    print("  N %d: evaluations %d -> %d, device ms %.0f -> %.0f" % (NA[b], eA[b], eB[b], tA[b] / 1e3, tB[b] / 1e3))
print("longest solve: A %.0f ms (N %d, %d evaluations), B %.0f ms (N %d, %d evaluations)" % (tA.max() / 1e3, NA[tA.argmax()], eA[tA.argmax()], tB.max() / 1e3, NB[tB.argmax()], eB[tB.argmax()]))
