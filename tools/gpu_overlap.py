"""Do two batches issued back to back on two contexts overlap on the device?  Per-trajectory start/end on the device's
constant clock for both; TOPAY_PRIVATE_STREAMS=1 gives every context its own bucket streams."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topay_amd import api
from harness import workload as wl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tb = wl.TablesBatch(S, 8, base_seed=42, nthreads=0)
worlds = [tb.world(s) for s in tb.scenarios]
slot = {s: k for k, s in enumerate(tb.scenarios)}
map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)
opts = []
for _ in range(2):
    o = api.MomaTrajOptBatch(device=0)
    w0 = worlds[0]
    o.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, np.stack([w.occ2d for w in worlds]),
                       np.stack([w.occ3d for w in worlds]))
    o.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
    opts.append(o)
for o in opts:
    o.reset(); o.optimize()
for rep in range(2):
    t0 = time.perf_counter()
    ti = []
    for o in opts:
        o.reset(); o.optimize_async(); ti.append(time.perf_counter() - t0)
    for o in opts:
        o.finish()
    t1 = time.perf_counter() - t0
    su = [o.start_us() for o in opts]
    us = [o.elapsed_us() for o in opts]
    N = opts[0].n_pieces()
    m = N > 0
    base = min(s[m].min() for s in su)
    print(f"rep {rep}: host issue times {ti}, both finished after {t1*1e3:.0f} ms")
    for k in range(2):
        s = su[k][m] - base; e = s + us[k][m]
        print(f"  batch {k}: first start {s.min()/1e3:.0f} ms, median start {np.median(s)/1e3:.0f}, last start {s.max()/1e3:.0f}, last end {e.max()/1e3:.0f}; kernel_ms {opts[k].last_kernel_ms()[0]:.0f}")
    T = max((su[k][m] - base + us[k][m]).max() for k in range(2))
    edges = np.linspace(0, T, 21)
    for k in range(2):
        s = su[k][m] - base; e = s + us[k][m]
        occ = [int(round(np.clip(np.minimum(e, edges[i + 1]) - np.maximum(s, edges[i]), 0, None).sum() / (edges[i + 1] - edges[i]))) for i in range(20)]
        print(f"  batch {k} resident waves per 5% bin of {T/1e3:.0f} ms:", occ)
