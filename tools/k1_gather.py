#!/usr/bin/env python3
"""K1 = the ESDF-gather kernel of the path: one stage-2 cost + gradient evaluation of every candidate (k_eval1..6 through
topay_eval_batch), timed on its own -- what BASELINE.json's north_star asks to be reported from rocprof against HBM peak.

    python3 tools/k1_gather.py [tables|hires] [repeats]

tables: the headline batch, 1024 scenarios x 8 candidates, one 5.4 MB map per scenario (the gathers are served by L2 /
        Infinity Cache);
hires : BASELINE configs[4], ONE 50 x 50 x 1.6 m map at 0.02 m -- a 4 GB 3-D field no cache holds, so the sphere
        gathers go to HBM (TOPAY_K1_HIRES_SIZE metres, default 50).
Prints one JSON line: ms per evaluation sweep of the batch (HIP events around the launches), algorithmic bytes per
sweep (11 280 N per evaluation: 13 N samples x (4 + 12 x 8) field values + coefficients / gradient in and out), achieved
GB/s.  Run it under `rocprofv3 --kernel-trace --stats` and `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
(tools/profile_k1.sh) for the per-kernel durations and the HBM-side traffic of the same launches.
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from topay_amd import api
from harness import workload as wl

kind = sys.argv[1] if len(sys.argv) > 1 else "tables"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
S, Cn = int(os.environ.get("TOPAY_K1_SCENARIOS", "1024")), 8
t0 = time.time()
opt = api.MomaTrajOptBatch(device=0)
if kind == "hires":
    # The occupancy of the cuboids map comes from the harness, the distance fields are built on the device (the CPU
    # construction of a 5e8-voxel field takes minutes); the init paths are straight 3-8 m segments between random
    # poses of the map (not collision-checked: the gathers of an evaluation do not depend on that).
    size = float(os.environ.get("TOPAY_K1_HIRES_SIZE", "50"))
    w = wl.World(wl.CUBOIDS, seed=42, size_xy=size, size_z=1.6, res=0.02, cloud_res=0.02, nthreads=-1)
    opt.build_esdf(w.origin, w.res, w.dims, w.min_b, w.max_b, w.occ2d, w.occ3d)
    rng = np.random.default_rng(42)
    lens, chunks = [], []
    for b in range(S * Cn):
        a = rng.uniform(-size / 2 + 2, size / 2 - 2, 2)
        while True:
            ang, d = rng.uniform(-np.pi, np.pi), rng.uniform(3.0, 8.0)
            g = a + d * np.array([np.cos(ang), np.sin(ang)])
            if np.all(np.abs(g) < size / 2 - 2):
                break
        k = max(2, int(np.ceil(d / 0.7)) + 1)
        t = np.linspace(0.0, 1.0, k)[:, None]
        q0, q1 = rng.uniform(-1.0, 1.0, 7), rng.uniform(-1.0, 1.0, 7)
        st = np.concatenate([a + t * (g - a), np.full((k, 1), ang), q0 + t * (q1 - q0)], axis=1)
        lens.append(k)
        chunks.append(st)
    lens, paths = np.array(lens, dtype=np.int32), np.concatenate(chunks)
    map_ids = None
    map_bytes = float(np.prod(w.dims)) * 8
else:
    tb = wl.TablesBatch(S, Cn, base_seed=42, nthreads=0, keep_esdf3d=0)
    worlds = [tb.world(s) for s in tb.scenarios]
    w0 = worlds[0]
    opt.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, np.stack([w.occ2d for w in worlds]), np.stack([w.occ3d for w in worlds]))
    slot = {s: k for k, s in enumerate(tb.scenarios)}
    lens, paths, map_ids = tb.lens, tb.paths, np.array([slot[s] for s in tb.scen], dtype=np.int32)
    map_bytes = float(np.prod(w0.dims)) * 8
opt.set_init_traj(lens, paths, map_ids=map_ids)
N = opt.n_pieces().astype(np.float64)
setup = time.time() - t0
opt.eval_batch(2, 2)                 # warm-up
opt.eval_batch(2, reps)
ms, launches = opt.last_kernel_ms()
abytes = float((11280.0 * N).sum())  # per sweep of the batch (one evaluation of every candidate)
gather = float((13.0 * N * 100 * 8).sum())
per = ms / reps
print(json.dumps({"workload": kind, "candidates": int(len(N)), "mean_pieces": float(N.mean()), "repeats": reps, "launches": launches,
                  "ms_per_sweep": per, "algorithmic_bytes_per_sweep": abytes, "esdf_gather_bytes_per_sweep": gather,
                  "achieved_GBps": abytes / (per * 1e-3) / 1e9, "frac_of_8TBps": abytes / (per * 1e-3) / 8e12,
                  "evals_per_s": len(N) / (per * 1e-3), "map_bytes_3d": map_bytes, "setup_s": setup}))
