"""Exposed latency of the ESDF gathers on BASELINE configs[4] (ONE 50 x 50 x 1.6 m map at 0.02 m: a 4 GB field no cache holds).

    TOPAY_LIB=tools/libs/libtopay_stamps.so python3 tools/gpu_stamps_hires.py [scenarios] [size_m]

The diagnostics build (-DTOPAY_STAMPS, tools/ab_lib.sh stamps -DTOPAY_STAMPS) reads the shader clock before and after an
s_waitcnt vmcnt(n) placed where the eight gathers of the sphere being finished are first needed (topay_eval.h): the latency a
wave is actually exposed to, everything the look-ahead and the other resident wave do not cover.  Whole solves of
scenarios x 8 candidates (128 -> one wave per SIMD, 512 -> two); the same on the cached maps is tools/gpu_stamps.py.
The world and the init paths are those of tools/k1_gather.py hires (field built on the device, straight 3-8 m segments).
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harness import workload as wl
from topay_amd import api

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
size = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
gpu = api.MomaTrajOptBatch(device=0, lib_path=os.environ.get("TOPAY_LIB", "tools/libs/libtopay_stamps.so"))
w = wl.World(wl.CUBOIDS, seed=42, size_xy=size, size_z=1.6, res=0.02, cloud_res=0.02, nthreads=-1)
gpu.build_esdf(w.origin, w.res, w.dims, w.min_b, w.max_b, w.occ2d, w.occ3d)
rng = np.random.default_rng(42)
lens, chunks = [], []
for b in range(S * 8):
    a = rng.uniform(-size / 2 + 2, size / 2 - 2, 2)
    while True:
        ang, d = rng.uniform(-np.pi, np.pi), rng.uniform(3.0, 8.0)
        g = a + d * np.array([np.cos(ang), np.sin(ang)])
        if np.all(np.abs(g) < size / 2 - 2):
            break
    k = max(2, int(np.ceil(d / 0.7)) + 1)
    t = np.linspace(0.0, 1.0, k)[:, None]
    q0, q1 = rng.uniform(-1.0, 1.0, 7), rng.uniform(-1.0, 1.0, 7)
    lens.append(k)
    chunks.append(np.concatenate([a + t * (g - a), np.full((k, 1), ang), q0 + t * (q1 - q0)], axis=1))
lens, paths = np.array(lens, dtype=np.int32), np.concatenate(chunks)
gpu.set_init_traj(lens, paths)
B = len(lens)
gpu.set_trace(64)
ok = gpu.optimize()
ms, _ = gpu.last_kernel_ms()
st = gpu.stats()
ev = (st[:, 2] + st[:, 5]).astype(float)
tot = np.zeros(16)
for b in range(B):
    tot += gpu.get_trace(b)[8:8 + 16].view(np.int64)[:16].astype(float)
cyc = tot / ev.sum()
print("map %g m, field %.2f GB; B %d, kernel %.1f ms, evals %d, N mean %.1f, success %.3f" % (size, np.prod(w.dims) * 8 / 1e9, B, ms, ev.sum(), gpu.n_pieces().mean(), ok.mean()))
print("cycles per evaluation + iteration: %.0f (sweep 1 %.0f, of which the manipulator block %.0f)" % (cyc[:10].sum(), cyc[4], cyc[13]))
arr = (C.c_longlong * 8)()
gpu.L.topay_debug_mani_stamps(arr)
v = np.array(list(arr), dtype=float)
calls = v[7] / 12.0
blk = v[:6].sum() / calls
wait = v[6] / calls
# calls of the block per evaluation (lane 0 of block 0 is the stamped one; the per-evaluation figure comes from the per-candidate stamps)
calls_per_eval = cyc[13] / blk if blk > 0 else 0.0
print("manipulator block: %.0f cycles per call, waiting for the gathers of the sphere being finished %.0f (%.1f %% of the block); "
      "%.2f calls per evaluation -> exposed gather latency %.1f %% of a wave's time" % (blk, wait, 100.0 * wait / blk, calls_per_eval, 100.0 * wait * calls_per_eval / cyc[:10].sum()))
