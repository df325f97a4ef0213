"""Timing of the GPU ESDF construction: the benchmark's per-scenario maps in one batched call, and one big map."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from topay_amd import api
from harness import workload as wl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
t = time.time(); tb = wl.TablesBatch(S, 1, base_seed=42, nthreads=0); cpu_s = time.time() - t
worlds = [tb.world(s) for s in tb.scenarios]
o2 = np.stack([w.occ2d for w in worlds]); o3 = np.stack([w.occ3d for w in worlds])
w0 = worlds[0]
opt = api.MomaTrajOptBatch(device=0)
for rep in range(2):
    t = time.time(); opt.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, o2, o3); wall = time.time() - t
_, e3, ms = opt.get_map(S - 1)
vox = S * int(np.prod(w0.dims))
print(f"{S} maps 200x200x16: kernels {ms:.1f} ms (wall incl. upload {wall*1e3:.0f} ms), {vox/ms/1e6:.2f} Gvoxel/s, "
      f"~{vox*100/ms/1e6:.0f} GB/s at 100 algorithmic B/voxel; last map equal {(e3 == worlds[-1].esdf3d).all()}; "
      f"CPU harness built maps+paths in {cpu_s:.1f} s")
size = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
t = time.time(); w = wl.World(wl.CUBOIDS, seed=3, size_xy=size, size_z=1.6, res=0.02, cloud_res=0.02); cpu_s = time.time() - t
opt2 = api.MomaTrajOptBatch(device=0)
for rep in range(2):
    opt2.build_esdf(w.origin, w.res, w.dims, w.min_b, w.max_b, w.occ2d, w.occ3d)
e2, e3, ms = opt2.get_map()
vox = int(np.prod(w.dims))
print(f"one map {tuple(w.dims)}: kernels {ms:.1f} ms, {vox/ms/1e6:.2f} Gvoxel/s, ~{vox*100/ms/1e6:.0f} GB/s; equal {(e3 == w.esdf3d).all()} {(e2 == w.esdf2d).all()}; CPU harness (cloud + EDT) {cpu_s:.1f} s")
