"""ESDF construction on the device (GridMap::updateESDF, grid_map.cpp:89-521) against the CPU restatement used by the
workload harness (harness/workload.hpp: fillESDF / updateESDF2d / updateESDF3d, same file:line citations):
bit-exact, as the arithmetic is integer index math, one division per envelope test and res * sqrt at the end."""
import numpy as np
import pytest

from conftest import EMU_LIB, set_map
from oracle import oracle as orc
from topay_amd import api
from harness import workload as wl


def _build_and_compare(opt, w):
    opt.build_esdf(w.origin, w.res, w.dims, w.min_b, w.max_b, w.occ2d, w.occ3d)
    e2, e3, ms = opt.get_map()
    assert (e2 == w.esdf2d).all() and (e3 == w.esdf3d).all()
    # the two front-end fields of updateESDF (inflate, critical -> critical-inflate; grid_map.cpp:211-423)
    inf, crit = opt.get_map_fields()
    inf_ref, crit_ref = wl.front_end_fields(w.occ2d, w.occ3d, w.dims, w.res)
    assert (inf == inf_ref).all() and (crit == crit_ref).all()
    assert (inf <= e2 + 1e-12).all() and (np.abs(inf - e2) > 0.05).any()   # inflating obstacles brings them closer
    return ms


def test_edt_kernel_sources_match_cpu_construction():
    for kind, seed, size in ((wl.CUBOIDS, 5, 4.0), (wl.TABLES, 6, 6.0)):
        w = wl.World(kind, seed=seed, size_xy=size, size_z=1.6, res=0.1, cloud_res=0.05)
        assert w.occ3d.sum() > 0
        _build_and_compare(api.MomaTrajOptBatch(lib_path=EMU_LIB), w)
        w.close()


def test_edt_edge_maps():
    """Empty map (no obstacle: every distance is DMAX-derived) and a single occupied voxel (analytic distances)."""
    w = wl.World(wl.CUBOIDS, seed=5, size_xy=4.0, size_z=1.6, res=0.1, cloud_res=0.05)
    emu = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    nx, ny, nz = (int(x) for x in w.dims)
    occ2 = np.zeros(nx * ny, dtype=np.int8)
    occ3 = np.zeros(nx * ny * nz, dtype=np.int8)
    occ3[(10 * ny + 12) * nz + 5] = 1
    occ2[10 * ny + 12] = 1
    emu.build_esdf(w.origin, w.res, w.dims, w.min_b, w.max_b, occ2, occ3)
    e2, e3, _ = emu.get_map()
    e3 = e3.reshape(nx, ny, nz)
    e2 = e2.reshape(nx, ny)
    X, Y, Z = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    d = w.res * np.sqrt((X - 10.0) ** 2 + (Y - 12.0) ** 2 + (Z - 5.0) ** 2)
    free = d > 0
    assert np.allclose(e3[free], d[free], rtol=0, atol=1e-12)
    d2 = w.res * np.sqrt((X[:, :, 0] - 10.0) ** 2 + (Y[:, :, 0] - 12.0) ** 2)
    assert np.allclose(e2[d2 > 0], d2[d2 > 0], rtol=0, atol=1e-12)
    # an explicit critical occupancy (a second pillar that exists only above the chassis height) enters the critical
    # field and not the others; inflate = distance to the disc of cells closer than the chassis radius
    crit_occ = occ2.copy()
    crit_occ[25 * ny + 30] = 1
    emu.build_esdf_fields(w.origin, w.res, w.dims, w.min_b, w.max_b, occ2, crit_occ, occ3)
    inf, crit = emu.get_map_fields()
    inf_ref, crit_ref = wl.front_end_fields(occ2, occ3, w.dims, w.res, occ2d_critical=crit_occ)
    assert (inf == inf_ref).all() and (crit == crit_ref).all()
    inf = inf.reshape(nx, ny)
    crit = crit.reshape(nx, ny)
    assert inf[10, 12] < 0 and inf[10, 12 + 8] > 0 and abs(inf[10, 12 + 8] - (d2[10, 12 + 8] - 0.4)) < 0.11
    assert crit[25, 30] < 0 and inf[25, 30] > 0
    w.close()


@pytest.mark.gpu
def test_edt_on_gpu_is_bit_exact_and_feeds_the_solver():
    opt = api.MomaTrajOptBatch(device=0)
    w = wl.World(wl.TABLES, seed=11, size_xy=20.0, size_z=1.6, res=0.1, cloud_res=0.05)      # the benchmark map size
    ms1 = _build_and_compare(opt, w)
    # the GPU-built map drives a solve exactly like the uploaded one
    ok_s, s, g = w.sample_scenario(77)
    lens, paths = w.init_paths(s, g, 4, 123)
    ok_a = opt.optimizeTraj(lens, paths)
    cost_a = opt.traj_cost.copy()
    ref = api.MomaTrajOptBatch(device=0)
    set_map(ref, w)
    ok_b = ref.optimizeTraj(lens, paths)
    assert (ok_a == ok_b).all() and (cost_a[ok_a] == ref.traj_cost[ok_b]).all()
    w.close()
    w2 = wl.World(wl.CUBOIDS, seed=12, size_xy=10.0, size_z=1.6, res=0.02, cloud_res=0.02)  # 500 x 500 x 80 cells
    ms2 = _build_and_compare(opt, w2)
    w2.close()
    print(f"EDT build: 200x200x16 {ms1:.2f} ms, 500x500x80 {ms2:.2f} ms")


@pytest.mark.gpu
def test_edt_batch_of_benchmark_maps():
    """64 'tables' maps of the benchmark size built in one call: every slot equals its CPU-built map bit for bit."""
    tb = wl.TablesBatch(64, 2, base_seed=900, nthreads=0)
    worlds = [tb.world(s) for s in tb.scenarios]
    o2 = np.stack([w.occ2d for w in worlds])
    o3 = np.stack([w.occ3d for w in worlds])
    w0 = worlds[0]
    opt = api.MomaTrajOptBatch(device=0)
    opt.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, o2, o3)
    for k in (0, 1, 31, 63):
        e2, e3, ms = opt.get_map(k)
        assert (e2 == worlds[k].esdf2d).all() and (e3 == worlds[k].esdf3d).all()
    print(f"EDT batch: 64 maps of 200x200x16 in {ms:.2f} ms")
    tb.close()
