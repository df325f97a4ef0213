"""Synthetic workload harness: deterministic maps, analytic EDT cases, collision-free scenarios."""
import ctypes as C

import numpy as np

from harness import workload as wl


def test_edt_single_voxel_is_analytic():
    L = wl.lib()
    nx, ny, nz, res = 9, 7, 5, 0.1
    occ2 = np.zeros(nx * ny, dtype=np.int8)
    occ3 = np.zeros(nx * ny * nz, dtype=np.int8)
    occ2[4 * ny + 3] = 1
    occ3[(4 * ny + 3) * nz + 2] = 1
    e2 = np.zeros(nx * ny)
    e3 = np.zeros(nx * ny * nz)
    L.wl_edt(occ2.ctypes.data_as(C.c_char_p), occ3.ctypes.data_as(C.c_char_p), nx, ny, nz, C.c_double(res),
             e2.ctypes.data_as(wl.c_dp), e3.ctypes.data_as(wl.c_dp), 2)
    X, Y = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    d2 = res * np.sqrt((X - 4) ** 2 + (Y - 3) ** 2)
    # signed distance: inside the occupied cell pos = 0, neg = res  =>  0 + (-res + res) = 0 (grid_map.cpp:204-206)
    assert np.allclose(e2.reshape(nx, ny), d2, atol=1e-12)
    X, Y, Z = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    d3 = res * np.sqrt((X - 4) ** 2 + (Y - 3) ** 2 + (Z - 2) ** 2)
    assert np.allclose(e3.reshape(nx, ny, nz), d3, atol=1e-12)


def test_worlds_are_deterministic_and_scenarios_are_free():
    a = wl.World(wl.CUBOIDS, seed=7)
    b = wl.World(wl.CUBOIDS, seed=7)
    c = wl.World(wl.CUBOIDS, seed=8)
    assert (a.esdf3d == b.esdf3d).all() and (a.esdf2d == b.esdf2d).all()
    assert not (a.esdf3d == c.esdf3d).all()
    assert list(a.dims) == [200, 200, 16] and a.res == 0.1
    ok, s, g = a.sample_scenario(123)
    assert ok and not a.collision(s) and not a.collision(g)
    assert 3.0 <= np.hypot(*(s[:2] - g[:2])) <= 8.0
    lens, paths = a.init_paths(s, g, 4, 99)
    assert len(lens) == 4 and paths.shape == (lens.sum(), 10)
    off = 0
    for l in lens:
        p = paths[off:off + l]
        off += l
        assert np.allclose(p[0, :2], s[:2]) and np.allclose(p[-1, :2], g[:2])
        assert np.allclose(p[0, 3:], s[3:]) and np.allclose(p[-1, 3:], g[3:])


def test_tables_batch_one_map_per_scenario():
    tb = wl.TablesBatch(3, 2, base_seed=5, nthreads=2)
    assert len(tb.lens) == 6 and list(tb.scen) == [0, 0, 1, 1, 2, 2]
    w0, w1 = tb.world(0), tb.world(1)
    assert not (w0.esdf2d == w1.esdf2d).all()
    tb.close()


def test_tables_batch_can_drop_cpu_esdf3d_without_changing_the_inputs():
    """bench.py keeps the CPU-built 3-D distance field only for the scenarios the CPU baseline solves; paths, occupancy
    grids and the 2-D field must not depend on that."""
    from harness import workload as wl

    a = wl.TablesBatch(4, 2, base_seed=4242, nthreads=4)
    b = wl.TablesBatch(4, 2, base_seed=4242, nthreads=4, keep_esdf3d=1)
    assert (a.lens == b.lens).all() and (a.paths == b.paths).all() and (a.scen == b.scen).all()
    for s in a.scenarios:
        wa, wb = a.world(s), b.world(s)
        assert (wa.occ3d == wb.occ3d).all() and (wa.occ2d == wb.occ2d).all() and (wa.esdf2d == wb.esdf2d).all()
        if s < 1:
            assert (wa.esdf3d == wb.esdf3d).all()
        else:
            assert wb.esdf3d is None
    a.close()
    b.close()
