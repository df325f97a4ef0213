"""Known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors for this path, so the oracle is pinned by closed-form facts:
banded solves against dense numpy, MINCO interpolation/continuity/energy identities, finite differences of the
FK and of the full cost, analytic ESDF cases and L-BFGS behaviour on textbook functions.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc

c_dp = C.POINTER(C.c_double)


def _dp(a):
    return a.ctypes.data_as(c_dp)


@pytest.fixture(scope="module")
def L():
    lib = orc.lib()
    lib.orc_smooth_l1.argtypes = [C.c_double, C.c_double, c_dp]
    lib.orc_reparam.argtypes = [C.c_double, C.c_double, c_dp]
    return lib


# ---- 1. BandedSystem (banded_system.hpp) ---------------------------------------------------------------
@pytest.mark.parametrize("n,bw", [(18, 6), (30, 6), (24, 3)])
def test_banded_solve_and_adjoint(L, n, bw):
    rng = np.random.default_rng(n)
    A = np.zeros((n, n))
    for i in range(n):
        for j in range(max(0, i - bw), min(n, i + bw + 1)):
            A[i, j] = rng.standard_normal()
        A[i, i] += 4.0 * bw  # diagonally dominant: no pivoting needed
    B = rng.standard_normal((n, 9))
    for adjoint, M in ((0, A), (1, A.T)):
        b = np.asfortranarray(B.copy())
        L.orc_banded_solve(_dp(np.ascontiguousarray(A)), n, bw, _dp(b), 9, adjoint)
        assert np.allclose(M @ b, B, rtol=1e-11, atol=1e-11)


# ---- 2/3. MINCO (minco.hpp:824-1069) -------------------------------------------------------------------
def _minco(L, N, head, tail, inner, T, ew, gdC=None, gdT=None):
    c = np.zeros(6 * N * 9)
    jerk = C.c_double(0)
    gC = np.zeros(6 * N * 9)
    gT = np.zeros(N)
    gP = np.zeros(9 * (N - 1))
    gTail = np.zeros(27)
    gT_io = None if gdT is None else gdT.copy()
    L.orc_minco(N, _dp(ew), _dp(head), _dp(tail), _dp(inner), _dp(T), _dp(c), C.byref(jerk), _dp(gC), _dp(gT),
                None if gdC is None else _dp(gdC), None if gT_io is None else _dp(gT_io), _dp(gP), _dp(gTail))
    return c.reshape(9, 6 * N), jerk.value, gC, gT, gT_io, gP.reshape(N - 1, 9), gTail.reshape(3, 9)


def _poly(c6, t, der=0):
    # c6: coefficients lowest order first
    out = 0.0
    for k in range(der, 6):
        f = 1.0
        for q in range(der):
            f *= (k - q)
        out += f * c6[k] * t ** (k - der)
    return out


def test_minco_interpolates_and_is_c4(L):
    rng = np.random.default_rng(3)
    N = 5
    head = rng.standard_normal(27)
    tail = rng.standard_normal(27)
    inner = rng.standard_normal((N - 1) * 9)
    T = rng.uniform(0.7, 2.0, N)
    ew = np.array([0.33] + [1.0] * 8)
    c, jerk, gC, gT, _, _, _ = _minco(L, N, head, tail, inner, T, ew)
    for d in range(9):
        cd = c[d].reshape(N, 6)
        # boundary position / velocity / acceleration (head/tail are 9x3 column-major: [col*9 + d])
        for der in range(3):
            assert np.isclose(_poly(cd[0], 0.0, der), head[der * 9 + d], atol=1e-9)
            assert np.isclose(_poly(cd[-1], T[-1], der), tail[der * 9 + d], atol=1e-8)
        for i in range(N - 1):
            assert np.isclose(_poly(cd[i], T[i], 0), inner[i * 9 + d], atol=1e-9)  # passes through the inner point
            for der in range(5):  # C4 at interior knots
                assert np.isclose(_poly(cd[i], T[i], der), _poly(cd[i + 1], 0.0, der), rtol=1e-8, atol=1e-7)
    # jerk energy = weighted integral of jerk^2 (Gauss-Legendre, exact for the degree-4 integrand)
    xg, wg = np.polynomial.legendre.leggauss(6)
    ref = 0.0
    for d in range(9):
        cd = c[d].reshape(N, 6)
        for i in range(N):
            ts = 0.5 * T[i] * (xg + 1)
            ref += ew[d] * 0.5 * T[i] * np.sum(wg * np.array([_poly(cd[i], t, 3) for t in ts]) ** 2)
    assert np.isclose(jerk, ref, rtol=1e-10)


def test_minco_adjoint_matches_finite_differences(L):
    rng = np.random.default_rng(4)
    N = 4
    head = rng.standard_normal(27)
    tail = rng.standard_normal(27)
    inner = rng.standard_normal((N - 1) * 9)
    T = rng.uniform(0.8, 1.8, N)
    ew = np.array([0.33] + [1.0] * 8)
    _, j0, gC, gT, _, _, _ = _minco(L, N, head, tail, inner, T, ew)
    _, _, _, _, gT_tot, gP, gTail = _minco(L, N, head, tail, inner, T, ew, gdC=gC, gdT=gT)
    h = 1e-6
    for i in range(N):
        Tp, Tm = T.copy(), T.copy()
        Tp[i] += h
        Tm[i] -= h
        fd = (_minco(L, N, head, tail, inner, Tp, ew)[1] - _minco(L, N, head, tail, inner, Tm, ew)[1]) / (2 * h)
        assert np.isclose(gT_tot[i], fd, rtol=1e-6, atol=1e-5)
    for k in rng.choice((N - 1) * 9, 10, replace=False):
        ip, im = inner.copy(), inner.copy()
        ip[k] += h
        im[k] -= h
        fd = (_minco(L, N, head, tail, ip, T, ew)[1] - _minco(L, N, head, tail, im, T, ew)[1]) / (2 * h)
        assert np.isclose(gP.reshape(-1)[k], fd, rtol=1e-6, atol=1e-5)
    # tail position of the arc-length dimension (the only gdTail entry the path uses, moma_traj_opt.cpp:948)
    tp, tm = tail.copy(), tail.copy()
    tp[1] += h
    tm[1] -= h
    fd = (_minco(L, N, head, tp, inner, T, ew)[1] - _minco(L, N, head, tm, inner, T, ew)[1]) / (2 * h)
    assert np.isclose(gTail[0, 1], fd, rtol=1e-6, atol=1e-5)


# ---- 4. C2 reparameterisations and smoothL1 (moma_traj_opt.h:745-830) ------------------------------------
def test_reparameterisations(L):
    out = np.zeros(6)
    for tau in (-3.0, -0.4, 0.0, 0.3, 2.5):
        L.orc_reparam(tau, 3.1, _dp(out))
        T = out[0]
        assert T > 0
        o2 = np.zeros(6)
        L.orc_reparam(T, 3.1, _dp(o2))
        if T > 0:
            # logC2(expC2(tau)) == tau  (orc_reparam returns logC2(v) in slot 1 for v > 0)
            assert np.isclose(o2[1], tau, atol=1e-12)
        h = 1e-6
        a, b = np.zeros(6), np.zeros(6)
        L.orc_reparam(tau + h, 3.1, _dp(a))
        L.orc_reparam(tau - h, 3.1, _dp(b))
        assert np.isclose(out[2], (a[0] - b[0]) / (2 * h), rtol=1e-6)      # dT/dtau
        assert np.isclose(out[5], (a[3] - b[3]) / (2 * h), rtol=1e-6)      # dq/dvq
        q = out[3]
        assert abs(q) < 3.1
        L.orc_reparam(q, 3.1, _dp(o2))
        assert np.isclose(o2[4], tau, atol=1e-9)                           # invSigmoidC2(sigmoidC2(v)) == v


def test_smooth_l1_is_c1(L):
    mu = 1e-3
    o = np.zeros(2)
    L.orc_smooth_l1(mu * (1 - 1e-9), mu, _dp(o))
    lo = o.copy()
    L.orc_smooth_l1(mu * (1 + 1e-9), mu, _dp(o))
    assert np.allclose(lo, o, rtol=1e-6)
    L.orc_smooth_l1(0.5, mu, _dp(o))
    assert np.isclose(o[0], 0.5 - mu / 2) and o[1] == 1.0
    L.orc_smooth_l1(1e-12, mu, _dp(o))
    assert 0 <= o[0] < 1e-20


# ---- 5. robot model (moma_param.h) ---------------------------------------------------------------------
def test_fk_zero_pose_and_collision_matrix(L):
    pts = np.zeros(48)
    n = L.orc_colli_pts(_dp(np.zeros(10)), _dp(pts))
    assert n == 12
    pts = pts.reshape(12, 4)
    # all spheres on the vertical line x = 0, y = 0.115
    assert np.allclose(pts[:, 0], 0.0, atol=1e-12) and np.allclose(pts[:, 1], 0.115, atol=1e-12)
    z = [0.2200, 0.3100, 0.4115, 0.4840, 0.5640, 0.6675, 0.7260, 0.7960, 0.8775, 0.9515, 1.0215, 1.1215]
    r = [0.06, 0.06, 0.08, 0.055, 0.055, 0.07, 0.055, 0.055, 0.06, 0.055, 0.055, 0.08]
    assert np.allclose(pts[:, 2], z, atol=1e-12) and np.allclose(pts[:, 3], r)
    cm = np.zeros(144, dtype=np.int32)
    L.orc_collision_matrix(cm.ctypes.data_as(C.POINTER(C.c_int)))
    cm = cm.reshape(12, 12)
    # self + adjacent spheres exempt, every other pair checked: 55 pairs
    expect = -np.ones((12, 12), dtype=int)
    for i in range(12):
        for j in range(12):
            if abs(i - j) <= 1:
                expect[i, j] = 1
    assert (cm == expect).all()
    assert (np.triu(cm, 1) == -1).sum() == 55


def test_colli_grads_is_jacobian_transpose(L):
    rng = np.random.default_rng(5)
    for _ in range(3):
        pos = np.concatenate([rng.uniform(-3, 3, 3), rng.uniform(-2, 2, 7)])
        g = rng.standard_normal(36)
        out = np.zeros(10)
        L.orc_colli_grads(_dp(pos), _dp(g), _dp(out))
        h = 1e-6
        for k in range(10):
            pp, pm = pos.copy(), pos.copy()
            pp[k] += h
            pm[k] -= h
            a, b = np.zeros(48), np.zeros(48)
            L.orc_colli_pts(_dp(pp), _dp(a))
            L.orc_colli_pts(_dp(pm), _dp(b))
            dP = (a.reshape(12, 4)[:, :3] - b.reshape(12, 4)[:, :3]) / (2 * h)
            assert np.isclose(out[k], np.sum(dP.reshape(-1) * g), rtol=1e-6, atol=1e-7)


# ---- 6. ESDF queries (grid_map.h:364-509) --------------------------------------------------------------
def test_esdf_interpolation_value_gradient_and_out_of_map():
    rng = np.random.default_rng(6)
    dims = np.array([12, 10, 8], dtype=np.int32)
    res = 0.1
    origin = np.array([-0.6, -0.5, 0.0])
    e2 = rng.uniform(0, 1, dims[0] * dims[1])
    e3 = rng.uniform(0, 1, dims[0] * dims[1] * dims[2])
    m = orc.MapView(origin, res, dims, origin, origin + dims * res, e2, e3)
    o = orc.Oracle(m)
    L = orc.lib()
    d = C.c_double(0)
    g = np.zeros(3)
    # at a cell centre the value is the stored voxel
    ix, iy, iz = 4, 5, 3
    p = origin + (np.array([ix, iy, iz]) + 0.5) * res
    L.orc_esdf_query(o.h, 3, _dp(p), C.byref(d), _dp(g))
    assert np.isclose(d.value, e3[(ix * dims[1] + iy) * dims[2] + iz])
    L.orc_esdf_query(o.h, 2, _dp(p[:2].copy()), C.byref(d), _dp(g))
    assert np.isclose(d.value, e2[ix * dims[1] + iy])
    # gradient = finite difference inside a cell; value continuous across a cell border
    for dim in (2, 3):
        p = (origin + np.array([0.437, 0.512, 0.333]))[:dim].copy()
        L.orc_esdf_query(o.h, dim, _dp(p), C.byref(d), _dp(g))
        g0 = g[:dim].copy()
        for a in range(dim):
            h = 1e-6
            pp, pm = p.copy(), p.copy()
            pp[a] += h
            pm[a] -= h
            dp, dm = C.c_double(0), C.c_double(0)
            gg = np.zeros(3)
            L.orc_esdf_query(o.h, dim, _dp(pp), C.byref(dp), _dp(gg))
            L.orc_esdf_query(o.h, dim, _dp(pm), C.byref(dm), _dp(gg))
            assert np.isclose(g0[a], (dp.value - dm.value) / (2 * h), rtol=1e-6, atol=1e-8)
        border = p.copy()
        border[0] = origin[0] + 5.5 * res  # cell-centre plane = interpolation cell border
        lo, hi = border.copy(), border.copy()
        lo[0] -= 1e-10
        hi[0] += 1e-10
        dl, dh = C.c_double(0), C.c_double(0)
        L.orc_esdf_query(o.h, dim, _dp(lo), C.byref(dl), _dp(g))
        L.orc_esdf_query(o.h, dim, _dp(hi), C.byref(dh), _dp(g))
        assert abs(dl.value - dh.value) < 1e-8
    # out of map (1e-4 margin): distance 0 AND zero gradient (reference quirk iv)
    p = np.array([origin[0] + 5e-5, 0.0, 0.3])
    L.orc_esdf_query(o.h, 3, _dp(p), C.byref(d), _dp(g))
    assert d.value == 0.0 and (g == 0).all()


# ---- 7. cost gradients vs finite differences ------------------------------------------------------------
def _fd_grad(o, stage, x):
    g = np.zeros_like(x)
    for i in range(len(x)):
        h = 1e-6 * max(1.0, abs(x[i]))
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        g[i] = (o.eval(stage, xp)[0] - o.eval(stage, xm)[0]) / (2 * h)
    return g


def test_stage2_gradient_matches_fd_with_exact_chain(cuboids_small):
    """With the mathematically exact Simpson weight on the producing sample the analytic stage-2 gradient equals
    finite differences, which pins MINCO, adjoint, FK, Jacobian-transpose, ESDF and penalty code together.  The
    reference-faithful gradient (default) over-weights that sample and only differs in the tau/theta/arc blocks."""
    cs = cuboids_small
    rng = np.random.default_rng(7)
    for b in (0, 2):
        p = cs["paths"][cs["offs"][b]:cs["offs"][b + 1]]
        o = orc.Oracle(cs["map"])
        n = o.set_init_traj(p)
        N = o.N
        x = o.get_x() + 0.05 * rng.standard_normal(n)
        o.set_alm([0.3, -0.2], [1e4, 1e4])
        gn = _fd_grad(o, 2, x)
        _, g_ref = o.eval(2, x)
        o.set_param("exact_chain", 1)
        _, g_exact = o.eval(2, x)
        scale = np.abs(gn).max()
        assert np.abs(g_exact - gn).max() < 2e-5 * scale
        # joints block is unaffected by the quirk
        assert np.abs(g_ref[3 * N - 1:] - gn[3 * N - 1:]).max() < 2e-5 * scale


def test_stage1_gradient_quirk(cuboids_small):
    """Stage 1 drops the path-tracking gradient of the piece's own samples (head(i*(2K+1)), moma_traj_opt.cpp:
    1175-1176): the joint block still matches finite differences, the theta/arc/tau blocks do not."""
    cs = cuboids_small
    p = cs["paths"][cs["offs"][1]:cs["offs"][2]]
    o = orc.Oracle(cs["map"])
    n = o.set_init_traj(p)
    N = o.N
    x = o.get_x()
    gn = _fd_grad(o, 1, x)
    _, g = o.eval(1, x)
    scale = np.abs(gn).max()
    assert np.abs(g[3 * N - 1:] - gn[3 * N - 1:]).max() < 1e-5 * scale
    assert np.abs(g[:3 * N - 1] - gn[:3 * N - 1]).max() > 1e-2 * scale


# ---- 8. L-BFGS (lbfgs.hpp) -------------------------------------------------------------------------------
def test_lbfgs_on_textbook_functions(L):
    L.orc_lbfgs_test.argtypes = [C.c_int, C.c_int, C.c_int, c_dp, c_dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for kind, n, stage in ((1, 20, 2), (0, 10, 2), (1, 20, 1)):
        x = np.full(n, -1.2) if kind == 0 else np.zeros(n)
        if kind == 0:
            x[1::2] = 1.0
        f = C.c_double(0)
        it = C.c_int(0)
        ev = C.c_int(0)
        ret = L.orc_lbfgs_test(kind, n, stage, _dp(x), C.byref(f), C.byref(it), C.byref(ev))
        assert ret in (0, 1)  # CONVERGENCE or STOP (past/delta test)
        if stage == 2:
            assert f.value < 1e-3 and np.allclose(x, 1.0, atol=0.05)
        else:  # stage-1 parameters stop early by design (delta 1e-2, past 2)
            assert it.value < 40
        assert ev.value >= it.value
