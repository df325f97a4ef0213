"""Several wavefronts per trajectory (topay_amd/csrc/topay_eval_mw.h): the workgroup of 2 or 4 waves that solves the long
candidates (launch classes N <= 42 / 64 / 170).

  * An evaluation is order-identical whatever the number of waves: for every N <= 64 the 2- and 4-wave kernels must
    return, bit for bit, what the one-wave kernel returns (cost, gradient, end-point error) -- both stages, ordinary
    points and the rare paths (joint velocity / acceleration rows, mean-time band, non-finite cost).
  * N = 65..170 exists only on four waves: per-evaluation parity against the
    oracle, capped solves against the oracle, and whole solves against the oracle's solver logic in the device's
    vector order (64 x waves threads, topay_class_of) fed with the device's evaluations -- bit for bit.
The CPU half runs the kernel sources in the lane emulator (its waves really run out of step between workgroup
barriers); the GPU half is the same on the MI355X, plus GPU == emulator.
"""
import numpy as np
import pytest

from conftest import EMU_LIB, serpentine_path, set_map
from oracle import oracle as orc
from topay_amd import api


def _points(o, n, N, rng):
    x0 = o.get_x().copy()
    yield x0
    yield x0 + 0.04 * rng.standard_normal(n)
    x = x0.copy()                       # rare paths: joint velocity / acceleration limits, mean-time band, folded arm
    x[:N] -= 1.6
    x[N - 1] += 2.5
    x[3 * N - 1:] += np.tile([0.0, 1.5, 0.0, 2.4, 0.0, 1.9, 0.0], N - 1)
    yield x


def _check_eval(opt, cs, paths, rng, stages=(1, 2), vs_oracle=True):
    """Every candidate of `paths`: all wave counts that hold it agree bit for bit, and agree with the oracle."""
    Ns = opt.n_pieces()
    for b, path in enumerate(paths):
        N = int(Ns[b])
        if N == 0:
            continue
        o = orc.Oracle(cs["map"])
        n = o.set_init_traj(path)
        assert o.N == N and np.allclose(opt.get_x(b), o.get_x(), rtol=0, atol=1e-12)
        waves = (1, 2, 4) if N <= 64 else (4,)
        for x in _points(o, n, N, rng):
            for stage in stages:
                lam, rho = [0.3, -0.2], [1e4, 2e4]
                res = [opt.eval(stage, b, x, lam, rho, waves=w) for w in waves]
                for r in res[1:]:
                    assert r[0] == res[0][0] and (r[1] == res[0][1]).all() and (r[2] == res[0][2]).all(), (N, stage, waves)
                fd, gd, ed = opt.eval(stage, b, x, lam, rho)            # the candidate's own launch class
                assert fd == res[0][0] and (gd == res[0][1]).all() and (ed == res[0][2]).all()
                if vs_oracle:
                    o.set_alm(lam, rho)
                    f, g = o.eval(stage, x)
                    assert abs(f - fd) <= 1e-11 * abs(f), (N, stage)
                    assert np.abs(g - gd).max() <= 1e-10 * np.abs(g).max(), (N, stage)


def _capped_solve_vs_device_order(lib, cs, paths, s1_it, s2_it, outer):
    lens = np.array([len(p) for p in paths], dtype=np.int32)
    p = api.default_params(api.load(lib))
    p.s1_lbfgs.max_iterations = s1_it
    p.s2_lbfgs.max_iterations = s2_it
    p.alm_max_outer = outer
    opt = api.MomaTrajOptBatch(params=p, device=0, lib_path=lib)
    set_map(opt, cs["world"])
    ok = opt.optimizeTraj(lens, np.concatenate(paths))
    st, alm = opt.stats(), opt.alm_state()
    xs = [opt.get_x(b) for b in range(len(paths))]
    ev = api.MomaTrajOptBatch(params=p, device=0, lib_path=lib)
    set_map(ev, cs["world"])
    ev.set_init_traj(lens, np.concatenate(paths))
    for b, path in enumerate(paths):
        o = orc.Oracle(cs["map"])
        o.set_param("s1_max_iterations", s1_it)
        o.set_param("s2_max_iterations", s2_it)
        o.set_param("alm_max_outer", outer)
        o.set_init_traj(path)
        nw, epl, _ = opt.class_of(o.N)
        okh = o.optimize_device_order(lambda stage, xx, lam, rho: ev.eval(stage, b, xx, lam, rho), epl=epl, nw=nw)
        so = o.stats()
        assert okh == bool(ok[b]) and [so[k] for k in api.STAT_KEYS] == list(st[b]), (b, o.N, nw)
        assert (o.get_x() == xs[b]).all() and (o.alm_state() == alm[b]).all(), (b, o.N, nw)
    return opt


def test_class_table():
    L = api.load(EMU_LIB)
    opt = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    assert [opt.class_of(N)[0] for N in (3, 10, 11, 21, 32)] == [1] * 5          # the common classes: one wave per trajectory
    # the long classes: evaluations on four waves, the SOLVER on the first of them (round 5) -- class_of describes the solver
    assert opt.class_of(170)[0] == 1 and opt.class_of(65)[0] == 1 and opt.class_of(33)[0] == 1
    with pytest.raises(api.TopayError):
        opt.class_of(171)
    for N in (3, 10, 11, 15, 16, 21, 22, 32, 33, 42, 43, 64, 65, 128, 129, 170):
        w, epl, k = opt.class_of(N)
        assert 10 * N - 8 <= 64 * w * epl                                          # the solver's vectors fit its threads


def test_multiwave_evaluation_is_order_identical_on_cpu(cuboids_small):
    """Kernel sources in the lane emulator: candidates of 4..11 pieces and a 33-piece one through the 1-, 2- and 4-wave
    kernels (bit-identical), a 95-piece, a 126-piece and a 170-piece one through four waves
    against the oracle."""
    cs = cuboids_small
    rng = np.random.default_rng(5)
    emu = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(emu, cs["world"])
    small = [cs["paths"][cs["offs"][b]:cs["offs"][b + 1]] for b in (0, 3, 5)]
    emu.set_init_traj(np.array([len(p) for p in small], dtype=np.int32), np.concatenate(small))
    _check_eval(emu, cs, small, rng)
    long_ = [serpentine_path(L) for L in (34.0, 99.0, 131.5, 177.0)]
    emu.set_init_traj(np.array([len(p) for p in long_], dtype=np.int32), np.concatenate(long_))
    assert list(emu.n_pieces()) == [33, 95, 126, 170]
    _check_eval(emu, cs, long_, rng, stages=(2,))


def test_multiwave_solver_equals_oracle_in_device_order_on_cpu(cuboids_small):
    """Capped solves of a 33-piece and a 95-piece candidate (one-wave solver on the first of four waves, 10 / 20 vector
    elements per lane) in the lane emulator against the oracle's solver logic with its vector arithmetic in that order:
    iterate, multipliers and counters bit for bit."""
    _capped_solve_vs_device_order(EMU_LIB, cuboids_small, [serpentine_path(34.0), serpentine_path(99.0)], 6, 4, 2)


@pytest.mark.gpu
def test_multiwave_evaluation_is_order_identical_on_gpu(cuboids_small):
    cs = cuboids_small
    rng = np.random.default_rng(6)
    gpu = api.MomaTrajOptBatch(device=0)
    set_map(gpu, cs["world"])
    small = [cs["paths"][cs["offs"][b]:cs["offs"][b + 1]] for b in range(len(cs["lens"]))]
    gpu.set_init_traj(cs["lens"], cs["paths"])
    _check_eval(gpu, cs, small, rng)
    long_ = [serpentine_path(L) for L in (20.0, 27.0, 34.0, 44.0, 50.0, 66.0, 67.0, 99.0, 120.0, 131.5, 133.0, 150.0, 177.0, 178.0)]
    gpu.set_init_traj(np.array([len(p) for p in long_], dtype=np.int32), np.concatenate(long_))
    N = gpu.n_pieces()
    assert N[2] == 33 and N[5] == 64 and N[6] == 65 and N[10] == 128 and N[11] == 144 and N[12] == 170 and N[13] == 0     # 171 pieces: refused
    _check_eval(gpu, cs, long_, rng)
    # GPU == lane emulator, every bit, on a four-wave evaluation
    emu = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(emu, cs["world"])
    emu.set_init_traj(np.array([len(long_[9])], dtype=np.int32), long_[9])
    o = orc.Oracle(cs["map"])
    n = o.set_init_traj(long_[9])
    x = o.get_x() + 0.03 * rng.standard_normal(n)
    for stage in (1, 2):
        a, b_ = gpu.eval(stage, 9, x, [0.1, 0.2], [1e4, 3e4]), emu.eval(stage, 0, x, [0.1, 0.2], [1e4, 3e4])
        assert a[0] == b_[0] and (a[1] == b_[1]).all() and (a[2] == b_[2]).all()


@pytest.mark.gpu
def test_long_candidates_up_to_170_pieces_on_gpu(cuboids_small):
    """N = 96, 128 and 170 (the reference has no bound on the pieces, moma_traj_opt.cpp:245, 300-321; this build's is 170):
    capped solves against the oracle -- counters identical, iterate to 1e-7, getTraj coefficients -- and whole
    solves of 33 / 64 / 96 / 128 / 170 pieces against the oracle's solver logic in the device's vector order, bit for bit."""
    cs = cuboids_small
    paths = [serpentine_path(L) for L in (100.0, 133.0, 177.0, 178.0)]
    lens = np.array([len(p) for p in paths], dtype=np.int32)
    p = api.default_params()
    p.s2_lbfgs.max_iterations = 8
    p.alm_max_outer = 1
    cap = api.MomaTrajOptBatch(params=p, device=0)
    set_map(cap, cs["world"])
    ok = cap.optimizeTraj(lens, np.concatenate(paths))
    N = cap.n_pieces()
    assert list(N) == [96, 128, 170, 0] and not ok[3] and np.isnan(cap.traj_cost[3])
    st = cap.stats()
    for k in range(3):
        o = orc.Oracle(cs["map"])
        o.set_param("s2_max_iterations", 8)
        o.set_param("alm_max_outer", 1)
        o.set_init_traj(paths[k])
        o.optimize()
        so = o.stats()
        assert list(st[k]) == [so[key] for key in api.STAT_KEYS], (k, list(st[k]), so)
        # Eight stage-2 iterations amplify the 1e-14 per-evaluation differences, the more the longer the candidate: the yardstick is the
        # oracle against ITSELF with one ulp added to one input coordinate (same code, same machine: 1.6e-9 / 9.0e-8 / 1.5e-4 at
        # 96 / 128 / 170 pieces) -- the device may differ from the oracle by twenty times that, at least 1e-7.  Bit-exactness in the
        # device's own order follows below.
        pp = paths[k].copy()
        pp[len(pp) // 2, 0] = np.nextafter(pp[len(pp) // 2, 0], 1e9)
        o1 = orc.Oracle(cs["map"])
        o1.set_param("s2_max_iterations", 8)
        o1.set_param("alm_max_outer", 1)
        o1.set_init_traj(pp)
        o1.optimize()
        own = np.abs(o1.get_x() - o.get_x()).max()
        dx = np.abs(cap.get_x(k) - o.get_x()).max()
        print(f"N {N[k]}: capped solve, iterate against the oracle {dx:.2e}; the oracle against itself with one ulp on one input {own:.2e}")
        bound = max(1e-7, 20.0 * own)
        assert dx <= bound, (N[k], dx, own)
        tr = cap.getTraj(k)
        d, c, kn = o.get_traj()
        assert np.abs(tr["durations"] - d).max() <= 10.0 * bound and np.abs(tr["knots_xy"] - kn).max() <= 10.0 * bound
        assert np.abs(tr["coeffs"] - c).max() <= max(1e-5, 100.0 * bound) * np.abs(c).max()
    # whole solves (to the solver's own stop), device order
    full = [serpentine_path(L) for L in (34.0, 66.0, 100.0, 133.0, 177.0)]
    opt = _capped_solve_vs_device_order(None, cs, full, 8000, 8000, 30)
    assert list(opt.n_pieces()) == [33, 64, 96, 128, 170]
    assert opt.stats()[:, 4].min() > 30      # real stage-2 runs


@pytest.mark.gpu
def test_multiwave_capped_solve_is_bit_identical_to_emulator(cuboids_small):
    """A 33-piece candidate on the several-waves kernel of its class: stage 1 capped at five, stage 2 at three iterations, GPU vs lane
    emulator -- trace of every evaluated cost, counters, iterate, coefficients."""
    cs = cuboids_small
    path = serpentine_path(34.0)
    p = api.default_params()
    p.s1_lbfgs.max_iterations = 5
    p.s2_lbfgs.max_iterations = 3
    p.alm_max_outer = 1
    res = []
    for lib in (None, EMU_LIB):
        o2 = api.MomaTrajOptBatch(params=p, device=0, lib_path=lib)
        set_map(o2, cs["world"])
        o2.set_init_traj(np.array([len(path)], dtype=np.int32), path)
        assert o2.class_of(int(o2.n_pieces()[0]))[1] >= 10   # (a long class: ten vector elements per lane of the one-wave solver)
        o2.set_trace(64)
        o2.optimize()
        res.append((o2.stats(), o2.get_trace(0), o2.get_x(0), o2.getTraj(0)["coeffs"]))
    g, e = res
    assert (g[0] == e[0]).all() and (g[1] == e[1]).all() and (g[2] == e[2]).all() and (g[3] == e[3]).all()


def _helper_wave_solves(lib, cs, paths, s1_it, s2_it, outer):
    """The same capped solves through the default kernels and through the helper-wave kernels (topay_set_latency_mode 2)."""
    lens = np.array([len(p) for p in paths], dtype=np.int32)
    p = api.default_params(api.load(lib))
    p.s1_lbfgs.max_iterations = s1_it
    p.s2_lbfgs.max_iterations = s2_it
    p.alm_max_outer = outer
    res = []
    for mode in (0, 2):
        opt = api.MomaTrajOptBatch(params=p, device=0, lib_path=lib)
        set_map(opt, cs["world"])
        opt.set_init_traj(lens, np.concatenate(paths))
        opt.set_latency_mode(mode)
        opt.set_trace(64)
        ok = opt.optimize()
        assert (opt.last_helper_launches() > 0) == (mode == 2)
        res.append(dict(ok=ok, st=opt.stats(), alm=opt.alm_state(), cost=opt.traj_cost.copy(),
                        x=[opt.get_x(b) for b in range(len(paths))], tr=[opt.get_trace(b) for b in range(len(paths))],
                        co=[opt.getTraj(b)["coeffs"] for b in range(len(paths)) if ok[b]], N=opt.n_pieces()))
    a, h = res
    assert (a["ok"] == h["ok"]).all() and (a["st"] == h["st"]).all() and (a["alm"] == h["alm"]).all()
    assert (np.nan_to_num(a["cost"]) == np.nan_to_num(h["cost"])).all()
    for b in range(len(paths)):
        assert (a["x"][b] == h["x"][b]).all() and (a["tr"][b] == h["tr"][b]).all(), (b, int(a["N"][b]))
    for ca, ch in zip(a["co"], h["co"]):
        assert (ca == ch).all()
    return a["N"]


def test_helper_wave_kernels_solve_to_the_same_bits_on_cpu(cuboids_small):
    """topay_set_latency_mode: four-wave workgroups whose extra waves only join the evaluations.  Candidates of every one-wave
    class (4..11, 14, 19 and 26 pieces; the 33-piece one stays on its own four-wave kernel) through capped solves in the lane
    emulator: every evaluated cost, the counters, the iterate, the ALM state and the coefficients equal the default kernels'."""
    cs = cuboids_small
    paths = [cs["paths"][cs["offs"][b]:cs["offs"][b + 1]] for b in range(len(cs["lens"]))] + [serpentine_path(L) for L in (14.0, 19.5, 27.0, 34.0)]
    N = _helper_wave_solves(EMU_LIB, cs, paths, 4, 3, 1)
    assert N.max() >= 33 and (N[6:9] > np.array([10, 15, 21])).all() and (N[6:9] <= np.array([15, 21, 32])).all(), N


@pytest.mark.gpu
def test_helper_wave_kernels_solve_to_the_same_bits_on_gpu(cuboids_small):
    """The same on the MI355X, to convergence (uncapped): the six candidates of the small fixture + one per one-wave class."""
    cs = cuboids_small
    paths = [cs["paths"][cs["offs"][b]:cs["offs"][b + 1]] for b in range(len(cs["lens"]))] + [serpentine_path(L) for L in (14.0, 19.5, 27.0, 34.0)]
    lens = np.array([len(p) for p in paths], dtype=np.int32)
    res = []
    for mode in (0, 1):
        opt = api.MomaTrajOptBatch(device=0)
        set_map(opt, cs["world"])
        opt.set_init_traj(lens, np.concatenate(paths))
        opt.set_latency_mode(mode)      # mode 1: this batch is small enough
        ok = opt.optimize()
        assert (opt.last_helper_launches() > 0) == (mode == 1)
        res.append((ok, opt.stats(), np.nan_to_num(opt.traj_cost), [opt.get_x(b) for b in range(len(paths))], opt.last_kernel_ms()[0]))
    a, h = res
    assert (a[0] == h[0]).all() and (a[1] == h[1]).all() and (a[2] == h[2]).all()
    for xa, xh in zip(a[3], h[3]):
        assert (xa == xh).all()
    print("solve of %d candidates: %.1f ms on the default kernels, %.1f ms with helper waves" % (len(paths), a[4], h[4]))


def test_rare_rows_of_a_piece_that_straddles_two_passes_keep_the_one_wave_order(cuboids_small):
    """Regression (round 4): the 637th evaluation of the 26-piece serpentine's solve -- joint velocity / acceleration rows
    flagged in a pass whose last piece continues in the next pass of the same round.  The several-waves gradient phase added a
    round's order-0 joint rows first and its rare rows afterwards; the one-wave phase does both pass by pass, so the two
    disagreed by 1.5e-10 in such a piece's joint rows (found by the helper-wave kernels, whose solves must reproduce the
    one-wave kernels' bit for bit).  tests/golden/order_identity_n26.npz: that decision vector and ALM state."""
    import os
    cs = cuboids_small
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "order_identity_n26.npz"))
    path = serpentine_path(27.0)
    emu = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(emu, cs["world"])
    emu.set_init_traj(np.array([len(path)], dtype=np.int32), path)
    assert int(emu.n_pieces()[0]) == 26 and len(fx["x"]) == 10 * 26 - 8
    r = {w: emu.eval(int(fx["stage"]), 0, fx["x"], fx["lam"], fx["rho"], waves=w) for w in (1, 2, 4)}
    for w in (2, 4):
        assert r[w][0] == r[1][0] and (r[w][1] == r[1][1]).all() and (r[w][2] == r[1][2]).all(), w
    o = orc.Oracle(cs["map"])
    o.set_init_traj(path)
    o.set_alm(fx["lam"], fx["rho"])
    f, g = o.eval(int(fx["stage"]), fx["x"])
    assert abs(f - r[1][0]) <= 1e-11 * abs(f) and np.abs(g - r[1][1]).max() <= 1e-10 * np.abs(g).max()


@pytest.mark.gpu
def test_helper_wave_kernels_on_512_benchmark_candidates():
    """64 scenarios x 8 candidates of the headline generator (4 to ~45 pieces, one map per scenario), solved to convergence by the
    default kernels and by the helper-wave kernels (mode 2: the batch is larger than the automatic rule allows): success, counters,
    cost, ALM state, the iterate and the feasibility gate's flags of every candidate are equal."""
    from harness import workload as wl
    tb = wl.TablesBatch(64, 8, base_seed=42, nthreads=8)
    worlds = [tb.world(s_) for s_ in tb.scenarios]
    slot = {s_: k for k, s_ in enumerate(tb.scenarios)}
    map_ids = np.array([slot[s_] for s_ in tb.scen], dtype=np.int32)
    res = []
    for mode in (0, 2):
        gpu = api.MomaTrajOptBatch(device=0)
        w0 = worlds[0]
        gpu.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, np.stack([w.occ2d for w in worlds]), np.stack([w.occ3d for w in worlds]))
        gpu.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
        gpu.set_latency_mode(mode)
        ok = gpu.optimize()
        assert (gpu.last_helper_launches() > 0) == (mode == 2)
        res.append((ok, gpu.stats(), np.nan_to_num(gpu.traj_cost), gpu.alm_state(), gpu.check_feasible(),
                    [gpu.get_x(b) for b in range(len(tb.lens))], gpu.n_pieces()))
        gpu.close()
    a, h = res
    assert a[6].max() > 32 and (a[6] <= 10).sum() > 100 and ((a[6] > 21) & (a[6] <= 32)).sum() > 3, np.bincount(a[6])
    for k in range(5):
        assert (a[k] == h[k]).all(), k
    for xa, xh in zip(a[5], h[5]):
        assert (xa == xh).all()
