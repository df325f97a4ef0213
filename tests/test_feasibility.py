"""Feasibility gate (printConstraintsSituations / checkFeasible, moma_traj_opt.h:948-1204) and the MomaTraj playback
it samples through (moma_traj_opt.h:26-137): oracle self-consistency, kernel sources (CPU lane emulator) vs oracle,
and — on the GPU — HIP vs emulator (bit-identical) and vs oracle."""
import numpy as np
import pytest

from conftest import EMU_LIB, set_map
from oracle import oracle as orc
from topay_amd import api
from harness import workload as wl


def _oracle_gate(m, path, x, alm):
    o = orc.Oracle(m)
    o.set_init_traj(path)
    o.set_alm(alm[:2], alm[2:])
    o.eval(2, x)                      # loads the spline of x into the oracle
    return o, o.check_feasible()


@pytest.fixture(scope="module")
def emu_partly_solved(cuboids_small):
    """Four candidates taken part of the way by the kernel sources in the lane emulator.  The gate and the playback
    are functions of the returned spline, whatever its state of convergence, and a full emulated solve takes most of a
    minute per candidate: the inner iterations are capped (the one full solve is in the too-fast test below)."""
    cs = cuboids_small
    p = api.default_params()
    p.s2_lbfgs.max_iterations = 40
    p.alm_max_outer = 2
    emu = api.MomaTrajOptBatch(params=p, lib_path=EMU_LIB)
    set_map(emu, cs["world"])
    ok = emu.optimizeTraj(cs["lens"][:4], cs["paths"][:cs["offs"][4]])
    # a candidate may end in one of the reference's L-BFGS error returns (the oracle does too, for about 1 % of the
    # benchmark candidates): it has no trajectory to gate
    solved = [int(b) for b in np.nonzero(ok)[0]]
    assert len(solved) >= 2
    return emu, solved


def test_oracle_playback_is_consistent(cuboids_small):
    """MomaTraj::getState at t = 0 / T reproduces the start state / the Simpson-integrated end knot of getTraj, joints
    and yaw follow the polynomial, and the report's extremes bound the sampled values."""
    cs = cuboids_small
    o = orc.Oracle(cs["map"])
    path = cs["paths"][cs["offs"][1]:cs["offs"][2]]
    o.set_init_traj(path)
    assert o.optimize()
    d, c, kn = o.get_traj()
    s0, sT = o.traj_state(0.0), o.traj_state(d.sum())
    assert np.allclose(s0[:3], path[0][:3], atol=1e-12) and np.allclose(s0[3:], path[0][3:], atol=1e-9)
    assert np.linalg.norm(sT[:2] - kn[-1]) < 1e-4      # 0.025 s panels vs the optimiser's T/12 panels
    assert np.allclose(sT[3:], path[-1][3:], atol=1e-6)
    f, st, rep = o.check_feasible()
    assert abs(rep[0]) <= 1.01 * 1.0 + 0.05 and rep[25] > 0.3 and (rep[26:] > 0).all()
    mid = o.traj_state(0.37 * d.sum())
    assert (np.abs(mid[3:]) <= np.abs(rep[4:11]) + 1e-12).all()


def test_gate_kernel_matches_oracle_on_cpu(cuboids_small, emu_partly_solved):
    cs, (emu, solved) = cuboids_small, emu_partly_solved
    paths = cs["paths"]
    f, st, rep = emu.check_feasible(report=True)
    assert (emu.check_feasible() == f).all()
    alm = emu.alm_state()
    for b in solved:
        _, (fo, so, ro) = _oracle_gate(cs["map"], paths[cs["offs"][b]:cs["offs"][b + 1]], emu.get_x(b), alm[b])
        assert fo == f[b] and so == st[b]
        # extremes: 1e-10 relative (libm vs deterministic sin/cos, scan vs running sum in car_seq)
        assert np.allclose(np.abs(ro), rep[b], rtol=1e-10, atol=1e-12)


def test_gate_rejects_a_too_fast_trajectory(cuboids_small):
    """Tighten the limits instead of editing the trajectory: with max_v scaled down the same candidate must fail on
    the velocity test and only there (the gate reads the limits from the parameter block)."""
    cs = cuboids_small
    lens, paths = cs["lens"][:1], cs["paths"][:cs["offs"][1]]
    p = api.default_params()
    emu = api.MomaTrajOptBatch(params=p, lib_path=EMU_LIB)
    set_map(emu, cs["world"])
    emu.optimizeTraj(lens, paths)
    f, st, rep = emu.check_feasible(report=True)
    assert f[0] and rep[0, 0] > 0.5
    p.s2_lbfgs.max_iterations = 1
    p.alm_max_outer = 1                 # stop early: the half-optimised trajectory still violates its limits
    emu2 = api.MomaTrajOptBatch(params=p, lib_path=EMU_LIB)
    set_map(emu2, cs["world"])
    emu2.optimizeTraj(lens, paths)
    f2, st2, rep2 = emu2.check_feasible(report=True)
    lim = np.array([1.0, 0.8, 1.25, 1.0])
    assert f2[0] == bool((rep2[0, :4] <= 1.01 * lim).all() and (rep2[0, 4:11] <= 1.01 * np.array([3.1, 2.26, 3.1, 2.355, 3.1, 2.23, 6.28])).all()
                         and (rep2[0, 11:18] <= 1.01 * 2.35).all() and (rep2[0, 18:25] <= 1.01 * 6.28).all() and rep2[0, 25] >= 0.99 * 0.4)


@pytest.mark.gpu
def test_gate_on_gpu_matches_emulator_and_oracle():
    world, start, goal, lens, paths = wl.tables_scenario(3, 8)
    offs = np.concatenate([[0], np.cumsum(lens)])
    res = {}
    for name, lib in (("gpu", None), ("emu", EMU_LIB)):
        opt = api.MomaTrajOptBatch(device=0, lib_path=lib)
        set_map(opt, world)
        nb = len(lens) if name == "gpu" else 2
        opt.optimizeTraj(lens[:nb], paths[:offs[nb]])
        res[name] = (opt, opt.check_feasible(report=True))
    (fg, sg, rg), (fe, se, re_) = res["gpu"][1], res["emu"][1]
    assert (fg[:2] == fe).all() and (sg[:2] == se).all() and (rg[:2] == re_).all()   # bit-identical to the CPU execution
    m = orc.MapView(world.origin, world.res, world.dims, world.min_b, world.max_b, world.esdf2d, world.esdf3d)
    gpu = res["gpu"][0]
    alm = gpu.alm_state()
    for b in range(len(lens)):
        if not np.isfinite(rg[b]).all():
            continue
        _, (fo, so, ro) = _oracle_gate(m, paths[offs[b]:offs[b + 1]], gpu.get_x(b), alm[b])
        assert fo == fg[b] and so == sg[b]
        assert np.allclose(np.abs(ro), rg[b], rtol=1e-9, atol=1e-11)


def test_playback_matches_oracle(cuboids_small, emu_partly_solved):
    """car_seq and getState of the kernel sources (CPU lane emulator) against the oracle's MomaTraj restatement."""
    cs, (emu, solved) = cuboids_small, emu_partly_solved
    paths = cs["paths"]
    alm = emu.alm_state()
    for b in solved[:2]:
        o, _ = _oracle_gate(cs["map"], paths[cs["offs"][b]:cs["offs"][b + 1]], emu.get_x(b), alm[b])
        T = emu.total_durations()[b]
        times = np.concatenate([[0.0, T, T + 1.0, -0.5], np.linspace(0, T, 37)])
        st, seq = emu.playback(b, times)
        seq_o = o.car_seq()
        assert len(seq) == len(seq_o) and np.allclose(seq, seq_o, rtol=0, atol=1e-11)
        for k, t in enumerate(times):
            assert np.allclose(st[k], o.traj_state(float(t)), rtol=0, atol=1e-11)


def test_mesh_poses_and_mesh_traj_match_oracle(cuboids_small, emu_partly_solved):
    """MomaParam::getMeshPose (moma_param.h:724-790) and Planner::toMeshMsg (planner.cpp:2003-2056) of the kernel sources
    against the oracle's restatement: random states (joints beyond their limits included: the reference clamps them),
    then the 1000-step message of two trajectories -- same number of states, poses, yaws and arc lengths."""
    cs, (emu, solved) = cuboids_small, emu_partly_solved
    rng = np.random.default_rng(0)
    st = np.concatenate([rng.uniform(-8, 8, (64, 2)), rng.uniform(-4, 4, (64, 1)), rng.uniform(-7, 7, (64, 7))], axis=1)
    st[0] = 0.0
    o = orc.Oracle(cs["map"])
    pe = emu.mesh_poses(st)
    po = np.array([o.mesh_pose(s) for s in st])
    assert np.abs(pe - po).max() < 1e-13
    assert np.allclose(np.linalg.norm(pe[:, :10, 3:], axis=2), 1.0, atol=1e-6)       # relative_R's 0.7071068 is not exactly orthonormal
    assert (pe[:, 10, 3:] == 0).all() and (pe[:, 9] == pe[:, 8]).all()
    assert np.allclose(pe[0, :, :3], [[0, 0, 0.0775], [0, 0.115, 0.0935], [0, 0.115, 0.334], [0, 0.115, 0.334], [0, 0.115, 0.59],
                                      [0, 0.115, 0.59], [0, 0.115, 0.8], [0, 0.115, 0.8], [0, 0.115, 0.944], [0, 0.115, 0.944],
                                      [0, 0.115, 1.1215]], atol=1e-5)   # zero pose: links stacked along z (offsets are +-1.5708, not pi/2)
    alm = emu.alm_state()
    for b in solved[:2]:
        o, _ = _oracle_gate(cs["map"], cs["paths"][cs["offs"][b]:cs["offs"][b + 1]], emu.get_x(b), alm[b])
        P, Y, A = o.mesh_traj(1000)
        Pe, Ye, Ae = emu.mesh_traj(b, 1000)
        assert len(Ye) == len(Y) and len(Y) in (1000, 1001)
        assert np.abs(P - Pe).max() < 1e-11 and np.abs(Y - Ye).max() < 1e-11 and np.abs(A - Ae).max() < 1e-11
        assert (np.diff(Ae) >= 0).all() and Ae[0] == 0.0


@pytest.mark.gpu
def test_mesh_traj_on_gpu_is_bit_identical_to_emulator(cuboids_small):
    cs = cuboids_small
    lens, paths = cs["lens"][:1], cs["paths"][:cs["offs"][1]]
    out = {}
    for name, lib in (("gpu", None), ("emu", EMU_LIB)):
        p = api.default_params(api.load(lib))
        p.s2_lbfgs.max_iterations = 20
        p.alm_max_outer = 1
        opt = api.MomaTrajOptBatch(params=p, device=0, lib_path=lib)
        set_map(opt, cs["world"])
        opt.optimizeTraj(lens, paths)
        out[name] = opt.mesh_traj(0, 1000)
    for a, b in zip(out["gpu"], out["emu"]):
        assert a.shape == b.shape and (a == b).all()


@pytest.mark.gpu
def test_playback_on_gpu_is_bit_identical_to_emulator(cuboids_small):
    cs = cuboids_small
    lens, paths = cs["lens"][:1], cs["paths"][:cs["offs"][1]]
    out = {}
    for name, lib in (("gpu", None), ("emu", EMU_LIB)):
        opt = api.MomaTrajOptBatch(device=0, lib_path=lib)
        set_map(opt, cs["world"])
        opt.optimizeTraj(lens, paths)
        T = opt.total_durations()[0]
        out[name] = opt.playback(0, np.linspace(-0.2, T + 0.2, 211))
    assert (out["gpu"][0] == out["emu"][0]).all() and (out["gpu"][1] == out["emu"][1]).all()


def test_gate_with_a_short_history_block_falls_back_to_the_separate_kernel(cuboids_small):
    """The in-solve gate borrows the candidate's L-BFGS history block as scratch.  With mem_size = 8 the block cannot hold
    the panels and sample times of the returned trajectory: the solving wave must NOT gate a truncated sweep (round 3 did,
    silently: violations in the tail went unsampled) -- it leaves the candidate to the separate kernel, whose scratch is
    sized from the trajectory, and the verdicts equal those of the same splines gated through that kernel alone."""
    cs = cuboids_small
    p = api.default_params()
    p.s1_lbfgs.mem_size = 8
    p.s2_lbfgs.mem_size = 8
    p.s2_lbfgs.max_iterations = 30
    p.alm_max_outer = 1
    emu = api.MomaTrajOptBatch(params=p, lib_path=EMU_LIB)
    set_map(emu, cs["world"])
    nb = 3
    ok = emu.optimizeTraj(cs["lens"][:nb], cs["paths"][:cs["offs"][nb]])
    feas, strict, rep = emu.check_feasible(report=True)
    # reference route: the same splines loaded into a second context (a loaded trajectory is always gated by the separate kernel)
    other = api.MomaTrajOptBatch(params=p, lib_path=EMU_LIB)
    set_map(other, cs["world"])
    other.set_init_traj(cs["lens"][:nb], cs["paths"][:cs["offs"][nb]])
    alm = emu.alm_state()
    for b in range(nb):
        other.load_solution(b, emu.get_x(b), alm[b][:2], alm[b][2:])
    f2, s2, r2 = other.check_feasible(report=True)
    for b in range(nb):
        if not ok[b]:
            continue
        assert feas[b] == f2[b] and strict[b] == s2[b]
        assert np.array_equal(rep[b], r2[b])
        assert np.isfinite(rep[b]).all()      # sampled, not skipped
    emu.close(); other.close()
