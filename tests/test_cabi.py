"""The drop-in boundary: the hipcc-built library exists, exports every symbol include/topay.h declares, its
parameter block has the reference defaults, and without a usable HIP device it fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import __graft_entry__ as entry
from topay_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    entry.build_hip()  # hipcc cross-compiles for gfx950 without a GPU
    return api.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "topay.h")).read()
    # function declarations look like "topay_status topay_xxx(" / "void topay_destroy(" / "const char* topay_last_error("
    return sorted(set(re.findall(r"^(?:topay_status|void|const char\*)\s+(topay_[a-z0-9_]+)\s*\(", text, flags=re.M)))


def test_library_exports_every_declared_symbol(hip_lib):
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/topay.h but not exported"


def test_default_params_are_the_reference_values(hip_lib):
    p = api.default_params(hip_lib)
    assert p.int_K == 12 and p.min_piece_num == 3 and p.relu_mu == 1e-3 and p.sample_interval == 1.5
    assert list(p.energy_weights) == [0.33] + [1.0] * 8
    assert (p.s1_time_weight, p.s1_path_pos_weight, p.s1_normal_past, p.s1_shot_path_past) == (20.0, 2e5, 2, 8)
    assert (p.s1_lbfgs.mem_size, p.s1_lbfgs.delta, p.s1_lbfgs.max_iterations) == (256, 1e-2, 8000)
    assert (p.s2_lbfgs.mem_size, p.s2_lbfgs.past, p.s2_lbfgs.delta, p.s2_lbfgs.min_step) == (256, 3, 1e-4, 1e-32)
    assert (p.s2_collision_weight, p.s2_mean_time_weight, p.s2_time_weight) == (5e5, 5000.0, 50.0)
    assert list(p.alm_init_rho) == [1e4, 1e4] and p.alm_tolerance == 0.01
    assert (p.max_v, p.max_a, p.max_w, p.max_dw) == (1.0, 0.8, 1.25, 1.0)
    assert min(r for r in p.colli_point_radius if r > 0) == 0.055  # moma_param.h:110-112 floor


def test_unsupported_parameters_are_rejected(hip_lib):
    p = api.default_params(hip_lib)
    p.int_K = 32
    h = C.c_void_p()
    assert hip_lib.topay_create(C.byref(p), 0, C.byref(h)) == -6  # TOPAY_ERR_UNSUPPORTED


def test_no_device_means_loud_failure(hip_lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = api.default_params(hip_lib)
    h = C.c_void_p()
    st = hip_lib.topay_create(C.byref(p), 0, C.byref(h))
    assert st == -2  # TOPAY_ERR_NO_DEVICE
    assert b"no CPU fallback" in hip_lib.topay_last_error() or b"hip" in hip_lib.topay_last_error().lower()
    with pytest.raises(api.TopayError):
        api.MomaTrajOptBatch()


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(api.TopayError):
        api.load(str(tmp_path / "libtopay_hip.so"))


def _build_demo(out):
    """examples/cabi_demo.cpp: a plain C++ caller of include/topay.h (no Python, no torch), linked against the library."""
    import subprocess

    lib_dir = os.path.join(ROOT, "topay_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "cabi_demo.cpp"), "-o", out,
                           os.path.join(lib_dir, "libtopay_hip.so"), "-Wl,-rpath," + lib_dir])


def test_cpp_caller_links_against_the_boundary(hip_lib, tmp_path):
    import subprocess

    exe = str(tmp_path / "cabi_demo")
    _build_demo(exe)
    # without a device the demo must stop at topay_create with the library's error text, not compute anything
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "topay_create" in r.stderr


@pytest.mark.gpu
def test_cpp_caller_runs_on_the_device(tmp_path):
    import subprocess

    exe = str(tmp_path / "cabi_demo")
    _build_demo(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("candidate")]
    assert len(lines) == 2 and all("success 1" in ln for ln in lines)
    # the multi-GPU exchange through the C-ABI (ncclAllGather called by the library, world size 1)
    assert "record gather over RCCL (world 1): ok" in r.stdout, r.stdout + r.stderr
    # the front-end chain through the C-ABI: 2-D path around the wall, dense path, joint-space search, trajectory
    fe = [ln for ln in r.stdout.splitlines() if ln.startswith("front-end:")]
    assert len(fe) == 2 and "joint search status 1" in fe[0] and "trajectory success 1" in fe[1], r.stdout
    # the several-processes mode of the exchange (tests/test_sharding.py runs it with two ranks where there are two GPUs) as one rank
    r = subprocess.run([exe, "--exchange", "0", "1", str(tmp_path / "comm.id")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rank 0 of 1 on device 0: record gather over RCCL, 2 records from 1 ranks: ok" in r.stdout, r.stdout + r.stderr


def test_header_is_plain_c99():
    """The boundary is a C ABI: include/topay.h must compile as C (a cgo / JNI / ctypes generator reads it as such)."""
    import subprocess

    src = '#include "topay.h"\nint main(void) { topay_params_t p; return (int)sizeof(p) == 0; }\n'
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-x", "c", "-"],
                   input=src, text=True, check=True)


def test_reference_side_adapter_compiles_against_the_header():
    """examples/moma_traj_opt_hip.h is the class a TopAY maintainer drops into the reference: the
    reference's optimizeTraj / getTraj / traj_cost / printConstraintsSituations surface over the C-ABI, written against
    the real GridMap accessors (grid_map.h:85-86, 179, 203, 206, 213-216).  Eigen and ROS are absent here, so it is
    syntax-checked against name-and-signature stand-ins (tests/stubs/); a signature drift of include/topay.h breaks it."""
    import subprocess

    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "tests", "stubs"), "-I" + os.path.join(ROOT, "examples"),
                        os.path.join(ROOT, "examples", "adapter_syntax_check.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_parameter_file_loader_follows_the_reference_keys():
    """params_from_yaml == MomaTrajOpt::init (moma_traj_opt.h:845-941): keys of the reference's optimizer.yaml land in
    the fields the kernels read; the keys the path has no use for are reported, unknown keys too."""
    text = """
planner_node:
  moma_traj_opt:
    int_K: 12
    sample_interval: 1.25
    mean_time_lowb: 0.4
    energy_weights: [0.5, 1, 1, 1, 1, 1, 1, 1, 2]
    first_stage:
      time_weight: 21.0
      lbgfs_normal_past: 4
      lbfgs: {mem_size: 128, delta: 2.0e-2}
    second_stage:
      collision_weight: 123456.0
      mean_time_weight: 7.0
      lbfgs: {past: 5, min_step: 1.0e-30, max_iterations: 77}
      alm_param:
        init_rho: [2.0e4, 3.0e4, 1, 1, 1, 1, 1, 1, 1]
        tolerance: [0.02]
      alm_data: {max_iter: 100}
    no_such_key: 1
"""
    d = api.default_params()
    p, ignored = api.params_from_yaml(text)
    assert p.sample_interval == 1.25 and p.energy_weights[0] == 0.5 and p.energy_weights[8] == 2.0
    assert p.s1_time_weight == 21.0 and p.s1_normal_past == 4 and p.s1_lbfgs.past == 4
    assert p.s1_lbfgs.mem_size == 128 and p.s1_lbfgs.delta == 2.0e-2 and p.s1_lbfgs.max_iterations == d.s1_lbfgs.max_iterations
    assert p.s2_collision_weight == 123456.0 and p.s2_mean_time_weight == 7.0
    assert p.s2_lbfgs.past == 5 and p.s2_lbfgs.min_step == 1.0e-30 and p.s2_lbfgs.max_iterations == 77
    assert list(p.alm_init_rho) == [2.0e4, 3.0e4] and p.alm_tolerance == 0.02 and list(p.alm_gamma) == list(d.alm_gamma)
    assert p.s2_mani_pos_weight == d.s2_mani_pos_weight and p.max_v == d.max_v       # untouched keys keep the defaults
    assert sorted(ignored) == ["mean_time_lowb", "no_such_key", "second_stage/alm_data/max_iter"]
    # the library's own loader (topay_params_from_yaml, what a C++ planner calls): same fields, same report
    from conftest import EMU_LIB
    pc, ignored_c = api.params_from_yaml_c(text, lib=api.load(EMU_LIB))
    assert bytes(pc) == bytes(p) and sorted(ignored_c) == sorted(ignored)
    ref = "/root/reference/src/planner/params/optimizer.yaml"
    if os.path.exists(ref):      # build container only: the reference's own file gives exactly the built-in defaults
        q, ign = api.params_from_yaml(ref)
        assert bytes(q) == bytes(d)
        qc, ign_c = api.params_from_yaml_c(ref, lib=api.load(EMU_LIB))
        assert bytes(qc) == bytes(d) and sorted(ign_c) == sorted(ign)
        assert sorted(ign) == ["first_stage/mean_time_weight", "mean_time_lowb", "mean_time_uppb", "second_stage/alm_data/epsilon_con",
                               "second_stage/alm_data/max_iter"]


def test_isa_lint_recognises_the_defect_pattern(tmp_path):
    """tools/isa_lint.py (run by build() on the product sources): a join block with a lane-wise copy ahead of its EXEC
    restore is reported, the regular shape is not."""
    import subprocess
    import sys

    bad = tmp_path / "bad.s"
    bad.write_text("k_x:\n\ts_and_saveexec_b64 s[4:5], vcc\n\ts_cbranch_execz .LBB0_2\n\tv_add_f64 v[0:1], v[0:1], v[2:3]\n.LBB0_2:\n"
                   "\tv_readlane_b32 s0, v255, 4\n\tv_accvgpr_write_b32 a86, v152\n\ts_or_b64 exec, exec, s[4:5]\n\ts_endpgm\n")
    good = tmp_path / "good.s"
    good.write_text("k_x:\n\ts_and_saveexec_b64 s[4:5], vcc\n\ts_cbranch_execz .LBB0_2\n\tv_add_f64 v[0:1], v[0:1], v[2:3]\n.LBB0_2:\n"
                    "\tv_readlane_b32 s0, v255, 4\n\ts_or_b64 exec, exec, s[4:5]\n\tv_accvgpr_write_b32 a86, v152\n\ts_endpgm\n")
    tool = os.path.join(ROOT, "tools", "isa_lint.py")
    assert subprocess.run([sys.executable, tool, str(bad)], capture_output=True).returncode == 1
    assert subprocess.run([sys.executable, tool, str(good)], capture_output=True).returncode == 0


def test_scenario_records_match_the_host_selection(hip_lib):
    """topay_scenario_records (planner.cpp:999-1010 inside the library, for C++ callers) against dist.scenario_records, the
    numpy restatement bench.py uses, on a solved batch of the lane emulator."""
    import numpy as np
    from conftest import EMU_LIB, set_map
    from harness import workload as wl
    from topay_amd import dist as tdist

    w, lens, paths, scen = wl.cuboids_batch(3, 2)
    p = api.default_params(api.load(EMU_LIB))
    p.s2_lbfgs.max_iterations = 10
    p.alm_max_outer = 1
    p.alm_tolerance = 10.0
    emu = api.MomaTrajOptBatch(params=p, lib_path=EMU_LIB)
    set_map(emu, w)
    ok = emu.optimizeTraj(lens, paths)
    so = np.array([7, 7, 9, 9, 4, 4], dtype=np.int32)
    recs, win = emu.scenario_records(so)
    acc = (ok & emu.check_feasible()).astype(np.int32)
    ref, rwin = tdist.scenario_records(np.array([7, 9, 4]), so, acc, emu.traj_cost, emu.n_pieces(), emu.total_durations(), return_winners=True)
    assert list(recs["scenario_id"]) == [7, 9, 4]
    for r, q in zip(recs, ref):
        assert r["scenario_id"] == q[0] and r["best_candidate"] == q[1] and r["status"] == q[2]
        if r["status"]:
            assert r["n_pieces"] == q[3] and r["cost"] == q[4] and r["duration"] == q[5]
    assert sorted(int(b) for b in win if b >= 0) == sorted(int(b) for b in rwin)
    with pytest.raises(api.TopayError):            # no communicator yet
        emu.gather_records(recs, 4, 1)
    w.close()


@pytest.mark.gpu
def test_record_gather_through_the_cabi_on_one_gpu():
    """topay_comm_unique_id / topay_comm_init / topay_gather_records: the library's own ncclAllGather (RCCL bound with dlopen),
    world size 1, beside a solve in flight on the same context."""
    import numpy as np
    from conftest import set_map
    from harness import workload as wl

    world, start, goal, lens, paths = wl.tables_scenario(2, 8)
    opt = api.MomaTrajOptBatch(device=0)
    set_map(opt, world)
    opt.optimizeTraj(lens, paths)
    recs, win = opt.scenario_records(np.full(len(lens), 5, dtype=np.int32))
    assert len(recs) == 1 and recs[0]["scenario_id"] == 5
    opt.comm_init(opt.comm_unique_id(), 1, 0)
    other = api.MomaTrajOptBatch(device=0)          # a second context keeps the device busy meanwhile
    set_map(other, world)
    other.set_init_traj(lens, paths)
    other.optimize_async()
    got = opt.gather_records(recs, 16, 1)
    other.finish()
    assert len(got) == 1 and got[0].tobytes() == recs[0].tobytes()
    empty = opt.gather_records(recs[:0], 16, 1)
    assert len(empty) == 0
    opt.comm_destroy()
    world.close()


def test_shared_map_slots_are_invalidated_when_the_owner_refills_or_goes_away(cuboids_small):
    """topay_share_maps hands out device pointers: when the owner refills a shared slot or is destroyed, the sharer must
    lose the slot (TOPAY_ERR_NO_MAP) instead of keeping a dangling descriptor."""
    from conftest import EMU_LIB, set_map
    cs = cuboids_small
    owner = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(owner, cs["world"])
    a = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    a.share_maps(owner, 0, 1)
    a.set_init_traj(cs["lens"][:2], cs["paths"][:cs["offs"][2]])
    x = a.get_x(0)
    f0 = a.eval(1, 0, x)[0]
    set_map(owner, cs["world"])                      # the owner refills slot 0: new buffers may replace the old ones
    with pytest.raises(api.TopayError):
        a.eval(1, 0, x)                              # the resident batch used the slot: it has to be set again
    with pytest.raises(api.TopayError):
        a.set_init_traj(cs["lens"][:2], cs["paths"][:cs["offs"][2]])   # ... and the slot is gone
    a.share_maps(owner, 0, 1)
    a.set_init_traj(cs["lens"][:2], cs["paths"][:cs["offs"][2]])
    assert a.eval(1, 0, x)[0] == f0
    owner.close()                                    # the owner goes away: same rule
    with pytest.raises(api.TopayError):
        a.set_init_traj(cs["lens"][:2], cs["paths"][:cs["offs"][2]])
    a.close()


def test_a_slot_shared_from_a_sharer_follows_the_context_that_holds_the_fields(cuboids_small):
    """A owns the fields, B shares A's slot, C shares the slot from B: C's descriptor points at A's buffers, so it is A's
    refill or destruction that must take the slot away from C -- B going away must not."""
    from conftest import EMU_LIB, set_map
    cs = cuboids_small
    a = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(a, cs["world"])
    b = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    b.share_maps(a, 0, 1)
    c = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    c.share_maps(b, 0, 1)
    c.set_init_traj(cs["lens"][:2], cs["paths"][:cs["offs"][2]])
    x = c.get_x(0)
    f0 = c.eval(1, 0, x)[0]
    b.close()                                        # the middle context goes away: the fields are A's and stay
    assert c.eval(1, 0, x)[0] == f0
    set_map(a, cs["world"])                          # the holder refills the slot: C loses it
    with pytest.raises(api.TopayError):
        c.eval(1, 0, x)
    with pytest.raises(api.TopayError):
        c.set_init_traj(cs["lens"][:2], cs["paths"][:cs["offs"][2]])
    c.share_maps(a, 0, 1)
    c.set_init_traj(cs["lens"][:2], cs["paths"][:cs["offs"][2]])
    assert c.eval(1, 0, x)[0] == f0
    a.close()
    with pytest.raises(api.TopayError):
        c.set_init_traj(cs["lens"][:2], cs["paths"][:cs["offs"][2]])
    c.close()


def test_a_field_with_a_single_z_layer_is_refused(cuboids_small):
    """The ESDF lookups of the evaluation fetch the two z-neighbours of a corner pair with one 16-byte load (round 5): with a
    single layer the pair would reach past the field, so such a map is refused when it is set (the reference's maps have 16)."""
    from conftest import EMU_LIB
    w = cuboids_small["world"]
    o = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    dims = np.array([w.dims[0], w.dims[1], 1], dtype=np.int32)
    e3 = np.ascontiguousarray(w.esdf3d.reshape(w.dims[0], w.dims[1], w.dims[2])[:, :, :1])
    with pytest.raises(api.TopayError):
        o.set_map(w.origin, w.res, dims, w.min_b, w.max_b, w.esdf2d, e3)
    o.close()
