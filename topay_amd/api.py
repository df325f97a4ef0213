"""Host-side mirror of the reference's `MomaTrajOpt` surface over the C-ABI (include/topay.h).

The reference interface (src/planner/include/planner/moma_traj_opt.h:646-674) is one C++ object per
candidate thread: `optimizeTraj(init_path, boundary_vel, boundary_acc) -> bool`, then `getTraj()` and the
public `traj_cost`.  `MomaTrajOptBatch` keeps those names and argument meanings but takes a *batch* of
candidates (one 64-lane wavefront each on the GPU).

This module only binds `topay_amd/lib/libtopay_hip.so` (built by hipcc for gfx950).  There is no CPU
path: if the library is missing, or no HIP device is usable, it raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libtopay_hip.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


class LbfgsParams(C.Structure):
    _fields_ = [("mem_size", C.c_int), ("past", C.c_int), ("max_iterations", C.c_int), ("max_linesearch", C.c_int),
                ("g_epsilon", C.c_double), ("delta", C.c_double), ("min_step", C.c_double), ("max_step", C.c_double),
                ("f_dec_coeff", C.c_double), ("s_curv_coeff", C.c_double), ("cautious_factor", C.c_double),
                ("machine_prec", C.c_double)]


class Params(C.Structure):
    """topay_params_t — mirrors MomaTrajOptParam (moma_traj_opt.h:441-564) + MomaParam constants."""
    _fields_ = [
        ("int_K", C.c_int), ("min_piece_num", C.c_int), ("relu_mu", C.c_double), ("sample_interval", C.c_double),
        ("energy_weights", C.c_double * 9),
        ("s1_time_weight", C.c_double), ("s1_moment_weight", C.c_double), ("s1_acc_weight", C.c_double),
        ("s1_domega_weight", C.c_double), ("s1_path_pos_weight", C.c_double),
        ("s1_normal_past", C.c_int), ("s1_shot_path_past", C.c_int), ("s1_shot_path_horizon", C.c_double),
        ("s1_lbfgs", LbfgsParams),
        ("s2_time_weight", C.c_double), ("s2_moment_weight", C.c_double), ("s2_acc_weight", C.c_double),
        ("s2_domega_weight", C.c_double), ("s2_collision_weight", C.c_double), ("s2_mani_colli_weight", C.c_double),
        ("s2_self_colli_weight", C.c_double), ("s2_mani_pos_weight", C.c_double), ("s2_mani_vel_weight", C.c_double),
        ("s2_mani_acc_weight", C.c_double), ("s2_mean_time_weight", C.c_double),
        ("s2_lbfgs", LbfgsParams),
        ("alm_init_lambda", C.c_double * 2), ("alm_init_rho", C.c_double * 2), ("alm_rho_max", C.c_double * 2),
        ("alm_gamma", C.c_double * 2), ("alm_tolerance", C.c_double), ("alm_max_outer", C.c_int),
        ("alm_work_budget", C.c_int),
        ("chassis_height", C.c_double), ("chassis_colli_radius", C.c_double),
        ("max_v", C.c_double), ("max_a", C.c_double), ("max_w", C.c_double), ("max_dw", C.c_double),
        ("colli_length", C.c_double * 8), ("colli_points", C.c_double * 16), ("colli_point_radius", C.c_double * 16),
        ("joint_pos_limit_max", C.c_double * 7), ("joint_vel_limit", C.c_double * 7), ("joint_acc_limit", C.c_double * 7),
        ("relative_R", C.c_double * 9), ("relative_t", C.c_double * 3),
    ]


class MeshParams(C.Structure):
    """topay_mesh_params_t — the mesh kinematics of MomaParam (moma_param.h:60-67, 77-90, 114-115)."""
    _fields_ = [("link_length", C.c_double * 7), ("joint_pos_limit_min", C.c_double * 7), ("joint_offset", C.c_double * 21),
                ("joint_dof_axis", C.c_double * 21)]


class CommId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class McrrtParams(C.Structure):
    """topay_mcrrt_params_t (include/topay.h): the search's parameters, deterministic caps and seed."""
    _fields_ = [("goal_sample_rate", C.c_double), ("check_colli_res", C.c_double), ("rs_turning_radius", C.c_double),
                ("max_iter", C.c_int), ("max_sample_tries", C.c_int), ("node_cap", C.c_int), ("reserved", C.c_int),
                ("seed", C.c_ulonglong)]


class Record(C.Structure):
    """topay_record_t: the 32-byte per-scenario record of the multi-GPU exchange."""
    _fields_ = [("scenario_id", C.c_int), ("best_candidate", C.c_int), ("status", C.c_int), ("n_pieces", C.c_int),
                ("cost", C.c_double), ("duration", C.c_double)]


class MapDesc(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("resolution", C.c_double), ("dims", C.c_int * 3),
                ("min_boundary", C.c_double * 3), ("max_boundary", C.c_double * 3)]


class TopayError(RuntimeError):
    pass


_libs = {}


def load(path=None):
    """Load the C-ABI library.  Default: the hipcc-built product library; raises if it is missing."""
    path = os.path.abspath(path or os.environ.get("TOPAY_LIB") or DEFAULT_LIB)   # TOPAY_LIB: diagnostic A/B of two builds
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise TopayError(
            f"{path} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()'). "
            "topay_amd has no CPU fallback.")
    L = C.CDLL(path)
    L.topay_last_error.restype = C.c_char_p
    L.topay_default_params.argtypes = [C.POINTER(Params)]
    L.topay_create.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)]
    L.topay_destroy.argtypes = [C.c_void_p]
    L.topay_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
    L.topay_default_mesh_params.argtypes = [C.POINTER(MeshParams)]
    L.topay_mesh_poses.argtypes = [C.c_void_p, C.POINTER(MeshParams), C.c_int, c_dp, c_dp]
    L.topay_mesh_traj.argtypes = [C.c_void_p, C.c_int, C.POINTER(MeshParams), C.c_int, C.c_int, c_dp, c_dp, c_dp, c_ip]
    L.topay_set_map.argtypes = [C.c_void_p, C.c_int, C.POINTER(MapDesc), c_dp, c_dp]
    L.topay_set_init_traj.argtypes = [C.c_void_p, C.c_int, c_ip, c_dp, c_dp, c_dp, c_ip]
    L.topay_reset.argtypes = [C.c_void_p]
    L.topay_optimize.argtypes = [C.c_void_p]
    L.topay_optimize_async.argtypes = [C.c_void_p]
    L.topay_synchronize.argtypes = [C.c_void_p]
    L.topay_get_batch.argtypes = [C.c_void_p, c_ip, c_dp, c_ip]
    L.topay_get_result.argtypes = [C.c_void_p, C.c_int, c_ip, c_dp, c_ip, c_dp, c_dp, c_dp]
    L.topay_get_results.argtypes = [C.c_void_p, C.c_int, c_ip, C.c_int, c_ip, c_dp, c_dp, c_dp]
    L.topay_get_polytraj_msg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.POINTER(C.c_float),
                                         C.POINTER(C.c_float), C.POINTER(C.c_int8), c_ip]
    L.topay_mcrrt_default_params.argtypes = [C.POINTER(McrrtParams)]
    L.topay_mcrrt_default_params.restype = None
    L.topay_mcrrt_plan.argtypes = [C.c_void_p, C.c_int, c_ip, c_ip, c_dp, c_dp, c_dp, C.POINTER(McrrtParams), C.c_ulonglong, C.c_int, c_ip, c_dp,
                                   c_ip, c_dp]
    L.topay_mcrrt_nodes.argtypes = [C.c_void_p, C.c_int, C.c_int, c_ip, c_ip, c_ip, c_dp, c_dp]
    L.topay_reeds_shepp.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, C.c_double, c_dp, c_ip, c_dp, c_dp]
    L.topay_dense_path.argtypes = [C.c_void_p, C.c_int, c_ip, c_dp, C.c_double, c_dp, c_dp, C.c_double, C.c_double, C.c_int, c_ip, c_dp]
    L.topay_connect_check_num.argtypes = [C.c_int, c_dp, c_dp, c_dp, C.c_double, c_ip]
    L.topay_connect_collision.argtypes = [C.c_void_p, C.c_int, C.c_int, c_ip, c_dp, c_dp, c_dp, c_ip]
    L.topay_get_stats.argtypes = [C.c_void_p, c_ip]
    L.topay_get_alm.argtypes = [C.c_void_p, c_dp]
    L.topay_whole_body_collision.argtypes = [C.c_void_p, C.c_int, C.c_int, c_dp, c_ip]
    L.topay_playback.argtypes = [C.c_void_p, C.c_int, C.c_int, c_dp, c_dp, C.c_int, c_dp, c_ip]
    L.topay_build_esdf.argtypes = [C.c_void_p, C.c_int, C.POINTER(MapDesc), C.POINTER(C.c_int8), C.POINTER(C.c_int8)]
    L.topay_get_map.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp]
    L.topay_build_esdf_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(MapDesc), C.POINTER(C.c_int8),
                                         C.POINTER(C.c_int8)]
    L.topay_build_esdf_fields.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(MapDesc), C.POINTER(C.c_int8), C.POINTER(C.c_int8),
                                          C.POINTER(C.c_int8)]
    L.topay_get_map_fields.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp]
    L.topay_get_total_durations.argtypes = [C.c_void_p, c_dp]
    L.topay_check_feasible.argtypes = [C.c_void_p, c_ip]
    L.topay_feasibility_report.argtypes = [C.c_void_p, c_ip, c_ip, c_dp]
    L.topay_get_elapsed_us.argtypes = [C.c_void_p, c_dp, c_dp, c_ip]
    L.topay_get_x.argtypes = [C.c_void_p, C.c_int, c_ip, c_dp]
    L.topay_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]
    L.topay_load_solution.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp]
    L.topay_gate_timeouts.argtypes = [C.c_void_p, c_ip]
    L.topay_class_of.argtypes = [C.c_int, c_ip, c_ip, c_ip]
    L.topay_set_groups.argtypes = [C.c_void_p, c_ip, C.c_int]
    if hasattr(L, "topay_set_latency_mode"):   # (older builds of the library under tools/libs, A/B runs)
        L.topay_set_latency_mode.argtypes = [C.c_void_p, C.c_int]
    L.topay_comm_unique_id.argtypes = [C.POINTER(CommId)]
    L.topay_comm_init.argtypes = [C.c_void_p, C.POINTER(CommId), C.c_int, C.c_int]
    L.topay_comm_destroy.argtypes = [C.c_void_p]
    L.topay_scenario_records.argtypes = [C.c_void_p, c_ip, C.c_int, C.POINTER(Record), c_ip, c_ip]
    L.topay_gather_records.argtypes = [C.c_void_p, C.POINTER(Record), C.c_int, C.c_int, C.POINTER(Record), c_ip]
    L.topay_cancel.argtypes = [C.c_void_p]
    L.topay_get_interrupted.argtypes = [C.c_void_p, c_ip]
    L.topay_workspace_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    L.topay_eval_waves.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]
    L.topay_eval_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, c_dp]
    L.topay_get_nmax.argtypes = [C.c_void_p, c_ip, c_ip]
    L.topay_check_feasible.argtypes = [C.c_void_p, c_ip]
    L.topay_last_kernel_ms.argtypes = [C.c_void_p, c_dp, c_ip]
    if hasattr(L, "topay_last_helper_launches"):
        L.topay_last_helper_launches.argtypes = [C.c_void_p, c_ip]
    L.topay_test_math.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp]
    L.topay_set_trace.argtypes = [C.c_void_p, C.c_int]
    L.topay_get_trace.argtypes = [C.c_void_p, C.c_int, c_dp]
    _libs[path] = L
    return L


def default_params(lib=None):
    L = lib or load()
    p = Params()
    _chk(L, L.topay_default_params(C.byref(p)))
    return p


def params_from_yaml(source, base=None, lib=None):
    """== MomaTrajOpt::init (moma_traj_opt.h:845-941): the optimiser parameters from the reference's parameter file
    (src/planner/params/optimizer.yaml: `planner_node: moma_traj_opt: ...`, the keys init() reads from the parameter
    server).  `source` is a path, a YAML text or an already parsed mapping; keys that are absent keep the value of
    `base` (default: topay_default_params).  Returns (Params, ignored): `ignored` lists the keys the reference reads
    but this path has no use for -- mean_time_lowb / mean_time_uppb (the reference's penalty uses the literals 0.5 and
    2.0, moma_traj_opt.cpp:1752-1769), first_stage/mean_time_weight (stage 1 has no mean-time term), alm_data/* (read
    into a struct optimizeTraj never consults) -- and any key init() does not know."""
    import yaml

    if isinstance(source, dict):
        doc = source
    else:
        text = open(source).read() if os.path.exists(str(source)) else str(source)
        doc = yaml.safe_load(text) or {}
    for k in ("planner_node", "moma_traj_opt"):
        if isinstance(doc, dict) and k in doc:
            doc = doc[k]
    p = Params()
    if base is not None:
        C.memmove(C.byref(p), C.byref(base), C.sizeof(Params))
    else:
        _chk(lib or load(), (lib or load()).topay_default_params(C.byref(p)))
    ignored = []

    def vec(dst, src, n):
        for i in range(min(n, len(src))):
            dst[i] = float(src[i])

    def lbfgs(dst, m, prefix):
        for k, v in m.items():
            if k in ("mem_size", "past", "max_iterations"):
                setattr(dst, k, int(v))
            elif k in ("g_epsilon", "min_step", "delta"):
                setattr(dst, k, float(v))
            else:
                ignored.append(prefix + k)

    for k, v in (doc or {}).items():
        if k in ("int_K", "min_piece_num"):
            setattr(p, k, int(v))
        elif k in ("relu_mu", "sample_interval"):
            setattr(p, k, float(v))
        elif k == "energy_weights":
            vec(p.energy_weights, v, 9)
        elif k == "first_stage":
            for kk, vv in v.items():
                if kk in ("time_weight", "moment_weight", "acc_weight", "domega_weight", "path_pos_weight"):
                    setattr(p, "s1_" + kk, float(vv))
                elif kk == "lbgfs_normal_past":          # (sic) also becomes the stage-1 `past`, moma_traj_opt.h:873
                    p.s1_normal_past = int(vv)
                    p.s1_lbfgs.past = int(vv)
                elif kk == "lbgfs_shot_path_past":
                    p.s1_shot_path_past = int(vv)
                elif kk == "shot_path_horizon":
                    p.s1_shot_path_horizon = float(vv)
                elif kk == "lbfgs":
                    lbfgs(p.s1_lbfgs, {a: b for a, b in vv.items() if a != "past"}, "first_stage/lbfgs/")
                    if "past" in vv:
                        ignored.append("first_stage/lbfgs/past")
                else:
                    ignored.append("first_stage/" + kk)
        elif k == "second_stage":
            for kk, vv in v.items():
                if kk in ("time_weight", "moment_weight", "acc_weight", "domega_weight", "collision_weight", "mani_colli_weight",
                          "self_colli_weight", "mani_pos_weight", "mani_vel_weight", "mani_acc_weight", "mean_time_weight"):
                    setattr(p, "s2_" + kk, float(vv))
                elif kk == "lbfgs":
                    lbfgs(p.s2_lbfgs, vv, "second_stage/lbfgs/")
                elif kk == "alm_param":
                    for a, b in vv.items():       # the reference sizes these vectors 9 and uses entries 0 and 1 (end-point x, y)
                        if a in ("init_lambda", "init_rho", "rho_max", "gamma"):
                            vec(getattr(p, "alm_" + a), b, 2)
                        elif a == "tolerance":
                            p.alm_tolerance = float(b[0])
                        else:
                            ignored.append("second_stage/alm_param/" + a)
                else:
                    ignored.extend(f"second_stage/{kk}/{a}" for a in vv) if isinstance(vv, dict) else ignored.append("second_stage/" + kk)
        else:
            ignored.append(k)
    return p, ignored


def params_from_yaml_c(source, base=None, lib=None):
    """The library's own loader (topay_params_from_yaml, for C / C++ callers): same mapping as params_from_yaml."""
    L = lib or load()
    p = Params()
    if base is not None:
        C.memmove(C.byref(p), C.byref(base), C.sizeof(Params))
    else:
        _chk(L, L.topay_default_params(C.byref(p)))
    buf = C.create_string_buffer(4096)
    L.topay_params_from_yaml.argtypes = [C.c_char_p, C.POINTER(Params), C.c_char_p, C.c_int]
    _chk(L, L.topay_params_from_yaml(str(source).encode(), C.byref(p), buf, 4096))
    ign = buf.value.decode()
    return p, ([x for x in ign.split("\n") if x] if ign else [])


def pack_records(records, per_rank, lib=None):
    """A rank's block of the record exchange (topay_pack_records): its records padded to per_rank entries whose
    scenario_id is INT32_MIN.  No device involved."""
    L = load(lib)
    rec = np.ascontiguousarray(records)
    out = (Record * per_rank)()
    L.topay_pack_records.argtypes = [C.POINTER(Record), C.c_int, C.c_int, C.POINTER(Record)]
    _chk(L, L.topay_pack_records(rec.ctypes.data_as(C.POINTER(Record)), len(rec), per_rank, out))
    return np.ctypeslib.as_array(out).copy()


def unpack_records(gathered, world, per_rank, lib=None):
    """The valid records of world x per_rank gathered entries, in rank order (topay_unpack_records)."""
    L = load(lib)
    g = np.ascontiguousarray(gathered)
    out = (Record * (per_rank * world))()
    nv = C.c_int(0)
    L.topay_unpack_records.argtypes = [C.POINTER(Record), C.c_int, C.c_int, C.POINTER(Record), C.POINTER(C.c_int)]
    _chk(L, L.topay_unpack_records(g.ctypes.data_as(C.POINTER(Record)), world, per_rank, out, C.byref(nv)))
    return np.ctypeslib.as_array(out)[:nv.value].copy()


def _chk(L, status):
    if status != 0:
        raise TopayError(f"topay status {status}: {L.topay_last_error().decode()}")


def _dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_ip) if a is not None else None


STAT_KEYS = ["stage1_ret", "stage1_iters", "stage1_evals", "stage2_last_ret", "stage2_iters", "stage2_evals",
             "alm_outer", "sum_bound"]


class MomaTrajOptBatch:
    """Batched stand-in for `MomaTrajOpt` (reference: moma_traj_opt.h:613-674)."""

    def __init__(self, params=None, device=0, lib_path=None):
        self.L = load(lib_path)
        self.opt_param = params if params is not None else default_params(self.L)
        h = C.c_void_p()
        _chk(self.L, self.L.topay_create(C.byref(self.opt_param), device, C.byref(h)))
        self.h = h
        self.batch = 0
        self.traj_cost = None
        self._map_dims = {}

    def close(self):
        if getattr(self, "h", None):
            self.L.topay_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- MomaTrajOpt(GridMap::Ptr): the shared read-only map
    def set_map(self, origin, resolution, dims, min_boundary, max_boundary, esdf2d, esdf3d, map_id=0):
        d = MapDesc()
        for i in range(3):
            d.origin[i] = origin[i]
            d.dims[i] = int(dims[i])
            d.min_boundary[i] = min_boundary[i]
            d.max_boundary[i] = max_boundary[i]
        d.resolution = resolution
        e2 = np.ascontiguousarray(esdf2d, dtype=np.float64)
        e3 = np.ascontiguousarray(esdf3d, dtype=np.float64)
        _chk(self.L, self.L.topay_set_map(self.h, map_id, C.byref(d), _dp(e2), _dp(e3)))
        self._map_dims[map_id] = tuple(int(x) for x in dims)

    # -- optimizeTraj lines 146-357
    def set_init_traj(self, path_len, init_paths, boundary_vel=None, boundary_acc=None, map_ids=None):
        pl = np.ascontiguousarray(path_len, dtype=np.int32)
        ip = np.ascontiguousarray(init_paths, dtype=np.float64)
        bv = None if boundary_vel is None else np.ascontiguousarray(boundary_vel, dtype=np.float64)
        ba = None if boundary_acc is None else np.ascontiguousarray(boundary_acc, dtype=np.float64)
        mi = None if map_ids is None else np.ascontiguousarray(map_ids, dtype=np.int32)
        self.batch = len(pl)
        _chk(self.L, self.L.topay_set_init_traj(self.h, self.batch, _ip(pl), _dp(ip), _dp(bv), _dp(ba), _ip(mi)))

    def reset(self):
        _chk(self.L, self.L.topay_reset(self.h))

    # -- optimizeTraj lines 359-497; returns the per-candidate bool of the reference
    def optimize(self):
        _chk(self.L, self.L.topay_optimize(self.h))
        succ = np.zeros(self.batch, dtype=np.int32)
        cost = np.zeros(self.batch)
        _chk(self.L, self.L.topay_get_batch(self.h, _ip(succ), _dp(cost), None))
        self.traj_cost = cost
        return succ.astype(bool)

    def optimize_within(self, budget_ms):
        """optimize with a wall-clock budget (topay_optimize_within): (success flags, timed_out)."""
        to = C.c_int(0)
        self.L.topay_optimize_within.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_int)]
        _chk(self.L, self.L.topay_optimize_within(self.h, float(budget_ms), C.byref(to)))
        return self.finish(), bool(to.value)

    def optimize_async(self):
        """Issue the solve without waiting (several contexts may be in flight on one GPU); pair with finish()."""
        _chk(self.L, self.L.topay_optimize_async(self.h))

    def finish(self):
        """Wait for optimize_async and fetch the per-candidate results, like optimize()."""
        _chk(self.L, self.L.topay_synchronize(self.h))
        succ = np.zeros(self.batch, dtype=np.int32)
        cost = np.zeros(self.batch)
        _chk(self.L, self.L.topay_get_batch(self.h, _ip(succ), _dp(cost), None))
        self.traj_cost = cost
        return succ.astype(bool)

    def optimizeTraj(self, path_len, init_paths, boundary_vel=None, boundary_acc=None, map_ids=None):
        self.set_init_traj(path_len, init_paths, boundary_vel, boundary_acc, map_ids)
        return self.optimize()

    def n_pieces(self):
        n = np.zeros(self.batch, dtype=np.int32)
        _chk(self.L, self.L.topay_get_batch(self.h, None, None, _ip(n)))
        return n

    # -- getTraj()
    def getTraj(self, i):
        n = C.c_int(0)
        _chk(self.L, self.L.topay_get_result(self.h, i, None, None, C.byref(n), None, None, None))
        N = n.value
        dur = np.zeros(N)
        coef = np.zeros(N * 54)
        knots = np.zeros((N + 1) * 2)
        succ = C.c_int(0)
        cost = C.c_double(0)
        _chk(self.L, self.L.topay_get_result(self.h, i, C.byref(succ), C.byref(cost), None, _dp(dur), _dp(coef), _dp(knots)))
        return dict(success=bool(succ.value), cost=cost.value, durations=dur, coeffs=coef.reshape(N, 9, 6),
                    knots_xy=knots.reshape(N + 1, 2))

    def getTrajs(self, idx, n_pieces=None):
        """getTraj() of a selection of candidates in one call / one copy (the planner's winners): dict of packed arrays
        piece_off[n+1], durations[P], coeffs[P, 9, 6], knots_xy[P + n, 2] (candidate k's knots start at piece_off[k] + k)."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        n = len(idx)
        npc = self.n_pieces() if n_pieces is None else n_pieces
        cap = int(npc[idx].sum()) if n else 0
        off = np.zeros(n + 1, dtype=np.int32)
        dur = np.zeros(cap)
        coef = np.zeros(cap * 54)
        kn = np.zeros(2 * (cap + n))
        _chk(self.L, self.L.topay_get_results(self.h, n, _ip(idx), cap, _ip(off), _dp(dur), _dp(coef), _dp(kn)))
        return dict(piece_off=off, durations=dur, coeffs=coef.reshape(cap, 9, 6), knots_xy=kn.reshape(cap + n, 2))

    def polytraj_msg(self, i):
        """Candidate i in the field layout of planner/msg/PolyTraj.msg: (order, coeff[N, 9, 6] f32, durations[N] f32, directions[N] i8)."""
        N = int(self.n_pieces()[i])
        order = C.c_ubyte(0)
        coef = np.zeros(max(N, 1) * 54, dtype=np.float32)
        dur = np.zeros(max(N, 1), dtype=np.float32)
        dirs = np.zeros(max(N, 1), dtype=np.int8)
        n = C.c_int(0)
        _chk(self.L, self.L.topay_get_polytraj_msg(self.h, i, max(N, 1), C.byref(order), coef.ctypes.data_as(C.POINTER(C.c_float)),
                                                   dur.ctypes.data_as(C.POINTER(C.c_float)), dirs.ctypes.data_as(C.POINTER(C.c_int8)),
                                                   C.byref(n)))
        return order.value, coef[:N * 54].reshape(N, 9, 6), dur[:N], dirs[:N]

    def stats(self):
        s = np.zeros(self.batch * 8, dtype=np.int32)
        _chk(self.L, self.L.topay_get_stats(self.h, _ip(s)))
        return s.reshape(self.batch, 8)

    def check_feasible(self, report=False):
        """printConstraintsSituations for every candidate (bool array); with report=True also checkFeasible's verdict
        and the 38 extreme values per candidate."""
        f = np.zeros(self.batch, dtype=np.int32)
        if not report:
            _chk(self.L, self.L.topay_check_feasible(self.h, _ip(f)))
            return f.astype(bool)
        st = np.zeros(self.batch, dtype=np.int32)
        rep = np.zeros(self.batch * 38)
        _chk(self.L, self.L.topay_feasibility_report(self.h, _ip(f), _ip(st), _dp(rep)))
        return f.astype(bool), st.astype(bool), rep.reshape(self.batch, 38)

    def total_durations(self):
        """Total duration of every candidate's trajectory (what the planner ranks successful candidates by)."""
        t = np.zeros(self.batch)
        _chk(self.L, self.L.topay_get_total_durations(self.h, _dp(t)))
        return t

    def build_esdf(self, origin, res, dims, min_b, max_b, occ2d, occ3d, map_id=0):
        """GridMap::updateESDF on the device from the occupancy grids; the slot is then usable like after set_map."""
        d = MapDesc()
        for k in range(3):
            d.origin[k] = origin[k]; d.dims[k] = int(dims[k]); d.min_boundary[k] = min_b[k]; d.max_boundary[k] = max_b[k]
        d.resolution = res
        o2 = np.ascontiguousarray(occ2d, dtype=np.int8)
        o3 = np.ascontiguousarray(occ3d, dtype=np.int8)
        _chk(self.L, self.L.topay_build_esdf(self.h, map_id, C.byref(d), o2.ctypes.data_as(C.POINTER(C.c_int8)),
                                             o3.ctypes.data_as(C.POINTER(C.c_int8))))
        self._map_dims[map_id] = tuple(int(x) for x in dims)

    def build_esdf_batch(self, origin, res, dims, min_b, max_b, occ2d_all, occ3d_all, first_map_id=0):
        """Same for a batch of equally sized maps: occ*_all are [n_maps, cells] arrays."""
        d = MapDesc()
        for k in range(3):
            d.origin[k] = origin[k]; d.dims[k] = int(dims[k]); d.min_boundary[k] = min_b[k]; d.max_boundary[k] = max_b[k]
        d.resolution = res
        o2 = np.ascontiguousarray(occ2d_all, dtype=np.int8)
        o3 = np.ascontiguousarray(occ3d_all, dtype=np.int8)
        n_maps = o2.shape[0]
        _chk(self.L, self.L.topay_build_esdf_batch(self.h, n_maps, first_map_id, C.byref(d), o2.ctypes.data_as(C.POINTER(C.c_int8)),
                                                   o3.ctypes.data_as(C.POINTER(C.c_int8))))
        for k in range(n_maps):
            self._map_dims[first_map_id + k] = tuple(int(x) for x in dims)

    def build_esdf_fields(self, origin, res, dims, min_b, max_b, occ2d, occ2d_critical, occ3d, map_id=0):
        """GridMap::updateESDF in full (also the inflate / critical 2-D fields); occ2d_critical may be None."""
        d = MapDesc()
        for k in range(3):
            d.origin[k] = origin[k]; d.dims[k] = int(dims[k]); d.min_boundary[k] = min_b[k]; d.max_boundary[k] = max_b[k]
        d.resolution = res
        o2 = np.ascontiguousarray(occ2d, dtype=np.int8)
        o3 = np.ascontiguousarray(occ3d, dtype=np.int8)
        oc = None if occ2d_critical is None else np.ascontiguousarray(occ2d_critical, dtype=np.int8)
        P8 = C.POINTER(C.c_int8)
        _chk(self.L, self.L.topay_build_esdf_fields(self.h, 1, map_id, C.byref(d), o2.ctypes.data_as(P8),
                                                    None if oc is None else oc.ctypes.data_as(P8), o3.ctypes.data_as(P8)))
        self._map_dims[map_id] = tuple(int(x) for x in dims)

    def get_map_fields(self, map_id=0):
        """(esdf2d_inflate, esdf2d_critical) of a map built on the device."""
        nx, ny, nz = self._map_dims[map_id]
        a = np.zeros(nx * ny)
        b = np.zeros(nx * ny)
        _chk(self.L, self.L.topay_get_map_fields(self.h, map_id, _dp(a), _dp(b)))
        return a, b

    def get_map(self, map_id=0):
        """(esdf2d, esdf3d, build_ms) of a resident map."""
        nx, ny, nz = self._map_dims[map_id]
        e2 = np.zeros(nx * ny)
        e3 = np.zeros(nx * ny * nz)
        ms = C.c_double(0)
        _chk(self.L, self.L.topay_get_map(self.h, map_id, _dp(e2), _dp(e3), C.byref(ms)))
        return e2, e3, ms.value

    def playback(self, i, times):
        """MomaTraj playback of candidate i: (states[len(times), 10], car_seq[n, 4])."""
        t = np.ascontiguousarray(times, dtype=np.float64)
        st = np.zeros((len(t), 10))
        cap = 4096
        seq = np.zeros((cap, 4))
        n = C.c_int(0)
        _chk(self.L, self.L.topay_playback(self.h, i, len(t), _dp(t), _dp(st), cap, _dp(seq), C.byref(n)))
        return st, seq[:min(n.value, cap)].copy()

    def whole_body_collision(self, states, map_id=0):
        """GridMap::isWholeBodyCollision for an [n, 10] array of states -> bool array."""
        st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 10)
        out = np.zeros(len(st), dtype=np.int32)
        _chk(self.L, self.L.topay_whole_body_collision(self.h, map_id, len(st), _dp(st), _ip(out)))
        return out.astype(bool)

    def dense_path(self, raw_paths, start_yaw, end_yaw, step_size=1.414, v_max=1.0, w_max=1.25, cap=None):
        """GraphSearch::getDensePath for a list of raw 2-D paths ([k_p, 2] arrays) -> list of [m_p, 4] arrays (x, y, theta, dt)."""
        lens = np.array([len(p) for p in raw_paths], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1, 2) for p in raw_paths]))
        sy = np.ascontiguousarray(start_yaw, dtype=np.float64)
        ey = np.ascontiguousarray(end_yaw, dtype=np.float64)
        if cap is None:
            seg = [np.linalg.norm(np.diff(np.asarray(p, dtype=np.float64).reshape(-1, 2), axis=0), axis=1) for p in raw_paths]
            cap = int(max(2 * (np.ceil(s_ / step_size).clip(1).sum() + 1) + 4 for s_ in seg))
        out = np.zeros((len(lens), cap, 4))
        n = np.zeros(len(lens), dtype=np.int32)
        _chk(self.L, self.L.topay_dense_path(self.h, len(lens), _ip(lens), _dp(flat), step_size, _dp(sy), _dp(ey), v_max, w_max, cap, _ip(n), _dp(out)))
        return [out[p, :min(n[p], cap)].copy() for p in range(len(lens))], n

    def connect_collision(self, rs_distance, car_pose_fn, q_from, q_to, check_res, map_id=0):
        """MCRRTs::connectCollision for a batch of edges.  car_pose_fn(e, fractions) -> [len(fractions), 3] car poses of
        edge e (the caller's Reeds-Shepp interpolation).  Returns (collide[n] bool, piece_num[n])."""
        rs = np.ascontiguousarray(rs_distance, dtype=np.float64)
        qf = np.ascontiguousarray(q_from, dtype=np.float64).reshape(-1, 7)
        qt = np.ascontiguousarray(q_to, dtype=np.float64).reshape(-1, 7)
        n = len(rs)
        pn = np.zeros(n, dtype=np.int32)
        _chk(self.L, self.L.topay_connect_check_num(n, _dp(rs), _dp(qf), _dp(qt), check_res, _ip(pn)))
        car = np.ascontiguousarray(np.concatenate([car_pose_fn(e, np.arange(pn[e]) / float(pn[e])) for e in range(n)]), dtype=np.float64)
        col = np.zeros(n, dtype=np.int32)
        _chk(self.L, self.L.topay_connect_collision(self.h, map_id, n, _ip(pn), _dp(car), _dp(qf), _dp(qt), _ip(col)))
        return col.astype(bool), pn

    def share_maps(self, owner, first_map_id=0, n_maps=1):
        """Use `owner`'s resident maps (same device) for these slots instead of a copy of the fields."""
        self.L.topay_share_maps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _chk(self.L, self.L.topay_share_maps(self.h, owner.h, int(first_map_id), int(n_maps)))
        self._map_owner = owner      # keeps the owner alive

    def plan2d_jps(self, start_xy, end_xy, threshold=0.5, map_ids=None, cap=512):
        """GraphSearch::plan2dJPS for a batch of (start, goal) pairs -> (list of [m, 2] paths (empty: none), stats [n, 2])."""
        a = np.ascontiguousarray(start_xy, dtype=np.float64).reshape(-1, 2)
        b = np.ascontiguousarray(end_xy, dtype=np.float64).reshape(-1, 2)
        n = len(a)
        mid = None if map_ids is None else np.ascontiguousarray(map_ids, dtype=np.int32)
        ln, st, out = np.zeros(n, dtype=np.int32), np.zeros((n, 2), dtype=np.int32), np.zeros((n, cap, 2))
        self.L.topay_plan2d_jps.argtypes = [C.c_void_p, C.c_int, c_ip, c_dp, c_dp, C.c_double, C.c_int, c_ip, c_dp, c_ip]
        _chk(self.L, self.L.topay_plan2d_jps(self.h, n, None if mid is None else _ip(mid), _dp(a), _dp(b), float(threshold), int(cap), _ip(ln), _dp(out),
                                             _ip(st)))
        return [out[p, :min(ln[p], cap)].copy() for p in range(n)], st, ln

    def mcrrt_params(self, **kw):
        p = McrrtParams()
        self.L.topay_mcrrt_default_params(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def mcrrt_plan(self, lens, car_paths, start, end, params=None, map_ids=None, first_instance=0, cap=None):
        """MCRRTs::plan for a batch of chassis paths (ragged car_paths [sum lens, 4] = (x, y, theta, dt); start / end [n, 10]).
        Returns (wb_paths: list of [m, 10] arrays (empty when no path), stats [n, 8], c_max [n])."""
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        n = len(lens)
        car = np.ascontiguousarray(car_paths, dtype=np.float64).reshape(-1, 4)
        st = np.ascontiguousarray(start, dtype=np.float64).reshape(n, 10)
        en = np.ascontiguousarray(end, dtype=np.float64).reshape(n, 10)
        cap = int(cap or max(2, int(lens.max()) if n else 2))
        mid = None if map_ids is None else np.ascontiguousarray(map_ids, dtype=np.int32)
        wl_ = np.zeros(n, dtype=np.int32)
        wb = np.zeros((n, cap, 10))
        stats = np.zeros((n, 8), dtype=np.int32)
        cmax = np.zeros(n)
        _chk(self.L, self.L.topay_mcrrt_plan(self.h, n, None if mid is None else _ip(mid), _ip(lens), _dp(car), _dp(st), _dp(en),
                                             None if params is None else C.byref(params), int(first_instance), cap, _ip(wl_), _dp(wb),
                                             _ip(stats), _dp(cmax)))
        return [wb[p, :wl_[p]].copy() for p in range(n)], stats, cmax

    def mcrrt_nodes(self, instance, n_nodes):
        """Node table of one instance of the last mcrrt_plan: dict of layer, state, parent, cost, q ([n_nodes, 7])."""
        m = int(n_nodes)
        layer, state, parent = (np.zeros(m, dtype=np.int32) for _ in range(3))
        cost, q = np.zeros(m), np.zeros((m, 7))
        _chk(self.L, self.L.topay_mcrrt_nodes(self.h, int(instance), m, _ip(layer), _ip(state), _ip(parent), _dp(cost), _dp(q)))
        return dict(layer=layer, state=state, parent=parent, cost=cost, q=q)

    def reeds_shepp(self, from_poses, to_poses, t=None, rho=1.0e-2):
        """ompl ReedsSheppStateSpace(rho): (distance [n], word [n], lengths [n, 5], pose [n, 3] or None)."""
        a = np.ascontiguousarray(from_poses, dtype=np.float64).reshape(-1, 3)
        b = np.ascontiguousarray(to_poses, dtype=np.float64).reshape(-1, 3)
        n = len(a)
        tt = None if t is None else np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (n,)))
        d, w, ln, po = np.zeros(n), np.zeros(n, dtype=np.int32), np.zeros((n, 5)), np.zeros((n, 3))
        _chk(self.L, self.L.topay_reeds_shepp(self.h, n, _dp(a), _dp(b), None if tt is None else _dp(tt), float(rho), _dp(d), _ip(w), _dp(ln),
                                              _dp(po)))
        return d, w, ln, (po if tt is not None else None)

    def alm_state(self):
        """(lambda0, lambda1, rho0, rho1) every candidate finished with."""
        a = np.zeros(self.batch * 4)
        _chk(self.L, self.L.topay_get_alm(self.h, _dp(a)))
        return a.reshape(self.batch, 4)

    def elapsed_us(self):
        """Device-measured optimisation time of every candidate (microseconds)."""
        t = np.zeros(self.batch)
        _chk(self.L, self.L.topay_get_elapsed_us(self.h, _dp(t), None, None))
        return t

    def start_us(self, raw=False):
        """When every candidate's solve began, on the device's constant clock (raw=True: comparable between contexts of
        one device; default: relative to the first start of this batch)."""
        t = np.zeros(self.batch)
        _chk(self.L, self.L.topay_get_elapsed_us(self.h, None, _dp(t), None))
        return t if raw else t - t.min()

    def hw_ids(self):
        h = np.zeros(self.batch, dtype=np.int32)
        _chk(self.L, self.L.topay_get_elapsed_us(self.h, None, None, _ip(h)))
        return h

    def get_x(self, i):
        n = C.c_int(0)
        _chk(self.L, self.L.topay_get_x(self.h, i, C.byref(n), None))
        x = np.zeros(n.value)
        _chk(self.L, self.L.topay_get_x(self.h, i, C.byref(n), _dp(x)))
        return x

    # -- firstStageCostCallback / secondStageCostCallback (test hook)
    def eval(self, stage, i, x, alm_lambda=None, alm_rho=None, waves=None):
        """(f, g, final_xy_error) of candidate i at x.  waves=1/2/4: by the kernel with that many wavefronts per trajectory
        (test hook topay_eval_waves) instead of the candidate's class default."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.zeros_like(x)
        f = C.c_double(0)
        e = np.zeros(2)
        lam = None if alm_lambda is None else np.ascontiguousarray(alm_lambda, dtype=np.float64)
        rho = None if alm_rho is None else np.ascontiguousarray(alm_rho, dtype=np.float64)
        if waves is None:
            _chk(self.L, self.L.topay_eval(self.h, stage, i, _dp(x), _dp(lam), _dp(rho), C.byref(f), _dp(g), _dp(e)))
        else:
            _chk(self.L, self.L.topay_eval_waves(self.h, stage, i, int(waves), _dp(x), _dp(lam), _dp(rho), C.byref(f), _dp(g), _dp(e)))
        return f.value, g, e

    def load_solution(self, i, x, alm_lambda=None, alm_rho=None):
        """Make the spline of the decision vector x candidate i's result (getTraj() after an evaluation at x)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        lam = None if alm_lambda is None else np.ascontiguousarray(alm_lambda, dtype=np.float64)
        rho = None if alm_rho is None else np.ascontiguousarray(alm_rho, dtype=np.float64)
        _chk(self.L, self.L.topay_load_solution(self.h, i, _dp(x), _dp(lam), _dp(rho)))

    def class_of(self, n_pieces):
        """(waves the solver's vectors are divided over, vector elements per thread of the solver, launch class index) of a candidate of
        n_pieces pieces: what the bits of its solve depend on (since round 5 one wave for every class; the classes from index 4 up
        -- more than 32 pieces -- evaluate on four waves)."""
        w, e, k = C.c_int(0), C.c_int(0), C.c_int(0)
        _chk(self.L, self.L.topay_class_of(int(n_pieces), C.byref(w), C.byref(e), C.byref(k)))
        return w.value, e.value, k.value

    def workspace_bytes(self):
        b = C.c_ulonglong(0)
        _chk(self.L, self.L.topay_workspace_bytes(self.h, C.byref(b)))
        return int(b.value)

    def set_groups(self, group_id, cancel_budget=2400):
        """Planning call (scenario) of every candidate and the cancellation window after a call's first feasible success, in
        piece-evaluations (2400 = the reference's 100 ms); None / 0 switches cancellation off."""
        g = None if group_id is None else np.ascontiguousarray(group_id, dtype=np.int32)
        _chk(self.L, self.L.topay_set_groups(self.h, _ip(g), int(cancel_budget) if g is not None else 0))

    def set_latency_mode(self, mode):
        """0: never (default); 1: batches with at most one candidate per SIMD (1024 on an MI355X) run on four-wave workgroups whose extra
        waves join the evaluations only (same bits, shorter solves); 2: every batch."""
        _chk(self.L, self.L.topay_set_latency_mode(self.h, int(mode)))

    def cancel(self):
        _chk(self.L, self.L.topay_cancel(self.h))

    def interrupted(self):
        out = np.zeros(self.batch, dtype=np.int32)
        _chk(self.L, self.L.topay_get_interrupted(self.h, _ip(out)))
        return out.astype(bool)

    # -- the multi-GPU exchange through the C-ABI (a C++ planner's path; bench.py uses torch.distributed for the same records)
    def comm_unique_id(self):
        cid = CommId()
        _chk(self.L, self.L.topay_comm_unique_id(C.byref(cid)))
        return cid

    def comm_init(self, cid, world, rank):
        _chk(self.L, self.L.topay_comm_init(self.h, C.byref(cid), world, rank))

    def comm_destroy(self):
        _chk(self.L, self.L.topay_comm_destroy(self.h))

    def scenario_records(self, scenario_of):
        """(records structured array, winner batch indices) of the solved batch: one 32-byte record per scenario."""
        so = np.ascontiguousarray(scenario_of, dtype=np.int32)
        cap = len(set(so.tolist()))
        recs = (Record * max(cap, 1))()
        n = C.c_int(0)
        win = np.zeros(max(cap, 1), dtype=np.int32)
        _chk(self.L, self.L.topay_scenario_records(self.h, _ip(so), cap, recs, C.byref(n), _ip(win)))
        return np.ctypeslib.as_array(recs)[:n.value].copy() if n.value else np.zeros(0, dtype=np.dtype(Record)), win[:n.value]

    def gather_records(self, records, per_rank, world):
        rec = np.ascontiguousarray(records)
        out = (Record * (per_rank * world))()
        nv = C.c_int(0)
        _chk(self.L, self.L.topay_gather_records(self.h, rec.ctypes.data_as(C.POINTER(Record)), len(rec), per_rank, out, C.byref(nv)))
        return np.ctypeslib.as_array(out)[:nv.value].copy()

    def gate_timeouts(self):
        n = C.c_int(0)
        _chk(self.L, self.L.topay_gate_timeouts(self.h, C.byref(n)))
        return n.value

    def _mesh_params(self):
        m = MeshParams()
        _chk(self.L, self.L.topay_default_mesh_params(C.byref(m)))
        return m

    def mesh_poses(self, states, mesh=None):
        """MomaParam::getMeshPose for n states: (n, 11, 7) = (x, y, z, qw, qx, qy, qz) of chassis, stump, 7 links, ee, ee point."""
        st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 10)
        out = np.zeros((len(st), 11, 7))
        m = mesh or self._mesh_params()
        _chk(self.L, self.L.topay_mesh_poses(self.h, C.byref(m), len(st), _dp(st), _dp(out)))
        return out

    def mesh_traj(self, i, res=1000, mesh=None):
        """Planner::toMeshMsg of candidate i: (parts (n, 11, 7), yaws (n), arc_lengths (n))."""
        cap = res + 1
        parts, yaws, arcs = np.zeros((cap, 11, 7)), np.zeros(cap), np.zeros(cap)
        n = np.zeros(1, dtype=np.int32)
        m = mesh or self._mesh_params()
        _chk(self.L, self.L.topay_mesh_traj(self.h, i, C.byref(m), res, cap, _dp(parts), _dp(yaws), _dp(arcs), _ip(n)))
        return parts[:n[0]], yaws[:n[0]], arcs[:n[0]]

    def set_params(self, params=None):
        """Push `opt_param` (or `params`) to the context: the reference's `opt_param` is a public member the planner may
        change between calls (moma_traj_opt.h:616)."""
        if params is not None:
            self.opt_param = params
        _chk(self.L, self.L.topay_set_params(self.h, C.byref(self.opt_param)))

    COST_TERMS = ("jerk", "time", "chassis_colli", "moment", "acc", "domega", "mani_colli", "self_colli", "mani_pos",
                  "mani_vel", "mani_acc", "mean_time", "endp")

    def cost_terms(self, i, x, alm_lambda, alm_rho):
        """Per-term breakdown of the stage-2 cost at x -- what the reference's DebugManager publishes under
        /debug_cost/<name> (moma_traj_opt.h:566-611, names as in 930-941).  Every term carries its own weight, so term k
        is the cost evaluated with every other weight at zero (the ALM term switched off by lambda = 0, rho = 1e-300):
        13 evaluations through topay_eval with topay_set_params in between; a debugging aid, not a hot path."""
        own = {"jerk": "energy_weights", "time": "s2_time_weight", "chassis_colli": "s2_collision_weight",
               "moment": "s2_moment_weight", "acc": "s2_acc_weight", "domega": "s2_domega_weight",
               "mani_colli": "s2_mani_colli_weight", "self_colli": "s2_self_colli_weight", "mani_pos": "s2_mani_pos_weight",
               "mani_vel": "s2_mani_vel_weight", "mani_acc": "s2_mani_acc_weight", "mean_time": "s2_mean_time_weight"}
        keep = Params()
        C.memmove(C.byref(keep), C.byref(self.opt_param), C.sizeof(Params))
        out = {}
        try:
            for name in self.COST_TERMS:
                q = Params()
                C.memmove(C.byref(q), C.byref(keep), C.sizeof(Params))
                for term, field in own.items():
                    if term == name:
                        continue
                    if field == "energy_weights":
                        for k in range(9):
                            q.energy_weights[k] = 0.0
                    else:
                        setattr(q, field, 0.0)
                _chk(self.L, self.L.topay_set_params(self.h, C.byref(q)))
                if name == "endp":
                    out[name] = self.eval(2, i, x, alm_lambda, alm_rho)[0]
                else:
                    out[name] = self.eval(2, i, x, [0.0, 0.0], [1e-300, 1e-300])[0]
        finally:
            _chk(self.L, self.L.topay_set_params(self.h, C.byref(keep)))
        return out

    def eval_batch(self, stage, repeats=1):
        f = np.zeros(self.batch)
        _chk(self.L, self.L.topay_eval_batch(self.h, stage, repeats, _dp(f)))
        return f

    def set_trace(self, cap):
        self._trace_cap = cap
        _chk(self.L, self.L.topay_set_trace(self.h, cap))

    def get_trace(self, i):
        out = np.zeros(self._trace_cap)
        _chk(self.L, self.L.topay_get_trace(self.h, i, _dp(out)))
        return out

    def test_math(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.zeros(4 * len(a))
        _chk(self.L, self.L.topay_test_math(self.h, len(a), _dp(a), _dp(b), _dp(out)))
        return out.reshape(-1, 4)

    def last_helper_launches(self):
        n = C.c_int(0)
        _chk(self.L, self.L.topay_last_helper_launches(self.h, C.byref(n)))
        return n.value

    def last_kernel_ms(self):
        ms = C.c_double(0)
        n = C.c_int(0)
        _chk(self.L, self.L.topay_last_kernel_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value
