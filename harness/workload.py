"""ctypes front-end of the synthetic workload generator (libtopay_workload.so).

Builds the BASELINE.json configurations: seeded "tables"/"cuboids" worlds with their 2-D/3-D ESDF,
start/goal scenarios, and front-end stand-in init paths (ragged P x 10 states per candidate).
Harness only: runs on the CPU, once per map/scenario, outside every timed region.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)

TABLES, CUBOIDS = 0, 1


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libtopay_workload.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.wl_world_create.restype = C.c_void_p
        L.wl_world_create.argtypes = [C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                      c_dp, C.c_int]
        L.wl_world_esdf2d.restype = c_dp
        L.wl_world_esdf3d.restype = c_dp
        L.wl_world_esdf2d.argtypes = [C.c_void_p]
        L.wl_world_occ2d.restype = C.POINTER(C.c_int8)
        L.wl_world_occ3d.restype = C.POINTER(C.c_int8)
        L.wl_world_occ2d.argtypes = [C.c_void_p]
        L.wl_world_occ3d.argtypes = [C.c_void_p]
        L.wl_world_esdf3d.argtypes = [C.c_void_p]
        L.wl_world_esdf3d_size.argtypes = [C.c_void_p]
        L.wl_world_esdf3d_size.restype = C.c_longlong
        L.wl_set_keep_esdf3d.argtypes = [C.c_int]
        L.wl_world_destroy.argtypes = [C.c_void_p]
        L.wl_world_desc.argtypes = [C.c_void_p, c_ip, c_dp, c_dp, c_dp, c_dp]
        L.wl_sample_start_goal_xy.argtypes = [C.c_uint64, C.c_double, c_dp, c_dp]
        L.wl_sample_arm.argtypes = [C.c_void_p, C.c_uint64, c_dp]
        L.wl_sample_scenario.argtypes = [C.c_void_p, C.c_uint64, c_dp, c_dp]
        L.wl_whole_body_collision.argtypes = [C.c_void_p, c_dp]
        L.wl_whole_body_tie_slack.argtypes = [C.c_void_p, c_dp]
        L.wl_whole_body_tie_slack.restype = C.c_double
        L.wl_init_paths.argtypes = [C.c_void_p, c_dp, c_dp, C.c_int, C.c_uint64, c_dp, C.c_int, c_ip]
        L.wl_tables_batch_create.restype = C.c_void_p
        L.wl_tables_batch_create.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double,
                                             C.c_int]
        L.wl_tables_batch_destroy.argtypes = [C.c_void_p]
        L.wl_tables_batch_ntraj.argtypes = [C.c_void_p]
        L.wl_tables_batch_nstates.argtypes = [C.c_void_p]
        L.wl_tables_batch_nstates.restype = C.c_longlong
        L.wl_tables_batch_get.argtypes = [C.c_void_p, c_ip, c_ip, c_dp]
        L.wl_tables_batch_world.argtypes = [C.c_void_p, C.c_int]
        L.wl_tables_batch_world.restype = C.c_void_p
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


class World:
    def __init__(self, kind, seed=42, size_xy=20.0, size_z=1.6, res=0.1, cloud_res=0.05, keepouts=None, nthreads=0,
                 _borrowed=None):
        L = lib()
        if _borrowed is not None:
            self.h = C.c_void_p(_borrowed)
            self._owned = False
        else:
            ko = np.zeros((0, 2)) if keepouts is None else np.ascontiguousarray(keepouts, dtype=np.float64).reshape(-1, 2)
            self.h = C.c_void_p(L.wl_world_create(kind, seed, size_xy, size_z, res, cloud_res, ko.shape[0], _dp(ko), nthreads))
            self._owned = True
        self.kind = kind
        self.size_xy = size_xy
        dims = np.zeros(3, dtype=np.int32)
        origin = np.zeros(3)
        mn = np.zeros(3)
        mx = np.zeros(3)
        r = C.c_double(0)
        L.wl_world_desc(self.h, dims.ctypes.data_as(c_ip), _dp(origin), C.byref(r), _dp(mn), _dp(mx))
        self.dims, self.origin, self.res, self.min_b, self.max_b = dims, origin, r.value, mn, mx
        n2 = int(dims[0]) * int(dims[1])
        n3 = n2 * int(dims[2])
        # nthreads < 0: occupancy grids only (no CPU distance fields; scenario sampling and init paths need them)
        self.esdf2d = np.ctypeslib.as_array(L.wl_world_esdf2d(self.h), shape=(n2,)) if nthreads >= 0 or _borrowed is not None else None
        # None when the batch generator was told to drop it (TablesBatch(keep_esdf3d=...))
        self.esdf3d = np.ctypeslib.as_array(L.wl_world_esdf3d(self.h), shape=(n3,)) if L.wl_world_esdf3d_size(self.h) == n3 else None
        self.occ2d = np.ctypeslib.as_array(L.wl_world_occ2d(self.h), shape=(n2,))
        self.occ3d = np.ctypeslib.as_array(L.wl_world_occ3d(self.h), shape=(n3,))

    def close(self):
        if self.h:
            if self._owned:
                lib().wl_world_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sample_scenario(self, seed):
        s = np.zeros(10)
        g = np.zeros(10)
        ok = lib().wl_sample_scenario(self.h, seed, _dp(s), _dp(g))
        return bool(ok), s, g

    def sample_arm(self, seed, state):
        st = np.ascontiguousarray(state, dtype=np.float64).copy()
        ok = lib().wl_sample_arm(self.h, seed, _dp(st))
        return bool(ok), st

    def collision(self, state):
        st = np.ascontiguousarray(state, dtype=np.float64)
        return bool(lib().wl_whole_body_collision(self.h, _dp(st)))

    def collision_tie_slack(self, state):
        """Smallest |lhs - rhs| over all threshold comparisons of isWholeBodyCollision for this state."""
        st = np.ascontiguousarray(state, dtype=np.float64)
        return float(lib().wl_whole_body_tie_slack(self.h, _dp(st)))

    def init_paths(self, start, goal, n_cand, seed, max_states=20000):
        start = np.ascontiguousarray(start, dtype=np.float64)
        goal = np.ascontiguousarray(goal, dtype=np.float64)
        out = np.zeros(max_states * 10)
        lens = np.zeros(n_cand, dtype=np.int32)
        made = lib().wl_init_paths(self.h, _dp(start), _dp(goal), n_cand, seed, _dp(out), max_states,
                                   lens.ctypes.data_as(c_ip))
        if made < 0:
            raise RuntimeError("init path buffer too small")
        lens = lens[:made]
        tot = int(lens.sum())
        return lens.copy(), out[: tot * 10].reshape(tot, 10).copy()


def sample_start_goal_xy(seed, size_xy=20.0):
    s = np.zeros(3)
    g = np.zeros(3)
    lib().wl_sample_start_goal_xy(seed, size_xy, _dp(s), _dp(g))
    return s, g


def tables_scenario(scenario_id, n_cand, base_seed=42, **world_kw):
    """Reference flow for the 'tables' scene (planner.cpp:498-548): sample start/goal, build the map with 1x1 m
    keep-outs at both, then rejection-sample the arm joints.  Returns (world, start, goal, lens, paths)."""
    seed = base_seed + scenario_id
    for attempt in range(50):
        s3, g3 = sample_start_goal_xy(seed * 1000 + attempt, world_kw.get("size_xy", 20.0))
        w = World(TABLES, seed=seed * 1000 + attempt, keepouts=[s3[:2], g3[:2]], **world_kw)
        start = np.zeros(10)
        goal = np.zeros(10)
        start[:3] = s3
        goal[:3] = g3
        ok1, goal = w.sample_arm(seed * 7919 + 2 * attempt, goal)
        ok2, start = w.sample_arm(seed * 7919 + 2 * attempt + 1, start)
        if not (ok1 and ok2):
            w.close()
            continue
        lens, paths = w.init_paths(start, goal, n_cand, seed * 104729 + attempt)
        if len(lens) == 0:
            w.close()
            continue
        return w, start, goal, lens, paths
    raise RuntimeError("could not build a tables scenario")


def cuboids_batch(n_scenarios, n_cand, map_seed=42, base_seed=42, first_scenario=0, **world_kw):
    """n_scenarios random start/goal pairs x n_cand candidates on ONE cuboids map (BASELINE config 3).
    Returns (world, lens, paths, scenario_of_traj)."""
    w = World(CUBOIDS, seed=map_seed, **world_kw)
    all_lens, all_paths, scen = [], [], []
    sid = first_scenario
    made = 0
    while made < n_scenarios:
        ok, s, g = w.sample_scenario(base_seed + sid)
        sid += 1
        if not ok:
            continue
        lens, paths = w.init_paths(s, g, n_cand, (base_seed + sid) * 104729)
        if len(lens) < n_cand:
            continue
        all_lens.append(lens)
        all_paths.append(paths)
        scen += [made] * len(lens)
        made += 1
    return w, np.concatenate(all_lens), np.concatenate(all_paths), np.asarray(scen, dtype=np.int32)


class TablesBatch:
    """S 'tables' scenarios, one map each, n_cand candidates per scenario (benchmark_tables batch)."""

    def __init__(self, n_scenarios, n_cand, base_seed=42, size_xy=20.0, size_z=1.6, res=0.1, cloud_res=0.05, nthreads=0,
                 keep_esdf3d=None):
        """keep_esdf3d = k: only the first k scenarios keep their CPU-built 3-D distance field (World.esdf3d is None for
        the others); the occupancy grids and the 2-D field are always kept."""
        L = lib()
        L.wl_set_keep_esdf3d(-1 if keep_esdf3d is None else int(keep_esdf3d))
        try:
            self.h = C.c_void_p(L.wl_tables_batch_create(n_scenarios, n_cand, base_seed, size_xy, size_z, res, cloud_res, nthreads))
        finally:
            L.wl_set_keep_esdf3d(-1)
        nt = L.wl_tables_batch_ntraj(self.h)
        ns = L.wl_tables_batch_nstates(self.h)
        self.lens = np.zeros(nt, dtype=np.int32)
        self.scen = np.zeros(nt, dtype=np.int32)
        paths = np.zeros(ns * 10)
        L.wl_tables_batch_get(self.h, self.lens.ctypes.data_as(c_ip), self.scen.ctypes.data_as(c_ip), _dp(paths))
        self.paths = paths.reshape(ns, 10)
        self.dts = np.zeros(ns)             # getDensePath's dt per state: (x, y, theta, dt) is MCRRTs::plan's car path
        L.wl_tables_batch_get_dt.argtypes = [C.c_void_p, c_dp]
        L.wl_tables_batch_get_dt(self.h, _dp(self.dts))
        self.scenarios = sorted(set(self.scen.tolist()))
        self._worlds = {}

    def world(self, sidx):
        if sidx not in self._worlds:
            wp = lib().wl_tables_batch_world(self.h, sidx)
            self._worlds[sidx] = World(TABLES, _borrowed=wp)
        return self._worlds[sidx]

    def close(self):
        if self.h:
            for w in self._worlds.values():
                w.close()
            lib().wl_tables_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def front_end_fields(occ2d, occ3d, dims, res, chassis_radius=0.4, occ2d_critical=None):
    """CPU construction of GridMap's esdf_buffer_2d_inflate and esdf_buffer_2d_critical (grid_map.cpp:211-423) from the
    occupancy grids -- the checker of topay_build_esdf_fields.  Returns (inflate, critical)."""
    L = lib()
    nx, ny, nz = (int(x) for x in dims)
    o2 = np.ascontiguousarray(occ2d, dtype=np.int8)
    o3 = np.ascontiguousarray(occ3d, dtype=np.int8)
    oc = None if occ2d_critical is None else np.ascontiguousarray(occ2d_critical, dtype=np.int8)
    inf = np.zeros(nx * ny)
    cr = np.zeros(nx * ny)
    P8 = C.POINTER(C.c_int8)
    L.wl_edt_front_end_fields.argtypes = [P8, P8, P8, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, c_dp, c_dp]
    L.wl_edt_front_end_fields.restype = None
    L.wl_edt_front_end_fields(o2.ctypes.data_as(P8), None if oc is None else oc.ctypes.data_as(P8), o3.ctypes.data_as(P8), nx, ny, nz,
                              float(res), float(chassis_radius), inf.ctypes.data_as(c_dp), cr.ctypes.data_as(c_dp))
    return inf, cr


def dense_path(raw_xy, start_yaw, end_yaw, step_size=1.414, v_max=1.0, w_max=1.25):
    """CPU restatement of GraphSearch::getDensePath (graph_search.cpp:119-176) -> [m, 4] array (x, y, theta, dt)."""
    L = lib()
    raw = np.ascontiguousarray(raw_xy, dtype=np.float64).reshape(-1, 2)
    cap = 4 * (int(np.ceil(np.linalg.norm(np.diff(raw, axis=0), axis=1) / step_size).clip(1).sum()) + 4)
    out = np.zeros((cap, 4))
    L.wl_dense_path.argtypes = [c_dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, c_dp, C.c_int]
    L.wl_dense_path.restype = C.c_int
    n = L.wl_dense_path(raw.ctypes.data_as(c_dp), len(raw), step_size, start_yaw, end_yaw, v_max, w_max, out.ctypes.data_as(c_dp), cap)
    return out[:n].copy()


# ---- layered joint-space search (MCRRTs::plan) and Reeds-Shepp restatements: libtopay_mcrrt.so --------------------------
_MLIB = None


class McrrtParams(C.Structure):
    """== topay_mcrrt_params_t (include/topay.h), == topay_wl::McrrtParams (harness/mcrrt.hpp)"""
    _fields_ = [("goal_sample_rate", C.c_double), ("check_colli_res", C.c_double), ("rs_turning_radius", C.c_double),
                ("max_iter", C.c_int), ("max_sample_tries", C.c_int), ("node_cap", C.c_int), ("reserved", C.c_int),
                ("seed", C.c_uint64)]

    def __init__(self, **kw):
        super().__init__(0.4, 0.01, 1.0e-2, 1000, 64, 2048, 0, 42)
        for k, v in kw.items():
            setattr(self, k, v)


class McrrtNodeRec(C.Structure):
    _fields_ = [("layer", C.c_int), ("state", C.c_int), ("parent", C.c_int), ("cost", C.c_double), ("q", C.c_double * 7)]


def mlib():
    global _MLIB
    if _MLIB is None:
        path = os.path.join(_HERE, "libtopay_mcrrt.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.wl_mcrrt_plan.argtypes = [C.c_void_p, c_dp, c_dp, C.c_int, c_dp, C.POINTER(McrrtParams), C.c_uint64, C.c_int, c_dp, c_ip, c_ip,
                                    c_dp, c_dp, C.c_int, C.POINTER(McrrtNodeRec)]
        L.wl_rs_path.argtypes = [C.c_double, c_dp, c_dp, c_ip, c_dp]
        L.wl_rs_path.restype = C.c_double
        L.wl_rs_interpolate.argtypes = [C.c_double, c_dp, c_dp, C.c_double, c_dp]
        L.wl_rs_interpolate.restype = None
        L.wl_mcrrt_u01.argtypes = [C.c_uint64] * 4
        L.wl_mcrrt_u01.restype = C.c_double
        _MLIB = L
    return _MLIB


def mcrrt_plan(world, start, end, car_path, params=None, inst=0, track_slack=False, want_nodes=True):
    """CPU restatement of MCRRTs::plan (mcrrts.cpp:5-231) on `world`.  car_path: [L, 4] (x, y, theta, dt).  Returns a dict:
    status (1 path / 0 none / -1 node pool full), wb_path [m, 10], stats (8 ints), c_max, min_slack, nodes (structured array:
    layer, state, parent, cost, q -- in creation order)."""
    L = mlib()
    prm = params or McrrtParams()
    cp = np.ascontiguousarray(car_path, dtype=np.float64).reshape(-1, 4)
    st = np.ascontiguousarray(start, dtype=np.float64)
    en = np.ascontiguousarray(end, dtype=np.float64)
    wb = np.zeros((len(cp), 10))
    wl_, stats = C.c_int(0), np.zeros(8, dtype=np.int32)
    cmax, slack = C.c_double(0), C.c_double(0)
    nodes = (McrrtNodeRec * prm.node_cap)() if want_nodes else None
    s = L.wl_mcrrt_plan(world.h, _dp(st), _dp(en), len(cp), _dp(cp), C.byref(prm), int(inst), int(track_slack), _dp(wb), C.byref(wl_),
                        stats.ctypes.data_as(c_ip), C.byref(cmax), C.byref(slack), prm.node_cap, nodes)
    out = dict(status=int(s), wb_path=wb[:wl_.value].copy(), stats=stats, c_max=cmax.value, min_slack=slack.value, nodes=None)
    if want_nodes:
        rec = np.dtype([("layer", "<i4"), ("state", "<i4"), ("parent", "<i4"), ("pad", "<i4"), ("cost", "<f8"), ("q", "<f8", (7,))])
        assert rec.itemsize == C.sizeof(McrrtNodeRec)
        out["nodes"] = np.frombuffer(nodes, dtype=rec, count=prm.node_cap)[:stats[1]].copy()
    return out


def plan2d_jps(world, start_xy, end_xy, threshold=0.5, cap=4096):
    """CPU restatement of GraphSearch::plan2dJPS (graph_search.cpp:53-117) -> ([m, 2] path, empty when none; (expanded nodes,
    jump points of the raw path))."""
    L = mlib()
    L.wl_plan2d_jps.argtypes = [C.c_void_p, c_dp, c_dp, C.c_double, C.c_int, c_dp, c_ip]
    a = np.ascontiguousarray(start_xy, dtype=np.float64)
    b = np.ascontiguousarray(end_xy, dtype=np.float64)
    out = np.zeros((cap, 2))
    st = np.zeros(2, dtype=np.int32)
    n = L.wl_plan2d_jps(world.h, _dp(a), _dp(b), float(threshold), cap, _dp(out), st.ctypes.data_as(c_ip))
    return out[:min(n, cap)].copy(), (int(st[0]), int(st[1]))


def rs_path(from_pose, to_pose, rho=1.0e-2):
    """ompl::base::ReedsSheppStateSpace(rho).reedsShepp(from, to) restated: (word 0..17, five signed lengths, distance)."""
    L = mlib()
    a = np.ascontiguousarray(from_pose, dtype=np.float64)
    b = np.ascontiguousarray(to_pose, dtype=np.float64)
    t, ln = C.c_int(0), np.zeros(5)
    d = L.wl_rs_path(rho, _dp(a), _dp(b), C.byref(t), _dp(ln))
    return t.value, ln, d


def rs_interpolate(from_pose, to_pose, t, rho=1.0e-2):
    L = mlib()
    a = np.ascontiguousarray(from_pose, dtype=np.float64)
    b = np.ascontiguousarray(to_pose, dtype=np.float64)
    out = np.zeros(3)
    L.wl_rs_interpolate(rho, _dp(a), _dp(b), float(t), _dp(out))
    return out
