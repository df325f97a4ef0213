"""TEST INFRASTRUCTURE — ctypes front-end of the CPU oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
See oracle/topay_oracle.hpp for the restatement and its reference citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


def _dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_ip) if a is not None else None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_create.restype = C.c_void_p
        L.orc_eval.restype = C.c_double
        L.orc_penalty.restype = C.c_double
        L.orc_traj_cost.restype = C.c_double
        L.orc_optimize_batch.restype = C.c_double
        _LIB = L
    return _LIB


class MapView:
    """Plain description of an ESDF map (same fields as topay_map_desc_t + buffers)."""

    def __init__(self, origin, res, dims, min_b, max_b, esdf2d, esdf3d):
        self.origin = np.ascontiguousarray(origin, dtype=np.float64)
        self.res = float(res)
        self.dims = np.ascontiguousarray(dims, dtype=np.int32)
        self.min_b = np.ascontiguousarray(min_b, dtype=np.float64)
        self.max_b = np.ascontiguousarray(max_b, dtype=np.float64)
        self.esdf2d = np.ascontiguousarray(esdf2d, dtype=np.float64)
        self.esdf3d = np.ascontiguousarray(esdf3d, dtype=np.float64)


class Oracle:
    """One `MomaTrajOpt` instance of the restatement."""

    def __init__(self, map_view=None):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create())
        self.map = None
        if map_view is not None:
            self.set_map(map_view)

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def set_param(self, name, v):
        r = self.L.orc_set_param(self.h, name.encode(), C.c_double(v))
        if r != 0:
            raise KeyError(name)

    def set_map(self, m):
        self.map = m  # keep buffers alive
        self.L.orc_set_map(self.h, _dp(m.origin), C.c_double(m.res), _ip(m.dims), _dp(m.min_b), _dp(m.max_b),
                           _dp(m.esdf2d), _dp(m.esdf3d))

    def set_init_traj(self, init_path, bvel=None, bacc=None):
        p = np.ascontiguousarray(init_path, dtype=np.float64).reshape(-1, 10)
        bvel = np.zeros(20) if bvel is None else np.ascontiguousarray(bvel, dtype=np.float64)
        bacc = np.zeros(20) if bacc is None else np.ascontiguousarray(bacc, dtype=np.float64)
        n = self.L.orc_set_init_traj(self.h, _dp(p), p.shape[0], _dp(bvel), _dp(bacc))
        self.n = n
        self.N = self.L.orc_piece_num(self.h)
        return n

    def get_x(self):
        x = np.zeros(self.n)
        self.L.orc_get_x(self.h, _dp(x))
        return x

    def set_x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        self.L.orc_set_x(self.h, _dp(x))

    def get_init_state(self):
        s = np.zeros(27)
        e = np.zeros(27)
        xy = np.zeros(2 * self.N)
        ss = np.zeros(10)
        es = np.zeros(10)
        past = C.c_int(0)
        self.L.orc_get_init_state(self.h, _dp(s), _dp(e), _dp(xy), _dp(ss), _dp(es), C.byref(past))
        return dict(start_pva=s, end_pva=e, init_inner_xy=xy.reshape(-1, 2), start_state=ss, end_state=es,
                    s1_past=past.value)

    def set_alm(self, lam, rho):
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        self.L.orc_set_alm(self.h, _dp(lam), _dp(rho))

    def alm_state(self):
        """(lambda0, lambda1, rho0, rho1) as the last optimize() left them (moma_traj_opt.cpp:451-459)."""
        a = np.zeros(4)
        self.L.orc_get_alm(self.h, _dp(a))
        return a

    def eval(self, stage, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.zeros_like(x)
        f = self.L.orc_eval(self.h, stage, _dp(x), _dp(g))
        return f, g

    def penalty(self, stage, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        gdC = np.zeros(6 * self.N * 9)
        gdT = np.zeros(self.N)
        c = self.L.orc_penalty(self.h, stage, _dp(x), _dp(gdC), _dp(gdT))
        return c, gdC.reshape(9, 6 * self.N), gdT

    def final_xy_error(self):
        e = np.zeros(2)
        self.L.orc_get_final_xy_error(self.h, _dp(e))
        return e

    def debug_terms(self):
        t = np.zeros(13)
        self.L.orc_get_debug_terms(self.h, _dp(t))
        names = ["jerk", "time", "chassis_colli", "moment", "acc", "domega", "mani_colli", "self_colli", "mani_pos",
                 "mani_vel", "mani_acc", "mean_time", "endp"]
        return dict(zip(names, t))

    def coeffs(self):
        c = np.zeros(6 * self.N * 9)
        T = np.zeros(self.N)
        self.L.orc_get_coeffs(self.h, _dp(c), _dp(T))
        return c.reshape(9, 6 * self.N), T

    def optimize(self):
        return bool(self.L.orc_optimize(self.h))

    def optimize_warm(self):
        """Stage-2 ALM loop only, from the x of set_x() and the (lambda, rho) of set_alm()."""
        return bool(self.L.orc_optimize_warm(self.h))

    def optimize_trace(self, cap=100000):
        tr = np.zeros(cap)
        ok = C.c_int(0)
        n = self.L.orc_optimize_trace(self.h, _dp(tr), cap, C.byref(ok))
        return bool(ok.value), tr[:min(n, cap)]

    def stats(self):
        s = np.zeros(8, dtype=np.int32)
        self.L.orc_get_stats(self.h, _ip(s))
        keys = ["stage1_ret", "stage1_iters", "stage1_evals", "stage2_last_ret", "stage2_iters", "stage2_evals",
                "alm_outer", "sum_bound"]
        return dict(zip(keys, s.tolist()))

    def traj_cost(self):
        return self.L.orc_traj_cost(self.h)

    def get_traj(self):
        N = self.L.orc_piece_num(self.h)
        d = np.zeros(N)
        c = np.zeros(N * 54)
        k = np.zeros((N + 1) * 2)
        self.L.orc_get_traj(self.h, _dp(d), _dp(c), _dp(k))
        return d, c.reshape(N, 9, 6), k.reshape(N + 1, 2)

    def check_feasible(self):
        """printConstraintsSituations / checkFeasible on the held trajectory -> (feasible, strict, report[38])."""
        rep = np.zeros(38)
        st = C.c_int(0)
        f = self.L.orc_check_feasible(self.h, _dp(rep), C.byref(st))
        return bool(f), bool(st.value), rep

    def car_seq(self):
        seq = np.zeros((4096, 4))
        n = self.L.orc_car_seq(self.h, _dp(seq), 4096)
        return seq[:n].copy()

    def mesh_pose(self, state):
        """MomaParam::getMeshPose (moma_param.h:724-790): 11 x (x, y, z, qw, qx, qy, qz)."""
        out = np.zeros(77)
        self.L.orc_mesh_pose(self.h, _dp(np.ascontiguousarray(state, dtype=np.float64)), _dp(out))
        return out.reshape(11, 7)

    def mesh_traj(self, res=1000):
        """Planner::toMeshMsg (planner.cpp:2003-2056): (parts n x 11 x 7, yaws n, arc_lengths n)."""
        cap = res + 8
        parts, yaws, arcs = np.zeros(cap * 77), np.zeros(cap), np.zeros(cap)
        self.L.orc_mesh_traj.restype = C.c_int
        n = self.L.orc_mesh_traj(self.h, C.c_int(res), C.c_int(cap), _dp(parts), _dp(yaws), _dp(arcs))
        return parts[:n * 77].reshape(n, 11, 7), yaws[:n], arcs[:n]

    def optimize_device_order(self, eval_fn, epl=12, nw=1):
        """This restatement's solver logic with the vector arithmetic in the device's order and the cost / gradient from
        `eval_fn(stage, x, lam, rho) -> (f, g, fxe)` (the device's evaluation hook): must reproduce a device solve bit for
        bit.  epl / nw = decision-vector elements per thread and waves per trajectory of the kernel that solved the
        candidate (topay_class_of; with one wave any epl >= the kernel's gives the same bits).  Returns success."""
        CB = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))

        def cb(_user, stage, n, x, lam, rho, g, fxe):
            xv = np.ctypeslib.as_array(x, shape=(n,)).copy()
            f, gv, e = eval_fn(stage, xv, [lam[0], lam[1]], [rho[0], rho[1]])
            np.ctypeslib.as_array(g, shape=(n,))[:] = gv
            fxe[0], fxe[1] = float(e[0]), float(e[1])
            return float(f)

        self._cb = CB(cb)
        self.L.orc_optimize_device_order.restype = C.c_int
        self.L.orc_optimize_device_order.argtypes = [C.c_void_p, C.c_int, C.c_int, CB, C.c_void_p]
        return bool(self.L.orc_optimize_device_order(self.h, epl, nw, self._cb, None))

    def traj_state(self, t):
        s = np.zeros(10)
        self.L.orc_traj_state(self.h, C.c_double(t), _dp(s))
        return s


def optimize_batch(m, path_len, paths, bvel=None, bacc=None, nthreads=1, alm_max_outer=0, maxN=0):
    """Thread-pool batch solve (cpu_baseline leg). Returns dict with success, cost, n_pieces, stats, seconds."""
    L = lib()
    path_len = np.ascontiguousarray(path_len, dtype=np.int32)
    B = len(path_len)
    paths = np.ascontiguousarray(paths, dtype=np.float64)
    bvel = np.zeros(B * 20) if bvel is None else np.ascontiguousarray(bvel, dtype=np.float64)
    bacc = np.zeros(B * 20) if bacc is None else np.ascontiguousarray(bacc, dtype=np.float64)
    success = np.zeros(B, dtype=np.int32)
    cost = np.zeros(B)
    npc = np.zeros(B, dtype=np.int32)
    stats = np.zeros(B * 8, dtype=np.int32)
    dur = coef = knots = None
    if maxN > 0:
        dur = np.zeros(B * maxN)
        coef = np.zeros(B * maxN * 54)
        knots = np.zeros(B * (maxN + 1) * 2)
    secs = L.orc_optimize_batch(_dp(m.origin), C.c_double(m.res), _ip(m.dims), _dp(m.min_b), _dp(m.max_b),
                                _dp(m.esdf2d), _dp(m.esdf3d), B, _ip(path_len), _dp(paths), _dp(bvel), _dp(bacc),
                                nthreads, alm_max_outer, _ip(success), _dp(cost), _ip(npc), _ip(stats), maxN,
                                _dp(dur), _dp(coef), _dp(knots))
    out = dict(success=success, cost=cost, n_pieces=npc, stats=stats.reshape(B, 8), seconds=secs)
    if maxN > 0:
        out.update(durations=dur.reshape(B, maxN), coeffs=coef.reshape(B, maxN, 9, 6), knots=knots.reshape(B, maxN + 1, 2))
    return out


def optimize_batch_maps(maps, map_id, path_len, paths, nthreads=1, gate=False):
    """Thread-pool batch solve where trajectory b runs against maps[map_id[b]] (cpu_baseline leg of bench.py)."""
    L = lib()
    M = len(maps)
    origin = np.ascontiguousarray(np.stack([m.origin for m in maps]), dtype=np.float64)
    res = np.ascontiguousarray([m.res for m in maps], dtype=np.float64)
    dims = np.ascontiguousarray(np.stack([m.dims for m in maps]), dtype=np.int32)
    mn = np.ascontiguousarray(np.stack([m.min_b for m in maps]), dtype=np.float64)
    mx = np.ascontiguousarray(np.stack([m.max_b for m in maps]), dtype=np.float64)
    P = C.POINTER(C.c_double)
    e2 = (P * M)(*[_dp(m.esdf2d) for m in maps])
    e3 = (P * M)(*[_dp(m.esdf3d) for m in maps])
    map_id = np.ascontiguousarray(map_id, dtype=np.int32)
    path_len = np.ascontiguousarray(path_len, dtype=np.int32)
    paths = np.ascontiguousarray(paths, dtype=np.float64)
    B = len(path_len)
    success = np.zeros(B, dtype=np.int32)
    cost = np.zeros(B)
    npc = np.zeros(B, dtype=np.int32)
    stats = np.zeros(B * 8, dtype=np.int32)
    each = np.zeros(B)
    dur = np.zeros(B) if gate else None
    gt = np.zeros(B, dtype=np.int32) if gate else None
    L.orc_optimize_batch_maps.restype = C.c_double
    secs = L.orc_optimize_batch_maps(M, _dp(origin), _dp(res), _ip(dims), _dp(mn), _dp(mx), e2, e3, _ip(map_id), B,
                                     _ip(path_len), _dp(paths), nthreads, _ip(success), _dp(cost), _ip(npc), _ip(stats),
                                     _dp(each), _dp(dur), _ip(gt))
    out = dict(success=success, cost=cost, n_pieces=npc, stats=stats.reshape(B, 8), seconds=secs, seconds_each=each)
    if gate:   # total duration and printConstraintsSituations verdict of every returned trajectory
        out.update(total_duration=dur, gate=gt)
    return out


def stage1_batch_maps(maps, map_id, path_len, paths, nthreads=1, xstride=10 * 128 - 8):
    """Stage 1 alone for a batch (the ALM loop is not entered): converged x per candidate (rows of xstride), cost, counters."""
    L = lib()
    M = len(maps)
    origin = np.ascontiguousarray(np.stack([m.origin for m in maps]), dtype=np.float64)
    res = np.ascontiguousarray([m.res for m in maps], dtype=np.float64)
    dims = np.ascontiguousarray(np.stack([m.dims for m in maps]), dtype=np.int32)
    mn = np.ascontiguousarray(np.stack([m.min_b for m in maps]), dtype=np.float64)
    mx = np.ascontiguousarray(np.stack([m.max_b for m in maps]), dtype=np.float64)
    P = C.POINTER(C.c_double)
    e2 = (P * M)(*[_dp(m.esdf2d) for m in maps])
    e3 = (P * M)(*[_dp(m.esdf3d) for m in maps])
    map_id = np.ascontiguousarray(map_id, dtype=np.int32)
    path_len = np.ascontiguousarray(path_len, dtype=np.int32)
    paths = np.ascontiguousarray(paths, dtype=np.float64)
    B = len(path_len)
    x = np.zeros((B, xstride))
    cost = np.zeros(B)
    npc = np.zeros(B, dtype=np.int32)
    stats = np.zeros(B * 3, dtype=np.int32)
    L.orc_stage1_batch_maps.restype = C.c_double
    secs = L.orc_stage1_batch_maps(M, _dp(origin), _dp(res), _ip(dims), _dp(mn), _dp(mx), e2, e3, _ip(map_id), B, _ip(path_len),
                                   _dp(paths), nthreads, xstride, _dp(x), _dp(cost), _ip(npc), _ip(stats))
    return dict(x=x, cost=cost, n_pieces=npc, stats=stats.reshape(B, 3), seconds=secs)


def group_cancel(group_id, clock, accepted, budget=2400):
    """The planner's cancellation rule (planner.cpp:829-952) on work clocks: bool array, True = interrupted."""
    L = lib()
    g = np.ascontiguousarray(group_id, dtype=np.int32)
    c = np.ascontiguousarray(clock, dtype=np.int64)
    a = np.ascontiguousarray(accepted, dtype=np.int32)
    out = np.zeros(len(g), dtype=np.int32)
    L.orc_group_cancel(len(g), _ip(g), c.ctypes.data_as(C.POINTER(C.c_longlong)), _ip(a), C.c_longlong(int(budget)), _ip(out))
    return out.astype(bool)
