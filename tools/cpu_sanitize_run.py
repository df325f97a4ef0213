import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topay_amd import api
from harness import workload as wl
LIB = sys.argv[1]
w, lens, paths, scen = wl.cuboids_batch(3, 2)
p = api.default_params(api.load(LIB))
p.s2_lbfgs.max_iterations = 25
p.alm_max_outer = 2
opt = api.MomaTrajOptBatch(params=p, lib_path=LIB)
opt.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, 0)
opt.optimizeTraj(lens, paths)
print("solve ok", opt.stats()[:, :3].tolist())
print("gate", opt.check_feasible())
# tables scenario with more candidates (persistent loop) and a map built by the EDT kernels
w1, _, _, lens1, paths1 = wl.tables_scenario(0, 12)
o2 = api.MomaTrajOptBatch(params=p, lib_path=LIB)
o2.build_esdf(w1.origin, w1.res, w1.dims, w1.min_b, w1.max_b, w1.occ2d, w1.occ3d)
o2.optimizeTraj(lens1, paths1)
print("tables ok", o2.stats()[:, :3].tolist())
st = np.zeros((4, 10)); st[:, 0] = np.linspace(-3, 3, 4)
print("wb", o2.whole_body_collision(st))
# output formats and front-end slice: mesh poses / MeshTraj, batched getTraj, PolyTraj layout, dense path, edge check,
# all five fields, per-term costs through topay_set_params
best = np.nonzero(opt.traj_cost == opt.traj_cost)[0][:2].astype(np.int32)
if len(best):
    print("getTrajs", len(opt.getTrajs(best)))
    print("mesh traj", [a.shape for a in opt.mesh_traj(int(best[0]), 200)])
    print("polytraj", len(opt.polytraj_msg(int(best[0]))))
rng = np.random.default_rng(0)
print("mesh poses", opt.mesh_poses(np.concatenate([rng.uniform(-8, 8, (9, 3)), rng.uniform(-7, 7, (9, 7))], axis=1)).shape)
print("terms", sorted(opt.cost_terms(0, opt.get_x(0), [0.1, 0.2], [1e4, 1e4]).items())[:2])
o2.build_esdf_fields(w1.origin, w1.res, w1.dims, w1.min_b, w1.max_b, w1.occ2d, None, w1.occ3d)
print("fields", [f.shape for f in o2.get_map_fields(0)[:4]])
raw = np.array([[0.0, 0.0], [1.0, 0.2], [2.5, 1.0], [2.6, 3.0]])
print("dense", np.asarray(o2.dense_path([raw], [0.3], [1.0])[0]).shape)
# round 3: the joint-space search + Reeds-Shepp, several waves per trajectory, cancellation, parameter file, shared maps
tb = wl.TablesBatch(2, 4, base_seed=31337, nthreads=2)
o3 = api.MomaTrajOptBatch(params=p, lib_path=LIB)
slot = {s: k for k, s in enumerate(tb.scenarios)}
for s_ in tb.scenarios:
    ww = tb.world(s_)
    o3.set_map(ww.origin, ww.res, ww.dims, ww.min_b, ww.max_b, ww.esdf2d, ww.esdf3d, slot[s_])
offs = np.concatenate([[0], np.cumsum(tb.lens)])
car = np.c_[tb.paths[:, :3], tb.dts]
mid = np.array([slot[s] for s in tb.scen], dtype=np.int32)
wbs, mst, _ = o3.mcrrt_plan(tb.lens, car, tb.paths[offs[:-1]], tb.paths[offs[1:] - 1], o3.mcrrt_params(seed=3, max_iter=150, node_cap=64), map_ids=mid)
print("mcrrt", mst[:, :3].tolist())
print("rs", o3.reeds_shepp(rng.uniform(-1, 1, (50, 3)), rng.uniform(-1, 1, (50, 3)), rng.uniform(0, 1, 50))[0][:3])
o4 = api.MomaTrajOptBatch(params=p, lib_path=LIB)
o4.share_maps(o3, 0, len(tb.scenarios))
o4.set_init_traj(tb.lens, tb.paths, map_ids=mid)
o4.set_groups(tb.scen, 200)
o4.optimize()
print("cancel", o4.interrupted().tolist())
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import serpentine_path
lp = [serpentine_path(34.0), serpentine_path(99.0)]
p2 = api.default_params(api.load(LIB)); p2.s1_lbfgs.max_iterations = 3; p2.s2_lbfgs.max_iterations = 3; p2.alm_max_outer = 1
o5 = api.MomaTrajOptBatch(params=p2, lib_path=LIB)
o5.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, 0)
o5.optimizeTraj(np.array([len(q) for q in lp], dtype=np.int32), np.concatenate(lp))
print("multi-wave", o5.n_pieces().tolist(), o5.stats()[:, :3].tolist())
lq = [serpentine_path(9.0), serpentine_path(19.5), serpentine_path(27.0)]
o6 = api.MomaTrajOptBatch(params=p2, lib_path=LIB)
o6.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, 0)
o6.set_init_traj(np.array([len(q) for q in lq], dtype=np.int32), np.concatenate(lq))
o6.set_latency_mode(2)
o6.optimize()
print("helper waves", o6.n_pieces().tolist(), o6.last_helper_launches(), o6.stats()[:, :3].tolist())
print("yaml", api.params_from_yaml_c("second_stage:\n  time_weight: 51.0\n  lbfgs: {past: 4}\n", lib=api.load(LIB))[0].s2_time_weight)
print("jps", [len(q) for q in o3.plan2d_jps(tb.paths[offs[:-1]][:3, :2], tb.paths[offs[1:] - 1][:3, :2], 0.5, map_ids=mid[:3])[0]])
