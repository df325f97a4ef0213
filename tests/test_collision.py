"""GridMap::isWholeBodyCollision (grid_map.h:613-650) on the device against the CPU restatement of the workload
harness (harness/workload.hpp: RobotModel::isWholeBodyCollision, same file:line citations)."""
import numpy as np
import pytest

from conftest import EMU_LIB, set_map
from topay_amd import api
from harness import workload as wl


def _states(w, n, seed):
    rng = np.random.default_rng(seed)
    st = np.zeros((n, 10))
    st[:, 0] = rng.uniform(w.min_b[0] - 0.5, w.max_b[0] + 0.5, n)      # some outside the map
    st[:, 1] = rng.uniform(w.min_b[1] - 0.5, w.max_b[1] + 0.5, n)
    st[:, 2] = rng.uniform(-np.pi, np.pi, n)
    qmax = np.array([3.1, 2.26, 3.1, 2.355, 3.1, 2.23, 6.28])
    st[:, 3:] = rng.uniform(-1.05, 1.05, (n, 7)) * qmax               # some beyond the joint limits
    st[: n // 4, 3:] *= 0.2                                            # and a share of tucked-in arms that can be free
    return st


def _check(opt, w, n, seed):
    st = _states(w, n, seed)
    got = opt.whole_body_collision(st)
    ref = np.array([w.collision(s) for s in st])
    assert 0.02 < (~ref).mean() < 0.98, "the sample must contain free and colliding states"
    # the decisions are threshold tests on distances that agree to ~1e-15: allow none to differ except exact ties
    assert (got == ref).mean() > 0.999
    return (got != ref).sum()


def test_whole_body_collision_kernel_sources_on_cpu():
    w = wl.World(wl.CUBOIDS, seed=42)
    emu = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(emu, w)
    assert _check(emu, w, 1500, 1) <= 1
    w.close()


@pytest.mark.gpu
def test_whole_body_collision_on_gpu():
    w = wl.World(wl.TABLES, seed=43)
    gpu = api.MomaTrajOptBatch(device=0)
    set_map(gpu, w)
    assert _check(gpu, w, 20000, 2) <= 2
    emu = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(emu, w)
    st = _states(w, 2000, 3)
    assert (gpu.whole_body_collision(st) == emu.whole_body_collision(st)).all()   # bit-identical decisions
    w.close()
