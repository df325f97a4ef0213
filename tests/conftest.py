import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_LIB = os.path.join(ROOT, "tests", "emu", "libtopay_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def built_libs():
    """CPU-side libraries: oracle, workload harness, lane-emulator build of the kernels (all test infrastructure)."""
    for d in ("oracle", os.path.join("topay_amd", "harness"), os.path.join("tests", "emu")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, d)])
    return True


@pytest.fixture(scope="session")
def cuboids_small(built_libs):
    """3 scenarios x 2 candidates on the seed-42 cuboids map: N = 4, 7, 8, 10, 8, 11 pieces."""
    from oracle import oracle as orc
    from topay_amd.harness import workload as wl

    w, lens, paths, scen = wl.cuboids_batch(3, 2)
    m = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
    offs = np.concatenate([[0], np.cumsum(lens)])
    return dict(world=w, map=m, lens=lens, paths=paths, offs=offs)


def set_map(opt, w, map_id=0):
    opt.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, map_id)
