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
    for d in ("oracle", "harness", os.path.join("tests", "emu")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, d)])
    return True


@pytest.fixture(scope="session")
def cuboids_small(built_libs):
    """3 scenarios x 2 candidates on the seed-42 cuboids map: N = 4, 7, 8, 10, 8, 11 pieces."""
    from oracle import oracle as orc
    from harness import workload as wl

    w, lens, paths, scen = wl.cuboids_batch(3, 2)
    m = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
    offs = np.concatenate([[0], np.cumsum(lens)])
    return dict(world=w, map=m, lens=lens, paths=paths, offs=offs)


def set_map(opt, w, map_id=0):
    opt.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, map_id)


def serpentine_path(total_len, step=1.3, row=17.0, dy=1.0):
    """Synthetic long init path inside the 20 x 20 m map: rows along x joined by short steps in y, `total_len` metres in
    all.  Its time allocation needs about 0.97 pieces per metre: 34 m -> N = 33, 50 m -> 48, 66 m -> 64 (the most this
    build solves), 67 m -> 65."""
    pts = [np.array([-8.5, -8.5])]
    done, d = 0.0, 1
    while done < total_len:
        run = 0.0
        while run < row - 1e-9 and done < total_len:
            s = min(step, row - run, total_len - done)
            pts.append(pts[-1] + np.array([d * s, 0.0]))
            run += s
            done += s
        if done >= total_len:
            break
        s = min(dy, total_len - done)
        pts.append(pts[-1] + np.array([0.0, s]))
        done += s
        d = -d
    states = []
    for k, p in enumerate(pts):
        th = 0.0 if k == 0 else np.arctan2(*(pts[k] - pts[k - 1])[::-1])
        q = np.linspace(0.2, -0.3, 7) * (k / max(1, len(pts) - 1))
        states.append(np.concatenate([p, [th], q]))
    return np.array(states)
