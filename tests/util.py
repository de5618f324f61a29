import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "qp_*.npz")))


from allocnet_amd.synth import (random_problem, corridor_problem, qp_corridor_problem, firi_scene,  # noqa: F401
                                firi_pack)


def rel_err(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
