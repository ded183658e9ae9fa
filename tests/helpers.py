import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def golden_mps_sets():
    g = golden("mps_pairs_9q.npz")
    n = int(g["n"])
    sets = {tag: [g[f"{tag}_{k}"] for k in range(n)] for tag in ("x0", "x1", "y0", "y1")}
    return [sets["x0"], sets["x1"]], [sets["y0"], sets["y1"]], g["z"]
