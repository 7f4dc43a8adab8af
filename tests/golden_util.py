"""Loaders for the committed golden fixtures (tests/golden/, produced by make_golden.py from
the reference's own control flow)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


def load_graph_npz(name):
    z = np.load(os.path.join(GOLDEN_DIR, name))
    return {k: z[k] for k in z.files}


def oracle_graph(O, z):
    return O.Graph(int(z["levels"].shape[0]), int(z["adj0"].shape[1]), int(z["adjU"].shape[1]),
                   int(z["max_level"]), int(z["entry"]), np.ascontiguousarray(z["levels"]),
                   np.ascontiguousarray(z["adj0"]), np.ascontiguousarray(z["upper_row"]),
                   np.ascontiguousarray(z["adjU"]))


def f32_distance(a, o):
    a = np.asarray(a, np.float32)
    o = np.asarray(o, np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        d = np.float32(1.0) - a / o
    return np.where(np.asarray(o) == 0, np.float32(0), d).astype(np.float32)
