"""Helpers to read the flat golden .npz files written by tests/golden/make_golden.py."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def cases(name):
    d = load(name)
    out = []
    for i in range(int(d["n_cases"])):
        pre = "c%02d_" % i
        out.append({k[len(pre):]: d[k] for k in d.files if k.startswith(pre)})
    return out
