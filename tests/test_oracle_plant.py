"""CPU: plant model and track transforms of the oracle (oracle/plant_ref.py) against vectors produced by the
reference's own Simulator / Map classes (tests/golden/plant_and_transforms.npz)."""
import numpy as np
import pytest

from oracle import lpv_ref as L, plant_ref as PR
from tests._golden import load


@pytest.mark.parametrize("shape", ["oval", "L_shape"])
def test_transforms(shape):
    g = load("plant_and_transforms")
    tab = L.TrackMap(shape, 0.2).PointAndTangent
    glob = np.array([PR.get_global_position(tab, s, e) for s, e in zip(g[shape + "_s"], g[shape + "_ey"])], float)
    assert np.max(np.abs(glob - g[shape + "_glob"])) <= 1e-12
    loc = np.array([PR.get_local_position(tab, float(g[shape + "_hw"]), float(g[shape + "_slack"]), *p) for p in g[shape + "_pts"]], float)
    assert np.max(np.abs(loc - g[shape + "_loc"])) <= 1e-12
    assert set(np.unique(loc[:, 3])) == {0.0, 1.0}          # inside and outside points are both covered
    # round trip (the reference's never-called unityTestChangeOfCoordinates, TRACK:433-460): tolerance 1e-8 on d^2
    inside = loc[:200, 3] == 1
    back = np.array([PR.get_global_position(tab, s, e)[:2] for s, e in loc[:200][inside][:, :2]], float)
    assert np.max(np.sum((back - g[shape + "_pts"][:200][inside][:, :2]) ** 2, axis=1)) <= 1e-8


def test_simulator_model():
    g = load("plant_and_transforms")
    st = g["sim_init"].copy()
    for u, ref in zip(g["sim_u"], g["sim_states"]):
        st = PR.simulator_f(st, u)
        assert np.max(np.abs(st - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))
