"""Interpolation tables against the known answers of the reference's own unit tests
(test/unit/src/interpolation_test.F90: data5 table, tolerance 1e-9)."""
import numpy as np
import pytest

from waiwera_amd.interpolation import Table

DATA5 = np.array([[0.0, 1.0], [2.1, 2.0], [3.7, 0.5], [6.3, -1.1], [8.9, -0.1]])
TOL = 1.0e-9


@pytest.mark.parametrize("kind,cases", [
    ("linear", [(-0.5, 1.0, 0), (0.0, 1.0, 0), (1.0, 1.4761904761904763, 1), (4.5, 0.007692307692307665, 3),
                (3.6, 0.59375, 2), (6.3, -1.1, 4), (10.0, -0.1, 5)]),
    ("step", [(-0.5, 1.0, 0), (0.0, 1.0, 0), (1.0, 1.0, 1), (4.5, 0.5, 3), (3.6, 2.0, 2), (6.3, -1.1, 4),
              (10.0, -0.1, 5)]),
])
def test_interpolate(kind, cases):
    t = Table(DATA5, kind)
    for x, y, idx in cases:
        assert abs(t.interpolate(x)[0] - y) < TOL, (kind, x)
        assert t.find(x) == idx, (kind, x)


@pytest.mark.parametrize("kind,averaging,cases", [
    ("linear", "endpoint", [((-0.5, -0.1), 1.0), ((-0.5, 0.1), 1.0238095238095237), ((0.1, 2.0), 1.5),
                            ((0.1, 3.0), 1.1019345238095237), ((3.1, 7.0), 0.11586538461538454),
                            ((8.0, 12.0), -0.27307692307692316), ((1.0, 1.0), 1.4761904761904763)]),
    ("step", "endpoint", [((-0.5, -0.1), 1.0), ((-0.5, 0.1), 1.0), ((0.1, 2.0), 1.0), ((0.1, 3.0), 1.5),
                          ((3.1, 7.0), 0.45), ((8.0, 12.0), -0.6), ((1.0, 1.0), 1.0)]),
    ("linear", "integrate", [((-0.5, -0.1), 1.0), ((-0.5, 0.1), 1.003968253968254), ((0.1, 2.0), 1.5),
                             ((0.1, 3.0), 1.5406660509031198), ((3.1, 7.0), -0.2530818540433925),
                             ((8.0, 12.0), -0.1389423076923077), ((9.0, 12.0), -0.1),
                             ((1.0, 1.0), 1.4761904761904763)]),
    ("step", "integrate", [((-0.5, -0.1), 1.0), ((-0.5, 0.1), 1.0), ((0.1, 2.0), 1.0), ((0.1, 3.0), 3.8 / 2.9),
                           ((3.1, 7.0), 1.73 / 3.9), ((8.0, 12.0), -0.325), ((1.0, 1.0), 1.0)]),
])
def test_average(kind, averaging, cases):
    t = Table(DATA5, kind, averaging)
    for interval, y in cases:
        assert abs(t.average(interval)[0] - y) < TOL, (kind, averaging, interval)


def test_array_values_and_bad_tables():
    t = Table([[0.0, 1.0, 2.0, 3.0], [1.0, 2.0, 3.0, 4.0], [2.0, 3.0, 4.0, 5.0]])
    assert t.dim == 3
    np.testing.assert_allclose(t.interpolate(0.5), [1.5, 2.5, 3.5], atol=TOL)
    np.testing.assert_allclose(t.interpolate(1.5), [2.5, 3.5, 4.5], atol=TOL)
    with pytest.raises(ValueError):
        Table([[0.0, 1.0], [2.0, 1.0], [1.0, 3.0]])
    with pytest.raises(ValueError):
        Table([[0.0, 1.0], [1.0, 1.0], [1.0, 3.0]])
