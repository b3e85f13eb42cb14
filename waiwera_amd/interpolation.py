"""Interpolation tables for time-dependent input data (source rate / enthalpy tables).

Host-side mirror of the reference's interpolation_table_type (src/interpolation.F90): a coordinate
axis with find() semantics val(index) <= x < val(index+1), index 0 below and size at/above the range
(:202-222), linear (:395-404) or step (:715-720) interpolants, clamped outside the data (:502-508),
and interval averages by end points (:585-602) or by integration (:606-658).  pchip is not here.
"""
import bisect

import numpy as np


class Table:
    def __init__(self, data, interpolation="linear", averaging="integrate"):
        data = np.atleast_2d(np.asarray(data, dtype=np.float64))
        if interpolation not in ("linear", "step"):
            raise NotImplementedError("interpolation %r" % interpolation)
        if averaging not in ("endpoint", "integrate"):
            raise ValueError("averaging %r" % averaging)
        self.x = data[:, 0].copy()
        self.v = data[:, 1:].copy()
        if np.any(np.diff(self.x) <= 0.0):
            # unsorted or duplicate abscissae are an error in the reference too (:330-352)
            raise ValueError("interpolation table coordinates must increase strictly")
        self.size, self.dim = self.x.size, self.v.shape[1]
        self.interpolation, self.averaging = interpolation, averaging
        self._xl = self.x.tolist()

    # index convention of interpolation_coordinate_find, 1-based like the reference
    def find(self, x):
        if x <= self.x[0]:
            return 0
        if x >= self.x[-1]:
            return self.size
        return bisect.bisect_right(self._xl, x)

    def _at_index(self, x, i):
        if i <= 0:
            return self.v[0].copy()
        if i >= self.size:
            return self.v[-1].copy()
        if self.interpolation == "step":
            return self.v[i - 1].copy()
        xi = (x - self.x[i - 1]) / (self.x[i] - self.x[i - 1])
        return (1.0 - xi) * self.v[i - 1] + xi * self.v[i]

    def interpolate(self, x):
        return self._at_index(x, self.find(x))

    def _integral(self, x1, x2, i):
        if self.interpolation == "step":
            return (x2 - x1) * self._at_index(x1, i)
        return 0.5 * (x2 - x1) * (self._at_index(x1, i) + self._at_index(x2, i))

    def average(self, interval):
        a, b = float(interval[0]), float(interval[1])
        if self.averaging == "endpoint":
            return 0.5 * (self.interpolate(a) + self.interpolate(b))
        dx = b - a
        if dx < 1.0e-15:
            return self.interpolate(0.5 * (a + b))
        total = np.zeros(self.dim)
        x1 = a
        for i in range(self.find(x1), self.size):
            x2, done = self.x[i], False
            if x2 > b:
                x2, done = b, True
            total += self._integral(x1, x2, i)
            x1 = x2
            if done:
                break
        if x1 < b:
            total += self._integral(x1, b, self.size)
        return total / dx
