"""Reference-path generators in the layout the engine consumes.

Host-side, NumPy, vectorised re-implementations of what the reference's drivers use to build ``ref_path``:

* ``calc_spline_course``            path_generator/cubic_spline_planner.py:311-323 (natural cubic spline through
  way-points, parametrised by cumulative chord length, sampled every ``ds``; yaw and curvature from the derivatives)
* ``calc_bezier_path`` / ``calc_4points_bezier_path``   path_generator/bezierPath.py:8-46
* ``generate_point_trajectory`` / ``generate_lemniscate_trajectory``   controllers/mppi_differential_drive.py:374-389
* ``racecar_lemniscate`` / ``racecar_circle``   controllers/mppi_race_car_obstacle.py:288-299, mppi_race_car.py:224-234

``ref_path_array`` packs columns into the ``[N, 3]`` (x, y, yaw) or ``[N, 4]`` (x, y, yaw, v) array that
``MPPIAlgorithms`` / ``MPPIRacecarController`` (and ``mppi_set_ref_path``) take.
"""
from __future__ import annotations

import numpy as np


class NaturalCubicSpline:
    """y(x) through the knots with zero second derivative at both ends (the boundary rows 1,0,..,0 / 0,..,0,1 of
    cubic_spline_planner.py:146-161); coefficients a + b dx + c dx^2 + d dx^3 per segment."""

    def __init__(self, x, y):
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        h = np.diff(x)
        if np.any(h < 0):
            raise ValueError("x coordinates must be sorted in ascending order")
        n = x.size
        A = np.zeros((n, n))
        B = np.zeros(n)
        A[0, 0] = A[n - 1, n - 1] = 1.0
        for i in range(1, n - 1):
            A[i, i - 1], A[i, i], A[i, i + 1] = h[i - 1], 2.0 * (h[i - 1] + h[i]), h[i]
            B[i] = 3.0 * (y[i + 1] - y[i]) / h[i] - 3.0 * (y[i] - y[i - 1]) / h[i - 1]
        c = np.linalg.solve(A, B)
        self.x, self.a, self.c = x, y, c
        self.d = (c[1:] - c[:-1]) / (3.0 * h)
        self.b = (y[1:] - y[:-1]) / h - h * (2.0 * c[:-1] + c[1:]) / 3.0

    def _seg(self, t):
        t = np.asarray(t, dtype=np.float64)
        i = np.clip(np.searchsorted(self.x, t, side="right") - 1, 0, self.x.size - 2)
        return i, t - self.x[i]

    def position(self, t):
        i, dx = self._seg(t)
        return self.a[i] + self.b[i] * dx + self.c[i] * dx ** 2 + self.d[i] * dx ** 3

    def first_derivative(self, t):
        i, dx = self._seg(t)
        return self.b[i] + 2.0 * self.c[i] * dx + 3.0 * self.d[i] * dx ** 2

    def second_derivative(self, t):
        i, dx = self._seg(t)
        return 2.0 * self.c[i] + 6.0 * self.d[i] * dx


def calc_spline_course(x, y, ds=0.1):
    """(rx, ry, ryaw, rk, s) like cubic_spline_planner.py:311-323, as arrays."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    knots = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(x), np.diff(y)))])
    sx, sy = NaturalCubicSpline(knots, x), NaturalCubicSpline(knots, y)
    s = np.arange(0, knots[-1], ds)
    dx, dy, ddx, ddy = sx.first_derivative(s), sy.first_derivative(s), sx.second_derivative(s), sy.second_derivative(s)
    return sx.position(s), sy.position(s), np.arctan2(dy, dx), (ddy * dx - ddx * dy) / (dx ** 2 + dy ** 2) ** 1.5, s


def calc_bezier_path(control_points, n_points=50):
    """bezierPath.py:33-46: Bernstein-basis evaluation of the control polygon at linspace(0, 1, n_points)."""
    from math import comb
    cp = np.asarray(control_points, dtype=np.float64)
    n = cp.shape[0] - 1
    t = np.linspace(0, 1, n_points)[:, None]
    basis = np.stack([comb(n, i) * t[:, 0] ** i * (1 - t[:, 0]) ** (n - i) for i in range(n + 1)], axis=1)
    return basis @ cp


def calc_4points_bezier_path(sx, sy, syaw, ex, ey, eyaw, offset):
    """bezierPath.py:8-30: (path[500,2], control_points[4,2])."""
    dist = np.hypot(sx - ex, sy - ey) / offset
    cp = np.array([[sx, sy], [sx + dist * np.cos(syaw), sy + dist * np.sin(syaw)],
                   [ex - dist * np.cos(eyaw), ey - dist * np.sin(eyaw)], [ex, ey]])
    return calc_bezier_path(cp, n_points=500), cp


def generate_point_trajectory(start_point, end_point, num_points=100):
    """controllers/mppi_differential_drive.py:385-389 -> (x, y, yaw)."""
    x = np.linspace(start_point[0], end_point[0], num_points)
    y = np.linspace(start_point[1], end_point[1], num_points)
    yaw = np.arctan2(end_point[1] - start_point[1], end_point[0] - start_point[0]) * np.ones(num_points)
    return x, y, yaw


def generate_lemniscate_trajectory(a, num_points=100):
    """controllers/mppi_differential_drive.py:374-383 -> (x, y, yaw)."""
    t = np.linspace(-np.pi, np.pi, num_points)
    x = a * np.cos(t) / (1 + np.sin(t) ** 2)
    y = a * np.sin(t) * np.cos(t) / (1 + np.sin(t) ** 2)
    return x, y, np.arctan2(np.gradient(y), np.gradient(x))


def racecar_lemniscate(num_points, radius):
    """controllers/mppi_race_car_obstacle.py:288-299 -> float32 [N,4] (x, y, yaw, v=5)."""
    t = np.linspace(0, 2 * np.pi, num_points, dtype=np.float32)
    x = radius * np.cos(t) / (1 + np.sin(t) ** 2)
    y = radius * np.sin(t) * np.cos(t) / (1 + np.sin(t) ** 2)
    yaw = np.arctan2(np.gradient(y), np.gradient(x))
    return np.stack([x, y, yaw, np.ones_like(t) * 5.0], axis=1)


def racecar_circle(num_points, radius):
    """controllers/mppi_race_car.py:224-234 -> float32 [N,4]."""
    ang = np.linspace(0, 2 * np.pi, num_points, dtype=np.float32)
    return np.stack([radius * np.cos(ang), radius * np.sin(ang), ang + np.pi / 2, np.ones_like(ang) * 5.0], axis=1)


def ref_path_array(x, y, yaw, v=None):
    """Columns -> the [N,3] / [N,4] array the controllers take (`np.array([cx, cy, cyaw]).T`, :419)."""
    cols = [np.asarray(x, np.float64), np.asarray(y, np.float64), np.asarray(yaw, np.float64)]
    if v is not None:
        cols.append(np.broadcast_to(np.asarray(v, np.float64), cols[0].shape))
    return np.stack(cols, axis=1)
