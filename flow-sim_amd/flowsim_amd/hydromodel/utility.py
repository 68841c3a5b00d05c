"""Small helpers of the reference's src/hydromodel/utility.py:4-35 that its public surface uses."""
import os

import numpy as np


def create_directory_if_not_exists(directory):
    if not os.path.exists(directory):
        os.makedirs(directory)


def manhattan_norm(vector):
    return np.sum(np.abs(np.asarray(vector, dtype=np.float64)))


def euclidean_norm(vector):
    return np.sum(np.square(np.asarray(vector, dtype=np.float64))) ** 0.5


def seconds_to_hms(seconds: int):
    if seconds < 0:
        return "0:00:00"
    s = int(seconds)
    return f"{s // 3600}:{(s % 3600) // 60:02d}:{s % 60:02d}"


def compute_curv(x_coords, y_coords):
    """Signed curvature of a plane polyline, (x' y'' - y' x'') / (x'^2 + y'^2)^1.5 with arc-length derivatives by
    np.gradient (utility.py:36-50)."""
    x, y = np.asarray(x_coords, dtype=np.float64), np.asarray(y_coords, dtype=np.float64)
    s = np.concatenate(([0.0], np.cumsum(np.hypot(np.diff(x), np.diff(y)))))
    x1, y1 = np.gradient(x, s), np.gradient(y, s)
    x2, y2 = np.gradient(x1, s), np.gradient(y1, s)
    return (x1 * y2 - y1 * x2) / (x1 ** 2 + y1 ** 2) ** 1.5
