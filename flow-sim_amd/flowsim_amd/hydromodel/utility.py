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
