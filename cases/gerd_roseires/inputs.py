"""CSV readers for the case (stand-ins for the pandas helpers of the reference's
cases/gerd_roseires/custom_functions.py:100-157; plotting / GIS export are out of scope)."""
import csv

import numpy as np

from src.hydromodel.cross_section import TrapezoidalSection


def _rows(path):
    with open(path, newline="", encoding="utf-8-sig") as f:
        return list(csv.reader(f))


def import_hydrograph(path, hr_to_s_conversion=True):
    """[time, flow] table; the file has a header line and a units line; hours -> seconds."""
    tab = np.array([[float(a), float(b)] for a, b in _rows(path)[2:]], dtype=np.float64)
    tab = tab[np.argsort(tab[:, 0], kind="stable")]
    if hr_to_s_conversion:
        tab[:, 0] *= 3600
    return tab


def import_table(path, sort_by_first=True):
    rows = [r for r in _rows(path)[1:] if r and all(c != "" for c in r)]
    tab = np.array([[float(c) for c in r] for r in rows], dtype=np.float64)
    return tab[np.argsort(tab[:, 0], kind="stable")] if sort_by_first else tab


def load_trapezoid_sections(path, n_main=None, n_fp=None):
    """Compound trapezoids fitted to the surveyed sections; section 53 is left out as upstream
    (custom_functions.py:137-139); n_main / n_fp override the tabulated roughness (the ensemble knob)."""
    rows = _rows(path)
    col = {name: i for i, name in enumerate(rows[0])}
    chainages, sections = [], []
    for r in rows[1:]:
        if r[col["file"]] == "53.csv":
            continue
        g = lambda k: float(r[col[k]])
        chainages.append(g("chainage"))
        sections.append(TrapezoidalSection(
            z_bed=g("z_min"), b_main=g("b_main"), m_main=g("m_main"),
            n_main=g("n_main") if n_main is None else n_main, z_bank=g("z_min") + g("h_bankfull"),
            b_fp_left=g("b_fp_left"), b_fp_right=g("b_fp_right"), m_fp=g("m_fp"),
            n_left=g("n_left") if n_fp is None else n_fp, n_right=g("n_right") if n_fp is None else n_fp))
    return chainages, sections


def grid_table(path):
    """2-D release table: first column = stage, header = second variable -> (X[n,2], y[n]) without blanks."""
    rows = _rows(path)
    second = [float(c) for c in rows[0][1:]]
    X, y = [], []
    for r in rows[1:]:
        for v, c in zip(second, r[1:]):
            if c != "" and c.lower() != "nan":
                X.append([float(r[0]), v]); y.append(float(c))
    return np.array(X), np.array(y)
