"""Host-side public surface against values the reference produced.

tests/golden/api_values.json was written by oracle/gen_api_values.py from the reference itself (imported in the build
container): recipes - constructor arguments, set-up calls, method calls with keyword arguments - and what each call
returned there, or the exception it raised.  The same recipes run here on the mirror package
(src/hydromodel -> flowsim_amd.hydromodel): hydraulics.*, utility.*, TrapezoidalSection (rectangle, trapezoid, compound),
IrregularSection (single channel, a section a levee splits in two), RatingCurve (set / fit, both types), Hydrograph,
LumpedStorage (every loss term and its derivatives, area curve, mass balance).

Tolerance: 1e-12 relative for closed forms; the finite-difference members of the polyline section (dR/dA, dA/dh with
dh = 1e-6 and what is built on them) amplify the last-bit differences of the area walk by 1/dh and get 1e-6."""
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.gen_api_values import run_recipe

RECIPES = json.load(open(os.path.join(GOLDEN, "api_values.json")))

FD_METHODS = {"dR_dA", "dA_dh", "dK_dA", "dSf_dA", "dSc_dA"}      # IrregularSection: central differences with dh = 1e-6


def label(r):
    if "function" in r:
        return f"{r['module']}.{r['function']}({', '.join(f'{k}={v}' for k, v in list(r['kwargs'].items())[:3])})"[:80]
    return f"{r['class']}({', '.join(f'{k}={v}' for k, v in list(r['init'].items())[:3])})"[:80]


def unwrap(v):
    if isinstance(v, dict) and "__float__" in v:
        return float(v["__float__"])
    return v


def same(got, want, tol, where):
    got, want = unwrap(got), unwrap(want)
    if isinstance(want, dict):
        assert isinstance(got, dict) and set(got) == set(want), where
        for k in want:
            same(got[k], want[k], tol, f"{where}[{k}]")
    elif isinstance(want, list):
        assert isinstance(got, list) and len(got) == len(want), f"{where}: {got} vs {want}"
        for i, (g, w) in enumerate(zip(got, want)):
            same(g, w, tol, f"{where}[{i}]")
    elif isinstance(want, float):
        assert isinstance(got, (int, float)), f"{where}: {got!r} vs {want!r}"
        if math.isnan(want) or math.isinf(want):
            assert (math.isnan(got) and math.isnan(want)) or got == want, f"{where}: {got} vs {want}"
        else:
            assert abs(got - want) <= tol * max(abs(want), 1e-30) + 1e-300, f"{where}: {got!r} vs {want!r}"
    else:
        assert got == want, f"{where}: {got!r} vs {want!r}"


def check(got, want, tol, where):
    if "raises" in want:
        assert got.get("raises") == want["raises"], f"{where}: reference raises {want['raises']}, got {got}"
    else:
        assert "value" in got, f"{where}: raised {got.get('raises')} where the reference returns {want['value']!r}"
        same(got["value"], want["value"], tol, where)


@pytest.mark.parametrize("rec", RECIPES, ids=[label(r) for r in RECIPES])
def test_mirror_reproduces_the_reference(rec):
    got = run_recipe(rec, "src.hydromodel")
    if "function" in rec:
        check(got, rec["expect"], 1e-12, label(rec))
        return
    for c, g, w in zip(rec["calls"], got, rec["expect"]):
        fd = rec["class"] == "IrregularSection" and c["method"] in FD_METHODS
        check(g, w, 1e-6 if fd else 1e-12, f"{rec['class']}.{c['method']}({c['kwargs']})")


def test_fixture_covers_every_public_function_of_hydraulics_and_utility():
    surface = json.load(open(os.path.join(GOLDEN, "api_surface.json")))
    called = {(r["module"], r["function"]) for r in RECIPES if "function" in r}
    for mod in ("hydraulics", "utility"):
        for fn in surface[mod]["functions"]:
            if fn == "create_directory_if_not_exists":
                continue
            assert (mod, fn) in called, f"{mod}.{fn} has no known-answer record"
