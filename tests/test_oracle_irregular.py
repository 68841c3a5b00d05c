"""Pins oracle/irregular_oracle.py (polyline sections) to the reference:
the `probe` table holds the reference's own IrregularSection methods evaluated at several stages
per node, and the full runs hold its PreissmannSolver output (oracle/gen_golden.py case_irregular)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import irregular_oracle as IO
from oracle import preissmann_oracle as O

CASES = ("irr_single", "irr_levee", "irr_mixed")
# A, P, T exact to rounding; the finite-difference quantities (dh = 1e-6) carry ~1e-10 cancellation noise
TOL = dict(A=1e-14, P=1e-14, T=1e-14, dA_dh=5e-9, n_eq=1e-14, K=1e-14, dR_dA=5e-9, dK_dA=5e-9,
           Sf=1e-13, dSf_dA=5e-9, dSf_dQ=1e-13, Sc=1e-13, dSc_dA=5e-9, dSc_dQ=1e-13)


@pytest.mark.parametrize("name", CASES)
def test_section_functions_match_reference_probe(name):
    fx, _ = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    X, Z, cnt = fx["geo_irr_x"], fx["geo_irr_z"], fx["geo_irr_npts"]
    worst = {k: 0.0 for k in TOL}
    multi = 0
    for row in fx["probe"]:
        i, h, Q = int(row[0]), row[1], row[2]
        if cnt[i] == 0:
            continue
        x, z = X[i, :cnt[i]], Z[i, :cnt[i]]
        rough = (fx["geo_n_left"][i], fx["geo_n_main"][i], fx["geo_n_right"][i], *fx["geo_irr_limits"][i])
        hw = h + z.min()
        multi += len(IO.subchannels(x, z, hw)) > 1
        A, P, R, T = IO.properties(x, z, hw)
        Sf, dSfA, dSfQ, K, dK = IO.friction(x, z, rough, h, Q)
        Sc, dScA, dScQ = IO.curvature_terms(x, z, rough, float(fx["geo_curvature"][i]), h, Q)
        mine = dict(A=A, P=P, T=T, dA_dh=IO.dA_dh(x, z, hw), n_eq=IO.equivalent_n(x, z, rough, hw),
                    K=IO.conveyance(x, z, rough, hw), dR_dA=IO.dR_dA(x, z, hw), dK_dA=IO.dK_dA(x, z, rough, hw),
                    Sf=Sf, dSf_dA=dSfA, dSf_dQ=dSfQ, Sc=Sc, dSc_dA=dScA, dSc_dQ=dScQ)
        for (k, v), ref in zip(mine.items(), row[3:]):
            worst[k] = max(worst[k], abs(v - ref) / max(abs(ref), 1e-300) if ref != 0 else abs(v))
    for k, tol in TOL.items():
        assert worst[k] <= tol, (name, k, worst[k])
    if name == "irr_levee":
        assert multi > 0            # the sub-channel conveyance path (cross_section.py:372-447) was exercised


@pytest.mark.parametrize("name", CASES)
def test_newton_run_matches_reference(name):
    fx, meta = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    p = O.problem_from_fixture(fx, meta)
    r = O.newton_run(p)
    assert np.array_equal(r["iters"], fx["iters"])
    assert np.max(np.abs(r["depth"] - fx["depth"]) / np.abs(fx["depth"])) <= 1e-10
    assert np.max(np.abs(r["flow"] - fx["flow"]) / np.abs(fx["flow"])) <= 1e-10


def test_water_surface_vertex_edge_is_dropped():
    """cross_section.py:262/:292: a vertex exactly at the stage is neither wet nor above."""
    x = np.array([0.0, 2.0, 4.0, 6.0]); z = np.array([3.0, 1.0, 0.0, 3.0])
    A, P, R, T = IO.properties(x, z, 1.0)
    # edge (1,2) ends AT the surface and is dropped; only edge (2,3), cut at the surface, remains
    assert T == pytest.approx(2.0 / 3.0, rel=1e-15) and A == pytest.approx(1.0 / 3.0, rel=1e-15)
    A2, _, _, T2 = IO.properties(x, z, 1.0 + 1e-9)
    assert A2 == pytest.approx(1.0 / 3.0 + 1.0, rel=1e-8) and T2 == pytest.approx(2.0 / 3.0 + 2.0, rel=1e-8)
