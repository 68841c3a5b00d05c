"""The device's unpivoted scalar-tridiagonal chunk-fold + tree elimination (modelled lane by lane in
tests/partition_model.py) against scipy.sparse.linalg.spsolve, the routine the reference calls
(preissmann.py:146), on Jacobians assembled by the oracle from the golden cases."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import GOLDEN
from oracle import preissmann_oracle as O
import partition_model as PM


@pytest.mark.parametrize("name", ["akbari", "example", "gerd", "bc_compound_normal", "synthetic_rect_512"])
@pytest.mark.parametrize("m", [2, 4, 8, 16])
def test_partition_solve_matches_superlu(name, m):
    fx, meta = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    p = O.problem_from_fixture(fx, meta, 0 if meta.get("B") else None)
    N = p.N
    rows, cols = O.csr_pattern(N)
    store = {"Y_prev": None} if p.ds.storage is not None else None
    h, Q = p.h0.copy(), p.Q0.copy()
    for it in range(3):          # three Newton iterations of step 1: large, medium and tiny updates
        R, data, _ = O.assemble(p, h, Q, p.h0, p.Q0, 1, store)
        J = sp.coo_matrix((data, (rows, cols)), shape=(2 * N, 2 * N)).tocsr()
        d = spla.spsolve(J, -R)
        d2, worst = PM.solve(data, R, m)
        for sl in (slice(0, None, 2), slice(1, None, 2)):
            assert np.max(np.abs(d2[sl] - d[sl])) <= 1e-10 * max(np.max(np.abs(d[sl])), 1e-300)
        assert worst > 1e-3          # no pivot lost more than three digits to cancellation
        h, Q = h + d[0::2], Q + d[1::2]


@pytest.mark.parametrize("name", ["akbari", "gerd", "synthetic_rect_512", "c3_4096"])
@pytest.mark.parametrize("W", [2, 4])
def test_continuant_cross_wave_step_matches_superlu(name, W):
    """The cross-wave step of the no-diagnostics multi-wave kernels (FS_XWAVE_CONT, fs_kernel.hpp): W wave segments + the upstream
    row solved as one small tridiagonal system by continuants with one reciprocal, instead of the pairwise fold - same bar as the
    fold above, on the same Jacobians and on the benchmark-size ones (c3_4096: 16 rows per lane, 4 waves: the flagship's layout)."""
    fx, meta = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    p = O.problem_from_fixture(fx, meta, 0 if meta.get("B") else None)
    N = p.N
    m = 16 if N > 2048 else (8 if N > 256 else 2)
    T = 64 * W
    while T * m < N:
        m *= 2
    rows, cols = O.csr_pattern(N)
    h, Q = p.h0.copy(), p.Q0.copy()
    for it in range(3):
        R, data, _ = O.assemble(p, h, Q, p.h0, p.Q0, 1, None)
        J = sp.coo_matrix((data, (rows, cols)), shape=(2 * N, 2 * N)).tocsr()
        d = spla.spsolve(J, -R)
        d1, _ = PM.solve(data, R, m, T=T)
        d2, _ = PM.solve(data, R, m, T=T, xwave=W)
        for sl in (slice(0, None, 2), slice(1, None, 2)):
            scale = max(np.max(np.abs(d[sl])), 1e-300)
            assert np.max(np.abs(d2[sl] - d[sl])) <= 1e-10 * scale
            assert np.max(np.abs(d2[sl] - d1[sl])) <= 1e-11 * scale         # and next to the fold it replaces
        h, Q = h + d[0::2], Q + d[1::2]
