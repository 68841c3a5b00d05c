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
