"""flowsim_amd.pipeline.step_pipelined: a batch handed over as host buffers and stepped as stream-ordered blocks of reaches
(upload of block i + 1 / stepping of block i / download of block i - 1 overlap) returns the bits of the single batch, for
blocks of unequal size too; a failing block surfaces as its exception instead of a hang."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _c3_batch(lo, nb, N, K):
    from flowsim_amd import BoundarySpec, PreissmannBatch, _abi as A
    from flowsim_amd.synthetic import c3_reach_parameters, inflow_table
    b_, n_, S0, Qb = c3_reach_parameters(lo, nb)
    L = (N - 1) * 250.0
    x = PreissmannBatch(nb, N, K + 1, section_mode="rect_uniform")
    x.set_scheme(0.6, 600.0, 250.0, 1e-6, 100)
    x.set_geometry_uniform(b_, n_, S0 * L, np.zeros(nb))
    x.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, K + 1, 600.0)))
    x.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S0, bed_level=np.zeros(nb))))
    return x


def _start_state(B, N):
    from flowsim_amd.synthetic import c3_reach_parameters, normal_depth_rect
    b_, n_, S0, Qb = c3_reach_parameters(0, B)
    hn = normal_depth_rect(b_, n_, S0, Qb)
    return np.repeat(hn[:, None], N, axis=1), np.repeat(Qb[:, None], N, axis=1)


@pytest.mark.parametrize("N, blocks", [(300, [96]), (300, [24, 24, 24, 24]), (1000, [50, 7, 1, 38]), (4096, [16, 16])])
def test_blocks_of_reaches_return_the_bits_of_one_batch(N, blocks):
    from flowsim_amd.pipeline import step_pipelined
    B, K = sum(blocks), 6
    h, Q = _start_state(B, N)
    with _c3_batch(0, B, N, K) as one:
        one.set_state(h, Q); one.step(K)
        h_ref, Q_ref = one.state()
        assert np.all(one.status() == 0)
    parts, lo = [], 0
    for nb in blocks:
        parts.append(_c3_batch(lo, nb, N, K)); lo += nb
    try:
        h2, Q2 = np.zeros_like(h), np.zeros_like(Q)
        step_pipelined(parts, h, Q, K, out=(h2, Q2))
        assert np.array_equal(h2, h_ref) and np.array_equal(Q2, Q_ref)
        assert all(np.all(p.status() == 0) and p.level == K for p in parts)
    finally:
        for p in parts:
            p.close()
    assert not np.array_equal(h_ref, h)            # the flood wave did move the state


def test_a_failing_block_raises_and_the_others_finish():
    from flowsim_amd.pipeline import step_pipelined
    N, K = 300, 3
    h, Q = _start_state(24, N)
    parts = [_c3_batch(0, 8, N, K), _c3_batch(8, 8, N, K), _c3_batch(16, 8, N, K)]
    try:
        h2, Q2 = np.zeros_like(h), np.zeros_like(Q)
        with pytest.raises(RuntimeError):
            step_pipelined(parts, h, Q, K + 5, out=(h2, Q2))       # more levels than the boundary table holds: every block refuses
        with pytest.raises(ValueError):
            step_pipelined(parts, h[:20], Q[:20], K, out=(h2, Q2))
    finally:
        for p in parts:
            p.close()
