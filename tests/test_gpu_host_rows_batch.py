"""A BATCH of boundaries evaluated on the host (FS_BC_HOST_ROW): every reach its own channel, its own node count and its own
plugin parameters; one gather kernel + one transfer brings the Newton vector at the reach ends to the host, the host evaluates
all plugins at once (numpy), one upload per side takes the rows back, one kernel launch makes the iteration
(include/flowsim_abi.h: fs_batch_get_boundary_iterate / fs_batch_set_host_rows / fs_batch_iterate).  The same rating curves exist
as a device kind (FS_BC_RATING_POWER, rating_curve.py:32-63): the two runs must agree to rounding, level by level, with the same
Newton counts - and reaches that converge early must sit still while the others go on."""
import numpy as np
import pytest

from oracle import preissmann_oracle as O

pytestmark = pytest.mark.gpu


def problems(B, seed=11):
    from test_gpu_instantiations import prismatic_problem
    rng = np.random.default_rng(seed)
    out = []
    for i in range(B):
        e = dict(index=i, cells_per_thread=2, waves_per_reach=1, full=0, long_reach=0)
        p = prismatic_problem(e, ("flow", "power"), trapezoid=bool(i % 2), n_steps=5)
        n = int(rng.integers(40, p.N + 1))                 # ragged: cut the reach (prismatic: any prefix is a reach of its own)
        cut = lambda a: np.asarray(a)[:n].copy()
        geo = {k: cut(v) for k, v in p.geo.items()}
        geo["z_bed"] = geo["z_bed"] - geo["z_bed"][-1]     # the downstream bed level stays 0
        p2 = O.Problem(geo=geo, h0=cut(p.h0), Q0=cut(p.Q0), us=O.BC("flow_hydrograph", bed_level=float(geo["z_bed"][0]), target=p.us.target),
                       ds=p.ds, theta=p.theta, dt=p.dt, dx=p.dx, nt=p.nt, tol=p.tol)
        out.append(p2)
    return out


def run_device(ps):
    from fixture_batch import hetero_batch_from_problems
    with hetero_batch_from_problems(ps, mode="table", history=False) as b:
        b.step(ps[0].nt - 1)
        assert np.all(b.status() == 0)
        return b.hydrographs(0, ps[0].nt), b.iterations(0, ps[0].nt), b.state()


def run_host_rows(ps):
    from fixture_batch import boundary_spec, hetero_batch_from_problems
    from flowsim_amd import BoundarySpec, _abi as A
    a = np.array([p.ds.rc["a"] for p in ps]); e = np.array([p.ds.rc["b"] for p in ps])
    with hetero_batch_from_problems(ps, mode="table", history=False) as b:
        b.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_HOST_ROW))       # batch-wide kind (flowsim_abi.h); upstream stays per reach
        launches = 0
        for k in range(1, ps[0].nt):
            while True:
                it = b.boundary_iterate()                                # [4, B]: h_0, Q_0, h_last, Q_last
                stage = it[2] + 0.0                                      # bed level 0, no stage shift
                res = it[3] - a * stage ** e                             # boundary.py:92-95: Q - discharge(stage)
                b.set_host_rows(A.DOWNSTREAM, -a * e * stage ** (e - 1.0), np.ones(len(ps)), res)
                launches += 1
                if b.iterate() == 0:
                    break
        assert np.all(b.status() == 0) and b.level == ps[0].nt - 1
        return b.hydrographs(0, ps[0].nt), b.iterations(0, ps[0].nt), b.state(), launches


def test_a_batch_of_host_evaluated_boundaries_follows_the_device_kind():
    ps = problems(96)
    assert len({p.N for p in ps}) > 20
    hyd_d, its_d, (h_d, Q_d) = run_device(ps)
    hyd_h, its_h, (h_h, Q_h), launches = run_host_rows(ps)
    assert np.array_equal(its_d, its_h)                                  # reach by reach, level by level
    assert launches == int(its_d[1:].max(axis=1).sum())                 # a level costs its slowest reach's iterations, no more
    assert len(set(its_d[1:].ravel().tolist())) > 1                      # (they do differ: early reaches waited)
    np.testing.assert_allclose(hyd_h, hyd_d, rtol=1e-11, atol=1e-12)
    for i, p in enumerate(ps):
        np.testing.assert_allclose(h_h[i, :p.N], h_d[i, :p.N], rtol=1e-11)
        np.testing.assert_allclose(Q_h[i, :p.N], Q_d[i, :p.N], rtol=1e-11, atol=1e-10)


def run_mixed(ps, host):
    """the reaches in `host` bring their downstream row from the host, the others keep the device's rating curve - one batch"""
    from fixture_batch import boundary_spec, hetero_batch_from_problems
    from flowsim_amd import BoundarySpec, _abi as A
    a = np.array([p.ds.rc["a"] for p in ps]); e = np.array([p.ds.rc["b"] for p in ps])
    with hetero_batch_from_problems(ps, mode="table", history=False) as b:
        b.set_boundary_per_reach(A.DOWNSTREAM, [BoundarySpec(A.BC_HOST_ROW) if r in host else boundary_spec(p.ds, p.nt) for r, p in enumerate(ps)])
        if host:
            with pytest.raises(RuntimeError, match="fs_batch_iterate"):      # one plugin reach: the batch advances iteration by iteration
                b.step(1)
        poison = np.where(np.isin(np.arange(len(ps)), sorted(host)), 0.0, np.nan)      # what is handed in for the device's reaches is not looked at
        for k in range(1, ps[0].nt):
            while True:
                it = b.boundary_iterate()
                stage = it[2] + 0.0
                res = it[3] - a * stage ** e
                if host:
                    b.set_host_rows(A.DOWNSTREAM, -a * e * stage ** (e - 1.0) + poison, np.ones(len(ps)) + poison, res + poison)
                if b.iterate() == 0:
                    break
        assert np.all(b.status() == 0) and b.level == ps[0].nt - 1
        return b.hydrographs(0, ps[0].nt), b.iterations(0, ps[0].nt), b.state()


def test_host_rows_on_some_reaches_of_a_batch_only():
    """FS_BC_HOST_ROW as a PER-REACH kind (fs_batch_set_bc_per_reach): the reference runs a RatingCurve subclass on one channel
    and closed-form boundaries on the next (boundary.py:56-141 belongs to the Boundary object); a batch now does the same.  Every
    third reach keeps the device kind; its results are those of the all-device batch advanced the same way bit for bit (its
    parameters are not touched by fs_batch_set_host_rows, whatever the caller's array holds there), the others follow to rounding
    with the same counts."""
    ps = problems(48, seed=23)
    host = {r for r in range(len(ps)) if r % 3}
    hyd_d, its_d, (h_d, Q_d) = run_mixed(ps, set())
    hyd_m, its_m, (h_m, Q_m) = run_mixed(ps, host)
    assert np.array_equal(its_d, its_m) and np.array_equal(its_d, run_device(ps)[1])
    for r, p in enumerate(ps):
        if r in host:
            np.testing.assert_allclose(hyd_m[:, :, r], hyd_d[:, :, r], rtol=1e-11, atol=1e-12)
            np.testing.assert_allclose(h_m[r, :p.N], h_d[r, :p.N], rtol=1e-11)
        else:
            assert np.array_equal(hyd_m[:, :, r], hyd_d[:, :, r]) and np.array_equal(h_m[r, :p.N], h_d[r, :p.N]) and np.array_equal(Q_m[r, :p.N], Q_d[r, :p.N])


def test_the_ends_come_back_as_they_are_on_the_device():
    """fs_batch_get_boundary_iterate against the full Newton vector (fs_batch_get_guess), ragged reaches, fp64"""
    from fixture_batch import hetero_batch_from_problems
    from flowsim_amd import BoundarySpec, _abi as A
    ps = problems(17, seed=5)
    with hetero_batch_from_problems(ps, mode="table", history=False) as b:
        b.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_HOST_ROW))
        it0 = b.boundary_iterate()
        b.set_host_rows(A.DOWNSTREAM, np.zeros(17), np.ones(17), np.zeros(17))
        b.iterate()
        it = b.boundary_iterate()
        hg, Qg = b.guess()
    assert np.array_equal(it0[0], [p.h0[0] for p in ps]) and np.array_equal(it0[3], [p.Q0[-1] for p in ps])
    for i, p in enumerate(ps):
        assert (it[0, i], it[1, i], it[2, i], it[3, i]) == (hg[i, 0], Qg[i, 0], hg[i, p.N - 1], Qg[i, p.N - 1])
