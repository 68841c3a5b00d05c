"""Host logic of flowsim_amd.pipeline.step_pipelined without a GPU: stand-in blocks record when each of their three stages
runs.  Within a stage the blocks must go in order, the stages of different blocks must be able to overlap, the rows of the
arrays must reach the right block, and a block that raises must neither hang the others nor be swallowed."""
import threading
import time

import numpy as np
import pytest

from flowsim_amd.pipeline import step_pipelined


class FakeBlock:
    log = []
    lock = threading.Lock()

    def __init__(self, B, N, name, fail_in=None, delay=0.01):
        self.B, self.N, self.name, self.fail_in, self.delay = B, N, name, fail_in, delay
        self.h = self.Q = None
        self.steps = 0

    def _mark(self, what):
        with FakeBlock.lock:
            FakeBlock.log.append((what, self.name, time.perf_counter()))

    def set_state(self, h, Q):
        self._mark("up+")
        if self.fail_in == "up":
            raise RuntimeError(f"block {self.name}: upload refused")
        time.sleep(self.delay)
        self.h, self.Q = h.copy(), Q.copy()
        self._mark("up-")

    def sync(self):
        pass

    def step(self, n, sync=True):
        self._mark("step+")
        if self.fail_in == "step":
            raise RuntimeError(f"block {self.name}: step refused")
        time.sleep(self.delay)
        self.h = self.h + n
        self.Q = self.Q * 2
        self.steps += n
        self._mark("step-")

    def state(self, out):
        self._mark("down+")
        time.sleep(self.delay)
        out[0][...] = self.h
        out[1][...] = self.Q
        self._mark("down-")


def test_rows_reach_their_block_and_stages_go_in_order():
    FakeBlock.log = []
    sizes = [5, 1, 7, 3]
    parts = [FakeBlock(b, 11, i) for i, b in enumerate(sizes)]
    B = sum(sizes)
    rng = np.random.default_rng(1)
    h, Q = rng.random((B, 11)), rng.random((B, 11))
    h2, Q2 = np.zeros_like(h), np.zeros_like(Q)
    step_pipelined(parts, h, Q, 4, out=(h2, Q2))
    assert np.array_equal(h2, h + 4) and np.array_equal(Q2, Q * 2)
    assert all(p.steps == 4 for p in parts)
    log = FakeBlock.log
    for stage in ("up", "step", "down"):
        spans = {name: [t for w, n, t in log if n == name and w.startswith(stage)] for name in range(len(parts))}
        for i in range(1, len(parts)):
            assert spans[i][0] >= spans[i - 1][1], f"{stage} of block {i} started before block {i - 1} was through"
    # the pipeline is one: block 1 uploads before block 0 has finished stepping
    t_up1 = [t for w, n, t in log if n == 1 and w == "up+"][0]
    t_step0_end = [t for w, n, t in log if n == 0 and w == "step-"][0]
    assert t_up1 < t_step0_end


@pytest.mark.parametrize("stage", ["up", "step"])
def test_a_block_that_raises_is_reported_and_nobody_hangs(stage):
    FakeBlock.log = []
    parts = [FakeBlock(2, 3, 0), FakeBlock(2, 3, 1, fail_in=stage), FakeBlock(2, 3, 2)]
    h, Q = np.ones((6, 3)), np.ones((6, 3))
    h2, Q2 = np.zeros_like(h), np.zeros_like(Q)
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="block 1"):
        step_pipelined(parts, h, Q, 1, out=(h2, Q2))
    assert time.perf_counter() - t0 < 5.0
    assert np.array_equal(h2[:2], h[:2] + 1) and np.array_equal(h2[4:], h[4:] + 1)     # the healthy blocks delivered


def test_mismatched_arrays_are_refused():
    parts = [FakeBlock(2, 3, 0), FakeBlock(2, 3, 1)]
    with pytest.raises(ValueError, match="4 reaches"):
        step_pipelined(parts, np.ones((5, 3)), np.ones((5, 3)), 1, out=(np.ones((5, 3)), np.ones((5, 3))))
