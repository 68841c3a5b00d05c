"""Host buffers in, host buffers out: a batch stepped as stream-ordered chunks of reaches.

`fs_batch_set_state` / `fs_batch_step` / `fs_batch_get_state` on ONE handle run back to back - the bus idles while the
kernel runs and the CUs idle while the state crosses the bus (flagship workload, 32 levels: 79 ms up, 190 ms stepping,
100 ms down).  Reaches are independent (`channel.py:37-39`: one Channel, one reach), so the batch can be cut into
blocks of reaches, one handle (= one HIP stream) per block, and the three stages overlap: block i + 1 uploads while block i
steps while block i - 1 downloads (the link is full duplex).  Within a stage the blocks go in order - two uploads never
share the link, two step kernels never share the CUs - which is what the events below enforce; ctypes releases the GIL
around every ABI call, so one host thread per block is enough.  The results are those of the single batch bit for bit
(a reach never sees its neighbours).  Measured: `tools/bench_pcie.py`, `profiles/round3/pcie.json`.
"""
import threading

__all__ = ["step_pipelined"]


def step_pipelined(parts, h, Q, n_steps, out):
    """parts: PreissmannBatch handles over consecutive blocks of reaches (scheme, geometry and boundaries set, in the order
    of the rows of h / Q); h, Q: [B, N] start state of all reaches; out = (h_out, Q_out): [B, N] arrays that receive the
    state after `n_steps` levels.  Returns nothing; raises what the first failing block raised."""
    h_out, Q_out = out
    offs = [0]
    for p in parts:
        offs.append(offs[-1] + p.B)
    if not (h.shape[0] == Q.shape[0] == h_out.shape[0] == Q_out.shape[0] == offs[-1]):
        raise ValueError(f"step_pipelined: the blocks hold {offs[-1]} reaches, the arrays {h.shape[0]} / {h_out.shape[0]}")
    n = len(parts)
    done = [[threading.Event() for _ in range(n)] for _ in range(3)]      # upload / step / download of block i is through
    errors = []

    def work(i):
        lo, hi = offs[i], offs[i + 1]
        stage = 0
        try:
            if i: done[0][i - 1].wait()
            parts[i].set_state(h[lo:hi], Q[lo:hi]); parts[i].sync(); done[0][i].set()
            stage = 1
            if i: done[1][i - 1].wait()
            parts[i].step(n_steps, sync=True); done[1][i].set()
            stage = 2
            if i: done[2][i - 1].wait()
            parts[i].state(out=(h_out[lo:hi], Q_out[lo:hi])); done[2][i].set()
        except BaseException as e:          # the blocks behind must not wait for ever
            errors.append((i, e))
            for s in range(stage, 3):
                done[s][i].set()

    threads = [threading.Thread(target=work, args=(i,), name=f"flowsim-block-{i}") for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise min(errors, key=lambda ie: ie[0])[1]
