"""Worker process of the fork-free host builder pool (mps.simulate_many): reads pickled (index, circuit, fidelity, zero)
tasks from stdin, writes pickled (index, tensors, fidelity, seconds) to stdout, until EOF or a ``None`` task.

A separate interpreter started with ``subprocess`` -- not ``fork`` (unsafe once the parent has initialised the GPU) and
not multiprocessing's spawn / forkserver (which re-import the parent's ``__main__``: the reference's driver scripts run
their work at module level).  LAPACK runs single-threaded here; the parallelism is one process per core, which, unlike
threads in one process, does not contend for OpenBLAS's internal buffer lock."""
import os
import pickle
import sys
import time


def main() -> None:
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    inp = os.fdopen(os.dup(0), "rb")
    out = os.fdopen(os.dup(1), "wb")
    os.dup2(2, 1)  # anything printed by accident goes to stderr, not into the protocol
    from .mps import simulate

    while True:
        try:
            task = pickle.load(inp)
        except EOFError:
            break
        if task is None:
            break
        idx, circuit, fidelity, zero = task[:4]
        max_bond = task[4] if len(task) > 4 else None
        t0 = time.perf_counter()
        try:
            m = simulate(circuit, fidelity, zero, max_bond)
            reply = (idx, m.tensors, m.fidelity, time.perf_counter() - t0, None)
        except Exception as exc:  # reported to the parent, which raises
            reply = (idx, None, 0.0, 0.0, repr(exc))
        pickle.dump(reply, out, protocol=4)
        out.flush()


if __name__ == "__main__":
    main()
