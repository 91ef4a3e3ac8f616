"""Collecting the results of worker processes (the multi-rank tests): a worker that has died is reported at once, with
its exit code, instead of after the queue's timeout -- and a progress line goes to stderr every minute (a GPU box
takes a long silence for a hang)."""
import queue
import sys
import time


def collect(procs, q, n_results, timeout):
    """n_results items from q (workers put one each); AssertionError when a worker exits without having delivered."""
    out, t0, last_note = [], time.time(), time.time()
    while len(out) < n_results:
        try:
            out.append(q.get(timeout=2.0))
            continue
        except queue.Empty:
            pass
        dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
        assert not dead, f"worker processes died before delivering: (index, exit code) = {dead}"
        finished = sum(p.exitcode == 0 for p in procs)
        assert not (finished == len(procs) and q.empty()), "all workers have exited, results are missing"
        assert time.time() - t0 < timeout, f"no result after {timeout} s ({len(out)} of {n_results} delivered)"
        if time.time() - last_note > 60:
            print(f"[mp_results] waiting for {n_results - len(out)} worker result(s), {time.time() - t0:.0f} s",
                  file=sys.stderr, flush=True)
            last_note = time.time()
    return out
