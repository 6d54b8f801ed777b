"""Small host utilities behind the names run_sample.py and the steps use from the reference's
misc/pyutils.py: a stdout tee (`Logger`, reference :6-17), a step stopwatch (`Timer`, :50-83) and
`to_one_hot` (:86-101).  Only the behaviour the label-generation steps rely on is provided."""
import contextlib
import sys
import time

import numpy as np


class Logger:
    """`Logger(path)` makes every later `print` go to the terminal and to `path` (the reference installs
    itself as sys.stdout the same way).  Works as a context manager too; `close()` restores sys.stdout."""

    def __init__(self, outfile):
        self._streams = (sys.stdout, open(outfile, "w"))
        sys.stdout = self

    def write(self, text):
        for s in self._streams:
            s.write(text)
        return len(text)

    def flush(self):
        for s in self._streams:
            with contextlib.suppress(ValueError):      # log file already closed
                s.flush()

    def isatty(self):
        return False

    def close(self):
        if sys.stdout is self:
            sys.stdout = self._streams[0]
        self._streams[1].close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Timer:
    """Stopwatch on the monotonic clock.  `Timer("step.make_cam:")` prints the message with the wall-clock
    date like the reference; `lapse()` returns the seconds since the previous lap (or the start) and starts
    a new lap; `elapsed()` the seconds since construction."""

    def __init__(self, starting_msg=None):
        self._t0 = self._lap = time.perf_counter()
        if starting_msg is not None:
            print(starting_msg, time.ctime())

    def elapsed(self):
        return time.perf_counter() - self._t0

    def lapse(self):
        now = time.perf_counter()
        dt, self._lap = now - self._lap, now
        return dt

    # progress estimates of the reference's training loops (misc/pyutils.py:61-76; step/train_cam.py, step/train_irn.py
    # call them): same names and attributes, on the monotonic clock; wall-clock dates only where one is printed
    def update_progress(self, progress):
        """progress in (0, 1]: sets `est_total`, `est_remaining` (seconds) and `est_finish` (epoch seconds)."""
        done = self.elapsed()
        self.est_total = done / progress
        self.est_remaining = self.est_total - done
        self.est_finish = int(time.time() + self.est_remaining)

    def str_estimated_complete(self):
        return time.ctime(self.est_finish)

    def get_stage_elapsed(self):
        return time.perf_counter() - self._lap

    def reset_stage(self):
        self._lap = time.perf_counter()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def to_one_hot(sparse_integers, maximum_val=None, dtype=np.bool_):
    """[...] ints -> [K, ...] one-hot planes, K = maximum_val or max+1 (misc/pyutils.py:86-101)."""
    ids = np.asarray(sparse_integers)
    k = int(ids.max()) + 1 if maximum_val is None else int(maximum_val)
    return (ids[None] == np.arange(k).reshape((k,) + (1,) * ids.ndim)).astype(dtype)
