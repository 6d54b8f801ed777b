"""API mirror of the reference misc/pyutils.py pieces the steps use: Logger tee (:6-17), Timer
(:50-83), to_one_hot (:86-101)."""
import sys
import time

import numpy as np


class Logger(object):
    """Replaces sys.stdout with a tee into `outfile` (misc/pyutils.py:6-17)."""

    def __init__(self, outfile):
        self.terminal = sys.stdout
        self.log = open(outfile, "w")
        sys.stdout = self

    def write(self, message):
        self.terminal.write(message)
        self.log.write(message)

    def flush(self):
        self.terminal.flush()


class Timer:
    """Wall-clock timer that prints a start stamp when given a message (misc/pyutils.py:50-83)."""

    def __init__(self, starting_msg=None):
        self.start = time.time()
        self.stage_start = self.start
        if starting_msg is not None:
            print(starting_msg, time.ctime(time.time()))

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        return

    def update_progress(self, progress):
        self.elapsed = time.time() - self.start
        self.est_total = self.elapsed / progress
        self.est_remaining = self.est_total - self.elapsed
        self.est_finish = int(self.start + self.est_total)

    def str_estimated_complete(self):
        return str(time.ctime(self.est_finish))

    def get_stage_elapsed(self):
        return time.time() - self.stage_start

    def reset_stage(self):
        self.stage_start = time.time()

    def lapse(self):
        out = time.time() - self.stage_start
        self.stage_start = time.time()
        return out


def to_one_hot(sparse_integers, maximum_val=None, dtype=np.bool_):
    """[...] ints -> [K, ...] one-hot (misc/pyutils.py:86-101)."""
    sparse_integers = np.asarray(sparse_integers)
    if maximum_val is None:
        maximum_val = int(np.max(sparse_integers)) + 1
    flat = sparse_integers.reshape(-1)
    one_hot = np.zeros((maximum_val, flat.shape[0]), dtype)
    one_hot[flat, np.arange(flat.shape[0])] = 1
    return one_hot.reshape([maximum_val] + list(sparse_integers.shape))
