"""Contact-schedule tables for the MPC (host-side mirror of the reference's
OffsetDurationGait, src/MPC_Ctrl/Gait.cpp:142-166 and :187-193)."""
import numpy as np


def mpc_table(n_segments, offsets, durations, iteration):
    """mpcTable[i*4 + leg] = 1 if leg is in stance at horizon step i.

    Gait.cpp:147-158: iter = (i + _iteration + 1) % n; progress = iter -
    offset (wrapped into [0, n)); stance iff progress < duration."""
    i = np.arange(n_segments)[:, None]
    it = (i + iteration + 1) % n_segments
    progress = it - np.asarray(offsets)[None, :]
    progress = np.where(progress < 0, progress + n_segments, progress)
    return (progress < np.asarray(durations)[None, :]).astype(np.uint8).reshape(-1)


class OffsetDurationGait:
    """Same public surface as the reference class for the MPC path."""

    def __init__(self, n_segments, offsets, durations, name="walk"):
        self.n = int(n_segments)
        self.offsets = tuple(int(x) for x in offsets)
        self.durations = tuple(int(x) for x in durations)
        self.name = name
        self.iteration = 0
        self.phase = 0.0

    def setIterations(self, iterations_per_mpc, current_iteration):
        # Gait.cpp:187-193
        self.iteration = (current_iteration // iterations_per_mpc) % self.n
        period = iterations_per_mpc * self.n
        self.phase = float(current_iteration % period) / float(period)

    def getMpcTable(self):
        return mpc_table(self.n, self.offsets, self.durations, self.iteration)
