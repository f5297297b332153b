"""Laser-scan dataset writer: the text files the reference's data collectors produce,
`<dir>/laser2d_<track>_<ctr>.txt` = one "hit.x hit.y" line per ray (Agent::sensor_hits_, the "robot frame" hit points)
followed by "throttle steering" without a trailing newline (FieldNavigators/collect_data/collect_data_random.cpp:65-96,
MeasurementMode::Laser2d), written from the batched environment: one file per agent and recorded step, numbered in
agent order within a step.  Values are formatted like `std::ostream << float` (six significant digits, %g).

Off the hot path: this reads the device buffers back once per recorded step.
"""
import os

import numpy as np

from . import _capi as capi


def _fmt(x):
    return "%g" % float(x)


class Laser2dWriter:
    def __init__(self, directory, track_name):
        self.directory, self.track_name, self.ctr = directory, track_name, 0
        os.makedirs(directory, exist_ok=True)  # collect_data_random.cpp:46-49

    def path(self, ctr):
        return os.path.join(self.directory, "laser2d_%s_%d.txt" % (self.track_name, ctr))

    def write_sample(self, hits_xy, throttle, steering):
        """One DataCollectorAgent::saveMeasurement call; hits_xy is [R, 2]."""
        lines = ["%s %s\n" % (_fmt(x), _fmt(y)) for x, y in hits_xy]
        with open(self.path(self.ctr), "w") as f:
            f.write("".join(lines) + "%s %s" % (_fmt(throttle), _fmt(steering)))
        self.ctr += 1

    def save(self, env, agents=None, skip_crashed=True):
        """Records the current observation and action of `agents` (default: all) of a BatchedEnvironment; crashed
        agents are skipped: the reference only records inside `while (... && !agent->crashed_)` (collect_data_random.cpp:177-183).  Returns the number
        of files written."""
        hits = env.hits()
        thr, steer = env.get(capi.F_THROTTLE), env.get(capi.F_STEER)
        crashed = env.get(capi.F_CRASHED)
        idx = range(env.N) if agents is None else agents
        n = 0
        for a in idx:
            if skip_crashed and crashed[a]:
                continue
            self.write_sample(hits[a], thr[a], steer[a])
            n += 1
        return n


def read_sample(path):
    """Parses one laser2d file back into (hits [R, 2] float32, throttle, steering)."""
    rows = [ln.split() for ln in open(path).read().split("\n") if ln.strip()]
    hits = np.array(rows[:-1], dtype=np.float32).reshape(-1, 2)
    return hits, float(rows[-1][0]), float(rows[-1][1])
