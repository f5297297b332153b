"""The resident form of okenv_step_packed (the C++ facade's Environment::step in a tight loop): a kernel that stays on the
GPU and takes one step per hand-over through mapped host memory.  Every step is compared with the oracle, through the start
of the resident kernel, other C-ABI calls in between (which stop it), pauses longer than its idle time, and steps the host
hands over too late (the kernel has left: the step is redone by a launch of its own)."""
import ctypes as C
import time

import numpy as np
import pytest

from test_gpu_packed_step import ORACLE_KEYS, REC, WITH_STATS, packed_step

pytestmark = pytest.mark.gpu


def make(gpu, oracle, N, R, track="Austin", seed=0):
    t = gpu.Track(track)
    fan = np.array([-70, -30, 0, 30, 70], dtype=np.float32) if R == 5 else gpu.default_ray_fan(R)
    dev = gpu.BatchedEnvironment.from_track(t, N, ray_angles_deg=fan)
    orc = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, t.P, N)
    rec = np.zeros(N, dtype=REC)
    rec["pos_x"], rec["pos_y"], rec["rot"] = t.x[idx], t.y[idx], t.heading[idx]
    orc.reset_agents(np.arange(N), t.x[idx], t.y[idx], t.heading[idx])
    return t, dev, orc, rec, rng


class Run:
    """The device side runs its steps back to back (that is what makes the kernel stay); the oracle replays them afterwards."""

    def __init__(self, gpu, oracle, t, dev, orc, rec, rng):
        self.gpu, self.oracle, self.t, self.dev, self.orc, self.rec, self.rng = gpu, oracle, t, dev, orc, rec, rng
        self.log = []

    def steps(self, n, pause=0.0):
        t, rec, N = self.t, self.rec, self.dev.N
        thr = self.rng.uniform(30, 100, (n, N)).astype(np.float32)
        steer = self.rng.uniform(-5, 5, (n, N)).astype(np.float32)
        where = self.rng.integers(0, t.P, (n, N))
        for i in range(n):
            if pause:
                time.sleep(pause)
            crashed = np.flatnonzero(rec["crashed"])  # crashed agents are put back first, as the applications do
            if crashed.size:
                idx = where[i, crashed]
                rec["pos_x"][crashed], rec["pos_y"][crashed], rec["rot"][crashed] = t.x[idx], t.y[idx], t.heading[idx]
                rec["speed"][crashed] = rec["acc"][crashed] = 0
                rec["crashed"][crashed] = rec["timed_out"][crashed] = 0
            rec["throttle"], rec["steer"] = thr[i], steer[i]
            hits = packed_step(self.gpu, self.dev, rec, WITH_STATS)
            self.log.append((crashed, where[i, crashed], thr[i], steer[i], rec.copy(), hits))

    def check(self):
        """Replays every logged step on the oracle and compares all fields, bit for bit."""
        t, orc, oracle = self.t, self.orc, self.oracle
        for n, (crashed, idx, thr, steer, rec, hits) in enumerate(self.log):
            if crashed.size:
                orc.reset_agents(crashed, t.x[idx], t.y[idx], t.heading[idx])
            orc.set(oracle.F_THR, thr)
            orc.set(oracle.F_STEER, steer)
            orc.step(1)
            o = orc.snapshot()
            for k, ok_ in ORACLE_KEYS.items():
                a, b = np.ascontiguousarray(rec[k]), np.ascontiguousarray(o[ok_])
                assert a.tobytes() == b.astype(a.dtype).tobytes(), (n, k)
            assert np.array_equal(hits[..., 0].view(np.uint32), o["rel_x"].view(np.uint32)), n
            assert np.array_equal(hits[..., 1].view(np.uint32), o["rel_y"].view(np.uint32)), n
        done = len(self.log)
        self.log = []
        return done


# What these tests assert about residency is only what the library decides by itself whatever the host's scheduler does: with
# OKENV_RESIDENT=1 every eligible step is handed to a resident kernel (packed_resident_steps counts the hand-overs, served or
# not); a hand-over that nobody answers in time (packed_fallbacks) is legitimate -- the kernel has left, the step is redone by a
# launch of its own -- and can be caused by a stall of the host at any moment, so its count is never pinned from above or to
# zero.  The hard assertion everywhere is the bit-for-bit comparison of every step with the oracle.
@pytest.mark.parametrize("N,R", [(1, 5), (15, 5), (50, 15), (64, 64)])
def test_resident_steps_match_oracle(gpu, oracle, monkeypatch, N, R):
    monkeypatch.setenv("OKENV_RESIDENT", "1")
    run = Run(gpu, oracle, *make(gpu, oracle, N, R, seed=N))
    assert run.dev.info()["agents_per_block"] == 1
    run.steps(300)
    info = run.dev.info()
    assert info["packed_resident_steps"] == 300 and 0 <= info["packed_fallbacks"] <= 300
    assert run.dev.step_count == 300
    assert run.check() == 300
    run.dev.close()


def test_resident_yields_to_other_calls_and_comes_back(gpu, oracle, monkeypatch):
    monkeypatch.setenv("OKENV_RESIDENT", "1")
    run = Run(gpu, oracle, *make(gpu, oracle, 15, 5, track="Silverstone", seed=3))
    dev = run.dev
    run.steps(240)
    assert dev.info()["packed_resident_steps"] == 240
    run.check()
    # any other call stops the kernel first and sees the state of the last step
    o = run.orc.snapshot()
    assert np.array_equal(dev.get(gpu.capi.F_POS_X).view(np.uint32), o["pos_x"].view(np.uint32))
    assert np.array_equal(np.asarray(dev.get(gpu.capi.F_HIT_X)).view(np.uint32).ravel(), o["hit_x"].view(np.uint32).ravel())
    assert dev.info()["packed_resident"] == 0
    run.steps(100)         # ... and it comes back
    assert dev.info()["packed_resident_steps"] == 340
    time.sleep(0.005)      # far longer than the kernel's patience (300 us): it has left; the next step starts a new one
    run.steps(1)
    run.steps(5, pause=0.001)
    assert dev.info()["packed_resident_steps"] == 346
    assert run.check() == 106
    dev.close()


def test_resident_heuristic_start(gpu, oracle, monkeypatch):
    """Without OKENV_RESIDENT the kernel becomes resident after a run of quick steps -- how many of a tight Python loop's steps
    count as quick is the host's business, so the counters are only required to be consistent; steps that are slow for certain
    (a sleep between them) never start or keep a resident kernel."""
    monkeypatch.delenv("OKENV_RESIDENT", raising=False)
    run = Run(gpu, oracle, *make(gpu, oracle, 15, 5, track="Silverstone", seed=3))
    dev = run.dev
    run.steps(200)
    served = dev.info()["packed_resident_steps"]
    assert 0 <= dev.info()["packed_fallbacks"] <= served <= 200
    print("heuristic start: %d of 200 tight-loop steps were handed to a resident kernel" % served)
    run.check()
    run.steps(5, pause=0.001)   # 1 ms between steps >> the 100 us that count as quick
    info = dev.info()
    assert info["packed_resident"] == 0 and info["packed_resident_steps"] == served
    assert run.check() == 5
    dev.close()


def test_resident_never_when_switched_off(gpu, oracle, monkeypatch):
    monkeypatch.setenv("OKENV_RESIDENT", "0")
    run = Run(gpu, oracle, *make(gpu, oracle, 15, 5, seed=5))
    run.steps(60)
    assert run.dev.info()["packed_resident"] == 0 and run.dev.info()["packed_resident_steps"] == 0
    run.check()
    run.dev.close()


def test_step_handed_over_too_late_is_redone(gpu, oracle, monkeypatch):
    monkeypatch.setenv("OKENV_RESIDENT", "1")
    monkeypatch.setenv("OKENV_RESIDENT_STALL_US", "700")   # every seventh hand-over comes after the kernel's 300 us of patience
    run = Run(gpu, oracle, *make(gpu, oracle, 15, 5, seed=9))
    run.steps(80)
    info = run.dev.info()
    assert info["packed_fallbacks"] >= 5 and info["packed_resident_steps"] > info["packed_fallbacks"]
    assert run.dev.step_count == 80
    assert run.check() == 80
    run.dev.close()


def test_mixed_calls_do_not_start_and_stop_a_kernel_each_time(gpu, oracle, monkeypatch):
    """step() and checkCollision() in turn: the collision-only pass is not served by the resident kernel, so none is started."""
    from test_gpu_packed_step import COLLIDE_ONLY
    monkeypatch.delenv("OKENV_RESIDENT", raising=False)
    run = Run(gpu, oracle, *make(gpu, oracle, 15, 5, seed=11))
    for _ in range(40):
        run.steps(1)
        packed_step(gpu, run.dev, run.rec.copy(), COLLIDE_ONLY)
    info = run.dev.info()
    assert info["packed_resident"] == 0 and info["packed_resident_steps"] == 0
    assert run.check() == 40
    run.dev.close()


def test_short_residencies_back_off(gpu, oracle, monkeypatch):
    """A caller that interrupts every run of quick steps soon after the kernel has become resident gets it less and less often
    (the number of quick steps needed quadruples after a residency that served fewer than 32)."""
    monkeypatch.delenv("OKENV_RESIDENT", raising=False)
    run = Run(gpu, oracle, *make(gpu, oracle, 15, 5, seed=13))
    for _ in range(12):
        run.steps(20)
        run.dev.get(gpu.capi.F_POS_X)   # stops a resident kernel
    served = run.dev.info()["packed_resident_steps"]
    assert served <= 30                 # without the back-off: four steps of every run of 20, i.e. 48 (a stalled host: fewer)
    assert run.check() == 240
    run.dev.close()
