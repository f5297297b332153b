"""A second, independent reading of the reference lines no reference output pins (its raycast exists only as CUDA,
`Environment.cpp` needs raylib): written here in numpy float32, straight from the reference text, sharing with the oracle
nothing but the trigonometry definition (`ok_sincosf`, exported by the oracle as `oracle_sincosf`) -- and compared with
the oracle (`oracle/okenv_oracle.c`) bit for bit.

  raySegmentIntersect / castRaysToSegmentsKernel   Environment/CollisionChecker.cu:8-35, 43-70
  ray build, hit transform, crash test             Environment/CollisionChecker.cu:115-128, 144-172
  checkAndUpdateStandstill, Environment::step      Environment/Environment.cpp:16-39, 125-142
  Agent::moveViaVelocity / moveViaAcceleration     Environment/Agent.cpp:82-98, 108-119

numpy's float32 scalars and arrays round every operation to IEEE single precision and never contract a*b+c, which is
the arithmetic the oracle is compiled for (-ffp-contract=off).  This does not turn "parity unpinned" into "pinned" (both
readings are ours); it removes the risk that the oracle's C restatement and the kernel share one misreading.
"""
import numpy as np
import pytest

import _oracle as O

f32 = np.float32
K_DEG2RAD = f32(np.pi / 180.0)  # Typedefs.h:10  kDeg2Rad = float(M_PI / 180.0F)
K_DT = f32(0.016)               # Agent.cpp:84,110
K_SENSOR_RANGE = f32(200.0)     # Agent.h:10
K_SPEED_LIMIT = f32(100.0)      # Agent.h:11
K_PERIOD = 200                  # Environment.h:19  DisplacementStats::kPeriod
K_DISP_THRESHOLD = f32(20.0)    # Environment.h:20  kDisplamentThreshold


def sincos(angle_rad):
    """The parity definition of sine / cosine (include/okenv_math.h ok_sincosf), evaluated by the oracle's export."""
    a = np.ascontiguousarray(angle_rad, dtype=f32).reshape(-1)
    s, c = np.zeros(a.size, dtype=f32), np.zeros(a.size, dtype=f32)
    O.lib().oracle_sincosf(a, s, c, a.size)
    return s.reshape(np.shape(angle_rad)), c.reshape(np.shape(angle_rad))


def cast_rays(ray_x, ray_y, ray_angle, active, hit_x, hit_y, segments):
    """castRaysToSegmentsKernel (CollisionChecker.cu:37-71), every ray in lock step: the loop over the segments is the
    kernel's own (sequential, shrinking min_t), vectorised over rays only.  Inactive rays keep their hit point."""
    ray_dy, ray_dx = sincos(ray_angle)                     # :47-48  cosf / sinf of the ray angle
    min_t = np.full(ray_x.shape, K_SENSOR_RANGE, dtype=f32)  # :49
    with np.errstate(all="ignore"):
        for seg_x1, seg_y1, seg_x2, seg_y2 in segments:     # :51
            # raySegmentIntersect, :19-34
            seg_dx = seg_x2 - seg_x1
            seg_dy = seg_y2 - seg_y1
            denom = ray_dx * seg_dy - ray_dy * seg_dx
            parallel = np.abs(denom) < f32(1e-8)
            t = ((seg_x1 - ray_x) * seg_dy - (seg_y1 - ray_y) * seg_dx) / denom
            s = ((seg_x1 - ray_x) * ray_dy - (seg_y1 - ray_y) * ray_dx) / denom
            hit = ~parallel & (t >= f32(0.0)) & (t <= min_t) & (s >= f32(0.0)) & (s <= f32(1.0))
            min_t = np.where(hit, t, min_t)                 # :65
    new_x = ray_x + min_t * ray_dx                          # :69
    new_y = ray_y + min_t * ray_dy                          # :70
    return np.where(active, new_x, hit_x), np.where(active, new_y, hit_y), min_t


class NumpyEnvironment:
    """N agents as arrays; one step() = Environment::step() without the render call."""

    def __init__(self, segments, n, ray_angles_deg, sensor_offset=0.0):
        self.seg = np.ascontiguousarray(segments, dtype=f32).reshape(-1, 4)
        self.n, self.rays = n, np.asarray(ray_angles_deg, dtype=f32)
        self.sensor_offset = f32(sensor_offset)
        z = lambda dt=f32: np.zeros(n, dtype=dt)  # noqa: E731
        self.pos_x, self.pos_y, self.rot, self.speed, self.acc = z(), z(), z(), z(), z()
        self.thr, self.steer = z(), z()
        self.mode = z(np.uint8)
        self.crashed, self.timed_out = z(bool), z(bool)
        self.disp_ctr = z(np.int64)
        self.disp_x, self.disp_y = z(), z()
        self.disp_to = z(bool)
        r = self.rays.size
        self.hit_x, self.hit_y = np.zeros((n, r), dtype=f32), np.zeros((n, r), dtype=f32)  # "define zeros" 
        self.rel_x, self.rel_y = np.zeros((n, r), dtype=f32), np.zeros((n, r), dtype=f32)

    def move(self, m):
        """Agent::move for the agents selected by mask m (Agent.cpp:21-47, 82-98, 108-119)."""
        vel = m & (self.mode == 0)
        accm = m & (self.mode == 1)
        both = vel | accm
        self.rot = np.where(both, self.rot + self.steer, self.rot)
        self.speed = np.where(vel, self.thr, self.speed)
        self.acc = np.where(accm, self.acc + self.thr, self.acc)
        sp = self.speed + self.acc * K_DT
        sp = np.where(sp < 0, f32(0), sp)
        sp = np.where(sp > K_SPEED_LIMIT, K_SPEED_LIMIT, sp)
        self.speed = np.where(accm, sp, self.speed)
        s, c = sincos(K_DEG2RAD * self.rot)
        self.pos_x = np.where(both, self.pos_x + c * self.speed * K_DT, self.pos_x)
        self.pos_y = np.where(both, self.pos_y + s * self.speed * K_DT, self.pos_y)

    def standstill(self, m):
        """checkAndUpdateStandstill (Environment.cpp:16-39) for the agents selected by m."""
        first = m & (self.disp_ctr == 0)
        period = m & ~first & (self.disp_ctr >= K_PERIOD)
        other = m & ~first & ~period
        self.disp_x = np.where(first, self.pos_x, self.disp_x)
        self.disp_y = np.where(first, self.pos_y, self.disp_y)
        dx, dy = self.pos_x - self.disp_x, self.pos_y - self.disp_y
        dist_moved = dx * dx + dy * dy
        timed = period & (dist_moved < K_DISP_THRESHOLD * K_DISP_THRESHOLD)
        self.disp_to = np.where(first | other, False, np.where(timed, True, self.disp_to))
        self.disp_ctr = np.where(first | other, self.disp_ctr + 1, np.where(period, 0, self.disp_ctr))

    def check_collision(self):
        """CollisionChecker::runCollisionKernel (CollisionChecker.cu:113-174)."""
        s, c = sincos(K_DEG2RAD * self.rot)
        ox = self.pos_x + self.sensor_offset * c                               # :121-124
        oy = self.pos_y + self.sensor_offset * s
        angle = K_DEG2RAD * (self.rot[:, None] + self.rays[None, :])          # :125
        active = np.broadcast_to(~self.crashed[:, None], angle.shape)         # :126
        rx = np.broadcast_to(ox[:, None], angle.shape).astype(f32)
        ry = np.broadcast_to(oy[:, None], angle.shape).astype(f32)
        self.hit_x, self.hit_y, _ = cast_rays(rx, ry, angle, active, self.hit_x, self.hit_y, self.seg)
        s2, c2 = sincos(self.rot * K_DEG2RAD)                                  # :148 agent_rot_rad
        xt, yt = self.hit_x - rx, self.hit_y - ry                             # :155-156
        self.rel_x = xt * c2[:, None] - yt * s2[:, None]                      # :157
        self.rel_y = xt * s2[:, None] + yt * c2[:, None]                      # :158
        n2 = self.rel_x * self.rel_x + self.rel_y * self.rel_y
        min_dist2 = np.full(self.n, K_SENSOR_RANGE * K_SENSOR_RANGE, dtype=f32)  # :150
        for i in range(self.rays.size):                                      # :161-164, NaN never wins
            min_dist2 = np.where(n2[:, i] < min_dist2, n2[:, i], min_dist2)
        self.crashed = self.crashed | (min_dist2 < f32(2.0))                  # :167-171

    def step(self):
        """Environment::step (Environment.cpp:125-145)."""
        alive = ~self.crashed
        self.move(alive)
        self.standstill(alive)
        timed = alive & self.disp_to
        self.crashed = self.crashed | timed
        self.timed_out = self.timed_out | timed
        self.check_collision()


def bits(a):
    return np.ascontiguousarray(a, dtype=f32).view(np.uint32)


@pytest.mark.parametrize("name", ["Silverstone", "Spa"])
def test_cast_rays_reading_equals_oracle(oracle, name):
    """castRaysToSegmentsKernel, literal loop, against the oracle's sweep: realistic rays, rays starting exactly on
    boundary points, rays along segments, rays from far outside."""
    t = O.Track(name)
    rng = np.random.default_rng(11)
    n = 1500
    idx = rng.integers(0, t.P, n)
    ox = (t.x[idx] + rng.normal(0, 10, n)).astype(f32)
    oy = (t.y[idx] + rng.normal(0, 10, n)).astype(f32)
    ang = rng.uniform(-np.pi, np.pi, n).astype(f32)
    k = n // 5
    sel = rng.integers(0, t.S, k)
    ox[:k], oy[:k] = t.segments[sel, 0], t.segments[sel, 1]
    ang[:k // 2] = np.arctan2(t.segments[sel[:k // 2], 3] - t.segments[sel[:k // 2], 1],
                              t.segments[sel[:k // 2], 2] - t.segments[sel[:k // 2], 0]).astype(f32)
    ox[k:2 * k] = rng.uniform(-400, 2000, k).astype(f32)
    oy[k:2 * k] = rng.uniform(-400, 1800, k).astype(f32)
    _, _, min_t = cast_rays(ox, oy, ang, np.ones(n, dtype=bool), ox, oy, t.segments)
    seg = np.ascontiguousarray(t.segments.reshape(-1))
    want = np.array([O.lib().oracle_cast_ray(float(ox[i]), float(oy[i]), float(ang[i]), seg, t.S) for i in range(n)], dtype=f32)
    assert np.array_equal(bits(min_t), bits(want))
    assert np.count_nonzero(want < 200.0) > n // 2


@pytest.mark.parametrize("mode", [0, 1])
def test_environment_step_reading_equals_oracle(oracle, mode):
    """260 Environment::step()s of 24 agents x 16 rays on Austin with per-step host actions: some agents drive into walls
    (collision crash, stale rays afterwards), some stand still (201-tick timeout), one uses a sensor offset-free fan like
    the rest; every field of both implementations compared bit for bit after every step."""
    t = O.Track("Austin")
    n, r = 24, 16
    fan = O.default_ray_fan(r)
    rng = np.random.default_rng(5 + mode)
    start = rng.integers(0, t.P, n)
    env = NumpyEnvironment(t.segments, n, fan)
    orc = O.OracleEnv(t.segments, n, r, fan, (t.x, t.y, t.heading))
    x0, y0, h0 = t.x[start].copy(), t.y[start].copy(), t.heading[start].copy()
    orc.reset_agents(np.arange(n, dtype=np.int32), x0, y0, h0)
    env.pos_x, env.pos_y, env.rot = x0.copy(), y0.copy(), h0.copy()
    env.mode[:] = mode
    orc.set(O.F_MODE, np.full(n, mode, dtype=np.uint8))
    saw_crash = saw_timeout = False
    for step in range(260):
        if mode == 0:
            thr = rng.uniform(0, 100, n).astype(f32)
        else:
            thr = rng.uniform(-0.3, 0.6, n).astype(f32)
        steer = rng.uniform(-5, 5, n).astype(f32)
        thr[:4], steer[:4] = 0, 0          # four agents never move: standstill timeout at the 201st tick
        if mode == 0:
            thr[4:8], steer[4:8] = 60, 3   # four agents circle into a wall
        env.thr, env.steer = thr.copy(), steer.copy()
        orc.set(O.F_THR, thr)
        orc.set(O.F_STEER, steer)
        env.step()
        orc.step(1)
        o = orc.snapshot()
        for k, mine in (("pos_x", env.pos_x), ("pos_y", env.pos_y), ("rot", env.rot), ("speed", env.speed), ("acc", env.acc),
                        ("hit_x", env.hit_x), ("hit_y", env.hit_y), ("rel_x", env.rel_x), ("rel_y", env.rel_y)):
            assert np.array_equal(bits(mine).reshape(-1), bits(o[k]).reshape(-1)), (step, k)
        assert np.array_equal(env.crashed, o["crashed"].astype(bool)), step
        assert np.array_equal(env.timed_out, o["timed_out"].astype(bool)), step
        assert np.array_equal(env.disp_ctr, o["disp_ctr"].astype(np.int64)), step
        assert np.array_equal(env.disp_to, o["disp_to"].astype(bool)), step
        saw_crash |= bool((env.crashed & ~env.timed_out).any())
        saw_timeout |= bool(env.timed_out.any())
    assert saw_timeout and (saw_crash or mode == 1)
