// okenv_kernels.h -- device code of the batched Environment step for gfx950 (MI355X).
//
// One launch advances every agent by `n_steps` Environment::step()s (reference
// Environment/Environment.cpp:125-149 minus render):
//
//   per agent, per step:  [optional on-device action source + reset]            (bench driver loop)
//                         Agent::move()                 Environment/Agent.cpp:21-47,82-98,108-119
//                         checkAndUpdateStandstill()    Environment/Environment.cpp:16-39,134-140
//   per ray,   per step:  ray build                     Environment/CollisionChecker.cu:115-128
//                         first-hit raycast             Environment/CollisionChecker.cu:37-71  (grid walk, ok_raycast.h)
//                         hit transform + crash test    Environment/CollisionChecker.cu:144-172
//
// Mapping to the machine (MI355X: 256 CUs x 4 SIMD32, 64-lane waves, 160 KB LDS per CU):
//   * lanes are rays.  An agent owns G = pow2ceil(R) <= 64 consecutive lanes of one wave (R = 64: one wave per
//     agent; R = 16: four agents per wave); fans wider than 64 rays loop inside the lane.  The per-agent
//     min over rays of the squared hit distance -- the crash test -- is a xor-shuffle reduction inside
//     those G lanes, no LDS, no atomics.
//   * the track is staged ONCE per workgroup into LDS as the compact "poly" image of ok_grid.h: a cell-major
//     stream of boundary points (8 B each, shared by chained segments) + a 4-byte header per grid cell,
//     90-115 KB for the config tracks; every point evaluation and ray-segment test then reads LDS,
//     never HBM.  Adjacent rays of a fan start in the same cell and fan out slowly, so most LDS reads of a
//     wave-instruction hit the same few addresses (broadcast).
//   * agent state is struct-of-arrays in HBM, read once at launch into registers, carried across the
//     launch's steps, written back once; observations (sensor_hits_, their norms) and world hit points are
//     written every step, agent-major/ray-minor, i.e. 256 contiguous bytes per wave-instruction at R = 64.
//   * agents never interact (rays test track segments only), so there is no inter-workgroup communication
//     and no grid barrier: a launch of n_steps is embarrassingly parallel over agents.
//
// Numerics: compiled with -ffp-contract=off; divisions and square roots are the IEEE correctly-rounded
// forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); sine/cosine come from ok_sincosf
// (include/okenv_math.h), shared bit-for-bit with the CPU oracle.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ok_raycast.h"

struct OkDeviceState
{
    float    *pos_x, *pos_y, *rot, *speed, *acc, *thr, *steer;
    uint8_t  *mode, *crashed, *timed_out, *disp_to;
    uint32_t *disp_ctr;
    float    *disp_x, *disp_y;
    float    *hit_x, *hit_y, *rel_x, *rel_y, *dist; // [N*R]
};

enum OkActionSource : int
{
    kActionsStored      = 0, // use thr/steer arrays as they are (set by the host between launches)
    kActionsPhiloxReset = 1, // bench recipe: per step reset crashed agents, draw U[0,100) x U[-5,5)
};

struct OkStepParams
{
    OkDeviceState st;
    int           N, R;
    int           G;        // lanes per agent (power of two, <= 64)
    int           rays_per_lane;
    const float  *ray_deg;  // [R]
    float         sensor_offset;
    // compact image in global memory: [slots | hdr | brk], byte offsets from `image` (ok_grid.h)
    const uint8_t *image;
    uint32_t       image_bytes, off_hdr, off_brk;
    float          side_tol;
    OkGridGeom     geom;
    // wide (global-memory) form
    const OkSeg    *g_segs;
    const uint32_t *g_refs32;
    const uint32_t *g_start;
    int             S;
    // stepping
    int      n_steps;
    int      do_move; // 0: CollisionChecker::checkCollision only
    int      action_source;
    uint32_t seed, agent_base, step_base;
    const float *cx, *cy, *chead;
    int          P;
};

enum OkGridMode : int
{
    kGridLds    = 0, // compact image staged into LDS
    kGridGlobal = 1, // wide CSR form read from global memory
    kGridBrute  = 2, // no grid: sweep all segments (the reference kernel's algorithm; testing/ablation)
};

// min over the G lanes that belong to one agent; every lane of the group receives the result.
// (a < b ? a : b) keeps the sequential loop's "NaN never wins" behaviour (CollisionChecker.cu:161-164).
__device__ __forceinline__ float okGroupMin(float v, const int G)
{
    for (int off = 1; off < G; off <<= 1)
    {
        const float o = __shfl_xor(v, off, 64);
        v             = (o < v) ? o : v;
    }
    return v;
}

template <int kMode>
__device__ __forceinline__ float okCastRay(const OkStepParams &p,
                                            const OkPolyView   &lds_view,
                                            const float         ox,
                                            const float         oy,
                                            const float         rdx,
                                            const float         rdy)
{
    if (kMode == kGridLds)
    {
#if defined(OKENV_ABLATE) && OKENV_ABLATE == 1
        return OK_SENSOR_RANGE * (0.5F + 0.25F * rdx); // timing experiment only: no raycast at all
#else
        return ok_cast_ray_poly<false>(lds_view, ox, oy, rdx, rdy, nullptr, nullptr, nullptr);
#endif
    }
    else if (kMode == kGridGlobal)
    {
        OkGridView32 v;
        v.g     = p.geom;
        v.segs  = p.g_segs;
        v.refs  = p.g_refs32;
        v.start = p.g_start;
        return ok_cast_ray_grid<false>(v, ox, oy, rdx, rdy, nullptr, nullptr);
    }
    else
    {
        float min_t = OK_SENSOR_RANGE;
        for (int j = 0; j < p.S; ++j)
        {
            const OkSeg sg = p.g_segs[j];
            float       t;
            if (ok_ray_segment(ox, oy, rdx, rdy, sg.x1, sg.y1, sg.x2, sg.y2, min_t, t))
                min_t = t;
        }
        return min_t;
    }
}

// Stage the compact grid image into LDS with 16-byte loads (image_bytes is a multiple of 16).
__device__ __forceinline__ void okStageImage(const OkStepParams &p, unsigned char *lds)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(p.image);
    uint4       *dst = reinterpret_cast<uint4 *>(lds);
    const int    n16 = static_cast<int>(p.image_bytes >> 4);
    for (int i = threadIdx.x; i < n16; i += blockDim.x)
        dst[i] = src[i];
    __syncthreads();
}

template <int kMode>
__global__ void __launch_bounds__(1024) okStepKernel(const OkStepParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ok_lds[];
    OkPolyView view{};
    if (kMode == kGridLds)
    {
        okStageImage(p, ok_lds);
        view.g        = p.geom;
        view.slots    = reinterpret_cast<const OkPoint *>(ok_lds);
        view.hdr      = reinterpret_cast<const uint32_t *>(ok_lds + p.off_hdr);
        view.brk      = reinterpret_cast<const uint32_t *>(ok_lds + p.off_brk);
        view.side_tol = p.side_tol;
    }

    const int  G     = p.G;
    const long gl    = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int  agent = static_cast<int>(gl / G);
    const int  rlane = static_cast<int>(gl % G);
    // Lanes past the last agent stay in the loop (shuffles need the whole group) but never touch memory.
    const bool agent_ok = agent < p.N;
    const int  a        = agent_ok ? agent : 0;

    // ---- agent state: one broadcast load per field, carried in registers across the launch ---------
    float    pos_x = p.st.pos_x[a], pos_y = p.st.pos_y[a], rot = p.st.rot[a];
    float    speed = p.st.speed[a], acc = p.st.acc[a];
    float    thr = p.st.thr[a], steer = p.st.steer[a];
    const int mode = p.st.mode[a];
    bool     crashed = p.st.crashed[a] != 0, timed_out = p.st.timed_out[a] != 0, disp_to = p.st.disp_to[a] != 0;
    uint32_t disp_ctr = p.st.disp_ctr[a];
    float    disp_x = p.st.disp_x[a], disp_y = p.st.disp_y[a];

    for (int s = 0; s < p.n_steps; ++s)
    {
        // ---- bench driver: reset crashed agents, draw the step's action --------
        if (p.action_source == kActionsPhiloxReset)
        {
            const ok_random_action ra =
                ok_draw_random_action(p.seed, p.agent_base + static_cast<uint32_t>(a), p.step_base + static_cast<uint32_t>(s));
            if (crashed)
            {
                // Agent::reset (Agent.cpp:123-135); DisplacementStats deliberately untouched
                const uint32_t idx = ok_index_from_word(ra.reset_word, static_cast<uint32_t>(p.P));
                pos_x              = p.cx[idx];
                pos_y              = p.cy[idx];
                rot                = p.chead[idx];
                acc                = 0.F;
                speed              = 0.F;
                crashed            = false;
                timed_out          = false;
            }
            thr   = ra.throttle;
            steer = ra.steer;
        }

        // ---- 1) kinematics + standstill (Environment.cpp:128-142) --------------------------------------
        if (p.do_move && !crashed)
        {
            bool moved = true;
            if (mode == 0)
            { // moveViaVelocity
                rot += steer;
                speed = thr;
            }
            else if (mode == 1)
            { // moveViaAcceleration
                rot += steer;
                acc += thr;
                speed += (acc * OK_DT);
                speed = (speed < 0.F) ? 0.F : speed;
                speed = (speed > OK_SPEED_LIMIT) ? OK_SPEED_LIMIT : speed;
            }
            else
            {
                moved = false; // MANUAL: empty in the reference
            }
            if (moved)
            {
                float sn, cs;
                ok_sincosf(OK_DEG2RAD * rot, &sn, &cs);
                const float dx = cs * speed * OK_DT;
                pos_x += dx;
                const float dy = sn * speed * OK_DT;
                pos_y += dy;
            }
            // checkAndUpdateStandstill
            if (disp_ctr == 0U)
            {
                disp_x   = pos_x;
                disp_y   = pos_y;
                disp_to  = false;
                disp_ctr = 1U;
            }
            else if (disp_ctr >= OK_DISP_PERIOD)
            {
                const float ddx = pos_x - disp_x, ddy = pos_y - disp_y;
                const float d2  = ddx * ddx + ddy * ddy;
                if (d2 < OK_DISP_THRESH2)
                    disp_to = true;
                disp_ctr = 0U;
            }
            else
            {
                disp_to = false;
                ++disp_ctr;
            }
            if (disp_to)
            {
                crashed   = true;
                timed_out = true;
            }
        }

        // ---- 2) collision pass (CollisionChecker.cu:113-174) -----------------------------------------
        float sr, cr;
        ok_sincosf(OK_DEG2RAD * rot, &sr, &cr);
        const float ox     = pos_x + p.sensor_offset * cr;
        const float oy     = pos_y + p.sensor_offset * sr;
        const bool  active = !crashed;
        float       min_d2 = OK_SENSOR_RANGE * OK_SENSOR_RANGE;
        for (int q = 0; q < p.rays_per_lane; ++q)
        {
            const int  r      = rlane + q * G;
            const bool ray_ok = agent_ok && (r < p.R);
            const long k      = static_cast<long>(a) * p.R + (ray_ok ? r : 0);
            float      hx, hy;
            if (active)
            {
                float rdy = 0.F, rdx = 1.F, min_t = OK_SENSOR_RANGE;
                if (ray_ok)
                {
                    const float angle = OK_DEG2RAD * (rot + p.ray_deg[r]);
                    ok_sincosf(angle, &rdy, &rdx);
                    min_t = okCastRay<kMode>(p, view, ox, oy, rdx, rdy);
                }
                hx = ox + min_t * rdx;
                hy = oy + min_t * rdy;
                if (ray_ok)
                {
                    p.st.hit_x[k] = hx;
                    p.st.hit_y[k] = hy;
                }
            }
            else
            {
                // stale world hit point of a crashed agent
                hx = ray_ok ? p.st.hit_x[k] : ox;
                hy = ray_ok ? p.st.hit_y[k] : oy;
            }
            const float xt = hx - ox;
            const float yt = hy - oy;
            const float rx = xt * cr - yt * sr;
            const float ry = xt * sr + yt * cr;
            const float n2 = rx * rx + ry * ry;
            if (ray_ok)
            {
                p.st.rel_x[k] = rx;
                p.st.rel_y[k] = ry;
                p.st.dist[k]  = __builtin_sqrtf(n2);
                if (n2 < min_d2)
                    min_d2 = n2;
            }
        }
        min_d2 = okGroupMin(min_d2, G);
        if (min_d2 < OK_CRASH_DIST2)
            crashed = true;
    }

    // ---- write the agent state back once (lane 0 of each group) ----------------------------------------
    if (agent_ok && rlane == 0)
    {
        p.st.pos_x[a]     = pos_x;
        p.st.pos_y[a]     = pos_y;
        p.st.rot[a]       = rot;
        p.st.speed[a]     = speed;
        p.st.acc[a]       = acc;
        p.st.thr[a]       = thr;
        p.st.steer[a]     = steer;
        p.st.crashed[a]   = crashed ? 1 : 0;
        p.st.timed_out[a] = timed_out ? 1 : 0;
        p.st.disp_to[a]   = disp_to ? 1 : 0;
        p.st.disp_ctr[a]  = disp_ctr;
        p.st.disp_x[a]    = disp_x;
        p.st.disp_y[a]    = disp_y;
    }
}

// ---- small service kernels ------------------------------------------------------------------------------

// Agent::reset for a list of agents (Agent.cpp:123-135).
__global__ void okResetKernel(OkDeviceState st, const int32_t *idx, const float *x, const float *y, const float *rot, int n, int N)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const int a = idx[i];
    if (a < 0 || a >= N)
        return;
    st.pos_x[a]     = x[i];
    st.pos_y[a]     = y[i];
    st.rot[a]       = rot[i];
    st.acc[a]       = 0.F;
    st.speed[a]     = 0.F;
    st.crashed[a]   = 0;
    st.timed_out[a] = 0;
    st.thr[a]       = 0.F;
    st.steer[a]     = 0.F;
}

// bench initial state
__global__ void okInitBenchKernel(OkDeviceState st, const float *cx, const float *cy, const float *chead, int P, int N, int R,
                                  uint32_t agent_base, int mode)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= N)
        return;
    const uint32_t idx = ok_start_index(agent_base + static_cast<uint32_t>(a), static_cast<uint32_t>(P));
    st.pos_x[a]        = cx[idx];
    st.pos_y[a]        = cy[idx];
    st.rot[a]          = chead[idx];
    st.acc[a]          = 0.F;
    st.speed[a]        = 0.F;
    st.thr[a]          = 0.F;
    st.steer[a]        = 0.F;
    st.crashed[a]      = 0;
    st.timed_out[a]    = 0;
    st.mode[a]         = static_cast<uint8_t>(mode);
    st.disp_ctr[a]     = 0U;
    st.disp_x[a]       = 0.F;
    st.disp_y[a]       = 0.F;
    st.disp_to[a]      = 0;
    for (int r = 0; r < R; ++r)
    {
        st.hit_x[static_cast<long>(a) * R + r] = 0.F;
        st.hit_y[static_cast<long>(a) * R + r] = 0.F;
    }
}

// RaceTrack::findNearestTrackIndexBruteForce (RaceTrack.cpp:16-31): one thread per query, centre line
// staged in LDS in chunks; strict '<' so the lowest index wins ties, like the sequential scan.
__global__ void okNearestIdxKernel(const float *cx, const float *cy, int P, const float *qx, const float *qy, int n, int32_t *out)
{
    __shared__ float sx[1024], sy[1024];
    const int        i  = blockIdx.x * blockDim.x + threadIdx.x;
    const float      px = (i < n) ? qx[i] : 0.F, py = (i < n) ? qy[i] : 0.F;
    float            bestv = 3.402823466e+38F; // FLT_MAX, as in the reference
    int              arg   = 0;
    for (int base = 0; base < P; base += 1024)
    {
        const int m = (P - base < 1024) ? (P - base) : 1024;
        __syncthreads();
        for (int j = threadIdx.x; j < m; j += blockDim.x)
        {
            sx[j] = cx[base + j];
            sy[j] = cy[base + j];
        }
        __syncthreads();
        for (int j = 0; j < m; ++j)
        {
            const float dx = px - sx[j], dy = py - sy[j];
            const float d  = dx * dx + dy * dy;
            if (d < bestv)
            {
                bestv = d;
                arg   = base + j;
            }
        }
    }
    if (i < n)
        out[i] = arg;
}

__global__ void okDebugSincosKernel(const float *x, float *s, float *c, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        ok_sincosf(x[i], &s[i], &c[i]);
}

template <int kMode>
__global__ void __launch_bounds__(1024)
okDebugCastKernel(const OkStepParams p, const float *ox, const float *oy, const float *ang, int n, float *out_t)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ok_lds[];
    OkPolyView view{};
    if (kMode == kGridLds)
    {
        okStageImage(p, ok_lds);
        view.g        = p.geom;
        view.slots    = reinterpret_cast<const OkPoint *>(ok_lds);
        view.hdr      = reinterpret_cast<const uint32_t *>(ok_lds + p.off_hdr);
        view.brk      = reinterpret_cast<const uint32_t *>(ok_lds + p.off_brk);
        view.side_tol = p.side_tol;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        float sn, cs;
        ok_sincosf(ang[i], &sn, &cs);
        out_t[i] = okCastRay<kMode>(p, view, ox[i], oy[i], cs, sn);
    }
}
