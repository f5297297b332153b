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
//     stream of boundary points (8 B each, shared by chained segments) + an 8-byte header per grid cell,
//     75-95 KB for the config tracks; every point evaluation and ray-segment test then reads LDS,
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

#include <type_traits>

#include "../../include/okenv.h"
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
    kActionsMlpPolicy   = 2, // EvolutionaryRacer: per step GeneticAgent::updateAction from the previous observation
    kActionsQLearning   = 3, // RLRacers/Q_Learning: epsilon-greedy action before the step, reward + table update after it
    kActionsController  = 4, // CMA-ES racers: per step CmaEsAgent::updateAction from the previous observation, fitness bookkeeping after it
};

// Rollout bookkeeping of the current-API callers (okenv_tracker_*)
struct OkTracker
{
    int32_t  *prev_idx;     // prev_track_idx_
    float    *fitness;      // fitness_ / running return
    float    *reward;       // of the last update
    uint32_t *ep_steps;
    float    *ep_return;    // fitness when the last episode ended
    uint8_t  *prev_crashed; // crashed_ as of the previous update
};

enum OkRewardKind : int
{
    kRewardStep     = 0, // ppo_sim.cpp:77-80
    kRewardProgress = 1, // main_eigen.cpp:147-158
};

struct OkStepParams
{
    OkDeviceState st;
    int           N, R;
    int           G;        // lanes per agent (power of two, <= 64)
    int           rays_per_lane;
    const float  *ray_deg;  // [R]
    float         sensor_offset;
    // compact image in global memory: [slots | hdr], byte offsets from `image` (ok_grid.h)
    const uint8_t *image;
    uint32_t       image_bytes, off_hdr;
    float          side_tol;
    OkGridGeom     geom;
    // front / back split (ok_grid.h: okClassifyFrontBack), cooperative kernel only: fb != 0 means `image` holds [front image |
    // back image] over the one geometry -- off_hdr / side_tol above are the front image's, the back image's slots start at
    // fb_back_off and its headers at fb_back_off_hdr (bytes from `image`) -- and a step walks the back image only for rays that
    // need it.  e_s / e_t: the exact test's ambiguity bounds; t12 / t34: those of the origin test (ok_raycast.h).
    uint32_t       fb, fb_back_off, fb_back_off_hdr;
    float          fb_back_side_tol, fb_e_s, fb_e_t, fb_t12, fb_t34;
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
    // device-side Environment::resetAgent of crashed agents at the start of a step (okenv_set_auto_reset):
    // reset_flags = OK_RESET_* | kAutoResetOn, inner lane boundaries as xy pairs [P][2]
    uint32_t     reset_flags, reset_seed;
    const float *lane_l, *lane_r;
    // epoch of the auto-reset draws: step_counter[0] + s.  Device-resident so that a captured hipGraph of the step
    // replays with advancing epochs: the last workgroup to finish a launch adds n_steps (step_counter[1] counts the
    // finished workgroups).
    uint32_t *step_counter;
    // EvolutionaryRacer policy weights, OK_MLP_WEIGHTS(R) floats per agent (layout in okenv_math.h)
    const float *mlp_w;
    // RLRacers/Q_Learning: per-agent table [N][243][3], current state / action / previous track index, the five
    // rays that feed the state, epsilon of this rollout
    float   *q_table;
    int32_t *q_state, *q_action, *q_prev_idx;
    // centre-line points bucketed by the cells of `geom` (CSR: point indices of cell c are cl_idx[cl_start[c] ..
    // cl_start[c + 1]), ascending); nullptr: scan the whole centre line every step
    const uint16_t *cl_start, *cl_idx;
    int      q_ray[5];
    float    q_epsilon;
    // CMA-ES racers (okenv_rollout_controller): the candidates' controller parameters [N][ctrl_num_params] (Controller.cpp's
    // parameters() order), the constant throttle and the steering scale of CmaEsAgent::updateAction, and the bookkeeping the
    // callers run after env.step() (okenv_tracker_*), all inside the step kernel
    const float *ctrl_params;
    int          ctrl_num_params, ctrl_hidden;
    float        ctrl_throttle, ctrl_steer_scale;
    OkTracker    trk;
    int          trk_kind;
    uint32_t     ctrl_lds_off; // byte offset in the workgroup's LDS where its agents' parameters are staged for the launch (0: read from global memory)
    // Packed host exchange (okenv_step_packed, the C++ facade's Environment::step): when set, the step kernel takes the
    // agents' state from `rec_in` and leaves state and sensor_hits_ (x, y pairs) in `rec_out` / `hits_xy_out`, all three in
    // host memory mapped into the device, so that a facade step is ONE kernel with no staging copies around it.
    const okenv_agent_record *rec_in;
    okenv_agent_record       *rec_out;
    float                    *hits_xy_out;
    int                       rec_with_stats; // DisplacementStats travel in the records (else the device keeps its own)
    // ... and the launch announces its end there as well: the last workgroup to finish stores `done_seq` to `done_flag`
    // (mapped host memory) after every workgroup's results have left the device; the host spins on that word instead of
    // waiting for the stream's completion signal (about 5 us less per facade step).  nullptr: no announcement.
    uint32_t *done_flag;
    uint32_t  done_seq;
    // cooperative kernel, tiny populations: workgroup b holds agents [b * agents_per_block, (b + 1) * agents_per_block) in its
    // first lanes and the rest of its lanes only help with staging the image.  0: agents are packed densely over the grid.
    int agents_per_block;
    // resident form of the packed exchange (okStepCoopKernel<.., true, true>): one 64-byte slot per agent in mapped host memory,
    // words 3, 7, 11, 15 = sequence number, the other twelve = the agent's record (word j at j + j / 3); idle_ticks: how long
    // (100 MHz ticks) a workgroup waits for work before it leaves
    const uint32_t *slots;
    uint32_t        idle_ticks;
    // Episodes of the population callers ("step everybody until every agent has crashed", genetic_learner_sim.cpp:75-95,
    // q_racer_sim.cpp:156-190; host side: okenv_episode_begin / _compact / _end).  Policy kernels only; all nullptr otherwise.
    //   active / n_active   the agents this launch steps (ascending ids): slot i of the grid holds agent active[i]
    //   settled             1 = the agent is crashed AND has taken one step as a crashed agent: from then on a step changes nothing
    //                       about it (it does not move, its DisplacementStats do not tick, its rays keep their stale hit points
    //                       seen from an origin that no longer moves, the MLP policy sees the same inputs) -- the host drops it from
    //                       `active`, a wave whose agents are all settled leaves the step loop
    //   crash_step          episode-local index (1, 2, ...) of the step in which the agent crashed (0: it was crashed when the
    //                       episode began, 0xFFFFFFFF: not yet); crash_thr / crash_steer: its action in that step.  The reference's
    //                       loop ends with the step T in which the LAST agent crashes; launches overrun T, and okenv_episode_end puts
    //                       back what the overrun changed (the action of the agents that crashed in step T)
    //   live                sum over steps of the agents that entered the step alive (one atomic per agent and launch)
    //   ep_step0            episode steps taken before this launch
    //   q_next_state        Q-learning: the state index seen in the crash step (it stays the "next state" of every later step);
    //                       crashed agents' table updates are NOT made by the step kernel but by okQSettleKernel, once T is known
#if defined(OKENV_STAMPS)
    unsigned long long *stamps; // diagnostic build: kStampWords words per wave of the launch (host: launchStep sizes it from the grid)
#endif
    const int32_t      *active;
    int                 n_active;
    uint8_t            *settled;
    uint32_t           *crash_step;
    float              *crash_thr, *crash_steer;
    unsigned long long *live;
    uint32_t            ep_step0;
    int32_t            *q_next_state;
};

// sequence numbers of packed steps (host and device count alike): 0 = nothing yet, 0xFFFFFFFF = "leave"
__host__ __device__ inline uint32_t okNextPackedSeq(uint32_t s)
{
    ++s;
    return (s == 0U || s == 0xFFFFFFFFU) ? 1U : s;
}

constexpr uint32_t kAutoResetOn = 0x80000000U;
// diagnostic build (-DOKENV_STAMPS), per wave: [0] policy, [1] pre-step, [3] phase 1, [5] phase 2, [6] epilogue (+ Q-learning)
// shader cycles summed over the launch's steps; [2] / [4] wave start / end on the 100 MHz clock; [8..17] walk-internal stamps
constexpr int kStampWords = 24;

// Compile-time policy selector of the step kernels, so that the headline path carries no policy registers.
enum OkPolicyKind : int
{
    kPolicyNone = 0, // actions stored by the host, or the bench driver's Philox actions
    kPolicyMlp  = 1, // EvolutionaryRacer
    kPolicyQ    = 2, // RLRacers/Q_Learning
    kPolicyCtrl = 3, // CMA-ES racers: controller before the step, fitness bookkeeping after it
};

enum OkGridMode : int
{
    kGridLds    = 0, // compact image staged into LDS
    kGridGlobal = 1, // wide CSR form read from global memory
    kGridBrute  = 2, // no grid: sweep all segments (the reference kernel's algorithm; testing/ablation)
};

// Cross-lane moves that stay in the VALU (no LDS round trip): DPP inside a row of 16 lanes, the gfx950 lane-swap
// instructions across rows.
template <int kCtrl>
__device__ __forceinline__ float okDppMove(const float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), kCtrl, 0xF, 0xF, true));
}
constexpr int kDppXor1       = 0xB1;  // quad_perm [1,0,3,2]
constexpr int kDppXor2       = 0x4E;  // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // lane i <- lane 7 - i of its group of 8
constexpr int kDppMirror     = 0x140; // lane i <- lane 15 - i of its row

// An index the compiler cannot see through: the addresses formed from it are formed where it is used instead of being kept in
// registers (or scratch) from the kernel's start.
__device__ __forceinline__ int okOpaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// min over the G lanes that belong to one agent; every lane of the group receives the result.
// (a < b ? a : b) keeps the sequential loop's "NaN never wins" behaviour (CollisionChecker.cu:161-164).
// The butterfly's partners are lane ^ 1, lane ^ 2, then the mirrored lane of the group of 8 / of the row (once the
// groups of 4 / 8 agree, the mirrored lane holds what lane ^ 4 / lane ^ 8 holds), then the other row / other half of
// the wave through v_permlane16_swap / v_permlane32_swap: six VALU steps, no ds_bpermute.
__device__ __forceinline__ float okGroupMin(float v, const int G)
{
#if defined(OKENV_SHUFFLE_GROUPMIN) // ablation: the ds_bpermute butterfly
    for (int off = 1; off < G; off <<= 1)
    {
        const float o = __shfl_xor(v, off, 64);
        v             = (o < v) ? o : v;
    }
    return v;
#endif
    float o;
    if (G > 1)
    {
        o = okDppMove<kDppXor1>(v);
        v = (o < v) ? o : v;
    }
    if (G > 2)
    {
        o = okDppMove<kDppXor2>(v);
        v = (o < v) ? o : v;
    }
    if (G > 4)
    {
        o = okDppMove<kDppHalfMirror>(v);
        v = (o < v) ? o : v;
    }
    if (G > 8)
    {
        o = okDppMove<kDppMirror>(v);
        v = (o < v) ? o : v;
    }
    // (inline asm: with the builtin, hipcc 7.2 drops the min that follows the swap -- it treats the two results as equal)
    if (G > 16)
    {
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        v = (a < b) ? a : b;
    }
    if (G > 32)
    {
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        v = (a < b) ? a : b;
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
        return ok_cast_ray_poly<false>(lds_view, ox, oy, rdx, rdy, nullptr, nullptr, nullptr);
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

// Stage the compact grid image into LDS with 16-byte loads (image_bytes is a multiple of 16).  Eight loads are issued
// back to back before their stores so that the L2 round trips overlap: with one launch per Environment step the
// staging is a visible part of the launch, and a small workgroup (few agents) has few lanes to spread it over.
__device__ __forceinline__ void okStageImage(const OkStepParams &p, unsigned char *lds)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(p.image);
    uint4       *dst = reinterpret_cast<uint4 *>(lds);
    const int    n16 = static_cast<int>(p.image_bytes >> 4);
    const int    bd  = static_cast<int>(blockDim.x);
    int          i   = static_cast<int>(threadIdx.x);
    constexpr int kDepth = 8;
    for (; i + (kDepth - 1) * bd < n16; i += kDepth * bd)
    {
        uint4 v[kDepth];
#pragma unroll
        for (int k = 0; k < kDepth; ++k)
            v[k] = src[i + k * bd];
#pragma unroll
        for (int k = 0; k < kDepth; ++k)
            dst[i + k * bd] = v[k];
    }
    for (; i < n16; i += bd)
        dst[i] = src[i];
    __syncthreads();
}

// End of a step launch, by the last workgroup to get here (step_counter[1] counts the finished ones): advance the
// device-side step counter once -- every workgroup has read it for the last time before its own arrival, so the writer runs
// after all readers -- and, for the packed host exchange, tell the host that all results are in its memory.
__device__ __forceinline__ void okFinishLaunch(const OkStepParams &p)
{
    const bool advance = (p.reset_flags & 0x80000000U) != 0U;
    if (!advance && p.done_flag == nullptr)
        return;
    if (p.done_flag != nullptr)
        __threadfence_system(); // each wave: its stores to the host have been acknowledged before it reports in
    __syncthreads();
    if (threadIdx.x == 0)
    {
        __threadfence();
        if (atomicAdd(&p.step_counter[1], 1U) == gridDim.x - 1U)
        {
            if (advance)
                p.step_counter[0] += static_cast<uint32_t>(p.n_steps);
            p.step_counter[1] = 0U;
            if (p.done_flag != nullptr)
            {
                __threadfence_system();
                __hip_atomic_store(p.done_flag, p.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Per-agent state carried in registers across the steps of one launch (every lane of the agent's group holds a
// copy; lane 0 of the group writes it back).
struct OkAgentRegs
{
    float    pos_x, pos_y, rot, speed, acc, thr, steer;
    int      mode;
    bool     crashed, timed_out, disp_to;
    uint32_t disp_ctr;
    float    disp_x, disp_y;
};

__device__ __forceinline__ OkAgentRegs okLoadAgent(const OkDeviceState &st, const int a)
{
    OkAgentRegs r;
    r.pos_x     = st.pos_x[a];
    r.pos_y     = st.pos_y[a];
    r.rot       = st.rot[a];
    r.speed     = st.speed[a];
    r.acc       = st.acc[a];
    r.thr       = st.thr[a];
    r.steer     = st.steer[a];
    r.mode      = st.mode[a];
    r.crashed   = st.crashed[a] != 0;
    r.timed_out = st.timed_out[a] != 0;
    r.disp_to   = st.disp_to[a] != 0;
    r.disp_ctr  = st.disp_ctr[a];
    r.disp_x    = st.disp_x[a];
    r.disp_y    = st.disp_y[a];
    return r;
}

__device__ __forceinline__ void okStoreAgent(const OkDeviceState &st, int a, const OkAgentRegs &r)
{
    // The write-back happens once per launch, after the step loop.  Without this the compiler keeps the thirteen addresses it
    // formed for okLoadAgent alive across the loop (26 VGPRs, or -- in the policy kernels, which have none to spare -- scratch);
    // an index it cannot see through makes it form them again here.
    asm volatile("" : "+v"(a));
    st.pos_x[a]     = r.pos_x;
    st.pos_y[a]     = r.pos_y;
    st.rot[a]       = r.rot;
    st.speed[a]     = r.speed;
    st.acc[a]       = r.acc;
    st.thr[a]       = r.thr;
    st.steer[a]     = r.steer;
    st.crashed[a]   = r.crashed ? 1 : 0;
    st.timed_out[a] = r.timed_out ? 1 : 0;
    st.disp_to[a]   = r.disp_to ? 1 : 0;
    st.disp_ctr[a]  = r.disp_ctr;
    st.disp_x[a]    = r.disp_x;
    st.disp_y[a]    = r.disp_y;
}

// Everything Environment::step does to one agent BEFORE the collision pass (Environment.cpp:128-142), preceded by
// the optional bench driver (reset of crashed agents + Philox action).  Also returns
// sin/cos(kDeg2Rad * rot_) of the pose the collision pass will see: Agent::move, the ray build and the hit transform
// all take the sine and cosine of that same angle (Agent.cpp:94-97, CollisionChecker.cu:121-124,157-158), so it is
// evaluated once per agent and step.
__device__ __forceinline__ void okAgentPreStep(const OkStepParams &p,
                                               OkAgentRegs        &r,
                                               const int           a,
                                               const int           s,
                                               float              &sn,
                                               float              &cs,
                                               const float         ray_deg = 0.F,
                                               float              *ray_sn  = nullptr,
                                               float              *ray_cs  = nullptr,
                                               const bool          have_drawn = false,
                                               const ok_random_action drawn = ok_random_action{}, // by value: a pointer here put it on the stack
                                               const bool          pair_split = false)
{
    if ((p.reset_flags & kAutoResetOn) != 0U && r.crashed)
    {
        // Environment::resetAgent (Environment.cpp:79-122) + Agent::reset (Agent.cpp:123-135): the step that follows
        // runs with the zeroed action, i.e. it is the "initial observation" step the callers take after a reset
        // (ppo_sim.cpp:58-60, main_eigen.cpp:127-128); DisplacementStats deliberately untouched
        const uint32_t      ag    = p.agent_base + static_cast<uint32_t>(a);
        const uint32_t      epoch = p.step_counter[0] + static_cast<uint32_t>(s);
        const ok_reset_draw d     = ok_draw_reset(p.reset_seed, ag, epoch, ag + epoch, static_cast<uint32_t>(p.P), p.reset_flags);
        ok_reset_pose(d, p.cx, p.cy, p.chead, p.lane_l, p.lane_r, &r.pos_x, &r.pos_y, &r.rot);
        r.acc       = 0.F;
        r.speed     = 0.F;
        r.crashed   = false;
        r.timed_out = false;
        r.thr       = 0.F;
        r.steer     = 0.F;
    }
    if (p.action_source == kActionsPhiloxReset)
    {
        // the caller may have drawn this step's action already (okStepCoopKernel draws a block of steps at a time)
        const ok_random_action ra =
            have_drawn ? drawn : ok_draw_random_action(p.seed, p.agent_base + static_cast<uint32_t>(a), p.step_base + static_cast<uint32_t>(s));
        if (r.crashed)
        {
            // Agent::reset (Agent.cpp:123-135); DisplacementStats deliberately untouched
            const uint32_t idx = ok_index_from_word(ra.reset_word, static_cast<uint32_t>(p.P));
            r.pos_x            = p.cx[idx];
            r.pos_y            = p.cy[idx];
            r.rot              = p.chead[idx];
            r.acc              = 0.F;
            r.speed            = 0.F;
            r.crashed          = false;
            r.timed_out        = false;
        }
        r.thr   = ra.throttle;
        r.steer = ra.steer;
    }
    const bool moves = p.do_move && !r.crashed && (r.mode == 0 || r.mode == 1); // MANUAL: empty in the reference
    if (moves)
    {
        if (r.mode == 0)
        { // moveViaVelocity (Agent.cpp:108-119)
            r.rot += r.steer;
            r.speed = r.thr;
        }
        else
        { // moveViaAcceleration (Agent.cpp:82-98)
            r.rot += r.steer;
            r.acc += r.thr;
            r.speed += (r.acc * OK_DT);
            r.speed = (r.speed < 0.F) ? 0.F : r.speed;
            r.speed = (r.speed > OK_SPEED_LIMIT) ? OK_SPEED_LIMIT : r.speed;
        }
    }
    // The lane's ray direction (CollisionChecker.cu:125, angle = kDeg2Rad * (rot_ + ray angle)) depends on nothing but
    // the heading just updated: evaluated here, next to the agent's own sine/cosine, the two independent fp64 chains
    // interleave instead of running back to back.
    // pair_split (the tail kernel, where lanes 2i and 2i + 1 hold the same agent AND the same ray): the even lane takes the
    // ray's angle, the odd lane the agent's, and they swap results -- one fp64 chain per lane instead of two.
    if (pair_split)
    {
        const bool odd = (threadIdx.x & 1U) != 0U;
        float      s1, c1;
        ok_sincosf(odd ? OK_DEG2RAD * r.rot : OK_DEG2RAD * (r.rot + ray_deg), &s1, &c1);
        const float s2 = okDppMove<kDppXor1>(s1), c2 = okDppMove<kDppXor1>(c1);
        sn      = odd ? s1 : s2;
        cs      = odd ? c1 : c2;
        *ray_sn = odd ? s2 : s1;
        *ray_cs = odd ? c2 : c1;
    }
    else
    {
        if (ray_sn != nullptr)
            ok_sincosf(OK_DEG2RAD * (r.rot + ray_deg), ray_sn, ray_cs);
        ok_sincosf(OK_DEG2RAD * r.rot, &sn, &cs);
    }
    if (moves)
    {
        const float dx = cs * r.speed * OK_DT;
        r.pos_x += dx;
        const float dy = sn * r.speed * OK_DT;
        r.pos_y += dy;
    }
    if (p.do_move && !r.crashed)
    {
        // checkAndUpdateStandstill (Environment.cpp:16-39)
        if (r.disp_ctr == 0U)
        {
            r.disp_x   = r.pos_x;
            r.disp_y   = r.pos_y;
            r.disp_to  = false;
            r.disp_ctr = 1U;
        }
        else if (r.disp_ctr >= OK_DISP_PERIOD)
        {
            const float ddx = r.pos_x - r.disp_x, ddy = r.pos_y - r.disp_y;
            const float d2  = ddx * ddx + ddy * ddy;
            if (d2 < OK_DISP_THRESH2)
                r.disp_to = true;
            r.disp_ctr = 0U;
        }
        else
        {
            r.disp_to = false;
            ++r.disp_ctr;
        }
        if (r.disp_to)
        {
            r.crashed   = true;
            r.timed_out = true;
        }
    }
}

// Hit transform of one ray (CollisionChecker.cu:144-166): writes sensor_hits_ (robot frame) and its norm, returns the
// squared norm for the crash test.
__device__ __forceinline__ float okRayEpilogue(const OkDeviceState &st,
                                               const long           k,
                                               const float          hx,
                                               const float          hy,
                                               const float          ox,
                                               const float          oy,
                                               const float          sr,
                                               const float          cr,
                                               float               &dist_out,
                                               float               *rel_x_out = nullptr,
                                               float               *rel_y_out = nullptr)
{
    const float xt = hx - ox;
    const float yt = hy - oy;
    const float rx = xt * cr - yt * sr;
    const float ry = xt * sr + yt * cr;
    const float n2 = rx * rx + ry * ry;
    st.rel_x[k]    = rx;
    st.rel_y[k]    = ry;
    if (rel_x_out != nullptr)
    {
        *rel_x_out = rx;
        *rel_y_out = ry;
    }
    dist_out       = __builtin_sqrtf(n2);
    st.dist[k]     = dist_out;
    return n2;
}

// GeneticAgent::updateAction (EvolutionaryRacer/GeneticAgent.hpp:37-107) with Network::infer
// (EvolutionaryRacer/Network.hpp:119-155): inputs [speed/100, normalizeAngleDeg(rot)/360, |hit_r|/200 ...] ->
// relu(x W1) -> (x W2) > 0 decode.  Runs inside the step kernel, before the move, from the previous step's
// observation: input r lives in the lane that owns ray r and is broadcast with wave shuffles; hidden unit u is
// accumulated by lane u % 32 of the agent's group in the fixed order j = 0..R+1, output k by lane k in the order
// i = 0..31 (the oracle accumulates in the same orders; Eigen's own order is unpinned).
// Must be called by every lane of the group.  kUnits = hidden units per owning lane = 32 / min(G, 32).
template <int kUnits>
__device__ __forceinline__ void
okMlpAction(const OkStepParams &p, const int a, const int rlane, const int G, OkAgentRegs &ag, const float dist_self, const bool ray_ok)
{
    const float *w1  = p.mlp_w + static_cast<size_t>(a) * OK_MLP_WEIGHTS(p.R);
    const float *w2  = w1 + (p.R + 2) * OK_MLP_HID_PAD;
    const float  x0  = ag.speed / 100.0F;
    const float  x1  = ok_normalize_angle_deg(ag.rot) / 360.0F;
    const float  xs  = ray_ok ? dist_self / 200.0F : 0.F;
    const int    own = OK_MLP_HID_PAD / kUnits; // lanes of the group that own hidden units (= min(G, 32))
    const int    ul  = rlane < own ? rlane : own - 1;
    // Weights stream from L2 / Infinity Cache (5.3 KB per agent and step).  They are fetched eight rows at a time before
    // the multiply-adds that consume them, so that a lane waits for one memory round trip per eight terms instead of one
    // per term; the additions keep their order (input 0, 1, 2, ... / hidden unit 0, 1, 2, ...), hence the same bits.
    constexpr int kAhead = 8;
    float         h[kUnits];
#pragma unroll
    for (int u = 0; u < kUnits; ++u)
    {
        const int unit = ul + u * own;
        float     acc  = 0.F;
        acc            = acc + x0 * w1[0 * OK_MLP_HID_PAD + unit];
        acc            = acc + x1 * w1[1 * OK_MLP_HID_PAD + unit];
        for (int j0 = 0; j0 < p.R; j0 += kAhead)
        {
            float w[kAhead];
#pragma unroll
            for (int k = 0; k < kAhead; ++k)
                w[k] = (j0 + k < p.R) ? w1[(2 + j0 + k) * OK_MLP_HID_PAD + unit] : 0.F;
#pragma unroll
            for (int k = 0; k < kAhead; ++k)
                if (j0 + k < p.R)
                    acc = acc + __shfl(xs, j0 + k, G) * w[k];
        }
        h[u] = (acc > 0.F) ? acc : 0.F;
    }
    const int kl = rlane < OK_MLP_OUT_PAD ? rlane : OK_MLP_OUT_PAD - 1;
    float     z  = 0.F;
#pragma unroll
    for (int u = 0; u < kUnits; ++u)
    {
        for (int l0 = 0; l0 < own; l0 += kAhead)
        {
            float w[kAhead];
#pragma unroll
            for (int k = 0; k < kAhead; ++k)
                w[k] = (l0 + k < own) ? w2[(l0 + k + u * own) * OK_MLP_OUT_PAD + kl] : 0.F;
#pragma unroll
            for (int k = 0; k < kAhead; ++k)
                if (l0 + k < own)
                { // hidden unit index l0 + k + u * own, visited in increasing order
                    const float hv = __shfl(h[u], l0 + k, G);
                    z              = z + hv * w[k];
                }
        }
    }
    float zs[OK_MLP_OUT];
#pragma unroll
    for (int k = 0; k < OK_MLP_OUT; ++k)
        zs[k] = __shfl(z, k, G);
    ok_ga_decode_action(zs, &ag.thr, &ag.steer);
}

// The same for a fan whose width is known at compile time and a group of >= 32 lanes (one hidden unit per lane).  The lane's
// column of w1 -- kR + 2 weights that never change during a launch -- is held in registers from the END of one step (requested
// after the raycast, whose temporaries are dead by then) to the policy at the start of the next: the memory round trip (the
// weights stream from L2 / Infinity Cache: 43 MB per 8192 agents, more than the L2s hold) runs under the step's epilogue
// instead of in front of the policy; the output lanes' column of w2 travels with it.  Same terms in the same order as
// okMlpAction, hence the same bits.
template <int kR>
struct OkMlpColumn
{
    float w[kR + 2];          // this lane's hidden unit: its weight for every input
    float v[OK_MLP_HID_PAD];  // this lane's output (lanes 0..5; the others hold a copy of lane 7's): its weight for every hidden unit
};

template <int kR>
__device__ __forceinline__ OkMlpColumn<kR> okMlpFetchColumn(const OkStepParams &p, const int a, const int rlane)
{
    const float *w1 = p.mlp_w + static_cast<size_t>(a) * OK_MLP_WEIGHTS(kR);
    const int    ul = rlane < OK_MLP_HID_PAD ? rlane : OK_MLP_HID_PAD - 1;
    const float *w2 = w1 + (kR + 2) * OK_MLP_HID_PAD;
    const int    kl = rlane < OK_MLP_OUT_PAD ? rlane : OK_MLP_OUT_PAD - 1;
    OkMlpColumn<kR> c;
#pragma unroll
    for (int j = 0; j < kR + 2; ++j)
        c.w[j] = w1[j * OK_MLP_HID_PAD + ul];
#pragma unroll
    for (int i = 0; i < OK_MLP_HID_PAD; ++i)
        c.v[i] = w2[i * OK_MLP_OUT_PAD + kl];
    return c;
}

template <int kR>
__device__ __forceinline__ void okMlpActionWide(const OkStepParams &p, const int a, const int rlane, const int G, OkAgentRegs &ag, const float dist_self,
                                                const bool ray_ok, const OkMlpColumn<kR> col) // by value: stays in registers
{
    const float  x0 = ag.speed / 100.0F;
    const float  x1 = ok_normalize_angle_deg(ag.rot) / 360.0F;
    const float  xs = ray_ok ? dist_self / 200.0F : 0.F;
    float        acc = 0.F;
    acc              = acc + x0 * col.w[0];
    acc              = acc + x1 * col.w[1];
#pragma unroll
    for (int j = 0; j < kR; ++j)
        acc = acc + __shfl(xs, j, G) * col.w[2 + j];
    const float h = (acc > 0.F) ? acc : 0.F;
    float       z = 0.F;
#pragma unroll
    for (int i = 0; i < OK_MLP_HID_PAD; ++i)
        z = z + __shfl(h, i, G) * col.v[i];
    float zs[OK_MLP_OUT];
#pragma unroll
    for (int k = 0; k < OK_MLP_OUT; ++k)
        zs[k] = __shfl(z, k, G);
    ok_ga_decode_action(zs, &ag.thr, &ag.steer);
}

__device__ __forceinline__ void
okPolicyAction(const OkStepParams &p, const int a, const int rlane, const int G, OkAgentRegs &ag, const float dist_self, const bool ray_ok)
{
    if (G >= 32)
        okMlpAction<1>(p, a, rlane, G, ag, dist_self, ray_ok);
    else if (G == 16)
        okMlpAction<2>(p, a, rlane, G, ag, dist_self, ray_ok);
    else
        okMlpAction<4>(p, a, rlane, G, ag, dist_self, ray_ok); // G == 8 (the C ABI refuses narrower fans)
}

// ---- CMA-ES racers' pieces of the fused rollout (okenv_rollout_controller) -----------------------------------------------

// CmaEsAgent::updateAction (CovarianceMatrixAdaptationEvolution/main_eigen.cpp:45-68, Controller.cpp:3-23) by the G lanes of the
// agent's group: tanh(fc3(tanh(fc2(tanh(fc1(x)))))), x = the agent's distances / kSensorRange (ray i's in lane i), fc1: R ->
// hidden, fc2: hidden -> hidden / 2, fc3: hidden / 2 -> 2 (only output 0 is used: steering; the throttle is a constant).  Lane r
// evaluates units r, r + G, ... of a layer; the previous layer's outputs come by shuffles; every sum starts from the bias and adds
// w * x in ascending input order (fp32, no FMA) -- the same terms in the same order as okControllerKernel and the oracle's
// ctrl_forward, whatever G.  At most kCtrlUnitsPerLane units of a layer per lane (hidden <= 4 G; the host checks it): with eight the
// kernel needs scratch.
constexpr int kCtrlUnitsPerLane = 4;

// One layer: out[q] = B[u] + sum over i of W[u * n_in + i] * input i, for the lane's units u = r + q * G < n_out; input i sits in
// lane i % G, register in[i / G].  Ascending i, one multiply and one add per term.
// kCtrlBatch: inputs whose weights are fetched together (the loads of a batch are in flight at once)
template <int kCtrlBatch, int kIn, int kOut>
__device__ __forceinline__ void
okCtrlLayer(float (&out)[kOut], const float (&in)[kIn], const float *W, const float *B, const int n_in, const int n_out, const int r, const int G)
{
#pragma unroll
    for (int q = 0; q < kOut; ++q)
    {
        const int u = r + q * G;
        out[q]      = (u < n_out) ? B[u] : 0.F;
    }
    const int g_shift = 31 - __builtin_clz(G);
    for (int i0 = 0; i0 < n_in; i0 += kCtrlBatch)
    {
        float w[kOut][kCtrlBatch], xs[kCtrlBatch];
#pragma unroll
        for (int q = 0; q < kOut; ++q)
        {
            const int u = r + q * G;
#pragma unroll
            for (int j = 0; j < kCtrlBatch; ++j)
                w[q][j] = (u < n_out && i0 + j < n_in) ? W[u * n_in + i0 + j] : 0.F;
        }
#pragma unroll
        for (int j = 0; j < kCtrlBatch; ++j)
        {
            const int i   = i0 + j;
            const int reg = i >> g_shift; // (the same in every lane)
            float     v   = in[0];
#pragma unroll
            for (int k = 1; k < kIn; ++k)
                v = (reg == k) ? in[k] : v;
            xs[j] = __shfl(v, i & (G - 1), G);
        }
#pragma unroll
        for (int j = 0; j < kCtrlBatch; ++j)
        {
            if (i0 + j < n_in)
            {
#pragma unroll
                for (int q = 0; q < kOut; ++q)
                    if (r + q * G < n_out)
                        out[q] = out[q] + w[q][j] * xs[j];
            }
        }
    }
}

// kUnits: units of a layer a lane may have to evaluate (1 when hidden <= G, the usual case: bigger batches fit the registers then)
template <int kCtrlBatch, int kUnits>
__device__ __forceinline__ void okCtrlAction(const OkStepParams &p, const float *prm, const int r, const int G, OkAgentRegs &ag, const float dist_self)
{
    const int    R = p.R, H = p.ctrl_hidden, H2 = H / 2;
    const float *w1 = prm, *b1 = w1 + H * R, *w2 = b1 + H, *b2 = w2 + H2 * H, *w3 = b2 + H2, *b3 = w3 + 2 * H2;
    const float  x[1] = {dist_self / OK_SENSOR_RANGE};
    constexpr int kUnits2 = kUnits > 1 ? kUnits / 2 : 1;
    float         a1[kUnits], a2[kUnits2], a3[1];
    okCtrlLayer<kCtrlBatch>(a1, x, w1, b1, R, H, r, G);
#pragma unroll
    for (int q = 0; q < kUnits; ++q)
        a1[q] = (r + q * G < H) ? ok_tanhf(a1[q]) : 0.F;
    okCtrlLayer<kCtrlBatch>(a2, a1, w2, b2, H, H2, r, G);
#pragma unroll
    for (int q = 0; q < kUnits2; ++q)
        a2[q] = (r + q * G < H2) ? ok_tanhf(a2[q]) : 0.F;
    okCtrlLayer<kCtrlBatch>(a3, a2, w3, b3, H2, 1, r, G); // output 0 (steering), by lane 0
    ag.thr   = p.ctrl_throttle;
    ag.steer = __shfl(ok_tanhf(a3[0]) * p.ctrl_steer_scale, 0, G);
}

// okTrackerKernel's update (begin == 0) on values carried in registers: the callers' loop body after env.step()
// (main_eigen.cpp:143-158, ppo_sim.cpp:77-80); `idx` is the nearest centre-line index of the new position (PROGRESS only).
struct OkTrackerRegs
{
    int      prev_idx;
    float    fitness, reward, ep_return;
    uint32_t ep_steps;
    bool     prev_crashed;
};

__device__ __forceinline__ void okTrackerStep(OkTrackerRegs &t, const int kind, const int idx, const bool crashed, const bool timed_out)
{
    float reward = 0.F;
    if (t.prev_crashed && !crashed)
    { // re-placed since the last update: this step was the new episode's initial observation
        t.fitness  = 0.F;
        t.ep_steps = 0U;
        t.prev_idx = idx;
    }
    else if (kind == kRewardStep)
    {
        reward = 1.F;
        t.fitness += 1.F;
        t.ep_steps += 1U;
    }
    else if (!crashed)
    {
        const int progress = idx - t.prev_idx;
        t.prev_idx         = idx;
        reward             = static_cast<float>(progress < 0 ? -progress : progress);
        t.fitness += reward;
        t.ep_steps += 1U;
    }
    else if (timed_out)
    {
        t.fitness = 0.F;
    }
    t.reward = reward;
    if (crashed && !t.prev_crashed)
        t.ep_return = t.fitness;
    t.prev_crashed = crashed;
}

template <int kMode>
__device__ __forceinline__ OkPolyView okSetupView(const OkStepParams &p, unsigned char *lds)
{
    OkPolyView view{};
    if (kMode == kGridLds)
    {
        okStageImage(p, lds);
        view.g        = p.geom;
        view.slots    = reinterpret_cast<const OkPoint *>(lds);
        view.hdr      = reinterpret_cast<const OkCellHdr *>(lds + p.off_hdr);
        view.side_tol = p.side_tol;
        view.e_s      = p.fb_e_s;
        view.e_t      = p.fb_e_t;
        view.e_s_over_e_t = p.fb_e_t > 0.F ? p.fb_e_s / p.fb_e_t : 0.F;
    }
    return view;
}

// chi(origin) = 1 for certain?  (front / back split: ok_grid.h has the argument, ok_raycast.h's okOriginChiScalar the test as one
// thread makes it.)  Here the pairs of the origin cell's front chunk are dealt to the G lanes of the agent's group, lane r taking
// pairs r, r + G, ...; the classes are the scalar test's, their parity and "any pair that cannot be told" come together by two
// ballots.  Every lane of the group returns the same answer.
// *clear: a radius around the origin inside which the answer cannot change -- chi is a function of the position alone and flips
// only across F, so it holds wherever no F segment can be reached: a lower bound of the distance to the F segments registered
// in the cell (each by its bounding box) and to the cell's own border (every other F segment lies beyond that).  The step loop
// repeats the test only when the origin has left that circle (an agent moves 1.6 px per step at most).
__device__ __forceinline__ bool okOriginChiGroup(const OkPolyView &front, const float ox, const float oy, const int r, const int G, const float t12,
                                                 const float t34, float *clear)
{
    const OkGridGeom &g      = front.g;
    const bool        inside = ox >= g.x0 && ox <= g.x1 && oy >= g.y0 && oy <= g.y1; // (false for NaN poses)
    int               ix     = (int)__builtin_floorf((ox - g.x0) * g.inv_cell);
    int               iy     = (int)__builtin_floorf((oy - g.y0) * g.inv_cell);
    ix                       = ix < 0 ? 0 : (ix >= g.nx ? g.nx - 1 : ix);
    iy                       = iy < 0 ? 0 : (iy >= g.ny ? g.ny - 1 : iy);
    const OkCellHdr hc       = front.hdr[inside ? iy * g.nx + ix : 0];
    const uint32_t  flags    = (hc.w0 >> OKFB_HDR_SHIFT_RC) & 15U;
    const bool      usable   = inside && (flags & OKFB_RC_CERT) != 0U;
    float           rx, ry;
    okCellRefPoint(g, ix, iy, (flags >> 2) & 3U, &rx, &ry);
    const uint32_t k0 = hc.w0 & OKPOLY_IDX_MASK;
    const uint32_t n8 = (((hc.w0 >> OKPOLY_IDX_BITS) & OKPOLY_N_MASK) + 7U) & ~7U;
    const uint32_t n  = usable ? (hc.w0 >> OKFB_HDR_NF_SHIFT) : 0U; // the slots of the cell's F segments: the first of its chunk
    bool           amb = false, odd = false;
    // distance to the cell's border (the cell's corners in the walk's own arithmetic)
    const float cx0 = g.x0 + static_cast<float>(ix) * g.cell, cy0 = g.y0 + static_cast<float>(iy) * g.cell;
    float       near = __builtin_fminf(__builtin_fminf(ox - cx0, (cx0 + g.cell) - ox), __builtin_fminf(oy - cy0, (cy0 + g.cell) - oy));
    for (uint32_t j = static_cast<uint32_t>(r); j + 1U < n; j += static_cast<uint32_t>(G))
    {
        if ((hc.brk >> (n8 - 2U - j)) & 1U) // no segment joins slots j and j + 1
            continue;
        const OkPoint pa = front.slots[k0 + j], pb = front.slots[k0 + j + 1U];
        const int     cls = okChiPairClass(pa, pb, rx, ry, ox, oy, t12, t34);
        amb               = amb || cls == 2;
        odd               = odd != (cls == 1);
        // Chebyshev distance to the segment's bounding box: never more than the distance to the segment
        const float bx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(pa.x, pb.x) - ox, ox - __builtin_fmaxf(pa.x, pb.x)), 0.F);
        const float by = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(pa.y, pb.y) - oy, oy - __builtin_fmaxf(pa.y, pb.y)), 0.F);
        near           = __builtin_fminf(near, __builtin_fmaxf(bx, by));
    }
    near = okGroupMin(near, G);
    // (an origin outside the grid box stays uncertified at least until it has covered the distance back to the box)
    const float outside = __builtin_fmaxf(__builtin_fmaxf(g.x0 - ox, ox - g.x1), __builtin_fmaxf(g.y0 - oy, oy - g.y1));
    *clear              = usable ? 0.99F * near - 1.0e-3F : (inside ? 0.F : 0.99F * outside - 1.0e-3F); // (slack: rounding here and in the later comparison)
    unsigned long long b_amb = __ballot(amb), b_odd = __ballot(odd);
    if (G < 64)
    {
        const int                base = static_cast<int>(__lane_id()) & ~(G - 1);
        const unsigned long long mask = ((1ULL << G) - 1ULL) << base;
        b_amb &= mask;
        b_odd &= mask;
    }
    const uint32_t chi = ((flags & OKFB_RC_CHI) ? 1U : 0U) ^ (static_cast<uint32_t>(__popcll(b_odd)) & 1U);
    return usable && b_amb == 0ULL && chi == 1U;
}

// ---- RLRacers/Q_Learning pieces shared by the step kernels ------------------------------------------------------------

// What the fused Q-learning carries from step to step in registers: current state / action / previous track index, and the Q
// values of the current state (the table lives in HBM / Infinity Cache and only this agent's lanes ever touch its part of it, so
// the row is read once per launch; after a step it is either patched with the value just learned or replaced by the next
// state's row, which the update reads anyway).
struct OkQCarry
{
    int state, action, prev;
};
// (the row's three values travel as separate variables c0, c1, c2: as members of the struct, hipcc turns the selects over them
// into an indexed load and puts the struct on the stack)

// RaceTrack::findNearestTrackIndexBruteForce (RaceTrack.cpp:16-31) for the position (px, py), by the G lanes of a group (lane r
// of it calling; every lane receives the result).  The centre line sits in LDS, bucketed by grid cell: the lanes look at the
// 3 x 3 cells around the position first; if the best point found there is closer than the block's nearest edge, no point
// outside can beat or tie it and the result is the brute-force argmin (strict '<' per lane over ascending indices, (distance,
// index) order across lanes: "lowest index wins").  Otherwise -- a position far from the track -- the lanes stride over the
// whole centre line.
__device__ __forceinline__ int okNearestBucketed(const OkStepParams &p, const float *lds_cx, const float *lds_cy, const uint16_t *lds_cstart,
                                                 const uint16_t *lds_cidx, const float px, const float py, const int r, const int G)
{
    float best = 3.402823466e+38F;
    int   bi   = 0x7FFFFFFF;
    bool  done = false;
    if (p.cl_start != nullptr)
    {
        const OkGridGeom &g  = p.geom;
        const int         ci = static_cast<int>((px - g.x0) * g.inv_cell), cj = static_cast<int>((py - g.y0) * g.inv_cell);
        if (px >= g.x0 && py >= g.y0 && ci < g.nx && cj < g.ny)
        {
            for (int c = r; c < 9; c += G)
            {
                const int ix = ci + (c % 3) - 1, iy = cj + (c / 3) - 1;
                if (ix < 0 || iy < 0 || ix >= g.nx || iy >= g.ny)
                    continue;
                const int cell = iy * g.nx + ix;
                for (int k = lds_cstart[cell]; k < lds_cstart[cell + 1]; ++k)
                {
                    const int   i  = lds_cidx[k];
                    const float dx = px - lds_cx[i], dy = py - lds_cy[i];
                    const float d2 = dx * dx + dy * dy;
                    if (d2 < best)
                    {
                        best = d2;
                        bi   = i;
                    }
                }
            }
            for (int off = 1; off < G; off <<= 1)
            {
                const float ob = __shfl_xor(best, off, 64);
                const int   oi = __shfl_xor(bi, off, 64);
                if (ob < best || (ob == best && oi < bi))
                {
                    best = ob;
                    bi   = oi;
                }
            }
            // distance from the position to the nearest edge of the 3 x 3 block (>= one cell), minus slack for the rounding of
            // the cell arithmetic and of d2
            const float ex = fminf(px - (g.x0 + static_cast<float>(ci - 1) * g.cell), (g.x0 + static_cast<float>(ci + 2) * g.cell) - px);
            const float ey = fminf(py - (g.y0 + static_cast<float>(cj - 1) * g.cell), (g.y0 + static_cast<float>(cj + 2) * g.cell) - py);
            const float m  = fminf(ex, ey) - 0.05F;
            done           = bi != 0x7FFFFFFF && m > 0.F && best < m * m;
        }
    }
    if (!done)
    {
        best = 3.402823466e+38F;
        bi   = 0x7FFFFFFF;
        for (int i = r; i < p.P; i += G)
        {
            const float dx = px - lds_cx[i], dy = py - lds_cy[i];
            const float d2 = dx * dx + dy * dy;
            if (d2 < best)
            {
                best = d2;
                bi   = i;
            }
        }
        for (int off = 1; off < G; off <<= 1)
        {
            const float ob = __shfl_xor(best, off, 64);
            const int   oi = __shfl_xor(bi, off, 64);
            if (ob < best || (ob == best && oi < bi))
            {
                best = ob;
                bi   = oi;
            }
        }
    }
    return (bi == 0x7FFFFFFF) ? 0 : bi;
}

// q_racer_sim.cpp:171-182 for one agent after Environment::step: reward from the progress along the centre line
// (QAgent.hpp:150-168), learn (QAgent.hpp:121-138) with the next state's row (n0, n1, n2, read by the caller), the carried row
// moved on.  `writer`: the one lane that stores the learned value.
__device__ __forceinline__ void okQLearnStep(const OkStepParams &p, float *q_row0, const bool writer, const bool crashed, const int next_state,
                                             const int nearest, const float n0, const float n1, const float n2, OkQCarry &qs, float &c0, float &c1,
                                             float &c2, float *mirror = nullptr) // mirror: the workgroup's copy of the agent's table (tail kernel)
{
    int         prev   = qs.prev; // (a local: the address of a member would put the whole struct on the stack)
    const float reward = ok_q_reward(crashed ? 1 : 0, nearest, &prev, p.P);
    qs.prev            = prev;
    // The next state's row as learn() sees it, i.e. BEFORE this step's update.  When the agent stays in its state that row is the
    // carried one -- and must be taken from there: the value read from the table may already contain this step's update where
    // several waves hold one agent (okStepTailKernel: each wave reads for itself, one lane of one wave writes).
    const bool  same   = next_state == qs.state;
    float       mq     = same ? c0 : n0;
    const float m1     = same ? c1 : n1;
    const float m2     = same ? c2 : n2;
    mq                 = (m1 > mq) ? m1 : mq;
    mq                 = (m2 > mq) ? m2 : mq;
    float      *cell   = q_row0 + qs.state * OK_Q_ACTIONS + qs.action;
    const float old_q  = (qs.action == 0) ? c0 : ((qs.action == 1) ? c1 : c2);
    const float new_q  = ok_q_learn(old_q, mq, reward);
    if (writer)
    {
        __hip_atomic_store(cell, new_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mirror != nullptr)
            mirror[qs.state * OK_Q_ACTIONS + qs.action] = new_q;
    }
    // the row carried into the next step: the current one with the learned value, or the next state's
    c0 = (qs.action == 0) ? new_q : c0;
    c1 = (qs.action == 1) ? new_q : c1;
    c2 = (qs.action == 2) ? new_q : c2;
    if (!crashed)
    {
        if (next_state != qs.state)
        {
            c0 = n0;
            c1 = n1;
            c2 = n2;
        }
        qs.state = next_state;
    }
}

// Generic step kernel: every lane casts its own ray(s) from start to end.  Used for the global-memory and
// brute-force forms, and for fans wider than 64 rays.
template <int kMode, int kPolicy>
__global__ void __launch_bounds__(1024) okStepKernel(const OkStepParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ok_lds[];
    const OkPolyView view = okSetupView<kMode>(p, ok_lds);

    const int  G     = p.G;
    const long gl    = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int  slot  = static_cast<int>(gl / G);
    const int  rlane = static_cast<int>(gl % G);
    // episodes (see OkStepParams): the grid covers the listed agents only; a wave whose agents are all settled stops stepping
    constexpr bool kEpisodes = kPolicy != kPolicyNone;
    const bool     episode   = kEpisodes && p.settled != nullptr;
    const bool     listed    = kEpisodes && p.active != nullptr;
    // Lanes past the last agent stay in the loop (shuffles need the whole group) but never touch memory.
    const bool agent_ok = listed ? slot < p.n_active : slot < p.N;
    const int  a        = agent_ok ? (listed ? p.active[slot] : slot) : 0;
    OkAgentRegs ag      = okLoadAgent(p.st, a);
    // the policy reads the previous observation of this lane's ray (one ray per lane whenever a policy is attached)
    const bool pol_ray = agent_ok && (rlane < p.R);
    float      last_dist = (kPolicy != kPolicyNone && pol_ray) ? p.st.dist[static_cast<long>(a) * p.R + rlane] : 0.F;
    bool       settled = !agent_ok || (episode && p.settled[a] != 0);
    uint32_t   live_n  = 0U;

    for (int s = 0; s < p.n_steps; ++s)
    {
        if (episode && __ballot(!settled) == 0ULL)
            break;
        const bool was_crashed = ag.crashed;
        if (kPolicy == kPolicyMlp)
            okPolicyAction(p, a, rlane, G, ag, last_dist, pol_ray);
        float sr, cr;
        okAgentPreStep(p, ag, a, s, sr, cr);
        // ---- collision pass (CollisionChecker.cu:113-174) ------------------------------------------------
        const float ox     = ag.pos_x + p.sensor_offset * cr;
        const float oy     = ag.pos_y + p.sensor_offset * sr;
        const bool  active = !ag.crashed;
        float       min_d2 = OK_SENSOR_RANGE * OK_SENSOR_RANGE;
        for (int q = 0; q < p.rays_per_lane; ++q)
        {
            const int  r      = rlane + q * G;
            const bool ray_ok = agent_ok && (r < p.R);
            long       k      = static_cast<long>(a) * p.R + (ray_ok ? r : 0);
            asm volatile("" : "+v"(k)); // (output addresses formed here, every step, not held across the raycast of the step before)
            float      hx = ox, hy = oy;
            if (ray_ok)
            {
                if (active)
                {
                    float rdy, rdx;
                    ok_sincosf(OK_DEG2RAD * (ag.rot + p.ray_deg[r]), &rdy, &rdx);
                    const float min_t = okCastRay<kMode>(p, view, ox, oy, rdx, rdy);
                    hx                = ox + min_t * rdx;
                    hy                = oy + min_t * rdy;
                    p.st.hit_x[k]     = hx;
                    p.st.hit_y[k]     = hy;
                }
                else
                { // stale world hit point of a crashed agent
                    hx = p.st.hit_x[k];
                    hy = p.st.hit_y[k];
                }
                const float n2 = okRayEpilogue(p.st, k, hx, hy, ox, oy, sr, cr, last_dist);
                if (n2 < min_d2)
                    min_d2 = n2;
            }
        }
        min_d2 = okGroupMin(min_d2, G);
        if (min_d2 < OK_CRASH_DIST2)
            ag.crashed = true;
        if (episode)
        {
            if (was_crashed)
                settled = true;
            else
            {
                ++live_n;
                if (ag.crashed && agent_ok && rlane == 0)
                {
                    const int ae      = okOpaque(a);
                    p.crash_step[ae]  = p.ep_step0 + static_cast<uint32_t>(s) + 1U;
                    p.crash_thr[ae]   = ag.thr;
                    p.crash_steer[ae] = ag.steer;
                }
            }
        }
    }
    if (agent_ok && rlane == 0)
        okStoreAgent(p.st, a, ag);
    if (episode && agent_ok && rlane == 0)
    {
        p.settled[okOpaque(a)] = settled ? 1 : 0;
        if (live_n != 0U)
            atomicAdd(p.live, static_cast<unsigned long long>(live_n));
    }
    okFinishLaunch(p);
}

// Cooperative step kernel for the LDS form (one ray per lane).
//
// A fan's rays differ in length by an order of magnitude (median first hit ~25 px, sensor range 200 px), and a
// wave costs as much as its longest ray, so with "each lane walks its own ray to the end" three quarters of the
// lane-cycles idle.  Here the collision pass of a step runs in two phases, both inside the wave:
//   phase 1  every lane walks its own ray over [0, T1] only (a few cells).  Most rays end there.
//   phase 2  the wave's unfinished rays (a ballot) are each cut into m equal parameter intervals,
//            m = min(kMaxSplit, 64 / unfinished); lane L takes interval L % m of pending ray L / m -- origin, direction
//            and progress come from the owner lane by ds_bpermute -- walks just that interval, the m lanes of a ray
//            min-combine by shuffles and the owner pulls the result back.  The intervals beyond a ray's true first
//            hit are speculative work done by lanes that would otherwise idle; the critical path of a step drops
//            from ~20 cells to ~4 + ~3.
// Exactness: ok_cast_poly_interval's contract (ok_raycast.h) -- the min over the intervals' results carries the same
// bits as a single walk.  No workgroup barrier and no LDS traffic besides the track image: waves of a workgroup drift
// apart freely, which is what hides their stalls (an earlier version compacted the unfinished rays of the whole
// workgroup through LDS; its two barriers per step idled a quarter of the wave time).
//
// LDS: [ image | Q-learning only: centre line, 8 B per point ]
#if !defined(OKENV_MAX_SPLIT)
#define OKENV_MAX_SPLIT 8
#endif
// Wave issue priority (s_setprio; a scheduling hint, never visible in results): 0 none, 1 by the amount of phase-2 work,
// 2 by how far the wave lags behind the leading wave of its SIMD (measured on C2: 16.2 / 15.6 / 13.3 us per step)
#if !defined(OKENV_PRIO)
#define OKENV_PRIO 2
#endif
#if !defined(OKENV_PRIO_N1)
#define OKENV_PRIO_N1 8
#define OKENV_PRIO_N2 16
#endif
#if !defined(OKENV_PRIO_STEP)
#define OKENV_PRIO_STEP 1
#endif
constexpr int kMaxSplit = OKENV_MAX_SPLIT; // intervals a pending ray is cut into at most

template <int kPolicy, bool kPacked = false, bool kResident = false, bool kDirect = false, int kG = 0, bool kDriver = false>
__global__ void __launch_bounds__(1024) okStepCoopKernel(const OkStepParams p_in, const uint32_t off_coop, const float phase1_range)
{
    // kDriver: the bench driver's launch (Philox actions + reset of crashed agents, kinematics on, no device-side resetAgent): the
    // three launch-time switches become constants
    OkStepParams p = p_in;
    if (kDriver)
    {
        p.action_source = kActionsPhiloxReset;
        p.do_move       = 1;
        p.reset_flags   = 0U;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char ok_lds[];
    // Q-learning scans the centre line every step (reward = progress along it): keep it in LDS, behind the image
    // LDS behind the image: four progress words (one per SIMD, see OKENV_PRIO below), then the Q-learning data
    uint32_t *lds_progress = reinterpret_cast<uint32_t *>(ok_lds + off_coop);
    float    *lds_cx     = reinterpret_cast<float *>(ok_lds + off_coop + 16);
    float    *lds_cy     = lds_cx + p.P;
    uint16_t *lds_cstart = reinterpret_cast<uint16_t *>(lds_cy + p.P);
    uint16_t *lds_cidx   = lds_cstart + (p.geom.nx * p.geom.ny + 1);
    if (kPolicy == kPolicyQ || kPolicy == kPolicyCtrl)
    {
        for (int i = threadIdx.x; i < p.P; i += blockDim.x)
        {
            lds_cx[i] = p.cx[i];
            lds_cy[i] = p.cy[i];
        }
        if (p.cl_start != nullptr)
        {
            for (int i = threadIdx.x; i <= p.geom.nx * p.geom.ny; i += blockDim.x)
                lds_cstart[i] = p.cl_start[i];
            for (int i = threadIdx.x; i < p.P; i += blockDim.x)
                lds_cidx[i] = p.cl_idx[i];
        }
    }
    if (threadIdx.x < 4)
        lds_progress[threadIdx.x] = 0U;

    const int G = kG > 0 ? kG : p.G; // (kG: the group width as a compile-time constant)
    // lane -> (agent, ray): densely over the grid, or -- tiny populations -- a few agents in the first lanes of every workgroup
    const long gl       = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int  in_block = static_cast<int>(threadIdx.x) / G;
    const int  slot     = p.agents_per_block > 0 ? static_cast<int>(blockIdx.x) * p.agents_per_block + in_block : static_cast<int>(gl / G);
    const int  r        = static_cast<int>(gl % G); // blockDim.x is a multiple of G
    // episodes (policy kernels): the grid covers the agents that are still worth stepping, slot i holds agent active[i]
    constexpr bool kEpisodes = kPolicy != kPolicyNone;
    const bool     episode   = kEpisodes && p.settled != nullptr;
    const bool     listed    = kEpisodes && p.active != nullptr;
    const bool agent_ok = listed ? slot < p.n_active : (slot < p.N && (p.agents_per_block <= 0 || in_block < p.agents_per_block));
    const int  a        = agent_ok ? (listed ? p.active[slot] : slot) : 0;
    // Which ray a lane works on.  Normally lane r of the group has ray r.  A group with spare lanes and no phase 1 (policy-free
    // kernels) is dealt out directly instead: ray q gets the `dm` consecutive lanes q * dm .. q * dm + dm - 1, lane j of them walks
    // the j-th of dm intervals of the ray and the first of them (j == 0) does the ray's epilogue -- the same cut as phase 2 makes,
    // without its ballot / permute / shuffles, because here it is the same for every step.
    // (kDirect is a template parameter and not a launch-time test because the headline instantiation has no register to spare:
    // the host picks it when phase1_range == 0, G >= 2 R and no policy is attached)
    constexpr bool direct = kDirect;
    int        dm = 1, ray = r, part = 0;
    if (direct)
    {
        dm   = G / p.R;
        dm   = dm > kMaxSplit ? kMaxSplit : dm;
        ray  = r / dm;
        part = r - ray * dm;
    }
    const bool ray_ok   = agent_ok && (ray < p.R);    // this lane walks (a part of) ray `ray`
    const bool out_ok   = ray_ok && part == 0;       // ... and is the one that finishes it (hit point, transform, outputs)
    const long k        = static_cast<long>(a) * p.R + (ray_ok ? ray : 0);
    // the agent's state is asked for before the image is staged, so that its round trip (to the host's memory over PCIe in the
    // packed exchange: 1.5 us) runs under the staging instead of after it
    const float ray_deg = p.ray_deg[ray_ok ? ray : 0];
    OkAgentRegs        ag{};
    okenv_agent_record rc_in{};
    if (!kResident)
    {
        ag = okLoadAgent(p.st, a);
        if (kPacked)
            rc_in = p.rec_in[a];
    }
    // CMA-ES racers: the group's controller parameters, read by every step of the launch, are staged behind the centre line when
    // the workgroup's groups fit there (the host decides: ctrl_lds_off): an LDS round trip per batch of weights instead of an L2 one
    if (kPolicy == kPolicyCtrl && p.ctrl_lds_off != 0U)
    {
        float       *s_ctrl = reinterpret_cast<float *>(ok_lds + p.ctrl_lds_off) + static_cast<size_t>(in_block) * p.ctrl_num_params;
        const float *src    = p.ctrl_params + static_cast<size_t>(a) * p.ctrl_num_params;
        for (int i = r; i < p.ctrl_num_params; i += G)
            s_ctrl[i] = src[i];
    }
    const OkPolyView view = okSetupView<kGridLds>(p, ok_lds); // ends with a barrier
    // front / back split (ok_grid.h): `view` is the front image then; the back image behind it is walked only by rays whose
    // origin is not certified to lie where back segments cannot come first, or whose front walk may have missed a crossing.
    // (the host decides per launch: p.fb)
    constexpr bool kFb = true;
    const bool     fb  = p.fb != 0U;
    constexpr bool kAmbW = kFb; // the front image's walks report candidates a crossing may hide behind (ok_first_hit_update)
    OkPolyView     view_back = view;
    if (fb)
    {
        view_back.slots    = reinterpret_cast<const OkPoint *>(ok_lds + p.fb_back_off);
        view_back.hdr      = reinterpret_cast<const OkCellHdr *>(ok_lds + p.fb_back_off_hdr);
        view_back.side_tol = p.fb_back_side_tol;
    }
#if OKENV_PRIO == 2
    // which of the CU's four SIMDs this wave runs on (HW_REG_HW_ID bits 5:4): waves of one SIMD compete for its issue slots
    const uint32_t my_simd = (__builtin_amdgcn_s_getreg((2 - 1) << 11 | 4 << 6 | 4)) & 3U;
#endif
    // Resident form (host side: startResident in okenv_capi.hip): one agent per workgroup and per wave (G == 64); the waves that
    // only helped with the staging leave, the agent's wave serves one step per pass of the loop below until it is told to go or
    // has waited idle_ticks for nothing.
    if (kResident && __ballot(agent_ok) == 0ULL)
        return;
    uint32_t res_seq = p.done_seq; // the sequence number this wave waits for
    for (;;) // one pass per step the resident form serves; every other form leaves at the bottom of the first
    {
        if (kResident)
        {
            // The slot is one 64-byte line of the host's memory, fetched by ONE load of 16 lanes.  Words 3, 7, 11, 15 hold the
            // sequence number, written by the host after the record words: a 16-byte piece of the line that shows the new number
            // was read after its record words were written, whatever the pieces a PCIe read may be served in.
            const int                lane = static_cast<int>(threadIdx.x) & 63;
            const uint32_t          *slot = p.slots + static_cast<size_t>(a) * 16U;
            const unsigned long long t0   = __builtin_amdgcn_s_memrealtime();
            uint32_t                 w    = 0U;
            bool                     go   = false;
            for (;;)
            {
                w = (lane < 16) ? __hip_atomic_load(slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0U;
                const uint32_t s0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w), 3));
                const uint32_t s1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w), 7));
                const uint32_t s2 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w), 11));
                const uint32_t s3 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w), 15));
                if (s0 == s1 && s1 == s2 && s2 == s3 && (s0 == res_seq || s0 == 0xFFFFFFFFU))
                {
                    go = s0 == res_seq;
                    break;
                }
                if (__builtin_amdgcn_s_memrealtime() - t0 > static_cast<unsigned long long>(p.idle_ticks))
                    break; // nobody has asked for a step for a long time (or ever will: the process may be gone)
                __builtin_amdgcn_s_sleep(2);
            }
            if (!go)
                break;
            uint32_t rw[sizeof(okenv_agent_record) / 4U];
#pragma unroll
            for (unsigned j = 0; j < sizeof(okenv_agent_record) / 4U; ++j)
                rw[j] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w), j + j / 3U));
            __builtin_memcpy(&rc_in, rw, sizeof(okenv_agent_record));
        }
        if (kPacked)
        { // the caller's Agent objects, as records in mapped host memory
            const okenv_agent_record &rc = rc_in;
            ag.pos_x                    = rc.pos_x;
            ag.pos_y                    = rc.pos_y;
            ag.rot                      = rc.rot;
            ag.speed                    = rc.speed;
            ag.acc                      = rc.acc;
            ag.thr                      = rc.throttle;
            ag.steer                    = rc.steer;
            ag.mode                     = rc.mode;
            ag.crashed                  = rc.crashed != 0;
            ag.timed_out                = rc.timed_out != 0;
            if (p.rec_with_stats)
            {
                ag.disp_x   = rc.disp_x;
                ag.disp_y   = rc.disp_y;
                ag.disp_ctr = rc.disp_ctr;
                ag.disp_to  = rc.disp_timed_out != 0;
            }
        }
        float       last_rel_x = 0.F, last_rel_y = 0.F; // kPacked: sensor_hits_ of the last step
        float       last_dist = (kPolicy != kPolicyNone && ray_ok) ? p.st.dist[k] : 0.F;
        OkQCarry    qs{};
        float       qc0 = 0.F, qc1 = 0.F, qc2 = 0.F; // Q values of the agent's current state
        float      *q_row0 = nullptr; // this agent's table
        if (kPolicy == kPolicyQ)
        {
            qs.state  = p.q_state[a];
            qs.action = p.q_action[a];
            qs.prev   = p.q_prev_idx[a];
            q_row0    = p.q_table + static_cast<size_t>(a) * (OK_Q_STATES * OK_Q_ACTIONS);
        }
        OkTrackerRegs trk{};
        if (kPolicy == kPolicyCtrl)
        { // the bookkeeping's state travels in registers for the launch (every lane of the group holds a copy, lane 0 stores it)
            trk.prev_idx     = p.trk.prev_idx[a];
            trk.fitness      = p.trk.fitness[a];
            trk.reward       = p.trk.reward[a];
            trk.ep_return    = p.trk.ep_return[a];
            trk.ep_steps     = p.trk.ep_steps[a];
            trk.prev_crashed = p.trk.prev_crashed[a] != 0;
        }
        // Q values of the agent's current state.  The table (47.8 MB at C5) lives in HBM / Infinity Cache and only this
        // group ever touches this agent's part of it, so the row is read once per launch and then carried in registers:
        // after a step it is either patched with the value just learned or replaced by the next state's row, which the
        // update needs anyway.  One memory round trip per step instead of three; loads and stores stay agent-scope because
        // lane 0 of the group writes what the others have read.
        if (kPolicy == kPolicyQ)
        {
            const float *row = q_row0 + qs.state * OK_Q_ACTIONS;
            qc0              = __hip_atomic_load(row + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            qc1              = __hip_atomic_load(row + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            qc2              = __hip_atomic_load(row + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }

#if defined(OKENV_STAMPS)
        unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long wprof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // walk-internal stamps: [0..4] phase 1, [5..9] phase 2
#define OK_STAMP(i)                                                                                                    \
        do                                                                                                                 \
        {                                                                                                                  \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                  \
            acc[i] += now_ - last_;                                                                                        \
            last_ = now_;                                                                                                  \
        } while (0)
#define OK_WPROF(o) , wprof + (o)
        unsigned long long last_ = __builtin_amdgcn_s_memtime();
        acc[2]                   = __builtin_amdgcn_s_memrealtime(); // wave start / end on the chip-wide 100 MHz clock
#else
#define OK_STAMP(i)
#define OK_WPROF(o) , nullptr
#endif
        ok_random_action ra_blk{}; // bench driver: this lane's share of the current block of drawn actions
        // waves without a single agent (small populations get workgroups of at least 256 lanes so that the image is staged
        // quickly) have nothing to step
        const int n_steps = (__ballot(agent_ok) != 0ULL) ? p.n_steps : 0;
        // BASELINE configs 3 / 4 (32-ray fan, 32-lane groups): the lane's w1 column travels from the end of a step to the next policy
        // (the host launches the kG == 32 policy instantiation for 32-ray fans only, so the condition is a compile-time one and the
        // column is dead across the raycast)
        constexpr bool wide_mlp = kPolicy == kPolicyMlp && kG == 32;
        OkMlpColumn<32> mlp_col;
        if (wide_mlp)
            mlp_col = okMlpFetchColumn<32>(p, a, r);
        // front / back split: the last origin test of this lane's agent (position, cleared radius, answer)
        float cert_ox = 0.F, cert_oy = 0.F, cert_clear = 0.F;
        bool  cert_state = false;
        bool      settled = !agent_ok || (episode && p.settled[a] != 0); // episodes: nothing left to do for this lane's agent
        uint32_t  live_n  = 0U;                                          // steps this agent entered alive
        for (int s = 0; s < n_steps; ++s)
        {
            // episodes: a wave whose agents are all settled is done with this launch (and with the episode)
            if (episode && __ballot(!settled) == 0ULL)
                break;
            const bool was_crashed = ag.crashed;
            // Q-learning inside an episode: a crashed agent's action draws and table updates are made by okQSettleKernel
            const bool q_frozen = kPolicy == kPolicyQ && episode && was_crashed;
#if OKENV_PRIO == 2
            // A launch ends with its slowest wave, and a wave is slow for many steps in a row (its agent sits where rays are long).
            // Waves of one SIMD share its issue slots: the further a wave lags behind the leader of its SIMD, the higher its
            // issue priority.  Purely a scheduling hint: no effect on results.
            {
                uint32_t leader = 0U;
                if ((threadIdx.x & 63U) == 0U)
                    leader = atomicMax(&lds_progress[my_simd], static_cast<uint32_t>(s));
                leader          = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(leader)));
                const int behind = static_cast<int>(leader) - s;
                if (behind >= 3 * OKENV_PRIO_STEP)
                    __builtin_amdgcn_s_setprio(3);
                else if (behind >= 2 * OKENV_PRIO_STEP)
                    __builtin_amdgcn_s_setprio(2);
                else if (behind >= OKENV_PRIO_STEP)
                    __builtin_amdgcn_s_setprio(1);
                else
                    __builtin_amdgcn_s_setprio(0);
            }
#endif
            if (kPolicy == kPolicyMlp)
            {
                if (wide_mlp)
                    okMlpActionWide<32>(p, a, r, G, ag, last_dist, ray_ok, mlp_col);
                else
                    okPolicyAction(p, a, r, G, ag, last_dist, ray_ok);
            }
            if (kPolicy == kPolicyCtrl)
            {
                if (p.ctrl_lds_off != 0U) // (LDS: short round trips, small batches; the address is formed here, every step)
                {
                    const float *prm = reinterpret_cast<const float *>(ok_lds + p.ctrl_lds_off) + static_cast<size_t>(okOpaque(in_block)) * p.ctrl_num_params;
                    if (p.ctrl_hidden <= G)
                        okCtrlAction<8, 1>(p, prm, r, G, ag, last_dist);
                    else
                        okCtrlAction<2, kCtrlUnitsPerLane>(p, prm, r, G, ag, last_dist);
                }
                else
                    okCtrlAction<4, kCtrlUnitsPerLane>(p, p.ctrl_params + static_cast<size_t>(okOpaque(a)) * p.ctrl_num_params, r, G, ag, last_dist);
            }
            if (kPolicy == kPolicyQ && !q_frozen)
            { // QLearnAgent::updateAction (QAgent.hpp:98-119) from the carried row of the current state
                qs.action = ok_q_choose_action(p.seed, p.agent_base + static_cast<uint32_t>(a), p.step_base + static_cast<uint32_t>(s), p.q_epsilon,
                                               qc0, qc1, qc2);
                ok_q_action_values(qs.action, &ag.thr, &ag.steer);
            }
            OK_STAMP(0);
            // Bench driver (kActionsPhiloxReset): an action depends on (seed, agent, step) only, so lane r of the agent's group
            // draws the action of step s_blk + r and one Philox evaluation serves G steps; each step then fetches its own
            // (same draws, same bits as one evaluation per step).
            ok_random_action ra_now{};
            bool             have_drawn = false;
            if (kPolicy == kPolicyNone && p.action_source == kActionsPhiloxReset)
            {
                const int idx = s & (G - 1);
                if (idx == 0)
                    ra_blk = ok_draw_random_action(p.seed, p.agent_base + static_cast<uint32_t>(a), p.step_base + static_cast<uint32_t>(s + r));
                if (G == 64)
                { // one agent per wave: the source lane is wave-uniform
                    ra_now.throttle   = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ra_blk.throttle), idx));
                    ra_now.steer      = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ra_blk.steer), idx));
                    ra_now.reset_word = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ra_blk.reset_word), idx));
                }
                else
                {
                    ra_now.throttle   = __shfl(ra_blk.throttle, idx, G);
                    ra_now.steer      = __shfl(ra_blk.steer, idx, G);
                    ra_now.reset_word = static_cast<uint32_t>(__shfl(static_cast<int>(ra_blk.reset_word), idx, G));
                }
                have_drawn = true;
            }
            float sr, cr;
            float rdx = 1.F, rdy = 0.F;
            // (direct dealing with an even number of lanes per ray: lanes 2i and 2i + 1 hold the same ray and share the two fp64
            // evaluations, as in the tail kernel)
            okAgentPreStep(p, ag, a, s, sr, cr, ray_deg, &rdy, &rdx, have_drawn, ra_now, direct && (dm & 1) == 0);
            const float ox     = ag.pos_x + p.sensor_offset * cr;
            const float oy     = ag.pos_y + p.sensor_offset * sr;
            const bool  casts  = ray_ok && !ag.crashed;
            // front / back split: is the agent's ray origin certainly on the side of the inner boundaries where no back segment can
            // be a ray's first hit?  (one answer per agent; its rays' front walks report ambiguous rejections in amb_ray)
            bool cert = false, amb_ray = false;
            if (fb)
            { // (the answer of the last test stands while the origin stays inside the circle that test cleared; a re-placed agent
              // is far outside it, a NaN pose fails the comparison)
                const float moved = __builtin_fabsf(ox - cert_ox) + __builtin_fabsf(oy - cert_oy);
                if (!(moved < cert_clear))
                {
                    cert_state = okOriginChiGroup(view, ox, oy, r, G, p.fb_t12, p.fb_t34, &cert_clear);
                    cert_ox    = ox;
                    cert_oy    = oy;
                }
                cert = cert_state;
            }
            OK_STAMP(1);
            float min_t = OK_SENSOR_RANGE; // the ray's first-hit parameter
            // ---- phase 1: own ray over [0, T1] ----------------------------------------------------------
            bool  unfinished = false;
            float t_reached  = 0.F;
            if (direct)
            { // the lane's own interval of its ray; the intervals tile [0, inf) whatever dt rounds to (last one open-ended)
                float       found = OK_SENSOR_RANGE;
                bool        amb1  = false;
                const float dt    = OK_SENSOR_RANGE * okRcpApprox(static_cast<float>(dm));
                const float ta    = static_cast<float>(part) * dt;
                const float tb    = (part + 1 == dm) ? OKRC_INF : static_cast<float>(part + 1) * dt;
                if (casts)
                {
                    // (lanes after the first leave the cell they start in to their neighbour when the ray entered it before ta)
                    const OkIntervalResult rd = ok_cast_poly_interval<false, kAmbW>(view, ox, oy, rdx, rdy, ta, tb, nullptr, nullptr, nullptr OK_WPROF(5), part > 0);
                    found = rd.min_t;
                    amb1  = rd.amb;
                }
                // min over the dm lanes of a ray (valid on the ray's first lane at least, which is the one that uses it)
                auto ray_min = [&](float v) {
                    if ((dm & (dm - 1)) == 0)
                        return okGroupMin(v, dm); // aligned groups of a power of two: DPP
                    for (int off = 1; off < dm; off <<= 1)
                    {
                        const float other = __shfl_down(v, off, 64);
                        if (part + off < dm && other < v)
                            v = other;
                    }
                    return v;
                };
                found = ray_min(found);
                if (fb)
                { // the back image for the rays that need it, cut like the front walk, from the front's first hit down
                    const int                first     = static_cast<int>(__lane_id()) - part;
                    const unsigned long long b_amb     = __ballot(amb1);
                    const bool               need_back = casts && (!cert || ((b_amb >> first) & ((1ULL << dm) - 1ULL)) != 0ULL);
                    if (__ballot(need_back) != 0ULL)
                    {
                        const float front_t = __shfl(found, first, 64);
                        float       back_t  = front_t;
                        if (need_back && ta <= front_t)
                            back_t = ok_cast_poly_interval<false>(view_back, ox, oy, rdx, rdy, ta, tb, nullptr, nullptr, nullptr, nullptr, part > 0, front_t).min_t;
                        found = ray_min(back_t);
                    }
                }
                min_t = found;
            }
            else if (casts && phase1_range > 0.F)
            {
                const OkIntervalResult r1 =
                    ok_cast_poly_interval<false, kAmbW>(view, ox, oy, rdx, rdy, 0.F, phase1_range, nullptr, nullptr, nullptr OK_WPROF(0));
                min_t      = r1.min_t;
                unfinished = !r1.conclusive;
                t_reached  = r1.t_reached;
                amb_ray    = kFb && r1.amb;
            }
            else
                unfinished = casts; // no phase 1 (spare lanes, but a policy that wants ray r on lane r): phase 2 cuts the whole ray
            OK_STAMP(3);
            // ---- phase 2: the wave's unfinished rays, cut into m intervals each, over the wave's 64 lanes ------
            // One pass of that dealing: the rays of the lanes with `want` set are cut into m = min(8, 64 / their number) intervals of
            // [t_from, t_limit] (the last one open-ended) over image `vw`, lane L walks interval L % m of pending ray L / m, the m
            // results are min-combined and pulled back by the owner into min_t.  Used twice: for the rays phase 1 left unfinished
            // (front image -- the only image without the front / back split), and for the rays that also need the back image (from
            // the origin to the front image's first hit, which bounds every walk).
            auto coop_pass = [&](auto amb_tag, const bool want, const OkPolyView &vw, const float t_from, const float t_limit, const bool bounded) {
                constexpr bool           kA      = decltype(amb_tag)::value;
                const unsigned long long pending = __ballot(want);
                if (pending == 0ULL)
                    return;
                const int lane = static_cast<int>(__lane_id());
                const int n    = __popcll(pending);
#if OKENV_PRIO == 1 // waves with much phase-2 work get issue priority
                if (n > OKENV_PRIO_N2)
                    __builtin_amdgcn_s_setprio(2);
                else if (n > OKENV_PRIO_N1)
                    __builtin_amdgcn_s_setprio(1);
#endif
                int       m    = 64 / n;
                m              = m > kMaxSplit ? kMaxSplit : m;
                // rank of a pending lane among the pending ones; rank -> lane through a forward permute
                const int rank  = static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(pending >> 32),
                                                                             __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(pending), 0U)));
                // (pending lanes go to slots 0..n-1 in lane order, the others fill n..63: a bijection, so no two lanes
                // write the same slot)
                const int owner_of_rank = __builtin_amdgcn_ds_permute((want ? rank : n + (lane - rank)) << 2, lane);
                // task of this lane: interval j of pending ray q
                // (lane + 0.5) / m is never within 1/16 of an integer, so the approximate reciprocal cannot misplace the floor
                const float inv_m = okRcpApprox(static_cast<float>(m));
                const int   q     = static_cast<int>((static_cast<float>(lane) + 0.5F) * inv_m);
                const int  j      = lane - q * m;
                const bool has    = q < n;
                const int  owner  = __shfl(owner_of_rank, has ? q : 0, 64);
                // (one agent per wave: every lane already holds the one origin)
                const float tox   = kG == 64 ? ox : __shfl(ox, owner, 64);
                const float toy   = kG == 64 ? oy : __shfl(oy, owner, 64);
                const float tdx   = __shfl(rdx, owner, 64);
                const float tdy   = __shfl(rdy, owner, 64);
                const float t0    = __shfl(t_from, owner, 64);
                const float tl    = bounded ? __shfl(t_limit, owner, 64) : OK_SENSOR_RANGE;
                float       found = OK_SENSOR_RANGE;
                bool        amb2  = false;
                if (has)
                {
                    // neighbouring lanes evaluate the shared bound with the same expression, the last interval is open-ended:
                    // the intervals tile [t0, inf) whatever dt rounds to
                    const float dt = (tl - t0) * inv_m;
                    const float ta = t0 + static_cast<float>(j) * dt;
                    const float tb = (j + 1 == m) ? OKRC_INF : t0 + static_cast<float>(j + 1) * dt;
                    // Every cell the ray entered before ta has been processed -- by phase 1 (t0 > 0) or by the lane of the interval before
                    // this one -- so the walk steps over its start cell unless the ray enters it inside [ta, tb): each cell of a pending
                    // ray is looked at once, not twice (3.3 -> 2.x cell iterations per wave-step in lock step).
                    const OkIntervalResult r2 =
                        ok_cast_poly_interval<false, kA>(vw, tox, toy, tdx, tdy, ta, tb, nullptr, nullptr, nullptr OK_WPROF(5), j > 0 || t0 > 0.F, tl);
                    found = r2.min_t;
                    amb2  = kA && r2.amb;
                }
                // min over the m lanes of a ray (consecutive lanes), then back to the owner
                const int first = want ? rank * m : 0;
                float     mine  = OK_SENSOR_RANGE;
                // (the owner gathering all m results in one round trip measured slower than this shuffle tree plus one pull)
#pragma unroll
                for (int off = 1; off < kMaxSplit; off <<= 1)
                {
                    const float other = __shfl_down(found, off, 64);
                    if (j + off < m && other < found)
                        found = other;
                }
                mine = __shfl(found, first, 64);
                if (want && mine < min_t)
                    min_t = mine;
                if (kA)
                { // an ambiguous rejection in any of the ray's m intervals (lanes first .. first + m - 1) is the ray's
                    const unsigned long long b_amb = __ballot(amb2);
                    if (want && ((b_amb >> first) & ((1ULL << m) - 1ULL)) != 0ULL)
                        amb_ray = true;
                }
#if OKENV_PRIO == 1
                __builtin_amdgcn_s_setprio(0);
#endif
            };
            if (!direct)
                coop_pass(std::integral_constant<bool, kAmbW>{}, unfinished, view, t_reached, OK_SENSOR_RANGE, false);
            // the back image for the rays that need it -- an origin that is not certified, a front walk that may have missed a crossing --
            // from the origin to the front image's first hit, dealt to the wave's lanes the same way (min over front and back = min over
            // all segments, whatever the origin)
            if (!direct && fb)
                coop_pass(std::false_type{}, casts && (!cert || amb_ray), view_back, 0.F, min_t, true);
            OK_STAMP(5);
            if (wide_mlp)
                mlp_col = okMlpFetchColumn<32>(p, a, r); // for the next step's policy: in flight during the epilogue

            // ---- hit point, transform, crash test (CollisionChecker.cu:69-70,144-172) -----------------------------------
            float min_d2 = OK_SENSOR_RANGE * OK_SENSOR_RANGE;
            if (out_ok)
            {
                // (an index the compiler cannot see through: the five output addresses are formed here, every step, instead of
                // being held in ten VGPRs across the raycast)
                long ke = k;
                asm volatile("" : "+v"(ke));
                float hx, hy;
                if (casts)
                {
                    hx                = ox + min_t * rdx;
                    hy                = oy + min_t * rdy;
                    p.st.hit_x[ke]    = hx;
                    p.st.hit_y[ke]    = hy;
                }
                else
                { // stale world hit point of a crashed agent
                    hx = p.st.hit_x[ke];
                    hy = p.st.hit_y[ke];
                }
                min_d2 = okRayEpilogue(p.st, ke, hx, hy, ox, oy, sr, cr, last_dist, kPacked ? &last_rel_x : nullptr, kPacked ? &last_rel_y : nullptr);
                min_d2 = (min_d2 < OK_SENSOR_RANGE * OK_SENSOR_RANGE) ? min_d2 : OK_SENSOR_RANGE * OK_SENSOR_RANGE;
            }
            min_d2 = okGroupMin(min_d2, G);
            if (min_d2 < OK_CRASH_DIST2)
                ag.crashed = true;
            if (kPolicy == kPolicyQ)
            { // q_racer_sim.cpp:171-182: next state, reward from the progress along the centre line, table update
                int next_state = 0, mult = 1;
#pragma unroll
                for (int i = 0; i < 5; ++i)
                {
                    next_state += ok_q_bin(__shfl(last_dist, p.q_ray[i], G)) * mult;
                    mult *= 3;
                }
                // The next state's row is asked for NOW: its round trip (the tables are 47.8 MB at C5: Infinity Cache or HBM) runs under
                // the nearest-index search below instead of after it.
                float n0 = 0.F, n1 = 0.F, n2 = 0.F;
                if (!q_frozen)
                {
                    const float *nrow = q_row0 + next_state * OK_Q_ACTIONS;
                    n0                = __hip_atomic_load(nrow + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    n1                = __hip_atomic_load(nrow + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    n2                = __hip_atomic_load(nrow + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const int nearest = okNearestBucketed(p, lds_cx, lds_cy, lds_cstart, lds_cidx, ag.pos_x, ag.pos_y, r, G);
                if (!q_frozen)
                {
                    okQLearnStep(p, q_row0, agent_ok && r == 0, ag.crashed, next_state, nearest, n0, n1, n2, qs, qc0, qc1, qc2);
                    if (ag.crashed && episode && agent_ok && r == 0)
                        p.q_next_state[okOpaque(a)] = next_state; // what every later step of this agent will see as its next state
                }
            }
            if (kPolicy == kPolicyCtrl)
            { // the callers' bookkeeping after env.step() (main_eigen.cpp:143-158 / ppo_sim.cpp:77-80)
                int idx = 0;
                if (p.trk_kind == kRewardProgress)
                    idx = okNearestBucketed(p, lds_cx, lds_cy, lds_cstart, lds_cidx, ag.pos_x, ag.pos_y, r, G);
                okTrackerStep(trk, p.trk_kind, idx, ag.crashed, ag.timed_out);
            }
            if (episode)
            {
                if (was_crashed)
                    settled = true; // this was its step as a crashed agent: nothing about it changes from here on
                else
                {
                    ++live_n;
                    if (ag.crashed && agent_ok && r == 0)
                    { // crashed in this step (wall or standstill timeout)
                        const int ae      = okOpaque(a);
                        p.crash_step[ae]  = p.ep_step0 + static_cast<uint32_t>(s) + 1U;
                        p.crash_thr[ae]   = ag.thr;
                        p.crash_steer[ae] = ag.steer;
                    }
                }
            }
            OK_STAMP(6);
        }
        if (episode && agent_ok && r == 0)
        {
            p.settled[okOpaque(a)] = settled ? 1 : 0;
            if (live_n != 0U)
                atomicAdd(p.live, static_cast<unsigned long long>(live_n));
        }
        if (kPolicy == kPolicyCtrl && agent_ok && r == 0 && n_steps > 0)
        {
            const int ae            = okOpaque(a);
            p.trk.prev_idx[ae]      = trk.prev_idx;
            p.trk.fitness[ae]       = trk.fitness;
            p.trk.reward[ae]        = trk.reward;
            p.trk.ep_return[ae]     = trk.ep_return;
            p.trk.ep_steps[ae]      = trk.ep_steps;
            p.trk.prev_crashed[ae]  = trk.prev_crashed ? 1 : 0;
        }
        if (kPolicy == kPolicyQ && agent_ok && r == 0)
        {
            const int ae     = okOpaque(a);
            p.q_state[ae]    = qs.state;
            p.q_action[ae]   = qs.action;
            p.q_prev_idx[ae] = qs.prev;
        }
#if defined(OKENV_STAMPS)
        acc[4] = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0 && p.stamps != nullptr)
        { // diagnostic build only: per-wave cycle sums
            unsigned long long *dbg = p.stamps;
            const long          w   = gl >> 6;
            for (int i = 0; i < 8; ++i)
                dbg[w * kStampWords + i] = acc[i];
        }
        if (p.stamps != nullptr)
        { // walk-internal stamps of the lane that spent the longest inside the walks (its view has the fewest gaps)
            unsigned long long *dbg = p.stamps;
            const long          w   = gl >> 6;
            unsigned long long  tot = 0;
            for (int i = 0; i < 10; ++i)
                tot += wprof[i];
            unsigned long long best = tot;
            for (int off = 1; off < 64; off <<= 1)
            {
                const unsigned long long o = __shfl_xor(best, off, 64);
                best                       = o > best ? o : best;
            }
            if (tot == best)
                for (int i = 0; i < 10; ++i)
                    dbg[w * kStampWords + 8 + i] = wprof[i];
        }
#endif
        if (agent_ok && r == 0)
            okStoreAgent(p.st, a, ag);
        if (kPacked)
        {
            if (out_ok)
            {
                p.hits_xy_out[2 * k]     = last_rel_x;
                p.hits_xy_out[2 * k + 1] = last_rel_y;
            }
            if (agent_ok && r == 0)
            {
                okenv_agent_record rc;
                rc.pos_x          = ag.pos_x;
                rc.pos_y          = ag.pos_y;
                rc.rot            = ag.rot;
                rc.speed          = ag.speed;
                rc.acc            = ag.acc;
                rc.throttle       = ag.thr;
                rc.steer          = ag.steer;
                rc.disp_x         = ag.disp_x;
                rc.disp_y         = ag.disp_y;
                rc.disp_ctr       = ag.disp_ctr;
                rc.mode           = static_cast<uint8_t>(ag.mode);
                rc.crashed        = ag.crashed ? 1 : 0;
                rc.timed_out      = ag.timed_out ? 1 : 0;
                rc.disp_timed_out = ag.disp_to ? 1 : 0;
                p.rec_out[a]      = rc;
            }
        }
        if (!kResident)
        {
            okFinishLaunch(p);
            break;
        }
        // resident: this wave is all that is left of its workgroup.  Its results have left the device before it reports in; the
        // last workgroup to do so answers the host (which asks for the next step only after that, so one counter is enough).
        __threadfence_system();
        if ((threadIdx.x & 63U) == 0U)
        {
            __threadfence();
            if (atomicAdd(&p.step_counter[1], 1U) == gridDim.x - 1U)
            {
                p.step_counter[1] = 0U;
                __threadfence_system();
                __hip_atomic_store(p.done_flag, res_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        res_seq = okNextPackedSeq(res_seq);
    }
}

// ---- the tail of an episode: few agents left, the whole machine free -----------------------------------------------------
//
// A rollout "until every agent has crashed" spends most of its steps on a handful of survivors (C3: under a tenth of the
// population after 200 of 700-1800 steps), and a step of the cooperative kernel costs one wave's dependent chain -- about
// 11 us for a 32-ray agent with the fused MLP -- however few agents there are.  This kernel shortens the chain instead of
// sharing it: ONE agent per workgroup, every ray cut into kTailSplit = 8 parameter intervals from its origin on, one interval
// per lane (32 rays: 256 lanes, four waves).  A lane walks 25 px; the eight lanes of a ray min-combine by DPP; the ray's first
// lane does its epilogue.  The waves of the workgroup meet once per step: the rays' distances (the next policy's inputs) and
// the waves' minima of the squared hit distances (the crash test) cross through LDS, double-buffered so that one barrier per
// step is enough.  Every wave evaluates the policy for itself (same inputs, same order, same bits), so nothing else has to be
// exchanged; with kR = 32 the lane's columns of both weight matrices stay in registers for the whole launch.
// Exactness: ok_cast_poly_interval's contract (the min over a ray's intervals carries the bits of a single walk) -- what the
// cooperative kernel's phase 2 and the direct dealing of small populations rely on as well.
// LDS: [ image | 16 B unused | dist[2][64] | dist / 200 [2][64] (the MLP's inputs, divided once by the ray's lane) | min[2][8] |
//        Q-learning: centre line + buckets ]
constexpr int kTailSplit     = 8;
constexpr int kTailLdsFloats = 4 * 64 + 2 * 16;
// Q-learning: one more wave per workgroup (it has no rays: while the others walk theirs it finds the nearest centre-line index of
// the new position, which depends on the move only), and the agent's whole table in LDS for the launch -- this workgroup is the
// only one that touches it, learned values are written through to memory -- so that the next state's row costs an LDS round trip
// instead of an L2 one: what is left of the step after its barrier is the table arithmetic.
constexpr int kTailQFloats   = OK_Q_STATES * OK_Q_ACTIONS + 7; // the table, nearest index [2], the next step's epsilon-greedy draw [2], padding

template <int kPolicy, int kR>
__global__ void __launch_bounds__(kPolicy == kPolicyQ ? 576 : 512) okStepTailKernel(const OkStepParams p, const uint32_t off_tail)
{
    static_assert(kPolicy != kPolicyNone, "policy kernels only");
    extern __shared__ __attribute__((aligned(16))) unsigned char ok_lds[];
    float    *s_dist     = reinterpret_cast<float *>(ok_lds + off_tail + 16);
    float    *s_xs       = s_dist + 2 * 64;
    float    *s_min      = s_xs + 2 * 64;
    float    *s_qtab     = s_min + 2 * 16;                                        // (Q-learning only, as is everything behind it)
    int      *s_near     = reinterpret_cast<int *>(s_qtab + OK_Q_STATES * OK_Q_ACTIONS);
    float    *lds_cx     = s_qtab + kTailQFloats;
    float    *lds_cy     = lds_cx + p.P;
    uint16_t *lds_cstart = reinterpret_cast<uint16_t *>(lds_cy + p.P);
    uint16_t *lds_cidx   = lds_cstart + (p.geom.nx * p.geom.ny + 1);
    if (kPolicy == kPolicyQ)
    {
        for (int i = threadIdx.x; i < p.P; i += blockDim.x)
        {
            lds_cx[i] = p.cx[i];
            lds_cy[i] = p.cy[i];
        }
        if (p.cl_start != nullptr)
        {
            for (int i = threadIdx.x; i <= p.geom.nx * p.geom.ny; i += blockDim.x)
                lds_cstart[i] = p.cl_start[i];
            for (int i = threadIdx.x; i < p.P; i += blockDim.x)
                lds_cidx[i] = p.cl_idx[i];
        }
    }
    const int  L      = static_cast<int>(threadIdx.x);
    const int  lane   = L & 63, wave = L >> 6, n_waves = static_cast<int>(blockDim.x) >> 6;
    const int  R      = kR > 0 ? kR : p.R;
    const int  ray    = L / kTailSplit, part = L % kTailSplit;
    const bool ray_ok = ray < R;
    const bool out_ok = ray_ok && part == 0;
    const bool listed = p.active != nullptr;
    const int  a      = listed ? p.active[blockIdx.x] : static_cast<int>(blockIdx.x); // (the grid is exactly the agents to step)
    const bool episode = p.settled != nullptr;
    const long k       = static_cast<long>(a) * R + (ray_ok ? ray : 0);
    const float ray_deg = p.ray_deg[ray_ok ? ray : 0];
    OkAgentRegs ag      = okLoadAgent(p.st, a);
    if (out_ok)
    { // the previous observation: the first policy's inputs
        const float d0 = p.st.dist[k];
        s_dist[ray]    = d0;
        s_xs[ray]      = d0 / 200.0F;
    }
    const OkPolyView view = okSetupView<kGridLds>(p, ok_lds); // ends with a barrier
    // front / back split (ok_grid.h), as in the cooperative kernel: `view` is the front image then, the back image is walked by the
    // rays that need it; every wave makes the agent's origin test for itself (same inputs, same answer)
    const bool fb        = p.fb != 0U;
    OkPolyView view_back = view;
    if (fb)
    {
        view_back.slots    = reinterpret_cast<const OkPoint *>(ok_lds + p.fb_back_off);
        view_back.hdr      = reinterpret_cast<const OkCellHdr *>(ok_lds + p.fb_back_off_hdr);
        view_back.side_tol = p.fb_back_side_tol;
    }
    float cert_ox = 0.F, cert_oy = 0.F, cert_clear = 0.F;
    bool  cert_state = false;

    OkQCarry qs{};
    float    qc0 = 0.F, qc1 = 0.F, qc2 = 0.F;
    float   *q_row0 = nullptr;
    if (kPolicy == kPolicyQ)
    {
        qs.state         = p.q_state[a];
        qs.action        = p.q_action[a];
        qs.prev          = p.q_prev_idx[a];
        q_row0           = p.q_table + static_cast<size_t>(a) * (OK_Q_STATES * OK_Q_ACTIONS);
        for (int i = L; i < OK_Q_STATES * OK_Q_ACTIONS; i += static_cast<int>(blockDim.x)) // (first read after the first step's barrier)
            s_qtab[i] = __hip_atomic_load(q_row0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float *row = q_row0 + qs.state * OK_Q_ACTIONS;
        qc0              = __hip_atomic_load(row + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        qc1              = __hip_atomic_load(row + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        qc2              = __hip_atomic_load(row + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the MLP's weights: hidden unit (lane & 31)'s column of w1, output (lane, clamped to 7)'s column of w2
    const float *w1 = p.mlp_w + static_cast<size_t>(a) * OK_MLP_WEIGHTS(R);
    const float *w2 = w1 + (R + 2) * OK_MLP_HID_PAD;
    const int    ul = lane & (OK_MLP_HID_PAD - 1);
    const int    kl = lane < OK_MLP_OUT_PAD ? lane : OK_MLP_OUT_PAD - 1;
    constexpr int kCols = kR > 0 ? kR + 2 : 1;
    float         wcol[kCols], vcol[OK_MLP_HID_PAD];
    if (kPolicy == kPolicyMlp)
    {
        if (kR > 0)
        {
#pragma unroll
            for (int j = 0; j < kCols; ++j)
                wcol[j] = w1[j * OK_MLP_HID_PAD + ul];
        }
#pragma unroll
        for (int i = 0; i < OK_MLP_HID_PAD; ++i)
            vcol[i] = w2[i * OK_MLP_OUT_PAD + kl];
    }

    bool     settled = episode && p.settled[a] != 0;
    uint32_t live_n  = 0U;
#if defined(OKENV_STAMPS)
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long twalk[5] = {0, 0, 0, 0, 0}; // inside the walk: set-up, cell entry, point loop, exact loop, leaving the cell
    unsigned long long tlast   = __builtin_amdgcn_s_memtime();
#define OK_TWALK twalk
    tacc[2]                    = __builtin_amdgcn_s_memrealtime();
#define OK_TSTAMP(i)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                  \
        tacc[i] += now_ - tlast;                                                                                       \
        tlast = now_;                                                                                                  \
    } while (0)
#else
#define OK_TSTAMP(i)
#define OK_TWALK nullptr
#endif
    for (int s = 0; s < p.n_steps; ++s)
    {
        if (episode && settled) // (the same in every lane of the workgroup: they all hold the one agent)
            break;
        const float *xs_in    = s_xs + (s & 1) * 64;         // written by the step before (or above)
        float       *dist_out = s_dist + ((s & 1) ^ 1) * 64;
        float       *xs_out   = s_xs + ((s & 1) ^ 1) * 64;
        float       *min_out  = s_min + ((s & 1) ^ 1) * 16;
        const bool   was_crashed = ag.crashed;
        const bool   q_frozen    = kPolicy == kPolicyQ && episode && was_crashed;
        if (kPolicy == kPolicyMlp)
        { // GeneticAgent::updateAction: the same terms in the same order as okMlpAction, inputs read from LDS instead of shuffled
            const float x0  = ag.speed / 100.0F;
            const float x1  = ok_normalize_angle_deg(ag.rot) / 360.0F;
            float       acc = 0.F;
            if (kR > 0)
            {
                acc = acc + x0 * wcol[0];
                acc = acc + x1 * wcol[1];
#pragma unroll
                for (int j = 0; j < (kR > 0 ? kR : 1); ++j)
                    acc = acc + xs_in[j] * wcol[kR > 0 ? 2 + j : 0];
            }
            else
            {
                acc = acc + x0 * w1[0 * OK_MLP_HID_PAD + ul];
                acc = acc + x1 * w1[1 * OK_MLP_HID_PAD + ul];
                for (int j = 0; j < R; ++j)
                    acc = acc + xs_in[j] * w1[(2 + j) * OK_MLP_HID_PAD + ul];
            }
            const float h = (acc > 0.F) ? acc : 0.F;
            float       z = 0.F;
            // (the whole wave holds the one agent: lane i's value by v_readlane, no LDS round trip)
#pragma unroll
            for (int i = 0; i < OK_MLP_HID_PAD; ++i)
                z = z + okFromBits(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(okBits(h)), i))) * vcol[i];
            float zs[OK_MLP_OUT];
#pragma unroll
            for (int q = 0; q < OK_MLP_OUT; ++q)
                zs[q] = okFromBits(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(okBits(z)), q)));
            ok_ga_decode_action(zs, &ag.thr, &ag.steer);
        }
        if (kPolicy == kPolicyQ && !q_frozen)
        { // the epsilon-greedy draw of this step was made by the wave without rays during the step before (it depends on the step's
          // number only); the first step of a launch makes its own
            const int drawn = (s == 0) ? ok_q_draw_action(p.seed, p.agent_base + static_cast<uint32_t>(a), p.step_base, p.q_epsilon) : s_near[2 + (s & 1)];
            qs.action       = drawn < 0 ? ok_q_argmax3(qc0, qc1, qc2) : drawn;
            ok_q_action_values(qs.action, &ag.thr, &ag.steer);
        }
        OK_TSTAMP(0);
        float sr, cr, rdx = 1.F, rdy = 0.F;
        static_assert(kTailSplit % 2 == 0, "lanes 2i and 2i + 1 share a ray");
        okAgentPreStep(p, ag, a, s, sr, cr, ray_deg, &rdy, &rdx, false, ok_random_action{}, true);
        const float ox    = ag.pos_x + p.sensor_offset * cr;
        const float oy    = ag.pos_y + p.sensor_offset * sr;
        const bool  casts = ray_ok && !ag.crashed;
        OK_TSTAMP(1);
        bool cert = false;
        if (fb)
        { // (cooperative kernel: the answer of the last origin test stands while the origin stays inside the circle it cleared)
            const float moved = __builtin_fabsf(ox - cert_ox) + __builtin_fabsf(oy - cert_oy);
            if (!(moved < cert_clear))
            {
                cert_state = okOriginChiGroup(view, ox, oy, lane, 64, p.fb_t12, p.fb_t34, &cert_clear);
                cert_ox    = ox;
                cert_oy    = oy;
            }
            cert = cert_state;
        }
        // the lane's interval of its ray; the intervals tile [0, inf) whatever dt rounds to (the last one is open-ended)
        float       found = OK_SENSOR_RANGE;
        bool        amb   = false;
        const float dt    = OK_SENSOR_RANGE * okRcpApprox(static_cast<float>(kTailSplit));
        const float ta    = static_cast<float>(part) * dt;
        const float tb    = (part + 1 == kTailSplit) ? OKRC_INF : static_cast<float>(part + 1) * dt;
        if (casts)
        {
            const OkIntervalResult rf = ok_cast_poly_interval<false, true>(view, ox, oy, rdx, rdy, ta, tb, nullptr, nullptr, nullptr, OK_TWALK, part > 0);
            found                     = rf.min_t;
            amb                       = rf.amb;
        }
        found = okGroupMin(found, kTailSplit);
        if (fb)
        { // the back image for the rays that need it, cut like the front walk; an interval beyond the front's first hit has nothing to add
            const unsigned long long b_amb     = __ballot(amb);
            const bool               amb_ray   = ((b_amb >> (lane & ~(kTailSplit - 1))) & ((1ULL << kTailSplit) - 1ULL)) != 0ULL;
            const bool               need_back = casts && (!cert || amb_ray); // (the same in the kTailSplit lanes of a ray)
            if (__ballot(need_back) != 0ULL)
            {
                float fb_found = found;
                if (need_back && ta <= found)
                    fb_found = ok_cast_poly_interval<false>(view_back, ox, oy, rdx, rdy, ta, tb, nullptr, nullptr, nullptr, nullptr, part > 0, found).min_t;
                found = okGroupMin(fb_found, kTailSplit);
            }
        }
        if (kPolicy == kPolicyQ && wave == n_waves - 1)
        { // the wave without rays: RaceTrack::findNearestTrackIndexBruteForce of the new position, while the others walk
            const int nearest_here = okNearestBucketed(p, lds_cx, lds_cy, lds_cstart, lds_cidx, ag.pos_x, ag.pos_y, lane, 64);
            if (lane == 0)
            {
                s_near[(s & 1) ^ 1]     = nearest_here;
                s_near[2 + ((s & 1) ^ 1)] = ok_q_draw_action(p.seed, p.agent_base + static_cast<uint32_t>(a), p.step_base + static_cast<uint32_t>(s) + 1U, p.q_epsilon);
            }
        }
        OK_TSTAMP(3);
        float min_d2 = OK_SENSOR_RANGE * OK_SENSOR_RANGE;
        if (out_ok)
        {
            long ke = k;
            asm volatile("" : "+v"(ke));
            float hx, hy;
            if (casts)
            {
                hx             = ox + found * rdx;
                hy             = oy + found * rdy;
                p.st.hit_x[ke] = hx;
                p.st.hit_y[ke] = hy;
            }
            else
            { // stale world hit point of a crashed agent
                hx = p.st.hit_x[ke];
                hy = p.st.hit_y[ke];
            }
            float d;
            min_d2        = okRayEpilogue(p.st, ke, hx, hy, ox, oy, sr, cr, d);
            min_d2        = (min_d2 < OK_SENSOR_RANGE * OK_SENSOR_RANGE) ? min_d2 : OK_SENSOR_RANGE * OK_SENSOR_RANGE;
            dist_out[ray] = d;
            if (kPolicy == kPolicyMlp)
                xs_out[ray] = d / 200.0F; // Network::infer's input (Network.hpp:131), formed once here instead of by every lane of every wave
        }
        min_d2 = okGroupMin(min_d2, 64);
        if (lane == 0)
            min_out[wave] = min_d2;
        OK_TSTAMP(5);
        __syncthreads(); // the one meeting of the step: distances and minima of all waves are in LDS
        OK_TSTAMP(6);
        for (int w = 0; w < n_waves; ++w)
        {
            const float o = min_out[w];
            min_d2        = (o < min_d2) ? o : min_d2;
        }
        if (min_d2 < OK_CRASH_DIST2)
            ag.crashed = true;
        if (kPolicy == kPolicyQ)
        { // q_racer_sim.cpp:171-182
            int next_state = 0, mult = 1;
#pragma unroll
            for (int i = 0; i < 5; ++i)
            {
                next_state += ok_q_bin(dist_out[p.q_ray[i]]) * mult;
                mult *= 3;
            }
            // the next state's row from the workgroup's copy of the table (a row other than the current state's: nobody writes it in
            // this step; the current state's own row is the carried one, okQLearnStep), the nearest index from the wave without rays
            const float *nrow    = s_qtab + next_state * OK_Q_ACTIONS;
            const float  n0      = nrow[0], n1 = nrow[1], n2 = nrow[2];
            const int    nearest = s_near[(s & 1) ^ 1];
            if (!q_frozen)
            {
                okQLearnStep(p, q_row0, L == 0, ag.crashed, next_state, nearest, n0, n1, n2, qs, qc0, qc1, qc2, s_qtab);
                if (ag.crashed && episode && L == 0)
                    p.q_next_state[okOpaque(a)] = next_state;
            }
        }
        OK_TSTAMP(7);
        if (episode)
        {
            if (was_crashed)
                settled = true;
            else
            {
                ++live_n;
                if (ag.crashed && L == 0)
                {
                    const int ae      = okOpaque(a);
                    p.crash_step[ae]  = p.ep_step0 + static_cast<uint32_t>(s) + 1U;
                    p.crash_thr[ae]   = ag.thr;
                    p.crash_steer[ae] = ag.steer;
                }
            }
        }
    }
#if defined(OKENV_STAMPS)
    tacc[4] = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && p.stamps != nullptr)
        for (int i = 0; i < 8; ++i)
            p.stamps[(static_cast<long>(blockIdx.x) * n_waves + wave) * kStampWords + i] = tacc[i];
    if (lane == 0 && p.stamps != nullptr)
        for (int i = 0; i < 5; ++i)
            p.stamps[(static_cast<long>(blockIdx.x) * n_waves + wave) * kStampWords + 8 + i] = twalk[i];
#endif
    if (L == 0)
    {
        okStoreAgent(p.st, a, ag);
        if (kPolicy == kPolicyQ)
        {
            const int ae     = okOpaque(a);
            p.q_state[ae]    = qs.state;
            p.q_action[ae]   = qs.action;
            p.q_prev_idx[ae] = qs.prev;
        }
        if (episode)
        {
            p.settled[okOpaque(a)] = settled ? 1 : 0;
            if (live_n != 0U)
                atomicAdd(p.live, static_cast<unsigned long long>(live_n));
        }
    }
    okFinishLaunch(p);
}

// ---- small service kernels ------------------------------------------------------------------------------

// Packed host exchange (okenv_step_packed): records <-> struct-of-arrays, one thread per agent; the pack side also
// interleaves sensor_hits_ as (x, y) pairs, one thread per ray.
__global__ void okUnpackRecordsKernel(OkDeviceState st, const okenv_agent_record *rec, int N, int with_stats)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= N)
        return;
    const okenv_agent_record r = rec[a];
    st.pos_x[a]     = r.pos_x;
    st.pos_y[a]     = r.pos_y;
    st.rot[a]       = r.rot;
    st.speed[a]     = r.speed;
    st.acc[a]       = r.acc;
    st.thr[a]       = r.throttle;
    st.steer[a]     = r.steer;
    st.mode[a]      = r.mode;
    st.crashed[a]   = r.crashed;
    st.timed_out[a] = r.timed_out;
    if (with_stats)
    {
        st.disp_x[a]   = r.disp_x;
        st.disp_y[a]   = r.disp_y;
        st.disp_ctr[a] = r.disp_ctr;
        st.disp_to[a]  = r.disp_timed_out;
    }
}

__global__ void okPackRecordsKernel(OkDeviceState st, okenv_agent_record *rec, float *hits_xy, int N, int R)
{
    const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < static_cast<long>(N) * R)
    {
        hits_xy[2 * i]     = st.rel_x[i];
        hits_xy[2 * i + 1] = st.rel_y[i];
    }
    if (i < N)
    {
        const int          a = static_cast<int>(i);
        okenv_agent_record r;
        r.pos_x          = st.pos_x[a];
        r.pos_y          = st.pos_y[a];
        r.rot            = st.rot[a];
        r.speed          = st.speed[a];
        r.acc            = st.acc[a];
        r.throttle       = st.thr[a];
        r.steer          = st.steer[a];
        r.disp_x         = st.disp_x[a];
        r.disp_y         = st.disp_y[a];
        r.disp_ctr       = st.disp_ctr[a];
        r.mode           = st.mode[a];
        r.crashed        = st.crashed[a];
        r.timed_out      = st.timed_out[a];
        r.disp_timed_out = st.disp_to[a];
        rec[a]           = r;
    }
}

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

// Environment::resetAgent (Environment.cpp:79-122) for a list of agents (idx == nullptr: agents 0..n-1); entry j of the
// call draws from Philox (agent_base + agent, epoch) and takes call-counter parity epoch + j (okenv_math.h).
__global__ void okResetRandomKernel(OkDeviceState st,
                                    const int32_t *idx,
                                    int            n,
                                    int            N,
                                    uint32_t       flags,
                                    uint32_t       seed,
                                    uint32_t       epoch,
                                    uint32_t       agent_base,
                                    const float   *cx,
                                    const float   *cy,
                                    const float   *chead,
                                    const float   *lane_l,
                                    const float   *lane_r,
                                    int            P)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    const int a = idx ? idx[j] : j;
    if (a < 0 || a >= N)
        return;
    if ((flags & OK_RESET_ONLY_DONE) != 0U && st.crashed[a] == 0)
        return;
    const ok_reset_draw d = ok_draw_reset(seed, agent_base + static_cast<uint32_t>(a), epoch, epoch + static_cast<uint32_t>(j),
                                          static_cast<uint32_t>(P), flags);
    float               x, y, rot;
    ok_reset_pose(d, cx, cy, chead, lane_l, lane_r, &x, &y, &rot);
    st.pos_x[a]     = x;
    st.pos_y[a]     = y;
    st.rot[a]       = rot;
    st.acc[a]       = 0.F;
    st.speed[a]     = 0.F;
    st.crashed[a]   = 0;
    st.timed_out[a] = 0;
    st.thr[a]       = 0.F;
    st.steer[a]     = 0.F;
}

// Agent::reset of every agent to one pose (genetic_learner_sim.cpp:65-70)
__global__ void okResetAllKernel(OkDeviceState st, float x, float y, float rot, int N)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= N)
        return;
    st.pos_x[a]     = x;
    st.pos_y[a]     = y;
    st.rot[a]       = rot;
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

// RaceTrack::findNearestTrackIndexBruteForce (RaceTrack.cpp:16-31).  kNearestLanes consecutive lanes share one query:
// lane l scans the centre-line points l, l + L, l + 2L, ... of each LDS-staged chunk, then the lanes combine
// (distance, index) pairs taking the smaller distance and, on equal distances, the lower index -- the element the
// sequential strict-'<' scan keeps.  Every thread of the workgroup must call it (barriers inside); `sx`/`sy` are two
// 1024-float LDS arrays; all lanes of a query receive the result.
constexpr int kNearestLanes = 16;

__device__ __forceinline__ int okNearestIdx(const float *cx, const float *cy, const int P, const float px, const float py, float *sx, float *sy)
{
    const int lane  = static_cast<int>(threadIdx.x) & (kNearestLanes - 1);
    float     bestv = 3.402823466e+38F; // FLT_MAX, as in the reference
    int       arg   = 0x7FFFFFFF;
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
        for (int j = lane; j < m; j += kNearestLanes)
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
    for (int off = 1; off < kNearestLanes; off <<= 1)
    {
        const float ov = __shfl_xor(bestv, off, 64);
        const int   oa = __shfl_xor(arg, off, 64);
        if (ov < bestv || (ov == bestv && oa < arg))
        {
            bestv = ov;
            arg   = oa;
        }
    }
    // no point closer than FLT_MAX (NaN or infinite coordinates): the sequential scan answers index 0
    return arg == 0x7FFFFFFF ? 0 : arg;
}

__global__ void okNearestIdxKernel(const float *cx, const float *cy, int P, const float *qx, const float *qy, int n, int32_t *out)
{
    __shared__ float sx[1024], sy[1024];
    const int        i   = (blockIdx.x * blockDim.x + threadIdx.x) / kNearestLanes;
    const float      px  = (i < n) ? qx[i] : 0.F, py = (i < n) ? qy[i] : 0.F;
    const int        arg = okNearestIdx(cx, cy, P, px, py, sx, sy);
    if (i < n && (threadIdx.x & (kNearestLanes - 1)) == 0)
        out[i] = arg;
}

// ---- CMA-ES controller ---------------------------------------------------------------
// CmaEsAgent::updateAction for the whole population (CovarianceMatrixAdaptationEvolution/main_eigen.cpp:45-68, Controller.cpp:
// 3-23): kLanes lanes per candidate, lane j evaluates unit j of a layer (its row of the weight matrix against the previous
// layer's outputs, fetched from the other lanes by shuffles), three layers = three rounds.  Sums start from the bias and add
// w * x in ascending input order (fp32, no FMA), as the oracle's ctrl_forward does.  One launch per Environment step, on the
// handle's stream: captured into the caller's HIP graph together with okenv_step and the tracker.
template <int kLanes>
__global__ void okControllerKernel(OkDeviceState st, const float *params, int num_params, int N, int R, int hidden, float throttle, float steering_scale)
{
    const int  gid  = blockIdx.x * blockDim.x + threadIdx.x;
    const int  a    = gid / kLanes;
    const int  j    = gid % kLanes;
    const bool ok   = a < N;
    const int  ac   = ok ? a : 0;
    const int  h2   = hidden / 2;
    const float *prm = params + static_cast<size_t>(ac) * num_params;
    const float *w1 = prm, *b1 = w1 + hidden * R, *w2 = b1 + hidden, *b2 = w2 + h2 * hidden, *w3 = b2 + h2, *b3 = w3 + 2 * h2;
    // layer 1: inputs are the agent's R distances / kSensorRange (every lane reads them; they sit in one or two cache lines)
    float a1 = 0.F;
    if (j < hidden)
    {
        float sum = b1[j];
        for (int i = 0; i < R; ++i)
            sum = sum + w1[j * R + i] * (st.dist[static_cast<long>(ac) * R + i] / OK_SENSOR_RANGE);
        a1 = ok_tanhf(sum);
    }
    // layer 2: lane j < hidden / 2
    float sum2 = (j < h2) ? b2[j] : 0.F;
    for (int i = 0; i < hidden; ++i)
    {
        const float xi = __shfl(a1, i, kLanes);
        if (j < h2)
            sum2 = sum2 + w2[j * hidden + i] * xi;
    }
    const float a2 = (j < h2) ? ok_tanhf(sum2) : 0.F;
    // layer 3: lanes 0 and 1 (only output 0 is used: steering; the throttle is a constant, main_eigen.cpp:65-67)
    float sum3 = (j < 2) ? b3[j] : 0.F;
    for (int i = 0; i < h2; ++i)
    {
        const float xi = __shfl(a2, i, kLanes);
        if (j < 2)
            sum3 = sum3 + w3[j * h2 + i] * xi;
    }
    if (ok && j == 0)
    {
        st.thr[a]   = throttle;
        st.steer[a] = ok_tanhf(sum3) * steering_scale;
    }
}

// ---- rollout bookkeeping ------------------------------------------------------------

// The callers' loop body after env.step() (begin != 0: the episode start of main_eigen.cpp:128-133).  kNearestLanes
// lanes per agent for the index-progress reward (they share the nearest-index scan, the first lane does the
// bookkeeping); the +1 reward needs no track index and runs one thread per agent.
__global__ void okTrackerKernel(OkDeviceState st, const float *cx, const float *cy, int P, OkTracker tr, int N, int kind, int begin)
{
    __shared__ float sx[1024], sy[1024];
    const int        lanes = (kind == kRewardProgress) ? kNearestLanes : 1;
    const int        gid   = blockIdx.x * blockDim.x + threadIdx.x;
    const int        i     = gid / lanes;
    const bool       ok    = i < N;
    int              idx   = 0;
    if (kind == kRewardProgress)
        idx = okNearestIdx(cx, cy, P, ok ? st.pos_x[i] : 0.F, ok ? st.pos_y[i] : 0.F, sx, sy);
    if (!ok || gid % lanes != 0)
        return;
    const bool crashed = st.crashed[i] != 0;
    if (begin)
    {
        tr.prev_idx[i]     = idx;
        tr.fitness[i]      = 0.F;
        tr.reward[i]       = 0.F;
        tr.ep_steps[i]     = 0U;
        tr.prev_crashed[i] = crashed ? 1 : 0;
        return;
    }
    const bool was_crashed = tr.prev_crashed[i] != 0;
    float      fitness     = tr.fitness[i];
    float      reward      = 0.F;
    if (was_crashed && !crashed)
    { // re-placed since the last update: this step was the new episode's initial observation
        fitness        = 0.F;
        tr.ep_steps[i] = 0U;
        tr.prev_idx[i] = idx;
    }
    else if (kind == kRewardStep)
    {
        reward = 1.F;
        fitness += 1.F;
        tr.ep_steps[i] += 1U;
    }
    else if (!crashed)
    {
        const int progress = idx - tr.prev_idx[i];
        tr.prev_idx[i]     = idx;
        reward             = static_cast<float>(progress < 0 ? -progress : progress);
        fitness += reward;
        tr.ep_steps[i] += 1U;
    }
    else if (st.timed_out[i] != 0)
    {
        fitness = 0.F;
    }
    tr.fitness[i] = fitness;
    tr.reward[i]  = reward;
    if (crashed && !was_crashed)
        tr.ep_return[i] = fitness;
    tr.prev_crashed[i] = crashed ? 1 : 0;
}

// ---- EvolutionaryRacer generation kernels ---------------------------------------

// Network() random initialisation (Network.hpp:105-106): U[-1,1) per real weight, zero in the padding.
__global__ void okGaInitWeightsKernel(float *w, int N, int R, int H, uint32_t seed, uint32_t agent_base)
{
    const int  per = OK_MLP_WEIGHTS(R);
    const long t   = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= static_cast<long>(N) * per)
        return;
    const int  a    = static_cast<int>(t / per), i = static_cast<int>(t % per);
    const bool real = ok_mlp_weight_is_real(static_cast<uint32_t>(i), R, H) != 0;
    w[t] = real ? ok_ga_initial_weight(seed, agent_base + static_cast<uint32_t>(a), static_cast<uint32_t>(i)) : 0.F;
}

// assignScores (MiscUtils.hpp:64-71) keeps the nearest index as a float; alive counter for the rollout's end test.
__global__ void okGaScoreKernel(const int32_t *nearest, float *score, int N)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N)
        score[i] = static_cast<float>(nearest[i]);
}

__global__ void okAliveCountKernel(const uint8_t *crashed, int N, int *out)
{
    const int i     = blockIdx.x * blockDim.x + threadIdx.x;
    const int alive = (i < N && crashed[i] == 0) ? 1 : 0;
    const unsigned long long m = __ballot(alive);
    if ((threadIdx.x & 63) == 0 && m)
        atomicAdd(out, __popcll(m));
}

// Agents whose position lies outside the grid box (the track's bounding box plus the builder's pad): out[0] counts those with
// crashed_ == false, out[1] all of them.  An agent out there casts no ray that can reach a segment within range once it is more
// than a sensor range away, can therefore never crash by lidar (it tunnelled through both boundary
// polylines) and only the standstill timeout can still end it.
__global__ void okOffGridCountKernel(const float *pos_x, const float *pos_y, const uint8_t *crashed, int N, float x0, float y0, float x1,
                                     float y1, int *out)
{
    const int  i   = blockIdx.x * blockDim.x + threadIdx.x;
    const bool off = i < N && !(pos_x[i] >= x0 && pos_x[i] <= x1 && pos_y[i] >= y0 && pos_y[i] <= y1); // NaN poses count as off the grid
    const unsigned long long m_all   = __ballot(off);
    const unsigned long long m_alive = __ballot(off && crashed[i] == 0);
    if ((threadIdx.x & 63) == 0 && m_all)
    {
        atomicAdd(out + 1, __popcll(m_all));
        if (m_alive)
            atomicAdd(out, __popcll(m_alive));
    }
}

// The kNumParents = 5 best agents (Mating.hpp:115-119): descending score, ties to the lower index.  One workgroup.
__global__ void __launch_bounds__(1024) okGaTopKernel(const float *score, int N, int32_t *parents, float *parent_score, int K)
{
    __shared__ float sv[1024];
    __shared__ int   si[1024];
    __shared__ int   chosen[16];
    for (int k = 0; k < K; ++k)
    {
        float best = -__builtin_huge_valf();
        int   arg  = 0x7FFFFFFF;
        for (int i = threadIdx.x; i < N; i += blockDim.x)
        {
            bool taken = false;
            for (int q = 0; q < k; ++q)
                taken |= (chosen[q] == i);
            const float v = score[i];
            if (!taken && (v > best || (v == best && i < arg)))
            {
                best = v;
                arg  = i;
            }
        }
        sv[threadIdx.x] = best;
        si[threadIdx.x] = arg;
        __syncthreads();
        for (int off = blockDim.x / 2; off > 0; off >>= 1)
        {
            if (static_cast<int>(threadIdx.x) < off)
            {
                const float ov = sv[threadIdx.x + off];
                const int   oi = si[threadIdx.x + off];
                if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x]))
                {
                    sv[threadIdx.x] = ov;
                    si[threadIdx.x] = oi;
                }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0)
        {
            chosen[k]       = si[0];
            parents[k]      = si[0];
            parent_score[k] = sv[0];
        }
        __syncthreads();
    }
}

// Parent pair of every offspring (Mating.hpp:128-152): offspring 0 clones the best, offspring 1 mates the best with
// itself, the rest draw two DIFFERENT parents with probability proportional to the parents' scores
// (std::discrete_distribution; uniform if all scores are zero).  pair[o] = first | second << 8, clone flag in bit 16.
// mate2AgentsSelective (Mating.hpp:52-99) for every weight of every offspring: 10 % mutation to U[-1,1), otherwise the
// dominant (higher score; the second on ties) parent's weight with probability 0.75, else the other's.
__global__ void okGaMateKernel(const float *w_old, float *w_new, const int32_t *parents, const float *parent_score, int K, int N, int R,
                               int H, uint32_t seed, uint32_t generation, uint32_t agent_base)
{
    const int  per = OK_MLP_WEIGHTS(R);
    const long t   = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= static_cast<long>(N) * per)
        return;
    const uint32_t o = static_cast<uint32_t>(t / per), i = static_cast<uint32_t>(t % per);
    float          ps[16];
    for (int k = 0; k < K; ++k)
        ps[k] = parent_score[k];
    const uint32_t og     = agent_base + o; // global offspring id keeps sharded islands' streams distinct
    const uint32_t pair   = ok_ga_parent_pair(ps, K, seed, og, generation);
    const uint32_t first  = pair & 0xFFU, second = (pair >> 8) & 0xFFU;
    const bool     clone  = (pair >> 16) & 1U;
    const uint32_t dom    = (ps[first] > ps[second]) ? first : second;
    const uint32_t sub    = (ps[first] > ps[second]) ? second : first;
    const float    wd     = w_old[static_cast<long>(parents[dom]) * per + i];
    const float    wsub   = w_old[static_cast<long>(parents[sub]) * per + i];
    const bool     real   = ok_mlp_weight_is_real(i, R, H) != 0;
    float          out    = wd;
    if (real && !clone)
    {
        const ok_u32x4 r = ok_philox4x32(og, i, 2U, generation, seed, 0x6F6B656EU);
        if (ok_u01(r.v[0]) < 0.1F)
            out = (ok_u01(r.v[1]) - 0.5F) * 2.F;
        else
            out = (ok_u01(r.v[2]) < 0.75F) ? wd : wsub;
    }
    w_new[t] = real ? out : 0.F;
}

// ---- episodes ("step everybody until every agent has crashed") -----------------------------------------------------------
// Host side: okenv_episode_begin / okenv_episode_compact / okenv_episode_end (include/okenv.h).

// Start of an episode: nobody is settled; an agent that is crashed already counts as crashed "in step 0".
__global__ void okEpisodeBeginKernel(const uint8_t *crashed, uint8_t *settled, uint32_t *crash_step, unsigned long long *live, int N)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a == 0)
        *live = 0ULL;
    if (a >= N)
        return;
    settled[a]    = 0;
    crash_step[a] = crashed[a] != 0 ? 0U : 0xFFFFFFFFU;
}

// counts[0] = agents alive, counts[1] = agents listed.  The list of the agents that are not settled yet, ascending.  One
// workgroup: every thread takes a contiguous run of ceil(N / 1024) agents, the runs' counts are scanned once (wave ballots
// are no help with runs; a shared-memory scan over the 16 waves' totals is), then every thread writes its run's entries.
__global__ void __launch_bounds__(1024) okEpisodeCompactKernel(const uint8_t *settled, const uint8_t *crashed, int N, int32_t *active, int32_t *counts)
{
    __shared__ int wave_keep[16], wave_live[16];
    const int t = static_cast<int>(threadIdx.x), lane = t & 63, wave = t >> 6;
    const int per = (N + 1023) / 1024;
    const int lo = t * per, hi = (lo + per < N) ? lo + per : N;
    int keep = 0, live = 0;
    for (int i = lo; i < hi; ++i)
    {
        keep += settled[i] == 0 ? 1 : 0;
        live += crashed[i] == 0 ? 1 : 0;
    }
    // inclusive scan of `keep` inside the wave (and the wave's sum of `live`)
    int scan = keep;
    for (int off = 1; off < 64; off <<= 1)
    {
        const int o = __shfl_up(scan, off, 64);
        if (lane >= off)
            scan += o;
    }
    int live_w = live;
    for (int off = 32; off > 0; off >>= 1)
        live_w += __shfl_xor(live_w, off, 64);
    if (lane == 63)
        wave_keep[wave] = scan;
    if (lane == 0)
        wave_live[wave] = live_w;
    __syncthreads();
    int before = scan - keep, total = 0, alive = 0;
    for (int w = 0; w < 16; ++w)
    {
        if (w < wave)
            before += wave_keep[w];
        total += wave_keep[w];
        alive += wave_live[w];
    }
    for (int i = lo; i < hi; ++i)
        if (settled[i] == 0)
            active[before++] = i;
    if (t == 0)
    {
        counts[0] = alive;
        counts[1] = total;
    }
}

// out[0] = T, the number of steps the reference's loop takes: it leaves after the step in which the last agent crashes
// (genetic_learner_sim.cpp:84-92, q_racer_sim.cpp:163-190; at least one step is always taken); when somebody is still alive
// (the caller's own step cap) every step taken counts.  out[1] = agents alive.  One workgroup.
__global__ void __launch_bounds__(1024) okEpisodeEndKernel(const uint8_t *crashed, const uint32_t *crash_step, int N, uint32_t steps_taken, uint32_t *out)
{
    __shared__ uint32_t s_max, s_alive;
    if (threadIdx.x == 0)
    {
        s_max   = 0U;
        s_alive = 0U;
    }
    __syncthreads();
    uint32_t m = 0U, alive = 0U;
    for (int i = threadIdx.x; i < N; i += blockDim.x)
    {
        if (crashed[i] == 0)
            ++alive;
        else
            m = crash_step[i] > m ? crash_step[i] : m; // (0xFFFFFFFF cannot occur for a crashed agent: see okenv_episode_begin)
    }
    atomicMax(&s_max, m);
    atomicAdd(&s_alive, alive);
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t T = s_alive != 0U ? steps_taken : (s_max < 1U ? 1U : s_max);
        T          = T > steps_taken ? steps_taken : T;
        out[0]     = T;
        out[1]     = s_alive;
    }
}

// The launches of an episode overrun T (they end at a launch boundary).  The only thing the overrun changes about an agent is the
// action of those that crashed in step T itself: the reference never asks their policy again, here they have taken their step as
// a crashed agent.  Put back the action of step T.  (Agents that crashed before T hold the action the policy gives a crashed
// agent -- the same in every step after the crash, in the reference's loop as well.)
__global__ void okEpisodeFixupKernel(OkDeviceState st, const uint32_t *crash_step, const float *crash_thr, const float *crash_steer, const uint32_t *T, int N)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= N || st.crashed[a] == 0 || crash_step[a] != T[0] || crash_step[a] == 0U)
        return;
    st.thr[a]   = crash_thr[a];
    st.steer[a] = crash_steer[a];
}

// Q-learning: what the reference's loop does with an agent after its crash, up to and including step T (q_racer_sim.cpp:158-182):
// every step an epsilon-greedy action from the row of its (no longer changing) state, then learn(state, action, -200, next state)
// with the next state it has seen ever since the crash step.  Only the agent's own table is involved, so the steps c + 1 .. T
// are replayed here once T is known; the step kernel leaves crashed agents' tables alone inside an episode.
// kSettleLanes lanes per agent: the Philox draws of sixteen steps are made side by side (they do not depend on the table), then
// every lane of the group takes the sixteen learn() steps in order from the shuffled draws (the same values in each lane).
constexpr int kSettleLanes = 16;

__global__ void okQSettleKernel(OkDeviceState st, float *q_table, const int32_t *q_state, int32_t *q_action, const int32_t *q_next_state,
                                const uint32_t *crash_step, const uint32_t *T_ptr, int N, uint32_t seed, uint32_t agent_base,
                                uint32_t step_base, float epsilon, int R, int r0, int r1, int r2, int r3, int r4)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int a   = gid / kSettleLanes, l = gid % kSettleLanes;
    if (a >= N || st.crashed[a] == 0) // (the same for all lanes of a group)
        return;
    const uint32_t T = T_ptr[0], c = crash_step[a];
    if (c >= T)
        return;
    float    *table = q_table + static_cast<size_t>(a) * (OK_Q_STATES * OK_Q_ACTIONS);
    const int sc    = q_state[a];
    // An agent that was crashed when the episode began (c == 0) never wrote q_next_state: what it sees as its next state in every
    // step is discretizeState() of its stale observation (q_racer_sim.cpp:173: the rays keep the hit points of the crash step, the
    // origin no longer moves), i.e. of the distances its last step left behind.  Right after okenv_q_begin_episode that IS q_state;
    // after a crash in an earlier okenv_rollout_q call outside an episode it is not (q_state stays the state before the crash, :177).
    int sn;
    if (c == 0U)
    {
        const float *d = st.dist + static_cast<long>(a) * R;
        sn             = ok_q_bin(d[r0]) + 3 * ok_q_bin(d[r1]) + 9 * ok_q_bin(d[r2]) + 27 * ok_q_bin(d[r3]) + 81 * ok_q_bin(d[r4]);
    }
    else
        sn = q_next_state[a];
    const bool same = sn == sc;
    float c0 = table[sc * OK_Q_ACTIONS + 0], c1 = table[sc * OK_Q_ACTIONS + 1], c2 = table[sc * OK_Q_ACTIONS + 2];
    const float n0 = table[sn * OK_Q_ACTIONS + 0], n1 = table[sn * OK_Q_ACTIONS + 1], n2 = table[sn * OK_Q_ACTIONS + 2];
    int action = q_action[a];
    for (uint32_t i0 = c + 1U; i0 <= T; i0 += kSettleLanes)
    { // episode step i is global step step_base + i - 1 of the Philox stream
        const uint32_t i    = i0 + static_cast<uint32_t>(l);
        const int      draw = i <= T ? ok_q_draw_action(seed, agent_base + static_cast<uint32_t>(a), step_base + i - 1U, epsilon) : -1;
        const int      n    = (T - i0 + 1U) < static_cast<uint32_t>(kSettleLanes) ? static_cast<int>(T - i0 + 1U) : kSettleLanes;
        // the sixteen draws, two bits each (3 = exploit), gathered into one word that every lane of the group holds
        uint32_t packed = static_cast<uint32_t>(draw < 0 ? 3 : draw) << (2 * l);
        for (int off = 1; off < kSettleLanes; off <<= 1)
            packed |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(packed), off, kSettleLanes));
        for (int k = 0; k < n; ++k)
        {
            const int d = static_cast<int>((packed >> (2 * k)) & 3U);
            action      = d == 3 ? ok_q_argmax3(c0, c1, c2) : d;
            float mq    = same ? c0 : n0;
            const float m1 = same ? c1 : n1, m2 = same ? c2 : n2;
            mq             = (m1 > mq) ? m1 : mq;
            mq             = (m2 > mq) ? m2 : mq;
            const float old_q = (action == 0) ? c0 : ((action == 1) ? c1 : c2);
            const float new_q = ok_q_learn(old_q, mq, -200.0F);
            c0                = (action == 0) ? new_q : c0;
            c1                = (action == 1) ? new_q : c1;
            c2                = (action == 2) ? new_q : c2;
        }
    }
    if (l == 0)
    {
        table[sc * OK_Q_ACTIONS + 0] = c0;
        table[sc * OK_Q_ACTIONS + 1] = c1;
        table[sc * OK_Q_ACTIONS + 2] = c2;
        q_action[a]                  = action;
        ok_q_action_values(action, &st.thr[a], &st.steer[a]);
    }
}

// ---- RLRacers/Q_Learning service kernels ----------------------------------------------------------------------

__global__ void okQInitTableKernel(float *q, long n)
{
    const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n)
        q[i] = OK_Q_INVALID; // QAgent.hpp:64-68
}

// start of an episode (q_racer_sim.cpp:132-154, after the initial env.step()): current state from the fresh
// observation, previous track index = the reset point's nearest index
__global__ void okQBeginEpisodeKernel(const float *dist, int R, int r0, int r1, int r2, int r3, int r4, int32_t *q_state, int32_t *q_prev,
                                      const int32_t *reset_nearest, int N)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= N)
        return;
    const float *d = dist + static_cast<long>(a) * R;
    q_state[a]     = ok_q_bin(d[r0]) + 3 * ok_q_bin(d[r1]) + 9 * ok_q_bin(d[r2]) + 27 * ok_q_bin(d[r3]) + 81 * ok_q_bin(d[r4]);
    q_prev[a]      = reset_nearest[0];
}

// shareCumulativeKnowledge (q_racer_sim.cpp:24-75), first half: per (state, action) the sum of the VALID entries over the
// agents, accumulated in agent order like the reference's loop, and their count.  One thread per table entry.
__global__ void okQTableSumsKernel(const float *q, int N, float *sum, float *count)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= OK_Q_STATES * OK_Q_ACTIONS)
        return;
    float total = OK_Q_INVALID, cnt = 0.F;
    for (int a = 0; a < N; ++a)
    {
        const float v = q[static_cast<size_t>(a) * (OK_Q_STATES * OK_Q_ACTIONS) + e];
        if (v != OK_Q_INVALID)
        {
            if (total == OK_Q_INVALID)
                total = 0.F;
            total += v;
            cnt += 1.F;
        }
    }
    sum[e]   = total;
    count[e] = cnt;
}

// second half: every agent's table becomes the mean (or stays invalid where nobody has a value yet)
__global__ void okQAssignAllKernel(float *q, int N, const float *sum, const float *count)
{
    const long t = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= static_cast<long>(N) * (OK_Q_STATES * OK_Q_ACTIONS))
        return;
    const int e = static_cast<int>(t % (OK_Q_STATES * OK_Q_ACTIONS));
    q[t]        = (count[e] > 0.F) ? sum[e] / count[e] : sum[e];
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
        view.hdr      = reinterpret_cast<const OkCellHdr *>(ok_lds + p.off_hdr);
        view.side_tol = p.side_tol;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        float sn, cs;
        ok_sincosf(ang[i], &sn, &cs);
        out_t[i] = okCastRay<kMode>(p, view, ox[i], oy[i], cs, sn);
    }
}

// Work the broad phase leaves per ray, for the population's CURRENT poses (measurement aid: bench.py's S_tested and SURVEY.md
// section 8d's valu_fraction): every live agent's rays are walked once, whole (ok_cast_poly_interval<true>), and the exact tests,
// grid cells and boundary points they meet are summed.  out[0] rays, [1] exact ray-segment tests, [2] cells, [3] points.
__global__ void __launch_bounds__(1024) okWorkStatsKernel(const OkStepParams p, unsigned long long *out)
{ // out[0..3]: rays, exact tests, cells, points; with the front / back split (p.fb) also out[4]: rays of a certified origin,
  // out[5]: rays whose front walk was ambiguous, out[6]: rays that walked the back image
    extern __shared__ __attribute__((aligned(16))) unsigned char ok_lds[];
    okStageImage(p, ok_lds);
    OkPolyView view{};
    view.g        = p.geom;
    view.slots    = reinterpret_cast<const OkPoint *>(ok_lds);
    view.hdr      = reinterpret_cast<const OkCellHdr *>(ok_lds + p.off_hdr);
    view.side_tol = p.side_tol;
    view.e_s      = p.fb_e_s;
    view.e_t      = p.fb_e_t;
    view.e_s_over_e_t = p.fb_e_t > 0.F ? p.fb_e_s / p.fb_e_t : 0.F;
    OkPolyView back = view;
    if (p.fb != 0U)
    {
        back.slots    = reinterpret_cast<const OkPoint *>(ok_lds + p.fb_back_off);
        back.hdr      = reinterpret_cast<const OkCellHdr *>(ok_lds + p.fb_back_off_hdr);
        back.side_tol = p.fb_back_side_tol;
    }
    unsigned long long rays = 0, tests = 0, cells = 0, points = 0, n_cert = 0, n_amb = 0, n_back = 0;
    const long         total = static_cast<long>(p.N) * p.R;
    for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * blockDim.x)
    {
        const int a = static_cast<int>(i / p.R), r = static_cast<int>(i % p.R);
        if (p.st.crashed[a] != 0)
            continue;
        float sn, cs, rdy, rdx;
        ok_sincosf(OK_DEG2RAD * p.st.rot[a], &sn, &cs);
        ok_sincosf(OK_DEG2RAD * (p.st.rot[a] + p.ray_deg[r]), &rdy, &rdx);
        const float ox = p.st.pos_x[a] + p.sensor_offset * cs, oy = p.st.pos_y[a] + p.sensor_offset * sn;
        uint32_t    t = 0, c = 0, pt = 0;
        if (p.fb != 0U)
        {
            const bool             cert = okOriginChiScalar(view, ox, oy, p.fb_t12, p.fb_t34) != 0;
            const OkIntervalResult rf   = ok_cast_poly_interval<true, true>(view, ox, oy, rdx, rdy, 0.F, OKRC_INF, &t, &c, &pt);
            n_cert += cert ? 1U : 0U;
            n_amb += rf.amb ? 1U : 0U;
            if (!cert || rf.amb)
            {
                (void)ok_cast_poly_interval<true>(back, ox, oy, rdx, rdy, 0.F, OKRC_INF, &t, &c, &pt, nullptr, false, rf.min_t);
                n_back += 1U;
            }
        }
        else
            (void)ok_cast_poly_interval<true>(view, ox, oy, rdx, rdy, 0.F, OKRC_INF, &t, &c, &pt);
        rays += 1;
        tests += t;
        cells += c;
        points += pt;
    }
    atomicAdd(&out[0], rays);
    atomicAdd(&out[1], tests);
    atomicAdd(&out[2], cells);
    atomicAdd(&out[3], points);
    if (p.fb != 0U)
    {
        atomicAdd(&out[4], n_cert);
        atomicAdd(&out[5], n_amb);
        atomicAdd(&out[6], n_back);
    }
}
