// okenv_capi.hip -- implementation of the C ABI declared in include/okenv.h.
//
// Host side of the batched Environment step: owns the device-resident struct-of-arrays state, builds and
// uploads the uniform grid, chooses the launch geometry and enqueues the kernels of okenv_kernels.h on the
// handle's HIP stream.  There is no CPU fallback anywhere in this file: without a usable GPU
// okenv_create fails with OKENV_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/okenv.h"
#include "Environment/RaceTrack.h"
#include "ok_grid.h"
#include "okenv_kernels.h"

namespace
{
thread_local std::string g_create_error = "";

constexpr size_t kLdsBudget = 160U * 1024U; // one workgroup may take the whole CU's LDS on gfx950
// LDS kept free behind the track image for the Q-learning kernel's copy of the centre line (8 B per point)
constexpr size_t kLdsReserve = 16U * 1024U;

struct EventPair
{
    hipEvent_t start, stop;
};

// spin-wait hint of the host loops that watch a word in mapped memory
inline void okCpuRelax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
}
} // namespace

struct okenv
{
    int         device{0};
    hipStream_t stream{nullptr};
    bool        own_stream{true};
    int         N{0}, R{0}, S{0}, G{1}, rays_per_lane{1};
    uint32_t    flags{0};
    int         grid_mode{kGridLds};
    OkGridHost  grid;
    OkPolyImage poly;
    size_t      image_bytes{0};
    void       *d_image{nullptr};
    OkSeg      *d_segs{nullptr};
    uint32_t   *d_refs32{nullptr}, *d_start{nullptr};
    float      *d_ray_deg{nullptr};
    float       sensor_offset{0.F};
    float      *d_cx{nullptr}, *d_cy{nullptr}, *d_chead{nullptr};
    int         P{0}, centerline_capacity{0};
    float      *d_lane_l{nullptr}, *d_lane_r{nullptr}; // left_bound_inner_ / right_bound_inner_, xy pairs
    int         lane_points{0}, lane_capacity{0};
    uint32_t    reset_flags{0}, reset_seed{0}, reset_agent_base{0}; // okenv_set_auto_reset
    // packed host exchange (okenv_step_packed): pinned host staging + device records, allocated on first use
    void       *h_stage{nullptr};
    void       *h_stage_device{nullptr}; // the same buffer as the device sees it (mapped host memory)
    void       *d_stage{nullptr};
    OkTracker   tracker{};
    int         tracker_kind{-1};
    // Environment steps taken by okenv_step / okenv_rollout_policy.  While auto-reset is on the device copy is the
    // authoritative one (it is the epoch of the reset draws and advances under graph replay); otherwise the host copy.
    uint32_t    step_count{0};
    uint32_t   *d_step_count{nullptr};
    OkDeviceState st{};
    std::vector<void *> allocations;
    int         block_threads{1024}, grid_blocks{1};
    // EvolutionaryRacer state
    int      mlp_hidden{0};
    // front / back split of the segment set (ok_grid.h): classification, the two images, and their copy on the device as one blob
    // [front | back]; fb_ok: the cooperative kernel's non-packed launches use it
    OkFrontBack       fbc;
    OkFrontBackImages fbi;
    bool              fb_ok{false};
    uint8_t          *d_image_fb{nullptr};
    size_t            fb_bytes{0}, fb_back_off{0};
    int      cus{256}; // compute units of the device (hipDeviceAttributeMultiprocessorCount), asked once in okenv_create
    float   *d_mlp_w{nullptr}, *d_mlp_w_new{nullptr}, *d_score{nullptr}, *d_parent_score{nullptr};
    int32_t *d_nearest{nullptr}, *d_parents{nullptr}, *d_alive{nullptr};
    // Q-learning state
    float   *d_q_table{nullptr};
    int32_t *d_q_state{nullptr}, *d_q_action{nullptr}, *d_q_prev{nullptr}, *d_q_reset_nearest{nullptr};
    void    *d_scratch{nullptr};       // staging for calls that take host arrays (every user synchronises before it returns)
    size_t   scratch_bytes{0};
    float   *d_ctrl_params{nullptr};   // CMA-ES controllers: [N][ctrl_num_params]
    int      ctrl_hidden{0}, ctrl_num_params{0};
    float   *d_q_reset_query{nullptr}; // (x, y) of the episode's reset point, the query of its nearest-index kernel
    float    q_reset_query[2]{0.F, 0.F}; // host copy the upload reads: lives as long as the handle
    float   *d_q_sums{nullptr};
    uint16_t *d_cl_start{nullptr}, *d_cl_idx{nullptr}; // centre line bucketed by grid cell (Q-learning's nearest index)
    size_t    cl_capacity{0};
    bool      cl_dirty{true};
    int      q_ray[5]{0, 0, 0, 0, 0};
    float    q_epsilon{0.F};
    unsigned long long *d_stamps{nullptr}; // -DOKENV_STAMPS builds: per-wave stamps of the last launch
    size_t    stamp_waves{0}, stamp_waves_cap{0};
    int       tail_max_agents{-1}; // episode lists up to this long are stepped by okStepTailKernel: -1 = what one round of
                                   // workgroups holds, 0 = never (OKENV_TAIL_MAX_AGENTS)
    // episodes (okenv_episode_begin / _compact / _end)
    bool      episode{false};
    int       n_active{-1};       // agents listed for the policy rollouts (-1: everybody, no list)
    int       ep_kind{0};         // policy of the episode's rollouts: 0 none yet, kPolicyMlp, kPolicyQ
    uint32_t  ep_steps{0};        // steps taken since okenv_episode_begin
    uint32_t  ep_q_seed{0}, ep_q_agent_base{0}, ep_q_step_base{0}; // Q-learning: the draws' key, global step of episode step 1
    float     ep_q_epsilon{0.F};
    int32_t  *d_active{nullptr}, *d_ep_counts{nullptr}, *d_q_next_state{nullptr};
    uint8_t  *d_settled{nullptr};
    uint32_t *d_crash_step{nullptr}, *d_ep_out{nullptr};
    float    *d_crash_thr{nullptr}, *d_crash_steer{nullptr};
    unsigned long long *d_live{nullptr};
    std::vector<float> host_cx, host_cy, host_chead, host_ray_deg;
    bool        coop{false};        // cooperative two-phase kernel (LDS form, one ray per lane)
    int         agents_per_block{0}; // coop, tiny populations: agents per workgroup (the other lanes only stage); 0 = dense
    uint32_t    packed_seq{0};      // okenv_step_packed: sequence number of the last launch's completion word
    // resident step kernel (okenv_step_packed called in quick succession, see startResident)
    hipStream_t resident_stream{nullptr};
    bool        resident{false};
    int         resident_mode{-1};  // OKENV_RESIDENT: 0 never, 1 from the first eligible step on, default: after a run of quick steps
    int         resident_steps{0}, resident_fallbacks{0}; // statistics (okenv_get_info)
    int         resident_stall_us{0}; // OKENV_RESIDENT_STALL_US, fault injection for the tests: the host dawdles this long before
                                      // it hands a step to the resident kernel, which has left by then
    int         resident_need{16};  // quick steps in a row that start it: kResidentStreak, more after residencies that ended early
    int         resident_served_now{0}; // steps the current residency has served
    int         packed_streak{0};   // packed steps in a row that came within kResidentGapUs of the one before
    std::chrono::steady_clock::time_point packed_last_end{};
    size_t      stage_slots_off{0}; // where the agents' slots lie in h_stage
    float       phase1_range{48.F}; // T1 of the cooperative kernel [px]
    std::string last_error;
    bool        timing{false};
    std::vector<EventPair> events;      // recorded pairs awaiting resolution
    double      timing_carry_ms{0.0};    // pairs resolved early (before a stream switch), not yet reported
    uint64_t    timing_carry_n{0};
    std::vector<EventPair> event_pool;  // reusable pairs
};

struct okenv_track
{
    std::unique_ptr<RaceTrack> track;
    std::vector<Segment2d>     segments;
};

namespace
{
int fail(okenv *h, const int code, const std::string &msg)
{
    if (h)
        h->last_error = msg;
    else
        g_create_error = msg;
    return code;
}

#define OK_HIP(h, call)                                                                                                \
    do                                                                                                                 \
    {                                                                                                                  \
        const hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess)                                                                                          \
            return fail((h), OKENV_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                        \
    } while (0)

template <class T>
int devAlloc(okenv *h, T **out, const size_t count)
{
    void *p = nullptr;
    OK_HIP(h, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    h->allocations.push_back(p); // (owned by the handle from here on, whatever happens next)
    OK_HIP(h, hipMemsetAsync(p, 0, std::max<size_t>(count, 1) * sizeof(T), h->stream));
    *out = static_cast<T *>(p);
    return OKENV_OK;
}

// Buffers that are created on first use, in groups: each one is allocated if it is not there yet, so that a call which ran out of
// memory half-way through its group can simply be repeated (the sizes depend on the handle's N and R only).
template <class T>
int devEnsure(okenv *h, T **out, const size_t count)
{
    return *out != nullptr ? OKENV_OK : devAlloc(h, out, count);
}

// Device staging space owned by the handle, grown on demand.  Each caller waits for the stream before it returns, so one
// buffer serves them all; the stream-ordered pool (hipMallocAsync) is not used anywhere: a tiny upload into pool memory
// that was freed again without a wait gave q_racer_sim's episodes two different outcomes from run to run.
int deviceScratch(okenv *h, const size_t bytes, void **out)
{
    if (bytes > h->scratch_bytes)
    {
        OK_HIP(h, hipStreamSynchronize(h->stream)); // nobody is still reading the old one
        uint8_t *p  = nullptr;
        const int rc = devAlloc(h, &p, bytes + bytes / 2U);
        if (rc != OKENV_OK)
            return rc;
        h->d_scratch     = p;
        h->scratch_bytes = bytes + bytes / 2U;
    }
    *out = h->d_scratch;
    return OKENV_OK;
}

int pow2ceil(int v)
{
    int p = 1;
    while (p < v)
        p <<= 1;
    return p;
}

struct FieldDesc
{
    void  *ptr;
    size_t bytes;
};

FieldDesc fieldOf(okenv *h, const int f)
{
    const size_t N = h->N, NR = static_cast<size_t>(h->N) * h->R;
    auto        &s = h->st;
    switch (f)
    {
    case OKENV_F_POS_X: return {s.pos_x, 4 * N};
    case OKENV_F_POS_Y: return {s.pos_y, 4 * N};
    case OKENV_F_ROT: return {s.rot, 4 * N};
    case OKENV_F_SPEED: return {s.speed, 4 * N};
    case OKENV_F_ACC: return {s.acc, 4 * N};
    case OKENV_F_THROTTLE: return {s.thr, 4 * N};
    case OKENV_F_STEER: return {s.steer, 4 * N};
    case OKENV_F_MODE: return {s.mode, N};
    case OKENV_F_CRASHED: return {s.crashed, N};
    case OKENV_F_TIMED_OUT: return {s.timed_out, N};
    case OKENV_F_DISP_CTR: return {s.disp_ctr, 4 * N};
    case OKENV_F_DISP_X: return {s.disp_x, 4 * N};
    case OKENV_F_DISP_Y: return {s.disp_y, 4 * N};
    case OKENV_F_DISP_TO: return {s.disp_to, N};
    case OKENV_F_HIT_X: return {s.hit_x, 4 * NR};
    case OKENV_F_HIT_Y: return {s.hit_y, 4 * NR};
    case OKENV_F_REL_X: return {s.rel_x, 4 * NR};
    case OKENV_F_REL_Y: return {s.rel_y, 4 * NR};
    case OKENV_F_DIST: return {s.dist, 4 * NR};
    case OKENV_F_REWARD: return {h->tracker.reward, h->tracker.reward ? 4 * N : 0};
    case OKENV_F_FITNESS: return {h->tracker.fitness, h->tracker.fitness ? 4 * N : 0};
    case OKENV_F_TRACK_IDX: return {h->tracker.prev_idx, h->tracker.prev_idx ? 4 * N : 0};
    case OKENV_F_EPISODE_STEPS: return {h->tracker.ep_steps, h->tracker.ep_steps ? 4 * N : 0};
    case OKENV_F_EPISODE_RETURN: return {h->tracker.ep_return, h->tracker.ep_return ? 4 * N : 0};
    case OKENV_F_PREV_CRASHED: return {h->tracker.prev_crashed, h->tracker.prev_crashed ? N : 0};
    default: return {nullptr, 0};
    }
}

OkStepParams baseParams(okenv *h)
{
    OkStepParams p{};
    p.st            = h->st;
    p.N             = h->N;
    p.R             = h->R;
    p.G             = h->G;
    p.rays_per_lane = h->rays_per_lane;
    p.ray_deg       = h->d_ray_deg;
    p.sensor_offset = h->sensor_offset;
    p.image         = static_cast<const uint8_t *>(h->d_image);
    p.image_bytes   = static_cast<uint32_t>(h->image_bytes);
    p.off_hdr       = static_cast<uint32_t>(h->poly.off_hdr);
    p.side_tol      = h->poly.side_tol;
    p.geom          = h->grid.g;
    p.g_segs        = h->d_segs;
    p.g_refs32      = h->d_refs32;
    p.g_start       = h->d_start;
    p.S             = h->S;
    p.n_steps       = 1;
    p.do_move       = 1;
    p.action_source = kActionsStored;
    p.cx            = h->d_cx;
    p.cy            = h->d_cy;
    p.chead         = h->d_chead;
    p.P             = h->P;
    p.reset_flags   = h->reset_flags;
    p.reset_seed    = h->reset_seed;
    p.agent_base    = h->reset_agent_base;
    p.step_counter  = h->d_step_count;
    p.agents_per_block = h->coop ? h->agents_per_block : 0;
    p.lane_l        = h->d_lane_l;
    p.lane_r        = h->d_lane_r;
    p.mlp_w         = h->d_mlp_w;
    p.q_table       = h->d_q_table;
    p.q_state       = h->d_q_state;
    p.q_action      = h->d_q_action;
    p.q_prev_idx    = h->d_q_prev;
    for (int i = 0; i < 5; ++i)
        p.q_ray[i] = h->q_ray[i];
    p.q_epsilon = h->q_epsilon;
    p.ctrl_params     = h->d_ctrl_params;
    p.ctrl_num_params = h->ctrl_num_params;
    p.ctrl_hidden     = h->ctrl_hidden;
    p.trk             = h->tracker;
    p.trk_kind        = h->tracker_kind;
    p.cl_start  = h->cl_dirty ? nullptr : h->d_cl_start;
    p.cl_idx    = h->cl_dirty ? nullptr : h->d_cl_idx;
    if (h->episode)
    { // read by the policy kernels only
        p.settled      = h->d_settled;
        p.crash_step   = h->d_crash_step;
        p.crash_thr    = h->d_crash_thr;
        p.crash_steer  = h->d_crash_steer;
        p.live         = h->d_live;
        p.q_next_state = h->d_q_next_state;
        p.ep_step0     = h->ep_steps;
        if (h->n_active >= 0)
        {
            p.active           = h->d_active;
            p.n_active         = h->n_active;
            p.agents_per_block = 0; // listed agents are packed densely
        }
    }
    return p;
}

// An episode ends without its end-of-episode corrections when agent state is changed from outside the policy rollouts.
void dropEpisode(okenv *h)
{
    h->episode  = false;
    h->n_active = -1;
    h->ep_kind  = 0;
}

int beginTiming(okenv *h, EventPair *ev)
{
    if (!h->timing)
        return OKENV_OK;
    if (!h->event_pool.empty())
    {
        *ev = h->event_pool.back();
        h->event_pool.pop_back();
    }
    else
    {
        OK_HIP(h, hipEventCreate(&ev->start));
        OK_HIP(h, hipEventCreate(&ev->stop));
    }
    OK_HIP(h, hipEventRecord(ev->start, h->stream));
    return OKENV_OK;
}

int endTiming(okenv *h, const EventPair &ev)
{
    if (!h->timing)
        return OKENV_OK;
    OK_HIP(h, hipEventRecord(ev.stop, h->stream));
    h->events.push_back(ev);
    return OKENV_OK;
}

// LDS the Q-learning kernel needs behind the track image: the centre line (8 B per point) and its cell buckets
size_t qLdsBytes(const okenv *h)
{
    const size_t cells = static_cast<size_t>(h->grid.g.nx) * h->grid.g.ny;
    return 8U * static_cast<size_t>(h->P) + 2U * (cells + 1U) + 2U * static_cast<size_t>(h->P) + 16U;
}

// LDS every cooperative launch needs behind the image (the per-SIMD progress words)
constexpr size_t kCoopLdsExtra = 16U;
size_t coopLdsBytes(const okenv *h)
{
    return h->image_bytes + kCoopLdsExtra;
}

// Buckets the centre-line points by the cells of the raycast grid (CSR, indices ascending inside a cell) and uploads
// them; a centre line with more than 65535 points keeps the full scan.
int buildCenterlineBuckets(okenv *h)
{
    if (!h->cl_dirty)
        return OKENV_OK;
    const OkGridGeom &g     = h->grid.g;
    const size_t      cells = static_cast<size_t>(g.nx) * g.ny;
    if (h->P <= 0 || h->P > 65535)
        return OKENV_OK; // stays dirty: the kernel scans the whole line
    std::vector<uint16_t> start(cells + 1U, 0), idx(static_cast<size_t>(h->P));
    std::vector<int>      cell_of(static_cast<size_t>(h->P));
    std::vector<uint32_t> count(cells, 0U);
    for (int i = 0; i < h->P; ++i)
    {
        // same arithmetic as the kernel's lookup; points outside the grid go to the nearest border cell, which only
        // makes their bucket a superset
        int ci = static_cast<int>((h->host_cx[i] - g.x0) * g.inv_cell), cj = static_cast<int>((h->host_cy[i] - g.y0) * g.inv_cell);
        ci     = ci < 0 ? 0 : (ci >= g.nx ? g.nx - 1 : ci);
        cj     = cj < 0 ? 0 : (cj >= g.ny ? g.ny - 1 : cj);
        cell_of[i] = cj * g.nx + ci;
        ++count[static_cast<size_t>(cell_of[i])];
    }
    uint32_t run = 0;
    for (size_t c = 0; c < cells; ++c)
    {
        start[c] = static_cast<uint16_t>(run);
        run += count[c];
    }
    start[cells] = static_cast<uint16_t>(run);
    std::vector<uint32_t> fill(cells, 0U);
    for (int i = 0; i < h->P; ++i)
    {
        const size_t c                = static_cast<size_t>(cell_of[i]);
        idx[start[c] + fill[c]++] = static_cast<uint16_t>(i);
    }
    const size_t need = cells + 1U + static_cast<size_t>(h->P);
    if (need > h->cl_capacity)
    {
        int rc;
        if ((rc = devAlloc(h, &h->d_cl_start, cells + 1U)) || (rc = devAlloc(h, &h->d_cl_idx, static_cast<size_t>(h->P))))
            return rc;
        h->cl_capacity = need;
    }
    OK_HIP(h, hipMemcpyAsync(h->d_cl_start, start.data(), 2U * (cells + 1U), hipMemcpyHostToDevice, h->stream));
    OK_HIP(h, hipMemcpyAsync(h->d_cl_idx, idx.data(), 2U * static_cast<size_t>(h->P), hipMemcpyHostToDevice, h->stream));
    OK_HIP(h, hipStreamSynchronize(h->stream));
    h->cl_dirty = false;
    return OKENV_OK;
}

// ---- resident step kernel ------------------------------------------------------------------------------------------------
// The C++ facade's Environment::step() is one okenv_step_packed per step for 1-50 agents.  Launched one by one, such a
// step costs ~19-22 us of which only ~6 us are the step: 3 us to enqueue the launch, ~3 us until its first wave runs, ~2-3 us
// to stage the track image into LDS again, and the completion.  When the steps follow each other closely the kernel
// therefore stays: every workgroup (one agent each) keeps the image in its LDS and watches its agent's 64-byte slot in
// mapped host memory; the host puts the record there, the workgroup steps it and answers through the same record / hits /
// completion word as a one-shot launch.  The kernel leaves by itself after kResidentIdleTicks without work -- so it can
// never outlive its process by more than that -- or when the host says so; every other entry point of the C ABI stops it
// first (OK_QUIESCE), so nothing else ever runs against the handle's state while it is resident.
constexpr uint32_t kResidentIdleTicks = 30000U; // 100 MHz ticks: 300 us
constexpr double   kResidentGapUs     = 100.0;  // the host treats the kernel as gone after this long without a step
constexpr int      kResidentStreak    = 16;     // quick steps in a row before the kernel is made resident
constexpr int      kResidentShort     = 32;     // a residency that served fewer steps than this quadruples that number
constexpr uint32_t kResidentExit      = 0xFFFFFFFFU;
constexpr int      kResidentMaxAgents = 64;

int stopResident(okenv *h)
{
    if (!h->resident)
        return OKENV_OK;
    volatile uint32_t *slots = reinterpret_cast<volatile uint32_t *>(static_cast<uint8_t *>(h->h_stage) + h->stage_slots_off);
    for (int i = 0; i < h->N; ++i)
        for (int q = 3; q < 16; q += 4)
            slots[16 * i + q] = kResidentExit;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    h->resident      = false;
    h->packed_streak = 0;
    // A residency that ends after a few steps was not worth its start and stop -- and a caller that waits for the whole
    // device between its steps (hipDeviceSynchronize) sits out the kernel's idle time whenever one is resident: back off.
    h->resident_need       = h->resident_served_now < kResidentShort ? std::min(h->resident_need * 4, 1 << 20) : kResidentStreak;
    h->resident_served_now = 0;
    OK_HIP(h, hipStreamSynchronize(h->resident_stream));
    // a step the kernel left half done (it timed out between two workgroups) may have left the finish counter behind
    OK_HIP(h, hipMemsetAsync(h->d_step_count + 1, 0, sizeof(uint32_t), h->stream));
    OK_HIP(h, hipStreamSynchronize(h->stream));
    return OKENV_OK;
}

#define OK_QUIESCE(h)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        if ((h) != nullptr && (h)->resident)                                                                           \
        {                                                                                                              \
            const int qrc_ = stopResident(h);                                                                          \
            if (qrc_ != OKENV_OK)                                                                                      \
                return qrc_;                                                                                           \
        }                                                                                                              \
    } while (0)

// sequence numbers of packed steps: 0 is "nothing yet" and kResidentExit the resident kernel's order to leave
uint32_t nextPackedSeq(okenv *h)
{
    h->packed_seq = okNextPackedSeq(h->packed_seq);
    return h->packed_seq;
}

size_t coopLdsBytes(const okenv *h);

// Policy-free launches of a population with spare lanes and no phase 1 use the kernel's direct dealing of intervals to lanes
// (okStepCoopKernel's kDirect).
bool directIntervals(const okenv *h)
{
    return h->phase1_range <= 0.F && h->G >= 2 * h->R;
}

void useFrontBack(const okenv *h, OkStepParams &p);

// Starts the resident kernel on a stream of its own; `p` carries the exchange pointers of okenv_step_packed.
int startResident(okenv *h, OkStepParams p, volatile uint32_t *slots)
{
    OK_HIP(h, hipStreamSynchronize(h->stream)); // whatever was enqueued against the state comes first
    if (!h->resident_stream)
        OK_HIP(h, hipStreamCreateWithFlags(&h->resident_stream, hipStreamNonBlocking));
    for (int i = 0; i < 16 * h->N; ++i)
        slots[i] = 0U;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    p.done_seq = okNextPackedSeq(h->packed_seq); // the first number the kernel waits for
    size_t   lds = coopLdsBytes(h);
    uint32_t off = static_cast<uint32_t>(h->image_bytes);
    if (h->fb_ok && h->fb_bytes + kCoopLdsExtra <= kLdsBudget)
    { // the front / back split (staged once for the kernel's whole residency)
        useFrontBack(h, p);
        off = static_cast<uint32_t>(h->fb_bytes);
        lds = h->fb_bytes + kCoopLdsExtra;
    }
    if (directIntervals(h))
        hipLaunchKernelGGL((okStepCoopKernel<kPolicyNone, true, true, true>), dim3(h->grid_blocks), dim3(h->block_threads), lds, h->resident_stream, p, off,
                           h->phase1_range);
    else
        hipLaunchKernelGGL((okStepCoopKernel<kPolicyNone, true, true>), dim3(h->grid_blocks), dim3(h->block_threads), lds, h->resident_stream, p, off,
                           h->phase1_range);
    OK_HIP(h, hipGetLastError());
    h->resident = true;
    return OKENV_OK;
}

// One step through the resident kernel: the records go into the agents' slots, the sequence number after them; then the
// host waits for the completion word.  *served is false when nobody answered (the kernel had left: idle for too long, e.g.
// because this thread was descheduled): the kernel is drained and the caller redoes the step with a launch of its own, which
// works from the same input records.
int stepResident(okenv *h, const okenv_agent_record *in, volatile uint32_t *slots, const volatile uint32_t *done_word, const bool just_started,
                 bool *served)
{
    if (h->resident_stall_us > 0 && h->resident_steps % 7 == 6)
    { // (tests only) every seventh resident step comes too late
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(h->resident_stall_us);
        while (std::chrono::steady_clock::now() < until)
        {
        }
    }
    const uint32_t seq = nextPackedSeq(h);
    const size_t   N   = static_cast<size_t>(h->N);
    for (size_t i = 0; i < N; ++i)
    { // record word j goes to slot word j + j / 3: words 3, 7, 11, 15 of a slot carry the sequence number
        uint32_t w[sizeof(okenv_agent_record) / 4U];
        std::memcpy(w, &in[i], sizeof(okenv_agent_record));
        for (unsigned j = 0; j < sizeof(okenv_agent_record) / 4U; ++j)
            slots[16U * i + j + j / 3U] = w[j];
    }
    std::atomic_thread_fence(std::memory_order_release); // records before sequence numbers (and x86 keeps store order)
    for (size_t i = 0; i < N; ++i)
        for (unsigned q = 3; q < 16U; q += 4U)
            slots[16U * i + q] = seq;
    bool       answered = false;
    const auto t_asked  = std::chrono::steady_clock::now();
    while (!answered)
    {
        for (int spin = 0; spin < 1024 && !answered; ++spin)
        {
            answered = *done_word == seq;
            if (!answered)
                okCpuRelax();
        }
        if (answered)
            break;
        // (a kernel launched a moment ago may still be on its way -- the first launch of a process loads the code object -- and
        // its patience only starts when it does)
        const double waited_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_asked).count();
        // Giving up on a kernel that is merely slow is safe because a packed step is a pure function of the input records: the
        // resident instantiation (kPacked && kResident) takes ALL agent state from the slot, never advances step_counter[0]
        // (auto-reset makes a handle ineligible) and writes only the exchange buffers and the per-ray arrays, which the redo
        // overwrites with the same values.  Anything added to the resident form that accumulates device state breaks this.
        if (waited_us > (just_started ? 2.0e6 : 2.5 * kResidentGapUs))
        { // ... and only a kernel that has really left is given up on: one that is still on its stream is merely slow (it polls
          // its slot every few hundred nanoseconds, so it has the step) and will answer -- up to a bound, so that a hung device
          // ends in the launch path's error and not in an endless wait
            if (!just_started && waited_us < 20000.0 && hipStreamQuery(h->resident_stream) == hipErrorNotReady)
                continue;
            break;
        }
        if (just_started && waited_us > 1000.0 && hipStreamQuery(h->resident_stream) != hipErrorNotReady)
            break; // its stream has drained (or failed): nobody is going to answer
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    ++h->resident_steps;
    ++h->resident_served_now;
    *served = answered;
    if (!answered)
    {
        ++h->resident_fallbacks;
        return stopResident(h);
    }
    return OKENV_OK;
}

// okenv_step_packed: the step kernel's last workgroup stores the launch's sequence number into mapped host memory once all
// results are there.  Spinning on that word returns about 5 us earlier than hipStreamSynchronize (which waits for the
// queue's completion signal: 10.9 us against 6.0 us for an empty kernel on this machine).  The stream is asked now and
// then, so that a failed launch ends in an error and not in an endless wait.
constexpr size_t kDoneWordBytes = 128;

int waitPackedDone(okenv *h, const volatile uint32_t *word, const uint32_t seq)
{
    for (;;)
    {
        for (int spin = 0; spin < 4096; ++spin)
        {
            if (*word == seq)
            {
                std::atomic_thread_fence(std::memory_order_acquire);
                return OKENV_OK;
            }
            okCpuRelax();
        }
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipErrorNotReady)
            continue;
        if (q != hipSuccess)
            return fail(h, OKENV_ERR_HIP, std::string("okenv_step_packed: ") + hipGetErrorString(q));
        // the stream has drained: the word is there by now, or the kernel never got to write it
        std::atomic_thread_fence(std::memory_order_acquire);
        if (*word == seq)
            return OKENV_OK;
        return fail(h, OKENV_ERR_HIP, "okenv_step_packed: the step kernel finished without announcing it");
    }
}

long tailLimit(const okenv *h, bool q_launch);

// Points a launch at the [front | back] images instead of the combined one (ok_grid.h: okClassifyFrontBack).
void useFrontBack(const okenv *h, OkStepParams &p)
{
    p.image            = h->d_image_fb;
    p.image_bytes      = static_cast<uint32_t>(h->fb_bytes);
    p.off_hdr          = static_cast<uint32_t>(h->fbi.front.off_hdr);
    p.side_tol         = h->fbi.front.side_tol;
    p.fb               = 1U;
    p.fb_back_off      = static_cast<uint32_t>(h->fb_back_off);
    p.fb_back_off_hdr  = static_cast<uint32_t>(h->fb_back_off + h->fbi.back.off_hdr);
    p.fb_back_side_tol = h->fbi.back.side_tol;
    p.fb_e_s           = h->fbc.e_s;
    p.fb_e_t           = h->fbc.e_t;
    p.fb_t12           = h->fbc.t12;
    p.fb_t34           = h->fbc.t34;
}

int launchStep(okenv *h, OkStepParams p) // (by value: the diagnostic build adds its stamp buffer)
{
    OK_HIP(h, hipSetDevice(h->device));
    EventPair ev{};
    int       rc = beginTiming(h, &ev);
    if (rc != OKENV_OK)
        return rc;
    dim3      grid(h->grid_blocks), block(h->block_threads);
    const int policy = p.action_source == kActionsMlpPolicy ? kPolicyMlp : kPolicyNone;
    float phase1 = h->phase1_range;
    if (p.active != nullptr)
    { // an episode's list: the grid covers the listed agents, spread over the CUs like a population of that size
        if (p.action_source == kActionsQLearning)
        { // ... and a list that has become short gets what a population that small gets from okenv_create: wider lane groups
          // whose spare lanes take intervals of the agent's rays, no phase 1 (16 rays, 64 listed agents: 9.3 against 11.8 us
          // per step; the fused MLP's steps gain nothing from it and keep their width)
            int G = h->G;
            while (G < 64 && static_cast<long>(p.n_active) * (2L * G) <= 512L * h->cus) // half of the machine's lanes, as in okenv_create
                G *= 2;
            if (G > h->G)
            {
                p.G    = G;
                phase1 = 0.F;
            }
        }
        const long lanes = static_cast<long>(p.n_active) * p.G;
        long       per   = ((((lanes + h->cus - 1) / h->cus) + 63) / 64) * 64;
        per              = per < 256 ? 256 : (per > 1024 ? 1024 : per);
        block            = dim3(static_cast<unsigned>(per));
        grid             = dim3(static_cast<unsigned>((lanes + per - 1) / per));
    }
#if defined(OKENV_STAMPS)
    { // diagnostic build: stamp space for every wave of THIS launch (the grid differs between populations, lists and forms)
        const size_t waves = static_cast<size_t>(grid.x) * (block.x / 64U);
        if (waves > h->stamp_waves_cap)
        {
            unsigned long long *d = nullptr;
            const int           src = devAlloc(h, &d, waves * kStampWords);
            if (src != OKENV_OK)
                return src;
            h->d_stamps        = d;
            h->stamp_waves_cap = waves;
        }
        h->stamp_waves = waves;
        p.stamps       = h->d_stamps;
    }
#endif
    // The tail of an episode: a short list is stepped one agent per workgroup, every ray cut into eight intervals
    // (okStepTailKernel).  Two such workgroups fit a CU's LDS; beyond about two rounds of them the cooperative kernel's shared
    // waves win again.
    const bool q_launch = p.action_source == kActionsQLearning;
    if (p.active != nullptr && (policy == kPolicyMlp || q_launch) && p.n_active > 0)
    {
        // (Q-learning: one more wave, without rays -- it looks up the nearest centre-line index while the others walk -- and the
        // agent's table in LDS)
        const unsigned lanes = static_cast<unsigned>(((h->R * kTailSplit + 63) / 64) * 64) + (q_launch ? 64U : 0U);
        size_t         lds   = h->image_bytes + 16U + sizeof(float) * kTailLdsFloats + (q_launch ? qLdsBytes(h) + sizeof(float) * kTailQFloats : 0U);
        if (p.n_active <= tailLimit(h, q_launch))
        {
            p.G = h->G; // (unused by the tail kernel; undo the widening above)
            const dim3 tgrid(static_cast<unsigned>(p.n_active)), tblock(lanes);
            uint32_t   off = static_cast<uint32_t>(h->image_bytes);
            // the front / back split while the list fits one round of workgroups with the larger image (fewer of them share a CU);
            // longer lists keep the combined image and their two workgroups per CU
            const size_t lds_fb = lds - h->image_bytes + h->fb_bytes;
            if (h->fb_ok && lds_fb <= kLdsBudget && static_cast<long>(p.n_active) <= static_cast<long>(kLdsBudget / lds_fb) * h->cus)
            {
                useFrontBack(h, p);
                off = static_cast<uint32_t>(h->fb_bytes);
                lds = lds_fb;
            }
#if defined(OKENV_STAMPS)
            { // [0] policy, [1] pre-step, [3] interval walk + min, [5] epilogue, [6] barrier, [7] crash test + Q-learning; [2] / [4] start / end
                const size_t waves = static_cast<size_t>(p.n_active) * (lanes / 64U);
                if (waves > h->stamp_waves_cap)
                {
                    unsigned long long *d = nullptr;
                    const int           src = devAlloc(h, &d, waves * kStampWords);
                    if (src != OKENV_OK)
                        return src;
                    h->d_stamps        = d;
                    h->stamp_waves_cap = waves;
                }
                h->stamp_waves = waves;
                p.stamps       = h->d_stamps;
            }
#endif
            if (q_launch)
                hipLaunchKernelGGL((okStepTailKernel<kPolicyQ, 0>), tgrid, tblock, lds, h->stream, p, off);
            else if (h->R == 32)
                hipLaunchKernelGGL((okStepTailKernel<kPolicyMlp, 32>), tgrid, tblock, lds, h->stream, p, off);
            else if (h->R == 15) // the reference's own fan (Agent.cpp:13-17; EvolutionaryRacer's 17-30-6 network): weights in registers as well
                hipLaunchKernelGGL((okStepTailKernel<kPolicyMlp, 15>), tgrid, tblock, lds, h->stream, p, off);
            else
                hipLaunchKernelGGL((okStepTailKernel<kPolicyMlp, 0>), tgrid, tblock, lds, h->stream, p, off);
            OK_HIP(h, hipGetLastError());
            return endTiming(h, ev);
        }
    }
#define OK_LAUNCH_GENERIC(MODE, LDS)                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        if (policy == kPolicyMlp)                                                                                      \
            hipLaunchKernelGGL((okStepKernel<MODE, kPolicyMlp>), grid, block, LDS, h->stream, p);                      \
        else                                                                                                           \
            hipLaunchKernelGGL((okStepKernel<MODE, kPolicyNone>), grid, block, LDS, h->stream, p);                     \
    } while (0)
    switch (h->grid_mode)
    {
    case kGridLds:
        if (h->coop)
        {
            size_t   lds = coopLdsBytes(h);
            uint32_t off = static_cast<uint32_t>(h->image_bytes);
            // the front / back split, when everything else the launch stages still fits behind it
            const bool has_extra = p.action_source == kActionsQLearning || p.action_source == kActionsController;
            if (h->fb_ok && h->fb_bytes + kCoopLdsExtra + (has_extra ? qLdsBytes(h) : 0U) <= kLdsBudget)
            {
                useFrontBack(h, p);
                off = static_cast<uint32_t>(h->fb_bytes);
                lds = h->fb_bytes + kCoopLdsExtra;
            }
            if (p.action_source == kActionsQLearning)
                hipLaunchKernelGGL(okStepCoopKernel<kPolicyQ>, grid, block, lds + qLdsBytes(h), h->stream, p, off, phase1);
            else if (p.action_source == kActionsController)
            { // the controllers' parameters of a workgroup's agents go into its LDS when they fit behind the centre line
                const size_t base  = ((lds + qLdsBytes(h) + 15U) / 16U) * 16U;
                const size_t stage = static_cast<size_t>(block.x / static_cast<unsigned>(p.G)) * static_cast<size_t>(h->ctrl_num_params) * sizeof(float);
                size_t       total = lds + qLdsBytes(h);
                if (base + stage <= kLdsBudget)
                {
                    p.ctrl_lds_off = static_cast<uint32_t>(base);
                    total          = base + stage;
                }
                hipLaunchKernelGGL(okStepCoopKernel<kPolicyCtrl>, grid, block, total, h->stream, p, off, phase1);
            }
            else if (policy == kPolicyMlp && h->G == 32 && h->R == 32) // C3 / C4's fan: group and fan width compile-time constants
                hipLaunchKernelGGL((okStepCoopKernel<kPolicyMlp, false, false, false, 32>), grid, block, lds, h->stream, p, off, h->phase1_range);
            else if (policy == kPolicyMlp)
                hipLaunchKernelGGL(okStepCoopKernel<kPolicyMlp>, grid, block, lds, h->stream, p, off, h->phase1_range);
            else if (p.rec_in != nullptr && directIntervals(h))
                hipLaunchKernelGGL((okStepCoopKernel<kPolicyNone, true, false, true>), grid, block, lds, h->stream, p, off, h->phase1_range);
            else if (p.rec_in != nullptr)
                hipLaunchKernelGGL((okStepCoopKernel<kPolicyNone, true>), grid, block, lds, h->stream, p, off, h->phase1_range);
            else if (directIntervals(h))
                hipLaunchKernelGGL((okStepCoopKernel<kPolicyNone, false, false, true>), grid, block, lds, h->stream, p, off, h->phase1_range);
            else if (h->G == 64 && p.action_source == kActionsPhiloxReset && p.do_move != 0 && p.reset_flags == 0U)
                // okenv_rollout_random without device-side resetAgent: its launch-time switches as constants (-1 %)
                hipLaunchKernelGGL((okStepCoopKernel<kPolicyNone, false, false, false, 64, true>), grid, block, lds, h->stream, p, off, h->phase1_range);
            else if (h->G == 64) // one agent per wave, the group width a compile-time constant (-1 % on 20-step launches)
                hipLaunchKernelGGL((okStepCoopKernel<kPolicyNone, false, false, false, 64>), grid, block, lds, h->stream, p, off, h->phase1_range);
            else
                hipLaunchKernelGGL(okStepCoopKernel<kPolicyNone>, grid, block, lds, h->stream, p, off, h->phase1_range);
        }
        else
            OK_LAUNCH_GENERIC(kGridLds, h->image_bytes);
        break;
    case kGridGlobal:
        OK_LAUNCH_GENERIC(kGridGlobal, 0);
        break;
    default:
        OK_LAUNCH_GENERIC(kGridBrute, 0);
        break;
    }
#undef OK_LAUNCH_GENERIC
    OK_HIP(h, hipGetLastError());
    return endTiming(h, ev);
}

// Longest episode list the tail kernel (okStepTailKernel) takes on this handle: one round of workgroups -- as many per CU as
// the LDS holds, times the device's compute units (256 on an MI355X in SPX mode; measured there, 32-ray MLP agents: 8.1 us per step up to 256 agents, 9.2 at 512 with two per CU, against
// 11.2-11.8 for the cooperative kernel; a second round loses: 17 us) -- or OKENV_TAIL_MAX_AGENTS; 0: the tail kernel does not apply.
long tailLimit(const okenv *h, const bool q_launch)
{
    if (h->grid_mode != kGridLds || !h->coop || h->tail_max_agents == 0)
        return 0;
    const unsigned lanes = static_cast<unsigned>(((h->R * kTailSplit + 63) / 64) * 64);
    const size_t   lds   = h->image_bytes + 16U + sizeof(float) * kTailLdsFloats + (q_launch ? qLdsBytes(h) + sizeof(float) * kTailQFloats : 0U);
    if (lanes > 512U || lds > kLdsBudget)
        return 0;
    const long fit = static_cast<long>(kLdsBudget / lds) * h->cus;
    return h->tail_max_agents > 0 ? std::min<long>(h->tail_max_agents, fit) : fit;
}

// The step counter also lives on the device (it is the epoch of the auto-reset draws and must advance when a captured
// graph of the step is replayed): okFinishLaunch at the end of the step kernels.
int advanceStepCount(okenv *h, const int n_steps)
{
    h->step_count += static_cast<uint32_t>(n_steps); // the device copy is advanced by the step kernel itself
    return OKENV_OK;
}

int copyAny(okenv *h, void *dst, const void *src, const size_t bytes)
{
    if (bytes == 0)
        return OKENV_OK;
    OK_HIP(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, h->stream));
    return OKENV_OK;
}
} // namespace

extern "C"
{
    const char *okenv_last_error(okenv_t h)
    {
        return h ? h->last_error.c_str() : g_create_error.c_str();
    }

    int okenv_create(okenv_t     *out,
                     const float *segments_xyxy,
                     int32_t      num_segments,
                     int32_t      num_agents,
                     int32_t      num_rays,
                     const float *ray_angles_deg,
                     int32_t      device,
                     uint32_t     flags,
                     float        grid_cell)
    {
        if (!out)
            return fail(nullptr, OKENV_ERR_INVALID, "okenv_create: out is NULL");
        *out = nullptr;
        // TrackSegments asserts num_segments > 0 (TrackSegments.cu:72); CollisionChecker needs >= 1 agent with >= 1 ray
        if (!segments_xyxy || num_segments <= 0)
            return fail(nullptr, OKENV_ERR_INVALID, "okenv_create: need at least one segment");
        if (num_agents <= 0 || num_rays <= 0 || !ray_angles_deg)
            return fail(nullptr, OKENV_ERR_INVALID, "okenv_create: need at least one agent and one ray");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            return fail(nullptr, OKENV_ERR_NO_DEVICE, "okenv_create: no HIP device available (there is no CPU fallback)");
        if (device < 0 || device >= ndev)
            return fail(nullptr, OKENV_ERR_INVALID, "okenv_create: device ordinal out of range");

        // every early return below releases what has been allocated so far (stream, device buffers)
        struct Destroy
        {
            void operator()(okenv *e) const { okenv_destroy(e); }
        };
        std::unique_ptr<okenv, Destroy> hp(new okenv);
        okenv                          *h = hp.get();
        h->device                = device;
        h->N                     = num_agents;
        h->R                     = num_rays;
        h->S                     = num_segments;
        h->flags                 = flags;
        OK_HIP(nullptr, hipSetDevice(device));
        OK_HIP(nullptr, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));

        // lanes per agent: the fan's width rounded up to a power of two -- and more when the population is far too
        // small to fill the machine: the spare lanes of an agent's group take intervals of its rays in phase 2, which
        // shortens the dependent chain of a step (11.5-13 us instead of 15.6 us per step for RL-sized populations).
        // Kept to half of the machine's lanes (compute units x 1024; the device is asked, an MI355X in CPX / DPX partition mode
        // shows 32 / 128 of its 256 CUs) so that the waves of a CU do not start competing for issue.
        {
            int cus = 0;
            OK_HIP(nullptr, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
            h->cus = cus > 0 ? cus : 256;
        }
        h->G = pow2ceil(num_rays) > 64 ? 64 : pow2ceil(num_rays);
        const int natural_g = h->G;
        while (h->G < 64 && static_cast<long>(num_agents) * (2L * h->G) <= 512L * h->cus)
            h->G *= 2;
        if (const char *env_g = std::getenv("OKENV_LANES_PER_AGENT"))
        { // tuning knob: fold the fan over fewer lanes (ray r, r+G, r+2G, ... share a lane) or spread it over more
            const int g = std::atoi(env_g);
            if (g >= 1 && g <= 64 && (g & (g - 1)) == 0)
                h->G = g;
        }
        // with spare lanes there is no phase 1: phase 2 cuts every ray into intervals from its origin on (a 4 px phase 1 in
        // front of it cost a second walk set-up per step: 6.5 -> 5.2 us for one five-ray agent, 8.4 -> 7.0 us at 4096 x 5)
        if (h->G > natural_g)
            h->phase1_range = 0.F;
        h->rays_per_lane = (num_rays + h->G - 1) / h->G;

        // ---- grid ------------------------------------------------------------------------------------
        const OkSeg *segs = reinterpret_cast<const OkSeg *>(segments_xyxy);
        bool         fits = false;
        // cell edge: 24 px when a wave holds one agent (all 64 rays leave one origin), 20 px when it holds several (measured:
        // Silverstone / Spa x 64 rays 4 % faster at 24, Monza x 32 rays 6 % faster at 20)
        // (round 3: 16-ray fans four to a wave -- BASELINE config 5 -- 24 px cells with a 32 px phase 1: 15.6 against 16.4 us per step)
        const bool  narrow16     = h->G == 16 && h->rays_per_lane == 1;
        // (wide fans, one agent per wave: 28 px since the front / back split halved the points per cell -- 9.5-10.0 us per C2 step at
        // 28-30 px against 10.4 at 24 and 10.7 at 20, profiles/r4/front_back_ab.txt; 16-ray fans stay at 24, 32-ray fans at 20)
        const bool  wide64       = h->G == 64 && h->rays_per_lane == 1 && num_rays > 32;
        const float cell_default = wide64 ? 28.F : (narrow16 ? 24.F : OKGRID_DEFAULT_CELL);
        if (narrow16)
            h->phase1_range = 32.F;
        h->grid = okBuildGridAuto(segs, static_cast<size_t>(num_segments), grid_cell > 0.F ? grid_cell : cell_default, kLdsBudget - kLdsReserve,
                                  &fits, &h->poly);
        if (flags & OKENV_FLAG_BRUTE_FORCE)
            h->grid_mode = kGridBrute;
        else if (!fits || (flags & OKENV_FLAG_FORCE_GLOBAL_GRID))
            h->grid_mode = kGridGlobal;
        else
            h->grid_mode = kGridLds;
        // The front / back split (ok_grid.h): the outer boundary polylines in an image of their own, walked only by the rays that
        // need it.  OKENV_FRONT_BACK=0 keeps every launch on the combined image (ablation; same results).
        const char *env_fb = std::getenv("OKENV_FRONT_BACK");
        if (h->grid_mode == kGridLds && (env_fb == nullptr || std::atoi(env_fb) != 0))
        {
            auto classify = [&]() {
                h->fbc = okClassifyFrontBack(segs, static_cast<size_t>(num_segments), h->grid, h->poly.max_seg_len);
                h->fbi = okBuildFrontBackImages(segs, static_cast<size_t>(num_segments), h->grid, h->fbc);
            };
            classify();
            // 32-lane groups (the EvolutionaryRacer shape): most of a generation's steps are taken by the tail kernel, one agent per
            // workgroup, and the drivers hand a list over to it at up to two workgroups per CU -- which the two images allow only when
            // they fit the CU's LDS twice.  With a default cell edge the smallest of 20 / 24 / 28 / 32 px that manages it is taken
            // (larger cells, smaller images; the cooperative kernel's step time is flat over that range: profiles/r4/front_back_ab.txt).
            const size_t tail_extra = 16U + sizeof(float) * kTailLdsFloats;
            auto twice = [&]() { return 2U * (h->fbi.front.bytes.size() + h->fbi.back.bytes.size() + tail_extra) <= kLdsBudget; };
            if (grid_cell <= 0.F && h->G == 32 && h->fbi.ok && !twice())
            {
                const OkGridHost       grid0 = h->grid;
                const OkPolyImage      poly0 = h->poly;
                const OkFrontBack      fbc0  = h->fbc;
                const OkFrontBackImages fbi0 = h->fbi;
                bool                   found = false;
                for (const float c : {24.F, 28.F, 32.F})
                {
                    bool        fits2 = false;
                    OkPolyImage poly2;
                    OkGridHost  grid2 = okBuildGridAuto(segs, static_cast<size_t>(num_segments), c, kLdsBudget - kLdsReserve, &fits2, &poly2);
                    if (!fits2)
                        break;
                    h->grid = grid2;
                    h->poly = poly2;
                    classify();
                    if (h->fbi.ok && twice())
                    {
                        found = true;
                        break;
                    }
                }
                if (!found)
                {
                    h->grid = grid0;
                    h->poly = poly0;
                    h->fbc  = fbc0;
                    h->fbi  = fbi0;
                }
            }
        }

        int rc;
        if ((rc = devAlloc(h, &h->d_segs, static_cast<size_t>(num_segments))) != OKENV_OK)
            return fail(nullptr, rc, h->last_error);
        OK_HIP(nullptr, hipMemcpyAsync(h->d_segs, segs, sizeof(OkSeg) * num_segments, hipMemcpyHostToDevice, h->stream));
        if (h->grid_mode == kGridLds)
        {
            h->image_bytes                  = h->poly.bytes.size();
            const std::vector<uint8_t> &img = h->poly.bytes;
            uint8_t *dimg = nullptr;
            if ((rc = devAlloc(h, &dimg, h->image_bytes)) != OKENV_OK)
                return fail(nullptr, rc, h->last_error);
            h->d_image = dimg;
            OK_HIP(nullptr, hipMemcpyAsync(dimg, img.data(), h->image_bytes, hipMemcpyHostToDevice, h->stream));
            // The front / back split (ok_grid.h; classified before the images were uploaded, above): its two images as one blob.
            if (h->fbi.ok && h->fbi.front.bytes.size() + h->fbi.back.bytes.size() <= kLdsBudget - kLdsReserve)
            {
                h->fb_back_off = h->fbi.front.bytes.size(); // (a multiple of 16: both parts of an image are 16-byte aligned)
                h->fb_bytes    = h->fb_back_off + h->fbi.back.bytes.size();
                if ((rc = devAlloc(h, &h->d_image_fb, h->fb_bytes)) != OKENV_OK)
                    return fail(nullptr, rc, h->last_error);
                OK_HIP(nullptr, hipMemcpyAsync(h->d_image_fb, h->fbi.front.bytes.data(), h->fb_back_off, hipMemcpyHostToDevice, h->stream));
                OK_HIP(nullptr, hipMemcpyAsync(h->d_image_fb + h->fb_back_off, h->fbi.back.bytes.data(), h->fbi.back.bytes.size(), hipMemcpyHostToDevice,
                                               h->stream));
                h->fb_ok = true;
            }
            OK_HIP(nullptr, hipStreamSynchronize(h->stream));
            const int lds_plain = static_cast<int>(h->image_bytes);
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepKernel<kGridLds, kPolicyNone>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_plain));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepKernel<kGridLds, kPolicyMlp>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_plain));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepCoopKernel<kPolicyNone>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepCoopKernel<kPolicyNone, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepCoopKernel<kPolicyNone, true, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepCoopKernel<kPolicyNone, false, false, false, 64, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepCoopKernel<kPolicyNone, false, false, false, 64>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepCoopKernel<kPolicyNone, false, false, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepCoopKernel<kPolicyNone, true, false, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepCoopKernel<kPolicyNone, true, true, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepCoopKernel<kPolicyMlp>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepCoopKernel<kPolicyQ>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepCoopKernel<kPolicyCtrl>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepCoopKernel<kPolicyMlp, false, false, false, 32>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepTailKernel<kPolicyMlp, 32>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepTailKernel<kPolicyMlp, 15>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepTailKernel<kPolicyMlp, 0>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okStepTailKernel<kPolicyQ, 0>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBudget)));
            OK_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(&okDebugCastKernel<kGridLds>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(h->image_bytes)));
        }
        else if (h->grid_mode == kGridGlobal)
        {
            if ((rc = devAlloc(h, &h->d_refs32, h->grid.refs.size())) != OKENV_OK ||
                (rc = devAlloc(h, &h->d_start, h->grid.start.size())) != OKENV_OK)
                return fail(nullptr, rc, h->last_error);
            OK_HIP(nullptr, hipMemcpyAsync(h->d_refs32, h->grid.refs.data(), 4U * h->grid.refs.size(), hipMemcpyHostToDevice, h->stream));
            OK_HIP(nullptr, hipMemcpyAsync(h->d_start, h->grid.start.data(), 4U * h->grid.start.size(), hipMemcpyHostToDevice, h->stream));
        }

        // ---- state -----------------------------------------------------------------------------------
        const size_t N = num_agents, NR = static_cast<size_t>(num_agents) * num_rays;
        auto        &s = h->st;
        if ((rc = devAlloc(h, &s.pos_x, N)) || (rc = devAlloc(h, &s.pos_y, N)) || (rc = devAlloc(h, &s.rot, N)) ||
            (rc = devAlloc(h, &s.speed, N)) || (rc = devAlloc(h, &s.acc, N)) || (rc = devAlloc(h, &s.thr, N)) ||
            (rc = devAlloc(h, &s.steer, N)) || (rc = devAlloc(h, &s.mode, N)) || (rc = devAlloc(h, &s.crashed, N)) ||
            (rc = devAlloc(h, &s.timed_out, N)) || (rc = devAlloc(h, &s.disp_to, N)) || (rc = devAlloc(h, &s.disp_ctr, N)) ||
            (rc = devAlloc(h, &s.disp_x, N)) || (rc = devAlloc(h, &s.disp_y, N)) || (rc = devAlloc(h, &s.hit_x, NR)) ||
            (rc = devAlloc(h, &s.hit_y, NR)) || (rc = devAlloc(h, &s.rel_x, NR)) || (rc = devAlloc(h, &s.rel_y, NR)) ||
            (rc = devAlloc(h, &s.dist, NR)) || (rc = devAlloc(h, &h->d_ray_deg, static_cast<size_t>(num_rays))) ||
            (rc = devAlloc(h, &h->d_step_count, 2))) // [0] steps, [1] finished workgroups of the running launch
            return fail(nullptr, rc, h->last_error);
        OK_HIP(nullptr, hipMemcpyAsync(h->d_ray_deg, ray_angles_deg, 4U * num_rays, hipMemcpyHostToDevice, h->stream));
        h->host_ray_deg.assign(ray_angles_deg, ray_angles_deg + num_rays);

        // ---- launch geometry -------------------------------------------------------------------------
        // Spread small populations over the CUs: aim for a workgroup on every CU before growing them to 1024 lanes.
        const long total_lanes = static_cast<long>(num_agents) * h->G;
        long       per_block   = (total_lanes + h->cus - 1) / h->cus;
        per_block              = ((per_block + 63) / 64) * 64;
        // at least four waves per workgroup: with a handful of agents the launch is dominated by staging the ~70-90 KB track
        // image into LDS, which a single wave does four times slower (waves without an agent leave right after it)
        if (per_block < 256)
            per_block = 256;
        if (per_block > 1024)
            per_block = 1024;
        if (const char *env_bt = std::getenv("OKENV_BLOCK_THREADS"))
        { // tuning knob: smaller workgroups (two per CU when the LDS image allows)
            const long bt = std::atol(env_bt);
            if (bt >= 64 && bt <= 1024 && bt % 64 == 0 && bt % h->G == 0)
                per_block = bt;
        }
        h->block_threads = static_cast<int>(per_block);
        h->grid_blocks   = static_cast<int>((total_lanes + per_block - 1) / per_block);
        h->coop          = h->grid_mode == kGridLds && h->rays_per_lane == 1;
        if (const char *env_coop = std::getenv("OKENV_COOP")) // tuning/ablation knob: 0 = every lane walks its own ray to the end
            h->coop = h->coop && std::atoi(env_coop) != 0;
        // up to one agent per CU with a wave each (the populations of the reference's applications: 1, 15, 30, 50): one agent per
        // workgroup, i.e. per CU -- four such waves on one CU take 7.8 us for a step, one alone 6.0 us -- and three more waves
        // that only help with the staging
        h->agents_per_block = 0;
        if (h->coop && h->G == 64 && num_agents <= h->cus && per_block == 256)
            h->agents_per_block = 1;
        if (const char *env_apb = std::getenv("OKENV_AGENTS_PER_BLOCK"))
        { // tuning knob; 0 = dense
            const int apb = std::atoi(env_apb);
            if (h->coop && apb >= 0 && static_cast<long>(apb) * h->G <= per_block)
                h->agents_per_block = apb;
        }
        if (h->agents_per_block > 0)
            h->grid_blocks = (num_agents + h->agents_per_block - 1) / h->agents_per_block;
        if (const char *env_stall = std::getenv("OKENV_RESIDENT_STALL_US"))
            h->resident_stall_us = std::atoi(env_stall);
        if (const char *env_res = std::getenv("OKENV_RESIDENT")) // 0: never keep the packed-step kernel resident, 1: from the first step on
            h->resident_mode = std::atoi(env_res);
        if (const char *env_tail = std::getenv("OKENV_TAIL_MAX_AGENTS")) // tuning / ablation knob; 0: never use the tail kernel
            h->tail_max_agents = std::atoi(env_tail);
        if (const char *env_t1 = std::getenv("OKENV_PHASE1_RANGE"))
        {
            const float t1 = static_cast<float>(std::atof(env_t1));
            if (t1 >= 0.F) // 0: no phase 1
                h->phase1_range = t1;
        }
        OK_HIP(nullptr, hipStreamSynchronize(h->stream));
        *out = hp.release();
        return OKENV_OK;
    }

    int okenv_destroy(okenv_t h)
    {
        if (!h)
            return OKENV_OK;
        (void)hipSetDevice(h->device);
        if (h->resident)
            (void)stopResident(h);
        if (h->resident_stream)
            (void)hipStreamDestroy(h->resident_stream);
        // a borrowed stream (okenv_set_stream) may already be gone, or be capturing: wait for the device instead of touching it
        if (h->own_stream)
            (void)hipStreamSynchronize(h->stream);
        else
            (void)hipDeviceSynchronize();
        for (void *p : h->allocations)
            (void)hipFree(p);
        if (h->h_stage)
            (void)hipHostFree(h->h_stage);
        for (auto &e : h->events)
        {
            (void)hipEventDestroy(e.start);
            (void)hipEventDestroy(e.stop);
        }
        for (auto &e : h->event_pool)
        {
            (void)hipEventDestroy(e.start);
            (void)hipEventDestroy(e.stop);
        }
        if (h->own_stream && h->stream)
            (void)hipStreamDestroy(h->stream);
        delete h;
        return OKENV_OK;
    }

    int okenv_get_info(okenv_t h, okenv_info *out)
    {
        if (!h || !out)
            return fail(h, OKENV_ERR_INVALID, "okenv_get_info: NULL argument");
        out->num_agents      = h->N;
        out->num_rays        = h->R;
        out->num_segments    = h->S;
        out->compute_units   = h->cus;
        out->front_back_bytes = h->fb_ok ? static_cast<int32_t>(h->fb_bytes) : 0;
        out->back_segments    = h->fb_ok ? static_cast<int32_t>(h->fbc.n_back) : 0;
        out->grid_nx         = h->grid.g.nx;
        out->grid_ny         = h->grid.g.ny;
        out->grid_cell       = h->grid.g.cell;
        out->grid_refs       = static_cast<int32_t>(h->grid.refs.size());
        out->grid_in_lds     = h->grid_mode == kGridLds ? 1 : 0;
        out->lds_bytes       = h->grid_mode == kGridLds ? static_cast<int32_t>(h->image_bytes) : 0;
        out->block_threads   = h->block_threads;
        out->grid_blocks     = h->grid_blocks;
        out->agents_per_block      = h->coop ? h->agents_per_block : 0;
        out->packed_resident       = h->resident ? 1 : 0;
        out->packed_resident_steps = h->resident_steps;
        out->packed_fallbacks      = h->resident_fallbacks;
        out->lanes_per_agent = h->G;
        out->device          = h->device;
        return OKENV_OK;
    }

    int okenv_set_sensor_offset(okenv_t h, float offset)
    {
        if (!h)
            return OKENV_ERR_INVALID;
        if (offset == h->sensor_offset)
            return OKENV_OK; // the C++ facade says it before every step: not a reason to stop a resident step kernel
        OK_QUIESCE(h);
        h->sensor_offset = offset;
        return OKENV_OK;
    }

    int okenv_set_centerline(okenv_t h, const float *x, const float *y, const float *heading_deg, int32_t num_points)
    {
        OK_QUIESCE(h);
        if (!h || !x || !y || !heading_deg || num_points <= 0)
            return fail(h, OKENV_ERR_INVALID, "okenv_set_centerline: bad argument");
        OK_HIP(h, hipSetDevice(h->device));
        int rc;
        if (num_points > h->centerline_capacity)
        { // grow only: repeated calls with the same track reuse the buffers
            if ((rc = devAlloc(h, &h->d_cx, static_cast<size_t>(num_points))) || (rc = devAlloc(h, &h->d_cy, static_cast<size_t>(num_points))) ||
                (rc = devAlloc(h, &h->d_chead, static_cast<size_t>(num_points))))
                return rc;
            h->centerline_capacity = num_points;
        }
        h->P        = num_points;
        h->cl_dirty = true;
        h->host_cx.resize(num_points), h->host_cy.resize(num_points), h->host_chead.resize(num_points);
        OK_HIP(h, hipMemcpy(h->host_cx.data(), x, 4U * num_points, hipMemcpyDefault));
        OK_HIP(h, hipMemcpy(h->host_cy.data(), y, 4U * num_points, hipMemcpyDefault));
        OK_HIP(h, hipMemcpy(h->host_chead.data(), heading_deg, 4U * num_points, hipMemcpyDefault));
        OK_HIP(h, hipMemcpyAsync(h->d_cx, x, 4U * num_points, hipMemcpyDefault, h->stream));
        OK_HIP(h, hipMemcpyAsync(h->d_cy, y, 4U * num_points, hipMemcpyDefault, h->stream));
        OK_HIP(h, hipMemcpyAsync(h->d_chead, heading_deg, 4U * num_points, hipMemcpyDefault, h->stream));
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_set_stream(okenv_t h, void *hip_stream)
    {
        OK_QUIESCE(h);
        if (!h)
            return OKENV_ERR_INVALID;
        // no synchronisation here: the call must be legal while the new stream is being captured into a graph; work
        // already queued on the old stream completes on its own (hipStreamDestroy defers), ordering is the caller's
        if (h->own_stream && h->stream)
        {
            // timing events recorded on the stream about to be destroyed are resolved first (nothing can be captured on a
            // stream this handle owns, so waiting for them is legal here) and carried into the next okenv_get_timing
            for (auto &e : h->events)
            {
                float ms = 0.F;
                if (hipEventSynchronize(e.stop) == hipSuccess && hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess)
                {
                    h->timing_carry_ms += ms;
                    ++h->timing_carry_n;
                }
                h->event_pool.push_back(e);
            }
            h->events.clear();
            (void)hipStreamDestroy(h->stream);
        }
        h->stream     = static_cast<hipStream_t>(hip_stream);
        h->own_stream = false;
        return OKENV_OK;
    }

    int okenv_sync(okenv_t h)
    {
        OK_QUIESCE(h);
        if (!h)
            return OKENV_ERR_INVALID;
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_set_field(okenv_t h, int32_t field, const void *src)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h || !src)
            return fail(h, OKENV_ERR_INVALID, "okenv_set_field: NULL argument");
        const FieldDesc d = fieldOf(h, field);
        if (!d.ptr)
            return fail(h, OKENV_ERR_INVALID, "okenv_set_field: unknown field (tracker fields exist after okenv_tracker_create)");
        int rc = copyAny(h, d.ptr, src, d.bytes);
        if (rc != OKENV_OK)
            return rc;
        OK_HIP(h, hipStreamSynchronize(h->stream)); // the caller may reuse src immediately
        return OKENV_OK;
    }

    int okenv_get_field(okenv_t h, int32_t field, void *dst)
    {
        OK_QUIESCE(h);
        if (!h || !dst)
            return fail(h, OKENV_ERR_INVALID, "okenv_get_field: NULL argument");
        const FieldDesc d = fieldOf(h, field);
        if (!d.ptr)
            return fail(h, OKENV_ERR_INVALID, "okenv_get_field: unknown field (tracker fields exist after okenv_tracker_create)");
        int rc = copyAny(h, dst, d.ptr, d.bytes);
        if (rc != OKENV_OK)
            return rc;
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    static int moveState(okenv_t h, const okenv_state_view *v, const bool upload)
    {
        if (!h || !v)
            return fail(h, OKENV_ERR_INVALID, "state view is NULL");
        const size_t N = h->N;
        auto        &s = h->st;
        struct Item
        {
            void  *dev;
            void  *host;
            size_t bytes;
        };
        const Item items[] = {{s.pos_x, v->pos_x, 4 * N},       {s.pos_y, v->pos_y, 4 * N},     {s.rot, v->rot, 4 * N},
                              {s.speed, v->speed, 4 * N},       {s.acc, v->acc, 4 * N},         {s.thr, v->throttle, 4 * N},
                              {s.steer, v->steer, 4 * N},       {s.mode, v->mode, N},           {s.crashed, v->crashed, N},
                              {s.timed_out, v->timed_out, N},   {s.disp_ctr, v->disp_ctr, 4 * N}, {s.disp_x, v->disp_x, 4 * N},
                              {s.disp_y, v->disp_y, 4 * N},     {s.disp_to, v->disp_timed_out, N}};
        for (const Item &it : items)
        {
            if (!it.host)
                continue;
            const int rc = upload ? copyAny(h, it.dev, it.host, it.bytes) : copyAny(h, it.host, it.dev, it.bytes);
            if (rc != OKENV_OK)
                return rc;
        }
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_upload_state(okenv_t h, const okenv_state_view *host_view)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        return moveState(h, host_view, true);
    }

    int okenv_download_state(okenv_t h, const okenv_state_view *host_view)
    {
        OK_QUIESCE(h);
        return moveState(h, host_view, false);
    }

    int okenv_set_actions(okenv_t h, const float *throttle, const float *steer)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h || !throttle || !steer)
            return fail(h, OKENV_ERR_INVALID, "okenv_set_actions: NULL argument");
        int rc;
        if ((rc = copyAny(h, h->st.thr, throttle, 4U * h->N)) || (rc = copyAny(h, h->st.steer, steer, 4U * h->N)))
            return rc;
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_reset_agents(okenv_t h, const int32_t *idx, const float *x, const float *y, const float *rot_deg, int32_t n)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h || n < 0 || (n > 0 && (!idx || !x || !y || !rot_deg)))
            return fail(h, OKENV_ERR_INVALID, "okenv_reset_agents: bad argument");
        if (n == 0)
            return OKENV_OK;
        OK_HIP(h, hipSetDevice(h->device));
        // one staging buffer: idx | x | y | rot
        const size_t bytes = static_cast<size_t>(n) * 16U;
        void        *stage = nullptr;
        const int    src   = deviceScratch(h, bytes, &stage);
        if (src != OKENV_OK)
            return src;
        char *b = static_cast<char *>(stage);
        OK_HIP(h, hipMemcpyAsync(b, idx, 4U * n, hipMemcpyHostToDevice, h->stream));
        OK_HIP(h, hipMemcpyAsync(b + 4U * n, x, 4U * n, hipMemcpyHostToDevice, h->stream));
        OK_HIP(h, hipMemcpyAsync(b + 8U * n, y, 4U * n, hipMemcpyHostToDevice, h->stream));
        OK_HIP(h, hipMemcpyAsync(b + 12U * n, rot_deg, 4U * n, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(okResetKernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->st, reinterpret_cast<const int32_t *>(b),
                           reinterpret_cast<const float *>(b + 4U * n), reinterpret_cast<const float *>(b + 8U * n),
                           reinterpret_cast<const float *>(b + 12U * n), n, h->N);
        OK_HIP(h, hipGetLastError());
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_set_lane_bounds(okenv_t h, const float *left_inner_xy, const float *right_inner_xy, int32_t num_points)
    {
        OK_QUIESCE(h);
        if (!h || !left_inner_xy || !right_inner_xy || num_points <= 0)
            return fail(h, OKENV_ERR_INVALID, "okenv_set_lane_bounds: bad argument");
        OK_HIP(h, hipSetDevice(h->device));
        if (num_points > h->lane_capacity)
        {
            int rc;
            if ((rc = devAlloc(h, &h->d_lane_l, 2U * static_cast<size_t>(num_points))) ||
                (rc = devAlloc(h, &h->d_lane_r, 2U * static_cast<size_t>(num_points))))
                return rc;
            h->lane_capacity = num_points;
        }
        h->lane_points = num_points;
        OK_HIP(h, hipMemcpyAsync(h->d_lane_l, left_inner_xy, 8U * num_points, hipMemcpyDefault, h->stream));
        OK_HIP(h, hipMemcpyAsync(h->d_lane_r, right_inner_xy, 8U * num_points, hipMemcpyDefault, h->stream));
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    // shared precondition of the two resetAgent entry points
    static int checkResetInputs(okenv_t h, const uint32_t flags, const char *who)
    {
        if (h->P <= 0)
            return fail(h, OKENV_ERR_STATE, std::string(who) + ": call okenv_set_centerline first");
        if ((flags & OK_RESET_RANDOM_POINT) == 0U && h->P <= static_cast<int>(OK_RESET_START_IDX))
            return fail(h, OKENV_ERR_STATE, std::string(who) + ": the centre line is shorter than RaceTrack::kStartingIdx");
        if ((flags & OK_RESET_RANDOM_POINT) != 0U && (flags & OK_RESET_RANDOM_LANE) != 0U && h->lane_points != h->P)
            return fail(h, OKENV_ERR_STATE, std::string(who) + ": lane randomisation needs okenv_set_lane_bounds with as many points as the centre line");
        return OKENV_OK;
    }

    int okenv_reset_random(okenv_t h, const int32_t *idx, int32_t n, uint32_t flags, uint32_t seed, uint32_t epoch, uint32_t agent_base)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h || n < 0)
            return fail(h, OKENV_ERR_INVALID, "okenv_reset_random: bad argument");
        if (!idx)
            n = h->N;
        if (n == 0)
            return OKENV_OK;
        int rc = checkResetInputs(h, flags, "okenv_reset_random");
        if (rc != OKENV_OK)
            return rc;
        OK_HIP(h, hipSetDevice(h->device));
        int32_t *didx = nullptr;
        if (idx)
        {
            void     *sp  = nullptr;
            const int src = deviceScratch(h, 4U * static_cast<size_t>(n), &sp);
            if (src != OKENV_OK)
                return src;
            didx = static_cast<int32_t *>(sp);
            OK_HIP(h, hipMemcpyAsync(didx, idx, 4U * static_cast<size_t>(n), hipMemcpyDefault, h->stream));
        }
        hipLaunchKernelGGL(okResetRandomKernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->st, didx, n, h->N, flags, seed, epoch,
                           agent_base, h->d_cx, h->d_cy, h->d_chead, h->d_lane_l, h->d_lane_r, h->P);
        OK_HIP(h, hipGetLastError());
        if (didx)
            OK_HIP(h, hipStreamSynchronize(h->stream)); // the caller may reuse idx, the next call the staging space
        return OKENV_OK;
    }

    int okenv_set_auto_reset(okenv_t h, int32_t enabled, uint32_t flags, uint32_t seed, uint32_t agent_base)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h)
            return OKENV_ERR_INVALID;
        uint32_t count = 0;
        int      rc    = okenv_get_step_count(h, &count); // folds the device copy back while it is still authoritative
        if (rc != OKENV_OK)
            return rc;
        if (!enabled)
        {
            h->reset_flags = 0;
            return OKENV_OK;
        }
        if ((rc = checkResetInputs(h, flags, "okenv_set_auto_reset")) != OKENV_OK)
            return rc;
        if ((rc = okenv_set_step_count(h, count)) != OKENV_OK)
            return rc;
        h->reset_flags      = (flags & (OK_RESET_RANDOM_POINT | OK_RESET_RANDOM_LANE | OK_RESET_RANDOM_HEADING)) | kAutoResetOn;
        h->reset_seed       = seed;
        h->reset_agent_base = agent_base;
        return OKENV_OK;
    }

    int okenv_get_step_count(okenv_t h, uint32_t *out)
    {
        OK_QUIESCE(h);
        if (!h || !out)
            return fail(h, OKENV_ERR_INVALID, "okenv_get_step_count: NULL argument");
        if ((h->reset_flags & kAutoResetOn) != 0U)
        {
            OK_HIP(h, hipSetDevice(h->device));
            OK_HIP(h, hipMemcpyAsync(&h->step_count, h->d_step_count, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
            OK_HIP(h, hipStreamSynchronize(h->stream));
        }
        *out = h->step_count;
        return OKENV_OK;
    }

    int okenv_set_step_count(okenv_t h, uint32_t value)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h)
            return OKENV_ERR_INVALID;
        h->step_count = value;
        OK_HIP(h, hipSetDevice(h->device));
        OK_HIP(h, hipMemcpyAsync(h->d_step_count, &h->step_count, sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_field_device_ptr(okenv_t h, int32_t field, void **ptr, uint64_t *bytes)
    {
        OK_QUIESCE(h);
        if (!h || !ptr)
            return fail(h, OKENV_ERR_INVALID, "okenv_field_device_ptr: NULL argument");
        const FieldDesc d = fieldOf(h, field);
        if (!d.ptr)
            return fail(h, OKENV_ERR_INVALID, "okenv_field_device_ptr: unknown field (tracker fields exist after okenv_tracker_create)");
        *ptr = d.ptr;
        if (bytes)
            *bytes = d.bytes;
        return OKENV_OK;
    }

    int okenv_get_hits(okenv_t h, float *out_xy)
    {
        OK_QUIESCE(h);
        if (!h || !out_xy)
            return fail(h, OKENV_ERR_INVALID, "okenv_get_hits: NULL argument");
        const size_t       NR = static_cast<size_t>(h->N) * h->R;
        std::vector<float> rx(NR), ry(NR);
        int                rc;
        if ((rc = copyAny(h, rx.data(), h->st.rel_x, 4U * NR)) || (rc = copyAny(h, ry.data(), h->st.rel_y, 4U * NR)))
            return rc;
        OK_HIP(h, hipStreamSynchronize(h->stream));
        for (size_t k = 0; k < NR; ++k)
        {
            out_xy[2 * k]     = rx[k];
            out_xy[2 * k + 1] = ry[k];
        }
        return OKENV_OK;
    }

    int okenv_get_distances(okenv_t h, float *out)
    {
        OK_QUIESCE(h);
        return okenv_get_field(h, OKENV_F_DIST, out);
    }

    int okenv_get_flags(okenv_t h, uint8_t *out)
    {
        OK_QUIESCE(h);
        if (!h || !out)
            return fail(h, OKENV_ERR_INVALID, "okenv_get_flags: NULL argument");
        std::vector<uint8_t> c(h->N), t(h->N);
        int                  rc;
        if ((rc = copyAny(h, c.data(), h->st.crashed, h->N)) || (rc = copyAny(h, t.data(), h->st.timed_out, h->N)))
            return rc;
        OK_HIP(h, hipStreamSynchronize(h->stream));
        for (int i = 0; i < h->N; ++i)
            out[i] = static_cast<uint8_t>((c[i] ? 1 : 0) | (t[i] ? 2 : 0));
        return OKENV_OK;
    }

    int okenv_step(okenv_t h, int32_t n_steps)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h || n_steps < 0)
            return fail(h, OKENV_ERR_INVALID, "okenv_step: bad argument");
        if (n_steps == 0)
            return OKENV_OK;
        OkStepParams p = baseParams(h);
        p.n_steps      = n_steps;
        const int rc   = launchStep(h, p);
        return rc == OKENV_OK ? advanceStepCount(h, n_steps) : rc;
    }

    int okenv_step_packed(okenv_t h, const okenv_agent_record *in, okenv_agent_record *out, float *sensor_hits_xy, uint32_t flags)
    {
        if (h)
            dropEpisode(h);
        if (!h || !in || !out || !sensor_hits_xy)
            return fail(h, OKENV_ERR_INVALID, "okenv_step_packed: NULL argument");
        OK_HIP(h, hipSetDevice(h->device));
        const size_t N = static_cast<size_t>(h->N), NR = N * static_cast<size_t>(h->R);
        const size_t rec_bytes = N * sizeof(okenv_agent_record), hit_bytes = NR * 2U * sizeof(float);
        // mapped host memory: [ in records | out records | hits | completion word | one 64-byte slot per agent (resident kernel) ]
        const size_t out_off   = (rec_bytes + 255U) & ~static_cast<size_t>(255U);
        const size_t hit_off   = 2U * out_off;
        const size_t done_off  = (hit_off + hit_bytes + 63U) & ~static_cast<size_t>(63U);
        const size_t slots_off = done_off + kDoneWordBytes;
        const size_t stage_all = slots_off + 64U * N;
        if (!h->h_stage)
        { // both buffers are committed together: a failed second allocation must not leave a half-initialised pair behind
            void *pinned = nullptr, *mapped = nullptr;
            // coherent (fine-grained): the device's stores go straight to the host's memory, in order with the completion word
            OK_HIP(h, hipHostMalloc(&pinned, stage_all, hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(pinned, 0, stage_all);
            uint8_t *d = nullptr;
            if (hipHostGetDevicePointer(&mapped, pinned, 0) != hipSuccess || devAlloc(h, &d, hit_off + hit_bytes) != OKENV_OK)
            {
                (void)hipHostFree(pinned);
                return fail(h, OKENV_ERR_HIP, "okenv_step_packed: cannot allocate the exchange buffers");
            }
            h->h_stage         = pinned;
            h->h_stage_device  = mapped;
            h->d_stage         = d;
            h->stage_slots_off = slots_off;
        }
        uint8_t *hs = static_cast<uint8_t *>(h->h_stage), *ds = static_cast<uint8_t *>(h->d_stage);
        if (h->grid_mode == kGridLds && h->coop)
        { // ONE kernel: it reads the records from, and writes records and sensor_hits_ to, the mapped host buffer
            uint8_t     *hm = static_cast<uint8_t *>(h->h_stage_device);
            OkStepParams p  = baseParams(h);
            p.rec_in         = reinterpret_cast<const okenv_agent_record *>(hm);
            p.rec_out        = reinterpret_cast<okenv_agent_record *>(hm + out_off);
            p.hits_xy_out    = reinterpret_cast<float *>(hm + hit_off);
            p.rec_with_stats = (flags & OKENV_PACKED_WITH_STATS) ? 1 : 0;
            p.done_flag      = reinterpret_cast<uint32_t *>(hm + done_off); // behind the hits, on a cache line of its own
            const volatile uint32_t *done_word = reinterpret_cast<const volatile uint32_t *>(hs + done_off);

            // steps that follow each other closely are served by a resident kernel (see startResident)
            const auto   t_in  = std::chrono::steady_clock::now();
            const bool   quick = h->packed_seq != 0U && std::chrono::duration<double, std::micro>(t_in - h->packed_last_end).count() < kResidentGapUs;
            const bool eligible = h->resident_mode != 0 && h->agents_per_block == 1 && h->N <= kResidentMaxAgents && h->own_stream && !h->timing &&
                                  (h->reset_flags & kAutoResetOn) == 0U && flags == OKENV_PACKED_WITH_STATS;
            // (a run of quick steps counts only while every one of them is of the kind the resident kernel serves: a caller that
            // alternates step() and checkCollision() must not start and stop a kernel on every other call)
            h->packed_streak = (quick && eligible) ? h->packed_streak + 1 : 0;
            if (h->resident && (!eligible || !quick))
            { // too long since the last step (the kernel may have left by itself), or a kind of step it does not serve
                const int src = stopResident(h);
                if (src != OKENV_OK)
                    return src;
            }
            const bool was_resident = h->resident;
            if (!h->resident && eligible && (h->resident_mode == 1 || h->packed_streak >= h->resident_need))
            {
                p.slots          = reinterpret_cast<const uint32_t *>(hm + slots_off);
                p.idle_ticks     = kResidentIdleTicks;
                const int src = startResident(h, p, reinterpret_cast<volatile uint32_t *>(hs + slots_off));
                if (src != OKENV_OK)
                    return src;
            }
            bool served = false;
            if (h->resident)
            {
                const int src = stepResident(h, in, reinterpret_cast<volatile uint32_t *>(hs + slots_off), done_word, !was_resident, &served);
                if (src != OKENV_OK)
                    return src;
            }
            if (!served)
            {
                std::memcpy(hs, in, rec_bytes);
                p.slots    = nullptr;
                p.done_seq = nextPackedSeq(h);
                if (flags & OKENV_PACKED_COLLIDE_ONLY)
                {
                    p.do_move     = 0;
                    p.reset_flags = 0;
                }
                const int rc = launchStep(h, p);
                if (rc != OKENV_OK)
                    return rc;
                const int wrc = waitPackedDone(h, done_word, p.done_seq);
                if (wrc != OKENV_OK)
                    return wrc;
            }
            if ((flags & OKENV_PACKED_COLLIDE_ONLY) == 0U)
                advanceStepCount(h, 1);
            const okenv_agent_record *src = reinterpret_cast<const okenv_agent_record *>(hs + out_off);
            for (size_t i = 0; i < N; ++i)
            {
                okenv_agent_record r = src[i];
                if ((flags & OKENV_PACKED_WITH_STATS) == 0U)
                { // the caller's DisplacementStats members stay as they were (`out` may alias `in`: read before writing)
                    r.disp_x         = in[i].disp_x;
                    r.disp_y         = in[i].disp_y;
                    r.disp_ctr       = in[i].disp_ctr;
                    r.disp_timed_out = in[i].disp_timed_out;
                }
                out[i] = r;
            }
            std::memcpy(sensor_hits_xy, hs + hit_off, hit_bytes);
            h->packed_last_end = std::chrono::steady_clock::now();
            return OKENV_OK;
        }
        std::memcpy(hs, in, rec_bytes);
        OK_HIP(h, hipMemcpyAsync(ds, hs, rec_bytes, hipMemcpyHostToDevice, h->stream));
        const unsigned blocks_n = static_cast<unsigned>((N + 255U) / 256U);
        hipLaunchKernelGGL(okUnpackRecordsKernel, dim3(blocks_n), dim3(256), 0, h->stream, h->st,
                           reinterpret_cast<const okenv_agent_record *>(ds), h->N, (flags & OKENV_PACKED_WITH_STATS) ? 1 : 0);
        OkStepParams p = baseParams(h);
        if (flags & OKENV_PACKED_COLLIDE_ONLY)
        {
            p.do_move     = 0;
            p.reset_flags = 0;
        }
        int rc = launchStep(h, p);
        if (rc != OKENV_OK)
            return rc;
        if ((flags & OKENV_PACKED_COLLIDE_ONLY) == 0U)
            advanceStepCount(h, 1);
        const size_t threads = NR > N ? NR : N;
        hipLaunchKernelGGL(okPackRecordsKernel, dim3(static_cast<unsigned>((threads + 255U) / 256U)), dim3(256), 0, h->stream, h->st,
                           reinterpret_cast<okenv_agent_record *>(ds + out_off), reinterpret_cast<float *>(ds + hit_off), h->N, h->R);
        OK_HIP(h, hipGetLastError());
        OK_HIP(h, hipMemcpyAsync(hs + out_off, ds + out_off, (hit_off - out_off) + hit_bytes, hipMemcpyDeviceToHost, h->stream));
        OK_HIP(h, hipStreamSynchronize(h->stream));
        if ((flags & OKENV_PACKED_WITH_STATS) != 0U)
            std::memcpy(out, hs + out_off, rec_bytes);
        else
        { // the caller's DisplacementStats members stay as they were
            const okenv_agent_record *src = reinterpret_cast<const okenv_agent_record *>(hs + out_off);
            for (size_t i = 0; i < N; ++i)
            {
                okenv_agent_record r = src[i];
                r.disp_x             = in[i].disp_x;
                r.disp_y             = in[i].disp_y;
                r.disp_ctr           = in[i].disp_ctr;
                r.disp_timed_out     = in[i].disp_timed_out;
                out[i]               = r;
            }
        }
        std::memcpy(sensor_hits_xy, hs + hit_off, hit_bytes);
        return OKENV_OK;
    }

    int okenv_collide(okenv_t h)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h)
            return OKENV_ERR_INVALID;
        OkStepParams p = baseParams(h);
        p.do_move      = 0;
        p.reset_flags  = 0;
        return launchStep(h, p);
    }

    int okenv_rollout_random(okenv_t h, int32_t n_steps, uint32_t seed, uint32_t agent_base, uint32_t step_base)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h || n_steps < 0)
            return fail(h, OKENV_ERR_INVALID, "okenv_rollout_random: bad argument");
        if (h->P <= 0)
            return fail(h, OKENV_ERR_STATE, "okenv_rollout_random: call okenv_set_centerline first");
        if (n_steps == 0)
            return OKENV_OK;
        OkStepParams p  = baseParams(h);
        p.n_steps       = n_steps;
        p.action_source = kActionsPhiloxReset;
        p.reset_flags   = 0; // this driver re-places crashed agents itself
        p.seed          = seed;
        p.agent_base    = agent_base;
        p.step_base     = step_base;
        return launchStep(h, p);
    }

    int okenv_init_bench_state(okenv_t h, uint32_t agent_base, int32_t mode)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h)
            return OKENV_ERR_INVALID;
        if (h->P <= 0)
            return fail(h, OKENV_ERR_STATE, "okenv_init_bench_state: call okenv_set_centerline first");
        OK_HIP(h, hipSetDevice(h->device));
        hipLaunchKernelGGL(okInitBenchKernel, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, h->st, h->d_cx, h->d_cy, h->d_chead,
                           h->P, h->N, h->R, agent_base, mode);
        OK_HIP(h, hipGetLastError());
        return OKENV_OK;
    }

    int okenv_nearest_track_idx(okenv_t h, const float *qx, const float *qy, int32_t n, int32_t *out)
    {
        OK_QUIESCE(h);
        if (!h || !out)
            return fail(h, OKENV_ERR_INVALID, "okenv_nearest_track_idx: NULL argument");
        if (h->P <= 0)
            return fail(h, OKENV_ERR_STATE, "okenv_nearest_track_idx: call okenv_set_centerline first");
        OK_HIP(h, hipSetDevice(h->device));
        const bool agents = (qx == nullptr);
        if (!agents && !qy) // every argument is validated before anything is allocated
            return fail(h, OKENV_ERR_INVALID, "okenv_nearest_track_idx: qy is NULL");
        if (agents)
            n = h->N;
        if (n <= 0)
            return OKENV_OK;
        void     *sp  = nullptr; // [ out: n x i32 | queries: 2n x f32 ]
        const int src = deviceScratch(h, 12U * static_cast<size_t>(n), &sp);
        if (src != OKENV_OK)
            return src;
        int32_t     *dout = static_cast<int32_t *>(sp);
        float       *dq   = reinterpret_cast<float *>(dout + n);
        const float *dqx = h->st.pos_x, *dqy = h->st.pos_y;
        if (!agents)
        {
            OK_HIP(h, hipMemcpyAsync(dq, qx, 4U * n, hipMemcpyDefault, h->stream));
            OK_HIP(h, hipMemcpyAsync(dq + n, qy, 4U * n, hipMemcpyDefault, h->stream));
            dqx = dq;
            dqy = dq + n;
        }
        hipLaunchKernelGGL(okNearestIdxKernel, dim3((n * kNearestLanes + 255) / 256), dim3(256), 0, h->stream, h->d_cx, h->d_cy, h->P, dqx, dqy, n,
                           dout);
        OK_HIP(h, hipGetLastError());
        OK_HIP(h, hipMemcpyAsync(out, dout, 4U * n, hipMemcpyDefault, h->stream));
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    // ---- rollout bookkeeping ---------------------------------------------------------------------------------------

    int okenv_tracker_create(okenv_t h, int32_t reward_kind)
    {
        OK_QUIESCE(h);
        if (!h || (reward_kind != OKENV_REWARD_STEP && reward_kind != OKENV_REWARD_PROGRESS))
            return fail(h, OKENV_ERR_INVALID, "okenv_tracker_create: unknown reward kind");
        if (h->P <= 0)
            return fail(h, OKENV_ERR_STATE, "okenv_tracker_create: call okenv_set_centerline first");
        OK_HIP(h, hipSetDevice(h->device));
        {
            const size_t N = static_cast<size_t>(h->N);
            int          rc;
            if ((rc = devEnsure(h, &h->tracker.prev_idx, N)) || (rc = devEnsure(h, &h->tracker.fitness, N)) ||
                (rc = devEnsure(h, &h->tracker.reward, N)) || (rc = devEnsure(h, &h->tracker.ep_steps, N)) ||
                (rc = devEnsure(h, &h->tracker.ep_return, N)) || (rc = devEnsure(h, &h->tracker.prev_crashed, N)))
                return rc;
        }
        h->tracker_kind = reward_kind;
        return OKENV_OK;
    }

    static int launchTracker(okenv_t h, const int begin, const char *who)
    {
        if (!h)
            return OKENV_ERR_INVALID;
        if (h->tracker_kind < 0)
            return fail(h, OKENV_ERR_STATE, std::string(who) + ": call okenv_tracker_create first");
        OK_HIP(h, hipSetDevice(h->device));
        const long threads = static_cast<long>(h->N) * (h->tracker_kind == kRewardProgress ? kNearestLanes : 1);
        hipLaunchKernelGGL(okTrackerKernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, h->stream, h->st, h->d_cx, h->d_cy,
                           h->P, h->tracker, h->N, h->tracker_kind, begin);
        OK_HIP(h, hipGetLastError());
        return OKENV_OK;
    }

    // Inside a controller episode the fused rollout carries the bookkeeping in registers and writes it back per launch: a tracker
    // call or new parameters from outside would be double counted, lost or half applied -- refused, loudly.
    static bool ctrlEpisodeRunning(okenv_t h)
    {
        return h != nullptr && h->episode && h->ep_kind == kPolicyCtrl;
    }

    int okenv_tracker_begin(okenv_t h)
    {
        OK_QUIESCE(h);
        if (ctrlEpisodeRunning(h))
            return fail(h, OKENV_ERR_STATE, "okenv_tracker_begin: a controller episode is running (okenv_rollout_controller does the bookkeeping); okenv_episode_end first");
        return launchTracker(h, 1, "okenv_tracker_begin");
    }

    int okenv_tracker_update(okenv_t h)
    {
        OK_QUIESCE(h);
        if (ctrlEpisodeRunning(h))
            return fail(h, OKENV_ERR_STATE, "okenv_tracker_update: a controller episode is running (okenv_rollout_controller does the bookkeeping); okenv_episode_end first");
        return launchTracker(h, 0, "okenv_tracker_update");
    }

    // ---- CMA-ES controller -----------------------------------------------------------------------------------------

    int okenv_controller_create(okenv_t h, int32_t hidden)
    {
        OK_QUIESCE(h);
        if (!h || hidden < 2 || hidden > OK_CTRL_MAX_HIDDEN || (hidden & 1) != 0)
            return fail(h, OKENV_ERR_INVALID, "okenv_controller_create: hidden width must be even and in [2, 64]");
        if (h->R > 64)
            return fail(h, OKENV_ERR_INVALID, "okenv_controller_create: at most 64 rays");
        OK_HIP(h, hipSetDevice(h->device));
        const int np = ok_controller_num_params(h->R, hidden, 2);
        if (!h->d_ctrl_params || np != h->ctrl_num_params)
        {
            const int rc = devAlloc(h, &h->d_ctrl_params, static_cast<size_t>(h->N) * np);
            if (rc != OKENV_OK)
                return rc;
        }
        h->ctrl_hidden     = hidden;
        h->ctrl_num_params = np;
        return OKENV_OK;
    }

    int okenv_controller_num_params(okenv_t h, int32_t *out)
    {
        OK_QUIESCE(h);
        if (!h || !out || !h->d_ctrl_params)
            return fail(h, OKENV_ERR_STATE, "okenv_controller_num_params: call okenv_controller_create first");
        *out = h->ctrl_num_params;
        return OKENV_OK;
    }

    int okenv_controller_set_params(okenv_t h, const float *params)
    {
        OK_QUIESCE(h);
        if (!h || !params || !h->d_ctrl_params)
            return fail(h, OKENV_ERR_STATE, "okenv_controller_set_params: call okenv_controller_create first");
        if (ctrlEpisodeRunning(h))
            return fail(h, OKENV_ERR_STATE, "okenv_controller_set_params: a controller episode is running; okenv_episode_end first");
        OK_HIP(h, hipSetDevice(h->device));
        return copyAny(h, h->d_ctrl_params, params, sizeof(float) * static_cast<size_t>(h->N) * h->ctrl_num_params);
    }

    int okenv_controller_act(okenv_t h, float throttle, float steering_scale)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h || !h->d_ctrl_params)
            return fail(h, OKENV_ERR_STATE, "okenv_controller_act: call okenv_controller_create first");
        OK_HIP(h, hipSetDevice(h->device));
        const int      lanes  = h->ctrl_hidden <= 16 ? 16 : (h->ctrl_hidden <= 32 ? 32 : 64);
        const unsigned blocks = static_cast<unsigned>((static_cast<long>(h->N) * lanes + 255) / 256);
        if (lanes == 16)
            hipLaunchKernelGGL(okControllerKernel<16>, dim3(blocks), dim3(256), 0, h->stream, h->st, h->d_ctrl_params, h->ctrl_num_params, h->N,
                               h->R, h->ctrl_hidden, throttle, steering_scale);
        else if (lanes == 32)
            hipLaunchKernelGGL(okControllerKernel<32>, dim3(blocks), dim3(256), 0, h->stream, h->st, h->d_ctrl_params, h->ctrl_num_params, h->N,
                               h->R, h->ctrl_hidden, throttle, steering_scale);
        else
            hipLaunchKernelGGL(okControllerKernel<64>, dim3(blocks), dim3(256), 0, h->stream, h->st, h->d_ctrl_params, h->ctrl_num_params, h->N,
                               h->R, h->ctrl_hidden, throttle, steering_scale);
        OK_HIP(h, hipGetLastError());
        return OKENV_OK;
    }

    int okenv_rollout_controller(okenv_t h, int32_t n_steps, float throttle, float steering_scale)
    {
        OK_QUIESCE(h);
        if (!h || n_steps < 0)
            return fail(h, OKENV_ERR_INVALID, "okenv_rollout_controller: bad argument");
        if (!h->d_ctrl_params)
            return fail(h, OKENV_ERR_STATE, "okenv_rollout_controller: call okenv_controller_create first");
        if (h->tracker_kind < 0)
            return fail(h, OKENV_ERR_STATE, "okenv_rollout_controller: call okenv_tracker_create first");
        if (h->grid_mode != kGridLds || !h->coop || h->rays_per_lane != 1 || h->G < 8)
            return fail(h, OKENV_ERR_STATE, "okenv_rollout_controller: needs the LDS form of the step kernel with at most 64 rays "
                                            "(use okenv_controller_act + okenv_step + okenv_tracker_update)");
        if (h->ctrl_hidden > kCtrlUnitsPerLane * h->G)
            return fail(h, OKENV_ERR_STATE, "okenv_rollout_controller: the hidden layer is too wide for this fan's lane groups (hidden <= 4 x lanes per agent); "
                                            "use okenv_controller_act + okenv_step + okenv_tracker_update");
        if (coopLdsBytes(h) + qLdsBytes(h) > kLdsBudget)
            return fail(h, OKENV_ERR_STATE, "okenv_rollout_controller: track image + centre line do not fit the CU's LDS");
        if (n_steps == 0)
            return OKENV_OK;
        OK_HIP(h, hipSetDevice(h->device));
        const int brc = buildCenterlineBuckets(h);
        if (brc != OKENV_OK)
            return brc;
        if (h->episode)
        {
            if (h->ep_kind != 0 && h->ep_kind != kPolicyCtrl)
                return fail(h, OKENV_ERR_STATE, "okenv_rollout_controller: the running episode belongs to another rollout (okenv_rollout_policy / _q)");
            if ((h->reset_flags & kAutoResetOn) != 0U)
                return fail(h, OKENV_ERR_STATE, "okenv_rollout_controller: episodes need auto-reset off");
            // (an episode stops stepping an agent once it has crashed and taken one more step; the +1-per-step reward keeps counting
            // for crashed agents, so its bookkeeping would fall behind the per-step loop's)
            if (h->tracker_kind != kRewardProgress)
                return fail(h, OKENV_ERR_STATE, "okenv_rollout_controller: inside an episode the bookkeeping must be OKENV_REWARD_PROGRESS");
            h->ep_kind = kPolicyCtrl;
        }
        OkStepParams p     = baseParams(h);
        p.n_steps          = n_steps;
        p.action_source    = kActionsController;
        p.ctrl_throttle    = throttle;
        p.ctrl_steer_scale = steering_scale;
        int rc             = OKENV_OK;
        if (!(h->episode && h->n_active == 0)) // (nobody left to step: the steps still count)
            rc = launchStep(h, p);
        if (rc == OKENV_OK && h->episode)
            h->ep_steps += static_cast<uint32_t>(n_steps);
        return rc == OKENV_OK ? advanceStepCount(h, n_steps) : rc;
    }

    // ---- EvolutionaryRacer ---------------------------------------------------------------------------------------

    int okenv_policy_mlp_create(okenv_t h, int32_t hidden, uint32_t seed, uint32_t agent_base)
    {
        OK_QUIESCE(h);
        if (!h || hidden < 1 || hidden > OK_MLP_HID_PAD)
            return fail(h, OKENV_ERR_INVALID, "okenv_policy_mlp_create: hidden width must be in [1, 32]");
        if (h->rays_per_lane != 1 || h->R < 5 || h->G < 8)
            return fail(h, OKENV_ERR_INVALID, "okenv_policy_mlp_create: the fused policy needs 5 <= rays <= 64");
        OK_HIP(h, hipSetDevice(h->device));
        const size_t total = static_cast<size_t>(h->N) * OK_MLP_WEIGHTS(h->R);
        int          rc;
        // (d_mlp_w, which the other entry points take as "a policy exists", comes last)
        if ((rc = devEnsure(h, &h->d_mlp_w_new, total)) || (rc = devEnsure(h, &h->d_score, static_cast<size_t>(h->N))) ||
            (rc = devEnsure(h, &h->d_nearest, static_cast<size_t>(h->N))) || (rc = devEnsure(h, &h->d_parents, 16U)) ||
            (rc = devEnsure(h, &h->d_parent_score, 16U)) || (rc = devEnsure(h, &h->d_alive, 4U)) || (rc = devEnsure(h, &h->d_mlp_w, total)))
            return rc;
        h->mlp_hidden = hidden;
        hipLaunchKernelGGL(okGaInitWeightsKernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, h->stream, h->d_mlp_w, h->N,
                           h->R, hidden, seed, agent_base);
        OK_HIP(h, hipGetLastError());
        return OKENV_OK;
    }

    int32_t okenv_policy_mlp_weights_per_agent(okenv_t h)
    {
        return h ? OK_MLP_WEIGHTS(h->R) : 0;
    }

    int okenv_policy_mlp_get_weights(okenv_t h, float *out)
    {
        OK_QUIESCE(h);
        if (!h || !out || !h->d_mlp_w)
            return fail(h, OKENV_ERR_STATE, "okenv_policy_mlp_get_weights: no policy");
        int rc = copyAny(h, out, h->d_mlp_w, sizeof(float) * static_cast<size_t>(h->N) * OK_MLP_WEIGHTS(h->R));
        if (rc != OKENV_OK)
            return rc;
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_policy_mlp_set_weights(okenv_t h, const float *in)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h || !in || !h->d_mlp_w)
            return fail(h, OKENV_ERR_STATE, "okenv_policy_mlp_set_weights: no policy");
        int rc = copyAny(h, h->d_mlp_w, in, sizeof(float) * static_cast<size_t>(h->N) * OK_MLP_WEIGHTS(h->R));
        if (rc != OKENV_OK)
            return rc;
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_rollout_policy(okenv_t h, int32_t n_steps)
    {
        OK_QUIESCE(h);
        if (!h || n_steps < 0)
            return fail(h, OKENV_ERR_INVALID, "okenv_rollout_policy: bad argument");
        if (!h->d_mlp_w)
            return fail(h, OKENV_ERR_STATE, "okenv_rollout_policy: call okenv_policy_mlp_create first");
        if (n_steps == 0)
            return OKENV_OK;
        if (h->episode)
        {
            if (h->ep_kind != 0 && h->ep_kind != kPolicyMlp)
                return fail(h, OKENV_ERR_STATE, "okenv_rollout_policy: the running episode belongs to another rollout (okenv_rollout_q / _controller)");
            if ((h->reset_flags & kAutoResetOn) != 0U)
                return fail(h, OKENV_ERR_STATE, "okenv_rollout_policy: episodes need auto-reset off");
            h->ep_kind = kPolicyMlp;
        }
        OkStepParams p  = baseParams(h);
        p.n_steps       = n_steps;
        p.action_source = kActionsMlpPolicy;
        int rc          = OKENV_OK;
        if (!(h->episode && h->n_active == 0)) // (nobody left to step: the steps still count)
            rc = launchStep(h, p);
        if (rc == OKENV_OK && h->episode)
            h->ep_steps += static_cast<uint32_t>(n_steps);
        return rc == OKENV_OK ? advanceStepCount(h, n_steps) : rc;
    }

    int okenv_episode_begin(okenv_t h)
    {
        OK_QUIESCE(h);
        if (!h)
            return OKENV_ERR_INVALID;
        if ((h->reset_flags & kAutoResetOn) != 0U)
            return fail(h, OKENV_ERR_STATE, "okenv_episode_begin: episodes need auto-reset off (a crashed agent stays crashed until the caller resets it)");
        OK_HIP(h, hipSetDevice(h->device));
        const size_t N = static_cast<size_t>(h->N);
        int          rc;
        if ((rc = devEnsure(h, &h->d_settled, N)) || (rc = devEnsure(h, &h->d_crash_step, N)) || (rc = devEnsure(h, &h->d_crash_thr, N)) ||
            (rc = devEnsure(h, &h->d_crash_steer, N)) || (rc = devEnsure(h, &h->d_active, N)) || (rc = devEnsure(h, &h->d_ep_counts, 2U)) ||
            (rc = devEnsure(h, &h->d_ep_out, 2U)) || (rc = devEnsure(h, &h->d_live, 1U)) || (rc = devEnsure(h, &h->d_q_next_state, N)))
            return rc;
        hipLaunchKernelGGL(okEpisodeBeginKernel, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, h->st.crashed, h->d_settled, h->d_crash_step,
                           h->d_live, h->N);
        OK_HIP(h, hipGetLastError());
        h->episode  = true;
        h->n_active = -1;
        h->ep_kind  = 0;
        h->ep_steps = 0U;
        // A population that fits the tail kernel (the reference's 50 agents, say) is listed from the start: nobody is settled yet,
        // so the list is 0 ... N-1 and its length is known without asking the device -- the very first rollout then already runs
        // one agent per workgroup, each leaving with its agent, instead of the cooperative kernel that okenv_episode_compact would
        // only replace after the first launch.
        const bool q    = h->d_q_table != nullptr && h->d_mlp_w == nullptr;
        const bool ctrl = h->d_ctrl_params != nullptr && h->d_mlp_w == nullptr && h->d_q_table == nullptr;
        if (!ctrl && static_cast<long>(h->N) <= tailLimit(h, q))
        {
            hipLaunchKernelGGL(okEpisodeCompactKernel, dim3(1), dim3(1024), 0, h->stream, h->d_settled, h->st.crashed, h->N, h->d_active, h->d_ep_counts);
            OK_HIP(h, hipGetLastError());
            h->n_active = h->N;
        }
        return OKENV_OK;
    }

    int okenv_episode_compact(okenv_t h, int32_t *alive_out, int32_t *listed_out)
    {
        OK_QUIESCE(h);
        if (!h || !h->episode)
            return fail(h, OKENV_ERR_STATE, "okenv_episode_compact: no episode is running (okenv_episode_begin)");
        OK_HIP(h, hipSetDevice(h->device));
        hipLaunchKernelGGL(okEpisodeCompactKernel, dim3(1), dim3(1024), 0, h->stream, h->d_settled, h->st.crashed, h->N, h->d_active, h->d_ep_counts);
        OK_HIP(h, hipGetLastError());
        int32_t counts[2] = {0, 0};
        OK_HIP(h, hipMemcpyAsync(counts, h->d_ep_counts, sizeof(counts), hipMemcpyDeviceToHost, h->stream));
        OK_HIP(h, hipStreamSynchronize(h->stream));
        h->n_active = counts[1];
        if (alive_out)
            *alive_out = counts[0];
        if (listed_out)
            *listed_out = counts[1];
        return OKENV_OK;
    }

    int okenv_episode_tail_limit(okenv_t h, int32_t *out)
    {
        OK_QUIESCE(h);
        if (!h || !out)
            return fail(h, OKENV_ERR_INVALID, "okenv_episode_tail_limit: NULL argument");
        const bool q    = h->ep_kind == kPolicyQ || (h->ep_kind == 0 && h->d_q_table != nullptr && h->d_mlp_w == nullptr);
        const bool ctrl = h->ep_kind == kPolicyCtrl || (h->ep_kind == 0 && h->d_ctrl_params != nullptr && h->d_mlp_w == nullptr && h->d_q_table == nullptr);
        *out            = ctrl ? 0 : static_cast<int32_t>(tailLimit(h, q)); // (the controller rollout has no one-agent-per-workgroup form)
        return OKENV_OK;
    }

    int okenv_episode_end(okenv_t h, int32_t *steps_out, uint64_t *live_agent_steps_out)
    {
        OK_QUIESCE(h);
        if (!h || !h->episode)
            return fail(h, OKENV_ERR_STATE, "okenv_episode_end: no episode is running (okenv_episode_begin)");
        OK_HIP(h, hipSetDevice(h->device));
        const unsigned blocks = static_cast<unsigned>((h->N + 255) / 256);
        hipLaunchKernelGGL(okEpisodeEndKernel, dim3(1), dim3(1024), 0, h->stream, h->st.crashed, h->d_crash_step, h->N, h->ep_steps, h->d_ep_out);
        if (h->ep_kind == kPolicyMlp || h->ep_kind == kPolicyCtrl)
            hipLaunchKernelGGL(okEpisodeFixupKernel, dim3(blocks), dim3(256), 0, h->stream, h->st, h->d_crash_step, h->d_crash_thr, h->d_crash_steer,
                               h->d_ep_out, h->N);
        else if (h->ep_kind == kPolicyQ)
            hipLaunchKernelGGL(okQSettleKernel, dim3(static_cast<unsigned>((static_cast<long>(h->N) * kSettleLanes + 255) / 256)), dim3(256), 0,
                               h->stream, h->st, h->d_q_table, h->d_q_state, h->d_q_action,
                               h->d_q_next_state, h->d_crash_step, h->d_ep_out, h->N, h->ep_q_seed, h->ep_q_agent_base, h->ep_q_step_base,
                               h->ep_q_epsilon, h->R, h->q_ray[0], h->q_ray[1], h->q_ray[2], h->q_ray[3], h->q_ray[4]);
        OK_HIP(h, hipGetLastError());
        uint32_t           out[2] = {0U, 0U};
        unsigned long long live   = 0ULL;
        OK_HIP(h, hipMemcpyAsync(out, h->d_ep_out, sizeof(out), hipMemcpyDeviceToHost, h->stream));
        OK_HIP(h, hipMemcpyAsync(&live, h->d_live, sizeof(live), hipMemcpyDeviceToHost, h->stream));
        OK_HIP(h, hipStreamSynchronize(h->stream));
        // the steps beyond T were taken by nobody who could still move: they do not count
        if (h->ep_kind == kPolicyMlp || h->ep_kind == kPolicyCtrl)
            h->step_count -= h->ep_steps - out[0];
        if (steps_out)
            *steps_out = static_cast<int32_t>(out[0]);
        if (live_agent_steps_out)
            *live_agent_steps_out = live;
        dropEpisode(h);
        return OKENV_OK;
    }

    int okenv_alive_count(okenv_t h, int32_t *out)
    {
        OK_QUIESCE(h);
        if (!h || !out)
            return fail(h, OKENV_ERR_INVALID, "okenv_alive_count: NULL argument");
        OK_HIP(h, hipSetDevice(h->device));
        int *d = h->d_alive;
        if (!d)
        {
            int rc = devAlloc(h, &h->d_alive, 4U);
            if (rc != OKENV_OK)
                return rc;
            d = h->d_alive;
        }
        OK_HIP(h, hipMemsetAsync(d, 0, sizeof(int), h->stream));
        hipLaunchKernelGGL(okAliveCountKernel, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, h->st.crashed, h->N, d);
        OK_HIP(h, hipGetLastError());
        OK_HIP(h, hipMemcpyAsync(out, d, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_off_grid_count(okenv_t h, int32_t *alive_off_grid, int32_t *all_off_grid)
    {
        OK_QUIESCE(h);
        if (!h)
            return OKENV_ERR_INVALID;
        OK_HIP(h, hipSetDevice(h->device));
        void     *sp = nullptr;
        const int rc = deviceScratch(h, 2U * sizeof(int), &sp);
        if (rc != OKENV_OK)
            return rc;
        int *d = static_cast<int *>(sp);
        OK_HIP(h, hipMemsetAsync(d, 0, 2U * sizeof(int), h->stream));
        const OkGridGeom &g = h->grid.g;
        hipLaunchKernelGGL(okOffGridCountKernel, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, h->st.pos_x, h->st.pos_y, h->st.crashed, h->N,
                           g.x0, g.y0, g.x1, g.y1, d);
        OK_HIP(h, hipGetLastError());
        int host[2] = {0, 0};
        OK_HIP(h, hipMemcpyAsync(host, d, 2U * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        OK_HIP(h, hipStreamSynchronize(h->stream));
        if (alive_off_grid)
            *alive_off_grid = host[0];
        if (all_off_grid)
            *all_off_grid = host[1];
        return OKENV_OK;
    }

    int okenv_reset_all(okenv_t h, float x, float y, float rot_deg)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h)
            return OKENV_ERR_INVALID;
        OK_HIP(h, hipSetDevice(h->device));
        hipLaunchKernelGGL(okResetAllKernel, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, h->st, x, y, rot_deg, h->N);
        OK_HIP(h, hipGetLastError());
        return OKENV_OK;
    }

    int okenv_ga_scores(okenv_t h, float *out)
    {
        OK_QUIESCE(h);
        if (!h)
            return OKENV_ERR_INVALID;
        if (!h->d_mlp_w || h->P <= 0)
            return fail(h, OKENV_ERR_STATE, "okenv_ga_scores: needs okenv_policy_mlp_create and okenv_set_centerline");
        OK_HIP(h, hipSetDevice(h->device));
        hipLaunchKernelGGL(okNearestIdxKernel, dim3((h->N * kNearestLanes + 255) / 256), dim3(256), 0, h->stream, h->d_cx, h->d_cy, h->P,
                           h->st.pos_x, h->st.pos_y, h->N, h->d_nearest);
        hipLaunchKernelGGL(okGaScoreKernel, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, h->d_nearest, h->d_score, h->N);
        OK_HIP(h, hipGetLastError());
        if (out)
        {
            int rc = copyAny(h, out, h->d_score, sizeof(float) * static_cast<size_t>(h->N));
            if (rc != OKENV_OK)
                return rc;
            OK_HIP(h, hipStreamSynchronize(h->stream));
        }
        return OKENV_OK;
    }

    int okenv_ga_scores_device(okenv_t h, const float **ptr)
    {
        OK_QUIESCE(h);
        if (!h || !ptr)
            return OKENV_ERR_INVALID;
        if (!h->d_score)
            return fail(h, OKENV_ERR_STATE, "okenv_ga_scores_device: call okenv_policy_mlp_create first");
        *ptr = h->d_score;
        return OKENV_OK;
    }

    int okenv_get_stream(okenv_t h, void **hip_stream)
    {
        OK_QUIESCE(h);
        if (!h || !hip_stream)
            return OKENV_ERR_INVALID;
        *hip_stream = static_cast<void *>(h->stream);
        return OKENV_OK;
    }

    int okenv_ga_select_mate(okenv_t h, uint32_t seed, uint32_t generation, uint32_t agent_base, int32_t *parents_out)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h)
            return OKENV_ERR_INVALID;
        if (!h->d_mlp_w)
            return fail(h, OKENV_ERR_STATE, "okenv_ga_select_mate: call okenv_policy_mlp_create first");
        OK_HIP(h, hipSetDevice(h->device));
        const int    K     = h->N < 5 ? h->N : 5; // kNumParents (Mating.hpp:118)
        const size_t total = static_cast<size_t>(h->N) * OK_MLP_WEIGHTS(h->R);
        hipLaunchKernelGGL(okGaTopKernel, dim3(1), dim3(1024), 0, h->stream, h->d_score, h->N, h->d_parents, h->d_parent_score, K);
        hipLaunchKernelGGL(okGaMateKernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, h->stream, h->d_mlp_w, h->d_mlp_w_new,
                           h->d_parents, h->d_parent_score, K, h->N, h->R, h->mlp_hidden, seed, generation, agent_base);
        OK_HIP(h, hipGetLastError());
        std::swap(h->d_mlp_w, h->d_mlp_w_new);
        if (parents_out)
        {
            OK_HIP(h, hipMemcpyAsync(parents_out, h->d_parents, sizeof(int32_t) * K, hipMemcpyDeviceToHost, h->stream));
            OK_HIP(h, hipStreamSynchronize(h->stream));
        }
        return OKENV_OK;
    }

    // ---- RLRacers/Q_Learning ------------------------------------------------------------------------------------

    int okenv_q_create(okenv_t h)
    {
        OK_QUIESCE(h);
        if (!h)
            return OKENV_ERR_INVALID;
        if (!h->coop || h->R < 5)
            return fail(h, OKENV_ERR_INVALID, "okenv_q_create: needs the LDS form with 5 <= rays <= 64");
        OK_HIP(h, hipSetDevice(h->device));
        const size_t n = static_cast<size_t>(h->N) * OK_Q_STATES * OK_Q_ACTIONS;
        int          rc;
        // (d_q_table, which the other entry points take as "the tables exist", comes last)
        if ((rc = devEnsure(h, &h->d_q_state, static_cast<size_t>(h->N))) || (rc = devEnsure(h, &h->d_q_action, static_cast<size_t>(h->N))) ||
            (rc = devEnsure(h, &h->d_q_prev, static_cast<size_t>(h->N))) || (rc = devEnsure(h, &h->d_q_reset_nearest, 4U)) ||
            (rc = devEnsure(h, &h->d_q_reset_query, 4U)) || (rc = devEnsure(h, &h->d_q_table, n)))
            return rc;
        hipLaunchKernelGGL(okQInitTableKernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, h->stream, h->d_q_table,
                           static_cast<long>(n));
        OK_HIP(h, hipGetLastError());
        // the five rays of the state: nearest to -70, -30, 0, 30, 70 degrees, ties to the lower index
        const float target[5] = {-70.F, -30.F, 0.F, 30.F, 70.F};
        for (int t = 0; t < 5; ++t)
        {
            int   arg  = 0;
            float best = std::fabs(h->host_ray_deg[0] - target[t]);
            for (int r = 1; r < h->R; ++r)
            {
                const float d = std::fabs(h->host_ray_deg[r] - target[t]);
                if (d < best)
                {
                    best = d;
                    arg  = r;
                }
            }
            h->q_ray[t] = arg;
        }
        return OKENV_OK;
    }

    int okenv_q_begin_episode(okenv_t h, int32_t reset_idx)
    {
        OK_QUIESCE(h);
        if (!h || !h->d_q_table)
            return fail(h, OKENV_ERR_STATE, "okenv_q_begin_episode: call okenv_q_create first");
        if (h->P <= 0 || reset_idx < 0 || reset_idx >= h->P)
            return fail(h, OKENV_ERR_INVALID, "okenv_q_begin_episode: needs a centre line and a valid reset index");
        const float x = h->host_cx[reset_idx], y = h->host_cy[reset_idx];
        int         rc = okenv_reset_all(h, x, y, h->host_chead[reset_idx]);
        if (rc != OKENV_OK)
            return rc;
        // prev_track_idx_ = findNearestTrackIndexBruteForce(reset point) (q_racer_sim.cpp:134-139); one query on the device
        // (the query lives in the handle on both sides: an asynchronous copy out of a local variable may still be reading it
        // after this function has returned)
        h->q_reset_query[0] = x;
        h->q_reset_query[1] = y;
        float *dq           = h->d_q_reset_query;
        OK_HIP(h, hipMemcpyAsync(dq, h->q_reset_query, 8U, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(okNearestIdxKernel, dim3(1), dim3(256), 0, h->stream, h->d_cx, h->d_cy, h->P, dq, dq + 1, 1, h->d_q_reset_nearest);
        OK_HIP(h, hipGetLastError());
        if ((rc = okenv_step(h, 1)) != OKENV_OK) // initial observation with the zero action Agent::reset leaves behind
            return rc;
        hipLaunchKernelGGL(okQBeginEpisodeKernel, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, h->st.dist, h->R, h->q_ray[0], h->q_ray[1],
                           h->q_ray[2], h->q_ray[3], h->q_ray[4], h->d_q_state, h->d_q_prev, h->d_q_reset_nearest, h->N);
        OK_HIP(h, hipGetLastError());
        return OKENV_OK;
    }

    int okenv_rollout_q(okenv_t h, int32_t n_steps, float epsilon, uint32_t seed, uint32_t agent_base, uint32_t step_base)
    {
        OK_QUIESCE(h);
        if (!h || n_steps < 0)
            return fail(h, OKENV_ERR_INVALID, "okenv_rollout_q: bad argument");
        if (!h->d_q_table)
            return fail(h, OKENV_ERR_STATE, "okenv_rollout_q: call okenv_q_create first");
        if (n_steps == 0)
            return OKENV_OK;
        if (coopLdsBytes(h) + qLdsBytes(h) > kLdsBudget)
            return fail(h, OKENV_ERR_STATE, "okenv_rollout_q: track image + centre line do not fit the CU's LDS");
        OK_HIP(h, hipSetDevice(h->device));
        const int brc = buildCenterlineBuckets(h);
        if (brc != OKENV_OK)
            return brc;
        h->q_epsilon    = epsilon;
        if (h->episode)
        {
            if (h->ep_kind != 0 && h->ep_kind != kPolicyQ)
                return fail(h, OKENV_ERR_STATE, "okenv_rollout_q: the running episode belongs to another rollout (okenv_rollout_policy / _controller)");
            if (h->ep_kind == 0)
            {
                h->ep_kind         = kPolicyQ;
                h->ep_q_seed       = seed;
                h->ep_q_agent_base = agent_base;
                h->ep_q_epsilon    = epsilon;
                h->ep_q_step_base  = step_base - h->ep_steps;
            }
            else if (h->ep_q_seed != seed || h->ep_q_agent_base != agent_base || h->ep_q_epsilon != epsilon ||
                     h->ep_q_step_base + h->ep_steps != step_base)
                return fail(h, OKENV_ERR_INVALID, "okenv_rollout_q: inside an episode seed, agent_base and epsilon must stay the same and "
                                                  "step_base advance by the steps taken");
        }
        OkStepParams p  = baseParams(h);
        p.n_steps       = n_steps;
        p.action_source = kActionsQLearning;
        p.reset_flags   = 0; // episodes are restarted by okenv_q_begin_episode
        p.seed          = seed;
        p.agent_base    = agent_base;
        p.step_base     = step_base;
        int rc          = OKENV_OK;
        if (!(h->episode && h->n_active == 0))
            rc = launchStep(h, p);
        if (rc == OKENV_OK && h->episode)
            h->ep_steps += static_cast<uint32_t>(n_steps);
        return rc;
    }

    int okenv_q_get_table(okenv_t h, float *out)
    {
        OK_QUIESCE(h);
        if (!h || !out || !h->d_q_table)
            return fail(h, OKENV_ERR_STATE, "okenv_q_get_table: no table");
        int rc = copyAny(h, out, h->d_q_table, sizeof(float) * static_cast<size_t>(h->N) * OK_Q_STATES * OK_Q_ACTIONS);
        if (rc != OKENV_OK)
            return rc;
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_q_set_table(okenv_t h, const float *in)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h || !in || !h->d_q_table)
            return fail(h, OKENV_ERR_STATE, "okenv_q_set_table: no table");
        int rc = copyAny(h, h->d_q_table, in, sizeof(float) * static_cast<size_t>(h->N) * OK_Q_STATES * OK_Q_ACTIONS);
        if (rc != OKENV_OK)
            return rc;
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_q_get_state(okenv_t h, int32_t *state, int32_t *action, int32_t *prev_idx)
    {
        OK_QUIESCE(h);
        if (!h || !h->d_q_table)
            return fail(h, OKENV_ERR_STATE, "okenv_q_get_state: no table");
        int rc;
        if ((state && (rc = copyAny(h, state, h->d_q_state, 4U * h->N))) || (action && (rc = copyAny(h, action, h->d_q_action, 4U * h->N))) ||
            (prev_idx && (rc = copyAny(h, prev_idx, h->d_q_prev, 4U * h->N))))
            return rc;
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    static int qSums(okenv_t h, float **d_out)
    {
        if (!h || !h->d_q_table)
            return fail(h, OKENV_ERR_STATE, "no Q table (call okenv_q_create first)");
        OK_HIP(h, hipSetDevice(h->device));
        if (!h->d_q_sums)
        {
            int rc = devAlloc(h, &h->d_q_sums, 2U * OK_Q_STATES * OK_Q_ACTIONS);
            if (rc != OKENV_OK)
                return rc;
        }
        constexpr int kEntries = OK_Q_STATES * OK_Q_ACTIONS;
        hipLaunchKernelGGL(okQTableSumsKernel, dim3((kEntries + 63) / 64), dim3(64), 0, h->stream, h->d_q_table, h->N, h->d_q_sums,
                           h->d_q_sums + kEntries);
        OK_HIP(h, hipGetLastError());
        *d_out = h->d_q_sums;
        return OKENV_OK;
    }

    int okenv_q_table_sums(okenv_t h, float *sum, float *count)
    {
        OK_QUIESCE(h);
        if (!sum || !count)
            return fail(h, OKENV_ERR_INVALID, "okenv_q_table_sums: NULL argument");
        float *d  = nullptr;
        int    rc = qSums(h, &d);
        if (rc != OKENV_OK)
            return rc;
        constexpr size_t kBytes = sizeof(float) * OK_Q_STATES * OK_Q_ACTIONS;
        if ((rc = copyAny(h, sum, d, kBytes)) || (rc = copyAny(h, count, d + OK_Q_STATES * OK_Q_ACTIONS, kBytes)))
            return rc;
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_q_assign_mean(okenv_t h, const float *sum, const float *count)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        if (!h || !h->d_q_table || !sum || !count)
            return fail(h, OKENV_ERR_STATE, "okenv_q_assign_mean: no Q table or NULL argument");
        OK_HIP(h, hipSetDevice(h->device));
        if (!h->d_q_sums)
        {
            int rc = devAlloc(h, &h->d_q_sums, 2U * OK_Q_STATES * OK_Q_ACTIONS);
            if (rc != OKENV_OK)
                return rc;
        }
        constexpr int    kEntries = OK_Q_STATES * OK_Q_ACTIONS;
        constexpr size_t kBytes   = sizeof(float) * kEntries;
        int              rc;
        if ((sum != h->d_q_sums && (rc = copyAny(h, h->d_q_sums, sum, kBytes))) ||
            (count != h->d_q_sums + kEntries && (rc = copyAny(h, h->d_q_sums + kEntries, count, kBytes))))
            return rc;
        const long total = static_cast<long>(h->N) * kEntries;
        hipLaunchKernelGGL(okQAssignAllKernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, h->stream, h->d_q_table, h->N,
                           h->d_q_sums, h->d_q_sums + kEntries);
        OK_HIP(h, hipGetLastError());
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }

    int okenv_q_share_knowledge(okenv_t h)
    {
        OK_QUIESCE(h);
        if (h)
            dropEpisode(h);
        float *d  = nullptr;
        int    rc = qSums(h, &d);
        if (rc != OKENV_OK)
            return rc;
        return okenv_q_assign_mean(h, d, d + OK_Q_STATES * OK_Q_ACTIONS);
    }

    int okenv_set_timing(okenv_t h, int32_t enabled)
    {
        OK_QUIESCE(h);
        if (!h)
            return OKENV_ERR_INVALID;
        h->timing = enabled != 0;
        return OKENV_OK;
    }

    int okenv_get_timing(okenv_t h, double *total_ms, uint64_t *launches)
    {
        OK_QUIESCE(h);
        if (!h || !total_ms || !launches)
            return fail(h, OKENV_ERR_INVALID, "okenv_get_timing: NULL argument");
        OK_HIP(h, hipStreamSynchronize(h->stream));
        double sum = 0.0;
        for (auto &e : h->events)
        {
            float ms = 0.F;
            OK_HIP(h, hipEventElapsedTime(&ms, e.start, e.stop));
            sum += ms;
        }
        *total_ms = sum + h->timing_carry_ms;
        *launches = h->events.size() + h->timing_carry_n;
        h->timing_carry_ms = 0.0;
        h->timing_carry_n  = 0;
        for (auto &e : h->events)
            h->event_pool.push_back(e);
        h->events.clear();
        return OKENV_OK;
    }

    // ---- track ------------------------------------------------------------------------------------------

    int okenv_track_load(okenv_track_t *out, const char *csv_path)
    {
        if (!out || !csv_path)
            return fail(nullptr, OKENV_ERR_INVALID, "okenv_track_load: NULL argument");
        *out = nullptr;
        std::unique_ptr<okenv_track> t(new okenv_track);
        t->track.reset(new RaceTrack(std::string(csv_path)));
        if (!t->track->loadedOk())
            return fail(nullptr, OKENV_ERR_IO, std::string("okenv_track_load: cannot read ") + csv_path);
        // TrackSegments (TrackSegments.cu:6-42): LI, LO, RI, RO polylines, then closers LI, RI, LO, RO
        const RaceTrack &rt   = *t->track;
        auto             run = [&](const std::vector<Vec2d> &poly) {
            for (size_t i = 0; i + 1 < poly.size(); ++i)
                t->segments.push_back({poly[i].x, poly[i].y, poly[i + 1].x, poly[i + 1].y});
        };
        auto closer = [&](const std::vector<Vec2d> &poly) {
            t->segments.push_back({poly.back().x, poly.back().y, poly.front().x, poly.front().y});
        };
        run(rt.left_bound_inner_);
        run(rt.left_bound_outer_);
        run(rt.right_bound_inner_);
        run(rt.right_bound_outer_);
        if (rt.left_bound_inner_.size() > 1)
        {
            closer(rt.left_bound_inner_);
            closer(rt.right_bound_inner_);
        }
        if (rt.left_bound_outer_.size() > 1)
        {
            closer(rt.left_bound_outer_);
            closer(rt.right_bound_outer_);
        }
        *out = t.release();
        return OKENV_OK;
    }

    int okenv_track_free(okenv_track_t t)
    {
        delete t;
        return OKENV_OK;
    }

    int32_t okenv_track_num_points(okenv_track_t t)
    {
        return t ? static_cast<int32_t>(t->track->track_data_points_.x_m.size()) : 0;
    }

    int32_t okenv_track_num_segments(okenv_track_t t)
    {
        return t ? static_cast<int32_t>(t->segments.size()) : 0;
    }

    int okenv_track_get(okenv_track_t t, int32_t which, float *out)
    {
        if (!t || !out)
            return OKENV_ERR_INVALID;
        const RaceTrack &rt  = *t->track;
        const auto      &d   = rt.track_data_points_;
        auto             vec = [&](const std::vector<float> &v) { std::memcpy(out, v.data(), v.size() * 4U); };
        auto             pts = [&](const std::vector<Vec2d> &v) {
            for (size_t i = 0; i < v.size(); ++i)
            {
                out[2 * i]     = v[i].x;
                out[2 * i + 1] = v[i].y;
            }
        };
        switch (which)
        {
        case 0: vec(d.x_m); break;
        case 1: vec(d.y_m); break;
        case 2: vec(d.w_tr_right_m); break;
        case 3: vec(d.w_tr_left_m); break;
        case 4: vec(rt.headings_); break;
        case 5: pts(rt.left_bound_inner_); break;
        case 6: pts(rt.left_bound_outer_); break;
        case 7: pts(rt.right_bound_inner_); break;
        case 8: pts(rt.right_bound_outer_); break;
        default: return OKENV_ERR_INVALID;
        }
        return OKENV_OK;
    }

    int okenv_track_queries(okenv_track_t t, const float *qx, const float *qy, int32_t n, float *out_boundary_distance,
                            float *out_lane_center_ratio)
    {
        if (!t || !qx || !qy || n < 0)
            return OKENV_ERR_INVALID;
        const RaceTrack &rt = *t->track;
        for (int32_t i = 0; i < n; ++i)
        {
            const Vec2d q{qx[i], qy[i]};
            if (out_boundary_distance)
                out_boundary_distance[i] = rt.getNearestDistanceToTrackBoundary(q);
            if (out_lane_center_ratio)
                out_lane_center_ratio[i] = rt.getDistanceToLaneCenter(q);
        }
        return OKENV_OK;
    }

    int okenv_track_segments(okenv_track_t t, float *out_xyxy)
    {
        if (!t || !out_xyxy)
            return OKENV_ERR_INVALID;
        std::memcpy(out_xyxy, t->segments.data(), t->segments.size() * sizeof(Segment2d));
        return OKENV_OK;
    }

#if defined(OKENV_STAMPS)
    // stamps of the last step launch: kStampWords words per wave; returns the number of waves copied (<= waves) or an error (< 0)
    __attribute__((visibility("default"))) int okenv_debug_stamps(okenv_t h, unsigned long long *out, int waves)
    {
        OK_QUIESCE(h);
        OK_HIP(h, hipStreamSynchronize(h->stream));
        const size_t n = std::min(static_cast<size_t>(waves < 0 ? 0 : waves), h->stamp_waves);
        if (n > 0)
            OK_HIP(h, hipMemcpy(out, h->d_stamps, sizeof(unsigned long long) * kStampWords * n, hipMemcpyDeviceToHost));
        return static_cast<int>(n);
    }
#endif

    // ---- device self-checks ----------------------------------------------------------------------------------

    int okenv_debug_sincos(int32_t device, const float *x, float *s, float *c, int32_t n)
    {
        if (!x || !s || !c || n < 0)
            return fail(nullptr, OKENV_ERR_INVALID, "okenv_debug_sincos: bad argument");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            return fail(nullptr, OKENV_ERR_NO_DEVICE, "okenv_debug_sincos: no HIP device");
        if (n == 0)
            return OKENV_OK;
        OK_HIP(nullptr, hipSetDevice(device));
        float *d = nullptr;
        OK_HIP(nullptr, hipMalloc(reinterpret_cast<void **>(&d), 12U * static_cast<size_t>(n)));
        OK_HIP(nullptr, hipMemcpy(d, x, 4U * static_cast<size_t>(n), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(okDebugSincosKernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, d, d + n, d + 2 * static_cast<size_t>(n), n);
        OK_HIP(nullptr, hipGetLastError());
        OK_HIP(nullptr, hipMemcpy(s, d + n, 4U * static_cast<size_t>(n), hipMemcpyDeviceToHost));
        OK_HIP(nullptr, hipMemcpy(c, d + 2 * static_cast<size_t>(n), 4U * static_cast<size_t>(n), hipMemcpyDeviceToHost));
        OK_HIP(nullptr, hipFree(d));
        return OKENV_OK;
    }

    static int workStats(okenv_t h, uint64_t *out, const int n_out, const bool split, const char *who)
    {
        if (!h || !out)
            return fail(h, OKENV_ERR_INVALID, std::string(who) + ": NULL argument");
        if (h->grid_mode != kGridLds)
            return fail(h, OKENV_ERR_STATE, std::string(who) + ": needs the LDS form of the grid");
        if (split && !h->fb_ok)
            return fail(h, OKENV_ERR_STATE, std::string(who) + ": the segment set has no front / back split (okenv_info.front_back_bytes == 0)");
        OK_HIP(h, hipSetDevice(h->device));
        void     *sp  = nullptr;
        const int src = deviceScratch(h, 8U * sizeof(unsigned long long), &sp);
        if (src != OKENV_OK)
            return src;
        OK_HIP(h, hipMemsetAsync(sp, 0, 8U * sizeof(unsigned long long), h->stream));
        OkStepParams p = baseParams(h);
        if (split)
            useFrontBack(h, p);
        OK_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&okWorkStatsKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(kLdsBudget)));
        const long total  = static_cast<long>(h->N) * h->R;
        const int  blocks = static_cast<int>(std::min<long>(256, (total + 1023) / 1024));
        hipLaunchKernelGGL(okWorkStatsKernel, dim3(blocks), dim3(1024), p.image_bytes, h->stream, p, static_cast<unsigned long long *>(sp));
        OK_HIP(h, hipGetLastError());
        unsigned long long host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        OK_HIP(h, hipMemcpyAsync(host, sp, sizeof(host), hipMemcpyDeviceToHost, h->stream));
        OK_HIP(h, hipStreamSynchronize(h->stream));
        for (int i = 0; i < n_out; ++i)
            out[i] = host[i];
        return OKENV_OK;
    }

    int okenv_work_stats(okenv_t h, uint64_t out[4])
    {
        OK_QUIESCE(h);
        return workStats(h, out, 4, false, "okenv_work_stats");
    }

    int okenv_work_stats_split(okenv_t h, uint64_t out[8])
    {
        OK_QUIESCE(h);
        return workStats(h, out, 8, true, "okenv_work_stats_split");
    }

    int okenv_debug_cast_rays(okenv_t h, const float *ox, const float *oy, const float *angle_rad, int32_t n, float *out_t)
    {
        OK_QUIESCE(h);
        if (!h || !ox || !oy || !angle_rad || !out_t || n < 0)
            return fail(h, OKENV_ERR_INVALID, "okenv_debug_cast_rays: bad argument");
        if (n == 0)
            return OKENV_OK;
        OK_HIP(h, hipSetDevice(h->device));
        void     *sp  = nullptr;
        const int src = deviceScratch(h, 16U * static_cast<size_t>(n), &sp);
        if (src != OKENV_OK)
            return src;
        float *d = static_cast<float *>(sp);
        OK_HIP(h, hipMemcpyAsync(d, ox, 4U * static_cast<size_t>(n), hipMemcpyHostToDevice, h->stream));
        OK_HIP(h, hipMemcpyAsync(d + n, oy, 4U * static_cast<size_t>(n), hipMemcpyHostToDevice, h->stream));
        OK_HIP(h, hipMemcpyAsync(d + 2 * static_cast<size_t>(n), angle_rad, 4U * static_cast<size_t>(n), hipMemcpyHostToDevice, h->stream));
        const OkStepParams p      = baseParams(h);
        const int          blocks = std::min(1024, (n + 1023) / 1024);
        float             *dt     = d + 3 * static_cast<size_t>(n);
        switch (h->grid_mode)
        {
        case kGridLds:
            hipLaunchKernelGGL(okDebugCastKernel<kGridLds>, dim3(blocks), dim3(1024), h->image_bytes, h->stream, p, d, d + n,
                               d + 2 * static_cast<size_t>(n), n, dt);
            break;
        case kGridGlobal:
            hipLaunchKernelGGL(okDebugCastKernel<kGridGlobal>, dim3(blocks), dim3(1024), 0, h->stream, p, d, d + n,
                               d + 2 * static_cast<size_t>(n), n, dt);
            break;
        default:
            hipLaunchKernelGGL(okDebugCastKernel<kGridBrute>, dim3(blocks), dim3(1024), 0, h->stream, p, d, d + n,
                               d + 2 * static_cast<size_t>(n), n, dt);
            break;
        }
        OK_HIP(h, hipGetLastError());
        OK_HIP(h, hipMemcpyAsync(out_t, dt, 4U * static_cast<size_t>(n), hipMemcpyDeviceToHost, h->stream));
        OK_HIP(h, hipStreamSynchronize(h->stream));
        return OKENV_OK;
    }
}
