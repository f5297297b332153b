/*
 * okenv.h -- C ABI of the MI355X-native batched Environment step (libokenv.so).
 *
 * This is the drop-in boundary for the hot path of goksanisil23/OpenKitchen (paths below are relative to
 * the reference tree): everything `Environment::step()` does per step -- Agent kinematics, the
 * standstill timeout, the 2-D raycast lidar against the RaceTrack's boundary segments and the
 * lidar-based crash test -- for N agents x R rays per launch, with the agent state resident on the GPU
 * as struct-of-arrays.  The C++ classes in include/Environment/ (same names and members as the
 * reference's) are thin hosts over these entry points; INTEGRATION.md shows the binding a maintainer of
 * the reference would add.
 *
 * Conventions: every function returns an int status (OKENV_OK == 0, negative on error) unless noted; no
 * exception crosses this boundary; `okenv_last_error` returns a message for the most recent failure on
 * the handle (or globally, for NULL).  Host buffers are caller-owned; device buffers are library-owned.
 * A handle owns one HIP stream; work is enqueued asynchronously and the `get`/`download` calls
 * synchronise.  A handle is used by one thread at a time; multi-GPU means one handle per device (one
 * process per GPU in bench.py).  Pointers passed to `okenv_set_field` / `okenv_get_field` /
 * `okenv_set_actions` may be host OR device pointers (hipMemcpyDefault).
 */
#ifndef OKENV_H
#define OKENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define OKENV_API
#else
#define OKENV_API __attribute__((visibility("default")))
#endif

#define OKENV_OK 0
#define OKENV_ERR_INVALID (-1)  /* bad argument */
#define OKENV_ERR_HIP (-2)      /* a HIP runtime call failed */
#define OKENV_ERR_NO_DEVICE (-3)/* no usable GPU: the product path has NO CPU fallback */
#define OKENV_ERR_IO (-4)       /* track CSV could not be read */
#define OKENV_ERR_STATE (-5)    /* call not valid in the handle's current state */

typedef struct okenv *okenv_t;
typedef struct okenv_track *okenv_track_t;

/* Agent::MovementMode, Environment/Agent.h:20-25 */
#define OKENV_MODE_VELOCITY 0
#define OKENV_MODE_ACCELERATION 1
#define OKENV_MODE_MANUAL 2

/* okenv_create flags */
#define OKENV_FLAG_NONE 0u
#define OKENV_FLAG_FORCE_GLOBAL_GRID 1u /* keep the grid in global memory even if it fits LDS (testing) */
#define OKENV_FLAG_BRUTE_FORCE 2u       /* sweep all segments like the reference kernel (testing / ablation) */

/*
 * State fields (struct-of-arrays).  One value per agent unless marked [N*R] (agent-major, ray-minor).
 * They mirror, field for field, what the reference keeps per Agent (Environment/Agent.h:56-81), per
 * DisplacementStats (Environment/Environment.h:17-27) and per Ray_ (Environment/Typedefs.h:91-99).
 */
enum okenv_field {
    OKENV_F_POS_X = 0,     /* f32  Agent::pos_.x                                    */
    OKENV_F_POS_Y = 1,     /* f32  Agent::pos_.y                                    */
    OKENV_F_ROT = 2,       /* f32  Agent::rot_ [deg], never wrapped                 */
    OKENV_F_SPEED = 3,     /* f32  Agent::speed_                                    */
    OKENV_F_ACC = 4,       /* f32  Agent::acceleration_                             */
    OKENV_F_THROTTLE = 5,  /* f32  Agent::current_action_.throttle_delta            */
    OKENV_F_STEER = 6,     /* f32  Agent::current_action_.steering_delta            */
    OKENV_F_MODE = 7,      /* u8   Agent::movement_mode_                            */
    OKENV_F_CRASHED = 8,   /* u8   Agent::crashed_                                  */
    OKENV_F_TIMED_OUT = 9, /* u8   Agent::timed_out_                                */
    OKENV_F_DISP_CTR = 10, /* u32  DisplacementStats::displacement_ctr              */
    OKENV_F_DISP_X = 11,   /* f32  DisplacementStats::init_pos.x                    */
    OKENV_F_DISP_Y = 12,   /* f32  DisplacementStats::init_pos.y                    */
    OKENV_F_DISP_TO = 13,  /* u8   DisplacementStats::displacement_timed_out        */
    OKENV_F_HIT_X = 14,    /* f32 [N*R] Ray_::hit_x (world frame, persists while crashed) */
    OKENV_F_HIT_Y = 15,    /* f32 [N*R] Ray_::hit_y                                 */
    OKENV_F_REL_X = 16,    /* f32 [N*R] Agent::sensor_hits_[r].x ("robot frame")    */
    OKENV_F_REL_Y = 17,    /* f32 [N*R] Agent::sensor_hits_[r].y                    */
    OKENV_F_DIST = 18,     /* f32 [N*R] Agent::sensor_hits_[r].norm()               */
    OKENV_F_COUNT = 19,
    /* rollout bookkeeping; exist after okenv_tracker_create (not part of okenv_state_view / snapshots) */
    OKENV_F_REWARD = 19,         /* f32  reward of the last okenv_tracker_update              */
    OKENV_F_FITNESS = 20,        /* f32  CmaEsAgent::fitness_ / return of the running episode  */
    OKENV_F_TRACK_IDX = 21,      /* i32  prev_track_idx_ (main_eigen.cpp:84); PROGRESS reward only, else 0 */
    OKENV_F_EPISODE_STEPS = 22,  /* u32  updates since the episode began                      */
    OKENV_F_EPISODE_RETURN = 23, /* f32  fitness at the end of the last finished episode     */
    OKENV_F_PREV_CRASHED = 24,   /* u8   crashed_ as the last okenv_tracker_update / _begin saw it (detects re-placed agents) */
    OKENV_F_COUNT_ALL = 25
};

/* Struct-of-pointers form of the per-agent state, for one-call upload/download by the C++ facade.
 * NULL members are skipped. */
typedef struct okenv_state_view {
    float *pos_x, *pos_y, *rot, *speed, *acc, *throttle, *steer;
    uint8_t *mode, *crashed, *timed_out;
    uint32_t *disp_ctr;
    float *disp_x, *disp_y;
    uint8_t *disp_timed_out;
} okenv_state_view;

typedef struct okenv_info {
    int32_t num_agents, num_rays, num_segments;
    int32_t grid_nx, grid_ny;
    float grid_cell;
    int32_t grid_refs;        /* total segment registrations */
    int32_t grid_in_lds;      /* 1: grid + segments staged in LDS per workgroup, 0: read from global memory */
    int32_t lds_bytes;        /* dynamic LDS per workgroup */
    int32_t block_threads;    /* workgroup size of the step kernel */
    int32_t grid_blocks;      /* workgroups per launch */
    int32_t lanes_per_agent;  /* G: power of two >= R, capped at 64 */
    int32_t device;
    int32_t agents_per_block; /* > 0: tiny population, that many agents per workgroup (the other lanes only stage)   */
    int32_t packed_resident;  /* 1: a resident step kernel is serving okenv_step_packed right now                    */
    int32_t packed_resident_steps; /* okenv_step_packed calls served by a resident kernel so far                     */
    int32_t packed_fallbacks; /* ... of which the resident kernel had left: redone by a launch of their own          */
    int32_t compute_units;    /* CUs of the device as HIP reports them (256 on an MI355X in SPX mode): what the launch geometry,  */
                              /* the lane-group rule for small populations and the tail-kernel hand-over point are sized by     */
    int32_t front_back_bytes; /* > 0: the segment set is split (ok_grid.h): bytes of the [front | back] images the cooperative     */
                              /* kernel stages instead of the combined image; 0: no split (not a track, or OKENV_FRONT_BACK=0)     */
    int32_t back_segments;    /* segments in the back image: looked at only by rays whose origin is not certified / ambiguous walks */
} okenv_info;

/* ---- lifetime ------------------------------------------------------------------------------------ */

/*
 * Replaces: TrackSegments::uploadToDevice (Environment/TrackSegments.cu:69-76) + CollisionChecker::Impl
 * constructor (Environment/CollisionChecker.cu:76-86) + Environment's displacement_stats_.resize
 * (Environment/Environment.cpp:61).  `segments_xyxy` is the reference's Segment2d[S] layout
 * (x1,y1,x2,y2); `ray_angles_deg` is Agent::sensor_ray_angles_ (all agents share one fan,
 * CollisionChecker.cu:82).  All state starts zeroed (mode VELOCITY, not crashed).
 * `grid_cell` <= 0 selects the default cell edge.
 * A segment set in TrackSegments order whose boundary polylines close (every track the reference builds) is additionally split
 * into the segments a ray from between the inner boundaries can hit first and the rest (the outer polylines, 3 px behind the
 * inner ones): the step kernels look at the rest only for rays whose origin is not certified to lie between the inner boundaries
 * -- same first hit, bit for bit, about half the points per ray (okenv_info.front_back_bytes / back_segments,
 * okenv_work_stats_split; environment variable OKENV_FRONT_BACK=0 switches it off).  Any other segment set is stepped as before.
 */
OKENV_API int okenv_create(okenv_t *out, const float *segments_xyxy, int32_t num_segments, int32_t num_agents,
                           int32_t num_rays, const float *ray_angles_deg, int32_t device, uint32_t flags,
                           float grid_cell);
/* Replaces ~CollisionChecker / ~TrackSegments (Environment/CollisionChecker.cu:88-94, TrackSegments.cu:44-51). */
OKENV_API int okenv_destroy(okenv_t h);
OKENV_API int okenv_get_info(okenv_t h, okenv_info *out);
/* Message for the last failure on `h` (or the last create failure if h is NULL).  Never NULL. */
OKENV_API const char *okenv_last_error(okenv_t h);
/* Agent::sensor_offset_ (Environment/Agent.h:61), shared by all agents; default 0. */
OKENV_API int okenv_set_sensor_offset(okenv_t h, float offset);
/* Centre line + headings (RaceTrack::track_data_points_.x_m/y_m, headings_), needed by the nearest-index
 * query and by the on-device reset in okenv_rollout_random. */
OKENV_API int okenv_set_centerline(okenv_t h, const float *x, const float *y, const float *heading_deg, int32_t num_points);
/* Use an externally owned hipStream_t (e.g. torch's current stream, or a stream being captured into a hipGraph)
 * instead of the private one.  Does not synchronise: ordering against work already queued is the caller's. */
OKENV_API int okenv_set_stream(okenv_t h, void *hip_stream);
OKENV_API int okenv_sync(okenv_t h);

/* ---- state access -------------------------------------------------------------------------------- */

OKENV_API int okenv_set_field(okenv_t h, int32_t field, const void *src);
OKENV_API int okenv_get_field(okenv_t h, int32_t field, void *dst); /* synchronises */
OKENV_API int okenv_upload_state(okenv_t h, const okenv_state_view *host_view);
OKENV_API int okenv_download_state(okenv_t h, const okenv_state_view *host_view); /* synchronises */
/* Agent::current_action_ for every agent (what callers write before Environment::step, Template/main.cpp:107-110). */
OKENV_API int okenv_set_actions(okenv_t h, const float *throttle, const float *steer);
/* Agent::reset (Environment/Agent.cpp:123-135) for agents idx[0..n): pose set, speed/acc/action zeroed,
 * crashed/timed_out cleared; DisplacementStats untouched, as in the reference.  Host arrays. */
OKENV_API int okenv_reset_agents(okenv_t h, const int32_t *idx, const float *x, const float *y, const float *rot_deg, int32_t n);
/* ---- device-side Environment::resetAgent ----------------------------- */

/* the booleans of Environment::resetAgent(agent, pick_random_point, randomize_lane, randomize_heading)
 * (Environment/Environment.h:47-50); values shared with include/okenv_math.h (OK_RESET_*) */
#define OKENV_RESET_RANDOM_POINT 1u
#define OKENV_RESET_RANDOM_LANE 2u
#define OKENV_RESET_RANDOM_HEADING 4u
#define OKENV_RESET_ONLY_DONE 8u /* batch call only: skip agents whose crashed_ flag is clear */

/* RaceTrack::left_bound_inner_ / right_bound_inner_ (Environment/RaceTrack.h:77) as xy pairs, what the lane
 * randomisation interpolates between (Environment/Environment.cpp:108-113).  Host or device pointers. */
OKENV_API int okenv_set_lane_bounds(okenv_t h, const float *left_inner_xy, const float *right_inner_xy, int32_t num_points);
/* Environment::resetAgent (Environment/Environment.cpp:79-122) for agents idx[0..n) (idx NULL: all agents, n ignored),
 * on the device.  raylib's GetRandomValue is replaced by Philox4x32 keyed (seed; agent_base + agent, epoch), see
 * ok_draw_reset in okenv_math.h; entry j of the call takes the reference's static call counter as epoch + j, so a loop
 * `for (agent : agents) env.resetAgent(agent, ...)` is one call.  idx may be a host or a device pointer.
 * Without OKENV_RESET_RANDOM_POINT every agent goes to RaceTrack::kStartingIdx with the track heading. */
OKENV_API int okenv_reset_random(okenv_t h, const int32_t *idx, int32_t n, uint32_t flags, uint32_t seed, uint32_t epoch,
                                 uint32_t agent_base);
/* Per-agent auto-reset for continuous training: while enabled, okenv_step and okenv_rollout_policy begin every step
 * by applying resetAgent(flags) to the agents whose crashed_ flag is set, drawing from Philox (seed; agent_base +
 * agent, step count) with call-counter parity agent + step count; that step then runs with the zeroed action, i.e.
 * it is the "initial observation" step callers take after a reset (RLRacers/PPO/ppo_sim.cpp:53-60).  The flags
 * of the crash stay readable until that next step. */
OKENV_API int okenv_set_auto_reset(okenv_t h, int32_t enabled, uint32_t flags, uint32_t seed, uint32_t agent_base);
/* Environment steps taken so far by okenv_step / okenv_rollout_policy on this handle (the auto-reset epoch).  While
 * auto-reset is on the count lives on the device and is advanced on the stream behind every step, so a captured
 * hipGraph of step launches replays with advancing epochs; reading it then synchronises. */
OKENV_API int okenv_get_step_count(okenv_t h, uint32_t *out);
OKENV_API int okenv_set_step_count(okenv_t h, uint32_t value);

/* ---- rollout bookkeeping of the current-API population callers -------- */

/* The per-step loop the living callers run after env.step() (CovarianceMatrixAdaptationEvolution/main_eigen.cpp:
 * 143-158, RLRacers/PPO/ppo_sim.cpp:73-88), for all agents on the device. */
#define OKENV_REWARD_STEP 0     /* +1 per step for every agent, crashed or not (ppo_sim.cpp:77-80)                   */
#define OKENV_REWARD_PROGRESS 1 /* |curr - prev nearest centre-line index| while alive, fitness := 0 once timed out  */
                                /* (main_eigen.cpp:147-158; the index difference is NOT wrapped at the lap seam)     */
OKENV_API int okenv_tracker_create(okenv_t h, int32_t reward_kind);
/* Start of an episode for every agent: prev_track_idx_ = findNearestTrackIndexBruteForce(pos_), fitness_ = 0
 * (main_eigen.cpp:128-133, CmaEsAgent::reset :70-74).  Call after the initial-observation step. */
OKENV_API int okenv_tracker_begin(okenv_t h);
/* The bookkeeping after one Environment::step.  An agent that was crashed at the previous update and is not now has
 * been re-placed (okenv_reset_random or auto-reset): its episode restarts here with reward 0 (that step was its
 * initial observation).  An agent that crashes in this step gets OKENV_F_EPISODE_RETURN := fitness. */
OKENV_API int okenv_tracker_update(okenv_t h);

/* The CMA-ES candidates' controllers on the device (CovarianceMatrixAdaptationEvolution/Controller.cpp:3-23: fc1 rays ->
 * hidden, fc2 hidden -> hidden / 2, fc3 hidden / 2 -> 2, tanh after each; main_eigen.cpp:18-19 uses hidden = 16), one
 * parameter vector per agent in the order of torch's parameters() (Controller.cpp:36-53).  hidden: even, 2..64. */
OKENV_API int okenv_controller_create(okenv_t h, int32_t hidden);
OKENV_API int okenv_controller_num_params(okenv_t h, int32_t *out);
/* Controller::set_params for every agent: params[num_agents][num_params], host or device pointer (e.g. the solver's
 * sample tensor). */
OKENV_API int okenv_controller_set_params(okenv_t h, const float *params);
/* CmaEsAgent::updateAction for every agent (main_eigen.cpp:58-68): input ||sensor_hits_[i]|| / kSensorRange of the last
 * step, throttle_delta = throttle, steering_delta = output[0] * steering_scale (100 and 5 in the reference).  One kernel on
 * the handle's stream, no synchronisation: it can be captured into a HIP graph next to okenv_step. */
OKENV_API int okenv_controller_act(okenv_t h, float throttle, float steering_scale);
/* The inner loop of the CMA-ES racers (main_eigen.cpp:135-160), n_steps iterations in ONE launch: for every agent
 * { CmaEsAgent::updateAction (= okenv_controller_act); Environment::step; the fitness bookkeeping (= okenv_tracker_update) } with
 * the controller, the step and the bookkeeping fused into the step kernel -- same results, bit for bit, as the three calls made
 * n_steps times.  Needs okenv_controller_create, okenv_tracker_create (either reward kind) and the centre line.  Inside an
 * episode (okenv_episode_begin / _compact / _end, below) later launches cover only the agents that can still change and
 * okenv_episode_end returns the loop's own length, as for okenv_rollout_policy; episodes need OKENV_REWARD_PROGRESS (the +1
 * reward keeps counting for crashed agents, which an episode no longer steps).  hidden <= 4 x the handle's lanes per agent. */
OKENV_API int okenv_rollout_controller(okenv_t h, int32_t n_steps, float throttle, float steering_scale);

/* ---- zero-copy access for device-side callers ------------------------ */

/* Device address and size of one library-owned struct-of-arrays field (okenv_field), valid for the handle's lifetime.
 * Work on it must be ordered against the handle's stream (okenv_set_stream / okenv_sync).  This is what the batched
 * Python binding wraps into tensors instead of copying sensor_hits_ / crashed_ out per step the way
 * Pybind/bindings.cpp:36-52 round-trips one agent. */
OKENV_API int okenv_field_device_ptr(okenv_t h, int32_t field, void **ptr, uint64_t *bytes);

/* Agent::sensor_hits_ as interleaved (x,y) pairs [N*R*2], Agent::sensor_hits_[r].norm() [N*R], and
 * crashed_/timed_out_ as bit0/bit1 of one byte per agent.  Synchronise. */
OKENV_API int okenv_get_hits(okenv_t h, float *out_xy);
OKENV_API int okenv_get_distances(okenv_t h, float *out);
OKENV_API int okenv_get_flags(okenv_t h, uint8_t *out);

/* ---- packed host exchange for callers that keep Agent objects on the host (the C++ facade) -------------------- */

/* One agent's mutable state as a record: the population crosses PCIe in ONE copy each way per step instead of one
 * copy per field (Environment::step of the facade was 40 small copies = 370 us for a single agent before this). */
typedef struct okenv_agent_record {
    float    pos_x, pos_y, rot, speed, acc, throttle, steer; /* Agent::pos_, rot_, speed_, acceleration_, current_action_ */
    float    disp_x, disp_y;                                 /* DisplacementStats::init_pos                              */
    uint32_t disp_ctr;                                       /* DisplacementStats::displacement_ctr                      */
    uint8_t  mode, crashed, timed_out, disp_timed_out;       /* movement_mode_, crashed_, timed_out_, displacement_timed_out */
} okenv_agent_record;                                        /* 44 bytes */

#define OKENV_PACKED_WITH_STATS 1u   /* the DisplacementStats members travel too (Environment::step); otherwise the   */
                                     /* device keeps its own and the record's are left untouched                     */
#define OKENV_PACKED_COLLIDE_ONLY 2u /* CollisionChecker::checkCollision(): no kinematics, no standstill bookkeeping  */
/* Upload `in[num_agents]`, run one Environment::step (or only the collision pass), download the new state into
 * `out[num_agents]` (may alias `in`) and Agent::sensor_hits_ as interleaved (x, y) pairs into sensor_hits_xy
 * [num_agents * num_rays * 2].  Host pointers; synchronises.
 * Up to 64 single-wave agents, OKENV_PACKED_WITH_STATS steps: when such calls follow each other within 100 us, a
 * resident step kernel takes them over (no launch per step; okenv_info.packed_resident).  It leaves after 300 us
 * without a step and before any other call on the handle does its work; results are the same bits either way.
 * Environment variable OKENV_RESIDENT: 0 = never, 1 = from the first eligible call on. */
OKENV_API int okenv_step_packed(okenv_t h, const okenv_agent_record *in, okenv_agent_record *out, float *sensor_hits_xy,
                                uint32_t flags);

/* ---- the hot path -------------------------------------------------------------------------------- */

/* Environment::step() x n_steps (Environment/Environment.cpp:125-149, minus render): move + standstill for
 * non-crashed agents, then the collision pass for all.  Uses the actions currently stored. */
OKENV_API int okenv_step(okenv_t h, int32_t n_steps);
/* CollisionChecker::checkCollision() alone (Environment/CollisionChecker.cu:197-200): ray build, first-hit
 * raycast, hit transform, crash flag; no kinematics. */
OKENV_API int okenv_collide(okenv_t h);
/* The bench driver loop on the device (shape of RLRacers/GuidedCostLearning/test.cpp:100-117; recipe in
 * SURVEY.md section 8d): per step, crashed agents are re-placed on a Philox-chosen centre-line point, every
 * agent draws throttle~U[0,100) and steer~U[-5,5) from Philox4x32 keyed (seed; agent_base+i, step_base+s),
 * then Environment::step.  Requires okenv_set_centerline. */
OKENV_API int okenv_rollout_random(okenv_t h, int32_t n_steps, uint32_t seed, uint32_t agent_base, uint32_t step_base);
/* Bench initial state: agent i on centre-line index ((agent_base+i)*2654435761 mod 2^32) mod P with the track
 * heading, speed 0, DisplacementStats and hit points zeroed, all agents in `mode`. */
OKENV_API int okenv_init_bench_state(okenv_t h, uint32_t agent_base, int32_t mode);
/* RaceTrack::findNearestTrackIndexBruteForce (Environment/RaceTrack.cpp:16-31) for n query points
 * (host or device pointers), or for every agent's current position when qx == NULL (n ignored). */
OKENV_API int okenv_nearest_track_idx(okenv_t h, const float *qx, const float *qy, int32_t n, int32_t *out);

/* ---- EvolutionaryRacer on the device (BASELINE configs 3 and 4) --------- */

/* genetic::Network() for every agent (EvolutionaryRacer/Network.hpp:99-107): (R+2) -> hidden -> 6, no biases, weights
 * U[-1,1) from Philox keyed (seed; agent_base+i, weight) in place of the unseeded Eigen Random().  hidden <= 32
 * (30 in the reference).  Needs 5 <= R <= 64.  Weights live on the device, okenv_policy_mlp_weights_per_agent()
 * floats per agent in the padded layout of include/okenv_math.h. */
OKENV_API int okenv_policy_mlp_create(okenv_t h, int32_t hidden, uint32_t seed, uint32_t agent_base);
OKENV_API int32_t okenv_policy_mlp_weights_per_agent(okenv_t h);
OKENV_API int okenv_policy_mlp_get_weights(okenv_t h, float *out);       /* host or device pointer */
OKENV_API int okenv_policy_mlp_set_weights(okenv_t h, const float *in);  /* host or device pointer */
/* n_steps x { GeneticAgent::updateAction for every agent (GeneticAgent.hpp:37-107, Network::infer Network.hpp:119-155)
 * from the previous step's observation; Environment::step } -- the inner loop of genetic_learner_sim.cpp:76-95, fused
 * into the step kernel.  Call okenv_step(h, 1) once after a reset for the initial observation (:75). */
OKENV_API int okenv_rollout_policy(okenv_t h, int32_t n_steps);
/* number of agents with crashed_ == false (the loop's all_done test, genetic_learner_sim.cpp:85-92) */
OKENV_API int okenv_alive_count(okenv_t h, int32_t *out);

/* ---- episodes: "step everybody until every agent has crashed" ----------------------------------------------------
 * The inner loop of both population callers (EvolutionaryRacer/genetic_learner_sim.cpp:76-95,
 * RLRacers/Q_Learning/q_racer_sim.cpp:156-190) runs until the step T in which the LAST agent crashes; most agents crash
 * long before that (a tenth to a fifth of the agent-steps of such a loop belong to agents still alive).  Between
 * okenv_episode_begin and okenv_episode_end the policy rollouts (okenv_rollout_policy, okenv_rollout_q) therefore
 *   - step only the agents that can still change: after an agent's first step as a crashed agent nothing about it changes
 *     any more (it does not move, its standstill counter does not tick, its rays keep their stale hit points, the MLP policy
 *     sees the same inputs), so it is dropped -- by a wave of the running launch as soon as all its agents are done, and from
 *     the launch grid by okenv_episode_compact;
 *   - may overrun T (launches end where the caller's n_steps end): okenv_episode_end finds T and leaves every agent, every
 *     Q table and the step count exactly as the reference's loop leaves them after step T, whatever the launches' lengths.
 *     Q-learning's per-step update of CRASHED agents (epsilon-greedy draw + learn with reward -200, q_racer_sim.cpp:158-182)
 *     is replayed per agent for its steps crash+1 .. T at that point.
 * Typical loop:   okenv_episode_begin(h);  okenv_episode_tail_limit(h, &tail);  listed = N;
 *                 do { okenv_rollout_policy(h, listed <= tail ? all the steps still allowed : n);
 *                      okenv_episode_compact(h, &alive, &listed); } while (alive > 0 && more steps allowed);
 *                 okenv_episode_end(h, &steps, &live);
 * Needs auto-reset off.  Any call that changes agent state from outside (set/upload/reset/step without a policy) ends the
 * episode without the end-of-episode corrections.  While a controller episode is running (okenv_rollout_controller inside
 * okenv_episode_begin / _end) okenv_tracker_begin, okenv_tracker_update and okenv_controller_set_params return OKENV_ERR_STATE:
 * the fused rollout carries the bookkeeping itself.  A population no larger than okenv_episode_tail_limit is listed from
 * okenv_episode_begin on (its first rollout already runs one agent per workgroup). */
OKENV_API int okenv_episode_begin(okenv_t h);
/* Rebuilds the list of agents the next rollouts step; *alive_out = agents with crashed_ == false (the loop's all_done test),
 * *listed_out = agents still stepped (alive ones + those that crashed in the last step taken).  Either may be NULL. */
OKENV_API int okenv_episode_compact(okenv_t h, int32_t *alive_out, int32_t *listed_out);
/* Longest list (okenv_episode_compact's *listed_out) that is stepped one agent per workgroup on this handle (0: never).  Such
 * a workgroup leaves as soon as its agent is done, so from there on the caller may ask for ALL the steps it still allows in
 * one rollout call: the launch ends with the step in which the last agent crashes, and no launch boundary is paid any more. */
OKENV_API int okenv_episode_tail_limit(okenv_t h, int32_t *out);
/* *steps_out = T (steps of the reference's loop; all steps taken if somebody is still alive), *live_agent_steps_out = sum over
 * the steps of the agents that entered the step alive.  Either may be NULL. */
OKENV_API int okenv_episode_end(okenv_t h, int32_t *steps_out, uint64_t *live_agent_steps_out);
/* Agents whose position lies outside the raycast grid's box (the track's bounding box plus a small pad): *alive_off_grid those
 * with crashed_ == false, *all_off_grid all of them (either may be NULL).  The crash test is lidar-only and a step can be 1.6 px
 * long, so an agent can tunnel through both boundary polylines (Environment/CollisionChecker.cu:167-171);
 * outside, nothing is within sensor range and only the standstill timeout (Environment.cpp:16-39) can still end it -- at speed it
 * never does, and "until every agent has crashed" then runs into the caller's step cap.  Measurement aid; synchronises. */
OKENV_API int okenv_off_grid_count(okenv_t h, int32_t *alive_off_grid, int32_t *all_off_grid);
/* Agent::reset of EVERY agent to one pose (genetic_learner_sim.cpp:65-70) */
OKENV_API int okenv_reset_all(okenv_t h, float x, float y, float rot_deg);
/* assignScores (EvolutionaryRacer/MiscUtils.hpp:64-71): score = nearest centre-line index as float, kept on the
 * device for okenv_ga_select_mate; `out` (N floats, host or device) may be NULL. */
OKENV_API int okenv_ga_scores(okenv_t h, float *out);
/* Where that score vector lives: the device address of the N floats okenv_ga_scores fills (library-owned, valid for the
 * handle's lifetime), and the hipStream_t the handle enqueues its work on.  With the two a multi-GPU caller runs its one
 * collective -- the per-generation all-gather of this vector, SURVEY.md section 8e -- straight from device memory and in
 * stream order behind the kernel that wrote it:  okenv_ga_scores(h, NULL); ncclAllGather(scores, colony, N, ncclFloat, comm,
 * stream);  (openkitchen_amd/csrc/apps/genetic_learner_sim.cpp --gpus N). */
OKENV_API int okenv_ga_scores_device(okenv_t h, const float **ptr);
OKENV_API int okenv_get_stream(okenv_t h, void **hip_stream);
/* chooseAndMateAgents (EvolutionaryRacer/Mating.hpp:108-166) with mate2AgentsSelective (:52-99): the 5 best agents
 * (ties to the lower index) become parents; offspring 0 clones the best, offspring 1 is the best mated with itself,
 * every other offspring draws two different parents proportionally to score; per weight 10 % mutation to U[-1,1),
 * else the dominant parent's weight with probability 0.75.  Draws come from Philox keyed (seed; agent_base+offspring,
 * weight, generation) instead of std::random_device.  parents_out (5 ints, host) may be NULL. */
OKENV_API int okenv_ga_select_mate(okenv_t h, uint32_t seed, uint32_t generation, uint32_t agent_base, int32_t *parents_out);

/* ---- RLRacers/Q_Learning on the device (BASELINE config 5) ----------------------- */

/* One 243 x 3 table per agent, every entry numeric_limits<float>::lowest() (RLRacers/Q_Learning/QAgent.hpp:31-36,64-68).
 * The state uses five rays; with a fan of more than five rays the ones nearest to -70, -30, 0, +30, +70 degrees are taken
 * (ties to the lower index) -- the reference fan is exactly those five (QAgent.hpp:56-62).  Needs R >= 5 and the LDS form. */
OKENV_API int okenv_q_create(okenv_t h);
/* Start of an episode (q_racer_sim.cpp:129-154): every agent reset onto centre-line point `reset_idx` with the track
 * heading, prev_track_idx = that point's nearest index, one Environment::step for the initial observation, current
 * state = discretizeState().  All agents must be in VELOCITY mode (the action map sets the speed). */
OKENV_API int okenv_q_begin_episode(okenv_t h, int32_t reset_idx);
/* n_steps x { updateAction (epsilon-greedy, QAgent.hpp:98-119); Environment::step; discretizeState; reward
 * (QAgent.hpp:150-168, nearest centre-line index); learn (QAgent.hpp:121-138) } for every agent, crashed ones included, as
 * q_racer_sim.cpp:158-182 does.  Random draws: Philox keyed (seed; agent_base+i, step_base+s).  Inside an episode
 * (okenv_episode_begin) the crashed agents' share of this is deferred to okenv_episode_end, with the same result; epsilon,
 * seed and agent_base must then stay the same for the whole episode and step_base advance with the steps taken. */
OKENV_API int okenv_rollout_q(okenv_t h, int32_t n_steps, float epsilon, uint32_t seed, uint32_t agent_base, uint32_t step_base);
OKENV_API int okenv_q_get_table(okenv_t h, float *out);      /* [N][243][3], host or device pointer */
OKENV_API int okenv_q_set_table(okenv_t h, const float *in);
/* current_state_idx_, current_action_idx_, prev_track_idx_ per agent (host arrays, any may be NULL) */
OKENV_API int okenv_q_get_state(okenv_t h, int32_t *state, int32_t *action, int32_t *prev_idx);

/* shareCumulativeKnowledge (q_racer_sim.cpp:24-75; off by default in the reference, :16): `okenv_q_table_sums` gives, per
 * (state, action), the sum of the valid entries over this handle's agents and their count (729 floats each, host or
 * device pointers); `okenv_q_assign_mean` sets every agent's table to sum/count (entries with count 0 stay invalid).  A
 * multi-GPU caller all-reduces the two vectors between the calls (5.8 KB); `okenv_q_share_knowledge` does both locally. */
OKENV_API int okenv_q_table_sums(okenv_t h, float *sum, float *count);
OKENV_API int okenv_q_assign_mean(okenv_t h, const float *sum, const float *count);
OKENV_API int okenv_q_share_knowledge(okenv_t h);

/* ---- measurement --------------------------------------------------------------------------------- */

/* When enabled, every step/collide/rollout launch is bracketed by HIP events on the handle's stream. */
OKENV_API int okenv_set_timing(okenv_t h, int32_t enabled);
/* Sum of the bracketed kernel durations [ms] and their count since the last call; synchronises, then clears. */
OKENV_API int okenv_get_timing(okenv_t h, double *total_ms, uint64_t *launches);

/* ---- host-side track construction (RaceTrack + TrackSegments, no GPU involved) -------------------- */

/* RaceTrack::RaceTrack(csv) (Environment/RaceTrack.cpp:3-14). */
OKENV_API int okenv_track_load(okenv_track_t *out, const char *csv_path);
OKENV_API int okenv_track_free(okenv_track_t t);
OKENV_API int32_t okenv_track_num_points(okenv_track_t t);
OKENV_API int32_t okenv_track_num_segments(okenv_track_t t);
/* which: 0 x_m, 1 y_m, 2 w_tr_right_m, 3 w_tr_left_m, 4 headings_ (P floats);
 *        5 left_bound_inner_, 6 left_bound_outer_, 7 right_bound_inner_, 8 right_bound_outer_ (2P floats, xy) */
OKENV_API int okenv_track_get(okenv_track_t t, int32_t which, float *out);
/* RaceTrack::getNearestDistanceToTrackBoundary and RaceTrack::getDistanceToLaneCenter (Environment/RaceTrack.h:36,39,
 * RaceTrack.cpp:33-72) for n query points (host pointers; either output may be NULL).  Host-side, like the reference's. */
OKENV_API int okenv_track_queries(okenv_track_t t, const float *qx, const float *qy, int32_t n, float *out_boundary_distance,
                                  float *out_lane_center_ratio);
/* TrackSegments::TrackSegments (Environment/TrackSegments.cu:6-42): 4*P segments, x1,y1,x2,y2 each. */
OKENV_API int okenv_track_segments(okenv_track_t t, float *out_xyxy);

/* Work the broad phase leaves for the population's current poses (measurement aid: SURVEY.md section 8d's S_tested): every
 * live agent's rays walked once through the grid; out[0] = rays, out[1] = exact ray-segment tests (the reference's sweep,
 * Environment/CollisionChecker.cu:49-67, makes num_segments per ray), out[2] = grid cells entered, out[3] = boundary points
 * evaluated by the skip rule.  LDS form of the grid only. */
OKENV_API int okenv_work_stats(okenv_t h, uint64_t out[4]);
/* The same walk as the step kernels make it with the front / back split of the segment set (okenv_info.front_back_bytes > 0;
 * openkitchen_amd/csrc/ok_grid.h: the outer boundary polylines sit in an image of their own and are walked only by rays whose
 * origin is not certified to lie between the inner boundaries, or whose front walk may have missed a crossing): out[0..3] as
 * above, front and back walks together; out[4] rays of a certified origin, out[5] rays whose front walk was ambiguous, out[6] rays
 * that walked the back image too; out[7] unused.  OKENV_ERR_STATE when the segment set has no split. */
OKENV_API int okenv_work_stats_split(okenv_t h, uint64_t out[8]);

/* ---- device self-checks used by the parity tests --------------------------------------------------- */

/* ok_sincosf evaluated on the GPU (n values, host pointers). */
OKENV_API int okenv_debug_sincos(int32_t device, const float *x, float *s, float *c, int32_t n);
/* First-hit parameter t for n arbitrary rays (origin, angle [rad]) through the handle's grid (host pointers). */
OKENV_API int okenv_debug_cast_rays(okenv_t h, const float *ox, const float *oy, const float *angle_rad, int32_t n, float *out_t);

#ifdef __cplusplus
}
#endif
#endif /* OKENV_H */
