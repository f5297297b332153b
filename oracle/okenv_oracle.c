/*
 * okenv_oracle.c -- CPU restatement of OpenKitchen's Environment step path.  TEST INFRASTRUCTURE.
 *
 * This file is the parity oracle and the CPU baseline ("port") of the project.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing under
 * openkitchen_amd/ links, imports or calls it, and the product path has no CPU fallback.
 *
 * It restates, in plain C and in the reference's fp32 operation order, compiled with
 * `-O2 -ffp-contract=off` (oracle/Makefile):
 *   RaceTrack construction ........ Environment/RaceTrack.cpp:3-14,87-114,127-164,166-196,198-229,257-307
 *   nearest centre-line index ..... Environment/RaceTrack.cpp:16-31
 *   TrackSegments flattening ...... Environment/TrackSegments.cu:6-42,53-67
 *   Agent kinematics .............. Environment/Agent.cpp:21-47,82-98,108-119,123-135
 *   standstill FSM ................ Environment/Environment.h:17-27, Environment/Environment.cpp:16-39
 *   Environment::step order ....... Environment/Environment.cpp:125-149
 *   ray build / raycast / epilogue  Environment/CollisionChecker.cu:8-35,37-71,113-174
 *
 * Pinning: the reference has no tests or golden vectors for this path.  The
 * restatement is pinned against the reference's own compilable translation units
 * (Environment/Agent.cpp, Environment/RaceTrack.cpp built into oracle/_ref/ by oracle/Makefile) in
 * tests/test_oracle_vs_ref.py, and through committed fixtures generated from them
 * (tests/golden/, generator tests/golden/make_golden.py).  The raycast itself exists in the reference
 * only as a CUDA kernel that cannot be built or run here; for it the oracle is a line-by-line
 * restatement and the pin is structural (known-answer geometric cases in tests/test_oracle_raycast.py).
 *
 * Trigonometry: `trig_mode` 0 uses ok_sincosf (include/okenv_math.h), the function the HIP kernels
 * use -- this is the parity definition.  `trig_mode` 1 uses glibc sinf/cosf, which is what the
 * reference's host code calls; it is used to validate kinematics bit-for-bit against the reference
 * objects and to bound the distance between the two definitions.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#include "../include/okenv_math.h"

#define ORACLE_API __attribute__((visibility("default")))

static int g_trig_mode = 0;

ORACLE_API void oracle_set_trig_mode(int mode) { g_trig_mode = mode; }
ORACLE_API int oracle_get_trig_mode(void) { return g_trig_mode; }

static inline void trig(float x, float *s, float *c)
{
    if (g_trig_mode == 1) {
        *s = sinf(x);
        *c = cosf(x);
    } else {
        ok_sincosf(x, s, c);
    }
}

ORACLE_API void oracle_sincosf(const float *x, float *s, float *c, int n)
{
    for (int i = 0; i < n; ++i) ok_sincosf(x[i], &s[i], &c[i]);
}

ORACLE_API void oracle_tanhf(const float *x, float *t, int n)
{
    for (int i = 0; i < n; ++i) t[i] = ok_tanhf(x[i]);
}

ORACLE_API void oracle_philox(uint32_t seed, uint32_t agent, uint32_t step, float *thr, float *steer, uint32_t *word)
{
    ok_random_action a = ok_draw_random_action(seed, agent, step);
    *thr = a.throttle;
    *steer = a.steer;
    *word = a.reset_word;
}

/* ============================================================================================ */
/* RaceTrack                                                                                     */
/* ============================================================================================ */

typedef struct oracle_track {
    int P;
    float *x, *y, *wr, *wl;      /* centre line and (scaled) half widths, RaceTrack::TrackData */
    float *heading;              /* degrees */
    float *li, *lo, *ri, *ro;    /* boundaries, xy interleaved, P points each */
} oracle_track;

ORACLE_API void oracle_track_free(oracle_track *t)
{
    if (!t) return;
    free(t->x); free(t->y); free(t->wr); free(t->wl); free(t->heading);
    free(t->li); free(t->lo); free(t->ri); free(t->ro);
    free(t);
}

/* Environment/RaceTrack.cpp:87-114 */
static void gradient(const float *in, float *out, int n)
{
    for (int i = 0; i < n; ++i) {
        if (i == 0) out[i] = in[i + 1] - in[i];
        else if (i == n - 1) out[i] = in[i] - in[i - 1];
        else out[i] = (in[i + 1] - in[i - 1]) / 2.0f;
    }
}

/* Environment/RaceTrack.cpp:166-196 */
static void extents(const float *x, const float *y, int n, float *minx, float *miny, float *maxx, float *maxy)
{
    *minx = FLT_MAX; *miny = FLT_MAX; *maxx = -FLT_MAX; *maxy = -FLT_MAX;
    for (int i = 0; i < n; ++i) {
        if (x[i] < *minx) *minx = x[i];
        if (x[i] > *maxx) *maxx = x[i];
    }
    for (int i = 0; i < n; ++i) {
        if (y[i] < *miny) *miny = y[i];
        if (y[i] > *maxy) *maxy = y[i];
    }
}

static float clamp_width(float w)
{
    /* Environment/RaceTrack.cpp:138-160: min(max(4, w) * 3, 17) */
    float m = (4.0f < w) ? w : 4.0f; /* std::max(4.0f, w) returns w only if 4 < w */
    float s = m * 3.0f;
    return (17.0f < s) ? 17.0f : s; /* std::min(s, 17): returns 17 only if 17 < s */
}

/* Environment/RaceTrack.cpp:127-164 */
static int parse_csv(const char *path, oracle_track *t)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int cap = 2048, n = 0;
    t->x = malloc(sizeof(float) * cap); t->y = malloc(sizeof(float) * cap);
    t->wr = malloc(sizeof(float) * cap); t->wl = malloc(sizeof(float) * cap);
    char line[1024];
    if (!fgets(line, sizeof line, f)) { fclose(f); return -2; } /* header */
    while (fgets(line, sizeof line, f)) {
        char *p = line;
        if (*p == '\n' || *p == '\0' || *p == '\r') continue;
        float v[4];
        for (int k = 0; k < 4; ++k) {
            char *end;
            v[k] = strtof(p, &end); /* std::stof == strtof */
            p = strchr(p, ',');
            if (p) ++p; else if (k < 3) { fclose(f); return -3; }
        }
        if (n == cap) {
            cap *= 2;
            t->x = realloc(t->x, sizeof(float) * cap); t->y = realloc(t->y, sizeof(float) * cap);
            t->wr = realloc(t->wr, sizeof(float) * cap); t->wl = realloc(t->wl, sizeof(float) * cap);
        }
        t->x[n] = v[0]; t->y[n] = v[1];
        t->wr[n] = clamp_width(v[2]);
        t->wl[n] = clamp_width(v[3]);
        ++n;
    }
    fclose(f);
    t->P = n;
    return 0;
}

ORACLE_API oracle_track *oracle_track_load(const char *csv_path)
{
    oracle_track *t = calloc(1, sizeof *t);
    if (parse_csv(csv_path, t) != 0 || t->P < 2) { oracle_track_free(t); return NULL; }
    const int P = t->P;
    /* Environment/RaceTrack.cpp:198-229 centerTrackPointsToWindow(extent, 1600, 1400) */
    float minx, miny, maxx, maxy;
    extents(t->x, t->y, P, &minx, &miny, &maxx, &maxy);
    const float window_w = 1600.0f, window_h = 1400.0f; /* Environment/Typedefs.h:7-8 */
    const float track_w = maxx - minx;
    const float track_h = maxy - miny;
    float sfx = window_w / track_w;
    float sfy = window_h / track_h;
    float sf = (sfy < sfx) ? sfy : sfx; /* std::min(sfx, sfy) */
    const float kScreenFitScale = 0.9f; /* constexpr float{0.9} */
    sf *= kScreenFitScale;
    for (int i = 0; i < P; ++i) {
        t->x[i] *= sf; t->y[i] *= sf; t->wl[i] *= sf; t->wr[i] *= sf;
    }
    extents(t->x, t->y, P, &minx, &miny, &maxx, &maxy);
    const float dcx = (window_w / 2.0f) - ((maxx + minx) / 2.0f);
    const float dcy = (window_h / 2.0f) - ((maxy + miny) / 2.0f);
    for (int i = 0; i < P; ++i) { t->x[i] += dcx; t->y[i] += dcy; }

    /* Environment/RaceTrack.cpp:257-307 calculateTrackLanes */
    float *dx = malloc(sizeof(float) * P), *dy = malloc(sizeof(float) * P);
    gradient(t->x, dx, P);
    gradient(t->y, dy, P);
    t->heading = malloc(sizeof(float) * P);
    for (int i = 0; i < P; ++i) {
        const float mag = sqrtf(dx[i] * dx[i] + dy[i] * dy[i]);
        dx[i] /= mag;
        dy[i] /= mag;
        /* :278  std::atan2(float,float) * 180.0F / M_PI : float*float, then a DOUBLE divide, stored to float */
        const float a = atan2f(dy[i], dx[i]) * 180.0f;
        t->heading[i] = (float)((double)a / M_PI);
    }
    t->li = malloc(sizeof(float) * 2 * P); t->lo = malloc(sizeof(float) * 2 * P);
    t->ri = malloc(sizeof(float) * 2 * P); t->ro = malloc(sizeof(float) * 2 * P);
    const float kBoundaryThickness = 3.0f;
    for (int i = 0; i < P; ++i) {
        t->ri[2 * i] = t->x[i] + t->wr[i] * dy[i];
        t->ri[2 * i + 1] = t->y[i] - t->wr[i] * dx[i];
        t->li[2 * i] = t->x[i] - t->wl[i] * dy[i];
        t->li[2 * i + 1] = t->y[i] + t->wl[i] * dx[i];
        t->ro[2 * i] = t->x[i] + (t->wr[i] + kBoundaryThickness) * dy[i];
        t->ro[2 * i + 1] = t->y[i] - (t->wr[i] + kBoundaryThickness) * dx[i];
        t->lo[2 * i] = t->x[i] - (t->wl[i] + kBoundaryThickness) * dy[i];
        t->lo[2 * i + 1] = t->y[i] + (t->wl[i] + kBoundaryThickness) * dx[i];
    }
    free(dx); free(dy);
    return t;
}

ORACLE_API int oracle_track_num_points(const oracle_track *t) { return t->P; }

/* which: 0 x, 1 y, 2 w_right, 3 w_left, 4 heading (P floats); 5 li, 6 lo, 7 ri, 8 ro (2P floats) */
ORACLE_API int oracle_track_get(const oracle_track *t, int which, float *out)
{
    const float *src[9] = {t->x, t->y, t->wr, t->wl, t->heading, t->li, t->lo, t->ri, t->ro};
    if (which < 0 || which > 8) return -1;
    const size_t n = (which < 5) ? (size_t)t->P : (size_t)2 * t->P;
    memcpy(out, src[which], n * sizeof(float));
    return 0;
}

/* Environment/TrackSegments.cu:6-42,53-67 -- order LI, LO, RI, RO runs then closers LI, RI, LO, RO.
 * out holds 4*P segments as x1,y1,x2,y2.  Returns the segment count. */
ORACLE_API int oracle_track_segments(const oracle_track *t, float *out)
{
    const int P = t->P;
    int s = 0;
    const float *poly[4] = {t->li, t->lo, t->ri, t->ro};
    for (int q = 0; q < 4; ++q) {
        for (int i = 0; i + 1 < P; ++i) {
            out[4 * s + 0] = poly[q][2 * i]; out[4 * s + 1] = poly[q][2 * i + 1];
            out[4 * s + 2] = poly[q][2 * i + 2]; out[4 * s + 3] = poly[q][2 * i + 3];
            ++s;
        }
    }
    if (P > 1) {
        const float *closers[4] = {t->li, t->ri, t->lo, t->ro};
        for (int q = 0; q < 4; ++q) {
            out[4 * s + 0] = closers[q][2 * (P - 1)]; out[4 * s + 1] = closers[q][2 * (P - 1) + 1];
            out[4 * s + 2] = closers[q][0]; out[4 * s + 3] = closers[q][1];
            ++s;
        }
    }
    return s;
}

/* Environment/RaceTrack.cpp:16-31; Vec2d::distanceSquared Environment/Typedefs.h:41-44 */
ORACLE_API void oracle_nearest_track_idx(const float *cx, const float *cy, int P, const float *qx, const float *qy,
                                         int n, int32_t *out)
{
    for (int j = 0; j < n; ++j) {
        float best = FLT_MAX;
        int bi = 0;
        for (int i = 0; i < P; ++i) {
            const float ddx = qx[j] - cx[i], ddy = qy[j] - cy[i];
            const float d = ddx * ddx + ddy * ddy;
            if (d < best) { best = d; bi = i; }
        }
        out[j] = bi;
    }
}

/* RaceTrack::getNearestDistanceToTrackBoundary (Environment/RaceTrack.cpp:33-51): the nearest inner-boundary POINT, left
 * and right interleaved per index; li / ri are xy pairs. */
ORACLE_API void oracle_boundary_distance(const float *li, const float *ri, int P, const float *qx, const float *qy, int n,
                                         float *out)
{
    for (int j = 0; j < n; ++j) {
        float min_distance = FLT_MAX;
        for (int i = 0; i < P; ++i) {
            float ddx = qx[j] - li[2 * i], ddy = qy[j] - li[2 * i + 1];
            float distance = ddx * ddx + ddy * ddy;
            if (distance < min_distance) min_distance = distance;
            ddx = qx[j] - ri[2 * i];
            ddy = qy[j] - ri[2 * i + 1];
            distance = ddx * ddx + ddy * ddy;
            if (distance < min_distance) min_distance = distance;
        }
        out[j] = sqrtf(min_distance);
    }
}

/* RaceTrack::getDistanceToLaneCenter (Environment/RaceTrack.cpp:53-72): distance to the nearest centre-line point as a
 * fraction of the lane width (left + right) at that point. */
ORACLE_API void oracle_lane_center_distance(const float *cx, const float *cy, const float *w_left, const float *w_right,
                                            int P, const float *qx, const float *qy, int n, float *out)
{
    for (int j = 0; j < n; ++j) {
        float min_distance = FLT_MAX;
        int min_dist_idx = 0;
        for (int i = 0; i < P; ++i) {
            const float ddx = qx[j] - cx[i], ddy = qy[j] - cy[i];
            const float distance = ddx * ddx + ddy * ddy;
            if (distance < min_distance) { min_distance = distance; min_dist_idx = i; }
        }
        const float lane_width = w_left[min_dist_idx] + w_right[min_dist_idx];
        out[j] = sqrtf(min_distance) / lane_width;
    }
}

/* ============================================================================================ */
/* Environment state (SoA mirror of N Agent objects + DisplacementStats + Ray_ hit points)       */
/* ============================================================================================ */

typedef struct oracle_env {
    int N, R, S;
    float *segs;        /* S * 4 */
    float *ray_deg;     /* R, Agent::sensor_ray_angles_ */
    float sensor_offset;
    /* Agent fields */
    float *pos_x, *pos_y, *rot, *speed, *acc, *thr, *steer;
    uint8_t *mode, *crashed, *timed_out;
    /* DisplacementStats */
    uint32_t *disp_ctr;
    float *disp_x, *disp_y;
    uint8_t *disp_to;
    /* Ray_ persistent fields (world-frame hit point), N*R each; zero before the first step
     * (the reference's pinned buffer is uninitialised -- defined as zeros) */
    float *hit_x, *hit_y;
    /* outputs of the epilogue: sensor_hits_ (robot frame) and their norms */
    float *rel_x, *rel_y, *dist;
    /* centre line for resets / nearest index */
    int P;
    float *cx, *cy, *chead;
    /* inner lane boundaries (xy pairs) for resetAgent's lane randomisation; auto-reset configuration and the running
     * step count that serves as its epoch (okenv_set_auto_reset in include/okenv.h) */
    float *lane_l, *lane_r;
    uint32_t reset_flags, reset_seed, reset_agent_base, step_count;
    int auto_reset;
    /* rollout bookkeeping of the CMA-ES / PPO callers (okenv_tracker_* in include/okenv.h) */
    int tracker_kind;
    int32_t *tr_prev_idx;
    float *tr_fitness, *tr_reward, *tr_ep_return;
    uint32_t *tr_ep_steps;
    uint8_t *tr_prev_crashed;
    /* EvolutionaryRacer: per-agent MLP weights (padded layout of okenv_math.h), scores */
    int mlp_hidden;
    float *mlp_w, *score;
    /* RLRacers/Q_Learning: table [N][243][3], state/action/prev index per agent, the five state rays */
    float *q_table;
    int32_t *q_state, *q_action, *q_prev;
    int q_ray[5];
} oracle_env;

#define ALLOC(T, n) ((T *)calloc((size_t)(n), sizeof(T)))

ORACLE_API oracle_env *oracle_env_create(const float *segs, int S, int N, int R, const float *ray_deg)
{
    oracle_env *e = ALLOC(oracle_env, 1);
    e->N = N; e->R = R; e->S = S;
    e->segs = ALLOC(float, 4 * (size_t)S); memcpy(e->segs, segs, sizeof(float) * 4 * (size_t)S);
    e->ray_deg = ALLOC(float, R); memcpy(e->ray_deg, ray_deg, sizeof(float) * R);
    e->sensor_offset = 0.0f; /* Environment/Agent.h:61 */
    e->pos_x = ALLOC(float, N); e->pos_y = ALLOC(float, N); e->rot = ALLOC(float, N);
    e->speed = ALLOC(float, N); e->acc = ALLOC(float, N); e->thr = ALLOC(float, N); e->steer = ALLOC(float, N);
    e->mode = ALLOC(uint8_t, N); e->crashed = ALLOC(uint8_t, N); e->timed_out = ALLOC(uint8_t, N);
    e->disp_ctr = ALLOC(uint32_t, N); e->disp_x = ALLOC(float, N); e->disp_y = ALLOC(float, N);
    e->disp_to = ALLOC(uint8_t, N);
    const size_t NR = (size_t)N * R;
    e->hit_x = ALLOC(float, NR); e->hit_y = ALLOC(float, NR);
    e->rel_x = ALLOC(float, NR); e->rel_y = ALLOC(float, NR); e->dist = ALLOC(float, NR);
    return e;
}

ORACLE_API void oracle_env_destroy(oracle_env *e)
{
    if (!e) return;
    free(e->segs); free(e->ray_deg);
    free(e->pos_x); free(e->pos_y); free(e->rot); free(e->speed); free(e->acc); free(e->thr); free(e->steer);
    free(e->mode); free(e->crashed); free(e->timed_out);
    free(e->disp_ctr); free(e->disp_x); free(e->disp_y); free(e->disp_to);
    free(e->hit_x); free(e->hit_y); free(e->rel_x); free(e->rel_y); free(e->dist);
    free(e->cx); free(e->cy); free(e->chead);
    free(e->lane_l); free(e->lane_r);
    free(e->tr_prev_idx); free(e->tr_fitness); free(e->tr_reward); free(e->tr_ep_return); free(e->tr_ep_steps);
    free(e->tr_prev_crashed);
    free(e->mlp_w); free(e->score);
    free(e->q_table); free(e->q_state); free(e->q_action); free(e->q_prev);
    free(e);
}

ORACLE_API void oracle_env_set_centerline(oracle_env *e, const float *cx, const float *cy, const float *head, int P)
{
    free(e->cx); free(e->cy); free(e->chead);
    e->P = P;
    e->cx = ALLOC(float, P); e->cy = ALLOC(float, P); e->chead = ALLOC(float, P);
    memcpy(e->cx, cx, sizeof(float) * P); memcpy(e->cy, cy, sizeof(float) * P); memcpy(e->chead, head, sizeof(float) * P);
}

ORACLE_API void oracle_env_set_sensor_offset(oracle_env *e, float off) { e->sensor_offset = off; }

ORACLE_API void oracle_env_set_lane_bounds(oracle_env *e, const float *left_inner_xy, const float *right_inner_xy, int P)
{
    free(e->lane_l); free(e->lane_r);
    e->lane_l = ALLOC(float, 2 * (size_t)P); e->lane_r = ALLOC(float, 2 * (size_t)P);
    memcpy(e->lane_l, left_inner_xy, sizeof(float) * 2 * (size_t)P);
    memcpy(e->lane_r, right_inner_xy, sizeof(float) * 2 * (size_t)P);
}

/* field ids shared with include/okenv.h (OKENV_F_*) */
enum { F_POS_X = 0, F_POS_Y, F_ROT, F_SPEED, F_ACC, F_THR, F_STEER, F_MODE, F_CRASHED, F_TIMED_OUT,
       F_DISP_CTR, F_DISP_X, F_DISP_Y, F_DISP_TO, F_HIT_X, F_HIT_Y, F_REL_X, F_REL_Y, F_DIST, F_COUNT,
       F_REWARD = F_COUNT, F_FITNESS, F_TRACK_IDX, F_EPISODE_STEPS, F_EPISODE_RETURN };

static void *field_ptr(oracle_env *e, int f, size_t *bytes)
{
    const size_t N = e->N, NR = (size_t)e->N * e->R;
    switch (f) {
    case F_POS_X: *bytes = 4 * N; return e->pos_x;
    case F_POS_Y: *bytes = 4 * N; return e->pos_y;
    case F_ROT: *bytes = 4 * N; return e->rot;
    case F_SPEED: *bytes = 4 * N; return e->speed;
    case F_ACC: *bytes = 4 * N; return e->acc;
    case F_THR: *bytes = 4 * N; return e->thr;
    case F_STEER: *bytes = 4 * N; return e->steer;
    case F_MODE: *bytes = N; return e->mode;
    case F_CRASHED: *bytes = N; return e->crashed;
    case F_TIMED_OUT: *bytes = N; return e->timed_out;
    case F_DISP_CTR: *bytes = 4 * N; return e->disp_ctr;
    case F_DISP_X: *bytes = 4 * N; return e->disp_x;
    case F_DISP_Y: *bytes = 4 * N; return e->disp_y;
    case F_DISP_TO: *bytes = N; return e->disp_to;
    case F_HIT_X: *bytes = 4 * NR; return e->hit_x;
    case F_HIT_Y: *bytes = 4 * NR; return e->hit_y;
    case F_REL_X: *bytes = 4 * NR; return e->rel_x;
    case F_REL_Y: *bytes = 4 * NR; return e->rel_y;
    case F_DIST: *bytes = 4 * NR; return e->dist;
    case F_REWARD: *bytes = 4 * N; return e->tr_reward;
    case F_FITNESS: *bytes = 4 * N; return e->tr_fitness;
    case F_TRACK_IDX: *bytes = 4 * N; return e->tr_prev_idx;
    case F_EPISODE_STEPS: *bytes = 4 * N; return e->tr_ep_steps;
    case F_EPISODE_RETURN: *bytes = 4 * N; return e->tr_ep_return;
    default: *bytes = 0; return NULL;
    }
}

ORACLE_API int oracle_env_set_field(oracle_env *e, int f, const void *src)
{
    size_t b; void *p = field_ptr(e, f, &b);
    if (!p) return -1;
    memcpy(p, src, b);
    return 0;
}

ORACLE_API int oracle_env_get_field(oracle_env *e, int f, void *dst)
{
    size_t b; void *p = field_ptr(e, f, &b);
    if (!p) return -1;
    memcpy(dst, p, b);
    return 0;
}

/* Environment/Agent.cpp:123-135 Agent::reset -- note: DisplacementStats are NOT touched (appendix A.5) */
static void agent_reset(oracle_env *e, int i, float x, float y, float rot)
{
    e->pos_x[i] = x; e->pos_y[i] = y; e->rot[i] = rot;
    e->acc[i] = 0.0f; e->speed[i] = 0.0f;
    e->crashed[i] = 0; e->timed_out[i] = 0;
    e->thr[i] = 0.0f; e->steer[i] = 0.0f;
}

ORACLE_API void oracle_env_reset_agents(oracle_env *e, const int32_t *idx, const float *x, const float *y,
                                        const float *rot, int n)
{
    for (int k = 0; k < n; ++k) agent_reset(e, idx[k], x[k], y[k], rot[k]);
}

/* GetRandomValue(lo, hi) of raylib (inclusive integer range; un-vendored, global state) drawn from one 32-bit word of the
 * shared Philox block: lo + floor(word * (hi - lo + 1) / 2^32). */
static int32_t rand_value_from_word(uint32_t word, int32_t lo, int32_t hi)
{
    const uint64_t span = (uint64_t)(hi - lo + 1);
    return lo + (int32_t)(((uint64_t)word * span) >> 32);
}

/* Environment::resetAgent (Environment/Environment.cpp:79-122) for one agent, restated line by line.  The three
 * GetRandomValue calls take words 0, 1, 2 of the Philox block (agent, epoch, 1, 0) -- only the generator is shared with the
 * device code (include/okenv_math.h), the arithmetic below is this file's own reading of the reference.  `ctr` stands for
 * the function-static call counter (:88). */
static void env_reset_agent(oracle_env *e, int i, uint32_t flags, uint32_t seed, uint32_t agent, uint32_t epoch, uint32_t ctr)
{
    const ok_u32x4 words = ok_philox4x32(agent, epoch, 1u, 0u, seed, 0x6F6B656Eu);
    const int pick_random_point = (flags & 1u) != 0u, randomize_lane = (flags & 2u) != 0u, randomize_heading = (flags & 4u) != 0u;
    /* :83  reset_idx = pick_random_point ? pickRandomResetTrackIdx() : RaceTrack::kStartingIdx (= 3, RaceTrack.h:18) */
    const int32_t reset_idx = pick_random_point ? rand_value_from_word(words.v[0], 0, e->P - 1) : 3;
    /* :86-101 */
    float heading_offset = 0.0f;
    if (pick_random_point && randomize_heading) {
        const float kHeadingRandomizationRangeDeg = 45.0f;
        heading_offset = (float)rand_value_from_word(words.v[1], 0, 45);
        if (ctr % 2u == 0u) heading_offset = (heading_offset + kHeadingRandomizationRangeDeg) * -1.0f;
        else heading_offset = heading_offset + kHeadingRandomizationRangeDeg;
    }
    /* :103-120 */
    float start_pos_x, start_pos_y;
    if (pick_random_point && randomize_lane) {
        const float lx = e->lane_l[2 * reset_idx], ly = e->lane_l[2 * reset_idx + 1];
        const float rx = e->lane_r[2 * reset_idx], ry = e->lane_r[2 * reset_idx + 1];
        const float alpha = (float)rand_value_from_word(words.v[2], 10, 90) / 100.0f;
        start_pos_x = lx * alpha + rx * (1.0f - alpha);
        start_pos_y = ly * alpha + ry * (1.0f - alpha);
    } else {
        start_pos_x = e->cx[reset_idx];
        start_pos_y = e->cy[reset_idx];
    }
    /* :121 */
    agent_reset(e, i, start_pos_x, start_pos_y, e->chead[reset_idx] + heading_offset);
}

/* batch form: entry j of the call stands for the (epoch + j)-th resetAgent call of the process */
ORACLE_API void oracle_env_reset_random(oracle_env *e, const int32_t *idx, int n, uint32_t flags, uint32_t seed,
                                        uint32_t epoch, uint32_t agent_base)
{
    if (!idx) n = e->N;
    for (int j = 0; j < n; ++j) {
        const int a = idx ? idx[j] : j;
        if (a < 0 || a >= e->N) continue;
        if ((flags & OK_RESET_ONLY_DONE) && !e->crashed[a]) continue;
        env_reset_agent(e, a, flags, seed, agent_base + (uint32_t)a, epoch, epoch + (uint32_t)j);
    }
}

ORACLE_API void oracle_env_set_auto_reset(oracle_env *e, int enabled, uint32_t flags, uint32_t seed, uint32_t agent_base)
{
    e->auto_reset = enabled;
    e->reset_flags = flags & (OK_RESET_RANDOM_POINT | OK_RESET_RANDOM_LANE | OK_RESET_RANDOM_HEADING);
    e->reset_seed = seed;
    e->reset_agent_base = agent_base;
}

ORACLE_API uint32_t oracle_env_get_step_count(const oracle_env *e) { return e->step_count; }
ORACLE_API void oracle_env_set_step_count(oracle_env *e, uint32_t v) { e->step_count = v; }

/* Environment/Agent.cpp:21-47 dispatch, :108-119 moveViaVelocity, :82-98 moveViaAcceleration.
 * `cos(kDeg2Rad * rot_) * speed_ * kDt` is ((cos * speed) * dt), all fp32. */
static void agent_move(oracle_env *e, int i)
{
    if (e->mode[i] == 0) {
        e->rot[i] += e->steer[i];
        e->speed[i] = e->thr[i];
    } else if (e->mode[i] == 1) {
        e->rot[i] += e->steer[i];
        e->acc[i] += e->thr[i];
        e->speed[i] += (e->acc[i] * OK_DT);
        e->speed[i] = (e->speed[i] < 0.0f) ? 0.0f : e->speed[i];
        e->speed[i] = (e->speed[i] > OK_SPEED_LIMIT) ? OK_SPEED_LIMIT : e->speed[i];
    } else {
        return; /* MANUAL is an empty stub, Environment/Agent.cpp:50-79 */
    }
    float s, c;
    trig(OK_DEG2RAD * e->rot[i], &s, &c);
    const float dx = c * e->speed[i] * OK_DT;
    e->pos_x[i] += dx;
    const float dy = s * e->speed[i] * OK_DT;
    e->pos_y[i] += dy;
}

/* Environment/Environment.cpp:16-39 */
static void standstill(oracle_env *e, int i)
{
    if (e->disp_ctr[i] == 0) {
        e->disp_x[i] = e->pos_x[i]; e->disp_y[i] = e->pos_y[i];
        e->disp_to[i] = 0;
        e->disp_ctr[i]++;
        return;
    }
    if (e->disp_ctr[i] >= OK_DISP_PERIOD) {
        const float ddx = e->pos_x[i] - e->disp_x[i], ddy = e->pos_y[i] - e->disp_y[i];
        const float d2 = ddx * ddx + ddy * ddy; /* Vec2d::distanceSquared */
        if (d2 < OK_DISP_THRESH2) e->disp_to[i] = 1;
        e->disp_ctr[i] = 0;
    } else {
        e->disp_to[i] = 0;
        e->disp_ctr[i]++;
    }
}

/* Environment/CollisionChecker.cu:8-35 */
static int ray_segment(float ox, float oy, float rdx, float rdy, float x1, float y1, float x2, float y2,
                       float range, float *out_t)
{
    const float sdx = x2 - x1;
    const float sdy = y2 - y1;
    const float denom = rdx * sdy - rdy * sdx;
    if (fabsf(denom) < OK_PARALLEL_EPS) return 0;
    const float t = ((x1 - ox) * sdy - (y1 - oy) * sdx) / denom;
    const float s = ((x1 - ox) * rdy - (y1 - oy) * rdx) / denom;
    if ((t >= 0.0f) && (t <= range) && (s >= 0.0f) && (s <= 1.0f)) { *out_t = t; return 1; }
    return 0;
}

/* Environment/CollisionChecker.cu:113-174 for agents [a0,a1): prologue (ray build), kernel, epilogue.
 * tests_out (optional) accumulates the ray-segment test count. */
static void collide_range(oracle_env *e, int a0, int a1)
{
    const int R = e->R, S = e->S;
    for (int i = a0; i < a1; ++i) {
        float sr, cr;
        trig(OK_DEG2RAD * e->rot[i], &sr, &cr);
        /* :121-124 origin = pos + sensor_offset * (cos, sin) */
        const float ox = e->pos_x[i] + e->sensor_offset * cr;
        const float oy = e->pos_y[i] + e->sensor_offset * sr;
        const int active = !e->crashed[i];
        float min_d2 = OK_SENSOR_RANGE * OK_SENSOR_RANGE;
        for (int r = 0; r < R; ++r) {
            const size_t k = (size_t)i * R + r;
            if (active) {
                /* :125 angle, :47-48 direction, :49-67 shrinking-range sweep, :69-70 hit point */
                const float angle = OK_DEG2RAD * (e->rot[i] + e->ray_deg[r]);
                float rdy, rdx;
                trig(angle, &rdy, &rdx);
                float min_t = OK_SENSOR_RANGE;
                const float *sg = e->segs;
                for (int j = 0; j < S; ++j, sg += 4) {
                    float t;
                    if (ray_segment(ox, oy, rdx, rdy, sg[0], sg[1], sg[2], sg[3], min_t, &t)) min_t = t;
                }
                e->hit_x[k] = ox + min_t * rdx;
                e->hit_y[k] = oy + min_t * rdy;
            }
            /* :144-166 epilogue runs for every ray, stale hit points included (appendix A.8) */
            const float xt = e->hit_x[k] - ox;
            const float yt = e->hit_y[k] - oy;
            const float hx = xt * cr - yt * sr;
            const float hy = xt * sr + yt * cr;
            e->rel_x[k] = hx; e->rel_y[k] = hy;
            const float n2 = hx * hx + hy * hy;
            e->dist[k] = sqrtf(n2); /* Vec2d::norm(), what every in-scope policy consumes */
            if (n2 < min_d2) min_d2 = n2;
        }
        if (min_d2 < OK_CRASH_DIST2) e->crashed[i] = 1;
    }
}

ORACLE_API void oracle_env_collide(oracle_env *e) { collide_range(e, 0, e->N); }

/* Environment/Environment.cpp:125-149 minus render(), for agents [a0,a1) */
static void step_range(oracle_env *e, int a0, int a1)
{
    for (int i = a0; i < a1; ++i) {
        if (!e->crashed[i]) {
            agent_move(e, i);
            standstill(e, i);
            if (e->disp_to[i]) { e->crashed[i] = 1; e->timed_out[i] = 1; }
        }
    }
    collide_range(e, a0, a1);
}

/* the caller-side idiom `if (agent->crashed_) env.resetAgent(agent, ...)` before a step
 * (RLRacers/GuidedCostLearning/test.cpp:104-110), applied to every agent when auto-reset is on */
static void auto_reset_pass(oracle_env *e)
{
    if (!e->auto_reset) return;
    for (int i = 0; i < e->N; ++i) {
        if (!e->crashed[i]) continue;
        const uint32_t ag = e->reset_agent_base + (uint32_t)i;
        env_reset_agent(e, i, e->reset_flags, e->reset_seed, ag, e->step_count, ag + e->step_count);
    }
}

/* Environment::step for every agent.  Agents are independent inside a step, so the tests may spread them over host
 * threads (oracle_set_threads; default 1 -- the CPU baseline of bench.py times the single-threaded path). */
#include <pthread.h>
static int g_step_threads = 1;
ORACLE_API void oracle_set_threads(int n) { g_step_threads = n < 1 ? 1 : n; }
typedef struct { oracle_env *e; int a0, a1; } step_job;
static void *step_job_main(void *p)
{
    step_job *j = (step_job *)p;
    step_range(j->e, j->a0, j->a1);
    return NULL;
}
static void step_all(oracle_env *e)
{
    int threads = g_step_threads > e->N ? e->N : g_step_threads;
    if (threads <= 1 || e->N * e->R < 512) { step_range(e, 0, e->N); return; }
    pthread_t th[64];
    step_job jobs[64];
    if (threads > 64) threads = 64;
    for (int t = 0; t < threads; ++t) {
        jobs[t].e = e;
        jobs[t].a0 = (int)((long long)e->N * t / threads);
        jobs[t].a1 = (int)((long long)e->N * (t + 1) / threads);
        pthread_create(&th[t], NULL, step_job_main, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
}

ORACLE_API void oracle_env_step(oracle_env *e, int n_steps)
{
    for (int s = 0; s < n_steps; ++s) {
        auto_reset_pass(e);
        step_all(e);
        e->step_count++;
    }
}

/* kinematics + standstill only (validated against the reference's Agent.o) */
ORACLE_API void oracle_env_move_only(oracle_env *e)
{
    for (int i = 0; i < e->N; ++i) {
        if (!e->crashed[i]) {
            agent_move(e, i);
            standstill(e, i);
            if (e->disp_to[i]) { e->crashed[i] = 1; e->timed_out[i] = 1; }
        }
    }
}

/*
 * The bench driver loop (shape of RLRacers/GuidedCostLearning/test.cpp:100-117):
 * per step: crashed agents are re-placed on a Philox-chosen centre-line point (Agent::reset), every
 * agent draws a fresh action, then Environment::step.  `agent_base` is the global id of agent 0 (for
 * sharded populations), `step_base` the global index of the first step.
 */
static void rollout_range(oracle_env *e, int a0, int a1, int n_steps, uint32_t seed, uint32_t agent_base,
                          uint32_t step_base)
{
    for (int s = 0; s < n_steps; ++s) {
        for (int i = a0; i < a1; ++i) {
            const ok_random_action a = ok_draw_random_action(seed, agent_base + (uint32_t)i, step_base + (uint32_t)s);
            if (e->crashed[i]) {
                const uint32_t idx = ok_index_from_word(a.reset_word, (uint32_t)e->P);
                agent_reset(e, i, e->cx[idx], e->cy[idx], e->chead[idx]);
            }
            e->thr[i] = a.throttle;
            e->steer[i] = a.steer;
        }
        step_range(e, a0, a1);
    }
}

ORACLE_API void oracle_env_rollout_random(oracle_env *e, int n_steps, uint32_t seed, uint32_t agent_base,
                                          uint32_t step_base)
{
    rollout_range(e, 0, e->N, n_steps, seed, agent_base, step_base);
}

/* bench recipe initial state: agent j on centre-line index ok_start_index(j), track heading, mode */
ORACLE_API void oracle_env_init_bench_state(oracle_env *e, uint32_t agent_base, int mode)
{
    for (int i = 0; i < e->N; ++i) {
        const uint32_t idx = ok_start_index(agent_base + (uint32_t)i, (uint32_t)e->P);
        agent_reset(e, i, e->cx[idx], e->cy[idx], e->chead[idx]);
        e->mode[i] = (uint8_t)mode;
        e->disp_ctr[i] = 0; e->disp_x[i] = 0.0f; e->disp_y[i] = 0.0f; e->disp_to[i] = 0;
    }
    memset(e->hit_x, 0, sizeof(float) * (size_t)e->N * e->R);
    memset(e->hit_y, 0, sizeof(float) * (size_t)e->N * e->R);
}

/*
 * Multi-threaded variant for the separately-labelled "all host cores" CPU row:
 * agents are partitioned into contiguous blocks, no shared writes.  Plain pthreads.
 */
#include <pthread.h>
typedef struct { oracle_env *e; int a0, a1, n_steps; uint32_t seed, agent_base, step_base; } mt_job;
static void *mt_main(void *p)
{
    mt_job *j = (mt_job *)p;
    rollout_range(j->e, j->a0, j->a1, j->n_steps, j->seed, j->agent_base, j->step_base);
    return NULL;
}

ORACLE_API void oracle_env_rollout_random_mt(oracle_env *e, int n_steps, uint32_t seed, uint32_t agent_base,
                                             uint32_t step_base, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > e->N) threads = e->N;
    pthread_t *th = malloc(sizeof(pthread_t) * threads);
    mt_job *jobs = malloc(sizeof(mt_job) * threads);
    for (int t = 0; t < threads; ++t) {
        jobs[t].e = e;
        jobs[t].a0 = (int)((long long)e->N * t / threads);
        jobs[t].a1 = (int)((long long)e->N * (t + 1) / threads);
        jobs[t].n_steps = n_steps; jobs[t].seed = seed; jobs[t].agent_base = agent_base; jobs[t].step_base = step_base;
        pthread_create(&th[t], NULL, mt_main, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

/* stand-alone raycast for known-answer tests: one ray against S segments, returns min_t */
ORACLE_API float oracle_cast_ray(float ox, float oy, float angle_rad, const float *segs, int S)
{
    float rdy, rdx;
    trig(angle_rad, &rdy, &rdx);
    float min_t = OK_SENSOR_RANGE;
    for (int j = 0; j < S; ++j) {
        float t;
        if (ray_segment(ox, oy, rdx, rdy, segs[4 * j], segs[4 * j + 1], segs[4 * j + 2], segs[4 * j + 3], min_t, &t))
            min_t = t;
    }
    return min_t;
}

/* ============================================================================================ */
/* EvolutionaryRacer                                        */
/* ============================================================================================ */

/* genetic::Network() (EvolutionaryRacer/Network.hpp:99-107), weights from the Philox stand-in of okenv_math.h */
ORACLE_API void oracle_ga_create(oracle_env *e, int hidden, uint32_t seed, uint32_t agent_base)
{
    const int per = OK_MLP_WEIGHTS(e->R);
    free(e->mlp_w); free(e->score);
    e->mlp_hidden = hidden;
    e->mlp_w = ALLOC(float, (size_t)e->N * per);
    e->score = ALLOC(float, e->N);
    for (int a = 0; a < e->N; ++a)
        for (int i = 0; i < per; ++i)
            e->mlp_w[(size_t)a * per + i] =
                ok_mlp_weight_is_real((uint32_t)i, e->R, hidden) ? ok_ga_initial_weight(seed, agent_base + (uint32_t)a, (uint32_t)i) : 0.0f;
}

ORACLE_API int oracle_ga_weights_per_agent(const oracle_env *e) { return OK_MLP_WEIGHTS(e->R); }
ORACLE_API void oracle_ga_get_weights(const oracle_env *e, float *out)
{
    memcpy(out, e->mlp_w, sizeof(float) * (size_t)e->N * OK_MLP_WEIGHTS(e->R));
}
ORACLE_API void oracle_ga_set_weights(oracle_env *e, const float *in)
{
    memcpy(e->mlp_w, in, sizeof(float) * (size_t)e->N * OK_MLP_WEIGHTS(e->R));
}

/* genetic::normalizeAngleDeg (EvolutionaryRacer/Network.hpp:16-27).  (An infinite or NaN angle makes the reference spin
 * forever; the bound keeps the oracle from hanging and is never reached by finite headings below 2.4e7 degrees.) */
static float ga_normalize_angle_deg(float angle)
{
    int guard = 65536;
    while (angle < 360.0f && guard-- > 0) angle += 360.0f;
    guard = 65536;
    while (angle >= 360.0f && guard-- > 0) angle -= 360.0f;
    return angle;
}

/* Network::sigmoid (EvolutionaryRacer/Network.hpp:162-165): 1.F / (1.F + exp(-x)) in fp32, written out with glibc's expf
 * (the reference's is Eigen's array exp, version unpinned). */
static float ga_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

/* GeneticAgent::updateAction's Sigmoid branch (EvolutionaryRacer/GeneticAgent.hpp:45-54): an output is active when
 * nn_output_[k] > kOutputActivationLim (0.5).  kAccelerationDelta 0.3, kSteeringDeltaLow 1, kSteeringDeltaHigh 4
 * (GeneticAgent.hpp:19-24). */
static void ga_decode_outputs(const float z[6], float *throttle_delta_out, float *steering_delta_out)
{
    float nn_output[6];
    for (int k = 0; k < 6; ++k) nn_output[k] = ga_sigmoid(z[k]);
    float throttle_delta = 0.0f, steering_delta = 0.0f;
    throttle_delta += (nn_output[0] > 0.5f) ? 0.3f : 0.0f;
    throttle_delta += (nn_output[1] > 0.5f) ? -0.3f : 0.0f;
    steering_delta += (nn_output[2] > 0.5f) ? 1.0f : 0.0f;  /* left soft  */
    steering_delta += (nn_output[3] > 0.5f) ? 4.0f : 0.0f;  /* left hard  */
    steering_delta += (nn_output[4] > 0.5f) ? -1.0f : 0.0f; /* right soft */
    steering_delta += (nn_output[5] > 0.5f) ? -4.0f : 0.0f; /* right hard */
    *throttle_delta_out = throttle_delta;
    *steering_delta_out = steering_delta;
}

/* For tests/test_math.py: the decode as the oracle does it (sigmoid written out) and as the product's header does it (one
 * comparison, include/okenv_math.h), on n sextuples of pre-activations. */
ORACLE_API void oracle_ga_decode(const float *z, int n, float *thr_oracle, float *steer_oracle, float *thr_product, float *steer_product)
{
    for (int i = 0; i < n; ++i) {
        ga_decode_outputs(z + 6 * (size_t)i, thr_oracle + i, steer_oracle + i);
        ok_ga_decode_action(z + 6 * (size_t)i, thr_product + i, steer_product + i);
    }
}

/* std::discrete_distribution over the parents' scores (EvolutionaryRacer/Mating.hpp:135-137) driven by one uniform draw:
 * weights are the (non-negative) scores; all-zero weights make the distribution uniform, as libstdc++ does. */
static int ga_draw_parent(const float *scores, int K, float u)
{
    float total = 0.0f;
    for (int k = 0; k < K; ++k) total += (scores[k] > 0.0f) ? scores[k] : 0.0f;
    if (!(total > 0.0f)) {
        const int k = (int)(u * (float)K);
        return k < K ? k : K - 1;
    }
    const float target = u * total;
    float running = 0.0f;
    for (int k = 0; k < K; ++k) {
        running += (scores[k] > 0.0f) ? scores[k] : 0.0f;
        if (target < running) return k;
    }
    return K - 1;
}

/* chooseAndMateAgents' parent choice (EvolutionaryRacer/Mating.hpp:128-152): network 0 is the best agent's clone,
 * network 1 the best mated with itself; every further one draws first_agent, then second_agent until it differs
 * ("Prevent self-mutation").  The mt19937 seeded from random_device is replaced by Philox draws (offspring, attempt, 3,
 * generation); after 16 equal draws the next parent is taken so that the loop ends.  *clone_out = exact copy. */
static void ga_choose_parents(const float *scores, int K, uint32_t seed, uint32_t offspring, uint32_t generation, int *first_out,
                              int *second_out, int *clone_out)
{
    *clone_out = 0;
    if (offspring == 0u) { *first_out = 0; *second_out = 0; *clone_out = 1; return; }
    if (offspring == 1u || K < 2) { *first_out = 0; *second_out = 0; return; }
    const ok_u32x4 r0 = ok_philox4x32(offspring, 0u, 3u, generation, seed, 0x6F6B656Eu);
    const int first_agent = ga_draw_parent(scores, K, ok_u01(r0.v[0]));
    int second_agent = -1;
    for (uint32_t attempt = 1u; attempt <= 16u && (second_agent == -1 || second_agent == first_agent); ++attempt) {
        const ok_u32x4 r = ok_philox4x32(offspring, attempt, 3u, generation, seed, 0x6F6B656Eu);
        second_agent = ga_draw_parent(scores, K, ok_u01(r.v[0]));
    }
    if (second_agent == first_agent) second_agent = (first_agent + 1) % K;
    *first_out = first_agent;
    *second_out = second_agent;
}

/* GeneticAgent::updateAction (GeneticAgent.hpp:37-107) + Network::infer (Network.hpp:119-155).  Accumulation orders:
 * hidden unit u over inputs j = 0..R+1, output k over hidden units i = 0..31 (zero-padded beyond the hidden width). */
static void ga_update_action(oracle_env *e, int a)
{
    const int R = e->R;
    const float *w1 = e->mlp_w + (size_t)a * OK_MLP_WEIGHTS(R);
    const float *w2 = w1 + (R + 2) * OK_MLP_HID_PAD;
    float h[OK_MLP_HID_PAD], z[OK_MLP_OUT];
    const float x0 = e->speed[a] / 100.0f;
    const float x1 = ga_normalize_angle_deg(e->rot[a]) / 360.0f;
    for (int u = 0; u < OK_MLP_HID_PAD; ++u) {
        float acc = 0.0f;
        acc = acc + x0 * w1[0 * OK_MLP_HID_PAD + u];
        acc = acc + x1 * w1[1 * OK_MLP_HID_PAD + u];
        for (int j = 0; j < R; ++j) {
            const float xj = e->dist[(size_t)a * R + j] / 200.0f;
            acc = acc + xj * w1[(2 + j) * OK_MLP_HID_PAD + u];
        }
        h[u] = (acc > 0.0f) ? acc : 0.0f;
    }
    for (int k = 0; k < OK_MLP_OUT; ++k) {
        float acc = 0.0f;
        for (int i = 0; i < OK_MLP_HID_PAD; ++i) acc = acc + h[i] * w2[i * OK_MLP_OUT_PAD + k];
        z[k] = acc;
    }
    ga_decode_outputs(z, &e->thr[a], &e->steer[a]);
}

/* genetic_learner_sim.cpp:76-95: n x { updateAction for all; env.step() } */
ORACLE_API void oracle_env_rollout_policy(oracle_env *e, int n_steps)
{
    for (int s = 0; s < n_steps; ++s) {
        for (int a = 0; a < e->N; ++a) ga_update_action(e, a);
        auto_reset_pass(e); /* off in the reference's loop; when on, a reset agent's step runs with the zeroed action */
        step_all(e);
        e->step_count++;
    }
}

/* CmaEsAgent::updateAction for every agent (CovarianceMatrixAdaptationEvolution/main_eigen.cpp:45-68): the input is
 * sensor_hits_[i].norm() / kSensorRange (:47-50), the controller tanh(fc3(tanh(fc2(tanh(fc1(x)))))) (Controller.cpp:16-23)
 * with fc1: R -> hidden, fc2: hidden -> hidden / 2, fc3: hidden / 2 -> 2 (:3-8, main_eigen.cpp:18-19); throttle_delta is a
 * constant and steering_delta = output[0] * scale (:65-67; 100 and 5 in the reference).  `params`: N rows in the order of
 * torch's parameters().  Sums run from the bias on, inputs ascending (see include/okenv_math.h on why that is a choice). */
static void ctrl_forward(const float *prm, const float *x, int in, int hidden, float *out2)
{
    const int h2 = hidden / 2;
    float a1[OK_CTRL_MAX_HIDDEN], a2[OK_CTRL_MAX_HIDDEN];
    const float *w1 = prm, *b1 = w1 + hidden * in, *w2 = b1 + hidden, *b2 = w2 + h2 * hidden, *w3 = b2 + h2, *b3 = w3 + 2 * h2;
    for (int j = 0; j < hidden; ++j) {
        float sum = b1[j];
        for (int i = 0; i < in; ++i) sum = sum + w1[j * in + i] * x[i];
        a1[j] = ok_tanhf(sum);
    }
    for (int j = 0; j < h2; ++j) {
        float sum = b2[j];
        for (int i = 0; i < hidden; ++i) sum = sum + w2[j * hidden + i] * a1[i];
        a2[j] = ok_tanhf(sum);
    }
    for (int j = 0; j < 2; ++j) {
        float sum = b3[j];
        for (int i = 0; i < h2; ++i) sum = sum + w3[j * h2 + i] * a2[i];
        out2[j] = ok_tanhf(sum);
    }
}

ORACLE_API int oracle_env_controller_act(oracle_env *e, const float *params, int hidden, float throttle, float steering_scale)
{
    if (hidden < 2 || hidden > OK_CTRL_MAX_HIDDEN || (hidden & 1) || e->R > 64) return -1;
    const int np = hidden * e->R + hidden + (hidden / 2) * hidden + hidden / 2 + 2 * (hidden / 2) + 2;
    for (int a = 0; a < e->N; ++a) {
        float x[64], out2[2];
        for (int i = 0; i < e->R; ++i) x[i] = e->dist[(size_t)a * e->R + i] / 200.0f;
        ctrl_forward(params + (size_t)a * np, x, e->R, hidden, out2);
        e->thr[a]   = throttle;
        e->steer[a] = out2[0] * steering_scale;
    }
    return 0;
}

ORACLE_API int oracle_env_alive_count(const oracle_env *e)
{
    int n = 0;
    for (int a = 0; a < e->N; ++a) n += e->crashed[a] ? 0 : 1;
    return n;
}

ORACLE_API void oracle_env_reset_all(oracle_env *e, float x, float y, float rot)
{
    for (int a = 0; a < e->N; ++a) agent_reset(e, a, x, y, rot);
}

/* assignScores (MiscUtils.hpp:64-71) */
ORACLE_API void oracle_ga_scores(oracle_env *e, float *out)
{
    for (int a = 0; a < e->N; ++a) {
        float best = FLT_MAX; int bi = 0;
        for (int i = 0; i < e->P; ++i) {
            const float dx = e->pos_x[a] - e->cx[i], dy = e->pos_y[a] - e->cy[i];
            const float d = dx * dx + dy * dy;
            if (d < best) { best = d; bi = i; }
        }
        e->score[a] = (float)bi;
        if (out) out[a] = e->score[a];
    }
}

/* chooseAndMateAgents (Mating.hpp:108-166) + mate2AgentsSelective (:52-99) with the Philox stand-ins */
ORACLE_API void oracle_ga_select_mate(oracle_env *e, uint32_t seed, uint32_t generation, uint32_t agent_base, int32_t *parents_out)
{
    const int N = e->N, per = OK_MLP_WEIGHTS(e->R), K = N < 5 ? N : 5;
    int parents[5]; float ps[5];
    for (int k = 0; k < K; ++k) { /* descending score, ties to the lower index */
        float best = -HUGE_VALF; int arg = -1;
        for (int a = 0; a < N; ++a) {
            int taken = 0;
            for (int q = 0; q < k; ++q) taken |= (parents[q] == a);
            if (!taken && (arg < 0 || e->score[a] > best)) { best = e->score[a]; arg = a; }
        }
        parents[k] = arg; ps[k] = best;
        if (parents_out) parents_out[k] = arg;
    }
    float *nw = ALLOC(float, (size_t)N * per);
    for (int o = 0; o < N; ++o) {
        const uint32_t og = agent_base + (uint32_t)o;
        int first_i, second_i, clone;
        ga_choose_parents(ps, K, seed, og, generation, &first_i, &second_i, &clone);
        const uint32_t first = (uint32_t)first_i, second = (uint32_t)second_i;
        const uint32_t dom = (ps[first] > ps[second]) ? first : second; /* n1 = agent_2 on ties (Mating.hpp:59) */
        const uint32_t sub = (ps[first] > ps[second]) ? second : first;
        const float *wd = e->mlp_w + (size_t)parents[dom] * per;
        const float *ws = e->mlp_w + (size_t)parents[sub] * per;
        for (int i = 0; i < per; ++i) {
            float out = wd[i];
            const int real = ok_mlp_weight_is_real((uint32_t)i, e->R, e->mlp_hidden);
            if (real && !clone) {
                const ok_u32x4 r = ok_philox4x32(og, (uint32_t)i, 2u, generation, seed, 0x6F6B656Eu);
                if (ok_u01(r.v[0]) < 0.1f) out = (ok_u01(r.v[1]) - 0.5f) * 2.0f;
                else out = (ok_u01(r.v[2]) < 0.75f) ? wd[i] : ws[i];
            }
            nw[(size_t)o * per + i] = real ? out : 0.0f;
        }
    }
    free(e->mlp_w);
    e->mlp_w = nw;
}

/* ============================================================================================ */
/* RLRacers/Q_Learning                                            */
/* ============================================================================================ */

#define Q_STATES 243
#define Q_ACTIONS 3
static const float kQInvalid = -FLT_MAX; /* numeric_limits<float>::lowest(), QAgent.hpp:36 */

ORACLE_API void oracle_q_create(oracle_env *e)
{
    free(e->q_table); free(e->q_state); free(e->q_action); free(e->q_prev);
    const size_t n = (size_t)e->N * Q_STATES * Q_ACTIONS;
    e->q_table = ALLOC(float, n);
    for (size_t i = 0; i < n; ++i) e->q_table[i] = kQInvalid; /* QAgent.hpp:64-68 */
    e->q_state = ALLOC(int32_t, e->N); e->q_action = ALLOC(int32_t, e->N); e->q_prev = ALLOC(int32_t, e->N);
    const float target[5] = {-70.0f, -30.0f, 0.0f, 30.0f, 70.0f}; /* the reference's five rays, QAgent.hpp:56-62 */
    for (int t = 0; t < 5; ++t) {
        int arg = 0; float best = fabsf(e->ray_deg[0] - target[t]);
        for (int r = 1; r < e->R; ++r) {
            const float d = fabsf(e->ray_deg[r] - target[t]);
            if (d < best) { best = d; arg = r; }
        }
        e->q_ray[t] = arg;
    }
}

/* QLearnAgent::discretizeState (QAgent.hpp:72-94) over the five state rays */
static int q_discretize(const oracle_env *e, int a)
{
    int state = 0, mult = 1;
    for (int i = 0; i < 5; ++i) {
        const float d = e->dist[(size_t)a * e->R + e->q_ray[i]];
        int bin;
        if (d < 5.0f) bin = 0;
        else if (d < 10.0f) bin = 1;
        else bin = 2;
        state += bin * mult;
        mult *= 3;
    }
    return state;
}

static int nearest_index(const oracle_env *e, float x, float y)
{
    float best = FLT_MAX; int bi = 0;
    for (int i = 0; i < e->P; ++i) {
        const float dx = x - e->cx[i], dy = y - e->cy[i];
        const float d = dx * dx + dy * dy;
        if (d < best) { best = d; bi = i; }
    }
    return bi;
}

/* q_racer_sim.cpp:129-154 */
ORACLE_API void oracle_q_begin_episode(oracle_env *e, int reset_idx)
{
    const float x = e->cx[reset_idx], y = e->cy[reset_idx];
    const int near = nearest_index(e, x, y);
    for (int a = 0; a < e->N; ++a) { agent_reset(e, a, x, y, e->chead[reset_idx]); e->q_prev[a] = near; }
    step_all(e);
    for (int a = 0; a < e->N; ++a) e->q_state[a] = q_discretize(e, a);
}

/* q_racer_sim.cpp:158-182 with QAgent.hpp:98-119 (updateAction), :150-168 (reward), :121-138 (learn) */
ORACLE_API void oracle_rollout_q(oracle_env *e, int n_steps, float epsilon, uint32_t seed, uint32_t agent_base, uint32_t step_base)
{
    for (int s = 0; s < n_steps; ++s) {
        for (int a = 0; a < e->N; ++a) {
            float *row = e->q_table + ((size_t)a * Q_STATES + e->q_state[a]) * Q_ACTIONS;
            const ok_u32x4 r = ok_philox4x32(agent_base + (uint32_t)a, step_base + (uint32_t)s, 4u, 0u, seed, 0x6F6B656Eu);
            int act;
            if (ok_u01(r.v[0]) < epsilon) {
                act = (int)(ok_u01(r.v[1]) * 3.0f);
                if (act > 2) act = 2;
            } else { /* std::max_element: first maximum */
                act = 0;
                if (row[1] > row[act]) act = 1;
                if (row[2] > row[act]) act = 2;
            }
            e->q_action[a] = act;
            /* kActionMap, QAgent.hpp:40-42 */
            e->thr[a] = (act == 0) ? 60.0f : 30.0f;
            e->steer[a] = (act == 0) ? 0.0f : (act == 1 ? 5.0f : -5.0f);
        }
        step_all(e);
        for (int a = 0; a < e->N; ++a) {
            const int next_state = q_discretize(e, a);
            float reward;
            if (e->crashed[a]) {
                reward = -200.0f;
            } else {
                const int near = nearest_index(e, e->pos_x[a], e->pos_y[a]);
                long prog = (long)near - (long)e->q_prev[a];
                e->q_prev[a] = near;
                if (prog < 0) prog = -prog;
                reward = (float)((prog > (long)(e->P / 2)) ? (long)e->P - prog : prog);
            }
            const float *nrow = e->q_table + ((size_t)a * Q_STATES + next_state) * Q_ACTIONS;
            float maxq = nrow[0];
            if (nrow[1] > maxq) maxq = nrow[1];
            if (nrow[2] > maxq) maxq = nrow[2];
            float *cell = e->q_table + ((size_t)a * Q_STATES + e->q_state[a]) * Q_ACTIONS + e->q_action[a];
            const float target = reward + 0.8f * maxq;
            const float old_q = *cell;
            if (old_q == kQInvalid || maxq == kQInvalid) *cell = reward;
            else *cell = old_q + 0.2f * (target - old_q);
            if (!e->crashed[a]) e->q_state[a] = next_state;
        }
    }
}

ORACLE_API void oracle_q_get_table(const oracle_env *e, float *out)
{
    memcpy(out, e->q_table, sizeof(float) * (size_t)e->N * Q_STATES * Q_ACTIONS);
}
ORACLE_API void oracle_q_set_table(oracle_env *e, const float *in)
{
    memcpy(e->q_table, in, sizeof(float) * (size_t)e->N * Q_STATES * Q_ACTIONS);
}
ORACLE_API void oracle_q_get_state(const oracle_env *e, int32_t *state, int32_t *action, int32_t *prev)
{
    memcpy(state, e->q_state, 4 * (size_t)e->N); memcpy(action, e->q_action, 4 * (size_t)e->N); memcpy(prev, e->q_prev, 4 * (size_t)e->N);
}

/* ============================================================================================ */
/* Rollout bookkeeping of the living population callers            */
/* ============================================================================================ */

/* reward kinds, as in include/okenv.h: 0 = +1 per step (RLRacers/PPO/ppo_sim.cpp:77-80),
 * 1 = index progress (CovarianceMatrixAdaptationEvolution/main_eigen.cpp:147-158) */
ORACLE_API void oracle_tracker_create(oracle_env *e, int kind)
{
    if (!e->tr_fitness) {
        e->tr_prev_idx = ALLOC(int32_t, e->N); e->tr_fitness = ALLOC(float, e->N); e->tr_reward = ALLOC(float, e->N);
        e->tr_ep_return = ALLOC(float, e->N); e->tr_ep_steps = ALLOC(uint32_t, e->N);
        e->tr_prev_crashed = ALLOC(uint8_t, e->N);
    }
    e->tracker_kind = kind;
}

/* main_eigen.cpp:128-133 (prev_track_idx_ after the initial-observation step) and :70-74 (fitness_ = 0 in reset) */
ORACLE_API void oracle_tracker_begin(oracle_env *e)
{
    for (int a = 0; a < e->N; ++a) {
        /* the +1 reward never reads the index (ppo_sim.cpp:82-86 is commented out): kept at 0 there */
        e->tr_prev_idx[a] = e->tracker_kind == 1 ? nearest_index(e, e->pos_x[a], e->pos_y[a]) : 0;
        e->tr_fitness[a] = 0.0f; e->tr_reward[a] = 0.0f; e->tr_ep_steps[a] = 0;
        e->tr_prev_crashed[a] = e->crashed[a];
    }
}

/* the loop body after env.step(): main_eigen.cpp:143-158 / ppo_sim.cpp:73-88; an agent that was crashed at the last
 * update and is not now was re-placed in between, so its episode starts over (reward 0 for its observation step) */
ORACLE_API void oracle_tracker_update(oracle_env *e)
{
    for (int a = 0; a < e->N; ++a) {
        const int crashed = e->crashed[a], was = e->tr_prev_crashed[a];
        float reward = 0.0f;
        if (was && !crashed) {
            e->tr_fitness[a] = 0.0f; e->tr_ep_steps[a] = 0;
            e->tr_prev_idx[a] = e->tracker_kind == 1 ? nearest_index(e, e->pos_x[a], e->pos_y[a]) : 0;
        } else if (e->tracker_kind == 0) {
            reward = 1.0f;
            e->tr_fitness[a] += 1.0f;
            e->tr_ep_steps[a]++;
        } else if (!crashed) {
            const int32_t curr = nearest_index(e, e->pos_x[a], e->pos_y[a]);
            const int32_t progress = curr - e->tr_prev_idx[a];
            e->tr_prev_idx[a] = curr;
            reward = (float)abs(progress);
            e->tr_fitness[a] += reward;
            e->tr_ep_steps[a]++;
        } else if (e->timed_out[a]) {
            e->tr_fitness[a] = 0.0f;
        }
        e->tr_reward[a] = reward;
        if (crashed && !was) e->tr_ep_return[a] = e->tr_fitness[a];
        e->tr_prev_crashed[a] = (uint8_t)crashed;
    }
}
