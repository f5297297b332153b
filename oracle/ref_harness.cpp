// ref_harness.cpp -- C-ABI shim around the REFERENCE's own compilable translation units.
// TEST INFRASTRUCTURE; builds only in the development container, where /root/reference exists.
//
// oracle/Makefile compiles Environment/Agent.cpp and Environment/RaceTrack.cpp from where they lie
// under /root/reference (nothing is copied into this repository) together with this file into
// oracle/_ref/libokref.so.  tests/test_oracle_vs_ref.py uses it to pin oracle/okenv_oracle.c
// bit-for-bit (track geometry, headings, nearest index, kinematics, Agent::reset), and
// tests/golden/make_golden.py uses it to generate the committed fixtures.
//
// The rest of the reference's path (Environment.cpp, CollisionChecker.cu, TrackSegments.cu) needs
// raylib / nvcc / the CUDA runtime and is unbuildable here.
#include <cstdint>
#include <cstring>
#include <vector>

#include "Agent.h"     // /root/reference/Environment/Agent.h
#include "RaceTrack.h" // /root/reference/Environment/RaceTrack.h

namespace
{
class PlainAgent : public Agent
{
  public:
    PlainAgent(Vec2d p, float rot) : Agent(p, rot, 0) {}
    void updateAction() override {}
};

void copyVec2(const std::vector<Vec2d> &v, float *out)
{
    for (size_t i = 0; i < v.size(); ++i)
    {
        out[2 * i]     = v[i].x;
        out[2 * i + 1] = v[i].y;
    }
}
} // namespace

extern "C"
{
    __attribute__((visibility("default"))) void *ref_track_load(const char *csv_path)
    {
        return new RaceTrack(std::string(csv_path));
    }

    __attribute__((visibility("default"))) void ref_track_free(void *t)
    {
        delete static_cast<RaceTrack *>(t);
    }

    __attribute__((visibility("default"))) int ref_track_num_points(void *t)
    {
        return static_cast<int>(static_cast<RaceTrack *>(t)->track_data_points_.x_m.size());
    }

    // which: 0 x, 1 y, 2 w_right, 3 w_left, 4 heading (P floats); 5 li, 6 lo, 7 ri, 8 ro (2P floats)
    __attribute__((visibility("default"))) int ref_track_get(void *tp, int which, float *out)
    {
        auto *t = static_cast<RaceTrack *>(tp);
        const auto &d = t->track_data_points_;
        switch (which)
        {
        case 0: std::memcpy(out, d.x_m.data(), d.x_m.size() * 4); return 0;
        case 1: std::memcpy(out, d.y_m.data(), d.y_m.size() * 4); return 0;
        case 2: std::memcpy(out, d.w_tr_right_m.data(), d.w_tr_right_m.size() * 4); return 0;
        case 3: std::memcpy(out, d.w_tr_left_m.data(), d.w_tr_left_m.size() * 4); return 0;
        case 4: std::memcpy(out, t->headings_.data(), t->headings_.size() * 4); return 0;
        case 5: copyVec2(t->left_bound_inner_, out); return 0;
        case 6: copyVec2(t->left_bound_outer_, out); return 0;
        case 7: copyVec2(t->right_bound_inner_, out); return 0;
        case 8: copyVec2(t->right_bound_outer_, out); return 0;
        default: return -1;
        }
    }

    __attribute__((visibility("default"))) void
    ref_nearest_track_idx(void *tp, const float *qx, const float *qy, int n, int32_t *out)
    {
        auto *t = static_cast<RaceTrack *>(tp);
        for (int i = 0; i < n; ++i)
        {
            out[i] = static_cast<int32_t>(t->findNearestTrackIndexBruteForce(Vec2d{qx[i], qy[i]}));
        }
    }

    // RaceTrack::getNearestDistanceToTrackBoundary / getDistanceToLaneCenter for n probe points
    __attribute__((visibility("default"))) void
    ref_track_queries(void *tp, const float *qx, const float *qy, int n, float *out_boundary, float *out_lane_center)
    {
        auto *t = static_cast<RaceTrack *>(tp);
        for (int i = 0; i < n; ++i)
        {
            out_boundary[i]    = t->getNearestDistanceToTrackBoundary(Vec2d{qx[i], qy[i]});
            out_lane_center[i] = t->getDistanceToLaneCenter(Vec2d{qx[i], qy[i]});
        }
    }

    // Drives the reference's Agent::move() n_steps times with the given per-step actions.
    // out arrays hold the state AFTER each step: x, y, rot, speed, acceleration.
    __attribute__((visibility("default"))) void ref_agent_rollout(int         mode,
                                                                  float       x0,
                                                                  float       y0,
                                                                  float       rot0,
                                                                  const float *thr,
                                                                  const float *steer,
                                                                  int         n_steps,
                                                                  float      *out_x,
                                                                  float      *out_y,
                                                                  float      *out_rot,
                                                                  float      *out_speed,
                                                                  float      *out_acc)
    {
        PlainAgent a(Vec2d{x0, y0}, rot0);
        a.setMovementMode(static_cast<Agent::MovementMode>(mode));
        for (int s = 0; s < n_steps; ++s)
        {
            a.current_action_.throttle_delta = thr[s];
            a.current_action_.steering_delta = steer[s];
            a.move();
            out_x[s]     = a.pos_.x;
            out_y[s]     = a.pos_.y;
            out_rot[s]   = a.rot_;
            out_speed[s] = a.speed_;
            out_acc[s]   = a.acceleration_;
        }
    }

    // Agent::reset semantics: returns the 9 fields after dirtying then resetting an agent.
    // out = {x, y, rot, speed, acc, crashed, timed_out, throttle, steer}
    __attribute__((visibility("default"))) void ref_agent_reset_probe(float x, float y, float rot, float *out)
    {
        PlainAgent a(Vec2d{1.F, 2.F}, 3.F);
        a.speed_         = 5.F;
        a.acceleration_  = 6.F;
        a.crashed_       = true;
        a.timed_out_     = true;
        a.current_action_ = {7.F, 8.F};
        a.reset(Vec2d{x, y}, rot);
        out[0] = a.pos_.x;
        out[1] = a.pos_.y;
        out[2] = a.rot_;
        out[3] = a.speed_;
        out[4] = a.acceleration_;
        out[5] = a.crashed_ ? 1.F : 0.F;
        out[6] = a.timed_out_ ? 1.F : 0.F;
        out[7] = a.current_action_.throttle_delta;
        out[8] = a.current_action_.steering_delta;
    }

    // Default sensor fan built by the reference's Agent constructor (Agent.cpp:8-19).
    __attribute__((visibility("default"))) int ref_agent_default_rays(float *out, int cap)
    {
        PlainAgent a(Vec2d{0.F, 0.F}, 0.F);
        const int  n = static_cast<int>(a.sensor_ray_angles_.size());
        for (int i = 0; i < n && i < cap; ++i)
        {
            out[i] = a.sensor_ray_angles_[i];
        }
        return n;
    }

    // Layout facts the SoA / Ray_ design relies on.
    __attribute__((visibility("default"))) void ref_layout_facts(uint32_t *out)
    {
        out[0] = sizeof(Ray_);
        out[1] = sizeof(Segment2d);
        float k = kDeg2Rad;
        std::memcpy(&out[2], &k, 4);
        out[3] = static_cast<uint32_t>(RaceTrack::kStartingIdx);
    }
}
