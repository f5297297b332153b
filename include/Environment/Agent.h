// Agent.h -- one simulated car: pose, kinematic state, lidar fan, flags and the current action.
//
// Source-compatible with the reference's Agent (reference Environment/Agent.h:7-94): callers derive from it,
// implement updateAction(), write current_action_ and read sensor_hits_ / crashed_ / pos_ / rot_ directly,
// so every public member keeps its name, type and default.  In this project the Agent object is a HOST-side
// mirror: Environment::step() gathers these fields into the device-resident struct-of-arrays state, runs
// the fused HIP step and scatters the results back (include/okenv.h).
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "Typedefs.h"

class OKENV_CLASS Agent
{
  public:
    static constexpr float kSensorRange{200.F};
    static constexpr float kSpeedLimit{100.F};
    static constexpr float kRotationLimit{360.F};

    struct Action
    {
        float throttle_delta{0.F}; // VELOCITY: the speed itself; ACCELERATION: added to acceleration_
        float steering_delta{0.F}; // [deg], added to rot_
    };

    enum class MovementMode
    {
        VELOCITY     = 0,
        ACCELERATION = 1,
        MANUAL       = 2 // keyboard control: a no-op here, as in the reference
    };

    Agent() = default;
    // builds the default sensor fan: -70..+70 degrees in steps of 10 (15 rays)
    Agent(Vec2d start_pos, float start_rot, int16_t id);
    virtual ~Agent() = default;

    // pose set, speed/acceleration/action zeroed, crashed_/timed_out_/completed_ cleared
    virtual void reset(const Vec2d &reset_pos, const float reset_rot);

    // Host-side kinematics with the same arithmetic (and the same sine/cosine) as the device step, for callers
    // that move an agent outside Environment::step().
    void move();
    void moveViaVelocity();
    void moveViaAcceleration();
    void moveViaUserInput();
    void setPose(const Vec2d pos, const float rot);
    bool isDone() const;
    void setMovementMode(const MovementMode mode) { movement_mode_ = mode; }
    inline void setHeadingDrawing(const bool draw_heading) { draw_agent_heading_ = draw_heading; }

    virtual void updateAction() = 0;

  public:
    Vec2d   pos_{};
    float   speed_{0.F};
    float   acceleration_{0.F};
    float   rot_{0.F}; // degrees, never wrapped
    float   radius_{9.0F};
    float   sensor_offset_{0.0F};
    int16_t id_{};
    int     color_[4]{80, 80, 80, 255};

    bool has_raycast_sensor_{true};
    bool manual_control_enabled_{true};
    bool draw_agent_heading_{true};

    std::vector<float> sensor_ray_angles_;
    float              sensor_range_{kSensorRange};

    bool crashed_{false};
    bool completed_{false};
    bool timed_out_{false};

    std::vector<Vec2d> sensor_hits_;      // "robot frame" hit points, one per ray, refreshed by every step
    std::vector<Pixel> pixels_until_hit_; // legacy, unused

    Action       current_action_{0.F, 0.F};
    MovementMode movement_mode_{MovementMode::VELOCITY};
};

template <typename TDerivedAgent>
inline std::vector<Agent *> createBaseAgentPtrs(const std::vector<std::unique_ptr<TDerivedAgent>> &derived_agents)
{
    std::vector<Agent *> out;
    out.reserve(derived_agents.size());
    for (const auto &a : derived_agents)
        out.push_back(a.get());
    return out;
}
