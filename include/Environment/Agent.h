// Agent.h -- one simulated car: pose, kinematic state, lidar fan, flags and the current action.
//
// Source-compatible with the reference's Agent (reference Environment/Agent.h:7-94): callers derive from it,
// implement updateAction(), write current_action_ and read sensor_hits_ / crashed_ / pos_ / rot_ directly,
// so every public member keeps its name, type and default value.  In this project the Agent object is a
// HOST-side mirror: Environment::step() gathers these fields into the device-resident struct-of-arrays
// state, runs the fused HIP step and scatters the results back (include/okenv.h, OKENV_F_* fields).
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "Typedefs.h"

class OKENV_CLASS Agent
{
  public:
    // ---- nested types -------------------------------------------------------------------------------------
    enum class MovementMode
    {
        VELOCITY     = 0, // throttle_delta IS the speed
        ACCELERATION = 1, // throttle_delta accumulates into acceleration_, which integrates into speed_
        MANUAL       = 2  // keyboard control: a no-op here, as in the reference
    };

    struct Action
    {
        float throttle_delta{0.F}; // meaning depends on the movement mode
        float steering_delta{0.F}; // [deg], added to rot_ every step
    };

    // ---- limits ---------------------------------------------------------------------------------------------
    static constexpr float kSensorRange{200.F};   // lidar range [px]; OK_SENSOR_RANGE on the device
    static constexpr float kSpeedLimit{100.F};    // ACCELERATION mode clamps speed_ to [0, kSpeedLimit]
    static constexpr float kRotationLimit{360.F}; // normalisation constant used by policies

    // ---- construction ---------------------------------------------------------------------------------------
    Agent() = default;
    Agent(Vec2d start_pos, float start_rot, int16_t id); // also builds the default fan: -70..+70 deg, every 10 deg
    virtual ~Agent() = default;

    // ---- what derived classes and callers use ------------------------------------------------------------------
    virtual void updateAction() = 0; // the policy: writes current_action_

    // pose set; speed_, acceleration_ and current_action_ zeroed; crashed_, timed_out_, completed_ cleared
    virtual void reset(const Vec2d &reset_pos, const float reset_rot);

    void setPose(const Vec2d pos, const float rot);
    bool isDone() const; // crashed_ || completed_
    void setMovementMode(const MovementMode mode) { movement_mode_ = mode; }
    inline void setHeadingDrawing(const bool draw_heading) { draw_agent_heading_ = draw_heading; }

    // Host-side kinematics with the arithmetic (and the sine/cosine) of the device step, for callers that move an
    // agent outside Environment::step().
    void move();
    void moveViaVelocity();
    void moveViaAcceleration();
    void moveViaUserInput();

  public:
    // ---- pose and kinematic state (uploaded before / downloaded after every step) -----------------------------
    Vec2d pos_{};
    float speed_{0.F};
    float acceleration_{0.F};
    float rot_{0.F}; // heading [deg], never wrapped

    // ---- appearance, identity -----------------------------------------------------------------------------------
    float   radius_{9.0F};        // drawing only: the crash test is purely lidar-based
    float   sensor_offset_{0.0F}; // lidar origin ahead of the centre, along the heading [px]
    int16_t id_{};
    AgentColor color_{}; // RGBA, dark gray

    bool has_raycast_sensor_{true};
    bool manual_control_enabled_{true};
    bool draw_agent_heading_{true};

    // ---- lidar --------------------------------------------------------------------------------------------------
    std::vector<float> sensor_ray_angles_; // [deg] relative to the heading; all agents of an Environment share one fan
    float              sensor_range_{kSensorRange};

    // ---- episode flags --------------------------------------------------------------------------------------------
    bool crashed_{false};   // set by the collision pass (a ray shorter than sqrt(2) px) or by the standstill timeout
    bool completed_{false}; // never set by the Environment
    bool timed_out_{false}; // set together with crashed_ when the standstill timeout fires

    // ---- observation ----------------------------------------------------------------------------------------------
    std::vector<Vec2d> sensor_hits_;      // one "robot frame" hit point per ray, refreshed by every step
    std::vector<Pixel> pixels_until_hit_; // legacy, unused

    // ---- control ------------------------------------------------------------------------------------------------
    Action       current_action_{0.F, 0.F};
    MovementMode movement_mode_{MovementMode::VELOCITY};
};

// Base-class pointers of a vector of owned derived agents, the form Environment's constructor takes.
template <typename TDerivedAgent>
inline std::vector<Agent *> createBaseAgentPtrs(const std::vector<std::unique_ptr<TDerivedAgent>> &derived_agents)
{
    std::vector<Agent *> out;
    out.reserve(derived_agents.size());
    for (const auto &a : derived_agents)
        out.push_back(a.get());
    return out;
}
