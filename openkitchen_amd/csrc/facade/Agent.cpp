// Agent.cpp -- see include/Environment/Agent.h.  Host-side mirror of what the device step does per agent
// (okenv_kernels.h: okAgentPreStep), with the same constants and the same ok_sincosf, so a caller that moves an
// agent by hand gets the bits the GPU would have produced (reference Environment/Agent.cpp:8-19,21-47,82-135).
#include "Environment/Agent.h"

#include <cassert>
#include <iostream>

#include "okenv_math.h"

Agent::Agent(Vec2d start_pos, float start_rot, int16_t id) : pos_{start_pos}, rot_{start_rot}, id_{id}
{
    if (has_raycast_sensor_)
    {
        for (int deg = -70; deg <= 70; deg += 10)
            sensor_ray_angles_.push_back(static_cast<float>(deg));
    }
}

void Agent::move()
{
    switch (movement_mode_)
    {
    case MovementMode::VELOCITY: moveViaVelocity(); break;
    case MovementMode::ACCELERATION: moveViaAcceleration(); break;
    case MovementMode::MANUAL: moveViaUserInput(); break;
    default:
        std::cerr << "Unimplemented movement mode." << std::endl;
        assert(false);
        break;
    }
}

void Agent::moveViaUserInput()
{
    // keyboard control needs a window; nothing to do in the headless build (it is a stub in the reference as well)
}

namespace
{
// pos += ((cos|sin)(kDeg2Rad * rot) * speed) * dt, x first, all fp32
void advancePose(Vec2d &pos, const float rot_deg, const float speed)
{
    float sn, cs;
    ok_sincosf(OK_DEG2RAD * rot_deg, &sn, &cs);
    const float dx = cs * speed * OK_DT;
    pos.x += dx;
    const float dy = sn * speed * OK_DT;
    pos.y += dy;
}
} // namespace

void Agent::moveViaAcceleration()
{
    rot_ += current_action_.steering_delta;
    acceleration_ += current_action_.throttle_delta;
    speed_ += (acceleration_ * OK_DT);
    speed_ = (speed_ < 0.F) ? 0.F : speed_;
    speed_ = (speed_ > kSpeedLimit) ? kSpeedLimit : speed_;
    advancePose(pos_, rot_, speed_);
}

void Agent::moveViaVelocity()
{
    rot_ += current_action_.steering_delta;
    speed_ = current_action_.throttle_delta;
    advancePose(pos_, rot_, speed_);
}

void Agent::setPose(const Vec2d pos, const float rot)
{
    pos_ = pos;
    rot_ = rot;
}

void Agent::reset(const Vec2d &reset_pos, const float reset_rot)
{
    pos_            = reset_pos;
    rot_            = reset_rot;
    acceleration_   = 0.F;
    speed_          = 0.F;
    crashed_        = false;
    timed_out_      = false;
    completed_      = false;
    current_action_ = Action{0.F, 0.F};
}

bool Agent::isDone() const
{
    return crashed_ || completed_;
}
