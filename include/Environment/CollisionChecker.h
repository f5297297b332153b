// CollisionChecker.h -- lidar raycast + crash test for a set of agents.
//
// Source-compatible with the reference's CollisionChecker (reference Environment/CollisionChecker.h:8-24).
// checkCollision() runs the collision pass alone (ray build, first-hit raycast against the segments, hit transform
// into sensor_hits_, crash flag) on the GPU through okenv_collide (include/okenv.h); kinematics are not touched.
#pragma once

#include <cstddef>
#include <memory>
#include <vector>

#include "Agent.h"
#include "Typedefs.h"

struct okenv;

class OKENV_CLASS CollisionChecker
{
  public:
    // `d_segments` may be a device pointer (as TrackSegments::getDeviceSegments() returns) or a host pointer.
    CollisionChecker(const Segment2d *d_segments, size_t num_segments, const std::vector<Agent *> &agents);
    ~CollisionChecker();

    void checkCollision();

    // one Ray_ per agent x ray: origin, world angle, world hit point, active flag (valid after checkCollision /
    // Environment::step)
    const Ray_ *getHostRays() const;
    size_t      getNumRays() const;

    // used by Environment: the underlying C-ABI handle, and one whole Environment::step for `agents` (kinematics,
    // standstill bookkeeping, collision pass) as a single packed exchange with the device.  The four arrays hold the
    // agents' DisplacementStats members and are updated in place.
    okenv *handle() const;
    void   stepAgents(const std::vector<Agent *> &agents, uint32_t *disp_ctr, float *disp_x, float *disp_y, uint8_t *disp_timed_out);

  private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};
