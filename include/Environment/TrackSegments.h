// TrackSegments.h -- the race track's boundary polylines flattened into Segment2d[4P].
//
// Source-compatible with the reference's TrackSegments (reference Environment/TrackSegments.h:8-37): the order is
// left-inner, left-outer, right-inner, right-outer runs followed by the four closing segments
// (reference Environment/TrackSegments.cu:6-42).  The array lives on the host (the grid is built from it) and,
// for callers that hand getDeviceSegments() to a CollisionChecker, in device memory too.
#pragma once

#include <cstddef>
#include <vector>

#include "RaceTrack.h"
#include "Typedefs.h"

class OKENV_CLASS TrackSegments
{
  public:
    explicit TrackSegments(const RaceTrack &race_track);
    ~TrackSegments();
    TrackSegments(const TrackSegments &)            = delete;
    TrackSegments &operator=(const TrackSegments &) = delete;

    const Segment2d *getDeviceSegments() const { return d_segments_; }
    size_t           getNumSegments() const { return segments_.size(); }
    const std::vector<Segment2d> &getHostSegments() const { return segments_; }

  private:
    std::vector<Segment2d> segments_;
    Segment2d             *d_segments_{nullptr};
};
