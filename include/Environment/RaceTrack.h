// RaceTrack.h -- race track geometry from a TUMFTM racetrack-database CSV.
//
// Source-compatible with the reference's RaceTrack (reference Environment/RaceTrack.h:15-80): same public
// data members and query methods, and bit-identical results (tests/test_track_parity.py compares every
// float with the reference's compiled RaceTrack.cpp and with the oracle).  Host-only setup code: it runs
// once per Environment; its output (centre line, headings, four boundary polylines) is what gets uploaded
// to the GPU.
#pragma once

#include <cstddef>
#include <string>
#include <vector>

#include "Typedefs.h"

class OKENV_CLASS RaceTrack
{
  public:
    static constexpr size_t kStartingIdx{3}; // centre-line index agents start from

    struct TrackData
    {
        std::vector<float> x_m;
        std::vector<float> y_m;
        std::vector<float> w_tr_right_m;
        std::vector<float> w_tr_left_m;
    };

    RaceTrack() = delete;
    explicit RaceTrack(const std::string &track_csv_path);

    // argmin over the centre line of the squared distance to query_pt; lowest index wins ties
    size_t findNearestTrackIndexBruteForce(const Vec2d &query_pt) const;
    // distance to the nearest inner-boundary POINT (left or right)
    float getNearestDistanceToTrackBoundary(const Vec2d &query_pt) const;
    // distance to the nearest centre-line point divided by the lane width there
    float getDistanceToLaneCenter(const Vec2d &query_pt) const;

    bool loadedOk() const { return loaded_ok_; }

  public:
    std::string        track_name_{};
    TrackData          track_data_points_{};
    std::vector<Vec2d> left_bound_inner_, left_bound_outer_, right_bound_inner_, right_bound_outer_;
    std::vector<Vec2d> start_line_, finish_line_;
    std::vector<float> headings_{};

  private:
    bool readCsv(const std::string &path);
    void fitToWindow(float window_width, float window_height);
    void buildLanes();

    bool loaded_ok_{false};
};
