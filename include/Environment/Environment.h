// Environment.h -- the step facade callers drive: move every agent, time out the ones standing still, cast the
// lidar fans against the track and flag crashes.
//
// Source-compatible with the reference's Environment (reference Environment/Environment.h:17-76): same public
// members and methods, the current constructor `(path, agents, draw_rays, hidden_window)` AND the legacy pair
// `Environment(path)` + `setAgent(Agent*)` that EvolutionaryRacer and RLRacers/Q_Learning still use
// (reference AutoEncoder/collect_data_racetrack/Environment.hpp:25-34).  step() is one fused HIP launch on the
// MI355X (include/okenv.h); the Raylib visualizer is replaced by a headless stub (Visualizer.h).
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "Agent.h"
#include "CollisionChecker.h"
#include "RaceTrack.h"
#include "ScreenGrabber.h"
#include "TrackSegments.h"
#include "Visualizer.h"

// Standstill bookkeeping: an agent that has moved less than 20 px over 200 steps is timed out
// (reference Environment/Environment.h:17-27).
struct DisplacementStats
{
    static constexpr uint32_t kPeriod{200};
    static constexpr float    kDisplamentThreshold{20.0F};

    bool     displacement_timed_out{false};
    uint32_t displacement_ctr{0U};
    Vec2d    init_pos{0.F, 0.F};
};

class OKENV_CLASS Environment
{
  public:
    // ---- construction -----------------------------------------------------------------------------------------
    // current surface: the population is fixed at construction (reference Environment/Environment.h:31-34)
    Environment(const std::string          &race_track_path,
                const std::vector<Agent *> &agents,
                const bool                  draw_rays     = true,
                const bool                  hidden_window = false);
    // legacy surface: agents are registered afterwards with setAgent(); device buffers are (re)built lazily at the
    // next step() because the ray count is unknown until then
    explicit Environment(const std::string &race_track_path);
    void     setAgent(Agent *agent);
    ~Environment();

    // ---- the step ---------------------------------------------------------------------------------------------
    // One Environment step for every registered agent: Agent::move + standstill check for agents that have not
    // crashed, then the collision pass for all of them, then (headless) render.  One fused HIP launch.
    void step();

    // ---- resets -----------------------------------------------------------------------------------------------
    // Resets the agent onto the track: the start point (index 3) or, if pick_random_point, a random centre-line
    // point, optionally at a random lateral position between the inner boundaries and with a heading offset of
    // +-(45..90) degrees alternating in sign.
    void    resetAgent(Agent *agent, const bool pick_random_point = true, const bool randomize_lane = false, const bool randomize_heading = false);
    int32_t pickRandomResetTrackIdx() const;
    // Seeds the generator behind pickRandomResetTrackIdx / resetAgent (the reference uses raylib's unseeded
    // GetRandomValue; here runs are reproducible).
    static void seedRandom(uint32_t seed);
    static int  randomValue(int lo, int hi); // uniform integer in [lo, hi]

    // ---- window-related members kept for source compatibility (headless here) --------------------------------------
    void drawSensorRanges(const std::vector<Vec2d> &sensor_hits);
    bool isEnterPressed() const;
    void saveImage(const std::string &filename) const;
    std::vector<uint8_t>            getRenderTargetHost() const { return screen_grabber_->getRenderTargetHost(); }
    ScreenGrabber::RenderTargetInfo getRenderTargetInfo() const { return screen_grabber_->getRenderTargetInfo(); }

  public:
    std::unique_ptr<RaceTrack>        race_track_;         // geometry; callers read track_data_points_ / headings_
    std::unique_ptr<TrackSegments>    track_segments_;     // the 4P boundary segments
    std::unique_ptr<env::Visualizer>  visualizer_;         // headless stub (user_draw_callback_ still fires)
    std::vector<Agent *>              agents_;             // not owned
    std::vector<DisplacementStats>    displacement_stats_; // one per agent, NOT touched by resets
    std::unique_ptr<CollisionChecker> collision_checker_{nullptr};
    std::unique_ptr<ScreenGrabber>    screen_grabber_{nullptr};

    bool draw_rays_{false};

  private:
    void ensureChecker();
    bool checker_stale_{true};
};
