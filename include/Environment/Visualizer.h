// Visualizer.h -- headless stand-in for the reference's Raylib visualizer (reference Environment/Visualizer.h).
//
// The hot path runs with the visualizer compiled out (BASELINE.json north_star); this class keeps the members
// callers touch (`env.visualizer_->user_draw_callback_ = ...`, setAgentToFollow, camera_) compiling.  render()
// draws nothing; it only invokes the user callback so that callers that count frames or log inside it keep working.
#pragma once

#include <cstdint>
#include <functional>
#include <vector>

#include "Agent.h"
#include "RaceTrack.h"

class CollisionChecker;

// ---- what callers use from the drawing library inside their draw callbacks, headless ------------------------------
// The reference's applications overlay text from `user_draw_callback_` (`DrawText(buffer, kScreenWidth - 150, 40, 20,
// YELLOW)`, RLRacers/PPO/ppo_sim.cpp:38-43) and draw their random numbers from the same library
// (`GetRandomValue(lo, hi)`, inclusive).  Without a window the former does nothing; the latter is served by the
// generator behind Environment::resetAgent, so Environment::seedRandom makes such callers reproducible.
// (Color and its constants: Typedefs.h)
inline void DrawText(const char * /*text*/, int /*x*/, int /*y*/, int /*font_size*/, Color /*color*/) {}
OKENV_CLASS int GetRandomValue(int min, int max);

namespace env
{
// The frame Visualizer::render draws in the reference (Visualizer.cpp:159-229), rasterised on the CPU into an RGBA8 buffer
// (openkitchen_amd/csrc/facade/Visualizer.cpp).  `rays` may be null (draw_rays_ off).
OKENV_CLASS void paintFrame(std::vector<uint8_t> &rgba, int width, int height, const RaceTrack &track, const std::vector<Agent *> &agents,
                            const CollisionChecker *rays);

class Visualizer
{
  public:
    explicit Visualizer(bool hidden_window = false) : hidden_window_(hidden_window) {}

    void setAgentToFollow(const Agent *agent) { agent_to_follow_ = agent; }
    void enableDrawingSensorRays() { draw_rays_ = true; }
    void disableDrawingSensorRays() { draw_rays_ = false; }
    void close() {}

    void render(const RaceTrack & /*track*/, const std::vector<Agent *> & /*agents*/, const CollisionChecker * /*rays*/)
    {
        if (user_draw_callback_)
            user_draw_callback_();
        ++frames_rendered_;
    }

  public:
    std::function<void()> user_draw_callback_{};
    const Agent          *agent_to_follow_{nullptr};
    bool                  draw_rays_{true};
    bool                  hidden_window_{false};
    unsigned long         frames_rendered_{0};
};
} // namespace env
