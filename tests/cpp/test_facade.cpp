// test_facade.cpp -- drives the C++ drop-in classes (include/Environment/*.h) the way the reference's callers do
// (shape of Template/main.cpp:71-125 and, for the legacy constructor, EvolutionaryRacer/genetic_learner_sim.cpp:28-38)
// and dumps the trajectory; tests/test_facade.py replays the same actions through the CPU oracle and compares
// every float bit for bit.
#include <cstdint>
#include <cstdio>
#include <memory>
#include <vector>

#include "Environment/Environment.h"

class ScriptedAgent : public Agent
{
  public:
    ScriptedAgent(const Vec2d p, const float rot, const int16_t id) : Agent(p, rot, id) {}
    void updateAction() override
    {
        // integer-valued actions: exactly representable, trivially reproduced by the Python side
        current_action_.throttle_delta = parked_ ? 0.F : static_cast<float>((id_ * 7 + step_ * 3) % 60) * (movement_mode_ == MovementMode::ACCELERATION ? 0.25F : 1.F);
        current_action_.steering_delta = static_cast<float>(((id_ * 5 + step_) % 7) - 3);
        ++step_;
    }
    bool parked_{false};
    int  step_{0};
};

static void dump(FILE *f, const std::vector<std::unique_ptr<ScriptedAgent>> &agents, const Environment &env)
{
    for (size_t i = 0; i < agents.size(); ++i)
    {
        const auto &a = agents[i];
        const float head[6] = {a->pos_.x, a->pos_.y, a->rot_, a->speed_, a->acceleration_, 0.F};
        fwrite(head, 4, 6, f);
        const uint32_t flags[3] = {a->crashed_ ? 1U : 0U, a->timed_out_ ? 1U : 0U, env.displacement_stats_[i].displacement_ctr};
        fwrite(flags, 4, 3, f);
        for (const auto &h : a->sensor_hits_)
        {
            const float xy[2] = {h.x, h.y};
            fwrite(xy, 4, 2, f);
        }
    }
}

int main(int argc, char **argv)
{
    if (argc != 5)
    {
        fprintf(stderr, "usage: test_facade <track.csv> <out.bin> <num_agents> <steps>\n");
        return 2;
    }
    const int n     = atoi(argv[3]);
    const int steps = atoi(argv[4]);
    std::vector<std::unique_ptr<ScriptedAgent>> agents;
    for (int16_t i = 0; i < n; ++i)
        agents.push_back(std::make_unique<ScriptedAgent>(Vec2d{0, 0}, 0, i));

    FILE *f = fopen(argv[2], "wb");
    {
        // ---- current constructor (Template/main.cpp:77) -------------------------------------------------------
        Environment env(argv[1], createBaseAgentPtrs(agents), /*draw_rays=*/true, /*hidden_window=*/true);
        int         callbacks = 0;
        env.visualizer_->user_draw_callback_ = [&callbacks]() { ++callbacks; };
        const auto &d = env.race_track_->track_data_points_;
        for (int i = 0; i < n; ++i)
        {
            const size_t idx = (static_cast<size_t>(i) * 37U + RaceTrack::kStartingIdx) % d.x_m.size();
            agents[i]->reset({d.x_m[idx], d.y_m[idx]}, env.race_track_->headings_[idx]);
            agents[i]->setMovementMode(i % 2 ? Agent::MovementMode::ACCELERATION : Agent::MovementMode::VELOCITY);
            agents[i]->parked_ = (i % 5 == 4);
        }
        env.step(); // initial observation (Template/main.cpp:103)
        dump(f, agents, env);
        for (int s = 0; s < steps; ++s)
        {
            for (auto &a : agents)
                a->updateAction();
            env.step();
            dump(f, agents, env);
            if (s == steps / 2)
            { // caller-side reset of whoever crashed, as every reference app does between episodes
                for (int i = 0; i < n; ++i)
                    if (agents[i]->crashed_ && i % 2 == 0)
                        agents[i]->reset({d.x_m[3], d.y_m[3]}, env.race_track_->headings_[3]);
            }
        }
        if (callbacks != steps + 1)
            return 3;
        // Ray_ view and the collision pass on its own
        const Ray_ *rays = env.collision_checker_->getHostRays();
        if (env.collision_checker_->getNumRays() != static_cast<size_t>(n) * agents[0]->sensor_ray_angles_.size() || rays == nullptr)
            return 4;
        env.collision_checker_->checkCollision();
        dump(f, agents, env);
        // resetAgent: default start point, and the randomised variants stay on the track
        env.resetAgent(agents[0].get(), false);
        if (agents[0]->pos_.x != d.x_m[3] || agents[0]->rot_ != env.race_track_->headings_[3] || agents[0]->crashed_)
            return 5;
        Environment::seedRandom(7);
        for (int k = 0; k < 50; ++k)
        {
            env.resetAgent(agents[1].get(), true, true, true);
            const size_t near = env.race_track_->findNearestTrackIndexBruteForce(agents[1]->pos_);
            if (agents[1]->pos_.distanceSquared({d.x_m[near], d.y_m[near]}) > 30.F * 30.F)
                return 6;
        }
    }
    {
        // ---- legacy constructor + setAgent (AutoEncoder/collect_data_racetrack/Environment.hpp:25-34) ------------
        Environment env(argv[1]);
        std::vector<std::unique_ptr<ScriptedAgent>> few;
        for (int16_t i = 0; i < 3; ++i)
        {
            few.push_back(std::make_unique<ScriptedAgent>(Vec2d{0, 0}, 0, i));
            env.setAgent(few.back().get());
        }
        const auto &d = env.race_track_->track_data_points_;
        for (auto &a : few)
            a->reset({d.x_m[3], d.y_m[3]}, env.race_track_->headings_[0]);
        for (int s = 0; s < 20; ++s)
        {
            for (auto &a : few)
                a->updateAction();
            env.step();
        }
        dump(f, few, env);
    }
    fclose(f);
    return 0;
}
