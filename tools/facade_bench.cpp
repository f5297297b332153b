// facade_bench.cpp -- microseconds per Environment::step() of the C++ drop-in classes (include/Environment/) for small
// populations, the regime every reference application lives in (Template: 1 agent, PPO / Reinforce: 15, EvolutionaryRacer: 50).
// Built and run by bench.py (`callers.facade_step_us`).   usage: facade_bench track.csv steps n1 [n2 ...]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "Environment/Environment.h"

class Driver : public Agent
{
  public:
    Driver(const Vec2d p, const float rot, const int16_t id) : Agent(p, rot, id)
    {
        sensor_ray_angles_ = {-70.F, -30.F, 0.F, 30.F, 70.F}; // the five-ray fan the RL applications use
    }
    void updateAction() override
    {
        current_action_.throttle_delta = 40.F;
        current_action_.steering_delta = static_cast<float>((id_ + step_++) % 5) - 2.F;
    }
    int step_{0};
};

int main(int argc, char **argv)
{
    if (argc < 4)
        return 2;
    const int steps = std::atoi(argv[2]);
    for (int k = 3; k < argc; ++k)
    {
        const int n = std::atoi(argv[k]);
        std::vector<std::unique_ptr<Driver>> agents;
        for (int16_t i = 0; i < n; ++i)
            agents.push_back(std::make_unique<Driver>(Vec2d{0, 0}, 0, i));
        Environment env(argv[1], createBaseAgentPtrs(agents), false, true);
        for (auto &a : agents)
            env.resetAgent(a.get(), true, false, false);
        auto loop = [&](int count) {
            for (int s = 0; s < count; ++s)
            {
                for (auto &a : agents)
                {
                    if (a->crashed_)
                        env.resetAgent(a.get(), true, false, false);
                    a->updateAction();
                }
                env.step();
            }
        };
        loop(200);
        const auto t0 = std::chrono::steady_clock::now();
        loop(steps);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps;
        std::printf("%d %.3f\n", n, us);
    }
    return 0;
}
