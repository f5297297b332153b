// genetic_learner_sim.cpp -- the EvolutionaryRacer application over the batched Environment (C ABI, include/okenv.h).
//
// Replaces the reference's EvolutionaryRacer/genetic_learner_sim.cpp: the same generation loop (:47-96 -- rollout until
// every agent has crashed, assignScores, saveBestAgentNetwork's regression check, colony average, chooseAndMateAgents,
// reset to the start line), with the per-agent work of the loop body (GeneticAgent::updateAction, Environment::step,
// scores, mating) done for the whole population on the GPU.  The reference runs 50 agents with the default 15-ray fan
// (kNumAgents, :18); population, fan and hidden width are command-line parameters here.  Window, score plot and the
// shared-memory queue of the original (VisUtils.hpp, spmc_queue.h) are not part of the path.
//
//   genetic_learner_sim track.csv [--agents N] [--rays R] [--hidden H] [--generations G] [--seed S] [--max-steps M]
//                                 [--steps-per-launch L] [--dump file]
//
// --dump writes, per generation, {int32 steps, float scores[N], int32 parents[5]} and finally the best agent's weights
// (the reference's agent_weights_{1,2}.txt, MiscUtils.hpp:52-59, as one padded block) -- what the parity test replays on
// the CPU oracle.
#include <cstdint>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "okenv.h"

namespace
{
struct Options
{
    std::string track;
    int         agents{50}, rays{15}, hidden{30}, generations{5}, max_steps{4000}, steps_per_launch{50};
    uint32_t    seed{1234};
    std::string dump;
};

bool parse(int argc, char **argv, Options &o)
{
    if (argc < 2)
        return false;
    o.track = argv[1];
    for (int i = 2; i + 1 < argc; i += 2)
    {
        const std::string k = argv[i];
        const char       *v = argv[i + 1];
        if (k == "--agents") o.agents = std::atoi(v);
        else if (k == "--rays") o.rays = std::atoi(v);
        else if (k == "--hidden") o.hidden = std::atoi(v);
        else if (k == "--generations") o.generations = std::atoi(v);
        else if (k == "--seed") o.seed = static_cast<uint32_t>(std::strtoul(v, nullptr, 10));
        else if (k == "--max-steps") o.max_steps = std::atoi(v);
        else if (k == "--steps-per-launch") o.steps_per_launch = std::atoi(v);
        else if (k == "--dump") o.dump = v;
        else return false;
    }
    return true;
}

#define CHECK(call)                                                                                                    \
    do                                                                                                                 \
    {                                                                                                                  \
        const int rc_ = (call);                                                                                        \
        if (rc_ != OKENV_OK)                                                                                           \
        {                                                                                                              \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, okenv_last_error(env));                           \
            return 2;                                                                                                  \
        }                                                                                                              \
    } while (0)
} // namespace

int main(int argc, char **argv)
{
    Options opt;
    if (!parse(argc, argv, opt))
    {
        std::fprintf(stderr, "Provide a file path for the track csv file [--agents N --rays R --hidden H --generations G --seed S "
                             "--max-steps M --steps-per-launch L --dump file]\n");
        return -1;
    }
    okenv_t       env = nullptr;
    okenv_track_t tr  = nullptr;
    if (okenv_track_load(&tr, opt.track.c_str()) != OKENV_OK)
    {
        std::fprintf(stderr, "cannot load %s\n", opt.track.c_str());
        return 2;
    }
    const int          P = okenv_track_num_points(tr), S = okenv_track_num_segments(tr);
    std::vector<float> seg(4 * static_cast<size_t>(S)), cx(P), cy(P), heading(P);
    okenv_track_segments(tr, seg.data());
    okenv_track_get(tr, 0, cx.data());
    okenv_track_get(tr, 1, cy.data());
    okenv_track_get(tr, 4, heading.data());
    // sensor fan: -70 ... +70 degrees (Agent.cpp:11-18 builds it every 10 degrees = 15 rays)
    std::vector<float> fan(opt.rays);
    for (int i = 0; i < opt.rays; ++i)
        fan[i] = opt.rays == 1 ? 0.F : -70.0F + 140.0F * static_cast<float>(i) / static_cast<float>(opt.rays - 1);
    CHECK(okenv_create(&env, seg.data(), S, opt.agents, opt.rays, fan.data(), 0, OKENV_FLAG_NONE, 0.0F));
    CHECK(okenv_set_centerline(env, cx.data(), cy.data(), heading.data(), P));
    const int N = opt.agents;
    // GeneticAgent drives by acceleration (GeneticAgent.hpp:28,34)
    std::vector<uint8_t> mode(N, OKENV_MODE_ACCELERATION);
    CHECK(okenv_set_field(env, OKENV_F_MODE, mode.data()));
    CHECK(okenv_policy_mlp_create(env, opt.hidden, opt.seed, 0));
    // start pose: centre-line point kStartingIdx (= 3) with the heading of point 0 (genetic_learner_sim.cpp:34-36)
    const float start_x = cx[3], start_y = cy[3], start_rot = heading[0];

    std::FILE *dump = opt.dump.empty() ? nullptr : std::fopen(opt.dump.c_str(), "wb");
    std::vector<float> scores(N), colony_avg_scores;
    float              prev_gen_best_score = 0.F, top_score_all_time = 0.F;
    std::vector<float> best_weights;
    for (int episode_idx = 0; episode_idx < opt.generations; ++episode_idx)
    {
        // agent.reset(...) for everybody, then one step for the initial observation (:65-75)
        CHECK(okenv_reset_all(env, start_x, start_y, start_rot));
        CHECK(okenv_step(env, 1));
        // { updateAction for all; env.step(); all_done? } (:76-93), steps_per_launch iterations per kernel launch, as an episode
        // (include/okenv.h): only the agents that can still change are stepped, and `iteration` is the reference's own count --
        // its loop leaves with the step in which the last agent crashes, wherever the launches end
        CHECK(okenv_episode_begin(env));
        int32_t tail = 0;
        CHECK(okenv_episode_tail_limit(env, &tail));
        int     iteration = 1;
        int32_t alive = N, listed = N;
        while (alive > 0 && iteration < opt.max_steps)
        { // a short list is stepped one agent per workgroup, each leaving with its agent: one launch for all that is left
            const int n = listed <= tail ? opt.max_steps - iteration : std::min(opt.steps_per_launch, opt.max_steps - iteration);
            CHECK(okenv_rollout_policy(env, n));
            iteration += n;
            CHECK(okenv_episode_compact(env, &alive, &listed));
        }
        int32_t loop_steps = 0;
        CHECK(okenv_episode_end(env, &loop_steps, nullptr));
        iteration = 1 + loop_steps;
        std::printf("------------ EPISODE %d DONE ---------------\n", episode_idx);
        // assignScores (MiscUtils.hpp:64-71)
        CHECK(okenv_ga_scores(env, scores.data()));
        // saveBestAgentNetwork (MiscUtils.hpp:26-62): best of this generation, regression check, keep the all-time best
        size_t top_scorer_agent_id = 0;
        float  best_score_current  = 0.F;
        float  current_avg         = 0.F;
        for (size_t i = 0; i < scores.size(); ++i)
        {
            if (scores[i] > best_score_current)
            {
                best_score_current  = scores[i];
                top_scorer_agent_id = i;
            }
            current_avg += scores[i];
        }
        current_avg /= static_cast<float>(scores.size());
        if (prev_gen_best_score > best_score_current)
        {
            std::fprintf(stderr, "!! Current gen. high score %f is less than previous gen. high score %f\n", best_score_current,
                         prev_gen_best_score);
            return 3; // the reference throws here
        }
        prev_gen_best_score = best_score_current;
        if (best_score_current > top_score_all_time)
        {
            const int per = okenv_policy_mlp_weights_per_agent(env);
            std::vector<float> all(static_cast<size_t>(N) * per);
            CHECK(okenv_policy_mlp_get_weights(env, all.data()));
            best_weights.assign(all.begin() + static_cast<long>(top_scorer_agent_id) * per, all.begin() + static_cast<long>(top_scorer_agent_id + 1) * per);
            top_score_all_time = best_score_current;
        }
        colony_avg_scores.push_back(current_avg);
        std::printf("top: %g  colony average: %g  steps: %d\n", best_score_current, current_avg, iteration);
        // chooseAndMateAgents (Mating.hpp:108-166)
        int32_t parents[5] = {-1, -1, -1, -1, -1};
        CHECK(okenv_ga_select_mate(env, opt.seed, static_cast<uint32_t>(episode_idx), 0, parents));
        if (dump)
        {
            const int32_t steps = iteration;
            std::fwrite(&steps, 4, 1, dump);
            std::fwrite(scores.data(), 4, scores.size(), dump);
            std::fwrite(parents, 4, 5, dump);
        }
    }
    if (dump)
    {
        std::fwrite(best_weights.data(), 4, best_weights.size(), dump);
        std::fclose(dump);
    }
    okenv_destroy(env);
    okenv_track_free(tr);
    return 0;
}
