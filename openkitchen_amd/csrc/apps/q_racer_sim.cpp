// q_racer_sim.cpp -- the tabular Q-learning application over the batched Environment (C ABI, include/okenv.h).
//
// Replaces the reference's RLRacers/Q_Learning/q_racer_sim.cpp: the same episode loop (:123-216 -- reset every agent to
// one centre-line point, initial observation, { updateAction; env.step(); discretizeState; reward; learn } until every
// agent has crashed, epsilon decay, next reset point, optional shareCumulativeKnowledge), with the loop body done for the
// whole population on the GPU (okenv_rollout_q).  The reference runs 30 agents with a five-ray fan (kNumAgents :12,
// QAgent.hpp:56-62); population and fan are parameters here.  raylib's GetRandomValue in pickResetPosition (:18-21) is
// replaced by a Philox draw keyed (seed; episode).
//
//   q_racer_sim track.csv [--agents N] [--rays R] [--episodes E] [--seed S] [--max-steps M] [--steps-per-launch L]
//                         [--share 0|1] [--dump file]
//
// --dump writes, per episode, {int32 steps, int32 reset_idx, float epsilon} and finally the Q tables [N][243][3] and the
// per-agent (state, action, prev_track_idx) -- what the parity test replays on the CPU oracle.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "okenv.h"
#include "okenv_math.h"

namespace
{
struct Options
{
    std::string track;
    int         agents{30}, rays{5}, episodes{5}, max_steps{4000}, steps_per_launch{50}, share{0};
    uint32_t    seed{1234};
    std::string dump;
};

bool parse(int argc, char **argv, Options &o)
{
    if (argc < 2 || (argc - 2) % 2 != 0) // (every option takes a value)
        return false;
    o.track = argv[1];
    for (int i = 2; i + 1 < argc; i += 2)
    {
        const std::string k = argv[i];
        const char       *v = argv[i + 1];
        if (k == "--agents") o.agents = std::atoi(v);
        else if (k == "--rays") o.rays = std::atoi(v);
        else if (k == "--episodes") o.episodes = std::atoi(v);
        else if (k == "--seed") o.seed = static_cast<uint32_t>(std::strtoul(v, nullptr, 10));
        else if (k == "--max-steps") o.max_steps = std::atoi(v);
        else if (k == "--steps-per-launch") o.steps_per_launch = std::atoi(v);
        else if (k == "--share") o.share = std::atoi(v);
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
        std::fprintf(stderr, "Provide a file path for the track csv file [--agents N --rays R --episodes E --seed S --max-steps M "
                             "--steps-per-launch L --share 0|1 --dump file]\n");
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
    // the reference's fan is {-70, -30, 0, 30, 70} (QAgent.hpp:56-62); wider fans spread evenly over the same range and the
    // state uses the five rays nearest to those angles (okenv_q_create)
    std::vector<float> fan(opt.rays);
    if (opt.rays == 5)
        fan = {-70.F, -30.F, 0.F, 30.F, 70.F};
    else
        for (int i = 0; i < opt.rays; ++i)
            fan[i] = -70.0F + 140.0F * static_cast<float>(i) / static_cast<float>(opt.rays - 1);
    CHECK(okenv_create(&env, seg.data(), S, opt.agents, opt.rays, fan.data(), 0, OKENV_FLAG_NONE, 0.0F));
    CHECK(okenv_set_centerline(env, cx.data(), cy.data(), heading.data(), P));
    const int            N = opt.agents;
    std::vector<uint8_t> mode(N, OKENV_MODE_VELOCITY); // the action map sets the speed directly (QAgent.hpp:40-42)
    CHECK(okenv_set_field(env, OKENV_F_MODE, mode.data()));
    CHECK(okenv_q_create(env));

    std::FILE *dump        = opt.dump.empty() ? nullptr : std::fopen(opt.dump.c_str(), "wb");
    if (!opt.dump.empty() && dump == nullptr)
    {
        std::fprintf(stderr, "cannot write %s\n", opt.dump.c_str());
        return 2;
    }
    float      epsilon     = 0.9F;  // QAgent.hpp:26
    const float kEpsilonDiscount = 0.05F; // QAgent.hpp:27
    int32_t    reset_idx   = 3;     // RaceTrack::kStartingIdx (q_racer_sim.cpp:114)
    uint32_t   steps_total = 0;
    for (int episode_idx = 0; episode_idx < opt.episodes; ++episode_idx)
    {
        std::printf("------------ EPISODE %d DONE ---------------\neps: %g\n", episode_idx, epsilon);
        // reset, initial observation, current_state_idx_ = discretizeState() (:129-154)
        CHECK(okenv_q_begin_episode(env, reset_idx));
        // The loop of :156-190 leaves with the step in which the last agent crashes.  Launches of steps_per_launch steps overrun
        // that step; run as an episode (include/okenv.h) they step only the agents that can still change, and
        // okenv_episode_end returns the loop's own length and leaves agents and tables as the loop does -- a crashed agent's
        // per-step draw and -200 update (:158-182) are made there, up to that last step and not beyond.
        CHECK(okenv_episode_begin(env));
        int32_t tail = 0;
        CHECK(okenv_episode_tail_limit(env, &tail));
        int     steps = 0;
        int32_t alive = N, listed = N;
        while (alive > 0 && steps < opt.max_steps)
        { // a short list is stepped one agent per workgroup, each leaving with its agent: one launch for all that is left
            const int n = listed <= tail ? opt.max_steps - steps : std::min(opt.steps_per_launch, opt.max_steps - steps);
            CHECK(okenv_rollout_q(env, n, epsilon, opt.seed, 0, steps_total + static_cast<uint32_t>(steps)));
            steps += n;
            CHECK(okenv_episode_compact(env, &alive, &listed));
        }
        int32_t loop_steps = 0;
        CHECK(okenv_episode_end(env, &loop_steps, nullptr));
        steps = loop_steps;
        steps_total += static_cast<uint32_t>(steps);
        if (dump)
        {
            const int32_t rec[2] = {steps, reset_idx};
            std::fwrite(rec, 4, 2, dump);
            std::fwrite(&epsilon, 4, 1, dump);
        }
        // epsilon decay once everybody is done (:192-205)
        epsilon = (epsilon > kEpsilonDiscount) ? epsilon - kEpsilonDiscount : 0.F;
        // pickResetPosition (:18-21): GetRandomValue(0, P - 1) -> Philox(counter = (episode, 0, 5, 0))
        const ok_u32x4 r = ok_philox4x32(static_cast<uint32_t>(episode_idx), 0U, 5U, 0U, opt.seed, 0x6F6B656EU);
        reset_idx        = static_cast<int32_t>(ok_index_from_word(r.v[0], static_cast<uint32_t>(P)));
        if (opt.share)
            CHECK(okenv_q_share_knowledge(env)); // shareCumulativeKnowledge (:24-75), off by default in the reference (:16)
    }
    if (dump)
    {
        std::vector<float>   table(static_cast<size_t>(N) * 243 * 3);
        std::vector<int32_t> st(N), ac(N), pv(N);
        CHECK(okenv_q_get_table(env, table.data()));
        CHECK(okenv_q_get_state(env, st.data(), ac.data(), pv.data()));
        std::fwrite(table.data(), 4, table.size(), dump);
        std::fwrite(st.data(), 4, st.size(), dump);
        std::fwrite(ac.data(), 4, ac.size(), dump);
        std::fwrite(pv.data(), 4, pv.size(), dump);
        std::fclose(dump);
    }
    okenv_destroy(env);
    okenv_track_free(tr);
    return 0;
}
