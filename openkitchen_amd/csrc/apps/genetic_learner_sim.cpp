// genetic_learner_sim.cpp -- the EvolutionaryRacer application over the batched Environment (C ABI, include/okenv.h).
//
// Replaces the reference's EvolutionaryRacer/genetic_learner_sim.cpp: the same generation loop (:47-96 -- rollout until
// every agent has crashed, assignScores, saveBestAgentNetwork's regression check and checkpoint, colony average,
// chooseAndMateAgents, reset to the start line), with the per-agent work of the loop body (GeneticAgent::updateAction,
// Environment::step, scores, mating) done for the whole population on the GPU.  The reference runs 50 agents with the default
// 15-ray fan (kNumAgents, :18); population, fan and hidden width are command-line parameters here.  Window, score plot and the
// shared-memory queue of the original (VisUtils.hpp, spmc_queue.h) are not part of the path.
//
//   genetic_learner_sim track.csv [--agents N] [--rays R] [--hidden H] [--generations G] [--seed S] [--max-steps M]
//                                 [--steps-per-launch L] [--dump file] [--save-dir D] [--init-from D]
//                                 [--gpus K] [--devices d0,d1,...] [--gather rccl|host]
//
// Multi-GPU (BASELINE config 4): --gpus K runs K independent island populations, one host thread + one
// okenv handle + one HIP stream per GPU (island g: device g unless --devices says otherwise, seed S + g, global agent ids
// g*N ... g*N + N - 1).  Nothing is exchanged inside a rollout.  Once per generation every island's fitness vector
// (MiscUtils.hpp:64-71's scores, N floats) is all-gathered over RCCL -- ncclAllGather straight from the device buffer
// okenv_ga_scores filled, on the handle's own stream, no host hop -- for the colony statistics; selection and mating stay
// local to the island.  K = 1 is not a special case: a one-rank communicator is created and the same call runs.
// `--gather host` is a REHEARSAL for machines with one GPU (RCCL refuses two ranks on one device): the islands may then
// share a device (--devices 0,0) and the gather is staged through host memory; everything else is the same code.
//
// --dump writes, per generation, {int32 steps, float scores[N], int32 parents[5], float colony_best, float colony_mean} and
// finally the all-time best agent's weights as one padded block -- what the parity tests replay on the CPU oracle.  With
// K > 1 island g writes <file>.island<g>.
// --save-dir D writes the all-time best agent's network whenever it improves, as saveBestAgentNetwork does
// (MiscUtils.hpp:52-59): D/agent_weights_1.txt ((R+2) x H) and D/agent_weights_2.txt (H x 6) in writeMatrixToFile's text
// format (Network.hpp:29-51), loadable by the reference's readMatrixFromFile; islands g > 0 write
// agent_weights_{1,2}.island<g>.txt.  --init-from D starts EVERY agent from those two files, like kInitFromCheckpoint
// (Network.hpp:97,108-116; `.txt.safe` is tried as well, the name the reference reads).
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "okenv.h"
#include "ga_checkpoint.h" // writeMatrixToFile / readMatrixFromFile, the padded weight layout

namespace
{
using namespace ga_checkpoint;

struct Options
{
    std::string      track;
    int              agents{50}, rays{15}, hidden{30}, generations{5}, max_steps{4000}, steps_per_launch{50};
    uint32_t         seed{1234};
    std::string      dump, save_dir, init_from;
    int              gpus{1};
    std::vector<int> devices;
    bool             gather_host{false};
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
        else if (k == "--hidden") o.hidden = std::atoi(v);
        else if (k == "--generations") o.generations = std::atoi(v);
        else if (k == "--seed") o.seed = static_cast<uint32_t>(std::strtoul(v, nullptr, 10));
        else if (k == "--max-steps") o.max_steps = std::atoi(v);
        else if (k == "--steps-per-launch") o.steps_per_launch = std::atoi(v);
        else if (k == "--dump") o.dump = v;
        else if (k == "--save-dir") o.save_dir = v;
        else if (k == "--init-from") o.init_from = v;
        else if (k == "--gpus") o.gpus = std::atoi(v);
        else if (k == "--devices")
        {
            for (const char *p = v; *p;)
            {
                o.devices.push_back(static_cast<int>(std::strtol(p, const_cast<char **>(&p), 10)));
                if (*p == ',')
                    ++p;
            }
        }
        else if (k == "--gather")
        {
            if (std::strcmp(v, "host") != 0 && std::strcmp(v, "rccl") != 0)
                return false;
            o.gather_host = std::strcmp(v, "host") == 0;
        }
        else return false;
    }
    if (o.gpus < 1 || o.agents < 1)
        return false;
    if (o.devices.empty())
        for (int g = 0; g < o.gpus; ++g)
            o.devices.push_back(g);
    return static_cast<int>(o.devices.size()) == o.gpus;
}

// what the islands share
struct Colony
{
    Options                  opt;
    int                      P{0}, S{0};
    std::vector<float>       seg, cx, cy, heading, fan;
    std::vector<ncclComm_t>  comms;
    // --gather host: the staging matrix [K][N] and a generation barrier
    std::vector<float>       host_colony;
    std::mutex               m;
    std::condition_variable  cv;
    int                      arrived{0};
    uint64_t                 phase{0};
    std::mutex               print_m;
    std::vector<int>         rc; // per island exit code
};

void hostBarrier(Colony &c)
{
    std::unique_lock<std::mutex> lk(c.m);
    const uint64_t               my = c.phase;
    if (++c.arrived == c.opt.gpus)
    {
        c.arrived = 0;
        ++c.phase;
        c.cv.notify_all();
    }
    else
        c.cv.wait(lk, [&] { return c.phase != my; });
}

#define CHECK(call)                                                                                                    \
    do                                                                                                                 \
    {                                                                                                                  \
        const int rc_ = (call);                                                                                        \
        if (rc_ != OKENV_OK)                                                                                           \
        {                                                                                                              \
            std::fprintf(stderr, "island %d: %s failed (%d): %s\n", g, #call, rc_, okenv_last_error(env));             \
            std::quick_exit(2); /* the other islands would wait in the collective for ever */                          \
        }                                                                                                              \
    } while (0)
#define CHECK_HIP(call)                                                                                                \
    do                                                                                                                 \
    {                                                                                                                  \
        const hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess)                                                                                          \
        {                                                                                                              \
            std::fprintf(stderr, "island %d: %s failed: %s\n", g, #call, hipGetErrorString(e_));                       \
            std::quick_exit(2);                                                                                        \
        }                                                                                                              \
    } while (0)
#define CHECK_NCCL(call)                                                                                               \
    do                                                                                                                 \
    {                                                                                                                  \
        const ncclResult_t r_ = (call);                                                                                \
        if (r_ != ncclSuccess)                                                                                         \
        {                                                                                                              \
            std::fprintf(stderr, "island %d: %s failed: %s\n", g, #call, ncclGetErrorString(r_));                      \
            std::quick_exit(2);                                                                                        \
        }                                                                                                              \
    } while (0)

// One island: the reference's main() from the Environment's construction on, for global agents g*N ... g*N + N - 1.
void runIsland(Colony &c, const int g)
{
    const Options &opt    = c.opt;
    const int      N      = opt.agents, K = opt.gpus, device = opt.devices[g];
    const uint32_t seed   = opt.seed + static_cast<uint32_t>(g);
    const uint32_t base   = static_cast<uint32_t>(g) * static_cast<uint32_t>(N);
    okenv_t        env    = nullptr;
    CHECK(okenv_create(&env, c.seg.data(), c.S, N, opt.rays, c.fan.data(), device, OKENV_FLAG_NONE, 0.0F));
    CHECK(okenv_set_centerline(env, c.cx.data(), c.cy.data(), c.heading.data(), c.P));
    // GeneticAgent drives by acceleration (GeneticAgent.hpp:28,34)
    std::vector<uint8_t> mode(N, OKENV_MODE_ACCELERATION);
    CHECK(okenv_set_field(env, OKENV_F_MODE, mode.data()));
    CHECK(okenv_policy_mlp_create(env, opt.hidden, seed, base));
    const int per = okenv_policy_mlp_weights_per_agent(env);
    if (!opt.init_from.empty())
    { // kInitFromCheckpoint (Network.hpp:108-116): every agent's Network() reads the same two files
        std::vector<float> w1, w2;
        int                r1 = 0, c1 = 0, r2 = 0, c2 = 0;
        const bool         ok1 = readMatrixFromFile(opt.init_from + "/agent_weights_1.txt", w1, r1, c1) ||
                         readMatrixFromFile(opt.init_from + "/agent_weights_1.txt.safe", w1, r1, c1);
        const bool ok2 = readMatrixFromFile(opt.init_from + "/agent_weights_2.txt", w2, r2, c2) ||
                         readMatrixFromFile(opt.init_from + "/agent_weights_2.txt.safe", w2, r2, c2);
        if (!ok1 || !ok2 || r1 != opt.rays + 2 || c1 != opt.hidden || r2 != opt.hidden || c2 != OK_MLP_OUT)
        {
            std::fprintf(stderr, "--init-from %s: need agent_weights_1.txt (%d x %d) and agent_weights_2.txt (%d x %d); found %d x %d and %d x %d\n",
                         opt.init_from.c_str(), opt.rays + 2, opt.hidden, opt.hidden, OK_MLP_OUT, r1, c1, r2, c2);
            std::quick_exit(2);
        }
        std::vector<float> all(static_cast<size_t>(N) * per);
        padWeights(w1, w2, opt.rays, opt.hidden, all.data());
        for (int a = 1; a < N; ++a)
            std::copy(all.begin(), all.begin() + per, all.begin() + static_cast<size_t>(a) * per);
        CHECK(okenv_policy_mlp_set_weights(env, all.data()));
    }
    // the island's end of the collective: its scores where okenv_ga_scores leaves them, the gathered matrix [K][N] beside them
    const float *d_scores = nullptr;
    void        *stream_v = nullptr;
    CHECK(okenv_ga_scores_device(env, &d_scores));
    CHECK(okenv_get_stream(env, &stream_v));
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    CHECK_HIP(hipSetDevice(device));
    float *d_colony = nullptr;
    CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&d_colony), sizeof(float) * static_cast<size_t>(K) * N));
    std::vector<float> colony(static_cast<size_t>(K) * N);

    // start pose: centre-line point kStartingIdx (= 3) with the heading of point 0 (genetic_learner_sim.cpp:34-36)
    const float start_x = c.cx[3], start_y = c.cy[3], start_rot = c.heading[0];
    const std::string suffix = K > 1 && g > 0 ? ".island" + std::to_string(g) : "";
    std::FILE *dump = opt.dump.empty() ? nullptr : std::fopen((opt.dump + (K > 1 ? ".island" + std::to_string(g) : "")).c_str(), "wb");
    if (!opt.dump.empty() && dump == nullptr)
    {
        std::fprintf(stderr, "island %d: cannot write %s\n", g, opt.dump.c_str());
        std::quick_exit(2);
    }
    std::vector<float> colony_avg_scores;
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
        // assignScores (MiscUtils.hpp:64-71) into the handle's device buffer, then the generation's ONE exchange: every island's
        // N scores to every island, from device memory to device memory, in stream order behind the kernel that wrote them
        CHECK(okenv_ga_scores(env, nullptr));
        if (!opt.gather_host)
            CHECK_NCCL(ncclAllGather(d_scores, d_colony, static_cast<size_t>(N), ncclFloat, c.comms[g], stream));
        else
        { // rehearsal on one device: the same matrix, staged through host memory between two thread barriers
            CHECK_HIP(hipMemcpyAsync(c.host_colony.data() + static_cast<size_t>(g) * N, d_scores, sizeof(float) * N, hipMemcpyDeviceToHost, stream));
            CHECK_HIP(hipStreamSynchronize(stream));
            hostBarrier(c);
            CHECK_HIP(hipMemcpyAsync(d_colony, c.host_colony.data(), sizeof(float) * static_cast<size_t>(K) * N, hipMemcpyHostToDevice, stream));
            CHECK_HIP(hipStreamSynchronize(stream));
            hostBarrier(c); // nobody overwrites its row before everybody has read the matrix
        }
        CHECK_HIP(hipMemcpyAsync(colony.data(), d_colony, sizeof(float) * static_cast<size_t>(K) * N, hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        const float *scores = colony.data() + static_cast<size_t>(g) * N; // this island's row of the gathered matrix
        // saveBestAgentNetwork (MiscUtils.hpp:26-62): best of this generation, regression check, keep the all-time best
        size_t top_scorer_agent_id = 0;
        float  best_score_current  = 0.F;
        float  current_avg         = 0.F;
        for (size_t i = 0; i < static_cast<size_t>(N); ++i)
        {
            if (scores[i] > best_score_current)
            {
                best_score_current  = scores[i];
                top_scorer_agent_id = i;
            }
            current_avg += scores[i];
        }
        current_avg /= static_cast<float>(N);
        if (prev_gen_best_score > best_score_current)
        {
            std::fprintf(stderr, "!! Current gen. high score %f is less than previous gen. high score %f\n", best_score_current,
                         prev_gen_best_score);
            std::quick_exit(3); // the reference throws here
        }
        prev_gen_best_score = best_score_current;
        if (best_score_current > top_score_all_time)
        {
            std::vector<float> all(static_cast<size_t>(N) * per);
            CHECK(okenv_policy_mlp_get_weights(env, all.data()));
            best_weights.assign(all.begin() + static_cast<long>(top_scorer_agent_id) * per, all.begin() + static_cast<long>(top_scorer_agent_id + 1) * per);
            top_score_all_time = best_score_current;
            if (!opt.save_dir.empty())
            { // genetic::writeMatrixToFile("agent_weights_1.txt", weights_1_) / ("agent_weights_2.txt", weights_2_)
                std::vector<float> w1, w2;
                unpadWeights(best_weights.data(), opt.rays, opt.hidden, w1, w2);
                const std::string f1 = opt.save_dir + "/agent_weights_1" + suffix + ".txt", f2 = opt.save_dir + "/agent_weights_2" + suffix + ".txt";
                if (!writeMatrixToFile(f1, w1, opt.rays + 2, opt.hidden) || !writeMatrixToFile(f2, w2, opt.hidden, OK_MLP_OUT))
                    std::quick_exit(2);
            }
        }
        colony_avg_scores.push_back(current_avg);
        // the colony over all islands (showColonyScore's numbers for the whole node)
        float  colony_best = 0.F;
        double colony_sum  = 0.0;
        for (const float v : colony)
        {
            colony_best = std::max(colony_best, v);
            colony_sum += v;
        }
        const float colony_mean = static_cast<float>(colony_sum / static_cast<double>(colony.size()));
        {
            std::lock_guard<std::mutex> lk(c.print_m);
            if (K == 1)
            {
                std::printf("------------ EPISODE %d DONE ---------------\n", episode_idx);
                std::printf("top: %g  colony average: %g  steps: %d\n", best_score_current, current_avg, iteration);
            }
            else
                std::printf("------------ EPISODE %d DONE --------------- island %d (device %d): top: %g  island average: %g  steps: %d | all %d islands: top: "
                            "%g  average: %g\n", episode_idx, g, device, best_score_current, current_avg, iteration, K, colony_best, colony_mean);
            std::fflush(stdout);
        }
        // chooseAndMateAgents (Mating.hpp:108-166), local to the island
        int32_t parents[5] = {-1, -1, -1, -1, -1};
        CHECK(okenv_ga_select_mate(env, seed, static_cast<uint32_t>(episode_idx), base, parents));
        if (dump)
        {
            const int32_t steps = iteration;
            std::fwrite(&steps, 4, 1, dump);
            std::fwrite(scores, 4, static_cast<size_t>(N), dump);
            std::fwrite(parents, 4, 5, dump);
            std::fwrite(&colony_best, 4, 1, dump);
            std::fwrite(&colony_mean, 4, 1, dump);
        }
    }
    if (dump)
    {
        std::fwrite(best_weights.data(), 4, best_weights.size(), dump);
        std::fclose(dump);
    }
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipFree(d_colony));
    okenv_destroy(env);
    c.rc[g] = 0;
}
} // namespace

int main(int argc, char **argv)
{
    Colony c;
    if (!parse(argc, argv, c.opt))
    {
        std::fprintf(stderr, "Provide a file path for the track csv file [--agents N --rays R --hidden H --generations G --seed S "
                             "--max-steps M --steps-per-launch L --dump file --save-dir D --init-from D --gpus K --devices d0,d1,.. "
                             "--gather rccl|host]\n");
        return -1;
    }
    const Options &opt = c.opt;
    okenv_track_t  tr  = nullptr;
    if (okenv_track_load(&tr, opt.track.c_str()) != OKENV_OK)
    {
        std::fprintf(stderr, "cannot load %s\n", opt.track.c_str());
        return 2;
    }
    c.P = okenv_track_num_points(tr);
    c.S = okenv_track_num_segments(tr);
    c.seg.resize(4 * static_cast<size_t>(c.S));
    c.cx.resize(c.P);
    c.cy.resize(c.P);
    c.heading.resize(c.P);
    okenv_track_segments(tr, c.seg.data());
    okenv_track_get(tr, 0, c.cx.data());
    okenv_track_get(tr, 1, c.cy.data());
    okenv_track_get(tr, 4, c.heading.data());
    // sensor fan: -70 ... +70 degrees (Agent.cpp:11-18 builds it every 10 degrees = 15 rays)
    c.fan.resize(opt.rays);
    for (int i = 0; i < opt.rays; ++i)
        c.fan[i] = opt.rays == 1 ? 0.F : -70.0F + 140.0F * static_cast<float>(i) / static_cast<float>(opt.rays - 1);
    c.rc.assign(opt.gpus, 1);
    if (opt.gather_host)
        c.host_colony.assign(static_cast<size_t>(opt.gpus) * opt.agents, 0.F);
    else
    { // one communicator per island, all created here (single process, one thread per device afterwards); also for --gpus 1
        c.comms.resize(opt.gpus);
        const ncclResult_t r = ncclCommInitAll(c.comms.data(), opt.gpus, opt.devices.data());
        if (r != ncclSuccess)
        {
            std::fprintf(stderr, "ncclCommInitAll over %d device(s) failed: %s (two islands on one device need --gather host)\n", opt.gpus,
                         ncclGetErrorString(r));
            return 2;
        }
    }
    std::vector<std::thread> islands;
    for (int g = 1; g < opt.gpus; ++g)
        islands.emplace_back(runIsland, std::ref(c), g);
    runIsland(c, 0);
    for (auto &t : islands)
        t.join();
    for (auto &comm : c.comms)
        ncclCommDestroy(comm);
    okenv_track_free(tr);
    return *std::max_element(c.rc.begin(), c.rc.end());
}
