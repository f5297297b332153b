// Environment.cpp -- TrackSegments, CollisionChecker and Environment of include/Environment/, hosted on the C ABI
// of include/okenv.h.  Callers keep mutating their Agent objects between steps (reset(), current_action_), so every
// step gathers the agents' fields into struct-of-arrays staging buffers, uploads them, runs the fused device step
// and scatters the results back: the same two PCIe crossings per step the reference pays
// (reference Environment/CollisionChecker.cu:130,142), but a few hundred bytes per agent instead of 24 B per ray
// each way, and everything between them is one kernel.  Throughput-oriented callers use the C ABI directly and
// leave the state on the device.
#include "Environment/Environment.h"

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <random>

#include "okenv.h"
#include "okenv_math.h"

namespace
{
[[noreturn]] void die(const char *what, okenv_t h)
{
    std::cerr << "okenv: " << what << ": " << okenv_last_error(h) << std::endl;
    std::terminate(); // GOX_ASSERT semantics of the reference: the step path has no recoverable errors
}

std::mt19937 &rng()
{
    static std::mt19937 gen(20260220U);
    return gen;
}
} // namespace

// ---- TrackSegments ------------------------------------------------------------------------------------------

TrackSegments::TrackSegments(const RaceTrack &rt)
{
    auto run = [&](const std::vector<Vec2d> &poly) {
        for (size_t i = 0; i + 1 < poly.size(); ++i)
            segments_.push_back({poly[i].x, poly[i].y, poly[i + 1].x, poly[i + 1].y});
    };
    auto closer = [&](const std::vector<Vec2d> &poly) {
        segments_.push_back({poly.back().x, poly.back().y, poly.front().x, poly.front().y});
    };
    run(rt.left_bound_inner_);
    run(rt.left_bound_outer_);
    run(rt.right_bound_inner_);
    run(rt.right_bound_outer_);
    if (rt.left_bound_inner_.size() > 1)
    {
        closer(rt.left_bound_inner_);
        closer(rt.right_bound_inner_);
    }
    if (rt.left_bound_outer_.size() > 1)
    {
        closer(rt.left_bound_outer_);
        closer(rt.right_bound_outer_);
    }
    GOX_ASSERT(!segments_.empty());
    void *d = nullptr;
    if (hipMalloc(&d, segments_.size() * sizeof(Segment2d)) == hipSuccess &&
        hipMemcpy(d, segments_.data(), segments_.size() * sizeof(Segment2d), hipMemcpyHostToDevice) == hipSuccess)
        d_segments_ = static_cast<Segment2d *>(d);
    else
        die("TrackSegments: device upload failed (no usable GPU?)", nullptr);
}

TrackSegments::~TrackSegments()
{
    if (d_segments_)
        (void)hipFree(d_segments_);
}

// ---- CollisionChecker ---------------------------------------------------------------------------------------

class CollisionChecker::Impl
{
  public:
    Impl(const Segment2d *segments, const size_t num_segments, const std::vector<Agent *> &agents) : agents_(agents)
    {
        GOX_ASSERT(!agents.empty());
        n_ = static_cast<int>(agents.size());
        r_ = static_cast<int>(agents[0]->sensor_ray_angles_.size()); // all agents share one fan (CollisionChecker.cu:82)
        GOX_ASSERT(r_ > 0);
        // the segments may live on the device (TrackSegments::getDeviceSegments) or on the host
        std::vector<Segment2d> host(num_segments);
        if (hipMemcpy(host.data(), segments, num_segments * sizeof(Segment2d), hipMemcpyDefault) != hipSuccess)
            die("CollisionChecker: cannot read the segment array", nullptr);
        int device = 0;
        (void)hipGetDevice(&device);
        if (okenv_create(&h_, reinterpret_cast<const float *>(host.data()), static_cast<int32_t>(num_segments), n_, r_,
                         agents[0]->sensor_ray_angles_.data(), device, OKENV_FLAG_NONE, 0.F) != OKENV_OK)
            die("okenv_create", nullptr);
        recs_.assign(static_cast<size_t>(n_), okenv_agent_record{});
        hits_.assign(static_cast<size_t>(n_) * r_ * 2U, 0.F);
        rays_.assign(static_cast<size_t>(n_) * r_, Ray_{0.F, 0.F, 0.F, 0.F, 0.F, true});
        fan_ = agents[0]->sensor_ray_angles_;
        // Step trace for record / replay checks: OKENV_TRACE_FILE=<path> makes every exchange append what went in and what
        // came out (tests/test_reference_binding_dropin.py replays it on the CPU oracle); OKENV_TRACE_STEPS bounds the file.
        if (const char *path = std::getenv("OKENV_TRACE_FILE"))
        {
            trace_ = std::fopen(path, "ab");
            if (const char *lim = std::getenv("OKENV_TRACE_STEPS"))
                trace_left_ = std::atol(lim);
        }
    }

    ~Impl()
    {
        if (trace_)
            std::fclose(trace_);
        okenv_destroy(h_);
    }

    // One trace record: {magic, n, r, flags} {sensor_offset} fan[r] in[n] out[n] hits[n * r * 2]
    void trace(const std::vector<okenv_agent_record> &in, const uint32_t flags)
    {
        if (!trace_ || trace_left_ <= 0)
            return;
        --trace_left_;
        const uint32_t head[4] = {0x4F4B5452U /* "OKTR" */, static_cast<uint32_t>(n_), static_cast<uint32_t>(r_), flags};
        const float    off     = agents_[0]->sensor_offset_;
        std::fwrite(head, 4, 4, trace_);
        std::fwrite(&off, 4, 1, trace_);
        std::fwrite(fan_.data(), 4, fan_.size(), trace_);
        std::fwrite(in.data(), sizeof(okenv_agent_record), in.size(), trace_);
        std::fwrite(recs_.data(), sizeof(okenv_agent_record), recs_.size(), trace_);
        std::fwrite(hits_.data(), 4, hits_.size(), trace_);
        std::fflush(trace_);
    }

    // Agent objects -> packed records -> device, one step (or only the collision pass), and back: two PCIe copies
    // per call (okenv_step_packed).  The four `disp_*` arrays (optional) carry Environment's DisplacementStats.
    void exchange(const std::vector<Agent *> &agents, const bool collide_only, uint32_t *disp_ctr, float *disp_x, float *disp_y,
                  uint8_t *disp_to)
    {
        const bool with_stats = disp_ctr != nullptr;
        for (int i = 0; i < n_; ++i)
        {
            const Agent        *a = agents[i];
            okenv_agent_record &r = recs_[static_cast<size_t>(i)];
            r.pos_x               = a->pos_.x;
            r.pos_y               = a->pos_.y;
            r.rot                 = a->rot_;
            r.speed               = a->speed_;
            r.acc                 = a->acceleration_;
            r.throttle            = a->current_action_.throttle_delta;
            r.steer               = a->current_action_.steering_delta;
            r.mode                = static_cast<uint8_t>(a->movement_mode_);
            r.crashed             = a->crashed_ ? 1 : 0;
            r.timed_out           = a->timed_out_ ? 1 : 0;
            if (with_stats)
            {
                r.disp_ctr       = disp_ctr[i];
                r.disp_x         = disp_x[i];
                r.disp_y         = disp_y[i];
                r.disp_timed_out = disp_to[i];
            }
        }
        // The reference reads sensor_offset_ and sensor_ray_angles_ per agent on every call (CollisionChecker.cu:113-128); the
        // device keeps ONE fan and ONE offset for the population, so anything else must not pass silently.
        for (int i = 0; i < n_; ++i)
        {
            if (agents[i]->sensor_offset_ != agents[0]->sensor_offset_ || agents[i]->sensor_ray_angles_ != fan_)
            {
                std::cerr << "okenv: agent " << i << " has its own sensor_offset_ / sensor_ray_angles_ (or the fan changed after the "
                          << "CollisionChecker was built): one fan and one offset per Environment are supported" << std::endl;
                std::terminate();
            }
        }
        if (okenv_set_sensor_offset(h_, agents[0]->sensor_offset_) != OKENV_OK)
            die("okenv_set_sensor_offset", h_);
        const uint32_t flags = (with_stats ? OKENV_PACKED_WITH_STATS : 0U) | (collide_only ? OKENV_PACKED_COLLIDE_ONLY : 0U);
        if (trace_ && trace_left_ > 0)
            trace_in_ = recs_;
        if (okenv_step_packed(h_, recs_.data(), recs_.data(), hits_.data(), flags) != OKENV_OK)
            die("okenv_step_packed", h_);
        trace(trace_in_, flags);
        for (int i = 0; i < n_; ++i)
        {
            Agent                    *a = agents[i];
            const okenv_agent_record &r = recs_[static_cast<size_t>(i)];
            a->pos_                     = {r.pos_x, r.pos_y};
            a->rot_                     = r.rot;
            a->speed_                   = r.speed;
            a->acceleration_            = r.acc;
            a->crashed_                 = r.crashed != 0;
            a->timed_out_               = r.timed_out != 0;
            a->sensor_hits_.resize(static_cast<size_t>(r_));
            for (int q = 0; q < r_; ++q)
            {
                const size_t k     = (static_cast<size_t>(i) * r_ + q) * 2U;
                a->sensor_hits_[q] = {hits_[k], hits_[k + 1]};
            }
            if (with_stats)
            {
                disp_ctr[i] = r.disp_ctr;
                disp_x[i]   = r.disp_x;
                disp_y[i]   = r.disp_y;
                disp_to[i]  = r.disp_timed_out;
            }
        }
        rays_valid_ = false;
    }

    // Ray_ view for visualisation, rebuilt on demand from the device state (reference CollisionChecker.cu:115-128)
    const Ray_ *hostRays()
    {
        if (!rays_valid_)
        {
            const size_t       nr = static_cast<size_t>(n_) * r_;
            std::vector<float> hx(nr), hy(nr);
            if (okenv_get_field(h_, OKENV_F_HIT_X, hx.data()) != OKENV_OK || okenv_get_field(h_, OKENV_F_HIT_Y, hy.data()) != OKENV_OK)
                die("okenv_get_field", h_);
            for (int i = 0; i < n_; ++i)
            {
                const Agent *a = agents_[i];
                float        sn, cs;
                ok_sincosf(OK_DEG2RAD * a->rot_, &sn, &cs);
                for (int r = 0; r < r_; ++r)
                {
                    Ray_ &ray  = rays_[static_cast<size_t>(i) * r_ + r];
                    ray.x      = a->pos_.x + a->sensor_offset_ * cs;
                    ray.y      = a->pos_.y + a->sensor_offset_ * sn;
                    ray.angle  = OK_DEG2RAD * (a->rot_ + agents_[0]->sensor_ray_angles_[r]);
                    ray.hit_x  = hx[static_cast<size_t>(i) * r_ + r];
                    ray.hit_y  = hy[static_cast<size_t>(i) * r_ + r];
                    ray.active = !a->crashed_;
                }
            }
            rays_valid_ = true;
        }
        return rays_.data();
    }

    okenv_t              h_{nullptr};
    std::vector<Agent *> agents_;
    int                  n_{0}, r_{0};
    std::vector<okenv_agent_record> recs_, trace_in_;
    std::vector<float>              hits_, fan_;
    std::FILE                      *trace_{nullptr};
    long                            trace_left_{100000};
    std::vector<Ray_>               rays_;
    bool                  rays_valid_{false};
};

CollisionChecker::CollisionChecker(const Segment2d *d_segments, size_t num_segments, const std::vector<Agent *> &agents)
    : impl_(std::make_unique<Impl>(d_segments, num_segments, agents))
{
}

CollisionChecker::~CollisionChecker() = default;

void CollisionChecker::checkCollision()
{
    impl_->exchange(impl_->agents_, true, nullptr, nullptr, nullptr, nullptr);
}

const Ray_ *CollisionChecker::getHostRays() const
{
    return impl_->hostRays();
}

size_t CollisionChecker::getNumRays() const
{
    return static_cast<size_t>(impl_->n_) * impl_->r_;
}

okenv *CollisionChecker::handle() const
{
    return impl_->h_;
}

void CollisionChecker::stepAgents(const std::vector<Agent *> &agents, uint32_t *disp_ctr, float *disp_x, float *disp_y, uint8_t *disp_timed_out)
{
    impl_->exchange(agents, false, disp_ctr, disp_x, disp_y, disp_timed_out);
}

// ---- Environment --------------------------------------------------------------------------------------------

Environment::Environment(const std::string &race_track_path, const std::vector<Agent *> &agents, const bool draw_rays, const bool hidden_window)
    : draw_rays_(draw_rays)
{
    race_track_     = std::make_unique<RaceTrack>(race_track_path);
    track_segments_ = std::make_unique<TrackSegments>(*race_track_);
    visualizer_     = std::make_unique<env::Visualizer>(hidden_window);
    screen_grabber_ = std::make_unique<ScreenGrabber>(kScreenWidth, kScreenHeight);
    // the frame is painted when somebody grabs it (Visualizer.cpp), from the state as it is then
    screen_grabber_->setPainter([this](std::vector<uint8_t> &rgba, const int w, const int h) {
        env::paintFrame(rgba, w, h, *race_track_, agents_, (draw_rays_ && collision_checker_) ? collision_checker_.get() : nullptr);
    });
    agents_         = agents;
    displacement_stats_.resize(agents.size());
    if (!agents_.empty())
        ensureChecker();
}

Environment::Environment(const std::string &race_track_path) : Environment(race_track_path, std::vector<Agent *>{}, true, true) {}

Environment::~Environment() = default;

void Environment::setAgent(Agent *agent)
{
    agents_.push_back(agent);
    displacement_stats_.resize(agents_.size());
    checker_stale_ = true; // population (and possibly the ray count) changed: rebuild at the next step
}

void Environment::ensureChecker()
{
    if (!checker_stale_ && collision_checker_)
        return;
    GOX_ASSERT(!agents_.empty());
    collision_checker_ = std::make_unique<CollisionChecker>(track_segments_->getDeviceSegments(), track_segments_->getNumSegments(), agents_);
    const auto &d      = race_track_->track_data_points_;
    if (okenv_set_centerline(collision_checker_->handle(), d.x_m.data(), d.y_m.data(), race_track_->headings_.data(),
                             static_cast<int32_t>(d.x_m.size())) != OKENV_OK)
        die("okenv_set_centerline", collision_checker_->handle());
    checker_stale_ = false;
}

void Environment::drawSensorRanges(const std::vector<Vec2d> & /*sensor_hits*/) {}

void Environment::step()
{
    ensureChecker();
    // DisplacementStats travel with the agents: members -> flat arrays -> device -> back
    const size_t          n = agents_.size();
    std::vector<uint32_t> ctr(n);
    std::vector<float>    ix(n), iy(n);
    std::vector<uint8_t>  to(n);
    for (size_t i = 0; i < n; ++i)
    {
        ctr[i] = displacement_stats_[i].displacement_ctr;
        ix[i]  = displacement_stats_[i].init_pos.x;
        iy[i]  = displacement_stats_[i].init_pos.y;
        to[i]  = displacement_stats_[i].displacement_timed_out ? 1 : 0;
    }
    collision_checker_->stepAgents(agents_, ctr.data(), ix.data(), iy.data(), to.data());
    for (size_t i = 0; i < n; ++i)
    {
        displacement_stats_[i].displacement_ctr       = ctr[i];
        displacement_stats_[i].init_pos               = {ix[i], iy[i]};
        displacement_stats_[i].displacement_timed_out = to[i] != 0;
    }
    visualizer_->render(*race_track_, agents_, draw_rays_ ? collision_checker_.get() : nullptr);
}

void Environment::seedRandom(const uint32_t seed)
{
    rng().seed(seed);
}

int Environment::randomValue(const int lo, const int hi)
{
    return std::uniform_int_distribution<int>(lo, hi)(rng());
}

// headless stand-in for the drawing library's inclusive integer draw (Visualizer.h)
int GetRandomValue(const int min, const int max)
{
    return std::uniform_int_distribution<int>(min, max)(rng());
}

int32_t Environment::pickRandomResetTrackIdx() const
{
    return randomValue(0, static_cast<int32_t>(race_track_->track_data_points_.x_m.size()) - 1);
}

void Environment::resetAgent(Agent *agent, const bool pick_random_point, const bool randomize_lane, const bool randomize_heading)
{
    const int32_t idx = pick_random_point ? pickRandomResetTrackIdx() : static_cast<int32_t>(RaceTrack::kStartingIdx);

    // heading offset of +-(45 + [0,45]) degrees, the sign alternating from call to call
    // (reference Environment/Environment.cpp:86-101)
    float         heading_offset = 0.F;
    static size_t flip           = 0;
    if (pick_random_point && randomize_heading)
    {
        constexpr float kRange{45.F};
        const float     magnitude = static_cast<float>(randomValue(0, static_cast<int>(kRange))) + kRange;
        heading_offset            = (flip % 2 == 0) ? magnitude * -1.F : magnitude;
        ++flip;
    }

    const auto &d = race_track_->track_data_points_;
    float       x = d.x_m[idx], y = d.y_m[idx];
    if (pick_random_point && randomize_lane)
    {
        // lateral position between the inner boundaries, alpha in [0.10, 0.90] (reference :106-113)
        const Vec2d l     = race_track_->left_bound_inner_[idx];
        const Vec2d r     = race_track_->right_bound_inner_[idx];
        const float alpha = static_cast<float>(randomValue(10, 90)) / 100.F;
        x                 = l.x * alpha + r.x * (1.F - alpha);
        y                 = l.y * alpha + r.y * (1.F - alpha);
    }
    agent->reset({x, y}, race_track_->headings_[idx] + heading_offset); // virtual: the derived reset runs
}

bool Environment::isEnterPressed() const
{
    return false; // no terminal polling in the headless build
}

void Environment::saveImage(const std::string &filename) const
{
    screen_grabber_->saveRenderTargetToFile(filename);
}
