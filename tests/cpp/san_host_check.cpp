// san_host_check.cpp -- the host-only parts of the facade under AddressSanitizer / UndefinedBehaviorSanitizer
// (tests/test_sanitizers.py compiles this file TOGETHER with facade/Visualizer.cpp, Agent.cpp and RaceTrack.cpp, so the PNG
// writer with its deflate, the software rasteriser and the track builder are instrumented; nothing here needs a GPU):
//   san_host_check <track.csv> <out.png> <out2.png>
// 1. RaceTrack from the CSV, its three queries on probe points inside and far outside the track;
// 2. the frame Visualizer::render would draw (track bands, agents also half outside the window and far outside it), painted by
//    env::paintFrame and written as a PNG;
// 3. ScreenGrabber's PNG writer on a synthetic frame with runs, gradients and single pixels (png_check.cpp's frame) at an odd size.
#include <cstdio>
#include <memory>
#include <vector>

#include "Environment/Agent.h"
#include "Environment/CollisionChecker.h"
#include "Environment/RaceTrack.h"
#include "Environment/ScreenGrabber.h"
#include "Environment/Visualizer.h"

// the frame is painted without rays here (no collision checker exists without a GPU); the rasteriser only asks these two
const Ray_ *CollisionChecker::getHostRays() const { return nullptr; }
size_t      CollisionChecker::getNumRays() const { return 0; }

class Plain : public Agent
{
  public:
    Plain(const Vec2d p, const float rot, const int16_t id) : Agent(p, rot, id) {}
    void updateAction() override {}
};

int main(int argc, char **argv)
{
    if (argc != 4)
        return 2;
    RaceTrack track(argv[1]);
    const auto &d = track.track_data_points_;
    if (d.x_m.size() < 100)
        return 3;
    float acc = 0.F;
    for (const Vec2d q : {Vec2d{d.x_m[3], d.y_m[3]}, Vec2d{d.x_m[50] + 7.F, d.y_m[50] - 4.F}, Vec2d{-5000.F, 9000.F}, Vec2d{1e9F, -1e9F}, Vec2d{0.F, 0.F}})
    {
        acc += static_cast<float>(track.findNearestTrackIndexBruteForce(q));
        acc += track.getNearestDistanceToTrackBoundary(q) * 1e-9F;
        acc += track.getDistanceToLaneCenter(q) * 1e-9F;
    }
    std::vector<std::unique_ptr<Plain>> agents;
    const Vec2d                         where[] = {{d.x_m[3], d.y_m[3]}, {d.x_m[400], d.y_m[400]}, {-3.F, 700.F}, {1598.F, 2.F}, {1603.F, 1405.F}, {-4000.F, 1e7F}, {800.F, 1399.5F}};
    int16_t                             id      = 0;
    for (const Vec2d p : where)
    {
        agents.push_back(std::make_unique<Plain>(p, 37.F * static_cast<float>(id), id));
        ++id;
    }
    agents[1]->color_ = BLUE;
    std::vector<Agent *> ptrs;
    for (auto &a : agents)
        ptrs.push_back(a.get());
    {
        ScreenGrabber g(kScreenWidth, kScreenHeight);
        g.setPainter([&](std::vector<uint8_t> &p, int w, int h) { env::paintFrame(p, w, h, track, ptrs, nullptr); });
        g.saveRenderTargetToFile(argv[2]);
    }
    {
        ScreenGrabber g(611, 397); // odd sizes: rows that do not end on the deflate's block or match boundaries
        g.setPainter([](std::vector<uint8_t> &p, int w, int h) {
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x)
                {
                    uint8_t *q = &p[(static_cast<size_t>(y) * w + x) * 4];
                    q[0]       = static_cast<uint8_t>((x / 100) * 16);
                    q[1]       = static_cast<uint8_t>((y > 200) ? 255 : (x * 7 + y * 3) % 251);
                    q[2]       = (x == y) ? 255 : 0;
                    q[3]       = 255;
                }
        });
        g.saveRenderTargetToFile(argv[3]);
    }
    std::printf("ok %g\n", static_cast<double>(acc));
    return 0;
}
