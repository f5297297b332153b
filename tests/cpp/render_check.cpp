// render_check.cpp -- grabs the headless render target of the C++ drop-in Environment after a few steps and saves it; prints
// what tests/test_gpu_render.py needs to know (agent poses, one ray's end points).
#include <cstdio>
#include <memory>
#include <vector>

#include "Environment/Environment.h"

class Plain : public Agent
{
  public:
    Plain(const Vec2d p, const float rot, const int16_t id) : Agent(p, rot, id) {}
    void updateAction() override { current_action_ = {20.F, 0.F}; }
};

int main(int argc, char **argv)
{
    if (argc != 3)
        return 2;
    std::vector<std::unique_ptr<Plain>> agents;
    for (int16_t i = 0; i < 2; ++i)
        agents.push_back(std::make_unique<Plain>(Vec2d{0, 0}, 0, i));
    agents[1]->color_ = BLUE;
    Environment env(argv[1], createBaseAgentPtrs(agents), /*draw_rays=*/true, /*hidden_window=*/true);
    const auto &d = env.race_track_->track_data_points_;
    agents[0]->reset({d.x_m[3], d.y_m[3]}, env.race_track_->headings_[3]);
    agents[1]->reset({d.x_m[400], d.y_m[400]}, env.race_track_->headings_[400]);
    for (int s = 0; s < 3; ++s)
    {
        for (auto &a : agents)
            a->updateAction();
        env.step();
    }
    env.saveImage(argv[2]);
    const auto  info = env.getRenderTargetInfo();
    const Ray_ *rays = env.collision_checker_->getHostRays();
    std::printf("%d %d %d\n", info.width, info.height, info.channels);
    for (auto &a : agents)
        std::printf("%f %f %f\n", a->pos_.x, a->pos_.y, a->rot_);
    std::printf("%f %f %f %f\n", rays[7].x, rays[7].y, rays[7].hit_x, rays[7].hit_y);
    std::printf("%f %f\n", d.x_m[200], d.y_m[200]);
    return 0;
}
