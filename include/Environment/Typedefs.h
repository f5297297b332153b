// Typedefs.h -- value types shared by the Environment facade.
//
// Source-compatible with the reference's Environment/Typedefs.h (names, members and layouts callers and
// the device code rely on: Vec2d, Pixel, Extent2d, Ray_ = 24 B, Segment2d = 16 B, kDeg2Rad, screen size,
// GOX_ASSERT).  Written for this project; nothing here touches a GPU.
#pragma once

#include <cstddef>

#include <cmath>
#include <exception>
#include <math.h>
#include <vector>

// the library is built with -fvisibility=hidden; the drop-in classes are its C++ surface
#define OKENV_CLASS __attribute__((visibility("default")))

constexpr int kScreenWidth  = 1600; // reference Environment/Typedefs.h:7
constexpr int kScreenHeight = 1400; // reference Environment/Typedefs.h:8

// float(M_PI / 180.0F): bit pattern 0x3C8EFA35, identical to OK_DEG2RAD in include/okenv_math.h
constexpr float kDeg2Rad{static_cast<float>(M_PI / 180.0F)};

constexpr int kLeftBarrierColor[4]{255, 0, 0, 255};
constexpr int kRightBarrierColor[4]{0, 0, 255, 255};

#define GOX_ASSERT(cond)                                                                                               \
    do                                                                                                                 \
    {                                                                                                                  \
        if (!(cond))                                                                                                   \
            std::terminate();                                                                                          \
    } while (0)

struct Pixel
{
    int x{};
    int y{};
};

struct Vec2d
{
    float x{};
    float y{};

    float squaredNorm() const { return x * x + y * y; }
    float norm() const { return std::sqrt(squaredNorm()); }
    float length() const { return norm(); }
    float distanceSquared(const Vec2d &o) const { return (x - o.x) * (x - o.x) + (y - o.y) * (y - o.y); }

    Vec2d operator+(const Vec2d &o) const { return {x + o.x, y + o.y}; }
    Vec2d operator-(const Vec2d &o) const { return {x - o.x, y - o.y}; }
    Vec2d operator*(const float k) const { return {x * k, y * k}; }
    Vec2d operator/(const Vec2d &o) const { return {x / o.x, y / o.y}; }
    Vec2d operator/(const float k) const { return {x / k, y / k}; }
};

// ---- the drawing library's vocabulary the callers use, headless (there is no window on this path) ----------------
// The applications written against the legacy Environment header (AutoEncoder/collect_data_racetrack/Environment.hpp:
// RLRacers/Q_Learning/q_racer_sim.cpp, QAgent.hpp) name positions `raylib::Vector2` and colours by raylib's constants.
namespace raylib
{
using Vector2 = ::Vec2d;
}
struct Color
{
    unsigned char r, g, b, a;
};
constexpr Color WHITE{255, 255, 255, 255}, BLACK{0, 0, 0, 255}, RED{230, 41, 55, 255}, GREEN{0, 228, 48, 255},
    BLUE{0, 121, 241, 255}, YELLOW{253, 249, 0, 255};

// Agent::color_: `int color_[4]` in the current reference header (Environment/Agent.h:63), a raylib Color in the legacy
// one (`greedy_agent->color_ = BLUE`, q_racer_sim.cpp:109).  Indexable like the former, assignable from the latter.
struct AgentColor
{
    int rgba[4]{80, 80, 80, 255}; // dark gray
    int       &operator[](const size_t i) { return rgba[i]; }
    const int &operator[](const size_t i) const { return rgba[i]; }
    AgentColor &operator=(const Color &c)
    {
        rgba[0] = c.r, rgba[1] = c.g, rgba[2] = c.b, rgba[3] = c.a;
        return *this;
    }
};

struct Extent2d
{
    float min_x;
    float min_y;
    float max_x;
    float max_y;

    bool isPointInside(const Vec2d &p) const { return p.x > min_x && p.y > min_y && p.x < max_x && p.y < max_y; }
};

// One lidar ray as the reference's CollisionChecker exposes it through getHostRays()
// (reference Environment/Typedefs.h:91-99): 24 bytes.
struct Ray_
{
    float x; // origin
    float y;
    float angle; // world angle [rad]
    float hit_x;
    float hit_y;
    bool  active{true};
};
static_assert(sizeof(Ray_) == 24, "Ray_ layout must match the reference");

// Track boundary segment, 16 bytes (reference Environment/Typedefs.h:101-105).
struct Segment2d
{
    float x1, y1;
    float x2, y2;
};
static_assert(sizeof(Segment2d) == 16, "Segment2d layout must match the reference");
