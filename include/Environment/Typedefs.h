// Typedefs.h -- value types shared by the Environment facade.
//
// Source-compatible with the reference's Environment/Typedefs.h (names, members and layouts callers and
// the device code rely on: Vec2d, Pixel, Extent2d, Ray_ = 24 B, Segment2d = 16 B, kDeg2Rad, screen size,
// GOX_ASSERT).  Written for this project; nothing here touches a GPU.
#pragma once

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
