// RaceTrack.cpp -- see include/Environment/RaceTrack.h.
//
// Every fp32 operation below follows the order of the reference's construction chain so that the
// resulting floats are bit-identical (reference Environment/RaceTrack.cpp: parse+clamp :127-164,
// extents :166-196, scale+centre :198-229, gradient :87-114, lanes+headings :257-307, queries :16-72).
#include "Environment/RaceTrack.h"

#include <algorithm>
#include <cctype>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>

namespace
{
struct Box
{
    float lo_x, lo_y, hi_x, hi_y;
};

// x first, then y, strict comparisons from +-FLT_MAX (reference :166-196)
Box boundsOf(const std::vector<float> &xs, const std::vector<float> &ys)
{
    constexpr float kBig = std::numeric_limits<float>::max();
    Box             b{kBig, kBig, -kBig, -kBig};
    for (const float v : xs)
    {
        if (v < b.lo_x)
            b.lo_x = v;
        if (v > b.hi_x)
            b.hi_x = v;
    }
    for (const float v : ys)
    {
        if (v < b.lo_y)
            b.lo_y = v;
        if (v > b.hi_y)
            b.hi_y = v;
    }
    return b;
}

// one-sided at the ends, central (divided by 2.0f) inside (reference :87-114)
std::vector<float> finiteDifference(const std::vector<float> &v)
{
    const size_t       n = v.size();
    std::vector<float> d(n);
    for (size_t i = 0; i < n; ++i)
    {
        if (i == 0)
            d[i] = v[1] - v[0];
        else if (i + 1 == n)
            d[i] = v[i] - v[i - 1];
        else
            d[i] = (v[i + 1] - v[i - 1]) / 2.0F;
    }
    return d;
}

// metres: at least 4, times 3, at most 17 -- applied BEFORE the pixel scaling (reference :138-160)
float limitedWidth(const float raw)
{
    constexpr float kScale{3.0F}, kMax{17.0F}, kMin{4.0F};
    return std::min(std::max(kMin, raw) * kScale, kMax);
}
} // namespace

RaceTrack::RaceTrack(const std::string &track_csv_path)
{
    // upper-cased file stem (reference :116-125)
    const size_t slash = track_csv_path.rfind('/');
    const size_t dot   = track_csv_path.rfind('.');
    track_name_        = track_csv_path.substr(slash + 1, dot - slash - 1);
    for (char &ch : track_name_)
        ch = static_cast<char>(std::toupper(static_cast<unsigned char>(ch)));

    loaded_ok_ = readCsv(track_csv_path);
    if (!loaded_ok_ || track_data_points_.x_m.size() < 2)
    {
        loaded_ok_ = false;
        return;
    }
    fitToWindow(static_cast<float>(kScreenWidth), static_cast<float>(kScreenHeight));
    buildLanes();
    start_line_  = {right_bound_outer_.front(), right_bound_outer_.back()};
    finish_line_ = {left_bound_outer_.front(), left_bound_outer_.back()};
}

bool RaceTrack::readCsv(const std::string &path)
{
    std::ifstream in(path);
    if (!in.is_open())
    {
        std::cerr << "Error opening file: " << path << std::endl;
        return false;
    }
    std::string row;
    std::getline(in, row); // header
    while (std::getline(in, row))
    {
        if (row.empty() || row == "\r")
            continue;
        std::istringstream cols(row);
        std::string        cell;
        float              v[4];
        for (float &f : v)
        {
            if (!std::getline(cols, cell, ','))
                return false;
            f = std::stof(cell);
        }
        track_data_points_.x_m.push_back(v[0]);
        track_data_points_.y_m.push_back(v[1]);
        track_data_points_.w_tr_right_m.push_back(limitedWidth(v[2]));
        track_data_points_.w_tr_left_m.push_back(limitedWidth(v[3]));
    }
    return true;
}

void RaceTrack::fitToWindow(const float window_width, const float window_height)
{
    auto     &d   = track_data_points_;
    const Box raw = boundsOf(d.x_m, d.y_m);
    const float span_x = raw.hi_x - raw.lo_x;
    const float span_y = raw.hi_y - raw.lo_y;
    float       scale  = std::min(window_width / span_x, window_height / span_y);
    constexpr float kScreenFitScale{0.9F}; // leave a border
    scale *= kScreenFitScale;
    for (size_t i = 0; i < d.x_m.size(); ++i)
    {
        d.x_m[i] *= scale;
        d.y_m[i] *= scale;
        d.w_tr_left_m[i] *= scale;
        d.w_tr_right_m[i] *= scale;
    }
    const Box   scaled  = boundsOf(d.x_m, d.y_m);
    const float shift_x = (window_width / 2.F) - ((scaled.hi_x + scaled.lo_x) / 2.F);
    const float shift_y = (window_height / 2.F) - ((scaled.hi_y + scaled.lo_y) / 2.F);
    for (size_t i = 0; i < d.x_m.size(); ++i)
    {
        d.x_m[i] += shift_x;
        d.y_m[i] += shift_y;
    }
}

void RaceTrack::buildLanes()
{
    const auto        &d  = track_data_points_;
    const size_t       n  = d.x_m.size();
    std::vector<float> tx = finiteDifference(d.x_m);
    std::vector<float> ty = finiteDifference(d.y_m);
    headings_.reserve(n);
    for (size_t i = 0; i < n; ++i)
    {
        const float len = std::sqrt(tx[i] * tx[i] + ty[i] * ty[i]);
        tx[i] /= len;
        ty[i] /= len;
        // float atan2, float multiply, then a DOUBLE divide by M_PI rounded back to float (reference :278)
        headings_.push_back(std::atan2(ty[i], tx[i]) * 180.0F / M_PI);
    }
    left_bound_inner_.resize(n);
    left_bound_outer_.resize(n);
    right_bound_inner_.resize(n);
    right_bound_outer_.resize(n);
    constexpr float kBoundaryThickness{3.F};
    for (size_t i = 0; i < n; ++i)
    {
        const float cx = d.x_m[i], cy = d.y_m[i];
        const float wr = d.w_tr_right_m[i], wl = d.w_tr_left_m[i];
        // the normal of the unit tangent (tx,ty): right = (+ty,-tx), left = (-ty,+tx)
        right_bound_inner_[i] = {cx + wr * ty[i], cy - wr * tx[i]};
        left_bound_inner_[i]  = {cx - wl * ty[i], cy + wl * tx[i]};
        right_bound_outer_[i] = {cx + (wr + kBoundaryThickness) * ty[i], cy - (wr + kBoundaryThickness) * tx[i]};
        left_bound_outer_[i]  = {cx - (wl + kBoundaryThickness) * ty[i], cy + (wl + kBoundaryThickness) * tx[i]};
    }
}

size_t RaceTrack::findNearestTrackIndexBruteForce(const Vec2d &query_pt) const
{
    const auto &d    = track_data_points_;
    float       best = std::numeric_limits<float>::max();
    size_t      arg  = 0;
    for (size_t i = 0; i < d.x_m.size(); ++i)
    {
        const float dist2 = query_pt.distanceSquared({d.x_m[i], d.y_m[i]});
        if (dist2 < best)
        {
            best = dist2;
            arg  = i;
        }
    }
    return arg;
}

float RaceTrack::getNearestDistanceToTrackBoundary(const Vec2d &query_pt) const
{
    float best = std::numeric_limits<float>::max();
    for (size_t i = 0; i < left_bound_inner_.size(); ++i)
    {
        best = std::min(best, query_pt.distanceSquared(left_bound_inner_[i]));
        best = std::min(best, query_pt.distanceSquared(right_bound_inner_[i]));
    }
    return std::sqrt(best);
}

float RaceTrack::getDistanceToLaneCenter(const Vec2d &query_pt) const
{
    const auto  &d   = track_data_points_;
    const size_t arg = findNearestTrackIndexBruteForce(query_pt);
    const float  d2  = query_pt.distanceSquared({d.x_m[arg], d.y_m[arg]});
    return std::sqrt(d2) / (d.w_tr_left_m[arg] + d.w_tr_right_m[arg]);
}
