// ScreenGrabber.h -- the render target without a window (reference Environment/ScreenGrabber.h grabs the OpenGL frame).
//
// There is no window on this path.  What the reference's Visualizer::render draws every step (Visualizer.cpp:159-229: the
// three shaded track bands, the active sensor rays, the agents) is rasterised on the CPU into an RGBA8 buffer of the window
// size WHEN somebody asks for it -- getRenderTargetHost(), saveRenderTargetToFile() -- through a painter the Environment
// installs; nothing is drawn per step.  Without a painter the target is blank.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "Typedefs.h"

class OKENV_CLASS ScreenGrabber
{
  public:
    // what Pybind/bindings.cpp:62-68 exposes: int extents, RGBA8, bytes per row
    struct RenderTargetInfo
    {
        int width{0};
        int height{0};
        int channels{4};

        size_t row_bytes() const { return static_cast<size_t>(width) * static_cast<size_t>(channels); }
    };
    // fills a width x height RGBA8 buffer (row-major, origin top left) with the current frame
    using Painter = std::function<void(std::vector<uint8_t> &rgba, int width, int height)>;

    ScreenGrabber(int width, int height) : info_{width, height, 4} {}

    void setPainter(Painter p) { painter_ = std::move(p); }

    std::vector<uint8_t> getRenderTargetHost() const
    {
        std::vector<uint8_t> rgba(info_.row_bytes() * static_cast<size_t>(info_.height), 0);
        if (painter_)
            painter_(rgba, info_.width, info_.height);
        return rgba;
    }
    RenderTargetInfo getRenderTargetInfo() const { return info_; }
    // PNG (8-bit RGBA, a small built-in deflate: no compression library needed); ".ppm" writes a binary PPM instead
    void saveRenderTargetToFile(const std::string &filename) const;

  private:
    RenderTargetInfo info_;
    Painter          painter_{};
};
