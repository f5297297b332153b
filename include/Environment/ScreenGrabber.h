// ScreenGrabber.h -- headless stand-in for the reference's OpenGL frame grabber (reference Environment/ScreenGrabber.h).
// Image observations are outside this project's scope; the class exists so that
// Environment keeps its public members.  The render target is an all-zero RGBA buffer of the window size.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

class ScreenGrabber
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

    ScreenGrabber(int width, int height) : info_{width, height, 4} {}

    std::vector<uint8_t> getRenderTargetHost() const { return std::vector<uint8_t>(info_.row_bytes() * static_cast<size_t>(info_.height), 0); }
    RenderTargetInfo     getRenderTargetInfo() const { return info_; }
    void                 saveRenderTargetToFile(const std::string & /*filename*/) const {}

  private:
    RenderTargetInfo info_;
};
