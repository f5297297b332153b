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
    struct RenderTargetInfo
    {
        size_t width{0};
        size_t height{0};
        size_t channels{4};
    };

    ScreenGrabber(size_t width, size_t height) : info_{width, height, 4} {}

    std::vector<uint8_t> getRenderTargetHost() const { return std::vector<uint8_t>(info_.width * info_.height * info_.channels, 0); }
    RenderTargetInfo     getRenderTargetInfo() const { return info_; }
    void                 saveRenderTargetToFile(const std::string & /*filename*/) const {}

  private:
    RenderTargetInfo info_;
};
