// IpcMsgs.h -- fixed-capacity message records that several of the reference's agents include through
// "Environment/IpcMsgs.h" (reference Environment/IpcMsgs.h; used by its shared-memory demos).
// Nothing on the step path reads them; the header exists so that those agents' sources keep compiling against this
// include directory.  Layouts follow the reference: a running index plus an RGBA frame, and an element count plus a
// fixed array of 2-D points.
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>

#include "Typedefs.h"

// One rendered frame: `idx` counts frames, `data` holds WIDTH x HEIGHT RGBA8 pixels.
template <size_t WIDTH, size_t HEIGHT>
struct ImageMsg
{
    static constexpr size_t kChannels = 4;

    size_t                                          idx;
    std::array<uint8_t, WIDTH * HEIGHT * kChannels> data;
};

// One lidar scan: the first `size` entries of `data` are the hit points.
template <size_t CAPACITY>
struct Laser2dMsg
{
    size_t                      size;
    std::array<Vec2d, CAPACITY> data;
};
