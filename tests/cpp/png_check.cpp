// png_check.cpp -- ScreenGrabber's PNG writer on a synthetic frame (flat areas, gradients, single pixels): no GPU involved.
#include "Environment/ScreenGrabber.h"

int main(int argc, char **argv)
{
    if (argc != 2)
        return 2;
    ScreenGrabber g(1600, 1400);
    g.setPainter([](std::vector<uint8_t> &p, int w, int h) {
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
            {
                uint8_t *q = &p[(static_cast<size_t>(y) * w + x) * 4];
                q[0] = static_cast<uint8_t>((x / 100) * 16);
                q[1] = static_cast<uint8_t>((y > 700) ? 255 : (x * 7 + y * 3) % 251);
                q[2] = (x == y) ? 255 : 0;
                q[3] = 255;
            }
    });
    g.saveRenderTargetToFile(argv[1]);
    return 0;
}
