// Visualizer.cpp -- software rendering of the frame the reference draws with Raylib (reference
// Environment/Visualizer.cpp:159-229), for the headless render target: track bands between the boundary polylines (right
// shoulder blue, left shoulder red, driving surface green), the active sensor rays in white, every agent as a disc in its
// colour with a yellow heading mark.  Host code, off the hot path: it runs only when a caller grabs the render target.
#include <algorithm>
#include <cmath>
#include <cstdio>

#include "Environment/Environment.h"

namespace
{
struct Canvas
{
    std::vector<uint8_t> &px;
    int                   w, h;

    void set(const int x, const int y, const Color c)
    {
        if (x < 0 || y < 0 || x >= w || y >= h)
            return;
        uint8_t *p = &px[(static_cast<size_t>(y) * w + x) * 4U];
        p[0] = c.r, p[1] = c.g, p[2] = c.b, p[3] = c.a;
    }
    void line(int x0, int y0, const int x1, const int y1, const Color c)
    { // Bresenham
        const int dx = std::abs(x1 - x0), sx = x0 < x1 ? 1 : -1, dy = -std::abs(y1 - y0), sy = y0 < y1 ? 1 : -1;
        int       err = dx + dy;
        for (int guard = 0; guard < 8 * (w + h); ++guard)
        {
            set(x0, y0, c);
            if (x0 == x1 && y0 == y1)
                break;
            const int e2 = 2 * err;
            if (e2 >= dy)
                err += dy, x0 += sx;
            if (e2 <= dx)
                err += dx, y0 += sy;
        }
    }
    void triangle(const Vec2d a, const Vec2d b, const Vec2d c, const Color col)
    { // either winding; pixel centres inside or on an edge
        const float minx = std::floor(std::min({a.x, b.x, c.x})), maxx = std::ceil(std::max({a.x, b.x, c.x}));
        const float miny = std::floor(std::min({a.y, b.y, c.y})), maxy = std::ceil(std::max({a.y, b.y, c.y}));
        const float area = (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x);
        if (!(std::fabs(area) > 1e-12F))
            return;
        for (int y = std::max(0, static_cast<int>(miny)); y <= std::min(h - 1, static_cast<int>(maxy)); ++y)
            for (int x = std::max(0, static_cast<int>(minx)); x <= std::min(w - 1, static_cast<int>(maxx)); ++x)
            {
                const float px_ = static_cast<float>(x) + 0.5F, py_ = static_cast<float>(y) + 0.5F;
                const float w0 = ((b.x - a.x) * (py_ - a.y) - (b.y - a.y) * (px_ - a.x)) / area;
                const float w1 = ((c.x - b.x) * (py_ - b.y) - (c.y - b.y) * (px_ - b.x)) / area;
                const float w2 = ((a.x - c.x) * (py_ - c.y) - (a.y - c.y) * (px_ - c.x)) / area;
                if (w0 >= 0.F && w1 >= 0.F && w2 >= 0.F)
                    set(x, y, col);
            }
    }
    void disc(const Vec2d c, const float r, const Color col)
    {
        for (int y = static_cast<int>(c.y - r) - 1; y <= static_cast<int>(c.y + r) + 1; ++y)
            for (int x = static_cast<int>(c.x - r) - 1; x <= static_cast<int>(c.x + r) + 1; ++x)
            {
                const float dx = static_cast<float>(x) + 0.5F - c.x, dy = static_cast<float>(y) + 0.5F - c.y;
                if (dx * dx + dy * dy <= r * r)
                    set(x, y, col);
            }
    }
    // shadeAreaBetweenCurves (Visualizer.cpp:88-111): a strip of quads between two polylines of equal length
    void band(const std::vector<Vec2d> &c1, const std::vector<Vec2d> &c2, const Color col)
    {
        const size_t n = std::min(c1.size(), c2.size());
        for (size_t i = 0; i + 1 < n; ++i)
        {
            triangle(c1[i], c2[i], c1[i + 1], col);
            triangle(c1[i + 1], c2[i], c2[i + 1], col);
        }
    }
};

uint32_t crc32(const uint8_t *d, size_t n, uint32_t crc = 0)
{
    crc = ~crc;
    for (size_t i = 0; i < n; ++i)
    {
        crc ^= d[i];
        for (int k = 0; k < 8; ++k)
            crc = (crc >> 1) ^ (0xEDB88320U & (0U - (crc & 1U)));
    }
    return ~crc;
}
void be32(std::vector<uint8_t> &v, const uint32_t x)
{
    v.push_back(static_cast<uint8_t>(x >> 24)), v.push_back(static_cast<uint8_t>(x >> 16)), v.push_back(static_cast<uint8_t>(x >> 8)), v.push_back(static_cast<uint8_t>(x));
}
void chunk(std::FILE *f, const char *type, const std::vector<uint8_t> &data)
{
    std::vector<uint8_t> head;
    be32(head, static_cast<uint32_t>(data.size()));
    std::fwrite(head.data(), 1, 4, f);
    std::vector<uint8_t> body(type, type + 4);
    body.insert(body.end(), data.begin(), data.end());
    std::fwrite(body.data(), 1, body.size(), f);
    std::vector<uint8_t> tail;
    be32(tail, crc32(body.data(), body.size()));
    std::fwrite(tail.data(), 1, 4, f);
}
} // namespace

namespace env
{
void paintFrame(std::vector<uint8_t> &rgba, const int width, const int height, const RaceTrack &track, const std::vector<Agent *> &agents,
                const CollisionChecker *rays)
{
    Canvas cv{rgba, width, height};
    for (size_t i = 3; i < rgba.size(); i += 4)
        rgba[i] = 255; // ClearBackground(BLACK)
    cv.band(track.right_bound_inner_, track.right_bound_outer_, Color{0, 0, 255, 255});
    cv.band(track.left_bound_inner_, track.left_bound_outer_, Color{255, 0, 0, 255});
    cv.band(track.left_bound_inner_, track.right_bound_inner_, Color{0, 255, 0, 255});
    if (rays != nullptr)
    {
        const Ray_  *r = rays->getHostRays();
        const size_t n = rays->getNumRays();
        for (size_t i = 0; i < n; ++i)
            if (r[i].active)
                cv.line(static_cast<int>(r[i].x), static_cast<int>(r[i].y), static_cast<int>(r[i].hit_x), static_cast<int>(r[i].hit_y), WHITE);
    }
    for (const Agent *a : agents)
    {
        const Color c{static_cast<unsigned char>(a->color_[0]), static_cast<unsigned char>(a->color_[1]), static_cast<unsigned char>(a->color_[2]),
                      static_cast<unsigned char>(a->color_[3])};
        cv.disc(a->pos_, a->radius_, c);
        if (a->draw_agent_heading_)
        { // heading mark (drawAgent, Visualizer.cpp:113-157)
            const float k = 0.01745329238474369049072265625F * a->rot_;
            cv.line(static_cast<int>(a->pos_.x), static_cast<int>(a->pos_.y), static_cast<int>(a->pos_.x + 2.F * a->radius_ * std::cos(k)),
                    static_cast<int>(a->pos_.y + 2.F * a->radius_ * std::sin(k)), YELLOW);
        }
    }
}
} // namespace env

void ScreenGrabber::saveRenderTargetToFile(const std::string &filename) const
{
    const std::vector<uint8_t> rgba = getRenderTargetHost();
    std::FILE                 *f    = std::fopen(filename.c_str(), "wb");
    if (!f)
        return;
    const int w = info_.width, h = info_.height;
    if (filename.size() > 4 && filename.compare(filename.size() - 4, 4, ".ppm") == 0)
    {
        std::fprintf(f, "P6\n%d %d\n255\n", w, h);
        for (size_t i = 0; i < rgba.size(); i += 4)
            std::fwrite(&rgba[i], 1, 3, f);
        std::fclose(f);
        return;
    }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::fwrite(sig, 1, 8, f);
    std::vector<uint8_t> ihdr;
    be32(ihdr, static_cast<uint32_t>(w)), be32(ihdr, static_cast<uint32_t>(h));
    ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0}); // 8 bits, RGBA, deflate, no filter, no interlace
    chunk(f, "IHDR", ihdr);
    // zlib stream, one deflate block with the fixed Huffman codes; every scanline is preceded by filter byte 0.  The only
    // matches looked for are repeats of the previous pixel (distance 4): frames are flat colour, so that alone takes a
    // 9 MB frame to a few hundred KB.
    std::vector<uint8_t> raw;
    raw.reserve(static_cast<size_t>(h) * (info_.row_bytes() + 1U));
    for (int y = 0; y < h; ++y)
    {
        raw.push_back(0);
        raw.insert(raw.end(), rgba.begin() + static_cast<long>(y) * static_cast<long>(info_.row_bytes()),
                   rgba.begin() + static_cast<long>(y + 1) * static_cast<long>(info_.row_bytes()));
    }
    std::vector<uint8_t> z{0x78, 0x01};
    uint32_t             bitbuf = 0;
    int                  bitcnt = 0;
    auto put = [&](uint32_t value, int nbits) { // LSB first
        bitbuf |= value << bitcnt;
        bitcnt += nbits;
        while (bitcnt >= 8)
        {
            z.push_back(static_cast<uint8_t>(bitbuf));
            bitbuf >>= 8;
            bitcnt -= 8;
        }
    };
    auto huff = [&](uint32_t code, int nbits) { // Huffman codes go MSB first
        uint32_t r = 0;
        for (int i = 0; i < nbits; ++i)
            r |= ((code >> i) & 1U) << (nbits - 1 - i);
        put(r, nbits);
    };
    auto symbol = [&](int sym) { // fixed literal/length code (RFC 1951, 3.2.6)
        if (sym < 144) huff(0x30U + sym, 8);
        else if (sym < 256) huff(0x190U + (sym - 144), 9);
        else if (sym < 280) huff(static_cast<uint32_t>(sym - 256), 7);
        else huff(0xC0U + (sym - 280), 8);
    };
    static const int kLenBase[29]  = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const int kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    put(1, 1); // final block
    put(1, 2); // fixed Huffman
    for (size_t i = 0; i < raw.size();)
    {
        size_t run = 0;
        if (i >= 4)
            while (i + run < raw.size() && run < 258 && raw[i + run] == raw[i + run - 4])
                ++run;
        if (run >= 3)
        {
            int c = 28;
            while (kLenBase[c] > static_cast<int>(run))
                --c;
            symbol(257 + c);
            put(static_cast<uint32_t>(static_cast<int>(run) - kLenBase[c]), kLenExtra[c]);
            huff(3, 5); // distance code 3 = distance 4, no extra bits
            i += run;
        }
        else
        {
            symbol(raw[i]);
            ++i;
        }
    }
    symbol(256); // end of block
    if (bitcnt > 0)
        put(0, 8 - bitcnt);
    uint32_t s1 = 1, s2 = 0;
    for (size_t off = 0; off < raw.size(); off += 4096U)
    { // Adler-32 of the uncompressed stream
        const size_t n = std::min<size_t>(4096U, raw.size() - off);
        for (size_t i = off; i < off + n; ++i)
        {
            s1 += raw[i];
            s2 += s1;
        }
        s1 %= 65521U;
        s2 %= 65521U;
    }
    be32(z, (s2 << 16) | s1);
    chunk(f, "IDAT", z);
    chunk(f, "IEND", {});
    std::fclose(f);
}
