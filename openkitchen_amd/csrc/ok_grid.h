// ok_grid.h -- host-side construction of the uniform grid the raycast walks (see ok_raycast.h).
//
// New functionality: the reference uploads the flat Segment2d array and every ray tests every segment
// (/root/reference Environment/TrackSegments.cu:69-76, Environment/CollisionChecker.cu:49-67).  Here the
// segments are binned once per Environment so that a ray only meets the few segments near its path.
//
// Registration rule (what the exactness argument in ok_raycast.h relies on): a segment is registered in
// every cell whose box, inflated by `margin` on all sides, is touched by the segment.  The grid's own
// box is the segments' bounding box inflated by 2*margin, so nothing lies outside it.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "ok_raycast.h"

// Default cell edge [px]: 16-24 px perform within a few percent of each other on the config tracks (measured on MI355X);
// smaller cells mean more cell steps, larger ones more points per cell.
#define OKGRID_DEFAULT_CELL 20.F
// 1: the cells' runs start in LDS columns chosen by cell position (okBuildPolyImage); 0: plain cell-major order (ablation)
#if !defined(OKGRID_ALIGN_COLUMNS)
#define OKGRID_ALIGN_COLUMNS 1
#endif

struct OkGridHost
{
    OkGridGeom            g{};
    float                 margin{0.F};
    std::vector<uint32_t> start; // CSR, ncell + 1
    std::vector<uint32_t> refs;  // segment indices, cell-major
    uint32_t              max_count{0};

    size_t numCells() const
    {
        return static_cast<size_t>(g.nx) * static_cast<size_t>(g.ny);
    }
    static size_t align16(const size_t v)
    {
        return (v + 15U) & ~static_cast<size_t>(15U);
    }
};

namespace okgrid
{
// Does segment (ax,ay)-(bx,by) touch the axis-aligned box [lx,hx]x[ly,hy]?  Liang-Barsky in fp64.
inline bool segmentTouchesBox(double ax, double ay, double bx, double by, double lx, double ly, double hx, double hy)
{
    double       t0 = 0.0, t1 = 1.0;
    const double dx = bx - ax, dy = by - ay;
    const double p[4] = {-dx, dx, -dy, dy};
    const double q[4] = {ax - lx, hx - ax, ay - ly, hy - ay};
    for (int i = 0; i < 4; ++i)
    {
        if (p[i] == 0.0)
        {
            if (q[i] < 0.0)
                return false;
        }
        else
        {
            const double r = q[i] / p[i];
            if (p[i] < 0.0)
            {
                if (r > t1)
                    return false;
                if (r > t0)
                    t0 = r;
            }
            else
            {
                if (r < t0)
                    return false;
                if (r < t1)
                    t1 = r;
            }
        }
    }
    return t0 <= t1;
}

inline bool finiteSeg(const OkSeg &s)
{
    return std::isfinite(s.x1) && std::isfinite(s.y1) && std::isfinite(s.x2) && std::isfinite(s.y2);
}
} // namespace okgrid

// Builds the grid with the given cell edge.  `max_cells` bounds nx*ny (the cell edge is enlarged if needed).
inline OkGridHost okBuildGrid(const OkSeg *segs, const size_t num_segments, float cell, const size_t max_cells = 1U << 20)
{
    OkGridHost out;
    double     minx = 1e300, miny = 1e300, maxx = -1e300, maxy = -1e300, maxabs = 0.0;
    size_t     finite = 0;
    for (size_t i = 0; i < num_segments; ++i)
    {
        if (!okgrid::finiteSeg(segs[i]))
            continue;
        ++finite;
        const double xs[2] = {segs[i].x1, segs[i].x2}, ys[2] = {segs[i].y1, segs[i].y2};
        for (int k = 0; k < 2; ++k)
        {
            minx   = std::fmin(minx, xs[k]);
            maxx   = std::fmax(maxx, xs[k]);
            miny   = std::fmin(miny, ys[k]);
            maxy   = std::fmax(maxy, ys[k]);
            maxabs = std::fmax(maxabs, std::fmax(std::fabs(xs[k]), std::fabs(ys[k])));
        }
    }
    if (finite == 0)
    {
        minx = miny = 0.0;
        maxx = maxy = 1.0;
    }
    // margin: 1/8 px, or 64 ulp of the largest coordinate if that is larger (keeps the argument valid for
    // segment sets far from the origin)
    const double ulp    = std::ldexp(1.0, static_cast<int>(std::floor(std::log2(std::fmax(maxabs, 1.0)))) - 23);
    const double margin = std::fmax(0.125, 64.0 * ulp);
    out.margin          = static_cast<float>(margin);
    const double pad    = 2.0 * margin;
    const double w = (maxx - minx) + 2.0 * pad, h = (maxy - miny) + 2.0 * pad;
    if (!(cell > 0.F))
        cell = OKGRID_DEFAULT_CELL;
    double c = cell;
    while (std::ceil(w / c) * std::ceil(h / c) > static_cast<double>(max_cells))
        c *= 1.25;
    out.g.cell     = static_cast<float>(c);
    out.g.inv_cell = 1.0F / out.g.cell;
    out.g.nx       = static_cast<int>(std::ceil(w / out.g.cell));
    out.g.ny       = static_cast<int>(std::ceil(h / out.g.cell));
    if (out.g.nx < 1)
        out.g.nx = 1;
    if (out.g.ny < 1)
        out.g.ny = 1;
    out.g.x0 = static_cast<float>(minx - pad);
    out.g.y0 = static_cast<float>(miny - pad);
    // upper corner as the traversal computes cell boundaries: x0 + n*cell in fp32
    out.g.x1 = out.g.x0 + static_cast<float>(out.g.nx) * out.g.cell;
    out.g.y1 = out.g.y0 + static_cast<float>(out.g.ny) * out.g.cell;

    const size_t          ncell = out.numCells();
    std::vector<uint32_t> count(ncell, 0U);
    const double          x0 = out.g.x0, y0 = out.g.y0, cd = out.g.cell;
    auto cellRange = [&](const OkSeg &s, int &ix0, int &ix1, int &iy0, int &iy1) {
        const double lx = std::fmin(s.x1, s.x2) - margin, hx = std::fmax(s.x1, s.x2) + margin;
        const double ly = std::fmin(s.y1, s.y2) - margin, hy = std::fmax(s.y1, s.y2) + margin;
        ix0 = static_cast<int>(std::floor((lx - x0) / cd)) - 1; // one extra cell absorbs fp32-vs-fp64 boundary drift
        ix1 = static_cast<int>(std::floor((hx - x0) / cd)) + 1;
        iy0 = static_cast<int>(std::floor((ly - y0) / cd)) - 1;
        iy1 = static_cast<int>(std::floor((hy - y0) / cd)) + 1;
        ix0 = ix0 < 0 ? 0 : ix0;
        iy0 = iy0 < 0 ? 0 : iy0;
        ix1 = ix1 >= out.g.nx ? out.g.nx - 1 : ix1;
        iy1 = iy1 >= out.g.ny ? out.g.ny - 1 : iy1;
    };
    // cell box in the traversal's own fp32 arithmetic (x0 + i*cell), inflated by the margin
    auto touches = [&](const OkSeg &s, const int ix, const int iy) {
        const double lx = static_cast<double>(out.g.x0 + static_cast<float>(ix) * out.g.cell) - margin;
        const double hx = static_cast<double>(out.g.x0 + static_cast<float>(ix + 1) * out.g.cell) + margin;
        const double ly = static_cast<double>(out.g.y0 + static_cast<float>(iy) * out.g.cell) - margin;
        const double hy = static_cast<double>(out.g.y0 + static_cast<float>(iy + 1) * out.g.cell) + margin;
        return okgrid::segmentTouchesBox(s.x1, s.y1, s.x2, s.y2, lx, ly, hx, hy);
    };
    for (int pass = 0; pass < 2; ++pass)
    {
        if (pass == 1)
        {
            out.start.assign(ncell + 1, 0U);
            for (size_t cidx = 0; cidx < ncell; ++cidx)
            {
                out.start[cidx + 1] = out.start[cidx] + count[cidx];
                if (count[cidx] > out.max_count)
                    out.max_count = count[cidx];
            }
            out.refs.assign(out.start[ncell], 0U);
            std::fill(count.begin(), count.end(), 0U);
        }
        for (size_t i = 0; i < num_segments; ++i)
        {
            const OkSeg &s = segs[i];
            if (!okgrid::finiteSeg(s))
                continue; // can never produce a valid hit: every comparison on NaN/Inf quotients fails
            int ix0, ix1, iy0, iy1;
            cellRange(s, ix0, ix1, iy0, iy1);
            for (int iy = iy0; iy <= iy1; ++iy)
            {
                for (int ix = ix0; ix <= ix1; ++ix)
                {
                    if (!touches(s, ix, iy))
                        continue;
                    const size_t cidx = static_cast<size_t>(iy) * out.g.nx + ix;
                    if (pass == 1)
                        out.refs[out.start[cidx] + count[cidx]] = static_cast<uint32_t>(i);
                    ++count[cidx];
                }
            }
        }
    }
    return out;
}

// The compact "poly" image staged into LDS: a cell-major stream of boundary points.
//   slots[]   8 B per point.  Each cell owns a contiguous, 16-byte aligned range of slots holding, run after run,
//             the points of the chained segments registered in it (a run of n chained segments = n + 1 points),
//             padded to an even count with a copy of its last point.
//   hdr[cell] 8 bytes: { first_slot | (n_slots << 20), brk }: a break bit per slot j, set when NO segment joins slot
//             first_slot + j - 1 to slot first_slot + j (first point of a run, padding, every j >= n_slots), stored in the
//             point loop's accumulator order (OkCellHdr in ok_raycast.h).  n_slots <= 32.
// Consecutive slots k, k+1 with the break bit of k+1 clear are one registered segment; because chained segments of the
// reference's boundary polylines share end points bit for bit, a run costs one point per segment.
struct OkPolyImage
{
    bool                 ok{false}; // encodable (index/count fields wide enough)
    uint32_t             num_slots{0};
    uint32_t             num_runs{0};
    std::vector<uint8_t> bytes;     // slots | hdr, each part 16-byte aligned
    size_t               off_hdr{0};
    uint32_t             max_slots_per_cell{0};
    uint32_t             pad_slots{0}; // slots spent on starting cells' runs in their LDS columns
    float                side_tol{0.F};
    float                max_seg_len{0.F};
};

inline OkPolyImage okBuildPolyImage(const OkSeg *segs, const size_t num_segments, const OkGridHost &grid)
{
    OkPolyImage img;
    auto sameBits = [](const float a, const float b) {
        uint32_t ua, ub;
        std::memcpy(&ua, &a, 4);
        std::memcpy(&ub, &b, 4);
        return ua == ub;
    };
    // chained[i]: segment i starts where segment i-1 ends, bit for bit
    std::vector<uint8_t> chained(num_segments, 0);
    double               max_len = 0.0;
    for (size_t i = 0; i < num_segments; ++i)
    {
        chained[i] = (i > 0 && sameBits(segs[i].x1, segs[i - 1].x2) && sameBits(segs[i].y1, segs[i - 1].y2)) ? 1 : 0;
        if (okgrid::finiteSeg(segs[i]))
            max_len = std::fmax(max_len, std::hypot(static_cast<double>(segs[i].x2) - segs[i].x1,
                                                    static_cast<double>(segs[i].y2) - segs[i].y1));
    }
    img.max_seg_len = static_cast<float>(max_len);
    const size_t           ncell = grid.numCells();
    std::vector<OkPoint>   slots;
    std::vector<uint8_t>   brk; // one byte per slot while building
    std::vector<OkCellHdr> hdr(ncell, OkCellHdr{0U, ~0U});
    bool                   encodable = true;
    std::vector<OkPoint>   cs; // the cell's slots before chunking
    std::vector<uint8_t>   cb;
    // Order of the cells' runs in the slot stream.  A lane reads its cell's slots 16 bytes at a time (ds_read_b128), all lanes of
    // a wave at the same offset into their cells; the LDS serves eight lanes per cycle, without a conflict when their 16-byte
    // pieces are the same piece or fall into different 16-byte columns of its 128-byte row.  The rays of a fan sit in a few
    // NEIGHBOURING cells at any time, so cell (cx, cy)'s run is made to start in column (cx + 2 cy) mod 8 -- different for any two
    // cells of a 3 x 3 block.  The stream is cell-major in no particular order (the headers carry the starts), so this is a
    // matter of emitting, at every point, a cell that wants the column the stream has reached, and padding (16 bytes per column
    // skipped) only when no such cell is left.  Cells without segments own no slots.
    const bool                       align_columns = OKGRID_ALIGN_COLUMNS != 0;
    std::vector<std::vector<size_t>> want(8);
    size_t                           to_emit = ncell;
    if (align_columns)
    {
        to_emit = 0;
        for (size_t c = ncell; c-- > 0;) // (descending, so that pop_back hands the cells of a column out in ascending order)
            if (grid.start[c + 1] > grid.start[c])
            {
                want[(c % static_cast<size_t>(grid.g.nx) + 2U * (c / static_cast<size_t>(grid.g.nx))) & 7U].push_back(c);
                ++to_emit;
            }
    }
    for (size_t step = 0; step < to_emit; ++step)
    {
        size_t c = step;
        if (align_columns)
        {
            const size_t col  = (slots.size() / 2U) & 7U;
            size_t       pick = col;
            while (want[pick].empty()) // the column reached, else the nearest one ahead (least padding); one is non-empty
                pick = (pick + 1U) & 7U;
            c = want[pick].back();
            want[pick].pop_back();
            for (size_t pc = col; pc != pick; pc = (pc + 1U) & 7U)
            { // two slots = one 16-byte column
                slots.push_back({0.F, 0.F});
                slots.push_back({0.F, 0.F});
                brk.push_back(1);
                brk.push_back(1);
                img.pad_slots += 2U;
            }
        }
        cs.clear();
        cb.clear();
        uint32_t       k     = grid.start[c];
        const uint32_t k_end = grid.start[c + 1];
        while (k < k_end)
        {
            const uint32_t seg0 = grid.refs[k];
            uint32_t       n    = 1;
            while (k + n < k_end && grid.refs[k + n] == seg0 + n && chained[seg0 + n])
                ++n;
            cs.push_back({segs[seg0].x1, segs[seg0].y1});
            cb.push_back(1);
            for (uint32_t j = 0; j < n; ++j)
            {
                cs.push_back({segs[seg0 + j].x2, segs[seg0 + j].y2});
                cb.push_back(0);
            }
            ++img.num_runs;
            k += n;
        }
        if (cs.size() > img.max_slots_per_cell)
            img.max_slots_per_cell = static_cast<uint32_t>(cs.size());
        // emit chunks of at most 32 slots; a continuation chunk repeats its predecessor's last point, and its header
        // lives in the slot stream right behind the predecessor's slots
        size_t pos         = 0;
        long   hdr_in_slot = -1; // slot index holding the header to fill (-1: the cell's own header)
        bool   first       = true;
        do
        {
            const uint32_t first_slot = static_cast<uint32_t>(slots.size()); // always even
            uint32_t       count      = 0U, bits = 0U;
            if (!first)
            { // overlap: the cut pair (cs[pos-1], cs[pos]) is examined in this chunk
                slots.push_back(cs[pos - 1]);
                brk.push_back(1);
                bits |= 1U;
                ++count;
            }
            while (pos < cs.size() && count < OKPOLY_MAX_SLOTS)
            {
                slots.push_back(cs[pos]);
                brk.push_back(cb[pos]);
                if (cb[pos])
                    bits |= 1U << count;
                ++pos;
                ++count;
            }
            if (count & 1U)
            {
                if (count == OKPOLY_MAX_SLOTS)
                { // cannot pad: give the last slot back to the next chunk
                    slots.pop_back();
                    brk.pop_back();
                    --pos;
                    --count;
                    bits &= ~(1U << count);
                }
                if (count & 1U)
                {
                    slots.push_back(slots.back());
                    brk.push_back(1);
                    bits |= 1U << count;
                    ++count;
                }
            }
            if (first_slot > OKPOLY_IDX_MASK)
                encodable = false;
            const bool      more = pos < cs.size();
            // break bits in the point loop's accumulator order (ok_raycast.h): the loop evaluates n8 = count rounded up to 8
            // slots and shifts one bit in per slot, so slot j ends up at bit n8 - 1 - j; everything that is not a pair of
            // chained points -- slot 0, run starts, padding, the slots from `count` on, the bits from n8 on -- is a break
            const uint32_t n8  = (count + 7U) & ~7U;
            uint32_t       rev = n8 < 32U ? (~0U << n8) : 0U;
            for (uint32_t j = 0; j < n8; ++j)
                if (j >= count || j == 0U || ((bits >> j) & 1U) != 0U)
                    rev |= 1U << (n8 - 1U - j);
            const OkCellHdr hv{first_slot | (count << OKPOLY_IDX_BITS) | ((more ? 1U : 0U) << (OKPOLY_IDX_BITS + 6)), rev};
            if (hdr_in_slot < 0)
                hdr[c] = hv;
            else
                std::memcpy(&slots[static_cast<size_t>(hdr_in_slot)], &hv, sizeof hv);
            if (more)
            { // reserve one slot for the next chunk's header plus one to keep chunks on even slots
                hdr_in_slot = static_cast<long>(slots.size());
                slots.push_back({0.F, 0.F});
                slots.push_back({0.F, 0.F});
                brk.push_back(1);
                brk.push_back(1);
            }
            first = false;
        } while (pos < cs.size() && encodable);
        if (!encodable)
            break;
    }
    for (int pad = 0; pad < 16; ++pad)
    { // the exact test of slot k reads k+1 and the point loop reads up to six slots past a chunk: keep those reads in bounds
        slots.push_back({0.F, 0.F});
        brk.push_back(1);
    }
    img.ok        = encodable;
    img.num_slots = static_cast<uint32_t>(slots.size());
    if (!encodable)
        return img;
    const size_t slot_b = OkGridHost::align16(slots.size() * sizeof(OkPoint));
    const size_t hdr_b  = OkGridHost::align16(hdr.size() * sizeof(OkCellHdr));
    img.off_hdr         = slot_b;
    img.bytes.assign(slot_b + hdr_b, 0);
    std::memcpy(img.bytes.data(), slots.data(), slots.size() * sizeof(OkPoint));
    std::memcpy(img.bytes.data() + img.off_hdr, hdr.data(), hdr.size() * sizeof(OkCellHdr));
    // side tolerance (ok_raycast.h).  Two parts:
    //  * any point met by a walk lies within A = range + two cell diagonals + the longest segment + margin of the ray
    //    origin; rounding differences between a point's true side and the reference's num_s / (num_s - denom) are below
    //    ~4 * 2^-23 * 4A = 2^-19 A; take 2^-17 * A (4x that);
    //  * the walk evaluates side(p) = p.x * dy - p.y * dx - c, c = o.x * dy - o.y * dx, with four roundings (o.y * dx, the
    //    FMA giving c, the two FMAs of a point) of intermediates no larger than 4M, M = the largest coordinate magnitude
    //    involved: the grid box's corners, and an origin at most one sensor range outside it (a walk only happens when the
    //    ray enters the box within the range).  That is <= 10 * 2^-24 * M away from the true side; take 2^-19 * M (3x that).
    const double A = 200.0 + 2.0 * 1.4143 * grid.g.cell + max_len + 2.0 * grid.margin;
    const double M = std::fmax(std::fmax(std::fabs(grid.g.x0), std::fabs(grid.g.x1)), std::fmax(std::fabs(grid.g.y0), std::fabs(grid.g.y1))) + 201.0;
    img.side_tol   = static_cast<float>(std::ldexp(A, -17) + std::ldexp(M, -19));
    return img;
}

// Picks the cell edge: `requested` if > 0, else the default; enlarged until the compact image fits
// `lds_budget` bytes (if it ever does).  `*fits_lds` tells whether `*image` is usable.
inline OkGridHost okBuildGridAuto(const OkSeg *segs,
                                  const size_t num_segments,
                                  const float  requested,
                                  const size_t lds_budget,
                                  bool        *fits_lds,
                                  OkPolyImage *image)
{
    float      cell = requested > 0.F ? requested : OKGRID_DEFAULT_CELL;
    OkGridHost g;
    for (int attempt = 0; attempt < 32; ++attempt)
    {
        g      = okBuildGrid(segs, num_segments, cell);
        *image = okBuildPolyImage(segs, num_segments, g);
        if (image->ok && image->bytes.size() <= lds_budget)
        {
            *fits_lds = true;
            return g;
        }
        if (num_segments * 8U + 64U > lds_budget)
            break; // the points alone do not fit: no cell size will help
        if (image->ok)
            cell *= 1.25F; // too big for LDS: fewer, larger cells
        else
            break; // more than 2^20 slots: the compact form cannot index them
    }
    *fits_lds = false;
    return okBuildGrid(segs, num_segments, requested > 0.F ? requested : OKGRID_DEFAULT_CELL);
}
