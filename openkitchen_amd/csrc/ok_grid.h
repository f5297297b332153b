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
#include <algorithm>
#include <utility>
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
// `include` (optional, one byte per segment): only segments with a non-zero byte are registered -- the box, the margin and the
// cells are still those of ALL segments, so grids built over subsets of one segment set share their geometry cell for cell
// (the front / back images below).
inline OkGridHost okBuildGrid(const OkSeg *segs, const size_t num_segments, float cell, const size_t max_cells = 1U << 20,
                              const uint8_t *include = nullptr)
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
            if (include != nullptr && include[i] == 0)
                continue;
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
        cell *= 1.25F; // too big for LDS (or more slots than the headers can index): fewer, larger cells
    }
    *fits_lds = false;
    return okBuildGrid(segs, num_segments, requested > 0.F ? requested : OKGRID_DEFAULT_CELL);
}

// ---- front / back split of the segment set (round 4) ----------------------------------------------------------------------
//
// Half of the points a ray meets belong to the OUTER boundary polylines, which run 3 px behind the inner ones and can be the
// first hit only for an agent that has already left the track.  The split below lets a step look at them only when that can
// matter, and keeps the result the reference's bits:
//
//   chi       The closed chains among the segments that stand for the inner boundaries (F) define chi(p) = parity of the number
//             of F crossings of any ray from p to infinity.  chi is constant on the components of the plane minus F and flips
//             across every F segment -- for ANY closed polygonal chains, simple or not (the inner boundaries of the config
//             tracks do cross themselves in tight corners).
//   back      A segment b outside F is a BACK segment when both its end points have chi = 0 and it keeps a distance of at least
//             kBackClearance from every F segment (so chi = 0 all along it, with room for every rounding in the tests).
//             Everything else -- F itself, outer pieces that cross an inner boundary or come close to it, segments that belong
//             to no closed chain -- is a FRONT segment.
//   claim     A ray whose origin has chi = 1 meets no back segment before its first F crossing: up to that crossing it runs where
//             chi = 1, and no back segment has a point there.  Its first hit over ALL segments is therefore its first hit over
//             the front segments, provided (i) chi(origin) = 1 is established robustly (ok_raycast.h: okOriginChi, from the
//             precomputed chi of a reference point of the origin's cell and the parity of the F segments the straight line from
//             there to the origin crosses), and (ii) no F crossing was MISSED by the fp32 test for rounding reasons (the walk
//             reports candidates it rejected by less than the error bounds e_s / e_t below as ambiguous).  A ray for which either
//             fails walks the back image too: min over front and back = min over all segments, as always.
//   cells     chi is established per step from the origin's cell: a cell is certifiable when every front segment registered in
//             it belongs to F, its front slots fit one chunk, and one of four reference points keeps >= kRefClearance from F.
//
// Which chains are "the inner boundaries" cannot be read off a bare segment array: the reference's TrackSegments order
// (Environment/TrackSegments.cu:11-39: runs LI, LO, RI, RO of n/4 - 1 segments, then the four closers) is assumed and CHECKED
// (closed chains found by bit-equal end points); any other input simply gets no split.  The choice only decides how much is
// gained, never what is computed.
struct OkFrontBack
{
    bool                 ok{false};
    std::vector<uint8_t> front, back; // per segment
    std::vector<uint8_t> chi_def;     // per segment: member of F
    std::vector<uint8_t> cell_flags;  // per cell: bit 0 certifiable, bit 1 chi(reference point), bits 2-3 reference point code
    float                e_s{0.F}, e_t{0.F}, t12{0.F}, t34{0.F};
    size_t               n_f{0}, n_back{0}, n_front_other{0}, n_cells_cert{0}, n_cells_with_front{0};
};
#define OKFB_CELL_CERT 1U
#define OKFB_CELL_CHI 2U
#define OKFB_HDR_SHIFT OKFB_HDR_SHIFT_RC // the cell flags in the front image's header word w0 (ok_raycast.h), the F slot count above them

namespace okgrid
{
constexpr double kBackClearance = 0.5; // px
constexpr double kRefClearance  = 1.0; // px

inline double pointSegDist2(const double px, const double py, const double ax, const double ay, const double bx, const double by)
{
    const double dx = bx - ax, dy = by - ay;
    const double l2 = dx * dx + dy * dy;
    double       t  = l2 > 0.0 ? ((px - ax) * dx + (py - ay) * dy) / l2 : 0.0;
    t               = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    const double qx = ax + t * dx - px, qy = ay + t * dy - py;
    return qx * qx + qy * qy;
}
inline double orient(const double ax, const double ay, const double bx, const double by, const double cx, const double cy)
{
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
}
inline double segSegDist2(const OkSeg &a, const OkSeg &b)
{
    const double o1 = orient(a.x1, a.y1, a.x2, a.y2, b.x1, b.y1), o2 = orient(a.x1, a.y1, a.x2, a.y2, b.x2, b.y2);
    const double o3 = orient(b.x1, b.y1, b.x2, b.y2, a.x1, a.y1), o4 = orient(b.x1, b.y1, b.x2, b.y2, a.x2, a.y2);
    if (((o1 > 0.0) != (o2 > 0.0) || o1 == 0.0 || o2 == 0.0) && ((o3 > 0.0) != (o4 > 0.0) || o3 == 0.0 || o4 == 0.0))
        return 0.0; // they cross or touch (collinear overlaps are caught by the point distances below as 0 too)
    double d = pointSegDist2(a.x1, a.y1, b.x1, b.y1, b.x2, b.y2);
    d        = std::fmin(d, pointSegDist2(a.x2, a.y2, b.x1, b.y1, b.x2, b.y2));
    d        = std::fmin(d, pointSegDist2(b.x1, b.y1, a.x1, a.y1, a.x2, a.y2));
    d        = std::fmin(d, pointSegDist2(b.x2, b.y2, a.x1, a.y1, a.x2, a.y2));
    return d;
}
// chi of point p: crossings of the horizontal ray from p to +x with the chi-defining segments (half-open rule on y, so a
// vertex on the ray's line counts once).  *robust = false when p lies (numerically) on one of the segments' lines where it matters.
inline int chiOf(const OkSeg *segs, const size_t n, const uint8_t *chi_def, const double px, const double py, bool *robust)
{
    int parity = 0;
    for (size_t i = 0; i < n; ++i)
    {
        if (!chi_def[i])
            continue;
        const OkSeg &s = segs[i];
        const bool   up = (s.y1 <= py) && (s.y2 > py), down = (s.y2 <= py) && (s.y1 > py);
        if (!up && !down)
            continue;
        const double c = orient(s.x1, s.y1, s.x2, s.y2, px, py);
        if (std::fabs(c) < 1e-7)
            *robust = false;
        if ((up && c > 0.0) || (down && c < 0.0))
            parity ^= 1;
    }
    return parity;
}
// the four reference point candidates of a cell, in the fp32 arithmetic the device repeats (ok_raycast.h: okCellRefPoint)
inline void cellRefPoint(const OkGridGeom &g, const int ix, const int iy, const unsigned code, float *rx, float *ry)
{
    const float fx = (code & 1U) ? 0.75F : 0.25F, fy = (code & 2U) ? 0.75F : 0.25F;
    *rx            = g.x0 + (static_cast<float>(ix) + fx) * g.cell;
    *ry            = g.y0 + (static_cast<float>(iy) + fy) * g.cell;
}
} // namespace okgrid

inline OkFrontBack okClassifyFrontBack(const OkSeg *segs, const size_t n, const OkGridHost &grid, const float max_seg_len)
{
    OkFrontBack fb;
    fb.front.assign(n, 1);
    fb.back.assign(n, 0);
    fb.chi_def.assign(n, 0);
    fb.cell_flags.assign(grid.numCells(), 0);
    if (n < 16 || (n % 4U) != 0U)
        return fb;
    for (size_t i = 0; i < n; ++i)
        if (!okgrid::finiteSeg(segs[i]))
            return fb; // (a set with non-finite segments is somebody's test, not a track)
    // ---- closed chains: next[i] = the one segment that starts, bit for bit, where i ends ----
    auto key = [](const float x, const float y) {
        uint32_t ux, uy;
        std::memcpy(&ux, &x, 4);
        std::memcpy(&uy, &y, 4);
        return (static_cast<uint64_t>(ux) << 32) | uy;
    };
    std::vector<std::pair<uint64_t, uint32_t>> starts(n);
    for (size_t i = 0; i < n; ++i)
        starts[i] = {key(segs[i].x1, segs[i].y1), static_cast<uint32_t>(i)};
    std::sort(starts.begin(), starts.end());
    std::vector<int32_t> next(n, -1), indeg(n, 0);
    for (size_t i = 0; i < n; ++i)
    {
        const uint64_t k  = key(segs[i].x2, segs[i].y2);
        auto           lo = std::lower_bound(starts.begin(), starts.end(), std::make_pair(k, 0U));
        if (lo != starts.end() && lo->first == k && (lo + 1 == starts.end() || (lo + 1)->first != k))
        {
            next[i] = static_cast<int32_t>(lo->second);
            ++indeg[lo->second];
        }
    }
    std::vector<int32_t> cyc(n, -1), stamp(n, -1);
    int32_t              n_cyc = 0;
    for (size_t i = 0; i < n; ++i)
    {
        if (stamp[i] >= 0)
            continue;
        int32_t j = static_cast<int32_t>(i);
        while (j >= 0 && stamp[j] < 0)
        {
            stamp[j] = static_cast<int32_t>(i);
            j        = next[j];
        }
        if (j >= 0 && stamp[j] == static_cast<int32_t>(i))
        { // closed back into this walk: j .. j is a cycle; clean if every member has exactly one predecessor
            bool    clean = true;
            int32_t m     = j;
            do
            {
                clean = clean && indeg[m] == 1;
                m     = next[m];
            } while (m != j);
            if (clean)
            {
                do
                {
                    cyc[m] = n_cyc;
                    m      = next[m];
                } while (m != j);
                ++n_cyc;
            }
        }
    }
    // ---- F: the chains of segment 0 (LI) and of segment 2 * (n/4 - 1) (RI) in TrackSegments order ----
    const size_t  run = n / 4U - 1U;
    const int32_t c_li = cyc[0], c_ri = cyc[2U * run];
    if (c_li < 0 || c_ri < 0)
        return fb;
    for (size_t i = 0; i < n; ++i)
        if (cyc[i] == c_li || cyc[i] == c_ri)
        {
            fb.chi_def[i] = 1;
            ++fb.n_f;
        }
    // ---- back segments: chi = 0 at both ends, clear of F ----
    const OkGridGeom &g = grid.g;
    const double      reach = okgrid::kBackClearance + grid.margin;
    for (size_t i = 0; i < n; ++i)
    {
        if (fb.chi_def[i])
            continue;
        const OkSeg &b = segs[i];
        bool         robust = true;
        if (okgrid::chiOf(segs, n, fb.chi_def.data(), b.x1, b.y1, &robust) != 0 || okgrid::chiOf(segs, n, fb.chi_def.data(), b.x2, b.y2, &robust) != 0 ||
            !robust)
        {
            ++fb.n_front_other;
            continue;
        }
        // F segments near b: those registered in the cells b's inflated bounding box overlaps
        int ix0 = static_cast<int>(std::floor((std::fmin(b.x1, b.x2) - reach - g.x0) / g.cell)) - 1;
        int ix1 = static_cast<int>(std::floor((std::fmax(b.x1, b.x2) + reach - g.x0) / g.cell)) + 1;
        int iy0 = static_cast<int>(std::floor((std::fmin(b.y1, b.y2) - reach - g.y0) / g.cell)) - 1;
        int iy1 = static_cast<int>(std::floor((std::fmax(b.y1, b.y2) + reach - g.y0) / g.cell)) + 1;
        ix0     = ix0 < 0 ? 0 : ix0;
        iy0     = iy0 < 0 ? 0 : iy0;
        ix1     = ix1 >= g.nx ? g.nx - 1 : ix1;
        iy1     = iy1 >= g.ny ? g.ny - 1 : iy1;
        bool clear = true;
        for (int iy = iy0; iy <= iy1 && clear; ++iy)
            for (int ix = ix0; ix <= ix1 && clear; ++ix)
            {
                const size_t c = static_cast<size_t>(iy) * g.nx + ix;
                for (uint32_t k = grid.start[c]; k < grid.start[c + 1]; ++k)
                {
                    const uint32_t f = grid.refs[k];
                    if (fb.chi_def[f] && okgrid::segSegDist2(b, segs[f]) < okgrid::kBackClearance * okgrid::kBackClearance)
                    {
                        clear = false;
                        break;
                    }
                }
            }
        if (clear)
        {
            fb.back[i]  = 1;
            fb.front[i] = 0;
            ++fb.n_back;
        }
        else
            ++fb.n_front_other;
    }
    if (fb.n_back * 8U < n) // next to nothing to leave for later: not worth a second image
        return fb;
    // ---- error bounds of the walk's exact test and of the origin test (u = 2^-24; derivations in ok_raycast.h) ----
    const double u = std::ldexp(1.0, -24);
    const double A = 200.0 + 2.0 * 1.4143 * g.cell + max_seg_len + 2.0 * grid.margin; // farthest point from an origin a walk can meet
    const double L = max_seg_len;
    const double D = 1.4143 * g.cell + 2.0 * grid.margin + L; // farthest F point from a point of the origin's cell ... that matters
    fb.e_s         = static_cast<float>(16.0 * u * A);
    fb.e_t         = static_cast<float>(16.0 * u * A * std::fmax(L, 1.0));
    fb.t12         = static_cast<float>(16.0 * u * std::fmax(L, 1.0) * D);
    fb.t34         = static_cast<float>(16.0 * u * D * D);
    // ---- cells: certifiable?  chi of the reference point ----
    for (int iy = 0; iy < g.ny; ++iy)
        for (int ix = 0; ix < g.nx; ++ix)
        {
            const size_t c         = static_cast<size_t>(iy) * g.nx + ix;
            bool         only_f    = true;
            bool         any_front = false;
            for (uint32_t k = grid.start[c]; k < grid.start[c + 1]; ++k)
            {
                const uint32_t sidx = grid.refs[k];
                if (fb.front[sidx])
                {
                    any_front = true;
                    only_f    = only_f && fb.chi_def[sidx] != 0;
                }
            }
            if (any_front)
                ++fb.n_cells_with_front;
            (void)only_f; // (other front segments may share the cell: the image lists the F segments first and says how many slots they take)
            for (unsigned code = 0; code < 4U; ++code)
            {
                float rx, ry;
                okgrid::cellRefPoint(g, ix, iy, code, &rx, &ry);
                bool good = true;
                // clear of every F segment around, and clearly off the supporting line of every F segment registered in the cell
                // (the device's d1 then needs no tolerance test of its own)
                for (int jy = std::max(0, iy - 1); jy <= std::min(g.ny - 1, iy + 1) && good; ++jy)
                    for (int jx = std::max(0, ix - 1); jx <= std::min(g.nx - 1, ix + 1) && good; ++jx)
                    {
                        const size_t cc = static_cast<size_t>(jy) * g.nx + jx;
                        for (uint32_t k = grid.start[cc]; k < grid.start[cc + 1]; ++k)
                        {
                            const OkSeg &f = segs[grid.refs[k]];
                            if (!fb.chi_def[grid.refs[k]])
                                continue;
                            if (okgrid::pointSegDist2(rx, ry, f.x1, f.y1, f.x2, f.y2) < okgrid::kRefClearance * okgrid::kRefClearance)
                                good = false;
                            else if (cc == c && std::fabs(okgrid::orient(f.x1, f.y1, f.x2, f.y2, rx, ry)) < 64.0 * fb.t12)
                                good = false;
                            if (!good)
                                break;
                        }
                    }
                if (!good)
                    continue;
                bool      robust = true;
                const int chi    = okgrid::chiOf(segs, n, fb.chi_def.data(), rx, ry, &robust);
                if (!robust)
                    continue;
                fb.cell_flags[c] = static_cast<uint8_t>(OKFB_CELL_CERT | (chi ? OKFB_CELL_CHI : 0U) | (code << 2));
                ++fb.n_cells_cert;
                break;
            }
        }
    fb.ok = true;
    return fb;
}

// The two images of a split segment set over ONE grid geometry: `front` carries the cell flags in its headers.
struct OkFrontBackImages
{
    bool        ok{false};
    OkGridHost  grid_front, grid_back;
    OkPolyImage front, back;
};

inline OkFrontBackImages okBuildFrontBackImages(const OkSeg *segs, const size_t n, const OkGridHost &grid, OkFrontBack &fb)
{
    OkFrontBackImages out;
    if (!fb.ok)
        return out;
    out.grid_front = okBuildGrid(segs, n, grid.g.cell, 1U << 20, fb.front.data());
    out.grid_back  = okBuildGrid(segs, n, grid.g.cell, 1U << 20, fb.back.data());
    // same geometry by construction (the box is that of all finite segments, the cell edge is given)
    if (out.grid_front.g.nx != grid.g.nx || out.grid_front.g.ny != grid.g.ny || out.grid_back.g.nx != grid.g.nx || out.grid_back.g.ny != grid.g.ny ||
        out.grid_front.g.x0 != grid.g.x0 || out.grid_front.g.y0 != grid.g.y0 || out.grid_front.g.cell != grid.g.cell)
        return out;
    // the front image lists a cell's F segments first (the origin test looks at those and only those), the other front segments --
    // outer pieces near the inner boundaries' swallowtails, segments of no closed chain -- behind them; chained runs never straddle the
    // two groups (they belong to different chains)
    std::vector<uint32_t> n_f_slots(grid.numCells(), 0U);
    auto sameBits = [](const float a, const float b) {
        uint32_t ua, ub;
        std::memcpy(&ua, &a, 4);
        std::memcpy(&ub, &b, 4);
        return ua == ub;
    };
    for (size_t c = 0; c < grid.numCells(); ++c)
    {
        uint32_t *lo = out.grid_front.refs.data() + out.grid_front.start[c], *hi = out.grid_front.refs.data() + out.grid_front.start[c + 1];
        uint32_t *mid = std::stable_partition(lo, hi, [&](const uint32_t sidx) { return fb.chi_def[sidx] != 0; });
        // slots the F part takes: a run of r chained segments = r + 1 points (okBuildPolyImage's rule)
        for (uint32_t *q = lo; q < mid;)
        {
            uint32_t r = 1;
            while (q + r < mid && q[r] == q[0] + r && sameBits(segs[q[0] + r].x1, segs[q[0] + r - 1].x2) && sameBits(segs[q[0] + r].y1, segs[q[0] + r - 1].y2))
                ++r;
            n_f_slots[c] += r + 1U;
            q += r;
        }
    }
    out.front = okBuildPolyImage(segs, n, out.grid_front);
    out.back  = okBuildPolyImage(segs, n, out.grid_back);
    if (!out.front.ok || !out.back.ok)
        return out;
    OkCellHdr *hdr = reinterpret_cast<OkCellHdr *>(out.front.bytes.data() + out.front.off_hdr);
    fb.n_cells_cert = 0;
    for (size_t c = 0; c < grid.numCells(); ++c)
    {
        uint32_t       flags    = fb.cell_flags[c];
        const uint32_t n_first  = (hdr[c].w0 >> OKPOLY_IDX_BITS) & OKPOLY_N_MASK;
        const bool     has_next = ((hdr[c].w0 >> (OKPOLY_IDX_BITS + 6)) & 1U) != 0U;
        // the F slots must all sit in the cell's first chunk (a chunk that continues ends on a point the next one repeats)
        if (n_f_slots[c] > (has_next ? n_first - 1U : n_first) || n_f_slots[c] > 32U)
            flags = 0U;
        fb.cell_flags[c] = static_cast<uint8_t>(flags);
        fb.n_cells_cert += (flags & OKFB_CELL_CERT) ? 1U : 0U;
        hdr[c].w0 |= (flags << OKFB_HDR_SHIFT) | ((flags & OKFB_CELL_CERT) ? (n_f_slots[c] << OKFB_HDR_NF_SHIFT) : 0U);
    }
    out.ok = true;
    return out;
}
