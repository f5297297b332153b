// ok_raycast.h -- ray/segment intersection and the uniform-grid first-hit traversal.
//
// Replaces the reference's brute-force sweep `castRaysToSegmentsKernel`
// (/root/reference Environment/CollisionChecker.cu:37-71, intersection :8-35) with an exactness-preserving broad
// phase: segments are binned into a uniform grid (ok_grid.h) and each ray walks the cells it crosses
// in order of increasing t (Amanatides-Woo DDA), looking only at the segments registered there, and stops
// once its current first hit lies inside the part of the ray already walked.
//
// Why the result is bit-identical to the brute-force sweep:
//  * each exact test (`ok_ray_segment`) performs the reference's fp32 operations in the reference's order
//    with no FMA contraction (this TU is compiled -ffp-contract=off), so every candidate's `t` carries the
//    same bits as in the sweep;
//  * the sweep's result is `min(200, min over valid t)` (order independent: SURVEY.md appendix A.6), so
//    it suffices that the set of exactly-tested segments contains the arg-min segment;
//  * a valid hit's point lies (to ~1e-4 px) on the segment; every segment is registered in every cell
//    that comes within `margin` (>= 0.125 px, >> any rounding in the walk) of it, so by the time the walk
//    has covered ray parameter t every segment that can produce a hit <= t has been looked at.  The walk ends
//    when min_t <= t_exit(current cell), or at the sensor range, or on leaving the grid's bounding box
//    (which contains every segment plus the margin);
//  * the compact ("poly") form additionally skips a segment WITHOUT the exact test when both its end
//    points lie clearly on the same side of the ray's supporting line: |side| > side_tol for both and equal
//    signs, where side(p) = cross(p - o, d).  The reference accepts a hit only if s = num_s/denom is in [0,1],
//    i.e. num_s and denom have equal signs and |num_s| <= |denom|; num_s is side(p1) and num_s - denom is
//    side(p2) up to rounding differences bounded by D ~ 4 ulp of the operands' magnitude, so an accepted hit
//    has side(p1) >= -D and side(p2) <= +D (or mirrored) and can never be skipped once side_tol > D.
//    ok_grid.h derives side_tol from the largest |p - o| a walk can meet, with a 16x safety factor.
//
// The code is `__host__ __device__` and free of GPU intrinsics so that tests/cpp can drive the very same
// traversal on the CPU against the oracle's brute force for millions of rays.
#pragma once

#include <stdint.h>

#include "../../include/okenv_math.h"

#if defined(__HIPCC__)
#define OKRC_HD __host__ __device__ __forceinline__
#else
#define OKRC_HD inline
#endif

#define OKRC_INF __builtin_huge_valf()

// Segment2d of the reference (Environment/Typedefs.h:101-105): x1,y1,x2,y2, 16 bytes.
struct OkSeg
{
    float x1, y1, x2, y2;
};

struct OkPoint
{
    float x, y;
};

// Geometry of a built grid (ok_grid.h builds it on the host).
struct OkGridGeom
{
    float x0, y0;   // lower corner of cell (0,0)
    float x1, y1;   // upper corner of the grid
    float cell;     // cell edge [px]
    float inv_cell; // 1 / cell
    int   nx, ny;
};

// Wide view: CSR starts and 32-bit segment indices, read from global memory (served by L2 / Infinity
// Cache).  Used when the compact image does not fit LDS, and by the host-side statistics.
struct OkGridView32
{
    OkGridGeom      g;
    const OkSeg    *segs;
    const uint32_t *refs;
    const uint32_t *start; // ncell + 1

    OKRC_HD void cellRange(const int c, uint32_t &k, uint32_t &k_end) const
    {
        k     = start[c];
        k_end = start[c + 1];
    }
    OKRC_HD OkSeg seg(const uint32_t k) const
    {
        return segs[refs[k]];
    }
};

// Compact "poly" view, staged into LDS by the step kernel (layout built by ok_grid.h: a cell-major stream of
// boundary points and an 8-byte header per cell).
struct OkCellHdr
{
    uint32_t w0;  // first_slot (20 bits) | n_slots << 20 (6 bits, even, <= 32) | has_next << 26
    uint32_t brk; // bit j set: NO segment joins slots first_slot + j - 1 and first_slot + j (run start, padding); bit 0 is set
};
// A cell with more than 32 slots continues in further chunks: when has_next is set, the next chunk's header occupies
// the slot right after this chunk's slots (slot index first_slot + n_slots, 8 bytes, followed by one unused slot).  A
// continuation chunk starts with a copy of its predecessor's last point, so the pair straddling the cut is examined.

struct OkPolyView
{
    OkGridGeom       g;
    const OkPoint   *slots;
    const OkCellHdr *hdr;
    float            side_tol;
};
#define OKPOLY_IDX_BITS 20
#define OKPOLY_IDX_MASK 0xFFFFFU
#define OKPOLY_MAX_SLOTS 32U
#define OKPOLY_N_MASK 0x3FU

struct OkPointPair // two consecutive slots, 16 bytes: one ds_read_b128
{
    OkPoint a, b;
};

// Environment/CollisionChecker.cu:8-35, operation for operation.
OKRC_HD bool ok_ray_segment(const float ox,
                            const float oy,
                            const float rdx,
                            const float rdy,
                            const float sx1,
                            const float sy1,
                            const float sx2,
                            const float sy2,
                            const float range,
                            float      &t_out)
{
    const float sdx   = sx2 - sx1;
    const float sdy   = sy2 - sy1;
    const float denom = rdx * sdy - rdy * sdx;
    if (__builtin_fabsf(denom) < OK_PARALLEL_EPS)
        return false;
    // Same four comparisons as the reference, evaluated t-first: a candidate beyond the current first hit (the outer
    // boundary behind the inner one, typically) is dropped before its second division.  The accepted set is identical.
    const float t = ((sx1 - ox) * sdy - (sy1 - oy) * sdx) / denom;
    if (!((t >= 0.0F) && (t <= range)))
        return false;
    const float s = ((sx1 - ox) * rdy - (sy1 - oy) * rdx) / denom;
    if ((s >= 0.0F) && (s <= 1.0F))
    {
        t_out = t;
        return true;
    }
    return false;
}

OKRC_HD float okRcpApprox(const float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0F / x;
#endif
}

// The cell walk shared by both forms: slab clip of [0, range] against the grid box, then DDA.
struct OkWalk
{
    int   ix, iy, step_x, step_y;
    float tmax_x, tmax_y, tdel_x, tdel_y;
    float t_out; // parameter at which the ray leaves the grid box or reaches the sensor range

    // Starts the walk at ray parameter max(t_start, entry into the grid box); returns false if [t_start, range]
    // misses the grid box altogether.
    OKRC_HD bool init(const OkGridGeom &g, const float ox, const float oy, const float rdx, const float rdy, const float t_start = 0.0F)
    {
        const bool  par_x  = __builtin_fabsf(rdx) < 1e-30F;
        const bool  par_y  = __builtin_fabsf(rdy) < 1e-30F;
        // The walk only has to be conservative (the registration margin absorbs its rounding), so the device
        // uses the 1-ulp hardware reciprocal instead of an IEEE division here.
        const float inv_dx = par_x ? 0.0F : okRcpApprox(rdx);
        const float inv_dy = par_y ? 0.0F : okRcpApprox(rdy);
        float       t_in   = t_start;
        t_out              = OK_SENSOR_RANGE;
        if (par_x)
        {
            if (!(ox >= g.x0 && ox <= g.x1))
                return false;
        }
        else
        {
            const float ta = (g.x0 - ox) * inv_dx;
            const float tb = (g.x1 - ox) * inv_dx;
            t_in           = __builtin_fmaxf(t_in, __builtin_fminf(ta, tb));
            t_out          = __builtin_fminf(t_out, __builtin_fmaxf(ta, tb));
        }
        if (par_y)
        {
            if (!(oy >= g.y0 && oy <= g.y1))
                return false;
        }
        else
        {
            const float ta = (g.y0 - oy) * inv_dy;
            const float tb = (g.y1 - oy) * inv_dy;
            t_in           = __builtin_fmaxf(t_in, __builtin_fminf(ta, tb));
            t_out          = __builtin_fminf(t_out, __builtin_fmaxf(ta, tb));
        }
        if (!(t_in <= t_out)) // also rejects NaN poses
            return false;
        const float px = ox + t_in * rdx;
        const float py = oy + t_in * rdy;
        ix             = (int)__builtin_floorf((px - g.x0) * g.inv_cell);
        iy             = (int)__builtin_floorf((py - g.y0) * g.inv_cell);
        ix             = ix < 0 ? 0 : (ix >= g.nx ? g.nx - 1 : ix);
        iy             = iy < 0 ? 0 : (iy >= g.ny ? g.ny - 1 : iy);
        step_x         = (rdx >= 0.0F) ? 1 : -1;
        step_y         = (rdy >= 0.0F) ? 1 : -1;
        const float bx = g.x0 + (float)(ix + (step_x > 0 ? 1 : 0)) * g.cell;
        const float by = g.y0 + (float)(iy + (step_y > 0 ? 1 : 0)) * g.cell;
        tmax_x         = par_x ? OKRC_INF : (bx - ox) * inv_dx;
        tmax_y         = par_y ? OKRC_INF : (by - oy) * inv_dy;
        tdel_x         = par_x ? OKRC_INF : g.cell * __builtin_fabsf(inv_dx);
        tdel_y         = par_y ? OKRC_INF : g.cell * __builtin_fabsf(inv_dy);
        return true;
    }
    OKRC_HD float exitT() const
    {
        return __builtin_fminf(tmax_x, tmax_y);
    }
    // moves to the next cell; false when the walk leaves the grid
    OKRC_HD bool advance(const OkGridGeom &g)
    {
        if (tmax_x < tmax_y)
        {
            ix += step_x;
            tmax_x += tdel_x;
            return ix >= 0 && ix < g.nx;
        }
        iy += step_y;
        tmax_y += tdel_y;
        return iy >= 0 && iy < g.ny;
    }
};

// First-hit parameter of one ray: min(OK_SENSOR_RANGE, min over valid t), wide form (every registered
// segment gets the exact test).  `tests`/`cells` (optional) count work for the host-side statistics.
template <bool kCount, class Grid>
OKRC_HD float ok_cast_ray_grid(const Grid  &grid,
                               const float  ox,
                               const float  oy,
                               const float  rdx,
                               const float  rdy,
                               uint32_t    *tests,
                               uint32_t    *cells)
{
    const OkGridGeom &g     = grid.g;
    float             min_t = OK_SENSOR_RANGE;
    OkWalk            w;
    if (!w.init(g, ox, oy, rdx, rdy))
        return min_t;
    // The walk visits at most nx + ny cells; the explicit bound makes termination unconditional.
    for (int guard = g.nx + g.ny + 2; guard > 0; --guard)
    {
        uint32_t k, k_end;
        grid.cellRange(w.iy * g.nx + w.ix, k, k_end);
        if (kCount)
        {
            *cells += 1;
            *tests += (k_end - k);
        }
        for (; k < k_end; ++k)
        {
            const OkSeg sg = grid.seg(k);
            float       t;
            if (ok_ray_segment(ox, oy, rdx, rdy, sg.x1, sg.y1, sg.x2, sg.y2, min_t, t))
                min_t = t;
        }
        if (__builtin_fminf(min_t, w.t_out) <= w.exitT())
            break;
        if (!w.advance(g))
            break;
    }
    return min_t;
}

// Approximate signed distance of p from the ray's supporting line (any rounding is fine: the skip rule is
// protected by side_tol).  Same sign convention as the reference's num_s.
OKRC_HD float ok_side(const OkPoint p, const float ox, const float oy, const float rdx, const float rdy)
{
    const float ax = p.x - ox;
    const float ay = p.y - oy;
    return __builtin_fmaf(ax, rdy, -(ay * rdx));
}

// true when the segment whose end points have sides s0, s1 cannot be hit (see the header comment): equal signs and
// both magnitudes above the tolerance.  median(s0, s1, 0) is the side of smaller magnitude when the signs agree and 0
// when they differ, so one v_med3_f32 and one compare decide it on the GPU.
OKRC_HD bool ok_same_side(const float s0, const float s1, const float tol)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fabsf(__builtin_amdgcn_fmed3f(s0, s1, 0.0F)) > tol;
#else
    const float lo = __builtin_fminf(s0, s1), hi = __builtin_fmaxf(s0, s1);
    const float med = (lo > 0.0F) ? lo : ((hi < 0.0F) ? hi : 0.0F);
    return __builtin_fabsf(med) > tol;
#endif
}

// exact test of the registered segment (slot k, slot k+1); returns the updated first-hit parameter
OKRC_HD float okExactSlot(const OkPolyView &v,
                          const uint32_t    k,
                          const float       ox,
                          const float       oy,
                          const float       rdx,
                          const float       rdy,
                          const float       min_t)
{
    const OkPoint a = v.slots[k];
    const OkPoint b = v.slots[k + 1];
    float         t;
    return ok_ray_segment(ox, oy, rdx, rdy, a.x, a.y, b.x, b.y, min_t, t) ? t : min_t;
}

OKRC_HD uint32_t okCountTrailingZeros(const uint32_t x)
{
    return static_cast<uint32_t>(__builtin_ctz(x));
}

// What a walk over part of a ray reports.
struct OkIntervalResult
{
    float min_t;      // smallest valid t among the segments looked at (OK_SENSOR_RANGE if none)
    float t_reached;  // the ray has been covered up to this parameter
    bool  conclusive; // min_t is the ray's final first-hit value: a hit inside the covered part, or the walk reached
                      // the sensor range / left the grid
};

// Walks the cells the ray crosses for parameters in [t_a, t_b] and reports the first hit among the segments
// registered there.  The whole ray is the interval [0, +inf).  Splitting a ray into intervals and taking the min
// of their min_t gives the same bits as one walk: every cell overlapping [t_a, t_b] is processed, a valid t found
// in ANY cell is a valid t of the ray (the exact test does not depend on the cell it was found in), and a walk
// only stops early on a hit that lies inside the part it has covered.
template <bool kCount>
OKRC_HD OkIntervalResult ok_cast_poly_interval(const OkPolyView &v,
                                               const float       ox,
                                               const float       oy,
                                               const float       rdx,
                                               const float       rdy,
                                               const float       t_a,
                                               const float       t_b,
                                               uint32_t         *tests,
                                               uint32_t         *cells,
                                               uint32_t         *points)
{
    const OkGridGeom &g = v.g;
    OkWalk            w;
    if (!w.init(g, ox, oy, rdx, rdy, t_a))
        return {OK_SENSOR_RANGE, OK_SENSOR_RANGE, true};
    const float tol   = v.side_tol;
    float       min_t = OK_SENSOR_RANGE;
    OkCellHdr   h     = v.hdr[w.iy * g.nx + w.ix];
    // The walk visits at most nx + ny cells; the explicit bound makes termination unconditional.
    for (int guard = g.nx + g.ny + 2; guard > 0; --guard)
    {
        // header of the cell the walk would enter next (clamped; unused if the walk ends first)
        const bool go_x = w.tmax_x < w.tmax_y;
        int        nx_i = w.ix + (go_x ? w.step_x : 0);
        int        ny_i = w.iy + (go_x ? 0 : w.step_y);
        nx_i            = nx_i < 0 ? 0 : (nx_i >= g.nx ? g.nx - 1 : nx_i);
        ny_i            = ny_i < 0 ? 0 : (ny_i >= g.ny ? g.ny - 1 : ny_i);
        const OkCellHdr h_next = v.hdr[ny_i * g.nx + nx_i];

        if (kCount)
            *cells += 1;
        OkCellHdr hc = h;
        while (true) // one pass per chunk of <= 32 slots; almost every cell is a single chunk
        {
            const uint32_t k0 = hc.w0 & OKPOLY_IDX_MASK;
            const uint32_t n  = (hc.w0 >> OKPOLY_IDX_BITS) & OKPOLY_N_MASK;
            if (kCount)
                *points += n;
            // Point loop: branch-free.  Bit j of `surv` = the pair (slot k0+j-1, slot k0+j) survived the side rule.
            uint32_t surv   = 0U;
            float    s_prev = 0.F;
            for (uint32_t i = 0; i < n; i += 2)
            {
                const OkPointPair pp = *reinterpret_cast<const OkPointPair *>(&v.slots[k0 + i]);
                const float       sa = ok_side(pp.a, ox, oy, rdx, rdy);
                const float       sb = ok_side(pp.b, ox, oy, rdx, rdy);
                s_prev               = (i == 0U) ? sa : s_prev; // the chunk's first slot has no predecessor (brk bit 0)
                const uint32_t m     = (ok_same_side(s_prev, sa, tol) ? 0U : 1U) | (ok_same_side(sa, sb, tol) ? 0U : 2U);
                surv |= m << i;
                s_prev = sb;
            }
            // Exact tests of the surviving segments, all lanes of a wave together: the long, division-heavy test stays
            // out of the point loop, and a wave runs it max-over-lanes(survivors) times per cell (typically 2: the inner
            // and the outer boundary).  Order does not matter: min is order independent.
            surv &= ~hc.brk;
            while (surv != 0U)
            {
                const uint32_t j = okCountTrailingZeros(surv);
                surv &= surv - 1U;
                if (kCount)
                    *tests += 1;
                min_t = okExactSlot(v, k0 + j - 1U, ox, oy, rdx, rdy, min_t);
            }
            if (((hc.w0 >> (OKPOLY_IDX_BITS + 6)) & 1U) == 0U)
                break;
            hc = *reinterpret_cast<const OkCellHdr *>(&v.slots[k0 + n]);
        }
        const float t_exit = w.exitT();
        if (__builtin_fminf(min_t, w.t_out) <= t_exit)
            return {min_t, t_exit, true};
        if (t_exit >= t_b)
            return {min_t, t_exit, false};
        if (!w.advance(g))
            return {min_t, OK_SENSOR_RANGE, true};
        h = h_next;
    }
    return {min_t, OK_SENSOR_RANGE, true};
}

// First-hit parameter of one whole ray, compact form.
template <bool kCount>
OKRC_HD float ok_cast_ray_poly(const OkPolyView &v,
                               const float       ox,
                               const float       oy,
                               const float       rdx,
                               const float       rdy,
                               uint32_t         *tests,
                               uint32_t         *cells,
                               uint32_t         *points)
{
    return ok_cast_poly_interval<kCount>(v, ox, oy, rdx, rdy, 0.0F, OKRC_INF, tests, cells, points).min_t;
}
