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

// Compact "poly" view, staged into LDS by the step kernel.  Chained segments (seg[i].p2 == seg[i+1].p1 bit for
// bit, as along the reference's four boundary polylines) share their end points:
//   pts[]          8 B per point; segment i is (pts[pidx(i)], pts[pidx(i) + 1])
//   hdr[cell]      first_run | (run_count << 20)
//   runs[r]        first_point | (n_segments << 20): segments (first_point + j, first_point + j + 1), j < n
struct OkPolyView
{
    OkGridGeom      g;
    const OkPoint  *pts;
    const uint32_t *hdr;
    const uint32_t *runs;
    float           side_tol;
};
#define OKPOLY_IDX_BITS 20
#define OKPOLY_IDX_MASK 0xFFFFFU

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
    const float t = ((sx1 - ox) * sdy - (sy1 - oy) * sdx) / denom;
    const float s = ((sx1 - ox) * rdy - (sy1 - oy) * rdx) / denom;
    if ((t >= 0.0F) && (t <= range) && (s >= 0.0F) && (s <= 1.0F))
    {
        t_out = t;
        return true;
    }
    return false;
}

// The cell walk shared by both forms: slab clip of [0, range] against the grid box, then DDA.
struct OkWalk
{
    int   ix, iy, step_x, step_y;
    float tmax_x, tmax_y, tdel_x, tdel_y;
    float t_out; // parameter at which the ray leaves the grid box or reaches the sensor range

    // returns false if the ray misses the grid box altogether
    OKRC_HD bool init(const OkGridGeom &g, const float ox, const float oy, const float rdx, const float rdy)
    {
        const bool  par_x  = __builtin_fabsf(rdx) < 1e-30F;
        const bool  par_y  = __builtin_fabsf(rdy) < 1e-30F;
        const float inv_dx = par_x ? 0.0F : 1.0F / rdx;
        const float inv_dy = par_y ? 0.0F : 1.0F / rdy;
        float       t_in   = 0.0F;
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

// true when the segment whose end points have sides s0, s1 cannot be hit (see the header comment)
OKRC_HD bool ok_same_side(const float s0, const float s1, const float tol)
{
    const float lo = __builtin_fminf(s0, s1);
    const float hi = __builtin_fmaxf(s0, s1);
    return (lo > tol) || (hi < -tol);
}

// Compact form.  `tests` counts exact tests, `cells` cells, `points` point evaluations (statistics only).
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
    const OkGridGeom &g     = v.g;
    float             min_t = OK_SENSOR_RANGE;
    OkWalk            w;
    if (!w.init(g, ox, oy, rdx, rdy))
        return min_t;
    const float tol = v.side_tol;
    for (int guard = g.nx + g.ny + 2; guard > 0; --guard)
    {
        const uint32_t h     = v.hdr[w.iy * g.nx + w.ix];
        uint32_t       r     = h & OKPOLY_IDX_MASK;
        const uint32_t r_end = r + (h >> OKPOLY_IDX_BITS);
        if (kCount)
            *cells += 1;
        for (; r < r_end; ++r)
        {
            const uint32_t run = v.runs[r];
            uint32_t       p   = run & OKPOLY_IDX_MASK;
            const uint32_t n   = run >> OKPOLY_IDX_BITS;
            OkPoint        p0  = v.pts[p];
            float          s0  = ok_side(p0, ox, oy, rdx, rdy);
            if (kCount)
                *points += n + 1;
            for (uint32_t j = 0; j < n; ++j)
            {
                ++p;
                const OkPoint p1 = v.pts[p];
                const float   s1 = ok_side(p1, ox, oy, rdx, rdy);
                if (!ok_same_side(s0, s1, tol))
                {
                    if (kCount)
                        *tests += 1;
                    float t;
                    if (ok_ray_segment(ox, oy, rdx, rdy, p0.x, p0.y, p1.x, p1.y, min_t, t))
                        min_t = t;
                }
                p0 = p1;
                s0 = s1;
            }
        }
        if (__builtin_fminf(min_t, w.t_out) <= w.exitT())
            break;
        if (!w.advance(g))
            break;
    }
    return min_t;
}
