// ok_raycast.h -- ray/segment intersection and the uniform-grid first-hit traversal.
//
// Replaces the reference's brute-force sweep `castRaysToSegmentsKernel`
// (/root/reference Environment/CollisionChecker.cu:37-71, intersection :8-35) with an exactness-preserving broad
// phase: segments are binned into a uniform grid (ok_grid.h) and each ray walks the cells it crosses
// in order of increasing t (Amanatides-Woo DDA), testing only the segments registered there, and stops
// once its current first hit lies inside the part of the ray already walked.
//
// Why the result is bit-identical to the brute-force sweep:
//  * each individual test (`ok_ray_segment`) performs the reference's fp32 operations in the
//    reference's order with no FMA contraction (this TU is compiled -ffp-contract=off), so every
//    candidate's `t` carries the same bits as in the sweep;
//  * the sweep's result is `min(200, min over valid t)` (order independent: SURVEY.md appendix A.6), so
//    it suffices that the candidate set contains the arg-min segment;
//  * a valid hit's point lies (to ~1e-4 px) on the segment; every segment is registered in every cell
//    that comes within `margin` (>= 0.125 px, >> any rounding in the walk) of it, so by the time the walk
//    has covered ray parameter t every segment that can produce a hit <= t has been tested.  The walk ends
//    when min_t <= t_exit(current cell), or at the sensor range, or on leaving the grid's bounding box
//    (which contains every segment plus the margin).
//
// The code is `__host__ __device__` and free of GPU intrinsics so that tests/cpp can drive the very same
// traversal on the CPU against the oracle's brute force for tens of millions of rays.
#pragma once

#include <stdint.h>

#include "../../include/okenv_math.h"

#if defined(__HIPCC__)
#define OKRC_HD __host__ __device__ __forceinline__
#else
#define OKRC_HD inline
#endif

// Segment2d of the reference (Environment/Typedefs.h:101-105): x1,y1,x2,y2, 16 bytes.
struct OkSeg
{
    float x1, y1, x2, y2;
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

// Compact view used when the whole structure is staged into LDS (or is small):
//   hdr[cell] = (first_ref << 16) | count   (both < 65536),  refs[k] = 16-bit segment index.
// Pointers may address LDS or global memory.
struct OkGridView16
{
    OkGridGeom      g;
    const OkSeg    *segs;
    const uint16_t *refs;
    const uint32_t *hdr;

    OKRC_HD void cellRange(const int c, uint32_t &k, uint32_t &k_end) const
    {
        const uint32_t h = hdr[c];
        k                = h >> 16;
        k_end            = k + (h & 0xFFFFU);
    }
    OKRC_HD OkSeg seg(const uint32_t k) const
    {
        return segs[refs[k]];
    }
};

// Wide view for segment sets too large for the compact form: CSR starts and 32-bit indices, read
// from global memory (served by L2 / Infinity Cache).
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

#define OKRC_INF __builtin_huge_valf()

// First-hit parameter of one ray: min(OK_SENSOR_RANGE, min over valid t).  `tests` (optional) counts
// ray-segment tests, `cells` the cells visited -- used by the host-side statistics in tests/.
template <bool kCount, class Grid>
OKRC_HD float ok_cast_ray_grid(const Grid       &grid,
                               const float       ox,
                               const float       oy,
                               const float       rdx,
                               const float       rdy,
                               uint32_t         *tests,
                               uint32_t         *cells)
{
    const OkGridGeom &g = grid.g;
    float min_t = OK_SENSOR_RANGE;

    // ---- clip the ray [0, range] against the grid box (slab method) --------------------------
    const bool  par_x  = __builtin_fabsf(rdx) < 1e-30F;
    const bool  par_y  = __builtin_fabsf(rdy) < 1e-30F;
    const float inv_dx = par_x ? 0.0F : 1.0F / rdx;
    const float inv_dy = par_y ? 0.0F : 1.0F / rdy;
    float       t_in   = 0.0F;
    float       t_out  = OK_SENSOR_RANGE;
    if (par_x)
    {
        if (!(ox >= g.x0 && ox <= g.x1))
            return min_t;
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
            return min_t;
    }
    else
    {
        const float ta = (g.y0 - oy) * inv_dy;
        const float tb = (g.y1 - oy) * inv_dy;
        t_in           = __builtin_fmaxf(t_in, __builtin_fminf(ta, tb));
        t_out          = __builtin_fminf(t_out, __builtin_fmaxf(ta, tb));
    }
    if (!(t_in <= t_out)) // also rejects NaN poses
        return min_t;

    // ---- start cell ---------------------------------------------------------------------------
    const float px = ox + t_in * rdx;
    const float py = oy + t_in * rdy;
    int         ix = (int)__builtin_floorf((px - g.x0) * g.inv_cell);
    int         iy = (int)__builtin_floorf((py - g.y0) * g.inv_cell);
    ix             = ix < 0 ? 0 : (ix >= g.nx ? g.nx - 1 : ix);
    iy             = iy < 0 ? 0 : (iy >= g.ny ? g.ny - 1 : iy);

    const int   step_x = (rdx >= 0.0F) ? 1 : -1;
    const int   step_y = (rdy >= 0.0F) ? 1 : -1;
    const float bx     = g.x0 + (float)(ix + (step_x > 0 ? 1 : 0)) * g.cell;
    const float by     = g.y0 + (float)(iy + (step_y > 0 ? 1 : 0)) * g.cell;
    float       tmax_x = par_x ? OKRC_INF : (bx - ox) * inv_dx;
    float       tmax_y = par_y ? OKRC_INF : (by - oy) * inv_dy;
    const float tdel_x = par_x ? OKRC_INF : g.cell * __builtin_fabsf(inv_dx);
    const float tdel_y = par_y ? OKRC_INF : g.cell * __builtin_fabsf(inv_dy);

    // The walk visits at most nx + ny cells; the explicit bound makes termination unconditional.
    for (int guard = g.nx + g.ny + 2; guard > 0; --guard)
    {
        uint32_t k, k_end;
        grid.cellRange(iy * g.nx + ix, k, k_end);
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
        const float t_exit = __builtin_fminf(tmax_x, tmax_y);
        if (__builtin_fminf(min_t, t_out) <= t_exit)
            break;
        if (tmax_x < tmax_y)
        {
            ix += step_x;
            tmax_x += tdel_x;
            if (ix < 0 || ix >= g.nx)
                break;
        }
        else
        {
            iy += step_y;
            tmax_y += tdel_y;
            if (iy < 0 || iy >= g.ny)
                break;
        }
    }
    return min_t;
}
