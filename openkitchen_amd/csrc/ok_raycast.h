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
    uint32_t w0;  // first_slot (15 bits: an image that fits the 160 KB of LDS has < 20 480 slots) | n_slots << 15 (6 bits, even, <= 32) |
                  // has_next << 21 | front / back split, front image only: cell flags << 22 (4 bits) | slots of F segments << 26 (6 bits)
    uint32_t brk; // break bits, in the point loop's accumulator order: with n8 = n_slots rounded up to 8, bit (n8 - 1 - j) is
                  // set when NO segment joins slots first_slot + j - 1 and first_slot + j (j = 0, run starts, padding, every
                  // j >= n_slots); the bits from n8 on are set too
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
    // front / back split only (ok_grid.h, okClassifyFrontBack): absolute error bounds of the exact test's folded numerators
    // fs (e_s) and ft (e_t) against their exact values -- a candidate rejected by less than that may be a crossing the fp32
    // arithmetic missed and is reported as ambiguous
    float            e_s, e_t, e_s_over_e_t;
};
#define OKPOLY_IDX_BITS 15
#define OKPOLY_IDX_MASK 0x7FFFU
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

// Bit pattern helpers (plain C++: memcpy compiles to nothing on both sides).
OKRC_HD uint32_t okBits(const float x)
{
    uint32_t u;
    __builtin_memcpy(&u, &x, 4);
    return u;
}
OKRC_HD float okFromBits(const uint32_t u)
{
    float x;
    __builtin_memcpy(&x, &u, 4);
    return x;
}

OKRC_HD int32_t okMin3i(const int32_t a, const int32_t b, const int32_t c)
{
    const int32_t m = a < b ? a : b;
    return m < c ? m : c;
}
OKRC_HD int32_t okMax3i(const int32_t a, const int32_t b, const int32_t c)
{
    const int32_t m = a > b ? a : b;
    return m > c ? m : c;
}

// The reference's test (ok_ray_segment) against the current first hit `min_t`, with the two IEEE divisions taken out of
// the common path: returns the updated first-hit parameter, bit for bit what
//     float t; return ok_ray_segment(ox, oy, rdx, rdy, a.x, a.y, b.x, b.y, min_t, t) ? t : min_t;
// returns.  num_t, num_s and denom are computed exactly as there.  When all three lie in a "regular" range
// (|denom| in [1e-8, 1e10), |num| in [1e-20, 1e10)) every quotient num/denom is a normal number of magnitude >= 1e-30
// whose sign is the product of the signs, so with the numerators' signs folded by denom's (fs, ft):
//     s >= 0  <=>  fs > 0                       s <= 1  <=>  fs <= |denom|
//         (a correctly rounded quotient of two floats exceeds 1 whenever the numerator exceeds the denominator:
//          the smallest such quotient is 1 + 1/(2^24 - 1) > 1 + 2^-24, which rounds up, never to 1)
//     t >= 0  <=>  ft > 0
//     t <= min_t  =>  ft <= (min_t * 1.00001) * |denom|      (rounding of t and of the products is < 4e-7 relative)
// The last line is only a filter: survivors compute t = num_t / denom (IEEE) and compare it with min_t exactly as the
// reference does, so it merely has to let every acceptable candidate through.  Anything irregular (zero or tiny
// numerators, huge or non-finite values) takes the reference's own sequence of two divisions.
//
// The range tests run on the bit patterns: non-negative floats order like their bits read as integers, NaNs and
// infinities sort above every finite value, and "0 < x <= y" for y >= 0 is the single unsigned comparison
// bits(x) - 1 < bits(y) (x = +0 wraps to the top, negative x and NaNs have bits above every non-negative finite y).
//
// kAmb (front / back split): *amb tells whether the candidate may be a crossing of the ray with this segment, in the exact
// geometry of the same fp32 coordinates, that the fp32 test MISSED.  With fs*, ft*, d* the exact values of the folded numerators and |denom|, an exact crossing at
// t* in (0, min_t) has 0 <= fs* <= d* and ft* > 0; the computed values differ from them by at most e_s, e_t (ok_grid.h:
// |fs - fs*| <= 6uA, |d - d*| <= 6uL, |ft - ft*| <= 8uAL with u = 2^-24, A the farthest a point can lie from an origin, L the
// longest segment; e_s = 16uA, e_t = 16uAL).  So a rejected candidate with fs < -e_s, fs > |denom| + e_s or ft < -e_t holds no such
// crossing: a missed crossing has fs within e_s of 0 or of |denom|, or |ft| <= e_t, and that is what is flagged (taken or not,
// in range or not: the superset costs nothing); every candidate on the irregular or near-parallel path is flagged as well.
template <bool kAmb = false>
OKRC_HD float ok_first_hit_update(const float   ox,
                                  const float   oy,
                                  const float   rdx,
                                  const float   rdy,
                                  const OkPoint a,
                                  const OkPoint b,
                                  const float   min_t,
                                  const float   e_s = 0.F,
                                  const float   e_t = 0.F,
                                  float        *amb = nullptr) // kAmb: running minimum of the candidates' distance from "missable"
{
    constexpr int32_t kBitsLo  = 0x1E3CE508; // 1e-20f
    constexpr int32_t kBitsHi  = 0x501502F9; // 1e10f
    constexpr int32_t kBitsEps = 0x322BCC77; // 1e-8f = OK_PARALLEL_EPS
    const float    sdx   = b.x - a.x;
    const float    sdy   = b.y - a.y;
    const float    denom = rdx * sdy - rdy * sdx;
    const float    ex    = a.x - ox;
    const float    ey    = a.y - oy;
    const float    num_t = ex * sdy - ey * sdx;
    const float    num_s = ex * rdy - ey * rdx;
    const uint32_t sg    = okBits(denom) & 0x80000000U;
    const uint32_t bd    = okBits(denom) ^ sg;       // bits of |denom|
    const uint32_t bfs   = okBits(num_s) ^ sg;       // bits of fs = num_s with denom's sign folded in
    const uint32_t bft   = okBits(num_t) ^ sg;
    const int32_t  bas   = static_cast<int32_t>(bfs & 0x7FFFFFFFU); // bits of |num_s|, |num_t|: non-negative as integers
    const int32_t  bat   = static_cast<int32_t>(bft & 0x7FFFFFFFU);
    // regular <=> |num_s|, |num_t| in [1e-20, 1e10) and |denom| in [1e-8, 1e10); |denom|'s lower bound is moved onto the
    // numerators' by the (signed) offset, so one three-way minimum and one three-way maximum decide it
    const int32_t lo = okMin3i(bas, bat, static_cast<int32_t>(bd) - (kBitsEps - kBitsLo));
    const int32_t hi = okMax3i(bas, bat, static_cast<int32_t>(bd));
    float         result = min_t;
    // kAmb: how far the candidate is from the two ways a crossing can be missed -- fs within e_s of 0 or of |denom| (the ray
    // passes a segment's end within rounding) and |ft| within e_t (the origin lies on the segment's line within rounding) -- as ONE
    // running minimum: near = min(| |fs - h| - h |, |ft| * e_s / e_t) with h = |denom| / 2 is <= e_s exactly when one of the two
    // holds.  Whether the candidate was taken, or lies beyond the current first hit, is not looked at: flagging a few more costs
    // nothing.  Five plain VALU operations, no lane masks.
    if (kAmb)
    {
        const float fs = okFromBits(bfs), ft = okFromBits(bft), h = 0.5F * okFromBits(bd);
        const float ws = __builtin_fabsf(__builtin_fabsf(fs - h) - h);
        *amb           = __builtin_fminf(*amb, __builtin_fminf(ws, __builtin_fabsf(ft) * e_t)); // (e_t here: the factor e_s / e_t)
    }
    if (lo >= kBitsLo && hi < kBitsHi)
    {
        const float lim = (min_t * 1.00001F) * okFromBits(bd);
        if ((bfs - 1U < bd) && (bft - 1U < okBits(lim)))
        { // s is in [0, 1]; t is positive and not clearly beyond min_t
            const float t = num_t / denom;
            result        = (t <= min_t) ? t : min_t;
        }
    }
    else
    {
        if (kAmb)
            *amb = 0.0F; // tiny, huge or non-finite values, a near-parallel pair whose ends are not clearly on one side: cannot tell
        if (!(okFromBits(bd) < OK_PARALLEL_EPS))
        { // the reference's own sequence (CollisionChecker.cu:23-33)
            const float t = num_t / denom;
            if ((t >= 0.0F) && (t <= min_t))
            {
                const float sq = num_s / denom;
                if ((sq >= 0.0F) && (sq <= 1.0F))
                    result = t;
            }
        }
    }
    return result;
}

// exact test of the registered segment (slot k, slot k+1); returns the updated first-hit parameter
template <bool kAmb = false>
OKRC_HD float okExactSlot(const OkPolyView &v,
                          const uint32_t    k,
                          const float       ox,
                          const float       oy,
                          const float       rdx,
                          const float       rdy,
                          const float       min_t,
                          float            *amb = nullptr)
{
    const OkPoint a = v.slots[k];
    const OkPoint b = v.slots[k + 1];
#if defined(OKRC_OLD_EXACT) // ablation: the reference's sequence on every survivor
    float t;
    return ok_ray_segment(ox, oy, rdx, rdy, a.x, a.y, b.x, b.y, min_t, t) ? t : min_t;
#else
    return ok_first_hit_update<kAmb>(ox, oy, rdx, rdy, a, b, min_t, v.e_s, v.e_s_over_e_t, amb);
#endif
}

OKRC_HD uint32_t okCountTrailingZeros(const uint32_t x)
{
    return static_cast<uint32_t>(__builtin_ctz(x));
}

// ---- front / back split: is the ray origin on the side of the inner boundaries where back segments cannot come first? --------
// (ok_grid.h, okClassifyFrontBack, has the argument.)  chi(origin) = chi(reference point of the origin's cell) XOR parity of the F
// segments that the straight line from the reference point to the origin crosses.  Both points lie in the cell, so only segments
// registered in the cell can be crossed, and in a certifiable cell every front segment IS an F segment: the pairs of the cell's
// (single) front chunk are all there is to test.
#define OKFB_HDR_SHIFT_RC (OKPOLY_IDX_BITS + 7) // cell flags in the front image's header word w0: certifiable, chi(ref), ref code (2 bits)
#define OKFB_HDR_NF_SHIFT (OKPOLY_IDX_BITS + 11) // ... and above them the number of the cell's slots that belong to F segments: they come
                                                 // first in the cell's (first) chunk, whatever other front segments follow them
#define OKFB_RC_CERT 1U
#define OKFB_RC_CHI 2U

// reference point `code` (0..3) of cell (ix, iy): the arithmetic ok_grid.h's okgrid::cellRefPoint repeats on the host
OKRC_HD void okCellRefPoint(const OkGridGeom &g, const int ix, const int iy, const uint32_t code, float *rx, float *ry)
{
    const float fx = (code & 1U) ? 0.75F : 0.25F, fy = (code & 2U) ? 0.75F : 0.25F;
    *rx            = g.x0 + (static_cast<float>(ix) + fx) * g.cell;
    *ry            = g.y0 + (static_cast<float>(iy) + fy) * g.cell;
}

// One F segment (a, b) against the line from r to o: 0 = certainly not crossed, 1 = certainly crossed, 2 = cannot tell.
//   d1, d2: r and o relative to the line through a, b;  d3, d4: a and b relative to the line through r, o.
// Crossed <=> (d1, d2 of opposite signs) and (d3, d4 of opposite signs); not crossed <=> one of the pairs has equal signs.  Each
// statement is only made when the values involved exceed their error bound (t12 for d1 / d2 -- and the host has picked r so that
// |d1| is 64 times that -- t34 for d3 / d4: 16 u L D and 16 u D^2 against roundings of ~5 u L D and ~5 u D^2, D = the farthest an
// F point registered in the cell can lie from a point of the cell, ok_grid.h).
OKRC_HD int okChiPairClass(const OkPoint a, const OkPoint b, const float rx, const float ry, const float ox, const float oy, const float t12, const float t34)
{
    const float sx = b.x - a.x, sy = b.y - a.y;
    const float d1 = sx * (ry - a.y) - sy * (rx - a.x);
    const float d2 = sx * (oy - a.y) - sy * (ox - a.x);
    const float wx = ox - rx, wy = oy - ry;
    const float d3 = wx * (a.y - ry) - wy * (a.x - rx);
    const float d4 = wx * (b.y - ry) - wy * (b.x - rx);
    const bool  c12 = __builtin_fabsf(d1) > t12 && __builtin_fabsf(d2) > t12; // both signs are the true ones
    const bool  c34 = __builtin_fabsf(d3) > t34 && __builtin_fabsf(d4) > t34;
    const bool  opp12 = (d1 > 0.F) != (d2 > 0.F), opp34 = (d3 > 0.F) != (d4 > 0.F);
    if ((c12 && !opp12) || (c34 && !opp34))
        return 0;
    if (c12 && c34) // (opp12 && opp34)
        return 1;
    return 2;
}

// The test as one thread makes it (the CPU tests; the kernels deal the pairs to the lanes of the agent's group: okenv_kernels.h,
// okOriginChiGroup -- same classes, same parity).  Returns 1 when chi(origin) = 1 is certain, 0 otherwise (chi = 0, an origin
// outside the grid, a cell that is not certifiable, any pair that cannot be told).
OKRC_HD int okOriginChiScalar(const OkPolyView &front, const float ox, const float oy, const float t12, const float t34)
{
    const OkGridGeom &g = front.g;
    if (!(ox >= g.x0 && ox <= g.x1 && oy >= g.y0 && oy <= g.y1)) // (NaN poses too)
        return 0;
    int ix = (int)__builtin_floorf((ox - g.x0) * g.inv_cell);
    int iy = (int)__builtin_floorf((oy - g.y0) * g.inv_cell);
    ix     = ix < 0 ? 0 : (ix >= g.nx ? g.nx - 1 : ix);
    iy     = iy < 0 ? 0 : (iy >= g.ny ? g.ny - 1 : iy);
    const OkCellHdr hc    = front.hdr[iy * g.nx + ix];
    const uint32_t  flags = (hc.w0 >> OKFB_HDR_SHIFT_RC) & 15U;
    if ((flags & OKFB_RC_CERT) == 0U)
        return 0;
    float rx, ry;
    okCellRefPoint(g, ix, iy, (flags >> 2) & 3U, &rx, &ry);
    const uint32_t k0 = hc.w0 & OKPOLY_IDX_MASK;
    const uint32_t n  = (hc.w0 >> OKPOLY_IDX_BITS) & OKPOLY_N_MASK;
    const uint32_t n8 = (n + 7U) & ~7U;
    const uint32_t nf = hc.w0 >> OKFB_HDR_NF_SHIFT; // the chunk's first nf slots are the F segments' (<= n)
    uint32_t       parity = (flags & OKFB_RC_CHI) ? 1U : 0U;
    for (uint32_t j = 0; j + 1U < nf; ++j)
    { // pair (slot j, slot j + 1) is a segment when the break bit of slot j + 1 is clear (bit n8 - 1 - (j + 1))
        if ((hc.brk >> (n8 - 2U - j)) & 1U)
            continue;
        const int cls = okChiPairClass(front.slots[k0 + j], front.slots[k0 + j + 1U], rx, ry, ox, oy, t12, t34);
        if (cls == 2)
            return 0;
        parity ^= static_cast<uint32_t>(cls);
    }
    return static_cast<int>(parity);
}

// What a walk over part of a ray reports.
struct OkIntervalResult
{
    float min_t;      // smallest valid t among the segments looked at (OK_SENSOR_RANGE if none)
    float t_reached;  // the ray has been covered up to this parameter
    bool  conclusive; // min_t is the ray's final first-hit value: a hit inside the covered part, or the walk reached
                      // the sensor range / left the grid
    bool  amb;        // kAmb walks: some candidate was rejected within rounding (ok_first_hit_update): a crossing may have been missed
};

struct alignas(16) OkVec4 // 16-byte aligned load unit (ds_read_b128): two consecutive slots
{
    float x, y, z, w;
};

// median of three; one v_med3_f32 on the GPU
OKRC_HD float okMed3(const float a, const float b, const float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(a, b, c);
#else
    const float lo = __builtin_fminf(a, b), hi = __builtin_fmaxf(a, b);
    return __builtin_fmaxf(lo, __builtin_fminf(hi, c));
#endif
}

// (acc << 1) | sign bit of d; one v_alignbit_b32 on the GPU
OKRC_HD uint32_t okShiftInSign(const uint32_t acc, const float d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(acc, okBits(d), 31U);
#else
    return (acc << 1) | (okBits(d) >> 31);
#endif
}

// Diagnostic build only (-DOKENV_STAMPS): shader-clock stamps inside the walk.  prof[0] walk set-up, [1] cell entry
// (header decode), [2] point loop, [3] exact loop, [4] leaving the cell.
#if defined(OKENV_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define OKRC_PROF_BEGIN() unsigned long long prof_last_ = __builtin_amdgcn_s_memtime()
#define OKRC_PROF(i)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        if (prof != nullptr)                                                                                           \
        {                                                                                                              \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                              \
            prof[i] += now_ - prof_last_;                                                                              \
            prof_last_ = now_;                                                                                         \
        }                                                                                                              \
    } while (0)
#else
#define OKRC_PROF_BEGIN() ((void)0)
#define OKRC_PROF(i) ((void)0)
#endif

// Walks the cells the ray crosses for parameters in [t_a, t_b] and reports the first hit among the segments
// registered there.  The whole ray is the interval [0, +inf).  Splitting a ray into intervals and taking the min
// of their min_t gives the same bits as one walk: every cell overlapping [t_a, t_b] is processed, a valid t found
// in ANY cell is a valid t of the ray (the exact test does not depend on the cell it was found in), and a walk
// only stops early on a hit that lies inside the part it has covered.
//
// Written for the way a gfx950 wave executes it: all 64 lanes run the loops in lock step, a wave is paced by its own
// instruction stream (one instruction per ~4 cycles) and by exposed LDS round trips (~150 cycles), and idle lanes are
// free.  Hence:
//   set-up      branch free (slab clip, start cell, DDA increments); the walk keeps a linear cell index and counts the
//               steps left to the grid's border, so a step is an add, not an (ix, iy) -> index multiplication
//   cell loop   the next cell's header is requested before the current cell is processed
//   point loop  eight slots per pass (four ds_read_b128), whatever the chunk's length: what is read past the chunk is
//               masked by the header's break bits.  A point's side is two FMAs: side(p) = p.x * dy - p.y * dx - c with
//               c = o.x * dy - o.y * dx formed once per walk (rounding analysis at side_tol in ok_grid.h); the skip
//               decision of a pair is the sign of tol - |med3(side, side', 0)|, shifted into a bit accumulator
//               (v_med3, v_sub, v_alignbit: no compare-to-mask round trip)
//   exact loop  pairs whose accumulator bit AND break bit are clear, ok_first_hit_update
//
// skip_unowned_start: for walks that CONTINUE a ray somebody else has covered up to t_a (the intervals of a ray cut across
// lanes; phase 2 after phase 1).  A walk over [t_x, t_a] ends with the cell in which it reaches t_a, so every cell the ray
// ENTERED before t_a has been processed by the time this walk's turn comes -- and the cell that contains t_a is, as a rule, one
// of them.  With the flag set the start cell is stepped over without being looked at when the ray entered it clearly before t_a
// (by more than kOwnEps, far above the rounding of the crossing parameters, far below a cell): each cell of a ray is then
// processed by the one walk in whose interval the ray enters it, instead of by two.  (A 25 px interval of a ray touches two or
// three 24 px cells; the first of them is the neighbour's last.)  Must be false for the walk that starts a ray.
// min_t0: a first hit already known from elsewhere (the front image's, when this walk covers the back image): the walk starts
// with it, so it ends as soon as the cells up to that parameter are covered and reports min(min_t0, what it finds).
template <bool kCount, bool kAmb = false>
OKRC_HD OkIntervalResult ok_cast_poly_interval(const OkPolyView &v,
                                               const float       ox,
                                               const float       oy,
                                               const float       rdx,
                                               const float       rdy,
                                               const float       t_a,
                                               const float       t_b,
                                               uint32_t         *tests,
                                               uint32_t         *cells,
                                               uint32_t         *points,
                                               unsigned long long *prof = nullptr,
                                               const bool        skip_unowned_start = false,
                                               const float       min_t0 = OK_SENSOR_RANGE)
{
    const OkGridGeom &g = v.g;
    OKRC_PROF_BEGIN();
    // ---- set-up: slab clip of [t_a, range] against the grid box (same arithmetic as OkWalk::init) ----
    const bool  par_x  = __builtin_fabsf(rdx) < 1e-30F;
    const bool  par_y  = __builtin_fabsf(rdy) < 1e-30F;
    const float inv_dx = par_x ? 0.0F : okRcpApprox(rdx);
    const float inv_dy = par_y ? 0.0F : okRcpApprox(rdy);
    const float tax = (g.x0 - ox) * inv_dx, tbx = (g.x1 - ox) * inv_dx;
    const float tay = (g.y0 - oy) * inv_dy, tby = (g.y1 - oy) * inv_dy;
    float       t_in  = t_a;
    float       t_out = OK_SENSOR_RANGE;
    t_in              = par_x ? t_in : __builtin_fmaxf(t_in, __builtin_fminf(tax, tbx));
    t_out             = par_x ? t_out : __builtin_fminf(t_out, __builtin_fmaxf(tax, tbx));
    t_in              = par_y ? t_in : __builtin_fmaxf(t_in, __builtin_fminf(tay, tby));
    t_out             = par_y ? t_out : __builtin_fminf(t_out, __builtin_fmaxf(tay, tby));
    const bool in_x   = !par_x || (ox >= g.x0 && ox <= g.x1);
    const bool in_y   = !par_y || (oy >= g.y0 && oy <= g.y1);
    if (!(in_x && in_y && (t_in <= t_out))) // the interval misses the grid box (also rejects NaN poses)
        return {min_t0, OK_SENSOR_RANGE, true, false};
    const float px = ox + t_in * rdx;
    const float py = oy + t_in * rdy;
    int         ix = (int)__builtin_floorf((px - g.x0) * g.inv_cell);
    int         iy = (int)__builtin_floorf((py - g.y0) * g.inv_cell);
    ix             = ix < 0 ? 0 : (ix >= g.nx ? g.nx - 1 : ix);
    iy             = iy < 0 ? 0 : (iy >= g.ny ? g.ny - 1 : iy);
    const bool  fwd_x = rdx >= 0.0F, fwd_y = rdy >= 0.0F;
    const float bx    = g.x0 + (float)(ix + (fwd_x ? 1 : 0)) * g.cell;
    const float by    = g.y0 + (float)(iy + (fwd_y ? 1 : 0)) * g.cell;
    float       tmax_x = par_x ? OKRC_INF : (bx - ox) * inv_dx;
    float       tmax_y = par_y ? OKRC_INF : (by - oy) * inv_dy;
    const float tdel_x = par_x ? OKRC_INF : g.cell * __builtin_fabsf(inv_dx);
    const float tdel_y = par_y ? OKRC_INF : g.cell * __builtin_fabsf(inv_dy);
    // linear cell index, its increments, and the steps left before the walk leaves the grid on either axis
    int       cell   = iy * g.nx + ix;
    const int lin_x  = fwd_x ? 1 : -1;
    const int lin_y  = fwd_y ? g.nx : -g.nx;
    int       left_x = fwd_x ? g.nx - 1 - ix : ix;
    int       left_y = fwd_y ? g.ny - 1 - iy : iy;
    const int last_cell = g.nx * g.ny - 1;

    const float tol    = v.side_tol;
    const float c_ray  = __builtin_fmaf(ox, rdy, -(oy * rdx)); // side(p) = p.x * dy - p.y * dx - c_ray
    const float neg_dx = -rdx;
    float       min_t  = min_t0;
    float       amb    = OKRC_INF; // kAmb: running minimum of the candidates' distance from being missable (ok_first_hit_update)

    // What leaving the current cell will mean is known when the cell is entered, except for hits found inside it:
    //   t_exit             parameter at which the ray leaves the cell
    //   cell_n, tnx, tny   the next cell (index clamped into the grid) and the DDA state after the step
    //   h_next             its header, requested now so that the LDS round trip overlaps the cell's own work
    //   leave              0 go on | 1 conclusive: range or grid box ends inside this cell (or, set later, a hit inside the covered
    //                      part) | 2 the interval's end t_b is reached, not conclusive | 3 conclusive: the walk leaves the grid
    // Every loop-carried quantity is a plain register value (no lane masks carried around the loop).
    float     t_exit, tnx, tny;
    int       cell_n, leave;
    OkCellHdr h_next;
#define OKRC_ENTER_CELL()                                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        const bool go_x = tmax_x < tmax_y;                                                                             \
        t_exit          = __builtin_fminf(tmax_x, tmax_y);                                                             \
        tnx             = go_x ? tmax_x + tdel_x : tmax_x;                                                             \
        tny             = go_x ? tmax_y : tmax_y + tdel_y;                                                             \
        left_x -= go_x ? 1 : 0;                                                                                        \
        left_y -= go_x ? 0 : 1;                                                                                        \
        cell_n = cell + (go_x ? lin_x : lin_y);                                                                        \
        cell_n = cell_n < 0 ? 0 : (cell_n > last_cell ? last_cell : cell_n);                                           \
        h_next = v.hdr[cell_n];                                                                                        \
        leave  = (left_x | left_y) >= 0 ? 0 : 3;                                                                       \
        leave  = (t_exit >= t_b) ? 2 : leave;                                                                          \
        leave  = (t_out <= t_exit) ? 1 : leave;                                                                        \
    } while (0)

    {
        // parameter at which the ray entered the start cell: the later of the two crossings before the upcoming ones (an axis
        // the ray does not move along has none)
        constexpr float kOwnEps = 1.0e-2F;
        const float tprev_x = par_x ? -OKRC_INF : tmax_x - tdel_x;
        const float tprev_y = par_y ? -OKRC_INF : tmax_y - tdel_y;
        if (skip_unowned_start && __builtin_fmaxf(tprev_x, tprev_y) < t_a - kOwnEps)
        { // the previous walk's: one step of the DDA, nothing read
            const bool  go_x = tmax_x < tmax_y;
            const float te   = __builtin_fminf(tmax_x, tmax_y);
            left_x -= go_x ? 1 : 0;
            left_y -= go_x ? 0 : 1;
            if (t_out <= te || (left_x | left_y) < 0) // the range, the grid box or the grid ends inside that cell
                return {min_t0, OK_SENSOR_RANGE, true, false};
            if (te >= t_b) // (an interval shorter than its start cell: every cell it touches is the previous walk's)
                return {min_t0, te, false, false};
            cell += go_x ? lin_x : lin_y;
            tmax_x = go_x ? tmax_x + tdel_x : tmax_x;
            tmax_y = go_x ? tmax_y : tmax_y + tdel_y;
        }
    }
    OkCellHdr hc = v.hdr[cell];
    OKRC_ENTER_CELL();
    if (kCount)
        *cells += 1;
    OKRC_PROF(0);
    // One pass per chunk of <= 32 slots (almost every cell is a single chunk; a cell's chain of chunks is finite by
    // construction of the image).  The walk visits at most nx + ny cells: the explicit bound on cell steps makes
    // termination unconditional.
    int guard = g.nx + g.ny + 2;
    while (true)
    {
        const uint32_t k0 = hc.w0 & OKPOLY_IDX_MASK;
        const uint32_t n  = (hc.w0 >> OKPOLY_IDX_BITS) & OKPOLY_N_MASK;
        if (kCount)
            *points += n;
        // Point loop.  After it, bit (n8 - 1 - j) of `skip` tells that the pair (slot k0+j-1, slot k0+j) lies clearly on one
        // side of the ray (n8 = n rounded up to 8); the header's break bits use the same positions.
        uint32_t skip = 0U;
        float    sp   = 0.F;
        OKRC_PROF(1);
        for (uint32_t i = 0; i < n; i += 8)
        {
            const OkVec4 *q  = reinterpret_cast<const OkVec4 *>(v.slots + k0 + i); // k0 + i is even, the slot array 16-byte aligned
            const OkVec4  q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const float   s0 = __builtin_fmaf(q0.x, rdy, __builtin_fmaf(q0.y, neg_dx, -c_ray));
            const float   s1 = __builtin_fmaf(q0.z, rdy, __builtin_fmaf(q0.w, neg_dx, -c_ray));
            const float   s2 = __builtin_fmaf(q1.x, rdy, __builtin_fmaf(q1.y, neg_dx, -c_ray));
            const float   s3 = __builtin_fmaf(q1.z, rdy, __builtin_fmaf(q1.w, neg_dx, -c_ray));
            const float   s4 = __builtin_fmaf(q2.x, rdy, __builtin_fmaf(q2.y, neg_dx, -c_ray));
            const float   s5 = __builtin_fmaf(q2.z, rdy, __builtin_fmaf(q2.w, neg_dx, -c_ray));
            const float   s6 = __builtin_fmaf(q3.x, rdy, __builtin_fmaf(q3.y, neg_dx, -c_ray));
            const float   s7 = __builtin_fmaf(q3.z, rdy, __builtin_fmaf(q3.w, neg_dx, -c_ray));
            skip             = okShiftInSign(skip, tol - __builtin_fabsf(okMed3(sp, s0, 0.F)));
            skip             = okShiftInSign(skip, tol - __builtin_fabsf(okMed3(s0, s1, 0.F)));
            skip             = okShiftInSign(skip, tol - __builtin_fabsf(okMed3(s1, s2, 0.F)));
            skip             = okShiftInSign(skip, tol - __builtin_fabsf(okMed3(s2, s3, 0.F)));
            skip             = okShiftInSign(skip, tol - __builtin_fabsf(okMed3(s3, s4, 0.F)));
            skip             = okShiftInSign(skip, tol - __builtin_fabsf(okMed3(s4, s5, 0.F)));
            skip             = okShiftInSign(skip, tol - __builtin_fabsf(okMed3(s5, s6, 0.F)));
            skip             = okShiftInSign(skip, tol - __builtin_fabsf(okMed3(s6, s7, 0.F)));
            sp               = s7;
        }
        // Exact tests of the surviving segments, all lanes of a wave together: the long test stays out of the point
        // loop, and a wave runs it max-over-lanes(survivors) times per chunk (typically 2: the inner and the outer
        // boundary).
        uint32_t       cand = ~(skip | hc.brk);
        const uint32_t top  = k0 + ((n + 7U) & ~7U) - 2U; // bit b <-> pair (slot top - b, slot top - b + 1)
        OKRC_PROF(2);
        while (cand != 0U)
        { // highest bit first = ascending slots, the order of the reference's sweep inside a run (it decides which of
          // two hits at t = +0 / -0 is kept)
            const uint32_t z = static_cast<uint32_t>(__builtin_clz(cand));
            cand &= ~(0x80000000U >> z);
            if (kCount)
                *tests += 1;
            min_t = okExactSlot<kAmb>(v, top - (31U - z), ox, oy, rdx, rdy, min_t, &amb);
        }
        OKRC_PROF(3);
        if (((hc.w0 >> (OKPOLY_IDX_BITS + 6)) & 1U) != 0U)
        { // the cell continues in another chunk: its header sits right behind this chunk's slots
            hc = *reinterpret_cast<const OkCellHdr *>(&v.slots[k0 + n]);
            continue;
        }
        // leave the cell
        leave = (min_t <= t_exit) ? 1 : leave; // first hit inside the covered part
        if (leave != 0 || --guard <= 0)
            break;
        cell   = cell_n;
        tmax_x = tnx;
        tmax_y = tny;
        hc     = h_next;
        OKRC_ENTER_CELL();
        if (kCount)
            *cells += 1;
        OKRC_PROF(4);
    }
#undef OKRC_ENTER_CELL
    // leave == 0 only if the guard ran out, which cannot happen; treat it like leaving the grid
    const float t_reached  = (leave == 1 || leave == 2) ? t_exit : OK_SENSOR_RANGE;
    const bool  conclusive = leave != 2;
    return {min_t, t_reached, conclusive, kAmb && amb <= v.e_s};
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
