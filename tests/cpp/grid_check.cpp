// grid_check.cpp -- host-side driver of the PRODUCT's traversal header (openkitchen_amd/csrc/ok_raycast.h,
// ok_grid.h) for CPU tests: casts arbitrary rays through the uniform grid and reports first-hit t plus
// work counters, so tests/test_grid_traversal.py can compare against the oracle's brute-force sweep for
// millions of rays without a GPU.  Test infrastructure: built by the test into tests/cpp/_build/.
#include <cstdint>
#include <vector>

#include "../../openkitchen_amd/csrc/ok_grid.h"

extern "C"
{
    // returns 0, fills out_t[n]; tests/cells may be null
    __attribute__((visibility("default"))) int gridcheck_cast(const float *segs_xyxy,
                                                              int          S,
                                                              float        cell,
                                                              const float *ox,
                                                              const float *oy,
                                                              const float *angle_rad,
                                                              int          n,
                                                              float       *out_t,
                                                              uint32_t    *out_tests,
                                                              uint32_t    *out_cells,
                                                              int32_t     *info /* nx, ny, nref, max_count, image bytes, runs, points */,
                                                              int          form /* 0 wide (exact test on every registered segment), 1 compact poly */,
                                                              uint32_t    *out_points)
    {
        const OkSeg *segs = reinterpret_cast<const OkSeg *>(segs_xyxy);
        OkGridHost   gh   = okBuildGrid(segs, static_cast<size_t>(S), cell);
        OkGridView32 v{};
        v.g     = gh.g;
        v.segs  = segs;
        v.refs  = gh.refs.data();
        v.start = gh.start.data();
        OkPolyImage img = okBuildPolyImage(segs, static_cast<size_t>(S), gh);
        OkPolyView  pv{};
        if (form >= 1)
        {
            if (!img.ok)
                return -1;
            pv.g        = gh.g;
            pv.slots    = reinterpret_cast<const OkPoint *>(img.bytes.data());
            pv.hdr      = reinterpret_cast<const OkCellHdr *>(img.bytes.data() + img.off_hdr);
            pv.side_tol = img.side_tol;
        }
        if (info)
        {
            info[0] = gh.g.nx;
            info[1] = gh.g.ny;
            info[2] = static_cast<int32_t>(gh.refs.size());
            info[3] = static_cast<int32_t>(gh.max_count);
            info[4] = static_cast<int32_t>(img.bytes.size());
            info[5] = static_cast<int32_t>(img.num_runs);
            info[6] = static_cast<int32_t>(img.num_slots);
        }
        for (int i = 0; i < n; ++i)
        {
            float s, c;
            ok_sincosf(angle_rad[i], &s, &c);
            uint32_t tests = 0, cells = 0, points = 0;
            if (form == 1)
            {
                out_t[i] = ok_cast_ray_poly<true>(pv, ox[i], oy[i], c, s, &tests, &cells, &points);
            }
            else if (form >= 2)
            { // the kernels' decompositions of a ray:
              //   form 2..8    the cooperative kernel: [0, T1], then m = form equal sub-intervals of what is left, every one of them
              //                stepping over its start cell unless the ray enters it inside the interval (skip_unowned_start)
              //   form 12..18  direct dealing / tail kernel: m = form - 10 equal intervals from the origin, all but the first stepping
              //                over an unowned start cell
              //   form 20 + m  m equal intervals from the origin, every one processing every cell it touches (the plain contract)
                const bool plain = form >= 20, from_origin = form >= 10;
                const int  m     = plain ? form - 20 : (from_origin ? form - 10 : form);
                float      best = OK_SENSOR_RANGE, t0 = 0.0F;
                bool       open = true;
                if (!from_origin)
                {
                    OkIntervalResult r1 = ok_cast_poly_interval<true>(pv, ox[i], oy[i], c, s, 0.0F, 48.0F, &tests, &cells, &points);
                    best = r1.min_t;
                    open = !r1.conclusive;
                    t0   = r1.t_reached;
                }
                if (open)
                {
                    const float dt = (OK_SENSOR_RANGE - t0) / (float)m;
                    for (int j = 0; j < m; ++j)
                    {
                        const float ta   = t0 + (float)j * dt;
                        const float tb   = (j == m - 1) ? OKRC_INF : t0 + (float)(j + 1) * dt;
                        const bool  skip = !plain && (j > 0 || t0 > 0.0F);
                        OkIntervalResult r2 = ok_cast_poly_interval<true>(pv, ox[i], oy[i], c, s, ta, tb, &tests, &cells, &points, nullptr, skip);
                        best = r2.min_t < best ? r2.min_t : best;
                    }
                }
                out_t[i] = best;
            }
            else
                out_t[i] = ok_cast_ray_grid<true>(v, ox[i], oy[i], c, s, &tests, &cells);
            if (out_tests)
                out_tests[i] = tests;
            if (out_cells)
                out_cells[i] = cells;
            if (out_points)
                out_points[i] = points;
        }
        return 0;
    }

    // ok_first_hit_update (division-free common path) next to the reference's sequence ok_ray_segment, element by element:
    // out_new[i] / out_ref[i] = updated first-hit parameter of candidate i
    __attribute__((visibility("default"))) int gridcheck_first_hit_update(int          n,
                                                                          const float *ox,
                                                                          const float *oy,
                                                                          const float *dx,
                                                                          const float *dy,
                                                                          const float *ax,
                                                                          const float *ay,
                                                                          const float *bx,
                                                                          const float *by,
                                                                          const float *min_t,
                                                                          float       *out_new,
                                                                          float       *out_ref)
    {
        for (int i = 0; i < n; ++i)
        {
            out_new[i] = ok_first_hit_update(ox[i], oy[i], dx[i], dy[i], OkPoint{ax[i], ay[i]}, OkPoint{bx[i], by[i]}, min_t[i]);
            float t;
            out_ref[i] = ok_ray_segment(ox[i], oy[i], dx[i], dy[i], ax[i], ay[i], bx[i], by[i], min_t[i], t) ? t : min_t[i];
        }
        return 0;
    }

    // The front / back split (ok_grid.h: okClassifyFrontBack) as the cooperative kernel uses it, ray by ray on the host:
    //   chi(origin) certain (okOriginChiScalar)?  front walk -- whole ray (parts = 1) or the kernel's decomposition ([0, 48], then
    //   `parts` intervals of what is left, start cells owned by one walk) -- reporting ambiguous rejections; rays whose origin is not
    //   certified or whose front walk was ambiguous walk the back image too, starting from the front's first hit.
    // out_flags: bit 0 origin certified, bit 1 front walk ambiguous, bit 2 back image walked.
    // info: [0] split exists, [1] F segments, [2] back segments, [3] other front segments, [4] certifiable cells, [5] cells holding
    //       front segments, [6] front image bytes, [7] back image bytes, [8] combined image bytes, [9] cells
    // force_back != 0: every ray walks the back image (what an uncertified agent costs; also "front + back = all segments")
    __attribute__((visibility("default"))) int gridcheck_cast_fb(const float *segs_xyxy, int S, float cell, const float *ox, const float *oy,
                                                                 const float *angle_rad, int n, float *out_t, uint32_t *out_flags, int32_t *info,
                                                                 int parts, int force_back, uint32_t *out_points)
    {
        const OkSeg *segs = reinterpret_cast<const OkSeg *>(segs_xyxy);
        bool         fits = false;
        OkPolyImage  combined;
        OkGridHost   gh = okBuildGridAuto(segs, static_cast<size_t>(S), cell, 160U * 1024U, &fits, &combined);
        if (!fits)
            return -1;
        OkFrontBack       fb   = okClassifyFrontBack(segs, static_cast<size_t>(S), gh, combined.max_seg_len);
        OkFrontBackImages imgs = okBuildFrontBackImages(segs, static_cast<size_t>(S), gh, fb);
        if (info)
        {
            info[0] = imgs.ok ? 1 : 0;
            info[1] = static_cast<int32_t>(fb.n_f);
            info[2] = static_cast<int32_t>(fb.n_back);
            info[3] = static_cast<int32_t>(fb.n_front_other);
            info[4] = static_cast<int32_t>(fb.n_cells_cert);
            info[5] = static_cast<int32_t>(fb.n_cells_with_front);
            info[6] = static_cast<int32_t>(imgs.front.bytes.size());
            info[7] = static_cast<int32_t>(imgs.back.bytes.size());
            info[8] = static_cast<int32_t>(combined.bytes.size());
            info[9] = static_cast<int32_t>(gh.numCells());
        }
        if (!imgs.ok)
            return 1;
        OkPolyView vf{}, vb{};
        vf.g        = gh.g;
        vf.slots    = reinterpret_cast<const OkPoint *>(imgs.front.bytes.data());
        vf.hdr      = reinterpret_cast<const OkCellHdr *>(imgs.front.bytes.data() + imgs.front.off_hdr);
        vf.side_tol = imgs.front.side_tol;
        vf.e_s      = fb.e_s;
        vf.e_t      = fb.e_t;
        vf.e_s_over_e_t = fb.e_s / fb.e_t;
        vb          = vf;
        vb.slots    = reinterpret_cast<const OkPoint *>(imgs.back.bytes.data());
        vb.hdr      = reinterpret_cast<const OkCellHdr *>(imgs.back.bytes.data() + imgs.back.off_hdr);
        vb.side_tol = imgs.back.side_tol;
        for (int i = 0; i < n; ++i)
        {
            float s, c;
            ok_sincosf(angle_rad[i], &s, &c);
            uint32_t  tests = 0, cells = 0, points = 0;
            const int cert  = okOriginChiScalar(vf, ox[i], oy[i], fb.t12, fb.t34);
            float     best  = OK_SENSOR_RANGE;
            bool      amb   = false;
            if (parts <= 1)
            {
                const OkIntervalResult r = ok_cast_poly_interval<true, true>(vf, ox[i], oy[i], c, s, 0.0F, OKRC_INF, &tests, &cells, &points);
                best                     = r.min_t;
                amb                      = r.amb;
            }
            else
            {
                const OkIntervalResult r1 = ok_cast_poly_interval<true, true>(vf, ox[i], oy[i], c, s, 0.0F, 48.0F, &tests, &cells, &points);
                best                      = r1.min_t;
                amb                       = r1.amb;
                if (!r1.conclusive)
                {
                    const float t0 = r1.t_reached, dt = (OK_SENSOR_RANGE - t0) / (float)parts;
                    for (int j = 0; j < parts; ++j)
                    {
                        const float            ta = t0 + (float)j * dt, tb = (j == parts - 1) ? OKRC_INF : t0 + (float)(j + 1) * dt;
                        const OkIntervalResult r2 =
                            ok_cast_poly_interval<true, true>(vf, ox[i], oy[i], c, s, ta, tb, &tests, &cells, &points, nullptr, j > 0 || t0 > 0.0F);
                        best = r2.min_t < best ? r2.min_t : best;
                        amb  = amb || r2.amb;
                    }
                }
            }
            uint32_t flags = (cert ? 1U : 0U) | (amb ? 2U : 0U);
            if (!cert || amb || force_back)
            { // the back image from the origin to the front's first hit: in one walk, or -- the cooperative kernel's second pass --
              // cut into `parts` intervals of [0, front hit], every walk bounded by the front hit
                const float limit = best;
                if (parts <= 1)
                    best = ok_cast_poly_interval<true, false>(vb, ox[i], oy[i], c, s, 0.0F, OKRC_INF, &tests, &cells, &points, nullptr, false, limit).min_t;
                else
                {
                    const float dt = limit / (float)parts;
                    for (int j = 0; j < parts; ++j)
                    {
                        const float ta = (float)j * dt, tb = (j == parts - 1) ? OKRC_INF : (float)(j + 1) * dt;
                        const float tj = ok_cast_poly_interval<true, false>(vb, ox[i], oy[i], c, s, ta, tb, &tests, &cells, &points, nullptr, j > 0, limit).min_t;
                        best           = tj < best ? tj : best;
                    }
                }
                flags |= 4U;
            }
            out_t[i] = best;
            if (out_flags)
                out_flags[i] = flags;
            if (out_points)
                out_points[i] = points;
        }
        return 0;
    }
}
