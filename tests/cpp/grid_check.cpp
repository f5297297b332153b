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
}
