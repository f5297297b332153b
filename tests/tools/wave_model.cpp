// wave_model.cpp -- development aid: lock-step (SIMD) trip-count model of the cooperative step's raycast.
//
// Replays the two-phase walk of okStepCoopKernel for waves of 64 rays on the CPU with the product's own traversal
// header, records per lane and per loop iteration what the lane did, and reduces that to what a 64-lane wave executes
// when all lanes run the loops in lock step (trip count of a loop = max over its active lanes).  Output: per wave-step
// averages of cell iterations, chunk iterations, point-pair iterations, exact-test iterations, and the lane utilisation
// of each, for a given cell size / phase-1 range / split width.  Used to rank design alternatives without GPU time.
//
// build: g++ -O2 -std=c++17 -ffp-contract=off -shared -fPIC -I include -I openkitchen_amd/csrc -o tools/_build/libwavemodel.so tools/wave_model.cpp
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../openkitchen_amd/csrc/ok_grid.h"

namespace
{
double g_cls[5] = {0, 0, 0, 0, 0};
int    g_front_only = 0; // walk the FRONT image of the front / back split (ok_grid.h) instead of the combined one
float  g_predict_margin = 0.F; // > 0: phase 2 cuts [t_reached, last step's hit distance + margin] instead of [t_reached, range];
                               // rays that do not end there get a second round over the rest (mock-up of a temporal prediction)
int    g_no_early_stop = 0; // mock-up of deferred exact tests: a walk does not end on a hit inside the covered part (its exact tests are
                           // still pending then), only at the end of its interval, of the range or of the grid
struct ChunkRec
{
    uint16_t pairs;  // point-pair iterations of the chunk
    uint16_t exact;  // exact tests run after it
    uint16_t slots;  // slots of the chunk
};
struct CellRec
{
    std::vector<ChunkRec> chunks;
};
struct LaneTrace
{
    std::vector<CellRec> cells;
    float                min_t      = OK_SENSOR_RANGE;
    float                t_reached  = 0.F;
    bool                 conclusive = true;
};

// ok_cast_poly_interval with recording (same control flow)
// skip_cells / max_cells: the cell-task form of phase 2 -- start the walk at t_a, step over `skip_cells` cells without looking
// at them, process at most `max_cells` cells (t_reached = the last one's exit; conclusive as usual or when the walk ended)
LaneTrace traceInterval(const OkPolyView &v, float ox, float oy, float rdx, float rdy, float t_a, float t_b, int pair_batch,
                        int skip_cells = 0, int max_cells = 1 << 30, bool skip_unowned_start = false)
{
    LaneTrace         tr;
    const OkGridGeom &g = v.g;
    OkWalk            w;
    if (!w.init(g, ox, oy, rdx, rdy, t_a))
        return tr;
    if (skip_unowned_start)
    { // the shipped rule (ok_cast_poly_interval): the start cell is the previous walk's when the ray entered it before t_a
        const float px = (w.tdel_x < OKRC_INF) ? w.tmax_x - w.tdel_x : -OKRC_INF, py = (w.tdel_y < OKRC_INF) ? w.tmax_y - w.tdel_y : -OKRC_INF;
        if (std::max(px, py) < t_a - 1.0e-2F)
        {
            const float te = w.exitT();
            if (w.t_out <= te || !w.advance(g))
            {
                tr.t_reached  = OK_SENSOR_RANGE;
                tr.conclusive = true;
                return tr;
            }
            if (te >= t_b)
            {
                tr.t_reached  = te;
                tr.conclusive = false;
                return tr;
            }
        }
    }
    for (int k = 0; k < skip_cells; ++k)
    {
        if (w.t_out <= w.exitT() || !w.advance(g))
        { // the range or the grid ends before this lane's cell: nothing to do, the ray is covered to the end
            tr.t_reached  = OK_SENSOR_RANGE;
            tr.conclusive = true;
            return tr;
        }
    }
    const float tol   = v.side_tol;
    float       min_t = OK_SENSOR_RANGE;
    OkCellHdr   h     = v.hdr[w.iy * g.nx + w.ix];
    for (int guard = g.nx + g.ny + 2; guard > 0; --guard)
    {
        CellRec   cr;
        OkCellHdr hc = h;
        while (true)
        {
            const uint32_t k0 = hc.w0 & OKPOLY_IDX_MASK;
            const uint32_t n  = (hc.w0 >> OKPOLY_IDX_BITS) & OKPOLY_N_MASK;
            uint32_t       skip = 0U;
            float          sp   = 0.F;
            const float    c_ray = __builtin_fmaf(ox, rdy, -(oy * rdx));
            for (uint32_t i = 0; i < n; i += 8)
                for (uint32_t j = 0; j < 8; ++j)
                {
                    const OkPoint pt = v.slots[k0 + i + j];
                    const float   sj = __builtin_fmaf(pt.x, rdy, __builtin_fmaf(pt.y, -rdx, -c_ray));
                    skip             = okShiftInSign(skip, tol - __builtin_fabsf(okMed3(sp, sj, 0.F)));
                    sp               = sj;
                }
            uint32_t       cand = ~(skip | hc.brk);
            const uint32_t top  = k0 + ((n + 7U) & ~7U) - 2U;
            ChunkRec c{static_cast<uint16_t>((n + 8 * pair_batch - 1) / (8 * pair_batch)), static_cast<uint16_t>(__builtin_popcount(cand)),
                       static_cast<uint16_t>(n)};
            while (cand != 0U)
            {
                const uint32_t z = static_cast<uint32_t>(__builtin_clz(cand));
                cand &= ~(0x80000000U >> z);
                const uint32_t k = top - (31U - z);
                // classify (statistics only)
                {
                    const OkPoint a = v.slots[k], b = v.slots[k + 1];
                    const float sdx = b.x - a.x, sdy = b.y - a.y, den = rdx * sdy - rdy * sdx;
                    const float tt = ((a.x - ox) * sdy - (a.y - oy) * sdx) / den, ss = ((a.x - ox) * rdy - (a.y - oy) * rdx) / den;
                    int cls = 0; // 0 accepted-range (0<=t<=min_t, s ok), 1 behind (t<0), 2 beyond current hit, 3 s outside, 4 parallel
                    if (__builtin_fabsf(den) < OK_PARALLEL_EPS) cls = 4;
                    else if (tt < 0.F) cls = 1;
                    else if (tt > min_t) cls = 2;
                    else if (!(ss >= 0.F && ss <= 1.F)) cls = 3;
                    g_cls[cls] += 1;
                }
                min_t = okExactSlot(v, k, ox, oy, rdx, rdy, min_t);
            }
            cr.chunks.push_back(c);
            if (((hc.w0 >> (OKPOLY_IDX_BITS + 6)) & 1U) == 0U)
                break;
            hc = *reinterpret_cast<const OkCellHdr *>(&v.slots[k0 + n]);
        }
        tr.cells.push_back(cr);
        const float t_exit = w.exitT();
        tr.min_t           = min_t;
        if (__builtin_fminf(g_no_early_stop ? OKRC_INF : min_t, w.t_out) <= t_exit)
        {
            tr.t_reached  = t_exit;
            tr.conclusive = true;
            return tr;
        }
        if (t_exit >= t_b || --max_cells <= 0)
        {
            tr.t_reached  = t_exit;
            tr.conclusive = min_t <= t_exit; // (deferred tests: known after the flush at the end of the walk)
            return tr;
        }
        if (!w.advance(g))
        {
            tr.t_reached  = OK_SENSOR_RANGE;
            tr.conclusive = true;
            return tr;
        }
        h = v.hdr[w.iy * g.nx + w.ix];
    }
    tr.conclusive = true;
    return tr;
}

int g_p1_max_cells = 0; // phase 1 ends after this many cells at the latest (0: at its range only)
int g_cells_goal = 0; // cell tasks: cells a round should cover per ray (0: one cell per lane)
int g_p2_mode = 0; // 0: equal parameter intervals with the start-cell ownership rule (shipped), 2: without it (round 2), 1: cell tasks (lane j of a ray takes the j-th cell after t_reached), rounds until done
double g_p2_rounds = 0, g_second_round_rays = 0, g_second_round_waves = 0;

struct WaveCount
{
    double cell_it = 0, chunk_it = 0, pair_it = 0, exact_it = 0;             // lock-step iterations
    double half_it = 0;                                                      // point loop in rounds of FOUR slots
    double pooled_it = 0;                                                    // exact loop if a chunk's candidates were pooled over the wave's lanes
    double cell_lanes = 0, chunk_lanes = 0, pair_lanes = 0, exact_lanes = 0; // active lane-iterations
    // exact loop with DEFERRED tests: a lane parks a chunk's candidates (queue of 1 or 2 chunks) and tests them when the queue is
    // full and another chunk with candidates arrives, or when its walk ends; a flush is a lock-step loop over the flushing lanes
    double defer1_it = 0, defer2_it = 0;
    void   addDeferred(const std::vector<LaneTrace> &lanes)
    {
        for (int depth = 1; depth <= 2; ++depth)
        {
            double                        &acc = depth == 1 ? defer1_it : defer2_it;
            std::vector<std::vector<int>>  q(lanes.size());
            size_t max_cells = 0;
            for (auto &l : lanes)
                max_cells = std::max(max_cells, l.cells.size());
            for (size_t c = 0; c < max_cells; ++c)
            {
                size_t max_chunks = 0;
                for (auto &l : lanes)
                    if (c < l.cells.size())
                        max_chunks = std::max(max_chunks, l.cells[c].chunks.size());
                for (size_t k = 0; k < max_chunks; ++k)
                {
                    int flush = 0;
                    for (size_t i = 0; i < lanes.size(); ++i)
                    {
                        const auto &l = lanes[i];
                        if (!(c < l.cells.size() && k < l.cells[c].chunks.size()) || l.cells[c].chunks[k].exact == 0)
                            continue;
                        if (static_cast<int>(q[i].size()) == depth)
                        { // queue full: the oldest chunk's candidates are tested now
                            flush = std::max(flush, q[i].front());
                            q[i].erase(q[i].begin());
                        }
                        q[i].push_back(l.cells[c].chunks[k].exact);
                    }
                    acc += flush;
                }
                // lanes whose walk ends with this cell flush what they hold (they have to report); the others' queues stay
                int flush_end = 0;
                for (size_t i = 0; i < lanes.size(); ++i)
                    if (c + 1 == lanes[i].cells.size())
                    {
                        int tot = 0;
                        for (int e : q[i])
                            tot += e;
                        flush_end = std::max(flush_end, tot);
                        q[i].clear();
                    }
                // (ends of different lanes fall into different cell iterations; a lane that ended waits, so the flushes of all lanes
                // that end within the same iteration run together -- and a smarter kernel would hold them until the LAST lane ends)
                acc += flush_end;
            }
        }
    }
    // the same with every lane holding everything until the wave's last lane has ended: one flush per walk phase
    double defer_all_it = 0;
    void   addDeferredAll(const std::vector<LaneTrace> &lanes)
    {
        int m = 0;
        for (auto &l : lanes)
        {
            int tot = 0;
            for (auto &c : l.cells)
                for (auto &k : c.chunks)
                    tot += k.exact;
            m = std::max(m, tot);
        }
        defer_all_it += m;
    }
    void   add(const std::vector<LaneTrace> &lanes)
    {
        addDeferred(lanes);
        addDeferredAll(lanes);
        size_t max_cells = 0;
        for (auto &l : lanes)
            max_cells = std::max(max_cells, l.cells.size());
        for (size_t c = 0; c < max_cells; ++c)
        {
            cell_it += 1;
            size_t max_chunks = 0;
            for (auto &l : lanes)
                if (c < l.cells.size())
                {
                    cell_lanes += 1;
                    max_chunks = std::max(max_chunks, l.cells[c].chunks.size());
                }
            for (size_t k = 0; k < max_chunks; ++k)
            {
                chunk_it += 1;
                int mp = 0, me = 0, mh = 0;
                for (auto &l : lanes)
                    if (c < l.cells.size() && k < l.cells[c].chunks.size())
                    {
                        chunk_lanes += 1;
                        mp = std::max<int>(mp, l.cells[c].chunks[k].pairs);
                        mh = std::max<int>(mh, (l.cells[c].chunks[k].slots + 3) / 4);
                        me = std::max<int>(me, l.cells[c].chunks[k].exact);
                        pair_lanes += l.cells[c].chunks[k].pairs;
                        exact_lanes += l.cells[c].chunks[k].exact;
                    }
                pair_it += mp;
                half_it += mh;
                exact_it += me;
                {
                    int tot = 0;
                    for (auto &l : lanes)
                        if (c < l.cells.size() && k < l.cells[c].chunks.size())
                            tot += l.cells[c].chunks[k].exact;
                    pooled_it += (tot + 63) / 64;
                }
            }
        }
    }
};
} // namespace

extern "C" __attribute__((visibility("default"))) void wavemodel_set_options(int front_only, int no_early_stop)
{
    g_front_only    = front_only;
    g_no_early_stop = no_early_stop;
}

extern "C" __attribute__((visibility("default"))) void wavemodel_set_predict(float margin)
{
    g_predict_margin = margin;
}

extern "C" __attribute__((visibility("default"))) void wavemodel_set_p2_mode(int mode, int cells_goal, int p1_max_cells)
{
    g_p2_mode      = mode;
    g_cells_goal   = cells_goal;
    g_p1_max_cells = p1_max_cells;
}

extern "C" __attribute__((visibility("default"))) int wavemodel_run(const float *segs_xyxy,
                                                                     int          S,
                                                                     float        cell,
                                                                     const float *pos_x,
                                                                     const float *pos_y,
                                                                     const float *rot_deg,
                                                                     int          n_agents,
                                                                     const float *ray_deg,
                                                                     int          R, // <= 64, one agent per wave (G = 64)
                                                                     float        phase1_range,
                                                                     int          max_split,
                                                                     int          pair_batch,
                                                                     double      *out /* 20 doubles */)
{
    const OkSeg *segs = reinterpret_cast<const OkSeg *>(segs_xyxy);
    OkGridHost   gh   = okBuildGrid(segs, static_cast<size_t>(S), cell);
    OkPolyImage  img  = okBuildPolyImage(segs, static_cast<size_t>(S), gh);
    if (!img.ok)
        return -1;
    OkPolyView pv{};
    pv.g        = gh.g;
    pv.slots    = reinterpret_cast<const OkPoint *>(img.bytes.data());
    pv.hdr      = reinterpret_cast<const OkCellHdr *>(img.bytes.data() + img.off_hdr);
    pv.side_tol = img.side_tol;
    OkFrontBack       fbc;
    OkFrontBackImages fbi;
    if (g_front_only)
    {
        fbc = okClassifyFrontBack(segs, static_cast<size_t>(S), gh, img.max_seg_len);
        fbi = okBuildFrontBackImages(segs, static_cast<size_t>(S), gh, fbc);
        if (!fbi.ok)
            return -2;
        pv.g        = fbi.grid_front.g;
        pv.slots    = reinterpret_cast<const OkPoint *>(fbi.front.bytes.data());
        pv.hdr      = reinterpret_cast<const OkCellHdr *>(fbi.front.bytes.data() + fbi.front.off_hdr);
        pv.side_tol = fbi.front.side_tol;
    }
    WaveCount p1, p2;
    for (double &x : g_cls) x = 0;
    double    n_pending = 0, waves_with_p2 = 0;
    for (int a = 0; a < n_agents; ++a)
    {
        float sr, cr;
        ok_sincosf(OK_DEG2RAD * rot_deg[a], &sr, &cr);
        const float            ox = pos_x[a], oy = pos_y[a];
        std::vector<LaneTrace> l1(64);
        std::vector<float>     dx(64), dy(64);
        std::vector<int>       pend;
        for (int r = 0; r < R; ++r)
        {
            ok_sincosf(OK_DEG2RAD * (rot_deg[a] + ray_deg[r]), &dy[r], &dx[r]);
            l1[r] = traceInterval(pv, ox, oy, dx[r], dy[r], 0.F, phase1_range, pair_batch, 0, g_p1_max_cells > 0 ? g_p1_max_cells : (1 << 30));
            if (!l1[r].conclusive)
                pend.push_back(r);
        }
        p1.add(l1);
        if (!pend.empty() && g_p2_mode == 1)
        {
            waves_with_p2 += 1;
            n_pending += pend.size();
            std::vector<float> t0(64), mt(64);
            std::vector<int>   done(64, 0);
            for (int r : pend)
            {
                t0[r] = l1[r].t_reached;
                mt[r] = l1[r].min_t;
            }
            while (!pend.empty())
            {
                g_p2_rounds += 1;
                const int n = static_cast<int>(pend.size());
                int       m = 64 / n;
                m           = m > max_split ? max_split : m;
                const int c = g_cells_goal > 0 ? (g_cells_goal + m - 1) / m : 1; // consecutive cells per lane
                std::vector<LaneTrace> l2;
                std::vector<int>       next;
                for (int q = 0; q < n; ++q)
                {
                    const int r       = pend[q];
                    float     reached = t0[r];
                    bool      ended   = false;
                    for (int j = 0; j < m; ++j)
                    {
                        LaneTrace t = traceInterval(pv, ox, oy, dx[r], dy[r], t0[r], OKRC_INF, pair_batch, done[r] + j * c, c);
                        mt[r]       = std::min(mt[r], t.min_t);
                        if (t.conclusive && t.t_reached >= OK_SENSOR_RANGE)
                            ended = true;
                        reached = std::max(reached, t.t_reached);
                        l2.push_back(std::move(t));
                    }
                    // (a walk restarted at a cell's exit parameter may land in the cell just left: one cell of overlap per round)
                    done[r] += m * c;
                    if (!(ended || mt[r] <= reached))
                        next.push_back(r);
                }
                p2.add(l2);
                pend.swap(next);
            }
        }
        else if (!pend.empty() && g_predict_margin > 0.F)
        {
            waves_with_p2 += 1;
            n_pending += pend.size();
            // last step's distances: the same fan from a pose 1.2 px behind along the heading, turned by 2 degrees (what a step of
            // the bench recipe changes at most)
            float srp, crp;
            ok_sincosf(OK_DEG2RAD * rot_deg[a], &srp, &crp);
            const float pox = ox - 1.2F * crp, poy = oy - 1.2F * srp;
            const int   n = static_cast<int>(pend.size());
            int         m = 64 / n;
            m             = m > max_split ? max_split : m;
            std::vector<LaneTrace> l2;
            std::vector<int>       next;
            std::vector<float>     from(64, 0.F), best(64, OK_SENSOR_RANGE);
            for (int q = 0; q < n; ++q)
            {
                const int r = pend[q];
                float     pdx, pdy;
                ok_sincosf(OK_DEG2RAD * (rot_deg[a] - 2.0F + ray_deg[r]), &pdy, &pdx);
                const float d_prev = ok_cast_ray_poly<false>(pv, pox, poy, pdx, pdy, nullptr, nullptr, nullptr);
                const float t0     = l1[r].t_reached;
                const bool  open   = !(d_prev < OK_SENSOR_RANGE);
                const float D      = open ? OK_SENSOR_RANGE : std::max(d_prev + g_predict_margin, t0 + 8.0F);
                const float dt     = (D - t0) / static_cast<float>(m);
                float       reached = t0;
                bool        ended   = false;
                best[r]             = l1[r].min_t;
                for (int j = 0; j < m; ++j)
                {
                    const float ta = t0 + static_cast<float>(j) * dt;
                    const float tb = (j + 1 == m && open) ? OKRC_INF : t0 + static_cast<float>(j + 1) * dt;
                    LaneTrace   t  = traceInterval(pv, ox, oy, dx[r], dy[r], ta, tb, pair_batch, 0, 1 << 30, true);
                    best[r]        = std::min(best[r], t.min_t);
                    if (t.conclusive && t.t_reached >= OK_SENSOR_RANGE)
                        ended = true;
                    reached = std::max(reached, t.t_reached);
                    l2.push_back(std::move(t));
                }
                if (!(ended || best[r] <= reached))
                {
                    next.push_back(r);
                    from[r] = reached;
                }
            }
            p2.add(l2);
            g_p2_rounds += 1;
            if (!next.empty())
            { // second round: the rest of the rays whose hit was not where it had been, cut like today's phase 2
                g_p2_rounds += 1;
                const int n2 = static_cast<int>(next.size());
                int       m2 = 64 / n2;
                m2           = m2 > max_split ? max_split : m2;
                std::vector<LaneTrace> l3;
                for (int q = 0; q < n2; ++q)
                {
                    const int   r  = next[q];
                    const float t0 = from[r];
                    const float dt = (OK_SENSOR_RANGE - t0) / static_cast<float>(m2);
                    for (int j = 0; j < m2; ++j)
                    {
                        const float ta = t0 + static_cast<float>(j) * dt;
                        const float tb = (j + 1 == m2) ? OKRC_INF : t0 + static_cast<float>(j + 1) * dt;
                        l3.push_back(traceInterval(pv, ox, oy, dx[r], dy[r], ta, tb, pair_batch, 0, 1 << 30, true));
                    }
                }
                p2.add(l3);
                g_second_round_rays += n2;
                g_second_round_waves += 1;
            }
        }
        else if (!pend.empty())
        {
            waves_with_p2 += 1;
            n_pending += pend.size();
            const int n = static_cast<int>(pend.size());
            int       m = 64 / n;
            m           = m > max_split ? max_split : m;
            std::vector<LaneTrace> l2;
            for (int lane = 0; lane < 64; ++lane)
            {
                const int q = lane / m, j = lane - q * m;
                if (q >= n)
                    continue;
                const int   r  = pend[q];
                const float t0 = l1[r].t_reached;
                const float dt = (OK_SENSOR_RANGE - t0) / static_cast<float>(m);
                const float ta = t0 + static_cast<float>(j) * dt;
                const float tb = (j + 1 == m) ? OKRC_INF : t0 + static_cast<float>(j + 1) * dt;
                l2.push_back(traceInterval(pv, ox, oy, dx[r], dy[r], ta, tb, pair_batch, 0, 1 << 30, g_p2_mode == 0));
            }
            p2.add(l2);
        }
    }
    const double W = n_agents;
    double      *o = out;
    *o++ = p1.cell_it / W;  *o++ = p1.chunk_it / W;  *o++ = p1.pair_it / W;  *o++ = p1.exact_it / W;
    *o++ = p1.cell_lanes / W;  *o++ = p1.chunk_lanes / W;  *o++ = p1.pair_lanes / W;  *o++ = p1.exact_lanes / W;
    *o++ = p2.cell_it / W;  *o++ = p2.chunk_it / W;  *o++ = p2.pair_it / W;  *o++ = p2.exact_it / W;
    *o++ = p2.cell_lanes / W;  *o++ = p2.chunk_lanes / W;  *o++ = p2.pair_lanes / W;  *o++ = p2.exact_lanes / W;
    *o++ = n_pending / W;  *o++ = waves_with_p2 / W;  *o++ = static_cast<double>(img.bytes.size());  *o++ = img.max_slots_per_cell;
    std::printf("    point loop: %.2f + %.2f rounds of 8 slots per wave-step; in rounds of 4 slots: %.2f + %.2f (= %.2f + %.2f of 8)\n", p1.pair_it / W, p2.pair_it / W,
                p1.half_it / W, p2.half_it / W, p1.half_it / W / 2, p2.half_it / W / 2);
    std::printf("    exact loop if a chunk's candidates were pooled over the 64 lanes: %.2f + %.2f passes (now %.2f + %.2f)\n", p1.pooled_it / W, p2.pooled_it / W,
                p1.exact_it / W, p2.exact_it / W);
    std::printf("    exact loop with deferred tests (%s early stop): queue of one chunk %.2f + %.2f, of two %.2f + %.2f, one flush per phase %.2f + %.2f passes\n",
                g_no_early_stop ? "walks WITHOUT" : "walks with", p1.defer1_it / W, p2.defer1_it / W, p1.defer2_it / W, p2.defer2_it / W, p1.defer_all_it / W, p2.defer_all_it / W);
    if (g_predict_margin > 0.F)
        std::printf("    prediction (margin %g px): waves with a second round %.3f, rays in it per wave-step %.2f\n", g_predict_margin, g_second_round_waves / W,
                    g_second_round_rays / W);
    g_second_round_rays = g_second_round_waves = 0;
    if (g_p2_mode == 1)
        std::printf("    phase-2 rounds per wave-step: %.2f\n", g_p2_rounds / W);
    g_p2_rounds = 0;
    std::printf("    exact tests per wave-step by outcome: accepted %.1f behind %.1f beyond-hit %.1f s-outside %.1f parallel %.1f\n", g_cls[0] / W, g_cls[1] / W, g_cls[2] / W, g_cls[3] / W, g_cls[4] / W);
    return 0;
}

// The tail kernel's arrangement (okStepTailKernel): one agent per workgroup, `split` equal pieces per ray with the ownership rule,
// eight rays (consecutive, or strided over the waves) per wave.  Prints, per agent-step, the lock-step counts of the mean wave and
// of the critical wave (the one with the largest modelled cost: the step waits for it at the barrier).
extern "C" __attribute__((visibility("default"))) int wavemodel_tail(const float *segs_xyxy, int S, float cell, const float *pos_x, const float *pos_y,
                                                                      const float *rot_deg, int n_agents, const float *ray_deg, int R, int split,
                                                                      int strided, int slot_lanes)
{
    const OkSeg *segs = reinterpret_cast<const OkSeg *>(segs_xyxy);
    OkGridHost   gh   = okBuildGrid(segs, static_cast<size_t>(S), cell);
    OkPolyImage  img  = okBuildPolyImage(segs, static_cast<size_t>(S), gh);
    if (!img.ok)
        return -1;
    OkPolyView pv{};
    pv.g        = gh.g;
    pv.slots    = reinterpret_cast<const OkPoint *>(img.bytes.data());
    pv.hdr      = reinterpret_cast<const OkCellHdr *>(img.bytes.data() + img.off_hdr);
    pv.side_tol = img.side_tol;
    const int rays_per_wave = 64 / split, n_waves = (R + rays_per_wave - 1) / rays_per_wave;
    auto cost = [&](const WaveCount &w) { return 150.0 * w.cell_it + 300.0 * (slot_lanes > 1 ? w.half_it * 4.0 / (8.0 * slot_lanes) * 2.0 : w.pair_it) + 450.0 * w.exact_it; };
    WaveCount mean_w, crit_w;
    double    mean_cost = 0, crit_cost = 0, first_cell_exact = 0, first_cell_pairs = 0;
    for (double &x : g_cls) x = 0;
    for (int a = 0; a < n_agents; ++a)
    {
        const float ox = pos_x[a], oy = pos_y[a];
        WaveCount   worst;
        double      worst_cost = -1;
        for (int w = 0; w < n_waves; ++w)
        {
            std::vector<LaneTrace> lanes;
            for (int i = 0; i < rays_per_wave; ++i)
            {
                const int r = strided ? w + i * n_waves : w * rays_per_wave + i;
                if (r >= R)
                    continue;
                float dx, dy;
                ok_sincosf(OK_DEG2RAD * (rot_deg[a] + ray_deg[r]), &dy, &dx);
                const float dt = OK_SENSOR_RANGE / static_cast<float>(split);
                for (int j = 0; j < split; ++j)
                {
                    const float ta = static_cast<float>(j) * dt, tb = (j + 1 == split) ? OKRC_INF : static_cast<float>(j + 1) * dt;
                    lanes.push_back(traceInterval(pv, ox, oy, dx, dy, ta, tb, 1, 0, 1 << 30, j > 0));
                    if (j == 0 && !lanes.back().cells.empty())
                        for (auto &c : lanes.back().cells[0].chunks)
                        {
                            first_cell_exact += c.exact;
                            first_cell_pairs += c.pairs;
                        }
                }
            }
            WaveCount wc;
            wc.add(lanes);
            mean_w.cell_it += wc.cell_it; mean_w.pair_it += wc.pair_it; mean_w.exact_it += wc.exact_it; mean_w.half_it += wc.half_it;
            mean_cost += cost(wc);
            if (cost(wc) > worst_cost)
            {
                worst_cost = cost(wc);
                worst      = wc;
            }
        }
        crit_w.cell_it += worst.cell_it; crit_w.pair_it += worst.pair_it; crit_w.exact_it += worst.exact_it; crit_w.half_it += worst.half_it;
        crit_cost += worst_cost;
    }
    const double A = n_agents, W = A * n_waves;
    std::printf("tail model cell %g split %d %s%s: mean wave: cells %.2f passes %.2f exact %.2f cost %.0f | critical wave: cells %.2f passes %.2f exact %.2f cost %.0f"
                " | origin cell per ray: passes %.2f exact %.2f\n",
                cell, split, strided ? "strided" : "consecutive", slot_lanes > 1 ? " slot-parallel" : "", mean_w.cell_it / W, mean_w.pair_it / W, mean_w.exact_it / W,
                mean_cost / W, crit_w.cell_it / A, crit_w.pair_it / A, crit_w.exact_it / A, crit_cost / A, first_cell_pairs / (A * R), first_cell_exact / (A * R));
    std::printf("    exact tests per agent-step by outcome: accepted %.1f behind %.1f beyond-hit %.1f s-outside %.1f parallel %.1f\n", g_cls[0] / A, g_cls[1] / A, g_cls[2] / A,
                g_cls[3] / A, g_cls[4] / A);
    return 0;
}
