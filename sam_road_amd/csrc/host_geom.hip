// Host-side geometry of the scene pipeline (no device code): the greedy radius NMS that turns the fused masks
// into graph points (reference graph_utils.py:572-591, called three times by graph_extraction.py:130-139).
// The reference walks a scipy KDTree from a Python loop (~1 s per CityScale scene, 83 % of the scene latency
// once pass 1 runs at 3300 tiles/s); this is the same algorithm, literally, on a uniform grid.
//
// Semantics kept exactly (candidates arrive ALREADY in the reference's processing order, i.e.
// np.argsort(scores)[::-1] computed by the caller so that numpy's tie order is preserved):
//     kept[:] = True
//     for i in order:  if not kept[i]: continue
//         for every j with |p_j - p_i| <= radius (j == i, earlier and later ones included):  kept[j] = force[j]
//         kept[i] = True
// force[j] = (score_j > 1.0): such a candidate can never be suppressed — with the u8 mask scores of the first two
// calls EVERY candidate is forced, so those calls suppress nothing (a quirk of the reference that is preserved).
#include <climits>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/samroad_hip.h"
#include "kdtree_emul.hpp"

extern "C" int srh_nms_points_host(const int32_t* xy, const uint8_t* force, int64_t n, int32_t radius, uint8_t* kept) try {
    if (n < 0 || radius < 0 || (n > 0 && (!xy || !force || !kept))) return SRH_ERR_BAD_ARG;
    if (n == 0) return 0;
    memset(kept, 1, (size_t)n);
    bool all_forced = true;
    for (int64_t i = 0; i < n && all_forced; ++i) all_forced = force[i] != 0;
    if (all_forced) return 0;                       // kept[j] = force[j] = True for every neighbour: nothing can change
    int32_t minx = xy[0], maxx = xy[0], miny = xy[1], maxy = xy[1];
    for (int64_t i = 1; i < n; ++i) {
        const int32_t x = xy[2 * i], y = xy[2 * i + 1];
        minx = x < minx ? x : minx; maxx = x > maxx ? x : maxx;
        miny = y < miny ? y : miny; maxy = y > maxy ? y : maxy;
    }
    bool none_forced = true;
    for (int64_t i = 0; i < n && none_forced; ++i) none_forced = force[i] == 0;
    const int64_t bw = (int64_t)maxx - minx + 1, bh = (int64_t)maxy - miny + 1;
    if (none_forced && bw * bh <= ((int64_t)1 << 26)) {
        // No forced candidate (the third call of extract_graph_points: priorities 1 / 0): a candidate is suppressed iff an EARLIER
        // kept one lies within the radius — nothing acts backwards, because a later point inside a kept point's ball is itself
        // suppressed before it is visited.  So one pass over the candidates with a per-pixel "inside the ball of a kept point"
        // map: a lookup per candidate, and per KEPT point the rows of its disc as memsets (exact integer half-widths).  142 k
        // candidates / 4.4 k survivors of a CityScale scene: the whole nms_points call 4.4 -> 1.8 ms (tools/prof_points.py).
        // one BIT per pixel (512 KiB for a 2048^2 scene: the candidates come in score order, i.e. at random places, and a byte map of
        // 4 MiB missed the L2 on nearly every lookup)
        const int64_t ws = (bw + 63) / 64;                          // words per row
        std::vector<uint64_t> sup((size_t)(ws * bh), 0);
        std::vector<int32_t> half((size_t)radius + 1);
        for (int32_t dy = 0; dy <= radius; ++dy) {
            int32_t hx = 0;
            while ((int64_t)(hx + 1) * (hx + 1) + (int64_t)dy * dy <= (int64_t)radius * radius) ++hx;
            half[(size_t)dy] = hx;
        }
        for (int64_t i = 0; i < n; ++i) {
            const int64_t x = xy[2 * i] - minx, y = xy[2 * i + 1] - miny;
            if ((sup[(size_t)(y * ws + (x >> 6))] >> (x & 63)) & 1) { kept[i] = 0; continue; }
            for (int32_t dy = -radius; dy <= radius; ++dy) {
                const int64_t yy = y + dy;
                if (yy < 0 || yy >= bh) continue;
                const int32_t hx = half[(size_t)(dy < 0 ? -dy : dy)];
                const int64_t x0 = x - hx < 0 ? 0 : x - hx, x1 = x + hx >= bw ? bw - 1 : x + hx;
                uint64_t* row = sup.data() + yy * ws;
                const int64_t w0 = x0 >> 6, w1 = x1 >> 6;
                const uint64_t m0 = ~(uint64_t)0 << (x0 & 63), m1 = ~(uint64_t)0 >> (63 - (x1 & 63));
                if (w0 == w1) row[w0] |= m0 & m1;
                else {
                    row[w0] |= m0;
                    for (int64_t w = w0 + 1; w < w1; ++w) row[w] = ~(uint64_t)0;
                    row[w1] |= m1;
                }
            }
        }
        return 0;
    }
    const int32_t cell = radius > 0 ? radius : 1;   // a ball of the radius touches at most 3 x 3 cells
    const int64_t gw = (int64_t)(maxx - minx) / cell + 1, gh = (int64_t)(maxy - miny) / cell + 1;
    if (gw * gh > (int64_t)1 << 28) return SRH_ERR_BAD_ARG;
    // counting sort of the candidates by cell (CSR: cell_start, cell_items)
    std::vector<int64_t> cell_start((size_t)(gw * gh) + 1, 0);
    std::vector<int64_t> cell_of((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t c = (int64_t)((xy[2 * i + 1] - miny) / cell) * gw + (xy[2 * i] - minx) / cell;
        cell_of[(size_t)i] = c;
        ++cell_start[(size_t)c + 1];
    }
    for (size_t c = 0; c < (size_t)(gw * gh); ++c) cell_start[c + 1] += cell_start[c];
    std::vector<int64_t> fill(cell_start.begin(), cell_start.end() - 1);
    std::vector<int64_t> items((size_t)n);
    for (int64_t i = 0; i < n; ++i) items[(size_t)fill[(size_t)cell_of[(size_t)i]]++] = i;
    const int64_t r2 = (int64_t)radius * radius;
    for (int64_t i = 0; i < n; ++i) {
        if (!kept[i]) continue;
        const int64_t x = xy[2 * i], y = xy[2 * i + 1];
        const int64_t cx = (x - minx) / cell, cy = (y - miny) / cell;
        for (int64_t yy = cy > 0 ? cy - 1 : 0; yy <= (cy + 1 < gh ? cy + 1 : gh - 1); ++yy)
            for (int64_t xx = cx > 0 ? cx - 1 : 0; xx <= (cx + 1 < gw ? cx + 1 : gw - 1); ++xx) {
                const int64_t c = yy * gw + xx;
                for (int64_t k = cell_start[(size_t)c]; k < cell_start[(size_t)c + 1]; ++k) {
                    const int64_t j = items[(size_t)k];
                    const int64_t dx = xy[2 * j] - x, dy = xy[2 * j + 1] - y;
                    if (dx * dx + dy * dy <= r2) kept[j] = force[j];
                }
            }
        kept[i] = 1;
    }
    return 0;
} catch (...) { return SRH_ERR_HIP; }      // std::bad_alloc / std::system_error (thread limit) must not cross the C ABI

// ---------------------------------------------------------------------------------------------------------------
// Pass-2 query builder for ALL tiles of a scene in one call (reference inferencer.py:148-176, executed per tile from
// Python there: rtree box query + scipy KDTree.query(k = K+1, distance_upper_bound = R)).  For every tile: the points
// inside the CLOSED box [x0,x1] x [y0,y1] in ascending point index (the order chosen HERE: the reference takes whatever
// order rtree.intersection yields and does not sort; its edge set does not depend on that order — pinned under three
// different orders by tests/test_refrun_golden.py — only the order of its edge list does), and for each of them its K nearest OTHER points of the same tile with distance STRICTLY below R (scipy's
// distance_upper_bound is exclusive), ascending by (distance, tile-local index); missing neighbours are -1.
// The neighbour SET is what the reference computes whenever it is unique.  It is not unique when the (K+1)-th and the
// (K+2)-th candidate are equidistant (scipy keeps whichever its heap met first) or when another point coincides with the
// source (scipy may then return the source itself as a neighbour): such source points are flagged in `ambiguous` and answered by
// the restatement of scipy's kd-tree in kdtree_emul.hpp (a tree of their tile's points), element for element.  Elsewhere the order INSIDE a set of equidistant neighbours is scipy-internal;
// nothing downstream depends on it (TopoNet has no positional encoding along the neighbour axis and the edge votes are
// keyed by (source, target)).
// Two-step protocol: srh_pass2_count -> caller allocates -> srh_pass2_fill.  Tiles are processed by worker threads.
// ---------------------------------------------------------------------------------------------------------------
#include <algorithm>
#include <atomic>
#include <thread>

// worker threads that are joined even when the scope is left by an exception (a std::system_error from a failed thread creation
// would otherwise destroy joinable std::threads = std::terminate before the C-ABI catch block is reached)
struct JoiningThreads : std::vector<std::thread> {
    ~JoiningThreads() { for (auto& t : *this) if (t.joinable()) t.join(); }
};

extern "C" int srh_pass2_count(const int64_t* pts, int64_t n, const int32_t* boxes, int32_t n_tiles, int64_t* counts) try {
    if (n < 0 || n_tiles < 0 || (n > 0 && !pts) || (n_tiles > 0 && (!boxes || !counts))) return SRH_ERR_BAD_ARG;
    // the points' rows and columns as two dense histogram prefix sums would not give a BOX count; what is cheap is to bucket the
    // points by row once (ascending y) and let every tile scan only the rows it covers: ~1/4 of the points of a CityScale scene
    std::vector<int64_t> order((size_t)n);
    for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [pts](int64_t a, int64_t b) { return pts[2 * a + 1] < pts[2 * b + 1]; });
    std::vector<int64_t> ys((size_t)n), xs((size_t)n);
    for (int64_t i = 0; i < n; ++i) { ys[(size_t)i] = pts[2 * order[(size_t)i] + 1]; xs[(size_t)i] = pts[2 * order[(size_t)i]]; }
    for (int32_t t = 0; t < n_tiles; ++t) {
        const int64_t x0 = boxes[4 * t], y0 = boxes[4 * t + 1], x1 = boxes[4 * t + 2], y1 = boxes[4 * t + 3];
        const size_t lo = (size_t)(std::lower_bound(ys.begin(), ys.end(), y0) - ys.begin());
        const size_t hi = (size_t)(std::upper_bound(ys.begin(), ys.end(), y1) - ys.begin());
        int64_t c = 0;
        for (size_t i = lo; i < hi; ++i) c += (xs[i] >= x0 && xs[i] <= x1);
        counts[t] = c;
    }
    return 0;
} catch (...) { return SRH_ERR_HIP; }      // std::bad_alloc / std::system_error (thread limit) must not cross the C ABI

// offsets[t] = sum(counts[:t]) (caller);  ids [total];  knn [total, K] tile-local neighbour index or -1;
// ambiguous [total]: per source point, 1 = its neighbours are not determined by distances alone (tie at the cut-off, a
// coincident point) and were decided the way the reference's scipy.spatial.KDTree(tile points).query decides them
// (kdtree_emul.hpp: the same tree, the same traversal, the same heaps), in scipy's output order
extern "C" int srh_pass2_fill(const int64_t* pts, int64_t n, const int32_t* boxes, int32_t n_tiles, int32_t K, int64_t radius,
                              const int64_t* offsets, int64_t* ids, int32_t* knn, uint8_t* ambiguous, int64_t* local,
                              int32_t n_threads) try {
    if (n < 0 || n_tiles < 0 || K <= 0 || radius < 0 || (n_tiles > 0 && (!boxes || !offsets || !ids || !knn || !ambiguous)))
        return SRH_ERR_BAD_ARG;
    const int64_t r2 = radius * radius;
    // ---- once per scene: every point's K+1 nearest OTHER points among ALL points (distance < radius, ascending by (distance,
    // index)) and whether its answer is ambiguous.  The tilings overlap heavily (a CityScale point lies in ~18 tiles) and for a
    // point at least `radius` away from all four edges of a tile every such neighbour is inside the tile too: its row in that
    // tile is the global answer re-indexed — order included, because a tile lists its points by ascending global index.
    // (Ambiguous points still get the per-tile kd-tree treatment below: the tie is broken by the TILE's tree.)
    // The worker threads first share this pre-pass (chunks of points), meet at a barrier, then take tiles one at a time from a
    // shared counter (tiles differ by 3x in points and in kd-tree work: a static split left the slowest thread 2x behind).
    std::vector<int32_t> gknn((size_t)n * K, -1);
    std::vector<uint8_t> gamb((size_t)n, 0);
    bool have_global = false;
    int64_t minx = 0, miny = 0, gw = 0, gh = 0;
    std::vector<int32_t> gcs, gpc, gorder;
    if (n > 0 && n < ((int64_t)1 << 30) && radius > 0) {
        int64_t maxx = pts[0], maxy = pts[1];
        minx = pts[0]; miny = pts[1];
        for (int64_t i = 1; i < n; ++i) {
            minx = std::min(minx, pts[2 * i]); maxx = std::max(maxx, pts[2 * i]);
            miny = std::min(miny, pts[2 * i + 1]); maxy = std::max(maxy, pts[2 * i + 1]);
        }
        gw = (maxx - minx) / radius + 1; gh = (maxy - miny) / radius + 1;
        if (gw * gh <= ((int64_t)1 << 24)) {
            gcs.assign((size_t)(gw * gh) + 1, 0); gpc.resize((size_t)n); gorder.resize((size_t)n);
            for (int64_t i = 0; i < n; ++i) {
                gpc[(size_t)i] = (int32_t)(((pts[2 * i + 1] - miny) / radius) * gw + (pts[2 * i] - minx) / radius);
                ++gcs[(size_t)gpc[(size_t)i] + 1];
            }
            for (size_t c = 0; c < (size_t)(gw * gh); ++c) gcs[c + 1] += gcs[c];
            std::vector<int32_t> fill(gcs.begin(), gcs.end() - 1);
            for (int64_t i = 0; i < n; ++i) gorder[(size_t)fill[(size_t)gpc[(size_t)i]]++] = (int32_t)i;
            have_global = true;
        }
    }
    auto global_knn = [&](int64_t i, std::vector<std::pair<int64_t, int32_t>>& cand) {
        cand.clear();
        const int64_t cx = gpc[(size_t)i] % gw, cy = gpc[(size_t)i] / gw;
        for (int64_t yy = std::max<int64_t>(cy - 1, 0); yy <= std::min<int64_t>(cy + 1, gh - 1); ++yy)
            for (int64_t xx = std::max<int64_t>(cx - 1, 0); xx <= std::min<int64_t>(cx + 1, gw - 1); ++xx)
                for (int32_t q = gcs[(size_t)(yy * gw + xx)]; q < gcs[(size_t)(yy * gw + xx) + 1]; ++q) {
                    const int32_t j = gorder[(size_t)q];
                    if (j == i) continue;
                    const int64_t dx = pts[2 * j] - pts[2 * i], dy = pts[2 * j + 1] - pts[2 * i + 1], d2 = dx * dx + dy * dy;
                    if (d2 < r2) cand.emplace_back(d2, j);
                }
        bool amb = false;
        const size_t keep = std::min<size_t>((size_t)K, cand.size());
        if (cand.size() > (size_t)K) {
            std::partial_sort(cand.begin(), cand.begin() + K + 1, cand.end());
            amb |= cand[(size_t)K].first == cand[(size_t)K - 1].first;
        } else {
            std::sort(cand.begin(), cand.end());
        }
        amb |= !cand.empty() && cand[0].first == 0;
        for (size_t q = 0; q < keep; ++q) gknn[(size_t)i * K + q] = cand[q].second;
        gamb[(size_t)i] = amb ? 1 : 0;
    };
    struct Scratch {
        std::vector<int32_t> gmap;                                      // global point index -> row in the current tile
        std::vector<int64_t> lx, ly;
        std::vector<std::pair<int64_t, int32_t>> cand;
        std::vector<int32_t> cstart, cfill, pcell, corder, res;
        std::vector<double> local;
        std::vector<srh_kd::detail::NodeInfo> pool;
        srh_kd::Tree tree;
    };
    auto do_tile = [&](int32_t t, Scratch& S) {
        auto& lx = S.lx; auto& ly = S.ly; auto& cand = S.cand; auto& cstart = S.cstart; auto& cfill = S.cfill; auto& pcell = S.pcell;
        auto& corder = S.corder; auto& gmap = S.gmap;
        bool any_amb = false;
        const int64_t x0 = boxes[4 * t], y0 = boxes[4 * t + 1], x1 = boxes[4 * t + 2], y1 = boxes[4 * t + 3];
        int64_t* tid = ids + offsets[t];
        int32_t* tk = knn + offsets[t] * K;
        lx.clear(); ly.clear();
        for (int64_t i = 0; i < n; ++i) {
            const int64_t x = pts[2 * i], y = pts[2 * i + 1];
            if (x >= x0 && x <= x1 && y >= y0 && y <= y1) { tid[lx.size()] = i; lx.push_back(x); ly.push_back(y); }
        }
        const int32_t m = (int32_t)lx.size();
        if (local) {
            int64_t* tl = local + offsets[t] * 2;
            for (int32_t i = 0; i < m; ++i) { tl[2 * i] = lx[i] - x0; tl[2 * i + 1] = ly[i] - y0; }
        }
        uint8_t* tamb = ambiguous + offsets[t];
        // uniform grid over the tile, cell side = radius: the neighbours of a point lie in its 3 x 3 cell block
        const int64_t cell = std::max<int64_t>(radius, 1);
        const int32_t tgw = (int32_t)((x1 - x0) / cell) + 1, tgh = (int32_t)((y1 - y0) / cell) + 1;
        cstart.assign((size_t)tgw * tgh + 1, 0);
        pcell.resize((size_t)m);
        for (int32_t i = 0; i < m; ++i) {
            pcell[(size_t)i] = (int32_t)((ly[i] - y0) / cell) * tgw + (int32_t)((lx[i] - x0) / cell);
            ++cstart[(size_t)pcell[(size_t)i] + 1];
        }
        for (size_t c = 0; c < (size_t)tgw * tgh; ++c) cstart[c + 1] += cstart[c];
        corder.resize((size_t)m);
        cfill.assign(cstart.begin(), cstart.end() - 1);
        for (int32_t i = 0; i < m; ++i) corder[(size_t)cfill[(size_t)pcell[(size_t)i]]++] = i;
        if (have_global) for (int32_t i = 0; i < m; ++i) gmap[(size_t)tid[i]] = i;
        for (int32_t i = 0; i < m; ++i) {
            bool amb = false;
            if (have_global && lx[i] - x0 >= radius && x1 - lx[i] >= radius && ly[i] - y0 >= radius && y1 - ly[i] >= radius) {
                // interior point: the global answer, re-indexed (every neighbour is inside the closed box)
                const int32_t* gk = gknn.data() + (size_t)tid[i] * K;
                for (int32_t q = 0; q < K; ++q) tk[(size_t)i * K + q] = gk[q] >= 0 ? gmap[(size_t)gk[q]] : -1;
                amb = gamb[(size_t)tid[i]] != 0;
                tamb[i] = amb ? 1 : 0;
                any_amb |= amb;
                continue;
            }
            cand.clear();
            const int32_t cx = pcell[(size_t)i] % tgw, cy = pcell[(size_t)i] / tgw;
            for (int32_t yy = std::max(cy - 1, 0); yy <= std::min(cy + 1, tgh - 1); ++yy)
                for (int32_t xx = std::max(cx - 1, 0); xx <= std::min(cx + 1, tgw - 1); ++xx)
                    for (int32_t q = cstart[(size_t)yy * tgw + xx]; q < cstart[(size_t)yy * tgw + xx + 1]; ++q) {
                        const int32_t j = corder[(size_t)q];
                        if (j == i) continue;
                        const int64_t dx = lx[j] - lx[i], dy = ly[j] - ly[i], d2 = dx * dx + dy * dy;
                        if (d2 < r2) cand.emplace_back(d2, j);
                    }
            const size_t keep = std::min<size_t>((size_t)K, cand.size());
            if (cand.size() > (size_t)K) {
                std::partial_sort(cand.begin(), cand.begin() + K + 1, cand.end());
                amb |= cand[K].first == cand[K - 1].first;          // cutoff falls inside a group of equidistant points
            } else {
                std::sort(cand.begin(), cand.end());
            }
            amb |= !cand.empty() && cand[0].first == 0;             // a point coinciding with the source
            for (size_t q = 0; q < (size_t)K; ++q) tk[(size_t)i * K + q] = q < keep ? cand[q].second : -1;
            tamb[i] = amb ? 1 : 0;
            any_amb |= amb;
        }
        if (any_amb) {
            // the reference's own query for those points: KDTree(tile-local points, ids ascending).query(p, k = K + 1,
            // distance_upper_bound = radius)[:, 1:] (reference inferencer.py:156-160)
            S.local.resize((size_t)m * 2);
            for (int32_t i = 0; i < m; ++i) { S.local[(size_t)i * 2] = (double)(lx[i] - x0); S.local[(size_t)i * 2 + 1] = (double)(ly[i] - y0); }
            srh_kd::build(S.tree, S.local.data(), m, 10);
            S.res.resize((size_t)K + 1);
            for (int32_t i = 0; i < m; ++i) {
                if (!tamb[i]) continue;
                srh_kd::query(S.tree, &S.local[(size_t)i * 2], K + 1, (double)radius, S.res.data(), nullptr, S.pool);
                for (int32_t q = 0; q < K; ++q) tk[(size_t)i * K + q] = S.res[(size_t)q + 1] < m ? S.res[(size_t)q + 1] : -1;
            }
        }
        if (have_global) for (int32_t i = 0; i < m; ++i) gmap[(size_t)tid[i]] = -1;
    };
    const int32_t nt = std::max<int32_t>(1, std::min<int32_t>(n_threads, std::max<int32_t>(n_tiles, 1)));
    std::atomic<int64_t> next_pt{0};
    std::atomic<int32_t> next_tile{0}, arrived{0};
    std::atomic<int> failed{0};
    auto worker = [&] {
        try {
            Scratch S;
            if (have_global) {
                for (;;) {
                    const int64_t i0 = next_pt.fetch_add(64);
                    if (i0 >= n) break;
                    for (int64_t i = i0; i < std::min(n, i0 + 64); ++i) global_knn(i, S.cand);
                }
                S.gmap.assign((size_t)n, -1);
            }
            arrived.fetch_add(1);
            while (arrived.load() < nt && !failed.load()) std::this_thread::yield();       // every point's global answer is complete
            for (;;) {
                const int32_t t = next_tile.fetch_add(1);
                if (t >= n_tiles || failed.load()) break;
                do_tile(t, S);
            }
        } catch (...) { failed.store(1); arrived.fetch_add(1); }
    };
    if (nt == 1) worker();
    else {
        JoiningThreads pool;
        try { for (int32_t w = 0; w < nt; ++w) pool.emplace_back(worker); }
        catch (...) { failed.store(1); throw; }      // a failed thread creation must not leave the started ones at the barrier
        for (auto& th : pool) th.join();
    }
    return failed.load() ? SRH_ERR_HIP : 0;
} catch (...) { return SRH_ERR_HIP; }      // std::bad_alloc / std::system_error (thread limit) must not cross the C ABI

// scipy.spatial.KDTree(points[n,2], leafsize).query(queries[nq,2], k, distance_upper_bound) restated (kdtree_emul.hpp): out_idx
// [nq,k] in scipy's output order (n = missing), tree_indices [n] (nullable) = scipy's tree.indices.  What srh_pass2_fill uses for
// tied cut-offs; exported so that the parity tests can pin tree structure and query results against scipy directly.
extern "C" int srh_kdtree_knn_host(const double* points, int64_t n, int32_t leafsize, const double* queries, int64_t nq, int32_t k,
                                   double distance_upper_bound, int32_t* out_idx, int32_t* tree_indices) try {
    if (n < 0 || nq < 0 || k <= 0 || leafsize <= 0 || n > 0x7fffffff || (n > 0 && !points) || (nq > 0 && (!queries || !out_idx)))
        return SRH_ERR_BAD_ARG;
    srh_kd::Tree tree;
    srh_kd::build(tree, points, (int32_t)n, leafsize);
    if (tree_indices) std::copy(tree.idx.begin(), tree.idx.end(), tree_indices);
    std::vector<srh_kd::detail::NodeInfo> pool;
    for (int64_t i = 0; i < nq; ++i) srh_kd::query(tree, queries + 2 * i, k, distance_upper_bound, out_idx + i * k, nullptr, pool);
    return 0;
} catch (...) { return SRH_ERR_HIP; }

// ---------------------------------------------------------------------------------------------------------------
// Directed edge votes (reference inferencer.py:209-221: a Python dict keyed by (src, tgt) accumulating score sums and
// counts over every valid pair of every tile, then mean > TOPO_THRESHOLD).  keys[i] = src * n_points + tgt and scores[i]
// arrive in the reference's visiting order (tile, point, neighbour slot).  A STABLE LSD radix sort by key keeps that
// order inside every key, so each sum is accumulated in float64 in exactly the reference's order: sums are bit-identical
// to the dict loop.  Outputs (capacity n): unique keys ascending, their sums and counts; *n_unique = how many.
// scratch: caller-provided, 2 * n int64 + n uint32 is NOT needed — the function allocates its own temporaries.
// ---------------------------------------------------------------------------------------------------------------
namespace {

// stable LSD radix sort of (key, index) pairs by key + the sequential float64 accumulation, on ONE contiguous range of votes.
// kcur / idx hold the range (n elements), ktmp / tmp are scratch of the same size; bits = key bits to sort on.  Writes the unique
// keys (ascending), sums, counts and first-vote indices to out_* and returns how many.
int64_t sort_accumulate_range(int64_t* kcur, uint32_t* idx, int64_t* ktmp, uint32_t* tmp, int64_t n, int64_t kmax, const double* scores,
                              int64_t* out_keys, double* out_sums, double* out_counts, int64_t* out_first) {
    // 11-bit digits: 3 passes cover 2^33 (n_points up to ~92k); more passes only if the keys need them.  (Two passes of 13-bit
    // digits were measured SLOWER, 12.2 vs 6.5 ms per CityScale scene: 8192 scatter streams defeat the write-combining.)
    const int BITS = 11, RAD = 1 << BITS;
    uint32_t hist[RAD];
    for (int shift = 0; shift < 63 && (kmax >> shift) != 0; shift += BITS) {
        std::fill(hist, hist + RAD, 0u);
        for (int64_t i = 0; i < n; ++i) ++hist[(size_t)((kcur[i] >> shift) & (RAD - 1))];
        bool single = false;                              // every key has the same digit (a thread's sub-range of the key space)
        for (int d = 0; d < RAD; ++d) if (hist[d] == (uint32_t)n) { single = true; break; }
        if (single) continue;
        uint32_t run = 0;
        for (int d = 0; d < RAD; ++d) { const uint32_t c = hist[d]; hist[d] = run; run += c; }
        for (int64_t i = 0; i < n; ++i) {
            const int64_t k = kcur[i];
            const uint32_t dst = hist[(size_t)((k >> shift) & (RAD - 1))]++;
            ktmp[dst] = k;
            tmp[dst] = idx[i];
        }
        std::swap(kcur, ktmp);
        std::swap(idx, tmp);
    }
    int64_t u = -1, prev = -1;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t j = idx[i];
        const int64_t k = kcur[i];
        // the sort is stable: the first vote of a run is the key's first visit = its insertion position in the reference's dict
        if (k != prev) { ++u; out_keys[u] = k; out_sums[u] = 0.0; out_counts[u] = 0.0; if (out_first) out_first[u] = j; prev = k; }
        out_sums[u] += scores[j];
        out_counts[u] += 1.0;
    }
    return u + 1;
}

}  // namespace

// n_threads > 1: the key space is cut into n_threads contiguous ranges of (nearly) equal vote counts by one stable partition
// pass on the keys' top 11 bits (per-thread histograms -> offsets -> scatter), then every thread sorts and accumulates its own
// range; the ranges' outputs are concatenated in key order.  Inside a key the votes keep their original order, so the sums are
// the same float64 sums, bit for bit, as with one thread.
extern "C" int srh_edge_vote_accumulate_mt(const int64_t* keys, const double* scores, int64_t n, int64_t* out_keys,
                                           double* out_sums, double* out_counts, int64_t* out_first, int64_t* n_unique,
                                           int32_t n_threads) try {
    if (n < 0 || !n_unique || (n > 0 && (!keys || !scores || !out_keys || !out_sums || !out_counts))) return SRH_ERR_BAD_ARG;
    *n_unique = 0;
    if (n == 0) return 0;
    if (n > 0x7fffffffLL) return SRH_ERR_UNSUPPORTED;
    int64_t kmax = 0;
    for (int64_t i = 0; i < n; ++i) { if (keys[i] < 0) return SRH_ERR_BAD_ARG; kmax = std::max(kmax, keys[i]); }
    // keys travel with their indices, so every pass reads and writes sequentially (an index-only sort gathers keys[idx[i]]
    // at random in each pass: 2x slower on ~500k votes)
    std::vector<uint32_t> idx((size_t)n), tmp((size_t)n);
    std::vector<int64_t> kcur((size_t)n), ktmp((size_t)n);
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n / 65536));     // below ~64k votes per thread it does not pay
    if (T == 1) {
        std::copy(keys, keys + n, kcur.begin());
        for (int64_t i = 0; i < n; ++i) idx[(size_t)i] = (uint32_t)i;
        *n_unique = sort_accumulate_range(kcur.data(), idx.data(), ktmp.data(), tmp.data(), n, kmax, scores, out_keys, out_sums,
                                          out_counts, out_first);
        return 0;
    }
    // ---- partition by the top bits (NB <= 2048 buckets), stable -----------------------------------------------------
    int kbits = 0;
    while ((kmax >> kbits) != 0) ++kbits;
    const int sh = std::max(0, kbits - 11);
    const int NB = (int)(kmax >> sh) + 1;
    std::vector<uint32_t> hist((size_t)T * NB, 0u);
    auto chunk = [&](int t) { return std::pair<int64_t, int64_t>(n * t / T, n * (t + 1) / T); };
    {
        JoiningThreads pool;
        for (int t = 0; t < T; ++t) pool.emplace_back([&, t] {
            uint32_t* h = hist.data() + (size_t)t * NB;
            const auto c = chunk(t);
            for (int64_t i = c.first; i < c.second; ++i) ++h[(size_t)(keys[i] >> sh)];
        });
        for (auto& th : pool) th.join();
    }
    std::vector<int64_t> bstart((size_t)NB + 1, 0);
    {
        int64_t run = 0;
        for (int b = 0; b < NB; ++b) {
            bstart[(size_t)b] = run;
            for (int t = 0; t < T; ++t) { const uint32_t c = hist[(size_t)t * NB + b]; hist[(size_t)t * NB + b] = (uint32_t)run; run += c; }
        }
        bstart[(size_t)NB] = run;
    }
    // thread r owns buckets [cut[r], cut[r+1]): contiguous key ranges with ~n / T votes each
    std::vector<int> cut((size_t)T + 1, NB);
    cut[0] = 0;
    for (int r = 1, b = 0; r < T; ++r) {
        while (b < NB && bstart[(size_t)b] < n * r / T) ++b;
        cut[(size_t)r] = b;
    }
    std::vector<int64_t> nu((size_t)T, 0);
    std::vector<std::vector<int64_t>> ok((size_t)T), of((size_t)T);
    std::vector<std::vector<double>> os((size_t)T), oc((size_t)T);
    {
        JoiningThreads pool;
        for (int t = 0; t < T; ++t) pool.emplace_back([&, t] {
            uint32_t* h = hist.data() + (size_t)t * NB;
            const auto c = chunk(t);
            for (int64_t i = c.first; i < c.second; ++i) {
                const int64_t k = keys[i];
                const uint32_t dst = h[(size_t)(k >> sh)]++;
                kcur[dst] = k;
                idx[dst] = (uint32_t)i;
            }
        });
        for (auto& th : pool) th.join();
    }
    {
        JoiningThreads pool;
        for (int r = 0; r < T; ++r) pool.emplace_back([&, r] {
            const int64_t lo = bstart[(size_t)cut[(size_t)r]], hi = bstart[(size_t)cut[(size_t)r + 1]], m = hi - lo;
            if (m <= 0) return;
            ok[(size_t)r].resize((size_t)m); os[(size_t)r].resize((size_t)m); oc[(size_t)r].resize((size_t)m); of[(size_t)r].resize((size_t)m);
            nu[(size_t)r] = sort_accumulate_range(kcur.data() + lo, idx.data() + lo, ktmp.data() + lo, tmp.data() + lo, m, kmax, scores,
                                                  ok[(size_t)r].data(), os[(size_t)r].data(), oc[(size_t)r].data(), of[(size_t)r].data());
        });
        for (auto& th : pool) th.join();
    }
    int64_t u = 0;
    for (int r = 0; r < T; ++r) {
        const size_t m = (size_t)nu[(size_t)r];
        std::copy(ok[(size_t)r].begin(), ok[(size_t)r].begin() + m, out_keys + u);
        std::copy(os[(size_t)r].begin(), os[(size_t)r].begin() + m, out_sums + u);
        std::copy(oc[(size_t)r].begin(), oc[(size_t)r].begin() + m, out_counts + u);
        if (out_first) std::copy(of[(size_t)r].begin(), of[(size_t)r].begin() + m, out_first + u);
        u += (int64_t)m;
    }
    *n_unique = u;
    return 0;
} catch (...) { return SRH_ERR_HIP; }      // std::bad_alloc / std::system_error (thread limit) must not cross the C ABI

extern "C" int srh_edge_vote_accumulate(const int64_t* keys, const double* scores, int64_t n, int64_t* out_keys,
                                        double* out_sums, double* out_counts, int64_t* out_first, int64_t* n_unique) try {
    return srh_edge_vote_accumulate_mt(keys, scores, n, out_keys, out_sums, out_counts, out_first, n_unique, 1);
} catch (...) { return SRH_ERR_HIP; }      // std::bad_alloc / std::system_error (thread limit) must not cross the C ABI

// ---------------------------------------------------------------------------------------------------------------
// Directed edge votes of ONE TopoNet batch (reference inferencer.py:206-221: the triple Python loop over tiles, source
// points and neighbour slots that fills the (src, tgt)-keyed dicts).  scores [nb, n_max, K] f32 = infer_toponet output
// (NaN already mapped to -100 as inferencer.py:206 does) of nb consecutive tiles whose query rows are offsets[0..nb] into
// ids / knn (srh_pass2_fill layout, knn = tile-local target or -1).  Appends, in the reference's visiting order,
// keys[*count] = ids[src] * n_points + ids[tgt] and the score as float64; *count is advanced.  Returns SRH_ERR_BAD_ARG if a
// valid pair's score is outside [0, 1] (the reference asserts that, inferencer.py:219).
// ---------------------------------------------------------------------------------------------------------------
extern "C" int srh_pass2_votes(const float* scores, int32_t nb, int64_t n_max, int32_t K, const int64_t* offsets,
                               const int64_t* ids, const int32_t* knn, int64_t n_points, int64_t* keys, double* votes,
                               int64_t capacity, int64_t* count) try {
    if (!scores || !offsets || !ids || !knn || !keys || !votes || !count || nb < 0 || K <= 0) return SRH_ERR_BAD_ARG;
    int64_t c = *count;
    for (int32_t b = 0; b < nb; ++b) {
        const int64_t a = offsets[b], n = offsets[b + 1] - a;
        if (n > n_max) return SRH_ERR_BAD_ARG;
        for (int64_t si = 0; si < n; ++si) {
            const int32_t* row = knn + (a + si) * K;
            const float* sc = scores + ((int64_t)b * n_max + si) * K;
            const int64_t src = ids[a + si] * n_points;
            for (int32_t pi = 0; pi < K; ++pi) {
                if (row[pi] < 0) continue;
                const float v = sc[pi];
                if (!(v >= 0.0f && v <= 1.0f)) return SRH_ERR_BAD_ARG;
                if (c >= capacity) return SRH_ERR_BAD_ARG;
                keys[c] = src + ids[a + row[pi]];
                votes[c] = (double)v;
                ++c;
            }
        }
    }
    *count = c;
    return 0;
} catch (...) { return SRH_ERR_HIP; }      // std::bad_alloc / std::system_error (thread limit) must not cross the C ABI

// ---------------------------------------------------------------------------------------------------------------
// Candidate pixels of a fused u8 mask (reference graph_extraction.py:24-28: `np.where(mask > threshold)` + the scores at
// those pixels), in np.where's row-major order.  Two-step protocol: call with xy = scores = NULL to get *n, allocate, call
// again.  xy int64 [n, 2] as (x, y) = (column, row), scores u8 [n].  The threshold is the reference's float
// (THRESHOLD * 255): a u8 value v is a candidate iff (float)v > threshold.
// ---------------------------------------------------------------------------------------------------------------
extern "C" int srh_mask_candidates(const uint8_t* mask, int32_t H, int32_t W, float threshold, int64_t* xy, uint8_t* scores,
                                   int64_t capacity, int64_t* n, int32_t n_threads) try {
    if (!mask || !n || H < 0 || W < 0 || ((xy == nullptr) != (scores == nullptr))) return SRH_ERR_BAD_ARG;
    int first = 256;                      // smallest u8 value that passes
    for (int v = 255; v >= 0; --v) { if ((float)v > threshold) first = v; else break; }
    *n = 0;
    if (first > 255 || H == 0 || W == 0) return 0;
    const uint8_t f8 = (uint8_t)first;
    // masks are sparse (a few % of the pixels pass): test 64-byte chunks with a byte-max reduction (vectorises to pmaxub)
    // and look at single pixels only inside chunks that contain a candidate
    auto chunk_has = [f8](const uint8_t* q, int len) {
        uint8_t m = 0;
        for (int i = 0; i < len; ++i) m = q[i] > m ? q[i] : m;
        return m >= f8;
    };
    // rows [y0, y1): count, or write from position `at` on (np.where order = row-major, so a band's candidates are contiguous)
    auto band = [&](int32_t y0, int32_t y1, bool write, int64_t at) -> int64_t {
        int64_t c = 0;
        for (int32_t y = y0; y < y1; ++y) {
            const uint8_t* row = mask + (size_t)y * W;
            for (int32_t x0 = 0; x0 < W; x0 += 64) {
                const int len = std::min(64, W - x0);
                if (!chunk_has(row + x0, len)) continue;
                for (int32_t x = x0; x < x0 + len; ++x) {
                    if (row[x] >= f8) {
                        if (write) { xy[2 * (at + c)] = x; xy[2 * (at + c) + 1] = y; scores[at + c] = row[x]; }
                        ++c;
                    }
                }
            }
        }
        return c;
    };
    // bands of rows on worker threads: count, prefix, then every thread writes its band (its 1/T of the mask is still in its L2)
    const int32_t T = std::max<int32_t>(1, std::min<int32_t>(std::min<int32_t>(n_threads, 64), H / 64));
    std::vector<int64_t> cnt((size_t)T, 0), at((size_t)T + 1, 0);
    auto rows = [&](int32_t t) { return std::pair<int32_t, int32_t>((int32_t)((int64_t)H * t / T), (int32_t)((int64_t)H * (t + 1) / T)); };
    std::atomic<int32_t> arrived{0};
    std::atomic<int> go{0};                  // 0: wait, 1: write, 2: stop (count only / capacity too small)
    auto worker = [&](int32_t t) {
        const auto r = rows(t);
        cnt[(size_t)t] = band(r.first, r.second, false, 0);
        arrived.fetch_add(1);
        if (t == 0) {
            while (arrived.load() < T) std::this_thread::yield();
            for (int32_t u = 0; u < T; ++u) at[(size_t)u + 1] = at[(size_t)u] + cnt[(size_t)u];
            go.store(xy && at[(size_t)T] <= capacity ? 1 : 2);
        } else {
            while (go.load() == 0) std::this_thread::yield();
        }
        if (go.load() == 1) band(r.first, r.second, true, at[(size_t)t]);
    };
    if (T == 1) worker(0);
    else {
        JoiningThreads pool;
        try { for (int32_t t = 1; t < T; ++t) pool.emplace_back(worker, t); }
        catch (...) { go.store(2); throw; }          // a failed thread creation must not leave the started ones spinning
        worker(0);
        for (auto& th : pool) th.join();
    }
    *n = at[(size_t)T];
    return (xy && at[(size_t)T] > capacity) ? SRH_ERR_BAD_ARG : 0;
} catch (...) { return SRH_ERR_HIP; }      // std::bad_alloc / std::system_error (thread limit) must not cross the C ABI

// ---------------------------------------------------------------------------------------------------------------
// The LAST of extract_graph_points' three nms_points calls (reference graph_extraction.py:136-139 -> graph_utils.py:572-591) with
// the two gathers in front of it: candidates = [xy_a[ord_a]; xy_b[ord_b]] (the keypoint and the road candidates, each already in
// ITS np.argsort(scores)[::-1] order — with u8 scores above 1.0 the first two nms_points calls keep everything and only reorder),
// visited in `order` (= np.argsort(priorities)[::-1] of [1]*na + [0]*nb, computed by the caller so that numpy's tie order is the
// reference's), none of them forced (priorities <= 1.0).  Writes the kept points (x, y) in visiting order.  The three numpy
// fancy-index gathers of ~140k rows and the boolean compaction cost more than the suppression itself.
// ---------------------------------------------------------------------------------------------------------------
extern "C" int srh_nms_merge_points(const int64_t* xy_a, const int64_t* ord_a, int64_t na, const int64_t* xy_b, const int64_t* ord_b,
                                    int64_t nb, const int64_t* order, int32_t radius, int64_t* out_xy, int64_t* n_out) try {
    const int64_t n = na + nb;
    if (!n_out || na < 0 || nb < 0 || radius < 0 || (na > 0 && (!xy_a || !ord_a)) || (nb > 0 && (!xy_b || !ord_b)) || (n > 0 && (!order || !out_xy)))
        return SRH_ERR_BAD_ARG;
    *n_out = 0;
    if (n == 0) return 0;
    std::vector<int32_t> xy((size_t)n * 2);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t j = order[i];
        if (j < 0 || j >= n) return SRH_ERR_BAD_ARG;
        const int64_t* src;
        if (j < na) { const int64_t q = ord_a[j]; if (q < 0 || q >= na) return SRH_ERR_BAD_ARG; src = xy_a + 2 * q; }
        else { const int64_t q = ord_b[j - na]; if (q < 0 || q >= nb) return SRH_ERR_BAD_ARG; src = xy_b + 2 * q; }
        if (src[0] < INT32_MIN || src[0] > INT32_MAX || src[1] < INT32_MIN || src[1] > INT32_MAX) return SRH_ERR_BAD_ARG;
        xy[(size_t)i * 2] = (int32_t)src[0]; xy[(size_t)i * 2 + 1] = (int32_t)src[1];
    }
    std::vector<uint8_t> force((size_t)n, 0), kept((size_t)n);
    const int rc = srh_nms_points_host(xy.data(), force.data(), n, radius, kept.data());
    if (rc != 0) return rc;
    int64_t c = 0;
    for (int64_t i = 0; i < n; ++i)
        if (kept[(size_t)i]) { out_xy[2 * c] = xy[(size_t)i * 2]; out_xy[2 * c + 1] = xy[(size_t)i * 2 + 1]; ++c; }
    *n_out = c;
    return 0;
} catch (...) { return SRH_ERR_HIP; }      // std::bad_alloc / std::system_error (thread limit) must not cross the C ABI

// ---------------------------------------------------------------------------------------------------------------
// Padded collate of one TopoNet batch (reference inferencer.py:179-185: graph_collate_fn-style zero padding of the per-tile
// point / pair / valid arrays to the longest tile) straight from the flat query arrays of srh_pass2_fill: tile b of the batch
// owns rows offsets[b] .. offsets[b+1] of local [*,2] (tile-local integer x, y) and knn [*,K] (tile-local target or -1).
// Writes points f32 [nb, n_max, 2], pairs i32 [nb, n_max, K, 2] = (row, target or row itself when invalid) and valid u8
// [nb, n_max, K]; rows beyond a tile's count are zero.
// ---------------------------------------------------------------------------------------------------------------
extern "C" int srh_pass2_pack(const int64_t* offsets, const int64_t* local, const int32_t* knn, int32_t nb, int64_t n_max, int32_t K,
                              float* points, int32_t* pairs, uint8_t* valid) try {
    if (!offsets || !local || !knn || !points || !pairs || !valid || nb < 0 || n_max < 0 || K <= 0) return SRH_ERR_BAD_ARG;
    for (int32_t b = 0; b < nb; ++b) {
        const int64_t a = offsets[b], n = offsets[b + 1] - a;
        if (n < 0 || n > n_max) return SRH_ERR_BAD_ARG;
        float* pt = points + (int64_t)b * n_max * 2;
        int32_t* pr = pairs + (int64_t)b * n_max * K * 2;
        uint8_t* vl = valid + (int64_t)b * n_max * K;
        for (int64_t r = 0; r < n; ++r) {
            pt[2 * r] = (float)local[2 * (a + r)];
            pt[2 * r + 1] = (float)local[2 * (a + r) + 1];
            const int32_t* row = knn + (a + r) * K;
            for (int32_t j = 0; j < K; ++j) {
                const bool ok = row[j] >= 0;
                vl[r * K + j] = ok ? 1 : 0;
                pr[(r * K + j) * 2] = (int32_t)r;
                pr[(r * K + j) * 2 + 1] = ok ? row[j] : (int32_t)r;
            }
        }
        std::memset(pt + 2 * n, 0, (size_t)(n_max - n) * 2 * sizeof(float));
        std::memset(pr + n * K * 2, 0, (size_t)(n_max - n) * K * 2 * sizeof(int32_t));
        std::memset(vl + n * K, 0, (size_t)(n_max - n) * K);
    }
    return 0;
} catch (...) { return SRH_ERR_HIP; }      // std::bad_alloc / std::system_error (thread limit) must not cross the C ABI

// The same collate WITHOUT padding (srh_toponet_ragged): the flat query arrays are already the concatenation of the tiles' point lists,
// so a row keeps its position; pairs become indices into the flat list (offsets[t] + tile-local index) and every row names its tile.
//   points f32 [R, 2], pairs i32 [R, K, 2], valid u8 [R, K], point_tile i32 [R]  with R = offsets[n_tiles] - offsets[0]; rows are
//   numbered from offsets[0] (the first tile of the range); tile numbers in point_tile count from 0 at that tile.
extern "C" int srh_pass2_pack_ragged(const int64_t* offsets, const int64_t* local, const int32_t* knn, int32_t n_tiles, int32_t K,
                                     float* points, int32_t* pairs, uint8_t* valid, int32_t* point_tile) try {
    if (!offsets || !local || !knn || !points || !pairs || !valid || !point_tile || n_tiles < 0 || K <= 0) return SRH_ERR_BAD_ARG;
    const int64_t base = offsets[0];
    for (int32_t t = 0; t < n_tiles; ++t) {
        const int64_t a = offsets[t], n = offsets[t + 1] - a;
        if (n < 0) return SRH_ERR_BAD_ARG;
        for (int64_t r = 0; r < n; ++r) {
            const int64_t g = a + r - base;                       // row in the outputs
            points[2 * g] = (float)local[2 * (a + r)];
            points[2 * g + 1] = (float)local[2 * (a + r) + 1];
            point_tile[g] = t;
            const int32_t* row = knn + (a + r) * K;
            for (int32_t j = 0; j < K; ++j) {
                const bool ok = row[j] >= 0;
                if (ok && row[j] >= n) return SRH_ERR_BAD_ARG;
                valid[g * K + j] = ok ? 1 : 0;
                pairs[(g * K + j) * 2] = (int32_t)g;
                pairs[(g * K + j) * 2 + 1] = ok ? (int32_t)(a - base + row[j]) : (int32_t)g;
            }
        }
    }
    return 0;
} catch (...) { return SRH_ERR_HIP; }

// ---------------------------------------------------------------------------------------------------------------
// srh_pass2_votes + srh_edge_vote_accumulate in ONE pass, without materialising the votes (reference inferencer.py:206-228: the
// (src, tgt)-keyed dicts).  The ~700k votes of a CityScale scene belong to only ~80k distinct edges, and every source point meets
// the same few dozen targets again and again (its neighbours, once per tile that contains it): instead of writing a key per vote
// and sorting them, the query rows are grouped by source point (a counting sort of ~50k rows) and each source point's votes are
// added into a small table of its targets, in ascending row order = the reference's visiting order for every key (a key occurs at
// most once per tile), so the float64 sums are the same sums, bit for bit.  Source points are dealt to worker threads in contiguous
// ranges; the outputs are concatenated in key order.
//   scores[b]      f32 [batch_nb[b], batch_n_max[b], K]: infer_toponet output of the tiles batch_tile0[b] .. + batch_nb[b] (NaN -> -100 done)
//   offsets        [n_tiles + 1] rows per tile into ids / knn (srh_pass2_fill layout); tiles outside every batch must be empty
//   out_*          unique keys src * n_points + tgt ascending, float64 sums, counts (as doubles), first-vote positions (the index
//                  the vote would have had in srh_pass2_votes' output); capacity >= number of valid slots is always enough
// Returns SRH_ERR_BAD_ARG if a valid pair's score is outside [0, 1] (inferencer.py:219 asserts that) or the arrays are inconsistent.
// ---------------------------------------------------------------------------------------------------------------
extern "C" int srh_pass2_vote_sums(const float* const* scores, const int32_t* batch_tile0, const int32_t* batch_nb,
                                   const int64_t* batch_n_max, int32_t n_batches, int32_t K, const int64_t* offsets, int32_t n_tiles,
                                   const int64_t* ids, const int32_t* knn, int64_t n_points, int64_t* out_keys, double* out_sums,
                                   double* out_counts, int64_t* out_first, int64_t capacity, int64_t* n_unique, int32_t n_threads) try {
    if (!n_unique || n_batches < 0 || n_tiles < 0 || K <= 0 || n_points < 0 || !offsets) return SRH_ERR_BAD_ARG;
    *n_unique = 0;
    const int64_t R = offsets[n_tiles] - offsets[0];
    if (R < 0) return SRH_ERR_BAD_ARG;
    if (R == 0) return 0;
    if (!scores || !batch_tile0 || !batch_nb || !batch_n_max || !ids || !knn || !out_keys || !out_sums || !out_counts) return SRH_ERR_BAD_ARG;
    const int64_t r0 = offsets[0];
    // what the grouped pass needs of a query row: its K scores, its K tile-local targets, its tile's first row (tile-local target ->
    // row -> global id), the tile's row count, and the position of its first vote in the visiting order
    struct Row { const float* sc; const int32_t* nb; int64_t tile_a; int64_t vbase; int32_t tile_n; };
    std::vector<Row> rows((size_t)R, Row{nullptr, nullptr, 0, 0, 0});
    for (int32_t b = 0; b < n_batches; ++b) {
        const int32_t t0 = batch_tile0[b], nb = batch_nb[b];
        if (t0 < 0 || nb < 0 || t0 + nb > n_tiles || !scores[b]) return SRH_ERR_BAD_ARG;
        for (int32_t t = t0; t < t0 + nb; ++t) {
            const int64_t a = offsets[t], n = offsets[t + 1] - a;
            if (n < 0 || n > batch_n_max[b] || n > 0x7fffffffLL) return SRH_ERR_BAD_ARG;
            for (int64_t si = 0; si < n; ++si) {
                Row& w = rows[(size_t)(a - r0 + si)];
                w.sc = scores[b] + ((int64_t)(t - t0) * batch_n_max[b] + si) * K;
                w.nb = knn + (a + si) * K;
                w.tile_a = a;
                w.tile_n = (int32_t)n;
            }
        }
    }
    std::vector<int64_t> start((size_t)n_points + 1, 0);
    int64_t votes = 0;
    for (int64_t r = 0; r < R; ++r) {
        Row& w = rows[(size_t)r];
        if (!w.sc) return SRH_ERR_BAD_ARG;                          // a non-empty tile that no batch covers
        const int64_t id = ids[r0 + r];
        if (id < 0 || id >= n_points) return SRH_ERR_BAD_ARG;
        ++start[(size_t)id + 1];
        w.vbase = votes;
        for (int32_t j = 0; j < K; ++j) votes += w.nb[j] >= 0;
    }
    for (int64_t i = 0; i < n_points; ++i) start[(size_t)i + 1] += start[(size_t)i];
    // rows in (source point, row) order — a stable counting sort; the grouped pass then walks `grp` front to back
    std::vector<Row> grp((size_t)R);
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t r = 0; r < R; ++r) grp[(size_t)fill[(size_t)ids[r0 + r]]++] = rows[(size_t)r];
    }
    // a worker thread pays for itself only with >= ~1 ms of work: waking an idle core and warming its caches costs more than the
    // ~0.3 ms per thread that a CityScale scene (48k rows, 700k votes, 2.6 ms on one core) would give eight of them — measured
    // SLOWER with every thread added (profiles/r03_host_stages.txt)
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, R / 40000));
    std::vector<int64_t> cut((size_t)T + 1, n_points);               // thread t owns source points [cut[t], cut[t+1]): ~R / T rows each
    cut[0] = 0;
    for (int t = 1, p = 0; t < T; ++t) {
        while (p < n_points && start[(size_t)p] < R * t / T) ++p;
        cut[(size_t)t] = p;
    }
    struct Entry { int64_t tgt; double sum, cnt; int64_t first; };
    std::vector<std::vector<int64_t>> ok((size_t)T), of((size_t)T);
    std::vector<std::vector<double>> os((size_t)T), oc((size_t)T);
    std::vector<int> bad((size_t)T, 0);
    auto work = [&](int t) {
        std::vector<Entry> tab;
        tab.reserve(64);
        std::vector<int32_t> slot((size_t)n_points, -1);            // target -> its entry in tab (reset after every source point)
        auto& k_ = ok[(size_t)t]; auto& s_ = os[(size_t)t]; auto& c_ = oc[(size_t)t]; auto& f_ = of[(size_t)t];
        const size_t guess = (size_t)((votes / std::max<int64_t>(1, T)) / 4 + 64);
        k_.reserve(guess); s_.reserve(guess); c_.reserve(guess); f_.reserve(guess);
        for (int64_t src = cut[(size_t)t]; src < cut[(size_t)t + 1]; ++src) {
            tab.clear();
            for (int64_t q = start[(size_t)src]; q < start[(size_t)src + 1]; ++q) {
                if (q + 6 < R) {               // a source point's rows lie in different tiles: one knn line and one score line each, far apart
                    __builtin_prefetch(grp[(size_t)q + 6].nb);
                    __builtin_prefetch(grp[(size_t)q + 6].sc);
                }
                const Row& w = grp[(size_t)q];
                int64_t pos = w.vbase;
                for (int32_t j = 0; j < K; ++j) {
                    const int32_t l = w.nb[j];
                    if (l < 0) continue;
                    const float v = w.sc[j];
                    if (!(v >= 0.0f && v <= 1.0f) || l >= w.tile_n) { bad[(size_t)t] = 1; return; }
                    const int64_t tgt = ids[w.tile_a + l];
                    if (tgt < 0 || tgt >= n_points) { bad[(size_t)t] = 1; return; }
                    int32_t si = slot[(size_t)tgt];
                    if (si < 0) { si = slot[(size_t)tgt] = (int32_t)tab.size(); tab.push_back(Entry{tgt, 0.0, 0.0, pos}); }
                    Entry& e = tab[(size_t)si];
                    e.sum += (double)v;
                    e.cnt += 1.0;
                    ++pos;
                }
            }
            for (const Entry& e : tab) slot[(size_t)e.tgt] = -1;
            std::sort(tab.begin(), tab.end(), [](const Entry& x, const Entry& y) { return x.tgt < y.tgt; });
            for (const Entry& e : tab) {
                k_.push_back(src * n_points + e.tgt); s_.push_back(e.sum); c_.push_back(e.cnt); f_.push_back(e.first);
            }
        }
    };
    if (T == 1) work(0);
    else {
        JoiningThreads pool;
        for (int t = 0; t < T; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    int64_t u = 0;
    for (int t = 0; t < T; ++t) { if (bad[(size_t)t]) return SRH_ERR_BAD_ARG; u += (int64_t)ok[(size_t)t].size(); }
    if (u > capacity) return SRH_ERR_BAD_ARG;
    u = 0;
    for (int t = 0; t < T; ++t) {
        const size_t m = ok[(size_t)t].size();
        std::copy(ok[(size_t)t].begin(), ok[(size_t)t].end(), out_keys + u);
        std::copy(os[(size_t)t].begin(), os[(size_t)t].end(), out_sums + u);
        std::copy(oc[(size_t)t].begin(), oc[(size_t)t].end(), out_counts + u);
        if (out_first) std::copy(of[(size_t)t].begin(), of[(size_t)t].end(), out_first + u);
        u += (int64_t)m;
    }
    *n_unique = u;
    return 0;
} catch (...) { return SRH_ERR_HIP; }      // std::bad_alloc / std::system_error (thread limit) must not cross the C ABI

// ---------------------------------------------------------------------------------------------------------------
// Edge list from the vote sums (reference inferencer.py:224-228): the directed edges whose MEAN score exceeds the threshold, in the
// insertion order of the reference's dict = ascending first-vote position.  keys = src * n_points + tgt (unique), sums / counts
// float64, first = position of each key's first vote (distinct, < first_bound).  The kept edges are dropped into a table indexed
// by first-vote position and read back in order (no comparison sort).  out_edges int64 [n, 2] (src, tgt).
// ---------------------------------------------------------------------------------------------------------------
extern "C" int srh_votes_to_edges(const int64_t* keys, const double* sums, const double* counts, const int64_t* first, int64_t n,
                                  int64_t n_points, double threshold, int64_t* out_edges, int64_t* n_edges) try {
    if (!n_edges || n < 0 || n_points <= 0 || (n > 0 && (!keys || !sums || !counts || !first || !out_edges))) return SRH_ERR_BAD_ARG;
    *n_edges = 0;
    if (n == 0) return 0;
    if (n > 0x7fffffffLL) return SRH_ERR_UNSUPPORTED;
    auto passes = [&](int64_t i) { return sums[i] / (counts[i] > 1.0 ? counts[i] : 1.0) > threshold; };      // np.maximum(cnts, 1.0)
    auto emit = [&](int64_t e, int64_t i) { out_edges[2 * e] = keys[i] / n_points; out_edges[2 * e + 1] = keys[i] % n_points; };
    int64_t bound = 0;
    bool dense = true;
    for (int64_t i = 0; i < n; ++i) { if (first[i] < 0) dense = false; bound = std::max(bound, first[i] + 1); }
    dense = dense && bound <= 64 * n + ((int64_t)1 << 20);
    // dense positions (one process: positions < number of votes): drop the kept edges into a table indexed by position, read it back
    if (dense) {
        std::vector<int32_t> at((size_t)bound, -1);
        for (int64_t i = 0; i < n && dense; ++i)
            if (passes(i)) {
                if (at[(size_t)first[i]] >= 0) dense = false;            // equal positions: keep numpy's stable order instead
                at[(size_t)first[i]] = (int32_t)i;
            }
        if (dense) {
            int64_t e = 0;
            for (int64_t p = 0; p < bound; ++p) if (at[(size_t)p] >= 0) emit(e++, at[(size_t)p]);
            *n_edges = e;
            return 0;
        }
    }
    // sparse or repeated positions (a multi-rank merge offsets every rank's positions by rank * 2^40, distributed.gather_edge_votes):
    // a stable sort by position, exactly np.argsort(first[keep], kind="stable")
    std::vector<std::pair<int64_t, int32_t>> kept;
    for (int64_t i = 0; i < n; ++i) if (passes(i)) kept.emplace_back(first[i], (int32_t)i);
    std::stable_sort(kept.begin(), kept.end(), [](const std::pair<int64_t, int32_t>& x, const std::pair<int64_t, int32_t>& y) { return x.first < y.first; });
    int64_t e = 0;
    for (const auto& kv : kept) emit(e++, kv.second);
    *n_edges = e;
    return 0;
} catch (...) { return SRH_ERR_HIP; }      // std::bad_alloc / std::system_error (thread limit) must not cross the C ABI
