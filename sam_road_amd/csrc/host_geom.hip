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
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/samroad_hip.h"

extern "C" int srh_nms_points_host(const int32_t* xy, const uint8_t* force, int64_t n, int32_t radius, uint8_t* kept) {
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
}
