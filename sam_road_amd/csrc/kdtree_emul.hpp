// k-nearest-neighbour queries with the TIE BEHAVIOUR of scipy.spatial.KDTree — host code, no device work.
//
// Why this exists.  The reference builds, per tile, `scipy.spatial.KDTree(patch_points)` and asks it for the 17 nearest
// points of every point within NEIGHBOR_RADIUS (reference inferencer.py:156-160).  Graph points are integer pixels, so the
// 16-th and 17-th neighbour of a point are often EQUIDISTANT; which one the reference keeps is not a property of the
// geometry but of scipy's kd-tree: the shape of the tree (which points share a leaf, in which order), the order in which
// the query visits leaves, and the `d < upper_bound` test that lets the first-visited of two equidistant candidates win.
// Round 2 answered those cases by calling scipy itself on every affected tile (~12 ms of interpreter time per CityScale
// scene).  This header restates the two scipy routines that decide the outcome, for m = 2 dimensions, p = 2, eps = 0 and no
// periodic box — scipy 1.15 `scipy/spatial/ckdtree/src/build.cxx` (build, partition_node_indices) and `query.cxx`
// (query_single_point, struct heap) — so that the library answers them itself.  tests/test_host_logic.py pins the
// restatement against scipy on lattices where nearly every cut-off is tied (tree structure, leaf order and query results).
//
// Faithfulness notes (each one is what makes a tie come out the same way):
//   * the tree is balanced with std::nth_element over point indices (comparator: the split coordinate only) followed by the
//     Hoare-style partition around the median value, and bounds are recomputed per node ("compact"), split dimension = the
//     FIRST dimension of largest extent; leafsize 10 (scipy.spatial.KDTree's default);
//   * std::nth_element leaves an implementation-defined order inside both halves; scipy's wheels and this library are both
//     built against libstdc++'s introselect, whose algorithm has not changed since GCC 4 — the test compares the resulting
//     index permutation with scipy's `tree.indices`;
//   * a leaf is scanned in index-array order with `if (d < bound)`: candidates at exactly the current k-th distance lose;
//   * both priority queues are scipy's own binary heap (push sifts up on `<`, remove sifts down preferring the left child
//     on ties), NOT std::priority_queue;
//   * the far child inherits the parent's side distances with the split dimension's entry replaced; the near / far pair is
//     swapped if that made the "far" one closer.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace srh_kd {

struct Node {
    int32_t split_dim;     // -1: leaf
    double split;
    int32_t start, end;    // range of `idx`
    int32_t less, greater; // children (node indices), -1 for a leaf
};

struct Tree {
    std::vector<Node> nodes;
    std::vector<int32_t> idx;      // scipy's tree.indices
    std::vector<double> data;      // [n][2]
    double mins[2], maxes[2];      // bounds of the whole point set (scipy's tree.mins / tree.maxes)
    int32_t n = 0;
};

namespace detail {

inline int32_t build(Tree& t, int32_t start, int32_t end, double* maxes, double* mins, int32_t leafsize) {
    const double* data = t.data.data();
    int32_t* indices = t.idx.data();
    const int32_t node_index = (int32_t)t.nodes.size();
    t.nodes.push_back(Node{-1, 0.0, start, end, -1, -1});
    if (end - start <= leafsize) return node_index;
    // recompute the node's bounds (compact_nodes=True)
    for (int i = 0; i < 2; ++i) maxes[i] = mins[i] = data[(size_t)indices[start] * 2 + i];
    for (int32_t j = start + 1; j < end; ++j)
        for (int i = 0; i < 2; ++i) {
            const double v = data[(size_t)indices[j] * 2 + i];
            maxes[i] = maxes[i] > v ? maxes[i] : v;
            mins[i] = mins[i] < v ? mins[i] : v;
        }
    int d = 0;
    double size = 0;
    for (int i = 0; i < 2; ++i)
        if (maxes[i] - mins[i] > size) { d = i; size = maxes[i] - mins[i]; }
    if (maxes[d] == mins[d]) return node_index;      // all points identical: leaf
    // balanced_tree=True: median by std::nth_element over indices, comparing the coordinate ONLY (no index tie-break: scipy
    // 1.15's comparator is `data[a*m+d] < data[b*m+d]`; a tie-broken comparator yields a different — wrong — permutation)
    const int32_t half = (end - start) / 2;
    auto cmp = [data, d](int32_t a, int32_t b) { return data[(size_t)a * 2 + d] < data[(size_t)b * 2 + d]; };
    std::nth_element(indices + start, indices + start + half, indices + end, cmp);
    double split = data[(size_t)indices[start + half] * 2 + d];
    int32_t p = start, q = end - 1;
    while (p <= q) {
        if (data[(size_t)indices[p] * 2 + d] < split) ++p;
        else if (data[(size_t)indices[q] * 2 + d] >= split) --q;
        else { std::swap(indices[p], indices[q]); ++p; --q; }
    }
    if (p == start) {
        // no point below the median value (it equals the node's minimum: duplicates).  scipy 1.15 moves the split to the next
        // representable double above it and partitions again, so every copy of the minimum goes left (observed: tree.split ==
        // nextafter(value, inf) on such nodes); older releases slid the split to a single point instead
        split = std::nextafter(split, __builtin_inf());
        p = start; q = end - 1;
        while (p <= q) {
            if (data[(size_t)indices[p] * 2 + d] < split) ++p;
            else if (data[(size_t)indices[q] * 2 + d] >= split) --q;
            else { std::swap(indices[p], indices[q]); ++p; --q; }
        }
    }
    const int32_t less = build(t, start, p, maxes, mins, leafsize);
    const int32_t greater = build(t, p, end, maxes, mins, leafsize);
    Node& nd = t.nodes[(size_t)node_index];
    nd.split_dim = d; nd.split = split; nd.less = less; nd.greater = greater;
    return node_index;
}

// scipy's `struct heap` (ckdtree/src/ordered_pair.h / query.cxx): a min-heap on `priority`
struct HeapItem { double priority; int64_t content; };
struct Heap {
    std::vector<HeapItem> h;
    int64_t n = 0;
    explicit Heap(size_t initial) { h.resize(initial ? initial : 1); }
    void push(const HeapItem& item) {
        ++n;
        if ((size_t)n > h.size()) h.resize(2 * h.size() + 1);
        int64_t i = n - 1;
        h[(size_t)i] = item;
        while (i > 0 && h[(size_t)i].priority < h[(size_t)((i - 1) / 2)].priority) {
            std::swap(h[(size_t)((i - 1) / 2)], h[(size_t)i]);
            i = (i - 1) / 2;
        }
    }
    const HeapItem& peek() const { return h[0]; }
    void remove() {
        h[0] = h[(size_t)(n - 1)];
        --n;
        int64_t i = 0, j = 1, k = 2;
        while ((j < n && h[(size_t)i].priority > h[(size_t)j].priority) || (k < n && h[(size_t)i].priority > h[(size_t)k].priority)) {
            const int64_t l = (k < n && h[(size_t)j].priority > h[(size_t)k].priority) ? k : j;
            std::swap(h[(size_t)l], h[(size_t)i]);
            i = l; j = 2 * i + 1; k = 2 * i + 2;
        }
    }
    HeapItem pop() { const HeapItem it = h[0]; remove(); return it; }
};

struct NodeInfo { int32_t node; double side[2]; double min_distance; };

}  // namespace detail

// scipy.spatial.cKDTree(points, leafsize) with compact_nodes = balanced_tree = True; points [n][2]
inline void build(Tree& t, const double* points, int32_t n, int32_t leafsize = 10) {
    t.n = n;
    t.data.assign(points, points + (size_t)n * 2);
    t.idx.resize((size_t)n);
    for (int32_t i = 0; i < n; ++i) t.idx[(size_t)i] = i;
    t.nodes.clear();
    if (n == 0) return;
    for (int i = 0; i < 2; ++i) t.mins[i] = t.maxes[i] = points[i];
    for (int32_t j = 1; j < n; ++j)
        for (int i = 0; i < 2; ++i) {
            t.mins[i] = std::min(t.mins[i], points[(size_t)j * 2 + i]);
            t.maxes[i] = std::max(t.maxes[i], points[(size_t)j * 2 + i]);
        }
    double maxes[2] = {t.maxes[0], t.maxes[1]}, mins[2] = {t.mins[0], t.mins[1]};
    detail::build(t, 0, n, maxes, mins, leafsize);
}

// tree.query(x, k=kmax, distance_upper_bound=dub) for ONE point: out_idx[kmax] in scipy's output order, n (= "missing") where
// fewer than kmax points lie within dub; out_d2 (nullable) the squared distances (inf for missing).  `pool` is scratch.
inline void query(const Tree& t, const double x[2], int32_t kmax, double dub, int32_t* out_idx, double* out_d2,
                  std::vector<detail::NodeInfo>& pool) {
    using namespace detail;
    const double* data = t.data.data();
    for (int32_t i = 0; i < kmax; ++i) { out_idx[i] = t.n; if (out_d2) out_d2[i] = __builtin_inf(); }
    if (t.n == 0) return;
    Heap q(12), neighbors((size_t)kmax);
    pool.clear();
    pool.reserve(t.nodes.size() + 1);        // pointers into the pool stay valid: at most one NodeInfo per tree node
    pool.push_back(NodeInfo{0, {0.0, 0.0}, 0.0});
    NodeInfo* ni1 = &pool[0];
    for (int i = 0; i < 2; ++i) {            // distance of the query to the root's bounding box (0 for a data point)
        const double s = std::max(0.0, std::max(x[i] - t.maxes[i], t.mins[i] - x[i]));
        ni1->side[i] = s * s;
        ni1->min_distance += ni1->side[i];
    }
    double bound = dub * dub;
    for (;;) {
        const Node& node = t.nodes[(size_t)ni1->node];
        if (node.split_dim == -1) {
            for (int32_t i = node.start; i < node.end; ++i) {
                const int32_t j = t.idx[(size_t)i];
                const double dx = data[(size_t)j * 2] - x[0], dy = data[(size_t)j * 2 + 1] - x[1];
                const double d = dx * dx + dy * dy;
                if (d < bound) {
                    if (neighbors.n == kmax) neighbors.remove();
                    neighbors.push(HeapItem{-d, (int64_t)j});
                    if (neighbors.n == kmax) bound = -neighbors.peek().priority;
                }
            }
            if (q.n == 0) break;
            ni1 = &pool[(size_t)q.pop().content];
        } else {
            if (ni1->min_distance > bound) break;      // the nearest remaining cell is too far: done
            pool.push_back(*ni1);                      // ni2 = copy of ni1 (init_plain)
            NodeInfo* ni2 = &pool.back();
            const int d = node.split_dim;
            double side;
            if (x[d] < node.split) { ni1->node = node.less; ni2->node = node.greater; side = node.split - x[d]; }
            else { ni1->node = node.greater; ni2->node = node.less; side = x[d] - node.split; }
            side = side * side;
            ni2->min_distance += side - ni2->side[d];
            ni2->side[d] = side;
            if (ni1->min_distance > ni2->min_distance) std::swap(ni1, ni2);
            if (ni2->min_distance <= bound) q.push(HeapItem{ni2->min_distance, (int64_t)(ni2 - pool.data())});
        }
    }
    // heapsort: furthest first out of the heap, filled from the back
    const int64_t nnb = neighbors.n;
    for (int64_t i = nnb - 1; i >= 0; --i) {
        const HeapItem it = neighbors.pop();
        out_idx[i] = (int32_t)it.content;
        if (out_d2) out_d2[i] = -it.priority;
    }
}

}  // namespace srh_kd
