// pt_bvh_sah.cpp -- a second BVH builder over the reference's node format (src/scene/bvh.rs:7-14: 32-byte nodes,
// children adjacent at left_child_index and left_child_index + 1, leaves with a range of the index permutation).
//
// The reference splits every node at the spatial midpoint of its longest axis (bvh.rs:101-124) down to two triangles.
// That tree is what pt_bvh_build reproduces node for node, and what every parity statement of this library refers to.
// This builder keeps the format and changes the splits: binned surface-area heuristic (16 bins per axis over the
// centroids, all three axes tried, cost = one step per node visit + one step per triangle tested), leaves of up to
// four triangles.  A traversal of it returns the same closest hit as a traversal of the reference tree whenever
// the hit is unambiguous; the two can differ where the shader's own arithmetic is ambiguous -- two triangles at exactly
// the same t (first tested wins, compute.wgsl:286-290, and the order of tests is the tree's), and rays that graze a
// node's box, where the float slab test (compute.wgsl:441-456) and the triangle test disagree about a hit.  Opt-in
// only, never the default.  No GPU calls.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ptrt.h"
#include "pt_layout.h"

namespace {

struct Box {
    float mn[3], mx[3];
    void reset()
    {
        for (int a = 0; a < 3; a++) { mn[a] = 3.40282347e+38f; mx[a] = -3.40282347e+38f; }
    }
    void grow(const Box &o)
    {
        for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], o.mn[a]); mx[a] = std::max(mx[a], o.mx[a]); }
    }
    float half_area() const
    {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;  // empty
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kBins = 16;         // (32 bins, leaves of 2 or 8, a node visit priced at 0.5 to 2 triangle tests: all within 2 % in steps per ray)
constexpr int kBinsMax = kBins;
#ifdef PT_TUNING
static uint32_t tuned_leaf() { const char *e = getenv("PTRT_GUARD_LEAF"); return e ? (uint32_t)atoi(e) : 4u; }
static const uint32_t kMaxLeaf = tuned_leaf();  // (tuning builds: leaf size of every tree this builder makes)
#else
constexpr uint32_t kMaxLeaf = 4;
#endif
constexpr float kTrav = 1.0f;

struct SahBuilder {
    const PtTriangle *tris;
    uint32_t *idx;
    std::vector<Box> tb;          // per triangle: its bounds
    std::vector<float> cen;       // per triangle: centroid of the bounds, 3 floats
    PtNode *nodes;
    uint32_t used = 0;

    void node_bounds(uint32_t first, uint32_t count, Box &b, Box &cb) const
    {
        b.reset();
        cb.reset();
        for (uint32_t i = 0; i < count; i++) {
            const uint32_t t = idx[first + i];
            b.grow(tb[t]);
            for (int a = 0; a < 3; a++) {
                cb.mn[a] = std::min(cb.mn[a], cen[3 * (size_t)t + a]);
                cb.mx[a] = std::max(cb.mx[a], cen[3 * (size_t)t + a]);
            }
        }
    }

    // Best binned split of idx[first .. first+count): axis, the bin boundary, its cost in steps relative to the node's area.
    bool best_split(uint32_t first, uint32_t count, const Box &b, const Box &cb, int &axis_out, int &bin_out, float &cost_out) const
    {
        float best = 3.40282347e+38f;
        bool found = false;
        for (int axis = 0; axis < 3; axis++) {
            const float lo = cb.mn[axis], ext = cb.mx[axis] - cb.mn[axis];
            if (!(ext > 0.0f)) continue;
            Box bins[kBinsMax];
            uint32_t n[kBinsMax];
            for (int k = 0; k < kBins; k++) { bins[k].reset(); n[k] = 0; }
            const float scale = (float)kBins / ext;
            for (uint32_t i = 0; i < count; i++) {
                const uint32_t t = idx[first + i];
                int k = (int)((cen[3 * (size_t)t + axis] - lo) * scale);
                k = std::min(std::max(k, 0), kBins - 1);
                bins[k].grow(tb[t]);
                n[k]++;
            }
            float right_area[kBinsMax];
            uint32_t right_n[kBinsMax];
            Box acc;
            acc.reset();
            uint32_t cnt = 0;
            for (int k = kBins - 1; k > 0; k--) {
                acc.grow(bins[k]);
                cnt += n[k];
                right_area[k] = acc.half_area();
                right_n[k] = cnt;
            }
            acc.reset();
            cnt = 0;
            for (int k = 1; k < kBins; k++) {  // split between bins k-1 and k
                acc.grow(bins[k - 1]);
                cnt += n[k - 1];
                if (cnt == 0 || right_n[k] == 0) continue;
                const float c = acc.half_area() * (float)cnt + right_area[k] * (float)right_n[k];
                if (c < best) { best = c; axis_out = axis; bin_out = k; found = true; }
            }
        }
        const float area = b.half_area();
        cost_out = (found && area > 0.0f) ? kTrav + best / area : 3.40282347e+38f;
        return found;
    }

    // Splits the node `nd` (a leaf over idx[first .. first+count)) if the heuristic says so: partitions its range of idx in
    // place and fills the two children.  False: it stays a leaf.
    bool split(const PtNode &nd, PtNode &left, PtNode &right)
    {
        const uint32_t first = nd.left_child_index, count = nd.triangle_count;
        if (count <= 1) return false;
        Box b, cb;
        node_bounds(first, count, b, cb);
        int axis = 0, bin = 0;
        float cost = 0.0f;
        uint32_t mid;
        const bool have = best_split(first, count, b, cb, axis, bin, cost);
        if (have && (count > kMaxLeaf || cost < (float)count)) {
            const float lo = cb.mn[axis], scale = (float)kBins / (cb.mx[axis] - cb.mn[axis]);
            uint32_t *begin = idx + first, *end = begin + count;
            uint32_t *m = std::partition(begin, end, [&](uint32_t t) {
                int k = (int)((cen[3 * (size_t)t + axis] - lo) * scale);
                k = std::min(std::max(k, 0), kBins - 1);
                return k < bin;
            });
            mid = (uint32_t)(m - begin);
        } else if (count > kMaxLeaf) {
            mid = count / 2;  // all centroids in one place: halve by index, so that no leaf outgrows the device's inline leaves
        } else {
            return false;  // a leaf
        }
        if (mid == 0 || mid == count) return false;
        Box lb, rb, dummy;
        node_bounds(first, mid, lb, dummy);
        node_bounds(first + mid, count - mid, rb, dummy);
        put(left, lb, first, mid);
        put(right, rb, first + mid, count - mid);
        return true;
    }
    static void put(PtNode &n, const Box &bb, uint32_t first, uint32_t count)
    {
        for (int a = 0; a < 3; a++) { n.min_corner[a] = bb.mn[a]; n.max_corner[a] = bb.mx[a]; }
        n.left_child_index = first;
        n.triangle_count = count;
    }

    // The subtree below out[root] (a leaf over its range), depth first, children appended to `out` (numbers local to it).
    void build_below(std::vector<PtNode> &out, uint32_t root)
    {
        std::vector<uint32_t> todo{root};
        while (!todo.empty()) {
            const uint32_t ni = todo.back();
            todo.pop_back();
            PtNode l, r;
            if (!split(out[ni], l, r)) continue;
            const uint32_t li = (uint32_t)out.size();
            out.push_back(l);
            out.push_back(r);
            out[ni].left_child_index = li;
            out[ni].triangle_count = 0;
            todo.push_back(li + 1);  // (left subtree first, like the reference's recursion: parents before children)
            todo.push_back(li);
        }
    }

    // The whole tree into `nodes`.  Large inputs: the top of the tree level by level on this thread until there are
    // enough subtrees, those by `n_threads` workers into arrays of their own (every subtree owns its range of idx), then
    // appended in the order the top produced them -- the same tree whatever the number of threads.
    uint32_t build(uint32_t n_tris, unsigned n_threads)
    {
        Box b, cb;
        node_bounds(0, n_tris, b, cb);
        std::vector<PtNode> top(1);
        put(top[0], b, 0, n_tris);
        std::vector<uint32_t> open{0};  // leaves of the top that may still be split
        const bool big = n_tris >= 65536, parallel = big && n_threads > 1;
        if (big) {  // (whatever the number of threads: the numbering of the nodes must not depend on it)
            const size_t want = 128;
            while (!open.empty() && open.size() < want) {
                std::vector<uint32_t> next;
                bool any = false;
                for (uint32_t ni : open) {
                    PtNode l, r;
                    if (top[ni].triangle_count < 4096u || !split(top[ni], l, r)) { next.push_back(ni); continue; }
                    const uint32_t li = (uint32_t)top.size();
                    top.push_back(l);
                    top.push_back(r);
                    top[ni].left_child_index = li;
                    top[ni].triangle_count = 0;
                    next.push_back(li);
                    next.push_back(li + 1);
                    any = true;
                }
                open.swap(next);
                if (!any) break;
            }
        }
        std::vector<std::vector<PtNode>> sub(open.size());
        auto work = [&](size_t k) {
            sub[k].push_back(top[open[k]]);
            build_below(sub[k], 0);
        };
        if (parallel) {
            std::atomic<size_t> next_job{0};
            std::atomic<bool> failed{false};
            std::vector<std::thread> pool;
            auto loop = [&]() {
                try {
                    for (size_t k; (k = next_job.fetch_add(1)) < sub.size();) work(k);
                } catch (...) {
                    failed = true;
                }
            };
            for (unsigned t = 1; t < n_threads; t++) pool.emplace_back(loop);
            loop();
            for (std::thread &t : pool) t.join();
            if (failed) throw std::bad_alloc();
        } else {
            for (size_t k = 0; k < sub.size(); k++) work(k);
        }
        // the top as it is, every subtree's descendants behind it (its root already has a place in the top)
        used = (uint32_t)top.size();
        std::copy(top.begin(), top.end(), nodes);
        for (size_t k = 0; k < sub.size(); k++) {
            const std::vector<PtNode> &sv = sub[k];
            if (sv.size() == 1) continue;  // stayed a leaf
            const uint32_t base = used - 1u;  // local number j >= 1 becomes base + j
            PtNode root = sv[0];
            root.left_child_index += base;
            nodes[open[k]] = root;
            for (size_t j = 1; j < sv.size(); j++) {
                PtNode n = sv[j];
                if (n.triangle_count == 0) n.left_child_index += base;
                nodes[base + j] = n;
            }
            used += (uint32_t)sv.size() - 1u;
        }
        return used;
    }
};

thread_local std::string g_sah_error;

}  // namespace

extern "C" {

// (message of the last failure on this thread; pt_host_error() of pt_host.cpp reports its own helpers only)
const char *pt_bvh_build_sah_error(void) { return g_sah_error.c_str(); }

// Shared body of the two entry points.  `pad` (may be null) is a per-triangle distance by which that triangle's bounds
// are grown on every side before the tree is built over them.
static int64_t build_sah_tree(const char *who, const void *tris_v, uint32_t n_tris, const float *pad, void *nodes_out, uint32_t *idx_out)
{
    try {
        if (n_tris == 0) return 0;
        if (!tris_v || !nodes_out || !idx_out) { g_sah_error = std::string(who) + ": null pointer"; return PT_ERR_INVALID; }
        SahBuilder sb;
        sb.tris = static_cast<const PtTriangle *>(tris_v);
        sb.idx = idx_out;
        sb.nodes = static_cast<PtNode *>(nodes_out);
        sb.tb.resize(n_tris);
        sb.cen.resize(3 * (size_t)n_tris);
        for (uint32_t t = 0; t < n_tris; t++) {
            idx_out[t] = t;
            const PtTriangle &tr = sb.tris[t];
            Box &b = sb.tb[t];
            for (int a = 0; a < 3; a++) {
                b.mn[a] = std::min(std::min(tr.a[a], tr.b[a]), tr.c[a]);
                b.mx[a] = std::max(std::max(tr.a[a], tr.b[a]), tr.c[a]);
                sb.cen[3 * (size_t)t + a] = 0.5f * b.mn[a] + 0.5f * b.mx[a];
                if (pad) {  // outwards, and one more float outwards for the rounding of the subtraction itself
                    b.mn[a] = std::nextafter(b.mn[a] - pad[t], -3.40282347e+38f);
                    b.mx[a] = std::nextafter(b.mx[a] + pad[t], 3.40282347e+38f);
                }
            }
        }
        unsigned n_threads = std::min(std::max(std::thread::hardware_concurrency(), 1u), 16u);
        if (const char *env = std::getenv("PTRT_BUILD_THREADS")) n_threads = (unsigned)std::max(1, std::atoi(env));
        const uint32_t used = sb.build(n_tris, n_threads);
        const size_t cap = (size_t)n_tris * 2u - 1u;
        for (size_t k = used; k < cap; k++) {
            PtNode &n = sb.nodes[k];
            for (int a = 0; a < 3; a++) { n.min_corner[a] = 3.40282347e+38f; n.max_corner[a] = -3.40282347e+38f; }
            n.left_child_index = 0;
            n.triangle_count = 0;
        }
        return (int64_t)used;
    } catch (const std::bad_alloc &) {
        try { g_sah_error.assign("out of memory"); } catch (...) {}
        return PT_ERR_NOMEM;
    } catch (...) {
        try { g_sah_error.assign("internal error"); } catch (...) {}
        return PT_ERR_INTERNAL;
    }
}

int64_t pt_bvh_build_sah(const void *tris_v, uint32_t n_tris, void *nodes_out, uint32_t *idx_out)
{
    return build_sah_tree("pt_bvh_build_sah", tris_v, n_tris, nullptr, nodes_out, idx_out);
}

int64_t pt_bvh_build_sah_padded(const void *tris_v, uint32_t n_tris, const float *pad, void *nodes_out, uint32_t *idx_out)
{
    if (n_tris != 0 && !pad) { g_sah_error = "pt_bvh_build_sah_padded: null pointer"; return PT_ERR_INVALID; }
    return build_sah_tree("pt_bvh_build_sah_padded", tris_v, n_tris, pad, nodes_out, idx_out);
}

}  // extern "C"
