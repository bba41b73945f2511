// pt_host.cpp -- host-side scene construction: the CPU code of the reference that produces
// the kernel's inputs (BVH build, plane tessellation, sphere packing, .obj ingestion).
// No GPU calls.  Built with -ffp-contract=off: Rust never fuses a*b+c, and the BVH must come
// out identical to the reference's, node for node.
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/ptrt.h"
#include "pt_layout.h"

namespace {

thread_local std::string g_host_error;

// for the catch-all guards: a message short enough for the string's in-place buffer, so that reporting an
// allocation failure does not allocate
int host_fail_static(int code, const char *msg) noexcept
{
    try { g_host_error.assign(msg); } catch (...) {}
    return code;
}

int host_fail(int code, const std::string &msg)
{
    g_host_error = msg;
    return code;
}

// Triangle::centroid, src/model.rs:42-48
inline float centroid_axis(const PtTriangle &t, int axis) { return (t.a[axis] + t.b[axis] + t.c[axis]) / 3.0f; }

inline PtNode empty_node()
{
    PtNode n;  // Node::default, bvh.rs:16-25
    for (int ax = 0; ax < 3; ax++) {
        n.min_corner[ax] = 3.40282347e+38f;
        n.max_corner[ax] = -3.40282347e+38f;
    }
    n.left_child_index = 0;
    n.triangle_count = 0;
    return n;
}

// Builds the tree of Bvh::from_triangles (src/scene/bvh.rs:34-162).  The node numbering is that of the
// reference's recursion: a split takes the next two free slots for its children, then the whole left
// subtree is built before the right one.  Big inputs are built by several threads with the same result:
// the top of the tree is split one node at a time (largest first; bounds of big nodes reduced in
// parallel), every remaining subtree is built independently with numbers relative to its root, and a last
// pass over the top of the tree hands out the global numbers in recursion order and relocates the
// subtrees.
struct BvhBuilder {
    uint32_t *idx;
    const PtTriangle *tris;
    unsigned n_threads = 1;

    // Bvh::update_bounds, src/scene/bvh.rs:68-89, over idx[first .. first+count).  f32::min/max ignore a NaN
    // operand; which of +0/-0 they return on a tie is not specified, here the earlier value stays.
    void bounds_of(uint32_t first, uint32_t count, float mn[3], float mx[3]) const
    {
        for (uint32_t i = 0; i < count; i++) {
            const PtTriangle &t = tris[idx[first + i]];
            const float *verts[3] = {t.a, t.b, t.c};
            for (const float *v : verts)
                for (int ax = 0; ax < 3; ax++) {
                    if (v[ax] < mn[ax]) mn[ax] = v[ax];
                    if (v[ax] > mx[ax]) mx[ax] = v[ax];
                }
        }
    }

    void update_bounds(PtNode &n) const
    {
        const uint32_t first = n.left_child_index, count = n.triangle_count;
        const unsigned parts = (n_threads > 1 && count >= (1u << 16)) ? n_threads : 1;
        if (parts == 1) {
            bounds_of(first, count, n.min_corner, n.max_corner);
            return;
        }
        // chunks in index order, merged in that order with the same strict comparisons: on a tie the value
        // of the earliest triangle stays, as in the single loop
        std::vector<PtNode> part(parts, empty_node());
        std::vector<std::thread> pool;
        for (unsigned k = 0; k < parts; k++) {
            const uint32_t lo = (uint32_t)((uint64_t)count * k / parts), hi = (uint32_t)((uint64_t)count * (k + 1) / parts);
            pool.emplace_back([this, &part, k, first, lo, hi] { bounds_of(first + lo, hi - lo, part[k].min_corner, part[k].max_corner); });
        }
        for (std::thread &t : pool) t.join();
        for (unsigned k = 0; k < parts; k++)
            for (int ax = 0; ax < 3; ax++) {
                if (part[k].min_corner[ax] < n.min_corner[ax]) n.min_corner[ax] = part[k].min_corner[ax];
                if (part[k].max_corner[ax] > n.max_corner[ax]) n.max_corner[ax] = part[k].max_corner[ax];
            }
    }

    // Bvh::subdivide, src/scene/bvh.rs:91-162, for one node: false when it stays a leaf, otherwise its
    // triangle range is partitioned and the two children come back with ranges and bounds.
    bool split(const PtNode &node, PtNode &left, PtNode &right) const
    {
        if (node.triangle_count <= 2) return false;
        const float extent[3] = {node.max_corner[0] - node.min_corner[0], node.max_corner[1] - node.min_corner[1],
                                 node.max_corner[2] - node.min_corner[2]};
        int axis = 0;
        if (extent[1] > extent[axis]) axis = 1;
        if (extent[2] > extent[axis]) axis = 2;
        const float split_position = node.min_corner[axis] + extent[axis] / 2.0f;

        // the reference's loop stops at i == j, so the element there is never compared and
        // lands on the right-hand side
        uint32_t i = node.left_child_index, j = node.left_child_index + node.triangle_count - 1u;
        while (i < j) {
            if (centroid_axis(tris[idx[i]], axis) < split_position) i++;
            else std::swap(idx[i], idx[j--]);
        }
        const uint32_t left_count = i - node.left_child_index;
        if (left_count == 0 || left_count == node.triangle_count) return false;
        left = right = empty_node();
        left.left_child_index = node.left_child_index;
        left.triangle_count = left_count;
        right.left_child_index = i;
        right.triangle_count = node.triangle_count - left_count;
        update_bounds(left);
        update_bounds(right);
        return true;
    }

    // The subtree below nodes[root], numbered from `used` on.  Same visiting order as the reference's
    // recursion with an explicit stack, so clustered input cannot overflow the call stack.
    template <typename Nodes>
    void build_subtree(Nodes &nodes, uint32_t root, uint32_t &used) const
    {
        std::vector<uint32_t> todo{root};
        PtNode l, r;
        while (!todo.empty()) {
            const uint32_t ni = todo.back();
            todo.pop_back();
            if (!split(nodes[ni], l, r)) continue;
            const uint32_t li = used;
            used += 2;
            store(nodes, li, l, r);
            nodes[ni].left_child_index = li;
            nodes[ni].triangle_count = 0;
            todo.push_back(li + 1);
            todo.push_back(li);
        }
    }
    static void store(PtNode *nodes, uint32_t li, const PtNode &l, const PtNode &r) { nodes[li] = l; nodes[li + 1] = r; }
    static void store(std::vector<PtNode> &nodes, uint32_t li, const PtNode &l, const PtNode &r)
    {
        if (nodes.size() < (size_t)li + 2) nodes.resize((size_t)li + 2);
        nodes[li] = l;
        nodes[li + 1] = r;
    }

    // returns the number of nodes used
    uint32_t build(PtNode *nodes, uint32_t n_tris)
    {
        PtNode root = empty_node();
        root.triangle_count = n_tris;
        update_bounds(root);
        if (n_threads <= 1 || n_tris < (1u << 16)) {
            nodes[0] = root;
            uint32_t used = 1;
            build_subtree(nodes, 0, used);
            return used;
        }

        // the top of the tree, split largest node first until there is enough independent work
        struct Top {
            PtNode rec;
            int left = -1, right = -1;  // children in `top`, or -1: a subtree for the threads (or a leaf)
            bool leaf = false;          // its split was tried here and failed: it stays as it is
        };
        std::vector<Top> top;
        top.push_back({root});
        auto smaller = [&top](int a, int b) { return top[a].rec.triangle_count < top[b].rec.triangle_count; };
        std::vector<int> open{0};  // heap of unsplit top nodes, largest on top
        const size_t wanted = (size_t)n_threads * 16;
        while (!open.empty() && open.size() < wanted) {
            std::pop_heap(open.begin(), open.end(), smaller);
            const int t = open.back();
            if (top[t].rec.triangle_count < 4096u) break;  // (the largest is small: nothing left worth splitting here)
            open.pop_back();
            PtNode l, r;
            if (!split(top[t].rec, l, r)) {  // stays a leaf; a second try would permute its triangle range again
                top[t].leaf = true;
                continue;
            }
            top[t].left = (int)top.size();
            top.push_back({l});
            top[t].right = (int)top.size();
            top.push_back({r});
            for (int c : {top[t].left, top[t].right}) {
                open.push_back(c);
                std::push_heap(open.begin(), open.end(), smaller);
            }
        }

        // every top node without children: its subtree, numbered from 1 relative to its root at 0
        std::vector<int> jobs;
        for (size_t t = 0; t < top.size(); t++)
            if (top[t].left < 0 && !top[t].leaf) jobs.push_back((int)t);
        std::sort(jobs.begin(), jobs.end(), [&top](int a, int b) { return top[a].rec.triangle_count > top[b].rec.triangle_count; });
        std::vector<std::vector<PtNode>> sub(top.size());
        std::atomic<size_t> next{0};
        const unsigned serial_threads = n_threads;
        n_threads = 1;  // inside the jobs bounds are computed by the job's own thread
        {
            std::vector<std::thread> pool;
            for (unsigned k = 0; k < serial_threads; k++)
                pool.emplace_back([&] {
                    for (size_t j; (j = next.fetch_add(1)) < jobs.size();) {
                        std::vector<PtNode> &mine = sub[jobs[j]];
                        mine.assign(1, top[jobs[j]].rec);
                        uint32_t used = 1;
                        build_subtree(mine, 0, used);
                        mine.resize(used);
                    }
                });
            for (std::thread &t : pool) t.join();
        }
        n_threads = serial_threads;

        // global numbers in recursion order: a split node's children take the next two slots, then come all
        // nodes below the left child, then all below the right one
        struct Move {
            int job;
            uint32_t base;  // global number of the subtree's local node 1
        };
        std::vector<Move> moves;
        uint32_t used = 1;
        nodes[0] = top[0].rec;
        std::vector<std::pair<int, uint32_t>> todo{{0, 0u}};  // (top node, its global number)
        while (!todo.empty()) {
            const auto [t, ni] = todo.back();
            todo.pop_back();
            if (top[t].left >= 0) {
                const uint32_t li = used;
                used += 2;
                nodes[ni] = top[t].rec;
                nodes[ni].left_child_index = li;
                nodes[ni].triangle_count = 0;
                todo.push_back({top[t].right, li + 1});
                todo.push_back({top[t].left, li});
            } else {
                if (top[t].leaf) {
                    nodes[ni] = top[t].rec;
                    continue;
                }
                moves.push_back({t, used});
                nodes[ni] = sub[t][0];
                if (sub[t].size() > 1) nodes[ni].left_child_index += used - 1u;  // an interior root: its children's number
                used += (uint32_t)sub[t].size() - 1u;
            }
        }
        next = 0;
        {
            std::vector<std::thread> pool;
            for (unsigned k = 0; k < n_threads; k++)
                pool.emplace_back([&] {
                    for (size_t m; (m = next.fetch_add(1)) < moves.size();) {
                        const std::vector<PtNode> &from = sub[moves[m].job];
                        const uint32_t shift = moves[m].base - 1u;
                        for (size_t k2 = 1; k2 < from.size(); k2++) {
                            PtNode n = from[k2];
                            if (n.triangle_count == 0) n.left_child_index += shift;  // interior: children's number; leaves keep their triangle range
                            nodes[moves[m].base + (k2 - 1)] = n;
                        }
                    }
                });
            for (std::thread &t : pool) t.join();
        }
        return used;
    }
};

struct ObjVertexRef {
    int v, vn;
};

bool parse_index(const char *&s, int count, int &out)
{
    char *end = nullptr;
    const long raw = std::strtol(s, &end, 10);
    if (end == s) return false;
    s = end;
    // range-checked as a long: a narrowing cast first would let 4294967297 wrap to a valid index
    if (raw == 0 || raw > (long)count || raw < -(long)count) return false;
    out = raw > 0 ? (int)(raw - 1) : count + (int)raw;
    return true;
}

// ---- Radiance .hdr (RGBE) decoding ------------------------------------------------------
// The reference hands the file's bytes to image 0.24.7's HdrDecoder in strict mode and converts every
// pixel with Rgbe8Pixel::to_hdr (src/texture.rs:179-188).  That crate is not part of the reference
// tree; this follows its published behaviour: "#?RADIANCE" signature, header lines up to an empty
// line, FORMAT must be 32-bit_rle_rgbe, dimensions "-Y <height> +X <width>", per-scanline choice
// between the per-component run-length coding (first pixel 2,2,<128,*) and the old coding in which
// a 1,1,1,n pixel repeats the previous one n times (consecutive markers multiply by 256).
struct HdrReader {
    const uint8_t *p;
    size_t n, at;
    bool byte(uint8_t &b)
    {
        if (at >= n) return false;
        b = p[at++];
        return true;
    }
    bool bytes(uint8_t *dst, size_t count)
    {
        if (n - at < count) return false;
        std::memcpy(dst, p + at, count);
        at += count;
        return true;
    }
    // one line without its '\n'; false at the end of the data with nothing read
    bool line(std::string &out)
    {
        out.clear();
        if (at >= n) return false;
        while (at < n && p[at] != '\n') out.push_back((char)p[at++]);
        if (at < n) at++;
        return true;
    }
};

std::string trimmed(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) a++;
    while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

bool parse_u32_strict(const std::string &s, uint32_t &v)
{
    // what str::parse::<u32> takes: an optional '+', then digits only, no overflow
    size_t i = (!s.empty() && s[0] == '+') ? 1 : 0;
    if (i >= s.size()) return false;
    uint64_t acc = 0;
    for (; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        acc = acc * 10 + (uint64_t)(s[i] - '0');
        if (acc > 0xFFFFFFFFull) return false;
    }
    v = (uint32_t)acc;
    return true;
}

// one colour component of a run-length coded scanline: literal groups of up to 128 values, runs of up to 127
bool hdr_component(HdrReader &r, uint8_t *rgbe, size_t width, int comp, std::string &why)
{
    size_t pos = 0;
    uint8_t buf[128];
    while (pos < width) {
        uint8_t rl;
        if (!r.byte(rl)) { why = "data ends inside a scanline"; return false; }
        if (rl == 0) { why = "zero-length group in a scanline"; return false; }  // (the crate would never finish here)
        if (rl <= 128) {
            if (pos + rl > width) { why = "scanline longer than the image is wide"; return false; }
            if (!r.bytes(buf, rl)) { why = "data ends inside a scanline"; return false; }
            for (size_t k = 0; k < rl; k++) rgbe[4 * (pos + k) + comp] = buf[k];
        } else {
            rl = (uint8_t)(rl - 128);
            if (pos + rl > width) { why = "scanline longer than the image is wide"; return false; }
            uint8_t value;
            if (!r.byte(value)) { why = "data ends inside a scanline"; return false; }
            for (size_t k = 0; k < rl; k++) rgbe[4 * (pos + k) + comp] = value;
        }
        pos += rl;
    }
    return true;
}

bool hdr_scanline(HdrReader &r, uint8_t *rgbe, size_t width, std::string &why)
{
    uint8_t fb[4];
    if (!r.bytes(fb, 4)) { why = "data ends before a scanline"; return false; }
    if (fb[0] == 2 && fb[1] == 2 && fb[2] < 128) {
        for (int comp = 0; comp < 4; comp++)
            if (!hdr_component(r, rgbe, width, comp, why)) return false;
        return true;
    }
    // old coding
    if (fb[0] == 1 && fb[1] == 1 && fb[2] == 1) { why = "scanline starts with a run marker"; return false; }
    std::memcpy(rgbe, fb, 4);
    size_t x = 1, mult = 1;
    uint8_t prev[4] = {fb[0], fb[1], fb[2], fb[3]};
    while (x < width) {
        uint8_t pix[4];
        if (!r.bytes(pix, 4)) { why = "data ends inside a scanline"; return false; }
        if (pix[0] == 1 && pix[1] == 1 && pix[2] == 1) {
            const size_t rl = (size_t)pix[3] * mult;
            mult *= 256;
            if (x + rl > width) { why = "scanline longer than the image is wide"; return false; }
            for (size_t k = 0; k < rl; k++) std::memcpy(rgbe + 4 * (x + k), prev, 4);
            x += rl;
        } else {
            mult = 1;
            std::memcpy(prev, pix, 4);
            std::memcpy(rgbe + 4 * x, pix, 4);
            x += 1;
        }
    }
    return true;
}

}  // namespace

extern "C" {

const char *pt_host_error(void) { return g_host_error.c_str(); }

static int64_t pt_bvh_build_impl(const void *tris, uint32_t n_tris, void *nodes_out, uint32_t *idx_out)
{
    if (n_tris == 0) return 0;  // Bvh::from_triangles returns empty vectors, bvh.rs:35-41
    if (!tris || !nodes_out || !idx_out) return host_fail(PT_ERR_INVALID, "pt_bvh_build: null pointer");
    PtNode *nodes = static_cast<PtNode *>(nodes_out);
    for (uint32_t k = 0; k < n_tris; k++) idx_out[k] = k;
    BvhBuilder b{idx_out, static_cast<const PtTriangle *>(tris)};
    b.n_threads = std::min(std::max(std::thread::hardware_concurrency(), 1u), 16u);
    if (const char *env = std::getenv("PTRT_BUILD_THREADS")) b.n_threads = (unsigned)std::max(1, std::atoi(env));
    uint32_t used = 0;
    try {
        used = b.build(nodes, n_tris);
    } catch (const std::exception &e) {
        return host_fail(PT_ERR_NOMEM, std::string("pt_bvh_build: ") + e.what());
    }
    const size_t cap = (size_t)n_tris * 2u - 1u;
    for (size_t k = used; k < cap; k++) nodes[k] = empty_node();  // Node::default; the reference truncates them away, bvh.rs:44, 64
    return (int64_t)used;
}

static int pt_plane_triangles_impl(const float q[3], const float u[3], const float v[3], const float albedo[3], uint32_t material,
                       void *tris_out)
{
    if (!q || !u || !v || !albedo || !tris_out) return host_fail(PT_ERR_INVALID, "pt_plane_triangles: null pointer");
    // Plane::normal, plane.rs:43-45: cgmath cross then normalize = v * (1 / sqrt(x*x + y*y + z*z))
    const float cx = u[1] * v[2] - u[2] * v[1], cy = u[2] * v[0] - u[0] * v[2], cz = u[0] * v[1] - u[1] * v[0];
    const float inv = 1.0f / std::sqrt(cx * cx + cy * cy + cz * cz);
    const float n[3] = {cx * inv, cy * inv, cz * inv};
    float qu[3], qv[3], quv[3];
    for (int ax = 0; ax < 3; ax++) {
        qu[ax] = q[ax] + u[ax];
        qv[ax] = q[ax] + v[ax];
        quv[ax] = qu[ax] + v[ax];
    }
    PtTriangle *out = static_cast<PtTriangle *>(tris_out);
    std::memset(out, 0, 2 * sizeof(PtTriangle));
    for (int k = 0; k < 2; k++) {
        const float *first = (k == 0) ? q : quv;  // plane.rs:18-38
        for (int ax = 0; ax < 3; ax++) {
            out[k].a[ax] = first[ax];
            out[k].b[ax] = qu[ax];
            out[k].c[ax] = qv[ax];
            out[k].na[ax] = out[k].nb[ax] = out[k].nc[ax] = n[ax];
            out[k].albedo[ax] = albedo[ax];
        }
        out[k].material = material;
    }
    return PT_OK;
}

static int pt_pack_spheres_impl(const float *spheres8, uint32_t n, void *sphere_data_out)
{
    if ((!spheres8 && n) || !sphere_data_out) return host_fail(PT_ERR_INVALID, "pt_pack_spheres: null pointer");
    PtSphereData *out = static_cast<PtSphereData *>(sphere_data_out);
    std::memset(out, 0, sizeof(PtSphereData));
    const uint32_t kept = std::min<uint32_t>(n, PT_MAX_SPHERES);  // sphere.rs:116-125
    out->sphere_count = kept;
    if (kept) std::memcpy(out->spheres, spheres8, (size_t)kept * sizeof(PtSphere));
    return PT_OK;
}

static int pt_obj_load_impl(const char *path, void *tris_out, uint32_t capacity, uint32_t *n_tris)
{
    if (!path || !n_tris) return host_fail(PT_ERR_INVALID, "pt_obj_load: null pointer");
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return host_fail(PT_ERR_INVALID, std::string("pt_obj_load: cannot open ") + path);
    std::vector<float> pos, nrm;
    std::vector<ObjVertexRef> corners;  // three per triangle after fan triangulation
    std::vector<char> line(1 << 16);
    bool bad = false;
    std::string why;
    while (std::fgets(line.data(), (int)line.size(), fp)) {
        const char *s = line.data();
        while (*s == ' ' || *s == '\t') s++;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            char *e = nullptr;
            s += 2;
            for (int k = 0; k < 3; k++) { pos.push_back(std::strtof(s, &e)); if (e == s) { bad = true; why = "malformed v line"; } s = e; }
        } else if (s[0] == 'v' && s[1] == 'n' && (s[2] == ' ' || s[2] == '\t')) {
            char *e = nullptr;
            s += 3;
            for (int k = 0; k < 3; k++) { nrm.push_back(std::strtof(s, &e)); if (e == s) { bad = true; why = "malformed vn line"; } s = e; }
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            s += 2;
            std::vector<ObjVertexRef> poly;
            for (;;) {
                while (*s == ' ' || *s == '\t') s++;
                if (*s == '\0' || *s == '\n' || *s == '\r' || *s == '#') break;
                ObjVertexRef r{-1, -1};
                if (!parse_index(s, (int)(pos.size() / 3), r.v)) { bad = true; why = "bad vertex index in f line"; break; }
                if (*s == '/') {
                    s++;
                    if (*s != '/') { int vt; char *e; (void)vt; std::strtol(s, &e, 10); s = e; }  // texture index unused
                    if (*s == '/') {
                        s++;
                        if (!parse_index(s, (int)(nrm.size() / 3), r.vn)) { bad = true; why = "bad normal index in f line"; break; }
                    }
                }
                poly.push_back(r);
            }
            if (bad) break;
            for (size_t k = 1; k + 1 < poly.size(); k++) {  // fan: (0, k, k+1)
                corners.push_back(poly[0]);
                corners.push_back(poly[k]);
                corners.push_back(poly[k + 1]);
            }
        }
        if (bad) break;
    }
    std::fclose(fp);
    if (bad) return host_fail(PT_ERR_INVALID, "pt_obj_load: " + why);
    const size_t count = corners.size() / 3;
    if (count > 0xFFFFFFFFull) return host_fail(PT_ERR_LIMIT, "pt_obj_load: too many triangles");
    *n_tris = (uint32_t)count;
    if (!tris_out) return PT_OK;
    if (capacity < count) return host_fail(PT_ERR_INVALID, "pt_obj_load: output buffer too small");

    // vertices without a vn reference: area-weighted smooth normals per position index
    std::vector<float> smooth;
    bool need_smooth = false;
    for (const ObjVertexRef &r : corners) need_smooth |= (r.vn < 0);
    if (need_smooth) {
        smooth.assign(pos.size(), 0.0f);
        for (size_t t = 0; t < count; t++) {
            const float *a = &pos[3 * corners[3 * t].v], *b = &pos[3 * corners[3 * t + 1].v], *c = &pos[3 * corners[3 * t + 2].v];
            const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            const float fn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            for (int k = 0; k < 3; k++)
                for (int ax = 0; ax < 3; ax++) smooth[3 * corners[3 * t + k].v + ax] += fn[ax];
        }
        for (size_t i = 0; i + 2 < smooth.size(); i += 3) {
            const float len = std::sqrt(smooth[i] * smooth[i] + smooth[i + 1] * smooth[i + 1] + smooth[i + 2] * smooth[i + 2]);
            if (len > 0.0f) { smooth[i] /= len; smooth[i + 1] /= len; smooth[i + 2] /= len; }
            else { smooth[i] = 0.0f; smooth[i + 1] = 1.0f; smooth[i + 2] = 0.0f; }
        }
    }
    PtTriangle *out = static_cast<PtTriangle *>(tris_out);
    std::memset(out, 0, count * sizeof(PtTriangle));
    for (size_t t = 0; t < count; t++) {
        float *pdst[3] = {out[t].a, out[t].b, out[t].c}, *ndst[3] = {out[t].na, out[t].nb, out[t].nc};
        for (int k = 0; k < 3; k++) {
            const ObjVertexRef r = corners[3 * t + k];
            std::memcpy(pdst[k], &pos[3 * r.v], 12);
            std::memcpy(ndst[k], r.vn >= 0 ? &nrm[3 * r.vn] : &smooth[3 * r.v], 12);
        }
        out[t].albedo[0] = out[t].albedo[1] = out[t].albedo[2] = 1.0f;  // model.rs:199-200
        out[t].material = 0;
    }
    return PT_OK;
}

static int pt_hdr_decode_impl(const void *data, size_t n_bytes, float *rgba_out, size_t capacity_pixels, uint32_t *width, uint32_t *height)
{
    if (!data || !width || !height) return host_fail(PT_ERR_INVALID, "pt_hdr_decode: null pointer");
    HdrReader r{static_cast<const uint8_t *>(data), n_bytes, 0};
    static const char kSignature[] = "#?RADIANCE";
    if (n_bytes < 10 || std::memcmp(r.p, kSignature, 10) != 0) return host_fail(PT_ERR_INVALID, "pt_hdr_decode: Radiance HDR signature not found");
    r.at = 10;
    std::string line;
    r.line(line);  // rest of the signature line
    for (;;) {
        if (!r.line(line)) return host_fail(PT_ERR_INVALID, "pt_hdr_decode: data ends inside the header");
        if (line.empty()) break;
        if (line[0] == '#') continue;
        const size_t eq = line.find('=');
        if (eq != std::string::npos && line.compare(0, eq, "FORMAT") == 0 && trimmed(line.substr(eq + 1)) != "32-bit_rle_rgbe")
            return host_fail(PT_ERR_INVALID, "pt_hdr_decode: unsupported FORMAT " + trimmed(line.substr(eq + 1)));
        // EXPOSURE, PIXASPECT, COLORCORR and unknown keys do not change the decoded pixels
    }
    if (!r.line(line)) return host_fail(PT_ERR_INVALID, "pt_hdr_decode: data ends before the dimensions line");
    std::vector<std::string> parts;
    for (size_t i = 0; i < line.size();) {
        while (i < line.size() && std::isspace((unsigned char)line[i])) i++;
        size_t j = i;
        while (j < line.size() && !std::isspace((unsigned char)line[j])) j++;
        if (j > i) parts.push_back(line.substr(i, j - i));
        i = j;
    }
    uint32_t w = 0, h = 0;
    if (parts.size() != 4) return host_fail(PT_ERR_INVALID, "pt_hdr_decode: malformed dimensions line");
    if (parts[0] != "-Y" || parts[2] != "+X") return host_fail(PT_ERR_INVALID, "pt_hdr_decode: unsupported orientation " + parts[0] + " " + parts[2]);
    if (!parse_u32_strict(parts[1], h) || !parse_u32_strict(parts[3], w)) return host_fail(PT_ERR_INVALID, "pt_hdr_decode: malformed dimensions line");
    if (w == 0 || h == 0) return host_fail(PT_ERR_INVALID, "pt_hdr_decode: empty image");
    if ((uint64_t)w * h * 3 > 0xFFFFFFFFull) return host_fail(PT_ERR_LIMIT, "pt_hdr_decode: image dimensions too large");
    *width = w;
    *height = h;
    if (!rgba_out) return PT_OK;
    if (capacity_pixels < (size_t)w * h) return host_fail(PT_ERR_INVALID, "pt_hdr_decode: output buffer too small");
    std::vector<uint8_t> row((size_t)w * 4);
    std::string why;
    for (uint32_t y = 0; y < h; y++) {
        if (!hdr_scanline(r, row.data(), w, why)) return host_fail(PT_ERR_INVALID, "pt_hdr_decode: " + why + " (row " + std::to_string(y) + ")");
        float *dst = rgba_out + (size_t)y * w * 4;
        for (uint32_t x = 0; x < w; x++) {
            // Rgbe8Pixel::to_hdr: mantissa * 2^(e - 136), black when e == 0; alpha 1 (texture.rs:184-185)
            const uint8_t *px = &row[4 * (size_t)x];
            const float scale = px[3] ? std::ldexp(1.0f, (int)px[3] - 136) : 0.0f;
            dst[4 * x + 0] = scale * (float)px[0];
            dst[4 * x + 1] = scale * (float)px[1];
            dst[4 * x + 2] = scale * (float)px[2];
            dst[4 * x + 3] = 1.0f;
        }
    }
    return PT_OK;
}


// Nothing may propagate out of an extern "C" function: a C or Rust caller cannot catch it.
int64_t pt_bvh_build(const void *tris, uint32_t n_tris, void *nodes_out, uint32_t *idx_out)
{
    try {
        return pt_bvh_build_impl(tris, n_tris, nodes_out, idx_out);
    } catch (const std::bad_alloc &) {
        return host_fail_static(PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return host_fail_static(PT_ERR_INTERNAL, "internal error");
    }
}
int pt_plane_triangles(const float q[3], const float u[3], const float v[3], const float albedo[3], uint32_t material,
                       void *tris_out)
{
    try {
        return pt_plane_triangles_impl(q, u, v, albedo, material, tris_out);
    } catch (const std::bad_alloc &) {
        return host_fail_static(PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return host_fail_static(PT_ERR_INTERNAL, "internal error");
    }
}
int pt_pack_spheres(const float *spheres8, uint32_t n, void *sphere_data_out)
{
    try {
        return pt_pack_spheres_impl(spheres8, n, sphere_data_out);
    } catch (const std::bad_alloc &) {
        return host_fail_static(PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return host_fail_static(PT_ERR_INTERNAL, "internal error");
    }
}
int pt_obj_load(const char *path, void *tris_out, uint32_t capacity, uint32_t *n_tris)
{
    try {
        return pt_obj_load_impl(path, tris_out, capacity, n_tris);
    } catch (const std::bad_alloc &) {
        return host_fail_static(PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return host_fail_static(PT_ERR_INTERNAL, "internal error");
    }
}
int pt_hdr_decode(const void *data, size_t n_bytes, float *rgba_out, size_t capacity_pixels, uint32_t *width, uint32_t *height)
{
    try {
        return pt_hdr_decode_impl(data, n_bytes, rgba_out, capacity_pixels, width, height);
    } catch (const std::bad_alloc &) {
        return host_fail_static(PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return host_fail_static(PT_ERR_INTERNAL, "internal error");
    }
}
}  // extern "C"
