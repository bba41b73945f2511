// pt_kernels.hip -- gfx950 kernels of the path-tracing hot path.
//
// One lane owns one pixel and walks it through all requested frames: camera ray
// (compute.wgsl:83-128), bounce loop (:130-207), closest hit over spheres and the flattened
// BVH (:214-302, :304-406, :441-456), material scatter with the shader's hash RNG
// (utils.wgsl), sky cube lookup (:209-212) and the fp32 running sum.  Every floating-point
// step follows pt_math.h so results are bit-identical to a scalar evaluation of the same
// sequence; the file must be compiled with -ffp-contract=off and without fast-math.
//
// CDNA4 mapping: wave64, one wave = an 8x8 pixel tile; the traversal stack holds 4-byte node
// tokens in LDS laid out [entry][lane] (conflict-free for any mix of stack depths); the two
// children of an interior node are one 64-byte aligned fetch; leaf triangles are stored in
// leaf order (no index indirection) as {a, b-a, c-a}, normals are fetched once per closest hit;
// sphere records are wave-uniform loads.  A lane that finishes a path starts its next
// sample/frame immediately, so a wave only idles when every lane has run out of frames.
#include "pt_kernels.h"

#include <type_traits>

#include <algorithm>
#include <cstddef>
#include <cstdlib>

namespace {

struct Ray {
    v3 o, d;
};

__device__ __forceinline__ v3 f4_xyz(float4 a) { return V3(a.x, a.y, a.z); }

// ---- traversal stack ---------------------------------------------------------------------
// Entries 0..PT_LDS_STACK-1 live in LDS at [entry][lane]; deeper entries (only trees deeper
// than PT_LDS_STACK, and then only the rays that actually get that deep) spill to a
// per-thread column of a global buffer.
template <bool OVERFLOW, int BLOCK>
struct Stack {
    uint32_t *lds;     // &smem[threadIdx.x]; entry e at lds[e * BLOCK]
    uint32_t *spill;   // &gstack[global thread]; entry e at spill[(e - PT_LDS_STACK) * pitch]
    size_t pitch;
    __device__ __forceinline__ void push(uint32_t sp, uint32_t v)
    {
        if (!OVERFLOW || sp < PT_LDS_STACK) lds[sp * BLOCK] = v;
        else spill[(size_t)(sp - PT_LDS_STACK) * pitch] = v;
    }
    __device__ __forceinline__ uint32_t pop(uint32_t sp)
    {
        if (!OVERFLOW || sp < PT_LDS_STACK) return lds[sp * BLOCK];
        return spill[(size_t)(sp - PT_LDS_STACK) * pitch];
    }
};

struct Counters {
    uint32_t rays = 0, paths = 0, interior = 0, tri_tests = 0, tri_accept = 0, sphere_tests = 0, sky = 0, bounces = 0, sphere_accept = 0;
    uint32_t walks = 0;  // rays that went on into the BVH after the sphere loop and the root visit
    uint32_t guarded = 0, rewalks = 0;  // pooled kernel: rays that set out on the guard tree; of their hits, those sent through the reference tree again
};

// Closest-hit record kept in registers during traversal; the full record (point, normal,
// attenuation) is derived once from it afterwards.
struct Closest {
    bool hit;
    bool is_tri;
    float t;
    uint32_t index;  // sphere index, or the walk-table position of the triangle's record; after a walk, PT_ID_TIE on top of
                     // it if a later triangle was accepted by the triangle test at exactly the held t (which of the two a walk
                     // keeps is the order of its tests, i.e. the tree's): consumers mask with PT_ID_INDEX
    float u, v;
};

#define PT_ID_TIE 0x10000000u
#define PT_ID_ABANDONED 0x08000000u  // a walk of the guard tree's four-child form that ran out of stack: its record says nothing
#define PT_ID_INDEX 0x07FFFFFFu      // (a table position is below 2^27)

struct HitRecord {
    bool hit;
    float t;
    v3 p, normal;
    bool front;
    v3 att;
    float material;
    uint32_t prim;
};

// hitAabb, compute.wgsl:441-456, with 1/direction hoisted out (same correctly rounded value), for the
// two children of a node at once.  q0..q2 hold the pair interleaved {l, r} per component (see the node
// layout in pt_api.hip), so the subtractions and multiplications are two-wide packed instructions;
// per child they are the same operations in the same order as the one-box form.
typedef float pt_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void hit_aabb_pair(v3 o, v3 inv, float4 q0, float4 q1, float4 q2, float &dl, float &dr)
{
    const pt_f2 t1x = (pt_f2{q0.x, q0.y} - o.x) * inv.x, t1y = (pt_f2{q0.z, q0.w} - o.y) * inv.y, t1z = (pt_f2{q1.x, q1.y} - o.z) * inv.z;
    const pt_f2 t2x = (pt_f2{q1.z, q1.w} - o.x) * inv.x, t2y = (pt_f2{q2.x, q2.y} - o.y) * inv.y, t2z = (pt_f2{q2.z, q2.w} - o.z) * inv.z;
    const float tmin_l = pt_max(pt_max(pt_min(t1x.x, t2x.x), pt_min(t1y.x, t2y.x)), pt_min(t1z.x, t2z.x));
    const float tmax_l = pt_min(pt_min(pt_max(t1x.x, t2x.x), pt_max(t1y.x, t2y.x)), pt_max(t1z.x, t2z.x));
    const float tmin_r = pt_max(pt_max(pt_min(t1x.y, t2x.y), pt_min(t1y.y, t2y.y)), pt_min(t1z.y, t2z.y));
    const float tmax_r = pt_min(pt_min(pt_max(t1x.y, t2x.y), pt_max(t1y.y, t2y.y)), pt_max(t1z.y, t2z.y));
    dl = (tmin_l > tmax_l || tmax_l < 0.0f) ? 99999.0f : tmin_l;
    dr = (tmin_r > tmax_r || tmax_r < 0.0f) ? 99999.0f : tmin_r;
}

// The 64-byte record at walk-table position `at` (a sibling pair or a triangle: one table, positions in 32-byte
// units from sc.nodes).  The byte offset is a 32-bit value added to a wave-uniform base, so the load needs no
// 64-bit vector arithmetic (pt_set_mesh keeps the table under 4 GiB).
__device__ __forceinline__ const float4 *pt_walk_record(const PtDeviceScene &sc, uint32_t at)
{
    return reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc.nodes) + (size_t)(uint32_t)(at << 5));
}

// hitScene, compute.wgsl:214-302, split into "begin" (spheres in array order, traversal set-up)
// and "step" (one interior-node or one leaf visit) so that a wave can interleave rays that are
// at different points of their walk.  The walk itself is the shader's: near child first by
// hitAabb entry distance, prune distance lowered by triangle hits only, strict '<' everywhere.
struct Walk {
    Closest rec;
    v3 inv;          // 1 / direction
    float nearest;   // prune distance (compute.wgsl:242)
    uint32_t node;   // token of the node to visit next
    uint32_t sp;     // stack entries in use
};

// Returns true if there is a BVH to walk (w.node/w.sp valid), false if the hit is already final.
// The root's child pair, re-read from the kernel-argument segment at the point of use (the scene block is the first
// member of every kernel's parameter block): fourteen scalar registers that are live for one box test instead of
// for the whole shading pass.
typedef const __attribute__((address_space(4))) uint32_t *PtKernargWords;
static_assert(offsetof(PtTraceParams, sc) == 0 && offsetof(PtRayQuery, sc) == 0, "the scene block leads the kernel arguments");
__device__ __forceinline__ void pt_fresh_root_pair(float4 &q0, float4 &q1, float4 &q2, uint32_t &k1, uint32_t &k2)
{
    PtKernargWords kw = (PtKernargWords)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kw));
    constexpr uint32_t o = offsetof(PtDeviceScene, root_pair) / 4;
    q0 = make_float4(__uint_as_float(kw[o + 0]), __uint_as_float(kw[o + 1]), __uint_as_float(kw[o + 2]), __uint_as_float(kw[o + 3]));
    q1 = make_float4(__uint_as_float(kw[o + 4]), __uint_as_float(kw[o + 5]), __uint_as_float(kw[o + 6]), __uint_as_float(kw[o + 7]));
    q2 = make_float4(__uint_as_float(kw[o + 8]), __uint_as_float(kw[o + 9]), __uint_as_float(kw[o + 10]), __uint_as_float(kw[o + 11]));
    k1 = kw[o + 12];
    k2 = kw[o + 13];
}

// Is pixel (x, y) outside both screen rectangles of PtTraceParams::cull (camera rays only)?
__device__ __forceinline__ bool pt_camera_ray_misses_bvh(const uint32_t (&cull)[8], uint32_t x, uint32_t y)
{
    const bool out0 = (x - cull[0] > cull[1]) || (y - cull[2] > cull[3]);
    const bool out1 = (x - cull[4] > cull[5]) || (y - cull[6] > cull[7]);
    return out0 && out1;
}

// The sphere half of hitScene (compute.wgsl:218-236): every sphere in array order, closest accepted root.
template <bool STATS>
__device__ __forceinline__ Closest hit_spheres(const PtDeviceScene &sc, v3 o, v3 d, Counters &cn)
{
    Closest rec;
    rec.hit = false;
    rec.is_tri = false;
    rec.t = 0.0f;
    rec.index = 0;
    rec.u = rec.v = 0.0f;
    // hitSphere, compute.wgsl:304-350
    const float a = dot3(d, d);
    const float inv_den = pt_rcp(2.0f * a);
    const float a4 = 4.0f * a;
    for (uint32_t i = 0; i < sc.n_spheres; i++) {
        const float4 s0 = sc.spheres[3 * i];
        const v3 c = f4_xyz(s0);
        const float r = s0.w;
        if (STATS) cn.sphere_tests++;
        const v3 oc = o - c;
        const float b = 2.0f * dot3(d, oc);
        const float cc = pt_fma(-r, r, dot3(oc, oc));
        const float disc = pt_fma(b, b, -(a4 * cc));
        if (disc < 0.0f) continue;
        const float sq = pt_sqrt(disc);
        float root = (-b - sq) * inv_den;
        if (root <= sc.t_min || root >= sc.t_max) root = (-b + sq) * inv_den;
        if (root <= sc.t_min || root >= sc.t_max) continue;
        const float mat = sc.spheres[3 * i + 1].w;
        if (mat >= 3.0f && mat < 4.0f) {  // gizmo ring filter, compute.wgsl:342-347
            const v3 p = v3_fma(root, d, o);
            const v3 on = (p - c) * sc.spheres[3 * i + 2].x;
            const bool front = dot3(d, on) < 0.0f;
            const v3 n = front ? on : -on;
            const float dn = dot3(d, n);
            if (!(dn <= 0.2f && dn >= -0.2f)) continue;
        }
        if (!rec.hit || root < rec.t) {
            rec.hit = true;
            rec.t = root;
            rec.index = i;
            if (STATS) cn.sphere_accept++;
        }
    }
    return rec;
}

// May this ray walk the guard tree?  Its origin within reach of the mesh, scaled by the direction's length where that
// exceeds one (the padding of the guard tree's boxes is sized for such rays), and a finite direction with no component
// near zero (so that every slab test is ordinary arithmetic: no infinities, no 0 * inf).  A NaN anywhere fails a
// comparison and sends the ray to the reference tree.
__device__ __forceinline__ bool pt_guard_in_reach(const PtDeviceScene &sc, v3 o, v3 d)
{
    const v3 oc = o - V3(sc.guard_center[0], sc.guard_center[1], sc.guard_center[2]);
    // (the hardware's approximate square roots, a few ulps: the comparison below leaves a thousand times that)
    const float dist = __builtin_amdgcn_sqrtf(dot3(oc, oc)) + sc.guard_radius;
    const float len = pt_max(1.0f, __builtin_amdgcn_sqrtf(dot3(d, d)));
    const float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
    return dist * len * 1.0001f <= sc.guard_reach && pt_min(pt_min(ax, ay), az) >= 1e-24f && pt_max(pt_max(ax, ay), az) <= 1e24f;
}

// Pooled kernel: does the ray of a path slot whose flag word (F_FS) is `fs` walk the guard tree (of a scene that has
// one)?  Asked twice with the same answer: by the walker that picks the ray up, and by the pass over hits that shades what
// the walk found.  Bit 2 of the flag word: a ray sent back for a walk of the reference tree.
__device__ __forceinline__ bool pt_ray_walks_guard(const PtDeviceScene &sc, v3 o, v3 d, uint32_t fs)
{
    return (fs & 4u) == 0u && pt_guard_in_reach(sc, o, d);
}

// Is the closest hit `c` of a guard-tree walk (a triangle) one that the reference walk is certain to return as well?
// The reference walk reaches a triangle through the boxes of its leaf and that leaf's ancestors, all of which contain
// the triangle's own bounds.  With the hit point `margin` inside those bounds on every axis -- margin being many times
// what the slab test's three roundings can move a box face for a ray in reach -- every one of those boxes is entered
// before t and left after it, in float as in exact arithmetic: none of them is pruned (the prune distance never drops
// below t, the smallest accepted t there is), the triangle is tested, and it wins with the same t, u, v.
__device__ __forceinline__ bool pt_guard_hit_is_certain(const PtDeviceScene &sc, const Ray &ray, const Closest &c)
{
    const float4 *tp = pt_walk_record(sc, c.index & PT_ID_INDEX);
    const v3 a = f4_xyz(tp[0]), b = a + f4_xyz(tp[1]), cc = a + f4_xyz(tp[2]);
    const v3 p = v3_fma(c.t, ray.d, ray.o);
    const float m = sc.guard_margin;
    const bool in_x = p.x - pt_min(pt_min(a.x, b.x), cc.x) >= m && pt_max(pt_max(a.x, b.x), cc.x) - p.x >= m;
    const bool in_y = p.y - pt_min(pt_min(a.y, b.y), cc.y) >= m && pt_max(pt_max(a.y, b.y), cc.y) - p.y >= m;
    const bool in_z = p.z - pt_min(pt_min(a.z, b.z), cc.z) >= m && pt_max(pt_max(a.z, b.z), cc.z) - p.z >= m;
    return in_x && in_y && in_z && c.t < 9999.0f;
}

// skip_root: the caller knows that this ray enters neither of the root's child boxes (a camera ray of a pixel outside
// their screen rectangles): the root visit is counted but not computed.
template <bool STATS, typename StackT>
__device__ __forceinline__ bool walk_begin_body(const PtDeviceScene &sc, const Ray &ray, Walk &w, StackT stack, Counters &cn, bool skip_root)
{
    const v3 o = ray.o, d = ray.d;
    w.rec = hit_spheres<STATS>(sc, o, d, cn);
    w.nearest = 9999.0f;
    w.node = sc.root_token;
    w.sp = 0;
    if (sc.n_nodes == 0) return false;
    if (skip_root) {  // (only ever set when the root is an interior node)
        if (STATS) cn.interior++;
        return false;
    }
    w.inv = pt_rcp3(d);
    if (sc.root_token & PT_TOKEN_LEAF) return true;
    // First visit (the root's two children) from wave-uniform registers instead of memory: most
    // rays of an outdoor scene end their walk right here.
    if (STATS) cn.interior++;
    float d1, d2;
    float4 r0, r1, r2;
    uint32_t k1, k2;
    pt_fresh_root_pair(r0, r1, r2, k1, k2);
    hit_aabb_pair(o, w.inv, r0, r1, r2, d1, d2);
    {  // near child first (selects: cheaper than a lane mask around four moves)
        const bool far_first = d1 > d2;
        const float n = far_first ? d2 : d1, f = far_first ? d1 : d2;
        const uint32_t kn = far_first ? k2 : k1, kf = far_first ? k1 : k2;
        d1 = n; d2 = f; k1 = kn; k2 = kf;
    }
    if (d1 > w.nearest) return false;  // stack is empty: walk over (compute.wgsl:263-265)
    w.node = k1;
    if (d2 < w.nearest) stack.push(w.sp++, k2);
    return true;
}

template <bool STATS, typename StackT>
__device__ __forceinline__ bool walk_begin(const PtDeviceScene &sc, const Ray &ray, Walk &w, StackT stack, Counters &cn, bool skip_root = false)
{
    const bool need = walk_begin_body<STATS>(sc, ray, w, stack, cn, skip_root);
    // (added, not branched on: an increment in a branch of its own gets merged with the neighbouring branches' increments
    // of other counters into one store through a selected address, which keeps the counters out of registers)
    if (STATS) cn.walks += need ? 1u : 0u;
    return need;
}

// One node visit.  Returns true when the walk is over (stack empty, compute.wgsl:264, 292).
template <bool STATS, typename StackT>
__device__ __forceinline__ bool walk_step(const PtDeviceScene &sc, const Ray &ray, Walk &w, StackT stack, Counters &cn)
{
    const v3 o = ray.o, d = ray.d;
    const uint32_t cur = w.node;
    if (!(cur & PT_TOKEN_LEAF)) {
        if (STATS) cn.interior++;
        const float4 *pn = pt_walk_record(sc, cur);  // children cur, cur+1: 64 contiguous bytes
        const float4 q0 = pn[0], q1 = pn[1], q2 = pn[2], q3 = pn[3];
        float d1, d2;
        hit_aabb_pair(o, w.inv, q0, q1, q2, d1, d2);
        uint32_t k1 = __float_as_uint(q3.x), k2 = __float_as_uint(q3.y);
        {  // near child first (selects: cheaper than a lane mask around four moves)
            const bool far_first = d1 > d2;
            const float n = far_first ? d2 : d1, f = far_first ? d1 : d2;
            const uint32_t kn = far_first ? k2 : k1, kf = far_first ? k1 : k2;
            d1 = n; d2 = f; k1 = kn; k2 = kf;
        }
        if (d1 > w.nearest) {
            if (w.sp == 0) return true;
            w.node = stack.pop(--w.sp);
        } else {
            w.node = k1;
            if (d2 < w.nearest) stack.push(w.sp++, k2);
        }
        return false;
    }
    uint32_t count = (cur >> 28) & 7u, first = cur & PT_TOKEN_LOW;
    if (count == 0) {
        const uint2 m = sc.node_meta[first];
        first = m.x;
        count = m.y;
    }
    // hitTriangle, compute.wgsl:352-392 (edges precomputed: same subtraction)
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t slot = first + 2u * k;  // (table position of the triangle's record)
        const float4 *tp = pt_walk_record(sc, slot);
        const float4 A4 = tp[0], E1 = tp[1], E2 = tp[2];
        if (STATS) cn.tri_tests++;
        const v3 e1 = f4_xyz(E1), e2 = f4_xyz(E2);
        const v3 h = cross3(d, e2);
        const float a = dot3(e1, h);
        if (a > -0.00001f && a < 0.00001f) continue;
        const float f = pt_rcp(a);
        const v3 s = o - f4_xyz(A4);
        const float u = f * dot3(s, h);
        if (u < 0.0f || u > 1.0f) continue;
        const v3 q = cross3(s, e1);
        const float v = f * dot3(d, q);
        if (v < 0.0f || u + v > 1.0f) continue;
        const float t = f * dot3(e2, q);
        if (t <= sc.t_min || t >= sc.t_max) continue;
        if (t == w.nearest) w.rec.index |= PT_ID_TIE;  // (see walk_triangle)
        if (!w.rec.hit || t < w.rec.t) {
            w.rec.hit = true;
            w.rec.is_tri = true;
            w.rec.t = t;
            w.rec.index = slot;
            w.rec.u = u;
            w.rec.v = v;
            w.nearest = t;
            if (STATS) cn.tri_accept++;
        }
    }
    if (w.sp == 0) return true;
    w.node = stack.pop(--w.sp);
    return false;
}

// The same walk in steps that each wait for memory only once: a step is one interior-node visit
// or ONE triangle of a leaf, and every lane's loads (child pair or triangle) are issued together
// before any of them is used, so a wave whose lanes are at interior nodes and at leaves pays one
// memory latency per step instead of one per kind.  Node and triangle order are unchanged.
struct WalkU : Walk {
    uint32_t leaf_left;  // triangles of the leaf being tested still to test; then w.node is the table position of the next
                         // one's record (0: w.node is a token still to be visited)
    uint32_t top;        // walks that keep the newest stack entry in a register (walk_consume): that entry, valid while
                         // sp != 0; entries 0 .. sp-2 are in the stack's memory
};

// Call whenever w.node has been set: a leaf with an inline triangle count goes straight to
// triangle mode.
__device__ __forceinline__ void walk_enter(WalkU &w)
{
    const uint32_t count = (w.node >> 28) & 7u;
    if ((w.node & PT_TOKEN_LEAF) && count != 0u) {
        w.node &= PT_TOKEN_LOW;
        w.leaf_left = count;
    }
}

// One visit of a four-child record of the guard tree (PtDeviceScene::guard_quads) loaded into q0..q3: the four slab
// tests on bounds dequantised on the fly -- t = byte * (scale / d) + (origin - o) / d per plane, which is the slab test of
// compute.wgsl:441-456 on the box origin + byte * scale up to a rounding that the guard tree's padding covers --, the
// children sorted by entry distance, the nearest entered and the others pushed farthest first.  Which order the children
// are visited in is free here: the guard walk only has to reach every box that a hit may lie in.
// The walk has room for `stack_cap` entries, whatever the tree's depth: a visit that could overflow it gives the walk up
// (PT_ID_ABANDONED on the record; the caller sends the ray through the reference tree, whose depth the stack IS sized
// for) -- a visit pushes at most three entries, and walks that come anywhere near a full stack do not occur in practice.
template <typename StackT>
__device__ __forceinline__ void walk_quad_visit(v3 o, v3 inv, float4 q0, float4 q1, float4 q2, float4 q3, WalkU &w, StackT stack, bool &next_node, uint32_t stack_cap)
{
    const float ax = q0.w * inv.x, ay = q3.z * inv.y, az = q3.w * inv.z;
    const float bx = (q0.x - o.x) * inv.x, by = (q0.y - o.y) * inv.y, bz = (q0.z - o.z) * inv.z;
    const uint32_t lox = __float_as_uint(q1.x), loy = __float_as_uint(q1.y), loz = __float_as_uint(q1.z);
    const uint32_t hix = __float_as_uint(q1.w), hiy = __float_as_uint(q2.x), hiz = __float_as_uint(q2.y);
    uint32_t k[4] = {__float_as_uint(q2.z), __float_as_uint(q2.w), __float_as_uint(q3.x), __float_as_uint(q3.y)};
    float dist[4];
    // (two children at a time: the twelve multiply-adds of a pair of children are six two-wide packed instructions)
    const pt_f2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az}, bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
#pragma unroll
    for (uint32_t c = 0; c < 4u; c += 2u) {
        auto two = [&](uint32_t word) { return pt_f2{(float)((word >> (8u * c)) & 0xFFu), (float)((word >> (8u * c + 8u)) & 0xFFu)}; };
        const pt_f2 tlx = __builtin_elementwise_fma(two(lox), ax2, bx2), thx = __builtin_elementwise_fma(two(hix), ax2, bx2);
        const pt_f2 tly = __builtin_elementwise_fma(two(loy), ay2, by2), thy = __builtin_elementwise_fma(two(hiy), ay2, by2);
        const pt_f2 tlz = __builtin_elementwise_fma(two(loz), az2, bz2), thz = __builtin_elementwise_fma(two(hiz), az2, bz2);
        const float tmin0 = pt_max(pt_max(pt_min(tlx.x, thx.x), pt_min(tly.x, thy.x)), pt_min(tlz.x, thz.x));
        const float tmax0 = pt_min(pt_min(pt_max(tlx.x, thx.x), pt_max(tly.x, thy.x)), pt_max(tlz.x, thz.x));
        const float tmin1 = pt_max(pt_max(pt_min(tlx.y, thx.y), pt_min(tly.y, thy.y)), pt_min(tlz.y, thz.y));
        const float tmax1 = pt_min(pt_min(pt_max(tlx.y, thx.y), pt_max(tly.y, thy.y)), pt_max(tlz.y, thz.y));
        dist[c] = (tmin0 > tmax0 || tmax0 < 0.0f || k[c] == 0xFFFFFFFFu) ? 99999.0f : tmin0;
        dist[c + 1u] = (tmin1 > tmax1 || tmax1 < 0.0f || k[c + 1u] == 0xFFFFFFFFu) ? 99999.0f : tmin1;
    }
    auto order = [&](uint32_t i, uint32_t j) {  // (dist[i], k[i]) <= (dist[j], k[j]) afterwards
        const bool swap = dist[i] > dist[j];
        const float dn = swap ? dist[j] : dist[i], df = swap ? dist[i] : dist[j];
        const uint32_t kn = swap ? k[j] : k[i], kf = swap ? k[i] : k[j];
        dist[i] = dn; dist[j] = df; k[i] = kn; k[j] = kf;
    };
    order(0, 1); order(2, 3); order(0, 2); order(1, 3); order(1, 2);
    if (dist[0] > w.nearest) {
        next_node = true;
    } else if (w.sp + 3u > stack_cap) {
        w.rec.index |= PT_ID_ABANDONED;
        w.sp = 0;
        next_node = true;  // (an empty stack: the walk ends here)
    } else {
        w.node = k[0];
        if (dist[3] < w.nearest) stack.push(w.sp++, k[3]);
        if (dist[2] < w.nearest) stack.push(w.sp++, k[2]);
        if (dist[1] < w.nearest) stack.push(w.sp++, k[1]);
    }
}

// stack_cap: entries the stack has room for (only a walk of the guard tree's four-child form looks at it).
template <bool STATS, typename StackT>
__device__ __forceinline__ bool walk_step_u(const PtDeviceScene &sc, const Ray &ray, WalkU &w, StackT stack, Counters &cn, uint32_t stack_cap)
{
    const v3 o = ray.o, d = ray.d;
    const bool tri = w.leaf_left != 0u;
    bool next_node = false;
    if (!tri && (w.node & PT_TOKEN_LEAF)) {
        // big leaf: (first, count) from the side table (rare: the builder splits down to two triangles)
        const uint2 m = sc.node_meta[w.node & PT_TOKEN_LOW];
        w.node = m.x;
        w.leaf_left = m.y;
        next_node = m.y == 0u;
    } else {
        const float4 *base = pt_walk_record(sc, w.node);
        // all four loads for every lane (node pairs and triangles are both 64-byte records): issued together
        float4 q0 = base[0], q1 = base[1], q2 = base[2], q3 = base[3];
        // (the empty asm uses the values here, so the compiler cannot sink part of the loads into the
        // interior-node branch, where they would cost that branch a second wait)
        asm volatile("" : : "v"(q0.w), "v"(q1.w), "v"(q2.w), "v"(q3.x), "v"(q3.y));
        if (!tri && !STATS && sc.guard_quads != 0u && w.node >= sc.guard_quads) {
            walk_quad_visit(o, w.inv, q0, q1, q2, q3, w, stack, next_node, stack_cap);
        } else if (!tri) {
            if (STATS) cn.interior++;
            float d1, d2;
            hit_aabb_pair(o, w.inv, q0, q1, q2, d1, d2);
            uint32_t k1 = __float_as_uint(q3.x), k2 = __float_as_uint(q3.y);
            {  // near child first (selects: cheaper than a lane mask around four moves)
                const bool far_first = d1 > d2;
                const float n = far_first ? d2 : d1, f = far_first ? d1 : d2;
                const uint32_t kn = far_first ? k2 : k1, kf = far_first ? k1 : k2;
                d1 = n; d2 = f; k1 = kn; k2 = kf;
            }
            if (d1 > w.nearest) {
                next_node = true;
            } else {
                w.node = k1;
                if (d2 < w.nearest) stack.push(w.sp++, k2);
            }
        } else {
            // hitTriangle, compute.wgsl:352-392 (edges precomputed: same subtraction)
            const uint32_t slot = w.node;
            w.node = slot + 2u;
            w.leaf_left--;
            next_node = w.leaf_left == 0u;
            if (STATS) cn.tri_tests++;
            // Straight-line form of the shader's early-outs: the same values, all computed, then one
            // test (a lane that would have left early only idles in a wave anyway).
            const v3 e1 = f4_xyz(q1), e2 = f4_xyz(q2);
            const v3 h = cross3(d, e2);
            const float a = dot3(e1, h);
            const float f = pt_rcp(a);
            const v3 s = o - f4_xyz(q0);
            const float u = f * dot3(s, h);
            const v3 q = cross3(s, e1);
            const float v = f * dot3(d, q);
            const float t = f * dot3(e2, q);
            const bool in_range = !(a > -0.00001f && a < 0.00001f) && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) && !(t <= sc.t_min || t >= sc.t_max);
            const bool ok = in_range && (!w.rec.hit || t < w.rec.t);
            if (in_range && t == w.nearest) w.rec.index |= PT_ID_TIE;  // (see walk_triangle)
            if (ok) {
                w.rec.hit = true;
                w.rec.is_tri = true;
                w.rec.t = t;
                w.rec.index = slot;
                w.rec.u = u;
                w.rec.v = v;
                w.nearest = t;
                if (STATS) cn.tri_accept++;
            }
        }
    }
    if (next_node) {
        if (w.sp == 0) return true;
        w.node = stack.pop(--w.sp);
    }
    if (!tri || next_node) walk_enter(w);
    return false;
}

// The same step in two halves, for a loop that keeps the NEXT record's loads in flight while it finishes the
// current step: walk_fetch issues the four loads of the record the walk stands at, walk_consume takes a loaded record,
// decides where the walk goes next (an interior lane from its box tests, a lane at a triangle without looking at the
// triangle: the next triangle of the leaf, or the stack), issues the loads of THAT record, and only then runs the
// triangle test, whose result (closest hit, prune distance) no address depends on.
struct WalkQ {
    float4 q0, q1, q2, q3;
};

__device__ __forceinline__ void walk_fetch(const PtDeviceScene &sc, const WalkU &w, WalkQ &q)
{
    // (a big leaf's token, flag | node index, is resolved through node_meta and has nothing to fetch here: its flag bit
    // leaves the 32-bit byte offset, so the loads read the node's own place in the table and are ignored)
    const float4 *base = pt_walk_record(sc, w.node);
    q.q0 = base[0]; q.q1 = base[1]; q.q2 = base[2]; q.q3 = base[3];
}

// Second half, for a lane that stood at triangle `slot` with its record in q: hitTriangle, compute.wgsl:352-392 (edges
// precomputed: same subtraction), straight-line as in walk_step_u.
// A hit that the test accepts at exactly the t already held marks the held record (PT_ID_TIE): which of the two a walk
// ends up with is the order of its tests, i.e. the tree's, and a guard-tree walk gives such a ray to the reference tree.
// (Compared with the prune distance: on the guard tree that is the held t whatever is held, on the reference tree the
// held triangle's.  A closer hit accepted later replaces the record and the mark with it.)
template <bool STATS>
__device__ __forceinline__ void walk_triangle(const PtDeviceScene &sc, v3 o, v3 d, WalkU &w, float4 q0, float4 q1, float4 q2, uint32_t slot, Counters &cn)
{
    if (STATS) cn.tri_tests++;
    const v3 e1 = f4_xyz(q1), e2 = f4_xyz(q2);
    const v3 h = cross3(d, e2);
    const float a = dot3(e1, h);
    const float f = pt_rcp(a);
    const v3 sv = o - f4_xyz(q0);
    const float u = f * dot3(sv, h);
    const v3 qq = cross3(sv, e1);
    const float v = f * dot3(d, qq);
    const float t = f * dot3(e2, qq);
    const bool in_range = !(a > -0.00001f && a < 0.00001f) && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) && !(t <= sc.t_min || t >= sc.t_max);
    const bool ok = in_range && (!w.rec.hit || t < w.rec.t);
    if (in_range && t == w.nearest) w.rec.index |= PT_ID_TIE;
    if (ok) {
        w.rec.hit = true;
        w.rec.is_tri = true;
        w.rec.t = t;
        w.rec.index = slot;
        w.rec.u = u;
        w.rec.v = v;
        w.nearest = t;
        if (STATS) cn.tri_accept++;
    }
}

// A whole step on the loaded record q (the record w.node points at), in the order that keeps the walk's chain of
// dependent fetches short: (1) where the walk goes next -- an interior lane from its two box tests, a lane at a triangle
// from the leaf's count, either of them possibly from the stack -- as one token; (2) the next record's loads, straight
// from that token (`prefetch` is wave-uniform); (3) everything no address depends on: the stack's bookkeeping, entering
// the node, the triangle test.  True when the walk is over.
// The newest stack entry lives in w.top.  A push stores the entry it displaces, a pop takes w.top and loads the entry
// that becomes the newest -- a value that nothing needs before the walk's next push or pop, a whole record fetch later:
// the read goes into `below` and is moved into w.top at the very end of the step, so no step waits for its own stack
// read.  A push onto an empty stack and a pop of the only entry (most of them: walks are shallow) touch no memory.
template <bool STATS, typename StackT>
__device__ __forceinline__ bool walk_consume(const PtDeviceScene &sc, v3 o, v3 d, WalkU &w, WalkQ q, StackT stack, Counters &cn, WalkQ &qn, bool prefetch)
{
    const bool tri = w.leaf_left != 0u;
    const uint32_t slot = w.node;
    // (the empty asm uses the loaded values here, so that the compiler waits for them once, before the branches)
    asm volatile("" : : "v"(q.q0.w), "v"(q.q1.w), "v"(q.q2.w), "v"(q.q3.x), "v"(q.q3.y));
    // (1)
    uint32_t next = slot + 2u;         // a lane at a triangle: the leaf's next triangle ...
    bool leave = w.leaf_left == 1u;    // ... unless this was the last one: the next node comes off the stack
    bool push = false;
    uint32_t far_child = 0;
    if (!tri) {
        if (w.node & PT_TOKEN_LEAF) {
            // big leaf: (first, count) from the side table (rare: the builder splits down to two triangles)
            const uint2 m = sc.node_meta[w.node & PT_TOKEN_LOW];
            next = m.x;
            w.leaf_left = m.y;
            leave = m.y == 0u;
        } else {
            if (STATS) cn.interior++;
            float d1, d2;
            hit_aabb_pair(o, w.inv, q.q0, q.q1, q.q2, d1, d2);
            const uint32_t kl = __float_as_uint(q.q3.x), kr = __float_as_uint(q.q3.y);
            // near child first (selects: cheaper than a lane mask around four moves)
            const bool far_first = d1 > d2;
            const float dn = far_first ? d2 : d1, df = far_first ? d1 : d2;
            next = far_first ? kr : kl;
            far_child = far_first ? kl : kr;
            leave = dn > w.nearest;
            push = !leave && df < w.nearest;
        }
    }
    const bool over = leave && w.sp == 0u, pops = leave && w.sp != 0u;
    if (pops) next = w.top;
    // (2) (a token's flag and count bits leave the 32-bit byte offset: a leaf's token addresses its first triangle, and
    // a big leaf's, flag | node index, the node's own place in the table -- loads that are ignored)
    if (prefetch && !over) {
        const float4 *base = pt_walk_record(sc, next);
        qn.q0 = base[0]; qn.q1 = base[1]; qn.q2 = base[2]; qn.q3 = base[3];
    }
    // (3)
    if (push) {
        if (w.sp != 0u) stack.push(w.sp - 1u, w.top);
        w.top = far_child;
        w.sp++;
    }
    uint32_t below = 0;
    bool below_is_top = false;
    if (pops) {
        w.sp--;
        // (of an empty stack: some value, never used; an entry beyond the LDS part is rare and goes to w.top at once)
        below_is_top = stack.peek_newest(w.sp, below, w.top);
    }
    if (tri) w.leaf_left--;
    w.node = next;
    if (!over && (!tri || leave)) walk_enter(w);
    if (tri) walk_triangle<STATS>(sc, o, d, w, q.q0, q.q1, q.q2, slot, cn);
    // (the empty asm pins the stack read's first use to this place, or the compiler folds the move back into the pop's
    // branch and waits there)
    asm volatile("" : "+v"(below));
    if (below_is_top) w.top = below;
    return over;
}

// walk_begin for the pooled kernel: same spheres + root visit, but the far child (if it has to be
// visited) is returned in k2 instead of being pushed, because the walk may run on another lane.
template <bool STATS>
__device__ __forceinline__ bool walk_begin_pool(const PtDeviceScene &sc, const Ray &ray, Walk &w, uint32_t &k2_out, Counters &cn, bool skip_root = false)
{
    struct OneSlot {
        uint32_t v;
        bool used;
        __device__ __forceinline__ void push(uint32_t, uint32_t x) { v = x; used = true; }
        __device__ __forceinline__ uint32_t pop(uint32_t) { return v; }
    };
    struct Ref {
        OneSlot *p;
        __device__ __forceinline__ void push(uint32_t sp, uint32_t x) { p->push(sp, x); }
        __device__ __forceinline__ uint32_t pop(uint32_t sp) { return p->pop(sp); }
    };
    OneSlot slot{0xFFFFFFFFu, false};
    const bool need = walk_begin<STATS>(sc, ray, w, Ref{&slot}, cn, skip_root);
    k2_out = slot.used ? slot.v : 0xFFFFFFFFu;
    // a root that is itself a leaf still needs its triangles tested
    return need;
}

template <bool STATS, typename StackT>
__device__ __forceinline__ Closest hit_scene(const PtDeviceScene &sc, const Ray &ray, StackT stack, Counters &cn)
{
    Walk w;
    if (walk_begin<STATS>(sc, ray, w, stack, cn))
        while (!walk_step<STATS>(sc, ray, w, stack, cn)) {}
    return w.rec;
}

// Expands the closest hit into the shader's HitRecord (compute.wgsl:334-340, 394-403).
__device__ __forceinline__ HitRecord make_record(const PtDeviceScene &sc, const Ray &ray, const Closest &c)
{
    HitRecord h;
    h.hit = c.hit;
    h.t = c.t;
    h.prim = 0xFFFFFFFFu;
    h.p = h.normal = h.att = V3(0.0f, 0.0f, 0.0f);
    h.front = false;
    h.material = 0.0f;
    if (!c.hit) return h;
    h.p = v3_fma(c.t, ray.d, ray.o);
    v3 on;
    const uint32_t index = c.index & PT_ID_INDEX;  // (without the walk's PT_ID_TIE mark)
    if (c.is_tri) {
        const float4 *tp = pt_walk_record(sc, index);  // table position of the triangle's record
        const float4 *tn = sc.tri_nrm + 3 * (size_t)((index - sc.tri_base) >> 1);
        const v3 an = f4_xyz(tn[0]), bn = f4_xyz(tn[1]), cnn = f4_xyz(tn[2]);
        const float w = (1.0f - c.u) - c.v;
        const v3 n = V3(pt_fma(cnn.x, c.v, pt_fma(bn.x, c.u, an.x * w)), pt_fma(cnn.y, c.v, pt_fma(bn.y, c.u, an.y * w)),
                        pt_fma(cnn.z, c.v, pt_fma(bn.z, c.u, an.z * w)));
        on = normalize3(n);
        h.att = V3(1.0f, 1.0f, 1.0f);  // triangle albedo/material are ignored: compute.wgsl:359-367
        h.material = 0.0f;
        if (sc.tri_materials) {  // opt-in: the record's own albedo and material (model.rs:66-67)
            h.att = V3(tn[0].w, tn[1].w, tn[2].w);
            h.material = tp[1].w;
        }
        h.prim = __float_as_uint(tp[0].w);
    } else {
        const float4 s0 = sc.spheres[3 * index], s1 = sc.spheres[3 * index + 1];
        on = (h.p - f4_xyz(s0)) * sc.spheres[3 * index + 2].x;
        h.att = f4_xyz(s1);
        h.material = s1.w;
        h.prim = 0x80000000u | index;
    }
    h.front = dot3(ray.d, on) < 0.0f;
    h.normal = h.front ? on : -on;
    return h;
}

// getBackgroundColor, compute.wgsl:209-212: cube face by major axis (ties z, y, x), nearest
// texel, clamp to edge.
__device__ __forceinline__ v3 sky_lookup(const PtDeviceScene &sc, v3 d)
{
    if (sc.sky_size == 0) return V3(0.0f, 0.0f, 0.0f);
    const float ax = pt_abs(d.x), ay = pt_abs(d.y), az = pt_abs(d.z);
    // (selects, not branches: the three cases are a few moves each, and a lane mask per case costs more)
    const bool zmaj = az >= ax && az >= ay, ymaj = !zmaj && ay >= ax;
    const float ma = zmaj ? az : ymaj ? ay : ax;
    const bool neg = (zmaj ? d.z : ymaj ? d.y : d.x) < 0.0f;
    const uint32_t face = (zmaj ? 4u : ymaj ? 2u : 0u) + (neg ? 1u : 0u);
    const float scoord = zmaj ? (neg ? -d.x : d.x) : ymaj ? d.x : (neg ? d.z : -d.z);
    const float tcoord = ymaj ? (neg ? -d.z : d.z) : -d.y;
    const float size = (float)sc.sky_size;
    const float inv_ma = pt_rcp(ma);
    const float fs = pt_fma(0.5f, scoord * inv_ma, 0.5f) * size;
    const float ft = pt_fma(0.5f, tcoord * inv_ma, 0.5f) * size;
    const uint32_t hi = sc.sky_size - 1u;
    const uint32_t i = (fs >= 0.0f) ? (fs < size ? (uint32_t)fs : hi) : 0u;
    const uint32_t j = (ft >= 0.0f) ? (ft < size ? (uint32_t)ft : hi) : 0u;
    const float4 tx = sc.sky[(face * sc.sky_size + j) * sc.sky_size + i];  // at most 6 * 16384^2 texels: 32-bit arithmetic
    return V3(tx.x, tx.y, tx.z);
}

// scatter / reflect / reflectance / refract: compute.wgsl:408-439
__device__ __forceinline__ v3 scatter_dir(v3 dir, v3 normal, float seed)
{
    v3 s = normal + pt_random_unit(dir.x * seed, dir.y * seed);
    if (pt_abs(s.x) < 1e-8f && pt_abs(s.y) < 1e-8f && pt_abs(s.z) < 1e-8f) s = normal;
    return s;
}
__device__ __forceinline__ v3 reflect_dir(v3 dir, v3 normal)
{
    const float k = 2.0f * dot3(dir, normal);
    return normalize3(v3_fma(-k, normal, dir));  // the fuzz term has weight 0 and is dropped
}
constexpr float kRefrFront = 1.0f / 1.5f, kRefrBack = 1.5f;
constexpr float r0_squared(float ref_idx)  // folded by the compiler with the same IEEE rounding
{
    return ((1.0f - ref_idx) / (1.0f + ref_idx)) * ((1.0f - ref_idx) / (1.0f + ref_idx));
}
__device__ __forceinline__ float reflectance(float cosine, bool front)
{
    const float r0s = front ? r0_squared(kRefrFront) : r0_squared(kRefrBack);
    const float x = 1.0f - cosine, x2 = x * x;
    const float p5 = (x2 * x2) * x;
    return pt_fma(1.0f - r0s, p5, r0s);
}
__device__ __forceinline__ v3 refract_dir(v3 dir, v3 normal, float eta)
{
    const float cos_t = pt_min(dot3(-dir, normal), 1.0f);
    const v3 perp = v3_fma(cos_t, normal, dir) * eta;
    const float par = -pt_sqrt(pt_abs(1.0f - dot3(perp, perp)));
    return v3_fma(par, normal, perp);
}
__device__ __forceinline__ uint32_t material_u32(float m)  // WGSL u32(f32): truncate, saturate, NaN -> 0
{
    if (!(m >= 0.0f)) return 0u;
    if (m >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)m;
}

// One iteration of rayColor's loop body after hitScene (compute.wgsl:140-203): either the path ends
// (returns true, `result` = its colour) or `cur` becomes the bounce ray and `color` the new throughput.
template <bool STATS, bool GIZMO>
__device__ __forceinline__ bool shade_ray(const PtDeviceScene &sc, Ray &cur, const Closest &c, v3 &color, float seed, uint32_t &bounce,
                                          uint32_t &limit, v3 initial_d, v3 &result, uint32_t &prim, Counters &cn)
{
    const HitRecord h = make_record(sc, cur, c);
    prim = h.prim;
    if (!h.hit) {
        // compute.wgsl:140-143
        if (STATS) cn.sky++;
        result = color * sky_lookup(sc, cur.d);
        return true;
    }
    if (STATS) cn.bounces++;
    bool path_done = false;
    const v3 dir = normalize3(cur.d);
    v3 bdir = dir;
    const uint32_t m = material_u32(h.material);
    if (m == 1u) {
        bdir = reflect_dir(dir, h.normal);
    } else if (m == 2u) {
        const float ri = h.front ? kRefrFront : kRefrBack;
        const float cos_t = pt_min(dot3(-dir, h.normal), 1.0f);
        const float sin_t = pt_sqrt(pt_fma(-cos_t, cos_t, 1.0f));
        const bool cannot = ri * sin_t > 1.0f;
        if (cannot || reflectance(cos_t, h.front) > pt_hash_rand(h.p.x, h.p.y)) bdir = reflect_dir(dir, h.normal);
        else bdir = refract_dir(dir, h.normal, ri);
    } else if (GIZMO && m == 3u) {
        const float dn = dot3(initial_d, h.normal);
        if (bounce == 0 && dn <= 0.2f && dn >= -0.2f) {
            result = h.att;
            path_done = true;
        }
        limit++;  // correction, compute.wgsl:193
    } else {
        bdir = scatter_dir(dir, h.normal, seed);
    }
    if (m <= 1u && dot3(bdir, h.normal) <= 0.0f) {
        // compute.wgsl:151-153, 161-163: absorbed ray still sees the sky along the incoming ray
        if (STATS) cn.sky++;
        result = (color * h.att) * sky_lookup(sc, cur.d);
        path_done = true;
    }
    // The bounce ray.  Written unconditionally: when the path ends here nobody reads cur, color or
    // bounce again before the next path overwrites them.
    if (m != 3u) color = color * h.att;
    cur.o = h.p;
    cur.d = bdir;
    bounce++;
    return path_done;
}

#ifdef PT_DIAG
// Diagnostic build only (make diag): per-wave start/end stamps, hardware id and loop trip count.
#define PT_DIAG_MAX_WAVES (1u << 18)
__device__ unsigned long long g_diag[8 * PT_DIAG_MAX_WAVES];
__device__ unsigned long long g_diag_marks[5 * 8 * 64];  // first 64 work items: realtime at every 1024th loop trip
#endif

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// A wave's counts onto its stripe of the launch's counters: one atomic per wave and counter (word k of the stripe =
// counter k of PT_NUM_COUNTERS; word 1, the paths, is counted on the host).
template <bool STATS>
__device__ __forceinline__ void pt_add_counters(unsigned long long *stripe_words, const Counters &cn, uint32_t lane)
{
    auto add = [&](uint32_t k, uint32_t v) {
        const uint32_t s = wave_sum(v);
        if (lane == 0 && s) atomicAdd(&stripe_words[k], (unsigned long long)s);
    };
    add(0, cn.rays);
    add(10, cn.guarded);
    add(11, cn.rewalks);
    if (STATS) {
        add(2, cn.interior); add(3, cn.tri_tests); add(4, cn.tri_accept); add(5, cn.sphere_tests);
        add(6, cn.sky); add(7, cn.bounces); add(8, cn.sphere_accept); add(9, cn.walks);
    }
}

// The kernel's parameter block is large (scene pointers, camera frame, the root's child boxes).  Kept
// in scalar registers for the whole persistent loop it overflows them, and every spilled value costs
// vector instructions in the walk loop.  The shading pass and the job hand-out therefore re-read the
// block from the kernel-argument segment (scalar loads, live only while they run); the asm statement
// keeps the compiler from hoisting those loads back out of the loop.
static_assert(sizeof(PtTraceParams) % 4 == 0, "parameter block is read word by word");
// A pointer rebuilt from plain words is a generic-address pointer, dereferenced through the slower flat path.  Every
// pointer of the block points to global memory: saying so (a cast to the global address space and back, which the
// compiler's address-space inference follows through the inlined callees) turns those accesses into global ones,
// so that no pointer has to be kept in scalar registers across the persistent loop just to keep its address space.
template <typename T>
__device__ __forceinline__ void pt_mark_global(T *&p)
{
    typedef __attribute__((address_space(1))) T *G;
    p = (T *)(G)(uintptr_t)p;
}
// Read-only for the whole launch and indexed wave-uniformly (the sphere records): constant address space, so that
// the loads are scalar loads again and do not make the pass wait for its outstanding vector loads.
template <typename T>
__device__ __forceinline__ void pt_mark_constant(T *&p)
{
    typedef __attribute__((address_space(4))) T *G;
    p = (T *)(G)(uintptr_t)p;
}
__device__ __forceinline__ void pt_fresh_params(PtTraceParams &L)
{
    PtKernargWords kw = (PtKernargWords)__builtin_amdgcn_kernarg_segment_ptr();  // the block is the kernel's only argument
    asm volatile("" : "+s"(kw));
    uint32_t *lw = reinterpret_cast<uint32_t *>(&L);
#pragma unroll
    for (uint32_t i = 0; i < sizeof(PtTraceParams) / 4; i++) lw[i] = kw[i];  // unused words are never loaded
    pt_mark_constant(L.sc.spheres); pt_mark_global(L.sc.nodes); pt_mark_global(L.sc.node_meta);
    pt_mark_global(L.sc.tri_nrm); pt_mark_global(L.sc.sky);
    pt_mark_global(L.items); pt_mark_global(L.partial); pt_mark_global(L.cost); pt_mark_global(L.accum);
    pt_mark_global(L.aov_prim); pt_mark_global(L.aov_t); pt_mark_global(L.counters); pt_mark_global(L.gstack);
    pt_mark_global(L.next_item);
}
#define PT_FRESH_PARAMS(L) \
    PtTraceParams L;       \
    pt_fresh_params(L)

// ---- the megakernel ---------------------------------------------------------------------
// GIZMO: some sphere has material 3 (only then is the path's first direction kept);
// SPP1: one sample per pixel and dispatch (the per-frame sample sum is the sample itself);
// DIRECT: no item table -- work item i is tile i with every frame, added straight onto the sums.
template <bool STATS, bool OVERFLOW, bool GIZMO, bool SPP1, bool DIRECT>
// (the variant that keeps nine counters per lane gets the registers of four waves per SIMD instead of five)
__global__ __launch_bounds__(PT_TRACE_BLOCK, STATS ? (PT_TRACE_WAVES_PER_EU < 4 ? PT_TRACE_WAVES_PER_EU : 4) : PT_TRACE_WAVES_PER_EU) void trace_kernel(const PtTraceParams P)
{
    // one wave per workgroup = one 8x8 pixel tile; lane -> (l&7, l>>3).  Single-wave groups
    // retire independently, so a slow wave never pins the LDS and wave slots of finished ones.
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid;
    // Work item = some consecutive frames of one tile.  Direct mode (no item table): item i is tile i
    // with every frame of the launch, added straight onto the stored sums.  Planned mode: the
    // table written by plan_kernel (costliest tiles first, split into more items the costlier they
    // are); each item leaves a partial sum that accumulate_kernel adds in item order.
    uint32_t tile, frame_begin, frame_end;
    if (!DIRECT) {
        const uint4 it = P.items[blockIdx.x];
        if (it.z == 0) return;  // unused slot of the over-sized grid
        tile = it.x;
        frame_begin = it.y;
        frame_end = it.y + it.z;
    } else {
        tile = blockIdx.x;
        frame_begin = 0;
        frame_end = P.n_frames;
    }
    uint32_t cost_trips = 0, cost_refills = 0;  // wave-uniform work counts: the next plan's cost estimate
    const uint32_t tby = tile / P.tiles_x, tbx = tile - tby * P.tiles_x;
    const uint32_t x = tbx * PT_TILE + (lane & 7u);
    const uint32_t y = tby * PT_TILE + (lane >> 3);

    // running sums of this item: touched once per frame, so they live in LDS, not in registers
    __shared__ float4 lds_acc[PT_TRACE_BLOCK];
    __shared__ uint32_t smem[PT_LDS_STACK * PT_TRACE_BLOCK];
    Stack<OVERFLOW, PT_TRACE_BLOCK> stack{&smem[tid], nullptr, 0};
    if (OVERFLOW) {
        stack.spill = P.gstack + ((size_t)blockIdx.x * PT_TRACE_BLOCK + tid);
        stack.pitch = (size_t)gridDim.x * PT_TRACE_BLOCK;
    }

#ifdef PT_DIAG
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t diag_iters = 0;
    unsigned long long diag_a = 0, diag_b = 0, diag_c = 0, diag_s = 0;
    uint32_t diag_steps = 0;
#endif
    Counters cn;
    const bool in_image = (x < P.cols) && (y < P.rows) && frame_begin < frame_end && P.spp > 0;
    bool active = in_image;

    const PtDeviceScene &sc = P.sc;
    const size_t pix = (size_t)y * P.width + x;
    {
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (active && DIRECT) acc = P.accum[pix];
        lds_acc[lane] = acc;
    }

    uint32_t frame = frame_begin, sample = 0;
    uint32_t sx = 0, sy = 0, sz = 0, sw = 0;  // HybridTaus state
    v3 frame_sum = V3(0.0f, 0.0f, 0.0f);
    // per-path state
    Ray cur;
    cur.o = cur.d = V3(0.0f, 0.0f, 0.0f);
    v3 initial_d = V3(0.0f, 0.0f, 0.0f);
    v3 color = V3(1.0f, 1.0f, 1.0f);
    float seed = 0.0f;
    uint32_t bounce = 0, limit = 0;  // limit = depth + correction
    bool need_new = true;
    bool want_aov = false;

    // Retire finished paths and start new ones until this lane holds a ray to trace (or is done).
    // path_done/result describe the path that just ended (ignored when need_new is set).
    auto advance = [&](const PtTraceParams &Q, bool path_done, v3 result) {
        while (active && (need_new || path_done || bounce >= limit)) {
            if (!need_new) {
                // rayColor returned: compute.wgsl:121
                const v3 path_color = path_done ? result : color;
                path_done = false;
                if (!SPP1) {
                    frame_sum = frame_sum + path_color;
                    sample++;
                }
                if (SPP1 || sample == Q.spp) {
                    // compute.wgsl:124-127, accumulated instead of stored as rgba8unorm
                    const v3 mean = (SPP1 ? path_color : frame_sum) * Q.inv_spp;
                    float4 acc = lds_acc[lane];
                    acc.x += mean.x; acc.y += mean.y; acc.z += mean.z; acc.w += 1.0f;
                    lds_acc[lane] = acc;
                    frame_sum = V3(0.0f, 0.0f, 0.0f);
                    sample = 0;
                    frame++;
                    if (frame == frame_end) { active = false; break; }
                }
            }
            if (sample == 0) {
                // compute.wgsl:85
                const uint32_t time = Q.time0 + frame * Q.stride;
                sx = x; sy = y; sz = x + time; sw = y + time;
            }
            // compute.wgsl:113-119, 131-136
            const float px = -0.5f + pt_hybrid_taus(sx, sy, sz, sw);
            const float py = -0.5f + pt_hybrid_taus(sx, sy, sz, sw);
            const v3 smp = v3_fma(py, Q.fr.dv, Q.fr.du * px);
            const v3 ploc = v3_fma((float)y, Q.fr.dv, v3_fma((float)x, Q.fr.du, Q.fr.pixel00));  // compute.wgsl:109
            cur.o = Q.fr.origin;
            cur.d = (ploc + smp) - Q.fr.origin;
            if (GIZMO) initial_d = cur.d;
            color = V3(1.0f, 1.0f, 1.0f);
            seed = pt_hybrid_taus(sx, sy, sz, sw);
            bounce = 0;
            limit = Q.depth;
            need_new = false;
            want_aov = (frame == 0 && sample == 0 && Q.aov_prim != nullptr);
        }
    };

    // Lane modes.  SHADE: the lane's ray (if any) has its closest hit and waits for shading and
    // for its next ray; WALK: mid-traversal; DONE: no frames left.  Shading and ray set-up cost
    // an order of magnitude more instructions than one node visit, and most rays need only a
    // couple of visits while a few need a hundred, so the wave keeps stepping the walkers and
    // only runs the shading block once enough lanes wait for it (or nobody is walking).
    enum : uint32_t { M_SHADE = 0, M_WALK = 1, M_DONE = 2 };
    uint32_t mode = active ? M_SHADE : M_DONE;
    WalkU w;
    w.rec.hit = w.rec.is_tri = false;
    w.rec.t = w.rec.u = w.rec.v = 0.0f;
    w.rec.index = 0;
    w.inv = V3(0.0f, 0.0f, 0.0f);
    w.nearest = 9999.0f;
    w.node = 0;
    w.sp = 0;
    w.leaf_left = 0;
    w.top = 0;
    // The guard tree (PtDeviceScene::guard_on), as in the pooled kernel: a ray in reach that has passed the reference root's
    // boxes walks it; a triangle found there is shaded only if the reference walk is certain to return the same one, and
    // otherwise the lane sets the same ray up again (redo) and walks the reference tree.
    bool on_guard = false, redo = false;

    for (;;) {
        const unsigned long long waiting = __ballot(mode == M_SHADE), walking = __ballot(mode == M_WALK);
        if ((waiting | walking) == 0ull) break;
        cost_trips++;
        if (walking == 0ull || (uint32_t)__popcll(waiting) >= P.refill_lanes) {
            cost_refills++;
#ifdef PT_DIAG
            diag_iters++;
            diag_s = clock64();
#endif
            // (the shading block re-reads the parameter block, so that the walk loop does not have to keep
            // all of it in scalar registers)
            // (stage by stage, so that no stage's scalars are live during another one)
            if (mode == M_SHADE) {
                bool path_done = false;
                v3 result = V3(0.0f, 0.0f, 0.0f);
                if constexpr (!STATS) {
                    if (!need_new && on_guard) {
                        if ((w.rec.index & PT_ID_ABANDONED) != 0u) {
                            redo = true;
                        } else if (w.rec.hit && w.rec.is_tri) {
                            PT_FRESH_PARAMS(L);
                            redo = (w.rec.index & PT_ID_TIE) != 0u || !pt_guard_hit_is_certain(L.sc, cur, w.rec);
                        }
                    }
                }
                if (!need_new && !redo) {
                    PT_FRESH_PARAMS(L);
                    uint32_t prim;
                    path_done = shade_ray<STATS, GIZMO>(L.sc, cur, w.rec, color, seed, bounce, limit, initial_d, result, prim, cn);
                    if (want_aov) {
                        L.aov_prim[pix] = prim;
                        L.aov_t[pix] = w.rec.hit ? w.rec.t : 0.0f;
                        want_aov = false;
                    }
                }
                if (!redo) {
                    PT_FRESH_PARAMS(L);
                    advance(L, path_done, result);
                }
                if (active) {
                    PT_FRESH_PARAMS(L);
                    if (redo) cn.rewalks++;
                    else cn.rays++;
                    mode = walk_begin<STATS>(L.sc, cur, w, stack, cn, !redo && bounce == 0u && pt_camera_ray_misses_bvh(L.cull, x, y)) ? M_WALK : M_SHADE;
                    w.leaf_left = 0;
                    on_guard = false;
                    if constexpr (!STATS) {
                        // (this kernel walks the guard tree in its four-child form only: that walk never needs more stack than
                        // there is, so the stack stays sized by the reference tree)
                        if (mode == M_WALK && !redo && L.sc.guard_quads != 0u && pt_guard_in_reach(L.sc, cur.o, cur.d)) {
                            on_guard = true;
                            cn.guarded++;
                            w.node = L.sc.guard_quads;
                            w.sp = 0;
                            if (w.rec.hit) w.nearest = pt_min(9999.0f, w.rec.t);
                        }
                    }
                    redo = false;
                    if (mode == M_WALK) walk_enter(w);
                } else {
                    mode = M_DONE;
                }
            }
#ifdef PT_DIAG
            { const unsigned long long t = clock64(); diag_b += t - diag_s; }
#endif
        }
#ifdef PT_DIAG
        diag_s = clock64();
        diag_steps++;
        {
            const unsigned long long wk = __ballot(mode == M_WALK);
            const unsigned long long lf = __ballot(mode == M_WALK && (w.leaf_left != 0u || (w.node & PT_TOKEN_LEAF)));
            diag_c += ((unsigned long long)__popcll(wk) << 40) | ((unsigned long long)__popcll(lf) << 20) | (lf ? 1ull : 0ull);
        }
        if ((diag_steps & 1023u) == 0 && blockIdx.x < 64 && (diag_steps >> 10) <= 8 && lane == 0)
            g_diag_marks[blockIdx.x * 8 + (diag_steps >> 10) - 1] = __builtin_amdgcn_s_memrealtime();
#endif
        if (mode == M_WALK) {
#ifdef PT_TUNING
            // (tuning builds: PTRT_POOL_START, which this kernel has no other use for, caps the stack that a walk of the four-child
            // form may use -- the way to exercise its giving up, which real scenes never trigger)
            const uint32_t stack_cap = min(OVERFLOW ? P.gstack_cap : (uint32_t)PT_LDS_STACK, P.pool_start ? P.pool_start : 0xFFFFFFFFu);
#else
            const uint32_t stack_cap = OVERFLOW ? P.gstack_cap : (uint32_t)PT_LDS_STACK;
#endif
            if (walk_step_u<STATS>(sc, cur, w, stack, cn, stack_cap)) mode = M_SHADE;
        }
#ifdef PT_DIAG
        { const unsigned long long t = clock64(); diag_a += t - diag_s; }
#endif
    }

    if (!DIRECT) P.partial[(size_t)blockIdx.x * 64u + lane] = lds_acc[lane];  // zeros for lanes outside the image
    else if (in_image) P.accum[pix] = lds_acc[lane];

    // Deterministic cost estimate (loop trips, a shading pass weighted like eight node visits), so
    // that the plan -- and with it the order of the partial sums -- does not depend on timing.
    if (P.cost && lane == 0) atomicAdd(&P.cost[tile], cost_trips + 8u * cost_refills);
#ifdef PT_DIAG
    {
        const uint32_t wid = blockIdx.x;
        uint32_t it = diag_iters;
        for (int off = 32; off > 0; off >>= 1) it = max(it, (uint32_t)__shfl_xor(it, off, 64));
        const uint32_t wrays = wave_sum(cn.rays);
        if (lane == 0 && wid < PT_DIAG_MAX_WAVES) {
            g_diag[8 * wid + 0] = diag_t0;
            g_diag[8 * wid + 1] = __builtin_amdgcn_s_memrealtime();
            g_diag[8 * wid + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
            g_diag[8 * wid + 3] = ((unsigned long long)wrays << 32) | it;
            g_diag[8 * wid + 4] = diag_a;
            g_diag[8 * wid + 5] = diag_b;
            g_diag[8 * wid + 6] = diag_c;
            g_diag[8 * wid + 7] = diag_steps;
        }
    }
#endif
    // one atomic per wave and counter
    unsigned long long *const stripe_words = &P.counters[(blockIdx.x % PT_COUNTER_STRIPES) * PT_COUNTER_PITCH];
    pt_add_counters<STATS>(stripe_words, cn, lane);
}

#ifndef PT_INLINE_SPHERE_HITS
#define PT_INLINE_SPHERE_HITS 1
#endif

// ---- the pooled megakernel -----------------------------------------------------------------
// Same per-ray arithmetic as trace_kernel, different division of labour.  In an outdoor scene only
// about one ray in ten needs a real BVH walk and those walks are long, so a wave that walks its own
// rays steps ~10 live lanes out of 64, and a tile's pixels finish at very different times.  Here
// neither paths nor pixels are tied to lanes:
//   * a workgroup of PT_PP_WAVES persistent waves keeps PT_PP_PATHS path slots with their whole state
//     in LDS, structure-of-arrays [field][slot]; a slot works on one JOB = one pixel of one work item
//     (its frames in order, so the pixel's sums are the same sequential sums as in trace_kernel) and
//     takes the next job when it is done: the group hands out the pixels of its open work item in
//     order and opens the next item from the launch's item counter;
//   * a slot id is always in exactly one place: one of the two READY rings (its ray has its closest
//     hit: one ring for hits, one for misses and slots that have no ray yet), the PENDING ring (its
//     ray survived the root visit and needs a walk), or in the hands of one lane;
//   * a SHADING PASS pops up to 64 slots from one READY ring (so all its lanes run the same kind of
//     code), shades them, sets up their next rays (spheres + the root's child pair) and pushes each
//     slot to PENDING or back to a READY ring;
//   * every lane also has a WALK SLOT in registers; a wave tops its free walk slots up from PENDING,
//     steps all its walkers a few node visits per trip and pushes finished ones to READY;
//   * pending rays go to the wave that already walks the most, so walks concentrate on few
//     well-filled waves; a wave that found nothing to do for a while ignores such preferences, so
//     they can never stall the group;
//   * the running sums live in global memory (the item's partial-sum row, or the accumulation buffer
//     itself in DIRECT mode) and are only touched by the lane that holds the slot.
// Rings are multi-producer/multi-consumer: producers reserve positions with an atomic add on the
// tail and then fill them; consumers claim positions with a compare-and-swap on the head and wait
// for the entry to be filled (0 = not yet); a producer waits for the entry it is about to fill to be
// empty.  Every wait is on a strictly older ring position, so waits cannot form a cycle.
#ifndef PT_PP_WAVES
#define PT_PP_WAVES 4
#endif
#define PT_PP_THREADS (64 * PT_PP_WAVES)
#ifndef PT_PP_PATHS
#define PT_PP_PATHS 320        // path slots of the plain variant (21 words per slot: four workgroups per CU)
#endif
#ifndef PT_PP_PATHS_WIDE
#define PT_PP_PATHS_WIDE 256   // ... of the variants with wider slots, so that four workgroups per CU still fit
#endif
#ifndef PT_PP_QCAP
#define PT_PP_QCAP 512   // ring capacity: a power of two >= PT_PP_PATHS
#endif
#ifndef PT_PP_STACK
#define PT_PP_STACK 8    // traversal stack entries per walk slot kept in LDS; deeper ones spill
#endif
#define PT_PP_MAX_SPIN (1u << 20)
// Scheduling thresholds of the pooled kernel.  They are compile-time constants (the values measured best on the
// benchmark scenes); only a -DPT_TUNING build (make tuning) reads them from the parameter block, where the PTRT_*
// environment variables of the sweep scripts put them.
#ifdef PT_TUNING
#define PT_TUNE(field, value) (field)
#else
#define PT_TUNE(field, value) (value)
#endif
#ifndef PT_PP_WAVES_PER_EU
#define PT_PP_WAVES_PER_EU 4
#endif
#define PT_PP_ITEM_BATCH 8u  // most work items claimed per trip to the launch's item counter (half as many in launches with few items per workgroup)
static_assert(PT_PP_ITEM_BATCH % 8u == 0, "item batches are staged four records at a time, whole or half");
static_assert(PT_PP_PATHS % PT_PP_WAVES == 0 && PT_PP_PATHS <= PT_PP_QCAP && (PT_PP_QCAP & (PT_PP_QCAP - 1)) == 0, "pool geometry");
static_assert(PT_PP_PATHS_WIDE % PT_PP_WAVES == 0 && PT_PP_PATHS_WIDE <= PT_PP_PATHS, "pool geometry");

enum : uint32_t {
    F_OX, F_OY, F_OZ, F_DX, F_DY, F_DZ,  // the path's current ray
    F_T, F_ID,                           // closest hit so far: t, index | is_tri << 30 | hit << 31
    F_U, F_V,                            // barycentrics; while PENDING: node token to visit, far child (or none)
    F_COLR, F_COLG, F_COLB, F_SEED, F_BOUNCE,
    F_FRAME,                             // frame being rendered
    F_FS,                                // have_ray | want_aov << 1 | reference tree only << 2 | sample << 3
    F_XY, F_FEND, F_PIDX,                // the job: pixel x | y << 16, end frame, index of its float4 of sums
    F_COST,                              // work estimate of the job so far
    F_BASE
};
template <bool GIZMO, bool SPP1>
struct PPFields {
    static constexpr uint32_t RNG = F_BASE;                      // HybridTaus state (4) + sample sum (3), only when spp > 1
    static constexpr uint32_t FSUM = F_BASE + 4u;
    static constexpr uint32_t IDIR = F_BASE + (SPP1 ? 0u : 7u);  // first direction (3) + bounce limit, only with a gizmo sphere
    static constexpr uint32_t LIMIT = IDIR + 3u;
    static constexpr uint32_t COUNT = IDIR + (GIZMO ? 4u : 0u);
    // Path slots per workgroup.  160 KiB of LDS hold four workgroups of 320 plain slots (21 words); with the 7 words
    // of spp > 1 or the 4 of a gizmo sphere only three such groups fit, so those variants take 256 slots and keep
    // four groups (and four waves per SIMD); with both extensions three groups of 320 are the better fit.
    static constexpr uint32_t PATHS = (GIZMO != !SPP1) ? (uint32_t)PT_PP_PATHS_WIDE : (uint32_t)PT_PP_PATHS;
    static constexpr uint32_t SEED = PATHS / PT_PP_WAVES;  // slots each wave gives their first job
};

typedef uint32_t pp_u32x4 __attribute__((ext_vector_type(4)));
enum : uint32_t { R_MISS = 0, R_HIT = 1, R_PEND = 2 };
// Ring entries are slot id + 1 in 16 bits.
typedef uint16_t pp_entry;
__device__ __forceinline__ uint32_t ring_load(pp_entry *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ring_store(pp_entry *p, uint32_t v) { __hip_atomic_store(p, (pp_entry)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <uint32_t NF, uint32_t PATHS>
struct PPShared {
    uint32_t st[NF][PATHS];
    uint32_t stack[PT_PP_STACK][PT_PP_THREADS];
    // rings: R_MISS = READY, ray missed everything (or no ray yet): sky, next path; R_HIT = READY, ray hit
    // something: material, bounce; R_PEND = PENDING, ray needs a walk
    pp_entry ring[3][PT_PP_QCAP];
    uint32_t head[3], tail[3];
    uint32_t busy[PT_PP_WAVES];  // walkers per wave, published every trip
    uint32_t live;               // slots that still have, or may still get, a job
    // the group's open work item: its pixels are handed out in order to slots that need a job
    uint32_t job_word;           // generation << 8 | pixels handed out (64 = all)
    uint32_t job_lock;           // held by the lane that opens the next item
    uint32_t no_more;            // the launch's item counter is exhausted
    uint32_t group;              // the workgroup's index, parked here for the end of the kernel
    pp_u32x4 item;               // x0 | y0 << 16, first frame, end frame, item index (read after an acquire fence)
    // Work items fetched ahead: the launch's item counter and the item table are global memory, two dependent round
    // trips that the group would otherwise pay every 64 pixels; they are paid once per PT_PP_ITEM_BATCH items (records
    // loaded side by side) and the others are opened from here.  Touched only by the holder of job_lock.
    uint32_t staged[PT_PP_ITEM_BATCH][4];  // {tile, first frame, frames, item index}
    uint32_t staged_pos;                   // next entry to open (PT_PP_ITEM_BATCH: none left)
};

// Hand-shake words are accessed with relaxed workgroup-scope atomics (plain ds_read/ds_write that the
// compiler neither caches nor reorders across the fences around them).
__device__ __forceinline__ uint32_t lds_load(uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// Same word read by every lane of the wave: hands the value over as a scalar, so that what is computed from it stays
// in scalar registers and scalar branches instead of lane masks.
__device__ __forceinline__ uint32_t lds_load_uniform(uint32_t *p) { return __builtin_amdgcn_readfirstlane(lds_load(p)); }

struct PPStack {
    uint32_t *lds;   // &sh.stack[0][tid]
    uint32_t *spill;
    uint32_t pitch;  // threads of the launch (at most 2^18): entry e of this thread is spill[(e - PT_PP_STACK) * pitch]
    __device__ __forceinline__ void push(uint32_t sp, uint32_t v)
    {
        if (sp < PT_PP_STACK) lds[sp * PT_PP_THREADS] = v;
        else spill[(size_t)(sp - PT_PP_STACK) * pitch] = v;
    }
    __device__ __forceinline__ uint32_t pop(uint32_t sp)
    {
        // the LDS read always (clamped), the spill read as a rare branch: keeps the common case a plain ds_read
        uint32_t v = lds[min(sp, (uint32_t)PT_PP_STACK - 1u) * PT_PP_THREADS];
        asm volatile("" : "+v"(v));  // keeps the two reads from being merged into one generic-address load
        if (sp >= PT_PP_STACK) v = spill[(size_t)(sp - PT_PP_STACK) * pitch];
        return v;
    }
    // The same read for a value that is not needed at once: nothing here makes the caller wait for the LDS read
    // (the statement that keeps the two reads apart sits in the rare branch).
    // Entry n - 1 of a stack of n entries, for a caller that needs it later: an entry of the LDS part is read into
    // `later` (true is returned; of an empty stack: whatever the last LDS entry holds -- no branch, no fault), so that
    // nothing but that LDS read stands between the caller and the value; an entry of the spilled part is loaded into
    // `now` on the spot (false).
    __device__ __forceinline__ bool peek_newest(uint32_t n, uint32_t &later, uint32_t &now)
    {
        const uint32_t e = n - 1u;
        later = lds[min(e, (uint32_t)PT_PP_STACK - 1u) * PT_PP_THREADS];
        const bool deep = (int32_t)e >= (int32_t)PT_PP_STACK;
        if (deep) now = spill[(size_t)(e - PT_PP_STACK) * pitch];
        return !deep;
    }
};

// Self-checks of the pooled kernel.  None of them is expected to fire; each costs a compare on a path that is taken
// anyway, and a violated one names itself in the launch's error word (bit per kind, read by the host after a
// synchronisation) instead of showing up later as a hang or a stray memory access.
// F_ID of a path slot: bit 31 the ray has hit something, bit 30 a triangle (else a sphere), bit 28 PT_ID_TIE, bits 27..0
// the sphere's index or the table position of the triangle's record.
#define PT_KERR_RING_OVERWRITE 1u  // a ring position was still occupied after the bounded wait (entry overwritten)
#define PT_KERR_RING_EMPTY 2u      // a claimed ring position was never filled (slot id lost)
#define PT_KERR_SLOT_ID 4u         // a ring handed out a slot id beyond the pool
#define PT_KERR_SUMS_INDEX 8u      // a slot's index into the sums lies outside the buffer
#define PT_KERR_JOB_WAIT 16u       // the hand-out of pixel jobs ran out of its bounded wait (lock held, or the claim kept losing)
__device__ __forceinline__ void pt_mark_error(uint32_t bit)
{
    PtTraceParams L;  // (re-read: only the word of the error pointer is loaded)
    pt_fresh_params(L);
    atomicOr(L.next_item + 2, bit);
}

__device__ __forceinline__ void ring_wait_empty(pp_entry *e)
{
    for (uint32_t spin = 0; ring_load(e) != 0u; spin++) {
        if (spin >= PT_PP_MAX_SPIN) { pt_mark_error(PT_KERR_RING_OVERWRITE); break; }
        __builtin_amdgcn_s_sleep(1);
    }
}

// Wave-level push: every lane with `mine` appends `id`.  Called from wave-uniform code.
__device__ __forceinline__ void ring_push(pp_entry *ring, uint32_t *tail, bool mine, uint32_t id, uint32_t lane, unsigned long long lt_mask)
{
    const unsigned long long m = __ballot(mine);
    if (m == 0ull) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the slot's state is written before its id shows up
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(tail, (uint32_t)__popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (mine) {
        pp_entry *e = &ring[(base + (uint32_t)__popcll(m & lt_mask)) & (PT_PP_QCAP - 1u)];
        // an older entry not yet taken out: never seen in practice; bounded so that a bug cannot hang the GPU
        ring_wait_empty(e);
        ring_store(e, id + 1u);
    }
}

// Three pushes at once (a lane appends to at most one of the rings): one fence, and the three tail
// reservations made by lanes 0..2 in the same instruction instead of one after the other.  `tails` and `rings`
// are the three rings' tail words and entry arrays, in ring order.
__device__ __forceinline__ void ring_push3(pp_entry (*rings)[PT_PP_QCAP], uint32_t *tails, bool m0, bool m1, bool m2, uint32_t id, uint32_t lane,
                                           unsigned long long lt_mask)
{
    const unsigned long long b0 = __ballot(m0), b1 = __ballot(m1), b2 = __ballot(m2);
    if ((b0 | b1 | b2) == 0ull) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the slots' state is written before their ids show up
    uint32_t base = 0;
    {
        uint32_t l = lane;
        asm volatile("" : "+v"(l));  // (otherwise "lane < 3" is computed once and sits in a scalar register pair for the whole kernel)
        const uint32_t n = l == 0 ? (uint32_t)__popcll(b0) : l == 1 ? (uint32_t)__popcll(b1) : (uint32_t)__popcll(b2);
        if (l < 3u && n != 0u) base = atomicAdd(&tails[l], n);
    }
    const uint32_t base0 = __shfl(base, 0, 64), base1 = __shfl(base, 1, 64), base2 = __shfl(base, 2, 64);
    if (m0 || m1 || m2) {
        const uint32_t k = m0 ? 0u : m1 ? 1u : 2u;
        const uint32_t pos = m0 ? base0 + (uint32_t)__popcll(b0 & lt_mask) : m1 ? base1 + (uint32_t)__popcll(b1 & lt_mask) : base2 + (uint32_t)__popcll(b2 & lt_mask);
        pp_entry *e = &rings[k][pos & (PT_PP_QCAP - 1u)];
        ring_wait_empty(e);
        ring_store(e, id + 1u);
    }
}

// Wave-level pop: the lanes of `want` take up to popcount(want) ids; returns 0xFFFFFFFF for no id.
__device__ __forceinline__ uint32_t ring_pop(pp_entry *ring, uint32_t *head, uint32_t *tail, unsigned long long want, uint32_t lane,
                                            unsigned long long lt_mask, uint32_t max_id)
{
    uint32_t h = 0, n = 0;
    if (lane == 0) {
        for (int attempt = 0; attempt < 4; attempt++) {
            h = lds_load(head);
            const uint32_t avail = lds_load(tail) - h;
            if (avail == 0u || avail > PT_PP_QCAP) break;  // empty (or a torn pair of reads: try next trip)
            const uint32_t m = min(avail, (uint32_t)__popcll(want));
            if (atomicCAS(head, h, h + m) == h) { n = m; break; }
        }
    }
    h = __builtin_amdgcn_readfirstlane(h);
    n = __builtin_amdgcn_readfirstlane(n);
    uint32_t id = 0xFFFFFFFFu;
    if (n == 0u) return id;
    const uint32_t rank = (uint32_t)__popcll(want & lt_mask);
    if (((want >> lane) & 1ull) && rank < n) {
        pp_entry *e = &ring[(h + rank) & (PT_PP_QCAP - 1u)];
        // reserved by its producer, about to be filled (bounded like the wait above; a slot lost here
        // stays "live", so the waves' idle limit ends the kernel and the error mark is set)
        uint32_t v = ring_load(e);
        for (uint32_t spin = 0; v == 0u && spin < PT_PP_MAX_SPIN; spin++) {
            __builtin_amdgcn_s_sleep(1);
            v = ring_load(e);
        }
        if (v != 0u) {
            ring_store(e, 0u);
            id = v - 1u;
            if (id >= max_id) { pt_mark_error(PT_KERR_SLOT_ID); id = 0xFFFFFFFFu; }
        } else {
            pt_mark_error(PT_KERR_RING_EMPTY);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return id;
}

// In long launches a tile's cost for the next plan is the sum over a fixed quarter of its pixels (mask 1: the even ones
// of the even rows; a tile starts at even coordinates, so every tile with a pixel in the image has one of them): a launch
// that measures pays one global atomic per pixel job that reports, and the bench has 16 million pixel jobs per launch.
// Short launches (mask 0) let every pixel report: few jobs, and few frames to average a tile's cost over.
__device__ __forceinline__ bool pt_cost_sample(uint32_t x, uint32_t y, uint32_t mask) { return ((x | y) & mask) == 0u; }

template <bool STATS, bool GIZMO, bool SPP1, bool DIRECT>
__global__ __launch_bounds__(PT_PP_THREADS, PT_PP_WAVES_PER_EU) void pool_kernel(const PtTraceParams P)
{
    using FL = PPFields<GIZMO, SPP1>;
    __shared__ PPShared<FL::COUNT, FL::PATHS> sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // the wave's index, as a scalar: comparisons with it need no lane masks
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const PtDeviceScene &sc = P.sc;
    for (uint32_t i = tid; i < PT_PP_QCAP; i += PT_PP_THREADS) { sh.ring[R_MISS][i] = 0; sh.ring[R_HIT][i] = 0; sh.ring[R_PEND][i] = 0; }
    if (tid == 0) {
        for (uint32_t i = 0; i < 3; i++) sh.head[i] = sh.tail[i] = 0;
        for (uint32_t i = 0; i < PT_PP_WAVES; i++) sh.busy[i] = 0;
        sh.live = FL::PATHS;
        sh.job_word = 64u;
        sh.job_lock = 0;
        sh.no_more = 0;
        sh.staged_pos = PT_PP_ITEM_BATCH;
        sh.group = blockIdx.x;
    }
    __syncthreads();

    PPStack stack{&sh.stack[0][tid], P.gstack ? P.gstack + ((size_t)blockIdx.x * PT_PP_THREADS + tid) : nullptr, gridDim.x * PT_PP_THREADS};
    Counters cn;

    // Gives each lane with `need` a job for slot p, in lane order, from the group's open work item
    // (opening the next one when its 64 pixels are handed out); false for the lanes left without a
    // job because the launch has no items left.  Wave-uniform control flow.
    auto assign_jobs = [&](bool need, uint32_t p) -> bool {
        PT_FRESH_PARAMS(L);
        float4 *const sums = DIRECT ? L.accum : L.partial;
        bool got = false;
        for (;;) {
            const unsigned long long nm = __ballot(need && !got);
            if (nm == 0ull) break;
            uint32_t first = 0, take = 0, x0 = 0, y0 = 0, fb = 0, fend = 0, index = 0;
            if (lane == 0) {
                uint32_t spin = 0;
                for (; spin < PT_PP_MAX_SPIN; spin++) {
                    const uint32_t jw = lds_load(&sh.job_word);  // generation << 8 | pixels handed out
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const uint32_t done = jw & 0xFFu;
                    if (done < 64u) {
                        // read the item first, claim second: the claim fails if the item was replaced meanwhile
                        const pp_u32x4 rec = sh.item;  // one 16-byte LDS read
                        x0 = rec.x & 0xFFFFu; y0 = rec.x >> 16; fb = rec.y; fend = rec.z; index = rec.w;
                        const uint32_t t = min((uint32_t)__popcll(nm), 64u - done);
                        if (atomicCAS(&sh.job_word, jw, jw + t) == jw) { first = done; take = t; break; }
                        continue;
                    }
                    if (lds_load(&sh.no_more) != 0u) break;
                    if (atomicCAS(&sh.job_lock, 0u, 1u) != 0u) { __builtin_amdgcn_s_sleep(1); continue; }
                    if (lds_load(&sh.job_word) == jw) {
                        // still handed out completely, and nobody else is replacing it: open the next item
                        uint32_t it = 0, tile = 0, f0 = 0, fc = 0;
                        bool have = false;
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // (the staged items were written by an earlier holder of the lock)
                        for (;;) {
                            uint32_t pos = lds_load(&sh.staged_pos);
                            if (pos >= PT_PP_ITEM_BATCH) {
                                // one trip to the launch's counter for a batch of items; their records are loaded side by side
                                const uint32_t nb = L.item_batch;  // (PT_PP_ITEM_BATCH or half of it: launch_pool)
                                const uint32_t base = atomicAdd(L.next_item, nb);
                                const uint32_t skip = PT_PP_ITEM_BATCH - nb;  // (the batch sits at the end of the staging array)
                                // (four records at a time: eight would take the registers of a spill)
#pragma nounroll
                                for (uint32_t k0 = skip; k0 < PT_PP_ITEM_BATCH; k0 += 4u) {
                                    uint4 recs[4];
#pragma unroll
                                    for (uint32_t k = 0; k < 4u; k++) {
                                        const uint32_t i = base + (k0 + k - skip);
                                        recs[k] = make_uint4(0u, 0u, 0u, 0u);
                                        if (i < L.n_items) {
                                            if (DIRECT) recs[k] = make_uint4(i, 0u, L.n_frames, 0u);
                                            else recs[k] = L.items[i];
                                        }
                                    }
#pragma unroll
                                    for (uint32_t k = 0; k < 4u; k++) {
                                        lds_store(&sh.staged[k0 + k][0], recs[k].x);
                                        lds_store(&sh.staged[k0 + k][1], recs[k].y);
                                        lds_store(&sh.staged[k0 + k][2], recs[k].z);
                                        lds_store(&sh.staged[k0 + k][3], base + (k0 + k - skip));
                                    }
                                }
                                pos = skip;
                            }
                            tile = lds_load(&sh.staged[pos][0]);
                            f0 = lds_load(&sh.staged[pos][1]);
                            fc = lds_load(&sh.staged[pos][2]);
                            it = lds_load(&sh.staged[pos][3]);
                            lds_store(&sh.staged_pos, pos + 1u);
                            if (it >= L.n_items) break;            // (items are claimed in rising order: none left)
                            if (fc != 0u) { have = true; break; }  // zero frames: an unused entry of the item table
                        }
                        if (!have) {
                            lds_store(&sh.no_more, 1u);
                        } else {
                            const uint32_t tby = tile / L.tiles_x, tbx = tile - tby * L.tiles_x;
                            sh.item = pp_u32x4{(tbx * PT_TILE) | ((tby * PT_TILE) << 16), f0, f0 + fc, it};
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            lds_store(&sh.job_word, ((jw >> 8) + 1u) << 8);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    lds_store(&sh.job_lock, 0u);
                }
                // (the loop is left by a claim or by the end of the items; running out of trips -- the lock never released,
                // the claim lost a million times -- would otherwise pass for "no items left" and drop the staged items silently)
                if (spin >= PT_PP_MAX_SPIN) pt_mark_error(PT_KERR_JOB_WAIT);
            }
            take = __builtin_amdgcn_readfirstlane(take);
            if (take == 0u) break;  // no items left
            first = __builtin_amdgcn_readfirstlane(first);
            x0 = __builtin_amdgcn_readfirstlane(x0); y0 = __builtin_amdgcn_readfirstlane(y0);
            fb = __builtin_amdgcn_readfirstlane(fb); fend = __builtin_amdgcn_readfirstlane(fend);
            index = __builtin_amdgcn_readfirstlane(index);
            const uint32_t rank = (uint32_t)__popcll(nm & lt_mask);
            if (need && !got && rank < take) {
                const uint32_t pix = first + rank;
                const uint32_t x = x0 + (pix & 7u), y = y0 + (pix >> 3);
                const bool in_image = (x < L.cols) && (y < L.rows) && L.spp > 0;
                const uint32_t pidx = DIRECT ? y * L.width + x : index * 64u + pix;
                if (!DIRECT) sums[pidx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // also the rows of pixels outside the image
                if (in_image) {
                    sh.st[F_XY][p] = x | (y << 16);
                    sh.st[F_FEND][p] = fend;
                    sh.st[F_PIDX][p] = pidx;
                    sh.st[F_COST][p] = 0u;
                    sh.st[F_FRAME][p] = fb;
                    sh.st[F_FS][p] = 0u;
                    if constexpr (!SPP1) { sh.st[FL::FSUM][p] = 0u; sh.st[FL::FSUM + 1][p] = 0u; sh.st[FL::FSUM + 2][p] = 0u; }
                    got = true;
                }
            }
        }
        return got;
    };

    // every wave gives its share of the slots their first job
    for (uint32_t base = 0; base < FL::SEED; base += 64u) {
        const bool need = base + lane < FL::SEED;
        const uint32_t p = wv * FL::SEED + base + lane;
        const bool got = assign_jobs(need, p);
        ring_push(sh.ring[R_MISS], &sh.tail[R_MISS], got, p, lane, lt_mask);
        const uint32_t n_dead = (uint32_t)__popcll(__ballot(need && !got));
        if (n_dead && lane == 0) atomicSub(&sh.live, n_dead);
    }

    // ---- walk slot ----
    bool wbusy = false;
    uint32_t wpath = 0, wsteps = 0;
    Ray wray;
    wray.o = wray.d = V3(0.0f, 0.0f, 0.0f);
    WalkU w;
    w.rec.hit = w.rec.is_tri = false;
    w.rec.t = w.rec.u = w.rec.v = 0.0f;
    w.rec.index = 0;
    w.inv = V3(0.0f, 0.0f, 0.0f);
    w.nearest = 9999.0f;
    w.node = 0;
    w.sp = 0;
    w.leaf_left = 0;
    w.top = 0;

#ifdef PT_DIAG
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long dg_trips = 0, dg_idle = 0, dg_shades = 0, dg_ready = 0, dg_shade_cyc = 0, dg_steps = 0, dg_busy = 0, dg_walk_cyc = 0,
                       dg_claim_cyc = 0, dg_claims = 0, dg_s = 0, dg_hit_passes = 0, dg_hit_lanes = 0, dg_hit_cyc = 0, dg_s2 = 0;
    unsigned long long dg_sec[7] = {0, 0, 0, 0, 0, 0, 0}, dg_rounds = 0, dg_round_lanes = 0, dg_extra[3] = {0, 0, 0};
    unsigned long long dg_hist[5] = {0, 0, 0, 0, 0};
    unsigned long long dg_len_n[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg_len_steps[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // finished walks by length
#endif
    // Every wave leaves the loop: normally when no slot is live any more, and in any case after
    // max_idle consecutive trips without anything to do (a lost hand-shake would otherwise hang the
    // GPU; the check after the loop reports it).
    uint32_t idle = 0;
    while (idle < PT_TUNE(P.max_idle, PT_POOL_MAX_IDLE)) {
        bool did_work = false;
        // (the wave's index, derived anew every trip behind an opaque copy of the thread id: two instructions, instead of a
        // scalar register held -- or spilled -- for the whole loop)
        uint32_t tid_now = tid;
        asm volatile("" : "+v"(tid_now));
        const uint32_t wv = __builtin_amdgcn_readfirstlane(tid_now >> 6);
        const bool relaxed = idle >= 2u;      // nothing to do twice in a row: shade whatever is ready
        const bool desperate = idle >= 32u;   // still nothing: walk whatever is pending
#ifdef PT_DIAG
        dg_trips++;
#endif
        // wave-uniform view of the pool
        uint32_t n_pend = lds_load_uniform(&sh.tail[R_PEND]) - lds_load_uniform(&sh.head[R_PEND]);
        if (n_pend > PT_PP_QCAP) n_pend = 0;
        uint32_t n_miss = lds_load_uniform(&sh.tail[R_MISS]) - lds_load_uniform(&sh.head[R_MISS]);
        if (n_miss > PT_PP_QCAP) n_miss = 0;
        uint32_t n_hit = lds_load_uniform(&sh.tail[R_HIT]) - lds_load_uniform(&sh.head[R_HIT]);
        if (n_hit > PT_PP_QCAP) n_hit = 0;
        const unsigned long long free_mask = __ballot(!wbusy);
        const uint32_t n_free = (uint32_t)__popcll(free_mask);

#ifdef PT_DIAG
        dg_s = clock64();
#endif
        // (a) top the walk slots up from PENDING.  Pending rays go to the wave that already has the
        // most walkers and still has room (ties to the lower wave index), so that walks concentrate
        // on few, well-filled waves; the others take rays only when so many wait that the leader
        // cannot hold them.
        // (walkers * waves + (waves - 1 - index) orders the waves by walkers, ties to the lower index; a full wave is out)
        int32_t best_other = -1;
#pragma nounroll  // (unrolled, the four comparisons with the wave's index are hoisted out of the main loop as lane masks)
        for (uint32_t o = 0; o < PT_PP_WAVES; o++) {
            const uint32_t ob = lds_load_uniform(&sh.busy[o]);
            const int32_t key = (o != wv && ob < 64u) ? (int32_t)(ob * PT_PP_WAVES + (PT_PP_WAVES - 1u - o)) : -1;
            best_other = max(best_other, key);
        }
        const bool leader = best_other < (int32_t)((64u - n_free) * PT_PP_WAVES + (PT_PP_WAVES - 1u - wv));
        if (n_pend != 0u && n_free != 0u &&
            (desperate || (leader ? (n_free >= PT_TUNE(P.pool_topup, PT_POOL_TOPUP) || n_pend >= n_free || n_free == 64u) : n_pend >= PT_TUNE(P.pool_start, PT_POOL_START) + 64u))) {
            const uint32_t id = ring_pop(sh.ring[R_PEND], &sh.head[R_PEND], &sh.tail[R_PEND], free_mask, lane, lt_mask, FL::PATHS);
            if (id != 0xFFFFFFFFu) {
                const uint32_t p = id;
                wpath = p;
                wray.o = V3(__uint_as_float(sh.st[F_OX][p]), __uint_as_float(sh.st[F_OY][p]), __uint_as_float(sh.st[F_OZ][p]));
                wray.d = V3(__uint_as_float(sh.st[F_DX][p]), __uint_as_float(sh.st[F_DY][p]), __uint_as_float(sh.st[F_DZ][p]));
                const uint32_t hid = sh.st[F_ID][p];
                w.rec.hit = (hid >> 31) != 0;
                w.rec.is_tri = false;
                w.rec.index = hid & PT_ID_INDEX;
                w.rec.t = __uint_as_float(sh.st[F_T][p]);
                w.rec.u = w.rec.v = 0.0f;
                w.inv = pt_rcp3(wray.d);
                w.nearest = 9999.0f;
                w.node = sh.st[F_U][p];
                w.sp = 0;
                w.leaf_left = 0;
                walk_enter(w);
                const uint32_t k2 = sh.st[F_V][p];
                if (k2 != 0xFFFFFFFFu) { w.top = k2; w.sp = 1u; }  // (the newest stack entry stays in a register)
                wsteps = 0;
                if constexpr (!STATS) {
                    // A ray within reach of the mesh leaves the reference tree here -- it has passed the reference root's
                    // boxes -- and walks the guard tree from its root, looking only for triangles in front of its sphere hit.
                    // (The pass over hits that shades this ray's hit makes the same test again to know which tree it was.)
                    PT_FRESH_PARAMS(LG);
                    if (LG.sc.guard_on != 0u && pt_ray_walks_guard(LG.sc, wray.o, wray.d, sh.st[F_FS][p])) {
                        w.node = LG.sc.guard_root;
                        w.sp = 0;
                        w.leaf_left = 0;
                        if (w.rec.hit) w.nearest = pt_min(9999.0f, w.rec.t);
                        cn.guarded++;
                    }
                }
                wbusy = true;
            }
            const unsigned long long gotm = __ballot(id != 0xFFFFFFFFu);
            if (gotm) did_work = true;
#ifdef PT_DIAG
            dg_claims += __popcll(gotm);
#endif
        }
#ifdef PT_DIAG
        dg_claim_cyc += clock64() - dg_s;
        dg_s = clock64();
#endif

        // (b) step the walkers a few node visits; finished ones go back to READY
        {
            const uint32_t nb = (uint32_t)__popcll(__ballot(wbusy));
            if (lane == 0) lds_store(&sh.busy[wv], nb);
        }
        if (__ballot(wbusy) != 0ull) {
            did_work = true;
            bool fin = false;
            __builtin_amdgcn_s_setprio(3);  // a walk step is short and waits on memory: let it issue ahead of shading
            // Two steps per trip of this loop, so that the record in hand and the record in flight swap registers
            // without copies: each half consumes one and has the next one's loads under way before its triangle test.
            {
                const uint32_t max_steps = PT_TUNE(P.pool_steps, PT_POOL_STEPS);
                WalkQ qa, qb;
                qa.q0 = qa.q1 = qa.q2 = qa.q3 = qb.q0 = qb.q1 = qb.q2 = qb.q3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (wbusy) walk_fetch(sc, w, qa);
                auto half_step = [&](const WalkQ &have, WalkQ &next, bool last) {
#ifdef PT_DIAG
                    dg_steps++;
                    dg_busy += __popcll(__ballot(wbusy));
                    {
                        const uint32_t nb = (uint32_t)__popcll(__ballot(wbusy));
                        dg_hist[nb <= 8 ? 0 : nb <= 16 ? 1 : nb <= 32 ? 2 : nb <= 48 ? 3 : 4]++;
                    }
#endif
                    if (wbusy) {
                        wsteps++;
                        if (walk_consume<STATS>(sc, wray.o, wray.d, w, have, stack, cn, next, !last)) {
                            const uint32_t p = wpath;
                            sh.st[F_T][p] = __float_as_uint(w.rec.t);
                            sh.st[F_ID][p] = (w.rec.hit ? 0x80000000u : 0u) | (w.rec.is_tri ? 0x40000000u : 0u) | w.rec.index;  // (PT_ID_TIE rides on the index)
                            sh.st[F_U][p] = __float_as_uint(w.rec.u);
                            sh.st[F_V][p] = __float_as_uint(w.rec.v);
                            sh.st[F_COST][p] += wsteps;
                            wbusy = false;
                            fin = true;
                        }
                    }
#ifdef PT_DIAG
                    {
                        const uint32_t ws = wsteps;
                        const bool ended = fin && !wbusy && ws != 0u;
                        const uint32_t b = ws <= 1u ? 0 : ws <= 2u ? 1 : ws <= 4u ? 2 : ws <= 8u ? 3 : ws <= 16u ? 4 : ws <= 32u ? 5 : ws <= 64u ? 6 : 7;
                        for (uint32_t q = 0; q < 8u; q++) {
                            dg_len_n[q] += __popcll(__ballot(ended && b == q));
                            dg_len_steps[q] += wave_sum((ended && b == q) ? ws : 0u);
                        }
                        if (ended) wsteps = 0;  // (counted once)
                    }
#endif
                };
                for (uint32_t k = 0;; k += 2u) {  // (at least one step: no entry test to hoist)
                    bool last = k + 1u >= max_steps;
                    half_step(qa, qb, last);
                    if (__ballot(wbusy) == 0ull || last) break;
                    last = k + 2u >= max_steps;
                    half_step(qb, qa, last);
                    if (__ballot(wbusy) == 0ull || last) break;
                }
            }
            __builtin_amdgcn_s_setprio(0);
            ring_push3(sh.ring, sh.tail, fin && !w.rec.hit, fin && w.rec.hit, false, wpath, lane, lt_mask);
            if (__ballot(fin)) {
                n_miss = lds_load_uniform(&sh.tail[R_MISS]) - lds_load_uniform(&sh.head[R_MISS]);
                if (n_miss > PT_PP_QCAP) n_miss = 0;
                n_hit = lds_load_uniform(&sh.tail[R_HIT]) - lds_load_uniform(&sh.head[R_HIT]);
                if (n_hit > PT_PP_QCAP) n_hit = 0;
            }
        }
#ifdef PT_DIAG
        dg_walk_cyc += clock64() - dg_s;
#endif

        // (c) shading pass over slots of one kind (hits, or misses and slots without a ray, so that the
        // wave skips the code of the other kind): with a wave's worth of them (more for a wave busy
        // walking), or when nothing else is left
        const uint32_t n_busy = (uint32_t)__popcll(__ballot(wbusy));
        const bool pass_hits = n_hit >= n_miss;
        const uint32_t n_ready = pass_hits ? n_hit : n_miss;
        if (n_ready != 0u && (relaxed || n_ready >= (n_busy >= PT_TUNE(P.pool_dense, PT_POOL_DENSE) ? PT_TUNE(P.pool_shade_dense, PT_POOL_SHADE_DENSE) : PT_TUNE(P.refill_lanes, PT_POOL_REFILL_LANES)) || (n_busy == 0u && n_pend == 0u))) {
#ifdef PT_DIAG
            dg_s = clock64();
#endif
            // The parameter block is re-read from the kernel-argument segment stage by stage (each PT_FRESH_PARAMS loads
            // only the words its stage uses, right there), so that no stage's scalars are live during another one.
#ifdef PT_TUNING
            uint32_t miss_iters;
            {
                PT_FRESH_PARAMS(L0);
                miss_iters = L0.pool_miss_iters;
            }
#else
            constexpr uint32_t miss_iters = PT_POOL_MISS_ITERS;
#endif
            bool to_pending = false, to_ready = false, to_hit = false, finished = false;
            uint32_t p = 0xFFFFFFFFu;
            // One body, compiled twice: in a pass over hits every lane has a ray that hit something, in a
            // pass over misses no lane does, so each copy drops the other kind's code.  The order inside is
            // chosen for memory latency: the loads whose values are needed last (the pixel's sums, the sky
            // texel of a miss) are issued first, and the next ray is set up before their values are used.
            auto shading_pass = [&](auto hits_tag) {
                constexpr bool HITS = decltype(hits_tag)::value;
                p = HITS ? ring_pop(sh.ring[R_HIT], &sh.head[R_HIT], &sh.tail[R_HIT], ~0ull, lane, lt_mask, FL::PATHS)
                         : ring_pop(sh.ring[R_MISS], &sh.head[R_MISS], &sh.tail[R_MISS], ~0ull, lane, lt_mask, FL::PATHS);
#ifdef PT_DIAG
                if (!HITS) { const unsigned long long t = clock64(); dg_sec[0] += t - dg_s; dg_s2 = t; }
#endif
                if (p == 0xFFFFFFFFu) return;
                const uint32_t xy = sh.st[F_XY][p], x = xy & 0xFFFFu, y = xy >> 16;
                const uint32_t frame_end = sh.st[F_FEND][p], pidx = sh.st[F_PIDX][p];
                float4 acc;
                uint32_t depth;
                {
                    PT_FRESH_PARAMS(LA);
                    if (pidx >= (DIRECT ? LA.width * LA.height : LA.n_items * 64u)) {
                        pt_mark_error(PT_KERR_SUMS_INDEX);  // the slot is dropped: its waves give up and the host reports it
                        p = 0xFFFFFFFFu;
                        return;
                    }
                    acc = (DIRECT ? LA.accum : LA.partial)[pidx];
                    depth = LA.depth;
                }
                const uint32_t fs = sh.st[F_FS][p];
                const bool had_ray = HITS || (fs & 1u) != 0;
                bool want_aov = (fs & 2u) != 0;
                uint32_t sample = fs >> 3;
                uint32_t frame = sh.st[F_FRAME][p];
                uint32_t bounce = 0, limit = depth;
                v3 color = V3(1.0f, 1.0f, 1.0f), initial_d = V3(0.0f, 0.0f, 0.0f);
                float seed = 0.0f;
                Ray cur;
                cur.o = cur.d = V3(0.0f, 0.0f, 0.0f);
                bool retire = false;                     // the path ends in this pass
                bool rewalk = false;                     // the ray in hand is not shaded: it walks the reference tree first
                v3 path_color = V3(0.0f, 0.0f, 0.0f);    // its colour (a miss: still to be multiplied by the sky texel)
                v3 sky = V3(1.0f, 1.0f, 1.0f);
                Closest c;
                c.hit = c.is_tri = false;
                c.index = 0;
                c.t = c.u = c.v = 0.0f;
                if constexpr (HITS) {
                    cur.o = V3(__uint_as_float(sh.st[F_OX][p]), __uint_as_float(sh.st[F_OY][p]), __uint_as_float(sh.st[F_OZ][p]));
                    cur.d = V3(__uint_as_float(sh.st[F_DX][p]), __uint_as_float(sh.st[F_DY][p]), __uint_as_float(sh.st[F_DZ][p]));
                    const uint32_t hid = sh.st[F_ID][p];
                    c.hit = true;
                    c.is_tri = ((hid >> 30) & 1u) != 0;
                    c.index = hid & PT_ID_INDEX;
                    c.t = __uint_as_float(sh.st[F_T][p]);
                    c.u = __uint_as_float(sh.st[F_U][p]);
                    c.v = __uint_as_float(sh.st[F_V][p]);
                    // A triangle found on the guard tree counts only if the reference walk is certain to return the same one;
                    // otherwise the same ray walks the reference tree (nothing of the path has changed yet).  (A stage of its
                    // own, so that its scalars are gone before the shading's are loaded.)
                    if constexpr (!STATS) {
                        PT_FRESH_PARAMS(LG);
                        if (LG.sc.guard_on != 0u)
                            rewalk = c.is_tri && pt_ray_walks_guard(LG.sc, cur.o, cur.d, fs) && ((hid & PT_ID_TIE) != 0u || !pt_guard_hit_is_certain(LG.sc, cur, c));
                    }
                }
                if (had_ray) {
                    PT_FRESH_PARAMS(L);
                    const PtDeviceScene &sc = L.sc;
                    if constexpr (!HITS) cur.d = V3(__uint_as_float(sh.st[F_DX][p]), __uint_as_float(sh.st[F_DY][p]), __uint_as_float(sh.st[F_DZ][p]));
                    color = V3(__uint_as_float(sh.st[F_COLR][p]), __uint_as_float(sh.st[F_COLG][p]), __uint_as_float(sh.st[F_COLB][p]));
                    if constexpr (!HITS) {
                        // compute.wgsl:140-143
                        if (STATS) cn.sky++;
                        sky = sky_lookup(sc, cur.d);
                        path_color = color;
                        retire = true;
                        if (want_aov) {
                            L.aov_prim[(size_t)y * L.width + x] = 0xFFFFFFFFu;
                            L.aov_t[(size_t)y * L.width + x] = 0.0f;
                            want_aov = false;
                        }
                    } else {
                        seed = __uint_as_float(sh.st[F_SEED][p]);
                        bounce = sh.st[F_BOUNCE][p];
                        if constexpr (GIZMO) {
                            initial_d = V3(__uint_as_float(sh.st[FL::IDIR][p]), __uint_as_float(sh.st[FL::IDIR + 1][p]), __uint_as_float(sh.st[FL::IDIR + 2][p]));
                            limit = sh.st[FL::LIMIT][p];
                        }
                        if (!rewalk) {
                            uint32_t prim;
                            v3 result = V3(0.0f, 0.0f, 0.0f);
                            const bool path_done = shade_ray<STATS, GIZMO>(sc, cur, c, color, seed, bounce, limit, initial_d, result, prim, cn);
                            if (want_aov) {
                                L.aov_prim[(size_t)y * L.width + x] = prim;
                                L.aov_t[(size_t)y * L.width + x] = c.t;
                                want_aov = false;
                            }
                            // rayColor returns: the path's colour, or the throughput when the bounces run out (compute.wgsl:121)
                            retire = path_done || bounce >= limit;
                            path_color = path_done ? result : color;
                        }
                    }
                }
#ifdef PT_DIAG
                if (!HITS) { const unsigned long long t = clock64(); dg_sec[1] += t - dg_s2; dg_s2 = t; }
#endif
                // frame and sample counters of the retiring path (compute.wgsl:112-128); its colour is added below
                bool frame_ends = false;
                if (retire) {
                    if constexpr (!SPP1) sample++;
                    bool last = true;
                    if constexpr (!SPP1) {
                        PT_FRESH_PARAMS(L);
                        last = sample == L.spp;
                    }
                    if (last) {
                        frame_ends = true;
                        sample = 0;
                        frame++;
                    }
                    finished = frame == frame_end;
                }
                const uint32_t cost = sh.st[F_COST][p] + 8u;  // a shading pass weighs like eight node visits
                if (finished) {
                    PT_FRESH_PARAMS(L);
                    if (L.cost && pt_cost_sample(x, y, L.cost_key >> 16)) atomicAdd(&L.cost[(y / PT_TILE) * (L.cost_key & 0xFFFFu) + x / PT_TILE], cost);
                } else if (HITS && retire) {
                    // The path is over and the job goes on: the pixel's next path is started by a pass over misses (a slot
                    // without a ray is one of theirs), where every lane starts a path, instead of here, where the few lanes
                    // that retire would have the whole wave run the RNG and the camera ray.
                    sh.st[F_FRAME][p] = frame;
                    sh.st[F_FS][p] = sample << 3;
                    sh.st[F_COST][p] = cost;
                    to_ready = true;
                } else {
                    if constexpr (!HITS) if (!had_ray || retire) {
                        // the pixel's next path (compute.wgsl:113-119, 131-136); depth >= 1, so it gets a ray
                        PT_FRESH_PARAMS(L);
                        uint32_t sx, sy, sz, sw;
                        const uint32_t time = L.time0 + frame * L.stride;  // compute.wgsl:85
                        sx = x; sy = y; sz = x + time; sw = y + time;
                        if constexpr (!SPP1) {
                            if (sample != 0) { sx = sh.st[FL::RNG][p]; sy = sh.st[FL::RNG + 1][p]; sz = sh.st[FL::RNG + 2][p]; sw = sh.st[FL::RNG + 3][p]; }
                        }
                        const float px = -0.5f + pt_hybrid_taus(sx, sy, sz, sw);
                        const float py = -0.5f + pt_hybrid_taus(sx, sy, sz, sw);
                        const v3 smp = v3_fma(py, L.fr.dv, L.fr.du * px);
                        const v3 ploc = v3_fma((float)y, L.fr.dv, v3_fma((float)x, L.fr.du, L.fr.pixel00));  // compute.wgsl:109
                        cur.o = L.fr.origin;
                        cur.d = (ploc + smp) - L.fr.origin;
                        if constexpr (GIZMO) initial_d = cur.d;
                        color = V3(1.0f, 1.0f, 1.0f);
                        seed = pt_hybrid_taus(sx, sy, sz, sw);
                        if constexpr (!SPP1) { sh.st[FL::RNG][p] = sx; sh.st[FL::RNG + 1][p] = sy; sh.st[FL::RNG + 2][p] = sz; sh.st[FL::RNG + 3][p] = sw; }
                        bounce = 0;
                        limit = L.depth;
                        want_aov = (frame == 0 && sample == 0 && L.aov_prim != nullptr);
                    }
#ifdef PT_DIAG
                    if (!HITS) { const unsigned long long t = clock64(); dg_sec[2] += t - dg_s2; dg_s2 = t; }
#endif
                    if (rewalk) cn.rewalks++;
                    else cn.rays++;
                    // spheres + root visit; a surviving ray goes to PENDING with its first node(s)
                    Walk nw;
                    uint32_t k2 = 0xFFFFFFFFu;
                    bool need_walk;
                    {
                        PT_FRESH_PARAMS(L);
                        need_walk = walk_begin_pool<STATS>(L.sc, cur, nw, k2, cn, !rewalk && bounce == 0u && pt_camera_ray_misses_bvh(L.cull, x, y));
                    }
#ifdef PT_DIAG
                    if (!HITS) { const unsigned long long t = clock64(); dg_sec[3] += t - dg_s2; dg_s2 = t; }
#endif
                    sh.st[F_OX][p] = __float_as_uint(cur.o.x); sh.st[F_OY][p] = __float_as_uint(cur.o.y); sh.st[F_OZ][p] = __float_as_uint(cur.o.z);
                    sh.st[F_DX][p] = __float_as_uint(cur.d.x); sh.st[F_DY][p] = __float_as_uint(cur.d.y); sh.st[F_DZ][p] = __float_as_uint(cur.d.z);
                    sh.st[F_T][p] = __float_as_uint(nw.rec.t);
                    sh.st[F_ID][p] = (nw.rec.hit ? 0x80000000u : 0u) | nw.rec.index;
                    sh.st[F_U][p] = need_walk ? nw.node : 0u;
                    sh.st[F_V][p] = need_walk ? k2 : 0u;
                    sh.st[F_COLR][p] = __float_as_uint(color.x); sh.st[F_COLG][p] = __float_as_uint(color.y); sh.st[F_COLB][p] = __float_as_uint(color.z);
                    sh.st[F_SEED][p] = __float_as_uint(seed);
                    sh.st[F_BOUNCE][p] = bounce;
                    sh.st[F_FRAME][p] = frame;
                    sh.st[F_FS][p] = 1u | (want_aov ? 2u : 0u) | (rewalk ? 4u : 0u) | (sample << 3);
                    sh.st[F_COST][p] = cost;
                    if constexpr (GIZMO) {
                        sh.st[FL::IDIR][p] = __float_as_uint(initial_d.x); sh.st[FL::IDIR + 1][p] = __float_as_uint(initial_d.y); sh.st[FL::IDIR + 2][p] = __float_as_uint(initial_d.z);
                        sh.st[FL::LIMIT][p] = limit;
                    }
                    to_pending = need_walk;
                    to_ready = !need_walk && !nw.rec.hit;
                    to_hit = !need_walk && nw.rec.hit;
                }
#ifdef PT_DIAG
                if (!HITS) { const unsigned long long t = clock64(); dg_sec[4] += t - dg_s2; dg_s2 = t; }
#endif
                // the retired path's colour: into the frame's sample sum, and at the frame's end into the pixel's sums
                if (retire) {
                    if constexpr (!HITS) path_color = path_color * sky;
                    if constexpr (!SPP1)
                        path_color = V3(__uint_as_float(sh.st[FL::FSUM][p]), __uint_as_float(sh.st[FL::FSUM + 1][p]), __uint_as_float(sh.st[FL::FSUM + 2][p])) + path_color;
                    if (frame_ends) {
                        // compute.wgsl:124-127, accumulated instead of stored as rgba8unorm
                        PT_FRESH_PARAMS(L);
                        const v3 mean = path_color * L.inv_spp;
                        acc.x += mean.x; acc.y += mean.y; acc.z += mean.z; acc.w += 1.0f;
                        (DIRECT ? L.accum : L.partial)[pidx] = acc;
                        path_color = V3(0.0f, 0.0f, 0.0f);
                    }
                    if constexpr (!SPP1) {
                        sh.st[FL::FSUM][p] = __float_as_uint(path_color.x);
                        sh.st[FL::FSUM + 1][p] = __float_as_uint(path_color.y);
                        sh.st[FL::FSUM + 2][p] = __float_as_uint(path_color.z);
                    }
                }
            };
            // A pass over misses keeps going in registers: a pixel whose ray missed everything usually sends
            // its next primary ray into the sky too, so the lane shades that one as well instead of sending the
            // slot through LDS and the ring again; lanes leave the loop when their new ray hits something or
            // needs a walk (state to LDS, slot to the matching ring) or when their job ends, and the loop ends
            // when too few lanes are left or after pool_miss_iters rounds.
            auto miss_pass = [&]() {
                p = ring_pop(sh.ring[R_MISS], &sh.head[R_MISS], &sh.tail[R_MISS], ~0ull, lane, lt_mask, FL::PATHS);
#ifdef PT_DIAG
                { const unsigned long long t = clock64(); dg_sec[0] += t - dg_s; dg_s2 = t; }
#endif
                bool mine = p != 0xFFFFFFFFu;  // this lane still carries its slot through the loop
                uint32_t x = 0, y = 0, frame_end = 0, pidx = 0, frame = 0, sample = 0, cost = 0;
                bool have_ray = false, want_aov = false, dirty = false;
                float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                v3 d = V3(0.0f, 0.0f, 0.0f), color = V3(1.0f, 1.0f, 1.0f), fsum = V3(0.0f, 0.0f, 0.0f);
                uint32_t rx = 0, ry = 0, rz = 0, rw = 0;  // HybridTaus state between the samples of a frame (spp > 1)
                if (mine) {
                    const uint32_t xy = sh.st[F_XY][p];
                    x = xy & 0xFFFFu;
                    y = xy >> 16;
                    frame_end = sh.st[F_FEND][p];
                    pidx = sh.st[F_PIDX][p];
                    {
                        PT_FRESH_PARAMS(LA);
                        if (pidx >= (DIRECT ? LA.width * LA.height : LA.n_items * 64u)) {
                            pt_mark_error(PT_KERR_SUMS_INDEX);  // the slot is dropped: its waves give up and the host reports it
                            pidx = 0;
                            mine = false;
                            p = 0xFFFFFFFFu;
                        } else {
                            acc = (DIRECT ? LA.accum : LA.partial)[pidx];
                        }
                    }
                    const uint32_t fs = sh.st[F_FS][p];
                    have_ray = (fs & 1u) != 0;
                    want_aov = (fs & 2u) != 0;
                    sample = fs >> 3;
                    frame = sh.st[F_FRAME][p];
                    cost = sh.st[F_COST][p];
                    if (have_ray) {
                        d = V3(__uint_as_float(sh.st[F_DX][p]), __uint_as_float(sh.st[F_DY][p]), __uint_as_float(sh.st[F_DZ][p]));
                        color = V3(__uint_as_float(sh.st[F_COLR][p]), __uint_as_float(sh.st[F_COLG][p]), __uint_as_float(sh.st[F_COLB][p]));
                    }
                    if constexpr (!SPP1) {
                        fsum = V3(__uint_as_float(sh.st[FL::FSUM][p]), __uint_as_float(sh.st[FL::FSUM + 1][p]), __uint_as_float(sh.st[FL::FSUM + 2][p]));
                        rx = sh.st[FL::RNG][p]; ry = sh.st[FL::RNG + 1][p]; rz = sh.st[FL::RNG + 2][p]; rw = sh.st[FL::RNG + 3][p];  // (unused by a frame's first sample)
                    }
                }
#ifdef PT_DIAG
                { const unsigned long long t = clock64(); dg_sec[1] += t - dg_s2; dg_s2 = t; }
#endif
                for (uint32_t round = 0;; round++) {
#ifdef PT_DIAG
                    dg_rounds++;
                    dg_round_lanes += __popcll(__ballot(mine));
#endif
                    PT_FRESH_PARAMS(L);
                    const PtDeviceScene &sc = L.sc;
                    if (mine) {
                        // the ray in hand missed everything: compute.wgsl:140-143
                        v3 sky = V3(1.0f, 1.0f, 1.0f), path_color = color;
                        const bool retire = have_ray;
                        bool frame_ends = false;
                        if (retire) {
                            if (STATS) cn.sky++;
                            sky = sky_lookup(sc, d);  // issued here, consumed after the next ray is set up
                            if (want_aov) {
                                L.aov_prim[(size_t)y * L.width + x] = 0xFFFFFFFFu;
                                L.aov_t[(size_t)y * L.width + x] = 0.0f;
                                want_aov = false;
                            }
                            // frame and sample counters of the retiring path (compute.wgsl:112-128)
                            if constexpr (!SPP1) sample++;
                            if (SPP1 || sample == L.spp) {
                                frame_ends = true;
                                sample = 0;
                                frame++;
                            }
                        }
                        cost += 8u;  // a shading round weighs like eight node visits
                        // the retired path's colour: into the frame's sample sum, and at the frame's end into the pixel's sums
                        // (run once the next ray is set up, so that the texel has had time to arrive, and before a sphere hit of
                        // that ray is shaded, so that the texel and the colour do not hold registers through the shading)
                        auto add_retired = [&]() {
                            if (retire) {
                                path_color = path_color * sky;
                                if constexpr (!SPP1) path_color = fsum + path_color;
                                if (frame_ends) {
                                    // compute.wgsl:124-127, accumulated instead of stored as rgba8unorm
                                    const v3 mean = path_color * L.inv_spp;
                                    acc.x += mean.x; acc.y += mean.y; acc.z += mean.z; acc.w += 1.0f;
                                    dirty = true;
                                    path_color = V3(0.0f, 0.0f, 0.0f);
                                }
                                if constexpr (!SPP1) fsum = path_color;
                            }
                        };
                        if (frame == frame_end) {
                            finished = true;
                            mine = false;
                            if (L.cost && pt_cost_sample(x, y, L.cost_key >> 16)) atomicAdd(&L.cost[(y / PT_TILE) * (L.cost_key & 0xFFFFu) + x / PT_TILE], cost);
                            add_retired();
                        } else {
                            // the pixel's next path (compute.wgsl:113-119, 131-136); depth >= 1, so it gets a ray
                            uint32_t sx = x, sy = y, sz = x + (L.time0 + frame * L.stride), sw = y + (L.time0 + frame * L.stride);  // compute.wgsl:85
                            if constexpr (!SPP1) {
                                if (sample != 0) { sx = rx; sy = ry; sz = rz; sw = rw; }
                            }
                            const float px = -0.5f + pt_hybrid_taus(sx, sy, sz, sw);
                            const float py = -0.5f + pt_hybrid_taus(sx, sy, sz, sw);
                            const v3 smp = v3_fma(py, L.fr.dv, L.fr.du * px);
                            const v3 ploc = v3_fma((float)y, L.fr.dv, v3_fma((float)x, L.fr.du, L.fr.pixel00));  // compute.wgsl:109
                            Ray cur;
                            cur.o = L.fr.origin;
                            cur.d = (ploc + smp) - L.fr.origin;
                            const float seed = pt_hybrid_taus(sx, sy, sz, sw);
                            if constexpr (!SPP1) { rx = sx; ry = sy; rz = sz; rw = sw; }
                            want_aov = (frame == 0 && sample == 0 && L.aov_prim != nullptr);
                            cn.rays++;
                            Walk nw;
                            uint32_t k2 = 0xFFFFFFFFu;
                            bool need_walk;
#ifdef PT_DIAG
                            { const unsigned long long t = clock64(); dg_sec[2] += t - dg_s2; dg_s2 = t; }
#endif
                            {
                                PT_FRESH_PARAMS(LW);
                                need_walk = walk_begin_pool<STATS>(LW.sc, cur, nw, k2, cn, pt_camera_ray_misses_bvh(LW.cull, x, y));  // (every ray of this loop is a camera ray)
                            }
#ifdef PT_DIAG
                            { const unsigned long long t = clock64(); dg_sec[3] += t - dg_s2; dg_s2 = t; }
#endif
                            add_retired();
                            // A primary ray that hits a sphere without needing a walk (the ground of an outdoor scene) is shaded
                            // right here, and its bounce ray gets its spheres + root visit too: if that one leaves the scene as
                            // well, the path never touches LDS or a ring, and the bounce ray is the miss this lane carries into
                            // the next round.  Tried on copies and kept only if the path goes on with a bounce ray (one that ends
                            // at the hit -- absorbed, out of bounces, a gizmo ring -- goes to a pass over hits like every other hit).
                            v3 lcol = V3(1.0f, 1.0f, 1.0f), lidir = cur.d;
                            uint32_t lbounce = 0, llimit = 0;
                            if constexpr (GIZMO) llimit = L.depth;
                            if constexpr (PT_INLINE_SPHERE_HITS && !STATS) {
                                if (!need_walk && nw.rec.hit) {
                                    PT_FRESH_PARAMS(LS);
                                    Ray br = cur;
                                    v3 bcol = lcol, result = V3(0.0f, 0.0f, 0.0f);
                                    uint32_t bb = 0, blimit = LS.depth, prim = 0;
                                    const bool path_done = shade_ray<STATS, GIZMO>(LS.sc, br, nw.rec, bcol, seed, bb, blimit, cur.d, result, prim, cn);
                                    if (!path_done && bb < blimit) {
                                        if (want_aov) {
                                            LS.aov_prim[(size_t)y * LS.width + x] = prim;
                                            LS.aov_t[(size_t)y * LS.width + x] = nw.rec.t;
                                            want_aov = false;
                                        }
                                        cur = br;
                                        lcol = bcol;
                                        lbounce = bb;
                                        if constexpr (GIZMO) llimit = blimit;
                                        cn.rays++;
                                        k2 = 0xFFFFFFFFu;
                                        PT_FRESH_PARAMS(LW);
                                        need_walk = walk_begin_pool<STATS>(LW.sc, cur, nw, k2, cn, false);
                                    }
                                }
                            }
                            if (need_walk || nw.rec.hit) {
                                // this ray goes on elsewhere: the slot's whole state to LDS
                                sh.st[F_OX][p] = __float_as_uint(cur.o.x); sh.st[F_OY][p] = __float_as_uint(cur.o.y); sh.st[F_OZ][p] = __float_as_uint(cur.o.z);
                                sh.st[F_DX][p] = __float_as_uint(cur.d.x); sh.st[F_DY][p] = __float_as_uint(cur.d.y); sh.st[F_DZ][p] = __float_as_uint(cur.d.z);
                                sh.st[F_T][p] = __float_as_uint(nw.rec.t);
                                sh.st[F_ID][p] = (nw.rec.hit ? 0x80000000u : 0u) | nw.rec.index;
                                sh.st[F_U][p] = need_walk ? nw.node : 0u;
                                sh.st[F_V][p] = need_walk ? k2 : 0u;
                                sh.st[F_COLR][p] = __float_as_uint(lcol.x); sh.st[F_COLG][p] = __float_as_uint(lcol.y); sh.st[F_COLB][p] = __float_as_uint(lcol.z);
                                sh.st[F_SEED][p] = __float_as_uint(seed);
                                sh.st[F_BOUNCE][p] = lbounce;
                                sh.st[F_FRAME][p] = frame;
                                sh.st[F_FS][p] = 1u | (want_aov ? 2u : 0u) | (sample << 3);
                                sh.st[F_COST][p] = cost;
                                if constexpr (GIZMO) {
                                    sh.st[FL::IDIR][p] = __float_as_uint(lidir.x); sh.st[FL::IDIR + 1][p] = __float_as_uint(lidir.y); sh.st[FL::IDIR + 2][p] = __float_as_uint(lidir.z);
                                    sh.st[FL::LIMIT][p] = llimit;
                                }
                                if constexpr (!SPP1) { sh.st[FL::RNG][p] = rx; sh.st[FL::RNG + 1][p] = ry; sh.st[FL::RNG + 2][p] = rz; sh.st[FL::RNG + 3][p] = rw; }
                                to_pending = need_walk;
                                to_hit = !need_walk;
                                mine = false;
                            } else {
                                // a miss (of the primary ray, or of the bounce ray off a sphere): stays in this lane's registers
                                d = cur.d;
                                color = lcol;
                                have_ray = true;
                            }
                        }
#ifdef PT_DIAG
                        { const unsigned long long t = clock64(); dg_sec[4] += t - dg_s2; dg_s2 = t; }
#endif
                    }
#ifdef PT_DIAG
                    { asm volatile("s_waitcnt vmcnt(0)"); const unsigned long long t = clock64(); dg_sec[5] += t - dg_s2; dg_s2 = t; }
#endif
                    const uint32_t n_mine = (uint32_t)__popcll(__ballot(mine));
                    if (n_mine < PT_TUNE(L.pool_miss_lanes, PT_POOL_MISS_LANES) || round + 1u >= PT_TUNE(L.pool_miss_iters, PT_POOL_MISS_ITERS)) break;
                }
                if (p != 0xFFFFFFFFu) {
                    if (dirty) {
                        PT_FRESH_PARAMS(LA);
                        (DIRECT ? LA.accum : LA.partial)[pidx] = acc;
                    }
                    if constexpr (!SPP1) {
                        // (a lane that left with a hit or a walk stored its RNG state above; the sample sum is common)
                        sh.st[FL::FSUM][p] = __float_as_uint(fsum.x); sh.st[FL::FSUM + 1][p] = __float_as_uint(fsum.y); sh.st[FL::FSUM + 2][p] = __float_as_uint(fsum.z);
                    }
                    if (mine) {
                        // still holding a missed ray when the loop ended: what the next pass over misses reads
                        sh.st[F_DX][p] = __float_as_uint(d.x); sh.st[F_DY][p] = __float_as_uint(d.y); sh.st[F_DZ][p] = __float_as_uint(d.z);
                        sh.st[F_COLR][p] = __float_as_uint(color.x); sh.st[F_COLG][p] = __float_as_uint(color.y); sh.st[F_COLB][p] = __float_as_uint(color.z);
                        sh.st[F_FRAME][p] = frame;
                        sh.st[F_FS][p] = (have_ray ? 1u : 0u) | (want_aov ? 2u : 0u) | (sample << 3);
                        sh.st[F_COST][p] = cost;
                        if constexpr (!SPP1) { sh.st[FL::RNG][p] = rx; sh.st[FL::RNG + 1][p] = ry; sh.st[FL::RNG + 2][p] = rz; sh.st[FL::RNG + 3][p] = rw; }
                        to_ready = true;
                    }
                }
            };
            if (pass_hits) shading_pass(std::true_type{});
            else if (miss_iters > 1u) miss_pass();
            else shading_pass(std::false_type{});
#ifdef PT_DIAG
            if (!pass_hits) { const unsigned long long t = clock64(); dg_sec[6] += t - dg_s2; dg_s2 = t; }
#endif
            const uint32_t id = p;
            // slots whose job is done take the next one (or die when the launch has no items left)
#ifdef PT_DIAG
            const unsigned long long dg_j0 = clock64();
#endif
            if (__ballot(finished) != 0ull) {
                const bool got = assign_jobs(finished, p);
                to_ready = to_ready || (finished && got);
                const uint32_t n_dead = (uint32_t)__popcll(__ballot(finished && !got));
                if (n_dead && lane == 0) atomicSub(&sh.live, n_dead);
            }
#ifdef PT_DIAG
            const unsigned long long dg_j1 = clock64();
            if (!pass_hits) { dg_extra[0] += dg_j1 - dg_j0; dg_extra[2] += __popcll(__ballot(finished)); }
#endif
            ring_push3(sh.ring, sh.tail, to_ready, to_hit, to_pending, p, lane, lt_mask);
#ifdef PT_DIAG
            if (!pass_hits) dg_extra[1] += clock64() - dg_j1;
#endif
            const unsigned long long gotm = __ballot(id != 0xFFFFFFFFu);
            if (gotm) did_work = true;
#ifdef PT_DIAG
            if (!pass_hits) dg_sec[6] += clock64() - dg_s2;

            dg_shades++;
            dg_ready += __popcll(gotm);
            dg_shade_cyc += clock64() - dg_s;
            if (pass_hits) { dg_hit_passes++; dg_hit_lanes += __popcll(gotm); dg_hit_cyc += clock64() - dg_s; }
#endif
        }

        if (lds_load_uniform(&sh.live) == 0u) break;
        if (did_work) {
            idle = 0;
        } else {
            idle++;
#ifdef PT_DIAG
            dg_idle++;
#endif
            __builtin_amdgcn_s_sleep(2);
        }
    }

#ifdef PT_DIAG
    {
        const uint32_t wid = blockIdx.x * PT_PP_WAVES + wv;
        if (lane == 0 && wid < PT_DIAG_MAX_WAVES) {
            g_diag[8 * wid + 0] = diag_t0;
            g_diag[8 * wid + 1] = __builtin_amdgcn_s_memrealtime();
            g_diag[8 * wid + 2] = dg_trips | (dg_idle << 32);
            g_diag_marks[(wid & 63u) * 8 + 0] = dg_hit_passes;  // a sample of 64 waves is enough for these
            g_diag_marks[(wid & 63u) * 8 + 1] = dg_hit_lanes;
            g_diag_marks[(wid & 63u) * 8 + 2] = dg_hit_cyc;
            g_diag_marks[(wid & 63u) * 8 + 3] = dg_shades;
            g_diag_marks[(wid & 63u) * 8 + 4] = dg_ready;
            g_diag_marks[(wid & 63u) * 8 + 5] = dg_shade_cyc;
            g_diag_marks[(wid & 63u) * 8 + 6] = dg_rounds;
            g_diag_marks[(wid & 63u) * 8 + 7] = dg_round_lanes;
            for (int k = 0; k < 7; k++) g_diag_marks[512 + (wid & 63u) * 8 + k] = dg_sec[k];
            for (int k = 0; k < 5; k++) g_diag_marks[1024 + (wid & 63u) * 8 + k] = dg_hist[k];
            for (int k = 0; k < 3; k++) g_diag_marks[1024 + (wid & 63u) * 8 + 5 + k] = dg_extra[k];
            for (int k = 0; k < 8; k++) { g_diag_marks[1536 + (wid & 63u) * 8 + k] = dg_len_n[k]; g_diag_marks[2048 + (wid & 63u) * 8 + k] = dg_len_steps[k]; }
            g_diag[8 * wid + 3] = dg_shades | (dg_ready << 32);
            g_diag[8 * wid + 4] = dg_shade_cyc;
            g_diag[8 * wid + 5] = dg_steps | (dg_busy << 32);
            g_diag[8 * wid + 6] = dg_walk_cyc;
            g_diag[8 * wid + 7] = dg_claim_cyc | (dg_claims << 40);
        }
    }
#endif
    // a wave that gave up (should never happen) leaves a mark for the host instead of hanging the GPU
    if ((__ballot(wbusy) != 0ull || lds_load(&sh.live) != 0u) && lane == 0) atomicAdd(P.next_item + 1, 1u);

    PtTraceParams LE;  // (re-read: the counters' address is not kept in registers through the loop)
    pt_fresh_params(LE);
    const uint32_t stripe = lds_load_uniform(&sh.group) % PT_COUNTER_STRIPES;
    pt_add_counters<STATS>(&LE.counters[stripe * PT_COUNTER_PITCH], cn, lane);
}

// ---- explicit ray queries ---------------------------------------------------------------
template <bool OVERFLOW>
__global__ __launch_bounds__(PT_BLOCK) void ray_query_kernel(const PtRayQuery Q)
{
    const uint32_t tid = threadIdx.x;
    const size_t i = (size_t)blockIdx.x * PT_BLOCK + tid;
    __shared__ uint32_t smem[PT_LDS_STACK * PT_BLOCK];
    Stack<OVERFLOW, PT_BLOCK> stack{&smem[tid], nullptr, 0};
    if (OVERFLOW) {
        stack.spill = Q.gstack + i;
        stack.pitch = (size_t)gridDim.x * PT_BLOCK;
    }
    if (i >= Q.n) return;
    Ray r;
    r.o = V3(Q.origins[3 * i], Q.origins[3 * i + 1], Q.origins[3 * i + 2]);
    r.d = V3(Q.dirs[3 * i], Q.dirs[3 * i + 1], Q.dirs[3 * i + 2]);
    Counters cn;
#ifdef PT_TUNING
    // (tuning builds: the ray's traversal steps ride in the upper bits of `front`, for tools/step_latency.py, and the walk
    // is the reference tree's unless `walked` is asked for: tools/guard_steps.py)
    constexpr bool count_steps = true;
    const bool try_guard = Q.walked != nullptr;
#else
    constexpr bool count_steps = false;
    const bool try_guard = true;
#endif
    // Where the mesh has a guard tree, the pooled kernel's scheme, ray by ray: a ray in reach walks the guard tree, a
    // triangle found there counts if it is certain to be the reference walk's as well, and everything else walks the
    // reference tree.  (This is also how that scheme is tested on rays aimed at edges, vertices and grazing angles.)
    Closest c;
    bool settled = false;
    if (try_guard && Q.sc.guard_quads != 0u && pt_guard_in_reach(Q.sc, r.o, r.d)) {
        WalkU w;
        w.rec = hit_spheres<false>(Q.sc, r.o, r.d, cn);
        w.nearest = w.rec.hit ? pt_min(9999.0f, w.rec.t) : 9999.0f;
        w.node = Q.sc.guard_quads;
        w.sp = 0;
        w.leaf_left = 0;
        w.top = 0;
        w.inv = pt_rcp3(r.d);
        [[maybe_unused]] uint32_t trips = 1;  // (reported by tuning builds)
        while (!walk_step_u<false>(Q.sc, r, w, stack, cn, OVERFLOW ? Q.gstack_cap : PT_LDS_STACK)) trips++;
        c = w.rec;
        settled = (c.index & PT_ID_ABANDONED) == 0u && (!(c.hit && c.is_tri) || ((c.index & PT_ID_TIE) == 0u && pt_guard_hit_is_certain(Q.sc, r, c)));
#ifdef PT_TUNING
        if (Q.walked) Q.walked[i] = (settled ? 1u : 2u) | (trips << 2);  // (tuning builds: the guard walk's steps ride along)
#else
        if (Q.walked) Q.walked[i] = settled ? 1u : 2u;
#endif
    } else if (Q.walked) {
        Q.walked[i] = 0u;
    }
    if (!settled) c = hit_scene<count_steps>(Q.sc, r, stack, cn);
    const HitRecord h = make_record(Q.sc, r, c);
    if (Q.prim) Q.prim[i] = h.prim;
    if (Q.t) Q.t[i] = h.t;
    if (Q.p) { Q.p[3 * i] = h.p.x; Q.p[3 * i + 1] = h.p.y; Q.p[3 * i + 2] = h.p.z; }
    if (Q.normal) { Q.normal[3 * i] = h.normal.x; Q.normal[3 * i + 1] = h.normal.y; Q.normal[3 * i + 2] = h.normal.z; }
#ifdef PT_TUNING
    if (Q.front) Q.front[i] = (h.front ? 1u : 0u) | ((cn.interior + cn.tri_tests) << 1);
#else
    if (Q.front) Q.front[i] = h.front ? 1u : 0u;
#endif
}

// ---- tiles that see nothing but the sky -----------------------------------------------------
// In an outdoor scene most pixels' camera rays leave the scene at once.  A tile whose 64 pixels can be PROVEN to do so
// for every jitter is taken out of the megakernels' work (no items, no path slots, no rings): its frames are summed by
// the plain loop of sky_tiles_kernel, straight onto the running sums in frame order -- the order of the shader's own
// frame-after-frame accumulation.  The proof is conservative and is made per pixel when a launch is planned:
//   * mesh: the pixel lies outside the screen rectangles of the root's child boxes (PtTraceParams::cull; none if there
//     is no mesh);
//   * every sphere: with theta the angle between the pixel's centre ray and the direction to the sphere's centre, alpha
//     a bound on the angle between the centre ray and any jittered ray of the pixel, L the distance of the centre and
//     r the radius (camera outside: L > r), either every ray points away from the centre (theta - alpha > pi/2: hitSphere,
//     compute.wgsl:304-350, then has b > 0 and c > 0, so sqrt(b^2 - 4ac) <= b in float arithmetic as well and both roots
//     are <= 0 <= t_min), or every ray's line passes the sphere (sin^2 > (r/L)^2 over [theta - alpha, theta + alpha]:
//     discriminant < 0).  Double arithmetic here against the kernels' float: the margins (1e-3 rad, 1e-5 in sin^2, the
//     rounding of the float direction added to alpha) are more than ten times the float evaluation's error bounds.
__device__ bool pt_pixel_sees_only_sky(const PtTraceParams &P, uint32_t x, uint32_t y)
{
    const PtDeviceScene &sc = P.sc;
    if (sc.n_nodes != 0u && !pt_camera_ray_misses_bvh(P.cull, x, y)) return false;
    if (sc.n_spheres == 0u) return true;
    if (!(sc.t_min >= 0.0f)) return false;
    const double o[3] = {P.fr.origin.x, P.fr.origin.y, P.fr.origin.z};
    const double du[3] = {P.fr.du.x, P.fr.du.y, P.fr.du.z}, dv[3] = {P.fr.dv.x, P.fr.dv.y, P.fr.dv.z};
    const double p0[3] = {P.fr.pixel00.x, P.fr.pixel00.y, P.fr.pixel00.z};
    double dc[3];
    for (int k = 0; k < 3; k++) dc[k] = (p0[k] + (double)x * du[k] + (double)y * dv[k]) - o[k];  // compute.wgsl:109-119 without the jitter
    auto len = [](const double *v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    const double ldc = len(dc), ldu = len(du), ldv = len(dv);
    // the jitter moves the direction by at most half of du plus half of dv; the float evaluation of the direction is
    // off by a few ulps of its largest intermediate
    const double rho = 0.525 * (ldu + ldv) + 1e-6 * (len(p0) + (double)P.width * ldu + (double)P.height * ldv + len(o));
    if (!(rho < 0.05 * ldc)) return false;
    const double alpha = 1.1 * rho / ldc;  // asin(s) <= 1.1 s up to s = 0.05
    const double pi = 3.14159265358979323846;
    for (uint32_t i = 0; i < sc.n_spheres; i++) {
        const float4 s0 = sc.spheres[3 * i];
        const double r = fabs((double)s0.w);
        const double oc[3] = {o[0] - (double)s0.x, o[1] - (double)s0.y, o[2] - (double)s0.z};
        const double L = len(oc);
        if (!(L > r * (1.0 + 1e-4)) || !(L > 0.0)) return false;  // camera inside or on the sphere (or not a number)
        double c = -(oc[0] * dc[0] + oc[1] * dc[1] + oc[2] * dc[2]) / (L * ldc);
        if (!(c == c)) return false;
        c = fmin(fmax(c, -1.0), 1.0);
        const double theta = acos(c), lo = theta - alpha, hi = theta + alpha;
        const bool away = lo > 0.5 * pi + 1e-3;
        const double sn = fmin(sin(lo), sin(hi)), q = r / L;
        const bool clear = lo > 0.0 && hi < pi && sn * sn > q * q + 1e-5;
        if (!(away || clear)) return false;
    }
    return true;
}

// One wave per tile: tile_sky[tile] = 1 if every pixel of the tile that is rendered sees only the sky.
__global__ __launch_bounds__(256) void classify_tiles_kernel(const PtTraceParams P, uint32_t *tile_sky)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile >= P.n_tiles) return;
    const uint32_t tby = tile / P.tiles_x, tbx = tile - tby * P.tiles_x;
    const uint32_t x = tbx * PT_TILE + (lane & 7u), y = tby * PT_TILE + (lane >> 3);
    bool sky = true;
    if (x < P.cols && y < P.rows) sky = pt_pixel_sees_only_sky(P, x, y);
    const bool all = __ballot(!sky) == 0ull;
    if (lane == 0) tile_sky[tile] = all ? 1u : 0u;
}

// One wave per flagged tile: every sample's camera ray (compute.wgsl:85, 109-119) goes straight to the
// sky lookup (compute.wgsl:140-143) and onto the pixel's running sums (compute.wgsl:124-127), frame after frame.  The
// arithmetic is the megakernels' own (advance + shade_ray for a ray that hit nothing).
// (32 registers: what four waves of the pooled kernel leave of a SIMD's 512, so that one wave of this loop fits beside them)
template <bool SPP1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) void sky_tiles_kernel(const PtTraceParams P, const uint32_t *tile_sky)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile >= P.n_tiles || tile_sky[tile] == 0u) return;  // (wave-uniform)
    const uint32_t tby = tile / P.tiles_x, tbx = tile - tby * P.tiles_x;
    const uint32_t x = tbx * PT_TILE + (lane & 7u), y = tby * PT_TILE + (lane >> 3);
    const bool in_image = (x < P.cols) && (y < P.rows);
    if (in_image) {
        const size_t pix = (size_t)y * P.width + x;
        float4 acc = P.accum[pix];
        const v3 ploc = v3_fma((float)y, P.fr.dv, v3_fma((float)x, P.fr.du, P.fr.pixel00));  // compute.wgsl:109
        for (uint32_t frame = 0; frame < P.n_frames; frame++) {
            const uint32_t time = P.time0 + frame * P.stride;
            uint32_t sx = x, sy = y, sz = x + time, sw = y + time;  // compute.wgsl:85
            v3 frame_sum = V3(0.0f, 0.0f, 0.0f);
            for (uint32_t sample = 0; sample < (SPP1 ? 1u : P.spp); sample++) {  // compute.wgsl:112-122; one RNG stream per pixel and frame
                const float px = -0.5f + pt_hybrid_taus(sx, sy, sz, sw);
                const float py = -0.5f + pt_hybrid_taus(sx, sy, sz, sw);
                const v3 smp = v3_fma(py, P.fr.dv, P.fr.du * px);
                const v3 d = (ploc + smp) - P.fr.origin;
                if (!SPP1) (void)pt_hybrid_taus(sx, sy, sz, sw);  // the path's seed: drawn, never used by a path that hits nothing
                const v3 path_color = V3(1.0f, 1.0f, 1.0f) * sky_lookup(P.sc, d);
                frame_sum = SPP1 ? path_color : frame_sum + path_color;
            }
            const v3 mean = frame_sum * P.inv_spp;
            acc.x += mean.x; acc.y += mean.y; acc.z += mean.z; acc.w += 1.0f;
        }
        P.accum[pix] = acc;
        if (P.aov_prim) {  // (non-null only in the launch that holds a call's first frame)
            P.aov_prim[pix] = 0xFFFFFFFFu;
            P.aov_t[pix] = 0.0f;
        }
    }
    const uint32_t rays = wave_sum(in_image ? P.n_frames * (SPP1 ? 1u : P.spp) : 0u);
    if (lane == 0 && rays) atomicAdd(&P.counters[(blockIdx.x % PT_COUNTER_STRIPES) * PT_COUNTER_PITCH], (unsigned long long)rays);
}

// Plans the next launch from the wave time each tile cost in the previous one: tiles sorted
// costliest first (1024 linear buckets; ties in arbitrary order), each split into
// ceil(cost / target) items of consecutive frames, target = total cost / wanted item count, and into
// at least ceil(n_frames / max_chunk) items (a pixel's frames run one after the other, so an item's
// frame count bounds how long it takes no matter how cheap its tile is), but never into items of fewer
// than min_chunk frames (every item has a fixed cost: a partial-sum row, a job hand-out per pixel).
// Without cost data (first launch) every tile gets ceil(n_frames / default_chunk) items.
// A tile flagged in tile_sky gets no items at all (sky_tiles_kernel renders it).
// The plan only affects scheduling and the (fixed, per-plan) order of partial sums.
// One workgroup.  tile_items[t] = (first item, item count); clears cost for the next measurement.
__global__ __launch_bounds__(1024) void plan_kernel(uint32_t *cost, uint32_t n_tiles, uint32_t n_frames, uint32_t wanted_items,
                                                    uint32_t default_chunk, uint32_t max_chunk, uint32_t min_chunk, uint4 *items,
                                                    uint32_t item_cap,
                                                    uint2 *tile_items, uint32_t *sorted, const uint32_t *tile_sky)
{
    __shared__ uint32_t hist[1024];
    __shared__ unsigned long long total;
    __shared__ uint32_t maxc, running;
    const uint32_t tid = threadIdx.x;
    hist[tid] = 0;
    if (tid == 0) { total = 0; maxc = 1; running = 0; }
    __syncthreads();
    unsigned long long sum = 0;
    uint32_t m = 0;
    for (uint32_t i = tid; i < n_tiles; i += 1024) { const uint32_t c = cost[i]; sum += c; m = max(m, c); }
    atomicAdd(&total, sum);
    atomicMax(&maxc, m);
    __syncthreads();
    const bool have_cost = total > 0;
    const float target = have_cost ? fmaxf(1.0f, (float)total / (float)max(wanted_items, 1u)) : 1.0f;
    const uint32_t uniform_items = (n_frames + max(default_chunk, 1u) - 1u) / max(default_chunk, 1u);
    const float scale = 1023.0f / (float)maxc;
    // bucket sort, bucket 0 = costliest
    for (uint32_t i = tid; i < n_tiles; i += 1024) atomicAdd(&hist[1023u - min(1023u, (uint32_t)((float)cost[i] * scale))], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t b = 0; b < 1024; b++) { const uint32_t c = hist[b]; hist[b] = run; run += c; }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n_tiles; i += 1024) {
        const uint32_t b = 1023u - min(1023u, (uint32_t)((float)cost[i] * scale));
        sorted[atomicAdd(&hist[b], 1u)] = i;
    }
    __syncthreads();
    // items per tile, exclusive prefix sum in sorted order, item records
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t j = base + tid;
        uint32_t t = 0, n = 0;
        if (j < n_tiles) {
            t = sorted[j];
            n = have_cost ? (uint32_t)ceilf((float)cost[t] / target) : uniform_items;
            if (max_chunk) n = max(n, (n_frames + max_chunk - 1u) / max_chunk);
            n = min(max(n, 1u), max(n_frames / max(min_chunk, 1u), 1u));
            if (tile_sky && tile_sky[t]) n = 0;  // rendered by sky_tiles_kernel
        }
        // block-wide exclusive scan of n (reuse hist)
        hist[tid] = n;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t v = tid >= off ? hist[tid - off] : 0;
            __syncthreads();
            hist[tid] += v;
            __syncthreads();
        }
        const uint32_t first = running + hist[tid] - n;
        __syncthreads();
        if (tid == 1023) running += hist[1023];
        if (j < n_tiles) {
            // if the table is full the tile collapses to the items that still fit (at least its
            // frames are never dropped: the host sizes the table so that this cannot happen)
            const uint32_t fit = first < item_cap ? min(n, item_cap - first) : 0;
            tile_items[t] = make_uint2(first, fit);
            for (uint32_t k = 0; k < fit; k++) {
                const uint32_t fb = (uint32_t)(((unsigned long long)n_frames * k) / fit);
                const uint32_t fe = (uint32_t)(((unsigned long long)n_frames * (k + 1)) / fit);
                items[first + k] = make_uint4(t, fb, fe - fb, 0);
            }
        }
        __syncthreads();
    }
    for (uint32_t i = running + tid; i < item_cap; i += 1024) items[i] = make_uint4(0, 0, 0, 0);
    for (uint32_t i = tid; i < n_tiles; i += 1024) cost[i] = 0;
}

// Adds each tile's partial sums onto the running sum, in item order (deterministic for a given plan).
__global__ __launch_bounds__(256) void accumulate_kernel(float4 *accum, const float4 *partial, const uint2 *tile_items, uint32_t width,
                                                         uint32_t tiles_x, uint32_t cols, uint32_t rows)
{
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u), y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const uint2 ti = tile_items[(y / PT_TILE) * tiles_x + x / PT_TILE];
    if (ti.y == 0u) return;  // a tile without items (its frames went straight onto the sums): nothing to add
    const uint32_t lane = (y % PT_TILE) * PT_TILE + x % PT_TILE;
    const size_t pix = (size_t)y * width + x;
    float4 a = accum[pix];
    for (uint32_t k = 0; k < ti.y; k++) {
        const float4 p = partial[(size_t)(ti.x + k) * 64u + lane];
        a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
    accum[pix] = a;
}

// ---- accumulation read-out ----------------------------------------------------------------
// mean = sum / frames (frames kept in .w); mode 1 also clamps and quantises like an
// rgba8unorm store: round(clamp(c,0,1)*255).
__global__ __launch_bounds__(256) void resolve_kernel(const float4 *accum, uint32_t n, float4 *out_f, uint32_t *out_u8, int mode)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = accum[i];
    float4 m = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (a.w > 0.0f) m = make_float4(a.x / a.w, a.y / a.w, a.z / a.w, 1.0f);
    if (mode == 0) {
        out_f[i] = m;
    } else {
        auto q = [](float c) -> uint32_t {
            c = pt_min(pt_max(c, 0.0f), 1.0f);  // NaN -> 0
            return (uint32_t)pt_rint(c * 255.0f);
        };
        out_u8[i] = q(m.x) | (q(m.y) << 8) | (q(m.z) << 16) | (q(m.w) << 24);
    }
}

// ---- the reference's display window: the newest n frames as rgba8unorm textures ------------------
// compute.wgsl:124-127 stores every frame's colour into an rgba8unorm texture and copy.wgsl:27-37 shows the
// mean of the newest n of them (renderer.rs:492-516 shifts the textures down by one before every frame).
// Here the frames live in a ring of packed texels instead of being copied from texture to texture.
__device__ __forceinline__ uint32_t unorm8_store(float4 c)
{
    auto q = [](float v) -> uint32_t {
        v = pt_min(pt_max(v, 0.0f), 1.0f);  // NaN -> 0
        return (uint32_t)pt_rint(v * 255.0f);
    };
    return q(c.x) | (q(c.y) << 8) | (q(c.z) << 16) | (q(c.w) << 24);
}

// One finished frame (colour in .xyz, 1 in .w where the dispatch covered the pixel, zeros elsewhere): its texel
// into the ring slot, its colour onto the running sums exactly as the megakernel adds it, and the frame
// buffer back to zero for the next frame.
__global__ __launch_bounds__(256) void window_push_kernel(float4 *frame, float4 *accum, uint32_t *slot, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 f = frame[i];
    slot[i] = unorm8_store(f);  // an uncovered pixel keeps the texture's initial zeros
    if (f.w != 0.0f) {
        float4 a = accum[i];
        a.x += f.x; a.y += f.y; a.z += f.z; a.w += f.w;
        accum[i] = a;
        frame[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// copy.wgsl:27-37: colour = sum over i < count of textures[i] (newest first), divided by count.
__global__ __launch_bounds__(256) void window_mean_kernel(const uint32_t *ring, uint32_t n, uint32_t ring_frames, uint32_t newest,
                                                          uint32_t count, float4 *out_f, uint32_t *out_u8)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 color = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t slot = newest;
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t t = ring[(size_t)slot * n + i];
        // unorm8 -> f32 as a texture load does it: c / 255
        color.x += (float)(t & 255u) / 255.0f;
        color.y += (float)((t >> 8) & 255u) / 255.0f;
        color.z += (float)((t >> 16) & 255u) / 255.0f;
        color.w += (float)(t >> 24) / 255.0f;
        slot = slot ? slot - 1u : ring_frames - 1u;
    }
    const float c = (float)count;
    color = make_float4(color.x / c, color.y / c, color.z / c, color.w / c);
    if (out_f) out_f[i] = color;
    if (out_u8) out_u8[i] = unorm8_store(color);
}

// ---- equirectangular -> cube map: shaders/equirectangular.wgsl:12-81 -----------------------
__global__ __launch_bounds__(256) void equirect_kernel(const float4 *src, uint32_t src_w, uint32_t src_h, float4 *dst, uint32_t size)
{
    const uint32_t x = blockIdx.x * 16u + (threadIdx.x & 15u);
    const uint32_t y = blockIdx.y * 16u + (threadIdx.x >> 4);
    const uint32_t f = blockIdx.z;
    if (x >= size || y >= size) return;
    v3 fw, up, rt;
    switch (f) {  // forward, up, right per face (equirectangular.wgsl:25-62)
    case 0: fw = V3(1, 0, 0); up = V3(0, 1, 0); rt = V3(0, 0, -1); break;
    case 1: fw = V3(-1, 0, 0); up = V3(0, 1, 0); rt = V3(0, 0, 1); break;
    case 2: fw = V3(0, -1, 0); up = V3(0, 0, 1); rt = V3(1, 0, 0); break;
    case 3: fw = V3(0, 1, 0); up = V3(0, 0, -1); rt = V3(1, 0, 0); break;
    case 4: fw = V3(0, 0, 1); up = V3(0, 1, 0); rt = V3(1, 0, 0); break;
    default: fw = V3(0, 0, -1); up = V3(0, 1, 0); rt = V3(-1, 0, 0); break;
    }
    const float fsz = (float)size;
    const float cu = pt_fma((float)x / fsz, 2.0f, -1.0f), cv = pt_fma((float)y / fsz, 2.0f, -1.0f);
    const v3 sph = normalize3(v3_fma(cv, up, v3_fma(cu, rt, fw)));
    const float eu = pt_fma(pt_atan2(sph.z, sph.x), 0.1591f, 0.5f);
    const float ev = pt_fma(pt_asin(sph.y), 0.3183f, 0.5f);
    const int32_t ix = (int32_t)(eu * (float)src_w), iy = (int32_t)(ev * (float)src_h);
    float4 texel = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (ix >= 0 && iy >= 0 && (uint32_t)ix < src_w && (uint32_t)iy < src_h) texel = src[(size_t)iy * src_w + (uint32_t)ix];
    dst[((size_t)f * size + y) * size + x] = texel;
}

}  // namespace

static size_t spill_entries(uint32_t stack_depth) { return stack_depth > PT_LDS_STACK ? stack_depth - PT_LDS_STACK : 0; }

size_t pt_trace_gstack_bytes(uint32_t n_blocks, uint32_t stack_depth)
{
    return (size_t)n_blocks * PT_TRACE_BLOCK * spill_entries(stack_depth) * sizeof(uint32_t);
}

static hipError_t launch_items(const PtTraceParams &p, uint32_t n_blocks, bool overflow, bool detailed_stats, hipStream_t stream)
{
    const dim3 grid(n_blocks, 1, 1), block(PT_TRACE_BLOCK, 1, 1);
#ifdef PT_TUNING
    static const size_t lds_pad = getenv("PTRT_LDS_PAD") ? (size_t)atoi(getenv("PTRT_LDS_PAD")) : 0;  // lowers occupancy
#else
    const size_t lds_pad = 0;
#endif
#define PT_LAUNCH(ST, OV, GZ, S1, DI) hipLaunchKernelGGL((trace_kernel<ST, OV, GZ, S1, DI>), grid, block, lds_pad, stream, p)
#define PT_LAUNCH_DI(OV, DI)                                                  \
    do {                                                                      \
        if (detailed_stats) PT_LAUNCH(true, OV, true, false, DI);             \
        else if (p.has_gizmo) { if (p.spp == 1) PT_LAUNCH(false, OV, true, true, DI); else PT_LAUNCH(false, OV, true, false, DI); } \
        else { if (p.spp == 1) PT_LAUNCH(false, OV, false, true, DI); else PT_LAUNCH(false, OV, false, false, DI); } \
    } while (0)
    const bool direct = p.items == nullptr;
    if (overflow) { if (direct) PT_LAUNCH_DI(true, true); else PT_LAUNCH_DI(true, false); }
    else { if (direct) PT_LAUNCH_DI(false, true); else PT_LAUNCH_DI(false, false); }
#undef PT_LAUNCH_DI
#undef PT_LAUNCH
    return hipGetLastError();
}

// Pooled variant: a fixed grid of persistent workgroups that take items from *sch.next_item.
uint32_t pt_pool_max_groups(uint32_t n_cus) { return n_cus * std::max(1, 4 * PT_PP_WAVES_PER_EU / PT_PP_WAVES); }

size_t pt_pool_gstack_bytes(uint32_t n_cus, uint32_t stack_depth)
{
    if (stack_depth <= PT_PP_STACK) return 0;
    return (size_t)pt_pool_max_groups(n_cus) * PT_PP_THREADS * (stack_depth - PT_PP_STACK) * sizeof(uint32_t);
}

template <typename K>
static uint32_t pool_groups_per_cu(K kernel)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, PT_PP_THREADS, 0) != hipSuccess || nb < 1) nb = 1;
    return (uint32_t)std::min(nb, std::max(1, 4 * PT_PP_WAVES_PER_EU / PT_PP_WAVES));
}

static hipError_t launch_pool(PtTraceParams p, uint32_t n_items, uint32_t stack_depth, bool detailed_stats, const PtSchedule &sch, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(sch.next_item, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    p.next_item = sch.next_item;
    p.n_items = n_items;
    p.gstack = stack_depth > PT_PP_STACK ? sch.pool_gstack : nullptr;
    p.gstack_cap = stack_depth;
    const dim3 block(PT_PP_THREADS, 1, 1);
    const uint32_t want = (n_items + PT_PP_WAVES - 1u) / PT_PP_WAVES;
    // items the launch really has: n_items is the table's size; a plan aims for wanted_items, or has one per tile and frame
    p.cost_key = p.tiles_x | ((p.n_frames >= 64u ? 1u : 0u) << 16);  // (one word for the pixel job's report: fewer scalar registers)
    const uint64_t est_items = p.items == nullptr ? n_items : (sch.max_chunk == 1u ? (uint64_t)p.n_tiles * p.n_frames : sch.wanted_items);
#define PT_LAUNCH(ST, GZ, S1, DI)                                                                              \
    do {                                                                                                       \
        static const uint32_t per_cu = pool_groups_per_cu(pool_kernel<ST, GZ, S1, DI>);                        \
        uint32_t groups = sch.pool_groups ? sch.pool_groups : per_cu * sch.n_cus;                              \
        groups = std::max(1u, std::min(std::min(groups, want), pt_pool_max_groups(sch.n_cus)));                \
        /* (a launch with fewer than 192 items per workgroup takes half batches: whole ones leave its last  */ \
        /* items to too few groups -- 16 frames of the bench scene 1.95 against 2.06 ms)                     */ \
        p.item_batch = est_items >= 192u * groups ? PT_PP_ITEM_BATCH : PT_PP_ITEM_BATCH / 2u;                   \
        hipLaunchKernelGGL((pool_kernel<ST, GZ, S1, DI>), dim3(groups, 1, 1), block, 0, stream, p);            \
    } while (0)
#define PT_LAUNCH_DI(DI)                                                  \
    do {                                                                  \
        if (detailed_stats) PT_LAUNCH(true, true, false, DI);             \
        else if (p.has_gizmo) { if (p.spp == 1) PT_LAUNCH(false, true, true, DI); else PT_LAUNCH(false, true, false, DI); } \
        else { if (p.spp == 1) PT_LAUNCH(false, false, true, DI); else PT_LAUNCH(false, false, false, DI); } \
    } while (0)
    if (p.items == nullptr) PT_LAUNCH_DI(true); else PT_LAUNCH_DI(false);
#undef PT_LAUNCH_DI
#undef PT_LAUNCH
    return hipGetLastError();
}

uint32_t pt_trace_num_tiles(uint32_t width, uint32_t height)
{
    return ((width + PT_TILE - 1u) / PT_TILE) * ((height + PT_TILE - 1u) / PT_TILE);
}

uint32_t pt_trace_item_cap(uint32_t n_tiles, uint32_t n_frames, const PtSchedule &sch)
{
    const uint64_t uniform = (uint64_t)n_tiles * ((n_frames + std::max(sch.default_chunk, 1u) - 1u) / std::max(sch.default_chunk, 1u));
    const uint64_t capped = sch.max_chunk ? (uint64_t)n_tiles * ((n_frames + sch.max_chunk - 1u) / sch.max_chunk) : 0;
    const uint64_t planned = (uint64_t)n_tiles + sch.wanted_items + capped + 1024u;
    return (uint32_t)std::min<uint64_t>(std::max(uniform, planned), 0x7FFFFFFFull);
}

__global__ void delay_kernel(uint32_t ticks);

hipError_t pt_launch_trace(PtTraceParams p, uint32_t stack_depth, bool detailed_stats, const PtSchedule &sch, hipStream_t stream)
{
    p.tiles_x = (p.width + PT_TILE - 1u) / PT_TILE;
    p.tiles_y = (p.height + PT_TILE - 1u) / PT_TILE;
    p.n_tiles = p.tiles_x * p.tiles_y;
    if (p.n_tiles == 0 || p.n_frames == 0) return hipSuccess;
    const bool overflow = spill_entries(stack_depth) > 0;
    p.cost = nullptr;
    // Few frames or no scheduling buffers: one item per tile, frames added in order onto the stored sums.
    // The pooled kernel has its own (shorter) LDS stack and spill buffer.
    const bool pool = sch.use_pool && sch.next_item && sch.n_cus && (stack_depth <= PT_PP_STACK || sch.pool_gstack) && p.width <= 0xFFFFu &&
                      p.height <= 0xFFFFu && p.depth >= 1u;  // it packs pixel coordinates into 16 bits each and gives every new path a ray
    const bool direct = p.n_frames < sch.min_planned_frames || !sch.items || !sch.partial || !sch.tile_items || !sch.sorted || !sch.cost;
    // the per-tile kernel's spill buffer has one column per lane of the launch's largest grid
    const bool spills = overflow && !pool;
    if (spills && (!p.gstack || p.gstack_cap < stack_depth || p.gstack_blocks < p.n_tiles)) return hipErrorInvalidValue;
    if (direct) {
        p.items = nullptr;
        p.partial = nullptr;
        if (pool) return launch_pool(p, p.n_tiles, stack_depth, detailed_stats, sch, stream);
        return launch_items(p, p.n_tiles, overflow, detailed_stats, stream);
    }
    hipError_t e;
    if (!*sch.cost_valid) {
        // No measurement yet: render the first frames directly to get one, then plan the rest.
        PtTraceParams first = p;
        first.items = nullptr;
        first.partial = nullptr;
        first.n_frames = std::min(p.n_frames, sch.probe_frames);
        first.cost = sch.cost;
        e = pool ? launch_pool(first, p.n_tiles, stack_depth, detailed_stats, sch, stream) : launch_items(first, p.n_tiles, overflow, detailed_stats, stream);
        if (e != hipSuccess) return e;
        *sch.cost_valid = true;
        p.time0 += first.n_frames * p.stride;
        p.n_frames -= first.n_frames;
        p.aov_prim = nullptr;
        if (p.n_frames == 0) return hipSuccess;
    }
    const uint32_t cap = pt_trace_item_cap(p.n_tiles, p.n_frames, sch);
    if (cap > sch.item_cap || (spills && p.gstack_blocks < cap)) return hipErrorInvalidValue;
    // Tiles that see only the sky (no per-ray statistics): flagged when the launch is planned,
    // left out of the plan, rendered by sky_tiles_kernel.  A plan that is run again keeps its flags.
    const bool sky_tiles = sch.sky_tiles && sch.tile_sky && p.spp >= 1u && p.depth >= 1u && !detailed_stats;
    if (!sch.reuse_plan) {
        if (sky_tiles) {
            hipLaunchKernelGGL(classify_tiles_kernel, dim3((p.n_tiles + 3u) / 4u), dim3(256), 0, stream, p, sch.tile_sky);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        } else if (sch.tile_sky && (e = hipMemsetAsync(sch.tile_sky, 0, (size_t)p.n_tiles * sizeof(uint32_t), stream)) != hipSuccess) return e;
        hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(1024), 0, stream, sch.cost, p.n_tiles, p.n_frames, sch.wanted_items, sch.default_chunk, sch.max_chunk,
                           sch.min_chunk, sch.items, cap, sch.tile_items, sch.sorted, sky_tiles ? sch.tile_sky : nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        p.cost = sch.cost;  // cleared by plan_kernel, refilled by this launch for the next plan
    } else if (sch.measure_cost) {
        if ((e = hipMemsetAsync(sch.cost, 0, (size_t)p.n_tiles * sizeof(uint32_t), stream)) != hipSuccess) return e;
        p.cost = sch.cost;
    }
    p.items = sch.items;
    p.partial = sch.partial;
    // The megakernel first (its workgroups take their places), then the sky loop into what is left of every CU.  The two
    // run on a pair of streams of the context's own that is known to reach the GPU through two hardware queues
    // (pt_streams_run_side_by_side); the caller's stream only forks into them and joins them again.
    const bool beside = sky_tiles && sch.main_stream && sch.side_stream && sch.fork_event && sch.start_event && sch.join_event && sch.main_event && p.cols != 0 && p.rows != 0;
    const hipStream_t ms = beside ? sch.main_stream : stream;
    // (the loop's stream waits for an event that the main stream records right before the megakernel, and then a few
    // tens of microseconds more: released together with the megakernel -- two queues start within the same microsecond
    // -- the loop's many small workgroups take places on the CUs that the megakernel's persistent groups then lack for
    // the whole launch: 15.5-16.3 instead of 14.4 ms)
    if (beside && ((e = hipEventRecord(sch.fork_event, stream)) != hipSuccess || (e = hipStreamWaitEvent(ms, sch.fork_event, 0)) != hipSuccess ||
                   (e = hipEventRecord(sch.start_event, ms)) != hipSuccess)) return e;
    e = pool ? launch_pool(p, cap, stack_depth, detailed_stats, sch, ms) : launch_items(p, cap, overflow, detailed_stats, ms);
    if (beside) {  // (also after a failed launch: the caller's stream must not run ahead of what did get started)
        hipError_t e2 = hipEventRecord(sch.main_event, ms);
        if (e2 == hipSuccess) e2 = hipStreamWaitEvent(stream, sch.main_event, 0);
        if (e == hipSuccess) e = e2;
    }
    if (e != hipSuccess) return e;
    if (p.cols == 0 || p.rows == 0) return hipSuccess;  // image smaller than one 16x16 group under PT_FLAG_REFERENCE_DISPATCH: no pixel is rendered
    if (sky_tiles) {
        const hipStream_t ss = beside ? sch.side_stream : stream;
        if (beside) {
            if ((e = hipStreamWaitEvent(ss, sch.start_event, 0)) != hipSuccess) return e;
            hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(1), 0, ss, 3000u);  // 30 us
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        if (p.spp == 1u) hipLaunchKernelGGL(sky_tiles_kernel<true>, dim3((p.n_tiles + 3u) / 4u), dim3(256), 0, ss, p, sch.tile_sky);
        else hipLaunchKernelGGL(sky_tiles_kernel<false>, dim3((p.n_tiles + 3u) / 4u), dim3(256), 0, ss, p, sch.tile_sky);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (beside) {
            if ((e = hipEventRecord(sch.join_event, ss)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(stream, sch.join_event, 0)) != hipSuccess) return e;
        }
    }
    hipLaunchKernelGGL(accumulate_kernel, dim3((p.cols + 63u) / 64u, (p.rows + 3u) / 4u), dim3(256), 0, stream, p.accum, p.partial,
                       sch.tile_items, p.width, p.tiles_x, p.cols, p.rows);
    return hipGetLastError();
}

// ---- do two streams reach the GPU through different hardware queues? -------------------------------
// HIP multiplexes a process's streams onto a few hardware queues (4 by default), and kernels of two streams that share
// one run one after the other.  Two one-thread kernels, one per stream, each raise a flag and wait -- for at most half a
// millisecond -- for the other's: both see it only if they were on the GPU at the same time.  (On one queue the first
// runs into its time limit before the second starts; the second then finds the first's flag, so BOTH must report.)
// Holds its stream for `ticks` of the 100 MHz clock (bounded by the trip count as well).
__global__ void delay_kernel(uint32_t ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t i = 0; i < 4096u; i++) {
        if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

__global__ void stream_probe_kernel(uint32_t *words, uint32_t mine)
{
    uint32_t *flag = words, *seen = words + 2;
    __hip_atomic_store(&flag[mine], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    uint32_t ok = 0;
    for (uint32_t i = 0; i < (1u << 16) && ok == 0u; i++) {  // (bounded twice: by the clock and by the trip count)
        ok = __hip_atomic_load(&flag[mine ^ 1u], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 50000ull) break;
        __builtin_amdgcn_s_sleep(16);
    }
    seen[mine] = ok;
}

hipError_t pt_streams_run_side_by_side(hipStream_t a, hipStream_t b, uint32_t *dev_words, bool *result)
{
    *result = false;
    hipError_t e;
    if ((e = hipMemsetAsync(dev_words, 0, 4 * sizeof(uint32_t), a)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(a)) != hipSuccess) return e;
    hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(1), 0, a, dev_words, 0u);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(1), 0, b, dev_words, 1u);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(a)) != hipSuccess || (e = hipStreamSynchronize(b)) != hipSuccess) return e;
    uint32_t h[4] = {0, 0, 0, 0};
    if ((e = hipMemcpy(h, dev_words, sizeof(h), hipMemcpyDeviceToHost)) != hipSuccess) return e;
    *result = h[2] != 0u && h[3] != 0u;
    return hipSuccess;
}

size_t pt_ray_query_gstack_bytes(uint32_t n, uint32_t stack_depth)
{
    const size_t blocks = ((size_t)n + PT_BLOCK - 1) / PT_BLOCK;
    return blocks * PT_BLOCK * spill_entries(stack_depth) * sizeof(uint32_t);
}

hipError_t pt_launch_ray_query(const PtRayQuery &q, uint32_t stack_depth, hipStream_t stream)
{
    if (q.n == 0) return hipSuccess;
    const dim3 grid((q.n + PT_BLOCK - 1) / PT_BLOCK, 1, 1), block(PT_BLOCK, 1, 1);
    const bool overflow = spill_entries(stack_depth) > 0;
    if (overflow && (!q.gstack || q.gstack_cap < stack_depth)) return hipErrorInvalidValue;
    if (overflow) hipLaunchKernelGGL((ray_query_kernel<true>), grid, block, 0, stream, q);
    else hipLaunchKernelGGL((ray_query_kernel<false>), grid, block, 0, stream, q);
    return hipGetLastError();
}

#ifdef PT_DIAG
extern "C" int pt_diag_read(void *host, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_diag), bytes, 0, hipMemcpyDeviceToHost);
}
extern "C" int pt_diag_clear(void)
{
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_diag)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * 8 * PT_DIAG_MAX_WAVES);
}
extern "C" int pt_diag_read_marks(void *host, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_diag_marks), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

hipError_t pt_launch_resolve(const float4 *accum, uint32_t n_pixels, float4 *out_f, uint32_t *out_u8, int mode, hipStream_t stream)
{
    if (n_pixels == 0) return hipSuccess;
    hipLaunchKernelGGL(resolve_kernel, dim3((n_pixels + 255u) / 256u), dim3(256), 0, stream, accum, n_pixels, out_f, out_u8, mode);
    return hipGetLastError();
}

hipError_t pt_launch_window_push(float4 *frame, float4 *accum, uint32_t *slot, uint32_t n_pixels, hipStream_t stream)
{
    if (n_pixels == 0) return hipSuccess;
    hipLaunchKernelGGL(window_push_kernel, dim3((n_pixels + 255u) / 256u), dim3(256), 0, stream, frame, accum, slot, n_pixels);
    return hipGetLastError();
}

hipError_t pt_launch_window_mean(const uint32_t *ring, uint32_t n_pixels, uint32_t ring_frames, uint32_t newest, uint32_t count,
                                 float4 *out_f, uint32_t *out_u8, hipStream_t stream)
{
    if (n_pixels == 0) return hipSuccess;
    hipLaunchKernelGGL(window_mean_kernel, dim3((n_pixels + 255u) / 256u), dim3(256), 0, stream, ring, n_pixels, ring_frames, newest, count, out_f, out_u8);
    return hipGetLastError();
}

hipError_t pt_launch_equirect(const float4 *src, uint32_t src_w, uint32_t src_h, float4 *dst, uint32_t size, hipStream_t stream)
{
    if (size == 0) return hipSuccess;
    hipLaunchKernelGGL(equirect_kernel, dim3((size + 15u) / 16u, (size + 15u) / 16u, 6), dim3(256), 0, stream, src, src_w, src_h, dst, size);
    return hipGetLastError();
}
