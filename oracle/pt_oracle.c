/*
 * pt_oracle.c -- CPU oracle for the path-tracing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A scalar C restatement of the reference's compute megakernel and of the host code
 * that feeds it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (libptrt.so) never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures, and it
 * cannot be compiled or run in this environment (no Rust toolchain, no Vulkan/WGSL
 * implementation).  The only pins are integer facts derivable from the source itself
 * (HybridTaus known-answer table, struct sizes/offsets, BVH invariants), which
 * tests/test_oracle_*.py check.  Everything floating-point rests on this restatement.
 *
 * Reference files followed (paths relative to the reference checkout):
 *   shaders/compute.wgsl:83-128   main            -> render_pixel()
 *   shaders/compute.wgsl:130-207  rayColor        -> ray_color()
 *   shaders/compute.wgsl:209-212  getBackgroundColor -> sky_lookup()
 *   shaders/compute.wgsl:214-302  hitScene        -> hit_scene()
 *   shaders/compute.wgsl:304-350  hitSphere       -> hit_sphere()
 *   shaders/compute.wgsl:352-406  hitTriangle     -> hit_triangle()
 *   shaders/compute.wgsl:408-439  scatter/reflect/reflectance/refract
 *   shaders/compute.wgsl:441-456  hitAabb         -> hit_aabb()
 *   shaders/include/utils.wgsl:3-5,15-22,33-58    rand/randomUnit/hybridTaus
 *   shaders/equirectangular.wgsl:12-81            -> pto_equirect_to_cube()
 *   src/scene/bvh.rs:34-162                       -> pto_bvh_build()
 *   src/scene/plane.rs:16-45                      -> pto_plane_triangles()
 *   src/renderer.rs:565-569 (floor-div dispatch)  -> PTO_FLAG_REFERENCE_DISPATCH
 *
 * Floating-point contract.  WGSL leaves operation fusing and the accuracy of
 * sin/cos/tan/pow/normalize to the implementation, and the shader's fract(sin(x)*43758)
 * hash amplifies one ulp of difference into a different random number, so results are
 * only reproducible against ONE fixed evaluation.  This file fixes it as follows; every
 * operation is IEEE-754 binary32, round-to-nearest-even, subnormals kept:
 *   - a*b+c, a*b-c, c-a*b patterns written in the shader are ONE fused multiply-add
 *     (fmaf) exactly where this file writes fmaf, and nowhere else (build with
 *     -ffp-contract=off);  dot(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x));
 *     cross(a,b).x = fma(a.y,b.z, -(a.z*b.y)) (cyclic);
 *   - '/' and sqrt are correctly rounded; at four places where the shader divides several
 *     values by one denominator (sphere normal / radius, the two sphere roots / 2a, the two
 *     cube-face coordinates / major axis, colour / spp) the quotient is value * (1/denominator),
 *     as WGSL's division accuracy allows;
 *     normalize(v) = v * (1/sqrt(dot(v,v)));
 *   - min/max are IEEE minNum/maxNum (fminf/fmaxf);
 *   - sin/cos: 3-term fused Cody-Waite reduction by pi/2 + degree-7/8 minimax
 *     polynomials (pto_sincosf below); tan = sin/cos; pow(x,5) = ((x*x)*(x*x))*x;
 *     atan2/asin (equirect only): pto_atan2f/pto_asinf below;
 *   - the zero-weighted fuzz term of reflect() (compute.wgsl:420-422) is dropped;
 *   - cube-map face selection and nearest texel follow the Vulkan rules (major axis
 *     ties resolved z, then y, then x; texel = floor(s*size) clamped to the face).
 * The HIP kernels evaluate the same sequence, so oracle and kernel agree bit for bit.
 *
 * Deliberate deviation from the reference's undefined behaviour: the traversal stack
 * is PTO_STACK_CAP entries (the shader's is 15 with no bounds check, compute.wgsl:240,
 * 273-274); rays whose stack would pass 15 are counted in stats.overflow15.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PTO_STACK_CAP 256
#define PTO_FLAG_REFERENCE_DISPATCH 1u
#define PTO_FLAG_TRIANGLE_MATERIALS 2u   /* extension: triangles use their own albedo/material (model.rs:66-67) */

typedef struct { float x, y, z; } v3;

/* ---- byte layouts (identical to the reference's Pod structs) ---------------------- */
typedef struct { float origin[3]; float focal; float forward[3]; float vfov;
                 float right[3]; uint32_t pad1; float up[3]; uint32_t pad2; } pto_camera;   /* camera.rs:73-84 */
typedef struct { uint32_t spp; uint32_t depth; float t_min; float t_max; } pto_settings;      /* renderer.rs:604-611 */
typedef struct { float center[3]; float radius; float albedo[3]; float material; } pto_sphere; /* sphere.rs:81-88 */
typedef struct { uint32_t count; uint32_t pad[3]; pto_sphere s[256]; } pto_sphere_data;       /* sphere.rs:105-111 */
typedef struct { float a[3], p0, b[3], p1, c[3], p2, na[3], p3, nb[3], p4, nc[3], p5;
                 float albedo[3]; uint32_t material; } pto_triangle;                           /* model.rs:51-68 */
typedef struct { float mn[3]; uint32_t left; float mx[3]; uint32_t count; } pto_node;          /* bvh.rs:7-14 */

_Static_assert(sizeof(pto_camera) == 64, "CameraBuffer is 64 B");
_Static_assert(sizeof(pto_settings) == 16, "Settings is 16 B");
_Static_assert(sizeof(pto_sphere) == 32, "SphereBuffer is 32 B");
_Static_assert(sizeof(pto_sphere_data) == 8208, "SphereDataBuffer is 8208 B");
_Static_assert(sizeof(pto_triangle) == 112, "TriangleBuffer is 112 B");
_Static_assert(sizeof(pto_node) == 32, "Node is 32 B");

typedef struct {
    uint32_t width, height;
    const pto_camera *camera;
    const pto_settings *settings;
    const pto_sphere_data *spheres;
    const pto_triangle *triangles; uint32_t n_tris;
    const pto_node *nodes;         uint32_t n_nodes;
    const uint32_t *tri_idx;       uint32_t n_idx;
    const float *sky;              uint32_t sky_size;   /* 6 faces, RGBA32F, +X -X +Y -Y +Z -Z */
    uint32_t flags;
} pto_scene;

typedef struct {
    uint64_t rays;          /* hitScene calls (compute.wgsl:138) */
    uint64_t paths;         /* rayColor calls */
    uint64_t interior;      /* interior-node steps (two hitAabb each) */
    uint64_t tri_tests;     /* hitTriangle calls */
    uint64_t tri_accept;    /* triangle hits that replaced the record */
    uint64_t sphere_tests;  /* hitSphere calls */
    uint64_t sky_lookups;   /* getBackgroundColor calls */
    uint64_t bounces;       /* material switch executions */
    uint64_t sphere_accept; /* sphere hits that replaced the record */
    uint64_t overflow15;    /* rays whose stack would exceed the shader's 15 */
    uint64_t max_stack;     /* deepest stack seen */
} pto_stats;

/* ---- canonical math --------------------------------------------------------------- */
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 vfma(float s, v3 a, v3 c) { return V(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z)); }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline v3 normalize3(v3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return vscale(a, inv); }
static inline v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }

/* sin and cos of x together.  k = nearest integer to x*2/pi; r = x - k*pi/2 in three
 * fused steps; polynomials on [-pi/4, pi/4]; quadrant from k mod 4.  |x| beyond ~1e5
 * loses accuracy gracefully but stays deterministic. */
void pto_sincosf(float x, float *s_out, float *c_out)
{
    const float k = rintf(x * 0.63661977236758134f);              /* 0x3f22f983 */
    float r = fmaf(-k, 1.57079637050628662f, x);                  /* 0x3fc90fdb */
    r = fmaf(-k, -4.37113882867379e-08f, r);                      /* 0xb33bbd2e */
    r = fmaf(-k, -1.71512451000588e-15f, r);                      /* 0xa6f72ced */
    const float s = r * r;
    float ps = fmaf(s, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(s, ps, -1.6666654611e-1f);
    ps = fmaf(s * r, ps, r);                                      /* sin(r) */
    float pc = fmaf(s, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(s, pc, 4.166664568298827e-2f);
    pc = fmaf(s * s, pc, fmaf(-0.5f, s, 1.0f));                   /* cos(r) */
    /* k is integral; beyond int range the quadrant is meaningless but must be defined */
    int q = (k >= -2147483648.0f && k < 2147483648.0f) ? (int)k : 0;
    float sv = (q & 1) ? pc : ps;
    float cv = (q & 1) ? ps : pc;
    if (q & 2) sv = -sv;
    if ((q + 1) & 2) cv = -cv;
    *s_out = sv; *c_out = cv;
}
float pto_sinf(float x) { float s, c; pto_sincosf(x, &s, &c); return s; }
float pto_cosf(float x) { float s, c; pto_sincosf(x, &s, &c); return c; }
float pto_tanf(float x) { float s, c; pto_sincosf(x, &s, &c); return s / c; }

/* atan on [0,1] by odd minimax polynomial, then octant fix-up (equirect only). */
static inline float atan_unit(float a)
{
    const float s = a * a;
    float p = fmaf(s, 0.00282363896258175373077393f, -0.0159569028764963150024414f);
    p = fmaf(s, p, 0.0425049886107444763183594f);
    p = fmaf(s, p, -0.0748900920152664184570312f);
    p = fmaf(s, p, 0.106347933411598205566406f);
    p = fmaf(s, p, -0.142027363181114196777344f);
    p = fmaf(s, p, 0.199926957488059997558594f);
    p = fmaf(s, p, -0.333331018686294555664062f);
    return fmaf(s * a, p, a);
}
float pto_atan2f(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float a = (mx == 0.0f) ? 0.0f : atan_unit(mn / mx);
    if (ay > ax) a = 1.57079637050628662f - a;
    if (x < 0.0f) a = 3.14159274101257324f - a;
    return (y < 0.0f) ? -a : a;
}
float pto_asinf(float x)
{
    /* asin(x) = atan2(x, sqrt((1-x)(1+x))) */
    return pto_atan2f(x, sqrtf((1.0f - x) * (1.0f + x)));
}

static inline float fract1(float x) { return x - floorf(x); }

/* utils.wgsl:3-5 */
static inline float hash_rand(float sx, float sy)
{
    const float d = fmaf(sy, 78.233f, sx * 12.9898f);
    return fract1(pto_sinf(d) * 43758.5453f);
}

/* utils.wgsl:15-22 */
static inline v3 random_unit(float sx, float sy)
{
    const float z = fmaf(hash_rand(sx, sy), 2.0f, -1.0f);
    const float a = (hash_rand(sx * z, sy * z) * 2.0f) * 3.14159274101257324f;
    const float r = sqrtf(fmaf(-z, z, 1.0f));
    float sa, ca;
    pto_sincosf(a, &sa, &ca);
    return V(r * ca, r * sa, z);
}

/* utils.wgsl:33-58 */
static inline uint32_t tau_step(uint32_t z, int s1, int s2, int s3, uint32_t m)
{
    const uint32_t b = ((z << s1) ^ z) >> s2;
    return ((z & m) << s3) ^ b;
}
float pto_hybrid_taus(uint32_t st[4])
{
    st[0] = tau_step(st[0], 13, 19, 12, 4294967294u);
    st[1] = tau_step(st[1], 2, 25, 4, 4294967288u);
    st[2] = tau_step(st[2], 3, 11, 17, 4294967280u);
    st[3] = 1664525u * st[3] + 1013904223u;
    return 2.3283064365387e-10f * (float)(st[0] ^ st[1] ^ st[2] ^ st[3]);
}

/* ---- scene intersection ----------------------------------------------------------- */
typedef struct { v3 o, d; } ray_t;
typedef struct {
    int hit; float t; v3 p; v3 normal; int front; v3 att; float material;
    uint32_t prim;  /* oracle-only AOV: sphere i -> 0x80000000|i, triangle -> its index */
} hit_t;

typedef struct { const pto_scene *sc; pto_stats st; float t_min, t_max; } ctx_t;

/* compute.wgsl:304-350 */
static hit_t hit_sphere(ctx_t *cx, ray_t ray, const pto_sphere *sp, uint32_t index)
{
    cx->st.sphere_tests++;
    const v3 center = ld3(sp->center);
    const v3 oc = vsub(ray.o, center);
    const float a = dot3(ray.d, ray.d);
    const float b = 2.0f * dot3(ray.d, oc);
    const float c = fmaf(-sp->radius, sp->radius, dot3(oc, oc));
    const float disc = fmaf(b, b, -((4.0f * a) * c));
    hit_t h; memset(&h, 0, sizeof h);
    h.att = ld3(sp->albedo); h.material = sp->material; h.prim = 0x80000000u | index;
    if (disc < 0.0f) return h;
    const float sq = sqrtf(disc), inv_den = 1.0f / (2.0f * a);
    float root = (-b - sq) * inv_den;
    if (root <= cx->t_min || root >= cx->t_max) root = (-b + sq) * inv_den;
    if (root <= cx->t_min || root >= cx->t_max) return h;
    h.hit = 1; h.t = root; h.p = vfma(root, ray.d, ray.o);
    const v3 pc = vsub(h.p, center);
    const v3 on = vscale(pc, 1.0f / sp->radius);
    h.front = dot3(ray.d, on) < 0.0f;
    h.normal = h.front ? on : vneg(on);
    /* u32(material): saturating, NaN -> 0 */
    if (sp->material >= 3.0f && sp->material < 4.0f) {
        const float dn = dot3(ray.d, h.normal);
        if (!(dn <= 0.2f && dn >= -0.2f)) h.hit = 0;
    }
    return h;
}

/* compute.wgsl:352-406 */
static hit_t hit_triangle(ctx_t *cx, ray_t ray, const pto_triangle *tr, uint32_t index)
{
    cx->st.tri_tests++;
    const v3 A = ld3(tr->a);
    const v3 e1 = vsub(ld3(tr->b), A), e2 = vsub(ld3(tr->c), A);
    const v3 hv = cross3(ray.d, e2);
    const float a = dot3(e1, hv);
    hit_t h; memset(&h, 0, sizeof h);
    h.att = V(1.0f, 1.0f, 1.0f); h.material = 0.0f; h.prim = index;   /* compute.wgsl:359-367: the record's own fields are ignored */
    if (cx->sc->flags & PTO_FLAG_TRIANGLE_MATERIALS) { h.att = ld3(tr->albedo); h.material = (float)tr->material; }
    if (a > -0.00001f && a < 0.00001f) return h;
    const float f = 1.0f / a;
    const v3 s = vsub(ray.o, A);
    const float u = f * dot3(s, hv);
    if (u < 0.0f || u > 1.0f) return h;
    const v3 q = cross3(s, e1);
    const float v = f * dot3(ray.d, q);
    if (v < 0.0f || u + v > 1.0f) return h;
    const float t = f * dot3(e2, q);
    if (t <= cx->t_min || t >= cx->t_max) return h;
    h.hit = 1; h.t = t; h.p = vfma(t, ray.d, ray.o);
    const float w = (1.0f - u) - v;
    const v3 an = ld3(tr->na), bn = ld3(tr->nb), cn = ld3(tr->nc);
    const v3 n = V(fmaf(cn.x, v, fmaf(bn.x, u, an.x * w)),
                   fmaf(cn.y, v, fmaf(bn.y, u, an.y * w)),
                   fmaf(cn.z, v, fmaf(bn.z, u, an.z * w)));
    const v3 on = normalize3(n);
    h.front = dot3(ray.d, on) < 0.0f;
    h.normal = h.front ? on : vneg(on);
    return h;
}

/* compute.wgsl:441-456 */
static float hit_aabb(ray_t ray, const pto_node *n)
{
    const v3 inv = V(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    const v3 t1 = vmul(vsub(ld3(n->mn), ray.o), inv);
    const v3 t2 = vmul(vsub(ld3(n->mx), ray.o), inv);
    const float tmin = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fminf(t1.z, t2.z));
    const float tmax = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fmaxf(t1.z, t2.z));
    if (tmin > tmax || tmax < 0.0f) return 99999.0f;
    return tmin;
}

/* compute.wgsl:214-302 */
static hit_t hit_scene(ctx_t *cx, ray_t ray)
{
    const pto_scene *sc = cx->sc;
    cx->st.rays++;
    hit_t rec; memset(&rec, 0, sizeof rec); rec.prim = 0xFFFFFFFFu;
    uint32_t nsph = sc->spheres ? sc->spheres->count : 0;
    if (nsph > 256) nsph = 256;
    for (uint32_t i = 0; i < nsph; i++) {
        hit_t h = hit_sphere(cx, ray, &sc->spheres->s[i], i);
        if (!h.hit) continue;
        if (!rec.hit || h.t < rec.t) { rec = h; cx->st.sphere_accept++; }
    }
    /* the shader reads bvhNodes[0] unconditionally; an empty BVH is skipped here */
    if (sc->n_nodes == 0) return rec;

    pto_node node = sc->nodes[0];
    pto_node stack[PTO_STACK_CAP];
    uint32_t sp = 0, deepest = 0;
    float nearest = 9999.0f;   /* lowered by triangle hits only: compute.wgsl:242, 288 */
    for (;;) {
        const uint32_t contents = node.left;
        if (node.count == 0) {
            cx->st.interior++;
            pto_node c1 = sc->nodes[contents], c2 = sc->nodes[contents + 1u];
            float d1 = hit_aabb(ray, &c1), d2 = hit_aabb(ray, &c2);
            if (d1 > d2) { float td = d1; d1 = d2; d2 = td; pto_node tn = c1; c1 = c2; c2 = tn; }
            if (d1 > nearest) {
                if (sp == 0) break;
                node = stack[--sp];
            } else {
                node = c1;
                if (d2 < nearest) {
                    if (sp >= PTO_STACK_CAP) abort();
                    stack[sp++] = c2;
                    if (sp > deepest) deepest = sp;
                }
            }
        } else {
            for (uint32_t i = 0; i < node.count; i++) {
                const uint32_t ti = sc->tri_idx[i + contents];
                hit_t h = hit_triangle(cx, ray, &sc->triangles[ti], ti);
                if (!h.hit) continue;
                if (!rec.hit || h.t < rec.t) { rec = h; nearest = rec.t; cx->st.tri_accept++; }
            }
            if (sp == 0) break;
            node = stack[--sp];
        }
    }
    if (deepest > cx->st.max_stack) cx->st.max_stack = deepest;
    if (deepest > 15) cx->st.overflow15++;
    return rec;
}

/* compute.wgsl:209-212 + Vulkan cube-map face selection, nearest, clamp-to-edge */
static v3 sky_lookup(ctx_t *cx, v3 d)
{
    const pto_scene *sc = cx->sc;
    cx->st.sky_lookups++;
    if (!sc->sky || sc->sky_size == 0) return V(0.0f, 0.0f, 0.0f);
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    uint32_t face; float scoord, tcoord, ma;
    if (az >= ax && az >= ay) { face = d.z < 0.0f ? 5u : 4u; scoord = d.z < 0.0f ? -d.x : d.x; tcoord = -d.y; ma = az; }
    else if (ay >= ax)        { face = d.y < 0.0f ? 3u : 2u; scoord = d.x; tcoord = d.y < 0.0f ? -d.z : d.z; ma = ay; }
    else                      { face = d.x < 0.0f ? 1u : 0u; scoord = d.x < 0.0f ? d.z : -d.z; tcoord = -d.y; ma = ax; }
    const float size = (float)sc->sky_size;
    const float inv_ma = 1.0f / ma;
    const float fs = fmaf(0.5f, scoord * inv_ma, 0.5f) * size;
    const float ft = fmaf(0.5f, tcoord * inv_ma, 0.5f) * size;
    const uint32_t hi = sc->sky_size - 1u;
    const uint32_t i = (fs >= 0.0f) ? (fs < size ? (uint32_t)fs : hi) : 0u;   /* NaN -> 0 */
    const uint32_t j = (ft >= 0.0f) ? (ft < size ? (uint32_t)ft : hi) : 0u;
    const float *tx = sc->sky + (((size_t)face * sc->sky_size + j) * sc->sky_size + i) * 4u;
    return V(tx[0], tx[1], tx[2]);
}

/* compute.wgsl:408-416 */
static v3 scatter(v3 dir, v3 normal, float seed)
{
    v3 s = vadd(normal, random_unit(dir.x * seed, dir.y * seed));
    if (fabsf(s.x) < 1e-8f && fabsf(s.y) < 1e-8f && fabsf(s.z) < 1e-8f) s = normal;
    return s;
}
/* compute.wgsl:418-423 (fuzz == 0 term dropped) */
static v3 reflect_dir(v3 dir, v3 normal)
{
    const float k = 2.0f * dot3(dir, normal);
    return normalize3(vfma(-k, normal, dir));
}
/* compute.wgsl:425-430 */
static float reflectance(float cosine, float ref_idx)
{
    const float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    const float r0s = r0 * r0;
    const float x = 1.0f - cosine, x2 = x * x;
    const float p5 = (x2 * x2) * x;
    return fmaf(1.0f - r0s, p5, r0s);
}
/* compute.wgsl:432-439 */
static v3 refract_dir(v3 dir, v3 normal, float eta)
{
    const float cos_t = fminf(dot3(vneg(dir), normal), 1.0f);
    const v3 perp = vscale(vfma(cos_t, normal, dir), eta);
    const float par = -sqrtf(fabsf(1.0f - dot3(perp, perp)));
    return vfma(par, normal, perp);
}

static inline uint32_t material_u32(float m)   /* WGSL u32(f32): truncate, saturate, NaN -> 0 */
{
    if (!(m >= 0.0f)) return 0u;
    if (m >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)m;
}

/* compute.wgsl:130-207.  first_prim/first_t receive the primary ray's hit (AOV). */
static v3 ray_color(ctx_t *cx, ray_t initial, uint32_t st[4], uint32_t depth, uint32_t *first_prim, float *first_t)
{
    v3 color = V(1.0f, 1.0f, 1.0f);
    const float seed = pto_hybrid_taus(st);
    uint32_t correction = 0;
    ray_t cur = initial;
    cx->st.paths++;
    for (uint32_t i = 0; i < depth + correction; i++) {
        const hit_t h = hit_scene(cx, cur);
#ifdef PTO_RAY_HOOK
        PTO_RAY_HOOK(cx, i);   /* analysis builds only (tests/csrc/walk_stats.c): after every hitScene, with the bounce index */
#endif
        if (i == 0 && first_prim) { *first_prim = h.hit ? h.prim : 0xFFFFFFFFu; *first_t = h.hit ? h.t : 0.0f; }
        if (!h.hit) { color = vmul(color, sky_lookup(cx, cur.d)); break; }
        cx->st.bounces++;
        v3 bounce;
        const v3 dir = normalize3(cur.d);
        switch (material_u32(h.material)) {
        case 0u:
            bounce = scatter(dir, h.normal, seed);
            if (dot3(bounce, h.normal) <= 0.0f) return vmul(vmul(color, h.att), sky_lookup(cx, cur.d));
            color = vmul(color, h.att);
            break;
        case 1u:
            bounce = reflect_dir(dir, h.normal);
            if (dot3(bounce, h.normal) <= 0.0f) return vmul(vmul(color, h.att), sky_lookup(cx, cur.d));
            color = vmul(color, h.att);
            break;
        case 2u: {
            const float ri = h.front ? (1.0f / 1.5f) : 1.5f;
            const float cos_t = fminf(dot3(vneg(dir), h.normal), 1.0f);
            const float sin_t = sqrtf(fmaf(-cos_t, cos_t, 1.0f));
            const int cannot = ri * sin_t > 1.0f;
            if (cannot || reflectance(cos_t, ri) > hash_rand(h.p.x, h.p.y)) bounce = reflect_dir(dir, h.normal);
            else bounce = refract_dir(dir, h.normal, ri);
            color = vmul(color, h.att);
            break; }
        case 3u: {
            const float dn = dot3(initial.d, h.normal);
            if (i == 0 && dn <= 0.2f && dn >= -0.2f) return h.att;
            bounce = dir;
            correction++;
            break; }
        default:
            bounce = scatter(dir, h.normal, seed);
            color = vmul(color, h.att);
            break;
        }
        cur.o = h.p; cur.d = bounce;
    }
    return color;
}

/* ---- camera frame (compute.wgsl:93-107), shared by every pixel -------------------- */
typedef struct { v3 origin, pixel00, du, dv; } frame_t;
static frame_t camera_frame(const pto_camera *cam, uint32_t W, uint32_t H)
{
    frame_t f;
    const float aspect = (float)W / (float)H;
    const float theta = cam->vfov * 0.017453292519943295f;
    const float h = pto_tanf(theta / 2.0f);
    const float vph = (2.0f * h) * cam->focal;
    const float vpw = aspect * vph;
    const v3 U = vscale(ld3(cam->right), vpw);
    const v3 Vv = vscale(ld3(cam->up), -vph);
    f.du = V(U.x / (float)W, U.y / (float)W, U.z / (float)W);
    f.dv = V(Vv.x / (float)H, Vv.y / (float)H, Vv.z / (float)H);
    f.origin = ld3(cam->origin);
    const v3 ul = vsub(vfma(cam->focal, ld3(cam->forward), f.origin), vscale(vadd(U, Vv), 0.5f));
    f.pixel00 = vadd(ul, vscale(vadd(f.du, f.dv), 0.5f));
    return f;
}

/* compute.wgsl:83-128 for one pixel; adds the per-dispatch mean into accum (RGBA, A += 1). */
static void render_pixel(ctx_t *cx, const frame_t *fr, uint32_t x, uint32_t y, uint32_t time,
                         float *accum, uint32_t *aov_prim, float *aov_t)
{
    const pto_scene *sc = cx->sc;
    uint32_t st[4] = { x, y, x + time, y + time };
    const v3 ploc = vfma((float)y, fr->dv, vfma((float)x, fr->du, fr->pixel00));
    v3 color = V(0.0f, 0.0f, 0.0f);
    const uint32_t spp = sc->settings->spp;
    const size_t pix = (size_t)y * sc->width + x;
    for (uint32_t i = 0; i < spp; i++) {
        const float px = -0.5f + pto_hybrid_taus(st);
        const float py = -0.5f + pto_hybrid_taus(st);
        const v3 sample = vfma(py, fr->dv, vscale(fr->du, px));
        ray_t ray; ray.o = fr->origin; ray.d = vsub(vadd(ploc, sample), fr->origin);
        uint32_t *fp = (i == 0 && aov_prim) ? &aov_prim[pix] : 0;
        float ft_dummy; float *ft = (i == 0 && aov_t) ? &aov_t[pix] : &ft_dummy;
        color = vadd(color, ray_color(cx, ray, st, sc->settings->depth, fp, ft));
    }
    color = vscale(color, 1.0f / (float)spp);
    float *a = accum + pix * 4u;
    a[0] += color.x; a[1] += color.y; a[2] += color.z; a[3] += 1.0f;
}

typedef struct {
    const pto_scene *sc; frame_t fr; uint32_t time, rows, cols;
    float *accum; uint32_t *aov_prim; float *aov_t;
    int tid, nthreads; pto_stats st;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    ctx_t cx; memset(&cx, 0, sizeof cx);
    cx.sc = j->sc; cx.t_min = j->sc->settings->t_min; cx.t_max = j->sc->settings->t_max;
    for (uint32_t y = (uint32_t)j->tid; y < j->rows; y += (uint32_t)j->nthreads)
        for (uint32_t x = 0; x < j->cols; x++)
            render_pixel(&cx, &j->fr, x, y, j->time, j->accum, j->aov_prim, j->aov_t);
    j->st = cx.st;
    return 0;
}

static void stats_add(pto_stats *d, const pto_stats *s)
{
    d->rays += s->rays; d->paths += s->paths; d->interior += s->interior; d->tri_tests += s->tri_tests;
    d->tri_accept += s->tri_accept; d->sphere_tests += s->sphere_tests; d->sky_lookups += s->sky_lookups;
    d->bounces += s->bounces; d->sphere_accept += s->sphere_accept; d->overflow15 += s->overflow15;
    if (s->max_stack > d->max_stack) d->max_stack = s->max_stack;
}

/* One dispatch: every pixel traces settings.spp samples seeded from `time`; the mean is
 * ADDED into accum[h][w][4].  aov_prim/aov_t (optional) receive the primary hit of each
 * pixel's first sample.  stats (optional) is ADDED to.  Returns 0, or -1 on bad input. */
int pto_render(const pto_scene *sc, uint32_t time, float *accum, uint32_t *aov_prim, float *aov_t,
               pto_stats *stats, int nthreads)
{
    if (!sc || !sc->camera || !sc->settings || !accum || sc->width == 0 || sc->height == 0) return -1;
    if (sc->n_nodes && (!sc->nodes || !sc->triangles || !sc->tri_idx)) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    uint32_t rows = sc->height, cols = sc->width;
    if (sc->flags & PTO_FLAG_REFERENCE_DISPATCH) { rows = (rows / 16u) * 16u; cols = (cols / 16u) * 16u; }
    job_t jobs[256]; pthread_t th[256];
    const frame_t fr = camera_frame(sc->camera, sc->width, sc->height);
    for (int t = 0; t < nthreads; t++) {
        jobs[t].sc = sc; jobs[t].fr = fr; jobs[t].time = time; jobs[t].rows = rows; jobs[t].cols = cols;
        jobs[t].accum = accum; jobs[t].aov_prim = aov_prim; jobs[t].aov_t = aov_t;
        jobs[t].tid = t; jobs[t].nthreads = nthreads; memset(&jobs[t].st, 0, sizeof(pto_stats));
    }
    if (nthreads == 1) worker(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; t++) if (pthread_create(&th[t], 0, worker, &jobs[t])) return -1;
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], 0);
    }
    if (stats) for (int t = 0; t < nthreads; t++) stats_add(stats, &jobs[t].st);
    return 0;
}

/* Closest hit of one explicit ray (analytic intersection tests). out = {hit, prim, t, p[3], n[3], front}. */
int pto_trace_ray(const pto_scene *sc, const float o[3], const float d[3], uint32_t *hit, uint32_t *prim,
                  float *t, float p[3], float n[3], uint32_t *front)
{
    if (!sc || !sc->settings) return -1;
    ctx_t cx; memset(&cx, 0, sizeof cx);
    cx.sc = sc; cx.t_min = sc->settings->t_min; cx.t_max = sc->settings->t_max;
    ray_t r; r.o = ld3(o); r.d = ld3(d);
    hit_t h = hit_scene(&cx, r);
    *hit = (uint32_t)h.hit; *prim = h.hit ? h.prim : 0xFFFFFFFFu; *t = h.t;
    p[0] = h.p.x; p[1] = h.p.y; p[2] = h.p.z; n[0] = h.normal.x; n[1] = h.normal.y; n[2] = h.normal.z;
    *front = (uint32_t)h.front;
    return 0;
}

float pto_hit_aabb(const float o[3], const float d[3], const float mn[3], const float mx[3])
{
    pto_node n; memset(&n, 0, sizeof n);
    memcpy(n.mn, mn, 12); memcpy(n.mx, mx, 12);
    ray_t r; r.o = ld3(o); r.d = ld3(d);
    return hit_aabb(r, &n);
}

void pto_sky_lookup(const float *sky, uint32_t size, const float d[3], float rgb[3])
{
    pto_scene sc; memset(&sc, 0, sizeof sc); sc.sky = sky; sc.sky_size = size;
    ctx_t cx; memset(&cx, 0, sizeof cx); cx.sc = &sc;
    v3 c = sky_lookup(&cx, ld3(d));
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

void pto_random_unit(float sx, float sy, float out[3])
{
    v3 r = random_unit(sx, sy); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float pto_hash_rand(float sx, float sy) { return hash_rand(sx, sy); }

/* ---- BVH build: src/scene/bvh.rs:34-162 ------------------------------------------- */
typedef struct { pto_node *nodes; uint32_t *idx; const pto_triangle *tris; uint32_t used; } bvh_t;

static inline float centroid_axis(const pto_triangle *t, int axis)   /* model.rs:42-48 */
{
    return (t->a[axis] + t->b[axis] + t->c[axis]) / 3.0f;
}

static void update_bounds(bvh_t *b, uint32_t ni)   /* bvh.rs:68-89 */
{
    pto_node *n = &b->nodes[ni];
    for (uint32_t i = 0; i < n->count; i++) {
        const pto_triangle *t = &b->tris[b->idx[n->left + i]];
        const float *vs[3] = { t->a, t->b, t->c };
        for (int k = 0; k < 3; k++)
            for (int ax = 0; ax < 3; ax++) {
                /* f32::min/max: a NaN vertex is ignored; on +0/-0 ties the earlier value stays
                 * (which zero Rust's minnum lowering returns is not specified) */
                if (vs[k][ax] < n->mn[ax]) n->mn[ax] = vs[k][ax];
                if (vs[k][ax] > n->mx[ax]) n->mx[ax] = vs[k][ax];
            }
    }
}

static void subdivide(bvh_t *b, uint32_t ni)   /* bvh.rs:91-162 */
{
    const pto_node node = b->nodes[ni];
    if (node.count <= 2) return;
    const float ext[3] = { node.mx[0] - node.mn[0], node.mx[1] - node.mn[1], node.mx[2] - node.mn[2] };
    int axis = 0;
    if (ext[1] > ext[axis]) axis = 1;
    if (ext[2] > ext[axis]) axis = 2;
    const float split = node.mn[axis] + ext[axis] / 2.0f;
    uint32_t i = node.left, j = node.left + node.count - 1u;
    while (i < j) {
        if (centroid_axis(&b->tris[b->idx[i]], axis) < split) i++;
        else { uint32_t tmp = b->idx[i]; b->idx[i] = b->idx[j]; b->idx[j] = tmp; j--; }
    }
    const uint32_t left_count = i - node.left;
    if (left_count == 0 || left_count == node.count) return;
    const uint32_t li = b->used++, ri = b->used++;
    b->nodes[li].left = node.left; b->nodes[li].count = left_count;
    b->nodes[ri].left = i;         b->nodes[ri].count = node.count - left_count;
    b->nodes[ni].left = li;        b->nodes[ni].count = 0;
    update_bounds(b, li); update_bounds(b, ri);
    subdivide(b, li); subdivide(b, ri);
}

/* nodes_out must hold 2*n_tris-1 nodes, idx_out n_tris entries.  Returns nodes used
 * (0 for an empty mesh), or -1 on bad arguments. */
int64_t pto_bvh_build(const pto_triangle *tris, uint32_t n_tris, pto_node *nodes_out, uint32_t *idx_out)
{
    if (n_tris == 0) return 0;
    if (!tris || !nodes_out || !idx_out) return -1;
    const size_t cap = (size_t)n_tris * 2u - 1u;
    for (size_t k = 0; k < cap; k++) {   /* Node::default, bvh.rs:16-25 */
        for (int ax = 0; ax < 3; ax++) { nodes_out[k].mn[ax] = 3.40282347e+38f; nodes_out[k].mx[ax] = -3.40282347e+38f; }
        nodes_out[k].left = 0; nodes_out[k].count = 0;
    }
    for (uint32_t k = 0; k < n_tris; k++) idx_out[k] = k;
    bvh_t b = { nodes_out, idx_out, tris, 0 };
    b.nodes[0].left = 0; b.nodes[0].count = n_tris; b.used = 1;
    update_bounds(&b, 0);
    subdivide(&b, 0);
    return (int64_t)b.used;
}

/* Plane -> two triangles: src/scene/plane.rs:16-45.  out receives 2 records of 112 B. */
void pto_plane_triangles(const float q[3], const float u[3], const float v[3], const float albedo[3],
                         uint32_t material, pto_triangle out[2])
{
    const v3 Q = ld3(q), U = ld3(u), Vv = ld3(v);
    /* cgmath cross + normalize: plain products and sums, v * (1/len)?  cgmath's
     * normalize is self * (1/magnitude); magnitude = sqrt(x*x+y*y+z*z) summed in order. */
    const v3 c = V(U.y * Vv.z - U.z * Vv.y, U.z * Vv.x - U.x * Vv.z, U.x * Vv.y - U.y * Vv.x);
    const float len = sqrtf(c.x * c.x + c.y * c.y + c.z * c.z);
    const float inv = 1.0f / len;
    const v3 n = vscale(c, inv);
    const v3 qu = vadd(Q, U), qv = vadd(Q, Vv), quv = vadd(qu, Vv);
    memset(out, 0, 2 * sizeof(pto_triangle));
    const v3 pa[2] = { Q, quv };
    for (int k = 0; k < 2; k++) {
        pto_triangle *t = &out[k];
        t->a[0] = pa[k].x; t->a[1] = pa[k].y; t->a[2] = pa[k].z;
        t->b[0] = qu.x; t->b[1] = qu.y; t->b[2] = qu.z;
        t->c[0] = qv.x; t->c[1] = qv.y; t->c[2] = qv.z;
        for (int ax = 0; ax < 3; ax++) {
            const float nv = ax == 0 ? n.x : ax == 1 ? n.y : n.z;
            t->na[ax] = nv; t->nb[ax] = nv; t->nc[ax] = nv; t->albedo[ax] = albedo[ax];
        }
        t->material = material;
    }
}

/* ---- equirectangular -> cube map: shaders/equirectangular.wgsl:12-81 --------------- */
static const float FACES[6][3][3] = {   /* forward, up, right */
    { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, -1 } },
    { { -1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } },
    { { 0, -1, 0 }, { 0, 0, 1 }, { 1, 0, 0 } },
    { { 0, 1, 0 }, { 0, 0, -1 }, { 1, 0, 0 } },
    { { 0, 0, 1 }, { 0, 1, 0 }, { 1, 0, 0 } },
    { { 0, 0, -1 }, { 0, 1, 0 }, { -1, 0, 0 } },
};

/* src: RGBA32F src_h x src_w; dst: 6 x size x size RGBA32F.  Texel loads outside the
 * source (eq_uv can reach 1.0) return zeros, as WGSL textureLoad may. */
void pto_equirect_to_cube(const float *src, uint32_t src_w, uint32_t src_h, float *dst, uint32_t size)
{
    const float fsz = (float)size;
    for (uint32_t f = 0; f < 6; f++)
        for (uint32_t y = 0; y < size; y++)
            for (uint32_t x = 0; x < size; x++) {
                const float cu = fmaf((float)x / fsz, 2.0f, -1.0f), cv = fmaf((float)y / fsz, 2.0f, -1.0f);
                const v3 fw = ld3(FACES[f][0]), up = ld3(FACES[f][1]), rt = ld3(FACES[f][2]);
                const v3 sph = normalize3(vfma(cv, up, vfma(cu, rt, fw)));
                const float eu = fmaf(pto_atan2f(sph.z, sph.x), 0.1591f, 0.5f);
                const float ev = fmaf(pto_asinf(sph.y), 0.3183f, 0.5f);
                const float fx = eu * (float)src_w, fy = ev * (float)src_h;
                const int32_t ix = (int32_t)fx, iy = (int32_t)fy;   /* vec2<i32>(): truncation */
                float *o = dst + (((size_t)f * size + y) * size + x) * 4u;
                if (ix >= 0 && iy >= 0 && (uint32_t)ix < src_w && (uint32_t)iy < src_h)
                    memcpy(o, src + ((size_t)iy * src_w + (uint32_t)ix) * 4u, 16);
                else memset(o, 0, 16);
            }
}
