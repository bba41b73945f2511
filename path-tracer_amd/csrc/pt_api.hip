// pt_api.hip -- implementation of the C ABI in include/ptrt.h: context, uploads, launches.
//
// Device data layout (all hipMalloc, lives for the context):
//   accum     width*height float4   running RGB sums, .w = frame count
//   spheres   3 float4 per sphere   {center, radius} {albedo, material} {1/radius, 0, 0, 0}
//   walk table: one array of 64-byte records, everything a traversal step reads, addressed in 32-byte units
//     sibling pair (l, l+1)         {lmin.x rmin.x lmin.y rmin.y} {lmin.z rmin.z lmax.x rmax.x}
//                                   {lmax.y rmax.y lmax.z rmax.z} {l token, r token, 0, 0}: the pair interleaved
//                                   component by component (two-wide slab tests), at the place of nodes l
//                                   and l+1; the array is stored one node late so that every pair
//                                   (odd, even index) is one 64-byte aligned block
//     triangle slot k               {a, original index} {b-a, material} {c-a, 0} {unused} behind the pairs, slot =
//                                   position in the BVH's index permutation, so a leaf's triangles are contiguous
//   node_meta uint2 per node        (table position of the first triangle, count), read only for leaves with > 7 triangles
//   tri_nrm   3 float4 per slot     vertex normals, touched once per closest hit
//   sky       6*size*size float4    cube faces +X,-X,+Y,-Y,+Z,-Z
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/ptrt.h"
#include "pt_kernels.h"

struct pt_ctx {
    int device = 0;
    uint32_t width = 0, height = 0;
    hipStream_t stream = nullptr;
    uint32_t flags = 0;
    std::string err;

    PtCamera camera{};
    PtSettings settings{1, 32, 0.0001f, 1000.0f};  // Renderer::new defaults, renderer.rs:411-416
    bool have_camera = false;

    float4 *accum_owned = nullptr, *accum = nullptr;
    uint32_t *aov_prim = nullptr;
    float *aov_t = nullptr;
    unsigned long long *counters = nullptr;

    float4 *spheres = nullptr;
    uint32_t n_spheres = 0;
    bool has_gizmo = false, mesh_has_gizmo = false;
    uint64_t paths = 0;  // rayColor calls since the last reset (every pixel runs every sample)

    float4 *nodes_alloc = nullptr;
    uint2 *node_meta = nullptr;
    float4 *tri_nrm = nullptr;
    uint32_t n_nodes = 0, root_token = 0, bvh_depth = 0, tri_base = 0;
    float4 root_pair[4] = {};
    uint32_t walk_depth = 0;  // deeper of the caller's tree and the second tree's pair form (the pooled kernel's stack bound)
    // the guard tree (PtDeviceScene::guard_on ...), if pt_set_mesh built one
    bool guard_on = false;
    float guard_center[3] = {}, guard_radius = 0.0f, guard_reach = 0.0f, guard_margin = 0.0f;
    uint32_t guard_root = 0, guard_nodes = 0, guard_depth = 0, guard_quads = 0;
    float guard_pad_median = 0.0f;

    float4 *sky = nullptr;
    uint32_t sky_size = 0;

    uint32_t *gstack = nullptr;
    size_t gstack_bytes = 0;
    uint32_t gstack_cap = 0;
    uint32_t *pool_gstack = nullptr;
    size_t pool_gstack_bytes = 0;
    uint32_t n_cus = 256;
    uint32_t n_tri_slots = 0;

    // work-item planning (see plan_kernel)
    uint32_t *tile_cost = nullptr, *tile_sorted = nullptr, *next_item = nullptr;
    uint2 *tile_items = nullptr;
    uint32_t *tile_sky = nullptr;  // per tile: sees only the sky (PtSchedule::tile_sky)
    // planned launches with sky-only tiles: the megakernel on main_stream, the tiles' loop beside it on side_stream
    // (the GPU's pair, g_streams: borrowed, never destroyed by a context)
    hipStream_t main_stream = nullptr, side_stream = nullptr;
    hipEvent_t fork_event = nullptr, start_event = nullptr, join_event = nullptr, main_event = nullptr;
    uint4 *items = nullptr;
    float4 *partial = nullptr;
    uint32_t item_cap = 0, n_tiles = 0, resident_waves = 5120;
    bool cost_valid = false;

    void *scratch = nullptr;  // read-back staging for resolve
    size_t scratch_bytes = 0;

    uint64_t frames = 0;
    bool last_static = false;  // the last render call took its scheduling memory from the static scratch array

    // The plan (item table) of the last planned launch, and what it was made for: an identical launch runs it
    // again instead of planning anew (nothing that decides a tile's cost has changed).
    uint64_t scene_generation = 0;  // bumped by everything that can change costs or the plan's buffers
    struct PlanKey {
        uint64_t generation = ~0ull;
        uint32_t n_frames = 0, wanted_items = 0, max_chunk = 0, min_chunk = 0;
        bool use_pool = false;
        bool operator==(const PlanKey &o) const
        {
            return generation == o.generation && n_frames == o.n_frames && wanted_items == o.wanted_items && max_chunk == o.max_chunk &&
                   min_chunk == o.min_chunk && use_pool == o.use_pool;
        }
    } plan_key;
    uint32_t plans_for_key = 0;  // plans made in a row for plan_key (the second one rests on a full measurement)

    // ProgressiveRendering (renderer.rs:613-646) when pt_set_window has enabled it
    uint32_t window_size = 0;       // sample_size; 0: off
    uint32_t ready_samples = 0;     // renderer.rs:422
    uint32_t window_shown = 0;      // textures the last frame's copy pass averaged
    uint64_t window_seq = 0;        // frames pushed into the ring so far
    uint32_t *window_ring = nullptr;  // window_size frames of packed rgba8unorm texels
    float4 *window_frame = nullptr;   // the frame being rendered (zero between frames)
};

// ---- static scratch for pt_render_frames_static ---------------------------------------------------------------
// Work-item table, partial sums and deep-stack spill space of a launch, for callers that cannot allocate (HIP graph
// capture, serving loops).  One copy per GPU, held from the moment the library's code is loaded onto that GPU until
// the process ends; every context on a GPU shares it (calls that overlap on one GPU race).  In host code the array's
// name is only a handle for hipGetSymbolAddress: g_static[device].ptr is the device address to use.
#define PT_STATIC_SCRATCH_BYTES (192ull << 20)
static __device__ __align__(256) unsigned char scratch[PT_STATIC_SCRATCH_BYTES];

namespace {

constexpr int kMaxDevices = 64;
struct StaticScratch {
    std::mutex lock;
    unsigned char *ptr = nullptr;     // device address of this GPU's copy of `scratch`
    const pt_ctx *last_user = nullptr;  // context whose plan the array holds
} g_static[kMaxDevices];

// Once per device, with that device current (pt_create calls it).
hipError_t static_scratch_init(int device)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(g_static[device].lock);
    if (g_static[device].ptr) return hipSuccess;
    void *p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p || (reinterpret_cast<uintptr_t>(p) & 255u)) return hipErrorInvalidSymbol;
    g_static[device].ptr = static_cast<unsigned char *>(p);
    return hipSuccess;
}

// ---- one stream pair per GPU ---------------------------------------------------------------------------------------
// Planned launches run the megakernel on `main` and the sky-only tiles' loop beside it on `side`; both are forked from
// and joined back into the caller's stream by events of the context.  The pair belongs to the device, not to a context:
// it is created once, by the first pt_create on that GPU, tested once (pt_streams_run_side_by_side: do the two reach
// the GPU through different hardware queues?) and then kept until the process ends, whatever contexts come and go.
// A runtime hands its few hardware queues (4 per process by default) to streams as they are created and destroyed, so
// a pair per context meant up to five stream creations per pt_create and a queue assignment that depended on the
// order in which a host made and dropped its renderers; one fixed pair has one assignment, checked once, and
// pt_stats reports the check's verdict.  Both streams have the default priority: a stream of another priority reaches
// the GPU through a queue that the command processor serves strictly before (or after) the caller's, and a wait of the
// preferred queue on an event of the other one is then a priority inversion.  Launches of several contexts simply
// queue up on the pair (they could not overlap anyway: a megakernel fills the GPU).
struct DeviceStreams {
    std::mutex lock;
    bool tried = false;          // creation and test have run (successfully or not)
    hipStream_t main = nullptr;  // null: creation failed, launches stay on the caller's stream
    hipStream_t side = nullptr;  // null: no stream was found that runs beside `main`; the loop runs behind the megakernel
    uint32_t *words = nullptr;   // 4 device words for the test kernels
} g_streams[kMaxDevices];

// Once per device, with that device current (pt_create calls it).  Failures are not errors: without the pair every
// launch runs on the caller's stream alone.
void device_streams_init(int device)
{
    DeviceStreams &ds = g_streams[device];
    std::lock_guard<std::mutex> g(ds.lock);
    if (ds.tried) return;
    ds.tried = true;
    if (hipMalloc((void **)&ds.words, 4 * sizeof(uint32_t)) != hipSuccess) { ds.words = nullptr; return; }
    if (hipStreamCreateWithFlags(&ds.main, hipStreamNonBlocking) != hipSuccess) { ds.main = nullptr; return; }
    // A candidate that shares the main stream's queue is set aside -- and kept until the search is over, so that the
    // next stream created is given another queue -- and the next one tried.
    hipStream_t rejected[4];
    int n_rejected = 0;
    for (int attempt = 0; attempt < 4 && !ds.side; attempt++) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) break;
        bool beside = false;
        if (pt_streams_run_side_by_side(ds.main, cand, ds.words, &beside) == hipSuccess && beside) ds.side = cand;
        else rejected[n_rejected++] = cand;
    }
    for (int k = 0; k < n_rejected; k++) (void)hipStreamDestroy(rejected[k]);
}

thread_local std::string g_create_error;

int fail(pt_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

// for the catch-all guards: a message short enough for the string's in-place buffer, so that reporting an
// allocation failure does not allocate
int fail_static(pt_ctx *ctx, int code, const char *msg) noexcept
{
    try {
        if (ctx) ctx->err.assign(msg);
        else g_create_error.assign(msg);
    } catch (...) {
    }
    return code;
}

// The calling thread's current device is put back when an entry point returns (a host that drives several GPUs
// from one thread must not find its later allocations redirected to the context's device).
struct DeviceGuard {
    int saved = -1;
    explicit DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

#define PT_HIP(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(ctx, e_ == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_HIP, "%s: %s", #call, \
                        hipGetErrorString(e_));                                                        \
    } while (0)

// (a status left behind by an earlier failed HIP call of anybody in the process would otherwise be reported by this
// entry point's first launch check: it is dropped on the way in)
#define PT_ENTER(ctx)                                                 \
    if (!(ctx)) return fail(nullptr, PT_ERR_INVALID, "null context"); \
    DeviceGuard device_guard_;                                        \
    (void)hipGetLastError();                                          \
    PT_HIP(ctx, hipSetDevice((ctx)->device))

template <typename T>
int dev_alloc(pt_ctx *ctx, T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) return PT_OK;
    PT_HIP(ctx, hipMalloc((void **)p, count * sizeof(T)));
    return PT_OK;
}
template <typename T>
void dev_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

int ensure_scratch(pt_ctx *ctx, size_t bytes)
{
    if (ctx->scratch_bytes >= bytes) return PT_OK;
    dev_free(ctx->scratch);
    ctx->scratch_bytes = 0;
    PT_HIP(ctx, hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return PT_OK;
}

void free_mesh(pt_ctx *ctx)
{
    dev_free(ctx->nodes_alloc);
    dev_free(ctx->node_meta);
    dev_free(ctx->tri_nrm);
    ctx->n_nodes = 0;
    ctx->n_tri_slots = 0;
    ctx->mesh_has_gizmo = false;
    ctx->root_token = 0;
    ctx->bvh_depth = 0;
    ctx->walk_depth = 0;
    ctx->guard_on = false;
    ctx->guard_nodes = ctx->guard_depth = 0;
    ctx->guard_quads = 0;
}

PtDeviceScene device_scene(const pt_ctx *ctx)
{
    PtDeviceScene sc{};
    sc.spheres = ctx->spheres;
    sc.n_spheres = ctx->n_spheres;
    sc.nodes = ctx->nodes_alloc ? ctx->nodes_alloc + 2 : nullptr;  // +1 node: see layout note above
    sc.node_meta = ctx->node_meta;
    sc.n_nodes = ctx->n_nodes;
    sc.root_token = ctx->root_token;
    for (int k = 0; k < 4; k++) sc.root_pair[k] = ctx->root_pair[k];
    sc.tri_base = ctx->tri_base;
    sc.tri_nrm = ctx->tri_nrm;
    sc.sky = ctx->sky;
    sc.sky_size = ctx->sky_size;
    sc.tri_materials = (ctx->flags & PT_FLAG_TRIANGLE_MATERIALS) ? 1u : 0u;
    sc.t_min = ctx->settings.t_min;
    sc.t_max = ctx->settings.t_max;
    sc.guard_on = (ctx->guard_on && !(ctx->flags & PT_FLAG_REFERENCE_WALK)) ? 1u : 0u;  // (the flag also works on a mesh that has the tree)
    for (int k = 0; k < 3; k++) sc.guard_center[k] = ctx->guard_center[k];
    sc.guard_radius = ctx->guard_radius;
    sc.guard_reach = ctx->guard_reach;
    sc.guard_margin = ctx->guard_margin;
    sc.guard_root = ctx->guard_root;
    sc.guard_quads = sc.guard_on ? ctx->guard_quads : 0u;
    return sc;
}

// Screen rectangles of the root's two child boxes (PtTraceParams::cull).  A camera ray of pixel (px, py), jittered by
// at most half a pixel, has the direction pixel00 + px du + py dv - origin (compute.wgsl:109-119); a point P of a box is
// seen along (px, py) = (a / c, b / c) where (a, b, c) solves [du dv pixel00-origin] (a, b, c)^T = P - origin, c > 0.
// The image of a convex box is the hull of its corners' images, so a pixel more than the jitter outside their
// bounding rectangle sees nothing of the box.  The margin: two pixels for the jitter, plus what the kernels' float
// arithmetic can move a ray or a box by, in pixels.  The direction is evaluated in float as pixel00 + x du + y dv +
// jitter - origin, every step rounding to an ulp of its largest intermediate M = |pixel00| + W |du| + H |dv| + |origin|:
// an error of a few 2^-24 M in world units = that over |du| (or |dv|) in pixels -- negligible near the origin at an
// ordinary field of view, more than two pixels for a narrow one far from it (5 degrees at 1080p, 1000 units out).  The
// slab test evaluates (bound - origin) / direction in float: it sees each box face up to a few ulps of B = the largest
// |bound| or |origin| away from where it is, which moves the face's image by that over (c |du|) pixels, c the corner's
// depth in units of |pixel00 - origin|.  Sixteen ulps each (the evaluations have at most five roundings) bound both;
// if the sum exceeds 64 pixels, or a box has a corner at or behind the camera, there is no statement for that box.
void camera_cull_rects(const pt_ctx *ctx, const pt_frame &fr, uint32_t cull[8])
{
    for (int k = 0; k < 2; k++) { cull[4 * k] = 0; cull[4 * k + 1] = 0xFFFFFFFFu; cull[4 * k + 2] = 0; cull[4 * k + 3] = 0xFFFFFFFFu; }
    if (ctx->n_nodes == 0 || (ctx->root_token & PT_TOKEN_LEAF)) return;
    const double m[3][3] = {{fr.du.x, fr.dv.x, (double)fr.pixel00.x - fr.origin.x},
                            {fr.du.y, fr.dv.y, (double)fr.pixel00.y - fr.origin.y},
                            {fr.du.z, fr.dv.z, (double)fr.pixel00.z - fr.origin.z}};
    const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                       m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return;
    const float4 *q = ctx->root_pair;  // {lmin.x rmin.x lmin.y rmin.y} {lmin.z rmin.z lmax.x rmax.x} {lmax.y rmax.y lmax.z rmax.z}
    auto len3 = [](double x, double y, double z) { return std::sqrt(x * x + y * y + z * z); };
    const double eps = 1.0 / 16777216.0;  // 2^-24: half an ulp of a float in [1, 2)
    const double step = std::min(len3(fr.du.x, fr.du.y, fr.du.z), len3(fr.dv.x, fr.dv.y, fr.dv.z));  // world units per pixel
    const double lo_o = len3(fr.origin.x, fr.origin.y, fr.origin.z);
    const double M = len3(fr.pixel00.x, fr.pixel00.y, fr.pixel00.z) + (double)ctx->width * len3(fr.du.x, fr.du.y, fr.du.z) +
                     (double)ctx->height * len3(fr.dv.x, fr.dv.y, fr.dv.z) + lo_o;
    if (!(step > 0.0) || !std::isfinite(M)) return;
    double cmin[2] = {1e300, 1e300}, B[2] = {lo_o, lo_o};
    for (int k = 0; k < 2; k++) {
        const double lo[3] = {k ? q[0].y : q[0].x, k ? q[0].w : q[0].z, k ? q[1].y : q[1].x};
        const double hi[3] = {k ? q[1].w : q[1].z, k ? q[2].y : q[2].x, k ? q[2].w : q[2].z};
        double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
        bool ok = true;
        for (int c = 0; c < 8 && ok; c++) {
            const double v[3] = {((c & 1) ? hi[0] : lo[0]) - fr.origin.x, ((c & 2) ? hi[1] : lo[1]) - fr.origin.y, ((c & 4) ? hi[2] : lo[2]) - fr.origin.z};
            // Cramer's rule
            const double a = (v[0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (v[1] * m[2][2] - m[1][2] * v[2]) + m[0][2] * (v[1] * m[2][1] - m[1][1] * v[2])) / det;
            const double b = (m[0][0] * (v[1] * m[2][2] - m[1][2] * v[2]) - v[0] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) + m[0][2] * (m[1][0] * v[2] - v[1] * m[2][0])) / det;
            const double cc = (m[0][0] * (m[1][1] * v[2] - v[1] * m[2][1]) - m[0][1] * (m[1][0] * v[2] - v[1] * m[2][0]) + v[0] * (m[1][0] * m[2][1] - m[1][1] * m[2][0])) / det;
            if (!(cc > 1e-6) || !std::isfinite(a) || !std::isfinite(b)) { ok = false; break; }  // at or behind the camera
            cmin[k] = std::min(cmin[k], cc);
            B[k] = std::max(B[k], std::max(std::max(std::fabs(v[0] + fr.origin.x), std::fabs(v[1] + fr.origin.y)), std::fabs(v[2] + fr.origin.z)));
            const double px = a / cc, py = b / cc;
            if (!std::isfinite(px) || !std::isfinite(py)) { ok = false; break; }
            x0 = std::min(x0, px); x1 = std::max(x1, px); y0 = std::min(y0, py); y1 = std::max(y1, py);
        }
        if (!ok) continue;
        const double W = ctx->width, H = ctx->height;
        const double margin = 2.0 + 16.0 * eps * (M / step + B[k] / (step * cmin[k]));
        if (!(margin <= 64.0)) continue;  // (also when it is not a number)
        const double ax0 = std::floor(x0 - margin), ax1 = std::ceil(x1 + margin), ay0 = std::floor(y0 - margin), ay1 = std::ceil(y1 + margin);
        if (ax1 < 0.0 || ay1 < 0.0 || ax0 > W - 1.0 || ay0 > H - 1.0) {  // entirely off screen
            cull[4 * k] = 0xFFFFFFFFu; cull[4 * k + 1] = 0; cull[4 * k + 2] = 0xFFFFFFFFu; cull[4 * k + 3] = 0;
            continue;
        }
        const uint32_t ix0 = (uint32_t)std::max(ax0, 0.0), ix1 = (uint32_t)std::min(ax1, W - 1.0);
        const uint32_t iy0 = (uint32_t)std::max(ay0, 0.0), iy1 = (uint32_t)std::min(ay1, H - 1.0);
        cull[4 * k] = ix0; cull[4 * k + 1] = ix1 - ix0; cull[4 * k + 2] = iy0; cull[4 * k + 3] = iy1 - iy0;
    }
}

int ensure_gstack(pt_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->gstack_bytes) return PT_OK;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // a launch in flight may still use the old buffer
    dev_free(ctx->gstack);
    ctx->gstack_bytes = 0;
    PT_HIP(ctx, hipMalloc((void **)&ctx->gstack, bytes));
    ctx->gstack_bytes = bytes;
    return PT_OK;
}

// Launch heuristics are constants.  Only a -DPT_TUNING build (make tuning; used by tools/sweep_*.sh) lets PTRT_*
// environment variables override them, read at every launch.
#ifdef PT_TUNING
inline uint32_t env_u32(const char *name, uint32_t fallback)
{
    const char *e = getenv(name);
    return e ? (uint32_t)strtoul(e, nullptr, 10) : fallback;
}
inline const char *env_str(const char *name) { return getenv(name); }
#else
inline uint32_t env_u32(const char *, uint32_t fallback) { return fallback; }
inline const char *env_str(const char *) { return nullptr; }
#endif

inline float u2f(uint32_t u)
{
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

}  // namespace

extern "C" {

const char *pt_last_error(const pt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static int pt_create_impl(int device, uint32_t width, uint32_t height, pt_ctx **out)
{
    if (!out) return fail(nullptr, PT_ERR_INVALID, "pt_create: null output pointer");
    *out = nullptr;
    DeviceGuard device_guard_;
    if (width == 0 || height == 0) return fail(nullptr, PT_ERR_INVALID, "pt_create: zero image size");
    if ((uint64_t)width * height > (1ull << 30)) return fail(nullptr, PT_ERR_LIMIT, "pt_create: image larger than 2^30 pixels");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(nullptr, PT_ERR_HIP, "pt_create: no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count || device >= kMaxDevices) return fail(nullptr, PT_ERR_INVALID, "pt_create: device %d out of range (%d present)", device, count);
    pt_ctx *ctx = new (std::nothrow) pt_ctx();
    if (!ctx) return fail(nullptr, PT_ERR_NOMEM, "pt_create: out of host memory");
    ctx->device = device;
    ctx->width = width;
    ctx->height = height;
    const size_t n = (size_t)width * height;
    int rc = PT_OK;
    auto step = [&](hipError_t he, const char *what) {
        if (rc == PT_OK && he != hipSuccess)
            rc = fail(nullptr, he == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_HIP, "pt_create: %s: %s", what, hipGetErrorString(he));
    };
    step(hipSetDevice(device), "hipSetDevice");
    if (rc == PT_OK) step(static_scratch_init(device), "address of the static scratch array");
    if (rc == PT_OK) step(hipMalloc((void **)&ctx->accum_owned, n * sizeof(float4)), "accumulation buffer");
    if (rc == PT_OK) step(hipMalloc((void **)&ctx->aov_prim, n * sizeof(uint32_t)), "aov buffer");
    if (rc == PT_OK) step(hipMalloc((void **)&ctx->aov_t, n * sizeof(float)), "aov buffer");
    if (rc == PT_OK) step(hipMalloc((void **)&ctx->counters, PT_COUNTER_STRIPES * PT_COUNTER_PITCH * sizeof(unsigned long long)), "counters");
    if (rc == PT_OK) step(hipMalloc((void **)&ctx->spheres, 3 * PT_MAX_SPHERES * sizeof(float4)), "sphere buffer");
    const size_t n_tiles = pt_trace_num_tiles(width, height);
    ctx->n_tiles = (uint32_t)n_tiles;
    if (rc == PT_OK) step(hipMalloc((void **)&ctx->tile_cost, n_tiles * sizeof(uint32_t)), "tile cost");
    if (rc == PT_OK) step(hipMalloc((void **)&ctx->next_item, 256), "item counter");
    if (rc == PT_OK) step(hipMemset(ctx->next_item, 0, 256), "clear");
    if (rc == PT_OK) step(hipMalloc((void **)&ctx->tile_sorted, n_tiles * sizeof(uint32_t)), "tile order");
    if (rc == PT_OK) step(hipMalloc((void **)&ctx->tile_items, n_tiles * sizeof(uint2)), "tile items");
    if (rc == PT_OK) step(hipMalloc((void **)&ctx->tile_sky, n_tiles * sizeof(uint32_t)), "tile flags");
    if (rc == PT_OK) step(hipMemset(ctx->tile_sky, 0, n_tiles * sizeof(uint32_t)), "tile flags");
    if (rc == PT_OK) step(hipMemset(ctx->tile_cost, 0, n_tiles * sizeof(uint32_t)), "clear");
    if (rc == PT_OK) {
        // the GPU's stream pair (created and tested by the first context on this GPU, shared by all of them)
        device_streams_init(device);
        ctx->main_stream = g_streams[device].main;
        ctx->side_stream = g_streams[device].main ? g_streams[device].side : nullptr;
        // (no pair that runs side by side: the loop runs on the launch stream behind the megakernel, as without sky-only tiles)
    }
    if (rc == PT_OK) step(hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming), "event");
    if (rc == PT_OK) step(hipEventCreateWithFlags(&ctx->join_event, hipEventDisableTiming), "event");
    if (rc == PT_OK) step(hipEventCreateWithFlags(&ctx->main_event, hipEventDisableTiming), "event");
    if (rc == PT_OK) step(hipEventCreateWithFlags(&ctx->start_event, hipEventDisableTiming), "event");
    if (rc == PT_OK) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        {
            ctx->resident_waves = (uint32_t)prop.multiProcessorCount * 4u * PT_TRACE_WAVES_PER_EU;
            ctx->n_cus = (uint32_t)prop.multiProcessorCount;
        }
    }
    if (rc == PT_OK) step(hipMemset(ctx->accum_owned, 0, n * sizeof(float4)), "clear");
    if (rc == PT_OK) step(hipMemset(ctx->aov_prim, 0xFF, n * sizeof(uint32_t)), "clear");
    if (rc == PT_OK) step(hipMemset(ctx->aov_t, 0, n * sizeof(float)), "clear");
    if (rc == PT_OK) step(hipMemset(ctx->counters, 0, PT_COUNTER_STRIPES * PT_COUNTER_PITCH * sizeof(unsigned long long)), "clear");
    if (rc != PT_OK) {
        pt_destroy(ctx);
        return rc;
    }
    ctx->accum = ctx->accum_owned;
    *out = ctx;
    return PT_OK;
}

void pt_destroy(pt_ctx *ctx)
{
    if (!ctx) return;
    DeviceGuard device_guard_;
    if (ctx->device >= 0 && ctx->device < kMaxDevices) {
        std::lock_guard<std::mutex> g(g_static[ctx->device].lock);
        if (g_static[ctx->device].last_user == ctx) g_static[ctx->device].last_user = nullptr;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    else (void)hipDeviceSynchronize();
    // (the GPU's stream pair is not the context's to destroy; every launch joined it back into the context's stream)
    if (ctx->main_event) (void)hipEventDestroy(ctx->main_event);
    if (ctx->start_event) (void)hipEventDestroy(ctx->start_event);
    if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
    if (ctx->join_event) (void)hipEventDestroy(ctx->join_event);
    dev_free(ctx->window_ring);
    dev_free(ctx->window_frame);
    dev_free(ctx->accum_owned);
    dev_free(ctx->aov_prim);
    dev_free(ctx->aov_t);
    dev_free(ctx->counters);
    dev_free(ctx->spheres);
    free_mesh(ctx);
    dev_free(ctx->sky);
    dev_free(ctx->gstack);
    dev_free(ctx->partial);
    dev_free(ctx->items);
    dev_free(ctx->tile_cost);
    dev_free(ctx->pool_gstack);
    dev_free(ctx->next_item);
    dev_free(ctx->tile_sorted);
    dev_free(ctx->tile_items);
    dev_free(ctx->tile_sky);
    dev_free(ctx->scratch);
    delete ctx;
}

static int pt_set_stream_impl(pt_ctx *ctx, void *hip_stream)
{
    PT_ENTER(ctx);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return PT_OK;
}

static int pt_set_flags_impl(pt_ctx *ctx, uint32_t flags)
{
    PT_ENTER(ctx);
    ctx->scene_generation++;  // a plan made before this call is not run again
    if (flags & ~(PT_FLAG_REFERENCE_DISPATCH | PT_FLAG_AOV | PT_FLAG_DETAILED_STATS | PT_FLAG_TILE_KERNEL | PT_FLAG_POOL_KERNEL | PT_FLAG_TRIANGLE_MATERIALS | PT_FLAG_REGROUP_SHORT | PT_FLAG_REBUILD_BVH | PT_FLAG_REFERENCE_WALK)) return fail(ctx, PT_ERR_INVALID, "pt_set_flags: unknown flag bits 0x%x", flags);
    ctx->flags = flags;
    return PT_OK;
}

// ---- the guard tree ---------------------------------------------------------------------------------------------
// The pooled kernel's default walk (PtDeviceScene::guard_on).  The reference tree splits at spatial midpoints and a walk
// of it takes about half as many steps again as a walk of a surface-area tree; but which hit a walk returns depends on
// its tree where the shader's float arithmetic is ambiguous, so a different tree alone is outside the bit-for-bit
// contract (PT_FLAG_REBUILD_BVH).  The guard tree is built so that a walk of it cannot lose a hit, and every hit it
// returns is checked to be the reference walk's:
//   * The triangle test (compute.wgsl:352-392) accepts a hit at float (t, u, v) when the exact ray point o + t d lies
//     within e of the triangle, e <= 23 u |d| |e1| |e2| (|o - a| + |edge|) / |det| to first order in u = 2^-24 (two
//     roundings per cross-product component, three per dot product, one for the reciprocal and one for each final
//     product), and |det| >= 1e-5 for an accepted hit.  Every triangle's bounds are padded by 48 u 1e5 |e1| |e2| reach
//     -- twice that bound for any ray with (|o - centre| + radius) max(1, |d|) <= reach -- plus an absolute term for the
//     slab test's own roundings, and the tree is built over the padded bounds.  The point o + t d of an accepted hit is
//     then inside every box above its triangle by more than the slab test can be wrong, so no such box is skipped or
//     pruned while the hit could still matter: the guard walk's closest hit is the closest of ALL accepted hits.
//   * That hit is the reference walk's too if the reference walk tests the triangle -- guaranteed when the hit point is
//     `margin` inside the triangle's own bounds, which every reference box above it contains (checked here for the
//     caller's tree) -- and if no other triangle is accepted at exactly the same t (flagged by the walk).
//   * Rays out of reach, rays with a zero or non-finite direction component, and hits that fail the check walk the
//     reference tree.
struct GuardPlan {
    bool on = false;
    float center[3] = {}, radius = 0.0f, reach = 0.0f, margin = 0.0f, pad_median = 0.0f;
    std::vector<float> pad;
};
static float float_above(double x)
{
    float f = (float)x;
    if ((double)f < x) f = std::nextafter(f, 3.40282347e+38f);
    return f;
}
static void plan_guard(const PtTriangle *tris, uint32_t n_tris, const PtNode *nodes, uint32_t n_nodes, const uint32_t *tri_idx,
                       const std::vector<int32_t> &depth, GuardPlan &gp)
{
    gp.on = false;
    // the mesh: finite, its bounds, its longest edge
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300}, edge_max = 0.0, abs_max = 0.0;
    for (uint32_t t = 0; t < n_tris; t++) {
        const float *v[3] = {tris[t].a, tris[t].b, tris[t].c};
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) {
                const double x = v[k][a];
                if (!std::isfinite(x)) return;
                mn[a] = std::min(mn[a], x);
                mx[a] = std::max(mx[a], x);
                abs_max = std::max(abs_max, std::fabs(x));
            }
    }
    if (!(abs_max < 1e15)) return;
    for (int a = 0; a < 3; a++) gp.center[a] = (float)(0.5 * (mn[a] + mx[a]));
    double radius = 0.0;
    std::vector<double> area2(n_tris), extent(n_tris);
    for (uint32_t t = 0; t < n_tris; t++) {
        const float *v[3] = {tris[t].a, tris[t].b, tris[t].c};
        double e[3] = {0.0, 0.0, 0.0}, ext = 0.0;
        for (int a = 0; a < 3; a++) {
            const double ab = (double)v[1][a] - v[0][a], ac = (double)v[2][a] - v[0][a], bc = (double)v[2][a] - v[1][a];
            e[0] += ab * ab; e[1] += ac * ac; e[2] += bc * bc;
            ext = std::max(ext, std::max(std::max(v[0][a], v[1][a]), v[2][a]) - (double)std::min(std::min(v[0][a], v[1][a]), v[2][a]));
        }
        for (int k = 0; k < 3; k++) {
            double r2 = 0.0;
            for (int a = 0; a < 3; a++) r2 += ((double)v[k][a] - gp.center[a]) * ((double)v[k][a] - gp.center[a]);
            radius = std::max(radius, std::sqrt(r2));
        }
        edge_max = std::max(edge_max, std::sqrt(std::max(e[0], std::max(e[1], e[2]))));
        area2[t] = std::sqrt(e[0]) * std::sqrt(e[1]);  // |e1| |e2|
        extent[t] = ext;
    }
    const double u = 5.9604644775390625e-08;  // 2^-24
    const double rb = (radius + edge_max) * 1.000001;
    const double reach = (double)env_u32("PTRT_GUARD_REACH", 8) * rb;  // (the factor: tuning builds only)
    if (!(rb > 0.0) || !(reach < 1e15)) return;
    const double o_max = std::max(std::max(std::fabs(gp.center[0]), std::fabs(gp.center[1])), std::fabs(gp.center[2])) + reach;  // no origin in reach has a larger coordinate
    const double slab = 8.0 * u * (abs_max + reach + o_max) + 6.0 * u * reach + 1e-30;
    gp.pad.resize(n_tris);
    for (uint32_t t = 0; t < n_tris; t++) gp.pad[t] = float_above(48.0 * u * 1e5 * reach * area2[t] * 1.000001 + slab);
    gp.radius = float_above(rb);
    gp.reach = (float)reach;
    if ((double)gp.reach > reach) gp.reach = std::nextafter(gp.reach, 0.0f);
    gp.margin = float_above(16.0 * u * (abs_max + o_max));
    // worth it?  a padding that is no small part of a triangle's size makes for a tree no better than the reference's
    {
        std::vector<float> p2(gp.pad);
        std::nth_element(p2.begin(), p2.begin() + n_tris / 2, p2.end());
        std::nth_element(extent.begin(), extent.begin() + n_tris / 2, extent.end());
        gp.pad_median = p2[n_tris / 2];
        if (!((double)gp.pad_median * 8.0 <= extent[n_tris / 2])) return;
    }
    // the caller's tree: every reachable box contains what is below it (children stored after their parent)
    for (uint32_t i = n_nodes; i-- > 0;) {
        if (depth[i] < 0) continue;
        const PtNode &n = nodes[i];
        for (int a = 0; a < 3; a++)
            if (!std::isfinite(n.min_corner[a]) || !std::isfinite(n.max_corner[a])) return;
        if (n.triangle_count == 0) {
            for (uint32_t c = n.left_child_index; c <= n.left_child_index + 1u; c++)
                for (int a = 0; a < 3; a++)
                    if (!(nodes[c].min_corner[a] >= n.min_corner[a]) || !(nodes[c].max_corner[a] <= n.max_corner[a])) return;
        } else {
            for (uint32_t k = 0; k < n.triangle_count; k++) {
                const PtTriangle &tr = tris[tri_idx[n.left_child_index + k]];
                const float *v[3] = {tr.a, tr.b, tr.c};
                for (int q = 0; q < 3; q++)
                    for (int a = 0; a < 3; a++)
                        if (!(v[q][a] >= n.min_corner[a]) || !(v[q][a] <= n.max_corner[a])) return;
            }
        }
    }
    gp.on = true;
}

extern "C" int64_t pt_bvh_build_sah_padded(const void *tris, uint32_t n_tris, const float *pad, void *nodes_out, uint32_t *idx_out);

static int pt_set_mesh_impl(pt_ctx *ctx, const void *tris_v, uint32_t n_tris, const void *nodes_v, uint32_t n_nodes, const uint32_t *tri_idx, uint32_t n_idx)
{
    PT_ENTER(ctx);
    ctx->scene_generation++;  // a plan made before this call is not run again
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // kernels in flight may still read the old mesh
    if (n_nodes == 0) {
        free_mesh(ctx);
        return PT_OK;
    }
    if (!tris_v || !nodes_v || !tri_idx || n_tris == 0 || n_idx == 0) return fail(ctx, PT_ERR_INVALID, "pt_set_mesh: null or empty input with n_nodes=%u", n_nodes);
    const PtTriangle *tris = static_cast<const PtTriangle *>(tris_v);
    const PtNode *nodes = static_cast<const PtNode *>(nodes_v);
    // PT_FLAG_REBUILD_BVH: the library's own tree over the caller's triangles instead of the caller's (the reference's)
    std::vector<PtNode> own_nodes;
    std::vector<uint32_t> own_idx;
    if (ctx->flags & PT_FLAG_REBUILD_BVH) {
        own_nodes.resize(2 * (size_t)n_tris - 1);
        own_idx.resize(n_tris);
        const int64_t used = pt_bvh_build_sah(tris_v, n_tris, own_nodes.data(), own_idx.data());
        if (used <= 0) return fail(ctx, used == PT_ERR_NOMEM ? PT_ERR_NOMEM : PT_ERR_INVALID, "pt_set_mesh: PT_FLAG_REBUILD_BVH: the tree could not be built");
        nodes = own_nodes.data();
        n_nodes = (uint32_t)used;
        tri_idx = own_idx.data();
        n_idx = n_tris;
    }
    // Walk table: node pairs (32 bytes per node, stored one node late) and, from the next 64-byte boundary on, one
    // 64-byte record per triangle slot; positions are 32-byte units counted from the array's second unit.
    const uint64_t tri_first_unit = ((uint64_t)n_nodes + 2) & ~1ull;  // even: 64-byte aligned
    uint64_t table_units = tri_first_unit + 2 * (uint64_t)n_idx;
    if (table_units > (1ull << 27)) return fail(ctx, PT_ERR_LIMIT, "pt_set_mesh: %u nodes and %u triangle slots need a walk table of more than 4 GiB", n_nodes, n_idx);
    const uint32_t tri_base = (uint32_t)tri_first_unit - 1u;

    // Validate, and find the deepest reachable node.  Children must be stored after their
    // parent (the reference's builder guarantees it, bvh.rs:131-134), which rules out cycles.
    std::vector<int32_t> depth(n_nodes, -1);
    depth[0] = 0;
    uint32_t max_depth = 0;
    for (uint32_t i = 0; i < n_nodes; i++) {
        if (depth[i] < 0) continue;  // unreachable: never visited by traversal
        const PtNode &n = nodes[i];
        max_depth = std::max<uint32_t>(max_depth, (uint32_t)depth[i]);
        if (n.triangle_count == 0) {
            const uint64_t l = n.left_child_index;
            if (l <= i || l + 1 >= n_nodes) return fail(ctx, PT_ERR_INVALID, "pt_set_mesh: node %u has children %llu,%llu outside (%u, %u)", i, (unsigned long long)l, (unsigned long long)l + 1, i, n_nodes);
            // the device layout stores sibling pairs interleaved in place, so a node can have one parent only
            if (depth[l] >= 0 || depth[l + 1] >= 0) return fail(ctx, PT_ERR_INVALID, "pt_set_mesh: node %llu or %llu has more than one parent (not a tree)", (unsigned long long)l, (unsigned long long)l + 1);
            depth[l] = depth[i] + 1;
            depth[l + 1] = depth[i] + 1;
        } else {
            const uint64_t end = (uint64_t)n.left_child_index + n.triangle_count;
            if (end > n_idx) return fail(ctx, PT_ERR_INVALID, "pt_set_mesh: leaf %u covers indices [%u, %llu) beyond %u", i, n.left_child_index, (unsigned long long)end, n_idx);
        }
    }
    for (uint32_t k = 0; k < n_idx; k++)
        if (tri_idx[k] >= n_tris) return fail(ctx, PT_ERR_INVALID, "pt_set_mesh: triangle index %u at position %u out of range (%u triangles)", tri_idx[k], k, n_tris);

    auto token_of = [&](uint32_t i) -> uint32_t {
        const PtNode &n = nodes[i];
        if (n.triangle_count == 0) return n.left_child_index;
        if (n.triangle_count <= 7u) return PT_TOKEN_LEAF | (n.triangle_count << 28) | (tri_base + 2u * n.left_child_index);
        return PT_TOKEN_LEAF | i;
    };

    // The traversal stacks are sized by the tree's depth; a degenerate tree (the reference's builder peels a
    // clump of identical triangles off one at a time) would ask for gigabytes of them.
    if (max_depth > PT_MAX_BVH_DEPTH)
        return fail(ctx, PT_ERR_LIMIT, "pt_set_mesh: BVH depth %u exceeds %u (the reference shader's own stack holds 15 entries)", max_depth, PT_MAX_BVH_DEPTH);

    // The guard tree (see plan_guard), for meshes the pooled kernel takes: its triangle records behind the reference
    // tree's (slots n_idx ..), its pairs behind those -- guard node j at unit guard_first_unit + j + 1, so that pairs
    // (odd, even) are aligned like the reference tree's.
    GuardPlan gp;
    std::vector<uint32_t> quad_nodes, quad_children, quad_of;  // per record: the node it stands for, its (up to) four children; per node: its record
    uint64_t quad_first_unit = 0;
    uint32_t q_depth = 0;
    std::vector<PtNode> g_nodes;
    std::vector<uint32_t> g_idx;
    uint32_t n_gnodes = 0, g_depth = 0;
    uint64_t guard_first_unit = 0;
    if (!(ctx->flags & (PT_FLAG_REBUILD_BVH | PT_FLAG_REFERENCE_WALK)) && n_tris >= PT_POOL_MIN_TRIANGLES && nodes[0].triangle_count == 0) {
        plan_guard(tris, n_tris, nodes, n_nodes, tri_idx, depth, gp);
        guard_first_unit = table_units;  // (even)
        if (gp.on && guard_first_unit + 2 * (uint64_t)n_tris + 2 * (uint64_t)n_tris + 2 > (1ull << 27)) gp.on = false;  // no room under 4 GiB: the reference tree alone
        if (gp.on) {
            g_nodes.resize(2 * (size_t)n_tris - 1);
            g_idx.resize(n_tris);
            const int64_t used = pt_bvh_build_sah_padded(tris_v, n_tris, gp.pad.data(), g_nodes.data(), g_idx.data());
            if (used == PT_ERR_NOMEM) return fail(ctx, PT_ERR_NOMEM, "pt_set_mesh: out of host memory");
            gp.on = used >= 3 && g_nodes[0].triangle_count == 0;
            n_gnodes = gp.on ? (uint32_t)used : 0u;
            // parents before children, leaves small enough for inline tokens: the builder's own guarantees, checked
            std::vector<uint32_t> gd(n_gnodes, 0u);
            for (uint32_t i = 0; i < n_gnodes && gp.on; i++) {
                const PtNode &n = g_nodes[i];
                g_depth = std::max(g_depth, gd[i]);
                if (n.triangle_count == 0) {
                    if (n.left_child_index <= i || (uint64_t)n.left_child_index + 1 >= n_gnodes) gp.on = false;
                    else gd[n.left_child_index] = gd[n.left_child_index + 1] = gd[i] + 1u;
                } else if (n.triangle_count > 7u || (uint64_t)n.left_child_index + n.triangle_count > n_tris) {
                    gp.on = false;
                }
            }
            if (g_depth > PT_MAX_BVH_DEPTH) gp.on = false;
        }
        if (gp.on) table_units = guard_first_unit + 2 * (uint64_t)n_tris + (((uint64_t)n_gnodes + 2) & ~1ull);
        // The same tree with four children to a record (PtDeviceScene::guard_quads): every interior node takes its two
        // children and, as long as it has fewer than four, replaces the largest interior one of them by that one's two.
        if (gp.on) {
            auto half_area = [&](uint32_t i) {
                const PtNode &n = g_nodes[i];
                const double dx = (double)n.max_corner[0] - n.min_corner[0], dy = (double)n.max_corner[1] - n.min_corner[1], dz = (double)n.max_corner[2] - n.min_corner[2];
                return dx * dy + dy * dz + dz * dx;
            };
            quad_of.assign(n_gnodes, 0xFFFFFFFFu);
            quad_nodes.push_back(0);
            quad_of[0] = 0;
            std::vector<uint32_t> qdepth{0};
            for (size_t qi = 0; qi < quad_nodes.size(); qi++) {
                const PtNode &n = g_nodes[quad_nodes[qi]];
                uint32_t ch[4] = {n.left_child_index, n.left_child_index + 1u, 0, 0}, nch = 2;
                while (nch < 4u) {
                    int best = -1;
                    double best_area = -1.0;
                    for (uint32_t c = 0; c < nch; c++)
                        if (g_nodes[ch[c]].triangle_count == 0 && half_area(ch[c]) > best_area) { best = (int)c; best_area = half_area(ch[c]); }
                    if (best < 0) break;
                    const uint32_t l = g_nodes[ch[best]].left_child_index;
                    ch[best] = l;
                    ch[nch++] = l + 1u;
                }
                for (uint32_t c = 0; c < 4u; c++) quad_children.push_back(c < nch ? ch[c] : 0xFFFFFFFFu);
                for (uint32_t c = 0; c < nch; c++)
                    if (g_nodes[ch[c]].triangle_count == 0) {
                        quad_of[ch[c]] = (uint32_t)quad_nodes.size();
                        quad_nodes.push_back(ch[c]);
                        qdepth.push_back(qdepth[qi] + 1u);
                        q_depth = std::max(q_depth, qdepth[qi] + 1u);
                    }
            }
            quad_first_unit = table_units;  // (even)
            if (quad_first_unit + 2 * (uint64_t)quad_nodes.size() > (1ull << 27)) quad_nodes.clear();  // no room: the pair form alone
            else table_units = quad_first_unit + 2 * (uint64_t)quad_nodes.size();
        }
    }
    const uint64_t guard_node_unit = guard_first_unit + 2 * (uint64_t)n_tris;  // (even) guard node j lives at unit guard_node_unit + j + 1
    const uint32_t n_slots = n_idx + (gp.on ? n_tris : 0u);

    // Sibling pairs (l, l+1), interleaved component by component so that the two slab tests of a node
    // visit are packed two-wide: {lmin.x rmin.x lmin.y rmin.y} {lmin.z rmin.z lmax.x rmax.x}
    // {lmax.y rmax.y lmax.z rmax.z} {l token, r token, 0, 0}, at the place of nodes l and l+1.
    std::vector<float4> h_nodes(2 * (size_t)table_units, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    std::vector<uint2> h_meta(n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) {
        const PtNode &n = nodes[i];
        h_meta[i] = make_uint2(n.triangle_count ? tri_base + 2u * n.left_child_index : n.left_child_index, n.triangle_count);
        if (depth[i] < 0 || n.triangle_count != 0) continue;
        const PtNode &a = nodes[n.left_child_index], &b = nodes[n.left_child_index + 1];
        float4 *rec = &h_nodes[2 * ((size_t)n.left_child_index + 1)];
        rec[0] = make_float4(a.min_corner[0], b.min_corner[0], a.min_corner[1], b.min_corner[1]);
        rec[1] = make_float4(a.min_corner[2], b.min_corner[2], a.max_corner[0], b.max_corner[0]);
        rec[2] = make_float4(a.max_corner[1], b.max_corner[1], a.max_corner[2], b.max_corner[2]);
        rec[3] = make_float4(u2f(token_of(n.left_child_index)), u2f(token_of(n.left_child_index + 1)), 0.0f, 0.0f);
    }
    // triangles as 64-byte records like the node pairs: a walk step reads one aligned record whichever it is, and a
    // random read of it is one 64-byte half line (48-byte records cost two of them three times in four)
    bool mesh_gizmo = false;
    float4 *const h_pos = h_nodes.data() + 2 * (size_t)tri_first_unit;
    std::vector<float4> h_nrm(3 * (size_t)n_slots);
    for (uint32_t k = 0; k < n_slots; k++) {
        const uint32_t ti = k < n_idx ? tri_idx[k] : g_idx[k - n_idx];  // (the guard tree's slots follow in its own leaf order)
        const PtTriangle &t = tris[ti];
        // edges as the shader computes them (compute.wgsl:353-354)
        h_pos[4 * (size_t)k] = make_float4(t.a[0], t.a[1], t.a[2], u2f(ti));
        h_pos[4 * (size_t)k + 1] = make_float4(t.b[0] - t.a[0], t.b[1] - t.a[1], t.b[2] - t.a[2], (float)t.material);
        h_pos[4 * (size_t)k + 2] = make_float4(t.c[0] - t.a[0], t.c[1] - t.a[1], t.c[2] - t.a[2], 0.0f);
        // albedo and material ride in the spare components; read only with PT_FLAG_TRIANGLE_MATERIALS
        h_nrm[3 * (size_t)k] = make_float4(t.na[0], t.na[1], t.na[2], t.albedo[0]);
        h_nrm[3 * (size_t)k + 1] = make_float4(t.nb[0], t.nb[1], t.nb[2], t.albedo[1]);
        h_nrm[3 * (size_t)k + 2] = make_float4(t.nc[0], t.nc[1], t.nc[2], t.albedo[2]);
        mesh_gizmo |= (t.material == 3u);
    }
    if (gp.on) {
        auto guard_token = [&](uint32_t i) -> uint32_t {
            const PtNode &n = g_nodes[i];
            if (n.triangle_count == 0) return (uint32_t)guard_node_unit + n.left_child_index;  // position of its children's pair (unit - 1)
            return PT_TOKEN_LEAF | (n.triangle_count << 28) | (tri_base + 2u * (n_idx + n.left_child_index));
        };
        for (uint32_t i = 0; i < n_gnodes; i++) {
            const PtNode &n = g_nodes[i];
            if (n.triangle_count != 0) continue;
            const PtNode &a = g_nodes[n.left_child_index], &b = g_nodes[n.left_child_index + 1];
            float4 *rec = &h_nodes[2 * (guard_node_unit + (size_t)n.left_child_index + 1)];
            rec[0] = make_float4(a.min_corner[0], b.min_corner[0], a.min_corner[1], b.min_corner[1]);
            rec[1] = make_float4(a.min_corner[2], b.min_corner[2], a.max_corner[0], b.max_corner[0]);
            rec[2] = make_float4(a.max_corner[1], b.max_corner[1], a.max_corner[2], b.max_corner[2]);
            rec[3] = make_float4(u2f(guard_token(n.left_child_index)), u2f(guard_token(n.left_child_index + 1)), 0.0f, 0.0f);
        }
        for (size_t qi = 0; qi < quad_nodes.size(); qi++) {
            const uint32_t *ch = &quad_children[4 * qi];
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (uint32_t c = 0; c < 4u && ch[c] != 0xFFFFFFFFu; c++)
                for (int a = 0; a < 3; a++) {
                    lo[a] = std::min(lo[a], (double)g_nodes[ch[c]].min_corner[a]);
                    hi[a] = std::max(hi[a], (double)g_nodes[ch[c]].max_corner[a]);
                }
            float org[3], scale[3];
            uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
            for (int a = 0; a < 3; a++) {
                org[a] = (float)lo[a];  // (the children's corners are floats: exact)
                int e = -126;
                if (hi[a] > lo[a]) e = std::max(-126, (int)std::ceil(std::log2((hi[a] - lo[a]) / 255.0)));
                while (std::ldexp(255.0, e) < hi[a] - lo[a]) e++;
                scale[a] = (float)std::ldexp(1.0, e);
                for (uint32_t c = 0; c < 4u; c++) {
                    uint32_t bl = 0, bh = 0;  // (an absent child: its token says so)
                    if (ch[c] != 0xFFFFFFFFu) {
                        const double l = ((double)g_nodes[ch[c]].min_corner[a] - lo[a]) / scale[a], h = ((double)g_nodes[ch[c]].max_corner[a] - lo[a]) / scale[a];
                        bl = (uint32_t)std::min(255.0, std::max(0.0, std::floor(l)));
                        bh = (uint32_t)std::min(255.0, std::max(0.0, std::ceil(h)));
                        // (origin + byte * scale is exact in double; the box may only grow)
                        if (lo[a] + bl * (double)scale[a] > (double)g_nodes[ch[c]].min_corner[a] || lo[a] + bh * (double)scale[a] < (double)g_nodes[ch[c]].max_corner[a])
                            return fail(ctx, PT_ERR_INTERNAL, "pt_set_mesh: a quantised box of the second tree does not contain its node");
                    }
                    qlo[a] |= bl << (8u * c);
                    qhi[a] |= bh << (8u * c);
                }
            }
            auto quad_token = [&](uint32_t i) -> uint32_t {
                if (i == 0xFFFFFFFFu) return 0xFFFFFFFFu;
                if (g_nodes[i].triangle_count == 0) return (uint32_t)(quad_first_unit - 1u) + 2u * quad_of[i];
                return guard_token(i);
            };
            float4 *rec = &h_nodes[2 * (quad_first_unit + 2 * qi)];
            rec[0] = make_float4(org[0], org[1], org[2], scale[0]);
            rec[1] = make_float4(u2f(qlo[0]), u2f(qlo[1]), u2f(qlo[2]), u2f(qhi[0]));
            rec[2] = make_float4(u2f(qhi[1]), u2f(qhi[2]), u2f(quad_token(ch[0])), u2f(quad_token(ch[1])));
            rec[3] = make_float4(u2f(quad_token(ch[2])), u2f(quad_token(ch[3])), scale[1], scale[2]);
        }
    }

    free_mesh(ctx);
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->nodes_alloc, h_nodes.size())) != PT_OK || (rc = dev_alloc(ctx, &ctx->node_meta, h_meta.size())) != PT_OK ||
        (rc = dev_alloc(ctx, &ctx->tri_nrm, h_nrm.size())) != PT_OK) {
        free_mesh(ctx);
        return rc;
    }
    PT_HIP(ctx, hipMemcpy(ctx->nodes_alloc, h_nodes.data(), h_nodes.size() * sizeof(float4), hipMemcpyHostToDevice));
    PT_HIP(ctx, hipMemcpy(ctx->node_meta, h_meta.data(), h_meta.size() * sizeof(uint2), hipMemcpyHostToDevice));
    PT_HIP(ctx, hipMemcpy(ctx->tri_nrm, h_nrm.data(), h_nrm.size() * sizeof(float4), hipMemcpyHostToDevice));
    ctx->n_nodes = n_nodes;
    ctx->n_tri_slots = n_idx;
    ctx->tri_base = tri_base;
    ctx->mesh_has_gizmo = mesh_gizmo;
    ctx->root_token = token_of(0);
    if (nodes[0].triangle_count == 0)
        for (int k = 0; k < 4; k++) ctx->root_pair[k] = h_nodes[2 * ((size_t)nodes[0].left_child_index + 1) + k];
    ctx->bvh_depth = max_depth;
    // (the pooled kernel walks the second tree's pair form; the four-child form, which the other kernels walk, gives a
    // walk up rather than outgrow the stack it finds, so those stay sized by the caller's tree)
    ctx->walk_depth = std::max(max_depth, gp.on ? g_depth : 0u);
    ctx->guard_on = gp.on;
    if (gp.on) {
        for (int k = 0; k < 3; k++) ctx->guard_center[k] = gp.center[k];
        ctx->guard_radius = gp.radius;
        ctx->guard_reach = gp.reach;
        ctx->guard_margin = gp.margin;
        ctx->guard_root = (uint32_t)guard_node_unit + g_nodes[0].left_child_index;
        ctx->guard_quads = quad_nodes.empty() ? 0u : (uint32_t)(quad_first_unit - 1u);
        if (env_str("PTRT_NO_QUADS")) ctx->guard_quads = 0;  // (tuning builds: the pair form everywhere)
        ctx->guard_nodes = n_gnodes;
        ctx->guard_depth = g_depth;
        ctx->guard_pad_median = gp.pad_median;
    }
    return PT_OK;
}

static int pt_set_sky_cube_impl(pt_ctx *ctx, const float *rgba32f, uint32_t size)
{
    PT_ENTER(ctx);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (size == 0) {
        dev_free(ctx->sky);
        ctx->sky_size = 0;
        return PT_OK;
    }
    if (!rgba32f) return fail(ctx, PT_ERR_INVALID, "pt_set_sky_cube: null texels");
    if (size > 16384u) return fail(ctx, PT_ERR_LIMIT, "pt_set_sky_cube: face size %u above 16384", size);
    const size_t texels = 6 * (size_t)size * size;
    if (ctx->sky_size != size) {
        dev_free(ctx->sky);
        ctx->sky_size = 0;
        int rc = dev_alloc(ctx, &ctx->sky, texels);
        if (rc != PT_OK) return rc;
    }
    PT_HIP(ctx, hipMemcpy(ctx->sky, rgba32f, texels * sizeof(float4), hipMemcpyHostToDevice));
    ctx->sky_size = size;
    return PT_OK;
}

static int pt_set_sky_equirect_impl(pt_ctx *ctx, const float *rgba32f, uint32_t w, uint32_t h, uint32_t cube_size)
{
    PT_ENTER(ctx);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!rgba32f || w == 0 || h == 0 || cube_size == 0) return fail(ctx, PT_ERR_INVALID, "pt_set_sky_equirect: null or empty input");
    if (cube_size > 16384u || w > 32768u || h > 32768u) return fail(ctx, PT_ERR_LIMIT, "pt_set_sky_equirect: size above limit");
    const size_t texels = 6 * (size_t)cube_size * cube_size;
    if (ctx->sky_size != cube_size) {
        dev_free(ctx->sky);
        ctx->sky_size = 0;
        int rc = dev_alloc(ctx, &ctx->sky, texels);
        if (rc != PT_OK) return rc;
    }
    float4 *src = nullptr;
    PT_HIP(ctx, hipMalloc((void **)&src, (size_t)w * h * sizeof(float4)));
    hipError_t e = hipMemcpy(src, rgba32f, (size_t)w * h * sizeof(float4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = pt_launch_equirect(src, w, h, ctx->sky, cube_size, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(src);
    if (e != hipSuccess) return fail(ctx, PT_ERR_HIP, "pt_set_sky_equirect: %s", hipGetErrorString(e));
    ctx->sky_size = cube_size;
    return PT_OK;
}

static int pt_read_sky_cube_impl(pt_ctx *ctx, float *out, size_t capacity_floats, uint32_t *size)
{
    PT_ENTER(ctx);
    if (size) *size = ctx->sky_size;
    if (!out) return size ? PT_OK : fail(ctx, PT_ERR_INVALID, "pt_read_sky_cube: null output");
    if (!ctx->sky_size) return fail(ctx, PT_ERR_STATE, "pt_read_sky_cube: no sky set");
    const size_t need = 6 * (size_t)ctx->sky_size * ctx->sky_size * 4;
    if (capacity_floats < need) return fail(ctx, PT_ERR_INVALID, "pt_read_sky_cube: room for %zu floats, the %u^2 cube has %zu", capacity_floats, ctx->sky_size, need);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(out, ctx->sky, need * sizeof(float), hipMemcpyDeviceToHost));
    return PT_OK;
}

static int pt_set_camera_impl(pt_ctx *ctx, const void *camera64)
{
    PT_ENTER(ctx);
    ctx->scene_generation++;  // a plan made before this call is not run again
    if (!camera64) return fail(ctx, PT_ERR_INVALID, "pt_set_camera: null pointer");
    std::memcpy(&ctx->camera, camera64, sizeof(PtCamera));
    ctx->have_camera = true;
    return PT_OK;
}

static int pt_set_spheres_impl(pt_ctx *ctx, const void *sphere_data8208)
{
    PT_ENTER(ctx);
    ctx->scene_generation++;  // a plan made before this call is not run again
    if (!sphere_data8208) return fail(ctx, PT_ERR_INVALID, "pt_set_spheres: null pointer");
    const PtSphereData *sd = static_cast<const PtSphereData *>(sphere_data8208);
    // the storage buffer holds 256 spheres; a larger count would read past it
    const uint32_t n = sd->sphere_count > PT_MAX_SPHERES ? PT_MAX_SPHERES : sd->sphere_count;
    std::vector<float4> packed(3 * (size_t)n);
    ctx->has_gizmo = false;
    for (uint32_t i = 0; i < n; i++) {
        const PtSphere &sp = sd->spheres[i];
        if (sp.material >= 3.0f && sp.material < 4.0f) ctx->has_gizmo = true;
        packed[3 * i] = make_float4(sp.center[0], sp.center[1], sp.center[2], sp.radius);
        packed[3 * i + 1] = make_float4(sp.albedo[0], sp.albedo[1], sp.albedo[2], sp.material);
        packed[3 * i + 2] = make_float4(1.0f / sp.radius, 0.0f, 0.0f, 0.0f);
    }
    // stream-ordered copy from pageable memory: the runtime stages the bytes before returning
    if (n) PT_HIP(ctx, hipMemcpyAsync(ctx->spheres, packed.data(), packed.size() * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    ctx->n_spheres = n;
    return PT_OK;
}

static int pt_set_settings_impl(pt_ctx *ctx, const void *settings16)
{
    PT_ENTER(ctx);
    ctx->scene_generation++;  // a plan made before this call is not run again
    if (!settings16) return fail(ctx, PT_ERR_INVALID, "pt_set_settings: null pointer");
    std::memcpy(&ctx->settings, settings16, sizeof(PtSettings));
    return PT_OK;
}

// pt_render_frames and pt_render_frames_static: the same launches; `use_static` takes the work-item table, the
// partial sums and the deep-stack spill space from this GPU's static scratch array instead of context-owned
// allocations that grow on demand.
static int render_frames_unguarded(pt_ctx *ctx, uint32_t time0, uint32_t stride, uint32_t n_frames, bool use_static)
{
    PT_ENTER(ctx);
    if (use_static && ctx->window_size) return fail(ctx, PT_ERR_STATE, "pt_render_frames_static: not available while a window is set (pt_set_window)");
    // A plan is run again only if its table is still where and what it was: not across a change of entry point, and
    // in the static array not after another context's call.
    unsigned char *static_base = nullptr;
    size_t static_used = 0;
    if (use_static) {
        StaticScratch &ss = g_static[ctx->device];
        std::lock_guard<std::mutex> g(ss.lock);
        if (!ss.ptr) return fail(ctx, PT_ERR_STATE, "pt_render_frames_static: the static scratch array has no address on device %d", ctx->device);
        static_base = ss.ptr;
        if (ss.last_user != ctx) ctx->scene_generation++;
        ss.last_user = ctx;
    }
    if (ctx->last_static != use_static) ctx->scene_generation++;
    ctx->last_static = use_static;
    // bump allocation inside the static array (256-byte granules); null when the array is too small
    auto carve = [&](size_t bytes) -> void * {
        const size_t at = (static_used + 255u) & ~(size_t)255u;
        if (bytes > PT_STATIC_SCRATCH_BYTES || at > PT_STATIC_SCRATCH_BYTES - bytes) return nullptr;
        static_used = at + bytes;
        return static_base + at;
    };
    if (!ctx->have_camera) return fail(ctx, PT_ERR_STATE, "pt_render: no camera set");
    if (n_frames == 0) return PT_OK;
    if (ctx->settings.samples_per_pixel == 0) return fail(ctx, PT_ERR_INVALID, "pt_render: samples_per_pixel is 0 (the shader would divide by zero)");

    PtTraceParams p{};
    p.sc = device_scene(ctx);
    p.fr = pt_camera_frame(reinterpret_cast<const float *>(&ctx->camera), ctx->width, ctx->height);
    camera_cull_rects(ctx, p.fr, p.cull);
    p.width = ctx->width;
    p.height = ctx->height;
    p.cols = ctx->width;
    p.rows = ctx->height;
    if (ctx->flags & PT_FLAG_REFERENCE_DISPATCH) {
        p.cols = (ctx->width / 16u) * 16u;
        p.rows = (ctx->height / 16u) * 16u;
    }
    p.spp = ctx->settings.samples_per_pixel;
    p.depth = ctx->settings.depth;
    p.inv_spp = 1.0f / (float)ctx->settings.samples_per_pixel;
    p.has_gizmo = ctx->has_gizmo || ((ctx->flags & PT_FLAG_TRIANGLE_MATERIALS) && ctx->mesh_has_gizmo);
    p.stride = stride;
    p.accum = ctx->accum;
    p.aov_prim = (ctx->flags & PT_FLAG_AOV) ? ctx->aov_prim : nullptr;
    p.aov_t = ctx->aov_t;
    p.counters = ctx->counters;
    // Which megakernel: the pooled one needs a launch long enough to fill its path pool and to plan by measured cost
    // (16 frames), and its spp > 1 variants run at three waves per SIMD instead of four and without the one-sample
    // variants' short cuts.  Measured on this GPU (tools/kernel_choice.py, 1080p, depth 8, pooled / per-tile, best of three
    // launches; profiles/r02/kernel_choice_scan_final.txt), one sample per frame: 2-12 frames 0.35-0.57 at every mesh
    // size; 16 frames 1.01-1.21; 32 frames 1.14-1.52; 96 frames 1.48-1.68; no triangles at all 0.95, Cornell (10
    // triangles, 16 spp) 0.83.  2 to 16 samples per frame, 768 to 69 k triangles: 16 frames and more 1.32-1.57, 4 and 8
    // frames 0.57-0.92 (profiles/r02/kernel_choice_scan_spp.txt).
    // Flags or PTRT_KERNEL force either; depth 0 and images wider or taller than 65535 always take the
    // per-tile kernel.
    const bool pool_possible = ctx->width <= 0xFFFFu && ctx->height <= 0xFFFFu && ctx->settings.depth >= 1u;
    const uint32_t launch_frames = ctx->window_size ? 1u : n_frames;  // a display window renders frame by frame
    // Several samples per pixel and frame (256 path slots per workgroup instead of 320, so that four workgroups
    // still share a CU; work items sized by samples) follow the same rule; a single frame of many samples (its samples
    // share one RNG stream per pixel and cannot be split) goes to the pooled kernel from 48 samples on big meshes.
    const uint32_t spp_now = std::max(ctx->settings.samples_per_pixel, 1u);
    bool use_pool = (ctx->n_tri_slots >= PT_POOL_MIN_TRIANGLES && launch_frames >= PT_POOL_MIN_FRAMES) ||
                    (launch_frames == 1u && spp_now >= PT_POOL_ONE_FRAME_MIN_SAMPLES && ctx->n_tri_slots >= PT_POOL_ONE_FRAME_MIN_TRIANGLES);
    // Short launches (one work item per tile and frame): since work items are claimed in batches they are the pooled
    // kernel's, too -- at one sample per frame from 8 frames on meshes of 512 triangles and more, and at any sample count
    // from 4 frames on big meshes (4 frames of 16 samples, pooled / per-tile: 0.81 / 0.95 / 1.20 / 1.40 / 0.97 at 768 /
    // 3 k / 8.7 k / 28 k / 69 k triangles).
    if ((spp_now == 1u && launch_frames >= PT_POOL_SHORT_MIN_FRAMES && ctx->n_tri_slots >= PT_POOL_SHORT_MIN_TRIANGLES) ||
        (launch_frames >= PT_POOL_SHORT_BIG_MIN_FRAMES && ctx->n_tri_slots >= PT_POOL_SHORT_BIG_MIN_TRIANGLES))
        use_pool = true;
    // (regrouped short launches, PT_FLAG_REGROUP_SHORT: like the long ones; pooled / per-tile time 0.77-0.96 at 4 to 15
    // frames on 192 to 69 k triangles)
    if ((ctx->flags & PT_FLAG_REGROUP_SHORT) && launch_frames >= 4u && ctx->n_tri_slots >= PT_POOL_MIN_TRIANGLES) use_pool = true;
    if (ctx->flags & PT_FLAG_POOL_KERNEL) use_pool = true;
    if (ctx->flags & PT_FLAG_TILE_KERNEL) use_pool = false;
    if (const char *env = env_str("PTRT_KERNEL")) {  // "pool" / "tile": tuning builds only, overrides the flags
        if (env[0] == 'p') use_pool = true;
        else if (env[0] == 't') use_pool = false;
    }
    use_pool = use_pool && pool_possible;
    // spill space for traversal stacks deeper than the kernels keep in LDS
    const size_t pgs = use_pool ? pt_pool_gstack_bytes(ctx->n_cus, ctx->walk_depth) : 0;
    uint32_t *pool_gstack = ctx->pool_gstack;
    if (use_static) {
        pool_gstack = pgs ? static_cast<uint32_t *>(carve(pgs)) : nullptr;
        if (pgs && !pool_gstack)
            return fail(ctx, PT_ERR_LIMIT, "pt_render_frames_static: a BVH of depth %u needs %zu bytes of traversal-stack spill space, beyond the %llu-byte static scratch", ctx->walk_depth, pgs, (unsigned long long)PT_STATIC_SCRATCH_BYTES);
    } else if (pgs > ctx->pool_gstack_bytes) {
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        dev_free(ctx->pool_gstack);
        ctx->pool_gstack_bytes = 0;
        PT_HIP(ctx, hipMalloc((void **)&ctx->pool_gstack, pgs));
        ctx->pool_gstack_bytes = pgs;
        pool_gstack = ctx->pool_gstack;
    }
    // Slots that must be ready before a wave runs a shading pass.  Per-tile kernel: long walks (big meshes)
    // favour shading early, short ones favour full passes (Cornell: 43 vs 37 Grays/s).  Pooled kernel: half a
    // wave's worth (48 for a wave that is busy walking); waiting for full passes was better only before misses
    // stayed in registers (bench scene 32.5 vs 31.5 Grays/s, 10 M-triangle field 4.76 vs 4.46).
    p.refill_lanes = env_u32("PTRT_REFILL_LANES", use_pool ? PT_POOL_REFILL_LANES : (ctx->n_tri_slots >= PT_TILE_EARLY_SHADE_TRIANGLES ? 24 : 48));

    PtSchedule sch{};
    sch.cost = ctx->tile_cost;
    sch.sorted = ctx->tile_sorted;
    sch.tile_items = ctx->tile_items;
    sch.tile_sky = ctx->tile_sky;
    sch.sky_tiles = !env_str("PTRT_NO_SKY_TILES");  // (tuning builds: A/B switches)
    sch.side_stream = env_str("PTRT_NO_SIDE_STREAM") ? nullptr : ctx->side_stream;
    sch.main_stream = ctx->main_stream;
    sch.main_event = ctx->main_event;
    sch.start_event = ctx->start_event;
    sch.fork_event = ctx->fork_event;
    sch.join_event = ctx->join_event;
    sch.cost_valid = &ctx->cost_valid;
    // Items per resident wave slot.  Pooled kernel: a pixel's frames run one after the other, so costly tiles need
    // short items, but every item costs 64 job hand-outs and a row of partial sums: one item per slot and four
    // frames of the launch, at most 64 (bench scene, Grays/s: 64 frames 26.2 / 27.8 / 30.6 / 30.3 at 8 / 10 / 16 / 21;
    // 128 frames 33.4 / 34.2 / 34.2 / 33.7 at 16 / 21 / 32 / 42; 256 frames 33.7 / 35.0 / 35.6 / 34.9 / 33.6 at
    // 32 / 42 / 64 / 96 / 128; 512 frames 36.1 / 36.6 / 35.2 at 64 / 85 / 128; with the sky-only tiles out of the plan
    // and sphere hits shaded in the pass over misses, 256 frames: 43.0 / 44.8 / 46.2 / 46.2 / 45.7 / 43.4 at
    // 24 / 32 / 40 / 48 / 56 / 96 with items of at least 2 frames, 44.6 at 64 with at least 4).  Per-tile kernel: every
    // item costs a wave its start-up and its ragged end.
    // A frame of several samples per pixel counts as that many frames here.
    const uint32_t spp = std::max(ctx->settings.samples_per_pixel, 1u);
    const uint32_t launch_samples = (uint32_t)std::min<uint64_t>((uint64_t)launch_frames * spp, 1u << 20);
    // (meshes of a million triangles and more -- every tile that sees the mesh is costly, and the launch is bound by
    // memory, not by the items' fixed costs -- get twice as many: 10 M-triangle field at 128 frames 6245 / 6332 / 6508 /
    // 6562 Mrays/s at 32 / 48 / 64 / 96 items per slot)
    const bool huge_mesh = ctx->n_tri_slots >= PT_POOL_HUGE_MESH_TRIANGLES;
    const uint32_t per_slot = env_u32("PTRT_ITEMS_PER_SLOT", use_pool ? (huge_mesh ? std::min(std::max(launch_samples / 2u, 6u), 96u)
                                                                               : std::min(std::max(launch_samples / 4u, 6u), 48u)) : 2);
    sch.wanted_items = env_u32("PTRT_WANTED_ITEMS", ctx->resident_waves * per_slot);
    sch.default_chunk = 32;
    sch.max_chunk = env_u32("PTRT_MAX_CHUNK", 0);
    // (the shortest planned launches, 16 to 23 frames, are bound by their costliest tiles' chains of frames: single-frame
    // items there, 1.96 -> 1.76 ms at 16 frames; from 24 frames on items of at least two frames are as good or better)
    sch.min_chunk = env_u32("PTRT_MIN_CHUNK", use_pool ? (spp == 1u ? (launch_frames < 24u ? 1u : 2u) : std::max(4u / spp, 1u)) : 1);
    // Launches of 2..15 frames: one work item per tile AND frame, the items' single-frame sums added in frame order,
    // which is bit for bit the sum frame after frame (0 + x == x) while the frames run side by side instead of one
    // after the other per pixel.  From 16 frames on: items of several frames by measured cost (sums regrouped).
    sch.min_planned_frames = env_u32("PTRT_MIN_PLANNED", 2);
    const uint32_t regroup_frames = (ctx->flags & PT_FLAG_REGROUP_SHORT) ? 4 : 16;
    sch.probe_frames = 4;
    if (env_str("PTRT_NO_PLAN")) sch.min_planned_frames = 0xFFFFFFFFu;  // tuning aid
    sch.use_pool = use_pool;
    sch.next_item = ctx->next_item;
    sch.pool_gstack = pool_gstack;
    sch.n_cus = ctx->n_cus;
    sch.pool_groups = env_u32("PTRT_POOL_GROUPS", 0);
    p.pool_start = env_u32("PTRT_POOL_START", PT_POOL_START);  // (the kernel reads these in tuning builds only)
    p.pool_topup = env_u32("PTRT_POOL_TOPUP", PT_POOL_TOPUP);
    p.pool_dense = env_u32("PTRT_POOL_DENSE", PT_POOL_DENSE);
    p.pool_shade_dense = env_u32("PTRT_POOL_SHADE_DENSE", PT_POOL_SHADE_DENSE);
    p.pool_miss_iters = env_u32("PTRT_POOL_MISS_ITERS", PT_POOL_MISS_ITERS);
    p.pool_miss_lanes = env_u32("PTRT_POOL_MISS_LANES", PT_POOL_MISS_LANES);
    p.pool_steps = env_u32("PTRT_POOL_STEPS", PT_POOL_STEPS);
    p.max_idle = PT_POOL_MAX_IDLE;

    // the per-lane ray counter is 32 bits: bound the frames of one launch
    const uint64_t per_frame = (uint64_t)ctx->settings.samples_per_pixel * ((uint64_t)ctx->settings.depth + PT_MAX_SPHERES);
    if (per_frame >= (1ull << 32)) return fail(ctx, PT_ERR_LIMIT, "pt_render: spp*depth too large");
    const uint32_t max_frames = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull / per_frame, 1u << 20);
    if (max_frames == 0) return fail(ctx, PT_ERR_LIMIT, "pt_render: spp*depth too large");

    if (ctx->window_size) {
        // Display emulation: one launch per frame, each frame quantised into the ring like the shader's
        // rgba8unorm store (compute.wgsl:124-127) and added to the running sums in the same order as usual.
        const uint32_t n_px = ctx->width * ctx->height;
        p.n_frames = 1;
        p.accum = ctx->window_frame;
        p.gstack_cap = 0;
        const size_t gs = use_pool ? 0 : pt_trace_gstack_bytes(ctx->n_tiles, ctx->bvh_depth);
        if (gs) {
            int rc = ensure_gstack(ctx, gs);
            if (rc != PT_OK) return rc;
            p.gstack_cap = ctx->bvh_depth;
            p.gstack_blocks = ctx->n_tiles;
        }
        p.gstack = ctx->gstack;
        sch.items = ctx->items;
        sch.partial = ctx->partial;
        sch.item_cap = ctx->item_cap;
        for (uint32_t f = 0; f < n_frames; f++) {
            // Renderer::render: the copy pass of this frame averages min(sample_size, ready_samples) textures,
            // counted before increment_ready_samples (renderer.rs:539-541, 560, 623-645)
            ctx->window_shown = std::min(ctx->window_size, ctx->ready_samples);
            ctx->ready_samples = std::min(ctx->ready_samples + 1u, ctx->window_size);
            p.time0 = time0 + f * stride;
            PT_HIP(ctx, pt_launch_trace(p, use_pool ? ctx->walk_depth : ctx->bvh_depth, (ctx->flags & PT_FLAG_DETAILED_STATS) != 0, sch, ctx->stream));
            uint32_t *slot = ctx->window_ring + (size_t)(ctx->window_seq % ctx->window_size) * n_px;
            PT_HIP(ctx, pt_launch_window_push(ctx->window_frame, ctx->accum, slot, n_px, ctx->stream));
            ctx->window_seq++;
            ctx->frames++;
            ctx->paths += (uint64_t)p.cols * p.rows * ctx->settings.samples_per_pixel;
            p.aov_prim = nullptr;  // the AOV belongs to the first frame of the call
        }
        return PT_OK;
    }

    const size_t static_mark = static_used;  // every batch carves its tables from here
    uint32_t done = 0;
    while (done < n_frames) {
        const uint32_t batch = std::min(n_frames - done, max_frames);
        p.time0 = time0 + done * stride;
        p.n_frames = batch;
        uint32_t blocks = ctx->n_tiles;
        PtSchedule bs = sch;  // this batch's schedule
        if (batch < regroup_frames && !env_str("PTRT_MAX_CHUNK")) {
            bs.max_chunk = 1;
            bs.min_chunk = 1;
            // one row of partial sums (and, for deep trees, one set of spilled stacks) per tile and frame: above the
            // limit the frames run one after the other, one item per tile, with no such rows at all
            const uint64_t cap = pt_trace_item_cap(ctx->n_tiles, batch, bs);
            const uint64_t bytes = cap * (64u * sizeof(float4) + sizeof(uint4)) + (use_pool ? 0 : pt_trace_gstack_bytes((uint32_t)cap, ctx->bvh_depth));
            const uint64_t limit = use_static ? PT_STATIC_SCRATCH_BYTES - static_mark - 4096u : PT_SHORT_LAUNCH_SCRATCH_LIMIT;
            if (bytes > limit) bs.min_planned_frames = regroup_frames;  // frame after frame, one item per tile
        }
        uint4 *items = ctx->items;
        float4 *partial = ctx->partial;
        uint32_t item_cap = ctx->item_cap;
        static_used = static_mark;
        if (batch >= bs.min_planned_frames) {
            const uint32_t cap = pt_trace_item_cap(ctx->n_tiles, batch, bs);
            blocks = std::max(blocks, cap);
            if (use_static) {
                items = static_cast<uint4 *>(carve((size_t)cap * sizeof(uint4)));
                partial = static_cast<float4 *>(carve((size_t)cap * 64u * sizeof(float4)));
                item_cap = cap;
                if (!items || !partial)
                    return fail(ctx, PT_ERR_LIMIT, "pt_render_frames_static: a launch of %u frames of %u x %u pixels plans up to %u work items (%zu bytes of table and partial sums), beyond the %llu-byte static scratch; render fewer frames per call",
                                batch, ctx->width, ctx->height, cap, (size_t)cap * (64u * sizeof(float4) + sizeof(uint4)), (unsigned long long)PT_STATIC_SCRATCH_BYTES);
            } else if (cap > ctx->item_cap) {
                PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
                dev_free(ctx->items);
                dev_free(ctx->partial);
                ctx->item_cap = 0;
                PT_HIP(ctx, hipMalloc((void **)&ctx->items, (size_t)cap * sizeof(uint4)));
                PT_HIP(ctx, hipMalloc((void **)&ctx->partial, (size_t)cap * 64u * sizeof(float4)));
                ctx->item_cap = cap;
                ctx->scene_generation++;  // the old plan went with the old table
                items = ctx->items;
                partial = ctx->partial;
                item_cap = cap;
            }
            pt_ctx::PlanKey key;
            key.generation = ctx->scene_generation;
            key.n_frames = batch;
            key.wanted_items = bs.wanted_items;
            key.max_chunk = bs.max_chunk;
            key.min_chunk = bs.min_chunk;
            key.use_pool = use_pool;
            // Identical launches one after the other (same scene, image, length; the time differs).  The per-tile
            // kernel plans each of them afresh from the launch before: its items are bound to waves up front, and a
            // plan gone stale costs it dearly (cost moves with the time: on coarse meshes some spans of 192 frames are
            // twice as expensive in some tiles, and a stale plan then took 26-52 ms instead of 15).  The pooled
            // kernel hands items out as slots fall free and loses less to a stale plan (20 ms instead of 15) while
            // skipping the planning and the cost measurement gains it 3-15 %: it plans launches 1 and 2, runs that
            // plan for six launches, measures during the seventh, plans the eighth anew, and so on.
            if (key == ctx->plan_key) {
                const uint32_t launch = ++ctx->plans_for_key;  // 2, 3, ... with this key
                if (use_pool && launch >= 3u && ctx->cost_valid && !env_str("PTRT_ALWAYS_PLAN")) {
                    const uint32_t phase = (launch - 3u) % PT_PLAN_REUSE_PERIOD;
                    bs.reuse_plan = phase + 1u < PT_PLAN_REUSE_PERIOD;
                    bs.measure_cost = phase + 2u == PT_PLAN_REUSE_PERIOD;
                }
            } else {
                ctx->plan_key = key;
                ctx->plans_for_key = 1;
            }
        }
        bs.items = items;
        bs.partial = partial;
        bs.item_cap = item_cap;
        // spill space for traversal stacks deeper than the per-tile kernel keeps in LDS
        const size_t gs = use_pool ? 0 : pt_trace_gstack_bytes(blocks, ctx->bvh_depth);
        p.gstack = ctx->gstack;
        if (gs && use_static) {
            p.gstack = static_cast<uint32_t *>(carve(gs));
            if (!p.gstack)
                return fail(ctx, PT_ERR_LIMIT, "pt_render_frames_static: a BVH of depth %u needs %zu bytes of traversal-stack spill space for this launch, beyond what is left of the %llu-byte static scratch", ctx->bvh_depth, gs, (unsigned long long)PT_STATIC_SCRATCH_BYTES);
        } else if (gs) {
            int rc = ensure_gstack(ctx, gs);
            if (rc != PT_OK) return rc;
            p.gstack = ctx->gstack;
        }
        p.gstack_cap = gs ? ctx->bvh_depth : 0;
        p.gstack_blocks = gs ? blocks : 0;
        PT_HIP(ctx, pt_launch_trace(p, use_pool ? ctx->walk_depth : ctx->bvh_depth, (ctx->flags & PT_FLAG_DETAILED_STATS) != 0, bs, ctx->stream));
        done += batch;
        ctx->frames += batch;
        ctx->paths += (uint64_t)p.cols * p.rows * batch * ctx->settings.samples_per_pixel;
        p.aov_prim = nullptr;  // the AOV belongs to the first frame of the call
    }
    return PT_OK;
}

static int render_frames_impl(pt_ctx *ctx, uint32_t time0, uint32_t stride, uint32_t n_frames, bool use_static)
{
    try {
        return render_frames_unguarded(ctx, time0, stride, n_frames, use_static);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}

int pt_render_frames(pt_ctx *ctx, uint32_t time0, uint32_t stride, uint32_t n_frames) { return render_frames_impl(ctx, time0, stride, n_frames, false); }

int pt_render_frames_static(pt_ctx *ctx, uint32_t time0, uint32_t stride, uint32_t n_frames) { return render_frames_impl(ctx, time0, stride, n_frames, true); }

int pt_render(pt_ctx *ctx, uint32_t time) { return pt_render_frames(ctx, time, 0, 1); }

static int pt_reset_impl(pt_ctx *ctx)
{
    PT_ENTER(ctx);
    const size_t n = (size_t)ctx->width * ctx->height;
    PT_HIP(ctx, hipMemsetAsync(ctx->accum, 0, n * sizeof(float4), ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->counters, 0, PT_COUNTER_STRIPES * PT_COUNTER_PITCH * sizeof(unsigned long long), ctx->stream));
    ctx->frames = 0;
    ctx->paths = 0;
    if (ctx->window_size) ctx->ready_samples = 1;  // reset_ready_samples, renderer.rs:635-637 (the textures keep their contents)
    PT_HIP(ctx, hipMemsetAsync(ctx->next_item + 1, 0, 2 * sizeof(uint32_t), ctx->stream));
    return PT_OK;
}

namespace {
// After a synchronisation: did a wave of the pooled kernel give up waiting on a hand-shake (its bounded
// waits exist so that a defect cannot hang the GPU)?  The image is then incomplete.
int check_gave_up(pt_ctx *ctx)
{
    uint32_t w[2] = {0, 0};  // waves that gave up, self-check bits (PT_KERR_* in pt_kernels.hip)
    PT_HIP(ctx, hipMemcpy(w, ctx->next_item + 1, sizeof w, hipMemcpyDeviceToHost));
    if (w[0] || w[1])
        return fail(ctx, PT_ERR_INTERNAL,
                    "pooled kernel: %u waves gave up waiting%s%s%s%s%s; the accumulated image is incomplete (pt_reset and use PT_FLAG_TILE_KERNEL)", w[0],
                    (w[1] & 1u) ? "; a ring entry was overwritten" : "", (w[1] & 2u) ? "; a claimed ring entry was never filled" : "",
                    (w[1] & 4u) ? "; a ring held a slot id beyond the pool" : "", (w[1] & 8u) ? "; a slot's sums index was out of range" : "",
                    (w[1] & 16u) ? "; the hand-out of pixel jobs ran out of its bounded wait" : "");
    return PT_OK;
}
}  // namespace

static int pt_sync_impl(pt_ctx *ctx)
{
    PT_ENTER(ctx);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_gave_up(ctx);
}

static int pt_read_sum_rgba32f_impl(pt_ctx *ctx, float *out)
{
    PT_ENTER(ctx);
    if (!out) return fail(ctx, PT_ERR_INVALID, "pt_read_sum_rgba32f: null output");
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(out, ctx->accum, (size_t)ctx->width * ctx->height * sizeof(float4), hipMemcpyDeviceToHost));
    return check_gave_up(ctx);
}

static int pt_read_mean_rgba32f_impl(pt_ctx *ctx, float *out)
{
    PT_ENTER(ctx);
    if (!out) return fail(ctx, PT_ERR_INVALID, "pt_read_mean_rgba32f: null output");
    const uint32_t n = ctx->width * ctx->height;
    int rc = ensure_scratch(ctx, (size_t)n * sizeof(float4));
    if (rc != PT_OK) return rc;
    PT_HIP(ctx, pt_launch_resolve(ctx->accum, n, static_cast<float4 *>(ctx->scratch), nullptr, 0, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(out, ctx->scratch, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    return check_gave_up(ctx);
}

static int pt_read_rgba8_impl(pt_ctx *ctx, uint8_t *out)
{
    PT_ENTER(ctx);
    if (!out) return fail(ctx, PT_ERR_INVALID, "pt_read_rgba8: null output");
    const uint32_t n = ctx->width * ctx->height;
    int rc = ensure_scratch(ctx, (size_t)n * sizeof(float4));
    if (rc != PT_OK) return rc;
    PT_HIP(ctx, pt_launch_resolve(ctx->accum, n, nullptr, static_cast<uint32_t *>(ctx->scratch), 1, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(out, ctx->scratch, (size_t)n * 4u, hipMemcpyDeviceToHost));
    return check_gave_up(ctx);
}

static int pt_set_window_impl(pt_ctx *ctx, uint32_t sample_size)
{
    PT_ENTER(ctx);
    if (sample_size > 256u) return fail(ctx, PT_ERR_LIMIT, "pt_set_window: sample_size %u exceeds MAX_NUMBER_OF_SAMPLES (256)", sample_size);
    if (ctx->accum != ctx->accum_owned && sample_size) return fail(ctx, PT_ERR_STATE, "pt_set_window: not available while pt_bind_accum is in effect");
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(ctx->window_ring);
    dev_free(ctx->window_frame);
    ctx->window_size = 0;
    ctx->ready_samples = 0;
    ctx->window_shown = 0;
    ctx->window_seq = 0;
    if (sample_size == 0) return PT_OK;
    const size_t n = (size_t)ctx->width * ctx->height;
    int rc = dev_alloc(ctx, &ctx->window_ring, n * sample_size);
    if (rc == PT_OK) rc = dev_alloc(ctx, &ctx->window_frame, n);
    if (rc != PT_OK) {
        dev_free(ctx->window_ring);
        dev_free(ctx->window_frame);
        return rc;
    }
    // wgpu textures start zeroed
    PT_HIP(ctx, hipMemsetAsync(ctx->window_ring, 0, n * sample_size * sizeof(uint32_t), ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->window_frame, 0, n * sizeof(float4), ctx->stream));
    ctx->window_size = sample_size;
    return PT_OK;
}

static int pt_read_window_impl(pt_ctx *ctx, float *rgba32f_out, uint8_t *rgba8_out, uint32_t *frames_shown)
{
    PT_ENTER(ctx);
    if (!ctx->window_size) return fail(ctx, PT_ERR_STATE, "pt_read_window: no window (pt_set_window)");
    if (frames_shown) *frames_shown = ctx->window_shown;
    if (!rgba32f_out && !rgba8_out) return PT_OK;
    if (ctx->window_shown == 0)
        return fail(ctx, PT_ERR_STATE, "pt_read_window: the last frame's copy pass averaged 0 textures (the reference divides 0 by 0 on its first frame)");
    const uint32_t n = ctx->width * ctx->height;
    int rc = ensure_scratch(ctx, (size_t)n * (sizeof(float4) + sizeof(uint32_t)));
    if (rc != PT_OK) return rc;
    float4 *out_f = static_cast<float4 *>(ctx->scratch);
    uint32_t *out_u8 = reinterpret_cast<uint32_t *>(out_f + n);
    const uint32_t newest = (uint32_t)((ctx->window_seq - 1u) % ctx->window_size);
    PT_HIP(ctx, pt_launch_window_mean(ctx->window_ring, n, ctx->window_size, newest, ctx->window_shown, rgba32f_out ? out_f : nullptr,
                                      rgba8_out ? out_u8 : nullptr, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (rgba32f_out) PT_HIP(ctx, hipMemcpy(rgba32f_out, out_f, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    if (rgba8_out) PT_HIP(ctx, hipMemcpy(rgba8_out, out_u8, (size_t)n * 4u, hipMemcpyDeviceToHost));
    return check_gave_up(ctx);
}

static int pt_read_aov_impl(pt_ctx *ctx, uint32_t *prim, float *t)
{
    PT_ENTER(ctx);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t n = (size_t)ctx->width * ctx->height;
    if (prim) PT_HIP(ctx, hipMemcpy(prim, ctx->aov_prim, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (t) PT_HIP(ctx, hipMemcpy(t, ctx->aov_t, n * sizeof(float), hipMemcpyDeviceToHost));
    return PT_OK;
}

static int pt_read_sky_tiles_impl(pt_ctx *ctx, uint32_t *out, size_t capacity, uint32_t *n_tiles)
{
    PT_ENTER(ctx);
    if (n_tiles) *n_tiles = ctx->n_tiles;
    if (!out) return n_tiles ? PT_OK : fail(ctx, PT_ERR_INVALID, "pt_read_sky_tiles: null output");
    if (capacity < ctx->n_tiles) return fail(ctx, PT_ERR_INVALID, "pt_read_sky_tiles: room for %zu tiles, the image has %u", capacity, ctx->n_tiles);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(out, ctx->tile_sky, (size_t)ctx->n_tiles * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return PT_OK;
}

static int pt_stats_impl(pt_ctx *ctx, pt_stats_t *out)
{
    PT_ENTER(ctx);
    if (!out) return fail(ctx, PT_ERR_INVALID, "pt_stats: null output");
    static_assert(PT_COUNTER_PITCH >= PT_NUM_COUNTERS, "a stripe holds every counter");
    std::vector<unsigned long long> stripes(PT_COUNTER_STRIPES * PT_COUNTER_PITCH);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(stripes.data(), ctx->counters, stripes.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long c[PT_NUM_COUNTERS] = {};
    for (uint32_t s = 0; s < PT_COUNTER_STRIPES; s++)
        for (uint32_t k = 0; k < PT_NUM_COUNTERS; k++) c[k] += stripes[s * PT_COUNTER_PITCH + k];
    out->rays = c[0];
    out->paths = ctx->paths;
    out->interior = c[2];
    out->tri_tests = c[3];
    out->tri_accept = c[4];
    out->sphere_tests = c[5];
    out->sky_lookups = c[6];
    out->bounces = c[7];
    out->sphere_accept = c[8];
    out->frames = ctx->frames;
    out->bvh_depth = ctx->bvh_depth;
    out->walks = c[9];
    out->sky_loop_beside = (ctx->main_stream && ctx->side_stream) ? 1u : 0u;
    out->guarded = c[10];
    out->rewalks = c[11];
    out->guard_nodes = ctx->guard_on ? ctx->guard_nodes : 0u;
    return check_gave_up(ctx);
}

void *pt_accum_device_ptr(pt_ctx *ctx) { return ctx ? ctx->accum : nullptr; }

static int pt_bind_accum_impl(pt_ctx *ctx, void *device_ptr)
{
    PT_ENTER(ctx);
    if (ctx->window_size && device_ptr) return fail(ctx, PT_ERR_STATE, "pt_bind_accum: not available while a window is set (pt_set_window)");
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (device_ptr && (reinterpret_cast<uintptr_t>(device_ptr) & 15u)) return fail(ctx, PT_ERR_INVALID, "pt_bind_accum: pointer not 16-byte aligned");
    ctx->accum = device_ptr ? static_cast<float4 *>(device_ptr) : ctx->accum_owned;
    return PT_OK;
}

static int pt_trace_rays_impl(pt_ctx *ctx, const float *origins, const float *dirs, uint32_t n, uint32_t *prim, float *t, float *p, float *normal, uint32_t *front,
                             uint32_t *walked)
{
    PT_ENTER(ctx);
    if (n == 0) return PT_OK;
    if (!origins || !dirs) return fail(ctx, PT_ERR_INVALID, "pt_trace_rays: null rays");
    if (n > (1u << 28)) return fail(ctx, PT_ERR_LIMIT, "pt_trace_rays: more than 2^28 rays");
    // one allocation: origins | dirs | prim | t | p | normal | front | walked
    const size_t fl = (size_t)n * sizeof(float);
    const size_t total = 3 * fl * 2 + fl + fl + 3 * fl * 2 + fl + fl;
    char *buf = nullptr;
    PT_HIP(ctx, hipMalloc((void **)&buf, total));
    PtRayQuery q{};
    q.sc = device_scene(ctx);
    char *cur = buf;
    q.origins = reinterpret_cast<float *>(cur); cur += 3 * fl;
    q.dirs = reinterpret_cast<float *>(cur); cur += 3 * fl;
    q.prim = reinterpret_cast<uint32_t *>(cur); cur += fl;
    q.t = reinterpret_cast<float *>(cur); cur += fl;
    q.p = reinterpret_cast<float *>(cur); cur += 3 * fl;
    q.normal = reinterpret_cast<float *>(cur); cur += 3 * fl;
    q.front = reinterpret_cast<uint32_t *>(cur); cur += fl;
    q.walked = reinterpret_cast<uint32_t *>(cur);
    q.n = n;
    uint32_t *gs_mem = nullptr;
    const uint32_t depth = ctx->bvh_depth;  // (a walk of the second tree's four-child form never needs more stack than it finds)
    const size_t gs = pt_ray_query_gstack_bytes(n, depth);
    hipError_t e = hipSuccess;
    if (gs) e = hipMalloc((void **)&gs_mem, gs);
    q.gstack = gs_mem;
    q.gstack_cap = gs ? depth : 0;
    if (e == hipSuccess) e = hipMemcpyAsync((void *)q.origins, origins, 3 * fl, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync((void *)q.dirs, dirs, 3 * fl, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = pt_launch_ray_query(q, depth, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && prim) e = hipMemcpy(prim, q.prim, fl, hipMemcpyDeviceToHost);
    if (e == hipSuccess && t) e = hipMemcpy(t, q.t, fl, hipMemcpyDeviceToHost);
    if (e == hipSuccess && p) e = hipMemcpy(p, q.p, 3 * fl, hipMemcpyDeviceToHost);
    if (e == hipSuccess && normal) e = hipMemcpy(normal, q.normal, 3 * fl, hipMemcpyDeviceToHost);
    if (e == hipSuccess && front) e = hipMemcpy(front, q.front, fl, hipMemcpyDeviceToHost);
    if (e == hipSuccess && walked) e = hipMemcpy(walked, q.walked, fl, hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    if (gs_mem) (void)hipFree(gs_mem);
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_HIP, "pt_trace_rays: %s", hipGetErrorString(e));
    return PT_OK;
}


// Nothing may propagate out of an extern "C" function: a C or Rust caller cannot catch it.
int pt_create(int device, uint32_t width, uint32_t height, pt_ctx **out)
{
    try {
        return pt_create_impl(device, width, height, out);
    } catch (const std::bad_alloc &) {
        return fail_static(nullptr, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(nullptr, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_set_stream(pt_ctx *ctx, void *hip_stream)
{
    try {
        return pt_set_stream_impl(ctx, hip_stream);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_set_flags(pt_ctx *ctx, uint32_t flags)
{
    try {
        return pt_set_flags_impl(ctx, flags);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_set_mesh(pt_ctx *ctx, const void *tris_v, uint32_t n_tris, const void *nodes_v, uint32_t n_nodes, const uint32_t *tri_idx, uint32_t n_idx)
{
    try {
        return pt_set_mesh_impl(ctx, tris_v, n_tris, nodes_v, n_nodes, tri_idx, n_idx);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_set_sky_cube(pt_ctx *ctx, const float *rgba32f, uint32_t size)
{
    try {
        return pt_set_sky_cube_impl(ctx, rgba32f, size);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_set_sky_equirect(pt_ctx *ctx, const float *rgba32f, uint32_t w, uint32_t h, uint32_t cube_size)
{
    try {
        return pt_set_sky_equirect_impl(ctx, rgba32f, w, h, cube_size);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_read_sky_cube(pt_ctx *ctx, float *out, size_t capacity_floats, uint32_t *size)
{
    try {
        return pt_read_sky_cube_impl(ctx, out, capacity_floats, size);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_set_camera(pt_ctx *ctx, const void *camera64)
{
    try {
        return pt_set_camera_impl(ctx, camera64);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_set_spheres(pt_ctx *ctx, const void *sphere_data8208)
{
    try {
        return pt_set_spheres_impl(ctx, sphere_data8208);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_set_settings(pt_ctx *ctx, const void *settings16)
{
    try {
        return pt_set_settings_impl(ctx, settings16);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_reset(pt_ctx *ctx)
{
    try {
        return pt_reset_impl(ctx);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_sync(pt_ctx *ctx)
{
    try {
        return pt_sync_impl(ctx);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_read_sum_rgba32f(pt_ctx *ctx, float *out)
{
    try {
        return pt_read_sum_rgba32f_impl(ctx, out);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_read_mean_rgba32f(pt_ctx *ctx, float *out)
{
    try {
        return pt_read_mean_rgba32f_impl(ctx, out);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_read_rgba8(pt_ctx *ctx, uint8_t *out)
{
    try {
        return pt_read_rgba8_impl(ctx, out);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_set_window(pt_ctx *ctx, uint32_t sample_size)
{
    try {
        return pt_set_window_impl(ctx, sample_size);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_read_window(pt_ctx *ctx, float *rgba32f_out, uint8_t *rgba8_out, uint32_t *frames_shown)
{
    try {
        return pt_read_window_impl(ctx, rgba32f_out, rgba8_out, frames_shown);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_read_aov(pt_ctx *ctx, uint32_t *prim, float *t)
{
    try {
        return pt_read_aov_impl(ctx, prim, t);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_read_sky_tiles(pt_ctx *ctx, uint32_t *out, size_t capacity, uint32_t *n_tiles)
{
    try {
        return pt_read_sky_tiles_impl(ctx, out, capacity, n_tiles);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_stats(pt_ctx *ctx, pt_stats_t *out)
{
    try {
        return pt_stats_impl(ctx, out);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_bind_accum(pt_ctx *ctx, void *device_ptr)
{
    try {
        return pt_bind_accum_impl(ctx, device_ptr);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_trace_rays(pt_ctx *ctx, const float *origins, const float *dirs, uint32_t n, uint32_t *prim, float *t, float *p, float *normal, uint32_t *front)
{
    try {
        return pt_trace_rays_impl(ctx, origins, dirs, n, prim, t, p, normal, front, nullptr);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
int pt_trace_rays_walked(pt_ctx *ctx, const float *origins, const float *dirs, uint32_t n, uint32_t *prim, float *t, float *p, float *normal, uint32_t *front,
                         uint32_t *walked)
{
    try {
        return pt_trace_rays_impl(ctx, origins, dirs, n, prim, t, p, normal, front, walked);
    } catch (const std::bad_alloc &) {
        return fail_static(ctx, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return fail_static(ctx, PT_ERR_INTERNAL, "internal error");
    }
}
}  // extern "C"
