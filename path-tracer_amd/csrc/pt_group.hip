// pt_group.hip -- several GPUs of one node behind one handle: the sample split and the sum of the per-GPU
// accumulation buffers for a host that drives all GPUs from one process (the Rust host calls this through the C ABI;
// hosts with one process per GPU use pt_accum_device_ptr / pt_bind_accum and their own collective instead).
//
// Frames are independent given their `time` value (compute.wgsl:85: the seed is (x, y, x + time, y + time)), so a job
// of F frames starting at time0 is dealt round-robin: member g of G renders time0 + g, time0 + g + G, ...  Every member
// is an ordinary context (pt_ctx) on its own GPU and stream with the scene replicated; nothing is exchanged while
// rendering.  pt_group_reduce then moves every member's sums onto one member.  Two ways, same result up to fp32
// summation order:
//   * peer copies (default): every other member's buffer is copied into a staging buffer on the root's GPU -- all
//     copies at once, each on its member's stream, so they travel over different xGMI links -- and the root adds the
//     staged buffers in member order.  The order is fixed, so the sum is bitwise the same from run to run.
//   * RCCL (PT_GROUP_RCCL): one ncclReduce(sum, float) per member inside ncclGroupStart/End.  librccl is loaded with
//     dlopen when the first such group is created, so the library itself has no link-time dependency on it.
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/ptrt.h"

namespace {

// ---- the few RCCL entry points the reduce needs (declared here: rccl.h is not needed to build the library) ----------
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;  // 0 = ncclSuccess
constexpr int kNcclFloat32 = 7, kNcclSum = 0;
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &why)
    {
        if (handle) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) { why = "librccl.so not found"; return false; }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(handle, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(handle, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(handle, "ncclGroupEnd"));
        Reduce = reinterpret_cast<decltype(Reduce)>(dlsym(handle, "ncclReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Reduce || !GetErrorString) {
            why = "librccl.so lacks an expected symbol";
            return false;
        }
        return true;
    }
} g_rccl;

thread_local std::string g_group_create_error;

// accum += staged, element by element (float4 = one pixel's sums and frame count)
__global__ __launch_bounds__(256) void group_add_kernel(float4 *accum, const float4 *staged, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 a = accum[i];
    const float4 b = staged[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    accum[i] = a;
}

}  // namespace

struct pt_group {
    std::vector<pt_ctx *> members;
    std::vector<int> devices;
    std::vector<hipStream_t> streams;   // one per member, on its device
    std::vector<hipEvent_t> events;     // one per member: "this member's buffer has arrived / is ready"
    std::vector<float4 *> staging;      // per member: its place on the current root's GPU (grown on demand)
    int staging_root = -1;
    std::vector<ncclComm_t> comms;      // PT_GROUP_RCCL only
    uint32_t width = 0, height = 0, flags = 0;
    std::string err;
};

namespace {

int gfail(pt_group *g, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    try {
        if (g) g->err = buf;
        else g_group_create_error = buf;
    } catch (...) {
    }
    return code;
}

struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

// a member's failure, with its own message
int member_fail(pt_group *g, int m, int code, const char *what)
{
    return gfail(g, code, "%s: member %d (GPU %d): %s", what, m, g->devices[(size_t)m], pt_last_error(g->members[(size_t)m]));
}

void free_staging(pt_group *g)
{
    if (g->staging_root < 0) return;
    (void)hipSetDevice(g->devices[(size_t)g->staging_root]);
    for (float4 *&p : g->staging) {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    g->staging_root = -1;
}

#define PT_G_EACH(g, what, call)                                  \
    do {                                                          \
        if (!(g)) return gfail(nullptr, PT_ERR_INVALID, "null group"); \
        for (int m_ = 0; m_ < (int)(g)->members.size(); m_++) {   \
            pt_ctx *ctx = (g)->members[(size_t)m_];               \
            const int rc_ = (call);                               \
            if (rc_ != PT_OK) return member_fail(g, m_, rc_, what); \
        }                                                         \
        return PT_OK;                                             \
    } while (0)

int group_create_impl(const int *devices, int n, uint32_t width, uint32_t height, uint32_t flags, pt_group **out)
{
    if (!out) return gfail(nullptr, PT_ERR_INVALID, "pt_group_create: null output pointer");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return gfail(nullptr, PT_ERR_INVALID, "pt_group_create: need 1 to 64 devices");
    if (flags & ~PT_GROUP_RCCL) return gfail(nullptr, PT_ERR_INVALID, "pt_group_create: unknown flag bits 0x%x", flags);
    DeviceGuard guard;
    pt_group *g = new (std::nothrow) pt_group();
    if (!g) return gfail(nullptr, PT_ERR_NOMEM, "pt_group_create: out of host memory");
    g->width = width;
    g->height = height;
    g->flags = flags;
    g->devices.assign(devices, devices + n);
    g->staging.assign((size_t)n, nullptr);
    int rc = PT_OK;
    for (int m = 0; m < n && rc == PT_OK; m++) {
        pt_ctx *ctx = nullptr;
        rc = pt_create(devices[m], width, height, &ctx);
        if (rc != PT_OK) { gfail(nullptr, rc, "pt_group_create: member %d (GPU %d): %s", m, devices[m], pt_last_error(nullptr)); break; }
        g->members.push_back(ctx);
        hipStream_t s = nullptr;
        hipEvent_t e = nullptr;
        hipError_t he = hipSetDevice(devices[m]);
        if (he == hipSuccess) he = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (he == hipSuccess) g->streams.push_back(s);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (he == hipSuccess) g->events.push_back(e);
        if (he != hipSuccess) { rc = gfail(nullptr, PT_ERR_HIP, "pt_group_create: member %d (GPU %d): %s", m, devices[m], hipGetErrorString(he)); break; }
        rc = pt_set_stream(ctx, s);
        if (rc != PT_OK) gfail(nullptr, rc, "pt_group_create: member %d: %s", m, pt_last_error(ctx));
    }
    // direct copies between the members' GPUs where the hardware allows (a refusal only makes copies take the long way)
    for (int a = 0; a < n && rc == PT_OK; a++)
        for (int b = 0; b < n; b++) {
            if (devices[a] == devices[b]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can && hipSetDevice(devices[a]) == hipSuccess) {
                const hipError_t he = hipDeviceEnablePeerAccess(devices[b], 0);
                if (he != hipSuccess) (void)hipGetLastError();  // (already enabled: fine)
            }
        }
    if (rc == PT_OK && (flags & PT_GROUP_RCCL)) {
        std::string why;
        if (!g_rccl.load(why)) rc = gfail(nullptr, PT_ERR_STATE, "pt_group_create: PT_GROUP_RCCL: %s", why.c_str());
        else {
            g->comms.assign((size_t)n, nullptr);
            const ncclResult_t nr = g_rccl.CommInitAll(g->comms.data(), n, devices);
            if (nr != 0) {
                g->comms.clear();
                rc = gfail(nullptr, PT_ERR_HIP, "pt_group_create: ncclCommInitAll: %s", g_rccl.GetErrorString(nr));
            }
        }
    }
    if (rc != PT_OK) {
        const std::string keep = g_group_create_error;
        pt_group_destroy(g);
        g_group_create_error = keep;
        return rc;
    }
    *out = g;
    return PT_OK;
}

// Member g's share of frames time0 .. time0 + n_frames - 1: time0 + g, stride G, count.
uint32_t share_count(uint32_t n_frames, uint32_t g, uint32_t G) { return n_frames > g ? (n_frames - g + G - 1u) / G : 0u; }

int group_reduce_impl(pt_group *g, int root)
{
    if (!g) return gfail(nullptr, PT_ERR_INVALID, "null group");
    const int n = (int)g->members.size();
    if (root < 0 || root >= n) return gfail(g, PT_ERR_INVALID, "pt_group_reduce: root %d out of range (%d members)", root, n);
    if (n == 1 && !(g->flags & PT_GROUP_RCCL)) return PT_OK;  // (a one-member RCCL group still runs its collective: a rehearsal on one GPU)
    DeviceGuard guard;
    const size_t px = (size_t)g->width * g->height, bytes = px * sizeof(float4);
    std::vector<void *> acc((size_t)n);
    for (int m = 0; m < n; m++) {
        acc[(size_t)m] = pt_accum_device_ptr(g->members[(size_t)m]);
        if (!acc[(size_t)m]) return gfail(g, PT_ERR_STATE, "pt_group_reduce: member %d has no accumulation buffer", m);
    }
#define PT_G_HIP(call)                                                                                   \
    do {                                                                                                 \
        const hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return gfail(g, e_ == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_HIP, "pt_group_reduce: %s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
    if (g->flags & PT_GROUP_RCCL) {
        // every member's reduce on its own stream, behind its rendering; the root reduces in place
        ncclResult_t nr = g_rccl.GroupStart();
        for (int m = 0; m < n && nr == 0; m++)
            nr = g_rccl.Reduce(acc[(size_t)m], acc[(size_t)m], px * 4u, kNcclFloat32, kNcclSum, root, g->comms[(size_t)m], g->streams[(size_t)m]);
        const ncclResult_t ne = g_rccl.GroupEnd();
        if (nr == 0) nr = ne;
        if (nr != 0) return gfail(g, PT_ERR_HIP, "pt_group_reduce: RCCL: %s", g_rccl.GetErrorString(nr));
    } else {
        if (g->staging_root != root) {
            free_staging(g);
            PT_G_HIP(hipSetDevice(g->devices[(size_t)root]));
            g->staging_root = root;
            for (int m = 0; m < n; m++)
                if (m != root) PT_G_HIP(hipMalloc((void **)&g->staging[(size_t)m], bytes));
        }
        // the staging buffers may still be read by the previous reduce's additions: the copies wait for the root's stream
        PT_G_HIP(hipSetDevice(g->devices[(size_t)root]));
        PT_G_HIP(hipEventRecord(g->events[(size_t)root], g->streams[(size_t)root]));
        for (int m = 0; m < n; m++) {
            if (m == root) continue;
            PT_G_HIP(hipSetDevice(g->devices[(size_t)m]));
            PT_G_HIP(hipStreamWaitEvent(g->streams[(size_t)m], g->events[(size_t)root], 0));
            PT_G_HIP(hipMemcpyPeerAsync(g->staging[(size_t)m], g->devices[(size_t)root], acc[(size_t)m], g->devices[(size_t)m], bytes, g->streams[(size_t)m]));
            PT_G_HIP(hipEventRecord(g->events[(size_t)m], g->streams[(size_t)m]));
        }
        PT_G_HIP(hipSetDevice(g->devices[(size_t)root]));
        for (int m = 0; m < n; m++) {  // member order: the sum's order is fixed
            if (m == root) continue;
            PT_G_HIP(hipStreamWaitEvent(g->streams[(size_t)root], g->events[(size_t)m], 0));
            hipLaunchKernelGGL(group_add_kernel, dim3((unsigned)((px + 255u) / 256u)), dim3(256), 0, g->streams[(size_t)root],
                               static_cast<float4 *>(acc[(size_t)root]), g->staging[(size_t)m], px);
            PT_G_HIP(hipGetLastError());
        }
    }
    // what has been handed over is not counted again: the other members start from zero
    for (int m = 0; m < n; m++) {
        if (m == root) continue;
        PT_G_HIP(hipSetDevice(g->devices[(size_t)m]));
        PT_G_HIP(hipMemsetAsync(acc[(size_t)m], 0, bytes, g->streams[(size_t)m]));
    }
#undef PT_G_HIP
    return PT_OK;
}

}  // namespace

extern "C" {

const char *pt_group_last_error(const pt_group *g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int pt_group_create(const int *devices, int n, uint32_t width, uint32_t height, uint32_t flags, pt_group **out)
{
    try {
        return group_create_impl(devices, n, width, height, flags, out);
    } catch (const std::bad_alloc &) {
        return gfail(nullptr, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return gfail(nullptr, PT_ERR_INTERNAL, "internal error");
    }
}

void pt_group_destroy(pt_group *g)
{
    if (!g) return;
    DeviceGuard guard;
    for (size_t m = 0; m < g->members.size(); m++) (void)pt_sync(g->members[m]);
    for (ncclComm_t c : g->comms)
        if (c) (void)g_rccl.CommDestroy(c);
    free_staging(g);
    for (size_t m = 0; m < g->members.size(); m++) pt_destroy(g->members[m]);  // (synchronises the member's stream first)
    for (size_t m = 0; m < g->streams.size(); m++) {  // (only members that came into being: their devices are valid)
        if (hipSetDevice(g->devices[m]) != hipSuccess) continue;
        if (m < g->events.size() && g->events[m]) (void)hipEventDestroy(g->events[m]);
        if (g->streams[m]) (void)hipStreamDestroy(g->streams[m]);
    }
    (void)hipGetLastError();  // nothing here may leave a stale status behind for the caller's next launch check
    delete g;
}

int pt_group_size(const pt_group *g) { return g ? (int)g->members.size() : 0; }

pt_ctx *pt_group_member(pt_group *g, int i) { return (g && i >= 0 && i < (int)g->members.size()) ? g->members[(size_t)i] : nullptr; }

int pt_group_set_flags(pt_group *g, uint32_t flags) { PT_G_EACH(g, "pt_group_set_flags", pt_set_flags(ctx, flags)); }
int pt_group_set_mesh(pt_group *g, const void *tris, uint32_t n_tris, const void *nodes, uint32_t n_nodes, const uint32_t *tri_idx, uint32_t n_idx)
{
    PT_G_EACH(g, "pt_group_set_mesh", pt_set_mesh(ctx, tris, n_tris, nodes, n_nodes, tri_idx, n_idx));
}
int pt_group_set_sky_cube(pt_group *g, const float *rgba32f, uint32_t size) { PT_G_EACH(g, "pt_group_set_sky_cube", pt_set_sky_cube(ctx, rgba32f, size)); }
int pt_group_set_sky_equirect(pt_group *g, const float *rgba32f, uint32_t w, uint32_t h, uint32_t cube_size)
{
    PT_G_EACH(g, "pt_group_set_sky_equirect", pt_set_sky_equirect(ctx, rgba32f, w, h, cube_size));
}
int pt_group_set_camera(pt_group *g, const void *camera64) { PT_G_EACH(g, "pt_group_set_camera", pt_set_camera(ctx, camera64)); }
int pt_group_set_spheres(pt_group *g, const void *sphere_data8208) { PT_G_EACH(g, "pt_group_set_spheres", pt_set_spheres(ctx, sphere_data8208)); }
int pt_group_set_settings(pt_group *g, const void *settings16) { PT_G_EACH(g, "pt_group_set_settings", pt_set_settings(ctx, settings16)); }
int pt_group_reset(pt_group *g) { PT_G_EACH(g, "pt_group_reset", pt_reset(ctx)); }
int pt_group_sync(pt_group *g) { PT_G_EACH(g, "pt_group_sync", pt_sync(ctx)); }

int pt_group_render_frames(pt_group *g, uint32_t time0, uint32_t n_frames)
{
    if (!g) return gfail(nullptr, PT_ERR_INVALID, "null group");
    const uint32_t G = (uint32_t)g->members.size();
    for (uint32_t m = 0; m < G; m++) {
        const uint32_t count = share_count(n_frames, m, G);
        if (count == 0) continue;
        const int rc = pt_render_frames(g->members[m], time0 + m, G, count);  // asynchronous on the member's stream
        if (rc != PT_OK) return member_fail(g, (int)m, rc, "pt_group_render_frames");
    }
    return PT_OK;
}

int pt_group_reduce(pt_group *g, int root)
{
    try {
        return group_reduce_impl(g, root);
    } catch (const std::bad_alloc &) {
        return gfail(g, PT_ERR_NOMEM, "out of memory");
    } catch (...) {
        return gfail(g, PT_ERR_INTERNAL, "internal error");
    }
}

int pt_group_stats(pt_group *g, pt_stats_t *out)
{
    if (!g) return gfail(nullptr, PT_ERR_INVALID, "null group");
    if (!out) return gfail(g, PT_ERR_INVALID, "pt_group_stats: null output");
    std::memset(out, 0, sizeof *out);
    out->sky_loop_beside = 1;
    for (int m = 0; m < (int)g->members.size(); m++) {
        pt_stats_t s;
        const int rc = pt_stats(g->members[(size_t)m], &s);
        if (rc != PT_OK) return member_fail(g, m, rc, "pt_group_stats");
        out->rays += s.rays; out->paths += s.paths; out->interior += s.interior; out->tri_tests += s.tri_tests;
        out->tri_accept += s.tri_accept; out->sphere_tests += s.sphere_tests; out->sky_lookups += s.sky_lookups;
        out->bounces += s.bounces; out->sphere_accept += s.sphere_accept; out->frames += s.frames; out->walks += s.walks;
        out->guarded += s.guarded; out->rewalks += s.rewalks;
        out->bvh_depth = s.bvh_depth;
        out->guard_nodes = s.guard_nodes;
        out->sky_loop_beside &= s.sky_loop_beside;
    }
    return PT_OK;
}

}  // extern "C"
