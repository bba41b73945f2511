/*
 * ptrt.h -- C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * The reference renderer has no FFI layer; its operator interface is the Rust struct
 * `Renderer` (src/renderer.rs:13-29) and the bind group it feeds to the compute shader
 * (shaders/compute.wgsl:72-81).  Each entry point below names the reference interface it
 * replaces.  All byte layouts are the reference's own #[repr(C)] Pod structs, passed as
 * `const void*` so a Rust caller hands over `bytemuck::bytes_of(..)` unchanged:
 *
 *   camera   64 B    CameraBuffer       src/scene/camera.rs:73-84
 *   settings 16 B    Settings           src/renderer.rs:604-611
 *   spheres  8208 B  SphereDataBuffer   src/scene/sphere.rs:105-111
 *   triangle 112 B   TriangleBuffer     src/model.rs:51-68
 *   node     32 B    Node               src/scene/bvh.rs:7-14
 *
 * Conventions: every function returns PT_OK (0) or a negative pt_status; nothing throws,
 * aborts or prints.  pt_last_error() gives the message of the last failure on that
 * context (or, with NULL, of the last failed pt_create on this thread).  A context is
 * bound to one GPU and must be used from one host thread at a time.  Calls that launch
 * work are asynchronous on the context's stream unless they return data to the host.
 * Launches of two frames and more may run part of their kernels on a pair of streams that the
 * library keeps per GPU (the first pt_create on a GPU creates them and runs two
 * half-millisecond-bounded test kernels on them; they stay until the process ends and are
 * shared by every context of that GPU); they are forked from and joined back into the
 * context's stream inside the call, so ordering on the context's stream is all a caller has
 * to think about.
 */
#ifndef PTRT_H
#define PTRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_ctx pt_ctx;

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1,   /* bad argument (null pointer, zero size, malformed BVH ...) */
    PT_ERR_HIP = -2,       /* a HIP runtime call failed; message carries hipGetErrorString */
    PT_ERR_NOMEM = -3,     /* host or device allocation failed */
    PT_ERR_STATE = -4,     /* call not valid in the context's current state */
    PT_ERR_LIMIT = -5,     /* input exceeds a documented limit */
    PT_ERR_INTERNAL = -6   /* a kernel reported a defect in itself (never expected; the image is incomplete) */
} pt_status;

/* pt_set_flags bits */
#define PT_FLAG_REFERENCE_DISPATCH 1u /* render only floor(w/16)*16 x floor(h/16)*16 pixels, as
                                         dispatch_workgroups(w/16, h/16, 1) does (renderer.rs:565-569) */
#define PT_FLAG_AOV 2u                /* record the primary hit of each pixel's first sample */
#define PT_FLAG_DETAILED_STATS 4u     /* count node steps / triangle tests too (slower kernel) */
/* Two implementations of the megakernel give the same image bit for bit (short launches) or up to fp32
   summation order (fused launches of 16 frames or more).  By default the library picks by mesh size,
   samples per frame and launch length (the pooled kernel for launches of many samples on meshes). */
#define PT_FLAG_TILE_KERNEL 8u        /* always the one-wave-per-tile kernel */
#define PT_FLAG_POOL_KERNEL 16u       /* always the pooled kernel (where it can run: depth >= 1, image <= 65535^2) */
/* Extension, off by default: triangles shade with the albedo and material of their own records
 * (TriangleBuffer, src/model.rs:66-67, 86-87) instead of the white Lambertian the shader gives every
 * triangle (compute.wgsl:359-367). */
#define PT_FLAG_TRIANGLE_MATERIALS 32u
/* Off by default: launches of 4 to 15 frames are split into work items of several frames like longer ones (sums equal
 * to separate dispatches up to fp32 summation order instead of bit for bit).  For hosts that render a few frames per
 * display frame and want them sooner: 8 frames of the 69 k-triangle 1080p scene 1.9 -> 1.2 ms, 15 frames 2.8 -> 1.7.
 * No counterpart in the reference, which renders one frame per dispatch. */
#define PT_FLAG_REGROUP_SHORT 64u
/* Off by default, and outside the bit-for-bit contract with the reference: pt_set_mesh (called while the flag is set)
 * ignores the caller's nodes and index permutation and walks a tree of its own over the same triangles
 * (pt_bvh_build_sah).  Primary-hit ids, t and radiance then agree with the reference tree's except on the ambiguous
 * cases named at pt_bvh_build_sah (none in 2 million primary rays of the 69 k-triangle scene; the converged images agree
 * far below 1e-3 RMSE), for about a third fewer traversal steps.  No counterpart in the reference. */
#define PT_FLAG_REBUILD_BVH 128u
/* Off by default: every ray walks the caller's tree only.  By default pt_set_mesh (for meshes of 128 triangles or more
 * whose triangles are small against the mesh) also builds a tree of its own, laid out so that a walk of it cannot lose a
 * hit, and every triangle hit found on it is checked to be the one a walk of the caller's tree returns; a ray that fails
 * the check, or starts too far from the mesh, walks the caller's tree.  Both megakernels and pt_trace_rays do this; results
 * are bit for bit the same either way (PT_FLAG_DETAILED_STATS always walks the caller's tree: its counters are that
 * walk's).  The flag is there to measure the difference and to save the second tree's memory (about as much again as the
 * first): set before pt_set_mesh it keeps the tree from being built, set afterwards it only keeps rays off it. */
#define PT_FLAG_REFERENCE_WALK 256u

/* Primitive ids reported by pt_read_aov / pt_trace_rays */
#define PT_PRIM_NONE 0xFFFFFFFFu      /* miss */
#define PT_PRIM_SPHERE 0x80000000u    /* sphere i is PT_PRIM_SPHERE | i; triangles are their index */

typedef struct pt_stats_t {
    uint64_t rays;          /* hitScene calls (compute.wgsl:138): the unit of Mrays/s */
    uint64_t paths;         /* rayColor calls */
    uint64_t interior;      /* interior-node steps          (PT_FLAG_DETAILED_STATS only) */
    uint64_t tri_tests;     /* hitTriangle calls            (PT_FLAG_DETAILED_STATS only) */
    uint64_t tri_accept;    /* accepted triangle hits       (PT_FLAG_DETAILED_STATS only) */
    uint64_t sphere_tests;  /* hitSphere calls              (PT_FLAG_DETAILED_STATS only) */
    uint64_t sky_lookups;   /* getBackgroundColor calls     (PT_FLAG_DETAILED_STATS only) */
    uint64_t bounces;       /* material switch executions   (PT_FLAG_DETAILED_STATS only) */
    uint64_t sphere_accept; /* sphere hits that replaced the record (PT_FLAG_DETAILED_STATS only) */
    uint64_t frames;        /* dispatch-equivalents accumulated since the last pt_reset */
    uint64_t bvh_depth;     /* deepest leaf of the uploaded BVH = traversal stack bound */
    uint64_t walks;         /* rays that entered the BVH below the root: they survived the sphere loop and the test of the
                               root's two child boxes (PT_FLAG_DETAILED_STATS only) */
    uint64_t sky_loop_beside; /* 1: this GPU's stream pair was found to run side by side when it was created, so planned
                               launches run the loop over sky-only tiles beside the megakernel; 0: behind it, on the
                               context's stream (same results, a few per cent slower) */
    uint64_t guarded;       /* rays that set out on the library's own tree instead of the caller's (see PT_FLAG_REFERENCE_WALK) */
    uint64_t rewalks;       /* of their hits, those that failed the check and walked the caller's tree as well */
    uint64_t guard_nodes;   /* nodes of that tree (0: none was built for this mesh) */
} pt_stats_t;

/* ---- lifetime: Renderer::new / drop (src/renderer.rs:32-37) ------------------------ */
/* Creates a renderer for a width x height image on HIP device `device`.  The reference
 * hard-codes 1920x1080 output textures (renderer.rs:157-176); here it is a parameter. */
int pt_create(int device, uint32_t width, uint32_t height, pt_ctx **out);
void pt_destroy(pt_ctx *ctx);
const char *pt_last_error(const pt_ctx *ctx);

/* Launch stream (hipStream_t); NULL = the default stream.  No counterpart: wgpu owns its queue. */
int pt_set_stream(pt_ctx *ctx, void *hip_stream);
int pt_set_flags(pt_ctx *ctx, uint32_t flags);

/* ---- static scene: the create_buffer_init uploads of Renderer::new (renderer.rs:220-241)
 * tris: n_tris x 112 B, nodes: n_nodes x 32 B, tri_idx: n_idx x u32 (bindings 3, 5, 4 of
 * compute.wgsl:75-77).  Copied; the caller's memory is not referenced afterwards.  The BVH is
 * validated (child/leaf ranges in bounds, children stored after their parent) so that a
 * malformed tree is rejected instead of hanging the GPU; so is a tree deeper than 4096 levels
 * (PT_ERR_LIMIT: traversal stacks are sized by the depth), and a mesh whose device records (32 B
 * per node + 64 B per index) would exceed 4 GiB, about 33 million triangles (PT_ERR_LIMIT: a
 * traversal step addresses them with a 32-bit byte offset).  n_nodes == 0 clears the mesh. */
int pt_set_mesh(pt_ctx *ctx, const void *tris, uint32_t n_tris, const void *nodes, uint32_t n_nodes,
                const uint32_t *tri_idx, uint32_t n_idx);

/* Sky cube map, binding 7 (compute.wgsl:79): 6 faces (+X,-X,+Y,-Y,+Z,-Z) of size x size RGBA32F
 * texels, row-major, as produced by CubeTexture::from_equirectangular_hdri (texture.rs:172-270). */
int pt_set_sky_cube(pt_ctx *ctx, const float *rgba32f, uint32_t size);
/* Same, but runs the equirectangular->cube kernel (shaders/equirectangular.wgsl:12-81) on the GPU
 * from a decoded w x h RGBA32F panorama (texture.rs:179-188). */
int pt_set_sky_equirect(pt_ctx *ctx, const float *rgba32f, uint32_t w, uint32_t h, uint32_t cube_size);
/* Copies the current cube map back to the host (6*size*size*4 floats; fewer than that in capacity_floats is
 * PT_ERR_INVALID).  *size (may be NULL) receives the face size, 0 without a sky; with out == NULL only that. */
int pt_read_sky_cube(pt_ctx *ctx, float *out, size_t capacity_floats, uint32_t *size);

/* ---- per-frame uniforms: Renderer::update_buffers (renderer.rs:518-548) ------------ */
int pt_set_camera(pt_ctx *ctx, const void *camera64);
int pt_set_spheres(pt_ctx *ctx, const void *sphere_data8208);
int pt_set_settings(pt_ctx *ctx, const void *settings16);

/* ---- the hot path: Renderer::render's compute pass (renderer.rs:562-570) ----------- */
/* One dispatch of compute.wgsl::main with the `time` uniform (binding 6) given explicitly
 * (the reference derives it from the wall clock, renderer.rs:518-522).  The per-pixel mean
 * over settings.spp samples is ADDED to the fp32 accumulation buffer and the frame count
 * goes up by one. */
int pt_render(pt_ctx *ctx, uint32_t time);
/* n_frames dispatch-equivalents in one launch: frame i uses time0 + i*stride; lanes start their
 * next frame as soon as their path ends.  Launches of fewer than 16 frames are bit-identical to
 * n_frames pt_render calls (every frame is summed on its own and the frames are added in order).
 * Launches of 16 frames or more are split into work items of several frames by measured cost
 * and summed per item: same samples, sums equal up to fp32 summation order, and several times
 * the throughput (1080p bench scene: 0.50 ms for one frame, 0.16 ms per frame in a 15-frame
 * launch, 0.09 ms in a 16-frame one, 0.05 ms from 256 frames on). */
int pt_render_frames(pt_ctx *ctx, uint32_t time0, uint32_t stride, uint32_t n_frames);
/* Device memory pt_render_frames allocates on demand and keeps until pt_destroy (it never shrinks): the work-item
 * table and partial sums of the largest launch so far, 1040 bytes per planned work item -- launches of 16 frames
 * or more plan tiles + 5120*k + 1024 items, k = frames/4 clamped to 6..64 for the pooled kernel and 2 for the
 * per-tile one (1080p: 45 MB at 16 frames, 375 MB at 256); launches of 2..15 frames plan tiles*(frames+1) + 11264
 * items as long as that stays below 640 MiB (1080p, 15 frames: 551 MB) and otherwise run frame after frame with no
 * such memory -- and, for BVHs deeper than the kernels' on-chip stacks (24 levels per-tile, 8 pooled), 4 bytes per
 * lane and extra level.  The first call that needs more than the context holds synchronises the stream, frees and
 * allocates. */
/* No counterpart in the reference.  For callers that cannot allocate device memory while rendering (serving loops,
 * harnesses that forbid hipMalloc): the same launches as pt_render_frames, bit for bit, but the work-item table, the
 * partial sums and the deep-stack spill space come from a static 192 MiB device array that the library holds once per
 * GPU for the life of the process, and no hipMalloc / hipFree is called during the call.  (Not exercised under HIP graph
 * capture: what a call launches depends on host-side state -- the plan's phase, the frame counter, the time values --
 * which a captured graph would freeze.)
 * A launch that needs more (see above; 1080p: up to about 100 frames per call) is rejected with PT_ERR_LIMIT and
 * nothing is rendered: ask for fewer frames per call.  Not available while a window is set.  All contexts of a GPU
 * share the array: calls that overlap in time on one GPU (different streams or threads) race; calls that follow one
 * another on a stream, or are separated by pt_sync, do not.  A plan is re-used only by the context that wrote it. */
int pt_render_frames_static(pt_ctx *ctx, uint32_t time0, uint32_t stride, uint32_t n_frames);
/* ProgressiveRendering::reset_ready_samples (renderer.rs:635-637): clears the accumulation. */
int pt_reset(pt_ctx *ctx);
int pt_sync(pt_ctx *ctx);

/* ---- results ----------------------------------------------------------------------- */
/* Raw accumulation buffer: width*height RGBA32F sums; A holds the frame count. */
int pt_read_sum_rgba32f(pt_ctx *ctx, float *out);
/* sum / frames: the image copy.wgsl:27-37 converges to, without its 8-bit quantisation. */
int pt_read_mean_rgba32f(pt_ctx *ctx, float *out);
/* Mean clamped to [0,1] and quantised like an rgba8unorm store (compute.wgsl:72,127). */
int pt_read_rgba8(pt_ctx *ctx, uint8_t *out);
/* ProgressiveRendering (renderer.rs:613-646) + copy.wgsl:27-37, for callers that want the picture the
 * reference puts on screen: with sample_size in 1..256 (MAX_NUMBER_OF_SAMPLES, renderer.rs:11) every frame
 * is also kept as an rgba8unorm texel image (the shader's store, compute.wgsl:124-127) in a ring of
 * sample_size frames, rendering proceeds one launch per frame, and ready_samples is tracked as
 * Renderer::render does (starts at 0; pt_reset sets it to 1 like reset_ready_samples).  0 turns it off
 * (default).  The fp32 sums are kept as without a window.  Not combinable with pt_bind_accum. */
int pt_set_window(pt_ctx *ctx, uint32_t sample_size);
/* What the last rendered frame's copy pass shows: the mean of the newest
 * n = min(sample_size, ready_samples before that frame) texel images, summed newest first in f32 and
 * divided by n (copy.wgsl:27-37); rgba8_out quantises that once more like a linear 8-bit surface.
 * Either output may be NULL; *frames_shown (may be NULL) receives n.  n == 0 (the very first frame
 * of a context, where the reference divides 0 by 0) is PT_ERR_STATE when an image is asked for. */
int pt_read_window(pt_ctx *ctx, float *rgba32f_out, uint8_t *rgba8_out, uint32_t *frames_shown);
/* Primary-hit primitive id and t per pixel (needs PT_FLAG_AOV during rendering). */
int pt_read_aov(pt_ctx *ctx, uint32_t *prim, float *t);
/* Per 8x8 tile (row-major, ceil(width/8) x ceil(height/8)): 1 if the most recently planned launch found that every
 * camera ray of the tile leaves the scene at once and summed its frames in a loop of its own (identical results,
 * frame after frame), else 0.  All zero while no launch has been planned (one-frame calls, PT_FLAG_DETAILED_STATS,
 * depth 0).  No counterpart in the reference: a diagnostic of this backend's scheduling.
 * *n_tiles (may be NULL) receives the tile count; with out == NULL only that; capacity below it is PT_ERR_INVALID. */
int pt_read_sky_tiles(pt_ctx *ctx, uint32_t *out, size_t capacity, uint32_t *n_tiles);
int pt_stats(pt_ctx *ctx, pt_stats_t *out);
/* Device address of the accumulation buffer (for collectives on it), and a way to make the
 * renderer accumulate into caller-owned device memory of width*height*16 bytes instead. */
void *pt_accum_device_ptr(pt_ctx *ctx);
int pt_bind_accum(pt_ctx *ctx, void *device_ptr);

/* Closest hit of n explicit rays through the same hitScene code (compute.wgsl:214-302).
 * origins/dirs: n x 3 floats.  Outputs (each may be NULL): prim n x u32, t n x f32,
 * p n x 3, normal n x 3, front n x u32. */
int pt_trace_rays(pt_ctx *ctx, const float *origins, const float *dirs, uint32_t n, uint32_t *prim,
                  float *t, float *p, float *normal, uint32_t *front);
/* The same, and per ray which tree answered (walked, n x u32, may be NULL): 0 the caller's tree alone, 1 the library's
 * own tree (see PT_FLAG_REFERENCE_WALK) with its hit taken, 2 that tree first and, its hit failing the check, the
 * caller's tree after it.  The other outputs are identical in all three cases; this one is a diagnostic and the handle by
 * which tests aim rays at the scheme's corner cases (edges, vertices, grazing angles).  No counterpart in the reference. */
int pt_trace_rays_walked(pt_ctx *ctx, const float *origins, const float *dirs, uint32_t n, uint32_t *prim,
                         float *t, float *p, float *normal, uint32_t *front, uint32_t *walked);

/* ---- several GPUs of one node behind one handle ------------------------------------- */
/* No counterpart in the reference, which drives one GPU.  Frames are independent given their `time` value
 * (compute.wgsl:85 seeds the RNG with (x, y, x + time, y + time)), so samples split across GPUs by frame: member g of
 * G renders time0 + g, time0 + g + G, ... into its own fp32 sums, the scene replicated on every GPU, nothing exchanged
 * while rendering; pt_group_reduce then adds every member's sums onto one member.  For a host that drives all GPUs
 * from one process and one thread; a host with one process per GPU uses pt_accum_device_ptr / pt_bind_accum and its
 * own collective instead.  Every member is an ordinary context on a stream of its own; all calls below are
 * asynchronous like their one-GPU counterparts (pt_group_sync waits for every member). */
typedef struct pt_group pt_group;
/* pt_group_create flag: sum with one RCCL ncclReduce(sum, float) per member inside ncclGroupStart/End (librccl is
 * loaded with dlopen at that moment; PT_ERR_STATE if it cannot be) instead of the default: peer copies of the members'
 * buffers onto the root's GPU, all at once, added there in member order -- a fixed summation order, so the result is
 * bitwise the same from run to run.  RCCL needs the devices to be distinct. */
#define PT_GROUP_RCCL 1u
/* devices: n HIP device ordinals, one member each (the same ordinal may appear more than once without PT_GROUP_RCCL:
 * its members then share that GPU). */
int pt_group_create(const int *devices, int n, uint32_t width, uint32_t height, uint32_t flags, pt_group **out);
void pt_group_destroy(pt_group *g);
/* Message of the last failed call on this group (with NULL: of the last failed pt_group_create on this thread). */
const char *pt_group_last_error(const pt_group *g);
int pt_group_size(const pt_group *g);
/* Member i's context, for the per-context calls (pt_read_*, pt_stats, pt_set_window ...); NULL if out of range.
 * Owned by the group: not to be destroyed, and its stream not to be changed. */
pt_ctx *pt_group_member(pt_group *g, int i);
/* The per-context setters, applied to every member (the reference uploads these to its one GPU:
 * renderer.rs:220-241, 518-548). */
int pt_group_set_flags(pt_group *g, uint32_t flags);
int pt_group_set_mesh(pt_group *g, const void *tris, uint32_t n_tris, const void *nodes, uint32_t n_nodes,
                      const uint32_t *tri_idx, uint32_t n_idx);
int pt_group_set_sky_cube(pt_group *g, const float *rgba32f, uint32_t size);
int pt_group_set_sky_equirect(pt_group *g, const float *rgba32f, uint32_t w, uint32_t h, uint32_t cube_size);
int pt_group_set_camera(pt_group *g, const void *camera64);
int pt_group_set_spheres(pt_group *g, const void *sphere_data8208);
int pt_group_set_settings(pt_group *g, const void *settings16);
/* Frames time0 .. time0 + n_frames - 1 dealt over the members: member g renders pt_render_frames(time0 + g, G, its
 * share) on its own GPU; the launches are queued on every GPU before any of them is waited for. */
int pt_group_render_frames(pt_group *g, uint32_t time0, uint32_t n_frames);
/* Adds every other member's sums (colour sums and frame counts) onto member `root`'s and clears the others', so that
 * nothing is counted twice when the group goes on rendering and reduces again.  Afterwards pt_read_mean_rgba32f of
 * pt_group_member(g, root) is the mean over all frames of all members since the last pt_group_reset; it differs from a
 * one-GPU rendering of the same frames by fp32 summation order only. */
int pt_group_reduce(pt_group *g, int root);
int pt_group_reset(pt_group *g);
int pt_group_sync(pt_group *g);
/* Counters summed over the members (bvh_depth: the members' common value; sky_loop_beside: 1 if it is 1 everywhere). */
int pt_group_stats(pt_group *g, pt_stats_t *out);

/* ---- host-side scene construction (no GPU needed) ---------------------------------- */
/* Bvh::from_triangles (src/scene/bvh.rs:34-66): same nodes and same index permutation.
 * nodes_out holds 2*n_tris-1 records of 32 B, idx_out n_tris u32.  Returns nodes used,
 * 0 for an empty mesh, negative pt_status on bad arguments. */
int64_t pt_bvh_build(const void *tris, uint32_t n_tris, void *nodes_out, uint32_t *idx_out);
/* No counterpart in the reference: another tree over the same triangles in the same node format -- binned surface-area
 * heuristic instead of the reference's midpoint split (bvh.rs:101-124), leaves of up to four triangles.  Same buffers and
 * return value as pt_bvh_build.  A traversal of it visits about two thirds of the nodes and triangles (69 k-triangle
 * scene: 22.5 instead of 34.3 steps per ray that enters the tree) and returns the same closest hit except where the
 * shader's arithmetic is itself ambiguous: two triangles at exactly the same t (the first one tested wins,
 * compute.wgsl:286-290, and the order of tests is the tree's) and rays grazing a node's box, where the float slab test
 * (compute.wgsl:441-456) and the triangle test disagree.  What PT_FLAG_REBUILD_BVH uses. */
int64_t pt_bvh_build_sah(const void *tris, uint32_t n_tris, void *nodes_out, uint32_t *idx_out);
/* The same builder over triangle bounds grown by pad[t] >= 0 on every side (n_tris floats): every box of the result
 * contains its triangles with at least that much room.  What pt_set_mesh builds its own tree with (see
 * PT_FLAG_REFERENCE_WALK); exported for tests.  No counterpart in the reference. */
int64_t pt_bvh_build_sah_padded(const void *tris, uint32_t n_tris, const float *pad, void *nodes_out, uint32_t *idx_out);
/* Plane::triangles (src/scene/plane.rs:16-41): writes two 112-byte records. */
int pt_plane_triangles(const float q[3], const float u[3], const float v[3], const float albedo[3],
                       uint32_t material, void *tris_out);
/* SphereDataBuffer::from(&Vec<Sphere>) (src/scene/sphere.rs:113-130): n spheres given as
 * n x {center[3], radius, albedo[3], material as f32}; clamps to 256. */
int pt_pack_spheres(const float *spheres8, uint32_t n, void *sphere_data8208_out);
/* Model::from_obj (src/model.rs:111-213) for the v / vn / f subset: first call with
 * tris_out == NULL to get the triangle count in *n_tris, then with a buffer of that many
 * 112-byte records.  Faces are fan-triangulated; vertices without `vn` get area-weighted
 * smooth normals (the reference would panic on a file without normals). */
int pt_obj_load(const char *path, void *tris_out, uint32_t capacity, uint32_t *n_tris);
/* Radiance .hdr bytes -> w x h RGBA32F panorama, as CubeTexture::from_equirectangular_hdri
 * decodes its `data` argument (src/texture.rs:179-188: HdrDecoder + Rgbe8Pixel::to_hdr, alpha 1);
 * the result is what pt_set_sky_equirect takes.  First call with rgba_out == NULL for the
 * size, then with room for capacity_pixels >= w*h pixels. */
int pt_hdr_decode(const void *data, size_t n_bytes, float *rgba_out, size_t capacity_pixels,
                  uint32_t *width, uint32_t *height);
/* Message of the last failed host-side helper on this thread. */
const char *pt_host_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PTRT_H */
