// pt_kernels.h -- launch interface between the C ABI (pt_api.hip) and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_layout.h"
#include "pt_math.h"

#define PT_BLOCK 256        // threads per workgroup of the small kernels
#define PT_TRACE_BLOCK 64   // megakernel: one wave per workgroup ...
#define PT_TILE 8u          // ... which owns an 8x8 pixel tile
#ifndef PT_TRACE_WAVES_PER_EU
#define PT_TRACE_WAVES_PER_EU 5   // register budget of the megakernel: 512 / 5 -> at most 96 VGPRs
#endif
#define PT_LDS_STACK 24     // traversal stack entries per lane kept in LDS (6 KiB per wave)
// Every wave ends with one atomic add per counter.  A single-frame 1080p launch is 32 400 waves: on one address
// those atomics alone took 0.39 ms (they serialise in L2), so the counters are striped over 256 cache lines and
// summed by the host.
#define PT_COUNTER_STRIPES 256u
#define PT_COUNTER_PITCH 16u  // 64-bit words per stripe (one 128-byte line), >= PT_NUM_COUNTERS
// The pooled kernel takes launches of at least PT_POOL_MIN_FRAMES frames on meshes of at least PT_POOL_MIN_TRIANGLES
// (pooled / per-tile, 1080p, depth 8, 192 to 69 k triangles, one sample per frame: 16 frames 1.01-1.21, 24 frames
// 1.01-1.38, 32 frames 1.14-1.52, 48 frames 1.31-1.53, 96 frames 1.48-1.68; 2 to 12 frames 0.35-0.57; no triangles at
// all 0.95, ten triangles at 16 samples per frame 0.83; 2 to 16 samples per frame, 768 to 69 k triangles: 16 frames and
// more 1.32-1.57, 4 and 8 frames 0.57-0.92), and single frames of at least PT_POOL_ONE_FRAME_MIN_SAMPLES samples on
// meshes of at least PT_POOL_ONE_FRAME_MIN_TRIANGLES (one frame of 64 or 256 samples: 1.08-1.17 from 8.7 k triangles
// on, 0.97-1.02 below).
#define PT_POOL_MIN_TRIANGLES 128u
#define PT_POOL_MIN_FRAMES 16u
// Launches of 2 to 15 frames at one sample per frame (one work item per tile and frame, sums exactly those of separate
// dispatches), pooled / per-tile time, 1080p, depth 8: 192 triangles 1.25 / 1.14 / 1.07 / 0.97 / 0.97 at 2 / 4 / 8 / 12 / 15
// frames; 768: 1.19 / 1.05 / 0.96 / 0.86 / 0.85; 3 k: 1.12 / 1.02 / 0.93 / 0.85 / 0.82; 12 k: 1.08 / 0.96 / 0.89 / 0.82 / 0.79;
// 69 k: 1.02 / 0.90 / 0.81 / 0.72 / 0.71.
#define PT_POOL_SHORT_MIN_FRAMES 8u
#define PT_POOL_SHORT_MIN_TRIANGLES 512u
#define PT_POOL_SHORT_BIG_MIN_FRAMES 4u
#define PT_POOL_SHORT_BIG_MIN_TRIANGLES 8192u
#define PT_POOL_HUGE_MESH_TRIANGLES (1u << 20)  // from here on the planner aims for twice as many work items
#define PT_POOL_ONE_FRAME_MIN_TRIANGLES 8192u
#define PT_POOL_ONE_FRAME_MIN_SAMPLES 48u
#define PT_TILE_EARLY_SHADE_TRIANGLES 32768u  // per-tile kernel: meshes from this size on shade with fewer lanes waiting
// Launches of 2..15 frames plan one work item (a 1 KiB row of partial sums) per tile and frame so that the frames
// run side by side; above this many bytes of such rows they run frame after frame instead (one item per tile, no
// rows), which costs time but no memory.
#define PT_SHORT_LAUNCH_SCRATCH_LIMIT (640ull << 20)
#define PT_PLAN_REUSE_PERIOD 8u  // pooled kernel, identical launches: of 8, six run the old plan, one also measures, one plans anew
// pooled kernel: scheduling thresholds (see PtTraceParams for what each one means)
#define PT_POOL_START 0u
#define PT_POOL_TOPUP 16u
#define PT_POOL_DENSE 16u
#define PT_POOL_SHADE_DENSE 32u
#define PT_POOL_REFILL_LANES 32u
#define PT_POOL_STEPS 12u  // (bench scene 49.3 / 49.5 / 49.1 / 47.2 Grays/s at 8 / 12 / 16 / 24 in the tuning build; 10 M-triangle field 5361 / 5490 / 5554 / 5538 Mrays/s)
#define PT_POOL_MISS_ITERS 32u
#define PT_POOL_MISS_LANES 48u
#define PT_POOL_MAX_IDLE (1u << 22)
#define PT_NUM_COUNTERS 12  // rays, paths, interior, tri_tests, tri_accept, sphere_tests, sky, bounces, sphere_accept, walks, guarded, rewalks

struct PtDeviceScene {
    // spheres: 3 float4 each {center, radius} {albedo, material} {1/radius, 0, 0, 0}
    const float4 *spheres;
    uint32_t n_spheres;
    // The walk table: one array of 64-byte records that a traversal step reads, addressed in 32-byte units from
    // `nodes`.  Position l (odd) = the sibling pair (l, l+1), 4 float4 interleaved {lmin.x rmin.x lmin.y rmin.y}
    // {lmin.z rmin.z lmax.x rmax.x} {lmax.y rmax.y lmax.z rmax.z} {l token, r token, 0, 0}; position tri_base + 2 k =
    // the triangle in leaf-order slot k, {a, original index} {b-a, material} {c-a, 0} {unused}.  A leaf token's low
    // bits are the position of its first triangle.  The whole table is under 4 GiB, so a position << 5 is a 32-bit
    // byte offset.
    const float4 *nodes;
    uint32_t tri_base;
    const uint2 *node_meta;  // (first, count) per node, read for big leaves only
    uint32_t n_nodes;
    uint32_t root_token;
    float4 root_pair[4];     // the root's two children (device node format), valid if the root is interior
    // vertex normals in leaf order, 3 float4 per slot {na, albedo.r} {nb, albedo.g} {nc, albedo.b}: read once per closest hit
    const float4 *tri_nrm;
    // sky cube: 6 faces of sky_size^2 RGBA32F
    const float4 *sky;
    uint32_t sky_size;
    uint32_t tri_materials;  // 1: triangles shade with their records' albedo and material (opt-in extension)
    float t_min, t_max;
    // The guard tree (guard_on == 0: none).  A second tree over the same triangles, built by the library with a
    // surface-area heuristic over triangle bounds padded by more than the triangle test's own rounding error can move a
    // hit point (for rays within `guard_reach` of the mesh), in the same table behind the reference tree: its pairs at
    // positions guard_root.. and its triangle records at slots n_idx.. in its own leaf order.  A ray within reach that
    // has passed the reference root's boxes walks it instead of the reference tree -- it cannot miss a triangle that the
    // triangle test accepts -- and its closest hit is then checked to be one the reference walk returns too (the hit point
    // strictly inside the triangle's own bounds, no second triangle at exactly the same t); a ray that fails the check,
    // or is out of reach, walks the reference tree.  The pooled kernel walks the pair form, the per-tile kernel and ray
    // queries the four-child form below.
    uint32_t guard_on;
    uint32_t guard_root;     // table position of the pair of the guard root's children (the guard root is always interior)
    // The guard tree once more, four children to a 64-byte record (the kernels that step one ray per lane walk this form:
    // half as many dependent fetches per walk).  Record j at position guard_quads + 2 j, j = 0 the root:
    //   {origin.x, origin.y, origin.z, scale.x} {lo.x, lo.y, lo.z, hi.x} {hi.y, hi.z, token 0, token 1} {token 2, token 3,
    //   scale.y, scale.z} -- lo / hi: one byte per child, bounds = origin + byte * scale (scale a power of two, lo rounded
    //   down, hi up: a box never shrinks); a token is a leaf token of the pair form, the position of another record, or
    //   0xFFFFFFFF (no child).  0: no such records.
    uint32_t guard_quads;
    float guard_center[3];   // centre of the mesh's bounds
    float guard_radius;      // no vertex is further from the centre than this (longest edge included)
    float guard_reach;       // (|origin - centre| + radius) * max(1, |direction|) at most this: the ray may walk the guard tree
    float guard_margin;      // a verified hit point lies this far inside the triangle's bounds on every axis
};

struct PtTraceParams {
    PtDeviceScene sc;
    pt_frame fr;
    uint32_t width, height;  // image size (accumulation buffer pitch = width)
    uint32_t cols, rows;     // pixels actually rendered (smaller under PT_FLAG_REFERENCE_DISPATCH)
    uint32_t spp, depth;
    bool has_gizmo;          // a sphere with material 3 exists (selects the kernel variant)
    float inv_spp;           // 1 / spp, rounded once on the host
    uint32_t time0, stride, n_frames;
    // Screen rectangles (in pixels, two pixels of margin) that contain everything a camera ray can see of the root's two
    // child boxes: {x0, x1 - x0, y0, y1 - y0} per child.  A camera ray of a pixel outside both cannot enter either box, so
    // its root visit is known to end the walk without being computed.  {0, 0xFFFFFFFF, 0, 0xFFFFFFFF}: no statement
    // (box partly behind the camera, no BVH, root is a leaf); {0xFFFFFFFF, 0, 0xFFFFFFFF, 0}: box entirely off screen.
    uint32_t cull[8];
    uint32_t refill_lanes;   // run the shading block once this many lanes of a wave wait for it
    uint32_t tiles_x, tiles_y, n_tiles;  // filled in by pt_launch_trace
    const uint4 *items;      // filled in by pt_launch_trace: {tile, first frame, frame count, 0} per work item
    float4 *partial;         // filled in by pt_launch_trace: 64 partial sums per work item
    uint32_t *cost;          // filled in by pt_launch_trace: per-tile work of this launch (node visits + 8 per shading pass)
    float4 *accum;
    uint32_t *aov_prim;  // may be null
    float *aov_t;
    unsigned long long *counters;  // PT_COUNTER_STRIPES x PT_COUNTER_PITCH: a workgroup adds to stripe blockIdx.x % PT_COUNTER_STRIPES
    uint32_t *gstack;              // global-memory traversal stack (deep trees only)
    uint32_t gstack_cap;           // deepest stack the buffer is sized for
    uint32_t gstack_blocks;        // workgroups (of the per-tile kernel) the buffer has columns for
    // pooled kernel only (filled in by pt_launch_trace)
    uint32_t *next_item;           // work-item counter, zero at launch
    uint32_t n_items;
    uint32_t item_batch;           // work items a workgroup claims per trip to next_item
    uint32_t cost_key;             // tiles_x | mask << 16: pixels with ((x | y) & mask) == 0 report their cost (0: all, 1: a quarter)
    uint32_t pool_start;           // an idle wave starts walking once this many rays of its group are pending
    uint32_t pool_topup;           // a walking wave takes more rays once this many of its walk slots are free
    uint32_t pool_dense;           // walkers that make a wave "walking" in that sense
    uint32_t pool_shade_dense;     // ready paths at which even a walking wave runs a shading pass
    uint32_t pool_steps;           // node visits per loop trip
    uint32_t pool_miss_iters;      // most rounds a pass over misses keeps its slots in registers (1: none)
    uint32_t pool_miss_lanes;      // ... as long as this many lanes still carry a slot
    uint32_t max_idle;             // consecutive idle trips after which a wave gives up
};

// Scheduling state owned by the context; every pointer is device memory.
struct PtSchedule {
    uint32_t *cost;        // n_tiles, zero-initialised
    uint32_t *sorted;      // n_tiles
    uint2 *tile_items;     // n_tiles
    uint32_t *tile_sky;    // n_tiles: 1 for a tile whose camera rays all provably leave the scene (set when a launch is planned)
    bool sky_tiles;        // such tiles get no work items: their frames are summed by a loop of their own, frame after frame
    // That loop waits on sky texels while the megakernel is bound by arithmetic, so it runs beside it: the megakernel
    // on main_stream, the loop on side_stream -- two streams of the context's own, checked at creation to run side by
    // side -- forked from and joined back into the launch stream by these events (null: all on the launch stream).
    hipStream_t main_stream, side_stream;
    hipEvent_t fork_event, start_event, join_event, main_event;
    uint4 *items;          // item_cap
    float4 *partial;       // item_cap * 64
    uint32_t item_cap;
    bool *cost_valid;      // host flag: cost holds a measurement of this image
    uint32_t wanted_items;        // work items a plan aims for (a multiple of the resident-wave capacity)
    uint32_t default_chunk;       // frames per item when there is no measurement to plan from
    uint32_t max_chunk;           // most frames a planned item may hold (0: no limit)
    uint32_t min_chunk;           // fewest frames a planned item may hold
    uint32_t min_planned_frames;  // shorter launches run one item per tile (exact sequential sums)
    uint32_t probe_frames;        // frames rendered directly to obtain the first measurement
    bool use_pool;                // pooled kernel (persistent workgroups, paths and rays pooled in LDS)
    uint32_t *next_item;          // one device word for the pooled kernel's item counter
    uint32_t *pool_gstack;        // pt_pool_gstack_bytes() of spill space, null if the tree is shallow enough
    uint32_t n_cus;               // compute units of the device
    uint32_t pool_groups;         // workgroups to launch; 0 = as many as are resident at once
    bool reuse_plan;              // the item table, tile table and buffers still hold the plan of an identical launch: run it again
    bool measure_cost;            // ... and measure the per-tile cost meanwhile, for a new plan next time
};

// stack_depth: deepest leaf of the BVH (bound on traversal stack entries).
hipError_t pt_launch_trace(PtTraceParams p, uint32_t stack_depth, bool detailed_stats, const PtSchedule &sch, hipStream_t stream);
uint32_t pt_trace_num_tiles(uint32_t width, uint32_t height);
// Item-table entries a launch of n_frames can need.
uint32_t pt_trace_item_cap(uint32_t n_tiles, uint32_t n_frames, const PtSchedule &sch);
// Bytes of global stack the per-tile kernel needs for a grid of n_blocks and this tree, 0 if the LDS stack suffices.
size_t pt_trace_gstack_bytes(uint32_t n_blocks, uint32_t stack_depth);
// Same for the pooled kernel (PtSchedule::pool_gstack).
size_t pt_pool_gstack_bytes(uint32_t n_cus, uint32_t stack_depth);

struct PtRayQuery {
    PtDeviceScene sc;
    const float *origins, *dirs;
    uint32_t n;
    uint32_t *prim;
    float *t, *p, *normal;
    uint32_t *front;
    uint32_t *walked;  // may be null: per ray 0 = reference tree only, 1 = guard tree and its answer taken, 2 = guard tree, then the reference tree
    uint32_t *gstack;
    uint32_t gstack_cap;
};
hipError_t pt_launch_ray_query(const PtRayQuery &q, uint32_t stack_depth, hipStream_t stream);
// Runs a pair of tiny kernels (at most 0.5 ms each) to find out whether kernels of streams a and b run at the same
// time, i.e. whether the two streams use different hardware queues.  dev_words: 4 device words of scratch.
hipError_t pt_streams_run_side_by_side(hipStream_t a, hipStream_t b, uint32_t *dev_words, bool *result);
size_t pt_ray_query_gstack_bytes(uint32_t n, uint32_t stack_depth);

// One finished frame into its slot of the rgba8unorm frame ring and onto the running sums; clears `frame`.
hipError_t pt_launch_window_push(float4 *frame, float4 *accum, uint32_t *slot, uint32_t n_pixels, hipStream_t stream);
// Mean of the newest `count` ring frames (newest at slot `newest`, older ones before it, wrapping), as copy.wgsl sums them.
hipError_t pt_launch_window_mean(const uint32_t *ring, uint32_t n_pixels, uint32_t ring_frames, uint32_t newest, uint32_t count,
                                 float4 *out_f, uint32_t *out_u8, hipStream_t stream);
// mode 0: mean rgba32f -> out_f; mode 1: clamp + rgba8unorm quantise -> out_u8
hipError_t pt_launch_resolve(const float4 *accum, uint32_t n_pixels, float4 *out_f, uint32_t *out_u8, int mode,
                             hipStream_t stream);
hipError_t pt_launch_equirect(const float4 *src, uint32_t src_w, uint32_t src_h, float4 *dst, uint32_t size,
                              hipStream_t stream);
