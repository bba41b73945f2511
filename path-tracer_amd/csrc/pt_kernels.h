// pt_kernels.h -- launch interface between the C ABI (pt_api.hip) and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_layout.h"
#include "pt_math.h"

#define PT_BLOCK 256        // 4 waves; each wave owns an 8x8 pixel tile of a 16x16 block tile
#define PT_LDS_STACK 24     // traversal stack entries per lane kept in LDS (24 KiB per block)
#define PT_NUM_COUNTERS 9   // rays, paths, interior, tri_tests, tri_accept, sphere_tests, sky, bounces, sphere_accept

struct PtDeviceScene {
    // spheres: 2 float4 each {center, radius} {albedo, material}
    const float4 *spheres;
    uint32_t n_spheres;
    // BVH: 2 float4 per node {min, token} {max, count}, indexed by the reference's node index
    const float4 *nodes;
    const uint2 *node_meta;  // (first, count) per node, read for big leaves only
    uint32_t n_nodes;
    uint32_t root_token;
    // triangles in leaf order: 3 float4 each {a, original index} {b-a, 0} {c-a, 0} / {na} {nb} {nc}
    const float4 *tri_pos;
    const float4 *tri_nrm;
    // sky cube: 6 faces of sky_size^2 RGBA32F
    const float4 *sky;
    uint32_t sky_size;
    float t_min, t_max;
};

struct PtTraceParams {
    PtDeviceScene sc;
    pt_frame fr;
    uint32_t width, height;  // image size (accumulation buffer pitch = width)
    uint32_t cols, rows;     // pixels actually rendered (smaller under PT_FLAG_REFERENCE_DISPATCH)
    uint32_t spp, depth;
    uint32_t time0, stride, n_frames;
    uint32_t chunk_frames;   // frames per work item; items of one tile write separate partial sums
    uint32_t n_chunks;       // filled in by pt_launch_trace
    uint32_t tiles_x, tiles_y, n_tiles;  // filled in by pt_launch_trace
    uint32_t chunk_base, launch_chunks;  // filled in by pt_launch_trace (chunks covered by one launch)
    const uint32_t *order;   // optional: n_tiles entries, rewritten after each launch (costliest tile first)
    uint32_t *cost;          // optional: n_tiles wave-time sums, zero before the first launch
    float4 *partial;         // n_chunks * width * height partial sums (needed when n_chunks > 1)
    float4 *accum;
    uint32_t *aov_prim;  // may be null
    float *aov_t;
    unsigned long long *counters;  // PT_NUM_COUNTERS
    uint32_t *gstack;              // global-memory traversal stack (deep trees only)
    uint32_t gstack_cap;
};

// stack_depth: deepest leaf of the BVH (bound on traversal stack entries).
// order_valid: in/out, whether p.order holds an order from an earlier launch of the same image size.
hipError_t pt_launch_trace(PtTraceParams p, uint32_t stack_depth, bool detailed_stats, bool *order_valid, hipStream_t stream);
// Work items per tile for a launch of n_frames (1 = frames are added straight onto the running sum).
uint32_t pt_trace_num_chunks(uint32_t n_frames, uint32_t chunk_frames, uint32_t stack_depth);
// Bytes of global stack pt_launch_trace needs for this image/tree, 0 if the LDS stack suffices.
size_t pt_trace_gstack_bytes(uint32_t width, uint32_t height, uint32_t stack_depth);

struct PtRayQuery {
    PtDeviceScene sc;
    const float *origins, *dirs;
    uint32_t n;
    uint32_t *prim;
    float *t, *p, *normal;
    uint32_t *front;
    uint32_t *gstack;
    uint32_t gstack_cap;
};
hipError_t pt_launch_ray_query(const PtRayQuery &q, uint32_t stack_depth, hipStream_t stream);
size_t pt_ray_query_gstack_bytes(uint32_t n, uint32_t stack_depth);

// mode 0: mean rgba32f -> out_f; mode 1: clamp + rgba8unorm quantise -> out_u8
hipError_t pt_launch_resolve(const float4 *accum, uint32_t n_pixels, float4 *out_f, uint32_t *out_u8, int mode,
                             hipStream_t stream);
hipError_t pt_launch_equirect(const float4 *src, uint32_t src_w, uint32_t src_h, float4 *dst, uint32_t size,
                              hipStream_t stream);
