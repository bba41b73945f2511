// pt_layout.h -- byte layouts shared with the reference's host code, and the device-side
// scene description the kernels read.
#pragma once

#include <stddef.h>
#include <stdint.h>

// ---- the reference's #[repr(C)] Pod structs, byte for byte --------------------------
struct PtCamera {  // CameraBuffer, src/scene/camera.rs:73-84
    float origin[3];
    float focal_length;
    float forward[3];
    float vfov;  // degrees
    float right[3];
    uint32_t pad1;
    float up[3];
    uint32_t pad2;
};
struct PtSettings {  // Settings, src/renderer.rs:604-611
    uint32_t samples_per_pixel;
    uint32_t depth;
    float t_min;
    float t_max;
};
struct PtSphere {  // SphereBuffer, src/scene/sphere.rs:81-88
    float center[3];
    float radius;
    float albedo[3];
    float material;  // 0 diffuse, 1 metal, 2 dielectric, 3 gizmo -- stored as f32
};
struct PtSphereData {  // SphereDataBuffer, src/scene/sphere.rs:105-111
    uint32_t sphere_count;
    uint32_t pad[3];
    PtSphere spheres[256];
};
struct PtTriangle {  // TriangleBuffer, src/model.rs:51-68
    float a[3], pad0;
    float b[3], pad1;
    float c[3], pad2;
    float na[3], pad3;
    float nb[3], pad4;
    float nc[3], pad5;
    float albedo[3];
    uint32_t material;  // never read by the shader (compute.wgsl:359-367)
};
struct PtNode {  // Node, src/scene/bvh.rs:7-14
    float min_corner[3];
    uint32_t left_child_index;  // interior: first child (second is +1); leaf: first index
    float max_corner[3];
    uint32_t triangle_count;  // 0 = interior
};

static_assert(sizeof(PtCamera) == 64 && offsetof(PtCamera, forward) == 16 && offsetof(PtCamera, vfov) == 28 &&
                  offsetof(PtCamera, right) == 32 && offsetof(PtCamera, up) == 48,
              "CameraBuffer layout");
static_assert(sizeof(PtSettings) == 16 && offsetof(PtSettings, t_min) == 8, "Settings layout");
static_assert(sizeof(PtSphere) == 32 && offsetof(PtSphere, albedo) == 16 && offsetof(PtSphere, material) == 28,
              "SphereBuffer layout");
static_assert(sizeof(PtSphereData) == 8208 && offsetof(PtSphereData, spheres) == 16, "SphereDataBuffer layout");
static_assert(sizeof(PtTriangle) == 112 && offsetof(PtTriangle, b) == 16 && offsetof(PtTriangle, na) == 48 &&
                  offsetof(PtTriangle, albedo) == 96 && offsetof(PtTriangle, material) == 108,
              "TriangleBuffer layout");
static_assert(sizeof(PtNode) == 32 && offsetof(PtNode, left_child_index) == 12 && offsetof(PtNode, max_corner) == 16 &&
                  offsetof(PtNode, triangle_count) == 28,
              "Node layout");

#define PT_MAX_SPHERES 256u
#define PT_MAX_BVH_DEPTH 4096u  // deepest tree pt_set_mesh accepts (traversal stacks are sized by it)

// ---- node reference ("token") used by the traversal stack ---------------------------
// bit 31 clear: interior node; the value is the index of its first child (second = +1).
// bit 31 set : leaf.  bits 30..28 = triangle count 1..7 and bits 27..0 = first slot in the
//              leaf-ordered triangle arrays; count field 0 = big leaf, bits 27..0 = node
//              index whose (first, count) are read from the side table.
#define PT_TOKEN_LEAF 0x80000000u
#define PT_TOKEN_LOW 0x0FFFFFFFu
