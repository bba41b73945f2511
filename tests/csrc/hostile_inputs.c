/* Test driver (host only): the oracle rendering scenes with hostile numbers -- NaN and infinite cameras, zero-area
 * and enormous triangles, zero and negative radii, NaN texels -- under AddressSanitizer and UBSan.  The GPU kernels
 * evaluate the same index arithmetic (sky texel, pixel, BVH entries), so an in-bounds oracle is the host-side
 * evidence that such inputs cannot fault the GPU.  Exit code 0 and "ok" on stdout when every case ran. */
#include "../../oracle/pt_oracle.c"

#include <stdio.h>
#include <stdlib.h>

static void run(const char *name, pto_scene *sc)
{
    float *accum = calloc((size_t)sc->width * sc->height * 4, sizeof(float));
    pto_stats st;
    for (uint32_t t = 0; t < 3; t++)
        if (pto_render(sc, 0xFFFFFFF0u + t * 7u, accum, NULL, NULL, &st, 2) != 0) { printf("%s: rejected\n", name); exit(2); }
    printf("%s: %llu rays\n", name, (unsigned long long)st.rays);
    free(accum);
}

int main(void)
{
    const float nan = NAN, inf = INFINITY;
    enum { W = 37, H = 23, SKY = 8 };
    float *sky = malloc(sizeof(float) * 6 * SKY * SKY * 4);
    for (int i = 0; i < 6 * SKY * SKY * 4; i++) sky[i] = (i % 97 == 0) ? nan : (i % 89 == 0 ? inf : 0.25f + (float)(i % 7));
    pto_triangle tris[8];
    memset(tris, 0, sizeof tris);
    const float q0[3] = {-1, -1, -2}, u0[3] = {2, 0, 0}, v0[3] = {0, 2, 0}, al[3] = {0.5f, 0.5f, 0.5f};
    pto_plane_triangles(q0, u0, v0, al, 0, &tris[0]);
    const float q1[3] = {0, 0, -1}, u1[3] = {0, 0, 0}, v1[3] = {0, 0, 0};             /* zero area */
    pto_plane_triangles(q1, u1, v1, al, 1, &tris[2]);
    const float q2[3] = {-1e30f, -1e30f, -3}, u2[3] = {2e30f, 0, 0}, v2[3] = {0, 2e30f, 0}; /* enormous */
    pto_plane_triangles(q2, u2, v2, al, 2, &tris[4]);
    const float q3[3] = {0.3f, 0.2f, -1.5f}, u3[3] = {1e-30f, 0, 0}, v3_[3] = {0, 1e-30f, 0}; /* subnormal cross products */
    pto_plane_triangles(q3, u3, v3_, al, 0, &tris[6]);
    pto_node nodes[15];
    uint32_t idx[8];
    const int64_t n_nodes = pto_bvh_build(tris, 8, nodes, idx);
    if (n_nodes <= 0) return 3;
    pto_sphere_data sd;
    memset(&sd, 0, sizeof sd);
    sd.count = 4;
    sd.s[0] = (pto_sphere){{0, 0, -1}, 0.5f, {0.8f, 0.3f, 0.3f}, 2.0f};
    sd.s[1] = (pto_sphere){{1, 0, -1}, 0.0f, {1, 1, 1}, 1.0f};       /* zero radius: 1/radius is inf */
    sd.s[2] = (pto_sphere){{-1, 0, -1}, -0.5f, {1, 1, 1}, 3.0f};     /* negative radius, gizmo */
    sd.s[3] = (pto_sphere){{0, nan, -1}, 0.5f, {1, 1, 1}, nan};      /* NaN centre and material */
    pto_settings set = {2, 6, 1e-4f, 1000.0f};
    pto_camera cams[6];
    for (int k = 0; k < 6; k++) {
        pto_camera c = {{0, 0, 1}, 1.0f, {0, 0, -1}, 75.0f, {1, 0, 0}, 0, {0, 1, 0}, 0};
        if (k == 1) c.origin[0] = nan;
        if (k == 2) c.forward[2] = inf;
        if (k == 3) c.vfov = 180.0f;          /* tan(90 deg): a huge viewport */
        if (k == 4) c.focal = 0.0f;
        if (k == 5) { c.right[0] = 1e38f; c.up[1] = -1e38f; }
        cams[k] = c;
    }
    for (int k = 0; k < 6; k++)
        for (int with_mesh = 0; with_mesh < 2; with_mesh++)
            for (int with_sky = 0; with_sky < 2; with_sky++) {
                pto_scene sc;
                memset(&sc, 0, sizeof sc);
                sc.width = W; sc.height = H;
                sc.camera = &cams[k]; sc.settings = &set; sc.spheres = &sd;
                if (with_mesh) { sc.triangles = tris; sc.n_tris = 8; sc.nodes = nodes; sc.n_nodes = (uint32_t)n_nodes; sc.tri_idx = idx; sc.n_idx = 8; }
                if (with_sky) { sc.sky = sky; sc.sky_size = SKY; }
                sc.flags = (k & 1) ? PTO_FLAG_TRIANGLE_MATERIALS : 0;
                char name[64];
                snprintf(name, sizeof name, "camera %d mesh %d sky %d", k, with_mesh, with_sky);
                run(name, &sc);
            }
    free(sky);
    printf("ok\n");
    return 0;
}
