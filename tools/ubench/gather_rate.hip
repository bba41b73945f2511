// How fast does a CU serve a chain of dependent, per-lane scattered reads of 64-byte records (the shape of a BVH
// walk: every lane follows its own chain, the next record's index comes out of the one just read)?  Variants:
//   own4   each lane reads its record with four 16-byte loads (what the walk does)
//   own2   two 16-byte loads (32 bytes of the record)
//   own1   one 16-byte load
//   quad   four lanes share the fetch of one record: four wave-instructions, each reading 16 records as 16 B per
//          lane (lanes 4k..4k+3 the four quarters of one record), staged through LDS so that every lane ends up
//          with its own record's 64 bytes
// Prints wave-cycles per chain step and records per microsecond chip-wide, for several table sizes, at 4 waves per
// SIMD (1024 workgroups of 256 threads) with all lanes and with 44 lanes of each wave active.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_rate.hip -o tools/ubench/gather_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t next_index(uint32_t idx, float4 a, uint32_t mask)
{
    return (idx * 2654435761u + __float_as_uint(a.x)) & mask;
}

template <int LOADS, int STRIDE = 4, bool MATE = false>
__global__ __launch_bounds__(256, 4) void own(const float4 *__restrict__ tab, uint32_t mask, uint32_t steps, uint32_t active,
                                             uint32_t *out, unsigned long long *cyc)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 747796405u & mask;
    float acc = 0.0f;
    const unsigned long long t0 = clock64();
    if (lane < active) {
        for (uint32_t s = 0; s < steps; s++) {
            const float4 *r = tab + (size_t)idx * STRIDE;
            float4 a = r[0], b = a, c = a, d = a;
            if (LOADS >= 2) b = r[1];
            if (LOADS >= 4) { c = r[2]; d = r[3]; }
            if (LOADS >= 8) { const float4 e = r[4], f = r[5], g = r[6], h = r[7]; acc += e.x + f.y + g.z + h.w; }
            acc += b.y + c.z + d.w;
            idx = (MATE && (s & 1u)) ? (idx ^ 1u) : next_index(idx, a, mask);
        }
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 256u + threadIdx.x] = idx + __float_as_uint(acc);
    if (lane == 0) cyc[blockIdx.x * 4u + (threadIdx.x >> 6)] = t1 - t0;
}

// the records in LDS (a 16 KiB table: 256 records), four ds_read_b128 per step
__global__ __launch_bounds__(256, 4) void lds4(const float4 *__restrict__ tab, uint32_t mask, uint32_t steps, uint32_t active,
                                              uint32_t *out, unsigned long long *cyc)
{
    __shared__ float4 top[256 * 4];
    for (uint32_t i = threadIdx.x; i < 1024u; i += 256u) top[i] = tab[i];
    __syncthreads();
    mask &= 255u;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 747796405u & mask;
    float acc = 0.0f;
    const unsigned long long t0 = clock64();
    if (lane < active) {
        for (uint32_t s = 0; s < steps; s++) {
            const float4 *r = top + (size_t)idx * 4u;
            const float4 a = r[0], b = r[1], c = r[2], d = r[3];
            acc += b.y + c.z + d.w;
            idx = next_index(idx, a, mask);
        }
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 256u + threadIdx.x] = idx + __float_as_uint(acc);
    if (lane == 0) cyc[blockIdx.x * 4u + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ __launch_bounds__(256, 4) void quad(const float4 *__restrict__ tab, uint32_t mask, uint32_t steps, uint32_t active,
                                              uint32_t *out, unsigned long long *cyc)
{
    __shared__ float4 stage[4][64 * 5];  // per wave: 64 records of 64 B, 80 B apart
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 747796405u & mask;
    float acc = 0.0f;
    float4 *const mine = &stage[wv][0];
    const unsigned long long t0 = clock64();
    for (uint32_t s = 0; s < steps; s++) {
        // (all lanes take part in the fetch; lanes past `active` fetch record 0 and ignore it)
        const uint32_t want = lane < active ? idx : 0u;
        float4 part[4];
#pragma unroll
        for (uint32_t i = 0; i < 4u; i++) {
            const uint32_t src = i * 16u + (lane >> 2);
            const uint32_t ridx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src * 4u), (int)want);
            part[i] = tab[(size_t)ridx * 4u + (lane & 3u)];
        }
#pragma unroll
        for (uint32_t i = 0; i < 4u; i++) mine[(i * 16u + (lane >> 2)) * 5u + (lane & 3u)] = part[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float4 a = mine[lane * 5u], b = mine[lane * 5u + 1u], c = mine[lane * 5u + 2u], d = mine[lane * 5u + 3u];
        __builtin_amdgcn_wave_barrier();
        if (lane < active) {
            acc += b.y + c.z + d.w;
            idx = next_index(idx, a, mask);
        }
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 256u + threadIdx.x] = idx + __float_as_uint(acc);
    if (lane == 0) cyc[blockIdx.x * 4u + wv] = t1 - t0;
}

int main(int argc, char **argv)
{
    const uint32_t groups = argc > 1 ? (uint32_t)atoi(argv[1]) : 1024u, steps = 2000;
    printf("%u workgroups of 4 waves (%.1f waves per SIMD)\n", groups, groups / 256.0);
    uint32_t *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&out, groups * 256 * 4));
    CHECK(hipMalloc(&cyc, groups * 4 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<unsigned long long> hc(groups * 4);
    uint32_t want_out[2][4] = {{0}};
    for (uint32_t kb : {16u, 2048u, 16384u}) {
        const uint32_t nrec = kb * 1024u / 64u;
        std::vector<float> h((size_t)nrec * 16);
        uint32_t x = 12345u;
        for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (float)(x >> 8) * (1.0f / 16777216.0f); }
        float4 *tab;
        CHECK(hipMalloc(&tab, (size_t)nrec * 64));
        CHECK(hipMemcpy(tab, h.data(), (size_t)nrec * 64, hipMemcpyHostToDevice));
        for (uint32_t active : {64u, 44u}) {
            for (int variant = 0; variant < 8; variant++) {
                const char *names[8] = {"own4", "own2", "own1", "quad", "lds4", "own4 one per 128 B line", "own8 (128 B records)", "own4, every other step to the line mate"};
                if (variant == 4 && kb != 16u) continue;
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(e0));
                    switch (variant) {
                    case 0: own<4><<<groups, 256>>>(tab, nrec - 1, steps, active, out, cyc); break;
                    case 1: own<2><<<groups, 256>>>(tab, nrec - 1, steps, active, out, cyc); break;
                    case 2: own<1><<<groups, 256>>>(tab, nrec - 1, steps, active, out, cyc); break;
                    case 3: quad<<<groups, 256>>>(tab, nrec - 1, steps, active, out, cyc); break;
                    case 4: lds4<<<groups, 256>>>(tab, nrec - 1, steps, active, out, cyc); break;
                    case 5: own<4, 8><<<groups, 256>>>(tab, nrec / 2 - 1, steps, active, out, cyc); break;
                    case 6: own<8, 8><<<groups, 256>>>(tab, nrec / 2 - 1, steps, active, out, cyc); break;
                    default: own<4, 4, true><<<groups, 256>>>(tab, nrec - 1, steps, active, out, cyc); break;
                    }
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                }
                CHECK(hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost));
                double sum = 0;
                for (auto c : hc) sum += (double)c;
                // every variant follows the same chains: the first outputs agree between own4 and quad
                uint32_t first[4];
                CHECK(hipMemcpy(first, out, sizeof first, hipMemcpyDeviceToHost));
                if (variant == 0) for (int k = 0; k < 4; k++) want_out[active == 64u][k] = first[k];
                const bool same = variant != 3 || (first[0] == want_out[active == 64u][0] && first[3] == want_out[active == 64u][3]);
                printf("table %6u KB  %2u lanes  %s  %7.0f cycles/step  %6.3f ms  %7.1f records/us%s\n", kb, active, names[variant],
                       sum / hc.size() / steps, ms, (double)groups * 4 * active * steps / (ms * 1e3), same ? "" : "  (RESULT DIFFERS)");
            }
        }
        CHECK(hipFree(tab));
    }
    return 0;
}
