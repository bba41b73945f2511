// Exhaustive check (all 2^32 bit patterns) of short reciprocal sequences against the correctly rounded
// IEEE quotient 1.0f / x that hipcc emits with -fhip-fp32-correctly-rounded-divide-sqrt.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt
//        -fno-gpu-flush-denormals-to-zero tools/ubench/rcp_exact.hip -o /tmp/rcp_exact
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float cand_a(float x)  // rcp + two Newton steps + fixup
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return __builtin_amdgcn_div_fixupf(r, x, 1.0f);
}
__device__ __forceinline__ float cand_b(float x)  // rcp + one Newton step + fixup
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return __builtin_amdgcn_div_fixupf(r, x, 1.0f);
}

__global__ void check(unsigned long long *bad_a, unsigned long long *bad_b, uint32_t *first_a, uint32_t *first_b)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float ref = 1.0f / x;
        const uint32_t rb = __float_as_uint(ref);
        const bool ref_nan = ref != ref;
        const uint32_t ex = (bits >> 23) & 0xFFu;
        const float a = cand_a(x), b = cand_b(x);
        if (!(ref_nan ? a != a : __float_as_uint(a) == rb)) { atomicAdd(&bad_a[ex], 1ull); atomicMin(first_a + ex, bits); }
        if (!(ref_nan ? b != b : __float_as_uint(b) == rb)) { atomicAdd(&bad_b[ex], 1ull); atomicMin(first_b + ex, bits); }
    }
}

int main()
{
    unsigned long long *ba, *bb;
    uint32_t *fa, *fb;
    hipMalloc(&ba, 256 * 8); hipMalloc(&bb, 256 * 8); hipMalloc(&fa, 256 * 4); hipMalloc(&fb, 256 * 4);
    hipMemset(ba, 0, 256 * 8); hipMemset(bb, 0, 256 * 8); hipMemset(fa, 0xFF, 256 * 4); hipMemset(fb, 0xFF, 256 * 4);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, ba, bb, fa, fb);
    unsigned long long ha[256], hb[256];
    uint32_t hfa[256], hfb[256];
    hipMemcpy(ha, ba, sizeof ha, hipMemcpyDeviceToHost); hipMemcpy(hb, bb, sizeof hb, hipMemcpyDeviceToHost);
    hipMemcpy(hfa, fa, sizeof hfa, hipMemcpyDeviceToHost); hipMemcpy(hfb, fb, sizeof hfb, hipMemcpyDeviceToHost);
    unsigned long long ta = 0, tb = 0;
    for (int e = 0; e < 256; e++) {
        ta += ha[e]; tb += hb[e];
        if (ha[e] || hb[e]) printf("exponent %3d: two-step %llu bad (first 0x%08x)  one-step %llu bad (first 0x%08x)\n", e, ha[e], hfa[e], hb[e], hfb[e]);
    }
    printf("total mismatches over 2^32 inputs: two-step %llu, one-step %llu\n", ta, tb);
    return 0;
}
