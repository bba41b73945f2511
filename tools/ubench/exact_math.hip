// Exhaustive check (all 2^32 bit patterns) of short reciprocal and square-root sequences against the correctly
// rounded results hipcc emits with -fhip-fp32-correctly-rounded-divide-sqrt (subnormals kept).  Prints, per binary
// exponent of the input, how many inputs each candidate gets wrong: the kernels may use a candidate only on the
// exponent range where that count is zero.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt
//        -fno-gpu-flush-denormals-to-zero tools/ubench/exact_math.hip -o tools/ubench/exact_math
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float rcp_nofix(float x)  // v_rcp_f32 + one Newton step
{
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_fix(float x) { return __builtin_amdgcn_div_fixupf(rcp_nofix(x), x, 1.0f); }
__device__ __forceinline__ float sqrt_short(float x)  // v_sqrt_f32, then the neighbour below / above if it is the better one
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
    const float rl = __builtin_fmaf(-lo, s, x), rh = __builtin_fmaf(-hi, s, x);
    float r = rl <= 0.0f ? lo : s;
    r = rh > 0.0f ? hi : r;
    return r;
}

#define NCAND 3
__global__ void check(unsigned long long *bad, uint32_t *first)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const uint32_t ex = (bits >> 23) & 0xFFu;
        const float refs[NCAND] = {1.0f / x, 1.0f / x, __builtin_sqrtf(x)};
        const float got[NCAND] = {rcp_nofix(x), rcp_fix(x), sqrt_short(x)};
        for (int c = 0; c < NCAND; c++) {
            const bool same = refs[c] != refs[c] ? got[c] != got[c] : __float_as_uint(got[c]) == __float_as_uint(refs[c]);
            if (!same) { atomicAdd(&bad[c * 512 + (bits >> 31) * 256 + ex], 1ull); atomicMin(&first[c * 512 + (bits >> 31) * 256 + ex], bits); }
        }
    }
}

int main()
{
    unsigned long long *bad, h[NCAND * 512];
    uint32_t *first, hf[NCAND * 512];
    hipMalloc(&bad, sizeof h); hipMalloc(&first, sizeof hf);
    hipMemset(bad, 0, sizeof h); hipMemset(first, 0xFF, sizeof hf);
    hipLaunchKernelGGL(check, dim3(8192), dim3(256), 0, 0, bad, first);
    if (hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hf, first, sizeof hf, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    const char *names[NCAND] = {"1/x: rcp + Newton step", "1/x: rcp + Newton step + div_fixup", "sqrt: v_sqrt + neighbour test"};
    for (int c = 0; c < NCAND; c++) {
        unsigned long long total = 0;
        printf("%s\n", names[c]);
        for (int s = 0; s < 2; s++)
            for (int e = 0; e < 256; e++)
                if (h[c * 512 + s * 256 + e]) { total += h[c * 512 + s * 256 + e]; printf("  sign %d exponent %3d: %llu wrong (first 0x%08x)\n", s, e, h[c * 512 + s * 256 + e], hf[c * 512 + s * 256 + e]); }
        printf("  total wrong over 2^32 inputs: %llu\n", total);
    }
    return 0;
}
