// Micro-benchmark: VALU issue rate of plain (unpacked) f32 FMA, integer ops and division on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; j++) { u0 = (u0 << 3) ^ u1; u1 = (u1 >> 5) ^ u2; u2 = (u2 << 7) ^ u3; u3 = (u3 >> 2) ^ u0; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) { x0 = a / x0; x1 = a / x1; x2 = b / x2; x3 = b / x3; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(u0 ^ u1 ^ u2 ^ u3);
}
int main()
{
    float *out; hipMalloc(&out, 256 * 64 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++)
        for (int bpc : {1, 2, 4, 8}) {
            int blocks = 256 * bpc, iters = 20000;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
                else if (mode == 1) k<1><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
                else k<2><<<blocks, 256>>>(out, iters / 10, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double per_wave = mode == 0 ? 64.0 * iters : mode == 1 ? 64.0 * iters : 16.0 * (iters / 10);
            double winstr = per_wave * blocks * 4;
            // cycles per wave-instruction per SIMD assuming 2.4 GHz
            double cyc = ms * 1e-3 * 2.4e9 * 1024 / winstr;
            printf("mode %d (%s) waves/SIMD %d: %.3f ms, %.2f SIMD-cycles per wave-op @2.4GHz, %.1f Gop/s lane\n", mode,
                   mode == 0 ? "fma" : mode == 1 ? "shift-xor (2 ops counted 1)" : "ieee div", bpc, ms, cyc, winstr * 64 / ms / 1e6);
        }
    return 0;
}
