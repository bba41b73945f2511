// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the two read patterns the renderer has:
// a coalesced float4 stream and scattered single float4 reads from a buffer far larger than the caches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void stream_read(const float4 *src, size_t n, float4 *out)
{
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.0f) out[0] = acc;
}
__global__ __launch_bounds__(256) void scatter_read(const float4 *src, size_t n, size_t reads_per_thread, float4 *out)
{
    unsigned long long s = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t k = 0; k < reads_per_thread; k++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const float4 v = src[(s >> 20) % n];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.0f) out[0] = acc;
}
int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t n = (size_t)2 << 30 >> 4;  // 2 GiB of float4
    float4 *src, *out;
    if (hipMalloc(&src, n * 16) != hipSuccess || hipMalloc(&out, 16) != hipSuccess) return 1;
    if (hipMemset(src, 0, n * 16) != hipSuccess) return 1;
    hipDeviceSynchronize();
    if (mode == 0) {
        stream_read<<<4096, 256>>>(src, n, out);
        printf("stream: %zu bytes read\n", n * 16);
    } else {
        const size_t rpt = 64;
        scatter_read<<<4096, 256>>>(src, n, rpt, out);
        printf("scatter: %zu reads of 16 B = %zu bytes\n", (size_t)4096 * 256 * rpt, (size_t)4096 * 256 * rpt * 16);
    }
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
