// What does a kernel that only streams cost at the sizes of this step's LayerNorm / CE launches?  (the
// ceiling the HBM-bound kernels are priced against is the 8 TB/s of the data sheet; this prints what a plain copy of the same
// bytes reaches at the same launch size, i.e. how much of the gap is launch ramp and tail at 32 - 320 MB per launch.)
//   hipcc --offload-arch=gfx950 -O3 -o stream_ceiling.bin stream_ceiling.hip && ./stream_ceiling.bin
// Kernels: copy (16-B loads/stores, one 2-KiB row per wave per trip), copy2 (two inputs, one output: the LN backward's shape
// without its arithmetic), read_only (rows reduced to one float per row: the CE forward's shape).  Durations by HIP events over
// 50 back-to-back launches (so they include the inter-launch gap a stream of such kernels pays) and, under rocprofv3
// --kernel-trace --stats, per kernel without it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int NIN>
__global__ __launch_bounds__(256) void copy_rows(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o, int64_t rows, int v16_per_row) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        for (int c = lane; c < v16_per_row; c += 64) {
            u32x4 v = a[row * v16_per_row + c];
            if (NIN == 2) { const u32x4 w = b[row * v16_per_row + c]; v.x ^= w.x; v.y ^= w.y; v.z ^= w.z; v.w ^= w.w; }
            o[row * v16_per_row + c] = v;
        }
    }
}
__global__ __launch_bounds__(256) void read_rows(const u32x4* __restrict__ a, float* __restrict__ o, int64_t rows, int v16_per_row) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        unsigned s = 0;
        for (int c = lane; c < v16_per_row; c += 64) { const u32x4 v = a[row * v16_per_row + c]; s += v.x ^ v.y ^ v.z ^ v.w; }
        for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k, 64);
        if (lane == 0) o[row] = (float)s;
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int64_t cap = (int64_t)1 << 30;
    u32x4 *a, *b, *o; float* f;
    CK(hipMalloc(&a, cap)); CK(hipMalloc(&b, cap)); CK(hipMalloc(&o, cap)); CK(hipMalloc(&f, 1 << 24));
    CK(hipMemset(a, 1, cap)); CK(hipMemset(b, 2, cap));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int64_t rows; int cols_bytes; int kind; int grid; };
    const Case cases[] = {
        {"copy   8192 x 2 KiB (LN fwd: 32 MB)", 8192, 2048, 1, 2048}, {"copy   8192 x 2 KiB, 1024 wg", 8192, 2048, 1, 1024}, {"copy   8192 x 2 KiB, 512 wg", 8192, 2048, 1, 512},
        {"copy2  8192 x 2 KiB (LN bwd w/o resid: 48 MB)", 8192, 2048, 2, 2048}, {"copy2  8192 x 2 KiB, 512 wg", 8192, 2048, 2, 512}, {"copy2  8192 x 2 KiB, 256 wg", 8192, 2048, 2, 256},
        {"copy   8192 x 4 KiB (large: 64 MB)", 8192, 4096, 1, 2048},
        {"copy   1229 x 128 KiB (CE: 322 MB)", 1229, 131072, 1, 1229}, {"copy   1229 x 128 KiB, 2048 wg", 1229, 131072, 1, 2048},
        {"read   1229 x 128 KiB (161 MB)", 1229, 131072, 0, 1229},
        {"copy   65536 x 2 KiB (256 MB)", 65536, 2048, 1, 2048}, {"copy   262144 x 2 KiB (1 GB)", 262144, 2048, 1, 2048},
    };
    for (const Case& c : cases) {
        const int v16 = c.cols_bytes / 16;
        auto launch = [&]() {
            if (c.kind == 1) hipLaunchKernelGGL(copy_rows<1>, dim3(c.grid), dim3(256), 0, 0, a, b, o, c.rows, v16);
            else if (c.kind == 2) hipLaunchKernelGGL(copy_rows<2>, dim3(c.grid), dim3(256), 0, 0, a, b, o, c.rows, v16);
            else hipLaunchKernelGGL(read_rows, dim3(c.grid), dim3(256), 0, 0, a, f, c.rows, v16);
        };
        for (int i = 0; i < 5; ++i) launch();
        CK(hipDeviceSynchronize());
        const int n = 50;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < n; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)c.rows * c.cols_bytes * (c.kind == 0 ? 1 : c.kind == 1 ? 2 : 3);
        printf("%-48s %8.2f us/launch  %6.2f TB/s (back to back, gaps included)\n", c.name, ms / n * 1e3, bytes / (ms / n * 1e-3) / 1e12);
    }
    return 0;
}
