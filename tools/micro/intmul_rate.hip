// Issue-rate probe: v_mul_lo_u32 vs v_mul_u32_u24 vs v_xor_b32 (independent chains, 8 per thread), gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/micro/intmul_rate.hip -o /tmp/intmul && /tmp/intmul
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ void k(unsigned* out, unsigned seed, int iters) {
    unsigned x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 8 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) x[i] = x[i] * 0x7feb352dU;
            else if (KIND == 1) x[i] = __umul24(x[i], 0x7feb35) ;
            else if (KIND == 2) x[i] = x[i] ^ (x[i] >> 15);
            else x[i] = __umulhi(x[i], 0x7feb352dU);
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> float run(unsigned* d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(256 * 8), dim3(256), 0, 0, d, 1u, iters);
    hipEventRecord(a); hipLaunchKernelGGL(k<KIND>, dim3(256 * 8), dim3(256), 0, 0, d, 1u, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 4096;
    const char* names[4] = {"v_mul_lo_u32", "v_mul_u32_u24", "xor-shift (2 ops)", "v_mul_hi_u32"};
    float t[4] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters)};
    for (int i = 0; i < 4; ++i) {
        const double ops = 256.0 * 8 * 256 * 8 * iters;   // thread-level ops
        printf("%-18s %8.3f ms  %7.1f Gop/s/thread-lanes  (%.2f cycles per wave-instruction per SIMD at 2.1 GHz, 8 waves/SIMD)\n", names[i], t[i], ops / t[i] / 1e6,
               t[i] * 1e-3 * 2.1e9 / (8.0 * 8 * iters * (256.0 * 8 * 4 / (256 * 4)) ));
    }
    return 0;
}
