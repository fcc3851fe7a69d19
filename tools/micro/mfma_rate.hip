// MFMA issue rate of ONE wave per SIMD (4 waves per CU, every CU busy), v_mfma_f32_32x32x16_bf16 and 16x16x32, by accumulator
// placement (architectural VGPRs / AGPRs) and dependence pattern.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define M32V(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define M32A(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define M16V(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* in, float* out, int iters, unsigned long long* cyc) {
    bf16x8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x4 d0 = {}, d1 = {}, d2 = {}, d3 = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { M32V(c0); M32V(c0); M32V(c0); M32V(c0); M32V(c0); M32V(c0); M32V(c0); M32V(c0); }            // one chain, VGPR
        if (MODE == 1) { M32V(c0); M32V(c1); M32V(c2); M32V(c3); M32V(c0); M32V(c1); M32V(c2); M32V(c3); }            // four chains, VGPR
        if (MODE == 2) { M32A(c0); M32A(c0); M32A(c0); M32A(c0); M32A(c0); M32A(c0); M32A(c0); M32A(c0); }            // one chain, AGPR
        if (MODE == 3) { M32A(c0); M32A(c1); M32A(c2); M32A(c3); M32A(c0); M32A(c1); M32A(c2); M32A(c3); }            // four chains, AGPR
        if (MODE == 4) { M16V(d0); M16V(d1); M16V(d2); M16V(d3); M16V(d0); M16V(d1); M16V(d2); M16V(d3); }            // 16x16x32, four chains
        if (MODE == 5) { M32V(c0); M32V(c1); M32V(c0); M32V(c1); M32V(c0); M32V(c1); M32V(c0); M32V(c1); }            // two chains alternating
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int r = 0; r < 4; ++r) s += d0[r] + d1[r] + d2[r] + d3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const bf16x8* in, float* out, unsigned long long* cyc, int iters, double flop_per_mfma) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, in, out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, in, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= 256;
    const double n = 8.0 * iters;
    printf("%-34s %8.1f us  %6.1f memtime-ticks/MFMA  %6.1f ns/MFMA  %7.1f TFLOP/s chip\n", name, ms * 1e3, mean / n, ms * 1e6 / n,
           n * flop_per_mfma * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
    bf16x8* in; float* out; unsigned long long* cyc;
    hipMalloc(&in, 512 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    std::vector<unsigned short> h(512 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3f00 + (i * 2654435761u >> 20) % 256 + ((i & 1) ? 0x8000 : 0));
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 20000;
    const double f32 = 2.0 * 32 * 32 * 16, f16 = 2.0 * 16 * 16 * 32;
    run<0>("32x32x16 one chain   VGPR", in, out, cyc, iters, f32);
    run<1>("32x32x16 four chains VGPR", in, out, cyc, iters, f32);
    run<5>("32x32x16 two chains  VGPR", in, out, cyc, iters, f32);
    run<2>("32x32x16 one chain   AGPR", in, out, cyc, iters, f32);
    run<3>("32x32x16 four chains AGPR", in, out, cyc, iters, f32);
    run<4>("16x16x32 four chains VGPR", in, out, cyc, iters, f16);
    return 0;
}
