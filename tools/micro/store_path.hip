// How fast can ONE CU store?  (the GEMM epilogues issue a 128-KB tile in ~4.1 us whether 32 or 256 workgroups store.)
// G workgroups of 512 threads (one per CU for G <= 256), each storing BYTES of its own region in 16-byte pieces, fire and forget,
// in two shapes: (a) fully contiguous (1 KiB per wave instruction), (b) the epilogue's shape (4 rows x 256 B per wave instruction,
// rows 8 KiB apart), (c) round 5: what a store straight from the 16x16x32 accumulators after one v_permlane16_swap per dword
// looks like — 16 rows x 64 B per wave instruction, the two halves of a row's 128-B line in consecutive instructions, (d) the same
// with the instructions of one line NOT adjacent (all first halves, then all second halves).  s_memtime around the issue loop (leader wave), median over workgroups; then the drain.
//   hipcc --offload-arch=gfx950 -O3 -o store_path.bin store_path.hip && ./store_path.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int SHAPE>
__global__ __launch_bounds__(512) void store_kernel(u32x4* out, int bytes, unsigned long long* t) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char* base = reinterpret_cast<char*>(out) + (size_t)blockIdx.x * (1 << 20);
    const u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
    const int n = bytes / (512 * 16);   // store instructions per thread
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        size_t off;
        if (SHAPE == 0) off = ((size_t)i * 8 + wave) * 1024 + lane * 16;
        else if (SHAPE == 1) off = ((size_t)(i * 4 + lane / 16) * 8192) + wave * 256 + (lane % 16) * 16;   // 4 rows x 256 B per instruction
        else if (SHAPE == 2) off = ((size_t)((i >> 2) * 16 + (lane & 15)) * 8192) + wave * 256 + (i & 3) * 64 + (lane >> 4) * 16;   // 16 rows x 64 B, a row's four pieces back to back
        else { const int n4 = n >> 2; off = ((size_t)((i % n4) * 16 + (lane & 15)) * 8192) + wave * 256 + (i / n4) * 64 + (lane >> 4) * 16; }   // piece by piece
        *reinterpret_cast<u32x4*>(base + off) = v;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = t1 - t0; t[blockIdx.x * 2 + 1] = t2 - t0; }
}
int main() {
    u32x4* out; unsigned long long* t;
    hipMalloc(&out, (size_t)256 << 20); hipMalloc(&t, 256 * 2 * 8);
    for (int shape = 0; shape < 4; ++shape)
        for (int g : {8, 32, 128, 256})
            for (int bytes : {131072, 262144}) {
                std::vector<unsigned long long> h(512);
                for (int rep = 0; rep < 3; ++rep) {
                    if (shape == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(g), dim3(512), 0, 0, out, bytes, t);
                    else if (shape == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(g), dim3(512), 0, 0, out, bytes, t);
                    else if (shape == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(g), dim3(512), 0, 0, out, bytes, t);
                    else hipLaunchKernelGGL(store_kernel<3>, dim3(g), dim3(512), 0, 0, out, bytes, t);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), t, g * 16, hipMemcpyDeviceToHost);
                std::vector<unsigned long long> a, b;
                for (int i = 0; i < g; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
                std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
                printf("shape %d  %3d workgroups  %3d KB each: issue %6llu cycles (%.1f B/clk/CU), drained %6llu cycles\n", shape, g, bytes >> 10,
                       a[g / 2], (double)bytes / a[g / 2], b[g / 2]);
            }
    return 0;
}
