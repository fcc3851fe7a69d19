// How far ahead of its MFMA must an LDS fragment read be issued at ONE wave per SIMD?  A loop of v_mfma_f32_32x32x16_bf16 whose
// A and B operands come from LDS (two ds_read_b128 per MFMA, or four ds_read_b64_tr_b16), read DIST MFMAs ahead into a register
// ring; cycles per MFMA (s_memtime) by DIST.   hipcc --offload-arch=gfx950 -O3 lds_mfma_dist.hip -o lds_mfma_dist.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

template <int DIST, int TR>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3f803f80u + i * 2654435761u % 64;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int R = DIST + 1;
    bf16x8 fa[R], fb[R];
    f32x16 acc = {};
    // conflict-free addressing: lane * 16 bytes within a 1-KiB piece (b128); tr: 8-byte pieces
    const char* base = smem + wave * 8192;
    auto rd = [&](int i, int slot) {
        const char* pa = base + ((i * 2) % 8) * 1024, *pb = base + ((i * 2 + 1) % 8) * 1024;
        if (TR) {
            s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + lane * 8));
            s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 512 + lane * 8));
            s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + lane * 8));
            s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + 512 + lane * 8));
            fa[slot] = __builtin_shufflevector(__builtin_bit_cast(bf16x4, a0), __builtin_bit_cast(bf16x4, a1), 0, 1, 2, 3, 4, 5, 6, 7);
            fb[slot] = __builtin_shufflevector(__builtin_bit_cast(bf16x4, b0), __builtin_bit_cast(bf16x4, b1), 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            fa[slot] = *reinterpret_cast<const bf16x8*>(pa + lane * 16);
            fb[slot] = *reinterpret_cast<const bf16x8*>(pb + lane * 16);
        }
    };
#pragma unroll
    for (int i = 0; i < DIST; ++i) rd(i, i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < R * 4; ++j) {   // a multiple of the ring size: static slots
            rd(j + DIST, (j + DIST) % R);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[j % R], fb[j % R], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = (t1 - t0) / (unsigned long long)(iters * R * 4);
}

template <int DIST, int TR>
void run(float* out, unsigned long long* cyc) {
    const int iters = 2000;
    (void)hipFuncSetAttribute((const void*)k<DIST, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((k<DIST, TR>), dim3(256), dim3(256), 65536, 0, out, iters, cyc);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= 256;
    printf("%s reads, issued %d MFMA(s) ahead: %6.1f cycles per MFMA\n", TR ? "4 x tr_b64" : "2 x b128  ", DIST, m);
}
int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    run<1, 0>(out, cyc); run<2, 0>(out, cyc); run<3, 0>(out, cyc); run<4, 0>(out, cyc); run<6, 0>(out, cyc); run<8, 0>(out, cyc);
    run<1, 1>(out, cyc); run<2, 1>(out, cyc); run<3, 1>(out, cyc); run<4, 1>(out, cyc); run<6, 1>(out, cyc); run<8, 1>(out, cyc);
    return 0;
}
