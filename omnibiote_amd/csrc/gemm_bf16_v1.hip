// bf16 GEMM, first structure (kept selectable with OBTE_GEMM=v1 for A/B runs; the default is gemm_bf16_v2.hip).
// bf16 GEMM on CDNA4 MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate, bf16 output with fused epilogues.
//
// Replaces the nn.Linear calls of the reference's block and readout (training/model.py:102,151,163,166,253)
// in all three passes: forward (A,B k-contiguous), dgrad (B k-strided), wgrad (A and B k-strided).
//
// Structure: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4 MFMA tiles),
// K step 64, two LDS stages filled by LDS-DMA (buffer_load ... lds, 16 B per lane, hardware zero fill past the
// end of the tensor), XOR-swizzled so that fragment reads are bank-conflict free:
//   k-contiguous operand : LDS image [128 rows][64 k], 128-B rows, 16-B chunk c of row r stored at c ^ ((r>>1)&7);
//                          fragments by ds_read_b128.
//   k-strided operand    : LDS image [64 k][128 mn], 256-B rows, chunk c of k-row r at c ^ f(r),
//                          f(r) = ((r&3) | ((r>>3)&1)<<2) << 1; fragments by ds_read_b64_tr_b16 (hardware
//                          transpose), two per fragment.
// LDS-DMA writes LDS linearly (wave base + lane*16), so the swizzle is applied to each lane's SOURCE address.
// The MFMA is issued with the operands swapped (computes C^T tiles) so that every lane ends up holding four
// consecutive output columns; the tile is then staged through LDS as bf16 and written with 16-B stores in
// full 128-B row segments, where the epilogue (GELU, residual add, GELU backward) is applied.
// Workgroup ids are remapped so that each XCD (own L2) works on a compact 8 x n group of tiles.
#include "common.h"

namespace obte_gemm_v1 {

constexpr int BM = 128, BN = 128, BKT = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;      // one operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // A + B
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;   // two stages = 64 KiB -> 2 workgroups per CU
constexpr int EPI_LD = 144;                   // bytes per staged output row: 64 bf16 + 16 B pad
constexpr int EPI_WAVE_BYTES = 64 * EPI_LD;

struct GemmParams {
    const bf16* a; const bf16* b; bf16* d; const bf16* aux; bf16* d2;
    int64_t M, N, K, lda, ldb, ldd;
    int64_t a_elems, b_elems;
    int tiles_m, tiles_n;
    float alpha;
    DropCfg drop;
    const float* rope_cos; const float* rope_sin; int64_t rope_T; int rope_hs;
};

__device__ __forceinline__ int kmaj_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int mnmaj_f(int krow) { return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1; }
__device__ __forceinline__ int mnmaj_off(int krow, int chunk) { return krow * 256 + ((chunk ^ mnmaj_f(krow)) << 4); }

// Per-lane byte offsets (relative to the tile origin in global memory) of the four LDS-DMA pieces a lane issues.
template <bool KMAJOR>
__device__ __forceinline__ void dma_offsets(int wave, int lane, int64_t ld, int (&voff)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = i * 4 + wave;  // 1-KiB piece of the 16-KiB image
        if (KMAJOR) {
            const int row = slot * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            voff[i] = (int)((row * ld + chunk * 8) * 2);
        } else {
            const int krow = slot * 4 + (lane >> 4);
            const int chunk = (lane & 15) ^ mnmaj_f(krow);
            voff[i] = (int)((krow * ld + chunk * 8) * 2);
        }
    }
}

__device__ __forceinline__ void dma_tile(const bf16* origin, int64_t elems_left, const int (&voff)[4], char* lds_tile, int wave) {
    // through inline asm (common.h lds_dma16): the builtin makes hipcc drain vmcnt in front of later LDS reads
    const i32x4_t rsrc = make_rsrc_words(origin, elems_left * 2);
    const uint32_t base = lds_addr_of(lds_tile) + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_dma16(rsrc, base + i * 4 * 1024, voff[i]);
}

// Fragment of 16 (m or n) x 32 (k) for v_mfma_f32_16x16x32_bf16: lane l holds index (l&15), k = 8*(l>>4)+j.
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 load_frag(const char* tile, int mn0, int s, int lane) {
    if (KMAJOR) {
        const int row = mn0 + (lane & 15);
        const int chunk = 4 * s + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(tile + kmaj_off(row, chunk));
    } else {
        const int li = lane & 15;
        const int krow = 32 * s + 8 * (lane >> 4) + (li >> 2);
        const int chunk = (mn0 >> 3) + ((li & 3) >> 1);
        const int sub = (li & 1) * 8;
        const bf16x4 lo = lds_read_tr16(tile + mnmaj_off(krow, chunk) + sub);
        const bf16x4 hi = lds_read_tr16(tile + mnmaj_off(krow + 4, chunk) + sub);
        return join8(lo, hi);
    }
}

template <bool A_KMAJOR, bool B_KMAJOR, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    // ---- workgroup -> tile: bijective XCD remap, then 8-row groups of tiles -------------------------------
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int group_sz = 8 * p.tiles_n;
    const int first_m = (wgid / group_sz) * 8;
    const int gsz = min(p.tiles_m - first_m, 8);
    const int tm = first_m + (wgid % group_sz) % gsz;
    const int tn = (wgid % group_sz) / gsz;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    int voff_a[4], voff_b[4];
    dma_offsets<A_KMAJOR>(wave, lane, p.lda, voff_a);
    dma_offsets<B_KMAJOR>(wave, lane, p.ldb, voff_b);

    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (int)((p.K + BKT - 1) / BKT);
    auto issue = [&](int t, int stage) {
        const int64_t k0 = (int64_t)t * BKT;
        const int64_t ao = A_KMAJOR ? (m0 * p.lda + k0) : (k0 * p.lda + m0);
        const int64_t bo = B_KMAJOR ? (n0 * p.ldb + k0) : (k0 * p.ldb + n0);
        char* st = smem + stage * STAGE_BYTES;
        dma_tile(p.a + ao, p.a_elems - ao, voff_a, st, wave);
        dma_tile(p.b + bo, p.b_elems - bo, voff_b, st + TILE_BYTES, wave);
    };

    // ---- main loop, software pipelined over half K-tiles (same scheme as gemm_bf16_v2.hip) ---------------------
    //   step t:  read F1(t) | MFMA F0(t) | wait tile t+1, barrier | read F0(t+1), issue DMA of tile t+2 | MFMA F1(t)
    auto load_frags = [&](int t, int ks, bf16x8 (&af)[4], bf16x8 (&bfr)[4]) {
        const char* ta = smem + (t & 1) * STAGE_BYTES;
        const char* tb = ta + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = load_frag<A_KMAJOR>(ta, wm * 64 + i * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) bfr[i] = load_frag<B_KMAJOR>(tb, wn * 64 + i * 16, ks, lane);
    };
    auto mma = [&](const bf16x8 (&af)[4], const bf16x8 (&bfr)[4]) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                // operands swapped: the accumulator holds C^T (row = n, col = m)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[ni][mi], 0, 0, 0);
    };
    bf16x8 a0[4], b0[4], a1[4], b1[4];
    issue(0, 0);
    if (nk > 1) {
        issue(1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile 0 landed; the 8 pieces of tile 1 stay in flight
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, a0, b0);
    for (int t = 0; t + 1 < nk; ++t) {
        load_frags(t, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): tile t+1 landed, F1(t) in registers
        __builtin_amdgcn_s_barrier();
        load_frags(t + 1, 0, a0, b0);
        if (t + 2 < nk) issue(t + 2, t & 1);  // the stage every wave has just finished reading
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    load_frags(nk - 1, 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, b0);
    mma(a1, b1);
    __syncthreads();

    // ---- epilogue: acc -> bf16 -> LDS (per-wave region) -> 16-B row-contiguous stores ---------------------
    char* stg = smem + wave * EPI_WAVE_BYTES;
    const int em = lane & 15, en = (lane >> 4) * 4;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = f2bf(acc[ni][mi][r] * p.alpha);
            *reinterpret_cast<bf16x4*>(stg + (mi * 16 + em) * EPI_LD + (ni * 16 + en) * 2) = v;
        }
    // interior tiles whose epilogue reads global memory: every such load before the first store (see gemm_bf16_v2.hip,
    // tile_epilogue: vmcnt retires in issue order, so a load behind a store waits for the store)
    constexpr bool READS = EPI == OBTE_EPI_ADD || EPI == OBTE_EPI_GELU_BWD || EPI == OBTE_EPI_ADD_DROPOUT || EPI == OBTE_EPI_ROPE_QK;
    if (READS && m0 + 128 <= p.M && n0 + 128 <= p.N) {
        const int row_l = lane >> 3, c8 = lane & 7;
        const int64_t n = n0 + wn * 64 + c8 * 8;
        const int64_t o0 = (m0 + wm * 64 + row_l) * p.ldd + n;
        bf16x8 r[8];
        f32x4 rc[8], rs[8];
        bool rot = false;
        if (EPI == OBTE_EPI_ROPE_QK) {
            rot = n < 2 * (p.N / 3);
            const uint32_t T32 = (uint32_t)p.rope_T, hs32 = (uint32_t)p.rope_hs, nu = (uint32_t)n;
            const uint32_t dd = (hs32 & (hs32 - 1)) == 0 ? (nu & (hs32 - 1)) : (nu % hs32);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const uint32_t mu = (uint32_t)(m0 + wm * 64 + row_l + it * 8);
                const uint32_t t = (T32 & (T32 - 1)) == 0 ? (mu & (T32 - 1)) : (mu % T32);
                rc[it] = *reinterpret_cast<const f32x4*>(p.rope_cos + t * (hs32 / 2) + dd / 2);
                rs[it] = *reinterpret_cast<const f32x4*>(p.rope_sin + t * (hs32 / 2) + dd / 2);
            }
        } else {
#pragma unroll
            for (int it = 0; it < 8; ++it) r[it] = *reinterpret_cast<const bf16x8*>(p.aux + o0 + (int64_t)it * 8 * p.ldd);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + (it * 8 + row_l) * EPI_LD + c8 * 16);
            if (EPI == OBTE_EPI_ADD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(r[it][j]) + bf2f(v[j]));
            } else if (EPI == OBTE_EPI_GELU_BWD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * bf2f(r[it][j]));
            } else if (EPI == OBTE_EPI_ROPE_QK) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xe = bf2f(v[2 * j]), xo = bf2f(v[2 * j + 1]);
                    const bf16 ne = f2bf(xe * rc[it][j] - xo * rs[it][j]), no = f2bf(xe * rs[it][j] + xo * rc[it][j]);
                    v[2 * j] = rot ? ne : v[2 * j];
                    v[2 * j + 1] = rot ? no : v[2 * j + 1];
                }
            } else if (EPI == OBTE_EPI_ADD_DROPOUT) {
                const int64_t m = m0 + wm * 64 + row_l + it * 8;
                const uint32_t rk = drop_rowkey((uint64_t)m, p.drop);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const uint32_t bits = drop_pair_bits(rk, (uint32_t)(n >> 1) + jj);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * jj + e;
                        const float t = drop_keep_bits(bits, (uint32_t)e, p.drop) ? bf2f(f2bf(bf2f(v[j]) * p.drop.scale)) : 0.f;
                        v[j] = f2bf(bf2f(r[it][j]) + t);
                    }
                }
            }
            *reinterpret_cast<bf16x8*>(p.d + o0 + (int64_t)it * 8 * p.ldd) = v;
        }
        return;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3);
        const int c8 = lane & 7;
        const int64_t m = m0 + wm * 64 + row;
        const int64_t n = n0 + wn * 64 + c8 * 8;
        if (m < p.M && n < p.N) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * EPI_LD + c8 * 16);
            const int64_t o = m * p.ldd + n;
            if (EPI == OBTE_EPI_GELU) {
                // one erf/exp evaluation yields both the activation (d2) and its derivative (d): the backward
                // epilogue is then a plain multiply
                bf16x8 g;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    f32x2_t act, der;
                    gelu_ref_both2(f32x2_t{bf2f(v[j]), bf2f(v[j + 1])}, act, der);
                    g[j] = f2bf(act[0]); g[j + 1] = f2bf(act[1]);
                    v[j] = f2bf(der[0]); v[j + 1] = f2bf(der[1]);
                }
                *reinterpret_cast<bf16x8*>(p.d2 + o) = g;
            } else if (EPI == OBTE_EPI_ADD) {
                const bf16x8 r = *reinterpret_cast<const bf16x8*>(p.aux + o);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(r[j]) + bf2f(v[j]));
            } else if (EPI == OBTE_EPI_ROPE_QK) {
                // packed c_attn output [.., 3C]: rotate the (even, odd) pairs of the q and k thirds (columns < 2N/3),
                // position = row % T; the v third passes through.  fp32 arithmetic on the bf16-rounded projection.
                if (n < 2 * (p.N / 3)) {
                    // position and in-head column by 32-bit arithmetic, masks when T / head_dim are powers of two (uniform
                    // branches): the 64-bit `%` this replaced was a software division per 16-byte chunk — 9 to 22 us of
                    // the 65-us c_attn GEMM
                    const uint32_t mu = (uint32_t)m, nu = (uint32_t)n, T32 = (uint32_t)p.rope_T, hs32 = (uint32_t)p.rope_hs;
                    const uint32_t t = (T32 & (T32 - 1)) == 0 ? (mu & (T32 - 1)) : (mu % T32);
                    const uint32_t dd = (hs32 & (hs32 - 1)) == 0 ? (nu & (hs32 - 1)) : (nu % hs32);
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.rope_cos + t * (hs32 / 2) + dd / 2);
                    const f32x4 sn = *reinterpret_cast<const f32x4*>(p.rope_sin + t * (hs32 / 2) + dd / 2);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xe = bf2f(v[2 * j]), xo = bf2f(v[2 * j + 1]);
                        v[2 * j] = f2bf(xe * c[j] - xo * sn[j]);
                        v[2 * j + 1] = f2bf(xe * sn[j] + xo * c[j]);
                    }
                }
            } else if (EPI == OBTE_EPI_ADD_DROPOUT) {
                const bf16x8 r = *reinterpret_cast<const bf16x8*>(p.aux + o);
                const uint32_t rk = drop_rowkey((uint64_t)m, p.drop);   // dropout element = (row m, column n + j)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const uint32_t bits = drop_pair_bits(rk, (uint32_t)(n >> 1) + jj);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * jj + e;
                        const float t = drop_keep_bits(bits, (uint32_t)e, p.drop) ? bf2f(f2bf(bf2f(v[j]) * p.drop.scale)) : 0.f;
                        v[j] = f2bf(bf2f(r[j]) + t);
                    }
                }
            } else if (EPI == OBTE_EPI_GELU_BWD) {
                const bf16x8 h = *reinterpret_cast<const bf16x8*>(p.aux + o);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * bf2f(h[j]));
            }
            *reinterpret_cast<bf16x8*>(p.d + o) = v;
        }
    }
}

template <bool AK, bool BK, int EPI>
int launch(const GemmParams& p, hipStream_t st) {
    static bool attr_set = false;  // idempotent; a race only repeats the call
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<AK, BK, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK, EPI>), dim3(p.tiles_m * p.tiles_n), dim3(256), SMEM_BYTES, st, p);
    OBTE_CHECK_LAUNCH("obte_gemm_bf16");
    return OBTE_OK;
}

template <bool AK, bool BK>
int dispatch_epi(const GemmParams& p, int epi, hipStream_t st) {
    switch (epi) {
        case OBTE_EPI_NONE: return launch<AK, BK, OBTE_EPI_NONE>(p, st);
        case OBTE_EPI_GELU: return launch<AK, BK, OBTE_EPI_GELU>(p, st);
        case OBTE_EPI_ADD: return launch<AK, BK, OBTE_EPI_ADD>(p, st);
        case OBTE_EPI_GELU_BWD: return launch<AK, BK, OBTE_EPI_GELU_BWD>(p, st);
        case OBTE_EPI_ADD_DROPOUT: return launch<AK, BK, OBTE_EPI_ADD_DROPOUT>(p, st);
        case OBTE_EPI_ROPE_QK: return launch<AK, BK, OBTE_EPI_ROPE_QK>(p, st);
    }
    obte_set_error("obte_gemm_bf16: unknown epilogue %d", epi);
    return OBTE_EINVAL;
}

}  // namespace obte_gemm_v1
using namespace obte_gemm_v1;

int obte_gemm_bf16_v1(const obte_gemm_args* g, obte_stream s) {
    OBTE_REQUIRE(g && g->a && g->b && g->d, "obte_gemm_bf16: null pointer");
    OBTE_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0, "obte_gemm_bf16: empty problem M=%lld N=%lld K=%lld",
                 (long long)g->M, (long long)g->N, (long long)g->K);
    OBTE_REQUIRE(g->lda % 8 == 0 && g->ldb % 8 == 0 && g->ldd % 8 == 0 && g->N % 8 == 0,
                 "obte_gemm_bf16: lda/ldb/ldd/N must be multiples of 8 (16-byte rows)");
    OBTE_REQUIRE(!(g->a_kmajor) || g->K % 64 == 0, "obte_gemm_bf16: k-contiguous A needs K %% 64 == 0 (K=%lld)", (long long)g->K);
    OBTE_REQUIRE(!(g->b_kmajor) || g->K % 64 == 0, "obte_gemm_bf16: k-contiguous B needs K %% 64 == 0 (K=%lld)", (long long)g->K);
    OBTE_REQUIRE(g->a_kmajor ? g->lda >= g->K : g->lda >= g->M, "obte_gemm_bf16: lda too small");
    OBTE_REQUIRE(g->b_kmajor ? g->ldb >= g->K : g->ldb >= g->N, "obte_gemm_bf16: ldb too small");
    OBTE_REQUIRE(g->ldd >= g->N, "obte_gemm_bf16: ldd too small");
    OBTE_REQUIRE(g->lda <= 1 << 20 && g->ldb <= 1 << 20, "obte_gemm_bf16: leading dimension too large");
    if (g->epilogue == OBTE_EPI_ADD || g->epilogue == OBTE_EPI_GELU_BWD || g->epilogue == OBTE_EPI_ADD_DROPOUT) OBTE_REQUIRE(g->aux, "obte_gemm_bf16: epilogue needs aux");
    if (g->epilogue == OBTE_EPI_ADD_DROPOUT) OBTE_REQUIRE(g->dropout_p >= 0.f && g->dropout_p < 1.f, "obte_gemm_bf16: dropout p must be in [0,1)");
    if (g->epilogue == OBTE_EPI_ROPE_QK)
        OBTE_REQUIRE(g->rope_cos && g->rope_sin && g->rope_T > 0 && g->rope_head_dim > 0 && g->rope_head_dim % 8 == 0 && g->N % 3 == 0 &&
                         (g->N / 3) % g->rope_head_dim == 0,
                     "obte_gemm_bf16: EPI_ROPE_QK needs cos/sin tables, T, head_dim %% 8 == 0 and N = 3 * n_head * head_dim");
    if (g->epilogue == OBTE_EPI_GELU) OBTE_REQUIRE(g->d2, "obte_gemm_bf16: GELU epilogue needs d2");
    if (g->epilogue != OBTE_EPI_NONE && g->epilogue != OBTE_EPI_ADD) OBTE_REQUIRE(g->alpha == 1.0f, "obte_gemm_bf16: alpha != 1 only with EPI_NONE / EPI_ADD");
    GemmParams p;
    p.a = (const bf16*)g->a; p.b = (const bf16*)g->b; p.d = (bf16*)g->d; p.aux = (const bf16*)g->aux; p.d2 = (bf16*)g->d2;
    p.M = g->M; p.N = g->N; p.K = g->K; p.lda = g->lda; p.ldb = g->ldb; p.ldd = g->ldd;
    p.a_elems = (g->a_kmajor ? g->M : g->K) * g->lda;
    p.b_elems = (g->b_kmajor ? g->N : g->K) * g->ldb;
    const int64_t tm = cdiv64(g->M, BM), tn = cdiv64(g->N, BN);
    OBTE_REQUIRE(tm * tn < (1ll << 30), "obte_gemm_bf16: too many tiles");
    p.tiles_m = (int)tm; p.tiles_n = (int)tn;
    p.alpha = g->alpha;
    p.rope_cos = g->rope_cos; p.rope_sin = g->rope_sin; p.rope_T = g->rope_T; p.rope_hs = g->rope_head_dim;
    p.drop = make_drop(g->epilogue == OBTE_EPI_ADD_DROPOUT ? g->dropout_p : 0.f, g->dropout_seed, (uint32_t)g->dropout_site);
    hipStream_t st = (hipStream_t)s;
    if (g->a_kmajor && g->b_kmajor) return dispatch_epi<true, true>(p, g->epilogue, st);
    if (g->a_kmajor && !g->b_kmajor) return dispatch_epi<true, false>(p, g->epilogue, st);
    if (!g->a_kmajor && g->b_kmajor) return dispatch_epi<false, true>(p, g->epilogue, st);
    return dispatch_epi<false, false>(p, g->epilogue, st);
}
