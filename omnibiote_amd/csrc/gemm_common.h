// Shared pieces of the bf16 MFMA GEMM structures (gemm_bf16_v2.hip: structures 2-5 and the grouped launch; gemm_bf16_v7.hip: the
// persistent continuous-ring structure): kernel parameters, LDS images and their swizzles, LDS-DMA piece offsets, fragment reads.
#pragma once
#include "common.h"

namespace obte_gemm_v2 {

constexpr int BM = 256, BKT = 64;
constexpr int NTHREADS = 512;
constexpr int A_TILE = BM * BKT * 2;            // 32 KiB
// Two tile widths.  BN = 128: 48 KiB per stage, 3-stage ring (loads two K-tiles ahead) — used where the grid
// would otherwise not fill the chip (N = 1024 outputs).  BN = 256: 64 KiB per stage, two stages — 131 FLOP per
// loaded byte instead of 87; a CU pulls ~70 GB/s from its L2 and ~25-30 GB/s from beyond it (measured), and at
// ~8 TFLOP/s per CU a 256x128 tile needs 94 GB/s: the wide tile is what keeps the MFMAs fed.
template <int BN> struct Cfg {
    static constexpr int B_TILE = BN * BKT * 2;
    static constexpr int STAGE = A_TILE + B_TILE;
    static constexpr int NSTAGE = BN == 128 ? 3 : 2;
    static constexpr int EPI_BYTES = 8 * 64 * 272;       // epilogue staging: 8 waves x 64 rows x (<=128 bf16 | 64 f32, + pad)
    static constexpr int SMEM = NSTAGE * STAGE > EPI_BYTES ? NSTAGE * STAGE : EPI_BYTES;   // 144 KiB / 136 KiB: one workgroup per CU
    static constexpr int NPB = BN / 64;                  // LDS-DMA pieces per wave for the B tile (A: 4); BN = 192: 3
    static constexpr int NJ = BN / 32;                   // 16-wide n sub-tiles per wave (wave tile 64 x BN/2)
};
constexpr int EPI_LD_F32 = 272;                 // bytes per staged f32 row: 64 f32 + 16 B pad

struct GemmParams {
    const bf16* a; const bf16* b; bf16* d; const bf16* aux; bf16* d2; float* slab;
    int64_t M, N, K, lda, ldb, ldd;
    int64_t store_rows;   // = M; 0 in the timing-only 'nostore' diagnostic
    int nt_store;         // non-temporal output stores (outputs that exceed the 256-MiB Infinity Cache)
    int delay_sleeps;     // debug build only: every second workgroup of a CU (odd hardware wave slot) sleeps this many x 3.4 us first
    unsigned long long* dbg_times;   // debug build only (OBTE_GEMM_TIMES=1): per workgroup, s_memrealtime at entry / first operands landed / loop end / stores issued / stores done
    int64_t a_elems, b_elems;
    int tiles_m, tiles_n, splits, k_per_split;   // k_per_split in K-tiles
    float alpha;
    DropCfg drop;
    const float* rope_cos; const float* rope_sin; int64_t rope_T; int rope_hs;
};

#ifdef OBTE_DEBUG_HOOKS
#define OBTE_GSTAMP(p, k) do { if ((p).dbg_times && threadIdx.x == 0) (p).dbg_times[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define OBTE_GSTAMP(p, k) do { } while (0)
#endif
__device__ __forceinline__ int kmaj_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int mn_f(int krow) { return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1; }
template <int ROWB>
__device__ __forceinline__ int mnmaj_off(int krow, int chunk) { return krow * ROWB + ((chunk ^ mn_f(krow)) << 4); }

// byte offsets, relative to the tile origin, of the LDS-DMA pieces this lane issues (piece = wave + 8*i)
template <bool KMAJOR, int MN, int NP>
__device__ __forceinline__ void dma_offsets(int wave, int lane, int64_t ld, int (&voff)[NP]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int piece = wave + 8 * i;  // 1 KiB of the image
        if (KMAJOR) {
            const int row = piece * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            voff[i] = (int)((row * ld + chunk * 8) * 2);
        } else {
            constexpr int CPR = MN / 8;          // 16-B chunks per k-row: 32 (A) or 16 (B)
            constexpr int RPP = 64 / CPR;        // k-rows per 1-KiB piece: 2 or 4
            const int krow = piece * RPP + lane / CPR;
            const int chunk = (lane % CPR) ^ mn_f(krow);
            voff[i] = (int)((krow * ld + chunk * 8) * 2);
        }
    }
}

template <int NP>
__device__ __forceinline__ void dma_tile(const bf16* origin, int64_t elems_left, const int (&voff)[NP], char* lds_tile, int wave) {
#ifdef OBTE_DMA_BUILTIN
    __amdgpu_buffer_rsrc_t rsrc = make_rsrc(origin, elems_left * 2);
#pragma unroll
    for (int i = 0; i < NP; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + (wave + 8 * i) * 1024), 16, voff[i], 0, 0, 0);
#else
    const i32x4_t rsrc = make_rsrc_words(origin, elems_left * 2);
    const uint32_t base = lds_addr_of(lds_tile) + wave * 1024;
#pragma unroll
    for (int i = 0; i < NP; ++i) lds_dma16(rsrc, base + 8 * i * 1024, voff[i]);
#endif
}

// 16 (m or n) x 32 (k) fragment for v_mfma_f32_16x16x32_bf16: lane l holds index (l&15), k = 8*(l>>4)+j.
template <bool KMAJOR, int MN>
__device__ __forceinline__ bf16x8 load_frag(const char* tile, int mn0, int s, int lane) {
    if (KMAJOR) {
        return *reinterpret_cast<const bf16x8*>(tile + kmaj_off(mn0 + (lane & 15), 4 * s + (lane >> 4)));
    } else {
        const int li = lane & 15;
        const int krow = 32 * s + 8 * (lane >> 4) + (li >> 2);
        const int chunk = (mn0 >> 3) + ((li & 3) >> 1);
        const int sub = (li & 1) * 8;
        const bf16x4 lo = lds_read_tr16(tile + mnmaj_off<MN * 2>(krow, chunk) + sub);
        const bf16x4 hi = lds_read_tr16(tile + mnmaj_off<MN * 2>(krow + 4, chunk) + sub);
        return join8(lo, hi);
    }
}

// ---- the half-tile ring's images (structures 3, 4, 7) ----------------------------------------------------------------------------
constexpr int H_TILE = BM * 32 * 2;             // 16 KiB per operand per half-stage
constexpr int H_STAGE = 2 * H_TILE;             // 32 KiB
constexpr int V3_RING = 4 * H_STAGE;            // 128 KiB
constexpr int V3_SMEM = V3_RING > Cfg<256>::EPI_BYTES ? V3_RING : Cfg<256>::EPI_BYTES;

__device__ __forceinline__ int kmaj32_off(int row, int chunk) { return row * 64 + ((chunk ^ ((4 - (row >> 2)) & 3)) << 4); }

template <bool KMAJOR>
__device__ __forceinline__ void dma_offsets_h(int wave, int lane, int64_t ld, int (&voff)[2]) {
    if (KMAJOR) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave + 8 * i) * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((4 - (row >> 2)) & 3);
            voff[i] = (int)((row * ld + chunk * 8) * 2);
        }
    } else {
        dma_offsets<false, 256, 2>(wave, lane, ld, voff);   // 32 k-rows x 512 B = pieces 0..15
    }
}

template <bool KMAJOR>
__device__ __forceinline__ bf16x8 load_frag_h(const char* tile, int mn0, int lane) {
    if (KMAJOR) return *reinterpret_cast<const bf16x8*>(tile + kmaj32_off(mn0 + (lane & 15), lane >> 4));
    return load_frag<false, 256>(tile, mn0, 0, lane);
}

// bijective XCD remap: workgroups that land on one XCD (bid % 8) get a contiguous range of work ids
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}

}  // namespace obte_gemm_v2
