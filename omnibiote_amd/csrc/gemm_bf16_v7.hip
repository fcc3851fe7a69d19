// bf16 GEMM, structure 7: the 256 x 256 half-tile ring (structure 3's main loop) as a PERSISTENT kernel whose ring never drains.
//
// Replaces the same nn.Linear calls as gemm_bf16_v2.hip (training/model.py:102,151,163,166 forward; their input gradients under
// loss.backward()) on the shapes where a tile's main loop is short: K = 1024 gives a 25-us loop per 256 x 256 tile, and around it
// structure 3 spends (per-workgroup clock stamps) 1.6 us filling the ring from empty, 4-8 us in an epilogue that is
// a chain of latencies — the slowest wave leaves the loop, conversion, staging through the ring's own LDS, a workgroup barrier,
// read back, addressing, stores — and ~3 us between a workgroup's exit and its successor's first MFMA: a quarter of the launch
// with the matrix pipe idle.  Here
//   * one workgroup per CU walks tiles v = blockIdx.x, + gridDim.x, ... (the same tile order and XCD placement as the
//     one-tile-per-workgroup launch: virtual block id v plays blockIdx.x's part);
//   * the LDS-DMA stream is CONTINUOUS across tiles: half-step g + 4 is issued during half-step g whatever tile it belongs to, so
//     the next tile's first four half-stages land under the current tile's last four half-steps and its first fragments are read
//     in the current tile's last half-step — no ring fill, no drain, no relaunch;
//   * the epilogue never touches the ring: each wave sends its 64 x 128 tile through 4 KiB of its own (the 32 KiB of LDS the
//     128-KiB ring leaves free), 16 rows at a time — eight 8-byte writes per lane, four 16-byte reads, four stores of 4 rows x 256 B
//     (the shape the CU's store path takes at full rate; 16 rows x 64 B straight from the accumulators measured 2.5x slower,
//     tools/micro/store_path.hip) — with NO workgroup barrier (a wave's LDS operations execute in order) and nothing to wait for
//     but its own data: the stores drain under the next tile's main loop;
//   * vmcnt retires in issue order, so the ring's counted waits name those stores: the first three half-steps after an epilogue
//     allow its S stores beside the two half-stages in flight (s_waitcnt vmcnt(8 + S)), from the fourth on the ring's vmcnt(8)
//     stands alone again (the stores are then older than everything it leaves in flight).
// Past the last tile of a workgroup the issue cursor points nowhere: its descriptors have zero records (the DMA writes zeros into
// a free slot and touches no memory), which keeps every wait of the loop the same immediate to the end.
// Whole tiles only (M, N multiples of 256; K a multiple of 64, >= 256), no split-K; x W^T and dy W layouts.
#include "gemm_common.h"
#include <type_traits>

using namespace obte_gemm_v2;

namespace {

constexpr int V7_STG_WAVE = 4096;                      // per wave: 16 rows x 256 B of output, 16-byte chunk c of row r at c ^ r
constexpr int V7_SMEM = V3_RING + 8 * V7_STG_WAVE;     // 160 KiB: the whole LDS of a CU

// tile of virtual block id v: the bijective XCD remap and 8-row tile groups of the one-tile-per-workgroup structures
__device__ __forceinline__ void tile_origin(const GemmParams& p, int v, int64_t& m0, int64_t& n0) {
    const int tid_ = xcd_remap(v, p.tiles_m * p.tiles_n);
    const int group_sz = 8 * p.tiles_n;
    const int first_m = (tid_ / group_sz) * 8;
    const int gsz = min(p.tiles_m - first_m, 8);
    m0 = (int64_t)(first_m + (tid_ % group_sz) % gsz) * BM;
    n0 = (int64_t)((tid_ % group_sz) / gsz) * 256;
}

template <bool A_KMAJOR, bool B_KMAJOR, int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_v7_kernel(GemmParams p) {
    constexpr int NJ = 8;
    constexpr int S_ST = EPI == OBTE_EPI_GELU ? 32 : 16;                            // stores of one epilogue per wave
    constexpr int WAIT_LAX = (((8 + S_ST) >> 4) << 14) | 0x0070 | ((8 + S_ST) & 15);   // s_waitcnt vmcnt(8 + S) lgkmcnt(0)
    constexpr bool READS = EPI == OBTE_EPI_ADD || EPI == OBTE_EPI_GELU_BWD || EPI == OBTE_EPI_ADD_DROPOUT || EPI == OBTE_EPI_ROWDOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int ntiles = p.tiles_m * p.tiles_n, G = (int)gridDim.x;
    const int nh = 2 * p.k_per_split;     // half-steps of 32 k per tile: even, >= 8 (host-checked)

    int voff_a[2], voff_b[2];
    dma_offsets_h<A_KMAJOR>(wave, lane, p.lda, voff_a);
    dma_offsets_h<B_KMAJOR>(wave, lane, p.ldb, voff_b);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- the issue cursor: the half-stage the LDS-DMA stream delivers next, four half-steps ahead of the MFMAs ----------------
    // Kept as the two operands' running origins (element offsets into A and B) and stepped by a constant per half-step — the
    // descriptors of a half-step are then a 64-bit add and a subtract each; formed from (m0, n0, k0) every time they cost ~50
    // dependent scalar instructions per half-step, all of them in front of the half-step's barrier.
    int v_i = blockIdx.x, u_i = 0;
    bool live_i = true;
    const int64_t step_a = A_KMAJOR ? 32 : 32 * p.lda, step_b = B_KMAJOR ? 32 : 32 * p.ldb;   // elements per half-step of 32 k
    int64_t ao_i, bo_i;
    auto cursor_tile = [&]() {
        int64_t m0_i, n0_i;
        tile_origin(p, v_i, m0_i, n0_i);
        ao_i = A_KMAJOR ? m0_i * p.lda : m0_i;
        bo_i = B_KMAJOR ? n0_i * p.ldb : n0_i;
    };
    cursor_tile();
    auto cursor_rsrc = [&](i32x4_t& ra, i32x4_t& rb) {
        ra = make_rsrc_words(p.a + ao_i, live_i ? (p.a_elems - ao_i) * 2 : 0);   // past the last tile: zero records, nothing is fetched
        rb = make_rsrc_words(p.b + bo_i, live_i ? (p.b_elems - bo_i) * 2 : 0);
    };
    auto cursor_advance = [&]() {
        ao_i += step_a; bo_i += step_b;
        if (++u_i == nh) {
            u_i = 0;
            v_i += G;
            live_i = v_i < ntiles;
            if (live_i) cursor_tile();
        }
    };

    f32x4 acc[NJ][4];
    bf16x8 a0[4], b0[NJ], a1[4], b1[NJ];
    // half-step with global index gu, fragments F(gu) in (af, bfr): group g = {A fragment g and B fragments 2g, 2g+1 of half-step
    // gu + 1 (ring slot (gu + 1) & 3), LDS-DMA piece g of the cursor's half-stage into slot gu & 3 (free: every wave holds F(gu) in
    // registers since the barrier), the 8 MFMAs of n sub-tiles 2g, 2g+1} — structure 3's interleave, pinned
    auto istep = [&](int gu, const bf16x8 (&af)[4], const bf16x8 (&bfr)[NJ], bf16x8 (&an)[4], bf16x8 (&bn)[NJ]) {
        const char* ta = smem + ((gu + 1) & 3) * H_STAGE;
        const char* tb = ta + H_TILE;
        i32x4_t ra, rb;
        cursor_rsrc(ra, rb);
        const uint32_t lds_st = lds_addr_of(smem + (gu & 3) * H_STAGE) + wave * 1024;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            an[g] = load_frag_h<A_KMAJOR>(ta, wm * 64 + g * 16, lane);
            bn[2 * g] = load_frag_h<B_KMAJOR>(tb, wn * 128 + (2 * g) * 16, lane);
            bn[2 * g + 1] = load_frag_h<B_KMAJOR>(tb, wn * 128 + (2 * g + 1) * 16, lane);
            if (g < 2) lds_dma16(ra, lds_st + 8 * g * 1024, voff_a[g]);
            else lds_dma16(rb, lds_st + H_TILE + 8 * (g - 2) * 1024, voff_b[g - 2]);
#pragma unroll
            for (int ni = 2 * g; ni < 2 * g + 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[ni][mi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        cursor_advance();
    };

    // ---- epilogue of the tile at (m0, n0): accumulators -> this wave's 4 KiB of staging -> whole 256-byte row pieces ----------
    // staged write of quad (ni, mi): row em, 8 bytes at column 16 ni + 4 g4, i.e. chunk 2 ni + (g4 >> 1), half g4 & 1;
    // staged read `it` of a round: row 4 it + g4, chunk lane & 15.  The GELU derivative (read again only in the backward) leaves
    // non-temporally, as in the other structures; outputs beyond the Infinity Cache (the readout) do not come here (eligibility).
    constexpr bool NT = EPI == OBTE_EPI_GELU;
    auto epilogue = [&](const int64_t m0, const int64_t n0) {
        // the lane parts of the epilogue's addresses are formed HERE, from a lane id the compiler cannot see through: hoisted to
        // kernel entry they are live across the whole tile loop, get spilled, and their reloads (vector-memory operations behind a
        // compiler vmcnt(0)) drain the LDS-DMA ring at every tile
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int em = lane_e & 15, g4 = lane_e >> 4;
        char* const stg = smem + V3_RING + wave * V7_STG_WAVE;
        const uint32_t wr_l = (uint32_t)(em * 256 + (g4 & 1) * 8);
        const int64_t tile_o = (m0 + wm * 64) * p.ldd + n0 + wn * 128;          // this wave's 64 x 128 tile
        const uint32_t ldd32 = (uint32_t)p.ldd;
        const uint32_t off_l = (uint32_t)g4 * ldd32 + (uint32_t)em * 8;        // this lane's chunk within a group of 4 rows
        auto o_of = [&](int mi, int it) { return tile_o + (int64_t)(off_l + (uint32_t)(mi * 16 + it * 4) * ldd32); };
        const uint32_t ncol = (uint32_t)(n0 + wn * 128) + (uint32_t)em * 8;    // first of this lane's 8 output columns
        auto row_of = [&](int mi, int it) { return (uint32_t)(m0 + wm * 64) + (uint32_t)(mi * 16 + it * 4 + g4); };
        bf16x8 raux[4][4];
        f32x4 rc[2][4], rs[2][4];
        auto load_aux = [&](int mi) {
            if (READS) {
#pragma unroll
                for (int it = 0; it < 4; ++it) raux[mi][it] = *reinterpret_cast<const bf16x8*>(p.aux + o_of(mi, it));
            } else if (EPI == OBTE_EPI_ROPE_QK) {   // lanes on the v third load the same (valid) table rows and leave their values alone
                const uint32_t T32 = (uint32_t)p.rope_T, hs32 = (uint32_t)p.rope_hs;
                const uint32_t dd = (hs32 & (hs32 - 1)) == 0 ? (ncol & (hs32 - 1)) : (ncol % hs32);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const uint32_t mu = row_of(mi, it);
                    const uint32_t t = (T32 & (T32 - 1)) == 0 ? (mu & (T32 - 1)) : (mu % T32);
                    rc[mi & 1][it] = *reinterpret_cast<const f32x4*>(p.rope_cos + t * (hs32 / 2) + dd / 2);
                    rs[mi & 1][it] = *reinterpret_cast<const f32x4*>(p.rope_sin + t * (hs32 / 2) + dd / 2);
                }
            }
        };
        auto stage_round = [&](int mi, bf16x8 (&st)[4]) {
            // (the eight write addresses are one XOR each from two lane values made opaque per round: kept across the four rounds
            //  they are seven more live registers through the tightest part of the epilogue — the RoPE form spilled them)
            uint32_t wl = wr_l, sw = (uint32_t)((g4 >> 1) ^ em) << 4;
            asm volatile("" : "+v"(wl), "+v"(sw));
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni)   // chunk (2 ni + (g4 >> 1)) ^ em = (2 ni) ^ ((g4 >> 1) ^ em): 2 ni has bit 0 clear
                *reinterpret_cast<bf16x4*>(stg + wl + (sw ^ (uint32_t)(ni << 5))) = __builtin_convertvector((EPI == OBTE_EPI_NONE || EPI == OBTE_EPI_ADD) ? acc[ni][mi] * p.alpha : acc[ni][mi], bf16x4);   // (alpha != 1 only with these two: validate_args)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const uint32_t r = (uint32_t)(4 * it + g4);
                st[it] = *reinterpret_cast<const bf16x8*>(stg + r * 256 + ((((uint32_t)em) ^ r) << 4));
            }
        };
        auto finish_round = [&](int mi, const bf16x8 (&st)[4]) {
            const bool rot = (int64_t)ncol < 2 * (p.N / 3);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                bf16x8 v = st[it];
                const int64_t o = o_of(mi, it);
                if (EPI == OBTE_EPI_GELU) {
                    bf16x8 gact;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        f32x2_t act, der;
                        gelu_ref_both2(f32x2_t{bf2f(v[j]), bf2f(v[j + 1])}, act, der);
                        gact[j] = f2bf(act[0]); gact[j + 1] = f2bf(act[1]);
                        v[j] = f2bf(der[0]); v[j + 1] = f2bf(der[1]);
                    }
#ifdef OBTE_DEBUG_HOOKS
                    if (p.store_rows == 0) asm volatile("" :: "v"(gact)); else
#endif
                    *reinterpret_cast<bf16x8*>(p.d2 + o) = gact;
                } else if (EPI == OBTE_EPI_ADD) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(raux[mi][it][j]) + bf2f(v[j]));
                } else if (EPI == OBTE_EPI_ROWDOT) {
                    // the wave tile's 128 columns are ONE head: the 16 lanes of a row piece hold the row's 128 values of it (common.h
                    // obte_gemm_rowdot_bf16: p.slab = the output, p.rope_T = T, heads of 128 columns)
                    float sdot = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) sdot += bf2f(v[j]) * bf2f(raux[mi][it][j]);
#pragma unroll
                    for (int o_ = 1; o_ < 16; o_ <<= 1) sdot += __shfl_xor(sdot, o_, 64);
                    if (em == 0) {
                        const uint32_t mu = row_of(mi, it), T32 = (uint32_t)p.rope_T;
                        const uint32_t bq = (T32 & (T32 - 1)) == 0 ? mu >> (31 - __builtin_clz(T32)) : mu / T32;   // batch element
                        const uint32_t tq = mu - bq * T32, Hn = (uint32_t)(p.N >> 7), head = ((uint32_t)n0 + (uint32_t)wn * 128u) >> 7;
                        p.slab[((int64_t)bq * Hn + head) * T32 + tq] = sdot;
                    }
                } else if (EPI == OBTE_EPI_GELU_BWD) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * bf2f(raux[mi][it][j]));
                } else if (EPI == OBTE_EPI_ROPE_QK) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xe = bf2f(v[2 * j]), xo = bf2f(v[2 * j + 1]);
                        const bf16 ne = f2bf(xe * rc[mi & 1][it][j] - xo * rs[mi & 1][it][j]), no = f2bf(xe * rs[mi & 1][it][j] + xo * rc[mi & 1][it][j]);
                        v[2 * j] = rot ? ne : v[2 * j];
                        v[2 * j + 1] = rot ? no : v[2 * j + 1];
                    }
                } else if (EPI == OBTE_EPI_ADD_DROPOUT) {
                    const uint32_t rk = drop_rowkey((uint64_t)row_of(mi, it), p.drop);   // dropout element = (row m, column n + j)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const uint32_t bits = drop_pair_bits(rk, (ncol >> 1) + jj);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int j = 2 * jj + e;
                            const float t = drop_keep_bits(bits, (uint32_t)e, p.drop) ? bf2f(f2bf(bf2f(v[j]) * p.drop.scale)) : 0.f;
                            v[j] = f2bf(bf2f(raux[mi][it][j]) + t);
                        }
                    }
                }
#ifdef OBTE_DEBUG_HOOKS
                if (p.store_rows == 0) { asm volatile("" :: "v"(v)); continue; }   // OBTE_GEMM_DEBUG=nostore: timing only
#endif
                if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p.d + o), "v"(v) : "memory");
                else *reinterpret_cast<bf16x8*>(p.d + o) = v;
            }
        };
        // rounds pipelined by one: round mi + 1 is converted, staged and read back before round mi's arithmetic and stores (the
        // wave's own LDS operations execute in order, so the staging can be rewritten as soon as its reads have been ISSUED);
        // the operands an epilogue reads (residual, GELU', RoPE tables) are requested two / one rounds ahead of their use
        bf16x8 sa[4], sb[4];
        load_aux(0);
        if (READS) load_aux(1);
        stage_round(0, sa);
        if (EPI == OBTE_EPI_ROPE_QK) load_aux(1);
        stage_round(1, sb);
        if (READS) load_aux(2);
        finish_round(0, sa);
        if (EPI == OBTE_EPI_ROPE_QK) load_aux(2);
        stage_round(2, sa);
        if (READS) load_aux(3);
        finish_round(1, sb);
        if (EPI == OBTE_EPI_ROPE_QK) load_aux(3);
        stage_round(3, sb);
        finish_round(2, sa);
        finish_round(3, sb);
    };

    // ---- fill the ring once, then walk the tiles -------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        i32x4_t ra, rb;
        cursor_rsrc(ra, rb);
        const uint32_t lds_st = lds_addr_of(smem + j * H_STAGE) + wave * 1024;
        lds_dma16(ra, lds_st, voff_a[0]); lds_dma16(ra, lds_st + 8 * 1024, voff_a[1]);
        lds_dma16(rb, lds_st + H_TILE, voff_b[0]); lds_dma16(rb, lds_st + H_TILE + 8 * 1024, voff_b[1]);
        cursor_advance();
    }
    __builtin_amdgcn_s_waitcnt(0x007C);   // vmcnt(12): half-stage 0 landed
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = load_frag_h<A_KMAJOR>(smem, wm * 64 + i * 16, lane);
#pragma unroll
    for (int i = 0; i < NJ; ++i) b0[i] = load_frag_h<B_KMAJOR>(smem + H_TILE, wn * 128 + i * 16, lane);

    int gu = 0;
    int lax = 0;                          // half-steps left in which the previous tile's stores may still be in flight
#ifdef OBTE_DEBUG_HOOKS
    // OBTE_GEMM_TIMES=1 (debug library): per workgroup, s_memrealtime (100 MHz) at entry / first loop start / exit, and the sums over
    // its tiles of the loop's and the epilogue's time as wave 0 sees them
    unsigned long long t_entry = 0, t_first = 0, t_a = 0, t_b = 0, sum_loop = 0, sum_epi = 0, n_t = 0;
    const bool stamping = p.dbg_times != nullptr && threadIdx.x == 0;
    if (stamping) t_entry = t_first = __builtin_amdgcn_s_memrealtime();
#endif
    for (int v_c = blockIdx.x; v_c < ntiles; v_c += G) {
        int64_t m0, n0;
        tile_origin(p, v_c, m0, n0);
#ifdef OBTE_DEBUG_HOOKS
        if (stamping) { t_a = __builtin_amdgcn_s_memrealtime(); if (n_t == 0) t_first = t_a; }
#endif
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int u = 0; u < nh; u += 2) {
            // half-stage gu + 1 landed (gu + 2, gu + 3 and — right after an epilogue — its stores stay in flight), F(gu) reads complete
            if (lax > 0) __builtin_amdgcn_s_waitcnt(WAIT_LAX); else __builtin_amdgcn_s_waitcnt(0x0078);
            __builtin_amdgcn_s_barrier();
            istep(gu, a0, b0, a1, b1);
            ++gu;
            if (lax > 1) __builtin_amdgcn_s_waitcnt(WAIT_LAX); else __builtin_amdgcn_s_waitcnt(0x0078);
            __builtin_amdgcn_s_barrier();
            istep(gu, a1, b1, a0, b0);
            ++gu;
            lax = lax > 2 ? lax - 2 : 0;
        }
#ifdef OBTE_DEBUG_HOOKS
        if (stamping) { t_b = __builtin_amdgcn_s_memrealtime(); sum_loop += t_b - t_a; }
#endif
        epilogue(m0, n0);
        lax = 3;

#ifdef OBTE_DEBUG_HOOKS
        if (stamping) { sum_epi += __builtin_amdgcn_s_memrealtime() - t_b; ++n_t; }
#endif
    }
#ifdef OBTE_DEBUG_HOOKS
    if (stamping) {
        unsigned long long* o = p.dbg_times + (size_t)blockIdx.x * 8;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        o[0] = t_entry; o[1] = t_first; o[2] = sum_loop; o[3] = sum_epi; o[4] = n_t; o[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the zero-record pieces of the last half-steps write LDS too: none may land after the workgroup has gone)
}

template <bool AK, bool BK, int EPI>
int launch7(const GemmParams& p, hipStream_t st) {
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)gemm_v7_kernel<AK, BK, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, V7_SMEM);
        return true;
    }();
    (void)attr_set;
    const int ntiles = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((gemm_v7_kernel<AK, BK, EPI>), dim3(ntiles < 256 ? ntiles : 256), dim3(NTHREADS), V7_SMEM, st, p);
    OBTE_CHECK_LAUNCH("obte_gemm_bf16");
    return OBTE_OK;
}

}  // namespace

bool obte_gemm_v7_has(bool a_kmajor, bool b_kmajor, int epilogue) {
    if (a_kmajor && b_kmajor)
        return epilogue == OBTE_EPI_NONE || epilogue == OBTE_EPI_GELU || epilogue == OBTE_EPI_ADD || epilogue == OBTE_EPI_ADD_DROPOUT ||
               epilogue == OBTE_EPI_ROPE_QK;
    if (a_kmajor && !b_kmajor) return epilogue == OBTE_EPI_NONE || epilogue == OBTE_EPI_GELU_BWD || epilogue == OBTE_EPI_ROWDOT;
    return false;
}

// whole 256 x 256 tiles, at least one per CU, eight half-steps or more per tile (the ring is refilled four half-steps ahead across tiles)
bool obte_gemm_v7_eligible(const obte_gemm_args* g) {
    return obte_gemm_v7_has(g->a_kmajor != 0, g->b_kmajor != 0, g->epilogue) && g->M % BM == 0 && g->N % 256 == 0 && g->K % BKT == 0 &&
           g->K >= 4 * BKT && (g->M / BM) * (g->N / 256) >= 256 && (g->epilogue != OBTE_EPI_ADD || g->aux != nullptr) &&
           g->ldd < (1ll << 24) &&                        // (32-bit element offsets inside a wave's 64-row tile; the tile origin is 64-bit)
           g->M * g->N * 2 <= (256ll << 20);              // (an output beyond the Infinity Cache wants non-temporal stores: structures 2 / 3)
}

#ifdef OBTE_DEBUG_HOOKS
#include <vector>
static void v7_report(const GemmParams& p, int epi, hipStream_t st) {
    const int n = p.tiles_m * p.tiles_n < 256 ? p.tiles_m * p.tiles_n : 256;
    std::vector<unsigned long long> h((size_t)n * 8);
    if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(h.data(), p.dbg_times, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    unsigned long long t0 = ~0ull, t1 = 0;
    double pro = 0, loop = 0, epi_t = 0, tiles = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned long long* r = &h[(size_t)i * 8];
        if (r[0] < t0) t0 = r[0];
        if (r[5] > t1) t1 = r[5];
        pro += (double)(r[1] - r[0]); loop += (double)r[2]; epi_t += (double)r[3]; tiles += (double)r[4];
    }
    fprintf(stderr, "[gemm v7 epi %d %lldx%lldx%lld, %d workgroups, us] span %.2f | prologue %.2f | per tile: loop %.2f, epilogue %.2f (%.1f tiles per workgroup)\n",
            epi, (long long)p.M, (long long)p.N, (long long)p.K, n, (t1 - t0) * 0.01, pro / n * 0.01, loop / tiles * 0.01, epi_t / tiles * 0.01, tiles / n);
}
#endif

int obte_gemm_v7_launch(const GemmParams& p, bool ak, bool bk, int epi, hipStream_t st) {
#ifdef OBTE_DEBUG_HOOKS
    struct Rep { const GemmParams& p; int epi; hipStream_t st; ~Rep() { if (p.dbg_times) v7_report(p, epi, st); } } rep{p, epi, st};
#endif
    if (ak && bk) {
        switch (epi) {
            case OBTE_EPI_NONE: return launch7<true, true, OBTE_EPI_NONE>(p, st);
            case OBTE_EPI_GELU: return launch7<true, true, OBTE_EPI_GELU>(p, st);
            case OBTE_EPI_ADD: return launch7<true, true, OBTE_EPI_ADD>(p, st);
            case OBTE_EPI_ADD_DROPOUT: return launch7<true, true, OBTE_EPI_ADD_DROPOUT>(p, st);
            case OBTE_EPI_ROPE_QK: return launch7<true, true, OBTE_EPI_ROPE_QK>(p, st);
        }
    } else if (ak && !bk) {
        switch (epi) {
            case OBTE_EPI_NONE: return launch7<true, false, OBTE_EPI_NONE>(p, st);
            case OBTE_EPI_GELU_BWD: return launch7<true, false, OBTE_EPI_GELU_BWD>(p, st);
            case OBTE_EPI_ROWDOT: return launch7<true, false, OBTE_EPI_ROWDOT>(p, st);
        }
    }
    obte_set_error("obte_gemm_bf16: structure 7 has no form for this layout / epilogue (%d %d %d)", (int)ak, (int)bk, epi);
    return OBTE_EINVAL;
}
