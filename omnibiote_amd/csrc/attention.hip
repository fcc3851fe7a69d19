// Fused attention for the OmniBioTE block (training/model.py:115-148), forward and backward, on CDNA4 MFMA
// (v_mfma_f32_32x32x16_bf16).  softmax(q k^T * scale + mask) v, non-causal, scale = 8/n_embd passed in.
//
// Data layout: q, k, v are read in place from the packed c_attn output [B*T, 3C] (row stride 3C, head h at
// column h*D); o is written as [B*T, C], so the reference's transpose(1,2).contiguous() copy never exists.
//
// Common structure of the three kernels: the operand that belongs to the workgroup's own rows stays in
// registers as MFMA B fragments (16 B per lane straight from global memory); the other side streams through
// LDS in 64-row (or 32-row) tiles, XOR-swizzled so that both row reads (ds_read_b128) and hardware-transposed
// reads (ds_read_b64_tr_b16) are bank-conflict free.  Scores are produced TRANSPOSED with the workgroup's own
// row index on the MFMA lane (S^T = K Q^T in forward/dQ, S = Q K^T with the key on the lane in dK/dV), so that
//   - softmax statistics are per lane (one cross-lane exchange with lane^32 per tile), and
//   - the f32 score accumulator, converted to bf16 in registers, already IS the B operand of the next MFMA
//     (O^T += V^T P^T, dQ^T += K^T dS^T, dV^T += dO^T P, dK^T += Q^T dS): no LDS round trip for P or dS.
// Masks: none | per-query key ranges [k_start,k_end) (block-diagonal document masks; KV tiles outside the
// workgroup's union range are skipped) | dense additive bf16 (any strides, stride-0 heads allowed).
// Backward is two kernels without atomics (bitwise reproducible): dQ per query block, dK/dV per key block,
// each recomputing P from the saved log-sum-exp.  The dK/dV kernel, in range mode, uses the symmetry of the
// reference's masks (query t may see key u  <=>  query u may see key t).
#include "attn_common.h"

namespace {
using namespace obte_attn;

// ==========================================================================================================
// forward
// ==========================================================================================================
// Queries per workgroup: 256 (eight waves, one workgroup per CU), with and without dropout.  Every workgroup streams the
// whole K and V of its (batch, head) through LDS, so the L2 -> LDS traffic is inversely proportional to the query block: at
// 128 queries the forward moved 268 MB per launch in 50 us = 5.4 TB/s, the chip-wide LDS-DMA ceiling — that, not MFMA,
// VALU, LDS or DMA latency, was what bounded it.  (The dropout variants ran 128 queries / four waves while every probability
// cost two hash rounds and 64-bit index arithmetic and the kernels needed the 512-register budget; with the row key
// hoisted and one hash per two keys — csrc/common.h — they fit 256 registers: forward 99 -> 58 us, backward 310 -> 189 us
// at p = 0.1, against 47 / 160 us without dropout.)
template <bool DROP> struct FwdShape { static constexpr int NW = 8; static constexpr int STAGES = 2; };   // a deeper ring (4 stages, 3 tiles ahead) measured slower: 48.9 vs 45.9 us
template <int D, int MODE, bool DROP>
__device__ __forceinline__ void attn_fwd_body(const AttnParams& p, char* smem) {
    constexpr int NW = FwdShape<DROP>::NW;
    constexpr int TB = 64 * 2 * D;  // bytes of one 64-row tile
    constexpr int NS = D / 16;      // k-steps over the head dim
    constexpr int ND = D / 32;      // 32-wide d tiles of the output
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const BlockId bid_ = p.q_blk_off ? block_id_rows(p) : block_id((int)((p.T + 32 * NW - 1) / (32 * NW)), p.H);
    if (bid_.blk < 0) return;
    const int hd = bid_.hd;
    const int64_t b = bid_.b;
    const int T = (int)p.T;
    const int C = p.H * D;
    const int64_t ld = 3 * (int64_t)C;
    const QSide qsd = q_side(p, b, hd);          // the queries of this batch element (attn_common.h): all T rows, or a gathered set
    const int Tq = qsd.n;
    if (bid_.blk * (32 * NW) >= Tq) return;      // (a gathered set shorter than the grid allows for: nothing to do, before any barrier)
    const int q_row = bid_.blk * (32 * NW) + wave * 32 + (lane & 31);
    const bool q_ok = q_row < Tq;
    const int q_c = q_ok ? q_row : Tq - 1;

    // the key range of this query FIRST: the tile range (and with it the first LDS-DMA) depends on it, and vmcnt retires in
    // issue order — loaded behind the Q fragments, it made the K/V stream wait for all of them
    int r_lo = 0, r_hi = T;
    if (MODE == MASK_RANGES || (MODE == MASK_DENSE && p.key_ranges)) {
        r_lo = p.key_ranges[(qsd.row0 + q_c) * 2];
        r_hi = p.key_ranges[(qsd.row0 + q_c) * 2 + 1];
    }
    const bf16* qptr = p.q_src + (qsd.row0 + q_c) * p.q_ld + hd * D;
    bf16x8 qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qptr + 16 * s + 8 * h);
    // dropout row = (batch, head, query): its key is formed once per lane, not once per probability
    // (the mask's row is the query's POSITION in its sequence: a gathered query keeps the mask it has in the whole attention)
    const uint32_t drop_rk = DROP ? drop_rowkey(((uint64_t)b * p.H + hd) * (uint64_t)T + (uint64_t)(p.q_pos ? p.q_pos[qsd.row0 + q_c] : q_c), p.drop) : 0u;

    int ks = 0, ke = T;
    if (MODE == MASK_RANGES) {
        ks = max(r_lo, 0);
        ke = min(r_hi, T);
    }
    int lo = ks, hi = ke;
    if (MODE == MASK_DENSE && p.key_ranges) {   // dense arithmetic, but the loop may skip what every row masks out
        lo = max(r_lo, 0);
        hi = min(r_hi, T);
        block_minmax<NW>(lo, hi, reinterpret_cast<int*>(smem + 2 * FwdShape<DROP>::STAGES * TB), wave, lane);
    }
    if (MODE == MASK_RANGES) block_minmax<NW>(lo, hi, reinterpret_cast<int*>(smem + 2 * FwdShape<DROP>::STAGES * TB), wave, lane);
    const int t_begin = lo / 64;
    int t_end = hi > lo ? (hi + 63) / 64 : t_begin;
    if (p.max_tiles) t_end = min(t_end, t_begin + p.max_tiles - 1);

    const bf16* kbase = p.qkv + b * T * ld + C + hd * D;
    const bf16* vbase = kbase + C;
    const bf16* mrow = nullptr;
    if (MODE == MASK_DENSE) mrow = p.mask + b * p.mask_sb + hd * p.mask_sh + (int64_t)q_c * p.mask_sq;
    const bool mvec = MODE == MASK_DENSE && mask_vec_ok(p);

    float m = -INFINITY, l = 0.f;
    f32x16 o[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    const float scale2 = p.scale * LOG2E;

    // K/V tiles by LDS-DMA (asm issue, see common.h): no staging registers, no ds_write — the forward is issue-bound
    TileDma<D, 64, NW> dma;
    dma.init(wave, lane, ld);
    // ring of R stages (K tile + V tile each), loaded R-1 tiles ahead; one workgroup per CU, so there is no second
    // workgroup to cover a wait and the prefetch distance has to
    constexpr int R = FwdShape<DROP>::STAGES;
    constexpr int OPS = 2 * TileDma<D, 64, NW>::NP;   // LDS-DMA instructions per wave per stage
    auto issue_kv = [&](int t) {
        const int64_t row0 = (int64_t)t * 64;
        const int64_t rows_left = ((int64_t)T - row0) * ld;
        char* st = smem + ((t - t_begin) % R) * 2 * TB;
        dma.issue(kbase + row0 * ld, (rows_left - (C + hd * D)) * 2, st, wave);
        dma.issue(vbase + row0 * ld, (rows_left - (2 * C + hd * D)) * 2, st + TB, wave);
    };
    // the first tile(s) go out while the Q fragments are still in flight; the builtin wait that follows covers both (hipcc
    // must KNOW the Q fragments have landed, or it keeps per-fragment vmcnt waits in front of the first MFMAs of every tile —
    // and vmcnt counts the asm-issued LDS-DMA too)
    static_assert(R == 2, "prologue written for a two-stage ring (one tile ahead)");
    if (t_begin < t_end) issue_kv(t_begin);
    prologue_wait_all();
    __syncthreads();

    for (int t = t_begin; t < t_end; ++t) {
        if (t + R - 1 < t_end) issue_kv(t + R - 1);   // the slot of tile t-1: every wave is past the barrier that ended it
        const char* Kt = smem + ((t - t_begin) % R) * 2 * TB;
        const char* Vt = Kt + TB;
        const int key0 = t * 64;

        // The softmax arithmetic is what bounds this kernel (measured: ~11 VALU instructions per MFMA), so it is kept
        // minimal: scores start from an inline-zero accumulator (no register clears), the scale is folded into the
        // exponent's FMA (max is taken on raw scores), and the running maximum is only moved — with the rescale of O
        // that goes with it — when some row's maximum grew by more than 2^6 (exact in exact arithmetic; P, l and O are
        // simply carried at up to 64x their usual magnitude in fp32 / bf16, whose precision is relative).
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 sc[2];
        {
            // every K fragment of the tile is requested before the first MFMA (and the order is pinned): left to itself hipcc
            // sinks each ds_read to just before the MFMA that uses it — read, lgkmcnt(0), MFMA, sixteen times over, the LDS
            // latency exposed every time (the forward has the registers for this since the subtile LDS image: 156 -> ~220)
            bf16x8 kfr[2][NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                kfr[0][s] = row_frag<D>(Kt, 0, s, lane);
                kfr[1][s] = row_frag<D>(Kt, 32, s, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            sc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0][0], qf[0], zero16, 0, 0, 0);
            sc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[1][0], qf[0], zero16, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < NS; ++s) {
                sc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0][s], qf[s], sc[0], 0, 0, 0);
                sc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[1][s], qf[s], sc[1], 0, 0, 0);
            }
        }
        if (MODE == MASK_DENSE) {   // scores in the log2 domain, plus the additive mask
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float mk[4];
                    mask4(mrow, key0 + 32 * mt + 8 * i + 4 * h, T, mvec, mk);
#pragma unroll
                    for (int j = 0; j < 4; ++j) sc[mt][4 * i + j] = sc[mt][4 * i + j] * scale2 + mk[j];
                }
        }
        const bool inside = key0 >= ks && key0 + 64 <= ke;
        if (!__all(inside)) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + 32 * mt + acc_row(r, h);
                    if (key < ks || key >= ke) sc[mt][r] = -INFINITY;
                }
        }
        float mx = sc[0][0];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[mt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float sx = MODE == MASK_DENSE ? 1.0f : scale2;   // what turns a stored score into log2 units
        const float m_new = fmaxf(m, mx * sx);
        float m_safe = m;
        if (__any(m == -INFINITY || m_new - m > 6.0f)) {
            m_safe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m - m_safe);
            l *= alpha;
            m = m_new;
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        float rs = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            // dropout, optional second output: the keep decisions of this wave's 32 queries for the 32 keys of this half tile, as one
            // word per KEY (bit = query): the compare's own lane mask is that word — lanes 0..31 hold key 8i + j, lanes 32..63 key
            // 8i + j + 4 — and v_writelane gathers the 32 words into lanes 0..31 for one 128-byte store.  The key-major backward
            // kernel then reads one word per key and slice instead of hashing every (query, key) pair again.
            uint32_t keyword = 0u;
            unsigned long long km[16];   // the sixteen compares' lane masks of this half tile (SGPR pairs)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // registers 4i .. 4i+3 hold four consecutive keys (a multiple of 4 onwards): two pairs, one hash each
                uint32_t bits[2] = {0u, 0u};
                if (DROP) {
                    const uint32_t g = (uint32_t)(key0 + 32 * mt + 8 * i + 4 * h) >> 1;
                    bits[0] = drop_pair_bits(drop_rk, g);
                    bits[1] = drop_pair_bits(drop_rk, g + 1);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * i + j;
                    float pv = fast_exp2(__builtin_fmaf(sc[mt][r], sx, -m_safe));
                    rs += pv;   // the normaliser uses the un-dropped probabilities
                    if (DROP) {
                        const bool kp = drop_keep_bits(bits[j >> 1], (uint32_t)(j & 1), p.drop);
                        pv = kp ? pv * p.drop.scale : 0.f;
                        km[r] = __ballot(kp);
                    }
                    sc[mt][r] = pv;
                }
            }
            if (DROP && p.drop_bits_out) {
                // lane 8i + j of `keyword` <- low word of mask (i, j), lane 8i + j + 4 <- its high word: two asm blocks of 16
                // v_writelane (this hipcc has no writelane builtin).  s_nop 3 in front: a v_writelane that reads an SGPR a
                // vector compare has just written needs four wait states, and the hazard recogniser does not look into asm —
                // without it the low word was the PREVIOUS compare's mask.
#define OBTE_WL(M, L) "v_writelane_b32 %0, %" #M ", " #L "\n\t"
                asm volatile("s_nop 3\n\t" OBTE_WL(1, 0) OBTE_WL(2, 4) OBTE_WL(3, 1) OBTE_WL(4, 5) OBTE_WL(5, 2) OBTE_WL(6, 6) OBTE_WL(7, 3) OBTE_WL(8, 7)
                             OBTE_WL(9, 8) OBTE_WL(10, 12) OBTE_WL(11, 9) OBTE_WL(12, 13) OBTE_WL(13, 10) OBTE_WL(14, 14) OBTE_WL(15, 11) OBTE_WL(16, 15)
                             : "+v"(keyword)
                             : "s"((uint32_t)km[0]), "s"((uint32_t)(km[0] >> 32)), "s"((uint32_t)km[1]), "s"((uint32_t)(km[1] >> 32)),
                               "s"((uint32_t)km[2]), "s"((uint32_t)(km[2] >> 32)), "s"((uint32_t)km[3]), "s"((uint32_t)(km[3] >> 32)),
                               "s"((uint32_t)km[4]), "s"((uint32_t)(km[4] >> 32)), "s"((uint32_t)km[5]), "s"((uint32_t)(km[5] >> 32)),
                               "s"((uint32_t)km[6]), "s"((uint32_t)(km[6] >> 32)), "s"((uint32_t)km[7]), "s"((uint32_t)(km[7] >> 32)));
                asm volatile("s_nop 3\n\t" OBTE_WL(1, 16) OBTE_WL(2, 20) OBTE_WL(3, 17) OBTE_WL(4, 21) OBTE_WL(5, 18) OBTE_WL(6, 22) OBTE_WL(7, 19) OBTE_WL(8, 23)
                             OBTE_WL(9, 24) OBTE_WL(10, 28) OBTE_WL(11, 25) OBTE_WL(12, 29) OBTE_WL(13, 26) OBTE_WL(14, 30) OBTE_WL(15, 27) OBTE_WL(16, 31)
                             : "+v"(keyword)
                             : "s"((uint32_t)km[8]), "s"((uint32_t)(km[8] >> 32)), "s"((uint32_t)km[9]), "s"((uint32_t)(km[9] >> 32)),
                               "s"((uint32_t)km[10]), "s"((uint32_t)(km[10] >> 32)), "s"((uint32_t)km[11]), "s"((uint32_t)(km[11] >> 32)),
                               "s"((uint32_t)km[12]), "s"((uint32_t)(km[12] >> 32)), "s"((uint32_t)km[13]), "s"((uint32_t)(km[13] >> 32)),
                               "s"((uint32_t)km[14]), "s"((uint32_t)(km[14] >> 32)), "s"((uint32_t)km[15]), "s"((uint32_t)(km[15] >> 32)));
#undef OBTE_WL
            }
            if (DROP && p.drop_bits_out && lane < 32 && q_row - (lane & 31) < T) {
                const int key = key0 + 32 * mt + lane;
                const int64_t nsl = ((int64_t)T + 31) / 32;
                if (key < T) p.drop_bits_out[(((int64_t)b * p.H + hd) * nsl + (q_row >> 5)) * T + key] = keyword;
            }
        }
        l += rs;
        // O^T += V^T P^T : P^T accumulators are the B operand as they stand.  The V fragments of key step kk+1 are requested
        // before the MFMAs of step kk (two register sets, order pinned).
        {
            bf16x8 vfr[2][ND];
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) vfr[0][dt] = tr_frag<D>(Vt, 0, dt, lane);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 pf = pack8(sc[kk >> 1], 8 * (kk & 1));
                if (kk + 1 < 4) {
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) vfr[(kk + 1) & 1][dt] = tr_frag<D>(Vt, 16 * (kk + 1), dt, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < ND; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[kk & 1][dt], pf, o[dt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!p.no_wait) dma_wait_leave(OPS * min(max(t_end - t - 2, 0), R - 2));   // tile t+1 landed; later ones may stay in flight
        __syncthreads();
    }

    const float l_tot = l + __shfl_xor(l, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (q_ok) {
        // A row whose every key carries a -1e9-class additive mask: the forward above reproduces the reference (the scores
        // vanish in fp32 beside the mask, the softmax is uniform), but m + log2(l) cannot hold log2(l) beside |m| ~ 1e9,
        // so an lse-based backward would turn that row into garbage for every key.  Its lse is stored as +inf instead:
        // the backward then gives the row exactly zero weight.  (The reference's own mask builder never produces such a
        // row; torch's fused SDPA backends return NaN for it.)
        const bool degenerate = m < -1.0e8f;
        if (h == 0) p.lse[qsd.stat0 + q_row] = (l_tot > 0.f && !degenerate) ? (m + __log2f(l_tot)) * LN2 : INFINITY;
    }
    {   // every tile has been read (the loop ends on a barrier): the stage memory carries the output rows out (wave_rows_out)
        bf16x4 ob[ND * 4];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) ob[4 * dt + i][j] = f2bf(o[dt][4 * i + j] * inv);
        const int64_t row0 = (int64_t)q_row - (lane & 31);   // the wave's first query
        wave_rows_out<D>(smem + wave * (32 * 2 * D), ob, p.o + (qsd.row0 + row0) * C + hd * D, C, (int)min((int64_t)32, (int64_t)Tq - row0), lane);
    }
}

// ==========================================================================================================
// backward, part 1: dQ (and delta).  Same shape as forward: 256 queries per workgroup, query on the lane.
//   S^T = K Q^T ; P^T = exp(S^T*scale + mask - lse) ; dP^T = V dO^T ; dS^T = P^T (dP^T - delta) ; dQ^T += K^T dS^T
// ==========================================================================================================
// 256 queries (eight waves) per workgroup, for the same reason as the forward: K/V traffic per query.
template <int D, int MODE, bool DROP>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnParams& p, char* smem) {
    constexpr int NW = FwdShape<DROP>::NW;
    constexpr int TB = 64 * 2 * D;
    constexpr int NS = D / 16, ND = D / 32;
    OBTE_STAMP(p, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const BlockId bid_ = p.q_blk_off ? block_id_rows(p) : block_id((int)((p.T + 32 * NW - 1) / (32 * NW)), p.H);
    if (bid_.blk < 0) return;
    const int hd = bid_.hd;
    const int64_t b = bid_.b;
    const int T = (int)p.T;
    const int C = p.H * D;
    const int64_t ld = 3 * (int64_t)C;
    const QSide qsd = q_side(p, b, hd);          // (see the forward)
    const int Tq = qsd.n;
    if (bid_.blk * (32 * NW) >= Tq) return;
    const int q_row = bid_.blk * (32 * NW) + wave * 32 + (lane & 31);
    const bool q_ok = q_row < Tq;
    const int q_c = q_ok ? q_row : Tq - 1;

    int r_lo = 0, r_hi = T;   // the key range first (see the forward)
    if (MODE == MASK_RANGES || (MODE == MASK_DENSE && p.key_ranges)) {
        r_lo = p.key_ranges[(qsd.row0 + q_c) * 2];
        r_hi = p.key_ranges[(qsd.row0 + q_c) * 2 + 1];
    }
    const bf16* qptr = p.q_src + (qsd.row0 + q_c) * p.q_ld + hd * D;
    const bf16* doptr = p.d_o + (qsd.row0 + q_c) * C + hd * D;
    bf16x8 qf[NS], dof[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        qf[s] = *reinterpret_cast<const bf16x8*>(qptr + 16 * s + 8 * h);
        dof[s] = *reinterpret_cast<const bf16x8*>(doptr + 16 * s + 8 * h);
    }
    const float lse2 = p.lse_in[qsd.stat0 + q_c] * LOG2E;
    // (the mask's row is the query's POSITION in its sequence: a gathered query keeps the mask it has in the whole attention)
    const uint32_t drop_rk = DROP ? drop_rowkey(((uint64_t)b * p.H + hd) * (uint64_t)T + (uint64_t)(p.q_pos ? p.q_pos[qsd.row0 + q_c] : q_c), p.drop) : 0u;
    const bf16* optr = p.o_in + (qsd.row0 + q_c) * C + hd * D;
    bf16x8 of[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) of[s] = *reinterpret_cast<const bf16x8*>(optr + 16 * s + 8 * h);

    // The key range was requested first, so it lands first (vmcnt retires in issue order): the tile range and the first K/V
    // tile's LDS-DMA go out while the 24 row-fragment loads above are still travelling, instead of one latency behind them.
    int ks = 0, ke = T;
    if (MODE == MASK_RANGES) {
        ks = max(r_lo, 0);
        ke = min(r_hi, T);
    }
    int lo = ks, hi = ke;
    if (MODE == MASK_DENSE && p.key_ranges) {
        lo = max(r_lo, 0);
        hi = min(r_hi, T);
        block_minmax<NW>(lo, hi, reinterpret_cast<int*>(smem + 4 * TB), wave, lane);
    }
    if (MODE == MASK_RANGES) block_minmax<NW>(lo, hi, reinterpret_cast<int*>(smem + 4 * TB), wave, lane);
    const int t_begin = lo / 64;
    int t_end = hi > lo ? (hi + 63) / 64 : t_begin;
    if (p.max_tiles) t_end = min(t_end, t_begin + p.max_tiles - 1);

    const bf16* kbase = p.qkv + b * T * ld + C + hd * D;
    const bf16* vbase = kbase + C;
    const bf16* mrow = nullptr;
    if (MODE == MASK_DENSE) mrow = p.mask + b * p.mask_sb + hd * p.mask_sh + (int64_t)q_c * p.mask_sq;
    const bool mvec = MODE == MASK_DENSE && mask_vec_ok(p);

    TileDma<D, 64, NW> dma;
    dma.init(wave, lane, ld);
    auto issue_kv = [&](int t, int stage) {
        const int64_t row0 = (int64_t)t * 64;
        const int64_t rows_left = ((int64_t)T - row0) * ld;
        char* st = smem + stage * 2 * TB;
        dma.issue(kbase + row0 * ld, (rows_left - (C + hd * D)) * 2, st, wave);
        dma.issue(vbase + row0 * ld, (rows_left - (2 * C + hd * D)) * 2, st + TB, wave);
    };
    if (t_begin < t_end) issue_kv(t_begin, 0);

    // delta = rowsum(O * dO), computed here from the dO fragments already in registers (it used to be a kernel of its
    // own: one 9-us launch per layer) and published for the dK/dV kernel, which runs after this one on the same stream
    float dl = 0.f;
    {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += bf2f(of[s][j]) * bf2f(dof[s][j]);
        dl += __shfl_xor(dl, 32, 64);
        if (h == 0 && q_ok) p.delta[qsd.stat0 + q_row] = dl;
    }

    f32x16 dq[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
    const float scale2 = p.scale * LOG2E;

    dma_wait_all();
    prologue_wait_all();
    __syncthreads();
    OBTE_STAMP(p, 1);

    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        if (t + 1 < t_end) issue_kv(t + 1, cur ^ 1);
        const char* Kt = smem + cur * 2 * TB;
        const char* Vt = Kt + TB;
        const int key0 = t * 64;
        // the 64-key tile is processed as two 32-key halves so that only one score/dP accumulator pair is live.
        // (Tried, measured no change — 173-176 us for the whole backward either way: issuing the K / V fragment reads two
        // k-steps ahead of their MFMAs and the transposed K fragments under the softmax arithmetic, pinned with
        // sched_barrier.  hipcc's own order is read -> wait -> MFMA per fragment, but the SIMD's second wave covers it.)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x16 sc, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Kt, 32 * mt, s, lane), qf[s], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Vt, 32 * mt, s, lane), dof[s], dp, 0, 0, 0);
            }
            // 32 keys that every query of the wave may see (the inside of a document): no per-element range test
            auto ds_rows = [&](auto checked) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float mk[4] = {0.f, 0.f, 0.f, 0.f};
                    if (MODE == MASK_DENSE) mask4(mrow, key0 + 32 * mt + 8 * i + 4 * h, T, mvec, mk);
                    uint32_t bits[2] = {0u, 0u};
                    if (DROP) {
                        const uint32_t g = (uint32_t)(key0 + 32 * mt + 8 * i + 4 * h) >> 1;
                        bits[0] = drop_pair_bits(drop_rk, g);
                        bits[1] = drop_pair_bits(drop_rk, g + 1);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * i + j;
                        const int key = key0 + 32 * mt + acc_row(r, h);
                        float x = sc[r] * scale2 - lse2;
                        if (MODE == MASK_DENSE) x += mk[j];
                        float pv = fast_exp2(x);
                        if (decltype(checked)::value && (key < ks || key >= ke)) pv = 0.f;
                        float dpd = dp[r];
                        if (DROP) dpd = drop_keep_bits(bits[j >> 1], (uint32_t)(j & 1), p.drop) ? dpd * p.drop.scale : 0.f;
                        sc[r] = pv * (dpd - dl);   // dS^T (without the scale factor)
                    }
                }
            };
            if (MODE != MASK_DENSE && __all(key0 + 32 * mt >= ks && key0 + 32 * mt + 32 <= ke)) ds_rows(std::false_type{});
            else ds_rows(std::true_type{});
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 dsf = pack8(sc, 8 * kk);
#pragma unroll
                for (int dt = 0; dt < ND; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<D>(Kt, 32 * mt + 16 * kk, dt, lane), dsf, dq[dt], 0, 0, 0);
            }
        }
        if (!p.no_wait) dma_wait_all();
        __syncthreads();
    }

    OBTE_STAMP(p, 2);
    {   // rotation table of the row first (one latency, nothing stored yet), then the rows leave through the stage memory
        RopeRow<D> rr;
        rr.load(p.rope_cos, p.rope_sin, p.q_pos ? p.q_pos[qsd.row0 + q_c] : q_c, h);   // (a gathered query rotates back at its own position)
        bf16x4 gb[ND * 4];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float g[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = dq[dt][4 * i + j] * p.scale;
                rr.apply(g, dt, i);
#pragma unroll
                for (int j = 0; j < 4; ++j) gb[4 * dt + i][j] = f2bf(g[j]);
            }
        const int64_t row0 = (int64_t)q_row - (lane & 31);   // the wave's first query
        wave_rows_out<D>(smem + wave * (32 * 2 * D), gb, p.dq_dst + (qsd.row0 + row0) * p.dq_ld + hd * D, p.dq_ld, (int)min((int64_t)32, (int64_t)Tq - row0), lane);
    }
#ifdef OBTE_DEBUG_HOOKS
    OBTE_STAMP(p, 3);
    if (p.dbg_times) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); OBTE_STAMP(p, 4); }
#endif
}

// ==========================================================================================================
// backward, part 2: dK and dV.  256 keys per workgroup (32 per wave), key on the lane; queries stream through
// LDS in 32-row tiles (Q and dO, plus their lse/delta).
//   S = Q K^T ; P = exp(S*scale + mask - lse) ; dP = dO V^T ; dS = P (dP - delta)
//   dV^T += dO^T P ; dK^T += Q^T dS
// ==========================================================================================================
// 256 keys (eight waves) per workgroup: the Q / dO tiles every workgroup streams are then shared by
// twice as many keys (half the L2 -> LDS traffic per key).
// Measured and dropped this round (A/B on one box, T = 1024, B = H = 8, key ranges): running the second half of the waves one
// phase late — every wave executes C(t-1) | A(t) B(t) per iteration, the first half places the step boundary between C and A,
// the second after B, so that one wave's softmax arithmetic always has its SIMD partner's MFMAs beside it
// (MI355X_MICROARCH.md "Two waves per SIMD" item 9; three Q/dO slots, P / dS carried over the loop edge, bitwise the same
// results) took 95.3 us against 89.5 un-staggered (333 vs 316 us at T = 4096): the loop is paced by fragment reads and their
// waits (320 KiB of LDS reads per step per CU, 40 KiB per wave), not by the vector pipe, and de-phasing the halves puts both
// LDS-heavy phases (A: 24 x b128, C: 32 x tr_b64) side by side.  Also measured and dropped: 64-row Q/dO stages (two MFMA tiles
// per barrier, half the barriers): 86.9 vs 87.0 us, 314.7 vs 312.5 at T = 4096 — the barrier count is not what parks the waves.
template <int D, int MODE, bool DROP, bool BITS = false>   // BITS: dropout decisions from the forward's keep bits (p.drop_bits_in), no hash
__device__ __forceinline__ void attn_bwd_dkdv_body(const AttnParams& p, char* smem) {
    constexpr int NW = FwdShape<DROP>::NW;
    constexpr int QB = 32 * 2 * D;  // bytes of one 32-row tile
    constexpr int NS = D / 16, ND = D / 32;
    constexpr int STAGE = 2 * QB + 384;  // Q tile, dO tile, 32 lse2 + 32 delta floats + 32 dropout row keys
    constexpr int VB = 32 * NW * 2 * D;  // the workgroup's own V rows, kept in LDS for the whole kernel (B operand of dP)
    OBTE_STAMP(p, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const BlockId bid_ = block_id((int)((p.T + 32 * NW - 1) / (32 * NW)), p.H);
    const int hd = bid_.hd;
    const int64_t b = bid_.b;
    const int T = (int)p.T;
    const int C = p.H * D;
    const int64_t ld = 3 * (int64_t)C;
    const int key = bid_.blk * (32 * NW) + wave * 32 + (lane & 31);
    const bool k_ok = key < T;
    const int key_c = k_ok ? key : T - 1;
    const QSide qsd = q_side(p, b, hd);   // the queries of this batch element: all T rows, or a gathered set (attn_common.h); query ranges
    const int Tq = qsd.n;                 // of the keys (query_bounds — required with a gathered set and a range mask) are in its indices

    int r_lo = 0, r_hi = Tq;   // the query range of this key first (see the forward)
    if (MODE == MASK_RANGES) {
        const int32_t* src = p.query_bounds ? p.query_bounds : p.key_ranges;   // no per-key table: symmetric mask
        r_lo = src[(b * T + key_c) * 2];
        r_hi = src[(b * T + key_c) * 2 + 1];
    } else if (MODE == MASK_DENSE && p.query_bounds) {
        r_lo = p.query_bounds[(b * T + key_c) * 2];
        r_hi = p.query_bounds[(b * T + key_c) * 2 + 1];
    }
    const bf16* kptr = p.qkv + (b * T + key_c) * ld + C + hd * D;
    bf16x8 kf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) kf[s] = *reinterpret_cast<const bf16x8*>(kptr + 16 * s + 8 * h);
    char* Vblk = smem + 2 * STAGE;
    {
        TileDma<D, 32 * NW, NW> dmv;
        dmv.init(wave, lane, ld);
        const int64_t row0 = (int64_t)bid_.blk * (32 * NW);
        dmv.issue(p.qkv + (b * T + row0) * ld + 2 * C + hd * D, (((int64_t)T - row0) * ld - (2 * C + hd * D)) * 2, Vblk, wave);
    }
    // by symmetry of the mask, the queries that see this key are the keys this position sees as a query
    int qs = 0, qe = Tq;
    if (MODE == MASK_RANGES) {
        qs = max(r_lo, 0);
        qe = min(r_hi, Tq);
    }
    if (!k_ok) { qs = 0; qe = 0; }
    int lo = k_ok ? qs : Tq, hi = k_ok ? qe : 0;
    if (MODE == MASK_RANGES) {
        block_minmax<NW>(lo, hi, reinterpret_cast<int*>(smem + 2 * STAGE + VB), wave, lane);
    } else if (MODE == MASK_DENSE && p.query_bounds) {
        if (k_ok) {
            lo = max(r_lo, 0);
            hi = min(r_hi, T);
        }
        block_minmax<NW>(lo, hi, reinterpret_cast<int*>(smem + 2 * STAGE + VB), wave, lane);
    } else {
        lo = 0; hi = Tq;
    }
    const int t_begin = lo / 32;
    int t_end = hi > lo ? (hi + 31) / 32 : t_begin;
    if (p.max_tiles) t_end = min(t_end, t_begin + p.max_tiles - 1);

    const int64_t qld = p.q_ld;
    const bf16* qbase = p.q_src + qsd.row0 * qld + hd * D;
    const bf16* dobase = p.d_o + qsd.row0 * C + hd * D;
    const float* lse_b = p.lse_in + qsd.stat0;
    const float* del_b = p.delta + qsd.stat0;
    const bf16* mcol = nullptr;
    if (MODE == MASK_DENSE) mcol = p.mask + b * p.mask_sb + hd * p.mask_sh + key_c;

    f32x16 dk[ND], dv[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
    const float scale2 = p.scale * LOG2E;

    TileDma<D, 32, NW> dmq, dmd;
    dmq.init(wave, lane, qld);
    dmd.init(wave, lane, C);
    auto issue_qd = [&](int t, char* stage) {
        const int64_t row0 = (int64_t)t * 32;
        dmq.issue(qbase + row0 * qld, (((int64_t)Tq - row0) * qld - hd * D) * 2, stage, wave);
        dmd.issue(dobase + row0 * C, (((int64_t)Tq - row0) * C - hd * D) * 2, stage + QB, wave);
    };
    float st_l = 0.f;  // threads 0..31: lse2 of row tid ; 32..63: delta of row tid-32 ; 64..95 (dropout): row key of row tid-64
    auto load_stats = [&](int q0) {
        if (tid < 64) {
            const int q = q0 + (tid & 31);
            float v = 0.f;
            if (q < Tq) v = tid < 32 ? lse_b[q] * LOG2E : del_b[q];
            else if (tid < 32) v = INFINITY;   // rows past the last query: p = exp2(x - inf) = 0
            st_l = v;
        } else if (DROP && tid < 96) {
            const int q = min(q0 + (tid & 31), Tq - 1);
            st_l = __uint_as_float(drop_rowkey(((uint64_t)b * p.H + hd) * (uint64_t)T + (uint64_t)(p.q_pos ? p.q_pos[qsd.row0 + q] : q), p.drop));
        }
    };
    auto store_stats = [&](char* stage) {
        if (tid < (DROP ? 96 : 64)) reinterpret_cast<float*>(stage + 2 * QB)[tid] = st_l;
    };
    if (t_begin < t_end) {
        issue_qd(t_begin, smem);
        load_stats(t_begin * 32);
        store_stats(smem);
    }
    dma_wait_all();
    prologue_wait_all();
    __syncthreads();
    OBTE_STAMP(p, 1);

    // dropout with the forward's keep bits (one word per key and 32-query slice): this lane's word of the current slice, the next
    // one requested a slice ahead and taken over beside the end-of-iteration wait (like the row constants)
    const uint32_t* keyw_src = (DROP && BITS) ? p.drop_bits_in + ((b * p.H + hd) * (((int64_t)T + 31) / 32)) * T + key_c : nullptr;
    uint32_t keyw = 0u, keyw_next = 0u;
    if (keyw_src && t_begin < t_end) keyw = keyw_src[(int64_t)t_begin * T];
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        const bool more = t + 1 < t_end;
        if (more) {
            issue_qd(t + 1, smem + (cur ^ 1) * STAGE);
            load_stats((t + 1) * 32);
            if (keyw_src) keyw_next = keyw_src[(int64_t)(t + 1) * T];
        }
        const char* Qt = smem + cur * STAGE;
        const char* Dt = Qt + QB;
        const float* stats = reinterpret_cast<const float*>(Qt + 2 * QB);
        const int q0 = t * 32;

        f32x16 sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#ifdef OBTE_DEBUG_HOOKS
        if (!(p.dbg_skip & 4))
#endif
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Qt, 0, s, lane), kf[s], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Dt, 0, s, lane), row_frag<D>(Vblk, 32 * wave, s, lane), dp, 0, 0, 0);
        }
#ifdef OBTE_DEBUG_HOOKS
        if (!(p.dbg_skip & 1))
#endif
        {
            // 32 queries that every key of the wave is seen by (the inside of a document): no per-element range test
            auto p_ds_rows = [&](auto checked) {
                // Dropout: the 32 bits that decide (query, key pair) are shared by the two lanes that hold the pair's keys, and here
                // every lane is ONE key against 16 queries — so the even lane of a pair hashes the queries of register groups
                // i = 0, 1, the odd lane those of i = 2, 3, and each takes the other half over by a quad permute: 8 hashes per lane
                // and slice instead of 16 (this kernel paid twice what the query-major kernels pay per element).
                uint32_t pairw[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
                const uint32_t pair_g = (uint32_t)key_c >> 1;
                const int odd_lane = lane & 1;
                // (the forward's keep bits, if it left them: one word per key and slice, bit = query — no hash at all)
                constexpr bool have_bits = DROP && BITS;
                const uint32_t kw = have_bits ? (keyw >> (4 * h)) : 0u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(stats + 8 * i + 4 * h);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(stats + 32 + 8 * i + 4 * h);
                    uint32_t w4[4] = {0u, 0u, 0u, 0u};
                    if (DROP && !have_bits) {
                        if (i < 2) {   // this lane's share: group i (even lane) or i + 2 (odd lane)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                pairw[i][j] = drop_pair_bits(__float_as_uint(stats[64 + 8 * (i + 2 * odd_lane) + 4 * h + j]), pair_g);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j)   // quad_perm [0,0,2,2]: from the even lane; [1,1,3,3]: from the odd one
                            w4[j] = i < 2 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pairw[i][j], 0xA0, 0xf, 0xf, false)
                                          : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pairw[i - 2][j], 0xF5, 0xf, 0xf, false);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * i + j;
                        const int q = q0 + 8 * i + 4 * h + j;
                        float x = sc[r] * scale2 - l4[j];
                        if (MODE == MASK_DENSE) { if (q < T) x += bf2f(mcol[(int64_t)q * p.mask_sq]) * LOG2E; }
                        float pv = fast_exp2(x);
                        if (decltype(checked)::value && (q < qs || q >= qe)) pv = 0.f;
                        float pd = pv, dpd = dp[r];
                        if (DROP) {   // (row key of query q from the stage's table, this lane's key as the column)
                            const bool kp = have_bits ? ((kw >> (8 * i + j)) & 1u) != 0u : drop_keep_bits(w4[j], (uint32_t)key_c, p.drop);
                            pd = kp ? pv * p.drop.scale : 0.f;
                            dpd = kp ? dpd * p.drop.scale : 0.f;
                        }
                        sc[r] = pd;                    // dropped probabilities feed dV
                        dp[r] = pv * (dpd - d4[j]);    // dS
                    }
                }
            };
            // (only where it pays and fits: the dense-mask and dropout forms sit at the 256-register limit and would spill)
            if (MODE != MASK_DENSE && !DROP && __all(q0 >= qs && q0 + 32 <= qe)) p_ds_rows(std::false_type{});
            else p_ds_rows(std::true_type{});
        }
#ifdef OBTE_DEBUG_HOOKS
        if (!(p.dbg_skip & 2))
#endif
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 pf = pack8(sc, 8 * kk);
            const bf16x8 dsf = pack8(dp, 8 * kk);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<D>(Dt, 16 * kk, dt, lane), pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<D>(Qt, 16 * kk, dt, lane), dsf, dk[dt], 0, 0, 0);
            }
        }
#ifdef OBTE_DEBUG_HOOKS
        if (p.dbg_skip & 2) { asm volatile("" ::"v"(sc), "v"(dp)); }
#endif
        if (more) store_stats(smem + (cur ^ 1) * STAGE);
        if (!p.no_wait) dma_wait_all();
        if (DROP && BITS) { asm volatile("" : "+v"(keyw_next)); keyw = keyw_next; }   // (its load is waited for HERE, not behind the next slice's requests)
#ifdef OBTE_DEBUG_HOOKS
        if (!(p.dbg_skip & 8))
#endif
        __syncthreads();
    }

    OBTE_STAMP(p, 2);
    {
        // dV needs no rotation: rounded first (64 fp32 registers become 32) and sent out through the wave's own V rows in LDS (no
        // other wave reads them), the rotation table of the key's row requested before anything is stored, then dK the same way
        const int64_t key0 = (int64_t)key - (lane & 31);   // the wave's first key
        const int rows_ok = (int)min((int64_t)32, (int64_t)T - key0);
        char* wl = Vblk + wave * (32 * 2 * D);
        bf16* dk0 = p.dqkv + (b * T + key0) * ld + C + hd * D;
        RopeRow<D> rr;
        rr.load(p.rope_cos, p.rope_sin, key_c, h);
        bf16x4 gb[ND * 4];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) gb[4 * dt + i][j] = f2bf(dv[dt][4 * i + j]);
        wave_rows_out<D>(wl, gb, dk0 + C, ld, rows_ok, lane);
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float g[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = dk[dt][4 * i + j] * p.scale;
                rr.apply(g, dt, i);
#pragma unroll
                for (int j = 0; j < 4; ++j) gb[4 * dt + i][j] = f2bf(g[j]);
            }
        wave_rows_out<D>(wl, gb, dk0, ld, rows_ok, lane);
    }
#ifdef OBTE_DEBUG_HOOKS
    OBTE_STAMP(p, 3);
    if (p.dbg_times) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); OBTE_STAMP(p, 4); }
#endif
}

// The kernels proper: the bodies above behind their own entry points, and — for a dense additive mask whose device flag
// (obte_mask_bounds) may say "this IS a range mask" — ONE entry point per pass that holds both bodies and branches on the flag
// (uniform).  The host cannot read the flag without a sync; launching both representations with a gate in each cost three
// empty launches per attention call (forward, dQ, dK/dV: ~6 us of stream time each, 2.3 ms per step with the reference's mask
// tensors).
#define OBTE_ATTN_KERNEL(NAME)                                                                                         \
    template <int D, int MODE, bool DROP>                                                                              \
    __global__ __launch_bounds__(FwdShape<DROP>::NW * 64, 1) void NAME##_kernel(AttnParams p) {                        \
        extern __shared__ __attribute__((aligned(16))) char smem[];                                                    \
        NAME##_body<D, MODE, DROP>(p, smem);                                                                           \
    }                                                                                                                  \
    template <int D, bool DROP>                                                                                        \
    __global__ __launch_bounds__(FwdShape<DROP>::NW * 64, 1) void NAME##_gated_kernel(AttnParams pr, AttnParams pd) {  \
        extern __shared__ __attribute__((aligned(16))) char smem[];                                                    \
        if (__builtin_amdgcn_readfirstlane(*pr.gate) == 1) NAME##_body<D, MASK_RANGES, DROP>(pr, smem);                \
        else NAME##_body<D, MASK_DENSE, DROP>(pd, smem);                                                               \
    }
OBTE_ATTN_KERNEL(attn_fwd)
OBTE_ATTN_KERNEL(attn_bwd_dq)
OBTE_ATTN_KERNEL(attn_bwd_dkdv)
#undef OBTE_ATTN_KERNEL
// dropout with the forward's keep bits (head size 128, key ranges or no mask)
template <int MODE>
__global__ __launch_bounds__(FwdShape<true>::NW * 64, 1) void attn_bwd_dkdv_bits_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_bwd_dkdv_body<128, MODE, true, true>(p, smem);
}
template __global__ void attn_bwd_dkdv_bits_kernel<MASK_NONE>(AttnParams);
template __global__ void attn_bwd_dkdv_bits_kernel<MASK_RANGES>(AttnParams);

// Explicit instantiations (implicit instantiation alone left some host stubs undefined with hipcc / ROCm 7.2).
#define OBTE_INST_ATTN(D, M)                                                       \
    template __global__ void attn_fwd_kernel<D, M, false>(AttnParams);             \
    template __global__ void attn_bwd_dq_kernel<D, M, false>(AttnParams);          \
    template __global__ void attn_bwd_dkdv_kernel<D, M, false>(AttnParams);        \
    template __global__ void attn_fwd_kernel<D, M, true>(AttnParams);              \
    template __global__ void attn_bwd_dq_kernel<D, M, true>(AttnParams);           \
    template __global__ void attn_bwd_dkdv_kernel<D, M, true>(AttnParams);
OBTE_INST_ATTN(64, 0) OBTE_INST_ATTN(64, 1) OBTE_INST_ATTN(64, 2)
OBTE_INST_ATTN(128, 0) OBTE_INST_ATTN(128, 1) OBTE_INST_ATTN(128, 2)
#undef OBTE_INST_ATTN
#define OBTE_INST_GATED(D)                                                                      \
    template __global__ void attn_fwd_gated_kernel<D, false>(AttnParams, AttnParams);           \
    template __global__ void attn_fwd_gated_kernel<D, true>(AttnParams, AttnParams);            \
    template __global__ void attn_bwd_dq_gated_kernel<D, false>(AttnParams, AttnParams);        \
    template __global__ void attn_bwd_dq_gated_kernel<D, true>(AttnParams, AttnParams);         \
    template __global__ void attn_bwd_dkdv_gated_kernel<D, false>(AttnParams, AttnParams);      \
    template __global__ void attn_bwd_dkdv_gated_kernel<D, true>(AttnParams, AttnParams);
OBTE_INST_GATED(64) OBTE_INST_GATED(128)
#undef OBTE_INST_GATED

template <typename K>
void set_smem(K kern, int bytes) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// a dense mask wins: key_ranges passed beside it only bound the loops (obte_mask_bounds)
int mask_mode(const int32_t* ranges, const obte_bf16* mask) { return mask ? MASK_DENSE : (ranges ? MASK_RANGES : MASK_NONE); }

template <int D>
int launch_fwd(const AttnParams& p, int mode, hipStream_t st) {
    const int smem = 2 * FwdShape<false>::STAGES * 64 * 2 * D + 64;   // >= the dropout variant's two stages
    const dim3 grid_d((unsigned)((p.q_blk_off ? cdiv64(p.stat_hs, 32 * FwdShape<true>::NW) + p.B : cdiv64(p.T, 32 * FwdShape<true>::NW) * p.B) * p.H)), block_d(64 * FwdShape<true>::NW);
    // (a gathered query set: only blocks that can hold queries — at most ceil(n / 256) + B of them — in a compact grid: block_id_rows)
    const dim3 grid((unsigned)((p.q_blk_off ? cdiv64(p.stat_hs, 32 * FwdShape<false>::NW) + p.B : cdiv64(p.T, 32 * FwdShape<false>::NW) * p.B) * p.H)), block(64 * FwdShape<false>::NW);
#define GO(M)                                                                                 \
    do {                                                                                      \
        if (p.drop.thresh16) {                                                                \
            set_smem(attn_fwd_kernel<D, M, true>, smem);                                      \
            hipLaunchKernelGGL((attn_fwd_kernel<D, M, true>), grid_d, block_d, smem, st, p);  \
        } else {                                                                              \
            set_smem(attn_fwd_kernel<D, M, false>, smem);                                     \
            hipLaunchKernelGGL((attn_fwd_kernel<D, M, false>), grid, block, smem, st, p);     \
        }                                                                                     \
    } while (0)
    if (mode == MASK_NONE) GO(MASK_NONE); else if (mode == MASK_RANGES) GO(MASK_RANGES); else GO(MASK_DENSE);
#undef GO
    OBTE_CHECK_LAUNCH("obte_attn_fwd");
    return OBTE_OK;
}

#ifdef OBTE_DEBUG_HOOKS
// OBTE_ATTN_TIMES=1 (debug build): where a dK/dV launch spends its time, from s_memrealtime stamps (100 MHz) of every workgroup
static void debug_report_times(const char* who, unsigned long long* dev, int n, hipStream_t st) {
    static std::vector<unsigned long long> h;
    h.resize((size_t)n * 8);
    if (hipStreamSynchronize(st) != hipSuccess) return;
    if (hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    unsigned long long t0 = ~0ull, tend = 0, first_done = ~0ull;
    double seg[4] = {0, 0, 0, 0};
    unsigned long long last_start = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned long long* r = &h[(size_t)i * 8];
        if (r[0] < t0) t0 = r[0];
        if (r[0] > last_start) last_start = r[0];
        if (r[4] > tend) tend = r[4];
        if (r[4] < first_done) first_done = r[4];
        for (int k = 0; k < 4; ++k) seg[k] += (double)(r[k + 1] - r[k]);
    }
    {   // loop time by XCD (workgroup id % 8) and its spread: who finishes late?
        double xs[8] = {0}, lo = 1e30, hi = 0; int xn[8] = {0};
        for (int i = 0; i < n; ++i) {
            const double d = (double)(h[(size_t)i * 8 + 2] - h[(size_t)i * 8 + 1]) * 0.01;
            xs[i & 7] += d; xn[i & 7]++; if (d < lo) lo = d; if (d > hi) hi = d;
        }
        fprintf(stderr, "[attn %s loop us] min %.2f max %.2f | by XCD:", who, lo, hi);
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %.2f", xn[x] ? xs[x] / xn[x] : 0.0);
        fprintf(stderr, "\n");
    }
    fprintf(stderr, "[attn %s times, us] %d workgroups: first entry -> last entry %.2f | mean prologue %.2f, loop %.2f, epilogue issue %.2f, store drain %.2f | "
            "first entry -> first done %.2f, -> last done %.2f\n", who, n, (last_start - t0) * 0.01, seg[0] / n * 0.01, seg[1] / n * 0.01, seg[2] / n * 0.01,
            seg[3] / n * 0.01, (first_done - t0) * 0.01, (tend - t0) * 0.01);
}
#endif

template <int D>
int launch_bwd(const AttnParams& p, int mode, hipStream_t st) {
    {
        const int smem = 4 * 64 * 2 * D + 64;
        const dim3 grid_d((unsigned)((p.q_blk_off ? cdiv64(p.stat_hs, 32 * FwdShape<true>::NW) + p.B : cdiv64(p.T, 32 * FwdShape<true>::NW) * p.B) * p.H)), block_d(64 * FwdShape<true>::NW);
        const dim3 grid((unsigned)((p.q_blk_off ? cdiv64(p.stat_hs, 32 * FwdShape<false>::NW) + p.B : cdiv64(p.T, 32 * FwdShape<false>::NW) * p.B) * p.H)), block(64 * FwdShape<false>::NW);
#define GO(M)                                                                                    \
    do {                                                                                         \
        if (p.drop.thresh16) {                                                                   \
            set_smem(attn_bwd_dq_kernel<D, M, true>, smem);                                      \
            hipLaunchKernelGGL((attn_bwd_dq_kernel<D, M, true>), grid_d, block_d, smem, st, p);  \
        } else {                                                                                 \
            set_smem(attn_bwd_dq_kernel<D, M, false>, smem);                                     \
            hipLaunchKernelGGL((attn_bwd_dq_kernel<D, M, false>), grid, block, smem, st, p);     \
        }                                                                                        \
    } while (0)
        if (mode == MASK_NONE) GO(MASK_NONE); else if (mode == MASK_RANGES) GO(MASK_RANGES); else GO(MASK_DENSE);
#undef GO
        OBTE_CHECK_LAUNCH("obte_attn_bwd(dq)");
#ifdef OBTE_DEBUG_HOOKS
        if (p.dbg_times) debug_report_times("dq", p.dbg_times, (int)(p.drop.thresh16 ? grid_d.x : grid.x), st);
#endif
    }
    {
        const int smem = 2 * (2 * 32 * 2 * D + 384) + 32 * FwdShape<false>::NW * 2 * D + 64;   // >= the dropout variant's
        const dim3 grid_d((unsigned)(cdiv64(p.T, 32 * FwdShape<true>::NW) * p.H * p.B)), block_d(64 * FwdShape<true>::NW);
        const dim3 grid((unsigned)(cdiv64(p.T, 32 * FwdShape<false>::NW) * p.H * p.B)), block(64 * FwdShape<false>::NW);
        if (p.drop.thresh16 && p.drop_bits_in && D == 128 && mode != MASK_DENSE) {
            if (mode == MASK_NONE) {
                set_smem(attn_bwd_dkdv_bits_kernel<MASK_NONE>, smem);
                hipLaunchKernelGGL((attn_bwd_dkdv_bits_kernel<MASK_NONE>), grid_d, block_d, smem, st, p);
            } else {
                set_smem(attn_bwd_dkdv_bits_kernel<MASK_RANGES>, smem);
                hipLaunchKernelGGL((attn_bwd_dkdv_bits_kernel<MASK_RANGES>), grid_d, block_d, smem, st, p);
            }
        } else {
#define GO(M)                                                                                      \
    do {                                                                                           \
        if (p.drop.thresh16) {                                                                     \
            set_smem(attn_bwd_dkdv_kernel<D, M, true>, smem);                                      \
            hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, M, true>), grid_d, block_d, smem, st, p);  \
        } else {                                                                                   \
            set_smem(attn_bwd_dkdv_kernel<D, M, false>, smem);                                     \
            hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, M, false>), grid, block, smem, st, p);     \
        }                                                                                          \
    } while (0)
        if (mode == MASK_NONE) GO(MASK_NONE); else if (mode == MASK_RANGES) GO(MASK_RANGES); else GO(MASK_DENSE);
#undef GO
        }
        OBTE_CHECK_LAUNCH("obte_attn_bwd(dkdv)");
#ifdef OBTE_DEBUG_HOOKS
        if (p.dbg_times) debug_report_times("dkdv", p.dbg_times, (int)(p.drop.thresh16 ? grid_d.x : grid.x), st);
#endif
    }
    return OBTE_OK;
}

}  // namespace

// Timing-only diagnostics (they make the kernels return WRONG results), compiled in only with -DOBTE_DEBUG_HOOKS (`make debug`
// builds libomnibiote_hip_debug.so; tools/attn_fixed_*.sh load it through OBTE_LIB_PATH): a stray environment variable cannot
// reach them in the shipped library.  OBTE_ATTN_DEBUG=tiles:N -> max_tiles = N + 1 (N tiles per workgroup); =nowait -> no_wait.
#ifdef OBTE_DEBUG_HOOKS
static void debug_warn_once(const char* what) {
    static bool said = false;
    if (!said) { said = true; fprintf(stderr, "libomnibiote_hip (DEBUG build): %s is active — attention results are WRONG, timing only\n", what); }
}
static int debug_no_wait() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("OBTE_ATTN_DEBUG"); v = (e && !strcmp(e, "nowait")) ? 1 : 0; if (v) debug_warn_once("OBTE_ATTN_DEBUG=nowait"); }
    return v;
}
static int debug_max_tiles() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("OBTE_ATTN_DEBUG");
        v = (e && !strncmp(e, "tiles:", 6)) ? atoi(e + 6) + 1 : 0;
        if (v) debug_warn_once("OBTE_ATTN_DEBUG=tiles:N");
    }
    return v;
}
static int debug_skip() {
    static int v = -1;
    { const char* e = getenv("OBTE_ATTN_SKIP"); v = e ? atoi(e) : 0; if (v) debug_warn_once("OBTE_ATTN_SKIP"); }
    return v;
}
static unsigned long long* debug_times_buffer(int64_t max_groups) {
    static int on = -1;
    static unsigned long long* buf = nullptr;
    static int64_t cap = 0;
    if (on < 0) { const char* e = getenv("OBTE_ATTN_TIMES"); on = (e && e[0] == '1') ? 1 : 0; }
    if (!on) return nullptr;
    if (cap < max_groups) {
        if (buf) (void)hipFree(buf);
        if (hipMalloc((void**)&buf, (size_t)max_groups * 64) != hipSuccess) { buf = nullptr; cap = 0; return nullptr; }
        cap = max_groups;
    }
    return buf;
}
#else
static int debug_no_wait() { return 0; }
static int debug_max_tiles() { return 0; }
static int debug_skip() { return 0; }
static unsigned long long* debug_times_buffer(int64_t) { return nullptr; }
#endif

static int check_common(const char* who, const void* qkv, int64_t B, int64_t T, int H, int D, const int32_t* ranges,
                        const obte_bf16* mask) {
    OBTE_REQUIRE(qkv, "%s: null qkv", who);
    OBTE_REQUIRE(B > 0 && T > 0 && H > 0 && B < 65536 && H < 65536, "%s: bad B/T/H", who);
    OBTE_REQUIRE(D == 64 || D == 128, "%s: head_dim must be 64 or 128 (got %d)", who, D);
    OBTE_REQUIRE(T < (1 << 24), "%s: T too large", who);
    OBTE_REQUIRE(B * H * ((T + 127) / 128) < (1ll << 30), "%s: too many (batch, head, row block) workgroups", who);
    return OBTE_OK;
}

// A dense additive mask that obte_mask_bounds found to be EXACTLY a range mask (ranges_exact[0] == 1 on the device: every
// row's allowed keys are one contiguous run of zeros shared by all heads, every key's queries one contiguous run) is served
// by the range kernels, which skip the element-by-element mask reads (forward 85 -> 55 us, backward 353 -> 174 us at the
// hot-path shape).  The host cannot know the flag without a sync, so BOTH representations are launched and the kernels of
// the one that does not apply return at once (AttnParams::gate).  Every mask the reference's trainer builds qualifies
// (train_encoder.py:25-57), and so do key-padding masks; any other additive mask takes the dense kernels as before.
template <int D>
int launch_fwd_gated(AttnParams p, const int32_t* flag, hipStream_t st) {
    AttnParams pr = p;
    pr.mask = nullptr; pr.gate = flag;          // the range-mode view of the same call
    p.gate = flag;
    const int smem = 2 * FwdShape<false>::STAGES * 64 * 2 * D + 64;
    const dim3 grid((unsigned)(cdiv64(p.T, 32 * FwdShape<false>::NW) * p.H * p.B)), block(64 * FwdShape<false>::NW);
    if (p.drop.thresh16) {
        set_smem(attn_fwd_gated_kernel<D, true>, smem);
        hipLaunchKernelGGL((attn_fwd_gated_kernel<D, true>), grid, block, smem, st, pr, p);
    } else {
        set_smem(attn_fwd_gated_kernel<D, false>, smem);
        hipLaunchKernelGGL((attn_fwd_gated_kernel<D, false>), grid, block, smem, st, pr, p);
    }
    OBTE_CHECK_LAUNCH("obte_attn_fwd(gated)");
    return OBTE_OK;
}
template <int D>
int launch_bwd_gated(AttnParams p, const int32_t* flag, hipStream_t st) {
    AttnParams pr = p;
    pr.mask = nullptr; pr.gate = flag;          // pr.query_bounds: the exact per-key query ranges
    p.gate = flag;
    const dim3 grid((unsigned)(cdiv64(p.T, 32 * FwdShape<false>::NW) * p.H * p.B)), block(64 * FwdShape<false>::NW);
    {
        const int smem = 4 * 64 * 2 * D + 64;
        if (p.drop.thresh16) {
            set_smem(attn_bwd_dq_gated_kernel<D, true>, smem);
            hipLaunchKernelGGL((attn_bwd_dq_gated_kernel<D, true>), grid, block, smem, st, pr, p);
        } else {
            set_smem(attn_bwd_dq_gated_kernel<D, false>, smem);
            hipLaunchKernelGGL((attn_bwd_dq_gated_kernel<D, false>), grid, block, smem, st, pr, p);
        }
        OBTE_CHECK_LAUNCH("obte_attn_bwd(dq, gated)");
    }
    {
        const int smem = 2 * (2 * 32 * 2 * D + 384) + 32 * FwdShape<false>::NW * 2 * D + 64;
        if (p.drop.thresh16) {
            set_smem(attn_bwd_dkdv_gated_kernel<D, true>, smem);
            hipLaunchKernelGGL((attn_bwd_dkdv_gated_kernel<D, true>), grid, block, smem, st, pr, p);
        } else {
            set_smem(attn_bwd_dkdv_gated_kernel<D, false>, smem);
            hipLaunchKernelGGL((attn_bwd_dkdv_gated_kernel<D, false>), grid, block, smem, st, pr, p);
        }
        OBTE_CHECK_LAUNCH("obte_attn_bwd(dkdv, gated)");
    }
    return OBTE_OK;
}

extern "C" int64_t obte_attn_drop_bits_bytes(int64_t B, int64_t T, int32_t n_head) {
    return B * n_head * ((T + 31) / 32) * T * 4;
}

extern "C" int obte_attn_fwd(const obte_attn_fwd_args* a, obte_stream s) {
    OBTE_REQUIRE(a, "obte_attn_fwd: null args");
    int rc = check_common("obte_attn_fwd", a->qkv, a->B, a->T, a->n_head, a->head_dim, a->key_ranges, a->mask);
    if (rc) return rc;
    OBTE_REQUIRE(a->o && a->lse, "obte_attn_fwd: null output");
    AttnParams p = {};
    p.qkv = (const bf16*)a->qkv; p.o = (bf16*)a->o; p.lse = a->lse;
    p.key_ranges = a->key_ranges; p.mask = (const bf16*)a->mask; p.mask_sb = a->mask_sb; p.mask_sh = a->mask_sh; p.mask_sq = a->mask_sq;
    p.B = a->B; p.T = a->T; p.H = a->n_head; p.scale = a->scale;
    p.q_src = p.qkv; p.q_ld = 3 * (int64_t)a->n_head * a->head_dim; p.stat_hs = a->T;   // the queries are the rows of qkv
    OBTE_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f, "obte_attn_fwd: dropout p must be in [0,1)");
    p.drop = make_drop(a->dropout_p, a->dropout_seed, OBTE_SITE_ATTN);
    p.max_tiles = debug_max_tiles(); p.no_wait = debug_no_wait();
    const int mode = mask_mode(a->key_ranges, a->mask);
    p.drop_bits_out = (p.drop.thresh16 != 0 && a->head_dim == 128 && mode != MASK_DENSE) ? a->drop_bits : nullptr;
    const int prof = obte_prof_begin((hipStream_t)s, 100, a->B * a->n_head, a->T, a->head_dim);
    if (mode == MASK_DENSE && a->key_ranges && a->ranges_exact)
        rc = a->head_dim == 128 ? launch_fwd_gated<128>(p, a->ranges_exact, (hipStream_t)s) : launch_fwd_gated<64>(p, a->ranges_exact, (hipStream_t)s);
    else
        rc = a->head_dim == 128 ? launch_fwd<128>(p, mode, (hipStream_t)s) : launch_fwd<64>(p, mode, (hipStream_t)s);
    obte_prof_end(prof, (hipStream_t)s);
    return rc;
}

#include <atomic>
static std::atomic<int>& attn_bwd_mode_ref() {
    static std::atomic<int> m{[] { const char* e = getenv("OBTE_ATTN_BWD"); return (e && !strcmp(e, "two")) ? 1 : 0; }()};
    return m;
}
static int attn_bwd_mode() { return attn_bwd_mode_ref().load(std::memory_order_relaxed); }
extern "C" int obte_attn_bwd_select(int mode) { return attn_bwd_mode_ref().exchange(mode == 1 ? 1 : 0); }
// 0 wherever the one-kernel form does not apply (head size, T > 8192, T whose packed rows exceed 32-bit byte offsets): the caller then
// allocates nothing and obte_attn_bwd takes the two-kernel form
extern "C" int64_t obte_attn_bwd_ws_bytes(int64_t B, int64_t T, int32_t n_head, int32_t head_dim) {
    if (B <= 0 || T <= 0 || n_head <= 0 || head_dim != 128 || T * 3 * n_head * 128 * 2 >= (1ll << 31)) return 0;
    return fused_bwd_ws_bytes(B, T, n_head);
}

static int attn_bwd_impl(const obte_attn_bwd_args* a, int delta_ready, obte_stream s);
extern "C" int obte_attn_bwd(const obte_attn_bwd_args* a, obte_stream s) { return attn_bwd_impl(a, 0, s); }
// common.h: the same with a->delta already holding rowsum(dO o O)
int obte_attn_bwd_delta_ready(const obte_attn_bwd_args* a, obte_stream s) { return attn_bwd_impl(a, 1, s); }
static int attn_bwd_impl(const obte_attn_bwd_args* a, int delta_ready, obte_stream s) {
    OBTE_REQUIRE(a, "obte_attn_bwd: null args");
    int rc = check_common("obte_attn_bwd", a->qkv, a->B, a->T, a->n_head, a->head_dim, a->key_ranges, a->mask);
    if (rc) return rc;
    OBTE_REQUIRE(a->o && a->d_o && a->lse && a->delta && a->dqkv, "obte_attn_bwd: null pointer");
    OBTE_REQUIRE((a->rope_cos == nullptr) == (a->rope_sin == nullptr), "obte_attn_bwd: pass both RoPE tables or neither");
    AttnParams p = {};
    p.qkv = (const bf16*)a->qkv; p.o_in = (const bf16*)a->o; p.d_o = (const bf16*)a->d_o; p.lse_in = a->lse; p.delta = a->delta;
    p.dqkv = (bf16*)a->dqkv; p.rope_cos = a->rope_cos; p.rope_sin = a->rope_sin; p.query_bounds = a->mask ? a->query_bounds : nullptr;
    p.key_ranges = a->key_ranges; p.mask = (const bf16*)a->mask; p.mask_sb = a->mask_sb; p.mask_sh = a->mask_sh; p.mask_sq = a->mask_sq;
    p.B = a->B; p.T = a->T; p.H = a->n_head; p.scale = a->scale;
    p.q_src = p.qkv; p.q_ld = 3 * (int64_t)a->n_head * a->head_dim; p.dq_dst = p.dqkv; p.dq_ld = p.q_ld; p.stat_hs = a->T;   // the queries are the rows of qkv
    OBTE_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f, "obte_attn_bwd: dropout p must be in [0,1)");
    p.drop = make_drop(a->dropout_p, a->dropout_seed, OBTE_SITE_ATTN);
    p.max_tiles = debug_max_tiles(); p.no_wait = debug_no_wait(); p.dbg_skip = debug_skip();
    p.delta_ready = delta_ready;
    p.dbg_times = debug_times_buffer(a->B * a->n_head * ((a->T + 127) / 128));
    const int mode = mask_mode(a->key_ranges, a->mask);
    p.drop_bits_in = (p.drop.thresh16 != 0 && a->head_dim == 128 && mode != MASK_DENSE) ? a->drop_bits : nullptr;
    const int prof = obte_prof_begin((hipStream_t)s, 101, a->B * a->n_head, a->T, a->head_dim);
    const int64_t fused_ws = a->head_dim == 128 ? fused_bwd_ws_bytes(a->B, a->T, a->n_head) : 0;   // 0: the one-kernel form does not apply
    if (mode != MASK_DENSE && a->head_dim == 128 && (p.drop.thresh16 == 0 || p.drop_bits_in) && a->ws && attn_bwd_mode() == 0 && fused_ws > 0 &&
        a->ws_bytes >= fused_ws && a->T * 3 * a->n_head * 128 * 2 < (1ll << 31)) {   // (its per-lane byte offsets are 32-bit)
        if (mode == MASK_RANGES) p.query_bounds = nullptr;   // a range mask without a dense tensor: symmetric (the key's own range)
        rc = launch_bwd_fused(p, mode, a->ws, (hipStream_t)s);
    } else if (mode == MASK_DENSE && a->key_ranges && a->query_bounds && a->ranges_exact)
        rc = a->head_dim == 128 ? launch_bwd_gated<128>(p, a->ranges_exact, (hipStream_t)s) : launch_bwd_gated<64>(p, a->ranges_exact, (hipStream_t)s);
    else
        rc = a->head_dim == 128 ? launch_bwd<128>(p, mode, (hipStream_t)s) : launch_bwd<64>(p, mode, (hipStream_t)s);
    obte_prof_end(prof, (hipStream_t)s);
    return rc;
}


// ---- obte_mask_bounds -------------------------------------------------------------------------------------------
namespace {
constexpr float MASKING = -3.0e4f;   // an additive entry at or below this is "masked": exp() of it underflows to 0

// one wave per (b, q) row: [first, last+1) over the keys any head allows; (0,T) and row_full = 1 if some head allows none.
// exact (nullable): cleared unless the row's allowed keys are, for every head alike, one contiguous run of exact zeros.
__global__ __launch_bounds__(256) void mask_row_bounds_kernel(const bf16* mask, int64_t sb, int64_t sh, int64_t sq, int64_t B, int H,
                                                               int64_t T, int32_t* kb, uint8_t* row_full, int32_t* exact) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= B * T) return;
    const int64_t b = row / T, q = row % T;
    const int heads = sh == 0 ? 1 : H;
    int lo = (int)T, hi = 0;
    bool full = false, is_range = true;
    int l0 = 0, u0 = 0;
    for (int h = 0; h < heads; ++h) {
        const bf16* r = mask + b * sb + h * sh + q * sq;
        int l = (int)T, u = 0, cnt = 0, nonzero = 0;
        for (int64_t k = lane; k < T; k += 64) {
            const float v = bf2f(r[k]);
            if (v > MASKING) { l = min(l, (int)k); u = max(u, (int)k + 1); ++cnt; nonzero += v != 0.f; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            l = min(l, __shfl_xor(l, o, 64)); u = max(u, __shfl_xor(u, o, 64));
            cnt += __shfl_xor(cnt, o, 64); nonzero += __shfl_xor(nonzero, o, 64);
        }
        if (u == 0) full = true;
        if (u == 0 || cnt != u - l || nonzero != 0) is_range = false;
        if (h == 0) { l0 = l; u0 = u; } else if (l != l0 || u != u0) is_range = false;
        lo = min(lo, l); hi = max(hi, u);
    }
    if (full) { lo = 0; hi = (int)T; }
    if (lane == 0) {
        kb[row * 2] = lo; kb[row * 2 + 1] = hi; row_full[row] = full ? 1 : 0;
        if (exact && !is_range) *exact = 0;   // plain store of the same value from every violating row
    }
}

// thread per (b, key), grid.y chunks of 64 queries: [first, last+1) over the queries that allow the key (or are "full" rows);
// cnt (nullable): the number of such queries, for the contiguity check of mask_col_check_kernel
__global__ __launch_bounds__(256) void mask_col_bounds_kernel(const bf16* mask, int64_t sb, int64_t sh, int64_t sq, int64_t B, int H,
                                                               int64_t T, const uint8_t* row_full, int32_t* qb, int32_t* cnt) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t b = blockIdx.z;
    if (k >= T) return;
    const int heads = sh == 0 ? 1 : H;
    const int64_t q0 = (int64_t)blockIdx.y * 64, q1 = min(q0 + 64, T);
    int lo = (int)T, hi = 0, n = 0;
    for (int64_t q = q0; q < q1; ++q) {
        bool ok = row_full[b * T + q] != 0;
        for (int h = 0; h < heads && !ok; ++h) ok = bf2f(mask[b * sb + h * sh + q * sq + k]) > MASKING;
        if (ok) { lo = min(lo, (int)q); hi = max(hi, (int)q + 1); ++n; }
    }
    if (hi > 0) {
        atomicMin(&qb[(b * T + k) * 2], lo);
        atomicMax(&qb[(b * T + k) * 2 + 1], hi);
        if (cnt) atomicAdd(&cnt[b * T + k], n);
    }
}
__global__ void mask_col_init_kernel(int32_t* qb, int32_t* cnt, int32_t* exact, int64_t n, int T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { qb[i * 2] = T; qb[i * 2 + 1] = 0; if (cnt) cnt[i] = 0; }   // a key no query can see: empty [T, 0)
    if (i == 0 && exact) *exact = 1;
}
// every key's queries must be one contiguous run as well (the dK/dV kernel walks [first, last+1) of its key)
__global__ void mask_col_check_kernel(const int32_t* qb, const int32_t* cnt, int32_t* exact, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int lo = qb[i * 2], hi = qb[i * 2 + 1];
        if (cnt[i] != (hi > lo ? hi - lo : 0)) *exact = 0;
    }
}
}  // namespace

extern "C" int obte_mask_bounds(const obte_bf16* mask, int64_t mask_sb, int64_t mask_sh, int64_t mask_sq, int64_t B, int32_t n_head,
                                int64_t T, int32_t* key_bounds, int32_t* query_bounds, uint8_t* row_scratch, int32_t* ranges_exact,
                                int32_t* col_scratch, obte_stream s) {
    OBTE_REQUIRE(mask && key_bounds && query_bounds && row_scratch, "obte_mask_bounds: null pointer");
    OBTE_REQUIRE(!ranges_exact || col_scratch, "obte_mask_bounds: ranges_exact needs col_scratch (int32 [B*T])");
    OBTE_REQUIRE(B > 0 && T > 0 && n_head > 0 && B < 65536 && T < (1 << 24), "obte_mask_bounds: bad B/T/H");
    hipStream_t st = (hipStream_t)s;
    int32_t* cnt = ranges_exact ? col_scratch : nullptr;
    hipLaunchKernelGGL(mask_col_init_kernel, dim3((unsigned)cdiv64(B * T, 256)), dim3(256), 0, st, query_bounds, cnt, ranges_exact, B * T, (int)T);
    OBTE_CHECK_LAUNCH("obte_mask_bounds(init)");
    hipLaunchKernelGGL(mask_row_bounds_kernel, dim3((unsigned)cdiv64(B * T, 4)), dim3(256), 0, st, (const bf16*)mask, mask_sb, mask_sh, mask_sq, B,
                       (int)n_head, T, key_bounds, row_scratch, ranges_exact);
    OBTE_CHECK_LAUNCH("obte_mask_bounds(rows)");
    hipLaunchKernelGGL(mask_col_bounds_kernel, dim3((unsigned)cdiv64(T, 256), (unsigned)cdiv64(T, 64), (unsigned)B), dim3(256), 0, st,
                       (const bf16*)mask, mask_sb, mask_sh, mask_sq, B, (int)n_head, T, (const uint8_t*)row_scratch, query_bounds, cnt);
    OBTE_CHECK_LAUNCH("obte_mask_bounds(columns)");
    if (ranges_exact) {
        hipLaunchKernelGGL(mask_col_check_kernel, dim3((unsigned)cdiv64(B * T, 256)), dim3(256), 0, st, (const int32_t*)query_bounds,
                           (const int32_t*)cnt, ranges_exact, B * T);
        OBTE_CHECK_LAUNCH("obte_mask_bounds(column check)");
    }
    return OBTE_OK;
}

// ---- queries at listed rows only (common.h: obte_attn_rows) ------------------------------------------------------------------------------
namespace {
// first index i in [0, n) with rows[i] >= v (n if none)
__device__ __forceinline__ int64_t rows_lower_bound(const int64_t* rows, int64_t n, int64_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (rows[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// thread i < B + 1: q_off; thread j < n: position and key range of gathered row j; thread m < B T: inverse index and, with a mask, the
// gathered rows that see key m (the positions of the key's own range: symmetric masks)
__global__ __launch_bounds__(256) void attn_rows_prep_kernel(const int64_t* rows, int64_t n, int64_t B, int64_t T, const int32_t* kr_full,
                                                              int32_t* q_off, int32_t* q_blk_off, int32_t* q_pos, int32_t* kr_rows, int32_t* qb_rows, int32_t* inv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i <= B) q_off[i] = (int32_t)rows_lower_bound(rows, n, i * T);   // (q_blk_off: attn_rows_blocks_kernel, once q_off is complete)
    if (i < n) {
        const int64_t r = rows[i];
        q_pos[i] = (int32_t)(r % T);
        if (kr_full) { kr_rows[2 * i] = kr_full[2 * r]; kr_rows[2 * i + 1] = kr_full[2 * r + 1]; }
    }
    if (i < B * T) {
        const int64_t at = rows_lower_bound(rows, n, i);
        inv[i] = (at < n && rows[at] == i) ? (int32_t)at : -1;
        if (kr_full) {
            const int64_t b = i / T, base = rows_lower_bound(rows, n, b * T);
            const int64_t qs = max((int64_t)kr_full[2 * i], (int64_t)0), qe = min((int64_t)kr_full[2 * i + 1], T);
            qb_rows[2 * i] = (int32_t)(rows_lower_bound(rows, n, b * T + qs) - base);
            qb_rows[2 * i + 1] = qe > qs ? (int32_t)(rows_lower_bound(rows, n, b * T + qe) - base) : qb_rows[2 * i];
        }
    }
}
// q_blk_off[b] = number of 256-query blocks of the batch elements before b (the compact grid of the query-major kernels): one
// workgroup, a running sum over chunks of 256 batch elements (as part of the kernel above it was one thread's chain of B binary
// searches: 40 of that launch's 50 us)
__global__ __launch_bounds__(256) void attn_rows_blocks_kernel(const int32_t* q_off, int32_t* q_blk_off, int64_t B) {
    __shared__ int32_t part[256];
    __shared__ int32_t carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < B; b0 += 256) {
        const int64_t b = b0 + tid;
        part[tid] = b < B ? (q_off[b + 1] - q_off[b] + 255) / 256 : 0;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {   // inclusive scan
            const int32_t v = tid >= o ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (b < B) q_blk_off[b + 1] = carry + part[tid];
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
    if (tid == 0) q_blk_off[0] = 0;
}
// one wave per row, 16 B per lane per step
__global__ __launch_bounds__(256) void rows_fill_strided_kernel(const bf16* src, const int32_t* inv, bf16* dst, int64_t total_rows, int64_t ld, int cols) {
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= total_rows) return;
    const int lane = threadIdx.x & 63;
    const int32_t j = inv[m];
    for (int c = lane * 8; c < cols; c += 512) {
        bf16x8 v = {};
        if (j >= 0) v = *reinterpret_cast<const bf16x8*>(src + (int64_t)j * cols + c);
        *reinterpret_cast<bf16x8*>(dst + m * ld + c) = v;
    }
}
__global__ __launch_bounds__(256) void rows_gather_strided_kernel(const bf16* src, int64_t ld, const int64_t* rows, bf16* dst, int64_t n_rows, int cols) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t r = rows[i];
    for (int c = lane * 8; c < cols; c += 512) *reinterpret_cast<bf16x8*>(dst + i * cols + c) = *reinterpret_cast<const bf16x8*>(src + r * ld + c);
}
}  // namespace

int obte_attn_rows_prep(const int64_t* rows, int64_t n, int64_t B, int64_t T, const int32_t* key_ranges_full, int32_t* q_off, int32_t* q_blk_off,
                        int32_t* q_pos, int32_t* key_ranges_rows, int32_t* query_bounds_rows, int32_t* inv, obte_stream s) {
    OBTE_REQUIRE(rows && n > 0 && n <= B * T && q_off && q_blk_off && q_pos && inv, "obte_attn_rows_prep: bad arguments");
    OBTE_REQUIRE(!key_ranges_full || (key_ranges_rows && query_bounds_rows), "obte_attn_rows_prep: a mask needs both output tables");
    hipLaunchKernelGGL(attn_rows_prep_kernel, dim3((unsigned)cdiv64(B * T + 1, 256)), dim3(256), 0, (hipStream_t)s, rows, n, B, T, key_ranges_full,
                       q_off, q_blk_off, q_pos, key_ranges_rows, query_bounds_rows, inv);
    OBTE_CHECK_LAUNCH("obte_attn_rows_prep");
    hipLaunchKernelGGL(attn_rows_blocks_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, q_off, q_blk_off, B);
    OBTE_CHECK_LAUNCH("obte_attn_rows_prep(blocks)");
    return OBTE_OK;
}
int obte_rows_fill_strided_bf16(const obte_bf16* src, const int32_t* inv, obte_bf16* dst, int64_t total_rows, int64_t ld, int32_t cols, obte_stream s) {
    OBTE_REQUIRE(src && inv && dst && total_rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && ld >= cols, "obte_rows_fill_strided_bf16: bad arguments");
    hipLaunchKernelGGL(rows_fill_strided_kernel, dim3((unsigned)cdiv64(total_rows, 4)), dim3(256), 0, (hipStream_t)s, (const bf16*)src, inv, (bf16*)dst, total_rows, ld, (int)cols);
    OBTE_CHECK_LAUNCH("obte_rows_fill_strided_bf16");
    return OBTE_OK;
}
int obte_rows_gather_strided_bf16(const obte_bf16* src, int64_t ld, const int64_t* rows, obte_bf16* dst, int64_t n_rows, int32_t cols, obte_stream s) {
    OBTE_REQUIRE(src && rows && dst && n_rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && ld >= cols, "obte_rows_gather_strided_bf16: bad arguments");
    hipLaunchKernelGGL(rows_gather_strided_kernel, dim3((unsigned)cdiv64(n_rows, 4)), dim3(256), 0, (hipStream_t)s, (const bf16*)src, ld, rows, (bf16*)dst, n_rows, (int)cols);
    OBTE_CHECK_LAUNCH("obte_rows_gather_strided_bf16");
    return OBTE_OK;
}

static int rows_common(const char* who, const obte_attn_rows* r, const void* q, int64_t B, int64_t T, const obte_bf16* mask, float dropout_p) {
    OBTE_REQUIRE(r && q && r->q_off && r->q_blk_off && r->q_pos && r->n > 0 && r->n <= B * T, "%s: bad row set", who);
    OBTE_REQUIRE(!mask && dropout_p >= 0.f && dropout_p < 1.f, "%s: the rows form takes key ranges or no mask; dropout p in [0,1)", who);
    OBTE_REQUIRE((r->key_ranges == nullptr) == (r->query_bounds == nullptr), "%s: a range mask needs both the rows' key ranges and the keys' row bounds", who);
    return OBTE_OK;
}
int obte_attn_fwd_rows(const obte_attn_fwd_args* a, const obte_attn_rows* r, const obte_bf16* q, obte_stream s) {
    OBTE_REQUIRE(a, "obte_attn_fwd_rows: null args");
    int rc = check_common("obte_attn_fwd_rows", a->qkv, a->B, a->T, a->n_head, a->head_dim, nullptr, nullptr);
    if (rc) return rc;
    rc = rows_common("obte_attn_fwd_rows", r, q, a->B, a->T, a->mask, a->dropout_p);
    if (rc) return rc;
    OBTE_REQUIRE(a->o && a->lse, "obte_attn_fwd_rows: null output");
    AttnParams p = {};
    p.qkv = (const bf16*)a->qkv; p.o = (bf16*)a->o; p.lse = a->lse;
    p.key_ranges = r->key_ranges;
    p.B = a->B; p.T = a->T; p.H = a->n_head; p.scale = a->scale;
    p.q_off = r->q_off; p.q_blk_off = r->q_blk_off; p.q_src = (const bf16*)q; p.q_ld = (int64_t)a->n_head * a->head_dim; p.q_pos = r->q_pos; p.stat_hs = r->n;
    p.drop = make_drop(a->dropout_p, a->dropout_seed, OBTE_SITE_ATTN);   // (hashed in both passes: no keep bits for a gathered query set)
    p.max_tiles = debug_max_tiles(); p.no_wait = debug_no_wait();
    const int mode = r->key_ranges ? MASK_RANGES : MASK_NONE;
    const int prof = obte_prof_begin((hipStream_t)s, 102, a->n_head * r->n, a->T, a->head_dim);   // kind 102 / 103: (queries x heads, keys, head size)
    rc = a->head_dim == 128 ? launch_fwd<128>(p, mode, (hipStream_t)s) : launch_fwd<64>(p, mode, (hipStream_t)s);
    obte_prof_end(prof, (hipStream_t)s);
    return rc;
}
int obte_attn_bwd_rows(const obte_attn_bwd_args* a, const obte_attn_rows* r, const obte_bf16* q, obte_bf16* dq, obte_stream s) {
    OBTE_REQUIRE(a, "obte_attn_bwd_rows: null args");
    int rc = check_common("obte_attn_bwd_rows", a->qkv, a->B, a->T, a->n_head, a->head_dim, nullptr, nullptr);
    if (rc) return rc;
    rc = rows_common("obte_attn_bwd_rows", r, q, a->B, a->T, a->mask, a->dropout_p);
    if (rc) return rc;
    OBTE_REQUIRE(a->o && a->d_o && a->lse && a->delta && a->dqkv && dq, "obte_attn_bwd_rows: null pointer");
    OBTE_REQUIRE((a->rope_cos == nullptr) == (a->rope_sin == nullptr), "obte_attn_bwd_rows: pass both RoPE tables or neither");
    AttnParams p = {};
    p.qkv = (const bf16*)a->qkv; p.o_in = (const bf16*)a->o; p.d_o = (const bf16*)a->d_o; p.lse_in = a->lse; p.delta = a->delta;
    p.dqkv = (bf16*)a->dqkv; p.rope_cos = a->rope_cos; p.rope_sin = a->rope_sin;
    p.key_ranges = r->key_ranges; p.query_bounds = r->query_bounds;
    p.B = a->B; p.T = a->T; p.H = a->n_head; p.scale = a->scale;
    p.q_off = r->q_off; p.q_blk_off = r->q_blk_off; p.q_src = (const bf16*)q; p.q_ld = (int64_t)a->n_head * a->head_dim; p.dq_dst = (bf16*)dq; p.dq_ld = p.q_ld;
    p.q_pos = r->q_pos; p.stat_hs = r->n;
    p.drop = make_drop(a->dropout_p, a->dropout_seed, OBTE_SITE_ATTN);
    p.max_tiles = debug_max_tiles(); p.no_wait = debug_no_wait(); p.dbg_skip = debug_skip();
    const int mode = r->key_ranges ? MASK_RANGES : MASK_NONE;
    const int prof = obte_prof_begin((hipStream_t)s, 103, a->n_head * r->n, a->T, a->head_dim);
    rc = a->head_dim == 128 ? launch_bwd<128>(p, mode, (hipStream_t)s) : launch_bwd<64>(p, mode, (hipStream_t)s);
    obte_prof_end(prof, (hipStream_t)s);
    return rc;
}
