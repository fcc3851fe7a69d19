// One-kernel attention backward for head_dim 128 (training/model.py:115-146 under loss.backward()): dQ, dK and dV from FIVE
// MFMA products per tile — S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS, dQ^T += K^T dS^T — where the two-kernel
// form of attention.hip recomputes S and dP in both kernels (seven products).  Structure (cdna_hip_programming.md, Appendix B
// "Attention backward"): a workgroup = 4 waves (one per SIMD, the whole 512-register file each) = 256 keys of one
// (batch, head); each wave keeps dK^T and dV^T of its 64 keys in 256 ACCUMULATOR registers (a[0:255], asm MFMAs with "+a"
// operands) while the workgroup sweeps the 32-row query slices; S and dP are formed with the KEY on the MFMA lane, so their
// accumulators, converted in place, already are the B operands of the dV^T / dK^T products; only dS crosses LDS, once, as
// the B operand of the dQ product, for which wave w owns the 32 head-dim columns 32 w .. 32 w + 31 over all 256 keys.
//
// Built with -mllvm -amdgpu-mfma-vgpr-form (csrc/Makefile): with a 512-register budget hipcc otherwise selects the
// AGPR-destination form for EVERY builtin MFMA and moves the S / dP tiles to the vector ALU through v_accvgpr_read, one
// instruction per element (measured in round 2: 350 copies per tile).  With the flag the builtin products (S, dP, dQ) live in
// architectural VGPRs and the resident dK / dV accumulators, only ever touched by the asm statements, in the AGPR half.
//
// dQ sums over the key blocks of a (batch, head), i.e. over T/256 workgroups: an ORDERED HAND-OFF, no atomics, bitwise
// reproducible.  The running fp32 dQ^T tile of every 32-query slice lives in a scratch buffer; the key blocks that take part in a
// slice add to it one after the other in a fixed order, the last one applies the softmax scale and the inverse RoPE and rounds
// once to bf16.  Every workgroup sweeps its slices in a ROTATED order (key block k starts k/nkb of the way round), so the
// members of one slice's chain reach it nsl/nkb iterations apart: nobody waits in steady state, and all workgroups of a
// (batch, head) start and finish together.  The order of a slice's chain is the order of the members' own visiting times
// (ties: the lower key block first) — every workgroup derives it from the per-key-block slice ranges the prep kernel publishes —
// so every wait is for a strictly earlier (time, key block) pair: no cycle, whatever is resident when.  Publication follows
// cdna_hip_programming.md Guideline 16, R1: write-through (sc1) stores of the tile, every storing wave's vmcnt(0), the
// workgroup's barrier, one lane's agent-scope add on the slice's counter; the consumer polls the counter with sc1 loads
// (the poll of the NEXT hand-off is issued an iteration ahead) and reads the tile with sc1 loads.  Spins are bounded.
#include "attn_common.h"

namespace {
using namespace obte_attn;

constexpr int FB_NW = 4;       // waves per workgroup (one per SIMD)
constexpr int FB_KEYS = 256;   // keys per workgroup: 64 per wave, two 32-key MFMA tiles
#ifndef FB_FIN_IN_LOOP
#define FB_FIN_IN_LOOP 1
#endif
#ifndef FB_RQ_AT
#define FB_RQ_AT 2   // the C0 step that requests the rotation entries
#endif
constexpr bool FIN_IN_LOOP = FB_FIN_IN_LOOP != 0;   // a chain's last member finishes its dQ tile where it forms it (0: in a pass after the loop, as in round 4)
#ifndef FB_RA
#define FB_RA 4
#define FB_RD 4
#define FB_RC 3
#endif

template <int D>
struct FusedShape {
    static constexpr int QB = 32 * 2 * D;            // one 32-row tile of Q or dO
    static constexpr int STAGE = 2 * QB + 384;       // Q tile, dO tile, three row constants of the 32 queries (-lse / scale, -delta, -lse log2 e)
    static constexpr int KBYTES = FB_KEYS * 2 * D;   // the workgroup's K rows (row reads for S, transposed reads for dQ)
    static constexpr int DSG = FB_KEYS * 8 + 32;     // dS image: one group of 4 queries = [256 keys][4 q] bf16 (+ 32 B: bank spread)
    static constexpr int DSB = 8 * DSG;              // one dS buffer (8 groups = 32 queries)
    static constexpr int NSTG = 2;
    static constexpr int TAB = 1024;                 // chain table: one byte per slice this workgroup sweeps (nsl <= 1024)
    static constexpr int ACC = 32 * D * 4;           // the dQ^T tile handed on to this workgroup, staged by LDS-DMA (each wave its own 4 KiB)
    static constexpr int RAW = FB_NW * 512;          // per wave: the next slice's lse / delta values and this slice's counter as LDS-DMA left them
    static constexpr int SMEM = NSTG * STAGE + KBYTES + 2 * DSB + ACC + 64 + TAB + 1024 + RAW;   // (+ the key blocks' slice ranges)
};

struct FusedParams {
    AttnParams a;
    float* dq_acc;         // fp32 [B*H][nsl + 1][4 waves][4][64 lanes][4]: the running dQ^T tile of every slice (+ one trash tile per (batch, head))
    int32_t* flags;        // int32 [B*H][nsl]: contributions completed per slice (zeroed by the prep kernel, every call)
    int32_t* kb_bounds;    // int32 [B][nkb][2]: the slices [t_begin, t_end) each key block sweeps (prep kernel)
    int32_t* status;       // the library's device status word (pinned host memory, common.h): OBTE_STATUS_ATTN_BWD_HANDOFF is OR-ed into it
                           // when a bounded spin gives up (a protocol bug or a workgroup that never ran); the host checks it at its next
                           // synchronising point (obte_device_status) and treats the launch's gradients as invalid
    int spin_limit;        // polls before a wait gives up (2^20: ~2 s; the fault-injection hook of the tests shortens it)
    int no_signal_slice;   // fault injection (tests): the counter of this slice of (batch 0, head 0) is never added to (-1: off)
    int nkb, nsl;
};

// one MFMA into a RESIDENT accumulator (AGPRs).  s_nop 1: the A / B operands may have been written by the vector ALU just
// before (cdna_hip_programming.md §5.7 item 2: hipcc pads nothing for an asm statement).
#ifndef FB_ACC_NOP
#define FB_ACC_NOP 1
#endif
__device__ __forceinline__ void mfma_acc(f32x16& acc, const bf16x8& a, const bf16x8& b) {
#if FB_ACC_NOP
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#else   // (timing experiments only: without the wait states a vector-ALU write of an operand right in front of the statement is a hazard)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#endif
}

// B fragment of the dQ product from the dS image: MFMA column = query (lane & 31), k element j = key row
// krow0 + 8 (j >> 2) + 4 (lane >> 5) + (j & 3) — the order tr_frag<D>() delivers the K^T A operand in.
// Image: group g (queries 4 g .. 4 g + 3) at g * DSG, key row r at + 8 r (4 bf16).  Lane 4 q' + p of a 16-lane group supplies
// row q', query group p of its 16 queries; the transposing read hands lane i query i of the four rows.
template <int D>
__device__ __forceinline__ bf16x8 ds_frag(const char* img, int krow0, int lane) {
    const int li = lane & 15, hh = lane >> 5, qhalf = (lane >> 4) & 1;
    const int g = 4 * qhalf + (li & 3);
    const char* t = img + g * FusedShape<D>::DSG + (krow0 + 4 * hh + (li >> 2)) * 8;
    return join8(lds_read_tr16(t), lds_read_tr16(t + 64));
}

// LDS reads as  <one VGPR base> + <immediate>: the bases of a slice (stage and dS-image parity are run-time values) are formed once
// per iteration and made opaque, so every fragment read of the loop is a bare ds_read with an offset field — left to itself hipcc
// re-derives stage + constant in scalar registers and adds the lane part per read (two VALU + two SALU instructions per read pair).
typedef __attribute__((address_space(3))) const bf16x8* lds_b128_t;
typedef __attribute__((address_space(3))) const f32x4* lds_f128_t;
__device__ __forceinline__ bf16x8 lds_row(uint32_t base, int imm) { return *(lds_b128_t)(uintptr_t)(base + (uint32_t)imm); }
__device__ __forceinline__ f32x4 lds_f4(uint32_t base, int imm) { return *(lds_f128_t)(uintptr_t)(base + (uint32_t)imm); }
__device__ __forceinline__ bf16x4 lds_tr(uint32_t base, int imm) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)(base + (uint32_t)imm));
    return __builtin_bit_cast(bf16x4, t);
}
__device__ __forceinline__ uint32_t opaque(uint32_t x) { asm volatile("" : "+v"(x)); return x; }

// Prep launch, three jobs by block range:
//  [0, nb_delta)   delta[b,h,q] = sum_d O[b,q,h,d] dO[b,q,h,d] (the softmax backward's row constant): one wave per (b, q) row of the
//                  [M, C] activations, 16 lanes hold one head's 128 values;
//  next nb_flags   zero the slices' hand-off counters (every call: the protocol counts from zero);
//  next B * nkb    the slice range [t_begin, t_end) of key block (b, kb): the union of the query ranges of its keys.
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(FusedParams fp, int mode, int nb_delta, int nb_flags) {
    const AttnParams& p = fp.a;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int T = (int)p.T, H = p.H;
    int blk = blockIdx.x;
    if (blk < nb_delta) {
        const int64_t row = (int64_t)blk * 4 + wave;
        if (row >= p.B * T) return;
        const int64_t b = row / T, q = row % T;
        const int C = H * 128;
        for (int c0 = 0; c0 < C; c0 += 512) {   // 64 lanes x 8 elements
            const int c = c0 + lane * 8;
            float s = 0.f;
            if (c < C) {
                const bf16x8 x = *reinterpret_cast<const bf16x8*>(p.o_in + row * C + c);
                const bf16x8 y = *reinterpret_cast<const bf16x8*>(p.d_o + row * C + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += bf2f(x[j]) * bf2f(y[j]);
            }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m, 64);
            if (c < C && (lane & 15) == 0) p.delta[(b * H + c / 128) * T + q] = s;
        }
        return;
    }
    blk -= nb_delta;
    if (blk < nb_flags) {
        const int64_t i = (int64_t)blk * 256 + tid;
        if (i < p.B * H * fp.nsl) fp.flags[i] = 0;
        return;
    }
    blk -= nb_flags;
    __shared__ int red[8];
    const int b = blk / fp.nkb, kb = blk % fp.nkb;
    const int key = kb * FB_KEYS + tid;
    int lo = T, hi = 0;
    if (key < T) {
        int qs = 0, qe = T;
        if (mode == MASK_RANGES) {
            const int32_t* src = p.query_bounds ? p.query_bounds : p.key_ranges;
            qs = max(src[((int64_t)b * T + key) * 2], 0);
            qe = min(src[((int64_t)b * T + key) * 2 + 1], T);
        }
        if (qe > qs) { lo = qs; hi = qe; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
    if (lane == 0) { red[wave] = lo; red[4 + wave] = hi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) { lo = min(lo, red[w]); hi = max(hi, red[4 + w]); }
        fp.kb_bounds[(b * fp.nkb + kb) * 2] = hi > lo ? lo / 32 : 0;
        fp.kb_bounds[(b * fp.nkb + kb) * 2 + 1] = hi > lo ? (hi + 31) / 32 : 0;
    }
}

// DROP: attention-probability dropout (model.py:83,121) from the keep bits the forward left behind (obte_attn_fwd_args::drop_bits:
// one word per KEY and 32-query slice, bit i = query 32 t + i kept) — with O = (P o M) V, M = keep / (1 - p):
//     dV^T += dO^T (P o M),   dS = P o (M o dP - delta) = (P o M) o dP - P delta,   delta = rowsum(dO o O) as without dropout,
// so per element one bit field extract, the scaled probability under the bit as a mask (it IS the dV operand), and an FMA in
// place of the multiply; both key tiles take the row-constant path of key tile 1 (dP chains from zero, -delta added in the
// arithmetic: the masked product must not carry the constant).  The lane's two words of the NEXT slice are loaded beside its tiles.
template <int D, int MODE, int SK = 0, bool DROP = false>   // SK: timing-only skip bits (debug library), compile-time so that the variants cost no branches
__global__ __launch_bounds__(FB_NW * 64, 1) void attn_bwd_fused_kernel(FusedParams fp) {
    static_assert(D == 128, "the one-kernel backward is written for head_dim 128");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using S = FusedShape<D>;
    constexpr int NS = D / 16, ND = D / 32;
    const AttnParams& p = fp.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const BlockId bid_ = block_id(fp.nkb, p.H);
    const int kb = bid_.blk, hd = bid_.hd;
    const int64_t b = bid_.b;
    const int T = (int)p.T;
    const int C = p.H * D;
    const int64_t ld = 3 * (int64_t)C;
    const int64_t bh = b * p.H + hd;
#ifdef OBTE_DEBUG_HOOKS
    unsigned long long t_entry = 0;
    if (p.dbg_times) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry) :: "memory");
#endif

    char* Kblk = smem + S::NSTG * S::STAGE;
    char* dsimg = Kblk + S::KBYTES;
    char* accst = dsimg + 2 * S::DSB;
    int* scratch = reinterpret_cast<int*>(accst + S::ACC);
    char* rawst = accst + S::ACC + 64 + S::TAB + 1024;

    // ---- this wave's 64 keys: two MFMA tiles of 32, key on the lane ----------------------------------------------
    int key[2], qs[2], qe[2];
    bool k_ok[2];
    bf16x8 vf[2][NS];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        key[kt] = kb * FB_KEYS + wave * 64 + 32 * kt + (lane & 31);
        k_ok[kt] = key[kt] < T;
        const int key_c = k_ok[kt] ? key[kt] : T - 1;
        qs[kt] = 0; qe[kt] = T;
        if (MODE == MASK_RANGES) {   // the queries that may see this key: the per-key table, or (symmetric masks) the key's own range
            const int32_t* src = p.query_bounds ? p.query_bounds : p.key_ranges;
            qs[kt] = max(src[(b * T + key_c) * 2], 0);
            qe[kt] = min(src[(b * T + key_c) * 2 + 1], T);
        }
        if (!k_ok[kt]) { qs[kt] = 0; qe[kt] = 0; }
        const bf16* vptr = p.qkv + (b * T + key_c) * ld + 2 * C + hd * D;
#pragma unroll
        for (int s = 0; s < NS; ++s) vf[kt][s] = *reinterpret_cast<const bf16x8*>(vptr + 16 * s + 8 * h);
    }
    // dropout: this lane's keep words, one per key tile and slice (slice t at + t * T words)
    const uint32_t* kw_src[2] = {nullptr, nullptr};
    uint32_t kw[2] = {0u, 0u}, kw_next[2] = {0u, 0u};
    if (DROP) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
            kw_src[kt] = p.drop_bits_in + ((b * p.H + hd) * (int64_t)fp.nsl) * T + (k_ok[kt] ? key[kt] : T - 1);
    }
    {   // the K rows of the block: LDS-DMA, zero-filled past T
        TileDma<D, FB_KEYS, FB_NW> dmk;
        dmk.init(wave, lane, ld);
        const int64_t row0 = (int64_t)kb * FB_KEYS;
        dmk.issue(p.qkv + (b * T + row0) * ld + C + hd * D, (((int64_t)T - row0) * ld - (C + hd * D)) * 2, Kblk, wave);
    }
    // The slices this key block sweeps: its own range straight from the prep kernel's table (two scalar loads: uniform), so that the
    // first slice's tiles can be requested beside the K rows and the V fragments — ONE memory round trip for the whole prologue
    // (it used to be three in a row: K / V / the table, then the chain places, then the first slice).
    // (a SCALAR load: it has its own counter, so reading it does not wait for the vector loads issued above)
    int t_begin, t_end;
    {
        const int32_t* bp = fp.kb_bounds + (b * fp.nkb + kb) * 2;
        unsigned long long both;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(both) : "s"(bp) : "memory");
        t_begin = (int)(both & 0xffffffffu); t_end = (int)(both >> 32);
    }
    const int n_sl = t_end - t_begin;
    const int rot = n_sl > 0 ? (int)(((unsigned)kb * (unsigned)n_sl) / (unsigned)fp.nkb) : 0;

    const bf16* qbase = p.qkv + b * T * ld + hd * D;
    const bf16* dobase = p.d_o + b * T * C + hd * D;
    const float* lse_b = p.lse_in + bh * T;
    const float* del_b = p.delta + bh * T;
    const float scale2 = p.scale * LOG2E;
    const float inv_scale = 1.0f / p.scale;

    f32x16 dk[2][ND], dv[2][ND];   // resident: 256 accumulator registers
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[kt][i][r] = 0.f; dv[kt][i][r] = 0.f; }

    // Q / dO tiles by LDS-DMA into a ring of three stages.  ONE buffer descriptor per tensor for the whole kernel (this batch
    // element's rows of this head: reads past row T return zeros); the slice enters through the per-lane byte offset, so issuing a
    // slice is four { v_add, s_mov m0, buffer_load ... lds } — the per-tile descriptors of attention.hip cost ~45 scalar
    // instructions per slice, in front of the first MFMA.
    TileDma<D, 32, FB_NW> dmq, dmd;
    dmq.init(wave, lane, ld);
    dmd.init(wave, lane, C);
    const i32x4_t rs_q = make_rsrc_words(qbase, ((int64_t)T * ld - hd * D) * 2);
    const i32x4_t rs_d = make_rsrc_words(dobase, ((int64_t)T * C - hd * D) * 2);
    const int q_step = (int)(32 * ld * 2), d_step = 32 * C * 2;   // bytes per slice
    static_assert(TileDma<D, 32, FB_NW>::NP == 2, "two pieces per wave and tile");
    // i = position in this workgroup's visiting order (stage i mod 3), t = the slice visited there
    auto stage_of = [&](int i) { return smem + (i % S::NSTG) * S::STAGE; };
    auto issue_piece = [&](int i, int t, int j) {   // j = 0, 1: Q pieces; 2, 3: dO pieces of this wave
        const uint32_t base = lds_addr_of(stage_of(i)) + wave * 1024;
        if (j < 2) lds_dma16(rs_q, base + FB_NW * j * 1024, dmq.voff[j] + t * q_step);
        else lds_dma16(rs_d, base + S::QB + FB_NW * (j - 2) * 1024, dmd.voff[j - 2] + t * d_step);
    };
    auto slice_at = [&](int i) { const int x = i + rot; return t_begin + (x >= n_sl ? x - n_sl : x); };   // i < n_sl
    // lse (threads 0..31) / delta (32..63) of the next slice: ONE load per thread, issued with the tile's LDS-DMA and not touched
    // until the end of the iteration (vmcnt retires in issue order: a use right after the load would also drain the DMA just issued)
    float st_l = 0.f;
    auto load_stats = [&](int q0) {
        if (tid < 96) {
            const int q = min(q0 + (tid & 31), T - 1);
            st_l = ((tid >> 5) == 1 ? del_b : lse_b)[q];
        }
    };
    // the loop's form: issued by EVERY wave (the counted waits below count the same operations in all four), through asm (hipcc
    // does not wait for it) and by LDS-DMA into the wave's own 256 bytes: NO load with a register destination stays in flight
    // across statements in this kernel.  (It used to: hipcc counts such a destination as written when the asm statement ends and
    // is free to copy it, and it did copy it in front of the end-of-iteration wait — harmless while the load had long returned,
    // the previous slice's lse / delta whenever the memory system was busy enough to make it late.)
    const float* const st_src = ((tid >> 5) == 1 ? del_b : lse_b);
    const uint32_t a_raw = lds_addr_of(rawst) + wave * 512;
    const float* const raw_st = reinterpret_cast<const float*>(rawst + wave * 512) + lane;
    const int* const raw_poll = reinterpret_cast<const int*>(rawst + wave * 512 + 256) + lane;
    auto load_stats_issue = [&](int q0) {
        const float* a = st_src + min(q0 + (tid & 31), T - 1);
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" : : "s"(a_raw), "v"(a) : "memory");
    };
    // row constants of a slice: -lse / scale and -delta are the INITIAL ACCUMULATORS of key tile 0's two chains (S - lse / scale,
    // dP - delta come out of the MFMAs ready); key tile 1 starts from zero and adds -lse log2(e), -delta in its arithmetic (its
    // chains would otherwise hold 32 more registers while tile 0's are still being consumed)
    auto store_stats = [&](char* stage, int q0) {
        if (tid < 96) {
            const int row = tid >> 5;
            float v = row == 0 ? -st_l * inv_scale : (row == 1 ? -st_l : -st_l * LOG2E);
            if (q0 + (tid & 31) >= T) v = row == 1 ? 0.f : -INFINITY;   // rows past T: p = exp2(-inf) = 0
            reinterpret_cast<float*>(stage + 2 * S::QB)[tid] = v;
        }
    };
    if (n_sl > 0) {
        const int t0 = slice_at(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_piece(0, t0, j);
        load_stats(t0 * 32);
        if (DROP) { kw[0] = kw_src[0][(int64_t)t0 * T]; kw[1] = kw_src[1][(int64_t)t0 * T]; }
    }
    // the visiting order and, for each slice, this workgroup's place in the slice's hand-off chain (header): tab[i] = place | last << 7
    // of the i-th slice visited, s(i) = t_begin + (i + rot) mod n — computed while the prologue's loads are in flight
    int* kbb = scratch + 16;                                        // [nkb][2] slice ranges of this batch element's key blocks
    uint8_t* tab = reinterpret_cast<uint8_t*>(scratch) + 64 + 1024;   // (the 1 KiB in between holds kbb: nkb <= 120)
    for (int i = tid; i < 2 * fp.nkb; i += FB_NW * 64) kbb[i] = fp.kb_bounds[b * fp.nkb * 2 + i];
    __syncthreads();
    for (int i = tid; i < n_sl; i += FB_NW * 64) {
        const unsigned x = (unsigned)(i + rot);
        const int s = t_begin + (int)(x >= (unsigned)n_sl ? x - (unsigned)n_sl : x);
        int place = 0, cnt = 0;
        for (int k2 = 0; k2 < fp.nkb; ++k2) {
            const int tb2 = kbb[2 * k2], te2 = kbb[2 * k2 + 1], n2 = te2 - tb2;
            if (s < tb2 || s >= te2) continue;
            ++cnt;
            if (k2 == kb) continue;
            const int rot2 = (int)(((unsigned)k2 * (unsigned)n2) / (unsigned)fp.nkb);
            int tau2 = s - tb2 - rot2;
            if (tau2 < 0) tau2 += n2;
            if (tau2 < i || (tau2 == i && k2 < kb)) ++place;
        }
        tab[i] = (uint8_t)(place | ((place == cnt - 1) ? 0x80 : 0));
    }
    // (published to every wave by the barrier that ends the prologue)
    if (n_sl > 0) store_stats(stage_of(0), slice_at(0) * 32);
    dma_wait_all();
    prologue_wait_all();
    __syncthreads();

    // ---- the hand-off of the dQ^T tiles (header) -------------------------------------------------------------------------------
    // this (batch, head)'s tiles through one buffer descriptor: tile of slice t at t * 16 KiB, this lane's four 16-byte pieces at
    // (4 wave + i) * 1 KiB + 16 lane
    const __amdgpu_buffer_rsrc_t rs_acc = make_rsrc(fp.dq_acc + bh * (fp.nsl + 1) * 4096, (int64_t)(fp.nsl + 1) * 16384);   // (+ 1: the trash tile, below)
    int32_t* const flag_b = fp.flags + bh * fp.nsl;
    const int mute = (fp.no_signal_slice >= 0 && bh == 0) ? fp.no_signal_slice : -2;   // fault injection (tests): this slice's counter is never added to
    const int acc_lane = wave * 4096 + lane * 16;
    // The loads of the hand-off are issued through inline asm: hipcc would otherwise put its own s_waitcnt vmcnt(0) in front of
    // their first use (and of unrelated instructions that reuse a register), which in this in-order queue also drains the LDS-DMA
    // just issued for the next slice.  Their completion is covered by the waits the loop has anyway (named at each use).
    const int32_t* const flag_s = flag_b;                                   // wave-uniform: an SGPR pair
    auto poll_issue = [&](int t) {                                          // counter of slice t -> the wave's raw area, NOT waited for
        const int off = t * 4;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2 sc1" : : "s"(a_raw + 256), "v"(off), "s"(flag_s) : "memory");
    };
    // the counter of slice t has reached `place` (every wave polls for itself: its own tile loads follow its own poll); `have` is
    // the value polled an iteration ago — in steady state it already suffices and nothing is loaded here
    // A wait that gives up is a hard failure of the launch, reported through the device status word: the wave goes on (with a tile
    // that is not the chain's: this launch's gradients are invalid and the host says so at its next synchronising point), and so that
    // one failure does not become nsl timeouts in a row, every later wait of this wave is cut short (`gave_up`) and the other waves
    // look at the word every 1024 polls.
    bool gave_up = false;
    auto wait_turn = [&](int t, int place, int have) {
        int spins = 0;
        while (have < place) {
            __builtin_amdgcn_s_sleep(8);
            const int off = t * 4;
            asm volatile("global_load_dword %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(have) : "v"(off), "s"(flag_s) : "memory");
            ++spins;
            if ((spins & 1023) == 0 && !gave_up)
                gave_up = (__hip_atomic_load(fp.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & OBTE_STATUS_ATTN_BWD_HANDOFF) != 0;
            if (gave_up || spins > fp.spin_limit) {
                if (lane == 0 && !gave_up) __hip_atomic_fetch_or(fp.status, OBTE_STATUS_ATTN_BWD_HANDOFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                gave_up = true;
                break;
            }
        }
    };
    // The tile so far is requested by LDS-DMA into this wave's own 4 KiB of a staging area (no VGPR destination: hipcc counts an
    // asm load's destination as written at the end of the statement and may copy or reuse the register while the data is still
    // in flight — tools/fused_audit.py caught exactly that with register-destination loads here), sc1 like every read of a
    // handed-off tile; it is read back by the wave that asked for it, behind that wave's own counted vmcnt — no barrier involved.
    const uint32_t a_acc = lds_addr_of(accst) + wave * 4096;
    const i32x4_t rs_accw = make_rsrc_words(fp.dq_acc + bh * (fp.nsl + 1) * 4096, (int64_t)(fp.nsl + 1) * 16384);
    auto acc_request = [&](int t, int i) {   // piece i of 4
        const int off = t * 16384 + acc_lane;
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen sc1 lds" : : "s"(a_acc + i * 1024), "v"(off + i * 1024), "s"(rs_accw) : "memory");   // (no instruction offset: it would move the LDS address too)
    };
    const uint32_t acc_rd = opaque(a_acc + lane * 16);
    auto acc_add = [&](f32x16& dq) {   // (behind the wait that covers the request)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 x = lds_f4(acc_rd, i * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) dq[4 * i + j] += x[j];
        }
    };
    // (register-destination form, for the epilogue only: each load is waited for in the statement that issues it)
    struct AccRegs { f32x4 x[4]; };
    auto load_acc_sync = [&](int t, AccRegs& r) {
        // (compiler-issued, plain loads: after the slice loop nothing hand-counted is in flight, so hipcc may put the loads of several
        //  tiles in flight and wait once — the asm form of round 4's first version waited inside every tile's statement, one
        //  dependent round trip per tile; plain because these tiles were last written by this very wave, same lanes and addresses,
        //  through this XCD's L2: the line there is the final one)
        const int off = t * 16384 + acc_lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) r.x[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_acc, off + i * 1024, 0, 0));
    };
    auto store_acc = [&](int t, const f32x16& dq, int i) {   // piece i of 4
        const f32x4 x = {dq[4 * i], dq[4 * i + 1], dq[4 * i + 2], dq[4 * i + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), rs_acc, t * 16384 + acc_lane + i * 1024, 0, 16);   // sc1: write-through
    };
    // the last member of a chain: softmax scale, inverse RoPE, one rounding to bf16; registers 4 i .. 4 i + 3 = head-dim columns
    // 32 wave + 8 i + 4 h .. + 3 of query 32 t + (lane & 31)
    typedef float v2f __attribute__((ext_vector_type(2)));
    struct RopeQ { v2f c[4], s[4]; };
    auto load_rope = [&](int t, RopeQ& r) {
        const int q = min(t * 32 + (lane & 31), T - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t at = (int64_t)q * (D / 2) + (32 * wave + 8 * i + 4 * h) / 2;
            r.c[i] = *reinterpret_cast<const v2f*>(p.rope_cos + at);
            r.s[i] = *reinterpret_cast<const v2f*>(p.rope_sin + at);
        }
    };
    // the loop's form: register-destination loads through asm, NOT waited for in the statement — the loop's own counted wait in A1
    // covers them (they are older than everything it leaves in flight) and every use is made to depend on that wait statement; that
    // hipcc leaves the registers alone in between is what tools/fused_audit.py checks on the ISA (tests/test_host_logic.py runs it)
    auto rope_request = [&](int t, RopeQ& r) {
        const int q = min(t * 32 + (lane & 31), T - 1);
        const int64_t at = (int64_t)q * (D / 2) + (32 * wave + 4 * h) / 2;   // piece i: + 4 i floats
        const float* pc = p.rope_cos + at;
        const float* ps = p.rope_sin + at;
        asm volatile("global_load_dwordx2 %0, %8, off\n\tglobal_load_dwordx2 %1, %8, off offset:16\n\t"
                     "global_load_dwordx2 %2, %8, off offset:32\n\tglobal_load_dwordx2 %3, %8, off offset:48\n\t"
                     "global_load_dwordx2 %4, %9, off\n\tglobal_load_dwordx2 %5, %9, off offset:16\n\t"
                     "global_load_dwordx2 %6, %9, off offset:32\n\tglobal_load_dwordx2 %7, %9, off offset:48"
                     : "=&v"(r.c[0]), "=&v"(r.c[1]), "=&v"(r.c[2]), "=&v"(r.c[3]), "=&v"(r.s[0]), "=&v"(r.s[1]), "=&v"(r.s[2]), "=&v"(r.s[3])
                     : "v"(pc), "v"(ps) : "memory");
    };
    auto store_final = [&](int t, const f32x16& dq, const RopeQ& r, int i) {   // piece i of 4
        const int q = t * 32 + (lane & 31);
        float g[4] = {dq[4 * i] * p.scale, dq[4 * i + 1] * p.scale, dq[4 * i + 2] * p.scale, dq[4 * i + 3] * p.scale};
        if (p.rope_cos) {
            const float e0 = g[0] * r.c[i].x + g[1] * r.s[i].x, o0 = -g[0] * r.s[i].x + g[1] * r.c[i].x;
            const float e1 = g[2] * r.c[i].y + g[3] * r.s[i].y, o1 = -g[2] * r.s[i].y + g[3] * r.c[i].y;
            g[0] = e0; g[1] = o0; g[2] = e1; g[3] = o1;
        }
        if (q < T) *reinterpret_cast<bf16x4*>(p.dqkv + (b * T + q) * ld + hd * D + 32 * wave + 8 * i + 4 * h) = bf16x4{f2bf(g[0]), f2bf(g[1]), f2bf(g[2]), f2bf(g[3])};
    };
    auto store_final4 = [&](int t, const f32x4& x, const RopeQ& r, int i) {   // the same from one piece's four values
        const int q = t * 32 + (lane & 31);
        float g[4] = {x[0] * p.scale, x[1] * p.scale, x[2] * p.scale, x[3] * p.scale};
        if (p.rope_cos) {
            const float e0 = g[0] * r.c[i].x + g[1] * r.s[i].x, o0 = -g[0] * r.s[i].x + g[1] * r.c[i].x;
            const float e1 = g[2] * r.c[i].y + g[3] * r.s[i].y, o1 = -g[2] * r.s[i].y + g[3] * r.c[i].y;
            g[0] = e0; g[1] = o0; g[2] = e1; g[3] = o1;
        }
        if (q < T) *reinterpret_cast<bf16x4*>(p.dqkv + (b * T + q) * ld + hd * D + 32 * wave + 8 * i + 4 * h) = bf16x4{f2bf(g[0]), f2bf(g[1]), f2bf(g[2]), f2bf(g[3])};
    };
    typedef __attribute__((address_space(3))) f32x4* lds_wf128_t;
    auto acc_park = [&](const f32x16& dq, int i) {   // piece i of the finished sum into this wave's staging area, where acc_add reads piece i from
        *(lds_wf128_t)(uintptr_t)(acc_rd + (uint32_t)(i * 1024)) = f32x4{dq[4 * i], dq[4 * i + 1], dq[4 * i + 2], dq[4 * i + 3]};
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- the slice loop, scheduled by hand ---------------------------------------------------------------------------------
    // One wave per SIMD: nothing hides a wave's vector-ALU or LDS latency except its own MFMAs, and hipcc left to itself runs
    // the phases one after the other (products, then all of the softmax arithmetic, then products: 3.1 us per slice against
    // 1.3 us of matrix-pipe time).  The iteration is therefore written as 80 SLOTS of one MFMA each, pinned in this order by
    // sched_barrier(0), every slot carrying the fragment reads of a later slot and its share of everything else:
    //   A0  16 slots  S0 / dP0 alternating (key tile 0)            beside: LDS-DMA of the next slice (one piece per slot pair), its lse / delta load
    //   D   16 slots  dQ^T tile of the PREVIOUS slice (its dS image was completed by that iteration's barrier)
    //                                                             beside: P0, dS0 of key tile 0, one element per slot
    //   A1  16 slots  S1 / dP1                                     beside: the previous slice's dQ contribution leaves
    //   C0  16 slots  dV0, dK0 per (query k-step, head-dim tile): resident accumulators, asm
    //                                                             beside: P1, dS1 of key tile 1, one element per slot; dS0 into the image
    //   C1   8 slots  dV1, dK1                                     beside: dS1 into the image
    //   ---- counted vmcnt (the next slice's tiles have landed) + the iteration's ONE barrier ----
    //   C1   8 slots                                               beside: the first fragments and row constants of the NEXT slice
    // (this order, not A0 A1 D C, because it keeps one key tile's S / dP pair live at a time: with both the loop needs ~30 more
    // registers than there are and hipcc spills the V fragments, reloading them behind vmcnt(0) in every slot)
    // The barrier sits where the matrix pipe has eight MFMAs with operands in hand, so nothing drains at the loop edge; that is
    // what the third stage is for (a wave ahead issues the DMA of slice t + 2 while a wave behind still reads stage t).
#define OBTE_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef OBTE_DEBUG_HOOKS
    // OBTE_ATTN_TIMES=1 (debug library): s_memtime at the phase boundaries of every slice, summed per workgroup (wave 0) — the SHARES
    // of the phases, not their lengths (each stamp drains the LDS reads in flight across it)
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    const bool stamping = p.dbg_times != nullptr;
#define OBTE_PHASE(k) do { if (stamping) { __builtin_amdgcn_sched_barrier(0); unsigned long long now_; \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); tsum[k] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); } } while (0)
    // timing-only (OBTE_ATTN_SKIP picks an instantiation, debug library): 1 D-phase reads, 2 A-phase reads, 4 softmax arithmetic,
    // 8 barrier, 16 DMA wait, 32 C-phase reads, 64 dS image + dQ stores, 128 DMA issue (results are wrong)
#define OBTE_SKIP(b) ((SK & (b)) != 0)
#else
#define OBTE_SKIP(b) false
#define OBTE_PHASE(k) do { } while (0)
#endif
    // state carried over the loop edge: the A-phase fragment ring (k-step n = 8 kt + s in slot n % 3, read two k-steps ahead),
    // key tile 0's two chains holding their row constants, and whether the slice needs the range test
    constexpr int RA = DROP ? 3 : FB_RA, RD = DROP ? 3 : FB_RD, RC = FB_RC;   // fragment rings: a fragment is read RING - 1 steps ahead of its MFMA
                                                                  // (dropout: the keep words and key tile 0's -delta chunks take the A ring's fourth slot; with it hipcc spills a V fragment)
    bf16x8 fq[RA], fd[RA], fk[RA];
    f32x16 sc0, dp0;
    bool inside = true;
    // lane parts of the fragment addresses (attn_common.h: row_frag, tr_frag; ds_frag above), formed once
    const int r31 = lane & 31, li = lane & 15;
    const uint32_t lrow0 = (16 * D) * (r31 >> 3) + 64 * (r31 & 7) + 16 * ((h ^ (r31 >> 2)) & 3);   // row reads, k-step even; odd: ^ 32
    const uint32_t ltr0 = 64 * (4 * h + (li >> 2)) + 16 * ((2 * ((lane >> 4) & 1) + ((li & 3) >> 1)) ^ h) + (li & 1) * 8;   // transposed reads, rows 0..7 of a group; 8..15: ^ 32, + 16 D
    const uint32_t lds0 = (4 * ((lane >> 4) & 1) + (li & 3)) * S::DSG + (4 * h + (li >> 2)) * 8;   // dS image
    const uint32_t a_smem = lds_addr_of(smem), a_k = lds_addr_of(Kblk), a_ds = lds_addr_of(dsimg);
    const uint32_t k_row[2] = {opaque(a_k + lrow0 + (16 * D) * (8 * wave)), opaque(a_k + (lrow0 ^ 32) + (16 * D) * (8 * wave))};   // this wave's 64 K rows
    const uint32_t k_tr[2] = {opaque(a_k + ltr0 + 512 * wave), opaque(a_k + (ltr0 ^ 32) + 16 * D + 512 * wave)};                   // K^T, head-dim tile = wave
    struct SliceBases { uint32_t row[2], tr[2], st; };   // of one stage: Q tile at +0, dO tile at +QB, row constants at +2 QB
    auto bases_of = [&](int i) {   // i = position in the visiting order
        const uint32_t a = a_smem + (i % S::NSTG) * S::STAGE;
        SliceBases b;
        b.row[0] = opaque(a + lrow0); b.row[1] = opaque(a + (lrow0 ^ 32));
        b.tr[0] = opaque(a + ltr0); b.tr[1] = opaque(a + (ltr0 ^ 32) + 16 * D);
        b.st = opaque(a + 2 * S::QB + 16 * h);
        return b;
    };
    auto rdA = [&](const SliceBases& sb, int n) {
        const int s = n & 7, kt = n >> 3, i = n % RA;
        if (OBTE_SKIP(2) && n > 2) return;
        fq[i] = lds_row(sb.row[s & 1], 512 * (s >> 1));
        fd[i] = lds_row(sb.row[s & 1], S::QB + 512 * (s >> 1));
        fk[i] = lds_row(k_row[s & 1], (16 * D) * (4 * kt) + 512 * (s >> 1));
    };
    // the row constants of the 16 query rows this lane's accumulator registers stand for, read straight into the accumulators
    auto row_init = [&](const SliceBases& sb, int which) {
        f32x16 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 x = lds_f4(sb.st, 128 * which + 32 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * i + j] = x[j];
        }
        return v;
    };
    // A slice some key of the wave is NOT seen by in full (a document boundary, keys past T): the score chain of a masked
    // (query, key) pair starts from -inf instead, so its P and dS come out as exact zeros with no test in the arithmetic.
    // query q0 + acc_row(r, h) inside [qs, qe)  <=>  (unsigned)(c_r - (qs - q0 - 4 h)) < qe - qs
    auto mask_init = [&](f32x16& sc, int kt, int q0) {
        const int m_lo = qs[kt] - q0 - 4 * h;
        const unsigned m_len = (unsigned)(qe[kt] - qs[kt]);
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = ((unsigned)((r & 3) + 8 * (r >> 2) - m_lo) < m_len) ? sc[r] : -INFINITY;
    };
#ifdef OBTE_DEBUG_HOOKS
    if (stamping) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast) :: "memory"); tsum[6] = tlast - t_entry; }   // slot 6: everything outside the slice loop (prologue here, the rest at the end)
#endif
    // loop-carried hand-off state: the previous slice (whose dQ^T tile this iteration forms and hands on), this workgroup's place
    // in that slice's chain, and the counter value polled for it an iteration ago
    int t_prev = 0, info_prev = 0;          // the slice visited in the previous iteration: its dQ^T tile is formed (D), joined with what the chain
                                            // handed on and stored (A1) in this one
    int t_sig = -1;                         // the slice whose tile was stored an iteration ago and has not been counted on yet (-1: none)
    int have_prev = 0;                      // the previous slice's counter as polled during the previous iteration
    for (int it = 0; it < n_sl; ++it) {
        // (the last iteration "prefetches" its own slice once more into the free stage: every iteration then issues the same
        //  operations, the counted waits are constants and the loop has no `is there a next slice` branches)
        const int t = slice_at(it), t_next = it + 1 < n_sl ? slice_at(it + 1) : t;
        const int cur = it & 1;
        const int info = tab[it];                       // place | last << 7 of slice t for this workgroup
        const bool last_p = (info_prev & 0x80) != 0, first_it = it == 0;
        const bool take_p = !first_it && (info_prev & 0x7f) > 0;   // somebody hands the previous slice's tile on to this workgroup
        int have_cur = 0;
        const int t_store = first_it ? fp.nsl : t_prev;
        const SliceBases sb = bases_of(it);
        const uint32_t ds_rd = opaque(a_ds + (cur ^ 1) * S::DSB + lds0);   // the previous slice's dS image (reads)
        char* img_cur = dsimg + cur * S::DSB;
        const int q0 = t * 32;
        // the previous slice's tile so far: once it is this workgroup's turn (in steady state the counter polled an iteration ago
        // already says so), four LDS-DMA pieces in A0's first slots; they are read back in A1
        wait_turn(t_prev, info_prev & 0x7f, have_prev);   // (place 0, the first iteration: no turn to wait for, the loop inside does not run)
        // A chain's last member finishes the previous slice's tile itself: the summed tile is parked in the wave's own staging area
        // (A1, where it would otherwise leave for the scratch buffer; the area is free between this iteration's read of the incoming
        // tile and the next iteration's request), the rotation entries of its queries are requested in C0, and the tile is scaled,
        // rotated back, rounded once and stored piece by piece in C1, where the registers of key tile 1's chains are free again.
        RopeQ rq;
        f32x4 fin_piece;
        const bool fin_p = FIN_IN_LOOP && last_p && !first_it;
        // the first fragments and the row constants of this slice (its tiles were published by the barrier that ended the previous iteration)
#pragma unroll
        for (int n = 0; n < RA - 1; ++n) rdA(sb, n);
        inside = __all(q0 >= qs[0] && q0 + 32 <= qe[0] && q0 >= qs[1] && q0 + 32 <= qe[1]);
        sc0 = row_init(sb, 0);
        dp0 = DROP ? zero16 : row_init(sb, 1);   // (dropout: the masked product must not carry -delta; it is added in the arithmetic)
        if (!inside) mask_init(sc0, 0, q0);
        const uint32_t kwh[2] = {kw[0] >> (4 * h), kw[1] >> (4 * h)};   // bit (r & 3) + 8 (r >> 2) = the query row of register r

        bf16x8 ka[RD], db[RD];          // D phase: K^T and dS^T fragments of key step ks in ring slot ks % RD
        auto rdD = [&](int ks) {
            if (OBTE_SKIP(1) && ks > 2) return;
            ka[ks % RD] = join8(lds_tr(k_tr[0], (16 * D) * (2 * ks)), lds_tr(k_tr[1], (16 * D) * (2 * ks)));
            db[ks % RD] = join8(lds_tr(ds_rd, 128 * ks), lds_tr(ds_rd, 128 * ks + 64));
        };
        bf16x8 cdo[RC], cq[RC];         // C phases: dO^T and Q^T fragments; step gg = 0..15 (C0 then C1) uses group gg % 8 = 4 kk + dt, ring slot gg % RC
        auto rdC = [&](int gg) {
            if (OBTE_SKIP(32) && gg > 1) return;
            const int g = gg & 7, off = (16 * D) * (2 * (g >> 2)) + 512 * (g & 3);
            cdo[gg % RC] = join8(lds_tr(sb.tr[0], S::QB + off), lds_tr(sb.tr[1], S::QB + off));
            cq[gg % RC] = join8(lds_tr(sb.tr[0], off), lds_tr(sb.tr[1], off));
        };
        f32x16 sc1, dp1, dq;
        uint32_t pw0[8], dw0[8], pw1[8], dw1[8];   // P and dS of the two key tiles as bf16 pairs (register pair 2 i, 2 i + 1 -> word i)
        // One element of P = exp2(S' scale2) and dS = P dP' (both chains started from their row constant), computed IN THE SLOT IT
        // IS WRITTEN IN: the empty asm statements make the results opaque there — without them hipcc sinks the arithmetic to its first
        // use, phase C, and the slots meant to hide it run empty (sched_barrier orders instructions, it does not stop IR-level sinking).
        // Two steps, one slot apart: a lone wave has nobody to cover the latency of v_exp_f32, so element r + 1's exponential is
        // issued in the slot that finishes element r.
        auto sm_exp = [&](f32x16& sc, int r) {
            if (OBTE_SKIP(4)) return;
            float pv = fast_exp2(sc[r] * scale2);   // the chain started from -lse / scale: nothing to subtract
            asm volatile("" : "+v"(pv));
            sc[r] = pv;
        };
        auto sm_fin = [&](f32x16& sc, f32x16& dp, uint32_t (&pw)[8], uint32_t (&dw)[8], int r) {
            if (OBTE_SKIP(4)) { if (r & 1) { pw[r >> 1] = __float_as_uint(sc[r]); dw[r >> 1] = __float_as_uint(dp[r]); } return; }
            float ds = sc[r] * dp[r];
            if (r & 1) {
                bf16x2 a = {f2bf(sc[r - 1]), f2bf(sc[r])}, c = {f2bf(dp[r - 1]), f2bf(ds)};
                uint32_t aw = __builtin_bit_cast(uint32_t, a), cw = __builtin_bit_cast(uint32_t, c);
                asm volatile("" : "+v"(aw), "+v"(cw));
                pw[r >> 1] = aw; dw[r >> 1] = cw;
            } else {
                asm volatile("" : "+v"(ds));
                dp[r] = ds;
            }
        };
        // key tile 1: the score chain starts from zero (-inf where the range mask excludes the pair); its row constants -lse log2(e) and -delta
        // arrive four query rows at a time, one chunk ahead of the arithmetic (c4 / d4 [i & 1] = rows 8 i + 4 h .. + 3 = registers 4 i .. 4 i + 3)
        f32x4 c4[2], d4[2];
        auto rd_const1 = [&](int i) {
            c4[i & 1] = lds_f4(sb.st, 256 + 32 * i);
            d4[i & 1] = lds_f4(sb.st, 128 + 32 * i);
        };
        auto sm1_exp = [&](int r) {
            if (OBTE_SKIP(4)) return;
            float pv = fast_exp2(__builtin_fmaf(sc1[r], scale2, c4[(r >> 2) & 1][r & 3]));
            asm volatile("" : "+v"(pv));
            sc1[r] = pv;
        };
        auto sm1_fin = [&](int r) {
            if (OBTE_SKIP(4)) { if (r & 1) { pw1[r >> 1] = __float_as_uint(sc1[r]); dw1[r >> 1] = __float_as_uint(dp1[r]); } return; }
            float ds = sc1[r] * (dp1[r] + d4[(r >> 2) & 1][r & 3]);
            if (r & 1) {
                bf16x2 a = {f2bf(sc1[r - 1]), f2bf(sc1[r])}, c = {f2bf(dp1[r - 1]), f2bf(ds)};
                uint32_t aw = __builtin_bit_cast(uint32_t, a), cw = __builtin_bit_cast(uint32_t, c);
                asm volatile("" : "+v"(aw), "+v"(cw));
                pw1[r >> 1] = aw; dw1[r >> 1] = cw;
            } else {
                asm volatile("" : "+v"(ds));
                dp1[r] = ds;
            }
        };
        // dropout forms (both key tiles): q = the scaled probability under its keep bit (the dV operand), dS = q dP + p (-delta)
        f32x4 d0c[2];                        // key tile 0's -delta chunks (rows 8 i + 4 h .. + 3), read one chunk ahead like tile 1's
        auto rd_const0 = [&](int i) { d0c[i & 1] = lds_f4(sb.st, 128 + 32 * i); };
        auto fin_drop = [&](f32x16& sc, f32x16& dp, uint32_t (&pw)[8], uint32_t (&dw)[8], const uint32_t kword, const float ndelta, int r) {
            const int bit = (r & 3) + 8 * (r >> 2);
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)kword, bit, 1);        // 0 / 0xffffffff
            const float pq = __uint_as_float(__float_as_uint(sc[r] * p.drop.scale) & m);
            float ds = __builtin_fmaf(pq, dp[r], sc[r] * ndelta);
            if (r & 1) {
                bf16x2 a = {f2bf(__uint_as_float(pw[r >> 1])), f2bf(pq)};   // (pw[r >> 1] holds q of the even element: below)
                bf16x2 c = {f2bf(dp[r - 1]), f2bf(ds)};
                uint32_t aw = __builtin_bit_cast(uint32_t, a), cw = __builtin_bit_cast(uint32_t, c);
                asm volatile("" : "+v"(aw), "+v"(cw));
                pw[r >> 1] = aw; dw[r >> 1] = cw;
            } else {
                uint32_t qbits = __float_as_uint(pq);
                asm volatile("" : "+v"(ds), "+v"(qbits));
                dp[r] = ds;
                pw[r >> 1] = qbits;          // parked until the odd element packs the pair
            }
        };
        auto frag_of = [](const uint32_t (&w)[8], int kk) {
            const u32x4 v = {w[4 * kk], w[4 * kk + 1], w[4 * kk + 2], w[4 * kk + 3]};
            return __builtin_bit_cast(bf16x8, v);
        };

        OBTE_PHASE(7);
        OBTE_SB();
        // ---- A0 (its first fragments and row constants were read behind the previous iteration's barrier) ----
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (n + RA - 1 < 8) rdA(sb, n + RA - 1);
            if (n >= 8 - (RD - 1)) rdD(n - (8 - (RD - 1)));   // D's first fragments
            if (n < 4 && take_p) acc_request(t_prev, n);   // (before everything else of this iteration: the A1 wait counts what follows.  Asking for the tile in every
                                                           //  iteration, so that the request is no branch, was measured: + 1.7 % — an LDS-DMA piece costs far more to issue than a branch)
            if (!OBTE_SKIP(128)) {
                if (n >= 4) issue_piece(it + 1, t_next, n - 4);
                if (n == 7) load_stats_issue(t_next * 32);
            }
            if (DROP && n == 7) { kw_next[0] = kw_src[0][(int64_t)t_next * T]; kw_next[1] = kw_src[1][(int64_t)t_next * T]; }
            if (DROP && n == 7) rd_const0(0);
            if (n == 7) poll_issue(t);                // this slice's counter, looked at an iteration from now (after the end-of-iteration wait)
            sc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[n % RA], fk[n % RA], sc0, 0, 0, 0);
            OBTE_SB();
            dp0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[n % RA], vf[0][n], dp0, 0, 0, 0);
            OBTE_SB();
        }
        OBTE_PHASE(0);
        // ---- D (previous slice) beside the softmax arithmetic of key tile 0 ----
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            if (ks + RD - 1 < 16) rdD(ks + RD - 1);
            if (ks >= 16 - (RA - 1)) rdA(sb, 8 + ks - (16 - (RA - 1)));   // A1's first fragments
            if (ks == 0) sm_exp(sc0, 0);
            if (ks + 1 < 16) sm_exp(sc0, ks + 1);
            if (DROP) {
                if ((ks & 3) == 0 && ks / 4 + 1 < 4) rd_const0(ks / 4 + 1);
                fin_drop(sc0, dp0, pw0, dw0, kwh[0], d0c[(ks >> 2) & 1][ks & 3], ks);
            } else {
                sm_fin(sc0, dp0, pw0, dw0, ks);
            }
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks % RD], db[ks % RD], ks == 0 ? zero16 : dq, 0, 0, 0);
            OBTE_SB();
        }
        OBTE_PHASE(1);
        // ---- A1; beside it the previous slice's dQ^T tile leaves: on to the next member of its chain, or (last member) to dqkv ----
#pragma unroll
        for (int n = 8; n < 16; ++n) {
            if (n + RA - 1 < 16) rdA(sb, n + RA - 1);
            if (n >= 16 - (RC - 1)) rdC(n - (16 - (RC - 1)));   // C0's first fragments
            // the tile so far (requested at the top of the iteration; younger than its four pieces: the next slice's four LDS-DMA, its
            // row-constant load and this slice's poll — past the last slice only the poll) joins this workgroup's contribution, then leaves
            if (n == 8 && take_p) {
#ifndef FB_EXP_NOACCWAIT   // (timing experiment only: how long the loop waits here for the tile handed on — results are wrong without the wait)
                if (DROP) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // (dropout: + the two keep-word loads)
#endif
                acc_add(dq);
            }
            if (n >= 9 && n < 13 && !OBTE_SKIP(64)) {
                if (fin_p) acc_park(dq, n - 9);   // the chain ends here: finished in C1
                else store_acc(t_store, dq, n - 9);   // (the first iteration has no previous slice: its four stores go to the trash tile — the loop's counted waits see the same four stores in every iteration)
            }
            if (n == 15) rd_const1(0);
            if (n == 8) { sc1 = zero16; if (!inside) mask_init(sc1, 1, q0); }
            sc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[n % RA], fk[n % RA], sc1, 0, 0, 0);
            OBTE_SB();
            dp1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[n % RA], vf[1][n - 8], n == 8 ? zero16 : dp1, 0, 0, 0);
            OBTE_SB();
        }
        OBTE_PHASE(2);
        // ---- C0: key tile 0's resident accumulators beside the softmax arithmetic of key tile 1; then C1 beside the dS image, the
        //      barrier and the next slice's first reads.  Group g = 4 kk + dt; fragment ring slot g & 1, read two groups ahead.
        char* rowp0 = img_cur + (64 * wave + (lane & 31)) * 8 + h * S::DSG;
        auto ds_words = [&](const uint32_t (&w)[8], int kt, int j) {   // j = 0..3: words 2 j, 2 j + 1 = registers 4 j .. 4 j + 3 = query group 2 j + h
            if (!OBTE_SKIP(64)) *reinterpret_cast<u32x2*>(rowp0 + kt * (32 * 8) + (2 * j) * S::DSG) = u32x2{w[2 * j], w[2 * j + 1]};
        };
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int kk = g >> 2, dt = g & 3;
            rdC(g + RC - 1);   // (the last ones: C1's first steps, the same fragments again)
            if (g == FB_RQ_AT && fin_p && p.rope_cos) rope_request(t_prev, rq);
            if (g >= 4) ds_words(dw0, 0, g - 4);
            if ((g & 1) == 0 && g / 2 + 1 < 4) rd_const1(g / 2 + 1);
            if (g == 0) sm1_exp(0);
            sm1_exp(2 * g + 1);
            if (DROP) fin_drop(sc1, dp1, pw1, dw1, kwh[1], d4[((2 * g) >> 2) & 1][(2 * g) & 3], 2 * g); else sm1_fin(2 * g);
            mfma_acc(dv[0][dt], cdo[g % RC], frag_of(pw0, kk)); OBTE_SB();
            if (2 * g + 2 < 16) sm1_exp(2 * g + 2);
            if (DROP) fin_drop(sc1, dp1, pw1, dw1, kwh[1], d4[((2 * g + 1) >> 2) & 1][(2 * g + 1) & 3], 2 * g + 1); else sm1_fin(2 * g + 1);
            mfma_acc(dk[0][dt], cq[g % RC], frag_of(dw0, kk)); OBTE_SB();
        }
        OBTE_PHASE(3);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int kk = g >> 2, dt = g & 3;
            if (8 + g + RC - 1 < 16) rdC(8 + g + RC - 1);
            if (g < 4) ds_words(dw1, 1, g);
            if (fin_p && !OBTE_SKIP(64)) {
                // (nothing younger than the request is in flight: every use of the entries hangs on this statement's outputs)
                if (g == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rq.c[0]), "+v"(rq.c[1]), "+v"(rq.c[2]), "+v"(rq.c[3]), "+v"(rq.s[0]), "+v"(rq.s[1]), "+v"(rq.s[2]), "+v"(rq.s[3]) :: "memory");
                if ((g & 1) == 0) fin_piece = lds_f4(acc_rd, (g >> 1) * 1024);
            }
            mfma_acc(dv[1][dt], cdo[(8 + g) % RC], frag_of(pw1, kk)); OBTE_SB();
            if (fin_p && (g & 1) == 1 && !OBTE_SKIP(64)) store_final4(t_prev, fin_piece, rq, g >> 1);
            mfma_acc(dk[1][dt], cq[(8 + g) % RC], frag_of(dw1, kk)); OBTE_SB();
        }
        OBTE_PHASE(4);
        // End of the iteration.  Everything this wave issued before this iteration's four tile stores (A1) is done: the next slice's
        // tiles and row constants have landed, this slice's counter has been read — and so are the PREVIOUS iteration's tile stores.
        // Those are what is counted on here: every wave's wait, the barrier, then ONE lane's add (Guideline 16, R1).  This
        // iteration's stores stay in flight (their acknowledgement takes longer than the 40 slots since) and are counted on an
        // iteration from now.  The same barrier publishes the next slice's tiles and this slice's dS image.
        if (!OBTE_SKIP(16)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // (four stores in EVERY iteration: the first one's go to the trash tile)
        if (DROP) { asm volatile("" : "+v"(kw_next[0]), "+v"(kw_next[1])); kw[0] = kw_next[0]; kw[1] = kw_next[1]; }   // (their loads are older than the stores the wait leaves in flight)
        st_l = *raw_st;          // (behind the wait: what the two LDS-DMA loads of this iteration left in this wave's raw area)
        have_cur = *raw_poll;
        store_stats(stage_of(it + 1), t_next * 32);
        if (!OBTE_SKIP(8)) __syncthreads();
        if (tid == 0 && t_sig >= 0 && t_sig != mute) __hip_atomic_fetch_add(flag_b + t_sig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        OBTE_PHASE(5);
        t_sig = (!first_it && !last_p) ? t_prev : -1;   // (nobody follows a chain's last member)
        t_prev = t; info_prev = info; have_prev = have_cur;
    }
#ifdef OBTE_DEBUG_HOOKS
    const unsigned long long t_loop_end = tlast;
#endif
#undef OBTE_PHASE
#undef OBTE_SB
#undef OBTE_SKIP
    if (n_sl > 0) {   // the last slice visited: its dQ^T tile (its dS image was completed by the loop's last barrier), handed on the same way
        const char* img = dsimg + ((n_sl - 1) & 1) * S::DSB;
        const bool take_p = (info_prev & 0x7f) > 0;
        if (take_p) {
            wait_turn(t_prev, info_prev & 0x7f, have_prev);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc_request(t_prev, i);
        }
        f32x16 dq = zero16;
#pragma unroll
        for (int ks = 0; ks < FB_KEYS / 16; ++ks)
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<D>(Kblk, 16 * ks, wave, lane), ds_frag<D>(img, 16 * ks, lane), dq, 0, 0, 0);
        if (take_p) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc_add(dq); }
        if (FIN_IN_LOOP && (info_prev & 0x80)) {
            RopeQ rq;
            if (p.rope_cos) load_rope(t_prev, rq);
#pragma unroll
            for (int i = 0; i < 4; ++i) store_final(t_prev, dq, rq, i);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) store_acc(t_prev, dq, i);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // every wave's tiles have left; and every wave is done with the K rows (all 256 feed each wave's dQ tiles): they now carry rows out
    if (tid == 0) {
        if (t_sig >= 0 && t_sig != mute) __hip_atomic_fetch_add(flag_b + t_sig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n_sl > 0 && !(info_prev & 0x80) && t_prev != mute) __hip_atomic_fetch_add(flag_b + t_prev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef OBTE_DEBUG_HOOKS
    unsigned long long t_a = 0, t_b = 0;   // (OBTE_ATTN_SKIP=1 / 2 / 3 with the stamps: slot 6 = prologue / + last slice and signals / + dK, dV rows)
    if (p.dbg_times) asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_a) :: "memory");
#endif
    {   // dV and dK rows leave through the wave's own K rows in LDS as whole 256-byte rows (wave_rows_out); dK rotated back
        char* wl = Kblk + wave * (64 * 2 * D);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int64_t key0 = (int64_t)key[kt] - (lane & 31);
            const int rows_ok = (int)max((int64_t)0, min((int64_t)32, (int64_t)T - key0));
            bf16* dk0 = p.dqkv + (b * T + key0) * ld + C + hd * D;
            RopeRow<D> rr;
            rr.load(p.rope_cos, p.rope_sin, k_ok[kt] ? key[kt] : T - 1, h);
            bf16x4 gb[ND * 4];
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) gb[4 * dt + i][j] = f2bf(dv[kt][dt][4 * i + j]);
            wave_rows_out<D>(wl, gb, dk0 + C, ld, rows_ok, lane);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float g[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = dk[kt][dt][4 * i + j] * p.scale;
                    rr.apply(g, dt, i);
#pragma unroll
                    for (int j = 0; j < 4; ++j) gb[4 * dt + i][j] = f2bf(g[j]);
                }
            wave_rows_out<D>(wl + 32 * 2 * D, gb, dk0, ld, rows_ok, lane);
        }
    }

#ifdef OBTE_DEBUG_HOOKS
    if (p.dbg_times) asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_b) :: "memory");
#endif
    // The slices whose chain this workgroup ENDED hold their complete fp32 sums, written by this very wave (same lanes, same
    // addresses): softmax scale, inverse RoPE, one rounding to bf16.  Done here and not in the loop, where the rotation-table
    // entries would hold 16 registers through its tightest phase — and AFTER the dK / dV rows have left, so that the 256
    // accumulator registers are free and eight tiles (with their rotation entries) travel per dependent round trip.
    if (!FIN_IN_LOOP) {
        constexpr int FIN = 8;
        int* lastlist = kbb;   // (the key blocks' ranges are no longer needed) compact list of those slices, in visiting order
        int n_last = 0;
        for (int i = tid; i < n_sl; i += FB_NW * 64)
            if (tab[i] & 0x80) {
                int rank = 0;
                for (int j = 0; j < i; ++j) rank += (tab[j] >> 7);
                lastlist[rank] = slice_at(i);
            }
        for (int j = 0; j < n_sl; ++j) n_last += (tab[j] >> 7);   // (every thread: uniform)
        n_last = __builtin_amdgcn_readfirstlane(n_last);
        __syncthreads();
        for (int base = 0; base < n_last; base += FIN) {
            // (FIN tiles per round trip: the later tiles' loads are in flight while the first one's statement waits)
            AccRegs r[FIN];
            RopeQ rq[FIN];
#pragma unroll
            for (int j = 0; j < FIN; ++j)
                if (base + j < n_last) {
                    const int t = __builtin_amdgcn_readfirstlane(lastlist[base + j]);
                    if (p.rope_cos) load_rope(t, rq[j]);
                    load_acc_sync(t, r[j]);
                }
#pragma unroll
            for (int j = 0; j < FIN; ++j)
                if (base + j < n_last) {
                    const int t = __builtin_amdgcn_readfirstlane(lastlist[base + j]);
                    f32x16 dq;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) dq[4 * i + e] = r[j].x[i][e];
#pragma unroll
                    for (int i = 0; i < 4; ++i) store_final(t, dq, rq[j], i);
                }
        }
    }
#ifdef OBTE_DEBUG_HOOKS
    if (p.dbg_times && tid == 0) {
        unsigned long long t_exit;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_exit) :: "memory");
        if (p.dbg_skip == 2) tsum[6] += t_a - t_loop_end;
        else if (p.dbg_skip == 3) tsum[6] += t_b - t_loop_end;
        else if (p.dbg_skip != 1) tsum[6] += t_exit - t_loop_end;
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) p.dbg_times[(size_t)blockIdx.x * 8 + k2] = tsum[k2];
    }
#endif
}

template __global__ void attn_bwd_fused_kernel<128, MASK_NONE>(FusedParams);
template __global__ void attn_bwd_fused_kernel<128, MASK_RANGES>(FusedParams);
template __global__ void attn_bwd_fused_kernel<128, MASK_NONE, 0, true>(FusedParams);
template __global__ void attn_bwd_fused_kernel<128, MASK_RANGES, 0, true>(FusedParams);
#ifdef OBTE_DEBUG_HOOKS
#define OBTE_FUSED_SKIPS(X) X(4) X(35) X(39) X(64) X(103) X(128) X(231) X(255)
#define X(m) template __global__ void attn_bwd_fused_kernel<128, MASK_RANGES, m>(FusedParams);
OBTE_FUSED_SKIPS(X)
#undef X
#endif

}  // namespace

namespace obte_attn {

// scratch: the running dQ^T tiles, the slices' counters, the key blocks' slice ranges
static void ws_layout(int64_t B, int64_t T, int H, int64_t& nkb, int64_t& nsl, int64_t& o_flags, int64_t& o_bounds, int64_t& total) {
    nkb = (T + FB_KEYS - 1) / FB_KEYS; nsl = (T + 31) / 32;
    auto up = [](int64_t x) { return (x + 255) & ~int64_t(255); };
    o_flags = up(B * H * (nsl + 1) * (32 * 128 * 4));   // per (batch, head): one tile per slice + a trash tile (the first iteration's stores, which stand for no slice)
    o_bounds = o_flags + up(B * H * nsl * 4);
    total = o_bounds + up(B * nkb * 2 * 4) + 256;
}
// 0: the one-kernel form does not apply to this shape — longer sequences take the two-kernel form.  (256 slices = T <= 8192: the list
// of the slices a workgroup finishes is kept in the 1 KiB of LDS the key blocks' ranges occupied, one int per slice, and a mask can
// make one key block finish all of its slices)
int64_t fused_bwd_ws_bytes(int64_t B, int64_t T, int H) {
    int64_t nkb, nsl, a, b2, total;
    ws_layout(B, T, H, nkb, nsl, a, b2, total);
    return (nsl <= 256 && nsl <= FusedShape<128>::TAB && nkb <= 120) ? total : 0;
}

// mode: MASK_NONE or MASK_RANGES.  ws: fused_bwd_ws_bytes() bytes.
int launch_bwd_fused(const AttnParams& p, int mode, void* ws, hipStream_t st) {
    FusedParams fp;
    fp.a = p;
    int64_t nkb, nsl, o_flags, o_bounds, total;
    ws_layout(p.B, p.T, p.H, nkb, nsl, o_flags, o_bounds, total);
    fp.nkb = (int)nkb; fp.nsl = (int)nsl;
    char* w = reinterpret_cast<char*>(ws);
    fp.dq_acc = reinterpret_cast<float*>(w);
    fp.flags = reinterpret_cast<int32_t*>(w + o_flags);
    fp.kb_bounds = reinterpret_cast<int32_t*>(w + o_bounds);
    fp.status = obte_status_word();
    if (!fp.status) { obte_set_error("obte_attn_bwd: the device status word could not be allocated"); return OBTE_ELAUNCH; }
    const bool inject = obte_fault_injection() == 1;
    fp.spin_limit = inject ? (1 << 10) : (1 << 20);
    fp.no_signal_slice = inject ? 0 : -1;
    // (delta already formed by the producer of dO — the row-dot epilogue of the projection's input gradient, csrc/block.cpp: the prep launch
    //  then only zeroes the counters and derives the slice ranges)
    const int nb_delta = p.delta_ready ? 0 : (int)cdiv64(p.B * p.T, 4), nb_flags = (int)cdiv64(p.B * p.H * nsl, 256);
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)(nb_delta + nb_flags + p.B * nkb)), dim3(256), 0, st, fp, mode, nb_delta, nb_flags);
    OBTE_CHECK_LAUNCH("obte_attn_bwd(prep)");
    const int smem = FusedShape<128>::SMEM;
    const dim3 grid((unsigned)(fp.nkb * p.H * p.B)), block(FB_NW * 64);
    if (p.drop.thresh16 != 0) {   // dropout: the forward's keep bits (the caller checked they are there)
        if (mode == MASK_NONE) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<128, MASK_NONE, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            hipLaunchKernelGGL((attn_bwd_fused_kernel<128, MASK_NONE, 0, true>), grid, block, smem, st, fp);
        } else {
            (void)hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<128, MASK_RANGES, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            hipLaunchKernelGGL((attn_bwd_fused_kernel<128, MASK_RANGES, 0, true>), grid, block, smem, st, fp);
        }
    } else if (mode == MASK_NONE) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<128, MASK_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipLaunchKernelGGL((attn_bwd_fused_kernel<128, MASK_NONE>), grid, block, smem, st, fp);
    } else {
        bool done = false;
#ifdef OBTE_DEBUG_HOOKS
#define X(m) if (p.dbg_skip == m) { (void)hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<128, MASK_RANGES, m>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); \
                                    hipLaunchKernelGGL((attn_bwd_fused_kernel<128, MASK_RANGES, m>), grid, block, smem, st, fp); done = true; }
        OBTE_FUSED_SKIPS(X)
#undef X
#endif
        if (!done) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<128, MASK_RANGES>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            hipLaunchKernelGGL((attn_bwd_fused_kernel<128, MASK_RANGES>), grid, block, smem, st, fp);
        }
    }
    OBTE_CHECK_LAUNCH("obte_attn_bwd(fused)");
#ifdef OBTE_DEBUG_HOOKS
    if (p.dbg_times) {
        const int n = (int)grid.x;
        std::vector<unsigned long long> hbuf((size_t)n * 8);
        if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(hbuf.data(), p.dbg_times, hbuf.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < n; ++i) for (int k = 0; k < 8; ++k) sum[k] += (double)hbuf[(size_t)i * 8 + k];
            const char* names[8] = {"A0", "D+SM0", "A1", "C0+SM1", "C1", "wait+barrier+signal", "OUTSIDE the loop (prologue + last slice + rows out + finishing; per slice share)", "top (tile request, first reads)"};
            double tot = 0; for (int k = 0; k < 8; ++k) tot += sum[k];
            fprintf(stderr, "[attn fused phases, cycles per slice per workgroup (%d slices)]", fp.nsl);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.0f (%.0f%%)", names[k], sum[k] / n / fp.nsl, 100.0 * sum[k] / tot);
            fprintf(stderr, " | total %.0f\n", tot / n / fp.nsl);
        }
    }
#endif
    return OBTE_OK;
}

}  // namespace obte_attn
