// Device helpers shared by the attention kernels (attention.hip: forward, dQ, dK/dV; attention_bwd_fused.hip: the one-kernel
// backward): parameter block, LDS tile image + fragment reads, LDS-DMA staging, workgroup remap, row epilogues.
#pragma once
#include "common.h"
#include <string.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

namespace obte_attn {


constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

enum { MASK_NONE = 0, MASK_RANGES = 1, MASK_DENSE = 2 };

struct AttnParams {
    const bf16* qkv; bf16* o; float* lse;                       // forward
    const bf16* o_in; const bf16* d_o; const float* lse_in; float* delta; bf16* dqkv;   // backward
    const float* rope_cos; const float* rope_sin;               // backward: inverse RoPE on dq, dk (nullable)
    const int32_t* key_ranges; const bf16* mask; int64_t mask_sb, mask_sh, mask_sq;
    const int32_t* query_bounds;   // per-key [first, last+1) query bounds (obte_mask_bounds), nullable: loop bounds in dense mode,
                                   // the exact query range of every key in range mode (else the mask is taken to be symmetric)
    const int32_t* gate;           // the *_gated_kernel entry points: device flag (obte_mask_bounds' ranges_exact), 1 = run the range-mode body
    int64_t B, T; int H; float scale;
    // The QUERY side (forward, dQ and dK/dV kernels of attention.hip; the one-kernel backward does not use these).  Normally the
    // queries are the T rows of every batch element inside qkv.  With q_off they are a separate, gathered set of rows per batch
    // element (the last block of a masked-LM step needs its attention output at the masked positions only: csrc/block.cpp):
    // batch element b owns q-side rows q_off[b] .. q_off[b + 1] - 1 of q_src / o / o_in / d_o / dq_dst and of key_ranges; lse and
    // delta hold entry (head, q-side row r) at head * stat_hs + r.  Keys and values are always the T rows of qkv.
    const int32_t* q_off;            // [B + 1], or null
    const int32_t* q_blk_off;        // with q_off, the query-major kernels: [B + 1] first 256-query block of every batch element in the compact
                                     // grid (the grid holds only blocks that have queries: an early-exit workgroup per empty block cost 3x the kernel)
    const bf16* q_src; int64_t q_ld; // Q rows (head h at column h * D): qkv, 3C without q_off
    bf16* dq_dst; int64_t dq_ld;     // dQ rows: dqkv, 3C without q_off
    const int32_t* q_pos;            // with q_off: the sequence position of every q-side row (inverse RoPE of dQ); null: the row's own index
    int64_t stat_hs;                 // lse / delta stride between heads: T without q_off (entry ((b H + head) T + q)), the row count with it
    DropCfg drop;   // attention-probability dropout (site 1); thresh16 == 0: off
    uint32_t* drop_bits_out;         // forward, nullable: keep bits in key-major order, uint32 [B*H][ceil(T/32)][T] (bit i = query 32 t + i)
    const uint32_t* drop_bits_in;    // backward, nullable: the same buffer
    int delta_ready;   // backward: p.delta already holds rowsum(dO o O) (the one-kernel form's prep launch skips it; the kernel pair forms its own)
    int no_wait;    // timing-only diagnostic (OBTE_ATTN_DEBUG=nowait): the tile loops do not wait for their LDS-DMA (results are wrong)
    int max_tiles;  // timing-only diagnostic (OBTE_ATTN_DEBUG=tiles:N): every workgroup stops after N tiles (results are wrong; 0 = off)
    int dbg_skip;   // timing-only diagnostic, debug build (OBTE_ATTN_SKIP=bits): dK/dV kernel — 1: no softmax arithmetic, 2: no phase-C MFMAs,
                    // 4: no phase-A MFMAs, 8: no per-tile barrier (results are wrong)
    unsigned long long* dbg_times;   // debug build (OBTE_ATTN_TIMES=1): per workgroup, s_memrealtime at entry / loop start / loop end / stores issued / stores done
};
// the query side of batch element b, head hd: first q-side row, number of queries, first lse / delta entry
struct QSide { int64_t row0; int n; int64_t stat0; };
__device__ __forceinline__ QSide q_side(const AttnParams& p, int64_t b, int hd) {
    QSide q;
    if (p.q_off) {
        const int o = __builtin_amdgcn_readfirstlane(p.q_off[b]), e = __builtin_amdgcn_readfirstlane(p.q_off[b + 1]);
        q.row0 = o; q.n = e - o; q.stat0 = (int64_t)hd * p.stat_hs + o;
    } else {
        q.row0 = b * p.T; q.n = (int)p.T; q.stat0 = (b * p.H + hd) * p.T;
    }
    return q;
}

#ifdef OBTE_DEBUG_HOOKS
#define OBTE_STAMP(p, k) do { if ((p).dbg_times && threadIdx.x == 0) (p).dbg_times[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define OBTE_STAMP(p, k) do { } while (0)
#endif

// LDS image of a [rows][D] bf16 tile: groups of 8 rows (16*D bytes), each cut into 8-row x 32-column subtiles of 512 B whose
// 64-B rows hold their four 16-B chunks XOR-ed with row bits 2-3 (cdna_hip_programming.md T10, image (a)).  Both kinds of
// fragment read are bank-conflict free, and — unlike plain rows with a 4-bit XOR — every read of a kernel is ONE of two
// lane-dependent base offsets plus an immediate: the address arithmetic leaves the tile loops and ~8 VGPRs with it.
template <int D>
__device__ __forceinline__ int swz(int row, int ch) {
    return (16 * D) * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

// A/B fragment of v_mfma_f32_32x32x16_bf16 read by rows: lane l -> tile row row0 + (l&31), k = 16s + 8(l>>5) + j.
// row0 % 32 == 0.  = tile + swz(row0 + (l&31), 2s + (l>>5)), written as lane base (two values: s even / odd) + immediate.
template <int D>
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int row0, int s, int lane) {
    const int r = lane & 31;
    const int base = (16 * D) * (r >> 3) + 64 * (r & 7) + 16 * ((((lane >> 5) ^ (r >> 2)) & 3) ^ (2 * (s & 1)));
    return *reinterpret_cast<const bf16x8*>(tile + base + (16 * D) * (row0 >> 3) + 512 * (s >> 1));
}

// A fragment of the TRANSPOSED tile: MFMA row = tile column 32*dt + (l&31), MFMA k element j = tile row
// krow0 + 8(j>>2) + 4(l>>5) + (j&3)  — the row order in which a 32x32 f32 accumulator, converted in place,
// serves as the other operand (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand").
// krow0 % 16 == 0.  Lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of a 4 x 16 block:
// = tile + swz(krow0 + 4(l>>5) + q [+ 8], 4dt + 2((l>>4)&1) + (p>>1)) + 8(p&1); the two reads differ by the bit-5 flip.
template <int D>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int krow0, int dt, int lane) {
    const int li = lane & 15, h = lane >> 5;
    const int c2 = 2 * ((lane >> 4) & 1) + ((li & 3) >> 1);
    const int base = 64 * (4 * h + (li >> 2)) + 16 * (c2 ^ h) + (li & 1) * 8;
    const char* t = tile + (16 * D) * (krow0 >> 3) + 512 * dt;
    const bf16x4 lo = lds_read_tr16(t + base);
    const bf16x4 hi = lds_read_tr16(t + (base ^ 32) + 16 * D);
    return join8(lo, hi);
}

// Global -> LDS staging of a ROWS x D tile by LDS-DMA (buffer_load ... lds, 16 B per lane, no VGPR round trip).
// The DMA writes LDS linearly (wave base + lane*16), so the image is produced by choosing the SOURCE chunk each lane fetches:
// LDS byte o = 1024*piece + 16*lane belongs to row 8*(o / 16D) + (o % 512)/64 and holds logical chunk
// 4*((o % 16D)/512) + (((o % 64)/16) ^ ((row>>2)&3)) of it — exactly what swz<D>() reads back.  Per 1-KiB piece the wave
// fetches 8 rows x 128 contiguous bytes (whole cache lines).  The descriptor ends at the end of this batch element's rows,
// so rows past T arrive as zeros.  NW waves; wave w issues pieces w, w+NW, ...  Completion: the issuing wave's
// s_waitcnt vmcnt, then the workgroup barrier.
template <int D, int ROWS, int NW = 4>
struct TileDma {
    static constexpr int PIECES = ROWS * 2 * D / 1024;   // 1-KiB pieces in the tile
    static constexpr int NP = PIECES >= NW ? PIECES / NW : 1;   // per wave (a tile with fewer pieces than waves: the first PIECES waves)
    static_assert(PIECES >= NW ? NP * NW == PIECES : true, "tile does not split evenly over the waves");
    static constexpr int PPG = D / 64;                    // pieces per 8-row group: 2 (D=128) or 1 (D=64)
    int voff[NP];
    __device__ __forceinline__ void init(int wave, int lane, int64_t row_stride) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int piece = wave + NW * i;
            const int row = 8 * (piece / PPG) + ((lane >> 2) & 7);
            const int chunk = 4 * (2 * (piece % PPG) + (lane >> 5)) + ((lane & 3) ^ ((row >> 2) & 3));
            voff[i] = (int)((row * row_stride + chunk * 8) * 2);
        }
    }
    // g: address of the tile's first row (head column applied); bytes_left: bytes from g to the end of the rows that
    // may be read (the batch element's last row)
    __device__ __forceinline__ void issue(const bf16* g, int64_t bytes_left, char* tile, int wave) const {
#ifdef OBTE_DMA_BUILTIN
        __amdgpu_buffer_rsrc_t rsrc = make_rsrc(g, bytes_left);
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (PIECES >= NW || wave < PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(tile + (wave + NW * i) * 1024), 16, voff[i], 0, 0, 0);
#else
        // through inline asm (common.h lds_dma16): with the builtin hipcc drains vmcnt in front of the next LDS read,
        // i.e. waits for the tile it was asked to prefetch before computing on the current one
        const i32x4_t rsrc = make_rsrc_words(g, bytes_left);
        const uint32_t base = lds_addr_of(tile) + wave * 1024;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (PIECES >= NW || wave < PIECES) lds_dma16(rsrc, base + NW * i * 1024, voff[i]);
#endif
    }
};

__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Before the main loop: the builtin form, so that hipcc's own bookkeeping also knows every global load of the prologue
// (the Q / dO / K fragments held in registers) has landed.  Otherwise it keeps its per-fragment `s_waitcnt vmcnt(7..0)`
// in front of the first MFMAs INSIDE the loop — harmless for its own loads after the first iteration, but vmcnt counts
// the asm-issued LDS-DMA too, so those waits drained the next tile's prefetch during the first MFMA phase of every tile.
// wait until at most k of this wave's vector-memory operations (the LDS-DMA pieces) are still in flight; k uniform
__device__ __forceinline__ void dma_wait_leave(int k) {
    switch (k) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void prologue_wait_all() { __builtin_amdgcn_s_waitcnt(0x0070); }

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Four consecutive additive-mask values of one query row (keys k0..k0+3), times log2(e).  A lane owns a query row and its
// 16 accumulator registers cover four runs of four consecutive keys, so the dense mask is read as four 8-byte loads per
// 32-key half instead of sixteen 2-byte ones (each of which touches 32 to 64 different cache lines per wave).
// vec: base pointer and all strides are multiples of 4 elements (uniform, checked once per kernel).
__device__ __forceinline__ void mask4(const bf16* mrow, int k0, int T, bool vec, float (&out)[4]) {
    if (vec) {
        bf16x4 v = {};
        if (k0 < T) v = *reinterpret_cast<const bf16x4*>(mrow + k0);   // T % 4 == 0 here: the run is inside or outside as a whole
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = bf2f(v[j]) * LOG2E;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = k0 + j < T ? bf2f(mrow[k0 + j]) * LOG2E : 0.f;
    }
}
__device__ __forceinline__ bool mask_vec_ok(const AttnParams& p) {
    return ((reinterpret_cast<uintptr_t>(p.mask) & 7) == 0) && (p.mask_sb % 4 == 0) && (p.mask_sh % 4 == 0) && (p.mask_sq % 4 == 0) && (p.T % 4 == 0);
}

// Workgroup -> (row block, head, batch).  The grid is one-dimensional and remapped so that the workgroups an XCD
// receives (block id % 8) cover a CONTIGUOUS range of work ids, with the row blocks of one (batch, head) adjacent: the
// blocks that stream the same K/V (or Q/dO) through LDS then share one L2 instead of pulling it into up to four.
struct BlockId { int blk, hd; int64_t b; };
// the compact grid of a gathered query set: workgroup w -> head w % H, block index w / H located in q_blk_off; blk < 0: nothing to do
__device__ __forceinline__ BlockId block_id_rows(const AttnParams& p) {
    const int w = blockIdx.x, H = p.H;
    const int idx = w / H;
    int lo = 0, hi = (int)p.B;            // last b with q_blk_off[b] <= idx
    if (idx >= p.q_blk_off[p.B]) return BlockId{-1, 0, 0};
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (p.q_blk_off[mid] <= idx) lo = mid; else hi = mid;
    }
    return BlockId{idx - p.q_blk_off[lo], w % H, (int64_t)lo};
}
__device__ __forceinline__ BlockId block_id(int nblk, int H) {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int pair = w / nblk;
    return BlockId{w % nblk, pair % H, (int64_t)(pair / H)};
}

// accumulator register -> row index inside the 32x32 tile, for lane half h
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int base) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f2bf(v[base + j]);
    return r;
}

// min/max over the workgroup (256 threads) through a small LDS scratch of 8 ints
template <int NW = 4>
__device__ __forceinline__ void block_minmax(int& lo, int& hi, int* scratch, int wave, int lane) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, 64));
        hi = max(hi, __shfl_xor(hi, o, 64));
    }
    if (lane == 0) { scratch[wave] = lo; scratch[NW + wave] = hi; }
    __syncthreads();
    lo = scratch[0]; hi = scratch[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) { lo = min(lo, scratch[w]); hi = max(hi, scratch[NW + w]); }
    __syncthreads();
    // the same value in every lane, and hipcc must know it: tile indices derived from these feed scalar operands
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
}


// Output / gradient rows leave through LDS.  The accumulators hold, per lane, 4 consecutive columns of ONE row (row = lane & 31):
// stored as they lie that is 8-byte pieces of 32 different rows per wave instruction — partial lines, 4.7 us of store issue in the
// dK/dV epilogue (s_memrealtime stamps, debug build).  Each wave instead writes its 32 x D block into a wave-private LDS region
// (16-byte chunk XOR row) and reads it back as whole rows: 16-byte pieces, four full 256-byte rows (D = 128) per store instruction.
// v[c] = the lane's columns 8 c + 4 h .. + 3; rows >= rows_ok (past T) are not stored.  No barrier: the region is the wave's own.
template <int D>
__device__ __forceinline__ void wave_rows_out(char* wl, const bf16x4 (&v)[D / 8], bf16* g0, int64_t ld, int rows_ok, int lane) {
    constexpr int NCH = D / 8;   // 16-byte chunks per row
    const int row = lane & 31, h = lane >> 5;
#pragma unroll
    for (int c = 0; c < NCH; ++c) *reinterpret_cast<bf16x4*>(wl + row * (2 * D) + ((c ^ (row & (NCH - 1))) * 16) + 8 * h) = v[c];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ps = 0; ps < (32 * NCH) / 64; ++ps) {
        const int idx = ps * 64 + lane, r = idx / NCH, c = idx % NCH;
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(wl + r * (2 * D) + ((c ^ (r & (NCH - 1))) * 16));
        if (r < rows_ok) *reinterpret_cast<bf16x8*>(g0 + (int64_t)r * ld + 8 * c) = x;
    }
}


// inverse RoPE on a gradient row at position t, four consecutive head-dim elements (d0 = 32 dt + 8 i + 4 h) at a time.  The
// rotation-table entries of the WHOLE row are loaded first (load()), before the first gradient store of the epilogue: vmcnt
// retires in issue order, so a table load issued behind a store waits for that store's acknowledgement — the former epilogue
// (load cos, wait, load sin, wait, rotate, store, sixteen times over) cost ~1 us per group, ~15 us per kernel.
template <int D>
struct RopeRow {
    float2 c[D / 8], s[D / 8];
    bool on;
    __device__ __forceinline__ void load(const float* cos_t, const float* sin_t, int64_t t, int h) {
        on = cos_t != nullptr;
        if (!on) return;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t at = t * (D / 2) + (32 * dt + 8 * i + 4 * h) / 2;
                c[4 * dt + i] = *reinterpret_cast<const float2*>(cos_t + at);
                s[4 * dt + i] = *reinterpret_cast<const float2*>(sin_t + at);
            }
    }
    __device__ __forceinline__ void apply(float (&g)[4], int dt, int i) const {
        if (!on) return;
        const float2 cc = c[4 * dt + i], ss = s[4 * dt + i];
        const float e0 = g[0] * cc.x + g[1] * ss.x, o0 = -g[0] * ss.x + g[1] * cc.x;
        const float e1 = g[2] * cc.y + g[3] * ss.y, o1 = -g[2] * ss.y + g[3] * cc.y;
        g[0] = e0; g[1] = o0; g[2] = e1; g[3] = o1;
    }
};


// attention_bwd_fused.hip: the one-kernel backward (head_dim 128, MASK_NONE / MASK_RANGES, no dropout)
int64_t fused_bwd_ws_bytes(int64_t B, int64_t T, int H);
int launch_bwd_fused(const AttnParams& p, int mode, void* ws, hipStream_t st);

}  // namespace obte_attn
