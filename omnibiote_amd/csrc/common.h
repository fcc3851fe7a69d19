// Shared device/host helpers for the gfx950 kernels of libomnibiote_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/omnibiote_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define WAVE 64

// ---- error plumbing ------------------------------------------------------------------------------------
void obte_set_error(const char* fmt, ...);
#define OBTE_REQUIRE(cond, ...)                     \
    do {                                            \
        if (!(cond)) {                              \
            obte_set_error(__VA_ARGS__);            \
            return OBTE_EINVAL;                     \
        }                                           \
    } while (0)
#define OBTE_CHECK_LAUNCH(name)                                                           \
    do {                                                                                  \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ != hipSuccess) {                                                           \
            obte_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));         \
            return OBTE_ELAUNCH;                                                          \
        }                                                                                 \
    } while (0)

// ---- GEMM with a row-dot epilogue (library-internal: the block backward's d(attention output) = dx1 W_proj, structure 7 only) -----------
// D = A B as OBTE_EPI_NONE (dy W layout: A k-contiguous, B not), and for every row m = b T + t and every head h (head_dim 128 columns):
// rowdot[(b H + h) T + t] = sum over the head's columns of D[m, c] * other[m, c] (both as stored, bf16; fp32 sums) — the softmax
// backward's delta = rowsum(dO o O), which the attention backward's prep launch otherwise forms by reading both tensors again.
// Returns OBTE_OK, an error, or 1: this shape does not take structure 7 (nothing was launched: run the plain product instead).
#define OBTE_EPI_ROWDOT 7
extern "C" int obte_gemm_rowdot_bf16(const obte_gemm_args* g, const obte_bf16* other, float* rowdot, int64_t T, int32_t head_dim, obte_stream s);   // (C linkage: tests/ call it directly)

int obte_attn_bwd_delta_ready(const obte_attn_bwd_args* a, obte_stream s);   // obte_attn_bwd with a->delta = rowsum(dO o O) already formed (the row-dot epilogue above)

// ---- attention with the queries at listed rows only (library-internal: csrc/block.cpp's rows form calls it; attention.hip) -----------
// The last block of a masked-LM step needs its attention output at the masked positions alone: the queries are a gathered set of
// rows per batch element (ascending positions), keys and values stay the T rows of qkv.  Everything on the query side is
// indexed by the gathered rows, n in all: q [n, C] (gathered rows of qkv's q third), o / d_o / dq [n, C], lse / delta [H][n].
struct obte_attn_rows {
    const int32_t* q_off;        // [B + 1]: batch element b owns gathered rows q_off[b] .. q_off[b + 1] - 1
    const int32_t* q_blk_off;    // [B + 1]: its first 256-row block in the compact grid of the query-major kernels
    const int32_t* q_pos;        // [n]: sequence position of every gathered row
    const int32_t* key_ranges;   // [n][2] key range of every gathered row, or null (no mask)
    const int32_t* query_bounds; // [B T][2]: for every key the gathered rows (indices within its batch element) that see it; null with no mask
    int64_t n;
};
// obte_attn_rows_prep fills the five arrays from the ascending row list (global rows b T + position) and the full-size key ranges
// (null: no mask; the masks are symmetric, SURVEY fact 5: a key's queries are the positions of its own range) and, for the
// backward's scatter, inv [B T]: the gathered index of every row or -1.
int obte_attn_rows_prep(const int64_t* rows, int64_t n, int64_t B, int64_t T, const int32_t* key_ranges_full, int32_t* q_off, int32_t* q_blk_off,
                        int32_t* q_pos, int32_t* key_ranges_rows, int32_t* query_bounds_rows, int32_t* inv, obte_stream s);
int obte_attn_fwd_rows(const obte_attn_fwd_args* a, const obte_attn_rows* r, const obte_bf16* q, obte_stream s);   // a->o, a->lse: gathered
int obte_attn_bwd_rows(const obte_attn_bwd_args* a, const obte_attn_rows* r, const obte_bf16* q, obte_bf16* dq, obte_stream s);   // a->o, d_o, lse, delta: gathered; a->dqkv: dK, dV thirds
// dst[m, 0:cols] (row stride ld) = src[inv[m]] (cols wide, dense) or zeros where inv[m] < 0; and the strided gather dst[i] = src[rows[i], 0:cols]
int obte_rows_fill_strided_bf16(const obte_bf16* src, const int32_t* inv, obte_bf16* dst, int64_t total_rows, int64_t ld, int32_t cols, obte_stream s);
// out[i] = (aux ? aux[i] : 0) + dropout(in[i]) on gathered rows: the mask element of (i, c) is (rows[i], c) of the whole activation
int obte_dropout_rows_bf16(const obte_bf16* in, const obte_bf16* aux, obte_bf16* out, const int64_t* rows, int64_t n_rows, int32_t cols, float p,
                           uint64_t seed, int32_t site, obte_stream s);
// RoPE in place on columns [0, ncols) of x (row stride ld; head boundaries aligned), the row's position pos[row] or row % T; dst[rows[i]] += src[i]
int obte_rope_cols_bf16(obte_bf16* x, int64_t ld, int32_t ncols, const float* cos_t, const float* sin_t, int64_t rows, int64_t T, const int32_t* pos,
                        int32_t head_dim, obte_stream s);
int obte_rows_add_bf16(const obte_bf16* src, const int64_t* rows, obte_bf16* dst, int64_t n_rows, int32_t cols, obte_stream s);
int obte_rows_gather_strided_bf16(const obte_bf16* src, int64_t ld, const int64_t* rows, obte_bf16* dst, int64_t n_rows, int32_t cols, obte_stream s);

// device status word (lib.cpp): pinned host memory kernels OR failure bits into; null if it could not be allocated
int32_t* obte_status_word();
int obte_fault_injection();   // the tests' fault-injection request (obte_fault_inject), 0 = none

// opt-in launch profiler (lib.cpp); idx < 0 = profiling off
int obte_prof_begin(hipStream_t st, int kind, int64_t d0, int64_t d1, int64_t d2);
void obte_prof_end(int idx, hipStream_t st);

// ---- bf16 <-> f32 -----------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }  // v_cvt_pk_bf16_f32: RNE, NaN-preserving

// ---- wave reductions (64 lanes) ----------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// The same sum through DPP (quad swaps, the two mirrors of a row of 16, row_bcast:15 / :31 of the wave64 form, lane 63 read back):
// seven ~8-cycle steps instead of six ds_bpermute round trips of ~64+ cycles each — in a one-row-per-wave kernel those round trips sit
// on the path between the row's load and its store.  Another summation order than wave_sum: equal to fp32 rounding, not bitwise.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_moved(float v) {   // lanes outside ROW_MASK (and sources outside the wave) read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_moved<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_moved<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_moved<0x141>(v);         // row_half_mirror
    v += dpp_moved<0x140>(v);         // row_mirror: every lane holds the sum of its row of 16
    v += dpp_moved<0x142, 0xa>(v);    // row_bcast:15 -> rows 1 and 3 add the row before them
    v += dpp_moved<0x143, 0xc>(v);    // row_bcast:31 -> rows 2 and 3 add lanes 0..31's total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- buffer resources --------------------------------------------------------------------------------------
// Raw buffer descriptor over [base, base+bytes): out-of-range lanes of a buffer load return 0 (and write 0 to
// LDS for the LDS-DMA form).  bytes is clipped to 32 bits; callers keep per-tile offsets far below 4 GiB by
// re-basing the descriptor at the tile origin.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t bytes) {
    uint32_t n = bytes <= 0 ? 0u : (bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}

// LDS-DMA issued through inline asm.  With the builtin, hipcc's wait-count pass treats every later LDS read that it
// cannot disambiguate (all ds_read_b64_tr_b16, and any read placed after an issue) as possibly aliasing the pending
// DMA and inserts s_waitcnt vmcnt(0) in front of it — which silently turns a counted-vmcnt ring into
// load-everything-then-compute.  Issued this way the compiler sees neither an LDS write nor a vmcnt event: ordering is
// entirely the kernel's explicit s_waitcnt vmcnt(N) + s_barrier, which is what the ring relies on anyway.
// rsrc: raw buffer descriptor words; lds_addr: wave-uniform LDS byte address of the 1-KiB destination (lane i lands at
// +16 i); voff: per-lane byte offset into the buffer (out-of-range lanes write zeros).
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_t make_rsrc_words(const void* base, int64_t bytes) {
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)bytes);
    const uint64_t b = (uint64_t)base;
    return i32x4_t{(int)(uint32_t)b, (int)(uint32_t)((b >> 32) & 0xFFFFu), (int)n, 0x00020000};
}
__device__ __forceinline__ void lds_dma16(i32x4_t rsrc, uint32_t lds_addr, int voff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");   // m0 is a reserved register: no kernel here mixes this with compiler-managed uses of it
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(p);
}

// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
// 4 x 16 block of 16-bit elements; lane i receives column i of the 4 rows (row q in element q).
__device__ __forceinline__ bf16x4 lds_read_tr16(const void* lds_addr) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_addr));
    return __builtin_bit_cast(bf16x4, t);
}

__device__ __forceinline__ bf16x8 join8(bf16x4 a, bf16x4 b) {
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// gelu_erf with the reference's constant 1.41421 (training/model.py:25) and its derivative, two elements at a time.
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, below fp32 resolution of the bf16-rounded results), branch-free: per
// element one v_rcp_f32, one v_exp_f32 and — on v_pk_{mul,fma}_f32, one issue slot per pair — five multiplies and seven FMAs
// (the ocml erff is branchy and several times longer).  The GELU epilogues are vector-ALU bound (both waves of a SIMD run this over
// 8192 elements each per 256 x 256 tile: ~9 us of a 36-us tile at K = 1024, round-5 stamps), so the sequence is kept minimal:
//   w = x sqrt(log2 e) / c          u = x / c is never formed: exp(-u^2) = exp2(-w^2), |u| = |w| / sqrt(log2 e)
//   g = exp2(-w w)                   (negation: a source modifier)
//   t = 1 / (1 + p' |w|)             (abs: a source modifier of the scalar FMA; p' = 0.3275911 / sqrt(log2 e))
//   q = a1 + t (a2 + t (a3 + t (a4 + t a5)))
//   erf|u| = 1 - q (t g),  Phi = 0.5 + 0.5 copysign(erf|u|, x),  gelu = x Phi,  gelu' = Phi + (x g) / (c sqrt(pi))
// Every GEMM structure calls this one function, so their GELU outputs agree bit for bit.
#define OBTE_GELU_C 1.41421f
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_ref_both2(f32x2_t x, f32x2_t& act, f32x2_t& der) {
    constexpr float K2 = 1.2011224087864498f / OBTE_GELU_C;        // sqrt(log2 e) / c
    constexpr float P2 = 0.3275911f / 1.2011224087864498f;
    const f32x2_t w = x * K2;
    const f32x2_t t2 = w * w;
    const f32x2_t gauss = {__builtin_amdgcn_exp2f(-t2[0]), __builtin_amdgcn_exp2f(-t2[1])};
    const f32x2_t t = {__builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(w[0]), P2, 1.0f)),
                       __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(w[1]), P2, 1.0f))};
    const f32x2_t q = 0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f)));
    const f32x2_t e_abs = 1.0f - q * (t * gauss);
    const f32x2_t e = {__builtin_copysignf(e_abs[0], x[0]), __builtin_copysignf(e_abs[1], x[1])};
    const f32x2_t half_cdf = e * 0.5f + 0.5f;
    act = x * half_cdf;
    der = half_cdf + (x * gauss) * (0.5641895835477563f / OBTE_GELU_C);
}

// ---- dropout: counter-based keep decision -------------------------------------------------------------------
// keep(row, col) is a pure function of (seed, site, row, col), so forward and backward regenerate the same mask with no
// mask tensor in memory, and a host-side restatement (oracle/omnibiote_ref.py dropout_keep) reproduces it bit for bit.
// Every dropout site is a matrix: row = token (embedding, projection outputs) or (batch, head, query) (attention
// probabilities), col = feature or key.  Two levels, so that the expensive part is paid once per ROW and once per PAIR
// of columns instead of once per element (the first form — two lowbias32 rounds and 64-bit index arithmetic per element —
// made the attention kernels 2.4x slower with dropout on):
//     rowkey = hash32(hash32(lo32(row) ^ s0) + hi32(row) * 0x9E3779B1 + s1)       once per row (per lane in attention)
//     bits   = hash32(rowkey ^ (col >> 1))                                        once per two columns
//     keep   = (col odd ? bits >> 16 : bits & 0xFFFF) >= thresh16                 thresh16 = round(p * 2^16)
// hash32 is the "lowbias32" integer finaliser.  p is resolved to 2^-16; the kept values are scaled by 1 / (1 - p).
struct DropCfg {
    uint32_t s0, s1, thresh16;   // thresh16 == 0  <=>  dropout off
    float scale;                 // 1 / (1 - p)
};
__host__ __device__ __forceinline__ uint32_t obte_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t drop_rowkey(uint64_t row, const DropCfg& c) {
    return obte_hash32(obte_hash32((uint32_t)row ^ c.s0) + (uint32_t)(row >> 32) * 0x9E3779B1U + c.s1);
}
// the 32 bits that decide columns 2g and 2g+1 of a row
__host__ __device__ __forceinline__ uint32_t drop_pair_bits(uint32_t rowkey, uint32_t g) { return obte_hash32(rowkey ^ g); }
__host__ __device__ __forceinline__ bool drop_keep_bits(uint32_t bits, uint32_t col, const DropCfg& c) {
    return ((col & 1u) ? (bits >> 16) : (bits & 0xFFFFu)) >= c.thresh16;
}
__host__ __device__ __forceinline__ bool drop_keep(uint32_t rowkey, uint32_t col, const DropCfg& c) {
    return drop_keep_bits(drop_pair_bits(rowkey, col >> 1), col, c);
}
static inline DropCfg make_drop(float p, uint64_t seed, uint32_t site) {
    DropCfg c;
    c.s0 = (uint32_t)seed ^ (site * 0x632BE5ABU);
    c.s1 = (uint32_t)(seed >> 32) + site * 0x9E3779B9U;
    c.thresh16 = p > 0.f ? (uint32_t)((double)p * 65536.0 + 0.5) : 0u;
    if (p > 0.f && c.thresh16 == 0u) c.thresh16 = 1u;
    c.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    return c;
}
enum { OBTE_SITE_EMBED = 0, OBTE_SITE_ATTN = 1, OBTE_SITE_RESID = 2, OBTE_SITE_MLP = 3, OBTE_SITE_USER = 7 };

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
