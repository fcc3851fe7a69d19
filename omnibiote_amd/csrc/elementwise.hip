// HBM-bound pieces of the path: RoPE on packed qkv, token-embedding gather / deterministic scatter-add,
// masked-LM cross entropy (forward + backward in one kernel), bf16 AdamW, sum of squares.
// All use 16-byte accesses per lane and grid sizes that fill 256 CUs.
#include "common.h"
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------------------------------------
// RoPE (training/model.py:39-50).  qkv [rows, 3C]; the q and k thirds are rotated in place.
// One thread = 8 consecutive elements = 4 (even, odd) pairs of one head.
// ---------------------------------------------------------------------------------------------------------
template <bool INVERSE>
__global__ __launch_bounds__(256) void rope_kernel(bf16* __restrict__ qkv, const float* __restrict__ cos_t,
                                                    const float* __restrict__ sin_t, int64_t rows, int64_t T, int C, int hs) {
    const int chunks_per_row = 2 * C / 8;  // q and k thirds
    const int64_t total = rows * chunks_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / chunks_per_row;
        const int col = (int)(i % chunks_per_row) * 8;  // 0 .. 2C-8: q then k, contiguous in the packed row
        const int d = col % hs;                         // C % hs == 0, so head boundaries align in both thirds
        const int64_t t = row % T;
        bf16* ptr = qkv + row * 3 * C + col;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(ptr);
        const f32x4 c = *reinterpret_cast<const f32x4*>(cos_t + t * (hs / 2) + d / 2);
        const f32x4 s = *reinterpret_cast<const f32x4*>(sin_t + t * (hs / 2) + d / 2);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xe = bf2f(v[2 * j]), xo = bf2f(v[2 * j + 1]);
            const float sj = INVERSE ? -s[j] : s[j];
            o[2 * j] = f2bf(xe * c[j] - xo * sj);
            o[2 * j + 1] = f2bf(xe * sj + xo * c[j]);
        }
        *reinterpret_cast<bf16x8*>(ptr) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Embedding gather (training/model.py:241).
// ---------------------------------------------------------------------------------------------------------
// Token ids outside [0, vocab) are clamped to the nearest valid row — in the forward gather AND in both backward
// kernels, so a bad id can never address memory outside wte / dwte (the reference raises a device-side assert instead;
// nothing on the host validates ids here, that would cost a sync).  The gradient of a clamped id lands on the row its
// forward read.  The stable argsort the backward receives is over the raw ids: every id < 0 sorts next to 0 and every
// id >= vocab next to vocab-1, so runs of equal clamped ids stay contiguous.
__device__ __forceinline__ int64_t clamp_tok(int64_t tok, int64_t vocab) { return tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok); }

__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ idx, const bf16* __restrict__ wte,
                                                         bf16* __restrict__ out, int64_t rows, int cols, int64_t vocab, DropCfg dc) {
    const int cpr = cols / 8;
    const int64_t total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cpr;
        const int c = (int)(i % cpr) * 8;
        const int64_t tok = clamp_tok(idx[r], vocab);
        bf16x8 v = *reinterpret_cast<const bf16x8*>(wte + tok * cols + c);
        if (dc.thresh16) {
            const uint32_t rk = drop_rowkey((uint64_t)r, dc);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const uint32_t bits = drop_pair_bits(rk, (uint32_t)(c >> 1) + jj);
#pragma unroll
                for (int e = 0; e < 2; ++e) v[2 * jj + e] = drop_keep_bits(bits, (uint32_t)e, dc) ? f2bf(bf2f(v[2 * jj + e]) * dc.scale) : f2bf(0.f);
            }
        }
        *reinterpret_cast<bf16x8*>(out + r * cols + c) = v;
    }
}

// element i*8 + j of the flat tensor is (row, column) = divmod(i*8 + j, cols); cols % 8 == 0, so a thread's 8 elements share a row
__global__ __launch_bounds__(256) void dropout_kernel(const bf16* in, bf16* out, int64_t n8, int64_t cols, DropCfg dc) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        bf16x8 v = reinterpret_cast<const bf16x8*>(in)[i];
        const int64_t row = (i * 8) / cols;
        const uint32_t c = (uint32_t)((i * 8) - row * cols);
        const uint32_t rk = drop_rowkey((uint64_t)row, dc);
        for (int jj = 0; jj < 4; ++jj) {
            const uint32_t bits = drop_pair_bits(rk, (c >> 1) + jj);
#pragma unroll
            for (int e = 0; e < 2; ++e) v[2 * jj + e] = drop_keep_bits(bits, (uint32_t)e, dc) ? f2bf(bf2f(v[2 * jj + e]) * dc.scale) : f2bf(0.f);
        }
        reinterpret_cast<bf16x8*>(out)[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Embedding backward: dwte[tok] = sum of dout rows with that token, fp32, fixed order (bitwise reproducible).
// The rows are visited in sorted-token order (order = stable argsort(idx)), cut into chunks of 32 positions.
// Pass A (one workgroup per chunk) sums each run of equal tokens inside its chunk; runs that touch a chunk
// edge shared with a neighbour holding the same token go to an fp32 slab (slot 0: the chunk's first run,
// slot 1: its last run), all others are written to dwte directly.  Pass B: the chunk where a spanning run
// starts adds the slabs of the following chunks in chunk order and writes the row.
// ---------------------------------------------------------------------------------------------------------
constexpr int EMB_CHUNK = 32;

// ACC: add to the existing row instead of overwriting it (each touched row is written by exactly one thread-set).
template <bool ACC>
__device__ __forceinline__ void embed_store_row(bf16* dst, const float (&acc)[8]) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(acc[j]);
    if (ACC) {
        const bf16x8 old = *reinterpret_cast<const bf16x8*>(dst);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(old[j]) + bf2f(o[j]));
    }
    *reinterpret_cast<bf16x8*>(dst) = o;
}

template <bool ACC>
__global__ __launch_bounds__(128) void embed_bwd_chunk_kernel(const int64_t* __restrict__ idx, const int32_t* __restrict__ order,
                                                               const bf16* __restrict__ dout, bf16* __restrict__ dwte,
                                                               float* __restrict__ slab, int64_t rows, int cols, int64_t vocab, DropCfg dc) {
    __shared__ int32_t s_row[EMB_CHUNK];
    __shared__ int64_t s_tok[EMB_CHUNK + 2];  // [0] = token before the chunk (or -1), [1..n] chunk, [n+1] = token after (or -1)
    const int64_t c = blockIdx.x;
    const int64_t p0 = c * EMB_CHUNK;
    const int n = (int)((rows - p0) < EMB_CHUNK ? (rows - p0) : EMB_CHUNK);
    if (threadIdx.x < n) {
        const int32_t r = order[p0 + threadIdx.x];
        s_row[threadIdx.x] = r;
        s_tok[threadIdx.x + 1] = clamp_tok(idx[r], vocab);
    }
    if (threadIdx.x == 64) s_tok[0] = p0 > 0 ? clamp_tok(idx[order[p0 - 1]], vocab) : -1;
    if (threadIdx.x == 65) s_tok[n + 1] = (p0 + n < rows) ? clamp_tok(idx[order[p0 + n]], vocab) : -1;
    __syncthreads();
    const bool left_open = s_tok[0] == s_tok[1];
    const bool right_open = s_tok[n + 1] == s_tok[n];
    for (int col = threadIdx.x * 8; col < cols; col += 128 * 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        int seg_start = 0;
        for (int i = 0; i < n; ++i) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(dout + (int64_t)s_row[i] * cols + col);
            if (dc.thresh16) {   // gradient of the dropped output: bf16(dout * scale) where kept, as autograd would form it
                const uint32_t rk = drop_rowkey((uint64_t)s_row[i], dc);
                for (int jj = 0; jj < 4; ++jj) {
                    const uint32_t bits = drop_pair_bits(rk, (uint32_t)(col >> 1) + jj);
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        if (drop_keep_bits(bits, (uint32_t)e, dc)) acc[2 * jj + e] += bf2f(f2bf(bf2f(v[2 * jj + e]) * dc.scale));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
            }
            const bool seg_end = (i == n - 1) || (s_tok[i + 2] != s_tok[i + 1]);
            if (seg_end) {
                const bool is_first = seg_start == 0, is_last = i == n - 1;
                if ((is_first && left_open) || (is_last && right_open)) {
                    float* dst = slab + ((c * 2 + (is_first ? 0 : 1)) * cols + col);
                    *reinterpret_cast<f32x4*>(dst) = f32x4{acc[0], acc[1], acc[2], acc[3]};
                    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
                } else {
                    embed_store_row<ACC>(dwte + s_tok[i + 1] * cols + col, acc);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = 0.f;
                seg_start = i + 1;
            }
        }
    }
}

template <bool ACC>
__global__ __launch_bounds__(128) void embed_bwd_span_kernel(const int64_t* __restrict__ idx, const int32_t* __restrict__ order,
                                                              bf16* __restrict__ dwte, const float* __restrict__ slab,
                                                              int64_t rows, int cols, int64_t nchunks, int64_t vocab) {
    const int64_t c = blockIdx.x;
    const int64_t p0 = c * EMB_CHUNK;
    const int64_t p1 = (p0 + EMB_CHUNK < rows) ? p0 + EMB_CHUNK : rows;
    const int64_t first = clamp_tok(idx[order[p0]], vocab), last = clamp_tok(idx[order[p1 - 1]], vocab);
    const bool single = first == last;
    const bool left_open = p0 > 0 && clamp_tok(idx[order[p0 - 1]], vocab) == first;
    const bool right_open = p1 < rows && clamp_tok(idx[order[p1]], vocab) == last;
    if (!right_open || (single && left_open)) return;  // no spanning run starts in this chunk
    const int slot0 = single ? 0 : 1;
    for (int col = threadIdx.x * 8; col < cols; col += 128 * 8) {
        float acc[8];
        const float* src = slab + ((c * 2 + slot0) * cols + col);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = src[j];
        for (int64_t cc = c + 1; cc < nchunks; ++cc) {
            const int64_t q0 = cc * EMB_CHUNK;
            const int64_t q1 = (q0 + EMB_CHUNK < rows) ? q0 + EMB_CHUNK : rows;
            if (clamp_tok(idx[order[q0]], vocab) != last) break;
            const float* s2 = slab + ((cc * 2) * cols + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += s2[j];
            if (clamp_tok(idx[order[q1 - 1]], vocab) != last) break;  // the run ended inside chunk cc
        }
        embed_store_row<ACC>(dwte + last * cols + col, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Masked-LM cross entropy, forward + backward (training/train_encoder.py:301-305).  One workgroup per row.
// Rows outside the MLM mask contribute exactly zero loss and zero gradient in the reference (loss *= mask),
// so their logits are not read; their dlogits row is written as zeros.
// ---------------------------------------------------------------------------------------------------------
// prev_mask (nullable): the MLM mask of the previous call on the SAME dlogits buffer.  Rows that were not masked then and
// are not masked now already hold zeros and are not touched at all, which removes ~85 % of the gradient write.
__global__ __launch_bounds__(256) void masked_ce_kernel(const bf16* __restrict__ logits, const int64_t* __restrict__ target,
                                                         const uint8_t* __restrict__ mlm_mask, const uint8_t* __restrict__ prev_mask,
                                                         const float* __restrict__ grad_scale, float row_scale,
                                                         float* __restrict__ row_loss, bf16* __restrict__ dlogits, int64_t vocab,
                                                         const int64_t* __restrict__ row_index, const float* __restrict__ row_scale_vec) {
    // row_index (nullable): compact form — workgroup r serves logits row row_index[r] (a masked position) and writes
    // gradient row r of a [n_masked, vocab] buffer; mlm_mask / prev_mask are not consulted
    __shared__ float red[8];
    const int64_t r = blockIdx.x;
    const int64_t src = row_index ? row_index[r] : r;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16* drow = dlogits + r * vocab;
    if (!row_index && mlm_mask && !mlm_mask[r]) {   // neither list nor mask: every row is served (obte_masked_ce_rows without a list)
        if (threadIdx.x == 0 && row_loss) row_loss[r] = 0.f;
        if (prev_mask && !prev_mask[r]) return;   // still zero from before
        const bf16x8 z = {};
        for (int64_t c = (int64_t)threadIdx.x * 8; c < vocab; c += 256 * 8) *reinterpret_cast<bf16x8*>(drow + c) = z;
        return;
    }
    const bf16* lrow = logits + src * vocab;
    // pass 1: online max / sum-exp (per thread), then combine
    float m = -INFINITY, l = 0.f;
    for (int64_t c = (int64_t)threadIdx.x * 8; c < vocab; c += 256 * 8) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(lrow + c);
        float mx = bf2f(v[0]);
#pragma unroll
        for (int j = 1; j < 8; ++j) mx = fmaxf(mx, bf2f(v[j]));
        const float mn = fmaxf(m, mx);
        float add = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) add += __expf(bf2f(v[j]) - mn);
        l = l * __expf(m - mn) + add;
        m = mn;
    }
    const float wm = wave_max(m);
    l = wave_sum(m == -INFINITY ? 0.f : l * __expf(m - wm));   // threads (or whole waves) without elements hold -inf
    if (lane == 0) { red[wave] = wm; red[4 + wave] = l; }
    __syncthreads();
    const float bm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float bl = red[4] * __expf(red[0] - bm) + red[5] * __expf(red[1] - bm) + red[6] * __expf(red[2] - bm) + red[7] * __expf(red[3] - bm);
    const float lse = bm + __logf(bl);
    int64_t tgt = target[src];
    tgt = tgt < 0 ? 0 : (tgt >= vocab ? vocab - 1 : tgt);
    const float rsc = row_scale_vec ? row_scale * row_scale_vec[r] : row_scale;   // per-row weight (compact form, several micro-batches in one call)
    if (threadIdx.x == 0 && row_loss) row_loss[r] = (lse - bf2f(lrow[tgt])) * rsc;
    // pass 2 (row is L2-resident): gradient
    const float gs = grad_scale ? rsc * grad_scale[0] : rsc;   // nullable device-side scale: NULL = 1
    for (int64_t c = (int64_t)threadIdx.x * 8; c < vocab; c += 256 * 8) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(lrow + c);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float p = __expf(bf2f(v[j]) - lse);
            if (c + j == tgt) p -= 1.0f;
            o[j] = f2bf(p * gs);
        }
        *reinterpret_cast<bf16x8*>(drow + c) = o;
    }
}

// The same row arithmetic with the row held in registers (vocab <= 256 threads x NCH x 8 elements: 65 536 at NCH = 32): every
// logit is loaded ONCE, all loads are in flight before the first reduction, and the gradient pass reads registers — the
// two-pass kernel above re-reads the row (from L2) chunk by chunk with each load queued behind the previous chunk's store
// (vmcnt retires in issue order).  Same operations in the same order per element; the per-thread online max / sum-exp is
// evaluated over the same chunks in the same order, so loss and gradients are bitwise those of the kernel above.
template <int NCH>
__global__ __launch_bounds__(256, 2) void masked_ce_regs_kernel(const bf16* __restrict__ logits, const int64_t* __restrict__ target,
                                                              const float* __restrict__ grad_scale, float row_scale,
                                                              float* __restrict__ row_loss, bf16* __restrict__ dlogits, int64_t vocab,
                                                              const int64_t* __restrict__ row_index, const float* __restrict__ row_scale_vec) {
    __shared__ float red[8];
    const int64_t r = blockIdx.x;
    const int64_t src = row_index ? row_index[r] : r;   // no list: the logits are already the listed rows ([n_rows, vocab])
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16* drow = dlogits + r * vocab;
    const bf16* lrow = logits + src * vocab;
    bf16x8 v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int64_t c = ((int64_t)i * 256 + threadIdx.x) * 8;
        v[i] = bf16x8{};
        if (c < vocab) v[i] = *reinterpret_cast<const bf16x8*>(lrow + c);
    }
    float m = -INFINITY, l = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int64_t c = ((int64_t)i * 256 + threadIdx.x) * 8;
        if (c < vocab) {
            float mx = bf2f(v[i][0]);
#pragma unroll
            for (int j = 1; j < 8; ++j) mx = fmaxf(mx, bf2f(v[i][j]));
            const float mn = fmaxf(m, mx);
            float add = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) add += __expf(bf2f(v[i][j]) - mn);
            l = l * __expf(m - mn) + add;
            m = mn;
        }
    }
    const float wm = wave_max(m);
    l = wave_sum(m == -INFINITY ? 0.f : l * __expf(m - wm));
    if (lane == 0) { red[wave] = wm; red[4 + wave] = l; }
    __syncthreads();
    const float bm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float bl = red[4] * __expf(red[0] - bm) + red[5] * __expf(red[1] - bm) + red[6] * __expf(red[2] - bm) + red[7] * __expf(red[3] - bm);
    const float lse = bm + __logf(bl);
    int64_t tgt = target[src];
    tgt = tgt < 0 ? 0 : (tgt >= vocab ? vocab - 1 : tgt);
    const float rsc = row_scale_vec ? row_scale * row_scale_vec[r] : row_scale;
    if (threadIdx.x == 0 && row_loss) row_loss[r] = (lse - bf2f(lrow[tgt])) * rsc;
    const float gs = grad_scale ? rsc * grad_scale[0] : rsc;   // nullable device-side scale: NULL = 1
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int64_t c = ((int64_t)i * 256 + threadIdx.x) * 8;
        if (c < vocab) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float p = __expf(bf2f(v[i][j]) - lse);
                if (c + j == tgt) p -= 1.0f;
                o[j] = f2bf(p * gs);
            }
            *reinterpret_cast<bf16x8*>(drow + c) = o;
        }
    }
}

// The same arithmetic again for vocab == 256 x NCH x 8 exactly (65 536 at NCH = 32), as a PERSISTENT kernel: one workgroup per CU,
// rows blockIdx.x, + gridDim.x, ...  The workgroup's NEXT row travels global -> LDS by LDS-DMA (no registers: 128 KiB of the CU's
// 160) while the current row is reduced, exponentiated and stored from registers, so its HBM reads run beside the current row's
// writes instead of after them, and the 1 229-row launch is 4.8 rows per workgroup (96 % balanced) instead of 2.4 rounds of 512
// resident workgroups (the third round 40 % full).  Each wave reads back exactly the bytes its own lanes requested, so the only
// synchronisation of the staging is the wave's own counted vmcnt; every scalar of the row (list entry, target, weights, the target's
// logit) comes through the scalar cache or LDS, so that NO compiler-issued vector load sits among the hand-counted ones.
template <int NCH, bool NT>
__global__ __launch_bounds__(256, 1) void masked_ce_pipe_kernel(const bf16* __restrict__ logits, const int64_t* __restrict__ target,
                                                                 const float* __restrict__ grad_scale, float row_scale,
                                                                 float* __restrict__ row_loss, bf16* __restrict__ dlogits, int64_t n_rows,
                                                                 const int64_t* __restrict__ row_index, const float* __restrict__ row_scale_vec) {
    constexpr int64_t vocab = (int64_t)256 * NCH * 8;
    extern __shared__ __attribute__((aligned(16))) char ce_smem[];   // [vocab] bf16 staging + the reduction scratch behind it
    float* red = reinterpret_cast<float*>(ce_smem + vocab * 2);     // [2][8]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t stride = gridDim.x;
    const uint32_t a_wave = lds_addr_of(ce_smem) + wave * 1024;     // chunk i of this wave: + 4096 i (the row's own byte order)
    const int voff = (int)threadIdx.x * 16;
    auto request = [&](int64_t r) {   // 32 LDS-DMA loads of 16 B per lane
        const int64_t src = row_index ? row_index[r] : r;
        const i32x4_t rs = make_rsrc_words(logits + src * vocab, vocab * 2);
#pragma unroll
        for (int i = 0; i < NCH; ++i)   // chunk i: + 4096 i bytes on both sides, through M0 and the scalar offset (no per-chunk VGPR)
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(a_wave + i * 4096), "v"(voff), "s"(rs), "s"(i * 4096) : "memory");
    };
    int64_t r = blockIdx.x;
    if (r >= n_rows) return;
    request(r);
    bool first = true;
    int par = 0;
    for (; r < n_rows; r += stride, par ^= 1) {
        const int64_t src = row_index ? row_index[r] : r;
        int64_t tgt = target[src];
        tgt = tgt < 0 ? 0 : (tgt >= vocab ? vocab - 1 : tgt);
        const float rsc = row_scale_vec ? row_scale * row_scale_vec[r] : row_scale;
        const float gs = grad_scale ? rsc * grad_scale[0] : rsc;
        // this row has landed (older than the 32 stores of the previous row, which may still be in flight)
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        first = false;
        bf16x8 v[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) v[i] = *reinterpret_cast<const bf16x8*>(ce_smem + wave * 1024 + lane * 16 + i * 4096);
        // (the target's logit is read by the thread whose own LDS-DMA fetched it: no barrier needed for the staging)
        const int tgt32 = (int)tgt;
        const bool owns_target = ((tgt32 >> 3) & 255) == (int)threadIdx.x;
        float xt = 0.f;
        if (owns_target) xt = bf2f(reinterpret_cast<const bf16*>(ce_smem)[tgt32]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the row is in registers: its staging may be overwritten
        if (r + stride < n_rows) request(r + stride);
        // ONE exponential per element: e = exp(x - running max of the thread at that chunk) is kept (fp32, 8 NCH registers: this
        // kernel has the whole register file of its SIMD) together with the chunk's maximum; the gradient pass multiplies it by
        // exp(chunk max - lse) * scale, one exponential per CHUNK.  (The one-row-per-workgroup kernels evaluate exp(x - lse) per
        // element a second time: 2 x 256 quarter-rate instructions per thread and row, which is what bounded them — 10 us of vector
        // ALU per row against 11 us of memory time at 6 TB/s.)  Equal to those kernels to fp32 rounding, not bitwise.
        float e[NCH][8], mc[NCH];
        float m = -INFINITY, l = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = bf2f(v[i][j]);
            float mx = x[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) mx = fmaxf(mx, x[j]);
            const float mn = fmaxf(m, mx);
            float add = 0.f;
            const float mnl = -mn * 1.4426950408889634f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { e[i][j] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[j], 1.4426950408889634f, mnl)); add += e[i][j]; }   // exp(x - mn): one FMA, one v_exp
            l = l * __expf(m - mn) + add;
            m = mn;
            mc[i] = mn;
            __builtin_amdgcn_sched_barrier(0);   // (chunk by chunk: left to itself hipcc converts the whole row first and spills; a spill reload is a vector load among the counted ones)
        }
        const float wm = wave_max(m);
        l = wave_sum(m == -INFINITY ? 0.f : l * __expf(m - wm));
        float* rd = red + par * 12;
        if (owns_target) rd[8] = xt;   // the target's logit, published with the wave partials (one barrier per row; scratch by row parity)
        if (lane == 0) { rd[wave] = wm; rd[4 + wave] = l; }
        __syncthreads();
        const float bm = fmaxf(fmaxf(rd[0], rd[1]), fmaxf(rd[2], rd[3]));
        const float bl = rd[4] * __expf(rd[0] - bm) + rd[5] * __expf(rd[1] - bm) + rd[6] * __expf(rd[2] - bm) + rd[7] * __expf(rd[3] - bm);
        const float lse = bm + __logf(bl);
        if (threadIdx.x == 0 && row_loss) row_loss[r] = (lse - rd[8]) * rsc;
        bf16* drow = dlogits + r * vocab + (int64_t)threadIdx.x * 8;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float f = __expf(mc[i] - lse) * gs;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(e[i][j] * f);
            // (asm: exactly one store instruction per chunk, whatever hipcc would have made of it — the wait above counts 32; the s_nop
            //  is the wait state an asm store of more than 8 bytes needs before its data registers are written again: hipcc pads only
            //  its own stores, and without it the next chunk's first conversion landed in this chunk's first dword)
            if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(drow + (int64_t)i * 256 * 8), "v"(o) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(drow + (int64_t)i * 256 * 8), "v"(o) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // (p - 1) * scale at the target: ONE element, rewritten by the thread whose chunk store just covered it (same wave, same
        // address: the two stores stay in order; for that wave the next row's wait counts one store too many, i.e. waits longer)
        if (owns_target) dlogits[r * vocab + tgt32] = f2bf((__expf(xt - lse) - 1.0f) * gs);
    }
}

// ---------------------------------------------------------------------------------------------------------
// AdamW in the reference's pure-bf16 regime (train_encoder.py:170,199): p, g, m, v all bf16; the update is
// evaluated in fp32 per element and each state is rounded once.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(bf16* __restrict__ p, const bf16* __restrict__ g, bf16* __restrict__ m,
                                                     bf16* __restrict__ v, int64_t n8, float lr, float b1, float b2, float eps,
                                                     float wd, float bc1, float bc2_sqrt, const float* __restrict__ clip) {
    const float cc = clip ? clip[0] : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        bf16x8 pp = reinterpret_cast<bf16x8*>(p)[i], mm = reinterpret_cast<bf16x8*>(m)[i], vv = reinterpret_cast<bf16x8*>(v)[i];
        const bf16x8 gg = reinterpret_cast<const bf16x8*>(g)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gj = bf2f(gg[j]) * cc;
            float pj = bf2f(pp[j]) * (1.0f - lr * wd);
            const float mj = bf2f(mm[j]) + (gj - bf2f(mm[j])) * (1.0f - b1);
            const float vj = bf2f(vv[j]) * b2 + gj * gj * (1.0f - b2);
            const float denom = sqrtf(vj) / bc2_sqrt + eps;
            pj -= (lr / bc1) * (mj / denom);
            pp[j] = f2bf(pj); mm[j] = f2bf(mj); vv[j] = f2bf(vj);
        }
        reinterpret_cast<bf16x8*>(p)[i] = pp; reinterpret_cast<bf16x8*>(m)[i] = mm; reinterpret_cast<bf16x8*>(v)[i] = vv;
    }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const bf16* __restrict__ g, int64_t n8, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const bf16x8 v = reinterpret_cast<const bf16x8*>(g)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = bf2f(v[j]); s += f * f; }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
// ---- multi-tensor forms: workgroup b serves chunk (b - first[t]) of tensor t; chunks of 16384 elements -----------
constexpr int MT_CHUNK = 16384;
struct MtTable {
    bf16* p[OBTE_MT_MAX]; const bf16* g[OBTE_MT_MAX]; bf16* m[OBTE_MT_MAX]; bf16* v[OBTE_MT_MAX];
    int64_t n[OBTE_MT_MAX]; float lr[OBTE_MT_MAX]; float wd[OBTE_MT_MAX]; float bc1[OBTE_MT_MAX]; float bc2s[OBTE_MT_MAX];
    float decay[OBTE_MT_MAX]; float step_size[OBTE_MT_MAX];   // 1 - lr*wd and lr / bias_correction1, formed in double on the host
    int first[OBTE_MT_MAX + 1];
    int count;
};

__device__ __forceinline__ int mt_find(const MtTable& t, int b) {
    int i = 0;
    while (i + 1 < t.count && b >= t.first[i + 1]) ++i;
    return i;
}

__global__ __launch_bounds__(256) void adamw_multi_kernel(MtTable t, float b1, float b2, float eps, const float* __restrict__ clip) {
    const int ti = mt_find(t, blockIdx.x);
    const int64_t base = (int64_t)(blockIdx.x - t.first[ti]) * MT_CHUNK;
    const int64_t end = min(base + (int64_t)MT_CHUNK, t.n[ti]);
    const float cc = clip ? clip[0] : 1.0f;
    const float lr = t.lr[ti], wd = t.wd[ti], bc1 = t.bc1[ti], bc2s = t.bc2s[ti];
    bf16* p = t.p[ti]; const bf16* g = t.g[ti]; bf16* m = t.m[ti]; bf16* v = t.v[ti];
    for (int64_t i = base + threadIdx.x * 8; i < end; i += 256 * 8) {
        bf16x8 pp = *reinterpret_cast<bf16x8*>(p + i), mm = *reinterpret_cast<bf16x8*>(m + i), vv = *reinterpret_cast<bf16x8*>(v + i);
        const bf16x8 gg = *reinterpret_cast<const bf16x8*>(g + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gj = bf2f(gg[j]) * cc;
            float pj = bf2f(pp[j]) * (1.0f - lr * wd);
            const float mj = bf2f(mm[j]) + (gj - bf2f(mm[j])) * (1.0f - b1);
            const float vj = bf2f(vv[j]) * b2 + gj * gj * (1.0f - b2);
            pj -= (lr / bc1) * (mj / (sqrtf(vj) / bc2s + eps));
            pp[j] = f2bf(pj); mm[j] = f2bf(mj); vv[j] = f2bf(vj);
        }
        *reinterpret_cast<bf16x8*>(p + i) = pp; *reinterpret_cast<bf16x8*>(m + i) = mm; *reinterpret_cast<bf16x8*>(v + i) = vv;
    }
}

// The reference's arithmetic, rounding for rounding: torch.optim.AdamW on bf16 parameters with bf16 moments
// (train_encoder.py:170,195-199) evaluates each tensor op in fp32 and rounds its result to bf16 before the next op
// (torch/optim/adamw.py, both the for-loop and the foreach form):
//     g   = bf16(g * clip)                        clip_grad_norm_'s in-place scaling (train_encoder.py:316)
//     p   = bf16(p * (1 - lr*wd))                 param.mul_
//     m   = bf16(m + (g - m) * (1 - beta1))       exp_avg.lerp_
//     v   = bf16(v * beta2);  v = bf16(v + (1 - beta2) * g * g)          exp_avg_sq.mul_().addcmul_()
//     d   = bf16(sqrt(v));  d = bf16(d / sqrt(bias_correction2));  d = bf16(d + eps)
//     p   = bf16(p + (-lr / bias_correction1) * (m / d))                  param.addcdiv_
// adamw_multi_kernel above rounds each state once per step instead (more accurate, not what the reference computes).
// Ties are COMMON here (m, g sit on coarse bf16 grids, so 0.9 m + 0.1 g lands exactly between two bf16 values in
// about 1.6 % of the elements), which makes the last fp32 bit — fused or unfused multiply-add, the association of
// value * m / d, 0.1f vs 1.0f - 0.9f — decide a whole bf16 ulp.  The sequence below is torch's CPU kernels', checked
// element for element against torch.optim.AdamW (tests/test_hip_ops.py): lerp = fma(w, g - m, m) (LerpKernel's
// vec::fmadd), addcmul = fma(w * g, g, v), addcdiv = p + ((value * m) / d); contraction is switched off so that hipcc
// fuses nothing else.  Measured: parameters and both moments bit-identical to torch's after every step.  w1 = float(1 - beta1), w2 = float(1 - beta2) are formed in double by the host.
__global__ __launch_bounds__(256) void adamw_multi_ref_kernel(MtTable t, float b2, float w1, float w2, float eps, const float* __restrict__ clip) {
#pragma clang fp contract(off)
    const int ti = mt_find(t, blockIdx.x);
    const int64_t base = (int64_t)(blockIdx.x - t.first[ti]) * MT_CHUNK;
    const int64_t end = min(base + (int64_t)MT_CHUNK, t.n[ti]);
    const float decay = t.decay[ti], neg_step = -t.step_size[ti], bc2s = t.bc2s[ti];
    const bool clipped = clip != nullptr;
    const float cc = clipped ? clip[0] : 1.0f;
    bf16* p = t.p[ti]; const bf16* g = t.g[ti]; bf16* m = t.m[ti]; bf16* v = t.v[ti];
    for (int64_t i = base + threadIdx.x * 8; i < end; i += 256 * 8) {
        bf16x8 pp = *reinterpret_cast<bf16x8*>(p + i), mm = *reinterpret_cast<bf16x8*>(m + i), vv = *reinterpret_cast<bf16x8*>(v + i);
        const bf16x8 gg = *reinterpret_cast<const bf16x8*>(g + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gj = clipped ? bf2f(f2bf(bf2f(gg[j]) * cc)) : bf2f(gg[j]);
            float pj = bf2f(f2bf(bf2f(pp[j]) * decay));
            const float m0 = bf2f(mm[j]);
            const float mj = bf2f(f2bf(__builtin_fmaf(w1, gj - m0, m0)));
            float vj = bf2f(f2bf(bf2f(vv[j]) * b2));
            const float wg = w2 * gj;
            vj = bf2f(f2bf(__builtin_fmaf(wg, gj, vj)));
            float d = bf2f(f2bf(sqrtf(vj)));
            d = bf2f(f2bf(d / bc2s));
            d = bf2f(f2bf(d + eps));
            const float num = neg_step * mj;
            const float upd = num / d;
            pj = pj + upd;
            pp[j] = f2bf(pj); mm[j] = f2bf(mj); vv[j] = f2bf(vj);
        }
        *reinterpret_cast<bf16x8*>(p + i) = pp; *reinterpret_cast<bf16x8*>(m + i) = mm; *reinterpret_cast<bf16x8*>(v + i) = vv;
    }
}

// each != 0: one sum per tensor (out[tensor index]) instead of one for all of them
__global__ __launch_bounds__(256) void sumsq_multi_kernel(MtTable t, float* __restrict__ out, int each) {
    __shared__ float red[4];
    const int ti = mt_find(t, blockIdx.x);
    const int64_t base = (int64_t)(blockIdx.x - t.first[ti]) * MT_CHUNK;
    const int64_t end = min(base + (int64_t)MT_CHUNK, t.n[ti]);
    const bf16* g = t.g[ti];
    float s = 0.f;
    for (int64_t i = base + threadIdx.x * 8; i < end; i += 256 * 8) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(g + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = bf2f(v[j]); s += f * f; }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + (each ? ti : 0), red[0] + red[1] + red[2] + red[3]);
}

int mt_build(const obte_mt_args* a, float beta1, float beta2, MtTable* t, const char* who, bool need_state) {
    OBTE_REQUIRE(a && a->count >= 1 && a->count <= OBTE_MT_MAX, "%s: count must be 1..%d", who, OBTE_MT_MAX);
    t->count = a->count;
    int blocks = 0;
    for (int i = 0; i < a->count; ++i) {
        OBTE_REQUIRE(a->g[i] && a->n[i] > 0 && a->n[i] % 8 == 0, "%s: tensor %d: null gradient or n not a multiple of 8", who, i);
        if (need_state) OBTE_REQUIRE(a->p[i] && a->m[i] && a->v[i] && a->step[i] >= 1, "%s: tensor %d: null state or step < 1", who, i);
        t->p[i] = (bf16*)a->p[i]; t->g[i] = (const bf16*)a->g[i]; t->m[i] = (bf16*)a->m[i]; t->v[i] = (bf16*)a->v[i];
        t->n[i] = a->n[i]; t->lr[i] = a->lr[i]; t->wd[i] = a->weight_decay[i];
        const int st = a->step[i] < 1 ? 1 : a->step[i];
        t->bc1[i] = 1.0f - powf(beta1, (float)st);
        t->bc2s[i] = sqrtf(1.0f - powf(beta2, (float)st));
        t->decay[i] = (float)(1.0 - (double)a->lr[i] * (double)a->weight_decay[i]);
        t->step_size[i] = (float)((double)a->lr[i] / (1.0 - pow((double)beta1, (double)st)));
        t->first[i] = blocks;
        blocks += (int)cdiv64(a->n[i], MT_CHUNK);
    }
    t->first[a->count] = blocks;
    return blocks;
}

inline unsigned stream_grid(int64_t work_items, int per_block) {
    int64_t b = cdiv64(work_items, per_block);
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int obte_rope_qk_inplace(obte_bf16* qkv, const float* cos_t, const float* sin_t, int64_t B, int64_t T,
                                    int n_head, int head_dim, int inverse, obte_stream s) {
    OBTE_REQUIRE(qkv && cos_t && sin_t, "obte_rope_qk_inplace: null pointer");
    OBTE_REQUIRE(B > 0 && T > 0 && n_head > 0 && head_dim % 8 == 0, "obte_rope_qk_inplace: head_dim must be a multiple of 8");
    const int C = n_head * head_dim;
    const int64_t rows = B * T, total = rows * (2 * C / 8);
    hipStream_t st = (hipStream_t)s;
    if (inverse)
        hipLaunchKernelGGL((rope_kernel<true>), dim3(stream_grid(total, 256)), dim3(256), 0, st, (bf16*)qkv, cos_t, sin_t, rows, T, C, head_dim);
    else
        hipLaunchKernelGGL((rope_kernel<false>), dim3(stream_grid(total, 256)), dim3(256), 0, st, (bf16*)qkv, cos_t, sin_t, rows, T, C, head_dim);
    OBTE_CHECK_LAUNCH("obte_rope_qk_inplace");
    return OBTE_OK;
}

static int check_p(const char* who, float p) {
    OBTE_REQUIRE(p >= 0.f && p < 1.f, "%s: dropout p must be in [0, 1) (got %g)", who, (double)p);
    return OBTE_OK;
}

extern "C" int obte_embedding_fwd_dropout(const int64_t* idx, const obte_bf16* wte, obte_bf16* out, int64_t rows, int cols,
                                          int64_t vocab, float p, uint64_t seed, obte_stream s) {
    OBTE_REQUIRE(idx && wte && out, "obte_embedding_fwd: null pointer");
    OBTE_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && vocab > 0, "obte_embedding_fwd: cols must be a multiple of 8");
    if (check_p("obte_embedding_fwd", p)) return OBTE_EINVAL;
    if (rows == 0) return OBTE_OK;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(stream_grid(rows * (cols / 8), 256)), dim3(256), 0, (hipStream_t)s, idx,
                       (const bf16*)wte, (bf16*)out, rows, cols, vocab, make_drop(p, seed, OBTE_SITE_EMBED));
    OBTE_CHECK_LAUNCH("obte_embedding_fwd");
    return OBTE_OK;
}

extern "C" int obte_embedding_fwd(const int64_t* idx, const obte_bf16* wte, obte_bf16* out, int64_t rows, int cols,
                                  int64_t vocab, obte_stream s) {
    return obte_embedding_fwd_dropout(idx, wte, out, rows, cols, vocab, 0.f, 0, s);
}

// the same on GATHERED rows: row i of in / out / aux is row rows[i] of the whole activation the mask is defined on (the rows form of
// the block: csrc/block.cpp), out = (aux ? aux : 0) + dropout(in)
__global__ __launch_bounds__(256) void dropout_rows_kernel(const bf16* __restrict__ in, const bf16* __restrict__ aux, bf16* __restrict__ out,
                                                           const int64_t* __restrict__ rows, int64_t n_rows, int cols, DropCfg cfg) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const uint32_t rk = drop_rowkey((uint64_t)rows[i], cfg);
    for (int c = lane * 8; c < cols; c += 512) {
        bf16x8 v = *reinterpret_cast<const bf16x8*>(in + i * cols + c);
        bf16x8 a = {};
        if (aux) a = *reinterpret_cast<const bf16x8*>(aux + i * cols + c);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const uint32_t bits = drop_pair_bits(rk, ((uint32_t)c >> 1) + jj);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * jj + e;
                const float t = drop_keep_bits(bits, (uint32_t)e, cfg) ? bf2f(f2bf(bf2f(v[j]) * cfg.scale)) : 0.f;
                v[j] = aux ? f2bf(bf2f(a[j]) + t) : f2bf(t);
            }
        }
        *reinterpret_cast<bf16x8*>(out + i * cols + c) = v;
    }
}

extern "C" int obte_dropout_bf16(const obte_bf16* in, obte_bf16* out, int64_t n, int64_t cols, float p, uint64_t seed, int32_t site,
                                 obte_stream s) {
    OBTE_REQUIRE(in && out && n > 0 && n % 8 == 0, "obte_dropout_bf16: null pointer or n not a positive multiple of 8");
    OBTE_REQUIRE(cols > 0 && cols % 8 == 0 && n % cols == 0 && cols < (1ll << 32), "obte_dropout_bf16: cols must be a multiple of 8 that divides n");
    if (check_p("obte_dropout_bf16", p)) return OBTE_EINVAL;
    hipLaunchKernelGGL(dropout_kernel, dim3(stream_grid(n / 8, 256)), dim3(256), 0, (hipStream_t)s, (const bf16*)in, (bf16*)out, n / 8,
                       cols, make_drop(p, seed, (uint32_t)site));
    OBTE_CHECK_LAUNCH("obte_dropout_bf16");
    return OBTE_OK;
}

extern "C" int64_t obte_embedding_bwd_ws_bytes(int64_t rows, int cols) {
    return cdiv64(rows, EMB_CHUNK) * 2 * (int64_t)cols * (int64_t)sizeof(float);
}

extern "C" int obte_embedding_bwd_dropout(const int64_t* idx, const int32_t* order, const obte_bf16* dout, obte_bf16* dwte,
                                          void* ws, int64_t rows, int cols, int64_t vocab, int accumulate, float p, uint64_t seed,
                                          obte_stream s) {
    OBTE_REQUIRE(idx && order && dout && dwte && ws, "obte_embedding_bwd: null pointer");
    if (check_p("obte_embedding_bwd", p)) return OBTE_EINVAL;
    const DropCfg dc = make_drop(p, seed, OBTE_SITE_EMBED);
    OBTE_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && vocab > 0, "obte_embedding_bwd: bad shape");
    OBTE_REQUIRE(rows < (1ll << 31), "obte_embedding_bwd: too many rows");
    hipStream_t st = (hipStream_t)s;
    if (!accumulate && hipMemsetAsync(dwte, 0, (size_t)vocab * cols * sizeof(obte_bf16), st) != hipSuccess) {
        obte_set_error("obte_embedding_bwd: memset failed");
        return OBTE_ELAUNCH;
    }
    const int64_t nchunks = cdiv64(rows, EMB_CHUNK);
    if (accumulate)
        hipLaunchKernelGGL(embed_bwd_chunk_kernel<true>, dim3((unsigned)nchunks), dim3(128), 0, st, idx, order, (const bf16*)dout,
                           (bf16*)dwte, (float*)ws, rows, cols, vocab, dc);
    else
        hipLaunchKernelGGL(embed_bwd_chunk_kernel<false>, dim3((unsigned)nchunks), dim3(128), 0, st, idx, order, (const bf16*)dout,
                           (bf16*)dwte, (float*)ws, rows, cols, vocab, dc);
    OBTE_CHECK_LAUNCH("obte_embedding_bwd(chunk)");
    if (accumulate)
        hipLaunchKernelGGL(embed_bwd_span_kernel<true>, dim3((unsigned)nchunks), dim3(128), 0, st, idx, order, (bf16*)dwte,
                           (const float*)ws, rows, cols, nchunks, vocab);
    else
        hipLaunchKernelGGL(embed_bwd_span_kernel<false>, dim3((unsigned)nchunks), dim3(128), 0, st, idx, order, (bf16*)dwte,
                           (const float*)ws, rows, cols, nchunks, vocab);
    OBTE_CHECK_LAUNCH("obte_embedding_bwd(span)");
    return OBTE_OK;
}

extern "C" int obte_embedding_bwd_acc(const int64_t* idx, const int32_t* order, const obte_bf16* dout, obte_bf16* dwte,
                                      void* ws, int64_t rows, int cols, int64_t vocab, int accumulate, obte_stream s) {
    return obte_embedding_bwd_dropout(idx, order, dout, dwte, ws, rows, cols, vocab, accumulate, 0.f, 0, s);
}

extern "C" int obte_embedding_bwd(const int64_t* idx, const int32_t* order, const obte_bf16* dout, obte_bf16* dwte,
                                  void* ws, int64_t rows, int cols, int64_t vocab, obte_stream s) {
    return obte_embedding_bwd_dropout(idx, order, dout, dwte, ws, rows, cols, vocab, 0, 0.f, 0, s);
}

extern "C" int obte_masked_ce_fwd_bwd_reuse(const obte_bf16* logits, const int64_t* target, const uint8_t* mlm_mask,
                                            const uint8_t* prev_mask, const float* grad_scale, float row_scale, float* row_loss,
                                            obte_bf16* dlogits, int64_t rows, int64_t vocab, obte_stream s) {
    OBTE_REQUIRE(logits && target && mlm_mask && grad_scale && dlogits, "obte_masked_ce_fwd_bwd: null pointer");
    OBTE_REQUIRE(rows > 0 && rows < (1ll << 31) && vocab > 0 && vocab % 8 == 0, "obte_masked_ce_fwd_bwd: vocab must be a multiple of 8");
    hipLaunchKernelGGL(masked_ce_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)s, (const bf16*)logits, target, mlm_mask,
                       prev_mask, grad_scale, row_scale, row_loss, (bf16*)dlogits, vocab, (const int64_t*)nullptr, (const float*)nullptr);
    OBTE_CHECK_LAUNCH("obte_masked_ce_fwd_bwd");
    return OBTE_OK;
}

// ---- rows by index (the rows form of the block: csrc/block.cpp) ---------------------------------------------------------
// one wave per row, 16 B per lane and step; gather: dst[i] = src[rows[i]]; scatter: dst[rows[i]] = src[i]
namespace {
__global__ __launch_bounds__(256) void rows_copy_kernel(const bf16* __restrict__ src, const int64_t* __restrict__ rows, bf16* __restrict__ dst,
                                                         int64_t n_rows, int64_t total_rows, int cols, int scatter) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= n_rows) return;
    int64_t r = rows[i];
    r = r < 0 ? 0 : (r >= total_rows ? total_rows - 1 : r);
    const bf16* s = src + (scatter ? i : r) * cols;
    bf16* d = dst + (scatter ? r : i) * cols;
    for (int c = lane * 8; c < cols; c += 64 * 8) *reinterpret_cast<bf16x8*>(d + c) = *reinterpret_cast<const bf16x8*>(s + c);
}
}  // namespace
extern "C" int obte_rows_gather_bf16(const obte_bf16* src, const int64_t* rows, obte_bf16* dst, int64_t n_rows, int64_t total_rows, int32_t cols, obte_stream s) {
    OBTE_REQUIRE(src && rows && dst, "obte_rows_gather_bf16: null pointer");
    OBTE_REQUIRE(n_rows > 0 && total_rows > 0 && cols > 0 && cols % 8 == 0, "obte_rows_gather_bf16: need n_rows, total_rows > 0 and cols %% 8 == 0");
    hipLaunchKernelGGL(rows_copy_kernel, dim3((unsigned)cdiv64(n_rows, 4)), dim3(256), 0, (hipStream_t)s, (const bf16*)src, rows, (bf16*)dst, n_rows, total_rows, cols, 0);
    OBTE_CHECK_LAUNCH("obte_rows_gather_bf16");
    return OBTE_OK;
}
extern "C" int obte_rows_scatter_bf16(const obte_bf16* src, const int64_t* rows, obte_bf16* dst, int64_t n_rows, int64_t total_rows, int32_t cols, obte_stream s) {
    OBTE_REQUIRE(src && rows && dst, "obte_rows_scatter_bf16: null pointer");
    OBTE_REQUIRE(n_rows > 0 && n_rows <= total_rows && cols > 0 && cols % 8 == 0, "obte_rows_scatter_bf16: need 0 < n_rows <= total_rows and cols %% 8 == 0");
    if (hipMemsetAsync(dst, 0, (size_t)total_rows * cols * 2, (hipStream_t)s) != hipSuccess) { obte_set_error("obte_rows_scatter_bf16: memset failed"); return OBTE_ELAUNCH; }
    hipLaunchKernelGGL(rows_copy_kernel, dim3((unsigned)cdiv64(n_rows, 4)), dim3(256), 0, (hipStream_t)s, (const bf16*)src, rows, (bf16*)dst, n_rows, total_rows, cols, 1);
    OBTE_CHECK_LAUNCH("obte_rows_scatter_bf16");
    return OBTE_OK;
}

extern "C" int obte_masked_ce_rows(const obte_bf16* logits, const int64_t* target, const int64_t* row_index, const float* grad_scale,
                                   float row_scale, const float* row_scale_vec, float* row_loss, obte_bf16* dlogits_rows, int64_t n_rows,
                                   int64_t total_rows, int64_t vocab, obte_stream s) {
    OBTE_REQUIRE(logits && target && dlogits_rows && row_loss, "obte_masked_ce_rows: null pointer");
    OBTE_REQUIRE(n_rows > 0 && n_rows <= total_rows && total_rows < (1ll << 31) && vocab > 0 && vocab % 8 == 0,
                 "obte_masked_ce_rows: need 0 < n_rows <= total_rows and vocab %% 8 == 0");
    OBTE_REQUIRE(row_index || n_rows == total_rows, "obte_masked_ce_rows: without a row list, logits and target hold exactly the n_rows listed rows");
    OBTE_REQUIRE((const void*)logits != (const void*)dlogits_rows, "obte_masked_ce_rows: dlogits_rows must not alias logits");
    const int prof = obte_prof_begin((hipStream_t)s, 112, n_rows, vocab, 1);   // algorithmic bytes = 4 * n_rows * vocab (read + write)
    // A/B switches, read once: ONE thread-safe static holds them all (this entry point is called from autograd's worker threads; with
    // several lazily set ints a second thread could see one of them set and another still at its placeholder — a grid of 0).
    // OBTE_CE_REGS=0: the two-pass kernel; OBTE_CE_PIPE=0: one workgroup per row (both bitwise the same results)
    struct CeCfg { int regs_on, pipe_on, nt_on, pipe_grid; };
    static const CeCfg cfg = [] {
        CeCfg c;
        const char* e = getenv("OBTE_CE_REGS"); c.regs_on = (e && e[0] == '0') ? 0 : 1;
        e = getenv("OBTE_CE_PIPE"); c.pipe_on = (e && e[0] == '0') ? 0 : 1;
        e = getenv("OBTE_CE_NT"); c.nt_on = (e && e[0] == '1') ? 1 : 0;
        e = getenv("OBTE_CE_GRID"); c.pipe_grid = e ? atoi(e) : 256;
        if (c.pipe_grid < 1 || c.pipe_grid > 4096) c.pipe_grid = 256;
        return c;
    }();
    const int regs_on = cfg.regs_on, pipe_on = cfg.pipe_on, nt_on = cfg.nt_on, pipe_grid = cfg.pipe_grid;
    if (regs_on && pipe_on && vocab == 256 * 32 * 8 && n_rows > pipe_grid) {
        const dim3 grid((unsigned)pipe_grid);
        const int smem = (int)(vocab * 2 + 128);
        if (nt_on) {
            (void)hipFuncSetAttribute((const void*)masked_ce_pipe_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            hipLaunchKernelGGL((masked_ce_pipe_kernel<32, true>), grid, dim3(256), smem, (hipStream_t)s, (const bf16*)logits, target, grad_scale, row_scale,
                               row_loss, (bf16*)dlogits_rows, n_rows, row_index, row_scale_vec);
        } else {
            (void)hipFuncSetAttribute((const void*)masked_ce_pipe_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            hipLaunchKernelGGL((masked_ce_pipe_kernel<32, false>), grid, dim3(256), smem, (hipStream_t)s, (const bf16*)logits, target, grad_scale, row_scale,
                               row_loss, (bf16*)dlogits_rows, n_rows, row_index, row_scale_vec);
        }
    } else if (regs_on && vocab <= 256 * 8 * 8)
        hipLaunchKernelGGL((masked_ce_regs_kernel<8>), dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)s, (const bf16*)logits, target,
                           grad_scale, row_scale, row_loss, (bf16*)dlogits_rows, vocab, row_index, row_scale_vec);
    else if (regs_on && vocab <= 256 * 32 * 8)
        hipLaunchKernelGGL((masked_ce_regs_kernel<32>), dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)s, (const bf16*)logits, target,
                           grad_scale, row_scale, row_loss, (bf16*)dlogits_rows, vocab, row_index, row_scale_vec);
    else
        hipLaunchKernelGGL(masked_ce_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)s, (const bf16*)logits, target,
                           (const uint8_t*)nullptr, (const uint8_t*)nullptr, grad_scale, row_scale, row_loss, (bf16*)dlogits_rows, vocab, row_index,
                           row_scale_vec);
    obte_prof_end(prof, (hipStream_t)s);
    OBTE_CHECK_LAUNCH("obte_masked_ce_rows");
    return OBTE_OK;
}

extern "C" int obte_masked_ce_fwd_bwd(const obte_bf16* logits, const int64_t* target, const uint8_t* mlm_mask,
                                      const float* grad_scale, float row_scale, float* loss_sum, float* row_loss,
                                      obte_bf16* dlogits, int64_t rows, int64_t vocab, obte_stream s) {
    (void)loss_sum;
    return obte_masked_ce_fwd_bwd_reuse(logits, target, mlm_mask, nullptr, grad_scale, row_scale, row_loss, dlogits, rows, vocab, s);
}

extern "C" int obte_adamw_bf16(obte_bf16* p, const obte_bf16* g, obte_bf16* m, obte_bf16* v, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int32_t step,
                               const float* clip_coef, obte_stream s) {
    OBTE_REQUIRE(p && g && m && v, "obte_adamw_bf16: null pointer");
    OBTE_REQUIRE(n > 0 && n % 8 == 0 && step >= 1, "obte_adamw_bf16: n must be a positive multiple of 8, step >= 1");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(stream_grid(n / 8, 256)), dim3(256), 0, (hipStream_t)s, (bf16*)p, (const bf16*)g,
                       (bf16*)m, (bf16*)v, n / 8, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, clip_coef);
    OBTE_CHECK_LAUNCH("obte_adamw_bf16");
    return OBTE_OK;
}

extern "C" int obte_sumsq_bf16(const obte_bf16* g, int64_t n, float* out, obte_stream s) {
    OBTE_REQUIRE(g && out, "obte_sumsq_bf16: null pointer");
    OBTE_REQUIRE(n > 0 && n % 8 == 0, "obte_sumsq_bf16: n must be a positive multiple of 8");
    hipLaunchKernelGGL(sumsq_kernel, dim3(stream_grid(n / 8, 256)), dim3(256), 0, (hipStream_t)s, (const bf16*)g, n / 8, out);
    OBTE_CHECK_LAUNCH("obte_sumsq_bf16");
    return OBTE_OK;
}

extern "C" int obte_adamw_multi_bf16(const obte_mt_args* a, float beta1, float beta2, float eps, const float* clip_coef, obte_stream s) {
    MtTable t;
    const int blocks = mt_build(a, beta1, beta2, &t, "obte_adamw_multi_bf16", true);
    if (blocks < 0) return blocks;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, t, beta1, beta2, eps, clip_coef);
    OBTE_CHECK_LAUNCH("obte_adamw_multi_bf16");
    return OBTE_OK;
}

extern "C" int obte_adamw_multi_bf16_ref(const obte_mt_args* a, double beta1, double beta2, double eps, const float* clip_coef, obte_stream s) {
    MtTable t;
    const int blocks = mt_build(a, (float)beta1, (float)beta2, &t, "obte_adamw_multi_bf16_ref", true);
    if (blocks < 0) return blocks;
    for (int i = 0; i < a->count; ++i) {   // Python forms these in double before they reach a kernel as fp32 scalars
        const double st = (double)(a->step[i] < 1 ? 1 : a->step[i]);
        const double lr = a->lr64[i] != 0.0 ? a->lr64[i] : (double)a->lr[i];
        const double wd = a->weight_decay64[i] != 0.0 ? a->weight_decay64[i] : (double)a->weight_decay[i];
        t.bc2s[i] = (float)sqrt(1.0 - pow(beta2, st));
        t.step_size[i] = (float)(lr / (1.0 - pow(beta1, st)));
        t.decay[i] = (float)(1.0 - lr * wd);
    }
    const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2);
    int64_t n_all = 0;
    for (int i = 0; i < a->count; ++i) n_all += a->n[i];
    const int prof = obte_prof_begin((hipStream_t)s, 113, n_all, 1, 1);        // algorithmic bytes = 14 * elements (p, g, m, v read; p, m, v written)
    hipLaunchKernelGGL(adamw_multi_ref_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, t, (float)beta2, w1, w2, (float)eps, clip_coef);
    obte_prof_end(prof, (hipStream_t)s);
    OBTE_CHECK_LAUNCH("obte_adamw_multi_bf16_ref");
    return OBTE_OK;
}

extern "C" int obte_sumsq_multi_bf16(const obte_mt_args* a, float* out, obte_stream s) {
    OBTE_REQUIRE(out, "obte_sumsq_multi_bf16: null output");
    MtTable t;
    const int blocks = mt_build(a, 0.9f, 0.999f, &t, "obte_sumsq_multi_bf16", false);
    if (blocks < 0) return blocks;
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, t, out, 0);
    OBTE_CHECK_LAUNCH("obte_sumsq_multi_bf16");
    return OBTE_OK;
}

extern "C" int obte_sumsq_multi_bf16_each(const obte_mt_args* a, float* out, obte_stream s) {
    OBTE_REQUIRE(out, "obte_sumsq_multi_bf16_each: null output");
    MtTable t;
    const int blocks = mt_build(a, 0.9f, 0.999f, &t, "obte_sumsq_multi_bf16_each", false);
    if (blocks < 0) return blocks;
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, t, out, 1);
    OBTE_CHECK_LAUNCH("obte_sumsq_multi_bf16_each");
    return OBTE_OK;
}

int obte_dropout_rows_bf16(const obte_bf16* in, const obte_bf16* aux, obte_bf16* out, const int64_t* rows, int64_t n_rows, int32_t cols, float p,
                           uint64_t seed, int32_t site, obte_stream s) {
    OBTE_REQUIRE(in && out && rows && n_rows > 0 && cols > 0 && cols % 8 == 0, "obte_dropout_rows_bf16: bad arguments");
    OBTE_REQUIRE(p > 0.f && p < 1.f, "obte_dropout_rows_bf16: p must be in (0,1)");
    hipLaunchKernelGGL(dropout_rows_kernel, dim3((unsigned)cdiv64(n_rows, 4)), dim3(256), 0, (hipStream_t)s, (const bf16*)in, (const bf16*)aux, (bf16*)out,
                       rows, n_rows, (int)cols, make_drop(p, seed, site));
    OBTE_CHECK_LAUNCH("obte_dropout_rows_bf16");
    return OBTE_OK;
}

// ---- the rows form of the block's c_attn (csrc/block.cpp): RoPE on ONE column block of a row-strided activation, the row's position taken
// from a list (gathered rows) or as row % T; and dst[rows[i]] += src[i] ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_cols_kernel(bf16* __restrict__ x, int64_t ld, int ncols, const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                        int64_t rows, int64_t T, const int32_t* __restrict__ pos, int hs) {
    const int cpr = ncols / 8;
    const int64_t total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int col = (int)(i % cpr) * 8;
        const int d = col % hs;
        const int64_t t = pos ? (int64_t)pos[row] : row % T;
        bf16* ptr = x + row * ld + col;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(ptr);
        const f32x4 c = *reinterpret_cast<const f32x4*>(cos_t + t * (hs / 2) + d / 2);
        const f32x4 sn = *reinterpret_cast<const f32x4*>(sin_t + t * (hs / 2) + d / 2);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // (the arithmetic of the GEMM epilogue OBTE_EPI_ROPE_QK and of rope_kernel)
            const float xe = bf2f(v[2 * j]), xo = bf2f(v[2 * j + 1]);
            o[2 * j] = f2bf(xe * c[j] - xo * sn[j]);
            o[2 * j + 1] = f2bf(xe * sn[j] + xo * c[j]);
        }
        *reinterpret_cast<bf16x8*>(ptr) = o;
    }
}
__global__ __launch_bounds__(256) void rows_add_kernel(const bf16* __restrict__ src, const int64_t* __restrict__ rows, bf16* __restrict__ dst, int64_t n_rows, int cols) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t r = rows[i];
    for (int c = lane * 8; c < cols; c += 512) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(src + i * cols + c);
        bf16x8 b = *reinterpret_cast<const bf16x8*>(dst + r * cols + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = f2bf(bf2f(a[j]) + bf2f(b[j]));
        *reinterpret_cast<bf16x8*>(dst + r * cols + c) = b;
    }
}
int obte_rope_cols_bf16(obte_bf16* x, int64_t ld, int32_t ncols, const float* cos_t, const float* sin_t, int64_t rows, int64_t T, const int32_t* pos,
                        int32_t head_dim, obte_stream s) {
    OBTE_REQUIRE(x && cos_t && sin_t && rows > 0 && ncols > 0 && ncols % 8 == 0 && ld % 8 == 0 && ld >= ncols && head_dim % 8 == 0 && ncols % head_dim == 0 && (pos || T > 0),
                 "obte_rope_cols_bf16: bad arguments");
    hipLaunchKernelGGL(rope_cols_kernel, dim3(stream_grid(rows * (ncols / 8), 256)), dim3(256), 0, (hipStream_t)s, (bf16*)x, ld, (int)ncols, cos_t, sin_t, rows, T, pos, (int)head_dim);
    OBTE_CHECK_LAUNCH("obte_rope_cols_bf16");
    return OBTE_OK;
}
int obte_rows_add_bf16(const obte_bf16* src, const int64_t* rows, obte_bf16* dst, int64_t n_rows, int32_t cols, obte_stream s) {
    OBTE_REQUIRE(src && rows && dst && n_rows > 0 && cols > 0 && cols % 8 == 0, "obte_rows_add_bf16: bad arguments");
    hipLaunchKernelGGL(rows_add_kernel, dim3((unsigned)cdiv64(n_rows, 4)), dim3(256), 0, (hipStream_t)s, (const bf16*)src, rows, (bf16*)dst, n_rows, (int)cols);
    OBTE_CHECK_LAUNCH("obte_rows_add_bf16");
    return OBTE_OK;
}
