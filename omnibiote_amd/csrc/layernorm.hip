// Weight-only LayerNorm forward/backward (training/model.py:63-72 -> F.layer_norm(x, (C,), w, None, 1e-5)).
// HBM-bound: one wave per row, 16-B loads/stores, the row is held in registers between the statistics passes
// so each element is read once and written once.  Algorithmic bytes per row: fwd 4*C (2 read + 2 written),
// bwd 8*C (dy, x, [dresid] read; dx written) + the dw partials.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int LN_BWD_MAX_BLOCKS = 512;   // rows of the fp32 partial workspace; the grid itself is 256 workgroups (see the pipelined kernel)

// FULL: cols == 512 * NCH exactly (1024, 2048: every hot-path shape).  Nothing is conditional then, so the row's loads AND the weight's
// are all in flight before the first use and the reductions go through DPP: one memory round trip per row.  (With the column tests
// hipcc waits for each 16-byte load before it issues the next and fetches the weights after the second reduction — four dependent
// round trips per row, found in the ISA in round 4.)
template <int NCH, bool FULL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                      bf16* __restrict__ y, float* __restrict__ mean,
                                                      float* __restrict__ rstd, int64_t rows, int cols, float eps) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const bf16* xr = x + row * cols;
    float v[NCH][8];
    bf16x8 wq[NCH];
    float sum = 0.f;
    if (FULL) {
        bf16x8 t[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) t[i] = *reinterpret_cast<const bf16x8*>(xr + (lane + 64 * i) * 8);
#pragma unroll
        for (int i = 0; i < NCH; ++i) wq[i] = *reinterpret_cast<const bf16x8*>(w + (lane + 64 * i) * 8);
        __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks the weight loads below the reductions)
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = bf2f(t[i][j]); sum += v[i][j]; }
    } else {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 8;
            if (c < cols) {
                const bf16x8 t = *reinterpret_cast<const bf16x8*>(xr + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[i][j] = bf2f(t[j]); sum += v[i][j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
            }
        }
    }
    const float mu = (FULL ? wave_sum_dpp(sum) : wave_sum(sum)) / (float)cols;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (FULL || c < cols) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mu; sq += d * d; }
        }
    }
    const float rs = 1.0f / sqrtf((FULL ? wave_sum_dpp(sq) : wave_sum(sq)) / (float)cols + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    bf16* yr = y + row * cols;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (FULL || c < cols) {
            const bf16x8 wv = FULL ? wq[i] : *reinterpret_cast<const bf16x8*>(w + c);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf((v[i][j] - mu) * rs * bf2f(wv[j]));
            *reinterpret_cast<bf16x8*>(yr + c) = o;
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*w ; partial dw per workgroup into ws[block][cols].
template <int NCH, bool HAS_RESID>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                      const bf16* __restrict__ w, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const bf16* __restrict__ dresid,
                                                      bf16* __restrict__ dx, float* __restrict__ ws, int64_t rows, int cols, int ws_accumulate) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);  // [4][cols]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float wv[NCH][8], dwacc[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        bf16x8 t = {};
        if (c < cols) t = *reinterpret_cast<const bf16x8*>(w + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) { wv[i][j] = bf2f(t[j]); dwacc[i][j] = 0.f; }
    }
    const float inv_c = 1.0f / (float)cols;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float xh[NCH][8], g[NCH][8];
        bf16x8 res[NCH];   // the residual gradient is fetched together with x and dy: behind the two reductions its latency was exposed once per row
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 8;
            res[i] = bf16x8{};
            if (HAS_RESID && c < cols) res[i] = *reinterpret_cast<const bf16x8*>(dresid + row * cols + c);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 8;
            if (c < cols) {
                const bf16x8 tx = *reinterpret_cast<const bf16x8*>(x + row * cols + c);
                const bf16x8 td = *reinterpret_cast<const bf16x8*>(dy + row * cols + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = bf2f(td[j]);
                    xh[i][j] = (bf2f(tx[j]) - mu) * rs;
                    g[i][j] = d * wv[i][j];
                    dwacc[i][j] += d * xh[i][j];
                    s1 += g[i][j];
                    s2 += g[i][j] * xh[i][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { xh[i][j] = 0.f; g[i][j] = 0.f; }
            }
        }
        s1 = wave_sum(s1) * inv_c;
        s2 = wave_sum(s2) * inv_c;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 8;
            if (c < cols) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float d = rs * (g[i][j] - s1 - xh[i][j] * s2);
                    if (HAS_RESID) d += bf2f(res[i][j]);
                    o[j] = f2bf(d);
                }
                *reinterpret_cast<bf16x8*>(dx + row * cols + c) = o;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < cols) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wave * cols + c + j] = dwacc[i][j];
        }
    }
    __syncthreads();
    // ws_accumulate: the partial row of this workgroup carries over from earlier calls (gradient accumulation over
    // micro-batches in fp32: one thread owns one address, the order of the additions is the order of the calls)
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float t = red[c] + red[cols + c] + red[2 * cols + c] + red[3 * cols + c];
        float* dst = ws + (int64_t)blockIdx.x * cols + c;
        *dst = ws_accumulate ? *dst + t : t;
    }
}

// ---- pipelined backward (default) -------------------------------------------------------------------------------------
// One row per wave at a time, as above, but the wave fetches row i+1 (x, dy, residual gradient, statistics) while it reduces
// and stores row i: the un-pipelined kernel is a chain of dependent load -> reduce -> store rounds per wave.  Measured at
// 8192 x 1024 with the residual gradient and the dw reduction: 24.0 us un-pipelined (512 workgroups) -> 23.3 (512), 22.3 (384),
// 21.4 (256 workgroups: eight rows per wave).  The forward got the same treatment and gained nothing (9.7 us either way at
// 512 / 1024 / 2048 workgroups: its 8192 waves are all resident at once, so it is one HBM round trip plus launch ramp as it
// stands) — not kept.  Same arithmetic per row; the weight-gradient partial sums group rows by workgroup, so their fp32
// summation order follows the grid size.
// DROP2: also write dropout(dx) (the library's counter-based mask, element (row, col) of `dc`) to dx_drop — the gradient the
// attention projection of the same block consumes (x1 = x + dropout(y W_proj^T): block.cpp), which used to be a pass of its own.
// FULL: cols == 512 * NCH exactly.  No column tests, the steady-state loop is peeled from the last row so that the prefetch is
// unconditional (a conditional prefetch makes hipcc join two wait states with vmcnt(0): the "prefetched" row was waited for before
// the current one was touched, and the kernel ran at one row in flight per wave), and the two reductions go through DPP.
template <int NCH, bool HAS_RESID, bool DROP2, bool FULL>
__global__ __launch_bounds__(256) void ln_bwd_pipe_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                           const bf16* __restrict__ w, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const bf16* __restrict__ dresid,
                                                           bf16* __restrict__ dx, float* __restrict__ ws, int64_t rows, int cols, int ws_accumulate,
                                                           bf16* __restrict__ dx_drop, DropCfg dc) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);  // [4][cols]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float wv[NCH][8], dwacc[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        bf16x8 t = {};
        if (FULL || c < cols) t = *reinterpret_cast<const bf16x8*>(w + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) { wv[i][j] = bf2f(t[j]); dwacc[i][j] = 0.f; }
    }
    const float inv_c = 1.0f / (float)cols;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t row = (int64_t)blockIdx.x * 4 + wave;
    bf16x8 cx[NCH], cd[NCH], cr[NCH], nx[NCH], nd[NCH], nr[NCH];
    float cmu = 0.f, crs = 0.f, nmu = 0.f, nrs = 0.f;
    auto fetch = [&](int64_t r, bf16x8 (&fx)[NCH], bf16x8 (&fd)[NCH], bf16x8 (&fr)[NCH], float& fmu, float& frs) {
        fmu = mean[r]; frs = rstd[r];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 8;
            fx[i] = bf16x8{}; fd[i] = bf16x8{}; fr[i] = bf16x8{};
            if (FULL || c < cols) {
                if (HAS_RESID) fr[i] = *reinterpret_cast<const bf16x8*>(dresid + r * cols + c);
                fx[i] = *reinterpret_cast<const bf16x8*>(x + r * cols + c);
                fd[i] = *reinterpret_cast<const bf16x8*>(dy + r * cols + c);
            }
        }
    };
    auto process = [&](int64_t r) {   // row r is in cx / cd / cr / cmu / crs
        const float mu = cmu, rs = crs;
        float xh[NCH][8], g[NCH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 8;
            if (FULL || c < cols) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = bf2f(cd[i][j]);
                    xh[i][j] = (bf2f(cx[i][j]) - mu) * rs;
                    g[i][j] = d * wv[i][j];
                    dwacc[i][j] += d * xh[i][j];
                    s1 += g[i][j];
                    s2 += g[i][j] * xh[i][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { xh[i][j] = 0.f; g[i][j] = 0.f; }
            }
        }
        s1 = (FULL ? wave_sum_dpp(s1) : wave_sum(s1)) * inv_c;
        s2 = (FULL ? wave_sum_dpp(s2) : wave_sum(s2)) * inv_c;
        const uint32_t rk = DROP2 ? drop_rowkey((uint64_t)r, dc) : 0u;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 8;
            if (FULL || c < cols) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float d = rs * (g[i][j] - s1 - xh[i][j] * s2);
                    if (HAS_RESID) d += bf2f(cr[i][j]);
                    o[j] = f2bf(d);
                }
                *reinterpret_cast<bf16x8*>(dx + r * cols + c) = o;
                if (DROP2) {
                    bf16x8 od;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const uint32_t bits = drop_pair_bits(rk, (uint32_t)(c >> 1) + jj);
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            od[2 * jj + e] = drop_keep_bits(bits, (uint32_t)e, dc) ? f2bf(bf2f(o[2 * jj + e]) * dc.scale) : f2bf(0.f);
                    }
                    *reinterpret_cast<bf16x8*>(dx_drop + r * cols + c) = od;
                }
            }
        }
    };
    if (row < rows) {
        fetch(row, cx, cd, cr, cmu, crs);
        for (; row + stride < rows; row += stride) {   // steady state: the next row's loads are in flight while this one is reduced and stored
            fetch(row + stride, nx, nd, nr, nmu, nrs);
            __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise issues these loads after the row's arithmetic: one row in flight per wave)
            process(row);
#pragma unroll
            for (int i = 0; i < NCH; ++i) { cx[i] = nx[i]; cd[i] = nd[i]; cr[i] = nr[i]; }
            cmu = nmu; crs = nrs;
        }
        process(row);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (FULL || c < cols) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wave * cols + c + j] = dwacc[i][j];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float t = red[c] + red[cols + c] + red[2 * cols + c] + red[3 * cols + c];
        float* dst = ws + (int64_t)blockIdx.x * cols + c;
        *dst = ws_accumulate ? *dst + t : t;
    }
}

// dw[c] = sum over the per-workgroup partial rows; 32 columns x 8 row-groups per workgroup, 128-B row segments.
// accumulate: dw = bf16(dw + bf16(sum)) — what autograd's `grad += new` would compute, without the extra launch.
__global__ __launch_bounds__(256) void ln_dw_reduce_kernel(const float* __restrict__ ws, bf16* __restrict__ dw, int nblk, int cols, int accumulate) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int b = part;
        for (; b + 24 < nblk; b += 32) {
            s0 += ws[(int64_t)b * cols + c];
            s1 += ws[(int64_t)(b + 8) * cols + c];
            s2 += ws[(int64_t)(b + 16) * cols + c];
            s3 += ws[(int64_t)(b + 24) * cols + c];
        }
        for (; b < nblk; b += 8) s0 += ws[(int64_t)b * cols + c];
    }
    red[part][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][cl];
        dw[c] = accumulate ? f2bf(bf2f(dw[c]) + bf2f(f2bf(t))) : f2bf(t);
    }
}

int nch_for(int cols) {
    const int chunks = (cols / 8 + 63) / 64;
    if (chunks <= 1) return 1;
    if (chunks <= 2) return 2;
    if (chunks <= 4) return 4;
    if (chunks <= 8) return 8;
    return 16;
}

}  // namespace

extern "C" int obte_layernorm_bwd_ws_rows(void) { return LN_BWD_MAX_BLOCKS; }

extern "C" int obte_layernorm_fwd(const obte_bf16* x, const obte_bf16* w, obte_bf16* y, float* mean, float* rstd,
                                  int64_t rows, int cols, float eps, obte_stream s) {
    OBTE_REQUIRE(x && w && y && mean && rstd, "obte_layernorm_fwd: null pointer");
    OBTE_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 4096, "obte_layernorm_fwd: cols must be a multiple of 8 and <= 4096 (got %d)", cols);
    if (rows == 0) return OBTE_OK;
    const dim3 grid((unsigned)cdiv64(rows, 4)), block(256);
    hipStream_t st = (hipStream_t)s;
    const int prof = obte_prof_begin(st, 110, rows, cols, 1);   // HBM-bound: algorithmic bytes = 4 * rows * cols
#define LN_FWD(N) do { if (cols == 512 * N) hipLaunchKernelGGL((ln_fwd_kernel<N, true>), grid, block, 0, st, (const bf16*)x, (const bf16*)w, (bf16*)y, mean, rstd, rows, cols, eps); \
                       else hipLaunchKernelGGL((ln_fwd_kernel<N, false>), grid, block, 0, st, (const bf16*)x, (const bf16*)w, (bf16*)y, mean, rstd, rows, cols, eps); } while (0)
    switch (nch_for(cols)) {
        case 1: LN_FWD(1); break;
        case 2: LN_FWD(2); break;
        case 4: LN_FWD(4); break;
        case 8: LN_FWD(8); break;
        default: LN_FWD(16); break;
    }
#undef LN_FWD
    obte_prof_end(prof, st);
    OBTE_CHECK_LAUNCH("obte_layernorm_fwd");
    return OBTE_OK;
}

extern "C" int obte_layernorm_bwd(const obte_bf16* dy, const obte_bf16* x, const obte_bf16* w, const float* mean,
                                  const float* rstd, const obte_bf16* dresid, obte_bf16* dx, obte_bf16* dw, float* ws,
                                  int64_t rows, int cols, obte_stream s) {
    return obte_layernorm_bwd_acc(dy, x, w, mean, rstd, dresid, dx, dw, ws, rows, cols, 0, s);
}

static int ln_bwd_impl(const obte_bf16* dy, const obte_bf16* x, const obte_bf16* w, const float* mean, const float* rstd,
                       const obte_bf16* dresid, obte_bf16* dx, obte_bf16* dw, float* ws, int64_t rows, int cols, int accumulate_dw,
                       int ws_acc, bool reduce, bool clear_tail, obte_stream s, obte_bf16* dx_drop = nullptr, DropCfg dc = DropCfg{0, 0, 0, 1.0f}) {
    OBTE_REQUIRE(dy && x && w && mean && rstd && dx && ws && (dw || !reduce), "obte_layernorm_bwd: null pointer");
    OBTE_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 4096, "obte_layernorm_bwd: bad shape rows=%lld cols=%d", (long long)rows, cols);
    // (one thread-safe static for the two A/B switches: the backward runs on autograd's worker threads)
    struct LnCfg { int blocks_cap, pipe_on; };
    static const LnCfg lcfg = [] {
        LnCfg c;
        const char* e = getenv("OBTE_LN_BLOCKS");   // timing experiments; the workspace admits up to LN_BWD_MAX_BLOCKS
        c.blocks_cap = e ? atoi(e) : 256;
        if (c.blocks_cap < 1 || c.blocks_cap > LN_BWD_MAX_BLOCKS) c.blocks_cap = 256;
        e = getenv("OBTE_LN_PIPE");                 // 0: the un-pipelined kernel (bitwise the same results)
        c.pipe_on = (e && e[0] == '0') ? 0 : 1;
        return c;
    }();
    const int blocks_cap = lcfg.blocks_cap;
    const int nblk = (int)(cdiv64(rows, 4) < blocks_cap ? cdiv64(rows, 4) : blocks_cap);
    const dim3 grid(nblk), block(256);
    const size_t smem = (size_t)4 * cols * sizeof(float);
    hipStream_t st = (hipStream_t)s;
    if (clear_tail && nblk < LN_BWD_MAX_BLOCKS &&
        hipMemsetAsync(ws + (size_t)nblk * cols, 0, (size_t)(LN_BWD_MAX_BLOCKS - nblk) * cols * sizeof(float), st) != hipSuccess) {
        obte_set_error("obte_layernorm_bwd_partial: memset failed");
        return OBTE_ELAUNCH;
    }
    const int prof = obte_prof_begin(st, 111, rows, cols, dresid ? 1 : 0);   // algorithmic bytes = (6 + 2 * has_resid) * rows * cols
    const int pipe_on = lcfg.pipe_on;
    const bool drop2 = dx_drop != nullptr && dc.thresh16 != 0;
#define LN_BWD_GO(K, ...) hipLaunchKernelGGL((K), grid, block, smem, st, (const bf16*)dy, (const bf16*)x, (const bf16*)w, mean, rstd, __VA_ARGS__)
#define LN_PIPE(N, R, D, F) LN_BWD_GO((ln_bwd_pipe_kernel<N, R, D, F>), (const bf16*)(R ? dresid : nullptr), (bf16*)dx, ws, rows, cols, ws_acc, (bf16*)(D ? dx_drop : nullptr), dc)
#define LN_PIPE_F(N, R, D) do { if (cols == 512 * N) LN_PIPE(N, R, D, true); else LN_PIPE(N, R, D, false); } while (0)
#define LN_BWD(N)                                                                                                           \
    do {                                                                                                                    \
        if (drop2) {                                                                                                        \
            if (dresid) LN_PIPE_F(N, true, true); else LN_PIPE_F(N, false, true);                                           \
        } else if (pipe_on) {                                                                                               \
            if (dresid) LN_PIPE_F(N, true, false); else LN_PIPE_F(N, false, false);                                         \
        } else {                                                                                                            \
            if (dresid) LN_BWD_GO((ln_bwd_kernel<N, true>), (const bf16*)dresid, (bf16*)dx, ws, rows, cols, ws_acc);        \
            else LN_BWD_GO((ln_bwd_kernel<N, false>), (const bf16*)nullptr, (bf16*)dx, ws, rows, cols, ws_acc);             \
        }                                                                                                                   \
    } while (0)
    switch (nch_for(cols)) {
        case 1: LN_BWD(1); break;
        case 2: LN_BWD(2); break;
        case 4: LN_BWD(4); break;
        case 8: LN_BWD(8); break;
        default: LN_BWD(16); break;
    }
#undef LN_BWD
#undef LN_PIPE_F
#undef LN_PIPE
#undef LN_BWD_GO
    if (!reduce) obte_prof_end(prof, st);
    OBTE_CHECK_LAUNCH("obte_layernorm_bwd");
    if (reduce) {
        // a carried-over workspace is summed over ALL its rows (workgroups of earlier calls may have used more of them)
        hipLaunchKernelGGL(ln_dw_reduce_kernel, dim3((cols + 31) / 32), dim3(256), 0, st, (const float*)ws, (bf16*)dw,
                           ws_acc || clear_tail ? LN_BWD_MAX_BLOCKS : nblk, cols, accumulate_dw);
        obte_prof_end(prof, st);
        OBTE_CHECK_LAUNCH("obte_layernorm_bwd(dw reduce)");
    }
    return OBTE_OK;
}

extern "C" int obte_layernorm_bwd_acc(const obte_bf16* dy, const obte_bf16* x, const obte_bf16* w, const float* mean,
                                      const float* rstd, const obte_bf16* dresid, obte_bf16* dx, obte_bf16* dw, float* ws,
                                      int64_t rows, int cols, int accumulate_dw, obte_stream s) {
    return ln_bwd_impl(dy, x, w, mean, rstd, dresid, dx, dw, ws, rows, cols, accumulate_dw, 0, true, false, s);
}

extern "C" int obte_layernorm_bwd_partial(const obte_bf16* dy, const obte_bf16* x, const obte_bf16* w, const float* mean,
                                          const float* rstd, const obte_bf16* dresid, obte_bf16* dx, obte_bf16* dw, float* partials,
                                          int64_t rows, int cols, int mode, obte_stream s) {
    OBTE_REQUIRE(mode >= OBTE_LN_PARTIAL_FIRST && mode <= OBTE_LN_PARTIAL_LAST, "obte_layernorm_bwd_partial: bad mode %d", mode);
    return ln_bwd_impl(dy, x, w, mean, rstd, dresid, dx, dw, partials, rows, cols, 0, mode != OBTE_LN_PARTIAL_FIRST,
                       mode == OBTE_LN_PARTIAL_LAST, mode == OBTE_LN_PARTIAL_FIRST, s);
}

extern "C" int obte_layernorm_bwd_dropout(const obte_bf16* dy, const obte_bf16* x, const obte_bf16* w, const float* mean,
                                          const float* rstd, const obte_bf16* dresid, obte_bf16* dx, obte_bf16* dx_dropped, obte_bf16* dw,
                                          float* ws_or_partials, int64_t rows, int cols, int partial_mode, int accumulate_dw,
                                          float p, uint64_t seed, int32_t site, obte_stream s) {
    OBTE_REQUIRE(dx_dropped, "obte_layernorm_bwd_dropout: null dx_dropped");
    OBTE_REQUIRE(p >= 0.f && p < 1.f, "obte_layernorm_bwd_dropout: dropout p must be in [0,1)");
    OBTE_REQUIRE(partial_mode == 0 || (partial_mode >= OBTE_LN_PARTIAL_FIRST && partial_mode <= OBTE_LN_PARTIAL_LAST), "obte_layernorm_bwd_dropout: bad partial_mode %d", partial_mode);
    if (p == 0.f) {   // nothing to drop: the plain backward, then a copy (callers normally do not come here with p = 0)
        const int rc = partial_mode ? obte_layernorm_bwd_partial(dy, x, w, mean, rstd, dresid, dx, dw, ws_or_partials, rows, cols, partial_mode, s)
                                    : obte_layernorm_bwd_acc(dy, x, w, mean, rstd, dresid, dx, dw, ws_or_partials, rows, cols, accumulate_dw, s);
        if (rc != OBTE_OK) return rc;
        if (hipMemcpyAsync(dx_dropped, dx, (size_t)rows * cols * 2, hipMemcpyDeviceToDevice, (hipStream_t)s) != hipSuccess) {
            obte_set_error("obte_layernorm_bwd_dropout: copy failed");
            return OBTE_ELAUNCH;
        }
        return OBTE_OK;
    }
    const DropCfg dc = make_drop(p, seed, (uint32_t)site);
    if (partial_mode)
        return ln_bwd_impl(dy, x, w, mean, rstd, dresid, dx, dw, ws_or_partials, rows, cols, 0, partial_mode != OBTE_LN_PARTIAL_FIRST,
                           partial_mode == OBTE_LN_PARTIAL_LAST, partial_mode == OBTE_LN_PARTIAL_FIRST, s, dx_dropped, dc);
    return ln_bwd_impl(dy, x, w, mean, rstd, dresid, dx, dw, ws_or_partials, rows, cols, accumulate_dw, 0, true, false, s, dx_dropped, dc);
}
