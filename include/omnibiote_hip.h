/* omnibiote_hip.h — C ABI of libomnibiote_hip.so: the MI355X (gfx950) implementation of the OmniBioTE
 * encoder-training hot path.
 *
 * The reference (nyuolab/OmniBioTE) is pure Python and defines no FFI of its own; each entry point below
 * names the reference code it replaces (paths relative to the reference tree).  Conventions:
 *   - every function returns 0 on success or a negative OBTE_E* code; obte_last_error() gives the text
 *     (thread-local).  Nothing here allocates, frees or synchronises: all buffers (incl. workspaces) are
 *     caller-owned device memory, borrowed for the duration of the call, and kernels are enqueued on the
 *     caller's stream (a hipStream_t passed as void*; NULL = the null stream).
 *   - bf16 tensors are passed as uint16_t* (bit pattern of bfloat16), dense row-major unless a stride
 *     argument says otherwise.  "rows" always means tokens (B*T).
 *   - re-entrant: callable from any thread (autograd's backward workers call in without the GIL).  Process-wide state is
 *     the thread-local error string plus two mutex-protected tables: the tuned GEMM plan table (obte_gemm_plan_set /
 *     _clear; read by every GEMM launch) and the opt-in launch profiler's records (obte_profile_*).  Nothing else persists
 *     between calls — except the device status word below, which is sticky until read.
 */
#ifndef OMNIBIOTE_HIP_H
#define OMNIBIOTE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t obte_bf16;
typedef void* obte_stream;

enum {
    OBTE_OK = 0,
    OBTE_EINVAL = -1,   /* bad shape / alignment / null pointer */
    OBTE_ELAUNCH = -2,  /* hipLaunch failure */
    OBTE_EUNSUPPORTED = -3
};

int obte_abi_version(void);
const char* obte_last_error(void);
/* sizeof of the public argument structs in declaration order (gemm, attn_fwd, attn_bwd, mt, block_desc): lets a
 * binding written in another language verify its struct layout at load time.  Returns the number of structs. */
int obte_struct_sizes(int64_t* out, int cap);

/* ---- device status (failures a kernel detects after it was launched) --------------------------------------------
 * A kernel cannot return an error code.  The ones that can detect an unrecoverable condition at run time OR a bit into one
 * word of pinned host memory; obte_device_status() returns that word (0 = nothing happened) and, with clear != 0, resets the
 * bits it returned.  It is a plain host read: call it where the work in question has already been synchronised with (after
 * the loss has been copied back, after a stream / device synchronise) — a set bit means the results of the launch that set it
 * are INVALID, and obte_last_error() then says which condition it was.  Negative: the word could not be allocated.
 * Replaces nothing in the reference (PyTorch raises from its own kernels' asserts the same way: at the next synchronise). */
enum {
    OBTE_STATUS_ATTN_BWD_HANDOFF = 1   /* obte_attn_bwd, one-kernel form: a bounded wait of the dQ hand-off chain gave up */
};
int obte_device_status(int clear);
/* Test hook (never set by the product): make the next launches of a kernel fail in a chosen way so that the reporting above can
 * be exercised.  what = 0: off; 1: the one-kernel attention backward never signals ONE hand-off counter and bounds its waits
 * at 2^10 polls instead of 2^20.  Process-wide; returns the previous value. */
int obte_fault_inject(int what);

/* ---- opt-in launch profiler (measurement only; off by default) ----------------------------------------------
 * While enabled, every obte_gemm_bf16 / obte_attn_fwd / obte_attn_bwd call is bracketed by two hipEvents on the
 * caller's stream.  obte_profile_collect synchronises those events and returns up to cap records:
 * ms[i] = elapsed milliseconds, dims[3*i..] = (M,N,K) for a GEMM or (B*H, T, head_dim) for attention,
 * kind[i] = a_kmajor*8 + b_kmajor*4 + epilogue (+ 32.. for a grouped launch) + 1000 * kernel structure (1 gemm_bf16_kernel,
 * 2 gemm_v2_kernel, 3 gemm_v3_kernel) for a GEMM, 100 = attention forward, 101 = attention backward; the HBM-bound
 * kernels record (rows, cols, flag): 110 LayerNorm forward, 111 LayerNorm backward (flag = residual gradient added),
 * 112 masked CE over (n_rows, vocab), 113 AdamW over (elements, 1, 1).
 * Returns the number of records written (records are cleared). */
int obte_profile_enable(int on);
int obte_profile_collect(double* ms, int64_t* dims, int32_t* kind, int cap);

/* ---- LayerNorm, weight only, eps inside (training/model.py:63-72; F.layer_norm) ------------------------- */
/* y = (x-mean)*rstd*w ; saves mean,rstd (fp32, one per row) for backward.  cols % 8 == 0, cols <= 4096. */
int obte_layernorm_fwd(const obte_bf16* x, const obte_bf16* w, obte_bf16* y, float* mean, float* rstd,
                       int64_t rows, int cols, float eps, obte_stream s);
/* dx = LN'(dy) (+ dresid if non-null); dw = sum_rows dy*xhat.  ws: fp32 [obte_layernorm_bwd_ws_rows()*cols]. */
int obte_layernorm_bwd_ws_rows(void);
int obte_layernorm_bwd(const obte_bf16* dy, const obte_bf16* x, const obte_bf16* w, const float* mean,
                       const float* rstd, const obte_bf16* dresid, obte_bf16* dx, obte_bf16* dw, float* ws,
                       int64_t rows, int cols, obte_stream s);
/* Same, with dw accumulated in place when accumulate_dw != 0: dw = bf16(dw + bf16(sum)) — the arithmetic of autograd's
 * `param.grad += new` (train_encoder.py:462, gradient accumulation over micro-batches) without its extra launch. */
int obte_layernorm_bwd_acc(const obte_bf16* dy, const obte_bf16* x, const obte_bf16* w, const float* mean,
                       const float* rstd, const obte_bf16* dresid, obte_bf16* dx, obte_bf16* dw, float* ws,
                       int64_t rows, int cols, int accumulate_dw, obte_stream s);

/* Weight gradient accumulated over several calls in fp32 (gradient accumulation over micro-batches, train_encoder.py:284-311):
 * `partials` is a caller-owned fp32 [obte_layernorm_bwd_ws_rows(), cols] buffer that persists between the calls.
 *   OBTE_LN_PARTIAL_FIRST  partials  = this call's per-workgroup sums (unused rows zeroed); dw untouched (may be NULL)
 *   OBTE_LN_PARTIAL_MORE   partials += this call's sums; dw untouched (may be NULL)
 *   OBTE_LN_PARTIAL_LAST   partials += this call's sums, then dw = bf16(sum of all partial rows)  (overwrites dw)
 * One reduction launch per optimizer step instead of one per micro-batch, and the sum over micro-batches is formed in fp32
 * (autograd's `grad += new` rounds to bf16 after every micro-batch).  dx as in obte_layernorm_bwd. */
enum { OBTE_LN_PARTIAL_FIRST = 1, OBTE_LN_PARTIAL_MORE = 2, OBTE_LN_PARTIAL_LAST = 3 };
int obte_layernorm_bwd_partial(const obte_bf16* dy, const obte_bf16* x, const obte_bf16* w, const float* mean,
                               const float* rstd, const obte_bf16* dresid, obte_bf16* dx, obte_bf16* dw, float* partials,
                               int64_t rows, int cols, int mode, obte_stream s);

/* The same backward with a second output: dx_dropped = dropout(dx), element (row, col) of (seed, site) as obte_dropout_bf16
 * forms it — the masked gradient the attention projection of the same block consumes (x1 = x + dropout(y W_proj^T),
 * model.py:151,179) without a pass of its own.  partial_mode 0: ws_or_partials is the fp32 scratch of obte_layernorm_bwd_acc
 * (accumulate_dw as there); else OBTE_LN_PARTIAL_* with ws_or_partials the persistent partial buffer. */
int obte_layernorm_bwd_dropout(const obte_bf16* dy, const obte_bf16* x, const obte_bf16* w, const float* mean,
                               const float* rstd, const obte_bf16* dresid, obte_bf16* dx, obte_bf16* dx_dropped, obte_bf16* dw,
                               float* ws_or_partials, int64_t rows, int cols, int partial_mode, int accumulate_dw,
                               float p, uint64_t seed, int32_t site, obte_stream s);

/* ---- bf16 GEMM on MFMA, fp32 accumulate (nn.Linear fwd/dgrad/wgrad: training/model.py:102,151,163,166,253)
 * D[M,N] = epilogue(alpha * sum_k A(m,k) * B(n,k)).
 *   a_kmajor=1: A(m,k) = a[m*lda + k]   (k contiguous)      a_kmajor=0: A(m,k) = a[k*lda + m]
 *   b_kmajor=1: B(n,k) = b[n*ldb + k]                        b_kmajor=0: B(n,k) = b[k*ldb + n]
 * forward  y = x W^T      : A=x (kmajor), B=W (kmajor)
 * dgrad    dx = dy W      : A=dy (kmajor), B=W (k = out-feature rows: b_kmajor=0)
 * wgrad    dW = dy^T x    : A=dy (a_kmajor=0, k = tokens), B=x (b_kmajor=0)
 * Requirements: lda, ldb, ldd % 8 == 0; a k-major operand needs K % 64 == 0 (a k-strided one may have any K:
 * the tail is zero-filled by the buffer bounds check); operands are dense (total size rows*ld).
 */
enum {
    OBTE_EPI_NONE = 0,      /* d = bf16(alpha*acc) */
    OBTE_EPI_GELU = 1,      /* h = bf16(acc); d = bf16(gelu'(h)) ; d2 = bf16(gelu_erf_1.41421(h))   (model.py:23-25,163-165):
                               the forward stores the activation and its derivative from one erf evaluation */
    OBTE_EPI_ADD = 2,       /* d = bf16(aux + bf16(alpha*acc))  residual add (model.py:179-180); aux may alias d
                               (gradient accumulation in place) */
    OBTE_EPI_GELU_BWD = 3,  /* d = bf16(bf16(acc) * aux)   aux = the derivative stored by EPI_GELU */
    OBTE_EPI_ROPE_QK = 5,   /* d = packed c_attn output with RoPE applied to its q and k thirds (model.py:102-108): N = 3C,
                               pairs (2j,2j+1) of each head rotated by (rope_cos, rope_sin)[row % rope_T][j], fp32 [T, hs/2];
                               all-zero sin = the reference's cos-only bf16 mode */
    OBTE_EPI_ADD_DROPOUT = 4 /* d = bf16(aux + dropout(bf16(acc)))   resid_dropout / mlp dropout (model.py:151,167);
                                element (m,n) is dropout element (row m, col n) of (dropout_seed, dropout_site) */
};
typedef struct {
    const obte_bf16* a; const obte_bf16* b; obte_bf16* d;
    const obte_bf16* aux;   /* [M,N] ld = ldd, for EPI_ADD / EPI_GELU_BWD */
    obte_bf16* d2;          /* [M,N] ld = ldd, for EPI_GELU */
    int64_t M, N, K;
    int64_t lda, ldb, ldd;
    int32_t a_kmajor, b_kmajor;
    int32_t epilogue;
    float alpha;
    float dropout_p; int32_t dropout_site; uint64_t dropout_seed;   /* EPI_ADD_DROPOUT only */
    const float* rope_cos; const float* rope_sin; int64_t rope_T; int32_t rope_head_dim;   /* EPI_ROPE_QK only */
} obte_gemm_args;
int obte_gemm_bf16(const obte_gemm_args* g, obte_stream s);
/* Same, with a caller-owned scratch buffer that enables split-K (fp32 partial tiles summed in a fixed order by a
 * second kernel) when the output has too few tiles to fill 256 CUs — the weight-gradient shapes.
 * obte_gemm_workspace_bytes returns the size that lets the library split as it prefers (0 = no split wanted);
 * a NULL or smaller workspace simply disables the split.  Split-K needs epilogue NONE or ADD and ldd == N. */
int64_t obte_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int obte_gemm_bf16_ws(const obte_gemm_args* g, void* workspace, int64_t workspace_bytes, obte_stream s);
/* Tuned plans.  The library holds five GEMM structures (variant 1: 128x128 tiles, two workgroups per CU; 2: the K-tile ring,
 * 256x128 / 256x192 / 256x256 tiles, one workgroup per CU, optional split-K; 3: the half-tile ring, 256x256; 4: the half-tile
 * ring at 256x128 and two workgroups per CU; 7: the 256x256 half-tile ring as a persistent kernel whose operand stream runs on
 * across tiles — whole tiles, at least one per CU, no split-K, the x W^T and dy W layouts; numbers 5 and 6 were structures
 * that lost every comparison and were removed); which is fastest depends on the shape
 * (tile quantisation against 256 CUs, K length, where the operands are served from).  A host-side tuner times the candidates once
 * per (layout, epilogue, M, N, K) and records the winner here; unknown shapes fall back to a built-in heuristic, and a plan a shape
 * cannot run (a borrowed near-match, edge tiles) falls back the same way.
 * obte_gemm_workspace_bytes_max: a workspace size that admits any recordable plan. */
int obte_gemm_plan_set(int a_kmajor, int b_kmajor, int epilogue, int64_t M, int64_t N, int64_t K, int variant, int bn,
                       int splits);
int obte_gemm_plan_clear(void);
int64_t obte_gemm_workspace_bytes_max(int64_t M, int64_t N, int64_t K);

/* Grouped launch: `count` (1..OBTE_GROUP_MAX) independent GEMMs in a single grid of 256x256 tiles, each tile running its
 * full K (>= 128) — no split-K workspace, no reduce launches.  The problems may MIX layouts (a_kmajor / b_kmajor per
 * problem), alpha and the two admissible epilogues (OBTE_EPI_NONE overwrite, OBTE_EPI_ADD accumulate into aux == d): tiles
 * are dealt so that every XCD gets its share of the long-K problems first.  Replaces, in one call, the four weight-gradient
 * products autograd issues for the nn.Linear layers of one block (training/model.py:102,151,163,166 under loss.backward(),
 * train_encoder.py:462: dW_mlp, dW_fc, dW_proj, dW_attn, K = tokens) together with the c_attn input gradient (K = 3C) on
 * the CUs those leave idle; and the readout's input gradient beside its weight gradient (model.py:253). */
#define OBTE_GROUP_MAX 6
int obte_gemm_grouped_bf16(const obte_gemm_args* gs, int count, obte_stream s);

/* ---- dropout (training/model.py:83-84,160,204) -------------------------------------------------------------------
 * Every dropout site of the path is a matrix and draws its mask from one counter-based generator: element (row, col) of
 * site `site` is kept iff the 16 bits it owns of hash32(rowkey(seed, site, row) ^ (col >> 1)) are >= p * 2^16 (two columns
 * share one 32-bit hash, a row's key is formed once; csrc/common.h drop_rowkey / drop_keep); kept values are scaled by
 * 1/(1-p) and rounded to bf16.  Forward and backward regenerate the mask from (seed, site) — nothing is stored.  Sites:
 * 0 embedding output (row = token position, col = feature), 1 attention probabilities (row = (b*H + h)*T + query,
 * col = key), 2 attention c_proj output, 3 MLP c_proj output (row = token position, col = feature), 7 free for callers.
 * The RNG stream necessarily differs from PyTorch's.
 * obte_dropout_bf16: out = dropout(in) on a flat [n / cols, cols] matrix (in may alias out); cols % 8 == 0. */
int obte_dropout_bf16(const obte_bf16* in, obte_bf16* out, int64_t n, int64_t cols, float p, uint64_t seed, int32_t site, obte_stream s);

/* ---- RoPE on the q and k thirds of a packed qkv activation, in place (training/model.py:39-50,108) ------
 * qkv: [rows = B*T, 3*C]; pairs (2j,2j+1) of each head; position = row % T.  cos/sin: fp32 [T, hs/2].
 * sin all-zero reproduces the reference's degenerate bf16 mode.  inverse=1 applies the
 * transpose (backward). */
int obte_rope_qk_inplace(obte_bf16* qkv, const float* cos_t, const float* sin_t, int64_t B, int64_t T,
                         int n_head, int head_dim, int inverse, obte_stream s);

/* ---- fused attention (training/model.py:115-148): softmax(q k^T * scale + mask) v, non-causal -------------
 * q,k,v are read from the packed [B*T, 3C] qkv buffer (row stride 3C; head h at column h*hs, k at +C, v at
 * +2C) and o is written as [B*T, C] (heads side by side: the "merge heads" copy of model.py:148 is free).
 * Mask, one of:  none;  key ranges int32 [B,T,2] = [k_start,k_end) per query (the block-diagonal masks of
 * train_encoder.py:25-57);  dense additive bf16 with element strides (mask_sb, mask_sh, mask_sq; key stride 1;
 * mask_sh = 0 for the reference's expand() view).  lse: fp32 [B,H,T] (natural log, of the scaled scores).
 * head_dim in {64,128}.  */
typedef struct {
    const obte_bf16* qkv; obte_bf16* o; float* lse;
    const int32_t* key_ranges; const obte_bf16* mask; int64_t mask_sb, mask_sh, mask_sq;
    int64_t B, T; int32_t n_head, head_dim; float scale;
    float dropout_p; uint64_t dropout_seed;     /* attention-probability dropout (site 1); p = 0 disables */
    const int32_t* ranges_exact;                /* nullable; with mask + key_ranges: the device flag obte_mask_bounds wrote */
    /* nullable, dropout only (head_dim 128, key ranges or no mask): uint32 [B*H][ceil(T/32)][T] (obte_attn_drop_bits_bytes) — the
     * keep decisions the forward takes, written once in key-major order: bit i of word (b*H+h, t, key) = keep(query 32 t + i, key).
     * Handed to obte_attn_bwd, its key-major kernel reads one word per key and 32 queries instead of hashing per element. */
    uint32_t* drop_bits;
} obte_attn_fwd_args;
int obte_attn_fwd(const obte_attn_fwd_args* a, obte_stream s);

typedef struct {
    const obte_bf16* qkv; const obte_bf16* o; const obte_bf16* d_o; const float* lse;
    float* delta;          /* workspace fp32 [B,H,T] */
    obte_bf16* dqkv;       /* out, packed like qkv (all three thirds written) */
    const float* rope_cos; const float* rope_sin;  /* nullable pair, fp32 [T, hs/2]: if given, dq and dk are
                                                      returned already multiplied by the transpose of the RoPE
                                                      map (i.e. gradients w.r.t. the un-rotated c_attn output) */
    const int32_t* key_ranges; const obte_bf16* mask; int64_t mask_sb, mask_sh, mask_sq;
    int64_t B, T; int32_t n_head, head_dim; float scale;
    float dropout_p; uint64_t dropout_seed;     /* must equal the forward call's */
    const int32_t* query_bounds;                /* nullable; with a dense mask: int32 [B,T,2] per KEY, see obte_mask_bounds */
    const int32_t* ranges_exact;                /* nullable; with mask + key_ranges + query_bounds: obte_mask_bounds' flag */
    /* optional scratch of obte_attn_bwd_ws_bytes() bytes: with it, head_dim 128, no dense mask and either dropout_p = 0 or the
     * forward's keep bits in drop_bits, the backward runs as ONE kernel that forms each score tile once (five MFMA products per tile instead of the seven of the
     * dQ + dK/dV kernel pair): per-key-block fp32 contributions to dQ land in the scratch and are summed in key-block order
     * (no atomics: bitwise reproducible).  NULL / too small: the two-kernel form. */
    void* ws; int64_t ws_bytes;
    const uint32_t* drop_bits;                  /* nullable: what the forward call wrote (obte_attn_fwd_args::drop_bits), same p and seed */
} obte_attn_bwd_args;
int obte_attn_bwd(const obte_attn_bwd_args* a, obte_stream s);
int64_t obte_attn_drop_bits_bytes(int64_t B, int64_t T, int32_t n_head);
int64_t obte_attn_bwd_ws_bytes(int64_t B, int64_t T, int32_t n_head, int32_t head_dim);
/* A/B switch (process-wide, measurement only): 0 = automatic (the one-kernel form wherever it applies), 1 = always the
 * two-kernel form.  Returns the previous value.  The environment variable OBTE_ATTN_BWD=two sets 1 at load time. */
int obte_attn_bwd_select(int mode);

/* Conservative bounds of a dense additive mask, so that the dense-mask kernels can skip key tiles without changing a
 * single value (the arithmetic still reads the mask element by element).  An entry counts as "masking" when it is
 * <= -3e4 (its softmax weight underflows to exactly 0 beside any row that has one allowed key).
 *   key_bounds[b,q]   = [first, last+1) over the keys some head allows for query q; (0,T) if any head's row allows none
 *                       (the reference then gets a softmax over the raw scores, so nothing may be skipped);
 *   query_bounds[b,k] = [first, last+1) over the queries that may give key k a non-zero weight (rows of the previous
 *                       kind count for every key).
 * Pass key_bounds as `key_ranges` TOGETHER with `mask` to obte_attn_fwd / obte_attn_bwd / the block descriptor (a mask
 * with key_ranges means "dense arithmetic, ranges only bound the loops"), and query_bounds in the field of that name.
 * ranges_exact (nullable, int32 [1] on the device; needs col_scratch int32 [B*T]): set to 1 iff the mask IS a range mask —
 * every row's allowed keys are one contiguous run of exact zeros, the same for all heads, everything else <= -3e4, and
 * every key's queries are one contiguous run too (block-diagonal document masks, key-padding masks).  Passed on in the
 * `ranges_exact` fields, it lets the attention entry points run the range kernels for such a mask: the kernels hold both
 * bodies and branch on the flag, so the host never reads it.
 * row_scratch: uint8 [B*T].  Replaces nothing in the reference (training/train_encoder.py:31-57 builds the mask, and
 * model.py:115-146 hands it to SDPA whole). */
int obte_mask_bounds(const obte_bf16* mask, int64_t mask_sb, int64_t mask_sh, int64_t mask_sq, int64_t B, int32_t n_head,
                     int64_t T, int32_t* key_bounds, int32_t* query_bounds, uint8_t* row_scratch, int32_t* ranges_exact,
                     int32_t* col_scratch, obte_stream s);

/* ---- token embedding (training/model.py:203,241) ------------------------------------------------------------
 * fwd: out[r,:] = wte[idx[r],:].  bwd: dwte (dense [V,C], fully written) = scatter-add of dout rows, summed in
 * fp32 in a fixed order (deterministic).  order = a stable argsort of idx (int32 [rows]); ws: fp32
 * [2*ceil(rows/32)*C + ...] see obte_embedding_bwd_ws_bytes. */
int obte_embedding_fwd(const int64_t* idx, const obte_bf16* wte, obte_bf16* out, int64_t rows, int cols,
                       int64_t vocab, obte_stream s);
/* with the embedding dropout of model.py:242 fused (site 0) */
int obte_embedding_fwd_dropout(const int64_t* idx, const obte_bf16* wte, obte_bf16* out, int64_t rows, int cols,
                               int64_t vocab, float p, uint64_t seed, obte_stream s);
int64_t obte_embedding_bwd_ws_bytes(int64_t rows, int cols);
int obte_embedding_bwd(const int64_t* idx, const int32_t* order, const obte_bf16* dout, obte_bf16* dwte,
                       void* ws, int64_t rows, int cols, int64_t vocab, obte_stream s);
/* accumulate != 0: dwte holds an existing gradient; only the touched rows are read-modified-written (no memset). */
int obte_embedding_bwd_acc(const int64_t* idx, const int32_t* order, const obte_bf16* dout, obte_bf16* dwte,
                           void* ws, int64_t rows, int cols, int64_t vocab, int accumulate, obte_stream s);
/* dout is the gradient of the DROPPED embedding output: the same mask (p, seed, site 0) is re-applied while summing */
int obte_embedding_bwd_dropout(const int64_t* idx, const int32_t* order, const obte_bf16* dout, obte_bf16* dwte,
                               void* ws, int64_t rows, int cols, int64_t vocab, int accumulate, float p, uint64_t seed,
                               obte_stream s);

/* ---- masked-LM cross entropy, forward + backward in one pass (training/train_encoder.py:301-305) -------------
 * loss_sum[0] += sum over rows with mlm_mask!=0 of (logsumexp(logits[r]) - logits[r,target[r]]) * row_scale
 * dlogits[r,:] = (softmax(logits[r]) - onehot(target[r])) * row_scale * grad_scale[0]  for masked rows, else 0,
 * where row_scale = 1/n_accum and grad_scale points at a device fp32 (1/mask_count), so no host sync is needed
 * (obte_masked_ce_rows also accepts grad_scale = NULL, meaning 1: its caller knows the count and folds it into row_scale).
 * vocab % 8 == 0, vocab <= 65536*2. */
int obte_masked_ce_fwd_bwd(const obte_bf16* logits, const int64_t* target, const uint8_t* mlm_mask,
                           const float* grad_scale, float row_scale, float* loss_sum, float* row_loss,
                           obte_bf16* dlogits, int64_t rows, int64_t vocab, obte_stream s);

/* Same, for a dlogits buffer that is reused across calls: prev_mask (nullable) is the mlm_mask of the previous call that
 * wrote this buffer; with prev_mask = NULL every row is written (use that for the first call on a fresh buffer).  Rows
 * unmasked both then and now are left untouched: they already hold zeros. */
int obte_masked_ce_fwd_bwd_reuse(const obte_bf16* logits, const int64_t* target, const uint8_t* mlm_mask,
                                 const uint8_t* prev_mask, const float* grad_scale, float row_scale, float* row_loss,
                                 obte_bf16* dlogits, int64_t rows, int64_t vocab, obte_stream s);

/* Compact form: the loss looks at the MLM-masked positions only (loss *= mask, train_encoder.py:304), so every other row of
 * d(logits) is an exact zero.  row_index: int64 [n_rows], ascending positions (rows of the dense [total_rows, vocab]
 * logits) that are masked; target is indexed by position; row_loss [n_rows] and dlogits_rows [n_rows, vocab] are
 * compact.  The readout's backward then contracts over n_rows instead of total_rows — the zero rows it leaves out
 * contribute nothing to either gradient.  row_scale_vec (nullable, fp32 [n_rows]): an extra weight per listed row, for a call
 * that covers several micro-batches, each normalised by its own count of masked tokens (train_encoder.py:305).
 * row_index = NULL (then n_rows == total_rows): logits [n_rows, vocab] and target [n_rows] already hold the listed rows alone
 * — the readout that computes the masked positions only.  dlogits_rows must not alias logits. */
int obte_masked_ce_rows(const obte_bf16* logits, const int64_t* target, const int64_t* row_index, const float* grad_scale,
                        float row_scale, const float* row_scale_vec, float* row_loss, obte_bf16* dlogits_rows, int64_t n_rows,
                        int64_t total_rows, int64_t vocab, obte_stream s);

/* ---- fused AdamW step, bf16 params/grads/moments as the reference trains (train_encoder.py:170,199,316-317) ---
 * One launch per tensor: p -= lr*(m_hat/(sqrt(v_hat)+eps) + wd*p), grads pre-multiplied by clip_coef[0]
 * (device fp32, 1.0 if no clipping).  step is 1-based.  */
int obte_adamw_bf16(obte_bf16* p, const obte_bf16* g, obte_bf16* m, obte_bf16* v, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int32_t step,
                    const float* clip_coef, obte_stream s);
/* sum of squares of a bf16 tensor accumulated into out[0] (fp32, atomics) — for clip_grad_norm_. */
int obte_sumsq_bf16(const obte_bf16* g, int64_t n, float* out, obte_stream s);
/* Multi-tensor forms: one launch covers up to OBTE_MT_MAX tensors (the optimizer step of the small config touches
 * 67 tensors, most of them 1 K-element LayerNorm weights: per-tensor launches cost more than the arithmetic).
 * Each tensor's n must be a multiple of 8. */
#define OBTE_MT_MAX 32
typedef struct {
    obte_bf16* p[OBTE_MT_MAX]; const obte_bf16* g[OBTE_MT_MAX]; obte_bf16* m[OBTE_MT_MAX]; obte_bf16* v[OBTE_MT_MAX];
    int64_t n[OBTE_MT_MAX]; float lr[OBTE_MT_MAX]; float weight_decay[OBTE_MT_MAX]; int32_t step[OBTE_MT_MAX];
    int32_t count;
    /* obte_adamw_multi_bf16_ref only: the same two hyper-parameters in double (0 = use the float field).  Python holds lr and
     * weight_decay as doubles and torch forms 1 - lr*wd and lr / bias_correction1 from them in double before rounding to fp32;
     * passing them through the float fields first can move step_size by one fp32 ulp. */
    double lr64[OBTE_MT_MAX]; double weight_decay64[OBTE_MT_MAX];
} obte_mt_args;
int obte_adamw_multi_bf16(const obte_mt_args* t, float beta1, float beta2, float eps, const float* clip_coef, obte_stream s);
int obte_sumsq_multi_bf16(const obte_mt_args* t, float* out, obte_stream s);
/* out[i] += sum of squares of tensor i (fp32 [count]): per-tensor norms, as torch's clip_grad_norm_ forms them before it
 * combines them (it rounds each to the gradients' dtype first: train_encoder.py:316 on bf16 gradients). */
int obte_sumsq_multi_bf16_each(const obte_mt_args* t, float* out, obte_stream s);
/* The same step with the reference's own rounding sequence: torch.optim.AdamW (which MuAdamW is underneath,
 * train_encoder.py:195-199) on bf16 tensors rounds after EVERY tensor op — clip scaling, p.mul_, exp_avg.lerp_,
 * exp_avg_sq.mul_().addcmul_(), sqrt, / sqrt(bias_correction2), + eps, addcdiv_ — where obte_adamw_multi_bf16 rounds each
 * state once.  This is the form the harness uses (FusedAdamW(rounding="reference")) so that loss curves track the
 * reference's step for step; scalars (1 - lr*wd, lr/bias_correction1, sqrt(bias_correction2)) are formed in double like
 * Python does. */
int obte_adamw_multi_bf16_ref(const obte_mt_args* t, double beta1, double beta2, double eps, const float* clip_coef, obte_stream s);

/* ---- whole transformer block (training/model.py:170-181), forward and backward, dropout 0 --------------------
 * One host call enqueues every kernel of the block, so Python crosses the boundary once per block and pass.
 * Activations saved for backward live in one caller-allocated buffer of obte_block_act_bytes() bytes. */
typedef struct {
    int64_t B, T; int32_t n_embd, n_head;
    const obte_bf16 *ln1_w, *attn_w, *proj_w, *ln2_w, *fc_w, *mlp_w;   /* parameters */
    const float *rope_cos, *rope_sin;                                   /* [T, hs/2] */
    const int32_t* key_ranges; const obte_bf16* mask; int64_t mask_sb, mask_sh, mask_sq;
    float dropout_p; uint64_t dropout_seed;   /* one seed per block call; sites 1-3 derive from it.  p = 0: no dropout */
    const int32_t* query_bounds;              /* nullable: obte_mask_bounds output for a dense mask (backward only) */
    /* backward only, optional: the two LayerNorm weight gradients accumulated over micro-batches in caller-owned fp32
     * partial buffers (see obte_layernorm_bwd_partial); ln_partial_mode 0 = off, else OBTE_LN_PARTIAL_*.  With FIRST / MORE
     * dln1_w / dln2_w are not written. */
    float *ln1_partials, *ln2_partials; int32_t ln_partial_mode;
    const int32_t* ranges_exact;              /* nullable: obte_mask_bounds' flag for a dense mask (see obte_mask_bounds) */
    /* optional (the LAST block of a masked-LM step): only these positions of the block's output are wanted — the
     * loss multiplies every other position by zero (train_encoder.py:304) and nothing after the last block mixes positions.
     * out_rows: int64 [n_out_rows], ascending distinct rows of the [B*T, C] activation.  ln_1 and the k and v thirds of c_attn run
     * on every position (keys and values of all of them are needed; without a dense mask the q third is formed for the listed
     * rows only); ln_2, c_fc + GELU and mlp.c_proj run on the listed rows only: y is
     * [n_out_rows, C].  Without a dense mask the attention itself runs with its QUERIES at the listed rows only (keys and values
     * of every position; the rows' key ranges decide what each sees), and the attention projection on those rows.  Backward: dy
     * is [n_out_rows, C]; every gradient of those products is contracted over / formed for the listed rows (the rows left out
     * would contribute exact zeros); dK and dV of every position, dQ of the listed rows (zeros elsewhere) feed c_attn's backward
     * as usual.  With a dense mask the attention runs on every position as in the whole block and its output is gathered.
     * Dropout: sites 1 (attention probabilities) and 2 (attention projection) keep the masks they have in the whole block — a
     * listed row's mask elements are those of its position; site 3 (the MLP projection) masks element (i, c) of the
     * [n_out_rows, C] output, forward and backward alike.  NULL / 0: the whole block on every position. */
    const int64_t* out_rows; int64_t n_out_rows;
    /* backward only, optional, dropout only (p > 0): the hand-off of the masked gradient between consecutive blocks.  Block i's
     * MLP projection needs dy under ITS (seed, site 3) mask, and dy is the dx of block i + 1 — so block i + 1's last LayerNorm
     * backward can write that masked copy beside dx instead of block i spending a pass on it.
     *   dy_masked      in : dropout(dy) under this block's (dropout_seed, site 3) mask, same shape as dy (NULL: computed here);
     *   dx_masked      out: dropout(dx) under (dx_mask_seed, site 3) — the seed of the block BELOW — [B*T, C] (NULL: not written). */
    const obte_bf16* dy_masked; obte_bf16* dx_masked; uint64_t dx_mask_seed;
} obte_block_desc;
int64_t obte_block_act_bytes(int64_t B, int64_t T, int32_t n_embd, int32_t n_head);
/* the same for a known dropout probability: with p = 0 the buffer ends before the attention dropout's keep bits (the largest
 * single region at long T); the other offsets do not move, so a buffer of obte_block_act_bytes() serves any p */
int64_t obte_block_act_bytes_p(int64_t B, int64_t T, int32_t n_embd, int32_t n_head, float dropout_p);
int64_t obte_block_bwd_ws_bytes(int64_t B, int64_t T, int32_t n_embd, int32_t n_head);
int obte_block_fwd(const obte_block_desc* d, const obte_bf16* x, obte_bf16* y, void* act, obte_stream s);
/* grads of the six parameters are written (not accumulated) to d*_w; dx to dx. */
int obte_block_bwd(const obte_block_desc* d, const obte_bf16* x, const obte_bf16* dy, const void* act, void* ws,
                   obte_bf16* dx, obte_bf16* dln1_w, obte_bf16* dattn_w, obte_bf16* dproj_w, obte_bf16* dln2_w,
                   obte_bf16* dfc_w, obte_bf16* dmlp_w, obte_stream s);

/* Same as obte_block_bwd; accumulate_matrices is a bit set: with bit 0 the four weight-matrix gradients are ADDED to
 * the contents of d*_w (bf16(old + bf16(new)), what autograd's accumulation would produce) instead of overwriting
 * them — gradient accumulation over micro-batches without the separate read-modify-write pass; with bit 1 the same
 * for the two LayerNorm weight gradients (dln1_w, dln2_w).  Unset bits: overwrite. */
int obte_block_bwd_acc(const obte_block_desc* d, const obte_bf16* x, const obte_bf16* dy, const void* act, void* ws,
                       obte_bf16* dx, obte_bf16* dln1_w, obte_bf16* dattn_w, obte_bf16* dproj_w, obte_bf16* dln2_w,
                       obte_bf16* dfc_w, obte_bf16* dmlp_w, int accumulate_matrices, obte_stream s);

/* rows of a [*, cols] bf16 matrix by index: dst[i] = src[rows[i]] (gather);  dst = 0, dst[rows[i]] = src[i] (scatter into
 * a zeroed [total_rows, cols]; rows ascending and distinct).  cols % 8 == 0.  Used by the rows form of the block above. */
int obte_rows_gather_bf16(const obte_bf16* src, const int64_t* rows, obte_bf16* dst, int64_t n_rows, int64_t total_rows, int32_t cols, obte_stream s);
int obte_rows_scatter_bf16(const obte_bf16* src, const int64_t* rows, obte_bf16* dst, int64_t n_rows, int64_t total_rows, int32_t cols, obte_stream s);

#ifdef __cplusplus
}
#endif
#endif
