// Host-side sequencing of one transformer block (training/model.py:170-181): a single C call enqueues every
// kernel of the block's forward (or backward) on the caller's stream, so the Python layer crosses the
// boundary once per block and pass.  Pre-LN residual wiring:
//     x1 = x  + c_proj(attn(rope(c_attn(ln_1(x)))))        out = x1 + mlp.c_proj(gelu(c_fc(ln_2(x1))))
// The residual adds live in the GEMM epilogues, GELU in c_fc's epilogue, GELU' in the mlp.c_proj dgrad epilogue,
// the residual-gradient adds in the LayerNorm backward kernels, inverse RoPE in the attention backward epilogue.
#include "common.h"

namespace {

inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

struct ActLayout {
    int64_t mean1, rstd1, h1, qkv, lse, y, x1, mean2, rstd2, h2, hpre, hact, x1r, r_off, r_boff, r_pos, r_kr, r_qb, r_inv, dropbits, total;
    // with_bits: the attention dropout's keep bits (B H ceil(T/32) T words: 134 MB per block at B = 8, T = 4096) are part of the
    // buffer only when dropout is on; they are the LAST region, so every other offset is the same either way
    ActLayout(int64_t B, int64_t T, int C, int H, bool with_bits = true) {
        const int64_t M = B * T;
        int64_t o = 0;
        auto take = [&](int64_t bytes) { int64_t r = o; o += align256(bytes); return r; };
        mean1 = take(M * 4); rstd1 = take(M * 4);
        h1 = take(M * C * 2);
        qkv = take(M * 3 * C * 2);
        lse = take(B * H * T * 4);
        y = take(M * C * 2);
        x1 = take(M * C * 2);
        mean2 = take(M * 4); rstd2 = take(M * 4);
        h2 = take(M * C * 2);
        hpre = take(M * 4 * C * 2);
        hact = take(M * 4 * C * 2);
        x1r = take(M * C * 2);      // rows form (obte_block_desc::out_rows): x1 at the wanted positions
        // rows form with the attention's queries at the wanted positions only (rows_attn below): the tables of obte_attn_rows_prep
        r_off = take((B + 1) * 4); r_boff = take((B + 1) * 4); r_pos = take(M * 4); r_kr = take(M * 8); r_qb = take(M * 8); r_inv = take(M * 4);
        dropbits = take(with_bits ? obte_attn_drop_bits_bytes(B, T, H) : 0);   // attention dropout: the forward's keep bits for the backward
        total = o;
    }
};

struct WsLayout {
    int64_t dhpre, dh, dx1, dyattn, dqkv, delta, lnws, dym, dym2, gemmws, gemmws_bytes, attnws, attnws_bytes, total;
    WsLayout(int64_t B, int64_t T, int C, int H) {
        const int64_t M = B * T;
        int64_t o = 0;
        auto take = [&](int64_t bytes) { int64_t r = o; o += align256(bytes); return r; };
        dhpre = take(M * 4 * C * 2);
        dh = take(M * C * 2);       // dh2, later dh1
        dx1 = take(M * C * 2);
        dyattn = take(M * C * 2);
        dqkv = take(M * 3 * C * 2);
        delta = take(B * H * T * 4);
        lnws = take((int64_t)obte_layernorm_bwd_ws_rows() * C * 4);
        dym = take(M * C * 2);      // dropout-masked copies of the two incoming gradients (only touched when dropout_p > 0);
        dym2 = take(M * C * 2);     // two buffers: both stay live until the grouped weight-gradient launch at the end
        gemmws_bytes = 0;
        const int64_t shapes[4][2] = {{C, 4 * C}, {4 * C, C}, {C, C}, {3 * C, C}};   // the four weight gradients
        for (auto& sh : shapes) {
            const int64_t b = obte_gemm_workspace_bytes(sh[0], sh[1], M);
            if (b > gemmws_bytes) gemmws_bytes = b;
        }
        gemmws = take(gemmws_bytes > 0 ? gemmws_bytes : 256);
        attnws_bytes = obte_attn_bwd_ws_bytes(B, T, H, C / H);   // the one-kernel attention backward's dQ contributions (0: not applicable)
        attnws = take(attnws_bytes > 0 ? attnws_bytes : 256);
        total = o;
    }
};

int gemm(const obte_bf16* a, const obte_bf16* b, obte_bf16* d, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
         int ak, int bk, int epi, const obte_bf16* aux, obte_bf16* d2, obte_stream s, void* ws = nullptr, int64_t ws_bytes = 0,
         float drop_p = 0.f, uint64_t drop_seed = 0, int drop_site = 0, int64_t ldd = 0) {
    obte_gemm_args g = {};
    g.a = a; g.b = b; g.d = d; g.aux = aux; g.d2 = d2;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldd = ldd > 0 ? ldd : N;
    g.a_kmajor = ak; g.b_kmajor = bk; g.epilogue = epi; g.alpha = 1.0f;
    if (epi == OBTE_EPI_ADD && drop_p > 0.f) {   // residual add with dropout on the projection output
        g.epilogue = OBTE_EPI_ADD_DROPOUT; g.dropout_p = drop_p; g.dropout_seed = drop_seed; g.dropout_site = drop_site;
    }
    return obte_gemm_bf16_ws(&g, ws, ws_bytes, s);
}

// The four weight gradients of a block go out as ONE grouped launch (obte_gemm_grouped_bf16) when their 256x256 tiles
// fill the chip reasonably (>= 70 % of the CU slots of the rounds they need); tiny widths keep the per-matrix split-K
// launches.  OBTE_GROUPED_WGRAD=0/1 overrides (A/B timing, tests).
bool use_grouped_wgrad(int C, int64_t M) {
    const char* e = getenv("OBTE_GROUPED_WGRAD");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1' && M >= 128;
    if (M < 1024) return false;
    auto t = [](int64_t m, int64_t n) { return ((m + 255) / 256) * ((n + 255) / 256); };
    const int64_t tiles = t(C, 4 * C) + t(4 * C, C) + t(C, C) + t(3 * C, C);
    const int64_t rounds = (tiles + 255) / 256;
    return tiles * 10 >= rounds * 256 * 7;
}

// dropout sites of one block (csrc/common.h OBTE_SITE_*): 1 attention probabilities, 2 attention c_proj, 3 MLP c_proj
enum { SITE_RESID = 2, SITE_MLP = 3 };

int check_desc(const char* who, const obte_block_desc* d) {
    OBTE_REQUIRE(d, "%s: null descriptor", who);
    OBTE_REQUIRE(d->B > 0 && d->T > 0 && d->n_head > 0 && d->n_embd > 0, "%s: bad shape", who);
    OBTE_REQUIRE(d->n_embd % d->n_head == 0, "%s: n_embd %% n_head != 0", who);
    const int hs = d->n_embd / d->n_head;
    OBTE_REQUIRE(hs == 64 || hs == 128, "%s: head size %d unsupported by the HIP path (64 or 128)", who, hs);
    OBTE_REQUIRE(d->n_embd % 64 == 0 && d->n_embd <= 4096, "%s: n_embd must be a multiple of 64 and <= 4096", who);
    OBTE_REQUIRE(d->ln1_w && d->attn_w && d->proj_w && d->ln2_w && d->fc_w && d->mlp_w && d->rope_cos && d->rope_sin,
                 "%s: null parameter", who);
    OBTE_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, "%s: dropout p must be in [0,1)", who);
    OBTE_REQUIRE((d->out_rows == nullptr) == (d->n_out_rows == 0) && d->n_out_rows >= 0 && d->n_out_rows <= d->B * d->T, "%s: out_rows / n_out_rows inconsistent", who);
    return OBTE_OK;
}

#define TRY(x) do { int rc_ = (x); if (rc_ != OBTE_OK) return rc_; } while (0)

// rows form with the attention projection on the wanted rows only (see obte_block_fwd); OBTE_ROWS_PROJ=0 keeps the projection on
// every row (A/B timing, tests)
// ... and the attention itself with its QUERIES at the wanted rows only (keys and values of every position; common.h obte_attn_rows):
// the wanted rows' attention output is all the rest of the block reads.  Key ranges or no mask; OBTE_ROWS_ATTN=0 keeps the full attention.
// ... and c_attn split by its output thirds: keys and values for every position, queries for the wanted rows only (the same products:
// W_attn's rows 0 .. C-1 make q, C .. 3C-1 make k and v); RoPE then runs as its own small passes (on the k third by position = row % T,
// on the gathered q rows by their positions) with the arithmetic of the fused epilogue.  OBTE_ROWS_QSPLIT=0 keeps the whole c_attn.
bool rows_attn(const obte_block_desc* d);
bool rows_qsplit(const obte_block_desc* d) {
    static const bool off = [] { const char* e = getenv("OBTE_ROWS_QSPLIT"); return e && e[0] == '0'; }();
    return rows_attn(d) && !off;
}
bool rows_proj(const obte_block_desc* d);
bool rows_attn(const obte_block_desc* d) {
    static const bool off = [] { const char* e = getenv("OBTE_ROWS_ATTN"); return e && e[0] == '0'; }();
    return rows_proj(d) && d->mask == nullptr && !off;
}
bool rows_proj(const obte_block_desc* d) {
    static const bool off = [] { const char* e = getenv("OBTE_ROWS_PROJ"); return e && e[0] == '0'; }();
    return d->out_rows != nullptr && !off;
}

}  // namespace

extern "C" int64_t obte_block_act_bytes(int64_t B, int64_t T, int32_t n_embd, int32_t n_head) {
    return ActLayout(B, T, n_embd, n_head).total;
}
extern "C" int64_t obte_block_act_bytes_p(int64_t B, int64_t T, int32_t n_embd, int32_t n_head, float dropout_p) {
    return ActLayout(B, T, n_embd, n_head, dropout_p > 0.f).total;
}
extern "C" int64_t obte_block_bwd_ws_bytes(int64_t B, int64_t T, int32_t n_embd, int32_t n_head) {
    return WsLayout(B, T, n_embd, n_head).total;
}

extern "C" int obte_block_fwd(const obte_block_desc* d, const obte_bf16* x, obte_bf16* y_out, void* act, obte_stream s) {
    TRY(check_desc("obte_block_fwd", d));
    OBTE_REQUIRE(x && y_out && act, "obte_block_fwd: null pointer");
    const int C = d->n_embd, H = d->n_head, hs = C / H;
    const int64_t M = d->B * d->T;
    const ActLayout L(d->B, d->T, C, H);
    char* A = (char*)act;
    float *mean1 = (float*)(A + L.mean1), *rstd1 = (float*)(A + L.rstd1), *mean2 = (float*)(A + L.mean2), *rstd2 = (float*)(A + L.rstd2);
    obte_bf16 *h1 = (obte_bf16*)(A + L.h1), *qkv = (obte_bf16*)(A + L.qkv), *yat = (obte_bf16*)(A + L.y), *x1 = (obte_bf16*)(A + L.x1),
              *h2 = (obte_bf16*)(A + L.h2), *hpre = (obte_bf16*)(A + L.hpre), *hact = (obte_bf16*)(A + L.hact);
    float* lse = (float*)(A + L.lse);

    TRY(obte_layernorm_fwd(x, d->ln1_w, h1, mean1, rstd1, M, C, 1e-5f, s));
    const bool q_split = rows_qsplit(d);
    if (q_split) {   // keys and values of every position (the k third rotated in place); the queries follow below, for the wanted rows only
        TRY(gemm(h1, d->attn_w + (int64_t)C * C, qkv + C, M, 2 * C, C, C, C, 1, 1, OBTE_EPI_NONE, nullptr, nullptr, s, nullptr, 0, 0.f, 0, 0, 3 * (int64_t)C));
        TRY(obte_rope_cols_bf16(qkv + C, 3 * (int64_t)C, C, d->rope_cos, d->rope_sin, M, d->T, nullptr, hs, s));
    } else
    {   // c_attn with RoPE on its q and k thirds fused in the epilogue (model.py:102-108)
        obte_gemm_args g = {};
        g.a = h1; g.b = d->attn_w; g.d = qkv;
        g.M = M; g.N = 3 * C; g.K = C; g.lda = C; g.ldb = C; g.ldd = 3 * C;
        g.a_kmajor = 1; g.b_kmajor = 1; g.epilogue = OBTE_EPI_ROPE_QK; g.alpha = 1.0f;
        g.rope_cos = d->rope_cos; g.rope_sin = d->rope_sin; g.rope_T = d->T; g.rope_head_dim = hs;
        TRY(obte_gemm_bf16(&g, s));
    }
    const bool r_attn = rows_attn(d);
    obte_attn_rows ar = {};
    if (r_attn) {   // tables of the row set (q_off, positions, the rows' key ranges, the keys' row bounds, inverse index): once per call
        ar.q_off = (const int32_t*)(A + L.r_off); ar.q_blk_off = (const int32_t*)(A + L.r_boff); ar.q_pos = (const int32_t*)(A + L.r_pos); ar.n = d->n_out_rows;
        ar.key_ranges = d->key_ranges ? (const int32_t*)(A + L.r_kr) : nullptr;
        ar.query_bounds = d->key_ranges ? (const int32_t*)(A + L.r_qb) : nullptr;
        TRY(obte_attn_rows_prep(d->out_rows, d->n_out_rows, d->B, d->T, d->key_ranges, (int32_t*)(A + L.r_off), (int32_t*)(A + L.r_boff), (int32_t*)(A + L.r_pos),
                                (int32_t*)(A + L.r_kr), (int32_t*)(A + L.r_qb), (int32_t*)(A + L.r_inv), s));
    }
    obte_attn_fwd_args af = {};
    af.qkv = qkv; af.o = yat; af.lse = lse; af.key_ranges = d->key_ranges; af.mask = d->mask;
    af.mask_sb = d->mask_sb; af.mask_sh = d->mask_sh; af.mask_sq = d->mask_sq;
    af.B = d->B; af.T = d->T; af.n_head = H; af.head_dim = hs; af.scale = 8.0f / (float)C;  // model.py:119
    af.dropout_p = d->dropout_p; af.dropout_seed = d->dropout_seed;
    af.drop_bits = (d->dropout_p > 0.f && !r_attn) ? (uint32_t*)(A + L.dropbits) : nullptr;   // (a gathered query set hashes in both passes)
    if (af.drop_bits) {   // words of key tiles the forward skips (pairs the mask excludes) stay defined whoever reads them
        if (hipMemsetAsync(af.drop_bits, 0, (size_t)obte_attn_drop_bits_bytes(d->B, d->T, H), (hipStream_t)s) != hipSuccess) {
            obte_set_error("obte_block_fwd: memset of the dropout keep bits failed");
            return OBTE_ELAUNCH;
        }
    }
    af.ranges_exact = d->ranges_exact;
    if (r_attn) {   // Q of the wanted rows gathered into the region of the full attention output (not formed in this form); the rows'
                    // attention output lands where the projection below expects its gathered input: the region of the full x1
        obte_bf16* qr = yat;
        if (q_split) {   // q = ln_1(x) W_q^T for the wanted rows, rotated at their positions (ln_1's rows gathered into the region x1r fills later)
            obte_bf16* h1r = (obte_bf16*)(A + L.x1r);
            TRY(obte_rows_gather_bf16(h1, d->out_rows, h1r, d->n_out_rows, M, C, s));
            TRY(gemm(h1r, d->attn_w, qr, d->n_out_rows, C, C, C, C, 1, 1, OBTE_EPI_NONE, nullptr, nullptr, s, (void*)(A + L.hpre), M * 4 * C * 2));
            TRY(obte_rope_cols_bf16(qr, C, C, d->rope_cos, d->rope_sin, d->n_out_rows, d->T, (const int32_t*)(A + L.r_pos), hs, s));
        } else {
            TRY(obte_rows_gather_strided_bf16(qkv, 3 * (int64_t)C, d->out_rows, qr, d->n_out_rows, C, s));
        }
        af.o = x1;
        TRY(obte_attn_fwd_rows(&af, &ar, qr, s));
    } else {
        TRY(obte_attn_fwd(&af, s));
    }
    // the attention projection and the MLP half: on every position, or (rows form) on the n wanted positions only — per-position
    // arithmetic, same results there.  Rows form without dropout: the projection too runs on the wanted rows (the attention output
    // and the block input gathered; x1 = x + y W_proj^T formed for those rows alone; the region of the full x1 keeps the gathered
    // attention output for the backward).  With dropout the projection's mask is defined on whole activations: all rows, then gather.
    int64_t Mm = M;
    const obte_bf16* x1m = x1;
    if (rows_proj(d)) {
        obte_bf16* x1r = (obte_bf16*)(A + L.x1r);
        obte_bf16* yr = x1;
        Mm = d->n_out_rows; x1m = x1r;
        TRY(obte_rows_gather_bf16(x, d->out_rows, x1r, Mm, M, C, s));
        if (!r_attn) TRY(obte_rows_gather_bf16(yat, d->out_rows, yr, Mm, M, C, s));   // (rows_attn: the attention wrote the wanted rows there itself)
        if (d->dropout_p > 0.f) {   // the projection's dropout mask (site 2) is defined on the whole activation: element (rows[i], c) for gathered row i
            obte_bf16* pr = (obte_bf16*)(A + L.h2);   // (ln_2's output region: written below)
            TRY(gemm(yr, d->proj_w, pr, Mm, C, C, C, C, 1, 1, OBTE_EPI_NONE, nullptr, nullptr, s, (void*)(A + L.hpre), M * 4 * C * 2));
            TRY(obte_dropout_rows_bf16(pr, x1r, x1r, d->out_rows, Mm, C, d->dropout_p, d->dropout_seed, SITE_RESID, s));
        } else {
            TRY(gemm(yr, d->proj_w, x1r, Mm, C, C, C, C, 1, 1, OBTE_EPI_ADD, x1r, nullptr, s, (void*)(A + L.hpre), M * 4 * C * 2));   // (split-K workspace: the MLP's regions are not written yet)
        }
    } else {
        TRY(gemm(yat, d->proj_w, x1, M, C, C, C, C, 1, 1, OBTE_EPI_ADD, x, nullptr, s, nullptr, 0, d->dropout_p, d->dropout_seed, SITE_RESID));
        if (d->out_rows) {
            obte_bf16* x1r = (obte_bf16*)(A + L.x1r);
            TRY(obte_rows_gather_bf16(x1, d->out_rows, x1r, d->n_out_rows, M, C, s));
            Mm = d->n_out_rows; x1m = x1r;
        }
    }
    TRY(obte_layernorm_fwd(x1m, d->ln2_w, h2, mean2, rstd2, Mm, C, 1e-5f, s));
    TRY(gemm(h2, d->fc_w, hpre, Mm, 4 * C, C, C, C, 1, 1, OBTE_EPI_GELU, nullptr, hact, s));
    // rows form: [n, C] over K = 4C is a handful of tiles — split-K, with the unused tail of the (M-row) hpre region as its workspace
    void* fws = nullptr;
    int64_t fws_bytes = 0;
    if (d->out_rows) {
        const int64_t used = align256(Mm * 4 * C * 2);
        fws = (void*)(A + L.hpre + used);
        fws_bytes = M * 4 * C * 2 - used;
        if (fws_bytes < (int64_t)(2 * Mm * C * 4)) { fws = nullptr; fws_bytes = 0; }
    }
    TRY(gemm(hact, d->mlp_w, y_out, Mm, C, 4 * C, 4 * C, 4 * C, 1, 1, OBTE_EPI_ADD, x1m, nullptr, s, fws, fws_bytes, d->dropout_p, d->dropout_seed, SITE_MLP));
    return OBTE_OK;
}

extern "C" int obte_block_bwd_acc(const obte_block_desc* d, const obte_bf16* x, const obte_bf16* dy, const void* act, void* ws,
                                  obte_bf16* dx, obte_bf16* dln1_w, obte_bf16* dattn_w, obte_bf16* dproj_w, obte_bf16* dln2_w,
                                  obte_bf16* dfc_w, obte_bf16* dmlp_w, int accumulate_matrices, obte_stream s);

extern "C" int obte_block_bwd(const obte_block_desc* d, const obte_bf16* x, const obte_bf16* dy, const void* act, void* ws,
                              obte_bf16* dx, obte_bf16* dln1_w, obte_bf16* dattn_w, obte_bf16* dproj_w, obte_bf16* dln2_w,
                              obte_bf16* dfc_w, obte_bf16* dmlp_w, obte_stream s) {
    return obte_block_bwd_acc(d, x, dy, act, ws, dx, dln1_w, dattn_w, dproj_w, dln2_w, dfc_w, dmlp_w, 0, s);
}

extern "C" int obte_block_bwd_acc(const obte_block_desc* d, const obte_bf16* x, const obte_bf16* dy, const void* act, void* ws,
                                  obte_bf16* dx, obte_bf16* dln1_w, obte_bf16* dattn_w, obte_bf16* dproj_w, obte_bf16* dln2_w,
                                  obte_bf16* dfc_w, obte_bf16* dmlp_w, int accumulate_matrices, obte_stream s) {
    // accumulate_matrices: bit 0 = the four matrices, bit 1 = the two LayerNorm weights: dW += ... straight into the .grad buffers
    const int acc_ln = (accumulate_matrices & 2) ? 1 : 0;
    accumulate_matrices &= 1;
    const int wepi = accumulate_matrices ? OBTE_EPI_ADD : OBTE_EPI_NONE;
    TRY(check_desc("obte_block_bwd", d));
    OBTE_REQUIRE(x && dy && act && ws && dx && dln1_w && dattn_w && dproj_w && dln2_w && dfc_w && dmlp_w, "obte_block_bwd: null pointer");
    const int C = d->n_embd, H = d->n_head, hs = C / H;
    const int64_t M = d->B * d->T;
    const ActLayout L(d->B, d->T, C, H);
    const WsLayout W(d->B, d->T, C, H);
    const char* A = (const char*)act;
    char* S = (char*)ws;
    const float *mean1 = (const float*)(A + L.mean1), *rstd1 = (const float*)(A + L.rstd1), *mean2 = (const float*)(A + L.mean2),
                *rstd2 = (const float*)(A + L.rstd2), *lse = (const float*)(A + L.lse);
    const obte_bf16 *h1 = (const obte_bf16*)(A + L.h1), *qkv = (const obte_bf16*)(A + L.qkv), *yat = (const obte_bf16*)(A + L.y),
                    *x1 = (const obte_bf16*)(A + L.x1), *h2 = (const obte_bf16*)(A + L.h2), *hpre = (const obte_bf16*)(A + L.hpre),
                    *hact = (const obte_bf16*)(A + L.hact);
    obte_bf16 *dhpre = (obte_bf16*)(S + W.dhpre), *dh = (obte_bf16*)(S + W.dh), *dx1 = (obte_bf16*)(S + W.dx1),
              *dyattn = (obte_bf16*)(S + W.dyattn), *dqkv = (obte_bf16*)(S + W.dqkv);
    float *delta = (float*)(S + W.delta), *lnws = (float*)(S + W.lnws);
    void* gws = W.gemmws_bytes > 0 ? (void*)(S + W.gemmws) : nullptr;

    obte_bf16* dym = (obte_bf16*)(S + W.dym);
    obte_bf16* dym2 = (obte_bf16*)(S + W.dym2);
    const bool drop = d->dropout_p > 0.f;
    // rows form (the model's last block): only the attention half's two weight gradients are left for the grouped launch — 64
    // tiles of K = M that keep a quarter of the chip busy for the whole launch while the input gradient's 512 short tiles finish on
    // the rest in a fifth of the time (686 us for what three balanced launches do in ~430: round-5 profile) — so they go out as
    // their own split-K launches with the tuned plans.  OBTE_GROUPED_LAST=1 restores the grouped form (A/B timing).
    static const bool grouped_last = [] { const char* e = getenv("OBTE_GROUPED_LAST"); return e && e[0] == '1'; }();
    const bool grouped_ok = use_grouped_wgrad(C, M);
    const bool grouped = grouped_ok && (d->out_rows == nullptr || (grouped_last && !rows_proj(d)));   // (the projection on the wanted rows is its own pair of launches)
    // rows form: the MLP half ran on Mm = n_out_rows positions (dy is [Mm, C]); its two weight gradients contract over those rows
    // and go out as their own launches, the grouped launch keeps the attention half's
    const bool rows_form = d->out_rows != nullptr;
    const bool rows_p = rows_proj(d);   // (implies rows_form, no dropout, and — below — the ungrouped form of the attention half)
    const bool r_attn = rows_attn(d);   // (implies rows_p: the attention's queries were the wanted rows only)
    const bool q_split = rows_qsplit(d);   // (implies r_attn: c_attn ran by its output thirds)
    const int64_t Mm = rows_form ? d->n_out_rows : M;
    const obte_bf16* x1m = rows_form ? (const obte_bf16*)(A + L.x1r) : x1;
    const bool grouped_mlp = grouped && !rows_form;
    // MLP: out = x1 + dropout(hact W_mlp^T): the projection sees dy masked by the same (seed, site 3) mask
    const obte_bf16* dy_mlp = dy;
    if (drop && d->dy_masked) {   // handed over by the block above: its last LayerNorm backward wrote dropout(dx) under this block's mask
        dy_mlp = d->dy_masked;
    } else if (drop) {   // (rows form: dy and the mask of site 3 are [Mm, C] — element (i, c) of the compact output, as in the forward)
        TRY(obte_dropout_bf16(dy, dym, Mm * C, C, d->dropout_p, d->dropout_seed, SITE_MLP, s));
        dy_mlp = dym;
    }
    TRY(gemm(dy_mlp, d->mlp_w, dhpre, Mm, 4 * C, C, C, 4 * C, 1, 0, OBTE_EPI_GELU_BWD, hpre, nullptr, s));      // dhpre = (dy W_mlp) * gelu'(h): hpre holds the derivative
    // rows form with enough rows for the grouped kernel's K: the MLP half's two weight gradients (K = Mm) share one launch below
    const bool pair_mlp = rows_form && grouped_ok && Mm >= 256;
    if (!grouped_mlp && !pair_mlp) TRY(gemm(dy_mlp, hact, dmlp_w, C, 4 * C, Mm, C, 4 * C, 0, 0, wepi, accumulate_matrices ? dmlp_w : nullptr, nullptr, s, gws, W.gemmws_bytes));               // dW_mlp = dy^T hact
    TRY(gemm(dhpre, d->fc_w, dh, Mm, C, 4 * C, 4 * C, C, 1, 0, OBTE_EPI_NONE, nullptr, nullptr, s, rows_form ? gws : nullptr, rows_form ? W.gemmws_bytes : 0));   // dh2 = dhpre W_fc (rows form: few tiles over K = 4C, split-K)
    if (!grouped_mlp && !pair_mlp) TRY(gemm(dhpre, h2, dfc_w, 4 * C, C, Mm, 4 * C, C, 0, 0, wepi, accumulate_matrices ? dfc_w : nullptr, nullptr, s, gws, W.gemmws_bytes));               // dW_fc = dhpre^T h2
    if (pair_mlp) {
        obte_gemm_args gp[2] = {};
        auto putp = [&](int i, const obte_bf16* a, const obte_bf16* b, obte_bf16* dw, int64_t m, int64_t n) {
            gp[i].a = a; gp[i].b = b; gp[i].d = dw; gp[i].aux = accumulate_matrices ? dw : nullptr;
            gp[i].M = m; gp[i].N = n; gp[i].K = Mm; gp[i].lda = m; gp[i].ldb = n; gp[i].ldd = n;
            gp[i].a_kmajor = 0; gp[i].b_kmajor = 0; gp[i].epilogue = wepi; gp[i].alpha = 1.0f;
        };
        putp(0, dhpre, h2, dfc_w, 4 * C, C);
        putp(1, dy_mlp, hact, dmlp_w, C, 4 * C);
        TRY(obte_gemm_grouped_bf16(gp, 2, s));
    }
    const int lnp = d->ln_partial_mode;
    if (lnp) OBTE_REQUIRE(d->ln1_partials && d->ln2_partials && lnp >= OBTE_LN_PARTIAL_FIRST && lnp <= OBTE_LN_PARTIAL_LAST,
                          "obte_block_bwd: ln_partial_mode needs both partial buffers and a valid mode");
    // dx1 = dy + LN2'(dh2); attention: x1 = x + dropout(y W_proj^T), so its projection sees dx1 under the (seed, site 2) mask:
    // with dropout on, the LayerNorm backward writes that masked copy beside dx1 (it used to be a pass of its own)
    const obte_bf16* dx1_proj = dx1;
    if (drop && !rows_form) {
        TRY(obte_layernorm_bwd_dropout(dh, x1, d->ln2_w, mean2, rstd2, dy, dx1, dym2, dln2_w, lnp ? d->ln2_partials : lnws, M, C, lnp, acc_ln,
                                       d->dropout_p, d->dropout_seed, SITE_RESID, s));
        dx1_proj = dym2;
    } else if (rows_form) {   // d x1 at the wanted rows ([Mm, C], staged in dyattn — free until the projection's input gradient), then scattered into zeros
        obte_bf16* dx1r = dyattn;
        if (lnp) TRY(obte_layernorm_bwd_partial(dh, x1m, d->ln2_w, mean2, rstd2, dy, dx1r, dln2_w, d->ln2_partials, Mm, C, lnp, s));
        else TRY(obte_layernorm_bwd_acc(dh, x1m, d->ln2_w, mean2, rstd2, dy, dx1r, dln2_w, lnws, Mm, C, acc_ln, s));
        TRY(obte_rows_scatter_bf16(dx1r, d->out_rows, dx1, Mm, M, C, s));
        if (drop && !rows_p) {   // the attention projection sees d x1 under the (seed, site 2) mask, which is defined on whole activations
            TRY(obte_dropout_bf16(dx1, dym2, M * C, C, d->dropout_p, d->dropout_seed, SITE_RESID, s));
            dx1_proj = dym2;
        }
        if (rows_p) {   // the projection ran on the wanted rows: its two gradients contract over / are formed for those rows only
            const obte_bf16* yr = x1;                                   // the gathered attention output (forward)
            obte_bf16* dyr = dym;                                       // d(attention output) at the wanted rows (dym: the MLP half's products above were its last readers)
            const obte_bf16* dxp = dx1r;
            if (drop) {   // d x1 of the wanted rows under the projection's mask (element (rows[i], c)), staged in dym2 (free: dq below is written after)
                TRY(obte_dropout_rows_bf16(dx1r, nullptr, dym2, d->out_rows, Mm, C, d->dropout_p, d->dropout_seed, SITE_RESID, s));
                dxp = dym2;
            }
            TRY(gemm(dxp, d->proj_w, dyr, Mm, C, C, C, C, 1, 0, OBTE_EPI_NONE, nullptr, nullptr, s, gws, W.gemmws_bytes));
            TRY(gemm(dxp, yr, dproj_w, C, C, Mm, C, C, 0, 0, wepi, accumulate_matrices ? dproj_w : nullptr, nullptr, s, gws, W.gemmws_bytes));   // dW_proj = dx1^T y over the wanted rows
            if (!r_attn) TRY(obte_rows_scatter_bf16(dyr, d->out_rows, dyattn, Mm, M, C, s));   // (dx1r, staged in dyattn, has been read by both products; rows_attn: the attention backward takes the gathered rows as they are)
        }
    } else if (lnp) {
        TRY(obte_layernorm_bwd_partial(dh, x1, d->ln2_w, mean2, rstd2, dy, dx1, dln2_w, d->ln2_partials, M, C, lnp, s));
    } else {
        TRY(obte_layernorm_bwd_acc(dh, x1, d->ln2_w, mean2, rstd2, dy, dx1, dln2_w, lnws, M, C, acc_ln, s));
    }
    bool delta_ready = false;
    if (!rows_p) {   // dy_attn = dx1 W_proj — where structure 7 takes the shape, with the softmax backward's delta = rowsum(dy_attn o y) formed in
                     // its epilogue (the attention backward's prep launch would otherwise read both tensors again to form it)
        obte_gemm_args g = {};
        g.a = dx1_proj; g.b = d->proj_w; g.d = dyattn; g.M = M; g.N = C; g.K = C; g.lda = C; g.ldb = C; g.ldd = C;
        g.a_kmajor = 1; g.b_kmajor = 0; g.epilogue = OBTE_EPI_NONE; g.alpha = 1.0f;
        const int rcd = obte_gemm_rowdot_bf16(&g, yat, delta, d->T, hs, s);
        if (rcd == OBTE_OK) delta_ready = true;
        else if (rcd == 1) TRY(obte_gemm_bf16(&g, s));
        else return rcd;
    }
    if (!grouped && !rows_p) TRY(gemm(dx1_proj, yat, dproj_w, C, C, M, C, C, 0, 0, wepi, accumulate_matrices ? dproj_w : nullptr, nullptr, s, gws, W.gemmws_bytes));                       // dW_proj = dx1^T y
    obte_attn_bwd_args ab = {};
    ab.qkv = qkv; ab.o = yat; ab.d_o = dyattn; ab.lse = lse; ab.delta = delta; ab.dqkv = dqkv;
    ab.rope_cos = d->rope_cos; ab.rope_sin = d->rope_sin;
    ab.key_ranges = d->key_ranges; ab.mask = d->mask; ab.mask_sb = d->mask_sb; ab.mask_sh = d->mask_sh; ab.mask_sq = d->mask_sq;
    ab.query_bounds = d->query_bounds;
    ab.ranges_exact = d->ranges_exact;
    ab.B = d->B; ab.T = d->T; ab.n_head = H; ab.head_dim = hs; ab.scale = 8.0f / (float)C;
    ab.dropout_p = d->dropout_p; ab.dropout_seed = d->dropout_seed;
    ab.drop_bits = (d->dropout_p > 0.f && !r_attn) ? (const uint32_t*)(A + L.dropbits) : nullptr;
    if (W.attnws_bytes > 0) { ab.ws = (void*)(S + W.attnws); ab.ws_bytes = W.attnws_bytes; }
    if (r_attn) {   // queries at the wanted rows: everything on the query side is the gathered set (forward: Q rows in the region of the
                    // full attention output, the rows' output in the region of the full x1; the tables of the row set in the buffer's tail)
        obte_attn_rows ar = {};
        ar.q_off = (const int32_t*)(A + L.r_off); ar.q_blk_off = (const int32_t*)(A + L.r_boff); ar.q_pos = (const int32_t*)(A + L.r_pos); ar.n = d->n_out_rows;
        ar.key_ranges = d->key_ranges ? (const int32_t*)(A + L.r_kr) : nullptr;
        ar.query_bounds = d->key_ranges ? (const int32_t*)(A + L.r_qb) : nullptr;
        ab.o = x1; ab.d_o = dym;
        obte_bf16* dqr = dym2;
        TRY(obte_attn_bwd_rows(&ab, &ar, yat, dqr, s));
        if (q_split) {   // c_attn's backward by thirds: dK / dV of every position against W's k and v rows, dQ of the wanted rows against its q rows
            const obte_bf16* dkv = dqkv + C;
            const obte_bf16* w_kv = d->attn_w + (int64_t)C * C;
            obte_bf16* tmp = dym;        // (d(attention output) of the wanted rows has been read by the attention backward)
            obte_bf16* h1r = dyattn;     // (d x1 of the wanted rows has been read by the projection's products)
            TRY(gemm(dkv, w_kv, dh, M, C, 2 * C, 3 * (int64_t)C, C, 1, 0, OBTE_EPI_NONE, nullptr, nullptr, s));                                   // dh1 = [dK dV] W_kv
            TRY(gemm(dqr, d->attn_w, tmp, Mm, C, C, C, C, 1, 0, OBTE_EPI_NONE, nullptr, nullptr, s, gws, W.gemmws_bytes));                         //       + dQ W_q at the wanted rows
            TRY(obte_rows_add_bf16(tmp, d->out_rows, dh, Mm, C, s));
            TRY(obte_rows_gather_bf16(h1, d->out_rows, h1r, Mm, M, C, s));
            obte_bf16* dw_kv = dattn_w + (int64_t)C * C;
            TRY(gemm(dkv, h1, dw_kv, 2 * C, C, M, 3 * (int64_t)C, C, 0, 0, wepi, accumulate_matrices ? dw_kv : nullptr, nullptr, s, gws, W.gemmws_bytes));   // dW_kv = [dK dV]^T ln_1(x)
            TRY(gemm(dqr, h1r, dattn_w, C, C, Mm, C, C, 0, 0, wepi, accumulate_matrices ? dattn_w : nullptr, nullptr, s, gws, W.gemmws_bytes));             // dW_q = dQ^T ln_1(x) over the wanted rows
        } else {
            TRY(obte_rows_fill_strided_bf16(dqr, (const int32_t*)(A + L.r_inv), dqkv, M, 3 * (int64_t)C, C, s));   // dqkv's q third: the rows' dQ, zeros elsewhere
        }
    } else {
        TRY(delta_ready ? obte_attn_bwd_delta_ready(&ab, s) : obte_attn_bwd(&ab, s));
    }
    // OBTE_GROUPED_DGRAD=0 keeps dh1 = dqkv W_attn as its own launch (A/B timing)
    const char* gd = getenv("OBTE_GROUPED_DGRAD");
    const bool group_dgrad = grouped && !(gd && gd[0] == '0');
    if (!group_dgrad && !q_split) TRY(gemm(dqkv, d->attn_w, dh, M, C, 3 * C, 3 * C, C, 1, 0, OBTE_EPI_NONE, nullptr, nullptr, s));             // dh1 = dqkv W_attn
    if (!grouped && !q_split) TRY(gemm(dqkv, h1, dattn_w, 3 * C, C, M, 3 * C, C, 0, 0, wepi, accumulate_matrices ? dattn_w : nullptr, nullptr, s, gws, W.gemmws_bytes));               // dW_attn = dqkv^T h1
    if (grouped) {
        // One grid: dW_fc = dhpre^T h2, dW_mlp = dy^T hact, dW_attn = dqkv^T h1, dW_proj = dx1^T y (K = tokens, full K
        // per tile) and, on the CUs those tiles leave idle, dh1 = dqkv W_attn (K = 3C).
        obte_gemm_args gs[5] = {};
        auto put = [&](int i, const obte_bf16* a, const obte_bf16* b, obte_bf16* dw, int64_t m, int64_t n) {
            gs[i].a = a; gs[i].b = b; gs[i].d = dw; gs[i].aux = accumulate_matrices ? dw : nullptr;
            gs[i].M = m; gs[i].N = n; gs[i].K = M; gs[i].lda = m; gs[i].ldb = n; gs[i].ldd = n;
            gs[i].a_kmajor = 0; gs[i].b_kmajor = 0; gs[i].epilogue = wepi; gs[i].alpha = 1.0f;
        };
        int np = 0;
        if (grouped_mlp) {
            put(np++, dhpre, h2, dfc_w, 4 * C, C);
            put(np++, dy_mlp, hact, dmlp_w, C, 4 * C);
        }
        put(np++, dqkv, h1, dattn_w, 3 * C, C);
        put(np++, dx1_proj, yat, dproj_w, C, C);
        if (group_dgrad) {
            gs[np].a = dqkv; gs[np].b = d->attn_w; gs[np].d = dh; gs[np].M = M; gs[np].N = C; gs[np].K = 3 * C;
            gs[np].lda = 3 * C; gs[np].ldb = C; gs[np].ldd = C; gs[np].a_kmajor = 1; gs[np].b_kmajor = 0;
            gs[np].epilogue = OBTE_EPI_NONE; gs[np].alpha = 1.0f;
            ++np;
        }
        TRY(obte_gemm_grouped_bf16(gs, np, s));
    }
    // dx = dx1 + LN1'(dh1); with dropout and a block below, also dropout(dx) under that block's (seed, site 3) mask
    if (drop && d->dx_masked)
        TRY(obte_layernorm_bwd_dropout(dh, x, d->ln1_w, mean1, rstd1, dx1, dx, d->dx_masked, dln1_w, lnp ? d->ln1_partials : lnws, M, C, lnp, acc_ln,
                                       d->dropout_p, d->dx_mask_seed, SITE_MLP, s));
    else if (lnp) TRY(obte_layernorm_bwd_partial(dh, x, d->ln1_w, mean1, rstd1, dx1, dx, dln1_w, d->ln1_partials, M, C, lnp, s));
    else TRY(obte_layernorm_bwd_acc(dh, x, d->ln1_w, mean1, rstd1, dx1, dx, dln1_w, lnws, M, C, acc_ln, s));
    return OBTE_OK;
}
