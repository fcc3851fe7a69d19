// bf16 GEMM, structure 8: the weight gradients of a transformer block as ONE persistent launch whose work divides evenly.
//
// Replaces what autograd issues for the four nn.Linear weights of a block under loss.backward() (training/model.py:102,151,163,166;
// train_encoder.py:462): dW = dY^T X, K = tokens, operands k-strided (a_kmajor = b_kmajor = 0).  The grouped launch of
// gemm_bf16_v2.hip gives every 256 x 256 output tile its full K: at n_embd = 1024 that is 192 tiles of 32 768 k on 256 CUs — 192 CUs
// run one 800-us tile each and 64 idle for three quarters of the launch (the short c_attn input gradient only fills 150 us of
// them): 77 % of the chip for the largest single item of the step.  Here the K range of every tile is cut into S ALIGNED parts
// (S = 4 at 192 tiles: 768 work items = three per CU, equal work), and
//   * one workgroup per CU walks its items through structure 7's continuous ring (the LDS-DMA stream runs on across items: no
//     fill, no drain between them);
//   * items are dealt part-major — round r, CU c: item 256 r + remap(c) -> (part, tile) — so that the CUs of an XCD work on
//     neighbouring tiles of the SAME part at the same k at the same time: the operand slices they share still meet in L2 (a plain
//     stream-K split would shift every workgroup's k phase and lose that);
//   * parts 0 .. S-2 of a tile leave as fp32 accumulator images in a scratch slab (the registers as they stand, one 16-byte piece per
//     lane: fully coalesced, no staging), published by the recipe the attention backward's hand-off uses (write-through sc1 stores,
//     every wave's vmcnt(0), the workgroup's barrier, one lane's agent-scope add on the tile's counter: Guideline 16, R1);
//   * part S-1 is the tile's owner: it comes last in item order (every producer of its tile belongs to an earlier item of some
//     workgroup: no wait in steady state, and no cycle whatever is resident when), polls the counter, adds the S-1 images IN PART
//     ORDER to its own accumulators (fp32, a fixed order: bitwise reproducible) and runs the epilogue — overwrite, or
//     bf16(grad + bf16(sum)) for gradient accumulation in place — through the wave-private staging of structure 7.
// A wait that gives up is reported through the library's device status word (include/omnibiote_hip.h): never a silent wrong gradient.
#include "gemm_common.h"

using namespace obte_gemm_v2;

namespace {

constexpr int V8_MAXP = OBTE_GROUP_MAX;
constexpr int V8_STG_WAVE = 4096;
constexpr int V8_SMEM = V3_RING + 8 * V8_STG_WAVE;       // 160 KiB
constexpr int64_t V8_SLAB = 256 * 256 * 4;               // one fp32 accumulator image of a tile

struct V8Problem {
    const bf16* a; const bf16* b; bf16* d; const bf16* aux;   // aux: the gradient accumulated into (== d), or null: overwrite
    int64_t lda, ldb, ldd, a_elems, b_elems;
    int tiles_m, tiles_n, first_tile;
    float alpha;
};
struct V8Params {
    V8Problem pr[V8_MAXP];
    int count, total_tiles, splits, nh_part;   // nh_part: half-steps of 32 k per item (even, >= 8)
    float* slabs;          // fp32 [total_tiles][splits - 1][256 x 256]
    int32_t* counters;     // int32 [total_tiles]: parts published per tile (zeroed by the launcher, every call)
    int32_t* status;       // the library's device status word
    int spin_limit;
};

struct V8Item { int prob; int64_t m0, n0; int part, tile; };

__device__ __forceinline__ V8Item item_of(const V8Params& P, int i) {
    V8Item it;
    it.part = i / P.total_tiles;
    it.tile = i - it.part * P.total_tiles;
    int pr = 0;
#pragma unroll
    for (int j = 1; j < V8_MAXP; ++j) pr += (j < P.count && it.tile >= P.pr[j].first_tile) ? 1 : 0;
    it.prob = pr;
    const V8Problem& q = P.pr[pr];
    const int local = it.tile - q.first_tile;
    const int group_sz = 8 * q.tiles_n;                 // 8-row groups of tiles: the tiles that run side by side share their panels
    const int first_m = (local / group_sz) * 8;
    const int gsz = min(q.tiles_m - first_m, 8);
    it.m0 = (int64_t)(first_m + (local % group_sz) % gsz) * 256;
    it.n0 = (int64_t)((local % group_sz) / gsz) * 256;
    return it;
}

__global__ __launch_bounds__(NTHREADS, 2) void gemm_v8_wgrad_kernel(V8Params P) {
    constexpr int NJ = 8;
    constexpr int WAIT_LAX = (((8 + 16) >> 4) << 14) | 0x0070 | ((8 + 16) & 15);   // vmcnt(8 + 16 stores) lgkmcnt(0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int G = (int)gridDim.x;
    const int n_items = P.total_tiles * P.splits;
    const int nh = P.nh_part;
    const int wmy = xcd_remap(blockIdx.x, G);           // the workgroups of an XCD take consecutive items of a round
    const int wm = wave >> 1, wn = wave & 1;

    // ---- issue cursor (four half-steps ahead of the MFMAs): running element offsets of the two k-strided operands ---------------
    int i_i = wmy, u_i = 0;
    bool live_i = i_i < n_items;
    int voff_a[2], voff_b[2];
    int64_t ao_i = 0, bo_i = 0, step_a = 0, step_b = 0, ae_i = 0, be_i = 0;
    const bf16 *pa_i = nullptr, *pb_i = nullptr;
    auto cursor_item = [&]() {
        const V8Item it = item_of(P, i_i);
        const V8Problem& q = P.pr[it.prob];
        const int64_t k0 = (int64_t)it.part * nh * 32;
        pa_i = q.a; pb_i = q.b; ae_i = q.a_elems; be_i = q.b_elems;
        ao_i = k0 * q.lda + it.m0; bo_i = k0 * q.ldb + it.n0;
        step_a = 32 * q.lda; step_b = 32 * q.ldb;
        dma_offsets_h<false>(wave, lane, q.lda, voff_a);   // (the leading dimensions differ between the problems of a group)
        dma_offsets_h<false>(wave, lane, q.ldb, voff_b);
    };
    if (live_i) cursor_item();
    auto cursor_rsrc = [&](i32x4_t& ra, i32x4_t& rb) {
        ra = make_rsrc_words(pa_i + ao_i, live_i ? (ae_i - ao_i) * 2 : 0);   // past the last item: zero records, nothing is fetched
        rb = make_rsrc_words(pb_i + bo_i, live_i ? (be_i - bo_i) * 2 : 0);
    };
    auto cursor_advance = [&]() {
        ao_i += step_a; bo_i += step_b;
        if (++u_i == nh) {
            u_i = 0;
            i_i += G;
            live_i = i_i < n_items;
            if (live_i) cursor_item();
        }
    };

    f32x4 acc[NJ][4];
    bf16x8 a0[4], b0[NJ], a1[4], b1[NJ];
    // Fragment addresses of the k-strided half image ([32 k][256], 512-byte k-rows, 16-byte chunk c of k-row r at c ^ f(r): gemm_common.h
    // load_frag): fragment i of an operand sits at (lane part) ^ (i << 5) — the 16 i rows of its origin move the chunk index by 2 i, in
    // bit fields the lane part's XOR does not carry into — and its second half 4 k-rows (2 KiB) further.  Formed per read from ONE
    // lane value per operand, made opaque per half-step: left to itself hipcc keeps all twelve addresses (and their hi halves) live
    // across the loop, and with the per-item cursor beside them the 256 registers no longer hold (fragments spilled INSIDE the loop).
    const int li = lane & 15, krow_l = 8 * (lane >> 4) + (li >> 2), c0_l = (li & 3) >> 1;
    const uint32_t la0 = (uint32_t)(krow_l * 512 + (li & 1) * 8 + ((((wm * 8) | c0_l) ^ mn_f(krow_l)) << 4));
    const uint32_t lb0 = (uint32_t)(krow_l * 512 + (li & 1) * 8 + ((((wn * 16) | c0_l) ^ mn_f(krow_l)) << 4)) + H_TILE;
    auto tr_frag = [](uint32_t base, int i) {
        const uint32_t a = base ^ (uint32_t)(i << 5);
        typedef __attribute__((address_space(3))) s16x4* lp;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)a);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(a + 2048));
        return join8(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi));
    };
    auto istep = [&](int gu, const bf16x8 (&af)[4], const bf16x8 (&bfr)[NJ], bf16x8 (&an)[4], bf16x8 (&bn)[NJ]) {
        const uint32_t slot_rd = lds_addr_of(smem) + (uint32_t)(((gu + 1) & 3) * H_STAGE);
        uint32_t la = la0 + slot_rd, lb = lb0 + slot_rd;
        asm volatile("" : "+v"(la), "+v"(lb));
        i32x4_t ra, rb;
        cursor_rsrc(ra, rb);
        const uint32_t lds_st = lds_addr_of(smem + (gu & 3) * H_STAGE) + wave * 1024;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            an[g] = tr_frag(la, g);
            bn[2 * g] = tr_frag(lb, 2 * g);
            bn[2 * g + 1] = tr_frag(lb, 2 * g + 1);
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

    // ---- a producer part: the accumulators as they stand -> the tile's slab image of this part, then one count on the tile -------
    auto publish = [&](const V8Item& it) {
        float* img = P.slabs + ((int64_t)it.tile * (P.splits - 1) + it.part) * (V8_SLAB / 4);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(img, V8_SLAB);
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int base = wave * 32768 + lane_e * 16;          // wave w: 32 quads x 1 KiB
#pragma unroll
        for (int ni = 0; ni < NJ; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[ni][mi]), rs, base + (ni * 4 + mi) * 1024, 0, 16);   // sc1: write-through
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave, then the barrier, then ONE lane's add (Guideline 16, R1)
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(P.counters + it.tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // ---- the owner part: wait for the S - 1 images of its tile, add them in part order --------------------------------------------
    auto gather = [&](const V8Item& it) {
        const int need = P.splits - 1;
        int have = 0, spins = 0;
        const int32_t* cnt = P.counters + it.tile;
        while (true) {
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(have) : "v"(cnt) : "memory");
            if (have >= need) break;
            __builtin_amdgcn_s_sleep(8);
            if (++spins > P.spin_limit) {
                if (lane == 0) __hip_atomic_fetch_or(P.status, OBTE_STATUS_GEMM_SPLIT_HANDOFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        __syncthreads();   // (every wave polled for itself; the barrier keeps the waves of the workgroup together for the ring's next barrier)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int base = wave * 32768 + lane_e * 16;
        for (int part = 0; part < need; ++part) {
            const float* img = P.slabs + ((int64_t)it.tile * need + part) * (V8_SLAB / 4);
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(img, V8_SLAB);
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {       // eight 16-byte loads in flight per lane, four times
                f32x4 t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, base + (qt * 8 + q) * 1024, 0, 16));   // sc1
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[(qt * 8 + q) >> 2][(qt * 8 + q) & 3] += t[q];
            }
        }
    };
    // ---- epilogue of a finished tile: this wave's 64 x 128 through its own 4 KiB of staging, whole 256-byte row pieces -----------
    auto epilogue = [&](const V8Item& it) {
        const V8Problem& q = P.pr[it.prob];
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int em = lane_e & 15, g4 = lane_e >> 4;
        char* const stg = smem + V3_RING + wave * V8_STG_WAVE;
        const uint32_t wr_l = (uint32_t)(em * 256 + (g4 & 1) * 8);
        const int64_t tile_o = (it.m0 + wm * 64) * q.ldd + it.n0 + wn * 128;
        const uint32_t ldd32 = (uint32_t)q.ldd;
        const uint32_t off_l = (uint32_t)g4 * ldd32 + (uint32_t)em * 8;
        auto o_of = [&](int mi, int r) { return tile_o + (int64_t)(off_l + (uint32_t)(mi * 16 + r * 4) * ldd32); };
        const bool add = q.aux != nullptr;
        bf16x8 raux[2][4];      // the gradient accumulated into, one round ahead of its use
        auto load_aux = [&](int mi) {
            if (add) {
#pragma unroll
                for (int r = 0; r < 4; ++r) raux[mi & 1][r] = *reinterpret_cast<const bf16x8*>(q.aux + o_of(mi, r));
            }
        };
        auto stage_round = [&](int mi, bf16x8 (&st)[4]) {
            uint32_t wl = wr_l, sw = (uint32_t)((g4 >> 1) ^ em) << 4;
            asm volatile("" : "+v"(wl), "+v"(sw));
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni)
                *reinterpret_cast<bf16x4*>(stg + wl + (sw ^ (uint32_t)(ni << 5))) = __builtin_convertvector(acc[ni][mi] * q.alpha, bf16x4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t row = (uint32_t)(4 * r + g4);
                st[r] = *reinterpret_cast<const bf16x8*>(stg + row * 256 + ((((uint32_t)em) ^ row) << 4));
            }
        };
        auto finish_round = [&](int mi, const bf16x8 (&st)[4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bf16x8 v = st[r];
                if (add) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(raux[mi & 1][r][j]) + bf2f(v[j]));
                }
                *reinterpret_cast<bf16x8*>(q.d + o_of(mi, r)) = v;
            }
        };
        bf16x8 sa[4], sb[4];
        load_aux(0);
        stage_round(0, sa);
        load_aux(1);
        stage_round(1, sb);
        finish_round(0, sa);
        load_aux(2);
        stage_round(2, sa);
        finish_round(1, sb);
        load_aux(3);
        stage_round(3, sb);
        finish_round(2, sa);
        finish_round(3, sb);
    };

    if (!live_i) return;
    // ---- fill the ring once, then walk the items ------------------------------------------------------------------------------------
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
    for (int i = 0; i < 4; ++i) a0[i] = tr_frag(la0 + lds_addr_of(smem), i);
#pragma unroll
    for (int i = 0; i < NJ; ++i) b0[i] = tr_frag(lb0 + lds_addr_of(smem), i);

    int gu = 0;
    int lax = 0;   // half-steps in which the previous item's 16 stores per wave may still be in flight beside the ring's two half-stages
    for (int i_c = wmy; i_c < n_items; i_c += G) {
        const V8Item it = item_of(P, i_c);
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int u = 0; u < nh; u += 2) {
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
        // The item's closing act issues vector-memory operations the ring's counted waits do not know about (a producer's 32
        // image stores and its drain, an owner's polls and image loads — both wait for vmcnt(0), which also lands the three
        // half-stages in flight: the next item starts from a full ring either way); only the epilogue's 16 stores stay in flight.
        if (P.splits > 1 && it.part < P.splits - 1) {
            publish(it);
            lax = 0;
        } else {
            if (P.splits > 1) gather(it);
            epilogue(it);
            lax = 3;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------------
// Which groups take this path: every problem a weight gradient (both operands k-strided), whole 256 x 256 tiles, one K for all,
// overwrite or accumulate; and a split that makes the items divide into full rounds of the 256 CUs (or no split where the tiles
// already do: n_embd = 2048 gives 768 tiles).
static int v8_plan(const obte_gemm_args* gs, int count, int* splits_out, int* tiles_out) {
    if (count < 1 || count > V8_MAXP) return 0;
    int64_t tiles = 0;
    for (int i = 0; i < count; ++i) {
        const obte_gemm_args* g = gs + i;
        if (g->a_kmajor || g->b_kmajor || g->M % 256 || g->N % 256 || g->K != gs[0].K || g->K % 64) return 0;
        if (g->epilogue != OBTE_EPI_NONE && g->epilogue != OBTE_EPI_ADD) return 0;
        if (g->ldd >= (1ll << 24)) return 0;
        tiles += (g->M / 256) * (g->N / 256);
    }
    if (tiles < 64 || tiles >= (1 << 20)) return 0;
    const int64_t nh = gs[0].K / 32;
    int best = 0;
    for (int s = 1; s <= 8; ++s) {      // the smallest split whose items fill whole rounds to >= 96 % and leave >= 8 half-steps a part
        if (nh % (2 * s) || nh / s < 8) continue;
        const int64_t items = tiles * s, rounds = (items + 255) / 256;
        if (items * 100 >= rounds * 256 * 96) { best = s; break; }
    }
    if (!best) return 0;
    *splits_out = best; *tiles_out = (int)tiles;
    return 1;
}

extern "C" int64_t obte_gemm_grouped_workspace_bytes(const obte_gemm_args* gs, int count) {
    int splits = 0, tiles = 0;
    if (!gs || !v8_plan(gs, count, &splits, &tiles)) return 0;
    return (int64_t)tiles * (splits - 1) * V8_SLAB + (((int64_t)tiles * 4 + 255) & ~255ll) + 256;
}

// returns 1 if the group was launched on this structure, 0 if it does not apply (the caller takes the full-K grouped launch), < 0 on error
int obte_gemm_v8_try(const obte_gemm_args* gs, int count, void* ws, int64_t ws_bytes, hipStream_t st) {
    int splits = 0, tiles = 0;
    if (!ws || !v8_plan(gs, count, &splits, &tiles)) return 0;
    const int64_t need = obte_gemm_grouped_workspace_bytes(gs, count);
    if (ws_bytes < need) return 0;
    V8Params P = {};
    int first = 0;
    for (int i = 0; i < count; ++i) {
        const obte_gemm_args* g = gs + i;
        V8Problem& q = P.pr[i];
        q.a = (const bf16*)g->a; q.b = (const bf16*)g->b; q.d = (bf16*)g->d;
        q.aux = g->epilogue == OBTE_EPI_ADD ? (const bf16*)g->aux : nullptr;
        if (g->epilogue == OBTE_EPI_ADD && !g->aux) { obte_set_error("obte_gemm_grouped_bf16: EPI_ADD needs aux"); return OBTE_EINVAL; }
        q.lda = g->lda; q.ldb = g->ldb; q.ldd = g->ldd;
        q.a_elems = g->K * g->lda; q.b_elems = g->K * g->ldb;
        q.tiles_m = (int)(g->M / 256); q.tiles_n = (int)(g->N / 256); q.first_tile = first;
        q.alpha = g->alpha;
        first += q.tiles_m * q.tiles_n;
    }
    for (int i = count; i < V8_MAXP; ++i) P.pr[i] = P.pr[0];
    P.count = count; P.total_tiles = tiles; P.splits = splits;
    P.nh_part = (int)(gs[0].K / 32 / splits);
    char* w = (char*)ws;
    P.slabs = (float*)w;
    P.counters = (int32_t*)(w + (int64_t)tiles * (splits - 1) * V8_SLAB);
    P.status = obte_status_word();
    if (!P.status) { obte_set_error("obte_gemm_grouped_bf16: the device status word could not be allocated"); return OBTE_ELAUNCH; }
    P.spin_limit = 1 << 20;
    if (splits > 1 && hipMemsetAsync(P.counters, 0, (size_t)tiles * 4, st) != hipSuccess) {
        obte_set_error("obte_gemm_grouped_bf16: memset of the split counters failed");
        return OBTE_ELAUNCH;
    }
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)gemm_v8_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, V8_SMEM);
        return true;
    }();
    (void)attr_set;
    const int items = tiles * splits;
    hipLaunchKernelGGL(gemm_v8_wgrad_kernel, dim3(items < 256 ? items : 256), dim3(NTHREADS), V8_SMEM, st, P);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { obte_set_error("obte_gemm_grouped_bf16(split): launch failed: %s", hipGetErrorString(e_)); return OBTE_ELAUNCH; }
    return 1;
}
