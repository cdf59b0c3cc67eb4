// k_attn4b.hip — attention BACKWARD for gfx950, second generation (the dQ kernel): the structure of k_attn4.hip — 4 waves per
// workgroup, one per SIMD, 64 queries (two 32-row blocks A / B) per wave, every wave its own software pipeline, the elementwise
// work placed by hand between the MFMAs — applied to
//   S^T = K Q^T,  dP^T = V dO^T   ([key][q], query on lanes)
//   dS^T = P^T * (dP^T - delta[q]) * scale,   P^T = exp2(S^T sc - lse2[q])
//   dQ^T[d][q] += K^T[d][key] . dS^T[key][q]                                            (TF:qwen3:185-207 backward, TF:esm:292-317)
// The pipeline's unit is (32-key step j, query block x), u = 2 j + x:
//   phase(u):   MFMAs  dQ(u - 1)  [8: K^T fragments x the packed dS of the previous unit]
//                      S, dP(u + 1) [16: K / V row fragments x this block's Q / dO fragments, which live in AGPRs]
//               VALU   the 16 + 16 scores of unit u -> its packed dS                         (4.5 instructions per element)
// i.e. 24 MFMAs (~790 cycles) beside ~100 issue slots: the matrix pipe is the bound, where the round 1-5 kernel (8 waves in lockstep
// per tile, every wave re-reading the tile) ran at 0.15 of it.  Only one block's S / dP is live beside the other's (64 registers),
// which is what lets Q, dO (128 AGPRs) and the dQ accumulators (128 AGPRs) stay resident.
// Staging as in the forward: K, V (row tiles) and K^T (transposed image) by LDS-DMA through buffer descriptors, one tile ahead of
// the barrier that publishes them, one barrier per 64-key tile; K / V in two slots, K^T in three (its last reader is half a tile
// behind the row tiles').  Key bits 2 / 3 swapped on the row side (k_attn4.hip): a dS fragment is 8 consecutive keys.
#include "bra_device.h"
#include "bra_api_internal.h"
#include "bra_attn.h"
#include "bra_attn4.h"

namespace bra {

template <int HD>
struct TB4 {
    static constexpr int CH = HD / 8, DS = HD / 16, DB = HD / 32;
    static constexpr int RSH = HD == 128 ? 0 : (HD == 64 ? 1 : 2);
    static constexpr int KBYTES = 64 * HD * 2, TBYTES = HD * 128;
    static constexpr int RSLOT = 2 * KBYTES;          // row tiles of one key tile: [K | V]
    static constexpr int OFF_T = 2 * RSLOT;           // K^T ring behind the two row slots
    static constexpr int SMEM = 2 * RSLOT + 3 * TBYTES;
    static constexpr int KPW = KBYTES / 4096, TPW = TBYTES / 4096;      // DMA pieces per wave: K, V (KPW each), K^T (TPW)
    static constexpr int NDMA = 2 * KPW + TPW;
    static constexpr int NDQ = 2 * DB;                // MFMAs of dQ(u - 1)
    static constexpr int NSD = 2 * DS;                // MFMAs of S, dP(u + 1)
    static constexpr int NG = NDQ + NSD;
    static constexpr int NITEM = 9 * (8 + 2);         // item slots of one unit's elementwise work (el_item)
    static constexpr int QD = 1;
    static constexpr int item0(int G, bool dma) {
        if (!dma) return (G * NITEM) / NG;
        return G <= NDMA ? G * QD : NDMA * QD + ((G - NDMA) * (NITEM - NDMA * QD)) / (NG - NDMA);
    }
};

// One unit's elementwise work as single-instruction items, software-pipelined through slots of nine (k_attn4.hip sm_item): slot n
// holds the four arguments of pair n (S and dP side), the two exponentials of pair n - 1, the two products and the pack of n - 2.
struct ElState {
    float x[2][2], t[2][2];     // by pair parity: exp2 arguments, (dP * scale - delta * scale)
    float e[2][2], tt[2][2];
};
template <int K>
__device__ __forceinline__ void el_item(const f32x16& s, const f32x16& dp, u32x4 (&ds)[2], ElState& st, float lse2, float dlt_s, float sc, float scale) {
    constexpr int NP = 8, n = K / 9, u = K % 9;
#ifdef BRA_A4_NOSM        // (timing probe: no elementwise work — garbage results)
    return;
#endif
    if constexpr (u < 4) {
        if constexpr (n < NP) {
            if constexpr (u < 2) { st.x[n & 1][u] = fmaf(s[2 * n + u], sc, -lse2); pin_f32(st.x[n & 1][u]); }
            else { st.t[n & 1][u - 2] = fmaf(dp[2 * n + u - 2], scale, -dlt_s); pin_f32(st.t[n & 1][u - 2]); }
        }
    } else if constexpr (u < 6) {
        if constexpr (n >= 1 && n <= NP) {
            st.e[(n - 1) & 1][u - 4] = fast_exp2(st.x[(n - 1) & 1][u - 4]); pin_f32(st.e[(n - 1) & 1][u - 4]);
            st.tt[(n - 1) & 1][u - 4] = st.t[(n - 1) & 1][u - 4];
        }
    } else if constexpr (n >= 2 && n <= NP + 1) {
        constexpr int pr = n - 2;
        if constexpr (u < 8) { st.e[pr & 1][u - 6] *= st.tt[pr & 1][u - 6]; pin_f32(st.e[pr & 1][u - 6]); }
        else {
            uint32_t w2 = pack_bf2(st.e[pr & 1][0], st.e[pr & 1][1]);
            pin_u32(w2);
            constexpr int g = pr >> 2, c4 = pr & 3;
            if constexpr (c4 == 0) ds[g].x = w2; else if constexpr (c4 == 1) ds[g].y = w2;
            else if constexpr (c4 == 2) ds[g].z = w2; else ds[g].w = w2;
        }
    }
}
template <int LO, int HI>
__device__ __forceinline__ void el_items(const f32x16& s, const f32x16& dp, u32x4 (&ds)[2], ElState& st, float lse2, float dlt_s, float sc, float scale) {
    if constexpr (LO < HI) {
        el_item<LO>(s, dp, ds, st, lse2, dlt_s, sc, scale);
        el_items<LO + 1, HI>(s, dp, ds, st, lse2, dlt_s, sc, scale);
    }
}

template <int HD>
struct CtxB4 {
    unsigned kfo[HD / 16];        // LDS byte offset of this lane's K (and, + KBYTES, V) row fragment of d-step ds in a 32-key half
    unsigned tfo[2][2];           // LDS byte offset of this lane's K^T fragment of (key half kb, k-slot group s2), d block 0, in a K^T tile
    float sc, scale;
};

// phase(u) of the hot loop.  s_cur / dp_cur: scores of unit u (complete); s_nxt / dp_nxt: receive unit u + 1; ds_prev: packed dS of
// unit u - 1 (accumulated into dq_prev = the dQ accumulators of ITS query block); ds_cur: receives unit u's.
//   ktp: K^T tile of unit u - 1's step (kbt = its key half); rows: [K | V] row tiles of unit u + 1's step (kbr = its key half);
//   qf / dof: Q / dO fragments of unit u + 1's block.
// Fragment reads in blocks of four behind a wait, first used four MFMAs later (k_attn4.hip step4).
template <int HD, bool DMA, typename DmaFn>
__device__ __forceinline__ void phase4(const CtxB4<HD>& cx, const f32x16& s_cur, const f32x16& dp_cur, f32x16& s_nxt, f32x16& dp_nxt,
                                       const u32x4 (&ds_prev)[2], u32x4 (&ds_cur)[2], f32x16 (&dq_prev)[HD / 32],
                                       const u32x4 (&qf)[HD / 16], const u32x4 (&dof)[HD / 16], float lse2, float dlt_s,
                                       const char* ktp, int kbt, const char* rows, int kbr, DmaFn&& dma) {
    using T = TB4<HD>;
    ElState st;
    u32x4 fr[2][4];                                   // [block parity][fragment of the block]
    // fragment f of the phase: f < NDQ: K^T fragment (s2 = f / DB, db = f % DB); else pair index p = (f - NDQ): K row (p even) / V row
    // (p odd) fragment of d-step p / 2
    auto read_frag = [&](int f) -> u32x4 {
#ifdef BRA_A4_NOLDS       // (timing probe: no fragment reads — garbage results)
        { u32x4 z = {cx.kfo[0], cx.kfo[1], cx.tfo[0][0], 0x3c003c00u}; return z; }
#endif
        if (f < T::NDQ) return ld16(ktp + cx.tfo[kbt][f / T::DB] + (f % T::DB) * 4096);
        const int p = f - T::NDQ;
        return ld16(rows + cx.kfo[p >> 1] + (p & 1) * T::KBYTES + kbr * (32 * HD * 2));
    };
#pragma unroll
    for (int f = 3; f >= 0; --f) fr[0][f] = read_frag(f);
#define BRA_P_GROUP(G)                                                                                                     \
    if constexpr ((G) < T::NG) {                                                                                           \
        constexpr int blk_ = (G) / 4;                                                                                      \
        if constexpr ((G) < T::NDQ) {                                                                                      \
            constexpr int s2_ = (G) / T::DB, db_ = (G) % T::DB;                                                            \
            mfma_o(dq_prev[db_], fr[blk_ & 1][(G) % 4], ds_prev[s2_]);                                                     \
        } else {                                                                                                           \
            constexpr int p_ = (G) - T::NDQ, ds_ = p_ >> 1;                                                                \
            if constexpr ((p_ & 1) == 0) {                                                                                 \
                if constexpr (ds_ == 0) { f32x16 z_ = {}; s_nxt = mfma_32x32x16(fr[blk_ & 1][(G) % 4], qf[0], z_); }        \
                else s_nxt = mfma_32x32x16(fr[blk_ & 1][(G) % 4], qf[ds_], s_nxt);                                         \
            } else {                                                                                                       \
                if constexpr (ds_ == 0) { f32x16 z_ = {}; dp_nxt = mfma_32x32x16(fr[blk_ & 1][(G) % 4], dof[0], z_); }      \
                else dp_nxt = mfma_32x32x16(fr[blk_ & 1][(G) % 4], dof[ds_], dp_nxt);                                      \
            }                                                                                                              \
        }                                                                                                                  \
        if constexpr ((G) % 4 == 0 && (G) + 4 < T::NG) {                                                                   \
            sched_fence();                                                                                                 \
            /* (youngest first: LDS reads return in order, so the next block's FIRST MFMA waits for the read issued last and the */ \
            /*  other three need no wait of their own) */                                                                  \
            _Pragma("unroll") for (int u_ = 3; u_ >= 0; --u_) fr[(blk_ + 1) & 1][u_] = read_frag((G) + 4 + u_);            \
        }                                                                                                                  \
        if constexpr (DMA && (G) < T::NDMA && !kNoDma) dma(G);                                                             \
        el_items<T::item0(G, DMA), T::item0((G) + 1, DMA)>(s_cur, dp_cur, ds_cur, st, lse2, dlt_s, cx.sc, cx.scale);        \
        sched_fence();                                                                                                     \
    }
    BRA_P_GROUP(0) BRA_P_GROUP(1) BRA_P_GROUP(2) BRA_P_GROUP(3) BRA_P_GROUP(4) BRA_P_GROUP(5) BRA_P_GROUP(6) BRA_P_GROUP(7)
    BRA_P_GROUP(8) BRA_P_GROUP(9) BRA_P_GROUP(10) BRA_P_GROUP(11) BRA_P_GROUP(12) BRA_P_GROUP(13) BRA_P_GROUP(14) BRA_P_GROUP(15)
    BRA_P_GROUP(16) BRA_P_GROUP(17) BRA_P_GROUP(18) BRA_P_GROUP(19) BRA_P_GROUP(20) BRA_P_GROUP(21) BRA_P_GROUP(22) BRA_P_GROUP(23)
#undef BRA_P_GROUP
}

// the same phase without the interleave, with run-time switches (block prologue / tail)
template <int HD>
__device__ __forceinline__ void cold_phase4(const CtxB4<HD>& cx, const f32x16& s_cur, const f32x16& dp_cur, f32x16& s_nxt, f32x16& dp_nxt,
                                            const u32x4 (&ds_prev)[2], u32x4 (&ds_cur)[2], f32x16 (&dq_prev)[HD / 32],
                                            const u32x4 (&qf)[HD / 16], const u32x4 (&dof)[HD / 16], float lse2, float dlt_s,
                                            const char* ktp, int kbt, const char* rows, int kbr, bool do_dq, bool do_el, bool do_sdp) {
    using T = TB4<HD>;
    if (do_dq) {
#pragma unroll
        for (int f = 0; f < T::NDQ; ++f) mfma_o(dq_prev[f % T::DB], ld16(ktp + cx.tfo[kbt][f / T::DB] + (f % T::DB) * 4096), ds_prev[f / T::DB]);
        mfma_drain();
    }
    if (do_el) {
        float e[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = fast_exp2(fmaf(s_cur[r], cx.sc, -lse2)) * fmaf(dp_cur[r], cx.scale, -dlt_s);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            ds_cur[g].x = pack_bf2(e[8 * g + 0], e[8 * g + 1]); ds_cur[g].y = pack_bf2(e[8 * g + 2], e[8 * g + 3]);
            ds_cur[g].z = pack_bf2(e[8 * g + 4], e[8 * g + 5]); ds_cur[g].w = pack_bf2(e[8 * g + 6], e[8 * g + 7]);
        }
    }
    if (do_sdp) {
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) {
            const u32x4 kf = ld16(rows + cx.kfo[ds] + kbr * (32 * HD * 2));
            const u32x4 vf = ld16(rows + cx.kfo[ds] + T::KBYTES + kbr * (32 * HD * 2));
            if (ds == 0) { f32x16 z = {}; s_nxt = mfma_32x32x16(kf, qf[0], z); dp_nxt = mfma_32x32x16(vf, dof[0], z); }
            else { s_nxt = mfma_32x32x16(kf, qf[ds], s_nxt); dp_nxt = mfma_32x32x16(vf, dof[ds], dp_nxt); }
        }
    }
}

// scores of masked keys -> kMasked (one block; qb_off = 32 for block B)
__device__ __forceinline__ void mask_scores4b(f32x16& s, uint32_t valid32, bool causal, int lim0, int h) {
    const uint32_t vb = valid32 >> (8 * h);
    const int lim = lim0 - 8 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int kk = 16 * (r >> 3) + (r & 7);
        bool ok = (vb >> kk) & 1u;
        if (causal) ok = ok && kk <= lim;
        s[r] = ok ? s[r] : kMasked;
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_dq4_kernel(AttnArgs a) {
    using T = TB4<HD>;
    BRA_DYN_SMEM(smem);                               // [2][K tile | V tile] [3][K^T tile]
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6), h = lane >> 5, l31 = lane & 31;
    int bx_, hq, b;
    attn_block_coords(0, a.causal, bx_, hq, b);
    const int hkv = hq / (a.Hq / a.Hkv);
    // key range in `nsp` parts (grids that cannot fill the chip): part sp visits tiles [tb, tb + ntile) and leaves its dQ in fp32 for
    // attn_sum_parts_kernel; tile / step / unit indices below are relative to tb
    const int nsp = a.nsplit > 1 ? a.nsplit : 1;
    const int sp = nsp > 1 ? bx_ % nsp : 0;
    if (nsp > 1) bx_ /= nsp;
    const int q0 = bx_ * 256, qw0 = q0 + wave * 64;
    const char* kb_ = uniform_ptr(a.k + b * a.k_sb + hkv * a.k_sh);
    const char* vb_ = uniform_ptr(a.v + b * a.v_sb + hkv * a.v_sh);
    const char* ktb = uniform_ptr(a.kt + b * a.kt_sb + hkv * a.kt_sh);

    CtxB4<HD> cx;
    cx.sc = a.scale * kLog2e;
    cx.scale = a.scale;
    {
        const int row = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);      // key bits 2 / 3 swapped (k_attn4.hip)
        const int sw = (row >> T::RSH) & (T::CH - 1);
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) cx.kfo[ds] = (unsigned)(row * (HD * 2) + (((2 * ds + h) ^ sw) << 4));
        const int swv = (l31 >> 1) & 7;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) cx.tfo[kb][s2] = (unsigned)(l31 * 128 + (((4 * kb + 2 * s2 + h) ^ swv) << 4));
    }
    int kv_end = a.Sk;
    if (a.causal) { const int last = q0 + 255 + a.q_off + 1; kv_end = last < kv_end ? last : kv_end; }
    const int ntile_all = kv_end > 0 ? (kv_end + 63) / 64 : 0;
    const int tb = nsp > 1 ? (ntile_all * sp) / nsp : 0;
    const int ntile = (nsp > 1 ? (ntile_all * (sp + 1)) / nsp : ntile_all) - tb;
    // DMA sources (byte offsets inside the (batch, kv-head) slices; rows beyond Sk are outside the descriptors: zeros)
    const int k_sbytes = (int)a.k_ss * 2, v_sbytes = (int)a.v_ss * 2;
    const BufDesc kdesc = make_bufdesc(kb_, (unsigned)((a.Sk - 1) * k_sbytes + HD * 2));
    const BufDesc vdesc = make_bufdesc(vb_, (unsigned)((a.Sk - 1) * v_sbytes + HD * 2));
    const BufDesc tdesc = make_bufdesc(ktb, (unsigned)(HD * (int)a.kt_sd * 2));
    unsigned ksrc[T::KPW], vsrc[T::KPW], tsrc[T::TPW];
#pragma unroll
    for (int i = 0; i < T::KPW; ++i) {
        const int u = 64 * (wave * T::KPW + i) + lane, row = u / T::CH, c = (u % T::CH) ^ ((row >> T::RSH) & (T::CH - 1));
        ksrc[i] = attn_mul24(row, k_sbytes) + (unsigned)(c * 16);
        vsrc[i] = attn_mul24(row, v_sbytes) + (unsigned)(c * 16);
    }
#pragma unroll
    for (int i = 0; i < T::TPW; ++i) {
        const int u = 64 * (wave * T::TPW + i) + lane, d = u >> 3, c = (u & 7) ^ ((d >> 1) & 7);
        tsrc[i] = attn_mul24(d, (int)a.kt_sd * 2) + (unsigned)(c * 16);
    }
    auto dma_piece = [&](int i, int tile, int rslot, int tslot) {
        if (i < T::KPW) dma16(kdesc, ksrc[i], (unsigned)((tb + tile) * 64) * (unsigned)k_sbytes, smem + rslot * T::RSLOT + (wave * T::KPW + i) * 1024);
        else if (i < 2 * T::KPW) dma16(vdesc, vsrc[i - T::KPW], (unsigned)((tb + tile) * 64) * (unsigned)v_sbytes,
                                       smem + rslot * T::RSLOT + T::KBYTES + (wave * T::KPW + i - T::KPW) * 1024);
        else dma16(tdesc, tsrc[i - 2 * T::KPW], (unsigned)((tb + tile) * 128), smem + T::OFF_T + tslot * T::TBYTES + (wave * T::TPW + i - 2 * T::KPW) * 1024);
    };

    int nunit_w = 0;                                  // this wave's units: two per 32-key step that holds a key one of its queries sees
    if (qw0 < a.Sq && ntile > 0) {
        int lastq = qw0 + 63; lastq = lastq < a.Sq ? lastq : a.Sq - 1;
        int lastk = a.causal ? lastq + a.q_off : a.Sk - 1;
        lastk = lastk < a.Sk ? lastk : a.Sk - 1;
        int ns = lastk >= 0 ? lastk / 32 + 1 - 2 * tb : 0;
        ns = ns > 0 ? ns : 0;
        ns = ns < 2 * ntile ? ns : 2 * ntile;
        nunit_w = 2 * ns;
    }

    // Q / dO fragments and the row statistics of this lane's two queries
    u32x4 qf[2][T::DS], dof[2][T::DS];
    float lse2[2], dlt_s[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = qw0 + 32 * qb + l31;
        qr = qr < a.Sq ? qr : a.Sq - 1;
        const bf16_t* qp = a.q + b * a.q_sb + (long)qr * a.q_ss + hq * a.q_sh;
        const bf16_t* dp = a.dout + b * a.do_sb + (long)qr * a.do_ss + hq * a.do_sh;
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) { qf[qb][ds] = ld16(qp + ds * 16 + 8 * h); dof[qb][ds] = ld16(dp + ds * 16 + 8 * h); }
        const long lidx = ((long)b * a.Hq + hq) * a.Sq + qr;
        lse2[qb] = a.lse[lidx] * kLog2e;
        dlt_s[qb] = a.delta[lidx] * a.scale;
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) { to_agpr(qf[qb][ds]); to_agpr(dof[qb][ds]); }
    f32x16 dq[2][T::DB];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < T::DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[qb][i][r] = 0.f;

    auto mask_byte = [&](int tile) -> int {
        int kj = (tb + tile) * 64 + lane;
        const bool in = kj < a.Sk;
        kj = in ? kj : a.Sk - 1;
        int v = a.kmask ? (int)a.kmask[(long)b * a.Sk + kj] : 1;
        return in ? v : 0;
    };

    f32x16 s[2], dp[2];                               // scores / dP of the unit in flight of block A / B
    u32x4 dsp[2][2];                                  // packed dS of the last unit of block A / B
    if (ntile > 0) {
#pragma unroll
        for (int i = 0; i < T::NDMA; ++i) dma_piece(i, 0, 0, 0);
        int mb0 = mask_byte(0), mb1 = mask_byte(1);
        wait_vmcnt<0>();
        raw_barrier();
        {
            const int t1 = ntile > 1 ? 1 : 0;
#pragma unroll
            for (int i = 0; i < T::NDMA; ++i) dma_piece(i, t1, 1, 1);
        }
        uint64_t vcur = wave_ballot(mb0 != 0), vnext = wave_ballot(mb1 != 0);      // validity words of tiles (u / 4) and the next
        int mb_pend = 0;
        auto noop = [](int) {};
        // masks of unit u (step j = u / 2, block x = u & 1), applied to its finished scores; vword = the validity word of tile j / 2
        auto prep = [&](f32x16& sx, int u, uint64_t vword) {
            const int j = u >> 1, x = u & 1, kv0s = 64 * tb + 32 * j;
            const uint32_t v32 = (uint32_t)(vword >> (32 * (j & 1)));
            const bool full = v32 == 0xffffffffu && (!a.causal || kv0s + 31 <= qw0 + 32 * x + a.q_off);
            if (!full) mask_scores4b(sx, v32, a.causal != 0, qw0 + 32 * x + l31 + a.q_off - kv0s, opaque_i(lane) >> 5);
        };
        const char* const tring = smem + T::OFF_T;
        // generic phase(u): locations from u (cold form)
        auto cold = [&](int u, uint64_t vw) {
            const int x = u & 1;
            const bool do_dq = u >= 1 && u <= nunit_w, do_el = u < nunit_w, do_sdp = u + 1 < nunit_w;
            const int up = u - 1, jp = up >> 1, un = u + 1, jn = un >> 1;
            const char* ktp = tring + (((jp >> 1) % 3 + 3) % 3) * T::TBYTES;
            const char* rows = smem + ((jn >> 1) & 1) * T::RSLOT;
            if (do_el) { if (x == 0) prep(s[0], u, vw); else prep(s[1], u, vw); }
            if (x == 0) cold_phase4<HD>(cx, s[0], dp[0], s[1], dp[1], dsp[1], dsp[0], dq[1], qf[1], dof[1], lse2[0], dlt_s[0], ktp, jp & 1, rows, jn & 1,
                                        do_dq, do_el, do_sdp);
            else cold_phase4<HD>(cx, s[1], dp[1], s[0], dp[0], dsp[0], dsp[1], dq[0], qf[0], dof[0], lse2[1], dlt_s[1], ktp, jp & 1, rows, jn & 1,
                                 do_dq, do_el, do_sdp);
        };
        if (nunit_w > 0) {
            // S, dP of unit 0 (a phase that does nothing else), then phases 0 .. 2
            cold_phase4<HD>(cx, s[1], dp[1], s[0], dp[0], dsp[0], dsp[1], dq[0], qf[0], dof[0], lse2[1], dlt_s[1], tring, 0, smem, 0, false, false, true);
            cold(0, vcur); cold(1, vcur); cold(2, vcur);
        }
        // iteration t: phases 4 t + 3 .. 4 t + 6.  Reads K^T(t) half 1, K^T(t + 1) half 0, the row tiles of t + 1; issues tile t + 2.
        // Hot loop: all four phases are full phases of this wave (4 t + 7 < nunit_w).
        const int tmain = nunit_w >= 8 ? (nunit_w - 4) / 4 : 0;
        int t = 0;
        // (tile parities compile-time: row slot = tile & 1; the K^T ring has three slots, so the hot loop is unrolled over six tiles'
        //  worth of slot indices by passing the ring slot as a run-time byte offset instead: only the K^T fragments pay a v_add)
#define BRA_HOT_ITER(PAR)                                                                                                    \
        {                                                                                                                  \
            wait_vmcnt<0>();                                                                                               \
            raw_barrier();                                                                                                 \
            mb_pend = mask_byte(t + 2);                                                                                    \
            int tn = t + 2;                                                                                                \
            tn = tn < ntile ? tn : ntile - 1;                                                                              \
            const int ts_new = (t + 2) % 3;                                                                                \
            auto dma = [&](int i) { dma_piece(i, tn, PAR, ts_new); };                                                      \
            const char* kt_t = tring + (t % 3) * T::TBYTES;                /* K^T(t) */                                   \
            const char* kt_n = tring + ((t + 1) % 3) * T::TBYTES;          /* K^T(t + 1) */                               \
            const char* rows_n = smem + (1 - (PAR)) * T::RSLOT;            /* K, V of tile t + 1 */                        \
            /* phase 4 t + 3 (x = 1, step 2 t + 1): dQ of (2 t + 1, A); S, dP of (2 t + 2, A) */                            \
            prep(s[1], 4 * t + 3, vcur);                                                                                   \
            phase4<HD, true>(cx, s[1], dp[1], s[0], dp[0], dsp[0], dsp[1], dq[0], qf[0], dof[0], lse2[1], dlt_s[1], kt_t, 1, rows_n, 0, dma);   \
            /* phase 4 t + 4 (x = 0, step 2 t + 2): dQ of (2 t + 1, B); S, dP of (2 t + 2, B) */                            \
            prep(s[0], 4 * t + 4, vnext);                                                                                  \
            phase4<HD, false>(cx, s[0], dp[0], s[1], dp[1], dsp[1], dsp[0], dq[1], qf[1], dof[1], lse2[0], dlt_s[0], kt_t, 1, rows_n, 0, noop); \
            /* phase 4 t + 5 (x = 1, step 2 t + 2): dQ of (2 t + 2, A); S, dP of (2 t + 3, A) */                            \
            prep(s[1], 4 * t + 5, vnext);                                                                                  \
            phase4<HD, false>(cx, s[1], dp[1], s[0], dp[0], dsp[0], dsp[1], dq[0], qf[0], dof[0], lse2[1], dlt_s[1], kt_n, 0, rows_n, 1, noop); \
            /* phase 4 t + 6 (x = 0, step 2 t + 3): dQ of (2 t + 2, B); S, dP of (2 t + 3, B) */                            \
            prep(s[0], 4 * t + 6, vnext);                                                                                  \
            phase4<HD, false>(cx, s[0], dp[0], s[1], dp[1], dsp[1], dsp[0], dq[1], qf[1], dof[1], lse2[0], dlt_s[0], kt_n, 0, rows_n, 1, noop); \
            vcur = vnext;                                                                                                  \
            vnext = wave_ballot(mb_pend != 0);                                                                             \
            ++t;                                                                                                           \
        }
        while (t + 1 < tmain) { BRA_HOT_ITER(0) BRA_HOT_ITER(1) }
        if (t < tmain) BRA_HOT_ITER(0)
#undef BRA_HOT_ITER
        for (; t < ntile; ++t) {
            wait_vmcnt<0>();
            raw_barrier();
            mb_pend = mask_byte(t + 2);
            int tn = t + 2;
            tn = tn < ntile ? tn : ntile - 1;
#pragma unroll
            for (int i = 0; i < T::NDMA; ++i) dma_piece(i, tn, t & 1, (t + 2) % 3);
            cold(4 * t + 3, vcur); cold(4 * t + 4, vnext); cold(4 * t + 5, vnext); cold(4 * t + 6, vnext);
            vcur = vnext;
            vnext = wave_ballot(mb_pend != 0);
        }
        wait_vmcnt<0>();
    }

    mfma_drain();
    if (nsp > 1) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qi = qw0 + 32 * qb + l31;
            float* op = a.part_o + ((((long)b * a.Hq + hq) * nsp + sp) * a.Sq + (qi < a.Sq ? qi : 0)) * HD;
#pragma unroll
            for (int db = 0; db < T::DB; ++db) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w = {dq[qb][db][4 * g + 0], dq[qb][db][4 * g + 1], dq[qb][db][4 * g + 2], dq[qb][db][4 * g + 3]};
                    if (qi < a.Sq) *reinterpret_cast<f32x4*>(op + db * 32 + 8 * g + 4 * h) = w;
                }
                sched_fence();
            }
        }
        return;
    }
    // ---- epilogue: dQ rows as bf16, 16-byte pieces (k_attn4.hip)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = qw0 + 32 * qb + l31;
        bf16_t* op = a.dq + b * a.dq_sb + (long)(qi < a.Sq ? qi : 0) * a.dq_ss + hq * a.dq_sh;
#pragma unroll
        for (int db = 0; db < T::DB; ++db) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                uint32_t a0 = pack_bf2(dq[qb][db][4 * g + 0], dq[qb][db][4 * g + 1]);
                uint32_t a1 = pack_bf2(dq[qb][db][4 * g + 2], dq[qb][db][4 * g + 3]);
                uint32_t b0 = pack_bf2(dq[qb][db][4 * g + 4], dq[qb][db][4 * g + 5]);
                uint32_t b1 = pack_bf2(dq[qb][db][4 * g + 6], dq[qb][db][4 * g + 7]);
                xhalf_pair(a0, b0);
                xhalf_pair(a1, b1);
                u32x4 w = {a0, a1, b0, b1};
                if (qi < a.Sq) st16(op + db * 32 + 8 * g + 8 * h, w);
            }
            sched_fence();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// dK / dV in the same structure, with the roles of keys and queries exchanged: a wave owns 64 KEYS (two 32-key blocks A / B, key
// on lanes; its K — and for dK its V — fragments live in AGPRs beside the 128 accumulator registers) and walks the (q-head of the
// group, 64-query tile) iterations of its workgroup; GQA's sum over the group needs no atomics.
//   S[q][key] = Q K^T,  dP[q][key] = dO V^T      (query on registers, key on lanes; Q / dO row fragments from LDS)
//   WHICH = 1:  dV^T[d][key] += dO^T[d][q] . P[q][key]           (16 MFMAs per unit: S, dV;   2.5 VALU per element)
//   WHICH = 2:  dK^T[d][key] += Q^T[d][q] . dS[q][key]           (24 MFMAs per unit: S, dP, dK; 4.5 VALU per element)
// Two launches that each recompute S, as in rounds 1-5: both accumulator sets beside both fragment sets exceed the register file.
// unit u = 4 it + 2 (query half of the tile) + (key block); phase(u): MFMAs {ACC(u - 1), S [, dP](u + 1)} beside the elementwise
// work of unit u.  The per-QUERY statistics (lse, delta) sit on the register index here: 16 (+ 16) values per lane and 32-query
// half, read from LDS at the start of the phase (two 8-value runs per lane: the row side's key-bit swap again).
template <int HD, int WHICH>
struct TK4 {
    static constexpr bool DK = WHICH == 2;
    static constexpr int CH = HD / 8, DS = HD / 16, DB = HD / 32;
    static constexpr int RSH = HD == 128 ? 0 : (HD == 64 ? 1 : 2);
    static constexpr int KBYTES = 64 * HD * 2, TBYTES = HD * 128;
    static constexpr int RSLOT = (DK ? 2 : 1) * KBYTES;            // row tiles of one query tile: Q [| dO]
    static constexpr int TSLOT = TBYTES + 512;                     // transposed tile (dO^T for dV, Q^T for dK) + lse2[64] + delta*scale[64]
    static constexpr int OFF_T = 2 * RSLOT;
    static constexpr int SMEM = 2 * RSLOT + 3 * TSLOT;
    static constexpr int KPW = KBYTES / 4096, TPW = TBYTES / 4096;
    static constexpr int NDMA = (DK ? 2 : 1) * KPW + TPW;
    static constexpr int NACC = 2 * DB;
    static constexpr int NSD = (DK ? 2 : 1) * DS;
    static constexpr int NG = NACC + NSD;
    static constexpr int IPS = DK ? 9 : 5;                         // items per slot of the elementwise pipeline
    static constexpr int NITEM = IPS * (8 + 2);
    static constexpr int item0(int G, bool dma) {
        if (!dma) return (G * NITEM) / NG;
        return G <= NDMA ? G : NDMA + ((G - NDMA) * (NITEM - NDMA)) / (NG - NDMA);
    }
};

// elementwise items of one unit (el_item's pipeline) with the statistics per register: lq / dq16 = lse2 / delta * scale of the 16
// queries this lane's registers hold
template <int WHICH, int K>
__device__ __forceinline__ void elkv_item(const f32x16& s, const f32x16& dp, u32x4 (&out)[2], ElState& st, const float (&lq)[16],
                                          const float (&dq16)[16], float sc, float scale) {
    constexpr int IPS = WHICH == 2 ? 9 : 5, NP = 8, n = K / IPS, u = K % IPS;
#ifdef BRA_A4_NOSM
    return;
#endif
    if constexpr (WHICH == 2) {
        if constexpr (u < 4) {
            if constexpr (n < NP) {
                if constexpr (u < 2) { st.x[n & 1][u] = fmaf(s[2 * n + u], sc, -lq[2 * n + u]); pin_f32(st.x[n & 1][u]); }
                else { st.t[n & 1][u - 2] = fmaf(dp[2 * n + u - 2], scale, -dq16[2 * n + u - 2]); pin_f32(st.t[n & 1][u - 2]); }
            }
        } else if constexpr (u < 6) {
            if constexpr (n >= 1 && n <= NP) {
                st.e[(n - 1) & 1][u - 4] = fast_exp2(st.x[(n - 1) & 1][u - 4]); pin_f32(st.e[(n - 1) & 1][u - 4]);
                st.tt[(n - 1) & 1][u - 4] = st.t[(n - 1) & 1][u - 4];
            }
        } else if constexpr (n >= 2 && n <= NP + 1) {
            constexpr int pr = n - 2;
            if constexpr (u < 8) { st.e[pr & 1][u - 6] *= st.tt[pr & 1][u - 6]; pin_f32(st.e[pr & 1][u - 6]); }
            else {
                uint32_t w2 = pack_bf2(st.e[pr & 1][0], st.e[pr & 1][1]);
                pin_u32(w2);
                constexpr int g = pr >> 2, c4 = pr & 3;
                if constexpr (c4 == 0) out[g].x = w2; else if constexpr (c4 == 1) out[g].y = w2;
                else if constexpr (c4 == 2) out[g].z = w2; else out[g].w = w2;
            }
        }
    } else {
        if constexpr (u < 2) {
            if constexpr (n < NP) { st.x[n & 1][u] = fmaf(s[2 * n + u], sc, -lq[2 * n + u]); pin_f32(st.x[n & 1][u]); }
        } else if constexpr (u < 4) {
            if constexpr (n >= 1 && n <= NP) { st.e[(n - 1) & 1][u - 2] = fast_exp2(st.x[(n - 1) & 1][u - 2]); pin_f32(st.e[(n - 1) & 1][u - 2]); }
        } else if constexpr (n >= 2 && n <= NP + 1) {
            constexpr int pr = n - 2;
            uint32_t w2 = pack_bf2(st.e[pr & 1][0], st.e[pr & 1][1]);
            pin_u32(w2);
            constexpr int g = pr >> 2, c4 = pr & 3;
            if constexpr (c4 == 0) out[g].x = w2; else if constexpr (c4 == 1) out[g].y = w2;
            else if constexpr (c4 == 2) out[g].z = w2; else out[g].w = w2;
        }
    }
}
template <int WHICH, int LO, int HI>
__device__ __forceinline__ void elkv_items(const f32x16& s, const f32x16& dp, u32x4 (&out)[2], ElState& st, const float (&lq)[16],
                                           const float (&dq16)[16], float sc, float scale) {
    if constexpr (LO < HI) {
        elkv_item<WHICH, LO>(s, dp, out, st, lq, dq16, sc, scale);
        elkv_items<WHICH, LO + 1, HI>(s, dp, out, st, lq, dq16, sc, scale);
    }
}

// the 16 per-query values of a lane: registers r = 0..7 <-> queries 8 h + r, r = 8..15 <-> 16 + 8 h + (r - 8) of the 32-query half
__device__ __forceinline__ void read_stats16(float (&v)[16], const float* base32, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(base32 + 16 * (g >> 1) + 8 * h + 4 * (g & 1));
        v[4 * g + 0] = q[0]; v[4 * g + 1] = q[1]; v[4 * g + 2] = q[2]; v[4 * g + 3] = q[3];
    }
}

// phase(u).  s_cur / dp_cur: unit u (complete); s_nxt / dp_nxt: unit u + 1; out_prev: packed P (dV) / dS (dK) of unit u - 1, accumulated
// into acc_prev (its key block's accumulators); out_cur: unit u's.
//   tp: transposed tile (+ statistics) of unit u - 1 (kbt = its query half); tc: the one of unit u (kbc); rows: row tiles of unit u + 1 (kbr);
//   kf / vf: K / V fragments of unit u + 1's key block.
template <int HD, int WHICH, bool DMA, bool STAT_PF, typename DmaFn>
__device__ __forceinline__ void phasekv(const CtxB4<HD>& cx, const f32x16& s_cur, const f32x16& dp_cur, f32x16& s_nxt, f32x16& dp_nxt,
                                        const u32x4 (&out_prev)[2], u32x4 (&out_cur)[2], f32x16 (&acc_prev)[HD / 32],
                                        const u32x4 (&kf)[HD / 16], const u32x4 (&vf)[HD / 16], const char* tp, int kbt,
                                        const float (&lq)[16], const float (&dq16)[16], float (&lq_pf)[16], float (&dq_pf)[16], const char* tn, int kbn,
                                        const char* rows, int kbr, int h, DmaFn&& dma) {
    // lq / dq16: the statistics of unit u's 32 queries, read from LDS a phase (or two) ago — read at the head of the phase that uses
    // them they cost a full LDS round trip in front of the first exponential.  STAT_PF: this phase reads the statistics of the NEXT
    // query half (tile tn, half kbn) into lq_pf / dq_pf, behind its second fragment block.
    using T = TK4<HD, WHICH>;
    ElState st;
    u32x4 fr[2][4];
    auto read_frag = [&](int f) -> u32x4 {
#ifdef BRA_A4_NOLDS
        { u32x4 z = {cx.kfo[0], cx.kfo[1], cx.tfo[0][0], 0x3c003c00u}; return z; }
#endif
        if (f < T::NACC) return ld16(tp + cx.tfo[kbt][f / T::DB] + (f % T::DB) * 4096);
        const int p = f - T::NACC;
        if constexpr (T::DK) return ld16(rows + cx.kfo[p >> 1] + (p & 1) * T::KBYTES + kbr * (32 * HD * 2));
        else return ld16(rows + cx.kfo[p] + kbr * (32 * HD * 2));
    };
#pragma unroll
    for (int f = 3; f >= 0; --f) fr[0][f] = read_frag(f);
#define BRA_KV_GROUP(G)                                                                                                    \
    if constexpr ((G) < T::NG) {                                                                                           \
        constexpr int blk_ = (G) / 4;                                                                                      \
        if constexpr ((G) < T::NACC) {                                                                                     \
            constexpr int s2_ = (G) / T::DB, db_ = (G) % T::DB;                                                            \
            mfma_o(acc_prev[db_], fr[blk_ & 1][(G) % 4], out_prev[s2_]);                                                   \
        } else if constexpr (T::DK) {                                                                                      \
            constexpr int p_ = (G) - T::NACC, ds_ = p_ >> 1;                                                               \
            if constexpr ((p_ & 1) == 0) {                                                                                 \
                if constexpr (ds_ == 0) { f32x16 z_ = {}; s_nxt = mfma_32x32x16(fr[blk_ & 1][(G) % 4], kf[0], z_); }        \
                else s_nxt = mfma_32x32x16(fr[blk_ & 1][(G) % 4], kf[ds_], s_nxt);                                         \
            } else {                                                                                                       \
                if constexpr (ds_ == 0) { f32x16 z_ = {}; dp_nxt = mfma_32x32x16(fr[blk_ & 1][(G) % 4], vf[0], z_); }       \
                else dp_nxt = mfma_32x32x16(fr[blk_ & 1][(G) % 4], vf[ds_], dp_nxt);                                       \
            }                                                                                                              \
        } else {                                                                                                           \
            constexpr int ds_ = (G) - T::NACC;                                                                             \
            if constexpr (ds_ == 0) { f32x16 z_ = {}; s_nxt = mfma_32x32x16(fr[blk_ & 1][(G) % 4], kf[0], z_); }            \
            else s_nxt = mfma_32x32x16(fr[blk_ & 1][(G) % 4], kf[ds_], s_nxt);                                             \
        }                                                                                                                  \
        if constexpr ((G) % 4 == 0 && (G) + 4 < T::NG) {                                                                   \
            sched_fence();                                                                                                 \
            _Pragma("unroll") for (int u_ = 3; u_ >= 0; --u_) fr[(blk_ + 1) & 1][u_] = read_frag((G) + 4 + u_);            \
        }                                                                                                                  \
        if constexpr (STAT_PF && (G) == 6) {                                                                               \
            read_stats16(lq_pf, reinterpret_cast<const float*>(tn + T::TBYTES) + 32 * kbn, h);                             \
            if constexpr (T::DK) read_stats16(dq_pf, reinterpret_cast<const float*>(tn + T::TBYTES + 256) + 32 * kbn, h);  \
        }                                                                                                                  \
        if constexpr (DMA && (G) < T::NDMA && !kNoDma) dma(G);                                                             \
        elkv_items<WHICH, T::item0(G, DMA), T::item0((G) + 1, DMA)>(s_cur, dp_cur, out_cur, st, lq, dq16, cx.sc, cx.scale); \
        sched_fence();                                                                                                     \
    }
    BRA_KV_GROUP(0) BRA_KV_GROUP(1) BRA_KV_GROUP(2) BRA_KV_GROUP(3) BRA_KV_GROUP(4) BRA_KV_GROUP(5) BRA_KV_GROUP(6) BRA_KV_GROUP(7)
    BRA_KV_GROUP(8) BRA_KV_GROUP(9) BRA_KV_GROUP(10) BRA_KV_GROUP(11) BRA_KV_GROUP(12) BRA_KV_GROUP(13) BRA_KV_GROUP(14) BRA_KV_GROUP(15)
    BRA_KV_GROUP(16) BRA_KV_GROUP(17) BRA_KV_GROUP(18) BRA_KV_GROUP(19) BRA_KV_GROUP(20) BRA_KV_GROUP(21) BRA_KV_GROUP(22) BRA_KV_GROUP(23)
#undef BRA_KV_GROUP
}

template <int HD, int WHICH>
__device__ __forceinline__ void cold_phasekv(const CtxB4<HD>& cx, const f32x16& s_cur, const f32x16& dp_cur, f32x16& s_nxt, f32x16& dp_nxt,
                                             const u32x4 (&out_prev)[2], u32x4 (&out_cur)[2], f32x16 (&acc_prev)[HD / 32],
                                             const u32x4 (&kf)[HD / 16], const u32x4 (&vf)[HD / 16], const char* tp, int kbt, const char* tc, int kbc,
                                             const char* rows, int kbr, int h, bool do_acc, bool do_el, bool do_sdp) {
    using T = TK4<HD, WHICH>;
    if (do_acc) {
#pragma unroll
        for (int f = 0; f < T::NACC; ++f) mfma_o(acc_prev[f % T::DB], ld16(tp + cx.tfo[kbt][f / T::DB] + (f % T::DB) * 4096), out_prev[f / T::DB]);
        mfma_drain();
    }
    if (do_el) {
        float lq[16], dq16[16], e[16];
        read_stats16(lq, reinterpret_cast<const float*>(tc + T::TBYTES) + 32 * kbc, h);
        if constexpr (T::DK) read_stats16(dq16, reinterpret_cast<const float*>(tc + T::TBYTES + 256) + 32 * kbc, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            e[r] = fast_exp2(fmaf(s_cur[r], cx.sc, -lq[r]));
            if constexpr (T::DK) e[r] *= fmaf(dp_cur[r], cx.scale, -dq16[r]);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            out_cur[g].x = pack_bf2(e[8 * g + 0], e[8 * g + 1]); out_cur[g].y = pack_bf2(e[8 * g + 2], e[8 * g + 3]);
            out_cur[g].z = pack_bf2(e[8 * g + 4], e[8 * g + 5]); out_cur[g].w = pack_bf2(e[8 * g + 6], e[8 * g + 7]);
        }
    }
    if (do_sdp) {
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) {
            const u32x4 qfr = ld16(rows + cx.kfo[ds] + kbr * (32 * HD * 2));
            if (ds == 0) { f32x16 z = {}; s_nxt = mfma_32x32x16(qfr, kf[0], z); } else s_nxt = mfma_32x32x16(qfr, kf[ds], s_nxt);
            if constexpr (T::DK) {
                const u32x4 dfr = ld16(rows + cx.kfo[ds] + T::KBYTES + kbr * (32 * HD * 2));
                if (ds == 0) { f32x16 z = {}; dp_nxt = mfma_32x32x16(dfr, vf[0], z); } else dp_nxt = mfma_32x32x16(dfr, vf[ds], dp_nxt);
            }
        }
    }
}

// scores of (query, key) pairs that do not exist -> kMasked.  Key on lanes (this lane's key is valid: kvalid), query on registers:
// register r <-> query 16 (r >> 3) + 8 h + (r & 7) of the half; visible iff that index >= lim (lim = key - q_off - first query - 8 h)
// and < nq (queries of the half that exist, minus 8 h)
__device__ __forceinline__ void mask_scores4k(f32x16& s, bool kvalid, bool causal, int lim, int nq) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qq = 16 * (r >> 3) + (r & 7);
        bool ok = kvalid && qq < nq;
        if (causal) ok = ok && qq >= lim;
        s[r] = ok ? s[r] : kMasked;
    }
}

template <int HD, int WHICH>
__global__ __launch_bounds__(256) void attn_dkv4_kernel(AttnArgs a) {
    using T = TK4<HD, WHICH>;
    constexpr bool DK = T::DK;
    BRA_DYN_SMEM(smem);                               // [2][Q rows | dO rows (dK)]  [3][transposed tile | lse2[64] | delta * scale[64]]
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6), h = lane >> 5, l31 = lane & 31;
    int bx_, hkv, b;
    attn_block_coords(0, 0, bx_, hkv, b);             // (causal: key block 0 is the heaviest — ascending order is heaviest first)
    const int group = a.Hq / a.Hkv;
    // the (q-head, query tile) iterations in `nsp` parts (bra_attn_bwd_split's nsplit_kv): part sp runs iterations [it_first, it_first + nit)
    // and leaves its accumulators in fp32 for attn_sum_parts_kernel; iteration / unit indices below are relative to it_first
    const int nsp = a.nsplit_kv > 1 ? a.nsplit_kv : 1;
    const int sp = nsp > 1 ? bx_ % nsp : 0;
    if (nsp > 1) bx_ /= nsp;
    const int k0 = bx_ * 256, kw0 = k0 + wave * 64;

    CtxB4<HD> cx;
    cx.sc = a.scale * kLog2e;
    cx.scale = a.scale;
    {
        const int row = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);      // query bits 2 / 3 swapped on the row side
        const int sw = (row >> T::RSH) & (T::CH - 1);
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) cx.kfo[ds] = (unsigned)(row * (HD * 2) + (((2 * ds + h) ^ sw) << 4));
        const int swv = (l31 >> 1) & 7;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) cx.tfo[kb][s2] = (unsigned)(l31 * 128 + (((4 * kb + 2 * s2 + h) ^ swv) << 4));
    }
    // this lane's two keys: fragments (B operands of S / dP) and validity
    u32x4 kf[2][T::DS], vf[2][T::DS];               // (vf: dK only; otherwise never written, never read)
    bool kvalid[2];
    int kj[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        kj[x] = kw0 + 32 * x + l31;
        const int kr = kj[x] < a.Sk ? kj[x] : a.Sk - 1;
        kvalid[x] = kj[x] < a.Sk;
        if (kvalid[x] && a.kmask) kvalid[x] = a.kmask[(long)b * a.Sk + kr] != 0;
        const bf16_t* kp = a.k + b * a.k_sb + (long)kr * a.k_ss + hkv * a.k_sh;
        const bf16_t* vp = a.v + b * a.v_sb + (long)kr * a.v_ss + hkv * a.v_sh;
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) {
            kf[x][ds] = ld16(kp + ds * 16 + 8 * h);
            if constexpr (DK) vf[x][ds] = ld16(vp + ds * 16 + 8 * h);
        }
    }
    const bool all_valid[2] = {wave_ballot(kvalid[0]) == ~0ull, wave_ballot(kvalid[1]) == ~0ull};
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) { to_agpr(kf[x][ds]); if constexpr (DK) to_agpr(vf[x][ds]); }
    f32x16 acc[2][T::DB];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < T::DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][i][r] = 0.f;

    // the workgroup's iterations: (q-head of the group, 64-query tile), tiles from the first one that can see a key of the workgroup
    int qt_begin = 0;
    if (a.causal) { const int first = k0 - a.q_off; qt_begin = first > 0 ? first / 64 : 0; }
    const int qt_end = (a.Sq + 63) / 64;
    const int per_head = qt_end > qt_begin ? qt_end - qt_begin : 0;
    const int nit_all = per_head * group;
    const int it_first = nsp > 1 ? (nit_all * sp) / nsp : 0;
    const int nit = (nsp > 1 ? (nit_all * (sp + 1)) / nsp : nit_all) - it_first;
    const int nunit = kw0 < a.Sk ? 4 * nit : 0;       // this wave's units (a wave without keys only takes part in the staging)

    // DMA sources: lane offsets inside the (batch, q-head) slices (loop constants); the slice bases change with the iteration's head
    const int q_sbytes = (int)a.q_ss * 2, do_sbytes = (int)a.do_ss * 2;
    const int t_sd2 = (int)(DK ? a.qt_sd : a.dot_sd) * 2;
    unsigned qsrc[T::KPW], dsrc[DK ? T::KPW : 1], tsrc[T::TPW];
#pragma unroll
    for (int i = 0; i < T::KPW; ++i) {
        const int u = 64 * (wave * T::KPW + i) + lane, row = u / T::CH, c = (u % T::CH) ^ ((row >> T::RSH) & (T::CH - 1));
        qsrc[i] = attn_mul24(row, q_sbytes) + (unsigned)(c * 16);
        if constexpr (DK) dsrc[i] = attn_mul24(row, do_sbytes) + (unsigned)(c * 16);
    }
#pragma unroll
    for (int i = 0; i < T::TPW; ++i) {
        const int u = 64 * (wave * T::TPW + i) + lane, d = u >> 3, c = (u & 7) ^ ((d >> 1) & 7);
        tsrc[i] = attn_mul24(d, t_sd2) + (unsigned)(c * 16);
    }
    auto it_head = [&](int it) -> int { return hkv * group + (per_head > 0 ? (it_first + it) / per_head : 0); };
    auto it_s0 = [&](int it) -> int { return (qt_begin + (per_head > 0 ? (it_first + it) % per_head : 0)) * 64; };
    float st_reg = 0.f;                               // staged statistic of the tile after next: lanes 0..63 of wave 0 lse2, of wave 1 delta * scale
    // the copies of one (q-head, query tile): descriptors of the head's slices + the tile's offsets, then NDMA pieces + the statistic
    struct Stage { BufDesc qd, dd, td; unsigned q_off, d_off, t_off; int rslot, tslot; long stat_idx; };
    auto stage_of = [&](int it, int rslot, int tslot) -> Stage {       // (beyond the last iteration: a harmless re-load of the last)
        const int itc = it < nit ? it : nit - 1;
        const int hq = it_head(itc), s0 = it_s0(itc);
        Stage g;
        g.qd = make_bufdesc(uniform_ptr(a.q + b * a.q_sb + hq * a.q_sh), (unsigned)((a.Sq - 1) * q_sbytes + HD * 2));
        g.dd = make_bufdesc(uniform_ptr(a.dout + b * a.do_sb + hq * a.do_sh), (unsigned)((a.Sq - 1) * do_sbytes + HD * 2));
        const bf16_t* tb = DK ? a.qt + b * a.qt_sb + hq * a.qt_sh : a.dot + b * a.dot_sb + hq * a.dot_sh;
        g.td = make_bufdesc(uniform_ptr(tb), (unsigned)(HD * t_sd2));
        g.q_off = (unsigned)s0 * (unsigned)q_sbytes; g.d_off = (unsigned)s0 * (unsigned)do_sbytes; g.t_off = (unsigned)(s0 * 2);
        g.rslot = rslot; g.tslot = tslot;
        int qq = s0 + lane; qq = qq < a.Sq ? qq : a.Sq - 1;
        g.stat_idx = ((long)b * a.Hq + hq) * a.Sq + qq;
        return g;
    };
    auto stage_piece = [&](const Stage& g, int i) {
        if (i < T::KPW) dma16(g.qd, qsrc[i], g.q_off, smem + g.rslot * T::RSLOT + (wave * T::KPW + i) * 1024);
        else if (DK && i < 2 * T::KPW) dma16(g.dd, dsrc[DK ? i - T::KPW : 0], g.d_off, smem + g.rslot * T::RSLOT + T::KBYTES + (wave * T::KPW + i - T::KPW) * 1024);
        else {
            const int j = i - (DK ? 2 : 1) * T::KPW;
            dma16(g.td, tsrc[j], g.t_off, smem + T::OFF_T + g.tslot * T::TSLOT + (wave * T::TPW + j) * 1024);
        }
    };
    auto stage_stat = [&](const Stage& g) {
        if (wave < 2) st_reg = wave == 0 ? a.lse[g.stat_idx] * kLog2e : a.delta[g.stat_idx] * a.scale;
    };
    auto stage = [&](int it, int rslot, int tslot) {
        const Stage g = stage_of(it, rslot, tslot);
#pragma unroll
        for (int i = 0; i < T::NDMA; ++i) stage_piece(g, i);
        stage_stat(g);
    };
    auto commit_stats = [&](int tslot) {               // (after the wait that covers the staged load, before the barrier that publishes it)
        if (wave < 2) reinterpret_cast<float*>(smem + T::OFF_T + tslot * T::TSLOT + T::TBYTES)[wave * 64 + lane] = st_reg;
    };

    f32x16 s[2], dp[2];
    u32x4 outp[2][2];
    if (nit > 0) {
        stage(0, 0, 0);
        wait_vmcnt<0>();
        commit_stats(0);
        raw_barrier();
        stage(1, 1, 1);
        auto noop = [](int) {};
        // masks of unit u (iteration u / 4, query half (u >> 1) & 1, key block u & 1)
        auto prep = [&](f32x16& sx, int u) {
            const int it = u >> 2, kbq = (u >> 1) & 1, x = u & 1;
            const int s0u = it_s0(it) + 32 * kbq, kwx = kw0 + 32 * x;
            const bool full = all_valid[x] && s0u + 31 < a.Sq && (!a.causal || kwx + 31 <= s0u + a.q_off);
            if (!full) {
                const int hh = opaque_i(lane) >> 5;
                mask_scores4k(sx, kvalid[x], a.causal != 0, kj[x] - a.q_off - s0u - 8 * hh, a.Sq - s0u - 8 * hh);
            }
        };
        const char* const tring = smem + T::OFF_T;
        auto cold = [&](int u) {
            const int x = u & 1;
            const bool do_acc = u >= 1 && u <= nunit, do_el = u < nunit, do_sdp = u + 1 < nunit;
            const int up = u > 0 ? u - 1 : 0, un = u + 1;
            const char* tp = tring + ((up >> 2) % 3) * T::TSLOT;
            const char* tc = tring + ((u >> 2) % 3) * T::TSLOT;
            const char* rows = smem + ((un >> 2) & 1) * T::RSLOT;
            if (do_el) { if (x == 0) prep(s[0], u); else prep(s[1], u); }
            if (x == 0) cold_phasekv<HD, WHICH>(cx, s[0], dp[0], s[1], dp[1], outp[1], outp[0], acc[1], kf[1], vf[1], tp, (up >> 1) & 1, tc, (u >> 1) & 1,
                                                rows, (un >> 1) & 1, h, do_acc, do_el, do_sdp);
            else cold_phasekv<HD, WHICH>(cx, s[1], dp[1], s[0], dp[0], outp[0], outp[1], acc[0], kf[0], vf[0], tp, (up >> 1) & 1, tc, (u >> 1) & 1,
                                         rows, (un >> 1) & 1, h, do_acc, do_el, do_sdp);
        };
        if (nunit > 0) {
            cold_phasekv<HD, WHICH>(cx, s[1], dp[1], s[0], dp[0], outp[0], outp[1], acc[0], kf[0], vf[0], tring, 0, tring, 0, smem, 0, h, false, false, true);
            cold(0); cold(1); cold(2);
        }
        // iteration t: phases 4 t + 3 .. 4 t + 6 (k_attn4b dQ kernel: the same ring discipline with query tiles in the place of key tiles)
        const int tmain = nunit >= 8 ? (nunit - 4) / 4 : 0;
        int t = 0;
        // statistics of the query halves in flight, by half of the tile (both key blocks of a half use the same values): the phase of
        // a half's SECOND key block reads the next half's (tile t + 1's were published by the barrier at the top of iteration t)
        float lq0[16], dq0[16], lq1[16], dq1[16];
        if (tmain > 0) {
            read_stats16(lq1, reinterpret_cast<const float*>(tring + T::TBYTES) + 32, h);                 // tile 0, half 1: phase 3
            if constexpr (DK) read_stats16(dq1, reinterpret_cast<const float*>(tring + T::TBYTES + 256) + 32, h);
        }
#define BRA_HOT_ITER(PAR)                                                                                                    \
        {                                                                                                                  \
            wait_vmcnt<0>();                                                                                               \
            commit_stats((t + 1) % 3);                                                                                     \
            raw_barrier();                                                                                                 \
            const Stage sg = stage_of(t + 2, PAR, (t + 2) % 3);                                                            \
            stage_stat(sg);                                                                                                \
            auto dma = [&](int i) { stage_piece(sg, i); };                                                                 \
            const char* tt_t = tring + (t % 3) * T::TSLOT;                                                                 \
            const char* tt_n = tring + ((t + 1) % 3) * T::TSLOT;                                                           \
            const char* rows_n = smem + (1 - (PAR)) * T::RSLOT;                                                            \
            prep(s[1], 4 * t + 3);                                                                                         \
            phasekv<HD, WHICH, true, true>(cx, s[1], dp[1], s[0], dp[0], outp[0], outp[1], acc[0], kf[0], vf[0], tt_t, 1, lq1, dq1, lq0, dq0, tt_n, 0, \
                                           rows_n, 0, h, dma);                                                             \
            prep(s[0], 4 * t + 4);                                                                                         \
            phasekv<HD, WHICH, false, false>(cx, s[0], dp[0], s[1], dp[1], outp[1], outp[0], acc[1], kf[1], vf[1], tt_t, 1, lq0, dq0, lq1, dq1, tt_n, 1, \
                                             rows_n, 0, h, noop);                                                          \
            prep(s[1], 4 * t + 5);                                                                                         \
            phasekv<HD, WHICH, false, true>(cx, s[1], dp[1], s[0], dp[0], outp[0], outp[1], acc[0], kf[0], vf[0], tt_n, 0, lq0, dq0, lq1, dq1, tt_n, 1, \
                                            rows_n, 1, h, noop);                                                           \
            prep(s[0], 4 * t + 6);                                                                                         \
            phasekv<HD, WHICH, false, false>(cx, s[0], dp[0], s[1], dp[1], outp[1], outp[0], acc[1], kf[1], vf[1], tt_n, 0, lq1, dq1, lq0, dq0, tt_n, 0, \
                                             rows_n, 1, h, noop);                                                          \
            ++t;                                                                                                           \
        }
        while (t + 1 < tmain) { BRA_HOT_ITER(0) BRA_HOT_ITER(1) }
        if (t < tmain) BRA_HOT_ITER(0)
#undef BRA_HOT_ITER
        for (; t < nit; ++t) {
            wait_vmcnt<0>();
            commit_stats((t + 1) % 3);
            raw_barrier();
            stage(t + 2, t & 1, (t + 2) % 3);
            cold(4 * t + 3); cold(4 * t + 4); cold(4 * t + 5); cold(4 * t + 6);
        }
        wait_vmcnt<0>();
    }

    mfma_drain();
    if (nsp > 1) {
        float* const part = DK ? a.part_dk : a.part_dv;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float* op = part + ((((long)b * a.Hkv + hkv) * nsp + sp) * a.Sk + (kj[x] < a.Sk ? kj[x] : 0)) * HD;
#pragma unroll
            for (int db = 0; db < T::DB; ++db) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w = {acc[x][db][4 * g + 0], acc[x][db][4 * g + 1], acc[x][db][4 * g + 2], acc[x][db][4 * g + 3]};
                    if (kj[x] < a.Sk) *reinterpret_cast<f32x4*>(op + db * 32 + 8 * g + 4 * h) = w;
                }
                sched_fence();
            }
        }
        return;
    }
    bf16_t* const outb = DK ? a.dk + b * a.dk_sb + hkv * a.dk_sh : a.dv + b * a.dv_sb + hkv * a.dv_sh;
    const long o_ss = DK ? a.dk_ss : a.dv_ss;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        bf16_t* op = outb + (long)(kj[x] < a.Sk ? kj[x] : 0) * o_ss;
#pragma unroll
        for (int db = 0; db < T::DB; ++db) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                uint32_t a0 = pack_bf2(acc[x][db][4 * g + 0], acc[x][db][4 * g + 1]);
                uint32_t a1 = pack_bf2(acc[x][db][4 * g + 2], acc[x][db][4 * g + 3]);
                uint32_t b0 = pack_bf2(acc[x][db][4 * g + 4], acc[x][db][4 * g + 5]);
                uint32_t b1 = pack_bf2(acc[x][db][4 * g + 6], acc[x][db][4 * g + 7]);
                xhalf_pair(a0, b0);
                xhalf_pair(a1, b1);
                u32x4 w = {a0, a1, b0, b1};
                if (kj[x] < a.Sk) st16(op + db * 32 + 8 * g + 8 * h, w);
            }
            sched_fence();
        }
    }
}

template <int HD>
int launch_dkv4(const AttnArgs& a, bra_stream_t st) {
    constexpr size_t smem_v = TK4<HD, 1>::SMEM, smem_k = TK4<HD, 2>::SMEM;
    const int ns = a.nsplit_kv > 1 ? a.nsplit_kv : 1;   // (the sums of the parts: attn_sum_parts_kernel, launched by the caller)
    const dim3 grid(((a.Sk + 255) / 256) * ns, a.Hkv, a.B);
    BRA_ALLOW_SMEM((attn_dkv4_kernel<HD, 1>), smem_v);
    BRA_LAUNCH((attn_dkv4_kernel<HD, 1>), grid, dim3(256), smem_v, st, a);
    int rc = BRA_LAUNCH_STATUS();
    if (rc) return rc;
    BRA_ALLOW_SMEM((attn_dkv4_kernel<HD, 2>), smem_k);
    BRA_LAUNCH((attn_dkv4_kernel<HD, 2>), grid, dim3(256), smem_k, st, a);
    return BRA_LAUNCH_STATUS();
}
template int launch_dkv4<128>(const AttnArgs&, bra_stream_t);
template int launch_dkv4<64>(const AttnArgs&, bra_stream_t);

template <int HD>
int launch_dq4(const AttnArgs& a, bra_stream_t st) {
    BRA_ALLOW_SMEM((attn_dq4_kernel<HD>), (size_t)TB4<HD>::SMEM);
    const int ns = a.nsplit > 1 ? a.nsplit : 1;      // (the sum of the parts: attn_sum_parts_kernel, launched by the caller)
    BRA_LAUNCH((attn_dq4_kernel<HD>), dim3(((a.Sq + 255) / 256) * ns, a.Hq, a.B), dim3(256), (size_t)TB4<HD>::SMEM, st, a);
    return BRA_LAUNCH_STATUS();
}
template int launch_dq4<128>(const AttnArgs&, bra_stream_t);
template int launch_dq4<64>(const AttnArgs&, bra_stream_t);

}  // namespace bra
