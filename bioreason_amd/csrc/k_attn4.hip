// k_attn4.hip — flash-style attention FORWARD for gfx950, second generation: 4 waves per workgroup, ONE wave per SIMD, 64 queries
// (two 32-row blocks A / B) per wave, the softmax of one 32-key step placed by hand inside the MFMA runs of its neighbours.
// Same arithmetic statement as attn_fwd_kernel (k_attn.hip; TF:qwen3:185-207 causal GQA hd 128, TF:esm:292-317 bidirectional hd 64):
//   S^T[key][q] = K[key][:] . Q[q][:]      (v_mfma_f32_32x32x16_bf16; a lane owns ONE query and 16 of the 32 keys of a step)
//   O^T[d][q]  += V^T[d][key] . P^T[key][q]  (the softmax registers ARE the B operand)
// What is different from the 8-wave kernel, and why (NOTES.md rounds 2 - 5: that kernel spends 7000 cycles per 64-key tile on 2048
// cycles of matrix work — both waves of a SIMD run the same phase between barriers, and every wave re-reads the whole tile from LDS):
//   * every K / V^T fragment read from LDS feeds TWO MFMAs (query blocks A and B): half the LDS bytes per FLOP;
//   * software pipeline over 32-key steps j:   phase A: PV(j-1) MFMAs  ||  row max of S(j), first exponentials of step j
//                                              phase B: QK(j+1) MFMAs  ||  remaining exponentials, row sums, bf16 packing
//     written as groups {1 MFMA + its share of the step's VALU work + at most one LDS read / DMA piece} separated by
//     sched_barrier(0): the source order IS the schedule (the compiler does not interleave a wave's softmax with its own MFMAs
//     by itself — measured in round 2);
//   * the exponentials do not wait for the step's row maximum: they are taken against the RUNNING maximum (speculatively), and a
//     wave-uniform slow path redoes the step and rescales O only when some row's maximum grew by more than 2^kThr ("defer-max");
//   * K and V^T tiles arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) into two slots each, issued one
//     whole tile ahead from inside phase A; one barrier per 64-key tile;
//   * the K-side MFMA rows are taken with bits 2 / 3 of the key index swapped, so that the 8 probabilities a lane packs for one
//     PV k-slot group are 8 CONSECUTIVE keys: the V^T fragment is one natural 16-byte unit of the transposed image
//     (no regrouping stores; the tile is copied by DMA as it lies in memory).
#include "bra_device.h"
#include "bra_api_internal.h"
#include "bra_attn.h"

namespace bra {

constexpr float kThr = 8.0f;            // log2 units: probabilities of a step are at most 2^8 against the running maximum
constexpr float kMasked = -3.0e38f;     // raw score of a masked key (finite: times `sc` it stays finite, exp2 gives 0)

#ifdef BRA_EMU
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float max2f(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float xhalf_max(float v) { return fmaxf(v, wave_shfl_xor(v, 32)); }
__device__ __forceinline__ float xhalf_sum(float v) { return v + wave_shfl_xor(v, 32); }
// lanes 0..31 receive {a (own), a of lane + 32}; lanes 32..63 receive {b of lane - 32, b (own)}: T21's widened row store
__device__ __forceinline__ void xhalf_pair(uint32_t& a, uint32_t& b) {
    const uint32_t ao = wave_shfl_xor_u32(a, 32), bo = wave_shfl_xor_u32(b, 32);
    if (bra_emu::lane_id() < 32) b = ao; else a = bo;
}
#else
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (fmaxf on a value the compiler cannot prove to be a quiet number is preceded by a canonicalising v_max_f32 x, x)
__device__ __forceinline__ float max2f(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float xhalf_max(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return max2f(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
}
__device__ __forceinline__ void xhalf_pair(uint32_t& a, uint32_t& b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);       // vdst = a: its upper half swaps with b's lower half
    a = r[0]; b = r[1];
}
#endif

// O^T += V^T . P^T with the accumulator pinned to the accumulator half of the register file (this file is built with
// -amdgpu-mfma-vgpr-form, so every builtin MFMA — the score tiles the softmax reads — has an architectural destination; left to the
// compiler the 128 O registers either take that half too or are copied around every branch).  An asm MFMA is opaque to the hazard
// recogniser: a VALU read of O (rescale, epilogue) must be preceded by mfma_drain(); MFMA -> MFMA on the same accumulator needs nothing.
#ifdef BRA_EMU
__device__ __forceinline__ void mfma_o(f32x16& o, const u32x4& a, const u32x4& b) { o = mfma_32x32x16(a, b, o); }
__device__ __forceinline__ void mfma_drain() {}
#else
__device__ __forceinline__ void mfma_o(f32x16& o, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory"); }
#endif
#ifdef BRA_EMU
__device__ __forceinline__ void to_agpr(u32x4&) {}
#else
__device__ __forceinline__ void to_agpr(u32x4& v) { asm volatile("" : "+a"(v)); }
#endif
#ifdef BRA_EMU
__device__ __forceinline__ void pin_u32_f32(uint32_t&, float&) {}
#else
__device__ __forceinline__ void pin_u32_f32(uint32_t& a, float& b) { asm volatile("" : "+v"(a), "+v"(b)); }
#endif

template <int HD>
struct T4 {
    static constexpr int CH = HD / 8;                 // 16-byte chunks per K row
    static constexpr int DS = HD / 16;                // contraction steps over d (QK^T)
    static constexpr int DB = HD / 32;                // 32-wide d blocks (PV)
    static constexpr int RSH = HD == 128 ? 0 : (HD == 64 ? 1 : 2);   // rows per swizzle step = 16 / CH
    static constexpr int KBYTES = 64 * HD * 2;        // [64 keys][HD]
    static constexpr int TBYTES = HD * 128;           // [HD][64 keys]
    static constexpr int KPW = KBYTES / 4096;         // 1-KiB DMA pieces per wave and tile
    static constexpr int TPW = TBYTES / 4096;
    static constexpr int NGA = 4 * DB;                // MFMAs of PV(j-1):  2 k-slot groups x DB x 2 query blocks
    static constexpr int NGB = 2 * DS;                // MFMAs of QK(j+1):  DS x 2 query blocks
    static constexpr int NG = NGA + NGB;
    static constexpr int NITEM = 24 + 112;            // single-instruction work items of one step's softmax (see sm_item)
};

// One step's softmax as NITEM single-instruction items in dependency order; group G of the step executes items
// [G * NITEM / NG, (G + 1) * NITEM / NG).  s: the step's raw scores (two query blocks), p: its packed probabilities [block][k-slot group],
// st: scratch that lives across items.
struct SmState {
    float mx[2];        // raw row maximum (this lane's 16 keys, then both halves)
    float t[2];         // mx * sc - m_run
    float e[2][2];      // the two exponentials of the pair being packed
    float x[2];
    float rs[2];        // this lane's row sums of the step
    uint32_t sw[2][2];
};

template <int K>
__device__ __forceinline__ void sm_item(const f32x16 (&s)[2], u32x4 (&p)[2][2], SmState& st, const float (&m_run)[2], float sc) {
    if constexpr (K < 16) {                           // row maximum, blocks interleaved: 8 v_max3 / v_max per block
        constexpr int c = K >> 1, qb = K & 1;
        if constexpr (c == 0) st.mx[qb] = max3f(s[qb][0], s[qb][1], s[qb][2]);
        else if constexpr (c < 7) st.mx[qb] = max3f(st.mx[qb], s[qb][2 * c + 1], s[qb][2 * c + 2]);
        else st.mx[qb] = max2f(st.mx[qb], s[qb][15]);
    } else if constexpr (K < 20) {                    // the other 16 keys of the row sit in lane ^ 32
        constexpr int qb = (K - 16) >> 1;
        if constexpr (((K - 16) & 1) == 0) st.mx[qb] = xhalf_max(st.mx[qb]);
    } else if constexpr (K < 24) {
        constexpr int qb = (K - 20) >> 1;
        if constexpr (((K - 20) & 1) == 0) st.t[qb] = fmaf(st.mx[qb], sc, -m_run[qb]);
    } else {
        constexpr int kk = K - 24, i = kk / 14, w = kk % 14, qb = w / 7, u = w % 7;      // pair i of block qb
        if constexpr (u == 0) st.x[0] = fmaf(s[qb][2 * i], sc, -m_run[qb]);
        else if constexpr (u == 1) st.e[qb][0] = fast_exp2(st.x[0]);
        else if constexpr (u == 2) st.rs[qb] += st.e[qb][0];
        else if constexpr (u == 3) st.x[1] = fmaf(s[qb][2 * i + 1], sc, -m_run[qb]);
        else if constexpr (u == 4) st.e[qb][1] = fast_exp2(st.x[1]);
        else if constexpr (u == 5) st.rs[qb] += st.e[qb][1];
        else {
            uint32_t w2 = pack_bf2(st.e[qb][0], st.e[qb][1]);
            // the speculative results are only USED on the no-rescale path: without a pin the compiler sinks every exponential, sum
            // and pack of the step out of the MFMA groups into that successor block (seen in the first build's ISA)
            pin_u32_f32(w2, st.rs[qb]);
            constexpr int g = i >> 2, c4 = i & 3;     // registers 8 g .. 8 g + 7 = the 8 keys of k-slot group g
            if constexpr (c4 == 0) p[qb][g].x = w2; else if constexpr (c4 == 1) p[qb][g].y = w2;
            else if constexpr (c4 == 2) p[qb][g].z = w2; else p[qb][g].w = w2;
        }
    }
}
template <int LO, int HI>
__device__ __forceinline__ void sm_items(const f32x16 (&s)[2], u32x4 (&p)[2][2], SmState& st, const float (&m_run)[2], float sc) {
    if constexpr (LO < HI) {
        sm_item<LO>(s, p, st, m_run, sc);
        sm_items<LO + 1, HI>(s, p, st, m_run, sc);
    }
}

// everything a step needs that does not change inside the tile loop
template <int HD>
struct Ctx4 {
    unsigned kfo[T4<HD>::DS];     // LDS byte offset of this lane's K fragment of d-step ds (row pi(lane & 31) of a 32-key half)
    unsigned vfo[2][2];           // LDS byte offset of this lane's V^T fragment of (key half kb, k-slot group s2), d block 0
    float sc;
};

// One pipeline step.  s_cur: scores of step j (complete); s_nxt: receives QK(j+1); p_prev: probabilities of step j-1 (PV(j-1)
// accumulates them into o); p_cur: receives the probabilities of step j.
//   vt: V^T tile holding step j-1 (kbv = its key half);  kt: K tile holding step j+1 (kbk = its key half).
// DMA: this step's phase A also issues the workgroup's next tile pair (dma(i), i < NDMA).
template <int HD, bool DO_PV, bool DO_QK, bool DMA, typename DmaFn>
__device__ __forceinline__ void step4(const Ctx4<HD>& cx, const f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2], const u32x4 (&p_prev)[2][2],
                                      u32x4 (&p_cur)[2][2], f32x16 (&o)[2][T4<HD>::DB], float (&m_run)[2], float (&l_run)[2],
                                      const u32x4 (&qf)[2][T4<HD>::DS], const char* vt, int kbv, const char* kt, int kbk, DmaFn&& dma) {
    using T = T4<HD>;
    constexpr int NDMA = T::KPW + T::TPW;
    SmState st;
    st.rs[0] = 0.f; st.rs[1] = 0.f;
    constexpr int VL = 2, KL = 2;                     // fragments read ahead of their first MFMA
    u32x4 vf[T::NGA / 2 + VL], kf[T::DS + KL];
    // ---- phase A: PV(j-1) ------------------------------------------------------------------------------------------------
    if constexpr (DO_PV) {
#pragma unroll
        for (int f = 0; f < VL; ++f) vf[f] = ld16(vt + cx.vfo[kbv][f / T::DB] + (f % T::DB) * 4096);
    }
#define BRA_A_GROUP(G)                                                                                                     \
    {                                                                                                                      \
        constexpr int f_ = (G) / 2, qb_ = (G) & 1, s2_ = f_ / T::DB, db_ = f_ % T::DB;                                      \
        if constexpr (DO_PV) mfma_o(o[qb_][db_], vf[f_], p_prev[qb_][s2_]);                            \
        if constexpr (DO_PV && qb_ == 0 && f_ + VL < T::NGA / 2)                                                           \
            vf[f_ + VL] = ld16(vt + cx.vfo[kbv][(f_ + VL) / T::DB] + ((f_ + VL) % T::DB) * 4096);                          \
        if constexpr (DO_QK && (G) >= T::NGA - KL) kf[(G) - (T::NGA - KL)] = ld16(kt + cx.kfo[(G) - (T::NGA - KL)] + kbk * (32 * HD * 2)); \
        if constexpr (DMA && (G) < NDMA) dma(G);                                                                           \
        sm_items<((G) * T::NITEM) / T::NG, (((G) + 1) * T::NITEM) / T::NG>(s_cur, p_cur, st, m_run, cx.sc);                 \
        sched_fence();                                                                                                     \
    }
    // (macro-unrolled: a `for` over G would make the MFMA / read indices run-time values until the unroller has run, and the
    //  groups' order would no longer be the written one)
    BRA_A_GROUP(0) BRA_A_GROUP(1) BRA_A_GROUP(2) BRA_A_GROUP(3) BRA_A_GROUP(4) BRA_A_GROUP(5) BRA_A_GROUP(6) BRA_A_GROUP(7)
    if constexpr (T::NGA > 8) {
        BRA_A_GROUP(8) BRA_A_GROUP(9) BRA_A_GROUP(10) BRA_A_GROUP(11) BRA_A_GROUP(12) BRA_A_GROUP(13) BRA_A_GROUP(14) BRA_A_GROUP(15)
    }
#undef BRA_A_GROUP
    // ---- phase B: QK(j+1) ------------------------------------------------------------------------------------------------
#define BRA_B_GROUP(G)                                                                                                     \
    {                                                                                                                      \
        constexpr int ds_ = (G) / 2, qb_ = (G) & 1;                                                                         \
        if constexpr (DO_QK) {                                                                                             \
            if constexpr (ds_ == 0) { f32x16 z_ = {}; s_nxt[qb_] = mfma_32x32x16(kf[0], qf[qb_][0], z_); }                  \
            else s_nxt[qb_] = mfma_32x32x16(kf[ds_], qf[qb_][ds_], s_nxt[qb_]);                                            \
            if constexpr (qb_ == 0 && ds_ + KL < T::DS) kf[ds_ + KL] = ld16(kt + cx.kfo[ds_ + KL] + kbk * (32 * HD * 2));   \
        }                                                                                                                  \
        sm_items<((T::NGA + (G)) * T::NITEM) / T::NG, ((T::NGA + (G) + 1) * T::NITEM) / T::NG>(s_cur, p_cur, st, m_run, cx.sc); \
        sched_fence();                                                                                                     \
    }
    BRA_B_GROUP(0) BRA_B_GROUP(1) BRA_B_GROUP(2) BRA_B_GROUP(3) BRA_B_GROUP(4) BRA_B_GROUP(5) BRA_B_GROUP(6) BRA_B_GROUP(7)
    if constexpr (T::NGB > 8) {
        BRA_B_GROUP(8) BRA_B_GROUP(9) BRA_B_GROUP(10) BRA_B_GROUP(11) BRA_B_GROUP(12) BRA_B_GROUP(13) BRA_B_GROUP(14) BRA_B_GROUP(15)
    }
#undef BRA_B_GROUP
    // ---- the rare path: some row's maximum grew past the threshold (always: the first step of a block) ---------------------
    const bool grow = st.t[0] > kThr || st.t[1] > kThr;
    if (wave_ballot(grow) != 0ull) {
        mfma_drain();
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float m_new = fmaxf(m_run[qb], st.mx[qb] * cx.sc);
            const float alpha = fast_exp2(m_run[qb] - m_new);
            m_run[qb] = m_new;
            l_run[qb] *= alpha;
#pragma unroll
            for (int db = 0; db < T::DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
            float rs = 0.f;
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { e[r] = fast_exp2(fmaf(s_cur[qb][r], cx.sc, -m_new)); rs += e[r]; }
            st.rs[qb] = rs;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                p_cur[qb][g].x = pack_bf2(e[8 * g + 0], e[8 * g + 1]); p_cur[qb][g].y = pack_bf2(e[8 * g + 2], e[8 * g + 3]);
                p_cur[qb][g].z = pack_bf2(e[8 * g + 4], e[8 * g + 5]); p_cur[qb][g].w = pack_bf2(e[8 * g + 6], e[8 * g + 7]);
            }
        }
    }
    l_run[0] += st.rs[0];
    l_run[1] += st.rs[1];
}

// The same step without the interleave and with run-time switches: the first step of a block, and the steps at which a wave runs
// out of visible keys while its workgroup still has tiles to stage (causal masks: the waves of a workgroup end at different steps).
// Exact online softmax here (the running maximum follows every step) — both forms are the same sum, the hot loop only defers
// rescaling.  Never on the critical path of a long loop: at most three of these per wave and block.
template <int HD>
__device__ __forceinline__ void cold_step4(const Ctx4<HD>& cx, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2], const u32x4 (&p_prev)[2][2],
                                           u32x4 (&p_cur)[2][2], f32x16 (&o)[2][T4<HD>::DB], float (&m_run)[2], float (&l_run)[2],
                                           const u32x4 (&qf)[2][T4<HD>::DS], const char* vt, int kbv, const char* kt, int kbk,
                                           bool do_pv, bool do_sm, bool do_qk) {
    using T = T4<HD>;
    if (do_pv) {
#pragma unroll
        for (int f = 0; f < T::NGA / 2; ++f) {
            const u32x4 vf = ld16(vt + cx.vfo[kbv][f / T::DB] + (f % T::DB) * 4096);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) mfma_o(o[qb][f % T::DB], vf, p_prev[qb][f / T::DB]);
        }
        mfma_drain();
    }
    if (do_sm) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = s_cur[qb][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s_cur[qb][r]);
            mx = xhalf_max(mx);
            const float m_new = fmaxf(m_run[qb], mx * cx.sc);
            const float alpha = fast_exp2(m_run[qb] - m_new);
            m_run[qb] = m_new;
            if (wave_ballot(alpha != 1.f) != 0ull) {
#pragma unroll
                for (int db = 0; db < T::DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
            }
            float rs = 0.f, e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { e[r] = fast_exp2(fmaf(s_cur[qb][r], cx.sc, -m_new)); rs += e[r]; }
            l_run[qb] = l_run[qb] * alpha + rs;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                p_cur[qb][g].x = pack_bf2(e[8 * g + 0], e[8 * g + 1]); p_cur[qb][g].y = pack_bf2(e[8 * g + 2], e[8 * g + 3]);
                p_cur[qb][g].z = pack_bf2(e[8 * g + 4], e[8 * g + 5]); p_cur[qb][g].w = pack_bf2(e[8 * g + 6], e[8 * g + 7]);
            }
        }
    }
    if (do_qk) {
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) {
            const u32x4 kf = ld16(kt + cx.kfo[ds] + kbk * (32 * HD * 2));
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (ds == 0) { f32x16 z = {}; s_nxt[qb] = mfma_32x32x16(kf, qf[qb][0], z); }
                else s_nxt[qb] = mfma_32x32x16(kf, qf[qb][ds], s_nxt[qb]);
            }
        }
    }
}

// scores of masked keys -> kMasked.  Register r of a lane in half h holds key 16 (r >> 3) + 8 h + (r & 7) of the step.
__device__ __forceinline__ void mask_scores4(f32x16 (&s)[2], uint32_t valid32, bool causal, int lim0, int h) {
    // lim0: (query of block A) + q_off - (first key of the step); block B's queries are 32 further on
    const uint32_t vb = valid32 >> (8 * h);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int lim = lim0 + 32 * qb - 8 * h;       // key (16 (r >> 3) + (r & 7)) visible iff <= lim
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = 16 * (r >> 3) + (r & 7);
            bool ok = (vb >> kk) & 1u;
            if (causal) ok = ok && kk <= lim;
            s[qb][r] = ok ? s[qb][r] : kMasked;
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_fwd4_kernel(AttnArgs a) {
    using T = T4<HD>;
    constexpr int SLOT = T::KBYTES + T::TBYTES;
    BRA_DYN_SMEM(smem);                               // [2][K tile | V^T tile]
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6), h = lane >> 5, l31 = lane & 31;
    int bx_, hq, b;
    attn_block_coords(0, a.causal, bx_, hq, b);
    const int hkv = hq / (a.Hq / a.Hkv);
    const int q0 = bx_ * 256, qw0 = q0 + wave * 64;
    const bf16_t* kb_ = a.k + b * a.k_sb + hkv * a.k_sh;
    const bf16_t* vtb = a.vt + b * a.vt_sb + hkv * a.vt_sh;

    Ctx4<HD> cx;
    cx.sc = a.scale * kLog2e;
    {
        // K fragment rows: MFMA row i = lane & 31 takes key pi(i) = i with bits 2 and 3 swapped
        const int row = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
        const int sw = (row >> T::RSH) & (T::CH - 1);
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) cx.kfo[ds] = (unsigned)(row * (HD * 2) + (((2 * ds + h) ^ sw) << 4));
        const int swv = (l31 >> 1) & 7;               // V^T rows d = 32 db + (lane & 31): (d >> 1) & 7 does not depend on db
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) cx.vfo[kb][s2] = (unsigned)(T::KBYTES + l31 * 128 + (((4 * kb + 2 * s2 + h) ^ swv) << 4));
    }
    // DMA sources of this wave's pieces: K piece i covers LDS units 64 (wave KPW + i) + lane of the K tile, V^T piece i likewise
    int krow[T::KPW]; unsigned kcol[T::KPW], vsrc[T::TPW];
#pragma unroll
    for (int i = 0; i < T::KPW; ++i) {
        const int u = 64 * (wave * T::KPW + i) + lane, row = u / T::CH, c = (u % T::CH) ^ ((row >> T::RSH) & (T::CH - 1));
        krow[i] = row; kcol[i] = (unsigned)(c * 8);
    }
#pragma unroll
    for (int i = 0; i < T::TPW; ++i) {
        const int u = 64 * (wave * T::TPW + i) + lane, d = u >> 3, c = (u & 7) ^ ((d >> 1) & 7);
        vsrc[i] = attn_mul24(d, (int)a.vt_sd) + (unsigned)(c * 8);
    }
    auto dma_k = [&](int i, int tile, int slot) {
        int rr = tile * 64 + krow[i];
        rr = rr < a.Sk ? rr : a.Sk - 1;
        glds16(kb_ + (attn_mul24(rr, (int)a.k_ss) + kcol[i]), smem + slot * SLOT + (wave * T::KPW + i) * 1024);
    };
    auto dma_v = [&](int i, int tile, int slot) {
        glds16(vtb + (vsrc[i] + (unsigned)(tile * 64)), smem + slot * SLOT + T::KBYTES + (wave * T::TPW + i) * 1024);
    };

    // this workgroup's key tiles, this wave's steps
    int kv_end = a.Sk;
    if (a.causal) { const int last = q0 + 255 + a.q_off + 1; kv_end = last < kv_end ? last : kv_end; }
    const int ntile = kv_end > 0 ? (kv_end + 63) / 64 : 0;
    int nstep_w = 0;
    if (qw0 < a.Sq && ntile > 0) {
        int lastq = qw0 + 63; lastq = lastq < a.Sq ? lastq : a.Sq - 1;
        int lastk = a.causal ? lastq + a.q_off : a.Sk - 1;
        lastk = lastk < a.Sk ? lastk : a.Sk - 1;
        nstep_w = lastk >= 0 ? lastk / 32 + 1 : 0;
        nstep_w = nstep_w < 2 * ntile ? nstep_w : 2 * ntile;
    }

    // Q fragments (B operand of S^T): row = query, 8 d per lane and d-step
    u32x4 qf[2][T::DS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = qw0 + 32 * qb + l31;
        qr = qr < a.Sq ? qr : a.Sq - 1;
        const bf16_t* qp = a.q + b * a.q_sb + (long)qr * a.q_ss + hq * a.q_sh;
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) qf[qb][ds] = ld16(qp + ds * 16 + 8 * h);
    }
    // the Q fragments live in the accumulator half of the register file for the whole block (MFMA B operands may be AGPRs): the
    // 256 architectural VGPRs are needed for two score sets, two probability sets and the fragments in flight
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) to_agpr(qf[qb][ds]);
    f32x16 o[2][T::DB];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < T::DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
    float m_run[2] = {kNeg, kNeg}, l_run[2] = {0.f, 0.f};

    // key validity of a tile as a 64-bit word (bit = key of the tile); requested one tile ahead
    auto mask_byte = [&](int tile) -> int {
        int kj = tile * 64 + lane;
        const bool in = kj < a.Sk;
        kj = in ? kj : a.Sk - 1;
        int v = a.kmask ? (int)a.kmask[(long)b * a.Sk + kj] : 1;
        return in ? v : 0;
    };

    f32x16 s0[2], s1[2];                              // scores of even / odd steps
    u32x4 p0[2][2], p1[2][2];                         // probabilities of even / odd steps
    if (ntile > 0) {
        // ---- prologue: tiles 0 (and K of tile 1), step 0 ------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < T::KPW; ++i) dma_k(i, 0, 0);
#pragma unroll
        for (int i = 0; i < T::TPW; ++i) dma_v(i, 0, 0);
        int mb_next = mask_byte(0);
        int mb_next2 = mask_byte(1);
        wait_vmcnt<0>();
        raw_barrier();
        {
            const int t1 = ntile > 1 ? 1 : 0;
#pragma unroll
            for (int i = 0; i < T::KPW; ++i) dma_k(i, t1, 1);
        }
        uint64_t vcur = wave_ballot(mb_next != 0), vnext = wave_ballot(mb_next2 != 0);
        int mb_pend = mask_byte(2);
        auto noop = [](int) {};
        auto prep = [&](f32x16 (&s)[2], int j, uint64_t vword) {          // masks of step j, applied to its finished scores
            const int kv0s = 32 * j;
            const uint32_t v32 = (uint32_t)(vword >> (32 * (j & 1)));
            const bool full = v32 == 0xffffffffu && (!a.causal || kv0s + 31 <= qw0 + a.q_off);
            if (!full) mask_scores4(s, v32, a.causal != 0, qw0 + l31 + a.q_off - kv0s, opaque_i(lane) >> 5);
        };
        if (nstep_w > 0) {
            // QK(0) into s0 (the "next" scores of a step that does nothing else), then step 0 without a PV
            cold_step4<HD>(cx, s1, s0, p1, p0, o, m_run, l_run, qf, smem, 0, smem, 0, false, false, true);
            prep(s0, 0, vcur);
            cold_step4<HD>(cx, s0, s1, p1, p0, o, m_run, l_run, qf, smem, 0, smem, 1, false, true, nstep_w > 1);
        }
        // ---- iteration t: steps 2 t + 1 and 2 t + 2.  Hot loop: the iterations in which this wave runs both steps in full --------
        const int tmain = nstep_w >= 4 ? (nstep_w - 2) / 2 : 0;             // 2 t + 3 < nstep_w
        int t = 0;
        for (; t < tmain; ++t) {
            wait_vmcnt<0>();                          // everything issued one iteration ago (K(t + 1), V(t)) has landed ...
            raw_barrier();                            // ... for every wave, and every wave is done with K(t) and V(t - 1)
            int tk = t + 2, tv = t + 1;               // (beyond the last tile: a harmless re-load of the last one into the free slot)
            tk = tk < ntile ? tk : ntile - 1;
            tv = tv < ntile ? tv : ntile - 1;
            const int ks = t & 1, vs = (t + 1) & 1;
            auto dma = [&](int i) {
                if (i < T::KPW) dma_k(i, tk, ks); else dma_v(i - T::KPW, tv, vs);
            };
            const char* slot_t = smem + (t & 1) * SLOT;             // tile t
            const char* slot_n = smem + ((t + 1) & 1) * SLOT;       // tile t + 1
            // step 2 t + 1: PV(2 t) from V(t) half 0, softmax of S(2 t + 1), QK(2 t + 2) from K(t + 1) half 0
            prep(s1, 2 * t + 1, vcur);
            step4<HD, true, true, true>(cx, s1, s0, p0, p1, o, m_run, l_run, qf, slot_t, 0, slot_n, 0, dma);
            // step 2 t + 2: PV(2 t + 1) from V(t) half 1, softmax of S(2 t + 2), QK(2 t + 3) from K(t + 1) half 1
            prep(s0, 2 * t + 2, vnext);
            step4<HD, true, true, false>(cx, s0, s1, p1, p0, o, m_run, l_run, qf, slot_t, 1, slot_n, 1, noop);
            vcur = vnext;
            vnext = wave_ballot(mb_pend != 0);
            mb_pend = mask_byte(t + 3);
        }
        // ---- the remaining iterations: this wave's last steps, then only its share of the staging -----------------------------
        for (; t < ntile; ++t) {
            wait_vmcnt<0>();
            raw_barrier();
            int tk = t + 2, tv = t + 1;
            tk = tk < ntile ? tk : ntile - 1;
            tv = tv < ntile ? tv : ntile - 1;
#pragma unroll
            for (int i = 0; i < T::KPW; ++i) dma_k(i, tk, t & 1);
#pragma unroll
            for (int i = 0; i < T::TPW; ++i) dma_v(i, tv, (t + 1) & 1);
            const char* slot_t = smem + (t & 1) * SLOT;
            const char* slot_n = smem + ((t + 1) & 1) * SLOT;
            const int j1 = 2 * t + 1, j2 = 2 * t + 2;
            if (j1 <= nstep_w) {
                if (j1 < nstep_w) prep(s1, j1, vcur);
                cold_step4<HD>(cx, s1, s0, p0, p1, o, m_run, l_run, qf, slot_t, 0, slot_n, 0, true, j1 < nstep_w, j1 + 1 < nstep_w);
            }
            if (j2 <= nstep_w) {
                if (j2 < nstep_w) prep(s0, j2, vnext);
                cold_step4<HD>(cx, s0, s1, p1, p0, o, m_run, l_run, qf, slot_t, 1, slot_n, 1, true, j2 < nstep_w, j2 + 1 < nstep_w);
            }
            vcur = vnext;
            vnext = wave_ballot(mb_pend != 0);
            mb_pend = mask_byte(t + 3);
        }
        wait_vmcnt<0>();                              // (the tail's redundant tile loads must not outlive the workgroup's LDS)
    }

    mfma_drain();
    // ---- epilogue: normalise, bf16, whole 16-byte pieces of a row per store (lane pairs exchange their 8-byte halves) ---------------
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = qw0 + 32 * qb + l31;
        const float l_tot = xhalf_sum(l_run[qb]);
        const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
        bf16_t* op = a.o + b * a.o_sb + (long)(qi < a.Sq ? qi : 0) * a.o_ss + hq * a.o_sh;
#pragma unroll
        for (int db = 0; db < T::DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                // this lane: d = 32 db + 8 g + 4 h + 0..3 (group g) and 32 db + 8 (g + 1) + 4 h + 0..3 (group g + 1)
                uint32_t a0 = pack_bf2(o[qb][db][4 * g + 0] * inv, o[qb][db][4 * g + 1] * inv);
                uint32_t a1 = pack_bf2(o[qb][db][4 * g + 2] * inv, o[qb][db][4 * g + 3] * inv);
                uint32_t b0 = pack_bf2(o[qb][db][4 * g + 4] * inv, o[qb][db][4 * g + 5] * inv);
                uint32_t b1 = pack_bf2(o[qb][db][4 * g + 6] * inv, o[qb][db][4 * g + 7] * inv);
                xhalf_pair(a0, b0);
                xhalf_pair(a1, b1);
                // lower half: [own group g | partner's group g] = d 32 db + 8 g + 0..7; upper half: d 32 db + 8 (g + 1) + 0..7
                u32x4 w = {a0, a1, b0, b1};
                if (qi < a.Sq) st16(op + db * 32 + 8 * g + 8 * h, w);
            }
        if (a.lse && h == 0 && qi < a.Sq)
            a.lse[((long)b * a.Hq + hq) * a.Sq + qi] = l_tot > 0.f ? (m_run[qb] + log2f(l_tot)) * kLn2 : kNeg;
    }
}

template <int HD>
int launch_fwd4(const AttnArgs& a, bra_stream_t st) {
    const size_t smem = 2 * (T4<HD>::KBYTES + T4<HD>::TBYTES);
    BRA_ALLOW_SMEM((attn_fwd4_kernel<HD>), smem);
    BRA_LAUNCH((attn_fwd4_kernel<HD>), dim3((a.Sq + 255) / 256, a.Hq, a.B), dim3(256), smem, st, a);
    return BRA_LAUNCH_STATUS();
}
template int launch_fwd4<128>(const AttnArgs&, bra_stream_t);
template int launch_fwd4<64>(const AttnArgs&, bra_stream_t);

}  // namespace bra
