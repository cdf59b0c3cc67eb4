// k_attn4.hip — flash-style attention FORWARD for gfx950, second generation: every wave runs its OWN software pipeline over 32-key
// steps, with the softmax of a step placed by hand inside the MFMA runs of its neighbours.  Two shapes of the same code:
//   NQB = 2:  4 waves per workgroup, one per SIMD, 64 queries (two 32-row blocks A / B) per wave, the whole register file;
//   NQB = 1:  8 waves per workgroup, two per SIMD, 32 queries per wave, 256 registers each.
// Same arithmetic statement as attn_fwd_kernel (k_attn.hip; TF:qwen3:185-207 causal GQA hd 128, TF:esm:292-317 bidirectional hd 64):
//   S^T[key][q] = K[key][:] . Q[q][:]      (v_mfma_f32_32x32x16_bf16; a lane owns ONE query and 16 of the 32 keys of a step)
//   O^T[d][q]  += V^T[d][key] . P^T[key][q]  (the softmax registers ARE the B operand)
// What is different from the round 1-5 kernel, and why (NOTES.md rounds 2 - 5: that kernel spends 7000 cycles per 64-key tile on 2048
// cycles of matrix work — both waves of a SIMD run the same phase between barriers, and every wave re-reads the whole tile from LDS):
//   * software pipeline over 32-key steps j:   phase A: PV(j-1) MFMAs  ||  first half of step j's exponentials
//                                              phase B: QK(j+1) MFMAs  ||  the rest, row sums, bf16 packing
//     written as groups {1 MFMA + its share of the step's VALU work + LDS reads / a DMA piece} separated by sched_barrier(0): the
//     source order IS the schedule (the compiler does not interleave a wave's softmax with its own MFMAs by itself — round 2).
//     One wave issues about one instruction per 5 cycles whatever it is (tools/ubench/mfma_fill.hip: up to 5 plain VALU
//     instructions hide under a 33-cycle MFMA, every further one costs 5 cycles, a transcendental counts twice), so the
//     instruction COUNT of a step is its cost: no row maximum in the hot path, fragment addresses as immediates, no per-step waits;
//   * the exponentials are taken against the RUNNING maximum (speculatively); a wave-uniform rare path redoes the step with the
//     true maximum and rescales O only when a lane's row sum says some exponential may have exceeded 2^8 ("defer-max");
//   * K and V^T tiles arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) into two slots, issued one whole
//     tile ahead from inside phase A; one barrier per 64-key tile;
//   * the K-side MFMA rows are taken with bits 2 / 3 of the key index swapped, so that the 8 probabilities a lane packs for one
//     PV k-slot group are 8 CONSECUTIVE keys: the V^T fragment is one natural 16-byte unit of the transposed image
//     (no regrouping stores; the tile is copied by DMA as it lies in memory);
//   * NQB = 2: every K / V^T fragment read from LDS feeds two MFMAs (blocks A and B).
#include "bra_device.h"
#include "bra_api_internal.h"
#include "bra_attn.h"
#include "bra_attn4.h"

namespace bra {

constexpr float kSumMax = 256.0f;       // a lane's 16 exponentials of one row and step may sum to 2^8 against the running maximum before
                                        // the step is redone with the true maximum (so no single one exceeds 2^8)

template <int HD, int NQB>
struct T4 {
    static constexpr int NW = 8 / NQB;                // waves per workgroup (256 queries)
    static constexpr int CH = HD / 8;                 // 16-byte chunks per K row
    static constexpr int DS = HD / 16;                // contraction steps over d (QK^T) = K fragments of a step
    static constexpr int DB = HD / 32;                // 32-wide d blocks (PV)
    static constexpr int NFV = 2 * DB;                // V^T fragments of a step: 2 k-slot groups x DB
    static constexpr int RSH = HD == 128 ? 0 : (HD == 64 ? 1 : 2);   // rows per swizzle step = 16 / CH
    static constexpr int KBYTES = 64 * HD * 2;        // [64 keys][HD]
    static constexpr int TBYTES = HD * 128;           // [HD][64 keys]
    static constexpr int KPW = KBYTES / 1024 / NW;    // 1-KiB DMA pieces per wave and tile
    static constexpr int TPW = TBYTES / 1024 / NW;
    static constexpr int NDMA = KPW + TPW;
    static constexpr int NGA = NFV * NQB;             // MFMAs of PV(j-1)
    static constexpr int NGB = DS * NQB;              // MFMAs of QK(j+1)
    static constexpr int NG = NGA + NGB;
    static constexpr int FB = 2;                      // fragments per read block (a block = 2 NQB MFMA groups)
    static constexpr int NITEM = 7 * (8 * NQB + 2);   // item slots of one step's softmax (see sm_item: 56 NQB instructions + pipeline fill)
    static constexpr int QD = HD == 128 ? 2 : 5;      // items in a group that also issues a DMA piece (and computes its address)
    // first item of group G: the groups that carry a DMA piece take QD items each, the others share the rest evenly
    static constexpr int item0(int G, bool dma) {
        if (!dma) return (G * NITEM) / NG;
        return G <= NDMA ? G * QD : NDMA * QD + ((G - NDMA) * (NITEM - NDMA * QD)) / (NG - NDMA);
    }
};

// One step's softmax as single-instruction items dealt out to the step's MFMA groups (T4::item0).  s: the step's raw scores, p: its
// packed probabilities [block][k-slot group], st: scratch that lives across items.
// There is no row maximum here: the exponentials are taken against the RUNNING maximum, and the step is redone (step4's rare
// path) when a lane's row sum says that some exponential may have exceeded 2^8.
// A wave issues in order, and a dependent instruction waits for its producer (fma -> exp -> add: first build, three stalls per
// pair): the pairs are software-pipelined through slots of seven items — slot n holds the two arguments of pair n, the two
// exponentials of pair n - 1, the two sums and the pack of pair n - 2 — so that no item reads a result of its own slot, and the
// compiler may order a group's items as it likes.  Two sum chains per block for the same reason.
template <int NQB>
struct SmState {
    float x[2][2] = {{0.f, 0.f}, {0.f, 0.f}};      // arguments of the pair in flight, by pair parity
    float e[2][2];      // its exponentials
    float rs[NQB][2];   // this lane's row sums of the step (16 keys per block), even / odd key of a pair
};

template <int NQB, int K>
__device__ __forceinline__ void sm_item(const f32x16 (&s)[NQB], u32x4 (&p)[NQB][2], SmState<NQB>& st, const float (&m_run)[NQB], float sc) {
    constexpr int NP = 8 * NQB, n = K / 7, u = K % 7;
#ifdef BRA_A4_NOSM        // (timing probes, tools/build_attn4_variant.sh: the step without its softmax work — results are garbage)
    return;
#endif
#ifdef BRA_A4_KEEP        // (timing probes: bit 0 arguments, 1 exponentials, 2 sums, 3 packs; a dropped kind passes its input on)
    constexpr int keep = BRA_A4_KEEP;
    if constexpr (u < 2) {
        if constexpr (n < NP) { constexpr int i = n / NQB, qb = n % NQB;
            if constexpr ((keep & 17) == 17) { st.x[n & 1][u] = fmaf(st.x[n & 1][u], sc, -m_run[qb]); pin_f32(st.x[n & 1][u]); }      /* argument NOT from a score register */
            else if constexpr ((keep & 33) == 33) { st.x[n & 1][u] = s[qb][2 * i + u] * sc; pin_f32(st.x[n & 1][u]); }               /* two-operand multiply instead of the fma */
            else if constexpr (keep & 1) { st.x[n & 1][u] = fmaf(s[qb][2 * i + u], sc, -m_run[qb]); if constexpr (!(keep & 64)) pin_f32(st.x[n & 1][u]); } else st.x[n & 1][u] = s[qb][2 * i + u]; }
        return;
    } else if constexpr (u < 4) {
        if constexpr (n >= 1 && n <= NP) { if constexpr (keep & 2) { st.e[(n - 1) & 1][u - 2] = fast_exp2(st.x[(n - 1) & 1][u - 2]); pin_f32(st.e[(n - 1) & 1][u - 2]); } else st.e[(n - 1) & 1][u - 2] = st.x[(n - 1) & 1][u - 2]; }
        return;
    } else if constexpr (n >= 2 && n <= NP + 1) {
        constexpr int pr = n - 2, i = pr / NQB, qb = pr % NQB;
        if constexpr (u < 6) { if constexpr (keep & 4) st.rs[qb][u - 4] += st.e[pr & 1][u - 4]; return; }
        if constexpr (!(keep & 8)) { pin_f32(st.e[pr & 1][0]); pin_f32(st.e[pr & 1][1]); return; }
    }
#endif
    if constexpr (u < 2) {
        // (pin_f32: instruction selection sinks a pure instruction down to its first use — sched_barrier does not hold it — so every
        //  result is tied to its slot by an empty volatile asm)
        if constexpr (n < NP) { constexpr int i = n / NQB, qb = n % NQB; st.x[n & 1][u] = fmaf(s[qb][2 * i + u], sc, -m_run[qb]); pin_f32(st.x[n & 1][u]); }
    } else if constexpr (u < 4) {
        if constexpr (n >= 1 && n <= NP) { st.e[(n - 1) & 1][u - 2] = fast_exp2(st.x[(n - 1) & 1][u - 2]); pin_f32(st.e[(n - 1) & 1][u - 2]); }
    } else if constexpr (n >= 2 && n <= NP + 1) {
        constexpr int pr = n - 2, i = pr / NQB, qb = pr % NQB;
        if constexpr (u < 6) st.rs[qb][u - 4] += st.e[pr & 1][u - 4];
        else {
            uint32_t w2 = pack_bf2(st.e[pr & 1][0], st.e[pr & 1][1]);
            // the speculative results are only USED on the no-rescale path: without a pin the compiler sinks every exponential, sum
            // and pack of the step out of the MFMA groups into that successor block (seen in the first build's ISA)
            pin_u32_f32(w2, st.rs[qb][0], st.rs[qb][1]);
            constexpr int g = i >> 2, c4 = i & 3;     // registers 8 g .. 8 g + 7 = the 8 keys of k-slot group g
            if constexpr (c4 == 0) p[qb][g].x = w2; else if constexpr (c4 == 1) p[qb][g].y = w2;
            else if constexpr (c4 == 2) p[qb][g].z = w2; else p[qb][g].w = w2;
        }
    }
}
template <int NQB, int LO, int HI>
__device__ __forceinline__ void sm_items(const f32x16 (&s)[NQB], u32x4 (&p)[NQB][2], SmState<NQB>& st, const float (&m_run)[NQB], float sc) {
    if constexpr (LO < HI) {
        sm_item<NQB, LO>(s, p, st, m_run, sc);
        sm_items<NQB, LO + 1, HI>(s, p, st, m_run, sc);
    }
}

// everything a step needs that does not change inside the tile loop
template <int HD>
struct Ctx4 {
    unsigned kfo[HD / 16];        // LDS byte offset of this lane's K fragment of d-step ds (row pi(lane & 31) of a 32-key half)
    unsigned vfo[2][2];           // LDS byte offset of this lane's V^T fragment of (key half kb, k-slot group s2), d block 0
    float sc;
};
template <int HD>
__device__ __forceinline__ u32x4 read_vfrag(const Ctx4<HD>& cx, const char* vt, int kbv, int f) {       // fragment f = (k-slot group, d block)
#ifdef BRA_A4_NOLDS       // (timing probe: no fragment reads — garbage results)
    u32x4 z = {cx.vfo[kbv][0], cx.vfo[kbv][1], (unsigned)f, 1u}; return z;
#endif
    return ld16(vt + cx.vfo[kbv][f / (HD / 32)] + (f % (HD / 32)) * 4096);
}
template <int HD>
__device__ __forceinline__ u32x4 read_kfrag(const Ctx4<HD>& cx, const char* kt, int kbk, int ds) {
#ifdef BRA_A4_NOLDS
    u32x4 z = {cx.kfo[0], cx.kfo[1], (unsigned)ds, 1u}; return z;
#endif
    return ld16(kt + cx.kfo[ds] + kbk * (32 * HD * 2));
}

// the packed probabilities of one block from its 16 exponentials
__device__ __forceinline__ void pack_p(u32x4 (&p)[2], const float (&e)[16]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        p[g].x = pack_bf2(e[8 * g + 0], e[8 * g + 1]); p[g].y = pack_bf2(e[8 * g + 2], e[8 * g + 3]);
        p[g].z = pack_bf2(e[8 * g + 4], e[8 * g + 5]); p[g].w = pack_bf2(e[8 * g + 6], e[8 * g + 7]);
    }
}

// One pipeline step of the hot loop.  s_cur: scores of step j (complete); s_nxt: receives QK(j+1); p_prev: probabilities of step j-1
// (PV(j-1) accumulates them into o); p_cur: receives the probabilities of step j.
//   vt: V^T tile holding step j-1 (kbv = its key half);  kt: K tile holding step j+1 (kbk = its key half).
// DMA: this step's phase A also issues the wave's pieces of the workgroup's next tile pair (dma(i), i < NDMA).
// LDS fragment reads: the compiler waits for a fragment with s_waitcnt lgkmcnt(0), i.e. for EVERY read in flight.  So reads go out
// as whole blocks right behind a wait — in the first group of every block of four groups, after its MFMA — and are first used
// four groups later: a wait finds nothing younger than four MFMAs.  VPRE_IN: the first block's fragments were read by the
// previous step (vpre); VPRE_OUT: read the next step's (vt_nxt, kbv_nxt) during this step's last block.
template <int HD, int NQB, bool DMA, bool VPRE_IN, bool VPRE_OUT, typename DmaFn>
__device__ __forceinline__ void step4(const Ctx4<HD>& cx, const f32x16 (&s_cur)[NQB], f32x16 (&s_nxt)[NQB], const u32x4 (&p_prev)[NQB][2],
                                      u32x4 (&p_cur)[NQB][2], f32x16 (&o)[NQB][HD / 32], float (&m_run)[NQB], float (&l_run)[NQB],
                                      const u32x4 (&qf)[NQB][HD / 16], const char* vt, int kbv, const char* kt, int kbk,
                                      u32x4 (&vpre)[2], const char* vt_nxt, int kbv_nxt, DmaFn&& dma) {
    using T = T4<HD, NQB>;
    constexpr int FB = T::FB;
    SmState<NQB> st;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) { st.rs[qb][0] = 0.f; st.rs[qb][1] = 0.f; }
    u32x4 vfr[2][FB], kfr[2][FB];                     // [block parity][fragment of the block]
    // ---- phase A: PV(j-1) ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int f = FB - 1; f >= 0; --f) vfr[0][f] = VPRE_IN ? vpre[f] : read_vfrag<HD>(cx, vt, kbv, f);
#define BRA_A_GROUP(G)                                                                                                     \
    if constexpr ((G) < T::NGA) {                                                                                          \
        constexpr int f_ = (G) / NQB, qb_ = (G) % NQB, s2_ = f_ / T::DB, db_ = f_ % T::DB, blk_ = f_ / FB;                  \
        mfma_o(o[qb_][db_], vfr[blk_ & 1][f_ % FB], p_prev[qb_][s2_]);                                                     \
        if constexpr ((G) % (2 * NQB) == 0) {                                                                              \
            sched_fence();                                                                                                 \
            _Pragma("unroll") for (int u_ = FB - 1; u_ >= 0; --u_) {      /* (youngest first: one wait per block) */         \
                if constexpr ((G) + 2 * NQB < T::NGA) vfr[(blk_ + 1) & 1][u_] = read_vfrag<HD>(cx, vt, kbv, (blk_ + 1) * FB + u_); \
                else kfr[0][u_] = read_kfrag<HD>(cx, kt, kbk, u_);                                                         \
            }                                                                                                              \
        }                                                                                                                  \
        if constexpr (DMA && (G) < T::NDMA && !kNoDma) dma(G);                                                             \
        sm_items<NQB, T::item0(G, DMA), T::item0((G) + 1, DMA)>(s_cur, p_cur, st, m_run, cx.sc);                            \
        sched_fence();                                                                                                     \
    }
    // (macro-unrolled: a `for` over G would make the MFMA / read indices run-time values until the unroller has run, and the
    //  groups' order would no longer be the written one)
    BRA_A_GROUP(0) BRA_A_GROUP(1) BRA_A_GROUP(2) BRA_A_GROUP(3) BRA_A_GROUP(4) BRA_A_GROUP(5) BRA_A_GROUP(6) BRA_A_GROUP(7)
    BRA_A_GROUP(8) BRA_A_GROUP(9) BRA_A_GROUP(10) BRA_A_GROUP(11) BRA_A_GROUP(12) BRA_A_GROUP(13) BRA_A_GROUP(14) BRA_A_GROUP(15)
#undef BRA_A_GROUP
    // ---- phase B: QK(j+1) ------------------------------------------------------------------------------------------------
#define BRA_B_GROUP(G)                                                                                                     \
    if constexpr ((G) < T::NGB) {                                                                                          \
        constexpr int ds_ = (G) / NQB, qb_ = (G) % NQB, blk_ = ds_ / FB;                                                   \
        if constexpr (ds_ == 0) { f32x16 z_ = {}; s_nxt[qb_] = mfma_32x32x16(kfr[0][0], qf[qb_][0], z_); }                  \
        else s_nxt[qb_] = mfma_32x32x16(kfr[blk_ & 1][ds_ % FB], qf[qb_][ds_], s_nxt[qb_]);                               \
        if constexpr ((G) % (2 * NQB) == 0) {                                                                              \
            sched_fence();                                                                                                 \
            _Pragma("unroll") for (int u_ = FB - 1; u_ >= 0; --u_) {                                                        \
                if constexpr ((G) + 2 * NQB < T::NGB) kfr[(blk_ + 1) & 1][u_] = read_kfrag<HD>(cx, kt, kbk, (blk_ + 1) * FB + u_); \
                else if constexpr (VPRE_OUT) vpre[u_] = read_vfrag<HD>(cx, vt_nxt, kbv_nxt, u_);                           \
            }                                                                                                              \
        }                                                                                                                  \
        sm_items<NQB, T::item0(T::NGA + (G), DMA), T::item0(T::NGA + (G) + 1, DMA)>(s_cur, p_cur, st, m_run, cx.sc);       \
        sched_fence();                                                                                                     \
    }
    BRA_B_GROUP(0) BRA_B_GROUP(1) BRA_B_GROUP(2) BRA_B_GROUP(3) BRA_B_GROUP(4) BRA_B_GROUP(5) BRA_B_GROUP(6) BRA_B_GROUP(7)
    BRA_B_GROUP(8) BRA_B_GROUP(9) BRA_B_GROUP(10) BRA_B_GROUP(11) BRA_B_GROUP(12) BRA_B_GROUP(13) BRA_B_GROUP(14) BRA_B_GROUP(15)
#undef BRA_B_GROUP
    // ---- the rare path: a lane's 16 exponentials of a row sum to more than 2^8 (or to NaN / infinity: the first step of a
    //      block, whose running maximum is still -1e30) — some of them may exceed what a deferred rescale is allowed to leave.
    //      Redo the step against the true maximum and rescale O and l.  Wave-uniform.
    bool grow = false;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) { st.rs[qb][0] += st.rs[qb][1]; grow = grow || !(st.rs[qb][0] <= kSumMax); }
    if (wave_ballot(grow) != 0ull) {
        mfma_drain();
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            float mx = max3f(s_cur[qb][0], s_cur[qb][1], s_cur[qb][2]);
#pragma unroll
            for (int c = 1; c < 7; ++c) mx = max3f(mx, s_cur[qb][2 * c + 1], s_cur[qb][2 * c + 2]);
            mx = xhalf_max(max2f(mx, s_cur[qb][15]));
            const float m_new = fmaxf(m_run[qb], mx * cx.sc);
            const float alpha = fast_exp2(m_run[qb] - m_new);
            m_run[qb] = m_new;
            l_run[qb] *= alpha;
#pragma unroll
            for (int db = 0; db < T::DB; ++db) {      // (one accumulator block at a time: the AGPR <-> VGPR copies of all of O in one
                                                      //  scheduling region need a hundred temporaries, and the lane constants of the
                                                      //  whole kernel get spilled around it)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
                sched_fence();
            }
            float rs = 0.f;
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { e[r] = fast_exp2(fmaf(s_cur[qb][r], cx.sc, -m_new)); rs += e[r]; }
            st.rs[qb][0] = rs;
            pack_p(p_cur[qb], e);
        }
    }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) l_run[qb] += st.rs[qb][0];
}

// The same step without the interleave and with run-time switches: the first step of a block, and the steps at which a wave runs
// out of visible keys while its workgroup still has tiles to stage (causal masks: the waves of a workgroup end at different steps).
// Exact online softmax here (the running maximum follows every step) — both forms are the same sum, the hot loop only defers
// rescaling.  Never on the critical path of a long loop: a handful of these per wave and block.
template <int HD, int NQB>
__device__ __forceinline__ void cold_step4(const Ctx4<HD>& cx, f32x16 (&s_cur)[NQB], f32x16 (&s_nxt)[NQB], const u32x4 (&p_prev)[NQB][2],
                                           u32x4 (&p_cur)[NQB][2], f32x16 (&o)[NQB][HD / 32], float (&m_run)[NQB], float (&l_run)[NQB],
                                           const u32x4 (&qf)[NQB][HD / 16], const char* vt, int kbv, const char* kt, int kbk,
                                           bool do_pv, bool do_sm, bool do_qk) {
    using T = T4<HD, NQB>;
    if (do_pv) {
#pragma unroll
        for (int f = 0; f < T::NFV; ++f) {
            const u32x4 vf = read_vfrag<HD>(cx, vt, kbv, f);
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) mfma_o(o[qb][f % T::DB], vf, p_prev[qb][f / T::DB]);
        }
        mfma_drain();
    }
    if (do_sm) {
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            float mx = s_cur[qb][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s_cur[qb][r]);
            mx = xhalf_max(mx);
            const float m_new = fmaxf(m_run[qb], mx * cx.sc);
            const float alpha = fast_exp2(m_run[qb] - m_new);
            m_run[qb] = m_new;
            if (wave_ballot(alpha != 1.f) != 0ull) {
#pragma unroll
                for (int db = 0; db < T::DB; ++db) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
                    sched_fence();
                }
            }
            float rs = 0.f, e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { e[r] = fast_exp2(fmaf(s_cur[qb][r], cx.sc, -m_new)); rs += e[r]; }
            l_run[qb] = l_run[qb] * alpha + rs;
            pack_p(p_cur[qb], e);
        }
    }
    if (do_qk) {
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) {
            const u32x4 kf = read_kfrag<HD>(cx, kt, kbk, ds);
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                if (ds == 0) { f32x16 z = {}; s_nxt[qb] = mfma_32x32x16(kf, qf[qb][0], z); }
                else s_nxt[qb] = mfma_32x32x16(kf, qf[qb][ds], s_nxt[qb]);
            }
        }
    }
}

// scores of masked keys -> kMasked.  Register r of a lane in half h holds key 16 (r >> 3) + 8 h + (r & 7) of the step.
template <int NQB>
__device__ __forceinline__ void mask_scores4(f32x16 (&s)[NQB], uint32_t valid32, bool causal, int lim0, int h) {
    // lim0: (query of block A) + q_off - (first key of the step); block B's queries are 32 further on
    const uint32_t vb = valid32 >> (8 * h);
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
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

template <int HD, int NQB>
__global__ __launch_bounds__(512 / NQB) void attn_fwd4_kernel(AttnArgs a) {
    using T = T4<HD, NQB>;
    constexpr int SLOT = T::KBYTES + T::TBYTES, QW = 32 * NQB;
    BRA_DYN_SMEM(smem);                               // [2][K tile | V^T tile]
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6), h = lane >> 5, l31 = lane & 31;
    int bx_, hq, b;
    attn_block_coords(0, a.causal, bx_, hq, b);
    const int hkv = hq / (a.Hq / a.Hkv);
    // key range in `nsp` parts (grids that cannot fill the chip: one prompt, a 256-query completion segment): part sp of a query block
    // visits its tiles [tb, tb + ntile) and leaves the unnormalised O (fp32) and (reference maximum, sum) per query for attn_combine_kernel
    const int nsp = a.nsplit > 1 ? a.nsplit : 1;
    const int sp = nsp > 1 ? bx_ % nsp : 0;
    if (nsp > 1) bx_ /= nsp;
    const int q0 = bx_ * 256, qw0 = q0 + wave * QW;
    // (uniform: the DMA source is this scalar base + a 32-bit lane offset, `global_load_lds_dwordx4 v, s[..]`)
    const char* kb_ = uniform_ptr(a.k + b * a.k_sb + hkv * a.k_sh);
    const char* vtb = uniform_ptr(a.vt + b * a.vt_sb + hkv * a.vt_sh);

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
    int krow[T::KPW]; unsigned kcol[T::KPW], vsrc[T::TPW];     // (byte offsets; host-checked to fit 32 bits, strides 24)
#pragma unroll
    for (int i = 0; i < T::KPW; ++i) {
        const int u = 64 * (wave * T::KPW + i) + lane, row = u / T::CH, c = (u % T::CH) ^ ((row >> T::RSH) & (T::CH - 1));
        krow[i] = row; kcol[i] = (unsigned)(c * 16);
    }
#pragma unroll
    for (int i = 0; i < T::TPW; ++i) {
        const int u = 64 * (wave * T::TPW + i) + lane, d = u >> 3, c = (u & 7) ^ ((d >> 1) & 7);
        vsrc[i] = attn_mul24(d, (int)a.vt_sd * 2) + (unsigned)(c * 16);
    }
    // this workgroup's key tiles, this wave's steps
    int kv_end = a.Sk;
    if (a.causal) { const int last = q0 + 255 + a.q_off + 1; kv_end = last < kv_end ? last : kv_end; }
    const int ntile_all = kv_end > 0 ? (kv_end + 63) / 64 : 0;
    const int tb = nsp > 1 ? (ntile_all * sp) / nsp : 0;                               // first tile of this part; tile / step indices below
    const int ntile = (nsp > 1 ? (ntile_all * (sp + 1)) / nsp : ntile_all) - tb;       // are relative to it
    const int k_sbytes = (int)a.k_ss * 2;
    // The copies go through buffer descriptors of the (batch, kv-head) slices: lane offset (a loop constant) + scalar tile offset, no
    // address arithmetic per piece, and K rows beyond Sk are out of the descriptor's range — the hardware delivers zeros (their
    // scores are masked anyway) where a flat load would need a per-lane row clamp.
    const BufDesc kdesc = make_bufdesc(kb_, (unsigned)((a.Sk - 1) * k_sbytes + HD * 2));
    const BufDesc vdesc = make_bufdesc(vtb, (unsigned)((HD - 1) * (int)a.vt_sd * 2 + (int)a.vt_sd * 2));
    unsigned ksrc[T::KPW];
#pragma unroll
    for (int i = 0; i < T::KPW; ++i) ksrc[i] = attn_mul24(krow[i], k_sbytes) + kcol[i];
    auto dma_k = [&](int i, int tile, int slot) {
        dma16(kdesc, ksrc[i], (unsigned)((tb + tile) * 64) * (unsigned)k_sbytes, smem + slot * SLOT + (wave * T::KPW + i) * 1024);
    };
    auto dma_v = [&](int i, int tile, int slot) {
        dma16(vdesc, vsrc[i], (unsigned)((tb + tile) * 128), smem + slot * SLOT + T::KBYTES + (wave * T::TPW + i) * 1024);
    };

    // (a wave's own steps: under a causal mask the earlier waves of a workgroup finish up to six steps before the last one and then
    //  only take part in the staging; giving every wave the last wave's step count — fully masked steps, exact zeros — was
    //  measured and is slower: masking a step costs more than an interleaved step)
    int nstep_w = 0;
    if (qw0 < a.Sq && ntile > 0) {
        int lastq = qw0 + QW - 1; lastq = lastq < a.Sq ? lastq : a.Sq - 1;
        int lastk = a.causal ? lastq + a.q_off : a.Sk - 1;
        lastk = lastk < a.Sk ? lastk : a.Sk - 1;
        nstep_w = lastk >= 0 ? lastk / 32 + 1 - 2 * tb : 0;
        nstep_w = nstep_w > 0 ? nstep_w : 0;
        nstep_w = nstep_w < 2 * ntile ? nstep_w : 2 * ntile;
    }

    // Q fragments (B operand of S^T): row = query, 8 d per lane and d-step
    u32x4 qf[NQB][T::DS];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        int qr = qw0 + 32 * qb + l31;
        qr = qr < a.Sq ? qr : a.Sq - 1;
        const bf16_t* qp = a.q + b * a.q_sb + (long)qr * a.q_ss + hq * a.q_sh;
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) qf[qb][ds] = ld16(qp + ds * 16 + 8 * h);
    }
    // the Q fragments live in the accumulator half of the register file for the whole block (MFMA B operands may be AGPRs): the
    // architectural VGPRs are needed for two score sets, two probability sets and the fragments in flight
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) to_agpr(qf[qb][ds]);
    f32x16 o[NQB][T::DB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int i = 0; i < T::DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
    float m_run[NQB], l_run[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) { m_run[qb] = kNeg; l_run[qb] = 0.f; }

    // key validity of a tile as a 64-bit word (bit = key of the tile); requested two tiles ahead
    auto mask_byte = [&](int tile) -> int {
        int kj = (tb + tile) * 64 + lane;
        const bool in = kj < a.Sk;
        kj = in ? kj : a.Sk - 1;
        int v = a.kmask ? (int)a.kmask[(long)b * a.Sk + kj] : 1;
        return in ? v : 0;
    };

    f32x16 s0[NQB], s1[NQB];                          // scores of even / odd steps
    u32x4 p0[NQB][2], p1[NQB][2];                     // probabilities of even / odd steps
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
        int mb_pend = 0;
        auto noop = [](int) {};
        auto prep = [&](f32x16 (&s)[NQB], int j, uint64_t vword) {        // masks of step j, applied to its finished scores
            const int kv0s = 64 * tb + 32 * j;
            const uint32_t v32 = (uint32_t)(vword >> (32 * (j & 1)));
            const bool full = v32 == 0xffffffffu && (!a.causal || kv0s + 31 <= qw0 + a.q_off);
            if (!full) mask_scores4<NQB>(s, v32, a.causal != 0, qw0 + l31 + a.q_off - kv0s, opaque_i(lane) >> 5);
        };
        if (nstep_w > 0) {
            // QK(0) into s0 (the "next" scores of a step that does nothing else), then step 0 without a PV
            cold_step4<HD, NQB>(cx, s1, s0, p1, p0, o, m_run, l_run, qf, smem, 0, smem, 0, false, false, true);
            prep(s0, 0, vcur);
            cold_step4<HD, NQB>(cx, s0, s1, p1, p0, o, m_run, l_run, qf, smem, 0, smem, 1, false, true, nstep_w > 1);
        }
        // ---- iteration t: steps 2 t + 1 and 2 t + 2.  Hot loop: the iterations in which this wave runs both steps in full --------
        // (2 t + 2 < nstep_w: both steps of the iteration are this wave's.  The QK^T of step 2 t + 3 is issued whether the wave
        //  needs it or not — at worst it reads a slot that holds no tile yet, into registers nobody reads — so that the interleaved
        //  loop covers everything but the wave's last one or two steps)
        const int tmain = nstep_w >= 3 ? (nstep_w - 1) / 2 : 0;
        int t = 0;
        u32x4 vpre[T::FB];                            // the first V^T fragments of an iteration's second step, read by its first
#if defined(BRA_DEBUG) && !defined(BRA_EMU)
        // cycle stamps (s_memtime) of the hot loop of ONE workgroup (bra_attn_set_probe): per wave the cycles spent waiting for the
        // DMA, at the barrier, in the two steps (masking included) and in the loop's tail; slot 6 = the whole loop, 7 = steps
        const bool probing = a.probe != nullptr && bx_ == (int)gridDim.x / 2 && hq == 0 && b == 0;
        unsigned long long pa0 = 0, pa1 = 0, pa2 = 0, pa3 = 0, pa4 = 0, pt0 = 0;
        const unsigned long long pstart = __builtin_amdgcn_s_memtime();
        unsigned long long pmin2 = ~0ull, pmin3 = ~0ull, plast = 0;
#define BRA_STAMP(acc) if (probing) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); plast = n_ - pt0; acc += plast; pt0 = n_; }
#define BRA_STAMP_MIN(m) if (probing) { m = plast < m ? plast : m; }
        if (probing) pt0 = __builtin_amdgcn_s_memtime();
#else
#define BRA_STAMP(acc)
#define BRA_STAMP_MIN(m)
#endif
        // (two iterations per trip with the slot parity a compile-time constant: every LDS address of the steps is then a lane
        //  constant + an immediate — with a run-time slot each of the 16 fragment reads of a step needs its own v_add)
#define BRA_HOT_ITER(PAR)                                                                                                    \
        {                                                                                                                  \
            wait_vmcnt<0>();              /* everything issued one iteration ago (K(t + 1), V(t)) has landed ... */           \
            BRA_STAMP(pa0)                                                                                                 \
            raw_barrier();                /* ... for every wave, and every wave is done with K(t) and V(t - 1) */             \
            BRA_STAMP(pa1)                                                                                                 \
            mb_pend = mask_byte(t + 2);   /* (consumed at the end of this iteration: long landed by then) */                 \
            int tk = t + 2, tv = t + 1;   /* (beyond the last tile: a harmless re-load of the last one into the free slot) */ \
            tk = tk < ntile ? tk : ntile - 1;                                                                              \
            tv = tv < ntile ? tv : ntile - 1;                                                                              \
            auto dma = [&](int i) {                                                                                        \
                if (i < T::KPW) dma_k(i, tk, PAR); else dma_v(i - T::KPW, tv, 1 - (PAR));                                   \
            };                                                                                                             \
            const char* slot_t = smem + (PAR) * SLOT;             /* tile t */                                             \
            const char* slot_n = smem + (1 - (PAR)) * SLOT;       /* tile t + 1 */                                         \
            /* step 2 t + 1: PV(2 t) from V(t) half 0, softmax of S(2 t + 1), QK(2 t + 2) from K(t + 1) half 0 */           \
            prep(s1, 2 * t + 1, vcur);                                                                                     \
            step4<HD, NQB, true, false, true>(cx, s1, s0, p0, p1, o, m_run, l_run, qf, slot_t, 0, slot_n, 0, vpre, slot_t, 1, dma);    \
            BRA_STAMP(pa2) BRA_STAMP_MIN(pmin2)                                                                            \
            /* step 2 t + 2: PV(2 t + 1) from V(t) half 1, softmax of S(2 t + 2), QK(2 t + 3) from K(t + 1) half 1 */       \
            prep(s0, 2 * t + 2, vnext);                                                                                    \
            step4<HD, NQB, false, true, false>(cx, s0, s1, p1, p0, o, m_run, l_run, qf, slot_t, 1, slot_n, 1, vpre, slot_t, 1, noop);  \
            BRA_STAMP(pa3) BRA_STAMP_MIN(pmin3)                                                                            \
            vcur = vnext;                                                                                                  \
            vnext = wave_ballot(mb_pend != 0);                                                                             \
            BRA_STAMP(pa4)                                                                                                 \
            ++t;                                                                                                           \
        }
        while (t + 1 < tmain) { BRA_HOT_ITER(0) BRA_HOT_ITER(1) }
        if (t < tmain) BRA_HOT_ITER(0)
#undef BRA_HOT_ITER
#if defined(BRA_DEBUG) && !defined(BRA_EMU)
        if (probing && lane == 0) {
            unsigned long long* pp = a.probe + wave * 8;
            a.probe[64 + wave * 2] = pmin2; a.probe[64 + wave * 2 + 1] = pmin3;
            pp[0] = pa0; pp[1] = pa1; pp[2] = pa2; pp[3] = pa3; pp[4] = pa4; pp[5] = (unsigned long long)tmain;
            pp[6] = __builtin_amdgcn_s_memtime() - pstart; pp[7] = (unsigned long long)nstep_w;
        }
#endif
#undef BRA_STAMP
#undef BRA_STAMP_MIN
        // ---- the remaining iterations: this wave's last steps, then only its share of the staging -----------------------------
        for (; t < ntile; ++t) {
            wait_vmcnt<0>();
            raw_barrier();
            mb_pend = mask_byte(t + 2);
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
                cold_step4<HD, NQB>(cx, s1, s0, p0, p1, o, m_run, l_run, qf, slot_t, 0, slot_n, 0, true, j1 < nstep_w, j1 + 1 < nstep_w);
            }
            if (j2 <= nstep_w) {
                if (j2 < nstep_w) prep(s0, j2, vnext);
                cold_step4<HD, NQB>(cx, s0, s1, p1, p0, o, m_run, l_run, qf, slot_t, 1, slot_n, 1, true, j2 < nstep_w, j2 + 1 < nstep_w);
            }
            vcur = vnext;
            vnext = wave_ballot(mb_pend != 0);
        }
        wait_vmcnt<0>();                              // (the tail's redundant tile loads must not outlive the workgroup's LDS)
    }

    mfma_drain();
    if (nsp > 1) {
        // this part's unnormalised O and (reference maximum, sum) per query; lane (query, h) owns d = 32 db + 8 g + 4 h + 0..3
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const int qi = qw0 + 32 * qb + l31;
            const float l_tot = xhalf_sum(l_run[qb]);
            const long row = (((long)b * a.Hq + hq) * nsp + sp) * a.Sq + (qi < a.Sq ? qi : 0);
            float* op = a.part_o + row * HD;
#pragma unroll
            for (int db = 0; db < T::DB; ++db) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w = {o[qb][db][4 * g + 0], o[qb][db][4 * g + 1], o[qb][db][4 * g + 2], o[qb][db][4 * g + 3]};
                    if (qi < a.Sq) *reinterpret_cast<f32x4*>(op + db * 32 + 8 * g + 4 * h) = w;
                }
                sched_fence();
            }
            if (h == 0 && qi < a.Sq) { a.part_ml[row * 2] = m_run[qb]; a.part_ml[row * 2 + 1] = l_tot; }
        }
        return;
    }
    // ---- epilogue: normalise, bf16, whole 16-byte pieces of a row per store (lane pairs exchange their 8-byte halves) ---------------
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int qi = qw0 + 32 * qb + l31;
        const float l_tot = xhalf_sum(l_run[qb]);
        const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
        bf16_t* op = a.o + b * a.o_sb + (long)(qi < a.Sq ? qi : 0) * a.o_ss + hq * a.o_sh;
#pragma unroll
        for (int db = 0; db < T::DB; ++db) {
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
#ifdef BRA_A4_NOSTORE      // (timing probe: what the per-lane row-strided O stores of the block cost — never true at run time, garbage results)
                if (a.Sq < 0) st16(op + db * 32 + 8 * g + 8 * h, w);
#else
                if (qi < a.Sq) st16(op + db * 32 + 8 * g + 8 * h, w);
#endif
            }
            sched_fence();
        }
        if (a.lse && h == 0 && qi < a.Sq)
            a.lse[((long)b * a.Hq + hq) * a.Sq + qi] = l_tot > 0.f ? (m_run[qb] + log2f(l_tot)) * kLn2 : kNeg;
    }
}

// (the 8-wave shape NQB = 1 — two waves per SIMD, 32 queries each — compiles from the same template but is not instantiated: with
//  128 architectural registers per wave the hd 128 form spills to scratch, and neither head size was faster than NQB = 2 on an
//  MI355X: the two waves of a SIMD end up serialised on the matrix pipe; NOTES.md round 6)
template <int HD>
int launch_fwd4(const AttnArgs& a, bra_stream_t st) {
    const size_t smem = 2 * (T4<HD, 2>::KBYTES + T4<HD, 2>::TBYTES);
    BRA_ALLOW_SMEM((attn_fwd4_kernel<HD, 2>), smem);
    const int ns = a.nsplit > 1 ? a.nsplit : 1;      // (the merge of the parts: attn_combine_kernel, launched by the caller)
    BRA_LAUNCH((attn_fwd4_kernel<HD, 2>), dim3(((a.Sq + 255) / 256) * ns, a.Hq, a.B), dim3(256), smem, st, a);
    return BRA_LAUNCH_STATUS();
}
template int launch_fwd4<128>(const AttnArgs&, bra_stream_t);
template int launch_fwd4<64>(const AttnArgs&, bra_stream_t);

}  // namespace bra
