// bra_decattn.h — device bodies of the shared-prefix decode attention (see k_decattn.hip for the design notes): one wave per
// item / per (sequence, q-head) merge.  Shared by the launched kernels (XS = 0: every operand comes from an earlier launch) and
// by the persistent decode step of k_persist.hip (XS = 1: qkv, the partials and o cross workgroups INSIDE the launch and are
// moved with the sc1 accesses of bra_gridsync.h).  The arithmetic is the same instruction for instruction.
#pragma once
#include "bra_device.h"
#include "bra_gridsync.h"

namespace bra {


constexpr float kNegA = -1.0e30f;
constexpr float kLog2eA = 1.4426950408889634f;

struct DecOneArgs {
    const bf16_t* qkv; long ldqkv;          // [B, (Hq + 2 Hkv) * hd] raw projections of the new token, B = R * copies
    const bf16_t* qw; const bf16_t* kw;     // per-head RMSNorm weights [hd]
    const float* cosT; const float* sinT;   // [npos, hd/2]
    const int* pos;                         // [B] rotary position of the new token
    const float* rope_rows;                 // optional [B, hd]: cos | sin rows of the current positions
    const bf16_t* kp; long kp_sr, kp_sh, kp_ss;     // prompt K: element strides over (prompt, kv-head, position)
    const bf16_t* vtp; long vt_sr, vt_sh, vt_sd;    // prompt V^T [R, Hkv, hd, pitch]
    const uint8_t* pmask;                   // [R, P] validity of prompt positions (left padding) or null
    bf16_t* kc;                             // completion K cache   [B, Hkv, C, hd]
    bf16_t* vct; long cp;                   // completion V^T cache [B, Hkv, hd, cp]
    float* part_o; float* part_ml;          // [B * Hq, nslot, hd], [B * Hq, nslot, 2]: slots = prompt chunks, completion chunks, new key
    bf16_t* o; long ldo;                    // attention output [B, Hq * hd]
    int R, copies, Hq, Hkv, P, C, t, nslot, npc, ncc_grid;
    float eps, scale;
    const int* t_ptr;                       // optional device-side t (graph replay: constant launch arguments)
    float inv_Hq, inv_npc, inv_Hkv, inv_ncc, inv_copies;   // reciprocals for the item decomposition (da_div)
    // More than 16 query rows per (prompt, kv-head) — Qwen3-4B: 8 rollouts x G = 4 — are handled as `rsplit` VIRTUAL prompts per
    // prompt: R and copies above are the virtual counts (R = prompts x rsplit, copies = rollouts / rsplit, copies x G <= 16); virtual
    // prompt r reads the K / V^T / mask of prompt r / rsplit; sequence b = r x copies + copy is unchanged (rollouts are consecutive)
    int rsplit; float inv_rsplit;
};

// The launch is latency-bound (one wave per item, a few microseconds in all): every instruction between kernel entry and the
// last K / V^T request is on its critical path.  So: item decomposition by reciprocal multiplication instead of integer
// division (~25 scalar instructions each), 24-bit multiplies and 32-bit offsets instead of 64-bit address arithmetic
// (v_mul_lo_u32 / v_mad_u64_u32 run at quarter rate), no load under a branch (the compiler waits for it at the join — the
// padding-mask byte used to cost every prompt item a full memory round trip before its first K request).
// w / d for 0 <= w < 2^21, inv = 1 / d rounded to float: (w + 0.5) / d is at least 0.5 / d away from an integer, the float
// error is below (w / d) 2^-22
__device__ __forceinline__ int da_div(int w, float inv) { return (int)(((float)w + 0.5f) * inv); }
#ifdef BRA_EMU
__device__ __forceinline__ unsigned da_mul24(int a, int b) { return ((unsigned)a & 0xffffffu) * ((unsigned)b & 0xffffffu); }
#else
__device__ __forceinline__ unsigned da_mul24(int a, int b) { return __umul24((unsigned)a, (unsigned)b); }
#endif

// per-head RMSNorm (weight nw) + rotate-half RoPE of the 8-dim slice this lane owns (dims 8 dl .. 8 dl + 7); the HD / 8 lanes
// of a row are consecutive, the rotation partner (dims +- HD / 2) is lane ^ (HD / 16)
template <int HD>
__device__ __forceinline__ void nr_slice(float (&x)[8], const bf16_t* nw, const float* cosr, const float* sinr, int dl, float eps) {
    constexpr int LPK = HD / 8;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
#pragma unroll
    for (int mk = LPK >> 1; mk >= 1; mk >>= 1) ss += wave_shfl_xor(ss, mk);
    const float rstd = rsqrtf(ss / (float)HD + eps);
    float w[8];
    unpack8(ld16(nw + dl * 8), w);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = round_bf(w[i] * round_bf(x[i] * rstd));
    const int hsl = (dl & (LPK / 2 - 1)) * 8;
    const bool upper = dl >= LPK / 2;
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cosr + hsl), c1 = *reinterpret_cast<const f32x4*>(cosr + hsl + 4);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sinr + hsl), s1 = *reinterpret_cast<const f32x4*>(sinr + hsl + 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float other = wave_shfl_xor(x[i], LPK / 2);
        const float c = i < 4 ? c0[i & 3] : c1[i & 3], s = i < 4 ? s0[i & 3] : s1[i & 3];
        x[i] = round_bf(upper ? x[i] * c + other * s : x[i] * c - other * s);
    }
}

// One 64-key chunk against the <= 16 query rows of a kv-head group.  MFMA row i of 16-key block kb is key
//   rel(kb, i) = 32 (kb / 2) + 8 (i / 4) + 4 (kb % 2) + (i % 4)
// so that the eight contraction slots a lane feeds into the PV product (its four scores of block 2 kk, then of block 2 kk + 1)
// are eight CONSECUTIVE keys: the V^T fragment of a lane is one 16-byte load.
// the K rows and V^T rows of one 64-key chunk, straight into MFMA fragments (requests only: nothing is consumed here).  The
// persistent step issues them BEFORE the grid barrier that hands over q (they do not depend on the new token).
template <int HD>
__device__ __forceinline__ void item_kv_issue(const bf16_t* kbase, const int kss, const bf16_t* vbase, const int vsd, const int key0,
                                              const int nkeys, u32x4 (&kf)[4][HD / 32], u32x4 (&vf)[HD / 16][2]) {
    constexpr int DS = HD / 32, DB = HD / 16;
    const int lane = lane_id();
    const int fr = lane & 15, fq = lane >> 4;
    // ---- K rows straight into MFMA A fragments: lane (fr, fq) holds K[key rel(kb, fr)][32 s + 8 fq .. +8]
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        int key = key0 + 32 * (kb >> 1) + 8 * (fr >> 2) + 4 * (kb & 1) + (fr & 3);
        key = key < nkeys ? key : nkeys - 1;
        const bf16_t* kr = kbase + (da_mul24(key, kss) + (unsigned)(fq * 8));
#pragma unroll
        for (int s = 0; s < DS; ++s) kf[kb][s] = ld16(kr + s * 32);
    }
    // ---- V^T fragments: lane (fr = d within block, fq) holds keys key0 + 32 kk + 8 fq .. +8 of row d
    const unsigned vlane = da_mul24(fr, vsd) + (unsigned)(key0 + fq * 8);        // lane part; the 16-row block steps a scalar base
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        const bf16_t* vr = vbase + (long)db * 16 * vsd + vlane;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) vf[db][kk] = ld16(vr + kk * 32);
    }
}

// PRE = 1: kf / vf were requested by the caller (item_kv_issue) before the call
template <int HD, int G, int XS = 0, int PRE = 0>
__device__ __forceinline__ void one_item(const DecOneArgs& a, const bf16_t* kbase, const int kss, const bf16_t* vbase, const int vsd,
                                         const int key0, const int nkeys, const uint8_t* mask, const int r, const int hkv,
                                         const int row_lo, const int row_hi, const int slot,
                                         u32x4 (&kf)[4][HD / 32], u32x4 (&vf)[HD / 16][2]) {
    constexpr int DS = HD / 32;           // 32-deep contraction steps over the head dim
    constexpr int DB = HD / 16;           // 16-wide output blocks over the head dim
    const int lane = lane_id();
    const int fr = lane & 15, fq = lane >> 4;
    const int rows = a.copies * G;
    const int qrow = fr < rows ? fr : rows - 1;
    const int copy = (int)((unsigned)qrow / (unsigned)G), g = (int)((unsigned)qrow % (unsigned)G);
    const int b = r * a.copies + copy, hq = hkv * G + g;
    // ---- the query row first: its norm / rotate chain runs while the K and V^T requests are in flight
    const unsigned qoff = da_mul24(b, (int)a.ldqkv) + (unsigned)(hq * HD);         // (XS: qkv comes from other workgroups of this launch)
    u32x4 qraw[DS], qwv[DS];
#pragma unroll
    for (int s = 0; s < DS; ++s) qraw[s] = xld16<XS>(a.qkv, (qoff + (unsigned)(s * 32 + fq * 8)) * 2u);
#pragma unroll
    for (int s = 0; s < DS; ++s) qwv[s] = ld16(a.qw + s * 32 + fq * 8);
    const float* cosr; const float* sinr;
    if (a.rope_rows) { cosr = a.rope_rows + (unsigned)(b * HD); sinr = cosr + HD / 2; }
    else { const int p = a.pos[b]; cosr = a.cosT + (long)p * (HD / 2); sinr = a.sinT + (long)p * (HD / 2); }
    f32x4 cs[DS / 2][4];                  // [half-dim slice][cos lo, cos hi, sin lo, sin hi]
#pragma unroll
    for (int s = 0; s < DS / 2; ++s) {
        const int hb = s * 32 + fq * 8;
        cs[s][0] = *reinterpret_cast<const f32x4*>(cosr + hb); cs[s][1] = *reinterpret_cast<const f32x4*>(cosr + hb + 4);
        cs[s][2] = *reinterpret_cast<const f32x4*>(sinr + hb); cs[s][3] = *reinterpret_cast<const f32x4*>(sinr + hb + 4);
    }
    // validity byte of this lane's position: requested unconditionally (without a mask: some readable byte, ignored below)
    const bool nomask = mask == nullptr;
    uint8_t mb;
    {
        const int key = key0 + lane;
        const uint8_t* mp = nomask ? reinterpret_cast<const uint8_t*>(a.qw) : mask + (unsigned)(key < nkeys ? key : nkeys - 1);
        mb = *mp;
    }
    if constexpr (PRE == 0) item_kv_issue<HD>(kbase, kss, vbase, vsd, key0, nkeys, kf, vf);
    // ---- validity of the 64 positions of this chunk (bit = offset inside the chunk)
    uint64_t vbits;
    {
        const int key = key0 + lane;
        vbits = wave_ballot((key < nkeys) & (nomask | (mb != 0)));
    }
    sched_fence();
    // ---- q: RMSNorm, RoPE, scale
    float qv[DS][8];
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        unpack8(qraw[s], qv[s]);
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += qv[s][i] * qv[s][i];
    }
    ss += wave_shfl_xor(ss, 16);
    ss += wave_shfl_xor(ss, 32);
    const float rstd = rsqrtf(ss / (float)HD + a.eps);
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        float w[8];
        unpack8(qwv[s], w);
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[s][i] = round_bf(w[i] * round_bf(qv[s][i] * rstd));
    }
    const float sc = a.scale * kLog2eA;
    u32x4 qf[DS];
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        const bool upper = s >= DS / 2;
        const int sp = s ^ (DS / 2), sh = s & (DS / 2 - 1);
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float cv = i < 4 ? cs[sh][0][i & 3] : cs[sh][1][i & 3], sv = i < 4 ? cs[sh][2][i & 3] : cs[sh][3][i & 3];
            const float rot = upper ? qv[s][i] * cv + qv[sp][i] * sv : qv[s][i] * cv - qv[sp][i] * sv;
            o[i] = round_bf(rot) * sc;
        }
        qf[s] = pack8(o);
    }
    // ---- scores: D[MFMA row 4 fq + j][query row fr] per 16-key block
    f32x4 sreg[4];
    float m = kNegA;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < DS; ++s) acc = mfma_16x16x32(kf[kb][s], qf[s], acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rel_ = 32 * (kb >> 1) + 8 * fq + 4 * (kb & 1) + j;
            const bool ok = (vbits >> rel_) & 1ull;
            acc[j] = ok ? acc[j] : kNegA;
            m = fmaxf(m, acc[j]);
        }
        sreg[kb] = acc;
    }
    m = fmaxf(m, wave_shfl_xor(m, 16));
    m = fmaxf(m, wave_shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pe = sreg[kb][j] > 0.5f * kNegA ? fast_exp2(sreg[kb][j] - m) : 0.f;
            sreg[kb][j] = pe;
            l += pe;
        }
    l += wave_shfl_xor(l, 16);
    l += wave_shfl_xor(l, 32);
    // ---- O^T[d][row] = V^T . P
    const unsigned base = da_mul24((int)da_mul24(b, a.Hq) + hq, a.nslot) + (unsigned)slot;
    const bool live = fr < rows && fr >= row_lo && fr < row_hi;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 pf;
            pf.x = pack_bf2(sreg[2 * kk][0], sreg[2 * kk][1]);
            pf.y = pack_bf2(sreg[2 * kk][2], sreg[2 * kk][3]);
            pf.z = pack_bf2(sreg[2 * kk + 1][0], sreg[2 * kk + 1][1]);
            pf.w = pack_bf2(sreg[2 * kk + 1][2], sreg[2 * kk + 1][3]);
            acc = mfma_16x16x32(vf[db][kk], pf, acc);
        }
        if (live) xst16<XS>(a.part_o, (base * HD + (unsigned)(db * 16 + 4 * fq)) * 4u, __builtin_bit_cast(u32x4, acc));   // d = 16 db + 4 fq + j
    }
    if (live && fq == 0) {
        u32x2 mlv; mlv.x = __builtin_bit_cast(uint32_t, m); mlv.y = __builtin_bit_cast(uint32_t, l);
        xst8<XS>(a.part_ml, base * 8u, mlv);
    }
}

// The new token of (sequence b, q-head hq): q / k / v of the new position, cache append (one q-head per kv group does it), and
// the new key's partial in the slot behind the completion chunks: max = its score, sum = 1, O = v.
template <int HD, int G, int XS = 0>
__device__ __forceinline__ void one_newkey(const DecOneArgs& a, const int b, const int hq, const int t) {
    constexpr int LPK = HD / 8;           // lanes of one 8-dims-per-lane row
    const int lane = lane_id();
    const int grp = lane / LPK, dl = lane % LPK;
    const int hkv = hq / G;
    const int Nq = a.Hq * HD, Nkv = a.Hkv * HD;
    // lane groups of LPK lanes: group 1 = the k row, group 2 = the v row, every other group = the q row
    const int which = grp == 1 ? 1 : (grp == 2 ? 2 : 0);
    const unsigned src = da_mul24(b, (int)a.ldqkv) + (unsigned)(which == 0 ? hq * HD : (which == 1 ? Nq + hkv * HD : Nq + Nkv + hkv * HD));
    float x[8], y[8];
    unpack8(xld16<XS>(a.qkv, (src + (unsigned)(dl * 8)) * 2u), x);
    const float* cosr; const float* sinr;
    if (a.rope_rows) { cosr = a.rope_rows + (long)b * HD; sinr = cosr + HD / 2; }
    else { const int p = a.pos[b]; cosr = a.cosT + (long)p * (HD / 2); sinr = a.sinT + (long)p * (HD / 2); }
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = x[i];
    nr_slice<HD>(y, which == 1 ? a.kw : a.qw, cosr, sinr, dl, a.eps);
    // score of the new key (always attendable): q . k_new, both rounded to bf16 as the cached rows are
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) d += y[i] * wave_shfl_xor(y[i], LPK);          // group 0 <-> group 1
#pragma unroll
    for (int mk = LPK >> 1; mk >= 1; mk >>= 1) d += wave_shfl_xor(d, mk);
    const float s_new = wave_shfl(d, 0) * a.scale * kLog2eA;
    if (hq % G == 0) {
        if (grp == 1) st16(a.kc + (((long)b * a.Hkv + hkv) * a.C + t) * HD + dl * 8, pack8(y));
        if (grp == 2) {
            bf16_t* vp = a.vct + (((long)b * a.Hkv + hkv) * HD + dl * 8) * a.cp + t;
#pragma unroll
            for (int i = 0; i < 8; ++i) vp[(long)i * a.cp] = f2bf(x[i]);
        }
    }
    const unsigned base = (unsigned)((b * a.Hq + hq) * a.nslot + a.npc + (t + 63) / 64);
    if (grp == 2) {
        const unsigned ob = (base * HD + (unsigned)(dl * 8)) * 4u;
        xst16<XS>(a.part_o, ob, __builtin_bit_cast(u32x4, f32x4{x[0], x[1], x[2], x[3]}));
        xst16<XS>(a.part_o, ob + 16u, __builtin_bit_cast(u32x4, f32x4{x[4], x[5], x[6], x[7]}));
    }
    if (lane == 0) {
        u32x2 mlv; mlv.x = __builtin_bit_cast(uint32_t, s_new); mlv.y = __builtin_bit_cast(uint32_t, 1.f);
        xst8<XS>(a.part_ml, base * 8u, mlv);
    }
}

// item `w` of the step, in dispatch order: the new-key items first (their q / k / v -> norm -> rotate -> dot chain is the longest
// dependent chain), then the prompt chunks, then the completion chunks.  Everything in the descriptor is wave-uniform.
struct DecItem {
    int kind;                              // 0 nothing to do, 1 new key of (b, hq), 2 one 64-key chunk
    int b, hq;                             // kind 1
    const bf16_t* kbase; int kss; const bf16_t* vbase; int vsd; int key0, nkeys; const uint8_t* mask;      // kind 2
    int r, hkv, row_lo, row_hi, slot;
};

template <int HD, int G>
__device__ __forceinline__ void dec_item_decode(const DecOneArgs& a, int w, const int t, DecItem& d) {
    const int nK = a.R * a.copies * a.Hq;
    const int nP = a.npc * a.Hkv * a.R;
    d.kind = 0;
    if (w >= nK && w < nK + nP) {              // prompt chunk: does not depend on t
        w -= nK;
        const int q1 = da_div(w, a.inv_npc), c = w - q1 * a.npc;
        const int r = da_div(q1, a.inv_Hkv), hkv = q1 - r * a.Hkv;
        const int rp = a.rsplit > 1 ? da_div(r, a.inv_rsplit) : r;          // the prompt whose K / V^T this virtual prompt reads
        d.kind = 2;
        d.kbase = a.kp + rp * a.kp_sr + hkv * a.kp_sh; d.kss = (int)a.kp_ss;
        d.vbase = a.vtp + rp * a.vt_sr + hkv * a.vt_sh; d.vsd = (int)a.vt_sd;
        d.key0 = c * 64; d.nkeys = a.P; d.mask = a.pmask ? a.pmask + (long)rp * a.P : nullptr;
        d.r = r; d.hkv = hkv; d.row_lo = 0; d.row_hi = 16; d.slot = c;
        return;
    }
    if (w < nK) {
        d.kind = 1;
        d.b = da_div(w, a.inv_Hq);
        d.hq = w - d.b * a.Hq;
        return;
    }
    {
        const int v = w - nK - nP;
        const int q1 = da_div(v, a.inv_ncc), c = v - q1 * a.ncc_grid;
        const int q2 = da_div(q1, a.inv_copies), copy = q1 - q2 * a.copies;
        const int r = da_div(q2, a.inv_Hkv), hkv = q2 - r * a.Hkv;
        if (c * 64 >= t || r >= a.R) return;                     // (graph replay sizes the grid for the longest completion)
        const int b = r * a.copies + copy;
        d.kind = 2;
        d.kbase = a.kc + ((long)b * a.Hkv + hkv) * a.C * HD; d.kss = HD;
        d.vbase = a.vct + ((long)b * a.Hkv + hkv) * HD * a.cp; d.vsd = (int)a.cp;
        d.key0 = c * 64; d.nkeys = t; d.mask = nullptr;
        d.r = r; d.hkv = hkv; d.row_lo = copy * G; d.row_hi = copy * G + G; d.slot = a.npc + c;
    }
}

// PRE = 1: the chunk's K / V^T fragments were requested earlier (item_kv_issue on the same descriptor)
template <int HD, int G, int XS, int PRE>
__device__ __forceinline__ void dec_item_run(const DecOneArgs& a, const DecItem& d, const int t, u32x4 (&kf)[4][HD / 32], u32x4 (&vf)[HD / 16][2]) {
    if (d.kind == 2)
        one_item<HD, G, XS, PRE>(a, d.kbase, d.kss, d.vbase, d.vsd, d.key0, d.nkeys, d.mask, d.r, d.hkv, d.row_lo, d.row_hi, d.slot, kf, vf);
    else if (d.kind == 1)
        one_newkey<HD, G, XS>(a, d.b, d.hq, t);
}

template <int HD, int G, int XS = 0>
__device__ __forceinline__ void dec_attn_item(const DecOneArgs& a, int w) {
    u32x4 kf[4][HD / 32], vf[HD / 16][2];
    const int nK = a.R * a.copies * a.Hq, nP = a.npc * a.Hkv * a.R;
    // (a prompt chunk does not wait for the device-side step counter)
    const int t = (w >= nK && w < nK + nP) ? 0 : (a.t_ptr ? a.t_ptr[0] : a.t);
    DecItem d;
    dec_item_decode<HD, G>(a, w, t, d);
    dec_item_run<HD, G, XS, 0>(a, d, t, kf, vf);
}

// one wave per (sequence, q-head): slots [0, npc + ceil(t / 64)] -> o.  Lane (p = lane / LR, d4 = lane % LR) owns dims
// 4 d4 .. 4 d4 + 3 of partial rows p, p + PR, ...: PR rows per 16-byte-per-lane instruction, all requested up front.
#ifdef BRA_EMU
__device__ __forceinline__ void da_pin(uint32_t&) {}
#else
__device__ __forceinline__ void da_pin(uint32_t& v) { asm volatile("" : "+v"(v)); }
#endif

// all-lanes maximum / sum over the wave: the four in-row steps are DPP row rotations (VALU moves), only the two cross-row steps
// go through the LDS crossbar.  Every lane ends with the same bits (each step adds the same two partial sums in either order).
__device__ __forceinline__ float da_ror_f(float v, int n) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    const uint32_t r = n == 8 ? row_ror_u32<8>(u) : (n == 4 ? row_ror_u32<4>(u) : (n == 2 ? row_ror_u32<2>(u) : row_ror_u32<1>(u)));
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float da_wave_max(float v) {
    v = fmaxf(v, da_ror_f(v, 8)); v = fmaxf(v, da_ror_f(v, 4)); v = fmaxf(v, da_ror_f(v, 2)); v = fmaxf(v, da_ror_f(v, 1));
    v = fmaxf(v, wave_shfl_xor(v, 16));
    return fmaxf(v, wave_shfl_xor(v, 32));
}
__device__ __forceinline__ float da_wave_sum(float v) {
    v += da_ror_f(v, 8); v += da_ror_f(v, 4); v += da_ror_f(v, 2); v += da_ror_f(v, 1);
    v += wave_shfl_xor(v, 16);
    return v + wave_shfl_xor(v, 32);
}

struct DaNoHook { __device__ __forceinline__ void operator()() const {} };

// `after_issue` runs once every request of the merge is in flight (the persistent step issues its look-ahead weight requests
// there: loads return in order, so anything requested BEFORE the partial rows would have to land before the merge can start)
template <int HD, int XS = 0, class Hook = DaNoHook>
__device__ __forceinline__ void dec_attn_merge_one(const DecOneArgs& a, const int hq, const int b, const Hook& after_issue = Hook()) {
    constexpr int LR = HD / 4, PR = 64 / LR, PRE = 24;
    const int lane = lane_id();
    const int t = a.t_ptr ? a.t_ptr[0] : a.t;
    const int nsl = a.npc + (t + 63) / 64 + 1;
    const int p = lane / LR, d4 = lane % LR;
    const unsigned base = (unsigned)((b * a.Hq + hq) * a.nslot);
    // uniform row bases + 32-bit lane offsets (nslot <= 256 rows of HD floats); every request of the launch is issued before
    // anything is consumed (sched_fence: the compiler otherwise waits for the (max, sum) pairs, starts the reduction and only
    // then — one quarter-rate 64-bit multiply each — requests the partial rows: two memory round trips in series)
    u32x2 mlw[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int c = lane + 64 * q4;
        mlw[q4] = xld8<XS>(a.part_ml, (base * 2u + (unsigned)((c < nsl ? c : nsl - 1) * 2)) * 4u);
    }
    u32x4 v0[PRE];
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int c = u * PR + p;
        v0[u] = xld16<XS>(a.part_o, (base * HD + (unsigned)((c < nsl ? c : nsl - 1) * HD + d4 * 4)) * 4u);
    }
    // sched_fence is ordered behind the requests, but pure arithmetic is not ordered behind IT: instruction selection orders a
    // block bottom-up by register pressure and would still start the reduction above the fence, each partial-row request sunk
    // to its first use.  Passing the (max, sum) words through an (empty) volatile asm ties everything derived from them to a
    // point behind the fence; it waits for those four loads only (in-order return counter), not for the partial rows.
    sched_fence();
    after_issue();
    float mc[4], lc[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        uint32_t wm = mlw[q4].x, wl = mlw[q4].y;             // (a bit_cast applied to `w.y` directly reads element 0 with this clang)
        da_pin(wm); da_pin(wl);
        mc[q4] = __builtin_bit_cast(float, wm);
        lc[q4] = __builtin_bit_cast(float, wl);
    }
    float m = kNegA;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        if (lane + 64 * q4 >= nsl) { mc[q4] = kNegA; lc[q4] = 0.f; }
        m = fmaxf(m, mc[q4]);
    }
    m = da_wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) { mc[q4] = fast_exp2(mc[q4] - m); l += lc[q4] * mc[q4]; }      // mc now holds the slot weight
    l = da_wave_sum(l);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int c = u * PR + p;                    // PRE * PR <= 96 slots: weight registers 0 and 1
        const float wsel = (c >> 6) == 0 ? mc[0] : mc[1];
        float w = wave_shfl(wsel, c & 63);
        w = c < nsl ? w : 0.f;
        const f32x4 ov = __builtin_bit_cast(f32x4, v0[u]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += w * ov[j];
    }
    for (int c0 = PRE * PR; c0 < nsl; c0 += 8 * PR) {
        u32x4 v[8];
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u * PR + p;
            const int cc = c < nsl ? c : nsl - 1;
            v[u] = xld16<XS>(a.part_o, ((base + (unsigned)cc) * HD + (unsigned)(d4 * 4)) * 4u);
            const float wsel = (cc >> 6) == 0 ? mc[0] : ((cc >> 6) == 1 ? mc[1] : ((cc >> 6) == 2 ? mc[2] : mc[3]));
            w[u] = wave_shfl(wsel, cc & 63);
            w[u] = c < nsl ? w[u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 ov = __builtin_bit_cast(f32x4, v[u]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += w[u] * ov[j];
        }
    }
#pragma unroll
    for (int mk = LR; mk < 64; mk <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += wave_shfl_xor(acc[j], mk);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (p == 0) {
        u32x2 ov;
        ov.x = pack_bf2(acc[0] * inv, acc[1] * inv);
        ov.y = pack_bf2(acc[2] * inv, acc[3] * inv);
        xst8<XS>(a.o, (da_mul24(b, (int)a.ldo) + (unsigned)(hq * HD + d4 * 4)) * 2u, ov);
    }
}


}  // namespace bra
