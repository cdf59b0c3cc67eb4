// k_attn.hip — flash-style attention for gfx950, forward and backward.
//   * bidirectional with key-padding mask, hd = 64      (NT-v2 encoder, TF:esm:292-317)
//   * causal GQA with key-padding mask, hd = 128        (Qwen3, TF:qwen3:185-207; fp32 softmax :202)
// Layout of the computation (v_mfma_f32_32x32x16_bf16, wave64):
//   S^T[key][q] = K[key][:] . Q[q][:]          -> every lane owns ONE query (lane & 31) and 16 keys
//                                                 per 32-key block; its partner lane ^ 32 owns the
//                                                 other 16, so row max / row sum are in-lane
//                                                 reductions plus one exchange with lane ^ 32.
//   O^T[d][q]   = V^T[d][key] . P^T[key][q]    -> the softmax registers ARE the MFMA B operand
//                                                 (keys on k-slots, query on lanes): no cross-lane
//                                                 movement between the two GEMMs; the running
//                                                 max / sum / rescale are per-lane scalars.
// Every contraction is "K-contiguous on both operands", so the producer kernels hand over
// V^T (forward), K^T (dQ) and Q^T / dO^T (dK,dV) as [B, H, hd, S_pad] images (transpose kernels in
// k_misc.hip); no transposing LDS reads are needed.
// K / V^T tiles are staged through LDS (register-prefetched, double-buffered, one barrier per
// 64-key tile) with a 16-byte-chunk XOR swizzle that makes the ds_read_b128 fragment reads
// conflict-free.
#include "bra_device.h"
#include "bra_api_internal.h"
#include "bra_attn.h"

namespace bra {

// load a [64][HD] row-major tile (rows r0.., clamped to nrows-1) into registers / LDS
template <int HD, int NT>
__device__ __forceinline__ void load_rows(u32x4 (&r)[(64 * (HD / 8)) / NT], const bf16_t* base, long row_stride,
                                          int r0, int nrows, int tid) {
    constexpr int CH = HD / 8;
    // round 5: `base` is uniform (one (batch, head) slice), the lane part is a 32-bit ELEMENT offset built with one full-rate 24-bit
    // multiply — the 64-bit `row * stride` of rounds 1-4 cost two quarter-rate multiplies and a 64-bit add per request, 60 of the
    // ~380 VALU instructions of a forward tile in a kernel that is VALU-bound (NOTES.md); the host checks that the slice fits
    // (attn_check: strides < 2^24, slice < 2^31 elements)
#pragma unroll
    for (int i = 0; i < (64 * CH) / NT; ++i) {
        int q = tid + NT * i;
        int row = q / CH, c = q % CH;
        int rr = r0 + row;
        rr = rr < nrows ? rr : nrows - 1;
        r[i] = ld16(base + (attn_mul24(rr, (int)row_stride) + (unsigned)(c * 8)));
    }
}
template <int HD, int NT>
__device__ __forceinline__ void store_rows(char* lds, const u32x4 (&r)[(64 * (HD / 8)) / NT], int tid) {
    constexpr int CH = HD / 8;
#pragma unroll
    for (int i = 0; i < (64 * CH) / NT; ++i) {
        int q = tid + NT * i;
        st16(lds + Tile<HD>::koff(q / CH, q % CH), r[i]);
    }
}
// load a [HD][64] tile of a transposed image (rows = d, 64 consecutive sequence positions from s0)
template <int HD, int NT>
__device__ __forceinline__ void load_trans(u32x4 (&r)[(HD * 8) / NT], const bf16_t* base, long d_stride, int s0,
                                           int tid) {
#pragma unroll
    for (int i = 0; i < (HD * 8) / NT; ++i) {
        int q = tid + NT * i;
        int d = q >> 3, c = q & 7;
        r[i] = ld16(base + (attn_mul24(d, (int)d_stride) + (unsigned)(s0 + c * 8)));
    }
}
template <int HD, int NT>
__device__ __forceinline__ void store_trans(char* lds, const u32x4 (&r)[(HD * 8) / NT], int tid) {
#pragma unroll
    for (int i = 0; i < (HD * 8) / NT; ++i) {
        int q = tid + NT * i;
#ifdef BRA_TRANS_B64
        // (rounds 1-4: the global chunk as it is; the reader then assembles its fragment from two 8-byte pieces 16 bytes apart)
        st16(lds + Tile<HD>::toff(q >> 3, q & 7), r[i]);
#else
        // round 5: positions regrouped per 16 as [0..3, 8..11 | 4..7, 12..15] — exactly the two 8-position fragments the 32x32 MFMA
        // register order asks for (frag_trans), each now ONE 16-byte unit: chunk c = 2 g + e (e: first / second half of the group)
        // puts its low half into unit 2 g at byte 8 e and its high half into unit 2 g + 1 at byte 8 e.  One ds_read_b128 per fragment
        // instead of two strided ds_read_b64 and the v_mov that glued them (48 per 64-key tile in the forward, 250 - 450 per tile loop in
        // the backward kernels); the staging side pays two 8-byte writes per chunk instead of one 16-byte write.
        {
            const int d = q >> 3, c = q & 7, g = c >> 1, e = c & 1;
            u32x2 lo, hi;
            lo.x = r[i].x; lo.y = r[i].y; hi.x = r[i].z; hi.y = r[i].w;
            st8(lds + Tile<HD>::toff(d, 2 * g) + 8 * e, lo);
            st8(lds + Tile<HD>::toff(d, 2 * g + 1) + 8 * e, hi);
        }
#endif
    }
}

// fragment readers -----------------------------------------------------------
// operand with rows on lanes (row = rbase + (lane & 31)) and 8 contiguous d at 16*ds + 8*(lane >> 5)
template <int HD>
__device__ __forceinline__ u32x4 frag_rows(const char* lds, int rbase, int ds, int lane) {
    return ld16(lds + Tile<HD>::koff(rbase + (lane & 31), ds * 2 + (lane >> 5)));
}
// operand from a transposed tile: row d = dbase + (lane & 31); sequence positions matching the
// register order of a 32x32 C/D fragment: {16s + 4h + 0..3, 16s + 4h + 8 + 0..3}, h = lane >> 5
template <int HD>
__device__ __forceinline__ u32x4 frag_trans(const char* lds, int dbase, int s, int lane) {
    const int d = dbase + (lane & 31), h = lane >> 5;
#ifdef BRA_TRANS_B64
    u32x2 p0 = ld8(lds + Tile<HD>::toff(d, 2 * s) + 8 * h);
    u32x2 p1 = ld8(lds + Tile<HD>::toff(d, 2 * s + 1) + 8 * h);
    u32x4 o; o.x = p0.x; o.y = p0.y; o.z = p1.x; o.w = p1.y;
    return o;
#else
    return ld16(lds + Tile<HD>::toff(d, 2 * s + h));          // [16 s + 4 h + 0..3 | 16 s + 8 + 4 h + 0..3] (store_trans)
#endif
}
// ---------------------------------------------------------------------------
// forward
// NW waves per workgroup (32 queries each); 8 waves = 2 per SIMD share one staged K / V^T tile
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(AttnArgs a) {
    using T = Tile<HD>;
    constexpr int NT = NW * 64, QROWS = NW * 32;
    BRA_DYN_SMEM(smem);   // [2][K tile | V^T tile]
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    int bx_, hq, b;
    attn_block_coords(a.legacy_order, a.causal, bx_, hq, b);
    const int hkv = hq / (a.Hq / a.Hkv);
    const int nsp = a.nsplit > 1 ? a.nsplit : 1;
    const int sp = nsp > 1 ? bx_ % nsp : 0;           // (grid x = query blocks x key parts: the parts of a block are neighbours)
    if (nsp > 1) bx_ /= nsp;
    const int q0 = bx_ * QROWS;
    const int qw0 = q0 + wave * 32;
    const int qi = qw0 + (lane & 31);                 // this lane's query
    const bf16_t* kb_ = a.k + b * a.k_sb + hkv * a.k_sh;
    const bf16_t* vtb = a.vt + b * a.vt_sb + hkv * a.vt_sh;

    // Q fragments (B operand of S^T): row = query, 8 d per lane per step
    u32x4 qf[T::DS];
    {
        int qr = qi < a.Sq ? qi : a.Sq - 1;
        const bf16_t* qp = a.q + b * a.q_sb + (long)qr * a.q_ss + hq * a.q_sh;
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) qf[ds] = ld16(qp + ds * 16 + 8 * h);
    }
    f32x16 o[T::DB];
#pragma unroll
    for (int i = 0; i < T::DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = kNeg, l_run = 0.f;
    const float sc = a.scale * kLog2e;

    int kv_end = a.Sk;
    if (a.causal) {
        int last = q0 + QROWS - 1 + a.q_off + 1;
        kv_end = last < kv_end ? last : kv_end;
    }
    const int ntile_all = kv_end > 0 ? (kv_end + 63) / 64 : 0;
    const int t_first = nsp > 1 ? (ntile_all * sp) / nsp : 0;
    const int ntile = nsp > 1 ? (ntile_all * (sp + 1)) / nsp : ntile_all;      // this part's tiles: [t_first, ntile)

    u32x4 rk[(64 * T::CH) / NT], rv[(HD * 8) / NT];
    if (ntile > t_first) {
        load_rows<HD, NT>(rk, kb_, a.k_ss, t_first * 64, a.Sk, tid);
        load_trans<HD, NT>(rv, vtb, a.vt_sd, t_first * 64, tid);
        store_rows<HD, NT>(smem + (t_first & 1) * (T::KBYTES + T::TBYTES), rk, tid);
        store_trans<HD, NT>(smem + (t_first & 1) * (T::KBYTES + T::TBYTES) + T::KBYTES, rv, tid);
    }
    __syncthreads();

    // key-validity word of the NEXT tile is requested one iteration ahead (its mask load is a dependent L2 round trip)
    uint64_t valid_next = ntile > t_first ? key_valid_word(a, b, t_first * 64, lane) : 0ull;
    for (int t = t_first; t < ntile; ++t) {
        const int kv0 = t * 64;
        const int tl = opaque_i(tid), ll = opaque_i(lane);      // per-iteration copies: keeps address math out of registers across the loop
        const char* sk = smem + (t & 1) * (T::KBYTES + T::TBYTES);
        const char* sv = sk + T::KBYTES;
        const bool more = t + 1 < ntile;
        if (more) {
            load_rows<HD, NT>(rk, kb_, a.k_ss, kv0 + 64, a.Sk, tl);
            load_trans<HD, NT>(rv, vtb, a.vt_sd, kv0 + 64, tl);
        }
        const uint64_t valid = valid_next;
        valid_next = key_valid_word(a, b, kv0 + 64, ll);
        // wave-uniform skip: every key of this tile is after every query of this wave
        const bool skip = a.causal && (kv0 > qw0 + 31 + a.q_off);
        if (!skip) {
            f32x16 st[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
                for (int ds = 0; ds < T::DS; ++ds)
                    st[kb] = mfma_32x32x16(frag_rows<HD>(sk, kb * 32, ds, ll), qf[ds], st[kb]);
                sched_fence();            // bounds how many fragment reads the scheduler keeps in flight (registers)
            }
            // tile max, probabilities, running statistics.  `full` (wave-uniform): every key of the tile is valid and
            // visible to every query of this wave, so the per-element mask logic (most of the VALU work of a masked
            // tile) is skipped; only diagonal tiles and tiles holding padded keys take the masked path.
            const bool full = valid == ~0ull && (!a.causal || kv0 + 63 <= qw0 + a.q_off);
            float mx = kNeg;
            if (full) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
                mx *= sc;                                   // sc > 0: max(s) * sc == max(s * sc)
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kl = kb * 32 + crow(r, h);
                        bool ok = (valid >> kl) & 1ull;
                        if (a.causal) ok = ok && (kv0 + kl <= qi + a.q_off);
                        float s = ok ? st[kb][r] * sc : kNeg;
                        st[kb][r] = s;
                        mx = fmaxf(mx, s);
                    }
            }
            mx = fmaxf(mx, wave_shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = fast_exp2(m_run - m_new);
            float rs = 0.f;
            if (full) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = fast_exp2(fmaf(st[kb][r], sc, -m_new));
                        st[kb][r] = p;
                        rs += p;
                    }
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float p = st[kb][r] > 0.5f * kNeg ? fast_exp2(st[kb][r] - m_new) : 0.f;
                        st[kb][r] = p;
                        rs += p;
                    }
            }
            rs += wave_shfl_xor(rs, 32);
            l_run = l_run * alpha + rs;
            m_run = m_new;
            if (wave_ballot(alpha != 1.f) != 0ull) {        // the running max moved for some query of this wave
#pragma unroll
                for (int i = 0; i < T::DB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            }
            // O^T += V^T . P^T
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kb = s >> 1, r0 = 8 * (s & 1);
                u32x4 pf;
                pf.x = pack_bf2(st[kb][r0 + 0], st[kb][r0 + 1]);
                pf.y = pack_bf2(st[kb][r0 + 2], st[kb][r0 + 3]);
                pf.z = pack_bf2(st[kb][r0 + 4], st[kb][r0 + 5]);
                pf.w = pack_bf2(st[kb][r0 + 6], st[kb][r0 + 7]);
#pragma unroll
                for (int db = 0; db < T::DB; ++db)
                    o[db] = mfma_32x32x16(frag_trans<HD>(sv, db * 32, s, ll), pf, o[db]);
                sched_fence();
            }
        }
        if (more) {
            char* nk = smem + ((t + 1) & 1) * (T::KBYTES + T::TBYTES);
            store_rows<HD, NT>(nk, rk, tl);
            store_trans<HD, NT>(nk + T::KBYTES, rv, tl);
        }
        __syncthreads();
    }

    if (nsp > 1) {
        // this part's unnormalised O and (max, sum) per query; lane (query, h) owns d = 32 db + 8 g + 4 h + 0..3
        if (qi < a.Sq) {
            const long row = (((long)b * a.Hq + hq) * nsp + sp) * a.Sq + qi;
            float* op = a.part_o + row * HD;
#pragma unroll
            for (int db = 0; db < T::DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w = {o[db][4 * g + 0], o[db][4 * g + 1], o[db][4 * g + 2], o[db][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(op + db * 32 + 8 * g + 4 * h) = w;
                }
            if (h == 0) { a.part_ml[row * 2] = m_run; a.part_ml[row * 2 + 1] = l_run; }
        }
        return;
    }
    if (qi < a.Sq) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        bf16_t* op = a.o + b * a.o_sb + (long)qi * a.o_ss + hq * a.o_sh;
#pragma unroll
        for (int db = 0; db < T::DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w.x = pack_bf2(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv);
                w.y = pack_bf2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
                st8(op + db * 32 + 8 * g + 4 * h, w);
            }
        if (a.lse && h == 0)
            a.lse[((long)b * a.Hq + hq) * a.Sq + qi] = l_run > 0.f ? (m_run + log2f(l_run)) * kLn2 : kNeg;
    }
}

// merges the key parts of a split forward (fixed order: deterministic): one lane per (query, 4 dims)
template <int HD>
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnArgs a) {
    constexpr int LR = HD / 4;                        // lanes per query row
    const long row = (long)blockIdx.x * (256 / LR) + (int)threadIdx.x / LR;     // (b, hq, qi) flattened
    const int d4 = (int)threadIdx.x % LR;
    const long nrow = (long)a.B * a.Hq * a.Sq;
    if (row >= nrow) return;
    const int qi = (int)(row % a.Sq);
    const long bh = row / a.Sq;
    const int hq = (int)(bh % a.Hq), b = (int)(bh / a.Hq);
    const int ns = a.nsplit;
    float m = kNeg;
    for (int s_ = 0; s_ < ns; ++s_) m = fmaxf(m, a.part_ml[((bh * ns + s_) * a.Sq + qi) * 2]);
    float l = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s_ = 0; s_ < ns; ++s_) {
        const long pr = (bh * ns + s_) * a.Sq + qi;
        const float ms = a.part_ml[pr * 2], ls = a.part_ml[pr * 2 + 1];
        const float w = ls > 0.f ? fast_exp2(ms - m) : 0.f;
        l += ls * w;
        const f32x4 ov = *reinterpret_cast<const f32x4*>(a.part_o + pr * HD + d4 * 4);
        acc += ov * w;
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
    u32x2 w2;
    w2.x = pack_bf2(acc[0] * inv, acc[1] * inv);
    w2.y = pack_bf2(acc[2] * inv, acc[3] * inv);
    st8(a.o + b * a.o_sb + (long)qi * a.o_ss + hq * a.o_sh + d4 * 4, w2);
    if (a.lse && d4 == 0) a.lse[bh * a.Sq + qi] = l > 0.f ? (m + log2f(l)) * kLn2 : kNeg;
}

// out[b, s, h, :] = sum over the parts of part[b, h, part, s, :] (fp32, in part order) rounded to bf16: the split backward kernels
template <int HD>
__global__ __launch_bounds__(256) void attn_sum_parts_kernel(const float* part, int ns, int B, int H, int S, bf16_t* out, long o_sb, long o_ss,
                                                             long o_sh) {
    constexpr int LR = HD / 4;
    const long row = (long)blockIdx.x * (256 / LR) + (int)threadIdx.x / LR;     // (b, h, s) flattened
    const int d4 = (int)threadIdx.x % LR;
    if (row >= (long)B * H * S) return;
    const int si = (int)(row % S);
    const long bh = row / S;
    const int hh = (int)(bh % H), b = (int)(bh / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s_ = 0; s_ < ns; ++s_) acc += *reinterpret_cast<const f32x4*>(part + ((bh * ns + s_) * S + si) * HD + d4 * 4);
    u32x2 w2;
    w2.x = pack_bf2(acc[0], acc[1]);
    w2.y = pack_bf2(acc[2], acc[3]);
    st8(out + b * o_sb + (long)si * o_ss + hh * o_sh + d4 * 4, w2);
}

// ---------------------------------------------------------------------------
// backward, dQ:   one workgroup = 128 queries of one (batch, q-head); loops over key tiles.
//   S^T = K Q^T, dP^T = V dO^T (both [key][q], query on lanes), dS^T = P^T * (dP^T - delta[q]) * scale
//   dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(AttnArgs a) {
    using T = Tile<HD>;
    constexpr int NT = NW * 64, QROWS = NW * 32;
    BRA_DYN_SMEM(smem);   // [2][K tile | V tile | K^T tile]
    constexpr int STAGE = 2 * T::KBYTES + T::TBYTES;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    int bx_, hq, b;
    attn_block_coords(a.legacy_order, a.causal, bx_, hq, b);
    const int hkv = hq / (a.Hq / a.Hkv);
    const int nsp = a.nsplit > 1 ? a.nsplit : 1;
    const int sp = nsp > 1 ? bx_ % nsp : 0;
    if (nsp > 1) bx_ /= nsp;
    const int q0 = bx_ * QROWS;
    const int qw0 = q0 + wave * 32;
    const int qi = qw0 + (lane & 31);
    const int qr = qi < a.Sq ? qi : a.Sq - 1;
    const bf16_t* kb_ = a.k + b * a.k_sb + hkv * a.k_sh;
    const bf16_t* vb_ = a.v + b * a.v_sb + hkv * a.v_sh;
    const bf16_t* ktb = a.kt + b * a.kt_sb + hkv * a.kt_sh;

    u32x4 qf[T::DS], dof[T::DS];
    {
        const bf16_t* qp = a.q + b * a.q_sb + (long)qr * a.q_ss + hq * a.q_sh;
        const bf16_t* dp = a.dout + b * a.do_sb + (long)qr * a.do_ss + hq * a.do_sh;
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) { qf[ds] = ld16(qp + ds * 16 + 8 * h); dof[ds] = ld16(dp + ds * 16 + 8 * h); }
    }
    const long lidx = ((long)b * a.Hq + hq) * a.Sq + qr;
    const float lse2 = a.lse[lidx] * kLog2e;
    const float dlt = a.delta[lidx];
    const float sc = a.scale * kLog2e;

    f32x16 dq[T::DB];
#pragma unroll
    for (int i = 0; i < T::DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

    int kv_end = a.Sk;
    if (a.causal) { int last = q0 + QROWS - 1 + a.q_off + 1; kv_end = last < kv_end ? last : kv_end; }
    const int ntile_all = kv_end > 0 ? (kv_end + 63) / 64 : 0;
    const int t_first = nsp > 1 ? (ntile_all * sp) / nsp : 0;
    const int ntile = nsp > 1 ? (ntile_all * (sp + 1)) / nsp : ntile_all;      // this part's key tiles: [t_first, ntile)

    u32x4 rk[(64 * T::CH) / NT], rv[(64 * T::CH) / NT], rt[(HD * 8) / NT];
    if (ntile > t_first) {
        char* s0_ = smem + (t_first & 1) * STAGE;
        load_rows<HD, NT>(rk, kb_, a.k_ss, t_first * 64, a.Sk, tid);
        load_rows<HD, NT>(rv, vb_, a.v_ss, t_first * 64, a.Sk, tid);
        load_trans<HD, NT>(rt, ktb, a.kt_sd, t_first * 64, tid);
        store_rows<HD, NT>(s0_, rk, tid);
        store_rows<HD, NT>(s0_ + T::KBYTES, rv, tid);
        store_trans<HD, NT>(s0_ + 2 * T::KBYTES, rt, tid);
    }
    __syncthreads();

    for (int t = t_first; t < ntile; ++t) {
        const int kv0 = t * 64;
        const int tl = opaque_i(tid), ll = opaque_i(lane);      // see forward
        const int hl = ll >> 5;
        const char* sk = smem + (t & 1) * STAGE;
        const char* sv = sk + T::KBYTES;
        const char* skt = sk + 2 * T::KBYTES;
        const bool more = t + 1 < ntile;
        if (more) {
            load_rows<HD, NT>(rk, kb_, a.k_ss, kv0 + 64, a.Sk, tl);
            load_rows<HD, NT>(rv, vb_, a.v_ss, kv0 + 64, a.Sk, tl);
            load_trans<HD, NT>(rt, ktb, a.kt_sd, kv0 + 64, tl);
        }
        const uint64_t valid = key_valid_word(a, b, kv0, lane);
        const bool skip = a.causal && (kv0 > qw0 + 31 + a.q_off);
        if (!skip) {
            const bool full = valid == ~0ull && (!a.causal || kv0 + 63 <= qw0 + a.q_off);   // wave-uniform, see forward
            // one 32-key half at a time (scores, dS, its dQ contribution): halves the live score registers
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 st, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ds = 0; ds < T::DS; ++ds) {
                    st = mfma_32x32x16(frag_rows<HD>(sk, kb * 32, ds, ll), qf[ds], st);
                    dp = mfma_32x32x16(frag_rows<HD>(sv, kb * 32, ds, ll), dof[ds], dp);
                }
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = fast_exp2(fmaf(st[r], sc, -lse2));
                        st[r] = p * (dp[r] - dlt) * a.scale;   // dS^T
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // (hl: a per-iteration copy of h — with the lane-constant h the compiler precomputes sixteen 64-bit
                        //  key masks at kernel entry, spills them, and reloads each behind a vmcnt(0) inside this loop)
                        const int kl = kb * 32 + crow(r, hl);
                        bool ok = (valid >> kl) & 1ull;
                        if (a.causal) ok = ok && (kv0 + kl <= qi + a.q_off);
                        const float p = ok ? fast_exp2(st[r] * sc - lse2) : 0.f;
                        st[r] = p * (dp[r] - dlt) * a.scale;   // dS^T
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int r0 = 8 * s2;
                    u32x4 sf;
                    sf.x = pack_bf2(st[r0 + 0], st[r0 + 1]);
                    sf.y = pack_bf2(st[r0 + 2], st[r0 + 3]);
                    sf.z = pack_bf2(st[r0 + 4], st[r0 + 5]);
                    sf.w = pack_bf2(st[r0 + 6], st[r0 + 7]);
#pragma unroll
                    for (int db = 0; db < T::DB; ++db)
                        dq[db] = mfma_32x32x16(frag_trans<HD>(skt, db * 32, kb * 2 + s2, ll), sf, dq[db]);
                }
                sched_fence();
            }
        }
        if (more) {
            char* nk = smem + ((t + 1) & 1) * STAGE;
            store_rows<HD, NT>(nk, rk, tl);
            store_rows<HD, NT>(nk + T::KBYTES, rv, tl);
            store_trans<HD, NT>(nk + 2 * T::KBYTES, rt, tl);
        }
        __syncthreads();
    }
    if (nsp > 1) {
        if (qi < a.Sq) {
            float* op = a.part_o + ((((long)b * a.Hq + hq) * nsp + sp) * a.Sq + qi) * HD;
#pragma unroll
            for (int db = 0; db < T::DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w = {dq[db][4 * g + 0], dq[db][4 * g + 1], dq[db][4 * g + 2], dq[db][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(op + db * 32 + 8 * g + 4 * h) = w;
                }
        }
        return;
    }
    if (qi < a.Sq) {
        bf16_t* op = a.dq + b * a.dq_sb + (long)qi * a.dq_ss + hq * a.dq_sh;
#pragma unroll
        for (int db = 0; db < T::DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w.x = pack_bf2(dq[db][4 * g + 0], dq[db][4 * g + 1]);
                w.y = pack_bf2(dq[db][4 * g + 2], dq[db][4 * g + 3]);
                st8(op + db * 32 + 8 * g + 4 * h, w);
            }
    }
}

// ---------------------------------------------------------------------------
// backward, dK/dV: one workgroup = 128 keys of one (batch, kv-head); loops over the group's
// q-heads and over query tiles (so GQA's sum over the group needs no atomics).
//   S[q][key] = Q K^T, dP[q][key] = dO V^T  (key on lanes, query on registers)
//   dV^T[d][key] += dO^T[d][q] . P[q][key]      dK^T[d][key] += Q^T[d][q] . dS[q][key]
// WHICH: 0 = dK and dV in one pass (hd <= 64), 1 = dV only, 2 = dK only.  At hd = 128 the two accumulator sets plus
// both score tiles exceed the register file (the one-pass form spills, and a spill reload drains the tile prefetch),
// so the work is split into two launches that each recompute S; every tile a variant does not need is neither
// loaded nor staged.
template <int HD, int WHICH, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(AttnArgs a) {
    using T = Tile<HD>;
    constexpr int NT = NW * 64, KROWS = NW * 32;      // NW waves of 32 keys
    constexpr bool DV = WHICH != 2, DK = WHICH != 1;
    BRA_DYN_SMEM(smem);   // [2][Q tile | dO tile (DK) | Q^T tile (DK) | dO^T tile (DV) | lse(64) delta(64)]
    constexpr int OFF_D = T::KBYTES;
    constexpr int OFF_QT = OFF_D + (DK ? T::KBYTES : 0);
    constexpr int OFF_DT = OFF_QT + (DK ? T::TBYTES : 0);
    constexpr int OFF_L = OFF_DT + (DV ? T::TBYTES : 0);
    constexpr int STAGE = OFF_L + 512;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    int bx_, hkv, b;
    attn_block_coords(a.legacy_order, 0, bx_, hkv, b);                // (causal: key block 0 is the heaviest — ascending order is heaviest first)
    const int group = a.Hq / a.Hkv;
    const int nsp = a.nsplit_kv > 1 ? a.nsplit_kv : 1;
    const int sp = nsp > 1 ? bx_ % nsp : 0;
    if (nsp > 1) bx_ /= nsp;
    const int k0 = bx_ * KROWS;
    const int kw0 = k0 + wave * 32;
    const int kj = kw0 + (lane & 31);                 // this lane's key
    const int kr = kj < a.Sk ? kj : a.Sk - 1;
    bool kvalid = kj < a.Sk;
    if (kvalid && a.kmask) kvalid = a.kmask[(long)b * a.Sk + kj] != 0;
    const bool all_valid = wave_ballot(kvalid) == ~0ull;

    u32x4 kf[T::DS], vf[DK ? T::DS : 1];
    {
        const bf16_t* kp = a.k + b * a.k_sb + (long)kr * a.k_ss + hkv * a.k_sh;
        const bf16_t* vp = a.v + b * a.v_sb + (long)kr * a.v_ss + hkv * a.v_sh;
#pragma unroll
        for (int ds = 0; ds < T::DS; ++ds) {
            kf[ds] = ld16(kp + ds * 16 + 8 * h);
            if (DK) vf[ds] = ld16(vp + ds * 16 + 8 * h);
        }
    }
    f32x16 dk[DK ? T::DB : 1], dv[DV ? T::DB : 1];
#pragma unroll
    for (int i = 0; i < T::DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { if (DK) dk[i][r] = 0.f; if (DV) dv[i][r] = 0.f; }
    const float sc = a.scale * kLog2e;

    // first query tile that can see any key of this workgroup
    int qt_begin = 0;
    if (a.causal) { int first = k0 - a.q_off; qt_begin = first > 0 ? first / 64 : 0; }
    const int qt_end = (a.Sq + 63) / 64;
    const int per_head = qt_end > qt_begin ? qt_end - qt_begin : 0;
    const int nit_all = per_head * group;
    const int it_first = nsp > 1 ? (nit_all * sp) / nsp : 0;
    const int nit = nsp > 1 ? (nit_all * (sp + 1)) / nsp : nit_all;          // this part's (q-head, query tile) iterations: [it_first, nit)

    u32x4 rq[(64 * T::CH) / NT], rd[DK ? (64 * T::CH) / NT : 1], rqt[DK ? (HD * 8) / NT : 1], rdt[DV ? (HD * 8) / NT : 1];
    float rl = 0.f;   // threads 0..63: lse, 64..127: delta
    auto issue = [&](int it, int tid) {
        const int hq = hkv * group + it / per_head;
        const int s0 = (qt_begin + it % per_head) * 64;
        load_rows<HD, NT>(rq, a.q + b * a.q_sb + hq * a.q_sh, a.q_ss, s0, a.Sq, tid);
        if constexpr (DK) {
            load_rows<HD, NT>(rd, a.dout + b * a.do_sb + hq * a.do_sh, a.do_ss, s0, a.Sq, tid);
            load_trans<HD, NT>(rqt, a.qt + b * a.qt_sb + hq * a.qt_sh, a.qt_sd, s0, tid);
        }
        if constexpr (DV) load_trans<HD, NT>(rdt, a.dot + b * a.dot_sb + hq * a.dot_sh, a.dot_sd, s0, tid);
        if (tid < 128) {
            int qq = s0 + (tid & 63);
            qq = qq < a.Sq ? qq : a.Sq - 1;
            const long li = ((long)b * a.Hq + hq) * a.Sq + qq;
            rl = tid < 64 ? a.lse[li] * kLog2e : a.delta[li];
        }
    };
    auto commit = [&](int buf, int tid) {
        char* s = smem + buf * STAGE;
        store_rows<HD, NT>(s, rq, tid);
        if constexpr (DK) {
            store_rows<HD, NT>(s + OFF_D, rd, tid);
            store_trans<HD, NT>(s + OFF_QT, rqt, tid);
        }
        if constexpr (DV) store_trans<HD, NT>(s + OFF_DT, rdt, tid);
        if (tid < 128) reinterpret_cast<float*>(s + OFF_L)[tid] = rl;
    };
    if (nit > it_first) { issue(it_first, tid); commit(it_first & 1, tid); }
    __syncthreads();

    for (int it = it_first; it < nit; ++it) {
        const int s0 = (qt_begin + it % per_head) * 64;
        const char* sq = smem + (it & 1) * STAGE;
        const char* sd = sq + OFF_D;
        const char* sqt = sq + OFF_QT;
        const char* sdt = sq + OFF_DT;
        const float* sl = reinterpret_cast<const float*>(sq + OFF_L);
        const bool more = it + 1 < nit;
        const int tl = opaque_i(tid), ll = opaque_i(lane);      // see forward
        if (more) issue(it + 1, tl);
        // wave-uniform skip: every query of this tile is before every key of this wave
        const bool skip = a.causal && (s0 + 63 + a.q_off < kw0);
        if (!skip) {
            // wave-uniform: all 32 keys of this wave valid and visible to all 64 queries of the tile
            const bool full = all_valid && s0 + 63 < a.Sq && (!a.causal || kw0 + 31 <= s0 + a.q_off);
            // one 32-query half at a time (scores, P / dS, its dV / dK contribution): halves the live score registers
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f32x16 st, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ds = 0; ds < T::DS; ++ds) {
                    st = mfma_32x32x16(frag_rows<HD>(sq, qb * 32, ds, ll), kf[ds], st);
                    if constexpr (DK) dp = mfma_32x32x16(frag_rows<HD>(sd, qb * 32, ds, ll), vf[ds], dp);
                }
                // the per-query statistics sit at sl[32 qb + crow(r, h)]: one per-iteration base (the lane's half h, re-derived from the
                // opaque lane id) + compile-time offsets, so that the reads are `ds_read_b32 base offset:imm`.  With `h` from outside the
                // loop the compiler hoisted all 64 addresses, ran out of registers and reloaded them from scratch in every tile
                // (round 5: 95 scratch loads per tile in the one-pass dK + dV kernel, behind the same counter as the tile prefetch)
                const int hl = ll >> 5;
                const float* slh = sl + 4 * hl;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qo = qb * 32 + (r & 3) + 8 * (r >> 2);
                        const float p = fast_exp2(fmaf(st[r], sc, -slh[qo]));
                        st[r] = p;                                                     // P
                        if constexpr (DK) dp[r] = p * (dp[r] - slh[64 + qo]) * a.scale;   // dS
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qo = qb * 32 + (r & 3) + 8 * (r >> 2);
                        const int qi = s0 + qo + 4 * hl;
                        bool ok = kvalid && qi < a.Sq;
                        if (a.causal) ok = ok && (kj <= qi + a.q_off);
                        const float p = ok ? fast_exp2(st[r] * sc - slh[qo]) : 0.f;
                        st[r] = p;                                                     // P
                        if constexpr (DK) dp[r] = p * (dp[r] - slh[64 + qo]) * a.scale;   // dS
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int r0 = 8 * s2, s = qb * 2 + s2;
                    if constexpr (DV) {
                        u32x4 pf;
                        pf.x = pack_bf2(st[r0 + 0], st[r0 + 1]); pf.y = pack_bf2(st[r0 + 2], st[r0 + 3]);
                        pf.z = pack_bf2(st[r0 + 4], st[r0 + 5]); pf.w = pack_bf2(st[r0 + 6], st[r0 + 7]);
#pragma unroll
                        for (int db = 0; db < T::DB; ++db) dv[db] = mfma_32x32x16(frag_trans<HD>(sdt, db * 32, s, ll), pf, dv[db]);
                    }
                    if constexpr (DK) {
                        u32x4 sf;
                        sf.x = pack_bf2(dp[r0 + 0], dp[r0 + 1]); sf.y = pack_bf2(dp[r0 + 2], dp[r0 + 3]);
                        sf.z = pack_bf2(dp[r0 + 4], dp[r0 + 5]); sf.w = pack_bf2(dp[r0 + 6], dp[r0 + 7]);
#pragma unroll
                        for (int db = 0; db < T::DB; ++db) dk[db] = mfma_32x32x16(frag_trans<HD>(sqt, db * 32, s, ll), sf, dk[db]);
                    }
                }
                sched_fence();
            }
        }
        if (more) commit((it + 1) & 1, tl);
        __syncthreads();
    }
    if (nsp > 1) {
        if (kj < a.Sk) {
            const long row = ((((long)b * a.Hkv + hkv) * nsp + sp) * a.Sk + kj) * HD;
#pragma unroll
            for (int db = 0; db < T::DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (DK) {
                        f32x4 w = {dk[db][4 * g + 0], dk[db][4 * g + 1], dk[db][4 * g + 2], dk[db][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(a.part_dk + row + db * 32 + 8 * g + 4 * h) = w;
                    }
                    if constexpr (DV) {
                        f32x4 w = {dv[db][4 * g + 0], dv[db][4 * g + 1], dv[db][4 * g + 2], dv[db][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(a.part_dv + row + db * 32 + 8 * g + 4 * h) = w;
                    }
                }
        }
        return;
    }
    if (kj < a.Sk) {
        bf16_t* kp = a.dk + b * a.dk_sb + (long)kj * a.dk_ss + hkv * a.dk_sh;
        bf16_t* vp = a.dv + b * a.dv_sb + (long)kj * a.dv_ss + hkv * a.dv_sh;
#pragma unroll
        for (int db = 0; db < T::DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                if constexpr (DK) {
                    w.x = pack_bf2(dk[db][4 * g + 0], dk[db][4 * g + 1]);
                    w.y = pack_bf2(dk[db][4 * g + 2], dk[db][4 * g + 3]);
                    st8(kp + db * 32 + 8 * g + 4 * h, w);
                }
                if constexpr (DV) {
                    w.x = pack_bf2(dv[db][4 * g + 0], dv[db][4 * g + 1]);
                    w.y = pack_bf2(dv[db][4 * g + 2], dv[db][4 * g + 3]);
                    st8(vp + db * 32 + 8 * g + 4 * h, w);
                }
            }
    }
}

// ---------------------------------------------------------------------------
// single-token decode attention over a KV cache (HF DynamicCache step, TF:generation/utils.py:2876-2925).
// HBM-bound: one wave per (sequence, kv-head, key chunk); lane-per-key dot products for the scores,
// lane-per-dimension accumulation for the values; chunk partials (max, sum, O) are merged by a
// second tiny kernel (split over the context so that B*Hkv waves are not the only parallelism).
struct DecodeArgs {
    const bf16_t* q;      // [B, Hq, hd]
    const bf16_t* kc;     // K cache [B, Hkv, Smax, hd]
    const bf16_t* vc;     // V cache [B, Hkv, Smax, hd]
    const uint8_t* kmask; // [B, Smax] validity of cached positions (left padding) or null
    float* part_o;        // [B, Hq, nchunk, hd]
    float* part_ml;       // [B, Hq, nchunk, 2]
    bf16_t* o;            // [B, Hq*hd]
    int B, Hq, Hkv, hd, Smax, len, chunk, nchunk;
    float scale;
    const int* t_ptr; int npc;   // merge only: device-side chunk count = npc + ceil((t+1)/64)
};

// partial kernel: a wave owns CK = 128 consecutive cached positions of one (sequence, kv-head).
// A K/V row (hd bf16) is covered by LPK = hd/8 lanes with one 16-byte load each, so one wave-load
// touches 64/LPK whole rows; every lane keeps its 8-dim slice of the (pre-scaled) queries of the
// G q-heads of the group in registers.  Two passes over the chunk (scores -> max -> exp/accumulate)
// keep everything statically indexed in registers.
template <int HD, int G>
__global__ __launch_bounds__(256) void attn_decode_partial_kernel(DecodeArgs a) {
    constexpr int CK = 128;
    constexpr int LPK = HD / 8;
    constexpr int KPI = 64 / LPK;
    constexpr int NIT = CK / KPI;
    const int lane = lane_id();
    const int c = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int hkv = (int)blockIdx.y, b = (int)blockIdx.z;
    if (c >= a.nchunk) return;
    const int s_begin = c * CK;
    int s_end = s_begin + CK;
    s_end = s_end < a.len ? s_end : a.len;
    const int kg = lane / LPK, dl = lane % LPK;
    const bf16_t* kb_ = a.kc + ((long)b * a.Hkv + hkv) * a.Smax * HD + dl * 8;
    const bf16_t* vb_ = a.vc + ((long)b * a.Hkv + hkv) * a.Smax * HD + dl * 8;
    const float sc = a.scale * kLog2e;
    float qv[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        unpack8(ld16(a.q + ((long)b * a.Hq + hkv * G + g) * HD + dl * 8), qv[g]);
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[g][i] *= sc;
    }
    float sco[NIT][G];
    float m[G];
#pragma unroll
    for (int g = 0; g < G; ++g) m[g] = kNeg;
    // validity of the chunk's 128 positions as two wave ballots (no per-key branch: a conditional byte load
    // inside the unrolled loop serialises the 32 K-row loads behind each other)
    uint64_t vbits[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int key = s_begin + hh * 64 + lane;
        bool okk = key < s_end;
        const uint8_t mb = a.kmask ? a.kmask[(long)b * a.Smax + (okk ? key : s_end - 1)] : (uint8_t)1;
        vbits[hh] = wave_ballot(okk && mb != 0);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int key = s_begin + it * KPI + kg;
        const int rel_ = it * KPI + kg;
        const bool ok = (vbits[rel_ >> 6] >> (rel_ & 63)) & 1ull;
        const int kc_ = key < s_end ? key : s_end - 1;
        float f[8];
        unpack8(ld16(kb_ + (long)kc_ * HD), f);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) d += f[i] * qv[g][i];
#pragma unroll
            for (int mk = LPK >> 1; mk >= 1; mk >>= 1) d += wave_shfl_xor(d, mk);
            d = ok ? d : kNeg;
            sco[it][g] = d;
            m[g] = fmaxf(m[g], d);
        }
    }
    float l[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int mk = 32; mk >= LPK; mk >>= 1) m[g] = fmaxf(m[g], wave_shfl_xor(m[g], mk));
        l[g] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[g][i] = 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int key = s_begin + it * KPI + kg;
        const int kc_ = key < s_end ? key : s_end - 1;
        float f[8];
        unpack8(ld16(vb_ + (long)kc_ * HD), f);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float p = sco[it][g] > 0.5f * kNeg ? exp2f(sco[it][g] - m[g]) : 0.f;
            l[g] += p;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[g][i] += p * f[i];
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int mk = 32; mk >= LPK; mk >>= 1) {
            l[g] += wave_shfl_xor(l[g], mk);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[g][i] += wave_shfl_xor(acc[g][i], mk);
        }
        const int hq = hkv * G + g;
        const long base = ((long)b * a.Hq + hq) * a.nchunk + c;
        if (kg == 0) {
            f32x4 lo = {acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
            f32x4 hi = {acc[g][4], acc[g][5], acc[g][6], acc[g][7]};
            *reinterpret_cast<f32x4*>(a.part_o + base * HD + dl * 8) = lo;
            *reinterpret_cast<f32x4*>(a.part_o + base * HD + dl * 8 + 4) = hi;
        }
        if (lane == 0) { a.part_ml[base * 2] = m[g]; a.part_ml[base * 2 + 1] = l[g]; }
    }
}

template <int HD>
__global__ __launch_bounds__(64) void attn_decode_merge_kernel(DecodeArgs a) {
    const int lane = lane_id();
    const int hq = (int)blockIdx.x, b = (int)blockIdx.y;
    const int nchunk = a.t_ptr ? a.npc + (a.t_ptr[0] + 64) / 64 : a.nchunk;
    const long base = ((long)b * a.Hq + hq) * nchunk;
    constexpr int EPL = HD / 64 > 0 ? HD / 64 : 1;
    constexpr int PRE = 48;                          // partial rows requested before anything is known about their weights
    const int nc = nchunk < 256 ? nchunk : 256;
    const bool dlive = lane * EPL < HD;
    // (1) every load of the kernel is issued here: the (max, sum) pairs and the first PRE partial rows are independent,
    //     so the merge costs one memory round trip instead of one per batch of rows
    float mc[4], lc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = lane + 64 * r;
        const int cc = c < nchunk ? c : 0;
        mc[r] = a.part_ml[(base + cc) * 2];
        lc[r] = a.part_ml[(base + cc) * 2 + 1];
    }
    float v0[PRE][EPL];
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int cc = u < nc ? u : nc - 1;
#pragma unroll
        for (int e = 0; e < EPL; ++e) v0[u][e] = a.part_o[(base + cc) * HD + (dlive ? lane * EPL : 0) + e];
    }
    float m = kNeg;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (lane + 64 * r >= nchunk) { mc[r] = kNeg; lc[r] = 0.f; }
        m = fmaxf(m, mc[r]);
    }
    m = wave_max<64>(m);
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { mc[r] = exp2f(mc[r] - m); l += lc[r] * mc[r]; }     // mc now holds the chunk weight
    l = wave_sum<64>(l);
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
#pragma unroll
    for (int u = 0; u < PRE; ++u) {                  // chunks 0 .. 47 live in lane u of weight register 0
        const float w = u < nc ? wave_shfl(mc[0], u) : 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] += v0[u][e] * w;
    }
    for (int c0 = PRE; c0 < nc; c0 += 8) {
        float v[8][EPL], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u;
            const int cc = c < nc ? c : nc - 1;
            const float wsel = (cc >> 6) == 0 ? mc[0] : ((cc >> 6) == 1 ? mc[1] : ((cc >> 6) == 2 ? mc[2] : mc[3]));
            w[u] = c < nc ? wave_shfl(wsel, cc & 63) : 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[u][e] = dlive ? a.part_o[(base + cc) * HD + lane * EPL + e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] += v[u][e] * w[u];
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (dlive)
#pragma unroll
        for (int e = 0; e < EPL; ++e) a.o[((long)b * a.Hq + hq) * HD + lane * EPL + e] = f2bf(acc[e] * inv);
}

}  // namespace bra

using namespace bra;

#ifdef BRA_EMU
static int g_attn_fwd4 = 1;
static unsigned long long* g_attn_probe = nullptr;
#else
static std::atomic<int> g_attn_fwd4{1};
static std::atomic<unsigned long long*> g_attn_probe{nullptr};
#endif

template <int HD>
static int launch_fwd(const AttnArgs& a, bra_stream_t st) {
    const size_t smem = 2 * (Tile<HD>::KBYTES + Tile<HD>::TBYTES);
    // whole 256-query workgroups, no key split: the 4-wave kernel with 64 queries per wave (k_attn4.hip)
    if constexpr (HD >= 64) {
        if (a.Sq > 128 && g_attn_fwd4) {
            int rc = launch_fwd4<HD>(a, st);
            if (rc || a.nsplit <= 1) return rc;
            const long rows = (long)a.B * a.Hq * a.Sq;
            constexpr int RPB = 256 / (HD / 4);
            BRA_LAUNCH((attn_combine_kernel<HD>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(256), 0, st, a);
            return BRA_LAUNCH_STATUS();
        }
    }
    if (HD >= 64 && a.Sq > 128) {
        const int ns = a.nsplit > 1 ? a.nsplit : 1;
        BRA_ALLOW_SMEM((attn_fwd_kernel<HD, (HD >= 64 ? 8 : 4)>), smem);
        BRA_LAUNCH((attn_fwd_kernel<HD, (HD >= 64 ? 8 : 4)>), dim3(((a.Sq + 255) / 256) * ns, a.Hq, a.B), dim3(512), smem, st, a);
        int rc = BRA_LAUNCH_STATUS();
        if (rc || ns == 1) return rc;
        const long rows = (long)a.B * a.Hq * a.Sq;
        constexpr int RPB = 256 / (HD / 4);
        BRA_LAUNCH((attn_combine_kernel<HD>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(256), 0, st, a);
        return BRA_LAUNCH_STATUS();
    }
    if (a.nsplit > 1) return BRA_ERR_UNSUPPORTED;
    BRA_ALLOW_SMEM((attn_fwd_kernel<HD, 4>), smem);
    BRA_LAUNCH((attn_fwd_kernel<HD, 4>), dim3((a.Sq + 127) / 128, a.Hq, a.B), dim3(256), smem, st, a);
    return BRA_LAUNCH_STATUS();
}
template <int HD>
static int launch_sum_parts(const float* part, int ns, int B, int H, int S, bf16_t* out, long sb, long ss, long sh, bra_stream_t st) {
    const long rows = (long)B * H * S;
    constexpr int RPB = 256 / (HD / 4);
    BRA_LAUNCH((attn_sum_parts_kernel<HD>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(256), 0, st, part, ns, B, H, S, out, sb, ss, sh);
    return BRA_LAUNCH_STATUS();
}

#ifdef BRA_EMU
static int g_attn_bwd4 = 3;
#else
static std::atomic<int> g_attn_bwd4{3};
#endif

template <int HD>
static int launch_dq(const AttnArgs& a, bra_stream_t st) {
    const size_t smem = 2 * (2 * Tile<HD>::KBYTES + Tile<HD>::TBYTES);
    // whole 256-query workgroups, no key split: the pipelined 4-wave kernel (k_attn4b.hip)
    if constexpr (HD >= 64) {
        if (a.Sq > 128 && (g_attn_bwd4 & 1)) {
            int rc = launch_dq4<HD>(a, st);
            if (rc || a.nsplit <= 1) return rc;
            return launch_sum_parts<HD>(a.part_o, a.nsplit, a.B, a.Hq, a.Sq, a.dq, a.dq_sb, a.dq_ss, a.dq_sh, st);
        }
    }
    if (HD >= 64 && a.Sq > 128) {
        const int ns = a.nsplit > 1 ? a.nsplit : 1;
        BRA_ALLOW_SMEM((attn_bwd_dq_kernel<HD, (HD >= 64 ? 8 : 4)>), smem);
        BRA_LAUNCH((attn_bwd_dq_kernel<HD, (HD >= 64 ? 8 : 4)>), dim3(((a.Sq + 255) / 256) * ns, a.Hq, a.B), dim3(512), smem, st, a);
        if (ns > 1) {
            int rc = BRA_LAUNCH_STATUS();
            if (rc) return rc;
            return launch_sum_parts<HD>(a.part_o, ns, a.B, a.Hq, a.Sq, a.dq, a.dq_sb, a.dq_ss, a.dq_sh, st);
        }
    } else if (a.nsplit > 1) {
        return BRA_ERR_UNSUPPORTED;
    } else {
        BRA_ALLOW_SMEM((attn_bwd_dq_kernel<HD, 4>), smem);
        BRA_LAUNCH((attn_bwd_dq_kernel<HD, 4>), dim3((a.Sq + 127) / 128, a.Hq, a.B), dim3(256), smem, st, a);
    }
    return BRA_LAUNCH_STATUS();
}
template <int HD, int WHICH, int NW>
static int launch_dkv_v(const AttnArgs& a, bra_stream_t st) {
    constexpr bool DV = WHICH != 2, DK = WHICH != 1;
    const size_t smem = 2 * (Tile<HD>::KBYTES + (DK ? Tile<HD>::KBYTES + Tile<HD>::TBYTES : 0) + (DV ? Tile<HD>::TBYTES : 0) + 512);
    BRA_ALLOW_SMEM((attn_bwd_dkv_kernel<HD, WHICH, NW>), smem);
    BRA_LAUNCH((attn_bwd_dkv_kernel<HD, WHICH, NW>), dim3((a.Sk + NW * 32 - 1) / (NW * 32), a.Hkv, a.B), dim3(NW * 64), smem, st, a);
    return BRA_LAUNCH_STATUS();
}
template <int HD>
static int launch_dkv(const AttnArgs& a, bra_stream_t st) {
    // whole 256-key workgroups, long query loops: the pipelined 4-wave kernels (k_attn4b.hip; bra_attn_set_bwd4 bit 1), with the
    // (q-head, query tile) loop in nsplit_kv parts for grids that cannot fill the chip
    if constexpr (HD >= 64) {
        if ((g_attn_bwd4 & 2) && a.Sk > 128 && a.Sq > 128) {
            int rc = launch_dkv4<HD>(a, st);
            if (rc || a.nsplit_kv <= 1) return rc;
            rc = launch_sum_parts<HD>(a.part_dk, a.nsplit_kv, a.B, a.Hkv, a.Sk, a.dk, a.dk_sb, a.dk_ss, a.dk_sh, st);
            if (rc) return rc;
            return launch_sum_parts<HD>(a.part_dv, a.nsplit_kv, a.B, a.Hkv, a.Sk, a.dv, a.dv_sb, a.dv_ss, a.dv_sh, st);
        }
    }
    if (a.nsplit_kv > 1) {
        // (one prompt: 144 four-wave workgroups with a triangular load — the (q-head, query tile) loop of every key block in parts;
        //  only the one-launch dK + dV form is split: the shapes that take the two 8-wave kernels fill the chip)
        const bool fused = HD < 128 || (a.Sk > 128 && (a.Sq <= 512 || (long)((a.Sk + 255) / 256) * a.Hkv * a.B < 256));
        if (!fused) return BRA_ERR_UNSUPPORTED;
        constexpr int NW4 = 4;
        const size_t smem = 2 * (Tile<HD>::KBYTES + Tile<HD>::KBYTES + Tile<HD>::TBYTES + Tile<HD>::TBYTES + 512);
        BRA_ALLOW_SMEM((attn_bwd_dkv_kernel<HD, 0, NW4>), smem);
        BRA_LAUNCH((attn_bwd_dkv_kernel<HD, 0, NW4>), dim3(((a.Sk + NW4 * 32 - 1) / (NW4 * 32)) * a.nsplit_kv, a.Hkv, a.B), dim3(NW4 * 64), smem, st, a);
        int rc = BRA_LAUNCH_STATUS();
        if (rc) return rc;
        rc = launch_sum_parts<HD>(a.part_dk, a.nsplit_kv, a.B, a.Hkv, a.Sk, a.dk, a.dk_sb, a.dk_ss, a.dk_sh, st);
        if (rc) return rc;
        return launch_sum_parts<HD>(a.part_dv, a.nsplit_kv, a.B, a.Hkv, a.Sk, a.dv, a.dv_sb, a.dv_ss, a.dv_sh, st);
    }
    if (HD < 128) return launch_dkv_v<HD, 0, 4>(a, st);
    constexpr int NW = HD >= 128 ? 8 : 4;              // 8 waves = 2 per SIMD share one staged query tile
    if (a.Sk <= 128) {
        int rc = launch_dkv_v<HD, 1, 4>(a, st);
        return rc ? rc : launch_dkv_v<HD, 2, 4>(a, st);
    }
    // short query loops (the completion segment of a shared-prompt pass: Sq = C) or grids that cannot fill the chip twice over
    // (one prompt: 72 workgroups of 256 keys): ONE launch that keeps dK and dV (4 waves, one per SIMD, the whole register file; S and
    // P computed once) instead of two that each recompute S — same arithmetic, bit-identical.  Long loops at full grids stay on the
    // two 8-wave kernels: there the second wave per SIMD is worth more than the third S (B = 8, S = 2436: 1.83 vs 2.16 ms).
    const long grid8 = (long)((a.Sk + 255) / 256) * a.Hkv * a.B;
    if (a.Sq <= 512 || grid8 < 256) return launch_dkv_v<HD, 0, 4>(a, st);
    int rc = launch_dkv_v<HD, 1, NW>(a, st);
    return rc ? rc : launch_dkv_v<HD, 2, NW>(a, st);
}

#ifdef BRA_EMU
static int g_attn_legacy_order = 0;
#else
static std::atomic<int> g_attn_legacy_order{0};
#endif

// the tile loaders address one (batch, head) slice with 32-bit element offsets built by a 24-bit multiply (load_rows / load_trans)
static bool attn_fit32(long rows, long stride) {
    return stride >= 0 && stride < (1L << 24) && rows < (1L << 24) && rows * stride < (1L << 31);
}

static int attn_check(int B, int Hq, int Hkv, int Sq, int Sk, int hd) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Sk <= 0 || Hq % Hkv) return BRA_ERR_ARG;
    if (hd != 32 && hd != 64 && hd != 128) return BRA_ERR_UNSUPPORTED;
    return 0;
}

extern "C" int bra_attn_fwd(const void* q, long q_sb, long q_ss, long q_sh, const void* k, long k_sb, long k_ss,
                            long k_sh, const void* vt, long vt_sb, long vt_sh, long vt_sd, void* o, long o_sb,
                            long o_ss, long o_sh, float* lse, const void* kmask, int B, int Hq, int Hkv, int Sq,
                            int Sk, int hd, int causal, int q_off, float scale, void* stream) {
    int e = attn_check(B, Hq, Hkv, Sq, Sk, hd);
    if (e) return e;
    if (!q || !k || !vt || !o || vt_sd % 8 || vt_sd < ((Sk + 63) / 64) * 64) return BRA_ERR_ARG;
    if (!attn_fit32(Sk, k_ss) || !attn_fit32(hd, vt_sd)) return BRA_ERR_UNSUPPORTED;
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.q_sb = q_sb; a.q_ss = q_ss; a.q_sh = q_sh;
    a.k = (const bf16_t*)k; a.k_sb = k_sb; a.k_ss = k_ss; a.k_sh = k_sh;
    a.vt = (const bf16_t*)vt; a.vt_sb = vt_sb; a.vt_sh = vt_sh; a.vt_sd = vt_sd;
    a.o = (bf16_t*)o; a.o_sb = o_sb; a.o_ss = o_ss; a.o_sh = o_sh;
    a.lse = lse; a.kmask = (const uint8_t*)kmask;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.Sq = Sq; a.Sk = Sk; a.causal = causal; a.q_off = q_off; a.scale = scale;
    a.legacy_order = g_attn_legacy_order;
#if defined(BRA_DEBUG) && !defined(BRA_EMU)
    a.probe = g_attn_probe;
#endif
    bra_stream_t st = (bra_stream_t)stream;
    if (hd == 128) return launch_fwd<128>(a, st);
    if (hd == 64) return launch_fwd<64>(a, st);
    return launch_fwd<32>(a, st);
}

// bra_attn_fwd with the key range of every query block cut into `nsplit` parts (2 .. 8) that run as separate workgroups, then one
// merge launch: for grids that cannot fill the chip (one prompt: 144 workgroups with a triangular load; a 256-query completion segment:
// 128).  part_o fp32 [B, Hq, nsplit, Sq, hd], part_ml fp32 [B, Hq, nsplit, Sq, 2]: caller-owned workspaces.  Same softmax / PV
// arithmetic per tile; the parts are merged in key order (TF:qwen3:185-207).  Needs Sq > 128 and hd >= 64.
extern "C" int bra_attn_fwd_split(const void* q, long q_sb, long q_ss, long q_sh, const void* k, long k_sb, long k_ss,
                                  long k_sh, const void* vt, long vt_sb, long vt_sh, long vt_sd, void* o, long o_sb,
                                  long o_ss, long o_sh, float* lse, const void* kmask, int B, int Hq, int Hkv, int Sq,
                                  int Sk, int hd, int causal, int q_off, float scale, int nsplit, float* part_o, float* part_ml,
                                  void* stream) {
    int e = attn_check(B, Hq, Hkv, Sq, Sk, hd);
    if (e) return e;
    if (!q || !k || !vt || !o || vt_sd % 8 || vt_sd < ((Sk + 63) / 64) * 64) return BRA_ERR_ARG;
    if (nsplit < 2 || nsplit > 8 || !part_o || !part_ml || o_sh % 4 || o_ss % 4 || o_sb % 4) return BRA_ERR_ARG;
    if (hd < 64 || Sq <= 128) return BRA_ERR_UNSUPPORTED;
    if (!attn_fit32(Sk, k_ss) || !attn_fit32(hd, vt_sd)) return BRA_ERR_UNSUPPORTED;
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.q_sb = q_sb; a.q_ss = q_ss; a.q_sh = q_sh;
    a.k = (const bf16_t*)k; a.k_sb = k_sb; a.k_ss = k_ss; a.k_sh = k_sh;
    a.vt = (const bf16_t*)vt; a.vt_sb = vt_sb; a.vt_sh = vt_sh; a.vt_sd = vt_sd;
    a.o = (bf16_t*)o; a.o_sb = o_sb; a.o_ss = o_ss; a.o_sh = o_sh;
    a.lse = lse; a.kmask = (const uint8_t*)kmask;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.Sq = Sq; a.Sk = Sk; a.causal = causal; a.q_off = q_off; a.scale = scale;
    a.legacy_order = g_attn_legacy_order;
    a.nsplit = nsplit; a.part_o = part_o; a.part_ml = part_ml;
    bra_stream_t st = (bra_stream_t)stream;
    if (hd == 128) return launch_fwd<128>(a, st);
    return launch_fwd<64>(a, st);
}

static int attn_bwd_impl(const void* q, long q_sb, long q_ss, long q_sh, const void* k, long k_sb, long k_ss,
                            long k_sh, const void* v, long v_sb, long v_ss, long v_sh, const void* dout, long do_sb,
                            long do_ss, long do_sh, const void* kt, long kt_sb, long kt_sh, long kt_sd,
                            const void* qt, long qt_sb, long qt_sh, long qt_sd, const void* dot, long dot_sb,
                            long dot_sh, long dot_sd, const float* lse, const float* delta, const void* kmask,
                            void* dq, long dq_sb, long dq_ss, long dq_sh, void* dk, long dk_sb, long dk_ss,
                            long dk_sh, void* dv, long dv_sb, long dv_ss, long dv_sh, int B, int Hq, int Hkv,
                            int Sq, int Sk, int hd, int causal, int q_off, float scale, int nsplit_dq, float* part_dq, int nsplit_kv, float* part_dk,
                            float* part_dv, void* stream) {
    int e = attn_check(B, Hq, Hkv, Sq, Sk, hd);
    if (e) return e;
    if (!q || !k || !v || !dout || !kt || !qt || !dot || !lse || !delta || !dq || !dk || !dv) return BRA_ERR_ARG;
    const int sk_pad = ((Sk + 63) / 64) * 64, sq_pad = ((Sq + 63) / 64) * 64;
    if (kt_sd % 8 || qt_sd % 8 || dot_sd % 8 || kt_sd < sk_pad || qt_sd < sq_pad || dot_sd < sq_pad) return BRA_ERR_ARG;
    if (!attn_fit32(Sq, q_ss) || !attn_fit32(Sk, k_ss) || !attn_fit32(Sk, v_ss) || !attn_fit32(Sq, do_ss) || !attn_fit32(hd, kt_sd) ||
        !attn_fit32(hd, qt_sd) || !attn_fit32(hd, dot_sd)) return BRA_ERR_UNSUPPORTED;
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.q_sb = q_sb; a.q_ss = q_ss; a.q_sh = q_sh;
    a.k = (const bf16_t*)k; a.k_sb = k_sb; a.k_ss = k_ss; a.k_sh = k_sh;
    a.v = (const bf16_t*)v; a.v_sb = v_sb; a.v_ss = v_ss; a.v_sh = v_sh;
    a.dout = (const bf16_t*)dout; a.do_sb = do_sb; a.do_ss = do_ss; a.do_sh = do_sh;
    a.kt = (const bf16_t*)kt; a.kt_sb = kt_sb; a.kt_sh = kt_sh; a.kt_sd = kt_sd;
    a.qt = (const bf16_t*)qt; a.qt_sb = qt_sb; a.qt_sh = qt_sh; a.qt_sd = qt_sd;
    a.dot = (const bf16_t*)dot; a.dot_sb = dot_sb; a.dot_sh = dot_sh; a.dot_sd = dot_sd;
    a.lse = const_cast<float*>(lse); a.delta = delta; a.kmask = (const uint8_t*)kmask;
    a.dq = (bf16_t*)dq; a.dq_sb = dq_sb; a.dq_ss = dq_ss; a.dq_sh = dq_sh;
    a.dk = (bf16_t*)dk; a.dk_sb = dk_sb; a.dk_ss = dk_ss; a.dk_sh = dk_sh;
    a.dv = (bf16_t*)dv; a.dv_sb = dv_sb; a.dv_ss = dv_ss; a.dv_sh = dv_sh;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.Sq = Sq; a.Sk = Sk; a.causal = causal; a.q_off = q_off; a.scale = scale;
    a.legacy_order = g_attn_legacy_order;
    if (nsplit_dq > 1) { if (nsplit_dq > 8 || !part_dq) return BRA_ERR_ARG; a.nsplit = nsplit_dq; a.part_o = part_dq; }
    if (nsplit_kv > 1) { if (nsplit_kv > 8 || !part_dk || !part_dv) return BRA_ERR_ARG; a.nsplit_kv = nsplit_kv; a.part_dk = part_dk; a.part_dv = part_dv; }
    if ((nsplit_dq > 1 || nsplit_kv > 1) && ((dq_sb | dq_ss | dq_sh | dk_sb | dk_ss | dk_sh | dv_sb | dv_ss | dv_sh) & 3)) return BRA_ERR_ARG;
    bra_stream_t st = (bra_stream_t)stream;
    int r;
    if (hd == 128) { r = launch_dq<128>(a, st); if (r) return r; return launch_dkv<128>(a, st); }
    if (hd == 64) { r = launch_dq<64>(a, st); if (r) return r; return launch_dkv<64>(a, st); }
    r = launch_dq<32>(a, st); if (r) return r; return launch_dkv<32>(a, st);
}

extern "C" int bra_attn_bwd(const void* q, long q_sb, long q_ss, long q_sh, const void* k, long k_sb, long k_ss,
                            long k_sh, const void* v, long v_sb, long v_ss, long v_sh, const void* dout, long do_sb,
                            long do_ss, long do_sh, const void* kt, long kt_sb, long kt_sh, long kt_sd,
                            const void* qt, long qt_sb, long qt_sh, long qt_sd, const void* dot, long dot_sb,
                            long dot_sh, long dot_sd, const float* lse, const float* delta, const void* kmask,
                            void* dq, long dq_sb, long dq_ss, long dq_sh, void* dk, long dk_sb, long dk_ss,
                            long dk_sh, void* dv, long dv_sb, long dv_ss, long dv_sh, int B, int Hq, int Hkv,
                            int Sq, int Sk, int hd, int causal, int q_off, float scale, void* stream) {
    return attn_bwd_impl(q, q_sb, q_ss, q_sh, k, k_sb, k_ss, k_sh, v, v_sb, v_ss, v_sh, dout, do_sb, do_ss, do_sh, kt, kt_sb, kt_sh, kt_sd,
                         qt, qt_sb, qt_sh, qt_sd, dot, dot_sb, dot_sh, dot_sd, lse, delta, kmask, dq, dq_sb, dq_ss, dq_sh, dk, dk_sb, dk_ss,
                         dk_sh, dv, dv_sb, dv_ss, dv_sh, B, Hq, Hkv, Sq, Sk, hd, causal, q_off, scale, 1, nullptr, 1, nullptr, nullptr, stream);
}

// bra_attn_bwd for grids that cannot fill the chip: the dQ kernel's key range in `nsplit_dq` parts (part_dq fp32 [B, Hq, nsplit_dq, Sq,
// hd]), the one-launch dK + dV kernel's (q-head, query tile) loop in `nsplit_kv` parts (part_dk / part_dv fp32 [B, Hkv, nsplit_kv, Sk,
// hd]); the parts are added in order by a sum launch each.  1 = no split for that kernel.  Same per-tile arithmetic.
extern "C" int bra_attn_bwd_split(const void* q, long q_sb, long q_ss, long q_sh, const void* k, long k_sb, long k_ss,
                                  long k_sh, const void* v, long v_sb, long v_ss, long v_sh, const void* dout, long do_sb,
                                  long do_ss, long do_sh, const void* kt, long kt_sb, long kt_sh, long kt_sd,
                                  const void* qt, long qt_sb, long qt_sh, long qt_sd, const void* dot, long dot_sb,
                                  long dot_sh, long dot_sd, const float* lse, const float* delta, const void* kmask,
                                  void* dq, long dq_sb, long dq_ss, long dq_sh, void* dk, long dk_sb, long dk_ss,
                                  long dk_sh, void* dv, long dv_sb, long dv_ss, long dv_sh, int B, int Hq, int Hkv,
                                  int Sq, int Sk, int hd, int causal, int q_off, float scale, int nsplit_dq, float* part_dq,
                                  int nsplit_kv, float* part_dk, float* part_dv, void* stream) {
    if (nsplit_dq < 1 || nsplit_kv < 1) return BRA_ERR_ARG;
    return attn_bwd_impl(q, q_sb, q_ss, q_sh, k, k_sb, k_ss, k_sh, v, v_sb, v_ss, v_sh, dout, do_sb, do_ss, do_sh, kt, kt_sb, kt_sh, kt_sd,
                         qt, qt_sb, qt_sh, qt_sd, dot, dot_sb, dot_sh, dot_sd, lse, delta, kmask, dq, dq_sb, dq_ss, dq_sh, dk, dk_sb, dk_ss,
                         dk_sh, dv, dv_sb, dv_ss, dv_sh, B, Hq, Hkv, Sq, Sk, hd, causal, q_off, scale, nsplit_dq, part_dq, nsplit_kv, part_dk,
                         part_dv, stream);
}

#ifdef BRA_DEBUG
extern "C" int bra_attn_set_block_order(int legacy) { g_attn_legacy_order = legacy ? 1 : 0; return 0; }
// A/B knob: 0 = the 8-wave forward of rounds 1-5 for every shape, 1 (default) = the 4-wave kernel where it applies
extern "C" int bra_attn_set_fwd4(int on) { g_attn_fwd4 = on ? 1 : 0; return 0; }
// attention backward: bit 0 = the pipelined dQ kernel, bit 1 = the pipelined dK / dV kernels (k_attn4b.hip) where they apply
// (default 3), 0 = the kernels of rounds 1-5
extern "C" int bra_attn_set_bwd4(int mask) { g_attn_bwd4 = mask; return 0; }
// 80 x 8-byte device buffer that one workgroup of the next 4-wave forward launches fills with cycle counts (k_attn4.hip), or null
extern "C" int bra_attn_set_probe(void* p) { g_attn_probe = (unsigned long long*)p; return 0; }
#endif

extern "C" int bra_attn_decode_nchunk(int len) { return (len + 127) / 128; }

extern "C" int bra_attn_decode(const void* q, const void* kc, const void* vc, const void* kmask, float* part_o,
                               float* part_ml, void* o, int B, int Hq, int Hkv, int hd, int Smax, int len,
                               float scale, void* stream) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || len <= 0 || len > Smax) return BRA_ERR_ARG;
    if (!q || !kc || !vc || !part_o || !part_ml || !o) return BRA_ERR_ARG;
    const int G = Hq / Hkv;
    DecodeArgs a = {(const bf16_t*)q, (const bf16_t*)kc, (const bf16_t*)vc, (const uint8_t*)kmask, part_o, part_ml,
                    (bf16_t*)o, B, Hq, Hkv, hd, Smax, len, 128, (len + 127) / 128, scale};
    bra_stream_t st = (bra_stream_t)stream;
    dim3 grid((a.nchunk + 3) / 4, Hkv, B);
#define BRA_DEC(HD_, G_)                                                                              \
    if (hd == HD_ && G == G_) {                                                                       \
        BRA_LAUNCH((attn_decode_partial_kernel<HD_, G_>), grid, dim3(256), 0, st, a);                 \
        int r = BRA_LAUNCH_STATUS();                                                                  \
        if (r) return r;                                                                              \
        BRA_LAUNCH((attn_decode_merge_kernel<HD_>), dim3(Hq, B), dim3(64), 0, st, a);                 \
        return BRA_LAUNCH_STATUS();                                                                   \
    }
    BRA_DEC(128, 1) BRA_DEC(128, 2) BRA_DEC(128, 4)
    BRA_DEC(64, 1) BRA_DEC(64, 2) BRA_DEC(64, 4)
    BRA_DEC(32, 1) BRA_DEC(32, 2) BRA_DEC(32, 4)
#undef BRA_DEC
    return BRA_ERR_UNSUPPORTED;
}

extern "C" int bra_attn_decode_merge(const float* part_o, const float* part_ml, void* o, int B, int Hq, int hd, int nchunk,
                                     const int* t_dev, int npc, void* stream) {
    if (B <= 0 || Hq <= 0 || nchunk <= 0 || !part_o || !part_ml || !o) return BRA_ERR_ARG;
    DecodeArgs a = {};
    a.part_o = const_cast<float*>(part_o); a.part_ml = const_cast<float*>(part_ml); a.o = (bf16_t*)o;
    a.B = B; a.Hq = Hq; a.hd = hd; a.nchunk = nchunk; a.t_ptr = t_dev; a.npc = npc;
    bra_stream_t st = (bra_stream_t)stream;
    if (hd == 128) BRA_LAUNCH((attn_decode_merge_kernel<128>), dim3(Hq, B), dim3(64), 0, st, a);
    else if (hd == 64) BRA_LAUNCH((attn_decode_merge_kernel<64>), dim3(Hq, B), dim3(64), 0, st, a);
    else if (hd == 32) BRA_LAUNCH((attn_decode_merge_kernel<32>), dim3(Hq, B), dim3(64), 0, st, a);
    else return BRA_ERR_UNSUPPORTED;
    return BRA_LAUNCH_STATUS();
}
