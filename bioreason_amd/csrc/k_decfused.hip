// k_decfused.hip — fused kernels of the single-token decode step (the body of HF's `_sample` loop,
// TF:generation/utils.py:2876-2925 -> Qwen3DecoderLayer.forward TF:qwen3:294-323 with a KV cache).
// At M = 8 rows every op of a decoder layer is a few microseconds of HBM streaming, so the step is bound by
// the NUMBER of dependent launches; these kernels cut a layer from 14 launches to 6:
//   dec_gemm<NORM>        RMSNorm prologue (TF:qwen3:59-64) + skinny weight-streaming GEMM  (x -> qkv, x -> logits)
//   dec_attn_partial      per-head q/k RMSNorm + RoPE (TF:qwen3:252-256) + KV-cache append + chunked attention
//   attn_decode_merge     (k_attn.hip) merge of the chunk partials
//   dec_gemm              o_proj + residual
//   dec_gemm<NORM, ACT>   RMSNorm + gate/up GEMM on row-interleaved weights + SwiGLU epilogue (TF:qwen3:81-83)
//   dec_gemm              down_proj + residual
// Weight loads are issued in explicit batches of 8 x 16 B per lane ahead of their MFMAs (the compiler otherwise
// interleaves them with the consumers and keeps only 2-8 in flight).
#include "bra_device.h"
#include "bra_api_internal.h"

namespace bra {

constexpr float kNegD = -1.0e30f;
constexpr float kLog2eD = 1.4426950408889634f;

struct DecGemmArgs {
    const bf16_t* x; long ldx;       // [M, K]
    const bf16_t* nw; float eps;     // RMSNorm weight [K] (NORM)
    const bf16_t* W; long ldw;       // [N, K]
    const bf16_t* res; long ldres;   // [M, N] or null
    void* out; long ldo;             // bf16 [M, N] | bf16 [M, N/2] (ACT) | f32 [M, N]
    int M, N, K;
};

__device__ __forceinline__ float silu_d(float x) { return x / (1.f + __expf(-x)); }

// NCOL = output columns per workgroup: 16, or 8 (upper half of the MFMA tile idle) so that N = 2048 projections
// still put a workgroup on every one of the 256 CUs (HBM streaming is per-CU latency bound at this size)
// MP = rows handled by the RMSNorm prologue (8 or 16, compile-time so that its loads carry no run-time branch)
template <int NORM, int ACT, int OUTF32, int NCOL, int MP>
__global__ __launch_bounds__(256) void dec_gemm_kernel(DecGemmArgs g) {
    BRA_DYN_SMEM(smem);                       // NORM: normalised x rows, bf16 [M][K + 8]
    __shared__ float red[4][64][4];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = (int)blockIdx.x * NCOL;
    const int rm = fr < g.M ? fr : g.M - 1;
    const long xpitch = g.K + 8;
    const bool wlive = fr < NCOL;                 // lanes that stream a weight row
    int rn = n0 + (wlive ? fr : 0); rn = rn < g.N ? rn : g.N - 1;
    const bf16_t* wp = g.W + (long)rn * g.ldw + fq * 8;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int nk = g.K / 32;
    const int nbatch = (nk - wave + 31) / 32 > 0 ? (nk - wave + 31) / 32 : 0;
    u32x4 wc[8];                                  // first weight batch: requested before the prologue (independent of x)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        int kk = wave + 4 * u; kk = kk < nk ? kk : nk - 1;
        wc[u] = (NCOL == 16 || wlive) ? ld16(wp + (long)kk * 32) : zero4;
    }
    if (NORM) {
        // thread t owns 16-byte column chunk j = t, t+256, .. of EVERY row: all row loads of a chunk are issued
        // together (one memory latency), the per-row sums of squares are combined across the block through LDS
        __shared__ float ssq[4][MP];
        float ss[MP];
#pragma unroll
        for (int r = 0; r < MP; ++r) ss[r] = 0.f;
        const int nch = g.K / 8;
        long roff[MP];
#pragma unroll
        for (int r = 0; r < MP; ++r) roff[r] = (long)(r < g.M ? r : g.M - 1) * g.ldx;      // clamped: no branch around a load
        for (int j = tid; j < nch; j += 256) {
            u32x4 xr[MP];
#pragma unroll
            for (int r = 0; r < MP; ++r) xr[r] = ld16(g.x + roff[r] + j * 8);
#pragma unroll
            for (int r = 0; r < MP; ++r) {
                float f[8];
                unpack8(xr[r], f);
#pragma unroll
                for (int i = 0; i < 8; ++i) ss[r] += f[i] * f[i];
            }
        }
#pragma unroll
        for (int r = 0; r < MP; ++r) ss[r] = wave_sum<64>(ss[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < MP; ++r) ssq[wave][r] = ss[r];
        }
        __syncthreads();
        float rstd[MP];
#pragma unroll
        for (int r = 0; r < MP; ++r) rstd[r] = rsqrtf((ssq[0][r] + ssq[1][r] + ssq[2][r] + ssq[3][r]) / (float)g.K + g.eps);
        for (int j = tid; j < nch; j += 256) {
            float w[8];
            unpack8(ld16(g.nw + j * 8), w);
            u32x4 xr[MP];
#pragma unroll
            for (int r = 0; r < MP; ++r) xr[r] = ld16(g.x + roff[r] + j * 8);
#pragma unroll
            for (int r = 0; r < MP; ++r) {
                float f[8];
                unpack8(xr[r], f);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] = w[i] * round_bf(f[i] * rstd[r]);
                st16(reinterpret_cast<bf16_t*>(smem) + (long)r * xpitch + j * 8, pack8(f));
            }
        }
        __syncthreads();
    }
    const bf16_t* xg = g.x + (long)rm * g.ldx + fq * 8;
    const bf16_t* xs = reinterpret_cast<const bf16_t*>(smem) + (long)rm * xpitch + fq * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // batches of 8 k-steps, double-buffered in registers: batch i+1 is requested before batch i is consumed
    for (int kb = 0; kb < nbatch; ++kb) {
        const int kt = wave + 32 * kb;
        u32x4 wn[8], x[8];
        const int ktn = kt + 32;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int kk = ktn + 4 * u; kk = kk < nk ? kk : nk - 1;
            wn[u] = (NCOL == 16 || wlive) ? ld16(wp + (long)kk * 32) : zero4;
        }
        if (!NORM) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { int kk = kt + 4 * u; kk = kk < nk ? kk : nk - 1; x[u] = ld16(xg + (long)kk * 32); }
        }
        sched_fence();
        if (NORM) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { int kk = kt + 4 * u; kk = kk < nk ? kk : nk - 1; x[u] = ld16(xs + (long)kk * 32); }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u32x4 wu = (kt + 4 * u < nk) ? wc[u] : zero4;        // steps past K contribute zero
            acc = mfma_16x16x32(wu, x[u], acc);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) wc[u] = wn[u];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc[r];
    __syncthreads();
    if (wave != 0) return;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = red[0][lane][r] + red[1][lane][r] + red[2][lane][r] + red[3][lane][r];
    const int m = fr;
    if (ACT) {
        // block columns = [8 gate | 8 up] of features 8*blk .. 8*blk+7: lanes fq<2 own gate, their partners (lane ^ 32) up
        float u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = wave_shfl_xor(v[r], 32);
        if (fq < 2 && m < g.M) {
            const int f0 = (int)blockIdx.x * 8 + 4 * fq;
            float a[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = round_bf(silu_d(round_bf(v[r]))) * round_bf(u[r]);
            bf16_t* op = (bf16_t*)g.out + (long)m * g.ldo + f0;
            u32x2 o; o.x = pack_bf2(a[0], a[1]); o.y = pack_bf2(a[2], a[3]);
            st8(op, o);
        }
        return;
    }
    const int n = n0 + 4 * fq;
    if (m >= g.M || n >= g.N || 4 * fq >= NCOL) return;
    if (OUTF32) {
        float* cp = (float*)g.out + (long)m * g.ldo + n;
        for (int r = 0; r < 4; ++r) if (n + r < g.N) cp[r] = v[r];
    } else {
        if (g.res) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] = round_bf(v[r]) + bf2f(g.res[(long)m * g.ldres + n + r]);
        }
        bf16_t* cp = (bf16_t*)g.out + (long)m * g.ldo + n;
        if (n + 3 < g.N) { u32x2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); st8(cp, o); }
        else for (int r = 0; r < 4; ++r) if (n + r < g.N) cp[r] = f2bf(v[r]);
    }
}

// ---------------------------------------------------------------------------
struct DecAttnArgs {
    const bf16_t* qkv; long ldqkv;        // [B, (Hq + 2 Hkv) * hd]  raw projections of the new token
    const bf16_t* qw; const bf16_t* kw;   // per-head RMSNorm weights [hd]
    const float* cosT; const float* sinT; // [npos, hd/2]
    const int* pos;                       // [B] rotary position of the new token
    bf16_t* kc; bf16_t* vc;               // KV cache [B, Hkv, Smax, hd]
    const uint8_t* kmask;                 // [B, Smax] or null
    float* part_o; float* part_ml;        // [B, Hq, nchunk, hd], [B, Hq, nchunk, 2]
    int B, Hq, Hkv, Smax, cur_len, nchunk;
    float eps, scale;
};

// per-head RMSNorm (optional weight) + rotate-half RoPE of the 8-dim slice this lane owns (dims 8*dl .. 8*dl+7);
// the rotation partner (dims +- hd/2) is lane ^ (LPK/2)
template <int HD>
__device__ __forceinline__ void norm_rope_slice(float (&x)[8], const bf16_t* nw, const float* cosr, const float* sinr,
                                                int dl, float eps) {
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
    const int hsl = (dl & (LPK / 2 - 1)) * 8;        // index into the half-dim cos/sin row
    const bool upper = dl >= LPK / 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float other = wave_shfl_xor(x[i], LPK / 2);
        const float c = cosr[hsl + i], s = sinr[hsl + i];
        x[i] = round_bf(upper ? x[i] * c + other * s : x[i] * c - other * s);
    }
}

template <int HD, int G>
__global__ __launch_bounds__(256) void dec_attn_partial_kernel(DecAttnArgs a) {
    constexpr int CK = 64, LPK = HD / 8, KPI = 64 / LPK, NIT = CK / KPI, GRP = NIT / 2;   // 64 positions per wave
    const int lane = lane_id();
    const int c = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int hkv = (int)blockIdx.y, b = (int)blockIdx.z;
    if (c >= a.nchunk) return;
    const int len = a.cur_len + 1;
    const int s_begin = c * CK;
    int s_end = s_begin + CK;
    s_end = s_end < len ? s_end : len;
    const int kg = lane / LPK, dl = lane % LPK;
    const int Nq = a.Hq * HD, Nkv = a.Hkv * HD;
    // the first half of the chunk's K rows is requested before anything else: its latency hides the q/k prologue
    bf16_t* kb_ = a.kc + ((long)b * a.Hkv + hkv) * a.Smax * HD + dl * 8;
    bf16_t* vb_ = a.vc + ((long)b * a.Hkv + hkv) * a.Smax * HD + dl * 8;
    u32x4 ra[GRP], rb[GRP], va[GRP], vb[GRP];     // all K and V rows of the chunk: one memory round trip
#pragma unroll
    for (int i = 0; i < GRP; ++i) {
        const int k0 = s_begin + i * KPI + kg, k1 = s_begin + (GRP + i) * KPI + kg;
        ra[i] = ld16(kb_ + (long)(k0 < s_end ? k0 : s_end - 1) * HD);
        rb[i] = ld16(kb_ + (long)(k1 < s_end ? k1 : s_end - 1) * HD);
    }
#pragma unroll
    for (int i = 0; i < GRP; ++i) {
        const int k0 = s_begin + i * KPI + kg, k1 = s_begin + (GRP + i) * KPI + kg;
        va[i] = ld16(vb_ + (long)(k0 < s_end ? k0 : s_end - 1) * HD);
        vb[i] = ld16(vb_ + (long)(k1 < s_end ? k1 : s_end - 1) * HD);
    }
    const bf16_t* row = a.qkv + (long)b * a.ldqkv;
    const int p = a.pos[b];
    const float* cosr = a.cosT + (long)p * (HD / 2);
    const float* sinr = a.sinT + (long)p * (HD / 2);
    // ---- queries of the G heads of this kv group
    const float sc = a.scale * kLog2eD;
    float qv[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        unpack8(ld16(row + (long)(hkv * G + g) * HD + dl * 8), qv[g]);
        norm_rope_slice<HD>(qv[g], a.qw, cosr, sinr, dl, a.eps);
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[g][i] *= sc;
    }
    // ---- the new key / value row (needed by the chunk that holds position cur_len; cheap, so every wave computes it)
    float kn[8], vn[8];
    unpack8(ld16(row + Nq + (long)hkv * HD + dl * 8), kn);
    norm_rope_slice<HD>(kn, a.kw, cosr, sinr, dl, a.eps);
    unpack8(ld16(row + Nq + Nkv + (long)hkv * HD + dl * 8), vn);
    const bool owns_new = a.cur_len >= s_begin && a.cur_len < s_begin + CK;
    if (owns_new && kg == 0) {
        st16(kb_ + (long)a.cur_len * HD, pack8(kn));
        st16(vb_ + (long)a.cur_len * HD, pack8(vn));
    }
    const int new_rel = owns_new ? a.cur_len - s_begin : -1;
    // ---- validity bits of the 128 positions
    uint64_t vbits;
    {
        const int key = s_begin + lane;
        const bool okk = key < s_end;
        const uint8_t mraw = a.kmask ? a.kmask[(long)b * a.Smax + (okk ? key : s_end - 1)] : (uint8_t)1;
        vbits = wave_ballot(okk && (mraw != 0 || key == a.cur_len));
    }
    float sco[NIT][G];
    float m[G];
#pragma unroll
    for (int g = 0; g < G; ++g) m[g] = kNegD;
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
#pragma unroll
        for (int i = 0; i < GRP; ++i) {
            const int it = grp * GRP + i;
            const int rel_ = it * KPI + kg;
            const bool ok = (vbits >> rel_) & 1ull;
            float f[8];
            unpack8(grp == 0 ? ra[i] : rb[i], f);
            if (rel_ == new_rel) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = kn[e];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d += f[e] * qv[g][e];
#pragma unroll
                for (int mk = LPK >> 1; mk >= 1; mk >>= 1) d += wave_shfl_xor(d, mk);
                d = ok ? d : kNegD;
                sco[it][g] = d;
                m[g] = fmaxf(m[g], d);
            }
        }
    }
    float l[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int mk = 32; mk >= LPK; mk >>= 1) m[g] = fmaxf(m[g], wave_shfl_xor(m[g], mk));
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    }
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
#pragma unroll
        for (int i = 0; i < GRP; ++i) {
            const int it = grp * GRP + i;
            const int rel_ = it * KPI + kg;
            float f[8];
            unpack8(grp == 0 ? va[i] : vb[i], f);
            if (rel_ == new_rel) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = round_bf(vn[e]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float pr = sco[it][g] > 0.5f * kNegD ? exp2f(sco[it][g] - m[g]) : 0.f;
                l[g] += pr;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][e] += pr * f[e];
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int mk = 32; mk >= LPK; mk >>= 1) {
            l[g] += wave_shfl_xor(l[g], mk);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[g][e] += wave_shfl_xor(acc[g][e], mk);
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

}  // namespace bra

using namespace bra;

extern "C" int bra_dec_gemm(const void* x, long ldx, const void* norm_w, float eps, const void* W, long ldw,
                            const void* res, long ldres, void* out, long ldo, int M, int N, int K, int act, int out_f32,
                            void* stream) {
    if (M <= 0 || M > 16 || N <= 0 || K <= 0 || K % 32 || ldx % 8 || ldw % 8 || !x || !W || !out) return BRA_ERR_ARG;
    if (act && (N % 16 || out_f32 || res)) return BRA_ERR_ARG;
    if (out_f32 && res) return BRA_ERR_ARG;
    DecGemmArgs g = {(const bf16_t*)x, ldx, (const bf16_t*)norm_w, eps, (const bf16_t*)W, ldw, (const bf16_t*)res, ldres, out, ldo, M, N, K};
    bra_stream_t st = (bra_stream_t)stream;
    const dim3 blk(256);
    const bool narrow = !act && (N + 15) / 16 < 256 && N % 8 == 0;        // fewer than one workgroup per CU: 8-column tiles
    const dim3 grid(narrow ? (N + 7) / 8 : (N + 15) / 16);
#define BRA_DG(NORM_, ACT_, F32_, NCOL_, SMEM_)                                                            \
    do {                                                                                                   \
        if (M <= 8) {                                                                                      \
            BRA_ALLOW_SMEM((dec_gemm_kernel<NORM_, ACT_, F32_, NCOL_, 8>), SMEM_);                         \
            BRA_LAUNCH((dec_gemm_kernel<NORM_, ACT_, F32_, NCOL_, 8>), grid, blk, SMEM_, st, g);           \
        } else {                                                                                           \
            BRA_ALLOW_SMEM((dec_gemm_kernel<NORM_, ACT_, F32_, NCOL_, 16>), SMEM_);                        \
            BRA_LAUNCH((dec_gemm_kernel<NORM_, ACT_, F32_, NCOL_, 16>), grid, blk, SMEM_, st, g);          \
        }                                                                                                  \
    } while (0)
    const size_t smem = norm_w ? (size_t)(M <= 8 ? 8 : 16) * (K + 8) * 2 : 0;
    if (norm_w) {
        if (act) BRA_DG(1, 1, 0, 16, smem);
        else if (out_f32) { if (narrow) BRA_DG(1, 0, 1, 8, smem); else BRA_DG(1, 0, 1, 16, smem); }
        else { if (narrow) BRA_DG(1, 0, 0, 8, smem); else BRA_DG(1, 0, 0, 16, smem); }
    } else {
        if (act) BRA_DG(0, 1, 0, 16, smem);
        else if (out_f32) { if (narrow) BRA_DG(0, 0, 1, 8, smem); else BRA_DG(0, 0, 1, 16, smem); }
        else { if (narrow) BRA_DG(0, 0, 0, 8, smem); else BRA_DG(0, 0, 0, 16, smem); }
    }
#undef BRA_DG
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_dec_attn_partial(const void* qkv, long ldqkv, const void* qw, const void* kw, const float* cosT,
                                    const float* sinT, const int* pos, void* kc, void* vc, const void* kmask,
                                    float* part_o, float* part_ml, int B, int Hq, int Hkv, int hd, int Smax, int cur_len,
                                    float eps, float scale, void* stream) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || cur_len < 0 || cur_len >= Smax) return BRA_ERR_ARG;
    if (!qkv || !qw || !kw || !cosT || !sinT || !pos || !kc || !vc || !part_o || !part_ml) return BRA_ERR_ARG;
    const int G = Hq / Hkv;
    DecAttnArgs a = {(const bf16_t*)qkv, ldqkv, (const bf16_t*)qw, (const bf16_t*)kw, cosT, sinT, pos, (bf16_t*)kc, (bf16_t*)vc,
                     (const uint8_t*)kmask, part_o, part_ml, B, Hq, Hkv, Smax, cur_len, (cur_len + 1 + 63) / 64, eps, scale};
    bra_stream_t st = (bra_stream_t)stream;
    dim3 grid((a.nchunk + 3) / 4, Hkv, B);
#define BRA_DA(HD_, G_)                                                                         \
    if (hd == HD_ && G == G_) {                                                                 \
        BRA_LAUNCH((dec_attn_partial_kernel<HD_, G_>), grid, dim3(256), 0, st, a);              \
        return BRA_LAUNCH_STATUS();                                                             \
    }
    BRA_DA(128, 1) BRA_DA(128, 2) BRA_DA(128, 4) BRA_DA(64, 1) BRA_DA(64, 2) BRA_DA(64, 4) BRA_DA(32, 1) BRA_DA(32, 2) BRA_DA(32, 4)
#undef BRA_DA
    return BRA_ERR_UNSUPPORTED;
}
