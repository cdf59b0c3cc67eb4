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
// The dec_gemm kernels of this file are the FIRST generation: the step functions use k_decgemm.hip's dec_gemm2 for
// M <= 8 rows (norm statistics as epilogue partials, all weight bytes requested up front) and fall back to these for
// 9..16 rows or hidden sizes with more than 256 column workgroups.  The attention kernels here are current:
// dec_attn_both = dec_attn_shared (one MFMA pass over the prompt K / V^T shared by the rollouts of a prompt) +
// dec_attn_partial (per-sequence completion keys) in one launch.
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

#ifdef BRA_EMU
__device__ __forceinline__ void da_stamp(unsigned long long*, int) {}
#else
__device__ __forceinline__ void da_stamp(unsigned long long* p, int slot) {
    if (p && lane_id() == 0) p[slot] = __builtin_amdgcn_s_memrealtime();
}
#endif

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
    int chunk_off, nchunk_tot;             // slot of this call's chunks inside the partial buffers (shared-prefix split)
    const int* t_ptr;                      // optional device-side cur_len (graph replay: the launch arguments stay constant)
    BRA_DBG_FIELD(unsigned long long* probe;)
    const float* rope_rows;                // optional [B, hd]: cos | sin of each sequence's CURRENT position (bra_rope_rows): no pos -> table hop
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
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cosr + hsl), c1 = *reinterpret_cast<const f32x4*>(cosr + hsl + 4);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sinr + hsl), s1 = *reinterpret_cast<const f32x4*>(sinr + hsl + 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float other = wave_shfl_xor(x[i], LPK / 2);
        const float c = i < 4 ? c0[i & 3] : c1[i & 3], s = i < 4 ? s0[i & 3] : s1[i & 3];
        x[i] = round_bf(upper ? x[i] * c + other * s : x[i] * c - other * s);
    }
}

// GT = q heads per kv head of the model; G = q heads THIS wave processes: GT (hsel = kv head) or 1 (hsel = q head: the VALU
// dot products of this body are the long pole of the launch, so the both-kernel gives every q head its own wave)
template <int HD, int GT, int G = GT>
__device__ __forceinline__ void dec_attn_partial_body(const DecAttnArgs& a, const int c, const int hsel, const int b) {
    constexpr int CK = 64, LPK = HD / 8, KPI = 64 / LPK, NIT = CK / KPI, GRP = NIT / 2;   // 64 positions per wave
    const int hkv = G == GT ? hsel : hsel / GT;
    const int hq0 = G == GT ? hkv * GT : hsel;
    const int lane = lane_id();
#ifdef BRA_DEBUG
    unsigned long long* pr = (a.probe && c == 0 && hkv == 0 && b == 0) ? a.probe + 8 : nullptr;
#else
    constexpr unsigned long long* pr = nullptr;
#endif
    da_stamp(pr, 0);
    int cur_len = a.cur_len, nchunk = a.nchunk, nchunk_tot = a.nchunk_tot;
    if (a.t_ptr) { cur_len = a.t_ptr[0]; nchunk = (cur_len + 64) / 64; nchunk_tot = a.chunk_off + nchunk; }
    if (c >= nchunk) return;
    const int len = cur_len + 1;
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
    const float* cosr; const float* sinr;
    if (a.rope_rows) { cosr = a.rope_rows + (long)b * HD; sinr = cosr + HD / 2; }
    else { const int p = a.pos[b]; cosr = a.cosT + (long)p * (HD / 2); sinr = a.sinT + (long)p * (HD / 2); }
    // ---- queries of the G heads of this kv group
    const float sc = a.scale * kLog2eD;
    float qv[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        unpack8(ld16(row + (long)(hq0 + g) * HD + dl * 8), qv[g]);
        norm_rope_slice<HD>(qv[g], a.qw, cosr, sinr, dl, a.eps);
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[g][i] *= sc;
    }
    // ---- the new key / value row (needed by the chunk that holds position cur_len; cheap, so every wave computes it)
    float kn[8], vn[8];
    unpack8(ld16(row + Nq + (long)hkv * HD + dl * 8), kn);
    norm_rope_slice<HD>(kn, a.kw, cosr, sinr, dl, a.eps);
    unpack8(ld16(row + Nq + Nkv + (long)hkv * HD + dl * 8), vn);
    const bool owns_new = cur_len >= s_begin && cur_len < s_begin + CK;
    if (owns_new && kg == 0 && hq0 == hkv * GT) {
        st16(kb_ + (long)cur_len * HD, pack8(kn));
        st16(vb_ + (long)cur_len * HD, pack8(vn));
    }
    const int new_rel = owns_new ? cur_len - s_begin : -1;
    // ---- validity bits of the 128 positions
    uint64_t vbits;
    {
        const int key = s_begin + lane;
        const bool okk = key < s_end;
        const uint8_t mraw = a.kmask ? a.kmask[(long)b * a.Smax + (okk ? key : s_end - 1)] : (uint8_t)1;
        vbits = wave_ballot(okk && (mraw != 0 || key == cur_len));
    }
    da_stamp(pr, 1);                                   // q / k prologue done, validity bits known
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
    da_stamp(pr, 2);                                   // scores
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
        const int hq = hq0 + g;
        const long base = ((long)b * a.Hq + hq) * nchunk_tot + a.chunk_off + c;
        if (kg == 0) {
            f32x4 lo = {acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
            f32x4 hi = {acc[g][4], acc[g][5], acc[g][6], acc[g][7]};
            *reinterpret_cast<f32x4*>(a.part_o + base * HD + dl * 8) = lo;
            *reinterpret_cast<f32x4*>(a.part_o + base * HD + dl * 8 + 4) = hi;
        }
        if (lane == 0) { a.part_ml[base * 2] = m[g]; a.part_ml[base * 2 + 1] = l[g]; }
    }
    da_stamp(pr, 3);
}

template <int HD, int G>
__global__ __launch_bounds__(256) void dec_attn_partial_kernel(DecAttnArgs a) {
    dec_attn_partial_body<HD, G>(a, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), (int)blockIdx.y, (int)blockIdx.z);
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
                                    float eps, float scale, int chunk_off, int nchunk_tot, const int* t_dev, void* stream) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || cur_len < 0 || cur_len >= Smax) return BRA_ERR_ARG;
    if (!qkv || !qw || !kw || !cosT || !sinT || !pos || !kc || !vc || !part_o || !part_ml) return BRA_ERR_ARG;
    const int G = Hq / Hkv;
    DecAttnArgs a = {(const bf16_t*)qkv, ldqkv, (const bf16_t*)qw, (const bf16_t*)kw, cosT, sinT, pos, (bf16_t*)kc, (bf16_t*)vc,
                     (const uint8_t*)kmask, part_o, part_ml, B, Hq, Hkv, Smax, cur_len, (cur_len + 1 + 63) / 64, eps, scale, chunk_off,
                     nchunk_tot > 0 ? nchunk_tot : (cur_len + 1 + 63) / 64, t_dev, BRA_DBG_INIT(nullptr) nullptr};
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

// ---------------------------------------------------------------------------
// Shared-prefix decode attention.  GRPO decodes G copies of ONE prompt together (grpo_trainer.py:107-116): their
// prompt K/V rows are identical, so the prompt part of the attention of all copies is one small matrix product per
// kv-head instead of G passes over the same cache:  S[key][row] = K_p[key][:] . Q[row][:]  with
// row = (copy, q-head of the group) <= 16 rows  ->  v_mfma_f32_16x16x32_bf16 with the K rows streamed straight into
// the A fragment;  O^T[d][row] += V_p^T[d][key] . P[key][row]  with the prompt's V^T image (kept from the prefill).
// One wave per (prompt, kv-head, 64-key chunk); the per-copy completion keys go through dec_attn_partial_kernel on the
// completion cache; both write (max, sum, O) partials that attn_decode_merge_kernel combines.
namespace bra {

struct DecSharedArgs {
    const bf16_t* qkv; long ldqkv;        // [B, (Hq + 2 Hkv) * hd] raw projections of the new token, B = R * copies
    const bf16_t* qw;                     // q_norm weight [hd]
    const float* cosT; const float* sinT; // [npos, hd/2]
    const int* pos;                       // [B]
    const bf16_t* kp; long kp_sr, kp_sh, kp_ss;     // prompt K: element strides over (prompt, kv-head, position)
    const bf16_t* vtp; long vt_sr, vt_sh, vt_sd;    // prompt V^T [R, Hkv, hd, pitch]
    const uint8_t* pmask;                 // [R, P] validity of prompt positions (left padding) or null
    float* part_o; float* part_ml;
    int R, copies, Hq, Hkv, P, nchunk_tot;
    float eps, scale;
    const int* t_ptr;                     // optional device-side completion index t: nchunk_tot = ceil(P/64) + ceil((t+1)/64)
    BRA_DBG_FIELD(unsigned long long* probe;)   // BRA_DEBUG only (bra_debug_set_probe): 100 MHz stamps of one prompt-part and one completion-part wave
    const float* rope_rows;               // optional [B, hd]: cos | sin of each sequence's current position
};


template <int HD, int G>
__device__ __forceinline__ void dec_attn_shared_body(const DecSharedArgs& a, const int c, const int hkv, const int r) {
    constexpr int DS = HD / 32;           // 32-deep contraction steps over the head dim
    constexpr int DB = HD / 16;           // 16-wide output blocks over the head dim
    const int lane = lane_id();
#ifdef BRA_DEBUG
    unsigned long long* pr = (a.probe && c == 0 && hkv == 0 && r == 0) ? a.probe : nullptr;
#else
    constexpr unsigned long long* pr = nullptr;
#endif
    da_stamp(pr, 0);
    const int fr = lane & 15, fq = lane >> 4;
    const int s0 = c * 64;
    const int rows = a.copies * G;                         // live query rows (<= 16)
    const int qrow = fr < rows ? fr : rows - 1;
    const int copy = qrow / G, g = qrow % G;
    const int b = r * a.copies + copy, hq = hkv * G + g;
    // ---- K rows of the chunk, straight into MFMA A fragments: lane (fr, fq) holds K[key][32 s + 8 fq .. +8]
    const bf16_t* kbase = a.kp + r * a.kp_sr + hkv * a.kp_sh;
    u32x4 kf[4][DS];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        int key = s0 + kb * 16 + fr; key = key < a.P ? key : a.P - 1;
#pragma unroll
        for (int s = 0; s < DS; ++s) kf[kb][s] = ld16(kbase + (long)key * a.kp_ss + s * 32 + fq * 8);
    }
    // ---- V^T fragments: lane (fr = d within block, fq) holds the 4+4 keys {32 kk + 4 fq + j, 32 kk + 16 + 4 fq + j}
    const bf16_t* vbase = a.vtp + r * a.vt_sr + hkv * a.vt_sh;
    u32x4 vf[DB][2];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16_t* vp = vbase + (long)(db * 16 + fr) * a.vt_sd + s0 + kk * 32 + fq * 4;
            const u32x2 lo = ld8(vp), hi = ld8(vp + 16);
            vf[db][kk].x = lo.x; vf[db][kk].y = lo.y; vf[db][kk].z = hi.x; vf[db][kk].w = hi.y;
        }
    // ---- query row: this lane's four 8-dim slices {32 s + 8 fq}; the rotation partner of slice s is slice s ^ (DS/2)
    const bf16_t* qp = a.qkv + (long)b * a.ldqkv + (long)hq * HD;
    float qv[DS][8];
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        unpack8(ld16(qp + s * 32 + fq * 8), qv[s]);
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += qv[s][i] * qv[s][i];
    }
    da_stamp(pr, 1);                                   // every request issued
    ss += wave_shfl_xor(ss, 16);
    ss += wave_shfl_xor(ss, 32);
    const float rstd = rsqrtf(ss / (float)HD + a.eps);
    da_stamp(pr, 2);                                   // q arrived
    const float* cosr; const float* sinr;
    if (a.rope_rows) { cosr = a.rope_rows + (long)b * HD; sinr = cosr + HD / 2; }
    else { const int p = a.pos[b]; cosr = a.cosT + (long)p * (HD / 2); sinr = a.sinT + (long)p * (HD / 2); }
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        float w[8];
        unpack8(ld16(a.qw + s * 32 + fq * 8), w);
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[s][i] = round_bf(w[i] * round_bf(qv[s][i] * rstd));
    }
    const float sc = a.scale * kLog2eD;
    u32x4 qf[DS];
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        const bool upper = s >= DS / 2;
        const int sp = s ^ (DS / 2);
        float o[8];
        const int hbase = (s & (DS / 2 - 1)) * 32 + fq * 8;            // index into the half-dim cos / sin row
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cosr + hbase), c1 = *reinterpret_cast<const f32x4*>(cosr + hbase + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sinr + hbase), s1 = *reinterpret_cast<const f32x4*>(sinr + hbase + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float cv = i < 4 ? c0[i & 3] : c1[i & 3], sv = i < 4 ? s0[i & 3] : s1[i & 3];
            const float rot = upper ? qv[s][i] * cv + qv[sp][i] * sv : qv[s][i] * cv - qv[sp][i] * sv;
            o[i] = round_bf(rot) * sc;
        }
        qf[s] = pack8(o);
    }
    da_stamp(pr, 3);                                   // q normalised + rotated (pos -> cos / sin round trips)
    // ---- validity of the 64 prompt positions of this chunk
    uint64_t vbits;
    {
        const int key = s0 + lane;
        const bool okk = key < a.P;
        const uint8_t mb = a.pmask ? a.pmask[(long)r * a.P + (okk ? key : a.P - 1)] : (uint8_t)1;
        vbits = wave_ballot(okk && mb != 0);
    }
    // ---- scores: D[key = 4 fq + j][row = fr] per 16-key block
    f32x4 sreg[4];
    float m = kNegD;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < DS; ++s) acc = mfma_16x16x32(kf[kb][s], qf[s], acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rel_ = kb * 16 + 4 * fq + j;
            const bool ok = (vbits >> rel_) & 1ull;
            acc[j] = ok ? acc[j] : kNegD;
            m = fmaxf(m, acc[j]);
        }
        sreg[kb] = acc;
    }
    m = fmaxf(m, wave_shfl_xor(m, 16));
    m = fmaxf(m, wave_shfl_xor(m, 32));
    da_stamp(pr, 4);                                   // scores (K arrived)
    float l = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pe = sreg[kb][j] > 0.5f * kNegD ? exp2f(sreg[kb][j] - m) : 0.f;
            sreg[kb][j] = pe;
            l += pe;
        }
    l += wave_shfl_xor(l, 16);
    l += wave_shfl_xor(l, 32);
    // ---- O^T[d][row] += V^T . P : k-slots of lane group fq <-> keys {32 kk + 4 fq + j} U {32 kk + 16 + 4 fq + j}
    const int ntot = a.t_ptr ? (a.P + 63) / 64 + (a.t_ptr[0] + 64) / 64 : a.nchunk_tot;
    const long base = ((long)b * a.Hq + hq) * ntot + c;
    const bool live = fr < rows;
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
        if (live) *reinterpret_cast<f32x4*>(a.part_o + base * HD + db * 16 + 4 * fq) = acc;   // d = 16 db + 4 fq + j
    }
    if (live && fq == 0) { a.part_ml[base * 2] = m; a.part_ml[base * 2 + 1] = l; }
    da_stamp(pr, 5);                                   // partials issued (V^T arrived)
}

template <int HD, int G>
__global__ __launch_bounds__(64) void dec_attn_shared_kernel(DecSharedArgs a) {
    dec_attn_shared_body<HD, G>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// both halves of the shared-prefix decode attention in ONE launch: single-wave workgroups [0, nsh) take the
// (prompt, kv-head, prompt chunk) items, the rest the (sequence, kv-head, completion chunk) items; they are independent
// (disjoint partial slots), so nothing orders them and a layer saves one dependent launch
template <int HD, int G>
__global__ __launch_bounds__(64) void dec_attn_both_kernel(DecSharedArgs s, DecAttnArgs a, int npc, int ncc) {
    const int w = (int)blockIdx.x;
    const int nsh = npc * s.Hkv * s.R;
    if (w < nsh) {
        dec_attn_shared_body<HD, G>(s, w % npc, (w / npc) % s.Hkv, w / (npc * s.Hkv));
    } else {
        const int v = w - nsh;                 // (sequence, q head, completion chunk): one q head per wave
        dec_attn_partial_body<HD, G, 1>(a, v % ncc, (v / ncc) % a.Hq, v / (ncc * a.Hq));
    }
}

}  // namespace bra

extern "C" int bra_dec_attn_shared(const void* qkv, long ldqkv, const void* qw, const float* cosT, const float* sinT,
                                   const int* pos, const void* kp, long kp_sr, long kp_sh, long kp_ss, const void* vtp,
                                   long vt_sr, long vt_sh, long vt_sd, const void* pmask, float* part_o, float* part_ml,
                                   int R, int copies, int Hq, int Hkv, int hd, int P, int nchunk_tot, float eps,
                                   float scale, const int* t_dev, void* stream) {
    if (R <= 0 || copies <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || P <= 0) return BRA_ERR_ARG;
    const int G = Hq / Hkv;
    if (copies * G > 16) return BRA_ERR_UNSUPPORTED;
    if (!qkv || !qw || !cosT || !sinT || !pos || !kp || !vtp || !part_o || !part_ml) return BRA_ERR_ARG;
    if (vt_sd < ((P + 63) / 64) * 64 || vt_sd % 4 || kp_ss % 8) return BRA_ERR_ARG;
    DecSharedArgs a = {(const bf16_t*)qkv, ldqkv, (const bf16_t*)qw, cosT, sinT, pos, (const bf16_t*)kp, kp_sr, kp_sh, kp_ss,
                       (const bf16_t*)vtp, vt_sr, vt_sh, vt_sd, (const uint8_t*)pmask, part_o, part_ml, R, copies, Hq, Hkv, P,
                       nchunk_tot, eps, scale, t_dev, BRA_DBG_INIT(nullptr) nullptr};
    bra_stream_t st = (bra_stream_t)stream;
    dim3 grid((P + 63) / 64, Hkv, R);
#define BRA_DS(HD_, G_)                                                                         \
    if (hd == HD_ && G == G_) {                                                                 \
        BRA_LAUNCH((dec_attn_shared_kernel<HD_, G_>), grid, dim3(64), 0, st, a);                \
        return BRA_LAUNCH_STATUS();                                                             \
    }
    BRA_DS(128, 1) BRA_DS(128, 2) BRA_DS(128, 4) BRA_DS(64, 1) BRA_DS(64, 2) BRA_DS(64, 4)   // hd 32: per-copy path
#undef BRA_DS
    return BRA_ERR_UNSUPPORTED;
}

#ifdef BRA_DEBUG
#ifdef BRA_EMU
static unsigned long long* g_debug_probe = nullptr;
#else
static std::atomic<unsigned long long*> g_debug_probe{nullptr};        // diagnostics knob (include/bioreason_hip_debug.h)
#endif
extern "C" int bra_debug_set_probe(void* p) { g_debug_probe = (unsigned long long*)p; return 0; }
#endif

extern "C" int bra_dec_attn_both(const void* qkv, long ldqkv, const void* qw, const void* kw, const float* cosT,
                                 const float* sinT, const int* pos, const void* kp, long kp_sr, long kp_sh, long kp_ss,
                                 const void* vtp, long vt_sr, long vt_sh, long vt_sd, const void* pmask, void* kc, void* vc,
                                 float* part_o, float* part_ml, int R, int copies, int Hq, int Hkv, int hd, int P, int C,
                                 int t, float eps, float scale, const int* t_dev, const float* rope_rows, void* stream) {
    if (R <= 0 || copies <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || P <= 0 || t < 0 || t >= C) return BRA_ERR_ARG;
    const int G = Hq / Hkv, B = R * copies;
    if (copies * G > 16) return BRA_ERR_UNSUPPORTED;
    if (!qkv || !qw || !kw || !cosT || !sinT || !pos || !kp || !vtp || !kc || !vc || !part_o || !part_ml) return BRA_ERR_ARG;
    if (vt_sd < ((P + 63) / 64) * 64 || vt_sd % 4 || kp_ss % 8) return BRA_ERR_ARG;
    const int npc = (P + 63) / 64, ncc = (t + 64) / 64, ntot = npc + ncc;
    DecSharedArgs s = {(const bf16_t*)qkv, ldqkv, (const bf16_t*)qw, cosT, sinT, pos, (const bf16_t*)kp, kp_sr, kp_sh, kp_ss,
                       (const bf16_t*)vtp, vt_sr, vt_sh, vt_sd, (const uint8_t*)pmask, part_o, part_ml, R, copies, Hq, Hkv, P,
                       ntot, eps, scale, t_dev, BRA_DBG_INIT((unsigned long long*)g_debug_probe) rope_rows};
    DecAttnArgs a = {(const bf16_t*)qkv, ldqkv, (const bf16_t*)qw, (const bf16_t*)kw, cosT, sinT, pos, (bf16_t*)kc,
                     (bf16_t*)vc, nullptr, part_o, part_ml, B, Hq, Hkv, C, t, ncc, eps, scale, npc, ntot, t_dev, BRA_DBG_INIT((unsigned long long*)g_debug_probe) rope_rows};
    bra_stream_t st = (bra_stream_t)stream;
    const dim3 grid(npc * Hkv * R + ncc * Hq * B);
#define BRA_DB(HD_, G_)                                                                         \
    if (hd == HD_ && G == G_) {                                                                 \
        BRA_LAUNCH((dec_attn_both_kernel<HD_, G_>), grid, dim3(64), 0, st, s, a, npc, ncc);     \
        return BRA_LAUNCH_STATUS();                                                             \
    }
    BRA_DB(128, 1) BRA_DB(128, 2) BRA_DB(128, 4) BRA_DB(64, 1) BRA_DB(64, 2) BRA_DB(64, 4)
#undef BRA_DB
    return BRA_ERR_UNSUPPORTED;
}
