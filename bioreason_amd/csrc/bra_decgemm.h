// bra_decgemm.h — argument record and tile epilogue of the weight-streaming decode projections (k_decgemm.hip), shared with the
// persistent decode step (k_persist.hip), which runs the same tiles inside one launch.
#pragma once
#include "bra_device.h"
#include "bra_gridsync.h"

namespace bra {

struct DecGemm2Args {
    const bf16_t* x; long ldx;          // [M, K]
    const float* ss_in; int nss_in;     // NORM: partial sums of squares of the rows of x, [8][nss_in] ([16][nss_in] for M > 8)
    const bf16_t* nw; float eps;        // NORM: RMSNorm weight [K]
    const bf16_t* W; long ldw;          // [N, K]
    const bf16_t* res; long ldres;      // [M, N] or null
    void* out; long ldo;                // bf16 [M, N] | bf16 [M, N/2] (ACT) | f32 [M, N]
    float* ss_out; int nss_out;         // partial sums of squares of the bf16 outputs, [8 | 16][nss_out], column = workgroup
    int M, N, K;
    int packed;                         // W is in fragment order (bra_dec_pack_weights): [tile][k-step][lane][8]
    const float* wscale;                // non-null: W is the fp8 e4m3 image of bra_dec_pack_weights_fp8 ([tile][k-step pair][lane][8 + 8 bytes]),
                                        // wscale[n] the fp32 scale of weight row n
    BRA_DBG_FIELD(unsigned long long* probe;)   // BRA_DEBUG only: timing probe (tools/dec_overhead_probe.py), 8 stamps per probed workgroup
    float inv_K;                        // 1 / K rounded on the host (NORM == 2: mean of squares = fma(sum, inv_K, eps))
};

// 100 MHz wall clock (s_memrealtime); compiled in only under BRA_DEBUG (one uniform branch per stamp when unused)
#if defined(BRA_EMU) || !defined(BRA_DEBUG)
__device__ __forceinline__ void dg2_stamp(const DecGemm2Args&, int) {}
#else
__device__ __forceinline__ void dg2_stamp(const DecGemm2Args& g, int slot) {
    if (g.probe && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))
        g.probe[(blockIdx.x == 0 ? 0 : 8) + slot] = __builtin_amdgcn_s_memrealtime();
}
#endif

__device__ __forceinline__ float silu_g(float x) { return x / (1.f + __expf(-x)); }

// RMSNorm row factor from the statistics partials one lane folds (NORM == 2): the eight chunks in a FIXED association and one
// explicit fma for mean + eps — under -ffast-math the optimiser is otherwise free to re-associate the sum and to form (or not
// form) the fma differently in every kernel this is inlined into, and the launched and the persistent step would disagree in the
// last bit of rstd (which an occasional bf16 rounding downstream then turns into a visible difference).  inv_K = 1 / K as a
// float rounded on the host (v_rcp_f32 of a run-time K and the constant-folded 1 / K of a templated K are not the same number).
__device__ __forceinline__ float dg2_fold_rstd(const f32x4 (&sq)[8], const int per, const float inv_K, const float eps) {
#pragma clang fp reassociate(off)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 q = sq[i];
        const float lo = q[0] + q[1], hi = q[2] + q[3];
        const float c = lo + hi;
        s = s + (4 * i < per ? c : 0.f);
    }
    s = s + wave_shfl_xor(s, 1);
    s = s + wave_shfl_xor(s, 2);
    s = s + wave_shfl_xor(s, 4);
    return rsqrtf(__builtin_fmaf(s, inv_K, eps));
}

// full-rate 24-bit multiply (v_mul_u32_u24; the 32- and 64-bit integer multiplies run at quarter rate): row index x leading dimension
#ifdef BRA_EMU
__device__ __forceinline__ unsigned dg2_mul24(int a, int b) { return ((unsigned)a & 0xffffffu) * ((unsigned)b & 0xffffffu); }
#else
__device__ __forceinline__ unsigned dg2_mul24(int a, int b) { return __umul24((unsigned)a, (unsigned)b); }
#endif

// K-reduction of one column tile: the NW per-wave partial products of a lane in wave order — a FIXED association (see
// dg2_fold_rstd: a re-associated sum differs in the last bit between the kernels this is inlined into)
template <int NW>
__device__ __forceinline__ void dg2_reduce(const float (*slab)[64][4], const int lane, float (&v)[4]) {
#pragma clang fp reassociate(off)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) sum = sum + slab[wv][lane][r];
        v[r] = sum;
    }
}

// epilogue of one column tile, executed by ONE wave on the K-reduced products v (lane: batch row fr, columns 4 fq .. + 3)
// FULLN: N is a multiple of the tile width (always so with packed weights): no ragged last tile, every column test folds away
// XS (bra_gridsync.h): 1 = out / res / ss_out are exchanged with other workgroups of the SAME launch (persistent decode step): sc1
// accesses through (uniform base, 32-bit byte offset); the arithmetic is unchanged
template <int MODE, int ACT, int OUTF32, int FULLN = 0, int XS = 0>
__device__ __forceinline__ void dg2_epilogue(const DecGemm2Args& g, float (&v)[4], int tile, int lane, bool have_res,
                                             const u32x2& resv) {
#pragma clang fp reassociate(off)
    constexpr int NCOL = MODE ? 8 : 16;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = tile * NCOL;
    const int N = FULLN ? 0x7fffffff : g.N;       // (column tests below read `N`)
    if (MODE) {                                   // second K-half of (n, m) sits at (n + 8, m + 8) = lane + 40
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += wave_shfl(v[r], lane + 40);
    }
    const int m = fr;
    if (ACT) {
        // tile rows = [8 gate | 8 up] of features 8*tile .. 8*tile+7: lanes fq < 2 own gate, partners (lane ^ 32) up
        float up[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) up[r] = wave_shfl_xor(v[r], 32);
        if (fq < 2 && m < g.M) {
            const int f0 = tile * 8 + 4 * fq;
            float a[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = round_bf(silu_g(round_bf(v[r]))) * round_bf(up[r]);
            u32x2 o; o.x = pack_bf2(a[0], a[1]); o.y = pack_bf2(a[2], a[3]);
            if constexpr (XS != 0) xst8<1>(g.out, (dg2_mul24(m, (int)g.ldo) + (unsigned)f0) * 2u, o);
            else st8((bf16_t*)g.out + (long)m * g.ldo + f0, o);
        }
        return;
    }
    const int n = n0 + 4 * fq;
    const bool live = m < g.M && n < N && 4 * fq < NCOL && (!MODE || fr < 8);
    if (OUTF32) {
        if (live) {
            float* cp = (float*)g.out + (long)m * g.ldo + n;
            if (n + 3 < N && (g.ldo & 3) == 0) { f32x4 o = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(cp) = o; }
            else for (int r = 0; r < 4; ++r) if (n + r < N) cp[r] = v[r];
        }
        if (g.ss_out) {
            // fp32 logits: `ss_out` [M][nss_out] receives the MAXIMUM of every 16-column tile instead of statistics — the sampler
            // (k_grpo.hip: sample_tiles_kernel) finds the top-k tiles among V / 16 maxima and only then touches their logits
            float mx = -3.0e38f;
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = (live && n + r < N) ? fmaxf(mx, v[r]) : mx;
            mx = fmaxf(mx, wave_shfl_xor(mx, 16));
            mx = fmaxf(mx, wave_shfl_xor(mx, 32));
            if (fq == 0 && m < g.M) g.ss_out[(long)m * g.nss_out + tile] = mx;
        }
        return;
    }
    float ss = 0.f;
    if (g.res) {
        if ((have_res || live) && n + 3 < N) {
            u32x2 rr = resv;                              // first tile of the workgroup: requested with the weights
            if (!have_res) {                                                   // later tiles (host: ldres % 4 == 0)
                if constexpr (XS != 0) rr = xld8<1>(g.res, (dg2_mul24(m, (int)g.ldres) + (unsigned)n) * 2u);
                else rr = ld8(g.res + (long)m * g.ldres + n);
            }
            v[0] = round_bf(v[0]) + bf_lo(rr.x); v[1] = round_bf(v[1]) + bf_hi(rr.x);
            v[2] = round_bf(v[2]) + bf_lo(rr.y); v[3] = round_bf(v[3]) + bf_hi(rr.y);
        } else if (live) {
            for (int r = 0; r < 4; ++r) if (n + r < N) v[r] = round_bf(v[r]) + bf2f(g.res[(long)m * g.ldres + n + r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = round_bf(v[r]); if (live && n + r < N) ss += v[r] * v[r]; }
    if (live) {
        bf16_t* cp = (bf16_t*)g.out + (long)m * g.ldo + n;
        if (n + 3 < N) {
            u32x2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
            if constexpr (XS != 0) xst8<1>(g.out, (dg2_mul24(m, (int)g.ldo) + (unsigned)n) * 2u, o);
            else st8(cp, o);
        }
        else for (int r = 0; r < 4; ++r) if (n + r < N) cp[r] = f2bf(v[r]);
    }
    if (g.ss_out) {
        if (!live) ss = 0.f;
        ss += wave_shfl_xor(ss, 16);
        ss += wave_shfl_xor(ss, 32);
        if (fq == 0 && (!MODE || fr < 8) && m < g.M) {
            if constexpr (XS != 0) xst4f<1>(g.ss_out, (dg2_mul24(m, g.nss_out) + (unsigned)tile) * 4u, ss);
            else g.ss_out[(long)m * g.nss_out + tile] = ss;
        }
    }
}

}  // namespace bra
