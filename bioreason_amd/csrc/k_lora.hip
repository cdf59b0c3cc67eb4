// k_lora.hip — the LoRA branch under training-mode dropout (PEFT: y += (alpha/r) * B(A(dropout(x))), lora_dropout = 0.05,
// train_dna_qwen.py:155-167, reason.py:266,376-388).  PEFT gives every target module its own nn.Dropout, so the q / k / v
// (and gate / up) adapters that share an input x see INDEPENDENT masks; the fused projections here therefore carry one
// mask stream per 32-column rank block.  Masks are never stored: keep(seed, m, k) is a counter-based hash of the element
// index, recomputed wherever the masked operand is needed —
//   lora_down_drop   t[m, r]   = s * sum_k keep_j(m,k)/(1-p) x[m,k] A[r,k]            (forward, replaces the x A^T GEMM)
//   lora_up_drop     dxl[m, k] = sum_j keep_j(m,k)/(1-p) sum_{r in j} dts[m,r] A[r,k]  (backward, the branch's input gradient)
//   wgrad_tn (DROP)  dA[r, k] += sum_m dts[m,r] keep_j(m,k)/(1-p) x[m,k]               (k_wgrad.hip)
// with j = r / 32.  torch's dropout scales in fp32 and rounds once to bf16; so does drop_apply8.
#include "bra_device.h"
#include "bra_api_internal.h"
#include "bra_dropout.h"

namespace bra {

struct LoraDownArgs {
    const bf16_t* x; long ldx;      // [M, K]
    const bf16_t* A; long lda;      // [R, K]
    bf16_t* t; long ldt;            // [M, R]
    int M, K, R;
    float alpha;
    DropCfg d;
    int nb_live;                    // rank blocks that belong to a target module; the padding blocks behind them produce zeros
    int steps_per;                  // K steps (of 128) per workgroup along gridDim.y (split-K form), nstep when gridDim.y == 1
    float* part;                    // split-K form: fp32 partial tiles [gridDim.y][M][R] (unscaled) instead of t
};

// workgroup = 32 rows of x against all R = 32 RB adapter rows; K in steps of 128 through LDS, the four waves take two of the
// eight 16-wide slices of every step each (M / 32 workgroups; the four partial tiles meet in LDS at the end).  The kernel is
// bound by the latency of its one-step-ahead prefetch, not by bytes or VALU work (tools/lora_bench.py): a step therefore
// carries 8 KB of x + 8 NL KB of A per workgroup, twice what the two-wave form had in flight.
// RB = rank blocks of the (padded) output, NL <= RB of them live: both compile-time, so that no branch sits between the
// accumulators and their MFMAs (a run-time block count moved them between AGPRs and VGPRs around every K step)
template <int RB, int NL>
__global__ __launch_bounds__(256) void lora_down_drop_kernel(LoraDownArgs g) {
    constexpr int KS = 128, XP = KS + 8;
    constexpr int XS_ELEMS = 2 * 32 * XP, AS_ELEMS = 2 * 32 * NL * XP;
    constexpr int RED_BYTES = 3 * NL * 64 * 16 * 4, STAGE_BYTES = (XS_ELEMS + AS_ELEMS) * 2;
    __shared__ __attribute__((aligned(16))) char lds[STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES];
    bf16_t* xs = reinterpret_cast<bf16_t*>(lds);                       // [2][32 * XP]
    bf16_t* as = xs + XS_ELEMS;                                        // [2][32 * NL * XP]
    float* red = reinterpret_cast<float*>(lds);                        // [3][NL][64][16], after the K loop
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int m0 = (int)blockIdx.x * 32;
    const int nstep = (g.K + KS - 1) / KS;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    // issue = requests only (raw values), commit zeroes what lies past K / R: a select on the loaded value inside issue, with
    // issue under `if (s + 1 < nstep)`, made the compiler wait for the next tile before the MFMAs of the current one
    u32x4 rx[2], ra[2 * NL];
    auto issue = [&](int s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 4, k = KS * s + 8 * (c & 15);
            int m = m0 + row; m = m < g.M ? m : g.M - 1;
            rx[i] = ld16(g.x + (long)m * g.ldx + (k < g.K ? k : 0));
        }
#pragma unroll
        for (int i = 0; i < 2 * NL; ++i) {                       // rows of the live blocks only
            const int c = tid + 256 * i, row = c >> 4, k = KS * s + 8 * (c & 15);
            ra[i] = ld16(g.A + (long)(row < g.R ? row : g.R - 1) * g.lda + (k < g.K ? k : 0));
        }
    };
    auto commit = [&](int buf, int s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, k = KS * s + 8 * (c & 15);
            st16(&xs[buf * 32 * XP + (c >> 4) * XP + 8 * (c & 15)], k < g.K ? rx[i] : zero4);
        }
#pragma unroll
        for (int i = 0; i < 2 * NL; ++i) {
            const int c = tid + 256 * i, row = c >> 4, k = KS * s + 8 * (c & 15);
            st16(&as[buf * 32 * NL * XP + (c >> 4) * XP + 8 * (c & 15)], (k < g.K && row < g.R) ? ra[i] : zero4);
        }
    };
    f32x16 acc[NL];
#pragma unroll
    for (int rb = 0; rb < NL; ++rb)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[rb][q] = 0.f;
    // split-K form (small M: M / 32 workgroups cannot fill the chip, and each would read all of A): gridDim.y workgroups per row
    // block take `steps_per` K steps each and leave fp32 partial tiles that lora_partial_reduce_kernel sums in a fixed order
    const int s_lo = (int)blockIdx.y * g.steps_per;
    const int s_hi = s_lo + g.steps_per < nstep ? s_lo + g.steps_per : nstep;
    issue(s_lo); commit(s_lo & 1, s_lo);
    __syncthreads();
    const int mrow = m0 + (lane & 31);                           // this lane's row of x (A-operand row)
    for (int s = s_lo; s < s_hi; ++s) {
        const int buf = s & 1;
        // requests without a branch around them (the last step re-requests its own tile and drops it); the fragment reads are
        // tied behind the fence through `lrow` so that the requests are not sunk to the commit behind the MFMAs
        issue(s + 1 < s_hi ? s + 1 : s);
        sched_fence();
        const int lrow = opaque_i(lane & 31);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int kk = 2 * wave + k2;
            const u32x4 xf = ld16(&xs[buf * 32 * XP + lrow * XP + 16 * kk + 8 * h]);
            const uint32_t e0 = (uint32_t)mrow * (uint32_t)g.K + (uint32_t)(KS * s + 16 * kk + 8 * h);
#pragma unroll
            for (int rb = 0; rb < NL; ++rb) {
                const u32x4 af = drop_apply8(xf, g.d.seed[rb], e0, g.d.thr16, g.d.inv_keep);
                const u32x4 bf = ld16(&as[buf * 32 * NL * XP + (32 * rb + lrow) * XP + 16 * kk + 8 * h]);
                acc[rb] = mfma_32x32x16(af, bf, acc[rb]);
            }
        }
        if (s + 1 < s_hi) commit(buf ^ 1, s + 1);
        __syncthreads();
    }
    if (wave != 0) {
#pragma unroll
        for (int rb = 0; rb < NL; ++rb)
#pragma unroll
            for (int q = 0; q < 16; ++q) red[(((wave - 1) * NL + rb) * 64 + lane) * 16 + q] = acc[rb][q];
    }
    __syncthreads();
    if (wave != 0) return;
    // one lane-dependent base per output (a 64-bit multiply each, once), then only wave-uniform offsets per store: the 128 stores of a
    // four-block tile used to carry three quarter-rate multiplies EACH (round 5: ~6000 cycles of address arithmetic on the one wave
    // that writes the tile, in a kernel that lasts 13 - 27 us)
    float* const pbase = g.part ? g.part + ((long)blockIdx.y * g.M + m0 + 4 * h) * g.R + (lane & 31) : nullptr;
    bf16_t* const tbase = g.t ? g.t + (long)(m0 + 4 * h) * g.ldt + (lane & 31) : nullptr;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int r = 32 * rb + (lane & 31);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int mq = (q & 3) + 8 * (q >> 2);                   // row of this register inside the tile, without the lane's half
            const int m = m0 + mq + 4 * h;
            float v = 0.f;
            if (rb < NL) {
                v = acc[rb < NL ? rb : 0][q];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += red[((w * NL + (rb < NL ? rb : 0)) * 64 + lane) * 16 + q];
            }
            if (m < g.M && r < g.R) {
                if (g.part) pbase[(long)mq * g.R + 32 * rb] = v;
                else tbase[(long)mq * g.ldt + 32 * rb] = f2bf(g.alpha * v);
            }
        }
    }
}

// t[m, r] = bf16(alpha * (part[0][m][r] + part[1][m][r] + ...)): the fixed-order sum of the split-K partial tiles; 4 columns per thread
__global__ __launch_bounds__(256) void lora_partial_reduce_kernel(const float* part, int ks, long MR, int R, float alpha, bf16_t* t, long ldt) {
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= MR) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int k = 0; k < ks; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(part + (long)k * MR + i4);
        a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
    }
    const long m = i4 / R;
    const int r = (int)(i4 - m * R);
    bf16_t* o = t + m * ldt + r;
    o[0] = f2bf(alpha * a0); o[1] = f2bf(alpha * a1); o[2] = f2bf(alpha * a2); o[3] = f2bf(alpha * a3);
}

struct LoraUpArgs {
    const bf16_t* dts; long ldd;    // [M, R]
    const bf16_t* AT; long ldat;    // [K, R]  (A transposed: row = input feature k)
    bf16_t* out; long ldo;          // [M, K]
    int M, K, R;
    int k_chunk;                    // columns per workgroup (multiple of 32)
    DropCfg d;
    int nb_live;                    // rank blocks that belong to a target module (padding blocks contribute nothing)
};

// wave = 32 rows of dts, walks 32-column tiles of the output; rank-32 product per target, masked, summed over targets
template <int RB, int NL>
__global__ __launch_bounds__(256) void lora_up_drop_kernel(LoraUpArgs g) {
    const int lane = lane_id(), wave = (int)threadIdx.x >> 6, h = lane >> 5;
    const int m_base = ((int)blockIdx.x * 4 + wave) * 32;
    if (m_base >= g.M) return;
    int mr = m_base + (lane & 31); mr = mr < g.M ? mr : g.M - 1;
    u32x4 df[NL][2];
#pragma unroll
    for (int rb = 0; rb < NL; ++rb)
#pragma unroll
        for (int c = 0; c < 2; ++c) df[rb][c] = ld16(g.dts + (long)mr * g.ldd + 32 * rb + 16 * c + 8 * h);
    uint32_t rowbase[8];                                       // m * K of the 8 rows this lane hashes (see below)
    {
        const uint32_t odd0 = (uint32_t)lane & 1u;             // column parity = lane parity (k0 is a multiple of 32)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = (int)(8 * odd0) + i;
            rowbase[i] = (uint32_t)(m_base + (q & 3) + 8 * (q >> 2) + 4 * h) * (uint32_t)g.K;
        }
    }
    const int k_lo = (int)blockIdx.y * g.k_chunk;
    int k_hi = k_lo + g.k_chunk; k_hi = k_hi < g.K ? k_hi : g.K;
    constexpr int UP_SP = 64 + 8;                               // staged row pitch (elements): 144 bytes
    __shared__ __attribute__((aligned(16))) bf16_t stage[4][32 * UP_SP];
    bf16_t* sw = stage[wave];
    const bool vec_ok = !(g.ldo & 7) && !((size_t)g.out & 15) && !(k_lo & 7);
    for (int k0 = k_lo; k0 < k_hi; k0 += 32) {
        const int kc = k0 + (lane & 31);
        const int kr = kc < g.K ? kc : g.K - 1;
        u32x4 af[NL][2];
#pragma unroll
        for (int rb = 0; rb < NL; ++rb)
#pragma unroll
            for (int c = 0; c < 2; ++c) af[rb][c] = ld16(g.AT + (long)kr * g.ldat + 32 * rb + 16 * c + 8 * h);
        float o[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) o[q] = 0.f;
        // a hash covers the column pair (k even, k odd) = this lane and lane ^ 1: each computes 8 of the 16 rows and they swap
        const uint32_t odd = (uint32_t)kr & 1u;
        const uint32_t fsh = odd ? 17u : 1u;                   // this column's 15-bit field inside a pair hash
#pragma unroll
        for (int rb = 0; rb < NL; ++rb) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            acc = mfma_32x32x16(df[rb][0], af[rb][0], acc);
            acc = mfma_32x32x16(df[rb][1], af[rb][1], acc);
            uint32_t hh[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t mine = drop_hash(g.d.seed[rb], (rowbase[i] + (uint32_t)kr) >> 1);     // even lane: register rows 0..7, odd: 8..15
                const uint32_t other = lane_swap1(mine);
                hh[i] = odd ? other : mine;
                hh[8 + i] = odd ? mine : other;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) o[q] += ((hh[q] >> fsh) & 0x7fffu) >= g.d.thr16 ? acc[q] : 0.f;
        }
        // round 6: the tile is turned through a wave-private LDS block and leaves as 16 bytes per lane — two 32-column tiles side by
        // side, 8 rows x 128 bytes per store instruction.  (Before: sixteen 2-byte stores per lane and tile, a wave instruction = 2 rows x
        // 64 bytes: lora_up_drop wrote its [M, K] output at ~1 TB/s, profiles/r6_j_sft_breakdown.md.)
        const int side = ((k0 - k_lo) >> 5) & 1;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            sw[((q & 3) + 8 * (q >> 2) + 4 * h) * UP_SP + 32 * side + (lane & 31)] = kc < g.K ? f2bf(o[q] * g.d.inv_keep) : (bf16_t)0;
        if (side == 1 || k0 + 32 >= k_hi) {
            wave_lds_sync();
            const int kf = k0 - 32 * side;                      // first column of the staged block
            const int ncol = 32 * (side + 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 8 * j + (lane >> 3), c8 = 8 * (lane & 7);
                const int m = m_base + row;
                const u32x4 v = ld16(&sw[row * UP_SP + c8]);
                if (m < g.M && c8 < ncol) {
                    bf16_t* op = g.out + (long)m * g.ldo + kf + c8;
                    if (vec_ok && kf + c8 + 8 <= g.K) st16(op, v);
                    else {
                        const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
                        for (int i = 0; i < 8; ++i) if (kf + c8 + i < g.K) op[i] = e[i];
                    }
                }
            }
            wave_lds_sync();
        }
    }
}

// mask image for tests / for injecting the same masks into the oracle: out[m, k] = keep(seed, m, k) as 0 / 1 bytes
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* out, long n, uint32_t seed, uint32_t thr16) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = drop_keep1(seed, (uint32_t)i, thr16) ? 1 : 0;
}

}  // namespace bra

using namespace bra;

static DropCfg make_cfg(float p, unsigned s0, unsigned s1, unsigned s2, unsigned s3) {
    DropCfg d;
    d.thr16 = drop_threshold(p);
    d.inv_keep = 1.f / (1.f - p);
    d.seed[0] = s0; d.seed[1] = s1; d.seed[2] = s2; d.seed[3] = s3;
    return d;
}

static int lora_down_launch(const void* x, long ldx, const void* A, long lda, void* t, long ldt, int M, int K, int R, float alpha, float p,
                            unsigned s0, unsigned s1, unsigned s2, unsigned s3, int nb_live, float* part, int ksplit, void* stream);

extern "C" int bra_lora_down_drop(const void* x, long ldx, const void* A, long lda, void* t, long ldt, int M, int K, int R,
                                  float alpha, float p, unsigned s0, unsigned s1, unsigned s2, unsigned s3, int nb_live, void* stream) {
    return lora_down_launch(x, ldx, A, lda, t, ldt, M, K, R, alpha, p, s0, s1, s2, s3, nb_live, nullptr, 1, stream);
}

extern "C" int bra_lora_down_splitk_plan(int M, int K) {
    // workgroups per row block so that the grid reaches ~2 per CU while every workgroup keeps >= 3 K steps (its prologue is one step)
    const int gx = (M + 31) / 32, nstep = (K + 127) / 128;
    // many row blocks (SFT / the full-row pass, M = 17 - 19 k): the chip is full either way, but a workgroup that walks 48 K steps alone
    // (down_proj, K = 6144) is a long latency chain — four K slices: 64.9 -> 52.6 us at M = 17 440, 65.6 -> 57.6 at 19 488; K = 2048: no
    // gain (profiles/r6_aa_lora_split_probe.txt)
    if (gx >= 192) return nstep >= 32 ? 4 : 1;
    if (gx <= 0 || nstep < 6) return 1;
    int want = 512 / gx, cap = nstep / 3;
    int ks = want < cap ? want : cap;
    if (ks < 2) return 1;
    const int per = (nstep + ks - 1) / ks;
    return (nstep + per - 1) / per;
}

extern "C" int bra_lora_down_drop_splitk(const void* x, long ldx, const void* A, long lda, void* t, long ldt, int M, int K, int R,
                                         float alpha, float p, unsigned s0, unsigned s1, unsigned s2, unsigned s3, int nb_live,
                                         float* part, int ksplit, void* stream) {
    if (ksplit <= 1) return lora_down_launch(x, ldx, A, lda, t, ldt, M, K, R, alpha, p, s0, s1, s2, s3, nb_live, nullptr, 1, stream);
    if (!part || ksplit > (K + 127) / 128 || ldt % 4) return BRA_ERR_ARG;
    return lora_down_launch(x, ldx, A, lda, t, ldt, M, K, R, alpha, p, s0, s1, s2, s3, nb_live, part, ksplit, stream);
}

static int lora_down_launch(const void* x, long ldx, const void* A, long lda, void* t, long ldt, int M, int K, int R, float alpha, float p,
                            unsigned s0, unsigned s1, unsigned s2, unsigned s3, int nb_live, float* part, int ksplit, void* stream) {
    if (M == 0) return 0;
    if (!x || !A || !t || M < 0 || K <= 0 || K % 8 || ldx % 8 || lda % 8 || (R != 32 && R != 64 && R != 128)) return BRA_ERR_ARG;
    if (!(p >= 0.f && p < 1.f) || (long)M * K >= (1l << 32)) return BRA_ERR_ARG;
    if (nb_live <= 0 || nb_live > R / 32) nb_live = R / 32;
    const int nstep = (K + 127) / 128;
    const int per = part ? (nstep + ksplit - 1) / ksplit : nstep;
    const int ny = part ? (nstep + per - 1) / per : 1;                 // no empty workgroup
    LoraDownArgs g = {(const bf16_t*)x, ldx, (const bf16_t*)A, lda, (bf16_t*)t, ldt, M, K, R, alpha, make_cfg(p, s0, s1, s2, s3), nb_live,
                      per, part};
    const dim3 grid((M + 31) / 32, ny);
    bra_stream_t st = (bra_stream_t)stream;
#define BRA_LD(RB_, NL_) BRA_LAUNCH((lora_down_drop_kernel<RB_, NL_>), grid, dim3(256), 0, st, g)
    if (R == 32) BRA_LD(1, 1);
    else if (R == 64) { if (nb_live == 1) BRA_LD(2, 1); else BRA_LD(2, 2); }
    else { if (nb_live == 3) BRA_LD(4, 3); else if (nb_live == 4) BRA_LD(4, 4); else { g.nb_live = 4; BRA_LD(4, 4); } }
#undef BRA_LD
    if (part) {
        const long MR = (long)M * R;
        BRA_LAUNCH(lora_partial_reduce_kernel, dim3((unsigned)((MR / 4 + 255) / 256)), dim3(256), 0, st, (const float*)part, ny, MR, R, alpha,
                   (bf16_t*)t, ldt);
    }
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_lora_up_drop(const void* dts, long ldd, const void* AT, long ldat, void* out, long ldo, int M, int K, int R,
                                float p, unsigned s0, unsigned s1, unsigned s2, unsigned s3, int nb_live, void* stream) {
    if (M == 0) return 0;
    if (!dts || !AT || !out || M < 0 || K <= 0 || ldd % 8 || ldat % 8 || (R != 32 && R != 64 && R != 128)) return BRA_ERR_ARG;
    if (!(p >= 0.f && p < 1.f) || (long)M * K >= (1l << 32)) return BRA_ERR_ARG;
    const int mblk = (M + 127) / 128;
    int splits = (1024 + mblk - 1) / mblk;                       // enough workgroups to fill the chip
    int k_chunk = ((K + splits - 1) / splits + 31) / 32 * 32;
    k_chunk = k_chunk < 128 ? 128 : k_chunk;
    if (nb_live <= 0 || nb_live > R / 32) nb_live = R / 32;
    LoraUpArgs g = {(const bf16_t*)dts, ldd, (const bf16_t*)AT, ldat, (bf16_t*)out, ldo, M, K, R, k_chunk, make_cfg(p, s0, s1, s2, s3), nb_live};
    const dim3 grid(mblk, (K + k_chunk - 1) / k_chunk);
    bra_stream_t st = (bra_stream_t)stream;
#define BRA_LU(RB_, NL_) BRA_LAUNCH((lora_up_drop_kernel<RB_, NL_>), grid, dim3(256), 0, st, g)
    if (R == 32) BRA_LU(1, 1);
    else if (R == 64) { if (nb_live == 1) BRA_LU(2, 1); else BRA_LU(2, 2); }
    else { if (nb_live == 3) BRA_LU(4, 3); else BRA_LU(4, 4); }        // (padding blocks hold zeros: treating them as live is exact, only slower)
#undef BRA_LU
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_dropout_mask(void* out, int M, int K, float p, unsigned seed, void* stream) {
    if (M <= 0 || K <= 0) return 0;
    if (!out || !(p >= 0.f && p < 1.f) || (long)M * K >= (1l << 32)) return BRA_ERR_ARG;
    const long n = (long)M * K;
    long grid = (n + 255) / 256; grid = grid > 4096 ? 4096 : grid;
    BRA_LAUNCH(dropout_mask_kernel, dim3((unsigned)grid), dim3(256), 0, (bra_stream_t)stream, (uint8_t*)out, n, (uint32_t)seed,
               drop_threshold(p));
    return BRA_LAUNCH_STATUS();
}
