// k_wgrad.hip — low-rank weight gradients straight from row-major activations.
//   C[n, r] += alpha * sum_m Y[m, n] * T[m, r]            Y [M, N] bf16 (large), T [M, R] bf16 (R = 32, 64 or 128)
// These are autograd's gradients of PEFT's lora_B (Y = dy, T = s * x A^T) and lora_A (Y = x, T = s * dy B, written
// transposed), train_dna_qwen.py:155-167 / reason.py:376-388.  The contraction runs over the token index m, which is the
// SLOW index of both operands, so neither is "K-contiguous" as an MFMA fragment wants.  The first version transposed
// Y and T in HBM (one extra read + write of every activation) and ran a split-K NT GEMM; this kernel instead stages
// row-major tiles in LDS with coalesced 16-byte accesses and builds the fragments with 2-byte LDS reads down a column:
// eight times more LDS instructions than a b128 read, which is irrelevant for a kernel whose only real cost is
// streaming Y once from HBM (≈16 FLOP per byte).
#include "bra_device.h"
#include "bra_api_internal.h"
#include "bra_dropout.h"

namespace bra {

struct WgradArgs {
    const bf16_t* Y; long ldy;
    const bf16_t* T; long ldt;
    float* C; long c_sn, c_sr;      // element strides of C over n and r: dB [N, R] = (ld, 1); dA [R, K] = (1, ld)
    int M, N, R;
    int m_chunk;                    // rows of m per workgroup (multiple of 32)
    float alpha;
    DropCfg d;                      // DROP: Y is used as keep_rb(m, n) / (1 - p) * Y[m, n], one mask stream per rank block
    int nb_live;                    // DROP: rank blocks that belong to a target module (the rest is padding: skipped)
};

constexpr int WG_YP = 128 + 8;      // LDS row pitch of the Y tile (elements): 272 bytes, odd multiple of 16

// one workgroup: 128 columns of Y x all R, over m in [blockIdx.y * m_chunk, + m_chunk); 4 waves x 32 columns
// NL = live rank blocks (<= RB; the padding blocks of a fused projection hold zeros in T): compile-time, so that no branch
// sits between the accumulators and their MFMAs
// RM: C is [R, N] (dA: n is its contiguous index) — the accumulator lanes then run along n.  Compile-time: as a run-time select
// between mfma(bf, af) and mfma(af, bf) it cost 16 accumulator-register moves and a full MFMA drain behind every MFMA
template <int RB, int DROP, int NL = RB, int RM = 0>
__global__ __launch_bounds__(256) void wgrad_tn_kernel(WgradArgs g) {
    // plain: the Y tile double-buffered.  DROP: one MASKED copy of the tile per live rank block (the mask is applied where a
    // thread holds 8 consecutive elements of a row — one hash per element pair, no exchange between lanes — and the fragment
    // reads below then differ per rank block only by their base address), single-buffered with two barriers per step
    constexpr int NYS = DROP ? NL : 2;
    __shared__ bf16_t ys[NYS][32 * WG_YP];
    constexpr int WG_TP = 32 * RB + 8, TPT = RB >= 2 ? RB / 2 : 1;      // T tile pitch; 16-byte chunks per thread
    __shared__ bf16_t ts[2][32 * WG_TP];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int n0 = (int)blockIdx.x * 128;
    const int m_lo = (int)blockIdx.y * g.m_chunk;
    int m_hi = m_lo + g.m_chunk;
    m_hi = m_hi < g.M ? m_hi : g.M;
    const int nstep = m_hi > m_lo ? (m_hi - m_lo + 31) / 32 : 0;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    // staging roles: Y tile 32 x 128 = 512 chunks of 16 bytes (2 per thread), T tile 32 x (32 RB) = 128 RB chunks
    const int yr = tid >> 4, yc = (tid & 15) * 8;                 // rows yr and yr + 16, column chunk yc
    const bool ycol_ok = n0 + yc + 8 <= g.N;
    // issue = requests only, raw values; the out-of-range zeroing happens in commit.  (A `cond ? v : 0` on the loaded value
    // inside issue, with issue under `if (s + 1 < nstep)`, made the compiler wait for the tile right there: the "prefetch" of
    // step s + 1 completed before the MFMAs of step s started, i.e. latency + compute per step instead of their maximum.)
    u32x4 ry[2], rt[TPT];
    auto issue = [&](int s) {
        const int m0 = m_lo + 32 * s;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + yr + 16 * i;
            const int mc = m < g.M ? m : g.M - 1;
            ry[i] = ld16(g.Y + (long)mc * g.ldy + (ycol_ok ? n0 + yc : 0));
        }
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
            const int c = tid + 256 * i, tr = c / (4 * RB), tc = (c % (4 * RB)) * 8;
            const int m = m0 + (tr < 32 ? tr : 0);
            const int mc = m < g.M ? m : g.M - 1;
            rt[i] = ld16(g.T + (long)mc * g.ldt + tc);
        }
    };
    auto commit = [&](int buf, int s) {
        const int m0 = m_lo + 32 * s;
        u32x4 yv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) yv[i] = (m0 + yr + 16 * i < m_hi && ycol_ok) ? ry[i] : zero4;
        if (DROP) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t e0 = (uint32_t)(m_lo + 32 * s + yr + 16 * i) * (uint32_t)g.N + (uint32_t)(n0 + yc);
#pragma unroll
                for (int rb = 0; rb < NL; ++rb)
                    st16(&ys[rb][(yr + 16 * i) * WG_YP + yc], drop_apply8(yv[i], g.d.seed[rb], e0, g.d.thr16, g.d.inv_keep));
            }
        } else {
            st16(&ys[buf][yr * WG_YP + yc], yv[0]);
            st16(&ys[buf][(yr + 16) * WG_YP + yc], yv[1]);
        }
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
            const int c = tid + 256 * i, tr = c / (4 * RB), tc = (c % (4 * RB)) * 8;
            const int m = m0 + (tr < 32 ? tr : 0);
            if (tr < 32) st16(&ts[buf][tr * WG_TP + tc], m < m_hi ? rt[i] : zero4);
        }
    };

    f32x16 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;

    if (nstep > 0) { issue(0); commit(0, 0); }
    __syncthreads();
    const int ncol0 = wave * 32 + (lane & 31);      // this lane's Y column inside the tile (A row)
    constexpr bool r_major = RM != 0;               // C is [R, N] (dA): n is its contiguous index -> lanes along n in the epilogue
    for (int s = 0; s < nstep; ++s) {
        const int buf = s & 1;
        // no branch around the requests (the last step re-requests its own tile and drops it); the fence is ordered behind them,
        // and the fragment reads below are tied behind the fence through `ncol` (instruction selection would otherwise sink
        // the requests to their first use, the commit after the MFMAs)
        issue(s + 1 < nstep ? s + 1 : s);
        sched_fence();
        const int ncol = opaque_i(ncol0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int mb = 16 * kk + 8 * h;          // this lane's 8 consecutive m of the k = 16 step
            u32x4 af0;
            if (!DROP) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    af0[j] = (uint32_t)ys[buf][(mb + 2 * j) * WG_YP + ncol] | ((uint32_t)ys[buf][(mb + 2 * j + 1) * WG_YP + ncol] << 16);
            }
#pragma unroll
            for (int rb = 0; rb < NL; ++rb) {                    // (a padding block's T columns are zero: nothing to add)
                u32x4 af = af0;
                if (DROP) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        af[j] = (uint32_t)ys[rb][(mb + 2 * j) * WG_YP + ncol] | ((uint32_t)ys[rb][(mb + 2 * j + 1) * WG_YP + ncol] << 16);
                }
                const int rc = rb * 32 + (lane & 31);
                uint32_t b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    b[j] = (uint32_t)ts[buf][(mb + 2 * j) * WG_TP + rc] | ((uint32_t)ts[buf][(mb + 2 * j + 1) * WG_TP + rc] << 16);
                const u32x4 bf = {b[0], b[1], b[2], b[3]};
                // lanes of the accumulator run along the operand given SECOND: the contiguous index of C goes there
                acc[rb] = r_major ? mfma_32x32x16(bf, af, acc[rb]) : mfma_32x32x16(af, bf, acc[rb]);
            }
        }
        if (DROP) __syncthreads();                   // every wave is done with the masked copies of step s
        if (s + 1 < nstep) commit(buf ^ 1, s + 1);
        __syncthreads();
    }
    if (nstep == 0) return;
    if (r_major) {
        // D[i][j]: i = r (register index), j = column n of Y (lane & 31): 32 consecutive floats of a row of C per atomic instruction
        const int n = n0 + wave * 32 + (lane & 31);
        // (round 5) one lane-dependent base, wave-uniform offsets per atomic: each used to carry two 64-bit multiplies
        float* const cb = g.C + (long)n * g.c_sn + (long)(4 * h) * g.c_sr;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int rq = rb * 32 + (q & 3) + 8 * (q >> 2);
                if (n < g.N && rq + 4 * h < g.R) atomicAdd(cb + (long)rq * g.c_sr, g.alpha * acc[rb][q]);
            }
        return;
    }
    // D[i][j]: i = A row = column n of Y (register index), j = B row = r (lane & 31)
    float* const cb = g.C + (long)(n0 + wave * 32 + 4 * h) * g.c_sn + (long)(lane & 31) * g.c_sr;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int r = rb * 32 + (lane & 31);
        if (r >= g.R) continue;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int nq = (q & 3) + 8 * (q >> 2);
            const int n = n0 + wave * 32 + nq + 4 * h;
            if (n < g.N) atomicAdd(cb + (long)nq * g.c_sn + (long)(32 * rb) * g.c_sr, g.alpha * acc[rb][q]);
        }
    }
}

}  // namespace bra

using namespace bra;

static int wgrad_launch(WgradArgs& g, int m_chunk, bool drop, void* stream) {
    if (m_chunk <= 0) {
        // enough workgroups to fill the chip twice, at least 256 rows each
        const int ntile = (g.N + 127) / 128;
        int splits = (512 + ntile - 1) / ntile;
        m_chunk = (g.M + splits - 1) / splits;
        m_chunk = m_chunk < 256 ? 256 : m_chunk;
    }
    g.m_chunk = (m_chunk + 31) / 32 * 32;
    const dim3 grid((g.N + 127) / 128, (g.M + g.m_chunk - 1) / g.m_chunk);
    bra_stream_t st = (bra_stream_t)stream;
    const bool rm = g.c_sn == 1;
#define BRA_WG(RB_, NL_)                                                                          \
    do {                                                                                          \
        if (drop && rm) BRA_LAUNCH((wgrad_tn_kernel<RB_, 1, NL_, 1>), grid, dim3(256), 0, st, g); \
        else if (drop) BRA_LAUNCH((wgrad_tn_kernel<RB_, 1, NL_, 0>), grid, dim3(256), 0, st, g);  \
        else if (rm) BRA_LAUNCH((wgrad_tn_kernel<RB_, 0, RB_, 1>), grid, dim3(256), 0, st, g);    \
        else BRA_LAUNCH((wgrad_tn_kernel<RB_, 0, RB_, 0>), grid, dim3(256), 0, st, g);            \
    } while (0)
    if (g.R == 32) BRA_WG(1, 1);
    else if (g.R == 64) { if (drop && g.nb_live == 1) BRA_WG(2, 1); else BRA_WG(2, 2); }
    else { if (drop && g.nb_live == 3) BRA_WG(4, 3); else BRA_WG(4, 4); }
#undef BRA_WG
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_wgrad_tn(const void* Y, long ldy, const void* T, long ldt, float* C, long c_sn, long c_sr, int M, int N,
                            int R, float alpha, int m_chunk, void* stream) {
    if (M == 0 || N == 0) return 0;
    if (!Y || !T || !C || M < 0 || N < 0 || N % 8 || ldy % 8 || ldt % 8 || (R != 32 && R != 64 && R != 128)) return BRA_ERR_ARG;
    WgradArgs g = {(const bf16_t*)Y, ldy, (const bf16_t*)T, ldt, C, c_sn, c_sr, M, N, R, 0, alpha, {}, R / 32};
    return wgrad_launch(g, m_chunk, false, stream);
}

// the same with Y masked per rank block: dA of a LoRA branch whose input went through dropout (k_lora.hip)
extern "C" int bra_wgrad_tn_drop(const void* Y, long ldy, const void* T, long ldt, float* C, long c_sn, long c_sr, int M,
                                 int N, int R, float alpha, int m_chunk, float p, unsigned s0, unsigned s1, unsigned s2,
                                 unsigned s3, int nb_live, void* stream) {
    if (M == 0 || N == 0) return 0;
    if (!Y || !T || !C || M < 0 || N < 0 || N % 8 || ldy % 8 || ldt % 8 || (R != 32 && R != 64 && R != 128)) return BRA_ERR_ARG;
    if (!(p >= 0.f && p < 1.f) || (long)M * N >= (1l << 32)) return BRA_ERR_ARG;
    if (nb_live <= 0 || nb_live > R / 32) nb_live = R / 32;
    WgradArgs g = {(const bf16_t*)Y, ldy, (const bf16_t*)T, ldt, C, c_sn, c_sr, M, N, R, 0, alpha, {}, nb_live};
    g.d.thr16 = drop_threshold(p);
    g.d.inv_keep = 1.f / (1.f - p);
    g.d.seed[0] = s0; g.d.seed[1] = s1; g.d.seed[2] = s2; g.d.seed[3] = s3;
    return wgrad_launch(g, m_chunk, true, stream);
}
