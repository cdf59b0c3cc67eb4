// k_decgemm.hip — weight-streaming projections of the single-token decode step, second generation.
// Arithmetic: y[M<=16, N] = rmsnorm(x; w, eps) W^T (+ res) | SwiGLU | fp32 logits — Qwen3DecoderLayer.forward with a
// KV cache (TF:qwen3:294-323), Qwen3RMSNorm (TF:qwen3:59-64), Qwen3MLP (TF:qwen3:81-83), tied lm_head (TF:qwen3:495).
//
// At M = 8 rows a projection is pure HBM streaming of its weight matrix (8.4 - 50 MB per launch), and a launch lasts
// only a few microseconds, so what decides its speed is how many bytes are in flight how soon after the launch:
//   * every lane requests ALL the 16-byte weight chunks it will consume (NL per lane, non-temporal) in its first
//     instructions, together with the activations, the residual and the norm statistics; nothing waits on anything
//     before the first MFMA;
//   * no RMSNorm prologue pass: the PRODUCER of x (o_proj / down_proj epilogue here, bra_row_sumsq after the embedding
//     gather) leaves per-workgroup partial sums of squares [8][nss]; every wave folds them in a fixed order
//     (deterministic) and normalises just the fragments it multiplies;
//   * N = hidden projections (o_proj, down_proj) use 8-column workgroups so that 256 workgroups exist, and fill the
//     16x16x32 MFMA by putting two K-halves on the two row halves of A and B ("diagonal" mode): the products
//     C[j][b] and C[j+8][b+8] are the two K-halves of output (b, j); all 64 lanes stream weights.
#include "bra_decgemm.h"
#include "bra_api_internal.h"

namespace bra {

// MODE 0: 16 output columns per tile, 32 k per step.  MODE 1: 8 columns, 64 k per step (diagonal).
// A workgroup walks the tiles blockIdx.x, + gridDim.x, ..: its waves split K, keep their (normalised) activation
// fragments in registers for all tiles, request tile i+1's weights before multiplying tile i, and hand the K-reduction
// and epilogue of tile i to wave i % NW through a double-buffered LDS slab (one barrier per tile).
// WIDE: 9 .. 16 batch rows (two prompts x 8 rollouts per GPU): MODE 0 only (the diagonal tiles hold 8 rows), folded norm or none
// PK: the weights are in fragment order (compile-time: with a run-time flag every weight address is built both ways and selected,
// ~100 VALU instructions between kernel entry and the first weight request of a launch that lasts 5-11 us)
// FAST (the shapes of the decode step): packed weights, K == NW * NL * KS exactly (one register round, no step is clamped or
// zeroed), every offset fits 32 bits (host-checked), no probe stamps.  With two waves per SIMD each prologue instruction costs
// ~8 cycles of the launch's critical path, so the path to the first request is kept to: kernel arguments, one 24-bit multiply
// per base, scalar base + 32-bit lane offset + immediate per request.
#define BRA_DG2_HEAD_PARAMS const bf16_t* h_x, const bf16_t* h_W, const bf16_t* h_res, const float* h_ss_in, int h_ldx, int h_ldres, int h_M, int h_N, \
                            int h_K, int h_nss_in
#define BRA_DG2_HEAD_ARGS(g) (g).x, (g).W, (g).res, (g).ss_in, (int)(g).ldx, (int)(g).ldres, (g).M, (g).N, (g).K, (g).nss_in
// F8 (round 5; FAST only): the packed weights are fp8 e4m3 with one fp32 scale per output row (bra_dec_pack_weights_fp8) — HALF the
// streamed bytes.  A 16-byte request then carries the lane's fragments of TWO consecutive k-steps (8 + 8 bytes), decoded to bf16 in
// registers (exact: v_cvt_scalef32_pk_bf16_fp8, 4 instructions per step) in front of the same bf16 MFMAs; the row scale multiplies the
// K-reduced products next to rstd.  Activations, accumulation and every epilogue are unchanged (W8A16).
template <int MODE, int NORM, int ACT, int OUTF32, int NW, int NL, int WIDE = 0, int PK = 0, int FAST = 0, int F8 = 0>
__global__ __launch_bounds__(NW * 64) void dec_gemm2_kernel(BRA_DG2_HEAD_PARAMS, DecGemm2Args g0) {
    // everything between kernel entry and the first weight / activation request arrives in SGPRs with the wave (kernel-argument
    // preload, -mllvm -amdgpu-kernarg-preload-count: the leading 14 dwords); the rest of the record is fetched from the argument
    // segment while the requests are in flight (a scalar load of the arguments is one memory round trip that every wave of a
    // 5 - 10 us launch otherwise waits for before it can form its first address)
    DecGemm2Args g = g0;
    g.x = h_x; g.W = h_W; g.res = h_res; g.ss_in = h_ss_in; g.ldx = h_ldx; g.ldres = h_ldres; g.M = h_M; g.N = h_N; g.K = h_K; g.nss_in = h_nss_in;
    static_assert(!WIDE || (MODE == 0 && NORM != 1), "wide rows: 16-column tiles, statistics applied in the epilogue");
    static_assert(!FAST || (PK && NORM != 1), "fast form: packed weights, folded norm or none");
    static_assert(!F8 || (FAST && NL % 2 == 0 && NW != 16 && !WIDE), "fp8 weights: fast form, two k-steps per 16-byte request");
    constexpr int NLW = F8 ? NL / 2 : NL;                               // 16-byte weight requests per lane and tile
    __shared__ float red[2][NW][64][4];
    constexpr int KS = MODE ? 64 : 32;
    constexpr int NCOL = MODE ? 8 : 16;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    if (!FAST) dg2_stamp(g, 0);
    const int nsteps = FAST ? NW * NL : g.K / KS;
    const int ntiles = (g.N + NCOL - 1) / NCOL;
    const int lrow = MODE ? (fr & 7) : fr;                            // A row = weight row, B row = batch row
    const int koff = MODE ? ((fr >> 3) * 32 + fq * 8) : fq * 8;
    const int xr = lrow < g.M ? lrow : g.M - 1;
    const bf16_t* xp = FAST ? g.x + (dg2_mul24(xr, (int)g.ldx) + (unsigned)koff) : g.x + (long)xr * g.ldx + koff;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int rounds = FAST ? 1 : (nsteps + NW * NL - 1) / (NW * NL);
    int tile = (int)blockIdx.x;
    int rn = tile * NCOL + lrow; rn = rn < g.N ? rn : g.N - 1;
    const bf16_t* wp = FAST ? g.W : g.W + (long)rn * g.ldw + koff;

    // epilogue operand of the first tile requested now (all waves, clamped address: no branch around a load)
    const int em = fr < g.M ? fr : g.M - 1;
    int en = tile * NCOL + 4 * fq; en = en + 3 < g.N ? en : (g.N >= 4 ? g.N - 4 : 0);
    u32x2 resv = {0u, 0u};
    if (!FAST && !ACT && !OUTF32 && g.res) resv = ld8(g.res + (long)em * g.ldres + en);

    // NORM: lane l folds partials [per * (l & 7), per * (l & 7) + per) of row l >> 3 (per <= 32, fixed order =>
    // run-to-run identical); requested first, they are the smallest and the first thing the MFMAs need
    // NORM == 2 ("folded"): the RMSNorm weight is already multiplied into the (packed) weights and the row factor rstd is
    // applied to the K-reduced products in the epilogue — y = rstd * (x (W . nw)^T) — so no wave loads norm weights or
    // statistics up front (the per-CU load path, 64 B/clk, is what bounds these kernels: the activation-side loads of
    // NORM == 1 are 3x the weight bytes); wave 0 alone folds the statistics, behind its weight requests, into LDS.
    const int per = NORM ? g.nss_in >> 3 : 0;
    f32x4 pv[8];
    __shared__ float rs_lds[WIDE ? 16 : 8];
    if (NORM == 1) {
        const float* pp = g.ss_in + (long)(lane >> 3) * g.nss_in + (lane & 7) * per;
#pragma unroll
        for (int i = 0; i < 8; ++i) pv[i] = *reinterpret_cast<const f32x4*>(pp + (4 * i < per ? 4 * i : 0));
    }
    float rstd = 1.f;

    if (rounds == 1) {
        int st[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u)      // MODE 0: a wave takes both 64-byte halves of a 128-byte line back to back;
            st[u] = PK ? wave * NL + u            // packed weights: NL consecutive KiB blocks per wave and tile
                             : (MODE ? wave + NW * u : 2 * (wave + NW * (u >> 1)) + (u & 1));
        long so[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) so[u] = FAST ? (long)(u * KS) : (long)(st[u] < nsteps ? st[u] : nsteps - 1) * KS;
        if (FAST) xp += (unsigned)(wave * (NL * KS));
        // request order = arrival order (one in-order counter per wave): statistics, activations, norm weights, then
        // the weight tile, so that the normalisation below runs while the weights are still in flight
        u32x4 w0[NLW], w1[NLW], x[NL], nv[NL];
        f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0;                      // F8: scales of the four weight rows behind this lane's products
#pragma unroll
        for (int u = 0; u < NL; ++u) x[u] = ld16(xp + so[u]);
        if (NORM == 1) {
#pragma unroll
            for (int u = 0; u < NL; ++u) nv[u] = ld16(g.nw + koff + so[u]);
        }
        // packed: lane l's 16 bytes of (tile, step) sit at ((tile * nsteps + step) * 64 + l) * 8 — one contiguous KiB per
        // wave-instruction (full 128-byte lines) instead of 16 row segments of 64 bytes
        long wo[NLW];
#pragma unroll
        for (int u = 0; u < NLW; ++u) wo[u] = FAST ? (long)(u * 512) : (PK ? (long)(st[u] < nsteps ? st[u] : nsteps - 1) * 512 : so[u]);
        const unsigned wlane = (unsigned)(wave * (NLW * 512) + lane * 8);         // FAST: the lane's offset inside every tile
        // F8: the row scales of tile t, requested by every wave with the tile's weights (no branch around a load; one KiB line
        // shared by the workgroup) — lane (fr, fq) holds the products of weight rows (4 fq + r) & (NCOL - 1)
        auto scale_of = [&](int t) -> f32x4 {
            return *reinterpret_cast<const f32x4*>(g.wscale + (unsigned)(t * NCOL + ((4 * fq) & (NCOL - 1))));
        };
        auto tile_base = [&](int t) -> const bf16_t* {
            if (FAST) return g.W + (long)t * (NW * NLW * 512) + wlane;          // scalar base + 32-bit lane offset (+ immediates)
            if (PK) return g.W + (long)t * nsteps * 512 + lane * 8;
            int rnn = t * NCOL + lrow; rnn = rnn < g.N ? rnn : g.N - 1;
            return g.W + (long)rnn * g.ldw + koff;
        };
        wp = tile_base(tile);
#pragma unroll
        for (int u = 0; u < NLW; ++u) w0[u] = ld16_nt(wp + wo[u]);
        if (F8) {
            sched_fence();                // the weight requests go out before anything waits for the scale pointer (argument segment)
            sc0 = scale_of(tile);
        }
        if (FAST && !ACT && !OUTF32) {            // needed last (epilogue of the first tile), requested last; no branch around the load
            const bf16_t* rp = g.res ? g.res : g.x;
            const unsigned ro = g.res ? dg2_mul24(em, (int)g.ldres) + (unsigned)en : 0u;
            resv = ld8(rp + ro);
        }
        sched_fence();                    // every request above is in flight before the first dependent instruction
        if (!FAST) dg2_stamp(g, 1);
        // folded norm: wave 0 requests the statistics partials behind its weights — all of them, unconditionally (a load that
        // sits under a uniform `4 i < per` test is issued and awaited one at a time: eight serial round trips on the wave
        // that every other wave then waits for at the first barrier) — and folds them in the shadow of the first tile's MFMAs
        constexpr int NPASS = WIDE ? 2 : 1;
        f32x4 sq[NPASS][8];
        if (NORM == 2 && wave == 0) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int srow = (lane >> 3) + 8 * pass;                                    // (rows past M fold row M - 1 again)
                const int sr = srow < g.M ? srow : g.M - 1;
                const float* pp = FAST ? g.ss_in + (dg2_mul24(sr, g.nss_in) + (unsigned)((lane & 7) * per))
                                       : g.ss_in + (long)sr * g.nss_in + (lane & 7) * per;
#pragma unroll
                for (int i = 0; i < 8; ++i) sq[pass][i] = *reinterpret_cast<const f32x4*>(pp + (4 * i < per ? 4 * i : 0));
            }
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass)
#pragma unroll
                for (int i = 0; i < 8; ++i) reg_fence(sq[pass][i]);
        }
        auto fold_stats = [&]() {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int srow = (lane >> 3) + 8 * pass;
                const float rs = dg2_fold_rstd(sq[pass], per, g.inv_K, g.eps);
                if ((lane & 7) == 0) rs_lds[srow] = rs;       // visible after the tile barrier
            }
        };
        if (NORM == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) reg_fence(pv[i]);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += 4 * i < per ? (pv[i][0] + pv[i][1]) + (pv[i][2] + pv[i][3]) : 0.f;
            s += wave_shfl_xor(s, 1); s += wave_shfl_xor(s, 2); s += wave_shfl_xor(s, 4);
            rstd = rsqrtf(wave_shfl(s, xr * 8) / (float)g.K + g.eps);
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                float xf[8], nf[8];
                unpack8(x[u], xf); unpack8(nv[u], nf);
#pragma unroll
                for (int i = 0; i < 8; ++i) xf[i] = nf[i] * round_bf(xf[i] * rstd);
                x[u] = pack8(xf);
            }
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) if (!FAST && st[u] >= nsteps) x[u] = zero4;        // steps past K contribute zero
        // weight registers ping-pong (w0 / w1): tile i+1 is requested before tile i is multiplied.  The requests sit
        // in straight-line code (no branch around a load: the compiler would wait for them at the join), so the last
        // one or two tiles are peeled.
        const int G = (int)gridDim.x;
        const int nmine = (ntiles - tile + G - 1) / G;
        auto issue = [&](u32x4 (&wn)[NLW], f32x4& scn, int t) {
            const bf16_t* wpn = tile_base(t);
#pragma unroll
            for (int u = 0; u < NLW; ++u) wn[u] = ld16_nt(wpn + wo[u]);
            if (F8) scn = scale_of(t);
        };
        auto compute = [&](u32x4 (&wc)[NLW], const f32x4& scc, int t, int it) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (F8 != 0) {
#pragma unroll
                for (int u = 0; u < NLW; ++u) {
                    acc = mfma_16x16x32(f8x8_to_bf16x8(wc[u].x, wc[u].y), x[2 * u], acc);
                    acc = mfma_16x16x32(f8x8_to_bf16x8(wc[u].z, wc[u].w), x[2 * u + 1], acc);
                }
            } else {
#pragma unroll
                for (int u = 0; u < NL; ++u) acc = mfma_16x16x32(wc[u], x[u], acc);
            }
            if (NORM == 2 && it == 0 && wave == 0) fold_stats();
            float (*slab)[64][4] = red[it & 1];
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[wave][lane][r] = acc[r];
            if (!FAST && it == 0) dg2_stamp(g, 2);
            __syncthreads();
            if (!FAST && it == 0) dg2_stamp(g, 3);
            if (wave == it % NW) {
                float v[4];
                dg2_reduce<NW>(slab, lane, v);
                if (NORM == 2) {
                    const int mrow = MODE ? (fr & 7) : fr;                   // batch row of this lane's products (both K-halves of MODE 1)
                    const float rsf = rs_lds[mrow < g.M ? mrow : g.M - 1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= rsf;
                }
                if (F8) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= scc[r];
                }
                dg2_epilogue<MODE, ACT, OUTF32, PK>(g, v, t, lane, it == 0, resv);
            }
            if (!FAST && it == 0) dg2_stamp(g, 4);
        };
        if (NW == 16) {                   // 16 waves x 12 chunks: no registers for a second weight set; the host launches one
            compute(w0, sc0, tile, 0);    // workgroup per tile (launch_dg2)
            return;
        }
        int it = 0;
        for (; it + 2 < nmine; it += 2) {
            issue(w1, sc1, tile + (it + 1) * G); compute(w0, sc0, tile + it * G, it);
            issue(w0, sc0, tile + (it + 2) * G); compute(w1, sc1, tile + (it + 1) * G, it + 1);
        }
        if (nmine - it == 2) {
            issue(w1, sc1, tile + (it + 1) * G); compute(w0, sc0, tile + it * G, it);
            compute(w1, sc1, tile + (it + 1) * G, it + 1);
        } else {
            compute(w0, sc0, tile + it * G, it);
        }
        return;
    }

    // K larger than one register round (NW * NL steps): one tile per iteration, fragments re-read every round
    for (int it = 0; tile < ntiles; ++it, tile += (int)gridDim.x) {
        rn = tile * NCOL + lrow; rn = rn < g.N ? rn : g.N - 1;
        wp = g.W + (long)rn * g.ldw + koff;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int rd = 0; rd < rounds; ++rd) {
            const int base = rd * NW * NL;
            int st[NL];
#pragma unroll
            for (int u = 0; u < NL; ++u) st[u] = base + (MODE ? wave + NW * u : 2 * (wave + NW * (u >> 1)) + (u & 1));
            u32x4 w[NL], x[NL], nv[NL];
            // (packed weights: chunk ((tile * nsteps + step) * 64 + lane) holds the lane's 8 elements of (tile, step) — K beyond one
            //  register round, e.g. Qwen3-4B's o / down projections, walks the same image round by round)
            const bf16_t* wpk = g.W + ((long)tile * nsteps * 64 + lane) * 8;
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int sc = st[u] < nsteps ? st[u] : nsteps - 1;
                w[u] = PK ? ld16_nt(wpk + (long)sc * 512) : ld16_nt(wp + (long)sc * KS);
            }
#pragma unroll
            for (int u = 0; u < NL; ++u) { const int sc = st[u] < nsteps ? st[u] : nsteps - 1; x[u] = ld16(xp + (long)sc * KS); }
            if (NORM == 1) {
#pragma unroll
                for (int u = 0; u < NL; ++u) { const int sc = st[u] < nsteps ? st[u] : nsteps - 1; nv[u] = ld16(g.nw + koff + (long)sc * KS); }
                if (it == 0 && rd == 0) {
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) s += 4 * i < per ? (pv[i][0] + pv[i][1]) + (pv[i][2] + pv[i][3]) : 0.f;
                    s += wave_shfl_xor(s, 1); s += wave_shfl_xor(s, 2); s += wave_shfl_xor(s, 4);
                    rstd = rsqrtf(wave_shfl(s, xr * 8) / (float)g.K + g.eps);
                }
#pragma unroll
                for (int u = 0; u < NL; ++u) {
                    float xf[8], nf[8];
                    unpack8(x[u], xf); unpack8(nv[u], nf);
#pragma unroll
                    for (int i = 0; i < 8; ++i) xf[i] = nf[i] * round_bf(xf[i] * rstd);
                    x[u] = pack8(xf);
                }
            }
#pragma unroll
            for (int u = 0; u < NL; ++u) acc = mfma_16x16x32(st[u] < nsteps ? w[u] : zero4, x[u], acc);
        }
        float (*slab)[64][4] = red[it & 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[wave][lane][r] = acc[r];
        __syncthreads();
        if (wave == it % NW) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
#pragma unroll
                for (int wv = 0; wv < NW; ++wv) s += slab[wv][lane][r];
                v[r] = s;
            }
            dg2_epilogue<MODE, ACT, OUTF32>(g, v, tile, lane, it == 0, resv);
        }
    }
}

// sums of squares of the rows of x [M <= 16, K] (after the embedding gather): ss[r][0] = sum, ss[r][1 .. nss) = 0
__global__ __launch_bounds__(256) void row_sumsq_kernel(const bf16_t* x, long ldx, int K, float* ss, int nss) {
    __shared__ float part[4];
    const int r = (int)blockIdx.x, tid = (int)threadIdx.x;
    float s = 0.f;
    for (int j = tid; j < K / 8; j += 256) {
        float f[8];
        unpack8(ld16(x + (long)r * ldx + j * 8), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += f[i] * f[i];
    }
    s = wave_sum<64>(s);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    for (int j = tid; j < nss; j += 256) ss[(long)r * nss + j] = j == 0 ? (part[0] + part[1]) + (part[2] + part[3]) : 0.f;
}

// W [N, K] row-major -> fragment order of dec_gemm2_kernel<MODE>: chunk ((tile * nsteps + step) * 64 + lane) holds the 8
// elements lane (fr, fq) feeds to the MFMA of (tile, step).  One thread per 16-byte chunk, coalesced on the write side.
template <int MODE>
__global__ __launch_bounds__(256) void dec_pack_kernel(const bf16_t* W, long ldw, int N, int K, const bf16_t* nw, bf16_t* out) {
    constexpr int KS = MODE ? 64 : 32, NCOL = MODE ? 8 : 16;
    const int nsteps = K / KS;
    const long nchunk = (long)(N / NCOL) * nsteps * 64;
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= nchunk) return;
    const int lane = (int)(c & 63);
    const long ts = c >> 6;
    const int step = (int)(ts % nsteps), tile = (int)(ts / nsteps);
    const int fr = lane & 15, fq = lane >> 4;
    const int row = tile * NCOL + (MODE ? (fr & 7) : fr);
    const int col = step * KS + (MODE ? ((fr >> 3) * 32 + fq * 8) : fq * 8);
    u32x4 v = ld16(W + (long)row * ldw + col);
    if (nw) {                                 // fold the RMSNorm weight of the projection's input: W[n, k] * nw[k], one rounding
        float f[8], s[8];
        unpack8(v, f); unpack8(ld16(nw + col), s);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] *= s[i];
        v = pack8(f);
    }
    st16(out + c * 8, v);
}

// fp8 image of a projection (BASELINE config 5: "fp8 weights"): scale[n] = max_k |W[n, k] nw[k]| / 448 (the e4m3
// maximum; 1 for an all-zero row), q[n, k] = e4m3(W[n, k] nw[k] / scale[n]) rounded to nearest even.  One wave per row.
__global__ __launch_bounds__(256) void dec_rowscale_fp8_kernel(const bf16_t* W, long ldw, int N, int K, const bf16_t* nw, float* scale) {
    const int row = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (row >= N) return;
    float mx = 0.f;
    for (int j = lane; j < K / 8; j += 64) {
        float f[8], s[8];
        unpack8(ld16(W + (long)row * ldw + j * 8), f);
        if (nw) { unpack8(ld16(nw + j * 8), s);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] *= s[i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(f[i]));
    }
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, wave_shfl_xor(mx, m));
    if (lane == 0) scale[row] = mx > 0.f ? mx * (1.0f / 448.0f) : 1.0f;
}

// fragment order of dec_gemm2_kernel<.., F8 = 1>: 16-byte chunk ((tile * nsteps / 2 + s2) * 64 + lane) = the lane's 8 elements of
// k-step 2 s2 followed by its 8 elements of k-step 2 s2 + 1, one byte each.  One thread per chunk.
template <int MODE>
__global__ __launch_bounds__(256) void dec_pack_fp8_kernel(const bf16_t* W, long ldw, int N, int K, const bf16_t* nw, const float* scale,
                                                           unsigned* out) {
    constexpr int KS = MODE ? 64 : 32, NCOL = MODE ? 8 : 16;
    const int npair = K / KS / 2;
    const long nchunk = (long)(N / NCOL) * npair * 64;
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= nchunk) return;
    const int lane = (int)(c & 63);
    const long ts = c >> 6;
    const int s2 = (int)(ts % npair), tile = (int)(ts / npair);
    const int fr = lane & 15, fq = lane >> 4;
    const int row = tile * NCOL + (MODE ? (fr & 7) : fr);
    const float inv = 1.0f / scale[row];
    unsigned o[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = (2 * s2 + h) * KS + (MODE ? ((fr >> 3) * 32 + fq * 8) : fq * 8);
        float f[8];
        unpack8(ld16(W + (long)row * ldw + col), f);
        if (nw) { float s[8]; unpack8(ld16(nw + col), s);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] *= s[i]; }
        unsigned b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = f32_to_e4m3(f[i] * inv);
        o[2 * h] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        o[2 * h + 1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
    }
    u32x4 v = {o[0], o[1], o[2], o[3]};
    *reinterpret_cast<u32x4*>(out + c * 4) = v;
}

// waves per workgroup and 16-byte chunks per lane of a projection with `nsteps` k-steps (one rule for the launcher and for the fp8
// packer, which must refuse what the fp8 kernel cannot stream)
static void dg2_waves_and_chunks(int nsteps, bool wide, int& nw, int& nl) {
    nw = (wide && nsteps >= 128) ? 16 : (nsteps >= 64 ? 8 : 4);
    const int spw = (nsteps + nw - 1) / nw;
    // (10: K = 2560, Qwen3-4B's hidden size — instantiated for the 8-wave form only; 4- and 16-wave shapes with ten steps per wave
    //  take nl = 8 in the generic multi-round loop, as before that form existed)
    nl = (spw >= 12 && spw % 12 == 0) ? 12 : ((spw == 10 && nw == 8) ? 10 : (spw > 4 ? 8 : 4));
}

template <int MODE, int NORM, int ACT, int OUTF32, int WIDE = 0>
static int launch_dg2(const DecGemm2Args& g, bra_stream_t st) {
    constexpr int KS = MODE ? 64 : 32;
    const int nsteps = g.K / KS;
    // (wide rows stream K = 6144 of down_proj in 32-deep steps — the 8-row form takes it in 64-deep diagonal steps — so that
    //  projection runs 16 waves to keep the single register round)
    int nw, nl;
    dg2_waves_and_chunks(nsteps, WIDE != 0, nw, nl);
    if (NORM == 2 && (nsteps + nw * nl - 1) / (nw * nl) != 1) return BRA_ERR_UNSUPPORTED;     // folded norm: single-round path only
    const int ntiles = MODE ? g.N / 8 : (g.N + 15) / 16;
    // one workgroup of 8 waves (two of 4) per CU, looping over the tiles; the 16-wave form takes one tile per workgroup
    const int gmax = nw == 16 ? ntiles : (nw >= 8 ? 256 : 512);
    const dim3 grid(ntiles < gmax ? ntiles : gmax);
    // (wide rows beyond one register round — Qwen3-4B's down projection, K = 9728 — take the generic multi-round loop like 8 rows;
    //  only the folded-norm form, refused above, is tied to the single-round path)
    // fast form: see the kernel; everything it assumes is checked here
    const bool fits32 = g.ldx < (1 << 24) && g.ldres < (1 << 24) && 16 * g.ldx < (1L << 30) && 16 * g.ldres + g.N < (1L << 30) &&
                        16L * g.nss_in < (1L << 24);
#ifdef BRA_DEBUG
    const bool probed = g.probe != nullptr;          // the fast form carries no stamps
#else
    const bool probed = false;
#endif
    const bool fast = NORM != 1 && (g.packed & 1) && nsteps == nw * nl && !probed && fits32;
    (void)fast;
#define BRA_DG2(NW_, NL_)                                                                                                      \
    do {                                                                                                                       \
        if constexpr (NORM != 1 && !WIDE && NW_ != 16 && NL_ % 2 == 0) {                                                       \
            if (g.wscale) {                                                                                                    \
                if (!fast) return BRA_ERR_UNSUPPORTED;                                                                         \
                BRA_LAUNCH((dec_gemm2_kernel<MODE, NORM, ACT, OUTF32, NW_, NL_, WIDE, 1, 1, 1>), grid, dim3(NW_ * 64), 0, st, BRA_DG2_HEAD_ARGS(g), g); \
                break;                                                                                                         \
            }                                                                                                                  \
        }                                                                                                                      \
        if (g.wscale) return BRA_ERR_UNSUPPORTED;                                                                              \
        if constexpr (NORM != 1) {                                                                                             \
            if (fast) {                                                                                                        \
                BRA_LAUNCH((dec_gemm2_kernel<MODE, NORM, ACT, OUTF32, NW_, NL_, WIDE, 1, 1>), grid, dim3(NW_ * 64), 0, st, BRA_DG2_HEAD_ARGS(g), g); \
                break;                                                                                                         \
            }                                                                                                                  \
        }                                                                                                                      \
        if (g.packed & 1) BRA_LAUNCH((dec_gemm2_kernel<MODE, NORM, ACT, OUTF32, NW_, NL_, WIDE, 1>), grid, dim3(NW_ * 64), 0, st, BRA_DG2_HEAD_ARGS(g), g); \
        else BRA_LAUNCH((dec_gemm2_kernel<MODE, NORM, ACT, OUTF32, NW_, NL_, WIDE, 0>), grid, dim3(NW_ * 64), 0, st, BRA_DG2_HEAD_ARGS(g), g);      \
    } while (0)
    if (nw == 16) {
        if constexpr (WIDE != 0) { if (nl == 12) BRA_DG2(16, 12); else if (nl == 8) BRA_DG2(16, 8); else BRA_DG2(16, 4); }
    }
    else if (nw == 8) { if (nl == 12) BRA_DG2(8, 12); else if (nl == 10) BRA_DG2(8, 10); else if (nl == 8) BRA_DG2(8, 8); else BRA_DG2(8, 4); }
    else { if (nl == 12) BRA_DG2(4, 12); else if (nl == 8) BRA_DG2(4, 8); else BRA_DG2(4, 4); }
#undef BRA_DG2
    return BRA_LAUNCH_STATUS();
}

}  // namespace bra

using namespace bra;

static int dec_gemm2_any(const void* x, long ldx, const float* ss_in, int nss_in, const void* norm_w, float eps,
                        const void* W, long ldw, const void* res, long ldres, void* out, long ldo, float* ss_out,
                        int nss_out, int M, int N, int K, int act, int out_f32, int packed, void* probe, void* stream,
                        const float* wscale = nullptr);

extern "C" int bra_dec_gemm2(const void* x, long ldx, const float* ss_in, int nss_in, const void* norm_w, float eps,
                             const void* W, long ldw, const void* res, long ldres, void* out, long ldo, float* ss_out,
                             int nss_out, int M, int N, int K, int act, int out_f32, void* stream) {
    return dec_gemm2_any(x, ldx, ss_in, nss_in, norm_w, eps, W, ldw, res, ldres, out, ldo, ss_out, nss_out, M, N, K, act, out_f32,
                         0, nullptr, stream);
}

extern "C" int bra_dec_gemm2_packed(const void* x, long ldx, const float* ss_in, int nss_in, const void* norm_w, float eps,
                                    const void* W, long ldw, const void* res, long ldres, void* out, long ldo, float* ss_out,
                                    int nss_out, int M, int N, int K, int act, int out_f32, int packed, void* stream) {
    return dec_gemm2_any(x, ldx, ss_in, nss_in, norm_w, eps, W, ldw, res, ldres, out, ldo, ss_out, nss_out, M, N, K, act, out_f32,
                         packed, nullptr, stream);
}

// fp8 (e4m3) weights with one fp32 scale per output row (bra_dec_pack_weights_fp8): y = rstd * scale[n] * (x Wq^T) ... — the same
// epilogues as bra_dec_gemm2_packed.  `norm_folded` != 0: ss_in / nss_in carry the statistics of x and the norm weight was folded
// into Wq before quantisation.  BRA_ERR_UNSUPPORTED outside the single-register-round shapes (the caller keeps bf16 weights there).
extern "C" int bra_dec_gemm2_fp8(const void* x, long ldx, const float* ss_in, int nss_in, float eps, const void* Wq, const float* wscale,
                                 const void* res, long ldres, void* out, long ldo, float* ss_out, int nss_out, int M, int N, int K,
                                 int act, int out_f32, int norm_folded, void* stream) {
    if (!Wq || !wscale) return BRA_ERR_ARG;
    // (norm_w only selects the folded-norm form here: its values are never read when the weights carry it)
    return dec_gemm2_any(x, ldx, norm_folded ? ss_in : nullptr, norm_folded ? nss_in : 0, norm_folded ? Wq : nullptr, eps, Wq, K, res, ldres,
                         out, ldo, ss_out, nss_out, M, N, K, act, out_f32, norm_folded ? 3 : 1, nullptr, stream, wscale);
}

#ifdef BRA_DEBUG
extern "C" int bra_dec_gemm2_probe(const void* x, long ldx, const float* ss_in, int nss_in, const void* norm_w, float eps,
                                   const void* W, long ldw, const void* res, long ldres, void* out, long ldo, float* ss_out,
                                   int nss_out, int M, int N, int K, int act, int out_f32, int packed, void* probe, void* stream) {
    return dec_gemm2_any(x, ldx, ss_in, nss_in, norm_w, eps, W, ldw, res, ldres, out, ldo, ss_out, nss_out, M, N, K, act, out_f32,
                         packed, probe, stream);
}
#endif

static int dec_gemm2_any(const void* x, long ldx, const float* ss_in, int nss_in, const void* norm_w, float eps,
                        const void* W, long ldw, const void* res, long ldres, void* out, long ldo, float* ss_out,
                        int nss_out, int M, int N, int K, int act, int out_f32, int packed, void* probe, void* stream,
                        const float* wscale) {
    (void)probe;
    if (wscale && (M > 8 || !(packed & 1) || (norm_w && !(packed & 2)))) return BRA_ERR_UNSUPPORTED;   // fp8 weights: <= 8 rows, packed, norm folded or absent
    if (M <= 0 || M > 16 || N <= 0 || K <= 0 || K % 32 || ldx % 8 || ldw % 8 || !x || !W || !out) return BRA_ERR_ARG;
    if (act && (N % 16 || out_f32 || res || ss_out)) return BRA_ERR_ARG;
    if (out_f32 && res) return BRA_ERR_ARG;            // (out_f32 with ss_out: per-tile maxima of the logits, 16-column tiles)
    if (norm_w && (!ss_in || nss_in < 32 || nss_in % 32 || nss_in > 256)) return BRA_ERR_ARG;
    if (res && ldres % 4) return BRA_ERR_ARG;
    const bool wide = M > 8;                     // 9 .. 16 rows: 16-column tiles everywhere (a diagonal tile holds 8 rows)
    // 8-column (diagonal) tiles where 16-column tiles would leave CUs idle — as long as the N / 8 statistics partials they emit still
    // fit the 256 a consumer folds (N = 2560, Qwen3-4B: 16-column tiles, 160 workgroups)
    const bool diag = !wide && !act && !out_f32 && N % 8 == 0 && K % 64 == 0 && (N + 15) / 16 < 256 && N / 8 <= 256;
    if (ss_out && nss_out < (diag ? N / 8 : (N + 15) / 16)) return BRA_ERR_ARG;
    DecGemm2Args g = {(const bf16_t*)x, ldx, ss_in, nss_in, (const bf16_t*)norm_w, eps, (const bf16_t*)W, ldw,
                      (const bf16_t*)res, ldres, out, ldo, ss_out, nss_out, M, N, K, packed, wscale, BRA_DBG_INIT((unsigned long long*)probe) 1.f / (float)K};
    if (packed && (N % (diag ? 8 : 16) || K % (diag ? 64 : 32))) return BRA_ERR_ARG;
    bra_stream_t st = (bra_stream_t)stream;
    if ((packed & 2) && !(packed & 1)) return BRA_ERR_ARG;
    if (wide) {
        // statistics only in the folded form (rstd in the epilogue); the weights must be packed for 16-column tiles
        if (norm_w && !(packed & 2)) return BRA_ERR_UNSUPPORTED;
        if (norm_w) {
            if (act) return launch_dg2<0, 2, 1, 0, 1>(g, st);
            if (out_f32) return launch_dg2<0, 2, 0, 1, 1>(g, st);
            return launch_dg2<0, 2, 0, 0, 1>(g, st);
        }
        if (act) return launch_dg2<0, 0, 1, 0, 1>(g, st);
        if (out_f32) return launch_dg2<0, 0, 0, 1, 1>(g, st);
        return launch_dg2<0, 0, 0, 0, 1>(g, st);
    }
    if (norm_w && (packed & 2)) {            // RMSNorm weight folded into the packed weights, rstd applied in the epilogue
        if (act) return launch_dg2<0, 2, 1, 0>(g, st);
        if (out_f32) return launch_dg2<0, 2, 0, 1>(g, st);
        return diag ? launch_dg2<1, 2, 0, 0>(g, st) : launch_dg2<0, 2, 0, 0>(g, st);
    }
    if (norm_w) {
        if (act) return launch_dg2<0, 1, 1, 0>(g, st);
        if (out_f32) return launch_dg2<0, 1, 0, 1>(g, st);
        return diag ? launch_dg2<1, 1, 0, 0>(g, st) : launch_dg2<0, 1, 0, 0>(g, st);
    }
    if (act) return launch_dg2<0, 0, 1, 0>(g, st);
    if (out_f32) return launch_dg2<0, 0, 0, 1>(g, st);
    return diag ? launch_dg2<1, 0, 0, 0>(g, st) : launch_dg2<0, 0, 0, 0>(g, st);
}

extern "C" int bra_dec_pack_weights_rows(const void* W, long ldw, int N, int K, int act, int out_f32, const void* norm_w, int rows,
                                         void* out, void* stream);
extern "C" int bra_dec_pack_weights(const void* W, long ldw, int N, int K, int act, int out_f32, const void* norm_w, void* out,
                                    void* stream) {
    return bra_dec_pack_weights_rows(W, ldw, N, K, act, out_f32, norm_w, 8, out, stream);
}

// rows = batch rows the copy will be streamed against: more than 8 selects the 16-column tile order for every projection
extern "C" int bra_dec_pack_weights_rows(const void* W, long ldw, int N, int K, int act, int out_f32, const void* norm_w, int rows,
                                         void* out, void* stream) {
    if (!W || !out || N <= 0 || K <= 0 || ldw % 8 || rows <= 0 || rows > 16) return BRA_ERR_ARG;
    const bool diag = rows <= 8 && !act && !out_f32 && N % 8 == 0 && K % 64 == 0 && (N + 15) / 16 < 256 && N / 8 <= 256;      // bra_dec_gemm2's rule
    if (N % (diag ? 8 : 16) || K % (diag ? 64 : 32)) return BRA_ERR_UNSUPPORTED;
    const long nchunk = (long)N * K / 8;
    const dim3 grid((unsigned)((nchunk + 255) / 256));
    if (diag) BRA_LAUNCH((dec_pack_kernel<1>), grid, dim3(256), 0, (bra_stream_t)stream, (const bf16_t*)W, ldw, N, K, (const bf16_t*)norm_w, (bf16_t*)out);
    else BRA_LAUNCH((dec_pack_kernel<0>), grid, dim3(256), 0, (bra_stream_t)stream, (const bf16_t*)W, ldw, N, K, (const bf16_t*)norm_w, (bf16_t*)out);
    return BRA_LAUNCH_STATUS();
}

// W [N, K] bf16 (optionally with the input's RMSNorm weight folded in) -> fp8 e4m3 image in bra_dec_gemm2_fp8's fragment order
// (out_q: N * K bytes) + one fp32 scale per row (out_scale: N floats).  Same tile rule as bra_dec_pack_weights for <= 8 batch rows;
// BRA_ERR_UNSUPPORTED when the shape is not a whole number of tiles and k-step pairs.
extern "C" int bra_dec_pack_weights_fp8(const void* W, long ldw, int N, int K, int act, int out_f32, const void* norm_w, void* out_q,
                                        float* out_scale, void* stream) {
    if (!W || !out_q || !out_scale || N <= 0 || K <= 0 || ldw % 8) return BRA_ERR_ARG;
    const bool diag = !act && !out_f32 && N % 8 == 0 && K % 64 == 0 && (N + 15) / 16 < 256 && N / 8 <= 256;      // bra_dec_gemm2's rule
    if (N % (diag ? 8 : 16) || K % (diag ? 128 : 64)) return BRA_ERR_UNSUPPORTED;
    {   // only what bra_dec_gemm2_fp8 streams: K exactly one register round of the (waves, chunks) the launcher picks
        int nw, nl;
        const int nsteps = K / (diag ? 64 : 32);
        dg2_waves_and_chunks(nsteps, false, nw, nl);
        if (nsteps != nw * nl || nl % 2 || nw == 16) return BRA_ERR_UNSUPPORTED;
    }
    bra_stream_t st = (bra_stream_t)stream;
    BRA_LAUNCH(dec_rowscale_fp8_kernel, dim3((N + 3) / 4), dim3(256), 0, st, (const bf16_t*)W, ldw, N, K, (const bf16_t*)norm_w, out_scale);
    int rc = BRA_LAUNCH_STATUS();
    if (rc) return rc;
    const long nchunk = (long)N * K / 16;
    const dim3 grid((unsigned)((nchunk + 255) / 256));
    if (diag) BRA_LAUNCH((dec_pack_fp8_kernel<1>), grid, dim3(256), 0, st, (const bf16_t*)W, ldw, N, K, (const bf16_t*)norm_w, out_scale, (unsigned*)out_q);
    else BRA_LAUNCH((dec_pack_fp8_kernel<0>), grid, dim3(256), 0, st, (const bf16_t*)W, ldw, N, K, (const bf16_t*)norm_w, out_scale, (unsigned*)out_q);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_row_sumsq(const void* x, long ldx, int M, int K, float* ss, int nss, void* stream) {
    if (M <= 0 || M > 16 || K <= 0 || K % 8 || ldx % 8 || !x || !ss || nss < 1) return BRA_ERR_ARG;
    BRA_LAUNCH(row_sumsq_kernel, dim3(M), dim3(256), 0, (bra_stream_t)stream, (const bf16_t*)x, ldx, K, ss, nss);
    return BRA_LAUNCH_STATUS();
}
