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
template <int MODE, int NORM, int ACT, int OUTF32, int NW, int NL, int WIDE = 0, int PK = 0, int FAST = 0>
__global__ __launch_bounds__(NW * 64) void dec_gemm2_kernel(BRA_DG2_HEAD_PARAMS, DecGemm2Args g0) {
    // everything between kernel entry and the first weight / activation request arrives in SGPRs with the wave (kernel-argument
    // preload, -mllvm -amdgpu-kernarg-preload-count: the leading 14 dwords); the rest of the record is fetched from the argument
    // segment while the requests are in flight (a scalar load of the arguments is one memory round trip that every wave of a
    // 5 - 10 us launch otherwise waits for before it can form its first address)
    DecGemm2Args g = g0;
    g.x = h_x; g.W = h_W; g.res = h_res; g.ss_in = h_ss_in; g.ldx = h_ldx; g.ldres = h_ldres; g.M = h_M; g.N = h_N; g.K = h_K; g.nss_in = h_nss_in;
    static_assert(!WIDE || (MODE == 0 && NORM != 1), "wide rows: 16-column tiles, statistics applied in the epilogue");
    static_assert(!FAST || (PK && NORM != 1), "fast form: packed weights, folded norm or none");
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
        u32x4 w0[NL], w1[NL], x[NL], nv[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) x[u] = ld16(xp + so[u]);
        if (NORM == 1) {
#pragma unroll
            for (int u = 0; u < NL; ++u) nv[u] = ld16(g.nw + koff + so[u]);
        }
        // packed: lane l's 16 bytes of (tile, step) sit at ((tile * nsteps + step) * 64 + l) * 8 — one contiguous KiB per
        // wave-instruction (full 128-byte lines) instead of 16 row segments of 64 bytes
        long wo[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) wo[u] = FAST ? (long)(u * 512) : (PK ? (long)(st[u] < nsteps ? st[u] : nsteps - 1) * 512 : so[u]);
        const unsigned wlane = (unsigned)(wave * (NL * 512) + lane * 8);          // FAST: the lane's offset inside every tile
        auto tile_base = [&](int t) -> const bf16_t* {
            if (FAST) return g.W + (long)t * (NW * NL * 512) + wlane;           // scalar base + 32-bit lane offset (+ immediates)
            if (PK) return g.W + (long)t * nsteps * 512 + lane * 8;
            int rnn = t * NCOL + lrow; rnn = rnn < g.N ? rnn : g.N - 1;
            return g.W + (long)rnn * g.ldw + koff;
        };
        wp = tile_base(tile);
#pragma unroll
        for (int u = 0; u < NL; ++u) w0[u] = ld16_nt(wp + wo[u]);
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
        auto issue = [&](u32x4 (&wn)[NL], int t) {
            const bf16_t* wpn = tile_base(t);
#pragma unroll
            for (int u = 0; u < NL; ++u) wn[u] = ld16_nt(wpn + wo[u]);
        };
        auto compute = [&](u32x4 (&wc)[NL], int t, int it) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NL; ++u) acc = mfma_16x16x32(wc[u], x[u], acc);
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
                dg2_epilogue<MODE, ACT, OUTF32, PK>(g, v, t, lane, it == 0, resv);
            }
            if (!FAST && it == 0) dg2_stamp(g, 4);
        };
        if (NW == 16) {                   // 16 waves x 12 chunks: no registers for a second weight set; the host launches one
            compute(w0, tile, 0);         // workgroup per tile (launch_dg2)
            return;
        }
        int it = 0;
        for (; it + 2 < nmine; it += 2) {
            issue(w1, tile + (it + 1) * G); compute(w0, tile + it * G, it);
            issue(w0, tile + (it + 2) * G); compute(w1, tile + (it + 1) * G, it + 1);
        }
        if (nmine - it == 2) {
            issue(w1, tile + (it + 1) * G); compute(w0, tile + it * G, it);
            compute(w1, tile + (it + 1) * G, it + 1);
        } else {
            compute(w0, tile + it * G, it);
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

template <int MODE, int NORM, int ACT, int OUTF32, int WIDE = 0>
static int launch_dg2(const DecGemm2Args& g, bra_stream_t st) {
    constexpr int KS = MODE ? 64 : 32;
    const int nsteps = g.K / KS;
    // (wide rows stream K = 6144 of down_proj in 32-deep steps — the 8-row form takes it in 64-deep diagonal steps — so that
    //  projection runs 16 waves to keep the single register round)
    const int nw = (WIDE && nsteps >= 128) ? 16 : (nsteps >= 64 ? 8 : 4);
    const int spw = (nsteps + nw - 1) / nw;
    // (10: K = 2560, Qwen3-4B's hidden size — instantiated for the 8-wave form only; 4- and 16-wave shapes with ten steps per wave
    //  take nl = 8 in the generic multi-round loop, as before that form existed)
    const int nl = (spw >= 12 && spw % 12 == 0) ? 12 : ((spw == 10 && nw == 8) ? 10 : (spw > 4 ? 8 : 4));
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
                        int nss_out, int M, int N, int K, int act, int out_f32, int packed, void* probe, void* stream);

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
                        int nss_out, int M, int N, int K, int act, int out_f32, int packed, void* probe, void* stream) {
    (void)probe;
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
                      (const bf16_t*)res, ldres, out, ldo, ss_out, nss_out, M, N, K, packed, BRA_DBG_INIT((unsigned long long*)probe) 1.f / (float)K};
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

extern "C" int bra_row_sumsq(const void* x, long ldx, int M, int K, float* ss, int nss, void* stream) {
    if (M <= 0 || M > 16 || K <= 0 || K % 8 || ldx % 8 || !x || !ss || nss < 1) return BRA_ERR_ARG;
    BRA_LAUNCH(row_sumsq_kernel, dim3(M), dim3(256), 0, (bra_stream_t)stream, (const bf16_t*)x, ldx, K, ss, nss);
    return BRA_LAUNCH_STATUS();
}
