// k_gemm.hip — bf16 MFMA GEMM for gfx950 (CDNA4), "NT" form:
//     C[M,N] = alpha * ( A[M,K] * B[N,K]^T  +  A2[M,K2] * B2[N,K2]^T ) (+ epilogue)
// Every linear layer of the hot path is this contraction (both operands
// K-contiguous): nn.Linear forward  y = x W^T  (TF:qwen3:225-236, TF:qwen3:81-83,
// TF:esm:336-338,402, dna_llm.py:159-160), its dgrad dx = dy (W^T)^T against a
// pre-transposed frozen weight, and the tied lm_head (TF:qwen3:495).  The
// second operand pair (A2,B2,K2) carries the LoRA update  y += (x A^T) B^T
// (PEFT, configured at train_dna_qwen.py:155-167 / reason.py:376-388) inside
// the same accumulators, so LoRA costs K2/K extra MFMA work and no extra pass
// over y.
//
// Structure: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, each
// 64x64 = 4x4 fragments of v_mfma_f32_16x16x32_bf16), BK = 64 (or 32), LDS
// double-buffered with a 16-byte-chunk XOR swizzle (conflict-free
// ds_read_b128), register-staged global->LDS pipeline with one barrier per
// K-step, XCD-aware grouped tile order.  Operand roles are swapped
// (D[n][m]) so every lane owns 4 consecutive output columns of one row.
#include <type_traits>

#include "bra_device.h"
#include "bra_api_internal.h"

namespace bra {

enum : int {
    EPI_BF16 = 0,     // C bf16 = rnd(alpha*acc [+bias]) [+res, rounded again]
    EPI_F32 = 1,      // C f32  = alpha*acc [+bias] (+ C when accumulate)
    EPI_LSE = 2,      // no C; per (row, 64-col chunk) running max / sum-exp of rnd(acc), + target logit
    EPI_DLOGIT = 3,   // C bf16 = rnd(coef[m] * ((n==tgt[m]) - exp(rnd(acc) - lse[m])))
    EPI_ATOMIC = 4,   // C f32 += alpha*acc with atomics (split-K)
    EPI_SWIGLU = 5,   // ring kernel only: B = [gate rows | up rows] ([2 F, K]); C bf16 [M, F] = rnd(rnd(silu(rnd(g))) * rnd(u)) (no-grad passes)
};

struct GemmArgs {
    const bf16_t* A;  long lda;
    const bf16_t* B;  long ldb;
    const bf16_t* A2; long lda2;
    const bf16_t* B2; long ldb2;
    void* C;          long ldc;
    int M, N, K, K2;
    float alpha;
    const bf16_t* bias;   // [N] or null
    const bf16_t* res;    // [M, ldres] or null
    long ldres;
    int accumulate;       // EPI_F32: C += result
    int split_k;          // number of K slices (EPI_ATOMIC)
    // EPI_LSE / EPI_DLOGIT
    const int* tgt;       // [M] target column per row (or -1)
    float* part_max;      // [M, nchunk]
    float* part_sum;      // [M, nchunk]
    float* tgt_logit;     // [M]
    const float* lse;     // [M]
    const float* coef;    // [M]
    int nchunk;
    int swiglu_F;         // EPI_SWIGLU: F (N = 2 F); B row of local tile row `loc` = (loc >> 5) * 16 + (loc & 15) + 128 tile_n (+ F when loc & 16)
};

template <int BK>
__device__ __forceinline__ int swz_chunk(int row, int c) {
    return BK == 64 ? (c ^ (row & 7)) : (c ^ ((row >> 2) & 3));
}

// ---- EPI_BF16 for a wave whose MI x 4 fragments all lie inside the matrix (every tile but the last tile row / column).
// The generic epilogues below walk the fragments one by one under `m < M` / `n < N` tests, and a load under a branch is issued
// and awaited alone: with a residual that was MI x 4 = 32 memory round trips in series at the end of every tile (one workgroup
// per CU: nothing else runs meanwhile), each with two 64-bit multiplies for its addresses.  Here: one base pointer per operand,
// row steps by addition, column steps as immediates, every residual word requested before the first is used.
#ifdef BRA_EMU
__device__ __forceinline__ void epi_pin(u32x2&) {}
#else
__device__ __forceinline__ void epi_pin(u32x2& v) { asm volatile("" : "+v"(v)); }
#endif
// alpha * acc, then + bias: two roundings, as the generic epilogue computes it (its add sits under `if (g.bias)`); -ffast-math
// would contract this into one fma here and the two epilogues would differ in the last bit
// (the backend fuses under the global fusion mode whatever a contract pragma says: an empty asm on the product separates them)
template <bool BIAS>
__device__ __forceinline__ float epi_scale_bias(float a, float alpha, float b) {
    float t = a * alpha;
    if (BIAS) {
#ifndef BRA_EMU
        asm volatile("" : "+v"(t));
#endif
        t = t + b;
    }
    return t;
}
template <int MI, bool BIAS>
__device__ __forceinline__ void epi_bf16_interior_b(const GemmArgs& g, f32x4 (&acc)[4][MI], int mw0, int nw0, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    const float alpha = g.alpha;
    const int col = nw0 + 4 * fq;
    bf16_t* cp = (bf16_t*)g.C + ((long)(mw0 + fr) * g.ldc + col);
    const long cstep = 16 * g.ldc;
    float bz[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) bz[ni][r] = 0.f;
    if (BIAS) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const u32x2 b = ld8(g.bias + col + 16 * ni);
            bz[ni][0] = bf_lo(b.x); bz[ni][1] = bf_hi(b.x); bz[ni][2] = bf_lo(b.y); bz[ni][3] = bf_hi(b.y);
        }
    }
    if (g.res) {
        const bf16_t* rp = g.res + ((long)(mw0 + fr) * g.ldres + col);
        const long rstep = 16 * g.ldres;
        u32x2 rv[MI][4];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) rv[mi][ni] = ld8(rp + mi * rstep + ni * 16);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) epi_pin(rv[mi][ni]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = epi_scale_bias<BIAS>(acc[ni][mi][r], alpha, bz[ni][r]);
                const u32x2 w = rv[mi][ni];
                v[0] = round_bf(v[0]) + bf_lo(w.x); v[1] = round_bf(v[1]) + bf_hi(w.x);
                v[2] = round_bf(v[2]) + bf_lo(w.y); v[3] = round_bf(v[3]) + bf_hi(w.y);
                u32x2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                st8(cp + mi * cstep + ni * 16, o);
            }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = epi_scale_bias<BIAS>(acc[ni][mi][r], alpha, bz[ni][r]);
            u32x2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
            st8(cp + mi * cstep + ni * 16, o);
        }
}
template <int MI>
__device__ __forceinline__ void epi_bf16_interior(const GemmArgs& g, f32x4 (&acc)[4][MI], int mw0, int nw0, int lane) {
    if (g.bias) epi_bf16_interior_b<MI, true>(g, acc, mw0, nw0, lane);
    else epi_bf16_interior_b<MI, false>(g, acc, mw0, nw0, lane);
}
__device__ __forceinline__ bool epi_interior(const GemmArgs& g, int mw0, int nw0, int rows) {
    return mw0 + rows <= g.M && nw0 + 64 <= g.N && !(g.ldc & 3) && !(g.ldres & 3);
}

// ---- the same interior epilogue with the wave's 16 MI x 64 block turned through LDS (round 6; VERDICT r5 #5: the direct form stores 8 bytes
// per lane — a wave instruction is 16 rows x 32 bytes, every 128-byte line of C is written by four instructions and the residual is
// fetched in the same shape: ~2.7 TB/s, 8 - 25 us of every one-prompt launch, profiles/r5_l_gemm_fixed_probe.txt).  Here the rounded
// bf16 product (alpha acc + bias: the FIRST rounding of the direct form) is written to a wave-private LDS block [16 MI][144 B] (row stride
// padded by 16 B), read back as 16-byte chunks with lane l on (row l >> 3, chunk l & 7), and stored / added to the residual as 8 rows x
// 128 bytes per wave instruction: full lines.  Same values bit for bit (the residual add works on the same rounded product).
// `lw`: 16 MI * 144 bytes of LDS private to this wave, free of pending reads (every kernel calls this behind its last K-loop barrier).
constexpr int EPI_RS = 144;
#if defined(BRA_DEBUG) && !defined(BRA_EMU)
__device__ int epi_via_lds = 1;                  // A/B knob of the debug library (bra_gemm_set_epi_lds): 0 = the direct 8-byte stores
#else
constexpr int epi_via_lds = 1;
#endif
__device__ __forceinline__ bool epi_lds_ok(const GemmArgs& g) {
    return !(g.ldc & 7) && !(g.ldres & 7) && !((size_t)g.C & 15) && !((size_t)g.res & 15);
}
template <int MI, bool BIAS>
__device__ __forceinline__ void epi_bf16_interior_lds_b(const GemmArgs& g, f32x4 (&acc)[4][MI], int mw0, int nw0, int lane, char* lw) {
    const int fr = lane & 15, fq = lane >> 4;
    const float alpha = g.alpha;
    const int rrow = lane >> 3, rch = lane & 7;
    // the first residual rows are requested before the staging (their latency hides behind it), the rest batch by batch: holding all
    // 2 MI x 4 words beside the accumulators spilled in the four-wave kernel (656 bytes of scratch at 5 x 8 fragments, tests/test_isa.py)
    constexpr int NB = 4;                               // 16-byte rows per batch
    constexpr int NJ = 2 * MI;
    const bf16_t* rp = g.res ? g.res + ((long)(mw0 + rrow) * g.ldres + nw0 + 8 * rch) : nullptr;
    const long rstep = 8 * g.ldres;
    u32x4 rv[NB];
    if (g.res) {
#pragma unroll
        for (int j = 0; j < NB && j < NJ; ++j) rv[j] = ld16(rp + j * rstep);
    }
    float bz[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) bz[ni][r] = 0.f;
    if (BIAS) {
        const int col = nw0 + 4 * fq;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const u32x2 b = ld8(g.bias + col + 16 * ni);
            bz[ni][0] = bf_lo(b.x); bz[ni][1] = bf_hi(b.x); bz[ni][2] = bf_lo(b.y); bz[ni][3] = bf_hi(b.y);
        }
    }
    char* wp = lw + fr * EPI_RS + 8 * fq;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = epi_scale_bias<BIAS>(acc[ni][mi][r], alpha, bz[ni][r]);
            u32x2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
            st8(wp + mi * (16 * EPI_RS) + ni * 32, o);
        }
    wave_lds_sync();
    sched_fence();
    const char* rdp = lw + rrow * EPI_RS + 16 * rch;
    bf16_t* cp = (bf16_t*)g.C + ((long)(mw0 + rrow) * g.ldc + nw0 + 8 * rch);
    const long cstep = 8 * g.ldc;
#pragma unroll
    for (int j0 = 0; j0 < NJ; j0 += NB) {
        u32x4 xv[NB];
#pragma unroll
        for (int j = 0; j < NB && j0 + j < NJ; ++j) xv[j] = ld16(rdp + (j0 + j) * (8 * EPI_RS));
        if (g.res) {
#pragma unroll
            for (int j = 0; j < NB && j0 + j < NJ; ++j) {
                float x[8], r8[8];
                unpack8(xv[j], x);
                unpack8(rv[j], r8);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] += r8[i];
                st16(cp + (j0 + j) * cstep, pack8(x));
            }
            if (j0 + NB < NJ) {
#pragma unroll
                for (int j = 0; j < NB && j0 + NB + j < NJ; ++j) rv[j] = ld16(rp + (j0 + NB + j) * rstep);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB && j0 + j < NJ; ++j) st16(cp + (j0 + j) * cstep, xv[j]);
        }
    }
    wave_lds_sync();                 // (a caller may reuse the block for its next column group)
}
template <int MI>
__device__ __forceinline__ void epi_bf16_interior_lds(const GemmArgs& g, f32x4 (&acc)[4][MI], int mw0, int nw0, int lane, char* lw) {
    if (g.bias) epi_bf16_interior_lds_b<MI, true>(g, acc, mw0, nw0, lane, lw);
    else epi_bf16_interior_lds_b<MI, false>(g, acc, mw0, nw0, lane, lw);
}

// ---- epilogue: lane owns C[m][n..n+3], m = m0 + wm*64 + 16*mi + fr, n = n0 + wn*64 + 16*ni + 4*fq
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[4][4], int m0, int n0, int wm, int wn,
                                              int lane, int tile_n) {
    if (EPI == EPI_BF16 && epi_interior(g, m0 + wm * 64, n0 + wn * 64, 64)) {
        epi_bf16_interior<4>(g, acc, m0 + wm * 64, n0 + wn * 64, lane);
        return;
    }
    const int fr = lane & 15, fq = lane >> 4;
    const float alpha = g.alpha;
    if (EPI == EPI_BF16 || EPI == EPI_F32 || EPI == EPI_ATOMIC || EPI == EPI_DLOGIT) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wm * 64 + mi * 16 + fr;
            if (m >= g.M) continue;
            float lse_m = 0.f, coef_m = 0.f; int tgt_m = -1;
            if (EPI == EPI_DLOGIT) { lse_m = g.lse[m]; coef_m = g.coef[m]; tgt_m = g.tgt[m]; }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wn * 64 + ni * 16 + fq * 4;
                if (n >= g.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r] * alpha;
                const bool full = (n + 3 < g.N);
                if (EPI == EPI_BF16) {
                    if (g.bias) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] += bf2f(g.bias[n + r]);
                    }
                    if (g.res) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < g.N) v[r] = round_bf(v[r]) + bf2f(g.res[(long)m * g.ldres + n + r]);
                    }
                    bf16_t* cp = (bf16_t*)g.C + (long)m * g.ldc + n;
                    if (full) {
                        u32x2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                        st8(cp, o);
                    } else {
                        for (int r = 0; r < 4; ++r) if (n + r < g.N) cp[r] = f2bf(v[r]);
                    }
                } else if (EPI == EPI_DLOGIT) {
                    bf16_t* cp = (bf16_t*)g.C + (long)m * g.ldc + n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = __expf(round_bf(v[r]) - lse_m);
                        v[r] = coef_m * (((n + r) == tgt_m ? 1.f : 0.f) - p);
                    }
                    if (full) {
                        u32x2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                        st8(cp, o);
                    } else {
                        for (int r = 0; r < 4; ++r) if (n + r < g.N) cp[r] = f2bf(v[r]);
                    }
                } else if (EPI == EPI_F32) {
                    float* cp = (float*)g.C + (long)m * g.ldc + n;
                    if (g.bias) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] += bf2f(g.bias[n + r]);
                    }
                    if (full) {
                        f32x4 o = {v[0], v[1], v[2], v[3]};
                        if (g.accumulate) o += *reinterpret_cast<const f32x4*>(cp);
                        *reinterpret_cast<f32x4*>(cp) = o;
                    } else {
                        for (int r = 0; r < 4; ++r)
                            if (n + r < g.N) cp[r] = g.accumulate ? cp[r] + v[r] : v[r];
                    }
                } else {  // EPI_ATOMIC
                    float* cp = (float*)g.C + (long)m * g.ldc + n;
                    for (int r = 0; r < 4; ++r) if (n + r < g.N) atomicAdd(cp + r, v[r]);
                }
            }
        }
    } else {  // EPI_LSE: each wave reduces its 64 columns per row
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wm * 64 + mi * 16 + fr;
            const int tgt_m = (m < g.M) ? g.tgt[m] : -1;
            float vmax = -3.0e38f;
            float vals[16];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wn * 64 + ni * 16 + fq * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = round_bf(acc[ni][mi][r] * alpha);
                    bool ok = (n + r) < g.N;
                    vals[ni * 4 + r] = ok ? x : -3.0e38f;
                    if (ok) vmax = fmaxf(vmax, x);
                    if (ok && (n + r) == tgt_m) g.tgt_logit[m] = x;
                }
            }
            // lanes fr, fr+16, fr+32, fr+48 hold the same row: reduce over fq
            vmax = fmaxf(vmax, wave_shfl_xor(vmax, 16));
            vmax = fmaxf(vmax, wave_shfl_xor(vmax, 32));
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += (vals[i] > -1.0e38f) ? __expf(vals[i] - vmax) : 0.f;
            s += wave_shfl_xor(s, 16);
            s += wave_shfl_xor(s, 32);
            const int chunk = tile_n * 2 + wn;
            if (fq == 0 && m < g.M && chunk < g.nchunk) {
                g.part_max[(long)m * g.nchunk + chunk] = vmax;
                g.part_sum[(long)m * g.nchunk + chunk] = s;
            }
        }
    }
}

// the same epilogues for a wave that owns MI x 4 fragments at (mw0, nw0): lane owns C[mw0 + 16 mi + fr][nw0 + 16 ni + 4 fq ..+3]
template <int EPI, int MI>
__device__ __forceinline__ void gemm_epilogue_w(const GemmArgs& g, f32x4 (&acc)[4][MI], int mw0, int nw0, int lane, char* lw = nullptr) {
    if (EPI == EPI_BF16 && epi_interior(g, mw0, nw0, 16 * MI)) {
        if (lw && epi_lds_ok(g) && epi_via_lds) epi_bf16_interior_lds<MI>(g, acc, mw0, nw0, lane, lw);
        else epi_bf16_interior<MI>(g, acc, mw0, nw0, lane);
        return;
    }
    const int fr = lane & 15, fq = lane >> 4;
    const float alpha = g.alpha;
    if (EPI == EPI_LSE) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = mw0 + mi * 16 + fr;
            const int tgt_m = (m < g.M) ? g.tgt[m] : -1;
            float vmax = -3.0e38f;
            float vals[16];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = nw0 + ni * 16 + fq * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = round_bf(acc[ni][mi][r] * alpha);
                    bool ok = (n + r) < g.N;
                    vals[ni * 4 + r] = ok ? x : -3.0e38f;
                    if (ok) vmax = fmaxf(vmax, x);
                    if (ok && (n + r) == tgt_m) g.tgt_logit[m] = x;
                }
            }
            vmax = fmaxf(vmax, wave_shfl_xor(vmax, 16));
            vmax = fmaxf(vmax, wave_shfl_xor(vmax, 32));
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += (vals[i] > -1.0e38f) ? __expf(vals[i] - vmax) : 0.f;
            s += wave_shfl_xor(s, 16);
            s += wave_shfl_xor(s, 32);
            const int chunk = nw0 >> 6;
            if (fq == 0 && m < g.M && chunk < g.nchunk) {
                g.part_max[(long)m * g.nchunk + chunk] = vmax;
                g.part_sum[(long)m * g.nchunk + chunk] = s;
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = mw0 + mi * 16 + fr;
        if (m >= g.M) continue;
        float lse_m = 0.f, coef_m = 0.f; int tgt_m = -1;
        if (EPI == EPI_DLOGIT) { lse_m = g.lse[m]; coef_m = g.coef[m]; tgt_m = g.tgt[m]; }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = nw0 + ni * 16 + fq * 4;
            if (n >= g.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r] * alpha;
            const bool full = (n + 3 < g.N);
            if (EPI == EPI_BF16 || EPI == EPI_DLOGIT) {
                if (EPI == EPI_BF16) {
                    if (g.bias) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] += bf2f(g.bias[n + r]);
                    }
                    if (g.res) {
                        if (full && !(g.ldres & 3)) {
                            const u32x2 rv = ld8(g.res + (long)m * g.ldres + n);
                            v[0] = round_bf(v[0]) + bf_lo(rv.x); v[1] = round_bf(v[1]) + bf_hi(rv.x);
                            v[2] = round_bf(v[2]) + bf_lo(rv.y); v[3] = round_bf(v[3]) + bf_hi(rv.y);
                        } else {
                            for (int r = 0; r < 4; ++r)
                                if (n + r < g.N) v[r] = round_bf(v[r]) + bf2f(g.res[(long)m * g.ldres + n + r]);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = __expf(round_bf(v[r]) - lse_m);
                        v[r] = coef_m * (((n + r) == tgt_m ? 1.f : 0.f) - p);
                    }
                }
                bf16_t* cp = (bf16_t*)g.C + (long)m * g.ldc + n;
                if (full) {
                    u32x2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                    st8(cp, o);
                } else {
                    for (int r = 0; r < 4; ++r) if (n + r < g.N) cp[r] = f2bf(v[r]);
                }
            } else if (EPI == EPI_F32) {
                float* cp = (float*)g.C + (long)m * g.ldc + n;
                if (g.bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] += bf2f(g.bias[n + r]);
                }
                if (full) {
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    if (g.accumulate) o += *reinterpret_cast<const f32x4*>(cp);
                    *reinterpret_cast<f32x4*>(cp) = o;
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < g.N) cp[r] = g.accumulate ? cp[r] + v[r] : v[r];
                }
            } else {  // EPI_ATOMIC
                float* cp = (float*)g.C + (long)m * g.ldc + n;
                for (int r = 0; r < 4; ++r) if (n + r < g.N) atomicAdd(cp + r, v[r]);
            }
        }
    }
}

// BM x 128 output tile, BM/64 x 2 waves of 64x64; PF = register prefetch depth in K-tiles (the global loads of
// tile t+PF are in flight while tile t is multiplied; LDS stays double-buffered).
template <int BM, int BK, int EPI, int PF>
__global__ __launch_bounds__(BM * 2) void gemm_nt_kernel(GemmArgs g) {
    constexpr int BN = 128, NT = BM * 2;
    constexpr int CH = BK / 8;              // 16-byte chunks per tile row
    constexpr int LPTA = (BM * CH) / NT;    // 16-byte loads per thread per K-step, A operand
    constexpr int LPTB = (BN * CH) / NT;    // ... B operand
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    BRA_DYN_SMEM(smem);                     // [2 buffers][A tile | B tile]

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;   // wave position in the (BM/64) x 2 grid

    // ---- tile order: XCD-aware, grouped along M so a group re-uses B panels from L2
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int ntiles = tiles_m * tiles_n;
    int bid = (int)blockIdx.x;
    int kslice = 0;
    if (EPI == EPI_ATOMIC) { kslice = bid / ntiles; bid -= kslice * ntiles; }
    bid = (int)xcd_remap((unsigned)bid, (unsigned)ntiles);
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int gid = bid / per_group;
    const int first_m = gid * GROUP;
    const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int tile_m = first_m + (bid % per_group) % gsize;
    const int tile_n = (bid % per_group) / gsize;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- K range (main + LoRA operand pair, optionally sliced)
    const int nk1 = g.K / BK, nk2 = g.K2 / BK;
    int kt_begin = 0, kt_end = nk1 + nk2;
    if (EPI == EPI_ATOMIC && g.split_k > 1) {
        int per = (kt_end + g.split_k - 1) / g.split_k;
        kt_begin = kslice * per;
        kt_end = kt_begin + per < kt_end ? kt_begin + per : kt_end;
        if (kt_begin >= kt_end) return;
    }

    // ---- per-thread global->LDS staging slots
    // register sets are named statically (S is a compile-time constant at every call site): a run-time index
    // into a register array would send it to scratch
    u32x4 ra[PF][LPTA], rb[PF][LPTB];

    auto issue_loads = [&](auto S, int kt) {
        constexpr int s = decltype(S)::value;
        const bf16_t* Ap; const bf16_t* Bp; long la, lb; int k0;
        if (kt < nk1) { Ap = g.A; Bp = g.B; la = g.lda; lb = g.ldb; k0 = kt * BK; }
        else { Ap = g.A2; Bp = g.B2; la = g.lda2; lb = g.ldb2; k0 = (kt - nk1) * BK; }
#pragma unroll
        for (int i = 0; i < LPTA; ++i) {
            const int q = tid + NT * i;
            int rm = m0 + q / CH; rm = rm < g.M ? rm : g.M - 1;      // clamp: rows past M are never stored
            ra[s][i] = ld16(Ap + (long)rm * la + k0 + (q % CH) * 8);
        }
#pragma unroll
        for (int i = 0; i < LPTB; ++i) {
            const int q = tid + NT * i;
            int rn = n0 + q / CH; rn = rn < g.N ? rn : g.N - 1;
            rb[s][i] = ld16(Bp + (long)rn * lb + k0 + (q % CH) * 8);
        }
    };
    auto write_lds = [&](auto S, int buf) {
        constexpr int s = decltype(S)::value;
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < LPTA; ++i) {
            const int q = tid + NT * i;
            st16(sa + (q / CH) * (BK * 2) + swz_chunk<BK>(q / CH, q % CH) * 16, ra[s][i]);
        }
#pragma unroll
        for (int i = 0; i < LPTB; ++i) {
            const int q = tid + NT * i;
            st16(sb + (q / CH) * (BK * 2) + swz_chunk<BK>(q / CH, q % CH) * 16, rb[s][i]);
        }
    };

    f32x4 acc[4][4];   // acc[ni][mi]: D[n = 16*ni + 4*(lane>>4) + r][m = 16*mi + (lane&15)]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf) {
        const char* sa = smem + buf * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            u32x4 fa[4], fb[4];
            const int c = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int rowa = wm * 64 + i * 16 + fr;
                int rowb = wn * 64 + i * 16 + fr;
                fa[i] = ld16(sa + rowa * (BK * 2) + swz_chunk<BK>(rowa, c) * 16);
                fb[i] = ld16(sb + rowb * (BK * 2) + swz_chunk<BK>(rowb, c) * 16);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma_16x16x32(fb[ni], fa[mi], acc[ni][mi]);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, PF - 1>;
    if (PF == 1) {
        issue_loads(S0{}, kt_begin);
        write_lds(S0{}, 0);
        __syncthreads();
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int buf = (kt - kt_begin) & 1;
            const bool more = (kt + 1 < kt_end);
            if (more) issue_loads(S0{}, kt + 1);
            compute(buf);
            if (more) write_lds(S0{}, buf ^ 1);
            __syncthreads();
        }
    } else {
        // tile kt_begin + j lives in register set j & 1 until it is written to LDS buffer j & 1
        issue_loads(S0{}, kt_begin);
        if (kt_begin + 1 < kt_end) issue_loads(S1{}, kt_begin + 1);
        write_lds(S0{}, 0);
        __syncthreads();
        for (int kt = kt_begin; kt < kt_end; kt += 2) {
            if (kt + 2 < kt_end) issue_loads(S0{}, kt + 2);      // set 0 is free: tile kt already sits in LDS buffer 0
            compute(0);
            if (kt + 1 < kt_end) write_lds(S1{}, 1);
            __syncthreads();
            if (kt + 1 < kt_end) {
                if (kt + 3 < kt_end) issue_loads(S1{}, kt + 3);
                compute(1);
                if (kt + 2 < kt_end) write_lds(S0{}, 0);
                __syncthreads();
            }
        }
    }

    gemm_epilogue<EPI>(g, acc, m0, n0, wm, wn, lane, tile_n);
}

// ---------------------------------------------------------------------------
// LDS-DMA variant for the large-M GEMMs: 256 x 128 output tile, 8 waves (4 x 2, 64 x 64 each), BK = 64, three
// 48 KiB LDS stages filled by `global_load_lds_dwordx4` (no VGPR round trip, no ds_write pass: the ds_write
// path tops out at ~80 B/clk/CU and made the register-staged kernel LDS-bound), counted vmcnt so that the
// copy of tile t+2 stays in flight across the barrier of tile t, one raw s_barrier per K-step.
// LDS image per stage: A rows [256][128 B] then B rows [128][128 B]; a DMA piece = 8 rows = 1 KiB written
// linearly by the 64 lanes, so the conflict-free XOR swizzle (chunk ^= row & 7) is applied on the SOURCE side:
// lane l lands at (row l>>3, physical chunk l&7) and therefore fetches logical chunk (l&7) ^ (l>>3).
// SKEW: fragment reads of the next half K-step are issued before the MFMAs of the current one (the barrier sits
// between the two halves), so LDS latency hides behind matrix work instead of stalling both waves of a SIMD at once.
// MI (round 4): fragments of 16 rows per wave, i.e. the tile is BM = 64 MI rows high — 256 (MI = 4, the original), 192 or 128.  One
// prompt x eight rollouts gives the step's GEMMs M = 2180 / 2048 rows: at N = 2048 a 256-row tile leaves 144 / 128 workgroups on 256
// CUs; 192-row tiles cover M = 2180 with 12 x 16 = 192 workgroups of 3/4 the work each, 128-row tiles cover M = 2048 with exactly 256.
// Same K order per output element in every variant (bit-identical results); the smaller tiles pay more fragment reads per MFMA
// (MI + 4 reads for 4 MI MFMAs) and more L2 -> LDS bytes per flop, which is why they are picked only where they fill the chip.
template <int EPI, int SKEW, int MI = 4>
__global__ __launch_bounds__(512, 2) void gemm_glds_kernel(GemmArgs g) {
    constexpr int BM = 64 * MI, BN = 128, BK = 64;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    BRA_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = uniform_i(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int ntiles = tiles_m * tiles_n;
    int bid = (int)blockIdx.x;
    int kslice = 0;
    if (EPI == EPI_ATOMIC) { kslice = bid / ntiles; bid -= kslice * ntiles; }
    bid = (int)xcd_remap((unsigned)bid, (unsigned)ntiles);
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int gid = bid / per_group;
    const int first_m = gid * GROUP;
    const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int tile_m = first_m + (bid % per_group) % gsize;
    const int tile_n = (bid % per_group) / gsize;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int nk1 = g.K / BK, nk2 = g.K2 / BK;
    int kt_begin = 0, kt_end = nk1 + nk2;
    if (EPI == EPI_ATOMIC && g.split_k > 1) {
        int per = (kt_end + g.split_k - 1) / g.split_k;
        kt_begin = kslice * per;
        kt_end = kt_begin + per < kt_end ? kt_begin + per : kt_end;
        if (kt_begin >= kt_end) return;
    }
    const int nt = kt_end - kt_begin;

    // per-lane source rows of this wave's MI A pieces and 2 B pieces (clamped: rows past M / N are never stored); a piece = 8 rows
    const int prow = lane >> 3, lchunk = ((lane & 7) ^ (lane >> 3)) * 8;
    int rowA[MI], rowB[2];         // rows only: nothing here may spill — a scratch reload in the K loop
#pragma unroll                     // carries a vmcnt(0) that would drain the DMA queue
    for (int j = 0; j < MI; ++j) { int r = m0 + wave * (8 * MI) + j * 8 + prow; rowA[j] = r < g.M ? r : g.M - 1; }
#pragma unroll
    for (int j = 0; j < 2; ++j) { int r = n0 + wave * 16 + j * 8 + prow; rowB[j] = r < g.N ? r : g.N - 1; }
    auto issue = [&](int kt, int stage) {
        char* sa = smem + stage * STAGE + wave * (8 * MI * 128);
        char* sb = smem + stage * STAGE + A_BYTES + wave * (16 * 128);
        const bool main = kt < nk1;
        const bf16_t* Ap = (main ? g.A : g.A2) + (long)(main ? kt : kt - nk1) * BK + lchunk;
        const bf16_t* Bp = (main ? g.B : g.B2) + (long)(main ? kt : kt - nk1) * BK + lchunk;
        const long la = main ? g.lda : g.lda2, lb = main ? g.ldb : g.ldb2;
#pragma unroll
        for (int j = 0; j < MI; ++j) glds16(Ap + (long)rowA[j] * la, sa + j * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(Bp + (long)rowB[j] * lb, sb + j * 1024);
    };
    constexpr int NDMA = MI + 2;   // wave-instructions per K-tile issue: the counted waits below keep ONE tile in flight

    f32x4 acc[4][MI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int stage) {
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            u32x4 fa[MI], fb[4];
            const int c = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int rowa = wm * (16 * MI) + i * 16 + fr;
                fa[i] = ld16(sa + rowa * 128 + swz_chunk<64>(rowa, c) * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int rowb = wn * 64 + i * 16 + fr;
                fb[i] = ld16(sb + rowb * 128 + swz_chunk<64>(rowb, c) * 16);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = mfma_16x16x32(fb[ni], fa[mi], acc[ni][mi]);
        }
    };

    issue(kt_begin, 0);
    if (nt > 1) { issue(kt_begin + 1, 1); wait_vmcnt<NDMA>(); } else { wait_vmcnt<0>(); }
    raw_barrier();
    int stage = 0;
    if (!SKEW) {
        for (int t = 0; t < nt; ++t) {
            int s2 = stage + 2; s2 = s2 >= 3 ? s2 - 3 : s2;
            const bool ahead = t + 2 < nt;
            if (ahead) issue(kt_begin + t + 2, s2);        // stage s2 was last read during step t-1: every wave is past that barrier
            compute(stage);
            if (ahead) wait_vmcnt<NDMA>(); else wait_vmcnt<0>();   // tile t+1 has landed; tile t+2 may still be in flight
            raw_barrier();
            stage = stage + 1 == 3 ? 0 : stage + 1;
        }
    } else {
        auto read_frags = [&](int stg, int kk, u32x4 (&fa)[MI], u32x4 (&fb)[4]) {
            const char* sa = smem + stg * STAGE;
            const char* sb = sa + A_BYTES;
            const int c = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int rowa = wm * (16 * MI) + i * 16 + fr;
                fa[i] = ld16(sa + rowa * 128 + swz_chunk<64>(rowa, c) * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int rowb = wn * 64 + i * 16 + fr;
                fb[i] = ld16(sb + rowb * 128 + swz_chunk<64>(rowb, c) * 16);
            }
        };
        auto mma = [&](const u32x4 (&fa)[MI], const u32x4 (&fb)[4]) {
            setprio(1);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = mfma_16x16x32(fb[ni], fa[mi], acc[ni][mi]);
            setprio(0);
        };
        u32x4 fa0[MI], fb0[4], fa1[MI], fb1[4];
        read_frags(0, 0, fa0, fb0);
        for (int t = 0; t < nt; ++t) {
            int s2 = stage + 2; s2 = s2 >= 3 ? s2 - 3 : s2;
            const int s1 = stage + 1 == 3 ? 0 : stage + 1;
            const bool ahead = t + 2 < nt;
            if (ahead) issue(kt_begin + t + 2, s2);
            read_frags(stage, 1, fa1, fb1);
            mma(fa0, fb0);
            if (ahead) wait_vmcnt<NDMA>(); else wait_vmcnt<0>();
            raw_barrier();                                   // all reads of `stage` are done, tile t+1 has landed
            if (t + 1 < nt) read_frags(s1, 0, fa0, fb0);
            mma(fa1, fb1);
            stage = s1;
        }
    }
    gemm_epilogue_w<EPI, MI>(g, acc, m0 + wm * (16 * MI), n0 + wn * 64, lane, smem + wave * (16 * MI * EPI_RS));
}

// ---------------------------------------------------------------------------
// fp8 x fp8 GEMM on the double-rate matrix path (round 6; BASELINE config 5 "fp8 weights (CDNA4 fp8 MFMA)"):
//   C[m, n] = sa[m] * sb[n] * sum_k A8[m, k] * B8[n, k]  (+ res),   A8 / B8 OCP e4m3 bytes, K % 128 == 0,
// sa = one fp32 scale per activation row (per token), sb = one per weight row (per output feature), both applied to the fp32
// accumulators in the epilogue.  The structure is gemm_glds_kernel's with the byte geometry unchanged — a K-tile is 128 fp8 = the
// same 128 bytes per row, LDS-DMA pieces of 8 rows, XOR swizzle on the source side, three stages, counted vmcnt — and ONE
// v_mfma_scale_f32_16x16x128_f8f6f4 per (16 x 16 fragment pair, K-tile): a lane's operand is the two 16-byte chunks 2 fq, 2 fq + 1 of
// its row (bra_device.h: mfma_fp8_16x16x128).  Per K-tile and wave: (MI + 4) x 2 ds_read_b128 for 4 MI MFMAs of 8 passes each —
// the same LDS bytes per matrix-pipe cycle as the bf16 kernel.  The K-tile is computed in two halves (fragment columns 0-1, then
// 2-3) with the reads of the second half issued before the MFMAs of the first.
struct Gemm8Args {
    const unsigned char* A; long lda;
    const unsigned char* B; long ldb;
    const float* sa;
    const float* sb;
    GemmArgs e;                       // C, ldc, M, N, alpha, bias, res, ldres of the shared epilogues (K / A / B unused)
    int K;
};

template <int EPI, int MI = 4>
__global__ __launch_bounds__(512, 2) void gemm_fp8_kernel(Gemm8Args g8) {
    constexpr int BM = 64 * MI, BN = 128, BKB = 128;                  // BKB: bytes (= fp8 elements) per row and K-tile
    constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB, STAGE = A_BYTES + B_BYTES;
    const GemmArgs& g = g8.e;
    BRA_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = uniform_i(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int ntiles = tiles_m * tiles_n;
    int bid = (int)xcd_remap(blockIdx.x, (unsigned)ntiles);
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int gid = bid / per_group;
    const int first_m = gid * GROUP;
    const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int tile_m = first_m + (bid % per_group) % gsize;
    const int tile_n = (bid % per_group) / gsize;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nt = g8.K / BKB;

    const int prow = lane >> 3, lchunk = ((lane & 7) ^ (lane >> 3)) * 16;
    int rowA[MI], rowB[2];
#pragma unroll
    for (int j = 0; j < MI; ++j) { int r = m0 + wave * (8 * MI) + j * 8 + prow; rowA[j] = r < g.M ? r : g.M - 1; }
#pragma unroll
    for (int j = 0; j < 2; ++j) { int r = n0 + wave * 16 + j * 8 + prow; rowB[j] = r < g.N ? r : g.N - 1; }
    auto issue = [&](int kt, int stage) {
        char* sa_ = smem + stage * STAGE + wave * (8 * MI * 128);
        char* sb_ = smem + stage * STAGE + A_BYTES + wave * (16 * 128);
        const unsigned char* Ap = g8.A + (long)kt * BKB + lchunk;
        const unsigned char* Bp = g8.B + (long)kt * BKB + lchunk;
#pragma unroll
        for (int j = 0; j < MI; ++j) glds16(Ap + (long)rowA[j] * g8.lda, sa_ + j * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(Bp + (long)rowB[j] * g8.ldb, sb_ + j * 1024);
    };
    constexpr int NDMA = MI + 2;

    f32x4 acc[4][MI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    auto read_a = [&](int stg, u32x4 (&fa)[MI][2]) {
        const char* sa_ = smem + stg * STAGE;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int rowa = wm * (16 * MI) + i * 16 + fr;
            fa[i][0] = ld16(sa_ + rowa * 128 + swz_chunk<64>(rowa, 2 * fq) * 16);
            fa[i][1] = ld16(sa_ + rowa * 128 + swz_chunk<64>(rowa, 2 * fq + 1) * 16);
        }
    };
    auto read_b = [&](int stg, int half, u32x4 (&fb)[2][2]) {
        const char* sb_ = smem + stg * STAGE + A_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rowb = wn * 64 + (2 * half + i) * 16 + fr;
            fb[i][0] = ld16(sb_ + rowb * 128 + swz_chunk<64>(rowb, 2 * fq) * 16);
            fb[i][1] = ld16(sb_ + rowb * 128 + swz_chunk<64>(rowb, 2 * fq + 1) * 16);
        }
    };
    auto mma = [&](int half, const u32x4 (&fa)[MI][2], const u32x4 (&fb)[2][2]) {
        setprio(1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[2 * half + i][mi] = mfma_fp8_16x16x128(fb[i][0], fb[i][1], fa[mi][0], fa[mi][1], acc[2 * half + i][mi]);
        setprio(0);
    };

    issue(0, 0);
    if (nt > 1) { issue(1, 1); wait_vmcnt<NDMA>(); } else { wait_vmcnt<0>(); }
    raw_barrier();
    int stage = 0;
    u32x4 fa[MI][2], fb0[2][2], fb1[2][2];
    read_a(0, fa);
    read_b(0, 0, fb0);
    for (int t = 0; t < nt; ++t) {
        int s2 = stage + 2; s2 = s2 >= 3 ? s2 - 3 : s2;
        const int s1 = stage + 1 == 3 ? 0 : stage + 1;
        const bool ahead = t + 2 < nt;
        if (ahead) issue(t + 2, s2);                       // stage s2 was last read during step t - 1: every wave is past that barrier
        read_b(stage, 1, fb1);
        mma(0, fa, fb0);
        mma(1, fa, fb1);
        if (ahead) wait_vmcnt<NDMA>(); else wait_vmcnt<0>();
        raw_barrier();                                     // all reads of `stage` are done, tile t + 1 has landed
        if (t + 1 < nt) { read_a(s1, fa); read_b(s1, 0, fb0); }
        stage = s1;
    }
    // the two scale vectors: lane owns C[mw0 + 16 mi + fr][nw0 + 16 ni + 4 fq .. + 3]
    const int mw0 = m0 + wm * (16 * MI), nw0 = n0 + wn * 64;
    float sm[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) { const int m = mw0 + mi * 16 + fr; sm[mi] = g8.sa[m < g.M ? m : g.M - 1]; }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        float sn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int n = nw0 + ni * 16 + 4 * fq + r; sn[r] = g8.sb[n < g.N ? n : g.N - 1]; }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ni][mi][r] *= sm[mi] * sn[r];
    }
    gemm_epilogue_w<EPI, MI>(g, acc, mw0, nw0, lane, smem + wave * (16 * MI * EPI_RS));
}

// ---------------------------------------------------------------------------
// FOUR waves with LARGE per-wave tiles (late round 4; opt-in: bra_gemm_set_variant(11..14)).  What the tuned library runs on the
// one-prompt shapes (profiles/r4_l_hipblaslt_kernels.txt): one wave per SIMD with the whole register file, a 16 WM x 16 WN tile per
// wave (2 x 2 waves: macro tile 32 WM x 32 WN), i.e. (WM + WN) fragment reads per WM x WN MFMAs — 0.33 for 5 x 8 against 0.5 / 0.75 of
// the eight-wave kernels above — and macro tiles of 160 rows for M = 2180 (14 x 16 = 224 workgroups on 256 CUs).
// Staging as in gemm_glds_kernel: LDS-DMA pieces of 8 rows x 128 B, XOR swizzle on the source side, the stage image is the
// (BM + BN) rows of A then B; every wave issues WM + WN pieces per K-tile; counted waits keep NS - 2 whole tiles in flight.
// K loop = the skewed form: the fragments of the second half K-step are requested before the MFMAs of the first, the barrier sits
// between the halves.  Same K order per output element as every other variant (bit-identical results).
// scheduling request for one MFMA cluster of NM instructions with ND LDS-DMA pieces and NR fragment reads issued in its gaps:
// {1 MFMA, 1 DMA piece + its pointer update} x ND, {1 MFMA, 1 fragment read} x NR, the remaining MFMAs
template <int ND, int NR, int NM>
__device__ __forceinline__ void sched_pipe() {
#ifndef BRA_EMU
    static_assert(ND + NR <= NM, "one memory instruction per MFMA gap");
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);     // VMEM read (the LDS-DMA piece)
        __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);      // VALU (64-bit pointer += BK)
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // DS read
    }
    if constexpr (NM - ND - NR > 0) __builtin_amdgcn_sched_group_barrier(0x8, NM - ND - NR, 0);
#endif
}

template <int EPI, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmArgs g) {
    static_assert(WN % 4 == 0, "the wave epilogue works on 64-column groups");
    constexpr int BM = 32 * WM, BN = 32 * WN, BK = 64;
    constexpr int A_BYTES = BM * BK * 2, STAGE = (BM + BN) * BK * 2;
    constexpr int NS = (160 * 1024) / STAGE >= 4 ? 4 : 3, LOOK = NS - 1;
    constexpr int NDMA = WM + WN;          // 1-KiB pieces per wave per K-tile: (BM + BN) / 8 rows-of-8 over 4 waves
    BRA_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = uniform_i(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int ntiles = tiles_m * tiles_n;
    int bid = (int)xcd_remap(blockIdx.x, (unsigned)ntiles);
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int gid = bid / per_group;
    const int first_m = gid * GROUP;
    const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int tile_m = first_m + (bid % per_group) % gsize;
    const int tile_n = (bid % per_group) / gsize;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int nk1 = g.K / BK, nk2 = g.K2 / BK;
    const int nt = nk1 + nk2;

    // piece j of this wave = rows 8 (wave NDMA + j) .. + 7 of the stage image; running source pointers (+ BK elements per K-tile),
    // rebuilt once where the stream crosses from (A, B) to the LoRA pair (A2, B2)
    const int prow = lane >> 3, lchunk = ((lane & 7) ^ (lane >> 3)) * 8;
    const bf16_t* src[NDMA];
    auto src_init = [&](bool main) {
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const int r = (wave * NDMA + j) * 8 + prow;              // row of the stage image (a piece never straddles A / B: BM % 8 == 0)
            if (r < BM) { int m = m0 + r; m = m < g.M ? m : g.M - 1; src[j] = (main ? g.A : g.A2) + (long)m * (main ? g.lda : g.lda2) + lchunk; }
            else { int n = n0 + r - BM; n = n < g.N ? n : g.N - 1; src[j] = (main ? g.B : g.B2) + (long)n * (main ? g.ldb : g.ldb2) + lchunk; }
        }
    };
    src_init(nk1 > 0);
    auto issue = [&](int kt, int stage) {
        if (kt == nk1 && kt > 0) src_init(false);                     // (wave-uniform, once)
        char* dst = smem + stage * STAGE + wave * (NDMA * 1024);
#pragma unroll
        for (int j = 0; j < NDMA; ++j) { glds16(src[j], dst + j * 1024); src[j] += BK; }
    };
    auto wait_tiles = [&](int fly) {
        if (fly >= 2 && LOOK >= 3) wait_vmcnt<2 * NDMA>();
        else if (fly >= 1) wait_vmcnt<NDMA>();
        else wait_vmcnt<0>();
    };

    f32x4 acc[WN][WM];             // acc[ni][mi]: D[n = 16 ni + 4 (lane >> 4) + r][m = 16 mi + (lane & 15)]
#pragma unroll
    for (int i = 0; i < WN; ++i)
#pragma unroll
        for (int j = 0; j < WM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    auto read_frags = [&](int stg, int kk, u32x4 (&fa)[WM], u32x4 (&fb)[WN]) {
        const char* sa = smem + stg * STAGE;
        const char* sb = sa + A_BYTES;
        const int c = kk * 4 + fq;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int rowa = wm * (16 * WM) + i * 16 + fr;
            fa[i] = ld16(sa + rowa * 128 + swz_chunk<64>(rowa, c) * 16);
        }
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const int rowb = wn * (16 * WN) + i * 16 + fr;
            fb[i] = ld16(sb + rowb * 128 + swz_chunk<64>(rowb, c) * 16);
        }
    };
    auto mma = [&](const u32x4 (&fa)[WM], const u32x4 (&fb)[WN]) {
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) acc[ni][mi] = mfma_16x16x32(fb[ni], fa[mi], acc[ni][mi]);
    };

#pragma unroll
    for (int j = 0; j < LOOK; ++j)
        if (j < nt) issue(j, j);
    wait_tiles((nt < LOOK ? nt : LOOK) - 1);
    raw_barrier();
    u32x4 fa0[WM], fb0[WN], fa1[WM], fb1[WN];
    read_frags(0, 0, fa0, fb0);
    int stage = 0;
    // One K-tile.  With one wave per SIMD nothing else covers the issue slots of the DMA pieces and fragment reads, so they are placed
    // INSIDE the MFMA clusters (sched_group_barrier: one memory instruction per MFMA gap) instead of between them; the steady-state
    // body has no branch between its first DMA piece and its last MFMA (the tail iterations, which issue nothing, are a second loop).
    auto body = [&](int t, auto ISSUE) {
        constexpr bool issuing = decltype(ISSUE)::value;
        int s2 = stage + LOOK; s2 = s2 >= NS ? s2 - NS : s2;
        const int s1 = stage + 1 == NS ? 0 : stage + 1;
        if (issuing && t + LOOK == nk1 && nk1 > 0) src_init(false);   // (wave-uniform, once: the stream crosses to the LoRA pair)
        sched_fence();
        if (issuing) {
            char* dst = smem + s2 * STAGE + wave * (NDMA * 1024);     // slot of K-tile t - 1: every wave is past the barrier behind its last read
#pragma unroll
            for (int j = 0; j < NDMA; ++j) { glds16(src[j], dst + j * 1024); src[j] += BK; }
        }
        read_frags(stage, 1, fa1, fb1);
        mma(fa0, fb0);
        sched_pipe<issuing ? NDMA : 0, NDMA, WM * WN>();
        sched_fence();
        if (issuing) wait_vmcnt<(LOOK - 1) * NDMA>();                 // K-tile t + 1 has landed, LOOK - 1 younger ones stay in flight
        else { const int left = nt - 2 - t; wait_tiles(left < LOOK - 1 ? left : LOOK - 1); }
        raw_barrier();
        sched_fence();
        if (t + 1 < nt) read_frags(s1, 0, fa0, fb0);                  // (uniform; false only in the very last iteration)
        mma(fa1, fb1);
        sched_pipe<0, NDMA, WM * WN>();
        sched_fence();
        stage = s1;
    };
    int t = 0;
    for (; t + LOOK < nt; ++t) body(t, std::true_type{});
    for (; t < nt; ++t) body(t, std::false_type{});
    // one call per 64-column group, the group index a compile-time constant: left as a `#pragma unroll` loop the (now larger) epilogue body
    // was NOT unrolled, `acc[4 * hcol]` became a run-time index and all 160 accumulators of the 5 x 8 tile went through scratch
    // (656 bytes; tests/test_isa.py)
    auto epi_group = [&](auto HC) {
        constexpr int hcol = decltype(HC)::value;
        gemm_epilogue_w<EPI, WM>(g, *reinterpret_cast<f32x4 (*)[4][WM]>(&acc[4 * hcol]), m0 + wm * (16 * WM), n0 + wn * (16 * WN) + 64 * hcol, lane,
                                 smem + wave * (16 * WM * EPI_RS));
    };
    epi_group(std::integral_constant<int, 0>{});
    if constexpr (WN / 4 > 1) epi_group(std::integral_constant<int, 1>{});
    static_assert(WN / 4 <= 2, "two 64-column groups per wave at most");
}

// ---- EPI_SWIGLU epilogue of the ring kernel (round 6): the wave's accumulator blocks are (gate, up, gate, up) of 2 x 16 features for
// 128 rows; act = bf16(bf16(silu(bf16(alpha g))) * bf16(alpha u)) — the roundings of the gate/up GEMM followed by bra_swiglu_fwd —
// goes through a wave-private LDS block [128][80 B] and leaves as 16-byte chunks, 16 rows x 64 bytes per wave instruction.
// Removes the [M, 2 F] store of the projection, its re-read and the SwiGLU launch from the no-grad passes (encoder FFN, reference
// pass, prompt pass): TF:qwen3:81-83, NT-v2 hub FFN.
constexpr int SWG_RS = 80;
__device__ __forceinline__ float silu_epi(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ void ring_epilogue_swiglu(const GemmArgs& g, f32x4 (&acc)[4][8], int mw0, int f0, int lane, char* lw) {
    const int fr = lane & 15, fq = lane >> 4;
    const float alpha = g.alpha;
    char* wp = lw + fr * SWG_RS + 8 * fq;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gb = round_bf(acc[2 * j][mi][r] * alpha), ub = round_bf(acc[2 * j + 1][mi][r] * alpha);
                v[r] = round_bf(silu_epi(gb)) * ub;
            }
            u32x2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
            st8(wp + mi * (16 * SWG_RS) + j * 32, o);
        }
    wave_lds_sync();
    sched_fence();
    const int rrow = lane >> 2, rch = lane & 3;
    const char* rdp = lw + rrow * SWG_RS + 16 * rch;
    bf16_t* cp = (bf16_t*)g.C + ((long)(mw0 + rrow) * g.ldc + f0 + 8 * rch);
    const long cstep = 16 * g.ldc;
    const bool full = mw0 + 128 <= g.M;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const u32x4 xv = ld16(rdp + jj * (16 * SWG_RS));
        if (full || mw0 + rrow + 16 * jj < g.M) st16(cp + jj * cstep, xv);
    }
}

// ---------------------------------------------------------------------------
// 256 x 256 output tile, 8 waves as 2 (M) x 4 (N), each wave 128 x 64 = 8 x 4 fragments of v_mfma_f32_16x16x32_bf16
// (128 accumulator registers), BK = 64.  Against the 256 x 128 kernel above: per-wave tile 128 x 64 instead of
// 64 x 64 (24 instead of 32 fragment reads per 64 MFMAs, half the L2 -> LDS bytes per flop), and a PHASED K loop in
// which the two wave groups of a SIMD pair (waves 0-3 / 4-7) run one barrier apart, so that one wave's 16-MFMA
// cluster always overlaps its partner's fragment reads + LDS-DMA issue instead of both waves stalling together.
//
// LDS: the whole 160 KiB as a RING of ten 16 KiB half-tile slots (128 rows x 128 B each, rows XOR-swizzled on the DMA
// source side as above).  The operand stream is, per K-tile t: B rows 0-127, B rows 128-255, A rows 0-127, A rows 128-255
// (stream index j = 4 t + h, slot j mod 10).  Phase g = 4 t + p (p = 0..3) does
//     fragment reads of K-tile t:  p = 0: B-sub 0 (4 reads) + A-sub 0 (8) | p = 1: B-sub 1 (4) | p = 2: A-sub 1 (8) | p = 3: none
//     LDS-DMA issue of stream index g + 7 (2 wave-instructions per wave)
//     s_waitcnt vmcnt(6): everything up to stream index g + 4 has landed (three half-tiles stay in flight)
//     barrier | 16 MFMAs of quadrant (A-sub, B-sub) = (0,0) (0,1) (1,1) (1,0) | barrier
// Hazards, by construction rather than by luck (MI355X guide: read a staged buffer one phase after the wait that retires it;
// restage a slot >= 2 phases after its last read):  the halves of K-tile t (indices <= 4t+3) are retired by the wait of
// phase 4t-1 and first read in phase 4t;  slot (j mod 10) is rewritten by index j+10 at phase j+3, its last reads
// happened at phase 4t+1 (B halves) / 4t+2 (A halves), i.e. >= 2 phases earlier for every h.
// PH2 = 1: TWO phases per K-tile instead of four — (B-sub 0, B-sub 1, A-sub 0 | quadrants (0,0) (0,1)) and (A-sub 1 | quadrants
// (1,1) (1,0)) — i.e. 32-MFMA clusters (512 cycles) against 16 / 8 fragment reads + two half-tile DMA issues: the load half of
// a phase then fits under the partner wave's MFMA half (with 16-MFMA clusters it took ~1.8x as long and the matrix pipe idled
// 45 % of the time), and there are half as many barriers.  Stream element j = 4t + h is issued at phase floor((j - 6) / 2):
// phase 2t issues (t+1, A halves), phase 2t+1 issues (t+2, B halves); the wait of phase g retires everything up to index
// 2g + 5 (two half-tiles stay in flight); slot reuse distance is >= 2 phases for every half, as in the four-phase form.
template <int EPI, int PH2 = 0>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(GemmArgs g) {
    constexpr int BM = 256, BN = 256, BK = 64, HALF = 128 * BK * 2, NSLOT = 10;
    BRA_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = uniform_i(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int ntiles = tiles_m * tiles_n;
    int bid = (int)blockIdx.x;
    int kslice = 0;
    if (EPI == EPI_ATOMIC) { kslice = bid / ntiles; bid -= kslice * ntiles; }
    bid = (int)xcd_remap((unsigned)bid, (unsigned)ntiles);
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int gid = bid / per_group;
    const int first_m = gid * GROUP;
    const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int tile_m = first_m + (bid % per_group) % gsize;
    const int tile_n = (bid % per_group) / gsize;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int nk1 = g.K / BK, nk2 = g.K2 / BK;
    int kt_begin = 0, kt_end = nk1 + nk2;
    if (EPI == EPI_ATOMIC && g.split_k > 1) {
        int per = (kt_end + g.split_k - 1) / g.split_k;
        kt_begin = kslice * per;
        kt_end = kt_begin + per < kt_end ? kt_begin + per : kt_end;
        if (kt_begin >= kt_end) return;
    }
    const int nt = kt_end - kt_begin;
    const int nstream = 4 * nt;

    // DMA source rows of this wave: half h, piece q (2 per half): local row 16 * wave + 8 * q + (lane >> 3)
    const int prow = lane >> 3, lchunk = ((lane & 7) ^ (lane >> 3)) * 8;
    int rowB[4], rowA[4];          // [half * 2 + piece], clamped (rows past M / N are computed and never stored)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int loc = (i >> 1) * 128 + wave * 16 + (i & 1) * 8 + prow;
        int rb = n0 + loc;
        // EPI_SWIGLU: the tile's 16 blocks of 16 B-rows alternate gate / up rows of the SAME 16 features, so that a wave's accumulator
        // blocks (ni = 0, 1) and (2, 3) hold gate and up of one feature in the same lane and register: the SwiGLU is register-local
        if (EPI == EPI_SWIGLU) rb = tile_n * 128 + (loc >> 5) * 16 + (loc & 15) + ((loc & 16) ? g.swiglu_F : 0);
        rowB[i] = rb < g.N ? rb : g.N - 1;
        int ra = m0 + loc; rowA[i] = ra < g.M ? ra : g.M - 1;
    }
    // stream element (K-tile tt relative to kt_begin, half H) -> slot `slot`; H is a compile-time constant at every call
    // site: phase p always issues half (p + 3) & 3, so no branch and no run-time register indexing surrounds the DMA.
    // Every (half, piece) walks its operand row 64 columns per K-tile, so its source address is a RUNNING pointer (+128 bytes per
    // issue) instead of base + row * ld + k recomputed per issue: the 64-bit multiplies (quarter-rate v_mul_lo_u32 / v_mad_u64_u32,
    // 24 per K-tile) sat in the load half of every phase, the part that has to fit under the partner wave's MFMAs.  The pointers
    // are rebuilt once, when the stream crosses from (A, B) to the second operand pair (A2, B2: the LoRA rank).
    const bf16_t* src[8];          // [half * 2 + piece]
    auto src_init = [&](int h, int kt) {
        const bool main = kt < nk1;
        const long kcol = (long)(main ? kt : kt - nk1) * BK + lchunk;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            if (h < 2) src[h * 2 + pc] = (main ? g.B : g.B2) + kcol + (long)rowB[h * 2 + pc] * (main ? g.ldb : g.ldb2);
            else src[h * 2 + pc] = (main ? g.A : g.A2) + kcol + (long)rowA[(h - 2) * 2 + pc] * (main ? g.lda : g.lda2);
        }
    };
#pragma unroll
    for (int h = 0; h < 4; ++h) src_init(h, kt_begin);
    auto issue = [&](int tt, auto H, int slot) {
        constexpr int h = decltype(H)::value;
        if (tt >= nt) return;
        if (kt_begin + tt == nk1 && tt > 0) src_init(h, nk1);          // (wave-uniform, taken once per half)
        char* dst = smem + slot * HALF + wave * 2048;
        glds16(src[h * 2], dst);
        glds16(src[h * 2 + 1], dst + 1024);
        src[h * 2] += BK;
        src[h * 2 + 1] += BK;
    };
    auto wrap = [&](int sl) { return sl >= NSLOT ? sl - NSLOT : sl; };
    // the wait of phase gph: stream indices <= gph + 4 have landed; issued so far = min(gph + 7, nstream - 1)
    auto wait_phase = [&](int gph) {
        const int fly = nstream - 5 - gph;                           // half-tiles that may stay in flight (cap 3)
        if (fly >= 3) wait_vmcnt<6>(); else if (fly == 2) wait_vmcnt<4>(); else if (fly == 1) wait_vmcnt<2>(); else wait_vmcnt<0>();
    };

    f32x4 acc[4][8];               // acc[ni][mi]: D[n = 16 ni + 4 (lane >> 4) + r][m = 16 mi + (lane & 15)]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    // fragment byte offsets inside a half-tile slot: row = base + 16 i + fr (row & 7 = fr & 7 for every i), k-half kk
    const int sw0 = ((fq ^ (fr & 7)) * 16), sw1 = (((4 + fq) ^ (fr & 7)) * 16);
    const int offA = fr * 128;                                        // + (64 s + 16 i) * 128 + sw{kk}
    const int offB = ((wc & 1) * 64 + fr) * 128;                      // + (32 s + 16 i) * 128 + sw{kk}
    u32x4 fa[4][2], fb[2][2][2];
    auto read_a = [&](int slotA, int s) {
        const char* base = smem + slotA * HALF + offA + s * (64 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[i][0] = ld16(base + i * 2048 + sw0); fa[i][1] = ld16(base + i * 2048 + sw1); }
    };
    auto read_b = [&](int slotB, auto S) {
        constexpr int s = decltype(S)::value;
        const char* base = smem + slotB * HALF + offB + s * (32 * 128);
#pragma unroll
        for (int i = 0; i < 2; ++i) { fb[s][i][0] = ld16(base + i * 2048 + sw0); fb[s][i][1] = ld16(base + i * 2048 + sw1); }
    };
    auto mma = [&](auto SA, auto SB) {
        constexpr int sa = decltype(SA)::value, sb = decltype(SB)::value;
        setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int ia = 0; ia < 4; ++ia)
                    acc[sb * 2 + ib][sa * 4 + ia] = mfma_16x16x32(fb[sb][ib][kk], fa[ia][kk], acc[sb * 2 + ib][sa * 4 + ia]);
        setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    if (PH2) {
        auto wait2 = [&](int gph) {                                       // landed: <= 2 gph + 5; issued: min(2 gph + 7, nstream - 1)
            const int fly = nstream - 6 - 2 * gph;
            if (fly >= 2) wait_vmcnt<4>(); else if (fly == 1) wait_vmcnt<2>(); else wait_vmcnt<0>();
        };
        issue(0, I0{}, 0); issue(0, I1{}, 1); issue(0, I2{}, 2); issue(0, I3{}, 3);
        issue(1, I0{}, 4); issue(1, I1{}, 5);
        wait2(-1);
        bare_barrier();
        if (wr == 1) bare_barrier();
        int rb2 = 0, gp = 0;
#pragma unroll 1
        for (int t = 0; t < nt; ++t) {
            const int sB = wrap(rb2 + (wc >> 1)), sA = wrap(rb2 + 2 + wr);
            // ---- phase 2t
            read_b(sB, I0{}); read_b(sB, I1{}); sched_fence(); read_a(sA, 0);
            issue(t + 1, I2{}, wrap(rb2 + 6)); issue(t + 1, I3{}, wrap(rb2 + 7)); wait2(gp); ++gp;
            sched_fence(); bare_barrier(); wait_lds(); sched_fence();
            mma(I0{}, I0{}); mma(I0{}, I1{});
            sched_fence(); bare_barrier();
            // ---- phase 2t + 1
            read_a(sA, 1);
            issue(t + 2, I0{}, wrap(rb2 + 8)); issue(t + 2, I1{}, wrap(rb2 + 9)); wait2(gp); ++gp;
            sched_fence(); bare_barrier(); wait_lds(); sched_fence();
            mma(I1{}, I1{}); mma(I1{}, I0{});
            sched_fence(); bare_barrier();
            rb2 = rb2 + 4 >= NSLOT ? rb2 + 4 - NSLOT : rb2 + 4;
        }
        if (wr == 0) bare_barrier();
        if constexpr (EPI == EPI_SWIGLU) ring_epilogue_swiglu(g, acc, m0 + wr * 128, tile_n * 128 + wc * 32, lane, smem + wave * (128 * SWG_RS));
        else gemm_epilogue_w<EPI, 8>(g, acc, m0 + wr * 128, n0 + wc * 64, lane, smem + wave * (128 * EPI_RS));
        return;
    }
    // prologue: stream indices 0..6 (K-tile 0 whole, K-tile 1 up to its first A half); K-tile 0 landed before phase 0
    issue(0, I0{}, 0); issue(0, I1{}, 1); issue(0, I2{}, 2); issue(0, I3{}, 3);
    issue(1, I0{}, 4); issue(1, I1{}, 5); issue(1, I2{}, 6);
    wait_phase(-1);
    bare_barrier();
    if (wr == 1) bare_barrier();                 // the second wave group runs one barrier behind the first
    int rbase = 0;                               // slot of (t, h = 0)
    int gph = 0;
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
        const int sB = wrap(rbase + (wc >> 1)), sA = wrap(rbase + 2 + wr);
        // ---- p = 0
        read_b(sB, I0{}); sched_fence(); read_a(sA, 0);
        issue(t + 1, I3{}, wrap(rbase + 7)); wait_phase(gph); ++gph;
        sched_fence(); bare_barrier(); wait_lds(); sched_fence();
        mma(I0{}, I0{});
        sched_fence(); bare_barrier();
        // ---- p = 1
        read_b(sB, I1{});
        issue(t + 2, I0{}, wrap(rbase + 8)); wait_phase(gph); ++gph;
        sched_fence(); bare_barrier(); wait_lds(); sched_fence();
        mma(I0{}, I1{});
        sched_fence(); bare_barrier();
        // ---- p = 2
        read_a(sA, 1);
        issue(t + 2, I1{}, wrap(rbase + 9)); wait_phase(gph); ++gph;
        sched_fence(); bare_barrier(); wait_lds(); sched_fence();
        mma(I1{}, I1{});
        sched_fence(); bare_barrier();
        // ---- p = 3
        issue(t + 2, I2{}, rbase); wait_phase(gph); ++gph;
        sched_fence(); bare_barrier(); sched_fence();
        mma(I1{}, I0{});
        sched_fence(); bare_barrier();
        rbase = rbase + 4 >= NSLOT ? rbase + 4 - NSLOT : rbase + 4;
    }
    if (wr == 0) bare_barrier();                 // re-join the groups (equal barrier counts)

    if constexpr (EPI == EPI_SWIGLU) ring_epilogue_swiglu(g, acc, m0 + wr * 128, tile_n * 128 + wc * 32, lane, smem + wave * (128 * SWG_RS));
    else gemm_epilogue_w<EPI, 8>(g, acc, m0 + wr * 128, n0 + wc * 64, lane, smem + wave * (128 * EPI_RS));
}

// ---------------------------------------------------------------------------
// Skinny GEMM for decode (M <= 16 rows): y[M,N] = alpha*(x W^T + x2 W2^T) (+bias)(+res).
// HBM-bound weight streaming: a workgroup owns 16 output columns; its 4 waves interleave over K in
// 32-deep steps (wave w takes steps w, w+4, ...: one round of the 4 waves reads 256 contiguous bytes of
// each of the 16 weight rows), every lane issues one 16-byte weight load per step straight into the
// MFMA A-fragment (no LDS round trip for data that is used once) with 8 steps in flight; x (M x K, a few
// KB, L2-resident) is read the same way as the B-fragment with rows clamped to M-1.  The four partial
// 16x16 tiles are combined through LDS.
template <int OUTF32>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs g) {
    __shared__ float red[4][64][4];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = (int)blockIdx.x * 16;
    int rn = n0 + fr; rn = rn < g.N ? rn : g.N - 1;
    int rm = fr < g.M ? fr : g.M - 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
        const bf16_t* wp = g.B + (long)rn * g.ldb + fq * 8;
        const bf16_t* xp = g.A + (long)rm * g.lda + fq * 8;
        const int nk = g.K / 32;
        int kt = wave;
        for (; kt + 28 < nk; kt += 32) {            // 8 steps of this wave in flight
            u32x4 w[8], x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { w[u] = ld16(wp + (long)(kt + 4 * u) * 32); x[u] = ld16(xp + (long)(kt + 4 * u) * 32); }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = mfma_16x16x32(w[u], x[u], acc);
        }
        for (; kt < nk; kt += 4) acc = mfma_16x16x32(ld16(wp + (long)kt * 32), ld16(xp + (long)kt * 32), acc);
    }
    if (g.K2 > 0) {
        const bf16_t* wp = g.B2 + (long)rn * g.ldb2 + fq * 8;
        const bf16_t* xp = g.A2 + (long)rm * g.lda2 + fq * 8;
        const int nk = g.K2 / 32;
        for (int kt = wave; kt < nk; kt += 4) acc = mfma_16x16x32(ld16(wp + (long)kt * 32), ld16(xp + (long)kt * 32), acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc[r];
    __syncthreads();
    if (wave != 0) return;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (red[0][lane][r] + red[1][lane][r] + red[2][lane][r] + red[3][lane][r]) * g.alpha;
    // lane holds D[n = n0 + 4*fq + r][m = fr]
    const int m = fr, n = n0 + 4 * fq;
    if (m >= g.M || n >= g.N) return;
    if (g.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] += bf2f(g.bias[n + r]);
    }
    if (OUTF32) {
        float* cp = (float*)g.C + (long)m * g.ldc + n;
        for (int r = 0; r < 4; ++r) if (n + r < g.N) cp[r] = g.accumulate ? cp[r] + v[r] : v[r];
    } else {
        if (g.res) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] = round_bf(v[r]) + bf2f(g.res[(long)m * g.ldres + n + r]);
        }
        bf16_t* cp = (bf16_t*)g.C + (long)m * g.ldc + n;
        if (n + 3 < g.N) { u32x2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); st8(cp, o); }
        else for (int r = 0; r < 4; ++r) if (n + r < g.N) cp[r] = f2bf(v[r]);
    }
}

static int launch_skinny(const GemmArgs& g, int out_f32, bra_stream_t stream) {
    const int grid = (g.N + 15) / 16;
    if (out_f32) BRA_LAUNCH((gemm_skinny_kernel<1>), dim3(grid), dim3(256), 0, stream, g);
    else BRA_LAUNCH((gemm_skinny_kernel<0>), dim3(grid), dim3(256), 0, stream, g);
    return BRA_LAUNCH_STATUS();
}

// tile variant: bit 0 = register prefetch depth 2, bit 1 = 256-row tiles (8 waves).  Chosen per call by
// pick_variant(); bra_gemm_set_variant(v >= 0) pins it (tuning / A-B measurements only).
// process-wide test / benchmark knobs (include/bioreason_hip.h): atomics, so that a launch racing a setter reads a whole value
#ifdef BRA_EMU
template <typename T> struct knob_t { T v; explicit knob_t(T x) : v(x) {} operator T() const { return v; } knob_t& operator=(T x) { v = x; return *this; } };
#else
template <typename T> struct knob_t {
    std::atomic<T> v;
    explicit knob_t(T x) : v(x) {}
    operator T() const { return v.load(std::memory_order_relaxed); }
    knob_t& operator=(T x) { v.store(x, std::memory_order_relaxed); return *this; }
};
#endif
static knob_t<int> g_forced_variant(-1);
static knob_t<int> g_forced_w4(0);             // 1..4: gemm_w4_kernel at 160x256 / 128x256 / 160x128 / 128x128 (bra_gemm_set_variant(11..14))
static knob_t<int> ring_min_fill_pct(60);
static knob_t<int> ring_two_phase(1);
static knob_t<int> ring_row_split(1);

static int pick_variant(const GemmArgs& g) {
    { const int fv = g_forced_variant; if (fv >= 0) return fv >= 6 ? 6 : fv; }
    // measured on MI355X (profiles/r1_gemm_variants.txt): the LDS-DMA kernel with skewed fragment reads wins on every
    // large-M shape of the path (850-1130 TFLOP/s vs 640-870 for the register-staged 128x128 kernel); it needs
    // K % 64 == 0 and enough 256x128 tiles to fill the chip, otherwise the 128x128 kernel keeps more CUs busy
    const long tiles256 = (long)((g.M + 255) / 256) * ((g.N + 127) / 128) * (g.split_k > 1 ? g.split_k : 1);
    if (g.K % 64 == 0 && g.K2 % 64 == 0 && g.split_k <= 1) {
        // 256 x 256 ring kernel: the fastest inner loop (measured 1.07-1.15x the 256 x 128 kernel per tile-flop,
        // profiles/r2_gemm_variants.txt), but half as many tiles — take it when a single partial round still beats two
        // rounds of the smaller tile (>= 140 tiles), or when its rounds fill most of the 256 CUs.  (round 6: the gate is 60 %, was 75 %:
        // SFT's M = 17 440 rows make the N = 2048 projections 552 tiles = 2.16 rounds (72 %), which the old gate sent to 192-row LDS-DMA
        // tiles; on the ring kernel the row split below covers the 40 remaining tiles with a short second launch: down 894 -> 1166,
        // d_gate_up 872 -> 1217, d_qkv 890 -> 1123 TFLOP/s, profiles/r6_k_gemm_sft_shapes.txt)
        const long t = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
        const long rounds = (t + 255) / 256;
        if (t >= 140 && (t <= 256 || 100 * t >= ring_min_fill_pct * rounds * 256)) return 6;
    }
    if (g.K % 64 == 0 && g.K2 % 64 == 0 && tiles256 >= 128) return 5;
    // (round 4) the LDS-DMA kernel at 128-row tiles where 256-row tiles are too few: encoder o / ffn-down at 2052 rows x N = 1024 —
    // 136 tiles: 276 / 409 TFLOP/s against 217 / 293 for the register-staged kernel (profiles/r4_gemm_variants_smallm.txt)
    const long tiles128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * (g.split_k > 1 ? g.split_k : 1);
    if (g.K % 64 == 0 && g.K2 % 64 == 0 && tiles128 >= 128) return 5;
    return 0;
}

template <int BM, int BK, int EPI, int PF>
static int launch_gemm_v(const GemmArgs& g, bra_stream_t stream) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + 127) / 128);
    int grid = tiles;
    if (EPI == EPI_ATOMIC) grid = tiles * (g.split_k > 0 ? g.split_k : 1);
    const size_t smem = 2 * (size_t)(BM + 128) * BK * 2;
    BRA_ALLOW_SMEM((gemm_nt_kernel<BM, BK, EPI, PF>), smem);
    BRA_LAUNCH((gemm_nt_kernel<BM, BK, EPI, PF>), dim3(grid), dim3(BM * 2), smem, stream, g);
    return BRA_LAUNCH_STATUS();
}

// tile height of the LDS-DMA kernel for this call: 256, 192 or 128 rows (see gemm_glds_kernel).  Cost model = rounds of 256 workgroups
// x rows per tile / relative efficiency of the tile; it reproduces the fastest height of every shape measured with
// tools/gemm_variants.py GV_SMALLM=1 (profiles/r4_gemm_variants_smallm.txt: M = 2180 -> 192 rows at N = 2048: +17-20 %; M = 2048 -> 128 rows: +39-42 %);
// ties go to the taller tile.  bra_gemm_set_variant(9 / 10) pins 192 / 128 for A/B runs (5 = 256).
static knob_t<int> g_forced_glds_rows(0);
static int pick_glds_rows(const GemmArgs& g) {
    { const int fr = g_forced_glds_rows; if (fr) return fr; }
    if (g_forced_variant >= 0) return 256;
    const long tn = (g.N + 127) / 128, sk = g.split_k > 1 ? g.split_k : 1;
    const int bms[3] = {256, 192, 128};
    const double eff[3] = {1.0, 0.95, 0.86};
    int best = 256;
    double best_cost = 1e30;
    for (int i = 0; i < 3; ++i) {
        const long t = ((g.M + bms[i] - 1) / bms[i]) * tn * sk;
        const double cost = (double)((t + 255) / 256) * bms[i] / eff[i];
        if (cost < best_cost * 0.97) { best_cost = cost; best = bms[i]; }
    }
    return best;
}

template <int EPI, int MI>
static int launch_glds_mi(const GemmArgs& g, bra_stream_t stream, bool skew) {
    constexpr int BM = 64 * MI;
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + 127) / 128);
    int grid = tiles;
    if (EPI == EPI_ATOMIC) grid = tiles * (g.split_k > 0 ? g.split_k : 1);
    const size_t smem = 3 * (size_t)(BM + 128) * 64 * 2;
    if (skew) {
        BRA_ALLOW_SMEM((gemm_glds_kernel<EPI, 1, MI>), smem);
        BRA_LAUNCH((gemm_glds_kernel<EPI, 1, MI>), dim3(grid), dim3(512), smem, stream, g);
    } else {
        BRA_ALLOW_SMEM((gemm_glds_kernel<EPI, 0, MI>), smem);
        BRA_LAUNCH((gemm_glds_kernel<EPI, 0, MI>), dim3(grid), dim3(512), smem, stream, g);
    }
    return BRA_LAUNCH_STATUS();
}

template <int EPI>
static int launch_glds(const GemmArgs& g, bra_stream_t stream) {
    const bool skew = pick_variant(g) >= 5;
    const int rows = pick_glds_rows(g);
    if (rows == 128) return launch_glds_mi<EPI, 2>(g, stream, skew);
    if (rows == 192) return launch_glds_mi<EPI, 3>(g, stream, skew);
    return launch_glds_mi<EPI, 4>(g, stream, skew);
}

// Which kernel for this call?  A launch takes rounds x (tile flops / sustained rate of the kernel on one CU + a fixed part: prologue
// latency + epilogue), rounds = ceil(tiles / 256) — fitted on tools/gemm_variants.py GV_SMALLM=1 (profiles/r4_n_gemm_variants_w4.txt:
// predicted / measured within 8 %, the argmin is the measured-fastest variant on 19 of the 20 one-prompt shapes).  Rates in TFLOP/s
// per CU: 256 x 256 ring 5.9; four-wave tiles 4.5; LDS-DMA tiles 4.27 / 4.56 / 3.75 at 256 / 192 / 128 rows.  Returns the four-wave
// configuration (1..4) when one of them is the argmin, else 0 = the ring / LDS-DMA choice of pick_variant() stands.
static knob_t<int> g_w4_auto(1);
static int pick_w4(const GemmArgs& g) {
    { const int f = g_forced_w4; if (f) return f; }
    if (!g_w4_auto || g_forced_variant >= 0 || g_forced_glds_rows) return 0;
    if (g.K % 64 || g.K2 % 64 || g.split_k > 1 || g.M < 128 || g.N < 128 || g.K + g.K2 < 256) return 0;
    // only where 256 x 256 tiles cannot fill more than one round of the chip (the one-prompt shapes the model was fitted on): with
    // several rounds the ring kernel's row split covers a partial last round better than ceil() says — SFT's M = 17 440 shapes measured
    // 6 % slower when the model was allowed to move them (254.1 vs 238.7 ms per step)
    if ((long)((g.M + 255) / 256) * ((g.N + 255) / 256) > 256) return 0;
    struct Cand { int bm, bn; double rate; int w4; };
    static const Cand cands[8] = {{256, 256, 5.9, 0}, {256, 128, 4.27, 0}, {192, 128, 4.56, 0}, {128, 128, 3.75, 0},
                                  {160, 256, 4.5, 1}, {128, 256, 4.5, 2}, {160, 128, 4.5, 3}, {128, 128, 4.5, 4}};
    const double kk = 2.0e-6 * (double)(g.K + g.K2);
    int best = 0;
    double best_t = 1e30;
    for (const Cand& c : cands) {
        const long tiles = (long)((g.M + c.bm - 1) / c.bm) * ((g.N + c.bn - 1) / c.bn);
        const double t = (double)((tiles + 255) / 256) * (kk * c.bm * c.bn / c.rate + 3.5);
        if (t < best_t) { best_t = t; best = c.w4; }
    }
    return best;
}

template <int EPI, int WM, int WN>
static int launch_w4_t(const GemmArgs& g, bra_stream_t stream) {
    constexpr int BM = 32 * WM, BN = 32 * WN, STAGE = (BM + BN) * 128, NS = (160 * 1024) / STAGE >= 4 ? 4 : 3;
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const size_t smem = (size_t)NS * STAGE;
    BRA_ALLOW_SMEM((gemm_w4_kernel<EPI, WM, WN>), smem);
    BRA_LAUNCH((gemm_w4_kernel<EPI, WM, WN>), dim3(tiles), dim3(256), smem, stream, g);
    return BRA_LAUNCH_STATUS();
}
template <int EPI>
static int launch_w4(const GemmArgs& g, bra_stream_t stream, int cfg) {
    if constexpr (EPI == EPI_BF16 || EPI == EPI_F32) {
        switch (cfg) {
            case 1: return launch_w4_t<EPI, 5, 8>(g, stream);
            case 2: return launch_w4_t<EPI, 4, 8>(g, stream);
            case 3: return launch_w4_t<EPI, 5, 4>(g, stream);
            default: return launch_w4_t<EPI, 4, 4>(g, stream);
        }
    }
    return BRA_ERR_UNSUPPORTED;
}

template <int EPI>
static int launch_ring(const GemmArgs& g, bra_stream_t stream) {
    const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    const size_t smem = 10 * (size_t)128 * 64 * 2;                  // the whole 160 KiB: ten half-tile slots
    if (ring_two_phase) {
        BRA_ALLOW_SMEM((gemm_ring_kernel<EPI, 1>), smem);
        BRA_LAUNCH((gemm_ring_kernel<EPI, 1>), dim3(tiles), dim3(512), smem, stream, g);
    } else {
        BRA_ALLOW_SMEM((gemm_ring_kernel<EPI, 0>), smem);
        BRA_LAUNCH((gemm_ring_kernel<EPI, 0>), dim3(tiles), dim3(512), smem, stream, g);
    }
    return BRA_LAUNCH_STATUS();
}

template <int BK, int EPI>
static int launch_gemm(const GemmArgs& g, bra_stream_t stream) {
    if (BK == 64 && (EPI == EPI_BF16 || EPI == EPI_F32)) { const int w4 = pick_w4(g); if (w4) return launch_w4<EPI>(g, stream, w4); }
    if (BK == 64 && EPI != EPI_ATOMIC && pick_variant(g) == 6) return launch_ring<EPI>(g, stream);
    if (BK == 64 && pick_variant(g) >= 4) return launch_glds<EPI>(g, stream);
    switch (pick_variant(g)) {
        case 1: return launch_gemm_v<128, BK, EPI, 2>(g, stream);
        case 2: return launch_gemm_v<256, BK, EPI, 1>(g, stream);
        case 3: return launch_gemm_v<256, BK, EPI, 2>(g, stream);
        default: return launch_gemm_v<128, BK, EPI, 1>(g, stream);
    }
}

// Tile-count quantisation of the 256 x 256 ring kernel: t tiles on 256 CUs take ceil(t / 256) rounds, and the N = 2048
// projections of the path (o, down and three of the four input-gradient GEMMs: 54 % of the step's GEMM FLOPs) have
// t = 77 x 8 = 616 = 2.41 rounds — the third round runs 104 tiles on 104 CUs.  When the last round is less than half full the
// rows are split: the ring kernel takes the tile-rows that fill whole rounds, the remaining rows go to the 256 x 128 LDS-DMA
// kernel, whose half-size tiles cover them in about 0.55 of a round (616 tiles: 2.55 rounds instead of 3).
// Returns the rows of the ring part, or 0 (no split).  Rows are independent in every epilogue this is applied to.
static int ring_split_rows(const GemmArgs& g) {
    if (g_forced_variant >= 0 || !ring_row_split || g.K % 64 || g.K2 % 64 || g.split_k > 1) return 0;
    if (pick_variant(g) != 6 || pick_w4(g)) return 0;
    const long tm = (g.M + 255) / 256, tn = (g.N + 255) / 256, t = tm * tn;
    const long R = t / 256, rem = t - R * 256;
    if (R < 1 || rem == 0 || 2 * rem >= 256) return 0;
    const long rm = (R * 256) / tn;                                  // tile-rows that fit into R whole rounds
    if (rm <= 0 || rm >= tm) return 0;
    const long rows2 = g.M - rm * 256;
    const long halves = ((rows2 + 255) / 256) * ((g.N + 127) / 128);
    if (halves < 32) return 0;                                       // (a launch for a handful of tiles costs more than it saves)
    // short, narrow GEMMs (o_proj and its input gradient: N = 2048, K = 2048 + 64) lose more to the second launch than the
    // half-empty round costs: measured 822 -> 786 and 910 -> 827 TFLOP/s with the split, every longer / wider shape gains 1-5 %
    if ((long)g.N * (g.K + g.K2) < 6l * 1024 * 1024) return 0;
    const double cost_split = (double)R + 0.55 * (double)((halves + 255) / 256);
    return cost_split < (double)(R + 1) - 0.15 ? (int)(rm * 256) : 0;
}

template <int EPI>
static int dispatch_bk(const GemmArgs& g, bra_stream_t stream) {
    if (EPI == EPI_BF16 || EPI == EPI_F32) {
        const int m1 = ring_split_rows(g);
        if (m1 > 0) {
            GemmArgs g1 = g, g2 = g;
            g1.M = m1;
            g2.M = g.M - m1;
            g2.A = g.A + (long)m1 * g.lda;
            if (g.A2) g2.A2 = g.A2 + (long)m1 * g.lda2;
            g2.C = EPI == EPI_F32 ? (void*)((float*)g.C + (long)m1 * g.ldc) : (void*)((bf16_t*)g.C + (long)m1 * g.ldc);
            if (g.res) g2.res = g.res + (long)m1 * g.ldres;
            const int e = launch_ring<EPI>(g1, stream);
            if (e) return e;
            return launch_glds<EPI>(g2, stream);
        }
    }
    if (g.K % 64 == 0 && g.K2 % 64 == 0) return launch_gemm<64, EPI>(g, stream);
    return launch_gemm<32, EPI>(g, stream);
}

static int check_common(const GemmArgs& g) {
    if (g.M <= 0 || g.N <= 0) return BRA_ERR_ARG;
    if (g.K < 0 || g.K2 < 0 || (g.K + g.K2) <= 0) return BRA_ERR_ARG;
    if (g.K % 32 || g.K2 % 32) return BRA_ERR_ARG;
    if (g.K && (g.lda % 8 || g.ldb % 8 || !g.A || !g.B)) return BRA_ERR_ARG;
    if (g.K2 && (g.lda2 % 8 || g.ldb2 % 8 || !g.A2 || !g.B2)) return BRA_ERR_ARG;
    return 0;
}

}  // namespace bra

using namespace bra;

#ifdef BRA_DEBUG      // tile-variant knobs for tests and A/B measurements (include/bioreason_hip_debug.h); the product build has none:
                      // the per-shape choice is a pure function of the call's arguments
extern "C" int bra_gemm_set_variant(int v) {
    bra::g_forced_glds_rows = 0;
    bra::g_forced_w4 = 0;
    bra::g_w4_auto = v == -2 ? 0 : 1;                                                      // -2: automatic choice WITHOUT the four-wave kernel (A/B)
    if (v == -2) v = -1;
    if (v >= 11 && v <= 14) { bra::g_forced_w4 = v - 10; v = 5; }                         // four waves, large per-wave tiles (opt-in)
    if (v == 9 || v == 10) { bra::g_forced_glds_rows = v == 9 ? 192 : 128; v = 5; }      // the LDS-DMA kernel at 192 / 128-row tiles
    bra::g_forced_variant = v;
    if (v == 6) bra::ring_two_phase = 0;                 // 6 = four-phase ring, 7 = two-phase ring (A/B measurements)
    if (v == 7 || v < 0) bra::ring_two_phase = 1;
    return 0;
}
extern "C" int bra_gemm_set_glds_rows(int rows) {
    if (rows != 0 && rows != 128 && rows != 192 && rows != 256) return BRA_ERR_ARG;
    bra::g_forced_glds_rows = rows;
    return 0;
}
extern "C" int bra_gemm_set_ring_fill(int pct) { bra::ring_min_fill_pct = pct; return 0; }
extern "C" int bra_gemm_set_epi_lds(int on) {
#ifndef BRA_EMU
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(bra::epi_via_lds), &on, sizeof(int));
#else
    (void)on;
    return 0;
#endif
}
extern "C" int bra_gemm_set_row_split(int on) { bra::ring_row_split = on; return 0; }
#endif

extern "C" int bra_gemm_bf16_nt(const void* A, long lda, const void* B, long ldb, const void* A2, long lda2,
                                const void* B2, long ldb2, int K2, void* C, long ldc, int M, int N, int K,
                                float alpha, const void* bias, const void* res, long ldres, int out_f32,
                                int accumulate, void* stream) {
    GemmArgs g = {};
    g.A = (const bf16_t*)A; g.lda = lda; g.B = (const bf16_t*)B; g.ldb = ldb;
    g.A2 = (const bf16_t*)A2; g.lda2 = lda2; g.B2 = (const bf16_t*)B2; g.ldb2 = ldb2;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.K2 = K2; g.alpha = alpha;
    g.bias = (const bf16_t*)bias; g.res = (const bf16_t*)res; g.ldres = ldres; g.accumulate = accumulate;
    if (M == 0 || N == 0) return 0;
    int e = check_common(g);
    if (e) return e;
    if (!C || ldc % 4) return BRA_ERR_ARG;
    if (out_f32 && res) return BRA_ERR_ARG;
    if (M <= 16) return launch_skinny(g, out_f32, (bra_stream_t)stream);   // decode: weight-streaming kernel
    if (out_f32) return dispatch_bk<EPI_F32>(g, (bra_stream_t)stream);
    return dispatch_bk<EPI_BF16>(g, (bra_stream_t)stream);
}

// fp8 x fp8 GEMM (gemm_fp8_kernel): C [M, N] bf16 (or fp32) = sa[m] sb[n] (A8 B8^T) (+ res).  A8 [M, K], B8 [N, K]: OCP e4m3 bytes, row
// strides in bytes (multiples of 16); K % 128 == 0.  The quantised images come from bra_quant_rows_fp8 / bra_swiglu_quant_fp8.
extern "C" int bra_gemm_fp8_nt(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, void* C, long ldc,
                               int M, int N, int K, const void* res, long ldres, int out_f32, void* stream) {
    if (M == 0 || N == 0) return 0;
    if (M < 0 || N < 0 || K <= 0 || K % 128 || !A8 || !B8 || !sa || !sb || !C || lda % 16 || ldb % 16 || ldc % 4) return BRA_ERR_ARG;
    if (out_f32 && res) return BRA_ERR_ARG;
    Gemm8Args g8 = {};
    g8.A = (const unsigned char*)A8; g8.lda = lda; g8.B = (const unsigned char*)B8; g8.ldb = ldb; g8.sa = sa; g8.sb = sb; g8.K = K;
    g8.e.C = C; g8.e.ldc = ldc; g8.e.M = M; g8.e.N = N; g8.e.K = K; g8.e.alpha = 1.f; g8.e.res = (const bf16_t*)res; g8.e.ldres = ldres;
    // tile height as for the bf16 LDS-DMA kernel: rounds of 256 workgroups x rows / relative efficiency
    const int rows = pick_glds_rows(g8.e);
    bra_stream_t st = (bra_stream_t)stream;
#define BRA_G8(EPI_, MI_)                                                                                             \
    do {                                                                                                              \
        constexpr int BM_ = 64 * MI_;                                                                                 \
        const int tiles = ((M + BM_ - 1) / BM_) * ((N + 127) / 128);                                                  \
        const size_t smem = 3 * (size_t)(BM_ + 128) * 128;                                                            \
        BRA_ALLOW_SMEM((gemm_fp8_kernel<EPI_, MI_>), smem);                                                           \
        BRA_LAUNCH((gemm_fp8_kernel<EPI_, MI_>), dim3(tiles), dim3(512), smem, st, g8);                                \
        return BRA_LAUNCH_STATUS();                                                                                   \
    } while (0)
    if (out_f32) { if (rows == 128) BRA_G8(EPI_F32, 2); if (rows == 192) BRA_G8(EPI_F32, 3); BRA_G8(EPI_F32, 4); }
    if (rows == 128) BRA_G8(EPI_BF16, 2);
    if (rows == 192) BRA_G8(EPI_BF16, 3);
    BRA_G8(EPI_BF16, 4);
#undef BRA_G8
}

// gate/up projection with the SwiGLU in its epilogue (gemm_ring_kernel<EPI_SWIGLU>): act [M, F] bf16 = swiglu(A W^T + A2 B2^T), W [2 F, K]
// = [gate rows | up rows] (B2 [2 F, K2] likewise: the LoRA rank part, optional).  The values are those of bra_gemm_bf16_nt followed by
// bra_swiglu_fwd, bit for bit; the [M, 2 F] intermediate never exists — for passes that keep nothing for a backward (encoder FFN,
// reference pass, prompt pass).  BRA_ERR_UNSUPPORTED unless K % 64 == 0, K2 % 64 == 0, F % 128 == 0, M > 16.
extern "C" int bra_gemm_swiglu_bf16_nt(const void* A, long lda, const void* W, long ldw, const void* A2, long lda2, const void* B2, long ldb2,
                                       int K2, void* C, long ldc, int M, int F, int K, float alpha, void* stream) {
    if (M == 0 || F == 0) return 0;
    GemmArgs g = {};
    g.A = (const bf16_t*)A; g.lda = lda; g.B = (const bf16_t*)W; g.ldb = ldw;
    g.A2 = (const bf16_t*)A2; g.lda2 = lda2; g.B2 = (const bf16_t*)B2; g.ldb2 = ldb2;
    g.C = C; g.ldc = ldc; g.M = M; g.N = 2 * F; g.K = K; g.K2 = K2; g.alpha = alpha; g.swiglu_F = F;
    int e = check_common(g);
    if (e) return e;
    if (!C || ldc % 8 || ((size_t)C & 15)) return BRA_ERR_ARG;
    if (K % 64 || K2 % 64 || F % 128 || M <= 16) return BRA_ERR_UNSUPPORTED;
    return launch_ring<EPI_SWIGLU>(g, (bra_stream_t)stream);
}

extern "C" int bra_gemm_bf16_nt_splitk(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M,
                                       int N, int K, float alpha, int split_k, void* stream) {
    GemmArgs g = {};
    g.A = (const bf16_t*)A; g.lda = lda; g.B = (const bf16_t*)B; g.ldb = ldb;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.split_k = split_k < 1 ? 1 : split_k;
    if (M == 0 || N == 0) return 0;
    int e = check_common(g);
    if (e) return e;
    if (!C) return BRA_ERR_ARG;
    return dispatch_bk<EPI_ATOMIC>(g, (bra_stream_t)stream);
}

extern "C" int bra_lmhead_lse_partials(const void* H, long ldh, const void* E, long lde, int M, int V, int K,
                                       const int* tgt, float* part_max, float* part_sum, float* tgt_logit,
                                       void* stream) {
    GemmArgs g = {};
    g.A = (const bf16_t*)H; g.lda = ldh; g.B = (const bf16_t*)E; g.ldb = lde;
    g.M = M; g.N = V; g.K = K; g.alpha = 1.f;
    g.tgt = tgt; g.part_max = part_max; g.part_sum = part_sum; g.tgt_logit = tgt_logit;
    g.nchunk = (V + 63) / 64;
    if (M == 0) return 0;
    int e = check_common(g);
    if (e) return e;
    if (!tgt || !part_max || !part_sum || !tgt_logit) return BRA_ERR_ARG;
    return dispatch_bk<EPI_LSE>(g, (bra_stream_t)stream);
}

extern "C" int bra_lmhead_dlogits(const void* H, long ldh, const void* E, long lde, int M, int V, int K,
                                  const int* tgt, const float* lse, const float* coef, void* dlogits, long ldd,
                                  void* stream) {
    GemmArgs g = {};
    g.A = (const bf16_t*)H; g.lda = ldh; g.B = (const bf16_t*)E; g.ldb = lde;
    g.C = dlogits; g.ldc = ldd; g.M = M; g.N = V; g.K = K; g.alpha = 1.f;
    g.tgt = tgt; g.lse = lse; g.coef = coef;
    if (M == 0) return 0;
    int e = check_common(g);
    if (e) return e;
    if (!tgt || !lse || !coef || !dlogits || ldd % 4) return BRA_ERR_ARG;
    return dispatch_bk<EPI_DLOGIT>(g, (bra_stream_t)stream);
}
