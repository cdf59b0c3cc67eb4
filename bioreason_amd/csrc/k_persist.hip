// k_persist.hip — persistent-grid building blocks of the rollout's token loop (HF `_sample` loop body,
// TF:generation/utils.py:2876-2925): the in-launch grid barrier / hand-off protocol of bra_gridsync.h and a probe that
// measures and word-checks it on the device (tools/gridbar_probe.py).
#define BRA_LANE_OPAQUE 1
#include "bra_gridsync.h"
#include "bra_decattn.h"
#include "bra_decgemm.h"
#include "bra_api_internal.h"

namespace bra {

#ifndef BRA_EMU
// One iteration = what a phase boundary of the persistent decode step does: every workgroup publishes a small slot (its 8 or
// 16 output columns), the grid synchronises, every workgroup reads ALL slots (the activation vector of the next projection).
//   mode 0: barrier only                     mode 1: sc1 stores -> barrier -> sc1 loads (the protocol of bra_gridsync.h)
//   mode 2: plain stores + agent release fence -> barrier -> agent acquire fence + plain loads (the guide's fence form)
//   mode 3: mode 1 with `wchunks` x 16 B per thread of a read-once weight stream requested BEFORE the barrier and consumed
//           after it (the prefetch the persistent step relies on)
// Every word read is compared with the value the producer wrote for THIS iteration (stale or torn data is counted in errs[0]);
// the slots are double-buffered by iteration parity, as the step's buffers are by phase.
template <int NT>
__global__ __launch_bounds__(NT) void gridbar_probe_kernel(GridSync* gs, unsigned* buf /* [2][nwg][32] */, unsigned* errs,
                                                           const u32x4* wts, unsigned long wts_chunks, int iters, int mode_in,
                                                           int wchunks, unsigned timeout_ticks) {
    int mode = mode_in;
    __shared__ unsigned flag;
    const int nwg = (int)gridDim.x, wg = (int)blockIdx.x, tid = (int)threadIdx.x;
    unsigned epoch = 0, bad = 0, sink = 0;
    const __amdgpu_buffer_rsrc_t rs = xs_rsrc(buf);
    const unsigned nwords = (unsigned)nwg * 32u;
    for (int it = 0; it < iters; ++it) {
        unsigned* slot = buf + (size_t)(it & 1) * nwords;
        u32x4 w[8];
        const bool spoll = mode >= 4;       // modes 4 / 5 = modes 1 / 3 with the barrier polled through the scalar path
        if (mode == 5) mode = 3;
        if (mode == 4) mode = 1;
        if (mode == 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned long c = ((unsigned long)it * nwg + wg) * (unsigned long)(NT * 8) + (unsigned long)u * NT + tid;
                w[u] = u < wchunks ? ld16_nt(wts + c % wts_chunks) : u32x4{0u, 0u, 0u, 0u};
            }
        }
        if (mode != 0 && tid < 16) {
            u32x2 v;
            v.x = (unsigned)it * 1000003u + (unsigned)wg * 64u + 2u * tid;
            v.y = v.x + 1u;
            if (mode == 2) *reinterpret_cast<u32x2*>(slot + wg * 32 + 2 * tid) = v;
            else xs_store8(rs, (unsigned)(((it & 1) * nwords + wg * 32 + 2 * tid) * 4), v);
        }
        if (mode == 2) {
            __syncthreads();
            if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        } else if (mode == 1) {
            gs_drain();
        } else if (mode == 3) {
            if (tid < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the storing wave drains (its weight chunks with it)
        }
        if (spoll ? !grid_barrier<1>(gs, epoch, nwg, timeout_ticks, &flag) : !grid_barrier<0>(gs, epoch, nwg, timeout_ticks, &flag)) break;
        if (mode == 2) {
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
        }
        if (mode != 0) {
            for (unsigned c = tid; c < nwords / 4; c += NT) {
                u32x4 v;
                if (mode == 2) v = *reinterpret_cast<const u32x4*>(slot + 4 * c);
                else v = xs_load16(rs, (unsigned)(((it & 1) * nwords + 4 * c) * 4));
                const unsigned src = (4 * c) >> 5, j = (4 * c) & 31u;
                const unsigned want = (unsigned)it * 1000003u + src * 64u + j;
                bad += (v.x != want) + (v.y != want + 1u) + (v.z != want + 2u) + (v.w != want + 3u);
            }
        }
        if (mode == 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) sink ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w;
        }
    }
    if (bad) atomicAdd(errs, bad);
    if (sink == 0x12345678u) errs[2] = sink;        // keeps the weight stream alive
    if (tid == 0 && wg == 0) errs[1] = epoch;
}
#endif


// ---------------------------------------------------------------------------------------------------------------------
// Chained launches (probe): kernel k of a dependent chain is launched on stream k % 2, requests its read-once "weights" at once,
// THEN waits on a device-side counter for kernel k - 1 (all of its workgroups have published), reads the predecessor's slots
// (sc1), publishes its own and bumps its counter.  Two kernels are resident at a time (the in-stream order keeps k + 2 behind k), so
// the weight stream of k + 1 overlaps the latency chain of k — what a launch boundary forbids and a grid barrier paid for with
// queueing (NOTES.md).  mode 0: the same kernels without the wait, on ONE stream (ordinary dependent launches).
#ifndef BRA_EMU
template <int NT, int WCH>
__global__ __launch_bounds__(NT, 2) void chain_probe_kernel(unsigned* done /* [n] counters, 64-byte apart */, unsigned* buf /* [2][nwg][32] */,
                                                            unsigned* errs, const u32x4* wts, unsigned long wts_chunks, int k, int chained,
                                                            unsigned timeout_ticks) {
    __shared__ unsigned ok_flag;
    const int nwg = (int)gridDim.x, wg = (int)blockIdx.x, tid = (int)threadIdx.x;
    u32x4 w[WCH];
#pragma unroll
    for (int u = 0; u < WCH; ++u) {
        const unsigned long c = ((unsigned long)k * nwg + wg) * (unsigned long)(NT * WCH) + (unsigned long)u * NT + tid;
        w[u] = ld16_nt(wts + c % wts_chunks);
    }
    sched_fence();
    if (chained && k > 0) {
        if (tid == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned ok = 1u;
            while (gs_load(done + 16 * (k - 1)) < (unsigned)nwg) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)timeout_ticks) { ok = 0u; gs_store(errs + 3, (unsigned)k); break; }
            }
            ok_flag = ok;
        }
        __syncthreads();
        if (!ok_flag) return;
    }
    const __amdgpu_buffer_rsrc_t rs = xs_rsrc(buf);
    const unsigned nwords = (unsigned)nwg * 32u;
    unsigned bad = 0, sink = 0;
    if (k > 0) {
        for (unsigned c = tid; c < nwords / 4; c += NT) {
            const u32x4 v = xs_load16(rs, (unsigned)((((k - 1) & 1) * nwords + 4 * c) * 4));
            const unsigned src = (4 * c) >> 5, j = (4 * c) & 31u;
            const unsigned want = (unsigned)(k - 1) * 1000003u + src * 64u + j;
            bad += (v.x != want) + (v.y != want + 1u) + (v.z != want + 2u) + (v.w != want + 3u);
        }
    }
#pragma unroll
    for (int u = 0; u < WCH; ++u) sink ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w;
    if (tid < 16) {
        u32x2 v;
        v.x = (unsigned)k * 1000003u + (unsigned)wg * 64u + 2u * tid + (sink == 0x12345678u ? 1u : 0u);
        v.y = v.x + 1u;
        xs_store8(rs, (unsigned)(((k & 1) * nwords + wg * 32 + 2 * tid) * 4), v);
        gs_drain();
    }
    if (bad) atomicAdd(errs, bad);
    __syncthreads();
    if (tid == 0) gs_add(done + 16 * k, 1u);
}
#endif

// =====================================================================================================================
// The decode step of the shared-prefix rollout as ONE launch: all decoder layers of Qwen3DecoderLayer.forward with a KV cache
// (TF:qwen3:294-323) for the new token of every sequence — what bra_qwen_decode_step_one issues as six launches per layer.
// One workgroup per CU (grid = number of 16-column qkv tiles = number of 8-column o / down tiles), eight waves; a layer is six
// phases separated by grid barriers (bra_gridsync.h):
//     qkv projection | attention items | merge | o projection + residual | gate/up + SwiGLU | down projection + residual
// Every phase runs the SAME tiles, K split, reduction order and epilogue as the launched kernels (dec_gemm2_kernel FAST shapes,
// dec_attn_item, dec_attn_merge_one): the step is bit-identical to the launched path and is tested against it.  What the single
// launch buys is not cheaper synchronisation (a barrier + hand-off costs about what a launch boundary does: ~3.8 us measured,
// tools/gridbar_probe.py) but that the weight stream no longer stops at the boundaries: with PF the fragment-packed weights of
// the NEXT phases are requested before the barrier that hands over their activations and sit in registers when it opens.
// Activations, statistics and attention partials cross workgroups through sc1 stores / sc1 loads; the K/V caches, weights and
// rope rows come from earlier launches (plain / non-temporal loads).
struct PLayer {                           // device-side layer table (one record per decoder layer)
    const bf16_t *Wqkv, *Wo, *Wgu, *Wd;     // fragment-packed rollout weights (ln1 folded into Wqkv, ln2 into Wgu)
    const bf16_t *qn, *kn;                  // per-head RMSNorm weights
    const bf16_t *kp, *vtp;                 // prompt K [R, Hkv, P, hd] and V^T [R, Hkv, hd, pitch]
    bf16_t *kc, *vct;                       // completion K cache and transposed completion V cache
};

constexpr int kPersistMaxLayers = 40;     // 40 x 80 B + the rest of the record < the 4 KiB kernel-argument segment
struct PersistArgs {
    // the layer table travels IN the kernel arguments: pointers read from the argument segment are known to be global, pointers
    // read from a table in memory are not — every access through them would be a FLAT instruction, which counts in lgkmcnt too, so
    // each LDS wait of a tile barrier would wait for the whole weight stream in flight
    PLayer layers[kPersistMaxLayers]; int L, M;
    DecOneArgs att;                         // per-layer pointer fields are patched from the table
    bf16_t *x, *h, *act; float *ssx, *ssh; int nss; float eps;
    GridSync* sync; unsigned timeout_ticks;
    int ph_lo, ph_hi;                       // phases [ph_lo, ph_hi) of the 6 L run in this launch (diagnostics; the whole step: 0, 6 L)
    unsigned long long* stamps;             // diagnostics (bra_persist_set_stamps): 100 MHz wall-clock stamps of workgroup 0 / wave 0,
                                            // [6 L][4]: phase start, own requests issued + computed, stores issued, stores drained
};

#ifndef BRA_EMU
template <int NL> struct Frag { u32x4 v[NL]; };

// wait until at most N of this wave's vector-memory operations are outstanding: with N look-ahead requests issued LAST, everything
// older (the phase's stores among them) has completed while the look-ahead stays in flight across the grid barrier
template <int N> __device__ __forceinline__ void wait_all_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// the wave's NL consecutive KiB blocks of one packed column tile (dec_gemm2_kernel FAST: tile_base + wo[u])
// blocks [U0, U1) of the NL: a tile can be requested in instalments (look-ahead windows hold ~48 KB per CU: what drains from
// the CU's request queue while the grid barrier is pending)
template <int NW, int NL, int U0 = 0, int U1 = NL>
__device__ __forceinline__ void pw_issue(Frag<NL>& w, const bf16_t* W, const int tile, const int wave, const int lane) {
    // (every path DEFINES the fragment: a fragment left untouched on one path would be carried around the layer loop)
    if (NW < 8 && wave >= NW) {
#pragma unroll
        for (int u = U0; u < U1; ++u) w.v[u] = u32x4{0u, 0u, 0u, 0u};
        return;
    }
    const bf16_t* p = W + (long)tile * (NW * NL * 512) + (unsigned)(wave * (NL * 512) + lane * 8);
#pragma unroll
    for (int u = U0; u < U1; ++u) w.v[u] = ld16_nt(p + u * 512);
}

// the wave's activation fragments (rows of another workgroup's output: sc1)
template <int MODE, int NW, int NL>
__device__ __forceinline__ void px_load(Frag<NL>& x, const bf16_t* X, const int ldx, const int M, const int wave, const int lane) {
    if (NW < 8 && wave >= NW) {
#pragma unroll
        for (int u = 0; u < NL; ++u) x.v[u] = u32x4{0u, 0u, 0u, 0u};
        return;
    }
    constexpr int KS = MODE ? 64 : 32;
    const int fr = lane & 15, fq = lane >> 4;
    const int lrow = MODE ? (fr & 7) : fr;
    const int koff = MODE ? ((fr >> 3) * 32 + fq * 8) : fq * 8;
    const int xr = lrow < M ? lrow : M - 1;
    const unsigned off = dg2_mul24(xr, ldx) + (unsigned)(koff + wave * (NL * KS));
#pragma unroll
    for (int u = 0; u < NL; ++u) x.v[u] = xld16<1>(X, (off + (unsigned)(u * KS)) * 2u);
}

// RMSNorm statistics partials of the input rows (wave 0; dec_gemm2_kernel NORM == 2)
__device__ __forceinline__ void ps_load(f32x4 (&sq)[8], const float* ss, const int nss, const int M, const int wave, const int lane) {
    if (wave != 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int per = nss >> 3;
    const int srow = lane >> 3;
    const int sr = srow < M ? srow : M - 1;
    const unsigned off = dg2_mul24(sr, nss) + (unsigned)((lane & 7) * per);
#pragma unroll
    for (int i = 0; i < 8; ++i) sq[i] = __builtin_bit_cast(f32x4, xld16<1>(ss, (off + (unsigned)(4 * i < per ? 4 * i : 0)) * 4u));
}

// residual words of the epilogue lanes of (tile): dec_gemm2_kernel's `resv`
template <int MODE>
__device__ __forceinline__ u32x2 pr_load(const bf16_t* res, const int ldres, const int M, const int N, const int tile, const int lane) {
    constexpr int NCOL = MODE ? 8 : 16;
    const int fr = lane & 15, fq = lane >> 4;
    const int em = fr < M ? fr : M - 1;
    int en = tile * NCOL + 4 * fq; en = en + 3 < N ? en : N - 4;
    return xld8<1>(res, (dg2_mul24(em, ldres) + (unsigned)en) * 2u);
}

// one column tile: MFMA chain over the wave's K slice, K-reduction through LDS in wave order, epilogue by wave it % NW —
// the body of dec_gemm2_kernel's compute()
template <int MODE, int NORM, int ACT, int NW, int NL>
__device__ __forceinline__ void pg_tile(const Frag<NL>& w, const Frag<NL>& x, const DecGemm2Args& g, const int tile, const int it,
                                        const u32x2& resv, const f32x4 (&sq)[8], float (&red)[2][8][64][4], float (&rs_lds)[8],
                                        const int wave, const int lane) {
    const int fr = lane & 15;
    if (wave < NW) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NL; ++u) acc = mfma_16x16x32(w.v[u], x.v[u], acc);
        if (NORM == 2 && it == 0 && wave == 0) {
            const float rs = dg2_fold_rstd(sq, g.nss_in >> 3, g.inv_K, g.eps);
            if ((lane & 7) == 0) rs_lds[lane >> 3] = rs;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[it & 1][wave][lane][r] = acc[r];
    }
    raw_barrier();
    if (wave == it % NW) {
        float v[4];
        dg2_reduce<NW>(red[it & 1], lane, v);
        if (NORM == 2) {
            const int mrow = MODE ? (fr & 7) : fr;
            const float rsf = rs_lds[mrow < g.M ? mrow : g.M - 1];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= rsf;
        }
        dg2_epilogue<MODE, ACT, 0, 1, 1>(g, v, tile, lane, true, resv);
    }
}

// PF: 0 = every phase requests its weights when it starts (the launched path inside one launch); 1 = weights of the following
// phases requested ahead (see the schedule in the loop); 2 = also the K / V^T fragments of the wave's attention item
template <int H, int NQ, int NKV, int F, int HD, int G, int PF>
__global__ __launch_bounds__(512) void decode_persist_kernel(PersistArgs a) {
    constexpr int NQKV = NQ + 2 * NKV;
    constexpr int NWG = NQKV / 16;                               // workgroups = qkv tiles
    constexpr int NL_QKV = H / 256;                              // 8 waves x NL x 32-deep steps
    constexpr int NW_O = (NQ / 64 >= 64) ? 8 : 4, NL_O = NQ / 64 / NW_O;
    constexpr int NL_D = F / 64 / 8;
    constexpr int NT_GU = 2 * F / 16 / NWG;
    constexpr int GU_A = 3;                                      // blocks of the first gate/up tile requested in the merge phase's window
    static_assert(H % 256 == 0 && (NL_QKV == 4 || NL_QKV == 8 || NL_QKV == 12), "qkv / gate-up: one register round of 8 waves");
    static_assert(H / 8 == NWG && (2 * F / 16) % NWG == 0 && NT_GU == 3, "tiles per workgroup: 1 qkv, 1 o, 3 gate/up, 1 down");
    static_assert(NQ % (64 * NW_O) == 0 && (NL_O == 4 || NL_O == 8 || NL_O == 12) && F % 512 == 0 && (NL_D == 4 || NL_D == 8 || NL_D == 12), "o / down");
    __shared__ float red[2][8][64][4];
    __shared__ float rs_lds[8];
    __shared__ unsigned bar_flag;
    // wave / workgroup ids are re-derived behind an (empty) volatile asm at the top of every phase: the optimiser otherwise treats
    // every id-derived offset of every phase as invariant of the layer loop, computes all of them at kernel entry and spills them
    // around the loop (several hundred dwords of scratch traffic inside the phases)
#define BRA_PIDS()                                                                                             \
    int wave, wg;                                                                                              \
    { int tw_ = (int)threadIdx.x >> 6; asm volatile("" : "+v"(tw_)); wave = __builtin_amdgcn_readfirstlane(tw_);   \
      wg = (int)blockIdx.x; asm volatile("" : "+s"(wg)); }
    unsigned epoch = 0;
    int phases = 0;
    const int M = a.M;
    DecOneArgs at = a.att;
    const int t = at.t_ptr ? at.t_ptr[0] : at.t;
    at.t = t; at.t_ptr = nullptr;
    {
        const int ncc = (t + 63) / 64;
        at.ncc_grid = ncc; at.inv_ncc = 1.f / (float)(ncc > 0 ? ncc : 1);
    }
    const int nitems = at.R * at.copies * at.Hq + at.npc * at.Hkv * at.R + at.ncc_grid * at.copies * at.Hkv * at.R;
    const u32x2 zero2 = {0u, 0u};
    f32x4 sq0[8];                             // (phases without a norm)
#pragma unroll
    for (int i = 0; i < 8; ++i) sq0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frag<NL_QKV> wq;                          // the only fragment carried from one layer to the next (PF)
#define BRA_PSTAMP(k_)                                                                        \
    do {                                                                                      \
        if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[phases * 4 + (k_)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    // phase window (diagnostics): a phase outside [ph_lo, ph_hi) is skipped, a barrier is taken only between two phases that run
    // (PF > 0 always runs from phase 0: the tests fold away, no fragment becomes conditionally defined)
#define BRA_PRUN() (PF > 0 || (phases >= a.ph_lo && phases < a.ph_hi))
#define BRA_PBAR()                                                                            \
    do {                                                                                      \
        const bool both_ = PF > 0 || (phases >= a.ph_lo && phases + 1 < a.ph_hi);             \
        ++phases;                                                                             \
        if (phases >= a.ph_hi) return;                                                        \
        if (both_ && !grid_barrier(a.sync, epoch, NWG, a.timeout_ticks, &bar_flag)) return;   \
    } while (0)
    const PLayer* Ls = a.layers;            // (in the kernel-argument segment)
    if (PF) { BRA_PIDS(); pw_issue<8, NL_QKV>(wq, Ls[0].Wqkv, wg, wave, lane_id()); sched_fence(); }
    for (int l = 0; l < a.L; ++l) {
        const PLayer Lr = Ls[l];
        Frag<NL_QKV> wg0, wg1;                // requested and consumed inside one layer: declared here, undefined at its start
        Frag<NL_O> wo;
        Frag<NL_D> wd;
        at.qw = Lr.qn; at.kw = Lr.kn; at.kp = Lr.kp; at.vtp = Lr.vtp; at.kc = Lr.kc; at.vct = Lr.vct;
        // ------------------------------------------------------------------ qkv = rmsnorm(x) Wqkv^T   (ln1 folded, rstd in the epilogue)
        {
            u32x4 kf[4][HD / 32], vf[HD / 16][2];
            DecItem it0;
            it0.kind = 0;
            if (BRA_PRUN()) {
            BRA_PIDS();
            DecGemm2Args g = {a.x, H, a.ssx, a.nss, nullptr, a.eps, Lr.Wqkv, H, nullptr, 0, (void*)at.qkv, NQKV, nullptr, 0, M, NQKV, H, 3, nullptr, nullptr, 1.f / (float)H};
            if (PF >= 2) {                  // the wave's attention item: its K / V^T chunk does not depend on the new token
                const int iw = wg + NWG * wave;
                if (iw < nitems) dec_item_decode<HD, G>(at, iw, t, it0);
                if (it0.kind == 2) item_kv_issue<HD>(it0.kbase, it0.kss, it0.vbase, it0.vsd, it0.key0, it0.nkeys, kf, vf);
                sched_fence();
            }
            BRA_PSTAMP(0);
            if (!PF) pw_issue<8, NL_QKV>(wq, Lr.Wqkv, wg, wave, lane_id());
            Frag<NL_QKV> xf;
            px_load<0, 8, NL_QKV>(xf, a.x, H, M, wave, lane_id());
            f32x4 sq[8];
            ps_load(sq, a.ssx, a.nss, M, wave, lane_id());
            pg_tile<0, 2, 0, 8, NL_QKV>(wq, xf, g, wg, 0, zero2, sq, red, rs_lds, wave, lane_id());
            BRA_PSTAMP(2);
            if (wave == 0) gs_drain();
            BRA_PSTAMP(3);
            }
            BRA_PBAR();
            // -------------------------------------------------------------- attention items: q/k norm + RoPE, cache append, partials
          if (BRA_PRUN()) {
            BRA_PIDS();
            BRA_PSTAMP(0);
            if (PF >= 2) {
                if (wg + NWG * wave < nitems) dec_item_run<HD, G, 1, 1>(at, it0, t, kf, vf);
                for (int i = wg + NWG * (wave + 8); i < nitems; i += NWG * 8) {
                    DecItem d;
                    dec_item_decode<HD, G>(at, i, t, d);
                    dec_item_run<HD, G, 1, 0>(at, d, t, kf, vf);
                }
            } else {
                for (int i = wg + NWG * wave; i < nitems; i += NWG * 8) {
                    DecItem d;
                    dec_item_decode<HD, G>(at, i, t, d);
                    dec_item_run<HD, G, 1, 0>(at, d, t, kf, vf);
                }
            }
            BRA_PSTAMP(2);
            gs_drain();
            BRA_PSTAMP(3);
          }
            BRA_PBAR();
        }
        // ------------------------------------------------------------------ merge of the partials -> o
        if (BRA_PRUN()) {
        BRA_PIDS();
        BRA_PSTAMP(0);
        // look-ahead requests (PF) go out BEHIND the phase's own requests: loads return in order, so a weight request issued first
        // would have to land before the phase's own operands can be used
        auto look_ahead = [&]() {           // o projection + first gate/up tile
            if (PF) {
                pw_issue<NW_O, NL_O>(wo, Lr.Wo, wg, wave, lane_id());
                pw_issue<8, NL_QKV, 0, GU_A>(wg0, Lr.Wgu, wg, wave, lane_id());
                sched_fence();
            }
        };
        {                                   // at most one (sequence, q-head) pair per wave: M Hq <= 128 < 8 NWG waves
            const int i = wg + NWG * wave;
            if (i < M * at.Hq) {
                const int b = da_div(i, at.inv_Hq);
                dec_attn_merge_one<HD, 1>(at, i - b * at.Hq, b);
            }
        }
        BRA_PSTAMP(2);
        // look-ahead requests go out LAST in a phase: a CU serves its waves' requests in arrival order, so a burst of weight
        // requests issued earlier delays the phase's own (latency-critical) operands — measured: merge 1.7 -> 5.5 us
        gs_drain();                         // the wave's stores of o have left before any look-ahead enters the CU's queue
        if (PF) {
            bare_barrier();                 // ... and so have every other wave's of this workgroup
            sched_fence();
            look_ahead();
        }
        BRA_PSTAMP(3);
        }
        BRA_PBAR();
        // ------------------------------------------------------------------ h = x + o Wo^T  (+ statistics of h)
        {
            if (BRA_PRUN()) {
            BRA_PIDS();
            BRA_PSTAMP(0);
            DecGemm2Args g = {at.o, NQ, nullptr, 0, nullptr, 0.f, Lr.Wo, NQ, a.x, H, (void*)a.h, H, a.ssh, a.nss, M, H, NQ, 1, nullptr, nullptr, 1.f / (float)NQ};
            if (!PF) pw_issue<NW_O, NL_O>(wo, Lr.Wo, wg, wave, lane_id());
            Frag<NL_O> xf;
            px_load<1, NW_O, NL_O>(xf, at.o, NQ, M, wave, lane_id());
            const u32x2 resv = pr_load<1>(a.x, H, M, H, wg, lane_id());
            pg_tile<1, 0, 0, NW_O, NL_O>(wo, xf, g, wg, 0, resv, sq0, red, rs_lds, wave, lane_id());
            BRA_PSTAMP(2);
            if (wave == 0) gs_drain();
            if (PF) {                       // look ahead (last, behind the epilogue wave's drained stores): first half of the down projection
                bare_barrier();
                sched_fence();
                pw_issue<8, NL_D, 0, NL_D / 2>(wd, Lr.Wd, wg, wave, lane_id());
                sched_fence();
            }
            BRA_PSTAMP(3);
            }
            BRA_PBAR();
        }
        // ------------------------------------------------------------------ act = silu(gate) * up of rmsnorm(h) Wgu^T  (ln2 folded)
        {
            if (BRA_PRUN()) {
            BRA_PIDS();
            BRA_PSTAMP(0);
            DecGemm2Args g = {a.h, H, a.ssh, a.nss, nullptr, a.eps, Lr.Wgu, H, nullptr, 0, (void*)a.act, F, nullptr, 0, M, 2 * F, H, 3, nullptr, nullptr, 1.f / (float)H};
            if (!PF) {
                pw_issue<8, NL_QKV>(wg0, Lr.Wgu, wg, wave, lane_id());
                pw_issue<8, NL_QKV>(wg1, Lr.Wgu, wg + NWG, wave, lane_id());
            }
            Frag<NL_QKV> xf;
            px_load<0, 8, NL_QKV>(xf, a.h, H, M, wave, lane_id());
            f32x4 sq[8];
            ps_load(sq, a.ssh, a.nss, M, wave, lane_id());
            if (PF) {                       // the rest of the first tile and the second tile, behind the phase's own operands
                sched_fence();
                pw_issue<8, NL_QKV, GU_A, NL_QKV>(wg0, Lr.Wgu, wg, wave, lane_id());
                pw_issue<8, NL_QKV>(wg1, Lr.Wgu, wg + NWG, wave, lane_id());
                sched_fence();
            }
            pg_tile<0, 2, 1, 8, NL_QKV>(wg0, xf, g, wg, 0, zero2, sq, red, rs_lds, wave, lane_id());
            pw_issue<8, NL_QKV>(wg0, Lr.Wgu, wg + 2 * NWG, wave, lane_id());
            pg_tile<0, 2, 1, 8, NL_QKV>(wg1, xf, g, wg + NWG, 1, zero2, sq, red, rs_lds, wave, lane_id());
            pg_tile<0, 2, 1, 8, NL_QKV>(wg0, xf, g, wg + 2 * NWG, 2, zero2, sq, red, rs_lds, wave, lane_id());
            BRA_PSTAMP(2);
            if (wave < 3) gs_drain();
            if (PF) {                       // look ahead (last): second half of the down projection
                bare_barrier();
                sched_fence();
                pw_issue<8, NL_D, NL_D / 2, NL_D>(wd, Lr.Wd, wg, wave, lane_id());
                sched_fence();
            }
            BRA_PSTAMP(3);
            }
            BRA_PBAR();
        }
        // ------------------------------------------------------------------ x = h + act Wd^T  (+ statistics of x)
        {
            if (BRA_PRUN()) {
            BRA_PIDS();
            BRA_PSTAMP(0);
            DecGemm2Args g = {a.act, F, nullptr, 0, nullptr, 0.f, Lr.Wd, F, a.h, H, (void*)a.x, H, a.ssx, a.nss, M, H, F, 1, nullptr, nullptr, 1.f / (float)F};
            if (!PF) pw_issue<8, NL_D>(wd, Lr.Wd, wg, wave, lane_id());
            Frag<NL_D> xf;
            px_load<1, 8, NL_D>(xf, a.act, F, M, wave, lane_id());
            const u32x2 resv = pr_load<1>(a.h, H, M, H, wg, lane_id());
            pg_tile<1, 0, 0, 8, NL_D>(wd, xf, g, wg, 0, resv, sq0, red, rs_lds, wave, lane_id());
            BRA_PSTAMP(2);
            if (wave == 0) gs_drain();
            if (PF) {                       // look ahead (last): the next layer's qkv tile
                bare_barrier();
                sched_fence();
                pw_issue<8, NL_QKV>(wq, Ls[l + 1 < a.L ? l + 1 : l].Wqkv, wg, wave, lane_id());
                sched_fence();
            }
            BRA_PSTAMP(3);
            }
            BRA_PBAR();
        }
    }
#undef BRA_PBAR
#undef BRA_PRUN
#undef BRA_PSTAMP
#undef BRA_PIDS
}
#endif  // !BRA_EMU

}  // namespace bra

using namespace bra;

extern "C" int bra_gridsync_bytes(void) { return (int)sizeof(GridSync); }

// iters iterations of {publish 128 B per workgroup, grid barrier, read all slots} on `nwg` workgroups of 512 threads (nwg must
// not exceed the number of CUs: every workgroup has to be resident).  sync = GridSync (zeroed here), buf = 2 * nwg * 32 words,
// errs = 4 words: [0] mismatching words, [1] barriers completed by workgroup 0, (sync->err[0] != 0: a barrier timed out).
extern "C" int bra_gridbar_probe(void* sync, void* buf, void* errs, const void* wts, long wts_bytes, int nwg, int iters, int mode,
                                 int wchunks, int timeout_us, void* stream) {
#ifdef BRA_EMU
    (void)sync; (void)buf; (void)errs; (void)wts; (void)wts_bytes; (void)nwg; (void)iters; (void)mode; (void)wchunks; (void)timeout_us; (void)stream;
    return BRA_ERR_UNSUPPORTED;       // workgroups of an emulated launch run one after another: nothing to synchronise
#else
    if (!sync || !buf || !errs || nwg <= 0 || iters <= 0 || mode < 0 || mode > 5 || wchunks < 0 || wchunks > 8) return BRA_ERR_ARG;
    if ((mode == 3 || mode == 5) && (!wts || wts_bytes < 16)) return BRA_ERR_ARG;
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return BRA_ERR_ARG;
    if (nwg > ncu) return BRA_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sync, 0, sizeof(GridSync), st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(errs, 0, 16, st);
    if (e != hipSuccess) return (int)e;
    const unsigned ticks = (unsigned)(timeout_us > 0 ? timeout_us : 20000) * 100u;           // s_memrealtime: 100 MHz
    BRA_LAUNCH((gridbar_probe_kernel<512>), dim3(nwg), dim3(512), 0, st, (GridSync*)sync, (unsigned*)buf, (unsigned*)errs,
               (const u32x4*)wts, (unsigned long)(wts_bytes / 16), iters, mode, wchunks, ticks);
    return BRA_LAUNCH_STATUS();
#endif
}

extern "C" int bra_persist_layer_desc_size(void) { return (int)sizeof(PLayer); }

#ifndef BRA_EMU
static std::atomic<unsigned long long*> g_persist_stamps{nullptr};
#endif
// diagnostics knob: device buffer of 6 L x 4 stamps filled by the next persistent launches (null: off)
extern "C" int bra_persist_set_stamps(void* p) {
#ifndef BRA_EMU
    g_persist_stamps = (unsigned long long*)p;
#else
    (void)p;
#endif
    return 0;
}

// All decoder layers of one shared-prefix decode step in ONE launch (see decode_persist_kernel).  `layers_dev`: HOST array of L
// records {Wqkv, Wo, Wgu, Wd (fragment-packed, norms folded), qn, kn, kp, vtp, kc, vct} (bra_persist_layer_desc_size() bytes
// each); every other argument as bra_qwen_decode_step_one.  x / ss_ws hold the embedded token rows and their RMSNorm statistics
// on entry and the last layer's output + statistics on exit (the caller runs the lm_head).  sync: bra_gridsync_bytes() bytes.
// prefetch: 0 none, 1 weights of the following phases, 2 also the K / V^T chunk of the attention items.  stop_after > 0 (tests):
// leave after that many phases.  BRA_ERR_UNSUPPORTED unless the shape is one of the instantiated models with <= 8 sequences and
// the device has one CU per workgroup.
extern "C" int bra_qwen_layers_persist(const void* layers_dev, int L, int R, int copies, int H, int Hq, int Hkv, int hd, int F, int P,
                                       long vt_pitch, int C, long cp, float eps, float scale, const float* cosT, const float* sinT,
                                       const int* pos, const float* rope_rows, const void* pmask, int t, const int* t_dev, void* x,
                                       void* qkv, void* o, void* h, void* act, float* ss_ws, int nss, float* part_o, float* part_ml,
                                       int nslot, void* sync, int prefetch, int stop_after, int timeout_us, void* stream) {
#ifdef BRA_EMU
    (void)layers_dev; (void)L; (void)R; (void)copies; (void)H; (void)Hq; (void)Hkv; (void)hd; (void)F; (void)P; (void)vt_pitch; (void)C; (void)cp;
    (void)eps; (void)scale; (void)cosT; (void)sinT; (void)pos; (void)rope_rows; (void)pmask; (void)t; (void)t_dev; (void)x; (void)qkv; (void)o;
    (void)h; (void)act; (void)ss_ws; (void)nss; (void)part_o; (void)part_ml; (void)nslot; (void)sync; (void)prefetch; (void)stop_after;
    (void)timeout_us; (void)stream;
    return BRA_ERR_UNSUPPORTED;       // a persistent grid needs concurrently resident workgroups
#else
    if (!layers_dev || L <= 0 || L > kPersistMaxLayers || R <= 0 || copies <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || P <= 0 || t < 0 || t >= C) return BRA_ERR_ARG;
    if (!cosT || !sinT || !pos || !x || !qkv || !o || !h || !act || !ss_ws || !part_o || !part_ml || !sync) return BRA_ERR_ARG;
    const int B = R * copies, G = Hq / Hkv;
    if (B > 8 || copies * G > 16) return BRA_ERR_UNSUPPORTED;
    const int Nq = Hq * hd, Nkv = Hkv * hd;
    const int npc = (P + 63) / 64, ncc = (t + 63) / 64;
    if (vt_pitch < (long)npc * 64 || vt_pitch % 8 || cp < ((C + 63) / 64) * 64 || cp % 8) return BRA_ERR_ARG;
    if (nslot < npc + (C + 63) / 64 + 1 || nslot > 256) return BRA_ERR_ARG;
    if (nss != 256) return BRA_ERR_UNSUPPORTED;                      // H / 8 statistics partials per row, folded 32 per lane
    const long kp_sr = (long)Hkv * P * hd, kp_sh = (long)P * hd, kp_ss = hd;
    const long vt_sr = (long)Hkv * hd * vt_pitch, vt_sh = (long)hd * vt_pitch, vt_sd = vt_pitch;
    const long lim = 1L << 30, f24 = 1L << 24;
    const long nitems = (long)npc * Hkv * R + (long)ncc * copies * Hkv * R + (long)R * copies * Hq;
    if (nitems >= (1 << 21) || vt_sd >= f24 || cp >= f24 || (long)(npc * 64) * kp_ss >= lim || (long)hd * vt_sd >= lim ||
        (long)hd * cp >= lim || (long)B * Hq * nslot * hd >= lim / 4)
        return BRA_ERR_UNSUPPORTED;
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return BRA_ERR_ARG;
    PersistArgs a;
    for (int i = 0; i < L; ++i) a.layers[i] = ((const PLayer*)layers_dev)[i];
    a.L = L; a.M = B;
    a.att = DecOneArgs{(const bf16_t*)qkv, (long)(Nq + 2 * Nkv), nullptr, nullptr, cosT, sinT, pos, rope_rows,
                       nullptr, kp_sr, kp_sh, kp_ss, nullptr, vt_sr, vt_sh, vt_sd, (const uint8_t*)pmask,
                       nullptr, nullptr, cp, part_o, part_ml, (bf16_t*)o, (long)Nq,
                       R, copies, Hq, Hkv, P, C, t, nslot, npc, ncc, eps, scale, t_dev,
                       1.f / (float)Hq, 1.f / (float)npc, 1.f / (float)Hkv, 1.f / (float)(ncc > 0 ? ncc : 1), 1.f / (float)copies};
    a.x = (bf16_t*)x; a.h = (bf16_t*)h; a.act = (bf16_t*)act; a.ssx = ss_ws; a.ssh = ss_ws + 8 * (long)nss; a.nss = nss; a.eps = eps;
    a.sync = (GridSync*)sync; a.timeout_ticks = (unsigned)(timeout_us > 0 ? timeout_us : 50000) * 100u;
    // stop_after (diagnostics): 0 = the whole step in one launch; k > 0 = phases [0, k); -1 = one launch per phase; -2 = one per layer
    // stop_after <= -1000: -(1000 + 8 p0 + p1): exactly the phases [p0, p1) in one launch
    const int nph = 6 * L;
    int first = 0;
    int win = stop_after == -1 ? 1 : (stop_after == -2 ? 6 : nph);
    int last = stop_after > 0 ? (stop_after < nph ? stop_after : nph) : nph;
    if (stop_after <= -1000) { const int c = -stop_after - 1000; first = c / 8; last = c % 8; win = nph; if (first >= last || last > nph) return BRA_ERR_ARG; }
    if (stop_after < 0 && prefetch > 0) return BRA_ERR_ARG;            // windows skip the requests a later phase relies on
    // stop_after <= -100 (diagnostics): -(100 + mask): one launch per phase, the ops whose mask bit is set (1 qkv, 2 attention, 4 o,
    // 8 gate/up, 16 down) by the LAUNCHED kernels (needs layers_host in `timeout_us`... no: see bra_qwen_layers_mixed)
    a.stamps = g_persist_stamps;
    hipStream_t st = (hipStream_t)stream;
#define BRA_PERSIST(H_, NQ_, NKV_, F_, HD_, G_)                                                                                   \
    if (H == H_ && Nq == NQ_ && Nkv == NKV_ && F == F_ && hd == HD_ && G == G_) {                                                 \
        constexpr int NWG = (NQ_ + 2 * NKV_) / 16;                                                                                \
        if (ncu < NWG) return BRA_ERR_UNSUPPORTED;                  /* every workgroup must be resident */                        \
        for (int lo = first; lo < last; lo += win) {                                                                                \
            a.ph_lo = lo; a.ph_hi = lo + win < last ? lo + win : last;                                                            \
            /* counters and generations restart with every launch; `err` is STICKY: a barrier that gave up in ANY token step */ \
            /* stays visible to the host's check after the rollout (the caller zeroes the whole buffer once per rollout), and */ \
            /* every later barrier returns at once instead of spinning to its own timeout */                                     \
            hipError_t e = hipMemsetAsync(sync, 0, offsetof(GridSync, err), st);                                                  \
            if (e != hipSuccess) return (int)e;                                                                                   \
            if (prefetch <= 0) BRA_LAUNCH((decode_persist_kernel<H_, NQ_, NKV_, F_, HD_, G_, 0>), dim3(NWG), dim3(512), 0, st, a); \
            else if (prefetch == 1) BRA_LAUNCH((decode_persist_kernel<H_, NQ_, NKV_, F_, HD_, G_, 1>), dim3(NWG), dim3(512), 0, st, a); \
            else BRA_LAUNCH((decode_persist_kernel<H_, NQ_, NKV_, F_, HD_, G_, 2>), dim3(NWG), dim3(512), 0, st, a);              \
        }                                                                                                                         \
        return BRA_LAUNCH_STATUS();                                                                                               \
    }
    BRA_PERSIST(2048, 2048, 1024, 6144, 128, 2)                     // Qwen3-1.7B (the reference's "Qwen3-1B", sh_reason.sh:46)
#undef BRA_PERSIST
    return BRA_ERR_UNSUPPORTED;
#endif
}

// n dependent probe kernels (see chain_probe_kernel): chained = 1 alternates stream_a / stream_b with device-side waits, 0 issues
// them all on stream_a.  done: n x 16 words (zeroed here on stream_a BEFORE the first launch; the caller orders stream_b behind it),
// buf: 2 x nwg x 32 words, errs: 4 words ([0] mismatching words, [3] kernel index of a timed-out wait).
extern "C" int bra_chain_probe(void* done, void* buf, void* errs, const void* wts, long wts_bytes, int nwg, int n, int chained, int wchunks,
                               int timeout_us, void* stream_a, void* stream_b) {
#ifdef BRA_EMU
    (void)done; (void)buf; (void)errs; (void)wts; (void)wts_bytes; (void)nwg; (void)n; (void)chained; (void)wchunks; (void)timeout_us; (void)stream_a; (void)stream_b;
    return BRA_ERR_UNSUPPORTED;
#else
    if (!done || !buf || !errs || !wts || wts_bytes < 16 || nwg <= 0 || n <= 0 || (wchunks != 4 && wchunks != 8 && wchunks != 12)) return BRA_ERR_ARG;
    hipStream_t sa = (hipStream_t)stream_a, sb = (hipStream_t)stream_b;
    const unsigned ticks = (unsigned)(timeout_us > 0 ? timeout_us : 20000) * 100u;
    for (int k = 0; k < n; ++k) {
        hipStream_t st = (chained && (k & 1)) ? sb : sa;
        if (wchunks == 4) BRA_LAUNCH((chain_probe_kernel<512, 4>), dim3(nwg), dim3(512), 0, st, (unsigned*)done, (unsigned*)buf, (unsigned*)errs, (const u32x4*)wts, (unsigned long)(wts_bytes / 16), k, chained, ticks);
        else if (wchunks == 8) BRA_LAUNCH((chain_probe_kernel<512, 8>), dim3(nwg), dim3(512), 0, st, (unsigned*)done, (unsigned*)buf, (unsigned*)errs, (const u32x4*)wts, (unsigned long)(wts_bytes / 16), k, chained, ticks);
        else BRA_LAUNCH((chain_probe_kernel<512, 12>), dim3(nwg), dim3(512), 0, st, (unsigned*)done, (unsigned*)buf, (unsigned*)errs, (const u32x4*)wts, (unsigned long)(wts_bytes / 16), k, chained, ticks);
    }
    return BRA_LAUNCH_STATUS();
#endif
}
