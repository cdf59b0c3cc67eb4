// k_persist.hip — persistent-grid building blocks of the rollout's token loop (HF `_sample` loop body,
// TF:generation/utils.py:2876-2925): the in-launch grid barrier / hand-off protocol of bra_gridsync.h and a probe that
// measures and word-checks it on the device (tools/gridbar_probe.py).
#include "bra_gridsync.h"
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
                                                           const u32x4* wts, unsigned long wts_chunks, int iters, int mode,
                                                           int wchunks, unsigned timeout_ticks) {
    __shared__ unsigned flag;
    const int nwg = (int)gridDim.x, wg = (int)blockIdx.x, tid = (int)threadIdx.x;
    unsigned epoch = 0, bad = 0, sink = 0;
    const __amdgpu_buffer_rsrc_t rs = xs_rsrc(buf);
    const unsigned nwords = (unsigned)nwg * 32u;
    for (int it = 0; it < iters; ++it) {
        unsigned* slot = buf + (size_t)(it & 1) * nwords;
        u32x4 w[8];
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
        if (!grid_barrier(gs, epoch, nwg, timeout_ticks, &flag)) break;
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
    if (!sync || !buf || !errs || nwg <= 0 || iters <= 0 || mode < 0 || mode > 3 || wchunks < 0 || wchunks > 8) return BRA_ERR_ARG;
    if (mode == 3 && (!wts || wts_bytes < 16)) return BRA_ERR_ARG;
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
