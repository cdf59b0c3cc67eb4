// bra_gridsync.h — in-launch synchronisation of a persistent grid (one workgroup per CU) on gfx950.
//
// MI355X has 8 XCDs with private, mutually non-coherent L2s and a per-CU L1 that other CUs' stores never refresh
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").  The protocol used here is the
// write-through one of the guide (cdna_hip_programming.md Guideline 16, recipe R1):
//   * every word that crosses workgroups is STORED with sc1 (write-through to the memory side, the line is dropped from the
//     producer's L2) and LOADED with sc1 (bypasses the consumer's L1) — xs_store8 / xs_load16 below; no cache-wide fences;
//   * every storing wave drains its stores (s_waitcnt vmcnt(0)) before the workgroup arrives at the barrier;
//   * arrival and release are agent-scope atomics on a two-level counter: 8 groups (blockIdx % 8 — the observed XCD of the
//     block, used for speed only, never for correctness), the last arriver of a group bumps the top counter, the last group
//     publishes the new generation to the 8 per-group generation words that the members poll (one lane per workgroup, relaxed
//     sc1 loads + s_sleep: polling waves starve the memory traffic of the producers, NOTES.md);
//   * every spin is bounded by the 100 MHz wall clock: a barrier that does not complete within `timeout_ticks` sets the error
//     word, and every workgroup that sees the error word leaves — a lost workgroup cannot hang the GPU.
// State (GridSync) is zeroed by the host before EVERY launch; epochs count barriers WITHIN a launch.
#pragma once
#include "bra_device.h"

namespace bra {

struct GridSync {                 // 64-byte lines: no two polled words share a line
    unsigned group_cnt[8][16];
    unsigned top_cnt[16];
    unsigned gen[8][16];
    unsigned err[16];             // [0] != 0: a barrier timed out (value = 1 + index of the barrier)
};

#ifndef BRA_EMU
typedef __attribute__((address_space(1))) unsigned gs_u32;

__device__ __forceinline__ unsigned gs_load(unsigned* p) {
    return __hip_atomic_load((gs_u32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gs_store(unsigned* p, unsigned v) {
    __hip_atomic_store((gs_u32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned gs_add(unsigned* p, unsigned v) {
    return __hip_atomic_fetch_add((gs_u32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// cross-workgroup payload accesses (sc1): a buffer resource over the whole address space of `base`
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xs_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 xs_load16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);          // aux 16 = sc1
}
__device__ __forceinline__ u32x2 xs_load8(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 16);
}
__device__ __forceinline__ float xs_load4f(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 16));
}
__device__ __forceinline__ void xs_store16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 16);
}
__device__ __forceinline__ void xs_store8(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, u32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, byte_off, 0, 16);
}
__device__ __forceinline__ void xs_store4f(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, byte_off, 0, 16);
}

// Barrier over all `nwg` workgroups of the launch.  EVERY thread of the workgroup calls it; waves that stored cross-workgroup
// data must have drained those stores themselves (gs_drain()) — the barrier does not wait for anybody's memory operations, so
// prefetches issued before it stay in flight across it.  `epoch` is a per-thread counter starting at 0.  `flag` is one LDS word.
// Returns false when the launch is being abandoned (timeout): the caller returns at once.
__device__ __forceinline__ void gs_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// scalar-path read of a polled word: s_load goes through the scalar data cache (invalidated first), not through the CU's vector
// memory queue — a poll is then not queued behind the weight requests the workgroup has in flight (EXPERIMENT: whether the
// XCD's L2 serves a fresh copy is what tools/gridbar_probe.py mode 4 / 5 checks)
__device__ __forceinline__ unsigned gs_load_scalar(const unsigned* p) {
    unsigned v;
    asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

template <int SPOLL = 0>
__device__ __forceinline__ bool grid_barrier(GridSync* gs, unsigned& epoch, int nwg, unsigned timeout_ticks, unsigned* flag) {
    ++epoch;
    __builtin_amdgcn_s_barrier();                      // every wave of the workgroup is past its (drained) stores
    if (threadIdx.x == 0) {
        const unsigned grp = blockIdx.x & 7u;
        const unsigned ngroups = nwg < 8 ? (unsigned)nwg : 8u;
        const unsigned members = ((unsigned)nwg - grp + 7u) >> 3;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        if (gs_add(&gs->group_cnt[grp][0], 1u) + 1u == members * epoch) {
            if (gs_add(&gs->top_cnt[0], 1u) + 1u == ngroups * epoch) {
#pragma unroll
                for (unsigned g = 0; g < 8; ++g) gs_store(&gs->gen[g][0], epoch);
            }
        }
        unsigned ok = 1u;
        while ((SPOLL ? gs_load_scalar(&gs->gen[grp][0]) : gs_load(&gs->gen[grp][0])) < epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)timeout_ticks || gs_load(&gs->err[0]) != 0u) {
                gs_store(&gs->err[0], epoch);
                ok = 0u;
                break;
            }
        }
        *flag = ok;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    return *flag != 0u;
}
#endif  // !BRA_EMU

// Access policy of code shared between the launched kernels and the persistent step: XS = 0 plain accesses (the operand was
// produced by an EARLIER launch), XS = 1 sc1 accesses (produced / consumed by other workgroups of THIS launch).  `base` must be
// wave-uniform (it becomes a buffer resource), `byte_off` < 2^31 is the lane's offset.
template <int XS> __device__ __forceinline__ u32x4 xld16(const void* base, unsigned byte_off) {
#ifndef BRA_EMU
    if constexpr (XS != 0) return xs_load16(xs_rsrc(base), byte_off);
#endif
    return ld16(reinterpret_cast<const char*>(base) + byte_off);
}
template <int XS> __device__ __forceinline__ u32x2 xld8(const void* base, unsigned byte_off) {
#ifndef BRA_EMU
    if constexpr (XS != 0) return xs_load8(xs_rsrc(base), byte_off);
#endif
    return ld8(reinterpret_cast<const char*>(base) + byte_off);
}
template <int XS> __device__ __forceinline__ float xld4f(const void* base, unsigned byte_off) {
#ifndef BRA_EMU
    if constexpr (XS != 0) return xs_load4f(xs_rsrc(base), byte_off);
#endif
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <int XS> __device__ __forceinline__ void xst16(void* base, unsigned byte_off, const u32x4& v) {
#ifndef BRA_EMU
    if constexpr (XS != 0) { xs_store16(xs_rsrc(base), byte_off, v); return; }
#endif
    st16(reinterpret_cast<char*>(base) + byte_off, v);
}
template <int XS> __device__ __forceinline__ void xst8(void* base, unsigned byte_off, const u32x2& v) {
#ifndef BRA_EMU
    if constexpr (XS != 0) { xs_store8(xs_rsrc(base), byte_off, v); return; }
#endif
    st8(reinterpret_cast<char*>(base) + byte_off, v);
}
template <int XS> __device__ __forceinline__ void xst4f(void* base, unsigned byte_off, float v) {
#ifndef BRA_EMU
    if constexpr (XS != 0) { xs_store4f(xs_rsrc(base), byte_off, v); return; }
#endif
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

}  // namespace bra
