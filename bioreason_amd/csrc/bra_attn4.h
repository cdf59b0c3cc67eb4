// bra_attn4.h — device helpers shared by the software-pipelined attention kernels (k_attn4.hip forward, k_attn4b.hip backward):
// single-instruction forms the compiler does not pick by itself, accumulators pinned to the accumulator half of the register file,
// LDS-DMA through buffer descriptors, scheduling pins.
#pragma once
#include "bra_device.h"
#include "bra_attn.h"

namespace bra {

#ifdef BRA_A4_NODMA       // (timing probe: the hot loop stages nothing — garbage results)
constexpr bool kNoDma = true;
#else
constexpr bool kNoDma = false;
#endif
constexpr float kMasked = -3.0e38f;     // raw score of a masked key (finite: times `sc` it stays finite, exp2 gives 0)

#ifdef BRA_EMU
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float max2f(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float xhalf_max(float v) { return fmaxf(v, wave_shfl_xor(v, 32)); }
__device__ __forceinline__ float xhalf_sum(float v) { return v + wave_shfl_xor(v, 32); }
// lanes 0..31 receive {a (own), a of lane + 32}; lanes 32..63 receive {b of lane - 32, b (own)}: T21's widened row store
__device__ __forceinline__ void xhalf_pair(uint32_t& a, uint32_t& b) {
    const uint32_t ao = wave_shfl_xor_u32(a, 32), bo = wave_shfl_xor_u32(b, 32);
    if (bra_emu::lane_id() < 32) b = ao; else a = bo;
}
#else
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (fmaxf on a value the compiler cannot prove to be a quiet number is preceded by a canonicalising v_max_f32 x, x)
__device__ __forceinline__ float max2f(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float xhalf_max(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return max2f(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
}
__device__ __forceinline__ void xhalf_pair(uint32_t& a, uint32_t& b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);       // vdst = a: its upper half swaps with b's lower half
    a = r[0]; b = r[1];
}
#endif

// O^T += V^T . P^T with the accumulator pinned to the accumulator half of the register file (this file is built with
// -amdgpu-mfma-vgpr-form, so every builtin MFMA — the score tiles the softmax reads — has an architectural destination; left to the
// compiler the O registers either take that half too or are copied around every branch).  An asm MFMA is opaque to the hazard
// recogniser: a VALU read of O (rescale, epilogue) must be preceded by mfma_drain(); MFMA -> MFMA on the same accumulator needs nothing.
#ifdef BRA_EMU
__device__ __forceinline__ void mfma_o(f32x16& o, const u32x4& a, const u32x4& b) { o = mfma_32x32x16(a, b, o); }
__device__ __forceinline__ void mfma_drain() {}
#else
__device__ __forceinline__ void mfma_o(f32x16& o, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory"); }
#endif
// a pointer the compiler can keep in scalar registers (everything derived from blockIdx through divisions is "divergent" to it)
#ifdef BRA_EMU
__device__ __forceinline__ const char* uniform_ptr(const void* p) { return (const char*)p; }
#else
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}
#endif
// 16 bytes per lane from (descriptor base + lane offset + scalar offset) to (wave-uniform LDS base + 16 lane): `buffer_load_dwordx4 ...
// offen lds`; a lane whose bytes lie beyond the descriptor's size receives zeros
#ifdef BRA_EMU
struct BufDesc { const char* base; unsigned bytes; };
__device__ __forceinline__ BufDesc make_bufdesc(const char* p, unsigned bytes) { BufDesc d = {p, bytes}; return d; }
__device__ __forceinline__ void dma16(const BufDesc& d, unsigned voff, unsigned soff, char* lds_wave_base) {
    const unsigned long long off = (unsigned long long)voff + soff;
    char* dst = lds_wave_base + bra_emu::lane_id() * 16;
    if (off + 16 <= d.bytes) memcpy(dst, d.base + off, 16); else memset(dst, 0, 16);
}
#else
typedef __amdgpu_buffer_rsrc_t BufDesc;
__device__ __forceinline__ BufDesc make_bufdesc(const char* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void dma16(const BufDesc& d, unsigned voff, unsigned soff, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
#endif
#ifdef BRA_EMU
__device__ __forceinline__ void to_agpr(u32x4&) {}
__device__ __forceinline__ void pin_u32_f32(uint32_t&, float&, float&) {}
__device__ __forceinline__ void pin_f32(float&) {}
__device__ __forceinline__ void pin_u32(uint32_t&) {}
#else
__device__ __forceinline__ void to_agpr(u32x4& v) { asm volatile("" : "+a"(v)); }
__device__ __forceinline__ void pin_u32_f32(uint32_t& a, float& b, float& c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }
__device__ __forceinline__ void pin_f32(float& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin_u32(uint32_t& a) { asm volatile("" : "+v"(a)); }
#endif

}  // namespace bra
