// bra_dropout.h — counter-based dropout masks shared by the LoRA kernels (k_lora.hip, k_wgrad.hip).
// keep(seed, idx) for element idx = m * K + k of a [M, K] activation: one 32-bit hash per PAIR of consecutive elements;
// element idx takes the 16-bit half (idx & 1) of the hash, whose upper 15 bits are compared with round(p * 32768)
// (p = 0.05 -> 1638 / 32768 = 0.04999).  Stateless, so forward, the input gradient and the weight gradient regenerate
// identical masks from (seed, index) alone.
// Cost is what shapes this file: the kernels that use it stream 80-240 MB per call and would be HBM-bound, but one mask per
// element and per target module is 40-160 M mask evaluations on a machine whose wave64 VALU retires one instruction per 4
// cycles — so the hash uses only full-rate instructions (v_mad_u32_u24 instead of the quarter-rate v_mul_lo_u32: two
// 24-bit multiply-add / xor-shift rounds, checked for keep rate, serial and cross-seed correlation and bucket uniformity
// against the former two-round 32-bit multiply hash), and both halves of a hash are turned into an AND mask with three
// packed 16-bit instructions (shift, subtract, arithmetic shift) instead of two compare / select chains.
#pragma once
#include "bra_device.h"

namespace bra {

struct DropCfg {
    uint32_t thr16;        // drop when the 15-bit field is < thr16 (name kept: the field is bits 1..15 of a 16-bit half)
    float inv_keep;        // 1 / (1 - p)
    uint32_t seed[4];      // one stream per 32-column rank block (= per target module of a fused projection)
};

__host__ __device__ inline uint32_t drop_threshold(float p) {
    float t = p * 32768.f + 0.5f;
    return t <= 0.f ? 0u : (t >= 32767.f ? 32767u : (uint32_t)t);
}

#ifdef BRA_EMU
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
#else
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
#endif

__device__ __forceinline__ uint32_t drop_hash(uint32_t seed, uint32_t pair) {
    const uint32_t x = pair ^ seed;
    uint32_t h = mul24(x, 0xB5297Bu) + (x >> 13);
    h ^= h >> 14;
    h = mul24(h, 0x68E31Du) + (h >> 11);
    h ^= h >> 15;
    return h;
}
__device__ __forceinline__ bool drop_field(uint32_t h, uint32_t idx, uint32_t thr16) {
    return (((idx & 1u) ? (h >> 17) : ((h & 0xffffu) >> 1))) >= thr16;
}
__device__ __forceinline__ bool drop_keep1(uint32_t seed, uint32_t idx, uint32_t thr16) {
    return drop_field(drop_hash(seed, idx >> 1), idx, thr16);
}
// 32-bit AND mask of the element pair a hash covers: 0xffff in a half that is kept, 0 in one that is dropped
typedef uint16_t drop_u16x2 __attribute__((ext_vector_type(2)));
typedef int16_t drop_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t drop_pair_mask(uint32_t h, uint32_t thr16) {
    const drop_u16x2 f = __builtin_bit_cast(drop_u16x2, h) >> (uint16_t)1;                       // 15-bit fields
    const drop_i16x2 t = {(int16_t)thr16, (int16_t)thr16};
    const drop_i16x2 d = __builtin_bit_cast(drop_i16x2, f) - t;                                   // negative <=> dropped
    return ~__builtin_bit_cast(uint32_t, d >> (int16_t)15);
}
// 8 consecutive bf16 elements starting at element index e0 (a multiple of 8): scaled by 1/(1-p) in fp32 and rounded once to
// bf16 — as torch.nn.functional.dropout does on a bf16 tensor — then masked (the scaled values do not depend on the mask
// stream: with several rank blocks on the same input the compiler computes them once)
__device__ __forceinline__ u32x4 drop_apply8(const u32x4& v, uint32_t seed, uint32_t e0, uint32_t thr16, float inv_keep) {
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t w = v[q];
        const uint32_t scaled = pack_bf2(bf_lo(w) * inv_keep, bf_hi(w) * inv_keep);
        o[q] = scaled & drop_pair_mask(drop_hash(seed, (e0 >> 1) + (uint32_t)q), thr16);
    }
    return o;
}

}  // namespace bra
