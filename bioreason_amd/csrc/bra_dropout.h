// bra_dropout.h — counter-based dropout masks shared by the LoRA kernels (k_lora.hip, k_wgrad.hip).
// keep(seed, idx) for element idx = m * K + k of a [M, K] activation: one 32-bit hash per PAIR of consecutive elements,
// 16 bits each compared with round(p * 65536) (p = 0.05 -> 3277 / 65536 = 0.050003).  Stateless, so forward, the input
// gradient and the weight gradient regenerate identical masks from (seed, index) alone.
#pragma once
#include "bra_device.h"

namespace bra {

struct DropCfg {
    uint32_t thr16;        // drop when the 16-bit field is < thr16
    float inv_keep;        // 1 / (1 - p)
    uint32_t seed[4];      // one stream per 32-column rank block (= per target module of a fused projection)
};

__host__ __device__ inline uint32_t drop_threshold(float p) {
    float t = p * 65536.f + 0.5f;
    return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)t);
}

// two multiply / xor-shift rounds: the second xor-shift folds the well-mixed high half into the low 16-bit field
__device__ __forceinline__ uint32_t drop_hash(uint32_t seed, uint32_t pair) {
    uint32_t h = (pair ^ seed) * 0x9E3779B1u;
    h ^= h >> 16; h *= 0x85ebca6bu;
    h ^= h >> 13;
    return h;
}
__device__ __forceinline__ bool drop_field(uint32_t h, uint32_t idx, uint32_t thr16) {
    return ((idx & 1u) ? (h >> 16) : (h & 0xffffu)) >= thr16;
}
__device__ __forceinline__ bool drop_keep1(uint32_t seed, uint32_t idx, uint32_t thr16) {
    const uint32_t h = drop_hash(seed, idx >> 1);
    return ((idx & 1u) ? (h >> 16) : (h & 0xffffu)) >= thr16;
}
// 8 consecutive bf16 elements starting at element index e0 (a multiple of 8): masked and scaled by 1/(1-p) in fp32,
// rounded once to bf16 — as torch.nn.functional.dropout does on a bf16 tensor
__device__ __forceinline__ u32x4 drop_apply8(const u32x4& v, uint32_t seed, uint32_t e0, uint32_t thr16, float inv_keep) {
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t h = drop_hash(seed, (e0 >> 1) + (uint32_t)q);
        const uint32_t w = v[q];
        const float lo = (h & 0xffffu) >= thr16 ? bf_lo(w) * inv_keep : 0.f;
        const float hi = (h >> 16) >= thr16 ? bf_hi(w) * inv_keep : 0.f;
        o[q] = pack_bf2(lo, hi);
    }
    return o;
}

}  // namespace bra
