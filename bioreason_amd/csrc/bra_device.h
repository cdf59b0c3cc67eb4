// bra_device.h — device-side helpers shared by every kernel in this library.
// gfx950 (CDNA4) only: wave64, MFMA bf16 16x16x32 / 32x32x16, 160 KiB LDS.
// (With -DBRA_EMU the same sources are compiled for the host test executor in
// tests/emu/; that build is test infrastructure and never ships.)
#pragma once
// Diagnostics (timing probes, tile-variant knobs, the persistent decode step) exist only in the -DBRA_DEBUG build
// (libbioreason_hip_debug.so, include/bioreason_hip_debug.h); the product library carries no process-wide mutable knob and no
// probe word in any kernel argument record.
#ifdef BRA_DEBUG
#define BRA_DBG_FIELD(decl) decl
#define BRA_DBG_INIT(x) x,
#else
#define BRA_DBG_FIELD(decl)
#define BRA_DBG_INIT(x)
#endif

#include <stdint.h>
#include <stddef.h>

#ifdef BRA_EMU
#include "bra_emu.h"
#else
#include <hip/hip_runtime.h>
#include <atomic>
#endif

namespace bra {

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;

// ---- bf16 <-> f32 -----------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
#ifdef BRA_EMU
__device__ __forceinline__ bf16_t f2bf(float f) {
    // round-to-nearest-even, NaN kept quiet (same rounding torch uses)
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
#else
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even): the compiler emits it for a plain cast
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    hw_bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
#endif
__device__ __forceinline__ float bf_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// round an f32 to the nearest bf16 and back (the reference rounds module outputs to bf16)
__device__ __forceinline__ float round_bf(float f) { return bf2f(f2bf(f)); }

// 8 bf16 <-> 8 floats
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
    v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
    v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
    return v;
}

// ---- wave64 primitives ------------------------------------------------
#ifdef BRA_EMU
__device__ __forceinline__ int lane_id() { return bra_emu::lane_id(); }
__device__ __forceinline__ uint32_t wave_shfl_u32(uint32_t v, int src) {
    const uint32_t* b = bra_emu::wave_exchange(&v, 1);
    return b[(src & 63) * 16];
}
__device__ __forceinline__ uint32_t wave_shfl_xor_u32(uint32_t v, int m) {
    return wave_shfl_u32(v, bra_emu::lane_id() ^ m);
}
__device__ __forceinline__ uint64_t wave_ballot(bool p) {
    uint32_t v = p ? 1u : 0u;
    const uint32_t* b = bra_emu::wave_exchange(&v, 1);
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) if (b[l * 16]) m |= (1ull << l);
    return m;
}
#else
#ifdef BRA_LANE_OPAQUE
// (k_persist.hip) every call yields a value the optimiser cannot relate to the others: inside a loop over decoder layers hipcc
// otherwise hoists ALL lane-derived address arithmetic of every phase to kernel entry and spills it around the loop (NOTES.md)
__device__ __forceinline__ int lane_id() { int l = (int)(threadIdx.x & 63); asm volatile("" : "+v"(l)); return l; }
#else
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
#endif
__device__ __forceinline__ uint32_t wave_shfl_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
__device__ __forceinline__ uint32_t wave_shfl_xor_u32(uint32_t v, int m) { return (uint32_t)__shfl_xor((int)v, m, 64); }
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __ballot(p); }
#endif
// exchange with lane ^ 1 (DPP quad permute: a VALU move, no LDS crossbar round trip)
#ifdef BRA_EMU
__device__ __forceinline__ uint32_t lane_swap1(uint32_t v) { return wave_shfl_xor_u32(v, 1); }
#else
__device__ __forceinline__ uint32_t lane_swap1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
}
#endif
// rotate within each row of 16 lanes (DPP row_ror: a VALU move); four of them (8, 4, 2, 1) all-reduce an idempotent operator
// over a row without touching the LDS crossbar
#ifdef BRA_EMU
template <int N> __device__ __forceinline__ uint32_t row_ror_u32(uint32_t v) {
    const uint32_t* b = bra_emu::wave_exchange(&v, 1);
    const int l = bra_emu::lane_id();
    return b[((l & ~15) | ((l + N) & 15)) * 16];
}
#else
template <int N> __device__ __forceinline__ uint32_t row_ror_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xF, 0xF, false);
}
#endif
// maximum of a 64-bit key over the 16 lanes of each row / over the wave (every lane receives it)
__device__ __forceinline__ uint64_t row_max_u64(uint64_t k) {
#define BRA_ROR_STEP(N)                                                                                        \
    { const uint64_t o = ((uint64_t)row_ror_u32<N>((uint32_t)(k >> 32)) << 32) | row_ror_u32<N>((uint32_t)k);  \
      k = o > k ? o : k; }
    BRA_ROR_STEP(8) BRA_ROR_STEP(4) BRA_ROR_STEP(2) BRA_ROR_STEP(1)
#undef BRA_ROR_STEP
    return k;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t k) {
    k = row_max_u64(k);
#pragma unroll
    for (int m = 16; m <= 32; m <<= 1) {
        const uint64_t o = ((uint64_t)wave_shfl_xor_u32((uint32_t)(k >> 32), m) << 32) | wave_shfl_xor_u32((uint32_t)k, m);
        k = o > k ? o : k;
    }
    return k;
}
__device__ __forceinline__ float wave_shfl_xor(float v, int m) {
    return __builtin_bit_cast(float, wave_shfl_xor_u32(__builtin_bit_cast(uint32_t, v), m));
}
__device__ __forceinline__ float wave_shfl(float v, int src) {
    return __builtin_bit_cast(float, wave_shfl_u32(__builtin_bit_cast(uint32_t, v), src));
}
__device__ __forceinline__ int wave_shfl_i(int v, int src) { return (int)wave_shfl_u32((uint32_t)v, src); }
__device__ __forceinline__ int wave_shfl_xor_i(int v, int m) { return (int)wave_shfl_xor_u32((uint32_t)v, m); }

// butterfly reductions over `width` consecutive lanes (width = 64, 32, 16, ...)
template <int WIDTH = 64>
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = WIDTH / 2; m >= 1; m >>= 1) v += wave_shfl_xor(v, m);
    return v;
}
template <int WIDTH = 64>
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = WIDTH / 2; m >= 1; m >>= 1) v = fmaxf(v, wave_shfl_xor(v, m));
    return v;
}

// block-wide sum for blocks of NWAVES waves (every thread must call)
template <int NWAVES>
__device__ __forceinline__ float block_sum(float v, float* red /* [NWAVES] shared */) {
    v = wave_sum<64>(v);
    if (NWAVES == 1) return v;
    int w = (int)(threadIdx.x >> 6);
    __syncthreads();
    if (lane_id() == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NWAVES; ++i) t += red[i];
    return t;
}
template <int NWAVES>
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max<64>(v);
    if (NWAVES == 1) return v;
    int w = (int)(threadIdx.x >> 6);
    __syncthreads();
    if (lane_id() == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NWAVES; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---- MFMA -------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16:  D[16x16] = A[16x32] * B[32x16] + C
//   A operand: lane l holds A[l&15][8*(l>>4) + j], j = 0..7   (4 VGPRs)
//   B operand: lane l holds B[8*(l>>4) + j][l&15]
//   C/D      : lane l, reg r holds D[4*(l>>4) + r][l&15]
// v_mfma_f32_32x32x16_bf16:  D[32x32] = A[32x16] * B[16x32] + C
//   A: lane l holds A[l&31][8*(l>>5) + j];  B: lane l holds B[8*(l>>5) + j][l&31]
//   C/D: lane l, reg r holds D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
#ifdef BRA_EMU
__device__ inline f32x4 mfma_16x16x32(const u32x4& a, const u32x4& b, f32x4 c) {
    uint32_t mine[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const uint32_t* x = bra_emu::wave_exchange(mine, 8);
    int l = bra_emu::lane_id();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            const uint32_t* la = x + (size_t)(row + 16 * (k >> 3)) * 16;
            const uint32_t* lb = x + (size_t)(col + 16 * (k >> 3)) * 16 + 4;
            int j = k & 7;
            uint32_t wa = la[j >> 1], wb = lb[j >> 1];
            float fa = (j & 1) ? bf_hi(wa) : bf_lo(wa);
            float fb = (j & 1) ? bf_hi(wb) : bf_lo(wb);
            acc += fa * fb;
        }
        c[r] = acc;
    }
    return c;
}
__device__ inline f32x16 mfma_32x32x16(const u32x4& a, const u32x4& b, f32x16 c) {
    uint32_t mine[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const uint32_t* x = bra_emu::wave_exchange(mine, 8);
    int l = bra_emu::lane_id();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            const uint32_t* la = x + (size_t)(row + 32 * (k >> 3)) * 16;
            const uint32_t* lb = x + (size_t)(col + 32 * (k >> 3)) * 16 + 4;
            int j = k & 7;
            uint32_t wa = la[j >> 1], wb = lb[j >> 1];
            float fa = (j & 1) ? bf_hi(wa) : bf_lo(wa);
            float fb = (j & 1) ? bf_hi(wb) : bf_lo(wb);
            acc += fa * fb;
        }
        c[r] = acc;
    }
    return c;
}
#else
__device__ __forceinline__ f32x4 mfma_16x16x32(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b),
                                                   c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x16(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b),
                                                   c, 0, 0, 0);
}
#endif

// ---- misc ---------------------------------------------------------------
#ifdef BRA_EMU
#define BRA_DYN_SMEM(name) char* name = bra_emu::dyn_smem()
__device__ __forceinline__ void sched_fence() {}
__device__ __forceinline__ void setprio(int) {}
#else
#define BRA_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#define setprio(p) __builtin_amdgcn_s_setprio(p)
#endif

// ---- asynchronous global -> LDS copy (LDS-DMA, `global_load_lds_dwordx4`) -----------------------------
// One wave-instruction moves 64 x 16 B: lane l's 16 bytes from its OWN global address to
// (wave-uniform LDS base) + 16*l.  It is a VMEM operation: completion is observed with s_waitcnt vmcnt(N)
// followed by a barrier, never by __syncthreads() alone.
#ifdef BRA_EMU
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    memcpy(lds_wave_base + bra_emu::lane_id() * 16, gsrc, 16);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {}
__device__ __forceinline__ void raw_barrier() { bra_emu::block_sync(); }
__device__ __forceinline__ void bare_barrier() { bra_emu::block_sync(); }
__device__ __forceinline__ void wait_lds() {}
// LDS words written by one lane of a wave and read by another lane of the SAME wave: the device needs the writes retired (lockstep
// does the rest), the emulator's lanes are fibers and need a rendezvous
__device__ __forceinline__ void wave_lds_sync() { bra_emu::wave_sync(); }
__device__ __forceinline__ int uniform_i(int v) { return v; }
#else
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// barrier that does NOT drain the VMEM queue (a __syncthreads() would wait for every LDS-DMA in flight)
__device__ __forceinline__ void raw_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
// s_barrier alone (LDS reads issued before it may still be in flight: wait_lds() after it, before their first use)
__device__ __forceinline__ void bare_barrier() { __builtin_amdgcn_s_barrier(); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }

// ---- fp8 (OCP e4m3fn: 1-4-3, bias 7, max 448, no infinities) — the rollout's optional weight format (BASELINE config 5) -------
// Every e4m3 value is exactly representable in bf16, so the decode is exact; the encode (bra_dec_pack_weights_fp8, offline) is
// round-to-nearest-even with saturation, in integer arithmetic so that the emulator and the device produce the same bytes.
__host__ __device__ inline float e4m3_to_f32(unsigned b) {
    const unsigned e = (b >> 3) & 15u, m = b & 7u;
    float v;
    if (e == 0) v = (float)m * (1.0f / 512.0f);                           // subnormal: m / 8 * 2^-6
    else { unsigned bits = ((e + 120u) << 23) | (m << 20); v = __builtin_bit_cast(float, bits); }      // 2^(e-7) * (1 + m/8)
    if (e == 15 && m == 7) v = __builtin_nanf("");
    return (b & 0x80u) ? -v : v;
}
__host__ __device__ inline unsigned f32_to_e4m3(float f) {
    const unsigned u = __builtin_bit_cast(unsigned, f);
    const unsigned sign = (u >> 24) & 0x80u;
    const unsigned a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return sign | 0x7fu;                             // NaN
    if (a >= 0x43e80000u) return sign | 0x7eu;                            // >= 464 = halfway to the next step above 448: saturate
    if (a < 0x3c800000u) {                                                // < 2^-6: subnormal grid, step 2^-9
        const float q = __builtin_rintf(__builtin_bit_cast(float, a) * 512.0f);    // exact product, ties to even
        return sign | (unsigned)q;                                        // q == 8 is 2^-6 = (e 1, m 0) = code 8
    }
    unsigned e = (a >> 23) - 120u;                                        // biased by 7
    unsigned m = (a >> 20) & 7u;
    const unsigned rem = a & 0xfffffu;
    if (rem > 0x80000u || (rem == 0x80000u && (m & 1u))) { if (++m == 8u) { m = 0; ++e; } }
    if (e > 15u || (e == 15u && m == 7u)) return sign | 0x7eu;
    return sign | (e << 3) | m;
}
// 8 fp8 (two dwords) -> 8 bf16 (the A fragment of one 16x16x32 MFMA step)
#ifdef BRA_EMU
__device__ inline u32x4 f8x8_to_bf16x8(unsigned lo, unsigned hi) {
    unsigned r[4];
    for (int i = 0; i < 4; ++i) {
        const unsigned w = i < 2 ? lo : hi;
        const unsigned b0 = (w >> ((i & 1) * 16)) & 0xffu, b1 = (w >> ((i & 1) * 16 + 8)) & 0xffu;
        r[i] = (__builtin_bit_cast(unsigned, e4m3_to_f32(b0)) >> 16) | (__builtin_bit_cast(unsigned, e4m3_to_f32(b1)) & 0xffff0000u);
    }
    u32x4 o = {r[0], r[1], r[2], r[3]};
    return o;
}
#else
__device__ __forceinline__ u32x4 f8x8_to_bf16x8(unsigned lo, unsigned hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    // v_cvt_scalef32_pk_bf16_fp8 with scale 1: two e4m3 -> two bf16 per instruction (exact)
    u32x4 o;
    o.x = __builtin_bit_cast(unsigned, (bf16x2_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false));
    o.y = __builtin_bit_cast(unsigned, (bf16x2_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true));
    o.z = __builtin_bit_cast(unsigned, (bf16x2_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false));
    o.w = __builtin_bit_cast(unsigned, (bf16x2_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true));
    return o;
}
#endif

// ---- fp8 (OCP e4m3) MFMA, K = 128 per instruction: v_mfma_scale_f32_16x16x128_f8f6f4 — the matrix pipe's double-rate form on
// gfx950 (MI355X_MICROARCH.md: MX-scaled K = 128 >= 4.66 PF/s; the non-scaled 16x16x32 fp8 form runs at the bf16 rate).  Operand layout
// found by experiment (tools/ubench/mfma_fp8_layout.hip, profiles/r6_n_mfma_fp8_layout.txt): lane l of operand a holds 32 consecutive
// bytes of row l & 15, k-group l >> 4 (the same for b with the column), a position of a meets the position of b with the same k-group
// and byte; C/D as every 16 x 16 shape (row 4 (l >> 4) + r of a's rows, column l & 15 of b's).  Block scales (E8M0, one byte per lane
// and operand) are passed as 2^0: the per-row / per-token fp32 scales of the GEMM are applied to the accumulators in its epilogue.
#ifdef BRA_EMU
__device__ inline f32x4 mfma_fp8_16x16x128(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, f32x4 c) {
    uint32_t mine[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const uint32_t* x = bra_emu::wave_exchange(mine, 16);
    const int l = bra_emu::lane_id();
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g) {
            const uint32_t* la = x + (size_t)(row + 16 * g) * 16;
            const uint32_t* lb = x + (size_t)(col + 16 * g) * 16 + 8;
            for (int j = 0; j < 32; ++j)
                acc += e4m3_to_f32((la[j >> 2] >> (8 * (j & 3))) & 0xffu) * e4m3_to_f32((lb[j >> 2] >> (8 * (j & 3))) & 0xffu);
        }
        c[r] = acc;
    }
    return c;
}
#else
__device__ __forceinline__ f32x4 mfma_fp8_16x16x128(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, f32x4 c) {
    typedef int i32x8_hw __attribute__((ext_vector_type(8)));
    const i32x8_hw a = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
    const i32x8_hw b = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}
#endif


// returns x but hides its origin from the compiler: address arithmetic built on it cannot be hoisted out of a loop
// (hipcc otherwise computes every lane-constant address at kernel entry and spills it around the tile loop)
#ifdef BRA_EMU
__device__ __forceinline__ int opaque_i(int x) { return x; }
#else
__device__ __forceinline__ int opaque_i(int x) { asm volatile("" : "+v"(x)); return x; }
#endif
// 2^x as the bare hardware instruction (v_exp_f32): no denormal rescue; softmax arguments are <= 0 and anything below
// 2^-126 is zero for a probability that is about to be rounded to bf16
#ifdef BRA_EMU
__device__ __forceinline__ float fast_exp2(float x) { return exp2f(x); }
#else
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#endif
// makes `v` opaque to the compiler at this point: nothing computed from it can be scheduled (or hoisted) above
#ifdef BRA_EMU
__device__ __forceinline__ void reg_fence(f32x4&) {}
#else
__device__ __forceinline__ void reg_fence(f32x4& v) { asm volatile("" : "+v"(v)); }
#endif
// streaming (read-once) 16-byte load: weights of the decode projections must not displace the activations in L2
#ifdef BRA_EMU
__device__ __forceinline__ u32x4 ld16_nt(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
#else
__device__ __forceinline__ u32x4 ld16_nt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
#endif
__device__ __forceinline__ void st16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ u32x2 ld8(const void* p) { return *reinterpret_cast<const u32x2*>(p); }
__device__ __forceinline__ void st8(void* p, const u32x2& v) { *reinterpret_cast<u32x2*>(p) = v; }

// XCD-aware bijective remap of a linear workgroup id (8 XCDs; block b runs on
// XCD b % 8): consecutive remapped ids share an XCD and therefore an L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned nx = 8;
    unsigned xcd = bid % nx, q = nwg / nx, r = nwg % nx;
    unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / nx;
}

}  // namespace bra

// ---- host side: launch + error plumbing ----------------------------------
#ifdef BRA_EMU
typedef void* bra_stream_t;
#define BRA_LAUNCH(kern, grid, block, smem, stream, ...) \
    bra_emu::launch((grid), (block), (smem), [=]() { kern(__VA_ARGS__); })
#define BRA_LAUNCH_STATUS() 0
#define BRA_ALLOW_SMEM(kern, bytes) ((void)0)
#else
typedef hipStream_t bra_stream_t;
#define BRA_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), (smem), (hipStream_t)(stream), __VA_ARGS__)
#define BRA_LAUNCH_STATUS() ((int)hipGetLastError())
// dynamic LDS above 64 KiB must be opted into once per kernel AND device (gfx950 has 160 KiB per CU): the attribute lives in
// the device's copy of the code object, so a process that drives several GPUs sets it on each; the flags are atomics because
// torch may call the library from more than one thread (autograd worker, RCCL watchdog)
#define BRA_ALLOW_SMEM(kern, bytes)                                                                       \
    do {                                                                                                  \
        static std::atomic<unsigned> done_{0u};                     /* bit d = set on device d (d < 32) */   \
        if ((bytes) > 65536) {                                                                            \
            int dev_ = 0;                                                                                 \
            (void)hipGetDevice(&dev_);                                                                    \
            const unsigned bit_ = 1u << (dev_ & 31);                                                      \
            if (!(done_.load(std::memory_order_relaxed) & bit_)) {                                        \
                (void)hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)(bytes));                                                  \
                done_.fetch_or(bit_, std::memory_order_relaxed);                                          \
            }                                                                                             \
        }                                                                                                 \
    } while (0)
#endif
