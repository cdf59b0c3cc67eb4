// k_quant.hip — per-row fp8 (OCP e4m3) images for the fp8 x fp8 GEMM (k_gemm.hip: gemm_fp8_kernel; BASELINE config 5):
//   weights     q[n, k] = e4m3(W[n, k] w[k] / s[n]),  s[n] = max_k |W[n, k] w[k]| / 448      (w = the input's RMSNorm weight, folded; the
//               rule and the rounding of bra_dec_pack_weights_fp8 — the token loop's image of the same weight holds the same bytes)
//   activations q[m, k] = e4m3(x[m, k] / a[m]),  a[m] = max_k |x[m, k]| / 448; the scale handed to the GEMM is a[m] (plain rows) or
//               rstd[m] a[m] (rows that feed a projection whose weights carry the norm weight: y = rstd (x (W w)^T), as the token loop)
//   SwiGLU rows the same over act = bf16(bf16(silu(g)) u), computed on the fly from the [gate | up] rows
// e4m3 encode: round to nearest even, saturating (enc8_e4m3: v_cvt_pk_fp8_f32 on the device, f32_to_e4m3 in the emulator — the same bytes).
#include "bra_device.h"
#include "bra_api_internal.h"

namespace bra {

__device__ __forceinline__ float silu_q(float x) { return x / (1.f + __expf(-x)); }

// 8 scaled values -> 8 e4m3 bytes.  Device: v_cvt_pk_fp8_f32 (round to nearest even; the inputs are clamped to +-448 first, so the
// instruction's overflow mode never matters) — the integer encoder costs ~25 VALU per element and made these kernels compute-bound
// (swiglu_quant 41.6 us for 2180 x 6144, profiles/r6_q_fp8_trace.txt).  Emulator: f32_to_e4m3.  Same bytes for every finite input
// (tests/test_fp8_gemm.py::test_quant_encoder_is_nearest_even_over_every_bf16_value).
__device__ __forceinline__ u32x2 enc8_e4m3(const float* f, float inv) {
#ifdef BRA_EMU
    unsigned b[8];
    for (int i = 0; i < 8; ++i) b[i] = f32_to_e4m3(f[i] * inv);
    u32x2 o = {b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24), b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24)};
    return o;
#else
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_fmed3f(f[i] * inv, -448.f, 448.f);
    unsigned lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], hi, true);
    u32x2 o = {lo, hi};
    return o;
#endif
}

// One wave per row, four rows per workgroup.  The row lives in registers: NV 16-byte chunks per lane, all requested before the first
// is used (a loop of dependent round trips measured 12 us for 2180 x 2048 — 1 TB/s; profiles/r6_o_gemm_fp8_probe.txt), one pass.
// K <= 512 NV; chunks past K / 8 are requested from chunk 0 and ignored.
template <int NV>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* x, long ldx, int M, int K, const bf16_t* colw, unsigned char* q,
                                                             long ldq, float* scale, int rms, float eps) {
    const int row = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (row >= M) return;
    const bf16_t* xr = x + (long)row * ldx;
    const int nch = K / 8;
    u32x4 xv[NV], wv[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) { const int j = lane + 64 * u; xv[u] = ld16(xr + (j < nch ? j : 0) * 8); }
    if (colw) {
#pragma unroll
        for (int u = 0; u < NV; ++u) { const int j = lane + 64 * u; wv[u] = ld16(colw + (j < nch ? j : 0) * 8); }
    }
    float f[NV][8];
    float mx = 0.f, ssq = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const bool live = lane + 64 * u < nch;
        unpack8(xv[u], f[u]);
#pragma unroll
        for (int i = 0; i < 8; ++i) { f[u][i] = live ? f[u][i] : 0.f; ssq += f[u][i] * f[u][i]; }
        if (colw) { float s8[8]; unpack8(wv[u], s8);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[u][i] *= s8[i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(f[u][i]));
    }
    for (int m = 32; m >= 1; m >>= 1) { mx = fmaxf(mx, wave_shfl_xor(mx, m)); ssq += wave_shfl_xor(ssq, m); }
    const float a = mx > 0.f ? mx * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / a;
    if (lane == 0) scale[row] = rms ? a * rsqrtf(ssq / (float)K + eps) : a;
    unsigned char* qr = q + (long)row * ldq;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = lane + 64 * u;
        const u32x2 o = enc8_e4m3(f[u], inv);
        if (j < nch) *reinterpret_cast<u32x2*>(qr + j * 8) = o;
    }
}

// F <= 512 NV: gate and up chunks of the row in registers, act computed once
template <int NV>
__global__ __launch_bounds__(256) void swiglu_quant_fp8_kernel(const bf16_t* gu, long ldgu, int M, int F, unsigned char* q, long ldq,
                                                               float* scale) {
    const int row = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (row >= M) return;
    const bf16_t* gr = gu + (long)row * ldgu;
    const int nch = F / 8;
    u32x4 gv[NV], uv[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) { const int j = lane + 64 * u; const int jj = j < nch ? j : 0; gv[u] = ld16(gr + jj * 8); uv[u] = ld16(gr + F + jj * 8); }
    float mx = 0.f;
    float act[NV][8];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const bool live = lane + 64 * u < nch;
        float g[8], w[8];
        unpack8(gv[u], g);
        unpack8(uv[u], w);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            act[u][i] = live ? round_bf(round_bf(silu_q(g[i])) * w[i]) : 0.f;
            mx = fmaxf(mx, fabsf(act[u][i]));
        }
    }
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, wave_shfl_xor(mx, m));
    const float a = mx > 0.f ? mx * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / a;
    if (lane == 0) scale[row] = a;
    unsigned char* qr = q + (long)row * ldq;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = lane + 64 * u;
        const u32x2 o = enc8_e4m3(act[u], inv);
        if (j < nch) *reinterpret_cast<u32x2*>(qr + j * 8) = o;
    }
}

}  // namespace bra

using namespace bra;

// x [M, K] bf16 (row stride ldx elements, K % 8 == 0) -> q [M, K] e4m3 bytes (row stride ldq bytes, % 8) + scale [M] fp32.
// colw (optional, bf16 [K]): multiplied into the row before quantisation (a weight matrix with its input's RMSNorm weight folded in).
// rms != 0: scale[m] = rstd[m] * absmax / 448 with rstd = rsqrt(mean_k x^2 + eps) — the row factor of a folded-norm projection rides on
// the activation scale (statistics over x itself, not x * colw).
extern "C" int bra_quant_rows_fp8(const void* x, long ldx, int M, int K, const void* colw, void* q, long ldq, float* scale, int rms,
                                  float eps, void* stream) {
    if (M == 0) return 0;
    if (!x || !q || !scale || M < 0 || K <= 0 || K % 8 || ldx % 8 || ldq % 8) return BRA_ERR_ARG;
    if (K > 512 * 24) return BRA_ERR_UNSUPPORTED;
#define BRA_QR(NV_) BRA_LAUNCH((quant_rows_fp8_kernel<NV_>), dim3((M + 3) / 4), dim3(256), 0, (bra_stream_t)stream, (const bf16_t*)x, ldx, M, \
                               K, (const bf16_t*)colw, (unsigned char*)q, ldq, scale, rms, eps)
    if (K <= 512 * 4) BRA_QR(4); else if (K <= 512 * 8) BRA_QR(8); else if (K <= 512 * 12) BRA_QR(12); else BRA_QR(24);
#undef BRA_QR
    return BRA_LAUNCH_STATUS();
}

// gu [M, 2 F] bf16 ([gate | up]) -> q [M, F] e4m3 of act = bf16(bf16(silu(gate)) up) + scale [M]
extern "C" int bra_swiglu_quant_fp8(const void* gu, long ldgu, int M, int F, void* q, long ldq, float* scale, void* stream) {
    if (M == 0) return 0;
    if (!gu || !q || !scale || M < 0 || F <= 0 || F % 8 || ldgu % 8 || ldq % 8) return BRA_ERR_ARG;
    if (F > 512 * 20) return BRA_ERR_UNSUPPORTED;
#define BRA_SQ(NV_) BRA_LAUNCH((swiglu_quant_fp8_kernel<NV_>), dim3((M + 3) / 4), dim3(256), 0, (bra_stream_t)stream, (const bf16_t*)gu, ldgu, \
                               M, F, (unsigned char*)q, ldq, scale)
    if (F <= 512 * 6) BRA_SQ(6); else if (F <= 512 * 12) BRA_SQ(12); else BRA_SQ(20);
#undef BRA_SQ
    return BRA_LAUNCH_STATUS();
}
