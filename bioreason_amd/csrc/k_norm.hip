// k_norm.hip — row normalisations, per-head QK-norm + RoPE, SwiGLU.  All are
// HBM-bound: one wave64 per row, 16-byte bf16x8 loads, fp32 statistics,
// wavefront butterfly reductions (no LDS, no block barrier).
#include "bra_device.h"
#include "bra_api_internal.h"

namespace bra {

// ---------------------------------------------------------------------------
// RMSNorm forward  (TF:models/qwen3/modeling_qwen3.py:59-64)
//   y = w * bf16( x_f32 * rsqrt(mean(x^2) + eps) )
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* x, long ldx, const bf16_t* w, bf16_t* y,
                                                          long ldy, float* rstd_out, int rows, int cols, float eps) {
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int lane = lane_id();
    const bool live = row < rows;
    const bf16_t* xr = x + (long)(live ? row : 0) * ldx;
    float ss = 0.f;
    for (int c = lane * 8; c < cols; c += 512) {
        float f[8];
        unpack8(ld16(xr + c), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
    }
    ss = wave_sum<64>(ss);
    const float rstd = rsqrtf(ss / (float)cols + eps);
    if (!live) return;
    if (rstd_out && lane == 0) rstd_out[row] = rstd;
    bf16_t* yr = y + (long)row * ldy;
    for (int c = lane * 8; c < cols; c += 512) {
        float f[8], g[8];
        unpack8(ld16(xr + c), f);
        unpack8(ld16(w + c), g);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = g[i] * round_bf(f[i] * rstd);
        st16(yr + c, pack8(f));
    }
}

// RMSNorm backward (input gradient only: norm weights are frozen under LoRA,
// train_dna_qwen.py:152-167).  dx = rstd * (g - xhat * mean(g * xhat)) [+ dres],
// g = w * dy, xhat = x * rstd.
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* dy, long lddy, const bf16_t* x, long ldx,
                                                          const bf16_t* w, const bf16_t* dres, long lddres,
                                                          bf16_t* dx, long lddx, int rows, int cols, float eps) {
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int lane = lane_id();
    const bool live = row < rows;
    const long r = live ? row : 0;
    const bf16_t* xr = x + r * ldx;
    const bf16_t* dyr = dy + r * lddy;
    float ss = 0.f, dot = 0.f;
    for (int c = lane * 8; c < cols; c += 512) {
        float f[8], d[8], g[8];
        unpack8(ld16(xr + c), f);
        unpack8(ld16(dyr + c), d);
        unpack8(ld16(w + c), g);
#pragma unroll
        for (int i = 0; i < 8; ++i) { ss += f[i] * f[i]; dot += g[i] * d[i] * f[i]; }
    }
    ss = wave_sum<64>(ss);
    dot = wave_sum<64>(dot);
    if (!live) return;
    const float rstd = rsqrtf(ss / (float)cols + eps);
    const float cmean = dot * rstd / (float)cols;   // mean(g * xhat)
    bf16_t* dxr = dx + r * lddx;
    for (int c = lane * 8; c < cols; c += 512) {
        float f[8], d[8], g[8], o[8];
        unpack8(ld16(xr + c), f);
        unpack8(ld16(dyr + c), d);
        unpack8(ld16(w + c), g);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = rstd * (g[i] * d[i] - f[i] * rstd * cmean);
        if (dres) {
            float e[8];
            unpack8(ld16(dres + r * lddres + c), e);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += e[i];
        }
        st16(dxr + c, pack8(o));
    }
}

// LayerNorm forward (TF:models/esm/modeling_esm.py:418,480,529; eps 1e-12)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* x, long ldx, const bf16_t* w,
                                                            const bf16_t* b, bf16_t* y, long ldy, int rows, int cols,
                                                            float eps) {
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int lane = lane_id();
    const bool live = row < rows;
    const bf16_t* xr = x + (long)(live ? row : 0) * ldx;
    float s = 0.f;
    for (int c = lane * 8; c < cols; c += 512) {
        float f[8];
        unpack8(ld16(xr + c), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += f[i];
    }
    const float mean = wave_sum<64>(s) / (float)cols;
    float v = 0.f;
    for (int c = lane * 8; c < cols; c += 512) {
        float f[8];
        unpack8(ld16(xr + c), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { float d = f[i] - mean; v += d * d; }
    }
    const float rstd = rsqrtf(wave_sum<64>(v) / (float)cols + eps);
    if (!live) return;
    bf16_t* yr = y + (long)row * ldy;
    for (int c = lane * 8; c < cols; c += 512) {
        float f[8], g[8], h[8];
        unpack8(ld16(xr + c), f);
        unpack8(ld16(w + c), g);
        unpack8(ld16(b + c), h);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (f[i] - mean) * rstd * g[i] + h[i];
        st16(yr + c, pack8(f));
    }
}

// ---------------------------------------------------------------------------
// SwiGLU  act = silu(gu[:, :F]) * gu[:, F:]   (TF:qwen3:81-83; NT-v2 hub FFN)
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }

__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* gu, long ldgu, bf16_t* act, long ldact,
                                                         int rows, int F) {
    const long nvec = (long)rows * (F / 8);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const long r = i / (F / 8);
        const int c = (int)(i % (F / 8)) * 8;
        float g[8], u[8];
        unpack8(ld16(gu + r * ldgu + c), g);
        unpack8(ld16(gu + r * ldgu + F + c), u);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = round_bf(silu_f(g[k])) * u[k];
        st16(act + r * ldact + c, pack8(g));
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* gu, long ldgu, const bf16_t* dact,
                                                         long lddact, bf16_t* dgu, long lddgu, int rows, int F) {
    const long nvec = (long)rows * (F / 8);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const long r = i / (F / 8);
        const int c = (int)(i % (F / 8)) * 8;
        float g[8], u[8], d[8], dg[8], du[8];
        unpack8(ld16(gu + r * ldgu + c), g);
        unpack8(ld16(gu + r * ldgu + F + c), u);
        unpack8(ld16(dact + r * lddact + c), d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float sg = 1.f / (1.f + __expf(-g[k]));
            const float si = g[k] * sg;
            du[k] = d[k] * si;
            dg[k] = d[k] * u[k] * (sg * (1.f + g[k] * (1.f - sg)));
        }
        st16(dgu + r * lddgu + c, pack8(dg));
        st16(dgu + r * lddgu + F + c, pack8(du));
    }
}

// ---------------------------------------------------------------------------
// Per-head (optional) RMSNorm + rotate-half RoPE on a fused QKV projection.
//   Qwen3: q = rope(q_norm(q_lin)), k = rope(k_norm(k_lin)), v = v_lin   (TF:qwen3:237-256, 140-170)
//   NT-v2: q = rope(q_lin * hd^-0.5), k = rope(k_lin), v = v_lin        (TF:esm:374-378)
// One thread per rotation pair (d, d + hd/2) of one (token, head); the pairs
// of one head are `hd/2` consecutive lanes, so the per-head sum of squares is a
// sub-wave butterfly.  Outputs go through (batch, seq, head) strides so the
// same kernel fills token-major buffers, head-major buffers or the KV cache.
struct RopeArgs {
    const bf16_t* qkv; long ldqkv;      // [T, (Hq + 2 Hkv) * hd]
    const bf16_t* qw; const bf16_t* kw; // per-head norm weights [hd] or null
    const float* cosT; const float* sinT;  // [npos, hd/2]
    const int* pos;                     // [T] rotary position of each token
    int T, S, Hq, Hkv, hd;
    float eps, qscale;
    bf16_t* q; long q_sb, q_ss, q_sh;
    bf16_t* k; long k_sb, k_ss, k_sh;
    bf16_t* v; long v_sb, v_ss, v_sh;
    int s_off;                          // sequence offset added to s in the K/V destination (cache append)
};

__global__ __launch_bounds__(256) void qk_norm_rope_fwd_kernel(RopeArgs a) {
    const int half = a.hd >> 1;
    const int H = a.Hq + 2 * a.Hkv;
    const long total = (long)a.T * H * half;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < total;
    const long cl = live ? idx : total - 1;
    const int d = (int)(cl % half);
    const int h = (int)((cl / half) % H);
    const int t = (int)(cl / ((long)half * H));
    const bf16_t* src = a.qkv + (long)t * a.ldqkv + (long)h * a.hd;
    float x1 = bf2f(src[d]), x2 = bf2f(src[d + half]);
    const bool is_q = h < a.Hq, is_k = !is_q && h < a.Hq + a.Hkv;
    const bf16_t* nw = is_q ? a.qw : (is_k ? a.kw : nullptr);
    // sum of squares over the head (all lanes run the butterfly; width = half)
    float ss = x1 * x1 + x2 * x2;
    for (int m = half >> 1; m >= 1; m >>= 1) ss += wave_shfl_xor(ss, m);
    if (nw) {
        const float rstd = rsqrtf(ss / (float)a.hd + a.eps);
        x1 = round_bf(bf2f(nw[d]) * round_bf(x1 * rstd));
        x2 = round_bf(bf2f(nw[d + half]) * round_bf(x2 * rstd));
    }
    if (is_q && a.qscale != 1.f) { x1 = round_bf(x1 * a.qscale); x2 = round_bf(x2 * a.qscale); }
    float o1 = x1, o2 = x2;
    if (is_q || is_k) {
        const int p = a.pos[t];
        const float c = a.cosT[(long)p * half + d], s = a.sinT[(long)p * half + d];
        o1 = x1 * c - x2 * s;
        o2 = x2 * c + x1 * s;
    }
    if (!live) return;
    const int b = t / a.S, sidx = t % a.S;
    bf16_t* dst;
    if (is_q) dst = a.q + b * a.q_sb + sidx * a.q_ss + (long)h * a.q_sh;
    else if (is_k) dst = a.k + b * a.k_sb + (sidx + a.s_off) * a.k_ss + (long)(h - a.Hq) * a.k_sh;
    else dst = a.v + b * a.v_sb + (sidx + a.s_off) * a.v_ss + (long)(h - a.Hq - a.Hkv) * a.v_sh;
    dst[d] = f2bf(o1);
    dst[d + half] = f2bf(o2);
}

// Backward of the above for the Qwen3 training path: inputs are dq/dk/dv (any
// strides), output is d(qkv_lin) token-major [T, (Hq+2Hkv)*hd].
struct RopeBwdArgs {
    const bf16_t* qkv; long ldqkv;      // saved forward input
    const bf16_t* qw; const bf16_t* kw;
    const float* cosT; const float* sinT;
    const int* pos;
    int T, S, Hq, Hkv, hd;
    float eps, qscale;
    const bf16_t* dq; long q_sb, q_ss, q_sh;
    const bf16_t* dk; long k_sb, k_ss, k_sh;
    const bf16_t* dv; long v_sb, v_ss, v_sh;
    bf16_t* dqkv; long lddqkv;
};

__global__ __launch_bounds__(256) void qk_norm_rope_bwd_kernel(RopeBwdArgs a) {
    const int half = a.hd >> 1;
    const int H = a.Hq + 2 * a.Hkv;
    const long total = (long)a.T * H * half;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < total;
    const long cl = live ? idx : total - 1;
    const int d = (int)(cl % half);
    const int h = (int)((cl / half) % H);
    const int t = (int)(cl / ((long)half * H));
    const bool is_q = h < a.Hq, is_k = !is_q && h < a.Hq + a.Hkv;
    const int b = t / a.S, sidx = t % a.S;
    const bf16_t* gsrc;
    if (is_q) gsrc = a.dq + b * a.q_sb + sidx * a.q_ss + (long)h * a.q_sh;
    else if (is_k) gsrc = a.dk + b * a.k_sb + sidx * a.k_ss + (long)(h - a.Hq) * a.k_sh;
    else gsrc = a.dv + b * a.v_sb + sidx * a.v_ss + (long)(h - a.Hq - a.Hkv) * a.v_sh;
    float g1 = bf2f(gsrc[d]), g2 = bf2f(gsrc[d + half]);
    if (is_q || is_k) {   // transpose of the rotation
        const int p = a.pos[t];
        const float c = a.cosT[(long)p * half + d], s = a.sinT[(long)p * half + d];
        const float r1 = g1 * c + g2 * s;
        const float r2 = g2 * c - g1 * s;
        g1 = r1; g2 = r2;
    }
    if (is_q && a.qscale != 1.f) { g1 *= a.qscale; g2 *= a.qscale; }
    const bf16_t* nw = is_q ? a.qw : (is_k ? a.kw : nullptr);
    const bf16_t* src = a.qkv + (long)t * a.ldqkv + (long)h * a.hd;
    const float x1 = bf2f(src[d]), x2 = bf2f(src[d + half]);
    float ss = x1 * x1 + x2 * x2;
    float dot = 0.f;
    float w1 = 1.f, w2 = 1.f;
    if (nw) { w1 = bf2f(nw[d]); w2 = bf2f(nw[d + half]); dot = w1 * g1 * x1 + w2 * g2 * x2; }
    for (int m = half >> 1; m >= 1; m >>= 1) { ss += wave_shfl_xor(ss, m); dot += wave_shfl_xor(dot, m); }
    if (nw) {
        const float rstd = rsqrtf(ss / (float)a.hd + a.eps);
        const float cm = dot * rstd / (float)a.hd;
        g1 = rstd * (w1 * g1 - x1 * rstd * cm);
        g2 = rstd * (w2 * g2 - x2 * rstd * cm);
    }
    if (!live) return;
    bf16_t* dst = a.dqkv + (long)t * a.lddqkv + (long)h * a.hd;
    dst[d] = f2bf(g1);
    dst[d + half] = f2bf(g2);
}

// ---- vectorised forms (hd = 64 or 128, every stride a multiple of 8 elements): HD / 8 lanes per head, 16 bytes per lane.
// Lane j of a head owns dims 8j .. 8j+7; its rotation partner (dims +- hd/2) is lane j ^ (LPH / 2).  Same arithmetic and
// rounding points as the scalar kernels above (only the order of the sum-of-squares butterfly differs).
template <int HD>
__global__ __launch_bounds__(256) void qk_norm_rope_fwd_vec_kernel(RopeArgs a) {
    constexpr int LPH = HD / 8, HALF = HD / 2;
    const int H = a.Hq + 2 * a.Hkv;
    const long nhead = (long)a.T * H;
    const long gl = (long)blockIdx.x * 256 + threadIdx.x;
    const long hidx = gl / LPH;
    const int j = (int)(gl % LPH);
    const bool live = hidx < nhead;
    const long hc = live ? hidx : nhead - 1;
    const int h = (int)(hc % H);
    const int t = (int)(hc / H);
    const bool is_q = h < a.Hq, is_k = !is_q && h < a.Hq + a.Hkv;
    const bool upper = j >= LPH / 2;
    const int dh = (j & (LPH / 2 - 1)) * 8;                     // index into the half-dim cos / sin row
    float x[8];
    unpack8(ld16(a.qkv + (long)t * a.ldqkv + (long)h * HD + j * 8), x);
    const bf16_t* nw = is_q ? a.qw : (is_k ? a.kw : nullptr);
    u32x4 nwv = {0u, 0u, 0u, 0u};
    if (nw) nwv = ld16(nw + j * 8);
    f32x4 c0 = {1.f, 1.f, 1.f, 1.f}, c1 = c0, s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (is_q || is_k) {
        const int p = a.pos[t];
        const float* cp = a.cosT + (long)p * HALF + dh;
        const float* sp = a.sinT + (long)p * HALF + dh;
        c0 = *reinterpret_cast<const f32x4*>(cp); c1 = *reinterpret_cast<const f32x4*>(cp + 4);
        s0 = *reinterpret_cast<const f32x4*>(sp); s1 = *reinterpret_cast<const f32x4*>(sp + 4);
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
#pragma unroll
    for (int m = LPH >> 1; m >= 1; m >>= 1) ss += wave_shfl_xor(ss, m);
    if (nw) {
        const float rstd = rsqrtf(ss / (float)HD + a.eps);
        float w[8];
        unpack8(nwv, w);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = round_bf(w[i] * round_bf(x[i] * rstd));
    }
    if (is_q && a.qscale != 1.f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = round_bf(x[i] * a.qscale);
    }
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float other = wave_shfl_xor(x[i], LPH / 2);
        const float c = i < 4 ? c0[i & 3] : c1[i & 3], sn = i < 4 ? s0[i & 3] : s1[i & 3];
        o[i] = (is_q || is_k) ? (upper ? x[i] * c + other * sn : x[i] * c - other * sn) : x[i];
    }
    if (!live) return;
    const int b = t / a.S, sidx = t % a.S;
    bf16_t* dst;
    if (is_q) dst = a.q + b * a.q_sb + sidx * a.q_ss + (long)h * a.q_sh;
    else if (is_k) dst = a.k + b * a.k_sb + (sidx + a.s_off) * a.k_ss + (long)(h - a.Hq) * a.k_sh;
    else dst = a.v + b * a.v_sb + (sidx + a.s_off) * a.v_ss + (long)(h - a.Hq - a.Hkv) * a.v_sh;
    st16(dst + j * 8, pack8(o));
}

template <int HD>
__global__ __launch_bounds__(256) void qk_norm_rope_bwd_vec_kernel(RopeBwdArgs a) {
    constexpr int LPH = HD / 8, HALF = HD / 2;
    const int H = a.Hq + 2 * a.Hkv;
    const long nhead = (long)a.T * H;
    const long gl = (long)blockIdx.x * 256 + threadIdx.x;
    const long hidx = gl / LPH;
    const int j = (int)(gl % LPH);
    const bool live = hidx < nhead;
    const long hc = live ? hidx : nhead - 1;
    const int h = (int)(hc % H);
    const int t = (int)(hc / H);
    const bool is_q = h < a.Hq, is_k = !is_q && h < a.Hq + a.Hkv;
    const bool upper = j >= LPH / 2;
    const int dh = (j & (LPH / 2 - 1)) * 8;
    const int b = t / a.S, sidx = t % a.S;
    const bf16_t* gsrc;
    if (is_q) gsrc = a.dq + b * a.q_sb + sidx * a.q_ss + (long)h * a.q_sh;
    else if (is_k) gsrc = a.dk + b * a.k_sb + sidx * a.k_ss + (long)(h - a.Hq) * a.k_sh;
    else gsrc = a.dv + b * a.v_sb + sidx * a.v_ss + (long)(h - a.Hq - a.Hkv) * a.v_sh;
    float g[8], x[8];
    unpack8(ld16(gsrc + j * 8), g);
    unpack8(ld16(a.qkv + (long)t * a.ldqkv + (long)h * HD + j * 8), x);
    const bf16_t* nw = is_q ? a.qw : (is_k ? a.kw : nullptr);
    float w[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (nw) unpack8(ld16(nw + j * 8), w);
    if (is_q || is_k) {   // transpose of the rotation: first half r1 = g1 c + g2 s, second half r2 = g2 c - g1 s
        const int p = a.pos[t];
        const float* cp = a.cosT + (long)p * HALF + dh;
        const float* sp = a.sinT + (long)p * HALF + dh;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp), c1 = *reinterpret_cast<const f32x4*>(cp + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float other = wave_shfl_xor(g[i], LPH / 2);
            const float c = i < 4 ? c0[i & 3] : c1[i & 3], sn = i < 4 ? s0[i & 3] : s1[i & 3];
            g[i] = upper ? g[i] * c - other * sn : g[i] * c + other * sn;
        }
    }
    if (is_q && a.qscale != 1.f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] *= a.qscale;
    }
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ss += x[i] * x[i]; if (nw) dot += w[i] * g[i] * x[i]; }
#pragma unroll
    for (int m = LPH >> 1; m >= 1; m >>= 1) { ss += wave_shfl_xor(ss, m); dot += wave_shfl_xor(dot, m); }
    if (nw) {
        const float rstd = rsqrtf(ss / (float)HD + a.eps);
        const float cm = dot * rstd / (float)HD;
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = rstd * (w[i] * g[i] - x[i] * rstd * cm);
    }
    if (!live) return;
    st16(a.dqkv + (long)t * a.lddqkv + (long)h * HD + j * 8, pack8(g));
}

}  // namespace bra

using namespace bra;

static inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int bra_rmsnorm_fwd(const void* x, long ldx, const void* w, void* y, long ldy, float* rstd, int rows,
                               int cols, float eps, void* stream) {
    if (rows == 0) return 0;
    if (!x || !w || !y || cols <= 0 || cols % 8 || ldx % 8 || ldy % 8) return BRA_ERR_ARG;
    BRA_LAUNCH(rmsnorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16_t*)x, ldx,
               (const bf16_t*)w, (bf16_t*)y, ldy, rstd, rows, cols, eps);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_rmsnorm_bwd(const void* dy, long lddy, const void* x, long ldx, const void* w, const void* dres,
                               long lddres, void* dx, long lddx, int rows, int cols, float eps, void* stream) {
    if (rows == 0) return 0;
    if (!dy || !x || !w || !dx || cols <= 0 || cols % 8 || ldx % 8 || lddy % 8 || lddx % 8 || (dres && lddres % 8))
        return BRA_ERR_ARG;
    BRA_LAUNCH(rmsnorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16_t*)dy, lddy,
               (const bf16_t*)x, ldx, (const bf16_t*)w, (const bf16_t*)dres, lddres, (bf16_t*)dx, lddx, rows, cols,
               eps);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_layernorm_fwd(const void* x, long ldx, const void* w, const void* b, void* y, long ldy, int rows,
                                 int cols, float eps, void* stream) {
    if (rows == 0) return 0;
    if (!x || !w || !b || !y || cols <= 0 || cols % 8 || ldx % 8 || ldy % 8) return BRA_ERR_ARG;
    BRA_LAUNCH(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16_t*)x, ldx,
               (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, ldy, rows, cols, eps);
    return BRA_LAUNCH_STATUS();
}

static inline int ew_grid(long nvec) {
    long g = (nvec + 255) / 256;
    if (g > 2048) g = 2048;   // grid-stride past 8 blocks per CU
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int bra_swiglu_fwd(const void* gu, long ldgu, void* act, long ldact, int rows, int F, void* stream) {
    if (rows == 0) return 0;
    if (!gu || !act || F <= 0 || F % 8 || ldgu % 8 || ldact % 8) return BRA_ERR_ARG;
    BRA_LAUNCH(swiglu_fwd_kernel, dim3(ew_grid((long)rows * (F / 8))), dim3(256), 0, stream, (const bf16_t*)gu, ldgu,
               (bf16_t*)act, ldact, rows, F);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_swiglu_bwd(const void* gu, long ldgu, const void* dact, long lddact, void* dgu, long lddgu,
                              int rows, int F, void* stream) {
    if (rows == 0) return 0;
    if (!gu || !dact || !dgu || F <= 0 || F % 8 || ldgu % 8 || lddact % 8 || lddgu % 8) return BRA_ERR_ARG;
    BRA_LAUNCH(swiglu_bwd_kernel, dim3(ew_grid((long)rows * (F / 8))), dim3(256), 0, stream, (const bf16_t*)gu, ldgu,
               (const bf16_t*)dact, lddact, (bf16_t*)dgu, lddgu, rows, F);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_qk_norm_rope_fwd(const void* qkv, long ldqkv, const void* qw, const void* kw, const float* cosT,
                                    const float* sinT, const int* pos, int T, int S, int Hq, int Hkv, int hd,
                                    float eps, float qscale, void* q, long q_sb, long q_ss, long q_sh, void* k,
                                    long k_sb, long k_ss, long k_sh, void* v, long v_sb, long v_ss, long v_sh,
                                    int s_off, void* stream) {
    if (T == 0) return 0;
    if (!qkv || !cosT || !sinT || !pos || !q || !k || !v || S <= 0 || T % S) return BRA_ERR_ARG;
    if (!pow2(hd) || hd < 2 || hd > 128) return BRA_ERR_UNSUPPORTED;
    RopeArgs a = {(const bf16_t*)qkv, ldqkv, (const bf16_t*)qw, (const bf16_t*)kw, cosT, sinT, pos, T, S, Hq, Hkv,
                  hd, eps, qscale, (bf16_t*)q, q_sb, q_ss, q_sh, (bf16_t*)k, k_sb, k_ss, k_sh, (bf16_t*)v, v_sb,
                  v_ss, v_sh, s_off};
    const long total = (long)T * (Hq + 2 * Hkv) * (hd / 2);
    const bool vec = (hd == 128 || hd == 64) && ldqkv % 8 == 0 && !((q_sb | q_ss | q_sh | k_sb | k_ss | k_sh | v_sb | v_ss | v_sh) & 7);
    if (vec) {
        const long lanes = (long)T * (Hq + 2 * Hkv) * (hd / 8);
        const dim3 grid((unsigned)((lanes + 255) / 256));
        if (hd == 128) BRA_LAUNCH((qk_norm_rope_fwd_vec_kernel<128>), grid, dim3(256), 0, stream, a);
        else BRA_LAUNCH((qk_norm_rope_fwd_vec_kernel<64>), grid, dim3(256), 0, stream, a);
        return BRA_LAUNCH_STATUS();
    }
    BRA_LAUNCH(qk_norm_rope_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_qk_norm_rope_bwd(const void* qkv, long ldqkv, const void* qw, const void* kw, const float* cosT,
                                    const float* sinT, const int* pos, int T, int S, int Hq, int Hkv, int hd,
                                    float eps, float qscale, const void* dq, long q_sb, long q_ss, long q_sh,
                                    const void* dk, long k_sb, long k_ss, long k_sh, const void* dv, long v_sb,
                                    long v_ss, long v_sh, void* dqkv, long lddqkv, void* stream) {
    if (T == 0) return 0;
    if (!qkv || !cosT || !sinT || !pos || !dq || !dk || !dv || !dqkv || S <= 0 || T % S) return BRA_ERR_ARG;
    if (!pow2(hd) || hd < 2 || hd > 128) return BRA_ERR_UNSUPPORTED;
    RopeBwdArgs a = {(const bf16_t*)qkv, ldqkv, (const bf16_t*)qw, (const bf16_t*)kw, cosT, sinT, pos, T, S, Hq,
                     Hkv, hd, eps, qscale, (const bf16_t*)dq, q_sb, q_ss, q_sh, (const bf16_t*)dk, k_sb, k_ss, k_sh,
                     (const bf16_t*)dv, v_sb, v_ss, v_sh, (bf16_t*)dqkv, lddqkv};
    const long total = (long)T * (Hq + 2 * Hkv) * (hd / 2);
    const bool vec = (hd == 128 || hd == 64) && ldqkv % 8 == 0 && lddqkv % 8 == 0 &&
                     !((q_sb | q_ss | q_sh | k_sb | k_ss | k_sh | v_sb | v_ss | v_sh) & 7);
    if (vec) {
        const long lanes = (long)T * (Hq + 2 * Hkv) * (hd / 8);
        const dim3 grid((unsigned)((lanes + 255) / 256));
        if (hd == 128) BRA_LAUNCH((qk_norm_rope_bwd_vec_kernel<128>), grid, dim3(256), 0, stream, a);
        else BRA_LAUNCH((qk_norm_rope_bwd_vec_kernel<64>), grid, dim3(256), 0, stream, a);
        return BRA_LAUNCH_STATUS();
    }
    BRA_LAUNCH(qk_norm_rope_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return BRA_LAUNCH_STATUS();
}
