// k_misc.hip — the small HBM-bound pieces around the GEMM / attention kernels:
// head transposes (K-contiguous images for the attention contractions), the
// attention-backward delta, token embedding + DNA-row scatter, bias-gradient
// column sums, fp32->bf16 pack/transposes of the trainable (LoRA / projection)
// parameters, the log-sum-exp merge of the fused lm_head, and AdamW.
#include "bra_device.h"
#include "bra_api_internal.h"

namespace bra {

// ---------------------------------------------------------------------------
// x[b, s, h, :] (any b/s/h strides, d contiguous) -> xT[b, h, d, s] with row pitch `pitch`
// (pitch = multiple of 64 >= S); positions s >= S are written as zeros so the consumer's
// tile loads never see non-finite padding.
template <int HD>
__global__ __launch_bounds__(256) void head_transpose_kernel(const bf16_t* x, long sb, long ss, long sh, bf16_t* xt,
                                                             long t_sb, long t_sh, long pitch, int S) {
    constexpr int LD = HD + 2;
    __shared__ bf16_t tile[64 * LD];
    const int tid = (int)threadIdx.x;
    const int s0 = (int)blockIdx.x * 64, h = (int)blockIdx.y, b = (int)blockIdx.z;
    const bf16_t* xp = x + b * sb + h * sh;
    constexpr int CH = HD / 8;
    for (int q = tid; q < 64 * CH; q += 256) {
        const int r = q / CH, c = q % CH;
        const int s = s0 + r;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (s < S) v = ld16(xp + (long)s * ss + c * 8);
        uint32_t* tp = reinterpret_cast<uint32_t*>(&tile[r * LD + c * 8]);   // LD even -> 4-byte aligned
        tp[0] = v.x; tp[1] = v.y; tp[2] = v.z; tp[3] = v.w;
    }
    __syncthreads();
    bf16_t* op = xt + b * t_sb + h * t_sh;
    for (int q = tid; q < HD * 8; q += 256) {
        const int d = q >> 3, c = q & 7;
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = tile[(c * 8 + 2 * i) * LD + d], hi = tile[(c * 8 + 2 * i + 1) * LD + d];
            w[i] = lo | (hi << 16);
        }
        u32x4 v = {w[0], w[1], w[2], w[3]};
        st16(op + (long)d * pitch + s0 + c * 8, v);
    }
}

// delta[b, h, s] = sum_d dO[b,s,h,d] * O[b,s,h,d]   (flash-attention backward preprocess)
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* dout, long do_sb, long do_ss, long do_sh,
                                                         const bf16_t* o, long o_sb, long o_ss, long o_sh,
                                                         float* delta, int B, int S, int H, int hd) {
    const int lpr = hd / 8;                       // lanes per (token, head) row: 4, 8 or 16
    const long total = (long)B * H * S * lpr;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < total;
    const long cl = live ? idx : total - 1;
    const int c = (int)(cl % lpr);
    const long row = cl / lpr;                    // (b*H + h)*S + s
    const int s = (int)(row % S);
    const int h = (int)((row / S) % H);
    const int b = (int)(row / ((long)S * H));
    float f[8], g[8];
    unpack8(ld16(dout + b * do_sb + (long)s * do_ss + h * do_sh + c * 8), f);
    unpack8(ld16(o + b * o_sb + (long)s * o_ss + h * o_sh + c * 8), g);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += f[i] * g[i];
    for (int m = lpr >> 1; m >= 1; m >>= 1) acc += wave_shfl_xor(acc, m);
    if (live && c == 0) delta[row] = acc;
}

// ---------------------------------------------------------------------------
// DNA-row scatter plan (dna_llm.py:163-177 + :216-229 without the per-sequence .item() syncs).
// The reference keeps the FIRST valid_len = attention_mask.sum() rows of every projected DNA
// sequence, concatenates them per sample in batch_idx_map order, then sample by sample, and
// writes the k-th row of that list over the k-th <|dna_pad|> token (row-major over [B, P]).
// Output: tok_src[t] = flat row (seq * Sd + pos) feeding token t, or -1 for an ordinary token;
// counts = {number of placeholder tokens, number of DNA feature rows} for the reference's
// mismatch ValueError (dna_llm.py:222-225).
__global__ __launch_bounds__(1024) void dna_scatter_plan_kernel(const int* ids, int ntok, int dna_id,
                                                                const uint8_t* dna_mask, int nseq, int Sd,
                                                                const int* seq_order, int* tok_src, int* counts) {
    __shared__ int feat_base[1025];      // exclusive prefix over seq_order (nseq <= 1024)
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // valid length per sequence (one wave per sequence, strided)
    for (int k = wave; k < nseq; k += 16) {
        const int seq = seq_order[k];
        int cnt = 0;
        for (int p = lane; p < Sd; p += 64) cnt += dna_mask[(long)seq * Sd + p] ? 1 : 0;
        for (int m = 32; m >= 1; m >>= 1) cnt += wave_shfl_xor_i(cnt, m);
        if (lane == 0) feat_base[k + 1] = cnt;
    }
    if (tid == 0) { feat_base[0] = 0; carry = 0; }
    __syncthreads();
    if (tid == 0) for (int k = 0; k < nseq; ++k) feat_base[k + 1] += feat_base[k];
    __syncthreads();
    const int nfeat = feat_base[nseq];
    for (int t0 = 0; t0 < ntok; t0 += 1024) {
        const int t = t0 + tid;
        const bool flag = t < ntok && ids[t] == dna_id;
        const uint64_t bal = wave_ballot(flag);
        const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __builtin_popcountll(bal);
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (t < ntok) {
            int src = -1;
            if (flag) {
                const int rank = off + before;
                if (rank < nfeat) {
                    int lo = 0, hi = nseq;   // largest k with feat_base[k] <= rank
                    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (feat_base[mid] <= rank) lo = mid; else hi = mid; }
                    src = seq_order[lo] * Sd + (rank - feat_base[lo]);
                } else {
                    src = -2;
                }
            }
            tok_src[t] = src;
        }
        __syncthreads();
        if (tid == 0) { int tot = 0; for (int w = 0; w < 16; ++w) tot += wsum[w]; carry += tot; }
        __syncthreads();
    }
    if (tid == 0) { counts[0] = carry; counts[1] = nfeat; }
}

// out[t,:] = tok_src[t] >= 0 ? dna_rows[tok_src[t],:] : E[ids[t],:]     (dna_llm.py:211,229)
__global__ __launch_bounds__(256) void embed_scatter_fwd_kernel(const int* ids, const int* tok_src, const bf16_t* E,
                                                                long lde, const bf16_t* dna, long ldd, bf16_t* out,
                                                                long ldo, int ntok, int H) {
    const int cpr = H / 8;
    const long total = (long)ntok * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i / cpr), c = (int)(i % cpr) * 8;
        const int src = tok_src ? tok_src[t] : -1;
        u32x4 v;
        if (src >= 0) v = ld16(dna + (long)src * ldd + c);
        else v = ld16(E + (long)ids[t] * lde + c);
        st16(out + (long)t * ldo + c, v);
    }
}

// d(dna_rows)[tok_src[t],:] = dout[t,:]; rows of dna_rows not referenced stay zero (caller zero-fills)
__global__ __launch_bounds__(256) void embed_scatter_bwd_kernel(const int* tok_src, const bf16_t* dout, long ldo,
                                                                bf16_t* ddna, long ldd, int ntok, int H) {
    const int cpr = H / 8;
    const long total = (long)ntok * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i / cpr), c = (int)(i % cpr) * 8;
        const int src = tok_src[t];
        if (src >= 0) st16(ddna + (long)src * ldd + c, ld16(dout + (long)t * ldo + c));
    }
}

// gather rows: out[i,:] = x[rows[i],:]
__global__ __launch_bounds__(256) void gather_rows_kernel(const int* rows, const bf16_t* x, long ldx, bf16_t* out,
                                                          long ldo, int n, int H) {
    const int cpr = H / 8;
    const long total = (long)n * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / cpr), c = (int)(i % cpr) * 8;
        st16(out + (long)r * ldo + c, ld16(x + (long)rows[r] * ldx + c));
    }
}
// scatter rows: out[rows[i],:] = x[i,:]   (rows unique; caller zero-fills out)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const int* rows, const bf16_t* x, long ldx, bf16_t* out,
                                                           long ldo, int n, int H) {
    const int cpr = H / 8;
    const long total = (long)n * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / cpr), c = (int)(i % cpr) * 8;
        st16(out + (long)rows[r] * ldo + c, ld16(x + (long)r * ldx + c));
    }
}

// ---------------------------------------------------------------------------
// 2-D transpose of a bf16 matrix: out[c, r] = in[r, c]  (activation / gradient images for the
// weight-gradient contractions, which run as NT GEMMs over the token dimension)
__global__ __launch_bounds__(256) void transpose2d_kernel(const bf16_t* in, long ldi, bf16_t* out, long ldo, int rows,
                                                          int cols) {
    __shared__ bf16_t tile[64 * 66];
    const int tid = (int)threadIdx.x;
    const int r0 = (int)blockIdx.y * 64, c0 = (int)blockIdx.x * 64;
    for (int q = tid; q < 64 * 8; q += 256) {
        const int r = q >> 3, c = (q & 7) * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < rows) {
            if (c0 + c + 7 < cols) v = ld16(in + (long)(r0 + r) * ldi + c0 + c);
            else {
                uint32_t w[4] = {0, 0, 0, 0};
                for (int i = 0; i < 8; ++i)
                    if (c0 + c + i < cols) w[i >> 1] |= (uint32_t)in[(long)(r0 + r) * ldi + c0 + c + i] << (16 * (i & 1));
                v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
            }
        }
        uint32_t* tp = reinterpret_cast<uint32_t*>(&tile[r * 66 + c]);
        tp[0] = v.x; tp[1] = v.y; tp[2] = v.z; tp[3] = v.w;
    }
    __syncthreads();
    for (int q = tid; q < 64 * 8; q += 256) {
        const int c = q >> 3, rr = (q & 7) * 8;      // output row = input column c; 8 consecutive input rows
        if (c0 + c >= cols) continue;
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = tile[(rr + 2 * i) * 66 + c], hi = tile[(rr + 2 * i + 1) * 66 + c];
            w[i] = lo | (hi << 16);
        }
        bf16_t* op = out + (long)(c0 + c) * ldo + r0 + rr;
        if (r0 + rr + 7 < rows) { u32x4 v = {w[0], w[1], w[2], w[3]}; st16(op, v); }
        else for (int i = 0; i < 8; ++i) if (r0 + rr + i < rows) op[i] = (bf16_t)(w[i >> 1] >> (16 * (i & 1)));
    }
}

// column sums: out[n] (+)= sum_m x[m, n]   (bias gradient of dna_projection, dna_llm.py:97)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* x, long ldx, float* out, int rows, int cols,
                                                     int rows_per_block) {
    const int c = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int r0 = (int)blockIdx.y * rows_per_block;
    int r1 = r0 + rows_per_block;
    r1 = r1 < rows ? r1 : rows;
    if (c >= cols) return;
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) acc += bf2f(x[(long)r * ldx + c]);
    atomicAdd(out + c, acc);
}

// ---------------------------------------------------------------------------
// pack table: fp32 master parameters -> bf16 working images (optionally transposed)
struct PackDesc {
    const float* src; long src_ld;
    bf16_t* dst; long dst_ld;
    int rows, cols, transpose, pad;
};
__global__ __launch_bounds__(256) void pack_params_kernel(const PackDesc* descs) {
    const PackDesc d = descs[blockIdx.y];
    const long total = (long)d.rows * d.cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / d.cols), c = (int)(i % d.cols);
        const bf16_t v = f2bf(d.src[(long)r * d.src_ld + c]);
        if (d.transpose) d.dst[(long)c * d.dst_ld + r] = v;
        else d.dst[(long)r * d.dst_ld + c] = v;
    }
}

// ---------------------------------------------------------------------------
// merge the per-chunk (max, sum-exp) partials of the fused lm_head into the row log-sum-exp and
// the log-probability of the target token (grpo_trainer.py:510-520; TF:loss/loss_utils.py:49-71)
__global__ __launch_bounds__(256) void lse_merge_kernel(const float* part_max, const float* part_sum,
                                                        const float* tgt_logit, float* lse, float* logp, int rows,
                                                        int nchunk) {
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int lane = lane_id();
    const bool live = row < rows;
    const long base = (long)(live ? row : 0) * nchunk;
    float m = -3.0e38f;
    for (int c = lane; c < nchunk; c += 64) m = fmaxf(m, part_max[base + c]);
    m = wave_max<64>(m);
    float s = 0.f;
    for (int c = lane; c < nchunk; c += 64) s += part_sum[base + c] * __expf(part_max[base + c] - m);
    s = wave_sum<64>(s);
    if (live && lane == 0) {
        const float l = m + __logf(s);
        lse[row] = l;
        if (logp) logp[row] = tgt_logit[row] - l;
    }
}

// fp32 <-> bf16 images of a flat gradient range: the optional bf16 transport of the data-parallel all-reduce (half the xGMI bytes of
// the 148 MB fp32 arena; SURVEY section 8e "148 MB fp32 (or 74 MB bf16) bucket").  dir 0: dst(bf16) = round(src(f32)); 1: dst(f32) = src(bf16)
__global__ __launch_bounds__(256) void cast_grad_kernel(const void* src, void* dst, long n, int dir) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (dir == 0) ((bf16_t*)dst)[i] = f2bf(((const float*)src)[i]);
    else ((float*)dst)[i] = bf2f(((const bf16_t*)src)[i]);
}

// out[0] = scale * sum(x[0..n)) — the mean reduction of the cross-entropy / per-token losses
__global__ __launch_bounds__(256) void vec_sum_kernel(const float* x, long n, float scale, float* out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) acc += x[i];
    acc = block_sum<4>(acc, red);
    if (threadIdx.x == 0) out[0] = acc * scale;
}

// ---------------------------------------------------------------------------
// AdamW over one flat fp32 arena (train_dna_qwen.py:393-411; ds_config_stage2.json:5-21), with the
// global-norm clip (max_grad_norm) read from a device scalar so no host sync is needed.
// `mask` (bytes, optional): 0 marks structural zeros of the packed LoRA layout (off-diagonal blocks of a fused
// B matrix, padding rows) — excluded from the norm and never updated.
// Two fixed-order stages (no atomics): data-parallel replicas must compute bit-identical clip factors from their
// bit-identical all-reduced gradients, or they drift apart.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, const uint8_t* mask, long n, float* ws) {
    __shared__ float part[4];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        if (!mask || mask[i]) acc += g[i] * g[i];
    acc = wave_sum<64>(acc);
    if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* ws, int nblk, float* out) {
    __shared__ float part[4];
    float acc = 0.f;
    for (int i = (int)threadIdx.x; i < nblk; i += 256) acc += ws[i];
    acc = wave_sum<64>(acc);
    if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v,
                                                    const uint8_t* mask, long n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2, const float* sumsq,
                                                    float max_norm, float grad_scale) {
    float clip = grad_scale;
    if (sumsq && max_norm > 0.f) {
        const float nrm = sqrtf(sumsq[0]) * grad_scale;
        const float c = max_norm / (nrm + 1e-6f);
        if (c < 1.f) clip *= c;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (mask && !mask[i]) continue;
        const float gi = g[i] * clip;
        float pi = p[i];
        pi *= (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] = pi - (lr / bc1) * mi / denom;
    }
}

}  // namespace bra

using namespace bra;

static inline int ew_grid(long n) {
    long g = (n + 255) / 256;
#ifdef BRA_EMU
    if (g > 32) g = 32;                  // (host executor: one fiber per thread — the kernels are grid-stride loops, a small grid is the same work)
#endif
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int bra_head_transpose(const void* x, long sb, long ss, long sh, void* xt, long t_sb, long t_sh,
                                  long pitch, int B, int S, int H, int hd, void* stream) {
    if (B <= 0 || S <= 0 || H <= 0) return 0;
    if (!x || !xt || pitch % 64 || pitch < ((S + 63) / 64) * 64) return BRA_ERR_ARG;
    dim3 grid((S + 63) / 64, H, B);
    if (hd == 128) BRA_LAUNCH((head_transpose_kernel<128>), grid, dim3(256), 0, stream, (const bf16_t*)x, sb, ss, sh, (bf16_t*)xt, t_sb, t_sh, pitch, S);
    else if (hd == 64) BRA_LAUNCH((head_transpose_kernel<64>), grid, dim3(256), 0, stream, (const bf16_t*)x, sb, ss, sh, (bf16_t*)xt, t_sb, t_sh, pitch, S);
    else if (hd == 32) BRA_LAUNCH((head_transpose_kernel<32>), grid, dim3(256), 0, stream, (const bf16_t*)x, sb, ss, sh, (bf16_t*)xt, t_sb, t_sh, pitch, S);
    else return BRA_ERR_UNSUPPORTED;
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_attn_delta(const void* dout, long do_sb, long do_ss, long do_sh, const void* o, long o_sb,
                              long o_ss, long o_sh, float* delta, int B, int S, int H, int hd, void* stream) {
    if (B <= 0 || S <= 0 || H <= 0) return 0;
    if (!dout || !o || !delta || (hd != 32 && hd != 64 && hd != 128)) return BRA_ERR_ARG;
    const long total = (long)B * H * S * (hd / 8);
    BRA_LAUNCH(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)dout,
               do_sb, do_ss, do_sh, (const bf16_t*)o, o_sb, o_ss, o_sh, delta, B, S, H, hd);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_dna_scatter_plan(const int* ids, int ntok, int dna_id, const void* dna_mask, int nseq, int Sd,
                                    const int* seq_order, int* tok_src, int* counts, void* stream) {
    if (!ids || !tok_src || !counts || ntok < 0 || nseq < 0 || nseq > 1024) return BRA_ERR_ARG;
    if (nseq > 0 && (!dna_mask || !seq_order || Sd <= 0)) return BRA_ERR_ARG;
    BRA_LAUNCH(dna_scatter_plan_kernel, dim3(1), dim3(1024), 0, stream, ids, ntok, dna_id, (const uint8_t*)dna_mask,
               nseq, Sd, seq_order, tok_src, counts);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_embed_scatter_fwd(const int* ids, const int* tok_src, const void* E, long lde, const void* dna,
                                     long ldd, void* out, long ldo, int ntok, int H, void* stream) {
    if (ntok == 0) return 0;
    if (!ids || !E || !out || H % 8 || lde % 8 || ldo % 8 || (tok_src && (!dna || ldd % 8))) return BRA_ERR_ARG;
    BRA_LAUNCH(embed_scatter_fwd_kernel, dim3(ew_grid((long)ntok * (H / 8))), dim3(256), 0, stream, ids, tok_src,
               (const bf16_t*)E, lde, (const bf16_t*)dna, ldd, (bf16_t*)out, ldo, ntok, H);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_embed_scatter_bwd(const int* tok_src, const void* dout, long ldo, void* ddna, long ldd, int ntok,
                                     int H, void* stream) {
    if (ntok == 0) return 0;
    if (!tok_src || !dout || !ddna || H % 8 || ldo % 8 || ldd % 8) return BRA_ERR_ARG;
    BRA_LAUNCH(embed_scatter_bwd_kernel, dim3(ew_grid((long)ntok * (H / 8))), dim3(256), 0, stream, tok_src,
               (const bf16_t*)dout, ldo, (bf16_t*)ddna, ldd, ntok, H);
    return BRA_LAUNCH_STATUS();
}

// out[r, i] = (add ? add[r, i] : 0) + sum_c src[(r * copies + c) * member_stride + i],  i < n (multiple of 8), fp32 accumulation,
// one rounding — the gradient that the `copies` sequences of a group send to the rows they SHARE (the prompt K / V of GRPO's G
// rollouts, grpo_trainer.py:107-116: the shared prompt is run once, the upstream gradients of its rows are the sum over the copies)
namespace bra {
__global__ __launch_bounds__(256) void group_sum_kernel(const bf16_t* src, long member_stride, int copies, const bf16_t* add, long add_stride,
                                                         bf16_t* out, long out_stride, int R, long n8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)R * n8) return;
    const int r = (int)(i / n8);
    const long c8 = i - (long)r * n8;
    float acc[8];
    if (add) unpack8(ld16(add + (long)r * add_stride + c8 * 8), acc);
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    }
    for (int c = 0; c < copies; ++c) {
        float f[8];
        unpack8(ld16(src + ((long)r * copies + c) * member_stride + c8 * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
    st16(out + (long)r * out_stride + c8 * 8, pack8(acc));
}
}  // namespace bra

extern "C" int bra_group_sum(const void* src, long member_stride, int copies, const void* add, long add_stride, void* out,
                             long out_stride, int R, long n, void* stream) {
    if (R == 0 || n == 0) return 0;
    if (!src || !out || copies <= 0 || n % 8 || member_stride % 8 || out_stride % 8 || (add && add_stride % 8)) return BRA_ERR_ARG;
    const long items = (long)R * (n / 8);
    BRA_LAUNCH(bra::group_sum_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, (const bra::bf16_t*)src, member_stride,
               copies, (const bra::bf16_t*)add, add_stride, (bra::bf16_t*)out, out_stride, R, n / 8);
    return BRA_LAUNCH_STATUS();
}

// out block ((r copies + c) inner + h) = src block (r inner + h) for every copy c: the K / V rows of a shared prompt placed in front
// of each rollout's own rows (one 16-byte load, `copies` 16-byte stores per thread)
namespace bra {
__global__ __launch_bounds__(256) void group_broadcast_kernel(const bf16_t* src, long src_blk_stride, bf16_t* out, long out_blk_stride,
                                                               int copies, int inner, long nblk, long n8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nblk * n8) return;
    const long blk = i / n8;
    const long c8 = i - blk * n8;
    const long r = blk / inner;
    const long h = blk - r * inner;
    const u32x4 v = ld16(src + blk * src_blk_stride + c8 * 8);
    for (int c = 0; c < copies; ++c) st16(out + ((r * copies + c) * inner + h) * out_blk_stride + c8 * 8, v);
}
}  // namespace bra

extern "C" int bra_group_broadcast(const void* src, long src_blk_stride, void* out, long out_blk_stride, int R, int copies, int inner,
                                   long n, void* stream) {
    if (R == 0 || n == 0 || inner == 0) return 0;
    if (!src || !out || copies <= 0 || inner < 0 || n % 8 || src_blk_stride % 8 || out_blk_stride % 8) return BRA_ERR_ARG;
    const long nblk = (long)R * inner, items = nblk * (n / 8);
    BRA_LAUNCH(bra::group_broadcast_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, (const bra::bf16_t*)src,
               src_blk_stride, (bra::bf16_t*)out, out_blk_stride, copies, inner, nblk, n / 8);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_gather_rows(const int* rows, const void* x, long ldx, void* out, long ldo, int n, int H,
                               void* stream) {
    if (n == 0) return 0;
    if (!rows || !x || !out || H % 8 || ldx % 8 || ldo % 8) return BRA_ERR_ARG;
    BRA_LAUNCH(gather_rows_kernel, dim3(ew_grid((long)n * (H / 8))), dim3(256), 0, stream, rows, (const bf16_t*)x,
               ldx, (bf16_t*)out, ldo, n, H);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_scatter_rows(const int* rows, const void* x, long ldx, void* out, long ldo, int n, int H,
                                void* stream) {
    if (n == 0) return 0;
    if (!rows || !x || !out || H % 8 || ldx % 8 || ldo % 8) return BRA_ERR_ARG;
    BRA_LAUNCH(scatter_rows_kernel, dim3(ew_grid((long)n * (H / 8))), dim3(256), 0, stream, rows, (const bf16_t*)x,
               ldx, (bf16_t*)out, ldo, n, H);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_transpose2d(const void* in, long ldi, void* out, long ldo, int rows, int cols, void* stream) {
    if (rows == 0 || cols == 0) return 0;
    if (!in || !out || ldi % 8 || ldo % 8) return BRA_ERR_ARG;
    BRA_LAUNCH(transpose2d_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, stream,
               (const bf16_t*)in, ldi, (bf16_t*)out, ldo, rows, cols);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_colsum(const void* x, long ldx, float* out, int rows, int cols, void* stream) {
    if (rows == 0 || cols == 0) return 0;
    if (!x || !out) return BRA_ERR_ARG;
    const int rpb = 256;
    BRA_LAUNCH(colsum_kernel, dim3((cols + 255) / 256, (rows + rpb - 1) / rpb), dim3(256), 0, stream,
               (const bf16_t*)x, ldx, out, rows, cols, rpb);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_pack_desc_size(void) { return (int)sizeof(PackDesc); }

extern "C" int bra_pack_params(const void* descs_dev, int ndesc, long max_elems, void* stream) {
    if (ndesc == 0) return 0;
    if (!descs_dev) return BRA_ERR_ARG;
    BRA_LAUNCH(pack_params_kernel, dim3(ew_grid(max_elems), ndesc), dim3(256), 0, stream,
               (const PackDesc*)descs_dev);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_lse_merge(const float* part_max, const float* part_sum, const float* tgt_logit, float* lse,
                             float* logp, int rows, int nchunk, void* stream) {
    if (rows == 0) return 0;
    if (!part_max || !part_sum || !lse || nchunk <= 0 || (logp && !tgt_logit)) return BRA_ERR_ARG;
    BRA_LAUNCH(lse_merge_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, part_max, part_sum, tgt_logit, lse,
               logp, rows, nchunk);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_cast_grad(const void* src, void* dst, long n, int to_f32, void* stream) {
    if (n == 0) return 0;
    if (!src || !dst || n < 0) return BRA_ERR_ARG;
    BRA_LAUNCH(cast_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, n, to_f32 ? 1 : 0);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_vec_sum(const float* x, long n, float scale, float* out, void* stream) {
    if (!x || !out || n <= 0) return BRA_ERR_ARG;
    BRA_LAUNCH(vec_sum_kernel, dim3(1), dim3(256), 0, stream, x, n, scale, out);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_sumsq(const float* g, const void* mask, long n, float* out, float* ws, void* stream) {
    if (!out || !ws) return BRA_ERR_ARG;
    if (n > 0 && !g) return BRA_ERR_ARG;
    int nblk = n > 0 ? ew_grid(n) : 1;
    nblk = nblk < 1024 ? nblk : 1024;                    // ws holds 1024 floats
    BRA_LAUNCH(sumsq_kernel, dim3(nblk), dim3(256), 0, stream, g, (const uint8_t*)mask, n, ws);
    BRA_LAUNCH(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, ws, nblk, out);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_adamw(float* p, const float* g, float* m, float* v, const void* mask, long n, float lr, float b1,
                         float b2, float eps, float wd, int step, const float* sumsq, float max_norm,
                         float grad_scale, void* stream) {
    if (n == 0) return 0;
    if (!p || !g || !m || !v || step < 1) return BRA_ERR_ARG;
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    BRA_LAUNCH(adamw_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, p, g, m, v, (const uint8_t*)mask, n, lr, b1, b2, eps, wd, bc1, bc2,
               sumsq, max_norm, grad_scale);
    return BRA_LAUNCH_STATUS();
}
