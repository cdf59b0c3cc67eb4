/* bioreason_hip.h — C ABI of libbioreason_hip.so (gfx950 / MI355X).
 *
 * The reference (bowang-lab/BioReason) has no FFI: its hot path is executed by
 * third-party Python (HF transformers Qwen3 / ESM modelling, torch SDPA, PEFT)
 * underneath bioreason/models/dna_llm.py and bioreason/trainer/grpo_trainer.py.
 * This header is the boundary introduced UNDER those modules: every entry
 * point names the reference arithmetic it replaces ("TF:" = the installed
 * transformers package, the reference's unpinned dependency).
 *
 * Conventions
 *   - plain C, POD arguments only: raw device pointers, sizes, element strides
 *     ("ld*" = leading dimension in ELEMENTS), `stream` = a hipStream_t.
 *   - bf16 tensors are raw 16-bit words; "f32" tensors are float.
 *   - every function returns 0 on success, a positive hipError_t if the launch
 *     failed, or a negative BRA_ERR_* for a rejected argument; nothing throws,
 *     nothing allocates, nothing synchronises; workspaces come from the caller.
 *   - all entry points are re-entrant and thread-safe.  There is NO process-wide mutable state in this library except
 *     an init-once "dynamic LDS opted in" flag per kernel and device and an init-once device-properties cache.  Which
 *     tile variant a GEMM call runs is a pure function of its arguments.  The knobs that pin a variant for tests and
 *     A/B measurements, the timing probes and the experimental persistent decode step are NOT part of this ABI: they
 *     exist only in libbioreason_hip_debug.so (built with -DBRA_DEBUG on request, `make debug`) and are declared in
 *     bioreason_hip_debug.h.
 */
#ifndef BIOREASON_HIP_H
#define BIOREASON_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define BRA_ERR_ARG (-1)
#define BRA_ERR_UNSUPPORTED (-2)

/* ---- GEMM family (k_gemm.hip) ------------------------------------------------
 * C[M,N] = alpha * (A[M,K] B[N,K]^T + A2[M,K2] B2[N,K2]^T) [+ bias[N]] [+ res[M,N]]
 * Replaces nn.Linear forward / dgrad everywhere on the path: Qwen3 q/k/v/o and MLP
 * (TF:models/qwen3/modeling_qwen3.py:225-236, :81-83), ESM q/k/v/o/FFN
 * (TF:models/esm/modeling_esm.py:336-338, :402, :448-466), dna_projection
 * (bioreason/models/dna_llm.py:97,159-160), tied lm_head (TF:qwen3:495); the
 * (A2,B2,K2) pair is PEFT-LoRA's  + (x A^T) B^T  (train_dna_qwen.py:155-167,
 * reason.py:376-388).  K, K2 multiples of 32; ld* multiples of 8.
 * out_f32: C is float (optionally accumulated into), else bf16.
 * With `res`, the product is rounded to bf16 before the add, as the reference's
 * `residual + module(x)` does (TF:qwen3:309,315). */
int bra_gemm_bf16_nt(const void* A, long lda, const void* B, long ldb, const void* A2, long lda2, const void* B2,
                     long ldb2, int K2, void* C, long ldc, int M, int N, int K, float alpha, const void* bias,
                     const void* res, long ldres, int out_f32, int accumulate, void* stream);

/* C[M,N] (f32, pre-zeroed or holding a running gradient) += alpha * A[M,K] B[N,K]^T with the K range cut
 * into `split_k` slices and combined by atomics: weight gradients of LoRA A/B and dna_projection, where
 * K is the token count (autograd of PEFT's lora_A / lora_B, and of dna_llm.py:159-160). */
int bra_gemm_bf16_nt_splitk(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                            float alpha, int split_k, void* stream);

/* Low-rank weight gradient from ROW-MAJOR operands (k_wgrad.hip): C[n, r] += alpha * sum_m Y[m, n] T[m, r], Y [M, N]
 * bf16, T [M, R] bf16 with R in {32, 64, 128}; C fp32 addressed by element strides (c_sn, c_sr), so the same call
 * writes dB [N_out, r] (Y = dy, T = s x A^T) and dA [r, K] transposed (Y = x, T = s dy B) — autograd of PEFT's
 * lora_B / lora_A.  m_chunk = rows per workgroup (0: chosen to fill the chip); combined by fp32 atomics. */
int bra_wgrad_tn(const void* Y, long ldy, const void* T, long ldt, float* C, long c_sn, long c_sr, int M, int N, int R,
                 float alpha, int m_chunk, void* stream);

/* LoRA branch under training-mode dropout (k_lora.hip; PEFT: y += (alpha/r) B(A(dropout(x))), lora_dropout 0.05,
 * reason.py:266,376-388).  Every 32-column rank block (= target module of a fused projection) has its own mask stream
 * s0..s3, as PEFT gives every target its own nn.Dropout; keep(seed, m, k) is a hash of the element index m*K + k, so no
 * mask is stored.  R in {32, 64, 128}; M*K < 2^32.
 *   bra_lora_down_drop: t[M,R] = alpha * (drop_j(x) A^T)              x [M,K], A [R,K]
 *   bra_lora_up_drop:   out[M,K] = sum_j drop_j'( dts[:, j] A[j, :] )  dts [M,R], AT [K,R]   (input gradient of the branch)
 *   bra_wgrad_tn_drop:  bra_wgrad_tn with Y = drop_rb(Y)               (dA of the branch)
 *   bra_dropout_mask:   out[M,K] bytes = keep(seed, m, k)              (tests: inject the same masks into the oracle)
 * nb_live = rank blocks that belong to a target module (<= R / 32; 0 = all): a fused projection pads its rank to 64 / 128
 * columns, and the padding blocks (zero rows of A, zero columns of dts) are skipped instead of masked. */
int bra_lora_down_drop(const void* x, long ldx, const void* A, long lda, void* t, long ldt, int M, int K, int R, float alpha,
                       float p, unsigned s0, unsigned s1, unsigned s2, unsigned s3, int nb_live, void* stream);
/* Split-K form of bra_lora_down_drop for small M (M / 32 row-block workgroups cannot fill 256 CUs and each would read all of A):
 * `ksplit` workgroups per row block take K / ksplit each, fp32 partial tiles go to `part` ([ksplit, M, R] floats, caller-provided)
 * and a second launch sums them in a FIXED order (deterministic) and scales / rounds once.  bra_lora_down_splitk_plan(M, K) returns
 * the split the host layer uses (1 = use the plain form). */
int bra_lora_down_splitk_plan(int M, int K);
int bra_lora_down_drop_splitk(const void* x, long ldx, const void* A, long lda, void* t, long ldt, int M, int K, int R, float alpha,
                              float p, unsigned s0, unsigned s1, unsigned s2, unsigned s3, int nb_live, float* part, int ksplit,
                              void* stream);
int bra_lora_up_drop(const void* dts, long ldd, const void* AT, long ldat, void* out, long ldo, int M, int K, int R, float p,
                     unsigned s0, unsigned s1, unsigned s2, unsigned s3, int nb_live, void* stream);
int bra_wgrad_tn_drop(const void* Y, long ldy, const void* T, long ldt, float* C, long c_sn, long c_sr, int M, int N, int R,
                      float alpha, int m_chunk, float p, unsigned s0, unsigned s1, unsigned s2, unsigned s3, int nb_live,
                      void* stream);
int bra_dropout_mask(void* out, int M, int K, float p, unsigned seed, void* stream);

/* Fused lm_head + log-softmax statistics WITHOUT materialising logits
 * (grpo_trainer.py:510-520 `_get_per_token_logps`; TF:loss/loss_utils.py:49-71 ForCausalLMLoss):
 * for every row m of H[M,K] (bf16) against E[V,K]: per 64-column chunk running max and sum-exp of the
 * bf16-rounded logits, and the logit of column tgt[m].  nchunk = ceil(V/64). */
int bra_lmhead_lse_partials(const void* H, long ldh, const void* E, long lde, int M, int V, int K, const int* tgt,
                            float* part_max, float* part_sum, float* tgt_logit, void* stream);
/* merge of the partials: lse[m], logp[m] = tgt_logit[m] - lse[m] */
int bra_lse_merge(const float* part_max, const float* part_sum, const float* tgt_logit, float* lse, float* logp,
                  int rows, int nchunk, void* stream);
/* backward of the above: dlogits[m,n] = coef[m] * ((n == tgt[m]) - exp(logit[m,n] - lse[m]))  (bf16) */
int bra_lmhead_dlogits(const void* H, long ldh, const void* E, long lde, int M, int V, int K, const int* tgt,
                       const float* lse, const float* coef, void* dlogits, long ldd, void* stream);

/* ---- normalisation / activation (k_norm.hip) --------------------------------- */
/* Qwen3RMSNorm.forward (TF:qwen3:59-64); rstd (f32 [rows]) optional */
int bra_rmsnorm_fwd(const void* x, long ldx, const void* w, void* y, long ldy, float* rstd, int rows, int cols,
                    float eps, void* stream);
/* its input gradient (+ optional residual-stream gradient dres); norm weights are frozen under LoRA */
int bra_rmsnorm_bwd(const void* dy, long lddy, const void* x, long ldx, const void* w, const void* dres, long lddres,
                    void* dx, long lddx, int rows, int cols, float eps, void* stream);
/* nn.LayerNorm forward of the ESM encoder (TF:esm:418,480,529) */
int bra_layernorm_fwd(const void* x, long ldx, const void* w, const void* b, void* y, long ldy, int rows, int cols,
                      float eps, void* stream);
/* SwiGLU: act = silu(gu[:, :F]) * gu[:, F:]  (Qwen3MLP TF:qwen3:81-83; NT-v2 hub EsmIntermediate) */
int bra_swiglu_fwd(const void* gu, long ldgu, void* act, long ldact, int rows, int F, void* stream);
int bra_swiglu_bwd(const void* gu, long ldgu, const void* dact, long lddact, void* dgu, long lddgu, int rows, int F,
                   void* stream);
/* per-head q_norm/k_norm (optional) + rotate-half RoPE on a fused QKV projection
 * (Qwen3Attention.forward TF:qwen3:252-256 + apply_rotary_pos_emb :140-170;
 *  EsmSelfAttention.forward TF:esm:366-378 with qscale = hd^-0.5 and no norm).
 * q/k/v destinations are addressed by (batch, seq, head) element strides; K/V rows are written at
 * sequence index s + s_off (KV-cache append). */
int bra_qk_norm_rope_fwd(const void* qkv, long ldqkv, const void* qw, const void* kw, const float* cosT,
                         const float* sinT, const int* pos, int T, int S, int Hq, int Hkv, int hd, float eps,
                         float qscale, void* q, long q_sb, long q_ss, long q_sh, void* k, long k_sb, long k_ss,
                         long k_sh, void* v, long v_sb, long v_ss, long v_sh, int s_off, void* stream);
int bra_qk_norm_rope_bwd(const void* qkv, long ldqkv, const void* qw, const void* kw, const float* cosT,
                         const float* sinT, const int* pos, int T, int S, int Hq, int Hkv, int hd, float eps,
                         float qscale, const void* dq, long q_sb, long q_ss, long q_sh, const void* dk, long k_sb,
                         long k_ss, long k_sh, const void* dv, long v_sb, long v_ss, long v_sh, void* dqkv,
                         long lddqkv, void* stream);

/* ---- attention (k_attn.hip) --------------------------------------------------
 * softmax(scale * Q K^T + mask) V with fp32 softmax: Qwen3 causal GQA (TF:qwen3:185-207, hd 128) and the
 * ESM bidirectional encoder (TF:esm:292-317, hd 64).  kmask[B,Sk] (bytes, 1 = attendable) is the
 * reference's 0/1 attention_mask; causal: key j visible to query i iff j <= i + q_off.
 * vt / kt / qt / dot are [B,H,hd,pitch] transposed images (bra_head_transpose), pitch % 64 == 0. */
int bra_attn_fwd(const void* q, long q_sb, long q_ss, long q_sh, const void* k, long k_sb, long k_ss, long k_sh,
                 const void* vt, long vt_sb, long vt_sh, long vt_sd, void* o, long o_sb, long o_ss, long o_sh,
                 float* lse, const void* kmask, int B, int Hq, int Hkv, int Sq, int Sk, int hd, int causal,
                 int q_off, float scale, void* stream);
/* bra_attn_fwd with every query block's key range cut into `nsplit` (2..8) parts that run as separate workgroups + one merge launch
 * (grids that cannot fill the 256 CUs: one prompt, a 256-query completion segment).  part_o fp32 [B, Hq, nsplit, Sq, hd] and part_ml
 * fp32 [B, Hq, nsplit, Sq, 2] are caller-owned workspaces; needs Sq > 128, hd >= 64. */
int bra_attn_fwd_split(const void* q, long q_sb, long q_ss, long q_sh, const void* k, long k_sb, long k_ss, long k_sh,
                       const void* vt, long vt_sb, long vt_sh, long vt_sd, void* o, long o_sb, long o_ss, long o_sh,
                       float* lse, const void* kmask, int B, int Hq, int Hkv, int Sq, int Sk, int hd, int causal,
                       int q_off, float scale, int nsplit, float* part_o, float* part_ml, void* stream);
int bra_attn_bwd(const void* q, long q_sb, long q_ss, long q_sh, const void* k, long k_sb, long k_ss, long k_sh,
                 const void* v, long v_sb, long v_ss, long v_sh, const void* dout, long do_sb, long do_ss, long do_sh,
                 const void* kt, long kt_sb, long kt_sh, long kt_sd, const void* qt, long qt_sb, long qt_sh,
                 long qt_sd, const void* dot, long dot_sb, long dot_sh, long dot_sd, const float* lse,
                 const float* delta, const void* kmask, void* dq, long dq_sb, long dq_ss, long dq_sh, void* dk,
                 long dk_sb, long dk_ss, long dk_sh, void* dv, long dv_sb, long dv_ss, long dv_sh, int B, int Hq,
                 int Hkv, int Sq, int Sk, int hd, int causal, int q_off, float scale, void* stream);
/* bra_attn_bwd for grids that cannot fill the 256 CUs: the dQ kernel's key range in `nsplit_dq` parts (part_dq fp32 [B, Hq, nsplit_dq,
 * Sq, hd]), the one-launch dK + dV kernel's (q-head, query tile) loop in `nsplit_kv` parts (part_dk / part_dv fp32 [B, Hkv, nsplit_kv,
 * Sk, hd]); the parts are added in order by a sum launch each; 1 = that kernel is not split. */
int bra_attn_bwd_split(const void* q, long q_sb, long q_ss, long q_sh, const void* k, long k_sb, long k_ss, long k_sh,
                 const void* v, long v_sb, long v_ss, long v_sh, const void* dout, long do_sb, long do_ss, long do_sh,
                 const void* kt, long kt_sb, long kt_sh, long kt_sd, const void* qt, long qt_sb, long qt_sh,
                 long qt_sd, const void* dot, long dot_sb, long dot_sh, long dot_sd, const float* lse,
                 const float* delta, const void* kmask, void* dq, long dq_sb, long dq_ss, long dq_sh, void* dk,
                 long dk_sb, long dk_ss, long dk_sh, void* dv, long dv_sb, long dv_ss, long dv_sh, int B, int Hq,
                 int Hkv, int Sq, int Sk, int hd, int causal, int q_off, float scale, int nsplit_dq, float* part_dq, int nsplit_kv, float* part_dk, float* part_dv, void* stream);
/* one decode step over the KV cache [B,Hkv,Smax,hd] (HF DynamicCache + sdpa, TF:generation/utils.py:2876-2925).
 * part_o f32 [B,Hq,nchunk,hd], part_ml f32 [B,Hq,nchunk,2], nchunk = bra_attn_decode_nchunk(len). */
int bra_attn_decode_nchunk(int len);
int bra_attn_decode(const void* q, const void* kc, const void* vc, const void* kmask, float* part_o, float* part_ml,
                    void* o, int B, int Hq, int Hkv, int hd, int Smax, int len, float scale, void* stream);

/* ---- native decode step (k_decode.hip) -------------------------------------------
 * One token per sequence through all L Qwen3 layers (the body of HF's `_sample` loop,
 * TF:generation/utils.py:2876-2925 -> TF:qwen3:367-427 with a KV cache), enqueued by native code so the
 * ~400 launches of a step are not paced by the Python interpreter.  `layers_host` is a HOST array of L
 * records {ln1, ln2, qn, kn, Wqkv, Wo, Wgu, Wd, A_qkv, B_qkv, A_o, B_o, A_gu, B_gu, A_d, B_d (device ptrs, LoRA
 * ones may be null); int r_qkv, r_o, r_gu, r_d; float s_qkv, s_o, s_gu, s_d; kc, vc; kp, vtp; int flags, pad; record 0:
 * head_packed, rope_rows, head_tmax (float [B, ceil(V / 16)] or null: the step functions that write logits also leave the
 * maximum of every 16-column tile there for bra_sample_tiles)}, bra_qwen_layer_desc_size() bytes each; the rest are device
 * pointers to caller-owned workspaces.  Writes the final-normed hidden rows [B, H] to `hid`. */
int bra_qwen_layer_desc_size(void);
int bra_qwen_decode_step(const void* layers_host, int L, int B, int H, int Hq, int Hkv, int hd, int F, int Smax,
                         float eps, float scale, const void* E, const void* norm_w, const float* cosT,
                         const float* sinT, const int* tok, const int* pos, const void* kmask, int cur_len,
                         int lora_on, void* x, void* xn, void* qkv, void* q, void* o, void* h, void* hn, void* gu,
                         void* act, void* t, float* part_o, float* part_ml, void* hid, void* stream);

/* Fused decode kernels (k_decfused.hip) and the 6-launch-per-layer step built from them.
 * bra_dec_gemm: out[M<=16, N] = rmsnorm(x; norm_w, eps) W^T (+res); act=1: W rows are [8 gate | 8 up] blocks and the
 *   output is silu(gate)*up [M, N/2] (Qwen3MLP TF:qwen3:81-83); out_f32: fp32 logits (tied lm_head TF:qwen3:495).
 * bra_dec_attn_partial: per-head q/k RMSNorm + RoPE of the new token (TF:qwen3:252-256), append of its K/V row at
 *   index cur_len, chunked attention over [0, cur_len]; bra_attn_decode_merge combines the chunk partials.
 * bra_qwen_decode_step_fused: layer records carry rollout weights (LoRA merged as PEFT merge_and_unload does,
 *   reason.py:428-446; gate/up row-interleaved); writes fp32 logits [B, V] if `logits` is non-null. */
int bra_dec_gemm(const void* x, long ldx, const void* norm_w, float eps, const void* W, long ldw, const void* res,
                 long ldres, void* out, long ldo, int M, int N, int K, int act, int out_f32, void* stream);
int bra_dec_attn_partial(const void* qkv, long ldqkv, const void* qw, const void* kw, const float* cosT,
                         const float* sinT, const int* pos, void* kc, void* vc, const void* kmask, float* part_o,
                         float* part_ml, int B, int Hq, int Hkv, int hd, int Smax, int cur_len, float eps, float scale,
                         int chunk_off, int nchunk_tot, const int* t_dev, void* stream);
/* Second-generation streaming projection (k_decgemm.hip): same arithmetic as bra_dec_gemm for M <= 8 rows, but the
 * RMSNorm statistics arrive as partial sums of squares ss_in[8][nss_in] (left by the producer's epilogue through
 * ss_out[8][nss_out], column = workgroup, or by bra_row_sumsq) instead of a per-workgroup prologue pass over x;
 * nss multiples of 32, unused columns zero.  `ss_ws` of the step functions = float [2][8][nss] zero-initialised,
 * nss >= H/8 (null: first-generation kernels). */
int bra_dec_gemm2(const void* x, long ldx, const float* ss_in, int nss_in, const void* norm_w, float eps, const void* W,
                  long ldw, const void* res, long ldres, void* out, long ldo, float* ss_out, int nss_out, int M, int N,
                  int K, int act, int out_f32, void* stream);
/* bra_dec_gemm2 with `packed` weights (bit 0: fragment order, bit 1: norm weight folded; see bra_dec_pack_weights) */
int bra_dec_gemm2_packed(const void* x, long ldx, const float* ss_in, int nss_in, const void* norm_w, float eps, const void* W,
                         long ldw, const void* res, long ldres, void* out, long ldo, float* ss_out, int nss_out, int M, int N,
                         int K, int act, int out_f32, int packed, void* stream);
/* fp8 (OCP e4m3) rollout weights — BASELINE config 5 ("GRPO fp8 weights"), opt-in: half the bytes of the token loop's weight stream.
 * bra_dec_pack_weights_fp8: W [N, K] bf16 (norm_w [K] optional: the input's RMSNorm weight multiplied in first) -> out_q, N * K bytes
 * in bra_dec_gemm2_fp8's fragment order, + out_scale [N] fp32, scale[n] = max_k |W[n,k] w[k]| / 448, q = e4m3(W w / scale) rounded to
 * nearest even.  bra_dec_gemm2_fp8: y = [rstd *] scale[n] * (x q^T) with bra_dec_gemm2_packed's epilogues (residual | SwiGLU | fp32
 * logits + tile maxima); the fp8 values are decoded to bf16 in registers (exact) in front of the bf16 MFMAs, activations stay bf16,
 * accumulation fp32 (W8A16: the step is HBM-bound, the fp8 MFMA rate is not what it needs).  M <= 8, single-register-round shapes
 * (every projection of Qwen3-1.7B); BRA_ERR_UNSUPPORTED otherwise — the caller keeps bf16 weights for that projection.
 * Replaces the same nn.Linear forwards as bra_dec_gemm2 (TF:qwen3:225-236, :81-83, :495) under weight-only quantisation. */
int bra_dec_pack_weights_fp8(const void* W, long ldw, int N, int K, int act, int out_f32, const void* norm_w, void* out_q,
                             float* out_scale, void* stream);
int bra_dec_gemm2_fp8(const void* x, long ldx, const float* ss_in, int nss_in, float eps, const void* Wq, const float* wscale,
                      const void* res, long ldres, void* out, long ldo, float* ss_out, int nss_out, int M, int N, int K, int act,
                      int out_f32, int norm_folded, void* stream);
/* gate/up projection with the SwiGLU in the GEMM epilogue (round 6): act [M, F] = bra_swiglu_fwd(bra_gemm_bf16_nt(A, W [2 F, K] = [gate | up]
 * rows (+ the LoRA rank part A2 B2^T))), bit for bit, without the [M, 2 F] intermediate — no-grad passes only (the backward of a training
 * forward needs gate and up).  BRA_ERR_UNSUPPORTED unless K % 64 == 0, K2 % 64 == 0, F % 128 == 0, M > 16 (callers fall back to the two
 * launches).  Replaces TF:qwen3:81-83 / the NT-v2 FFN in the encoder, the reference pass and the prompt pass. */
int bra_gemm_swiglu_bf16_nt(const void* A, long lda, const void* W, long ldw, const void* A2, long lda2, const void* B2, long ldb2, int K2,
                            void* C, long ldc, int M, int F, int K, float alpha, void* stream);
/* ---- fp8 x fp8 GEMM on the fp8 MFMA path (round 6; BASELINE config 5 "fp8 weights (CDNA4 fp8 MFMA)"; k_gemm.hip gemm_fp8_kernel,
 * k_quant.hip) — the prompt pass and the no-grad reference pass of an fp8 rollout run: the behaviour policy's MFMA-bound passes on
 * v_mfma_scale_f32_16x16x128_f8f6f4 (OCP e4m3 operands, fp32 accumulation, block scales 2^0), per-row fp32 scales in the epilogue.
 * bra_quant_rows_fp8: x [M, K] bf16 -> q [M, K] e4m3 (ldq bytes) + scale [M]: a = max_k |x colw| / 448 (1 for a zero row),
 *   q = e4m3(x colw / a) round-to-nearest-even, saturating; colw (optional, bf16 [K]) is multiplied in first — with W and the input's
 *   RMSNorm weight this is bra_dec_pack_weights_fp8's rule, so the token loop's image of the weight holds the same bytes in another
 *   order.  rms != 0: scale = a * rsqrt(mean_k x^2 + eps): the row factor of a folded-norm projection, y = rstd (x (W w)^T).
 * bra_swiglu_quant_fp8: [gate | up] rows -> the e4m3 image + scale of act = bf16(bf16(silu(gate)) up) (bra_swiglu_fwd's roundings).
 * bra_gemm_fp8_nt: C [M, N] (bf16, or fp32 with out_f32) = sa[m] sb[n] (A8 B8^T) (+ res bf16); K % 128 == 0, lda / ldb bytes % 16.
 * Replaces the nn.Linear forwards of TF:qwen3:225-236, :81-83 under W8A8 per-row / per-token quantisation (opt-in; the bf16 path
 * never calls these). */
int bra_quant_rows_fp8(const void* x, long ldx, int M, int K, const void* colw, void* q, long ldq, float* scale, int rms, float eps,
                       void* stream);
int bra_swiglu_quant_fp8(const void* gu, long ldgu, int M, int F, void* q, long ldq, float* scale, void* stream);
int bra_gemm_fp8_nt(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, void* C, long ldc, int M, int N,
                    int K, const void* res, long ldres, int out_f32, void* stream);
/* W [N, K] -> the fragment order bra_dec_gemm2 streams with `packed` = 1 (same act / out_f32 flags as the projection it feeds:
 * they select the tile mode): every wave-instruction of the weight stream then reads one contiguous KiB of full 128-byte lines
 * instead of 16 row segments of 64 bytes.  out: N * K bf16.  BRA_ERR_UNSUPPORTED when N, K are not tile multiples.
 * norm_w (optional, [K]): the RMSNorm weight of the projection's INPUT is multiplied in (W[n,k] * w[k], rounded once); such a
 * copy is streamed with packed = 3: y = rstd_row * (x W'^T), the row factor applied to the reduced products — no wave loads
 * norm weights or statistics ahead of the MFMAs (TF:qwen3:59-64 rounds the normalised activation to bf16 instead: the two
 * differ by bf16 rounding placement only; rollout path). */
int bra_dec_pack_weights(const void* W, long ldw, int N, int K, int act, int out_f32, const void* norm_w, void* out, void* stream);
/* the same for a given number of batch rows: bra_dec_gemm2 takes up to 16 rows (two prompts x 8 rollouts per GPU); above 8 rows
 * every projection streams 16-column tiles (the 8-column "diagonal" tiles of the N = hidden projections hold 8 rows), the
 * RMSNorm must be folded (packed = 3) and the statistics arrays have 16 rows */
int bra_dec_pack_weights_rows(const void* W, long ldw, int N, int K, int act, int out_f32, const void* norm_w, int rows, void* out,
                              void* stream);
int bra_row_sumsq(const void* x, long ldx, int M, int K, float* ss, int nss, void* stream);
/* `t_dev` / `len_dev` (optional device int): when given, the attention kernels read the current length from it and
 * the host-side `cur_len` / `t` only size the grids (pass the maximum); the launch arguments are then identical for
 * every step, so one captured hipGraph replays the whole rollout (HF `_sample` loop, TF:generation/utils.py:2876-2925). */
/* shared-prefix decode attention: the `copies` sequences of each of R prompts (GRPO's G rollouts, grpo_trainer.py:107-116)
 * attend to ONE copy of the prompt K / V^T; writes chunk partials 0 .. ceil(P/64)-1 of every (sequence, q-head) */
int bra_dec_attn_shared(const void* qkv, long ldqkv, const void* qw, const float* cosT, const float* sinT, const int* pos,
                        const void* kp, long kp_sr, long kp_sh, long kp_ss, const void* vtp, long vt_sr, long vt_sh,
                        long vt_sd, const void* pmask, float* part_o, float* part_ml, int R, int copies, int Hq, int Hkv,
                        int hd, int P, int nchunk_tot, float eps, float scale, const int* t_dev, void* stream);
/* bra_dec_attn_shared + bra_dec_attn_partial (on the completion caches kc / vc [B, Hkv, C, hd], index t) in one launch */
int bra_dec_attn_both(const void* qkv, long ldqkv, const void* qw, const void* kw, const float* cosT, const float* sinT,
                      const int* pos, const void* kp, long kp_sr, long kp_sh, long kp_ss, const void* vtp, long vt_sr,
                      long vt_sh, long vt_sd, const void* pmask, void* kc, void* vc, float* part_o, float* part_ml, int R,
                      int copies, int Hq, int Hkv, int hd, int P, int C, int t, float eps, float scale, const int* t_dev,
                      const float* rope_rows, void* stream);
int bra_attn_decode_merge(const float* part_o, const float* part_ml, void* o, int B, int Hq, int hd, int nchunk,
                          const int* t_dev, int npc, void* stream);
int bra_qwen_decode_step_fused(const void* layers_host, int L, int B, int H, int Hq, int Hkv, int hd, int F, int Smax,
                               int V, float eps, float scale, const void* E, const void* norm_w, const float* cosT,
                               const float* sinT, const int* tok, const int* pos, const void* kmask, int cur_len,
                               const int* len_dev, int embed_done, void* x, void* qkv, void* o, void* h, void* act, float* ss_ws,
                               int nss, float* part_o, float* part_ml, float* logits, void* stream);

/* shared-prefix step: B = R * copies sequences grouped by prompt; layer records additionally carry kp (prompt K
 * [R,Hkv,P,hd]) and vtp (prompt V^T [R,Hkv,hd,vt_pitch]); kc / vc are the per-sequence COMPLETION caches [B,Hkv,C,hd]
 * and t the number of completion tokens already cached.  6 launches per layer (qkv, attention, merge, o, gate/up + SwiGLU, down). */
int bra_qwen_decode_step_shared(const void* layers_host, int L, int R, int copies, int H, int Hq, int Hkv, int hd, int F,
                                int P, long vt_pitch, int C, int V, float eps, float scale, const void* E,
                                const void* norm_w, const float* cosT, const float* sinT, const int* tok, const int* pos,
                                const void* pmask, int t, const int* t_dev, int embed_done, void* x, void* qkv, void* o, void* h,
                                void* act, float* ss_ws, int nss, float* part_o, float* part_ml, float* logits, void* stream);

/* Shared-prefix decode attention, second generation (k_decattn.hip; same operator as bra_dec_attn_both + bra_attn_decode_merge:
 * Qwen3Attention.forward TF:qwen3:231-284 on one new token per sequence with a KV cache, TF:generation/utils.py:2876-2925).
 * Items kernel: one-wave items score 64-key chunks of the shared prompt K / V^T and of each sequence's completion cache on
 * MFMA (the completion V cache is kept TRANSPOSED: vct [B, Hkv, hd, cp], cp >= ceil64(C), zeroed by the caller at allocation;
 * kc [B, Hkv, C, hd]; t = completion tokens already cached), one more item per (sequence, q-head) appends the new k / v and
 * emits the new key's own partial; merge kernel: one wave per (sequence, q-head).  o bf16 [B, Hq * hd].
 * nslot = partial slots per (sequence, q-head) >= ceil(P/64) + ceil(C/64) + 1, <= 256. */
int bra_dec_attn_one(const void* qkv, long ldqkv, const void* qw, const void* kw, const float* cosT, const float* sinT,
                     const int* pos, const float* rope_rows, const void* kp, long kp_sr, long kp_sh, long kp_ss,
                     const void* vtp, long vt_sr, long vt_sh, long vt_sd, const void* pmask, void* kc, void* vct, long cp,
                     float* part_o, float* part_ml, int nslot, void* o, long ldo, int R, int copies,
                     int Hq, int Hkv, int hd, int P, int C, int t, float eps, float scale, const int* t_dev, void* stream);
/* bra_qwen_decode_step_shared on bra_dec_attn_one.  The layer records' vc fields hold the transposed completion V caches. */
int bra_qwen_decode_step_one(const void* layers_host, int L, int R, int copies, int H, int Hq, int Hkv, int hd, int F,
                             int P, long vt_pitch, int C, long cp, int V, float eps, float scale, const void* E,
                             const void* norm_w, const float* cosT, const float* sinT, const int* tok, const int* pos,
                             const void* pmask, int t, const int* t_dev, int embed_done, void* x, void* qkv, void* o, void* h,
                             void* act, float* ss_ws, int nss, float* part_o, float* part_ml, int nslot, float* logits,
                             void* stream);

/* ---- data movement around the kernels (k_misc.hip) ---------------------------- */
int bra_head_transpose(const void* x, long sb, long ss, long sh, void* xt, long t_sb, long t_sh, long pitch, int B,
                       int S, int H, int hd, void* stream);
int bra_attn_delta(const void* dout, long do_sb, long do_ss, long do_sh, const void* o, long o_sb, long o_ss,
                   long o_sh, float* delta, int B, int S, int H, int hd, void* stream);
/* DNALLMModel.process_dna_embeddings regrouping + `text_inputs_embeds[mask] = dna_embeds_flat`
 * (dna_llm.py:163-177, :216-229) as a device-side plan: tok_src[t] = projected-DNA row feeding token t or -1;
 * counts = {#placeholder tokens, #DNA feature rows} for the reference's mismatch ValueError (:222-225). */
int bra_dna_scatter_plan(const int* ids, int ntok, int dna_id, const void* dna_mask, int nseq, int Sd,
                         const int* seq_order, int* tok_src, int* counts, void* stream);
/* embed_tokens(input_ids) with the DNA rows written over the placeholders (dna_llm.py:211,229) and its backward */
int bra_embed_scatter_fwd(const int* ids, const int* tok_src, const void* E, long lde, const void* dna, long ldd,
                          void* out, long ldo, int ntok, int H, void* stream);
int bra_embed_scatter_bwd(const int* tok_src, const void* dout, long ldo, void* ddna, long ldd, int ntok, int H,
                          void* stream);
int bra_gather_rows(const int* rows, const void* x, long ldx, void* out, long ldo, int n, int H, void* stream);
int bra_scatter_rows(const int* rows, const void* x, long ldx, void* out, long ldo, int n, int H, void* stream);
int bra_transpose2d(const void* in, long ldi, void* out, long ldo, int rows, int cols, void* stream);
int bra_colsum(const void* x, long ldx, float* out, int rows, int cols, void* stream);
/* fp32 master -> bf16 working images of the trainable parameters; descs_dev = device array of
 * {const float* src; long src_ld; bf16* dst; long dst_ld; int rows, cols, transpose, pad;} */
int bra_pack_desc_size(void);
int bra_pack_params(const void* descs_dev, int ndesc, long max_elems, void* stream);
/* AdamW over one flat arena with device-side global-norm clip (train_dna_qwen.py:393-411, :1003 clip 1.0) */
/* out[0] = scale * sum(x): the `mean` of F.cross_entropy (TF:loss/loss_utils.py:32-46) */
int bra_vec_sum(const float* x, long n, float scale, float* out, void* stream);
/* dst = bf16(src fp32) (to_f32 = 0) or dst = fp32(src bf16) (to_f32 = 1) over n elements: the optional bf16 transport of the gradient
 * all-reduce (DDP / ZeRO-2 reduce of ~37 M trainable parameters, train_dna_qwen.py:985-1005, ds_config_stage2.json:22-34) */
int bra_cast_grad(const void* src, void* dst, long n, int to_f32, void* stream);
/* out[0] = sum of squares of g (global-norm clip, max_grad_norm); ws = float[1024] scratch.  Fixed summation order,
 * no atomics: replicas holding identical gradients get identical norms.
 * mask (bytes, optional): 0 = structural zero of the packed LoRA layout, excluded from norm and update */
int bra_sumsq(const float* g, const void* mask, long n, float* out, float* ws, void* stream);
int bra_adamw(float* p, const float* g, float* m, float* v, const void* mask, long n, float lr, float b1, float b2,
              float eps, float wd, int step, const float* sumsq, float max_norm, float grad_scale, void* stream);

/* ---- GRPO arithmetic (k_grpo.hip) --------------------------------------------- */
/* temperature -> top-k -> top-p -> multinomial, or argmax when do_sample == 0
 * (TF:generation/logits_process.py:238,473,542; TF:generation/utils.py:2897-2925; grpo_trainer.py:384-391) */
/* finished[B] (bytes, in/out): a finished row emits pad_id; a row that emits eos_id becomes finished
 * (HF: next = next*unfinished + pad*(1-unfinished); unfinished &= next != eos).  tokens_out[row*ldt + *step_ptr]
 * receives the token (the [B, C] completion matrix), step_ptr is a device int so a replayed launch advances.
 * eos_id2: a second stop token (HF accepts a list; Qwen3's generation_config has two), -1 = none.
 * do_sample with top_k outside 1..64 is BRA_ERR_UNSUPPORTED: the temperature / top-p / multinomial stage runs over at
 * most 64 survivors (HF's top_k = 0 "disabled" would need a full-vocabulary sort). */
int bra_sample(const float* logits, long ldl, int B, int V, float temperature, int top_k, float top_p,
               int do_sample, unsigned seed, const int* step_ptr, void* finished, int pad_id, int eos_id, int eos_id2,
               int* out_ids, float* out_logp, int* tokens_out, long ldt, void* ws, void* stream);
/* ws (optional, bra_sample_ws_floats(B, top_k) 4-byte words): enables the two-stage top-k (64 vocabulary slices
 * per row in parallel, then a merge) instead of one workgroup per row scanning the vocabulary k times */
int bra_sample_ws_floats(int B, int top_k);
/* bra_sample on the two-stage path whose drawing wave also gathers x[b] = E[token] (HF embed_tokens of the next step,
 * TF:qwen3:380) and its RMSNorm statistic ss[8][nss] (bra_row_sumsq layout): the decode step functions then start at
 * the first projection (`embed_done`).  E may be null (sampling only).  BRA_ERR_UNSUPPORTED without ws / outside
 * 4096 <= V <= 262144. */
int bra_sample_embed(const float* logits, long ldl, int B, int V, float temperature, int top_k, float top_p, int do_sample,
                     unsigned seed, const int* step_ptr, void* finished, int pad_id, int eos_id, int eos_id2, int* out_ids,
                     float* out_logp, int* tokens_out, long ldt, void* ws, const void* E, long lde, int H, void* x, long ldx,
                     float* ss, int nss, void* stream);
/* synthetic EOS schedule for benchmarks / tests with random-init weights (SURVEY 8d "straggler run"): logits[b, token]
 * is raised above every other entry when *step_ptr == at[b], so row b draws `token` at that step.  B <= 64. */
int bra_force_token(float* logits, long ldl, int B, int V, int token, const int* step_ptr, const int* at, void* stream);
/* ---- sampler over tile maxima (round 4; k_grpo.hip: sample_tiles_kernel) -------
 * tmax [B, ldm >= ceil(V / 16)] fp32: the maximum of every 16-column tile of the logits — left by the lm_head epilogue of the
 * decode step functions (layer record 0: `head_tmax`) or computed by bra_tile_max.  Every one of the k best logits of a row
 * lies in one of the k best tiles, so the top-k stage scans V / 16 maxima + 16 k logits instead of V logits; tokens, ties and
 * draws are those of bra_sample / bra_sample_embed on the same inputs (TF:generation/logits_process.py:238,473,542). */
int bra_tile_max(const float* logits, long ldl, int B, int V, float* tmax, long ldm, void* stream);
/* bra_sample_embed over (logits, tmax) in two launches.  `step_ptr` null: the step index is the launch argument `step` (a
 * token loop issued launch by launch needs no device-side counter).  pos0 / pos_out / cosT / sinT / hd / rope_rows (optional):
 * pos_out[b] = pos0[b] + step, rope_rows[b] = (cos | sin) row of that position — the state bra_advance_counters would have
 * left for the decode step that follows this draw (HF: cache_position += 1, TF:generation/utils.py:979-984). */
int bra_sample_tiles(const float* logits, long ldl, const float* tmax, long ldm, int B, int V, float temperature, int top_k,
                     float top_p, int do_sample, unsigned seed, const int* step_ptr, int step, void* finished, int pad_id,
                     int eos_id, int eos_id2, int* out_ids, float* out_logp, int* tokens_out, long ldt, void* ws,
                     const void* E, long lde, int H, void* x, long ldx, float* ss, int nss, const int* pos0, int* pos_out,
                     const float* cosT, const float* sinT, int hd, float* rope_rows, void* stream);
/* sampling with top_k = 0 (HF: top-k disabled) or top_k > 64: thresholds of TopKLogitsWarper / TopPLogitsWarper found by bisection over
 * the row's logits, multinomial over the survivors (one workgroup per row; exact, ~0.1-0.3 ms per token: a functional path — GRPO's
 * top_k = 20 takes bra_sample_tiles).  Ties of equal logits at the top-p boundary are kept or dropped together. */
int bra_sample_full(const float* logits, long ldl, int B, int V, float temperature, int top_k, float top_p, unsigned seed,
                    const int* step_ptr, int step, void* finished, int pad_id, int eos_id, int eos_id2, int* out_ids,
                    float* out_logp, int* tokens_out, long ldt, void* stream);
/* bra_force_token that also raises the token's tile maximum; `step_ptr` null: the step index is `step` */
int bra_force_token_tiles(float* logits, long ldl, int B, int V, int token, const int* step_ptr, int step, const int* at,
                          float* tmax, long ldm, void* stream);
/* counters of the replayed token loop: pos[0..n) += 1, a[0] += 1, b[0] += 1 (a, b optional); when `rope_rows` is given, also
 * rope_rows[i][0..hd/2) = cosT[pos[i]], [hd/2..hd) = sinT[pos[i]] for the NEW positions (see bra_rope_rows) */
int bra_advance_counters(int* pos, int n, int* a, int* b, const float* cosT, const float* sinT, int hd, float* rope_rows,
                         void* stream);
/* rope_rows [n, hd] fp32 = (cos | sin) table rows of the positions pos[0..n): the decode attention kernels read the row of
 * their sequence directly instead of hopping pos -> table (one dependent memory round trip per layer less); optional last
 * pointer argument of bra_dec_attn_both */
int bra_rope_rows(const float* cosT, const float* sinT, const int* pos, int n, int hd, float* rope_rows, void* stream);
/* completion mask up to and including the first EOS (grpo_trainer.py:605-609) */
int bra_eos_mask(const int* ids, int B, int C, int eos_id, int* mask, int* lengths, void* stream);
/* rewards [N,F] -> sum over F -> (r - mean_group) / (std_group + 1e-4), groups of G (grpo_trainer.py:682-691) */
int bra_group_advantage(const float* rewards, int N, int F, int G, float* adv, float* grp_mean, float* grp_std,
                        void* stream);
/* compute_loss (grpo_trainer.py:786-814): out3 = {loss, mean_kl, clip_ratio}; dlogp = dloss/dlogp */
int bra_grpo_loss(const float* logp, const float* old_logp, const float* ref_logp, const float* adv, const int* mask,
                  int B, int C, float eps_lo, float eps_hi, float beta, float* out3, float* dlogp, void* stream);

/* out[r, 0..n) = (add ? add[r] : 0) + sum over the `copies` members c of group r of src[(r copies + c) member_stride + 0..n), bf16
 * in / out, fp32 accumulation: the gradient the G rollouts of a GRPO group (grpo_trainer.py:107-116) send to the rows they share
 * when the prompt is run once for the group (the shared prompt's K / V rows; the last prompt row's hidden state). */
int bra_group_sum(const void* src, long member_stride, int copies, const void* add, long add_stride, void* out, long out_stride,
                  int R, long n, void* stream);

/* The inverse placement: block (r inner + h) of src (n contiguous bf16 elements, blocks src_blk_stride apart) is written to blocks
 * ((r copies + c) inner + h) of out (out_blk_stride apart) for c in [0, copies): a shared prompt's K / V rows [R, Hkv, P, hd] put in
 * front of every rollout's own rows in a [R copies, Hkv, P + C, hd] cache (the completion rows of the shared-prompt passes attend to
 * [prompt | own]; replaces `expand` + copy of the reference's batched layout, grpo_trainer.py:107-116 / TF:qwen3:185-207). */
int bra_group_broadcast(const void* src, long src_blk_stride, void* out, long out_blk_stride, int R, int copies, int inner, long n,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BIOREASON_HIP_H */
