// k_decode.hip — host-side orchestration of ONE Qwen3 decode step (all layers) in native code.
// The per-token loop of HF generate (TF:generation/utils.py:2876-2925 -> Qwen3Model.forward TF:qwen3:367-427
// with a DynamicCache) is ~400 small launches; issuing them from Python costs more than they take to run, so
// the step is one C-ABI call that enqueues every kernel on the caller's stream.  No device code here.
#include "bra_device.h"
#include "bra_api_internal.h"
#include "../../include/bioreason_hip.h"
#ifdef BRA_DEBUG
#include "../../include/bioreason_hip_debug.h"
#endif

namespace {

struct Layer {
    const void *ln1, *ln2, *qn, *kn, *Wqkv, *Wo, *Wgu, *Wd;
    const void *A_qkv, *B_qkv, *A_o, *B_o, *A_gu, *B_gu, *A_d, *B_d;   // LoRA images (null = no adapter)
    int r_qkv, r_o, r_gu, r_d;                                         // padded ranks
    float s_qkv, s_o, s_gu, s_d;                                       // alpha / r
    void *kc, *vc;                                                     // KV cache [B, Hkv, Smax, hd]
    const void *kp, *vtp;                                              // shared prompt K [R,Hkv,P,hd] and V^T [R,Hkv,hd,pitch]
    int flags;                                                         // bit 0: Wqkv / Wo / Wgu / Wd are fragment-packed (bra_dec_pack_weights); bit 1: ln1 / ln2 folded
                                                                       // into Wqkv / Wgu; bit 2 (record 0): final norm folded into head_packed
    int pad_;
    const void* head_packed;                                           // record 0 only: fragment-packed lm_head matrix, or null
    const float* rope_rows;                                            // record 0 only: [B, hd] cos | sin rows of the current positions, or null
    float* head_tmax;                                                  // record 0 only: [B, ceil(V / 16)] maxima of the logits' 16-column tiles (sampler), or null
    // fp8 rollout weights (bra_dec_pack_weights_fp8; flags bit 3: Wqkv / Wo / Wgu / Wd are e4m3 images, bit 4 (record 0): head_packed is):
    // one fp32 scale per output row of each projection
    const float *sc_qkv, *sc_o, *sc_gu, *sc_d, *sc_head;
};

}  // namespace

extern "C" int bra_qwen_layer_desc_size(void) { return (int)sizeof(Layer); }

// x, xn, h, hn: [B,H]; qkv: [B,Nq+2Nkv]; q, o: [B,Nq]; gu: [B,2F]; act: [B,F]; t: [B,128]; hid: [B,H]
extern "C" int bra_qwen_decode_step(const void* layers_host, int L, int B, int H, int Hq, int Hkv, int hd, int F,
                                    int Smax, float eps, float scale, const void* E, const void* norm_w,
                                    const float* cosT, const float* sinT, const int* tok, const int* pos,
                                    const void* kmask, int cur_len, int lora_on, void* x, void* xn, void* qkv, void* q,
                                    void* o, void* h, void* hn, void* gu, void* act, void* t, float* part_o,
                                    float* part_ml, void* hid, void* stream) {
    const Layer* ls = (const Layer*)layers_host;
    const int Nq = Hq * hd, Nkv = Hkv * hd, Nqkv = Nq + 2 * Nkv;
    int rc;
#define CK(call) do { rc = (call); if (rc) return rc; } while (0)
    CK(bra_embed_scatter_fwd(tok, nullptr, E, H, nullptr, 0, x, H, B, H, stream));
    // lin: y = in W^T (+ LoRA) (+res)
    auto lin = [&](const void* in, int K, const void* W, int N, const void* A, const void* Bm, int r, float s,
                   const void* res, void* out) -> int {
        if (A && lora_on) {
            int e = bra_gemm_bf16_nt(in, K, A, K, nullptr, 0, nullptr, 0, 0, t, r, B, r, K, s, nullptr, nullptr, 0, 0, 0, stream);
            if (e) return e;
            return bra_gemm_bf16_nt(in, K, W, K, t, r, Bm, r, r, out, N, B, N, K, 1.f, nullptr, res, N, 0, 0, stream);
        }
        return bra_gemm_bf16_nt(in, K, W, K, nullptr, 0, nullptr, 0, 0, out, N, B, N, K, 1.f, nullptr, res, N, 0, 0, stream);
    };
    for (int li = 0; li < L; ++li) {
        const Layer& l = ls[li];
        CK(bra_rmsnorm_fwd(x, H, l.ln1, xn, H, nullptr, B, H, eps, stream));
        CK(lin(xn, H, l.Wqkv, Nqkv, l.A_qkv, l.B_qkv, l.r_qkv, l.s_qkv, nullptr, qkv));
        // q -> [B,1,Hq,hd]; k,v appended to the cache at sequence index cur_len
        CK(bra_qk_norm_rope_fwd(qkv, Nqkv, l.qn, l.kn, cosT, sinT, pos, B, 1, Hq, Hkv, hd, eps, 1.f, q, (long)Nq, (long)Nq,
                                (long)hd, l.kc, (long)Hkv * Smax * hd, (long)hd, (long)Smax * hd, l.vc, (long)Hkv * Smax * hd,
                                (long)hd, (long)Smax * hd, cur_len, stream));
        CK(bra_attn_decode(q, l.kc, l.vc, kmask, part_o, part_ml, o, B, Hq, Hkv, hd, Smax, cur_len + 1, scale, stream));
        CK(lin(o, Nq, l.Wo, H, l.A_o, l.B_o, l.r_o, l.s_o, x, h));
        CK(bra_rmsnorm_fwd(h, H, l.ln2, hn, H, nullptr, B, H, eps, stream));
        CK(lin(hn, H, l.Wgu, 2 * F, l.A_gu, l.B_gu, l.r_gu, l.s_gu, nullptr, gu));
        CK(bra_swiglu_fwd(gu, 2 * F, act, F, B, F, stream));
        CK(lin(act, F, l.Wd, H, l.A_d, l.B_d, l.r_d, l.s_d, h, x));
    }
    CK(bra_rmsnorm_fwd(x, H, norm_w, hid, H, nullptr, B, H, eps, stream));
#undef CK
    return 0;
}

// The four projections of a decoder layer at decode time, in either generation of the streaming GEMM: `ss` != null selects
// bra_dec_gemm2 (k_decgemm.hip), whose RMSNorm statistics travel as partial sums of squares from epilogue to consumer.
struct StepGemms {
    bool v2; float* ssx; float* ssh; int nss;
    int B, H, Nq, Nqkv, F, V; float eps; void* stream;
};
static StepGemms step_gemms(float* ss_ws, int nss, int B, int H, int Nq, int Nqkv, int F, int V, float eps, void* stream) {
    StepGemms s;
    const int nblk = (H % 8 == 0 && (H + 15) / 16 < 256 && H / 8 <= 256) ? H / 8 : (H + 15) / 16;      // workgroups of an N = H projection (bra_dec_gemm2's tile rule)
    s.v2 = ss_ws && B <= 16 && nss >= 32 && nss % 32 == 0 && nss <= 256 && nss >= nblk;
    // two statistics arrays [rows][nss], rows = 8 or (9 .. 16 sequences) 16
    s.ssx = ss_ws; s.ssh = ss_ws ? ss_ws + (B > 8 ? 16 : 8) * (long)nss : nullptr; s.nss = nss;
    s.B = B; s.H = H; s.Nq = Nq; s.Nqkv = Nqkv; s.F = F; s.V = V; s.eps = eps; s.stream = stream;
    return s;
}
static int sg_begin(const StepGemms& s, const void* x) {
    return s.v2 ? bra_row_sumsq(x, s.H, s.B, s.H, s.ssx, s.nss, s.stream) : 0;
}
static int sg_qkv(const StepGemms& s, const Layer& l, const void* x, void* qkv) {
    const int pk = l.flags & 3;
    if (s.v2 && (l.flags & 8))
        return bra_dec_gemm2_fp8(x, s.H, s.ssx, s.nss, s.eps, l.Wqkv, l.sc_qkv, nullptr, 0, qkv, s.Nqkv, nullptr, 0, s.B, s.Nqkv, s.H, 0, 0, 1, s.stream);
    if (s.v2) return bra_dec_gemm2_packed(x, s.H, s.ssx, s.nss, l.ln1, s.eps, l.Wqkv, s.H, nullptr, 0, qkv, s.Nqkv, nullptr, 0, s.B, s.Nqkv, s.H, 0, 0, pk, s.stream);
    if (pk) return BRA_ERR_UNSUPPORTED;
    return bra_dec_gemm(x, s.H, l.ln1, s.eps, l.Wqkv, s.H, nullptr, 0, qkv, s.Nqkv, s.B, s.Nqkv, s.H, 0, 0, s.stream);
}
// h = x + o Wo^T;  act = swiglu(rmsnorm(h) Wgu^T);  x = h + act Wd^T
static int sg_tail(const StepGemms& s, const Layer& l, const void* o, void* x, void* h, void* act) {
    int rc;
    const int pk = l.flags & 1;
    if (s.v2 && (l.flags & 8)) {
        if ((rc = bra_dec_gemm2_fp8(o, s.Nq, nullptr, 0, 0.f, l.Wo, l.sc_o, x, s.H, h, s.H, s.ssh, s.nss, s.B, s.H, s.Nq, 0, 0, 0, s.stream))) return rc;
        if ((rc = bra_dec_gemm2_fp8(h, s.H, s.ssh, s.nss, s.eps, l.Wgu, l.sc_gu, nullptr, 0, act, s.F, nullptr, 0, s.B, 2 * s.F, s.H, 1, 0, 1, s.stream))) return rc;
        return bra_dec_gemm2_fp8(act, s.F, nullptr, 0, 0.f, l.Wd, l.sc_d, h, s.H, x, s.H, s.ssx, s.nss, s.B, s.H, s.F, 0, 0, 0, s.stream);
    }
    if (s.v2) {
        if ((rc = bra_dec_gemm2_packed(o, s.Nq, nullptr, 0, nullptr, 0.f, l.Wo, s.Nq, x, s.H, h, s.H, s.ssh, s.nss, s.B, s.H, s.Nq, 0, 0, pk, s.stream))) return rc;
        if ((rc = bra_dec_gemm2_packed(h, s.H, s.ssh, s.nss, l.ln2, s.eps, l.Wgu, s.H, nullptr, 0, act, s.F, nullptr, 0, s.B, 2 * s.F, s.H, 1, 0, l.flags & 3, s.stream))) return rc;
        return bra_dec_gemm2_packed(act, s.F, nullptr, 0, nullptr, 0.f, l.Wd, s.F, h, s.H, x, s.H, s.ssx, s.nss, s.B, s.H, s.F, 0, 0, pk, s.stream);
    }
    if (pk) return BRA_ERR_UNSUPPORTED;
    if ((rc = bra_dec_gemm(o, s.Nq, nullptr, 0.f, l.Wo, s.Nq, x, s.H, h, s.H, s.B, s.H, s.Nq, 0, 0, s.stream))) return rc;
    if ((rc = bra_dec_gemm(h, s.H, l.ln2, s.eps, l.Wgu, s.H, nullptr, 0, act, s.F, s.B, 2 * s.F, s.H, 1, 0, s.stream))) return rc;
    return bra_dec_gemm(act, s.F, nullptr, 0.f, l.Wd, s.F, h, s.H, x, s.H, s.B, s.H, s.F, 0, 0, s.stream);
}
// `tmax` (optional, [B, ceil(V / 16)]): the projection's epilogue also leaves the maximum of every 16-column tile of the logits
// (bra_sample_tiles); the first-generation kernel has no such epilogue — bra_tile_max then computes them in a launch of its own
static int sg_head(const StepGemms& s, const void* x, const void* norm_w, const void* E, const void* Epacked, int folded, float* logits,
                   float* tmax, const float* sc_head = nullptr) {
    const int nt = (s.V + 15) / 16;
    if (s.v2 && Epacked && sc_head)      // fp8 lm_head image (final norm folded)
        return bra_dec_gemm2_fp8(x, s.H, s.ssx, s.nss, s.eps, Epacked, sc_head, nullptr, 0, logits, s.V, tmax, tmax ? nt : 0, s.B, s.V, s.H, 0, 1, 1, s.stream);
    if (s.v2 && Epacked)
        return bra_dec_gemm2_packed(x, s.H, s.ssx, s.nss, norm_w, s.eps, Epacked, s.H, nullptr, 0, logits, s.V, tmax, tmax ? nt : 0, s.B, s.V, s.H, 0, 1, folded ? 3 : 1, s.stream);
    if (s.v2) return bra_dec_gemm2(x, s.H, s.ssx, s.nss, norm_w, s.eps, E, s.H, nullptr, 0, logits, s.V, tmax, tmax ? nt : 0, s.B, s.V, s.H, 0, 1, s.stream);
    int rc = bra_dec_gemm(x, s.H, norm_w, s.eps, E, s.H, nullptr, 0, logits, s.V, s.B, s.V, s.H, 0, 1, s.stream);
    if (rc || !tmax) return rc;
    return bra_tile_max(logits, s.V, s.B, s.V, tmax, nt, s.stream);
}

// Fused variant (k_decfused.hip): 6 launches per layer.  The layer records carry ROLLOUT weights in
// Wqkv / Wo / Wgu / Wd: LoRA already merged (W + s B A, the same merge PEFT's merge_and_unload performs,
// reason.py:428-446) and gate/up rows interleaved in blocks of 8 for the SwiGLU epilogue; the LoRA fields are unused.
// Ends with the final RMSNorm + tied lm_head into fp32 logits [B, V] when `logits` is non-null.
extern "C" int bra_qwen_decode_step_fused(const void* layers_host, int L, int B, int H, int Hq, int Hkv, int hd, int F,
                                          int Smax, int V, float eps, float scale, const void* E, const void* norm_w,
                                          const float* cosT, const float* sinT, const int* tok, const int* pos,
                                          const void* kmask, int cur_len, const int* len_dev, int embed_done, void* x, void* qkv, void* o, void* h, void* act,
                                          float* ss_ws, int nss, float* part_o, float* part_ml, float* logits, void* stream) {
    const Layer* ls = (const Layer*)layers_host;
    const int Nq = Hq * hd, Nkv = Hkv * hd, Nqkv = Nq + 2 * Nkv;
    const int nchunk = (cur_len + 1 + 63) / 64;
    int rc;
#define CK(call) do { rc = (call); if (rc) return rc; } while (0)
    const StepGemms sg = step_gemms(ss_ws, nss, B, H, Nq, Nqkv, F, V, eps, stream);
    if (!(embed_done && sg.v2)) {          // else bra_sample_embed already left x = E[tok] and its RMSNorm statistics
        CK(bra_embed_scatter_fwd(tok, nullptr, E, H, nullptr, 0, x, H, B, H, stream));
        CK(sg_begin(sg, x));
    }
    for (int li = 0; li < L; ++li) {
        const Layer& l = ls[li];
        CK(sg_qkv(sg, l, x, qkv));
        CK(bra_dec_attn_partial(qkv, Nqkv, l.qn, l.kn, cosT, sinT, pos, l.kc, l.vc, kmask, part_o, part_ml, B, Hq, Hkv, hd,
                                Smax, cur_len, eps, scale, 0, 0, len_dev, stream));
        CK(bra_attn_decode_merge(part_o, part_ml, o, B, Hq, hd, nchunk, len_dev, 0, stream));
        CK(sg_tail(sg, l, o, x, h, act));
    }
    if (logits) CK(sg_head(sg, x, norm_w, E, L > 0 ? ls[0].head_packed : nullptr, L > 0 ? (ls[0].flags & 4) : 0, logits, L > 0 ? ls[0].head_tmax : nullptr, (L > 0 && (ls[0].flags & 16)) ? ls[0].sc_head : nullptr));
#undef CK
    return 0;
}


// Shared-prefix variant: the B = R * copies sequences are grouped by prompt; the prompt part of the attention reads
// ONE copy of the prompt K / V^T (bra_dec_attn_shared), the completion part the per-sequence completion cache
// kc / vc [B, Hkv, C, hd] at index `t` (number of completion tokens already cached).  6 launches per layer.
// With `t_dev` (device int) the kernels read t from memory and the host `t` only sizes the grids (pass C - 1): the
// launch arguments are then identical for every step, so the step can be captured once in a hipGraph and replayed.
extern "C" int bra_qwen_decode_step_shared(const void* layers_host, int L, int R, int copies, int H, int Hq, int Hkv, int hd,
                                           int F, int P, long vt_pitch, int C, int V, float eps, float scale, const void* E,
                                           const void* norm_w, const float* cosT, const float* sinT, const int* tok,
                                           const int* pos, const void* pmask, int t, const int* t_dev, int embed_done, void* x, void* qkv, void* o,
                                           void* h, void* act, float* ss_ws, int nss, float* part_o, float* part_ml, float* logits, void* stream) {
    const Layer* ls = (const Layer*)layers_host;
    const int B = R * copies;
    const int Nq = Hq * hd, Nkv = Hkv * hd, Nqkv = Nq + 2 * Nkv;
    const int npc = (P + 63) / 64, ncc = (t + 1 + 63) / 64, ntot = npc + ncc;
    int rc;
#define CK(call) do { rc = (call); if (rc) return rc; } while (0)
    const StepGemms sg = step_gemms(ss_ws, nss, B, H, Nq, Nqkv, F, V, eps, stream);
    if (!(embed_done && sg.v2)) {          // else bra_sample_embed already left x = E[tok] and its RMSNorm statistics
        CK(bra_embed_scatter_fwd(tok, nullptr, E, H, nullptr, 0, x, H, B, H, stream));
        CK(sg_begin(sg, x));
    }
    for (int li = 0; li < L; ++li) {
        const Layer& l = ls[li];
        CK(sg_qkv(sg, l, x, qkv));
        CK(bra_dec_attn_both(qkv, Nqkv, l.qn, l.kn, cosT, sinT, pos, l.kp, (long)Hkv * P * hd, (long)P * hd, (long)hd, l.vtp,
                             (long)Hkv * hd * vt_pitch, (long)hd * vt_pitch, vt_pitch, pmask, l.kc, l.vc, part_o, part_ml, R,
                             copies, Hq, Hkv, hd, P, C, t, eps, scale, t_dev, L > 0 ? ls[0].rope_rows : nullptr, stream));
        CK(bra_attn_decode_merge(part_o, part_ml, o, B, Hq, hd, ntot, t_dev, npc, stream));
        CK(sg_tail(sg, l, o, x, h, act));
    }
    if (logits) CK(sg_head(sg, x, norm_w, E, L > 0 ? ls[0].head_packed : nullptr, L > 0 ? (ls[0].flags & 4) : 0, logits, L > 0 ? ls[0].head_tmax : nullptr, (L > 0 && (ls[0].flags & 16)) ? ls[0].sc_head : nullptr));
#undef CK
    return 0;
}

// Shared-prefix variant on k_decattn.hip (bra_dec_attn_one: items kernel + merge kernel): 6 launches per layer.  The layer
// records' `vc` fields hold the TRANSPOSED completion V caches [B, Hkv, hd, cp].
extern "C" int bra_qwen_decode_step_one(const void* layers_host, int L, int R, int copies, int H, int Hq, int Hkv, int hd,
                                        int F, int P, long vt_pitch, int C, long cp, int V, float eps, float scale, const void* E,
                                        const void* norm_w, const float* cosT, const float* sinT, const int* tok,
                                        const int* pos, const void* pmask, int t, const int* t_dev, int embed_done, void* x, void* qkv, void* o,
                                        void* h, void* act, float* ss_ws, int nss, float* part_o, float* part_ml, int nslot,
                                        float* logits, void* stream) {
    const Layer* ls = (const Layer*)layers_host;
    const int B = R * copies;
    const int Nq = Hq * hd, Nkv = Hkv * hd, Nqkv = Nq + 2 * Nkv;
    int rc;
#define CK(call) do { rc = (call); if (rc) return rc; } while (0)
    const StepGemms sg = step_gemms(ss_ws, nss, B, H, Nq, Nqkv, F, V, eps, stream);
    if (!(embed_done && sg.v2)) {          // else bra_sample_embed already left x = E[tok] and its RMSNorm statistics
        CK(bra_embed_scatter_fwd(tok, nullptr, E, H, nullptr, 0, x, H, B, H, stream));
        CK(sg_begin(sg, x));
    }
    for (int li = 0; li < L; ++li) {
        const Layer& l = ls[li];
        CK(sg_qkv(sg, l, x, qkv));
        CK(bra_dec_attn_one(qkv, Nqkv, l.qn, l.kn, cosT, sinT, pos, L > 0 ? ls[0].rope_rows : nullptr, l.kp, (long)Hkv * P * hd,
                            (long)P * hd, (long)hd, l.vtp, (long)Hkv * hd * vt_pitch, (long)hd * vt_pitch, vt_pitch, pmask, l.kc, l.vc, cp,
                            part_o, part_ml, nslot, o, Nq, R, copies, Hq, Hkv, hd, P, C, t, eps, scale, t_dev, stream));
        CK(sg_tail(sg, l, o, x, h, act));
    }
    if (logits) CK(sg_head(sg, x, norm_w, E, L > 0 ? ls[0].head_packed : nullptr, L > 0 ? (ls[0].flags & 4) : 0, logits, L > 0 ? ls[0].head_tmax : nullptr, (L > 0 && (ls[0].flags & 16)) ? ls[0].sc_head : nullptr));
#undef CK
    return 0;
}

#ifdef BRA_DEBUG      // the persistent decode step is an opt-in experiment: libbioreason_hip_debug.so only
// bra_qwen_decode_step_one with the layer loop as ONE persistent launch (k_persist.hip: bra_qwen_layers_persist) — embed +
// statistics (unless the sampler left them), one launch for all decoder layers, lm_head: 2-3 launches per token instead of ~170.
// `layers_host` as above (its record 0 supplies the packed lm_head); `layers_dev` the device table of the persistent kernel.
// Returns BRA_ERR_UNSUPPORTED where the persistent kernel does (callers fall back to bra_qwen_decode_step_one).
extern "C" int bra_qwen_decode_step_persist(const void* layers_host, const void* layers_dev, int L, int R, int copies, int H, int Hq,
                                            int Hkv, int hd, int F, int P, long vt_pitch, int C, long cp, int V, float eps, float scale,
                                            const void* E, const void* norm_w, const float* cosT, const float* sinT, const int* tok,
                                            const int* pos, const void* pmask, int t, const int* t_dev, int embed_done, void* x,
                                            void* qkv, void* o, void* h, void* act, float* ss_ws, int nss, float* part_o,
                                            float* part_ml, int nslot, float* logits, void* sync, int prefetch, int stop_after,
                                            int timeout_us, void* stream) {
    const Layer* ls = (const Layer*)layers_host;
    const int B = R * copies;
    const int Nq = Hq * hd, Nkv = Hkv * hd, Nqkv = Nq + 2 * Nkv;
    int rc;
#define CK(call) do { rc = (call); if (rc) return rc; } while (0)
    const StepGemms sg = step_gemms(ss_ws, nss, B, H, Nq, Nqkv, F, V, eps, stream);
    if (!sg.v2 || L <= 0 || !(ls[0].flags & 1) || !(ls[0].flags & 2)) return BRA_ERR_UNSUPPORTED;     // packed + folded weights only
    if (!(embed_done && sg.v2)) {
        CK(bra_embed_scatter_fwd(tok, nullptr, E, H, nullptr, 0, x, H, B, H, stream));
        CK(sg_begin(sg, x));
    }
    if (stop_after <= -100) {
        // diagnostics: phase by phase, the ops whose mask bit is set (1 qkv, 2 attention, 4 o, 8 gate/up, 16 down) by the LAUNCHED
        // kernels, the others by single-phase windows of the persistent kernel — localises a numerical difference between the two
        const int mask = -stop_after - 100;
        char* ld = (char*)layers_dev;
        const int rec = bra_persist_layer_desc_size();
        for (int li = 0; li < L; ++li) {
            const Layer& l = ls[li];
            auto win = [&](int p0, int p1) -> int {        // phases [p0, p1) of layer li on a one-layer table
                return bra_qwen_layers_persist(ld + (long)li * rec, 1, R, copies, H, Hq, Hkv, hd, F, P, vt_pitch, C, cp, eps, scale, cosT,
                                               sinT, pos, ls[0].rope_rows, pmask, t, t_dev, x, qkv, o, h, act, ss_ws, nss, part_o,
                                               part_ml, nslot, sync, 0, -1000 - (p0 * 8 + p1), timeout_us, stream);
            };
            if (mask & 1) CK(sg_qkv(sg, l, x, qkv)); else CK(win(0, 1));
            if (mask & 2) CK(bra_dec_attn_one(qkv, Nqkv, l.qn, l.kn, cosT, sinT, pos, ls[0].rope_rows, l.kp, (long)Hkv * P * hd,
                                              (long)P * hd, (long)hd, l.vtp, (long)Hkv * hd * vt_pitch, (long)hd * vt_pitch, vt_pitch, pmask,
                                              l.kc, l.vc, cp, part_o, part_ml, nslot, o, Nq, R, copies, Hq, Hkv, hd, P, C, t, eps, scale,
                                              t_dev, stream));
            else CK(win(1, 3));
            const int pk = l.flags & 1;
            if (mask & 4) CK(bra_dec_gemm2_packed(o, sg.Nq, nullptr, 0, nullptr, 0.f, l.Wo, sg.Nq, x, sg.H, h, sg.H, sg.ssh, sg.nss, sg.B, sg.H, sg.Nq, 0, 0, pk, stream));
            else CK(win(3, 4));
            if (mask & 8) CK(bra_dec_gemm2_packed(h, sg.H, sg.ssh, sg.nss, l.ln2, sg.eps, l.Wgu, sg.H, nullptr, 0, act, sg.F, nullptr, 0, sg.B, 2 * sg.F, sg.H, 1, 0, l.flags & 3, stream));
            else CK(win(4, 5));
            if (mask & 16) CK(bra_dec_gemm2_packed(act, sg.F, nullptr, 0, nullptr, 0.f, l.Wd, sg.F, h, sg.H, x, sg.H, sg.ssx, sg.nss, sg.B, sg.H, sg.F, 0, 0, pk, stream));
            else CK(win(5, 6));
        }
    } else
    CK(bra_qwen_layers_persist(layers_dev, L, R, copies, H, Hq, Hkv, hd, F, P, vt_pitch, C, cp, eps, scale, cosT, sinT, pos,
                               ls[0].rope_rows, pmask, t, t_dev, x, qkv, o, h, act, ss_ws, nss, part_o, part_ml, nslot, sync,
                               prefetch, stop_after, timeout_us, stream));
    if (logits) CK(sg_head(sg, x, norm_w, E, ls[0].head_packed, ls[0].flags & 4, logits, ls[0].head_tmax, (ls[0].flags & 16) ? ls[0].sc_head : nullptr));
#undef CK
    return 0;
}
#endif  // BRA_DEBUG
