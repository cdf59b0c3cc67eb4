// k_decattn.hip — decode attention of the shared-prefix rollout: an items kernel and a merge kernel.
// The step of HF's `_sample` loop (TF:generation/utils.py:2876-2925) runs Qwen3Attention.forward (TF:qwen3:231-284) on
// ONE new token per sequence: q/k RMSNorm + RoPE (TF:qwen3:252-256), KV-cache append, softmax(q K^T) V over the prompt
// and the completion so far.  GRPO decodes `copies` rollouts of each prompt together (grpo_trainer.py:107-116): they share
// ONE copy of the prompt K / V^T.  At 8 sequences this op moves ~10-17 MB and is bound by latency and by the ~16 B/clk a CU
// can fill, not by HBM bandwidth, so the work is cut into many independent one-wave workgroups ("items"):
//   prompt item      (prompt r, kv-head, 64-key chunk): S^T = K Q^T and O^T = V^T P on MFMA for the <= 16 query rows
//                    (copies x q-heads of the group) that share those keys; q is requested first so that its norm / rotate
//                    chain runs under the K / V^T flight;
//   completion item  (sequence, kv-head, 64-key chunk of the keys cached so far): the same MFMA body on the sequence's own
//                    K rows and V^T rows (the completion V cache is kept TRANSPOSED, [B, Hkv, hd, cp], for this);
//   new-key item     (sequence, q-head): normalises + rotates q and the NEW k, appends k / v to the caches and emits the new
//                    key's own partial (max = its score, sum = 1, O = v) — no item ever reads the row being appended;
// each writes (max, sum, O) partials; dec_attn_merge_kernel (one wave per (sequence, q-head), every load issued before the
// first is used, two partial rows per 16-byte-per-lane instruction) combines them after the kernel boundary.
// A single-launch form (tail waves waiting on arrival counters, partials published write-through) was built and measured:
// 20-29 us per layer against 13.5-15.8 for two launches — waves that poll starve the items' memory traffic (NOTES.md).
#include "bra_decattn.h"
#include "bra_api_internal.h"

namespace bra {

template <int HD, int G>
__global__ __launch_bounds__(64) void dec_attn_items_kernel(DecOneArgs a) {
    dec_attn_item<HD, G, 0>(a, (int)blockIdx.x);           // dispatch order = workgroup id
}

// the merge reads nine fields of the record: all of them lead the argument list (14 dwords, delivered in SGPRs with the wave:
// kernel-argument preload, -mllvm -amdgpu-kernarg-preload-count) — nothing is fetched from the argument segment, which the host
// has just written, before the requests go out (-0.4 us per launch).  The items kernel needs nine pointers before its first
// request: they do not fit, it keeps the record.
template <int HD>
__global__ __launch_bounds__(64) void dec_attn_merge_kernel(float* m_part_ml, float* m_part_o, bf16_t* m_o, int m_ldo, int m_Hq, int m_nslot,
                                                            int m_npc, int m_t, const int* m_t_ptr, DecOneArgs a0) {
    DecOneArgs a = a0;
    a.part_ml = m_part_ml; a.part_o = m_part_o; a.o = m_o; a.ldo = m_ldo; a.Hq = m_Hq; a.nslot = m_nslot; a.npc = m_npc; a.t = m_t; a.t_ptr = m_t_ptr;
    dec_attn_merge_one<HD, 0>(a, (int)blockIdx.x, (int)blockIdx.y);
}

}  // namespace bra

using namespace bra;

// nslot = partial slots per (sequence, q-head) >= ceil(P / 64) + ceil(C / 64) + 1, <= 256.
extern "C" int bra_dec_attn_one(const void* qkv, long ldqkv, const void* qw, const void* kw, const float* cosT, const float* sinT,
                                const int* pos, const float* rope_rows, const void* kp, long kp_sr, long kp_sh, long kp_ss,
                                const void* vtp, long vt_sr, long vt_sh, long vt_sd, const void* pmask, void* kc, void* vct, long cp,
                                float* part_o, float* part_ml, int nslot, void* o, long ldo, int R, int copies,
                                int Hq, int Hkv, int hd, int P, int C, int t, float eps, float scale, const int* t_dev, void* stream) {
    if (R <= 0 || copies <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || P <= 0 || t < 0 || t >= C) return BRA_ERR_ARG;
    const int G = Hq / Hkv;
    // more than 16 query rows per (prompt, kv-head): split the rollouts of a prompt into `rsplit` virtual prompts of <= 16 / G each
    int rsplit = 1;
    while (rsplit <= copies && (copies % rsplit || (copies / rsplit) * G > 16)) ++rsplit;
    if (rsplit > copies || G > 16) return BRA_ERR_UNSUPPORTED;
    const int R_phys = R;
    R *= rsplit; copies /= rsplit;
    (void)R_phys;
    if (!qkv || !qw || !kw || !cosT || !sinT || !pos || !kp || !vtp || !kc || !vct || !part_o || !part_ml || !o) return BRA_ERR_ARG;
    const int npc = (P + 63) / 64, ncc = (t + 63) / 64;
    if (vt_sd < (long)npc * 64 || vt_sd % 8 || kp_ss % 8 || cp < ((C + 63) / 64) * 64 || cp % 8 || ldqkv % 8 || ldo % 4) return BRA_ERR_ARG;
    if (nslot < npc + (C + 63) / 64 + 1 || nslot > 256) return BRA_ERR_ARG;
    DecOneArgs a = {(const bf16_t*)qkv, ldqkv, (const bf16_t*)qw, (const bf16_t*)kw, cosT, sinT, pos, rope_rows,
                    (const bf16_t*)kp, kp_sr, kp_sh, kp_ss, (const bf16_t*)vtp, vt_sr, vt_sh, vt_sd, (const uint8_t*)pmask,
                    (bf16_t*)kc, (bf16_t*)vct, cp, part_o, part_ml, (bf16_t*)o, ldo,
                    R, copies, Hq, Hkv, P, C, t, nslot, npc, ncc, eps, scale, t_dev,
                    1.f / (float)Hq, 1.f / (float)npc, 1.f / (float)Hkv, 1.f / (float)(ncc > 0 ? ncc : 1), 1.f / (float)copies,
                    rsplit, 1.f / (float)rsplit};
    bra_stream_t st = (bra_stream_t)stream;
    const int nitems = npc * Hkv * R + ncc * copies * Hkv * R + R * copies * Hq;
    // 24-bit factors / 32-bit element offsets inside one (prompt, kv-head) block, one activation matrix, the partial buffers
    const long lim = 1L << 30, f24 = 1L << 24;
    const long Bq = (long)R * copies;
    if (nitems >= (1 << 21) || kp_ss >= f24 || vt_sd >= f24 || cp >= f24 || ldqkv >= f24 || (long)(npc * 64) * kp_ss >= lim ||
        (long)hd * vt_sd >= lim || (long)hd * cp >= lim || Bq * ldqkv >= lim || Bq * Hq * nslot * hd >= lim || Bq * hd >= lim)
        return BRA_ERR_UNSUPPORTED;
#define BRA_DO(HD_, G_)                                                                                   \
    if (hd == HD_ && G == G_) {                                                                           \
        BRA_LAUNCH((dec_attn_items_kernel<HD_, G_>), dim3(nitems), dim3(64), 0, st, a);                   \
        BRA_LAUNCH((dec_attn_merge_kernel<HD_>), dim3(Hq, R * copies), dim3(64), 0, st, a.part_ml, a.part_o, a.o, (int)a.ldo, a.Hq,   \
                   a.nslot, a.npc, a.t, a.t_ptr, a);                                                      \
        return BRA_LAUNCH_STATUS();                                                                       \
    }
    BRA_DO(128, 1) BRA_DO(128, 2) BRA_DO(128, 4) BRA_DO(64, 1) BRA_DO(64, 2) BRA_DO(64, 4)
#undef BRA_DO
    return BRA_ERR_UNSUPPORTED;
}
