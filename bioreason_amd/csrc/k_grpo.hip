// k_grpo.hip — rollout-side and loss-side GRPO arithmetic:
//   * next-token sampler: temperature -> top-k -> top-p -> multinomial / argmax
//     (HF warpers TF:generation/logits_process.py:238,473,542 and _sample
//      TF:generation/utils.py:2897-2925 with the GenerationConfig of grpo_trainer.py:384-391)
//   * first-EOS completion mask            (grpo_trainer.py:605-609)
//   * group-normalised advantages          (grpo_trainer.py:682-699)
//   * clipped-ratio + k3-KL loss, fwd+bwd  (grpo_trainer.py:786-814)
#include "bra_device.h"
#include "bra_api_internal.h"

namespace bra {

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// uniform in [0,1) from (seed, step, row): counter-based, so replaying a captured graph with a new
// `step` word in device memory yields a new draw without host involvement.
__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t step, uint32_t row) {
    uint32_t h = hash_u32(seed ^ hash_u32(step * 0x9e3779b9u + 0x85ebca6bu) ^ hash_u32(row + 0xc2b2ae35u));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// One workgroup per sequence.  logits: fp32 [B, V].  Top-k by k rounds of a block-wide arg-max
// under the strict total order (value desc, index asc) — k <= 64.
// stage 1 of the sampler: per (row, vocabulary slice) top-k under the order (value desc, index asc).  64 slices per row keep
// the whole chip busy instead of one workgroup per sequence scanning 152k logits k times.  A candidate is a 64-bit key
// (order-preserving image of the value | 0x7fffffff - index): the order is then an unsigned maximum.  The slice is split over the
// 16 lane-rows of the workgroup; each row extracts ITS top-k with k rounds of a row-wide maximum (DPP rotations, no barrier,
// no LDS), then one row merges the 16 sorted lists the same way.  (First form: k rounds of a block-wide arg-max with two barriers
// and a serial section each, 26 us per token at Qwen3's vocabulary.)
constexpr int kSlices = 64;
#ifdef BRA_EMU
__device__ __forceinline__ void pin_u32(uint32_t&) {}
__device__ __forceinline__ void pin_u32x4(u32x4&) {}
#else
__device__ __forceinline__ void pin_u32x4(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin_u32(uint32_t& v) { asm volatile("" : "+v"(v)); }
#endif
__device__ __forceinline__ uint32_t ord_f32(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float unord_f32(uint32_t o) {
    const uint32_t u = (o >> 31) ? (o ^ 0x80000000u) : ~o;
    return __builtin_bit_cast(float, u);
}
__device__ __forceinline__ uint64_t cand_key(float v, int idx) { return ((uint64_t)ord_f32(v) << 32) | (uint32_t)(0x7fffffff - idx); }

// EPT = candidates per thread: a slice holds at most 256 * EPT elements (10 covers Qwen3's 151 936-entry vocabulary; every
// round of the extraction rescans a winner lane's EPT registers, so the count is kept as small as the vocabulary allows)
// (nsl = gridDim.x slices per row: 64 over the logits themselves, 8 over the V / 16 tile maxima of bra_sample_tiles)
template <int EPT>
__global__ __launch_bounds__(256) void topk_slices_kernel(const float* logits, long ldl, int V, int k, float* cand_v,
                                                          int* cand_i) {
    __shared__ uint64_t lists[16][64];            // [lane-row of the workgroup][rank]
    const int row = (int)blockIdx.y, sl = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63;
    const int kSlices = (int)gridDim.x;
    const int per = (V + kSlices - 1) / kSlices;
    const int lo = sl * per, hi = (lo + per) < V ? (lo + per) : V;
    const float* lr = logits + (long)row * ldl;
    // all EPT requests first, from clamped addresses: `i < hi ? cand_key(lr[i], i) : 0` compiles to EPT branches with one load
    // and a full wait each — ten memory round trips in series at Qwen3's vocabulary
    uint64_t key[EPT];
    float val[EPT];
    const int last = hi > lo ? hi - 1 : (V > 0 ? V - 1 : 0);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = lo + tid + 256 * e;
        val[e] = lr[i < hi ? i : last];
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = lo + tid + 256 * e;
        const uint64_t kk = cand_key(val[e], i);
        const uint64_t live = i < hi ? ~0ull : 0ull;
        key[e] = kk & live;
    }
    uint64_t best = 0ull;
#pragma unroll
    for (int e = 0; e < EPT; ++e) best = key[e] > best ? key[e] : best;
    const int lrow = tid >> 4;                    // 0 .. 15
    for (int round = 0; round < k; ++round) {
        const uint64_t w = row_max_u64(best);
        if ((lane & 15) == 0) lists[lrow][round] = w;
        if (best == w && w != 0ull) {             // (keys are unique: the index is part of them)
            best = 0ull;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                key[e] = key[e] == w ? 0ull : key[e];
                best = key[e] > best ? key[e] : best;
            }
        }
    }
    __syncthreads();
    if (tid >= 16) return;
    // lane = list: k rounds over the list heads
    int ptr = 0;
    uint64_t head = lists[tid][0];
    for (int round = 0; round < k; ++round) {
        const uint64_t w = row_max_u64(head);
        if (tid == 0) {
            const long o = ((long)row * kSlices + sl) * k + round;
            cand_v[o] = w ? unord_f32((uint32_t)(w >> 32)) : -3.0e38f;
            cand_i[o] = w ? 0x7fffffff - (int)(uint32_t)w : 0x7fffffff;
        }
        if (head == w && w != 0ull) {
            ++ptr;
            head = ptr < k ? lists[tid][ptr] : 0ull;
        }
    }
}

// temperature / top-p / draw over the k survivors (value-descending), executed by one thread
// (HF: TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper, multinomial — TF:generation/logits_process.py, utils.py:2905-2925)
// `p` = k floats of LDS scratch (a per-thread array indexed at run time would live in scratch memory: every access a
// round trip through the memory system, in a single-thread serial section)
// `step` / `was_finished`: the step word and the row's finished flag, read by the caller (the merge kernel requests them with its
// first loads: read here they were two more memory round trips in the single-thread tail of every token).  Returns the token.
__device__ inline int sample_pick(float* p, const float* top_v, const int* top_i, int k, int row, float temperature, float top_p,
                                  int do_sample, uint32_t seed, int step, int was_finished, uint8_t* finished, int pad_id, int eos_id,
                                  int eos_id2, int* out_ids, float* out_logp, int* tokens_out, long ldt) {
    int choice = top_i[0];
    float lp = 0.f;
    if (do_sample) {
        // temperature, softmax over the top-k survivors
        const float inv_t = 1.f / temperature;
        const float mx = top_v[0] * inv_t;
        float z = 0.f;
        for (int j = 0; j < k; ++j) { p[j] = __expf(top_v[j] * inv_t - mx); z += p[j]; }
        for (int j = 0; j < k; ++j) p[j] /= z;
        // top-p (HF: sort ascending, drop tokens whose cumulative prob <= 1 - top_p, keep >= 1)
        int keep = k;
        if (top_p < 1.f) {
            float cum = 0.f;
            for (int j = k - 1; j >= 1; --j) {   // ascending order = from the tail of our descending list
                cum += p[j];
                if (cum <= 1.f - top_p) keep = j; else break;
            }
        }
        float z2 = 0.f;
        for (int j = 0; j < keep; ++j) z2 += p[j];
        const float u = uniform01(seed, (uint32_t)step, (uint32_t)row) * z2;
        float acc = 0.f;
        int pick = keep - 1;
        for (int j = 0; j < keep; ++j) { acc += p[j]; if (u < acc) { pick = j; break; } }
        choice = top_i[pick];
        lp = __logf(p[pick] / z2);
    }
    if (was_finished) choice = pad_id;                // HF: finished sequences emit pad_token_id
    out_ids[row] = choice;
    if (out_logp) out_logp[row] = lp;
    if (tokens_out) tokens_out[(long)row * ldt + step] = choice;
    if (finished && ((eos_id >= 0 && choice == eos_id) || (eos_id2 >= 0 && choice == eos_id2))) finished[row] = 1;   // unfinished &= (token != eos)
    return choice;
}

template <int NT>
__global__ __launch_bounds__(NT) void sample_kernel(const float* logits, long ldl, int V, const int* cand_idx, float temperature,
                                                    int top_k, float top_p, int do_sample, uint32_t seed,
                                                    const int* step_ptr, uint8_t* finished, int pad_id,
                                                    int eos_id, int eos_id2, int* out_ids, float* out_logp, int* tokens_out,
                                                    long ldt) {
    __shared__ float s_val[NT / 64];
    __shared__ int s_idx[NT / 64];
    __shared__ float top_v[64];
    __shared__ int top_i[64];
    const int row = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lr = logits + (long)row * ldl;
    const int k = do_sample ? (top_k > 0 ? (top_k < 64 ? top_k : 64) : 64) : 1;
    float last_v = 3.0e38f;
    int last_i = -1;
    for (int round = 0; round < k; ++round) {
        float bv = -3.0e38f;
        int bi = 0x7fffffff;
        for (int j = tid; j < V; j += NT) {
            const float v = lr[j];
            const int i = cand_idx ? cand_idx[(long)row * ldl + j] : j;     // original vocabulary index
            // candidates strictly after (last_v, last_i) in the order (value desc, index asc)
            const bool after = (v < last_v) || (v == last_v && i > last_i);
            if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
        }
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = wave_shfl_xor(bv, m);
            const int oi = wave_shfl_xor_i(bi, m);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NT / 64; ++w)
                if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) { bv = s_val[w]; bi = s_idx[w]; }
            top_v[round] = bv; top_i[round] = bi;
        }
        __syncthreads();
        last_v = top_v[round];
        last_i = top_i[round];
    }
    __shared__ float pick_ws[64];
    if (tid == 0)
        sample_pick(pick_ws, top_v, top_i, k, row, temperature, top_p, do_sample, seed, step_ptr ? step_ptr[0] : 0,
                    finished ? (int)finished[row] : 0, finished, pad_id, eos_id, eos_id2, out_ids, out_logp, tokens_out, ldt);
}

// stage 2 of the sampler when stage 1 ran: ONE wave per sequence merges the 64 slice lists (each already in the
// order value desc / index asc) — lane = slice, k rounds of a wave arg-max over the list heads, no barriers — then lane 0
// draws, and the wave gathers the chosen token's embedding row into the decode step's input x [B, H] together with its
// RMSNorm statistic (bra_row_sumsq layout), which is what the next launch of the token loop consumes.
__global__ __launch_bounds__(64) void sample_merge_kernel(const float* cand_v, const int* cand_i, int k, float temperature,
                                                          float top_p, int do_sample, uint32_t seed, const int* step_ptr,
                                                          uint8_t* finished, int pad_id, int eos_id, int eos_id2, int* out_ids,
                                                          float* out_logp, int* tokens_out, long ldt, const bf16_t* E,
                                                          long lde, int H, bf16_t* x, long ldx, float* ss, int nss) {
    BRA_DYN_SMEM(smem);                           // values [64][k] | indices [64][k]
    __shared__ float top_v[64];
    __shared__ int top_i[64];
    __shared__ int s_choice;
    __shared__ float pick_ws[64];
    float* sv = reinterpret_cast<float*>(smem);
    int* si = reinterpret_cast<int*>(smem) + kSlices * k;
    const int row = (int)blockIdx.x, lane = lane_id();
    const long base = (long)row * kSlices * k;
    // the step word and the finished flag travel with the candidate loads (clamped pointers: no branch around a load)
    uint32_t step_w = *reinterpret_cast<const uint32_t*>(step_ptr ? (const void*)step_ptr : (const void*)cand_i);
    uint32_t fin_w = *(finished ? finished + row : reinterpret_cast<const uint8_t*>(cand_i));
    if ((k & 3) == 0) {
        // lane = slice: its k candidates are contiguous; all 16-byte loads are requested before the first is used
        const f32x4* gv = reinterpret_cast<const f32x4*>(cand_v + base + (long)lane * k);
        const u32x4* gi = reinterpret_cast<const u32x4*>(cand_i + base + (long)lane * k);
        f32x4 tv[16];
        u32x4 ti[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int qq = 4 * q < k ? q : 0; tv[q] = gv[qq]; ti[q] = gi[qq]; }
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (4 * q < k) {
                *reinterpret_cast<f32x4*>(sv + lane * k + 4 * q) = tv[q];
                *reinterpret_cast<u32x4*>(si + lane * k + 4 * q) = ti[q];
            }
    } else {
        for (int j = lane; j < kSlices * k; j += 64) { sv[j] = cand_v[base + j]; si[j] = cand_i[base + j]; }
    }
    pin_u32(step_w); pin_u32(fin_w);              // (keeps the two loads up there: instruction selection sinks a load to its first use)
    const int step_v = step_ptr ? (int)step_w : 0, fin_v = finished ? (int)(fin_w & 0xffu) : 0;
    __syncthreads();
    int ptr = 0;
    uint64_t head = si[lane * k] != 0x7fffffff ? cand_key(sv[lane * k], si[lane * k]) : 0ull;
    for (int round = 0; round < k; ++round) {
        const uint64_t w = wave_max_u64(head);
        if (lane == 0) {
            top_v[round] = w ? unord_f32((uint32_t)(w >> 32)) : -3.0e38f;
            top_i[round] = w ? 0x7fffffff - (int)(uint32_t)w : 0x7fffffff;
        }
        if (head == w && w != 0ull) {             // the list that supplied the winner moves to its next entry
            ++ptr;
            head = (ptr < k && si[lane * k + ptr] != 0x7fffffff) ? cand_key(sv[lane * k + ptr], si[lane * k + ptr]) : 0ull;
        }
    }
    __syncthreads();
    if (lane == 0)
        s_choice = sample_pick(pick_ws, top_v, top_i, k, row, temperature, top_p, do_sample, seed, step_v, fin_v, finished, pad_id,
                               eos_id, eos_id2, out_ids, out_logp, tokens_out, ldt);
    __syncthreads();
    if (!E) return;
    const int tok = s_choice;
    float acc = 0.f;
    // the embedding row (H <= 2048: four 16-byte chunks per lane) is requested in one go, then stored
    const int nch = H / 8;
    const bf16_t* er = E + (long)tok * lde;
    u32x4 ev[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int j = lane + 64 * u; ev[u] = ld16(er + (j < nch ? j : 0) * 8); }
#pragma unroll
    for (int u = 0; u < 4; ++u) pin_u32x4(ev[u]);          // (or the load of a chunk is sunk under its `j < nch` test again)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u;
        if (j < nch) {
            st16(x + (long)row * ldx + j * 8, ev[u]);
            float f[8];
            unpack8(ev[u], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += f[i] * f[i];
        }
    }
    for (int j = lane + 256; j < nch; j += 64) {
        const u32x4 v = ld16(er + j * 8);
        st16(x + (long)row * ldx + j * 8, v);
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += f[i] * f[i];
    }
    acc = wave_sum<64>(acc);
    if (ss) for (int c = lane; c < nss; c += 64) ss[(long)row * nss + c] = c == 0 ? acc : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Sampler over TILE MAXIMA (round 4).  The lm_head projection's epilogue (bra_decgemm.h) leaves, next to the fp32 logits, the
// maximum of every 16-column tile: tmax [B, ceil(V / 16)].  Under the order (value desc, index asc) every one of the k best
// logits of a row lies in one of the k best TILES under (tile maximum desc, tile index asc): a tile T that holds one of them and
// is not among those would have k tiles in front of it, each with an element that precedes T's best element (a larger value, or
// the same value at a smaller index — tiles are contiguous index ranges), hence k elements in front of one of the k best.  So
//   stage 1  topk_slices_kernel over the 9 496 maxima of a row (8 slices) instead of 151 936 logits (64 slices),
//   stage 2  ONE wave per row: merges the 8 slice lists into the k best tiles, gathers their 16 k logits, extracts the k best
//            (the same list the 64-slice path produces: same order, same ties), draws, gathers the embedding row of the token
//            and — the launch that used to follow every decode step — moves the row's rotary position on: pos_out = pos0 + step
//            and the (cos | sin) row of that position for the decode attention of the step that follows.
constexpr int kTileSlices = 8;

__global__ __launch_bounds__(256) void tile_max_kernel(const float* logits, long ldl, int V, float* tmax, long ldm) {
    const int row = (int)blockIdx.y, tile = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int ntiles = (V + 15) / 16;
    if (tile >= ntiles) return;
    const float* lr = logits + (long)row * ldl;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { const int c = tile * 16 + j; v[j] = lr[c < V ? c : V - 1]; }
    float m = v[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) m = fmaxf(m, v[j]);
    tmax[(long)row * ldm + tile] = m;
}

// NL = list entries per lane in stage 2a (nsl * k <= 64 * NL), NU = gathered logits per lane (16 k <= 64 * NU)
template <int NL, int NU>
__global__ __launch_bounds__(64) void sample_tiles_kernel(const float* logits, long ldl, int V, const float* cand_v, const int* cand_i,
                                                          int nsl, int k, float temperature, float top_p, int do_sample, uint32_t seed,
                                                          const int* step_ptr, int step_arg, uint8_t* finished, int pad_id, int eos_id,
                                                          int eos_id2, int* out_ids, float* out_logp, int* tokens_out, long ldt,
                                                          const bf16_t* E, long lde, int H, bf16_t* x, long ldx, float* ss, int nss,
                                                          const int* pos0, int* pos_out, const float* cosT, const float* sinT, int hd,
                                                          float* rope_rows) {
    __shared__ float sv[kTileSlices * 64];
    __shared__ int si[kTileSlices * 64];
    __shared__ int tsel[64];
    __shared__ float top_v[64];
    __shared__ int top_i[64];
    __shared__ int s_choice;
    __shared__ float pick_ws[64];
    const int row = (int)blockIdx.x, lane = lane_id();
    const long base = (long)row * nsl * k;
    const int nlist = nsl * k;
    // every first-round request up front, from clamped addresses (no branch around a load)
    uint32_t step_w = *reinterpret_cast<const uint32_t*>(step_ptr ? (const void*)step_ptr : (const void*)cand_i);
    uint32_t fin_w = *(finished ? finished + row : reinterpret_cast<const uint8_t*>(cand_i));
    uint32_t pos_w = *reinterpret_cast<const uint32_t*>(pos0 ? (const void*)(pos0 + row) : (const void*)cand_i);
    float lv[NL];
    int li[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int j = lane + 64 * u;
        const int jj = j < nlist ? j : nlist - 1;
        lv[u] = cand_v[base + jj];
        li[u] = cand_i[base + jj];
    }
    pin_u32(step_w); pin_u32(fin_w); pin_u32(pos_w);
    const int step_v = step_ptr ? (int)step_w : step_arg, fin_v = finished ? (int)(fin_w & 0xffu) : 0;
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int j = lane + 64 * u;
        if (j < nlist) { sv[j] = lv[u]; si[j] = li[u]; }
    }
    // the rotary row of the position this token is fed at: requested now, stored at the end (hd <= 128: two floats per lane)
    const int p_new = (int)pos_w + step_v;
    const int half = hd >> 1;
    float rr[2] = {0.f, 0.f};
    if (rope_rows) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int d = lane + 64 * u;
            const int dc = d < hd ? d : 0;
            const float* tp = dc < half ? cosT + ((long)p_new * half + dc) : sinT + ((long)p_new * half + dc - half);
            rr[u] = *tp;
        }
    }
    __syncthreads();
    // ---- stage 2a: lane = slice list (sorted value desc / tile index asc); k rounds over the heads -> the k best tiles
    {
        int ptr = 0;
        const bool mine = lane < nsl;
        const int l0 = mine ? lane * k : 0;
        uint64_t head = (mine && si[l0] != 0x7fffffff) ? cand_key(sv[l0], si[l0]) : 0ull;
        for (int round = 0; round < k; ++round) {
            const uint64_t w = row_max_u64(head);                 // nsl <= 16: the lists sit in the first lane-row
            if (lane == 0) tsel[round] = w ? 0x7fffffff - (int)(uint32_t)w : -1;
            if (head == w && w != 0ull) {
                ++ptr;
                head = (ptr < k && si[l0 + ptr] != 0x7fffffff) ? cand_key(sv[l0 + ptr], si[l0 + ptr]) : 0ull;
            }
        }
    }
    __syncthreads();
    // ---- stage 2b: the logits of those tiles (16 k <= 64 NU values), all requested before the first is used
    const float* lr = logits + (long)row * ldl;
    float gv[NU];
    int gi[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = lane + 64 * u;
        const int tq = j >> 4;
        const int t = tsel[tq < k ? tq : k - 1];
        const int idx = t * 16 + (j & 15);
        const bool ok = tq < k && t >= 0 && idx < V;
        gi[u] = ok ? idx : -1;
        gv[u] = lr[ok ? idx : 0];
    }
    uint64_t key[NU];
    uint64_t best = 0ull;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        key[u] = gi[u] >= 0 ? cand_key(gv[u], gi[u]) : 0ull;
        best = key[u] > best ? key[u] : best;
    }
    for (int round = 0; round < k; ++round) {
        const uint64_t w = wave_max_u64(best);
        if (lane == 0) {
            top_v[round] = w ? unord_f32((uint32_t)(w >> 32)) : -3.0e38f;
            top_i[round] = w ? 0x7fffffff - (int)(uint32_t)w : 0x7fffffff;
        }
        if (best == w && w != 0ull) {
            best = 0ull;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                key[u] = key[u] == w ? 0ull : key[u];
                best = key[u] > best ? key[u] : best;
            }
        }
    }
    __syncthreads();
    if (lane == 0)
        s_choice = sample_pick(pick_ws, top_v, top_i, k, row, temperature, top_p, do_sample, seed, step_v, fin_v, finished, pad_id,
                               eos_id, eos_id2, out_ids, out_logp, tokens_out, ldt);
    __syncthreads();
    if (rope_rows) {
        if (hd <= 128) {
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int d = lane + 64 * u; if (d < hd) rope_rows[(long)row * hd + d] = rr[u]; }
        } else {
            for (int d = lane; d < hd; d += 64)
                rope_rows[(long)row * hd + d] = d < half ? cosT[(long)p_new * half + d] : sinT[(long)p_new * half + d - half];
        }
    }
    if (pos_out && lane == 0) pos_out[row] = p_new;
    if (!E) return;
    const int tok = s_choice;
    float acc = 0.f;
    const int nch = H / 8;
    const bf16_t* er = E + (long)tok * lde;
    u32x4 ev[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int j = lane + 64 * u; ev[u] = ld16(er + (j < nch ? j : 0) * 8); }
#pragma unroll
    for (int u = 0; u < 4; ++u) pin_u32x4(ev[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u;
        if (j < nch) {
            st16(x + (long)row * ldx + j * 8, ev[u]);
            float f[8];
            unpack8(ev[u], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += f[i] * f[i];
        }
    }
    for (int j = lane + 256; j < nch; j += 64) {
        const u32x4 v = ld16(er + j * 8);
        st16(x + (long)row * ldx + j * 8, v);
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += f[i] * f[i];
    }
    acc = wave_sum<64>(acc);
    if (ss) for (int c = lane; c < nss; c += 64) ss[(long)row * nss + c] = c == 0 ? acc : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The tile-maxima sampler as ONE launch (round 6; VERDICT r5 #9: two launches were 9.6 + 17.9 us per token for no bytes).  One
// 1024-thread workgroup per row:
//   A  every thread holds EPT tile maxima (strided: tile = tid + 1024 e); each of the 64 lane-rows of 16 extracts ITS k best with k
//      rounds of a row-wide maximum (as topk_slices_kernel) -> 64 sorted lists in LDS;
//   B  wave 0, lane = list: k rounds over the 64 list heads -> the k best tiles (a strict total order: the same k tiles whatever the
//      partition into lists);  C / D  as sample_tiles_kernel: the 16 k logits of those tiles, k rounds, the draw, the embedding row.
// Against the two launches this drops one launch boundary, the round trip of the candidate lists through memory and one of the four
// k-round stages.  Wave 0 alone gathers the embedding row so that the RMSNorm statistic is summed in the order of
// sample_tiles_kernel (bit-identical x / ss, hence identical tokens downstream).
template <int EPT, int NU>
__global__ __launch_bounds__(1024) void sample_tiles_one_kernel(const float* logits, long ldl, int V, const float* tmax, long ldm, int ntiles,
                                                                int k, float temperature, float top_p, int do_sample, uint32_t seed,
                                                                const int* step_ptr, int step_arg, uint8_t* finished, int pad_id, int eos_id,
                                                                int eos_id2, int* out_ids, float* out_logp, int* tokens_out, long ldt,
                                                                const bf16_t* E, long lde, int H, bf16_t* x, long ldx, float* ss, int nss,
                                                                const int* pos0, int* pos_out, const float* cosT, const float* sinT, int hd,
                                                                float* rope_rows) {
    __shared__ uint64_t lists[64][64];            // [lane-row of the workgroup][rank]
    __shared__ int tsel[64];
    __shared__ float top_v[64];
    __shared__ int top_i[64];
    __shared__ float pick_ws[64];
    const int row = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* tr = tmax + (long)row * ldm;
    // every first-round request up front, from clamped addresses (no branch around a load)
    float val[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + 1024 * e;
        val[e] = tr[i < ntiles ? i : ntiles - 1];
    }
    uint32_t step_w = 0, fin_w = 0, pos_w = 0;
    float rr[2] = {0.f, 0.f};
    if (wave == 0) {
        step_w = *reinterpret_cast<const uint32_t*>(step_ptr ? (const void*)step_ptr : (const void*)tr);
        fin_w = *(finished ? finished + row : reinterpret_cast<const uint8_t*>(tr));
        pos_w = *reinterpret_cast<const uint32_t*>(pos0 ? (const void*)(pos0 + row) : (const void*)tr);
    }
    // ---- A: per lane-row extraction
    {
        uint64_t key[EPT];
        uint64_t best = 0ull;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + 1024 * e;
            const uint64_t kk = cand_key(val[e], i);
            key[e] = i < ntiles ? kk : 0ull;
            best = key[e] > best ? key[e] : best;
        }
        const int lrow = tid >> 4;                // 0 .. 63
        for (int round = 0; round < k; ++round) {
            const uint64_t w = row_max_u64(best);
            if ((lane & 15) == 0) lists[lrow][round] = w;
            if (best == w && w != 0ull) {
                best = 0ull;
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    key[e] = key[e] == w ? 0ull : key[e];
                    best = key[e] > best ? key[e] : best;
                }
            }
        }
    }
    __syncthreads();
    if (wave != 0) return;
    pin_u32(step_w); pin_u32(fin_w); pin_u32(pos_w);
    const int step_v = step_ptr ? (int)step_w : step_arg, fin_v = finished ? (int)(fin_w & 0xffu) : 0;
    // the rotary row of the position this token is fed at: requested now, stored at the end (hd <= 128: two floats per lane)
    const int p_new = (int)pos_w + step_v;
    const int half = hd >> 1;
    if (rope_rows) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int d = lane + 64 * u;
            const int dc = d < hd ? d : 0;
            const float* tp = dc < half ? cosT + ((long)p_new * half + dc) : sinT + ((long)p_new * half + dc - half);
            rr[u] = *tp;
        }
    }
    // ---- B: lane = list; k rounds over the heads -> the k best tiles
    {
        int ptr = 0;
        uint64_t head = lists[lane][0];
        for (int round = 0; round < k; ++round) {
            const uint64_t w = wave_max_u64(head);
            if (lane == 0) tsel[round] = w ? 0x7fffffff - (int)(uint32_t)w : -1;
            if (head == w && w != 0ull) {
                ++ptr;
                head = ptr < k ? lists[lane][ptr] : 0ull;
            }
        }
    }
    __syncthreads();                              // (one wave left: an LDS fence)
    // ---- C: the logits of those tiles (16 k <= 64 NU values), all requested before the first is used
    const float* lr = logits + (long)row * ldl;
    float gv[NU];
    int gi[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = lane + 64 * u;
        const int tq = j >> 4;
        const int t = tsel[tq < k ? tq : k - 1];
        const int idx = t * 16 + (j & 15);
        const bool ok = tq < k && t >= 0 && idx < V;
        gi[u] = ok ? idx : -1;
        gv[u] = lr[ok ? idx : 0];
    }
    uint64_t key[NU];
    uint64_t best = 0ull;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        key[u] = gi[u] >= 0 ? cand_key(gv[u], gi[u]) : 0ull;
        best = key[u] > best ? key[u] : best;
    }
    for (int round = 0; round < k; ++round) {
        const uint64_t w = wave_max_u64(best);
        if (lane == 0) {
            top_v[round] = w ? unord_f32((uint32_t)(w >> 32)) : -3.0e38f;
            top_i[round] = w ? 0x7fffffff - (int)(uint32_t)w : 0x7fffffff;
        }
        if (best == w && w != 0ull) {
            best = 0ull;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                key[u] = key[u] == w ? 0ull : key[u];
                best = key[u] > best ? key[u] : best;
            }
        }
    }
    __syncthreads();                              // (one wave left: an LDS fence)
    int tok = 0;
    if (lane == 0)
        tok = sample_pick(pick_ws, top_v, top_i, k, row, temperature, top_p, do_sample, seed, step_v, fin_v, finished, pad_id,
                          eos_id, eos_id2, out_ids, out_logp, tokens_out, ldt);
    tok = wave_shfl_i(tok, 0);
    if (rope_rows) {
        if (hd <= 128) {
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int d = lane + 64 * u; if (d < hd) rope_rows[(long)row * hd + d] = rr[u]; }
        } else {
            for (int d = lane; d < hd; d += 64)
                rope_rows[(long)row * hd + d] = d < half ? cosT[(long)p_new * half + d] : sinT[(long)p_new * half + d - half];
        }
    }
    if (pos_out && lane == 0) pos_out[row] = p_new;
    if (!E) return;
    float acc = 0.f;
    const int nch = H / 8;
    const bf16_t* er = E + (long)tok * lde;
    u32x4 ev[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int j = lane + 64 * u; ev[u] = ld16(er + (j < nch ? j : 0) * 8); }
#pragma unroll
    for (int u = 0; u < 4; ++u) pin_u32x4(ev[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u;
        if (j < nch) {
            st16(x + (long)row * ldx + j * 8, ev[u]);
            float f[8];
            unpack8(ev[u], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += f[i] * f[i];
        }
    }
    for (int j = lane + 256; j < nch; j += 64) {
        const u32x4 v = ld16(er + j * 8);
        st16(x + (long)row * ldx + j * 8, v);
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += f[i] * f[i];
    }
    acc = wave_sum<64>(acc);
    if (ss) for (int c = lane; c < nss; c += 64) ss[(long)row * nss + c] = c == 0 ? acc : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------------
// General sampler: top_k = 0 (HF: top-k "disabled") or top_k > 64 — the warped distribution can then have thousands of survivors,
// so the fast paths above (temperature / top-p / multinomial over <= 64 sorted survivors) do not apply.  One workgroup per row,
// everything by passes over the row's logits (L2-resident: 600 KB at Qwen3's vocabulary):
//   m = max, kept_k = { l >= the top_k-th largest value }   (TopKLogitsWarper keeps ties of the k-th value: `scores < kth` is removed)
//   kept_p = { l in kept_k : mass of kept_k strictly above l  <  top_p }   (TopPLogitsWarper on the ascending sort: token i is removed
//            iff cumsum_i <= 1 - top_p, i.e. iff the mass sorted AFTER it is >= top_p; min_tokens_to_keep = 1 holds: nothing is above
//            the maximum.  Tokens of EQUAL value at the boundary are kept or dropped together here; HF's stable sort may split them)
//   draw   = multinomial over kept_p in INDEX order (same distribution as HF's; the uniform comes from the same counter hash)
// The two thresholds are found by bisection on the order-preserving integer image of the float (<= 32 passes each): exact, slow
// (~0.1-0.3 ms per token) — a functional path for configurations outside GRPO's top_k = 20 (TF:generation/logits_process.py:238,473,542).
template <int NT>
__device__ __forceinline__ float blk_sum_f(float v, float* red) {
    v = wave_sum<64>(v);
    __syncthreads();
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

template <int NT>
__global__ __launch_bounds__(NT) void sample_full_kernel(const float* logits, long ldl, int V, float temperature, int top_k, float top_p,
                                                         uint32_t seed, const int* step_ptr, int step_arg, uint8_t* finished, int pad_id,
                                                         int eos_id, int eos_id2, int* out_ids, float* out_logp, int* tokens_out, long ldt) {
    __shared__ float red[NT / 64];
    __shared__ float scan[NT];
    const int row = (int)blockIdx.x, tid = (int)threadIdx.x;
    const float* lr = logits + (long)row * ldl;
    const float inv_t = 1.f / temperature;
    float m = -3.0e38f;
    for (int j = tid; j < V; j += NT) m = fmaxf(m, lr[j]);
    m = block_max<NT / 64>(m, red);
    // ---- top-k: the largest x with count(ord(l) >= x) >= k  (= image of the k-th largest value)
    uint32_t kmin = 0u;
    if (top_k > 0 && top_k < V) {
        uint32_t lo = 0u, hi = ord_f32(m);
        while (lo < hi) {
            const uint32_t mid = lo + (uint32_t)(((uint64_t)hi - lo + 1ull) >> 1);
            float c = 0.f;
            for (int j = tid; j < V; j += NT) c += ord_f32(lr[j]) >= mid ? 1.f : 0.f;     // (counts < 2^24: exact in fp32)
            c = blk_sum_f<NT>(c, red);
            if (c >= (float)top_k) lo = mid; else hi = mid - 1u;
        }
        kmin = lo;
    }
    auto e_of = [&](float l) { return __expf((l - m) * inv_t); };
    float zk = 0.f;
    for (int j = tid; j < V; j += NT) { const float l = lr[j]; zk += ord_f32(l) >= kmin ? e_of(l) : 0.f; }
    zk = blk_sum_f<NT>(zk, red);
    // ---- top-p: the smallest x >= kmin with  mass(kept_k, ord(l) > x) < top_p * zk
    uint32_t pmin = kmin;
    if (top_p < 1.f) {
        const float P = top_p * zk;
        uint32_t lo = kmin, hi = ord_f32(m);               // ok(hi) holds: nothing lies above the maximum
        while (lo < hi) {
            const uint32_t mid = lo + (uint32_t)(((uint64_t)hi - lo) >> 1);
            float ms = 0.f;
            for (int j = tid; j < V; j += NT) { const float l = lr[j]; ms += ord_f32(l) > mid ? e_of(l) : 0.f; }      // (mid >= kmin: all of these are in kept_k)
            ms = blk_sum_f<NT>(ms, red);
            if (ms < P) hi = mid; else lo = mid + 1u;
        }
        pmin = lo;
    }
    // ---- multinomial over { ord(l) >= pmin } in index order: thread t owns indices [t per, (t + 1) per)
    const int per = (V + NT - 1) / NT;
    const int j0 = tid * per, j1 = (j0 + per) < V ? (j0 + per) : V;
    float mine = 0.f;
    for (int j = j0; j < j1; ++j) { const float l = lr[j]; mine += ord_f32(l) >= pmin ? e_of(l) : 0.f; }
    scan[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        const int step = step_ptr ? step_ptr[0] : step_arg;
        float zf = 0.f;
        for (int t = 0; t < NT; ++t) zf += scan[t];
        const float u = uniform01(seed, (uint32_t)step, (uint32_t)row) * zf;
        // the thread whose index range holds the draw (the last range with mass takes a draw that rounding pushed past the total)
        float run = 0.f, before = 0.f;
        int owner = 0;
        for (int t = 0; t < NT; ++t) {
            if (scan[t] > 0.f) {
                owner = t; before = run;
                if (u < run + scan[t]) break;
                run += scan[t];
            }
        }
        const int o0 = owner * per, o1 = (o0 + per) < V ? (o0 + per) : V;
        int pick = 0;
        float ep = 1.f, a2 = 0.f;
        for (int j = o0; j < o1; ++j) {
            const float l = lr[j];
            if (ord_f32(l) >= pmin) {
                const float e = e_of(l);
                pick = j; ep = e;
                if (u < before + a2 + e) break;
                a2 += e;
            }
        }
        int choice = pick;
        const int was_finished = finished ? (int)finished[row] : 0;
        if (was_finished) choice = pad_id;
        out_ids[row] = choice;
        if (out_logp) out_logp[row] = __logf(ep / zf);
        if (tokens_out) tokens_out[(long)row * ldt + step] = choice;
        if (finished && ((eos_id >= 0 && choice == eos_id) || (eos_id2 >= 0 && choice == eos_id2))) finished[row] = 1;
    }
}

// synthetic EOS schedule (bench / tests: random-init weights never emit EOS on their own): row b's logit of `token` is
// raised above everything else at the step its schedule names, so the sampler — greedy or warped — draws it there.
__global__ __launch_bounds__(64) void force_token_kernel(float* logits, long ldl, int B, int token, const int* step_ptr, int step_arg,
                                                         const int* at, float* tmax, long ldm) {
    const int b = (int)threadIdx.x;
    const int step = step_ptr ? step_ptr[0] : step_arg;
    if (b < B && at[b] == step) {
        logits[(long)b * ldl + token] = 1.0e30f;
        if (tmax) tmax[(long)b * ldm + (token >> 4)] = 1.0e30f;       // (the maximum of the token's 16-column tile)
    }
}

// counters of the replayed token loop: pos[0 .. n) += 1 (rotary positions), a[0] += 1, b[0] += 1 (step index, cache length)
// (one workgroup of up to 1024 threads: with 64 threads the n * hd row elements were 16 dependent position -> table round trips
//  in series, 6.5 us per token for 4 KB)
__global__ __launch_bounds__(1024) void advance_counters_kernel(int* pos, int n, int* a, int* b, const float* cosT, const float* sinT,
                                                                int hd, float* rows, int nt) {
    const int i = (int)threadIdx.x;                // (nt = blockDim.x as an argument: blockDim lives in the implicit arguments, which are
                                                   //  not among the preloaded dwords — reading it is a memory round trip at entry)
    if (rows) {                                   // one pass: every lane reads the positions it needs before anyone bumps them
        const int half = hd / 2;
        for (int e = i; e < n * hd; e += nt) {
            const int s = e / hd, d = e % hd;
            const int p = pos[s] + 1;
            rows[e] = d < half ? cosT[(long)p * half + d] : sinT[(long)p * half + d - half];
        }
        __syncthreads();
    }
    for (int j = i; j < n; j += nt) pos[j] += 1;
    if (i == 0) { if (a) a[0] += 1; if (b) b[0] += 1; }
}

__global__ __launch_bounds__(64) void rope_rows_kernel(const float* cosT, const float* sinT, const int* pos, int n, int hd, float* rows) {
    const int half = hd / 2;
    for (int e = (int)threadIdx.x; e < n * hd; e += 64) {
        const int s = e / hd, d = e % hd;
        const int p = pos[s];
        rows[e] = d < half ? cosT[(long)p * half + d] : sinT[(long)p * half + d - half];
    }
}

// completion_mask[b, c] = c <= first_eos(b) ; also lengths[b] = mask.sum()
__global__ __launch_bounds__(64) void eos_mask_kernel(const int* ids, int C, int eos_id, int* mask, int* lengths) {
    const int b = (int)blockIdx.x, lane = lane_id();
    int first = C;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const bool e = c < C && ids[(long)b * C + c] == eos_id;
        const uint64_t bal = wave_ballot(e);
        if (bal && first == C) first = c0 + __builtin_ctzll(bal);
    }
    for (int c = lane; c < C; c += 64) mask[(long)b * C + c] = c <= first ? 1 : 0;
    if (lane == 0 && lengths) lengths[b] = first < C ? first + 1 : C;
}

// advantages over all gathered rewards: rewards [N, F] -> sum over F -> per group of G mean / unbiased std
__global__ __launch_bounds__(64) void group_advantage_kernel(const float* rewards, int F, int G, float* adv,
                                                             float* grp_mean, float* grp_std) {
    const int grp = (int)blockIdx.x, lane = lane_id();
    float r = 0.f;
    if (lane < G) for (int f = 0; f < F; ++f) r += rewards[((long)grp * G + lane) * F + f];
    const float mean = wave_sum<64>(lane < G ? r : 0.f) / (float)G;
    const float d = lane < G ? r - mean : 0.f;
    const float var = wave_sum<64>(d * d) / (float)(G - 1);   // torch.std default: unbiased (NaN for G == 1)
    const float sd = sqrtf(var);
    if (lane < G) adv[(long)grp * G + lane] = (r - mean) / (sd + 1e-4f);
    if (lane == 0) { if (grp_mean) grp_mean[grp] = mean; if (grp_std) grp_std[grp] = sd; }
}

// loss = mean_b( sum_c(ptl * mask) / sum_c(mask) ),  ptl = -min(r A, clip(r) A) + beta * k3
// outputs: out[0] loss, out[1] mean_kl, out[2] clip_ratio; dlogp [B, C] = d loss / d logp
__global__ __launch_bounds__(256) void grpo_loss_kernel(const float* logp, const float* old_logp, const float* ref_logp,
                                                        const float* adv, const int* mask, int B, int C, float eps_lo,
                                                        float eps_hi, float beta, float* out, float* dlogp) {
    __shared__ float red[4];
    const int tid = (int)threadIdx.x;
    float loss_acc = 0.f, kl_acc = 0.f, clip_num = 0.f, mask_tot = 0.f;
    for (int b = 0; b < B; ++b) {
        float ms = 0.f;
        for (int c = tid; c < C; c += 256) ms += (float)mask[(long)b * C + c];
        ms = block_sum<4>(ms, red);
        const float A = adv[b];
        float ls = 0.f, ks = 0.f, cs = 0.f;
        for (int c = tid; c < C; c += 256) {
            const long i = (long)b * C + c;
            const float lp = logp[i];
            const float ol = old_logp ? old_logp[i] : lp;
            const float mk = (float)mask[i];
            const float c1 = __expf(lp - ol);
            const float c2 = fminf(fmaxf(c1, 1.f - eps_lo), 1.f + eps_hi);
            const float l1 = c1 * A, l2 = c2 * A;
            float ptl = -fminf(l1, l2);
            // d(-min(l1,l2))/dlp ; d c1/dlp = c1 ; clamp passes gradient inside [lo, hi]; ties split evenly
            const float g1 = -A * c1;
            const float g2 = (c1 >= 1.f - eps_lo && c1 <= 1.f + eps_hi) ? -A * c1 : 0.f;
            float g = l1 < l2 ? g1 : (l1 > l2 ? g2 : 0.5f * (g1 + g2));
            if (beta != 0.f && ref_logp) {
                const float dl = ref_logp[i] - lp;
                const float e = __expf(dl);
                const float kl = e - dl - 1.f;
                ptl += beta * kl;
                g += beta * (1.f - e);
                ks += kl * mk;
            }
            ls += ptl * mk;
            cs += (l1 < l2 ? 1.f : 0.f) * mk;
            if (dlogp) dlogp[i] = ms > 0.f ? g * mk / (ms * (float)B) : 0.f;
        }
        ls = block_sum<4>(ls, red);
        ks = block_sum<4>(ks, red);
        cs = block_sum<4>(cs, red);
        loss_acc += ls / ms;
        kl_acc += ks / ms;
        clip_num += cs;
        mask_tot += ms;
    }
    if (tid == 0) {
        out[0] = loss_acc / (float)B;
        out[1] = kl_acc / (float)B;
        out[2] = clip_num / mask_tot;
    }
}

}  // namespace bra

using namespace bra;

extern "C" int bra_sample_ws_floats(int B, int top_k) {     // workspace size in 4-byte words (values + indices)
    const int k = top_k > 0 ? (top_k < 64 ? top_k : 64) : 64;
    return 2 * B * kSlices * k;
}

// `E` (optional, with x / ss): the wave that draws the token also gathers its embedding row into x [B, H] and writes the
// row's RMSNorm statistic (bra_row_sumsq layout) — the first two launches of the next decode step.  Needs the two-stage
// path (ws given, 4096 <= V <= 64 * 4096); returns BRA_ERR_UNSUPPORTED otherwise.
extern "C" int bra_sample_embed(const float* logits, long ldl, int B, int V, float temperature, int top_k, float top_p,
                                int do_sample, unsigned seed, const int* step_ptr, void* finished, int pad_id, int eos_id, int eos_id2,
                                int* out_ids, float* out_logp, int* tokens_out, long ldt, void* ws, const void* E, long lde,
                                int H, void* x, long ldx, float* ss, int nss, void* stream) {
    if (B == 0) return 0;
    if (!logits || !out_ids || V <= 0 || (do_sample && temperature <= 0.f)) return BRA_ERR_ARG;
    if (!(ws && V <= kSlices * 4096 && V >= 4096)) return BRA_ERR_UNSUPPORTED;
    if (do_sample && (top_k <= 0 || top_k > 64)) return BRA_ERR_UNSUPPORTED;   // the warpers run over <= 64 survivors
    if (E && (!x || H % 8 || lde % 8 || ldx % 8)) return BRA_ERR_ARG;
    const int k = do_sample ? (top_k > 0 ? (top_k < 64 ? top_k : 64) : 64) : 1;
    float* cv = (float*)ws;
    int* ci = (int*)ws + (long)B * kSlices * k;
    if ((V + kSlices - 1) / kSlices <= 2560) BRA_LAUNCH(topk_slices_kernel<10>, dim3(kSlices, B), dim3(256), 0, stream, logits, ldl, V, k, cv, ci);
    else BRA_LAUNCH(topk_slices_kernel<16>, dim3(kSlices, B), dim3(256), 0, stream, logits, ldl, V, k, cv, ci);
    int r = BRA_LAUNCH_STATUS();
    if (r) return r;
    BRA_LAUNCH(sample_merge_kernel, dim3(B), dim3(64), (size_t)kSlices * k * 8, stream, (const float*)cv, (const int*)ci, k,
               temperature, top_p, do_sample, (uint32_t)seed, step_ptr, (uint8_t*)finished, pad_id, eos_id, eos_id2, out_ids, out_logp,
               tokens_out, ldt, (const bf16_t*)E, lde, H, (bf16_t*)x, ldx, ss, nss);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_sample(const float* logits, long ldl, int B, int V, float temperature, int top_k, float top_p,
                          int do_sample, unsigned seed, const int* step_ptr, void* finished, int pad_id, int eos_id, int eos_id2,
                          int* out_ids, float* out_logp, int* tokens_out, long ldt, void* ws, void* stream) {
    if (B == 0) return 0;
    if (!logits || !out_ids || V <= 0 || (do_sample && temperature <= 0.f)) return BRA_ERR_ARG;
    if (do_sample && (top_k <= 0 || top_k > 64)) return BRA_ERR_UNSUPPORTED;       // the warpers run over <= 64 survivors
    if (ws && V <= kSlices * 4096 && V >= 4096)
        return bra_sample_embed(logits, ldl, B, V, temperature, top_k, top_p, do_sample, seed, step_ptr, finished, pad_id,
                                eos_id, eos_id2, out_ids, out_logp, tokens_out, ldt, ws, nullptr, 0, 0, nullptr, 0, nullptr, 0, stream);
    BRA_LAUNCH((sample_kernel<1024>), dim3(B), dim3(1024), 0, stream, logits, ldl, V, (const int*)nullptr, temperature, top_k, top_p,
               do_sample, (uint32_t)seed, step_ptr, (uint8_t*)finished, pad_id, eos_id, eos_id2, out_ids, out_logp, tokens_out, ldt);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_force_token(float* logits, long ldl, int B, int V, int token, const int* step_ptr, const int* at,
                               void* stream) {
    if (B == 0) return 0;
    if (!logits || !step_ptr || !at || token < 0 || token >= V || B > 64) return BRA_ERR_ARG;
    BRA_LAUNCH(force_token_kernel, dim3(1), dim3(64), 0, stream, logits, ldl, B, token, step_ptr, 0, at, (float*)nullptr, 0L);
    return BRA_LAUNCH_STATUS();
}

// bra_force_token for the tile-maxima sampler: also raises the maximum of the token's tile; `step_ptr` null -> the step index
// is the launch argument `step` (token loop issued launch by launch: no device-side counter)
extern "C" int bra_force_token_tiles(float* logits, long ldl, int B, int V, int token, const int* step_ptr, int step, const int* at,
                                     float* tmax, long ldm, void* stream) {
    if (B == 0) return 0;
    if (!logits || !at || token < 0 || token >= V || B > 64 || (tmax && ldm < (V + 15) / 16)) return BRA_ERR_ARG;
    BRA_LAUNCH(force_token_kernel, dim3(1), dim3(64), 0, stream, logits, ldl, B, token, step_ptr, step, at, tmax, ldm);
    return BRA_LAUNCH_STATUS();
}

// temperature -> top-k -> top-p -> multinomial for ANY top_k: 0 = no top-k filter (HF's "disabled"), > 64 as given (see
// sample_full_kernel: exact thresholds by bisection, a slow functional path; 1..64 is what bra_sample / bra_sample_tiles serve).
// `step_ptr` null: the step index is `step`.
extern "C" int bra_sample_full(const float* logits, long ldl, int B, int V, float temperature, int top_k, float top_p, unsigned seed,
                               const int* step_ptr, int step, void* finished, int pad_id, int eos_id, int eos_id2, int* out_ids,
                               float* out_logp, int* tokens_out, long ldt, void* stream) {
    if (B == 0) return 0;
    if (!logits || !out_ids || V <= 0 || temperature <= 0.f || top_k < 0 || !(top_p > 0.f)) return BRA_ERR_ARG;
    if (V >= (1 << 24)) return BRA_ERR_UNSUPPORTED;
    BRA_LAUNCH((sample_full_kernel<1024>), dim3(B), dim3(1024), 0, stream, logits, ldl, V, temperature, top_k, top_p > 1.f ? 1.f : top_p,
               (uint32_t)seed, step_ptr, step, (uint8_t*)finished, pad_id, eos_id, eos_id2, out_ids, out_logp, tokens_out, ldt);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_tile_max(const float* logits, long ldl, int B, int V, float* tmax, long ldm, void* stream) {
    if (B == 0) return 0;
    if (!logits || !tmax || V <= 0 || ldm < (V + 15) / 16) return BRA_ERR_ARG;
    const int ntiles = (V + 15) / 16;
    BRA_LAUNCH(tile_max_kernel, dim3((ntiles + 255) / 256, B), dim3(256), 0, stream, logits, ldl, V, tmax, ldm);
    return BRA_LAUNCH_STATUS();
}

// 0 = the two launches of round 4 (the product path); 1 = the one-launch form (sample_tiles_one_kernel, V <= 262 144), reachable
// through the debug library's bra_sample_set_one_launch.  Measured on one MI355X (profiles/r6_l_sampler_ab.txt, r6_m_sampler_trace.txt):
// the one launch takes 31.2 us against 9.6 + 18.2 us for the two (its extraction stage puts 16 waves of 64-bit compare / select
// rounds on 8 CUs where the two-launch form spreads them over 64), the token step is the same within 0.5 us (1.1858 / 1.1841 against
// 1.1862 / 1.1848 ms): the boundary it removes is worth what the concentration costs.  Kept for the A/B, not chosen.
static int g_sample_one_launch = 0;
#ifdef BRA_DEBUG
extern "C" int bra_sample_set_one_launch(int on) { g_sample_one_launch = on; return 0; }
#endif

// The sampler over tile maxima (see sample_tiles_kernel): logits fp32 [B, V] with tmax [B, ceil(V / 16)] = the maximum of every
// 16-column tile (lm_head epilogue of the decode step, or bra_tile_max).  Same tokens as bra_sample / bra_sample_embed for the
// same inputs.  ws: bra_sample_ws_floats words.  `step_ptr` null: the step index is `step`.  E / x / ss as bra_sample_embed.
// pos0 / pos_out / cosT / sinT / hd / rope_rows (optional): pos_out[b] = pos0[b] + step and rope_rows[b] = (cos | sin) row of that
// position — what bra_advance_counters would leave for the decode step that follows this draw.
// BRA_ERR_UNSUPPORTED: fewer than 4096 tiles' worth of vocabulary is fine, but V / 16 > 8 * 4096, top_k outside 1..64.
extern "C" int bra_sample_tiles(const float* logits, long ldl, const float* tmax, long ldm, int B, int V, float temperature, int top_k,
                                float top_p, int do_sample, unsigned seed, const int* step_ptr, int step, void* finished, int pad_id,
                                int eos_id, int eos_id2, int* out_ids, float* out_logp, int* tokens_out, long ldt, void* ws,
                                const void* E, long lde, int H, void* x, long ldx, float* ss, int nss, const int* pos0, int* pos_out,
                                const float* cosT, const float* sinT, int hd, float* rope_rows, void* stream) {
    if (B == 0) return 0;
    if (!logits || !tmax || !out_ids || !ws || V <= 0 || (do_sample && temperature <= 0.f)) return BRA_ERR_ARG;
    if (do_sample && (top_k <= 0 || top_k > 64)) return BRA_ERR_UNSUPPORTED;
    if (E && (!x || H % 8 || lde % 8 || ldx % 8)) return BRA_ERR_ARG;
    if (rope_rows && (!pos0 || !cosT || !sinT || hd <= 0 || hd % 2)) return BRA_ERR_ARG;
    if (pos_out && !pos0) return BRA_ERR_ARG;
    const int ntiles = (V + 15) / 16;
    if (ldm < ntiles) return BRA_ERR_ARG;
    if ((ntiles + kTileSlices - 1) / kTileSlices > 4096) return BRA_ERR_UNSUPPORTED;
    const int k = do_sample ? top_k : 1;
    if (g_sample_one_launch && ntiles <= 1024 * 16) {
#define BRA_ST1(EPT_, NU_)                                                                                                          \
        BRA_LAUNCH((sample_tiles_one_kernel<EPT_, NU_>), dim3(B), dim3(1024), 0, stream, logits, ldl, V, tmax, ldm, ntiles, k,        \
                   temperature, top_p, do_sample, (uint32_t)seed, step_ptr, step, (uint8_t*)finished, pad_id, eos_id, eos_id2,     \
                   out_ids, out_logp, tokens_out, ldt, (const bf16_t*)E, lde, H, (bf16_t*)x, ldx, ss, nss, pos0, pos_out, cosT,    \
                   sinT, hd, rope_rows)
        if (ntiles <= 1024 * 10) { if (k <= 20) BRA_ST1(10, 5); else BRA_ST1(10, 16); }
        else { if (k <= 20) BRA_ST1(16, 5); else BRA_ST1(16, 16); }
#undef BRA_ST1
        return BRA_LAUNCH_STATUS();
    }
    float* cv = (float*)ws;
    int* ci = (int*)ws + (long)B * kTileSlices * k;
    const int per = (ntiles + kTileSlices - 1) / kTileSlices;
    if (per <= 1280) BRA_LAUNCH(topk_slices_kernel<5>, dim3(kTileSlices, B), dim3(256), 0, stream, tmax, ldm, ntiles, k, cv, ci);
    else if (per <= 2560) BRA_LAUNCH(topk_slices_kernel<10>, dim3(kTileSlices, B), dim3(256), 0, stream, tmax, ldm, ntiles, k, cv, ci);
    else BRA_LAUNCH(topk_slices_kernel<16>, dim3(kTileSlices, B), dim3(256), 0, stream, tmax, ldm, ntiles, k, cv, ci);
    int r = BRA_LAUNCH_STATUS();
    if (r) return r;
#define BRA_ST(NL_, NU_)                                                                                                        \
    BRA_LAUNCH((sample_tiles_kernel<NL_, NU_>), dim3(B), dim3(64), 0, stream, logits, ldl, V, (const float*)cv, (const int*)ci,  \
               kTileSlices, k, temperature, top_p, do_sample, (uint32_t)seed, step_ptr, step, (uint8_t*)finished, pad_id, eos_id, \
               eos_id2, out_ids, out_logp, tokens_out, ldt, (const bf16_t*)E, lde, H, (bf16_t*)x, ldx, ss, nss, pos0, pos_out,    \
               cosT, sinT, hd, rope_rows)
    if (k <= 20) BRA_ST(3, 5);
    else BRA_ST(8, 16);
#undef BRA_ST
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_advance_counters(int* pos, int n, int* a, int* b, const float* cosT, const float* sinT, int hd, float* rope_rows,
                                    void* stream) {
    if (n < 0 || (n > 0 && !pos)) return BRA_ERR_ARG;
    if (rope_rows && (!cosT || !sinT || hd <= 0 || hd % 2)) return BRA_ERR_ARG;
    const long ne = rope_rows ? (long)n * hd : 0;
    const int nt = ne <= 64 ? 64 : (ne >= 1024 ? 1024 : (int)((ne + 63) / 64 * 64));
    BRA_LAUNCH(advance_counters_kernel, dim3(1), dim3(nt), 0, stream, pos, n, a, b, cosT, sinT, hd, rope_rows, nt);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_rope_rows(const float* cosT, const float* sinT, const int* pos, int n, int hd, float* rope_rows, void* stream) {
    if (n <= 0) return 0;
    if (!cosT || !sinT || !pos || !rope_rows || hd <= 0 || hd % 2) return BRA_ERR_ARG;
    BRA_LAUNCH(rope_rows_kernel, dim3(1), dim3(64), 0, stream, cosT, sinT, pos, n, hd, rope_rows);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_eos_mask(const int* ids, int B, int C, int eos_id, int* mask, int* lengths, void* stream) {
    if (B == 0 || C == 0) return 0;
    if (!ids || !mask) return BRA_ERR_ARG;
    BRA_LAUNCH(eos_mask_kernel, dim3(B), dim3(64), 0, stream, ids, C, eos_id, mask, lengths);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_group_advantage(const float* rewards, int N, int F, int G, float* adv, float* grp_mean,
                                   float* grp_std, void* stream) {
    if (N == 0) return 0;
    if (!rewards || !adv || G <= 0 || G > 64 || N % G || F <= 0) return BRA_ERR_ARG;
    BRA_LAUNCH(group_advantage_kernel, dim3(N / G), dim3(64), 0, stream, rewards, F, G, adv, grp_mean, grp_std);
    return BRA_LAUNCH_STATUS();
}

extern "C" int bra_grpo_loss(const float* logp, const float* old_logp, const float* ref_logp, const float* adv,
                             const int* mask, int B, int C, float eps_lo, float eps_hi, float beta, float* out3,
                             float* dlogp, void* stream) {
    if (B <= 0 || C <= 0 || !logp || !adv || !mask || !out3) return BRA_ERR_ARG;
    BRA_LAUNCH(grpo_loss_kernel, dim3(1), dim3(256), 0, stream, logp, old_logp, ref_logp, adv, mask, B, C, eps_lo,
               eps_hi, beta, out3, dlogp);
    return BRA_LAUNCH_STATUS();
}
