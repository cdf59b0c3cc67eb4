/* bioreason_hip_debug.h — diagnostics and experiments of libbioreason_hip_debug.so (the same sources built with -DBRA_DEBUG:
 * `make -C bioreason_amd/csrc debug`).  NOT part of the product ABI (bioreason_hip.h): nothing here is needed to run the hot path,
 * and everything here is either process-wide mutable state (the knobs), a timing probe, or the opt-in persistent decode step that
 * measured slower than the launched kernels.  Used by tests/ (pinning a tile variant so that every variant is
 * compared with every other), tools/ (probes) and bench.py's A/B flags.
 *
 * The knobs are held in atomics (a concurrent launch sees the old or the new value, never a torn one) and select between
 * bit-identical tilings: results do not depend on them.
 */
#ifndef BIOREASON_HIP_DEBUG_H
#define BIOREASON_HIP_DEBUG_H
#include "bioreason_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- tile-variant knobs (k_gemm.hip, k_attn.hip) ------------------------------------------------------------------------ */
/* tuning knob for A/B measurements: pins the tile variant of bra_gemm_* (0-3: register-staged 128/256-row tiles x prefetch
 * depth, 4/5: 256 x 128 LDS-DMA kernel without / with skewed fragment reads, 6 / 7: 256 x 256 ring kernel with four / two phases
 * per K-tile, 9 / 10: the LDS-DMA kernel at 192 / 128-row tiles, 11 - 14: the four-wave kernel with 80 x 128 / 64 x 128 / 80 x 64 /
 * 64 x 64 per-wave tiles = macro tiles 160 x 256 / 128 x 256 / 160 x 128 / 128 x 128); v = -1 restores the built-in per-shape
 * choice (a cost model over all of them), v = -2 the same choice WITHOUT the four-wave kernel (rounds 1 - 4a, for A/B runs).
 * Process-wide: meant for benchmarks and tests, not for concurrent callers. */
int bra_gemm_set_variant(int v);
/* minimum fill (percent of 256 CUs busy, averaged over the rounds of 256 x 256 tiles) at which the per-shape choice takes the
 * ring kernel (default 60; 75 until round 6) */
int bra_gemm_set_ring_fill(int pct);
/* row split of the per-shape choice (default on): when the last round of 256 x 256 tiles would be less than half full, the
 * tile-rows that fill whole rounds go to the ring kernel and the remaining rows to the 256 x 128 kernel (two launches) */
int bra_gemm_set_row_split(int on);
/* interior bf16 epilogue of the tiled GEMMs through LDS (full 128-byte lines per row; default on) or as direct 8-byte stores (0): same
 * values bit for bit, A/B timing */
int bra_gemm_set_epi_lds(int on);
/* bra_sample_tiles as ONE launch (sample_tiles_one_kernel; 1) or as the two launches of round 4 (0, the default): same tokens, A/B timing */
int bra_sample_set_one_launch(int on);
/* tile height of the LDS-DMA kernel (round 4): 0 = chosen per call (256 / 192 / 128 rows, whichever fills the 256 CUs best),
 * 256 = rounds 1-3 (A/B measurements); bra_gemm_set_variant(9 / 10) pins 192 / 128 together with the kernel */
int bra_gemm_set_glds_rows(int rows);
/* attention forward / backward: 1 = launch order of rounds 1-3 (query / key block index fastest), 0 (default) = block index
 * slowest and, under a causal mask, heaviest blocks first (A/B measurements) */
int bra_attn_set_block_order(int legacy);
/* attention forward for grids of whole 256-query workgroups: 1 (default) = the pipelined kernel of k_attn4.hip (4 waves of 64
 * queries), 0 = the 8-wave kernel of rounds 1-5 for every shape (A/B measurements) */
int bra_attn_set_fwd4(int on);
/* attention backward: bit 0 = the pipelined dQ kernel, bit 1 = the pipelined dK / dV kernels of k_attn4b.hip for grids of whole
 * 256-row workgroups (default: both set), 0 = the kernels of rounds 1-5 for every shape (A/B measurements) */
int bra_attn_set_bwd4(int mask);
/* 80 x 8-byte device buffer (or null) that one mid-sequence workgroup of the following 4-wave forward launches fills per wave with
 * cycle counts of its hot loop: DMA wait, barrier, step 1, step 2, loop tail, iterations, whole kernel, steps (diagnostics) */
int bra_attn_set_probe(void* buf);


/* ---- timing probes of the token loop (k_decgemm.hip, k_decfused.hip) ---------------------------------------------------- */
/* bra_dec_gemm2_packed with a timing probe: `probe` (device, 16 x 8 bytes or null) receives 100 MHz wall-clock stamps of the first
 * and the last workgroup — kernel entry, requests issued, products ready, barrier passed, epilogue issued (diagnostics) */
int bra_dec_gemm2_probe(const void* x, long ldx, const float* ss_in, int nss_in, const void* norm_w, float eps, const void* W,
                        long ldw, const void* res, long ldres, void* out, long ldo, float* ss_out, int nss_out, int M, int N,
                        int K, int act, int out_f32, int packed, void* probe, void* stream);
/* diagnostics: 16 x 8-byte device buffer that bra_dec_attn_both stamps with the 100 MHz wall clock (one prompt-part wave:
 * entry, requests issued, q arrived, q rotated, scores, partials issued; one completion-part wave: entry, prologue, scores,
 * end); null switches the probe off.  Process-wide. */
int bra_debug_set_probe(void* p);

/* ---- persistent-grid building blocks (k_persist.hip, bra_gridsync.h) -------------------------------------------------
 * In-launch grid barrier + write-through hand-off used by the persistent decode step (the body of HF's `_sample` loop,
 * TF:generation/utils.py:2876-2925, kept inside one launch).  bra_gridsync_bytes: size of the synchronisation record the
 * caller provides.  bra_gridbar_probe: `iters` x {publish 128 B per workgroup, grid barrier, read and word-check every slot}
 * on `nwg` resident workgroups (mode 0 barrier only, 1 sc1 write-through protocol, 2 fence protocol, 3 protocol 1 with
 * `wchunks` x 16 B per thread of a read-once stream prefetched across the barrier); errs[0] = mismatching words, errs[1] =
 * barriers completed; every spin is bounded by `timeout_us` (the record's error word is set instead of hanging). */
int bra_gridsync_bytes(void);
/* All decoder layers of one shared-prefix decode step (bra_qwen_decode_step_one's layer loop: Qwen3DecoderLayer.forward with a
 * KV cache, TF:qwen3:294-323) in ONE launch of one workgroup per CU: six phases per layer separated by in-launch grid barriers,
 * the same tiles / K split / reduction order / epilogues as the launched kernels (bit-identical results), the weights of the
 * following phases requested before the barrier that hands over their activations (prefetch 1; 2: also the K / V^T chunk of the
 * wave's attention item).  layers_dev: HOST array of L <= 40 records of bra_persist_layer_desc_size() bytes {Wqkv, Wo, Wgu, Wd
 * (fragment-packed, norms folded), qn, kn, kp, vtp, kc, vct} (device pointers; the table is copied into the kernel arguments).  x / ss_ws: embedded token rows + statistics in, last layer's
 * output + statistics out.  sync: bra_gridsync_bytes() bytes (zeroed by the call).  Every spin is bounded by timeout_us; a
 * timed-out launch leaves a non-zero word at sync + 1088 (GridSync::err).  BRA_ERR_UNSUPPORTED: shape not instantiated, more
 * than 8 sequences, fewer CUs than workgroups. */
int bra_persist_layer_desc_size(void);
/* diagnostics knob (process-wide, atomic): device buffer of 6 L x 4 100-MHz wall-clock stamps of workgroup 0 filled by the following
 * persistent launches (per phase: start, -, stores issued, stores drained); null turns it off */
int bra_persist_set_stamps(void* p);
int bra_qwen_layers_persist(const void* layers_dev, int L, int R, int copies, int H, int Hq, int Hkv, int hd, int F, int P,
                            long vt_pitch, int C, long cp, float eps, float scale, const float* cosT, const float* sinT,
                            const int* pos, const float* rope_rows, const void* pmask, int t, const int* t_dev, void* x,
                            void* qkv, void* o, void* h, void* act, float* ss_ws, int nss, float* part_o, float* part_ml,
                            int nslot, void* sync, int prefetch, int stop_after, int timeout_us, void* stream);
/* bra_qwen_decode_step_one with its layer loop replaced by bra_qwen_layers_persist: embed + statistics (unless embed_done), ONE
 * launch for all decoder layers, lm_head.  layers_host as for bra_qwen_decode_step_one (packed + folded weights required),
 * layers_dev / sync / prefetch / stop_after / timeout_us as for bra_qwen_layers_persist.  Bit-identical logits. */
int bra_qwen_decode_step_persist(const void* layers_host, const void* layers_dev, int L, int R, int copies, int H, int Hq, int Hkv,
                                 int hd, int F, int P, long vt_pitch, int C, long cp, int V, float eps, float scale, const void* E,
                                 const void* norm_w, const float* cosT, const float* sinT, const int* tok, const int* pos,
                                 const void* pmask, int t, const int* t_dev, int embed_done, void* x, void* qkv, void* o, void* h,
                                 void* act, float* ss_ws, int nss, float* part_o, float* part_ml, int nslot, float* logits,
                                 void* sync, int prefetch, int stop_after, int timeout_us, void* stream);
/* probe: n dependent kernels that each stream `wchunks` x 16 B per thread of read-once data, as ordinary launches on one stream
 * (chained 0) or alternating two streams with device-side waits on per-kernel completion counters (chained 1): see k_persist.hip */
int bra_chain_probe(void* done, void* buf, void* errs, const void* wts, long wts_bytes, int nwg, int n, int chained, int wchunks,
                    int timeout_us, void* stream_a, void* stream_b);
int bra_gridbar_probe(void* sync, void* buf, void* errs, const void* wts, long wts_bytes, int nwg, int iters, int mode,
                      int wchunks, int timeout_us, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BIOREASON_HIP_DEBUG_H */
