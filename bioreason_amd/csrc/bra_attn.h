// bra_attn.h — declarations shared by the attention kernel files (k_attn.hip: the 8-wave forward, the backward kernels, decode;
// k_attn4.hip: the 4-wave forward with 64 queries per wave).
#pragma once
#include "bra_device.h"

namespace bra {

constexpr float kNeg = -1.0e30f;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct AttnArgs {
    const bf16_t* q;  long q_sb, q_ss, q_sh;     // [B, Sq, Hq, hd] via strides
    const bf16_t* k;  long k_sb, k_ss, k_sh;     // [B, Sk, Hkv, hd] via strides
    const bf16_t* v;  long v_sb, v_ss, v_sh;     // (backward only)
    const bf16_t* vt; long vt_sb, vt_sh, vt_sd;  // V^T [B, Hkv, hd, pitch]
    const bf16_t* kt; long kt_sb, kt_sh, kt_sd;  // K^T (dQ kernel)
    const bf16_t* qt; long qt_sb, qt_sh, qt_sd;  // Q^T [B, Hq, hd, pitch] (dKV kernel)
    const bf16_t* dot; long dot_sb, dot_sh, dot_sd;  // dO^T [B, Hq, hd, pitch] (dKV kernel)
    const bf16_t* dout; long do_sb, do_ss, do_sh;    // dO
    bf16_t* o;        long o_sb, o_ss, o_sh;
    bf16_t* dq;       long dq_sb, dq_ss, dq_sh;
    bf16_t* dk;       long dk_sb, dk_ss, dk_sh;
    bf16_t* dv;       long dv_sb, dv_ss, dv_sh;
    float* lse;                                  // [B, Hq, Sq] natural-log LSE of the scaled scores
    const float* delta;                          // [B, Hq, Sq] rowsum(dO * O)
    const uint8_t* kmask;                        // [B, Sk] 1 = key may be attended, or null
    int B, Hq, Hkv, Sq, Sk;
    int causal, q_off;                           // causal: key j visible to query i iff j <= i + q_off
    int legacy_order;                            // A/B knob (bra_attn_set_block_order): block index fastest, as rounds 1-3 launched
    // forward with the key range cut into `nsplit` parts (grids that cannot fill the chip: one prompt, the 256-query completion
    // segment): part s of a query block visits its tiles [ntile s / nsplit, ntile (s + 1) / nsplit) and leaves the unnormalised
    // O (fp32) and (running max, sum) per query; attn_combine_kernel merges the parts in order
    int nsplit;
    float* part_o;                               // [B, Hq, nsplit, Sq, hd]
    float* part_ml;                              // [B, Hq, nsplit, Sq, 2]
    // backward: the dQ kernel splits its key range the same way (`nsplit`, fp32 parts in `part_o`), the dK / dV kernel its loop
    // over (q-head, query tile) iterations (`nsplit_kv`, parts [B, Hkv, nsplit_kv, Sk, hd]); attn_sum_parts_kernel adds the parts in order
    int nsplit_kv;
    float* part_dk;
    float* part_dv;
    float scale;
    BRA_DBG_FIELD(unsigned long long* probe;)    // k_attn4.hip, debug build: cycle stamps of one workgroup's hot loop (bra_attn_set_probe)
};

template <int HD>
struct Tile {
    static constexpr int CH = HD / 8;            // 16-byte chunks per K row
    static constexpr int RPB = 16 / CH > 0 ? 16 / CH : 1;
    static constexpr int DB = HD / 32;           // 32-wide d blocks
    static constexpr int DS = HD / 16;           // 16-deep contraction steps over d
    static constexpr int KBYTES = 64 * HD * 2;   // [64 rows][HD]   (row = key or query)
    static constexpr int TBYTES = HD * 64 * 2;   // [HD rows][64]   (transposed image)
    __device__ static __forceinline__ int koff(int row, int chunk) {   // byte offset in a [64][HD] tile
        return row * (HD * 2) + ((chunk ^ ((row / RPB) & (CH - 1))) << 4);
    }
    __device__ static __forceinline__ int toff(int d, int chunk) {     // byte offset of 16-byte unit `chunk` of row d in a [HD][64] tile
        return d * 128 + ((chunk ^ ((d >> 1) & 7)) << 4);
    }
};

// full-rate 24-bit multiply (v_mul_u32_u24): row index x stride, both below 2^24 (host-checked)
#ifdef BRA_EMU
__device__ __forceinline__ unsigned attn_mul24(int a, int b) { return ((unsigned)a & 0xffffffu) * ((unsigned)b & 0xffffffu); }
#else
__device__ __forceinline__ unsigned attn_mul24(int a, int b) { return __umul24((unsigned)a, (unsigned)b); }
#endif

// sequence index (within a 32-block) held in register r of a 32x32 C/D fragment
__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ uint64_t key_valid_word(const AttnArgs& a, int b, int kv0, int lane) {
    int kj = kv0 + lane;
    bool ok = kj < a.Sk;
    if (ok && a.kmask) ok = a.kmask[(long)b * a.Sk + kj] != 0;
    return wave_ballot(ok);
}


// Workgroup -> (sequence block, head, batch row) in DISPATCH order (x fastest, then y, z): the sequence block is the SLOWEST
// coordinate and, under a causal mask, blocks are taken heaviest first — query block i of a causal pass visits i + 1 key tiles
// (key block j is visited by the query tiles behind it), so with the block index fastest the heaviest workgroup of the last
// (batch, head) pair started last and the chip idled behind it (B = 8, S = 2436: ~87 tile-units of makespan for 49 of work per slot).
// Neighbouring workgroups are the heads of one batch row: the q-heads of a kv-group still share their K / V tiles in L2.
__device__ __forceinline__ void attn_block_coords(int legacy, int heavy_is_last, int& blk, int& head, int& b) {
    if (legacy) { blk = (int)blockIdx.x; head = (int)blockIdx.y; b = (int)blockIdx.z; return; }
    const int nblk = (int)gridDim.x, nh = (int)gridDim.y, nb = (int)gridDim.z;
    const int id = (int)blockIdx.x + nblk * ((int)blockIdx.y + nh * (int)blockIdx.z);
    const int per = nh * nb;
    const int x = id / per, rem = id - x * per;
    blk = heavy_is_last ? nblk - 1 - x : x;
    b = rem / nh;
    head = rem - b * nh;
}


// k_attn4.hip: forward for grids of whole 256-query workgroups (no key split): 4 waves, one per SIMD, 64 queries per wave, every
// wave with its own software pipeline over 32-key steps
template <int HD> int launch_fwd4(const AttnArgs& a, bra_stream_t st);
extern template int launch_fwd4<128>(const AttnArgs&, bra_stream_t);
extern template int launch_fwd4<64>(const AttnArgs&, bra_stream_t);

// k_attn4b.hip: the dQ kernel in the same structure (unit = 32-key step x query block)
template <int HD> int launch_dq4(const AttnArgs& a, bra_stream_t st);
extern template int launch_dq4<128>(const AttnArgs&, bra_stream_t);
extern template int launch_dq4<64>(const AttnArgs&, bra_stream_t);
// ... and dK / dV (two launches: dV, then dK), 256 keys per workgroup
template <int HD> int launch_dkv4(const AttnArgs& a, bra_stream_t st);
extern template int launch_dkv4<128>(const AttnArgs&, bra_stream_t);
extern template int launch_dkv4<64>(const AttnArgs&, bra_stream_t);

}  // namespace bra
