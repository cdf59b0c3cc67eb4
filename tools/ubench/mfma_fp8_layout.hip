// Operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (e4m3) operands, found by experiment (the ISA tables are not on disk):
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_fp8_layout.hip -o tools/ubench/mfma_fp8_layout && ./mfma_fp8_layout   (GPU box)
// 1. A one-hot at (lane, byte), B all ones -> which C row lights up;  B one-hot, A all ones -> which C column
// 2. A one-hot at (la, ja) x B one-hot at (lb, jb) -> C != 0 iff the two positions carry the same k
// 3. the scale operand: the E8M0 byte (opsel 0) of each lane scales that lane's 32 elements (2^(s - 127))
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void one(const unsigned char* a, const unsigned char* b, float* c, int sa, int sb) {
    const int l = threadIdx.x;
    v8i av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = ((const int*)a)[l * 8 + i]; bv[i] = ((const int*)b)[l * 8 + i]; }
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) c[l * 4 + r] = acc[r];
}
// block = A position (lane la = blockIdx.x >> 5, byte ja = blockIdx.x & 31); loops over all B positions; match[pa][pb] = any C != 0
__global__ void pairs(unsigned char* match) {
    const int l = threadIdx.x, pa = blockIdx.x, la = pa >> 5, ja = pa & 31;
    v8i av = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l == la) av[ja >> 2] = 0x38 << (8 * (ja & 3));
    for (int pb = 0; pb < 2048; ++pb) {
        const int lb = pb >> 5, jb = pb & 31;
        v8i bv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (l == lb) bv[jb >> 2] = 0x38 << (8 * (jb & 3));
        v4f acc = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        const bool nz = acc[0] != 0.f || acc[1] != 0.f || acc[2] != 0.f || acc[3] != 0.f;
        const unsigned long long m = __ballot(nz);
        if (l == 0) match[(size_t)pa * 2048 + pb] = m ? 1 : 0;
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main() {
    unsigned char *da, *db, *dm; float* dc;
    CK(hipMalloc(&da, 2048)); CK(hipMalloc(&db, 2048)); CK(hipMalloc(&dc, 256 * 4)); CK(hipMalloc(&dm, 2048 * 2048));
    std::vector<unsigned char> ha(2048), hb(2048); std::vector<float> hc(256);
    // 1. rows of A positions / columns of B positions; C layout assumed col = lane & 15, row = 4 (lane >> 4) + reg
    int bad_row = 0, bad_col = 0;
    for (int side = 0; side < 2; ++side)
        for (int p = 0; p < 2048; ++p) {
            std::fill(ha.begin(), ha.end(), side ? 0x38 : 0); std::fill(hb.begin(), hb.end(), side ? 0 : 0x38);
            (side ? hb : ha)[p] = 0x38;
            CK(hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice));
            one<<<1, 64>>>(da, db, dc, 0x7f7f7f7f, 0x7f7f7f7f);
            CK(hipMemcpy(hc.data(), dc, 1024, hipMemcpyDeviceToHost));
            // which rows / columns are non-zero
            int rows = 0, cols = 0, r0 = -1, c0 = -1;
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hc[l * 4 + r] != 0.f) {
                const int row = 4 * (l >> 4) + r, col = l & 15;
                if (!(rows >> row & 1)) { rows |= 1 << row; r0 = row; }
                if (!(cols >> col & 1)) { cols |= 1 << col; c0 = col; }
            }
            const int lane = p >> 5;
            if (side == 0) { if (__builtin_popcount(rows) != 1 || r0 != (lane & 15) || cols != 0xffff) { if (bad_row++ < 8) printf("A pos lane %d byte %d -> rows %04x cols %04x\n", lane, p & 31, rows, cols); } }
            else { if (__builtin_popcount(cols) != 1 || c0 != (lane & 15) || rows != 0xffff) { if (bad_col++ < 8) printf("B pos lane %d byte %d -> rows %04x cols %04x\n", lane, p & 31, rows, cols); } }
        }
    printf("A position (lane, byte) -> C row lane & 15: %s (%d exceptions)\n", bad_row ? "NO" : "yes", bad_row);
    printf("B position (lane, byte) -> C col lane & 15 (C: col = lane & 15, row = 4 (lane >> 4) + reg): %s (%d exceptions)\n", bad_col ? "NO" : "yes", bad_col);
    // 2. k pairing
    pairs<<<2048, 64>>>(dm);
    std::vector<unsigned char> hm((size_t)2048 * 2048);
    CK(hipMemcpy(hm.data(), dm, hm.size(), hipMemcpyDeviceToHost));
    int bad_pair = 0;
    for (int pa = 0; pa < 2048; ++pa) for (int pb = 0; pb < 2048; ++pb) {
        const bool want = ((pa >> 5) >> 4) == ((pb >> 5) >> 4) && (pa & 31) == (pb & 31);
        if ((hm[(size_t)pa * 2048 + pb] != 0) != want && bad_pair++ < 16)
            printf("pair A (lane %d, byte %d) x B (lane %d, byte %d): match %d, hypothesis %d\n", pa >> 5, pa & 31, pb >> 5, pb & 31, hm[(size_t)pa * 2048 + pb], (int)want);
    }
    printf("A (la, ja) pairs with B (lb, jb) iff la >> 4 == lb >> 4 and ja == jb: %s (%d exceptions)\n", bad_pair ? "NO" : "yes", bad_pair);
    // 3. values and scales: A = small integers, B = small integers, C against the host; then the scale bytes
    srand(7);
    const unsigned char enc[5] = {0x00, 0x38, 0x40, 0xB8, 0xC0};   // 0, 1, 2, -1, -2
    const float dec[5] = {0.f, 1.f, 2.f, -1.f, -2.f};
    std::vector<float> fa(2048), fb(2048);
    for (int p = 0; p < 2048; ++p) { int i = rand() % 5, j = rand() % 5; ha[p] = enc[i]; fa[p] = dec[i]; hb[p] = enc[j]; fb[p] = dec[j]; }
    CK(hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice));
    for (int sc = 0; sc < 3; ++sc) {
        const int sa = sc == 0 ? 0x7f7f7f7f : (sc == 1 ? 0x7f7f7f80 : 0x7f7f7f7f), sb = sc == 2 ? 0x7f7f7f7d : 0x7f7f7f7f;
        const float mul = sc == 0 ? 1.f : (sc == 1 ? 2.f : 0.25f);
        one<<<1, 64>>>(da, db, dc, sa, sb);
        CK(hipMemcpy(hc.data(), dc, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int row = 4 * (l >> 4) + r, col = l & 15;
            float want = 0.f;
            for (int g = 0; g < 4; ++g) for (int j = 0; j < 32; ++j) want += fa[(16 * g + row) * 32 + j] * fb[(16 * g + col) * 32 + j];
            if (hc[l * 4 + r] != want * mul && bad++ < 4) printf("scale case %d: C[%d][%d] = %g, want %g\n", sc, row, col, hc[l * 4 + r], want * mul);
        }
        printf("values, scale A byte0 = 0x%02x, scale B byte0 = 0x%02x: C = %g x (A B^T) %s\n", sa & 0xff, sb & 0xff, mul, bad ? "NO" : "exact");
    }
    return 0;
}
