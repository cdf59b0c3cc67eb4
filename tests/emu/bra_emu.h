// bra_emu.h — TEST INFRASTRUCTURE ONLY.
//
// A CPU executor for the kernel sources under bioreason_amd/csrc/: the SAME
// .hip files are compiled with the host clang++ and -DBRA_EMU, and every
// workgroup is run as a set of cooperative fibers (one per work-item) so that
// __syncthreads(), wave64 shuffles and MFMA fragments behave as on gfx950.
// It exists so that the index arithmetic of the HIP kernels can be exercised
// by `pytest -m "not gpu"` on a box without a GPU.  It is never loaded by the
// product package (bioreason_amd/_lib.py only opens libbioreason_hip.so) and
// nothing measured or shipped runs through it.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <string.h>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace bra_emu {
struct Idx { unsigned x, y, z; };
extern thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

void launch(dim3 grid, dim3 block, size_t dyn_smem, const std::function<void()>& body);
void block_sync();
void wave_sync();
int lane_id();                      // linear thread id & 63
int wave_live_lanes();
// wave exchange: every live lane of the wave deposits n 32-bit words, then may
// read any lane's words after the call returns.  Returned pointer is valid
// until the lane's next exchange.
const uint32_t* wave_exchange(const uint32_t* mine, int n);   // -> base[lane*16 + i]
char* dyn_smem();
}  // namespace bra_emu

#define threadIdx (bra_emu::t_threadIdx)
#define blockIdx (bra_emu::t_blockIdx)
#define blockDim (bra_emu::t_blockDim)
#define gridDim (bra_emu::t_gridDim)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

static inline void __syncthreads() { bra_emu::block_sync(); }

static inline float atomicAdd(float* p, float v) {
    uint32_t* ip = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
    for (;;) {
        float f;
        memcpy(&f, &old, 4);
        float nf = f + v;
        uint32_t nu;
        memcpy(&nu, &nf, 4);
        if (__atomic_compare_exchange_n(ip, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
    }
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
