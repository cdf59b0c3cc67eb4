// bra_emu.cpp — TEST INFRASTRUCTURE ONLY (see bra_emu.h).
// Cooperative-fiber executor: one fiber per work-item, workgroups spread over
// host threads.  Barriers and wave64 exchanges are generation counters; a
// fiber that cannot proceed yields to the round-robin scheduler.
#include "bra_emu.h"

#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#if !defined(__x86_64__)
#error "bra_emu's fiber switch is written for x86-64 (System V ABI)"
#endif

// Fiber switch: callee-saved registers + stack pointer, nothing else (glibc's swapcontext also saves the signal mask — one
// rt_sigprocmask system call per switch, and a kernel under this executor switches fibers at every barrier and wave exchange).
extern "C" void bra_emu_switch(void** save_sp, void* const* load_sp);
__asm__(
    ".text\n"
    ".globl bra_emu_switch\n"
    ".type bra_emu_switch,@function\n"
    "bra_emu_switch:\n"
    "    pushq %rbp\n"
    "    pushq %rbx\n"
    "    pushq %r12\n"
    "    pushq %r13\n"
    "    pushq %r14\n"
    "    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq (%rsi), %rsp\n"
    "    popq %r15\n"
    "    popq %r14\n"
    "    popq %r13\n"
    "    popq %r12\n"
    "    popq %rbx\n"
    "    popq %rbp\n"
    "    ret\n"
    ".size bra_emu_switch, .-bra_emu_switch\n");

namespace bra_emu {

thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

namespace {
constexpr int kMaxThreads = 1024;
constexpr int kMaxWaves = kMaxThreads / 64;
constexpr size_t kStackBytes = 256 * 1024;
constexpr int kXWords = 16;  // 32-bit words per lane per exchange

struct Fiber {
    void* sp = nullptr;          // saved stack pointer while the fiber is switched out
    char* stack = nullptr;
    bool done = false;
    unsigned coll = 0;  // number of wave exchanges this lane has done
};

struct Worker {
    void* sched = nullptr;       // saved stack pointer of the scheduler loop
    std::vector<Fiber> fibers;
    int nthreads = 0;
    int cur = 0;
    int alive = 0;
    int block_arrived = 0;
    uint64_t block_gen = 0;
    int wave_alive[kMaxWaves];
    int wave_arrived[kMaxWaves];
    uint64_t wave_gen[kMaxWaves];
    uint32_t* xbuf = nullptr;  // [2][kMaxWaves][64][kXWords]
    char* smem = nullptr;
    size_t smem_cap = 0;
    const std::function<void()>* body = nullptr;
    uint64_t progress = 0;
};

thread_local Worker* W = nullptr;

void yield_to_sched() {
    Worker* w = W;
    bra_emu_switch(&w->fibers[w->cur].sp, &w->sched);
}

// fiber stacks are kept for the life of the process (a launch of 512-thread workgroups would otherwise map and fault 128 MB per worker)
std::mutex g_stack_mu;
std::vector<char*> g_stack_pool;
char* stack_get() {
    {
        std::lock_guard<std::mutex> lk(g_stack_mu);
        if (!g_stack_pool.empty()) { char* p = g_stack_pool.back(); g_stack_pool.pop_back(); return p; }
    }
    return (char*)aligned_alloc(64, kStackBytes);
}
void stack_put(char* p) {
    std::lock_guard<std::mutex> lk(g_stack_mu);
    g_stack_pool.push_back(p);
}

void fiber_entry() {
    Worker* w = W;
    (*w->body)();
    Fiber& f = w->fibers[w->cur];
    f.done = true;
    w->alive--;
    w->wave_alive[w->cur / 64]--;
    w->progress++;
    // a pending block barrier may now be complete
    if (w->alive > 0 && w->block_arrived == w->alive) {
        w->block_arrived = 0;
        w->block_gen++;
    }
    int wv = w->cur / 64;
    if (w->wave_alive[wv] > 0 && w->wave_arrived[wv] == w->wave_alive[wv]) {
        w->wave_arrived[wv] = 0;
        w->wave_gen[wv]++;
    }
    bra_emu_switch(&f.sp, &w->sched);
    abort();                     // (a finished fiber is never resumed)
}

void set_thread_idx(Worker* w, int lin) {
    unsigned bx = t_blockDim.x, by = t_blockDim.y;
    t_threadIdx.x = lin % bx;
    t_threadIdx.y = (lin / bx) % by;
    t_threadIdx.z = lin / (bx * by);
    (void)w;
}

void run_block(Worker* w) {
    int T = w->nthreads;
    w->alive = T;
    w->block_arrived = 0;
    w->block_gen = 0;
    for (int i = 0; i < kMaxWaves; ++i) {
        w->wave_alive[i] = 0;
        w->wave_arrived[i] = 0;
        w->wave_gen[i] = 0;
    }
    for (int i = 0; i < T; ++i) w->wave_alive[i / 64]++;
    for (int i = 0; i < T; ++i) {
        Fiber& f = w->fibers[i];
        f.done = false;
        f.coll = 0;
        // first switch into the fiber: six zeroed callee-saved registers are popped, `ret` enters fiber_entry with the stack
        // aligned as after a call (one dummy return-address slot below the 16-byte aligned top)
        uint64_t* sp = (uint64_t*)(((uintptr_t)f.stack + kStackBytes) & ~(uintptr_t)15);
        *--sp = 0;
        *--sp = (uint64_t)(uintptr_t)&fiber_entry;
        for (int r = 0; r < 6; ++r) *--sp = 0;
        f.sp = sp;
    }
    while (w->alive > 0) {
        uint64_t before = w->progress;
        for (int i = 0; i < T; ++i) {
            if (w->fibers[i].done) continue;
            w->cur = i;
            set_thread_idx(w, i);
            bra_emu_switch(&w->sched, &w->fibers[i].sp);
        }
        if (w->progress == before && w->alive > 0) {
            fprintf(stderr, "bra_emu: deadlock in block (%u,%u,%u): %d fibers alive, %d at barrier\n",
                    t_blockIdx.x, t_blockIdx.y, t_blockIdx.z, w->alive, w->block_arrived);
            abort();
        }
    }
}

}  // namespace

void block_sync() {
    Worker* w = W;
    uint64_t my = w->block_gen;
    w->progress++;
    if (++w->block_arrived == w->alive) {
        w->block_arrived = 0;
        w->block_gen++;
        return;
    }
    while (w->block_gen == my) yield_to_sched();
}

void wave_sync() {
    Worker* w = W;
    int wv = w->cur / 64;
    uint64_t my = w->wave_gen[wv];
    w->progress++;
    if (++w->wave_arrived[wv] == w->wave_alive[wv]) {
        w->wave_arrived[wv] = 0;
        w->wave_gen[wv]++;
        return;
    }
    while (w->wave_gen[wv] == my) yield_to_sched();
}

int lane_id() { return W->cur & 63; }
int wave_live_lanes() { return W->wave_alive[W->cur / 64]; }

const uint32_t* wave_exchange(const uint32_t* mine, int n) {
    Worker* w = W;
    Fiber& f = w->fibers[w->cur];
    int wv = w->cur / 64, lane = w->cur & 63;
    int par = f.coll & 1;
    f.coll++;
    uint32_t* base = w->xbuf + ((size_t)(par * kMaxWaves + wv) * 64) * kXWords;
    for (int i = 0; i < n; ++i) base[lane * kXWords + i] = mine[i];
    wave_sync();
    return base;
}

char* dyn_smem() { return W->smem; }


void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body) {
    int T = (int)(block.x * block.y * block.z);
    if (T <= 0 || T > kMaxThreads) {
        fprintf(stderr, "bra_emu: bad block size %d\n", T);
        abort();
    }
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    int nworkers = (int)std::min<size_t>(nblocks, std::max(1u, std::thread::hardware_concurrency()));
    const char* env = getenv("BRA_EMU_THREADS");
    if (env) nworkers = std::max(1, std::min(nworkers, atoi(env)));
    std::atomic<size_t> next{0};
    auto work = [&]() {
        Worker w;
        W = &w;
        w.nthreads = T;
        w.body = &body;
        w.fibers.resize(T);
        for (int i = 0; i < T; ++i) w.fibers[i].stack = stack_get();
        w.xbuf = (uint32_t*)malloc(sizeof(uint32_t) * 2 * kMaxWaves * 64 * kXWords);
        w.smem_cap = dyn_smem_bytes + 64;
        w.smem = (char*)aligned_alloc(64, (w.smem_cap + 63) / 64 * 64);
        t_blockDim = {block.x, block.y, block.z};
        t_gridDim = {grid.x, grid.y, grid.z};
        for (;;) {
            size_t b = next.fetch_add(1);
            if (b >= nblocks) break;
            t_blockIdx.x = (unsigned)(b % grid.x);
            t_blockIdx.y = (unsigned)((b / grid.x) % grid.y);
            t_blockIdx.z = (unsigned)(b / ((size_t)grid.x * grid.y));
            run_block(&w);
        }
        for (int i = 0; i < T; ++i) stack_put(w.fibers[i].stack);
        free(w.xbuf);
        free(w.smem);
        W = nullptr;
    };
    if (nworkers == 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < nworkers; ++i) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
}

}  // namespace bra_emu
