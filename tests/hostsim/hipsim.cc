// tests/hostsim/hipsim.cc -- TEST INFRASTRUCTURE ONLY: the fiber scheduler behind hip/hip_runtime.h
// of this directory, plus the three symbols the kernel sources expect from psgpu_core.hip.
//
// A launch runs its workgroups one after the other.  Inside a workgroup every work-item is a fiber
// (own stack, hand-written switch); a fiber runs until it waits -- at __syncthreads() or inside a cross-lane operation --
// and then hands over to the next fiber of the workgroup in the chosen order.  Waiting is a
// generation counter per barrier object (one for the workgroup, one per wavefront), so a fiber that
// is resumed early simply hands over again; if a whole round goes by without progress the kernel
// has divergent barriers and the run aborts with a message.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <vector>

#include "psgpu.h"

// The fiber switch: callee-saved registers and the stack pointer (System V x86-64).  ucontext's swapcontext
// would do, but it makes two signal-mask system calls per switch and a search of one utterance is ~10^7 switches.
#if !defined(__x86_64__)
#error "tests/hostsim needs x86-64 (the fiber switch below)"
#endif
extern "C" void hipsim_switch(void **save_sp, void *load_sp);
asm(R"(
    .text
    .globl hipsim_switch
    .type hipsim_switch,@function
hipsim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipsim_switch, .-hipsim_switch
)");

namespace hipsim {

struct Barrier { int count = 0, size = 0; unsigned gen = 0; };

struct Fiber {
    void *sp = nullptr;
    char *stack = nullptr;
    int tid = 0, done = 0;
};

struct Wave {
    Barrier bar;
    uint32_t slot[64];
    uint64_t ballot = 0;
};

struct Block {
    std::vector<Fiber> fib;
    std::vector<Wave> waves;
    std::vector<int> order;          // scheduling order: position -> tid
    std::vector<int> where;          // tid -> position
    Barrier bar;
    int n = 0, n_done = 0, idle = 0;
    void *main_sp = nullptr;
    const std::function<void()> *body = nullptr;
    dim3 bdim;
};

thread_local Fiber *cur = nullptr;
thread_local dim3 v_threadIdx, v_blockIdx, v_blockDim, v_gridDim;
static thread_local Block *g_blk = nullptr;
static const size_t kStack = 96 * 1024;

static void set_ids(Block *b, int tid)
{
    v_threadIdx = dim3(tid % b->bdim.x, (tid / b->bdim.x) % b->bdim.y, tid / (b->bdim.x * b->bdim.y));
}

static void switch_to_next()
{
    Block *b = g_blk;
    Fiber *me = cur;
    int pos = b->where[me->tid];
    for (int k = 1; k <= b->n; ++k) {
        Fiber *f = &b->fib[b->order[(pos + k) % b->n]];
        if (f->done) continue;
        if (f == me) return;
        cur = f;
        set_ids(b, f->tid);
        hipsim_switch(&me->sp, f->sp);
        return;
    }
    // every other fiber has finished
    if (me->done) { cur = nullptr; hipsim_switch(&me->sp, b->main_sp); }
}

static void wait_on(Barrier &bar)
{
    Block *b = g_blk;
    const unsigned my = bar.gen;
    if (++bar.count == bar.size) { bar.count = 0; ++bar.gen; b->idle = 0; return; }
    while (bar.gen == my) {
        if (++b->idle > 4 * b->n + 8) {
            fprintf(stderr, "hipsim: no work-item can make progress (divergent barrier or a work-item left the kernel "
                            "while others wait); block (%u), work-item %d\n", v_blockIdx.x, cur->tid);
            abort();
        }
        switch_to_next();
    }
}

void sync_block() { wait_on(g_blk->bar); }

uint32_t xchg_wave(uint32_t v, int f(int, int), int arg)
{
    Wave &w = g_blk->waves[cur->tid >> 6];
    const int lane = cur->tid & 63;
    w.slot[lane] = v;
    wait_on(w.bar);
    const int src = f(lane, arg);
    const uint32_t r = (src >= 0 && src < w.bar.size) ? w.slot[src] : v;
    wait_on(w.bar);
    return r;
}

uint64_t ballot_wave(bool pred)
{
    Wave &w = g_blk->waves[cur->tid >> 6];
    const int lane = cur->tid & 63;
    if (lane == 0) w.ballot = 0;         // lanes arrive in any order: clear before the first barrier of the op
    wait_on(w.bar);
    if (pred) w.ballot |= 1ull << lane;
    wait_on(w.bar);
    const uint64_t r = w.ballot;
    wait_on(w.bar);
    return r;
}

static void fiber_main()
{
    Block *b = g_blk;
    Fiber *me = cur;
    (*b->body)();
    me->done = 1;
    ++b->n_done;
    b->idle = 0;
    // barriers count the work-items still in the kernel (as the hardware barrier does)
    b->bar.size = b->n - b->n_done;
    if (b->bar.size && b->bar.count == b->bar.size) { b->bar.count = 0; ++b->bar.gen; }
    Wave &w = b->waves[me->tid >> 6];
    --w.bar.size;
    if (w.bar.size && w.bar.count == w.bar.size) { w.bar.count = 0; ++w.bar.gen; }
    for (;;) switch_to_next();           // never returns into a finished fiber
}

static uint32_t g_seed = 1;
static uint32_t rnd() { g_seed = g_seed * 1664525u + 1013904223u; return g_seed >> 8; }

void launch(dim3 grid, dim3 block, const std::function<void()> &body)
{
    const int n = (int)(block.x * block.y * block.z);
    const char *ord = getenv("PSGPU_SIM_ORDER");      // "fwd" (default) | "rev" | "shuffle[:seed]"
    Block b;
    b.n = n; b.bdim = block; b.body = &body;
    b.fib.resize(n);
    b.waves.resize((n + 63) / 64);
    b.order.resize(n); b.where.resize(n);
    for (int i = 0; i < n; ++i) b.fib[i].stack = (char *)malloc(kStack);
    if (ord && !strncmp(ord, "shuffle", 7)) g_seed = ord[7] == ':' ? (uint32_t)atoi(ord + 8) : 12345u;
    v_blockDim = block; v_gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        v_blockIdx = dim3(bx, by, bz);
        b.n_done = 0; b.idle = 0;
        b.bar = Barrier(); b.bar.size = n;
        for (size_t w = 0; w < b.waves.size(); ++w) { b.waves[w].bar = Barrier(); b.waves[w].bar.size = std::min(64, n - (int)w * 64); }
        for (int i = 0; i < n; ++i) b.order[i] = i;
        if (ord && !strcmp(ord, "rev")) std::reverse(b.order.begin(), b.order.end());
        else if (ord && !strncmp(ord, "shuffle", 7)) for (int i = n - 1; i > 0; --i) std::swap(b.order[i], b.order[rnd() % (i + 1)]);
        for (int i = 0; i < n; ++i) b.where[b.order[i]] = i;
        for (int i = 0; i < n; ++i) {
            Fiber &f = b.fib[i];
            f.tid = i; f.done = 0;
            // first switch "returns" into fiber_main with the stack aligned as after a call
            uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
            void **q = (void **)(top - 64);
            for (int k = 0; k < 6; ++k) q[k] = nullptr;
            q[6] = (void *)fiber_main;
            f.sp = q;
        }
        g_blk = &b;
        cur = &b.fib[b.order[0]];
        set_ids(&b, cur->tid);
        hipsim_switch(&b.main_sp, cur->sp);
        g_blk = nullptr;
    }
    for (int i = 0; i < n; ++i) free(b.fib[i].stack);
}

}  // namespace hipsim

// ---- what the kernel sources take from psgpu_core.hip -----------------------------------------------
static thread_local char g_err[512];
void psgpu_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
int psgpu_check_device() { return PSGPU_OK; }
extern "C" const char *psgpu_last_error(void) { return g_err; }
