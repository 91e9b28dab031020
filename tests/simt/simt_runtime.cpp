// tests/simt/simt_runtime.cpp — fiber scheduler + host-runtime stubs behind tests/simt/hip/hip_runtime.h.
// TEST INFRASTRUCTURE (see the header).  One OS thread; a workgroup = blockDim fibers; workgroups run sequentially.
#include <hip/hip_runtime.h>
#if !defined(__x86_64__)
#include <ucontext.h>
#endif
#include <chrono>
#include <random>
#include <string>
#include <vector>

// dynamic LDS windows of the kernel sources (each `extern __shared__ ... name[]` needs one definition)
thread_local __attribute__((aligned(16))) char ba_smem[160 * 1024];
thread_local __attribute__((aligned(16))) char bp_smem[160 * 1024];
thread_local __attribute__((aligned(16))) char mg_smem[160 * 1024];
thread_local __attribute__((aligned(16))) unsigned char sel_smem[160 * 1024];

// ---- fiber switch.  glibc's swapcontext saves / restores the signal mask with a system call on every switch (half of the run time
//      of the emulated tests was kernel time); a fiber here only needs its callee-saved registers and its stack pointer.
#if defined(__x86_64__)
extern "C" void simt_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl simt_switch
    .type simt_switch,@function
simt_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size simt_switch,.-simt_switch
)");
struct FiberCtx { void* sp = nullptr; };
static void ctx_make(FiberCtx& c, char* stack, size_t size, void (*entry)()) {
    // the frame simt_switch pops: [mxcsr | x87 cw][r15 r14 r13 r12 rbx rbp][return address = entry][0]; at `entry` the stack is
    // aligned as after a call (rsp = 16k + 8)
    uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
    uint64_t* a = (uint64_t*)(top - 16);
    a[1] = 0;
    a[0] = (uint64_t)(uintptr_t)entry;
    for (int k = 1; k <= 6; ++k) a[-k] = 0;
    uint32_t* fp = (uint32_t*)(a - 7);
    fp[0] = 0x1F80;                    // mxcsr: default rounding, exceptions masked
    fp[1] = 0x037F;                    // x87 control word
    c.sp = (void*)(a - 7);
}
static inline void ctx_switch(FiberCtx& from, FiberCtx& to) { simt_switch(&from.sp, to.sp); }
#else
struct FiberCtx { ucontext_t uc; };
static void ctx_make(FiberCtx& c, char* stack, size_t size, void (*entry)()) {
    getcontext(&c.uc);
    c.uc.uc_stack.ss_sp = stack; c.uc.uc_stack.ss_size = size; c.uc.uc_link = nullptr;
    makecontext(&c.uc, entry, 0);
}
static inline void ctx_switch(FiberCtx& from, FiberCtx& to) { swapcontext(&from.uc, &to.uc); }
#endif

namespace simt {
thread_local Lane* cur = nullptr;
thread_local dim3 g_block, g_grid;
thread_local const char* kernarg_end = nullptr;

namespace {
enum { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber {
    FiberCtx ctx;
    char* stack = nullptr;   // from the thread's pool (below)
    Lane lane;
    int state = RUN;
    unsigned gen = 0;        // generation of the barrier it waits on
};
struct WaveState {
    int alive = 0, arrived = 0;
    unsigned gen = 0;
    uint64_t buf[2][64];
};
struct BlockState {
    std::vector<Fiber> f;
    std::vector<WaveState> w;
    int alive = 0, arrived = 0;
    unsigned gen = 0;
    FiberCtx sched;
    int running = -1;
    const std::function<void()>* body = nullptr;
};
thread_local BlockState* B = nullptr;
const size_t kStack = 256 * 1024;
// fiber stacks are kept per host thread and re-used by every launch: allocating (and zero-filling) 256 KB per lane and launch was
// most of the run time of the emulated tests (page faults).  Uninitialised on purpose: only the pages a fiber touches get mapped.
struct StackPool {
    std::vector<char*> s;
    ~StackPool() { for (char* p : s) free(p); }
    char* get(size_t i) {
        while (s.size() <= i) { void* p = nullptr; if (posix_memalign(&p, 64, kStack)) abort(); s.push_back((char*)p); }
        return s[i];
    }
};
thread_local StackPool g_stacks;

void release_block() { B->arrived = 0; ++B->gen; }
void release_wave(WaveState& w) { w.arrived = 0; ++w.gen; }

void fiber_main() {
    Fiber& me = B->f[B->running];
    (*B->body)();
    me.state = DONE;
    --B->alive;
    WaveState& w = B->w[me.lane.wave];
    --w.alive;
    // a finished lane no longer takes part in barriers: release whoever was waiting for it
    if (B->alive > 0 && B->arrived == B->alive) release_block();
    if (w.alive > 0 && w.arrived == w.alive) release_wave(w);
    ctx_switch(me.ctx, B->sched);
}

void yield_to_scheduler() {
    Fiber& me = B->f[B->running];
    ctx_switch(me.ctx, B->sched);
}
}  // namespace

void sync_block() {
    Fiber& me = B->f[B->running];
    const unsigned g = B->gen;
    if (++B->arrived == B->alive) { release_block(); return; }
    me.state = WAIT_BLOCK; me.gen = g;
    yield_to_scheduler();
}
void sync_wave() {
    Fiber& me = B->f[B->running];
    WaveState& w = B->w[me.lane.wave];
    const unsigned g = w.gen;
    if (++w.arrived == w.alive) { release_wave(w); return; }
    me.state = WAIT_WAVE; me.gen = g;
    yield_to_scheduler();
}
uint64_t* xchg(unsigned parity) { return B->w[cur->wave].buf[parity & 1]; }
long long clock() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body) {
    if (lds_bytes > 160 * 1024) { fprintf(stderr, "simt: %zu bytes of dynamic LDS requested\n", lds_bytes); abort(); }
    const char* ord = getenv("SIMT_ORDER");
    const int order = !ord ? 0 : (!strcmp(ord, "reverse") ? 1 : (!strcmp(ord, "shuffle") ? 2 : 0));
    const int nthr = (int)(block.x * block.y * block.z);
    const int nwave = (nthr + 63) / 64;
    g_block = block; g_grid = grid;
    BlockState bs;
    bs.f.resize(nthr);
    for (int t = 0; t < nthr; ++t) bs.f[t].stack = g_stacks.get(t);
    std::vector<int> perm(nthr);
    std::mt19937 rng(12345);
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        B = &bs;
        bs.body = &body;
        bs.w.assign(nwave, WaveState());
        bs.alive = nthr; bs.arrived = 0; bs.gen = 0;
        for (int t = 0; t < nthr; ++t) {
            Fiber& f = bs.f[t];
            f.state = RUN; f.gen = 0;
            f.lane.flat = t; f.lane.lane = t & 63; f.lane.wave = t >> 6; f.lane.xcnt = 0;
            f.lane.tid.x = t % block.x; f.lane.tid.y = (t / block.x) % block.y; f.lane.tid.z = t / (block.x * block.y);
            f.lane.bid.x = bx; f.lane.bid.y = by; f.lane.bid.z = bz;
            bs.w[f.lane.wave].alive++;
            ctx_make(f.ctx, f.stack, kStack, fiber_main);
        }
        for (int t = 0; t < nthr; ++t) perm[t] = order == 1 ? nthr - 1 - t : t;
        int stalled = 0;
        while (bs.alive > 0) {
            if (order == 2) std::shuffle(perm.begin(), perm.end(), rng);
            bool progressed = false;
            for (int k = 0; k < nthr; ++k) {
                Fiber& f = bs.f[perm[k]];
                if (f.state == DONE) continue;
                if (f.state == WAIT_BLOCK) { if (bs.gen == f.gen) continue; f.state = RUN; }
                if (f.state == WAIT_WAVE) { if (bs.w[f.lane.wave].gen == f.gen) continue; f.state = RUN; }
                bs.running = perm[k];
                cur = &f.lane;
                progressed = true;
                ctx_switch(bs.sched, f.ctx);
            }
            if (!progressed && ++stalled > 2) {
                fprintf(stderr, "simt: deadlock in block (%u,%u,%u): %d lanes alive, %d at the block barrier; a barrier or a "
                        "wavefront collective sits in divergent control flow\n", bx, by, bz, bs.alive, bs.arrived);
                for (int w = 0; w < nwave; ++w)
                    fprintf(stderr, "   wave %d: alive %d, arrived at wave sync %d\n", w, bs.w[w].alive, bs.w[w].arrived);
                abort();
            }
            if (progressed) stalled = 0;
        }
    }
    B = nullptr; cur = nullptr;
}
}  // namespace simt

// ---------------------------------------------------------------------------------------------- host runtime stubs
struct simt_node { dim3 grid, block; size_t lds; std::function<void()> body; };
struct simt_graph { std::vector<simt_node> nodes; };
struct simt_stream { int id; simt_graph* cap = nullptr; };
struct simt_event { long long t; };
static thread_local hipError_t g_last = hipSuccess;
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); if (*p) memset(*p, 0xCD, n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < h; ++r) memmove((char*)d + r * dp, (const char*)s + r * sp, w);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new simt_stream{1}; return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new simt_stream{1}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new simt_event{0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new simt_event{0}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = simt::clock(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((b->t - a->t) * 1e-6); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
namespace simt {
bool capture_launch(hipStream_t s, dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body) {
    if (!s || !s->cap) return false;
    s->cap->nodes.push_back(simt_node{grid, block, lds_bytes, body});
    return true;
}
}
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
    if (!s || s->cap) return hipErrorInvalidValue;
    s->cap = new simt_graph();
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) {
    if (!s || !s->cap) return hipErrorInvalidValue;
    *g = s->cap; s->cap = nullptr;
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* x, hipGraph_t g, void*, void*, size_t) { *x = new simt_graph(*g); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t x) { delete x; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t x, hipStream_t) {
    for (const simt_node& n : x->nodes) simt::launch(n.grid, n.block, n.lds, n.body);
    return hipSuccess;
}
hipError_t hipGetLastError() { hipError_t e = g_last; g_last = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess (simt)" : "hip error (simt)"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
