/* tests/emu/hip_emu.cpp -- TEST INFRASTRUCTURE ONLY: fiber scheduler for tests/emu/hip/hip_runtime.h */
#include "hip/hip_runtime.h"

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace hipemu {

struct Fiber {
    ucontext_t ctx;
    char *stack;
    int state;          /* 0 runnable, 1 waiting wave, 2 waiting block, 3 done */
    unsigned gen_wait;  /* generation waited on */
};

static std::vector<Fiber> fibers;
static ucontext_t sched_ctx;
static unsigned cur;                 /* current fiber index */
static unsigned nthreads;
static std::vector<unsigned> wave_arrived, wave_gen;
static unsigned block_arrived, block_gen;
static std::vector<uint64_t> slots;  /* 64 per wave */
static void (*g_tramp)(void *);
static void *g_args;
static const size_t STACK = 256 * 1024;

unsigned lane_id() { return cur & 63u; }
uint64_t *wave_slots() { return &slots[(cur >> 6) * 64]; }

static unsigned wave_size(unsigned w) { return std::min<unsigned>(64u, nthreads - w * 64); }
static unsigned live_in_wave(unsigned w)
{
    unsigned c = 0;
    for (unsigned i = w * 64; i < std::min<unsigned>(nthreads, w * 64 + 64); i++) c += fibers[i].state != 3;
    return c;
}

static void yield_to_sched() { swapcontext(&fibers[cur].ctx, &sched_ctx); }

void wave_sync()
{
    unsigned w = cur >> 6;
    /* all lanes of the wave (that have not exited) must arrive; exited lanes never arrive:
       calling a wave collective after some lanes returned is a test failure */
    if (live_in_wave(w) != wave_size(w)) { fprintf(stderr, "hipemu: wave collective after partial wave exit\n"); abort(); }
    wave_arrived[w]++;
    if (wave_arrived[w] == wave_size(w)) {
        wave_arrived[w] = 0;
        wave_gen[w]++;
        return;
    }
    fibers[cur].state = 1;
    fibers[cur].gen_wait = wave_gen[w];
    yield_to_sched();
}

void block_sync()
{
    block_arrived++;
    unsigned live = 0;
    for (unsigned i = 0; i < nthreads; i++) live += fibers[i].state != 3;
    if (live != nthreads) { fprintf(stderr, "hipemu: __syncthreads after partial block exit\n"); abort(); }
    if (block_arrived == nthreads) {
        block_arrived = 0;
        block_gen++;
        return;
    }
    fibers[cur].state = 2;
    fibers[cur].gen_wait = block_gen;
    yield_to_sched();
}

static void fiber_main()
{
    g_tramp(g_args);
    fibers[cur].state = 3;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}

void launch(void (*tramp)(void *), void *args, dim3 grid, dim3 block)
{
    g_tramp = tramp;
    g_args = args;
    nthreads = block.x;
    blockDim = block;
    gridDim = grid;
    unsigned nw = (nthreads + 63) / 64;
    if (fibers.size() < nthreads) {
        size_t old = fibers.size();
        fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; i++) fibers[i].stack = (char *)malloc(STACK);
    }
    for (unsigned bb = 0; bb < grid.x * grid.y; bb++) {
        const unsigned b = bb % grid.x, by = bb / grid.x;      // 2-D grids: x fastest
        wave_arrived.assign(nw, 0);
        wave_gen.assign(nw, 0);
        slots.assign((size_t)nw * 64, 0);
        block_arrived = 0;
        block_gen = 0;
        for (unsigned t = 0; t < nthreads; t++) {
            Fiber &f = fibers[t];
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = STACK;
            f.ctx.uc_link = &sched_ctx;
            f.state = 0;
            makecontext(&f.ctx, fiber_main, 0);
        }
        unsigned done = 0;
        while (done < nthreads) {
            unsigned progressed = 0;
            for (unsigned t = 0; t < nthreads; t++) {
                Fiber &f = fibers[t];
                if (f.state == 3) continue;
                if (f.state == 1 && wave_gen[t >> 6] == f.gen_wait) continue;
                if (f.state == 2 && block_gen == f.gen_wait) continue;
                f.state = 0;
                cur = t;
                threadIdx.x = t; threadIdx.y = 0; threadIdx.z = 0;
                blockIdx.x = b; blockIdx.y = by; blockIdx.z = 0;
                swapcontext(&sched_ctx, &f.ctx);
                progressed++;
                if (f.state == 3) done++;
            }
            if (!progressed) { fprintf(stderr, "hipemu: deadlock (divergent barrier?) in block %u\n", b); abort(); }
        }
    }
}

}  // namespace hipemu
