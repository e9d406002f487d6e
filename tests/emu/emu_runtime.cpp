// TEST-ONLY fiber scheduler behind tests/emu/hip/hip_runtime.h (see the header).
#include <hip/hip_runtime.h>
#include <ucontext.h>

#include <cstdlib>
#include <deque>
#include <vector>

namespace hipemu {

dim3 tIdx, bIdx, bDim, gDim;

enum { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    int state = DONE;
};

static constexpr size_t kStack = 256 * 1024;
static std::vector<Fiber> fibers;
static ucontext_t sched_ctx;
static int cur = -1;
static const std::function<void()>* body = nullptr;
static unsigned xbuf[16][64][2];
static unsigned short bbuf[16][64][2][8];

// Counted asynchronous loads (tsii_common.h: async_load16 / async_load4 / async_wait<N>): issued loads land only when a wait
// retires them, oldest first, exactly as vmcnt counts them on the chip; until then the destination holds a NaN pattern, so a wait
// count that is too large (a register read before its load is guaranteed to have landed) shows up as NaNs in a parity test
// instead of passing on a host that completes every load at issue.
struct PendingLoad {
    void* dst;
    const void* src;
    int bytes;
};
static std::vector<std::deque<PendingLoad>> pending;

void async_issue(void* dst, const void* src, int bytes) {
    std::memset(dst, 0xFF, (size_t)bytes);
    pending[cur].push_back(PendingLoad{dst, src, bytes});
}
void async_retire(int keep) {
    std::deque<PendingLoad>& q = pending[cur];
    while ((int)q.size() > keep) {
        std::memcpy(q.front().dst, q.front().src, (size_t)q.front().bytes);
        q.pop_front();
    }
}

static void entry() {
    (*body)();
    pending[cur].clear();          // loads still in flight at the end of a thread were never consumed
    fibers[cur].state = DONE;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}

static void yield_to_sched(int st) {
    fibers[cur].state = st;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}

void syncthreads() { async_retire(0); yield_to_sched(WAIT_BLOCK); }     // __syncthreads() = s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier
void barrier_only() { yield_to_sched(WAIT_BLOCK); }                     // lds_barrier(): loads in flight stay in flight
static long spins = 0;
void yield() {
    if (++spins > 200000000L) { std::fprintf(stderr, "hipemu: livelock (spin-wait never satisfied)\n"); std::abort(); }
    yield_to_sched(RUN);
}
static void wavesync() { yield_to_sched(WAIT_WAVE); }

int lane_id() { return cur & 63; }

unsigned exchange32(unsigned v, int mode, int arg, int width) {
    const int lane = cur & 63, wave = cur >> 6;
    xbuf[wave][lane][0] = v;
    wavesync();
    int src = lane;
    const int base = lane - (lane % width);
    const int rel = lane % width;
    if (mode == 0) { if (rel + arg < width) src = lane + arg; }
    else if (mode == 1) { if (rel - arg >= 0) src = lane - arg; }
    else if (mode == 2) { int r2 = rel ^ arg; if (r2 < width) src = base + r2; }
    else { src = base + (((arg % width) + width) % width); }
    unsigned r = xbuf[wave][src][0];
    wavesync();
    return r;
}

f32x16_emu mfma_32x32x2(float a, float b, f32x16_emu c) {
    const int lane = cur & 63, wave = cur >> 6;
    std::memcpy(&xbuf[wave][lane][0], &a, 4);
    std::memcpy(&xbuf[wave][lane][1], &b, 4);
    wavesync();
    const int hi = lane >> 5, col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            std::memcpy(&av, &xbuf[wave][row + 32 * k][0], 4);
            std::memcpy(&bv, &xbuf[wave][col + 32 * k][1], 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    wavesync();
    return c;
}

f32x4_emu mfma_16x16x4(float a, float b, f32x4_emu c) {
    const int lane = cur & 63, wave = cur >> 6;
    std::memcpy(&xbuf[wave][lane][0], &a, 4);
    std::memcpy(&xbuf[wave][lane][1], &b, 4);
    wavesync();
    const int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            std::memcpy(&av, &xbuf[wave][row + 16 * k][0], 4);
            std::memcpy(&bv, &xbuf[wave][col + 16 * k][1], 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    wavesync();
    return c;
}

f32x16_emu mfma_32x32x16_bf16(bf16x8_emu a, bf16x8_emu b, f32x16_emu c) {
    const int lane = cur & 63, wave = cur >> 6;
    std::memcpy(&bbuf[wave][lane][0][0], &a, 16);
    std::memcpy(&bbuf[wave][lane][1][0], &b, 16);
    wavesync();
    const int hi = lane >> 5, col = lane & 31;
    auto f = [](unsigned short h) { unsigned u = (unsigned)h << 16; float v; std::memcpy(&v, &u, 4); return v; };
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        // the 16 exact bf16 products are summed wide and rounded into the fp32 accumulator once (the hardware's internal
        // order / width is not documented; GPU-side accuracy is pinned by tests/test_parity_r2.py on the real chip)
        double acc = 0.0;
        for (int k = 0; k < 16; ++k)
            acc += (double)f(bbuf[wave][row + 32 * (k >> 3)][0][k & 7]) * (double)f(bbuf[wave][col + 32 * (k >> 3)][1][k & 7]);
        c[r] = (float)((double)c[r] + acc);
    }
    wavesync();
    return c;
}

static void run_block(int nthreads) {
    for (int t = 0; t < nthreads; ++t) {
        Fiber& f = fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())entry, 0);
        f.state = RUN;
    }
    int live = nthreads;
    const int nwaves = (nthreads + 63) / 64;
    while (live > 0) {
        bool progressed = false;
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = fibers[t];
            if (f.state != RUN) continue;
            cur = t;
            tIdx.x = t % bDim.x;
            tIdx.y = (t / bDim.x) % bDim.y;
            tIdx.z = t / (bDim.x * bDim.y);
            swapcontext(&sched_ctx, &f.ctx);
            progressed = true;
            if (f.state == DONE) { --live; spins = 0; }
        }
        bool released = false;
        for (int w = 0; w < nwaves; ++w) {  // wave collectives
            int waiting = 0, alive = 0;
            for (int t = w * 64; t < std::min(nthreads, w * 64 + 64); ++t) {
                if (fibers[t].state != DONE) ++alive;
                if (fibers[t].state == WAIT_WAVE) ++waiting;
            }
            if (alive > 0 && waiting == alive) {
                for (int t = w * 64; t < std::min(nthreads, w * 64 + 64); ++t)
                    if (fibers[t].state == WAIT_WAVE) fibers[t].state = RUN;
                released = true;
            }
        }
        {
            int waiting = 0;
            for (int t = 0; t < nthreads; ++t) if (fibers[t].state == WAIT_BLOCK) ++waiting;
            if (live > 0 && waiting == live) {
                for (int t = 0; t < nthreads; ++t) if (fibers[t].state == WAIT_BLOCK) fibers[t].state = RUN;
                released = true;
                spins = 0;
            }
        }
        if (!progressed && !released && live > 0) {
            std::fprintf(stderr, "hipemu: deadlock (divergent barrier / collective) in block (%u,%u,%u)\n",
                         bIdx.x, bIdx.y, bIdx.z);
            std::abort();
        }
    }
}

static long launches_by_threads[1025];

void launch(dim3 grid, dim3 block, const std::function<void()>& fn) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 1024) ++launches_by_threads[nthreads];
    if (nthreads > 1024) { std::fprintf(stderr, "hipemu: block too large\n"); std::abort(); }
    if ((int)fibers.size() < nthreads) {
        size_t old = fibers.size();
        fibers.resize(nthreads);
        for (size_t i = old; i < fibers.size(); ++i) fibers[i].stack = (char*)std::malloc(kStack);
        pending.resize(fibers.size());
    }
    body = &fn;
    bDim = block;
    gDim = grid;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                bIdx = dim3(x, y, z);
                run_block(nthreads);
            }
    body = nullptr;
}

}  // namespace hipemu

// test introspection: kernel launches so far with the given block size (e.g. 768 = the producer / consumer GEMM)
extern "C" long hipemu_launches(int block_threads) {
    return (block_threads >= 0 && block_threads <= 1024) ? hipemu::launches_by_threads[block_threads] : -1;
}
