// hipemu runtime: fiber scheduler for the host-side HIP stand-in (tests only).
#include <hip/hip_runtime.h>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <mutex>

namespace hipemu {

thread_local Block* g_blk = nullptr;
thread_local hipemu_uint3 g_tid, g_bid;
thread_local dim3 g_bdim, g_gdim;

static void fiber_entry() {
    Block* b = g_blk;
    (*b->body)();
    b->fibers[b->cur].state = DONE;
    swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}

static void set_tid(Block* b, int t) {
    g_tid.x = t % b->bdim.x;
    g_tid.y = (t / b->bdim.x) % b->bdim.y;
    g_tid.z = t / (b->bdim.x * b->bdim.y);
}

static void run_block(Block* b) {
    g_blk = b;
    g_bdim = b->bdim;
    g_gdim = b->gdim;
    g_bid.x = b->bidx.x; g_bid.y = b->bidx.y; g_bid.z = b->bidx.z;
    int n = b->nthreads;
    int nw = (n + WAVE - 1) / WAVE;
    for (int w = 0; w < nw; ++w) {
        memset(b->waves[w].valid, 0, sizeof(b->waves[w].valid));
        memset(b->waves[w].pub, 0, sizeof(b->waves[w].pub));
        memset(b->waves[w].gen, 0, sizeof(b->waves[w].gen));
    }
    for (int t = 0; t < n; ++t) {
        Fiber& f = b->fibers[t];
        f.state = RUN;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK_BYTES;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    int live = n;
    while (live > 0) {
        bool progress = false;
        for (int t = 0; t < n; ++t) {
            Fiber& f = b->fibers[t];
            if (f.state != RUN) continue;
            b->cur = t;
            set_tid(b, t);
            swapcontext(&b->sched, &f.ctx);
            progress = true;
            if (f.state == DONE) --live;     // its last published slot stays readable (see WaveView::peer_valid)
        }
        // block barrier: every live fiber must be AT_BARRIER
        {
            int at = 0;
            for (int t = 0; t < n; ++t) at += (b->fibers[t].state == AT_BARRIER);
            if (at > 0 && at == live) {
                for (int t = 0; t < n; ++t)
                    if (b->fibers[t].state == AT_BARRIER) b->fibers[t].state = RUN;
                progress = true;
            }
        }
        // wave rendezvous
        for (int w = 0; w < nw; ++w) {
            int lo = w * WAVE, hi = lo + WAVE < n ? lo + WAVE : n;
            int lw = 0, at = 0;
            unsigned gen = 0; bool gen_ok = true, first = true;
            for (int t = lo; t < hi; ++t) {
                if (b->fibers[t].state == DONE) continue;
                ++lw;
                if (b->fibers[t].state == AT_WAVE) {
                    ++at;
                    unsigned g = b->waves[w].gen[t - lo];
                    if (first) { gen = g; first = false; } else if (g != gen) gen_ok = false;
                }
            }
            if (at > 0 && at == lw) {
                if (!gen_ok) {
                    fprintf(stderr, "hipemu: wave %d lanes disagree on collective count (divergent collective)\n", w);
                    abort();
                }
                for (int t = lo; t < hi; ++t)
                    if (b->fibers[t].state == AT_WAVE) b->fibers[t].state = RUN;
                progress = true;
            }
        }
        if (!progress) {
            int nb = 0, nwv = 0;
            for (int t = 0; t < n; ++t) { nb += b->fibers[t].state == AT_BARRIER; nwv += b->fibers[t].state == AT_WAVE; }
            fprintf(stderr,
                    "hipemu: DEADLOCK in block (%u,%u,%u): live=%d at_barrier=%d at_wave_collective=%d "
                    "(collective or barrier reached by only part of the threads)\n",
                    b->bidx.x, b->bidx.y, b->bidx.z, live, nb, nwv);
            abort();
        }
    }
    g_blk = nullptr;
}

static void segv_handler(int sig) {
    void* bt[48];
    int n = backtrace(bt, 48);
    const char msg[] = "hipemu: fatal signal inside an emulated kernel; backtrace:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(bt, n, 2);
    if (g_blk) fprintf(stderr, "hipemu: block (%u,%u,%u) thread %d\n", g_blk->bidx.x, g_blk->bidx.y, g_blk->bidx.z, g_blk->cur);
    signal(sig, SIG_DFL);
    raise(sig);
}

void trace(const char* name) {
    static bool installed = false;
    if (!installed && getenv("HIPEMU_TRACE")) {
        installed = true;
        static char altstack[1 << 16];
        stack_t ss; ss.ss_sp = altstack; ss.ss_size = sizeof(altstack); ss.ss_flags = 0;
        sigaltstack(&ss, nullptr);
        struct sigaction sa; memset(&sa, 0, sizeof(sa));
        sa.sa_handler = segv_handler; sa.sa_flags = SA_ONSTACK;
        sigaction(SIGSEGV, &sa, nullptr);
    }
    static const bool on = getenv("HIPEMU_TRACE") != nullptr;
    if (on) { fprintf(stderr, "hipemu: launch %s\n", name); fflush(stderr); }
}

static int n_workers() {
    const char* e = getenv("HIPEMU_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : n;
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    int nthreads = (int)(block.x * block.y * block.z);
    if (nblocks == 0 || nthreads == 0) return;
    if (nthreads > 1024) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
    int nwk = n_workers();
    if ((size_t)nwk > nblocks) nwk = (int)nblocks;
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        Block b;
        b.nthreads = nthreads;
        b.bdim = block;
        b.gdim = grid;
        b.body = &body;
        b.fibers.resize(nthreads);
        b.waves.resize((nthreads + WAVE - 1) / WAVE);
        b.dyn.resize(shmem + 16);
        for (auto& f : b.fibers) f.stack = (char*)malloc(STACK_BYTES);
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nblocks) break;
            b.bidx.x = (unsigned)(i % grid.x);
            b.bidx.y = (unsigned)((i / grid.x) % grid.y);
            b.bidx.z = (unsigned)(i / ((size_t)grid.x * grid.y));
            run_block(&b);
        }
        for (auto& f : b.fibers) free(f.stack);
    };
    if (nwk == 1) { worker(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < nwk; ++i) th.emplace_back(worker);
    for (auto& t : th) t.join();
}

}  // namespace hipemu
