// Deterministic SPMD emulator for ONE workgroup (TEST INFRASTRUCTURE ONLY -- never linked into the product).
//
// NT fibers (ucontext) run the same function; the only yield points are the workgroup barrier and the
// wave-level shuffle.  The scheduler visits fibers in a configurable order (forward / reverse / strided) so a
// missing barrier shows up as an order-dependent result.
#pragma once
#include <ucontext.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace emu {

struct Runtime {
    int nt = 0;
    int cur = -1;
    int order = 0;  // 0 forward, 1 reverse, 2 stride-7
    std::vector<ucontext_t> ctx;
    std::vector<char*> stacks;
    std::vector<char> done;
    ucontext_t main_ctx;
    // workgroup barrier
    int bar_count = 0;
    long bar_gen = 0;
    // per-wave rendezvous for shuffles
    std::vector<int> wave_count;
    std::vector<long> wave_gen;
    std::vector<double> exch;
    std::vector<double> exch2;
    std::function<void()> body;
    long n_barriers = 0;
    long progress = 0;  // bumped whenever a barrier releases or a fiber finishes
};

inline Runtime*& rt() {
    static Runtime* r = nullptr;
    return r;
}

inline int tid() { return rt()->cur; }

inline void yield() {
    Runtime* r = rt();
    swapcontext(&r->ctx[r->cur], &r->main_ctx);
}

inline void block_barrier() {
    Runtime* r = rt();
    long g = r->bar_gen;
    if (++r->bar_count == r->nt) {
        r->bar_count = 0;
        r->bar_gen++;
        r->n_barriers++;
        r->progress++;
    } else {
        while (r->bar_gen == g) yield();
    }
}

inline void wave_barrier() {
    Runtime* r = rt();
    int w = r->cur / 64;
    int lanes = (r->nt - w * 64) < 64 ? (r->nt - w * 64) : 64;
    long g = r->wave_gen[w];
    if (++r->wave_count[w] == lanes) {
        r->wave_count[w] = 0;
        r->wave_gen[w]++;
        r->progress++;
    } else {
        while (r->wave_gen[w] == g) yield();
    }
}

inline double shfl_xor(double v, int mask) {
    Runtime* r = rt();
    int me = r->cur;
    r->exch[me] = v;
    wave_barrier();
    int src = (me & ~63) | ((me & 63) ^ mask);
    double out = (src < r->nt) ? r->exch[src] : v;
    wave_barrier();
    return out;
}

// value of lane `src` of the calling thread's wavefront, in every lane
inline double wave_bcast(double v, int src) {
    Runtime* r = rt();
    int me = r->cur;
    r->exch[me] = v;
    wave_barrier();
    double out = r->exch[(me & ~63) | (src & 63)];
    wave_barrier();
    return out;
}

// D = A(16x4) * B(4x16) + C with the gfx950 v_mfma_f64_16x16x4_f64 lane layout:
// a: A[i = l&15][k = l>>4], b: B[k = l>>4][j = l&15], c[reg]: C[row = (l>>4) + 4*reg][col = l&15]
inline void mfma_f64_16x16x4(double a, double b, double* c) {
    Runtime* r = rt();
    int me = r->cur, base = me & ~63, l = me & 63;
    r->exch[me] = a;
    r->exch2[me] = b;
    wave_barrier();
    for (int reg = 0; reg < 4; ++reg) {
        int row = (l >> 4) + 4 * reg, col = l & 15;
        double acc = c[reg];
        for (int k = 0; k < 4; ++k) acc += r->exch[base + 16 * k + row] * r->exch2[base + 16 * k + col];
        c[reg] = acc;
    }
    wave_barrier();
}

inline void trampoline() {
    Runtime* r = rt();
    r->body();
    r->done[r->cur] = 1;
    swapcontext(&r->ctx[r->cur], &r->main_ctx);
}

// Run `body` as a workgroup of nt threads.  `order` selects the fiber visiting order.
inline long run_block(int nt, int order, std::function<void()> body) {
    Runtime R;
    Runtime* saved = rt();
    rt() = &R;
    R.nt = nt;
    R.order = order;
    R.body = body;
    R.ctx.resize(nt);
    R.done.assign(nt, 0);
    R.stacks.resize(nt);
    R.wave_count.assign((nt + 63) / 64, 0);
    R.wave_gen.assign((nt + 63) / 64, 0);
    R.exch.assign(nt, 0.0);
    R.exch2.assign(nt, 0.0);
    const size_t STK = 256 * 1024;
    for (int i = 0; i < nt; ++i) {
        R.stacks[i] = (char*)malloc(STK);
        getcontext(&R.ctx[i]);
        R.ctx[i].uc_stack.ss_sp = R.stacks[i];
        R.ctx[i].uc_stack.ss_size = STK;
        R.ctx[i].uc_link = &R.main_ctx;
        makecontext(&R.ctx[i], (void (*)())trampoline, 0);
    }
    int ndone = 0;
    while (ndone < nt) {
        ndone = 0;
        const long progress_before = R.progress;
        for (int k = 0; k < nt; ++k) {
            int i = k;
            if (order == 1) i = nt - 1 - k;
            else if (order == 2) i = (int)(((long)k * 7 + 3) % nt);  // nt is a multiple of 64: 7 is coprime
            if (R.done[i]) { ndone++; continue; }
            R.cur = i;
            swapcontext(&R.main_ctx, &R.ctx[i]);
            if (R.done[i]) { ndone++; R.progress++; }
        }
        if (ndone < nt && R.progress == progress_before) {
            fprintf(stderr, "emu: DEADLOCK -- a full scheduler pass made no progress (non-uniform barrier?)\n");
            fprintf(stderr, "emu: %d of %d threads at the workgroup barrier; wavefront rendezvous counts:", R.bar_count, nt);
            for (size_t w = 0; w < R.wave_count.size(); ++w) fprintf(stderr, " %d", R.wave_count[w]);
            fprintf(stderr, "\n");
            abort();
        }
    }
    for (int i = 0; i < nt; ++i) free(R.stacks[i]);
    long nb = R.n_barriers;
    rt() = saved;
    return nb;
}

}  // namespace emu
