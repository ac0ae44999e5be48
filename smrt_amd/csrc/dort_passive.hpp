// The per-pair driver (passive mode) in its fused / prep / finish shapes, with the pieces the active driver shares
// (pair_setup, pruning, per-layer failure records).
// Part of the DORT device code (see dort_device.hpp for the overview and the reference map).
#pragma once
#include "spmd.hpp"
#include <math.h>
#include <string.h>
#include "dort_gauss_jordan.hpp"
#include "dort_interface_dense.hpp"

namespace smrt {

// ------------------------------------------------------------------------------------------------------------
// DORT option prune_deep_snowpack (smrt/rtsolver/dort.py:443-452): the reference stops assembling its boundary system
// after the layer in which the running optical depth sum_l min|beta_l| thickness_l passes the threshold and cuts the
// rows / unknowns of everything below.  Here: the number of layers the bottom-up recursion starts from.  The
// eigenvalues (singular values) of all the layers of this pair (and azimuth mode) lie in the staging area of the
// pipeline; tau is an LDS scratch of Lmax doubles.  Workgroup-uniform result.
// ------------------------------------------------------------------------------------------------------------
template <int NT>
SMRT_DEV int pruned_layer_count(const DevStage& stg, long long item0, int L, const double* thick, double* tau,
                                double limit) {
    const int t = tid();
    const int lane = t % SMRT_LANES, wave = t / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    for (int l = wave; l < L; l += NW) {
        const long long item = item0 + l;
        const int N = stage_rows(stg.n[item]);   // < 0: the diagonalisation of this layer failed (see first_failed_layer)
        double m = 1e300;
        for (int r = lane; r < N; r += SMRT_LANES) {
            const double sg = stg.sigma[item * stg.vec_stride + r];
            m = sg < m ? sg : m;
        }
        for (int k = 1; k < SMRT_LANES; k <<= 1) { const double o = shfl_xor(m, k); m = o < m ? o : m; }
        if (lane == 0) tau[l] = (N > 0) ? m * thick[l] : -1.0;
    }
    block_sync();
    double acc = 0.0;
    int keep = L;
    for (int l = 0; l < L; ++l) {
        if (tau[l] < 0.0) break;  // a failed layer above the cut: the reference reaches it too (keep everything,
                                  // the failure is reported by the caller)
        acc += tau[l];
        if (acc > limit) { keep = l + 1; break; }
    }
    block_sync();
    return keep;
}

// After a round of prep + Jacobi over the layers [0, layer_hi) under prune_deep_snowpack (one wavefront per pair):
// done[p] = 1 when the running optical depth sum_l min|beta_l| thickness_l of EVERY azimuth mode of pair p has passed the
// threshold within the layers processed so far (or a layer above failed: the finish kernel reports that, the layers
// below do not matter), so that the next rounds leave the pair alone (dort.py:443-452: the reference stops assembling
// at the cut).  The decision uses the very singular values the finish kernel's pruned_layer_count will use.
SMRT_DEV void prune_mark_pair(const DevBatch& b, const DevStage& stg, long long p, int* done) {
    const int lane = tid() % SMRT_LANES;
    const long long gp = global_pair(b, p);
    const int si = (int)(gp % b.S);
    const int L = b.n_layers[si];
    const int nmodes = (b.mode == 1) ? b.m_max + 1 : 1;
    const double* thick = b.thickness + (long long)si * b.Lmax;
    const int upto = b.layer_hi < L ? b.layer_hi : L;
    bool all_cut = true;
    for (int m = 0; m < nmodes; ++m) {
        double acc = 0.0;
        bool cut = false;
        for (int l = 0; l < upto && !cut; ++l) {   // uniform over the wavefront
            const long long item = (p * nmodes + m) * b.Lmax + l;
            const int N = stage_rows(stg.n[item]);
            if (N < 0) { cut = true; break; }   // failed layer: nothing below it is needed
            if (N == 0) break;                  // not processed (cannot happen above the running round)
            double mn = 1e300;
            for (int r = lane; r < N; r += SMRT_LANES) { const double sg = stg.sigma[item * stg.vec_stride + r]; mn = sg < mn ? sg : mn; }
            for (int k = 1; k < SMRT_LANES; k <<= 1) { const double o = shfl_xor(mn, k); mn = o < mn ? o : mn; }
            acc += mn * thick[l];
            if (acc > b.prune_tau) cut = true;
        }
        all_cut = all_cut && cut;
    }
    if (lane == 0) done[p] = all_cut ? 1 : 0;
}

// The prep and Jacobi kernels of the pipelines record a failed layer (renormalisation beyond 30 %, albedo >= 1, no
// convergence) as n[item] = -status instead of failing the pair: the reference diagonalises its layers from the top
// inside the loop that assembles the boundary system (dort.py:312-336) and never reaches the layers that
// prune_deep_snowpack cuts away, so only a failure among the kept layers counts -- the first one from the top.
SMRT_DEV int first_failed_layer(const DevStage& stg, long long item0, int n_kept) {
    for (int l = 0; l < n_kept; ++l) {
        const int n = stg.n[item0 + l];
        if (n < 0) return -n;
    }
    return ST_OK;
}

// ------------------------------------------------------------------------------------------------------------
// the per-pair solve (passive mode, azimuth mode 0, 2 polarisations)
// ------------------------------------------------------------------------------------------------------------
template <int NT>
SMRT_DEV void fail_pair(const DevBatch& b, long long p, int code, int out_stride) {
    const int t = tid();
    for (int i = t; i < out_stride; i += NT) b.out[p * out_stride + i] = NAN;
    if (t == 0) b.status[p] = code;
}


// ---- stages 0 and 1 of a pair, shared by the passive and the active drivers ---------------------------------------
// Layer scalars (one thread per layer), Gauss-Legendre sines, number of streams per layer and the air streams
// (streams.py:136-223).  s.ints[0..7] must be zero on entry.  Returns the status, uniform over the workgroup; on
// ST_OK s.ints[4] = most refringent layer, s.ints[5] = n_air.
template <int NT>
SMRT_DEV int pair_setup(const DevBatch& b, const Lds& s, double frequency, int L /* layers of the snowpack; s.ints[6] on return */, const double* thickness,
                        const double* fracvol, const double* temperature, const double* mp1, const double* mp2,
                        const int* kinds = nullptr /* this snowpack's row of b.layer_kind, or null */,
                        long long gp = 0 /* global pair f * S + s: row of the host-evaluated emmodel arrays */) {
    const int t = tid();
    const int nmax = b.n_max_stream;
    for (int l = t; l < L; l += NT) {
        cplx ee; double ks, ka, pa, pb, krho = 0.0; int bad = 0;
        const int kind = kinds ? kinds[l] : b.emmodel + 16 * b.micro;   // emmodel + 16 * microstructure of this layer
        if (__builtin_expect((kind & 15) == EM_HOST || (kind & 15) == EM_IBA_HOST || (kind & 15) == EM_RAYLEIGH_HOST, 0)) {   // scalars from the caller (smrt_batch.host_layer)
            if (b.host_layer) {
                const double* h = b.host_layer + (gp * b.Lmax + l) * 4;   // (l: still the index in the input arrays here)
                ks = h[0]; ka = h[1]; ee = cmk(h[2], h[3]);
                if (!(ka >= 0.0) || !(ee.re > 0.0)) bad = 1;
            } else { ks = ka = 0.0; ee = cmk(1.0, 0.0); bad = 1; }
            pa = pb = 0.0;
            if ((kind & 15) == EM_RAYLEIGH_HOST) pa = 1.5 * ks;   // the Rayleigh phase matrix of the dmrt kinds (rayleigh.py:127)
            if ((kind & 15) == EM_IBA_HOST) {
                // IBA's phase function with the caller's coefficient: pa / pb as layer_em leaves them for IBA (iba.py:228-244)
                const double coeff = b.host_coeff ? b.host_coeff[gp * b.Lmax + l] : -1.0;
                if (!(coeff >= 0.0)) bad = 1;
                const double kfac = 2.0 * (2.0 * kPi * frequency / kCSpeed) * csqrt_(ee).re;
                const double p1 = mp1[l], fv = fracvol[l];
                if ((kind >> 4) == MS_EXP) { pa = coeff * fv * (1.0 - fv) * 8.0 * kPi * p1 * p1 * p1; pb = 0.5 * kfac * kfac * p1 * p1; }
                else { pa = coeff; pb = 0.5 * kfac * kfac; }
                if ((kind >> 4) >= MS_EXPC) {   // complex wavenumber: k^2 = 4 k0^2 eps_eff sin^2(Theta / 2); Im / Re travels in pc
                    const double k0 = 2.0 * kPi * frequency / kCSpeed;
                    pb = 2.0 * k0 * k0 * ee.re;
                    krho = ee.im / ee.re;
                    if (!(krho >= 0.0 && krho < 0.5)) bad = 1;
                }
            }
        } else
        layer_em(kind & 15, kind >> 4, frequency, fracvol[l], temperature[l], mp1[l], mp2[l], &ee, &ks, &ka, &pa, &pb, &bad,
                 __builtin_expect(b.liquid_water != nullptr, 0) ? b.liquid_water[(gp % b.S) * b.Lmax + l] : 0.0);
        s.eps_re[l] = ee.re; s.eps_im[l] = ee.im; s.ks[l] = ks; s.ka[l] = ka; s.pa[l] = pa; s.pb[l] = pb;
        s.pc[l] = (double)(((kind & 15) == EM_IBA_INV || (kind & 15) == EM_IBA_HOST) ? (kind & ~15) | EM_IBA   // the phase function is IBA's either way
                           : (kind & 15) == EM_RAYLEIGH_HOST ? (kind & ~15) | EM_DMRT : kind)                 // ... or Rayleigh's
                  + krho;   // (MS_EXPC / MS_TSC: Im / Re of the squared wavenumber, < 0.5, rides in the fraction; 0 otherwise)
        s.slab_re[l] = s.slab_im[l] = s.slab_th[l] = 0.0; s.lo[l] = (double)l;
        s.thick[l] = thickness[l];
        s.BT[l] = b.rayleigh_jeans ? temperature[l] : planck_radiance(frequency, temperature[l]);
        if (bad || !(ks >= 0.0)) lds_max(&s.ints[0], ST_INPUT);
    }
    for (int j = t; j < nmax; j += NT) {
        const double m = b.gl_mu[j];
        s.gmu[j] = m; s.gsin[j] = sqrt(1.0 - m * m);
    }
    if (t == 0) s.ints[6] = L;
    block_sync();
    if (s.ints[0] != ST_OK) return s.ints[0];
    if (b.coherent) {
        // DORT option process_coherent_layers (smrt/interface/coherent_flat.py:16-57): a layer with k Re(n) d < 3 pi / 4
        // at this frequency is taken out and becomes the slab of the interface on top of the layer that follows it;
        // the per-layer tables are compacted in place (lo[] keeps the index in the input arrays), everything after
        // this point -- streams, staging items, the layer loops of the three kernels -- sees the shorter snowpack
        if (t == 0) {
            const double k0 = 2.0 * kPi * frequency / kCSpeed;
            int j = 0, bad = 0;
            bool pending = false, prev = false;
            double pr = 0.0, pi = 0.0, pth = 0.0;
            for (int l = 0; l < L; ++l) {
                const cplx e = cmk(s.eps_re[l], s.eps_im[l]);
                const bool coh = k0 * csqrt_(e).re * s.thick[l] < 0.75 * kPi;
                if (coh) {
                    if (l == L - 1 || prev) { bad = 1; break; }   // the last layer / two in a row: not supported (:26,:34)
                    pr = e.re; pi = e.im; pth = s.thick[l]; pending = true;
                } else {
                    s.eps_re[j] = s.eps_re[l]; s.eps_im[j] = s.eps_im[l]; s.ks[j] = s.ks[l]; s.ka[j] = s.ka[l];
                    s.pa[j] = s.pa[l]; s.pb[j] = s.pb[l]; s.pc[j] = s.pc[l]; s.BT[j] = s.BT[l]; s.thick[j] = s.thick[l];
                    s.slab_re[j] = pending ? pr : 0.0; s.slab_im[j] = pending ? pi : 0.0; s.slab_th[j] = pending ? pth : 0.0;
                    s.lo[j] = (double)l;
                    pending = false;
                    ++j;
                }
                prev = coh;
            }
            s.ints[6] = j;
            if (bad) s.ints[0] = ST_COHERENT;
        }
        block_sync();
        if (s.ints[0] != ST_OK) return s.ints[0];
        L = s.ints[6];
    }
    if (t == 0) {
        int ks_ = 0;
        for (int l = 1; l < L; ++l)  // np.argmax on complex: lexicographic, first maximum
            if (s.eps_re[l] > s.eps_re[ks_] || (s.eps_re[l] == s.eps_re[ks_] && s.eps_im[l] > s.eps_im[ks_])) ks_ = l;
        s.ints[4] = ks_;
    }
    block_sync();
    {
        const cplx estar = cmk(s.eps_re[s.ints[4]], s.eps_im[s.ints[4]]);
        for (int l = t; l < L; l += NT) {
            const double ri = csqrt_(cdiv(estar, cmk(s.eps_re[l], s.eps_im[l]))).re;
            int n = 0;
            for (int j = 0; j < nmax; ++j) n += (ri * s.gsin[j] < 1.0) ? 1 : 0;
            s.ri[l] = ri; s.nl[l] = (double)n;
            if (n < 2) lds_max(&s.ints[0], ST_INPUT);
            // the phase matrix of a host-evaluated layer was sampled on the caller's streams: same count or nothing
            if (((int)s.pc[l] & 15) == EM_HOST && !(b.host_streams && b.host_phase && b.host_streams[gp * b.Lmax + (int)s.lo[l]] == n))
                lds_max(&s.ints[0], ST_INPUT);
        }
        if (t == NT - 1) {
            const double ria = csqrt_(estar).re;
            int n = 0;
            for (int j = 0; j < nmax; ++j) {
                const double rs = ria * s.gsin[j];
                if (rs < 1.0) { s.outmu[n] = sqrt(1.0 - rs * rs); ++n; }
            }
            s.ints[5] = n;
            if (n < 1) lds_max(&s.ints[0], ST_INPUT);
        }
    }
    block_sync();
    return s.ints[0];
}

#ifdef SMRT_EMU_DEBUG
#include <cstdio>
#define SMRT_DUMP(tag, M, NN) do { block_sync(); if (t == 0) { char fn[128]; snprintf(fn, 128, "/tmp/dump_l%d_%s.bin", l, tag); FILE* f = fopen(fn, "wb"); for (int c_ = 0; c_ < (NN); ++c_) fwrite((M) + c_ * LD, 8, (NN), f); fclose(f);} block_sync(); } while (0)
#else
#define SMRT_DUMP(tag, M, NN) do {} while (0)
#endif
// Optional per-stage cycle accounting (profiling builds only): thread 0 accumulates s_memtime deltas.
#ifdef SMRT_STAGE_TIMING
#define SMRT_STAGE(k) do { const long long now_ = cycle_counter(); stage_acc[stage_cur] += (double)(now_ - stage_t0); stage_t0 = now_; stage_cur = (k); } while (0)
#else
#define SMRT_STAGE(k) do {} while (0)
#endif
enum { SG_SETUP = 0, SG_ASSEMBLE, SG_CHOL, SG_BTL, SG_JACOBI, SG_TRI, SG_R1, SG_LU1, SG_R45, SG_LU2, SG_R78, SG_OUT, SG_COUNT };

// MODE 0: the whole solve in one workgroup (fused).  MODE 1 ("prep"): per layer assemble X+-, factorise, form
// B = L+^T L- and park L+, B, d in the staging area.  MODE 2 ("finish"): pick up L+, B' (rotated by the Jacobi
// kernel) and the singular values, build the eigenvectors and run the layer recursion.
// MODE 3 ("finish", two LDS slots): the same recursion with only two N x N matrices in LDS, so that TWO workgroups
// share a CU (the Gauss-Jordan panels are wavefront-serial: a second resident workgroup fills the idle SIMDs).
//   slot X: B' -> Ep' -> Wk (matrix of solve 1) -> t Q t -> W (matrix of solve 2) -> K
//   slot R: R~ (carried between layers) -> right-hand side of solve 1 -> Y (right-hand side of solve 2) -> next R~
//   global: L+ is used where it lies in the staging area; Em' -> G overwrites the item's B slot, F the item's L
//   slot (both dead by then); the 16x16 diagonal-block inverses of L+ come from the prep kernel.
template <int NT, int CH, int MODE = 0>
SMRT_DEV void dort_pair_passive(const DevBatch& b, long long p, double* lds_base, double* gmem_mat = nullptr,
                                const DevStage* stg = nullptr) {
    constexpr int P = 2;
    constexpr int JW = (NT / SMRT_LANES >= 4) ? 4 : NT / SMRT_LANES;  // wavefronts rotating columns (one per SIMD)
    constexpr int GS = 8;                                             // lanes per Jacobi column pair
    constexpr int RPL = (64 * CH + GS - 1) / GS;                      // rows per lane (N <= 64 CH)
    const int t = tid();
    const int lane = t % SMRT_LANES, wave = t / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    const int nphi = 9;  // m_max = 0 -> 16 azimuth samples (emmodel/common.py:401-414), 9 distinct by symmetry
    const LdsPlan plan = make_plan(b.n_max_stream, P, b.Lmax, b.n_theta, nphi, gmem_mat == nullptr ? 1 : 0, 0,
                                   MODE == 1 ? (CH == 1 ? 3 : 1) : (MODE == 3 ? 2 : 0),
                                   (gmem_mat != nullptr && MODE != 1) ? ((CH > 2 && MODE == 2) ? 3 : (MODE == 2 && b.jac_in_lds ? 2 : b.jac_in_lds)) : 0);
    Lds s = carve(lds_base, gmem_mat == nullptr ? lds_base : gmem_mat, plan);
    // matrix-core variants of the dense steps: always on the LDS path; on the global-workspace path for N <= 128 when
    // the LDS Jacobi buffer exists (it doubles as the scratch of the blocked Cholesky / triangular solve)
    // (the prep half only needs the 512-double Cholesky scratch, which its slim plan has)
    // N > 128 (CH > 2): always, with the scratch of the blocked solvers behind the work matrices in the global workspace
    const bool dense_mfma = (CH == 1) || (CH == 2 && (plan.o_jac >= 0 || MODE == 1)) || (CH > 2);
    double* dense_scratch = (CH > 2) ? gmem_mat + plan.mat_doubles
                                     : ((CH == 1 || MODE == 1) ? s.gj : lds_base + (plan.o_jac >= 0 ? plan.o_jac : 0));
#ifdef SMRT_STAGE_TIMING
    double sub_acc_store[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    s.sub_acc = sub_acc_store;
#endif
    // N > 128 finish kernel: LDS staging buffers of the matrix-core products (make_plan jac_in_lds = 3)
    double* big_stage = (CH > 2 && MODE == 2 && plan.o_jac >= 0) ? lds_base + plan.o_jac : nullptr;
    const int LD = plan.LD;
    const int nmax = b.n_max_stream;
    const int out_stride = P * b.n_theta;
    // the LDS-resident prep kernel keeps X+- / L+- as packed lower triangles (make_plan slim = 3)
    constexpr bool PK = (MODE == 1 && CH == 1);

    const long long gp = global_pair(b, p);
    const int fi = (int)(gp / b.S), si = (int)(gp % b.S);
    const double frequency = b.frequency[fi];
    int L = b.n_layers[si];
    const double* thickness = b.thickness + (long long)si * b.Lmax;
    const double* fracvol = b.frac_volume + (long long)si * b.Lmax;
    const double* temperature = b.temperature + (long long)si * b.Lmax;
    const double* mp1 = b.p1 + (long long)si * b.Lmax;
    const double* mp2 = b.p2 + (long long)si * b.Lmax;

#ifdef SMRT_STAGE_TIMING
    double stage_acc[SG_COUNT];
    for (int k = 0; k < SG_COUNT; ++k) stage_acc[k] = 0.0;
    long long stage_t0 = cycle_counter();
    int stage_cur = SG_SETUP;
#endif
    // ---- stage 0: layer scalars, azimuth table, Gauss-Legendre sines -------------------------------------
    if (t < 8) s.ints[t] = 0;
    block_sync();
    if (MODE >= 2) {  // a failure recorded by the prep or Jacobi kernel
        const int prev = b.status[p];
        if (prev != ST_OK) { fail_pair<NT>(b, p, prev, out_stride); return; }
    }
    for (int k = t; k < nphi && MODE < 2; k += NT) {  // azimuth table of the phase-matrix assembly
        const double ph = kPi * (double)k / (double)(nphi - 1);
        const double c = cos(ph), sn = sin(ph);
        s.cphi[k] = c; s.s2phi[k] = sn * sn;
        s.wphi[k] = ((k == 0 || k == nphi - 1) ? 1.0 : 2.0) / (double)(2 * (nphi - 1));
    }
    {
        const int st = pair_setup<NT>(b, s, frequency, L, thickness, fracvol, temperature, mp1, mp2,
                                      b.layer_kind ? b.layer_kind + (long long)si * b.Lmax : nullptr, gp);
        if (st != ST_OK) { fail_pair<NT>(b, p, st, out_stride); return; }
        L = s.ints[6];   // fewer than the snowpack's under process_coherent_layers
    }
    const int n_air = s.ints[5];

    if (MODE != 1 && b.want_layer_out) {
        double* lo = b.layer_out + p * (long long)b.Lmax * 5;
        for (int l = t; l < b.Lmax; l += NT) {
            const bool in = l < L;
            lo[l * 5 + 0] = in ? s.eps_re[l] : 0.0; lo[l * 5 + 1] = in ? s.eps_im[l] : 0.0;
            lo[l * 5 + 2] = in ? s.ks[l] : 0.0; lo[l * 5 + 3] = in ? s.ka[l] : 0.0;
            lo[l * 5 + 4] = in ? s.nl[l] + (b.coherent ? 1024.0 * s.lo[l] : 0.0) : 0.0;   // + 1024 x index in the input (smrt_dort.h)
        }
    }
    if (MODE != 1 && b.want_stream_out) {
        double* so = b.stream_out + p * (long long)(1 + nmax);
        if (t == 0) so[0] = (double)n_air;
        for (int j = t; j < nmax; j += NT) so[1 + j] = (j < n_air) ? s.outmu[j] : 0.0;
    }

    double n3 = 0.0;
    int n_sweeps = 0;
    // layers kept by prune_deep_snowpack (finish kernels: s.pa is free there)
    int Lk = L;
    if (MODE >= 2 && b.prune_tau > 0.0) Lk = pruned_layer_count<NT>(*stg, p * (long long)b.Lmax, L, s.thick, s.pa, b.prune_tau);
    if (MODE >= 2) {
        const int bad = first_failed_layer(*stg, p * (long long)b.Lmax, Lk);
        if (bad != ST_OK) { fail_pair<NT>(b, p, bad, out_stride); return; }
    }
    // prep kernel: a layer that cannot be diagonalised is recorded and skipped (uniform)
    auto layer_failed = [&](int l, int code) {
        block_sync();
        if (t == 0) { stg->n[p * (long long)b.Lmax + l] = -code; s.ints[0] = ST_OK; }
        block_sync();
    };
    bool rough_surface = false;   // the surface is a rough interface: R~ (air side) and s are in M3 / svec after the loop
    // ---- bottom-up over the layers -------------------------------------------------------------------------
    if (MODE == 1 && b.pair_done && b.pair_done[p]) return;   // the cut of this pair lies above this round's layers (uniform)
    for (int l = Lk - 1; l >= 0; --l) {
        if (MODE == 1 && (l < b.layer_lo || l >= b.layer_hi)) continue;   // not in this round of the prep kernel
        const int n = (int)s.nl[l];
        const int N = n * P;
        n3 += (double)N * N * N;
        const cplx el = cmk(s.eps_re[l], s.eps_im[l]);
        const double ks = s.ks[l], ke = s.ks[l] + s.ka[l];
        const double Bl = s.BT[l];
        const int nu = (l > 0) ? (int)s.nl[l - 1] : 0;
        const int Nu = nu * P;

        SMRT_STAGE(SG_SETUP);
        // -- stream cosines of this layer and of the layer above
        for (int j = t; j < n; j += NT) { const double rs = s.ri[l] * s.gsin[j]; s.mu[j] = sqrt(1.0 - rs * rs); }
        if (l > 0)
            for (int j = t; j < nu; j += NT) { const double rs = s.ri[l - 1] * s.gsin[j]; s.muu[j] = sqrt(1.0 - rs * rs); }
        if (MODE != 1 && l == Lk - 1) {
            // what the last layer sees below: nothing (rtsolver_utils.py:548-551,601-603), or a substrate: specular
            // reflection R_sub on the diagonal and its emission (1 - R_sub) B(T_sub) (rtsolver_utils.py:544-547,
            // 579-584; dort.py:429-441); or, when deeper layers were pruned, the reflection of the interface to the
            // first dropped layer and nothing coming up through it (dort.py:446-452)
            for_2d<NT>(N, N, [&](int r, int c) { s.M3[c * LD + r] = 0.0; });
            block_sync();
            for (int r = t; r < N; r += NT) {
                double Rs = 0.0, src = 0.0;
                if (Lk < L) {
                    const double rs = s.ri[l] * s.gsin[r >> 1];
                    double Rv, Rh, Tv, Th;
                    interface_RT(frequency, el, cmk(s.eps_re[l + 1], s.eps_im[l + 1]), sqrt(1.0 - rs * rs),
                                 cmk(s.slab_re[l + 1], s.slab_im[l + 1]), s.slab_th[l + 1], &Rv, &Rh, &Tv, &Th);
                    Rs = (r & 1) ? Rh : Rv;
                } else if (b.sub_kind == SUB_HOST) {
                    // rough substrate evaluated by the caller (smrt_dort.h): dense reflection matrix of mode 0 (its other
                    // entries are copied below) and the emissivity diagonal
                    const int NE = 3 * nmax;
                    Rs = b.host_substrate[gp * (long long)NE * NE + (long long)r * NE + r];
                    const double Ts = b.sub_T[si];
                    if (Ts > 0.0) src = b.host_substrate_coh[gp * (long long)NE + r] * (b.rayleigh_jeans ? Ts : planck_radiance(frequency, Ts));
                } else if (b.sub_kind != SUB_NONE) {
                    const long long gpi = gp;
                    const double q1 = b.sub_p1[gpi], q2 = b.sub_p2[gpi];
                    if (b.sub_kind == SUB_FLAT) {
                        const double rs = s.ri[l] * s.gsin[r >> 1];
                        double Rv, Rh;
                        fresnel_RvRh(el, cmk(q1, q2), sqrt(1.0 - rs * rs), &Rv, &Rh);
                        Rs = (r & 1) ? Rh : Rv;
                    } else Rs = (r & 1) ? q2 : q1;
                    const double Ts = b.sub_T[si];
                    if (Ts > 0.0) src = (1.0 - Rs) * (b.rayleigh_jeans ? Ts : planck_radiance(frequency, Ts));
                }
                s.M3[r * LD + r] = Rs;
                s.svec[r] = src;
            }
            {
                // a DENSE start of the recursion, from the caller (smrt_dort.h): the other entries of a rough substrate's
                // reflection matrix, or -- deeper layers pruned right above a rough interface -- that interface's dense
                // reflection seen from this layer (Rbot of its slot: specular + diffuse), what the reference's truncated
                // system keeps (dort.py:443-452, rtsolver_utils.py:567-597).  One copy loop for both.
                const long long NE = 3LL * nmax;
                const double* H = nullptr;
                if (Lk == L) { if (b.sub_kind == SUB_HOST) H = b.host_substrate + gp * NE * NE; }
                else if (__builtin_expect(b.host_itf_slot != nullptr, 0)) {
                    const int hsb = b.host_itf_slot[gp * b.Lmax + (int)s.lo[l + 1]];
                    if (hsb >= 0) H = b.host_itf + ((gp * b.host_itf_slots + hsb) * 4 + 2) * NE * NE;
                }
                if (__builtin_expect(H != nullptr, 0)) {
                    block_sync();
                    for_2d<NT>(N, N, [&](int r, int c) { s.M3[c * LD + r] = H[r * NE + c]; });
                }
            }
        }
        block_sync();
        // -- weights (streams.py:324-330), per-row copies, interface diagonals
        for (int j = t; j < n; j += NT) {
            double w;
            if (j == 0) w = 1.0 - 0.5 * (s.mu[0] + s.mu[1]);
            else if (j == n - 1) w = fabs(0.5 * (s.mu[n - 2] + s.mu[n - 1]));
            else w = fabs(0.5 * (s.mu[j - 1] - s.mu[j + 1]));
            s.w[j] = w;
            if (MODE < 2) {
                s.mrow[2 * j] = s.mu[j]; s.mrow[2 * j + 1] = s.mu[j];
                s.wrow[2 * j] = w; s.wrow[2 * j + 1] = w;
            }
            if (MODE == 1) continue;  // the interfaces belong to the finish kernel
            double Rv, Rh, Tv, Th;
            const cplx eup = (l > 0) ? cmk(s.eps_re[l - 1], s.eps_im[l - 1]) : cmk(1.0, 0.0);
            const cplx slab = cmk(s.slab_re[l], s.slab_im[l]);   // a coherent layer collapsed into this interface, if any
            interface_RT(frequency, el, eup, s.mu[j], slab, s.slab_th[l], &Rv, &Rh, &Tv, &Th);
            s.Rtop[2 * j] = Rv; s.Rtop[2 * j + 1] = Rh;
            s.Ttop[2 * j] = Tv; s.Ttop[2 * j + 1] = Th;
        }
        if (MODE != 1 && l > 0)
            for (int j = t; j < nu; j += NT) {
                double Rv, Rh, Tv, Th;
                interface_RT(frequency, cmk(s.eps_re[l - 1], s.eps_im[l - 1]), el, s.muu[j], cmk(s.slab_re[l], s.slab_im[l]),
                             s.slab_th[l], &Rv, &Rh, &Tv, &Th);
                s.Rbu[2 * j] = Rv; s.Rbu[2 * j + 1] = Rh;
                s.Tbu[2 * j] = Tv; s.Tbu[2 * j + 1] = Th;
            }

        // a rough interface on top of this layer (evaluated by the caller): the layer step runs with a transparent top and
        // the interface is composed afterwards (dort_interface_dense.hpp)
        const int hs = (MODE != 1) ? host_interface_slot(b, gp, (int)s.lo[l]) : -1;
        if (hs >= 0) {
            block_sync();
            for (int r = t; r < N; r += NT) { s.Rtop[r] = 0.0; s.Ttop[r] = 1.0; s.Rbu[r] = 0.0; s.Tbu[r] = 1.0; }
        }
        SMRT_STAGE(SG_ASSEMBLE);
        if (MODE < 2) {
        // (a layer the Rayleigh kernel will diagonalise in closed form needs neither matrix: its row sums below are two
        //  sums over the streams -- 18 -> ~4 ms of prep kernel per 7168 solves of the configs[2] shape)
        const bool ray_direct = MODE == 1 && b.rayleigh_direct && stg->Linv && em_has_rayleigh_phase((int)s.pc[l] & 15);   // (uniform)
        // -- phase matrix, azimuth mode 0: S+ = P(mu,+mu') + P(mu,-mu') -> M0, S- = P(+) - P(-) -> M1
        //    (lower triangle by stream blocks; the matrices are symmetric)
        if (!ray_direct) {
            const int T = n * (n + 1) / 2;
            const double pa = s.pa[l], pb = s.pb[l];
            const int lo = (int)s.lo[l];   // the layer's index in the input arrays
            const int em_l = (int)s.pc[l] & 15, ms_l = (int)s.pc[l] >> 4;   // this layer's emmodel and microstructure
            const double fv = fracvol[lo], q1 = mp1[lo], q2 = mp2[lo];
            const double krho = __builtin_expect(ms_l >= MS_EXPC, 0) ? s.pc[l] - (double)(int)s.pc[l] : 0.0;   // Im / Re of k^2 (pair_setup)
            for (int idx = t; idx < T; idx += NT) {
                int i = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
                while ((i + 1) * (i + 2) / 2 <= idx) ++i;
                while (i * (i + 1) / 2 > idx) --i;
                const int j = idx - i * (i + 1) / 2;
                const double mi = s.mu[i], mj = s.mu[j];
                double pvv_p, pvh_p, phv_p, phh_p, pvv_m, pvh_m, phv_m, phh_m;
                if (em_l == EM_HOST) {  // mode 0 of the caller's ft_even_phase(mu, +-mu'), compressed (smrt_dort.h)
                    const int NE = b.host_ne;
                    const double* hp = b.host_phase + ((gp * b.Lmax + lo) * (long long)b.host_modes) * 2 * NE * NE;   // lo: the layer's index in the input
                    const double* hm = hp + (long long)NE * NE;
                    const int r0 = 2 * i, c0 = 2 * j;
                    pvv_p = hp[r0 * NE + c0]; pvh_p = hp[r0 * NE + c0 + 1];
                    phv_p = hp[(r0 + 1) * NE + c0]; phh_p = hp[(r0 + 1) * NE + c0 + 1];
                    pvv_m = hm[r0 * NE + c0]; pvh_m = hm[r0 * NE + c0 + 1];
                    phv_m = hm[(r0 + 1) * NE + c0]; phh_m = hm[(r0 + 1) * NE + c0 + 1];
                } else if (em_l != EM_IBA) {  // closed form, rayleigh.py:70-76; even in mu'
                    const double a2 = mi * mi, b2 = mj * mj;
                    pvv_p = pa * (0.5 * a2 * b2 + (1.0 - a2) * (1.0 - b2));
                    pvh_p = pa * 0.5 * a2; phv_p = pa * 0.5 * b2; phh_p = pa * 0.5;
                    pvv_m = pvv_p; pvh_m = pvh_p; phv_m = phv_p; phh_m = phh_p;
                } else {
                    const double sisj = sqrt(1.0 - mi * mi) * sqrt(1.0 - mj * mj);
                    const double mm = mi * mj;
                    const double a2 = mi * mi, b2 = mj * mj;
                    pvv_p = pvh_p = phv_p = phh_p = pvv_m = pvh_m = phv_m = phh_m = 0.0;
                    for (int k = 0; k < nphi; ++k) {
                        const double c = s.cphi[k], s2 = s.s2phi[k], wk = s.wphi[k];
                        double ct_p = mm + sisj * c;       // cos(scattering angle), mu' = +mu_j
                        double ct_m = -mm + sisj * c;      // mu' = -mu_j
                        // (only these two bounds can be crossed, by rounding, at mu = mu': cos(ti - tj) and -cos(ti - tj);
                        // the other two are +-cos(ti + tj) with ti + tj < pi)
                        ct_p = ct_p > 1.0 ? 1.0 : ct_p;
                        ct_m = ct_m < -1.0 ? -1.0 : ct_m;
                        double Cp, Cm;
                        if (ms_l == MS_EXP) {
                            // pa wk / dp^2 and pa wk / dm^2 with ONE reciprocal (quarter rate, plus its Newton steps)
                            const double dp = 1.0 + pb * (1.0 - ct_p), dm = 1.0 + pb * (1.0 - ct_m);
                            const double dp2 = dp * dp, dm2 = dm * dm;
                            const double q = (pa * wk) * fast_rcp(dp2 * dm2);
                            Cp = dm2 * q; Cm = dp2 * q;
                        } else {
                            Cp = pa * wk * ft_corr(ms_l, pb * (1.0 - ct_p), fv, q1, q2, krho);
                            Cm = pa * wk * ft_corr(ms_l, pb * (1.0 - ct_m), fv, q1, q2, krho);
                        }
                        const double fvv_p = c * mm + sisj, fvv_m = -c * mm + sisj;
                        pvv_p += fvv_p * fvv_p * Cp; pvv_m += fvv_m * fvv_m * Cm;
                        const double s2p = s2 * Cp, s2m = s2 * Cm;          // (a2, b2 are applied once, after the loop)
                        pvh_p += s2p; pvh_m += s2m;
                        const double c2 = c * c;
                        phh_p += c2 * Cp; phh_m += c2 * Cm;
                    }
                    phv_p = b2 * pvh_p; phv_m = b2 * pvh_m;
                    pvh_p *= a2; pvh_m *= a2;
                }
                const int r0 = 2 * i, c0 = 2 * j;
                s.M0[sidx<PK>(r0, c0, LD)] = pvv_p + pvv_m;             s.M1[sidx<PK>(r0, c0, LD)] = pvv_p - pvv_m;
                if (!PK || i > j) {   // (V, H) of a diagonal stream block lies above the diagonal: its mirror (H, V) is written below
                    s.M0[sidx<PK>(r0, c0 + 1, LD)] = pvh_p + pvh_m;     s.M1[sidx<PK>(r0, c0 + 1, LD)] = pvh_p - pvh_m;
                }
                s.M0[sidx<PK>(r0 + 1, c0, LD)] = phv_p + phv_m;         s.M1[sidx<PK>(r0 + 1, c0, LD)] = phv_p - phv_m;
                s.M0[sidx<PK>(r0 + 1, c0 + 1, LD)] = phh_p + phh_m;     s.M1[sidx<PK>(r0 + 1, c0 + 1, LD)] = phh_p - phh_m;
            }
        }
        block_sync();
        // -- energy-conserving renormalisation (dort.py:782-819): norm_r = ks / (c sum_c S+[r,c] w_c), c = 1/2
        for (int r = t; r < N; r += NT) {
            double rs = 0.0;
            if (ray_direct) {
                // S+ = 2 pa (u1 u1^T / 2 + u2 u2^T) (rayleigh.py:70-76): row (i, V) sums to pa (a W2 + 2 (1 - a)(W0 - W2) + a W0),
                // a = mu_i^2, row (i, H) to pa (W2 + W0), with W0 = sum_j w_j, W2 = sum_j w_j mu_j^2
                double W0 = 0.0, W2 = 0.0;
                for (int j = 0; j < n; ++j) { const double wj = s.wrow[2 * j], mj = s.mu[j]; W0 += wj; W2 += wj * mj * mj; }
                const double a2 = s.mu[r >> 1] * s.mu[r >> 1];
                rs = s.pa[l] * ((r & 1) ? (W2 + W0) : (a2 * W2 + 2.0 * (1.0 - a2) * (W0 - W2) + a2 * W0));
            } else {
            for (int c = 0; c <= r; ++c) rs += s.M0[sidx<PK>(r, c, LD)] * s.wrow[c];
            for (int c = r + 1; c < N; ++c) rs += s.M0[sidx<PK>(c, r, LD)] * s.wrow[c];
            }
            double nr = 1.0;
            if (b.normalization != 0 && ks != 0.0) {
                nr = ks / (0.5 * rs);
                if (b.normalization == 1 && !(fabs(nr - 1.0) <= 0.3)) lds_max(&s.ints[0], ST_NORM);
            }
            const double uu = sqrt(nr * s.wrow[r] / s.mrow[r]);
            s.u[r] = uu;
            s.d[r] = uu / s.wrow[r];
        }
        block_sync();
        if (s.ints[0] != ST_OK) {
            if (MODE == 1) { layer_failed(l, s.ints[0]); continue; }
            fail_pair<NT>(b, p, s.ints[0], out_stride); return;
        }
        if (ray_direct) {
            // a layer with a Rayleigh phase matrix: X- is diagonal and X+ diagonal minus rank two -- no Cholesky, no B;
            // dort_rayleigh_kernel.hpp diagonalises it from these scalars (its header has the layout of the slot)
            const long long item = p * (long long)b.Lmax + l;
            double* gI = stg->Linv + item * stg->linv_stride;
            for (int r = t; r < N; r += NT) { gI[2 + n + r] = s.u[r]; stg->d[item * stg->vec_stride + r] = s.d[r]; }
            for (int j = t; j < n; j += NT) gI[2 + j] = s.mu[j];
            if (t == 0) { gI[0] = ke; gI[1] = s.pa[l]; stg->n[item] = N + kStageDirect; }
            block_sync();
            continue;
        }
        // -- X+- = M^-1/2 T (ke I - c N S+- W) T^-1 M^-1/2, symmetric positive definite (lower triangles)
        for_2d<NT>(N, N, [&](int r, int c) {
            if (r >= c) {
                const double uu = 0.5 * s.u[r] * s.u[c];
                const double dg = (r == c) ? ke / s.mrow[r] : 0.0;
                s.M0[sidx<PK>(r, c, LD)] = dg - uu * s.M0[sidx<PK>(r, c, LD)];
                s.M1[sidx<PK>(r, c, LD)] = dg - uu * s.M1[sidx<PK>(r, c, LD)];
            }
        });
        block_sync();
        SMRT_STAGE(SG_CHOL);
        if (!(dense_mfma ? chol2_mfma<NT, PK>(s.M0, s.M1, dense_scratch, &s.ints[2], N, LD,
                                       (MODE == 1 && CH <= 2 && stg->Linv) ? stg->Linv + (p * (long long)b.Lmax + l) * stg->linv_stride : nullptr)
                      : chol2<NT>(s.M0, s.M1, N, LD))) {
            if (MODE == 1) { layer_failed(l, ST_ALBEDO); continue; }
            fail_pair<NT>(b, p, ST_ALBEDO, out_stride); return;
        }
        SMRT_STAGE(SG_BTL);
        if (MODE == 1) {  // B = L+^T L- straight from the accumulators into the staging area
            lt_times_l_mfma<NT, PK>(s.M0, s.M1, stg->B + (p * (long long)b.Lmax + l) * stg->mat_stride, N, LD,
#ifdef SMRT_NO_COLUMN_REVERSAL
                                false);
#else
                                true);
#endif
        } else {
        if (dense_mfma) lt_times_l_mfma<NT>(s.M0, s.M1, s.M2, N, LD);     // B = L+^T L-
        else lt_times_l<NT>(s.M0, s.M1, s.M2, N, LD);
        }
        }  // MODE < 2
        if (MODE == 1) {  // park L+ and d for the finish kernel (B is already there)
            const long long item = p * (long long)b.Lmax + l;
            double* gL = stg->L + item * stg->mat_stride;
            for_2d<NT>(N, N, [&](int r, int c) { gL[c * LD + r] = (PK && r < c) ? 0.0 : s.M0[sidx<PK>(r, c, LD)]; });
            for (int r = t; r < N; r += NT) stg->d[item * stg->vec_stride + r] = s.d[r];
            if (t == 0) stg->n[item] = N;
            block_sync();
            continue;
        }
        SMRT_STAGE(SG_JACOBI);
        if (MODE == 0) {
            // global-workspace kernels: the rotations run on an LDS copy of B when one fits next to the vectors
            double* Jm = (plan.o_jac >= 0) ? lds_base + plan.o_jac : s.M2;
            if (Jm != s.M2) { for_2d<NT>(N, N, [&](int r, int c) { Jm[c * LD + r] = s.M2[c * LD + r]; }); block_sync(); }
            if (!jacobi_onesided<NT, JW, GS, RPL>(Jm, N, LD, s.sigma, s.rsig, &s.ints[1], &n_sweeps, s.sub_acc)) {
                fail_pair<NT>(b, p, ST_EIGEN, out_stride); return;
            }
            if (Jm != s.M2) { for_2d<NT>(N, N, [&](int r, int c) { s.M2[c * LD + r] = Jm[c * LD + r]; }); block_sync(); }
        } else {  // MODE 2 / 3: pick up L+, B' = B V, d and the singular values
            const long long item = p * (long long)b.Lmax + l;
            const double* gL = stg->L + item * stg->mat_stride;
            const double* gB = stg->B + item * stg->mat_stride;
            if (MODE == 3) for_2d<NT>(N, N, [&](int r, int c) { s.M0[c * LD + r] = gB[c * LD + r]; });   // B' -> slot X
            else for_2d<NT>(N, N, [&](int r, int c) { s.M0[c * LD + r] = gL[c * LD + r]; s.M2[c * LD + r] = gB[c * LD + r]; });
            for (int r = t; r < N; r += NT) {
                s.d[r] = stg->d[item * stg->vec_stride + r];
                const double sg = stg->sigma[item * stg->vec_stride + r];
                s.sigma[r] = sg; s.rsig[r] = 1.0 / sg;
            }
            block_sync();
        }
        SMRT_STAGE(SG_TRI);
        double* F = s.M2; double* G = s.M1; double* Rt = s.M3; double* Wk = s.M0;
        double r1a[RowTiles<NT>::RPW][16];   // MODE 3: rows of R~ of this wavefront's row tile (A operands of R1)
        if (MODE == 3) {
            const long long item = p * (long long)b.Lmax + l;
            double* gL = stg->L + item * stg->mat_stride;   // L+, later F
            double* gB = stg->B + item * stg->mat_stride;   // (B' is in slot X by now) Em', later G
            // R~ goes into registers now, which frees slot R for L+ during the triangular stage (its transposed walk in
            // the solve is uncoalesced in global memory) and, after that, for the LDS copy of F
            r1_load<NT>(s.M3, r1a, s.cvec, s.svec, Bl, N, LD);
            for_2d<NT>(N, N, [&](int r, int c) { s.M3[c * LD + r] = gL[c * LD + r]; });
            block_sync();
            l_times_m_mfma<NT>(s.M3, s.M0, gB, N, LD);                                          // Em' = L+ B'
            lt_solve_mfma<NT>(s.M3, s.M0, stg->Linv + item * stg->linv_stride, N, LD, true);    // Ep' = L+^-T B'
            // F, G to global memory (A operands and elementwise terms of the second GEMM pass) and to slots R, X
            // (B operands of the first one)
            for_2d<NT>(N, N, [&](int i, int c) {
                const double ep = s.M0[c * LD + i], em = gB[c * LD + i] * s.rsig[c];
                const double hd = 0.5 * s.d[i];
                const double fv = hd * (ep + em), gv = hd * (ep - em);
                gL[c * LD + i] = fv; s.M3[c * LD + i] = fv;
                gB[c * LD + i] = gv; s.M0[c * LD + i] = gv;
            });
            F = gL; G = gB;
        } else {
        if (dense_mfma) l_times_m_mfma<NT>(s.M0, s.M2, s.M1, N, LD);      // Em' = L+ B'
        else l_times_m<NT>(s.M0, s.M2, s.M1, N, LD);
        if (dense_mfma) lt_solve_mfma<NT>(s.M0, s.M2, dense_scratch, N, LD);       // Ep' = L+^-T B'
        else lt_solve<NT>(s.M0, s.M2, N, LD);
        // -- F = (Ep - Em)/2 -> M2, G = (Ep + Em)/2 -> M1, with Ep = d Ep', Em = -d Em' / sigma
        for_2d<NT>(N, N, [&](int i, int c) {
            const double ep = s.M2[c * LD + i], em = s.M1[c * LD + i] * s.rsig[c];
            const double hd = 0.5 * s.d[i];
            s.M2[c * LD + i] = hd * (ep + em);
            s.M1[c * LD + i] = hd * (ep - em);
        });
        }
        for (int c = t; c < N; c += NT) s.t[c] = exp(-s.sigma[c] * s.thick[l]);
        block_sync();
        SMRT_DUMP("F", F, N); SMRT_DUMP("G", G, N); SMRT_DUMP("Rt", Rt, N);

        SMRT_STAGE(SG_R1);
        if (MODE == 3) {
            r1_compute<NT>(s.M3, s.M0, r1a, N, LD);   // Wk -> slot X, R~ F - G -> slot R
        } else if (CH == 1) {
            r1_mfma<NT>(F, G, Rt, Wk, s.cvec, s.svec, Bl, N, LD);
        } else if (CH > 2 && big_stage) {   // one staged operand matrix per pass (dort_dense.hpp)
            r1_mfma_big<NT, 16 * CH, 1>(F, G, Rt, Wk, s.cvec, s.svec, Bl, N, LD, big_stage, plan.stage_bufs, 64 * ((plan.NMAX + 3) / 4));
            r1_mfma_big<NT, 16 * CH, 2>(F, G, Rt, Wk, s.cvec, s.svec, Bl, N, LD, big_stage, plan.stage_bufs, 64 * ((plan.NMAX + 3) / 4));
        } else if (CH >= 2 && dense_mfma) {
            r1_mfma_big<NT, 16 * CH>(F, G, Rt, Wk, s.cvec, s.svec, Bl, N, LD);
        } else {
            r1_rows<NT, CH>(F, G, Rt, Wk, s.cvec, s.svec, Bl, N, LD);
        }
        SMRT_DUMP("M1", Wk, N); SMRT_DUMP("RHS", Rt, N);
        SMRT_STAGE(SG_LU1);
        // -- x+ = Q t x- + q : solve (F - Rt G) [Q | q] = [Rt F - G | c]
        if (MODE == 3) {  // the solution t Q t stays in slot X (one pass over the matrix instead of three)
            if (!gj_solve_b16<NT, false, CH>(Wk, Rt, s.cvec, s, N, LD, true, s.t, s.t, true)) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
        } else
        if (!gj_solve<NT, false, CH>(Wk, Rt, s.cvec, s, N, LD, MODE == 2)) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
        SMRT_STAGE(SG_R45);
        double* Q = (MODE == 3) ? Wk : Rt;
        SMRT_DUMP("Q", Q, N);
        if (MODE != 3) for_2d<NT>(N, N, [&](int r, int c) { Q[c * LD + r] *= s.t[r] * s.t[c]; });
        for (int r = t; r < N; r += NT) s.tq[r] = s.t[r] * s.cvec[r];
        block_sync();
        if (MODE == 3) {
            r45_mfma2<NT>(F, G, Q, Rt, Wk, s.Rtop, s.tq, s.upb, s.g, Bl, N, LD);   // Y -> slot R, W -> slot X (over Q)
        } else if (CH == 1) {
            r45_mfma<NT>(F, G, Q, Wk, s.Rtop, s.tq, s.upb, s.g, Bl, N, LD);
        } else if (CH >= 2 && dense_mfma) {
            if (CH > 2) {   // two passes with one operand array each (register budget, see r45_mfma_big)
                r45_mfma_big<NT, false, 16 * CH, 1>(F, G, Q, Wk, s.Rtop, s.tq, s.upb, s.g, Bl, N, LD, nullptr, big_stage, plan.stage_bufs, 64 * ((plan.NMAX + 3) / 4));
                r45_mfma_big<NT, false, 16 * CH, 2>(F, G, Q, Wk, s.Rtop, s.tq, s.upb, s.g, Bl, N, LD, nullptr, big_stage, plan.stage_bufs, 64 * ((plan.NMAX + 3) / 4));
            } else r45_mfma_big<NT, false, 16 * CH>(F, G, Q, Wk, s.Rtop, s.tq, s.upb, s.g, Bl, N, LD, nullptr);
        } else {
            r45_rows<NT, CH, false>(F, G, Q, Wk, s.Rtop, s.tq, s.upb, s.g, Bl, N, LD, nullptr);
        }
        SMRT_DUMP("Y", Wk, N); SMRT_DUMP("W", F, N);
        SMRT_STAGE(SG_LU2);
        // -- K = Y W^-1  (solve W^T K^T = Y^T on the transposed view; K lands in Wk in normal storage)
        if (MODE == 3) {  // A = W (slot X), B = Y (slot R); K is left in slot X
            if (!gj_solve_b16<NT, true, CH>(Wk, Rt, nullptr, s, N, LD, true, nullptr, nullptr, true)) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
        } else
        if (!gj_solve<NT, true, CH>(F, Wk, nullptr, s, N, LD, MODE == 2)) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
        SMRT_STAGE(SG_R78);
        double* K = Wk;
        SMRT_DUMP("K", K, N);
        // -- upwelling intensity just below the top interface of layer l: up = F tq + B - K g
        for (int i = t; i < N; i += NT) {
            double acc = s.upb[i];
            for (int k = 0; k < N; ++k) acc -= K[k * LD + i] * s.g[k];
            s.up[i] = acc;
        }
        block_sync();
        if (l > 0 || hs >= 0) {
            // reflection matrix and source seen from the bottom of layer l-1 (streams paired by index); under a rough
            // interface first as if the layer above had the same streams and no interface ...
            const int Nue = (hs >= 0) ? N : Nu;
            const int nc = (N < Nue) ? N : Nue;
            for_2d<NT>(Nue, Nue, [&](int i, int j) {
                double v = (i == j) ? s.Rbu[i] : 0.0;
                if (i < nc && j < nc) v += s.Ttop[i] * K[j * LD + i] * s.Tbu[j];
                s.M3[j * LD + i] = v;
            });
            for (int i = t; i < Nue; i += NT) s.svec[i] = (i < nc) ? s.Ttop[i] * s.up[i] : 0.0;
            block_sync();
            if (hs >= 0) {   // ... then composed with the caller's matrices: R~ and s as the medium above (layer l - 1 or the air) sees them
                const int Nabove = (l > 0) ? Nu : n_air * P;
                if (!interface_dense_step<NT>(host_interface_matrices(b, gp, hs, 0, 1), 3 * nmax, K, s.M3, s.up, s.svec, s.gj,
                                              N, Nabove, LD, true)) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
            }
        }
        rough_surface = (l == 0 && hs >= 0);
    }

    if (MODE == 1) {
        if (t == 0) b.status[p] = ST_OK;
#ifdef SMRT_STAGE_TIMING
        SMRT_STAGE(SG_OUT);   // the prep kernel's own stages: slots 0..3 (the finish kernel fills the others)
        if (t == 0 && b.stage_out) for (int k = 0; k < 4; ++k) b.stage_out[p * 16 + k] = stage_acc[k];
#endif
        return;
    }
    SMRT_STAGE(SG_OUT);
    // ---- emerging brightness temperature at the air streams, then at the sensor angles ---------------------
    {
        // atmosphere (rtsolver_utils.py:251-260,302-305): isotropic downwelling radiation I_dn enters through the
        // surface (dort.py:391-395), is reflected by it (dort.py:484) and by the snowpack (K_0 of the top layer is
        // still in the work matrix), and the result is tb_up + transmittance * (...)
        const bool atm = (b.atm_down != nullptr);
        const double Idn = atm ? (b.rayleigh_jeans ? b.atm_down[fi] : planck_radiance(frequency, b.atm_down[fi])) : 0.0;
        const double Iup = atm ? (b.rayleigh_jeans ? b.atm_up[fi] : planck_radiance(frequency, b.atm_up[fi])) : 0.0;
        const double trans = atm ? b.atm_trans[fi] : 1.0;
        const double* K0 = s.M0;
        const cplx e0 = cmk(s.eps_re[0], s.eps_im[0]);
        for (int i = t; i < n_air * P; i += NT) {
            double I0 = s.Ttop[i] * s.up[i];  // dort.py:484
            if (rough_surface) {   // I0 = R~_air I_sky + s with the composed matrices (dense R_air, T_top, T_air inside)
                double acc = 0.0;
                for (int j = 0; j < n_air * P; ++j) acc += s.M3[j * LD + i];
                I0 = acc * Idn + s.svec[i];
            } else
            if (atm && Idn != 0.0) {
                double acc = 0.0;
                const cplx slab0 = cmk(s.slab_re[0], s.slab_im[0]);
                double Rv, Rh, Tv, Th;
                for (int j = 0; j < n_air; ++j) {
                    interface_RT(frequency, cmk(1.0, 0.0), e0, s.outmu[j], slab0, s.slab_th[0], &Rv, &Rh, &Tv, &Th);
                    acc += K0[(2 * j) * LD + i] * Tv + K0[(2 * j + 1) * LD + i] * Th;
                }
                interface_RT(frequency, cmk(1.0, 0.0), e0, s.outmu[i >> 1], slab0, s.slab_th[0], &Rv, &Rh, &Tv, &Th);
                I0 += ((i & 1) ? Rh : Rv) * Idn + s.Ttop[i] * acc * Idn;
            }
            if (atm) I0 = Iup + trans * I0;
            s.tb[i] = b.rayleigh_jeans ? I0 : planck_inverse(frequency, I0);
        }
    }
    block_sync();
    for (int idx = t; idx < P * b.n_theta; idx += NT) {
        const int pol = idx / b.n_theta, it = idx % b.n_theta;
        const double um = cos(b.theta[it]);
        // outmu is descending; a virtual node mu = 1 holding mean(V,H) of the steepest stream is prepended when the
        // request is steeper than every stream (rtsolver_utils.py:191-198); linear inter/extrapolation otherwise
        double x0, x1, y0, y1;
        const double top = 0.5 * (s.tb[0] + s.tb[1]);
        if (um > s.outmu[0]) { x0 = 1.0; y0 = top; x1 = s.outmu[0]; y1 = s.tb[pol]; }
        else if (n_air == 1) { x0 = 1.0; y0 = top; x1 = s.outmu[0]; y1 = s.tb[pol]; }
        else {
            int k = 0;  // segment [outmu[k+1], outmu[k]] containing um, clamped for extrapolation
            while (k < n_air - 2 && um < s.outmu[k + 1]) ++k;
            x0 = s.outmu[k]; y0 = s.tb[2 * k + pol]; x1 = s.outmu[k + 1]; y1 = s.tb[2 * (k + 1) + pol];
        }
        b.out[p * out_stride + idx] = y0 + (y1 - y0) * ((um - x0) / (x1 - x0));
    }
    if (t == 0) { b.status[p] = ST_OK; if (b.n3_out) b.n3_out[p] = n3; }
#ifdef SMRT_STAGE_TIMING
    SMRT_STAGE(SG_OUT);
    if (t == 0 && b.stage_out) {
        for (int k = (MODE >= 2 ? 4 : 0); k < 16; ++k) b.stage_out[p * 16 + k] = (k < SG_COUNT) ? stage_acc[k] : 0.0;   // (0..3: the prep kernel's)
        b.stage_out[p * 16 + 12] = (double)n_sweeps;
        for (int k = 0; k < 3; ++k) b.stage_out[p * 16 + 13 + k] = sub_acc_store[k];
    }
#endif
}

}  // namespace smrt
