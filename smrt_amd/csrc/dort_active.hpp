// DORT hot path, ACTIVE mode (backscatter): device code (gfx950 / CDNA4), one workgroup per (snowpack, frequency).
//
// What is computed (paths relative to /root/reference):
//   azimuth modes m = 0..m_max        smrt/rtsolver/dort.py:209-261 (mode loop, coherent subtraction, cos/sin(m phi))
//   incident streams / intensity_0    smrt/rtsolver/dort.py:210-239 (two bracketing streams per incidence angle,
//                                     power 1/(2 pi outweight), x2 for m > 0)
//   phase matrix modes, 3 pols        smrt/emmodel/common.py:9-131 (IBA: discrete Fourier sums over 2^k azimuths),
//                                     smrt/emmodel/rayleigh.py:52-127 (DMRT: closed forms m = 0, 1, 2)
//   third Stokes component            smrt/rtsolver/dort.py:716-737 (U extinction / normalisation), :925-953 (half-rank
//                                     reduction with the sign flip of the U columns of beta and the U rows of Ed),
//                                     smrt/core/fresnel.py:417-474 (R_U, T_U)
//   interpolation                     smrt/rtsolver/rtsolver_utils.py:199-239 (active branch)
//
// Design.  For m >= 1 the reduced matrices alpha -/+ beta D are NOT symmetrisable by the mode-0 scaling alone: the
// (V|H, U) cross blocks of the phase matrix obey P_(U,x)(mu_i, mu_j) = 2 P_(x,U)(mu_j, mu_i).  Scaling the U
// component by sqrt(2) on top of sqrt(norm w / mu) removes the factor: X+ and X- are then exactly symmetric and
// (for albedo < 1) positive definite, so every mode runs through the same Cholesky x 2 -> one-sided Jacobi ->
// triangular recovery pipeline as the passive solve -- no general non-symmetric eigensolver is needed.
// There are no thermal sources in active mode: the layer recursion carries only the reflection matrix, and the
// answer is read off the top-level reflection matrix K_0 at the incident streams (backscatter: scattered stream ==
// incident stream).  The coherent (no-scattering) solution is diagonal and is a per-stream scalar recursion.
#pragma once
#include "dort_device.hpp"

namespace smrt {

// Flat interface, Maezawa & Miyauchi 2009 Fresnel (core/fresnel.py:99-146): power R and T for V, H and the
// coherency term U (fresnel.py:417-474).
// With slab_thickness > 0: the coherent interface of process_coherent_layers (coherent_flat.py:76-147, see coherent_slab).
SMRT_DEV void fresnel_RT3(cplx e1, cplx e2, double mu1, double* R3, double* T3, double frequency = 0.0,
                          cplx es = cplx{0.0, 0.0}, double slab_thickness = 0.0) {
    if (slab_thickness > 0.0) {
        const SlabRT q = coherent_slab(frequency, e1, e2, mu1, es, slab_thickness);
        const double nt = csqrt_(cdiv(e2, e1)).re;
        R3[0] = cabs2(q.Rv); R3[1] = cabs2(q.Rh);
        R3[2] = q.Rv.re * q.Rh.re + q.Rv.im * q.Rh.im;
        T3[0] = cabs2(q.Tv) * q.mu_t / mu1 / nt;
        T3[1] = cabs2(q.Th) * q.mu_t / mu1 * nt;
        T3[2] = q.mu_t / mu1 * ((1.0 + q.Rv.re) * (1.0 + q.Rh.re) + q.Rv.im * q.Rh.im);
        return;
    }
    cplx n1 = csqrt_(e1);
    double kz2 = n1.re * n1.re * (1.0 - mu1 * mu1);
    cplx kyi = cscale(csqrt_(cmk(e1.re - kz2, e1.im)), -1.0);
    cplx kyt = cscale(csqrt_(cmk(e2.re - kz2, e2.im)), -1.0);
    cplx rh = cdiv(csub(kyi, kyt), cadd(cconj(kyi), kyt));
    cplx num = cmul(cconj(n1), csub(cmul(e2, kyi), cmul(e1, kyt)));
    cplx den = cmul(n1, cadd(cmul(e2, cconj(kyi)), cmul(cconj(e1), kyt)));
    cplx rv = cdiv(num, den);
    R3[0] = cabs2(rv);
    R3[1] = cabs2(rh);
    R3[2] = rv.re * rh.re + rv.im * rh.im;
    T3[0] = 1.0 - R3[0];
    T3[1] = 1.0 - R3[1];
    const double mu2 = -kyt.re / csqrt_(e2).re;
    T3[2] = mu2 / mu1 * ((1.0 + rv.re) * (1.0 + rh.re) + rv.im * rh.im);
}

// MODE 0: everything in one workgroup.  MODE 1 / 3: the "prep" and "finish" halves of the three-kernel pipeline (see
// dort_pair_passive): staging items are (pair, mode, layer) -> item = (p (m_max + 1) + m) Lmax + l; the Jacobi kernel
// in between is the passive one.
template <int NT, int CH, int MODE = 0>
SMRT_DEV void dort_pair_active(const DevBatch& b, long long p, double* lds_base, double* gmem_mat = nullptr,
                               const DevStage* stg = nullptr) {
    constexpr int JW = (NT / SMRT_LANES >= 4) ? 4 : NT / SMRT_LANES;
    constexpr int GS = 8;
    constexpr int RPL = (64 * CH + GS - 1) / GS;
    const int t = tid();
    const int m_max = b.m_max;
    const int nsamp = azimuth_samples(m_max);
    const int nphi = nsamp / 2 + 1;
    const int nmax = b.n_max_stream;
    const LdsPlan plan = make_plan(nmax, 3, b.Lmax, b.n_theta, nphi, gmem_mat == nullptr ? 1 : 0,
                                   active_doubles(nmax, b.Lmax, b.n_theta, MODE < 2), MODE == 1 ? 1 : (MODE == 3 ? 2 : 0),
                                   (gmem_mat != nullptr && MODE != 1) ? ((CH > 2 && MODE == 2) ? 3 : (MODE == 2 && b.jac_in_lds ? 2 : b.jac_in_lds)) : 0);
    Lds s = carve(lds_base, gmem_mat == nullptr ? lds_base : gmem_mat, plan);
    // matrix-core variants of the dense steps: always on the LDS path; on the global-workspace path for N <= 128 when
    // the LDS Jacobi buffer exists (it doubles as the scratch of the blocked Cholesky / triangular solve)
    // N > 128 (CH > 2): always, with the scratch of the blocked solvers behind the work matrices in the global workspace
    const bool dense_mfma = (CH == 1) || (CH == 2 && (plan.o_jac >= 0 || MODE == 1)) || (CH > 2);
    double* dense_scratch = (CH > 2) ? gmem_mat + plan.mat_doubles
                                     : ((CH == 1 || MODE == 1) ? s.gj : lds_base + (plan.o_jac >= 0 ? plan.o_jac : 0));
#ifdef SMRT_STAGE_TIMING
    double sub_acc_store[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    s.sub_acc = sub_acc_store;
    double stage_acc[SG_COUNT];
    for (int k = 0; k < SG_COUNT; ++k) stage_acc[k] = 0.0;
    long long stage_t0 = cycle_counter();
    int stage_cur = SG_SETUP;
#endif
    // N > 128 finish kernel: LDS staging buffers of the matrix-core products (make_plan jac_in_lds = 3)
    double* big_stage = (CH > 2 && MODE == 2 && plan.o_jac >= 0) ? lds_base + plan.o_jac : nullptr;
    const int LD = plan.LD;
    const int out_stride = 9 * b.n_theta;
    const int NI = 2 * b.n_theta;                 // capacity of the incident stream list
    double* total = s.act;                        // [9][NI]
    double* coh = total + 9 * NI;                 // [2][NI]
    int* inc = (int*)(coh + 2 * NI);              // [NI], then the count
    double* norm0 = coh + 2 * NI + (NI + 2) / 2 + 1;   // [Lmax][2 nmax]; absent in the finish kernels (MODE >= 2)
    double* dsg = s.g;                            // row signs of the down-going eigenvectors (not in the prep kernel)
    auto su_of = [](int r, int P) { return (P == 3 && r % 3 == 2) ? 1.4142135623730951 : 1.0; };  // sqrt(2) on U rows

    const long long gp = global_pair(b, p);
    const int fi = (int)(gp / b.S), si = (int)(gp % b.S);
    const double frequency = b.frequency[fi];
    int L = b.n_layers[si];
    const double* thickness = b.thickness + (long long)si * b.Lmax;
    const double* fracvol = b.frac_volume + (long long)si * b.Lmax;
    const double* temperature = b.temperature + (long long)si * b.Lmax;
    const double* mp1 = b.p1 + (long long)si * b.Lmax;
    const double* mp2 = b.p2 + (long long)si * b.Lmax;

    if (t < 8) s.ints[t] = 0;
    block_sync();
    if (MODE >= 2) {  // a failure recorded by the prep or Jacobi kernel
        const int prev = b.status[p];
        if (prev != ST_OK) { fail_pair<NT>(b, p, prev, out_stride); return; }
    }
    for (int k = t; k < nphi && MODE < 2; k += NT) {
        const double ph = kPi * (double)k / (double)(nphi - 1);
        s.cphi[k] = cos(ph); s.sphi[k] = sin(ph);
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

    // ---- incident streams: the two streams bracketing every incidence angle (dort.py:210-226), sorted, unique ----
    if (t == 0 && MODE != 1) {
        int cnt = 0;
        for (int it = 0; it < b.n_theta; ++it) {
            const double mi = cos(b.theta[it]);
            int i0 = 0;
            while (i0 < n_air && s.outmu[i0] > mi) ++i0;  // np.searchsorted(-outmu, -mu_inc)
            int cand[2], nc = 0;
            if (i0 == 0) cand[nc++] = 0;
            else if (i0 == n_air) cand[nc++] = n_air - 1;
            else { cand[nc++] = i0 - 1; cand[nc++] = i0; }
            for (int q = 0; q < nc; ++q) {
                int pos = 0;
                while (pos < cnt && inc[pos] < cand[q]) ++pos;
                if (pos < cnt && inc[pos] == cand[q]) continue;
                for (int z = cnt; z > pos; --z) inc[z] = inc[z - 1];
                inc[pos] = cand[q];
                ++cnt;
            }
        }
        inc[NI] = cnt;
    }
    for (int k = t; k < 9 * NI && MODE != 1; k += NT) total[k] = 0.0;
    block_sync();
    const int ninc = (MODE != 1) ? inc[NI] : 0;

    // ---- coherent (no scattering) mode-0 solution: diagonal, one thread per (incident stream, V|H) -------------
    for (int idx = t; idx < 2 * ninc; idx += NT) {
        const int jn = idx >> 1, pol = idx & 1;
        const int j = inc[jn];
        double Rt = 0.0, K = 0.0, Ttop0 = 0.0;
        // prune_deep_snowpack (dort.py:443-452) acts on this solve with ITS eigenvalues beta = ke / mu: the smallest
        // one belongs to the steepest stream
        int Lc = L;
        if (MODE >= 2 && b.prune_tau > 0.0) {
            double acc = 0.0;
            for (int l = 0; l < L; ++l) {
                const double rs0 = s.ri[l] * s.gsin[0];
                acc += (s.ks[l] + s.ka[l]) / sqrt(1.0 - rs0 * rs0) * s.thick[l];
                if (acc > b.prune_tau) { Lc = l + 1; break; }
            }
        }
        if (Lc < L) {  // reflection of the interface to the first dropped layer, nothing from below
            if (j < (int)s.nl[Lc - 1]) {
                const double rs = s.ri[Lc - 1] * s.gsin[j];
                double R3[3], T3[3];
                fresnel_RT3(cmk(s.eps_re[Lc - 1], s.eps_im[Lc - 1]), cmk(s.eps_re[Lc], s.eps_im[Lc]), sqrt(1.0 - rs * rs), R3, T3,
                            frequency, cmk(s.slab_re[Lc], s.slab_im[Lc]), s.slab_th[Lc]);
                Rt = R3[pol];
                if (__builtin_expect(b.host_itf_slot != nullptr, 0)) {   // a rough interface there: its specular part
                    const int hsb = host_interface_slot(b, gp, (int)s.lo[Lc]);
                    if (hsb >= 0) Rt = host_interface_specular(b, gp, hsb)[2 * 3 * nmax + 2 * j + pol];
                }
            }
        } else
        if (b.sub_kind == SUB_HOST && j < (int)s.nl[L - 1]) {   // specular part of a rough substrate, from the caller (mode 0)
            Rt = b.host_substrate_coh[(gp * (long long)(m_max + 1)) * (3 * nmax) + 2 * j + pol];
        } else
        if (b.sub_kind == SUB_FLAT && j < (int)s.nl[L - 1]) {  // specular reflection of the substrate
            const double rs = s.ri[L - 1] * s.gsin[j];
            double R3[3], T3[3];
            fresnel_RT3(cmk(s.eps_re[L - 1], s.eps_im[L - 1]), cmk(b.sub_p1[gp], b.sub_p2[gp]), sqrt(1.0 - rs * rs), R3, T3);
            Rt = R3[pol];
        }
        for (int l = Lc - 1; l >= 0; --l) {
            const int n = (int)s.nl[l];
            const cplx el = cmk(s.eps_re[l], s.eps_im[l]);
            const cplx eup = (l > 0) ? cmk(s.eps_re[l - 1], s.eps_im[l - 1]) : cmk(1.0, 0.0);
            K = 0.0;
            double Tt = 0.0;
            // a rough interface on top of this layer: its specular parts come from the caller (smrt_dort.h)
            const int hsc = host_interface_slot(b, gp, (int)s.lo[l]);
            const double* hspec = (hsc >= 0) ? host_interface_specular(b, gp, hsc) : nullptr;
            if (j < n) {
                const double rs = s.ri[l] * s.gsin[j];
                const double mu = sqrt(1.0 - rs * rs);
                double R3[3], T3[3];
                fresnel_RT3(el, eup, mu, R3, T3, frequency, cmk(s.slab_re[l], s.slab_im[l]), s.slab_th[l]);
                if (hspec) { R3[pol] = hspec[2 * j + pol]; T3[pol] = hspec[3 * nmax + 2 * j + pol]; }
                const double tt = exp(-(s.ks[l] + s.ka[l]) * s.thick[l] / mu);
                const double y = tt * tt * Rt;
                K = y / (1.0 - R3[pol] * y);
                Tt = T3[pol];
            }
            if (l > 0) {
                const int nu = (int)s.nl[l - 1];
                if (j < nu) {
                    const double rs = s.ri[l - 1] * s.gsin[j];
                    const double muu = sqrt(1.0 - rs * rs);
                    double R3[3], T3[3];
                    fresnel_RT3(eup, el, muu, R3, T3, frequency, cmk(s.slab_re[l], s.slab_im[l]), s.slab_th[l]);
                    if (hspec) { R3[pol] = hspec[2 * 3 * nmax + 2 * j + pol]; T3[pol] = hspec[3 * 3 * nmax + 2 * j + pol]; }
                    Rt = R3[pol] + Tt * K * T3[pol];
                } else Rt = 0.0;
            } else Ttop0 = Tt;
        }
        double Ra[3], Ta[3];
        fresnel_RT3(cmk(1.0, 0.0), cmk(s.eps_re[0], s.eps_im[0]), s.outmu[j], Ra, Ta, frequency, cmk(s.slab_re[0], s.slab_im[0]), s.slab_th[0]);
        {
            const int hs0 = host_interface_slot(b, gp, (int)s.lo[0]);
            if (hs0 >= 0) {
                const double* hspec = host_interface_specular(b, gp, hs0);
                Ra[pol] = hspec[2 * 3 * nmax + 2 * j + pol]; Ta[pol] = hspec[3 * 3 * nmax + 2 * j + pol];
            }
        }
        coh[pol * NI + jn] = Ra[pol] + Ttop0 * K * Ta[pol];
    }
    block_sync();

    double n3 = 0.0;
    int n_sweeps = 0;
    if (MODE == 1 && b.pair_done && b.pair_done[p]) return;   // every azimuth mode of this pair is cut above this round's layers
    for (int m = 0; m <= m_max; ++m) {
        const int P = (m == 0) ? 2 : 3;
        const double cc = (m == 0) ? 0.5 : 0.25;  // dort.py:716-721
        // azimuth weights of this mode: cosine sums for the even entries, sine sums for the (V|H, U) cross entries
        for (int k = t; k < nphi && MODE < 2; k += NT) {
            const double ph = kPi * (double)k / (double)(nphi - 1);
            const bool end = (k == 0 || k == nphi - 1);
            const double base = ((m == 0) ? 1.0 : 2.0) / (double)nsamp;
            s.wphi[k] = base * (end ? 1.0 : 2.0) * cos((double)m * ph);
            s.swphi[k] = end ? 0.0 : base * 2.0 * sin((double)m * ph);
        }
        block_sync();
        int Lk = L;  // layers kept by prune_deep_snowpack for this azimuth mode (s.pa is free in the finish kernels)
        if (MODE >= 2 && b.prune_tau > 0.0)
            Lk = pruned_layer_count<NT>(*stg, (p * (long long)(m_max + 1) + m) * b.Lmax, L, s.thick, s.pa, b.prune_tau);
        if (MODE >= 2) {  // failures recorded per layer by the prep / Jacobi kernels: only the kept layers count; the
                          // mode-0 assembly of a layer also provides the normalisation of its higher modes
            int bad = first_failed_layer(*stg, (p * (long long)(m_max + 1) + m) * b.Lmax, Lk);
            if (bad == ST_OK && m > 0) bad = first_failed_layer(*stg, (p * (long long)(m_max + 1)) * b.Lmax, Lk);
            if (bad != ST_OK) { fail_pair<NT>(b, p, bad, out_stride); return; }
        }
        auto layer_failed = [&](int l, int code) {   // prep kernel: record and skip the layer (uniform)
            block_sync();
            if (t == 0) { stg->n[(p * (long long)(m_max + 1) + m) * b.Lmax + l] = -code; s.ints[0] = ST_OK; }
            block_sync();
        };
        for (int l = Lk - 1; l >= 0; --l) {
            if (MODE == 1 && (l < b.layer_lo || l >= b.layer_hi)) continue;   // not in this round of the prep kernel
            const int n = (int)s.nl[l];
            const int N = n * P;
            n3 += (double)N * N * N;
            const cplx el = cmk(s.eps_re[l], s.eps_im[l]);
            const double ks = s.ks[l], ke = s.ks[l] + s.ka[l];
            const int nu = (l > 0) ? (int)s.nl[l - 1] : 0;
            const int Nu = nu * P;
            SMRT_STAGE(SG_SETUP);

            for (int j = t; j < n; j += NT) { const double rs = s.ri[l] * s.gsin[j]; s.mu[j] = sqrt(1.0 - rs * rs); }
            if (l > 0)
                for (int j = t; j < nu; j += NT) { const double rs = s.ri[l - 1] * s.gsin[j]; s.muu[j] = sqrt(1.0 - rs * rs); }
            const long long item = (p * (long long)(m_max + 1) + m) * b.Lmax + l;   // staging slot (MODE 1 / 3)
            if (MODE != 1 && l == Lk - 1) for_2d<NT>(N, N, [&](int r, int c) { s.M3[c * LD + r] = 0.0; });
            if (MODE != 1) for (int r = t; r < N; r += NT) { s.svec[r] = 0.0; s.tq[r] = 0.0; }
            block_sync();
            if (MODE != 1 && l == Lk - 1 && (Lk < L || b.sub_kind == SUB_FLAT)) {
                // substrate: R_sub (V, H, U) on the diagonal; with deeper layers pruned: the interface to the first
                // dropped layer instead (dort.py:446-452)
                const cplx ebelow = (Lk < L) ? cmk(s.eps_re[l + 1], s.eps_im[l + 1]) : cmk(b.sub_p1[gp], b.sub_p2[gp]);
                for (int j = t; j < n; j += NT) {
                    double R3[3], T3[3];
                    if (Lk < L) fresnel_RT3(el, ebelow, s.mu[j], R3, T3, frequency, cmk(s.slab_re[l + 1], s.slab_im[l + 1]), s.slab_th[l + 1]);
                    else fresnel_RT3(el, ebelow, s.mu[j], R3, T3);
                    for (int q = 0; q < P; ++q) s.M3[(P * j + q) * LD + P * j + q] = R3[q];
                }
            }
            if (MODE != 1 && l == Lk - 1) {
                // a DENSE start of the recursion, from the caller (smrt_dort.h): the reflection matrix of this mode of a rough
                // substrate, or -- deeper layers pruned right above a rough interface -- that interface's reflection seen
                // from this layer (Rbot of its slot), what the reference's truncated system keeps (dort.py:443-452,
                // rtsolver_utils.py:567-597).  One copy loop for both (the pointer is uniform over the workgroup).
                const long long NE = 3LL * nmax;
                const double* H = nullptr;
                if (Lk == L) { if (b.sub_kind == SUB_HOST) H = b.host_substrate + (gp * (long long)(m_max + 1) + m) * NE * NE; }
                else if (__builtin_expect(b.host_itf_slot != nullptr, 0)) {
                    const int hsb = b.host_itf_slot[gp * b.Lmax + (int)s.lo[l + 1]];
                    if (hsb >= 0) H = b.host_itf + (((gp * b.host_itf_slots + hsb) * (long long)(m_max + 1) + m) * 4 + 2) * NE * NE;
                }
                if (__builtin_expect(H != nullptr, 0)) {
                    block_sync();
                    for_2d<NT>(N, N, [&](int r, int c) { s.M3[c * LD + r] = H[r * NE + c]; });
                }
            }
            for (int j = t; j < n; j += NT) {
                double w;
                if (j == 0) w = 1.0 - 0.5 * (s.mu[0] + s.mu[1]);
                else if (j == n - 1) w = fabs(0.5 * (s.mu[n - 2] + s.mu[n - 1]));
                else w = fabs(0.5 * (s.mu[j - 1] - s.mu[j + 1]));
                s.w[j] = w;
                double R3[3], T3[3];
                const cplx eup = (l > 0) ? cmk(s.eps_re[l - 1], s.eps_im[l - 1]) : cmk(1.0, 0.0);
                fresnel_RT3(el, eup, s.mu[j], R3, T3, frequency, cmk(s.slab_re[l], s.slab_im[l]), s.slab_th[l]);
                for (int q = 0; q < P; ++q) {
                    const int r = P * j + q;
                    if (MODE < 2) { s.mrow[r] = s.mu[j]; s.wrow[r] = w; }
                    if (MODE != 1) {
                        s.Rtop[r] = R3[q]; s.Ttop[r] = T3[q];
                        dsg[r] = (q == 2) ? -1.0 : 1.0;
                    }
                }
            }
            if (MODE != 1 && l > 0)
                for (int j = t; j < nu; j += NT) {
                    double R3[3], T3[3];
                    fresnel_RT3(cmk(s.eps_re[l - 1], s.eps_im[l - 1]), el, s.muu[j], R3, T3, frequency, cmk(s.slab_re[l], s.slab_im[l]), s.slab_th[l]);
                    for (int q = 0; q < P; ++q) { s.Rbu[P * j + q] = R3[q]; s.Tbu[P * j + q] = T3[q]; }
                }

            // a rough interface on top of this layer (evaluated by the caller): transparent top for the layer step, the
            // interface is composed afterwards (dort_interface_dense.hpp)
            const int hs = (MODE != 1) ? host_interface_slot(b, gp, (int)s.lo[l]) : -1;
            if (hs >= 0) {
                block_sync();
                for (int r = t; r < N; r += NT) { s.Rtop[r] = 0.0; s.Ttop[r] = 1.0; s.Rbu[r] = 0.0; s.Tbu[r] = 1.0; }
            }
            if (MODE < 2) {
            // -- phase matrix of mode m: S+ = P(mu,+mu') + P(mu,-mu') D -> M0, S- = P(+) - P(-) D -> M1, lower
            //    triangle (rows = scattered stream/polarisation, columns = incident); D = -1 on the U columns
            {
                const int T = n * (n + 1) / 2;
                const double pa = s.pa[l], pb = s.pb[l];
                const int lo = (int)s.lo[l];   // the layer's index in the input arrays
                const double fv = fracvol[lo], q1 = mp1[lo], q2 = mp2[lo];
                const int em_l = (int)s.pc[l] & 15, ms_l = (int)s.pc[l] >> 4;   // this layer's emmodel and microstructure
                for (int idx = t; idx < T; idx += NT) {
                    int i = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
                    while ((i + 1) * (i + 2) / 2 <= idx) ++i;
                    while (i * (i + 1) / 2 > idx) --i;
                    const int j = idx - i * (i + 1) / 2;
                    const double mi = s.mu[i], mj = s.mu[j];
                    const double sis = sqrt(1.0 - mi * mi), sjs = sqrt(1.0 - mj * mj);
                    double pp[3][3], pm[3][3];
                    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) { pp[a][c] = 0.0; pm[a][c] = 0.0; }
                    if (em_l == EM_HOST) {  // mode m of the caller's ft_even_phase(mu, +-mu'), compressed (smrt_dort.h)
                        const int NE = b.host_ne;
                        const double* hp = b.host_phase + ((gp * b.Lmax + lo) * (long long)b.host_modes + m) * 2 * NE * NE;
                        const double* hm = hp + (long long)NE * NE;
                        for (int a = 0; a < P; ++a)
                            for (int c = 0; c < P; ++c) {
                                pp[a][c] = hp[(3 * i + a) * NE + 3 * j + c];   // the caller compresses every mode
                                pm[a][c] = hm[(3 * i + a) * NE + 3 * j + c];   // with three polarisations
                            }
                    } else if (em_l != EM_IBA) {  // rayleigh.py:70-127 (with the sign convention of :121-124)
                        for (int sgn = 0; sgn < 2; ++sgn) {
                            const double x = sgn ? -mj : mj;
                            double (&q)[3][3] = sgn ? pm : pp;
                            const double a2 = mi * mi, x2 = x * x;
                            if (m == 0) {
                                q[0][0] = pa * (0.5 * a2 * x2 + (1.0 - a2) * (1.0 - x2));
                                q[0][1] = pa * 0.5 * a2; q[1][0] = pa * 0.5 * x2; q[1][1] = pa * 0.5;
                            } else if (m == 1) {
                                const double cs = mi * sis, ci = x * sjs;
                                q[0][0] = pa * 2.0 * cs * ci;
                                q[0][2] = -pa * cs * sjs;
                                q[2][0] = -pa * 2.0 * sis * ci;
                                q[2][2] = pa * sis * sjs;
                            } else if (m == 2) {
                                q[0][0] = pa * 0.5 * a2 * x2; q[0][1] = -pa * 0.5 * a2;
                                q[1][0] = -pa * 0.5 * x2; q[1][1] = pa * 0.5;
                                q[0][2] = -pa * 0.5 * a2 * x; q[1][2] = pa * 0.5 * x;
                                q[2][0] = -pa * mi * x2; q[2][1] = pa * mi; q[2][2] = pa * mi * x;
                            }
                        }
                    } else {
                        const double sisj = sis * sjs, mm = mi * mj;
                        for (int k = 0; k < nphi; ++k) {
                            const double c = s.cphi[k], sn = s.sphi[k], cw = s.wphi[k], sw = s.swphi[k];
                            for (int sgn = 0; sgn < 2; ++sgn) {
                                const double x = sgn ? -mj : mj;
                                double (&q)[3][3] = sgn ? pm : pp;
                                double ct = (sgn ? -mm : mm) + sisj * c;  // cosine of the scattering angle
                                ct = ct > 1.0 ? 1.0 : (ct < -1.0 ? -1.0 : ct);
                                double C;
                                if (ms_l == MS_EXP) { const double dp = 1.0 + pb * (1.0 - ct); C = pa / (dp * dp); }
                                else C = pa * ft_corr(ms_l, pb * (1.0 - ct), fv, q1, q2);
                                const double fvv = c * mi * x + sisj, fvh = sn * mi, fhv = -sn * x, fhh = c;
                                const double Cc = C * cw, Cs = C * sw;
                                q[0][0] += fvv * fvv * Cc; q[0][1] += fvh * fvh * Cc;
                                q[1][0] += fhv * fhv * Cc; q[1][1] += fhh * fhh * Cc;
                                if (P == 3) {
                                    q[2][2] += (fvv * fhh + fvh * fhv) * Cc;
                                    q[0][2] -= fvh * fvv * Cs; q[1][2] -= fhh * fhv * Cs;
                                    q[2][0] += 2.0 * fvv * fhv * Cs; q[2][1] += 2.0 * fvh * fhh * Cs;
                                }
                            }
                        }
                    }
                    for (int a = 0; a < P; ++a)
                        for (int c = 0; c < P; ++c) {
                            if (i == j && a < c) continue;  // above the diagonal
                            const double dm = (c == 2) ? -pm[a][c] : pm[a][c];
                            s.M0[(P * j + c) * LD + P * i + a] = pp[a][c] + dm;
                            s.M1[(P * j + c) * LD + P * i + a] = pp[a][c] - dm;
                        }
                }
            }
            block_sync();
            // -- renormalisation (dort.py:782-819): mode 0 fixes norm_r = ks / (c sum_c S+[r,c] w_c); the higher modes
            //    reuse it, with sqrt(norm_V norm_H) on the U rows (dort.py:728-737)
            for (int r = t; r < N; r += NT) {
                double nr = 1.0;
                if (m == 0) {
                    double rs = 0.0;
                    for (int c = 0; c <= r; ++c) rs += s.M0[c * LD + r] * s.wrow[c];
                    for (int c = r + 1; c < N; ++c) rs += s.M0[r * LD + c] * s.wrow[c];
                    if (b.normalization != 0 && ks != 0.0) {
                        nr = ks / (0.5 * rs);
                        if (b.normalization == 1 && !(fabs(nr - 1.0) <= 0.3)) lds_max(&s.ints[0], ST_NORM);
                    }
                    norm0[l * 2 * nmax + r] = nr;
                } else {
                    const int j = r / 3, q = r - 3 * j;
                    const double nv = norm0[l * 2 * nmax + 2 * j], nh = norm0[l * 2 * nmax + 2 * j + 1];
                    nr = (q == 0) ? nv : ((q == 1) ? nh : sqrt(nv * nh));
                }
                const double uu = sqrt(nr * s.wrow[r] / s.mrow[r]);
                s.u[r] = uu;
                s.d[r] = su_of(r, P) * uu / s.wrow[r];
            }
            block_sync();
            if (s.ints[0] != ST_OK) {
                if (MODE == 1) { layer_failed(l, s.ints[0]); continue; }
                fail_pair<NT>(b, p, s.ints[0], out_stride); return;
            }
            // -- X+- (lower triangles), symmetric positive definite thanks to the sqrt(2) scaling of U
            for_2d<NT>(N, N, [&](int r, int c) {
                if (r >= c) {
                    const double uu = cc * (s.u[r] / su_of(r, P)) * (s.u[c] * su_of(c, P));
                    const double dg = (r == c) ? ke / s.mrow[r] : 0.0;
                    s.M0[c * LD + r] = dg - uu * s.M0[c * LD + r];
                    s.M1[c * LD + r] = dg - uu * s.M1[c * LD + r];
                }
            });
            block_sync();
            if (!(dense_mfma ? chol2_mfma<NT>(s.M0, s.M1, dense_scratch, &s.ints[2], N, LD, (MODE == 1 && CH == 1) ? stg->Linv + item * stg->linv_stride : nullptr)
                          : chol2<NT>(s.M0, s.M1, N, LD))) {
                if (MODE == 1) { layer_failed(l, ST_ALBEDO); continue; }
                fail_pair<NT>(b, p, ST_ALBEDO, out_stride); return;
            }
            if (MODE == 1) {  // B = L+^T L- (columns reversed), L+ and d to the staging area; the Jacobi kernel is next
                lt_times_l_mfma<NT>(s.M0, s.M1, stg->B + item * stg->mat_stride, N, LD, true);
                double* gL = stg->L + item * stg->mat_stride;
                for_2d<NT>(N, N, [&](int r, int c) { gL[c * LD + r] = s.M0[c * LD + r]; });
                for (int r = t; r < N; r += NT) stg->d[item * stg->vec_stride + r] = s.d[r];
                if (t == 0) stg->n[item] = N;
                block_sync();
                continue;
            }
            if (dense_mfma) lt_times_l_mfma<NT>(s.M0, s.M1, s.M2, N, LD);     // B = L+^T L-
            else lt_times_l<NT>(s.M0, s.M1, s.M2, N, LD);
            {
                double* Jm = (plan.o_jac >= 0) ? lds_base + plan.o_jac : s.M2;
                if (Jm != s.M2) { for_2d<NT>(N, N, [&](int r, int c) { Jm[c * LD + r] = s.M2[c * LD + r]; }); block_sync(); }
                if (!jacobi_onesided<NT, JW, GS, RPL>(Jm, N, LD, s.sigma, s.rsig, &s.ints[1], &n_sweeps, nullptr)) {
                    fail_pair<NT>(b, p, ST_EIGEN, out_stride); return;
                }
                if (Jm != s.M2) { for_2d<NT>(N, N, [&](int r, int c) { s.M2[c * LD + r] = Jm[c * LD + r]; }); block_sync(); }
            }
            }  // MODE < 2
            double* F = s.M2; double* G = s.M1; double* Rt = s.M3; double* Wk = s.M0;
            SMRT_STAGE(SG_TRI);
            double r1a[RowTiles<NT>::RPW][16];   // MODE 3: rows of R~ D of this wavefront's row tile
            if (MODE == 2) {  // four-matrix finish (global workspace): pick up L+, B' = B V, d and the singular values
                const double* gL = stg->L + item * stg->mat_stride;
                const double* gB = stg->B + item * stg->mat_stride;
                for_2d<NT>(N, N, [&](int r, int c) { s.M0[c * LD + r] = gL[c * LD + r]; s.M2[c * LD + r] = gB[c * LD + r]; });
                for (int r = t; r < N; r += NT) {
                    s.d[r] = stg->d[item * stg->vec_stride + r];
                    const double sg = stg->sigma[item * stg->vec_stride + r];
                    s.sigma[r] = sg; s.rsig[r] = 1.0 / sg;
                }
                block_sync();
            }
            if (MODE == 3) {  // two LDS slots (X = M0, R = M3), F and G in the item's dead staging slots
                double* gL = stg->L + item * stg->mat_stride;   // L+, later F
                double* gB = stg->B + item * stg->mat_stride;   // B', later Em' and G
                for_2d<NT>(N, N, [&](int r, int c) { s.M0[c * LD + r] = gB[c * LD + r]; });
                for (int r = t; r < N; r += NT) {
                    s.d[r] = stg->d[item * stg->vec_stride + r];
                    const double sg = stg->sigma[item * stg->vec_stride + r];
                    s.sigma[r] = sg; s.rsig[r] = 1.0 / sg;
                }
                block_sync();
                // R~ D into registers, L+ into the freed slot R for the triangular stage, then F / G into slots R / X
                // (see dort_pair_passive, MODE 3)
                r1_load<NT>(s.M3, r1a, s.cvec, s.svec, 0.0, N, LD, dsg);
                for_2d<NT>(N, N, [&](int r, int c) { s.M3[c * LD + r] = gL[c * LD + r]; });
                block_sync();
                l_times_m_mfma<NT>(s.M3, s.M0, gB, N, LD);                                // Em' = L+ B'
                lt_solve_mfma<NT>(s.M3, s.M0, stg->Linv + item * stg->linv_stride, N, LD, true);      // Ep' = L+^-T B'
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
            for_2d<NT>(N, N, [&](int i, int c) {
                const double ep = s.M2[c * LD + i], em = s.M1[c * LD + i] * s.rsig[c];
                const double hd = 0.5 * s.d[i];
                s.M2[c * LD + i] = hd * (ep + em);   // F = (Ep - Em)/2 with Em = -d Em'/sigma
                s.M1[c * LD + i] = hd * (ep - em);   // G = (Ep + Em)/2
                s.M3[c * LD + i] *= dsg[c];          // reflection matrix seen from this layer times D
            });
            }
            for (int c = t; c < N; c += NT) s.t[c] = exp(-s.sigma[c] * s.thick[l]);
            block_sync();
            SMRT_STAGE(SG_R1);
            // -- Q = (F - R~ D G)^-1 (R~ D F - G)
            if (MODE == 3) r1_compute<NT>(s.M3, s.M0, r1a, N, LD);
            else if (CH == 1) r1_mfma<NT>(F, G, Rt, Wk, s.cvec, s.svec, 0.0, N, LD);
            else if (CH > 2 && big_stage) {   // one staged operand matrix per pass (dort_dense.hpp)
                r1_mfma_big<NT, 16 * CH, 1>(F, G, Rt, Wk, s.cvec, s.svec, 0.0, N, LD, big_stage, plan.stage_bufs, 64 * ((plan.NMAX + 3) / 4));
                r1_mfma_big<NT, 16 * CH, 2>(F, G, Rt, Wk, s.cvec, s.svec, 0.0, N, LD, big_stage, plan.stage_bufs, 64 * ((plan.NMAX + 3) / 4));
            } else if (CH >= 2 && dense_mfma) r1_mfma_big<NT, 16 * CH>(F, G, Rt, Wk, s.cvec, s.svec, 0.0, N, LD);
            else r1_rows<NT, CH>(F, G, Rt, Wk, s.cvec, s.svec, 0.0, N, LD);
            double* K = Wk;
            SMRT_STAGE(SG_LU1);
            if (MODE == 3) {
                if (!gj_solve_b16<NT, false, CH>(Wk, Rt, nullptr, s, N, LD, true, s.t, s.t, true)) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
                // -- Y = F tQt + G -> slot R ; W = (D G - Rtop F) tQt + (D F - Rtop G) -> slot X (over tQt) ; K = Y W^-1
                r45_mfma2<NT, true>(F, G, Wk, Rt, Wk, s.Rtop, s.tq, s.up, s.cvec, 0.0, N, LD, dsg);
                if (!gj_solve_b16<NT, true, CH>(Wk, Rt, nullptr, s, N, LD, true, nullptr, nullptr, true)) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
            } else {
            if (!gj_solve<NT, false, CH>(Wk, Rt, nullptr, s, N, LD, MODE == 2)) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
            double* Q = Rt;
            SMRT_STAGE(SG_R45);
            for_2d<NT>(N, N, [&](int r, int c) { Q[c * LD + r] *= s.t[r] * s.t[c]; });
            block_sync();
            // -- Y = F tQt + G ; W = (D G - Rtop F) tQt + (D F - Rtop G) ; K = Y W^-1
            if (CH == 1) r45_mfma<NT, true>(F, G, Q, Wk, s.Rtop, s.tq, s.up, s.cvec, 0.0, N, LD, dsg);
            else if (CH > 2) {   // two passes with one operand array each (register budget, see r45_mfma_big)
                r45_mfma_big<NT, true, 16 * CH, 1>(F, G, Q, Wk, s.Rtop, s.tq, s.up, s.cvec, 0.0, N, LD, dsg, big_stage, plan.stage_bufs, 64 * ((plan.NMAX + 3) / 4));
                r45_mfma_big<NT, true, 16 * CH, 2>(F, G, Q, Wk, s.Rtop, s.tq, s.up, s.cvec, 0.0, N, LD, dsg, big_stage, plan.stage_bufs, 64 * ((plan.NMAX + 3) / 4));
            } else if (CH == 2 && dense_mfma) r45_mfma_big<NT, true, 16 * CH>(F, G, Q, Wk, s.Rtop, s.tq, s.up, s.cvec, 0.0, N, LD, dsg);
            else r45_rows<NT, CH, true>(F, G, Q, Wk, s.Rtop, s.tq, s.up, s.cvec, 0.0, N, LD, dsg);
            SMRT_STAGE(SG_LU2);
            if (!gj_solve<NT, true, CH>(F, Wk, nullptr, s, N, LD, MODE == 2)) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
            }
            SMRT_STAGE(SG_R78);
            if (l > 0 || hs >= 0) {
                const int Nue = (hs >= 0) ? N : Nu;
                const int nc = (N < Nue) ? N : Nue;
                for_2d<NT>(Nue, Nue, [&](int i, int j) {
                    double v = (i == j) ? s.Rbu[i] : 0.0;
                    if (i < nc && j < nc) v += s.Ttop[i] * K[j * LD + i] * s.Tbu[j];
                    s.M3[j * LD + i] = v;
                });
                block_sync();
                if (hs >= 0) {   // composed with the caller's matrices of this azimuth mode: the reflection matrix the medium above sees
                    const int Nabove = (l > 0) ? Nu : n_air * P;
                    if (!interface_dense_step<NT>(host_interface_matrices(b, gp, hs, m, m_max + 1), 3 * nmax, K, s.M3, nullptr, nullptr,
                                                  s.gj, N, Nabove, LD, false)) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
                }
            }
            if (l == 0) {
                // -- read the backscatter of this mode off the reflection matrix of the whole snowpack
                //    (dort.py:228-259): I0up = [Rbot_air + Ttop_0 K_0 Tbot_air] intensity_0 at the incident streams
                const double cm = cos((double)m * b.phi), sm = sin((double)m * b.phi);
                for (int idx = t; idx < ninc * P * P; idx += NT) {
                    const int jn = idx / (P * P), rem = idx - jn * P * P, po = rem / P, pi = rem - po * P;
                    const int j = inc[jn];
                    double Ra[3], Ta[3];
                    fresnel_RT3(cmk(1.0, 0.0), cmk(s.eps_re[0], s.eps_im[0]), s.outmu[j], Ra, Ta, frequency, cmk(s.slab_re[0], s.slab_im[0]), s.slab_th[0]);
                    double w;  // outweight (streams.py:324-330 on outmu)
                    if (n_air == 1) w = 1.0;
                    else if (j == 0) w = 1.0 - 0.5 * (s.outmu[0] + s.outmu[1]);
                    else if (j == n_air - 1) w = fabs(0.5 * (s.outmu[n_air - 2] + s.outmu[n_air - 1]));
                    else w = fabs(0.5 * (s.outmu[j - 1] - s.outmu[j + 1]));
                    const double power = ((m == 0) ? 1.0 : 2.0) / (2.0 * kPi * w);
                    const int r = P * j + po, c = P * j + pi;
                    double v = ((po == pi) ? Ra[po] : 0.0) + s.Ttop[r] * K[c * LD + r] * Ta[pi];
                    if (hs >= 0) v = s.M3[c * LD + r];   // rough surface: R_air + T_top X T_air, composed above
                    v *= power;
                    if (po < 2 && pi < 2 && po == pi) v -= coh[po * NI + jn] * ((m == 0) ? 1.0 : 2.0) / (2.0 * kPi * w);
                    if (m > 0) v *= (po < 2) ? cm : sm;
                    total[(po * 3 + pi) * NI + jn] += v;
                }
                block_sync();
            }
        }
    }

    if (MODE == 1) {
        if (t == 0) b.status[p] = ST_OK;
        return;
    }
    // ---- interpolation to the incidence angles (rtsolver_utils.py:199-239, active branch) -------------------------
    for (int idx = t; idx < 9 * b.n_theta; idx += NT) {
        const int a = idx / b.n_theta, it = idx - a * b.n_theta;
        const int po = a / 3, pi = a - 3 * po;
        const double um = cos(b.theta[it]);
        const double xtop = s.outmu[inc[0]];
        bool any_above = false;
        for (int q = 0; q < b.n_theta; ++q) any_above = any_above || (cos(b.theta[q]) > xtop);
        // virtual node mu = 1: co = mean(VV, HH), cross = mean(HV, VH) of the steepest incident stream
        double ytop;
        if (po < 2 && pi < 2) {
            if (po == pi) ytop = 0.5 * (total[0 * NI] + total[4 * NI]);
            else ytop = 0.5 * (total[3 * NI] + total[1 * NI]);
        } else ytop = total[a * NI];
        const double* y = total + a * NI;
        double res;
        if (um > xtop || (ninc == 1 && any_above)) {
            res = ytop + (y[0] - ytop) * ((um - 1.0) / (xtop - 1.0));
        } else if (ninc == 1) {
            res = y[0];
        } else {
            int k = 0;
            while (k < ninc - 2 && um < s.outmu[inc[k + 1]]) ++k;
            const double x0 = s.outmu[inc[k]], x1 = s.outmu[inc[k + 1]];
            res = y[k] + (y[k + 1] - y[k]) * ((um - x0) / (x1 - x0));
        }
        b.out[p * out_stride + idx] = res;
    }
    if (t == 0) { b.status[p] = ST_OK; if (b.n3_out) b.n3_out[p] = n3; }
#ifdef SMRT_STAGE_TIMING
    SMRT_STAGE(SG_OUT);
    if (t == 0 && b.stage_out) {
        for (int k = 4; k < 16; ++k) b.stage_out[p * 16 + k] = (k < SG_COUNT) ? stage_acc[k] : 0.0;
        for (int k = 0; k < 3; ++k) b.stage_out[p * 16 + 13 + k] = sub_acc_store[k];
    }
#endif
}

}  // namespace smrt
