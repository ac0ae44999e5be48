// DORT hot path, device code (gfx950 / CDNA4).  One workgroup solves one (snowpack, frequency) pair with every
// N x N matrix (N = streams x polarisations <= 64 on the LDS path) resident in LDS.
//
// What is computed is fixed by the reference (paths relative to /root/reference); HOW is our own design:
//   layer electromagnetics   smrt/emmodel/iba.py:85-265, dmrt_qca_shortrange.py:65-112, permittivity/ice.py:52-73,
//                            permittivity/generic_mixing_formula.py:117-145, microstructure_model/*.py
//   streams                  smrt/rtsolver/streams.py:136-223,300-330
//   interfaces (Flat)        smrt/core/fresnel.py:99-146,417-474, smrt/rtsolver/rtsolver_utils.py:473-644
//   phase matrix modes       smrt/emmodel/common.py:9-131 (IBA: discrete azimuth mean), rayleigh.py:52-127
//   eigenproblem             smrt/rtsolver/dort.py:699-749 (matrix A), :891-962 (half-rank reduction)
//   boundary conditions      smrt/rtsolver/dort.py:263-488
//   Planck / interpolation   smrt/core/lib.py:594-620, smrt/rtsolver/rtsolver_utils.py:179-239
//
// Design: for azimuth mode 0 the reduced problem (alpha-beta)(alpha+beta) is similar to X- X+ with
// X+-, both symmetric positive definite (diagonal similarity by sqrt(norm*w/mu)).  With X+ = L+ L+^T and
// X- = L- L-^T, the singular values of B = L+^T L- are the eigenvalues beta, and the eigenvectors follow from
// B' = B V (one-sided Jacobi, wavefront-parallel column rotations in LDS) by one triangular solve and one
// triangular product.  The boundary system is solved by a bottom-up layer reflection-matrix recursion
// (two pivoted N x N solves per layer) instead of a banded LU of the global (2 N L) system.
//
// All storage is column-major with an ODD leading dimension LD so that row- and column-wise wavefront accesses
// are both LDS bank-conflict free (ds_read_b64: 32 eight-byte slots per 32-lane group).
#pragma once
#include "spmd.hpp"
#include <math.h>
#include <string.h>

namespace smrt {

// ------------------------------------------------------------------------------------------------------------
// batch descriptor as seen by the device
// ------------------------------------------------------------------------------------------------------------
struct DevBatch {
    int S, Lmax, F, n_theta;
    int emmodel, micro, mode, n_max_stream, m_max, normalization, rayleigh_jeans;
    int want_layer_out, want_stream_out;
    int jac_in_lds;  // global-workspace kernels: the Jacobi stage runs on an LDS copy of B (host: it fits)
    long long pair_begin, pair_count;
    const int* n_layers;
    const double* thickness;
    const double* frac_volume;
    const double* temperature;
    const double* p1;
    const double* p2;
    const double* frequency;
    const double* theta;
    const double* gl_mu;  // [n_max_stream] positive Gauss-Legendre nodes of order 2 n_max, descending
    int sub_kind;                         // 0 none, 1 flat (p1 + i p2 = permittivity), 2 reflector (p1, p2 = R_V, R_H)
    const double *sub_p1, *sub_p2;        // [F][S]
    const double* sub_T;                  // [S], <= 0: no emission
    const double *atm_down, *atm_up, *atm_trans;  // [F] or null
    double phi;
    double prune_tau;  // > 0: optical depth beyond which the deeper layers are dropped (dort.py:443-452); pipeline only
    double* out;
    int* status;
    double* layer_out;
    double* stream_out;
    double* n3_out;  // [pair_count] sum_l N_l^3 (work counter for the roofline)
    double* stage_out;  // [pair_count][16] shader cycles per stage (only written by -DSMRT_STAGE_TIMING builds)
};

// Staging area of the three-kernel pipeline (prep -> jacobi -> finish): per (pair, layer) the Cholesky factor L+,
// the matrix B = L+^T L- (replaced in place by B' = B V), the row scaling d, the singular values and N.
struct DevStage {
    double* L;
    double* B;
    double* d;
    double* sigma;
    int* n;
    long long mat_stride;  // doubles per matrix slot (NMAX * LD)
    int vec_stride;        // doubles per vector slot (NMAX)
    double* Linv;          // [item][4][256] inverses of the 16x16 diagonal blocks of L+ (written by the prep kernel)
};

constexpr double kCSpeed = 299792458.0;
constexpr double kPlanck = 6.62607015e-34;
constexpr double kBoltzmann = 1.380649e-23;
constexpr double kFreezing = 273.15;
constexpr double kPi = 3.14159265358979323846;

enum { EM_IBA = 0, EM_DMRT = 1, EM_QCACP = 2, EM_NONSCAT = 3 };  // every emmodel but IBA has a Rayleigh phase matrix
enum { MS_EXP = 0, MS_SHS = 1 };
enum { ST_OK = 0, ST_EIGEN = 1, ST_NORM = 2, ST_ALBEDO = 3, ST_SINGULAR = 4, ST_INPUT = 5 };
enum { SUB_NONE = 0, SUB_FLAT = 1, SUB_REFLECTOR = 2 };

// ------------------------------------------------------------------------------------------------------------
// LDS layout (shared by host sizing code and the kernel)
// ------------------------------------------------------------------------------------------------------------
struct LdsPlan {
    int NMAX, LD, nmax, Lmax, nphi, ntheta;
    int slim;             // 0: full layout, 1: prep kernel, 2: two-slot finish kernel (see make_plan)
    int matrices_in_lds;  // 1: the four N x N work matrices are LDS-resident; 0: they live in a global workspace
    int mat_doubles;      // doubles of matrix workspace per workgroup (4 * NMAX * LD)
    int o_M[4];
    int o_rowvec;   // 17 vectors of NMAX
    int o_strvec;   // 6 vectors of nmax
    int o_layvec;   // 11 vectors of Lmax
    int o_phi;      // 5 vectors of nphi
    int o_tb;       // NMAX
    int o_int;      // 16 ints (8 doubles)
    int o_gj;       // scratch of the blocked solvers (block inverses of Cholesky / triangular solve, Gauss-Jordan bookkeeping)
    int o_act;      // active mode only: per-layer mode-0 normalisation, mode totals, incident stream list
    int o_jac;      // global-workspace kernels: an NMAX x LD LDS buffer for the Jacobi stage, or -1 if it does not fit
    int total;      // doubles
};

#if defined(SMRT_HOST_EMU)
#define SMRT_HD inline
#else
#define SMRT_HD __host__ __device__ inline
#endif

// doubles of the active-mode region: norm0[Lmax][2 nmax], total[9][2 ntheta], coherent[2][2 ntheta], incident list
SMRT_HD int active_doubles(int n_max_stream, int Lmax, int ntheta) {
    return 2 * n_max_stream * Lmax + 9 * 2 * ntheta + 2 * 2 * ntheta + (2 * ntheta + 2) / 2 + 1;
}
// azimuth samples of the discrete Fourier decomposition of the phase function (emmodel/common.py:401-414)
SMRT_HD int azimuth_samples(int m_max) {
    int e = 4, v = 1;
    while (v < m_max + 1) { v *= 2; ++e; }
    return 1 << e;
}

// slim = 1: the "prep" kernel of the split pipeline -- two work matrices (X+- -> L+-), four row vectors and the
// Cholesky scratch only, so that TWO workgroups fit in the 160 KB of a CU.
// slim = 2: the two-slot "finish" kernel -- two work matrices (X, R), all row vectors, Gauss-Jordan bookkeeping only.
// jac_in_lds (global-workspace kernels only): reserve one LDS matrix for the Jacobi stage.
SMRT_HD LdsPlan make_plan(int n_max_stream, int P, int Lmax, int ntheta, int nphi, int matrices_in_lds = 1,
                          int act_doubles = 0, int slim = 0, int jac_in_lds = 0) {
    LdsPlan p;
    p.nmax = n_max_stream;
    p.NMAX = n_max_stream * P;
    p.LD = (p.NMAX + 1) | 1;  // odd (bank-conflict free rows and columns) and at least one padding row
    p.Lmax = Lmax;
    p.nphi = nphi;
    p.ntheta = ntheta;
    p.matrices_in_lds = matrices_in_lds;
    const int nmat = slim ? 2 : 4;
    p.mat_doubles = nmat * p.NMAX * p.LD;
    int o = 0;
    for (int i = 0; i < 4; ++i) { p.o_M[i] = (i < nmat ? i : 0) * p.NMAX * p.LD; }
    if (slim == 2) p.o_M[3] = p.NMAX * p.LD;  // the two-slot finish kernel: M0 = X, M3 = R (M1, M2 live in global memory)
    if (matrices_in_lds) o = p.mat_doubles;
    p.slim = slim;
    p.o_rowvec = o; o += (slim == 1 ? 4 : slim == 2 ? 14 : 17) * p.NMAX;  // slim 2: no mrow / wrow / u
    p.o_strvec = o; o += 6 * p.nmax;
    p.o_layvec = o; o += 11 * Lmax;
    p.o_phi = o; o += (slim == 2 ? 0 : 5 * nphi);
    p.o_tb = o; o += p.NMAX;
    p.o_int = o; o += 8;
    p.o_gj = o; o += slim == 1 ? 520 : slim == 2 ? 88 : !matrices_in_lds ? 2 * p.NMAX / 2 + p.NMAX / 2 + 32 : ((16 * p.NMAX + 8 > 1024 + 8) ? 16 * p.NMAX + 8 : 1024 + 8) + (p.NMAX + 8 + 1) / 2;
    p.o_act = o; o += act_doubles;
    p.o_jac = -1;
    // 1: a whole matrix (Jacobi stage of the fused kernel); 2: only the 16 NMAX doubles of scratch that the blocked
    // triangular solve needs (finish half of the global-workspace pipeline: small LDS, several workgroups per CU)
    if (jac_in_lds && !matrices_in_lds) { p.o_jac = o; o += (jac_in_lds == 2) ? 16 * p.NMAX : p.NMAX * p.LD; }
    p.total = o;
    return p;
}

struct Lds {
    double *M0, *M1, *M2, *M3;
    double *mrow, *wrow, *u, *d, *sigma, *rsig, *t, *Rtop, *Ttop, *Rbu, *Tbu, *cvec, *tq, *svec, *g, *upb, *up;
    double *gmu, *gsin, *outmu, *mu, *w, *muu;
    double *eps_re, *eps_im, *ks, *ka, *pa, *pb, *pc, *BT, *thick, *ri, *nl;
    double *cphi, *s2phi, *wphi, *sphi, *swphi;
    double* tb;
    double* act;  // active-mode region (see active_doubles)
    int* ints;  // [0] status  [1] jacobi flag  [2] pivot  [3] pivot fail  [4] kstar  [5] n_air
    double* gj;  // blocked Gauss-Jordan scratch
    int gj_nmax;
    double* sub_acc;  // profiling builds: [0] GJ panel cycles, [1] GJ update cycles, [2] GJ permutation cycles
};

SMRT_DEV Lds carve(double* base, double* mat_base, const LdsPlan& p) {
    Lds s;
    s.M0 = mat_base + p.o_M[0]; s.M1 = mat_base + p.o_M[1]; s.M2 = mat_base + p.o_M[2]; s.M3 = mat_base + p.o_M[3];
    const int n = p.NMAX;
    double* v = base + p.o_rowvec - (p.slim == 2 ? 3 * n : 0);  // slim 2: mrow / wrow / u do not exist (never touched)
    s.mrow = v; s.wrow = v + n; s.u = v + 2 * n; s.d = v + 3 * n; s.sigma = v + 4 * n; s.rsig = v + 5 * n;
    s.t = v + 6 * n; s.Rtop = v + 7 * n; s.Ttop = v + 8 * n; s.Rbu = v + 9 * n; s.Tbu = v + 10 * n;
    s.cvec = v + 11 * n; s.tq = v + 12 * n; s.svec = v + 13 * n; s.g = v + 14 * n; s.upb = v + 15 * n;
    s.up = v + 16 * n;
    v = base + p.o_strvec;
    const int m = p.nmax;
    s.gmu = v; s.gsin = v + m; s.outmu = v + 2 * m; s.mu = v + 3 * m; s.w = v + 4 * m; s.muu = v + 5 * m;
    v = base + p.o_layvec;
    const int L = p.Lmax;
    s.eps_re = v; s.eps_im = v + L; s.ks = v + 2 * L; s.ka = v + 3 * L; s.pa = v + 4 * L; s.pb = v + 5 * L;
    s.pc = v + 6 * L; s.BT = v + 7 * L; s.thick = v + 8 * L; s.ri = v + 9 * L; s.nl = v + 10 * L;
    v = base + p.o_phi;
    s.cphi = v; s.s2phi = v + p.nphi; s.wphi = v + 2 * p.nphi; s.sphi = v + 3 * p.nphi; s.swphi = v + 4 * p.nphi;
    s.act = base + p.o_act;
    s.tb = base + p.o_tb;
    s.ints = (int*)(base + p.o_int);
    s.gj = base + p.o_gj;
    s.gj_nmax = p.NMAX;
    s.sub_acc = nullptr;
    return s;
}

// ------------------------------------------------------------------------------------------------------------
// complex helpers
// ------------------------------------------------------------------------------------------------------------
struct cplx { double re, im; };
SMRT_DEV cplx cmk(double a, double b) { cplx z; z.re = a; z.im = b; return z; }
SMRT_DEV cplx cadd(cplx a, cplx b) { return cmk(a.re + b.re, a.im + b.im); }
SMRT_DEV cplx csub(cplx a, cplx b) { return cmk(a.re - b.re, a.im - b.im); }
SMRT_DEV cplx cmul(cplx a, cplx b) { return cmk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
SMRT_DEV cplx cscale(cplx a, double s) { return cmk(a.re * s, a.im * s); }
SMRT_DEV cplx cconj(cplx a) { return cmk(a.re, -a.im); }
SMRT_DEV double cabs2(cplx a) { return a.re * a.re + a.im * a.im; }
SMRT_DEV cplx cdiv(cplx a, cplx b) {
    double d = 1.0 / cabs2(b);
    return cmk((a.re * b.re + a.im * b.im) * d, (a.im * b.re - a.re * b.im) * d);
}
SMRT_DEV cplx csqrt_(cplx z) {  // principal branch
    if (z.re == 0.0 && z.im == 0.0) return cmk(0.0, 0.0);
    double m = sqrt(cabs2(z));
    double tt = sqrt(0.5 * (fabs(z.re) + m));
    if (z.re >= 0.0) return cmk(tt, z.im / (2.0 * tt));
    return cmk(fabs(z.im) / (2.0 * tt), z.im >= 0.0 ? tt : -tt);
}

// ------------------------------------------------------------------------------------------------------------
// layer electromagnetics
// ------------------------------------------------------------------------------------------------------------
SMRT_DEV cplx ice_permittivity(double frequency, double T) {  // Maetzler 2006, permittivity/ice.py:52-73
    double fg = frequency * 1e-9;
    double tc = T - kFreezing;
    double er = 3.1884 + 9.1e-4 * tc;
    double th = 300.0 / T - 1.0;
    double alpha = (0.00504 + 0.0062 * th) * exp(-22.1 * th);
    double eb = exp(335.0 / T);
    double betam = (0.0207 / T) * (eb / ((eb - 1.0) * (eb - 1.0))) + 1.16e-11 * fg * fg;
    double dbeta = exp(-9.963 + 0.0372 * tc);
    return cmk(er, alpha / fg + (betam + dbeta) * fg);
}

SMRT_DEV double sinc_(double x) { return x == 0.0 ? 1.0 : sin(x) / x; }

// FT of the autocorrelation function at wavenumber k (k2 = k*k)
SMRT_DEV double ft_corr(int micro, double k2, double fv, double p1, double p2) {
    if (micro == MS_EXP) {  // exponential.py:53-58
        double x = k2 * p1 * p1;
        double den = 1.0 + x;
        return fv * (1.0 - fv) * 8.0 * kPi * p1 * p1 * p1 / (den * den);
    }
    // sticky hard spheres, sticky_hard_spheres.py:63-130
    double f = fv, tau = p2, radius = p1;
    double x = sqrt(k2) * radius;
    double tt = 0.0;
    if (isfinite(tau) && f > 0.0) {
        double disc = 36 * tau * tau * f * f - 72 * tau * f * f - 72 * tau * tau * f + 30 * f * f + 72 * tau * f +
                      36 * tau * tau - 12 * f;
        tt = (6 * tau * f - 6 * f - 6 * tau + sqrt(disc)) / (f * (f - 1.0));
    }
    double vd = 4.0 / 3.0 * kPi * radius * radius * radius;
    double fr = f / (1.0 - f);
    double c1 = 1.0 - tt * f + 3.0 * fr;
    double c2 = 3.0 - tt * (1.0 - f);
    if (fabs(x) <= 1e-3) {
        double den = fr * (c1 + c2) + 1.0;
        return f * vd / (den * den);
    }
    double vint = 3.0 * (sinc_(x) - cos(x)) / (x * x);
    double psi = sinc_(x) / vint;
    double a = fr * (c1 + c2 * psi) + cos(x) / vint;
    double b = fr * x + sin(x) / vint;
    return f * vd / (a * a + b * b);
}

SMRT_DEV double shs_t(double f, double tau, int* bad) {  // sticky_hard_spheres.py:132-167
    if (isinf(tau)) return 0.0;
    double a = f / 12.0, b = -(tau + f / (1.0 - f)), c = (1.0 + 0.5 * f) / ((1.0 - f) * (1.0 - f));
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) { *bad = 1; return 0.0; }
    double sq = sqrt(disc);
    double tt = (-b - sq) / (2.0 * a);
    if (tt * f * (1.0 - f) > 1.0 + 2.0 * f) tt = (-b + sq) / (2.0 * a);
    if (tt * f * (1.0 - f) > 1.0 + 2.0 * f) *bad = 1;
    return tt;
}

SMRT_DEV double planck_radiance(double frequency, double T) {  // core/lib.py:594-607
    if (!(T > 1e-10)) return 0.0;
    return (2.0 * kPlanck / (kCSpeed * kCSpeed)) * frequency * frequency * frequency /
           expm1((kPlanck / kBoltzmann) * frequency / T);
}
SMRT_DEV double planck_inverse(double frequency, double radiance) {  // core/lib.py:610-620
    if (!(radiance > 1e-40)) return 0.0;
    double x = (2.0 * kPlanck / (kCSpeed * kCSpeed)) * frequency * frequency * frequency / radiance;
    return (kPlanck / kBoltzmann) * frequency / log1p(x);
}

// One layer: effective permittivity, ks, ka and the parameters of its phase function.
// pa/pb/pc: IBA+exponential -> C(cosT) = pa / (1 + pb (1 - cosT))^2 ; IBA+SHS -> pa = iba_coeff, pb = kfac^2/2;
// DMRT -> pa = 1.5 ks.
SMRT_DEV void layer_em(const DevBatch& b, double frequency, double fv, double T, double p1, double p2, cplx* eps_eff,
                       double* ks, double* ka, double* pa, double* pb, int* bad) {
    cplx es = ice_permittivity(frequency, T);
    if (T > kFreezing) *bad = 1;
    double k0 = 2.0 * kPi * frequency / kCSpeed;
    if (b.emmodel == EM_IBA) {
        // Polder-van Santen, spheres: 2x^2 + bx - eps e0 = 0 (generic_mixing_formula.py:117-145), e0 = 1
        cplx bq = csub(csub(es, cmk(2.0, 0.0)), cscale(csub(es, cmk(1.0, 0.0)), 3.0 * fv));
        cplx disc = cadd(cmul(bq, bq), cscale(es, 8.0));
        cplx ee = cscale(csub(csqrt_(disc), bq), 0.25);
        if (ee.im < -1e-10) *bad = 1;
        *eps_eff = ee;
        // mean squared field ratio with depolarisation factors 1/3 (iba.py:152-162)
        cplx app = cadd(cscale(ee, 2.0 / 3.0), cmk(1.0 / 3.0, 0.0));
        cplx den = cadd(app, cscale(csub(es, cmk(1.0, 0.0)), 1.0 / 3.0));
        double y2 = cabs2(cdiv(app, den));
        double coeff = (1.0 / (4.0 * kPi)) * cabs2(csub(es, cmk(1.0, 0.0))) * y2 * (k0 * k0) * (k0 * k0);
        cplx sq = csqrt_(ee);
        *ka = 2.0 * k0 * sq.im;  // iba.py:265
        // ks: Romberg on 65 samples of mu = 1 - j/32 (iba.py:176-226; scipy.integrate.romb), |sqrt(eps)| here
        double nabs2 = sqrt(cabs2(ee));  // |sqrt(eps)|^2 = |eps|
        double S[7];
        for (int i = 0; i < 7; ++i) S[i] = 0.0;
        double yend = 0.0;
        for (int j = 0; j <= 64; ++j) {
            double mu = 1.0 - j * 0.03125;
            double k2 = 4.0 * k0 * k0 * (0.5 * (1.0 - mu)) * nabs2;
            double y = coeff * ft_corr(b.micro, k2, fv, p1, p2) * (mu * mu + 1.0);
            if (j == 0 || j == 64) { yend += 0.5 * y; continue; }
            int tz = 0;
            while (((j >> tz) & 1) == 0) ++tz;
            for (int i = 6 - tz; i <= 6; ++i) S[i] += y;
        }
        double R[7];
        for (int i = 0; i < 7; ++i) R[i] = (double)(64 >> i) * 0.03125 * (yend + S[i]);
        double pw = 1.0;
        for (int j = 1; j <= 6; ++j) {
            pw *= 4.0;
            for (int i = 0; i <= 6 - j; ++i) R[i] = (pw * R[i + 1] - R[i]) / (pw - 1.0);
        }
        *ks = 0.25 * R[0];
        double kfac = 2.0 * k0 * sq.re;  // iba.py:233
        if (b.micro == MS_EXP) {
            *pa = coeff * fv * (1.0 - fv) * 8.0 * kPi * p1 * p1 * p1;
            *pb = 0.5 * kfac * kfac * p1 * p1;
        } else {
            *pa = coeff;
            *pb = 0.5 * kfac * kfac;
        }
    } else if (b.emmodel == EM_NONSCAT) {
        // non-scattering medium (nonscattering.py): Polder-van Santen permittivity, absorption only
        cplx bq = csub(csub(es, cmk(2.0, 0.0)), cscale(csub(es, cmk(1.0, 0.0)), 3.0 * fv));
        cplx ee = cscale(csub(csqrt_(cadd(cmul(bq, bq), cscale(es, 8.0))), bq), 0.25);
        *eps_eff = ee;
        *ka = 2.0 * k0 * csqrt_(ee).im;
        *ks = 0.0; *pa = 0.0; *pb = 0.0;
    } else if (b.emmodel == EM_QCACP) {
        // DMRT QCA-CP short range as in DMRT-ML (dmrt_qcacp_shortrange.py:63-125), dense_snow_correction="auto"
        double f = fv;
        cplx e0 = cmk(1.0, 0.0), e1 = es;
        if (f > 0.5) { f = 1.0 - f; e0 = es; e1 = cmk(1.0, 0.0); }
        int tb = 0;
        const double tt = shs_t(f, p2, &tb);
        if (tb) *bad = 1;
        const cplx de = csub(e1, e0);
        const cplx bq = csub(cscale(de, (1.0 - 4.0 * f) / 3.0), e0);
        const cplx cq = cscale(cmul(e0, de), -(1.0 - f) / 3.0);
        const cplx disc = csqrt_(csub(cmul(bq, bq), cscale(cq, 4.0)));
        cplx ee0 = cscale(csub(disc, bq), 0.5);
        if (ee0.re < 1.0) ee0 = cscale(cadd(disc, bq), -0.5);
        const double x = k0 * p1;  // 2 pi radius / lambda (vacuum wavelength)
        const double x3 = x * x * x;
        const double den = 1.0 + 2.0 * f - tt * f * (1.0 - f);
        const double omf4 = (1.0 - f) * (1.0 - f) * (1.0 - f) * (1.0 - f);
        const double shape = omf4 / (den * den);
        const cplx corr = cdiv(de, cadd(cmk(1.0, 0.0), cscale(cdiv(de, cscale(ee0, 3.0)), 1.0 - f)));
        const cplx fac = cadd(cmk(1.0, 0.0), cscale(cmul(cmul(cmk(0.0, 2.0 / 9.0 * x3), csqrt_(ee0)), corr), shape));
        const cplx ee = cadd(e0, cmul(csub(ee0, e0), fac));
        *eps_eff = ee;
        const double sqim = csqrt_(ee).im;
        const double albedo = 2.0 / 9.0 * x3 * f / (2.0 * sqim) * cabs2(corr) * shape;
        const double beta = 2.0 * k0 * sqim;
        *ks = albedo * beta;
        *ka = beta - albedo * beta;
        *pa = 1.5 * albedo * beta;
        *pb = 0.0;
    } else {
        // DMRT QCA short range (dmrt_qca_shortrange.py:65-112), dense_snow_correction="auto"
        double f = fv;
        cplx e0 = cmk(1.0, 0.0), e1 = es;
        if (f > 0.5) { f = 1.0 - f; e0 = es; e1 = cmk(1.0, 0.0); }
        int tb = 0;
        double tt = shs_t(f, p2, &tb);
        if (tb) *bad = 1;
        cplx y = cdiv(csub(e1, e0), cadd(e1, cscale(e0, 2.0)));
        cplx fy = cscale(y, f);
        double kk = k0 * csqrt_(e0).re;
        double kr3 = (kk * p1) * (kk * p1) * (kk * p1);
        double den = 1.0 + 2.0 * f - tt * f * (1.0 - f);
        double omf4 = (1.0 - f) * (1.0 - f) * (1.0 - f) * (1.0 - f);
        cplx one_m_fy = csub(cmk(1.0, 0.0), fy);
        // Eeff = e0 + 3 fy e0/(1-fy) * (1 + 2j/3 kr3 y (1-f)^4 / ((1-fy) den^2))
        cplx corr = cdiv(cscale(cmul(cmk(0.0, 2.0 / 3.0 * kr3 * omf4 / (den * den)), y), 1.0), one_m_fy);
        cplx fac = cadd(cmk(1.0, 0.0), corr);
        cplx ee = cadd(e0, cmul(cdiv(cmul(cscale(fy, 3.0), e0), one_m_fy), fac));
        *eps_eff = ee;
        double Ks = 2.0 / (9.0 * f) * kk * kr3 * (cabs2(csub(cdiv(ee, e0), cmk(1.0, 0.0))) * omf4 / (den * den));
        double beta = 2.0 * kk * csqrt_(ee).im;
        *ks = Ks;
        *ka = beta - Ks;
        *pa = 1.5 * Ks;
        *pb = 0.0;
    }
}

// Flat interface, Maezawa & Miyauchi 2009 "rigorous" Fresnel (core/fresnel.py:99-146): power R for V and H.
SMRT_DEV void fresnel_RvRh(cplx e1, cplx e2, double mu1, double* Rv, double* Rh) {
    cplx n1 = csqrt_(e1);
    double kz2 = n1.re * n1.re * (1.0 - mu1 * mu1);
    cplx kyi = cscale(csqrt_(cmk(e1.re - kz2, e1.im)), -1.0);
    cplx kyt = cscale(csqrt_(cmk(e2.re - kz2, e2.im)), -1.0);
    cplx rh = cdiv(csub(kyi, kyt), cadd(cconj(kyi), kyt));
    cplx num = cmul(cconj(n1), csub(cmul(e2, kyi), cmul(e1, kyt)));
    cplx den = cmul(n1, cadd(cmul(e2, cconj(kyi)), cmul(cconj(e1), kyt)));
    cplx rv = cdiv(num, den);
    *Rv = cabs2(rv);
    *Rh = cabs2(rh);
}

// ------------------------------------------------------------------------------------------------------------
// dense kernels in LDS.  Element (r, c) of every matrix lives at [c * LD + r].
// ------------------------------------------------------------------------------------------------------------
// Power-of-two 2-D tiling of an (R rows) x (C columns) index space over the workgroup without integer division:
// a wavefront covers RW = pow2 >= min(R, 64) rows and 64 / RW columns at a time (consecutive lanes -> consecutive
// rows -> consecutive LDS addresses).  body(r, c) is called for every r < R, c < C exactly once.
struct Tile2D { int rmask, cshift, cols_per_wave; };
SMRT_DEV Tile2D make_tile(int R) {
    Tile2D t;
    int sh = 6;                       // RW = 64
    if (R <= 32) sh = 5;
    if (R <= 16) sh = 4;
    if (R <= 8) sh = 3;
    if (R <= 4) sh = 2;
    t.rmask = (1 << sh) - 1; t.cshift = sh; t.cols_per_wave = SMRT_LANES >> sh;
    return t;
}
template <int NT, class Body>
SMRT_DEV void for_2d(int R, int C, Body body) {
    const int t = tid();
    const int lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    const Tile2D tl = make_tile(R);
    const int rl = lane & tl.rmask, cs = lane >> tl.cshift;
    for (int r0 = 0; r0 < R; r0 += SMRT_LANES) {
        const int r = r0 + rl;
        for (int c = wave * tl.cols_per_wave + cs; c < C; c += NW * tl.cols_per_wave)
            if (r < R) body(r, c);
    }
}

template <int NT>
SMRT_DEV bool chol2(double* A, double* Bm, int N, int LD) {
    // Two right-looking Cholesky factorisations side by side (lower triangles, in place), ONE barrier per column:
    // during step k column k stays unscaled (read-only) and the trailing update carries the 1/pivot factor; the
    // columns are scaled by 1/sqrt(pivot) in a final pass.
    const int t = tid();
    for (int k = 0; k < N; ++k) {
        const double akk = A[k * LD + k], bkk = Bm[k * LD + k];
        if (!(akk > 0.0) || !(bkk > 0.0)) return false;  // uniform: every thread reads the same words
        const int m = N - k - 1;
        if (m > 0) {
            const double ra = fast_rcp(akk), rb = fast_rcp(bkk);
            for_2d<NT>(m, m, [&](int ri, int ci) {
                const int i = k + 1 + ri, j = k + 1 + ci;
                if (i >= j) {
                    A[j * LD + i] -= A[k * LD + i] * (A[k * LD + j] * ra);
                    Bm[j * LD + i] -= Bm[k * LD + i] * (Bm[k * LD + j] * rb);
                }
            });
            block_sync();
        }
    }
    for_2d<NT>(N, N, [&](int i, int k) {
        if (i > k) {
            A[k * LD + i] *= fast_rsqrt(A[k * LD + k]);
            Bm[k * LD + i] *= fast_rsqrt(Bm[k * LD + k]);
        }
    });
    block_sync();
    for (int k = t; k < N; k += NT) {
        const double akk = A[k * LD + k], bkk = Bm[k * LD + k];
        A[k * LD + k] = akk * fast_rsqrt(akk);
        Bm[k * LD + k] = bkk * fast_rsqrt(bkk);
    }
    block_sync();
    return true;
}

// C = Lp^T Lm for lower-triangular Lp, Lm
template <int NT>
SMRT_DEV void lt_times_l(const double* Lp, const double* Lm, double* C, int N, int LD) {
    for_2d<NT>(N, N, [&](int i, int j) {
        double acc = 0.0;
        for (int k = (i > j ? i : j); k < N; ++k) acc += Lp[i * LD + k] * Lm[j * LD + k];
        C[j * LD + i] = acc;
    });
    block_sync();
}

// One-sided (Hestenes) Jacobi, two-level ordering.
//
// The columns are cut into NB = 2 * (number of wavefronts) blocks of m columns.  An outer round-robin over the
// blocks gives every wavefront one pair of blocks (I, J) per outer step; inside the step the wavefront rotates all
// m*m cross pairs (m inner steps of m disjoint pairs) -- and, at the first outer step of a sweep, the pairs inside
// its two blocks -- touching only its own 2m columns, so the inner steps need a wavefront-level sync only.  One
// workgroup barrier per OUTER step (NB-1 per sweep instead of N-1).
// GS lanes own one column pair and keep their RPL rows of both columns in registers between the three dot products
// and the rotation.  A sweep in which no pair had cos^2 > 1e-15 before its rotation is the last one (the residual
// non-orthogonality is second order).  On exit sigma[c] = |column c|, rsig[c] = 1/sigma[c].
#ifndef SMRT_JACOBI_EXIT_COS2
#define SMRT_JACOBI_EXIT_COS2 1e-15
#endif
// rotations between columns whose cosine is already below 1e-13 are skipped in the split-pipeline kernel: they
// cannot change the result at the 1e-9 relative level of the parity requirement (1e-6 K), and in the last sweeps
// most pairs are in that state (saves the update + store half of the step)
#ifndef SMRT_JACOBI_SKIP_COS2
#define SMRT_JACOBI_SKIP_COS2 1e-26
#endif
template <int GS, int RPL>
SMRT_DEV void rotate_pair(double* Bm, int LD, int N, int p, int q, bool valid, int sub, int slot, int* flag) {
    // Branch-free: every lane always loads and stores its RPL rows.  Rows >= N of a column are padding inside the
    // N_max x LD buffer (never read by any other stage), loads from them are masked to zero with a select instead
    // of being predicated (per-element exec-mask branches were costing more than the arithmetic).
    double x[RPL], y[RPL];
    double a = 0.0, bb = 0.0, gg = 0.0, a2 = 0.0, bb2 = 0.0, gg2 = 0.0;  // two accumulators: half the FMA chain
    double* cp = Bm + p * LD;
    double* cq = Bm + q * LD;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int r0 = sub + i * GS;
        const int r = r0 < LD - 1 ? r0 : LD - 1;  // row LD-1 is always padding (LD = N_max + 1)
        const bool in = valid && (r0 < N);
        const double xv = cp[r], yv = cq[r];
        x[i] = in ? xv : 0.0;
        y[i] = in ? yv : 0.0;
        if (i & 1) { a2 += x[i] * x[i]; bb2 += y[i] * y[i]; gg2 += x[i] * y[i]; }
        else { a += x[i] * x[i]; bb += y[i] * y[i]; gg += x[i] * y[i]; }
    }
    a += a2; bb += bb2; gg += gg2;
    a = group_sum<GS>(a); bb = group_sum<GS>(bb); gg = group_sum<GS>(gg);
    const double g2 = gg * gg, ab = a * bb;
    if (valid && g2 > 1e-30 * ab) {
        // tan of the rotation angle: t = 2 g sign(d) / (|d| + sqrt(d^2 + 4 g^2)), d = b - a
        const double dd = bb - a;
        const double hh = dd * dd + 4.0 * g2;
        const double h = hh * fast_rsqrt1(hh);
        const double tt = (dd >= 0.0 ? 2.0 : -2.0) * gg * fast_rcp1(fabs(dd) + h);  // angle only: 1 Newton step
        const double c = fast_rsqrt(1.0 + tt * tt), sn = c * tt;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r0 = sub + i * GS;
            const int r = r0 < LD - 1 ? r0 : LD - 1;
            cp[r] = c * x[i] - sn * y[i];
            cq[r] = sn * x[i] + c * y[i];
        }
        if (sub == 0 && g2 > SMRT_JACOBI_EXIT_COS2 * ab) lds_or(flag, 1);
    }
}

template <int NT, int JW, int GS, int RPL>
SMRT_DEV bool jacobi_onesided(double* Bm, int N, int LD, double* sigma, double* rsig, int* flag, int* n_sweeps,
                              double* sub_acc = nullptr) {
    // JW = wavefronts that take part (the others only meet the workgroup barriers): with few lanes per pair and many
    // rows per lane the fixed per-rotation cost (index math, reductions, rotation parameters) is amortised better
    // than by spreading every pair over more lanes of more wavefronts.
    const int t = tid();
    const int lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES;
    constexpr int NB = 2 * JW;               // column blocks
    constexpr int SLOTS = SMRT_LANES / GS;   // column pairs a wavefront rotates at once
    const int slot = lane / GS, sub = lane % GS;
    const int m = (N + NB - 1) / NB;         // columns per block
    const int me = m + (m & 1);              // even player count of the in-block tournament
    bool converged = false;
#ifdef SMRT_STAGE_TIMING
    long long tj0 = cycle_counter();
#define SMRT_JSUB(k) do { const long long n_ = cycle_counter(); if (t == 0 && sub_acc) sub_acc[k] += (double)(n_ - tj0); tj0 = n_; } while (0)
#else
#define SMRT_JSUB(k) do {} while (0)
#endif
    for (int sweep = 0; sweep < 40 && !converged; ++sweep) {
        block_sync();  // everyone has read the previous flag
        if (t == 0) *flag = 0;
        block_sync();
        SMRT_JSUB(5);
        for (int s = 0; s < NB - 1; ++s) {
            if (wave >= JW) { block_sync(); continue; }
            int I, J;
            if (wave == 0) { I = NB - 1; J = s; }
            else {
                I = s + wave; if (I >= NB - 1) I -= NB - 1;
                J = s - wave; if (J < 0) J += NB - 1;
            }
            const int i0 = I * m, j0 = J * m;
            if (s == 0 && m > 1) {
                // pairs inside block I and inside block J: (me - 1) steps of me/2 pairs per block
                const int half = me / 2;
                for (int u = 0; u < me - 1; ++u) {
                    for (int ps0 = 0; ps0 < 2 * half; ps0 += SLOTS) {  // uniform trip count over the wavefront
                        const int ps = ps0 + slot;
                        const int base = (ps < half) ? i0 : j0;
                        const int k = (ps < half) ? ps : ps - half;
                        int a, b;
                        if (k == 0) { a = me - 1; b = u; }
                        else {
                            a = u + k; if (a >= me - 1) a -= me - 1;
                            b = u - k; if (b < 0) b += me - 1;
                        }
                        const int p = base + a, q = base + b;
                        const bool valid = (ps < 2 * half) && (a < m) && (b < m) && (p < N) && (q < N);
                        rotate_pair<GS, RPL>(Bm, LD, N, valid ? p : 0, valid ? q : 0, valid, sub, slot, flag);
                    }
                    wave_sync_lds();
                }
                SMRT_JSUB(3);
            }
            for (int j = 0; j < m; ++j) {
                for (int ps0 = 0; ps0 < m; ps0 += SLOTS) {  // uniform trip count over the wavefront
                    const int ps = ps0 + slot;
                    int bq = ps + j; if (bq >= m) bq -= m;
                    const int p = i0 + ps, q = j0 + bq;
                    const bool valid = (ps < m) && (p < N) && (q < N);
                    rotate_pair<GS, RPL>(Bm, LD, N, valid ? p : 0, valid ? q : 0, valid, sub, slot, flag);
                }
                wave_sync_lds();
            }
            SMRT_JSUB(3);
            block_sync();
            SMRT_JSUB(4);
        }
        converged = (*flag == 0);
        ++*n_sweeps;
    }
    block_sync();
    // column norms
    {
        constexpr int NG = NT / GS;
        const int grp = t / GS;
        const int rounds2 = (N + NG - 1) / NG;
        for (int rd = 0; rd < rounds2; ++rd) {
            const int c = grp + rd * NG;
            double a = 0.0;
            if (c < N)
                for (int r = sub; r < N; r += GS) { const double xx = Bm[c * LD + r]; a += xx * xx; }
            a = group_sum<GS>(a);
            if (c < N && sub == 0) { const double rs = fast_rsqrt(a); sigma[c] = a * rs; rsig[c] = rs; }
        }
    }
    block_sync();
    return converged;
}

// C = Lp * Bm (Lp lower triangular)
template <int NT>
SMRT_DEV void l_times_m(const double* Lp, const double* Bm, double* C, int N, int LD) {
    for_2d<NT>(N, N, [&](int i, int c) {
        double acc = 0.0;
        for (int k = 0; k <= i; ++k) acc += Lp[k * LD + i] * Bm[c * LD + k];
        C[c * LD + i] = acc;
    });
    block_sync();
}

// Bm <- Lp^-T Bm (back substitution with the upper-triangular Lp^T, all columns at once)
template <int NT>
SMRT_DEV void lt_solve(const double* Lp, double* Bm, int N, int LD) {
    for (int i = N - 1; i >= 1; --i) {
        const double rd = fast_rcp(Lp[i * LD + i]);
        for_2d<NT>(i, N, [&](int r, int c) { Bm[c * LD + r] -= Lp[r * LD + i] * (Bm[c * LD + i] * rd); });
        block_sync();
    }
    for_2d<NT>(N, N, [&](int i, int c) { Bm[c * LD + i] *= fast_rcp(Lp[i * LD + i]); });
    block_sync();
}

// Solve A X = Bm (+ one extra right-hand-side vector v, may be null) by LU with partial pivoting; X overwrites
// Bm / v, A is destroyed.  TR selects the storage view: element (r, c) at [c*LD + r] (false) or [r*LD + c]
// (true, i.e. the routine then solves A^T X^T = Bm^T on the same buffers).
// Every thread scans the pivot column itself (LDS broadcast reads): no cross-lane reduction and no barrier
// between the search and the row swap.
template <bool TR>
SMRT_DEV double& at(double* M, int r, int c, int LD) { return TR ? M[r * LD + c] : M[c * LD + r]; }

template <int NT, bool TR>
SMRT_DEV bool lu_solve(double* A, double* Bm, double* v, double* udiag, int N, int LD) {
    // Gauss-Jordan elimination with partial pivoting: every step eliminates column k from ALL other rows, so there
    // is no back-substitution phase (one third fewer barriers than LU + back substitution; the path is latency
    // bound, not flop bound).  Column k is never written once step k starts: the row swap skips it (the multipliers
    // are taken from the unswapped column) and the pivot goes to udiag[k]; that makes the per-wavefront pivot
    // search race-free against the swap of faster wavefronts without an extra barrier.
    const int t = tid();
    const int lane = t & (SMRT_LANES - 1);
    const int nv = (v != nullptr) ? 1 : 0;
    for (int k = 0; k < N; ++k) {
        // pivot row: every wavefront finds it on its own (one LDS load per lane, DPP arg-max on a key made of the
        // magnitude bits with the row index in the 8 low mantissa bits: exactness of the choice is irrelevant)
        unsigned long long key = 0ull;
        for (int r = k + lane; r < N; r += SMRT_LANES) {
            const double xr = fabs(at<TR>(A, r, k, LD));
            unsigned long long bits;
            memcpy(&bits, &xr, 8);
            bits = (bits & ~0xFFull) | (unsigned long long)(255 - (r - k < 255 ? r - k : 255));
            if (bits > key) key = bits;
        }
        key = wave_max_u64(key);
        if (key < 256ull) return false;  // zero column: singular, uniform exit
        const int p = k + 255 - (int)(key & 0xFFull);
        const double pv = at<TR>(A, p, k, LD);
        const double akk = at<TR>(A, k, k, LD);
        if (!(fabs(pv) > 0.0 && fabs(pv) < 1e300)) return false;  // uniform
        if (p != k) {
            const int na = N - k - 1;
            for (int idx = t; idx < na + N + nv; idx += NT) {
                if (idx < na) {
                    const int c = k + 1 + idx;
                    const double xx = at<TR>(A, k, c, LD);
                    at<TR>(A, k, c, LD) = at<TR>(A, p, c, LD);
                    at<TR>(A, p, c, LD) = xx;
                } else if (idx < na + N) {
                    const int c = idx - na;
                    const double xx = at<TR>(Bm, k, c, LD);
                    at<TR>(Bm, k, c, LD) = at<TR>(Bm, p, c, LD);
                    at<TR>(Bm, p, c, LD) = xx;
                } else {
                    const double xx = v[k]; v[k] = v[p]; v[p] = xx;
                }
            }
            block_sync();
        }
        if (t == 0) udiag[k] = pv;
        const double rp = fast_rcp(pv);
        const int m = N - k - 1;
        for_2d<NT>(N - 1, m + N + nv, [&](int ri, int cc) {
            const int r = ri + (ri >= k ? 1 : 0);  // every row but k
            const double l = ((r == p) ? akk : at<TR>(A, r, k, LD)) * rp;
            if (cc < m) {
                const int c = k + 1 + cc;
                at<TR>(A, r, c, LD) -= l * at<TR>(A, k, c, LD);
            } else if (cc < m + N) {
                const int c = cc - m;
                at<TR>(Bm, r, c, LD) -= l * at<TR>(Bm, k, c, LD);
            } else {
                v[r] -= l * v[k];
            }
        });
        block_sync();
    }
    for_2d<NT>(N, N + nv, [&](int i, int c) {
        const double rd = fast_rcp(udiag[i]);
        if (c < N) at<TR>(Bm, i, c, LD) *= rd;
        else v[i] *= rd;
    });
    block_sync();
    return true;
}

// ---- 16x16 tile GEMM on the FP64 matrix core ---------------------------------------------------------------
// c (the 16x16 tile at tile-row ti, tile-column tj, in MFMA accumulator layout) += sum_k A[i][k] B[k][j];
// fa(i, k) / fb(k, j) fetch operands (they must return 0 outside the matrix); K is rounded up to 4.
template <class FA, class FB>
SMRT_DEV void gemm_tile(double (&c)[4], int K, int ti, int tj, FA fa, FB fb) {
    const int lane = tid() & (SMRT_LANES - 1), lr = lane & 15, lk = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 4) mfma_f64_16x16x4(fa(ti * 16 + lr, k0 + lk), fb(k0 + lk, tj * 16 + lr), c);
}
// store / load an accumulator tile to a column-major matrix (element (r, c) at [c*LD + r]), rows/cols < N only
template <class F>
SMRT_DEV void tile_foreach(int ti, int tj, int N, F f) {
    const int lane = tid() & (SMRT_LANES - 1), lr = lane & 15, lk = lane >> 4;
    const int col = tj * 16 + lr;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = ti * 16 + lk + 4 * reg;
        if (row < N && col < N) f(reg, row, col);
    }
}

// C = Lp^T Lm (both lower triangular; whatever sits above their diagonals is ignored)
// reverse_cols: column c of the product is stored as column N-1-c.  The column norms of B = L+^T L- grow with the
// column index (beta ~ ke / mu, mu descending); the one-sided Jacobi converges in fewer sweeps when the large
// columns come first (de Rijk), and nothing downstream depends on the order of the eigenpairs.
template <int NT>
SMRT_DEV void lt_times_l_mfma(const double* Lp, const double* Lm, double* C, int N, int LD, bool reverse_cols = false) {
    const int wave = tid() / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    const int RT = (N + 15) >> 4;
    for (int tix = wave; tix < RT * RT; tix += NW) {
        const int ti = tix % RT, tj = tix / RT;
        double c[4] = {0.0, 0.0, 0.0, 0.0};
        const int kmin = 16 * (ti > tj ? ti : tj);  // k >= max(i, j)
        const int lane = tid() & (SMRT_LANES - 1), lr = lane & 15, lk = lane >> 4;
        const int i = ti * 16 + lr, j = tj * 16 + lr;
        const int ic = i < N ? i : N - 1, jc = j < N ? j : N - 1;
        for (int k0 = kmin; k0 < N; k0 += 4) {
            const int k = k0 + lk, kc = k < N ? k : N - 1;
            const double av = Lp[ic * LD + kc], bv = Lm[jc * LD + kc];
            mfma_f64_16x16x4((i < N && k < N && k >= i) ? av : 0.0, (j < N && k < N && k >= j) ? bv : 0.0, c);
        }
        tile_foreach(ti, tj, N, [&](int reg, int row, int col) { C[(reverse_cols ? N - 1 - col : col) * LD + row] = c[reg]; });
    }
    block_sync();
}

// C = Lp * Bm (Lp lower triangular)
template <int NT>
SMRT_DEV void l_times_m_mfma(const double* Lp, const double* Bm, double* C, int N, int LD) {
    const int wave = tid() / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    const int RT = (N + 15) >> 4;
    for (int tix = wave; tix < RT * RT; tix += NW) {
        const int ti = tix % RT, tj = tix / RT;
        double c[4] = {0.0, 0.0, 0.0, 0.0};
        const int lane = tid() & (SMRT_LANES - 1), lr = lane & 15, lk = lane >> 4;
        const int i = ti * 16 + lr, j = tj * 16 + lr;
        const int ic = i < N ? i : N - 1, jc = j < N ? j : N - 1;
        const int kend = (ti * 16 + 16 < N) ? ti * 16 + 16 : N;  // k <= i
        for (int k0 = 0; k0 < kend; k0 += 4) {
            const int k = k0 + lk, kc = k < N ? k : N - 1;
            const double av = Lp[kc * LD + ic], bv = Bm[jc * LD + kc];
            mfma_f64_16x16x4((i < N && k <= i) ? av : 0.0, (j < N && k < N) ? bv : 0.0, c);
        }
        tile_foreach(ti, tj, N, [&](int reg, int row, int col) { C[col * LD + row] = c[reg]; });
    }
    block_sync();
}

// ---- the two "row block times matrix" passes of the layer recursion on the matrix core (N <= 64) -------------
// Every wavefront owns one (or, for small workgroups, a few) 16-row tile(s): it first pulls the A operands of its
// rows -- the whole 16 x N row block, 16 registers per lane -- into registers, the workgroup synchronises, and only
// then are results written; that is what makes the in-place updates (rows of Rt, rows of F) safe.
template <int NT>
struct RowTiles {
    static constexpr int NW = NT / SMRT_LANES;
    static constexpr int RPW = (4 + NW - 1) / NW;               // row tiles per wavefront (RT <= 4)
    static constexpr int CS = (NW >= 4) ? NW / 4 : 1;           // wavefronts sharing one row tile (column split)
};

// Wk = F - Rt G ; Rt <- Rt F - G (in place) ; cvec = (Rt 1) B - B + svec
template <int NT>
SMRT_DEV void r1_mfma(const double* F, const double* G, double* Rt, double* Wk, double* cvec, const double* svec,
                      double Bl, int N, int LD) {
    using RTc = RowTiles<NT>;
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
    double a[RTc::RPW][16];
    int tis[RTc::RPW];
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        tis[o] = ti;
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        double rs = 0.0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double x = Rt[kc * LD + ic];
            a[o][kk] = (ti < RT && i < N && k < N) ? x : 0.0;
            rs += a[o][kk];
        }
        rs += shfl_xor(rs, 16);
        rs += shfl_xor(rs, 32);
        const bool owner = (RTc::NW >= 4) ? (wave < 4) : true;
        if (owner && ti < RT && lk == 0 && i < N) cvec[i] = rs * Bl - Bl + svec[i];
    }
    block_sync();
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = tis[o];
        if (ti >= RT) continue;
        const int cs = (RTc::NW >= 4) ? (wave >> 2) : 0;
        for (int tj = cs; tj < RT; tj += RTc::CS) {
            double c1[4] = {0.0, 0.0, 0.0, 0.0}, c2[4] = {0.0, 0.0, 0.0, 0.0};
            const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                if (4 * kk < N) {
                    const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
                    const bool in = (j < N && k < N);
                    const double gv = G[jc * LD + kc], fv = F[jc * LD + kc];
                    mfma_f64_16x16x4(a[o][kk], in ? gv : 0.0, c1);
                    mfma_f64_16x16x4(a[o][kk], in ? fv : 0.0, c2);
                }
            }
            tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                Wk[col * LD + row] = F[col * LD + row] - c1[reg];
                Rt[col * LD + row] = c2[reg] - G[col * LD + row];
            });
        }
    }
    block_sync();
}

// Y = F tQt + G -> Wk ; W = (G - Rtop F) tQt + (F - Rtop G) -> over F (in place)
// upb = F tq + B ; g = (G - Rtop F) tq + (1 - Rtop) B
// r1_mfma in two halves for the two-slot finish kernel.  r1_load pulls the A operands (the rows of R~ of this
// wavefront's row tile) into registers and writes cvec -- after it slot R is free.  r1_compute runs the MFMA loops
// with both B operands in LDS (Gl in slot X, Fl in slot R: staged there by the F/G formation), keeps the results in
// registers until every wavefront is done reading, and then writes Wk = F - R~ G over Gl and R~ F - G over Fl.
template <int NT>
SMRT_DEV void r1_load(const double* Rt, double (&a)[RowTiles<NT>::RPW][16], double* cvec, const double* svec, double Bl,
                      int N, int LD, const double* colsign = nullptr /* active mode: R~ D, D = +-1 per column */) {
    using RTc = RowTiles<NT>;
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        double rs = 0.0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double x = Rt[kc * LD + ic] * (colsign ? colsign[kc] : 1.0);
            a[o][kk] = (ti < RT && i < N && k < N) ? x : 0.0;
            rs += a[o][kk];
        }
        rs += shfl_xor(rs, 16);
        rs += shfl_xor(rs, 32);
        const bool owner = (RTc::NW >= 4) ? (wave < 4) : true;
        if (owner && ti < RT && lk == 0 && i < N) cvec[i] = rs * Bl - Bl + svec[i];
    }
    block_sync();
}

template <int NT>
SMRT_DEV void r1_compute(double* Fl /* slot R */, double* Gl /* slot X */, const double (&a)[RowTiles<NT>::RPW][16],
                         int N, int LD) {
    using RTc = RowTiles<NT>;
    constexpr int MAXTJ = (4 + RTc::CS - 1) / RTc::CS;
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
    double r1[RTc::RPW][MAXTJ][4], r2[RTc::RPW][MAXTJ][4];
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        const int cs = (RTc::NW >= 4) ? (wave >> 2) : 0;
#pragma unroll
        for (int q = 0; q < MAXTJ; ++q) {
            const int tj = cs + q * RTc::CS;
            double c1[4] = {0.0, 0.0, 0.0, 0.0}, c2[4] = {0.0, 0.0, 0.0, 0.0};
            if (ti < RT && tj < RT) {
                const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    if (4 * kk < N) {
                        const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
                        const bool in = (j < N && k < N);
                        const double gv = Gl[jc * LD + kc], fv = Fl[jc * LD + kc];
                        mfma_f64_16x16x4(a[o][kk], in ? gv : 0.0, c1);
                        mfma_f64_16x16x4(a[o][kk], in ? fv : 0.0, c2);
                    }
                }
                tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                    c1[reg] = Fl[col * LD + row] - c1[reg];   // Wk
                    c2[reg] = c2[reg] - Gl[col * LD + row];   // R~ F - G
                });
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { r1[o][q][reg] = c1[reg]; r2[o][q][reg] = c2[reg]; }
        }
    }
    block_sync();  // every wavefront has finished reading F and G
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        const int cs = (RTc::NW >= 4) ? (wave >> 2) : 0;
#pragma unroll
        for (int q = 0; q < MAXTJ; ++q) {
            const int tj = cs + q * RTc::CS;
            if (ti < RT && tj < RT)
                tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                    Gl[col * LD + row] = r1[o][q][reg];
                    Fl[col * LD + row] = r2[o][q][reg];
                });
        }
    }
    block_sync();
}

// SIGNED (azimuth modes m >= 1, three polarisations): the down-going eigenvectors carry the row signs
// dsg = (+1, +1, -1) per (V, H, U) (dort.py:951-953), i.e. W = (D G - Rtop F) tQt + (D F - Rtop G).
template <int NT, bool SIGNED = false>
SMRT_DEV void r45_mfma(double* F, const double* G, const double* Q, double* Wk, const double* Rtop, const double* tq,
                       double* upb, double* gvec, double Bl, int N, int LD, const double* dsg = nullptr) {
    using RTc = RowTiles<NT>;
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
    double af[RTc::RPW][16], aw[RTc::RPW][16];
    int tis[RTc::RPW];
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        tis[o] = ti;
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        const double rt = Rtop[ic];
        const double sg = SIGNED ? dsg[ic] : 1.0;
        double vy = 0.0, vg = 0.0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double fv = F[kc * LD + ic], gv = G[kc * LD + ic], tk = tq[kc];
            const bool in = (ti < RT && i < N && k < N);
            af[o][kk] = in ? fv : 0.0;
            aw[o][kk] = in ? (SIGNED ? sg * gv : gv) - rt * fv : 0.0;
            vy += af[o][kk] * tk;
            vg += aw[o][kk] * tk;
        }
        vy += shfl_xor(vy, 16); vy += shfl_xor(vy, 32);
        vg += shfl_xor(vg, 16); vg += shfl_xor(vg, 32);
        const bool owner = (RTc::NW >= 4) ? (wave < 4) : true;
        if (owner && ti < RT && lk == 0 && i < N) { upb[i] = vy + Bl; gvec[i] = vg + (1.0 - rt) * Bl; }
    }
    block_sync();
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = tis[o];
        if (ti >= RT) continue;
        const int cs = (RTc::NW >= 4) ? (wave >> 2) : 0;
        for (int tj = cs; tj < RT; tj += RTc::CS) {
            double cy[4] = {0.0, 0.0, 0.0, 0.0}, cw[4] = {0.0, 0.0, 0.0, 0.0};
            const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                if (4 * kk < N) {
                    const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
                    const double qv = Q[jc * LD + kc];
                    const double bop = (j < N && k < N) ? qv : 0.0;
                    mfma_f64_16x16x4(af[o][kk], bop, cy);
                    mfma_f64_16x16x4(aw[o][kk], bop, cw);
                }
            }
            tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                const double fic = F[col * LD + row], gic = G[col * LD + row];
                Wk[col * LD + row] = cy[reg] + gic;
                F[col * LD + row] = cw[reg] + (SIGNED ? dsg[row] * fic : fic) - Rtop[row] * gic;
            });
        }
    }
    block_sync();
}

// Two-slot variant for the finish kernel whose F and G live in global memory: Y and W are held in registers until
// every wavefront has finished reading Q, then Y goes to Yout and W OVER Q (Wout == Q is allowed).
template <int NT, bool SIGNED = false>
SMRT_DEV void r45_mfma2(const double* F, const double* G, const double* Q, double* Yout, double* Wout,
                        const double* Rtop, const double* tq, double* upb, double* gvec, double Bl, int N, int LD,
                        const double* dsg = nullptr) {
    using RTc = RowTiles<NT>;
    constexpr int MAXTJ = (4 + RTc::CS - 1) / RTc::CS;
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
    double ys[RTc::RPW][MAXTJ][4], ws[RTc::RPW][MAXTJ][4];
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        const double rt = Rtop[ic];
        const double sg = SIGNED ? dsg[ic] : 1.0;   // row sign of the down-going eigenvectors (+-1)
        double af[16], aw[16];
        double vy = 0.0, vg = 0.0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double fv = F[kc * LD + ic], gv = G[kc * LD + ic], tk = tq[kc];
            const bool in = (ti < RT && i < N && k < N);
            af[kk] = in ? fv : 0.0;
            aw[kk] = in ? (SIGNED ? sg * gv : gv) - rt * fv : 0.0;
            vy += af[kk] * tk;
            vg += aw[kk] * tk;
        }
        vy += shfl_xor(vy, 16); vy += shfl_xor(vy, 32);
        vg += shfl_xor(vg, 16); vg += shfl_xor(vg, 32);
        const bool owner = (RTc::NW >= 4) ? (wave < 4) : true;
        if (owner && ti < RT && lk == 0 && i < N) { upb[i] = vy + Bl; gvec[i] = vg + (1.0 - rt) * Bl; }
        const int cs = (RTc::NW >= 4) ? (wave >> 2) : 0;
#pragma unroll
        for (int q = 0; q < MAXTJ; ++q) {
            const int tj = cs + q * RTc::CS;
            double cy[4] = {0.0, 0.0, 0.0, 0.0}, cw[4] = {0.0, 0.0, 0.0, 0.0};
            if (ti < RT && tj < RT) {
                const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    if (4 * kk < N) {
                        const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
                        const double qv = Q[jc * LD + kc];
                        const double bop = (j < N && k < N) ? qv : 0.0;
                        mfma_f64_16x16x4(af[kk], bop, cy);
                        mfma_f64_16x16x4(aw[kk], bop, cw);
                    }
                }
                // the elementwise terms + G (for Y) and + F - Rtop G (for W) of this tile: the row block of F and G is
                // already in registers in A-operand layout, so they are added as four more k-steps against an identity
                // B operand instead of being re-read from global memory in accumulator layout (uncoalesced)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int kk = 4 * tj + q4;                      // k = 4 kk + lk runs over the columns of tile tj
                    const double idb = (4 * q4 + lk == lr) ? 1.0 : 0.0;
                    double gk = 0.0, fk = 0.0;
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2)
                        if (k2 == kk) { fk = af[k2]; gk = (SIGNED ? sg : 1.0) * (aw[k2] + rt * af[k2]); }
                    mfma_f64_16x16x4(gk, idb, cy);
                    mfma_f64_16x16x4((SIGNED ? sg * fk : fk) - rt * gk, idb, cw);
                }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { ys[o][q][reg] = cy[reg]; ws[o][q][reg] = cw[reg]; }
        }
    }
    block_sync();  // every wavefront has read Q
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        const int cs = (RTc::NW >= 4) ? (wave >> 2) : 0;
#pragma unroll
        for (int q = 0; q < MAXTJ; ++q) {
            const int tj = cs + q * RTc::CS;
            if (ti < RT && tj < RT)
                tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                    Yout[col * LD + row] = ys[o][q][reg];
                    Wout[col * LD + row] = ws[o][q][reg];
                });
        }
    }
    block_sync();
}

// ---- the two row-block passes for 64 < N <= 128 (global-workspace kernels): same algorithm, 32 k-groups per row,
// up to eight row tiles, one row tile per wavefront at a time (its A operands in registers), operands from wherever
// the matrices live (all pointers are generic).
template <int NT>
SMRT_DEV void r1_mfma_big(const double* F, const double* G, double* Rt, double* Wk, double* cvec, const double* svec,
                          double Bl, int N, int LD) {
    constexpr int NW = NT / SMRT_LANES;
    constexpr int RPW = (NW >= 8) ? 1 : (8 + NW - 1) / NW;
    constexpr int CS = (NW > 8) ? NW / 8 : 1;
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
    for (int o = 0; o < RPW; ++o) {
        const int ti = (NW >= 8) ? (wave & 7) : (wave + o * NW);
        const int cs = (NW >= 8) ? (wave >> 3) : 0;
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        double a[32];
        double rs = 0.0;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double x = Rt[kc * LD + ic];
            a[kk] = (ti < RT && i < N && k < N) ? x : 0.0;
            rs += a[kk];
        }
        rs += shfl_xor(rs, 16);
        rs += shfl_xor(rs, 32);
        if (cs == 0 && ti < RT && lk == 0 && i < N) cvec[i] = rs * Bl - Bl + svec[i];
        block_sync();  // column-split wavefronts share a row tile: everybody has its A operands before anybody writes
        if (ti < RT) {
            for (int tj = cs; tj < RT; tj += CS) {
                double c1[4] = {0.0, 0.0, 0.0, 0.0}, c2[4] = {0.0, 0.0, 0.0, 0.0};
                const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) {
                    if (4 * kk < N) {
                        const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
                        const bool in = (j < N && k < N);
                        const double gv = G[jc * LD + kc], fv = F[jc * LD + kc];
                        mfma_f64_16x16x4(a[kk], in ? gv : 0.0, c1);
                        mfma_f64_16x16x4(a[kk], in ? fv : 0.0, c2);
                    }
                }
                tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                    Wk[col * LD + row] = F[col * LD + row] - c1[reg];
                    Rt[col * LD + row] = c2[reg] - G[col * LD + row];
                });
            }
        }
    }
    block_sync();
}

template <int NT, bool SIGNED>
SMRT_DEV void r45_mfma_big(double* F, const double* G, const double* Q, double* Wk, const double* Rtop, const double* tq,
                           double* upb, double* gvec, double Bl, int N, int LD, const double* dsg) {
    constexpr int NW = NT / SMRT_LANES;
    constexpr int RPW = (NW >= 8) ? 1 : (8 + NW - 1) / NW;
    constexpr int CS = (NW > 8) ? NW / 8 : 1;
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
    for (int o = 0; o < RPW; ++o) {
        const int ti = (NW >= 8) ? (wave & 7) : (wave + o * NW);
        const int cs = (NW >= 8) ? (wave >> 3) : 0;
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        const double rt = Rtop[ic];
        const double sg = SIGNED ? dsg[ic] : 1.0;
        double af[32], aw[32];
        double vy = 0.0, vg = 0.0;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double fv = F[kc * LD + ic], gv = G[kc * LD + ic], tk = tq[kc];
            const bool in = (ti < RT && i < N && k < N);
            af[kk] = in ? fv : 0.0;
            aw[kk] = in ? (SIGNED ? sg * gv : gv) - rt * fv : 0.0;
            vy += af[kk] * tk;
            vg += aw[kk] * tk;
        }
        vy += shfl_xor(vy, 16); vy += shfl_xor(vy, 32);
        vg += shfl_xor(vg, 16); vg += shfl_xor(vg, 32);
        if (cs == 0 && ti < RT && lk == 0 && i < N) { upb[i] = vy + Bl; gvec[i] = vg + (1.0 - rt) * Bl; }
        block_sync();
        if (ti < RT) {
            for (int tj = cs; tj < RT; tj += CS) {
                double cy[4] = {0.0, 0.0, 0.0, 0.0}, cw[4] = {0.0, 0.0, 0.0, 0.0};
                const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) {
                    if (4 * kk < N) {
                        const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
                        const double qv = Q[jc * LD + kc];
                        const double bop = (j < N && k < N) ? qv : 0.0;
                        mfma_f64_16x16x4(af[kk], bop, cy);
                        mfma_f64_16x16x4(aw[kk], bop, cw);
                    }
                }
                tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                    const double fic = F[col * LD + row], gic = G[col * LD + row];
                    Wk[col * LD + row] = cy[reg] + gic;
                    F[col * LD + row] = cw[reg] + (SIGNED ? dsg[row] * fic : fic) - Rtop[row] * gic;
                });
            }
        }
    }
    block_sync();
}

// ---- the same two passes without the matrix core (N > 64: CH column chunks of 64 per lane) --------------------
// rows-per-wavefront register blocking
constexpr int RB = 2;

// Wk = F - Rt G ; Rt <- Rt F - G (row-wise in place) ; cvec = (Rt 1) B - B + svec
template <int NT, int CH>
SMRT_DEV void r1_rows(const double* F, const double* G, double* Rt, double* Wk, double* cvec, const double* svec,
                      double Bl, int N, int LD) {
    const int t = tid(), lane = t % SMRT_LANES, wave = t / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    for (int i0 = wave * RB; i0 < N; i0 += NW * RB) {
        double a1[RB][CH], a2[RB][CH], rsum[RB];
        for (int bb = 0; bb < RB; ++bb) { rsum[bb] = 0.0; for (int ch = 0; ch < CH; ++ch) { a1[bb][ch] = 0.0; a2[bb][ch] = 0.0; } }
        for (int k = 0; k < N; ++k) {
            double fk[CH], gk[CH];
            for (int ch = 0; ch < CH; ++ch) {
                const int c = ch * SMRT_LANES + lane;
                fk[ch] = (c < N) ? F[c * LD + k] : 0.0;
                gk[ch] = (c < N) ? G[c * LD + k] : 0.0;
            }
            for (int bb = 0; bb < RB; ++bb) {
                const int i = i0 + bb;
                const double r = (i < N) ? Rt[k * LD + i] : 0.0;
                rsum[bb] += r;
                for (int ch = 0; ch < CH; ++ch) { a1[bb][ch] += r * gk[ch]; a2[bb][ch] += r * fk[ch]; }
            }
        }
        wave_sync();  // every lane has read rows i0.. of Rt before they are overwritten
        for (int bb = 0; bb < RB; ++bb) {
            const int i = i0 + bb;
            if (i < N) {
                for (int ch = 0; ch < CH; ++ch) {
                    const int c = ch * SMRT_LANES + lane;
                    if (c < N) {
                        Wk[c * LD + i] = F[c * LD + i] - a1[bb][ch];
                        Rt[c * LD + i] = a2[bb][ch] - G[c * LD + i];
                    }
                }
                if (lane == 0) cvec[i] = rsum[bb] * Bl - Bl + svec[i];
            }
        }
    }
    block_sync();
}

// Y = F tQt + G -> Wk ; W = (D G - Rtop F) tQt + (D F - Rtop G) -> over F (row-wise in place; D = 1 unless SIGNED)
// upb = F tq + B ; g = (D G - Rtop F) tq + (1 - Rtop) B
template <int NT, int CH, bool SIGNED>
SMRT_DEV void r45_rows(double* F, const double* G, const double* Q, double* Wk, const double* Rtop, const double* tq,
                       double* upb, double* gvec, double Bl, int N, int LD, const double* dsg) {
    const int t = tid(), lane = t % SMRT_LANES, wave = t / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    for (int i0 = wave * RB; i0 < N; i0 += NW * RB) {
        double ay[RB][CH], aw[RB][CH], vy[RB], vg[RB];
        for (int bb = 0; bb < RB; ++bb) { vy[bb] = 0.0; vg[bb] = 0.0; for (int ch = 0; ch < CH; ++ch) { ay[bb][ch] = 0.0; aw[bb][ch] = 0.0; } }
        for (int k = 0; k < N; ++k) {
            double tk[CH];
            for (int ch = 0; ch < CH; ++ch) {
                const int c = ch * SMRT_LANES + lane;
                tk[ch] = (c < N) ? Q[c * LD + k] : 0.0;
            }
            const double tqk = tq[k];
            for (int bb = 0; bb < RB; ++bb) {
                const int i = i0 + bb;
                double fik = 0.0, wik = 0.0;
                if (i < N) {
                    fik = F[k * LD + i];
                    const double gik = G[k * LD + i];
                    wik = (SIGNED ? dsg[i] * gik : gik) - Rtop[i] * fik;
                }
                vy[bb] += fik * tqk; vg[bb] += wik * tqk;
                for (int ch = 0; ch < CH; ++ch) { ay[bb][ch] += fik * tk[ch]; aw[bb][ch] += wik * tk[ch]; }
            }
        }
        wave_sync();  // every lane has read rows i0.. of F before they are overwritten
        for (int bb = 0; bb < RB; ++bb) {
            const int i = i0 + bb;
            if (i < N) {
                const double rt = Rtop[i];
                for (int ch = 0; ch < CH; ++ch) {
                    const int c = ch * SMRT_LANES + lane;
                    if (c < N) {
                        const double fic = F[c * LD + i], gic = G[c * LD + i];
                        Wk[c * LD + i] = ay[bb][ch] + gic;
                        F[c * LD + i] = aw[bb][ch] + (SIGNED ? dsg[i] * fic : fic) - rt * gic;
                    }
                }
                if (lane == 0) { upb[i] = vy[bb] + Bl; gvec[i] = vg[bb] + (1.0 - rt) * Bl; }
            }
        }
    }
    block_sync();
}

// ---- blocked Cholesky of two SPD matrices side by side on the matrix core (N <= 64) --------------------------
// Right-looking with 16-column blocks: the 16x16 diagonal block is factorised (and its inverse formed) by one
// wavefront per matrix with lane = row and the row in registers; the panel below (L_IJ = A_IJ inv(L_JJ)^T) and the
// trailing update (A_IK -= L_IJ L_KJ^T) are MFMA tile GEMMs.  3 workgroup barriers per block column (12 for N = 64)
// instead of one per column, and the O(N^3) part runs on the matrix core.
template <int NT>
SMRT_DEV bool chol2_mfma(double* A0, double* A1, double* inv /* [2][256] */, int* fail, int N, int LD,
                         double* inv_out = nullptr /* [4][256]: inverses of the diagonal blocks of the first factor */) {
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    constexpr int NW = NT / SMRT_LANES;
    const int RT = (N + 15) >> 4;
    if (t == 0) *fail = 0;
    block_sync();
    for (int J = 0; J < RT; ++J) {
        const int b0 = J * 16;
        // (a) diagonal block: L_JJ and its inverse
        for (int mi = wave; mi < 2; mi += NW) {
            double* A = mi ? A1 : A0;
            double row[16];
            const int gi = b0 + lr;                      // lanes 16..63 mirror lanes 0..15 (harmless duplicates)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int gj = b0 + j;
                const int gic = gi < N ? gi : N - 1, gjc = gj < N ? gj : N - 1;
                const double v = A[gjc * LD + gic];
                row[j] = (gi < N && gj < N) ? v : ((lr == j) ? 1.0 : 0.0);   // identity padding
            }
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double akk = wave_bcast(row[k], k);
                if (!(akk > 0.0)) ok = false;
                const double rk = fast_rsqrt(ok ? akk : 1.0);
                const double lik = row[k] * rk;             // L[i][k] for i >= k (lane k: sqrt(akk))
                row[k] = lik;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j > k) { const double ljk = wave_bcast(lik, j); row[j] -= lik * ljk; }
            }
            if (!ok && lane == 0) *fail = 1;
            if (lane < 16) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int gj = b0 + j;
                    if (gi < N && gj < N && j <= lr) A[gj * LD + gi] = row[j];
                }
            }
            // inverse of L_JJ by forward substitution, lane = column of the inverse; L[i][k] = row[k] of lane i
            double x[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                double acc = (i == lr) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < i) { const double lik2 = wave_bcast(row[k], i); acc -= lik2 * ((k >= lr) ? x[k] : 0.0); }
                const double dii = wave_bcast(row[i], i);
                x[i] = (i >= lr) ? acc * fast_rcp(dii) : 0.0;
            }
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 16; ++i) inv[mi * 256 + lr * 16 + i] = x[i];   // (L^-1)[i][j = lr]
                if (mi == 0 && inv_out) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) inv_out[J * 256 + lr * 16 + i] = x[i];
                }
            }
        }
        block_sync();
        if (*fail) return false;  // uniform
        // (b) panel below the diagonal block: L_IJ = A_IJ inv(L_JJ)^T
        {
            const int nt_ = 2 * (RT - 1 - J);
            for (int tix = wave; tix < nt_; tix += NW) {
                const int mi = tix & 1, I = J + 1 + (tix >> 1);
                double* A = mi ? A1 : A0;
                double c[4] = {0.0, 0.0, 0.0, 0.0};
                const int i = I * 16 + lr, ic = i < N ? i : N - 1;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = 4 * kk + lk, gk = b0 + k, gkc = gk < N ? gk : N - 1;
                    const double av = A[gkc * LD + ic];
                    const double bv = inv[mi * 256 + k * 16 + lr];          // (L^-1)[lr][k] = invT[k][lr]
                    mfma_f64_16x16x4((i < N && gk < N) ? av : 0.0, bv, c);
                }
                tile_foreach(I, J, N, [&](int reg, int row_, int col) { A[col * LD + row_] = c[reg]; });
            }
        }
        block_sync();
        // (c) trailing update A_IK -= L_IJ L_KJ^T, I >= K > J
        {
            const int nb = RT - 1 - J;
            const int ntri = nb * (nb + 1) / 2;
            for (int tix = wave; tix < 2 * ntri; tix += NW) {
                const int mi = tix & 1;
                int q = tix >> 1, Ir = 0;
                while ((Ir + 1) * (Ir + 2) / 2 <= q) ++Ir;
                const int Kr = q - Ir * (Ir + 1) / 2;
                const int I = J + 1 + Ir, K = J + 1 + Kr;
                double* A = mi ? A1 : A0;
                double c[4] = {0.0, 0.0, 0.0, 0.0};
                const int i = I * 16 + lr, ic = i < N ? i : N - 1;
                const int j = K * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int gk = b0 + 4 * kk + lk, gkc = gk < N ? gk : N - 1;
                    const double av = A[gkc * LD + ic], bv = A[gkc * LD + jc];
                    mfma_f64_16x16x4((i < N && gk < N) ? av : 0.0, (j < N && gk < N) ? bv : 0.0, c);
                }
                tile_foreach(I, K, N, [&](int reg, int row_, int col) { A[col * LD + row_] -= c[reg]; });
            }
        }
        block_sync();
    }
    return true;
}

// ---- blocked triangular solve on the matrix core: Bm <- Lp^-T Bm  (N <= 64) ---------------------------------
// The (up to four) 16x16 diagonal blocks of Lp are inverted once (one wavefront per block, lane = column of the
// inverse, forward substitution); then, block row by block row from the bottom,
//   X_I = inv(L_II)^T (B_I - sum_{J>I} L_JI^T X_J)
// is two MFMA GEMMs per 16x16 tile -- the accumulator layout of the first is exactly the B-operand layout of the
// second (c[reg] = R[lk + 4 reg][lr] = B[k = 4 kk + lk][j = lr] for kk = reg), so nothing moves between them.
template <int NT>
SMRT_DEV void lt_solve_mfma(const double* Lp, double* Bm, double* inv /* [4][16*16] */, int N, int LD,
                            bool have_inv = false /* inv already holds the block inverses (from chol2_mfma) */) {
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    constexpr int NW = NT / SMRT_LANES;
    const int RT = (N + 15) >> 4;
    // inverse of the diagonal blocks: inv[I][j*16 + i] = (L_II^-1)[i][j]  (identity padding beyond N).
    // lane (mod 16) = row i of the block with the row in registers; entries of other rows come by wave_bcast
    // (loading the block through broadcast LDS reads made the compiler hoist all 136 loads into registers).
    for (int I = wave; I < RT && !have_inv; I += NW) {
        const int b0 = I * 16, gi = b0 + lr;
        double row[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int gk = b0 + k;
            const int gic = gi < N ? gi : N - 1, gkc = gk < N ? gk : N - 1;
            const double v = Lp[gkc * LD + gic];
            row[k] = (gi < N && gk < N && k <= lr) ? v : ((k == lr) ? 1.0 : 0.0);
        }
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double acc = (i == lr) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < i) { const double lik = wave_bcast(row[k], i); acc -= lik * ((k >= lr) ? x[k] : 0.0); }
            const double dii = wave_bcast(row[i], i);
            x[i] = (i >= lr) ? acc * fast_rcp(dii) : 0.0;
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) inv[I * 256 + lr * 16 + i] = x[i];
        }
    }
    if (!have_inv) block_sync();
    for (int I = RT - 1; I >= 0; --I) {
        for (int tj = wave; tj < RT; tj += NW) {
            double c[4] = {0.0, 0.0, 0.0, 0.0};
            const int i = I * 16 + lr, j = tj * 16 + lr;
            const int ic = i < N ? i : N - 1, jc = j < N ? j : N - 1;
            // acc = sum_{k in later blocks} L[k][i] X[k][j]
            for (int k0 = (I + 1) * 16; k0 < N; k0 += 4) {
                const int k = k0 + lk, kc = k < N ? k : N - 1;
                const double av = Lp[ic * LD + kc], bv = Bm[jc * LD + kc];
                mfma_f64_16x16x4((i < N && k < N) ? av : 0.0, (j < N && k < N) ? bv : 0.0, c);
            }
            // R = B_I - acc in accumulator layout
            double r[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = I * 16 + lk + 4 * reg;
                const int rowc = row < N ? row : N - 1;
                const double bv = Bm[jc * LD + rowc];
                r[reg] = ((row < N && j < N) ? bv : 0.0) - c[reg];
            }
            // X = inv(L_II)^T R : A[i][k] = inv[k][i]
            double x[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = 4 * kk + lk;                          // row of inv(L_II)
                const double av = inv[I * 256 + lr * 16 + k];       // (L_II^-1)[k][lr]
                mfma_f64_16x16x4(av, r[kk], x);
            }
            tile_foreach(I, tj, N, [&](int reg, int row, int col) { Bm[col * LD + row] = x[reg]; });
        }
        block_sync();
    }
}

// ---- 16-wide blocked Gauss-Jordan (N <= 128): ONE workgroup barrier per 16 columns ---------------------------------
// Gauss-Jordan with implicit partial pivoting: per block one wavefront factorises the panel (lane = row, arg-max over
// the rows not used yet by DPP on a 32-bit key; rows are never swapped, the permutation is undone once at the end) and
// tracks the columns u_j = T[:, p_j] - e_pj of the accumulated row transformation T, so that the whole block update is
// C <- C + U R_P with the ORIGINAL pivot rows R_P (no inverse to form); the pivot rows come out normalised, so after
// the last block row perm[k] of B is row k of the solution.  (An earlier version used 4-column blocks with a side
// buffer for U and a copy of the pivot rows, two barriers per block.)  Design points of this one:
//   * block width 16 = one MFMA tile column = four chained v_mfma_f64_16x16x4 per tile (the C tile is loaded and
//     stored once per 16 eliminated columns instead of once per 4);
//   * the multipliers u_j are written into the panel's own, now dead, columns of A -- no side buffer;
//   * the pivot rows of the running block are NOT touched by the tile updates (stores to them are masked), so they
//     can be read in place as the B operand by every wavefront; their own new values R_P + U_P R_P are computed as
//     one extra "virtual" tile per column tile, kept in registers across the block barrier and stored after it.
// Every wavefront owns fixed absolute column tiles of [A | B] for the whole solve, so the only cross-wavefront
// traffic per block is the panel (u columns, permutation, row states), published by the one barrier.  The panel of
// block k+1 is factorised by the owner of that column tile right after it has updated the tile (look-ahead).
// RPLN = rows per lane: 1 for N <= 64 (lane = row), 2 for N <= 128 (lane holds rows lane and lane + 64).
template <bool TR, int RPLN>
SMRT_DEV_NOINLINE bool gj_panel16(double* A, int N, int LD, int k, int lane, int* perm, int* rowblk) {
    // x[r][s] holds panel column s of row (lane + 64 r) until the column has been a pivot column, its multiplier u_s
    // afterwards: both kinds of slot receive the same update x[s] += u_j * x[s][pivot row], so a step treats all
    // slots but the pivot one alike.  The loop is unrolled by four only, with the slots rotated by four after every
    // group (the pivot slot index stays a compile-time constant): a fully unrolled panel is ~18 KB of straight-line
    // code that is executed once per call and does not live in the instruction cache next to the rest of the kernel.
    const int k0 = 16 * k;
    const int nbk = (N - k0 < 16) ? N - k0 : 16;
    double x[RPLN][16];
    bool used[RPLN], mine[RPLN];
#pragma unroll
    for (int r = 0; r < RPLN; ++r) {
        const int row = lane + 64 * r;
        const int rc = row < N ? row : N - 1;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int cc = (k0 + j < N) ? k0 + j : N - 1;
            const double v = at<TR>(A, rc, cc, LD);
            x[r][j] = (row < N && j < nbk) ? v : 0.0;
        }
        used[r] = (row < N) ? (rowblk[rc] >= 0) : true;
#ifdef SMRT_GJ_DIAG_PIVOT
        // numerical experiment: pivots only from the 16 rows of the diagonal block -- what a panel
        // built from a 16 x 16 inverse and MFMA products would do
        if (row < k0 || row >= k0 + 16) used[r] = true;
#endif
        mine[r] = false;
    }
    bool ok = true;
    int pj_store = 0;
    int grp = 0;
    for (; grp * 4 < nbk; ++grp) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = grp * 4 + q;
            if (j < nbk) {   // uniform
                // arg-max over the unused rows: float magnitude bits with (255 - row) in the 8 low mantissa bits
                unsigned key = 0u;
#pragma unroll
                for (int r = 0; r < RPLN; ++r) {
                    if (!used[r]) {
                        const float xr = (float)fabs(x[r][q]);
                        unsigned kr;
                        memcpy(&kr, &xr, 4);
                        kr = (kr & ~0xFFu) | (unsigned)(255 - (lane + 64 * r));
                        key = kr > key ? kr : key;
                    }
                }
                key = wave_max_u32(key);
                if (key < 256u) ok = false;
                const int p = ok ? 255 - (int)(key & 0xFFu) : 0;
                if (lane == j) pj_store = p;
                const int pl = p & 63, ps = p >> 6;   // lane and slot of the pivot row (uniform)
                double pvq = x[0][q];
                if (RPLN > 1) pvq = ps ? x[RPLN - 1][q] : x[0][q];
                const double rpv = fast_rcp(ok ? wave_bcast(pvq, pl) : 1.0);
                double pr[16];
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) {
                    if (s2 != q) {
                        double src = x[0][s2];
                        if (RPLN > 1) src = ps ? x[RPLN - 1][s2] : x[0][s2];
                        pr[s2] = wave_bcast(src, pl);
                    }
                }
#if !defined(SMRT_HOST_EMU)
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int r = 0; r < RPLN; ++r) {
                    const bool isp = (lane == pl) && (r == ps);
                    if (isp) { used[r] = true; mine[r] = true; }
                    // the pivot row itself is scaled by 1/pivot: a - (1 - 1/pv) a = a / pv, i.e. the same update as
                    // every other row with the multiplier 1 - 1/pv
                    const double uj = isp ? rpv - 1.0 : -(x[r][q] * rpv);
#pragma unroll
                    for (int s2 = 0; s2 < 16; ++s2)
                        if (s2 != q) x[r][s2] = __builtin_fma(uj, pr[s2], x[r][s2]);
                    x[r][q] = uj;
                }
            }
        }
        // rotate the slots left by four: slot s now holds what slot s + 4 held
#pragma unroll
        for (int r = 0; r < RPLN; ++r) {
            const double t0 = x[r][0], t1 = x[r][1], t2 = x[r][2], t3 = x[r][3];
#pragma unroll
            for (int s2 = 0; s2 < 12; ++s2) x[r][s2] = x[r][s2 + 4];
            x[r][12] = t0; x[r][13] = t1; x[r][14] = t2; x[r][15] = t3;
        }
    }
    // after grp rotations slot s holds column (s + 4 grp) mod 16
#pragma unroll
    for (int r = 0; r < RPLN; ++r) {
        const int row = lane + 64 * r;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const int j = (s2 + 4 * grp) & 15;
            if (row < N && j < nbk) at<TR>(A, row, k0 + j, LD) = ok ? x[r][s2] : 0.0;
        }
        if (mine[r] && ok) rowblk[row] = k;
    }
    if (lane < nbk) perm[k0 + lane] = pj_store;
    return ok;
}

// result_in_A: leave the solution in A (one pass and one barrier less than copying it back over Bm), optionally scaled
// X[k][c] * rs[k] * cs[c] on the way (the t Q t scaling of the recursion).
template <int NT, bool TR>
SMRT_DEV bool gj_solve_b16(double* A, double* Bm, double* v, const Lds& s, int N, int LD, bool result_in_A = false,
                           const double* rs = nullptr, const double* cs = nullptr) {
    const int t = tid();
    const int lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    const int NMX = s.gj_nmax;
    int* perm = (int*)s.gj;                     // [NMX + 16] pivot row of every column
    int* rowblk = perm + NMX + 16;              // [NMX] block in which the row was a pivot row, -1 before
    int* fail = rowblk + NMX;
    const bool has_v = (v != nullptr);
    const int RT = (N + 15) >> 4;
    const int lr = lane & 15, lk = lane >> 4;
    for (int r = t; r < NMX; r += NT) rowblk[r] = -1;
    if (t == 0) *fail = 0;
    block_sync();
#ifdef SMRT_STAGE_TIMING
    long long tg0 = cycle_counter();
#define SMRT_GSUB(k) do { const long long n_ = cycle_counter(); if (t == 0 && s.sub_acc) s.sub_acc[k] += (double)(n_ - tg0); tg0 = n_; } while (0)
#else
#define SMRT_GSUB(k) do {} while (0)
#endif
    auto panel = [&](int kb) { return (N > 64) ? gj_panel16<TR, 2>(A, N, LD, kb, lane, perm, rowblk) : gj_panel16<TR, 1>(A, N, LD, kb, lane, perm, rowblk); };
    if (wave == 0) { if (!panel(0) && lane == 0) *fail = 1; }
    SMRT_GSUB(0);
    block_sync();
    if (*fail) return false;  // uniform

    for (int k = 0; k < RT; ++k) {
        const int k0 = 16 * k;
        const int nbk = (N - k0 < 16) ? N - k0 : 16;
        // one absolute column tile g of [A | B]: all row tiles (pivot rows masked) + the virtual pivot-row tile -> pvt
        auto do_tile = [&](int g, double (&pvt)[4]) {
            double* Mat = (g < RT) ? A : Bm;
            const int col = ((g < RT) ? g : g - RT) * 16 + lr;
            const bool cin = col < N;
            const int colc = cin ? col : 0;
            double bop[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int j = 4 * kk + lk;
                const int pr = (j < nbk) ? perm[k0 + j] : 0;
                const double x = at<TR>(Mat, pr, colc, LD);
                bop[kk] = (cin && j < nbk) ? x : 0.0;
            }
            for (int ti = 0; ti < RT; ++ti) {
                double c[4];
                bool keep[4];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = ti * 16 + lk + 4 * reg;
                    const int rowc = row < N ? row : 0;
                    const double x = at<TR>(Mat, rowc, colc, LD);
                    keep[reg] = cin && row < N && rowblk[rowc] != k;
                    c[reg] = keep[reg] ? x : 0.0;
                }
                const int arow = ti * 16 + lr;
                const int arowc = arow < N ? arow : 0;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int j = 4 * kk + lk;
                    const int jc = (j < nbk) ? j : 0;
                    const double x = at<TR>(A, arowc, k0 + jc, LD);
                    mfma_f64_16x16x4((arow < N && j < nbk) ? x : 0.0, bop[kk], c);
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = ti * 16 + lk + 4 * reg;
                    if (keep[reg]) at<TR>(Mat, row, col, LD) = c[reg];
                }
            }
            // new pivot rows: R_P + U_P R_P with U_P[j][i] = u_i[p_j]
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = lk + 4 * reg;
                const int pr = (j < nbk) ? perm[k0 + j] : 0;
                const double x = at<TR>(Mat, pr, colc, LD);
                pvt[reg] = (cin && j < nbk) ? x : 0.0;
            }
            const int prl = (lr < nbk) ? perm[k0 + lr] : 0;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int j = 4 * kk + lk;
                const int jc = (j < nbk) ? j : 0;
                const double x = at<TR>(A, prl, k0 + jc, LD);
                mfma_f64_16x16x4((lr < nbk && j < nbk) ? x : 0.0, bop[kk], pvt);
            }
        };
        auto store_pivot_rows = [&](int g, const double (&pvt)[4]) {
            double* Mat = (g < RT) ? A : Bm;
            const int col = ((g < RT) ? g : g - RT) * 16 + lr;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = lk + 4 * reg;
                if (col < N && j < nbk) at<TR>(Mat, perm[k0 + j], col, LD) = pvt[reg];
            }
        };

        // Work distribution of block k: the owner of the next panel (wavefront gnext mod NW) takes only that column
        // tile and then factorises the panel; the other live column tiles ([A | B] minus the dead A tiles) go round
        // robin to the remaining wavefronts.  A column tile is read and written by exactly one wavefront per block
        // (pivot rows included: only the tile's own pivot-row entries serve as its B operand), so the new pivot rows
        // are stored right away and the block barrier is the only synchronisation.
        const int gnext = k + 1;
        const bool has_next = gnext < RT;
        const int owner = has_next ? (gnext % NW) : -1;
        if (has_next && wave == owner) {
            double tmp[4];
            do_tile(gnext, tmp);
            wave_sync_lds();
            store_pivot_rows(gnext, tmp);
            wave_sync_lds();
            if (!panel(k + 1) && lane == 0) *fail = 1;
        }
        const bool worker = (NW == 1) || !has_next || wave != owner;
        const int nworkers = (NW == 1 || !has_next) ? NW : NW - 1;
        const int widx = (NW == 1 || !has_next) ? wave : (wave - owner - 1 + NW) % NW;
        if (worker) {
            int idx = 0;
            for (int g = (has_next ? gnext + 1 : RT); g < 2 * RT; ++g, ++idx) {
                if (idx % nworkers != widx) continue;
                double tmp[4];
                do_tile(g, tmp);
                wave_sync_lds();
                store_pivot_rows(g, tmp);
                wave_sync_lds();
            }
            if (has_v && widx == 0) {  // extra right-hand side: same transformation, rows lane and lane + 64
                double acc[2] = {0.0, 0.0};
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    const int row = lane + 64 * r2;
                    if (row < N) {
                        acc[r2] = v[row];
                        for (int j = 0; j < nbk; ++j) acc[r2] += at<TR>(A, row, k0 + j, LD) * v[perm[k0 + j]];
                    }
                }
                wave_sync_lds();
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
                    if (lane + 64 * r2 < N) v[lane + 64 * r2] = acc[r2];
            }
        }
        block_sync();
        if (*fail) return false;  // uniform
    }
    block_sync();
    SMRT_GSUB(1);
    // ---- undo the implicit row permutation: row perm[k] of B is row k of the solution (A is free scratch now)
    if (rs) for_2d<NT>(N, N, [&](int k, int c) { at<TR>(A, k, c, LD) = at<TR>(Bm, perm[k], c, LD) * (rs[k] * cs[c]); });
    else for_2d<NT>(N, N, [&](int k, int c) { at<TR>(A, k, c, LD) = at<TR>(Bm, perm[k], c, LD); });
    double vk[2] = {0.0, 0.0};   // N <= 128 <= 2 NT
    if (has_v) {
        if (t < N) vk[0] = v[perm[t]];
        if (t + NT < N) vk[1] = v[perm[t + NT]];
    }
    block_sync();
    if (!result_in_A) for_2d<NT>(N, N, [&](int k, int c) { at<TR>(Bm, k, c, LD) = at<TR>(A, k, c, LD); });
    if (has_v) {
        if (t < N) v[t] = vk[0];
        if (t + NT < N) v[t + NT] = vk[1];
    }
    block_sync();
    SMRT_GSUB(2);
    return true;
}

// the Gauss-Jordan entry point of the drivers (solution copied back over Bm)
template <int NT, bool TR>
SMRT_DEV bool gj_solve(double* A, double* Bm, double* v, const Lds& s, int N, int LD) {
    return gj_solve_b16<NT, TR>(A, Bm, v, s, N, LD);
}

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
        const int N = stg.n[item];   // < 0: the diagonalisation of this layer failed (see first_failed_layer)
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
SMRT_DEV int pair_setup(const DevBatch& b, const Lds& s, double frequency, int L, const double* thickness,
                        const double* fracvol, const double* temperature, const double* mp1, const double* mp2) {
    const int t = tid();
    const int nmax = b.n_max_stream;
    for (int l = t; l < L; l += NT) {
        cplx ee; double ks, ka, pa, pb; int bad = 0;
        layer_em(b, frequency, fracvol[l], temperature[l], mp1[l], mp2[l], &ee, &ks, &ka, &pa, &pb, &bad);
        s.eps_re[l] = ee.re; s.eps_im[l] = ee.im; s.ks[l] = ks; s.ka[l] = ka; s.pa[l] = pa; s.pb[l] = pb;
        s.thick[l] = thickness[l];
        s.BT[l] = b.rayleigh_jeans ? temperature[l] : planck_radiance(frequency, temperature[l]);
        if (bad || !(ks >= 0.0)) lds_max(&s.ints[0], ST_INPUT);
    }
    for (int j = t; j < nmax; j += NT) {
        const double m = b.gl_mu[j];
        s.gmu[j] = m; s.gsin[j] = sqrt(1.0 - m * m);
    }
    block_sync();
    if (s.ints[0] != ST_OK) return s.ints[0];
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
                                   MODE == 1 ? 1 : (MODE == 3 ? 2 : 0),
                                   (gmem_mat != nullptr && MODE != 1) ? (MODE == 2 && b.jac_in_lds ? 2 : b.jac_in_lds) : 0);
    Lds s = carve(lds_base, gmem_mat == nullptr ? lds_base : gmem_mat, plan);
    // matrix-core variants of the dense steps: always on the LDS path; on the global-workspace path for N <= 128 when
    // the LDS Jacobi buffer exists (it doubles as the scratch of the blocked Cholesky / triangular solve)
    // (the prep half only needs the 512-double Cholesky scratch, which its slim plan has)
    const bool dense_mfma = (CH == 1) || (CH == 2 && (plan.o_jac >= 0 || MODE == 1));
    double* dense_scratch = (CH == 1 || MODE == 1) ? s.gj : lds_base + (plan.o_jac >= 0 ? plan.o_jac : 0);
#ifdef SMRT_STAGE_TIMING
    double sub_acc_store[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    s.sub_acc = sub_acc_store;
#endif
    const int LD = plan.LD;
    const int nmax = b.n_max_stream;
    const int out_stride = P * b.n_theta;

    const long long gp = b.pair_begin + p;
    const int fi = (int)(gp / b.S), si = (int)(gp % b.S);
    const double frequency = b.frequency[fi];
    const int L = b.n_layers[si];
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
        const int st = pair_setup<NT>(b, s, frequency, L, thickness, fracvol, temperature, mp1, mp2);
        if (st != ST_OK) { fail_pair<NT>(b, p, st, out_stride); return; }
    }
    const int n_air = s.ints[5];

    if (MODE != 1 && b.want_layer_out) {
        double* lo = b.layer_out + p * (long long)b.Lmax * 5;
        for (int l = t; l < b.Lmax; l += NT) {
            const bool in = l < L;
            lo[l * 5 + 0] = in ? s.eps_re[l] : 0.0; lo[l * 5 + 1] = in ? s.eps_im[l] : 0.0;
            lo[l * 5 + 2] = in ? s.ks[l] : 0.0; lo[l * 5 + 3] = in ? s.ka[l] : 0.0;
            lo[l * 5 + 4] = in ? s.nl[l] : 0.0;
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
    // ---- bottom-up over the layers -------------------------------------------------------------------------
    for (int l = Lk - 1; l >= 0; --l) {
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
                    double Rv, Rh;
                    fresnel_RvRh(el, cmk(s.eps_re[l + 1], s.eps_im[l + 1]), sqrt(1.0 - rs * rs), &Rv, &Rh);
                    Rs = (r & 1) ? Rh : Rv;
                } else if (b.sub_kind != SUB_NONE) {
                    const long long gpi = b.pair_begin + p;
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
            double Rv, Rh;
            const cplx eup = (l > 0) ? cmk(s.eps_re[l - 1], s.eps_im[l - 1]) : cmk(1.0, 0.0);
            fresnel_RvRh(el, eup, s.mu[j], &Rv, &Rh);
            s.Rtop[2 * j] = Rv; s.Rtop[2 * j + 1] = Rh;
            s.Ttop[2 * j] = 1.0 - Rv; s.Ttop[2 * j + 1] = 1.0 - Rh;
        }
        if (MODE != 1 && l > 0)
            for (int j = t; j < nu; j += NT) {
                double Rv, Rh;
                fresnel_RvRh(cmk(s.eps_re[l - 1], s.eps_im[l - 1]), el, s.muu[j], &Rv, &Rh);
                s.Rbu[2 * j] = Rv; s.Rbu[2 * j + 1] = Rh;
                s.Tbu[2 * j] = 1.0 - Rv; s.Tbu[2 * j + 1] = 1.0 - Rh;
            }

        SMRT_STAGE(SG_ASSEMBLE);
        if (MODE < 2) {
        // -- phase matrix, azimuth mode 0: S+ = P(mu,+mu') + P(mu,-mu') -> M0, S- = P(+) - P(-) -> M1
        //    (lower triangle by stream blocks; the matrices are symmetric)
        {
            const int T = n * (n + 1) / 2;
            const double pa = s.pa[l], pb = s.pb[l];
            const double fv = fracvol[l], q1 = mp1[l], q2 = mp2[l];
            for (int idx = t; idx < T; idx += NT) {
                int i = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
                while ((i + 1) * (i + 2) / 2 <= idx) ++i;
                while (i * (i + 1) / 2 > idx) --i;
                const int j = idx - i * (i + 1) / 2;
                const double mi = s.mu[i], mj = s.mu[j];
                double pvv_p, pvh_p, phv_p, phh_p, pvv_m, pvh_m, phv_m, phh_m;
                if (b.emmodel != EM_IBA) {  // closed form, rayleigh.py:70-76; even in mu'
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
                        ct_p = ct_p > 1.0 ? 1.0 : (ct_p < -1.0 ? -1.0 : ct_p);
                        ct_m = ct_m > 1.0 ? 1.0 : (ct_m < -1.0 ? -1.0 : ct_m);
                        double Cp, Cm;
                        if (b.micro == MS_EXP) {
                            const double dp = 1.0 + pb * (1.0 - ct_p), dm = 1.0 + pb * (1.0 - ct_m);
                            Cp = pa * fast_rcp(dp * dp); Cm = pa * fast_rcp(dm * dm);   // 1 / (dp dm)^2 without the IEEE division
                        } else {
                            Cp = pa * ft_corr(MS_SHS, pb * (1.0 - ct_p), fv, q1, q2);
                            Cm = pa * ft_corr(MS_SHS, pb * (1.0 - ct_m), fv, q1, q2);
                        }
                        Cp *= wk; Cm *= wk;
                        const double fvv_p = c * mm + sisj, fvv_m = -c * mm + sisj;
                        pvv_p += fvv_p * fvv_p * Cp; pvv_m += fvv_m * fvv_m * Cm;
                        pvh_p += s2 * a2 * Cp; pvh_m += s2 * a2 * Cm;
                        phv_p += s2 * b2 * Cp; phv_m += s2 * b2 * Cm;
                        phh_p += c * c * Cp; phh_m += c * c * Cm;
                    }
                }
                const int r0 = 2 * i, c0 = 2 * j;
                s.M0[c0 * LD + r0] = pvv_p + pvv_m;             s.M1[c0 * LD + r0] = pvv_p - pvv_m;
                s.M0[(c0 + 1) * LD + r0] = pvh_p + pvh_m;       s.M1[(c0 + 1) * LD + r0] = pvh_p - pvh_m;
                s.M0[c0 * LD + r0 + 1] = phv_p + phv_m;         s.M1[c0 * LD + r0 + 1] = phv_p - phv_m;
                s.M0[(c0 + 1) * LD + r0 + 1] = phh_p + phh_m;   s.M1[(c0 + 1) * LD + r0 + 1] = phh_p - phh_m;
            }
        }
        block_sync();
        // -- energy-conserving renormalisation (dort.py:782-819): norm_r = ks / (c sum_c S+[r,c] w_c), c = 1/2
        for (int r = t; r < N; r += NT) {
            double rs = 0.0;
            for (int c = 0; c <= r; ++c) rs += s.M0[c * LD + r] * s.wrow[c];
            for (int c = r + 1; c < N; ++c) rs += s.M0[r * LD + c] * s.wrow[c];
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
        // -- X+- = M^-1/2 T (ke I - c N S+- W) T^-1 M^-1/2, symmetric positive definite (lower triangles)
        for_2d<NT>(N, N, [&](int r, int c) {
            if (r >= c) {
                const double uu = 0.5 * s.u[r] * s.u[c];
                const double dg = (r == c) ? ke / s.mrow[r] : 0.0;
                s.M0[c * LD + r] = dg - uu * s.M0[c * LD + r];
                s.M1[c * LD + r] = dg - uu * s.M1[c * LD + r];
            }
        });
        block_sync();
        SMRT_STAGE(SG_CHOL);
        if (!(dense_mfma ? chol2_mfma<NT>(s.M0, s.M1, dense_scratch, &s.ints[2], N, LD,
                                       (MODE == 1 && CH == 1) ? stg->Linv + (p * (long long)b.Lmax + l) * 1024 : nullptr)
                      : chol2<NT>(s.M0, s.M1, N, LD))) {
            if (MODE == 1) { layer_failed(l, ST_ALBEDO); continue; }
            fail_pair<NT>(b, p, ST_ALBEDO, out_stride); return;
        }
        SMRT_STAGE(SG_BTL);
        if (MODE == 1) {  // B = L+^T L- straight from the accumulators into the staging area
            lt_times_l_mfma<NT>(s.M0, s.M1, stg->B + (p * (long long)b.Lmax + l) * stg->mat_stride, N, LD,
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
            for_2d<NT>(N, N, [&](int r, int c) { gL[c * LD + r] = s.M0[c * LD + r]; });
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
            lt_solve_mfma<NT>(s.M3, s.M0, stg->Linv + item * 1024, N, LD, true);                // Ep' = L+^-T B'
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
        } else if (CH == 2 && dense_mfma) {
            r1_mfma_big<NT>(F, G, Rt, Wk, s.cvec, s.svec, Bl, N, LD);
        } else {
            r1_rows<NT, CH>(F, G, Rt, Wk, s.cvec, s.svec, Bl, N, LD);
        }
        SMRT_DUMP("M1", Wk, N); SMRT_DUMP("RHS", Rt, N);
        SMRT_STAGE(SG_LU1);
        // -- x+ = Q t x- + q : solve (F - Rt G) [Q | q] = [Rt F - G | c]
        if (MODE == 3) {  // the solution t Q t stays in slot X (one pass over the matrix instead of three)
            if (!gj_solve_b16<NT, false>(Wk, Rt, s.cvec, s, N, LD, true, s.t, s.t)) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
        } else
        if (!(CH <= 2 ? gj_solve<NT, false>(Wk, Rt, s.cvec, s, N, LD) : lu_solve<NT, false>(Wk, Rt, s.cvec, s.sigma, N, LD))) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
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
        } else if (CH == 2 && dense_mfma) {
            r45_mfma_big<NT, false>(F, G, Q, Wk, s.Rtop, s.tq, s.upb, s.g, Bl, N, LD, nullptr);
        } else {
            r45_rows<NT, CH, false>(F, G, Q, Wk, s.Rtop, s.tq, s.upb, s.g, Bl, N, LD, nullptr);
        }
        SMRT_DUMP("Y", Wk, N); SMRT_DUMP("W", F, N);
        SMRT_STAGE(SG_LU2);
        // -- K = Y W^-1  (solve W^T K^T = Y^T on the transposed view; K lands in Wk in normal storage)
        if (MODE == 3) {  // A = W (slot X), B = Y (slot R); K is left in slot X
            if (!gj_solve_b16<NT, true>(Wk, Rt, nullptr, s, N, LD, true)) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
        } else
        if (!(CH <= 2 ? gj_solve<NT, true>(F, Wk, nullptr, s, N, LD) : lu_solve<NT, true>(F, Wk, nullptr, s.sigma, N, LD))) { fail_pair<NT>(b, p, ST_SINGULAR, out_stride); return; }
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
        if (l > 0) {
            // reflection matrix and source seen from the bottom of layer l-1 (streams paired by index)
            const int nc = (N < Nu) ? N : Nu;
            for_2d<NT>(Nu, Nu, [&](int i, int j) {
                double v = (i == j) ? s.Rbu[i] : 0.0;
                if (i < nc && j < nc) v += s.Ttop[i] * K[j * LD + i] * s.Tbu[j];
                s.M3[j * LD + i] = v;
            });
            for (int i = t; i < Nu; i += NT) s.svec[i] = (i < nc) ? s.Ttop[i] * s.up[i] : 0.0;
            block_sync();
        }
    }

    if (MODE == 1) {
        if (t == 0) b.status[p] = ST_OK;
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
            if (atm && Idn != 0.0) {
                double acc = 0.0;
                for (int j = 0; j < n_air; ++j) {
                    double Rv, Rh;
                    fresnel_RvRh(cmk(1.0, 0.0), e0, s.outmu[j], &Rv, &Rh);
                    acc += K0[(2 * j) * LD + i] * (1.0 - Rv) + K0[(2 * j + 1) * LD + i] * (1.0 - Rh);
                }
                double Rv, Rh;
                fresnel_RvRh(cmk(1.0, 0.0), e0, s.outmu[i >> 1], &Rv, &Rh);
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
        for (int k = 0; k < 16; ++k) b.stage_out[p * 16 + k] = (k < SG_COUNT) ? stage_acc[k] : 0.0;
        b.stage_out[p * 16 + 12] = (double)n_sweeps;
        for (int k = 0; k < 3; ++k) b.stage_out[p * 16 + 13 + k] = sub_acc_store[k];
        b.stage_out[p * 16 + 0] = sub_acc_store[3]; b.stage_out[p * 16 + 1] = sub_acc_store[4]; b.stage_out[p * 16 + 2] = sub_acc_store[5];  // (overrides setup/assemble/cholesky slots in this debug build)
    }
#endif
}

// ---- zero-padded Jacobi step for the split pipeline ----------------------------------------------------------
// The matrix is padded with zero rows/columns up to CP = NB*m columns and RPL*GS rows, so no lane ever needs a
// validity test or a masked load (a zero column never rotates: g = 0).  Column norms are tracked in LDS (a rotation
// changes them by -/+ t g exactly) and refreshed once per sweep, so a step needs ONE dot product and ONE group sum.
template <int GS, int RPL>
SMRT_DEV void rotate_pair_padded(double* Bm, int LD, int p, int q, int sub, double* nrm, int* flag, double skip2,
                                 double exit2) {
    double x[RPL], y[RPL];
    double gg = 0.0, gg2 = 0.0;
    double* cp = Bm + p * LD;
    double* cq = Bm + q * LD;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int r = sub + i * GS;
        x[i] = cp[r];
        y[i] = cq[r];
        if (i & 1) gg2 += x[i] * y[i]; else gg += x[i] * y[i];
    }
    const double a = nrm[p], bb = nrm[q];
    gg = group_sum<GS>(gg + gg2);
    const double g2 = gg * gg, ab = a * bb;
    if (g2 > skip2 * ab) {
        const double dd = bb - a;
        const double hh = dd * dd + 4.0 * g2;
        const double h = hh * fast_rsqrt1(hh);
        const double tt = (dd >= 0.0 ? 2.0 : -2.0) * gg * fast_rcp1(fabs(dd) + h);
        const double c = fast_rsqrt(1.0 + tt * tt), sn = c * tt;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = sub + i * GS;
            cp[r] = c * x[i] - sn * y[i];
            cq[r] = sn * x[i] + c * y[i];
        }
        if (sub == 0) {
            nrm[p] = a - tt * gg;
            nrm[q] = bb + tt * gg;
            if (g2 > exit2 * ab) lds_or(flag, 1);
        }
    }
}

// Two-level ordering as jacobi_onesided, on a zero-padded LDS matrix (rows < RPL*GS <= LD, columns < NB*m).
// skip2 / exit2: squared-cosine thresholds below which a rotation is skipped / does not count against convergence.
// Passive brightness temperatures (1e-6 K of ~250 K) tolerate 1e-26 / 1e-15; the backscatter is a small difference
// of intensities (coherent part subtracted, azimuth modes cancelling in cross-pol), so active mode uses 1e-30 / 1e-22
// (5e-10 -> 2e-11 relative error on the fixtures, about a third of a sweep more).
template <int NT, int JW, int GS, int RPL>
SMRT_DEV bool jacobi_padded(double* Bm, int N, int LD, double* sigma, double* nrm, int* flag,
                            double skip2 = SMRT_JACOBI_SKIP_COS2, double exit2 = SMRT_JACOBI_EXIT_COS2) {
    const int t = tid();
    const int lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES;
    constexpr int NB = 2 * JW;
    constexpr int SLOTS = SMRT_LANES / GS;
    const int slot = lane / GS, sub = lane % GS;
    const int m = (N + NB - 1) / NB;
    const int CP = NB * m;
    const int me = m + (m & 1);
    constexpr int NG = NT / GS;
    const int grp = t / GS;
    bool converged = false;
    for (int sweep = 0; sweep < 40 && !converged; ++sweep) {
        // refresh the tracked squared column norms (also the first computation)
        for (int c0 = 0; c0 < CP; c0 += NG) {  // uniform trip count (the group sum is a wavefront operation)
            const int c = c0 + grp;
            const int cc = c < CP ? c : CP;
            double a = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) { const double xx = Bm[cc * LD + sub + i * GS]; a += xx * xx; }
            a = group_sum<GS>(a);
            if (sub == 0 && c < CP) nrm[c] = a;
        }
        if (t == 0) { *flag = 0; nrm[CP] = 0.0; }
        block_sync();
        for (int s = 0; s < NB - 1; ++s) {
            if (wave >= JW) { block_sync(); continue; }
            int I, J;
            if (wave == 0) { I = NB - 1; J = s; }
            else {
                I = s + wave; if (I >= NB - 1) I -= NB - 1;
                J = s - wave; if (J < 0) J += NB - 1;
            }
            const int i0 = I * m, j0 = J * m;
            if (s == 0 && m > 1) {
                const int half = me / 2;
                for (int u = 0; u < me - 1; ++u) {
                    for (int ps0 = 0; ps0 < 2 * half; ps0 += SLOTS) {
                        const int ps = ps0 + slot;
                        const int base = (ps < half) ? i0 : j0;
                        const int k = (ps < half) ? ps : ps - half;
                        int a, b;
                        if (k == 0) { a = me - 1; b = u; }
                        else {
                            a = u + k; if (a >= me - 1) a -= me - 1;
                            b = u - k; if (b < 0) b += me - 1;
                        }
                        const bool valid = (ps < 2 * half) && (a < m) && (b < m);
                        // an idle slot rotates the (all-zero) last padding column with itself: a no-op
                        rotate_pair_padded<GS, RPL>(Bm, LD, valid ? base + a : CP, valid ? base + b : CP, sub, nrm, flag, skip2, exit2);
                    }
                    wave_sync_lds();
                }
            }
            // cross pairs (I_a, J_(a+j)): the lane group of slot a keeps column I_a (and its tracked norm) in
            // registers for all m inner steps -- loaded once, stored once -- only the J column moves through LDS
            for (int ps0 = 0; ps0 < m; ps0 += SLOTS) {
                const int ps = ps0 + slot;
                const bool valid = ps < m;
                const int pc = valid ? i0 + ps : CP;
                double* cp = Bm + pc * LD;
                double x[RPL];
#pragma unroll
                for (int i = 0; i < RPL; ++i) x[i] = cp[sub + i * GS];
                double a = nrm[pc];
                for (int j = 0; j < m; ++j) {
                    int bq = ps + j; if (bq >= m) bq -= m;
                    const int qc = valid ? j0 + bq : CP;
                    double* cq = Bm + qc * LD;
                    double y[RPL];
                    double gg = 0.0, gg2 = 0.0;
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        y[i] = cq[sub + i * GS];
                        if (i & 1) gg2 += x[i] * y[i]; else gg += x[i] * y[i];
                    }
                    const double bb = nrm[qc];
                    gg = group_sum<GS>(gg + gg2);
                    const double g2 = gg * gg, ab = a * bb;
                    if (g2 > skip2 * ab) {
                        const double dd = bb - a;
                        const double hh = dd * dd + 4.0 * g2;
                        const double h = hh * fast_rsqrt1(hh);
                        const double tt = (dd >= 0.0 ? 2.0 : -2.0) * gg * fast_rcp1(fabs(dd) + h);
                        const double c = fast_rsqrt(1.0 + tt * tt), sn = c * tt;
#pragma unroll
                        for (int i = 0; i < RPL; ++i) {
                            const double xn = c * x[i] - sn * y[i];
                            cq[sub + i * GS] = sn * x[i] + c * y[i];
                            x[i] = xn;
                        }
                        if (sub == 0) {
                            nrm[qc] = bb + tt * gg;
                            if (g2 > exit2 * ab) lds_or(flag, 1);
                        }
                        a -= tt * gg;
                    }
                    wave_sync_lds();  // the J columns just written are read by other lane groups in the next step
                }
#pragma unroll
                for (int i = 0; i < RPL; ++i) cp[sub + i * GS] = x[i];
                if (sub == 0) nrm[pc] = a;
            }
            block_sync();
        }
        converged = (*flag == 0);
        block_sync();  // everyone has read the flag before it is cleared again
    }
    for (int c0 = 0; c0 < N; c0 += NG) {
        const int c = c0 + grp;
        const int cc = c < N ? c : N - 1;
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) { const double xx = Bm[cc * LD + sub + i * GS]; a += xx * xx; }
        a = group_sum<GS>(a);
        if (sub == 0 && c < N) sigma[c] = a * fast_rsqrt(a);
    }
    block_sync();
    return converged;
}

// ---- Jacobi kernel of the split pipeline: one (pair, layer) item per workgroup, ONE matrix in LDS, so that four
// workgroups share a CU and hide each other's dependency latency (the rotation sequence of one matrix is strictly
// sequential: N-1 steps per sweep).
// LDJ: leading dimension of the LDS matrix inside the Jacobi kernel, == 8 (mod 32) eight-byte slots: the column
// pairs a 32-lane group rotates together are ADJACENT columns (8 lanes each), so consecutive columns must start 8
// bank slots apart to be conflict-free (with the odd LD of the other kernels they overlapped: 41 % of the LDS cycles
// of this kernel were bank conflicts, profiles/r1f_pmc_counters.txt).
#ifndef SMRT_JACOBI_GS
#define SMRT_JACOBI_GS 8     // lanes per column pair in the Jacobi kernel
#endif
#ifndef SMRT_JACOBI_NT
#define SMRT_JACOBI_NT 256   // threads per workgroup of the Jacobi kernel
#endif
struct JacobiPlan { int NMAX, LD, LDJ, NCOL, o_sigma, o_rsig, o_int, total; };
SMRT_HD JacobiPlan make_jacobi_plan(int n_max_stream, int P) {
    JacobiPlan p;
    p.NMAX = n_max_stream * P;
    p.LD = (p.NMAX + 1) | 1;                            // layout of the staged matrices in global memory (make_plan)
    // padded rows = RPL * GS with the rows-per-lane count dort_jacobi_item dispatches on
    const int rows = p.NMAX > 64 ? 128 : p.NMAX > 32 ? 64 : p.NMAX > 16 ? 32 : p.NMAX > 8 ? 16 : 8;
    p.LDJ = ((rows + 31) / 32) * 32 + SMRT_JACOBI_GS;
    p.NCOL = ((p.NMAX + 7) / 8) * 8 + 1;                // NB*ceil(N/NB) <= this - 1, plus the idle-slot column
    int o = p.NCOL * p.LDJ;
    p.o_sigma = o; o += p.NMAX + 16;
    p.o_rsig = o; o += p.NMAX + 16; // tracked column norms (padded columns included)
    p.o_int = o; o += 4;
    p.total = o;
    return p;
}

template <int NT, int RPL>
SMRT_DEV void dort_jacobi_item_impl(const DevBatch& b, const DevStage& stg, long long item, double* lds) {
    constexpr int JW = (NT / SMRT_LANES >= 4) ? 4 : NT / SMRT_LANES;
    constexpr int GS = SMRT_JACOBI_GS;
    constexpr int NB = 2 * JW;
    const int t = tid();
    const int nmodes = (b.mode == 1) ? b.m_max + 1 : 1;   // active: items are (pair, azimuth mode, layer)
    const long long p = item / ((long long)b.Lmax * nmodes);
    const int l = (int)(item % b.Lmax);
    const long long gp = b.pair_begin + p;
    const int si = (int)(gp % b.S);
    if (l >= b.n_layers[si]) return;          // uniform
    if (b.status[p] != ST_OK) return;         // the prep kernel flagged this pair (uniform)
    const JacobiPlan plan = make_jacobi_plan(b.n_max_stream, b.mode == 1 ? 3 : 2);
    const int LD = plan.LD, LDJ = plan.LDJ;
    const int N = stg.n[item];
    if (N <= 0) return;                       // the prep kernel flagged this layer (uniform)
    double* M = lds;
    double* sigma = lds + plan.o_sigma;
    double* nrm = lds + plan.o_rsig;
    int* ints = (int*)(lds + plan.o_int);
    double* gB = stg.B + item * stg.mat_stride;
    // load B and zero the padding: rows N..RPL*GS-1 of every used column, columns N..CP (CP = NB*m, plus the idle
    // slot column CP itself)
    const int m = (N + NB - 1) / NB;
    const int CP = NB * m;
    for_2d<NT>(RPL * GS, CP + 1, [&](int r, int c) { M[c * LDJ + r] = (r < N && c < N) ? gB[c * LD + r] : 0.0; });
    if (t == 0) ints[0] = 0;
    block_sync();
    const bool ok = (b.mode == 1) ? jacobi_padded<NT, JW, GS, RPL>(M, N, LDJ, sigma, nrm, &ints[0], 1e-30, 1e-22)
                                  : jacobi_padded<NT, JW, GS, RPL>(M, N, LDJ, sigma, nrm, &ints[0]);
    if (!ok) { if (t == 0) stg.n[item] = -ST_EIGEN; return; }   // per layer, like the prep kernel's failures
#ifdef SMRT_SORT_EIGENPAIRS
    // eigenpairs in ascending order of the singular value (the order of the streams in the no-scattering limit, where
    // column c then belongs to row c): rank by counting, the dead norm buffer holds the permutation
    int* src = (int*)nrm;
    for (int r = t; r < N; r += NT) {
        const double sg = sigma[r];
        int rank = 0;
        for (int j = 0; j < N; ++j) { const double sj = sigma[j]; rank += (sj < sg || (sj == sg && j < r)) ? 1 : 0; }
        src[rank] = r;
    }
    block_sync();
    for_2d<NT>(N, N, [&](int r, int c) { gB[c * LD + r] = M[src[c] * LDJ + r]; });
    for (int r = t; r < N; r += NT) stg.sigma[item * stg.vec_stride + r] = sigma[src[r]];
#else
    for_2d<NT>(N, N, [&](int r, int c) { gB[c * LD + r] = M[c * LDJ + r]; });
    for (int r = t; r < N; r += NT) stg.sigma[item * stg.vec_stride + r] = sigma[r];
#endif
}

template <int NT>
SMRT_DEV void dort_jacobi_item(const DevBatch& b, const DevStage& stg, long long item, double* lds) {
    const int NMAX = b.n_max_stream * (b.mode == 1 ? 3 : 2);   // rows per lane = ceil(NMAX / 8): RPL * 8 <= NMAX (< LD)
    constexpr int G = SMRT_JACOBI_GS;   // rows per lane = padded rows / lanes per column pair
    if (NMAX > 64) dort_jacobi_item_impl<NT, 128 / G>(b, stg, item, lds);
    else if (NMAX > 32) dort_jacobi_item_impl<NT, 64 / G>(b, stg, item, lds);
    else if (NMAX > 16) dort_jacobi_item_impl<NT, 32 / G>(b, stg, item, lds);
    else if (NMAX > 8) dort_jacobi_item_impl<NT, 16 / G>(b, stg, item, lds);
    else dort_jacobi_item_impl<NT, 8 / G>(b, stg, item, lds);
}

}  // namespace smrt
