// Batch / staging descriptors as the device sees them and the LDS layout shared by the host sizing code and the kernels.
// Part of the DORT device code (see dort_device.hpp for the overview and the reference map).
#pragma once
#include "spmd.hpp"
#include <math.h>
#include <string.h>

#if defined(SMRT_HOST_EMU)
#define SMRT_HD_EARLY inline
#else
#define SMRT_HD_EARLY __host__ __device__ inline
#endif

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
    const long long* pair_map;  // null: workgroup p solves pair pair_begin + p of the flattened f * S + s list; else
                                // pair pair_map[pair_begin + p] (sparse selections, smrt_dort_upload_pairs)
    const int* n_layers;
    const double* thickness;
    const double* frac_volume;
    const double* temperature;
    const double* p1;
    const double* p2;
    const double* frequency;
    const double* theta;
    const int* layer_kind;  // [S][Lmax] emmodel + 16 * microstructure of every layer, or null: b.emmodel / b.micro everywhere
    // emmodels evaluated by the caller (layers of kind EM_HOST), indexed by the global pair f * S + s (smrt_dort.h)
    const double* host_layer;    // [F * S][Lmax][4] ks, ka, Re eps, Im eps
    const double* host_coeff;    // [F * S][Lmax] layers of kind EM_IBA_HOST: the coefficient of IBA's phase matrix
    const int* host_streams;     // [F * S][Lmax]
    const double* host_phase;    // [F * S][Lmax][host_modes][2][host_ne * host_ne]
    int host_modes, host_ne;
    int coherent;  // DORT option process_coherent_layers
    const double* host_substrate;      // SUB_HOST: [F * S][m_max + 1][NE * NE] dense bottom reflection, NE = 3 n_max_stream
    const double* host_substrate_coh;  //           [F * S][m_max + 1][NE] its specular diagonal
    // rough interfaces evaluated by the caller (smrt_dort.h: SMRT_INTERFACE_HOST), indexed by the global pair
    const int* host_itf_slot;          // [F * S][Lmax], -1: Flat; or null
    const double* host_itf;            // [F * S][slots][modes][4][NE * NE]
    const double* host_itf_coh;        // [F * S][slots][4][NE]
    int host_itf_slots;
    const double* liquid_water;  // [S][Lmax] liquid water of wet layers (water / (ice + water) volume), or null: dry snow
    const double* gl_mu;  // [n_max_stream] positive Gauss-Legendre nodes of order 2 n_max, descending
    int sub_kind;                         // 0 none, 1 flat (p1 + i p2 = permittivity), 2 reflector (p1, p2 = R_V, R_H)
    const double *sub_p1, *sub_p2;        // [F][S]
    const double* sub_T;                  // [S], <= 0: no emission
    const double *atm_down, *atm_up, *atm_trans;  // [F] or null
    double phi;
    double jacobi_skip2, jacobi_exit2;  // squared-cosine thresholds of the Jacobi kernel (host: per mode, env override)
    double prune_tau;  // > 0: optical depth beyond which the deeper layers are dropped (dort.py:443-452); pipeline only
    // Rounds of the pipelines under prune_deep_snowpack: the prep and Jacobi kernels of one round cover the layers
    // [layer_lo, layer_hi) of the pairs whose cut has not been reached yet (pair_done[p] == 0; null: every pair), so the
    // layers below a cut are never diagonalised -- like in the reference, which stops assembling there.  Otherwise
    // layer_lo = 0 and layer_hi = Lmax.
    int layer_lo, layer_hi;
    const int* pair_done;
    // Order in which the workgroups of a launch take the pairs of the chunk: dispatch[w] = pair slot of workgroup w (a
    // permutation of 0 .. pair_count - 1, sorted by estimated cost at upload time), or null: w itself.  Everything else --
    // outputs, staging items, pair_done -- stays indexed by the pair slot.
    const int* dispatch;
    double* out;
    int* status;
    double* layer_out;
    double* stream_out;
    double* n3_out;  // [pair_count] sum_l N_l^3 (work counter for the roofline)
    double* stage_out;  // [pair_count][16] shader cycles per stage (only written by -DSMRT_STAGE_TIMING builds)
    // Layers with a Rayleigh phase matrix (DMRT-QCA short range, ...) skip Cholesky / B / the Jacobi sweeps: the prep kernel
    // stages their scalars, dort_rayleigh_kernel.hpp diagonalises them in closed form and the strip finish kernels take
    // A+ = D V as it is (host: only where a strip finish kernel consumes the staging area)
    int rayleigh_direct;
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
    double* Linv;          // [item][linv_stride] inverses of the 16x16 diagonal blocks of L+, 256 doubles each (written by the prep kernel)
    double* ws;            // [pair][4096] one 64 x 64 matrix per pair (register-resident finish kernel), or null
    int linv_stride;       // 1024 (N <= 64: four blocks) or 2048 (64 < N <= 128: eight blocks)
    // the symmetric eigensolver (dort_eig_kernel.hpp), or null / 0: the Jacobi kernel diagonalises
    double* eig_e;         // [item][2 * vec_stride] off-diagonal of the tridiagonal form (+ the scale), then the tau of the reflectors
    double* eig_rot;       // [item][rot_stride] the plane rotations of the QL iterations
    long long rot_stride;
};

// stg.n[item] of a layer diagonalised by the Rayleigh kernel: its row count + kStageDirect (the Jacobi / eigensolver
// kernels see a count beyond their sizes and leave; the finish kernel reads the flag)
constexpr int kStageDirect = 4096;
SMRT_DEV int stage_rows(int n) { return n > 0 ? (n & (kStageDirect - 1)) : n; }
SMRT_DEV bool stage_direct(int n) { return n > kStageDirect; }

// index into the flattened (frequency-major) pair list of the batch for the p-th workgroup of a launch
SMRT_DEV long long global_pair(const DevBatch& b, long long p) {
    return b.pair_map ? b.pair_map[b.pair_begin + p] : b.pair_begin + p;
}

// pair slot handled by the w-th workgroup (or grid-stride step) of a launch
SMRT_DEV long long dispatched_pair(const DevBatch& b, long long w) { return b.dispatch ? (long long)b.dispatch[w] : w; }

// staging item of the blk-th workgroup of a Jacobi launch that covers the layers [layer_lo, layer_hi) of every
// (pair, azimuth mode): item = (pair * modes + mode) * Lmax + layer, pairs in dispatch order
SMRT_DEV long long jacobi_item_of_block(const DevBatch& b, long long blk) {
    const int span = b.layer_hi - b.layer_lo;
    long long row = blk / span;   // pair * modes + mode
    if (b.dispatch) {
        const int nmodes = (b.mode == 1) ? b.m_max + 1 : 1;
        row = (long long)b.dispatch[row / nmodes] * nmodes + row % nmodes;
    }
    return row * b.Lmax + b.layer_lo + (blk % span);
}

constexpr double kCSpeed = 299792458.0;
constexpr double kPlanck = 6.62607015e-34;
constexpr double kBoltzmann = 1.380649e-23;
constexpr double kFreezing = 273.15;
constexpr double kPi = 3.14159265358979323846;

enum { EM_IBA = 0, EM_DMRT = 1, EM_QCACP = 2, EM_NONSCAT = 3, EM_HOST = 4, EM_IBA_INV = 5, EM_IBA_HOST = 6, EM_RAYLEIGH_HOST = 7 };  // 1-3 have a Rayleigh phase matrix; 4: host arrays;
// 5: IBA on the inverted medium (layer_em); pair_setup files such a layer as EM_IBA once its coefficients are computed
// emmodels whose azimuth mode 0 is the Rayleigh phase matrix (closed form in the prep kernel; dort_rayleigh_kernel.hpp)
SMRT_HD_EARLY bool em_has_rayleigh_phase(int em) { return em == EM_DMRT || em == EM_QCACP || em == EM_NONSCAT || em == EM_RAYLEIGH_HOST; }
enum { MS_EXP = 0, MS_SHS = 1, MS_SPHERE = 2, MS_TS = 3, MS_EXPC = 4, MS_SHSC = 5, MS_SPHEREC = 6, MS_TSC = 7 };   // 4 + model: the model at a complex wavenumber (smrt_dort.h)
enum { ST_OK = 0, ST_EIGEN = 1, ST_NORM = 2, ST_ALBEDO = 3, ST_SINGULAR = 4, ST_INPUT = 5, ST_COHERENT = 6 };
enum { SUB_NONE = 0, SUB_FLAT = 1, SUB_REFLECTOR = 2, SUB_HOST = 3 };

// ------------------------------------------------------------------------------------------------------------
// LDS layout (shared by host sizing code and the kernel)
// ------------------------------------------------------------------------------------------------------------
struct LdsPlan {
    int NMAX, LD, nmax, Lmax, nphi, ntheta;
    int slim;             // 0: full layout, 1: prep kernel, 2: two-slot finish kernel (see make_plan)
    int matrices_in_lds;  // 1: the four N x N work matrices are LDS-resident; 0: they live in a global workspace
    int mat_doubles;      // doubles of matrix workspace per workgroup (4 * NMAX * LD)
    int scratch_doubles;  // global-workspace kernels with N > 128: scratch of the blocked solvers behind the matrices
    int o_M[4];
    int o_rowvec;   // 17 vectors of NMAX
    int o_strvec;   // 6 vectors of nmax
    int o_layvec;   // 15 vectors of Lmax
    int o_phi;      // 5 vectors of nphi
    int o_tb;       // NMAX
    int o_int;      // 16 ints (8 doubles)
    int o_gj;       // scratch of the blocked solvers (block inverses of Cholesky / triangular solve, Gauss-Jordan bookkeeping)
    int o_act;      // active mode only: per-layer mode-0 normalisation, mode totals, incident stream list
    int stage_bufs; // N > 128 finish kernels: operand staging buffers at o_jac (make_plan jac_in_lds = 3), else 0
    int o_jac;      // global-workspace kernels: an NMAX x LD LDS buffer for the Jacobi stage, or -1 if it does not fit
    int total;      // doubles
};

#if defined(SMRT_HOST_EMU)
#define SMRT_HD inline
#else
#define SMRT_HD __host__ __device__ inline
#endif

// doubles of the active-mode region: total[9][2 ntheta], coherent[2][2 ntheta], incident list, then -- only in the
// kernels that assemble the phase matrices (fused, prep) -- the mode-0 normalisation norm0[Lmax][2 nmax]
SMRT_HD int active_doubles(int n_max_stream, int Lmax, int ntheta, bool with_norm0 = true) {
    return 9 * 2 * ntheta + 2 * 2 * ntheta + (2 * ntheta + 2) / 2 + 1 + (with_norm0 ? 2 * n_max_stream * Lmax : 0);
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
    // slim = 3: the prep kernel of the LDS-resident passive pipeline -- like slim = 1, but its two symmetric matrices are
    // stored as packed lower triangles (sidx<true>, dort_dense.hpp): 34 KB instead of 67 KB at 32 streams, so that THREE
    // workgroups share a CU
    const bool packed = (slim == 3);
    if (packed) slim = 1;
    const int nmat = slim ? 2 : 4;
    const int one = packed ? (p.LD * (p.LD + 1)) / 2 : p.NMAX * p.LD;
    p.mat_doubles = nmat * one;
    // (one 16 x 16 block inverse per block row of the triangular solve: whole blocks, also when 16 does not divide NMAX)
    p.scratch_doubles = !matrices_in_lds ? 256 * ((p.NMAX + 15) / 16) : 0;
    int o = 0;
    for (int i = 0; i < 4; ++i) { p.o_M[i] = (i < nmat ? i : 0) * one; }
    if (slim == 2) p.o_M[3] = p.NMAX * p.LD;  // the two-slot finish kernel: M0 = X, M3 = R (M1, M2 live in global memory)
    if (matrices_in_lds) o = p.mat_doubles;
    p.slim = slim;
    p.o_rowvec = o; o += (slim == 1 ? 4 : slim == 2 ? 14 : 17) * p.NMAX;  // slim 2: no mrow / wrow / u
    p.o_strvec = o; o += 6 * p.nmax;
    p.o_layvec = o; o += 15 * Lmax;
    p.o_phi = o; o += (slim == 2 ? 0 : 5 * nphi);
    p.o_tb = o; o += p.NMAX;
    p.o_int = o; o += 8;
    // Gauss-Jordan bookkeeping: perm [NMAX + 16] + rowblk [NMAX] + two flags (ints), then the 16 x 16 inverse of the
    // running diagonal block (fast panel): NMAX + 11 + 256 doubles
    const int gjd = p.NMAX + 12 + 256;
    const int full = ((16 * p.NMAX + 8 > 1024 + 8) ? 16 * p.NMAX + 8 : 1024 + 8) + (p.NMAX + 8 + 1) / 2;
    p.o_gj = o; o += slim == 1 ? 520 : slim == 2 ? gjd : !matrices_in_lds ? gjd + p.NMAX / 2 + 32 : (full > gjd ? full : gjd);
    p.o_act = o; o += act_doubles;
    p.o_jac = -1;
    // 1: a whole matrix (Jacobi stage of the fused kernel); 2: only the 16 NMAX doubles of scratch that the blocked
    // triangular solve needs (finish half of the global-workspace pipeline: small LDS, several workgroups per CU)
    // 3: the operand staging of the N > 128 finish kernels (r45_mfma_big): buffers of one 16-column operand tile in
    // matrix-core lane order, 64 doubles per group of four rows
    // (two buffers while they fit the 160 KB of a CU next to the rest, else one)
    p.stage_bufs = 0;
    if (jac_in_lds && !matrices_in_lds) {
        p.o_jac = o;
        if (jac_in_lds == 3) {
            const int one = 64 * ((p.NMAX + 3) / 4);
            p.stage_bufs = ((o + 2 * one) * 8 <= 160 * 1024) ? 2 : 1;
            o += p.stage_bufs * one;
        } else o += (jac_in_lds == 2) ? 16 * p.NMAX : p.NMAX * p.LD;
    }
    p.total = o;
    return p;
}

struct Lds {
    double *M0, *M1, *M2, *M3;
    double *mrow, *wrow, *u, *d, *sigma, *rsig, *t, *Rtop, *Ttop, *Rbu, *Tbu, *cvec, *tq, *svec, *g, *upb, *up;
    double *gmu, *gsin, *outmu, *mu, *w, *muu;
    double *eps_re, *eps_im, *ks, *ka, *pa, *pb, *pc, *BT, *thick, *ri, *nl;
    double *slab_re, *slab_im, *slab_th, *lo;  // coherent slab on top of the layer (thickness 0: none); index of the layer in the input
    double *cphi, *s2phi, *wphi, *sphi, *swphi;
    double* tb;
    double* act;  // active-mode region (see active_doubles)
    int* ints;  // [0] status  [1] jacobi flag  [2] pivot  [3] pivot fail  [4] kstar  [5] n_air  [6] layers after process_coherent_layers
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
    s.slab_re = v + 11 * L; s.slab_im = v + 12 * L; s.slab_th = v + 13 * L; s.lo = v + 14 * L;
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

}  // namespace smrt
