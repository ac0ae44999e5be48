// Rough interfaces evaluated by the caller (smrt_dort.h: SMRT_INTERFACE_HOST): composition of a dense interface with the
// reflection matrix of everything below it.  Part of the DORT device code (see dort_device.hpp).
//
// The layer under a rough interface runs its usual step with a TRANSPARENT top (R_top = 0, T = 1, same streams), which
// leaves K0 and up0 with  I_up = K0 I_dn + up0  at the top of the layer for total intensities.  The interface
// (smrt/rtsolver/dort.py:356-427 with the matrices of rtsolver_utils.py:473-642)
//     I_dn(layer) = Rtop I_up(layer) + Tbot I_dn(above),      I_up(above) = Rbot I_dn(above) + Ttop I_up(layer)
// then gives, seen from the medium above,
//     R~' = Rbot + Ttop X Tbot,   s' = Ttop x,       (I - K0 Rtop) [X | x] = [K0 | up0]
// -- the same linear system as the reference's banded one, eliminated interface by interface.  One pivoted N x N solve
// and two rectangular products per rough interface and azimuth mode; plain loops (a compatibility route like the
// host-evaluated emmodels and substrates, not a fast path).
#pragma once
#include "spmd.hpp"
#include "dort_dense.hpp"

namespace smrt {

// slot of the interface on top of layer `l_input` (index in the input arrays) of global pair gp, or -1 (Flat)
SMRT_DEV int host_interface_slot(const DevBatch& b, long long gp, int l_input) {
    return b.host_itf_slot ? b.host_itf_slot[gp * b.Lmax + l_input] : -1;
}
// the four matrices [Rtop, Ttop, Rbot, Tbot] of a slot and azimuth mode, each NE x NE row-major
SMRT_DEV const double* host_interface_matrices(const DevBatch& b, long long gp, int slot, int m, int nmodes) {
    const long long NE = 3LL * b.n_max_stream;
    return b.host_itf + (((gp * b.host_itf_slots + slot) * nmodes + m) * 4) * NE * NE;
}
SMRT_DEV const double* host_interface_specular(const DevBatch& b, long long gp, int slot) {
    const long long NE = 3LL * b.n_max_stream;
    return b.host_itf_coh + ((gp * b.host_itf_slots + slot) * 4) * NE;
}

// Km: K0 (N x N, element (i, j) at [j LD + i]) -> destroyed;  Am: scratch N x N -> R~' (Nu x Nu);  up: up0 (N) -> destroyed;
// svec: s' (Nu) out;  udiag: N doubles of scratch.  with_source = false: no thermal terms (active mode), up / svec untouched.
// Returns false when the elimination meets a vanishing pivot (uniform).
template <int NT>
SMRT_DEV bool interface_dense_step(const double* H, int NE, double* Km, double* Am, double* up, double* svec, double* udiag,
                                   int N, int Nu, int LD, bool with_source) {
    const double* Rtop = H;
    const double* Ttop = H + (long long)NE * NE;
    const double* Rbot = H + 2LL * NE * NE;
    const double* Tbot = H + 3LL * NE * NE;
    const int t = tid();
    // A = I - K0 Rtop
    for_2d<NT>(N, N, [&](int i, int j) {
        double acc = (i == j) ? 1.0 : 0.0;
        for (int k = 0; k < N; ++k) acc -= Km[k * LD + i] * Rtop[k * NE + j];
        Am[j * LD + i] = acc;
    });
    block_sync();
    if (!lu_solve<NT, false>(Am, Km, with_source ? up : nullptr, udiag, N, LD)) return false;   // Km = X, up = x
    // P1 = X Tbot (N x Nu) -> Am
    for_2d<NT>(N, Nu, [&](int a, int j) {
        double acc = 0.0;
        for (int c = 0; c < N; ++c) acc += Km[c * LD + a] * Tbot[c * NE + j];
        Am[j * LD + a] = acc;
    });
    block_sync();
    // R~' = Rbot + Ttop P1 (Nu x Nu) -> Km, then to Am;  s' = Ttop x
    for_2d<NT>(Nu, Nu, [&](int i, int j) {
        double acc = Rbot[i * NE + j];
        for (int a = 0; a < N; ++a) acc += Ttop[i * NE + a] * Am[j * LD + a];
        Km[j * LD + i] = acc;
    });
    if (with_source)
        for (int i = t; i < Nu; i += NT) {
            double acc = 0.0;
            for (int a = 0; a < N; ++a) acc += Ttop[i * NE + a] * up[a];
            svec[i] = acc;
        }
    block_sync();
    for_2d<NT>(Nu, Nu, [&](int i, int j) { Am[j * LD + i] = Km[j * LD + i]; });
    block_sync();
    return true;
}

}  // namespace smrt
