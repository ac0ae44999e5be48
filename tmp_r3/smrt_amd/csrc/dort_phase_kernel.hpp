// The emmodel protocol's ft_even_phase on the device (smrt/emmodel/common.py:349-399 for IBA: discrete Fourier
// decomposition in azimuth of the phase matrix, with the mirror / sign rules of generic_ft_even_matrix :56-131;
// smrt/emmodel/rayleigh.py:52-127 for the Rayleigh emmodels: closed forms, modes 0..2): the azimuthal modes of the phase
// matrix of ONE layer on arbitrary grids of scattered / incident cosines -- what a foreign rtsolver asks an emmodel for.
// The DORT kernels do not call this (they assemble the symmetrised S+- combinations on their own stream sets in place);
// it shares the layer electromagnetics and the microstructure functions with them.
// Part of the DORT device code (see dort_device.hpp for the overview and the reference map).
#pragma once
#include "dort_physics.hpp"

namespace smrt {

struct PhaseRequest {
    int em, ms;                       // emmodel, microstructure model of the layer
    double frequency, frac_volume, temperature, p1, p2;
    const double* mu_s; int n_s;      // cosines of the scattered directions (any sign)
    const double* mu_i; int n_i;      // cosines of the incident directions
    int m_max, npol, nsamp;           // nsamp: azimuth samples, azimuth_samples(m_max) = the reference's estimate
    double* out;                      // [npol][npol][m_max + 1][n_s][n_i]
    int* status;                      // 0, or ST_INPUT for an invalid layer
};

// one thread per (scattered, incident) pair
SMRT_DEV void ft_even_phase_entry(const PhaseRequest& q, int is, int ii) {
    cplx ee; double ks, ka, pa, pb; int bad = 0;
    layer_em(q.em, q.ms, q.frequency, q.frac_volume, q.temperature, q.p1, q.p2, &ee, &ks, &ka, &pa, &pb, &bad);
    if (bad || !(ks >= 0.0)) { if (is == 0 && ii == 0) *q.status = ST_INPUT; return; }
    const double mi = q.mu_s[is], x = q.mu_i[ii];
    const double sis = sqrt(1.0 - mi * mi), sjs = sqrt(1.0 - x * x);
    const int P = q.npol;
    const long long plane = (long long)q.n_s * q.n_i, at0 = (long long)is * q.n_i + ii;
    for (int m = 0; m <= q.m_max; ++m) {
        double e[3][3];
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) e[a][c] = 0.0;
        if (q.em == EM_NONSCAT) {
            // null phase matrix
        } else if (q.em != EM_IBA && q.em != EM_IBA_INV) {   // Rayleigh closed forms times 1.5 ks (pa); the minus on the (V|H, U) column: :121-124
            const double a2 = mi * mi, x2 = x * x;
            if (m == 0) {
                e[0][0] = pa * (0.5 * a2 * x2 + (1.0 - a2) * (1.0 - x2));
                e[0][1] = pa * 0.5 * a2; e[1][0] = pa * 0.5 * x2; e[1][1] = pa * 0.5;
            } else if (m == 1) {
                const double cs = mi * sis, ci = x * sjs;
                e[0][0] = pa * 2.0 * cs * ci; e[0][2] = -pa * cs * sjs;
                e[2][0] = -pa * 2.0 * sis * ci; e[2][2] = pa * sis * sjs;
            } else if (m == 2) {
                e[0][0] = pa * 0.5 * a2 * x2; e[0][1] = -pa * 0.5 * a2; e[1][0] = -pa * 0.5 * x2; e[1][1] = pa * 0.5;
                e[0][2] = -pa * 0.5 * a2 * x; e[1][2] = pa * 0.5 * x;
                e[2][0] = -pa * mi * x2; e[2][1] = pa * mi; e[2][2] = pa * mi * x;
            }
        } else {
            // IBA: p(phi) = C(cos Theta) x (products of the scattering amplitudes), sampled on [0, pi] and mirrored: the
            // even entries are cosine sums, the (V|H, U) / (U, V|H) entries -/+ sine sums
            const int nphi = q.nsamp / 2 + 1;
            const double base = ((m == 0) ? 1.0 : 2.0) / (double)q.nsamp;
            for (int k = 0; k < nphi; ++k) {
                const double ph = kPi * (double)k / (double)(nphi - 1);
                const double c = cos(ph), sn = sin(ph);
                const bool end = (k == 0 || k == nphi - 1);
                const double cw = base * (end ? 1.0 : 2.0) * cos((double)m * ph);
                const double sw = end ? 0.0 : base * 2.0 * sin((double)m * ph);
                double ct = mi * x + sis * sjs * c;   // cosine of the scattering angle
                ct = ct > 1.0 ? 1.0 : (ct < -1.0 ? -1.0 : ct);
                double C;
                if (q.ms == MS_EXP) { const double dp = 1.0 + pb * (1.0 - ct); C = pa / (dp * dp); }
                else C = pa * ft_corr(MS_SHS, pb * (1.0 - ct), q.frac_volume, q.p1, q.p2);
                const double fvv = c * mi * x + sis * sjs, fvh = sn * mi, fhv = -sn * x, fhh = c;
                const double Cc = C * cw, Cs = C * sw;
                e[0][0] += fvv * fvv * Cc; e[0][1] += fvh * fvh * Cc;
                e[1][0] += fhv * fhv * Cc; e[1][1] += fhh * fhh * Cc;
                if (P == 3) {
                    e[2][2] += (fvv * fhh + fvh * fhv) * Cc;
                    e[0][2] -= fvh * fvv * Cs; e[1][2] -= fhh * fhv * Cs;
                    e[2][0] += 2.0 * fvv * fhv * Cs; e[2][1] += 2.0 * fvh * fhh * Cs;
                }
            }
        }
        for (int a = 0; a < P; ++a)
            for (int c = 0; c < P; ++c)
                q.out[((long long)(a * P + c) * (q.m_max + 1) + m) * plane + at0] = e[a][c];
    }
}

}  // namespace smrt
