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

// The device code in reading order:
#include "dort_layout.hpp"         // descriptors, LDS plan
#include "dort_physics.hpp"        // layer electromagnetics, Fresnel, Planck
#include "dort_dense.hpp"          // Cholesky, triangular kernels, GEMM passes, in-kernel Jacobi
#include "dort_gauss_jordan.hpp"   // blocked Gauss-Jordan
#include "dort_passive.hpp"        // per-pair driver, passive mode (dort_active.hpp: active mode)
#include "dort_jacobi_kernel.hpp"  // Jacobi kernel of the pipelines
#include "dort_jacobi_big.hpp"     // ... for matrices larger than LDS (N > 128)
