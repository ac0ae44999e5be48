// The symmetric eigensolver of the N <= 64 pipelines (dort_eig_kernel.hpp): tridiag -> chase -> vectors in place of the
// one-sided Jacobi kernel, on the same staging items (pair, [azimuth mode,] layer).
#include <cstdlib>
#include "dort_ctx.hpp"
#include "dort_device.hpp"
#include "dort_eig_kernel.hpp"

using namespace smrt;

// One wavefront per item; a launch holds the instantiations of the padded row counts of ITS size class (LO, HI] only, so
// that the small items run with the registers (and, in tridiag, the LDS) of their own size; an item of another class
// leaves at once (like dort_jacobi_kernel).
#define SMRT_EIG_ROWS(NP_, CALL) \
    if constexpr (LO < (NP_) && (NP_) - 8 < HI) if (rows <= (NP_)) { CALL; return; }
#define SMRT_EIG_ROWS16(NP_, CALL) \
    if constexpr (LO < (NP_) && (NP_) - 16 < HI) if (rows <= (NP_)) { CALL; return; }

template <int LO, int HI>
__global__ __launch_bounds__(64) void dort_eig_gram_kernel(DevBatch b, DevStage st) {
    const long long item = uniform(jacobi_item_of_block(b, (long long)blockIdx.x));
    const int rows = uniform(eig_item_rows(b, st, item));
    if (rows <= LO || rows > HI) return;
    SMRT_EIG_ROWS16(16, (eig_gram_item<16>(st, item))) SMRT_EIG_ROWS16(32, (eig_gram_item<32>(st, item)))
    SMRT_EIG_ROWS16(48, (eig_gram_item<48>(st, item))) SMRT_EIG_ROWS16(64, (eig_gram_item<64>(st, item)))
}

template <int LO, int HI>
__global__ __launch_bounds__(64) void dort_eig_tridiag_kernel(DevBatch b, DevStage st) {
    const long long item = uniform(jacobi_item_of_block(b, (long long)blockIdx.x));
    const int rows = uniform(eig_item_rows(b, st, item));
    if (rows <= LO || rows > HI) return;
    SMRT_EIG_ROWS(8, (eig_tridiag_item<8>(st, item))) SMRT_EIG_ROWS(16, (eig_tridiag_item<16>(st, item)))
    SMRT_EIG_ROWS(24, (eig_tridiag_item<24>(st, item))) SMRT_EIG_ROWS(32, (eig_tridiag_item<32>(st, item)))
    SMRT_EIG_ROWS(40, (eig_tridiag_item<40>(st, item))) SMRT_EIG_ROWS(48, (eig_tridiag_item<48>(st, item)))
    SMRT_EIG_ROWS(56, (eig_tridiag_item<56>(st, item))) SMRT_EIG_ROWS(64, (eig_tridiag_item<64>(st, item)))
}

template <int LO, int HI>
__global__ __launch_bounds__(64) void dort_eig_vectors_kernel(DevBatch b, DevStage st) {
    __shared__ __attribute__((aligned(16))) double ring[2 * kEigRingSlots];
    const long long item = uniform(jacobi_item_of_block(b, (long long)blockIdx.x));
    const int rows = uniform(eig_item_rows(b, st, item));
    if (rows <= LO || rows > HI) return;
    SMRT_EIG_ROWS(8, (eig_vectors_item<8>(st, item, ring))) SMRT_EIG_ROWS(16, (eig_vectors_item<16>(st, item, ring)))
    SMRT_EIG_ROWS(24, (eig_vectors_item<24>(st, item, ring))) SMRT_EIG_ROWS(32, (eig_vectors_item<32>(st, item, ring)))
    SMRT_EIG_ROWS(40, (eig_vectors_item<40>(st, item, ring))) SMRT_EIG_ROWS(48, (eig_vectors_item<48>(st, item, ring)))
    SMRT_EIG_ROWS(56, (eig_vectors_item<56>(st, item, ring))) SMRT_EIG_ROWS(64, (eig_vectors_item<64>(st, item, ring)))
}

// one LANE per item: d and e of the kEigChaseLanes items of a wavefront in LDS, element i of lane t at [kEigChaseLanes i + t].
// Items in their staging order.  Measured and dropped (profiles/r6_eig_steps.txt 7): lists of the items by size class (filled by
// the tridiag kernel with atomics) so that the 64 items of a wavefront have similar sizes -- a simulation of the wavefront
// schedule promised 3700 -> 2700 steps per wavefront --, one launch per class with the LDS of its size (3.75 ms: every launch of
// this latency chain pays its own tail) or one launch through the lists (3.41 ms small classes first, 3.15 ms largest first)
// against 3.18 ms as it is.  The chase of a pass on a high-priority stream of its own (events around it), so that its long
// chains start as early as the dispatcher allows while other passes fill the chip: 29.4 against 29.2 ms per step -- dropped.
__global__ __launch_bounds__(kEigChaseLanes) void dort_eig_chase_kernel(DevBatch b, DevStage st, long long items) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    const long long blk = (long long)blockIdx.x * kEigChaseLanes + threadIdx.x;
    if (blk >= items) return;
    const long long item = jacobi_item_of_block(b, blk);
    if (eig_item_rows(b, st, item) <= 0) return;
    const int nmax = st.vec_stride;
    eig_chase_lane(st, item, smrt_lds + threadIdx.x, smrt_lds + kEigChaseLanes * nmax + threadIdx.x);
}

namespace smrt_launch {
template <int LO, int HI>
static hipError_t go_gram(smrt_dort_ctx* ctx, const DevBatch& c, long long items) {
    hipLaunchKernelGGL((dort_eig_gram_kernel<LO, HI>), dim3((unsigned)items), dim3(64), 0, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}
template <int LO, int HI>
static hipError_t go_tridiag(smrt_dort_ctx* ctx, const DevBatch& c, long long items) {
    hipLaunchKernelGGL((dort_eig_tridiag_kernel<LO, HI>), dim3((unsigned)items), dim3(64), 0, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}
template <int LO, int HI>
static hipError_t go_vectors(smrt_dort_ctx* ctx, const DevBatch& c, long long items) {
    hipLaunchKernelGGL((dort_eig_vectors_kernel<LO, HI>), dim3((unsigned)items), dim3(64), 0, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}

// Size classes by row count: (0, 32], (32, 48], (48, 64] -- the register rows of tridiag / vectors are 64 / 96 / 128 registers
// (3 / 3 / 2 wavefronts per SIMD).  Measured and dropped: a class (48, 56] of its own held to three wavefronts per SIMD by
// __launch_bounds__ (52 / 12 B of scratch): tridiag 2.61 against 2.02 ms for the two classes, vectors 3.34 against 3.47.
hipError_t eig(smrt_dort_ctx* ctx, const DevBatch& c, long long items) {
    const int nmax = ctx->nmax_rows;
    hipError_t e;
    if ((e = go_gram<0, 32>(ctx, c, items)) != hipSuccess) return e;
    if (nmax > 32 && (e = go_gram<32, 48>(ctx, c, items)) != hipSuccess) return e;
    if (nmax > 48 && (e = go_gram<48, 64>(ctx, c, items)) != hipSuccess) return e;
    if ((e = go_tridiag<0, 32>(ctx, c, items)) != hipSuccess) return e;
    if (nmax > 32 && (e = go_tridiag<32, 48>(ctx, c, items)) != hipSuccess) return e;
    if (nmax > 48 && (e = go_tridiag<48, 64>(ctx, c, items)) != hipSuccess) return e;
    {
        const size_t lds = (size_t)2 * kEigChaseLanes * nmax * sizeof(double);
        e = hipFuncSetAttribute((const void*)dort_eig_chase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(dort_eig_chase_kernel, dim3((unsigned)((items + kEigChaseLanes - 1) / kEigChaseLanes)), dim3(kEigChaseLanes), lds, ctx->stream, c, ctx->stage, items);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if ((e = go_vectors<0, 32>(ctx, c, items)) != hipSuccess) return e;
    if (nmax > 32 && (e = go_vectors<32, 48>(ctx, c, items)) != hipSuccess) return e;
    if (nmax > 48 && (e = go_vectors<48, 64>(ctx, c, items)) != hipSuccess) return e;
    return hipSuccess;
}
}  // namespace smrt_launch
