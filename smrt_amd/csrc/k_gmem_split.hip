// Three-kernel pipeline for 64 < N <= 128 (passive and active): prep and the four-matrix finish on the per-workgroup
// hipcc-flags: -mllvm -disable-machine-licm
// (fewer loop-invariant values hoisted and spilled: active 32 streams 1496 -> 1573 solves/s, profiles/r5_nolicm_ab.txt)
// global workspace (L2 / Infinity-Cache resident, grid-stride over the pairs so that it stays bounded); the shared
// Jacobi kernel with its 128-column LDS matrix runs in between.
#include <cstdlib>
#include "dort_ctx.hpp"
#include "dort_active.hpp"

using namespace smrt;

#ifndef SMRT_GMEM_FINISH_WAVES
#define SMRT_GMEM_FINISH_WAVES 2   // wavefronts per SIMD the finish kernels leave room for
#endif

#ifndef SMRT_GMEM_PREP_WAVES
#define SMRT_GMEM_PREP_WAVES 2
#endif
template <int NT>
__global__ __launch_bounds__(NT, SMRT_GMEM_PREP_WAVES) void dort_prep_kernel_gmem(DevBatch b, DevStage st, double* workspace, long long ws_stride) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    double* mat = workspace + (long long)blockIdx.x * ws_stride;
    for (long long p = blockIdx.x; p < b.pair_count; p += gridDim.x) {
        dort_pair_passive<NT, 2, 1>(b, dispatched_pair(b, p), smrt_lds, mat, &st);
        __syncthreads();
    }
}
template <int NT>
__global__ __launch_bounds__(NT, SMRT_GMEM_FINISH_WAVES) void dort_finish_kernel_gmem(DevBatch b, DevStage st, double* workspace, long long ws_stride) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    double* mat = workspace + (long long)blockIdx.x * ws_stride;
    for (long long p = blockIdx.x; p < b.pair_count; p += gridDim.x) {
        dort_pair_passive<NT, 2, 2>(b, dispatched_pair(b, p), smrt_lds, mat, &st);
        __syncthreads();
    }
}
template <int NT>
__global__ __launch_bounds__(NT) void dort_active_prep_kernel_gmem(DevBatch b, DevStage st, double* workspace, long long ws_stride) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    double* mat = workspace + (long long)blockIdx.x * ws_stride;
    for (long long p = blockIdx.x; p < b.pair_count; p += gridDim.x) {
        dort_pair_active<NT, 2, 1>(b, dispatched_pair(b, p), smrt_lds, mat, &st);
        __syncthreads();
    }
}
template <int NT>
__global__ __launch_bounds__(NT, SMRT_GMEM_FINISH_WAVES) void dort_active_finish_kernel_gmem(DevBatch b, DevStage st, double* workspace, long long ws_stride) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    double* mat = workspace + (long long)blockIdx.x * ws_stride;
    for (long long p = blockIdx.x; p < b.pair_count; p += gridDim.x) {
        dort_pair_active<NT, 2, 2>(b, dispatched_pair(b, p), smrt_lds, mat, &st);
        __syncthreads();
    }
}

namespace smrt_launch {

template <class K>
static hipError_t go(K kern, smrt_dort_ctx* ctx, const DevBatch& c, unsigned grid, size_t lds, unsigned nt = 256) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, ctx->stream, c, ctx->stage, (double*)ctx->d_work.p, ctx->ws_stride);
    return hipGetLastError();
}

hipError_t prep_gmem(smrt_dort_ctx* ctx, const DevBatch& c, unsigned grid, bool active) {
    return active ? go(dort_active_prep_kernel_gmem<256>, ctx, c, grid, ctx->prep_lds_bytes)
                  : go(dort_prep_kernel_gmem<256>, ctx, c, grid, ctx->prep_lds_bytes);
}
hipError_t finish_gmem(smrt_dort_ctx* ctx, const DevBatch& c, unsigned grid, bool active) {
    // One workgroup of EIGHT wavefronts per CU instead of two of four: the same wavefronts per SIMD and registers per
    // wavefront, but one row tile per wavefront at N = 128, twice the wavefronts on every solve of a pair, and half the
    // global workspace in flight (135 MB instead of 270: it stays in the Infinity Cache).  configs[2] shape: 3863 -> 3972
    // solves/s on 448 pairs, 3261 -> 3690 on 1792, 3685 -> 3785 on 7168.  SMRT_DORT_GMEM_FINISH_256=1: the old shape.
    static const bool wide = getenv("SMRT_DORT_GMEM_FINISH_256") == nullptr;
    if (wide)
        return active ? go(dort_active_finish_kernel_gmem<512>, ctx, c, grid, ctx->finish2_lds_bytes, 512)
                      : go(dort_finish_kernel_gmem<512>, ctx, c, grid, ctx->finish2_lds_bytes, 512);
    return active ? go(dort_active_finish_kernel_gmem<256>, ctx, c, grid, ctx->finish2_lds_bytes)
                  : go(dort_finish_kernel_gmem<256>, ctx, c, grid, ctx->finish2_lds_bytes);
}

}  // namespace smrt_launch
