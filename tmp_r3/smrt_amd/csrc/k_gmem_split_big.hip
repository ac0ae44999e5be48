// Prep and finish kernels of the three-kernel pipeline for 128 < N <= 384 (BASELINE configs[3]: 128 streams, N = 256 /
// 384): CH = 4 or 6 row chunks of 64, work matrices and solver scratch in the per-workgroup global workspace, every
// dense step on the matrix core; the Jacobi kernel in between is dort_jacobi_big_kernel.
#include <cstdlib>
#include "dort_ctx.hpp"
#include "dort_active.hpp"

using namespace smrt;

#ifndef SMRT_BIG_FINISH_WAVES
#define SMRT_BIG_FINISH_WAVES 2   // wavefronts per SIMD the finish kernels leave room for (2: <= 256 VGPRs, two workgroups per CU)
#endif

// (finish kernels, MODE 2: <= 256 VGPRs so that two workgroups share a CU -- their LDS is small; the operands of their
// matrix-core passes come from L2, whose latency a second workgroup hides)
template <int NT, int CH, int MODE>
__global__ __launch_bounds__(NT, (MODE == 2 ? SMRT_BIG_FINISH_WAVES : 1)) void dort_passive_big_kernel(DevBatch b, DevStage st, double* workspace, long long ws_stride) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    double* mat = workspace + (long long)blockIdx.x * ws_stride;
    for (long long p = blockIdx.x; p < b.pair_count; p += gridDim.x) {
        dort_pair_passive<NT, CH, MODE>(b, dispatched_pair(b, p), smrt_lds, mat, &st);
        __syncthreads();
    }
}
template <int NT, int CH, int MODE>
__global__ __launch_bounds__(NT, (MODE == 2 ? SMRT_BIG_FINISH_WAVES : 1)) void dort_active_big_kernel(DevBatch b, DevStage st, double* workspace, long long ws_stride) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    double* mat = workspace + (long long)blockIdx.x * ws_stride;
    for (long long p = blockIdx.x; p < b.pair_count; p += gridDim.x) {
        dort_pair_active<NT, CH, MODE>(b, dispatched_pair(b, p), smrt_lds, mat, &st);
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

hipError_t prep_gmem_big(smrt_dort_ctx* ctx, const DevBatch& c, unsigned grid, bool active, int ch) {
    // eight wavefronts here too (one workgroup per CU either way): configs[3] shape 86.1 -> 89.1 solves/s; on the
    // 65 ... 128 path the same change loses 1 %.  SMRT_DORT_BIG_PREP_256=1: the old shape
    static const bool wide = getenv("SMRT_DORT_BIG_PREP_256") == nullptr;
    if (wide) {
        if (active) return ch <= 4 ? go(dort_active_big_kernel<512, 4, 1>, ctx, c, grid, ctx->prep_lds_bytes, 512)
                                   : go(dort_active_big_kernel<512, 6, 1>, ctx, c, grid, ctx->prep_lds_bytes, 512);
        return ch <= 4 ? go(dort_passive_big_kernel<512, 4, 1>, ctx, c, grid, ctx->prep_lds_bytes, 512)
                       : go(dort_passive_big_kernel<512, 6, 1>, ctx, c, grid, ctx->prep_lds_bytes, 512);
    }
    if (active) return ch <= 4 ? go(dort_active_big_kernel<256, 4, 1>, ctx, c, grid, ctx->prep_lds_bytes)
                               : go(dort_active_big_kernel<256, 6, 1>, ctx, c, grid, ctx->prep_lds_bytes);
    return ch <= 4 ? go(dort_passive_big_kernel<256, 4, 1>, ctx, c, grid, ctx->prep_lds_bytes)
                   : go(dort_passive_big_kernel<256, 6, 1>, ctx, c, grid, ctx->prep_lds_bytes);
}
hipError_t finish_gmem_big(smrt_dort_ctx* ctx, const DevBatch& c, unsigned grid, bool active, int ch) {
    // one workgroup of eight wavefronts per CU instead of two of four (see finish_gmem in k_gmem_split.hip): configs[3]
    // shape 77 -> 86 solves/s (512 and 1024 pairs), passive 128 streams 1910 -> 2690.  SMRT_DORT_BIG_FINISH_256=1: the old shape
    static const bool wide = getenv("SMRT_DORT_BIG_FINISH_256") == nullptr;
    if (wide) {
        if (active) return ch <= 4 ? go(dort_active_big_kernel<512, 4, 2>, ctx, c, grid, ctx->finish2_lds_bytes, 512)
                                   : go(dort_active_big_kernel<512, 6, 2>, ctx, c, grid, ctx->finish2_lds_bytes, 512);
        return ch <= 4 ? go(dort_passive_big_kernel<512, 4, 2>, ctx, c, grid, ctx->finish2_lds_bytes, 512)
                       : go(dort_passive_big_kernel<512, 6, 2>, ctx, c, grid, ctx->finish2_lds_bytes, 512);
    }
    if (active) return ch <= 4 ? go(dort_active_big_kernel<256, 4, 2>, ctx, c, grid, ctx->finish2_lds_bytes)
                               : go(dort_active_big_kernel<256, 6, 2>, ctx, c, grid, ctx->finish2_lds_bytes);
    return ch <= 4 ? go(dort_passive_big_kernel<256, 4, 2>, ctx, c, grid, ctx->finish2_lds_bytes)
                   : go(dort_passive_big_kernel<256, 6, 2>, ctx, c, grid, ctx->finish2_lds_bytes);
}

}  // namespace smrt_launch
