// Fused kernels on the per-workgroup global workspace: N > 128 (CH = 4: N <= 256, CH = 6: N <= 384; scalar dense steps)
// and the fallback for 64 < N <= 128 when the pipeline is switched off (CH = 2).
#include "dort_ctx.hpp"
#include "dort_active.hpp"

using namespace smrt;

template <int NT, int CH>
__global__ __launch_bounds__(NT) void dort_passive_kernel_gmem(DevBatch b, double* workspace, long long ws_stride) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    double* mat = workspace + (long long)blockIdx.x * ws_stride;
    for (long long p = blockIdx.x; p < b.pair_count; p += gridDim.x) {
        dort_pair_passive<NT, CH>(b, dispatched_pair(b, p), smrt_lds, mat);
        __syncthreads();
    }
}
template <int NT, int CH>
__global__ __launch_bounds__(NT) void dort_active_kernel_gmem(DevBatch b, double* workspace, long long ws_stride) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    double* mat = workspace + (long long)blockIdx.x * ws_stride;
    for (long long p = blockIdx.x; p < b.pair_count; p += gridDim.x) {
        dort_pair_active<NT, CH>(b, dispatched_pair(b, p), smrt_lds, mat);
        __syncthreads();
    }
}

namespace smrt_launch {

template <class K>
static hipError_t go(K kern, smrt_dort_ctx* ctx, const DevBatch& d) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)ctx->gmem_grid), dim3(256), ctx->lds_bytes, ctx->stream, d, (double*)ctx->d_work.p,
                       ctx->ws_stride);
    return hipGetLastError();
}

hipError_t fused_gmem(smrt_dort_ctx* ctx, const DevBatch& d, int ch, bool active) {
    if (active) {
        if (ch <= 2) return go(dort_active_kernel_gmem<256, 2>, ctx, d);
        if (ch <= 4) return go(dort_active_kernel_gmem<256, 4>, ctx, d);
        return go(dort_active_kernel_gmem<256, 6>, ctx, d);
    }
    if (ch <= 2) return go(dort_passive_kernel_gmem<256, 2>, ctx, d);
    if (ch <= 4) return go(dort_passive_kernel_gmem<256, 4>, ctx, d);
    return go(dort_passive_kernel_gmem<256, 6>, ctx, d);
}

}  // namespace smrt_launch
