// Everything of a pair in ONE kernel, one workgroup per pair, matrices in LDS (N <= 64): the shape the pipelines grew
// out of, kept behind smrt_dort_set_pipeline(ctx, 0) as an independent implementation for cross-checks.
#include "dort_ctx.hpp"
#include "dort_active.hpp"

using namespace smrt;

template <int NT>
__global__ __launch_bounds__(NT) void dort_passive_kernel(DevBatch b) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_pair_passive<NT, 1>(b, dispatched_pair(b, (long long)blockIdx.x), smrt_lds);
}
template <int NT>
__global__ __launch_bounds__(NT) void dort_active_kernel(DevBatch b) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_pair_active<NT, 1>(b, dispatched_pair(b, (long long)blockIdx.x), smrt_lds);
}

namespace smrt_launch {

template <class K>
static hipError_t go(K kern, smrt_dort_ctx* ctx, const DevBatch& d, int nt) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)d.pair_count), dim3(nt), ctx->lds_bytes, ctx->stream, d);
    return hipGetLastError();
}

hipError_t fused(smrt_dort_ctx* ctx, const DevBatch& d, int nt, bool active) {
    if (active) return nt == 64 ? go(dort_active_kernel<64>, ctx, d, 64) : go(dort_active_kernel<256>, ctx, d, 256);
    return nt == 64 ? go(dort_passive_kernel<64>, ctx, d, 64) : go(dort_passive_kernel<256>, ctx, d, 256);
}

}  // namespace smrt_launch
