// Active mode (backscatter) through the three-kernel pipeline, 3 x streams <= 64: prep and two-slot finish
// (the Jacobi kernel in between is the shared one), see dort_active.hpp.
#include "dort_ctx.hpp"
#include "dort_active.hpp"

using namespace smrt;

template <int NT>
__global__ __launch_bounds__(NT, (NT <= 256 ? 2 : 1)) void dort_active_prep_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_pair_active<NT, 1, 1>(b, dispatched_pair(b, (long long)blockIdx.x), smrt_lds, nullptr, &st);
}
template <int NT>
__global__ __launch_bounds__(NT, (NT <= 256 ? 2 : 1)) void dort_active_finish_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_pair_active<NT, 1, 3>(b, dispatched_pair(b, (long long)blockIdx.x), smrt_lds, nullptr, &st);
}

namespace smrt_launch {

template <class K>
static hipError_t go(K kern, smrt_dort_ctx* ctx, const DevBatch& c, int nt, size_t lds) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)c.pair_count), dim3(nt), lds, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}

hipError_t active_prep(smrt_dort_ctx* ctx, const DevBatch& c, int nt) {
    return nt == 64 ? go(dort_active_prep_kernel<64>, ctx, c, 64, ctx->prep_lds_bytes)
                    : go(dort_active_prep_kernel<256>, ctx, c, 256, ctx->prep_lds_bytes);
}
hipError_t active_finish(smrt_dort_ctx* ctx, const DevBatch& c, int nt) {
    return nt == 64 ? go(dort_active_finish_kernel<64>, ctx, c, 64, ctx->finish2_lds_bytes)
                    : go(dort_active_finish_kernel<256>, ctx, c, 256, ctx->finish2_lds_bytes);
}

}  // namespace smrt_launch
