// The Jacobi kernel for N > 128 (matrix in the staging area, two column blocks at a time in LDS), see
// dort_jacobi_big.hpp.
#include "dort_ctx.hpp"
#include "dort_device.hpp"

using namespace smrt;

__global__ __launch_bounds__(SMRT_JACOBI_BIG_NT) void dort_jacobi_big_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_jacobi_big_item<SMRT_JACOBI_BIG_NT>(b, st, jacobi_item_of_block(b, (long long)blockIdx.x), smrt_lds);
}

namespace smrt_launch {
hipError_t jacobi_big(smrt_dort_ctx* ctx, const DevBatch& c, long long items) {
    auto kern = dort_jacobi_big_kernel;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->jacobi_lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)items), dim3(SMRT_JACOBI_BIG_NT), ctx->jacobi_lds, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}
}  // namespace smrt_launch
