// The Jacobi kernel shared by all the three-kernel pipelines: one (pair, [azimuth mode,] layer) item per workgroup,
// see dort_jacobi_kernel.hpp.
#include "dort_ctx.hpp"
#include "dort_device.hpp"

using namespace smrt;

template <int NT>
__global__ __launch_bounds__(NT) void dort_jacobi_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_jacobi_item<NT>(b, st, jacobi_item_of_block(b, (long long)blockIdx.x), smrt_lds);
}

namespace smrt_launch {
hipError_t jacobi(smrt_dort_ctx* ctx, const DevBatch& c, long long items) {
    auto kern = dort_jacobi_kernel<SMRT_JACOBI_NT>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->jacobi_lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)items), dim3(SMRT_JACOBI_NT), ctx->jacobi_lds, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}
}  // namespace smrt_launch
