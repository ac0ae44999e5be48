// The Jacobi kernel shared by all the three-kernel pipelines: one (pair, [azimuth mode,] layer) item per workgroup,
// see dort_jacobi_kernel.hpp.
#include <cstdlib>
#include "dort_ctx.hpp"
#include "dort_device.hpp"

using namespace smrt;

template <int NT>
__global__ __launch_bounds__(NT) void dort_jacobi_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_jacobi_item<NT>(b, st, jacobi_item_of_block(b, (long long)blockIdx.x), smrt_lds);
}

namespace smrt_launch {
template <int NT>
static hipError_t go(smrt_dort_ctx* ctx, const DevBatch& c, long long items) {
    auto kern = dort_jacobi_kernel<NT>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->jacobi_lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)items), dim3(NT), ctx->jacobi_lds, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}
// N <= 64: four workgroups of four wavefronts share a CU.  64 < N <= 128: the 128-column matrix fills the LDS of a CU, so
// the one resident workgroup brings eight wavefronts (16 column blocks, eight block pairs per round) instead of four.
hipError_t jacobi(smrt_dort_ctx* ctx, const DevBatch& c, long long items) {
    static const bool wide = getenv("SMRT_DORT_JACOBI_256") == nullptr;
    return (ctx->nmax_rows > 64 && wide) ? go<512>(ctx, c, items) : go<SMRT_JACOBI_NT>(ctx, c, items);
}
}  // namespace smrt_launch
