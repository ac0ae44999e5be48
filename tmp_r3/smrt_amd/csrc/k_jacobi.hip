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
// A wavefront rotates eight column pairs per step, so an item of N columns occupies ceil(N / 16) of these wavefronts
// (jacobi_waves) and the others only keep the barriers.  Measured and dropped (round 3, headline batch): a second launch
// with 192 threads for the items of at most 48 rows, 60 % of the batch (10.7 + 10.8 ms against 20.8 ms for the one
// launch: the wavefronts of a workgroup are spread evenly over the SIMDs whatever its size -- tools/micro/simd_placement.hip
// -- and the kernel is not bound by the instructions the fourth wavefront would have issued).
hipError_t jacobi(smrt_dort_ctx* ctx, const DevBatch& c, long long items) {
    static const bool wide = getenv("SMRT_DORT_JACOBI_256") == nullptr;
    return (ctx->nmax_rows > 64 && wide) ? go<512>(ctx, c, items) : go<SMRT_JACOBI_NT>(ctx, c, items);
}
}  // namespace smrt_launch
