// The closed-form diagonalisation of the layers with a Rayleigh phase matrix (dort_rayleigh_kernel.hpp): one workgroup of
// two wavefronts per staging item (pair, layer), passive mode; an item of another kind leaves at once.
#include "dort_ctx.hpp"
#include "dort_device.hpp"
#include "dort_rayleigh_kernel.hpp"

using namespace smrt;

__global__ __launch_bounds__(128) void dort_rayleigh_kernel(DevBatch b, DevStage st) {
    __shared__ __attribute__((aligned(16))) double lds[4 * 64 + 9 * 128 + 8];
    dort_rayleigh_item<128>(b, st, jacobi_item_of_block(b, (long long)blockIdx.x), lds);
}

namespace smrt_launch {
hipError_t rayleigh(smrt_dort_ctx* ctx, const DevBatch& c, long long items) {
    hipLaunchKernelGGL(dort_rayleigh_kernel, dim3((unsigned)items), dim3(128), 0, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}
}  // namespace smrt_launch
