// ft_even_phase of one layer on the device (the emmodel protocol for foreign rtsolvers), see dort_phase_kernel.hpp.
#include "dort_ctx.hpp"
#include "dort_phase_kernel.hpp"

using namespace smrt;

__global__ __launch_bounds__(256) void dort_ft_even_phase_kernel(PhaseRequest q) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)q.n_s * q.n_i) return;
    ft_even_phase_entry(q, (int)(idx / q.n_i), (int)(idx % q.n_i));
}

namespace smrt_launch {
hipError_t ft_even_phase(smrt_dort_ctx* ctx, const PhaseRequest& q) {
    const long long n = (long long)q.n_s * q.n_i;
    hipLaunchKernelGGL(dort_ft_even_phase_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, q);
    return hipGetLastError();
}
}  // namespace smrt_launch
