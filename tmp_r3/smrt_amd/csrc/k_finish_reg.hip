// Register-resident finish kernel of the LDS pipeline (passive, N <= 64): one wavefront per pair, see dort_finish_reg.hpp.
#include <cstdio>
#include "dort_ctx.hpp"
#include "dort_device.hpp"
#include "dort_finish_reg.hpp"

using namespace smrt;

// one wavefront per workgroup and per SIMD: the whole register file (512 VGPRs: 256 architectural + 256 accumulation)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void dort_finish_reg_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_pair_passive_reg(b, dispatched_pair(b, (long long)blockIdx.x), smrt_lds, st);
}

namespace smrt_launch {

hipError_t finish_reg(smrt_dort_ctx* ctx, const DevBatch& c) {
    const size_t lds = ctx->finish_reg_lds_bytes;
    hipError_t e = hipFuncSetAttribute((const void*)dort_finish_reg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dort_finish_reg_kernel, dim3((unsigned)c.pair_count), dim3(64), lds, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}

}  // namespace smrt_launch
