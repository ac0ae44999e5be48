// Strip finish kernel of the 64 < N <= 128 pipeline (passive): one workgroup of eight wavefronts per pair, see dort_finish_strip.hpp.
// hipcc-flags: -mllvm -disable-machine-licm
// (machine-level loop-invariant code motion hoists the f64 literals and address parts of the whole layer body out of the
// layer loop and spills them: 192 -> 112 / 115 -> 31 spilled registers without it)
#include <cstdio>
#include "dort_ctx.hpp"
#include "dort_device.hpp"
#include "dort_finish_strip.hpp"

using namespace smrt;

// eight wavefronts, two per SIMD: 256 registers each; the LDS of the CU belongs to the one resident workgroup
__global__ __launch_bounds__(512, 2) void dort_finish_strip_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_pair_passive_strip(b, dispatched_pair(b, (long long)blockIdx.x), smrt_lds, st);
}

// the same recursion on four wavefronts for N <= 64 (LDS pipeline): 44 KB of LDS, three workgroups per CU, three wavefronts per SIMD
#ifndef SMRT_STRIP4_WAVES
#define SMRT_STRIP4_WAVES 3
#endif
// (two instances: batches without layers in closed form -- the headline -- run one that has the Cholesky form of the stage only)
__global__ __launch_bounds__(256, SMRT_STRIP4_WAVES) void dort_finish_strip4_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_pair_passive_strip4<false>(b, dispatched_pair(b, (long long)blockIdx.x), smrt_lds, st);
}
__global__ __launch_bounds__(256, SMRT_STRIP4_WAVES) void dort_finish_strip4_direct_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_pair_passive_strip4<true>(b, dispatched_pair(b, (long long)blockIdx.x), smrt_lds, st);
}

namespace smrt_launch {

hipError_t finish_strip4(smrt_dort_ctx* ctx, const DevBatch& c) {
    const size_t lds = ctx->finish_strip4_lds_bytes;
    auto* const kernel = c.rayleigh_direct ? dort_finish_strip4_direct_kernel : dort_finish_strip4_kernel;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3((unsigned)c.pair_count), dim3(256), lds, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}

hipError_t finish_strip(smrt_dort_ctx* ctx, const DevBatch& c) {
    const size_t lds = ctx->finish_strip_lds_bytes;
    hipError_t e = hipFuncSetAttribute((const void*)dort_finish_strip_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dort_finish_strip_kernel, dim3((unsigned)c.pair_count), dim3(512), lds, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}

}  // namespace smrt_launch
