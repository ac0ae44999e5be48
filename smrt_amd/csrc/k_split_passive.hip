// Kernels of the LDS-resident passive pipeline (streams x polarisations N <= 64): prep and finish, see dort_passive.hpp.
// hipcc-flags: -mllvm -disable-machine-licm
// (fewer loop-invariant values hoisted and spilled: prep kernel 6.15 -> 6.02 ms on the headline batch, profiles/r5_nolicm_ab.txt)
#include <cstdio>
#include "dort_ctx.hpp"
#include "dort_device.hpp"

using namespace smrt;

#ifndef SMRT_PREP_WAVES
#define SMRT_PREP_WAVES 3   // wavefronts per SIMD the prep kernel leaves room for: its LDS plan (two packed lower
                           // triangles, 45 KB at 32 streams) lets three workgroups share a CU
#endif
template <int NT>
__global__ __launch_bounds__(NT, SMRT_PREP_WAVES) void dort_prep_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_pair_passive<NT, 1, 1>(b, dispatched_pair(b, (long long)blockIdx.x), smrt_lds, nullptr, &st);
}
// the same kernel for 64 < N <= 128: the two packed lower triangles of a 64-stream layer (N = 128) are 134 KB, one
// workgroup of eight wavefronts per CU (two per SIMD) instead of the global-workspace prep kernel's two of four
__global__ __launch_bounds__(512, 2) void dort_prep_kernel_wide(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_pair_passive<512, 1, 1>(b, dispatched_pair(b, (long long)blockIdx.x), smrt_lds, nullptr, &st);
}
// four N x N matrices in LDS: one workgroup per CU (kept for A/B runs, smrt_dort_set_pipeline(ctx, 2))
template <int NT>
__global__ __launch_bounds__(NT) void dort_finish_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_pair_passive<NT, 1, 2>(b, dispatched_pair(b, (long long)blockIdx.x), smrt_lds, nullptr, &st);
}
// two LDS slots + F, G in the (dead) staging slots of the item: two workgroups per CU
// (second launch-bound argument on HIP = wavefronts per SIMD the compiler must leave room for: 2 -> <= 256 VGPRs)
template <int NT>
__global__ __launch_bounds__(NT, 2) void dort_finish2_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_pair_passive<NT, 1, 3>(b, dispatched_pair(b, (long long)blockIdx.x), smrt_lds, nullptr, &st);
}

namespace smrt_launch {

template <class K>
static hipError_t go(K kern, smrt_dort_ctx* ctx, const DevBatch& c, int nt, size_t lds) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)c.pair_count), dim3(nt), lds, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}

hipError_t prep_wide(smrt_dort_ctx* ctx, const DevBatch& c) { return go(dort_prep_kernel_wide, ctx, c, 512, ctx->prep_wide_lds_bytes); }

hipError_t prep(smrt_dort_ctx* ctx, const DevBatch& c, int nt) {
    return nt == 64 ? go(dort_prep_kernel<64>, ctx, c, 64, ctx->prep_lds_bytes)
                    : go(dort_prep_kernel<256>, ctx, c, 256, ctx->prep_lds_bytes);
}

hipError_t finish(smrt_dort_ctx* ctx, const DevBatch& c, int nt, bool two_slot) {
    if (two_slot)
        return nt == 64 ? go(dort_finish2_kernel<64>, ctx, c, 64, ctx->finish2_lds_bytes)
                        : go(dort_finish2_kernel<256>, ctx, c, 256, ctx->finish2_lds_bytes);
    return nt == 64 ? go(dort_finish_kernel<64>, ctx, c, 64, ctx->lds_bytes)
                    : go(dort_finish_kernel<256>, ctx, c, 256, ctx->lds_bytes);
}

void occupancy_report(smrt_dort_ctx* ctx, int nt) {   // resident workgroups per CU as the runtime sees them
    int op = 0, of2 = 0, of = 0;
    if (nt == 64) {
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&op, dort_prep_kernel<64>, 64, ctx->prep_lds_bytes);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&of2, dort_finish2_kernel<64>, 64, ctx->finish2_lds_bytes);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&of, dort_finish_kernel<64>, 64, ctx->lds_bytes);
    } else {
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&op, dort_prep_kernel<256>, 256, ctx->prep_lds_bytes);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&of2, dort_finish2_kernel<256>, 256, ctx->finish2_lds_bytes);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&of, dort_finish_kernel<256>, 256, ctx->lds_bytes);
    }
    fprintf(stderr, "occupancy (workgroups/CU), %d threads: prep lds=%zu -> %d | finish2 lds=%zu -> %d | finish lds=%zu -> %d\n",
            nt, ctx->prep_lds_bytes, op, ctx->finish2_lds_bytes, of2, ctx->lds_bytes, of);
}

}  // namespace smrt_launch
