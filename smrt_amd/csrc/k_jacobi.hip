// The Jacobi kernel shared by all the three-kernel pipelines: one (pair, [azimuth mode,] layer) item per workgroup,
// see dort_jacobi_kernel.hpp.
#include <cstdlib>
#include "dort_ctx.hpp"
#include "dort_device.hpp"

using namespace smrt;

template <int NT, int LO, int HI>
__global__ __launch_bounds__(NT) void dort_jacobi_kernel(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_jacobi_item<NT, LO, HI>(b, st, jacobi_item_of_block(b, (long long)blockIdx.x), smrt_lds);
}

// 64 < N <= 128 with sixteen lanes per column pair: sixteen wavefronts (four per SIMD) on the one matrix a CU's LDS holds
__global__ __launch_bounds__(1024) void dort_jacobi_kernel16(DevBatch b, DevStage st) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    dort_jacobi_item16<1024>(b, st, jacobi_item_of_block(b, (long long)blockIdx.x), smrt_lds);
}

namespace smrt_launch {
static hipError_t go16(smrt_dort_ctx* ctx, const DevBatch& c, long long items, size_t lds) {
    hipError_t e = hipFuncSetAttribute((const void*)dort_jacobi_kernel16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dort_jacobi_kernel16, dim3((unsigned)items), dim3(1024), lds, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}
template <int NT, int LO, int HI>
static hipError_t go(smrt_dort_ctx* ctx, const DevBatch& c, long long items, size_t lds) {
    auto kern = dort_jacobi_kernel<NT, LO, HI>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)items), dim3(NT), lds, ctx->stream, c, ctx->stage);
    return hipGetLastError();
}
// N <= 64: one launch per size class of dort_jacobi_kernel.hpp:jacobi_classes (every launch covers all the items and an
// item of another class leaves at once) -- each class with the registers of its own instantiations and the LDS of its
// own largest item, i.e. with more workgroups per CU than one kernel for all sizes can have.  SMRT_DORT_JACOBI_ONE=1:
// the single launch of the earlier versions (A/B switch).
// 64 < N <= 128: the 128-column matrix fills the LDS of a CU, so the one resident workgroup brings eight wavefronts
// (16 column blocks, eight block pairs per round) instead of four.
// A wavefront rotates eight column pairs per step, so an item of N columns occupies ceil(N / 16) of these wavefronts
// (jacobi_waves) and the others only keep the barriers.  Measured and dropped (round 3, headline batch): a second launch
// with 192 threads for the items of at most 48 rows WITH THE LDS AND THE REGISTERS OF THE LARGE ONES (10.7 + 10.8 ms
// against 20.8 ms for the one launch: the wavefronts of a workgroup are spread evenly over the SIMDs whatever its size
// -- tools/micro/simd_placement.hip -- and the kernel is not bound by the instructions the fourth wavefront would have issued).
hipError_t jacobi(smrt_dort_ctx* ctx, const DevBatch& c, long long items) {
    static const bool wide = getenv("SMRT_DORT_JACOBI_256") == nullptr;
    static const bool one = getenv("SMRT_DORT_JACOBI_ONE") != nullptr;
    // SMRT_DORT_JACOBI_1024=1: sixteen wavefronts with sixteen lanes per column pair.  Measured (round 5, configs[2] shape,
    // profiles/r5_jacobi_n128.txt): 101.9 against 102.3 ms per 1792-pair chunk -- nothing: at N = 128 a step moves 64 J columns
    // of 1 KB through LDS each way, and 64 KB of ds_write_b64 at ~85 B / clk are ~770 of the step's ~1000 cycles whatever the
    // number of lanes that issue them.  Kept as an experiment switch, off by default.
    static const bool sixteen = getenv("SMRT_DORT_JACOBI_1024") != nullptr;
    if (ctx->nmax_rows > 64 && sixteen && ctx->jacobi16_lds > 0) return go16(ctx, c, items, ctx->jacobi16_lds);
    if (ctx->nmax_rows > 64) return wide ? go<512, 0, 128>(ctx, c, items, ctx->jacobi_lds) : go<SMRT_JACOBI_NT, 0, 128>(ctx, c, items, ctx->jacobi_lds);
    if (one) return go<SMRT_JACOBI_NT, 0, 128>(ctx, c, items, ctx->jacobi_lds);
    const int P = c.mode == 1 ? 3 : 2;
    JacobiClass cls[4];
    const int n = jacobi_classes(c.n_max_stream * P, cls);
    for (int i = 0; i < n; ++i) {
        const size_t lds = (size_t)make_jacobi_plan(c.n_max_stream, P, cls[i].hi).total * sizeof(double);
        hipError_t e = cls[i].hi == 32 ? go<jacobi_class_nt(32), 0, 32>(ctx, c, items, lds)
                     : cls[i].hi == 48 ? go<jacobi_class_nt(48), 32, 48>(ctx, c, items, lds)
                     : cls[i].hi == 56 ? go<jacobi_class_nt(56), 48, 56>(ctx, c, items, lds)
                                       : go<jacobi_class_nt(64), 56, 64>(ctx, c, items, lds);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
}  // namespace smrt_launch
