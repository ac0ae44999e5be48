// Work estimate per pair for cost-weighted sharding: sum over the layers (and azimuth modes) of N_l^3
// with the actual stream counts, from the first stage of the solve only (layer permittivities -> streams per layer).
#include "dort_ctx.hpp"
#include "dort_device.hpp"

using namespace smrt;

template <int NT>
__global__ __launch_bounds__(NT) void dort_cost_kernel(DevBatch b, double* cost) {
    extern __shared__ __attribute__((aligned(16))) double smrt_lds[];
    const long long p = blockIdx.x;
    const int P = b.mode == 1 ? 3 : 2;
    const LdsPlan plan = make_plan(b.n_max_stream, P, b.Lmax, b.n_theta, 1, 0, 0, 1);
    Lds s = carve(smrt_lds, smrt_lds, plan);
    const int t = tid();
    if (t < 8) s.ints[t] = 0;
    block_sync();
    const long long gp = global_pair(b, p);
    const int fi = (int)(gp / b.S), si = (int)(gp % b.S);
    int L = b.n_layers[si];
    const long long o = (long long)si * b.Lmax;
    const int st = pair_setup<NT>(b, s, b.frequency[fi], L, b.thickness + o, b.frac_volume + o, b.temperature + o, b.p1 + o, b.p2 + o,
                                  b.layer_kind ? b.layer_kind + o : nullptr, gp);
    if (st == ST_OK) L = s.ints[6];
    if (t == 0) {
        double c = 0.0;
        if (st == ST_OK)
            for (int l = 0; l < L; ++l) {
                const double n2 = 2.0 * s.nl[l], n3 = 3.0 * s.nl[l];
                c += n2 * n2 * n2 + (b.mode == 1 ? b.m_max * n3 * n3 * n3 : 0.0);
            }
        cost[p] = c;
    }
}

// see prune_mark_pair (dort_passive.hpp): one wavefront per pair
__global__ __launch_bounds__(64) void dort_prune_mark_kernel(DevBatch b, DevStage stg, int* done) {
    prune_mark_pair(b, stg, (long long)blockIdx.x, done);
}

namespace smrt_launch {
hipError_t prune_mark(smrt_dort_ctx* ctx, const DevBatch& c, int* done_dev) {
    hipLaunchKernelGGL(dort_prune_mark_kernel, dim3((unsigned)c.pair_count), dim3(64), 0, ctx->stream, c, ctx->stage, done_dev);
    return hipGetLastError();
}
hipError_t pair_cost(smrt_dort_ctx* ctx, const DevBatch& d, double* cost_dev) {
    const int P = d.mode == 1 ? 3 : 2;
    const size_t lds = (size_t)make_plan(d.n_max_stream, P, d.Lmax, d.n_theta, 1, 0, 0, 1).total * sizeof(double);
    auto kern = dort_cost_kernel<64>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)d.pair_count), dim3(64), lds, ctx->stream, d, cost_dev);
    return hipGetLastError();
}
}  // namespace smrt_launch
