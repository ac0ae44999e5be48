// Host-side context of libsmrt_dort.so and the launcher interface between its translation units.
// The kernels are instantiated in separate .hip files (k_*.hip) so that they compile in parallel; dort_hip.hip holds the
// C ABI (include/smrt_dort.h), the buffers and the chunk loops and calls the launchers declared here.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

#include "dort_layout.hpp"

namespace smrt { struct PhaseRequest; }

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct smrt_dort_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    DevBuf d_nl, d_thick, d_fv, d_temp, d_p1, d_p2, d_freq, d_theta, d_gl, d_out, d_status, d_layer, d_stream, d_n3, d_stage, d_work, d_stL, d_stB, d_std, d_sts, d_stn, d_sti, d_sub1, d_sub2, d_subT, d_atm, d_pairmap, d_kind, d_phase, d_done, d_hostlayer, d_hostcoeff, d_hoststreams, d_hostphase, d_dispatch, d_regws, d_itfslot, d_itf, d_itfcoh, d_lw, d_ste, d_strot;
    smrt::DevBatch dev{};
    bool uploaded = false;
    int out_stride = 0;
    int nt = 256;
    size_t lds_bytes = 0;
    size_t prep_lds_bytes = 0;
    size_t finish2_lds_bytes = 0;
    size_t finish_reg_lds_bytes = 0;
    size_t prep_wide_lds_bytes = 0;
    bool prep_wide = false;     // 64 < N <= 128, passive: the LDS-resident prep kernel (packed triangles) with eight wavefronts
    size_t finish_strip_lds_bytes = 0, finish_strip4_lds_bytes = 0;
    bool finish_strip4 = false; // ... its four-wavefront instance on the LDS pipeline (N <= 64) instead of the register-resident kernel
    bool finish_strip = false;  // strip finish kernel (passive, 64 < N <= 128, Flat interfaces): one workgroup of eight wavefronts per pair
    bool finish_reg = false;    // register-resident finish kernel (passive, N <= 64, Flat interfaces): one wavefront per pair
    int finish_mode = -1;       // -1: the default choice; 0: never the register-resident finish kernel; 1: whenever supported
    bool finish2 = true;        // two-slot finish kernel (set_pipeline(2) selects the LDS-resident one)
    float last_ms = 0.f;
    double total_ms = 0.0;
    int64_t n_launch = 0;
    bool timing_pending = false;
    // per-kernel HIP-event time of the three-kernel pipelines (smrt_dort_kernel_breakdown: events around every prep / Jacobi /
    // finish launch of the NEXT smrt_dort_launch; summed per kind when it is read)
    bool breakdown_on = false;
    std::vector<hipEvent_t> bd_events;   // pool, reused
    std::vector<int> bd_kind;            // per recorded interval [2 k, 2 k + 1]: 0 prep, 1 Jacobi, 2 finish
    size_t bd_used = 0;
    int max_lds = 0;
    bool split = true;          // three-kernel pipeline on the LDS path (fused single kernel if false)
    long long chunk_pairs = 0;  // pairs per pipeline pass (bounds the staging area)
    // Concurrent pipeline passes (LDS pipelines): chunk c runs prep -> Jacobi -> finish on lane stream c % lanes with its own
    // staging region, so that the kernels of different chunks overlap on the chip (the prep, Jacobi and finish kernels
    // stall on different resources and each launch has a tail); joined into `stream` before the launch returns.
    int lanes = 1;
    hipStream_t lane_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lane_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr;
    size_t jacobi_lds = 0;
    int diag_mode = -1;         // smrt_dort_set_diagonalisation: -1 default, 0 Jacobi, 1 symmetric eigensolver where built
    bool eig = false;           // this batch: the symmetric eigensolver (k_eig.hip) between prep and finish
    size_t jacobi16_lds = 0;    // 64 < N <= 128: LDS of the sixteen-wavefront Jacobi kernel (sixteen lanes per column pair), 0: it does not fit
    smrt::DevStage stage{};
    bool gmem_path = false;
    bool gmem_split = false;    // 64 < N: three-kernel pipeline on the global workspace
    bool big = false;           // ... with the kernels for N > 128 (matrix larger than LDS in the Jacobi kernel)
    int jac_in_lds = 0;
    bool active = false;
    int gmem_grid = 0;
    long long ws_stride = 0;
    int nmax_rows = 0;
    // multi-GPU (dort_comm.hip): the RCCL communicator this context is a rank of, and the root's gather buffers
    void* comm = nullptr;
    int comm_world = 0, comm_rank = 0;
    DevBuf d_gather_out, d_gather_status, d_scalar;
};

#ifndef SMRT_JACOBI_NT
#define SMRT_JACOBI_NT 256   // threads per workgroup of the Jacobi kernel
#endif

// Kernel launchers, one translation unit each (asynchronous on ctx->stream; the returned error is the launch error).
// `c` is the DevBatch of one chunk (pair_begin / pair_count / output pointers already offset).
namespace smrt_launch {
// k_split_passive.hip: prep and finish kernels of the LDS-resident pipeline (N <= 64), nt = 64 or 256
hipError_t prep(smrt_dort_ctx* ctx, const smrt::DevBatch& c, int nt);
hipError_t prep_wide(smrt_dort_ctx* ctx, const smrt::DevBatch& c);   // the same for 64 < N <= 128: 512 threads, one workgroup per CU
hipError_t finish(smrt_dort_ctx* ctx, const smrt::DevBatch& c, int nt, bool two_slot);
void occupancy_report(smrt_dort_ctx* ctx, int nt);
// k_finish_reg.hip: the register-resident finish kernel of the same pipeline, one wavefront per pair
hipError_t finish_reg(smrt_dort_ctx* ctx, const smrt::DevBatch& c);
// k_finish_strip.hip: the strip finish kernel of the 64 < N <= 128 pipeline (passive), one workgroup of eight wavefronts per pair
hipError_t finish_strip(smrt_dort_ctx* ctx, const smrt::DevBatch& c);
hipError_t finish_strip4(smrt_dort_ctx* ctx, const smrt::DevBatch& c);   // N <= 64: four wavefronts per pair
// k_jacobi.hip: one workgroup per staging item (pair, [azimuth mode,] layer)
hipError_t jacobi(smrt_dort_ctx* ctx, const smrt::DevBatch& c, long long items);
// k_eig.hip: the symmetric eigensolver on the same items (N <= 64): tridiag, chase, vectors
hipError_t eig(smrt_dort_ctx* ctx, const smrt::DevBatch& c, long long items);
// k_rayleigh.hip: the layers with a Rayleigh phase matrix in closed form (passive; the other kernels skip them)
hipError_t rayleigh(smrt_dort_ctx* ctx, const smrt::DevBatch& c, long long items);
// k_split_active.hip
hipError_t active_prep(smrt_dort_ctx* ctx, const smrt::DevBatch& c, int nt);
hipError_t active_finish(smrt_dort_ctx* ctx, const smrt::DevBatch& c, int nt);
// k_gmem_split.hip: 64 < N <= 128, work matrices in the per-workgroup global workspace, grid-stride over the pairs
hipError_t prep_gmem(smrt_dort_ctx* ctx, const smrt::DevBatch& c, unsigned grid, bool active);
hipError_t finish_gmem(smrt_dort_ctx* ctx, const smrt::DevBatch& c, unsigned grid, bool active);
// k_gmem_split_big.hip: the same for 128 < N <= 384 (ch = 4 or 6 row chunks of 64), k_jacobi_big.hip: its Jacobi kernel
hipError_t prep_gmem_big(smrt_dort_ctx* ctx, const smrt::DevBatch& c, unsigned grid, bool active, int ch);
hipError_t finish_gmem_big(smrt_dort_ctx* ctx, const smrt::DevBatch& c, unsigned grid, bool active, int ch);
hipError_t jacobi_big(smrt_dort_ctx* ctx, const smrt::DevBatch& c, long long items);
// k_fused.hip / k_gmem_fused.hip: everything of a pair in one workgroup
hipError_t fused(smrt_dort_ctx* ctx, const smrt::DevBatch& d, int nt, bool active);
hipError_t fused_gmem(smrt_dort_ctx* ctx, const smrt::DevBatch& d, int ch, bool active);
// k_phase.hip: ft_even_phase of one layer (emmodel protocol)
hipError_t ft_even_phase(smrt_dort_ctx* ctx, const smrt::PhaseRequest& q);
// k_cost.hip: which pairs have reached their prune_deep_snowpack cut within the layers processed so far
hipError_t prune_mark(smrt_dort_ctx* ctx, const smrt::DevBatch& c, int* done_dev);
// k_cost.hip: sum of N_l^3 per pair from the stream counts alone
hipError_t pair_cost(smrt_dort_ctx* ctx, const smrt::DevBatch& d, double* cost_dev);
}  // namespace smrt_launch
