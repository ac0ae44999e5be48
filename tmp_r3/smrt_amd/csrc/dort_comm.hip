// Multi-GPU part of libsmrt_dort.so: the gather of the result rows over RCCL (include/smrt_dort.h).  RCCL is loaded
// lazily with dlopen, so a single-GPU user never pays for it (and the library loads where RCCL is absent).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/smrt_dort.h"
#include "dort_ctx.hpp"
#include "dort_host_common.hpp"

namespace {

struct RcclApi {
    void* handle = nullptr;
    std::string err;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

template <class F>
bool sym(RcclApi& a, F& f, const char* name) {
    f = (F)dlsym(a.handle, name);
    if (!f) a.err = std::string("librccl has no symbol ") + name;
    return f != nullptr;
}

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) { api.err = std::string("cannot load librccl: ") + dlerror(); return; }
        bool ok = sym(api, api.GetUniqueId, "ncclGetUniqueId") && sym(api, api.CommInitRank, "ncclCommInitRank") &&
                  sym(api, api.CommInitAll, "ncclCommInitAll") && sym(api, api.CommDestroy, "ncclCommDestroy") &&
                  sym(api, api.GroupStart, "ncclGroupStart") && sym(api, api.GroupEnd, "ncclGroupEnd") &&
                  sym(api, api.Send, "ncclSend") && sym(api, api.Recv, "ncclRecv") &&
                  sym(api, api.AllReduce, "ncclAllReduce") && sym(api, api.GetErrorString, "ncclGetErrorString");
        if (!ok) { dlclose(api.handle); api.handle = nullptr; }
    });
    return api;
}

}  // namespace

#define COMM_API(ctx_)                                               \
    RcclApi& R = rccl();                                             \
    if (!R.handle) { if (ctx_) (ctx_)->err = R.err; return -1; }
#define NCCLCHK(call)                                                                     \
    do {                                                                                  \
        ncclResult_t r_ = (call);                                                         \
        if (r_ != ncclSuccess) {                                                          \
            ctx->err = std::string(#call) + ": " + R.GetErrorString(r_);                  \
            return -1;                                                                    \
        }                                                                                 \
    } while (0)
#define HIPCHK(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            return -1;                                                                    \
        }                                                                                 \
    } while (0)

extern "C" {

int32_t smrt_dort_comm_unique_id(char* id) {
    if (!id) return -1;
    COMM_API((smrt_dort_ctx*)nullptr);
    static_assert(sizeof(ncclUniqueId) == SMRT_COMM_ID_BYTES, "id size");
    ncclUniqueId u;
    if (R.GetUniqueId(&u) != ncclSuccess) return -1;
    memcpy(id, &u, sizeof(u));
    return 0;
}

int32_t smrt_dort_comm_init(smrt_dort_ctx* ctx, int32_t world, int32_t rank, const char* id) {
    if (!ctx || !id) return -1;
    COMM_API(ctx);
    if (world < 1 || rank < 0 || rank >= world) { ctx->err = "invalid world / rank"; return -1; }
    if (ctx->comm) { ctx->err = "the context already has a communicator"; return -1; }
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    NCCLCHK(R.CommInitRank(&comm, world, u, rank));
    ctx->comm = comm; ctx->comm_world = world; ctx->comm_rank = rank;
    return 0;
}

int32_t smrt_dort_comm_init_all(smrt_dort_ctx** ctxs, int32_t n) {
    if (!ctxs || n < 1) return -1;
    smrt_dort_ctx* ctx = ctxs[0];
    if (!ctx) return -1;
    COMM_API(ctx);
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->comm) { ctx->err = "null context or communicator already set"; return -1; }
        devs[i] = ctxs[i]->device;
    }
    std::vector<ncclComm_t> comms(n);
    NCCLCHK(R.CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) { ctxs[i]->comm = comms[i]; ctxs[i]->comm_world = n; ctxs[i]->comm_rank = i; }
    return 0;
}

int32_t smrt_dort_comm_destroy(smrt_dort_ctx* ctx) {
    if (!ctx) return -1;
    if (!ctx->comm) return 0;
    COMM_API(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    NCCLCHK(R.CommDestroy((ncclComm_t)ctx->comm));
    ctx->comm = nullptr; ctx->comm_world = 0; ctx->comm_rank = 0;
    return 0;
}

int32_t smrt_dort_gather(smrt_dort_ctx* ctx, int32_t root, const int64_t* counts, double* out, int32_t* status) {
    if (!ctx) return -1;
    COMM_API(ctx);
    if (!ctx->comm) { ctx->err = "no communicator: call smrt_dort_comm_init first"; return -1; }
    if (!ctx->uploaded) { ctx->err = "no batch uploaded"; return -1; }
    const int world = ctx->comm_world, rank = ctx->comm_rank;
    if (!counts || root < 0 || root >= world) { ctx->err = "invalid root / counts"; return -1; }
    if (counts[rank] != ctx->dev.pair_count) { ctx->err = "counts[own rank] differs from the uploaded pair count"; return -1; }
    HIPCHK(hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    const size_t stride = (size_t)ctx->out_stride;
    // the transfers come from smrt_dort_gather_plan (host arithmetic, tested with 2..8 ranks on the CPU)
    std::vector<smrt_gather_op> ops((size_t)world);
    int64_t own_off = 0, total = 0;
    const int n_ops = smrt_host::gather_plan(world, root, rank, counts, ops.data(), world, &own_off, &total);
    if (n_ops < 0) { ctx->err = "negative count"; return -1; }
    if (rank != root) {
        NCCLCHK(R.GroupStart());
        for (int k = 0; k < n_ops; ++k) {
            NCCLCHK(R.Send(ctx->dev.out, (size_t)ops[k].rows * stride, ncclDouble, ops[k].peer, comm, ctx->stream));
            NCCLCHK(R.Send(ctx->dev.status, (size_t)ops[k].rows, ncclInt32, ops[k].peer, comm, ctx->stream));
        }
        NCCLCHK(R.GroupEnd());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return 0;
    }
    HIPCHK(ctx->d_gather_out.reserve(sizeof(double) * (size_t)total * stride));
    HIPCHK(ctx->d_gather_status.reserve(sizeof(int32_t) * (size_t)total));
    double* gout = (double*)ctx->d_gather_out.p;
    int32_t* gst = (int32_t*)ctx->d_gather_status.p;
    NCCLCHK(R.GroupStart());
    for (int k = 0; k < n_ops; ++k) {
        NCCLCHK(R.Recv(gout + (size_t)ops[k].offset_rows * stride, (size_t)ops[k].rows * stride, ncclDouble, ops[k].peer, comm, ctx->stream));
        NCCLCHK(R.Recv(gst + ops[k].offset_rows, (size_t)ops[k].rows, ncclInt32, ops[k].peer, comm, ctx->stream));
    }
    NCCLCHK(R.GroupEnd());
    const int64_t off = own_off;
    HIPCHK(hipMemcpyAsync(gout + (size_t)off * stride, ctx->dev.out, sizeof(double) * (size_t)counts[root] * stride,
                          hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(gst + off, ctx->dev.status, sizeof(int32_t) * (size_t)counts[root], hipMemcpyDeviceToDevice, ctx->stream));
    if (out) HIPCHK(hipMemcpyAsync(out, gout, sizeof(double) * (size_t)total * stride, hipMemcpyDeviceToHost, ctx->stream));
    if (status) HIPCHK(hipMemcpyAsync(status, gst, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int32_t smrt_dort_gather_plan(int32_t world, int32_t root, int32_t rank, const int64_t* counts, smrt_gather_op* ops,
                              int32_t capacity, int64_t* own_offset_rows, int64_t* total_rows) {
    return smrt_host::gather_plan(world, root, rank, counts, ops, capacity < 0 ? 0 : capacity, own_offset_rows, total_rows);
}

int32_t smrt_dort_comm_allreduce_max(smrt_dort_ctx* ctx, double* values, int32_t n) {
    if (!ctx) return -1;
    COMM_API(ctx);
    if (!ctx->comm) { ctx->err = "no communicator: call smrt_dort_comm_init first"; return -1; }
    if (n < 0 || (n > 0 && !values)) { ctx->err = "invalid value buffer"; return -1; }
    HIPCHK(hipSetDevice(ctx->device));
    const size_t m = (size_t)(n > 0 ? n : 1);
    HIPCHK(ctx->d_scalar.reserve(sizeof(double) * m));
    double zero = 0.0;
    HIPCHK(hipMemcpyAsync(ctx->d_scalar.p, n > 0 ? values : &zero, sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
    NCCLCHK(R.AllReduce(ctx->d_scalar.p, ctx->d_scalar.p, m, ncclDouble, ncclMax, (ncclComm_t)ctx->comm, ctx->stream));
    if (n > 0) HIPCHK(hipMemcpyAsync(values, ctx->d_scalar.p, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
