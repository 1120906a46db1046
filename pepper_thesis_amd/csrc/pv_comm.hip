// pv_comm.hip — the one exchange step of the multi-GPU path behind the C-ABI: pv_gather.
//
// Regions shard across ranks with no data-path collective; at the end the per-window prediction rows (and whatever keys the
// caller packs next to them) are gathered to ONE rank. The reference has no counterpart (every caller process writes its own
// prediction file, pepper_variant/modules/python/RunInference.py:101-106; the only process-group site is
// pepper/modules/python/models/predict_distributed_gpu.py:124-129): this is the hand-off a multi-rank launcher uses when it
// wants one result stream.
//   counts : ncclAllGather of one int64 per rank (every rank learns the row counts)
//   payload: grouped ncclSend / ncclRecv to `dst` (point-to-point over xGMI: 7 links per GPU, so the 7 sends of an
//            8-GPU node to rank 0 run on 7 different links), the destination's own rows by a device copy
// RCCL is resolved at run time with dlopen (librccl.so.1, the copy a host that also carries torch.distributed has already
// mapped), so libpepper_hip.so has no link-time dependency on it and single-GPU users never load it.
#include "pv_common.hpp"

#include <dlfcn.h>

#include <mutex>

namespace {

typedef struct { char internal[128]; } nccl_uid;
typedef void* nccl_comm;
enum { NCCL_CHAR = 0, NCCL_INT64 = 4 };

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(nccl_uid*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

char g_dlerr[256] = "symbol missing";

void rccl_open(Rccl& r) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.h) break;
    }
    if (!r.h) {
        const char* e = dlerror();  // one call: a second one returns NULL
        if (e) snprintf(g_dlerr, sizeof g_dlerr, "%s", e);
        return;
    }
#define PV_SYM(field, sym) r.field = (decltype(r.field))dlsym(r.h, sym)
    PV_SYM(GetUniqueId, "ncclGetUniqueId"); PV_SYM(CommInitRank, "ncclCommInitRank"); PV_SYM(CommDestroy, "ncclCommDestroy");
    PV_SYM(AllGather, "ncclAllGather"); PV_SYM(Send, "ncclSend"); PV_SYM(Recv, "ncclRecv");
    PV_SYM(GroupStart, "ncclGroupStart"); PV_SYM(GroupEnd, "ncclGroupEnd"); PV_SYM(GetErrorString, "ncclGetErrorString");
#undef PV_SYM
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) {
        dlclose(r.h);
        r.h = nullptr;
    }
}

// contexts may create communicators from several threads: the library is opened exactly once
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_open(r); });
    return r.h ? &r : nullptr;
}

}  // namespace

struct pv_comm {
    nccl_comm comm = nullptr;
    int rank = 0, world = 1;
    int64_t* d_counts = nullptr;  // [2 * world + 2] pairs {rows, receive capacity (dst only, else -1)}: the last pair is this rank's own
    int64_t* h_counts = nullptr;  // pinned [2 * world + 2]
};

#define PV_NCCL(call)                                                                                                  \
    do {                                                                                                               \
        int e__ = (call);                                                                                              \
        if (e__ != 0) {                                                                                                \
            pv_set_error("%s failed: %s", #call, (R->GetErrorString ? R->GetErrorString(e__) : "RCCL error"));        \
            return PV_ERR_HIP;                                                                                         \
        }                                                                                                              \
    } while (0)

extern "C" int pv_comm_unique_id(pv_ctx* ctx, char* id128) {
    PV_CHECK(ctx && id128, PV_ERR_INVALID, "null argument");
    Rccl* R = rccl();
    PV_CHECK(R, PV_ERR_STATE, "RCCL (librccl.so.1) could not be loaded: %s", g_dlerr);
    PV_HIP(hipSetDevice(ctx->device));
    nccl_uid u;
    PV_NCCL(R->GetUniqueId(&u));
    memcpy(id128, u.internal, 128);
    return PV_OK;
}

extern "C" int pv_comm_create(pv_ctx* ctx, const char* id128, int rank, int world, pv_comm** out) {
    PV_CHECK(ctx && id128 && out && world >= 1 && rank >= 0 && rank < world, PV_ERR_INVALID, "bad argument");
    Rccl* R = rccl();
    PV_CHECK(R, PV_ERR_STATE, "RCCL (librccl.so.1) could not be loaded");
    PV_HIP(hipSetDevice(ctx->device));
    pv_comm* c = new pv_comm();
    c->rank = rank; c->world = world;
    nccl_uid u;
    memcpy(u.internal, id128, 128);
    int e = R->CommInitRank(&c->comm, world, u, rank);
    if (e != 0) {
        pv_set_error("ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(e) : "RCCL error");
        delete c;
        return PV_ERR_HIP;
    }
    if (hipMalloc((void**)&c->d_counts, (size_t)(2 * world + 2) * sizeof(int64_t)) != hipSuccess ||
        hipHostMalloc((void**)&c->h_counts, (size_t)(2 * world + 2) * sizeof(int64_t), hipHostMallocDefault) != hipSuccess) {
        pv_comm_destroy(c);  // frees the communicator and whichever buffer did get allocated
        pv_set_error("pv_comm_create: allocation failed");
        return PV_ERR_HIP;
    }
    *out = c;
    return PV_OK;
}

extern "C" void pv_comm_destroy(pv_comm* c) {
    if (!c) return;
    Rccl* R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    delete c;
}

extern "C" int pv_gather_counts(pv_ctx* ctx, pv_comm* c, int64_t n_rows, int64_t* counts_out, void* stream) {
    PV_CHECK(ctx && c && n_rows >= 0 && counts_out, PV_ERR_INVALID, "bad argument");
    Rccl* R = rccl();
    PV_CHECK(R, PV_ERR_STATE, "RCCL (librccl.so.1) could not be loaded");
    PV_HIP(hipSetDevice(ctx->device));
    hipStream_t st = pv_pick_stream(ctx, stream);
    int64_t* mine = c->h_counts + 2 * c->world;
    mine[0] = n_rows;
    PV_HIP(hipMemcpyAsync(c->d_counts + 2 * c->world, mine, sizeof(int64_t), hipMemcpyHostToDevice, st));
    PV_NCCL(R->AllGather(c->d_counts + 2 * c->world, c->d_counts, 1, NCCL_INT64, c->comm, st));
    PV_HIP(hipMemcpyAsync(c->h_counts, c->d_counts, (size_t)c->world * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    PV_HIP(hipStreamSynchronize(st));
    for (int r = 0; r < c->world; r++) counts_out[r] = c->h_counts[r];
    return PV_OK;
}

extern "C" int pv_gather(pv_ctx* ctx, pv_comm* c, const void* d_send, int64_t n_rows, int row_bytes, void* d_recv,
                         int64_t recv_capacity_rows, int64_t* counts_out, int dst, void* stream) {
    PV_CHECK(ctx && c && n_rows >= 0 && row_bytes > 0 && dst >= 0 && dst < c->world, PV_ERR_INVALID, "bad argument");
    PV_CHECK(n_rows == 0 || d_send, PV_ERR_INVALID, "null send buffer");
    Rccl* R = rccl();
    PV_CHECK(R, PV_ERR_STATE, "RCCL (librccl.so.1) could not be loaded");
    PV_HIP(hipSetDevice(ctx->device));
    hipStream_t st = pv_pick_stream(ctx, stream);
    // 1. every rank learns every row count AND the destination's capacity, so that all ranks take the same decision before
    //    anybody is committed to a send (a destination that bails out alone would leave its peers blocked in ncclSend)
    int64_t* mine = c->h_counts + 2 * c->world;
    mine[0] = n_rows;
    mine[1] = c->rank == dst ? (d_recv ? recv_capacity_rows : (int64_t)0) : (int64_t)-1;
    PV_HIP(hipMemcpyAsync(c->d_counts + 2 * c->world, mine, 2 * sizeof(int64_t), hipMemcpyHostToDevice, st));
    PV_NCCL(R->AllGather(c->d_counts + 2 * c->world, c->d_counts, 2, NCCL_INT64, c->comm, st));
    PV_HIP(hipMemcpyAsync(c->h_counts, c->d_counts, (size_t)(2 * c->world) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    PV_HIP(hipStreamSynchronize(st));
    int64_t total = 0;
    for (int r = 0; r < c->world; r++) {
        if (counts_out) counts_out[r] = c->h_counts[2 * r];
        total += c->h_counts[2 * r];
    }
    const int64_t cap = c->h_counts[2 * dst + 1];
    if (total > cap) {  // the same verdict on every rank: nobody sends, nobody waits
        pv_set_error("pv_gather: %lld rows arrive, the receive buffer of rank %d holds %lld", (long long)total, dst, (long long)cap);
        return PV_ERR_CAPACITY;
    }
    // 2. payload: point-to-point to dst, rank-major. Once the group is open it is always closed again: a failing call inside
    //    is remembered, the remaining calls are skipped and GroupEnd still runs.
    PV_NCCL(R->GroupStart());
    int bad = 0;
    const char* what = "";
    if (c->rank == dst) {
        int64_t off = 0;
        for (int r = 0; r < c->world && !bad; r++) {
            const size_t nb = (size_t)c->h_counts[2 * r] * (size_t)row_bytes;
            char* to = (char*)d_recv + (size_t)off * (size_t)row_bytes;
            if (r == dst) {
                if (nb && hipMemcpyAsync(to, d_send, nb, hipMemcpyDeviceToDevice, st) != hipSuccess) { bad = -1; what = "hipMemcpyAsync"; }
            } else if (nb) {
                bad = R->Recv(to, nb, NCCL_CHAR, r, c->comm, st);
                what = "ncclRecv";
            }
            off += c->h_counts[2 * r];
        }
    } else if (n_rows) {
        bad = R->Send(d_send, (size_t)n_rows * (size_t)row_bytes, NCCL_CHAR, dst, c->comm, st);
        what = "ncclSend";
    }
    const int end = R->GroupEnd();
    if (bad) {
        pv_set_error("pv_gather: %s failed: %s", what, bad > 0 && R->GetErrorString ? R->GetErrorString(bad) : "HIP error");
        return PV_ERR_HIP;
    }
    PV_NCCL(end);
    return PV_OK;
}
