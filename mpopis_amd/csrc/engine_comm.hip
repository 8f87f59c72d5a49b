// engine_comm.hip -- the ONE collective of the path, behind the C ABI.
//
// Trials shard over GPUs at trial granularity (trial k -> rank (k-1) mod G, no data-path collective).  When all
// trials are done, the per-trial summary records (the 12 quantities printed at src/examples/car_example.jl:144-155,
// :287-302) travel to rank 0 with ONE RCCL gather over xGMI; rank 0 computes AVE/STD/MED/L95/U95/MIN/MAX
// (:328-410).  The payload is a few hundred bytes per rank, so the collective is latency-bound and bandwidth is
// irrelevant -- it exists so that a Julia host (no torch.distributed) gets its summary table.
//
// RCCL is bound at run time (dlopen) so that single-GPU users of libmpopis_hip.so never load it: a process that
// already has a librccl mapped (e.g. one that imported torch) reuses that copy.
#include "engine.h"
#include "engine_handle.h"
#include <dlfcn.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

namespace {

// the subset of <rccl/rccl.h> that is used (signatures as declared there: :187,:220,:260,:678,:745,:145)
struct RcclUniqueId { char internal[128]; };
typedef void* RcclComm;
enum { kRcclSuccess = 0, kRcclFloat64 = 8 };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*CommCount)(const RcclComm, int*) = nullptr;
    int (*Gather)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;       // RCCL extension
    int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
std::string g_rccl_error;

bool load_rccl() {
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;   // already mapped by the host process?
    if (!lib) for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) { g_rccl_error = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?"); return false; }
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = (int (*)(RcclUniqueId*))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(RcclComm*, int, RcclUniqueId, int))dlsym(lib, "ncclCommInitRank");
    a.CommDestroy = (int (*)(RcclComm))dlsym(lib, "ncclCommDestroy");
    a.CommCount = (int (*)(const RcclComm, int*))dlsym(lib, "ncclCommCount");
    a.Gather = (int (*)(const void*, void*, size_t, int, int, RcclComm, hipStream_t))dlsym(lib, "ncclGather");
    a.AllGather = (int (*)(const void*, void*, size_t, int, RcclComm, hipStream_t))dlsym(lib, "ncclAllGather");
    a.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather) { g_rccl_error = "librccl lacks the expected symbols"; return false; }
    g_rccl = a;
    return true;
}

std::string rccl_err(const char* what, int rc) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s failed: %s (%d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "rccl error", rc);
    return buf;
}

}  // namespace

extern "C" {

int mpopis_comm_unique_id(char* id128) {
    if (!id128) return MPOPIS_ERR_ARG;
    if (!load_rccl()) return MPOPIS_ERR_HIP;
    RcclUniqueId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != kRcclSuccess) { g_rccl_error = rccl_err("ncclGetUniqueId", rc); return MPOPIS_ERR_HIP; }
    memcpy(id128, id.internal, MPOPIS_COMM_ID_BYTES);
    return MPOPIS_OK;
}

int mpopis_comm_init(mpopis_handle* h, const char* id128, int32_t rank, int32_t world) {
    if (!h || world < 1 || rank < 0 || rank >= world) { if (h) h->err = "mpopis_comm_init: need 0 <= rank < world"; return MPOPIS_ERR_ARG; }
    if (h->comm) { h->err = "communicator already initialised"; return MPOPIS_ERR_ARG; }
    h->comm_rank = rank; h->comm_world = world;
    if (world == 1 && !id128) return MPOPIS_OK;          // nothing to talk to: gather degenerates to a copy (no RCCL loaded)
    if (!id128) { h->err = "mpopis_comm_init: unique id required for world > 1"; return MPOPIS_ERR_ARG; }
    if (!load_rccl()) { h->err = g_rccl_error; return MPOPIS_ERR_HIP; }
    if (hipSetDevice(h->cfg.device) != hipSuccess) { h->err = "hipSetDevice failed"; return MPOPIS_ERR_HIP; }
    RcclUniqueId id;
    memcpy(id.internal, id128, MPOPIS_COMM_ID_BYTES);
    RcclComm c = nullptr;
    const int rc = g_rccl.CommInitRank(&c, world, id, rank);
    if (rc != kRcclSuccess) { h->err = rccl_err("ncclCommInitRank", rc); h->comm_world = 1; h->comm_rank = 0; return MPOPIS_ERR_HIP; }
    h->comm = c;
    return MPOPIS_OK;
}

// How many ranks the handle's communicator really spans, asked of RCCL itself (ncclCommCount): 0 = no RCCL communicator is bound
// (world == 1 without an id, or mpopis_comm_init never ran / failed).  A launcher uses it to tell "the gather went over RCCL with N ranks"
// from "it silently ran some other way".
int mpopis_comm_count(mpopis_handle* h, int32_t* ranks) {
    if (!h || !ranks) { if (h) h->err = "mpopis_comm_count: null argument"; return MPOPIS_ERR_ARG; }
    *ranks = 0;
    if (!h->comm) return MPOPIS_OK;
    if (!g_rccl.CommCount) { h->err = "librccl lacks ncclCommCount"; return MPOPIS_ERR_HIP; }
    int n = 0;
    const int rc = g_rccl.CommCount((RcclComm)h->comm, &n);
    if (rc != kRcclSuccess) { h->err = rccl_err("ncclCommCount", rc); return MPOPIS_ERR_HIP; }
    *ranks = (int32_t)n;
    return MPOPIS_OK;
}

int mpopis_comm_destroy(mpopis_handle* h) {
    if (!h) return MPOPIS_ERR_ARG;
    if (h->comm) {
        (void)hipSetDevice(h->cfg.device);
        (void)hipStreamSynchronize(h->stream);
        (void)g_rccl.CommDestroy((RcclComm)h->comm);
        h->comm = nullptr;
    }
    h->comm_rank = 0; h->comm_world = 1;
    return MPOPIS_OK;
}

// Every rank contributes n_local records (rows of MPOPIS_RECORD_LEN doubles) padded to n_max rows, preceded by its row
// count; rank 0 receives world x (1 + n_max*RECORD_LEN) doubles and unpacks them into out[world][n_max][RECORD_LEN] and
// counts[world].  ncclGather when the library has it (RCCL extension), otherwise ncclAllGather (same bytes per link:
// the payload is < 1 kB per rank).
int mpopis_gather_summary(mpopis_handle* h, const double* records, int32_t n_local, int32_t n_max, double* out, int32_t* counts) {
    if (!h || n_local < 0 || n_max < n_local || (n_local > 0 && !records)) { if (h) h->err = "mpopis_gather_summary: bad arguments"; return MPOPIS_ERR_ARG; }
    const int world = h->comm_world, rank = h->comm_rank;
    const size_t RL = MPOPIS_RECORD_LEN, per = 1 + (size_t)n_max * RL;
    if (rank == 0 && (!out || !counts)) { h->err = "mpopis_gather_summary: rank 0 needs out and counts"; return MPOPIS_ERR_ARG; }
    std::vector<double> send(per, 0.0);
    send[0] = (double)n_local;
    if (n_local) memcpy(send.data() + 1, records, sizeof(double) * (size_t)n_local * RL);
    std::vector<double> recv;
    if (world == 1 && !h->comm) {
        recv = send;
    } else {
        if (!h->comm) { h->err = "mpopis_gather_summary: call mpopis_comm_init first"; return MPOPIS_ERR_ARG; }
        if (hipSetDevice(h->cfg.device) != hipSuccess) { h->err = "hipSetDevice failed"; return MPOPIS_ERR_HIP; }
        double *d_send = nullptr, *d_recv = nullptr;
        if (hipMalloc((void**)&d_send, sizeof(double) * per) != hipSuccess || hipMalloc((void**)&d_recv, sizeof(double) * per * world) != hipSuccess) {
            if (d_send) (void)hipFree(d_send);
            h->err = "hipMalloc failed"; return MPOPIS_ERR_HIP;
        }
        int rc = kRcclSuccess; const char* what = "ncclGather";
        hipError_t e = hipMemcpyAsync(d_send, send.data(), sizeof(double) * per, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            if (g_rccl.Gather) rc = g_rccl.Gather(d_send, d_recv, per, kRcclFloat64, 0, (RcclComm)h->comm, h->stream);
            else { what = "ncclAllGather"; rc = g_rccl.AllGather(d_send, d_recv, per, kRcclFloat64, (RcclComm)h->comm, h->stream); }
        }
        if (e == hipSuccess && rc == kRcclSuccess && rank == 0) {
            recv.resize(per * world);
            e = hipMemcpyAsync(recv.data(), d_recv, sizeof(double) * per * world, hipMemcpyDeviceToHost, h->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        (void)hipFree(d_send); (void)hipFree(d_recv);
        if (rc != kRcclSuccess) { h->err = rccl_err(what, rc); return MPOPIS_ERR_HIP; }
        if (e != hipSuccess) { h->err = std::string("gather copy failed: ") + hipGetErrorString(e); return MPOPIS_ERR_HIP; }
    }
    if (rank == 0) {
        for (int r = 0; r < world; ++r) {
            const double* p = recv.data() + (size_t)r * per;
            counts[r] = (int32_t)p[0];
            memcpy(out + (size_t)r * n_max * RL, p + 1, sizeof(double) * (size_t)n_max * RL);
        }
    }
    return MPOPIS_OK;
}

}  // extern "C"
