// ptb_comm.hip -- the halo exchange of a sharded merge as a C entry point over RCCL (`ptb_halo_exchange(ncclComm_t, ...)`).
//
// One image over N GPUs (parallel.ShardedTileMerger; no reference counterpart -- the specification is the single-device result of
// inference/tiles.py:321-346): every rank sends the partial sums of the rectangles it does not own to the owners and receives the
// ones it owns, point to point.  All transfers of a rank are posted as ONE RCCL group (ncclGroupStart / ncclSend / ncclRecv /
// ncclGroupEnd): every pair progresses concurrently, each on its own xGMI link, both directions of a link at once.  No collective:
// a ring all-reduce of the 524 MB accumulator would be per-link bound (~6 ms against ~0.3 ms of kernels per rank).
//
// RCCL is bound at run time (dlopen of the librccl the process already has -- torch ships one -- else the system's): the library
// itself has no link-time dependency on it, single-GPU users never load it.
#include <dlfcn.h>

#include <cstring>
#include <string>

#include <rccl/rccl.h>

#include "ptb_common.h"

namespace ptb {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
};

static RcclApi load_rccl() {
    RcclApi a;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) {                      // the copy that is already in the process first (two RCCLs do not mix)
        a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (a.handle) break;
    }
    for (const char* n : names) {
        if (a.handle) break;
        a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!a.handle) return a;
#define PTB_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, name))
    PTB_SYM(get_unique_id, "ncclGetUniqueId");
    PTB_SYM(comm_init_rank, "ncclCommInitRank");
    PTB_SYM(comm_destroy, "ncclCommDestroy");
    PTB_SYM(group_start, "ncclGroupStart");
    PTB_SYM(group_end, "ncclGroupEnd");
    PTB_SYM(send, "ncclSend");
    PTB_SYM(recv, "ncclRecv");
    PTB_SYM(error_string, "ncclGetErrorString");
#undef PTB_SYM
    a.ok = a.get_unique_id && a.comm_init_rank && a.comm_destroy && a.group_start && a.group_end && a.send && a.recv;
    return a;
}

static RcclApi& rccl() {
    static RcclApi api = load_rccl();
    return api;
}

static int rccl_fail(ncclResult_t r, const char* what) {
    const RcclApi& a = rccl();
    set_error_text((std::string(what) + ": " + (a.error_string ? a.error_string(r) : "RCCL error")).c_str());
    return PTB_ELAUNCH;
}

}  // namespace ptb

using namespace ptb;

// 1 when an RCCL library could be bound, 0 otherwise (ptb_last_hip_error has no text for that: there is simply no library).
extern "C" int ptb_rccl_available(void) { return rccl().ok ? 1 : 0; }

// id: 128 bytes (NCCL_UNIQUE_ID_BYTES), produced on ONE rank and handed to all others by whatever channel the job has
// (torch.distributed's store / a broadcast).
extern "C" int ptb_rccl_unique_id(void* id128) {
    if (!id128) return PTB_EINVAL;
    if (!rccl().ok) return PTB_EUNSUPPORTED;
    ncclUniqueId id;
    const ncclResult_t r = rccl().get_unique_id(&id);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return PTB_OK;
}

// Collective over the nranks processes (one per GPU; the HIP current device of the calling thread is the rank's GPU).
extern "C" int ptb_rccl_comm_init(const void* id128, int nranks, int rank, void** comm) {
    if (!id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks) return PTB_EINVAL;
    if (!rccl().ok) return PTB_EUNSUPPORTED;
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    const ncclResult_t r = rccl().comm_init_rank(&c, nranks, id, rank);
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
    *comm = c;
    return PTB_OK;
}

extern "C" int ptb_rccl_comm_destroy(void* comm) {
    if (!comm) return PTB_EINVAL;
    if (!rccl().ok) return PTB_EUNSUPPORTED;
    const ncclResult_t r = rccl().comm_destroy(static_cast<ncclComm_t>(comm));
    return r == ncclSuccess ? PTB_OK : rccl_fail(r, "ncclCommDestroy");
}

// All outgoing and incoming halo rectangles of this rank (packed, contiguous fp32 buffers: ptb_halo_pack fills the outgoing ones,
// ptb_rect_add / ptb_band_plan_finish_rank consume the incoming ones) as one RCCL group on `stream`.  Stream-ordered, returns at once.
extern "C" int ptb_halo_exchange(void* comm, int n_sends, const float* const* send_bufs, const int64_t* send_counts, const int* send_peers,
                                 int n_recvs, float* const* recv_bufs, const int64_t* recv_counts, const int* recv_peers, ptb_stream_t stream) {
    if (!comm || n_sends < 0 || n_recvs < 0 || (n_sends && (!send_bufs || !send_counts || !send_peers)) ||
        (n_recvs && (!recv_bufs || !recv_counts || !recv_peers))) return PTB_EINVAL;
    if (!rccl().ok) return PTB_EUNSUPPORTED;
    if (n_sends + n_recvs == 0) return PTB_OK;
    for (int k = 0; k < n_sends; ++k) if (!send_bufs[k] || send_counts[k] < 0 || send_peers[k] < 0) return PTB_EINVAL;
    for (int k = 0; k < n_recvs; ++k) if (!recv_bufs[k] || recv_counts[k] < 0 || recv_peers[k] < 0) return PTB_EINVAL;
    const RcclApi& a = rccl();
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    hipStream_t s = (hipStream_t)stream;
    ncclResult_t r = a.group_start();
    if (r != ncclSuccess) return rccl_fail(r, "ncclGroupStart");
    ncclResult_t first_bad = ncclSuccess;
    for (int k = 0; k < n_sends && first_bad == ncclSuccess; ++k)
        first_bad = a.send(send_bufs[k], (size_t)send_counts[k], ncclFloat32, send_peers[k], c, s);
    for (int k = 0; k < n_recvs && first_bad == ncclSuccess; ++k)
        first_bad = a.recv(recv_bufs[k], (size_t)recv_counts[k], ncclFloat32, recv_peers[k], c, s);
    r = a.group_end();                                   // (always closed, also after a failed post)
    if (first_bad != ncclSuccess) return rccl_fail(first_bad, "ncclSend / ncclRecv");
    return r == ncclSuccess ? PTB_OK : rccl_fail(r, "ncclGroupEnd");
}
