// Direct RCCL communicator: one ncclComm_t per process, created from a unique id that the caller
// distributes over the framed-TCP control plane. Every call enqueues on the hipStream_t it is
// given and returns at once; completion is stream-ordered, so a collective on a bucket overlaps
// with whatever the compute stream is doing. Asynchronous errors (a peer that died, a link that
// went down) are polled with dtf_rccl_async_error and turned into a clean abort by the caller,
// where the reference blocks for ever in barrier.wait() (reference centralized/server.py:95).
//
// Return codes: 0 ok, > 0 ncclResult_t, -1 bad argument.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>

#define DTF_API extern "C" __attribute__((visibility("default")))

namespace {

struct RcclComm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

inline bool dtype_of(int code, ncclDataType_t* out) {
  switch (code) {
    case 0: *out = ncclFloat32; return true;
    case 1: *out = ncclBfloat16; return true;
    case 2: *out = ncclFloat16; return true;
    case 3: *out = ncclInt32; return true;
    case 4: *out = ncclInt64; return true;
    case 5: *out = ncclUint8; return true;
    case 6: *out = ncclFloat64; return true;
    default: return false;
  }
}

inline bool op_of(int code, ncclRedOp_t* out) {
  switch (code) {
    case 0: *out = ncclSum; return true;
    case 1: *out = ncclProd; return true;
    case 2: *out = ncclMax; return true;
    case 3: *out = ncclMin; return true;
    case 4: *out = ncclAvg; return true;
    default: return false;
  }
}

}  // namespace

DTF_API int dtf_rccl_version() {
  int v = 0;
  return ncclGetVersion(&v) == ncclSuccess ? v : -1;
}

DTF_API int dtf_rccl_unique_id_bytes() { return (int)sizeof(ncclUniqueId); }

// Rank 0 calls this and ships the bytes to every other rank.
DTF_API int dtf_rccl_unique_id(void* out, int capacity) {
  if (!out || capacity < (int)sizeof(ncclUniqueId)) return -1;
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return (int)r;
  std::memcpy(out, &id, sizeof(id));
  return 0;
}

// Collective over all ranks of the job. `device` is the HIP device of this process.
DTF_API void* dtf_rccl_init(const void* id_bytes, int rank, int world, int device, int* rc) {
  int dummy;
  if (!rc) rc = &dummy;
  *rc = -1;
  if (!id_bytes || world < 1 || rank < 0 || rank >= world) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof(id));
  RcclComm* c = new (std::nothrow) RcclComm();
  if (!c) return nullptr;
  c->rank = rank; c->world = world; c->device = device;
  const ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  *rc = (int)r;
  if (r != ncclSuccess) { delete c; return nullptr; }
  return c;
}

DTF_API int dtf_rccl_all_reduce(void* h, const void* send, void* recv, unsigned long long count, int dtype, int op,
                                hipStream_t stream) {
  ncclDataType_t dt; ncclRedOp_t ro;
  if (!h || !dtype_of(dtype, &dt) || !op_of(op, &ro)) return -1;
  return (int)ncclAllReduce(send, recv, count, dt, ro, ((RcclComm*)h)->comm, stream);
}

// recv_count elements land on every rank; send holds world * recv_count.
DTF_API int dtf_rccl_reduce_scatter(void* h, const void* send, void* recv, unsigned long long recv_count, int dtype,
                                    int op, hipStream_t stream) {
  ncclDataType_t dt; ncclRedOp_t ro;
  if (!h || !dtype_of(dtype, &dt) || !op_of(op, &ro)) return -1;
  return (int)ncclReduceScatter(send, recv, recv_count, dt, ro, ((RcclComm*)h)->comm, stream);
}

DTF_API int dtf_rccl_all_gather(void* h, const void* send, void* recv, unsigned long long send_count, int dtype,
                                hipStream_t stream) {
  ncclDataType_t dt;
  if (!h || !dtype_of(dtype, &dt)) return -1;
  return (int)ncclAllGather(send, recv, send_count, dt, ((RcclComm*)h)->comm, stream);
}

DTF_API int dtf_rccl_broadcast(void* h, const void* send, void* recv, unsigned long long count, int dtype, int root,
                               hipStream_t stream) {
  ncclDataType_t dt;
  if (!h || !dtype_of(dtype, &dt)) return -1;
  return (int)ncclBroadcast(send, recv, count, dt, root, ((RcclComm*)h)->comm, stream);
}

DTF_API int dtf_rccl_reduce(void* h, const void* send, void* recv, unsigned long long count, int dtype, int op,
                            int root, hipStream_t stream) {
  ncclDataType_t dt; ncclRedOp_t ro;
  if (!h || !dtype_of(dtype, &dt) || !op_of(op, &ro)) return -1;
  return (int)ncclReduce(send, recv, count, dt, ro, root, ((RcclComm*)h)->comm, stream);
}

// count elements go to (and come from) every rank.
DTF_API int dtf_rccl_all_to_all(void* h, const void* send, void* recv, unsigned long long count, int dtype,
                                hipStream_t stream) {
  ncclDataType_t dt;
  if (!h || !dtype_of(dtype, &dt)) return -1;
  return (int)ncclAllToAll(send, recv, count, dt, ((RcclComm*)h)->comm, stream);
}

DTF_API int dtf_rccl_send(void* h, const void* buf, unsigned long long count, int dtype, int peer,
                          hipStream_t stream) {
  ncclDataType_t dt;
  if (!h || !dtype_of(dtype, &dt)) return -1;
  return (int)ncclSend(buf, count, dt, peer, ((RcclComm*)h)->comm, stream);
}

DTF_API int dtf_rccl_recv(void* h, void* buf, unsigned long long count, int dtype, int peer, hipStream_t stream) {
  ncclDataType_t dt;
  if (!h || !dtype_of(dtype, &dt)) return -1;
  return (int)ncclRecv(buf, count, dt, peer, ((RcclComm*)h)->comm, stream);
}

// Sends and receives posted between group_start and group_end are issued as one operation, so the
// transfers to different peers run concurrently, each over its own xGMI link.
DTF_API int dtf_rccl_group_start() { return (int)ncclGroupStart(); }
DTF_API int dtf_rccl_group_end() { return (int)ncclGroupEnd(); }

// *err receives the pending asynchronous error (0 = none).
DTF_API int dtf_rccl_async_error(void* h, int* err) {
  if (!h || !err) return -1;
  ncclResult_t e = ncclSuccess;
  const ncclResult_t r = ncclCommGetAsyncError(((RcclComm*)h)->comm, &e);
  *err = (int)e;
  return (int)r;
}

DTF_API int dtf_rccl_rank(void* h) { return h ? ((RcclComm*)h)->rank : -1; }
DTF_API int dtf_rccl_world(void* h) { return h ? ((RcclComm*)h)->world : -1; }

DTF_API const char* dtf_rccl_error_string(int code) {
  return code < 0 ? "bad argument" : ncclGetErrorString((ncclResult_t)code);
}

// Sub-communicator by ncclCommSplit: every rank of the parent calls this; ranks with the same colour >= 0 form a
// group ordered by key, a rank with colour < 0 takes part in the call and gets no communicator (*out = null, 0).
DTF_API int dtf_rccl_split(void* h, int color, int key, void** out) {
  if (!h || !out) return -1;
  RcclComm* p = (RcclComm*)h;
  ncclComm_t nc = nullptr;
  const ncclResult_t r = ncclCommSplit(p->comm, color < 0 ? NCCL_SPLIT_NOCOLOR : color, key, &nc, nullptr);
  *out = nullptr;
  if (r != ncclSuccess) return (int)r;
  if (color < 0 || nc == nullptr) return 0;
  RcclComm* c = new RcclComm();
  c->comm = nc; c->device = p->device;
  ncclResult_t q = ncclCommUserRank(nc, &c->rank);
  if (q == ncclSuccess) q = ncclCommCount(nc, &c->world);
  if (q != ncclSuccess) { ncclCommDestroy(nc); delete c; return (int)q; }
  *out = c;
  return 0;
}

// Tears the communicator down without waiting for outstanding operations (after a failure).
DTF_API int dtf_rccl_abort(void* h) {
  if (!h) return -1;
  RcclComm* c = (RcclComm*)h;
  const ncclResult_t r = ncclCommAbort(c->comm);
  delete c;
  return (int)r;
}

DTF_API int dtf_rccl_destroy(void* h) {
  if (!h) return -1;
  RcclComm* c = (RcclComm*)h;
  const ncclResult_t r = ncclCommDestroy(c->comm);
  delete c;
  return (int)r;
}
