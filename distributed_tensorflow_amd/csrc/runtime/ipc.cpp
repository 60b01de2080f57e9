// Peer memory over HIP IPC: a rank exports the allocation behind a device pointer, every other
// rank of the node maps it, and the hand-written collectives (csrc/kernels/p2p.hip) then read
// peer buffers directly over xGMI with plain loads. Also fine-grained flag arrays for the
// device-side barrier, and interprocess events.
//
// Return codes: 0 ok, > 0 hipError_t, -1 bad argument.
#include <hip/hip_runtime.h>

#include <cstring>
#include <map>
#include <mutex>
#include <string>

#define DTF_API extern "C" __attribute__((visibility("default")))

namespace {

struct Opened {
  void* base = nullptr;
  int refs = 0;
};

std::mutex g_mu;
std::map<std::string, Opened> g_opened;   // handle bytes -> mapping (a handle may be opened once per process)

}  // namespace

DTF_API int dtf_ipc_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }
DTF_API int dtf_ipc_event_handle_bytes() { return (int)sizeof(hipIpcEventHandle_t); }

// Exports the allocation that contains `ptr`. *offset is ptr's distance from the allocation base;
// the importer adds it back.
DTF_API int dtf_ipc_export(void* ptr, void* handle_out, unsigned long long* offset, unsigned long long* alloc_bytes) {
  if (!ptr || !handle_out || !offset) return -1;
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr);
  if (e != hipSuccess) return (int)e;
  hipIpcMemHandle_t h;
  e = hipIpcGetMemHandle(&h, base);
  if (e != hipSuccess) return (int)e;
  std::memcpy(handle_out, &h, sizeof(h));
  *offset = (unsigned long long)((char*)ptr - (char*)base);
  if (alloc_bytes) *alloc_bytes = size;
  return 0;
}

// Maps a peer's allocation; *ptr_out = mapped base + offset.
DTF_API int dtf_ipc_open(const void* handle, unsigned long long offset, void** ptr_out) {
  if (!handle || !ptr_out) return -1;
  const std::string key((const char*)handle, sizeof(hipIpcMemHandle_t));
  std::lock_guard<std::mutex> lk(g_mu);
  Opened& o = g_opened[key];
  if (!o.base) {
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof(h));
    void* base = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { g_opened.erase(key); return (int)e; }
    o.base = base;
  }
  ++o.refs;
  *ptr_out = (char*)o.base + offset;
  return 0;
}

DTF_API int dtf_ipc_close(const void* handle) {
  if (!handle) return -1;
  const std::string key((const char*)handle, sizeof(hipIpcMemHandle_t));
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_opened.find(key);
  if (it == g_opened.end()) return -1;
  if (--it->second.refs > 0) return 0;
  const hipError_t e = hipIpcCloseMemHandle(it->second.base);
  g_opened.erase(it);
  return (int)e;
}

// Fine-grained (coherent across agents) device memory for barrier flags; zero-filled. There is no fallback to
// ordinary (coarse-grained) memory: writes of another process to such memory are not guaranteed to become
// visible while a kernel runs, so every cross-process barrier on it would time out.
DTF_API int dtf_ipc_alloc_flags(unsigned long long bytes, void** out) {
  if (!out || bytes == 0) return -1;
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return (int)e;
  }
  e = hipMemset(p, 0, bytes);
  if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
  e = hipDeviceSynchronize();
  *out = p;
  return (int)e;
}

DTF_API int dtf_ipc_free(void* p) { return p ? (int)hipFree(p) : -1; }

// Reads `bytes` from device memory into host memory (status words, tests).
DTF_API int dtf_ipc_read(void* host_dst, const void* dev_src, unsigned long long bytes) {
  if (!host_dst || !dev_src) return -1;
  return (int)hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost);
}

// ---- interprocess events: stream-ordered hand-offs between processes without a host round trip
DTF_API int dtf_ipc_event_create(void** ev_out, void* handle_out) {
  if (!ev_out || !handle_out) return -1;
  hipEvent_t ev;
  hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess);
  if (e != hipSuccess) return (int)e;
  hipIpcEventHandle_t h;
  e = hipIpcGetEventHandle(&h, ev);
  if (e != hipSuccess) { (void)hipEventDestroy(ev); return (int)e; }
  std::memcpy(handle_out, &h, sizeof(h));
  *ev_out = ev;
  return 0;
}

DTF_API int dtf_ipc_event_open(const void* handle, void** ev_out) {
  if (!handle || !ev_out) return -1;
  hipIpcEventHandle_t h;
  std::memcpy(&h, handle, sizeof(h));
  hipEvent_t ev;
  const hipError_t e = hipIpcOpenEventHandle(&ev, h);
  if (e != hipSuccess) return (int)e;
  *ev_out = ev;
  return 0;
}

DTF_API int dtf_ipc_event_record(void* ev, hipStream_t stream) { return ev ? (int)hipEventRecord((hipEvent_t)ev, stream) : -1; }
DTF_API int dtf_ipc_event_wait(void* ev, hipStream_t stream) { return ev ? (int)hipStreamWaitEvent(stream, (hipEvent_t)ev, 0) : -1; }
DTF_API int dtf_ipc_event_destroy(void* ev) { return ev ? (int)hipEventDestroy((hipEvent_t)ev) : -1; }
