// Native batch loader (host runtime): worker threads gather dataset rows by index into
// pinned staging buffers, optionally converting uint8 pixels to float32 in [0, 1], so that
// the training loop only issues one asynchronous host-to-device copy per batch.
//
// It is the host half of the input pipeline that the reference delegates to tf.data
// (`from_tensor_slices -> shard -> batch -> shuffle`, reference initializer.py:43-48); the
// index arithmetic (shard / batch / shuffle order) stays in Python, the byte moving is here.
//
//   h = dtf_loader_create(x, n_rows, x_row_bytes, y, y_row_bytes, n_slots, max_batch, to_float, n_threads)
//   slot = dtf_loader_submit(h, indices, count)       // returns immediately
//   dtf_loader_wait(h, slot, &xp, &yp)                // blocks until that batch is assembled
//   ... copy from xp / yp ...
//   dtf_loader_release(h, slot)
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#define DTF_API extern "C" __attribute__((visibility("default")))

namespace {

struct Slot {
  void* x = nullptr;
  void* y = nullptr;
  bool pinned = false;
  int state = 0;  // 0 free, 1 queued/running, 2 ready
  std::vector<int64_t> idx;
};

struct Loader {
  const uint8_t* x; const uint8_t* y;
  int64_t n_rows, x_row, y_row, max_batch;
  int to_float;
  std::vector<Slot> slots;
  std::deque<int> queue;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> threads;
  bool stop = false;

  int64_t out_row() const { return to_float ? x_row * 4 : x_row; }

  void run() {
    for (;;) {
      int s;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return stop || !queue.empty(); });
        if (stop && queue.empty()) return;
        s = queue.front();
        queue.pop_front();
      }
      Slot& sl = slots[s];
      const int64_t n = (int64_t)sl.idx.size();
      for (int64_t i = 0; i < n; ++i) {
        int64_t r = sl.idx[i];
        if (r < 0 || r >= n_rows) r = 0;
        const uint8_t* src = x + r * x_row;
        if (to_float) {
          float* dst = (float*)sl.x + i * x_row;
          for (int64_t k = 0; k < x_row; ++k) dst[k] = (float)src[k] * (1.0f / 255.0f);
        } else {
          memcpy((uint8_t*)sl.x + i * x_row, src, (size_t)x_row);
        }
        if (y) memcpy((uint8_t*)sl.y + i * y_row, y + r * y_row, (size_t)y_row);
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        sl.state = 2;
      }
      cv_done.notify_all();
    }
  }
};

void* alloc_staging(size_t bytes, bool* pinned) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess && p) { *pinned = true; return p; }
  (void)hipGetLastError();
  *pinned = false;
  if (posix_memalign(&p, 4096, bytes ? bytes : 4096) != 0) return nullptr;
  return p;
}

}  // namespace

DTF_API void* dtf_loader_create(const void* x, int64_t n_rows, int64_t x_row_bytes, const void* y, int64_t y_row_bytes,
                                int n_slots, int64_t max_batch, int to_float, int n_threads, int try_pinned) {
  if (!x || n_rows <= 0 || x_row_bytes <= 0 || n_slots <= 0 || max_batch <= 0) return nullptr;
  Loader* L = new Loader();
  L->x = (const uint8_t*)x; L->y = (const uint8_t*)y;
  L->n_rows = n_rows; L->x_row = x_row_bytes; L->y_row = y ? y_row_bytes : 0; L->max_batch = max_batch;
  L->to_float = to_float;
  L->slots.resize(n_slots);
  for (auto& s : L->slots) {
    const size_t xb = (size_t)(L->out_row() * max_batch), yb = (size_t)(L->y_row * max_batch);
    if (try_pinned) {
      s.x = alloc_staging(xb, &s.pinned);
      bool p2 = false;
      s.y = yb ? alloc_staging(yb, &p2) : nullptr;
      s.pinned = s.pinned && (yb == 0 || p2);
    } else {
      if (posix_memalign(&s.x, 4096, xb ? xb : 4096) != 0) s.x = nullptr;
      if (yb && posix_memalign(&s.y, 4096, yb) != 0) s.y = nullptr;
    }
    if (!s.x || (yb && !s.y)) { delete L; return nullptr; }
  }
  if (n_threads < 1) n_threads = 1;
  for (int i = 0; i < n_threads; ++i) L->threads.emplace_back([L] { L->run(); });
  return L;
}

DTF_API int dtf_loader_is_pinned(void* h) { return h && ((Loader*)h)->slots[0].pinned ? 1 : 0; }

DTF_API int dtf_loader_submit(void* h, const int64_t* indices, int64_t count) {
  Loader* L = (Loader*)h;
  if (!L || count <= 0 || count > L->max_batch) return -1;
  std::unique_lock<std::mutex> lk(L->mu);
  int s = -1;
  for (size_t i = 0; i < L->slots.size(); ++i) if (L->slots[i].state == 0) { s = (int)i; break; }
  if (s < 0) return -2;  // every slot is in flight: wait + release one first
  L->slots[s].idx.assign(indices, indices + count);
  L->slots[s].state = 1;
  L->queue.push_back(s);
  lk.unlock();
  L->cv_work.notify_one();
  return s;
}

DTF_API int dtf_loader_wait(void* h, int slot, void** x, void** y, int64_t* count) {
  Loader* L = (Loader*)h;
  if (!L || slot < 0 || slot >= (int)L->slots.size()) return -1;
  std::unique_lock<std::mutex> lk(L->mu);
  if (L->slots[slot].state == 0) return -1;
  L->cv_done.wait(lk, [&] { return L->slots[slot].state == 2; });
  *x = L->slots[slot].x; *y = L->slots[slot].y; *count = (int64_t)L->slots[slot].idx.size();
  return 0;
}

DTF_API int dtf_loader_release(void* h, int slot) {
  Loader* L = (Loader*)h;
  if (!L || slot < 0 || slot >= (int)L->slots.size()) return -1;
  std::lock_guard<std::mutex> lk(L->mu);
  L->slots[slot].state = 0;
  return 0;
}

DTF_API void dtf_loader_destroy(void* h) {
  Loader* L = (Loader*)h;
  if (!L) return;
  {
    std::lock_guard<std::mutex> lk(L->mu);
    L->stop = true;
  }
  L->cv_work.notify_all();
  for (auto& t : L->threads) t.join();
  for (auto& s : L->slots) {
    if (s.pinned) { if (s.x) (void)hipHostFree(s.x); if (s.y) (void)hipHostFree(s.y); }
    else { free(s.x); free(s.y); }
  }
  delete L;
}
