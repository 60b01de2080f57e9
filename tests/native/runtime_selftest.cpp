// Host-only self test of the native runtime (framed TCP transport, batch loader), meant to be built with
// -fsanitize=address,undefined and with -fsanitize=thread (tests/test_native_sanitizers_cpu.py does both) so that
// the runtime's buffer handling and its thread hand-offs run under the sanitizers without Python in the process.
// Exit code 0 = every check passed.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

extern "C" {
int dtf_tcp_listen(const char* addr, int port, int backlog);
int dtf_tcp_bound_port(int fd);
int dtf_tcp_accept(int listen_fd, double timeout_s);
int dtf_tcp_connect(const char* addr, int port, double timeout_s, double retry_interval_s);
int dtf_tcp_send_msg(int fd, uint32_t kind, const void* meta, uint32_t meta_len, const void* payload,
                     uint64_t payload_len, double timeout_s);
int dtf_tcp_recv_head(int fd, uint32_t* kind, uint32_t* meta_len, uint64_t* payload_len, double timeout_s);
int dtf_tcp_recv_body(int fd, void* meta, uint32_t meta_len, void* payload, uint64_t payload_len, double timeout_s);
int dtf_tcp_close(int fd);
void* dtf_loader_create(const void* x, int64_t n_rows, int64_t x_row_bytes, const void* y, int64_t y_row_bytes,
                        int n_slots, int64_t max_batch, int to_float, int n_threads, int try_pinned);
int dtf_loader_submit(void* h, const int64_t* indices, int64_t count);
int dtf_loader_wait(void* h, int slot, void** x, void** y, int64_t* count);
int dtf_loader_release(void* h, int slot);
void dtf_loader_destroy(void* h);
}

#define CHECK(c)                                                              \
  do {                                                                        \
    if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); exit(1); } \
  } while (0)

static void transport_round_trips() {
  const int lfd = dtf_tcp_listen("127.0.0.1", 0, 8);
  CHECK(lfd >= 0);
  const int port = dtf_tcp_bound_port(lfd);
  CHECK(port > 0);
  const size_t sizes[] = {0, 1, 13, 4096, (1u << 20) + 7};
  std::thread server([&] {
    const int fd = dtf_tcp_accept(lfd, 20.0);
    CHECK(fd >= 0);
    for (size_t n : sizes) {                         // echo every message back with kind + 1
      uint32_t kind = 0, ml = 0;
      uint64_t pl = 0;
      CHECK(dtf_tcp_recv_head(fd, &kind, &ml, &pl, 20.0) == 0);
      CHECK(pl == n);
      std::vector<char> meta(ml ? ml : 1), body(pl ? pl : 1);
      CHECK(dtf_tcp_recv_body(fd, ml ? meta.data() : nullptr, ml, pl ? body.data() : nullptr, pl, 20.0) == 0);
      CHECK(dtf_tcp_send_msg(fd, kind + 1, meta.data(), ml, pl ? body.data() : nullptr, pl, 20.0) == 0);
    }
    // the peer closes: the next read must report a closed connection, not hang or crash
    uint32_t kind = 0, ml = 0;
    uint64_t pl = 0;
    CHECK(dtf_tcp_recv_head(fd, &kind, &ml, &pl, 20.0) != 0);
    dtf_tcp_close(fd);
  });
  const int fd = dtf_tcp_connect("127.0.0.1", port, 20.0, 0.05);
  CHECK(fd >= 0);
  uint32_t k = 7;
  for (size_t n : sizes) {
    std::vector<unsigned char> body(n ? n : 1), back(n ? n : 1);
    for (size_t i = 0; i < n; ++i) body[i] = (unsigned char)(i * 131 + n);
    const char meta[] = "{\"step\": 3}";
    CHECK(dtf_tcp_send_msg(fd, k, meta, sizeof(meta) - 1, n ? body.data() : nullptr, n, 20.0) == 0);
    uint32_t kind = 0, ml = 0;
    uint64_t pl = 0;
    CHECK(dtf_tcp_recv_head(fd, &kind, &ml, &pl, 20.0) == 0);
    CHECK(kind == k + 1 && ml == sizeof(meta) - 1 && pl == n);
    std::vector<char> m2(ml);
    CHECK(dtf_tcp_recv_body(fd, m2.data(), ml, n ? back.data() : nullptr, pl, 20.0) == 0);
    CHECK(memcmp(m2.data(), meta, ml) == 0);
    CHECK(n == 0 || memcmp(body.data(), back.data(), n) == 0);
    ++k;
  }
  // a read with nothing on the wire times out instead of blocking
  uint32_t kind = 0, ml = 0;
  uint64_t pl = 0;
  CHECK(dtf_tcp_recv_head(fd, &kind, &ml, &pl, 0.2) != 0);
  dtf_tcp_close(fd);
  server.join();
  dtf_tcp_close(lfd);
  // connecting to a port nobody listens on gives up after its timeout
  CHECK(dtf_tcp_connect("127.0.0.1", port, 0.3, 0.05) < 0);
}

static void loader_batches(int to_float, int n_threads) {
  const int64_t rows = 257, row_bytes = 49, y_bytes = 4;
  std::vector<uint8_t> x(rows * row_bytes);
  std::vector<int32_t> y(rows);
  for (int64_t i = 0; i < rows * row_bytes; ++i) x[i] = (uint8_t)(i * 7 + 3);
  for (int64_t i = 0; i < rows; ++i) y[i] = (int32_t)(i * 5);
  void* L = dtf_loader_create(x.data(), rows, row_bytes, y.data(), y_bytes, 3, 32, to_float, n_threads, 0);
  CHECK(L != nullptr);
  std::vector<int> in_flight;
  for (int b = 0; b < 40; ++b) {
    std::vector<int64_t> idx(1 + (b * 11) % 32);
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (b * 37 + (int64_t)i * 13) % rows;
    int s = dtf_loader_submit(L, idx.data(), (int64_t)idx.size());
    if (s == -2) {                                   // every slot busy: drain the oldest, then submit again
      void *px, *py;
      int64_t cnt;
      CHECK(dtf_loader_wait(L, in_flight.front(), &px, &py, &cnt) == 0);
      CHECK(dtf_loader_release(L, in_flight.front()) == 0);
      in_flight.erase(in_flight.begin());
      s = dtf_loader_submit(L, idx.data(), (int64_t)idx.size());
    }
    CHECK(s >= 0);
    in_flight.push_back(s);
    void *px, *py;
    int64_t cnt;
    CHECK(dtf_loader_wait(L, s, &px, &py, &cnt) == 0);
    CHECK(cnt == (int64_t)idx.size());
    for (int64_t i = 0; i < cnt; ++i) {
      const uint8_t* src = x.data() + idx[i] * row_bytes;
      for (int64_t k = 0; k < row_bytes; ++k) {
        if (to_float) CHECK(((float*)px)[i * row_bytes + k] == (float)src[k] * (1.0f / 255.0f));
        else CHECK(((uint8_t*)px)[i * row_bytes + k] == src[k]);
      }
      CHECK(((int32_t*)py)[i] == y[idx[i]]);
    }
  }
  CHECK(dtf_loader_submit(L, nullptr, 0) == -1);    // rejected, not dereferenced
  dtf_loader_destroy(L);                            // with batches still marked ready: threads join, memory is freed
}

int main() {
  transport_round_trips();
  loader_batches(0, 1);
  loader_batches(1, 3);
  puts("runtime self test ok");
  return 0;
}
