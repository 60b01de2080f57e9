// Framed TCP transport (host runtime).
//
// The reference moves every message as a 4-byte big-endian length followed by
// a pickle, receives with `data += packet` (quadratic copying) and cannot
// carry more than 4 GiB (reference centralized/network.py:4-28). This
// transport keeps the idea -- a length-prefixed frame on a blocking stream
// socket -- and fixes the mechanics:
//
//   frame = | magic u32 | kind u32 | meta_len u32 | reserved u32 | payload_len u64 |  (24 bytes, big endian)
//           | meta (meta_len bytes, JSON) | payload (payload_len raw tensor bytes) |
//
//   * the payload is received straight into a caller-provided buffer (no
//     concatenation, no unpickling: tensors travel as raw bytes);
//   * payloads are 64-bit sized;
//   * every blocking call takes a timeout (poll), so a dead peer surfaces as an
//     error code instead of the reference's endless wait;
//   * connect() retries on ECONNREFUSED the way the reference worker does
//     (reference centralized/client.py:55-62), but bounded.
//
// All functions are extern "C", thread safe per socket, and release nothing
// they did not create. Return codes: >= 0 success, -1 system error (errno
// kept), -2 timeout, -3 peer closed the connection, -4 protocol error.
#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <stdint.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

#define DTF_API extern "C" __attribute__((visibility("default")))

namespace {

constexpr uint32_t kMagic = 0x44544631u;  // "DTF1"

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

uint64_t to_be64(uint64_t v) { return ((uint64_t)htonl((uint32_t)(v & 0xffffffffu)) << 32) | htonl((uint32_t)(v >> 32)); }
uint64_t from_be64(uint64_t v) { return to_be64(v); }

// waits until fd is ready for `events`; deadline < 0 means no timeout
int wait_fd(int fd, short events, double deadline) {
  for (;;) {
    int ms = -1;
    if (deadline >= 0) {
      double left = deadline - now_s();
      if (left <= 0) return -2;
      ms = (int)(left * 1000.0) + 1;
    }
    pollfd p{fd, events, 0};
    int r = poll(&p, 1, ms);
    if (r > 0) return 0;
    if (r == 0) return -2;
    if (errno != EINTR) return -1;
  }
}

int send_all(int fd, const void* buf, uint64_t n, double deadline) {
  const char* p = (const char*)buf;
  while (n > 0) {
    ssize_t k = ::send(fd, p, n > (1u << 30) ? (1u << 30) : (size_t)n, MSG_NOSIGNAL);
    if (k > 0) { p += k; n -= (uint64_t)k; continue; }
    if (k < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) {
      int w = wait_fd(fd, POLLOUT, deadline);
      if (w != 0) return w;
      continue;
    }
    if (k < 0 && errno == EINTR) continue;
    if (k < 0 && (errno == EPIPE || errno == ECONNRESET)) return -3;
    return -1;
  }
  return 0;
}

int recv_all(int fd, void* buf, uint64_t n, double deadline) {
  char* p = (char*)buf;
  while (n > 0) {
    int w = wait_fd(fd, POLLIN, deadline);
    if (w != 0) return w;
    ssize_t k = ::recv(fd, p, n > (1u << 30) ? (1u << 30) : (size_t)n, 0);
    if (k > 0) { p += k; n -= (uint64_t)k; continue; }
    if (k == 0) return -3;
    if (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK) continue;
    if (errno == ECONNRESET) return -3;
    return -1;
  }
  return 0;
}

void tune(int fd) {
  // non-blocking + poll(): every send and receive honours its deadline
  int fl = fcntl(fd, F_GETFL, 0);
  if (fl >= 0) fcntl(fd, F_SETFL, fl | O_NONBLOCK);
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
  int sz = 8 << 20;
  setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &sz, sizeof(sz));
  setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &sz, sizeof(sz));
}

int resolve(const char* addr, int port, sockaddr_in* out) {
  memset(out, 0, sizeof(*out));
  out->sin_family = AF_INET;
  out->sin_port = htons((uint16_t)port);
  if (inet_pton(AF_INET, addr, &out->sin_addr) == 1) return 0;
  if (strcmp(addr, "localhost") == 0) { inet_pton(AF_INET, "127.0.0.1", &out->sin_addr); return 0; }
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  if (getaddrinfo(addr, nullptr, &hints, &res) != 0 || !res) return -1;
  out->sin_addr = ((sockaddr_in*)res->ai_addr)->sin_addr;
  freeaddrinfo(res);
  return 0;
}

struct Header {
  uint32_t magic, kind, meta_len, reserved;
  uint64_t payload_len;
};

}  // namespace

DTF_API int dtf_tcp_listen(const char* addr, int port, int backlog) {
  sockaddr_in sa;
  if (resolve(addr, port, &sa) != 0) return -1;
  int fd = socket(AF_INET, SOCK_STREAM, 0);
  if (fd < 0) return -1;
  int one = 1;
  setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
  if (bind(fd, (sockaddr*)&sa, sizeof(sa)) != 0 || listen(fd, backlog) != 0) {
    int e = errno; close(fd); errno = e;
    return -1;
  }
  return fd;
}

// Port the listening socket is bound to (useful with port 0 = pick a free one).
DTF_API int dtf_tcp_bound_port(int fd) {
  sockaddr_in sa;
  socklen_t len = sizeof(sa);
  if (getsockname(fd, (sockaddr*)&sa, &len) != 0) return -1;
  return ntohs(sa.sin_port);
}

DTF_API int dtf_tcp_accept(int listen_fd, double timeout_s) {
  int w = wait_fd(listen_fd, POLLIN, timeout_s < 0 ? -1.0 : now_s() + timeout_s);
  if (w != 0) return w;
  int fd = accept(listen_fd, nullptr, nullptr);
  if (fd < 0) return -1;
  tune(fd);
  return fd;
}

DTF_API int dtf_tcp_connect(const char* addr, int port, double timeout_s, double retry_interval_s) {
  sockaddr_in sa;
  if (resolve(addr, port, &sa) != 0) return -1;
  const double deadline = timeout_s < 0 ? -1.0 : now_s() + timeout_s;
  for (;;) {
    int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    if (connect(fd, (sockaddr*)&sa, sizeof(sa)) == 0) { tune(fd); return fd; }
    int e = errno;
    close(fd);
    if (e != ECONNREFUSED && e != ETIMEDOUT && e != EINTR) { errno = e; return -1; }
    if (deadline >= 0 && now_s() + retry_interval_s > deadline) return -2;
    timespec ts;
    ts.tv_sec = (time_t)retry_interval_s;
    ts.tv_nsec = (long)((retry_interval_s - (double)ts.tv_sec) * 1e9);
    nanosleep(&ts, nullptr);
  }
}

DTF_API int dtf_tcp_send_msg(int fd, uint32_t kind, const void* meta, uint32_t meta_len, const void* payload,
                             uint64_t payload_len, double timeout_s) {
  const double deadline = timeout_s < 0 ? -1.0 : now_s() + timeout_s;
  Header h;
  h.magic = htonl(kMagic); h.kind = htonl(kind); h.meta_len = htonl(meta_len); h.reserved = 0;
  h.payload_len = to_be64(payload_len);
  int r = send_all(fd, &h, sizeof(h), deadline);
  if (r != 0) return r;
  if (meta_len) { r = send_all(fd, meta, meta_len, deadline); if (r != 0) return r; }
  if (payload_len) { r = send_all(fd, payload, payload_len, deadline); if (r != 0) return r; }
  return 0;
}

DTF_API int dtf_tcp_recv_head(int fd, uint32_t* kind, uint32_t* meta_len, uint64_t* payload_len, double timeout_s) {
  const double deadline = timeout_s < 0 ? -1.0 : now_s() + timeout_s;
  Header h;
  int r = recv_all(fd, &h, sizeof(h), deadline);
  if (r != 0) return r;
  if (ntohl(h.magic) != kMagic) return -4;
  *kind = ntohl(h.kind);
  *meta_len = ntohl(h.meta_len);
  *payload_len = from_be64(h.payload_len);
  return 0;
}

DTF_API int dtf_tcp_recv_body(int fd, void* meta, uint32_t meta_len, void* payload, uint64_t payload_len,
                              double timeout_s) {
  const double deadline = timeout_s < 0 ? -1.0 : now_s() + timeout_s;
  int r = 0;
  if (meta_len) { r = recv_all(fd, meta, meta_len, deadline); if (r != 0) return r; }
  if (payload_len) { r = recv_all(fd, payload, payload_len, deadline); if (r != 0) return r; }
  return 0;
}

DTF_API int dtf_tcp_close(int fd) {
  if (fd < 0) return 0;
  shutdown(fd, SHUT_RDWR);
  return close(fd);
}

DTF_API int dtf_last_errno() { return errno; }
