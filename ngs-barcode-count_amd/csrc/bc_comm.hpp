// bc_comm.hpp -- the transport under the job's ONE cross-GPU exchange.
//
// The reference has no distributed code: it is one process whose workers share one Results map (main.rs:93-120,
// info.rs:661-808).  Across GPUs that map is cut into one private result structure per rank plus a single exchange at
// the end of the job (bc_exchange.hpp).  Everything that exchange sends goes through the two calls below, so that the
// same exchange logic runs over RCCL on xGMI (bc_comm.hip: ncclSend / ncclRecv groups on the engine's stream) and over
// a directory of message files (here) -- the latter serves machines without peer-to-peer access, several ranks on one
// GPU (the 1-GPU test box) and the CPU-only tests of the exchange logic, where the buffers are host memory.
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

namespace bc {

void set_error(const std::string& msg);  // bc_plan.cpp (tests/emu provides its own)

struct Transport {
  int rank = 0, world = 1;
  virtual ~Transport() {}
  // Every rank calls with the same shape.  Slice r of `send` (send_bytes[r] bytes, slices back to back in rank order)
  // goes to rank r; `recv` receives, back to back in rank order, the slices the ranks addressed to this one
  // (recv_bytes[r] from rank r -- the caller knows them, from an earlier exchange of the counts if need be).  The
  // buffers live in the memory space of the exchange: device memory for an engine, host memory in the CPU tests.
  // Returns 0 or a BC_ERR_* status (message set).
  virtual int all_to_all_v(const void* send, const uint64_t* send_bytes, void* recv, const uint64_t* recv_bytes) = 0;
  // The same for small control data that is always in host memory (counts, counters, overflow lists).
  virtual int host_all_to_all_v(const void* send, const uint64_t* send_bytes, void* recv, const uint64_t* recv_bytes) = 0;

  // --- built on the two above -------------------------------------------------------------------------------------
  // one u64 per peer each way (what the variable-size exchanges need first)
  int exchange_counts(const uint64_t* to_peer, uint64_t* from_peer) {
    std::vector<uint64_t> eight((size_t)world, 8);
    return host_all_to_all_v(to_peer, eight.data(), from_peer, eight.data());
  }
  // element-wise sum of `n` u64 onto rank root (other ranks' vals are left as they were)
  int reduce_sum_u64(uint64_t* vals, int n, int root) {
    std::vector<uint64_t> sb((size_t)world, 0), rb((size_t)world, 0);
    sb[(size_t)root] = (uint64_t)n * 8;
    if (rank == root) rb.assign((size_t)world, (uint64_t)n * 8);
    std::vector<uint64_t> got(rank == root ? (size_t)world * (size_t)n : 1);
    const int rc = host_all_to_all_v(vals, sb.data(), got.data(), rb.data());
    if (rc) return rc;
    if (rank == root) {
      for (int k = 0; k < n; ++k) {
        uint64_t s = 0;
        for (int r = 0; r < world; ++r) s += got[(size_t)r * n + k];
        vals[k] = s;
      }
    }
    return 0;
  }
  int barrier() {
    std::vector<uint64_t> a((size_t)world, 1), b((size_t)world, 0);
    return exchange_counts(a.data(), b.data());
  }
};

// Messages as files in a directory every rank can reach (/dev/shm or any shared file system): message k from rank a
// to rank b is <dir>/m<k>_<a>_<b>, written under a temporary name and renamed, read and removed by its receiver.  Host
// memory only; bc_comm.hip wraps it with pinned staging for device buffers.
struct HostDirTransport : Transport {
  std::string dir;
  uint64_t seq = 0;
  double timeout_s = 300.0;
  HostDirTransport(const std::string& d, int r, int w) : dir(d) {
    rank = r;
    world = w;
    if (const char* ev = getenv("BC_COMM_TIMEOUT_S")) timeout_s = atof(ev);
  }
  std::string name(uint64_t k, int a, int b) const {
    char buf[96];
    snprintf(buf, sizeof buf, "/m%llu_%d_%d", (unsigned long long)k, a, b);
    return dir + buf;
  }
  static int write_all(int fd, const uint8_t* p, uint64_t n) {
    while (n) {
      const ssize_t w = ::write(fd, p, (size_t)(n > (1u << 30) ? (1u << 30) : n));
      if (w < 0) {
        if (errno == EINTR) continue;
        return -6;
      }
      p += w;
      n -= (uint64_t)w;
    }
    return 0;
  }
  static int read_all(int fd, uint8_t* p, uint64_t n) {
    while (n) {
      const ssize_t r = ::read(fd, p, (size_t)(n > (1u << 30) ? (1u << 30) : n));
      if (r < 0) {
        if (errno == EINTR) continue;
        return -6;
      }
      if (r == 0) return -1;  // shorter than announced
      p += r;
      n -= (uint64_t)r;
    }
    return 0;
  }
  int all_to_all_v(const void* send, const uint64_t* send_bytes, void* recv, const uint64_t* recv_bytes) override {
    return host_all_to_all_v(send, send_bytes, recv, recv_bytes);
  }
  int host_all_to_all_v(const void* send, const uint64_t* send_bytes, void* recv, const uint64_t* recv_bytes) override {
    const uint64_t k = seq++;
    const uint8_t* sp = (const uint8_t*)send;
    uint8_t* rp = (uint8_t*)recv;
    uint64_t s_off = 0;
    std::vector<uint64_t> r_off((size_t)world, 0);
    for (int r = 1; r < world; ++r) r_off[(size_t)r] = r_off[(size_t)r - 1] + recv_bytes[r - 1];
    for (int r = 0; r < world; ++r) {
      const uint64_t n = send_bytes[r];
      if (r == rank) {
        if (n != recv_bytes[r]) {
          set_error("exchange: a rank's slice for itself differs in length from what it expects");
          return -6;
        }
        if (n) memmove(rp + r_off[(size_t)r], sp + s_off, (size_t)n);
      } else if (n) {
        const std::string fin = name(k, rank, r), tmp = fin + ".part";
        const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
        if (fd < 0 || write_all(fd, sp + s_off, n) != 0 || ::close(fd) != 0 || ::rename(tmp.c_str(), fin.c_str()) != 0) {
          set_error("exchange: cannot write " + fin + ": " + strerror(errno));
          return -6;
        }
      }
      s_off += n;
    }
    for (int r = 0; r < world; ++r) {
      if (r == rank || recv_bytes[r] == 0) continue;
      const std::string fin = name(k, r, rank);
      struct timespec t0;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      int fd = -1;
      for (unsigned spin = 0;; ++spin) {
        fd = ::open(fin.c_str(), O_RDONLY);
        if (fd >= 0) break;
        if (errno != ENOENT) {
          set_error("exchange: cannot open " + fin + ": " + strerror(errno));
          return -6;
        }
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_s) {
          set_error("exchange: rank " + std::to_string(r) + " did not deliver message " + std::to_string((unsigned long long)k) +
                    " within the time limit (BC_COMM_TIMEOUT_S)");
          return -6;
        }
        usleep(spin < 200 ? 50 : 1000);
      }
      const int rc = read_all(fd, rp + r_off[(size_t)r], recv_bytes[r]);
      ::close(fd);
      ::unlink(fin.c_str());
      if (rc != 0) {
        set_error("exchange: message " + fin + " is shorter than announced");
        return -6;
      }
    }
    return 0;
  }
};

}  // namespace bc
