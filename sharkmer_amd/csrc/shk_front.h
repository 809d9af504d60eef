// shk_front.h — internals shared by the plain-C++ host files (shk_front.cpp, shk_inflate.cpp) and shk_host.hip.
// Nothing here is part of the C ABI (include/shk.h is).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace shk {

std::string fmt(const char *f, ...) __attribute__((format(printf, 1, 2)));

// `b as char` of the reference's "Invalid character '{}'" (encoding.rs:353-356): byte b is the scalar U+00b, so a byte
// ≥ 0x80 prints as two UTF-8 bytes
inline std::string byte_as_char(uint8_t b) {
  if (b < 0x80) return std::string(1, (char)b);
  const char two[2] = {(char)(0xC0 | (b >> 6)), (char)(0x80 | (b & 0x3F))};
  return std::string(two, 2);
}

// CPUs this process may really keep busy: the hardware's, the affinity mask's, and the container's CFS quota
// (cgroup v2 cpu.max / v1 cpu.cfs_quota_us) — whichever is smallest.
uint32_t usable_cpus();

// the message of the last failed shk_run_files / shk_validate_args / shk_pack_reads on this thread (shk_run_error)
std::string &run_error();

// A small persistent pool: parallel_for over [0, n), one parallel_for at a time.
struct Pool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::function<void(uint32_t)> job;
  uint32_t n_jobs = 0, next = 0, running = 0;
  bool quit = false;
  std::mutex use;
  explicit Pool(uint32_t T);
  ~Pool();
  void run();
  void parallel_for(uint32_t n, std::function<void(uint32_t)> f);
  uint32_t size() const { return (uint32_t)th.size(); }
};

}  // namespace shk
