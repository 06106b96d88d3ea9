// RcclRendezvous.h — how the ranks of a CLI run hand the 128-byte RCCL unique id from rank 0 to the others.
// The CLI (bench_test/bench_micro24.cpp) has no channel between its processes but the file system.  The file is unique to the
// RUN, not to a time window: its name carries the launcher's identity — TORCHELASTIC_RUN_ID when the launcher exports one, the
// rendezvous endpoint (MASTER_ADDR, MASTER_PORT) and the parent process id, which all ranks of one launcher share and two
// launches do not — so a second run on the same (static) port never reads the first run's id, however close in time, and a
// rank that arrives minutes after rank 0 published still accepts the file.  Rank 0 removes a leftover of that name before it
// builds its op, publishes atomically (write + rename), and removes the file once the communicator exists (every rank has
// read it by then: ncclCommInitRank is collective).
#ifndef HOMULATOR_RCCL_RENDEZVOUS_H
#define HOMULATOR_RCCL_RENDEZVOUS_H
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>
#include <thread>

namespace hrv {
static const size_t kIdBytes = 128;

inline std::string sanitize(const char *s) {
  std::string o;
  for (; s && *s; ++s) o += (isalnum((unsigned char)*s) || *s == '-' || *s == '.') ? *s : '_';
  return o;
}
// HOMULATOR_RCCL_ID_FILE overrides the whole name (the caller then owns uniqueness)
inline std::string idPath() {
  if (const char *e = getenv("HOMULATOR_RCCL_ID_FILE")) return e;
  const char *run = getenv("TORCHELASTIC_RUN_ID"), *addr = getenv("MASTER_ADDR"), *port = getenv("MASTER_PORT");
  return "/tmp/homulator_rccl_" + sanitize(addr ? addr : "local") + "_" + sanitize(port ? port : "0") + "_" + sanitize(run ? run : "norun") + "_" +
         std::to_string((long)getppid()) + ".id";
}
inline void removeStale(const std::string &path) { (void)unlink(path.c_str()); (void)unlink((path + ".tmp").c_str()); }
inline void publish(const std::string &path, const char *id) {
  const std::string tmp = path + ".tmp";
  { std::ofstream f(tmp, std::ios::binary | std::ios::trunc); f.write(id, kIdBytes); if (!f) throw std::runtime_error("cannot write " + tmp); }
  if (rename(tmp.c_str(), path.c_str())) throw std::runtime_error("cannot publish the RCCL id at " + path);
}
// waits until the file exists with its full size (rename is atomic: a visible file is complete)
inline void fetch(const std::string &path, char *id, unsigned timeoutMs) {
  for (unsigned waited = 0;; waited += 50) {
    struct stat st;
    if (stat(path.c_str(), &st) == 0 && st.st_size == (off_t)kIdBytes) {
      std::ifstream f(path, std::ios::binary);
      if (f.read(id, kIdBytes)) return;
    }
    if (waited >= timeoutMs) throw std::runtime_error("no RCCL id from rank 0 at " + path + " after " + std::to_string(timeoutMs / 1000) + " s");
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
  }
}
}  // namespace hrv
#endif
