// RcclRendezvous.h — how the ranks of a CLI run hand the 128-byte RCCL unique id from rank 0 to the others.
// The CLI (bench_test/bench_micro24.cpp) has no channel between its processes but the file system.  The file is unique to the
// RUN, not to a time window: its name carries the launcher's identity — TORCHELASTIC_RUN_ID when the launcher exports one, the
// rendezvous endpoint (MASTER_ADDR, MASTER_PORT) and the parent process id, which all ranks of one launcher share and two
// launches do not — so a second run on the same (static) port never reads the first run's id, however close in time, and a
// rank that arrives minutes after rank 0 published still accepts the file.  Rank 0 removes a leftover of that name before it
// builds its op, publishes atomically (write + rename), and removes the file once the communicator exists (every rank has
// read it by then: ncclCommInitRank is collective).  A restarted attempt of the SAME launcher (torchelastic: same agent pid, same run id)
// gets a name of its own through TORCHELASTIC_RESTART_COUNT, so a non-zero rank of attempt k+1 cannot pick up the id file that a dead
// attempt k left behind before rank 0 has removed it.  The file is created exclusively, without following symlinks, mode 0600.
#ifndef HOMULATOR_RCCL_RENDEZVOUS_H
#define HOMULATOR_RCCL_RENDEZVOUS_H
#include <fcntl.h>
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
  const char *run = getenv("TORCHELASTIC_RUN_ID"), *addr = getenv("MASTER_ADDR"), *port = getenv("MASTER_PORT"), *restart = getenv("TORCHELASTIC_RESTART_COUNT");
  return "/tmp/homulator_rccl_" + sanitize(addr ? addr : "local") + "_" + sanitize(port ? port : "0") + "_" + sanitize(run ? run : "norun") + "_r" +
         sanitize(restart ? restart : "0") + "_" + std::to_string((long)getppid()) + ".id";
}
inline void removeStale(const std::string &path) { (void)unlink(path.c_str()); (void)unlink((path + ".tmp").c_str()); }
inline void publish(const std::string &path, const char *id) {
  const std::string tmp = path + ".tmp";
  {  // exclusive create (removeStale ran before), never through a symlink someone else planted at the fixed /tmp name
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) throw std::runtime_error("cannot create " + tmp);
    const ssize_t w = write(fd, id, kIdBytes);
    const int c = close(fd);
    if (w != (ssize_t)kIdBytes || c) { (void)unlink(tmp.c_str()); throw std::runtime_error("cannot write " + tmp); }
  }
  if (rename(tmp.c_str(), path.c_str())) throw std::runtime_error("cannot publish the RCCL id at " + path);
}
// waits until the file exists with its full size (rename is atomic: a visible file is complete)
inline void fetch(const std::string &path, char *id, unsigned timeoutMs) {
  for (unsigned waited = 0;; waited += 50) {
    struct stat st;
    if (lstat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_size == (off_t)kIdBytes) {
      const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
      if (fd >= 0) {
        const ssize_t r = read(fd, id, kIdBytes);
        (void)close(fd);
        if (r == (ssize_t)kIdBytes) return;
      }
    }
    if (waited >= timeoutMs) throw std::runtime_error("no RCCL id from rank 0 at " + path + " after " + std::to_string(timeoutMs / 1000) + " s");
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
  }
}
}  // namespace hrv
#endif
