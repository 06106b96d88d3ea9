// mock_rccl.cpp — TEST DOUBLE of the eight RCCL entry points the backend uses (homulator_amd/csrc/hm_backend.hip: rccl_load).
// RCCL refuses two ranks on one device, and the test box has one GPU: with this library (selected by HOMULATOR_RCCL_LIB, never
// by default) the ranks are THREADS of one process, every rank with its own hm_ctx / HIP stream / HBM pool, and the backend's
// RCCL code path — hm_comm_unique_id, the collective hm_comm_init_rccl, the grouped ncclSend / ncclRecv of every exchange with
// their counts, datatype, peers and streams — executes exactly as on a node.  Only the wire differs: a send is matched with
// the peer's receive by a rendezvous of the threads and copied device to device.
// What a real run would turn into a HANG is an ERROR here: a receive nobody sends to, a send nobody receives, a size the two
// sides disagree on, a peer out of range, ranks entering a different number of groups (rendezvous time-out).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Post { int peer; const void *ptr; size_t bytes; bool taken; };
struct Group {
  int world = 0, attached = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long generation = 0;
  bool failed = false;
  std::vector<std::vector<Post>> sends;  // [rank]
  // returns false on time-out or if any rank flagged a failure
  bool barrier() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned long gen = generation;
    if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); return !failed; }
    if (!cv.wait_for(lk, std::chrono::seconds(60), [&] { return generation != gen; })) { failed = true; cv.notify_all(); return false; }
    return !failed;
  }
};
struct Comm { Group *g; int rank; };
struct Op { bool send; void *ptr; size_t bytes; int peer; Comm *comm; hipStream_t stream; };

std::mutex g_registry_lock;
std::map<std::string, Group *> g_groups;
std::atomic<unsigned long> g_next_id{1};
std::atomic<long> g_group_calls{0}, g_bytes{0}, g_errors{0};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local std::string t_error;
thread_local void *t_last_comm = nullptr;  // a group without operations still takes part in the rendezvous (one communicator per thread here)

size_t type_size(ncclDataType_t t) {
  switch (t) {
  case ncclInt8: case ncclUint8: return 1;
  case ncclFloat16: case ncclBfloat16: return 2;
  case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
  case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
  default: return 0;
  }
}
ncclResult_t fail(const std::string &msg) {
  t_error = "mock rccl: " + msg;
  g_errors++;
  return ncclInternalError;
}

ncclResult_t run_group() {
  std::vector<Op> ops;
  ops.swap(t_ops);
  g_group_calls++;
  Comm *c = ops.empty() ? static_cast<Comm *>(t_last_comm) : ops[0].comm;
  if (!c) return ncclSuccess;
  const hipStream_t stream = ops.empty() ? nullptr : ops[0].stream;
  for (const Op &o : ops) {
    if (o.comm != c) return fail("one group mixes communicators");
    if (o.stream != ops[0].stream) return fail("one group mixes streams");
    if (o.peer < 0 || o.peer >= c->g->world) return fail("peer " + std::to_string(o.peer) + " out of range");
  }
  Group *g = c->g;
  if (!ops.empty() && hipStreamSynchronize(stream) != hipSuccess) return fail("hipStreamSynchronize failed");
  {
    std::lock_guard<std::mutex> lk(g->m);
    g->sends[c->rank].clear();
    for (const Op &o : ops)
      if (o.send) g->sends[c->rank].push_back({o.peer, o.ptr, o.bytes, false});
  }
  std::string problem;
  if (!g->barrier()) problem = "rendezvous failed before the copies (a rank did not enter this exchange, or failed)";
  if (problem.empty())
    for (const Op &o : ops) {
      if (o.send) continue;
      Post *match = nullptr;
      {
        std::lock_guard<std::mutex> lk(g->m);
        for (Post &p : g->sends[o.peer])
          if (p.peer == c->rank && !p.taken) { match = &p; p.taken = true; break; }
      }
      if (!match) { problem = "rank " + std::to_string(c->rank) + " receives from rank " + std::to_string(o.peer) + ", which sends nothing to it: a hang over RCCL"; break; }
      if (match->bytes != o.bytes) { problem = "rank " + std::to_string(c->rank) + " expects " + std::to_string(o.bytes) + " bytes from rank " + std::to_string(o.peer) + ", which sends " + std::to_string(match->bytes); break; }
      if (hipMemcpy(o.ptr, match->ptr, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) { problem = "hipMemcpy failed"; break; }
      g_bytes += (long)o.bytes;
    }
  if (!problem.empty()) { std::lock_guard<std::mutex> lk(g->m); g->failed = true; }
  (void)hipDeviceSynchronize();
  const bool ok = g->barrier();
  if (problem.empty() && ok) {
    std::lock_guard<std::mutex> lk(g->m);
    for (const Post &p : g->sends[c->rank])
      if (!p.taken) { problem = "rank " + std::to_string(c->rank) + " sends " + std::to_string(p.bytes) + " bytes to rank " + std::to_string(p.peer) + ", which does not receive them: a hang over RCCL"; g->failed = true; break; }
  }
  if (!g->barrier() && problem.empty()) problem = "another rank failed in this exchange";
  if (!problem.empty()) return fail(problem);
  return ncclSuccess;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id, 0, sizeof *id);
  const unsigned long v = g_next_id++;
  std::memcpy(id->internal, "MOCKRCCL", 8);
  std::memcpy(id->internal + 8, &v, sizeof v);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
  if (!out || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
  if (std::memcmp(id.internal, "MOCKRCCL", 8)) return fail("unique id was not drawn from this library");
  Group *g;
  {
    std::lock_guard<std::mutex> lk(g_registry_lock);
    Group *&slot = g_groups[std::string(id.internal, sizeof id.internal)];
    if (!slot) { slot = new Group; slot->world = world; slot->sends.resize(world); }
    g = slot;
  }
  {
    std::unique_lock<std::mutex> lk(g->m);
    if (g->world != world) return fail("ranks disagree on the world size");
    g->attached++;
    g->cv.notify_all();
    if (!g->cv.wait_for(lk, std::chrono::seconds(60), [&] { return g->attached >= g->world; }))   // collective, like the real call
      return fail("ncclCommInitRank: only " + std::to_string(g->attached) + " of " + std::to_string(world) + " ranks arrived");
  }
  Comm *made = new Comm{g, rank};
  t_last_comm = made;
  *out = reinterpret_cast<ncclComm_t>(made);
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete reinterpret_cast<Comm *>(comm);
  return ncclSuccess;
}
ncclResult_t ncclGroupStart() { t_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return fail("ncclGroupEnd without ncclGroupStart");
  if (--t_depth) return ncclSuccess;
  return run_group();
}
static ncclResult_t enqueue(bool send, void *ptr, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  if (!comm || (!ptr && count)) return ncclInvalidArgument;
  const size_t ts = type_size(type);
  if (!ts) return fail("unknown datatype");
  t_ops.push_back({send, ptr, count * ts, peer, reinterpret_cast<Comm *>(comm), stream});
  if (t_depth == 0) return run_group();
  return ncclSuccess;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  return enqueue(true, const_cast<void *>(buf), count, type, peer, comm, stream);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  return enqueue(false, buf, count, type, peer, comm, stream);
}
const char *ncclGetErrorString(ncclResult_t r) {
  if (r == ncclSuccess) return "no error";
  return t_error.empty() ? "mock rccl: error" : t_error.c_str();
}
// test hooks
void mock_rccl_stats(long *group_calls, long *bytes, long *errors) {
  if (group_calls) *group_calls = g_group_calls;
  if (bytes) *bytes = g_bytes;
  if (errors) *errors = g_errors;
}
}
