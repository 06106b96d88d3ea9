// mock_rccl.cpp — TEST DOUBLE of the eight RCCL entry points the backend uses (homulator_amd/csrc/hm_backend.hip: rccl_load).
// RCCL refuses two ranks on one device, and the test box has one GPU: with this library (selected by HOMULATOR_RCCL_LIB, never
// by default) the ranks are THREADS of one process, every rank with its own hm_ctx / HIP stream / HBM pool, and the backend's
// RCCL code path — hm_comm_unique_id, the collective hm_comm_init_rccl, the grouped ncclSend / ncclRecv of every exchange with
// their counts, datatype, peers and streams — executes exactly as on a node.  Only the wire differs: a send is matched with
// the peer's receive by a rendezvous of the threads and copied device to device.
// What a real run would turn into a HANG is an ERROR here (after a 45 s time-out): a receive nobody sends to, a send nobody
// receives; a size the two sides disagree on and a peer out of range are reported at once.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
// one FIFO of posted sends per (source, destination) pair of a communicator: a send is posted without blocking, the receiver
// copies and acknowledges it; nobody else takes part (no global barrier: like RCCL, ranks only meet the peers they exchange with)
struct Post { const void *ptr; size_t bytes; unsigned long seq; };
struct Group {
  int world = 0, attached = 0;
  std::mutex m;
  std::condition_variable cv;
  std::vector<std::vector<Post>> box;         // [src * world + dst]: posted, not yet received
  std::vector<unsigned long> posted, acked;   // [src * world + dst]: counters
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
static std::chrono::seconds timeout_s() { const char *e = getenv("MOCK_RCCL_TIMEOUT_S"); return std::chrono::seconds(e ? atoi(e) : 45); }
const auto TIMEOUT = timeout_s();

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
  if (ops.empty()) return ncclSuccess;   // an empty group is a no-op, as over RCCL
  Comm *c = ops[0].comm;
  Group *g = c->g;
  const int W = g->world, me = c->rank;
  for (const Op &o : ops) {
    if (o.comm != c) return fail("one group mixes communicators");
    if (o.stream != ops[0].stream) return fail("one group mixes streams");
    if (o.peer < 0 || o.peer >= W) return fail("peer " + std::to_string(o.peer) + " out of range");
  }
  if (hipStreamSynchronize(ops[0].stream) != hipSuccess) return fail("hipStreamSynchronize failed");  // the double is synchronous
  std::vector<std::pair<int, unsigned long>> mine;  // (destination, sequence number) of my sends
  {
    std::lock_guard<std::mutex> lk(g->m);
    for (const Op &o : ops)
      if (o.send) {
        const size_t slot = (size_t)me * W + o.peer;
        g->box[slot].push_back({o.ptr, o.bytes, ++g->posted[slot]});
        mine.emplace_back(o.peer, g->posted[slot]);
      }
    g->cv.notify_all();
  }
  std::string problem;
  for (const Op &o : ops) {
    if (o.send) continue;
    const size_t slot = (size_t)o.peer * W + me;
    Post p{};
    {
      std::unique_lock<std::mutex> lk(g->m);
      if (!g->cv.wait_for(lk, TIMEOUT, [&] { return !g->box[slot].empty(); })) {
        problem = "rank " + std::to_string(me) + " receives from rank " + std::to_string(o.peer) + ", which sends nothing to it: a hang over RCCL";
        break;
      }
      p = g->box[slot].front();
      g->box[slot].erase(g->box[slot].begin());
    }
    if (p.bytes != o.bytes) problem = "rank " + std::to_string(me) + " expects " + std::to_string(o.bytes) + " bytes from rank " + std::to_string(o.peer) + ", which sends " + std::to_string(p.bytes);
    // a device-to-device hipMemcpy may return before the copy has run: the sender is told only when the bytes have really left its
    // buffer (with per-digit pipelined exchanges its next exchange packs into the same staging buffer right away; round 3 saw one run
    // in ten corrupted by acknowledging too early)
    else if (hipMemcpy(o.ptr, p.ptr, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess || hipDeviceSynchronize() != hipSuccess) problem = "hipMemcpy failed";
    else g_bytes += (long)o.bytes;
    {
      std::lock_guard<std::mutex> lk(g->m);
      g->acked[slot] = p.seq;   // also after a failed check: the sender must not wait for ever
      g->cv.notify_all();
    }
    if (!problem.empty()) break;
  }
  (void)hipDeviceSynchronize();
  for (const auto &s : mine) {   // my send buffers may be reused only when the receivers have copied them
    const size_t slot = (size_t)me * W + s.first;
    std::unique_lock<std::mutex> lk(g->m);
    if (!g->cv.wait_for(lk, TIMEOUT, [&] { return g->acked[slot] >= s.second; }) && problem.empty())
      problem = "rank " + std::to_string(me) + " sends to rank " + std::to_string(s.first) + ", which does not receive: a hang over RCCL";
  }
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
    if (!slot) { slot = new Group; slot->world = world; slot->box.resize((size_t)world * world); slot->posted.assign((size_t)world * world, 0); slot->acked.assign((size_t)world * world, 0); }
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
  *out = reinterpret_cast<ncclComm_t>(new Comm{g, rank});
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
