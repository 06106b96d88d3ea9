// Arch.h — the execution backend behind the reference's Arch surface.
// Upstream, Arch (include/Arch.h:154-299, src/Arch.cpp) is the cycle model: per-cluster fetch / decode /
// issue / commit front ends that advance a cycle counter.  Here the same public method set drives REAL
// execution through the C ABI of include/homulator_hip.h: issueIns() queues a stage, update() launches the
// next stage on the GPU, simulateComplete() says whether everything was launched, getCycle() returns
// elapsed DEVICE NANOSECONDS (not cycles), shownStat() prints measured counters in the upstream format.
// backend = "count" executes nothing (no GPU needed): it only accounts instructions, for the structural
// parity tests against the compiled reference (tests/golden/structural.json).
// backend = "sim" keeps what upstream's Arch was: update() advances the build's own cycle model of the reference
// accelerator (include/SimModel.h) by one cycle, getCycle() returns CYCLES, shownStat() prints the reference's stat block.
#ifndef HOMULATOR_ARCH_H
#define HOMULATOR_ARCH_H
#include "Basic.h"
#include "Config.h"
#include "Instruction.h"
#include "Statistic.h"
#include "SimModel.h"

struct hm_ctx;
class DataMap;        // upstream's use-count map (include/mem.h:12-109) and per-cluster memory controller (:465-651): the cycle
class MemController;  // model's types, kept as opaque names so that upstream-shaped callers compile; never instantiated here

// one dispatched stage: all limb-level instructions of one stage key, in limb order
struct Stage {
  std::string name;  // stage key, e.g. "ModUp_INTT"
  ins_ops kind;
  std::vector<Instruction *> ins;
  uint32_t cluster0 = 0;  // placement of limb 0 (limb l runs on (cluster0 + l) % cluster upstream)
};

// a fill of input limbs with the deterministic synthetic stream
struct InputFill {
  std::vector<AddrType> addrs;
  std::vector<uint32_t> mods;
  uint64_t seed;
  bool shared = false;  // batch > 1: one copy serves every op of the batch (the evaluation key)
};

class Arch {
public:
  enum Backend { BACKEND_HIP = 0, BACKEND_COUNT = 1, BACKEND_SIM = 2 };

  explicit Arch(Config *cfg);
  ~Arch();

  // ---- build-specific setup (called by the Operation constructors)
  void bindParams(uint32_t maxLevel, uint32_t curLevel, uint32_t alpha);  // creates the HIP context (N from the config)
  // multi-GPU (config keys `world`, `rank`): limb-polys are sharded by modulus, q_i -> i % world and
  // p_j -> (curLevel + j) % world, i.e. extended limb e -> e % world — upstream's `limb % cluster` placement
  // (include/Driver.h:158,178).  One of the two transports must be set before prepare() when world > 1.
  uint32_t owner(uint32_t modId) const { return modId < maxLevel_ ? modId % world_ : (curLevel_ + (modId - maxLevel_)) % world_; }
  uint32_t rank() const { return rank_; }
  uint32_t world() const { return world_; }
  void commInitRccl(const void *uniqueId128);
  void commInitExternal(void *fn, void *user);
  void registerLimbs(const std::vector<AddrType> &limbStarts);  // gives every limb-poly a place in HBM
  void addInputFill(const InputFill &f) { fills.push_back(f); }
  // continuous execution: before every run, limb-polys `dst` of this op are copied (device to device, stream-ordered after
  // the producer's work) from limb-polys `srcAddrs` of `src`; the synthetic fill of `dst` is dropped
  void bindInput(const std::vector<AddrType> &dst, Arch *src, const std::vector<AddrType> &srcAddrs);
  uint64_t modulus(uint32_t modId) const;
  // constants of a base conversion (host side): qhat_inv[n_in]
  std::vector<uint64_t> bconvScale(const std::vector<uint32_t> &inMods);
  Backend backend() const { return backendKind; }
  // backend = sim: the literal reference program of the op (host/src/SimProgram.cpp); replaces any earlier one
  void loadSim(SimProgram &&program);
  SimModel *simModel() { return sim; }
  hm_ctx *context() { return ctx; }

  // ---- the reference's execution surface
  // upstream's two issue entry points, upstream's signatures (include/Arch.h:234-237, src/Arch.cpp:1000-1017): one instruction
  // group for a unit of a cluster / for MAC port (h, w) of the BCONV or HPIP array.  Each group joins the plan as a stage of its
  // own (stages of equal kind and dependency depth are coalesced into one launch anyway).  Upstream's Driver issues every BCONV /
  // HPIP group to ALL ports (include/Driver.h:307-320): the replicas (h, w) != (0, 0) are accounted, not executed again.
  void issueIns(uint32_t index, const std::string &name, std::vector<Instruction *> &insg);
  void issueIns(uint32_t index, uint32_t h, uint32_t w, std::vector<Instruction *> &insg, bool hpip);
  void setDataMap(DataMap *) {}                                   // include/Arch.h:165: the functional backends track no use counts
  std::vector<MemController *> getMemController() { return {}; }  // :244
  // the build's own form: a whole stage at once (what host/include/Driver.h issues)
  void issueIns(uint32_t cluster, const std::string &unit, const Stage &stage);
  void update();                // launch the next queued stage (src/Arch.cpp:912-929 advanced one cycle)
  bool simulateComplete();      // include/Arch.h:246-269
  unsigned long long getCycle();         // elapsed device time in ns
  unsigned long long getcompletedIns();  // upstream instructions retired so far
  void state();
  void shownStat();  // src/Arch.cpp:1019-1022

  // ---- whole-plan execution (bench / tests): prepare once, run many times
  void prepare();                       // allocate HBM, fill inputs, fuse and coalesce stages into launches
  void run();                           // enqueue every launch once (asynchronous)
  void sync();
  double timedRun(uint32_t iters);      // ns per iteration, device time
  bool readLimbs(const std::vector<AddrType> &addrs, uint64_t *host, uint32_t copy = 0);  // download limbs (N words each) of op `copy` of the batch
  bool writeLimbs(const std::vector<AddrType> &addrs, const uint64_t *host, uint32_t copy = 0);  // upload limbs (real ciphertexts / keys); shared key limbs live in copy 0
  // asynchronous helpers on the op's stream (tests of the chain ordering, streaming callers):
  void refill(const std::vector<AddrType> &addrs, uint64_t seed);              // new synthetic input data (all ops of the batch)
  void snapshot(const std::vector<AddrType> &addrs, uint32_t slot);             // device-side copy of the limbs as they are NOW in stream order
  bool readSnapshot(uint32_t slot, uint64_t *host);                             // synchronises, then downloads the slot
  uint32_t batch() const { return batch_; }
  size_t launchCount() const { return launches.size(); }
  std::string stageTimes(uint32_t iters);  // one line per launch: "<kind> <stage names> <ns>", each launch timed alone
  std::string planText() const;  // one line per launch: kind, stage names, limb count, exchange lists (tests)
  unsigned long long algorithmicBytes() const { return algBytes; }
  Statistic *stats() { return stat; }
  uint32_t N() const { return n; }

private:
  struct Launch;
  Config *config;
  Backend backendKind;
  bool fuse;
  bool shardFused = false;      // sharded: conversion + first pass on column slices, transform x key kernel on the owner (config key shard_fused)
  bool pipelineDigits = false;  // sharded: per-digit exchanges on the exchange stream (config key pipeline_digits, default 1 when world > 1)
  bool shardGather = false;     // sharded (round 5): the conversions' inputs are replicated (all-gather) and every rank converts for its own output limbs with the one-GPU kernels: config key shard_plan
  bool fuseHpip = true;
  bool fuseBconv = true;  // config key fuse_bconv: the ModUp conversion runs inside the first pass of the fused transform x key kernel   // config key fuse_hpip: the ModUp transforms' last pass runs inside the inner-product kernel
  bool fuseIpInv = true;     // pass (7b): the inner product's special limbs leave as the first pass of the ModDown's inverse transform
  bool packBconvIn = true;   // pass (11): inverse transforms that feed only base conversions store the split-30 packed form
  bool fuseAuto = true;      // pass (12): an automorphism read only as a transform's input / a fused transform's addend is gathered by that transform
  bool fuseModDown = true;   // pass (9): the ModDown conversion inside the merged transform's first pass (default: by batch size)
  uint32_t n = 0, logN = 0, clusterCount = 1;
  uint32_t maxLevel_ = 0, curLevel_ = 0, world_ = 1, rank_ = 0;
  uint32_t batch_ = 1;  // config key `batch`: independent ops (own inputs, shared evaluation key) carried by every launch
  bool commReady = false;
  bool useGraph = false;
  unsigned runCount = 0;
  void *graph = nullptr;  // hm_graph*: the whole plan captured once, replayed by run()
  std::vector<void *> sliceBuffers;
  std::map<uint32_t, std::pair<uint64_t *, size_t>> snapshots;  // slot -> (device buffer, limbs)
  hm_ctx *ctx = nullptr;
  void *hostParams = nullptr;  // hm::Params: moduli / roots / conversion constants on the host (both backends)
  uint64_t *pool = nullptr;  // all limb-polys, [limb][N]
  std::map<AddrType, uint32_t> limbIndex;
  std::vector<Stage> stages;
  std::vector<InputFill> fills;
  struct Binding { std::vector<AddrType> dst; Arch *src; std::vector<AddrType> srcAddrs; std::vector<uint32_t> dstLimbs, srcLimbs, mods; };
  std::vector<Binding> bindings;
  std::vector<Launch *> launches;
  size_t nextLaunch = 0;
  bool prepared = false;
  unsigned long long completedIns = 0, elapsedNs = 0, algBytes = 0;
  Statistic *stat;
  SimModel *sim = nullptr;

  uint32_t limbOf(AddrType a) const;
  void buildLaunches();
  void fusePasses(std::vector<Stage> &st);
  void enqueue(Launch &l);
  void replicateForBatch();
};
#endif
