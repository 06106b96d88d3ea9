// Arch.cpp — execution backend: turns queued stages into GPU launches through the C ABI.
#include "Arch.h"

#include <algorithm>
#include <chrono>
#include <fstream>
#include <iterator>
#include <cstring>
#include <set>

#include "../../homulator_amd/csrc/hm_params.h"
#include "../../include/homulator_hip.h"

// the back-end's capability table for this ring size (hm_capability: host-side data, no GPU): which fused forms have kernels.  The fusion
// passes below ask it instead of naming ring sizes or digit widths (round 6).
static uint32_t cap(uint32_t logN, const char *name) {
  uint64_t v = 0;
  return hm_capability(logN, name, &v) == HM_OK ? (uint32_t)v : 0u;
}

// one C-ABI call, with its argument arrays prebuilt so that run() is a tight loop of calls
struct Arch::Launch {
  enum Kind { L_NTT, L_INTT, L_EWE, L_BCONV, L_AUTO, L_NTT_SUBSCALE, L_TENSOR, L_EXCH_IN, L_EXCH_OUT, L_REPLICATE, L_IP, L_NTT_IP,
              L_BCONV_COL, L_EXCH_IN_COL, L_EXCH_OUT_COL } kind;   // round 4: conversion + first pass on a rank's column slice, between the transposed-domain exchanges
  Launch *xin = nullptr, *xout = nullptr;   // sharded BCONV: the exchange launches around it (they share its slice buffers)
  int recordSlot = -1;            // exchange launches of a pipelined sharded plan: the mark set behind them (hm_exchange_mark)
  std::vector<int> waitSlots;     // marks the compute stream waits for before this launch (hm_exchange_wait)
  std::vector<uint8_t> ipCoeff;   // L_NTT_IP: per (limb, digit) 1 = transformed inside the kernel (a = source, c = first-pass scratch)
  std::vector<uint8_t> ipInv;     // L_NTT_IP (7b): per limb, 1 = the outputs leave as the first pass of their inverse transform
  bool secondOnly = false;        // L_INTT (7b): hm_ntt_second_pass — the first pass was run by the inner-product kernel
  std::vector<uint8_t> outPacked; // L_INTT (11): per limb-poly, 1 = stored in the split-30 packed form of the conversions' inputs
  std::vector<uint32_t> inGalois, addGalois;   // (12) L_INTT: per limb-poly, the input / L_NTT_SUBSCALE: the addend is read through X -> X^g (0: as stored); empty: none
  uint32_t xGalois = 0;                        // (12) L_NTT_IP: the evaluation-form digits, L_IP: the x operands are read through X -> X^g
  uint32_t ipTerms = 0, ipOuts = 0;
  std::string name;
  std::string statKey;
  int opcode = 0;
  uint32_t galois = 0;
  std::vector<uint32_t> a, b, c, d, out, out1, out2, mods, inMods;
  struct Prob { std::vector<uint32_t> in, inMods, out, outMods, epA, epB; std::vector<uint64_t> epK; bool epi = false, epAdd = false, inPacked = false; };
  std::vector<Prob> probs;  // BCONV: independent conversions batched into one launch
  // multi-GPU: exchange steps (limb list + owner of each limb) and the coefficient-slice buffers of a sharded BCONV
  std::vector<uint32_t> exLimbs, exOwners;
  uint64_t *slicesIn = nullptr, *slicesOut = nullptr;
  uint32_t logLen = 0;
  std::vector<uint64_t> k, mixK, addK;  // mixK / addK: prologue and addend constants of the merged ModDown + rescale transform
  bool hasK = false;
  unsigned long long refInstructions = 0;
  unsigned long long bytes = 0;  // operand limb-polys read + written x N x 8
};

static hm::Params &hostP(void *p) { return *static_cast<hm::Params *>(p); }

Arch::Arch(Config *cfg) : config(cfg) {
  n = cfg->getValue("N");
  logN = 0;
  while ((1u << logN) < n) ++logN;
  clusterCount = cfg->getValueOr("cluster", 1);
  uint32_t b = cfg->getValueOr("backend", BACKEND_HIP);
  if (const char *e = getenv("HOMULATOR_BACKEND")) b = std::string(e) == "count" ? BACKEND_COUNT : std::string(e) == "sim" ? BACKEND_SIM : BACKEND_HIP;
  backendKind = b == BACKEND_COUNT ? BACKEND_COUNT : b == BACKEND_SIM ? BACKEND_SIM : BACKEND_HIP;
  world_ = cfg->getValueOr("world", 0);
  rank_ = cfg->getValueOr("rank", 0);
  if (world_ == 0) {
    // no explicit `world` key: one rank per GPU when started by a multi-process launcher (torch.distributed.run, mpirun
    // wrappers: WORLD_SIZE / RANK / LOCAL_RANK).  The CLI's [cluster] argument doubles as the GPU count:
    // when it was given on the command line it must agree with the launcher.
    world_ = 1;
    const char *ws = getenv("WORLD_SIZE"), *rk = getenv("RANK"), *lr = getenv("LOCAL_RANK");
    // ... and only for the CLI (it sets `launcher_env`): a library user who builds an op inside a torchrun job without saying
    // `world` gets a plain one-GPU op on the device they asked for, not a silently sharded one on LOCAL_RANK
    if (ws && atoi(ws) > 1 && b == BACKEND_HIP && cfg->getValueOr("launcher_env", 0)) {
      world_ = (uint32_t)atoi(ws);
      rank_ = rk ? (uint32_t)atoi(rk) : 0;
      if (lr && !getenv("HOMULATOR_DEVICE") && !cfg->hasKey("device")) cfg->setValue("device", (uint32_t)atoi(lr));
      if (cfg->getValueOr("cluster_from_argv", 0) && clusterCount != world_)
        throw std::runtime_error("[cluster] = " + std::to_string(clusterCount) + " GPUs requested, but the launcher started " + std::to_string(world_) + " ranks");
    }
  }
  if (rank_ >= world_) throw std::runtime_error("rank must be below world");
  if (world_ & (world_ - 1)) throw std::runtime_error("world must be a power of two");
  // HIP-graph replay of the whole plan: no gain in steady state (the op is GPU-bound), but the host prepares a launch set in ~20 us instead of
  // ~0.3 ms: +3-5 % on a 20-step timed region of batched instances (bench.py turns it on for them), -4 % one op at a time (replay overhead)
  useGraph = cfg->getValueOr("graph", 0) != 0;
  if (const char *e = getenv("HOMULATOR_GRAPH")) useGraph = std::string(e) != "0";
  batch_ = std::max<uint32_t>(1, cfg->getValueOr("batch", 1));
  if (const char *e = getenv("HOMULATOR_BATCH")) batch_ = std::max(1, atoi(e));
  fuse = cfg->getValueOr("fuse", 1) != 0;
  if (const char *e = getenv("HOMULATOR_FUSE")) fuse = std::string(e) != "0";
  // the HPIP unit as a fused NTT-epilogue x evaluation-key MAC.  Upstream switches its HPIP unit with
  // `hasHPIPU` (src/Arch.cpp:10; 0 in the shipped files, where the inner product runs on the EWE); that key keeps describing
  // the SIMULATED machine (backend = sim).  On the GPU the fused kernel is a scheduling decision of the backend: `fuse_hpip`.
  fuseHpip = cfg->getValueOr("fuse_hpip", 1) != 0;
  if (const char *e = getenv("HOMULATOR_FUSE_HPIP")) fuseHpip = std::string(e) != "0";
  // ... and the ModUp base conversion inside the first pass of that transform (BConvOut_(j) never reaches HBM): N = 2^16, up to 15
  // input limbs per digit, one GPU (a sharded conversion works on coefficient slices)
  fuseBconv = cfg->getValueOr("fuse_bconv", 1) != 0;
  if (const char *e = getenv("HOMULATOR_FUSE_BCONV")) fuseBconv = std::string(e) != "0";
  // ... and the ModDown conversion inside the first pass of the merged ModDown + rescale transform (round 4, pass 9): built, bit-exact, and
  // measured SLOWER on MI355X in both modes (profiles/r04_fuse_ab.txt: batch 10 conversion 18.5 + transform 51.7 us per op against 3.6 + 70.8
  // fused; one at a time 24.9 + 70.5 against 9.5 + 89.2): the separate conversion kernel converts to all 35 outputs of a key from inputs it
  // loads and splits ONCE, the fused form loads the 15 input tiles again for every pair of outputs, and the 73 MB of ModdownBConvOut traffic it
  // saves is worth less than that.  Opt-in (config key fuse_moddown = 1).
  // (7b, round 5) InnerProOut -> ModDownINTTOut (src/Operation.cpp:294-445): the special limbs of the key-switch sum are read by nothing but
  // the ModDown's inverse transform, whose first pass is a ROW pass over the 16 rows the inner-product workgroup already owns: the kernel
  // runs it on its accumulators and the INTT launch keeps the COL pass.  N = 2^16, one GPU.  Config key fuse_ip_inv (default 1).
  fuseIpInv = cfg->getValueOr("fuse_ip_inv", 1) != 0;
  if (const char *e = getenv("HOMULATOR_FUSE_IP_INV")) fuseIpInv = std::string(e) != "0";
  // (11, round 5) the inverse transforms whose outputs are read by base conversions only (ModUp_DecompOut, ModDownBConvStep1) store them in
  // the split-30 packed form the conversions multiply with: two instructions per value in the producer instead of two per value and reading
  // workgroup (18 per value in a 35-output ModUp digit).  Config key pack_bconv_in (default 1).
  packBconvIn = cfg->getValueOr("pack_bconv_in", 1) != 0;
  if (const char *e = getenv("HOMULATOR_PACK_BCONV_IN")) packBconvIn = std::string(e) != "0";
  // (12, round 6) an automorphism whose output only feeds inverse transforms' inputs / fused forward transforms' addends is read through by them
  // (hrotate: AUTO_Key(0) -> the final add).  Config key fuse_auto (default 1).
  fuseAuto = cfg->getValueOr("fuse_auto", 1) != 0;
  if (const char *e = getenv("HOMULATOR_FUSE_AUTO")) fuseAuto = std::string(e) != "0";
  fuseModDown = cfg->getValueOr("fuse_moddown", 0) != 0;
  if (const char *e = getenv("HOMULATOR_FUSE_MODDOWN")) fuseModDown = std::string(e) != "0";
  // sharded runs: the exchanges of digit j+1 run on the context's exchange stream while digit j converts and transforms (SURVEY.md 7:
  // 2 beta + 2 all-to-alls per key switch instead of 4, same order on every rank).  The per-digit transforms must then stay separate
  // launches, so the fused NTT x key kernel (which needs all digits) is not used.
  pipelineDigits = world_ > 1 && cfg->getValueOr("pipeline_digits", 1) != 0;
  if (const char *e = getenv("HOMULATOR_PIPELINE_DIGITS")) pipelineDigits = world_ > 1 && std::string(e) != "0";
  // round 4: the fused kernels serve the sharded plan too — the ModUp conversion AND the first pass of its transforms run on the slice
  // holder's COLUMN slice (exchange in the transposed domain), the limb owner runs the transform x key kernel's second pass (config key
  // shard_fused, default 1; needs world <= the column tiles of a limb-poly the back-end deals out: cap_col_slices).  Otherwise the per-digit transforms stay launches of their own.
  shardFused = world_ > 1 && cfg->getValueOr("shard_fused", 1) != 0 && fuseHpip && fuseBconv && world_ <= cap(logN, "cap_col_slices");
  if (const char *e = getenv("HOMULATOR_SHARD_FUSED")) shardFused = shardFused && std::string(e) != "0";
  // Round 5: a second sharded plan, chosen by the bytes it moves.  `gather`: the conversions' INPUT limbs (the l scaled
  // limbs of the ModUp, the 2 alpha of the ModDown) are replicated to every rank (hm_replicate_limbs: one collective each) and every rank
  // runs the ONE-GPU kernels — conversion inside the first pass, transform x key — for the output limbs it owns: per-rank ingress
  // (l + 2 alpha)(G - 1) / G limb-polys against (2 beta E + 2 alpha)(G - 1) / G^2 for the all-to-all pair around every conversion: half the
  // bytes at G = 2, the same at G = 4, twice at G = 8 — and 3 collectives per key switch instead of 2 beta + 3.  The links are point to
  // point (one 77 GB/s link per pair and direction), so at G <= 4 the bytes decide.  Config key shard_plan: 0 = by rank count (gather up to 4
  // ranks, all-to-all above), 1 = all-to-all (the transposed-domain plan of round 4), 2 = gather.
  {
    uint32_t plan = cfg->getValueOr("shard_plan", 0);
    if (const char *e = getenv("HOMULATOR_SHARD_PLAN")) plan = (uint32_t)atoi(e);
    shardGather = world_ > 1 && (plan == 2 || (plan == 0 && world_ <= 4));
    if (shardGather) { shardFused = false; pipelineDigits = false; }
  }
  if (pipelineDigits && !shardFused) fuseHpip = false;
  stat = new Statistic();
}

Arch::~Arch() {
  for (Launch *l : launches) delete l;
  if (graph) hm_graph_destroy(static_cast<hm_graph *>(graph));
  if (ctx)
    for (void *p : sliceBuffers) hm_free(ctx, p);
  if (ctx)
    for (auto &kv : snapshots) hm_free(ctx, kv.second.first);
  if (ctx) {
    if (pool) hm_free(ctx, pool);
    hm_destroy(ctx);
  }
  delete static_cast<hm::Params *>(hostParams);
  delete stat;
  delete sim;
}

void Arch::loadSim(SimProgram &&program) {
  if (backendKind != BACKEND_SIM) throw std::runtime_error("loadSim: backend is not sim");
  delete sim;
  sim = nullptr;
  sim = new SimModel(config, std::move(program));
}

void Arch::commInitRccl(const void *id) {
  if (!ctx) throw std::runtime_error("commInitRccl: no HIP context");
  if (hm_comm_init_rccl(ctx, (int)rank_, (int)world_, id) != HM_OK) throw std::runtime_error(std::string("hm_comm_init_rccl: ") + hm_last_error(ctx));
  commReady = true;
}
void Arch::commInitExternal(void *fn, void *user) {
  if (!ctx) throw std::runtime_error("commInitExternal: no HIP context");
  if (hm_comm_init_external(ctx, (int)rank_, (int)world_, reinterpret_cast<hm_exchange_fn>(fn), user) != HM_OK)
    throw std::runtime_error(std::string("hm_comm_init_external: ") + hm_last_error(ctx));
  commReady = true;
}

void Arch::bindParams(uint32_t maxLevel, uint32_t curLevel, uint32_t alpha) {
  maxLevel_ = maxLevel;
  curLevel_ = curLevel;
  if (hostParams) return;
  hm::Params *hp = new hm::Params;
  // config key `chain_bits` (build-specific; the reference pins no modulus): 0 = the default chain (primes h 2^32 + 1 below 2^60: the
  // word-wise Montgomery back-end); b in [21, 60] = the maxLevel + alpha largest primes = 1 mod 2N below 2^b — 60 is SURVEY.md 8(d)'s
  // chain as written, 36 a chain of 36-bit words as upstream's `elementBitWidth` models (config/config_4.cfg:9).  hm_create picks the
  // arithmetic back-end from the chain it is handed.
  const uint32_t chainBits = config->getValueOr("chain_bits", 0);
  std::vector<uint64_t> chain;
  if (chainBits) chain = hm::Params::chain_below(logN, chainBits, maxLevel + alpha);
  const uint64_t *cq = chainBits ? chain.data() : nullptr, *cp = chainBits ? chain.data() + maxLevel : nullptr;
  hp->init(logN, maxLevel, alpha, cq, cp, nullptr, /*forGeneric: the host side only needs moduli and conversion constants*/ chainBits != 0);
  hostParams = hp;
  if (backendKind == BACKEND_HIP) {
    hm_params p = {logN, maxLevel, alpha, (int32_t)config->getValueOr("device", 0), cq, cp, nullptr};
    if (const char *e = getenv("HOMULATOR_DEVICE")) p.device = atoi(e);
    if (hm_create(&ctx, &p) != HM_OK)
      throw std::runtime_error(std::string("HIP backend unavailable: ") + hm_last_error(nullptr));
  }
}

uint64_t Arch::modulus(uint32_t modId) const { return hostP(hostParams).mod.at(modId); }

std::vector<uint64_t> Arch::bconvScale(const std::vector<uint32_t> &inMods) {
  std::vector<uint64_t> qh(inMods.size()), tb(inMods.size());
  hostP(hostParams).bconv_consts(inMods.data(), (uint32_t)inMods.size(), nullptr, 0, qh.data(), tb.data());
  return qh;
}

void Arch::registerLimbs(const std::vector<AddrType> &limbStarts) {
  for (AddrType a : limbStarts)
    if (!limbIndex.count(a)) {
      uint32_t idx = (uint32_t)limbIndex.size();
      limbIndex[a] = idx;
    }
}

uint32_t Arch::limbOf(AddrType a) const {
  auto it = limbIndex.find(a);
  if (it == limbIndex.end()) throw std::runtime_error("address " + std::to_string(a) + " is not a registered limb");
  return it->second;
}

void Arch::bindInput(const std::vector<AddrType> &dst, Arch *src, const std::vector<AddrType> &srcAddrs) {
  if (prepared) throw std::runtime_error("bindInput after prepare()");
  if (!src || src == this) throw std::runtime_error("bindInput: bad producer");
  if (dst.size() != srcAddrs.size()) throw std::runtime_error("bindInput: " + std::to_string(srcAddrs.size()) + " limb-polys produced, " + std::to_string(dst.size()) + " expected (levels differ)");
  if (src->n != n || src->batch_ != batch_ || src->world_ != world_ || src->rank_ != rank_ || src->backendKind != backendKind)
    throw std::runtime_error("bindInput: producer and consumer differ in N, batch, sharding or backend");
  bindings.push_back(Binding{dst, src, srcAddrs, {}, {}, {}});
}

void Arch::issueIns(uint32_t, const std::string &, const Stage &stage) {
  if (prepared) throw std::runtime_error("issueIns after prepare()");
  stages.push_back(stage);
}
void Arch::issueIns(uint32_t index, const std::string &name, std::vector<Instruction *> &insg) {
  if (insg.empty()) return;
  Stage st;
  st.name = name + "_" + std::to_string(stages.size());
  st.kind = insg[0]->ops;
  st.ins = insg;
  st.cluster0 = index;
  issueIns(index, name, st);
}
void Arch::issueIns(uint32_t index, uint32_t h, uint32_t w, std::vector<Instruction *> &insg, bool hpip) {
  if (h == 0 && w == 0) issueIns(index, hpip ? "HPIP" : "BCONV", insg);
}

// ---------------------------------------------------------------------------------------------------
// fusion passes on the stage list (fuse = 1).  All of them preserve every value that a later stage or the
// caller can observe except the intermediates they eliminate (listed in DESIGN.md §5).
// ---------------------------------------------------------------------------------------------------
void Arch::fusePasses(std::vector<Stage> &st) {
  // consumers of every address
  std::map<AddrType, int> uses;
  auto operands = [](Instruction *i) {
    std::vector<AddrType> v;
    if (i->ops == BCONV_STEP2) v.assign(i->operandList.begin(), i->operandList.end() - 1);
    else if (i->ops == MULT) {
      const int m[9] = {3, 15, 7, 5, 5, 1, 5, 1, 13};
      for (int b = 0; b < 4; ++b)
        if (m[i->opcode] & (1 << b)) v.push_back(i->operandList[b]);
    } else v.push_back(i->operandList[0]);
    return v;
  };
  std::map<AddrType, Instruction *> producer;
  for (auto &s : st)
    for (Instruction *i : s.ins) {
      for (AddrType a : operands(i)) uses[a]++;
      producer[i->OutputOperand] = i;
    }
  std::set<Instruction *> dead;
  // (1) pass-through NTT records: consumers read the source directly
  std::map<AddrType, AddrType> alias;
  for (auto &s : st)
    for (Instruction *i : s.ins)
      if ((i->ops == NTT) && i->passthrough) {
        alias[i->OutputOperand] = i->operandList[0];
        dead.insert(i);
      }
  for (auto &s : st)
    for (Instruction *i : s.ins) {
      if (dead.count(i)) continue;
      for (AddrType &a : i->operandList) {
        auto al = alias.find(a);
        if (al != alias.end()) a = al->second;
      }
    }
  // (2) INTT followed by a single MUL_CONST consumer: the constant goes into the INTT epilogue
  for (auto &s : st)
    for (Instruction *i : s.ins) {
      if (i->ops != MULT || i->opcode != EWE_MUL_CONST || dead.count(i)) continue;
      auto p = producer.find(i->operandList[0]);
      if (p == producer.end() || p->second->ops != INTT || uses[i->operandList[0]] != 1 || p->second->hasConstant) continue;
      p->second->hasConstant = true;
      p->second->constant = i->constant;
      p->second->OutputOperand = i->OutputOperand;
      p->second->refInstructions += i->refInstructions;
      producer[i->OutputOperand] = p->second;
      dead.insert(i);
    }
  // (3) EWE chains: SUB then MUL_CONST -> SUB_SCALE ; SUB_SCALE then ADD -> SUB_SCALE_ADD
  for (auto &s : st)
    for (Instruction *i : s.ins) {
      if (i->ops != MULT || dead.count(i)) continue;
      if (i->opcode == EWE_MUL_CONST) {
        auto p = producer.find(i->operandList[0]);
        if (p == producer.end() || p->second->ops != MULT || p->second->opcode != EWE_SUB || dead.count(p->second) ||
            uses[i->operandList[0]] != 1)
          continue;
        Instruction *sub = p->second;
        i->opcode = EWE_SUB_SCALE;
        i->operandList[0] = sub->operandList[0];
        i->operandList[2] = sub->operandList[2];
        i->refInstructions += sub->refInstructions;
        dead.insert(sub);
      } else if (i->opcode == EWE_ADD) {
        for (int side = 0; side < 2; ++side) {
          const int me = side ? 2 : 0, other = side ? 0 : 2;
          auto p = producer.find(i->operandList[me]);
          if (p == producer.end() || p->second->ops != MULT || p->second->opcode != EWE_SUB_SCALE || dead.count(p->second) ||
              uses[i->operandList[me]] != 1)
            continue;
          Instruction *ss = p->second;
          const AddrType addend = i->operandList[other];
          i->opcode = EWE_SUB_SCALE_ADD;
          i->operandList[0] = ss->operandList[0];
          i->operandList[2] = ss->operandList[2];
          i->operandList[3] = addend;
          i->hasConstant = true;
          i->constant = ss->constant;
          i->refInstructions += ss->refInstructions;
          dead.insert(ss);
          break;
        }
      }
    }
  // (4) forward NTT whose only consumer is (minuend - x) * k [+ addend]: the epilogue moves into the transform's
  //     last pass (ModDowNTT + ModDownSub + final add; Rescale_NTT + Rescale_SUB + Rescale_Mul)
  for (auto &s : st)
    for (Instruction *i : s.ins) {
      if (i->ops != MULT || dead.count(i) || (i->opcode != EWE_SUB_SCALE && i->opcode != EWE_SUB_SCALE_ADD)) continue;
      auto p = producer.find(i->operandList[2]);
      if (p == producer.end() || p->second->ops != NTT || p->second->passthrough || dead.count(p->second) ||
          p->second->fusedSubScale || uses[i->operandList[2]] != 1)
        continue;
      Instruction *t = p->second;
      const AddrType minuend = i->operandList[0], addend = i->opcode == EWE_SUB_SCALE_ADD ? i->operandList[3] : 0;
      const AddrType out = i->OutputOperand;
      if (out == t->operandList[0] || out == minuend || out == addend) continue;  // out doubles as first-pass scratch
      t->fusedSubScale = true;
      t->fMinuend = minuend;
      t->fAddend = addend;
      t->hasConstant = true;
      t->constant = i->constant;
      t->OutputOperand = out;
      t->refInstructions += i->refInstructions;
      producer[out] = t;
      dead.insert(i);
    }
  // (4b) ModDown finish followed by the rescale of the same limb.  T: h = (ip - NTT(conv)) * kT + d and
  //      R: out = (h - NTT(r)) * kR with r = INTT(last limb of h) are both linear in the coefficient domain:
  //      out = (ip - NTT(conv + kT^-1 * r)) * (kT kR) + d * kR — ONE transform per limb instead of two, and h never
  //      exists (the last limb keeps T: r comes from it).  Bit-identical: everything is exact modular arithmetic.
  for (auto &s : st)
    for (Instruction *R : s.ins) {
      if (R->ops != NTT || !R->fusedSubScale || R->fAddend || R->fMix || dead.count(R)) continue;
      auto p = producer.find(R->fMinuend);
      if (p == producer.end() || p->second == R || !p->second->fusedSubScale || p->second->fMix || dead.count(p->second) ||
          p->second->mod_id != R->mod_id || uses[R->fMinuend] != 1)
        continue;
      Instruction *T = p->second;
      const uint64_t q = modulus(R->mod_id);
      // R survives (its stage comes after the INTT that produces r, so the stage list stays a topological order)
      R->fMix = R->operandList[0];
      R->fMixConst = hm::invmod(T->constant, q);
      R->operandList[0] = T->operandList[0];
      R->fMinuend = T->fMinuend;
      R->fAddend = T->fAddend;
      R->fAddendConst = T->fAddend ? R->constant : 0;
      R->constant = hm::mulmod(T->constant, R->constant, q);
      R->refInstructions += T->refInstructions;
      dead.insert(T);
    }
  // (4c) the residue r = INTT(h_last) that the merged records mix in.  h_last = (ip - NTT(conv)) * kT + d is only ever
  //      needed in coefficient form, where it is (INTT(ip) - conv) * kT + INTT(d): conv never has to be transformed and
  //      brought back.  INTT(ip) and INTT(d) depend on nothing after the inner product, so they join the ModDown INTT
  //      launch (equal dependency depth), and one element-wise SUB_SCALE_ADD on the last limb replaces the two
  //      latency-bound single-limb transforms T_last and INTT(h_last).
  {
    struct Rewrite { Instruction *T, *X; size_t stage; };
    std::vector<Rewrite> todo;
    for (size_t si = 0; si < st.size(); ++si)
      for (Instruction *X : st[si].ins) {
        if (X->ops != INTT || X->hasConstant || dead.count(X)) continue;
        auto p = producer.find(X->operandList[0]);
        if (p == producer.end() || dead.count(p->second) || !p->second->fusedSubScale || p->second->fMix || p->second->ops != NTT ||
            uses[X->operandList[0]] != 1)
          continue;
        bool mixedIn = false;
        for (auto &s2 : st)
          for (Instruction *i : s2.ins) mixedIn |= !dead.count(i) && i->fMix == X->OutputOperand;
        if (mixedIn) todo.push_back(Rewrite{p->second, X, si});
      }
    AddrType fresh = limbIndex.empty() ? 1 : limbIndex.rbegin()->first + 1;
    for (const Rewrite &w : todo) {
      Instruction *T = w.T, *X = w.X;
      const AddrType h = X->operandList[0], r = X->OutputOperand;
      Instruction *ia = new Instruction("INTT", INTT, T->level_id);   // wa = INTT(ip_last), kept in h's limb
      ia->mod_id = T->mod_id; ia->operandList = {T->fMinuend}; ia->OutputOperand = h; ia->refInstructions = T->refInstructions;
      st[w.stage].ins.push_back(ia);
      AddrType wb = 0;
      if (T->fAddend) {                                               // wb = INTT(d_last), in a limb of its own
        wb = fresh++;
        registerLimbs({wb});
        Instruction *ib = new Instruction("INTT", INTT, T->level_id);
        ib->mod_id = T->mod_id; ib->operandList = {T->fAddend}; ib->OutputOperand = wb;
        st[w.stage].ins.push_back(ib);
      }
      Instruction *e = new Instruction("MULT", MULT, T->level_id);    // r = (wa - conv_last) * kT [+ wb]
      e->mod_id = T->mod_id;
      e->opcode = wb ? EWE_SUB_SCALE_ADD : EWE_SUB_SCALE;
      e->operandList = {h, 0, T->operandList[0], wb};
      e->hasConstant = true; e->constant = T->constant;
      e->OutputOperand = r; e->refInstructions = X->refInstructions;
      st[w.stage].ins.push_back(e);
      producer[h] = ia; producer[r] = e;
      dead.insert(T); dead.insert(X);
    }
  }
  // (5) tensor product: d1 = p*s + r*t (MAC2) with d0 = p*t and d2 = r*s (MUL) of the same limb -> one pass
  {
    std::map<std::pair<AddrType, AddrType>, Instruction *> muls;
    for (auto &s : st)
      for (Instruction *i : s.ins)
        if (i->ops == MULT && i->opcode == EWE_MUL && !dead.count(i)) muls[{i->operandList[0], i->operandList[1]}] = i;
    for (auto &s : st)
      for (Instruction *i : s.ins) {
        if (i->ops != MULT || i->opcode != EWE_MAC2 || dead.count(i)) continue;
        const AddrType P = i->operandList[0], S = i->operandList[1], R = i->operandList[2], T = i->operandList[3];
        auto u = muls.find({P, T}), v = muls.find({R, S});
        if (u == muls.end() || v == muls.end() || u->second->mod_id != i->mod_id || v->second->mod_id != i->mod_id) continue;
        i->fusedTensor = true;
        i->extraOutputs = {u->second->OutputOperand, v->second->OutputOperand};
        i->refInstructions += u->second->refInstructions + v->second->refInstructions;
        dead.insert(u->second);
        dead.insert(v->second);
      }
  }
  // (6) inner product with the evaluation key: the chain MAC2 / MAC_ADD ... of one key collapses into a single sum
  //     of products, and the two keys (same ext operands) into one two-output record (the HPIP unit's job)
  {
    struct Dot { std::vector<AddrType> x, y; std::vector<Instruction *> members; };
    std::map<Instruction *, Dot> dots;
    auto single = [&](AddrType a) { return uses[a] == 1; };
    std::vector<Instruction *> order;
    for (auto &s : st)
      for (Instruction *i : s.ins) order.push_back(i);
    for (Instruction *i : order) {
      if (i->ops != MULT || dead.count(i) || i->fusedTensor) continue;
      Dot d;
      if (i->opcode == EWE_MUL) { d.x = {i->operandList[0]}; d.y = {i->operandList[1]}; }
      else if (i->opcode == EWE_MAC2) { d.x = {i->operandList[0], i->operandList[2]}; d.y = {i->operandList[1], i->operandList[3]}; }
      else if (i->opcode == EWE_MAC_ADD) {
        auto p = producer.find(i->operandList[2]);
        if (p == producer.end() || !dots.count(p->second) || !single(i->operandList[2]) || p->second->mod_id != i->mod_id) continue;
        d = dots[p->second];
        d.x.push_back(i->operandList[0]);
        d.y.push_back(i->operandList[1]);
      } else continue;
      d.members.push_back(i);
      dots[i] = d;
    }
    // keep the dots that end a chain (nobody extends them) and have a partner with the same x list or >= 3 terms
    std::set<Instruction *> extended;
    for (auto &kv : dots)
      for (size_t m = 0; m + 1 < kv.second.members.size(); ++m) extended.insert(kv.second.members[m]);
    std::map<std::pair<uint32_t, std::vector<AddrType>>, Instruction *> byX;
    for (Instruction *i : order) {
      auto it = dots.find(i);
      if (it == dots.end() || extended.count(i) || it->second.x.size() > 4) continue;
      Dot &d = it->second;
      auto key = std::make_pair(i->mod_id, d.x);
      auto partner = byX.find(key);
      if (partner == byX.end()) { byX[key] = i; continue; }
      Instruction *a = partner->second;  // first key
      Dot &da = dots[a];
      if (a->ops == IP) continue;        // already paired
      // `i` (the later one) carries the fused record so that it is scheduled after every member of both chains
      const AddrType outFirst = a->OutputOperand, outSecond = i->OutputOperand;
      i->ops = IP;
      i->ipX = d.x;
      i->ipY = {da.y, d.y};
      i->OutputOperand = outFirst;
      i->extraOutputs = {outSecond};
      producer[outFirst] = i;
      for (Instruction *m : da.members) { i->refInstructions += m->refInstructions; dead.insert(m); }
      for (Instruction *m : d.members)
        if (m != i) { i->refInstructions += m->refInstructions; dead.insert(m); }
      byX.erase(partner);
    }
  }
  // (7) HPIP as SURVEY.md 8f-2 specifies it: a forward transform whose only reader is an inner-product record moves INTO that
  //     record (ModUp_NTT_(j) + InnerProOut: src/Operation.cpp:190-414).  The kernel runs the digit's ROW pass and multiplies
  //     its registers into both keys' accumulators; the extended digit (NTTOut_beta(j)) is never written or read back — its
  //     buffer only serves the first pass as scratch.
  if (fuseHpip) {
    std::map<AddrType, std::vector<Instruction *>> readers;
    for (auto &s : st)
      for (Instruction *i : s.ins) {
        if (dead.count(i)) continue;
        if (i->ops == IP && !i->ipX.empty()) {
          for (AddrType x : i->ipX) readers[x].push_back(i);
          for (auto &y : i->ipY) for (AddrType yy : y) readers[yy].push_back(i);
        } else {
          for (AddrType a : operands(i)) readers[a].push_back(i);
          if (i->fusedSubScale) { readers[i->fMinuend].push_back(i); if (i->fAddend) readers[i->fAddend].push_back(i); if (i->fMix) readers[i->fMix].push_back(i); }
        }
      }
    for (auto &s : st)
      for (Instruction *ip : s.ins) {
        if (ip->ops != IP || ip->ipX.empty() || dead.count(ip)) continue;
        ip->ipSrc = ip->ipX;
        ip->ipCoeff.assign(ip->ipX.size(), 0);
        std::vector<Instruction *> conv(ip->ipX.size(), nullptr);
        // widest digit the fused conversion + first pass takes (0: none at this ring size); config key fuse_bconv_max_in caps it below what the
        // back-end offers (A/B runs: 15 = the plan of rounds 3-5, where wider digits kept a conversion launch of their own)
        // (default: the widest digit for which the fused form measured faster at this ring size, cap_bconv_col_pref_in)
        const uint32_t maxConvIn = std::min<uint32_t>(cap(logN, "cap_bconv_col_max_in"), config->getValueOr("fuse_bconv_max_in", cap(logN, "cap_bconv_col_pref_in")));
        bool allConv = fuseBconv && (world_ == 1 || shardFused || shardGather) && maxConvIn != 0;
        for (size_t j = 0; j < ip->ipX.size(); ++j) {
          auto p = producer.find(ip->ipX[j]);
          if (p == producer.end()) continue;
          Instruction *t = p->second;
          if (t->ops != NTT || t->passthrough || t->fusedSubScale || dead.count(t) || t->mod_id != ip->mod_id) continue;
          auto &rd = readers[ip->ipX[j]];
          if (rd.size() != 1 || rd[0] != ip) continue;
          ip->ipSrc[j] = t->operandList[0];
          ip->ipCoeff[j] = 1;
          ip->refInstructions += t->refInstructions;
          dead.insert(t);
          // (8) is the transform's input a conversion output that nobody else reads?
          auto pb = producer.find(t->operandList[0]);
          auto &rb = readers[t->operandList[0]];
          if (pb != producer.end() && pb->second->ops == BCONV_STEP2 && !dead.count(pb->second) && rb.size() == 1 && rb[0] == t &&
              pb->second->operandList.size() - 1 <= maxConvIn && pb->second->mod_id == ip->mod_id)
            conv[j] = pb->second;
          else allConv = false;
        }
        if (allConv && std::find(ip->ipCoeff.begin(), ip->ipCoeff.end(), 1) != ip->ipCoeff.end()) {
          ip->ipConvIn.assign(ip->ipX.size(), {});
          ip->ipConvMods.assign(ip->ipX.size(), {});
          for (size_t j = 0; j < ip->ipX.size(); ++j) {
            if (!conv[j]) continue;
            ip->ipConvIn[j].assign(conv[j]->operandList.begin(), conv[j]->operandList.end() - 1);
            ip->ipConvMods[j] = conv[j]->inMods;
            ip->refInstructions += conv[j]->refInstructions * (unsigned long long)config->getValueOr("bconv_num_high", 1) * config->getValueOr("bconv_num_width", 1);
            dead.insert(conv[j]);
          }
        }
      }
  }
  // (7b, round 5) an inner-product record (HPIP form) whose outputs are read by inverse transforms and nothing else — the special limbs of
  //      the key-switch sum (ModDownINTTOut_Key(k)) and, with (4c), the last Q limb (the rescale residue's INTT) — hands them over as the
  //      FIRST pass of that inverse transform: a ROW pass over the 16 rows of the limb-poly the workgroup has just accumulated (the forward
  //      ROW pass's last round and the inverse ROW pass's first are the same round, so the pass runs from the registers).  The kernel stores
  //      the pass's hand-off into the INTT's output limb, the INTT record keeps its COL pass (hm_ntt_second_pass, with its scale), and the
  //      evaluation-form sums of those limbs (InnerProduceOut_Key{k}[0 .. alpha)) are never written or read back.
  if (fuseHpip && fuseIpInv && (world_ == 1 || shardGather) && cap(logN, "cap_ip_inverse_out")) {
    std::map<AddrType, std::vector<Instruction *>> readers;
    for (auto &s : st)
      for (Instruction *i : s.ins) {
        if (dead.count(i)) continue;
        if (i->ops == IP && !i->ipX.empty()) {
          for (AddrType x : (i->ipSrc.empty() ? i->ipX : i->ipSrc)) readers[x].push_back(i);
          for (auto &cin : i->ipConvIn) for (AddrType x : cin) readers[x].push_back(i);
          for (auto &y : i->ipY) for (AddrType yy : y) readers[yy].push_back(i);
        } else {
          for (AddrType a : operands(i)) readers[a].push_back(i);
          if (i->fusedSubScale) { readers[i->fMinuend].push_back(i); if (i->fAddend) readers[i->fAddend].push_back(i); if (i->fMix) readers[i->fMix].push_back(i); }
        }
      }
    for (auto &s : st)
      for (Instruction *ip : s.ins) {
        if (ip->ops != IP || ip->ipX.empty() || dead.count(ip) || ip->ipInvOut) continue;
        if (std::find(ip->ipCoeff.begin(), ip->ipCoeff.end(), 1) == ip->ipCoeff.end()) continue;   // the fused transform x key kernel only
        std::vector<AddrType *> outs = {&ip->OutputOperand};
        for (AddrType &o : ip->extraOutputs) outs.push_back(&o);
        std::vector<Instruction *> inv;
        for (AddrType *o : outs) {
          auto &rd = readers[*o];
          if (rd.size() != 1 || rd[0]->ops != INTT || dead.count(rd[0]) || rd[0]->mod_id != ip->mod_id || rd[0]->secondOnly || rd[0]->operandList[0] != *o) break;
          inv.push_back(rd[0]);
        }
        if (inv.size() != outs.size()) continue;
        for (size_t k = 0; k < outs.size(); ++k) {
          *outs[k] = inv[k]->OutputOperand;                      // the hand-off lands where the inverse transform finishes in place
          inv[k]->operandList[0] = inv[k]->OutputOperand;
          inv[k]->secondOnly = true;
        }
        ip->ipInvOut = true;
      }
  }
  // (9, round 4) the ModDown side of (8): a fused forward transform (ModDowNTT + ModDownSub [+ rescale]) whose input is a P -> Q conversion
  //     output that nobody else reads takes the conversion into its first pass (src/Operation.cpp:489-590): ModdownBConvOut_Key(k) is
  //     never written or read back.  The last limb of a key keeps its conversion: the rescale residue is formed from it element-wise (4c).
  if (fuseBconv && fuseModDown && (world_ == 1 || shardGather) && cap(logN, "cap_bconv_col_max_in_mix")) {
    const uint32_t maxMixIn = cap(logN, "cap_bconv_col_max_in_mix");
    std::map<AddrType, std::vector<Instruction *>> readers;
    for (auto &s : st)
      for (Instruction *i : s.ins) {
        if (dead.count(i)) continue;
        if (i->ops == IP && !i->ipX.empty()) {
          for (AddrType x : (i->ipSrc.empty() ? i->ipX : i->ipSrc)) readers[x].push_back(i);
          for (auto &cin : i->ipConvIn) for (AddrType x : cin) readers[x].push_back(i);
          for (auto &y : i->ipY) for (AddrType yy : y) readers[yy].push_back(i);
        } else {
          for (AddrType a : operands(i)) readers[a].push_back(i);
          if (i->fusedSubScale) { readers[i->fMinuend].push_back(i); if (i->fAddend) readers[i->fAddend].push_back(i); if (i->fMix) readers[i->fMix].push_back(i); }
        }
      }
    for (auto &s : st)
      for (Instruction *t : s.ins) {
        if (t->ops != NTT || !t->fusedSubScale || t->passthrough || dead.count(t)) continue;
        auto pb = producer.find(t->operandList[0]);
        if (pb == producer.end() || pb->second->ops != BCONV_STEP2 || dead.count(pb->second) || pb->second->mod_id != t->mod_id) continue;
        auto &rb = readers[t->operandList[0]];
        if (rb.size() != 1 || rb[0] != t || pb->second->operandList.size() - 1 > (t->fMix ? maxMixIn : cap(logN, "cap_bconv_col_max_in"))) continue;
        t->fConvIn.assign(pb->second->operandList.begin(), pb->second->operandList.end() - 1);
        t->fConvMods = pb->second->inMods;
        t->refInstructions += pb->second->refInstructions * (unsigned long long)config->getValueOr("bconv_num_high", 1) * config->getValueOr("bconv_num_width", 1);
        dead.insert(pb->second);
      }
  }
  // (10, round 4) r = (wa - conv_last) * kT [+ wb] (4c) directly behind the conversion that produces conv_last: the one-limb element-wise
  //      launch becomes the conversion kernel's epilogue (hm_bconv_desc::sub_from)
  if (fuseBconv && (world_ == 1 || shardGather)) {
    std::map<AddrType, int> nread;
    for (auto &s : st)
      for (Instruction *i : s.ins) {
        if (dead.count(i)) continue;
        if (i->ops == IP && !i->ipX.empty()) {
          for (AddrType x : (i->ipSrc.empty() ? i->ipX : i->ipSrc)) nread[x]++;
          for (auto &cin : i->ipConvIn) for (AddrType x : cin) nread[x]++;
          for (auto &y : i->ipY) for (AddrType yy : y) nread[yy]++;
        } else {
          for (AddrType a : operands(i)) nread[a]++;
          if (i->fusedSubScale) { nread[i->fMinuend]++; if (i->fAddend) nread[i->fAddend]++; if (i->fMix) nread[i->fMix]++; }
          for (AddrType x : i->fConvIn) nread[x]++;
        }
      }
    for (auto &s : st)
      for (size_t ei = 0; ei < s.ins.size(); ++ei) {
        Instruction *e = s.ins[ei];
        if (e->ops != MULT || dead.count(e) || (e->opcode != EWE_SUB_SCALE && e->opcode != EWE_SUB_SCALE_ADD)) continue;
        auto pb = producer.find(e->operandList[2]);
        if (pb == producer.end() || pb->second->ops != BCONV_STEP2 || dead.count(pb->second) || pb->second->fusedEpi || pb->second->mod_id != e->mod_id ||
            nread[e->operandList[2]] != 1)
          continue;
        Instruction *cv = pb->second;
        cv->fusedEpi = true;
        cv->fSubFrom = e->operandList[0];
        cv->fAdd = e->opcode == EWE_SUB_SCALE_ADD ? e->operandList[3] : 0;
        cv->hasConstant = true;
        cv->constant = e->constant;
        cv->OutputOperand = e->OutputOperand;
        cv->refExtra += e->refInstructions;   // (a conversion's own count is scaled by the MAC ports at launch time, the epilogue's is not)
        producer[cv->OutputOperand] = cv;
        // the conversion now reads what e read (the inverse transforms of 4c, queued in e's stage): it takes e's place in the stage list,
        // which stays a topological order
        for (auto &s2 : st) s2.ins.erase(std::remove(s2.ins.begin(), s2.ins.end(), cv), s2.ins.end());
        std::replace(s.ins.begin(), s.ins.end(), e, cv);
        dead.insert(e);
      }
  }
  // (11, round 5) split-30 packed conversion inputs.  A limb-poly that an inverse transform writes and that nothing but base conversions read
  //      (as a conversion INPUT: a separate conversion, a conversion inside a transform x key record or inside a fused transform) is
  //      stored packed; a conversion takes packed inputs only if all of them are.
  if (packBconvIn && (world_ == 1 || shardGather)) {
    struct Conv { std::vector<AddrType> in; Instruction *ins; int digit; };   // digit: index into ipConvIn, -1 = the record's own conversion
    std::vector<Conv> convs;
    std::map<AddrType, int> otherReads;   // reads of an address that are not a conversion input
    for (auto &s : st)
      for (Instruction *i : s.ins) {
        if (dead.count(i)) continue;
        if (i->ops == IP && !i->ipX.empty()) {
          const auto &src = i->ipSrc.empty() ? i->ipX : i->ipSrc;
          for (size_t j = 0; j < src.size(); ++j) {
            const bool conv = j < i->ipConvIn.size() && !i->ipConvIn[j].empty();
            if (conv) convs.push_back(Conv{i->ipConvIn[j], i, (int)j});
            else otherReads[src[j]]++;
          }
          for (auto &y : i->ipY) for (AddrType yy : y) otherReads[yy]++;
          continue;
        }
        if (i->ops == INTT && i->secondOnly) continue;   // (7b) its one operand is its own output: the in-place second pass is no other reader of it
        if (i->ops == BCONV_STEP2) convs.push_back(Conv{std::vector<AddrType>(i->operandList.begin(), i->operandList.end() - 1), i, -1});
        else if (!i->fConvIn.empty()) {
          if (i->fMix) { for (AddrType a : i->fConvIn) otherReads[a]++; }   // (the fused conversion with the mix prologue takes plain inputs)
          else convs.push_back(Conv{i->fConvIn, i, -1});
        } else for (AddrType a : operands(i)) otherReads[a]++;
        if (i->fusedSubScale) { otherReads[i->fMinuend]++; if (i->fAddend) otherReads[i->fAddend]++; if (i->fMix) otherReads[i->fMix]++; }
        if (i->fusedEpi) { otherReads[i->fSubFrom]++; if (i->fAdd) otherReads[i->fAdd]++; }
      }
    std::set<AddrType> cand;
    for (auto &s : st)
      for (Instruction *x : s.ins)
        if (x->ops == INTT && !dead.count(x) && !otherReads.count(x->OutputOperand)) cand.insert(x->OutputOperand);
    for (bool changed = true; changed;) {   // a conversion with one plain input keeps all of its inputs plain
      changed = false;
      for (const Conv &cv : convs) {
        bool all = true;
        for (AddrType a : cv.in) all &= cand.count(a) != 0;
        if (all) continue;
        for (AddrType a : cv.in) changed |= cand.erase(a) != 0;
      }
    }
    std::set<AddrType> read;
    for (const Conv &cv : convs) {
      if (cv.in.empty() || !cand.count(cv.in[0])) continue;
      if (cv.digit >= 0) { cv.ins->ipConvPacked.resize(cv.ins->ipConvIn.size(), 0); cv.ins->ipConvPacked[(size_t)cv.digit] = 1; }
      else cv.ins->inPacked = true;
      read.insert(cv.in.begin(), cv.in.end());
    }
    for (auto &s : st)
      for (Instruction *x : s.ins)
        if (x->ops == INTT && !dead.count(x) && read.count(x->OutputOperand)) x->packedOut = true;
  }
  // (12) round 6: an automorphism whose output is read ONLY as the input of inverse transforms, as the addend of fused forward transforms and / or as
  //      the evaluation-form digits of transform x key records folds into those readers (hm_ntt_desc.in_galois, hm_ntt_fused_desc.addend_galois,
  //      hm_ntt_ip_desc.x_galois): the index map takes aligned blocks to aligned blocks, so a kernel gathers through it with its own 16-byte loads,
  //      and AUTOOutput is never written or read back.  hrotate: AUTO_Key(1) -> ModUp_INTT + the key product's own digits, AUTO_Key(0) -> the final
  //      add inside ModDowNTT's epilogue: 6 -> 5 launches, 140 limb-polys less traffic.  Config key fuse_auto (default 1).
  //      The key product takes ONE Galois element per launch: its records fold only if every evaluation-form digit of every such record of the op
  //      is the output of a foldable automorphism by the same element.  Sharded plans fold the same way: a limb-poly's transforms and key product run on
  //      the rank that owns the limb, where the automorphism's source limb lives too.
  if (fuseAuto) {
    struct Reader { Instruction *ins; int role; size_t digit; };   // role 0: INTT input, 1: fused forward transform's addend, 2: evaluation-form digit of a
    std::map<AddrType, std::vector<Reader>> readers;               // transform x key record, -1: anything else
    std::vector<Reader> ownDigits;
    for (auto &s : st)
      for (Instruction *i : s.ins) {
        if (dead.count(i)) continue;
        if (i->ops == IP && !i->ipX.empty()) {
          // a transform x key record reads its own digits in evaluation form, a plain inner-product record (no digit transformed inside) all of them
          const bool anyT = std::find(i->ipCoeff.begin(), i->ipCoeff.end(), 1) != i->ipCoeff.end(), can = !i->ipXGalois;   // (any plan: a limb-poly's key product runs on the rank that owns the limb, and so does the automorphism's source limb)
          const auto &src = i->ipSrc.empty() ? i->ipX : i->ipSrc;
          for (size_t j = 0; j < src.size(); ++j) {
            const bool own = can && (!anyT || (!i->ipCoeff[j] && !(j < i->ipConvIn.size() && !i->ipConvIn[j].empty())));
            readers[src[j]].push_back({i, own ? 2 : -1, j});
            if (own) ownDigits.push_back({i, 2, j});
          }
          for (auto &v : i->ipConvIn) for (AddrType a : v) readers[a].push_back({i, -1, 0});
          for (auto &y : i->ipY) for (AddrType a : y) readers[a].push_back({i, -1, 0});
          continue;
        }
        const bool plainIntt = i->ops == INTT && !i->secondOnly && i->fConvIn.empty() && !i->inGalois;
        for (AddrType a : operands(i)) readers[a].push_back({i, plainIntt && a == i->operandList[0] ? 0 : -1, 0});
        for (AddrType a : i->fConvIn) readers[a].push_back({i, -1, 0});
        if (i->fusedSubScale) {
          readers[i->fMinuend].push_back({i, -1, 0});
          if (i->fAddend) readers[i->fAddend].push_back({i, i->ops == NTT && !i->fMix && i->fConvIn.empty() && !i->fAddendGalois ? 1 : -1, 0});
          if (i->fMix) readers[i->fMix].push_back({i, -1, 0});
        }
        if (i->fusedEpi) { readers[i->fSubFrom].push_back({i, -1, 0}); if (i->fAdd) readers[i->fAdd].push_back({i, -1, 0}); }
      }
    // (the readers will read the automorphism's SOURCE, and later than the automorphism did: nothing may write that source from the automorphism's
    // stage on — the reference's operations never write their inputs; a program that does keeps its launch)
    std::map<AddrType, size_t> lastWrite;
    for (size_t si = 0; si < st.size(); ++si)
      for (Instruction *i : st[si].ins) {
        if (dead.count(i)) continue;
        lastWrite[i->OutputOperand] = si;
        for (AddrType o : i->extraOutputs) lastWrite[o] = si;
      }
    std::map<AddrType, Instruction *> cand;   // output address -> the automorphism that every reader can read through
    bool anyOwn = false;
    for (size_t si = 0; si < st.size(); ++si)
      for (Instruction *A : st[si].ins) {
        if (A->ops != AUTO || dead.count(A) || A->galois <= 1) continue;
        { auto w = lastWrite.find(A->operandList[0]); if (w != lastWrite.end() && w->second >= si) continue; }
        auto r = readers.find(A->OutputOperand);
        if (r == readers.end() || r->second.empty()) continue;   // nobody reads it inside the op: a result
        bool ok = true;
        for (const Reader &x : r->second) ok &= x.role >= 0 && x.ins->mod_id == A->mod_id && x.ins->OutputOperand != A->operandList[0];
        if (!ok) continue;
        cand[A->OutputOperand] = A;
        for (const Reader &x : r->second) anyOwn |= x.role == 2;
      }
    if (anyOwn) {   // one Galois element per key-product launch
      uint32_t g0 = 0;
      bool uniform = true;
      for (const Reader &x : ownDigits) {
        auto c = cand.find((x.ins->ipSrc.empty() ? x.ins->ipX : x.ins->ipSrc)[x.digit]);
        if (c == cand.end() || (g0 && c->second->galois != g0)) { uniform = false; break; }
        g0 = c->second->galois;
      }
      if (!uniform)
        for (auto it = cand.begin(); it != cand.end();) {
          bool own = false;
          for (const Reader &x : readers[it->first]) own |= x.role == 2;
          it = own ? cand.erase(it) : std::next(it);
        }
    }
    for (auto &kv : cand) {
      Instruction *A = kv.second;
      auto &rd = readers[kv.first];
      for (const Reader &x : rd) {
        if (x.role == 0) { x.ins->operandList[0] = A->operandList[0]; x.ins->inGalois = A->galois; }
        else if (x.role == 1) { x.ins->fAddend = A->operandList[0]; x.ins->fAddendGalois = A->galois; }
        else {
          x.ins->ipX[x.digit] = A->operandList[0];
          if (!x.ins->ipSrc.empty()) x.ins->ipSrc[x.digit] = A->operandList[0];
          x.ins->ipXGalois = A->galois;
        }
      }
      rd.front().ins->refInstructions += A->refInstructions;
      dead.insert(A);
    }
  }
  // drop dead instructions and empty stages; upstream instructions of eliminated pass-through records are
  // accounted on the first surviving instruction so that the retired total still matches getTotalIns()
  unsigned long long orphan = 0;
  std::vector<Stage> keep;
  for (auto &s : st) {
    Stage t = s;
    t.ins.clear();
    for (Instruction *i : s.ins) {
      if (!dead.count(i)) t.ins.push_back(i);
      else if (i->passthrough) orphan += i->refInstructions;
    }
    if (!t.ins.empty()) keep.push_back(t);
  }
  if (!keep.empty()) keep[0].ins[0]->refInstructions += orphan;
  st.swap(keep);
}

// ---------------------------------------------------------------------------------------------------
// stage -> launch
// ---------------------------------------------------------------------------------------------------
void Arch::buildLaunches() {
  std::vector<Stage> st = stages;
  if (fuse) fusePasses(st);
  const unsigned long long LP = (unsigned long long)n * 8;
  // upstream issues every BCONV group to all MAC ports (include/Driver.h:307-320): same accounting here
  const unsigned long long bconvPorts = (unsigned long long)config->getValueOr("bconv_num_high", 1) * config->getValueOr("bconv_num_width", 1);
  const int useMask[9] = {3, 15, 7, 5, 5, 1, 5, 1, 13};

  // ---- 1. split every stage into parts one C-ABI call can express (same kind / opcode / direction)
  struct Part { std::string name; int key; std::vector<Instruction *> ins; int depth = 0; };
  std::vector<Part> parts;
  for (const Stage &s : st) {
    size_t first = parts.size();
    for (Instruction *i : s.ins) {
      const bool nip = i->ops == IP && std::find(i->ipCoeff.begin(), i->ipCoeff.end(), 1) != i->ipCoeff.end();
      int key = nip ? 400 + (int)i->ipX.size() * 10 + (int)i->ipY.size() : i->ops == IP && !i->ipX.empty() ? 300 + (int)i->ipX.size() * 10 + (int)i->ipY.size() : i->fusedTensor ? 200 : i->fusedSubScale ? (i->fMix ? 203 : 201) + (i->fConvIn.empty() ? 0 : 4) : i->ops == MULT ? 100 + i->opcode : (i->ops == NTT && i->passthrough) ? 100 + EWE_COPY : i->ops == AUTO ? 1000 + (int)i->galois : (int)i->ops + (i->secondOnly ? 5000 : 0);
      size_t p = first;
      for (; p < parts.size(); ++p)
        if (parts[p].key == key) break;
      if (p == parts.size()) parts.push_back(Part{s.name, key, {}, 0});
      parts[p].ins.push_back(i);
    }
  }
  // ---- 2. dependency depth of every part (RAW, WAR and WAW through limb addresses).  With fuse = 1 parts
  // of equal depth and kind are coalesced into one launch: the two keys of a ModDown, the beta digits of a
  // ModUp, D0/D2 of the tensor product ...  The reference dispatches stage by stage (Operation.cpp:947-964);
  // the stage ORDER it fixes is only a topological order of this graph.
  auto reads = [&](Instruction *i) {
    std::vector<AddrType> v;
    if (i->ops == IP && !i->ipX.empty()) {
      v = i->ipSrc.empty() ? i->ipX : i->ipSrc;
      for (size_t j = 0; j < i->ipConvIn.size(); ++j)
        if (!i->ipConvIn[j].empty()) { v[j] = i->ipConvIn[j][0]; v.insert(v.end(), i->ipConvIn[j].begin() + 1, i->ipConvIn[j].end()); }
      for (auto &y : i->ipY) v.insert(v.end(), y.begin(), y.end());
      return v;
    }
    if (i->ops == BCONV_STEP2) v.assign(i->operandList.begin(), i->operandList.end() - 1);
    else if (i->ops == MULT) { for (int b = 0; b < 4; ++b) if (useMask[i->opcode] & (1 << b)) v.push_back(i->operandList[b]); }
    else v.push_back(i->operandList[0]);
    if (i->fusedSubScale) { v.push_back(i->fMinuend); if (i->fAddend) v.push_back(i->fAddend); if (i->fMix) v.push_back(i->fMix); }
    if (!i->fConvIn.empty()) { v[0] = i->fConvIn[0]; v.insert(v.end(), i->fConvIn.begin() + 1, i->fConvIn.end()); }
    if (i->fusedEpi) { v.push_back(i->fSubFrom); if (i->fAdd) v.push_back(i->fAdd); }
    return v;
  };
  auto writes = [&](Instruction *i) {
    std::vector<AddrType> v = {i->OutputOperand};
    v.insert(v.end(), i->extraOutputs.begin(), i->extraOutputs.end());
    for (size_t j = 0; j < i->ipCoeff.size(); ++j)
      if (i->ipCoeff[j]) v.push_back(i->ipX[j]);   // first-pass scratch of a digit transformed inside the inner product
    return v;
  };
  std::map<AddrType, int> writerDepth, readerDepth;
  int serial = 0;
  for (Part &p : parts) {
    int d = 0;
    for (Instruction *i : p.ins) {
      for (AddrType a : reads(i)) { auto w = writerDepth.find(a); if (w != writerDepth.end()) d = std::max(d, w->second + 1); }
      for (AddrType o : writes(i)) {
        auto w = writerDepth.find(o); if (w != writerDepth.end()) d = std::max(d, w->second + 1);
        auto r = readerDepth.find(o); if (r != readerDepth.end()) d = std::max(d, r->second + 1);
      }
    }
    if (!fuse) d = serial++;  // unfused: one launch per stage part, upstream's order
    p.depth = d;
    for (Instruction *i : p.ins) {
      for (AddrType a : reads(i)) { int &r = readerDepth[a]; r = std::max(r, d); }
      for (AddrType o : writes(i)) writerDepth[o] = d;
    }
  }
  // ---- multi-GPU: who holds each limb-poly.  Inputs: by the modulus they were filled for; everything else: by
  // the modulus of the instruction that writes it.
  std::map<AddrType, uint32_t> ownerOfAddr;
  if (world_ > 1) {
    for (const InputFill &f : fills)
      for (size_t i = 0; i < f.addrs.size(); ++i) ownerOfAddr[f.addrs[i]] = owner(f.mods[i]);
    for (const Part &p : parts)
      for (Instruction *i : p.ins)
        for (AddrType o : writes(i)) ownerOfAddr[o] = owner(i->mod_id);
  }
  const uint32_t logLen = logN - (uint32_t)__builtin_ctz(world_);
  // ---- 3. coalesce and emit, level by level
  std::map<AddrType, int> slotOfAddr;   // pipelined sharded plan: the exchange mark behind which an address is valid on this rank
  int nextSlot = 0;
  int maxDepth = 0;
  for (const Part &p : parts) maxDepth = std::max(maxDepth, p.depth);
  for (int d = 0; d <= maxDepth; ++d) {
    std::vector<int> keysDone;
    for (size_t pi = 0; pi < parts.size(); ++pi) {
      if (parts[pi].depth != d) continue;
      const int key = parts[pi].key;
      if (std::find(keysDone.begin(), keysDone.end(), key) != keysDone.end()) continue;
      keysDone.push_back(key);
      std::vector<const Part *> group;
      for (size_t pj = pi; pj < parts.size(); ++pj)
        if (parts[pj].depth == d && parts[pj].key == key) group.push_back(&parts[pj]);
      Instruction *f = group[0]->ins[0];
      std::vector<Part> mineParts;
      bool nipSharded = false;
      if (world_ > 1 && f->ops == IP && !shardGather)
        for (const Part *g : group)
          for (Instruction *i : g->ins)
            for (auto &cin : i->ipConvIn) nipSharded |= !cin.empty();
      if (nipSharded) {
        // ---- sharded ModUp with the fused kernels (round 4).  Per digit j: limbs -> COLUMN slices of the digit's limbs (all-to-all), conversion +
        // first pass on this rank's columns for EVERY output limb (hm_bconv_col), column slices -> limbs of the first-pass hand-off
        // (all-to-all back); then ONE transform x key launch over this rank's extended limbs (second pass + MAC with both keys).
        // Every rank issues every exchange (the lists come from the global graph), also a rank that owns no extended limb.
        struct Dig { std::vector<uint32_t> in, inMods, hand, handMods; };
        std::vector<Dig> digs;
        for (const Part *g : group)
          for (Instruction *i : g->ins)
            for (size_t j = 0; j < i->ipConvIn.size(); ++j) {
              if (i->ipConvIn[j].empty()) continue;
              std::vector<uint32_t> in;
              for (AddrType x : i->ipConvIn[j]) in.push_back(limbOf(x));
              Dig *dg = nullptr;
              for (auto &q : digs) if (q.in == in && q.inMods == i->ipConvMods[j]) dg = &q;
              if (!dg) { digs.push_back(Dig{in, i->ipConvMods[j], {}, {}}); dg = &digs.back(); }
              dg->hand.push_back(limbOf(i->ipX[j]));
              dg->handMods.push_back(i->mod_id);
            }
        const uint32_t per = (uint32_t)limbIndex.size();
        std::vector<Launch *> front, back;
        std::vector<int> outSlots;
        const uint32_t nTiles = cap(logN, "cap_col_slices") / world_;   // column tiles of a limb-poly per rank
        for (size_t dj = 0; dj < digs.size(); ++dj) {
          Dig &dg = digs[dj];
          Launch *XI = new Launch, *BC = new Launch, *XO = new Launch;
          XI->kind = Launch::L_EXCH_IN_COL; BC->kind = Launch::L_BCONV_COL; XO->kind = Launch::L_EXCH_OUT_COL;
          XI->statKey = XO->statKey = "XCHG"; BC->statKey = "BCONV";
          BC->name = "ModUp_BCONV_COL_(" + std::to_string(dj) + ")";
          XI->name = BC->name + ":limbs->columns"; XO->name = BC->name + ":columns->limbs";
          for (uint32_t c = 0; c < batch_; ++c) {   // the ops of a batch share the exchanges
            for (size_t x = 0; x < dg.in.size(); ++x) { XI->exLimbs.push_back(dg.in[x] + c * per); XI->exOwners.push_back(owner(dg.inMods[x])); }
            for (size_t x = 0; x < dg.hand.size(); ++x) { XO->exLimbs.push_back(dg.hand[x] + c * per); XO->exOwners.push_back(owner(dg.handMods[x])); }
          }
          std::vector<uint32_t> inRows(XI->exLimbs.size()), outRows(XO->exLimbs.size());
          hm_slice_rows(XI->exOwners.data(), (uint32_t)inRows.size(), world_, inRows.data());
          hm_slice_rows(XO->exOwners.data(), (uint32_t)outRows.size(), world_, outRows.data());
          for (uint32_t c = 0; c < batch_; ++c) {
            Launch::Prob q{{}, dg.inMods, {}, dg.handMods};
            for (size_t x = 0; x < dg.in.size(); ++x) q.in.push_back(inRows[c * dg.in.size() + x]);
            for (size_t x = 0; x < dg.hand.size(); ++x) q.out.push_back(outRows[c * dg.hand.size() + x]);
            BC->probs.push_back(q);
          }
          BC->galois = rank_ * nTiles;    // first column tile of this rank's slice
          BC->logLen = nTiles;            // ... and how many
          BC->refInstructions = 0;        // (accounted on the transform x key launch, as in the one-GPU plan)
          XI->bytes = LP * XI->exLimbs.size() / world_;
          XO->bytes = LP * XO->exLimbs.size() / world_;
          BC->bytes = (XI->bytes + XO->bytes);
          if (pipelineDigits) {
            XI->recordSlot = nextSlot++;
            BC->waitSlots = {XI->recordSlot};
            XO->recordSlot = nextSlot++;
          }
          outSlots.push_back(XO->recordSlot);
          BC->xin = XI; BC->xout = XO;
          (pipelineDigits ? front : back).push_back(XI);
          back.push_back(BC);
          back.push_back(XO);
          algBytes += BC->bytes;
        }
        // this rank's extended limbs
        Launch *L = new Launch;
        L->kind = Launch::L_NTT_IP; L->statKey = "NTT";
        L->ipTerms = (uint32_t)f->ipX.size(); L->ipOuts = (uint32_t)f->ipY.size();
        unsigned long long lp = 0;
        for (const Part *g : group) {
          bool any = false;
          for (Instruction *i : g->ins) {
            L->refInstructions += owner(i->mod_id) == rank_ ? i->refInstructions : 0;
            if (owner(i->mod_id) != rank_) continue;
            any = true;
            for (size_t j = 0; j < i->ipX.size(); ++j) {
              const bool conv = j < i->ipConvIn.size() && !i->ipConvIn[j].empty();
              L->a.push_back(limbOf(i->ipSrc[j])); L->c.push_back(limbOf(i->ipX[j]));
              L->ipCoeff.push_back(conv ? 2 : i->ipCoeff[j]);
              lp += i->ipCoeff[j] ? 1 : 1;
            }
            for (auto &y : i->ipY) for (AddrType yy : y) L->b.push_back(limbOf(yy));
            L->out.push_back(limbOf(i->OutputOperand));
            for (AddrType o : i->extraOutputs) L->out.push_back(limbOf(o));
            L->mods.push_back(i->mod_id);
            if (i->ipXGalois) L->xGalois = i->ipXGalois;   // (12)
            lp += (unsigned long long)L->ipTerms * L->ipOuts + L->ipOuts;
          }
          if (any) L->name += (L->name.empty() ? "" : "+") + g->name;
        }
        L->bytes = lp * LP;
        if (pipelineDigits) for (int sl : outSlots) L->waitSlots.push_back(sl);
        launches.insert(launches.end(), front.begin(), front.end());
        launches.insert(launches.end(), back.begin(), back.end());
        if (!L->mods.empty()) { algBytes += L->bytes; launches.push_back(L); }
        else {   // a rank without extended limbs still waits for nothing: its exchanges are complete on their own stream
          delete L;
        }
        continue;
      }
      if (world_ > 1 && (f->ops != BCONV_STEP2 || shardGather)) {
        // (a) operands written on another rank (the rescale's r = INTT(x_last); with the gather plan also the conversions' inputs): replicate them first — every
        //     rank derives the same list from the global graph, so the collective is entered by all
        std::vector<AddrType> need;
        for (const Part *g : group)
          for (Instruction *i : g->ins)
            for (AddrType a : reads(i)) {
              auto oo = ownerOfAddr.find(a);
              if (oo != ownerOfAddr.end() && oo->second != owner(i->mod_id) && std::find(need.begin(), need.end(), a) == need.end()) need.push_back(a);
            }
        if (!need.empty()) {
          Launch *R = new Launch;
          R->kind = Launch::L_REPLICATE; R->statKey = "XCHG"; R->name = "replicate";
          for (AddrType a : need) { R->exLimbs.push_back(limbOf(a)); R->exOwners.push_back(ownerOfAddr[a]); }
          R->bytes = LP * need.size();
          if (pipelineDigits) {
            for (AddrType a : need) { auto it = slotOfAddr.find(a); if (it != slotOfAddr.end() && std::find(R->waitSlots.begin(), R->waitSlots.end(), it->second) == R->waitSlots.end()) R->waitSlots.push_back(it->second); }
            R->recordSlot = nextSlot++;
            for (AddrType a : need) slotOfAddr[a] = R->recordSlot;
          }
          launches.push_back(R);
        }
        // (b) keep the instructions whose modulus this rank owns
        for (const Part *g : group) {
          Part m{g->name, g->key, {}, g->depth};
          for (Instruction *i : g->ins)
            if (owner(i->mod_id) == rank_) m.ins.push_back(i);
          if (!m.ins.empty()) mineParts.push_back(m);
        }
        group.clear();
        for (const Part &m : mineParts) group.push_back(&m);
        if (group.empty()) continue;
        f = group[0]->ins[0];
      }
      // pipelined sharded plan: parts that depend on different exchanges (digit j's transforms read what exchange XO_j delivered)
      // become launches of their own, each waiting for its own mark; the conversions are split by input basis (= by digit; the two
      // keys of a ModDown share theirs and stay together).  Exchange-in launches of all digits are issued first.
      std::vector<std::vector<const Part *>> subgroups;
      auto slotsOf = [&](const Part *g) {
        std::set<int> ss;
        for (Instruction *i : g->ins)
          for (AddrType a : reads(i)) { auto it = slotOfAddr.find(a); if (it != slotOfAddr.end()) ss.insert(it->second); }
        return ss;
      };
      if (pipelineDigits) {
        std::vector<std::set<int>> sigs;
        std::vector<std::vector<uint32_t>> bases;
        for (const Part *g : group) {
          const std::set<int> sg = slotsOf(g);
          const std::vector<uint32_t> bs = f->ops == BCONV_STEP2 ? g->ins[0]->inMods : std::vector<uint32_t>();
          size_t k = 0;
          for (; k < subgroups.size(); ++k) if (sigs[k] == sg && bases[k] == bs) break;
          if (k == subgroups.size()) { subgroups.emplace_back(); sigs.push_back(sg); bases.push_back(bs); }
          subgroups[k].push_back(g);
        }
      } else subgroups.push_back(group);
      std::vector<Launch *> front, back;
      for (const std::vector<const Part *> &subgroup : subgroups) {
      const std::vector<const Part *> &group = subgroup;
      f = group[0]->ins[0];
      Launch *L = new Launch;
      if (pipelineDigits)
        for (const Part *g : group) for (int sl : slotsOf(g)) if (std::find(L->waitSlots.begin(), L->waitSlots.end(), sl) == L->waitSlots.end()) L->waitSlots.push_back(sl);
      for (const Part *g : group) L->name += (L->name.empty() ? "" : "+") + g->name;
      size_t count = 0;
      for (const Part *g : group)
        for (Instruction *i : g->ins) { L->refInstructions += i->refInstructions * (i->ops == BCONV_STEP2 ? bconvPorts : 1ull) + i->refExtra; ++count; }
      if (f->ops == IP && std::find(f->ipCoeff.begin(), f->ipCoeff.end(), 1) != f->ipCoeff.end()) {
        L->kind = Launch::L_NTT_IP; L->statKey = "NTT";
        L->ipTerms = (uint32_t)f->ipX.size(); L->ipOuts = (uint32_t)f->ipY.size();
        unsigned long long lp = 0;
        for (const Part *g : group)
          for (Instruction *i : g->ins) {
            for (size_t j = 0; j < i->ipX.size(); ++j) {
              L->a.push_back(limbOf(i->ipSrc[j])); L->c.push_back(limbOf(i->ipX[j])); L->ipCoeff.push_back(i->ipCoeff[j]);
              lp += i->ipCoeff[j] ? 3 : 1;       // transformed digit: source read, hand-off written and read; own limb: read
            }
            for (size_t j = 0; j < i->ipConvIn.size(); ++j) {   // (8): the digit's conversion runs inside its first pass
              if (i->ipConvIn[j].empty()) continue;
              std::vector<uint32_t> in;
              for (AddrType x : i->ipConvIn[j]) in.push_back(limbOf(x));
              Launch::Prob *pr = nullptr;
              const bool pk = j < i->ipConvPacked.size() && i->ipConvPacked[j];
              for (auto &q : L->probs) if (q.in == in && q.inMods == i->ipConvMods[j] && q.inPacked == pk) pr = &q;
              if (!pr) { L->probs.push_back(Launch::Prob{in, i->ipConvMods[j], {}, {}}); pr = &L->probs.back(); pr->inPacked = pk; }
              pr->out.push_back(limbOf(i->ipX[j]));      // the hand-off limb of (limb, digit)
              pr->outMods.push_back(i->mod_id);
              lp -= 1;                                      // the converted limb is neither written nor read: source = the conversion's inputs
            }
            for (auto &y : i->ipY) for (AddrType yy : y) L->b.push_back(limbOf(yy));
            L->out.push_back(limbOf(i->OutputOperand));
            for (AddrType o : i->extraOutputs) L->out.push_back(limbOf(o));
            L->mods.push_back(i->mod_id);
            L->ipInv.push_back(i->ipInvOut ? 1 : 0);
            if (i->ipXGalois) L->xGalois = i->ipXGalois;   // (12): uniform over the op's records by construction
            lp += (unsigned long long)L->ipTerms * L->ipOuts + L->ipOuts;
          }
        for (auto &q : L->probs) lp += q.in.size();
        L->bytes = lp * LP;
      } else if (f->ops == IP && !f->ipX.empty()) {
        L->kind = Launch::L_IP; L->statKey = "EWE";
        L->ipTerms = (uint32_t)f->ipX.size(); L->ipOuts = (uint32_t)f->ipY.size();
        for (const Part *g : group)
          for (Instruction *i : g->ins) {
            for (AddrType x : i->ipX) L->a.push_back(limbOf(x));
            for (auto &y : i->ipY) for (AddrType yy : y) L->b.push_back(limbOf(yy));
            L->out.push_back(limbOf(i->OutputOperand));
            for (AddrType o : i->extraOutputs) L->out.push_back(limbOf(o));
            L->mods.push_back(i->mod_id);
            if (i->ipXGalois) L->xGalois = i->ipXGalois;   // (12): uniform over the op's records by construction
          }
        L->bytes = (unsigned long long)(L->ipTerms * (1 + L->ipOuts) + L->ipOuts) * LP * count;
      } else if (f->fusedTensor) {
        L->kind = Launch::L_TENSOR; L->statKey = "EWE";
        for (const Part *g : group)
          for (Instruction *i : g->ins) {  // a = c00 (P), b = c10 (T), c = c01 (R), d = c11 (S)
            L->a.push_back(limbOf(i->operandList[0])); L->b.push_back(limbOf(i->operandList[3]));
            L->c.push_back(limbOf(i->operandList[2])); L->d.push_back(limbOf(i->operandList[1]));
            L->out.push_back(limbOf(i->extraOutputs[0])); L->out1.push_back(limbOf(i->OutputOperand)); L->out2.push_back(limbOf(i->extraOutputs[1]));
            L->mods.push_back(i->mod_id);
          }
        L->bytes = 7 * LP * count;
      } else if (f->fusedSubScale) {
        L->kind = Launch::L_NTT_SUBSCALE; L->statKey = "NTT";
        bool anyAddend = false;   // the addend is per limb-poly (hrotate: key 0 adds the rotated c0, key 1 nothing)
        bool anyAddGalois = false;   // (12) ... and key 0's addend through the automorphism
        for (const Part *g : group)
          for (Instruction *i : g->ins) { anyAddend |= i->fAddend != 0; anyAddGalois |= i->fAddendGalois != 0; }
        for (const Part *g : group)
          for (Instruction *i : g->ins) {
            L->a.push_back(limbOf(i->operandList[0])); L->b.push_back(limbOf(i->fMinuend));
            if (anyAddend) L->c.push_back(i->fAddend ? limbOf(i->fAddend) : HM_NO_LIMB);
            if (anyAddGalois) L->addGalois.push_back(i->fAddendGalois);
            L->out.push_back(limbOf(i->OutputOperand)); L->mods.push_back(i->mod_id); L->k.push_back(i->constant);
            if (f->fMix) {  // the part key keeps merged and plain records apart
              L->d.push_back(limbOf(i->fMix)); L->mixK.push_back(i->fMixConst);
              if (anyAddend) L->addK.push_back(i->fAddendConst ? i->fAddendConst : 1);
            }
          }
        L->hasK = true;
        L->bytes = (anyAddend ? 4 : 3) * LP * count;
        for (const Part *g : group)
          for (Instruction *i : g->ins) {   // (9): the conversion of this limb-poly runs inside its first pass
            if (i->fConvIn.empty()) continue;
            std::vector<uint32_t> in;
            for (AddrType x : i->fConvIn) in.push_back(limbOf(x));
            Launch::Prob *pr = nullptr;
            for (auto &q : L->probs) if (q.in == in && q.inMods == i->fConvMods && q.inPacked == i->inPacked) pr = &q;
            if (!pr) { L->probs.push_back(Launch::Prob{in, i->fConvMods, {}, {}}); pr = &L->probs.back(); pr->inPacked = i->inPacked; L->bytes += LP * in.size(); }
            pr->out.push_back(limbOf(i->OutputOperand));   // the hand-off lands in the output limb
            pr->outMods.push_back(i->mod_id);
            L->bytes -= LP;                                  // the converted limb-poly is neither written nor read
          }
      } else if (f->ops == NTT && f->passthrough) {  // unfused mode: materialise the copy
        L->kind = Launch::L_EWE; L->opcode = EWE_COPY; L->statKey = "EWE";
        for (const Part *g : group)
          for (Instruction *i : g->ins) { L->a.push_back(limbOf(i->operandList[0])); L->out.push_back(limbOf(i->OutputOperand)); L->mods.push_back(i->mod_id); }
        L->bytes = 2 * LP * count;
      } else if (f->ops == NTT || f->ops == INTT) {
        L->kind = f->ops == NTT ? Launch::L_NTT : Launch::L_INTT; L->statKey = "NTT";
        L->secondOnly = f->secondOnly;
        bool anyInGalois = false;
        for (const Part *g : group)
          for (Instruction *i : g->ins) anyInGalois |= i->inGalois != 0;
        for (const Part *g : group)
          for (Instruction *i : g->ins) {
            if (anyInGalois) L->inGalois.push_back(i->inGalois);
            L->a.push_back(limbOf(i->operandList[0])); L->out.push_back(limbOf(i->OutputOperand)); L->mods.push_back(i->mod_id);
            L->k.push_back(i->hasConstant ? i->constant : 1);
            L->hasK |= i->hasConstant;
            if (f->ops == INTT) L->outPacked.push_back(i->packedOut ? 1 : 0);
          }
        L->bytes = 2 * LP * count;
      } else if (f->ops == AUTO) {
        L->kind = Launch::L_AUTO; L->statKey = "AUTO"; L->galois = f->galois;
        for (const Part *g : group)
          for (Instruction *i : g->ins) { L->a.push_back(limbOf(i->operandList[0])); L->out.push_back(limbOf(i->OutputOperand)); }
        L->bytes = 2 * LP * count;
      } else if (f->ops == MULT) {
        L->kind = Launch::L_EWE; L->opcode = f->opcode; L->statKey = "EWE";
        const int m = useMask[f->opcode];
        int nops = 1;
        for (int b = 0; b < 4; ++b) nops += (m >> b) & 1;
        // entries of one modulus side by side: records that share an operand (pmult: c0 x pt and c1 x pt of a limb) then sit 128 workgroups apart
        // in dispatch order, on the same XCD, and the second reader finds the shared limb-poly in L2 instead of fetching it again
        std::vector<Instruction *> recs;
        for (const Part *g : group) recs.insert(recs.end(), g->ins.begin(), g->ins.end());
        std::stable_sort(recs.begin(), recs.end(), [](const Instruction *x, const Instruction *y) { return x->mod_id < y->mod_id; });
        for (Instruction *i : recs) {
          auto get = [&](int b) { return (m & (1 << b)) ? limbOf(i->operandList[b]) : 0u; };
          L->a.push_back(get(0)); L->b.push_back(get(1)); L->c.push_back(get(2)); L->d.push_back(get(3));
          L->out.push_back(limbOf(i->OutputOperand)); L->mods.push_back(i->mod_id);
          L->k.push_back(i->hasConstant ? i->constant : 0);
          L->hasK |= i->hasConstant;
        }
        L->bytes = (unsigned long long)nops * LP * count;
      } else if (f->ops == BCONV_STEP2) {
        L->kind = Launch::L_BCONV; L->statKey = "BCONV";
        for (const Part *g : group) {
          // one conversion per distinct (input limbs) inside the part
          for (Instruction *i : g->ins) {
            std::vector<uint32_t> in;
            for (size_t x = 0; x + 1 < i->operandList.size(); ++x) in.push_back(limbOf(i->operandList[x]));
            Launch::Prob *pr = nullptr;
            for (auto &q : L->probs) if (q.in == in && q.inMods == i->inMods && q.epi == i->fusedEpi && q.epAdd == (i->fAdd != 0) && q.inPacked == i->inPacked) pr = &q;
            if (!pr) { L->probs.push_back(Launch::Prob{in, i->inMods, {}, {}}); pr = &L->probs.back(); pr->epi = i->fusedEpi; pr->epAdd = i->fAdd != 0; pr->inPacked = i->inPacked; }
            pr->out.push_back(limbOf(i->OutputOperand));
            pr->outMods.push_back(i->mod_id);
            if (i->fusedEpi) {   // (10): out = (fSubFrom - conv) * k [+ fAdd]
              pr->epA.push_back(limbOf(i->fSubFrom)); pr->epK.push_back(i->constant);
              if (i->fAdd) pr->epB.push_back(limbOf(i->fAdd));
              L->bytes += LP * (i->fAdd ? 2 : 1);
            }
          }
        }
        if (world_ > 1 && batch_ > 1 && !shardGather) {  // sharded batch: the ops of the batch share the exchanges around the conversion
          const uint32_t per = (uint32_t)limbIndex.size();
          const size_t p0 = L->probs.size();
          for (uint32_t c = 1; c < batch_; ++c)
            for (size_t i = 0; i < p0; ++i) {
              Launch::Prob q = L->probs[i];
              for (uint32_t &x : q.in) x += c * per;
              for (uint32_t &x : q.out) x += c * per;
              L->probs.push_back(q);
            }
          L->refInstructions *= batch_;
        }
        for (auto &q : L->probs) L->bytes += LP * (q.in.size() + q.out.size());
        if (world_ > 1 && !shardGather) {
          // limb-sharded -> coefficient slices -> convert every output on this rank's slice -> limb-sharded
          Launch *XI = new Launch, *XO = new Launch;
          XI->kind = Launch::L_EXCH_IN; XO->kind = Launch::L_EXCH_OUT; XI->statKey = XO->statKey = "XCHG";
          XI->name = L->name + ":limbs->slices"; XO->name = L->name + ":slices->limbs";
          std::vector<uint32_t> inOwn, outOwn;
          for (auto &q : L->probs) {
            for (size_t x = 0; x < q.in.size(); ++x) {
              size_t pos = std::find(XI->exLimbs.begin(), XI->exLimbs.end(), q.in[x]) - XI->exLimbs.begin();
              if (pos == XI->exLimbs.size()) { XI->exLimbs.push_back(q.in[x]); XI->exOwners.push_back(owner(q.inMods[x])); }
            }
            for (size_t x = 0; x < q.out.size(); ++x) { XO->exLimbs.push_back(q.out[x]); XO->exOwners.push_back(owner(q.outMods[x])); }
          }
          std::vector<uint32_t> inRows(XI->exLimbs.size()), outRows(XO->exLimbs.size());
          hm_slice_rows(XI->exOwners.data(), (uint32_t)inRows.size(), world_, inRows.data());
          hm_slice_rows(XO->exOwners.data(), (uint32_t)outRows.size(), world_, outRows.data());
          size_t o = 0;
          for (auto &q : L->probs) {
            for (uint32_t &x : q.in) x = inRows[std::find(XI->exLimbs.begin(), XI->exLimbs.end(), x) - XI->exLimbs.begin()];
            for (uint32_t &x : q.out) x = outRows[o++];
          }
          L->logLen = XI->logLen = XO->logLen = logLen;
          XI->bytes = LP * XI->exLimbs.size() / world_;
          XO->bytes = LP * XO->exLimbs.size() / world_;
          L->bytes /= world_;
          if (pipelineDigits) {
            XI->waitSlots = L->waitSlots;   // (its inputs come from the compute stream; marks matter only if an exchange produced them)
            XI->recordSlot = nextSlot++;
            L->waitSlots = {XI->recordSlot};
            XO->recordSlot = nextSlot++;
            for (const Part *g : group)
              for (Instruction *i : g->ins) slotOfAddr[i->OutputOperand] = XO->recordSlot;
          }
          L->xin = XI; L->xout = XO;
          (pipelineDigits ? front : back).push_back(XI);
          algBytes += L->bytes;
          back.push_back(L);
          back.push_back(XO);
          continue;
        }
      } else {
        delete L;
        throw std::runtime_error("no unit executes op " + f->GetOpName());
      }
      algBytes += L->bytes;
      back.push_back(L);
      }  // subgroups
      launches.insert(launches.end(), front.begin(), front.end());
      launches.insert(launches.end(), back.begin(), back.end());
    }
  }
}

// batch > 1: every launch carries the limb-polys of `batch` independent ops.  Op c lives in its own copy of the
// buffer plan (limb index + c * limbs-per-op) with its own inputs (fill seed + c * kBatchSeedStride); buffers
// marked shared (the evaluation key: all ops of a batch are under the same key) exist once.  Larger launches
// amortise the per-launch latency and pair up same-modulus limb-polys across ops (one twiddle fetch per pair).
static const uint64_t kBatchSeedStride = 100000;
void Arch::replicateForBatch() {
  const uint32_t per = (uint32_t)limbIndex.size();
  std::set<uint32_t> sharedLimbs;
  for (const InputFill &f : fills)
    if (f.shared)
      for (AddrType a : f.addrs) sharedLimbs.insert(limbOf(a));
  auto rep = [&](std::vector<uint32_t> &v, bool isLimb) {
    const size_t n0 = v.size();
    for (uint32_t c = 1; c < batch_; ++c)
      for (size_t i = 0; i < n0; ++i) v.push_back(isLimb && v[i] != HM_NO_LIMB && !sharedLimbs.count(v[i]) ? v[i] + c * per : v[i]);
  };
  for (Launch *l : launches) {
    if (world_ > 1 && !shardGather && (l->kind == Launch::L_BCONV || l->kind == Launch::L_EXCH_IN || l->kind == Launch::L_EXCH_OUT)) continue;  // built batched
    if (l->kind == Launch::L_BCONV_COL || l->kind == Launch::L_EXCH_IN_COL || l->kind == Launch::L_EXCH_OUT_COL) continue;           // built batched
    if (l->kind == Launch::L_REPLICATE) {
      const size_t n0 = l->exLimbs.size();
      for (uint32_t c = 1; c < batch_; ++c)
        for (size_t i = 0; i < n0; ++i) { l->exLimbs.push_back(l->exLimbs[i] + c * per); l->exOwners.push_back(l->exOwners[i]); }
      l->bytes *= batch_;
      continue;
    }
    if (l->kind == Launch::L_IP || l->kind == Launch::L_NTT_IP) {
      // entry e: ipTerms x limbs, ipTerms * ipOuts y limbs, ipOuts outputs.  Entry-major order (entry e of every op
      // side by side): the ops share the key limbs, so the second and later readers of a key chunk find it in L2
      auto inter = [&](std::vector<uint32_t> &v, size_t width, bool isLimb) {
        const size_t n0 = v.size() / width;
        std::vector<uint32_t> o;
        for (size_t e = 0; e < n0; ++e)
          for (uint32_t c = 0; c < batch_; ++c)
            for (size_t w = 0; w < width; ++w) {
              const uint32_t x = v[e * width + w];
              o.push_back(isLimb && c && !sharedLimbs.count(x) ? x + c * per : x);
            }
        v.swap(o);
      };
      inter(l->a, l->ipTerms, true); inter(l->b, (size_t)l->ipTerms * l->ipOuts, true); inter(l->out, l->ipOuts, true); inter(l->mods, 1, false);
      if (l->kind == Launch::L_NTT_IP) {
        const size_t p0 = l->probs.size();
        for (uint32_t c = 1; c < batch_; ++c)
          for (size_t i = 0; i < p0; ++i) {
            Launch::Prob q = l->probs[i];
            for (uint32_t &x : q.in) x += c * per;
            for (uint32_t &x : q.out) x += c * per;
            for (uint32_t &x : q.epA) x += c * per;
            for (uint32_t &x : q.epB) x += c * per;
            l->probs.push_back(q);
          }
        inter(l->c, l->ipTerms, true);
        std::vector<uint32_t> f(l->ipCoeff.begin(), l->ipCoeff.end());
        inter(f, l->ipTerms, false);
        l->ipCoeff.assign(f.begin(), f.end());
        std::vector<uint32_t> fi(l->ipInv.begin(), l->ipInv.end());
        inter(fi, 1, false);
        l->ipInv.assign(fi.begin(), fi.end());
      }
    } else {
      {
        const size_t n0 = l->outPacked.size();
        for (uint32_t c = 1; c < batch_; ++c)
          for (size_t i = 0; i < n0; ++i) l->outPacked.push_back(l->outPacked[i]);
      }
      rep(l->a, true); rep(l->b, true); rep(l->c, true); rep(l->d, true);
      rep(l->out, true); rep(l->out1, true); rep(l->out2, true); rep(l->mods, false);
      rep(l->inGalois, false); rep(l->addGalois, false);
      for (std::vector<uint64_t> *kv : {&l->k, &l->mixK, &l->addK}) {
        const size_t k0 = kv->size();
        for (uint32_t c = 1; c < batch_; ++c)
          for (size_t i = 0; i < k0; ++i) kv->push_back((*kv)[i]);
      }
      const size_t p0 = l->probs.size();
      for (uint32_t c = 1; c < batch_; ++c)
        for (size_t i = 0; i < p0; ++i) {
          Launch::Prob q = l->probs[i];
          for (uint32_t &x : q.in) x += c * per;
          for (uint32_t &x : q.out) x += c * per;
          for (uint32_t &x : q.epA) x += c * per;
          for (uint32_t &x : q.epB) x += c * per;
          l->probs.push_back(q);
        }
    }
    l->refInstructions *= batch_;
    l->bytes *= batch_;
  }
  algBytes *= batch_;
}

void Arch::prepare() {
  if (prepared) return;
  prepared = true;
  buildLaunches();
  if (batch_ > 1) {
    replicateForBatch();
    stat->setStat("Batch", batch_);
  }
  stat->setStat("Launches", launches.size());
  stat->setStat("LimbPolys_resident", limbIndex.size());
  stat->setStat("HBM_stage_bytes", algBytes);
  if (backendKind != BACKEND_HIP) return;
  const size_t bytes = (size_t)limbIndex.size() * batch_ * n * 8;
  void *p = nullptr;
  if (hm_malloc(ctx, bytes, &p) != HM_OK) throw std::runtime_error(std::string("hm_malloc: ") + hm_last_error(ctx));
  pool = static_cast<uint64_t *>(p);
  if (world_ > 1) {
    if (!commReady) throw std::runtime_error("world > 1 but no transport was set (commInitRccl / commInitExternal)");
    if (pipelineDigits && hm_exchange_stream(ctx, 1) != HM_OK) throw std::runtime_error(std::string("hm_exchange_stream: ") + hm_last_error(ctx));
    for (Launch *bc : launches) {
      if (bc->kind != Launch::L_BCONV_COL || !bc->xin) continue;
      Launch *xi = bc->xin, *xo = bc->xout;
      void *si = nullptr, *so = nullptr;   // limb-poly layout: a row per limb-poly, this rank's columns valid
      if (hm_malloc(ctx, (size_t)xi->exLimbs.size() * n * 8, &si) != HM_OK || hm_malloc(ctx, (size_t)xo->exLimbs.size() * n * 8, &so) != HM_OK)
        throw std::runtime_error(std::string("hm_malloc (column slices): ") + hm_last_error(ctx));
      sliceBuffers.push_back(si); sliceBuffers.push_back(so);
      xi->slicesIn = bc->slicesIn = static_cast<uint64_t *>(si);
      bc->slicesOut = xo->slicesOut = static_cast<uint64_t *>(so);
    }
    for (Launch *bc : launches) {
      if (bc->kind != Launch::L_BCONV || !bc->xin) continue;
      Launch *xi = bc->xin, *xo = bc->xout;
      void *si = nullptr, *so = nullptr;
      if (hm_malloc(ctx, (size_t)xi->exLimbs.size() * (n / world_) * 8, &si) != HM_OK || hm_malloc(ctx, (size_t)xo->exLimbs.size() * (n / world_) * 8, &so) != HM_OK)
        throw std::runtime_error(std::string("hm_malloc (slices): ") + hm_last_error(ctx));
      sliceBuffers.push_back(si); sliceBuffers.push_back(so);
      xi->slicesIn = bc->slicesIn = static_cast<uint64_t *>(si);
      bc->slicesOut = xo->slicesOut = static_cast<uint64_t *>(so);
    }
  }
  std::set<AddrType> bound;
  for (Binding &b : bindings) {
    if (!b.src->prepared) throw std::runtime_error("bindInput: prepare the producer first");
    for (size_t i = 0; i < b.dst.size(); ++i)
      for (uint32_t c = 0; c < batch_; ++c) {
        b.dstLimbs.push_back(limbOf(b.dst[i]) + c * (uint32_t)limbIndex.size());
        b.srcLimbs.push_back(b.src->limbOf(b.srcAddrs[i]) + c * (uint32_t)b.src->limbIndex.size());
        b.mods.push_back(0);
      }
    bound.insert(b.dst.begin(), b.dst.end());
  }
  for (const InputFill &f : fills)
    for (uint32_t c = 0; c < (f.shared ? 1u : batch_); ++c) {
      if (!f.addrs.empty() && bound.count(f.addrs[0])) continue;  // this input comes from another op
      std::vector<uint32_t> limbs;
      for (AddrType a : f.addrs) limbs.push_back(limbOf(a) + c * (uint32_t)limbIndex.size());
      if (hm_fill_uniform(ctx, pool, limbs.data(), f.mods.data(), (uint32_t)limbs.size(), f.seed + c * kBatchSeedStride) != HM_OK)
        throw std::runtime_error(std::string("hm_fill_uniform: ") + hm_last_error(ctx));
    }
  hm_sync(ctx);
}

static const char *const kLaunchKindNames[] = {"NTT", "INTT", "EWE", "BCONV", "AUTO", "NTT_SUBSCALE", "TENSOR", "EXCH_IN", "EXCH_OUT", "REPLICATE", "IP", "NTT_IP",
                                               "BCONV_COL", "EXCH_IN_COL", "EXCH_OUT_COL"};

// Per-launch device time (SURVEY.md §8d "per-stage hipEvent times", exchange time at N > 1): every launch of the plan
// bracketed by its own event pair, in plan order so that the data dependencies (and, sharded, the collectives) line up.
std::string Arch::stageTimes(uint32_t iters) {
  if (!prepared) prepare();
  if (backendKind != BACKEND_HIP) return "";
  std::vector<uint64_t> total(launches.size(), 0);
  for (uint32_t it = 0; it < iters; ++it)
    for (size_t i = 0; i < launches.size(); ++i) {
      hm_timer_start(ctx);
      enqueue(*launches[i]);
      if (launches[i]->recordSlot >= 0) hm_exchange_wait(ctx, (uint32_t)launches[i]->recordSlot);   // timed alone: the compute stream's timer covers the exchange
      uint64_t ns = 0;
      hm_timer_stop(ctx, &ns);
      total[i] += ns;
    }
  std::string out;
  for (size_t i = 0; i < launches.size(); ++i)
    out += std::string(kLaunchKindNames[launches[i]->kind]) + " " + launches[i]->name + " " + std::to_string(total[i] / (iters ? iters : 1)) + "\n";
  return out;
}

std::string Arch::planText() const {
  const char *const *names = kLaunchKindNames;
  std::string out;
  for (const Launch *l : launches) {
    size_t cnt = l->out.size();
    if (l->kind == Launch::L_BCONV || l->kind == Launch::L_BCONV_COL) { cnt = 0; for (auto &q : l->probs) cnt += q.out.size(); }
    if (l->kind == Launch::L_IP || l->kind == Launch::L_NTT_IP) cnt = l->mods.size();
    out += std::string(names[l->kind]) + " " + l->name + " n=" + std::to_string(cnt) + " ref=" + std::to_string(l->refInstructions);
    // pass 11: limb-polys an inverse transform stores split-30 packed / conversions (separate or inside a transform's first pass) that read packed inputs
    const size_t po = (size_t)std::count(l->outPacked.begin(), l->outPacked.end(), 1), pi = (size_t)std::count_if(l->probs.begin(), l->probs.end(), [](const Launch::Prob &q) { return q.inPacked; });
    if (po) out += " packed_out=" + std::to_string(po);
    if (pi) out += " packed_in=" + std::to_string(pi) + "/" + std::to_string(l->probs.size());
    if (l->secondOnly) out += " second_pass_only";
    // pass 12: limb-polys whose input (INTT) / addend (fused forward transform) is read through an automorphism
    for (const auto *gv : {&l->inGalois, &l->addGalois}) {
      uint32_t cnt = 0, g = 0;
      for (uint32_t x : *gv) if (x > 1) { ++cnt; g = x; }
      if (cnt) out += std::string(gv == &l->inGalois ? " auto_in=" : " auto_addend=") + std::to_string(cnt) + "/g" + std::to_string(g);
    }
    if (l->xGalois) out += " auto_x=g" + std::to_string(l->xGalois);
    if (l->recordSlot >= 0) out += " mark=" + std::to_string(l->recordSlot);
    if (!l->waitSlots.empty()) { out += " wait="; for (int w : l->waitSlots) out += std::to_string(w) + ","; }
    if (!l->exLimbs.empty()) {
      out += " limbs=";
      for (size_t i = 0; i < l->exLimbs.size(); ++i) out += std::to_string(l->exLimbs[i]) + ":" + std::to_string(l->exOwners[i]) + ",";
    }
    out += "\n";
  }
  return out;
}

void Arch::enqueue(Launch &l) {
  hm_status st = HM_OK;
  for (int sl : l.waitSlots)
    if (hm_exchange_wait(ctx, (uint32_t)sl) != HM_OK) throw std::runtime_error("stage " + l.name + ": " + hm_last_error(ctx));
  const uint32_t cnt = (uint32_t)l.out.size();
  std::vector<hm_bconv_desc> descs;
  switch (l.kind) {
  case Launch::L_NTT:
    st = hm_ntt(ctx, pool, l.a.data(), pool, l.out.data(), l.mods.data(), cnt, 0, nullptr);
    break;
  case Launch::L_INTT: {
    const bool anyPacked = std::find(l.outPacked.begin(), l.outPacked.end(), 1) != l.outPacked.end();
    hm_ntt_desc d = {pool, l.a.data(), pool, l.out.data(), l.mods.data(), cnt, 1, l.hasK ? l.k.data() : nullptr, l.secondOnly ? 1 : 0,
                     anyPacked ? l.outPacked.data() : nullptr, l.inGalois.empty() ? nullptr : l.inGalois.data()};
    st = hm_ntt_ex(ctx, &d);
    break;
  }
  case Launch::L_NTT_SUBSCALE:
    if (!l.mixK.empty() || !l.probs.empty() || !l.addGalois.empty()) {
      for (auto &q : l.probs)
        descs.push_back(hm_bconv_desc{pool, q.in.data(), q.inMods.data(), (uint32_t)q.in.size(), pool, q.out.data(), q.outMods.data(), (uint32_t)q.out.size(), 0,
                                    nullptr, nullptr, nullptr, nullptr, nullptr, q.inPacked ? 1u : 0u});
      hm_ntt_fused_desc d = {pool, l.a.data(), l.mixK.empty() ? nullptr : pool, l.mixK.empty() ? nullptr : l.d.data(), l.mixK.empty() ? nullptr : l.mixK.data(), pool, l.b.data(),
                             l.c.empty() ? nullptr : pool, l.c.empty() ? nullptr : l.c.data(), l.addK.empty() ? nullptr : l.addK.data(), pool, l.out.data(), l.mods.data(), cnt,
                             l.k.data(), descs.empty() ? nullptr : descs.data(), (uint32_t)descs.size(), l.addGalois.empty() ? nullptr : l.addGalois.data()};
      st = hm_ntt_mix_sub_scale(ctx, &d);
    } else {
      st = hm_ntt_sub_scale(ctx, pool, l.a.data(), pool, l.b.data(), l.c.empty() ? nullptr : pool, l.c.empty() ? nullptr : l.c.data(), pool,
                            l.out.data(), l.mods.data(), cnt, l.k.data());
    }
    break;
  case Launch::L_IP:
    if (l.xGalois) {
      const hm_ip_desc d = {pool, l.a.data(), pool, l.b.data(), pool, l.out.data(), l.mods.data(), (uint32_t)l.mods.size(), l.ipTerms, l.ipOuts, l.xGalois};
      st = hm_inner_product_ex(ctx, &d);
    } else
    st = hm_inner_product(ctx, pool, l.a.data(), pool, l.b.data(), pool, l.out.data(), l.mods.data(), (uint32_t)l.mods.size(), l.ipTerms, l.ipOuts);
    break;
  case Launch::L_NTT_IP: {
    for (auto &q : l.probs)
      descs.push_back(hm_bconv_desc{pool, q.in.data(), q.inMods.data(), (uint32_t)q.in.size(), pool, q.out.data(), q.outMods.data(), (uint32_t)q.out.size(), 0,
                                    nullptr, nullptr, nullptr, nullptr, nullptr, q.inPacked ? 1u : 0u});
    hm_ntt_ip_desc d = {pool, l.a.data(), l.ipCoeff.data(), pool, l.c.data(), pool, l.b.data(), pool, l.out.data(), l.mods.data(),
                        (uint32_t)l.mods.size(), l.ipTerms, l.ipOuts, descs.empty() ? nullptr : descs.data(), (uint32_t)descs.size(),
                        std::find(l.ipInv.begin(), l.ipInv.end(), 1) != l.ipInv.end() ? l.ipInv.data() : nullptr, l.xGalois};
    st = hm_ntt_inner_product(ctx, &d);
    break;
  }
  case Launch::L_TENSOR:
    st = hm_tensor(ctx, pool, l.a.data(), pool, l.b.data(), pool, l.c.data(), pool, l.d.data(), pool, l.out.data(), pool, l.out1.data(), pool,
                   l.out2.data(), l.mods.data(), cnt);
    break;
  case Launch::L_EXCH_IN:
    st = hm_limbs_to_slices(ctx, pool, l.exLimbs.data(), l.exOwners.data(), (uint32_t)l.exLimbs.size(), l.slicesIn);
    break;
  case Launch::L_EXCH_OUT:
    st = hm_slices_to_limbs(ctx, l.slicesOut, pool, l.exLimbs.data(), l.exOwners.data(), (uint32_t)l.exLimbs.size());
    break;
  case Launch::L_EXCH_IN_COL:
    st = hm_limbs_to_colslices(ctx, pool, l.exLimbs.data(), l.exOwners.data(), (uint32_t)l.exLimbs.size(), l.slicesIn);
    break;
  case Launch::L_EXCH_OUT_COL:
    st = hm_colslices_to_limbs(ctx, l.slicesOut, pool, l.exLimbs.data(), l.exOwners.data(), (uint32_t)l.exLimbs.size());
    break;
  case Launch::L_BCONV_COL:
    for (auto &q : l.probs)
      descs.push_back(hm_bconv_desc{l.slicesIn, q.in.data(), q.inMods.data(), (uint32_t)q.in.size(), l.slicesOut, q.out.data(), q.outMods.data(), (uint32_t)q.out.size(), 0});
    st = hm_bconv_col(ctx, descs.data(), (uint32_t)descs.size(), l.galois, l.logLen);
    break;
  case Launch::L_REPLICATE:
    st = hm_replicate_limbs(ctx, pool, l.exLimbs.data(), l.exOwners.data(), (uint32_t)l.exLimbs.size());
    break;
  case Launch::L_AUTO:
    st = hm_automorph(ctx, pool, l.a.data(), pool, l.out.data(), cnt, l.galois);
    break;
  case Launch::L_EWE:
    st = hm_ewe(ctx, l.opcode, pool, l.a.data(), pool, l.b.empty() ? nullptr : l.b.data(), pool, l.c.empty() ? nullptr : l.c.data(), pool,
                l.d.empty() ? nullptr : l.d.data(), pool, l.out.data(), l.mods.data(), cnt, l.hasK ? l.k.data() : nullptr);
    break;
  case Launch::L_BCONV:
    for (auto &q : l.probs)
      descs.push_back(hm_bconv_desc{l.slicesIn ? l.slicesIn : pool, q.in.data(), q.inMods.data(), (uint32_t)q.in.size(),
                                    l.slicesOut ? l.slicesOut : pool, q.out.data(), q.outMods.data(), (uint32_t)q.out.size(), l.logLen,
                                    q.epi ? pool : nullptr, q.epi ? q.epA.data() : nullptr, q.epAdd ? pool : nullptr, q.epAdd ? q.epB.data() : nullptr,
                                    q.epi ? q.epK.data() : nullptr, q.inPacked ? 1u : 0u});
    st = hm_bconv_batch(ctx, descs.data(), (uint32_t)descs.size());
    break;
  }
  if (st == HM_OK && l.recordSlot >= 0) st = hm_exchange_mark(ctx, (uint32_t)l.recordSlot);
  if (st != HM_OK) throw std::runtime_error("stage " + l.name + ": " + hm_last_error(ctx));
}

void Arch::update() {
  if (backendKind == BACKEND_SIM) {
    if (!sim) throw std::runtime_error("sim backend: no program loaded");
    sim->step();
    return;
  }
  if (!prepared) prepare();
  if (nextLaunch >= launches.size()) return;
  Launch &l = *launches[nextLaunch++];
  if (backendKind == BACKEND_HIP) {
    hm_timer_start(ctx);
    enqueue(l);
    if (l.recordSlot >= 0) hm_exchange_wait(ctx, (uint32_t)l.recordSlot);
    uint64_t ns = 0;
    hm_timer_stop(ctx, &ns);
    elapsedNs += ns;
    stat->increaseStat(l.statKey + "_(0)", ns);  // per-unit busy time, ns (upstream: busy cycles per cluster)
  }
  stat->increaseStat(l.statKey + "_launches");
  completedIns += l.refInstructions;
}

bool Arch::simulateComplete() { return sim ? sim->complete() : prepared && nextLaunch >= launches.size(); }
unsigned long long Arch::getCycle() { return sim ? sim->cycle() : elapsedNs; }
unsigned long long Arch::getcompletedIns() { return sim ? sim->completedIns() : completedIns; }

void Arch::state() {
  if (sim) { std::cout << sim->describe(); return; }
  std::cout << "launched " << nextLaunch << " of " << launches.size() << " stages\n";
  if (nextLaunch < launches.size()) std::cout << "next: " << launches[nextLaunch]->name << "\n";
}

void Arch::shownStat() {
  if (sim) {  // the reference's block: same keys, same order (std::map), same values
    std::cout << "Start outPut statistic informations:\n";
    std::cout << "=====================================\n";
    for (const auto &kv : sim->stats()) std::cout << kv.first << " :\t" << kv.second << "\n";
    return;
  }
  stat->setStat("Total_ns", elapsedNs);
  // measured HBM bytes of the op from the profiler ("rocprof HBM counters in the same key : value block"; upstream's
  // HBM_(c) / MEM_(c) counters, src/mem.cpp:68-69,105-107).  A process cannot read its own rocprofv3 counters: the figures come from
  // the counter file a profiling pass left (profiles/roofline_inputs.json, written by tools/make_roofline_inputs.py), named by
  // HOMULATOR_COUNTER_FILE, and are printed only if that file describes this launch shape (hmult at N = 2^16 with this batch).
  if (const char *path = getenv("HOMULATOR_COUNTER_FILE")) {
    std::ifstream f(path);
    std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    auto num = [&](const std::string &key, double &v) {
      size_t p = text.find("\"" + key + "\"");
      if (p == std::string::npos) return false;
      p = text.find(':', p);
      if (p == std::string::npos) return false;
      v = atof(text.c_str() + p + 1);
      return true;
    };
    double fetch = 0, write = 0, b = 0;
    if (num("whole_op_fetch_kib", fetch) && num("whole_op_write_kib", write) && num("whole_op_batch", b) && (uint32_t)b == batch_ && n == 65536 && maxLevel_ == 45 && curLevel_ == 35 &&
        std::any_of(launches.begin(), launches.end(), [](const Launch *l) { return l->kind == Launch::L_TENSOR; })) {
      stat->setStat("HBM_fetch_KiB_per_op", (unsigned long long)(2 * fetch));   // FETCH_SIZE counts 64 B per 128-B request on gfx950
      stat->setStat("HBM_write_KiB_per_op", (unsigned long long)write);
      stat->setStat("HBM_bytes_per_op", (unsigned long long)((2 * fetch + write) * 1024));
    }
  }
  stat->showStat();
}

void Arch::run() {
  if (backendKind != BACKEND_HIP) return;
  for (Binding &b : bindings) {  // stream-ordered after the producer; one gather-copy launch per bound ciphertext part
    if (hm_wait_for(ctx, b.src->ctx) != HM_OK) throw std::runtime_error(std::string("hm_wait_for: ") + hm_last_error(ctx));
    if (hm_ewe(ctx, EWE_COPY, b.src->pool, b.srcLimbs.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pool, b.dstLimbs.data(),
               b.mods.data(), (uint32_t)b.dstLimbs.size(), nullptr) != HM_OK)
      throw std::runtime_error(std::string("bound input copy: ") + hm_last_error(ctx));
    // back edge: the producer's NEXT pass overwrites these limbs (its transforms even use their output as first-pass
    // scratch), so its stream waits until this copy has read them
    if (hm_wait_for(b.src->ctx, ctx) != HM_OK) throw std::runtime_error(std::string("hm_wait_for (back edge): ") + hm_last_error(ctx));
  }
  // single GPU: the plan is a fixed sequence of kernels -> captured into a HIP graph on the second run (the first
  // one warms the base-conversion table cache, which allocates) and replayed with one launch afterwards.
  // Sharded runs enqueue directly: the RCCL groups stay outside graphs.
  if (useGraph && world_ == 1) {
    if (graph) {
      if (hm_graph_launch(ctx, static_cast<hm_graph *>(graph)) != HM_OK) throw std::runtime_error(std::string("hm_graph_launch: ") + hm_last_error(ctx));
      return;
    }
    if (runCount++ >= 1) {
      hm_graph *g = nullptr;
      if (hm_capture_begin(ctx) != HM_OK) throw std::runtime_error(std::string("hm_capture_begin: ") + hm_last_error(ctx));
      for (Launch *l : launches) enqueue(*l);
      if (hm_capture_end(ctx, &g) != HM_OK) throw std::runtime_error(std::string("hm_capture_end: ") + hm_last_error(ctx));
      graph = g;
      if (hm_graph_launch(ctx, g) != HM_OK) throw std::runtime_error(std::string("hm_graph_launch: ") + hm_last_error(ctx));
      return;
    }
  }
  for (Launch *l : launches) enqueue(*l);
}
void Arch::sync() {
  if (ctx) hm_sync(ctx);
}
double Arch::timedRun(uint32_t iters) {
  if (!prepared) prepare();
  if (backendKind != BACKEND_HIP) return 0.0;
  hm_timer_start(ctx);
  for (uint32_t i = 0; i < iters; ++i) run();
  uint64_t ns = 0;
  hm_timer_stop(ctx, &ns);
  return (double)ns / iters;
}

void Arch::refill(const std::vector<AddrType> &addrs, uint64_t seed) {
  if (!prepared) prepare();
  if (backendKind != BACKEND_HIP || addrs.empty()) return;
  const InputFill *f = nullptr;
  for (const InputFill &x : fills)
    if (x.addrs == addrs) f = &x;
  if (!f) throw std::runtime_error("refill: not an input of this op");
  for (uint32_t c = 0; c < (f->shared ? 1u : batch_); ++c) {
    std::vector<uint32_t> limbs;
    for (AddrType a : addrs) limbs.push_back(limbOf(a) + c * (uint32_t)limbIndex.size());
    if (hm_fill_uniform(ctx, pool, limbs.data(), f->mods.data(), (uint32_t)limbs.size(), seed + c * kBatchSeedStride) != HM_OK)
      throw std::runtime_error(std::string("hm_fill_uniform: ") + hm_last_error(ctx));
  }
}
void Arch::snapshot(const std::vector<AddrType> &addrs, uint32_t slot) {
  if (!prepared) prepare();
  if (backendKind != BACKEND_HIP) return;
  auto &sn = snapshots[slot];
  if (sn.second != addrs.size()) {
    if (sn.first) hm_free(ctx, sn.first);
    void *p = nullptr;
    if (hm_malloc(ctx, addrs.size() * (size_t)n * 8, &p) != HM_OK) throw std::runtime_error(std::string("hm_malloc (snapshot): ") + hm_last_error(ctx));
    sn = {static_cast<uint64_t *>(p), addrs.size()};
  }
  for (size_t i = 0; i < addrs.size(); ++i)
    if (hm_memcpy_d2d(ctx, sn.first + i * n, pool + (size_t)limbOf(addrs[i]) * n, (size_t)n * 8) != HM_OK)
      throw std::runtime_error(std::string("hm_memcpy_d2d: ") + hm_last_error(ctx));
}
bool Arch::readSnapshot(uint32_t slot, uint64_t *host) {
  auto it = snapshots.find(slot);
  if (it == snapshots.end() || backendKind != BACKEND_HIP) return false;
  return hm_memcpy_d2h(ctx, host, it->second.first, it->second.second * (size_t)n * 8) == HM_OK;
}

// real data in: overwrites limb-polys (inputs, evaluation-key limbs) after prepare(); stream-ordered after everything enqueued so far
bool Arch::writeLimbs(const std::vector<AddrType> &addrs, const uint64_t *host, uint32_t copy) {
  if (backendKind != BACKEND_HIP || !pool || copy >= batch_) return false;
  sync();
  for (size_t i = 0; i < addrs.size(); ++i)
    if (hm_memcpy_h2d(ctx, pool + ((size_t)limbOf(addrs[i]) + (size_t)copy * limbIndex.size()) * n, host + i * n, (size_t)n * 8) != HM_OK) return false;
  return true;
}
bool Arch::readLimbs(const std::vector<AddrType> &addrs, uint64_t *host, uint32_t copy) {
  if (backendKind != BACKEND_HIP || !pool || copy >= batch_) return false;
  for (size_t i = 0; i < addrs.size(); ++i)
    if (hm_memcpy_d2h(ctx, host + i * n, pool + ((size_t)limbOf(addrs[i]) + (size_t)copy * limbIndex.size()) * n, (size_t)n * 8) != HM_OK) return false;
  return true;
}
