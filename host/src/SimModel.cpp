// SimModel.cpp — cycle model of the reference accelerator (see include/SimModel.h for what it is and is not).
//
// One step() is one pass of the reference's main loop (src/Operation.cpp:1046-1051): the driver offers each cluster's
// next DRAM line to its HBM port, then every cluster in turn advances its units, its four front ends and its memory
// controller (src/Arch.cpp:912-929).  Clusters see each other's scratchpads (NoC) and share the use counts, so the order
// "cluster 0 completely, then cluster 1, ..." is part of the model.
//
// Reference behaviour that shapes the numbers and is kept on purpose (file:line of the reference):
//  * front ends take one cycle per stage because a cycle runs write-back, commit, issue, decode, fetch in that order
//    (src/Arch.cpp:400-409, 518-524, 589-595, 667-681); the two EWE slots share one queue (:427-446);
//  * a unit only advances while something is committed to it (src/Arch.cpp:215-232), the base-conversion array always;
//  * a pipeline is a rigid shift register: it moves only when its last stage is free (include/Components.h:70-113);
//  * the scoreboard looks at operands 1 and 2 only (include/recodeboard.h:33-46).  Single-operand instructions (NTT, AUTO)
//    and the table operand of a conversion never match a pending output, so only EWE instructions ever wait on it;
//  * every base-conversion group is run by all bconv_num_high x bconv_num_width MAC ports (include/Driver.h:307-320), each
//    port consuming the operands' use counts again; a port marks its output pending on the scoreboard only when no further
//    group was decoded meanwhile (src/Arch.cpp:774-777); "BCONV_OutMem_Stall" counts successful write-backs (:790-793);
//  * an operand counts as ready when its line is resident, or when nobody will read it any more (src/mem.cpp:39-44);
//    reading decrements the line in this cluster, in every other cluster that holds it, and the global count (:55-67);
//  * a write-back first clears the "to be produced" mark, then may still fail on a full unit FIFO (include/mem.h:546-571);
//  * the first add to a counter stores 1 whatever the amount, counters are 32-bit (include/Staistics.h:21-28).
#include "SimModel.h"

namespace {
struct Port {
  int32_t ins = -1;
  bool full() const { return ins >= 0; }
};

// include/Components.h:13-148 as a ring buffer: slot(i) is stage i, a shift is one step of the head
struct Pipe {
  std::vector<int32_t> ring;
  uint32_t delay = 0, head = 0, used = 0;
  bool overlap = true, executed = false;
  Port *in = nullptr;
  Port out;
  void init(uint32_t d, bool ov = true) {
    if (d == 0) throw std::runtime_error("sim backend: a pipeline delay of 0 cycles is not modelled");
    delay = d; overlap = ov; ring.assign(d, -1);
  }
  int32_t &slot(uint32_t i) { uint32_t k = head + i; return ring[k >= delay ? k - delay : k]; }
  bool running() const { return in->full() || used > 0; }
  void update() {
    executed = false;
    int32_t &last = slot(delay - 1);
    if (last >= 0 && !out.full()) { out.ins = last; last = -1; used--; }
    if (used > 0 && slot(delay - 1) < 0) { head = head ? head - 1 : delay - 1; executed = true; }
    if ((overlap && used == delay) || (!overlap && used > 0)) return;
    int32_t &first = slot(0);
    if (first < 0 && in->full()) { first = in->ins; in->ins = -1; used++; }
  }
};

// include/mem.h:130-212: the DRAM-side port, same register structure, payload = one line address
struct HbmPort {
  std::vector<long long> ring;
  uint32_t delay = 0, head = 0, used = 0;
  bool executed = false, inSignal = false, outSignal = false;
  long long inVal = -1, outVal = -1;
  void init(uint32_t d) {
    if (d == 0) throw std::runtime_error("sim backend: offDelay = 0 is not modelled");
    delay = d; ring.assign(d, -1);
  }
  long long &slot(uint32_t i) { uint32_t k = head + i; return ring[k >= delay ? k - delay : k]; }
  bool offer(long long a) {
    if (inSignal) return false;
    inVal = a; inSignal = true;
    return true;
  }
  void update() {
    executed = false;
    long long &last = slot(delay - 1);
    if (last >= 0 && !outSignal) { outVal = last; outSignal = true; last = -1; used--; }
    if (used > 0 && slot(delay - 1) < 0) { head = head ? head - 1 : delay - 1; executed = true; }
    long long &first = slot(0);
    if (first < 0 && inSignal) { first = inVal; used++; inSignal = false; }
  }
};

struct Counter {
  bool present = false;
  uint32_t v = 0;
  void inc() { if (present) v++; else { present = true; v = 1; } }
  void add(uint32_t n) { if (present) v += n; else { present = true; v = 1; } }
};

struct Slot {  // one fetch / decode / issue front-end lane
  bool fetch = false, decode = false, issue = false;
  int32_t decodeHold = -1, issueHold = -1;
  uint32_t remaining = 0;  // base conversion: instructions of the decoded group still to issue (issued last to first)
  size_t cursor = 0;       // base conversion: this port's position in the cluster's group list
  std::vector<int32_t> inflight;  // base conversion: committed, not yet written back (FIFO)
  size_t inflightHead = 0;
};

struct Cluster {
  // units
  Port eweIn[4], nttIn, autoIn;
  Pipe mul[4], add[2];
  Port fuse[2];
  std::vector<Pipe> nttPipes, autoPipes;  // in data-flow order
  int nttStallIndex = -1;
  std::vector<Port> macIn;
  std::vector<Pipe> mac;
  unsigned long long exeEWE = 0, exeNTT = 0, exeAUTO = 0, exeBCONV = 0;
  // front ends
  Slot ewe[2], ntt, aut;
  std::vector<Slot> conv;
  size_t eweCursor = 0, nttCursor = 0, autoCursor = 0;
  uint32_t flightEWE = 0, flightNTT = 0, flightAUTO = 0;
  // memory side
  HbmPort hbm;
  std::vector<int32_t> line;  // scratchpad: remaining reads of the resident line at this address, -1 = not resident
  uint32_t freeLines = 0;
  std::vector<uint8_t> board;  // scoreboard: pending output addresses
  std::vector<AddrType> fromUnits, fromNoC;
  struct DramEntry { bool has; AddrType a; };
  std::vector<DramEntry> fromDram;
  size_t dramCursor = 0;
  Counter stEweMem, stAutoMem, stNttMem, stConvMem, stEweOut, stAutoOut, stNttOut, stConvOut, stMem, stHbm;
};
}  // namespace

struct SimModel::Impl {
  SimProgram prog;
  std::vector<Cluster> cl;
  uint32_t C = 0, batchSize = 0, bh = 0, bw = 0, depthUnits = 0, depthDram = 0, depthNoC = 0;
  bool hasHPIP = false;
  Counter stNoC;
  unsigned long long *completed = nullptr;

  // ---- scratchpad (include/mem.h:259-463; physical line ids and the valid bit do not influence anything observable)
  bool memInsert(Cluster &k, AddrType a) {
    if (k.line[a] >= 0) return true;
    if (k.freeLines == 0) return false;
    k.line[a] = (int32_t)prog.uses[a];
    k.freeLines--;
    return true;
  }
  void memRead(Cluster &k, AddrType a) {  // validAddr :401-448
    if (k.line[a] > 0 && --k.line[a] == 0) { k.line[a] = -1; k.freeLines++; }
  }
  void memReadRemote(Cluster &k, AddrType a) {  // processAddr :355-372
    if (k.line[a] > 0) k.line[a]--;
    else if (k.line[a] == 0) { k.line[a] = -1; k.freeLines++; }
  }
  static bool queued(const std::vector<AddrType> &q, AddrType a) { return std::find(q.begin(), q.end(), a) != q.end(); }

  void nocRequest(Cluster &k, AddrType a) {  // src/mem.cpp:78-100
    if (queued(k.fromNoC, a) || prog.uses[a] == 0 || k.fromNoC.size() >= depthNoC) return;
    stNoC.inc();
    k.fromNoC.push_back(a);
  }
  bool unitWriteback(Cluster &k, AddrType a) {  // include/mem.h:546-571
    prog.pendingOut[a] = 0;
    if (queued(k.fromUnits, a) || prog.uses[a] == 0) return true;
    if (k.fromUnits.size() >= depthUnits) return false;
    k.fromUnits.push_back(a);
    return true;
  }

  // src/mem.cpp:22-75
  bool operandsReady(uint32_t c, uint32_t first, uint32_t count, Counter &stall) {
    Cluster &k = cl[c];
    for (uint32_t n = 0; n < count; n++) {
      const SimIns &i = prog.ins[first + n];
      for (int o = 0; o < i.nIn; o++) {
        const AddrType a = i.op[o];
        if (a == 0) continue;
        if (prog.pendingOut[a]) return false;
        if (k.line[a] >= 0 || prog.uses[a] == 0) continue;
        for (uint32_t id = 0; id < C; id++)
          if (id != c && cl[id].line[a] >= 0) nocRequest(k, a);
        stall.inc();
        return false;
      }
    }
    for (uint32_t n = 0; n < count; n++) {
      const SimIns &i = prog.ins[first + n];
      for (int o = 0; o < i.nIn; o++) {
        const AddrType a = i.op[o];
        memRead(k, a);
        for (uint32_t id = 0; id < C; id++)
          if (id != c && cl[id].line[a] >= 0) memReadRemote(cl[id], a);
        if (prog.uses[a] > 0) prog.uses[a]--;
      }
      k.stMem.add(i.nIn * batchSize);
    }
    return true;
  }
  bool boardClear(const Cluster &k, const SimIns &i) const { return !(k.board[i.op[1]] || k.board[i.op[2]]); }

  // ---- units (src/Components.cpp)
  void unitsAdvance(Cluster &k) {
    if (k.flightEWE) {  // EWE::update :122-150
      bool run = false;
      for (Pipe &a : k.add)
        if (a.running()) { a.update(); run |= a.executed; }
      for (int f = 0; f < 2; f++) {
        Port &x = k.mul[2 * f].out, &y = k.mul[2 * f + 1].out;
        if (x.full() && y.full() && !k.fuse[f].full()) { k.fuse[f].ins = y.ins; x.ins = y.ins = -1; }
      }
      for (Pipe &m : k.mul)
        if (m.running()) { m.update(); run |= m.executed; }
      k.exeEWE += run;
    }
    if (k.flightAUTO) {  // AUTOU::update :238-253
      bool run = false;
      for (size_t s = k.autoPipes.size(); s-- > 0;)
        if (k.autoPipes[s].running()) { k.autoPipes[s].update(); run |= k.autoPipes[s].executed; }
      k.exeAUTO += run;
    }
    {  // BCONVU::update :307-324
      bool run = false;
      for (Pipe &m : k.mac)
        if (m.running()) { m.update(); run |= m.executed; }
      k.exeBCONV += run;
    }
    if (k.flightNTT) {  // NTTU::update :504-569
      bool run = false;
      for (size_t s = k.nttPipes.size(); s-- > 0;) {
        Pipe &p = k.nttPipes[s];
        if (!p.running()) continue;
        p.update();
        if (p.executed) run = (int)s != k.nttStallIndex;
      }
      k.exeNTT += run;
    }
  }

  // ---- single-instruction front ends (EWE slots, NTT, AUTO): src/Arch.cpp:400-665
  template <class Ready> void laneCommit(Cluster &k, Slot &s, uint32_t &flight, Ready &&portsFree, bool markBoard = true) {
    if (!s.issue || !portsFree(s.issueHold)) return;
    if (markBoard) k.board[prog.ins[s.issueHold].out] = 1;
    flight++;
    s.issue = false;
    s.issueHold = -1;
  }
  void laneIssue(const Cluster &k, Slot &s, bool useBoard) {
    if (!s.decode || s.issue) return;
    if (useBoard && !boardClear(k, prog.ins[s.decodeHold])) return;
    s.issueHold = s.decodeHold; s.decodeHold = -1;
    s.issue = true; s.decode = false;
  }
  void laneDecode(uint32_t c, Slot &s, const std::vector<uint32_t> &queue, size_t &cursor, Counter &stall) {
    if (!s.fetch || s.decode || cursor >= queue.size()) return;
    const uint32_t id = queue[cursor];
    if (!operandsReady(c, id, 1, stall)) return;
    cursor++;
    s.decodeHold = (int32_t)id;
    s.fetch = false; s.decode = true;
  }
  bool retire(Cluster &k, Port &out, Counter &stall, uint32_t &flight) {  // handleSignalOutput :968-998
    if (!out.full()) return false;
    const SimIns &i = prog.ins[out.ins];
    if (!unitWriteback(k, i.out)) { stall.inc(); return false; }
    k.board[i.out] = 0;
    out.ins = -1;
    flight--;
    (*completed)++;
    return true;
  }

  void frontEnds(uint32_t c) {
    Cluster &k = cl[c];
    // EWE :400-516
    for (Pipe &a : k.add) retire(k, a.out, k.stEweOut, k.flightEWE);
    for (int n = 0; n < 2; n++) {
      Slot &s = k.ewe[n];
      laneCommit(k, s, k.flightEWE, [&](int32_t id) {
        if (k.eweIn[2 * n].full() || k.eweIn[2 * n + 1].full()) return false;
        k.eweIn[2 * n].ins = k.eweIn[2 * n + 1].ins = id;
        return true;
      });
      laneIssue(k, s, true);
      laneDecode(c, s, prog.ewe[c], k.eweCursor, k.stEweMem);
      s.fetch = true;
    }
    // AUTO :518-586
    retire(k, k.autoPipes.back().out, k.stAutoOut, k.flightAUTO);
    laneCommit(k, k.aut, k.flightAUTO, [&](int32_t id) { if (k.autoIn.full()) return false; k.autoIn.ins = id; return true; });
    laneIssue(k, k.aut, false);
    laneDecode(c, k.aut, prog.aut[c], k.autoCursor, k.stAutoMem);
    k.aut.fetch = true;
    // NTT :588-665
    retire(k, k.nttPipes.back().out, k.stNttOut, k.flightNTT);
    laneCommit(k, k.ntt, k.flightNTT, [&](int32_t id) { if (k.nttIn.full()) return false; k.nttIn.ins = id; return true; });
    laneIssue(k, k.ntt, false);
    laneDecode(c, k.ntt, prog.ntt[c], k.nttCursor, k.stNttMem);
    k.ntt.fetch = true;
    // BCONV :667-797 — every port walks the whole group list of the cluster
    const std::vector<SimProgram::Group> &groups = prog.bconv[c];
    for (size_t pt = 0; pt < k.conv.size(); pt++) {  // write-back (handleMultiOutput :931-966): in commit order per port
      Slot &s = k.conv[pt];
      Port &out = k.mac[pt].out;
      if (!out.full()) continue;
      if (s.inflightHead >= s.inflight.size() || s.inflight[s.inflightHead] != out.ins)
        throw std::runtime_error("Computation Out of Order for HPIP, causing computational error!\n");
      const SimIns &i = prog.ins[out.ins];
      if (!unitWriteback(k, i.out)) continue;
      out.ins = -1;
      k.board[i.out] = 0;
      s.inflightHead++;
      if (s.inflightHead == s.inflight.size()) { s.inflight.clear(); s.inflightHead = 0; }
      (*completed)++;
      k.stConvOut.inc();
    }
    for (size_t pt = 0; pt < k.conv.size(); pt++) {
      Slot &s = k.conv[pt];
      if (s.issue && !k.macIn[pt].full()) {  // commit :754-782
        k.macIn[pt].ins = s.issueHold;
        s.inflight.push_back(s.issueHold);
        if (!s.decode) k.board[prog.ins[s.issueHold].out] = 1;
        s.issue = false;
        s.issueHold = -1;
      }
      if (s.decode && !s.issue) {  // issue :723-752: last instruction of the group first; the table operand never waits
        s.issueHold = s.decodeHold + (int32_t)s.remaining - 1;
        s.issue = true;
        if (--s.remaining == 0) { s.decode = false; s.decodeHold = -1; }
      }
      if (s.fetch && !s.decode && s.cursor < groups.size()) {  // decode :697-721
        const SimProgram::Group g = groups[s.cursor];
        if (operandsReady(c, g.first, g.count, k.stConvMem)) {
          s.cursor++;
          s.decodeHold = (int32_t)g.first;
          s.remaining = g.count;
          s.fetch = false; s.decode = true;
        }
      }
      s.fetch = true;
    }
  }

  // ---- memory controller (src/mem.cpp:102-147)
  void memoryAdvance(Cluster &k) {
    k.hbm.update();
    if (k.hbm.executed) k.stHbm.inc();
    if (!k.fromNoC.empty() && memInsert(k, k.fromNoC.front())) k.fromNoC.erase(k.fromNoC.begin());
    if (!k.fromUnits.empty() && memInsert(k, k.fromUnits.front())) k.fromUnits.erase(k.fromUnits.begin());
    if (!k.fromDram.empty() && k.fromDram.front().has && memInsert(k, k.fromDram.front().a)) k.fromDram.erase(k.fromDram.begin());
    if (k.hbm.outSignal && k.fromDram.size() < depthDram) {  // include/mem.h:576-593: an entry nobody reads stays as an EMPTY slot
      const AddrType a = (AddrType)k.hbm.outVal;
      k.fromDram.push_back({prog.uses[a] != 0, a});
      k.hbm.outSignal = false;
      k.hbm.outVal = -1;
    }
  }
};

SimModel::SimModel(Config *cfg, SimProgram &&program) : m(new Impl) {
  m->prog = std::move(program);
  m->completed = &completed_;
  m->C = cfg->getValue("cluster");
  if (m->C != m->prog.cluster) throw std::runtime_error("sim backend: program built for another cluster count");
  m->batchSize = cfg->getValue("batchSize");
  m->bh = cfg->getValue("bconv_num_high");
  m->bw = cfg->getValue("bconv_num_width");
  m->depthUnits = cfg->getValue("memUnitsFifo");
  m->depthDram = cfg->getValue("memDramFifo");
  m->depthNoC = m->C;  // include/mem.h:521
  m->hasHPIP = cfg->getValue("hasHPIPU") == 1;
  if (cfg->getValue("ewe_num_mul") != 4 || cfg->getValue("ewe_num_add") != 2)
    throw std::runtime_error("sim backend: ewe_num_mul = 4 and ewe_num_add = 2 are the only shape upstream's Arch wires (src/Arch.cpp:170-174, 469)");
  const uint32_t memSize = cfg->getValue("memSize"), bits = cfg->getValue("elementBitWidth");
  const uint32_t lines = uint32_t(memSize / (float(m->batchSize * bits) / (8 * 1024 * 1024)));  // include/mem.h:512-513
  const uint32_t bfly = cfg->getValue("butterfly_delay"), intra = cfg->getValue("intraTrans_delay"), inter = cfg->getValue("interTrans_delay");
  const uint32_t nttStall = cfg->getValue("ntt_stall_delay");
  const size_t words = (size_t)m->prog.maxAddr + 2;
  m->cl.resize(m->C);
  for (Cluster &k : m->cl) {
    for (int i = 0; i < 4; i++) { k.mul[i].init(cfg->getValue("ewe_mult_delay")); k.mul[i].in = &k.eweIn[i]; }
    for (int i = 0; i < 2; i++) { k.add[i].init(cfg->getValue("ewe_madd_delay")); k.add[i].in = &k.fuse[i]; }
    // src/Components.cpp:397-431: step1, intra, step2, inter, [stall], step1, intra, step2
    std::vector<std::pair<uint32_t, bool>> chain = {{bfly * cfg->getValue("phase1_step1_depth"), true}, {intra, true},
                                                    {bfly * cfg->getValue("phase1_step2_depth"), true}, {inter, true}};
    if (nttStall > 0) { k.nttStallIndex = (int)chain.size(); chain.push_back({nttStall, false}); }
    chain.push_back({bfly * cfg->getValue("phase2_step1_depth"), true});
    chain.push_back({intra, true});
    chain.push_back({bfly * cfg->getValue("phase2_step2_depth"), true});
    k.nttPipes.resize(chain.size());
    for (size_t s = 0; s < chain.size(); s++) {
      k.nttPipes[s].init(chain[s].first, chain[s].second);
      k.nttPipes[s].in = s == 0 ? &k.nttIn : &k.nttPipes[s - 1].out;
    }
    k.autoPipes.resize(cfg->getValue("auto_stages"));
    if (k.autoPipes.empty()) throw std::runtime_error("sim backend: auto_stages = 0");
    for (size_t s = 0; s < k.autoPipes.size(); s++) {
      k.autoPipes[s].init(cfg->getValue("auto_delay"));
      k.autoPipes[s].in = s == 0 ? &k.autoIn : &k.autoPipes[s - 1].out;
    }
    const size_t ports = (size_t)m->bh * m->bw;
    k.macIn.resize(ports); k.mac.resize(ports); k.conv.resize(ports);
    for (uint32_t h = 0; h < m->bh; h++)
      for (uint32_t w = 0; w < m->bw; w++) {
        Pipe &p = k.mac[h * m->bw + w];
        p.init(cfg->getValue("bconv_mac_delay") + w * cfg->getValue("bconv_fifo_delay"));  // :278-292
        p.in = &k.macIn[h * m->bw + w];
      }
    k.hbm.init(cfg->getValue("offDelay"));
    k.line.assign(words, -1);
    k.board.assign(words, 0);
    k.freeLines = lines;
  }
  total_ = m->prog.totalInstructions(m->bh * m->bw);
}
SimModel::~SimModel() { delete m; }

void SimModel::step() {
  for (uint32_t c = 0; c < m->C; c++) {  // Driver::IssueDataFromDramToChip, include/Driver.h:327-340
    Cluster &k = m->cl[c];
    const std::vector<AddrType> &list = m->prog.dram[c];
    if (k.dramCursor < list.size() && k.hbm.offer((long long)list[k.dramCursor])) k.dramCursor++;
  }
  for (uint32_t c = 0; c < m->C; c++) {
    m->unitsAdvance(m->cl[c]);
    m->frontEnds(c);
    m->memoryAdvance(m->cl[c]);
  }
  cycle_++;
}

bool SimModel::complete() const {  // include/Arch.h:246-269 with src/Arch.cpp:311-375
  for (uint32_t c = 0; c < m->C; c++) {
    const Cluster &k = m->cl[c];
    if (k.eweCursor < m->prog.ewe[c].size() || k.flightEWE) return false;
    for (const Slot &s : k.ewe)
      if (s.decodeHold >= 0 || s.issueHold >= 0) return false;
    if (k.nttCursor < m->prog.ntt[c].size() || k.flightNTT || k.ntt.decodeHold >= 0 || k.ntt.issueHold >= 0) return false;
    for (const Slot &s : k.conv)
      if (s.cursor < m->prog.bconv[c].size() || s.issueHold >= 0 || s.decodeHold >= 0 || s.inflightHead < s.inflight.size()) return false;
  }
  return true;
}

std::map<std::string, uint32_t> SimModel::stats() const {
  std::map<std::string, uint32_t> out;
  auto put = [&](const std::string &key, const Counter &c) { if (c.present) out[key] = c.v; };
  for (uint32_t c = 0; c < m->C; c++) {
    const Cluster &k = m->cl[c];
    const std::string t = "_(" + std::to_string(c) + ")";
    put("EWE_MEM_Stall" + t, k.stEweMem); put("AUTO_MEM_Stall" + t, k.stAutoMem); put("NTT_MEM_Stall" + t, k.stNttMem);
    put("BCONV_MEM_Stall" + t, k.stConvMem); put("EWE_OutMem_Stall" + t, k.stEweOut); put("AUTO_OutMem_Stall" + t, k.stAutoOut);
    put("NTT_OutMem_Stall" + t, k.stNttOut); put("BCONV_OutMem_Stall" + t, k.stConvOut); put("MEM" + t, k.stMem); put("HBM" + t, k.stHbm);
    out["EWE" + t] = (uint32_t)k.exeEWE; out["NTT" + t] = (uint32_t)k.exeNTT;   // src/Arch.h:280-295 (set at the end, always present)
    out["AUTO" + t] = (uint32_t)k.exeAUTO; out["BCONV" + t] = (uint32_t)k.exeBCONV;
    if (m->hasHPIP) out["HPIP" + t] = 0;
  }
  put("NoC_Mem_Chip", m->stNoC);
  return out;
}

std::string SimModel::describe() const {
  std::string out;
  for (uint32_t c = 0; c < m->C; c++) {
    const Cluster &k = m->cl[c];
    size_t convLeft = 0, convFlight = 0;
    for (const Slot &s : k.conv) { convLeft = std::max(convLeft, m->prog.bconv[c].size() - s.cursor); convFlight += s.inflight.size() - s.inflightHead; }
    out += "cluster " + std::to_string(c) + ": queued EWE " + std::to_string(m->prog.ewe[c].size() - k.eweCursor) + " NTT " +
           std::to_string(m->prog.ntt[c].size() - k.nttCursor) + " AUTO " + std::to_string(m->prog.aut[c].size() - k.autoCursor) +
           " BCONV groups " + std::to_string(convLeft) + "; in flight EWE " + std::to_string(k.flightEWE) + " NTT " + std::to_string(k.flightNTT) +
           " AUTO " + std::to_string(k.flightAUTO) + " BCONV " + std::to_string(convFlight) + "; free scratchpad lines " + std::to_string(k.freeLines) +
           "; FIFOs units " + std::to_string(k.fromUnits.size()) + "/" + std::to_string(m->depthUnits) + " NoC " + std::to_string(k.fromNoC.size()) + "/" +
           std::to_string(m->depthNoC) + " DRAM " + std::to_string(k.fromDram.size()) + "/" + std::to_string(m->depthDram) + "; DRAM lines to fetch " +
           std::to_string(m->prog.dram[c].size() - k.dramCursor) + "\n";
  }
  return out;
}
