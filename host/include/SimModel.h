// SimModel.h — `backend = sim`: the build's own cycle model of the reference accelerator.
//
// The reference IS a cycle model (src/Arch.cpp:912-929 update loop, :400-897 per-unit front ends, src/Components.cpp
// unit pipelines, src/mem.cpp + include/mem.h scratchpad / NoC / HBM port, include/recodeboard.h scoreboard,
// include/Driver.h placement).  This backend reproduces its cycle counts and its stat block from the timing constants of
// the same .cfg, so one binary gives both simulated-accelerator cycles (`backend = sim`) and measured MI355X time
// (`backend = hip`).  It is written for speed, not as a transcription: instructions are integer ids in flat arrays, every
// address-keyed container (use counts, pending outputs, scoreboard, scratchpad lines) is a dense array indexed by the
// line address, pipelines are ring buffers with a moving head, queues are cursors into the literal program.  The
// reference needs 57 s for `config_4_N15.cfg hmult 16 10 4` (14 816 cycles, 316 416 instructions); this model needs a
// fraction of a second, which also makes the N = 2^16 points (hours upstream) runnable.
//
// Behaviour that is observable in cycle counts or counters is kept literally, including the reference's quirks
// (DESIGN.md §10); behaviour that is not observable (instruction names, program counters, operand
// format bits, physical line ids) is dropped.
#ifndef HOMULATOR_SIMMODEL_H
#define HOMULATOR_SIMMODEL_H
#include "Basic.h"
#include "Config.h"

// one 256-coefficient instruction of the literal stream (reference: include/Instruction.h)
struct SimIns {
  AddrType op[4];
  AddrType out;
  uint8_t nIn;   // 1: NTT / INTT / AUTO, 2: BCONV (input, table), 4: EWE (0 = fake operand)
};

// the literal program: what include/Driver.h leaves in the per-cluster FIFOs after every dispatchInstructions()
struct SimProgram {
  uint32_t cluster = 0;
  std::vector<SimIns> ins;
  // single-instruction groups, in queue order (ids into `ins`)
  std::vector<std::vector<uint32_t>> ewe, ntt, aut;
  // base-conversion groups: [first, first + count) in `ins`, replicated upstream to every MAC port (Driver.h:307-320)
  struct Group { uint32_t first, count; };
  std::vector<std::vector<Group>> bconv;
  std::vector<std::vector<AddrType>> dram;       // per cluster: lines fetched from DRAM, first-touch order (Driver.h:107-153)
  std::vector<uint32_t> uses;                    // DataMap::inputDataAddr: remaining reads per line address
  std::vector<uint8_t> usesKey;                  // ... and whether the key exists at all
  std::vector<uint8_t> pendingOut;               // DataMap::outDataAddr: lines some instruction has yet to produce
  AddrType maxAddr = 0;
  unsigned long long totalInstructions(uint32_t bconvPorts) const;
};

class SimModel {
public:
  SimModel(Config *cfg, SimProgram &&prog);
  ~SimModel();
  void step();                 // one cycle: DRAM feed (Driver::IssueDataFromDramToChip) + Arch::update
  bool complete() const;       // Arch::simulateComplete (EWE, NTT and BCONV drained; AUTO is not looked at upstream)
  unsigned long long cycle() const { return cycle_; }
  unsigned long long completedIns() const { return completed_; }
  unsigned long long totalIns() const { return total_; }
  // the stat block in the reference's key set, order and (32-bit, first-increment-stores-1) arithmetic
  std::map<std::string, uint32_t> stats() const;
  // what is stuck where (printed at upstream's dead-lock exit instead of its Arch::state dump): per cluster the instructions still
  // queued per unit, in flight per unit, free scratchpad lines, FIFO fills and DRAM lines not yet fetched
  std::string describe() const;

private:
  struct Impl;
  Impl *m;
  unsigned long long cycle_ = 0, completed_ = 0, total_ = 0;
};
#endif
