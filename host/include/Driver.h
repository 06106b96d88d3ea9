// Driver.h — stage placement and dispatch, with the reference's entry points (include/Driver.h:71-105
// dispatchInstructions, :287-325 IssueInsFromDramToChip, :327-340 IssueDataFromDramToChip, :370
// getTotalIns).  Upstream routes every 256-coefficient instruction to a cluster FIFO; here a whole stage is
// one unit of dispatch (one GPU launch covers all limbs x all N), and the placement rule `limb % cluster`
// (include/Driver.h:158,178) is recorded on the stage: it is the multi-GPU limb-sharding rule.
#ifndef HOMULATOR_DRIVER_H
#define HOMULATOR_DRIVER_H
#include "Arch.h"
#include "Basic.h"
#include "Config.h"
#include "Instruction.h"

class Driver {
private:
  uint32_t cluster;
  uint32_t bconvh, bconvw;
  std::vector<Stage> pending;
  unsigned long long totalIns = 0;
  uint32_t unnamed = 0;

public:
  explicit Driver(Config *cfg) {
    cluster = cfg->getValue("cluster");
    bconvh = cfg->getValue("bconv_num_high");
    bconvw = cfg->getValue("bconv_num_width");
  }

  // ---- upstream's entry point, upstream's signature (include/Driver.h:71): map[level][batch] = instruction group.
  // Upstream routes each 256-coefficient group to a cluster FIFO (include/Driver.h:155-246); here a STAGE is the unit of
  // dispatch (one GPU launch covers all limbs x all N), so the per-batch groups of a level are coalesced: the instructions of
  // batch 0 are the stage's limb records and the records of the other batches add their upstream-instruction counts to them.
  // The build's own generators emit one record per limb (map[level].size() == 1).  Throws like upstream on an empty stage
  // (:72-74) or an op kind no unit executes (:100-104).
  void dispatchInstructions(const std::vector<std::vector<INSGROUP>> &map) { dispatchInstructions(std::string(), map); }
  void dispatchInstructions(const std::string &stageName, const std::vector<std::vector<INSGROUP>> &map) {
    if (map.empty() || map[0].empty() || map[0][0].empty()) throw std::runtime_error("Empty instruction map provided.");
    std::vector<INSGROUP> perLimb;
    for (const auto &level : map) {
      if (level.empty()) continue;
      INSGROUP g = level[0];
      for (size_t b = 1; b < level.size(); ++b)
        for (size_t k = 0; k < level[b].size() && k < g.size(); ++k) g[k]->refInstructions += level[b][k]->refInstructions;
      perLimb.push_back(std::move(g));
    }
    dispatchInstructions(stageName, perLimb);
  }
  // the compact form the build's Operation classes use: map[level] = the limb's record(s), plus the stage key
  void dispatchInstructions(const std::string &stageName, const std::vector<INSGROUP> &map) {
    if (map.empty() || map[0].empty()) throw std::runtime_error("Empty instruction map provided.");
    const std::string &opName = map[0][0]->GetOpName();
    if (opName != "NTT" && opName != "INTT" && opName != "AUTO" && opName != "MULT" && opName != "BCONV_STEP2" &&
        opName != "IP") {
      std::cout << opName << "\n";
      throw std::runtime_error("This instruction generation error, as not exist corresponding component.");
    }
    Stage st;
    st.name = stageName.empty() ? ("stage_" + std::to_string(unnamed++)) : stageName;
    st.kind = map[0][0]->ops;
    for (const auto &g : map)
      for (Instruction *ins : g) st.ins.push_back(ins);
    pending.push_back(std::move(st));
  }

  // hands every pending stage to the backend; the instruction total replicates upstream's accounting: every
  // BCONV group is issued to all bconv_num_high x bconv_num_width MAC ports (include/Driver.h:307-320)
  void IssueInsFromDramToChip(Arch *arch) {
    for (const Stage &st : pending) {
      const bool bconv = st.kind == BCONV_STEP2;
      const std::string unit = bconv ? "BCONV" : st.kind == AUTO ? "AUTO" : (st.kind == NTT || st.kind == INTT) ? "NTT" : st.kind == IP ? "HPIP" : "EWE";
      arch->issueIns(0, unit, st);
      for (Instruction *ins : st.ins) totalIns += ins->refInstructions * (bconv ? (unsigned long long)bconvh * bconvw : 1ull);
    }
    pending.clear();
  }
  // upstream's signature (include/Driver.h:327-340: one group of DRAM lines per cluster and call goes to the scratchpad).
  // Operands are resident in HBM from the start and the sim backend feeds its own model (SimModel::step), so nothing moves here.
  void IssueDataFromDramToChip(std::vector<MemController *> memC) { (void)memC; }
  unsigned long long getTotalIns() const { return totalIns; }
  uint32_t getCluster() const { return cluster; }
};
#endif
