#include "InsGen.h"

std::vector<std::string> ins_ops_name = {"NTT",  "INTT", "MULT", "MADD", "MSUB",     "BCONV_STEP1", "BCONV_STEP2",
                                         "AUTO", "DS",   "PRNG", "IP",   "FETCH_RF", "STORE_RF"};

InsGen::InsGen(Config *cfg) {
  batchSize = cfg->getValue("batchSize");
  batchCount = cfg->getValue("N") / batchSize;  // reference: src/InsGen.cpp:9-13
}

Instruction *InsGen::make(const std::string &name, ins_ops op, uint32_t level, uint32_t modId,
                          std::initializer_list<INSGROUP *> deps) {
  Instruction *ins = new Instruction(name, op, level);
  ins->mod_id = modId;
  ins->refInstructions = batchCount;
  for (INSGROUP *d : deps)
    if (d) ins->depsInsList.insert(ins->depsInsList.end(), d->begin(), d->end());
  return ins;
}

static void record(std::map<AddrType, std::vector<Instruction *>> *m, Instruction *ins) {
  if (m) (*m)[ins->OutputOperand].push_back(ins);  // producer lookup used by KeySwitch / Rescale
}

INSGROUP InsGen::GenNTT(uint32_t levelId, std::string name, INSGROUP *dep, bool ntt, AddrType in, AddrType out,
                        uint32_t modId, bool passthrough) {
  Instruction *ins = make(name, ntt ? NTT : INTT, levelId, modId, {dep});
  ins->operandList = {in};
  ins->OutputOperand = out;
  ins->passthrough = passthrough;
  record(DataInsMap, ins);
  return {ins};
}

INSGROUP InsGen::GenAUTO(uint32_t levelId, std::string name, INSGROUP *dep, AddrType in, AddrType out, uint32_t galois,
                         uint32_t modId) {
  Instruction *ins = make(name, AUTO, levelId, modId, {dep});
  ins->operandList = {in};
  ins->OutputOperand = out;
  ins->galois = galois;
  record(DataInsMap, ins);
  return {ins};
}

INSGROUP InsGen::GenEWE(uint32_t levelId, std::string name, INSGROUP *d1, INSGROUP *d2, INSGROUP *d3, INSGROUP *d4,
                        AddrType op1, AddrType op2, AddrType op3, AddrType op4, AddrType out, ewe_opcode opcode,
                        uint32_t modId, bool hasConstant, uint64_t constant) {
  Instruction *ins = make(name, MULT, levelId, modId, {d1, d2, d3, d4});  // upstream tags every EWE op MULT
  ins->operandList = {op1, op2, op3, op4};
  ins->OutputOperand = out;
  ins->opcode = opcode;
  ins->hasConstant = hasConstant;
  ins->constant = constant;
  record(DataInsMap, ins);
  return {ins};
}

INSGROUP InsGen::GenBCONV(uint32_t levelId, uint32_t InLevel, std::string name, std::vector<INSGROUP> deps,
                          std::vector<AddrType> inAddrs, std::vector<uint32_t> inMods, AddrType tableAddr,
                          AddrType out, uint32_t outMod) {
  if (inAddrs.size() != InLevel || inMods.size() != InLevel) throw std::runtime_error("GenBCONV: input list size mismatch");
  Instruction *ins = new Instruction(name, BCONV_STEP2, levelId);
  ins->mod_id = outMod;
  ins->operandList = inAddrs;
  ins->operandList.push_back(tableAddr);
  ins->inMods = inMods;
  ins->OutputOperand = out;
  ins->refInstructions = (unsigned long long)batchCount * InLevel;  // InLevel instructions per batch (src/InsGen.cpp:279-311)
  for (auto &g : deps) ins->depsInsList.insert(ins->depsInsList.end(), g.begin(), g.end());
  record(DataInsMap, ins);
  return {ins};
}

INSGROUP InsGen::GenHPIP(uint32_t levelId, std::string name, INSGROUP *d1, INSGROUP *d2, AddrType op1, AddrType op2,
                         AddrType out, uint32_t modId) {
  Instruction *ins = make(name, IP, levelId, modId, {d1, d2});
  ins->operandList = {op1, op2};
  ins->OutputOperand = out;
  record(DataInsMap, ins);
  return {ins};
}
