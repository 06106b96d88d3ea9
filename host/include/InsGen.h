// InsGen.h — stage emitters with the reference's five entry points (include/InsGen.h:46-98:
// GenNTT, GenAUTO, GenEWE, GenBCONV, GenHPIP), at limb granularity (see Instruction.h).
#ifndef HOMULATOR_INSGEN_H
#define HOMULATOR_INSGEN_H
#include "Basic.h"
#include "Config.h"
#include "Instruction.h"

class Arch;
class InsGen {
private:
  Arch *arch_ = nullptr;
  uint64_t keySeed_ = 0;
  uint32_t batchSize, batchCount;
  std::vector<AddrType> *DataPool = nullptr;
  std::map<AddrType, std::vector<Instruction *>> *DataInsMap = nullptr;
  Instruction *make(const std::string &name, ins_ops op, uint32_t level, uint32_t modId,
                    std::initializer_list<INSGROUP *> deps);

public:
  explicit InsGen(Config *cfg);
  void setGlobalDatapPoll(std::vector<AddrType> *pool) { DataPool = pool; }
  void setGlobalDataInsMap(std::map<AddrType, std::vector<Instruction *>> *map) { DataInsMap = map; }
  uint32_t getbatchCount() const { return batchCount; }
  // What real execution needs and the timing model never did travels with the generator, so that the sub-builders keep
  // the reference's constructor signatures (include/Operation.h:48-54, 156-162): the backend (moduli and conversion
  // constants of the context) and the seed of the synthetic evaluation key.
  void setBackend(Arch *a) { arch_ = a; }
  Arch *backend() const { return arch_; }
  void setKeySeed(uint64_t s) { keySeed_ = s; }
  uint64_t keySeed() const { return keySeed_; }

  // ntt == true: forward; false: inverse.  scale (inverse only): extra epilogue constant, 0 = none.
  INSGROUP GenNTT(uint32_t levelId, std::string name, INSGROUP *depInsGroup, bool ntt, AddrType op1AddrStart,
                  AddrType opOutAddrStart, uint32_t modId, bool passthrough = false);
  INSGROUP GenAUTO(uint32_t levelId, std::string name, INSGROUP *depInsGroup, AddrType op1AddrStart,
                   AddrType opOutAddrStart, uint32_t galois, uint32_t modId = 0);
  // address 0 = unused operand, exactly like upstream
  INSGROUP GenEWE(uint32_t levelId, std::string name, INSGROUP *dep1, INSGROUP *dep2, INSGROUP *dep3, INSGROUP *dep4,
                  AddrType op1, AddrType op2, AddrType op3, AddrType op4, AddrType out, ewe_opcode opcode,
                  uint32_t modId, bool hasConstant = false, uint64_t constant = 0);
  // one output limb of a base conversion: InLevel inputs (already scaled), table `tableAddr` (address token
  // kept for the address plan; the real table is derived from the bases)
  INSGROUP GenBCONV(uint32_t levelId, uint32_t InLevel, std::string name, std::vector<INSGROUP> depInsGroupList,
                    std::vector<AddrType> op1AddrStartList, std::vector<uint32_t> inMods, AddrType tableAddr,
                    AddrType opOutAddrStart, uint32_t outMod);
  // dedicated inner-product unit (src/InsGen.cpp:356-406; dead code upstream, hasHPIPU = 0)
  INSGROUP GenHPIP(uint32_t levelId, std::string name, INSGROUP *dep1, INSGROUP *dep2, AddrType op1, AddrType op2,
                   AddrType out, uint32_t modId);
};
#endif
