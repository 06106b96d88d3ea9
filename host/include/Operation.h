// Operation.h — the FHE operation layer with the reference's class names, constructor signatures and
// simulate() entry (include/Operation.h:18-321): KeySwitch, TensorCompute, Rescale sub-builders and the op
// classes HMULT, HROTATE, HADD, PMULT, PADD, each `ctor(label, maxLevel, curLevel, alpha, Config*, Arch*)`.
// The constructors build the stage graph (same stage keys, same order, same buffer names as upstream's
// src/Operation.cpp) with the mathematically correct wiring of SURVEY.md Appendix A / C; simulate() executes
// it on the backend and prints the upstream banner and stat block.
#ifndef HOMULATOR_OPERATION_H
#define HOMULATOR_OPERATION_H
#include "Addr.h"
#include "Arch.h"
#include "Basic.h"
#include "Context.h"
#include "Driver.h"
#include "InsGen.h"

typedef std::map<std::string, std::vector<INSGROUP>> StageMap;  // stage key -> [level] -> instruction group

class KeySwitch {
private:
  std::vector<AddrType> *DataPool;
  std::map<AddrType, std::vector<Instruction *>> *DataInsMap;
  InsGen *insGenPointer;
  std::vector<AddrType> preAddr;
  uint32_t Level, Alpha, Beta, dnum, MaxLevel;
  AddrManage *memMange;
  Arch *arch;
  std::string baseName;
  StageMap KeySwicthInsMap;
  std::vector<std::string> KeySwitchInsMapName;

  uint32_t digitSize(uint32_t beta) const { return std::min(Alpha, Level - beta * Alpha); }
  uint32_t extMod(uint32_t t) const { return t < Level ? t : MaxLevel + (t - Level); }

public:
  KeySwitch(std::string labelName, uint32_t maxlevel, uint32_t level, uint32_t alpha,
            const std::vector<AddrType> &inputPolynomialAddress, std::vector<AddrType> *pool,
            std::map<AddrType, std::vector<Instruction *>> *map, InsGen *insgen, AddrManage *memoryMange);
  std::pair<StageMap, std::vector<std::string>> getInsMap() { return {KeySwicthInsMap, KeySwitchInsMapName}; }

  void ModUpINTT();
  void ModUpDecompFusionBConvStep1(uint32_t beta);
  void ModUpBConvStep2(uint32_t beta);
  void ModUpNTT(uint32_t beta);
  void InnerProduceOperation(uint64_t evkSeed);
  void ModDownINTT();
  void ModDownBConvStep1();
  void ModDownBConvStep2();
  void ModDowNTT();
  void ModDownSub();
};

class TensorCompute {
private:
  InsGen *insGenPointer;
  uint32_t currentLevel;
  AddrManage *memMange;
  std::string baseName;
  StageMap TensorComputeInsMap;
  std::vector<std::string> TensorComputeInsMapName;
  std::vector<AddrType> ciph1_c0, ciph1_c1, ciph2_c0, ciph2_c1;

public:
  TensorCompute(std::string labelName, uint32_t level, Ciphertext *cipher1, Ciphertext *cipher2,
                std::vector<AddrType> *pool, std::map<AddrType, std::vector<Instruction *>> *map, InsGen *insgen,
                AddrManage *memoryMange);
  void computeD0();
  void computeD1();
  void computeD2();
  std::pair<StageMap, std::vector<std::string>> getInsMap() { return {TensorComputeInsMap, TensorComputeInsMapName}; }
};

class Rescale {
private:
  std::map<AddrType, std::vector<Instruction *>> *DataInsMap;
  std::vector<AddrType> preAddr;
  InsGen *insGenPointer;
  uint32_t currentLevel;
  AddrManage *memMange;
  Arch *arch;
  std::string baseName;
  StageMap RescaleInsMap;
  std::vector<std::string> RescaleInsMapName;

public:
  Rescale(std::string labelName, uint32_t level, const std::vector<AddrType> &inputPolynomialAddress,
          std::vector<AddrType> *pool, std::map<AddrType, std::vector<Instruction *>> *map, InsGen *insgen,
          AddrManage *memoryMange);
  void NTTOps();
  void SubOps();
  void MulOps();
  std::pair<StageMap, std::vector<std::string>> getInsMap() { return {RescaleInsMap, RescaleInsMapName}; }
};

// common part of the five op classes: generators, driver, address plan, synthetic inputs, simulate()
class OperationBase {
protected:
  std::vector<AddrType> Datapool;
  std::map<AddrType, std::vector<Instruction *>> DataInsMap;
  InsGen *insgener;
  Driver *driver;
  AddrManage *addrManager = nullptr;
  Arch *arch;
  Config *config;
  std::string opName;       // HMULT, HROTATE, ...
  std::string label;        // constructor label ("test_hmult", ...): part of the rescale buffer names
  uint32_t maxLevel_, level_, alpha_;
  uint32_t batchSize, N;
  uint64_t seed;
  std::map<std::string, std::vector<AddrType>> namedInputs;   // ct1.c0, ct1.c1, ... for readBuffer
  std::map<std::string, std::vector<AddrType>> namedOutputs;  // out.c0, out.c1

  OperationBase(const std::string &op, Config *cfg, Arch *_arch, uint32_t maxLevel, uint32_t curLevel, uint32_t alpha);
  void dispatch(std::pair<StageMap, std::vector<std::string>> m);
  void inputCiphertext(const std::string &name, Ciphertext *ct, uint64_t seed);
  void inputPlaintext(const std::string &name, Plaintext *pt, uint64_t seed);
  void finishConstruction();  // registers every temporary with the backend

public:
  virtual ~OperationBase();
  bool simulate();   // upstream entry: banner, execute, stat block
  // backend = sim: run the cycle model to completion without the banner / progress output; false = no instruction retired
  // for 2000 cycles (upstream's dead-lock exit).  simulate() on the sim backend is this loop plus upstream's stdout.
  bool simulateCycles(bool verbose = false);
  void prepare();    // issue stages + allocate/fill/fuse (idempotent)
  double execute(uint32_t iters);  // ns per iteration of the whole op (device time)
  std::vector<AddrType> bufferAddrs(const std::string &name) const;  // named buffer (Malloc name, input or output alias)
  bool readBuffer(const std::string &name, uint64_t *host, uint32_t copy = 0);  // copy: op of the batch (config key `batch`)
  bool writeBuffer(const std::string &name, const uint64_t *host, uint32_t copy = 0);  // real data for an input / key buffer ([limbs][N], fully reduced)
  unsigned long long totalInstructions();
  Arch *getArch() { return arch; }
  std::vector<std::string> bufferNames() const;
  // continuous execution: this op's input ciphertext `input` ("ct1", "ct2") is the output ciphertext of `producer`
  void bindInput(const std::string &input, OperationBase *producer);
  uint32_t outputLevel() const;  // limbs of out.c0
  const std::string &name() const { return opName; }
};

class HMULT : public OperationBase {
  Ciphertext *c1, *c2;
public:
  HMULT(std::string labelName, uint32_t maxLevel, uint32_t currentLevel, uint32_t alpha, Config *cfg, Arch *_arch);
};
class HROTATE : public OperationBase {
  Ciphertext *ciph;
public:
  HROTATE(std::string labelName, uint32_t maxLevel, uint32_t currentLevel, uint32_t alpha, Config *cfg, Arch *_arch);
};
class HADD : public OperationBase {
  Ciphertext *c1, *c2;
public:
  HADD(std::string labelName, uint32_t maxLevel, uint32_t currentLevel, uint32_t alpha, Config *cfg, Arch *_arch);
};
class PMULT : public OperationBase {
  Ciphertext *ctx; Plaintext *ptx;
public:
  PMULT(std::string labelName, uint32_t maxLevel, uint32_t currentLevel, uint32_t alpha, Config *cfg, Arch *_arch);
};
class PADD : public OperationBase {
  Ciphertext *ctx; Plaintext *ptx;
public:
  PADD(std::string labelName, uint32_t maxLevel, uint32_t currentLevel, uint32_t alpha, Config *cfg, Arch *_arch);
};

// Continuous multi-op execution.  Upstream cannot chain operations ("NotSuppotr the continuous
// operation simulate", src/Operation.cpp:636): every op starts from freshly allocated inputs.  A chain keeps the
// ciphertext resident in HBM: op k+1's first input ciphertext IS op k's output (copied device to device,
// stream-ordered, no host round trip); its level follows the data (an hmult's rescale drops one limb).  Second
// operands (ct2 of hmult / hadd, the plaintext of pmult / padd) and the evaluation keys are synthetic as for single ops.
class OpChain {
  std::vector<Config *> cfgs;
  std::vector<Arch *> archs;
  std::vector<OperationBase *> ops;
public:
  // ops: comma-separated list of hmult | hrotate | hadd | pmult | padd, e.g. "hmult,hrotate,hadd,hmult"
  OpChain(const std::string &cfgPath, const std::string &opList, uint32_t maxLevel, uint32_t curLevel, uint32_t alpha,
          const std::map<std::string, uint32_t> &overrides = {});
  ~OpChain();
  size_t size() const { return ops.size(); }
  OperationBase *op(size_t i) { return ops.at(i); }
  void prepare();
  void run();                        // one pass over the whole chain, asynchronous
  void sync();
  double execute(uint32_t iters);    // wall ns per pass over the chain (host clock around enqueue + sync)
  bool simulate();                   // CLI: per-op banners and stat blocks, then the chain total
};
#endif
