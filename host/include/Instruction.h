// Instruction.h — one LIMB-level operation of a stage.
// The reference's Instruction (include/Instruction.h:26-188) is one 256-coefficient batch of one limb and
// carries only address tokens and an op kind; a GPU launch covers a whole stage, so this build's record is
// per limb, stands for `refInstructions` upstream instructions (batchCount of them; x InLevel for BCONV),
// and carries what the timing model lacks: modulus id, EWE opcode, constants, Galois element, input basis.
#ifndef HOMULATOR_INSTRUCTION_H
#define HOMULATOR_INSTRUCTION_H
#include "Basic.h"

enum ins_ops { NTT, INTT, MULT, MADD, MSUB, BCONV_STEP1, BCONV_STEP2, AUTO, DS, PRNG, IP, FETCH_RF, STORE_RF };
extern std::vector<std::string> ins_ops_name;

// the arithmetic of an EWE ("MULT"-tagged) instruction; numbering = include/homulator_hip.h hm_ewe_op
enum ewe_opcode {
  EWE_MUL = 0, EWE_MAC2 = 1, EWE_MAC_ADD = 2, EWE_ADD = 3, EWE_SUB = 4,
  EWE_MUL_CONST = 5, EWE_SUB_SCALE = 6, EWE_COPY = 7, EWE_SUB_SCALE_ADD = 8
};

class Instruction {
public:
  ins_ops ops;
  std::string Name;
  uint32_t level_id = 0;   // limb index inside its buffer (reference: level_id)
  uint32_t mod_id = 0;     // modulus id (q_i: i, p_j: maxLevel + j)
  std::vector<AddrType> operandList;  // up to 4 operands (0 = "fake operand", src/mem.cpp:30); BCONV: the inputs
  AddrType OutputOperand = 0;
  ewe_opcode opcode = EWE_MUL;
  bool hasConstant = false;
  uint64_t constant = 0;   // per-limb constant of MUL_CONST / SUB_SCALE, or the INTT epilogue scale
  uint32_t galois = 0;
  bool passthrough = false;  // an NTT-kind instruction whose input is already in evaluation form (copy)
  std::vector<uint32_t> inMods;  // BCONV: modulus ids of the inputs
  unsigned long long refInstructions = 0;  // upstream instructions this record stands for
  unsigned long long refExtra = 0;         // ... of records folded into a BCONV record (not scaled by its MAC-port count)
  // set by the backend's fusion passes (Arch::fusePasses), never by the generators:
  bool fusedSubScale = false;          // forward NTT whose epilogue is out = (minuend - NTT(in)) * constant [+ addend]
  AddrType fMinuend = 0, fAddend = 0;  // fAddend == 0: no addend
  // merged ModDown + rescale: the transform's input is in + fMixConst * fMix, the addend is scaled by fAddendConst
  AddrType fMix = 0;                   // 0: no prologue
  uint64_t fMixConst = 0, fAddendConst = 0;  // fAddendConst == 0: addend as is
  // (9) the transform's input operandList[0] is this base conversion, computed inside the transform's first pass (never written)
  std::vector<AddrType> fConvIn;
  std::vector<uint32_t> fConvMods;     // ... its input moduli
  // (10) a BCONV_STEP2 record with the element-wise epilogue out = (fSubFrom - conv) * constant [+ fAdd] (the rescale residue of 4c)
  bool ipInvOut = false;     // IP record (7b, round 5): its outputs leave the kernel as the first pass of their inverse transform
  bool secondOnly = false;   // INTT record (7b): the first pass has been run into its output limb by the producing IP record
  bool packedOut = false;    // INTT record (11, round 5): the output is stored in the split-30 packed form (only base conversions read it)
  bool inPacked = false;     // BCONV record / fused transform with fConvIn (11): the conversion's inputs are stored packed
  std::vector<uint8_t> ipConvPacked;   // IP record (11): per digit, the inputs of the digit's fused conversion are stored packed
  // (12, round 6) an AUTO record folded into its readers: an INTT record reads its input, a fused forward transform its addend, through X -> X^g
  uint32_t inGalois = 0, fAddendGalois = 0;   // 0: as stored
  uint32_t ipXGalois = 0;                     // ... and a transform x key record its evaluation-form digits (hm_ntt_ip_desc.x_galois)
  bool fusedEpi = false;
  AddrType fSubFrom = 0, fAdd = 0;
  bool fusedTensor = false;            // MAC2 record that also produces d0 -> extraOutputs[0] and d2 -> extraOutputs[1]
  std::vector<AddrType> extraOutputs;
  // fused inner product (ops == IP): out_k = sum_j ipX[j] * ipY[k][j]; out_0 = OutputOperand, out_1 = extraOutputs[0]
  std::vector<AddrType> ipX;
  std::vector<std::vector<AddrType>> ipY;
  // fused NTT-epilogue x key MAC (Arch::fusePasses (7)): digit j with ipCoeff[j] set is still in coefficient
  // form at ipSrc[j] and goes through the forward transform inside the inner-product kernel; ipX[j] (the buffer the separate
  // transform would have written: NTTOut_beta(j)) then only serves as the scratch of its first pass
  std::vector<AddrType> ipSrc;
  std::vector<uint8_t> ipCoeff;
  // (8) the base conversion that produces such a digit moves into the transform's first pass as well: ipConvIn[j] = the conversion's
  // input limbs, ipConvMods[j] their moduli (empty: digit j is not converted inside the kernel)
  std::vector<std::vector<AddrType>> ipConvIn;
  std::vector<std::vector<uint32_t>> ipConvMods;
  std::vector<Instruction *> depsInsList;

  Instruction(std::string name, ins_ops op, uint32_t level) : ops(op), Name(std::move(name)), level_id(level) {}
  const std::string &GetOpName() const { return ins_ops_name[ops]; }
  const std::string &GetName() const { return Name; }
  uint32_t getinputCount() const { return (uint32_t)operandList.size(); }
  AddrType getOperand(uint32_t i) const { return operandList[i]; }
};

typedef std::vector<Instruction *> INSGROUP;
#endif
