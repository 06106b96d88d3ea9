// Operation.cpp — stage-graph builders.  Structure (stage keys, order, buffer names, limb counts) follows the
// reference's src/Operation.cpp, cited per function; the wiring is the mathematically correct one of
// SURVEY.md Appendix A (upstream only needs shapes, and mislabels several operands: Appendix C).
#include "Operation.h"
#include "SimProgram.h"

#include <chrono>
#include <sstream>

static std::string S(uint32_t v) { return std::to_string(v); }

// =====================================================================================================
// KeySwitch — reference: KeySwitch::KeySwitch src/Operation.cpp:9-54 (stage order), beta = ceil(l/alpha) :22
// =====================================================================================================
KeySwitch::KeySwitch(std::string labelName, uint32_t maxlevel, uint32_t level, uint32_t alpha,
                     const std::vector<AddrType> &inputPolynomialAddress, std::vector<AddrType> *pool,
                     std::map<AddrType, std::vector<Instruction *>> *map, InsGen *insgen, AddrManage *memoryMange) {
  Arch *arch_ = insgen->backend();             // same signature as upstream (include/Operation.h:48-54): the backend
  const uint64_t evkSeed = insgen->keySeed();  // and the key seed travel with the generator
  DataInsMap = map;
  DataPool = pool;
  MaxLevel = maxlevel;
  Level = level;
  Alpha = alpha;
  Beta = (Level + Alpha - 1) / Alpha;
  dnum = (MaxLevel + Alpha - 1) / Alpha;
  insGenPointer = insgen;
  memMange = memoryMange;
  arch = arch_;
  preAddr = inputPolynomialAddress;
  baseName = labelName + "_KeySwitch";

  ModUpINTT();
  memMange->MallocMem("ModUpDecompOffset", 1);
  memMange->MallocMem("ModUpDecompOut", Level);
  for (uint32_t be = 0; be < Beta; be++) {
    ModUpDecompFusionBConvStep1(be);
    ModUpBConvStep2(be);
    ModUpNTT(be);
  }
  InnerProduceOperation(evkSeed);
  ModDownINTT();
  ModDownBConvStep1();
  ModDownBConvStep2();
  ModDowNTT();
  ModDownSub();
}

// reference: KeySwitch::ModUpINTT :63-102 — one INTT per input limb; throws when the input has no producer
void KeySwitch::ModUpINTT() {
  memMange->MallocMem("ModUpINTTOut", Level);
  std::vector<INSGROUP> out;
  for (uint32_t l = 0; l < Level; l++) {
    auto prod = DataInsMap->find(preAddr[l]);
    if (prod == DataInsMap->end()) throw std::runtime_error("Error! This dependece need exists!\n\n");
    out.push_back(insGenPointer->GenNTT(l, baseName + "_ModUp_INTT(" + S(l) + ")_", &prod->second, false, preAddr[l],
                                        memMange->getAddr("ModUpINTTOut")[l], l));
  }
  KeySwicthInsMap["ModUp_INTT"] = out;
  KeySwitchInsMapName.push_back("ModUp_INTT");
}

// reference: ModUpDecompFusionBConvStep1 :104-135 — y_i = x_i * [(Q_Dj/q_i)^-1]_{q_i} for the limbs of digit j
void KeySwitch::ModUpDecompFusionBConvStep1(uint32_t beta) {
  const uint32_t dj = digitSize(beta);
  std::vector<uint32_t> inMods;
  for (uint32_t a = 0; a < dj; a++) inMods.push_back(beta * Alpha + a);
  const std::vector<uint64_t> qhatInv = arch->bconvScale(inMods);
  std::vector<INSGROUP> out;
  for (uint32_t a = 0; a < dj; a++) {
    const uint32_t cur = beta * Alpha + a;
    out.push_back(insGenPointer->GenEWE(cur, baseName + "_decompFusionBConvStep1_beta(" + S(beta) + ")_Level(" + S(a) + ")_",
                                        &KeySwicthInsMap["ModUp_INTT"][cur], nullptr, nullptr, nullptr,
                                        memMange->getAddr("ModUpINTTOut")[cur], memMange->getAddr("ModUpDecompOffset")[0], 0, 0,
                                        memMange->getAddr("ModUpDecompOut")[cur], EWE_MUL_CONST, cur, true, qhatInv[a]));
  }
  const std::string key = "ModUp_DecompOut" + S(beta) + ")";  // sic: upstream's key has the stray parenthesis (:133)
  KeySwicthInsMap[key] = out;
  KeySwitchInsMapName.push_back(key);
}

// reference: ModUpBConvStep2 :137-188 — every limb of the extended basis outside the digit, d_j-deep MAC each
void KeySwitch::ModUpBConvStep2(uint32_t beta) {
  const uint32_t dj = digitSize(beta), E = Level + Alpha, lo = beta * Alpha;
  memMange->MallocMemOneBatch("BConvMap_(" + S(beta) + ")", 1);
  memMange->MallocMem("BConvOut_(" + S(beta) + ")", E - dj);
  std::vector<AddrType> inAddrs;
  std::vector<uint32_t> inMods;
  std::vector<INSGROUP> deps;
  const std::string depKey = "ModUp_DecompOut" + S(beta) + ")";
  for (uint32_t a = 0; a < dj; a++) {
    inAddrs.push_back(memMange->getAddr("ModUpDecompOut")[lo + a]);
    inMods.push_back(lo + a);
    deps.push_back(KeySwicthInsMap[depKey][a]);
  }
  std::vector<INSGROUP> out;
  uint32_t o = 0;
  for (uint32_t t = 0; t < E; t++) {
    if (t >= lo && t < lo + dj) continue;
    out.push_back(insGenPointer->GenBCONV(o, dj, baseName + "_ModUpBConv_beta(" + S(beta) + ")_outLevel(" + S(t) + ")_", deps,
                                          inAddrs, inMods, memMange->getAddr("BConvMap_(" + S(beta) + ")")[0],
                                          memMange->getAddr("BConvOut_(" + S(beta) + ")")[o], extMod(t)));
    o++;
  }
  const std::string key = "ModUp_BCONV_(" + S(beta) + ")";
  KeySwicthInsMap[key] = out;
  KeySwitchInsMapName.push_back(key);
}

// reference: ModUpNTT :190-292 — E NTT instructions per digit.  Upstream also spends an NTT on each of the
// digit's own limbs; mathematically those are the original evaluation-form input limbs, so here they are
// pass-through records (a copy, or nothing once the consumers are redirected) that still count as NTTs.
void KeySwitch::ModUpNTT(uint32_t beta) {
  const uint32_t dj = digitSize(beta), E = Level + Alpha, lo = beta * Alpha;
  memMange->MallocMem("NTTOut_beta(" + S(beta) + ")", E);
  std::vector<INSGROUP> out;
  uint32_t o = 0;
  for (uint32_t t = 0; t < E; t++) {
    const AddrType dst = memMange->getAddr("NTTOut_beta(" + S(beta) + ")")[t];
    const std::string name = baseName + "_ModUp_NTT_beta(" + S(beta) + ")_Level(" + S(t) + ")_";
    if (t >= lo && t < lo + dj) {
      out.push_back(insGenPointer->GenNTT(t, name, &KeySwicthInsMap["ModUp_DecompOut" + S(beta) + ")"][t - lo], true, preAddr[t],
                                          dst, t, /*passthrough=*/true));
    } else {
      out.push_back(insGenPointer->GenNTT(t, name, &KeySwicthInsMap["ModUp_BCONV_(" + S(beta) + ")"][o], true,
                                          memMange->getAddr("BConvOut_(" + S(beta) + ")")[o], dst, extMod(t)));
      o++;
    }
  }
  const std::string key = "ModUp_NTT_(" + S(beta) + ")";
  KeySwicthInsMap[key] = out;
  KeySwitchInsMapName.push_back(key);
}

// reference: InnerProduceOperation :294-414 — acc_k = sum_j ext_j * evk_{j,k}; beta = 1: one product (:314-352);
// beta > 1: beta-1 MAC groups, the first with two products (:355-411).  The last group writes
// InnerProduceOut_Key<k> (upstream writes a temp and reads a buffer nobody produced: Appendix C item 3).
void KeySwitch::InnerProduceOperation(uint64_t evkSeed) {
  const uint32_t E = Level + Alpha;
  std::vector<uint32_t> extMods;
  for (uint32_t t = 0; t < E; t++) extMods.push_back(extMod(t));
  for (uint32_t k = 0; k < 2; k++) {
    const std::string K = S(k);
    memMange->MallocMem("InnerProduceOut_Key" + K, E);
    for (uint32_t be = 0; be < Beta; be++) {
      memMange->MallocMem("IP_Key" + K + "_" + S(be), E);
      // evaluation key limbs are inputs: deterministic synthetic stream (same layout as the oracle's synth_evk)
      arch->addInputFill(InputFill{memMange->getAddr("IP_Key" + K + "_" + S(be)), extMods, evkSeed + (be * 2 + k) * 1000ull, /*shared=*/true});
      if (Beta != 1 && be <= Beta - 2) memMange->MallocMem("InnerProduceOut_temp(" + S(be) + ")_Key" + K, E);
    }
    auto ext = [&](uint32_t j) { return memMange->getAddr("NTTOut_beta(" + S(j) + ")"); };
    auto key = [&](uint32_t j) { return memMange->getAddr("IP_Key" + K + "_" + S(j)); };
    if (Beta == 1) {
      std::vector<INSGROUP> g;
      for (uint32_t ml = 0; ml < E; ml++)
        g.push_back(insGenPointer->GenEWE(ml, baseName + "_InnerProducOperation(0)_Level(" + S(ml) + ")_Key(" + K + ")",
                                          &KeySwicthInsMap["ModUp_NTT_(0)"][ml], nullptr, nullptr, nullptr, ext(0)[ml], key(0)[ml],
                                          0, 0, memMange->getAddr("InnerProduceOut_Key" + K)[ml], EWE_MUL, extMods[ml]));
      KeySwicthInsMap["InnerProOut_(0)_Key" + K] = g;
      KeySwitchInsMapName.push_back("InnerProOut_(0)_Key" + K);
    } else {
      for (uint32_t be = 0; be < Beta - 1; be++) {
        std::vector<INSGROUP> g;
        const std::string outKey = (be < Beta - 2) ? "InnerProduceOut_temp(" + S(be) + ")_Key" + K : "InnerProduceOut_Key" + K;
        for (uint32_t ml = 0; ml < E; ml++) {
          const std::string name = baseName + "_InnerProducOperation(" + S(be) + ")_Level(" + S(ml) + ")_Key(" + K + ")";
          const AddrType outaddr = memMange->getAddr(outKey)[ml];
          if (be == 0)
            g.push_back(insGenPointer->GenEWE(ml, name, &KeySwicthInsMap["ModUp_NTT_(0)"][ml], nullptr,
                                              &KeySwicthInsMap["ModUp_NTT_(1)"][ml], nullptr, ext(0)[ml], key(0)[ml], ext(1)[ml],
                                              key(1)[ml], outaddr, EWE_MAC2, extMods[ml]));
          else
            g.push_back(insGenPointer->GenEWE(ml, name, &KeySwicthInsMap["ModUp_NTT_(" + S(be + 1) + ")"][ml], nullptr,
                                              &KeySwicthInsMap["InnerProOut_(" + S(be - 1) + ")_Key" + K][ml], nullptr,
                                              ext(be + 1)[ml], key(be + 1)[ml],
                                              memMange->getAddr("InnerProduceOut_temp(" + S(be - 1) + ")_Key" + K)[ml], 0, outaddr,
                                              EWE_MAC_ADD, extMods[ml]));
        }
        KeySwicthInsMap["InnerProOut_(" + S(be) + ")_Key" + K] = g;
        KeySwitchInsMapName.push_back("InnerProOut_(" + S(be) + ")_Key" + K);
      }
    }
  }
}

// reference: ModDownINTT :417-445 — INTT of the alpha special-prime limbs of each inner-product output
// (extended limb order = Q limbs then P limbs: Appendix A (3))
void KeySwitch::ModDownINTT() {
  const std::string last = "InnerProOut_(" + S(Beta == 1 ? 0 : Beta - 2) + ")_Key";
  for (uint32_t k = 0; k < 2; k++) {
    memMange->MallocMem("INTTOut_ModDown_Key(" + S(k) + ")", Alpha);
    std::vector<INSGROUP> g;
    for (uint32_t l = 0; l < Alpha; l++)
      g.push_back(insGenPointer->GenNTT(l, baseName + "_ModDown_INTT(" + S(l) + ")_Key(" + S(k) + ")",
                                        &KeySwicthInsMap[last + S(k)][Level + l], false,
                                        memMange->getAddr("InnerProduceOut_Key" + S(k))[Level + l],
                                        memMange->getAddr("INTTOut_ModDown_Key(" + S(k) + ")")[l], MaxLevel + l));
    KeySwicthInsMap["ModDownINTTOut_Key(" + S(k) + ")"] = g;
    KeySwitchInsMapName.push_back("ModDownINTTOut_Key(" + S(k) + ")");
  }
}

// reference: ModDownBConvStep1 :447-487 — y_p = a_p * [(P/p)^-1]_p
void KeySwitch::ModDownBConvStep1() {
  memMange->MallocMem("ModDownBConvStep1_Ref", 2);
  std::vector<uint32_t> pMods;
  for (uint32_t l = 0; l < Alpha; l++) pMods.push_back(MaxLevel + l);
  const std::vector<uint64_t> phatInv = arch->bconvScale(pMods);
  for (uint32_t k = 0; k < 2; k++) {
    memMange->MallocMem("ModDownBConvStep1_Key(" + S(k) + ")", Alpha);
    std::vector<INSGROUP> g;
    for (uint32_t l = 0; l < Alpha; l++)
      g.push_back(insGenPointer->GenEWE(l, baseName + "_ModDownBConvStep1_Level(" + S(l) + ")_Key(" + S(k) + ")",
                                        &KeySwicthInsMap["ModDownINTTOut_Key(" + S(k) + ")"][l], nullptr, nullptr, nullptr,
                                        memMange->getAddr("INTTOut_ModDown_Key(" + S(k) + ")")[l],
                                        memMange->getAddr("ModDownBConvStep1_Ref")[k], 0, 0,
                                        memMange->getAddr("ModDownBConvStep1_Key(" + S(k) + ")")[l], EWE_MUL_CONST, MaxLevel + l,
                                        true, phatInv[l]));
    KeySwicthInsMap["ModDownBConvStep1_Key(" + S(k) + ")"] = g;
    KeySwitchInsMapName.push_back("ModDownBConvStep1_Key(" + S(k) + ")");
  }
}

// reference: ModDownBConvStep2 :489-519 — P -> Q conversion, alpha inputs per output limb
void KeySwitch::ModDownBConvStep2() {
  memMange->MallocMemOneBatch("ModdownBConvMap", 1);
  std::vector<uint32_t> pMods;
  for (uint32_t l = 0; l < Alpha; l++) pMods.push_back(MaxLevel + l);
  for (uint32_t k = 0; k < 2; k++) {
    memMange->MallocMem("ModdownBConvOut_Key" + S(k), Level);
    std::vector<INSGROUP> g;
    for (uint32_t ol = 0; ol < Level; ol++)
      g.push_back(insGenPointer->GenBCONV(ol, Alpha, baseName + "_ModDownBConv_outLevel(" + S(ol) + ")_Key(" + S(k) + ")",
                                          KeySwicthInsMap["ModDownBConvStep1_Key(" + S(k) + ")"],
                                          memMange->getAddr("ModDownBConvStep1_Key(" + S(k) + ")"), pMods,
                                          memMange->getAddr("ModdownBConvMap")[0],
                                          memMange->getAddr("ModdownBConvOut_Key" + S(k))[ol], ol));
    KeySwicthInsMap["ModDown_BCONV_Key(" + S(k) + ")"] = g;
    KeySwitchInsMapName.push_back("ModDown_BCONV_Key(" + S(k) + ")");
  }
}

// reference: ModDowNTT :521-546 (passes ntt=false there: a labelling slip, Appendix C item 5)
void KeySwitch::ModDowNTT() {
  for (uint32_t k = 0; k < 2; k++) {
    memMange->MallocMem("NTTOut_ModDown_Key(" + S(k) + ")", Level);
    std::vector<INSGROUP> g;
    for (uint32_t l = 0; l < Level; l++)
      g.push_back(insGenPointer->GenNTT(l, baseName + "_ModDown_NTT(" + S(l) + ")_Key(" + S(k) + ")",
                                        &KeySwicthInsMap["ModDown_BCONV_Key(" + S(k) + ")"][l], true,
                                        memMange->getAddr("ModdownBConvOut_Key" + S(k))[l],
                                        memMange->getAddr("NTTOut_ModDown_Key(" + S(k) + ")")[l], l));
    KeySwicthInsMap["ModDownNTTOut_Key(" + S(k) + ")"] = g;
    KeySwitchInsMapName.push_back("ModDownNTTOut_Key(" + S(k) + ")");
  }
}

// reference: ModDownSub :548-590 — ks_k,i = (acc_k,i - w_k,i) * [P^-1]_{q_i}
void KeySwitch::ModDownSub() {
  const std::string last = "InnerProOut_(" + S(Beta == 1 ? 0 : Beta - 2) + ")_Key";
  for (uint32_t k = 0; k < 2; k++) {
    memMange->MallocMem("KeySwitchFinalOutput_Key(" + S(k) + ")", Level);
    std::vector<INSGROUP> g;
    for (uint32_t l = 0; l < Level; l++) {
      const uint64_t q = arch->modulus(l);
      unsigned __int128 P = 1;
      for (uint32_t p = 0; p < Alpha; p++) P = (P * (arch->modulus(MaxLevel + p) % q)) % q;
      // P^-1 mod q by Fermat
      uint64_t base = (uint64_t)P, e = q - 2, r = 1;
      for (; e; e >>= 1) {
        if (e & 1) r = (uint64_t)(((unsigned __int128)r * base) % q);
        base = (uint64_t)(((unsigned __int128)base * base) % q);
      }
      g.push_back(insGenPointer->GenEWE(l, baseName + "_ModDownSub_Level(" + S(l) + ")_Key(" + S(k) + ")",
                                        &KeySwicthInsMap["ModDownNTTOut_Key(" + S(k) + ")"][l], nullptr,
                                        &KeySwicthInsMap[last + S(k)][l], nullptr,
                                        memMange->getAddr("InnerProduceOut_Key" + S(k))[l], 0,
                                        memMange->getAddr("NTTOut_ModDown_Key(" + S(k) + ")")[l], 0,
                                        memMange->getAddr("KeySwitchFinalOutput_Key(" + S(k) + ")")[l], EWE_SUB_SCALE, l, true, r));
    }
    KeySwicthInsMap["KeySwitchFinalOutput_Key(" + S(k) + ")"] = g;
    KeySwitchInsMapName.push_back("KeySwitchFinalOutput_Key(" + S(k) + ")");
  }
}

// =====================================================================================================
// TensorCompute — reference: src/Operation.cpp:592-739 (d0 = c00*c10, d1 = c00*c11 + c01*c10, d2 = c01*c11)
// =====================================================================================================
TensorCompute::TensorCompute(std::string labelName, uint32_t level, Ciphertext *cipher1, Ciphertext *cipher2,
                             std::vector<AddrType> *, std::map<AddrType, std::vector<Instruction *>> *, InsGen *insgen,
                             AddrManage *memoryMange) {
  currentLevel = level;
  insGenPointer = insgen;
  memMange = memoryMange;
  baseName = labelName + "_TensorCompute";
  ciph1_c0 = cipher1->getC0Addr(); ciph1_c1 = cipher1->getC1Addr();
  ciph2_c0 = cipher2->getC0Addr(); ciph2_c1 = cipher2->getC1Addr();
  computeD0();
  computeD1();
  computeD2();
}
void TensorCompute::computeD0() {  // :624-660
  memMange->MallocMem("TensorD0Out", currentLevel);
  std::vector<INSGROUP> g;
  for (uint32_t l = 0; l < currentLevel; l++)
    g.push_back(insGenPointer->GenEWE(l, baseName + "_D0_Level(" + S(l) + ")", nullptr, nullptr, nullptr, nullptr, ciph1_c0[l],
                                      ciph2_c0[l], 0, 0, memMange->getAddr("TensorD0Out")[l], EWE_MUL, l));
  TensorComputeInsMap["TensorCompute_INS_D0"] = g;
  TensorComputeInsMapName.push_back("TensorCompute_INS_D0");
}
void TensorCompute::computeD1() {  // :662-699
  memMange->MallocMem("TensorD1Out", currentLevel);
  std::vector<INSGROUP> g;
  for (uint32_t l = 0; l < currentLevel; l++)
    g.push_back(insGenPointer->GenEWE(l, baseName + "_D1_Level(" + S(l) + ")", nullptr, nullptr, nullptr, nullptr, ciph1_c0[l],
                                      ciph2_c1[l], ciph1_c1[l], ciph2_c0[l], memMange->getAddr("TensorD1Out")[l], EWE_MAC2, l));
  TensorComputeInsMap["TensorCompute_INS_D1"] = g;
  TensorComputeInsMapName.push_back("TensorCompute_INS_D1");
}
void TensorCompute::computeD2() {  // :701-739
  memMange->MallocMem("TensorD2Out", currentLevel);
  std::vector<INSGROUP> g;
  for (uint32_t l = 0; l < currentLevel; l++)
    g.push_back(insGenPointer->GenEWE(l, baseName + "_D2_Level(" + S(l) + ")", nullptr, nullptr, nullptr, nullptr, ciph1_c1[l],
                                      ciph2_c1[l], 0, 0, memMange->getAddr("TensorD2Out")[l], EWE_MUL, l));
  TensorComputeInsMap["TensorCompute_INS_D2"] = g;
  TensorComputeInsMapName.push_back("TensorCompute_INS_D2");
}

// =====================================================================================================
// Rescale — reference: src/Operation.cpp:741-911.  r = INTT_{q_last}(x_last);
// x'_i = (x_i - NTT_{q_i}(r)) * [q_last^-1]_{q_i}.  Upstream issues ONE forward NTT per polynomial (:810-822);
// the maths needs one per remaining limb.  The extra NTTs are emitted with refInstructions = 0 so that the
// instruction total still equals upstream's, and write into <base>_Rescale_Mul_Offset (a constants token
// upstream; same buffer list, every stage writes its own buffer).
// =====================================================================================================
Rescale::Rescale(std::string labelName, uint32_t level, const std::vector<AddrType> &inputPolynomialAddress,
                 std::vector<AddrType> *, std::map<AddrType, std::vector<Instruction *>> *map, InsGen *insgen,
                 AddrManage *memoryMange) {
  Arch *arch_ = insgen->backend();  // upstream's signature (include/Operation.h:156-162)
  DataInsMap = map;
  currentLevel = level;
  insGenPointer = insgen;
  memMange = memoryMange;
  arch = arch_;
  preAddr = inputPolynomialAddress;
  baseName = labelName + "_Rescale";
  if (currentLevel < 2) throw std::runtime_error("Rescale needs at least two limbs");
  NTTOps();
  SubOps();
  MulOps();
}
void Rescale::NTTOps() {  // :766-825
  // all five buffers up front, in upstream's allocation order (:768-769, :828, :881-882)
  memMange->MallocMem(baseName + "_ResINTTOut", 1);
  memMange->MallocMem(baseName + "_ResNTTOut", 1);
  memMange->MallocMem(baseName + "_Rescale_SubOut", currentLevel - 1);
  memMange->MallocMem(baseName + "_Rescale_MulOut", currentLevel - 1);
  memMange->MallocMem(baseName + "_Rescale_Mul_Offset", currentLevel - 1);
  const uint32_t last = currentLevel - 1;
  auto prod = DataInsMap->find(preAddr[last]);
  if (prod == DataInsMap->end()) throw std::runtime_error("Error! This dependece need exists!\n\n");
  std::vector<INSGROUP> intt;
  intt.push_back(insGenPointer->GenNTT(0, baseName + "_Rescale_INTT(0)_", &prod->second, false, preAddr[last],
                                       memMange->getAddr(baseName + "_ResINTTOut")[0], last));
  RescaleInsMap["Rescale_INTT"] = intt;
  RescaleInsMapName.push_back("Rescale_INTT");

  const uint64_t qlast = arch->modulus(last);
  std::vector<INSGROUP> ntt;
  for (uint32_t l = 0; l < currentLevel - 1; l++) {
    // the forward transform accepts inputs below 4 q_l (lazy butterflies), so r in [0, q_last) needs no
    // separate reduction mod q_l as long as q_last < 4 q_l — true for any chain of same-size primes
    if (qlast >= 4 * arch->modulus(l)) throw std::runtime_error("Rescale: q_last >= 4 q_l is not supported");
    INSGROUP g = insGenPointer->GenNTT(l, baseName + "_Rescale_NTT_level(" + S(l) + ")", &intt[0], true,
                                       memMange->getAddr(baseName + "_ResINTTOut")[0],
                                       memMange->getAddr(baseName + "_Rescale_Mul_Offset")[l], l);
    if (l > 0) g[0]->refInstructions = 0;
    ntt.push_back(g);
  }
  RescaleInsMap["Rescale_NTT"] = ntt;
  RescaleInsMapName.push_back("Rescale_NTT");
}
void Rescale::SubOps() {  // :827-875
  std::vector<INSGROUP> g;
  for (uint32_t l = 0; l < currentLevel - 1; l++) {
    g.push_back(insGenPointer->GenEWE(l, baseName + "_Rescale_Sub_Level(" + S(l) + ")", &RescaleInsMap["Rescale_NTT"][l], nullptr,
                                      nullptr, nullptr, preAddr[l], 0, memMange->getAddr(baseName + "_Rescale_Mul_Offset")[l], 0,
                                      memMange->getAddr(baseName + "_Rescale_SubOut")[l], EWE_SUB, l));
  }
  RescaleInsMap["Rescale_SUB"] = g;
  RescaleInsMapName.push_back("Rescale_SUB");
}
void Rescale::MulOps() {  // :877-910
  const uint64_t qlast = arch->modulus(currentLevel - 1);
  std::vector<INSGROUP> g;
  for (uint32_t l = 0; l < currentLevel - 1; l++) {
    const uint64_t q = arch->modulus(l);
    uint64_t base = qlast % q, e = q - 2, r = 1;
    for (; e; e >>= 1) {
      if (e & 1) r = (uint64_t)(((unsigned __int128)r * base) % q);
      base = (uint64_t)(((unsigned __int128)base * base) % q);
    }
    g.push_back(insGenPointer->GenEWE(l, baseName + "_Rescale_Mul_Level(" + S(l) + ")", &RescaleInsMap["Rescale_SUB"][l], nullptr,
                                      nullptr, nullptr, memMange->getAddr(baseName + "_Rescale_SubOut")[l],
                                      memMange->getAddr(baseName + "_Rescale_Mul_Offset")[l], 0, 0,
                                      memMange->getAddr(baseName + "_Rescale_MulOut")[l], EWE_MUL_CONST, l, true, r));
  }
  RescaleInsMap["Rescale_Mul"] = g;
  RescaleInsMapName.push_back("Rescale_Mul");
}

// =====================================================================================================
// OperationBase
// =====================================================================================================
OperationBase::OperationBase(const std::string &op, Config *cfg, Arch *_arch, uint32_t maxLevel, uint32_t curLevel, uint32_t alpha)
    : arch(_arch), config(cfg), opName(op), maxLevel_(maxLevel), level_(curLevel), alpha_(alpha) {
  insgener = new InsGen(cfg);
  driver = new Driver(cfg);
  batchSize = cfg->getValue("batchSize");
  N = cfg->getValue("N");
  seed = cfg->getValueOr("seed", 0x484F4D55u);
  insgener->setGlobalDatapPoll(&Datapool);
  insgener->setGlobalDataInsMap(&DataInsMap);
  insgener->setBackend(arch);
  insgener->setKeySeed(seed + 10000);
  arch->bindParams(maxLevel, curLevel, alpha);
  Datapool.push_back(BASEADDRESS);
}
OperationBase::~OperationBase() {
  for (auto &kv : DataInsMap)
    for (Instruction *i : kv.second) delete i;
  delete addrManager;
  delete driver;
  delete insgener;
}
void OperationBase::dispatch(std::pair<StageMap, std::vector<std::string>> m) {
  for (auto &key : m.second) driver->dispatchInstructions(key, m.first[key]);
}
void OperationBase::inputCiphertext(const std::string &name, Ciphertext *ct, uint64_t s) {
  std::vector<uint32_t> mods;
  for (uint32_t l = 0; l < ct->level(); l++) mods.push_back(l);
  arch->registerLimbs(ct->getC0Addr());
  arch->registerLimbs(ct->getC1Addr());
  arch->addInputFill(InputFill{ct->getC0Addr(), mods, s});
  arch->addInputFill(InputFill{ct->getC1Addr(), mods, s + 1000});
  namedInputs[name + ".c0"] = ct->getC0Addr();
  namedInputs[name + ".c1"] = ct->getC1Addr();
}
void OperationBase::inputPlaintext(const std::string &name, Plaintext *pt, uint64_t s) {
  std::vector<uint32_t> mods;
  for (uint32_t l = 0; l < pt->getC0Addr().size(); l++) mods.push_back(l);
  arch->registerLimbs(pt->getC0Addr());
  arch->addInputFill(InputFill{pt->getC0Addr(), mods, s});
  namedInputs[name] = pt->getC0Addr();
}
void OperationBase::finishConstruction() {
  for (const std::string &n : addrManager->names()) arch->registerLimbs(addrManager->getAddr(n));
}
std::vector<AddrType> OperationBase::bufferAddrs(const std::string &name) const {
  auto i = namedInputs.find(name);
  if (i != namedInputs.end()) return i->second;
  auto o = namedOutputs.find(name);
  if (o != namedOutputs.end()) return o->second;
  return addrManager->getAddr(name);
}
std::vector<std::string> OperationBase::bufferNames() const {
  std::vector<std::string> v;
  for (auto &kv : namedInputs) v.push_back(kv.first);
  for (auto &kv : namedOutputs) v.push_back(kv.first);
  for (auto &n : addrManager->names()) v.push_back(n);
  return v;
}
bool OperationBase::readBuffer(const std::string &name, uint64_t *host, uint32_t copy) { return arch->readLimbs(bufferAddrs(name), host, copy); }
bool OperationBase::writeBuffer(const std::string &name, const uint64_t *host, uint32_t copy) { prepare(); return arch->writeLimbs(bufferAddrs(name), host, copy); }
unsigned long long OperationBase::totalInstructions() { prepare(); return driver->getTotalIns(); }
void OperationBase::bindInput(const std::string &input, OperationBase *producer) {
  for (const char *part : {".c0", ".c1"}) {
    auto in = namedInputs.find(input + part);
    auto out = producer->namedOutputs.find(std::string("out") + part);
    if (in == namedInputs.end()) throw std::runtime_error(opName + " has no input ciphertext " + input);
    if (out == producer->namedOutputs.end()) throw std::runtime_error(producer->opName + " has no output ciphertext");
    arch->bindInput(in->second, producer->arch, out->second);
  }
}
uint32_t OperationBase::outputLevel() const {
  auto o = namedOutputs.find("out.c0");
  if (o == namedOutputs.end()) throw std::runtime_error(opName + " has no output ciphertext");
  return (uint32_t)o->second.size();
}

void OperationBase::prepare() {
  driver->IssueInsFromDramToChip(arch);
  arch->prepare();
}
double OperationBase::execute(uint32_t iters) {
  prepare();
  return arch->timedRun(iters);
}

// reference: HMULT::simulate src/Operation.cpp:1025-1112 (the five simulate() bodies are textual copies).
// The stdout contract is kept: banner, start time, [progress], completion block, stat
// block.  There is no per-cycle loop to report on; each update() launches one stage on the GPU.
// backend = sim: the reference's loop itself (src/Operation.cpp:1046-1087), on the build's own cycle model
bool OperationBase::simulateCycles(bool verbose) {
  if (arch->backend() != Arch::BACKEND_SIM) throw std::runtime_error("simulateCycles: backend is not sim");
  driver->IssueInsFromDramToChip(arch);
  arch->loadSim(buildSimProgram(opName, label, level_, alpha_, config, *addrManager, namedInputs));
  const unsigned long long TotalIns = driver->getTotalIns();
  if (arch->simModel()->totalIns() != TotalIns)
    throw std::runtime_error("sim backend: literal program has " + std::to_string(arch->simModel()->totalIns()) + " instructions, the stage graph accounts for " + std::to_string(TotalIns));
  unsigned long long exeInsCycle = 0, traced = ~0ull;
  const bool trace = getenv("HOMULATOR_SIM_TRACE") != nullptr;  // "<cycle> <retired>" whenever the count moves (oracle/ref_dump.cpp prints the same)
  time_t periodTime = time(0);
  while (!arch->simulateComplete()) {
    driver->IssueDataFromDramToChip(arch->getMemController());
    arch->update();
    const unsigned long long cycle = arch->getCycle();
    if (trace && arch->getcompletedIns() != traced) {
      traced = arch->getcompletedIns();
      std::fprintf(stderr, "%llu %llu\n", cycle, traced);
    }
    if (cycle % 2000 == 0) {
      const unsigned long long exeins = arch->getcompletedIns() - exeInsCycle;
      if (exeins == 0) {
        if (verbose) {
          std::cout << "We have executed " << exeins << " instruction(s) in this period!\n";
          arch->state();
          std::cout << "\n";
        }
        return false;
      }
      if (verbose) {
        std::cout << "\nFHE-Sim running " << cycle << " cycles!\n";
        std::cout << "We have executed " << arch->getcompletedIns() << " instructions!\n";
        const unsigned long long remainIns = TotalIns - arch->getcompletedIns();
        std::cout << "Remaining " << remainIns << " instructions!\n";
        std::cout << "We have executed " << exeins << " instruction(s) in this period!\n";
        const time_t nowTime = time(0);
        const double speed = static_cast<double>(exeins) / (nowTime - periodTime);
        std::cout << "Estimated time remaining " << static_cast<double>(remainIns) / speed / 60 << " minutes\n";
        periodTime = nowTime;
      }
      exeInsCycle = arch->getcompletedIns();
    }
  }
  return true;
}

bool OperationBase::simulate() {
  if (arch->backend() == Arch::BACKEND_SIM) {
    std::cout << "\n\nWelcome! Start simulating " << opName << "!\n\n";
    time_t t0 = time(0);
    std::cout << "Start time: " << ctime(&t0) << std::endl;
    simulateCycles(true);
    time_t t1 = time(0);
    std::cout << "\n\nCompleted Simulate!\n";
    std::cout << "FHE-Sim Total simulated\t" << arch->getCycle() << " cycles!\n\n";
    std::cout << "End time: " << ctime(&t1) << std::endl;
    std::cout << "The simulator total cost\t" << static_cast<double>(t1 - t0) / 60 << " Minutes!\n";
    arch->shownStat();
    return true;
  }
  driver->IssueInsFromDramToChip(arch);
  const unsigned long long TotalIns = driver->getTotalIns();
  std::cout << "\n\nWelcome! Start simulating " << opName << "!\n\n";
  time_t currentTime = time(0);
  std::cout << "Start time: " << ctime(&currentTime) << std::endl;
  arch->prepare();
  arch->run();   // untimed warm-up of the whole plan (first-use table uploads, code-object loads); the plan is idempotent
  arch->sync();
  while (!arch->simulateComplete()) {
    driver->IssueDataFromDramToChip(arch->getMemController());
    arch->update();
  }
  arch->sync();
  const unsigned long long ns = arch->getCycle();
  std::cout << "\nFHE-Sim running " << ns << " cycles!\n";  // unit: device nanoseconds (see Arch.h)
  std::cout << "We have executed " << arch->getcompletedIns() << " instructions!\n";
  std::cout << "Remaining " << TotalIns - arch->getcompletedIns() << " instructions!\n";
  time_t currentTime2 = time(0);
  std::cout << "\n\nCompleted Simulate!\n";
  std::cout << "FHE-Sim Total simulated\t" << ns << " cycles!\n\n";
  std::cout << "End time: " << ctime(&currentTime2) << std::endl;
  std::cout << "The simulator total cost\t" << static_cast<double>(currentTime2 - currentTime) / 60 << " Minutes!\n";
  arch->shownStat();
  return true;
}

// =====================================================================================================
// op classes
// =====================================================================================================
// reference: HMULT::HMULT :913-1023.  Wiring: KS(d2); out0 = d0 + ks0; out1 = d1 + ks1 (Appendix C item 1)
HMULT::HMULT(std::string labelName, uint32_t maxLevel, uint32_t currentLevel, uint32_t alpha, Config *cfg, Arch *_arch)
    : OperationBase("HMULT", cfg, _arch, maxLevel, currentLevel, alpha) {
  label = labelName;
  c1 = new Ciphertext(currentLevel, N, Datapool, batchSize);
  c2 = new Ciphertext(currentLevel, N, Datapool, batchSize);
  inputCiphertext("ct1", c1, seed);
  inputCiphertext("ct2", c2, seed + 2000);
  addrManager = new AddrManage(Datapool.back() + 1, batchSize);
  addrManager->setGlobalDatapPoll(&Datapool);

  TensorCompute tcm(labelName, currentLevel, c1, c2, &Datapool, &DataInsMap, insgener, addrManager);
  dispatch(tcm.getInsMap());

  KeySwitch ksw(labelName, maxLevel, currentLevel, alpha, addrManager->getAddr("TensorD2Out"), &Datapool, &DataInsMap, insgener,
                addrManager);
  dispatch(ksw.getInsMap());

  StageMap hadd;
  std::vector<std::string> haddNames;
  for (uint32_t k = 0; k < 2; k++) {
    addrManager->MallocMem("HMULTHaddOutput(" + S(k) + ")", currentLevel);
    std::vector<INSGROUP> g;
    const auto ks = addrManager->getAddr("KeySwitchFinalOutput_Key(" + S(k) + ")");
    const auto d = addrManager->getAddr(k == 0 ? "TensorD0Out" : "TensorD1Out");
    for (uint32_t l = 0; l < currentLevel; l++)
      g.push_back(insgener->GenEWE(l, labelName + "_HMULTHadd_Level(" + S(l) + ")_k(" + S(k) + ")", nullptr, nullptr, nullptr, nullptr,
                                   ks[l], 0, d[l], 0, addrManager->getAddr("HMULTHaddOutput(" + S(k) + ")")[l], EWE_ADD, l));
    hadd["HMULT_Hadd_Key(" + S(k) + ")"] = g;
    haddNames.push_back("HMULT_Hadd_Key(" + S(k) + ")");
  }
  dispatch({hadd, haddNames});

  for (uint32_t k = 0; k < 2; k++) {
    Rescale res(labelName + "_" + S(k), currentLevel, addrManager->getAddr("HMULTHaddOutput(" + S(k) + ")"), &Datapool, &DataInsMap,
                insgener, addrManager);
    dispatch(res.getInsMap());
    namedOutputs[k == 0 ? "out.c0" : "out.c1"] = addrManager->getAddr(labelName + "_" + S(k) + "_Rescale_Rescale_MulOut");
  }
  finishConstruction();
}

// reference: HROTATE::HROTATE :1271-1358.  Wiring: c'_k = sigma_g(c_k); KS(c'_1); out0 = c'_0 + ks0; out1 = ks1
// (Appendix C item 2).  The Galois element is the config key `galois` (default 5 = rotation by one slot).
HROTATE::HROTATE(std::string labelName, uint32_t maxLevel, uint32_t currentLevel, uint32_t alpha, Config *cfg, Arch *_arch)
    : OperationBase("HROTATE", cfg, _arch, maxLevel, currentLevel, alpha) {
  label = labelName;
  ciph = new Ciphertext(currentLevel, N, Datapool, batchSize);
  inputCiphertext("ct1", ciph, seed);
  addrManager = new AddrManage(Datapool.back() + 1, batchSize);
  addrManager->setGlobalDatapPoll(&Datapool);
  const uint32_t galois = cfg->getValueOr("galois", 5);

  StageMap autoMap;
  std::vector<std::string> autoNames;
  for (uint32_t k = 0; k < 2; k++) {
    addrManager->MallocMem("AUTOOutput(" + S(k) + ")", currentLevel);
    const auto src = k == 0 ? ciph->getC0Addr() : ciph->getC1Addr();
    std::vector<INSGROUP> g;
    for (uint32_t l = 0; l < currentLevel; l++)
      g.push_back(insgener->GenAUTO(l, labelName + "_AUTO_Level(" + S(l) + ")_k(" + S(k) + ")", nullptr, src[l],
                                    addrManager->getAddr("AUTOOutput(" + S(k) + ")")[l], galois, l));
    autoMap["AUTO_Key(" + S(k) + ")"] = g;
    autoNames.push_back("AUTO_Key(" + S(k) + ")");
  }
  dispatch({autoMap, autoNames});

  KeySwitch ksw(labelName, maxLevel, currentLevel, alpha, addrManager->getAddr("AUTOOutput(1)"), &Datapool, &DataInsMap, insgener,
                addrManager);
  dispatch(ksw.getInsMap());

  addrManager->MallocMem("HROTATEOutput(1)", currentLevel);
  std::vector<INSGROUP> g;
  for (uint32_t l = 0; l < currentLevel; l++)
    g.push_back(insgener->GenEWE(l, labelName + "_HROTATEadd_Level(" + S(l) + ")", nullptr, nullptr, nullptr, nullptr,
                                 addrManager->getAddr("KeySwitchFinalOutput_Key(0)")[l], 0, addrManager->getAddr("AUTOOutput(0)")[l], 0,
                                 addrManager->getAddr("HROTATEOutput(1)")[l], EWE_ADD, l));
  driver->dispatchInstructions("HROTATE_Hadd", g);
  namedOutputs["out.c0"] = addrManager->getAddr("HROTATEOutput(1)");
  namedOutputs["out.c1"] = addrManager->getAddr("KeySwitchFinalOutput_Key(1)");
  finishConstruction();
}

// reference: HADD::HADD :1114-1176
HADD::HADD(std::string labelName, uint32_t maxLevel, uint32_t currentLevel, uint32_t alpha, Config *cfg, Arch *_arch)
    : OperationBase("HADD", cfg, _arch, maxLevel, currentLevel, alpha) {
  label = labelName;
  c1 = new Ciphertext(currentLevel, N, Datapool, batchSize);
  c2 = new Ciphertext(currentLevel, N, Datapool, batchSize);
  inputCiphertext("ct1", c1, seed);
  inputCiphertext("ct2", c2, seed + 2000);
  addrManager = new AddrManage(Datapool.back() + 1, batchSize);
  addrManager->setGlobalDatapPoll(&Datapool);
  for (uint32_t k = 0; k < 2; k++) {
    addrManager->MallocMem("HADDOutput(" + S(k) + ")", currentLevel);
    const auto a = k == 0 ? c1->getC0Addr() : c1->getC1Addr(), b = k == 0 ? c2->getC0Addr() : c2->getC1Addr();
    std::vector<INSGROUP> g;
    for (uint32_t l = 0; l < currentLevel; l++)
      g.push_back(insgener->GenEWE(l, labelName + "_HADD_Level(" + S(l) + ")_k(" + S(k) + ")", nullptr, nullptr, nullptr, nullptr, a[l],
                                   0, b[l], 0, addrManager->getAddr("HADDOutput(" + S(k) + ")")[l], EWE_ADD, l));
    driver->dispatchInstructions("HADD_Key(" + S(k) + ")", g);
    namedOutputs[k == 0 ? "out.c0" : "out.c1"] = addrManager->getAddr("HADDOutput(" + S(k) + ")");
  }
  finishConstruction();
}

// reference: PMULT::PMULT :1460-1523
PMULT::PMULT(std::string labelName, uint32_t maxLevel, uint32_t currentLevel, uint32_t alpha, Config *cfg, Arch *_arch)
    : OperationBase("PMULT", cfg, _arch, maxLevel, currentLevel, alpha) {
  label = labelName;
  ctx = new Ciphertext(currentLevel, N, Datapool, batchSize);
  ptx = new Plaintext(currentLevel, N, Datapool, batchSize);
  inputCiphertext("ct1", ctx, seed);
  inputPlaintext("pt", ptx, seed + 4000);
  addrManager = new AddrManage(Datapool.back() + 1, batchSize);
  addrManager->setGlobalDatapPoll(&Datapool);
  for (uint32_t k = 0; k < 2; k++) {
    addrManager->MallocMem("HMult" + S(k) + "Out", currentLevel);
    const auto a = k == 0 ? ctx->getC0Addr() : ctx->getC1Addr(), p = ptx->getC0Addr();
    std::vector<INSGROUP> g;
    for (uint32_t l = 0; l < currentLevel; l++)
      g.push_back(insgener->GenEWE(l, labelName + "_PMULT_Level(" + S(l) + ")_k(" + S(k) + ")", nullptr, nullptr, nullptr, nullptr, a[l],
                                   p[l], 0, 0, addrManager->getAddr("HMult" + S(k) + "Out")[l], EWE_MUL, l));
    driver->dispatchInstructions("PMULT_Key(" + S(k) + ")", g);
    namedOutputs[k == 0 ? "out.c0" : "out.c1"] = addrManager->getAddr("HMult" + S(k) + "Out");
  }
  finishConstruction();
}

// reference: PADD::PADD :1625-1680 (upstream adds the plaintext to both components; only c0 takes it)
PADD::PADD(std::string labelName, uint32_t maxLevel, uint32_t currentLevel, uint32_t alpha, Config *cfg, Arch *_arch)
    : OperationBase("PADD", cfg, _arch, maxLevel, currentLevel, alpha) {
  label = labelName;
  ctx = new Ciphertext(currentLevel, N, Datapool, batchSize);
  ptx = new Plaintext(currentLevel, N, Datapool, batchSize);
  inputCiphertext("ct1", ctx, seed);
  inputPlaintext("pt", ptx, seed + 4000);
  addrManager = new AddrManage(Datapool.back() + 1, batchSize);
  addrManager->setGlobalDatapPoll(&Datapool);
  for (uint32_t k = 0; k < 2; k++) {
    addrManager->MallocMem("PADDOutput(" + S(k) + ")", currentLevel);
    const auto a = k == 0 ? ctx->getC0Addr() : ctx->getC1Addr(), p = ptx->getC0Addr();
    std::vector<INSGROUP> g;
    for (uint32_t l = 0; l < currentLevel; l++)
      g.push_back(insgener->GenEWE(l, labelName + "_PADD_Level(" + S(l) + ")_k(" + S(k) + ")", nullptr, nullptr, nullptr, nullptr, a[l],
                                   0, p[l], 0, addrManager->getAddr("PADDOutput(" + S(k) + ")")[l], k == 0 ? EWE_ADD : EWE_COPY, l));
    driver->dispatchInstructions("PADD_Key(" + S(k) + ")", g);
    namedOutputs[k == 0 ? "out.c0" : "out.c1"] = addrManager->getAddr("PADDOutput(" + S(k) + ")");
  }
  finishConstruction();
}

// =====================================================================================================
// continuous execution
// =====================================================================================================
static OperationBase *makeOp(const std::string &o, uint32_t maxLevel, uint32_t level, uint32_t alpha, Config *cfg, Arch *arch) {
  if (o == "hmult") return new HMULT("test_hmult", maxLevel, level, alpha, cfg, arch);
  if (o == "hrotate") return new HROTATE("test_hrotate", maxLevel, level, alpha, cfg, arch);
  if (o == "hadd") return new HADD("test_hadd", maxLevel, level, alpha, cfg, arch);
  if (o == "pmult") return new PMULT("test_pmult", maxLevel, level, alpha, cfg, arch);
  if (o == "padd") return new PADD("test_ADD", maxLevel, level, alpha, cfg, arch);
  throw std::runtime_error("Error operation requirement, please double confirm!");
}

OpChain::OpChain(const std::string &cfgPath, const std::string &opList, uint32_t maxLevel, uint32_t curLevel, uint32_t alpha,
                 const std::map<std::string, uint32_t> &overrides) {
  std::stringstream ss(opList);
  std::string name;
  uint32_t level = curLevel;
  try {
    while (std::getline(ss, name, ',')) {
      if (name.empty()) continue;
      if (level == 0) throw std::runtime_error("chain: no limbs left for " + name);
      Config *cfg = new Config(cfgPath);
      cfgs.push_back(cfg);
      cfg->getValue("N");
      for (auto &kv : overrides) cfg->setValue(kv.first, kv.second);
      // every op draws its own second operand; the first op's first operand is the chain input
      cfg->setValue("seed", cfg->getValueOr("seed", 0x484F4D55u) + 31u * (uint32_t)ops.size());
      Arch *arch = new Arch(cfg);
      archs.push_back(arch);
      OperationBase *op = makeOp(name, maxLevel, level, alpha, cfg, arch);
      if (!ops.empty()) op->bindInput("ct1", ops.back());
      ops.push_back(op);
      level = op->outputLevel();
    }
    if (ops.empty()) throw std::runtime_error("chain: empty op list");
  } catch (...) {
    for (auto *o : ops) delete o;
    for (auto *a : archs) delete a;
    for (auto *c : cfgs) delete c;
    throw;
  }
}
OpChain::~OpChain() {
  for (auto *o : ops) delete o;
  for (auto *a : archs) delete a;
  for (auto *c : cfgs) delete c;
}
void OpChain::prepare() { for (auto *o : ops) o->prepare(); }
void OpChain::run() { for (auto *a : archs) a->run(); }
void OpChain::sync() { for (auto *a : archs) a->sync(); }
double OpChain::execute(uint32_t iters) {
  prepare();
  run();
  sync();
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t i = 0; i < iters; ++i) run();
  sync();
  return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / (iters ? iters : 1);
}
bool OpChain::simulate() {
  if (!archs.empty() && archs[0]->backend() == Arch::BACKEND_SIM) {  // the reference cannot chain: every op is simulated on its own
    unsigned long long cycles = 0;
    for (size_t i = 0; i < ops.size(); ++i) { ops[i]->simulate(); cycles += archs[i]->getCycle(); }
    std::cout << "\nChain of " << ops.size() << " operations, simulated one by one: " << cycles << " cycles in total\n";
    return true;
  }
  prepare();
  for (auto *o : ops) o->simulate();
  const double ns = execute(20);
  std::cout << "\nChain of " << ops.size() << " operations, ciphertext resident in HBM: " << (unsigned long long)ns << " ns per pass\n";
  return true;
}
