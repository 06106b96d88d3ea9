// SimProgram.cpp — the literal instruction stream of the reference for one operation, as input of the `sim` backend.
//
// The functional backends execute the mathematically correct stage graph at limb granularity (host/src/Operation.cpp).
// The reference's cycle counts belong to a different program: its own 256-coefficient instructions with its own operand
// wiring, quirks included.  This file re-derives that program from the address plan (the same
// AddrManage names and lines the Operation constructors allocated) and the reference's generator and placement rules:
//   generators   src/InsGen.cpp:17-140 (GenNTT / GenAUTO / GenEWE), :263-313 (GenBCONV), use counts :155-185
//   op wiring    src/Operation.cpp:63-590 (KeySwitch), :624-739 (TensorCompute), :766-911 (Rescale), :913-1023 (HMULT),
//                :1114-1176 (HADD), :1271-1358 (HROTATE), :1453-1523 (PMULT), :1618-1680 (PADD)
//   placement    include/Driver.h:71-105 (by op kind), :155-246 (NTT / AUTO: level % cluster; EWE: round robin per
//                instruction; BCONV: tiles of bconv_num_high batches x bconv_num_width levels, one cluster per row of tiles),
//                :107-153 (first touch of an address that no instruction produces queues a DRAM fetch on that cluster)
// Instruction names, dependency pointers, operand format bits and program counters do not influence timing and are not
// generated.
#include "SimProgram.h"

namespace {
typedef std::vector<AddrType> Addrs;
typedef std::vector<std::vector<uint32_t>> StageIds;            // [level][batch] -> instruction id
typedef std::vector<std::vector<SimProgram::Group>> StageGroups;  // [level][batch] -> base-conversion group

class Builder {
public:
  Builder(Config *cfg, const AddrManage &am) : am_(am) {
    p.cluster = cfg->getValue("cluster");
    B = cfg->getValue("N") / cfg->getValue("batchSize");
    bh = cfg->getValue("bconv_num_high");
    bw = cfg->getValue("bconv_num_width");
    if (cfg->getValue("entryCount") != 1)
      throw std::runtime_error("sim backend: entryCount must be 1 (upstream's Driver rejects partial DRAM entries, include/Driver.h:140-149)");
    p.ewe.resize(p.cluster); p.ntt.resize(p.cluster); p.aut.resize(p.cluster); p.bconv.resize(p.cluster); p.dram.resize(p.cluster);
  }
  Addrs buf(const std::string &name) const { return am_.getAddr(name); }

  // ---- generators: one stage = [level][batch]
  std::vector<uint32_t> genLine(AddrType in, AddrType out) {  // GenNTT / GenAUTO: one single-operand instruction per batch line
    std::vector<uint32_t> ids;
    for (uint32_t b = 0; b < B; b++) {
      ids.push_back(emit({in + b, 0, 0, 0}, out + b, 1));
      use(in + b);
      produce(out + b);
    }
    return ids;
  }
  std::vector<uint32_t> genEWE(AddrType o1, AddrType o2, AddrType o3, AddrType o4, AddrType out) {
    std::vector<uint32_t> ids;
    for (uint32_t idx = 0; idx < 2 * (B / 2); idx++) {
      ids.push_back(emit({o1 + idx, o2 + idx, o3 + idx, o4 + idx}, out + idx, 4));
      AddrType s[4] = {o1, o2, o3, o4};  // every DISTINCT start operand is one read of its line (the fake operand 0 included)
      std::sort(s, s + 4);
      for (int k = 0; k < 4; k++)
        if (k == 0 || s[k] != s[k - 1]) use(s[k] + idx);
      produce(out + idx);
    }
    return ids;
  }
  std::vector<SimProgram::Group> genBCONV(uint32_t inLevel, const Addrs &in, AddrType table, AddrType out) {
    std::vector<SimProgram::Group> gs;
    for (uint32_t b = 0; b < B; b++) {
      SimProgram::Group g{(uint32_t)p.ins.size(), inLevel};
      for (uint32_t il = 0; il < inLevel; il++) {
        emit({in.at(il) + b, table, 0, 0}, out + b, 2);
        use(in[il] + b);
        use(table);
      }
      produce(out + b);
      gs.push_back(g);
    }
    return gs;
  }

  // ---- placement
  void dispatchLine(const StageIds &st, std::vector<std::vector<uint32_t>> &queue) {  // NTT / INTT / AUTO
    for (uint32_t l = 0; l < st.size(); l++) {
      const uint32_t c = l % p.cluster;
      for (uint32_t id : st[l]) { queue[c].push_back(id); touch(id, c); }
    }
  }
  void dispatchNTT(const StageIds &st) { need(st); dispatchLine(st, p.ntt); }
  void dispatchAUTO(const StageIds &st) { need(st); dispatchLine(st, p.aut); }
  void dispatchEWE(const StageIds &st) {
    need(st);
    for (const auto &lv : st)
      for (uint32_t id : lv) {
        p.ewe[eweNext].push_back(id);
        touch(id, eweNext);
        eweNext = (eweNext + 1) % p.cluster;
      }
  }
  void dispatchBCONV(const StageGroups &st) {
    if (st.empty() || st[0].empty()) throw std::runtime_error("Empty instruction map provided.");
    const uint32_t wTiles = (uint32_t)((st.size() + bw - 1) / bw), hTiles = (uint32_t)((st[0].size() + bh - 1) / bh);
    for (uint32_t ht = 0; ht < hTiles; ht++) {
      for (uint32_t wt = 0; wt < wTiles; wt++)
        for (uint32_t ho = 0; ho < bh; ho++) {
          const uint32_t b = ht * bh + ho;
          if (b >= st[0].size()) continue;
          for (uint32_t wo = 0; wo < bw; wo++) {
            const uint32_t l = wt * bw + wo;
            if (l >= st.size()) continue;
            const SimProgram::Group g = st[l][b];
            p.bconv[bconvNext].push_back(g);
            for (uint32_t k = 0; k < g.count; k++) touch(g.first + k, bconvNext);
          }
        }
      bconvNext = (bconvNext + 1) % p.cluster;
    }
  }
  SimProgram take() {
    p.maxAddr = top;
    grow(top + 1);
    return std::move(p);
  }
  uint32_t batchCount() const { return B; }

private:
  SimProgram p;
  const AddrManage &am_;
  uint32_t B = 0, bh = 0, bw = 0, eweNext = 0, bconvNext = 0;
  AddrType top = 0;
  std::vector<uint8_t> touched;  // Driver::dispathTimes

  static void need(const StageIds &st) {
    if (st.empty() || st[0].empty()) throw std::runtime_error("Empty instruction map provided.");
  }
  void grow(AddrType a) {
    if (a > top) top = a;
    if (a >= p.uses.size()) {
      const size_t n = (size_t)a + 4096;
      p.uses.resize(n, 0); p.usesKey.resize(n, 0); p.pendingOut.resize(n, 0); touched.resize(n, 0);
    }
  }
  void use(AddrType a) { grow(a); p.uses[a]++; p.usesKey[a] = 1; }
  void produce(AddrType a) { grow(a); p.pendingOut[a] = 1; }
  uint32_t emit(std::initializer_list<AddrType> ops, AddrType out, uint8_t nIn) {
    SimIns i{};
    int k = 0;
    for (AddrType a : ops) i.op[k++] = a;
    i.out = out;
    i.nIn = nIn;
    for (int q = 0; q < nIn; q++) grow(i.op[q]);
    grow(out);
    p.ins.push_back(i);
    return (uint32_t)p.ins.size() - 1;
  }
  // include/Driver.h:107-153: the first instruction that names an address decides; an address that is some instruction's
  // output at that moment is never fetched
  void touch(uint32_t id, uint32_t c) {
    const SimIns &i = p.ins[id];
    for (int k = 0; k < i.nIn; k++) {
      const AddrType a = i.op[k];
      if (touched[a]) continue;
      touched[a] = 1;
      if (!p.pendingOut[a]) p.dram[c].push_back(a);
    }
  }
};

std::string S(uint32_t v) { return std::to_string(v); }

// src/Operation.cpp:9-590.  `pre`: the limbs the key switch is fed (TensorD0Out for hmult, AUTOOutput(0) for hrotate: the
// reference's wiring, Appendix C items 1-2).  Generation is complete before the first dispatch, as upstream.
void keySwitch(Builder &b, uint32_t level, uint32_t alpha, const Addrs &pre) {
  const uint32_t beta = (level + alpha - 1) / alpha, E = level + alpha;
  std::vector<std::pair<char, StageIds>> order;  // 'N' / 'E' stages in KeySwitchInsMapName order; 'B' marks a conversion
  std::vector<StageGroups> conv;
  auto pushIds = [&](char kind, StageIds s) { order.emplace_back(kind, std::move(s)); };
  auto pushConv = [&](StageGroups s) { order.emplace_back('B', StageIds()); conv.push_back(std::move(s)); };

  {  // ModUp_INTT :63-102
    StageIds s;
    const Addrs out = b.buf("ModUpINTTOut");
    for (uint32_t l = 0; l < level; l++) s.push_back(b.genLine(pre.at(l), out[l]));
    pushIds('N', s);
  }
  const Addrs inttOut = b.buf("ModUpINTTOut"), decomp = b.buf("ModUpDecompOut");
  const AddrType decompOffset = b.buf("ModUpDecompOffset")[0];
  for (uint32_t be = 0; be < beta; be++) {
    const uint32_t d = std::min(alpha, level - be * alpha);
    {  // ModUp_DecompOut<be>) :104-135
      StageIds s;
      for (uint32_t a = 0; a < d; a++) s.push_back(b.genEWE(inttOut[be * alpha + a], decompOffset, 0, 0, decomp[be * alpha + a]));
      pushIds('E', s);
    }
    const Addrs convOut = b.buf("BConvOut_(" + S(be) + ")");
    {  // ModUp_BCONV_(be) :137-188 — the inputs are ModUpDecompOut[0 .. d), not the digit's own limbs (Appendix C item 4)
      StageGroups s;
      const AddrType table = b.buf("BConvMap_(" + S(be) + ")")[0];
      for (uint32_t ol = d; ol < E; ol++) s.push_back(b.genBCONV(d, decomp, table, convOut.at(ol - d)));
      pushConv(s);
    }
    {  // ModUp_NTT_(be) :190-292 — BConvOut is indexed with l, and past its end the input keeps the previous level's value
      StageIds s;
      const Addrs out = b.buf("NTTOut_beta(" + S(be) + ")");
      AddrType in = 0;
      for (uint32_t l = 0; l < E; l++) {
        if (l < d) in = decomp[l];
        else if (l < convOut.size()) in = convOut[l];
        s.push_back(b.genLine(in, out[l]));
      }
      pushIds('N', s);
    }
  }
  for (uint32_t k = 0; k < 2; k++) {  // InnerProOut_(be)_Key<k> :294-414
    auto key = [&](uint32_t j) { return b.buf("IP_Key" + S(k) + "_" + S(j)); };
    auto ext = [&](uint32_t j) { return b.buf("NTTOut_beta(" + S(j) + ")"); };
    if (beta == 1) {
      StageIds s;
      const Addrs x = ext(0), y = key(0), out = b.buf("InnerProduceOut_Key" + S(k));
      for (uint32_t ml = 0; ml < E; ml++) s.push_back(b.genEWE(x[ml], y[ml], 0, 0, out[ml]));
      pushIds('E', s);
    } else {
      for (uint32_t be = 0; be + 1 < beta; be++) {  // the last group writes ..._temp(beta-2); ..._Key<k> is never produced (item 3)
        StageIds s;
        const uint32_t j = be == 0 ? 0 : be + 1;
        const Addrs x = ext(j), y = key(j), out = b.buf("InnerProduceOut_temp(" + S(be) + ")_Key" + S(k));
        const Addrs x2 = be == 0 ? ext(1) : b.buf("InnerProduceOut_temp(" + S(be - 1) + ")_Key" + S(k));
        const Addrs y2 = be == 0 ? key(1) : Addrs(E, 0);
        for (uint32_t ml = 0; ml < E; ml++) s.push_back(b.genEWE(x[ml], y[ml], x2[ml], y2[ml], out[ml]));
        pushIds('E', s);
      }
    }
  }
  for (uint32_t k = 0; k < 2; k++) {  // ModDownINTTOut_Key(k) :417-445 — reads InnerProduceOut_Key<k>[0 .. alpha)
    StageIds s;
    const Addrs in = b.buf("InnerProduceOut_Key" + S(k)), out = b.buf("INTTOut_ModDown_Key(" + S(k) + ")");
    for (uint32_t l = 0; l < alpha; l++) s.push_back(b.genLine(in[l], out[l]));
    pushIds('N', s);
  }
  for (uint32_t k = 0; k < 2; k++) {  // ModDownBConvStep1_Key(k) :447-487
    StageIds s;
    const Addrs in = b.buf("INTTOut_ModDown_Key(" + S(k) + ")"), out = b.buf("ModDownBConvStep1_Key(" + S(k) + ")");
    const AddrType ref = b.buf("ModDownBConvStep1_Ref")[k];
    for (uint32_t l = 0; l < alpha; l++) s.push_back(b.genEWE(in[l], ref, 0, 0, out[l]));
    pushIds('E', s);
  }
  for (uint32_t k = 0; k < 2; k++) {  // ModDown_BCONV_Key(k) :489-519
    StageGroups s;
    const Addrs in = b.buf("ModDownBConvStep1_Key(" + S(k) + ")"), out = b.buf("ModdownBConvOut_Key" + S(k));
    const AddrType table = b.buf("ModdownBConvMap")[0];
    for (uint32_t ol = 0; ol < level; ol++) s.push_back(b.genBCONV(alpha, in, table, out[ol]));
    pushConv(s);
  }
  for (uint32_t k = 0; k < 2; k++) {  // ModDownNTTOut_Key(k) :521-546
    StageIds s;
    const Addrs in = b.buf("ModdownBConvOut_Key" + S(k)), out = b.buf("NTTOut_ModDown_Key(" + S(k) + ")");
    for (uint32_t l = 0; l < level; l++) s.push_back(b.genLine(in[l], out[l]));
    pushIds('N', s);
  }
  for (uint32_t k = 0; k < 2; k++) {  // KeySwitchFinalOutput_Key(k) :548-590
    StageIds s;
    const Addrs a = b.buf("NTTOut_ModDown_Key(" + S(k) + ")"), ip = b.buf("InnerProduceOut_Key" + S(k)),
                out = b.buf("KeySwitchFinalOutput_Key(" + S(k) + ")");
    for (uint32_t l = 0; l < level; l++) s.push_back(b.genEWE(a[l], ip[alpha + l], 0, 0, out[l]));
    pushIds('E', s);
  }
  size_t nextConv = 0;
  for (auto &st : order) {
    if (st.first == 'N') b.dispatchNTT(st.second);
    else if (st.first == 'E') b.dispatchEWE(st.second);
    else b.dispatchBCONV(conv[nextConv++]);
  }
}

// one EWE stage per ciphertext component, generated and dispatched like HADD / PMULT / PADD and the closing additions
StageIds eweStage(Builder &b, uint32_t level, const Addrs &o1, const Addrs &o2, const Addrs &o3, const Addrs &o4, const Addrs &out) {
  StageIds s;
  const Addrs zero(level, 0);
  const Addrs &a = o1.empty() ? zero : o1, &c = o2.empty() ? zero : o2, &d = o3.empty() ? zero : o3, &e = o4.empty() ? zero : o4;
  for (uint32_t l = 0; l < level; l++) s.push_back(b.genEWE(a[l], c[l], d[l], e[l], out[l]));
  return s;
}
}  // namespace

unsigned long long SimProgram::totalInstructions(uint32_t bconvPorts) const {
  unsigned long long t = 0;
  for (uint32_t c = 0; c < cluster; c++) {
    t += ewe[c].size() + ntt[c].size() + aut[c].size();
    for (const Group &g : bconv[c]) t += (unsigned long long)g.count * bconvPorts;
  }
  return t;
}

SimProgram buildSimProgram(const std::string &op, const std::string &label, uint32_t level, uint32_t alpha, Config *cfg,
                           const AddrManage &am, const std::map<std::string, std::vector<AddrType>> &in) {
  Builder b(cfg, am);
  const Addrs none;
  auto input = [&](const char *n) -> const Addrs & {
    auto it = in.find(n);
    if (it == in.end()) throw std::runtime_error(std::string("sim backend: missing input ") + n);
    return it->second;
  };
  if (op == "HMULT") {
    const Addrs &c00 = input("ct1.c0"), &c01 = input("ct1.c1"), &c10 = input("ct2.c0"), &c11 = input("ct2.c1");
    const StageIds d0 = eweStage(b, level, c00, c10, none, none, b.buf("TensorD0Out"));   // :624-661
    const StageIds d1 = eweStage(b, level, c00, c11, c01, c10, b.buf("TensorD1Out"));     // :663-700
    const StageIds d2 = eweStage(b, level, c01, c11, none, none, b.buf("TensorD2Out"));   // :702-739
    b.dispatchEWE(d0); b.dispatchEWE(d1); b.dispatchEWE(d2);
    keySwitch(b, level, alpha, b.buf("TensorD0Out"));                                     // :953-957
    StageIds add[2];
    for (uint32_t k = 0; k < 2; k++)                                                      // :975-1000: ks0 + D1, ks1 + D2
      add[k] = eweStage(b, level, b.buf("KeySwitchFinalOutput_Key(" + S(k) + ")"), none, b.buf(k == 0 ? "TensorD1Out" : "TensorD2Out"),
                        none, b.buf("HMULTHaddOutput(" + S(k) + ")"));
    b.dispatchEWE(add[0]); b.dispatchEWE(add[1]);
    for (uint32_t k = 0; k < 2; k++) {                                                    // Rescale :766-911: ONE transform pair
      const std::string base = label + "_" + S(k) + "_Rescale";
      const Addrs pre = b.buf("HMULTHaddOutput(" + S(k) + ")");
      const StageIds intt{b.genLine(pre.at(level - 1), b.buf(base + "_ResINTTOut")[0])};
      const StageIds ntt{b.genLine(b.buf(base + "_ResINTTOut")[0], b.buf(base + "_ResNTTOut")[0])};
      const Addrs res(level - 1, b.buf(base + "_ResNTTOut")[0]);
      const StageIds sub = eweStage(b, level - 1, res, pre, none, none, b.buf(base + "_Rescale_SubOut"));
      const StageIds mul = eweStage(b, level - 1, b.buf(base + "_Rescale_SubOut"), b.buf(base + "_Rescale_Mul_Offset"), none, none,
                                    b.buf(base + "_Rescale_MulOut"));
      b.dispatchNTT(intt); b.dispatchNTT(ntt); b.dispatchEWE(sub); b.dispatchEWE(mul);
    }
  } else if (op == "HROTATE") {
    StageIds a[2];
    for (uint32_t k = 0; k < 2; k++) {                                                    // :1302-1324
      const Addrs &src = input(k == 0 ? "ct1.c0" : "ct1.c1"), out = b.buf("AUTOOutput(" + S(k) + ")");
      for (uint32_t l = 0; l < level; l++) a[k].push_back(b.genLine(src[l], out[l]));
    }
    b.dispatchAUTO(a[0]); b.dispatchAUTO(a[1]);
    keySwitch(b, level, alpha, b.buf("AUTOOutput(0)"));                                   // :1326-1337
    b.dispatchEWE(eweStage(b, level, b.buf("KeySwitchFinalOutput_Key(1)"), none, b.buf("AUTOOutput(1)"), none, b.buf("HROTATEOutput(1)")));
  } else if (op == "HADD") {
    StageIds s[2];
    for (uint32_t k = 0; k < 2; k++)                                                      // :1146-1170
      s[k] = eweStage(b, level, input(k == 0 ? "ct1.c0" : "ct1.c1"), none, input(k == 0 ? "ct2.c0" : "ct2.c1"), none, b.buf("HADDOutput(" + S(k) + ")"));
    b.dispatchEWE(s[0]); b.dispatchEWE(s[1]);
  } else if (op == "PMULT") {
    StageIds s[2];
    for (uint32_t k = 0; k < 2; k++)                                                      // :1487-1518
      s[k] = eweStage(b, level, input(k == 0 ? "ct1.c0" : "ct1.c1"), input("pt"), none, none, b.buf("HMult" + S(k) + "Out"));
    b.dispatchEWE(s[0]); b.dispatchEWE(s[1]);
  } else if (op == "PADD") {
    StageIds s[2];
    for (uint32_t k = 0; k < 2; k++)                                                      // :1650-1674
      s[k] = eweStage(b, level, input(k == 0 ? "ct1.c0" : "ct1.c1"), none, input("pt"), none, b.buf("PADDOutput(" + S(k) + ")"));
    b.dispatchEWE(s[0]); b.dispatchEWE(s[1]);
  } else {
    throw std::runtime_error("sim backend: unknown operation " + op);
  }
  return b.take();
}
