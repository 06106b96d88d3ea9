// arch_shim.cpp — INTEGRATION.md §B made real (TEST INFRASTRUCTURE): an upstream-shaped `Arch` whose execution side is nothing
// but calls into include/homulator_hip.h, driven the way upstream's Driver drives its Arch (include/Arch.h:234-237 issueIns,
// src/Arch.cpp:912-929 update, include/Arch.h:246-269 simulateComplete, :271 getCycle), for ONE hybrid key switch
// (KeySwitch::KeySwitch src/Operation.cpp:9-54: INTT -> per digit {scale, BConv, NTT} -> inner product -> ModDown {INTT, scale,
// BConv, NTT, sub}).  Nothing of host/ is linked: this is what a maintainer who keeps upstream's own Operation / InsGen / Driver
// would write.  Instruction is upstream's record (op kind, limb, operand / output line addresses) plus the four things the timing
// model never needed: modulus id, EWE opcode, per-limb constant, bases of a conversion.
//   usage: arch_shim <logN> <L> <ell> <alpha> <d.bin> <evk.bin> <out.bin>      (d: [ell][N], evk: [beta][2][ell+alpha][N], out: [2][ell][N])
#include <cstdio>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/homulator_hip.h"

typedef unsigned long long AddrType;
enum ins_ops { NTT, INTT, MULT, BCONV_STEP2 };
struct Instruction {
  ins_ops ops; uint32_t level_id; std::vector<AddrType> operandList; AddrType OutputOperand;
  uint32_t mod_id = 0; int opcode = 0; uint64_t constant = 0; std::vector<uint32_t> inMods;   // what upstream lacks
};
typedef std::vector<Instruction *> INSGROUP;

static void ck(hm_ctx *c, hm_status s, const char *what) { if (s) throw std::runtime_error(std::string(what) + ": " + hm_last_error(c)); }

class Arch {   // upstream's public surface (the part Driver and Operation::simulate use)
  hm_ctx *hip = nullptr;
  uint64_t *pool = nullptr;
  uint32_t N;
  struct Stage { ins_ops kind; int opcode; INSGROUP ins; };
  std::vector<Stage> queue;
  size_t next = 0;
  unsigned long long ns = 0, completed = 0;
public:
  Arch(uint32_t logN, uint32_t L, uint32_t K, uint32_t limbs) : N(1u << logN) {
    hm_params p = {logN, L, K, 0, nullptr, nullptr, nullptr};
    if (hm_create(&hip, &p)) throw std::runtime_error(std::string("hm_create: ") + hm_last_error(nullptr));
    void *d = nullptr;
    ck(hip, hm_malloc(hip, (size_t)limbs * N * 8, &d), "hm_malloc");
    pool = static_cast<uint64_t *>(d);
  }
  ~Arch() { hm_free(hip, pool); hm_destroy(hip); }
  hm_ctx *ctx() { return hip; }
  uint64_t *limb(AddrType a) { return pool + (size_t)a * N; }   // line address -> limb-poly (one upstream "limb" = N words here)
  // include/Arch.h:234: one group for a unit of a cluster.  Groups of one kind that arrive back to back form a stage (one launch).
  void issueIns(uint32_t, const std::string &, INSGROUP &insg) {
    if (insg.empty()) return;
    if (queue.empty() || queue.back().kind != insg[0]->ops || queue.back().opcode != insg[0]->opcode || insg[0]->ops == BCONV_STEP2 || closed) { queue.push_back(Stage{insg[0]->ops, insg[0]->opcode, {}}); closed = false; }
    queue.back().ins.insert(queue.back().ins.end(), insg.begin(), insg.end());
  }
  // :236: MAC port (h, w) of the BCONV array; upstream's Driver replicates every group to all ports (include/Driver.h:307-320)
  void issueIns(uint32_t c, uint32_t h, uint32_t w, INSGROUP &insg, bool) { if (h == 0 && w == 0) issueIns(c, "BCONV", insg); }
  void stageBoundary() { closed = true; }   // (upstream: the next dispatchInstructions call)
  bool simulateComplete() { return next >= queue.size(); }
  unsigned long long getCycle() { return ns; }             // device nanoseconds
  unsigned long long getcompletedIns() { return completed; }
  void update() {                                            // src/Arch.cpp:912-929 advanced one cycle; here: the next stage runs
    if (simulateComplete()) return;
    const Stage &s = queue[next++];
    std::vector<uint32_t> a, b, c, d, out, mods;
    std::vector<uint64_t> k;
    for (Instruction *i : s.ins) {   // operand slots as upstream's GenEWE has them: (op1 x op2) + (op3 x op4), 0 = the fake operand
      a.push_back((uint32_t)i->operandList[0]); b.push_back(i->operandList.size() > 1 ? (uint32_t)i->operandList[1] : 0);
      c.push_back(i->operandList.size() > 2 ? (uint32_t)i->operandList[2] : 0); d.push_back(i->operandList.size() > 3 ? (uint32_t)i->operandList[3] : 0);
      out.push_back((uint32_t)i->OutputOperand);
      mods.push_back(i->mod_id); k.push_back(i->constant);
    }
    const uint32_t n = (uint32_t)s.ins.size();
    ck(hip, hm_timer_start(hip), "hm_timer_start");
    switch (s.kind) {
    case NTT:  ck(hip, hm_ntt(hip, pool, a.data(), pool, out.data(), mods.data(), n, 0, nullptr), "hm_ntt"); break;
    case INTT: ck(hip, hm_ntt(hip, pool, a.data(), pool, out.data(), mods.data(), n, 1, nullptr), "hm_ntt(inverse)"); break;
    case MULT: {
      ck(hip, hm_ewe(hip, s.opcode, pool, a.data(), pool, b.data(), pool, c.data(), pool, d.data(), pool, out.data(), mods.data(), n, k.data()), "hm_ewe");
      break;
    }
    case BCONV_STEP2: {   // one conversion: the group's instructions share the input limbs, one output limb each
      const Instruction *f = s.ins[0];
      std::vector<uint32_t> in(f->operandList.begin(), f->operandList.end());
      ck(hip, hm_bconv(hip, pool, in.data(), f->inMods.data(), (uint32_t)in.size(), pool, out.data(), mods.data(), n), "hm_bconv");
      break;
    }
    }
    uint64_t t = 0;
    ck(hip, hm_timer_stop(hip, &t), "hm_timer_stop");
    ns += t; completed += n;
  }
private:
  bool closed = false;
};

static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((unsigned __int128)a * b % q); }
static uint64_t powmod(uint64_t a, uint64_t e, uint64_t q) { uint64_t r = 1; for (; e; e >>= 1, a = mulmod(a, a, q)) if (e & 1) r = mulmod(r, a, q); return r; }

int main(int argc, char **argv) {
  if (argc < 8) { fprintf(stderr, "usage: %s logN L ell alpha d.bin evk.bin out.bin\n", argv[0]); return 2; }
  const uint32_t logN = atoi(argv[1]), L = atoi(argv[2]), ell = atoi(argv[3]), K = atoi(argv[4]), N = 1u << logN, E = ell + K, beta = (ell + K - 1) / K;
  try {
    // buffer plan in limb-polys (upstream: AddrManage::MallocMem, include/Addr.h:29-48)
    AddrType top = 0;
    auto alloc = [&](uint32_t n) { AddrType a = top; top += n; return a; };
    const AddrType D = alloc(ell), EVK = alloc(beta * 2 * E), INTTo = alloc(ell), DEC = alloc(ell), BC = alloc(beta * E), EXT = alloc(beta * E),
                   IP = alloc(2 * E), TMP = alloc(2 * E), MI = alloc(2 * K), MS = alloc(2 * K), MB = alloc(2 * ell), MN = alloc(2 * ell), KS = alloc(2 * ell);
    Arch arch(logN, L, K, (uint32_t)top);
    hm_ctx *ctx = arch.ctx();
    auto load = [&](const char *path, AddrType at, size_t limbs) {
      FILE *f = fopen(path, "rb"); if (!f) throw std::runtime_error(std::string("cannot open ") + path);
      std::vector<uint64_t> h(limbs * N);
      if (fread(h.data(), 8, h.size(), f) != h.size()) throw std::runtime_error("short read");
      fclose(f);
      ck(ctx, hm_memcpy_h2d(ctx, arch.limb(at), h.data(), h.size() * 8), "hm_memcpy_h2d");
    };
    load(argv[5], D, ell);
    load(argv[6], EVK, (size_t)beta * 2 * E);
    std::vector<uint32_t> ext(E);
    for (uint32_t t = 0; t < E; ++t) ext[t] = t < ell ? t : L + (t - ell);
    std::vector<uint64_t> q(E);
    for (uint32_t t = 0; t < E; ++t) ck(ctx, hm_get_modulus(ctx, ext[t], &q[t]), "hm_get_modulus");
    std::vector<Instruction *> all;
    auto ins = [&](ins_ops op, uint32_t level, std::vector<AddrType> in, AddrType out, uint32_t mod, int opcode = 0, uint64_t k = 0) {
      Instruction *i = new Instruction{op, level, std::move(in), out}; i->mod_id = mod; i->opcode = opcode; i->constant = k; all.push_back(i); return i; };
    auto dispatch = [&](std::vector<INSGROUP> &map, const char *unit) {   // Driver::dispatchInstructions + IssueInsFromDramToChip, one stage
      for (INSGROUP &g : map) { if (g[0]->ops == BCONV_STEP2) arch.issueIns(0, 0, 0, g, false); else arch.issueIns(0, unit, g); }
      arch.stageBoundary();
    };
    // ModUp_INTT (src/Operation.cpp:63-102)
    { std::vector<INSGROUP> m; for (uint32_t l = 0; l < ell; ++l) m.push_back({ins(INTT, l, {D + l}, INTTo + l, l)}); dispatch(m, "NTT"); }
    for (uint32_t j = 0; j < beta; ++j) {
      const uint32_t lo = j * K, hi = std::min(ell, lo + K), dj = hi - lo;
      std::vector<uint32_t> dmods; for (uint32_t l = lo; l < hi; ++l) dmods.push_back(l);
      std::vector<uint64_t> qh(dj);
      ck(ctx, hm_bconv_consts(ctx, dmods.data(), dj, nullptr, 0, qh.data(), nullptr), "hm_bconv_consts");
      // ModUp_DecompOut<j>) (:104-135): x q_hat^-1
      { std::vector<INSGROUP> m; for (uint32_t l = lo; l < hi; ++l) m.push_back({ins(MULT, l, {INTTo + l}, DEC + l, l, HM_OP_MUL_CONST, qh[l - lo])}); dispatch(m, "EWE"); }
      // ModUp_BCONV_(j) (:137-188): every limb of the extended basis outside the digit
      { INSGROUP g; std::vector<AddrType> in; for (uint32_t l = lo; l < hi; ++l) in.push_back(DEC + l);
        for (uint32_t t = 0; t < E; ++t) if (t < lo || t >= hi) { Instruction *i = ins(BCONV_STEP2, t, in, BC + j * E + t, ext[t]); i->inMods = dmods; g.push_back(i); }
        std::vector<INSGROUP> m = {g}; dispatch(m, "BCONV"); }
      // ModUp_NTT_(j) (:190-292): the converted limbs; the digit's own limbs are the input itself (evaluation form)
      { std::vector<INSGROUP> m; for (uint32_t t = 0; t < E; ++t) if (t < lo || t >= hi) m.push_back({ins(NTT, t, {BC + j * E + t}, EXT + j * E + t, ext[t])}); dispatch(m, "NTT"); }
    }
    auto extAddr = [&](uint32_t j, uint32_t t) { const uint32_t lo = j * K, hi = std::min(ell, lo + K); return (t >= lo && t < hi) ? D + t : EXT + j * E + t; };
    auto key = [&](uint32_t j, uint32_t k, uint32_t t) { return EVK + ((AddrType)j * 2 + k) * E + t; };
    // InnerProOut_(.)_Key<k> (:294-414): one product, then MAC groups
    for (uint32_t k = 0; k < 2; ++k) {
      std::vector<INSGROUP> m;
      if (beta == 1) { for (uint32_t t = 0; t < E; ++t) m.push_back({ins(MULT, t, {extAddr(0, t), key(0, k, t)}, IP + k * E + t, ext[t], HM_OP_MUL)}); dispatch(m, "EWE"); continue; }
      for (uint32_t be = 0; be + 1 < beta; ++be) {
        m.clear();
        for (uint32_t t = 0; t < E; ++t) {
          const AddrType out = (be + 2 == beta) ? IP + k * E + t : TMP + k * E + t;
          if (be == 0) { Instruction *i = ins(MULT, t, {extAddr(0, t), key(0, k, t), extAddr(1, t)}, out, ext[t], HM_OP_MAC2); i->operandList.push_back(key(1, k, t)); m.push_back({i}); }
          else m.push_back({ins(MULT, t, {extAddr(be + 1, t), key(be + 1, k, t), TMP + k * E + t}, out, ext[t], HM_OP_MAC_ADD)});   // + the previous group's sum
        }
        dispatch(m, "EWE");
      }
    }
    // ModDown (:417-590)
    std::vector<uint32_t> pmods; for (uint32_t p = 0; p < K; ++p) pmods.push_back(L + p);
    std::vector<uint64_t> ph(K);
    ck(ctx, hm_bconv_consts(ctx, pmods.data(), K, nullptr, 0, ph.data(), nullptr), "hm_bconv_consts");
    { std::vector<INSGROUP> m; for (uint32_t k = 0; k < 2; ++k) for (uint32_t p = 0; p < K; ++p) m.push_back({ins(INTT, p, {IP + k * E + ell + p}, MI + k * K + p, L + p)}); dispatch(m, "NTT"); }
    { std::vector<INSGROUP> m; for (uint32_t k = 0; k < 2; ++k) for (uint32_t p = 0; p < K; ++p) m.push_back({ins(MULT, p, {MI + k * K + p}, MS + k * K + p, L + p, HM_OP_MUL_CONST, ph[p])}); dispatch(m, "EWE"); }
    for (uint32_t k = 0; k < 2; ++k) {
      INSGROUP g; std::vector<AddrType> in; for (uint32_t p = 0; p < K; ++p) in.push_back(MS + k * K + p);
      for (uint32_t l = 0; l < ell; ++l) { Instruction *i = ins(BCONV_STEP2, l, in, MB + k * ell + l, l); i->inMods = pmods; g.push_back(i); }
      std::vector<INSGROUP> m = {g}; dispatch(m, "BCONV");
    }
    { std::vector<INSGROUP> m; for (uint32_t k = 0; k < 2; ++k) for (uint32_t l = 0; l < ell; ++l) m.push_back({ins(NTT, l, {MB + k * ell + l}, MN + k * ell + l, l)}); dispatch(m, "NTT"); }
    { std::vector<INSGROUP> m;
      for (uint32_t k = 0; k < 2; ++k) for (uint32_t l = 0; l < ell; ++l) {
        uint64_t P = 1; for (uint32_t p = 0; p < K; ++p) P = mulmod(P, q[ell + p] % q[l], q[l]);
        m.push_back({ins(MULT, l, {IP + k * E + l, 0, MN + k * ell + l}, KS + k * ell + l, l, HM_OP_SUB_SCALE, powmod(P, q[l] - 2, q[l]))});
      }
      dispatch(m, "EWE"); }
    // HMULT::simulate's loop (src/Operation.cpp:1046): one update per "cycle"
    while (!arch.simulateComplete()) arch.update();
    ck(ctx, hm_sync(ctx), "hm_sync");
    std::vector<uint64_t> h((size_t)2 * ell * N);
    ck(ctx, hm_memcpy_d2h(ctx, h.data(), arch.limb(KS), h.size() * 8), "hm_memcpy_d2h");
    FILE *f = fopen(argv[7], "wb"); fwrite(h.data(), 8, h.size(), f); fclose(f);
    printf("key switch through the C ABI: %llu limb records in %llu ns of device time\n", arch.getcompletedIns(), arch.getCycle());
    for (Instruction *i : all) delete i;
  } catch (const std::exception &e) { fprintf(stderr, "arch_shim: %s\n", e.what()); return 1; }
  return 0;
}
