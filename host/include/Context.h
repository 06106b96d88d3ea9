// Context.h — ciphertext / plaintext address holders with the reference's address plan
// (include/Context.h:56-73, 124-136): c0 and c1 limbs interleaved at a stride of BS lines, the next object
// starting at `Datapool.back() + 1`.  That plan makes consecutive objects overlap by BS-1 lines upstream
// (the addresses are only tokens there).  The tokens are kept bit for bit, but
// Arch gives every limb START address its own N words of HBM, so nothing aliases in this build.
#ifndef HOMULATOR_CONTEXT_H
#define HOMULATOR_CONTEXT_H
#include "Basic.h"

class Polynominal {
private:
  AddrType addressStart = 0, addressEnd = 0;
  uint32_t length;

public:
  explicit Polynominal(uint32_t size) : length(size) {}
  void setAddr(AddrType start, uint32_t batchCount) { addressStart = start; addressEnd = start + batchCount; }
  AddrType getAddressStart() const { return addressStart; }
  AddrType getAddressEnd() const { return addressEnd; }
  uint32_t size() const { return length; }
};

class Ciphertext {
private:
  uint32_t CurrentLevel;
  std::vector<Polynominal> c0, c1;

public:
  Ciphertext(uint32_t level, uint32_t N, std::vector<AddrType> &Datapool, uint32_t BS) : CurrentLevel(level) {
    AddrType AddressStart = Datapool.back() + 1;
    for (uint32_t l = 0; l < CurrentLevel; l++) {
      c0.emplace_back(N);
      c0[l].setAddr(AddressStart, BS);
      Datapool.push_back(AddressStart + BS);
      c1.emplace_back(N);
      c1[l].setAddr(AddressStart + BS, BS);
      Datapool.push_back(AddressStart + BS);
      AddressStart += 2 * BS;
    }
  }
  uint32_t level() const { return CurrentLevel; }
  std::vector<AddrType> getC0Addr() const { std::vector<AddrType> t; for (auto &p : c0) t.push_back(p.getAddressStart()); return t; }
  std::vector<AddrType> getC1Addr() const { std::vector<AddrType> t; for (auto &p : c1) t.push_back(p.getAddressStart()); return t; }
};

class Plaintext {
private:
  uint32_t CurrentLevel;
  std::vector<Polynominal> c0;

public:
  Plaintext(uint32_t level, uint32_t N, std::vector<AddrType> &Datapool, uint32_t BS) : CurrentLevel(level) {
    AddrType AddressStart = Datapool.back() + 1;
    for (uint32_t l = 0; l < CurrentLevel; l++) {
      c0.emplace_back(N);
      c0[l].setAddr(AddressStart, BS);
      Datapool.push_back(AddressStart + BS);
      AddressStart += BS;
    }
  }
  std::vector<AddrType> getC0Addr() const { std::vector<AddrType> t; for (auto &p : c0) t.push_back(p.getAddressStart()); return t; }
};
#endif
