// Addr.h — named bump allocator in units of 256-coefficient lines, limb stride = batchSize lines; same
// names, same `Malloc <name> from <first> to <last>` stdout and same error behaviour as the reference's
// include/Addr.h:18-78.  In this build the lines are real: Arch maps line a to HBM (see Arch.h).
#ifndef HOMULATOR_ADDR_H
#define HOMULATOR_ADDR_H
#include "Basic.h"

class AddrManage {
private:
  AddrType addr, origin;
  uint32_t BatchSize;
  std::map<std::string, std::vector<AddrType>> dataMap;
  std::vector<std::string> order;
  std::vector<AddrType> *DataPool = nullptr;

public:
  AddrManage(AddrType PreAddr, uint32_t BSCount) : addr(PreAddr), origin(PreAddr), BatchSize(BSCount) {}
  AddrManage(const AddrManage &) = delete;
  AddrManage &operator=(const AddrManage &) = delete;

  void setGlobalDatapPoll(std::vector<AddrType> *pool) { DataPool = pool; }

  void MallocMem(const std::string &name, uint32_t polyCount) {
    if (dataMap.count(name)) {
      std::cerr << "Error: Name already exists in dataMap." << std::endl;
      return;
    }
    std::vector<AddrType> temp;
    for (uint32_t p = 0; p < polyCount; ++p) {
      temp.push_back(addr);
      if (DataPool) DataPool->push_back(addr);
      addr += BatchSize;
    }
    dataMap[name] = temp;
    order.push_back(name);
    std::cout << "Malloc " << name << " from " << temp[0] << " to " << temp.back() << std::endl;
  }
  // upstream allocates the base-conversion tables with this (Addr.h:50-66); each entry still advances the
  // bump pointer by one limb stride (updateBatchAddr, Addr.h:26), so every address stays congruent to the
  // origin modulo the stride.  Kept so the address plan and the stdout match.
  void MallocMemOneBatch(const std::string &name, uint32_t BatchCount) {
    if (dataMap.count(name)) {
      std::cerr << "Error: Name already exists in dataMap." << std::endl;
      return;
    }
    std::vector<AddrType> temp;
    for (uint32_t p = 0; p < BatchCount; ++p) {
      temp.push_back(addr);
      if (DataPool) DataPool->push_back(addr);
      addr += BatchSize;
    }
    dataMap[name] = temp;
    order.push_back(name);
    std::cout << "Malloc " << name << " from " << temp[0] << " to " << temp.back() << std::endl;
  }

  std::vector<AddrType> getAddr(const std::string &name) const {
    auto it = dataMap.find(name);
    if (it != dataMap.end()) return it->second;
    std::cout << name << std::endl;
    throw std::runtime_error("Cannot find this data key, please confirm!\n");
  }
  bool has(const std::string &name) const { return dataMap.count(name) != 0; }
  AddrType getLatestAddr() const { return addr; }
  AddrType getOrigin() const { return origin; }
  uint32_t getBatchSize() const { return BatchSize; }
  const std::vector<std::string> &names() const { return order; }
};
#endif
