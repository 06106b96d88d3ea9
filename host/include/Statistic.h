// Statistic.h — named counters printed as `key :\t value` (format of the reference's
// include/Staistics.h:30-37).  Values here are measured (nanoseconds, bytes, launches), 64-bit.
#ifndef HOMULATOR_STATISTIC_H
#define HOMULATOR_STATISTIC_H
#include "Basic.h"

class Statistic {
private:
  std::map<std::string, unsigned long long> statMap;

public:
  void increaseStat(const std::string &key, unsigned long long count = 1) { statMap[key] += count; }
  void setStat(const std::string &key, unsigned long long v) { statMap[key] = v; }
  unsigned long long getStat(const std::string &key) const {
    auto it = statMap.find(key);
    return it == statMap.end() ? 0 : it->second;
  }
  void showStat() const {
    std::cout << "Start outPut statistic informations:\n";
    std::cout << "=====================================\n";
    for (const auto &kv : statMap) std::cout << kv.first << " :\t" << kv.second << "\n";
  }
};
#endif
