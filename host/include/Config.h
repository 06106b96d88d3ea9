// Config.h — `key = value` configuration, same file format, echo and error behaviour as the
// reference's include/Config.h:1-26 + src/Config.cpp:4-52 (the shipped .cfg files load unchanged).
#ifndef HOMULATOR_CONFIG_H
#define HOMULATOR_CONFIG_H
#include "Basic.h"

class Config {
private:
  std::map<std::string, uint32_t> configMap;

public:
  explicit Config(std::string file);

  uint32_t getValue(std::string key) {
    auto it = configMap.find(key);
    if (it == configMap.end()) throw std::runtime_error("Can not find this key!\n");
    return it->second;
  }
  // keys the reference's files do not have (backend, fuse, gpus, galois, seed, ...) are read with a default
  uint32_t getValueOr(const std::string &key, uint32_t dflt) const {
    auto it = configMap.find(key);
    return it == configMap.end() ? dflt : it->second;
  }
  bool hasKey(const std::string &key) const { return configMap.count(key) != 0; }
  void setValue(std::string name, uint32_t count) { configMap[name] = count; }
};
#endif
