#include "Config.h"

static std::string trim(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

Config::Config(std::string file) {
  std::ifstream in(file);
  if (!in.is_open()) {  // same behaviour as upstream: report and continue with an empty map
    std::cerr << "Error opening config file." << std::endl;
    return;
  }
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '#') continue;
    size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    std::string key = trim(line.substr(0, eq)), value = trim(line.substr(eq + 1));
    configMap[key] = static_cast<uint32_t>(std::stoi(value));
  }
  std::cout << "Configuration details are as follow:\n\n*****************************************\n\n";
  for (const auto &kv : configMap)
    std::cout << std::left << std::setw(20) << kv.first << " " << std::right << std::setw(20) << kv.second << "\n";
  std::cout << "\n*****************************************\n\n";
}
