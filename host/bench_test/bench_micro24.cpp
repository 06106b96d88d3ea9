// bench_micro24.cpp — the CLI, same argv contract, messages and exit codes as the reference's only main()
// (bench_test/bench_micro24.cpp:5-52):  <cfg> <op> <maxLevel> <curLevel> <alpha> [cluster]
#include "Arch.h"
#include "Basic.h"
#include "Operation.h"
#include "RcclRendezvous.h"

#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <fstream>
#include <thread>

extern "C" int hm_comm_unique_id(void *out128);

// [cluster] doubles as the GPU count: `torchrun --nproc-per-node 8 ./Homulator.run <cfg> hmult 45 35 15 8` (or
// any launcher that sets WORLD_SIZE / RANK / LOCAL_RANK) runs ONE op sharded over the ranks.  The 128-byte RCCL id goes
// from rank 0 to the others through a file whose name is unique to the run (RcclRendezvous.h).
static void rcclRendezvous(Arch *arch, const std::string &path) {
  char id[hrv::kIdBytes];
  if (arch->rank() == 0) {
    if (hm_comm_unique_id(id)) throw std::runtime_error("hm_comm_unique_id failed (is librccl.so available?)");
    hrv::publish(path, id);
  } else {
    const char *t = getenv("HOMULATOR_RCCL_ID_TIMEOUT_S");
    hrv::fetch(path, id, (t ? (unsigned)atoi(t) : 600u) * 1000u);
  }
  arch->commInitRccl(id);   // collective: when it returns, every rank has read the file
  if (arch->rank() == 0) hrv::removeStale(path);
}

int main(int argc, char *argv[]) {
  if (argc < 6) {
    std::cerr << "Usage: " << argv[0] << " <path>" << std::endl;
    return 1;
  }
  std::string path = argv[1];
  std::string ops = argv[2];
  Config *config = new Config(path);
  uint32_t maxlevel = std::atoi(argv[3]);
  uint32_t currentlevel = std::atoi(argv[4]);
  uint32_t alpha = std::atoi(argv[5]);
  if (argc > 6) {
    config->setValue("cluster", std::atoi(argv[6]));
    config->setValue("cluster_from_argv", 1);
  }
  // only the CLI infers its rank from a launcher's environment (WORLD_SIZE / RANK / LOCAL_RANK); library users say world / rank
  config->setValue("launcher_env", 1);
  const std::string idFile = hrv::idPath();
  if (const char *ws = getenv("WORLD_SIZE")) {   // rank 0 of a launcher run: a leftover of an aborted run with this name goes first
    const char *rk = getenv("RANK");
    if (atoi(ws) > 1 && (!rk || atoi(rk) == 0)) hrv::removeStale(idFile);
  }

  if (ops.find(',') != std::string::npos) {  // build extension: "hmult,hrotate,hadd" = a chain with the ciphertext resident in HBM
    try {
      OpChain chain(path, ops, maxlevel, currentlevel, alpha, argc > 6 ? std::map<std::string, uint32_t>{{"cluster", (uint32_t)std::atoi(argv[6])}} : std::map<std::string, uint32_t>{});
      chain.simulate();
    } catch (const std::exception &e) {
      std::cout << e.what() << "\n";
    }
    return 0;
  }
  Arch *arch = nullptr;
  try {
    arch = new Arch(config);
  } catch (const std::exception &e) {
    std::cerr << e.what() << std::endl;
    return 1;
  }
  if (ops == "hmult") {
    HMULT *hmult = new HMULT("test_hmult", maxlevel, currentlevel, alpha, config, arch);
    if (arch->world() > 1) rcclRendezvous(arch, idFile);
    hmult->simulate();
  } else if (ops == "hrotate") {
    HROTATE *hrotate = new HROTATE("test_hrotate", maxlevel, currentlevel, alpha, config, arch);
    if (arch->world() > 1) rcclRendezvous(arch, idFile);
    hrotate->simulate();
  } else if (ops == "hadd") {
    HADD *hadd = new HADD("test_hadd", maxlevel, currentlevel, alpha, config, arch);
    if (arch->world() > 1) rcclRendezvous(arch, idFile);
    hadd->simulate();
  } else if (ops == "pmult") {
    PMULT *pmult = new PMULT("test_pmult", maxlevel, currentlevel, alpha, config, arch);
    if (arch->world() > 1) rcclRendezvous(arch, idFile);
    pmult->simulate();
  } else if (ops == "padd") {
    PADD *padd = new PADD("test_ADD", maxlevel, currentlevel, alpha, config, arch);
    if (arch->world() > 1) rcclRendezvous(arch, idFile);
    padd->simulate();
  } else {
    std::cout << "Error operation requirement, please double confirm!\n";
  }
}
