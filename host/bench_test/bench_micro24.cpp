// bench_micro24.cpp — the CLI, same argv contract, messages and exit codes as the reference's only main()
// (bench_test/bench_micro24.cpp:5-52):  <cfg> <op> <maxLevel> <curLevel> <alpha> [cluster]
#include "Arch.h"
#include "Basic.h"
#include "Operation.h"

#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <fstream>
#include <thread>

extern "C" int hm_comm_unique_id(void *out128);

// [cluster] doubles as the GPU count: `torchrun --nproc-per-node 8 ./Homulator.run <cfg> hmult 45 35 15 8` (or
// any launcher that sets WORLD_SIZE / RANK / LOCAL_RANK) runs ONE op sharded over the ranks.  The 128-byte RCCL id goes
// from rank 0 to the others through a file (HOMULATOR_RCCL_ID_FILE, default /tmp/homulator_rccl_<MASTER_PORT>.id): the CLI
// has no other channel between its processes.  Files older than this process are leftovers of earlier runs and ignored.
static void rcclRendezvous(Arch *arch) {
  const char *envp = getenv("HOMULATOR_RCCL_ID_FILE"), *port = getenv("MASTER_PORT");
  const std::string path = envp ? envp : std::string("/tmp/homulator_rccl_") + (port ? port : "0") + ".id";
  const time_t started = time(nullptr) - 5;
  char id[128];
  if (arch->rank() == 0) {
    if (hm_comm_unique_id(id)) throw std::runtime_error("hm_comm_unique_id failed (is librccl.so available?)");
    const std::string tmp = path + ".tmp";
    { std::ofstream f(tmp, std::ios::binary); f.write(id, sizeof id); }
    if (rename(tmp.c_str(), path.c_str())) throw std::runtime_error("cannot publish the RCCL id at " + path);
  } else {
    for (int waited = 0;; ++waited) {
      struct stat st;
      if (stat(path.c_str(), &st) == 0 && st.st_size == (off_t)sizeof id && st.st_mtime >= started) {
        std::ifstream f(path, std::ios::binary);
        if (f.read(id, sizeof id)) break;
      }
      if (waited > 1200) throw std::runtime_error("no RCCL id from rank 0 at " + path + " after 120 s");
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
  }
  arch->commInitRccl(id);
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
    if (arch->world() > 1) rcclRendezvous(arch);
    hmult->simulate();
  } else if (ops == "hrotate") {
    HROTATE *hrotate = new HROTATE("test_hrotate", maxlevel, currentlevel, alpha, config, arch);
    if (arch->world() > 1) rcclRendezvous(arch);
    hrotate->simulate();
  } else if (ops == "hadd") {
    HADD *hadd = new HADD("test_hadd", maxlevel, currentlevel, alpha, config, arch);
    if (arch->world() > 1) rcclRendezvous(arch);
    hadd->simulate();
  } else if (ops == "pmult") {
    PMULT *pmult = new PMULT("test_pmult", maxlevel, currentlevel, alpha, config, arch);
    if (arch->world() > 1) rcclRendezvous(arch);
    pmult->simulate();
  } else if (ops == "padd") {
    PADD *padd = new PADD("test_ADD", maxlevel, currentlevel, alpha, config, arch);
    if (arch->world() > 1) rcclRendezvous(arch);
    padd->simulate();
  } else {
    std::cout << "Error operation requirement, please double confirm!\n";
  }
}
