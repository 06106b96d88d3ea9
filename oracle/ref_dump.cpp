// ref_dump.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
// A second `main` for the unmodified reference simulator, compiled together with the reference's own sources where
// they lie (oracle/Makefile: ref_dump -> oracle/_ref/ref_dump).  It opens the reference's classes to print what the
// stock binary keeps private, so that the build's own `sim` backend (host/src/SimModel.cpp, host/src/SimProgram.cpp)
// can be checked against the reference at a finer grain than the final cycle count:
//   ref_dump <cfg> <op> <L> <l> <alpha> ins     the literal per-cluster instruction queues and DRAM fetch lists
//   ref_dump <cfg> <op> <L> <l> <alpha> trace   completed instructions after every simulated cycle
// Nothing of the reference is copied: this file only calls it.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <sstream>
#include <stdio.h>
#include <string>
#include <typeinfo>
#include <vector>
#define private public
#include "Arch.h"
#include "Basic.h"
#include "Driver.h"
#include "Operation.h"
#undef private

static long g_probe[3] = {0, 0, 0};
template <class OP> static int go(OP *op, Arch *arch, const std::string &mode, uint32_t cluster) {
  Driver *driver = op->driver;
  if (mode == "ins") {
    // DRAM fetch lists (first-touch order per cluster) and the datamap's use counts
    for (uint32_t c = 0; c < cluster; c++) {
      std::printf("DRAM %u %zu\n", c, driver->dispatchedAddrHolder[c].size());
      for (auto &e : driver->dispatchedAddrHolder[c]) {
        for (auto a : e) std::printf("%llu ", a);
        std::printf("\n");
      }
    }
    std::printf("INPUTS %zu\n", op->datamap->inputDataAddr.size());
    for (auto &kv : op->datamap->inputDataAddr) std::printf("%llu %u\n", kv.first, kv.second);
    std::printf("OUTPUTS %zu\n", op->datamap->outDataAddr.size());
    for (auto &kv : op->datamap->outDataAddr) std::printf("%llu\n", kv.first);
    for (uint32_t c = 0; c < cluster; c++)
      for (const char *t : {"EWE", "NTT", "AUTO", "BCONV"}) {
        auto &q = driver->sentInsFIFO[c][t];
        std::printf("QUEUE %u %s %zu\n", c, t, q.size());
        for (auto &g : q) {
          std::printf("G %zu\n", g.size());
          for (Instruction *i : g) {
            std::printf("I %s |", i->GetInsName().c_str());
            for (uint32_t k = 0; k < i->getinputCount(); k++) std::printf(" %llu", i->getOperand(k));
            std::printf(" -> %llu\n", i->getOperandOut());
          }
        }
      }
    return 0;
  }
  driver->IssueInsFromDramToChip(arch);
  std::printf("TOTAL %llu\n", driver->getTotalIns());
  unsigned long long last = ~0ull;
  while (!arch->simulateComplete()) {
    driver->IssueDataFromDramToChip(op->memControlList);
    arch->update();
    if (mode == "probe") {  // front-end state of one cluster around a cycle window: ref_dump ... probe [cluster] <c> <from> <to>
      const unsigned c = (unsigned)g_probe[0];
      if (arch->getCycle() >= (unsigned long long)g_probe[1] && arch->getCycle() <= (unsigned long long)g_probe[2]) {
        std::printf("cyc %llu AUTO q=%zu f=%d d=%d i=%d io=%d commit=%zu | NTT q=%zu d=%d i=%d io=%d commit=%zu | EWE q=%zu commit=%zu\n", arch->getCycle(),
                    arch->IssuePortFromInsGenLayerOt[c]["AUTO"].size(), (int)arch->autofetchflag[c], (int)arch->autoDecodeflag[c],
                    (int)arch->autoIssueflag[c], (int)arch->autouIOs[c]->GetSignal(), arch->CommitInstAUTO[c].size(),
                    arch->IssuePortFromInsGenLayerOt[c]["NTT"].size(), (int)arch->nttDecodeflag[c], (int)arch->nttIssueflag[c],
                    (int)arch->nttuIOs[c]->GetSignal(), arch->CommitInstNTT[c].size(), arch->IssuePortFromInsGenLayerOt[c]["EWE"].size(),
                    arch->CommitInstEWE[c].size());
        std::printf("   auto stages used/out:");
        for (auto *st : arch->autous[c]->autous) std::printf(" %u/%d", st->used_pipeline, (int)st->output->GetSignal());
        std::printf("  units fifo %zu noc %zu dram %zu\n", arch->chipMemControl[c]->addrFromUnits.size(), arch->chipMemControl[c]->addrFromNoC.size(), arch->chipMemControl[c]->addrFromDram.size());
        if (arch->autoDecodeflag[c] && !arch->autoDecodeHold[c].empty()) {
          Instruction *i = arch->autoDecodeHold[c][0];
          std::printf("   decodeHold %s op0=%llu raw1=%llu raw2=%llu out=%llu\n", i->GetInsName().c_str(), i->getOperand(0), i->getOperand(1), i->getOperand(2), i->getOperandOut());
        }
      }
      continue;
    }
    if (arch->getcompletedIns() != last) {
      last = arch->getcompletedIns();
      std::printf("%llu %llu\n", arch->getCycle(), last);
    }
    if (arch->getCycle() > 50000000ull) break;
  }
  std::printf("CYCLES %llu\n", arch->getCycle());
  arch->shownStat();
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 7) {
    std::fprintf(stderr, "usage: %s <cfg> <op> <L> <l> <alpha> ins|trace [cluster]\n", argv[0]);
    return 1;
  }
  std::stringstream sink;
  std::streambuf *old = std::cout.rdbuf(sink.rdbuf());  // the constructors print the config echo and Malloc lines
  Config *config = new Config(argv[1]);
  const std::string ops = argv[2], mode = argv[6];
  const uint32_t L = std::atoi(argv[3]), l = std::atoi(argv[4]), a = std::atoi(argv[5]);
  if (mode == "probe") {
    if (argc < 10) { std::fprintf(stderr, "probe needs <cluster> <from> <to>\n"); return 1; }
    for (int i = 0; i < 3; i++) g_probe[i] = std::atol(argv[7 + i]);
  } else if (argc > 7) config->setValue("cluster", std::atoi(argv[7]));
  const uint32_t cluster = config->getValue("cluster");
  Arch *arch = new Arch(config);
  int rc = 1;
  if (ops == "hmult") rc = go(new HMULT("test_hmult", L, l, a, config, arch), arch, mode, cluster);
  else if (ops == "hrotate") rc = go(new HROTATE("test_hrotate", L, l, a, config, arch), arch, mode, cluster);
  else if (ops == "hadd") rc = go(new HADD("test_hadd", L, l, a, config, arch), arch, mode, cluster);
  else if (ops == "pmult") rc = go(new PMULT("test_pmult", L, l, a, config, arch), arch, mode, cluster);
  else if (ops == "padd") rc = go(new PADD("test_ADD", L, l, a, config, arch), arch, mode, cluster);
  std::cout.rdbuf(old);
  if (mode != "ins") {  // pass the stat block through
    const std::string s = sink.str();
    const size_t p = s.find("Start outPut statistic");
    if (p != std::string::npos) std::fputs(s.substr(p).c_str(), stdout);
  }
  return rc;
}
