// capi.cpp — C entry points of the host layer (see homulator_host.h).
#include "../include/homulator_host.h"

#include <cstring>
#include <sstream>

#include "Operation.h"
#include "RcclRendezvous.h"

struct hh_op {
  Config *cfg = nullptr;
  Arch *arch = nullptr;
  OperationBase *op = nullptr;
};
static thread_local std::string g_err;
const char *hh_last_error(void) { return g_err.c_str(); }

struct QuietScope {  // swallow std::cout while constructing (the reference prints its config and Malloc plan)
  std::streambuf *old = nullptr;
  std::ostringstream sink;
  explicit QuietScope(bool on) { if (on) old = std::cout.rdbuf(sink.rdbuf()); }
  ~QuietScope() { if (old) std::cout.rdbuf(old); }
};

#define HH_TRY(body)                      \
  try { body; return 0; }                 \
  catch (const std::exception &e) { g_err = e.what(); return 1; }

int hh_op_create(hh_op **out, const char *cfg_path, const char *op_name, uint32_t maxLevel, uint32_t curLevel, uint32_t alpha,
                 int backend, int fuse, int device, const char *overrides, int quiet) {
  if (!out || !cfg_path || !op_name) { g_err = "null argument"; return 1; }
  *out = nullptr;
  hh_op *h = new hh_op;
  try {
    QuietScope qs(quiet != 0);
    h->cfg = new Config(cfg_path);
    h->cfg->getValue("N");  // a missing file leaves the map empty: fail here like upstream would at first use
    h->cfg->setValue("backend", (uint32_t)backend);
    h->cfg->setValue("fuse", (uint32_t)fuse);
    h->cfg->setValue("device", (uint32_t)device);
    if (overrides) {
      std::stringstream ss(overrides);
      std::string kv;
      while (std::getline(ss, kv, ';')) {
        size_t eq = kv.find('=');
        if (eq != std::string::npos) h->cfg->setValue(kv.substr(0, eq), (uint32_t)std::stoul(kv.substr(eq + 1)));
      }
    }
    if (curLevel == 0 || curLevel > maxLevel || alpha == 0) throw std::runtime_error("need 0 < curLevel <= maxLevel and alpha > 0");
    h->arch = new Arch(h->cfg);
    const std::string o = op_name;
    if (o == "hmult") h->op = new HMULT("test_hmult", maxLevel, curLevel, alpha, h->cfg, h->arch);
    else if (o == "hrotate") h->op = new HROTATE("test_hrotate", maxLevel, curLevel, alpha, h->cfg, h->arch);
    else if (o == "hadd") h->op = new HADD("test_hadd", maxLevel, curLevel, alpha, h->cfg, h->arch);
    else if (o == "pmult") h->op = new PMULT("test_pmult", maxLevel, curLevel, alpha, h->cfg, h->arch);
    else if (o == "padd") h->op = new PADD("test_ADD", maxLevel, curLevel, alpha, h->cfg, h->arch);
    else throw std::runtime_error("Error operation requirement, please double confirm!");
  } catch (const std::exception &e) {
    g_err = e.what();
    hh_op_destroy(h);
    return 1;
  }
  *out = h;
  return 0;
}
void hh_op_destroy(hh_op *h) {
  if (!h) return;
  delete h->op;
  delete h->arch;
  delete h->cfg;
  delete h;
}
int hh_op_simulate(hh_op *h) { HH_TRY(h->op->simulate()) }
int hh_op_sim_run(hh_op *h, uint64_t *cycles, uint64_t *retired, int *drained) {
  HH_TRY(const bool ok = h->op->simulateCycles(false); if (cycles) *cycles = h->arch->getCycle(); if (retired) *retired = h->arch->getcompletedIns();
         if (drained) *drained = ok ? 1 : 0)
}
int hh_op_sim_stats(hh_op *h, char *out, uint32_t cap) {
  HH_TRY(if (!h->arch->simModel()) throw std::runtime_error("no cycle model: backend is not sim, or hh_op_sim_run was not called");
         std::string s; for (const auto &kv : h->arch->simModel()->stats()) s += kv.first + " " + std::to_string(kv.second) + "\n";
         if (s.size() + 1 > cap) throw std::runtime_error("buffer too small"); std::memcpy(out, s.c_str(), s.size() + 1))
}
int hh_op_execute(hh_op *h, uint32_t iters, double *ns) { HH_TRY(double t = h->op->execute(iters); if (ns) *ns = t) }
int hh_op_enqueue(hh_op *h, uint32_t iters) { HH_TRY(h->op->prepare(); for (uint32_t i = 0; i < iters; ++i) h->arch->run()) }
int hh_op_sync(hh_op *h) { HH_TRY(h->arch->sync()) }
int hh_op_total_instructions(hh_op *h, uint64_t *t) { HH_TRY(*t = h->op->totalInstructions()) }
int hh_op_launch_count(hh_op *h, uint64_t *n) { HH_TRY(h->op->prepare(); *n = h->arch->launchCount()) }
int hh_op_stage_bytes(hh_op *h, uint64_t *b) { HH_TRY(h->op->prepare(); *b = h->arch->algorithmicBytes()) }
int hh_op_buffer_limbs(hh_op *h, const char *name, uint32_t *n) { HH_TRY(*n = (uint32_t)h->op->bufferAddrs(name).size()) }
int hh_op_read_buffer(hh_op *h, const char *name, uint64_t *host) {
  HH_TRY(if (!h->op->readBuffer(name, host)) throw std::runtime_error("buffer not readable (count backend or not prepared)"))
}
int hh_op_buffer_names(hh_op *h, char *out, uint32_t cap) {
  HH_TRY(std::string s; for (auto &n : h->op->bufferNames()) s += n + "\n"; if (s.size() + 1 > cap) throw std::runtime_error("buffer too small");
         memcpy(out, s.c_str(), s.size() + 1))
}
int hh_op_read_buffer_copy(hh_op *h, const char *name, uint32_t copy, uint64_t *host) {
  HH_TRY(if (!h->op->readBuffer(name, host, copy)) throw std::runtime_error("buffer not readable (count backend, not prepared, or copy >= batch)"))
}
int hh_op_write_buffer(hh_op *h, const char *name, uint32_t copy, const uint64_t *host) {
  HH_TRY(if (!h->op->writeBuffer(name, host, copy)) throw std::runtime_error("buffer not writable (not the hip backend, or copy >= batch)"))
}
uint32_t hh_op_batch(hh_op *h) { return h->arch->batch(); }
uint32_t hh_op_N(hh_op *h) { return h->arch->N(); }
int hh_op_plan(hh_op *h, char *out, uint32_t cap) {
  HH_TRY(h->op->prepare(); std::string s = h->arch->planText(); if (s.size() + 1 > cap) throw std::runtime_error("buffer too small");
         memcpy(out, s.c_str(), s.size() + 1))
}
int hh_op_stage_times(hh_op *h, uint32_t iters, char *out, uint32_t cap) {
  HH_TRY(h->op->prepare(); std::string s = h->arch->stageTimes(iters); if (s.size() + 1 > cap) throw std::runtime_error("buffer too small");
         memcpy(out, s.c_str(), s.size() + 1))
}
extern "C" int hm_get_counter(hm_ctx *, const char *, uint64_t *);
extern "C" const char *hm_last_error(const hm_ctx *);
int hh_op_backend_counter(hh_op *h, const char *name, uint64_t *value) {
  HH_TRY(h->op->prepare(); if (!h->arch->context()) throw std::runtime_error("no backend context (not the hip backend)");
         if (hm_get_counter(h->arch->context(), name, value)) throw std::runtime_error(hm_last_error(h->arch->context())))
}
int hh_op_bind_input(hh_op *dst, const char *input, hh_op *src) { HH_TRY(dst->op->bindInput(input, src->op)) }

struct hh_chain {
  OpChain *chain = nullptr;
  std::vector<hh_op *> views;  // non-owning per-op handles for the read / plan calls
};
int hh_chain_create(hh_chain **out, const char *cfg_path, const char *op_list, uint32_t maxLevel, uint32_t curLevel, uint32_t alpha,
                    const char *overrides, int quiet) {
  if (!out || !cfg_path || !op_list) { g_err = "null argument"; return 1; }
  *out = nullptr;
  try {
    QuietScope qs(quiet != 0);
    std::map<std::string, uint32_t> ov;
    if (overrides) {
      std::stringstream ss(overrides);
      std::string kv;
      while (std::getline(ss, kv, ';')) {
        size_t eq = kv.find('=');
        if (eq != std::string::npos) ov[kv.substr(0, eq)] = (uint32_t)std::stoul(kv.substr(eq + 1));
      }
    }
    hh_chain *h = new hh_chain;
    h->chain = new OpChain(cfg_path, op_list, maxLevel, curLevel, alpha, ov);
    for (size_t i = 0; i < h->chain->size(); ++i) {
      hh_op *v = new hh_op;
      v->op = h->chain->op(i);
      v->arch = v->op->getArch();
      h->views.push_back(v);
    }
    *out = h;
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}
void hh_chain_destroy(hh_chain *h) {
  if (!h) return;
  for (hh_op *v : h->views) delete v;  // views own nothing
  delete h->chain;
  delete h;
}
uint32_t hh_chain_size(hh_chain *h) { return (uint32_t)h->chain->size(); }
hh_op *hh_chain_op(hh_chain *h, uint32_t i) { return i < h->views.size() ? h->views[i] : nullptr; }
int hh_chain_execute(hh_chain *h, uint32_t iters, double *ns) { HH_TRY(double t = h->chain->execute(iters); if (ns) *ns = t) }
int hh_chain_simulate(hh_chain *h) { HH_TRY(h->chain->simulate()) }
int hh_chain_enqueue(hh_chain *h, uint32_t iters) { HH_TRY(h->chain->prepare(); for (uint32_t i = 0; i < iters; ++i) h->chain->run()) }
int hh_chain_sync(hh_chain *h) { HH_TRY(h->chain->sync()) }
int hh_op_refill(hh_op *h, const char *input, uint64_t seed) {
  HH_TRY(h->op->prepare(); h->arch->refill(h->op->bufferAddrs(std::string(input) + ".c0"), seed);
         h->arch->refill(h->op->bufferAddrs(std::string(input) + ".c1"), seed + 1000))
}
int hh_op_snapshot(hh_op *h, const char *name, uint32_t slot) { HH_TRY(h->op->prepare(); h->arch->snapshot(h->op->bufferAddrs(name), slot)) }
int hh_op_snapshot_read(hh_op *h, uint32_t slot, uint64_t *host) {
  HH_TRY(if (!h->arch->readSnapshot(slot, host)) throw std::runtime_error("no such snapshot"))
}
extern "C" int hm_comm_unique_id(void *);
int hh_comm_unique_id(void *out) { if (hm_comm_unique_id(out)) { g_err = "hm_comm_unique_id failed (is librccl.so available?)"; return 1; } return 0; }
int hh_op_comm_init_rccl(hh_op *h, const void *id) { HH_TRY(h->arch->commInitRccl(id)) }
int hh_op_comm_init_external(hh_op *h, void *fn, void *user) { HH_TRY(h->arch->commInitExternal(fn, user)) }

int hh_rccl_id_path(char *out, uint32_t cap) {
  const std::string p = hrv::idPath();
  if (!out || p.size() + 1 > cap) { g_err = "buffer too small"; return 1; }
  memcpy(out, p.c_str(), p.size() + 1);
  return 0;
}
int hh_rccl_id_publish(const char *path, const void *id128) { HH_TRY(hrv::publish(path, static_cast<const char *>(id128))); }
int hh_rccl_id_fetch(const char *path, void *id128, uint32_t timeout_ms) { HH_TRY(hrv::fetch(path, static_cast<char *>(id128), timeout_ms)); }
int hh_rccl_id_remove(const char *path) { HH_TRY(hrv::removeStale(path)); }
