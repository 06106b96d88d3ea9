// hm_dispatch.cpp — the ONE C ABI of include/homulator_hip.h in front of the two arithmetic back-ends (round 5).
//
// The kernels exist twice, built from the same sources (hm_modarith.h, HM_GENERIC):
//   libhm_m32.so   "mont32":  every modulus is q = h 2^32 + 1 — word-wise Montgomery reduction, six multiplies per butterfly, one-word
//                             twiddles: the fast path and the default chain;
//   libhm_gen.so   "generic": any distinct primes = 1 mod 2N between 2^20 and 2^60 — Shoup / Barrett arithmetic (SURVEY.md 8d's chain,
//                             a chain of 36-bit words as the reference's configuration models, a chain a caller's FHE library hands over).
// Both export the full ABI.  This library (libhomulator_hip.so, what callers link and what the host layer binds) maps them privately
// (dlopen RTLD_LOCAL, from its own directory), and hm_create picks one per CONTEXT from the chain it is given: mont32 when every modulus
// fits it (or the chain is the default), generic otherwise; HOMULATOR_ARITH=generic|mont32 forces one (A/B runs on one chain).
// hm_get_counter(ctx, "arith") says which runs (0 = mont32, 1 = generic).  A context handle of this library wraps the back-end's own.
// No compute happens here and there is still no CPU fallback: a back-end that cannot be loaded is an error of hm_create.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/homulator_hip.h"
#include "hm_params.h"
#include "hm_caps.h"

struct hm_ctx {
  int arith;        // 0 = mont32, 1 = generic
  hm_ctx *impl;     // the back-end's context
  std::string err;  // errors raised here (as opposed to inside the back-end)
  bool own_err = false;
  hipEvent_t ev_done = nullptr;   // hm_wait_for across back-ends
  int device = 0;
};
struct hm_graph {
  int arith;
  hm_graph *impl;
};
#define HM_IMPL(c) ((c)->impl)
#define HM_API(c) (g_api[(c)->arith])
#define HM_FORWARDED(c) (const_cast<hm_ctx *>(c)->own_err = false)
#include "hm_dispatch_gen.inc"

static void *g_handle[2] = {nullptr, nullptr};
static std::mutex g_lock;
static thread_local std::string g_create_err;
static const char *const kLibName[2] = {"libhm_m32.so", "libhm_gen.so"};
static const char *const kArithName[2] = {"mont32", "generic"};

// the back-ends live beside this library
static std::string own_dir() {
  Dl_info info;
  if (dladdr(reinterpret_cast<const void *>(&own_dir), &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t s = p.rfind('/');
    return s == std::string::npos ? std::string(".") : p.substr(0, s);
  }
  return ".";
}
static const char *load_backend(int arith) {   // nullptr = loaded
  std::lock_guard<std::mutex> g(g_lock);
  if (g_handle[arith]) return nullptr;
  static thread_local std::string msg;
  const std::string path = own_dir() + "/" + kLibName[arith];
  void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    const char *why = dlerror();   // (one call: dlerror() clears the message it returns)
    msg = std::string("cannot load the ") + kArithName[arith] + " arithmetic back-end " + path + ": " + (why ? why : "?") +
          " (build it: make -C homulator_amd/csrc); there is no CPU fallback";
    return msg.c_str();
  }
  if (const char *missing = hm_api_load(h, g_api[arith])) {
    msg = path + " does not export " + missing;
    dlclose(h);
    return msg.c_str();
  }
  g_handle[arith] = h;
  return nullptr;
}

extern "C" const char *hm_version(void) { return "homulator-hip 0.2 (gfx950; arithmetic back-ends: mont32, generic)"; }
extern "C" const char *hm_last_error(const hm_ctx *c) {
  if (!c) return g_create_err.c_str();
  if (c->own_err) return c->err.c_str();
  return g_api[c->arith].last_error(c->impl);
}

extern "C" hm_status hm_create(hm_ctx **out, const hm_params *p) {
  if (!out || !p) { g_create_err = "hm_create: null argument"; return HM_ERR_ARG; }
  *out = nullptr;
  // q and p go together; K == 0 needs no p (a default chain with no special moduli has neither)
  if ((p->q == nullptr && p->p != nullptr) || (p->q != nullptr && p->p == nullptr && p->K != 0)) { g_create_err = "hm_create: q and p go together"; return HM_ERR_ARG; }
  // which back-end: every modulus h 2^32 + 1 (or the default chain) -> mont32; anything else the generic one.  The chain itself is
  // validated by the back-end (primality, 1 mod 2N, range, duplicates): here only the form of the words is looked at.
  int arith = 0;
  for (uint32_t i = 0; p->q && i < p->L; ++i)
    if ((p->q[i] & 0xffffffffull) != 1 || (p->q[i] >> 32) == 0) arith = 1;
  for (uint32_t i = 0; p->p && i < p->K; ++i)
    if ((p->p[i] & 0xffffffffull) != 1 || (p->p[i] >> 32) == 0) arith = 1;
  if (const char *e = getenv("HOMULATOR_ARITH")) {
    if (!strcmp(e, "generic")) arith = 1;
    else if (!strcmp(e, "mont32")) {
      if (arith == 1) { g_create_err = "hm_create: HOMULATOR_ARITH=mont32, but the chain holds a modulus that is not h 2^32 + 1"; return HM_ERR_ARG; }
    } else if (*e) { g_create_err = "hm_create: HOMULATOR_ARITH is mont32 or generic"; return HM_ERR_ARG; }
  }
  if (const char *err = load_backend(arith)) { g_create_err = std::string("hm_create: ") + err; return HM_ERR_HIP; }
  hm_ctx *impl = nullptr;
  const hm_status st = g_api[arith].create(&impl, p);
  if (st != HM_OK) { g_create_err = g_api[arith].last_error(nullptr); return st; }
  hm_ctx *c = new hm_ctx;
  c->arith = arith;
  c->impl = impl;
  c->device = p->device;
  *out = c;
  return HM_OK;
}
extern "C" void hm_destroy(hm_ctx *c) {
  if (!c) return;
  g_api[c->arith].destroy(c->impl);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  delete c;
}

// work enqueued on `c` after this call starts only when everything enqueued so far on `producer` has finished.  The two contexts may
// run on different back-ends, so the event is recorded and awaited here, on the streams the back-ends hand out.
extern "C" hm_status hm_wait_for(hm_ctx *c, hm_ctx *producer) {
  if (!c || !producer) return HM_ERR_ARG;
  if (c == producer) return HM_OK;
  if (c->arith == producer->arith) { c->own_err = false; return g_api[c->arith].wait_for(c->impl, producer->impl); }
  c->own_err = true;
  if (c->device != producer->device) { c->err = "hm_wait_for: contexts on different devices"; return HM_ERR_ARG; }
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess && !producer->ev_done) e = hipEventCreateWithFlags(&producer->ev_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventRecord(producer->ev_done, static_cast<hipStream_t>(g_api[producer->arith].stream(producer->impl)));
  if (e == hipSuccess) e = hipStreamWaitEvent(static_cast<hipStream_t>(g_api[c->arith].stream(c->impl)), producer->ev_done, 0);
  if (e != hipSuccess) { c->err = std::string("hm_wait_for: ") + hipGetErrorString(e); return HM_ERR_HIP; }
  c->own_err = false;
  return HM_OK;
}

extern "C" hm_status hm_capture_end(hm_ctx *c, hm_graph **out) {
  if (!c || !out) return HM_ERR_ARG;
  c->own_err = false;
  hm_graph *impl = nullptr;
  const hm_status st = g_api[c->arith].capture_end(c->impl, &impl);
  if (st != HM_OK) return st;
  *out = new hm_graph{c->arith, impl};
  return HM_OK;
}
extern "C" hm_status hm_graph_launch(hm_ctx *c, hm_graph *g) {
  if (!c || !g) return HM_ERR_ARG;
  if (g->arith != c->arith) { c->own_err = true; c->err = "hm_graph_launch: the graph was captured from a context of the other arithmetic back-end"; return HM_ERR_ARG; }
  c->own_err = false;
  return g_api[c->arith].graph_launch(c->impl, g->impl);
}
extern "C" void hm_graph_destroy(hm_graph *g) {
  if (!g) return;
  g_api[g->arith].graph_destroy(g->impl);
  delete g;
}

// context-free helpers: the same in both back-ends; served by whichever is loaded (mont32 is tried first)
static HmApi *any_backend() {
  for (int a = 0; a < 2; ++a)
    if (g_handle[a]) return &g_api[a];
  for (int a = 0; a < 2; ++a)
    if (!load_backend(a)) return &g_api[a];
  return nullptr;
}
// the capability table is host-side data: no back-end is loaded for it
extern "C" hm_status hm_capability(uint32_t logN, const char *name, uint64_t *value) {
  if (!name || !value) return HM_ERR_ARG;
  return hm_cap_by_name(logN, name, value) ? HM_ERR_ARG : HM_OK;
}
extern "C" hm_status hm_comm_unique_id(void *out128) {
  HmApi *a = any_backend();
  return a ? a->comm_unique_id(out128) : HM_ERR_HIP;
}
extern "C" hm_status hm_slice_rows(const uint32_t *owners, uint32_t n, uint32_t world, uint32_t *rows) {
  HmApi *a = any_backend();
  return a ? a->slice_rows(owners, n, world, rows) : HM_ERR_HIP;
}
