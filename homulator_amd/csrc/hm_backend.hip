// hm_backend.hip — HIP kernels (gfx950) + context + the C ABI of include/homulator_hip.h.
// There is no CPU fallback in this library: every compute entry point launches a HIP kernel.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/homulator_hip.h"
#include "hm_elem_core.h"
#include "hm_modarith.h"
#include "hm_ntt_core.h"
#include "hm_params.h"

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------

// XCD-aware block -> (limb entry, tile) map.  Blocks are dealt round-robin over the 8 XCDs, so blocks
// b and b+8 share an L2; all tiles of one limb-poly get the same b % 8, and both passes of a
// transform use the same map, so the second pass finds the first pass's output in that XCD's L2.
__device__ __forceinline__ bool hm_block_map(uint32_t tiles_per_limb, uint32_t n_limbs, uint32_t &entry, uint32_t &tile) {
  uint32_t b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
  entry = (slot / tiles_per_limb) * 8u + xcd;
  tile = slot % tiles_per_limb;
  return entry < n_limbs;
}

template <int LOGR, bool STRIDED, bool INV, int MODE>
__device__ __forceinline__ void hm_ntt_pass_body(const HmNttArgs &a, const HmTw *scale) {
  __shared__ __attribute__((aligned(16))) uint64_t lds[HM_LDS_WORDS];
  uint32_t entry, tile;
  if (!hm_block_map(1u << (a.logN - HM_TILE_LOG), a.n_limbs, entry, tile)) return;
  const int tid = threadIdx.x;
  const HmLimb lb = a.limb[entry];
  const size_t N = (size_t)1 << a.logN;
  const uint64_t q = a.mods[lb.mod].q;
  const HmTw *twl = a.tw + (size_t)lb.mod * N;
  const uint32_t s0 = STRIDED ? 0u : (a.logN - 8u);
  const uint32_t prefix0 = STRIDED ? 0u : (tile << (HM_TILE_LOG - LOGR));
  // the first pass of a transform reads `in`, the second works in place on `out`
  constexpr bool FIRST = (STRIDED != INV);
  const uint64_t *src = FIRST ? a.in + (size_t)lb.in * N : a.out + (size_t)lb.out * N;
  uint64_t *dst = a.out + (size_t)lb.out * N;

  hm_tile_load<LOGR, STRIDED>(tid, lds, src, tile);
  __syncthreads();
  using RS = HmRounds<LOGR>;
  if constexpr (!INV) {
    hm_ntt_round<LOGR, STRIDED, RS::nb[0], RS::k[0], false>(tid, lds, twl, s0, prefix0, q);
    __syncthreads();
    hm_ntt_round<LOGR, STRIDED, RS::nb[1], RS::k[1], false>(tid, lds, twl, s0, prefix0, q);
    __syncthreads();
    if constexpr (RS::n > 2) {
      hm_ntt_round<LOGR, STRIDED, RS::nb[2], RS::k[2], false>(tid, lds, twl, s0, prefix0, q);
      __syncthreads();
    }
  } else {
    if constexpr (RS::n > 2) {
      hm_ntt_round<LOGR, STRIDED, RS::nb[2], RS::k[2], true>(tid, lds, twl, s0, prefix0, q);
      __syncthreads();
    }
    hm_ntt_round<LOGR, STRIDED, RS::nb[1], RS::k[1], true>(tid, lds, twl, s0, prefix0, q);
    __syncthreads();
    hm_ntt_round<LOGR, STRIDED, RS::nb[0], RS::k[0], true>(tid, lds, twl, s0, prefix0, q);
    __syncthreads();
  }
  HmTw sc = {0, 0};
  if constexpr (MODE == 2) sc = scale[entry];
  hm_tile_store<LOGR, STRIDED, MODE>(tid, lds, dst, tile, q, sc);
}

template <int LOGR, bool STRIDED, bool INV, int MODE>
__global__ void __launch_bounds__(HM_THREADS) k_ntt_pass(HmNttArgs a) {
  hm_ntt_pass_body<LOGR, STRIDED, INV, MODE>(a, nullptr);
}
template <int LOGR>
__global__ void __launch_bounds__(HM_THREADS) k_intt_final(HmNttArgs a, HmScale s) {
  hm_ntt_pass_body<LOGR, true, true, 2>(a, s.c);
}

template <int OP>
__global__ void __launch_bounds__(256) k_ewe(HmEweArgs a) {
  const uint32_t N = 1u << a.logN;
  const uint32_t per_limb = N / 512;  // blocks per limb: 256 threads x 2 coefficients
  uint32_t entry = blockIdx.x / per_limb, chunk = blockIdx.x % per_limb;
  if (entry >= a.n_limbs) return;
  const HmEweLimb lb = a.limb[entry];
  const HmMod m = a.mods[lb.mod];
  const HmTw k = a.k[entry];
  const size_t x = (size_t)chunk * 512 + 2 * threadIdx.x;
  constexpr int uses = hm_ewe_uses(OP);
  ulonglong2 va = {0, 0}, vb = {0, 0}, vc = {0, 0}, vd = {0, 0};
  if (uses & 1) va = *reinterpret_cast<const ulonglong2 *>(a.a + (size_t)lb.a * N + x);
  if (uses & 2) vb = *reinterpret_cast<const ulonglong2 *>(a.b + (size_t)lb.b * N + x);
  if (uses & 4) vc = *reinterpret_cast<const ulonglong2 *>(a.c + (size_t)lb.c * N + x);
  if (uses & 8) vd = *reinterpret_cast<const ulonglong2 *>(a.d + (size_t)lb.d * N + x);
  ulonglong2 r;
  r.x = hm_ewe_one<OP>(va.x, vb.x, vc.x, vd.x, k, m);
  r.y = hm_ewe_one<OP>(va.y, vb.y, vc.y, vd.y, k, m);
  *reinterpret_cast<ulonglong2 *>(a.out + (size_t)lb.out * N + x) = r;
}

__global__ void __launch_bounds__(128) k_bconv(HmBconvArgs a) {
  uint32_t x = blockIdx.x * 128 + threadIdx.x;
  uint32_t t0 = blockIdx.y * a.out_per_block;
  uint32_t t1 = min(t0 + a.out_per_block, a.n_out);
  hm_bconv_thread(a, x, t0, t1);
}

struct HmAutoArgs {
  const uint64_t *in;
  uint64_t *out;
  uint32_t logN, n_limbs, g;
  HmLimb limb[HM_MAX_LIMBS];
};
__global__ void __launch_bounds__(256) k_automorph(HmAutoArgs a) {
  const uint32_t N = 1u << a.logN;
  const uint32_t per_limb = N / 256;
  uint32_t entry = blockIdx.x / per_limb;
  uint32_t i = (blockIdx.x % per_limb) * 256 + threadIdx.x;
  const HmLimb lb = a.limb[entry];
  a.out[(size_t)lb.out * N + i] = a.in[(size_t)lb.in * N + hm_auto_src(i, a.g, a.logN)];
}

struct HmFillArgs {
  uint64_t *out;
  const HmMod *mods;
  uint64_t seed;
  uint32_t logN, n_limbs;
  HmLimb limb[HM_MAX_LIMBS];
};
__global__ void __launch_bounds__(256) k_fill(HmFillArgs a) {
  const uint32_t N = 1u << a.logN;
  const uint32_t per_limb = N / 256;
  uint32_t entry = blockIdx.x / per_limb;
  uint32_t x = (blockIdx.x % per_limb) * 256 + threadIdx.x;
  const HmLimb lb = a.limb[entry];
  a.out[(size_t)lb.out * N + x] = hm_synth(a.seed + lb.aux, x, a.mods[lb.mod].q);
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct hm_ctx {
  hm::Params P;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  HmTw *d_tw_fwd = nullptr, *d_tw_inv = nullptr;
  HmMod *d_mods = nullptr;
  std::map<std::vector<uint32_t>, uint64_t *> bconv_tables;  // key: n_in, in_ids..., out_ids...
  std::string err;
};

static thread_local std::string g_create_err;

static hm_status fail(hm_ctx *c, hm_status st, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_err = buf;
  return st;
}
#define HM_HIP(c, expr)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail((c), HM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

extern "C" const char *hm_version(void) { return "homulator-hip 0.1 (gfx950)"; }
extern "C" const char *hm_last_error(const hm_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

extern "C" hm_status hm_create(hm_ctx **out, const hm_params *p) {
  if (!out || !p) return fail(nullptr, HM_ERR_ARG, "hm_create: null argument");
  *out = nullptr;
  std::unique_ptr<hm_ctx> c(new hm_ctx);
  try {
    c->P.init(p->logN, p->L, p->K, p->q, p->p, p->psi);
  } catch (const std::exception &e) {
    return fail(nullptr, HM_ERR_ARG, "hm_create: %s", e.what());
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, HM_ERR_HIP, "hm_create: no HIP device available (%s); this backend has no CPU fallback",
                hipGetErrorString(e));
  if (p->device < 0 || p->device >= ndev) return fail(nullptr, HM_ERR_ARG, "hm_create: device %d out of range", p->device);
  c->device = p->device;
  hm_ctx *cc = c.get();
  HM_HIP(nullptr, hipSetDevice(cc->device));
  HM_HIP(nullptr, hipStreamCreateWithFlags(&cc->stream, hipStreamNonBlocking));
  HM_HIP(nullptr, hipEventCreate(&cc->ev0));
  HM_HIP(nullptr, hipEventCreate(&cc->ev1));
  const uint32_t M = cc->P.L + cc->P.K, N = cc->P.N;
  HM_HIP(nullptr, hipMalloc(&cc->d_tw_fwd, sizeof(HmTw) * (size_t)M * N));
  HM_HIP(nullptr, hipMalloc(&cc->d_tw_inv, sizeof(HmTw) * (size_t)M * N));
  HM_HIP(nullptr, hipMalloc(&cc->d_mods, sizeof(HmMod) * M));
  std::vector<HmTw> tmp(N);
  for (uint32_t m = 0; m < M; ++m) {
    cc->P.make_table(m, false, tmp.data());
    HM_HIP(nullptr, hipMemcpy(cc->d_tw_fwd + (size_t)m * N, tmp.data(), sizeof(HmTw) * N, hipMemcpyHostToDevice));
    cc->P.make_table(m, true, tmp.data());
    HM_HIP(nullptr, hipMemcpy(cc->d_tw_inv + (size_t)m * N, tmp.data(), sizeof(HmTw) * N, hipMemcpyHostToDevice));
  }
  HM_HIP(nullptr, hipMemcpy(cc->d_mods, cc->P.modc.data(), sizeof(HmMod) * M, hipMemcpyHostToDevice));
  *out = c.release();
  return HM_OK;
}

extern "C" void hm_destroy(hm_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto &kv : c->bconv_tables) (void)hipFree(kv.second);
  (void)hipFree(c->d_tw_fwd);
  (void)hipFree(c->d_tw_inv);
  (void)hipFree(c->d_mods);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" hm_status hm_get_modulus(const hm_ctx *c, uint32_t m, uint64_t *q) {
  if (!c || !q || m >= c->P.L + c->P.K) return HM_ERR_ARG;
  *q = c->P.mod[m];
  return HM_OK;
}
extern "C" hm_status hm_get_psi(const hm_ctx *c, uint32_t m, uint64_t *psi) {
  if (!c || !psi || m >= c->P.L + c->P.K) return HM_ERR_ARG;
  *psi = c->P.psi[m];
  return HM_OK;
}

extern "C" hm_status hm_malloc(hm_ctx *c, size_t bytes, void **dptr) {
  if (!c || !dptr) return HM_ERR_ARG;
  HM_HIP(c, hipSetDevice(c->device));
  HM_HIP(c, hipMalloc(dptr, bytes));
  return HM_OK;
}
extern "C" hm_status hm_free(hm_ctx *c, void *dptr) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipStreamSynchronize(c->stream));
  HM_HIP(c, hipFree(dptr));
  return HM_OK;
}
extern "C" hm_status hm_memcpy_h2d(hm_ctx *c, void *dst, const void *src, size_t bytes) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HM_HIP(c, hipStreamSynchronize(c->stream));
  return HM_OK;
}
extern "C" hm_status hm_memcpy_d2h(hm_ctx *c, void *dst, const void *src, size_t bytes) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HM_HIP(c, hipStreamSynchronize(c->stream));
  return HM_OK;
}
extern "C" hm_status hm_memcpy_d2d(hm_ctx *c, void *dst, const void *src, size_t bytes) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return HM_OK;
}
extern "C" hm_status hm_sync(hm_ctx *c) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipStreamSynchronize(c->stream));
  return HM_OK;
}
extern "C" void *hm_stream(hm_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" hm_status hm_timer_start(hm_ctx *c) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipEventRecord(c->ev0, c->stream));
  return HM_OK;
}
extern "C" hm_status hm_timer_stop(hm_ctx *c, uint64_t *ns) {
  if (!c || !ns) return HM_ERR_ARG;
  HM_HIP(c, hipEventRecord(c->ev1, c->stream));
  HM_HIP(c, hipEventSynchronize(c->ev1));
  float ms = 0;
  HM_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *ns = (uint64_t)((double)ms * 1e6);
  return HM_OK;
}

// ------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------
static inline uint32_t limb_at(const uint32_t *l, uint32_t i) { return l ? l[i] : i; }

static hm_status check_limbs(hm_ctx *c, const char *what, const uint32_t *l, uint32_t n) {
  for (uint32_t i = 0; l && i < n; ++i)
    if (l[i] > 0xFFFFu) return fail(c, HM_ERR_ARG, "%s: limb index %u exceeds 65535", what, l[i]);
  return HM_OK;
}
static hm_status check_mods(hm_ctx *c, const char *what, const uint32_t *m, uint32_t n) {
  if (!m) return fail(c, HM_ERR_ARG, "%s: mod_ids is null", what);
  for (uint32_t i = 0; i < n; ++i)
    if (m[i] >= c->P.L + c->P.K) return fail(c, HM_ERR_ARG, "%s: mod id %u out of range", what, m[i]);
  return HM_OK;
}

template <int LOG1>
static void launch_ntt(hm_ctx *c, const HmNttArgs &a, const HmScale *sc, bool inverse) {
  const uint32_t tiles = c->P.N >> HM_TILE_LOG;
  dim3 grid(((a.n_limbs + 7) / 8) * 8 * tiles), block(HM_THREADS);
  if (!inverse) {
    hipLaunchKernelGGL((k_ntt_pass<LOG1, true, false, 0>), grid, block, 0, c->stream, a);
    hipLaunchKernelGGL((k_ntt_pass<8, false, false, 1>), grid, block, 0, c->stream, a);
  } else {
    hipLaunchKernelGGL((k_ntt_pass<8, false, true, 0>), grid, block, 0, c->stream, a);
    hipLaunchKernelGGL((k_intt_final<LOG1>), grid, block, 0, c->stream, a, *sc);
  }
}

extern "C" hm_status hm_ntt(hm_ctx *c, const uint64_t *in, const uint32_t *in_limbs, uint64_t *out,
                            const uint32_t *out_limbs, const uint32_t *mod_ids, uint32_t n, int inverse,
                            const uint64_t *scale) {
  if (!c) return HM_ERR_ARG;
  if (!in || !out) return fail(c, HM_ERR_ARG, "hm_ntt: null buffer");
  if (scale && !inverse) return fail(c, HM_ERR_ARG, "hm_ntt: scale is only defined for the inverse transform");
  hm_status st;
  if ((st = check_limbs(c, "hm_ntt", in_limbs, n)) || (st = check_limbs(c, "hm_ntt", out_limbs, n)) ||
      (st = check_mods(c, "hm_ntt", mod_ids, n)))
    return st;
  HM_HIP(c, hipSetDevice(c->device));
  for (uint32_t base = 0; base < n; base += HM_MAX_LIMBS) {
    const uint32_t cnt = std::min<uint32_t>(HM_MAX_LIMBS, n - base);
    HmNttArgs a;
    a.in = in; a.out = out;
    a.tw = inverse ? c->d_tw_inv : c->d_tw_fwd;
    a.mods = c->d_mods;
    a.logN = c->P.logN; a.n_limbs = cnt;
    HmScale sc;
    for (uint32_t i = 0; i < cnt; ++i) {
      const uint32_t g = base + i, m = mod_ids[g];
      a.limb[i] = HmLimb{(uint16_t)limb_at(in_limbs, g), (uint16_t)limb_at(out_limbs, g), (uint16_t)m, 0};
      if (inverse) {
        const uint64_t q = c->P.mod[m];
        uint64_t k = c->P.modc[m].ninv;
        if (scale) {
          if (scale[g] >= q) return fail(c, HM_ERR_ARG, "hm_ntt: scale[%u] is not reduced", g);
          k = hm::mulmod(k, scale[g], q);
        }
        sc.c[i] = HmTw{k, hm::shoup(k, q)};
      }
    }
    switch (c->P.logN - 8) {
    case 5: launch_ntt<5>(c, a, &sc, inverse); break;
    case 6: launch_ntt<6>(c, a, &sc, inverse); break;
    case 7: launch_ntt<7>(c, a, &sc, inverse); break;
    case 8: launch_ntt<8>(c, a, &sc, inverse); break;
    case 9: launch_ntt<9>(c, a, &sc, inverse); break;
    default: return fail(c, HM_ERR_UNSUPPORTED, "hm_ntt: logN %u", c->P.logN);
    }
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}

extern "C" hm_status hm_automorph(hm_ctx *c, const uint64_t *in, const uint32_t *in_limbs, uint64_t *out,
                                  const uint32_t *out_limbs, uint32_t n, uint32_t galois) {
  if (!c) return HM_ERR_ARG;
  if (!in || !out) return fail(c, HM_ERR_ARG, "hm_automorph: null buffer");
  if (!(galois & 1) || galois >= 2 * c->P.N) return fail(c, HM_ERR_ARG, "hm_automorph: galois element must be odd and < 2N");
  hm_status st;
  if ((st = check_limbs(c, "hm_automorph", in_limbs, n)) || (st = check_limbs(c, "hm_automorph", out_limbs, n))) return st;
  HM_HIP(c, hipSetDevice(c->device));
  for (uint32_t base = 0; base < n; base += HM_MAX_LIMBS) {
    const uint32_t cnt = std::min<uint32_t>(HM_MAX_LIMBS, n - base);
    HmAutoArgs a;
    a.in = in; a.out = out; a.logN = c->P.logN; a.n_limbs = cnt; a.g = galois;
    for (uint32_t i = 0; i < cnt; ++i)
      a.limb[i] = HmLimb{(uint16_t)limb_at(in_limbs, base + i), (uint16_t)limb_at(out_limbs, base + i), 0, 0};
    hipLaunchKernelGGL(k_automorph, dim3(cnt * (c->P.N / 256)), dim3(256), 0, c->stream, a);
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}

template <int OP>
static void launch_ewe(hm_ctx *c, const HmEweArgs &a) {
  hipLaunchKernelGGL((k_ewe<OP>), dim3(a.n_limbs * (c->P.N / 512)), dim3(256), 0, c->stream, a);
}

extern "C" hm_status hm_ewe(hm_ctx *c, int op, const uint64_t *pa, const uint32_t *la, const uint64_t *pb,
                            const uint32_t *lb, const uint64_t *pc, const uint32_t *lc, const uint64_t *pd,
                            const uint32_t *ld, uint64_t *out, const uint32_t *lo, const uint32_t *mod_ids,
                            uint32_t n, const uint64_t *k) {
  if (!c) return HM_ERR_ARG;
  if (op < 0 || op >= HM_EWE_NOPS) return fail(c, HM_ERR_ARG, "hm_ewe: bad opcode %d", op);
  const int uses = hm_ewe_uses(op);
  if (((uses & 1) && !pa) || ((uses & 2) && !pb) || ((uses & 4) && !pc) || ((uses & 8) && !pd) || !out)
    return fail(c, HM_ERR_ARG, "hm_ewe: opcode %d is missing an operand", op);
  const bool needs_k = op == HM_EWE_MUL_CONST || op == HM_EWE_SUB_SCALE || op == HM_EWE_SUB_SCALE_ADD;
  if (needs_k && !k) return fail(c, HM_ERR_ARG, "hm_ewe: opcode %d needs constants", op);
  hm_status st;
  if ((st = check_limbs(c, "hm_ewe", la, n)) || (st = check_limbs(c, "hm_ewe", lb, n)) ||
      (st = check_limbs(c, "hm_ewe", lc, n)) || (st = check_limbs(c, "hm_ewe", ld, n)) ||
      (st = check_limbs(c, "hm_ewe", lo, n)) || (st = check_mods(c, "hm_ewe", mod_ids, n)))
    return st;
  HM_HIP(c, hipSetDevice(c->device));
  for (uint32_t base = 0; base < n; base += HM_MAX_LIMBS) {
    const uint32_t cnt = std::min<uint32_t>(HM_MAX_LIMBS, n - base);
    HmEweArgs a;
    a.a = pa; a.b = pb; a.c = pc; a.d = pd; a.out = out;
    a.mods = c->d_mods; a.logN = c->P.logN; a.n_limbs = cnt; a.op = (uint32_t)op;
    for (uint32_t i = 0; i < cnt; ++i) {
      const uint32_t g = base + i, m = mod_ids[g];
      a.limb[i] = HmEweLimb{(uint16_t)limb_at(la, g), (uint16_t)limb_at(lb, g), (uint16_t)limb_at(lc, g),
                            (uint16_t)limb_at(ld, g), (uint16_t)limb_at(lo, g), (uint16_t)m};
      a.k[i] = HmTw{0, 0};
      if (needs_k) {
        if (k[g] >= c->P.mod[m]) return fail(c, HM_ERR_ARG, "hm_ewe: k[%u] is not reduced", g);
        a.k[i] = HmTw{k[g], hm::shoup(k[g], c->P.mod[m])};
      }
    }
    switch (op) {
    case HM_EWE_MUL: launch_ewe<HM_EWE_MUL>(c, a); break;
    case HM_EWE_MAC2: launch_ewe<HM_EWE_MAC2>(c, a); break;
    case HM_EWE_MAC_ADD: launch_ewe<HM_EWE_MAC_ADD>(c, a); break;
    case HM_EWE_ADD: launch_ewe<HM_EWE_ADD>(c, a); break;
    case HM_EWE_SUB: launch_ewe<HM_EWE_SUB>(c, a); break;
    case HM_EWE_MUL_CONST: launch_ewe<HM_EWE_MUL_CONST>(c, a); break;
    case HM_EWE_SUB_SCALE: launch_ewe<HM_EWE_SUB_SCALE>(c, a); break;
    case HM_EWE_COPY: launch_ewe<HM_EWE_COPY>(c, a); break;
    case HM_EWE_SUB_SCALE_ADD: launch_ewe<HM_EWE_SUB_SCALE_ADD>(c, a); break;
    }
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}

extern "C" hm_status hm_bconv_consts(hm_ctx *c, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids,
                                     uint32_t n_out, uint64_t *qhat_inv, uint64_t *table) {
  if (!c) return HM_ERR_ARG;
  if (!in_ids || n_in == 0 || (n_out && !out_ids)) return fail(c, HM_ERR_ARG, "hm_bconv_consts: bad basis");
  hm_status st;
  if ((st = check_mods(c, "hm_bconv_consts", in_ids, n_in)) || (n_out && (st = check_mods(c, "hm_bconv_consts", out_ids, n_out))))
    return st;
  std::vector<uint64_t> qh(n_in), tb((size_t)n_in * std::max<uint32_t>(n_out, 1));
  c->P.bconv_consts(in_ids, n_in, out_ids, n_out, qh.data(), tb.data());
  if (qhat_inv) memcpy(qhat_inv, qh.data(), 8ull * n_in);
  if (table && n_out) memcpy(table, tb.data(), 8ull * n_in * n_out);
  return HM_OK;
}

extern "C" hm_status hm_bconv(hm_ctx *c, const uint64_t *in, const uint32_t *in_limbs, const uint32_t *in_ids,
                              uint32_t n_in, uint64_t *out, const uint32_t *out_limbs, const uint32_t *out_ids,
                              uint32_t n_out) {
  if (!c) return HM_ERR_ARG;
  if (!in || !out) return fail(c, HM_ERR_ARG, "hm_bconv: null buffer");
  if (n_in == 0 || n_in > HM_BCONV_MAX_IN) return fail(c, HM_ERR_ARG, "hm_bconv: n_in %u not in [1,%d]", n_in, HM_BCONV_MAX_IN);
  if (n_out == 0 || n_out > HM_BCONV_MAX_OUT) return fail(c, HM_ERR_ARG, "hm_bconv: n_out %u not in [1,%d]", n_out, HM_BCONV_MAX_OUT);
  hm_status st;
  if ((st = check_limbs(c, "hm_bconv", in_limbs, n_in)) || (st = check_limbs(c, "hm_bconv", out_limbs, n_out)) ||
      (st = check_mods(c, "hm_bconv", in_ids, n_in)) || (st = check_mods(c, "hm_bconv", out_ids, n_out)))
    return st;
  for (uint32_t i = 0; i < n_in; ++i)
    for (uint32_t t = 0; t < n_out; ++t)
      if (in_ids[i] == out_ids[t]) return fail(c, HM_ERR_ARG, "hm_bconv: modulus %u is in both bases", in_ids[i]);
  HM_HIP(c, hipSetDevice(c->device));
  // conversion tables are cached per (input basis, output basis); built and uploaded on first use
  std::vector<uint32_t> key;
  key.push_back(n_in);
  key.insert(key.end(), in_ids, in_ids + n_in);
  key.insert(key.end(), out_ids, out_ids + n_out);
  auto it = c->bconv_tables.find(key);
  if (it == c->bconv_tables.end()) {
    std::vector<uint64_t> qh(n_in), tb((size_t)n_in * n_out);
    c->P.bconv_consts(in_ids, n_in, out_ids, n_out, qh.data(), tb.data());
    uint64_t *d = nullptr;
    HM_HIP(c, hipMalloc(&d, 8ull * n_in * n_out));
    HM_HIP(c, hipMemcpy(d, tb.data(), 8ull * n_in * n_out, hipMemcpyHostToDevice));
    it = c->bconv_tables.emplace(key, d).first;
  }
  HmBconvArgs a;
  a.in = in; a.out = out; a.table = it->second; a.mods = c->d_mods;
  a.logN = c->P.logN; a.n_in = n_in; a.n_out = n_out;
  a.out_per_block = n_out <= 8 ? n_out : (n_out + 3) / 4;
  for (uint32_t i = 0; i < n_in; ++i) a.in_limb[i] = (uint16_t)limb_at(in_limbs, i);
  for (uint32_t t = 0; t < n_out; ++t) {
    a.out_limb[t] = (uint16_t)limb_at(out_limbs, t);
    a.out_mod[t] = (uint16_t)out_ids[t];
  }
  dim3 grid(c->P.N / 128, (n_out + a.out_per_block - 1) / a.out_per_block);
  hipLaunchKernelGGL(k_bconv, grid, dim3(128), 0, c->stream, a);
  HM_HIP(c, hipGetLastError());
  return HM_OK;
}

extern "C" hm_status hm_fill_uniform(hm_ctx *c, uint64_t *out, const uint32_t *out_limbs, const uint32_t *mod_ids,
                                     uint32_t n, uint64_t seed) {
  if (!c) return HM_ERR_ARG;
  if (!out) return fail(c, HM_ERR_ARG, "hm_fill_uniform: null buffer");
  hm_status st;
  if ((st = check_limbs(c, "hm_fill_uniform", out_limbs, n)) || (st = check_mods(c, "hm_fill_uniform", mod_ids, n))) return st;
  HM_HIP(c, hipSetDevice(c->device));
  for (uint32_t base = 0; base < n; base += HM_MAX_LIMBS) {
    const uint32_t cnt = std::min<uint32_t>(HM_MAX_LIMBS, n - base);
    HmFillArgs a;
    a.out = out; a.mods = c->d_mods; a.seed = seed + base; a.logN = c->P.logN; a.n_limbs = cnt;
    for (uint32_t i = 0; i < cnt; ++i)
      a.limb[i] = HmLimb{0, (uint16_t)limb_at(out_limbs, base + i), (uint16_t)mod_ids[base + i], (uint16_t)i};
    hipLaunchKernelGGL(k_fill, dim3(cnt * (c->P.N / 256)), dim3(256), 0, c->stream, a);
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}
