// tools/bconv_mfma.hip — EXPERIMENT (round 6): the fast base conversion's multiply-accumulate on the int8 matrix pipe.
//
//   out_t[x] = sum_i y_i[x] * w_{i,t} mod q_t      (K4: InsGen::GenBCONV src/InsGen.cpp:263-313, BCONVU src/Components.cpp:268-295)
//
// is a true matrix product per coefficient (n_in x n_out), the one stage of the path that is; it holds a third of the op's VALU instructions
// (k_bconv / k_bconv_col: four v_mad_u64_u32 per (input, output, coefficient) on 30-bit halves) while the matrix pipe idles.  Here the 60-bit
// operands are taken as their eight BYTES, recoded to signed digits (x = sum_a s_a 2^(8a), s_a in [-128, 127]: (x + C) ^ C with
// C = 0x0080808080808080), and one v_mfma_i32_32x32x32_i8 contracts K = 32 = (4 input limbs x 8 bytes) for a tile of 32 coefficients x 32
// rows, a row being (output t, diagonal d = a + b): the table side is Toeplitz-banded on the host (A[(t, d)][(i, a)] = digit d - a of
// w_{i,t} 2^64 mod q_t).  Four MFMAs (16 limbs) leave every lane with the 15 diagonal sums D_0 .. D_14 of ONE (coefficient, output): the C
// layout (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) puts diagonal d = reg of output t = lane >> 5 on a lane when the
// rows are numbered (t, d) -> row (d & 3) + 8 (d >> 2) + 4 t.  The VALU then only recombines: sum_d D_d 2^(8d) (12 multiply-adds with powers of
// two on biased, non-negative sums), one 128-bit subtraction of the bias, one Montgomery reduction — about 40 instructions per (coefficient,
// output) where the split-30 form takes 60 multiply-adds + 30.  Results are bit-identical to hm_bconv_batch (exact integer arithmetic: the
// matrix pipe accumulates int32, |D_d| < 2^21).  north_star says "no MFMA" because the path is not a contraction; this stage is one.
//
// Stand-alone: runs hm_bconv_batch (k_bconv<15>) and this kernel on the same inputs — P conversions of 15 -> 35 limbs at N = 2^16, the shape of
// the ModDown conversion of a batch of 10 hmults (2 keys each) — compares bit for bit and prints both device times.
//   /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I homulator_amd/csrc -I include tools/bconv_mfma.hip -o tools/bconv_mfma \
//       -L homulator_amd/lib -lhomulator_hip -Wl,-rpath,'$ORIGIN/../homulator_amd/lib'
//   tools/bconv_mfma [n_prob = 20] [n_in = 15] [n_out = 35] [blocks of 32 coefficients per wave = 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "homulator_hip.h"
#include "hm_modarith.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// pointers read out of the problem record are generic to the compiler: FLAT loads / stores, which also count on the LDS counter — every
// s_waitcnt for an A fragment then waits for the previous iteration's result stores.  Say that they are global.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const ulonglong2 __attribute__((address_space(1))) *GlobalCV2;
typedef ulonglong2 __attribute__((address_space(1))) *GlobalV2;
#else
typedef const ulonglong2 *GlobalCV2;
typedef ulonglong2 *GlobalV2;
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HM(x) do { if ((x) != HM_OK) { fprintf(stderr, "%s: %s\n", #x, hm_last_error(ctx)); exit(1); } } while (0)

static constexpr uint64_t kRecode = 0x0080808080808080ull;   // x -> (x + C) ^ C: bytes 0..6 become signed digits, byte 7 (< 17) stays
static constexpr int BIAS_LOG = 22;                          // |D_d| <= 16 limbs x 8 byte pairs x 2^14 = 2^21: D_d + 2^22 is positive and below 2^23

struct MfmaProb {
  const uint64_t *in;    // [n_in limbs][N], the limbs at in_limb[]
  uint64_t *out;
  const v4i *afrag;      // device: [pairs][STEPS][64 lanes] 16-byte A fragments (the Toeplitz-banded table)
  const uint64_t *qn;    // device: [n_out] x {q, unused}
  uint32_t n_in, n_out;
  uint32_t in_limb[16], out_limb[64];
};

// one workgroup = WAVES waves sharing the LDS copy of a conversion's A fragments; a wave walks its blocks of 32 coefficients two at a time
// (two independent accumulator chains per A fragment: the four MFMAs of a chain depend on each other); per block pair: the data fragments
// of all STEPS (16 limbs) in registers, then every output pair: A fragments from LDS, 2 x STEPS MFMAs, recombination, reduction, store
// the data fragments of TWO blocks at once: lane c of a k half takes coefficients 2c and 2c + 1 of a 64-coefficient span with ONE 16-byte
// load per limb (8-byte accesses run at half the rate on gfx950); block 0 = the even coefficients, block 1 = the odd ones
template <int STEPS>
__device__ __forceinline__ void load_b2(const MfmaProb &p, size_t N, uint32_t x, int kh, v4i (&bf0)[STEPS], v4i (&bf1)[STEPS]) {
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    uint64_t y0[2], y1[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const uint32_t i = 4 * s + 2 * kh + e;
      ulonglong2 v = {0, 0};
      if (i < p.n_in) v = *(GlobalCV2)(uintptr_t)(p.in + (size_t)p.in_limb[i] * N + x);
      y0[e] = i < p.n_in ? (v.x + kRecode) ^ kRecode : 0;
      y1[e] = i < p.n_in ? (v.y + kRecode) ^ kRecode : 0;
    }
    bf0[s] = v4i{(int)(uint32_t)y0[0], (int)(uint32_t)(y0[0] >> 32), (int)(uint32_t)y0[1], (int)(uint32_t)(y0[1] >> 32)};
    bf1[s] = v4i{(int)(uint32_t)y1[0], (int)(uint32_t)(y1[0] >> 32), (int)(uint32_t)y1[1], (int)(uint32_t)(y1[1] >> 32)};
  }
}
// V = sum_d D'_d 2^(8d) mod 2^128, D'_d = D_d + 2^22 in [0, 2^23): pairs of diagonals in 32-bit words (P_m = D'_2m + 2^8 D'_2m+1 < 2^32), the
// even pairs ARE the four words of one 128-bit number and the odd pairs of another, 16 bits up; then minus the bias, then one Montgomery reduction
__device__ __forceinline__ uint64_t recombine(const v16i &acc, uint64_t klo, uint64_t khi, uint64_t q) {
  uint32_t P[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) P[m] = ((uint32_t)acc[2 * m + 1] << 8) + (uint32_t)acc[2 * m];
  const unsigned __int128 E = (unsigned __int128)((uint64_t)P[0] | ((uint64_t)P[2] << 32)) | ((unsigned __int128)((uint64_t)P[4] | ((uint64_t)P[6] << 32)) << 64);
  const unsigned __int128 O = (unsigned __int128)((uint64_t)P[1] | ((uint64_t)P[3] << 32)) | ((unsigned __int128)((uint64_t)P[5] | ((uint64_t)P[7] << 32)) << 64);
  unsigned __int128 V = E + (O << 16);
  V -= ((unsigned __int128)khi << 64) | klo;
  HmMod m;
  m.q = q;
  m.nqinv = 0;
  return hm_redc_wide<16>(V, m);
}
template <int STEPS, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_bconv_mfma(const MfmaProb *probs, uint32_t logN, uint32_t pairs, uint32_t blocksPerWave) {
  extern __shared__ v4i lds_a[];   // [pairs][STEPS][64] A fragments, then [2 pairs] x {q, out limb}
  const MfmaProb &p = probs[blockIdx.y];
  const size_t N = (size_t)1 << logN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, kh = lane >> 5;
  uint64_t *lds_q = reinterpret_cast<uint64_t *>(lds_a + pairs * STEPS * 64);   // [2 pairs][2]: q_t, out limb of t
  for (uint32_t i = threadIdx.x; i < pairs * STEPS * 64; i += 64 * WAVES) lds_a[i] = p.afrag[i];
  for (uint32_t t = threadIdx.x; t < 2 * pairs; t += 64 * WAVES) { lds_q[2 * t] = t < p.n_out ? p.qn[2 * t] : 1; lds_q[2 * t + 1] = t < p.n_out ? p.out_limb[t] : 0xFFFFFFFFu; }
  __syncthreads();
  v16i bias;
#pragma unroll
  for (int r = 0; r < 16; ++r) bias[r] = 1 << BIAS_LOG;
  // the bias as a 128-bit constant: 2^22 sum_{d < 16} 2^(8d) mod 2^128
  unsigned __int128 K128 = 0;
#pragma unroll
  for (int d = 0; d < 16; ++d)
    if (BIAS_LOG + 8 * d < 128) K128 += (unsigned __int128)1 << (BIAS_LOG + 8 * d);   // (the bias of diagonals 14 and 15 lies above 2^128: nothing to take back)
  const uint64_t klo = (uint64_t)K128, khi = (uint64_t)(K128 >> 64);
  for (uint32_t b = 0; b < blocksPerWave; b += 2) {
    const uint32_t x0 = ((blockIdx.x * WAVES + wave) * blocksPerWave + b) * 32 + 2 * c;   // (blocksPerWave is even and the grid covers N exactly)
    v4i bf0[STEPS], bf1[STEPS];
    load_b2<STEPS>(p, N, x0, kh, bf0, bf1);
    for (uint32_t pr = 0; pr < pairs; ++pr) {
      v16i acc0 = bias, acc1 = bias;
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const v4i af = lds_a[(pr * STEPS + s) * 64 + lane];
#if defined(ABL_NO_MFMA)   // timing-only ablations (wrong results): where the time goes
        acc0[s] += af[0] ^ bf0[s][1]; acc1[s] += af[1] ^ bf1[s][2];
#else
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf1[s], acc1, 0, 0, 0);
#endif
      }
      const uint32_t t = 2 * pr + kh;
      const uint64_t q = lds_q[2 * t], ol = lds_q[2 * t + 1];
#if defined(ABL_NO_RECOMB)
      uint64_t r0 = 0, r1 = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) { r0 ^= (uint32_t)acc0[r]; r1 += (uint32_t)acc1[r]; }
#else
      const uint64_t r0 = recombine(acc0, klo, khi, q), r1 = recombine(acc1, klo, khi, q);
#endif
      if (ol != 0xFFFFFFFFu) {   // (an odd basis: the last pair's second output does not exist)
        ulonglong2 r = {r0, r1};
        *(GlobalV2)(uintptr_t)(p.out + (size_t)ol * N + x0) = r;
      }
    }
  }
}

int main(int argc, char **argv) {
  const uint32_t n_prob = argc > 1 ? atoi(argv[1]) : 20, n_in = argc > 2 ? atoi(argv[2]) : 15, n_out = argc > 3 ? atoi(argv[3]) : 35;
  const uint32_t logN = 16, N = 1u << logN, L = n_out, K = n_in;
  if (n_in > 16 || n_out > 64) { fprintf(stderr, "n_in <= 16, n_out <= 64\n"); return 1; }
  hm_ctx *ctx = nullptr;
  hm_params prm = {logN, L, K, 0, nullptr, nullptr, nullptr};
  if (hm_create(&ctx, &prm) != HM_OK) { fprintf(stderr, "hm_create: %s\n", hm_last_error(nullptr)); return 1; }
  hipStream_t S = (hipStream_t)hm_stream(ctx);
  std::vector<uint32_t> in_ids(n_in), out_ids(n_out);
  for (uint32_t i = 0; i < n_in; ++i) in_ids[i] = L + i;      // the special basis P ...
  for (uint32_t t = 0; t < n_out; ++t) out_ids[t] = t;         // ... to the Q limbs: the ModDown conversion's shape
  std::vector<uint64_t> q(n_out), tb((size_t)n_in * n_out);
  for (uint32_t t = 0; t < n_out; ++t) HM(hm_get_modulus(ctx, out_ids[t], &q[t]));
  HM(hm_bconv_consts(ctx, in_ids.data(), n_in, out_ids.data(), n_out, nullptr, tb.data()));   // table[i][t] = [Q_D / q_i] mod q_t
  // the Toeplitz-banded A fragments: lane l (row rho = l & 31, k half kh = l >> 5) of (pair, step s) holds, for limbs i = 4 s + 2 kh + e (e = 0, 1)
  // and bytes a = 0 .. 7, digit (d - a) of wm_{i,t}, with t = 2 pair + ((rho >> 2) & 1), d = (rho & 3) + 4 (rho >> 3); wm = w 2^64 mod q_t, recoded
  const uint32_t pairs = (n_out + 1) / 2, STEPS = 4;
  std::vector<int8_t> af((size_t)pairs * STEPS * 64 * 16, 0);
  for (uint32_t pr = 0; pr < pairs; ++pr)
    for (uint32_t s = 0; s < STEPS; ++s)
      for (uint32_t l = 0; l < 64; ++l) {
        const uint32_t rho = l & 31, kh = l >> 5, t = 2 * pr + ((rho >> 2) & 1), d = (rho & 3) + 4 * (rho >> 3);
        for (uint32_t j = 0; j < 16; ++j) {
          const uint32_t i = 4 * s + 2 * kh + (j >> 3), a = j & 7;
          int8_t v = 0;
          if (i < n_in && t < n_out && d >= a && d - a <= 7) {
            const uint64_t wm = (uint64_t)((((unsigned __int128)tb[(size_t)i * n_out + t]) << 64) % q[t]);
            const uint64_t rec = (wm + kRecode) ^ kRecode;
            v = (int8_t)(uint8_t)(rec >> (8 * (d - a)));
          }
          af[(((size_t)pr * STEPS + s) * 64 + l) * 16 + j] = v;
        }
      }
  std::vector<uint64_t> qn(2 * n_out);
  for (uint32_t t = 0; t < n_out; ++t) { qn[2 * t] = q[t]; qn[2 * t + 1] = 0; }
  void *d_af = nullptr, *d_qn = nullptr, *d_in = nullptr, *d_ref = nullptr, *d_out = nullptr, *d_probs = nullptr;
  HM(hm_malloc(ctx, af.size(), &d_af));
  HM(hm_malloc(ctx, qn.size() * 8, &d_qn));
  HM(hm_memcpy_h2d(ctx, d_af, af.data(), af.size()));
  HM(hm_memcpy_h2d(ctx, d_qn, qn.data(), qn.size() * 8));
  const size_t LP = (size_t)N * 8;
  HM(hm_malloc(ctx, LP * n_in * n_prob, &d_in));
  HM(hm_malloc(ctx, LP * n_out * n_prob, &d_ref));
  HM(hm_malloc(ctx, LP * n_out * n_prob, &d_out));
  std::vector<uint32_t> fl(n_in * n_prob), fm(n_in * n_prob);
  for (uint32_t k = 0; k < n_in * n_prob; ++k) { fl[k] = k; fm[k] = in_ids[k % n_in]; }
  for (uint32_t base = 0; base < fl.size(); base += 128)
    HM(hm_fill_uniform(ctx, (uint64_t *)d_in, fl.data() + base, fm.data() + base, std::min<uint32_t>(128, (uint32_t)fl.size() - base), 77 + base));
  // the extreme operand on a few coefficients: every input at q_i - 1
  {
    std::vector<uint64_t> edge(8);
    for (uint32_t i = 0; i < n_in; ++i) {
      uint64_t qi; HM(hm_get_modulus(ctx, in_ids[i], &qi));
      for (auto &e : edge) e = qi - 1;
      HM(hm_memcpy_h2d(ctx, (char *)d_in + LP * i, edge.data(), 64));
    }
  }
  std::vector<std::vector<uint32_t>> il(n_prob), ol(n_prob);
  std::vector<hm_bconv_desc> descs(n_prob);
  std::vector<MfmaProb> probs(n_prob);
  for (uint32_t pb = 0; pb < n_prob; ++pb) {
    il[pb].resize(n_in); ol[pb].resize(n_out);
    for (uint32_t i = 0; i < n_in; ++i) il[pb][i] = pb * n_in + i;
    for (uint32_t t = 0; t < n_out; ++t) ol[pb][t] = pb * n_out + t;
    memset(&descs[pb], 0, sizeof descs[pb]);
    descs[pb].in = (const uint64_t *)d_in; descs[pb].in_limbs = il[pb].data(); descs[pb].in_ids = in_ids.data(); descs[pb].n_in = n_in;
    descs[pb].out = (uint64_t *)d_ref; descs[pb].out_limbs = ol[pb].data(); descs[pb].out_ids = out_ids.data(); descs[pb].n_out = n_out;
    MfmaProb &m = probs[pb];
    memset(&m, 0, sizeof m);
    m.in = (const uint64_t *)d_in; m.out = (uint64_t *)d_out; m.afrag = (const v4i *)d_af; m.qn = (const uint64_t *)d_qn; m.n_in = n_in; m.n_out = n_out;
    for (uint32_t i = 0; i < n_in; ++i) m.in_limb[i] = il[pb][i];
    for (uint32_t t = 0; t < n_out; ++t) m.out_limb[t] = ol[pb][t];
  }
  HM(hm_malloc(ctx, sizeof(MfmaProb) * n_prob, &d_probs));
  HM(hm_memcpy_h2d(ctx, d_probs, probs.data(), sizeof(MfmaProb) * n_prob));
  constexpr int WAVES = 8;
  const uint32_t BPW = argc > 4 ? atoi(argv[4]) : 8;   // blocks of 32 coefficients per wave (even)
  const dim3 grid(N / (WAVES * BPW * 32), n_prob), block(64 * WAVES);
  const size_t ldsBytes = (size_t)pairs * STEPS * 64 * 16 + (size_t)pairs * 2 * 16;
  CK(hipFuncSetAttribute((const void *)k_bconv_mfma<4, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_us = [&](auto fn, int iters) {
    for (int i = 0; i < 3; ++i) fn();
    CK(hipEventRecord(e0, S));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, S));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / iters;
  };
  const double ref_us = time_us([&] { HM(hm_bconv_batch(ctx, descs.data(), n_prob)); }, 20);
  const double mfma_us = time_us([&] { hipLaunchKernelGGL((k_bconv_mfma<4, WAVES>), grid, block, ldsBytes, S, (const MfmaProb *)d_probs, logN, pairs, BPW); }, 20);
  CK(hipGetLastError());
  HM(hm_sync(ctx));
  std::vector<uint64_t> a((size_t)N * n_out), b((size_t)N * n_out);
  size_t bad = 0;
  for (uint32_t pb = 0; pb < n_prob; pb += (n_prob > 1 ? n_prob - 1 : 1)) {   // first and last conversion, every limb
    HM(hm_memcpy_d2h(ctx, a.data(), (char *)d_ref + LP * n_out * pb, LP * n_out));
    HM(hm_memcpy_d2h(ctx, b.data(), (char *)d_out + LP * n_out * pb, LP * n_out));
    for (size_t k = 0; k < a.size(); ++k) bad += a[k] != b[k];
    if (bad) { for (size_t k = 0; k < a.size(); ++k) if (a[k] != b[k]) { printf("first mismatch: conversion %u limb %zu x %zu: ref %llx mfma %llx\n", pb, k / N, k % N, (unsigned long long)a[k], (unsigned long long)b[k]); break; } break; }
  }
  printf("bconv %u -> %u limbs, N = 2^%u, %u conversions per launch: k_bconv<%u> (hm_bconv_batch) %.1f us, k_bconv_mfma %.1f us: x%.2f; %s\n", n_in, n_out, logN, n_prob, n_in,
         ref_us, mfma_us, ref_us / mfma_us, bad ? "RESULTS DIFFER" : "bit-identical");
  hm_destroy(ctx);
  return bad ? 2 : 0;
}
