/*
 * homulator_hip.h — C ABI of the MI355X (gfx950) execution backend for Homulator's FHE datapath.
 *
 * The reference (FHE-ACCELE/Homulator) has no FFI or plugin interface: its execution boundary is the
 * C++ seam between the graph builders (Operation / InsGen / Driver) and the cycle model (Arch).
 * Each entry point below states which reference interface it replaces.  Plain C types only; device
 * buffers are raw device pointers (`uint64_t *` to [limb][N] words, limb-major, 8-byte words, values
 * fully reduced in [0, q) at every call boundary); no exceptions cross the boundary: every call returns
 * hm_status (0 = ok) and hm_last_error() returns the message.
 *
 * A context is bound to one GPU and one HIP stream and is not thread-safe.  All hm_* compute calls are
 * asynchronous on the context's stream; hm_sync() waits.
 */
#ifndef HOMULATOR_HIP_H
#define HOMULATOR_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hm_ctx hm_ctx;
typedef int hm_status;
enum { HM_OK = 0, HM_ERR_ARG = 1, HM_ERR_HIP = 2, HM_ERR_UNSUPPORTED = 3, HM_ERR_COMM = 4 };

/* Parameter set.  N = 2^logN (13..17), L Q-primes, K special primes (the reference sizes the special
 * basis as exactly alpha limbs, src/Operation.cpp:160,193,297-304).  q == NULL selects the default
 * chain: the L+K largest primes below 2^60 that are 1 mod 2^32 (hence 1 mod 2N), descending, first L = Q,
 * next K = P; psi == NULL selects the smallest primitive 2N-th root of each prime.  Explicit moduli must be
 * distinct primes q = h 2^32 + 1 below 2^60 as well: the transforms reduce word-wise in Montgomery form, and
 * q^-1 = 1 mod 2^32 makes a reduction step one multiply (six 32-bit multiplies per butterfly instead of nine);
 * hm_create fails with the modulus in hm_last_error otherwise.  "mod id" m: m < L -> q[m],
 * m >= L -> p[m-L].  Replaces: Arch::Arch(Config*) include/Arch.h:154 / src/Arch.cpp:8-168 (the
 * construction of the execution resources from N / cluster). */
typedef struct hm_params {
  uint32_t logN, L, K;
  int32_t device;           /* HIP device ordinal */
  const uint64_t *q, *p;    /* optional explicit moduli */
  const uint64_t *psi;      /* optional explicit roots, [L+K] */
} hm_params;

hm_status hm_create(hm_ctx **ctx, const hm_params *params);
void hm_destroy(hm_ctx *ctx);
const char *hm_last_error(const hm_ctx *ctx); /* ctx may be NULL: last error of hm_create */
const char *hm_version(void);

hm_status hm_get_modulus(const hm_ctx *ctx, uint32_t mod_id, uint64_t *q);
hm_status hm_get_psi(const hm_ctx *ctx, uint32_t mod_id, uint64_t *psi);

/* Device memory (replaces the scratchpad/HBM address space of include/mem.h:465-651 and the
 * AddrManage line addresses of include/Addr.h:29-66: one reference "limb" = N words here). */
hm_status hm_malloc(hm_ctx *ctx, size_t bytes, void **dptr);
hm_status hm_free(hm_ctx *ctx, void *dptr);
hm_status hm_memcpy_h2d(hm_ctx *ctx, void *dst, const void *src, size_t bytes);
hm_status hm_memcpy_d2h(hm_ctx *ctx, void *dst, const void *src, size_t bytes);
hm_status hm_memcpy_d2d(hm_ctx *ctx, void *dst, const void *src, size_t bytes);
hm_status hm_sync(hm_ctx *ctx);
void *hm_stream(hm_ctx *ctx); /* the context's hipStream_t */
/* work enqueued on `ctx` after this call starts only when everything enqueued so far on `producer` has finished
 * (an event on the producer's stream; no host synchronisation): chains ops that live in different contexts */
hm_status hm_wait_for(hm_ctx *ctx, hm_ctx *producer);

/* Limb lists: every compute call takes n limb-polys; operand X of limb i lives at
 * X_base + X_limbs[i] * N (X_limbs == NULL means 0,1,..,n-1); mod_ids[i] selects the modulus. */

/* K1 — NTT / INTT.  Replaces Arch::issueIns(cluster, "NTT"/"INTT", group) include/Arch.h:276 for
 * the instructions of InsGen::GenNTT (src/InsGen.cpp:17-44): one call = one stage of limb-NTTs.
 * inverse != 0: result is multiplied by N^-1 and, if scale != NULL, by scale[i] (a host array of n
 * residues mod the limb's modulus): this fuses the "x q_hat^-1" EWE stage that follows every INTT in
 * the key switch (src/Operation.cpp:104-135, 447-487).  in == out is allowed. */
hm_status hm_ntt(hm_ctx *ctx, const uint64_t *in, const uint32_t *in_limbs, uint64_t *out,
                 const uint32_t *out_limbs, const uint32_t *mod_ids, uint32_t n, int inverse,
                 const uint64_t *scale);

/* K1 fused with the element-wise stage that consumes it: out_i = (minuend_i - NTT(in_i)) * k[i] (+ addend_i).
 * One call = ModDowNTT + ModDownSub (+ the final add), src/Operation.cpp:521-590, 967-1005, or Rescale_NTT +
 * Rescale_SUB + Rescale_Mul, :806-910: the transformed limbs never travel to HBM.  `out` also serves as the
 * scratch of the first pass, so it must not alias `in`, `minuend` or `addend`. */
hm_status hm_ntt_sub_scale(hm_ctx *ctx, const uint64_t *in, const uint32_t *in_limbs, const uint64_t *minuend,
                           const uint32_t *minuend_limbs, const uint64_t *addend, const uint32_t *addend_limbs,
                           uint64_t *out, const uint32_t *out_limbs, const uint32_t *mod_ids, uint32_t n,
                           const uint64_t *k);

/* The same with a linear prologue and a constant on the addend — ModDown and rescale merged into ONE transform per
 * limb (both are linear in the coefficient domain):
 *     x   = NTT(in + mix_k * mix)                       mix, mix_k optional (both or neither)
 *     out = (minuend - x) * k  [+ addend [* addend_k]]   addend optional; addend_k optional (NULL = 1)
 * With in = the P->Q conversion, mix = r = INTT(last limb of the key-switch sum), mix_k = P, k = (P q_last)^-1,
 * addend = the tensor part, addend_k = q_last^-1 this is ModDowNTT + ModDownSub + add + Rescale_NTT + Rescale_SUB +
 * Rescale_Mul of one limb (src/Operation.cpp:521-590, 967-1005, 806-910) in one pass over HBM.  All constants are
 * host arrays of n residues mod the limb's modulus.  `out` must not alias `in`, `mix`, `minuend` or `addend`.
 * addend_limbs[i] == HM_NO_LIMB drops the addend for limb-poly i only (both this call and hm_ntt_sub_scale). */
#define HM_NO_LIMB 0xFFFFFFFFu
struct hm_bconv_desc;
typedef struct hm_ntt_fused_desc {
  const uint64_t *in;      const uint32_t *in_limbs;
  const uint64_t *mix;     const uint32_t *mix_limbs;     const uint64_t *mix_k;
  const uint64_t *minuend; const uint32_t *minuend_limbs;
  const uint64_t *addend;  const uint32_t *addend_limbs;  const uint64_t *addend_k;
  uint64_t *out;           const uint32_t *out_limbs;
  const uint32_t *mod_ids; uint32_t n;
  const uint64_t *k;
  /* optional (NULL / 0: none; round 4): `in` of some or all limb-polys is a base conversion that has not been computed yet —
   * ModDown_BCONV_Key(k) + ModDowNTT + ModDownSub [+ the rescale] (src/Operation.cpp:489-590) in one call.  conv[j] describes
   * conversion j exactly as for hm_bconv_batch, except that conv[j].out must be `out` and conv[j].out_limbs[t] the OUTPUT limb
   * (out_limbs[i]) of the limb-poly i its output t feeds: the conversion runs inside the first pass of that transform (with the
   * mix prologue of limb-poly i), so ModdownBConvOut_Key(k) is never written to HBM or read back; `in` / in_limbs of the covered
   * limb-polys are ignored, the others are transformed as usual.  N = 2^15 or 2^16 and n_in <= 15 (HM_ERR_UNSUPPORTED otherwise).
   * Bit-identical to hm_bconv_batch + the call without conv. */
  const struct hm_bconv_desc *conv; uint32_t n_conv;
  /* optional (NULL: none; round 6): addend_galois[i] = g > 1 reads the addend of limb-poly i through the automorphism X -> X^g (evaluation form, as
   * hm_automorph): out = (minuend - NTT(in)) * k + automorph_g(addend) [* addend_k] — hrotate's final add takes AUTOOutput(0) this way and
   * AUTO_Key(0) (src/Operation.cpp hrotate, InsGen::GenAUTO src/InsGen.cpp:46-71) is never written; 0 / 1 = the addend as stored.  g odd, below 2N.
   * Needs an addend; not with mix or conv (HM_ERR_UNSUPPORTED).  Bit-identical to hm_automorph + the call without it. */
  const uint32_t *addend_galois;
} hm_ntt_fused_desc;
hm_status hm_ntt_mix_sub_scale(hm_ctx *ctx, const hm_ntt_fused_desc *desc);

/* K2 — automorphism X -> X^galois in evaluation form.  Replaces issueIns(..., "AUTO", ...) for
 * InsGen::GenAUTO (src/InsGen.cpp:46-71).  in must not alias out. */
hm_status hm_automorph(hm_ctx *ctx, const uint64_t *in, const uint32_t *in_limbs, uint64_t *out,
                       const uint32_t *out_limbs, uint32_t n, uint32_t galois);

/* K3 — element-wise engine.  Replaces issueIns(..., "MULT", ...) for InsGen::GenEWE
 * (src/InsGen.cpp:77-125); upstream has no opcode, the stages of src/Operation.cpp need these: */
enum hm_ewe_op {
  HM_OP_MUL = 0,           /* out = a*b            TensorCompute D0/D2 :634-660, 712-738; PMULT */
  HM_OP_MAC2 = 1,          /* out = a*b + c*d      TensorCompute D1 :673-699; inner product :355-411 */
  HM_OP_MAC_ADD = 2,       /* out = a*b + c        inner product, later groups :355-411 */
  HM_OP_ADD = 3,           /* out = a + c          HMULT add :967-1005, HADD, PADD, HROTATE add :1339-1357 */
  HM_OP_SUB = 4,           /* out = a - c          Rescale_SUB :831-875 */
  HM_OP_MUL_CONST = 5,     /* out = a*k[i]         ModUp_DecompOut :104-135, ModDownBConvStep1 :447-487, Rescale_Mul :877-910 */
  HM_OP_SUB_SCALE = 6,     /* out = (a - c)*k[i]   KeySwitchFinalOutput :548-590 */
  HM_OP_COPY = 7,          /* out = a */
  HM_OP_SUB_SCALE_ADD = 8  /* out = (a - c)*k[i] + d   fused ModDownSub + final add */
};
hm_status hm_ewe(hm_ctx *ctx, int op, const uint64_t *a, const uint32_t *a_limbs, const uint64_t *b,
                 const uint32_t *b_limbs, const uint64_t *c, const uint32_t *c_limbs, const uint64_t *d,
                 const uint32_t *d_limbs, uint64_t *out, const uint32_t *out_limbs,
                 const uint32_t *mod_ids, uint32_t n, const uint64_t *k);

/* K3, tensor product in one pass over the four input polynomials: o0 = a*b, o1 = a*d + c*b, o2 = c*d
 * (TensorCompute::computeD0/D1/D2, src/Operation.cpp:624-739 with a = c00, b = c10, c = c01, d = c11). */
hm_status hm_tensor(hm_ctx *ctx, const uint64_t *a, const uint32_t *a_limbs, const uint64_t *b, const uint32_t *b_limbs,
                    const uint64_t *c, const uint32_t *c_limbs, const uint64_t *d, const uint32_t *d_limbs,
                    uint64_t *o0, const uint32_t *o0_limbs, uint64_t *o1, const uint32_t *o1_limbs, uint64_t *o2,
                    const uint32_t *o2_limbs, const uint32_t *mod_ids, uint32_t n);

/* K5 — inner product with the evaluation key in one pass: for limb i, out[i][k] = sum_{j < n_terms}
 * x[i][j] * y[i][k][j], k < n_out (n_terms <= 4 digits, n_out <= 2 keys).  Limb lists are row-major:
 * x_limbs[i * n_terms + j], y_limbs[(i * n_out + k) * n_terms + j], out_limbs[i * n_out + k].  Replaces
 * issueIns(cluster, h, w, group, hpip = true) include/Arch.h:277 for InsGen::GenHPIP (src/InsGen.cpp:356-406), and
 * the beta-1 EWE MAC groups per key of KeySwitch::InnerProduceOperation (src/Operation.cpp:294-414): every ext limb is
 * read once for both keys and each output is reduced once. */
hm_status hm_inner_product(hm_ctx *ctx, const uint64_t *x, const uint32_t *x_limbs, const uint64_t *y,
                           const uint32_t *y_limbs, uint64_t *out, const uint32_t *out_limbs,
                           const uint32_t *mod_ids, uint32_t n, uint32_t n_terms, uint32_t n_out);
/* The same with its arguments in one record (round 6) and x_galois = g > 1: the x operands are read through the automorphism X -> X^g (as
 * hm_automorph would have stored them; 0 / 1: as stored) — a one-digit key switch of hrotate, whose Q limbs multiply the rotated c1 itself with the
 * key, then needs no AUTO_Key(1) launch (InsGen::GenAUTO src/InsGen.cpp:46-71).  Same reference interface as hm_inner_product. */
typedef struct hm_ip_desc {
  const uint64_t *x;  const uint32_t *x_limbs;
  const uint64_t *y;  const uint32_t *y_limbs;
  uint64_t *out;      const uint32_t *out_limbs;
  const uint32_t *mod_ids;
  uint32_t n, n_terms, n_out;
  uint32_t x_galois;
} hm_ip_desc;
hm_status hm_inner_product_ex(hm_ctx *ctx, const hm_ip_desc *desc);

/* K1 x K5 — the HPIP unit as a fused NTT-epilogue x evaluation-key MAC: for extended limb i,
 *     out[i][k] = sum_{j < n_terms} X_j[i] * y[i][k][j],   X_j[i] = NTT(x[i][j]) if x_is_coeff[i][j] else x[i][j]
 * i.e. ModUp_NTT_(j) + InnerProOut_(.)_Key<k> of src/Operation.cpp:190-414 in one call: per extended limb the n_terms forward
 * transforms run back to back and both keys accumulate in registers, so the extended digits (NTTOut_beta(j)) are never
 * written to HBM or read back.  Replaces issueIns(cluster, h, w, group, hpip = true) include/Arch.h:277 for
 * InsGen::GenHPIP (src/InsGen.cpp:356-406) together with the NTT stage that feeds it (HPIP: src/Components.cpp:571-668).
 * Row-major lists: x_limbs / x_is_coeff / hand_limbs [i * n_terms + j], y_limbs [(i * n_out + k) * n_terms + j],
 * out_limbs [i * n_out + k].  x_is_coeff[i][j] != 0: x[i][j] is in coefficient form (a converted limb) and is transformed;
 * hand + hand_limbs[i][j] * N is N words of scratch for its first pass (contents undefined afterwards); 0: x[i][j] is already
 * in evaluation form (a digit's own limbs); 2 (round 4): transformed as for 1, but the first pass has already been run into the hand-off
 * limb by the caller (hm_bconv_col, possibly on another rank: see hm_colslices_to_limbs): only the second pass runs here.  `out` must not alias x, hand or y.  Bit-identical to hm_ntt + hm_inner_product. */
struct hm_bconv_desc;
typedef struct hm_ntt_ip_desc {
  const uint64_t *x;    const uint32_t *x_limbs;   const uint8_t *x_is_coeff;
  uint64_t *hand;       const uint32_t *hand_limbs;
  const uint64_t *y;    const uint32_t *y_limbs;
  uint64_t *out;        const uint32_t *out_limbs;
  const uint32_t *mod_ids;
  uint32_t n, n_terms, n_out;
  /* optional (NULL / 0: none): the transformed digits are base conversions that have not been computed yet —
   * ModUp_BCONV_(j) + ModUp_NTT_(j) + the inner product (src/Operation.cpp:137-414) in one call.  conv[k] describes conversion k
   * exactly as for hm_bconv_batch, except that conv[k].out + conv[k].out_limbs[t] * N must be the hand-off limb
   * (hand + hand_limbs[i][j] * N) of the (limb, digit) its output t feeds: the conversion runs INSIDE the first pass of that
   * transform (the workgroup of an output limb's column tile converts its own coefficients from the input tiles), so the
   * converted limb-polys (BConvOut_(j)) are never written to HBM or read back; x / x_limbs of those digits are ignored.
   * The conversions need not cover every transformed digit: a transformed (limb, digit) whose hand-off limb is not an output of
   * any conv[k] is read from x / x_limbs as usual (one call may mix both kinds, e.g. a ModUp whose first digit is wider than the
   * fused conversion admits).
   * N = 2^15 or 2^16 and n_in <= 15 (HM_ERR_UNSUPPORTED otherwise: convert with hm_bconv_batch first).  Bit-identical to
   * hm_bconv_batch + hm_ntt_inner_product. */
  const struct hm_bconv_desc *conv; uint32_t n_conv;
  /* optional (NULL: none; round 5): out_inverse[i] != 0 — the outputs of limb i are only ever read by an inverse transform (the special
   * limbs of the key-switch sum: InnerProOut -> ModDownINTTOut_Key(k), src/Operation.cpp:294-445).  The kernel then runs the FIRST pass
   * of that inverse transform on its accumulators (the ROW pass over the 16 rows the workgroup already owns) and stores the pass's
   * hand-off at out + out_limbs[i][k] * N instead of the evaluation-form sum; hm_ntt_second_pass(.., inverse = 1, ..) on those limbs
   * finishes the transform.  The sums themselves never reach HBM.  N = 2^16 (HM_ERR_UNSUPPORTED otherwise).  Bit-identical to the call
   * without it followed by hm_ntt(.., inverse = 1, ..). */
  const uint8_t *out_inverse;
  /* optional (0: none; round 6): x_galois = g > 1 reads the digits that arrive in EVALUATION form (x_is_coeff == 0: a limb's own digit) through the
   * automorphism X -> X^g, as hm_automorph would have stored them: with hm_ntt_desc.in_galois on the ModUp INTT, hrotate's rotated c1 (AUTO_Key(1),
   * InsGen::GenAUTO src/InsGen.cpp:46-71) is never written.  g odd, below 2N.  Bit-identical to hm_automorph + the call without it. */
  uint32_t x_galois;
} hm_ntt_ip_desc;
hm_status hm_ntt_inner_product(hm_ctx *ctx, const hm_ntt_ip_desc *desc);
/* K1, second half: the last pass of a transform whose first pass has already been run into `buf` by another call — the inverse ROW pass of
 * hm_ntt_inner_product's out_inverse limbs (inverse != 0: the COL pass, x N^-1 [x scale[i]]) or the first pass of hm_bconv_col
 * (inverse == 0: the ROW pass).  In place on buf + limbs[i] * N.  Same reference interface as hm_ntt: Arch::issueIns(cluster, "NTT" / "INTT",
 * group) include/Arch.h:276 for the second half of InsGen::GenNTT's instructions (src/InsGen.cpp:17-44; NTTU stages 9-16,
 * src/Components.cpp:380-436). */
hm_status hm_ntt_second_pass(hm_ctx *ctx, uint64_t *buf, const uint32_t *limbs, const uint32_t *mod_ids, uint32_t n, int inverse,
                             const uint64_t *scale);
/* K1 with its options in one record (round 5): hm_ntt (second_pass_only == 0) or hm_ntt_second_pass (!= 0; `in` is ignored), plus
 * out_packed (inverse only; NULL: none): out_packed[i] != 0 stores limb i in the split-30 packed form that the base conversions take with
 * hm_bconv_desc.in_packed — for the inverse transforms whose outputs feed nothing but base conversions (ModUp_INTT + ModUp_DecompOut,
 * ModDownINTTOut + ModDownBConvStep1: src/Operation.cpp:72-135, 417-487).  Same reference interface as hm_ntt. */
typedef struct hm_ntt_desc {
  const uint64_t *in;  const uint32_t *in_limbs;
  uint64_t *out;       const uint32_t *out_limbs;
  const uint32_t *mod_ids; uint32_t n;
  int inverse;         const uint64_t *scale;
  int second_pass_only;
  const uint8_t *out_packed;
  /* optional (NULL: none; round 6; inverse only, not with second_pass_only): in_galois[i] = g > 1 reads limb-poly i of `in` through the automorphism
   * X -> X^g: out_i = INTT(automorph_g(in_i)) — AUTO_Key(1) + ModUp_INTT of hrotate in one pass over HBM (the index map takes aligned blocks to
   * aligned blocks, so the transform's own 16-byte loads serve).  0 / 1 = as stored.  A limb-poly read this way must not be a limb-poly the same call writes
   * (neither in place nor another entry's output: HM_ERR_ARG). */
  const uint32_t *in_galois;
} hm_ntt_desc;
hm_status hm_ntt_ex(hm_ctx *ctx, const hm_ntt_desc *desc);

/* K4 — fast base conversion, matrix step: out_t = sum_i in_i * [Q_D / q_i]_t mod t for the input
 * basis in_ids (n_in <= 32) and output basis out_ids (n_out <= 64).  `in` must already hold
 * y_i = x_i * [(Q_D/q_i)^-1]_{q_i} (hm_ntt's scale or HM_OP_MUL_CONST with hm_bconv_consts).
 * Replaces issueIns(cluster, h, w, group, ...) include/Arch.h:277 for InsGen::GenBCONV
 * (src/InsGen.cpp:263-313; stages src/Operation.cpp:137-188, 489-519). */
hm_status hm_bconv(hm_ctx *ctx, const uint64_t *in, const uint32_t *in_limbs, const uint32_t *in_ids,
                   uint32_t n_in, uint64_t *out, const uint32_t *out_limbs, const uint32_t *out_ids,
                   uint32_t n_out);
/* several independent conversions in ONE launch (the beta digits of a ModUp, the two keys of a ModDown:
 * src/Operation.cpp:31-35 loops over the digits, :489-519 over the keys) */
typedef struct hm_bconv_desc hm_bconv_desc;
struct hm_bconv_desc {
  const uint64_t *in; const uint32_t *in_limbs; const uint32_t *in_ids; uint32_t n_in;
  uint64_t *out; const uint32_t *out_limbs; const uint32_t *out_ids; uint32_t n_out;
  uint32_t log_len; /* coefficients per limb in `in`/`out` = 2^log_len; 0 = N.  N/world for coefficient slices */
  /* optional epilogue (round 4; sub_from == NULL: none): out_t = (sub_from_t - conv_t) * sub_k[t] [+ add_t] instead of conv_t — the
   * rescale residue r = (INTT(ip_last) - conv_last) * P^-1 + INTT(d_last) of the merged ModDown + rescale (src/Operation.cpp:521-590,
   * 806-822) formed by the conversion kernel itself: one launch less than conversion + HM_OP_SUB_SCALE_ADD.  sub_k: host array of n_out
   * residues; add may be NULL.  hm_bconv_batch only (not the conversions of hm_ntt_ip_desc / hm_ntt_fused_desc). */
  const uint64_t *sub_from; const uint32_t *sub_from_limbs; const uint64_t *add; const uint32_t *add_limbs; const uint64_t *sub_k;
  /* round 5: != 0 — the inputs are stored in the split-30 packed form (x mod 2^30) | ((x >> 30) << 32), as hm_ntt_ex writes them for limbs
   * with out_packed set: the conversion multiplies the two 30-bit halves of every input anyway, and a value that is stored split is not
   * shifted and masked again by each of the workgroups that read it (one per pair of output limbs).  All three conversion entry points
   * (hm_bconv_batch, hm_bconv_col, the conv lists of hm_ntt_ip_desc / hm_ntt_fused_desc). */
  uint32_t in_packed;
};
hm_status hm_bconv_batch(hm_ctx *ctx, const hm_bconv_desc *descs, uint32_t n_desc);
/* K4 + first pass of K1 in one kernel, as a call of its own (round 4): every output of every conversion is converted AND taken through the
 * COL pass of its forward transform; descs[k].out + out_limbs[t] * N receives the first pass's hand-off (what hm_ntt_inner_product reads
 * for a digit with x_is_coeff = 2), the converted limb-polys never exist.  tile0 / n_tiles: the column tiles (16 columns x2 of index
 * i = x1 * 256 + x2 each) to work on; n_tiles = 0: all N / 4096 of them; a power of two that divides tile0.  A rank of a sharded run passes
 * its column slice (tile0 = rank * 16 / world, n_tiles = 16 / world at N = 2^16).  N = 2^15 or 2^16, n_in <= 15.  Same reference
 * interface as hm_bconv_batch + hm_ntt: issueIns(cluster, h, w, group, ...) include/Arch.h:277 for InsGen::GenBCONV (src/InsGen.cpp:263-313)
 * and issueIns(cluster, name, group) include/Arch.h:276 for the first half of InsGen::GenNTT (src/InsGen.cpp:17-44). */
hm_status hm_bconv_col(hm_ctx *ctx, const hm_bconv_desc *descs, uint32_t n_desc, uint32_t tile0, uint32_t n_tiles);
/* host-side constants of a conversion: qhat_inv[n_in], table[n_in][n_out] (either may be NULL) */
hm_status hm_bconv_consts(hm_ctx *ctx, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids,
                          uint32_t n_out, uint64_t *qhat_inv, uint64_t *table);

/* ---- multi-GPU: limb sharding + the one exchange of the path ----------------------------
 * One process per GPU.  Limb-polys are sharded by modulus over the `world` ranks (the reference places limb l
 * on cluster l % cluster, include/Driver.h:158,178); only base conversion mixes limbs, so around it the data
 * goes limb-sharded -> coefficient-sharded -> limb-sharded with two all-to-alls (the reference's only
 * "communication" is the modelled inter-cluster NoC fetch, src/mem.cpp:78-100).
 * Transport: RCCL over xGMI (hm_comm_init_rccl; the 128-byte unique id comes from hm_comm_unique_id on rank 0
 * and is distributed by the caller, e.g. through torch.distributed), or a caller-supplied exchange function
 * (hm_comm_init_external) used by the tests to run several ranks over gloo on one GPU. */
typedef int (*hm_exchange_fn)(void *user, const void *send_dev, const size_t *send_off, const size_t *send_bytes,
                              void *recv_dev, const size_t *recv_off, const size_t *recv_bytes);
/* ^ per peer p: send_bytes[p] bytes at send_dev + send_off[p] go to rank p; recv_bytes[p] bytes from rank p land at
 *   recv_dev + recv_off[p]; the entries of the calling rank itself are 0.  Returns 0 on success. */
/* hm_comm_init_rccl and hm_comm_init_external are COLLECTIVE: every rank of the communicator calls them at the same time (ncclCommInitRank is
 * collective anyway; since round 6 both also carry one word from every rank to every peer over the new communicator — the
 * "replicate_split_bytes" option, which decides the shape of hm_replicate_limbs' collective and must agree on all ranks: a mismatch fails
 * the call with HM_ERR_COMM on every rank instead of hanging the first replicate). */
hm_status hm_comm_unique_id(void *out128);
hm_status hm_comm_init_rccl(hm_ctx *ctx, int rank, int world, const void *unique_id128);
hm_status hm_comm_init_external(hm_ctx *ctx, int rank, int world, hm_exchange_fn fn, void *user);
hm_status hm_comm_info(const hm_ctx *ctx, int *rank, int *world);
/* rows of the slice layout: slices are grouped by owner rank, in list order inside an owner */
hm_status hm_slice_rows(const uint32_t *owners, uint32_t n, uint32_t world, uint32_t *rows);
/* n limb-polys, owners[i] = rank that holds limb i in `buf` at index limbs[i] (other entries of limbs[] are
 * ignored on this rank).  Afterwards slices[rows[i]][0 .. N/world) = coefficients [rank*N/world, ...) of limb i,
 * for ALL n limbs.  slices: device, n * N/world words. */
hm_status hm_limbs_to_slices(hm_ctx *ctx, const uint64_t *buf, const uint32_t *limbs, const uint32_t *owners,
                             uint32_t n, uint64_t *slices);
/* the reverse: every rank holds its coefficient slice of all n limbs; afterwards the owner of limb i holds the
 * whole limb at buf + limbs[i] * N. */
hm_status hm_slices_to_limbs(hm_ctx *ctx, const uint64_t *slices, uint64_t *buf, const uint32_t *limbs,
                             const uint32_t *owners, uint32_t n);
/* The same two exchanges in the TRANSPOSED domain (round 4; same bytes on the wire): with i = x1 * 256 + x2, rank p's slice of a limb-poly
 * is the column block x2 in [p * 256 / world, (p + 1) * 256 / world) of every row x1.  The first pass of a forward transform (length N / 256
 * over x1) is local to a column, so the slice holder runs base conversion AND first pass on its columns (hm_bconv_col) and the exchange
 * back carries the first pass's hand-off: the fused kernels of the one-GPU plan serve the sharded one, first-pass work is spread over all
 * ranks and the limb owner only runs hm_ntt_inner_product's second pass (x_is_coeff = 2).  `slices`: n rows of N words in the limb-poly
 * layout (row rows[i] of hm_slice_rows; only this rank's columns are valid).  world <= 16.  Reference: the placement of every unit by
 * limb % cluster, HPIP included (include/Driver.h:155-246), and its one communication, the inter-cluster fetch (src/mem.cpp:78-100). */
hm_status hm_limbs_to_colslices(hm_ctx *ctx, const uint64_t *buf, const uint32_t *limbs, const uint32_t *owners,
                                uint32_t n, uint64_t *slices);
hm_status hm_colslices_to_limbs(hm_ctx *ctx, const uint64_t *slices, uint64_t *buf, const uint32_t *limbs,
                                const uint32_t *owners, uint32_t n);

/* Exchange / compute overlap inside ONE op ("overlap digit j+1's exchange with digit j's NTT"; upstream's digit loop
 * src/Operation.cpp:31-35, its only communication src/mem.cpp:78-100).  hm_exchange_stream(ctx, 1): from now on the three exchange
 * calls below run on a second stream of the context — each starts once everything enqueued on the compute stream BEFORE the call
 * has finished, and the compute stream does not wait for it.  hm_exchange_mark(ctx, slot) marks the end of the exchanges issued
 * so far; hm_exchange_wait(ctx, slot) makes the compute stream wait for that mark (slot < 256).  Exchanges keep their issue order
 * among themselves (one stream, one staging buffer), so every rank still enters the collectives in the same order.  With an
 * external transport the exchange stream is synchronised on the host inside the call (correct, no overlap). */
hm_status hm_exchange_stream(hm_ctx *ctx, int enable);
hm_status hm_exchange_mark(hm_ctx *ctx, uint32_t slot);
hm_status hm_exchange_wait(hm_ctx *ctx, uint32_t slot);

/* every rank ends up with a full copy of the n limbs (owner -> everybody).  The one place the hmult path needs
 * it is the rescale: r = INTT(x_last) lives on one rank and every other limb's NTT reads it (src/Operation.cpp:
 * 806-822). */
hm_status hm_replicate_limbs(hm_ctx *ctx, uint64_t *buf, const uint32_t *limbs, const uint32_t *owners, uint32_t n);

/* Synthetic data: limb i = counter-based SplitMix64 stream (seed + i), uniform in
 * [0, q).  Bench/test tooling; the reference has no data at all. */
hm_status hm_fill_uniform(hm_ctx *ctx, uint64_t *out, const uint32_t *out_limbs, const uint32_t *mod_ids,
                          uint32_t n, uint64_t seed);

/* HIP graphs: record everything enqueued on the context's stream between begin and end into a graph and replay it
 * with one launch (the launch-bound inner loop of an op is a fixed sequence of kernels).  Calls that allocate
 * (first use of a base-conversion table) must have run once before the capture.  Exchanges through an external
 * transport cannot be captured. */
typedef struct hm_graph hm_graph;
hm_status hm_capture_begin(hm_ctx *ctx);
hm_status hm_capture_end(hm_ctx *ctx, hm_graph **graph);
hm_status hm_graph_launch(hm_ctx *ctx, hm_graph *graph);
/* Destroy a context's graphs BEFORE hm_destroy(ctx): a graph's kernel nodes hold device addresses of the context's launch
 * tables.  (hm_destroy detaches graphs that are still alive, so a late hm_graph_destroy is safe; replaying one is not.) */
void hm_graph_destroy(hm_graph *graph);

/* Execution options of a context (A/B measurements, tests; every option has a working default):
 *   "ntt_fused_small"  transform launches of at most this many limb-poly entries (default 96, N = 2^16) run both passes in
 *                ONE launch: the workgroups of a limb-poly meet at a counter in their XCD's L2 between the passes
 *                (k_ntt_fused8; 2-5 us faster than two kernels up to ~100 limb-polys); 0 switches it off.  hm_create switches it off
 *                by itself when an XCD cannot hold the 16 workgroups of a limb-poly at once (counter "ntt_fused_slots_per_xcd"),
 *                and a rendezvous that times out switches it off for the rest of the context's life (HM_ERR_DEVICE on the next
 *                synchronising call; graphs that hold such launches are refused from then on).  Env HOMULATOR_NTT_FUSED_SMALL.
 *   "ntt_fused_test_spread"  test hook: one-launch transforms take the agent-scope path of their rendezvous, as if the
 *                dispatcher had spread every limb-poly over two XCDs (slow; results unchanged; counted by "ntt_cross_xcd").
 *   "ntt_fused_test_timeout"  test hook: tile 0 of every limb-poly withholds its arrival and the waiters spin briefly: every
 *                one-launch transform times out (the error path above, on the GPU).
 *   "ntt_small_limbs"  two-kernel transform launches (single passes; "ntt_fused_small" 0) of at most this many limb-poly entries
 *                (default 64, N = 2^16) use the small-launch geometry (512-thread workgroups, 8 coefficients per thread);
 *                0 switches it off.  Env HOMULATOR_NTT_SMALL_LIMBS.
 *   "ntt_small_mode"   which passes of such a launch use it: bit 0 = COL, bit 1 = ROW (default 3; the hand-off is the same).
 *   "nip_small_limbs"  hm_ntt_inner_product launches of at most this many limb records (default 64, N = 2^16) run in the same
 *                small-launch geometry (k_ntt_row_ip8); 0 switches it off.  Env HOMULATOR_NIP_SMALL.
 *   "bconv_col_outs"   output limbs per workgroup of the fused conversion + first pass: 0 = by launch size (default), 1, 2.
 *                Env HOMULATOR_BCOL_OUTS.
 *   "bconv_col_merge"  small calls of the fused conversion (at most 4 096 workgroups, one output per workgroup): digits of different width run
 *                ONE launch of the widest digit's kernel, the narrower ones with zero table columns for the inputs they lack (default 1: two
 *                launches that each leave the chip part empty become one; 0 = one launch per digit width).  Env HOMULATOR_BCOL_MERGE.
 *   "replicate_split_bytes"  hm_replicate_limbs of a list with ONE owner and at least this many bytes, on >= 4 ranks, runs as
 *                scatter + exchange of chunks (every link carries 2 / (W - 1) of the list); default 2 MiB, 0 = never.  Every
 *                rank of a communicator must use the same value.  Env HOMULATOR_REPLICATE_SPLIT.
 * Counters:
 *   "arith"          the arithmetic back-end serving this context: 0 = word-wise Montgomery (q = h 2^32 + 1), 1 = generic (Shoup / Barrett).
 *   "ntt_cross_xcd"  limb-polys of one-launch transforms whose workgroups were NOT all placed on one XCD and took the
 *                    agent-scope hand-off (slow, still correct); expected 0 under the dispatcher's observed round-robin placement.
 *   "ntt_fused_small", "ntt_fused_slots_per_xcd"  the threshold in force (0 = the one-launch form is off) and the guard's figure.
 * No reference counterpart: the reference's Arch has no tunables besides the .cfg keys (src/Arch.cpp:8-168). */
hm_status hm_set_option(hm_ctx *ctx, const char *name, uint64_t value);
/* What the back-end has kernels for at ring size N = 2^logN (round 6: one table, homulator_amd/csrc/hm_caps.h, instead of ring-size tests in
 * the callers).  Needs no context and no GPU; the same names are served by hm_get_counter for a context's own ring size:
 *   "cap_small_geometry"        1: the 8-coefficient passes, the one-launch transform and the small-launch transform x key kernel exist
 *   "cap_bconv_col_max_in"      widest digit (input limbs) of the fused conversion + first pass (hm_bconv_col, the conv lists of
 *                               hm_ntt_ip_desc / hm_ntt_fused_desc); 0: convert with hm_bconv_batch first
 *   "cap_bconv_col_pref_in"     widest digit for which that fused form is also the faster plan on MI355X (wider ones: hm_bconv_batch + the
 *                               transform's first pass; a planner's default, not a limit of the entry points)
 *   "cap_bconv_col_max_in_mix"  ... when the conversion carries the mix prologue (hm_ntt_fused_desc.conv with mix)
 *   "cap_ip_inverse_out"        1: hm_ntt_ip_desc.out_inverse is served
 *   "cap_col_slices"            ranks the column tiles of a limb-poly can be dealt to (hm_limbs_to_colslices / hm_bconv_col with a tile range)
 * Replaces: nothing upstream — the reference sizes its units from the .cfg (src/Arch.cpp:8-168) and has one shape of each. */
hm_status hm_capability(uint32_t logN, const char *name, uint64_t *value);
hm_status hm_get_counter(hm_ctx *ctx, const char *name, uint64_t *value); /* synchronises */

/* Timing on the context's stream (replaces Arch::getCycle include/Arch.h:271: elapsed device time in
 * nanoseconds instead of simulated cycles). */
hm_status hm_timer_start(hm_ctx *ctx);
hm_status hm_timer_stop(hm_ctx *ctx, uint64_t *elapsed_ns); /* synchronises */

#ifdef __cplusplus
}
#endif
#endif
