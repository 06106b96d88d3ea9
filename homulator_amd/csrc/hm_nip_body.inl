// hm_nip_body.inl — body of the fused transform x key kernel (k_ntt_row_ip) for ONE geometry of the passes; included by hm_backend.hip once
// per geometry, inside that geometry's namespace (hm16: 16 coefficients per thread, 256-thread workgroups — the throughput form; hm8: 8 per
// thread, 512-thread workgroups on the same 4096-coefficient tiles — the small-launch form, N = 2^16).  No include guard on purpose.
// INVOUT (round 5): 0 = no limb of the launch hands its outputs over as the first pass of their inverse transform (HM_NIP_INV_OUT), 1 = all of
// them do, 2 = per limb (the record says).  mont32 runs a mixed launch as ONE kernel (2: 155 VGPRs); in the generic build that form spills
// (the Shoup twiddles take twice the registers), so there the two kinds of limbs are two launches (0 and 1).
// TLR: log2 of the ROW tile a workgroup owns (HM_TL_ROW = 4096 coefficients = 16 rows; 11 = 8 rows works too — twice the workgroups of
// half the work — and measured level: not instantiated)
// XG (round 6): the digits that arrive in evaluation form (a limb's own digit) are read through the automorphism X -> X^a.x_galois — hrotate's
// rotated c1 is then never written (AUTO_Key(1) folds into the ModUp INTT and into this kernel); instantiations of their own
template <int OUTS, int INVOUT, int TLR = HM_TL_ROW, bool XG = false>
__device__ __forceinline__ void hm_nip_body(const HmNipArgs &a) {
  constexpr int TL = TLR, LOGR = HM_ROW_LOG, R2 = HmRounds<LOGR>::n - 1;
  __shared__ __attribute__((aligned(16))) uint64_t lds[HmLds<TL, LOGR, false>::WORDS];
  uint32_t entry, tile;
  if (!hm_block_map(1u << (a.logN - TL), a.n_limbs, a.logG, entry, tile)) return;
  const HmConstNipLimb rec = (HmConstNipLimb)(uintptr_t)a.limb + entry;
  const uint32_t mod = rec->mod;
  if (mod == HM_NTT_NONE) return;
  const size_t N = (size_t)1 << a.logN;
  const HmMod m = HM_CONST_MODS(a.mods)[mod];
  const HmW *twl = a.tw + (size_t)mod * N;
  const uint32_t s0 = a.logN - LOGR, prefix0 = tile << (TL - LOGR);
  const HmW *twt = a.twist + ((size_t)mod * (N >> LOGR) + prefix0) * 3;
  const uint32_t mask = rec->coeff_mask;
#if HM_NIP_WIDE
  typedef hm_u128 Acc;
#else
  typedef uint64_t Acc;
#endif
  Acc acc[OUTS][HM_EPT];
#pragma unroll
  for (int k = 0; k < OUTS; ++k)
#pragma unroll
    for (int i = 0; i < HM_EPT; ++i) acc[k][i] = 0;
#pragma unroll 1
  for (uint32_t j = 0; j < a.n_terms; ++j) {
    HmNttState st;
    // a thread id the compiler cannot see through: otherwise the ~60 lane offsets of the pass are hoisted out of the digit
    // loop as loop invariants and live (spilled) beside the accumulators
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    __builtin_assume(tid >= 0 && tid < (1 << TL) / HM_EPT);
    const uint32_t xl = rec->x[j];
    const uint64_t *y[OUTS];
#pragma unroll
    for (int k = 0; k < OUTS; ++k) y[k] = a.y + (size_t)rec->y[k][j] * N;
#if HM_NIP_PREFETCH
    uint64_t e[OUTS][HM_EPT];
    hm_ph_key_load<TL, LOGR, R2, OUTS>(e, tid, y, tile);
#endif
#if defined(HM_ABL_NIP_NOTRANSFORM)   // timing-only ablation: every digit taken as if already in evaluation form (loads + key MAC only)
    if (false) {
#else
    if (mask & (1u << j)) {   // wave-uniform
#endif
      const uint64_t *src = a.hand + (size_t)xl * N;
      const HmTw sc = {0, 0};
      const HmEpi ep = hm_epi_none();
      if (j) __syncthreads();   // the previous digit's last round has read the tile
      int nsync = 0;
      hm_ntt_pass_phases<TL, LOGR, false, false, 5, HM_NIP_LD_AUX>(st, tid, lds, src, nullptr, tile, twl, twt, s0, prefix0, m.q, sc, ep, [&] { hm_pass_sync<false>(nsync++); });
#if HM_NIP_WIDE
      hm_ph_below_2q(st, m.q);
#endif
    } else {
      if constexpr (XG) {
        HmEpi ep = hm_epi_none();
        ep.g = a.x_galois; ep.logN = a.logN;
        hm_ph_load_global_auto<TL, LOGR, false, R2, HM_NIP_LD_AUX>(st, tid, a.x + (size_t)xl * N, tile, ep);
      } else hm_ph_load_global<TL, LOGR, false, R2, HM_NIP_LD_AUX>(st, tid, a.x + (size_t)xl * N, tile);
    }
#if HM_NIP_PREFETCH
    hm_ph_mac_regs<OUTS, Acc>(st, acc, e, m, j);
#else
#if defined(HM_ABL_NIP_NOMAC)          // timing-only ablation: transforms only (no key loads, no products)
#pragma unroll
    for (int k = 0; k < OUTS; ++k)
#pragma unroll
      for (int i = 0; i < HM_EPT; ++i) acc[k][i] += st.v[i] + k;
#else
    hm_ph_mac<TL, LOGR, R2, OUTS, HM_NIP_MAC_CH, Acc>(st, acc, tid, y, tile, m, j);
#endif
#endif
  }
  uint64_t *out[OUTS];
#pragma unroll
  for (int k = 0; k < OUTS; ++k) out[k] = a.out + (size_t)rec->out[k] * N;
  if (INVOUT == 1 || (INVOUT == 2 && (mask & HM_NIP_INV_OUT))) {   // wave-uniform
    // Round 5 (InnerProOut -> ModDownINTTOut, src/Operation.cpp:294-445): the special limbs of the key-switch sum are only ever read by
    // the ModDown's inverse transform, whose first pass is a ROW pass over exactly this workgroup's 16 rows.  The reduced accumulators
    // sit in the registers of the forward ROW pass's last round, which is the inverse ROW pass's first: the inverse pass runs from
    // them (key 0, then key 1, on the tile this workgroup owns) and stores its hand-off where the ModDown INTT launch runs the
    // remaining COL pass (hm_ntt_second_pass); InnerProduceOut_Key{k}'s special limbs are never written or read back.
    const HmW *twl_i = a.tw_inv + (size_t)mod * N;
    const HmW *twt_i = a.twist_inv + ((size_t)mod * (N >> LOGR) + prefix0) * 3;
    // (straight-line code for the two keys: a loop with barriers inside is not unrolled and the accumulators end up in scratch)
    auto inverse_first_pass = [&](auto KK) {
      constexpr int k = decltype(KK)::value;
      HmNttState st;
#pragma unroll
      for (int i = 0; i < HM_EPT; ++i) st.v[i] = hm_mac_final(acc[k][i], m);
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));   // (lane offsets recomputed per pass, as in the digit loop)
      __builtin_assume(tid >= 0 && tid < (1 << TL) / HM_EPT);
      __syncthreads();   // the previous pass's last round has read the tile
      const HmTw sc = {0, 0};
      const HmEpi ep = hm_epi_none();
      int nsync = 0;
      hm_ntt_pass_phases<TL, LOGR, false, true, 0, 0, 0, HM_EPI_CHUNK, true>(st, tid, lds, nullptr, out[k], tile, twl_i, twt_i, s0, prefix0, m.q, sc, ep, [&] { hm_pass_sync<false>(nsync++); });
    };
    inverse_first_pass(HmNipKey<0>());
    if constexpr (OUTS == 2) inverse_first_pass(HmNipKey<1>());
    return;
  }
  hm_ph_mac_store<TL, LOGR, R2, OUTS, Acc, HM_NIP_ST_AUX>(acc, threadIdx.x, out, tile, m);
}

