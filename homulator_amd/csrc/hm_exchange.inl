// hm_exchange.inl — the multi-GPU side of the C ABI: RCCL / external transports, limb <-> slice exchanges in both domains,
// the replicate (all-gather of whole limb-polys).  Included by hm_backend.hip (one translation unit per arithmetic back-end); no include guard.
// ------------------------------------------------------------------------------------------------
// multi-GPU exchange
// ------------------------------------------------------------------------------------------------
extern "C" hm_status hm_comm_unique_id(void *out128) {
  if (!out128) return HM_ERR_ARG;
  if (const char *e = rccl_load()) return fail(nullptr, HM_ERR_COMM, "hm_comm_unique_id: %s", e);
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail(nullptr, HM_ERR_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
  memcpy(out128, &id, sizeof id);
  return HM_OK;
}
static hm_status verify_replicate_split(hm_ctx *c);
extern "C" hm_status hm_comm_init_rccl(hm_ctx *c, int rank, int world, const void *id128) {
  if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return HM_ERR_ARG;
  if (c->P.N % ((uint32_t)world * 512u)) return fail(c, HM_ERR_ARG, "hm_comm_init: world %d does not divide N / 512", world);
  if (const char *e = rccl_load()) return fail(c, HM_ERR_COMM, "hm_comm_init_rccl: %s", e);
  HM_HIP(c, hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) return fail(c, HM_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString(r));
  c->rank = rank; c->world = world; c->ext_fn = nullptr;
  return verify_replicate_split(c);
}
extern "C" hm_status hm_comm_init_external(hm_ctx *c, int rank, int world, hm_exchange_fn fn, void *user) {
  if (!c || !fn || world < 1 || rank < 0 || rank >= world) return HM_ERR_ARG;
  if (c->P.N % ((uint32_t)world * 512u)) return fail(c, HM_ERR_ARG, "hm_comm_init: world %d does not divide N / 512", world);
  c->rank = rank; c->world = world; c->ext_fn = fn; c->ext_user = user;
  return verify_replicate_split(c);
}
extern "C" hm_status hm_comm_info(const hm_ctx *c, int *rank, int *world) {
  if (!c) return HM_ERR_ARG;
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return HM_OK;
}
extern "C" hm_status hm_slice_rows(const uint32_t *owners, uint32_t n, uint32_t world, uint32_t *rows) {
  if (!owners || !rows || world == 0) return HM_ERR_ARG;
  std::vector<uint32_t> count(world + 1, 0);
  for (uint32_t i = 0; i < n; ++i) {
    if (owners[i] >= world) return HM_ERR_ARG;
    count[owners[i] + 1]++;
  }
  for (uint32_t r = 0; r < world; ++r) count[r + 1] += count[r];  // first row of each owner
  std::vector<uint32_t> next(count.begin(), count.end() - 1);
  for (uint32_t i = 0; i < n; ++i) rows[i] = next[owners[i]]++;
  return HM_OK;
}

static hm_status ensure_stage(hm_ctx *c, size_t words) {
  if (c->stage_words >= words) return HM_OK;
  HM_HIP(c, hipStreamSynchronize(c->stream));
  if (c->xstream) HM_HIP(c, hipStreamSynchronize(c->xstream));
  (void)hipFree(c->stage_send);
  (void)hipFree(c->stage_recv);
  c->stage_send = c->stage_recv = nullptr;
  HM_HIP(c, hipMalloc(&c->stage_send, words * 8));
  HM_HIP(c, hipMalloc(&c->stage_recv, words * 8));
  c->stage_words = words;
  return HM_OK;
}
static hm_status chunk_copy(hm_ctx *c, hipStream_t S, const uint64_t *src, uint64_t *dst, uint32_t len, const std::vector<uint32_t> &so,
                            const std::vector<uint32_t> &dof) {
  for (size_t base = 0; base < so.size(); base += HM_MAX_CHUNKS / 2) {
    const uint32_t cnt = (uint32_t)std::min<size_t>(HM_MAX_CHUNKS / 2, so.size() - base);
    HmChunkArgs a;
    a.src = src; a.dst = dst; a.len = len; a.n_chunks = cnt;
    for (uint32_t i = 0; i < cnt; ++i) { a.src_off[i] = so[base + i]; a.dst_off[i] = dof[base + i]; }
    hipLaunchKernelGGL(k_chunk_copy, dim3(cnt * (len / 512)), dim3(256), 0, S, a);
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}
// the exchange itself: per peer p, send_bytes[p] bytes at send + send_off[p] go to rank p and recv_bytes[p] bytes
// from rank p land at recv + recv_off[p].  Nothing is sent to self (callers place their own part directly).
static hm_status all_to_all(hm_ctx *c, hipStream_t S, const uint64_t *send, std::vector<size_t> send_off, std::vector<size_t> send_bytes,
                            uint64_t *recv, std::vector<size_t> recv_off, std::vector<size_t> recv_bytes) {
  if (c->world == 1) return HM_OK;
  send_bytes[c->rank] = recv_bytes[c->rank] = 0;
  if (c->ext_fn) {
    HM_HIP(c, hipStreamSynchronize(S));   // the external transports move the bytes on the host's clock
    if (c->ext_fn(c->ext_user, send, send_off.data(), send_bytes.data(), recv, recv_off.data(), recv_bytes.data()))
      return fail(c, HM_ERR_COMM, "external exchange failed");
    return HM_OK;
  }
  if (!c->comm) return fail(c, HM_ERR_COMM, "no communicator: call hm_comm_init_rccl first");
  ncclResult_t r = g_rccl.GroupStart();
  for (int p = 0; p < c->world && r == ncclSuccess; ++p) {
    if (send_bytes[p]) r = g_rccl.Send((const char *)send + send_off[p], send_bytes[p] / 8, ncclUint64, p, c->comm, S);
    if (recv_bytes[p] && r == ncclSuccess) r = g_rccl.Recv((char *)recv + recv_off[p], recv_bytes[p] / 8, ncclUint64, p, c->comm, S);
  }
  ncclResult_t e = g_rccl.GroupEnd();
  if (r == ncclSuccess) r = e;
  if (r != ncclSuccess) return fail(c, HM_ERR_COMM, "RCCL exchange: %s", g_rccl.GetErrorString(r));
  return HM_OK;
}

// hm_replicate_limbs chooses between one exchange and scatter + exchange of chunks from replicate_split_bytes, which every process reads
// from ITS environment (HOMULATOR_REPLICATE_SPLIT) or option: ranks that disagree would enter collectives of different shape and hang.
// So the first thing a new communicator carries is every rank's threshold to every other rank (8 bytes per pair, over the transport
// itself); a mismatch fails hm_comm_init_* with HM_ERR_COMM on every rank, and hm_set_option refuses to change the value afterwards.
static hm_status verify_replicate_split(hm_ctx *c) {
  const uint32_t W = (uint32_t)c->world, me = (uint32_t)c->rank;
  if (W < 2) return HM_OK;
  hm_status st;
  if ((st = ensure_stage(c, 512u * W))) return st;
  std::vector<uint64_t> mine(512u * W, c->replicate_split_bytes), got(512u * W, 0);
  HM_HIP(c, hipMemcpyAsync(c->stage_send, mine.data(), 8u * mine.size(), hipMemcpyHostToDevice, c->stream));
  std::vector<size_t> so(W, 0), sb(W, 8), ro(W, 0), rb(W, 8);
  for (uint32_t p = 0; p < W; ++p) { so[p] = ro[p] = (size_t)p * 4096; }
  if ((st = all_to_all(c, c->stream, c->stage_send, so, sb, c->stage_recv, ro, rb))) return st;
  HM_HIP(c, hipMemcpyAsync(got.data(), c->stage_recv, 8u * got.size(), hipMemcpyDeviceToHost, c->stream));
  HM_HIP(c, hipStreamSynchronize(c->stream));
  for (uint32_t p = 0; p < W; ++p)
    if (p != me && got[(size_t)p * 512] != c->replicate_split_bytes) {
      st = fail(c, HM_ERR_COMM, "hm_comm_init: rank %u holds replicate_split_bytes = %llu, rank %u holds %llu (HOMULATOR_REPLICATE_SPLIT / hm_set_option must agree on every rank)",
                me, (unsigned long long)c->replicate_split_bytes, p, (unsigned long long)got[(size_t)p * 512]);
      if (c->comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
      c->ext_fn = nullptr; c->world = 1; c->rank = 0;
      return st;
    }
  return HM_OK;
}

extern "C" hm_status hm_limbs_to_slices(hm_ctx *c, const uint64_t *buf, const uint32_t *limbs, const uint32_t *owners,
                                        uint32_t n, uint64_t *slices) {
  if (!c) return HM_ERR_ARG;
  if (!buf || !limbs || !owners || !slices) return fail(c, HM_ERR_ARG, "hm_limbs_to_slices: null argument");
  const uint32_t W = (uint32_t)c->world, me = (uint32_t)c->rank, len = c->P.N / W;
  std::vector<uint32_t> rows(n);
  if (hm_slice_rows(owners, n, W, rows.data())) return fail(c, HM_ERR_ARG, "hm_limbs_to_slices: owner out of range");
  HM_HIP(c, hipSetDevice(c->device));
  std::vector<uint32_t> cnt(W, 0);
  for (uint32_t i = 0; i < n; ++i) cnt[owners[i]]++;
  const uint32_t mine = cnt[me];
  // send block for rank p: my limbs' slice p, in list order -> [mine][len]; laid out rank after rank (self skipped)
  hm_status st = ensure_stage(c, (size_t)std::max<uint32_t>(mine, 1) * W * len);
  if (st) return st;
  std::vector<size_t> send_off(W, 0), send_bytes(W, 0), recv_off(W, 0), recv_bytes(W, 0);
  size_t so_acc = 0;
  for (uint32_t p = 0; p < W; ++p) {
    send_off[p] = so_acc;
    send_bytes[p] = p == me ? 0 : (size_t)mine * len * 8;
    so_acc += send_bytes[p];
  }
  // receive straight into the slice rows of each source (rows of one owner are contiguous)
  uint32_t first = 0;
  for (uint32_t p = 0; p < W; ++p) { recv_off[p] = (size_t)first * len * 8; recv_bytes[p] = (size_t)cnt[p] * len * 8; first += cnt[p]; }
  std::vector<uint32_t> so, dof, so_self, do_self;
  // chunk units are `len` words: limb l slice p starts at (l * W + p) chunks of buf
  for (uint32_t p = 0; p < W; ++p) {
    uint32_t j = 0;
    for (uint32_t i = 0; i < n; ++i) {
      if (owners[i] != me) continue;
      if (p == me) { so_self.push_back(limbs[i] * W + p); do_self.push_back(rows[i]); }
      else { so.push_back(limbs[i] * W + p); dof.push_back((uint32_t)(send_off[p] / 8 / len) + j); }
      ++j;
    }
  }
  hipStream_t S;
  if ((st = exchange_stream_begin(c, &S))) return st;
  if ((st = chunk_copy(c, S, buf, slices, len, so_self, do_self))) return st;
  if ((st = chunk_copy(c, S, buf, c->stage_send, len, so, dof))) return st;
  // a rank's own rows must not be overwritten by the receive: recv_off[me] block is skipped by all_to_all
  return all_to_all(c, S, c->stage_send, send_off, send_bytes, slices, recv_off, recv_bytes);
}

extern "C" hm_status hm_slices_to_limbs(hm_ctx *c, const uint64_t *slices, uint64_t *buf, const uint32_t *limbs,
                                        const uint32_t *owners, uint32_t n) {
  if (!c) return HM_ERR_ARG;
  if (!buf || !limbs || !owners || !slices) return fail(c, HM_ERR_ARG, "hm_slices_to_limbs: null argument");
  const uint32_t W = (uint32_t)c->world, me = (uint32_t)c->rank, len = c->P.N / W;
  std::vector<uint32_t> rows(n);
  if (hm_slice_rows(owners, n, W, rows.data())) return fail(c, HM_ERR_ARG, "hm_slices_to_limbs: owner out of range");
  HM_HIP(c, hipSetDevice(c->device));
  std::vector<uint32_t> cnt(W, 0);
  for (uint32_t i = 0; i < n; ++i) cnt[owners[i]]++;
  const uint32_t mine = cnt[me];
  hm_status st = ensure_stage(c, (size_t)std::max<uint32_t>(mine, 1) * W * len);
  if (st) return st;
  // send: the rows of owner p (contiguous in `slices`) go to rank p as they are
  std::vector<size_t> send_off(W, 0), send_bytes(W, 0), recv_off(W, 0), recv_bytes(W, 0);
  uint32_t first = 0;
  for (uint32_t p = 0; p < W; ++p) { send_off[p] = (size_t)first * len * 8; send_bytes[p] = (size_t)cnt[p] * len * 8; first += cnt[p]; }
  // receive: from rank s the slice s of each of my limbs, [mine][len], into the staging buffer
  size_t ro_acc = 0;
  for (uint32_t p = 0; p < W; ++p) {
    recv_off[p] = ro_acc;
    recv_bytes[p] = p == me ? 0 : (size_t)mine * len * 8;
    ro_acc += recv_bytes[p];
  }
  hipStream_t S;
  if ((st = exchange_stream_begin(c, &S))) return st;
  if ((st = all_to_all(c, S, slices, send_off, send_bytes, c->stage_recv, recv_off, recv_bytes))) return st;
  std::vector<uint32_t> so, dof, so_self, do_self;
  for (uint32_t p = 0; p < W; ++p) {
    uint32_t j = 0;
    for (uint32_t i = 0; i < n; ++i) {
      if (owners[i] != me) continue;
      if (p == me) { so_self.push_back(rows[i]); do_self.push_back(limbs[i] * W + p); }
      else { so.push_back((uint32_t)(recv_off[p] / 8 / len) + j); dof.push_back(limbs[i] * W + p); }
      ++j;
    }
  }
  if ((st = chunk_copy(c, S, slices, buf, len, so_self, do_self))) return st;
  return chunk_copy(c, S, c->stage_recv, buf, len, so, dof);
}

// ---- the same exchanges in the TRANSPOSED domain (round 4).  Index i = x1 * 256 + x2: rank p's slice of a limb-poly is the column block
// x2 in [p * cw, (p + 1) * cw), cw = 256 / world, of every row x1.  A length-(N/256) transform over x1 (the COL pass) is local to a column, so
// the slice holder runs base conversion AND first transform pass (hm_bconv_col) on its columns and the exchange back carries the first
// pass's hand-off; the limb owner only runs the second pass.  `slices` keeps the limb-poly layout (n rows of N words, only this rank's
// columns are valid): the kernels address it like any limb-poly.  Same bytes on the wire as the contiguous slices.
static hm_status col_copy(hm_ctx *c, hipStream_t S, const uint64_t *src, uint64_t *dst, uint32_t rows, uint32_t cw, uint32_t ss, uint32_t ds,
                          const std::vector<uint32_t> &so, const std::vector<uint32_t> &dof) {
  for (size_t base = 0; base < so.size(); base += HM_MAX_CHUNKS / 2) {
    const uint32_t cnt = (uint32_t)std::min<size_t>(HM_MAX_CHUNKS / 2, so.size() - base);
    HmColArgs a;
    a.src = src; a.dst = dst; a.rows = rows; a.cw = cw; a.src_stride = ss; a.dst_stride = ds; a.n_chunks = cnt;
    for (uint32_t i = 0; i < cnt; ++i) { a.src_off[i] = so[base + i]; a.dst_off[i] = dof[base + i]; }
    const uint32_t per = (rows * cw / 2 + 255) / 256;
    hipLaunchKernelGGL(k_col_copy, dim3(cnt * per), dim3(256), 0, S, a);
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}
static hm_status col_geometry(hm_ctx *c, const char *what, uint32_t &rowsN, uint32_t &cw) {
  const uint32_t W = (uint32_t)c->world;
  if (W > (c->P.N >> HM_TL_COL) || (256u % W)) return fail(c, HM_ERR_UNSUPPORTED, "%s: column slices need world <= N / 4096 (a rank holds one first-pass tile at least)", what);
  rowsN = c->P.N >> HM_ROW_LOG;
  cw = 256u / W;
  return HM_OK;
}
extern "C" hm_status hm_limbs_to_colslices(hm_ctx *c, const uint64_t *buf, const uint32_t *limbs, const uint32_t *owners, uint32_t n, uint64_t *slices) {
  if (!c) return HM_ERR_ARG;
  if (!buf || !limbs || !owners || !slices) return fail(c, HM_ERR_ARG, "hm_limbs_to_colslices: null argument");
  const uint32_t W = (uint32_t)c->world, me = (uint32_t)c->rank, len = c->P.N / W, N = c->P.N;
  uint32_t R = 0, cw = 0;
  hm_status st = col_geometry(c, "hm_limbs_to_colslices", R, cw);
  if (st) return st;
  std::vector<uint32_t> rows(n);
  if (hm_slice_rows(owners, n, W, rows.data())) return fail(c, HM_ERR_ARG, "hm_limbs_to_colslices: owner out of range");
  auto U = [N](uint32_t limb, uint32_t col) { return (uint32_t)(((size_t)limb * N + col) / 2); };   // 16-byte units
  HM_HIP(c, hipSetDevice(c->device));
  std::vector<uint32_t> cnt(W, 0);
  for (uint32_t i = 0; i < n; ++i) cnt[owners[i]]++;
  const uint32_t mine = cnt[me];
  // staging: send = [peer][my limbs][R][cw] (self skipped), recv = [source][its limbs][R][cw]
  if ((st = ensure_stage(c, (size_t)std::max<uint32_t>(std::max(mine, n), 1) * W * len))) return st;
  std::vector<size_t> send_off(W, 0), send_bytes(W, 0), recv_off(W, 0), recv_bytes(W, 0);
  size_t acc = 0;
  for (uint32_t p = 0; p < W; ++p) { send_off[p] = acc; send_bytes[p] = p == me ? 0 : (size_t)mine * len * 8; acc += send_bytes[p]; }
  uint32_t first = 0;
  for (uint32_t p = 0; p < W; ++p) { recv_off[p] = (size_t)first * len * 8; recv_bytes[p] = (size_t)cnt[p] * len * 8; first += cnt[p]; }
  std::vector<uint32_t> so, dof, so_self, do_self;
  for (uint32_t p = 0; p < W; ++p) {
    uint32_t j = 0;
    for (uint32_t i = 0; i < n; ++i) {
      if (owners[i] != me) continue;
      if (p == me) { so_self.push_back(U(limbs[i], p * cw)); do_self.push_back(U(rows[i], me * cw)); }
      else { so.push_back(U(limbs[i], p * cw)); dof.push_back((uint32_t)((send_off[p] / 8 + (size_t)j * len) / 2)); }
      ++j;
    }
  }
  hipStream_t S;
  if ((st = exchange_stream_begin(c, &S))) return st;
  if ((st = col_copy(c, S, buf, slices, R, cw, 256, 256, so_self, do_self))) return st;                 // my own limbs' block: straight into place
  if ((st = col_copy(c, S, buf, c->stage_send, R, cw, 256, cw, so, dof))) return st;                   // pack
  if ((st = all_to_all(c, S, c->stage_send, send_off, send_bytes, c->stage_recv, recv_off, recv_bytes))) return st;
  so.clear(); dof.clear();
  for (uint32_t p = 0; p < W; ++p) {                                                                    // unpack: source p's limbs, my columns
    if (p == me) continue;
    uint32_t j = 0;
    for (uint32_t i = 0; i < n; ++i) {
      if (owners[i] != p) continue;
      so.push_back((uint32_t)((recv_off[p] / 8 + (size_t)j * len) / 2)); dof.push_back(U(rows[i], me * cw));
      ++j;
    }
  }
  return col_copy(c, S, c->stage_recv, slices, R, cw, cw, 256, so, dof);
}
extern "C" hm_status hm_colslices_to_limbs(hm_ctx *c, const uint64_t *slices, uint64_t *buf, const uint32_t *limbs, const uint32_t *owners, uint32_t n) {
  if (!c) return HM_ERR_ARG;
  if (!buf || !limbs || !owners || !slices) return fail(c, HM_ERR_ARG, "hm_colslices_to_limbs: null argument");
  const uint32_t W = (uint32_t)c->world, me = (uint32_t)c->rank, len = c->P.N / W, N = c->P.N;
  uint32_t R = 0, cw = 0;
  hm_status st = col_geometry(c, "hm_colslices_to_limbs", R, cw);
  if (st) return st;
  std::vector<uint32_t> rows(n);
  if (hm_slice_rows(owners, n, W, rows.data())) return fail(c, HM_ERR_ARG, "hm_colslices_to_limbs: owner out of range");
  auto U = [N](uint32_t limb, uint32_t col) { return (uint32_t)(((size_t)limb * N + col) / 2); };   // 16-byte units
  HM_HIP(c, hipSetDevice(c->device));
  std::vector<uint32_t> cnt(W, 0);
  for (uint32_t i = 0; i < n; ++i) cnt[owners[i]]++;
  const uint32_t mine = cnt[me];
  if ((st = ensure_stage(c, (size_t)std::max<uint32_t>(std::max(mine, n), 1) * W * len))) return st;
  // send: to owner p my column block of each of ITS limbs (compact, in list order); receive: from every rank its block of each of MY limbs
  std::vector<size_t> send_off(W, 0), send_bytes(W, 0), recv_off(W, 0), recv_bytes(W, 0);
  uint32_t first = 0;
  for (uint32_t p = 0; p < W; ++p) { send_off[p] = (size_t)first * len * 8; send_bytes[p] = (size_t)cnt[p] * len * 8; first += cnt[p]; }
  size_t acc = 0;
  for (uint32_t p = 0; p < W; ++p) { recv_off[p] = acc; recv_bytes[p] = p == me ? 0 : (size_t)mine * len * 8; acc += recv_bytes[p]; }
  std::vector<uint32_t> so, dof, so_self, do_self;
  for (uint32_t p = 0; p < W; ++p) {
    uint32_t j = 0;
    for (uint32_t i = 0; i < n; ++i) {
      if (owners[i] != p) continue;
      if (p == me) { so_self.push_back(U(rows[i], me * cw)); do_self.push_back(U(limbs[i], me * cw)); }
      else { so.push_back(U(rows[i], me * cw)); dof.push_back((uint32_t)((send_off[p] / 8 + (size_t)j * len) / 2)); }
      ++j;
    }
  }
  hipStream_t S;
  if ((st = exchange_stream_begin(c, &S))) return st;
  if ((st = col_copy(c, S, slices, buf, R, cw, 256, 256, so_self, do_self))) return st;
  if ((st = col_copy(c, S, slices, c->stage_send, R, cw, 256, cw, so, dof))) return st;
  if ((st = all_to_all(c, S, c->stage_send, send_off, send_bytes, c->stage_recv, recv_off, recv_bytes))) return st;
  so.clear(); dof.clear();
  for (uint32_t p = 0; p < W; ++p) {
    if (p == me) continue;
    uint32_t j = 0;
    for (uint32_t i = 0; i < n; ++i) {
      if (owners[i] != me) continue;
      so.push_back((uint32_t)((recv_off[p] / 8 + (size_t)j * len) / 2)); dof.push_back(U(limbs[i], p * cw));
      ++j;
    }
  }
  return col_copy(c, S, c->stage_recv, buf, R, cw, cw, 256, so, dof);
}

extern "C" hm_status hm_replicate_limbs(hm_ctx *c, uint64_t *buf, const uint32_t *limbs, const uint32_t *owners, uint32_t n) {
  if (!c) return HM_ERR_ARG;
  if (!buf || !limbs || !owners) return fail(c, HM_ERR_ARG, "hm_replicate_limbs: null argument");
  const uint32_t W = (uint32_t)c->world, me = (uint32_t)c->rank, N = c->P.N;
  if (W == 1) return HM_OK;
  HM_HIP(c, hipSetDevice(c->device));
  std::vector<uint32_t> cnt(W, 0);
  for (uint32_t i = 0; i < n; ++i) {
    if (owners[i] >= W) return fail(c, HM_ERR_ARG, "hm_replicate_limbs: owner out of range");
    cnt[owners[i]]++;
  }
  hm_status st = ensure_stage(c, (size_t)std::max<uint32_t>(n, 1) * N);
  if (st) return st;
  // send: my limbs packed [mine][N] (the same block to every peer); receive: block of source s = its limbs, list order
  std::vector<size_t> send_off(W, 0), send_bytes(W, 0), recv_off(W, 0), recv_bytes(W, 0);
  size_t acc = 0;
  for (uint32_t p = 0; p < W; ++p) {
    send_bytes[p] = p == me ? 0 : (size_t)cnt[me] * N * 8;
    recv_off[p] = acc;
    recv_bytes[p] = p == me ? 0 : (size_t)cnt[p] * N * 8;
    acc += recv_bytes[p];
  }
  std::vector<uint32_t> so, dof;
  uint32_t j = 0;
  for (uint32_t i = 0; i < n; ++i)
    if (owners[i] == me) { so.push_back(limbs[i]); dof.push_back(j++); }
  hipStream_t S;
  if ((st = exchange_stream_begin(c, &S))) return st;
  if ((st = chunk_copy(c, S, buf, c->stage_send, N, so, dof))) return st;
  // one owner, many ranks, a list worth splitting: scatter + exchange of chunks.  The packed list [n][N] is cut into W - 1 runs of whole
  // 512-word blocks; peer k (the ranks other than the owner, in rank order) gets run k from the owner, then sends it to the other peers.
  // Every rank takes the same decision from the same lists.
  const uint32_t owner = n ? owners[0] : 0;
  const size_t total = (size_t)n * N * 8;
  if (W >= 4 && n && cnt[owner] == n && c->replicate_split_bytes && total >= c->replicate_split_bytes) {
    const size_t run = ((total / 4096 + (W - 2)) / (W - 1)) * 4096;
    auto run_of = [&](uint32_t p) { return (size_t)(p < owner ? p : p - 1) * run; };               // byte offset of peer p's run
    auto len_of = [&](uint32_t p) { const size_t o = run_of(p); return o >= total ? (size_t)0 : std::min(run, total - o); };
    std::vector<size_t> so1(W, 0), sb1(W, 0), ro1(W, 0), rb1(W, 0), so2(W, 0), sb2(W, 0), ro2(W, 0), rb2(W, 0);
    for (uint32_t p = 0; p < W; ++p) {
      if (p == me || p == owner) continue;
      if (me == owner) { so1[p] = run_of(p); sb1[p] = len_of(p); }             // phase 1: the owner's runs go out
      else { so2[p] = run_of(me); sb2[p] = len_of(me); ro2[p] = run_of(p); rb2[p] = len_of(p); }   // phase 2: my run to the other peers, theirs to me
    }
    if (me != owner) { ro1[owner] = run_of(me); rb1[owner] = len_of(me); }
    if ((st = all_to_all(c, S, c->stage_send, so1, sb1, c->stage_recv, ro1, rb1))) return st;
    // phase 2 sends FROM and receives INTO stage_recv inside one group: a rank sends the run it received in phase 1 (bytes [run_of(me),
    // + len)) and receives the other peers' runs at THEIR offsets — disjoint ranges of the one buffer, ordered behind phase 1 on the stream
    if ((st = all_to_all(c, S, c->stage_recv, so2, sb2, c->stage_recv, ro2, rb2))) return st;
    if (me == owner) return HM_OK;
    so.clear(); dof.clear();
    for (uint32_t i = 0; i < n; ++i) { so.push_back(i); dof.push_back(limbs[i]); }
    return chunk_copy(c, S, c->stage_recv, buf, N, so, dof);
  }
  if ((st = all_to_all(c, S, c->stage_send, send_off, send_bytes, c->stage_recv, recv_off, recv_bytes))) return st;
  so.clear(); dof.clear();
  for (uint32_t p = 0; p < W; ++p) {
    if (p == me) continue;
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; ++i)
      if (owners[i] == p) { so.push_back((uint32_t)(recv_off[p] / 8 / N) + k++); dof.push_back(limbs[i]); }
  }
  return chunk_copy(c, S, c->stage_recv, buf, N, so, dof);
}
