"""ctypes binding of include/homulator_hip.h (libhomulator_hip.so).  No CPU fallback: a missing library or a
machine without a HIP device is an error, never a silent detour."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HOMULATOR_HIP_LIB") or os.path.join(_HERE, "lib", "libhomulator_hip.so")  # override: A/B builds
# the two arithmetic back-ends libhomulator_hip.so maps from its own directory (hm_dispatch.cpp): word-wise Montgomery for chains of
# primes h 2^32 + 1 (the default), Shoup / Barrett for any other NTT-friendly chain below 2^60
BACKEND_LIBS = ("libhm_m32.so", "libhm_gen.so")

OP_MUL, OP_MAC2, OP_MAC_ADD, OP_ADD, OP_SUB, OP_MUL_CONST, OP_SUB_SCALE, OP_COPY, OP_SUB_SCALE_ADD = range(9)

# every symbol include/homulator_hip.h declares
SYMBOLS = [
    "hm_create", "hm_destroy", "hm_last_error", "hm_version", "hm_get_modulus", "hm_get_psi", "hm_malloc", "hm_free",
    "hm_memcpy_h2d", "hm_memcpy_d2h", "hm_memcpy_d2d", "hm_sync", "hm_stream", "hm_wait_for", "hm_ntt", "hm_ntt_sub_scale", "hm_ntt_mix_sub_scale", "hm_tensor", "hm_inner_product", "hm_automorph", "hm_ewe",
    "hm_bconv", "hm_bconv_batch", "hm_bconv_consts", "hm_fill_uniform", "hm_timer_start", "hm_timer_stop", "hm_comm_unique_id", "hm_comm_init_rccl", "hm_comm_init_external",
    "hm_capture_begin", "hm_capture_end", "hm_graph_launch", "hm_graph_destroy", "hm_comm_info", "hm_slice_rows", "hm_limbs_to_slices", "hm_slices_to_limbs", "hm_replicate_limbs",
    "hm_set_option", "hm_get_counter", "hm_ntt_inner_product", "hm_exchange_stream", "hm_exchange_mark", "hm_exchange_wait",
    "hm_bconv_col", "hm_limbs_to_colslices", "hm_colslices_to_limbs", "hm_ntt_second_pass", "hm_ntt_ex", "hm_capability", "hm_inner_product_ex",
]


class hm_ntt_fused_desc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_limbs", C.c_void_p), ("mix", C.c_void_p), ("mix_limbs", C.c_void_p), ("mix_k", C.c_void_p),
                ("minuend", C.c_void_p), ("minuend_limbs", C.c_void_p), ("addend", C.c_void_p), ("addend_limbs", C.c_void_p),
                ("addend_k", C.c_void_p), ("out", C.c_void_p), ("out_limbs", C.c_void_p), ("mod_ids", C.c_void_p), ("n", C.c_uint32),
                ("k", C.c_void_p), ("conv", C.c_void_p), ("n_conv", C.c_uint32), ("addend_galois", C.c_void_p)]


class hm_ip_desc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_limbs", C.c_void_p), ("y", C.c_void_p), ("y_limbs", C.c_void_p), ("out", C.c_void_p), ("out_limbs", C.c_void_p),
                ("mod_ids", C.c_void_p), ("n", C.c_uint32), ("n_terms", C.c_uint32), ("n_out", C.c_uint32), ("x_galois", C.c_uint32)]


class hm_ntt_ip_desc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_limbs", C.c_void_p), ("x_is_coeff", C.c_void_p), ("hand", C.c_void_p), ("hand_limbs", C.c_void_p),
                ("y", C.c_void_p), ("y_limbs", C.c_void_p), ("out", C.c_void_p), ("out_limbs", C.c_void_p), ("mod_ids", C.c_void_p),
                ("n", C.c_uint32), ("n_terms", C.c_uint32), ("n_out", C.c_uint32), ("conv", C.c_void_p), ("n_conv", C.c_uint32),
                ("out_inverse", C.c_void_p), ("x_galois", C.c_uint32)]


class hm_bconv_desc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_limbs", C.c_void_p), ("in_ids", C.c_void_p), ("n_in", C.c_uint32),
                ("out", C.c_void_p), ("out_limbs", C.c_void_p), ("out_ids", C.c_void_p), ("n_out", C.c_uint32),
                ("log_len", C.c_uint32), ("sub_from", C.c_void_p), ("sub_from_limbs", C.c_void_p), ("add", C.c_void_p), ("add_limbs", C.c_void_p),
                ("sub_k", C.c_void_p), ("in_packed", C.c_uint32)]


class hm_ntt_desc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_limbs", C.c_void_p), ("out", C.c_void_p), ("out_limbs", C.c_void_p), ("mod_ids", C.c_void_p), ("n", C.c_uint32),
                ("inverse", C.c_int), ("scale", C.c_void_p), ("second_pass_only", C.c_int), ("out_packed", C.c_void_p),
                ("in_galois", C.c_void_p)]


class hm_params(C.Structure):
    _fields_ = [("logN", C.c_uint32), ("L", C.c_uint32), ("K", C.c_uint32), ("device", C.c_int32),
                ("q", C.c_void_p), ("p", C.c_void_p), ("psi", C.c_void_p)]


_lib = None


def load():
    """Load the HIP backend.  Raises if it has not been built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                          "(make -C homulator_amd/csrc); homulator_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.hm_create.argtypes = [C.POINTER(vp), C.POINTER(hm_params)]
    L.hm_destroy.argtypes = [vp]
    L.hm_last_error.restype = C.c_char_p
    L.hm_last_error.argtypes = [vp]
    L.hm_version.restype = C.c_char_p
    L.hm_get_modulus.argtypes = [vp, u32, C.POINTER(u64)]
    L.hm_get_psi.argtypes = [vp, u32, C.POINTER(u64)]
    L.hm_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.hm_free.argtypes = [vp, vp]
    L.hm_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.hm_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.hm_memcpy_d2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.hm_sync.argtypes = [vp]
    L.hm_stream.restype = vp
    L.hm_stream.argtypes = [vp]
    L.hm_wait_for.argtypes = [vp, vp]
    L.hm_ntt.argtypes = [vp, vp, vp, vp, vp, vp, u32, i32, vp]
    L.hm_ntt_second_pass.argtypes = [vp, vp, vp, vp, u32, i32, vp]
    L.hm_ntt_ex.argtypes = [vp, C.POINTER(hm_ntt_desc)]
    L.hm_inner_product_ex.argtypes = [vp, C.POINTER(hm_ip_desc)]
    L.hm_ntt_sub_scale.argtypes = [vp] + [vp] * 9 + [u32, vp]
    L.hm_ntt_mix_sub_scale.argtypes = [vp, C.POINTER(hm_ntt_fused_desc)]
    L.hm_tensor.argtypes = [vp] + [vp] * 15 + [u32]
    L.hm_inner_product.argtypes = [vp] + [vp] * 7 + [u32, u32, u32]
    L.hm_automorph.argtypes = [vp, vp, vp, vp, vp, u32, u32]
    L.hm_ewe.argtypes = [vp, i32] + [vp] * 11 + [u32, vp]
    L.hm_bconv.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, u32]
    L.hm_bconv_batch.argtypes = [vp, C.POINTER(hm_bconv_desc), u32]
    L.hm_bconv_consts.argtypes = [vp, vp, u32, vp, u32, vp, vp]
    L.hm_fill_uniform.argtypes = [vp, vp, vp, vp, u32, u64]
    L.hm_ntt_inner_product.argtypes = [vp, C.POINTER(hm_ntt_ip_desc)]
    L.hm_set_option.argtypes = [vp, C.c_char_p, u64]
    L.hm_get_counter.argtypes = [vp, C.c_char_p, C.POINTER(u64)]
    L.hm_capability.argtypes = [u32, C.c_char_p, C.POINTER(u64)]
    L.hm_comm_init_external.argtypes = [vp, i32, i32, vp, vp]
    L.hm_timer_start.argtypes = [vp]
    L.hm_timer_stop.argtypes = [vp, C.POINTER(u64)]
    _lib = L
    return L


class HmError(RuntimeError):
    pass


def capability(logN, name):
    """the back-end's capability table (homulator_amd/csrc/hm_caps.h) for ring size 2^logN: needs no context and no GPU"""
    v = C.c_uint64()
    if load().hm_capability(int(logN), name.encode(), C.byref(v)) != 0:
        raise HmError(f"hm_capability: unknown capability {name}")
    return v.value


def _u32(a):
    if a is None:
        return None, None
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.uint32))
    return arr, arr.ctypes.data_as(C.c_void_p)


def _u64(a):
    if a is None:
        return None, None
    arr = np.ascontiguousarray(np.asarray([int(x) for x in a], dtype=np.uint64))
    return arr, arr.ctypes.data_as(C.c_void_p)


class DeviceBuf:
    """[n_limbs][N] uint64 words in HBM, limb-major (a plain hipMalloc'd pointer)."""

    def __init__(self, ctx, n_limbs):
        self.ctx, self.n_limbs = ctx, n_limbs
        p = C.c_void_p()
        ctx._ck(ctx.L.hm_malloc(ctx.h, 8 * ctx.N * n_limbs, C.byref(p)))
        self.ptr = p.value

    def limb_ptr(self, limb):
        return self.ptr + 8 * self.ctx.N * limb

    def upload(self, arr, limb0=0):
        a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, self.ctx.N)
        assert limb0 + a.shape[0] <= self.n_limbs
        self.ctx._ck(self.ctx.L.hm_memcpy_h2d(self.ctx.h, self.limb_ptr(limb0), a.ctypes.data_as(C.c_void_p), a.nbytes))
        return self

    def download(self, limb0=0, n=None):
        n = self.n_limbs - limb0 if n is None else n
        out = np.empty((n, self.ctx.N), dtype=np.uint64)
        self.ctx._ck(self.ctx.L.hm_memcpy_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.limb_ptr(limb0), out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.L.hm_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """One GPU, one parameter set.  Thin, 1:1 with the C ABI."""

    def __init__(self, logN, L, K, device=0, q=None, p=None):
        """q / p: the L chain moduli and the K special moduli (distinct primes = 1 mod 2N below 2^60; hm_create runs them on the
        word-wise Montgomery back-end if they all are h 2^32 + 1, on the generic one otherwise); default: the library's own chain"""
        self.L = load()
        self.h = C.c_void_p()
        qa = None if q is None else np.ascontiguousarray(np.asarray(q, dtype=np.uint64))
        pa = None if p is None else np.ascontiguousarray(np.asarray(p, dtype=np.uint64))
        if (qa is not None and len(qa) != L) or (pa is not None and len(pa) != K):
            raise ValueError("q needs L entries and p needs K")
        prm = hm_params(logN, L, K, device, None if qa is None else qa.ctypes.data_as(C.c_void_p), None if pa is None else pa.ctypes.data_as(C.c_void_p), None)
        st = self.L.hm_create(C.byref(self.h), C.byref(prm))
        if st != 0:
            raise HmError(f"hm_create failed ({st}): {self.L.hm_last_error(None).decode()}")
        self.logN, self.N, self.nQ, self.K = logN, 1 << logN, L, K
        self.chain = None if qa is None else [int(x) for x in qa] + ([] if pa is None else [int(x) for x in pa])
        q = C.c_uint64()
        self.moduli = []
        for m in range(L + K):
            self._ck(self.L.hm_get_modulus(self.h, m, C.byref(q)))
            self.moduli.append(q.value)

    def _ck(self, st):
        if st != 0:
            raise HmError(f"hm error {st}: {self.L.hm_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.L.hm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc(self, n_limbs):
        return DeviceBuf(self, n_limbs)

    def from_host(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, self.N)
        return self.alloc(a.shape[0]).upload(a)

    def sync(self):
        self._ck(self.L.hm_sync(self.h))

    def ext_ids(self, ell):
        return list(range(ell)) + [self.nQ + i for i in range(self.K)]

    # ---- compute calls (device pointers + limb lists)
    def ntt(self, src, dst, mod_ids, inverse=False, in_limbs=None, out_limbs=None, scale=None, out_packed=None, in_galois=None):
        """out_packed (inverse only): per limb, store the split-30 packed form the base conversions take with in_packed (hm_ntt_ex);
        in_galois (inverse only): per limb, read the input through the automorphism X -> X^g (hm_ntt_ex)"""
        n = len(mod_ids)
        k1, pi = _u32(in_limbs)
        k2, po = _u32(out_limbs)
        k3, pm = _u32(mod_ids)
        k4, ps = _u64(scale)
        if out_packed is None and in_galois is None:
            self._ck(self.L.hm_ntt(self.h, src.ptr, pi, dst.ptr, po, pm, n, 1 if inverse else 0, ps))
            return
        pk = None if out_packed is None else np.ascontiguousarray(np.asarray(out_packed, dtype=np.uint8))
        k5, pg = _u32(in_galois)
        d = hm_ntt_desc(src.ptr, pi, dst.ptr, po, pm, n, 1 if inverse else 0, ps, 0, None if pk is None else pk.ctypes.data_as(C.c_void_p), pg)
        self._ck(self.L.hm_ntt_ex(self.h, C.byref(d)))

    def ntt_sub_scale(self, src, minuend, out, mod_ids, k, addend=None, in_limbs=None, minuend_limbs=None, addend_limbs=None,
                      out_limbs=None):
        keep = [_u32(x) for x in (in_limbs, minuend_limbs, addend_limbs, out_limbs, mod_ids)]
        kk, pk = _u64(k)
        self._ck(self.L.hm_ntt_sub_scale(self.h, src.ptr, keep[0][1], minuend.ptr, keep[1][1], None if addend is None else addend.ptr,
                                         keep[2][1], out.ptr, keep[3][1], keep[4][1], len(mod_ids), pk))

    def ntt_mix_sub_scale(self, src, minuend, out, mod_ids, k, mix=None, mix_k=None, addend=None, addend_k=None, in_limbs=None,
                          mix_limbs=None, minuend_limbs=None, addend_limbs=None, out_limbs=None, conv=None, addend_galois=None):
        """out = (minuend - NTT(src + mix_k * mix)) * k + addend * addend_k (merged ModDown + rescale of one limb).
        conv = [(src, in_limbs, in_ids, out_limbs_of_the_fed_limb_polys, out_ids), ...]: `src` of those limb-polys is this base conversion,
        computed inside the transform's first pass (src may then be None)"""
        keep = [_u32(x) for x in (in_limbs, mix_limbs, minuend_limbs, addend_limbs, out_limbs, mod_ids)]
        ks = [_u64(x) for x in (mix_k, addend_k, k)]
        kg = _u32(addend_galois)   # per limb: read the addend through the automorphism X -> X^g
        ptr = lambda v: None if v is None else v.ptr
        descs, keep2 = None, []
        if conv:
            descs = (hm_bconv_desc * len(conv))()
            for dd, (csrc, c_in_limbs, in_ids, c_out_limbs, out_ids, *pk) in zip(descs, conv):
                arrs = [_u32(c_in_limbs), _u32(in_ids), _u32(c_out_limbs), _u32(out_ids)]
                keep2.append(arrs)
                dd.in_packed = 1 if pk and pk[0] else 0
                dd.in_, dd.in_limbs, dd.in_ids, dd.n_in = csrc.ptr, arrs[0][1], arrs[1][1], len(in_ids)
                dd.out, dd.out_limbs, dd.out_ids, dd.n_out, dd.log_len = out.ptr, arrs[2][1], arrs[3][1], len(out_ids), 0
        d = hm_ntt_fused_desc(ptr(src) if src is not None else out.ptr, keep[0][1], ptr(mix), keep[1][1], ks[0][1], minuend.ptr, keep[2][1], ptr(addend), keep[3][1], ks[1][1],
                              out.ptr, keep[4][1], keep[5][1], len(mod_ids), ks[2][1], C.cast(descs, C.c_void_p) if conv else None, len(conv) if conv else 0,
                              kg[1])
        self._ck(self.L.hm_ntt_mix_sub_scale(self.h, C.byref(d)))

    def tensor(self, a, b, c, d, o0, o1, o2, mod_ids, limbs=None):
        ls = limbs or [None] * 7
        keep = [_u32(x) for x in ls] + [_u32(mod_ids)]
        self._ck(self.L.hm_tensor(self.h, a.ptr, keep[0][1], b.ptr, keep[1][1], c.ptr, keep[2][1], d.ptr, keep[3][1], o0.ptr, keep[4][1],
                                  o1.ptr, keep[5][1], o2.ptr, keep[6][1], keep[7][1], len(mod_ids)))

    def inner_product(self, x, x_limbs, y, y_limbs, out, out_limbs, mod_ids, n_terms, n_out, x_galois=0):
        keep = [_u32(v) for v in (x_limbs, y_limbs, out_limbs, mod_ids)]
        if x_galois:   # the x operands through the automorphism X -> X^g (hm_inner_product_ex)
            d = hm_ip_desc(x.ptr, keep[0][1], y.ptr, keep[1][1], out.ptr, keep[2][1], keep[3][1], len(mod_ids), n_terms, n_out, int(x_galois))
            self._ck(self.L.hm_inner_product_ex(self.h, C.byref(d)))
            return
        self._ck(self.L.hm_inner_product(self.h, x.ptr, keep[0][1], y.ptr, keep[1][1], out.ptr, keep[2][1], keep[3][1], len(mod_ids),
                                         n_terms, n_out))

    def ntt_second_pass(self, buf, mod_ids, inverse=False, limbs=None, scale=None):
        """the last pass of transforms whose first pass another call has already run into `buf` (in place)"""
        k1, pl = _u32(limbs)
        k2, pm = _u32(mod_ids)
        k3, ps = _u64(scale)
        self._ck(self.L.hm_ntt_second_pass(self.h, buf.ptr, pl, pm, len(mod_ids), 1 if inverse else 0, ps))

    def ntt_inner_product(self, x, x_limbs, x_is_coeff, hand, hand_limbs, y, y_limbs, out, out_limbs, mod_ids, n_terms, n_out, conv=None, out_inverse=None,
                          x_galois=0):
        """out[i][k] = sum_j (NTT(x[i][j]) if x_is_coeff[i][j] else x[i][j]) * y[i][k][j]: the HPIP unit as a fused NTT-epilogue x key MAC.
        conv = [(src, in_limbs, in_ids, hand_out_limbs, out_ids), ...]: the transformed digits are these base conversions, computed inside
        the transforms' first pass (their outputs are hand-off limbs of `hand`); x_galois = g: the evaluation-form digits are read through X -> X^g"""
        keep = [_u32(v) for v in (x_limbs, hand_limbs, y_limbs, out_limbs, mod_ids)]
        flags = np.ascontiguousarray(np.asarray(x_is_coeff, dtype=np.uint8))
        descs, keep2 = None, []
        if conv:
            descs = (hm_bconv_desc * len(conv))()
            for dd, (src, in_limbs, in_ids, out_limbs_, out_ids, *pk) in zip(descs, conv):
                arrs = [_u32(in_limbs), _u32(in_ids), _u32(out_limbs_), _u32(out_ids)]
                keep2.append(arrs)
                dd.in_packed = 1 if pk and pk[0] else 0
                dd.in_, dd.in_limbs, dd.in_ids, dd.n_in = src.ptr, arrs[0][1], arrs[1][1], len(in_ids)
                dd.out, dd.out_limbs, dd.out_ids, dd.n_out, dd.log_len = hand.ptr, arrs[2][1], arrs[3][1], len(out_ids), 0
        inv = None if out_inverse is None else np.ascontiguousarray(np.asarray(out_inverse, dtype=np.uint8))
        d = hm_ntt_ip_desc(x.ptr, keep[0][1], flags.ctypes.data_as(C.c_void_p), None if hand is None else hand.ptr, keep[1][1], y.ptr, keep[2][1],
                           out.ptr, keep[3][1], keep[4][1], len(mod_ids), n_terms, n_out,
                           C.cast(descs, C.c_void_p) if conv else None, len(conv) if conv else 0,
                           None if inv is None else inv.ctypes.data_as(C.c_void_p), int(x_galois))
        self._ck(self.L.hm_ntt_inner_product(self.h, C.byref(d)))

    def automorph(self, src, dst, n, galois, in_limbs=None, out_limbs=None):
        k1, pi = _u32(in_limbs)
        k2, po = _u32(out_limbs)
        self._ck(self.L.hm_automorph(self.h, src.ptr, pi, dst.ptr, po, n, galois))

    def ewe(self, op, out, mod_ids, a=None, b=None, c=None, d=None, la=None, lb=None, lc=None, ld=None, lo=None, k=None):
        keep = [_u32(x) for x in (la, lb, lc, ld, lo, mod_ids)]
        kk, pk = _u64(k)
        ptr = lambda x: None if x is None else x.ptr
        self._ck(self.L.hm_ewe(self.h, op, ptr(a), keep[0][1], ptr(b), keep[1][1], ptr(c), keep[2][1], ptr(d), keep[3][1],
                               out.ptr, keep[4][1], keep[5][1], len(mod_ids), pk))

    def bconv(self, src, in_ids, dst, out_ids, in_limbs=None, out_limbs=None):
        k1, pil = _u32(in_limbs)
        k2, pii = _u32(in_ids)
        k3, pol = _u32(out_limbs)
        k4, poi = _u32(out_ids)
        self._ck(self.L.hm_bconv(self.h, src.ptr, pil, pii, len(in_ids), dst.ptr, pol, poi, len(out_ids)))

    def bconv_batch(self, probs, log_len=0):
        """several conversions in one launch: probs = [(src, in_limbs, in_ids, dst, out_limbs, out_ids[, in_packed]), ...]"""
        keep, descs = [], (hm_bconv_desc * len(probs))()
        for d, (src, in_limbs, in_ids, dst, out_limbs, out_ids, *pk) in zip(descs, probs):
            arrs = [_u32(in_limbs), _u32(in_ids), _u32(out_limbs), _u32(out_ids)]
            keep.append(arrs)
            d.in_packed = 1 if pk and pk[0] else 0
            d.in_, d.in_limbs, d.in_ids, d.n_in = src.ptr, arrs[0][1], arrs[1][1], len(in_ids)
            d.out, d.out_limbs, d.out_ids, d.n_out, d.log_len = dst.ptr, arrs[2][1], arrs[3][1], len(out_ids), log_len
        self._ck(self.L.hm_bconv_batch(self.h, descs, len(probs)))

    def bconv_consts(self, in_ids, out_ids):
        k1, pii = _u32(in_ids)
        k2, poi = _u32(out_ids)
        qh = np.empty(len(in_ids), dtype=np.uint64)
        tb = np.empty((len(in_ids), max(1, len(out_ids))), dtype=np.uint64)
        self._ck(self.L.hm_bconv_consts(self.h, pii, len(in_ids), poi, len(out_ids), qh.ctypes.data_as(C.c_void_p),
                                        tb.ctypes.data_as(C.c_void_p)))
        return qh, tb[:, :len(out_ids)]

    def fill_uniform(self, dst, mod_ids, seed, out_limbs=None):
        k1, pol = _u32(out_limbs)
        k2, pm = _u32(mod_ids)
        self._ck(self.L.hm_fill_uniform(self.h, dst.ptr, pol, pm, len(mod_ids), int(seed) & (2 ** 64 - 1)))

    def set_option(self, name, value):
        self._ck(self.L.hm_set_option(self.h, name.encode(), int(value)))

    def counter(self, name):
        v = C.c_uint64()
        self._ck(self.L.hm_get_counter(self.h, name.encode(), C.byref(v)))
        return v.value

    def timer_start(self):
        self._ck(self.L.hm_timer_start(self.h))

    def timer_stop(self):
        ns = C.c_uint64()
        self._ck(self.L.hm_timer_stop(self.h, C.byref(ns)))
        return ns.value
