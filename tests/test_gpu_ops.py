"""GPU parity of whole operations through the C++ host layer (Operation -> Driver -> Arch -> C ABI -> HIP
kernels) against the CPU oracle on the same seeded synthetic inputs, bit-exact.
 * unfused (fuse=0): every named stage buffer of the reference's buffer plan is compared;
 * fused (fuse=1, the bench path): the operation outputs and every buffer that still exists.
Covers BASELINE configs #1 (N=2^15, 16/10/4), #3 hmult and #4 hrotate (N=2^16, 45/35/15), beta = 1, uneven
last digits, and hadd / pmult / padd."""
import os

import numpy as np
import pytest

from oracle.homoracle import Oracle

pytestmark = pytest.mark.gpu
SEED = 0x484F4D55
_oracles = {}


def oracle(logN, L, K, chain="mont32"):
    key = (logN, L, K, chain)
    if key not in _oracles:
        _oracles[key] = Oracle(logN, L, K, chain=chain)
        _oracles[key].set_threads(8)
    return _oracles[key]


# The ops run on both arithmetic back-ends: config key `chain_bits` hands the host layer (and hm_create) the largest primes = 1 mod 2N
# below 2^bits instead of the default chain of primes h 2^32 + 1; 60 = SURVEY.md 8(d)'s chain as written, 36 = 36-bit words as upstream's
# elementBitWidth models (config/config_4.cfg:9)
def chain_overrides(chain, base=None):
    ov = dict(base or {})
    if chain != "mont32":
        ov["chain_bits"] = 60 if chain == "survey" else int(chain)
    return ov or None


def inputs(o, ell):
    return o.synth_ct(ell, SEED), o.synth_ct(ell, SEED + 2000), o.synth_evk(ell, SEED + 10000)


def check_keyswitch_buffers(op, dd, ell, K, beta, fused, hpip=True, bconv=True, ip_rows=None, packed=True):   # bconv: True | "residue" | "moddown"
    """hpip (fused only): the ModUp transforms' last pass runs inside the inner-product kernel: NTTOut_beta(j) is
    then only first-pass scratch and InnerProduceOut_Key{k} is compared with the oracle's `ip` dump instead.  bconv (fused only,
    round 4): the ModDown conversion runs inside the first pass of the transform that consumes it: ModdownBConvOut_Key{k} no longer
    exists (the results are compared through the outputs).  ip_rows (fused, round 5, N = 2^16): the rows of InnerProduceOut_Key{k} that still
    hold the evaluation-form sum — the special limbs (and an hmult's last Q limb) leave the kernel as the first pass of their inverse
    transform (pass 7b) and are compared through the outputs"""
    E = ell + K
    if not fused:
        assert np.array_equal(op.read("ModUpINTTOut"), dd["modup_intt"])
        for k in range(2):
            assert np.array_equal(op.read(f"INTTOut_ModDown_Key({k})"), dd["moddown_intt"][k])
    got_d = op.read("ModUpDecompOut")
    if fused and packed:   # pass (11): limb-polys that only base conversions read are stored split-30 packed: (x mod 2^30) | ((x >> 30) << 32)
        got_d = (got_d & np.uint64(0x3FFFFFFF)) | ((got_d >> np.uint64(32)) << np.uint64(30))
    assert np.array_equal(got_d, dd["modup_decomp"])
    for j in range(beta):
        if fused and hpip:
            continue
        got = op.read(f"NTTOut_beta({j})")
        lo, hi = j * K, min(ell, (j + 1) * K)
        sel = [t for t in range(E) if fused is False or not (lo <= t < hi)]   # fused: the digit's own limbs are aliased away
        assert np.array_equal(got[sel], dd["ext"][j][sel]), f"NTTOut_beta({j})"
    for k in range(2):
        rows_ip = slice(0, E) if ip_rows is None else slice(0, ip_rows)
        assert np.array_equal(op.read(f"InnerProduceOut_Key{k}")[rows_ip], dd["ip"][k][rows_ip]), f"InnerProduceOut_Key{k}"
        if bconv == "moddown":      # pass 9: the conversion runs inside its consumer, the buffer is never written
            continue
        got_c = op.read(f"ModdownBConvOut_Key{k}")
        rows = slice(0, ell - 1) if (fused and bconv == "residue") else slice(0, ell)   # pass 10 (hmult): the last limb's conversion goes
        assert np.array_equal(got_c[rows], dd["moddown_bconv"][k][rows])                # straight into the rescale residue
        if not fused:   # fused: the ModDown NTT output only exists inside the fused transform's epilogue
            assert np.array_equal(op.read(f"NTTOut_ModDown_Key({k})"), dd["moddown_ntt"][k])


CASES = [("config_4_N15.cfg", 15, 16, 10, 4), ("config_4_N15.cfg", 15, 8, 8, 8), ("config_4_N15.cfg", 15, 6, 5, 2),
         ("config_4.cfg", 16, 45, 35, 15)]


@pytest.mark.parametrize("chain", ["mont32", "survey", 36])
@pytest.mark.parametrize("cfg,logN,L,ell,alpha", CASES)
@pytest.mark.parametrize("fuse", [False, True, "no_hpip", "no_bconv", "moddown", "no_ip_inv", "no_pack"])
def test_hmult_bit_exact(cfg, logN, L, ell, alpha, fuse, chain):
    """fuse = True is the bench path (ModUp conversion + transforms + key MAC in one C-ABI call: k_bconv_col, k_ntt_row_ip);
    "no_bconv" = the same with the conversion as its own launch (fuse_bconv = 0); "no_hpip" = fused plan with separate ModUp
    transforms and inner product (fuse_hpip = 0)"""
    from homulator_amd import host
    if chain == 36 and (logN != 15 or fuse not in (False, True)):
        pytest.skip("the 36-bit chain runs the N = 2^15 configurations unfused and fully fused")
    o = oracle(logN, L, alpha, chain)
    ct1, ct2, evk = inputs(o, ell)
    ids = list(range(ell))
    hpip = fuse in (True, "no_bconv", "moddown", "no_ip_inv", "no_pack")   # "no_bconv": fused transform x key kernel fed by a separate conversion launch; "moddown": pass 9 on
    mode = fuse
    # pass 7b (fused transform x key kernel; N = 2^15 too since round 6): the special limbs and the last Q limb of the key-switch sum leave as the first pass of
    # their inverse transform; "no_ip_inv" keeps them in evaluation form (every row compared)
    ip_rows = ell - 1 if (hpip and fuse != "no_ip_inv") else None
    op = host.Op(cfg, "hmult", L, ell, alpha, fuse=bool(fuse),
                 overrides=chain_overrides(chain, {"fuse_hpip": 0} if fuse == "no_hpip" else {"fuse_bconv": 0} if fuse == "no_bconv" else {"fuse_moddown": 1} if fuse == "moddown" else {"fuse_ip_inv": 0} if fuse == "no_ip_inv" else {"pack_bconv_in": 0} if fuse == "no_pack" else None))
    fuse = bool(fuse)
    op.execute(1)
    assert op.backend_counter("arith") == (1 if os.environ.get("HOMULATOR_ARITH") == "generic" else 0 if chain == "mont32" else 1)
    assert np.array_equal(op.read("ct1.c0"), ct1[0]) and np.array_equal(op.read("ct2.c1"), ct2[1])
    d0 = o.ewe(0, ids, ct1[0], ct2[0])
    d1 = o.ewe(1, ids, ct1[0], ct2[1], ct1[1], ct2[0])
    d2 = o.ewe(0, ids, ct1[1], ct2[1])
    assert np.array_equal(op.read("TensorD0Out"), d0)
    assert np.array_equal(op.read("TensorD1Out"), d1)
    assert np.array_equal(op.read("TensorD2Out"), d2)
    k0, k1, dd = o.keyswitch(ell, d2, evk, dump=True)
    check_keyswitch_buffers(op, dd, ell, alpha, o.beta(ell), fuse, hpip, bconv="moddown" if mode == "moddown" else "residue" if mode in (True, "no_hpip", "no_ip_inv", "no_pack") else True, ip_rows=ip_rows, packed=mode != "no_pack")
    if not fuse:
        assert np.array_equal(op.read("KeySwitchFinalOutput_Key(0)"), k0)
        assert np.array_equal(op.read("KeySwitchFinalOutput_Key(1)"), k1)
    if not fuse:   # fused: ModDown finish and rescale are ONE transform per limb and the rescale residue is formed in
        # coefficient form, so the key-switch sum is never materialised
        assert np.array_equal(op.read("HMULTHaddOutput(0)"), o.ewe(3, ids, k0, None, d0))
        assert np.array_equal(op.read("HMULTHaddOutput(1)"), o.ewe(3, ids, k1, None, d1))
    exp = o.hmult(ell, ct1, ct2, evk, rescale=True)
    assert np.array_equal(op.read("out.c0"), exp[0])
    assert np.array_equal(op.read("out.c1"), exp[1])
    # idempotence: a second execution of the same plan reproduces the outputs (no stage clobbers an input)
    op.execute(2)
    assert np.array_equal(op.read("out.c0"), exp[0]) and np.array_equal(op.read("out.c1"), exp[1])
    op.close()


@pytest.mark.parametrize("fuse", [True, "no_bconv", "wide"])
def test_hmult_mixed_conversion_launch(fuse):
    """N = 2^16, l = 20, alpha = 16: digits of 16 and 4 limbs.  With the fused conversion capped at 15 input limbs (config key
    fuse_bconv_max_in: the plan of rounds 3-5) the 16-limb digit keeps its own conversion, so ONE transform x key launch mixes digits
    converted inside their first pass with digits that arrive converted (their first pass was skipped and the result
    silently wrong).  "wide" (round 6; fuse_bconv_max_in = 32): both digits convert inside their first pass, the 16-limb one in two input groups
    (the planner's default at N = 2^16 stops at 15 limbs, where the fused form measured faster: cap_bconv_col_pref_in)."""
    from homulator_amd import host
    L, ell, alpha = 45, 20, 16
    o = oracle(16, L, alpha)
    ct1, ct2, evk = inputs(o, ell)
    op = host.Op("config_4.cfg", "hmult", L, ell, alpha, overrides={"fuse_bconv": 0} if fuse == "no_bconv" else {"fuse_bconv_max_in": 32 if fuse == "wide" else 15})
    kinds = [ln.split()[0] + ":" + ln.split()[1] for ln in op.plan()]
    assert any(k.startswith("BCONV:ModUp_BCONV") for k in kinds) == (fuse != "wide"), kinds
    op.execute(1)
    ids = list(range(ell))
    d2 = o.ewe(0, ids, ct1[1], ct2[1])
    k0, k1, dd = o.keyswitch(ell, d2, evk, dump=True)
    for k in range(2):   # (rows [0, l - 1): the last Q limb and the special limbs leave as the first pass of their inverse transform, pass 7b)
        assert np.array_equal(op.read(f"InnerProduceOut_Key{k}")[:ell - 1], dd["ip"][k][:ell - 1]), f"InnerProduceOut_Key{k}"
    exp = o.hmult(ell, ct1, ct2, evk, rescale=True)
    assert np.array_equal(op.read("out.c0"), exp[0]) and np.array_equal(op.read("out.c1"), exp[1])
    op.close()


@pytest.mark.parametrize("cfg,logN,L,ell,alpha", CASES)
@pytest.mark.parametrize("fuse", [False, True, "auto-launch"])
@pytest.mark.parametrize("chain", ["mont32", "survey"])
def test_hrotate_bit_exact(cfg, logN, L, ell, alpha, fuse, chain):
    """fuse = True: the default plan — since round 6 (pass 12) the ModUp INTT, the key product and the final add read the ciphertext THROUGH the
    automorphism and neither AUTOOutput is ever written; "auto-launch": the same plan with both automorphisms as a launch (fuse_auto = 0)"""
    from homulator_amd import host
    o = oracle(logN, L, alpha, chain)
    ct1, _, evk = inputs(o, ell)
    ov = dict(chain_overrides(chain) or {})
    if fuse == "auto-launch":
        ov["fuse_auto"] = 0
    op = host.Op(cfg, "hrotate", L, ell, alpha, fuse=bool(fuse), overrides=ov or None)
    folded = any(" auto_addend=" in ln for ln in op.plan())
    assert folded == (fuse is True)
    launched = " ".join(ln for ln in op.plan() if ln.startswith("AUTO"))
    assert ("AUTO_Key(1)" in launched) == (fuse is not True)   # (a one-digit key switch too: its Q limbs' plain inner products gather as well)
    op.execute(1)
    r0, r1 = o.automorph_eval(ct1[0], 5), o.automorph_eval(ct1[1], 5)
    if "AUTO_Key(0)" in launched:
        assert np.array_equal(op.read("AUTOOutput(0)"), r0)
    if "AUTO_Key(1)" in launched:
        assert np.array_equal(op.read("AUTOOutput(1)"), r1)
    k0, k1, dd = o.keyswitch(ell, r1, evk, dump=True)
    check_keyswitch_buffers(op, dd, ell, alpha, o.beta(ell), fuse, ip_rows=ell if fuse else None)   # 7b: the special limbs
    exp = o.hrotate(ell, ct1, 5, evk)
    assert np.array_equal(op.read("out.c0"), exp[0])
    assert np.array_equal(op.read("out.c1"), exp[1])
    op.close()


def test_hrotate_other_galois_element():
    from homulator_amd import host
    o = oracle(15, 6, 2)
    ct1, _, evk = inputs(o, 5)
    g = pow(5, 3, 2 * o.N)
    op = host.Op("config_4_N15.cfg", "hrotate", 6, 5, 2, overrides={"galois": g})
    op.execute(1)
    exp = o.hrotate(5, ct1, g, evk)
    assert np.array_equal(op.read("out.c0"), exp[0]) and np.array_equal(op.read("out.c1"), exp[1])
    op.close()


@pytest.mark.parametrize("chain", ["mont32", "survey"])
@pytest.mark.parametrize("cfg,logN,L,ell,alpha", [("config_4_N15.cfg", 15, 16, 10, 4), ("config_4.cfg", 16, 45, 35, 15)])
def test_hadd_pmult_padd_bit_exact(cfg, logN, L, ell, alpha, chain):
    from homulator_amd import host
    o = oracle(logN, L, alpha, chain)
    ct1, ct2, _ = inputs(o, ell)
    pt = o.fill_uniform(list(range(ell)), SEED + 4000)
    for name, exp in (("hadd", o.hadd(ell, ct1, ct2)), ("pmult", o.pmult(ell, ct1, pt)), ("padd", o.padd(ell, ct1, pt))):
        op = host.Op(cfg, name, L, ell, alpha, overrides=chain_overrides(chain))
        op.execute(1)
        assert np.array_equal(op.read("out.c0"), exp[0]), name
        assert np.array_equal(op.read("out.c1"), exp[1]), name
        op.close()


def test_cli_runs_on_gpu_and_prints_upstream_format():
    import os, subprocess
    from homulator_amd import host
    cli = os.path.join(host.ROOT, "host", "Homulator.run")
    r = subprocess.run([cli, os.path.join(host.CONFIG_DIR, "config_4_N15.cfg"), "hmult", "16", "10", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Welcome! Start simulating HMULT!" in r.stdout and "Remaining 0 instructions!" in r.stdout
    stat = r.stdout.split("Start outPut statistic informations:")[1]
    assert "NTT_(0) :\t" in stat and "BCONV_(0) :\t" in stat and "EWE_(0) :\t" in stat


def test_graph_replay_matches_direct_launches():
    """config key graph = 1: the launch plan is captured into a HIP graph on the second run and replayed afterwards"""
    from homulator_amd import host
    o = oracle(15, 6, 2)
    ct1, ct2, evk = inputs(o, 5)
    exp = o.hmult(5, ct1, ct2, evk)
    op = host.Op("config_4_N15.cfg", "hmult", 6, 5, 2, overrides={"graph": 1})
    for _ in range(4):   # run 1 direct, run 2 captures + launches, runs 3-4 replay
        op.execute(1)
        assert np.array_equal(op.read("out.c0"), exp[0]) and np.array_equal(op.read("out.c1"), exp[1])
    op.close()


BATCH_SEED_STRIDE = 100000   # host/src/Arch.cpp kBatchSeedStride


@pytest.mark.parametrize("cfg,logN,L,ell,alpha,batch", [("config_4_N15.cfg", 15, 6, 5, 2, 3), ("config_4_N15.cfg", 15, 16, 10, 4, 2),
                                                         ("config_4.cfg", 16, 45, 35, 15, 3)])
@pytest.mark.parametrize("opname", ["hmult", "hrotate"])
def test_batched_ops_bit_exact(cfg, logN, L, ell, alpha, batch, opname):
    """config key batch = B: every launch carries B independent ops (own inputs, ONE evaluation key); op c must equal
    the oracle on the inputs of seed + c * stride, and the stage byte / instruction accounting scales by B"""
    from homulator_amd import host
    o = oracle(logN, L, alpha)
    evk = o.synth_evk(ell, SEED + 10000)
    single = host.Op(cfg, opname, L, ell, alpha, backend=host.BACKEND_COUNT)
    op = host.Op(cfg, opname, L, ell, alpha, overrides={"batch": batch})
    assert op.batch == batch
    op.execute(2)
    assert op.launch_count() == single.launch_count()
    assert op.stage_bytes() == batch * single.stage_bytes()
    for c in range(batch):
        s = SEED + c * BATCH_SEED_STRIDE
        ct1, ct2 = o.synth_ct(ell, s), o.synth_ct(ell, s + 2000)
        assert np.array_equal(op.read("ct1.c1", copy=c), ct1[1])
        exp = o.hmult(ell, ct1, ct2, evk, rescale=True) if opname == "hmult" else o.hrotate(ell, ct1, 5, evk)
        assert np.array_equal(op.read("out.c0", copy=c), exp[0]), f"copy {c}"
        assert np.array_equal(op.read("out.c1", copy=c), exp[1]), f"copy {c}"
    with pytest.raises(host.HostError):
        op.read("out.c0", copy=batch)
    op.close()
    single.close()


def test_concurrent_instances_stay_bit_exact():
    """bench.py's throughput mode: instances with their own HBM pool and HIP stream run concurrently on one GPU; their
    kernels interleave on the chip, the results must not (200 interleaved enqueues, outputs checked afterwards)"""
    from homulator_amd import host
    o = oracle(15, 16, 4)
    evk = o.synth_evk(10, SEED + 10000)
    ops = [host.Op("config_4_N15.cfg", "hmult", 16, 10, 4, overrides={"seed": SEED + 7 * i, "batch": 1 + i}) for i in range(3)]
    for it in range(200):
        ops[it % 3].enqueue(1)
    for op in ops:
        op.sync()
    for i, op in enumerate(ops):
        for c in range(op.batch):
            s = SEED + 7 * i + c * BATCH_SEED_STRIDE
            exp = o.hmult(10, o.synth_ct(10, s), o.synth_ct(10, s + 2000), o.synth_evk(10, SEED + 7 * i + 10000))
            assert np.array_equal(op.read("out.c0", copy=c), exp[0]) and np.array_equal(op.read("out.c1", copy=c), exp[1]), (i, c)
        op.close()


def test_bench_timed_region_configuration_is_bit_exact():
    """exactly what bench.py times: BASELINE configs[1] (config_4.cfg hmult 45/35/15, N = 2^16), two instances in flight with their own
    pool and stream, 10 hmults per launch, the plan replayed as a HIP graph, enqueued alternately; afterwards every one of the 20 ops
    equals the oracle on its own inputs"""
    from homulator_amd import host
    L, ell, alpha = 45, 35, 15
    o = oracle(16, L, alpha)
    o.set_threads(16)
    ops = [host.Op("config_4.cfg", "hmult", L, ell, alpha, overrides={"seed": SEED + 7 * i, "batch": 10, "graph": 1}) for i in range(2)]
    for it in range(12):   # per instance: direct, capture, then four replays
        ops[it % 2].enqueue(1)
    for op in ops:
        op.sync()
    for i, op in enumerate(ops):
        evk = o.synth_evk(ell, SEED + 7 * i + 10000)
        for c in range(op.batch):
            s = SEED + 7 * i + c * BATCH_SEED_STRIDE
            exp = o.hmult(ell, o.synth_ct(ell, s), o.synth_ct(ell, s + 2000), evk, rescale=True)
            assert np.array_equal(op.read("out.c0", copy=c), exp[0]) and np.array_equal(op.read("out.c1", copy=c), exp[1]), (i, c)
        op.close()
