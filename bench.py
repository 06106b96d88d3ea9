#!/usr/bin/env python3
"""bench.py — hmult + key-switch throughput on MI355X (BASELINE.json metric), one JSON line on stdout.

  python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full hmult (tensor -> hybrid key switch -> add -> rescale) at config_4.cfg, L=45, l=35,
alpha=15 (BASELINE configs[2]) over synthetic limbs already resident in HBM.  The op runs through the same path
as the CLI: C++ Operation/Driver -> C ABI -> HIP kernels.  `roofline` is measured live with HIP events on the
backend's own stream; `cpu_baseline` times the CPU oracle ("port": the reference has no arithmetic to time,
SURVEY.md §0) on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG, OP, L, ELL, ALPHA = "config_4.cfg", "hmult", 45, 35, 15
LOGN = 16
LP = (1 << LOGN) * 8                       # one limb-poly, bytes
HMULT_ALG_BYTES = 1_102_577_664            # 2 103 LP
HROTATE_ALG_BYTES = 883_425_280            # 1 685 LP
NTT_ALG_BYTES = 2 * LP                     # per limb-NTT
EVK_BYTES = 2 * 3 * 50 * LP                # evaluation key of configs[2]: 2 beta E = 300 limb-polys = 157 286 400 B
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md
BFLY_PER_LIMB_NTT = (1 << LOGN) // 2 * LOGN  # butterflies of one limb-NTT: N/2 per stage, logN stages


def roofline_inputs():
    """figures that come from profiler runs of HEAD, committed under profiles/ (never typed into this file):
    `ntt_sweep50_traffic_bytes` = HBM bytes of the 50-limb forward sweep from the PMC counters (separate --pmc passes,
    FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950); `wave_butterfly_ns` = full-chip time per
    wave-butterfly of the shipped butterfly mix (tools/bflyrate: 1024 SIMDs busy, no memory traffic)."""
    try:
        with open(os.path.join(ROOT, "profiles", "roofline_inputs.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _warm(ctx, launch, seconds=0.25):
    """the same launches, untimed, for a quarter of a second: the chip ramps its clocks over the first ~100 ms of load (the timed
    region of the op gets the same treatment in main())"""
    t0, i = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            launch(i)
            i += 1
        ctx.sync()


def measure_ntt_inop(batch, iters=12, sets=2):
    """the ModUp forward transforms at the launch shape of the timed region: the 115 converted limb-polys of each of `batch` ops
    (sum over digits of E - d_j, same-modulus limb-polys of the ops and digits side by side) in ONE hm_ntt call, timed alone
    with HIP events on the backend's stream; rotating over two buffer pairs.  Returns (ns per call, limb-polys per call)."""
    from homulator_amd import hip
    ctx = hip.Context(LOGN, L, ALPHA)
    ext = ctx.ext_ids(ELL)
    ids = []
    for j in range(-(-ELL // ALPHA)):
        own = set(range(j * ALPHA, min(ELL, (j + 1) * ALPHA)))
        ids += [m for t, m in enumerate(ext) if t not in own]
    ids = ids * batch
    n = len(ids)
    bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(sets)]
    for i, (a, _) in enumerate(bufs):
        ctx.fill_uniform(a, ids, 1 + i)
    _warm(ctx, lambda i: ctx.ntt(*bufs[i % sets], ids))
    ctx.timer_start()
    for i in range(iters):
        a, b = bufs[i % sets]
        ctx.ntt(a, b, ids)
    ns = ctx.timer_stop() / iters
    for a, b in bufs:
        a.free(); b.free()
    ctx.close()
    return ns, n


def measure_ntt_sweep(n_limbs, iters=48, sets=6, device=0, which=None):
    """forward NTT sweep over the extended basis (l + alpha limbs): device time per sweep, HIP events on the
    backend stream.  One sweep = one hm_ntt call = ONE launch (k_ntt_fused8: both passes, the hand-off through the XCD's L2 behind a
    rendezvous on XCD-local atomics; launches above 96 limb-polys run the two pass kernels k_ntt_col + k_ntt_row).
    The sweeps rotate over `sets` input/output buffer pairs (6 x 2 x 26 MB = 315 MB > the 256 MiB Infinity Cache), so
    that every sweep reads its input from HBM rather than from a cache the previous replay left warm.
    N > 1 (round 6): every rank sweeps the limbs of the extended basis it owns (limb e -> e % N), all ranks at once; `sets` grows so that a
    rank's rotation still exceeds its Infinity Cache."""
    from homulator_amd import hip
    ctx = hip.Context(LOGN, L, ALPHA, device=device)
    ids = ctx.ext_ids(ELL)[:n_limbs] if which is None else [ctx.ext_ids(ELL)[e] for e in which]   # which: a rank's own limbs of the extended basis (N > 1)
    n_limbs = len(ids)
    bufs = [(ctx.alloc(n_limbs), ctx.alloc(n_limbs)) for _ in range(sets)]
    for i, (a, _) in enumerate(bufs):
        ctx.fill_uniform(a, ids, 1 + i)
    _warm(ctx, lambda i: ctx.ntt(*bufs[i % sets], ids))
    ctx.timer_start()
    for i in range(iters):
        a, b = bufs[i % sets]
        ctx.ntt(a, b, ids)
    ns = ctx.timer_stop() / iters
    # the same sweep IN PLACE (hm_ntt allows in == out), rotating over the 12 buffers (315 MB): the hand-off stores land on lines the input
    # loads have just brought into the L2, so its loads need no fill from memory (information beside the contract's out-of-place figure)
    flat = [x for pair in bufs for x in pair]
    for i, x in enumerate(flat):
        ctx.fill_uniform(x, ids, 31 + i)
    _warm(ctx, lambda i: ctx.ntt(flat[i % len(flat)], flat[i % len(flat)], ids))
    ctx.timer_start()
    for i in range(iters):
        ctx.ntt(flat[i % len(flat)], flat[i % len(flat)], ids)
    ns_inplace = ctx.timer_stop() / iters
    cross = ctx.counter("ntt_cross_xcd")   # limb-polys whose workgroups were spread over XCDs (agent-scope path of the rendezvous): 0 expected
    one_launch = ctx.counter("ntt_fused_small")
    for a, b in bufs:
        a.free(); b.free()
    ctx.close()
    return ns, cross, one_launch, ns_inplace


def measure_second_op(opn, streams, batch, steps, device, extra=None):
    """a second figure beside the headline, through the same instances x batch x HIP-graph shape as the timed region of the headline op and
    timed the same way (wall clock around enqueue + sync); own instances, closed afterwards.  opn = "hrotate": BASELINE configs[3];
    extra = {"chain_bits": 60}: the headline op on SURVEY.md 8(d)'s chain as written (the generic arithmetic back-end)"""
    from homulator_amd import host
    extra = dict(extra or {})
    ops = [host.Op(CFG, opn, L, ELL, ALPHA, device=device,
                   overrides={"seed": host.SEED + 31 * (i + 1), **({"batch": batch} if batch > 1 else {}), "graph": 1, **extra}) for i in range(streams)]
    tail = host.Op(CFG, opn, L, ELL, ALPHA, device=device, overrides={"seed": host.SEED + 998, **extra}) if batch > 1 else None

    def run(n):
        for i in range(n // batch):
            ops[i % streams].enqueue(1)
        if n % batch:
            (tail if tail is not None else ops[0]).enqueue(n % batch if tail is not None else 1)

    def sync_all():
        for o in ops:
            o.sync()
        if tail is not None:
            tail.sync()
    if tail is not None:   # (before the pre-warm: the one-op instance's pool and key copy must not sweep the caches right in front of the timed regions)
        tail.enqueue(1)
    sync_all()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < float(os.environ.get("HOMULATOR_PREWARM_S", "0.6")):   # plan, graph capture, clock ramp: the same pre-warm the headline gets
        run(streams * batch)
        sync_all()
    dts = []
    for _ in range(3):   # three timed regions of `steps` ops; the median counts (a single 20-step region is 4 ms)
        t0 = time.perf_counter()
        run(steps)
        sync_all()
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[1]
    cross = sum(o.backend_counter("ntt_cross_xcd") for o in ops + ([tail] if tail is not None else []))
    arith = ops[0].backend_counter("arith")
    launches = ops[0].launch_count()
    auto_launches = sum(1 for ln in ops[0].plan() if ln.startswith("AUTO"))
    for o in ops:
        o.close()
    if tail is not None:
        tail.close()
    ms = dt / steps * 1e3
    if opn in ELEMENTWISE_LP:   # the single-stage element-wise ops: operands + results, no key
        alg = ELEMENTWISE_LP[opn] * ELL * LP
        return {"workload": f"{CFG} {opn} L={L} l={ELL}", "ops_per_s": steps / dt, "ops_per_s_min_median_max": [steps / max(dts), steps / dt, steps / min(dts)], "regions": 3,
                "us_per_op": ms * 1e3, "steps": steps, "streams": streams, "batch": batch, "launches_per_op": launches, "algorithmic_bytes": alg,
                "achieved_gbs": alg / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    alg = HMULT_ALG_BYTES if opn == "hmult" else HROTATE_ALG_BYTES
    evk_once = alg - EVK_BYTES * (1 - 1 / batch)
    return {"workload": f"{CFG} {opn} L={L} l={ELL} alpha={ALPHA}" + (" (BASELINE configs[3]: automorphism + full hybrid key switch)" if opn == "hrotate" else ""),
            "moduli": MODULI_NOTE[arith], "ops_per_s": steps / dt, "ops_per_s_min_median_max": [steps / max(dts), steps / dt, steps / min(dts)], "regions": 3,
            "ms_per_step": ms, "steps": steps, "streams": streams, "batch": batch, "launches_per_op": launches,
            **({"automorphism_launches": auto_launches, "automorphism_note": "the ModUp INTT, the key product and the final add read the ciphertext THROUGH the automorphism "
                "(planner pass 12, config key fuse_auto): no AUTO launch on one GPU; the algorithmic bytes still charge SURVEY 8(d)'s automorphism stage"} if opn == "hrotate" else {}),
            "frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_evk_once": evk_once / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ntt_cross_xcd": cross}


def cpu_baseline(opn="hmult", runs=5):
    """the CPU functional path (the oracle: the reference has no arithmetic to time) on the identical
    inputs, single thread and all host cores, median of `runs` after one warm-up each."""
    from oracle.homoracle import Oracle
    from homulator_amd import host
    cores = min(16, os.cpu_count() or 1)
    o = Oracle(LOGN, L, ALPHA)
    ct1, ct2, evk = o.synth_ct(ELL, host.SEED), o.synth_ct(ELL, host.SEED + 2000), o.synth_evk(ELL, host.SEED + 10000)
    run = (lambda: o.hmult(ELL, ct1, ct2, evk)) if opn == "hmult" else (lambda: o.hrotate(ELL, ct1, 5, evk))

    def median_s(threads):
        o.set_threads(threads)
        run()  # warm
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            run()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]
    multi, single = median_s(cores), median_s(1)
    return {"value": 1.0 / multi, "unit": "ops/s", "cores": cores, "kind": "port",
            "single_thread_value": 1.0 / single,
            "note": "the port is plain scalar C (oracle/homoracle.c): 64 x 64 -> 128-bit products through unsigned __int128 with a Shoup / Barrett reduction per product, fully "
                    "reduced values everywhere (no lazy ranges), no vector code, radix-2 transforms, OpenMP over limbs only.  A reported baseline, not a tuned CPU "
                    "library: the GPU / CPU ratio says nothing about kernel quality (the roofline fractions do)",
            "sample": f"median of {runs} full {opn} ops (N=2^16, l=35, alpha=15) on the CPU oracle after one warm-up: "
                      f"{cores} threads (OpenMP over limbs) = `value`, 1 thread = `single_thread_value`"}


EXCHANGE_KINDS = ("EXCH_IN", "EXCH_OUT", "REPLICATE", "EXCH_IN_COL", "EXCH_OUT_COL")   # the launch kinds that are collectives (host/src/Arch.cpp kLaunchKindNames)


def sharded_plan_note(op, world, batch):
    """which of the two sharded plans the host layer chose for this rank count (config key shard_plan), read off the op's own plan"""
    kinds = [ln.split()[0] for ln in op.plan()]
    n_coll = sum(1 for k in kinds if k in EXCHANGE_KINDS)
    if "BCONV_COL" not in kinds and not any(k.startswith("EXCH") for k in kinds):
        return (f"limbs sharded over {world} GPUs (limb e -> e % {world}), gather plan: the inputs of both base conversions and the rescale residue are "
                f"all-gathered over RCCL ({n_coll} collectives per key switch), every rank then runs the one-GPU fused kernels on the limbs it owns; "
                f"{batch} hmults per launch share the collectives")
    return (f"limbs sharded over {world} GPUs (limb e -> e % {world}), all-to-all plan: RCCL all-to-all around every digit's ModUp conversion and around the "
            f"ModDown conversion on column slices (2 beta + 2 per key switch, digit j+1's exchange on the exchange stream beside digit j's conversion and "
            f"transform) + replicate of the rescale residue ({n_coll} collectives per key switch); {batch} hmults per launch share the exchanges")


def exchange_overlap(stage_rows, ops_per_launch, us_per_step, instances, pipelined):
    """hidden / exposed exchange time per op.  The stage rows time every launch ALONE (an exchange launch together with the wait for
    its own mark).  What the compute launches alone do not explain of the step time is the exposed exchange time; the rest of the
    exchange time ran beside compute: with per-digit pipelined exchanges (the default when sharded: digit j+1's all-to-all on the
    context's exchange stream while digit j converts and transforms) already inside ONE instance, with --sharded-streams 2 also across
    instances.  One instance without pipelining hides nothing by construction."""
    ex = sum(ns for kind, _, ns in stage_rows if kind in EXCHANGE_KINDS) * 1e-3 / ops_per_launch
    comp = sum(ns for kind, _, ns in stage_rows if kind not in EXCHANGE_KINDS) * 1e-3 / ops_per_launch
    n_coll = sum(1 for kind, _, _ in stage_rows if kind in EXCHANGE_KINDS)
    if instances <= 1 and not pipelined:
        return {"hidden_us_per_op": 0.0, "exposed_us_per_op": round(ex, 2), "instances_in_flight": 1, "pipelined_per_digit": False, "collectives_per_launch": n_coll,
                "note": "one sharded instance, bulk-synchronous exchanges on the op's own stream: nothing is hidden (DESIGN.md section 7)"}
    exposed = min(ex, max(0.0, us_per_step - comp))
    return {"hidden_us_per_op": round(ex - exposed, 2), "exposed_us_per_op": round(exposed, 2), "instances_in_flight": instances,
            "pipelined_per_digit": bool(pipelined), "collectives_per_launch": n_coll,
            "note": "estimate: step time minus the compute launches timed alone = exposed exchange time; exchanges run on the context's exchange stream, ordered against compute by marks"}


def roofline_op(batched_rows, batch, rin):
    """NTT_IP = k_bconv_col<n_in> (conversion + first pass) + k_ntt_row_ip (second pass + key MAC, both keys, all digits; round 5: + the first
    inverse pass of the ModDown on the 15 special limbs and the last Q limb).  Algorithmic limb-polys per op at 45/35/15 (SURVEY.md 8d
    conventions: every operand read once, every result written once; beta = 3, E = 50, l = 35): conversion inputs 35 + hand-off written and
    read 2 x 115 + own-digit limbs 35 + outputs 2 x 50 = 400 per op, + the key 2 x 3 x 50 = 300 ONCE per launch (the figure of rounds 3 / 4,
    kept so that the fractions stay comparable)."""
    if not batched_rows:
        return None
    us = [ns * 1e-3 / batch for kind, _, ns in batched_rows if kind == "NTT_IP"]
    if not us:
        return None
    alg = (400 + 300 / batch) * LP
    pmc = rin.get("ntt_ip_bytes_per_op") if rin.get("whole_op_batch") == batch else None
    return {"kernel": "NTT_IP launch = k_bconv_col2<15, 8, false, true> + k_bconv_col2<5, 8, false, true> + k_ntt_row_ip<2, 2> (ModUp conversion from split-30 packed inputs + transforms + key MAC + the ModDown's first inverse pass on the special limbs, in one C-ABI call)", "bound": "valu",
            "us_per_op": us[0], "algorithmic_bytes_per_op_evk_once": alg, "achieved": alg / (us[0] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg / (us[0] * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": pmc,
            "traffic_frac_of_peak": None if not pmc else pmc / (us[0] * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "valu_wave_instructions_per_op": rin.get("ntt_ip_valu_per_op"), "source": rin.get("whole_op_source"),
            "note": "timed alone on the chip (stage_us_per_op_batched); `bound`: VALU issue — per-kernel SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES, wait and LDS counters of the batched op at this round's HEAD: profiles/r06_pmc_kernels_batch10.txt"}


# limb-polys per level of the reference's single-stage ops (src/Operation.cpp:1114-1176 HADD, :1455-1523 PMULT, :1618-1680 PADD): every operand read
# once, every result written once — hadd: two ciphertexts in, one out; pmult: a ciphertext and a plaintext in, a ciphertext out; padd: the
# plaintext joins c0, c1 is carried over
ELEMENTWISE_LP = {"hadd": 6, "pmult": 5, "padd": 5}
DEFAULT_BATCH = 10
# which arithmetic back-end the context chose for the chain it was given (hm_get_counter "arith")
MODULI_NOTE = {
    0: "mont32: the default chain, 45 + 15 largest primes h*2^32 + 1 below 2^60 (word-wise Montgomery reduction, q^-1 = 1 mod 2^32: six 32-bit multiplies per butterfly, one-word twiddles; libhm_m32.so)",
    1: "generic: SURVEY.md 8(d)'s chain as written, the 45 + 15 largest primes = 1 mod 2N below 2^60 (Shoup / Barrett arithmetic for any NTT-friendly chain below 2^60: nine multiplies per butterfly, two-word twiddles; libhm_gen.so)",
}


def pick_batch(steps, streams, default=DEFAULT_BATCH):
    """ops per launch: a FIXED default (10), whatever --steps is; only when K is too small to give every in-flight instance one
    full launch is the batch cut to K // streams.  Steps that do not fill a launch run through the one-op instance, so exactly K
    hmults are timed.  (Round 2 searched for a batch that divides K: the reported rate then hinged on --steps.)"""
    return max(1, min(default, steps // max(1, streams)))


def loaded_hip_library():
    """the backend build this process really mapped (an A/B build named by HOMULATOR_HIP_LIB must show up here, not beside the in-tree one)"""
    libs = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libhomulator_hip" in ln or "/libhm_" in ln})
    return [os.path.relpath(p, ROOT) if p.startswith(ROOT) else p for p in libs]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: this process starts the N ranks itself — fresh child processes with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, the same argv — relays rank 0's ONE JSON line and returns the worst exit code.  Decided
    before anything touches the GPU: this parent never imports torch and never initialises HIP (a process that has must not start or become
    another program on this pool).  Under torch.distributed.run (WORLD_SIZE set) nothing of this runs."""
    import socket
    import subprocess
    with socket.socket() as sk:   # a free rendezvous port on the loop-back interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0) or None))
    import threading
    import time as _t
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # like torch.distributed.run: a rank that fails takes the job down (the others would wait for it in a collective for ever)
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            _t.sleep(2.0)   # let the failing rank's peers print what they know
            for p in procs:
                if p.poll() is None:
                    p.kill()   # exactly the processes started above
            break
        _t.sleep(0.05)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    for ln in "".join(c for c in chunks if c).splitlines():   # ONE JSON line on stdout; what libraries print there (gloo's connection notice) goes to stderr
        print(ln, file=sys.stdout if ln.startswith("{") else sys.stderr)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc]
    if bad:
        print(f"[bench] ranks failed (rank, exit code): {bad}", file=sys.stderr)
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--sharded-streams", type=int, default=1,
                    help="N > 1 only: sharded instances in flight per rank, each with its own communicator (default 1)")
    ap.add_argument("--streams", type=int, default=2,
                    help="independent hmult instances in flight on one GPU (own inputs, HBM pool and HIP stream each); "
                         "the K timed steps are dealt round-robin over them.  1 = one op at a time (latency mode)")
    ap.add_argument("--batch", type=int, default=0,
                    help="independent hmults carried by every launch of an instance (config key `batch`: own inputs, one "
                         "evaluation key); a step is still ONE hmult, an enqueue advances `batch` steps.  0 = the fixed "
                         "default of 10 (pick_batch)")
    ap.add_argument("--op", default=OP, choices=["hmult", "hrotate"],
                    help="hmult = BASELINE.json's metric (default); hrotate = BASELINE configs[3], for information")
    ap.add_argument("--graph", type=int, default=1,
                    help="1 GPU: the throughput instances replay their launch plan as ONE captured HIP graph per enqueue (config key `graph`): "
                         "the host then takes ~20 us per launch set instead of ~0.3 ms of argument preparation, which matters for a 20-step "
                         "timed region (one launch set per instance); the one-op-at-a-time instance always enqueues directly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:   # started plainly (`python bench.py --gpus N`): be the launcher
        raise SystemExit(self_launch(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    # rehearsal on a one-GPU box (never the driver's configuration): HOMULATOR_DIST_BACKEND=gloo puts every rank on GPU 0
    # and carries the exchanges over gloo, because RCCL refuses two ranks per device
    rehearsal = world > 1 and os.environ.get("HOMULATOR_DIST_BACKEND", "nccl") == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    red_dev = "cpu" if rehearsal else "cuda"
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from homulator_amd import host
    # N > 1: ONE hmult whose limb-polys are sharded over the N GPUs (limb e -> e % N), RCCL all-to-all around the two
    # base conversions + one replicate in the rescale: strong scaling of the op's latency
    # sharded: one instance by default; --sharded-streams 2 keeps two sharded instances in flight, each with its OWN communicator and
    # stream, so that the exchanges of one run while the other computes (opt-in: not yet measured on a node)
    streams = args.streams if world == 1 else max(1, args.sharded_streams)
    # sharded: the ops of a batch share the exchanges around each base conversion; a batch that divides --steps, so that
    # exactly --steps hmults are timed without a second (communicating) instance for the remainder
    if world == 1:
        batch = args.batch if args.batch > 0 else pick_batch(args.steps, streams)
    else:   # per-rank launches shrink with the rank count: a proportionally larger batch keeps them (and the exchanges) big
        target = max(1, min(16, (args.batch or 4) * world // 2))
        batch = max(d for d in range(1, target + 1) if args.steps % d == 0)
    opn = args.op
    alg_bytes = HMULT_ALG_BYTES if opn == "hmult" else HROTATE_ALG_BYTES
    ops = [host.Op(CFG, opn, L, ELL, ALPHA, device=local_rank, rank=rank, world=world,
                   overrides={"seed": host.SEED + 7 * i, **({"batch": batch} if batch > 1 else {}), **({"graph": 1} if world == 1 and args.graph else {})})
           for i in range(streams)]
    op = ops[0]
    # steps that do not fill a batch run through a one-op instance, so that EXACTLY --steps hmults are timed
    tail_op = host.Op(CFG, opn, L, ELL, ALPHA, device=local_rank, overrides={"seed": host.SEED + 999}) if batch > 1 and world == 1 else None
    transport = "none"
    if world > 1:
        from homulator_amd import dist as hdist
        transport = "gloo-rehearsal" if rehearsal else os.environ.get("HOMULATOR_TRANSPORT", "rccl")
        if transport == "rccl":
            # the HIP library's own RCCL communicator: ncclSend/ncclRecv groups on its stream (the product path).
            # init_rccl agrees on go / no-go BEFORE the collective ncclCommInitRank (every rank enters it or none does)
            # and again after it; a failure after the GPU call is fatal (no in-process retry).
            for o in ops:   # one communicator per instance, created in the same order on every rank
                if not hdist.init_rccl(o, device=red_dev):
                    print(f"[bench] rank {rank}: no RCCL unique id could be drawn; using torch.distributed NCCL staging", file=sys.stderr)
                    transport = "torch-nccl-staging"
                    break
        if transport != "rccl":
            trs = [hdist.GlooTransport() if rehearsal else hdist.TorchNcclTransport() for _ in ops]   # kept alive: the C side holds the callbacks
            for o, tr in zip(ops, trs):
                o.comm_init_external(tr.cfunc)

    def run(n):   # n hmult steps, round-robin over the in-flight instances; asynchronous
        for i in range(n // batch):
            ops[i % streams].enqueue(1)
        if n % batch:
            if tail_op is not None:
                tail_op.enqueue(n % batch)
            else:   # sharded warm-up only (the timed step count is a multiple of the batch): one more full batch
                ops[0].enqueue(1)

    def sync_all():
        for o in ops:
            o.sync()
        if tail_op is not None:
            tail_op.sync()

    # untimed: the chip ramps its clocks over the first ~100 ms of load; a 20-step timed region is 5 ms.  Run the same
    # steps for about 0.3 s first, then the W warm-up steps the contract asks for.  Sharded runs MUST do the same number
    # of passes on every rank (each pass enters the exchanges): rank 0 times a probe pass and broadcasts the count.
    run(streams * batch)
    sync_all()
    t_pre = time.perf_counter()
    run(streams * batch)
    sync_all()
    n_pre = max(1, min(1000, int(float(os.environ.get("HOMULATOR_PREWARM_S", "0.6")) / max(time.perf_counter() - t_pre, 1e-4))))
    if dist is not None:
        cnt = torch.tensor([n_pre], device=red_dev)
        dist.broadcast(cnt, src=0)
        n_pre = int(cnt.item())
    for _ in range(n_pre):
        run(streams * batch)
    if tail_op is not None:   # (the one-op instance's plan and tables: before the warm-up steps, so that the W steps are the last untimed work — run behind
        tail_op.enqueue(1)    # them, its 1 GB pool and own key copy swept the caches right in front of the timed region: its first launches ran 3-4 % slow)
    sync_all()
    run(max(args.warmup, streams * batch))
    sync_all()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    sync_all()
    barrier()
    dt = time.perf_counter() - t0
    # two more regions of the same K steps, timed the same way: min / median / max of the three beside `value` (which stays the contract's
    # single region above: K = 20 is one 3.7 ms graph launch per instance, and it moves by +-4 % from region to region)
    region_rates = [args.steps / dt]
    for _ in range(2):
        barrier()
        t1 = time.perf_counter()
        run(args.steps)
        sync_all()
        barrier()
        d1 = time.perf_counter() - t1
        if dist is not None:
            tr = torch.tensor([d1], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
            d1 = float(tr.item())
        region_rates.append(args.steps / d1)
    # the same steady state over ten launches per instance (information: the contract's K = 20 is one launch per instance)
    sustained = None
    if world == 1:
        n_sus = 10 * streams * batch
        barrier()
        t1 = time.perf_counter()
        run(n_sus)
        sync_all()
        barrier()
        sustained = n_sus / (time.perf_counter() - t1)
    single = None
    if world == 1 and (streams > 1 or batch > 1):   # latency mode beside it: one op at a time
        lat_op = tail_op if tail_op is not None else op
        lat_op.enqueue(max(2, args.warmup))   # untimed: this instance's key and tables back into the caches the batched instances have just swept
        lat_op.sync()
        barrier()
        t1 = time.perf_counter()
        lat_op.enqueue(args.steps)
        lat_op.sync()
        barrier()
        single = args.steps / (time.perf_counter() - t1)
    # N > 1, beside the sharded figure: what the same N GPUs do with INDEPENDENT ops (every rank runs its own two one-GPU instances of
    # `rep_batch` hmults per launch, no exchange): the node's capacity for a stream of unrelated ops, where `value` is ONE op's limbs spread
    # over the GPUs.  Information only; any failure here leaves the line without the field.
    replicas = None
    if world > 1:
        rep_batch, rep_launches, reps, ok = 10, 4, [], 1.0
        try:
            reps = [host.Op(CFG, opn, L, ELL, ALPHA, device=local_rank, overrides={"seed": host.SEED + 31 * (i + 1), "batch": rep_batch, "graph": 1, "world": 1, "rank": 0}) for i in range(2)]
            for _ in range(3):
                for r_ in reps:
                    r_.enqueue(1)
            for r_ in reps:
                r_.sync()
        except Exception as e:  # noqa: BLE001
            ok = 0.0
            print("replicas leg skipped:", repr(e), file=sys.stderr, flush=True)
        flag = torch.tensor([ok], dtype=torch.float64, device=red_dev)   # every rank takes the same branch: a rank that failed must not leave the others in a collective
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) == 1.0:
            barrier()
            t1 = time.perf_counter()
            try:
                for _ in range(rep_launches):
                    for r_ in reps:
                        r_.enqueue(1)
                for r_ in reps:
                    r_.sync()
            except Exception as e:  # noqa: BLE001
                ok = 0.0
                print("replicas leg failed:", repr(e), file=sys.stderr, flush=True)
            barrier()
            res = torch.tensor([time.perf_counter() - t1, -ok], dtype=torch.float64, device=red_dev)
            dist.all_reduce(res, op=dist.ReduceOp.MAX)
            if float(res[1].item()) == -1.0:
                replicas = {"ops_per_s": world * 2 * rep_launches * rep_batch / float(res[0].item()), "instances_per_gpu": 2, "batch": rep_batch,
                            "note": "independent one-GPU instances on every rank, no exchange (weak scaling of unrelated ops); `value` is one op sharded over the GPUs"}
        for r_ in reps:
            try:
                r_.close()
            except Exception:  # noqa: BLE001
                pass
    # per-launch device time of one op, each launch bracketed by its own event pair (collective when sharded)
    stage_rows = (tail_op if tail_op is not None else op).stage_times(5)   # sharded: of one batch
    batched_rows = ops[0].stage_times(3) if world == 1 and batch > 1 else None   # per launch of `batch` ops
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # N > 1: the NTT sweep of the extended basis with its limbs sharded like the op's (limb e -> rank e % N), every rank on its own limbs at
    # the same time (north_star: "HBM GB/s for the NTT sweep ... at 1/2/4/8 GPUs"); per-rank device times gathered on rank 0
    sharded_sweep = None
    if world > 1:
        own = [e for e in range(ELL + ALPHA) if e % world == rank]
        barrier()
        ns_r, cross_r, one_r, ns_ip_r = measure_ntt_sweep(len(own), sets=max(6, 6 * world // 2), device=local_rank, which=own)
        tt = torch.zeros(3 * world, dtype=torch.float64, device=red_dev)
        tt[3 * rank], tt[3 * rank + 1], tt[3 * rank + 2] = ns_r, len(own), cross_r
        dist.all_reduce(tt)
        sharded_sweep = {"ns": [float(tt[3 * r].item()) for r in range(world)], "limbs": [int(tt[3 * r + 1].item()) for r in range(world)],
                         "cross": int(sum(tt[3 * r + 2].item() for r in range(world))), "one_launch": one_r, "ns_inplace_rank0": ns_ip_r}
    comm_seen = None
    if world > 1:
        try:
            comm_seen = {"world": op.backend_counter("comm_world"), "ranks_seen": op.backend_counter("comm_ranks_seen"), "transport": ["none", "rccl", "external"][op.backend_counter("comm_transport")]}
        except Exception as e:  # noqa: BLE001
            comm_seen = {"error": str(e)[:200]}
    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = args.steps / dt   # whole-job rate: the N GPUs complete `steps` sharded hmults together
        sweep_limbs = ELL + ALPHA
        if sharded_sweep is None:
            ntt_ns, sweep_cross, sweep_one_launch, ntt_ns_inplace = measure_ntt_sweep(sweep_limbs)
        else:   # this rank's own launch is the `roofline` kernel (per GPU); the aggregate over the ranks rides beside it
            sweep_limbs = sharded_sweep["limbs"][0]
            ntt_ns, sweep_cross, sweep_one_launch, ntt_ns_inplace = sharded_sweep["ns"][0], sharded_sweep["cross"], sharded_sweep["one_launch"], sharded_sweep["ns_inplace_rank0"]
        # limb-polys of one-launch transforms that took the agent-scope path of the rendezvous in the op instances (timed region, warm-up, stage timings)
        op_cross = sum(o.backend_counter("ntt_cross_xcd") for o in ops + ([tail_op] if tail_op is not None else []))
        arith = op.backend_counter("arith")
        inop_ns, inop_limbs = measure_ntt_inop(batch) if world == 1 else (None, None)
        achieved = NTT_ALG_BYTES * sweep_limbs / ntt_ns  # B/ns = GB/s
        rin = roofline_inputs()
        # second ceiling (SURVEY.md §8d "measure and print both"): VALU issue.  One limb-NTT = N/2 * logN butterflies; a
        # wave executes 64 at a time; `wave_butterfly_ns` is the measured full-chip time per wave-butterfly
        wb_ns = rin.get("wave_butterfly_ns")
        valu_floor_ns = wb_ns * BFLY_PER_LIMB_NTT / 64 * sweep_limbs if wb_ns else None
        hbm_floor_ns = NTT_ALG_BYTES * sweep_limbs / HBM_PEAK_GBS
        out = {
            "metric": "hmult+key-switch ops/sec" if opn == "hmult" else "hrotate+key-switch ops/sec", "value": value, "unit": "ops/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{CFG} {opn} L={L} l={ELL} alpha={ALPHA} (N=2^16, beta=3, " + ("full hybrid key switch + rescale)" if opn == "hmult" else "automorphism + full hybrid key switch)"),
                       "parallelism": "single GPU" if world == 1 else sharded_plan_note(op, world, batch),
                       "launches_per_op": op.launch_count(), "streams": streams, "batch": batch, "transport": transport,
                       "hip_graph": bool(world == 1 and args.graph),
                       "moduli": MODULI_NOTE[arith],
                       "evk_note": "the ops of a batch share ONE evaluation key: the 157 MB key stream that the algorithmic figure charges per op is read from HBM once per batch, the other readers hit cache",
                       "streams_note": "`streams` instances in flight (own HBM pool / HIP stream each), each carrying `batch` independent hmults per launch (own inputs, one evaluation key); a step is one hmult"},
            "hip_library": loaded_hip_library(),
            # N > 1: what the communicator of the HIP library itself reports (hm_get_counter comm_*): ranks_seen = ncclCommCount over RCCL
            "comm": comm_seen,
            # one-launch transforms: limb-polys whose workgroups were spread over XCDs (the rendezvous' slow agent-scope path; 0 expected) after
            # the sweep and in the op instances, and whether the form was in force (0 = off: the guard of hm_create or a time-out)
            "ntt_cross_xcd": {"after_sweep": sweep_cross, "after_timed_region": op_cross, "ntt_fused_small": sweep_one_launch},
            "value_min_median_max": None if world > 1 else [min(region_rates), sorted(region_rates)[1], max(region_rates)],   # three K-step regions; `value` is the first
            "single_stream_ops_per_s": single,
            "independent_replicas": replicas,
            "sustained_ops_per_s": sustained,
            "launches_in_timed_region_per_instance": args.steps // (batch * streams),
            "stage_us": [[kind, name, round(ns * 1e-3, 2)] for kind, name, ns in stage_rows],
            "stage_us_per_op_batched": None if not batched_rows else [[kind, name, round(ns * 1e-3 / batch, 2)] for kind, name, ns in batched_rows],
            "exchange_us_per_op": round(sum(ns for kind, _, ns in stage_rows if kind in EXCHANGE_KINDS) * 1e-3 / (batch if world > 1 else 1), 2),
            "exchange_overlap": exchange_overlap(stage_rows, batch if world > 1 else 1, ms * 1e3, streams if world > 1 else 1,
                                                 world > 1 and any(" mark=" in ln for ln in op.plan())),   # the op's own plan says whether its exchanges are pipelined
            "hmult_hbm_gbs_algorithmic": alg_bytes / (ms * 1e-3) / 1e9,
            "hmult_frac_of_hbm_peak": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # the same with the evaluation key (2 beta E = 300 limb-polys) charged ONCE per launch of `batch` ops instead of once per
            # op: what the algorithmic figure describes when the ops of a launch share their key, as they do here
            "hmult_alg_bytes_evk_once": alg_bytes - EVK_BYTES * (1 - 1 / batch),
            "hmult_frac_evk_once": (alg_bytes - EVK_BYTES * (1 - 1 / batch)) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # what the chip moved: FETCH_SIZE x 2 + WRITE_SIZE of one op at the timed region's launch shape (profiler run, not this run)
            "measured_hbm": None if opn != "hmult" or not rin.get("whole_op_bytes") or rin.get("whole_op_batch") != batch or rin.get("whole_op_instances") != streams else {
                "bytes_per_op": rin["whole_op_bytes"], "tb_per_s": rin["whole_op_bytes"] / (ms * 1e-3) / 1e12,
                "frac_of_peak": rin["whole_op_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "batch": rin["whole_op_batch"], "instances": rin["whole_op_instances"], "source": rin.get("whole_op_source"),
                "note": "bytes from the committed rocprofv3 counter passes of the same launch shape, time from this run"},
            # the op's dominant launch (ModUp conversion + transforms + key MAC in one C-ABI call): algorithmic bytes with the key charged once per
            # launch of `batch` ops, HBM bytes of its kernels from the committed counter passes, time from this run's per-launch events
            "roofline_op": roofline_op(batched_rows, batch, rin) if world == 1 and opn == "hmult" else None,
            "roofline": {"bound": "valu" if valu_floor_ns and valu_floor_ns > hbm_floor_ns else "hbm",   # the ceiling with the larger floor for this launch, measured
                         "contract_bound": "hbm",   # ... `achieved` / `peak` / `frac` / `traffic` are the HBM figures the task's contract asks for, whatever binds
                         "binding_ceiling": "valu" if valu_floor_ns and valu_floor_ns > hbm_floor_ns else "hbm",
                         "kernel": f"forward NTT sweep, {sweep_limbs} limbs = ONE launch, k_ntt_fused8<8, false, 0, 1, true> (COL pass on non-temporal input loads, per-limb rendezvous on XCD-local atomics, ROW pass)" + ("" if world == 1 else f": this rank's limbs of the 50-limb extended basis (limb e -> rank e % {world}), all ranks sweeping at once"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": rin.get("ntt_sweep50_traffic_bytes") if sweep_limbs == 50 and world == 1 else None, "us_per_launch": ntt_ns * 1e-3,
                         "algorithmic_bytes_per_launch": NTT_ALG_BYTES * sweep_limbs,
                         "working_set": "6 rotating buffer pairs, 315 MB: past the 256 MiB Infinity Cache",
                         "hbm": {"floor_us": hbm_floor_ns * 1e-3, "frac": hbm_floor_ns / ntt_ns},
                         "valu": None if valu_floor_ns is None else {
                             "floor_us": valu_floor_ns * 1e-3, "frac": valu_floor_ns / ntt_ns, "unit": "wave-butterflies/ns",
                             "achieved": BFLY_PER_LIMB_NTT / 64 * sweep_limbs / ntt_ns, "peak": 1.0 / wb_ns,
                             "source": rin.get("wave_butterfly_source")},
                         # N > 1: the whole 50-limb sweep over the N GPUs = every rank's launch at once; the slowest rank's time counts
                         "sharded": None if sharded_sweep is None else {
                             "limbs_per_rank": sharded_sweep["limbs"], "us_per_rank": [round(x * 1e-3, 2) for x in sharded_sweep["ns"]],
                             "aggregate_gbs": NTT_ALG_BYTES * (ELL + ALPHA) / max(sharded_sweep["ns"]),
                             "frac_per_gpu": [NTT_ALG_BYTES * n / t / HBM_PEAK_GBS for n, t in zip(sharded_sweep["limbs"], sharded_sweep["ns"])],
                             "frac_of_aggregate_peak": NTT_ALG_BYTES * (ELL + ALPHA) / max(sharded_sweep["ns"]) / (HBM_PEAK_GBS * world)},
                         # the OP-level figures to lead with (the sweep above is the contract's `kernel`): the timed region's bytes over its time,
                         # with the evaluation key charged once per launch (the ops of a batch share it), and by the profiler's counters
                         "op_frac_evk_once": (alg_bytes - EVK_BYTES * (1 - 1 / batch)) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "op_measured_frac": None if opn != "hmult" or not rin.get("whole_op_bytes") or rin.get("whole_op_batch") != batch or rin.get("whole_op_instances") != streams
                         else rin["whole_op_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "op_note": "op_frac_evk_once = (SURVEY.md 8(d) bytes of the op, key once per launch) / step time / 8 TB/s; op_measured_frac = (2 x FETCH_SIZE + WRITE_SIZE per op "
                                    "from the committed counter passes of this launch shape) / step time / 8 TB/s; `achieved` / `frac` describe the kernel named in `kernel`",
                         "in_place": {"us_per_launch": ntt_ns_inplace * 1e-3, "achieved": NTT_ALG_BYTES * sweep_limbs / ntt_ns_inplace,
                                      "frac": NTT_ALG_BYTES * sweep_limbs / ntt_ns_inplace / HBM_PEAK_GBS,
                                      "note": "the same 50-limb sweep with in == out (k_ntt_fused8<false, 0, 1, false>): the hand-off stays on L2 lines the input loads brought in; "
                                              "the op's own transforms are out of place, which is what `achieved` / `frac` above describe"},
                         "in_op": None if not inop_ns else {
                             "kernel": "the ModUp forward transforms of one launch of the timed region (115 limb-polys per op x batch) as ONE hm_ntt call = k_ntt_col + k_ntt_row, timed alone with HIP events on the backend's stream",
                             "limbs_per_launch_group": inop_limbs, "us": inop_ns * 1e-3, "us_per_limb": inop_ns * 1e-3 / inop_limbs,
                             "achieved": NTT_ALG_BYTES * inop_limbs / inop_ns, "frac": NTT_ALG_BYTES * inop_limbs / inop_ns / HBM_PEAK_GBS,
                             "valu_frac": None if not wb_ns else wb_ns * BFLY_PER_LIMB_NTT / 64 * inop_limbs / inop_ns},
                         "real_traffic": {
                             "floor_bytes_per_limb_ntt": 4 * LP,
                             "note": "a limb-poly (512 KiB) does not fit one CU's LDS (160 KiB), so a transform is two passes through global memory with a hand-off between "
                                     "them (4 limb-polys per limb-NTT + row twiddles = the floor above).  Since round 4 launches of up to 96 limb-polys run both passes in ONE kernel: the "
                                     "workgroups of a limb-poly share an XCD, meet at a counter in that XCD's L2 (returning atomics without the agent-scope bit) and read the hand-off "
                                     "from the L2 that holds it (write-back: profiles/r04_l2_handoff.txt); larger launches stay two kernels, where the rendezvous costs more than it saves"},
                         "note": "`achieved`/`peak`/`frac`/`traffic` are the HBM figures of the task's contract (`contract_bound`); `bound` names the ceiling "
                                 "with the larger floor for this launch (the 64-bit modular butterflies are integer VALU work)"},
        }
        if world == 1 and opn == "hmult":   # configs[3] at the same launch shape, so that the driver's run times it too
            out["hrotate"] = measure_second_op("hrotate", streams, batch, args.steps, local_rank)
            # ... and the headline op on SURVEY.md 8(d)'s chain as written (largest primes = 1 mod 2N below 2^60): the generic arithmetic
            # back-end behind the same ABI (`value` is the default chain of primes h 2^32 + 1: config.moduli)
            if arith == 0:
                g = measure_second_op("hmult", streams, batch, args.steps, local_rank, {"chain_bits": 60})
                out["generic_chain"] = g
                out["generic_chain_ops_per_s"] = g["ops_per_s"]
                out["generic_chain_frac_evk_once"] = g["frac_evk_once"]
        if world == 1 and opn == "hmult":   # the reference's other three ops (one element-wise stage each): rate and fraction of the HBM peak, same launch shape
            # (50 launches per instance: a launch of ten element-wise ops is ~15 us)
            out["elementwise"] = {o_: measure_second_op(o_, streams, batch, 50 * streams * batch, local_rank) for o_ in ("hadd", "pmult", "padd")}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(opn)
        try:  # the reference's own answer for the same command line, from the build's cycle model (backend = sim): host only
            t0 = time.time()
            sim_op = host.Op(CFG, opn, L, ELL, ALPHA, backend=host.BACKEND_SIM)
            sim = sim_op.sim_run()
            sim_op.close()
            out["reference_model"] = {"cycles": sim["cycles"], "instructions": sim["retired"], "drained": sim["drained"], "host_seconds": round(time.time() - t0, 2),
                                      "note": "simulated accelerator of the .cfg (4 clusters): what the reference simulator prints for this op (the reference's own run of this configuration with MALLOC_PERTURB_ set — its clean run: it reads uninitialised scoreboard operands, include/recodeboard.h:33-46 — took 3 h 18 min and gave the same cycles and counters: tests/golden/structural.json slow_points" + ("" if opn == "hmult" else "; for hrotate the stock binary can differ in a few counters for that reason") + "); not a GPU measurement"}
        except Exception as e:  # never let the side figure take the bench line down
            out["reference_model"] = {"error": str(e)[:200]}
    for o in ops:
        o.close()
    if tail_op is not None:
        tail_op.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
