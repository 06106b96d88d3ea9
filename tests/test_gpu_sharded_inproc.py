"""BASELINE configs[4]: hmult 45/35/15 with the limb-polys sharded over EIGHT ranks, executed for real on the one GPU of
the test box: 8 ranks = 8 threads of this process (own HIP context, stream and HBM pool each), exchanges through
homulator_amd.dist.InProcessGroup (device-to-device copies between the ranks' staging buffers).  A GPU box admits at
most 6 processes on its card, so 8 gloo processes are not an option; the ownership rule (upstream's `limb % cluster`,
include/Driver.h:158,178), the all-to-all pairs around both base conversions, the replicate of the rescale residue
and the per-pair byte counts are exactly those of an 8-GPU run — only the wire differs (RCCL over xGMI there).
Ranks that own NO limb of a list (the 2-limb rescale INTT at 8 ranks) send and receive nothing; both sides of every
pair must agree on its size (checked by the transport)."""
import numpy as np
import pytest

from oracle.homoracle import Oracle

pytestmark = pytest.mark.gpu
SEED = 0x484F4D55


def assemble(ops, name, n_limbs, N, copy=0):
    full = np.zeros((n_limbs, N), dtype=np.uint64)
    seen = np.zeros(n_limbs, dtype=int)
    for op in ops:
        mine = op.read(name, copy=copy)
        for l in op.owned(n_limbs):
            full[l] = mine[l]
            seen[l] += 1
    assert (seen == 1).all(), "every limb has exactly one owner"
    return full


@pytest.mark.parametrize("world,cfg,opname,L,ell,alpha,logN,batch,fused,plan", [
    (8, "config_4.cfg", "hmult", 45, 35, 15, 16, 1, 1, 0),       # BASELINE configs[4], round-4 plan (the choice above 4 ranks): the fused kernels on column slices
    (8, "config_4.cfg", "hmult", 45, 35, 15, 16, 1, 0, 0),       # ... and the round-3 plan (shard_fused = 0)
    (8, "config_4.cfg", "hrotate", 45, 35, 15, 16, 1, 1, 0),
    (8, "config_4_N15.cfg", "hmult", 16, 10, 4, 15, 3, 1, 0),    # batched: the ops of a batch share every exchange; N = 2^15: one 32-column tile per rank
    (4, "config_4_N15.cfg", "hmult", 6, 5, 2, 15, 1, 1, 1),      # fewer limbs than ranks in some lists
    (16, "config_4.cfg", "hmult", 45, 35, 15, 16, 1, 1, 0),      # 16 ranks: one 16-column first-pass tile per rank
    (8, "config_4.cfg", "hmult", 45, 35, 15, 16, 1, 1, 2),       # round-5 gather plan forced at 8 ranks: 3 collectives, one-GPU kernels per rank
    (8, "config_4.cfg", "hrotate", 45, 35, 15, 16, 1, 1, 2),
    (4, "config_4.cfg", "hmult", 45, 35, 15, 16, 2, 1, 0),       # ... and where it is the automatic choice, batched
    (4, "config_4_N15.cfg", "hmult", 6, 5, 2, 15, 1, 1, 2),      # ranks that own no limb of a gathered list
    (8, "config_4.cfg", "hmult", 45, 35, 15, 16, 2, 1, 0),       # two ops' residues = 2 MiB from one owner: replicated as scatter + exchange of chunks
    (4, "config_4_N15.cfg", "hmult", 16, 10, 4, 15, 3, 1, -2),   # (plan < 0: that plan with HOMULATOR_REPLICATE_SPLIT=1) ... in the gather plan, uneven chunks
    (16, "config_4_N15.cfg", "hmult", 6, 5, 2, 15, 1, 1, -2),    # ... 15 peers, 128 blocks: a short last chunk
])
def test_sharded_in_process(world, cfg, opname, L, ell, alpha, logN, batch, fused, plan, monkeypatch):
    from homulator_amd import host
    from homulator_amd.dist import run_in_process

    split = plan < 0 or (batch > 1 and logN == 16)
    if plan < 0:
        monkeypatch.setenv("HOMULATOR_REPLICATE_SPLIT", "1")
        plan = -plan

    def make(r):
        ov = {**({"batch": batch} if batch > 1 else {}), **({} if fused else {"shard_fused": 0}), **({"shard_plan": plan} if plan else {})}
        return host.Op(cfg, opname, L, ell, alpha, rank=r, world=world, overrides=ov or None)

    def body(r, op):
        op.execute(1)
        op.execute(1)   # the plan must be re-runnable
        return True

    ops, res, grp = run_in_process(world, make, body)
    assert all(res) and not grp.failed
    kinds = [ln.split()[0] for ln in ops[0].plan()]
    gather = plan == 2 or (plan == 0 and world <= 4)
    if gather:
        assert "BCONV_COL" not in kinds and "NTT_IP" in kinds and not any(k.startswith("EXCH") for k in kinds)
        assert grp.calls[0] // 2 == (3 if opname == "hmult" else 2) + (1 if split and world >= 4 else 0), grp.calls
    else:
        assert ("BCONV_COL" in kinds and "NTT_IP" in kinds and "NTT" not in kinds) if fused else ("BCONV_COL" not in kinds and "NTT" in kinds)
    assert len(set(grp.calls)) == 1 and grp.calls[0] > 0, grp.calls          # every rank entered every exchange
    if not gather and fused:
        beta = -(-ell // alpha)
        assert grp.calls[0] // 2 == 2 * beta + 2 + (opname == "hmult") + (1 if split else 0), grp.calls
    N = 1 << logN
    n_out = ell - 1 if opname == "hmult" else ell
    o = Oracle(logN, L, alpha)
    o.set_threads(8)
    evk = o.synth_evk(ell, SEED + 10000)
    for c in range(batch):
        S = SEED + c * 100000
        ct1, ct2 = o.synth_ct(ell, S), o.synth_ct(ell, S + 2000)
        exp = o.hmult(ell, ct1, ct2, evk) if opname == "hmult" else o.hrotate(ell, ct1, 5, evk)
        assert np.array_equal(assemble(ops, "out.c0", n_out, N, c), exp[0])
        assert np.array_equal(assemble(ops, "out.c1", n_out, N, c), exp[1])
    if world == 8 and opname == "hmult" and batch == 1 and not gather:   # (both plans: the same exchanges, in the transposed domain or not)
        # every pair of one exchange carries slices of N/8 coefficients; rank 0's ingress per hmult is
        # 7/8 of (its share of) the exchanged limb-polys: nonzero, identical on the two runs, below the 13.7 MiB bound + replicate
        runs = 2       # execute(1) twice
        assert grp.calls[0] % runs == 0 and grp.calls[0] // runs == 9, grp.calls  # per-digit pipelined: 2 beta + 2 = 8 all-to-alls + 1 replicate per hmult
        per_op = grp.bytes_recv[0] / runs
        assert 0 < per_op < 20 * 2 ** 20, per_op
    if cfg == "config_4.cfg" and opname == "hmult" and batch == 1 and fused:
        # the per-collective budget of DESIGN.md section 7 (tools/shard_budget.py derives it from the plans) against the bytes the transport moved
        import os
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
        from shard_budget import received_per_rank
        assert [b // 2 for b in grp.bytes_recv] == received_per_rank(world, 2 if gather else 1), (grp.bytes_recv, world, gather)
    for op in ops:
        op.close()


def test_bench_flow_rehearsal_four_ranks():
    """the driver's multi-GPU bench flow (torch.distributed.run, batch selection, transport agreement, MAX-reduce, one JSON
    line on rank 0) executed once with 4 ranks on the one GPU over gloo (HOMULATOR_DIST_BACKEND=gloo; 4 + this process
    stay within the box's 6-process limit — the reason it is not 8)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HOMULATOR_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "8", "--warmup", "2"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 4 and d["steps"] == 8 and d["value"] > 0
    assert d["config"]["transport"] == "gloo-rehearsal" and d["exchange_us_per_op"] > 0
    rep = d["independent_replicas"]    # beside the sharded figure: the same ranks running unrelated one-GPU ops (here all four on the one GPU)
    assert rep and rep["ops_per_s"] > 0 and rep["instances_per_gpu"] == 2
    ov = d["exchange_overlap"]   # over gloo the exchanges are host-synchronous: whatever the estimate calls hidden is noise, the sum is the exchange time
    assert "gather plan" in d["config"]["parallelism"] and ov["collectives_per_launch"] == 3     # 4 ranks: the automatic choice
    assert ov["instances_in_flight"] == 1 and not ov["pipelined_per_digit"] and ov["hidden_us_per_op"] >= 0.0
    assert abs(ov["hidden_us_per_op"] + ov["exposed_us_per_op"] - d["exchange_us_per_op"]) < 0.05


def test_bench_flow_rehearsal_two_sharded_instances():
    """--sharded-streams 2 (opt-in): two sharded instances per rank, a transport each, enqueued alternately in the same order on every
    rank; 2 ranks over gloo.  The bench line carries the hidden / exposed estimate."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HOMULATOR_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29537", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--sharded-streams", "2"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["streams"] == 2
    ov = d["exchange_overlap"]
    assert ov["instances_in_flight"] == 2 and ov["hidden_us_per_op"] >= 0 and abs(ov["hidden_us_per_op"] + ov["exposed_us_per_op"] - d["exchange_us_per_op"]) < 0.05


def test_ranks_that_disagree_on_the_replicate_threshold_fail_at_init():
    """hm_replicate_limbs takes one exchange or scatter + exchange of chunks from `replicate_split_bytes`, which every process
    reads from ITS environment / option: ranks that disagree would enter collectives of different shape and hang in RCCL.  Round 6: the first
    thing a new communicator carries is every rank's threshold to every peer; a mismatch fails hm_comm_init_* with HM_ERR_COMM on EVERY rank
    (nobody is left inside a collective), and the option is fixed once the communicator exists."""
    import ctypes as C
    import threading
    from homulator_amd import hip
    from homulator_amd.dist import InProcessGroup
    world = 4
    grp = InProcessGroup(world)
    ctxs = [hip.Context(13, 3, 2) for _ in range(world)]
    ctxs[2].set_option("replicate_split_bytes", 12345)   # one rank with another threshold
    errs = [None] * world

    def work(r):
        fn = grp.transport(r)
        st = ctxs[r].L.hm_comm_init_external(ctxs[r].h, r, world, C.cast(fn, C.c_void_p), None)
        errs[r] = (st, ctxs[r].L.hm_last_error(ctxs[r].h).decode())
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(st == 4 for st, _ in errs), errs                      # HM_ERR_COMM everywhere
    assert all("replicate_split_bytes" in msg for _, msg in errs), errs
    assert all(c.counter("comm_world") == 1 and c.counter("comm_transport") == 0 for c in ctxs)   # no communicator was kept
    # agreeing ranks: the communicator is made, reports itself, and the option is fixed from then on
    ctxs[2].set_option("replicate_split_bytes", 2 << 20)
    grp2 = InProcessGroup(world)

    def work2(r):
        fn = grp2.transport(r)
        errs[r] = ctxs[r].L.hm_comm_init_external(ctxs[r].h, r, world, C.cast(fn, C.c_void_p), None)
        ctxs[r]._keep = fn
    th = [threading.Thread(target=work2, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert errs == [0] * world
    assert all(c.counter("comm_world") == world and c.counter("comm_ranks_seen") == world and c.counter("comm_transport") == 2 for c in ctxs)
    with pytest.raises(hip.HmError, match="fixed once the communicator exists"):
        ctxs[0].set_option("replicate_split_bytes", 1)
    ctxs[0].set_option("replicate_split_bytes", 2 << 20)   # the value in force: accepted
    for c in ctxs:
        c.close()
