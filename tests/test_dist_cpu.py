"""N > 1 path on the CPU (no GPU): (1) the sharded launch plans that the C++ host layer builds for every rank are
collectively consistent (same exchange sequence, same limb lists, work partitioned exactly); (2) the slice-row
layout helper of the HIP library; (3) GlooTransport under a real world_size-2 gloo group."""
import ctypes as C
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(world, args, timeout=600):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")] + args, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    return [p.returncode for p in procs], outs


def test_gloo_transport_world2():
    rcs, outs = launch(2, ["transport"])
    assert rcs == [0, 0], outs


@pytest.mark.parametrize("opname", ["hmult", "hrotate"])
@pytest.mark.parametrize("pipeline", [1, 0])
@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("world,batch", [(2, 1), (4, 1), (8, 1), (2, 4), (8, 16)])
def test_sharded_plans_are_collectively_consistent(opname, world, batch, pipeline, fused):
    """fused = 1 (round 4, default): the ModUp runs on the fused kernels — per digit limbs -> COLUMN slices, conversion + first pass on the
    rank's columns (BCONV_COL), column slices -> limbs of the hand-off, then ONE transform x key launch (NTT_IP) per rank: beta + 1 ModUp
    launches per rank instead of 2 beta + 1; fused = 0 (shard_fused = 0): the round-3 plan (conversion on contiguous slices, per-digit
    transforms, inner product).  pipeline = 1 (default when sharded): per-digit exchanges, 2 beta + 2 all-to-alls per key switch, every
    launch behind an exchange waits for that exchange's mark only; pipeline = 0 (fused = 0 only changes the grouping): one pair per stage"""
    from homulator_amd import host
    L, ell, alpha = 45, 35, 15
    beta = -(-ell // alpha)
    ov = {"shard_plan": 1, "pipeline_digits": pipeline, **({"batch": batch} if batch > 1 else {})}   # (auto picks the gather plan up to 4 ranks)
    if not fused:
        ov.update({"shard_fused": 0, "fuse_hpip": 0 if pipeline else 1})
    single = host.Op("config_4.cfg", opname, L, ell, alpha, backend=host.BACKEND_COUNT, overrides=ov)
    total_ref = sum(int(re.search(r"ref=(\d+)", ln).group(1)) for ln in single.plan())
    plans = []
    for r in range(world):
        o = host.Op("config_4.cfg", opname, L, ell, alpha, backend=host.BACKEND_COUNT, rank=r, world=world, overrides=ov)
        plans.append(o.plan())
    XK = ("EXCH_IN", "EXCH_OUT", "REPLICATE", "EXCH_IN_COL", "EXCH_OUT_COL")
    coll = [[ln for ln in pl if ln.split()[0] in XK] for pl in plans]
    # every rank enters the same collectives, in the same order, with the same limb:owner lists
    assert all(c == coll[0] for c in coll)
    kind = lambda ln: ln.split()[0]
    cnt = lambda pl, k: sum(1 for ln in pl if kind(ln) == k)
    if fused:
        # ModUp: one column exchange pair and one BCONV_COL per digit on EVERY rank, one NTT_IP on every rank that owns an extended limb;
        # ModDown: the contiguous-slice conversion as before
        for pl in plans:
            assert cnt(pl, "EXCH_IN_COL") == cnt(pl, "BCONV_COL") == cnt(pl, "EXCH_OUT_COL") == beta
            assert cnt(pl, "EXCH_IN") == cnt(pl, "BCONV") == cnt(pl, "EXCH_OUT") == 1
            assert cnt(pl, "NTT_IP") == 1 and cnt(pl, "NTT") == 0 and cnt(pl, "IP") == 0
        if pipeline:
            marks = [int(re.search(r"mark=(\d+)", ln).group(1)) for ln in plans[0] if " mark=" in ln]
            assert sorted(marks) == list(range(len(marks)))
            xin = {re.match(r"EXCH_IN_COL (\S+):", ln).group(1): int(re.search(r"mark=(\d+)", ln).group(1)) for ln in plans[0] if kind(ln) == "EXCH_IN_COL"}
            xout = [int(re.search(r"mark=(\d+)", ln).group(1)) for ln in plans[0] if kind(ln) == "EXCH_OUT_COL"]
            for pl in plans:
                for ln in pl:
                    if kind(ln) == "BCONV_COL":   # a conversion waits for its own digit's exchange-in only
                        assert re.search(r"wait=([\d,]+)", ln).group(1).strip(",") == str(xin[ln.split()[1]])
                    if kind(ln) == "NTT_IP":      # the transform x key launch needs every digit's hand-off
                        assert sorted(int(x) for x in re.search(r"wait=([\d,]+)", ln).group(1).strip(",").split(",")) == sorted(xout)
            first_bc = next(i for i, ln in enumerate(plans[0]) if kind(ln) == "BCONV_COL")
            assert sum(1 for ln in plans[0][:first_bc] if kind(ln) == "EXCH_IN_COL") == beta   # all digits' exchange-in are in flight before the first conversion
    else:
        n_bconv = sum(1 for ln in plans[0] if ln.startswith("BCONV"))
        assert sum(1 for ln in coll[0] if ln.startswith("EXCH_IN")) == n_bconv == sum(1 for ln in coll[0] if ln.startswith("EXCH_OUT"))
        # one all-to-all pair for ModUp (all digits) and one for ModDown (both keys), or 2 beta + 2 pipelined per digit
        assert n_bconv == (beta + 1 if pipeline else 2)
        if pipeline:   # marks: every exchange sets one, in issue order; a conversion waits for its own exchange-in, a transform for its digit's exchange-out
            marks = [int(re.search(r"mark=(\d+)", ln).group(1)) for ln in plans[0] if " mark=" in ln]
            assert sorted(marks) == list(range(len(marks)))
            xin = {re.match(r"EXCH_IN (\S+):", ln).group(1): int(re.search(r"mark=(\d+)", ln).group(1)) for ln in plans[0] if ln.startswith("EXCH_IN")}
            xout = {re.match(r"EXCH_OUT (\S+):", ln).group(1): int(re.search(r"mark=(\d+)", ln).group(1)) for ln in plans[0] if ln.startswith("EXCH_OUT")}
            for ln in plans[0]:
                if ln.startswith("BCONV"):
                    assert re.search(r"wait=([\d,]+)", ln).group(1).strip(",") == str(xin[ln.split()[1]])
            for j in range(beta):
                for pl in plans:
                    for ln in pl:
                        if ln.startswith(f"NTT ModUp_NTT_({j})"):
                            assert re.search(r"wait=([\d,]+)", ln).group(1).strip(",") == str(xout[f"ModUp_BCONV_({j})"])
            first_bconv = next(i for i, ln in enumerate(plans[0]) if ln.startswith("BCONV"))
            assert sum(1 for ln in plans[0][:first_bconv] if ln.startswith("EXCH_IN")) == beta   # all digits' exchange-in are in flight before the first conversion
    if opname == "hmult":
        assert sum(1 for ln in coll[0] if ln.startswith("REPLICATE")) == 1   # rescale's r
    # owners follow limb % world on the exchanged limbs, and the element-wise work is partitioned exactly
    EW = ("NTT", "INTT", "EWE", "AUTO", "NTT_SUBSCALE", "TENSOR", "IP", "NTT_IP")
    per_rank = []
    for pl in plans:
        per_rank.append(sum(int(re.search(r" n=(\d+)", ln).group(1)) for ln in pl if kind(ln) in EW))
    # (against the one-GPU plan with the conversions as launches of their own, as in every sharded plan: with fuse_bconv the residue's
    # element-wise step is the ModDown conversion's epilogue and is not a limb-poly of an EWE launch)
    single_conv = host.Op("config_4.cfg", opname, L, ell, alpha, backend=host.BACKEND_COUNT, overrides={**ov, "fuse_bconv": 0})
    n_single = sum(int(re.search(r" n=(\d+)", ln).group(1)) for ln in single_conv.plan() if kind(ln) in EW)
    if opname == "hrotate":   # pass 12: a sharded plan reads through the automorphism wherever one GPU does (the plain inner product of the unfused plan cannot)
        keeps = ov.get("fuse_hpip") == 0
        assert all(sum(1 for ln in pl if kind(ln) == "AUTO" and "AUTO_Key(0)" not in ln) == (1 if keeps else 0) and any(" auto_addend=" in ln for ln in pl)
                   for pl in plans)
        if keeps and not any(kind(ln) == "AUTO" for ln in single_conv.plan()):
            n_single += ell * batch   # the l limb-polys of AUTO_Key(1) per op
    assert sum(per_rank) == n_single
    assert max(per_rank) - min(per_rank) <= 12 * batch   # balanced up to the remainder limbs of each stage
    # the instruction accounting of the sharded plans adds up to the one-GPU total
    assert sum(int(re.search(r"ref=(\d+)", ln).group(1)) for pl in plans for ln in pl if kind(ln) not in ("BCONV",)) + \
        sum(int(re.search(r"ref=(\d+)", ln).group(1)) for ln in plans[0] if kind(ln) == "BCONV") == total_ref


@pytest.mark.parametrize("opname", ["hmult", "hrotate"])
@pytest.mark.parametrize("world,batch,plan", [(2, 1, 0), (4, 1, 0), (8, 1, 2), (2, 4, 2), (4, 16, 0)])
def test_gather_plan_is_collectively_consistent(opname, world, batch, plan):
    """shard_plan = 2 (and the automatic choice up to 4 ranks): the conversions' INPUTS are replicated (all-gather of the ModUp INTT's
    limbs, of the inner product's P-limbs after their inverse transform, and of the rescale residue), every rank then runs the
    one-GPU kernels on the limbs it owns: one NTT_IP launch (conversion + transform + key product of its extended limbs), one ModDown
    conversion of its output limbs.  3 collectives per hmult (2 per hrotate) instead of 2 beta + 3, no column slices."""
    from homulator_amd import host
    L, ell, alpha = 45, 35, 15
    ov = {**({"shard_plan": plan} if plan else {}), **({"batch": batch} if batch > 1 else {})}
    single = host.Op("config_4.cfg", opname, L, ell, alpha, backend=host.BACKEND_COUNT, overrides=ov)
    plans = [host.Op("config_4.cfg", opname, L, ell, alpha, backend=host.BACKEND_COUNT, rank=r, world=world, overrides=ov).plan() for r in range(world)]
    kind = lambda ln: ln.split()[0]
    num = lambda ln, key: int(re.search(rf" {key}=(\d+)", ln).group(1))
    coll = [[ln for ln in pl if kind(ln) in ("EXCH_IN", "EXCH_OUT", "REPLICATE", "EXCH_IN_COL", "EXCH_OUT_COL")] for pl in plans]
    assert all(c == coll[0] for c in coll)
    assert [kind(ln) for ln in coll[0]] == ["REPLICATE"] * (3 if opname == "hmult" else 2)
    # what is gathered: every limb of the ModUp input (ell per op), every P-limb of both keys (2 alpha per op), the two residues
    sizes = [len(re.search(r"limbs=(\S+)", ln).group(1).strip(",").split(",")) for ln in coll[0]]
    assert sizes == [ell * batch, 2 * alpha * batch] + ([2 * batch] if opname == "hmult" else [])
    for i, ln in enumerate(coll[0]):   # the first two lists span every rank (limb e -> e % world); the residue is limb ell-1's: one owner
        owners = {int(x.split(":")[1]) for x in re.search(r"limbs=(\S+)", ln).group(1).strip(",").split(",")}
        assert owners == (set(range(world)) if i < 2 else {(ell - 1) % world}), (i, owners)
    for pl in plans:
        ks = [kind(ln) for ln in pl]
        assert ks.count("NTT_IP") == 1 and ks.count("BCONV") == 1 and "BCONV_COL" not in ks and "NTT" not in ks and "IP" not in ks
        # the order on every rank: gather -> transform x key -> inverse -> gather -> conversion (-> gather) -> merged ModDown + rescale transform
        assert ks.index("REPLICATE") < ks.index("NTT_IP") < ks.index("BCONV") < ks.index("NTT_SUBSCALE")
    # the work of every launch kind is partitioned exactly (conversions included: each rank converts the output limbs it owns) ...
    # (hrotate: the gather plan's ranks run the one-GPU kernels on the limb-polys they own, so they read through the automorphism exactly as one
    # GPU does — pass 12: no AUTO launch on any rank)
    if opname == "hrotate":
        assert not any(kind(ln) == "AUTO" for pl in plans for ln in pl) and all(any(" auto_x=" in ln for ln in pl) for pl in plans)
    for k in ("TENSOR", "AUTO", "INTT", "NTT_IP", "BCONV", "NTT_SUBSCALE"):
        assert sum(num(ln, "n") for pl in plans for ln in pl if kind(ln) == k) == sum(num(ln, "n") for ln in single.plan() if kind(ln) == k), k
    per_rank = [sum(num(ln, "n") for ln in pl if kind(ln) != "REPLICATE") for pl in plans]
    assert max(per_rank) - min(per_rank) <= 12 * batch
    # ... and so is the instruction accounting
    assert sum(num(ln, "ref") for pl in plans for ln in pl) == sum(num(ln, "ref") for ln in single.plan())


def test_plan_choice_by_world_size():
    """shard_plan = 0: gather up to 4 ranks (its 3 collectives move fewer bytes per rank there), all-to-all on column slices above"""
    from homulator_amd import host
    for world, gather in ((2, True), (4, True), (8, False), (16, False)):
        pl = host.Op("config_4.cfg", "hmult", 45, 35, 15, backend=host.BACKEND_COUNT, rank=0, world=world).plan()
        ks = [ln.split()[0] for ln in pl]
        assert ("BCONV_COL" not in ks and ks.count("REPLICATE") == 3) if gather else ("BCONV_COL" in ks and ks.count("REPLICATE") == 1), (world, ks)


def test_slice_rows_layout():
    from homulator_amd import hip
    lib = hip.load()
    lib.hm_slice_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    owners = np.array([0, 1, 2, 3, 0, 1, 2, 3, 0, 1], dtype=np.uint32)
    rows = np.empty_like(owners)
    assert lib.hm_slice_rows(owners.ctypes.data, len(owners), 4, rows.ctypes.data) == 0
    assert rows.tolist() == [0, 3, 6, 8, 1, 4, 7, 9, 2, 5]     # grouped by owner, list order inside an owner
    assert sorted(rows.tolist()) == list(range(10))
    bad = np.array([0, 5], dtype=np.uint32)
    assert lib.hm_slice_rows(bad.ctypes.data, 2, 4, rows.ctypes.data) != 0
