#!/usr/bin/env python3
"""Per-collective byte and latency budget of the two sharded plans, derived from the launch plans that the host
layer itself builds for every rank (count backend: no GPU needed).  For every collective of one hmult: what a rank RECEIVES (max over
ranks), per peer and in total; then a time model on MI355X's xGMI mesh and the predicted whole-job rate against the one-GPU rate.

Model (stated, not measured: no multi-GPU node is available to this build; the driver's SCALE run is the measurement):
  * every pair of GPUs has its own xGMI link, 76.8 GB/s per direction (MI355X_MICROARCH.md: 7 links x 153.6 GB/s bidirectional);
    LINK_EFF of it is assumed reachable by a grouped ncclSend / ncclRecv; a collective's wire time = its largest per-peer message / that;
  * T_LAT per collective (group launch + completion, not overlapped);
  * compute: the one-GPU stage times of `bench.py` (`stage_us_per_op_batched`, r05) divided by the rank count for the stages that are
    partitioned by limb or by column, plus LAUNCH_FLOOR per launch and per batch;
  * a replicate whose limbs all have ONE owner (the rescale residues) and that is >= 2 MiB on >= 4 ranks runs as scatter + exchange of chunks
    (`replicate_split_bytes`): two collectives, each link carries 2 / (G - 1) of the list; printed for the batch it applies to;
  * gather plan: nothing overlaps (bulk-synchronous); all-to-all plan: the per-digit pipeline hides PIPE_HIDE of the ModUp exchanges.
Usage: python tools/shard_budget.py > profiles/r05_shard_budget.txt"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import host  # noqa: E402

L, ELL, ALPHA, LOGN = 45, 35, 15, 16
N = 1 << LOGN
LIMB = N * 8
LINK_GBS, LINK_EFF, T_LAT_US, LAUNCH_FLOOR_US, PIPE_HIDE = 76.8, 0.8, 20.0, 6.0, 0.5
ONE_GPU_STAGE_US = {"TENSOR": 23.0, "INTT": 18.2 + 8.2, "NTT_IP": 91.0, "BCONV": 17.7, "NTT_SUBSCALE": 50.5}   # r05, batch 10, per op
ONE_GPU_OPS = 5330.0    # bench.py value at the closing r05 box (two instances x batch 10)
XK = ("EXCH_IN", "EXCH_OUT", "REPLICATE", "EXCH_IN_COL", "EXCH_OUT_COL")


def plans(world, plan, batch=1):
    ov = {"shard_plan": plan, **({"batch": batch} if batch > 1 else {})}
    return [host.Op("config_4.cfg", "hmult", L, ELL, ALPHA, backend=host.BACKEND_COUNT, rank=r, world=world, overrides=ov).plan() for r in range(world)]


def received(kind, owners, world, me):
    """bytes rank `me` receives in one collective over the limb list with these owners: (total, largest per peer)"""
    cnt = [owners.count(p) for p in range(world)]
    if kind == "REPLICATE":            # every owner sends its limbs whole to every peer
        per_peer = [cnt[p] * LIMB if p != me else 0 for p in range(world)]
    elif kind.startswith("EXCH_IN"):   # limbs -> slices: from every peer its limbs' 1 / world of the coefficients (or columns)
        per_peer = [cnt[p] * LIMB // world if p != me else 0 for p in range(world)]
    else:                              # slices -> limbs: for every limb I own, every peer's slice
        per_peer = [cnt[me] * LIMB // world if p != me else 0 for p in range(world)]
    return sum(per_peer), max(per_peer)


def collectives(pls):
    """(kind, name, owners of the limb list) of every collective of rank 0's plan (the lists are identical on every rank: tests/test_dist_cpu.py)"""
    out = []
    for ln in [l for l in pls[0] if l.split()[0] in XK]:
        owners = [int(x.split(":")[1]) for x in re.search(r"limbs=(\S+)", ln).group(1).strip(",").split(",")]
        out.append((ln.split()[0], ln.split()[1], owners))
    return out


def budget(world, plan):
    pls = plans(world, plan)
    rows = []
    for kind, name, owners in collectives(pls):
        rec = [received(kind, owners, world, me) for me in range(world)]
        rows.append((kind, name, len(owners), max(r[0] for r in rec), max(r[1] for r in rec)))
    return pls, rows


def received_per_rank(world, plan):
    """bytes every rank receives per hmult (one-phase replicate): what tests/test_gpu_sharded_inproc.py compares with the bytes its transport moved"""
    cs = collectives(plans(world, plan))
    return [sum(received(kind, owners, world, me)[0] for kind, _, owners in cs) for me in range(world)]


def model(world, plan, rows, pls, batch):
    link = LINK_GBS * LINK_EFF * 1e3    # bytes per us
    wire = [(r[4] / link) for r in rows]
    n_coll = len(rows)
    for i, r in enumerate(rows):        # the one-owner replicate in two phases when the batch makes it big enough
        if r[0] == "REPLICATE" and r[2] == 2 and world >= 4 and r[2] * batch * LIMB >= (2 << 20):
            wire[i] = 2 * (r[4] / (world - 1)) / link
            n_coll += 1
    hide = PIPE_HIDE if plan == 1 else 0.0
    exch = sum(w * (1 - hide if rows[i][0].endswith("_COL") else 1) for i, w in enumerate(wire)) + n_coll * T_LAT_US / batch
    n_launch = sum(1 for l in pls[0] if l.split()[0] not in XK)
    comp = sum(ONE_GPU_STAGE_US.values()) / world + n_launch * LAUNCH_FLOOR_US / batch
    return comp, exch


def main():
    print(__doc__.split("Usage")[0])
    print(f"hmult {L}/{ELL}/{ALPHA}, N = 2^{LOGN}: one limb-poly = {LIMB >> 10} KiB; one GPU: {ONE_GPU_OPS:.0f} hmult/s ({1e6 / ONE_GPU_OPS:.0f} us per op)\n")
    for world in (2, 4, 8):
        for plan, pname in ((2, "gather"), (1, "all-to-all on column slices")):
            pls, rows = budget(world, plan)
            auto = (plan == 2) == (world <= 4)
            print(f"== {world} GPUs, plan {plan} ({pname}){'   <- shard_plan = 0 picks this one' if auto else ''}")
            print(f"   {'collective':58s} {'limbs':>5s} {'received/rank':>14s} {'largest peer':>13s} {'wire us':>8s}")
            for kind, name, n, tot, peer in rows:
                print(f"   {(kind + ' ' + name)[:58]:58s} {n:5d} {tot / 2**20:10.2f} MiB {peer / 2**20:9.2f} MiB {peer / (LINK_GBS * LINK_EFF * 1e3):8.1f}")
            tot = sum(r[3] for r in rows)
            print(f"   total received per rank and op: {tot / 2**20:.2f} MiB in {len(rows)} collectives")
            for batch in (1, 4, 16):
                comp, exch = model(world, plan, rows, pls, batch)
                t = comp + exch
                print(f"   model, batch {batch:2d}: compute {comp:6.1f} us + exchange {exch:6.1f} us = {t:6.1f} us per op -> {1e6 / t:6.0f} hmult/s = x{1e6 / t / ONE_GPU_OPS:.2f} of one GPU")
            print()


if __name__ == "__main__":
    main()
