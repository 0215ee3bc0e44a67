"""The N>1 path on CPU: two processes over gloo run the frame-parallel wave schedule
(openhevc_amd/parallel.py) with the CPU checker as executor; every rank must end up with the same
reference pictures as a single process that decodes ALL ranks' pictures itself."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _digest(planes):
    h = hashlib.md5()
    for pl in planes:
        h.update(np.ascontiguousarray(pl).tobytes())
    return h.hexdigest()


def _worker(rank, world, port, w, h, out_path, batched=False, p2p=False):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    from openhevc_amd import frame as F
    from openhevc_amd import parallel as P
    from oracle_backend import OracleBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = F.pic_params(w, h)
    plan = P.make_step_plan(world, rank, n_waves=3, n_tail=2, seed=5)
    be = OracleBackend(p, plan)
    if batched:                                        # a second, different chain runs in lockstep with the first
        plan2 = P.make_step_plan(world, rank, n_waves=3, n_tail=2, seed=6)
        be2 = OracleBackend(p, plan2)
        chains = [(plan, be, None), (plan2, be2, dist.new_group())]
        if p2p:                                        # a third chain on the first one's communicator, one chain per point-to-point
            P.P2P_CHAINS = 1                           # group call: the exchange of a wave is then cut into several calls
            plan3 = P.make_step_plan(world, rank, n_waves=3, n_tail=2, seed=7)
            chains.append((plan3, OracleBackend(p, plan3), None))
    for _ in range(2):                                 # two steps: buffers are reused across steps
        if batched:
            P.run_steps_batched(chains, dist, P.exchange_map(world, rank, 3, 2) if p2p else None)
        else:
            P.run_step(plan, be, dist)
    res = {str(n): _digest(be.picture(n)) for n in be.store.names()}
    torch.save(res, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def _grouped_worker(rank, world, port, w, h, out_path, gop, p2p, n_chains):
    """the production layout: the chains of a lockstep group share a GroupStore, a wave is exchanged as ONE all-gather or ONE
    message per peer (openhevc_amd/parallel.py _exchange_grouped), accounted in a Comm"""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    from openhevc_amd import frame as F
    from openhevc_amd import parallel as P
    from oracle_backend import OracleBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = F.pic_params(w, h)
    gs = P.GroupStore(torch, torch.device("cpu"), p, world, n_chains, 3, 2)
    chains = []
    for k in range(n_chains):
        plan = P.make_step_plan(world, rank, n_waves=3, n_tail=2, seed=5 + k, gop=gop)
        chains.append((plan, OracleBackend(p, plan, group=(gs, k)), None))
    comm = P.Comm(torch)
    ex = P.exchange_map(world, rank, 3, 2, gop=gop) if p2p else None
    for _ in range(2):
        P.run_steps_batched(chains, dist, ex, comm)
    be = chains[0][1]
    res = {str(n): _digest(be.picture(n)) for n in be.store.names()}
    res["comm"] = dict(sent=comm.bytes_sent, recv=comm.bytes_recv, messages=comm.messages, collectives=comm.collectives, half_bytes=gs.half_bytes)
    torch.save(res, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def _single_process_expectation(world, w, h, gop="ra"):
    """one process plays every rank in turn (wave by wave), no collectives"""
    sys.path.insert(0, HERE)
    from openhevc_amd import frame as F
    from openhevc_amd import parallel as P
    from oracle_backend import OracleBackend
    p = F.pic_params(w, h)
    plans = [P.make_step_plan(world, r, n_waves=3, n_tail=2, seed=5, gop=gop) for r in range(world)]
    backs = [OracleBackend(p, pl) for pl in plans]
    for _ in range(2):
        for wv in range(3):
            for r in range(world):
                backs[r].execute(plans[r].waves[wv].name)
            for r in range(world):                     # replicate by plain copies
                half = backs[r].final_half(plans[r].waves[wv].name)
                for o in range(world):
                    if o != r:
                        backs[o].store.waves[wv][half][r].copy_(backs[r].store.waves[wv][half][r])
                        backs[o].set_final_half(("ref", wv, r), half)
        for r in range(world):
            for pic in plans[r].tail:
                backs[r].execute(pic.name)
    return [{str(n): _digest(b.picture(n)) for n in b.store.names()} for b in backs]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("batched", [False, True], ids=["chain_per_stream", "lockstep_batches"])
def test_two_ranks_gloo_match_single_process(tmp_path, batched):
    world, w, h = 2, 128, 72
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, _free_port(), w, h, out, batched), nprocs=world, join=True)
    want = _single_process_expectation(world, w, h)
    for r in range(world):
        got = torch.load(f"{out}.{r}")
        assert got == want[r], f"rank {r} pictures differ from the single-process decode"
    # the exchanged reference pictures are identical on both ranks, the tails are rank-local
    a, b = torch.load(f"{out}.0"), torch.load(f"{out}.1")
    for k in a:
        if k.startswith("('ref'"):
            assert a[k] == b[k]


@pytest.mark.timeout(900)
def test_four_ranks_point_to_point_exchange(tmp_path):
    """lockstep batches with the reference pictures sent only to the ranks that reference them (exchange_map): every picture a
    rank DECODES must equal the single-process decode; pictures it never references may be missing in its DPB"""
    world, w, h = 4, 128, 72
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, _free_port(), w, h, out, True, True), nprocs=world, join=True)
    want = _single_process_expectation(world, w, h)
    from openhevc_amd import parallel as P
    for r in range(world):
        got = torch.load(f"{out}.{r}")
        plan = P.make_step_plan(world, r, n_waves=3, n_tail=2, seed=5)
        mine = {str(pic.name) for pic in plan.pictures()} | {str(ref) for pic in plan.pictures() for ref in pic.refs}
        for k in mine:
            assert got[k] == want[r][k], f"rank {r}: picture {k} differs from the single-process decode"
        ex = P.exchange_map(world, r, 3, 2)
        assert all(r not in s and r not in t for s, t in ex)
        assert sum(len(s) for s, _ in ex) < 3 * (world - 1)        # fewer transfers than replicating everything


@pytest.mark.timeout(900)
@pytest.mark.parametrize("gop,p2p", [("ra", False), ("ra", True), ("ldp", True), ("ldp", False), ("intra", True)])
def test_grouped_exchange_one_message_per_peer_and_wave(tmp_path, gop, p2p):
    """three chains in one lockstep group over two ranks: pictures equal the single-process decode for every GOP shape, and the
    exchange is counted: per wave ONE all-gather, or one send + one receive per peer that references the pictures — independent of
    the number of chains — carrying n_chains pictures each"""
    world, w, h, n_chains = 2, 128, 72, 3
    out = str(tmp_path / "res")
    mp.spawn(_grouped_worker, args=(world, _free_port(), w, h, out, gop, p2p, n_chains), nprocs=world, join=True)
    want = _single_process_expectation(world, w, h, gop)
    from openhevc_amd import parallel as P
    for r in range(world):
        got = torch.load(f"{out}.{r}")
        comm = got.pop("comm")
        plan = P.make_step_plan(world, r, n_waves=3, n_tail=2, seed=5, gop=gop)
        mine = {str(pic.name) for pic in plan.pictures()} | {str(ref) for pic in plan.pictures() for ref in pic.refs}
        for k in mine:
            assert got[k] == want[r][k], f"{gop}: rank {r}: picture {k} differs from the single-process decode"
        steps, waves = 2, 3
        if not p2p:
            assert comm["collectives"] == steps * waves and comm["messages"] == 0
            assert comm["sent"] == comm["recv"] == steps * waves * n_chains * comm["half_bytes"] * (world - 1)
        else:
            ex = P.exchange_map(world, r, 3, 2, gop=gop)
            n_msg = sum(len(a) + len(b) for a, b in ex)
            assert comm["messages"] == steps * n_msg and comm["collectives"] == 0
            assert comm["sent"] == steps * sum(len(a) for a, _ in ex) * n_chains * comm["half_bytes"]
            if gop == "intra":
                assert n_msg == 0                            # nothing references anything: nothing moves
            else:
                assert 0 < n_msg <= 2 * waves * (world - 1)


def test_gop_shapes():
    from openhevc_amd import parallel as P
    for world in (1, 2, 4):
        for r in range(world):
            ldp = P.make_step_plan(world, r, gop="ldp")
            assert [p.slice_type for p in ldp.waves] == [0, 1, 1, 1] and all(len(p.refs) == 1 for p in ldp.waves[1:] + ldp.tail)
            assert all(p.refs[0] == ("ref", p.name[1] - 1, (r + 1) % world) for p in ldp.waves[1:])     # picture n on GPU n mod G reads GPU n-1's
            intra = P.make_step_plan(world, r, gop="intra")
            assert all(p.slice_type == 0 and not p.refs for p in intra.pictures())
            assert P.pictures_per_step(ldp) == P.pictures_per_step(intra) == P.pictures_per_step(P.make_step_plan(world, r))


def test_plan_is_balanced_and_cross_rank():
    from openhevc_amd import parallel as P
    for world in (1, 2, 4, 8):
        plans = [P.make_step_plan(world, r) for r in range(world)]
        assert len({P.pictures_per_step(p) for p in plans}) == 1          # weak scaling: same work per rank
        if world > 1:
            for p in plans:
                assert any(ref[2] != p.rank for pic in p.waves[1:] for ref in pic.refs)   # needs another GPU's picture
        for p in plans:
            for pic in p.pictures():
                for ref in pic.refs:
                    assert ref[0] == "ref" and ref != pic.name


def test_turnstile_orders_the_exchanges_of_concurrent_threads():
    """bench.py at N > 1: every stream's host thread issues its own exchanges, P.Turnstile puts them in ONE order
    (call k of lane l = ticket k * lanes + l) whatever the threads' pace; a failing lane releases the others"""
    import random
    import threading
    import time
    from openhevc_amd import parallel as P
    lanes, calls = 4, 25
    ts = P.Turnstile(lanes, timeout=30.0)
    order, errs = [], []

    def work(lane):
        rng = random.Random(lane)
        try:
            for k in range(calls):
                time.sleep(rng.random() * 0.002 * (lane + 1))
                ts.run(lane, lambda: order.append((k, lane)))
        except BaseException as exc:            # noqa: BLE001
            errs.append(exc)
    ths = [threading.Thread(target=work, args=(lane,)) for lane in range(lanes)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs and order == [(k, lane) for k in range(calls) for lane in range(lanes)]

    ts = P.Turnstile(2, timeout=30.0)
    got = []

    def waiter():
        try:
            ts.run(1, lambda: got.append("ran"))
        except RuntimeError as exc:
            got.append(type(exc.__cause__).__name__)
    t = threading.Thread(target=waiter)
    t.start()
    with pytest.raises(ValueError):
        ts.run(0, lambda: (_ for _ in ()).throw(ValueError("lane 0 dies")))
    t.join(timeout=10)
    assert got == ["ValueError"]
