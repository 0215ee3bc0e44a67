"""Frame-parallel wave schedule across the GPUs of one node (one process per GPU).

The reference parallelises pictures with frame threads that wait on row progress of their
reference pictures (pthread_frame.c:479-513, hevc.c:1951-1958).  Across GPUs the same
decomposition is: pictures are the units, decoded reference pictures are the one exchange step.
Per STEP every rank decodes the same amount of work (weak scaling):

    wave 0      one I picture per rank                         -> all-gather of the finished pictures
    wave 1..R-1 one reference B picture per rank, predicted    -> all-gather
                from two pictures of the previous wave, one of
                them decoded by ANOTHER rank
    tail        T non-reference B pictures per rank, predicted from pictures of the last two waves

All pictures of one wave are mutually independent, so N GPUs decode a wave concurrently; the
all-gather (RCCL over xGMI with the NCCL backend; one collective per wave instead of N broadcasts)
replicates the wave's finished reference pictures into every GPU's DPB.  Non-reference pictures
are never exchanged.  The collective is issued on the engine's stream, so no host sync is needed.

This module holds only the schedule and the exchange; executing a picture is delegated to a
backend (the HIP engine in production; tests drive the same schedule with a CPU checker over
gloo).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Tuple


@dataclass
class PicturePlan:
    name: Tuple            # ("ref", wave, rank) or ("tail", k)
    slice_type: int        # 0 I, 2 B
    refs: List[Tuple]      # names of the reference pictures (list index == OhFrame.ref_pics slot)
    seed: int


@dataclass
class StepPlan:
    world: int
    rank: int
    waves: List[PicturePlan] = field(default_factory=list)     # this rank's reference picture of each wave
    tail: List[PicturePlan] = field(default_factory=list)      # this rank's non-reference pictures

    def pictures(self):
        return self.waves + self.tail


GOPS = ("ra", "ldp", "intra")


def make_step_plan(world, rank, n_waves=4, n_tail=12, seed=1, gop="ra"):
    """Deterministic plan; every rank can compute every rank's plan (used for validation).  gop:
      "ra"     the random-access shape above: I, reference B pictures on two pictures of the wave before (one of them another
               rank's), non-reference B pictures on the last two waves
      "ldp"    low delay P: every picture after the I picture is a P picture on ONE picture of the wave before — the NEIGHBOUR
               rank's, i.e. picture n of the chain is decoded on GPU n mod G from what GPU (n - 1) mod G produced; the tail's
               P pictures read the last wave
      "intra"  all intra: nothing references anything, nothing is exchanged"""
    assert gop in GOPS, gop
    plan = StepPlan(world, rank)
    for w in range(n_waves):
        s = seed * 1000003 + w * 1009 + rank * 7919
        if w == 0 or gop == "intra":
            plan.waves.append(PicturePlan(("ref", w, rank), 0, [], s))
        elif gop == "ldp":
            plan.waves.append(PicturePlan(("ref", w, rank), 1, [("ref", w - 1, (rank + 1) % world)], s))
        else:
            refs = [("ref", w - 1, rank), ("ref", w - 1, (rank + 1) % world)]
            plan.waves.append(PicturePlan(("ref", w, rank), 2, refs, s))
    last = n_waves - 1
    for k in range(n_tail):
        s = seed * 1000003 + 500 * 1009 + k * 104729 + rank * 7919
        if gop == "intra":
            plan.tail.append(PicturePlan(("tail", k), 0, [], s))
            continue
        if gop == "ldp":
            plan.tail.append(PicturePlan(("tail", k), 1, [("ref", last, (rank + k) % world)], s))
            continue
        a = ("ref", last, rank)
        b = ("ref", last - (k % 2) if last else 0, (rank + 1 + k) % world)
        plan.tail.append(PicturePlan(("tail", k), 2, [a, b] if a != b else [a, ("ref", last, (rank + 1) % world)], s))
    return plan


class Backend:
    """What the schedule needs from an executor."""

    def wave_tensor(self, wave):          # torch uint8 tensor [2 halves][world][half_bytes]
        raise NotImplementedError

    def execute(self, name):              # enqueue the picture's work list
        raise NotImplementedError

    def execute_batch(self, items):       # items: [(backend, name)], mutually independent pictures
        for be, name in items:
            be.execute(name)

    def final_half(self, name):
        raise NotImplementedError

    def set_final_half(self, name, half):
        raise NotImplementedError


def run_step(plan: StepPlan, backend: Backend, dist=None, group=None):
    """Enqueue one step.  dist: torch.distributed module (initialised) or None for world == 1;
    group: the process group this chain's collectives use (one group per chain in flight)."""
    for w, pic in enumerate(plan.waves):
        backend.execute(pic.name)
        if plan.world > 1:
            half = backend.final_half(pic.name)
            buf = backend.wave_tensor(w)[half]                    # [world][half_bytes], contiguous
            dist.all_gather_into_tensor(buf.view(-1), buf[plan.rank], group=group)
            for r in range(plan.world):
                if r != plan.rank:
                    backend.set_final_half(("ref", w, r), half)   # same SPS => same half on every rank
    for pic in plan.tail:
        backend.execute(pic.name)


def exchange_map(world, rank, n_waves, n_tail, gop="ra"):
    """Who references whose reference pictures: per wave (send_to, recv_from) of this rank.  A decoder knows its reference
    picture sets ahead of time, so a finished picture only has to reach the GPUs that will read it — the general case is the
    all-gather (everybody), this is the same exchange restricted to the readers."""
    need = {}
    for r in range(world):
        for pic in make_step_plan(world, r, n_waves=n_waves, n_tail=n_tail, gop=gop).pictures():
            for ref in pic.refs:
                if ref[2] != r:
                    need.setdefault((ref[1], ref[2]), set()).add(r)
    out = []
    for w in range(n_waves):
        send_to = sorted(need.get((w, rank), ()))
        recv_from = sorted(src for (ww, src), dsts in need.items() if ww == w and rank in dsts)
        out.append((send_to, recv_from))
    return out


def _staged(dist, group, tensor):
    """rehearsal only: the gloo backend cannot move device memory, so the transfer goes through a host copy (lets the
    N > 1 control flow run with several ranks on ONE GPU; RCCL moves the device buffers themselves)"""
    return tensor.is_cuda and dist.get_backend(group) == "gloo"


P2P_CHAINS = 8          # chains per batch_isend_irecv call (bounds the operations inside one RCCL group call)


class Turnstile:
    """One issue order for the exchanges of several host threads, the same on every rank.

    Each lockstep group of chains has its own host thread (hand-over and launches of one HIP stream), its own communicator and its
    own exchange stream; RCCL wants the operations of different communicators ENQUEUED in the same relative order on every rank
    (their kernels may share hardware queues).  Lane l's k-th exchange holds ticket k * lanes + l and is issued when every smaller
    ticket has been: all lanes make the same number of calls (same plan), a lane only ever waits for lanes that do not wait for it,
    and only the short enqueue is serialised — hand-over and launches of the lanes stay concurrent."""

    def __init__(self, lanes, timeout=600.0):
        import threading
        self.lanes, self.timeout = lanes, timeout
        self.calls = [0] * lanes
        self.next = 0
        self.failed = None
        self.cv = threading.Condition()

    def run(self, lane, fn):
        import time
        with self.cv:
            ticket = self.calls[lane] * self.lanes + lane
            deadline = time.monotonic() + self.timeout
            while self.next != ticket and self.failed is None:
                self.cv.wait(timeout=1.0)
                if self.next != ticket and self.failed is None and time.monotonic() > deadline:      # on every pass: a lane that is notified once a second would otherwise never time out
                    self.failed = TimeoutError(f"exchange ticket {ticket}: ticket {self.next} was never issued")
                    self.cv.notify_all()
            if self.failed is not None:
                raise RuntimeError("exchange order broken: another stream's thread failed") from self.failed
        try:
            fn()
        except BaseException as exc:
            self.fail(exc)
            raise
        with self.cv:
            self.calls[lane] += 1
            self.next += 1
            self.cv.notify_all()

    def fail(self, exc):
        """a lane's thread died: release the others (they raise)"""
        with self.cv:
            if self.failed is None:
                self.failed = exc
            self.cv.notify_all()


class Comm:
    """The exchange's own stream and its account.  The transfers of a wave are enqueued on `stream` behind an event recorded on the
    engine's stream after the wave's passes, and the engine's stream waits for the event that ends them — hipStreamWaitEvent both
    ways, no host wait — so the copy engines / xGMI links work while the other streams' passes (and this stream's hand-over of the
    next batch) go on.  bytes / messages / time are per rank (bench.py prints them)."""

    def __init__(self, torch, device=None, turnstile=None, lane=0):
        self.torch = torch
        self.turnstile, self.lane = turnstile, lane        # several groups on threads of their own: issue order (Turnstile)
        self.stream = torch.cuda.Stream(device) if device is not None and torch.cuda.is_available() else None
        self.bytes_sent = self.bytes_recv = self.messages = self.collectives = 0
        self.host_s = 0.0                                  # time of exchanges that ran on the host (CPU tensors / staged rehearsal)
        self._events = []

    def reset(self):
        self.bytes_sent = self.bytes_recv = self.messages = self.collectives = 0
        self.host_s = 0.0
        self._events = []

    def ms(self):
        """device time between the first and last operation of every exchange (call after a synchronize) + host-side exchanges"""
        return sum(a.elapsed_time(b) for a, b in self._events) + self.host_s * 1e3


def _exchange_grouped(chains, w, dist, exchange, comm):
    """One wave's reference pictures of ALL chains of a lockstep group: the group's buffers (GroupStore) keep what a rank
    contributes — its picture of every chain — contiguous, so the wave costs ONE all-gather, or with exchange_map ONE message per
    peer that references this rank's pictures, however many chains are in flight."""
    import time
    pl0, be0, group = chains[0]
    gs = be0.store.group
    torch = gs.torch
    half = be0.final_half(pl0.waves[w].name)
    assert all(be.final_half(pl.waves[w].name) == half for pl, be, _ in chains), "the chains of a group finish a wave in the same half"
    assert half == gs.final_half, "the group's buffers were laid out for pictures that finish in the other half (SAO on / off)"
    buf = gs.final[w]                                      # [world][n_chains][half_bytes], contiguous
    rank, world = pl0.rank, pl0.world
    send_to, recv_from = (list(r for r in range(world) if r != rank),) * 2 if exchange is None else exchange[w]
    on_dev = buf.is_cuda
    stage = _staged(dist, group, buf)
    cur = torch.cuda.current_stream() if on_dev else None
    use_stream = on_dev and comm is not None and comm.stream is not None
    t0 = t1 = None
    t_host = time.perf_counter()
    if use_stream:
        ready = torch.cuda.Event()
        ready.record(cur)
        comm.stream.wait_event(ready)                      # the wave's passes first
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx = torch.cuda.stream(comm.stream) if use_stream else _null_context()
    with ctx:
        if t0 is not None:
            t0.record()
        if exchange is None:
            if send_to:
                if stage:
                    host = buf.cpu()
                    dist.all_gather_into_tensor(host.view(-1), host[rank].clone().view(-1), group=group)
                    buf.copy_(host)
                else:
                    dist.all_gather_into_tensor(buf.view(-1), buf[rank].view(-1), group=group)
                if comm is not None:
                    comm.collectives += 1
        else:
            ops, landed = [], []
            mine = buf[rank].cpu() if stage and send_to else buf[rank]
            ops += [dist.P2POp(dist.isend, mine, dst, group) for dst in send_to]
            for src in recv_from:
                into = buf[src].cpu() if stage else buf[src]
                ops.append(dist.P2POp(dist.irecv, into, src, group))
                if stage:
                    landed.append((buf[src], into))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for dev, host in landed:
                dev.copy_(host)
            if comm is not None:
                comm.messages += len(ops)
        if t1 is not None:
            t1.record()
    if use_stream:
        cur.wait_event(t1)                                 # the next batch reads what arrived
        comm._events.append((t0, t1))
    elif comm is not None:
        comm.host_s += time.perf_counter() - t_host
    if comm is not None:
        per_rank = buf[rank].numel()
        comm.bytes_sent += per_rank * len(send_to)
        comm.bytes_recv += per_rank * len(recv_from)
    for pl, be, _ in chains:
        for src in recv_from:
            be.set_final_half(("ref", w, src), half)


class _null_context:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def run_steps_batched(chains, dist=None, exchange=None, comm=None):
    """Enqueue one step of EVERY chain in lockstep.  chains: list of (plan, backend, process group); the
    chains are independent GOPs with the same schedule, so wave w of all of them is one batch of
    mutually independent pictures (Backend.execute_batch: one launch per pass over the whole batch), and
    so is the union of their tails.  The exchange after a wave is one all-gather per chain, or — with exchange =
    exchange_map(...) — one batch of point-to-point transfers to the ranks that reference the pictures."""
    plan0, be0 = chains[0][0], chains[0][1]
    batches = [[(be, pl.waves[w].name) for pl, be, _ in chains] for w in range(len(plan0.waves))]
    tail = [(be, pic.name) for pl, be, _ in chains for pic in pl.tail]
    # a backend that hands work lists over per picture takes the NEXT batch's lists while the GPU still prepares / runs this one's
    # (what a decoder's host thread does: it is already parsing picture n + 1); the step after this one starts with wave 0 again
    ahead = getattr(be0, "hand_over_ahead", None)
    following = batches[1:] + ([tail] if tail else []) + [batches[0]]
    for w in range(len(plan0.waves)):
        if ahead is not None:
            be0.execute_batch(batches[w], then=following[w])
        else:
            be0.execute_batch(batches[w])
        if plan0.world > 1 and getattr(be0.store, "group", None) is not None:
            if comm is not None and comm.turnstile is not None:
                comm.turnstile.run(comm.lane, lambda: _exchange_grouped(chains, w, dist, exchange, comm))
            else:
                _exchange_grouped(chains, w, dist, exchange, comm)
        elif plan0.world > 1 and exchange is None:        # everybody gets everything: one all-gather per chain
            for pl, be, group in chains:
                half = be.final_half(pl.waves[w].name)
                buf = be.wave_tensor(w)[half]
                if _staged(dist, group, buf):
                    host = buf.cpu()
                    dist.all_gather_into_tensor(host.view(-1), host[pl.rank].clone(), group=group)
                    buf.copy_(host)
                else:
                    dist.all_gather_into_tensor(buf.view(-1), buf[pl.rank], group=group)
                for r in range(pl.world):
                    if r != pl.rank:
                        be.set_final_half(("ref", w, r), half)
        elif plan0.world > 1:                             # only to the ranks that reference the picture: one P2P batch per wave
            send_to, recv_from = exchange[w]
            by_group, done, landed = {}, [], []            # a batch must stay inside one communicator
            for k, (pl, be, group) in enumerate(chains):
                half = be.final_half(pl.waves[w].name)
                buf = be.wave_tensor(w)[half]
                # one point-to-point group per P2P_CHAINS chains: the cut falls between the same chains on every rank, so the
                # sends of a group always meet their receives in the peer's group of the same number
                ops = by_group.setdefault((id(group), k // P2P_CHAINS), [])
                stage = _staged(dist, group, buf)
                mine = buf[pl.rank].cpu() if stage and send_to else buf[pl.rank]
                ops += [dist.P2POp(dist.isend, mine, dst, group) for dst in send_to]
                for src in recv_from:
                    into = buf[src].cpu() if stage else buf[src]
                    ops.append(dist.P2POp(dist.irecv, into, src, group))
                    if stage:
                        landed.append((buf[src], into))
                done += [(be, ("ref", w, src), half) for src in recv_from]
            for ops in by_group.values():
                if ops:
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
            for dev, host in landed:
                dev.copy_(host)
            for be, name, half in done:
                be.set_final_half(name, half)
    if tail:
        if ahead is not None:
            be0.execute_batch(tail, then=batches[0])
        else:
            be0.execute_batch(tail)


def pictures_per_step(plan: StepPlan):
    return len(plan.waves) + len(plan.tail)


# ------------------------------------------------------------------------------------------------
# production backend: HIP engine + torch CUDA tensors (torch is plumbing: memory, stream, RCCL)
# ------------------------------------------------------------------------------------------------
def default_synth_knobs():
    """generator knobs of the benchmark stream (reported with every number)"""
    return dict(intra_pct=12, skip_pct=35, bi_pct=45, frac_mv_pct=75, mv_range=160, cbf_pct=60, split_pct=45,
                qp_base=30, qp_var=5, sao_pct=50)


class GroupStore:
    """The picture buffers of the chains that advance in LOCKSTEP on one stream, laid out for the exchange:
         wave w : final[w]   uint8 tensor [world][n_chains][half_bytes]   the FINISHED half of every rank's reference picture
                  scratch[w] uint8 tensor [n_chains][half_bytes]          the other half of THIS rank's pictures (a picture is
                                                                          reconstructed in half 0 and SAO writes half 1)
    so everything one rank contributes to a wave — its reference picture of EVERY chain — is one contiguous block: one message per
    peer and wave (or one all-gather per wave) whatever the number of chains in flight (_exchange_grouped); pictures of other ranks
    are only ever read, they need no second half (at 8 ranks: 0.9 GB of wave buffers per 4K Main 10 chain instead of 1.6).
         tail   : uint8 tensor [n_chains][n_tail][2][half_bytes]   (non-reference pictures, never exchanged)
    final_half: which half a finished picture lives in — 1 with SAO (every picture of the streams carries SAO parameters), else 0."""

    def __init__(self, torch, device, params, world, n_chains, n_waves, n_tail, final_half=None):
        from . import frame as F
        self.torch = torch
        self.half_bytes = F.half_layout(params)[0]
        self.n_chains = n_chains
        self.final_half = (1 if params.sao_enabled else 0) if final_half is None else final_half
        self.final = [torch.zeros((world, n_chains, self.half_bytes), dtype=torch.uint8, device=device) for _ in range(n_waves)]
        self.scratch = [torch.zeros((n_chains, self.half_bytes), dtype=torch.uint8, device=device) for _ in range(n_waves)]
        self.tail = torch.zeros((n_chains, max(n_tail, 1), 2, self.half_bytes), dtype=torch.uint8, device=device)


class PictureStore:
    """Names -> picture buffers laid out for the wave all-gather:
         wave w : uint8 tensor [2 halves][world][half_bytes]   (reference pictures of every rank)
         tail   : uint8 tensor [n_tail][2][half_bytes]         (this rank's non-reference pictures)
    group = (GroupStore, k): chain k's views into a lockstep group's buffers instead of buffers of its own."""

    def __init__(self, torch, device, params, plan, n_waves, n_tail, group=None):
        from . import frame as F
        self.half_bytes, self.strides, self.offsets = F.half_layout(params)
        self.params, self.plan = params, plan
        self.group, self.n_waves = None, n_waves
        if group is not None:
            self.group, self.k = group
            self.waves = None                              # the group's buffers: GroupStore.final / .scratch
            self.tail = self.group.tail[self.k]
            return
        self.waves = [torch.zeros((2, plan.world, self.half_bytes), dtype=torch.uint8, device=device) for _ in range(n_waves)]
        self.tail = torch.zeros((max(n_tail, 1), 2, self.half_bytes), dtype=torch.uint8, device=device)

    def halves(self, name):
        """(tensor of half 0, tensor of half 1) of a picture"""
        if name[0] == "ref":
            _, w, r = name
            if self.group is not None:
                fin = self.group.final[w][r][self.k]
                if r != self.plan.rank:
                    return fin, fin                        # another rank's picture: read only, one buffer
                scr = self.group.scratch[w][self.k]
                return (scr, fin) if self.group.final_half else (fin, scr)
            return self.waves[w][0][r], self.waves[w][1][r]
        return self.tail[name[1]][0], self.tail[name[1]][1]

    def names(self):
        out = [("ref", w, r) for w in range(self.n_waves) for r in range(self.plan.world)]
        return out + [p.name for p in self.plan.tail]


def host_work_lists(params, plan, knobs=None, pinned_by=None):
    """The host side of one GOP: {picture name: frame.FrameCopy} + {name: stats}.  What the reference's CTU loop would have recorded
    (here: the synthetic generator); picture ids are placeholders, EngineBackend binds them to its own pictures.
    pinned_by: the engine library — the lists are held in page-locked memory from oh_host_alloc, boundary strengths packed, and
    handed over by DMA (frame.FrameCopy)."""
    from . import frame as F
    knobs = dict(default_synth_knobs(), **(knobs or {}))
    rec = F.Recorder(params)
    lists, stats = {}, {}
    for pic in plan.pictures():
        sp = F.synth_params(pic.slice_type, pic.seed, n_refs=max(len(pic.refs), 0), **knobs)
        f = rec.synth(sp, 0, list(range(1, 1 + len(pic.refs))))
        stats[pic.name] = frame_stats(f)
        lists[pic.name] = F.FrameCopy(f, pinned_by=pinned_by)
    rec.close()
    return lists, stats


class EngineBackend(Backend):
    def __init__(self, torch, device_index, params, plan, knobs=None, engine=None, host_lists=None, resident=True, group=None):
        """engine: share another backend's engine (its stream, picture ids and batches); None = own engine
        on the current torch stream.
        host_lists: (lists, stats) of host_work_lists() to decode (several chains may share one host copy: every chain binds
        the lists to its OWN pictures); None = generate this plan's.
        resident: True = every work list is uploaded once, now, and execute() replays it (kernel-only timing);
        False = every execute() uploads the picture's work list first (validation, list preparation, H2D) and releases it
        right after the passes are enqueued — what a decoder does per picture."""
        from .engine import Engine
        self.torch, self.params, self.plan = torch, params, plan
        dev = torch.device("cuda", device_index)
        torch.cuda.set_device(dev)
        self.own_engine = engine is None
        self.engine = engine if engine is not None else Engine(device_index, stream=torch.cuda.current_stream().cuda_stream)
        self.store = PictureStore(torch, dev, params, plan, len(plan.waves), len(plan.tail), group=group)   # group: (GroupStore, chain index)
        self.ids: Dict[Tuple, int] = {}
        for name in self.store.names():
            h0, h1 = self.store.halves(name)
            self.ids[name] = self.engine.pic_wrap(params, h0.data_ptr(), h1.data_ptr(), self.store.half_bytes)
        lists, self.stats = host_lists if host_lists is not None else host_work_lists(params, plan, knobs)
        self.host_lists = lists                                   # keeps the arrays alive
        self.headers = {pic.name: lists[pic.name].with_ids(self.ids[pic.name], [self.ids[r] for r in pic.refs]) for pic in plan.pictures()}
        self.frames: Dict[Tuple, object] = {}
        self.resident = False
        self.upload_s, self.uploads = 0.0, 0                      # host time spent inside oh_frame_upload
        self._ahead = {}                                          # chunk key -> work lists handed over ahead of their execute
        if resident:
            self.make_resident()

    def make_resident(self):
        self.drop_ahead()
        for name, hdr in self.headers.items():
            self.frames[name] = self.engine.frame_upload(hdr)
        self.resident = True

    def drop_resident(self):
        self.drop_ahead()
        self.engine.sync()
        for df in self.frames.values():
            self.engine.frame_free(df)
        self.frames = {}
        self.resident = False

    def wave_tensor(self, wave):
        return self.store.waves[wave]

    def execute(self, name):
        self.execute_batch([(self, name)])

    hand_over_ahead = True                                    # run_steps_batched: execute_batch(items, then=next batch)

    def _hand_over(self, chunk):
        """oh_frames_upload of up to 32 work lists (validation, H2D on the copy stream, list preparation kernels behind it)"""
        import time
        key = tuple((id(be), name) for be, name in chunk)
        if key in self._ahead:
            return
        t0 = time.perf_counter()
        self._ahead[key] = self.engine.frames_upload([be.headers[name] for be, name in chunk])
        self.upload_s += time.perf_counter() - t0
        self.uploads += len(chunk)

    def execute_batch(self, items, then=None):
        assert all(be.engine is self.engine for be, _ in items), "a batch runs on one engine"
        if all(be.resident for be, _ in items):
            self.engine.frames_execute([be.frames[name] for be, name in items])
            return
        import time
        # one launch per pass covers at most 32 pictures: hand over that many, run them, let go.  The lists of the chunk that
        # comes NEXT (of this batch, else the first of `then`) are handed over before this chunk's passes are enqueued, so their
        # copy and preparation run while this chunk executes and the host never waits for a preparation it has just started.
        chunks = [items[c0:c0 + 32] for c0 in range(0, len(items), 32)]
        for i, chunk in enumerate(chunks):
            self._hand_over(chunk)
            dfs = self._ahead.pop(tuple((id(be), name) for be, name in chunk))
            nxt = chunks[i + 1] if i + 1 < len(chunks) else (then[:32] if then else None)
            if nxt:
                self._hand_over(nxt)
            t1 = time.perf_counter()
            self.engine.frames_execute(dfs)
            t2 = time.perf_counter()
            for df in dfs:
                self.engine.frame_release(df)
            self.execute_s = getattr(self, "execute_s", 0.0) + (t2 - t1)
            self.release_s = getattr(self, "release_s", 0.0) + (time.perf_counter() - t2)

    def drop_ahead(self):
        """work lists handed over ahead of a batch that never ran (end of a run)"""
        for dfs in self._ahead.values():
            for df in dfs:
                self.engine.frame_release(df)
        self._ahead = {}

    def final_half(self, name):
        return self.engine.pic_final_half(self.ids[name])

    def set_final_half(self, name, half):
        self.engine.pic_set_final_half(self.ids[name], half)

    def close(self):
        self.drop_ahead()
        self.drop_resident()
        if self.own_engine:
            self.engine.close()


def frame_stats(f):
    """sample counts of one work list that the algorithmic byte model needs:
    inter samples by number of reference lists, coded (residual) samples, intra samples."""
    import ctypes as C

    import numpy as np

    from . import frame as F
    p = f.p
    nplanes = F.n_planes(p)
    total = sum(F.plane_dims(p, c)[0] * F.plane_dims(p, c)[1] for c in range(nplanes))
    chroma_factor = total / float(p.width * p.height)           # 1.5 for 4:2:0
    st = dict(samples=total, luma=p.width * p.height, uni=0.0, bi=0.0, coded=0, intra=0, n_pu=int(f.n_pu),
              n_tu=int(f.n_tu), n_intra=int(f.n_intra), n_levels=int(f.n_levels))
    if f.n_pu:
        pu = np.frombuffer(C.string_at(f.pu, int(f.n_pu) * C.sizeof(F.OhPu)), dtype=np.dtype(
            [("x", "<u2"), ("y", "<u2"), ("w", "u1"), ("h", "u1"), ("r0", "u1"), ("r1", "u1"), ("mv", "<i2", (4,)), ("wp", "<u2"), ("rs", "<u2")]))
        area = pu["w"].astype(np.int64) * pu["h"]
        both = (pu["r0"] != F.OH_NO_REF) & (pu["r1"] != F.OH_NO_REF)
        st["bi"] = float(area[both].sum()) * chroma_factor
        st["uni"] = float(area[~both].sum()) * chroma_factor
    if f.n_tu:
        tu = np.frombuffer(C.string_at(f.tu, int(f.n_tu) * C.sizeof(F.OhTu)), dtype=np.dtype(
            [("x", "<u2"), ("y", "<u2"), ("c", "u1"), ("l2", "u1"), ("kind", "u1"), ("fl", "u1"), ("off", "<u4")]))
        st["coded"] = int((1 << (2 * tu["l2"].astype(np.int64))).sum())
    if f.n_intra:
        it = np.frombuffer(C.string_at(f.intra, int(f.n_intra) * C.sizeof(F.OhIntra)), dtype=np.dtype(
            [("x", "<u2"), ("y", "<u2"), ("c", "u1"), ("l2", "u1"), ("mode", "u1"), ("av", "u1"), ("tu", "<u4")]))
        st["intra"] = int((1 << (2 * it["l2"].astype(np.int64))).sum())
    return st


def algorithmic_bytes(stats, bytes_per_sample):
    """per pass, the bytes an ideal implementation must move for this picture.
       inter     (n_ref + 1)*b per inter-predicted sample (reference reads + prediction write)
       residual  2 B per coded sample (int16 coefficients)
       intra     b per intra sample (prediction write)
       deblock   2b per sample over the V and H launches together (b per launch)
       sao       2b per sample (read deblocked, write output)
    Their sum is the survey's 2*f_c + (n_ref + 5)*b per sample."""
    b = bytes_per_sample
    return dict(inter=(2 * stats["uni"] + 3 * stats["bi"]) * b, residual=2.0 * stats["coded"], intra=stats["intra"] * b,
                deblock_v=stats["samples"] * b, deblock_h=stats["samples"] * b, sao=2.0 * stats["samples"] * b)
