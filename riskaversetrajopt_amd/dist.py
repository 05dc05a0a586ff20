"""Sample-axis sharding across GPUs: one process per GPU; the data-path collective is RCCL behind the C ABI
(``rato_comm_*`` of include/rato_saa.h, bound by ``device_comm``); torch.distributed is the control plane only
(rendezvous, shipping the RCCL unique id, barriers) and the transport of the gloo CPU tests.

Every ★ function of the hot path is independent per sample, so each rank owns a
contiguous block of samples and all per-sample outputs (G, g_up, Z) stay
sharded.  The only cross-sample quantities are (1) the mean of the final-
constraint linearization (drone_risk.py:294-296) and (2) the Monte-Carlo
statistics (fraction satisfied, VaR, CVaR).  Both are served by ONE collective
per evaluation: an all-gather of the packed record

    [ fp64 partial sums (n_sums) | fp32 Z shard (M_local) ]

after which every rank sums the partial sums in rank order (deterministic) and
runs the same exact selection on the full Z.  The payload is tiny (4 MB at
M = 1e6), i.e. latency-bound; an all-gather uses all point-to-point xGMI links
at once, which suits this size better than a ring all-reduce of histograms.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  Single process -> (0, 1, 0)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("RATO_FORCE_DIST") == "1"     # exercise the collective path on a single GPU
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # RATO_DIST_BACKEND=gloo: functional test of the multi-rank path where the ranks share one GPU
            backend = os.environ.get("RATO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if os.environ.get("RATO_SINGLE_GPU") == "1":
            local = 0                                   # every rank on cuda:0 (tests on a 1-GPU box; gloo only)
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_bounds(M_total, rank, world, equal=False):
    """Contiguous block of samples owned by ``rank`` (remainder spread over the first ranks).

    The exchange (``exchange`` / ``exchange_record`` / ``gather_concat``) and the sharded SCP solve need EQUAL shards:
    one all-gather of fixed-size records, ``M_total = M_local * world``.  ``equal=True`` refuses a batch that does
    not divide (drop ``M_total % world`` samples or pad the batch first); ``check_equal_shards`` is what the
    exchange itself calls, so that unequal shards fail on every rank instead of hanging RCCL."""
    base, rem = divmod(M_total, world)
    if equal and rem:
        raise ValueError(f"M_total={M_total} does not divide over {world} ranks: the sample-sharded path needs equal "
                         f"shards (use {M_total - rem} or {M_total + world - rem} samples)")
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def check_equal_shards(M_local, group=None, what="sample shards"):
    """Collective: every rank must hold the same value (by default: its number of samples).  One tiny all-gather, done
    once per Record / Model.shard(); raises on EVERY rank when the values differ (a mismatched RCCL all-gather would hang
    or corrupt silently, gloo errors on one rank only).  ``what`` names the quantity in the message."""
    if not (dist.is_available() and dist.is_initialized()):
        return [int(M_local)]
    world = dist.get_world_size(group)
    dev = "cpu" if dist.get_backend(group) == "gloo" else torch.device("cuda", torch.cuda.current_device())
    mine = torch.tensor([int(M_local)], dtype=torch.int64, device=dev)
    every = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(every, mine, group=group)
    counts = [int(v) for v in every.cpu()]
    if len(set(counts)) != 1:
        raise ValueError(f"{what} differ across ranks: {counts}; the exchange needs them equal on every rank")
    return counts


# ---- the collective itself: RCCL behind the C ABI (include/rato_saa.h: rato_comm_*) -----------------------------
_COMMS = {}


def device_comm(group=None):
    """The library's own RCCL communicator for ``group`` (created on first use; None when unavailable).

    torch.distributed only ships the 128-byte unique id (one broadcast) and agrees on success; the data-path
    collective is ``rato_comm_exchange`` / ``rato_comm_allgather`` on the caller's stream.  Used when every rank owns
    its own GPU (backend "nccl").  ``RATO_COMM=torch`` keeps torch.distributed's own collective instead (A/B, debug);
    gloo runs (CPU tests, ranks sharing one GPU) never get a communicator: RCCL refuses two ranks on one device."""
    if group in _COMMS:
        return _COMMS[group]
    comm = None
    if os.environ.get("RATO_COMM", "rccl") != "torch" and dist.get_backend(group) == "nccl":
        import ctypes as C
        import sys
        from . import _lib
        lib = _lib.load()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        dev = torch.device("cuda", torch.cuda.current_device())

        def agree(flag):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return int(t.item()) == 1

        status = lib.rato_comm_available()
        # every rank must be able to bind librccl BEFORE any of them enters the collective rato_comm_init: a rank that
        # cannot would return at once and leave the others blocked inside ncclCommInitRank
        if agree(status == 0):
            ident = (C.c_uint8 * 128)()
            status = lib.rato_comm_unique_id(ident) if rank == 0 else 0
            box = [bytes(ident) if (rank == 0 and status == 0) else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            handle = C.c_void_p()
            if box[0] is not None:          # the same on every rank (broadcast): all enter the collective init, or none
                status = lib.rato_comm_init(C.byref(handle), box[0], rank, world)
            else:
                status = -3
            if agree(status == 0):
                comm = handle
            elif status == 0:
                lib.rato_comm_destroy(handle)
        if comm is None:
            msg = (f"rato_comm_init failed (status {status} on rank {rank}): the library's RCCL communicator is not "
                   f"available")
            if os.environ.get("RATO_STRICT_COMM") == "1":
                # bench.py --strict-comm: a scaling run must not silently report another transport
                raise RuntimeError(msg + " and RATO_STRICT_COMM=1 forbids the torch.distributed fallback")
            if rank == 0:
                print("warning: " + msg + ": falling back to torch.distributed's RCCL collective", file=sys.stderr)
    _COMMS[group] = comm
    return comm


def transport(group=None):
    """What carries the exchange of ``group``: 'rato_comm (RCCL behind the C ABI)', 'torch.distributed (nccl)',
    'torch.distributed (gloo)' or 'none (single process)' -- reported in bench.py's line."""
    if not (dist.is_available() and dist.is_initialized()):
        return "none (single process)"
    if _COMMS.get(group) is not None:
        return "rato_comm (RCCL behind the C ABI)"
    return f"torch.distributed ({dist.get_backend(group)})"


def destroy_comms():
    from . import _lib
    for comm in _COMMS.values():
        if comm is not None:
            _lib.load().rato_comm_destroy(comm)
    _COMMS.clear()


def pack_record(sums64, Z32):
    """-> uint8 tensor [8*n_sums + 4*M_local] (sums first: 8-byte aligned)."""
    a = sums64.contiguous().view(torch.uint8).reshape(-1)
    b = Z32.contiguous().view(torch.uint8).reshape(-1)
    return torch.cat([a, b])


def unpack_records(buf, world, n_sums, M_local):
    """buf: uint8 [world * rec] -> (sums (world, n_sums) fp64, Z (world*M_local,) fp32)."""
    rec = 8 * n_sums + 4 * M_local
    buf = buf.view(world, rec)
    sums = buf[:, :8 * n_sums].contiguous().view(torch.float64).view(world, n_sums)
    Z = buf[:, 8 * n_sums:].contiguous().view(torch.float32).reshape(world * M_local)
    return sums, Z


def exchange(sums64, Z32, group=None, agreed=False):
    """The single collective of an evaluation.  ``sums64``: rank-local fp64 sums
    (any shape), ``Z32``: rank-local fp32 Z (M_local,) — equal M_local on all
    ranks (verified on every call unless ``agreed``: see ``gather_concat``).
    Returns (total_sums (like sums64), Z_all (world*M_local,))."""
    if not (dist.is_available() and dist.is_initialized()) or \
            (dist.get_world_size(group) == 1 and os.environ.get("RATO_FORCE_DIST") != "1"):
        return sums64, Z32
    world = dist.get_world_size(group)
    n_sums, M_local = sums64.numel(), Z32.numel()
    rec = pack_record(sums64.reshape(-1).to(torch.float64), Z32.to(torch.float32))
    out = gather_concat(rec, group, agreed, what="exchange record lengths (8 n_sums + 4 M_local bytes)")
    sums, Z_all = unpack_records(out, world, n_sums, M_local)
    total = sums[0].clone()
    for r in range(1, world):          # fixed (rank) order: bitwise identical on every rank
        total += sums[r]
    return total.view(sums64.shape), Z_all


class Record:
    """One rank's [fp64 sums | fp32 Z row] record in ONE buffer, so that the producers (sum_partials, the
    linearize / eval kernels) write straight into what the all-gather sends: no pack step, and one unpack launch
    (rato_unpack_records) on the receiving side.  ``z_row`` >= M_local is the producer's row stride (ld)."""

    def __init__(self, n_sums, M_local, device, z_row=None):
        z_row = M_local if z_row is None else int(z_row)
        if z_row < M_local:
            raise ValueError("z_row < M_local")
        z_row += z_row % 2                               # record length: a multiple of 8 bytes
        self.n_sums, self.M_local, self.rec_bytes = int(n_sums), int(M_local), 8 * int(n_sums) + 4 * z_row
        self.buf = torch.zeros(self.rec_bytes, dtype=torch.uint8, device=device)
        self.sums = self.buf[:8 * n_sums].view(torch.float64)
        self.Z_row = self.buf[8 * n_sums:].view(torch.float32)
        self.Z = self.Z_row[:M_local]
        self._all = self._Z_all = self._total = None

    def _buffers(self, world):
        if self._all is None or self._all.numel() != world * self.rec_bytes:
            dev = self.buf.device
            self._all = torch.empty(world * self.rec_bytes, dtype=torch.uint8, device=dev)
            self._Z_all = torch.empty(world * self.M_local, dtype=torch.float32, device=dev)
            self._total = torch.empty(self.n_sums, dtype=torch.float64, device=dev)
        return self._all, self._Z_all, self._total


def exchange_record(rec, group=None):
    """``exchange`` for a Record: -> (total sums (n_sums,) fp64, Z_all (world * M_local,) fp32)."""
    if not (dist.is_available() and dist.is_initialized()) or \
            (dist.get_world_size(group) == 1 and os.environ.get("RATO_FORCE_DIST") != "1"):
        return rec.sums, rec.Z
    world = dist.get_world_size(group)
    if not getattr(rec, "_checked", False):
        check_equal_shards(rec.M_local, group)           # once per record: unequal shards fail loudly on every rank
        rec._checked = True
    all_, Z_all, total = rec._buffers(world)
    comm = device_comm(group) if rec.buf.is_cuda else None
    if comm is not None:                                 # RCCL all-gather + unpack behind the C ABI, caller's stream
        from . import _lib
        _lib.check(_lib.load().rato_comm_exchange(comm, _lib.ptr(rec.buf), _lib.ptr(all_), rec.rec_bytes, rec.n_sums,
                                                  rec.M_local, _lib.ptr(total), _lib.ptr(Z_all),
                                                  _lib.current_stream()), "rato_comm_exchange")
        return total, Z_all
    if _staged(rec.buf, group):                          # gloo + device tensors: stage through the host
        host = torch.empty(all_.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(host, rec.buf.cpu(), group=group)
        all_.copy_(host)
    else:
        dist.all_gather_into_tensor(all_, rec.buf, group=group)
    if all_.is_cuda:
        from . import _lib
        lib = _lib.load()
        _lib.check(lib.rato_unpack_records(_lib.ptr(all_), world, rec.n_sums, rec.M_local, rec.rec_bytes,
                                           _lib.ptr(total), _lib.ptr(Z_all), _lib.current_stream()),
                   "rato_unpack_records")
    else:
        # HOST tensors only (the world-size-2 gloo tests of the exchange logic, which run without a GPU): the same
        # byte layout re-sliced with torch views.  This is record plumbing, not a CPU version of any kernel of the
        # hot path -- those exist on the device only and raise RatoError without the HIP library.
        v = all_.view(world, rec.rec_bytes)
        sums = v[:, :8 * rec.n_sums].contiguous().view(torch.float64).view(world, rec.n_sums)
        total.copy_(sums[0])
        for r in range(1, world):
            total += sums[r]
        Z_all.copy_(v[:, 8 * rec.n_sums:].contiguous().view(torch.float32)[:, :rec.M_local].reshape(-1))
    return total, Z_all


def comm_selfcheck(rec, group=None, exchange_fn=None):
    """Collective, before a multi-rank timed region: is the exchange the run is about to time CORRECT?

    The product exchange of ``rec`` (``exchange_record``: rato_comm_exchange -- RCCL behind the C ABI -- when every rank
    owns a GPU) is compared, bit for bit, with torch.distributed's own all-gather of the same records unpacked by
    plain tensor views; then the ranks compare digests of what they hold (the gathered Z and the rank-ordered totals
    must be IDENTICAL on every rank: each runs the same exact selection on them).  Every verdict is agreed with
    MIN / SUM all-reduces, so all ranks return the same dict:
      ok, bitwise_vs_torch_all_gather, identical_on_every_rank, rccl_ranks (ranks whose exchange went through
      rato_comm), world, transport.
    ``exchange_fn``: the exchange under test (default ``exchange_record``; the gloo tests inject a corrupted one)."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"ok": True, "world": 1, "rccl_ranks": 0, "transport": transport(group), "skipped": "single process"}
    world = dist.get_world_size(group)
    total, Z_all = (exchange_fn or exchange_record)(rec, group)
    total, Z_all = total.clone(), Z_all.clone()
    # reference leg: torch's collective on the raw records, unpacked on the host with views (control plane + plumbing)
    buf = rec.buf.cpu() if (rec.buf.is_cuda and dist.get_backend(group) == "gloo") else rec.buf
    ref = torch.empty(world * rec.rec_bytes, dtype=torch.uint8, device=buf.device)
    dist.all_gather_into_tensor(ref, buf.contiguous(), group=group)
    v = ref.cpu().view(world, rec.rec_bytes)
    sums = v[:, :8 * rec.n_sums].contiguous().view(torch.float64).view(world, rec.n_sums)
    total_ref = sums[0].clone()
    for r in range(1, world):
        total_ref += sums[r]
    Z_ref = v[:, 8 * rec.n_sums:].contiguous().view(torch.float32)[:, :rec.M_local].reshape(-1)
    same = (torch.equal(total.cpu().view(torch.int64), total_ref.view(torch.int64))
            and torch.equal(Z_all.cpu().view(torch.int32), Z_ref.view(torch.int32)))
    # digest of what this rank holds: exact integer sums of the bit patterns
    digest = torch.stack([Z_all.cpu().view(torch.int32).to(torch.int64).sum(),
                          (total.cpu().view(torch.int64) >> 11).sum()])
    dev = "cpu" if dist.get_backend(group) == "gloo" else rec.buf.device
    every = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(every, digest.to(dev), group=group)
    every = every.cpu().view(world, 2)
    identical = bool((every == every[0:1]).all())
    flags = torch.tensor([1 if same else 0, 1 if identical else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN, group=group)
    n_rccl = torch.tensor([1 if _COMMS.get(group) is not None else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(n_rccl, op=dist.ReduceOp.SUM, group=group)
    same, identical = bool(int(flags[0])), bool(int(flags[1]))
    return {"ok": same and identical, "bitwise_vs_torch_all_gather": same, "identical_on_every_rank": identical,
            "rccl_ranks": int(n_rccl.item()), "world": world, "transport": transport(group)}


# ---- small helpers for the sharded cutting-plane oracle (cvar_cuts.py) ---------------------------------------
def _staged(t, group):
    """gloo has no device collectives: stage through the host (tests: two ranks on one GPU)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def gather_concat(t, group=None, agreed=False, what="buffer lengths"):
    """1-D tensor, equal length on every rank -> the concatenation in rank order (on every rank).

    A mismatched RCCL all-gather hangs or corrupts silently, so the equal length is verified with one tiny collective
    (``check_equal_shards``) on EVERY call -- unless the caller states ``agreed=True``: the length follows from
    quantities the ranks have already agreed on collectively (``Model.shard()`` checks M; the cutting-plane solver
    checks its (M, S, n_u) once when it is built).  ``agreed`` is a property of the call site, the same on every rank;
    whether the check runs is never decided from a rank-local value (a per-rank cache keyed on the local length would
    send a rank whose length changed into the check while the others are already in the all-gather)."""
    world = dist.get_world_size(group)
    src = t.contiguous()
    if not agreed:
        check_equal_shards(src.numel(), group, what=what)
    comm = device_comm(group) if src.is_cuda else None
    if comm is not None:
        from . import _lib
        out = torch.empty(world * src.numel(), dtype=src.dtype, device=src.device)
        _lib.check(_lib.load().rato_comm_allgather(comm, _lib.ptr(src), _lib.ptr(out), src.numel() * src.element_size(),
                                                   _lib.current_stream()), "rato_comm_allgather")
        return out
    if _staged(src, group):
        host = src.cpu()
        out = torch.empty(world * host.numel(), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host, group=group)
        return out.to(t.device)
    out = torch.empty(world * src.numel(), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)
    return out


def sum_in_rank_order(t, group=None, agreed=False):
    """Sum of a small tensor over the ranks, added in rank order: bitwise identical on every rank."""
    world = dist.get_world_size(group)
    parts = gather_concat(t.reshape(-1), group, agreed, what="lengths of the summed vectors").view(world, -1)
    total = parts[0].clone()
    for r in range(1, world):
        total += parts[r]
    return total.view(t.shape)


def any_rank(flag, device, group=None):
    """Collective: True on every rank iff ``flag`` is true on at least one (one 4-byte MAX all-reduce)."""
    t = torch.tensor([1 if flag else 0], dtype=torch.int32)
    if dist.get_backend(group) != "gloo":
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t.item()))


def broadcast_from_rank0(arr, device, group=None):
    """numpy float64 array -> rank 0's copy on every rank."""
    import numpy as np
    t = torch.as_tensor(np.ascontiguousarray(arr, dtype=np.float64))
    if dist.get_backend(group) != "gloo":
        t = t.to(device)
    dist.broadcast(t, src=0, group=group)
    return t.cpu().numpy()


# ---- the N > 1 step as a two-slot software pipeline ---------------------------------------------------------------------
def _record_stream(obj, stream, _depth=0):
    """Mark every CUDA tensor reachable from ``obj`` (dict / list / tuple nesting) as in use on ``stream``."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif _depth < 4:
        if isinstance(obj, dict):
            for v in obj.values():
                _record_stream(v, stream, _depth + 1)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                _record_stream(v, stream, _depth + 1)


class PipelinedSteps:
    """[hot kernel of step n + 1]  beside  [exchange + statistics of step n].

    The one exchange of an evaluation and the exact selection on the gathered Z (``M_total`` samples on EVERY rank) are
    small, latency-bound work behind a kernel that saturates the store path: issued on the same stream they are paid
    in full on every step (C5: 0.154 ms of kernel + the sums + an all-gather + a selection over 1e6 samples).  Here a step
    ``produce``s into one of ``n_slots`` output slots on the caller's stream, and the ``consume`` of the PREVIOUS step --
    exchange, unpack, statistics, on the previous slot -- is issued to a side stream right behind it, so that it runs
    beside the producer of this step.  A slot is reused only after its consumer has finished (events).  ``drain()`` issues
    the last consumer and waits for both streams: results are complete, and identical to the serial order bit for bit
    (same kernels on the same buffers; tests/test_dist_gloo.py, tests/test_gpu_dist.py).

    Memory: ``produce`` should write into persistent per-slot buffers (bench.py does).  If it returns freshly allocated
    tensors instead, the consumer reads them on the side stream after this object has dropped its last reference, and the
    caching allocator would hand the block to the next main-stream allocation while the side stream still reads it: every
    tensor reachable from what ``produce`` returned (dicts / lists / tuples are walked) is therefore marked as in use on
    the side stream (``Tensor.record_stream``) before the consumer is issued.

    Host tensors / no GPU (the gloo tests of the logic): no streams, the same ISSUE order -- produce(n + 1), then
    consume(n)."""

    ISSUED_KEPT = 256                                     # the issue log is a diagnostic: bounded

    def __init__(self, n_slots=2, device=None, high_priority=False):
        import collections
        self.n_slots = int(n_slots)
        self.cuda = bool(device is not None and torch.device(device).type == "cuda")
        self.pending = None
        self.results = [None] * self.n_slots
        # ("produce" | "consume", step): the order work was issued in (the last ISSUED_KEPT entries)
        self.issued = collections.deque(maxlen=self.ISSUED_KEPT)
        self._step = 0
        if self.cuda:
            self.main = torch.cuda.current_stream(device)
            # high_priority: the consumer's few workgroups are dispatched ahead of the producer's queue of tiles (A/B knob)
            self.side = torch.cuda.Stream(device, priority=-1) if high_priority else torch.cuda.Stream(device)
            self.ev_done = [torch.cuda.Event() for _ in range(self.n_slots)]
            self.ev_free = [torch.cuda.Event() for _ in range(self.n_slots)]

    def step(self, produce, consume):
        """``produce(slot)`` -> anything (handed to ``consume(slot, produced)`` one step later, whose return value lands in
        ``results[slot]``).  -> slot used."""
        slot = self._step % self.n_slots
        if self.cuda:
            self.main.wait_event(self.ev_free[slot])     # the consumer that last read this slot has finished
        out = produce(slot)
        self.issued.append(("produce", self._step))
        if self.cuda:
            self.ev_done[slot].record(self.main)
        self._flush()                                     # the previous step's consumer: beside this step's producer
        self.pending = (slot, consume, out, self._step)
        self._step += 1
        return slot

    def _flush(self):
        if self.pending is None:
            return
        slot, consume, out, n = self.pending
        self.pending = None
        if self.cuda:
            _record_stream(out, self.side)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ev_done[slot])
                self.results[slot] = consume(slot, out)
                self.ev_free[slot].record(self.side)
        else:
            self.results[slot] = consume(slot, out)
        self.issued.append(("consume", n))

    def drain(self):
        self._flush()
        if self.cuda:
            self.side.synchronize()
            self.main.synchronize()
