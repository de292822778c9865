"""Sharded MSM, sharded inner-product prover and sharded verdicts across the GPUs of one
node: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in
the CPU tests).

The MSM sum_i e_i * P_i is a sum over independent pairs, so it shards with no data-path
exchange: every rank runs the full single-GPU MSM on its own contiguous shard and
produces ONE 64-byte partial point.  The only exchange step is an all_gather of those
64-byte partials (EC addition is not an RCCL reduction operator, so "reduce" =
all_gather of raw bytes + a local fold with bpmi_ec_sum); it is latency-bound (64 B per
rank), link bandwidth is irrelevant.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world, rank):
    """Contiguous shard [lo, hi) of n items for `rank`; sizes differ by at most one."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_bytes(data: bytes, group=None):
    """All ranks contribute `data` (same length everywhere); returns the list of all
    contributions, in rank order.  One collective."""
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    if backend == "nccl":
        # one flat receive buffer and ONE device-to-host copy (a list of per-rank tensors costs a
        # synchronising copy per rank, which is most of the step's communication time at 64 bytes)
        flat = torch.empty(world * len(data), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(flat, mine, group=group)
        raw = flat.cpu().numpy().tobytes()
        return [raw[i * len(data): (i + 1) * len(data)] for i in range(world)]
    outs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(outs, mine, group=group)
    return [bytes(t.numpy().tobytes()) for t in outs]


class ShardedMSM:
    """msm(pts_bytes, scalar_bytes, n) -> 64 bytes and fold(points_bytes, k) -> 64 bytes are
    the two engine operations used; by default they are the HIP engine's.

    On RCCL the exchange runs on its OWN stream with its own small engine context (one more bpmi_ctx on that
    stream): the all_gather of the 64-byte partials and the fold kernel never queue behind the MSM kernels
    that the compute engine has already enqueued for the next step, so a pipelined caller
    (bpmi_msm_dev_enqueue / bpmi_msm_finish, bench.py) keeps both of its MSMs in flight while it combines."""

    def __init__(self, engine=None, group=None, msm=None, fold=None):
        if engine is None and (msm is None or fold is None):
            from .engine import default_engine
            engine = default_engine()
        self.engine = engine
        self.msm = msm or engine.msm_bytes
        self.fold = fold or engine.ec_sum_bytes
        self.msm_dev = getattr(engine, "msm_dev", None)
        self.device_fold = fold is None and hasattr(engine, "ec_sum_dev")
        self.group = group
        self._comm = None
        self._inflight = False      # the RCCL exchange uses ONE set of staging buffers: one combine_begin at a time
        self._ev = None
        self.exchange_us_sum, self.exchange_count = 0.0, 0

    def _comm_setup(self, world):
        """(stream, engine on that stream, pinned 64-byte staging, device send buffer, device receive buffer)"""
        if self._comm is None or self._comm[4].numel() != 64 * world:
            from .engine import Engine
            dev = torch.device("cuda", torch.cuda.current_device())
            stream = torch.cuda.Stream(dev)
            eng2 = Engine(device=dev.index, stream=stream.cuda_stream)
            pin = torch.empty(64, dtype=torch.uint8).pin_memory()
            mine = torch.empty(64, dtype=torch.uint8, device=dev)
            flat = torch.empty(64 * world, dtype=torch.uint8, device=dev)
            self._comm = (stream, eng2, pin, mine, flat, torch.empty(64, dtype=torch.uint8, device=dev), torch.empty(64, dtype=torch.uint8).pin_memory())
        return self._comm

    def combine(self, partial: bytes) -> bytes:
        """partial = this rank's 64-byte partial result -> the global result on every rank."""
        if not dist.is_initialized():
            return partial            # a lone process; with a process group -- even of one rank -- the collective runs
        if self.device_fold and dist.get_backend(self.group) == "nccl":
            # RCCL: gather straight into one device buffer and fold it there -- no copy of the partials back to
            # the host and up again; everything on the exchange stream (see the class docstring)
            return self.combine_wait(self.combine_begin(partial))
        if len(partial) != 64:
            raise ValueError("ShardedMSM: a partial result is one 64-byte point, got %d bytes" % len(partial))
        parts = all_gather_bytes(partial, self.group)
        return self.fold(b"".join(parts), len(parts))

    def combine_begin(self, partial: bytes):
        """Start combine() and return a handle for combine_wait(): on RCCL the copy up, the all_gather and the fold kernel
        are only QUEUED on the exchange stream, so a pipelined caller can enqueue its next MSM before it waits -- the fold
        kernel may have to wait for the running accumulate kernel to free a wave slot, and nobody should wait with it."""
        if len(partial) != 64:
            raise ValueError("ShardedMSM: a partial result is one 64-byte point, got %d bytes" % len(partial))
        if not dist.is_initialized():
            return ("done", partial)
        if self.device_fold and dist.get_backend(self.group) == "nccl":
            if self._inflight:
                raise RuntimeError("ShardedMSM.combine_begin: the previous exchange has not been collected (combine_wait) -- "
                                   "its staging buffers are still in use")
            world = dist.get_world_size(self.group)
            stream, eng2, pin, mine, flat, d_out, pin_out = self._comm_setup(world)
            # The exchange buffers, the exchange stream and the fold's ctx were created on the device that was current at the FIRST exchange and
            # are cached together (so they always agree with each other); what can drift is the process's CURRENT device -- RCCL would then run
            # the collective with another current device than the one its buffers live on.  Checked at every exchange.
            cur = torch.cuda.current_device()
            if cur != eng2.device or flat.numel() != 64 * world:
                raise RuntimeError("ShardedMSM: the exchange buffers and the fold ctx live on device %d, the current device is %d "
                                   "(set the device once, before the first exchange)" % (eng2.device, cur))
            self._inflight = True
            pin.copy_(torch.frombuffer(bytearray(partial), dtype=torch.uint8))
            if self._ev is None:
                self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            with torch.cuda.stream(stream):
                self._ev[0].record(stream)
                mine.copy_(pin, non_blocking=True)
                dist.all_gather_into_tensor(flat, mine, group=self.group)
                # same stream: the fold is ordered behind the collective
                eng2._ck(eng2.lib.bpmi_ec_sum_dev_enqueue(eng2.ctx, flat.data_ptr(), world, d_out.data_ptr()))
                pin_out.copy_(d_out, non_blocking=True)
                self._ev[1].record(stream)
            return ("rccl", None)
        return ("done", self.combine(partial))

    def combine_wait(self, handle) -> bytes:
        kind, value = handle
        if kind == "done":
            return value
        stream, pin_out = self._comm[0], self._comm[6]
        stream.synchronize()
        self._inflight = False
        # device-side duration of the exchange: copy up, all_gather, fold kernel, copy down (waits for wave slots included)
        self.exchange_us_sum += self._ev[0].elapsed_time(self._ev[1]) * 1e3
        self.exchange_count += 1
        return bytes(pin_out.numpy().tobytes())

    def exchange_us(self, reset=True):
        """Mean device microseconds of the RCCL exchanges (combine_begin .. combine_wait) since the last reset; None if there were none."""
        v = self.exchange_us_sum / self.exchange_count if self.exchange_count else None
        if reset:
            self.exchange_us_sum, self.exchange_count = 0.0, 0
        return v

    def multiexp_replicated(self, pts: bytes, scalars: bytes, n: int) -> bytes:
        """Every rank holds the full input; each computes its shard, then combine()."""
        if not dist.is_initialized():
            return self.msm(pts, scalars, n)
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        lo, hi = shard_bounds(n, world, rank)
        part = self.msm(pts[64 * lo: 64 * hi], scalars[32 * lo: 32 * hi], hi - lo)
        return self.combine(part)

    def multiexp_local_dev(self, d_pts, d_scalars, n_local: int) -> bytes:
        """Each rank already holds its own shard in device memory (weak scaling)."""
        return self.combine(self.msm_dev(d_pts, d_scalars, n_local))


def shard_verdicts(verify_one, items, group=None):
    """Batch verification sharded by proof: rank r verifies items[lo:hi]; verdict bytes
    (1 = accepted) are exchanged with one all_gather.  `verify_one(item) -> bool`."""
    if not dist.is_initialized():
        return [bool(verify_one(it)) for it in items]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (len(items) + world - 1) // world
    lo, hi = min(rank * per, len(items)), min((rank + 1) * per, len(items))
    mine = bytearray(per)
    for j, it in enumerate(items[lo:hi]):
        mine[j] = 1 if verify_one(it) else 0
    parts = all_gather_bytes(bytes(mine), group)
    flat = b"".join(parts)
    return [bool(flat[i]) for i in range(len(items))]


def cyclic_shard(seq, world, rank):
    """Element i lives on rank i mod world: i and i + n/2 are then
    on the same rank in every halving round while the local length is >= 2, and the folded
    vector is again cyclic, so the inner-product argument never moves data between rounds."""
    return seq[rank::world]


class ShardedFastNIProver2:
    """FastNIProver2 (reference src/innerproduct/inner_product_prover.py:48-110) with g, h, a, b
    distributed cyclically over the ranks: every constructor vector is THIS rank's
    cyclic_shard of the global one; u, P and the transcript are the same on all ranks.

    Per round each rank computes the L and R of its own shard (the cl * u / cr * u terms are
    linear in the shard too), ONE all_gather moves 128 bytes per rank, every rank folds the
    partials with bpmi_ec_sum, hashes the same transcript and folds its shard with the same
    challenge.  When one element per rank is left the ranks exchange it (bpmi_ipa_export,
    192 bytes each) and all finish the last log2(world) rounds redundantly.  The Proof2 is
    bit-identical to the single-GPU prover's."""

    def __init__(self, g, h, u, P, a, b, group, transcript=None, h_scale=None, engine=None, process_group=None, state=None):
        """state: an already device-resident IpaState over this rank's shard (engine.ipa_create_dev);
        g, h, a, b are then ignored."""
        from .utils.transcript import Transcript
        if state is None:
            assert len(g) == len(h) == len(a) == len(b)
            assert len(a) & (len(a) - 1) == 0          # 0 is allowed: a global vector shorter than the number of ranks (prove())
        self.state = state
        self.g, self.h, self.u, self.P, self.a, self.b, self.group = g, h, u, P, a, b, group
        self.h_scale = h_scale
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        assert self.world & (self.world - 1) == 0, "the cyclic layout needs a power-of-two number of ranks"
        if engine is None:
            from .engine import default_engine
            engine = default_engine()
        self.engine = engine
        self.transcript = Transcript()
        if transcript:
            self.transcript.digest += transcript
            self.init_transcript_length = len(transcript.split(b"&"))
        else:
            self.init_transcript_length = 1

    def _combine(self, Lb, Rb):
        """This rank's partial L, R -> the global ones: ONE all_gather of 128 bytes per rank."""
        parts = all_gather_bytes(Lb + Rb, self.pg)
        k = len(parts)
        return (self.engine.ec_sum_bytes(b"".join(p[:64] for p in parts), k),
                self.engine.ec_sum_bytes(b"".join(p[64:] for p in parts), k))

    def _layout(self, n_local):
        """Local lengths of all ranks (one 8-byte all_gather).  The cyclic layout needs the same power-of-two length
        >= 1 on every rank; a global vector SHORTER than the number of ranks (some shards empty, the others one element)
        is legal input -- the argument is then tiny and every rank runs all of it (see _prove_replicated)."""
        lens = [int.from_bytes(p, "little") for p in all_gather_bytes(int(n_local).to_bytes(8, "little"), self.pg)]
        if min(lens) == 0:
            if max(lens) > 1 or self.state is not None:
                raise ValueError("ShardedFastNIProver2: rank shards of lengths %r are not a cyclic layout "
                                 "(a vector shorter than the %d ranks must be given as host lists, one element or none per rank)" % (lens, self.world))
            return lens, True
        if len(set(lens)) != 1:
            raise ValueError("ShardedFastNIProver2: the cyclic layout needs equal shard lengths on all ranks, got %r" % (lens,))
        return lens, False

    def _prove_replicated(self, lens):
        """Global length < number of ranks: the ranks that hold an element publish it (192 or 224 bytes), every rank
        rebuilds the global vectors and runs the whole (at most log2(world) - 1 rounds) argument itself."""
        from .ec import pack_points, pack_scalars
        q = self.group.q
        have = len(self.a) == 1
        rec = bytes(224)
        if have:
            hs = 1 if self.h_scale is None else int(self.h_scale[0]) % q
            rec = pack_points(self.g) + pack_points(self.h) + pack_scalars(self.a, q) + pack_scalars(self.b, q) + hs.to_bytes(32, "little")
        parts = [p for p, ln in zip(all_gather_bytes(rec, self.pg), lens) if ln]
        n = len(parts)
        if n & (n - 1) or any(lens[i] == 0 for i in range(n)):
            raise ValueError("ShardedFastNIProver2: %d elements on ranks %r are not a cyclic layout of a power-of-two vector"
                             % (n, [i for i, ln in enumerate(lens) if ln]))
        scaled = any(p[192:224] != (1).to_bytes(32, "little") for p in parts)
        return self.engine.ipa_create(b"".join(p[:64] for p in parts), b"".join(p[64:128] for p in parts),
                                      b"".join(p[128:160] for p in parts), b"".join(p[160:192] for p in parts), n, self.u.to_le64(),
                                      b"".join(p[192:224] for p in parts) if scaled else None)

    def prove(self):
        from .ec import pack_points, pack_scalars
        from .innerproduct._rounds import run_rounds
        from .innerproduct.inner_product_verifier import Proof2
        from .utils.utils import ModP
        q = self.group.q
        eng = self.engine
        ub = self.u.to_le64()
        state = self.state
        replicated = False
        if self.world > 1:
            lens, replicated = self._layout(len(state) if state is not None else len(self.a))
        if replicated:
            state = self._prove_replicated(lens)
        elif state is None:
            state = eng.ipa_create(pack_points(self.g), pack_points(self.h), pack_scalars(self.a, q),
                                   pack_scalars(self.b, q), len(self.a), ub,
                                   None if self.h_scale is None else pack_scalars(self.h_scale, q))
        xs, Ls, Rs = [], [], []
        try:
            run_rounds(state, self.transcript, q, xs, Ls, Rs, self._combine if (self.world > 1 and not replicated) else None)
            if self.world > 1 and not replicated:
                g1, h1, a1, b1 = state.export()           # this rank's last element = global index `rank`
                state.close()
                parts = all_gather_bytes(g1 + h1 + a1 + b1, self.pg)
                state = eng.ipa_create(b"".join(p[:64] for p in parts), b"".join(p[64:128] for p in parts),
                                       b"".join(p[128:160] for p in parts), b"".join(p[160:192] for p in parts),
                                       self.world, ub)
                run_rounds(state, self.transcript, q, xs, Ls, Rs)
            a, b = state.finish()
        finally:
            state.close()
        return Proof2(ModP(a, q), ModP(b, q), xs, Ls, Rs, self.transcript.digest, self.init_transcript_length)
