"""Data-parallel runtime: one process per GPU, gradients averaged with RCCL over xGMI.

The reference's only parallelism is single-process ``nn.DataParallel`` (train_multi_stylegan.py:67-70), which
re-broadcasts 415 MB of parameters per forward and reduces gradients onto GPU 0.  Here every rank owns a full
replica and the one exchange per optimiser step is a bucketed all-reduce:

* gradients live inside a few flat fp32 buffers (``param.grad`` are views), so a bucket is ready to send the moment
  its last gradient has been accumulated -- no gather copy -- and zeroing / norm / clipping touch a handful of
  tensors instead of hundreds;
* buckets are launched from post-accumulate-grad hooks on a side stream and overlap the rest of backward;
* parameters that can never receive a gradient (the generator's dead second-stream convolutions) are not
  registered at all, so no bucket waits for them.

xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by one link, so buckets are
sized (default 32 MiB) to keep several of them in flight rather than one huge one, and RCCL is left to pick the
algorithm per message.
"""
import os
from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _FORCE_COLLECTIVES) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


# MSG_FORCE_COLLECTIVES=1: run every collective even with ONE rank (a one-rank RCCL communicator is legal).  That is how
# the RCCL code paths -- broadcast at start-up, hook -> side stream -> async all-reduce -> wait, the scalar all-reduce of
# the path-length mean -- are exercised on a single-GPU box (tests/test_hip_ddp.py); results are unchanged.
_FORCE_COLLECTIVES = bool(int(os.environ.get("MSG_FORCE_COLLECTIVES", "0")))


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def collectives_active() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_COLLECTIVES)


def broadcast_module(module: torch.nn.Module, src: int = 0) -> None:
    """Identical initial replicas: parameters and buffers of ``module`` are overwritten with rank ``src``'s."""
    if not collectives_active():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)
    # `.data` writes bump no version counter: any kernel-side weight image cached by an earlier forward is stale now
    from . import conv_ops
    conv_ops.invalidate_weight_cache(list(module.parameters()))


def all_reduce_mean(t: torch.Tensor) -> torch.Tensor:
    if not collectives_active():
        return t
    out = t.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out / world_size()


def global_top_k(scores: torch.Tensor, v: float):
    """Top-k over the GLOBAL batch (reference loss.py:436-443 ranks the batch gathered on device 0): every rank
    contributes its [B] scores, the k = max(1, int(world * B * v)) largest of all of them are kept.  Returns the indices
    of this rank's kept samples and the weight k_local * world / k that makes the ranks' averaged mean-losses equal the
    mean over the k kept samples.  A rank that keeps nothing returns index [0] with weight 0 (its backward still runs,
    so the gradient exchange stays aligned).  One small all-gather and one device->host copy of k indices."""
    world, rank = world_size(), (dist.get_rank() if collectives_active() else 0)
    local = scores.reshape(-1)
    everything = [torch.empty_like(local) for _ in range(world)]
    if world > 1:
        dist.all_gather(everything, local.contiguous())
    else:
        everything = [local]
    flat = torch.cat(everything)
    k = max(1, int(flat.shape[0] * v))
    kept = torch.topk(flat, k=k).indices
    lo = rank * local.shape[0]
    mine = kept[(kept >= lo) & (kept < lo + local.shape[0])] - lo
    if mine.numel() == 0:
        return torch.zeros(1, dtype=torch.long, device=scores.device), 0.0
    return mine, mine.numel() * world / k


class _Bucket:
    __slots__ = ("flat", "params", "offsets", "pending", "work", "ready", "shard")

    def __init__(self, flat, params, offsets):
        self.flat, self.params, self.offsets = flat, params, offsets
        self.pending, self.work, self.ready = len(params), None, False
        self.shard = None                    # exchange == "reduce_scatter": this rank's 1/world slice of the reduced bucket


class _GradSlot:
    """Where a parameter's gradient lives inside its reducer's flat store (see grad_destination)."""
    __slots__ = ("reducer", "bucket", "offset", "taken")

    def __init__(self, reducer, bucket, offset):
        self.reducer, self.bucket, self.offset, self.taken = reducer, bucket, offset, False


_DIRECT_GRADS = True     # gradients are written straight into the flat store (False: through autograd's accumulation adds)


def grad_destination(p: torch.Tensor) -> Optional[torch.Tensor]:
    """A fresh view of the flat-store slice that holds ``p``'s gradient, for a backward kernel to write its result into --
    or None when the result has to go through autograd's own accumulation (no armed reducer, a second contribution to the
    same parameter in this backward, a higher-order pass).

    With ~280 parameters per optimiser step, autograd's AccumulateGrad costs one tiny `grad += incoming` launch each
    (1.4 ms of device time and as much host time per iteration).  While a reducer is armed for a labelled backward its
    parameters' ``.grad`` are None instead: AccumulateGrad then ADOPTS the incoming tensor instead of adding it, and when that
    tensor already is the parameter's slice of the flat store -- because the kernel that computed it wrote it there -- the
    accumulation costs nothing and the store stays the single home of the gradients (hooks fire as usual)."""
    slot = p.__dict__.get("_msg_grad_slot") if isinstance(p, torch.nn.Parameter) else None
    if slot is None:
        return None
    red = slot.reducer
    if not (red._armed and red._detached) or slot.taken or p.grad is not None or torch.is_grad_enabled():
        return None
    slot.taken = True
    return slot.bucket.flat[slot.offset:slot.offset + p.numel()].view(p.shape)


BUCKET_ALIGN = 64        # elements (256 bytes): every parameter's slice of a flat store starts on this boundary


def bucket_layout(params) -> Tuple[List[int], int]:
    """Element offsets of `params` in a flat store and its total length.  Slices start on 256-byte boundaries: the
    parameters themselves become views of such a store (multi_stylegan_amd.optim) and the kernels' 16-byte fast paths
    test their operands' alignment (an unaligned weight sent msg_modulate_backward down its scalar path, 2.4x slower)."""
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off = (off + p.numel() + BUCKET_ALIGN - 1) // BUCKET_ALIGN * BUCKET_ALIGN
    return offsets, off


class GradBucketReducer:
    """Flat-bucket gradient store + overlapped all-reduce for one set of parameters.

    Usage per optimiser step:  ``zero_grad()`` -> ``arm()`` -> backward -> ``finish()`` (waits, averages) ->
    ``clip_(max_norm)`` -> optimiser step.  With world size 1 the collectives vanish but the flat store, the
    one-pass zeroing and the sync-free clipping stay.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20,
                 overlap: bool = True, group=None, split_key=None, exchange: Optional[str] = None):
        """``split_key(parameter)``: parameters with different keys never share a bucket (the trainer passes the
        optimiser hyper-parameters, so that a bucket can be stepped by one launch -- multi_stylegan_amd.optim).
        ``exchange``: "all_reduce" (default; RCCL picks the algorithm per message) or "reduce_scatter" -- every bucket is
        reduce-scattered during backward (each rank receives 1/world of the summed bucket), the clip norm is taken from
        the local shards (+ one scalar all-reduce) and the shards are all-gathered back in ``finish``; same result on
        every rank, the two half-exchanges striped over all xGMI links.  Env: MSG_DDP_EXCHANGE."""
        params = [p for p in params if p.requires_grad]
        self.group, self.overlap = group, overlap
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or collectives_active()      # collectives are issued (see MSG_FORCE_COLLECTIVES)
        self.exchange = exchange or os.environ.get("MSG_DDP_EXCHANGE", "all_reduce")
        if self.exchange not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"exchange must be 'all_reduce' or 'reduce_scatter', not {self.exchange!r}")
        self.buckets: List[_Bucket] = []
        self._armed = False
        self._next = 0
        self._bucket_of = {}
        # Which buckets a given kind of backward fills: `arm(label)` learns, the first time a label is seen, which
        # parameters received a gradient (agreed over the ranks), and from then on buckets that this kind of backward never
        # completes ("cold": the biases in the R1 step, whose graph reaches no bias) are exchanged LAST instead of stalling
        # every bucket behind them until finish().  See _launch_ready.
        self._learned = {}
        self._label = None
        self._fired = None
        self._order: List[int] = []
        self._cold: List[int] = []
        self._shard_sumsq: Optional[torch.Tensor] = None
        self._detached = False              # the parameters' .grad are None for the armed backward (see grad_destination)
        self._zeroed = True                 # the store is all zeros (fresh, or zero_grad() since the last labelled backward)
        self.direct = _DIRECT_GRADS
        self._slot_of = {}
        cap = max(1, bucket_bytes // 4)
        # gradients become ready roughly in reverse registration order: fill buckets from the back
        chunk, size = [], 0
        groups = []
        key_of = split_key if split_key is not None else (lambda _p: None)
        for p in reversed(params):
            if size and (size + p.numel() > cap or key_of(p) != key_of(chunk[-1])):
                groups.append(chunk); chunk, size = [], 0
            chunk.append(p); size += p.numel()
        if chunk:
            groups.append(chunk)
        for ps in groups:
            dev = ps[0].device
            offsets, total = bucket_layout(ps)
            if self.exchange == "reduce_scatter":                             # equal, 256-byte-aligned shards
                unit = self.world * BUCKET_ALIGN
                total = (total + unit - 1) // unit * unit
            flat = torch.zeros(total, dtype=torch.float32, device=dev)        # (padding stays zero: zero gradients)
            for p, off in zip(ps, offsets):
                assert p.dtype == torch.float32 and p.device == dev
                p.grad = flat[off:off + p.numel()].view_as(p)
            bucket = _Bucket(flat, ps, offsets)
            if self.exchange == "reduce_scatter" and self.active:
                bucket.shard = torch.zeros(total // self.world, dtype=torch.float32, device=dev)
            self.buckets.append(bucket)
            for p, off in zip(ps, offsets):
                self._bucket_of[p] = bucket
                self._slot_of[p] = len(self._slot_of)
                p.__dict__["_msg_grad_slot"] = _GradSlot(self, bucket, off)
                p.register_post_accumulate_grad_hook(self._on_grad)
        self.comm_stream = None
        if self.active and params and params[0].is_cuda:
            self.comm_stream = torch.cuda.Stream(device=params[0].device)

    # -------------------------------------------------------------------------------------------------
    def numel(self) -> int:
        return sum(b.flat.numel() for b in self.buckets)

    def zero_grad(self) -> None:
        self._zeroed = True
        for b in self.buckets:
            b.flat.zero_()
            for p in b.params:                      # re-attach if something replaced .grad (e.g. set_to_none)
                if p.grad is None or p.grad.data_ptr() < b.flat.data_ptr() or \
                        p.grad.data_ptr() >= b.flat.data_ptr() + b.flat.numel() * 4:
                    self._reattach(b)
                    break

    def _reattach(self, b: _Bucket) -> None:
        for p, off in zip(b.params, b.offsets):
            view = b.flat[off:off + p.numel()].view_as(p)
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            p.grad = view

    def arm(self, label: Optional[str] = None) -> None:
        """Call right before the backward whose gradients this reducer owns.  ``label`` names the KIND of backward (the
        trainer passes "d", "r1", "g", "pl", ...): backwards of one kind reach the same parameters on every rank and in
        every iteration, which is what lets cold buckets be taken out of the in-order launch sequence."""
        self._armed = True
        self._shard_sumsq = None
        self._next = 0                                   # collectives are issued in ONE order on every rank: self._order
        self._label = label
        plan = self._learned.get(label) if label is not None else None
        self._fired = set() if (label is not None and plan is None) else None
        for k, b in enumerate(self.buckets):
            b.pending = len(b.params) if plan is None else plan[k]
            b.work, b.ready = None, False
        dirty_before = not self._zeroed
        self._zeroed = False                             # labelled or not: the backward that follows writes the store
        if plan is None:
            self._order, self._cold = list(range(len(self.buckets))), []
        else:
            self._order = [k for k, n in enumerate(plan) if n > 0]
            self._cold = [k for k, n in enumerate(plan) if n == 0]
        if label is not None and self.direct and self.buckets and self.buckets[0].flat.is_cuda:
            # (labelled = the trainer's backwards; the store was zeroed by zero_grad(), so a parameter this backward does not
            #  reach reads as a zero gradient once finish() has re-attached the views)
            # The direct route OVERWRITES the store's slices, so it has no accumulate semantics: a labelled backward
            # must follow a zero_grad() (two armed backwards in a row would drop the first one's directly written gradients
            # and keep the ones routed through AccumulateGrad).  Enforced, not assumed.
            if dirty_before:
                raise RuntimeError("GradBucketReducer.arm(label): zero_grad() must run before every labelled backward "
                                   "(gradients are written, not accumulated, into the flat store)")
            self._detached = True
            for b in self.buckets:
                for q in b.params:
                    q.grad = None
                    q.__dict__["_msg_grad_slot"].taken = False

    def _attach_all(self) -> None:
        """Every parameter's .grad is its view of the flat store again (after a backward with detached gradients)."""
        self._detached = False
        for b in self.buckets:
            for q, off in zip(b.params, b.offsets):
                g = q.grad
                if g is None:
                    q.grad = b.flat[off:off + q.numel()].view_as(q)
                elif g.data_ptr() != b.flat.data_ptr() + 4 * off:
                    view = b.flat[off:off + q.numel()].view_as(q)
                    view.copy_(g)
                    q.grad = view

    def disarm(self) -> None:
        self._armed = False
        if self._detached:
            self._attach_all()

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        # ANY gradient that reaches the store dirties it -- also one that arrives while the reducer is not armed (plain
        # AccumulateGrad into the attached views, e.g. the discriminator's parameters during a generator step that does not
        # skip their weight gradients): the next arm(label) must then see a zero_grad() first
        self._zeroed = False
        if not self._armed:
            return
        b = self._bucket_of[p]
        if p.grad is not None and p.grad.data_ptr() != self._view_ptr(b, p):
            if self._detached:                     # a gradient whose producer did not write the store: move it in (this
                slot = p.__dict__["_msg_grad_slot"]                      # parameter only; the others are still to arrive)
                view = b.flat[slot.offset:slot.offset + p.numel()].view_as(p)
                view.copy_(p.grad)
                p.grad = view
            else:
                self._reattach(b)                  # autograd swapped the tensor (out-of-place accumulation)
        if b.ready:
            # the bucket is already on the wire: this gradient would be lost (or race with the collective).  Can only happen
            # if a backward of this label reached a parameter that the learned plan says it never reaches.
            raise RuntimeError(f"gradient of a parameter arrived after its bucket was exchanged (label {self._label!r}); "
                               "backwards with one label must reach the same parameters -- use distinct labels")
        if self._fired is not None:
            self._fired.add(self._slot_of[p])
        b.pending -= 1
        if b.pending == 0 and self.overlap and self.active:
            self._launch_ready()

    def _view_ptr(self, b, p):
        for q, off in zip(b.params, b.offsets):
            if q is p:
                return b.flat.data_ptr() + off * 4
        raise KeyError

    def _launch_ready(self) -> None:
        """Launch every bucket that is complete AND whose predecessors in ``self._order`` have been launched.  The order in
        which gradients become ready can differ between ranks (style mixing changes the generator's graph per rank), but
        collectives must be issued in one order everywhere: the hot buckets in index order, then (in finish) the cold
        ones.  Correctness never depends on the plan -- a bucket the plan calls hot but that stays incomplete is sent by
        finish(), and every bucket is exchanged exactly once -- only the overlap does."""
        while self._next < len(self._order) and self.buckets[self._order[self._next]].pending == 0:
            self._launch(self.buckets[self._order[self._next]])
            self._next += 1

    def _launch(self, b: _Bucket) -> None:
        b.ready = True
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream(b.flat.device))
            with torch.cuda.stream(self.comm_stream):
                b.work = self._collective(b)
        else:
            b.work = self._collective(b)

    def _collective(self, b: _Bucket):
        if self.exchange == "reduce_scatter":
            return dist.reduce_scatter_tensor(b.shard, b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _learn(self) -> None:
        """First backward of a label: agree over the ranks on the parameters it reached (a parameter counts if ANY rank saw
        its gradient) and turn that into per-bucket counts."""
        seen = torch.zeros(len(self._slot_of), dtype=torch.float32, device=self.buckets[0].flat.device)
        if self._fired:
            seen[sorted(self._fired)] = 1.0
        if self.active:
            dist.all_reduce(seen, op=dist.ReduceOp.MAX, group=self.group)
        hit = seen.cpu().tolist()
        self._learned[self._label] = [sum(1 for p in b.params if hit[self._slot_of[p]] > 0) for b in self.buckets]

    def finish(self, average: bool = True) -> float:
        """Reduce whatever has not been sent yet, wait for everything, turn sums into means.  With average=False the
        buckets keep the SUM over ranks and the factor that still has to be applied (1 / world) is returned, for a
        caller that can fold it into a later pass (the trainer folds it into fused Adam's grad_scale and saves one
        read-modify-write of every gradient per optimiser step)."""
        self._armed = False
        if self._detached:
            self._attach_all()
        if not self.active:
            return 1.0
        for k in self._order[self._next:] + self._cold:   # whatever is left (incomplete hot buckets, then the cold ones)
            self._launch(self.buckets[k])
        self._next = len(self._order)
        # Work.wait() orders the CURRENT stream behind the collective (which runs on the backend's own stream).  The second
        # half of the reduce_scatter exchange runs on comm_stream, so the waits must happen with comm_stream current --
        # waiting on the main stream left the shard norms free to read shards the reduce-scatters had not written yet.
        if self.comm_stream is not None:
            main = torch.cuda.current_stream(self.buckets[0].flat.device)
            with torch.cuda.stream(self.comm_stream):
                for b in self.buckets:
                    b.work.wait()
                if self.exchange == "reduce_scatter":
                    self._gather_shards()
                    self._shard_sumsq.record_stream(main)          # allocated on comm_stream, consumed on the main stream
            main.wait_stream(self.comm_stream)
        else:
            for b in self.buckets:
                b.work.wait()
            if self.exchange == "reduce_scatter":
                self._gather_shards()
        if self._fired is not None:
            self._learn()
            self._fired = None
        inv = 1.0 / self.world
        if not average:
            return inv
        torch._foreach_mul_([b.flat for b in self.buckets], inv)
        if self._shard_sumsq is not None:
            self._shard_sumsq = self._shard_sumsq * (inv * inv)
        return 1.0

    def _gather_shards(self) -> None:
        """reduce_scatter exchange, second half: the squared norm of this rank's shards (the global norm is then one scalar
        all-reduce away: every element of the reduced gradient lives in exactly one rank's shard), and the shards back
        into the flat buckets of every rank."""
        sq = torch.stack([torch.linalg.vector_norm(b.shard) for b in self.buckets]).square().sum()
        dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=self.group)
        self._shard_sumsq = sq
        for b in self.buckets:
            dist.all_gather_into_tensor(b.flat, b.shard, group=self.group)

    def grad_norm(self) -> torch.Tensor:
        if self._shard_sumsq is not None:
            return self._shard_sumsq.sqrt()              # (of the rank SUM, like the flat buckets hold after finish(False))
        norms = torch._foreach_norm([b.flat for b in self.buckets])
        return torch.linalg.vector_norm(torch.stack(norms))

    def clip_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ semantics on the flat store; no host synchronisation."""
        total = self.grad_norm()
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        torch._foreach_mul_([b.flat for b in self.buckets], coef)
        self._shard_sumsq = None                         # (the cached norm described the unclipped buckets)
        return total
