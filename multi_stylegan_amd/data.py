"""The data feed of the training step: host batches reach the GPU without ever stalling an iteration.

The reference feeds ``_gan_training`` from a ``DataLoader(num_workers=..., pin_memory=True)`` (train_multi_stylegan.py:60-63
over dataset/tlfm_dataset.py:128-198) and moves each batch with ``.to(device)`` at the top of the iteration
(model_wrapper.py:253-256) -- on the compute stream, i.e. in front of the iteration's first kernel, and synchronously
whenever the batch is pageable.  Here:

* ``DevicePrefetcher`` wraps ANY iterable of host batches (a DataLoader, a list, a generator; tensors or nested tuples of
  tensors; pinned or pageable).  A background thread pulls the next batch, stages it in a pinned slot, issues the H2D copy
  on a dedicated copy stream and hands the consumer a device tensor guarded by an event -- ``depth`` batches ahead
  (double buffering by default).  The compute stream only ever waits on an event that completed tens of milliseconds
  earlier; the slot is recycled once the compute stream has passed the last kernel that read it.
* ``SyntheticBatches`` draws the batches on the device itself (``torch.rand`` in the dataset's [0, 1] range,
  dataset/tlfm_dataset.py:187,191): what the benchmark and smoke runs use when there is no dataset.

``ModelWrapper.train`` / ``_gan_training`` put every host iterable behind a ``DevicePrefetcher`` themselves.
"""
import queue
import threading
from typing import Any, Iterable, Iterator, Optional, Union

import torch


def _map(fn, batch):
    if isinstance(batch, torch.Tensor):
        return fn(batch)
    if isinstance(batch, (list, tuple)):
        return type(batch)(_map(fn, b) for b in batch)
    if isinstance(batch, dict):
        return {k: _map(fn, v) for k, v in batch.items()}
    return batch


def _tensors(batch, out=None):
    out = [] if out is None else out
    if isinstance(batch, torch.Tensor):
        out.append(batch)
    elif isinstance(batch, (list, tuple)):
        for b in batch:
            _tensors(b, out)
    elif isinstance(batch, dict):
        for b in batch.values():
            _tensors(b, out)
    return out


def _stage(pinned: torch.Tensor, src: torch.Tensor) -> None:
    """pageable -> pinned on ONE core, without the GIL: numpy's copy loop.  (torch's CPU copy of a 25 MB tensor fans out over
    the intra-op thread pool, whose workers then compete with the thread that is launching the step's kernels.)"""
    try:
        import numpy
        numpy.copyto(pinned.numpy(), src.detach().contiguous().numpy())
    except (TypeError, RuntimeError):                     # dtypes numpy does not know (bfloat16)
        pinned.copy_(src)


class _Slot:
    """One in-flight batch: pinned staging tensors, device tensors, `ready` (copy done) and `released` (compute stream is
    past the consumer's last use) events."""

    def __init__(self, device):
        self.device = device
        self.host, self.dev, self.out = [], [], []
        self.ready = torch.cuda.Event()
        self.released: Optional[torch.cuda.Event] = None
        self.batch: Any = None
        self.keep: list = []                                # pinned source tensors of the copy in flight

    def fit(self, tensors, wire) -> None:
        """`wire[k]`: the dtype tensor k crosses the bus in (its own, or the prefetcher's transfer dtype)."""
        ok = len(tensors) == len(self.host) and all(h.shape == t.shape and h.dtype == w and o.dtype == t.dtype
                                                    for h, t, w, o in zip(self.host, tensors, wire, self.out))
        if not ok:                                          # first use, or a ragged last batch: (re)allocate this slot
            self.host = [torch.empty(t.shape, dtype=w, pin_memory=True) for t, w in zip(tensors, wire)]
            self.dev = [torch.empty(t.shape, dtype=w, device=self.device) for t, w in zip(tensors, wire)]
            # what the consumer sees: the landed tensor itself, or its widening back to the batch's dtype
            self.out = [d if w == t.dtype else torch.empty(t.shape, dtype=t.dtype, device=self.device)
                        for d, t, w in zip(self.dev, tensors, wire)]


class DevicePrefetcher:
    """``for batch in DevicePrefetcher(loader, device)``: device-resident batches, copied ``depth`` iterations ahead on a
    copy stream from pinned staging buffers.  A yielded batch is valid until the consumer asks for the next one (its device
    buffer is then refilled, ordered behind everything the consumer has launched so far); clone it to keep it longer.
    Batches that already live on the device pass through without a copy, but still at most ``depth`` ahead of the
    consumer (a lazy or infinite device-side generator is not drained into HBM)."""

    def __init__(self, loader: Iterable, device: Union[str, torch.device] = "cuda", depth: int = 2,
                 transfer_dtype: Optional[torch.dtype] = None):
        """``transfer_dtype``: floating-point host tensors cross the bus in this type and arrive as their original dtype
        (e.g. ``torch.bfloat16`` when the models compute in bf16 storage: the discriminator's first layer rounds its input to
        bf16 anyway, so training is bit-identical while the copy moves half the bytes).  Default: the batch's own dtype.
        Measured on the MI355X boxes: a 25 MB fp32 batch per step costs ~1.0 ms of a 110 ms step (0.9 %) even though it is
        copied a whole iteration ahead on its own stream (page-locked source or staged alike; the thread and queue cost
        nothing) -- against 1.6-2.0 ms for `.to(device)` at the top of the step."""
        self.loader, self.device, self.depth = loader, torch.device(device), max(2, int(depth))
        if self.device.type != "cuda":
            raise ValueError("DevicePrefetcher stages batches for a GPU; iterate the loader directly on the CPU")
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.transfer_dtype = transfer_dtype

    def __len__(self) -> int:
        return len(self.loader)

    def __iter__(self) -> Iterator:
        dev = self.device
        slots = [_Slot(dev) for _ in range(self.depth)]
        free = queue.Queue()
        for s in slots:
            free.put(s)
        # Bounded: batches that need no copy take no slot, so the slot hand-shake alone would let a loader that yields
        # device tensors lazily (or forever) be drained as fast as this thread runs -- an epoch resident in HBM.  At most
        # `depth` batches wait here, whatever they are.
        ready: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        _END, _ERR = object(), object()

        def put(item) -> bool:
            """Blocking put that gives up when the consumer has gone away."""
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    continue
            return False

        def worker():
            try:
                torch.cuda.set_device(dev)
                for batch in self.loader:
                    tensors = _tensors(batch)
                    if not tensors or all(t.is_cuda for t in tensors):
                        if not put((None, batch)):                        # nothing to copy
                            return
                        continue
                    slot = None
                    while slot is None:                                   # a free slot (the consumer returns them)
                        if stop.is_set():
                            return
                        try:
                            slot = free.get(timeout=0.05)
                        except queue.Empty:
                            continue
                    wire = [self.transfer_dtype if (self.transfer_dtype is not None and t.is_floating_point() and
                                                    not t.is_cuda) else t.dtype for t in tensors]
                    slot.fit(tensors, wire)
                    with torch.cuda.stream(self.copy_stream):
                        if slot.released is not None:                     # kernels that still read the device buffers
                            self.copy_stream.wait_event(slot.released)
                        slot.keep = []
                        for h, d, t, o in zip(slot.host, slot.dev, tensors, slot.out):
                            if t.is_cuda:
                                d.copy_(t, non_blocking=True)
                            elif h.dtype != t.dtype:
                                slot.ready.synchronize()
                                h.copy_(t)                                # narrowing cast into the pinned buffer
                                d.copy_(h, non_blocking=True)
                                o.copy_(d)                                # widened back on the device (copy stream)
                            elif t.is_pinned():
                                # already page-locked (DataLoader(pin_memory=True), the reference's loader): no staging; the
                                # source stays referenced until the slot is refilled, i.e. well past the copy
                                slot.keep.append(t)
                                d.copy_(t, non_blocking=True)
                            else:
                                # the pinned tensor of the slot's previous batch may still be the source of an in-flight
                                # copy: `ready` of that batch is behind us only once the copy stream has reached it
                                slot.ready.synchronize()
                                _stage(h, t)                              # pageable -> pinned
                                d.copy_(h, non_blocking=True)
                        slot.ready.record(self.copy_stream)
                    it = iter(slot.out)
                    slot.batch = _map(lambda _t: next(it), batch)
                    if not put((slot, slot.batch)):
                        return
                put((_END, None))
            except BaseException as exc:                                  # surfaces in the consumer
                put((_ERR, exc))

        thread = threading.Thread(target=worker, name="msg-prefetch", daemon=True)
        thread.start()
        try:
            while True:
                slot, batch = ready.get()
                if slot is _END:
                    return
                if slot is _ERR:
                    raise batch
                if slot is not None:
                    torch.cuda.current_stream(dev).wait_event(slot.ready)
                    for t in slot.out:
                        t.record_stream(torch.cuda.current_stream(dev))
                yield batch
                # the consumer is back for its next batch: every kernel that reads this one has been enqueued, so the slot
                # can be refilled behind an event recorded on the compute stream now
                if slot is not None:
                    slot.released = torch.cuda.Event()
                    slot.released.record(torch.cuda.current_stream(dev))
                    free.put(slot)
        finally:
            stop.set()
            thread.join(timeout=5.0)


class SyntheticBatches:
    """``steps`` batches ``[B, 2, 3, H, W]`` drawn on the device, uniform in [0, 1] like the dataset's normalised
    frames (dataset/tlfm_dataset.py:187,191).  ``fresh=False`` re-uses one resident batch (the benchmark's default)."""

    def __init__(self, steps: int, batch_size: int, resolution: int, device: Union[str, torch.device] = "cuda",
                 seed: int = 1234, fresh: bool = True, sequence_length: int = 3, channels: int = 2):
        self.steps, self.fresh = int(steps), fresh
        self.shape = (batch_size, channels, sequence_length, resolution, resolution)
        self.device = torch.device(device)
        self.generator = torch.Generator(device=self.device).manual_seed(seed)
        self._resident = None if fresh else torch.rand(self.shape, device=self.device, generator=self.generator)

    def __len__(self) -> int:
        return self.steps

    def __iter__(self) -> Iterator[torch.Tensor]:
        for _ in range(self.steps):
            yield torch.rand(self.shape, device=self.device, generator=self.generator) if self.fresh else self._resident


def prefetch(loader: Iterable, device: Union[str, torch.device], depth: int = 2,
             transfer_dtype: Optional[torch.dtype] = None) -> Iterable:
    """``loader`` behind a DevicePrefetcher when that helps (a GPU target and a loader that is not already one of this
    module's device-side feeds); otherwise the loader itself."""
    device = torch.device(device)
    if device.type != "cuda" or isinstance(loader, (DevicePrefetcher, SyntheticBatches)):
        return loader
    return DevicePrefetcher(loader, device, depth, transfer_dtype)
