"""A training step as HIP graphs: what `bench.py` times, as a reusable object.

    step = CapturedTrainStep(lambda: criterion(model(batch, cidx, batch.batch), y), flat, opt, clip_norm=1.0)
    for items in loader:                       # shape-stable batches through DeviceCollator(static=True): fixed input addresses
        collator(items).wait()
        loss = step()                          # graph A (zero, forward, backward, pack) | all-reduce when world > 1 | graph B (clip, optimiser)

The eager step of the stage-2 model is host-bound (~190 launches, 2.5-3 ms of Python per step at ESOL shapes, more on a loaded host); replayed from two
graphs it is the GPU's 2.2 ms.  Everything the captured step needs is in place in this package: no host synchronisation in the models when the batch comes
from `DeviceCollator` (the sizes ride on the index tensor), edge counts stay on the device, `FlatAdam` reads its learning rate from the device (a scheduler's
change reaches the captured launch) and `FlatGradients.clip_grad_norm_` never visits the host.  The all-reduce sits BETWEEN the two graphs, issued
synchronously on the replay stream (legal between replays; collectives that can precede a capture go through `parallel.all_reduce_group_stream`: DESIGN.md §6).

Reference: the step Lightning runs for `conan_fgw/src/model/graph_embeddings/common.py:246-266` under `trainer.py:177` (gradient_clip_val)."""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist

from .parallel import FlatGradients, _world, all_reduce_group_stream


class CapturedTrainStep:
    def __init__(self, loss_fn: Callable[[], torch.Tensor], flat: FlatGradients, optimizer, clip_norm: Optional[float] = None, warmup: int = 3,
                 force_collective: bool = False):
        """`loss_fn()` runs the forward pass and returns the 0-dim loss; it must read its inputs from FIXED device tensors (the views of a
        `DeviceCollator(static=True)`, or tensors updated in place).  `optimizer`: `parallel.FlatAdam` or a capturable torch optimiser
        (`torch.optim.Adam(..., capturable=True)`).  `warmup` eager steps run first (allocator pools, autograd's stream bookkeeping); they DO
        update the parameters.  Capture happens on a stream of its own; every rank of a process group must construct this object at the same point."""
        if not flat.flat.is_cuda:
            raise RuntimeError("CapturedTrainStep captures HIP graphs: GPU only")
        self.flat, self.opt, self.clip = flat, optimizer, clip_norm
        self.world = _world()
        self.collective = self.world > 1 or (force_collective and dist.is_available() and dist.is_initialized())
        dev = flat.flat.device
        self._seed = torch.full((), 1.0 / self.world if self.collective else 1.0, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.loss = None
        self._loss_fn = loss_fn
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(max(1, warmup)):
                self._eager()
            torch.cuda.synchronize(dev)
            if self.world > 1:                                    # every rank enters capture with its queue and its process group idle
                t = torch.zeros(1, device=dev)
                all_reduce_group_stream(t)
                torch.cuda.synchronize(dev)
            flat.suspend_overlap(True)
            try:
                self.graph_a = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_a, stream=self.stream, capture_error_mode="thread_local"):
                    self._fwd_bwd()
                    flat.pack()
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, stream=self.stream, pool=self.graph_a.pool(), capture_error_mode="thread_local"):
                    self._update()
            finally:
                flat.suspend_overlap(False)
        torch.cuda.current_stream(dev).wait_stream(self.stream)

    def _fwd_bwd(self):
        self.flat.zero()
        loss = self._loss_fn()
        self.flat.backward(loss, grad_scale=self._seed)           # seeded with 1 / world: the SUM all-reduce yields the mean
        self.loss = loss.detach()

    def _update(self):
        if self.clip is not None:
            self.flat.clip_grad_norm_(self.clip)
        self.opt.step()

    def _eager(self):
        self._fwd_bwd()
        self.flat.all_reduce_mean(force=self.collective and self.world == 1, prescaled=self.collective)
        self._update()

    def __call__(self) -> torch.Tensor:
        """One step.  Returns the loss tensor of the captured forward (a fixed device tensor: read it — `.item()` — only when needed)."""
        cur = torch.cuda.current_stream(self.flat.flat.device)
        self.stream.wait_stream(cur)                               # inputs written on the caller's stream (the collator's landing copy) are ordered before the replay
        with torch.cuda.stream(self.stream):
            self.graph_a.replay()
            if self.collective:
                dist.all_reduce(self.flat.flat, op=dist.ReduceOp.SUM)
            self.graph_b.replay()
        cur.wait_stream(self.stream)
        return self.loss
