"""Data parallelism for the ConAN hot path: one process per GPU, molecules sharded across ranks, the gradients averaged
through ONE flat fp32 buffer (RCCL over xGMI through torch.distributed's "nccl" backend).

The reference uses Lightning's "ddp_find_unused_parameters_false" (conan_fgw/src/trainer.py:315-319) with a
DistributedSampler(shuffle=False) (data/datamodules.py:40-41).  The gradient payload is ~1.1 MB (SchNet-128), i.e. the
collective is latency-bound on xGMI (7 point-to-point links per GPU): it is issued on one contiguous buffer that the
parameters' .grad tensors alias afterwards — never per parameter.  In the eager step the buffer is cut into two buckets in
the order autograd produces the gradients: the first (output-side layers) is reduced on RCCL's own stream while the rest of
the backward still runs, the second right after backward.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(num_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block of molecules owned by `rank` (all K conformers of a molecule stay on one rank)."""
    base, rem = divmod(num_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_reduce_group_stream(t: torch.Tensor, op=None):
    """All-reduce issued with async_op=True and waited for on the calling stream: the collective and the completion event of
    its work object live on the process group's OWN stream, never on the caller's.

    Why this matters (round-3 SIGABRT of the RCCL step, diagnosed in round 4): ProcessGroupNCCL runs a *synchronous* collective
    directly on the caller's current stream and records the work's end event there; its watchdog thread polls that event
    (hipEventQuery) every 100 ms until it retires the work.  If the caller starts a HIP-graph capture on the same stream inside
    that window, HIP answers the watchdog's query with hipErrorCapturedEvent ("operation not permitted on an event last recorded
    in a capturing stream": the check looks at the *stream's* capture state, not at when the event was recorded), the watchdog
    thread throws and the process aborts.  Every collective issued before a capture on the same stream therefore goes through
    here; the one between two graph *replays* may stay synchronous (a replay is a launch, not a capture)."""
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM if op is None else op, async_op=True)
    w.wait()
    return t


def barrier_group_stream(device: torch.device):
    """dist.barrier() with the same property as all_reduce_group_stream (the NCCL barrier is an all-reduce on the caller's stream)."""
    t = torch.zeros(1, device=device)
    all_reduce_group_stream(t)
    if t.is_cuda:
        torch.cuda.synchronize(device)


class FlatGradients:
    """One contiguous fp32 buffer for all gradients: packed with one concatenation kernel per bucket, averaged across ranks
    with one all-reduce per bucket, and aliased back as every parameter's .grad for the optimizer.

    zero() drops the .grad tensors instead of clearing them: autograd's AccumulateGrad then adopts each freshly produced
    gradient without a per-parameter `grad += g` kernel (57 launches per step for SchNet), and the pack is one launch.

    Overlap (eager steps, world_size > 1): `enable_overlap()` registers post-accumulate hooks that only count.  The first
    backward records the order (and stream) in which gradients appear; `calibrate()` then lays the buffer out in that order
    and puts the leading gradients produced on the calling stream — about `early_fraction` of the bytes — into bucket 0.
    From then on the hook that completes bucket 0 packs it and starts its all-reduce asynchronously (the collective runs on
    the process group's stream, behind the work already queued on the current one) while backward continues;
    `all_reduce_mean()` packs and reduces bucket 1 and waits for both.  Gradients produced on other streams (the covalent
    branch runs on a side stream) always land in bucket 1, which is reduced after backward() has joined every stream.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._layout(list(range(len(self.params))), len(self.params))
        self._hooks: list = []
        self._overlap = False
        self._order: Optional[list] = None          # calibration record: (param index, produced on the calling stream?)
        self._pending = 0
        self._fired: set = set()
        self._work = None
        self.last_allreduce_launches = 0

    # ------------------------------------------------------------------------------------------------ layout
    def _layout(self, order: List[int], n_early: int):
        """Parameters in `order`; the first n_early of them form bucket 0 (== everything when no overlap is configured)."""
        self._order_idx = order
        self._views: List[Optional[torch.Tensor]] = [None] * len(self.params)
        off = 0
        self._split = 0
        for k, i in enumerate(order):
            p = self.params[i]
            self._views[i] = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            if k + 1 == n_early:
                self._split = off
        self._early = set(order[:n_early])
        self._n_early = n_early

    def zero(self):
        for p in self.params:
            p.grad = None
        self._pending = self._n_early if self._overlap else 0
        self._fired = set()
        self._work = None

    # ------------------------------------------------------------------------------------------------ pack
    def _pack(self, idx: List[int], out: torch.Tensor):
        pieces = []
        for i in idx:
            p, v = self.params[i], self._views[i]
            g = p.grad
            if g is None:
                g = torch.zeros_like(v)                 # parameter not reached by this step's graph
            elif g.data_ptr() == v.data_ptr():
                g = g.clone()                           # already aliased (packed twice): keep the value
            pieces.append(g.reshape(-1).to(torch.float32))
        if pieces:
            torch.cat(pieces, out=out)
        for i in idx:
            self.params[i].grad = self._views[i]

    def pack(self):
        """Gather the .grad tensors autograd produced into the flat buffer and alias them to it."""
        self._flush_deferred()
        self._pack(self._order_idx, self.flat)

    def _flush_deferred(self, only=None):
        """Weight gradients whose slab reduction was deferred (ops.deferred_weight_gradients) become valid here, in one launch.
        Deferral hands autograd tensors that are filled at this point, which is sound only if autograd adopted them as .grad: checked.

        `only` (a set of parameter indices): verify just those parameters and leave the other flushed entries for the next call.
        The early-bucket hook uses it: it runs inside the post-accumulate hook of the LAST early parameter, when sibling outputs
        of the same autograd node (the other weights of a fused two-layer function, a bias whose hook order differs) may not have
        been accumulated yet — their .grad is still None, which is not a copy.  They are verified at the end of backward."""
        if not self.flat.is_cuda:
            return
        from . import ops
        ops.flush_weight_gradients()
        if ops._flushed:
            grads = {p.grad.data_ptr() for p in self.params if p.grad is not None}
            for i, p in enumerate(self.params):
                key = p.data_ptr()
                want = ops._flushed.get(key)
                if want is None or (only is not None and i not in only):
                    continue
                dw_ptr, db_ptr = want
                if only is not None and db_ptr is not None and db_ptr not in grads and p.grad is not None and p.grad.data_ptr() == dw_ptr:
                    continue                                   # the bias of this layer has not been accumulated yet: checked at the end
                if p.grad is None or p.grad.data_ptr() != dw_ptr or (db_ptr is not None and db_ptr not in grads):
                    raise RuntimeError("a deferred weight gradient was copied by autograd before it was final (the parameter already had a "
                                       ".grad, or something else held the tensor): call FlatGradients.zero() before backward, or do not "
                                       "use deferred_weight_gradients here")
                del ops._flushed[key]
            if only is None:
                ops._flushed.clear()

    def backward(self, loss: torch.Tensor, grad_scale: Optional[torch.Tensor] = None):
        """loss.backward() with the slab reductions of all weight gradients batched into one launch (flushed before any gradient is
        packed or reduced).  Equivalent to `loss.backward()` in results.  `grad_scale` (a 0-dim tensor, e.g. 1 / world_size): the seed of the
        backward pass — the ranks' SUM all-reduce then yields the mean directly and no scaling pass over the flat buffer is needed
        (`all_reduce_mean(prescaled=True)`)."""
        if self.flat.is_cuda:
            from . import ops
            with ops.deferred_weight_gradients():
                loss.backward(grad_scale)
            self._flush_deferred()                               # (the context flushed; this verifies the adoption)
        else:
            loss.backward(grad_scale)

    # ------------------------------------------------------------------------------------------------ overlap
    def enable_overlap(self, early_fraction: float = 0.5):
        """Start recording the gradient production order; call `calibrate()` after one backward.  May be called again to re-cut the
        buffer at another fraction (the hooks are registered once)."""
        self._frac = float(early_fraction)
        self._order = []
        self._overlap = False
        if self._hooks:
            return
        main = torch.cuda.current_stream() if self.flat.is_cuda else None
        index = {id(p): i for i, p in enumerate(self.params)}

        def hook(p):
            i = index[id(p)]
            if self._order is not None:
                on_main = (not self.flat.is_cuda) or torch.cuda.current_stream() == main
                self._order.append((i, on_main))
            elif self._overlap and i in self._early:
                self._fired.add(i)
                self._pending -= 1
                if self._pending == 0:
                    self._fire_early()
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(hook))

    def calibrate(self):
        """Fix the buffer layout from the recorded production order.  Returns (gradients in bucket 0, gradients in total)."""
        rec, self._order = self._order or [], None
        seen, order, on_main = set(), [], []
        for i, m in rec:
            if i not in seen:
                seen.add(i)
                order.append(i)
                on_main.append(m)
        total, early_bytes, early = self.flat.numel(), 0, []
        for k, i in enumerate(order):           # bucket 0 = the first calling-stream gradients up to the byte target
            if not on_main[k]:
                continue                        # produced on a side stream: reduced after backward has joined the streams
            if early_bytes >= self._frac * total:
                break
            early_bytes += self.params[i].numel()
            early.append(i)
        es = set(early)
        order = early + [i for i in order if i not in es] + [i for i in range(len(self.params)) if i not in seen]
        self._overlap = 0 < len(early) < len(order) and _world() > 1
        if _world() > 1:
            # every rank must cut the buffer at the same element, or the two collectives would not match: agree on the layout
            # (min == max of a layout checksum over the ranks), else fall back to the single all-reduce everywhere
            chk = float(sum((k + 1) * (i + 1) for k, i in enumerate(early)) % 1000003) if self._overlap else -1.0
            t = torch.tensor([chk, -chk], dtype=torch.float64, device=self.flat.device)
            all_reduce_group_stream(t, op=dist.ReduceOp.MAX)
            if t[0].item() != chk or -t[1].item() != chk or chk < 0:
                self._overlap = False
        self._layout(order, len(early) if self._overlap else len(order))
        for p in self.params:                   # .grad views of the old layout are stale
            p.grad = None
        return (len(early) if self._overlap else 0), len(order)

    def suspend_overlap(self, flag: bool = True):
        """Turn the early bucket off / on again (e.g. while a HIP graph is captured: the collective is issued between replays)."""
        if flag and self._overlap:
            self._overlap, self._suspended = False, True
        elif not flag and getattr(self, "_suspended", False):
            self._overlap, self._suspended = True, False

    def _fire_early(self):
        # the early bucket's weight gradients must be final before they travel; only parameters whose hook has fired are verified
        # here (an unfired sibling of the same autograd node has .grad None until its own AccumulateGrad runs)
        self._flush_deferred(only=self._fired)
        idx = self._order_idx[: self._n_early]
        self._pack(idx, self.flat[: self._split])
        self._work = dist.all_reduce(self.flat[: self._split], op=dist.ReduceOp.SUM, async_op=True)

    # ------------------------------------------------------------------------------------------------ reduce
    def all_reduce_mean(self, force: bool = False, group_stream: bool = True, prescaled: bool = False):
        """Pack, sum across ranks, divide by the world size (`prescaled`: the backward pass was seeded with 1 / world_size — `backward(loss,
        grad_scale=...)` — so the sum already is the mean and the scaling pass is skipped).  `force`: issue the collective even in a 1-rank group (the same RCCL
        call the multi-rank step makes; used to exercise the path on one GPU).  `group_stream` (default): the collective runs on the
        process group's own stream (all_reduce_group_stream: safe to follow with a HIP-graph capture on the calling stream);
        False = synchronous call on the calling stream (two event hops fewer; only where no capture can follow)."""
        world = _world()
        self.last_allreduce_launches = 0
        if self._overlap and self._work is not None:
            self._pack(self._order_idx[self._n_early:], self.flat[self._split:])
            w2 = dist.all_reduce(self.flat[self._split:], op=dist.ReduceOp.SUM, async_op=True)
            self._work.wait(); w2.wait()
            self._work = None
            self.last_allreduce_launches = 2
            if not prescaled:
                self.flat.mul_(1.0 / world)
            return
        self.pack()
        if world > 1 or (force and dist.is_available() and dist.is_initialized()):
            if group_stream:
                all_reduce_group_stream(self.flat)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.last_allreduce_launches = 1
            if not prescaled:
                self.flat.mul_(1.0 / world)


    # ------------------------------------------------------------------------------------------------ clip
    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_(params, max_norm) on the packed buffer (call after pack() / all_reduce_mean(): the .grad tensors alias
        it), without the host round trip and the ~100-tensor foreach kernels: what Lightning runs for the reference's
        Trainer(gradient_clip_val=1.0) (trainer.py:177) between backward and the optimiser step.  Returns the total norm (0-dim, on the device)."""
        if not self.flat.is_cuda:
            return _clip_cpu(self.flat, max_norm)
        from ._lib import call, ptr, stream_ptr
        if getattr(self, "_clip_ws", None) is None:
            dev = self.flat.device
            self._clip_ws = (torch.zeros(2, dtype=torch.float32, device=dev), torch.zeros(256, dtype=torch.float64, device=dev),
                             torch.zeros(1, dtype=torch.int32, device=dev))
        out, part, ticket = self._clip_ws
        call("conan_grad_clip_flat", ptr(self.flat), self.flat.numel(), float(max_norm), ptr(out), ptr(part), ptr(ticket), stream_ptr())
        return out[0]


def _clip_cpu(flat: torch.Tensor, max_norm: float) -> torch.Tensor:
    """The same rule on a CPU buffer (gloo tests of the step's host logic)."""
    nrm = flat.double().norm()
    coef = torch.clamp(max_norm / (nrm + 1e-6), max=1.0)
    flat.mul_(coef.to(flat.dtype))
    return nrm.to(torch.float32)


class _LiveGroup(dict):
    """The param_group of FlatAdam: `group["lr"] = x` (what a scheduler or user code does with a plain float) lands in the device scalar the
    kernel reads, so that a captured launch follows it; the entry itself stays the device tensor (torch's own idiom for capturable optimisers:
    schedulers `fill_` a tensor lr in place)."""

    def __init__(self, owner, *a, **kw):
        super().__init__(*a, **kw)
        self._owner = owner

    def __setitem__(self, key, value):
        if key == "lr" and "lr" in self and isinstance(self["lr"], torch.Tensor) and value is not self["lr"]:
            self["lr"].fill_(float(value))
            return
        super().__setitem__(key, value)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam (defaults: no amsgrad, no maximize) for the parameters of a `FlatGradients`, as ONE launch per step
    (conan_adam_flat_step) instead of torch's multi-tensor kernels over ~100 parameter tensors.

    The parameters are moved into one flat fp32 buffer in the gradient buffer's order and every `Parameter.data` is re-pointed at its slice
    (the module keeps working unchanged: `state_dict`, `load_state_dict` and the kernels see the same tensors), the two moments are flat
    buffers of the same layout, the step counter lives on the device so that a captured HIP graph replays the launch.  `step()` expects the
    gradients in `flat.flat` — i.e. after `FlatGradients.pack()` / `all_reduce_mean()`.  GPU only (there is no CPU path in this package).

    It is a `torch.optim.Optimizer` as far as the reference's training loop needs one (common.py:253-262: Adam + ReduceLROnPlateau; Lightning's
    checkpoint of `optimizer_states`):
    * `param_groups[0]["lr"]` is a 0-dim fp64 tensor ON THE DEVICE which the kernel reads (never a by-value argument): schedulers `fill_` it, a plain
      `group["lr"] = 1e-4` is routed into it — a captured HIP graph follows both between replays;
    * `state_dict()` / `load_state_dict()` speak torch.optim.Adam's format (per-parameter `step`, `exp_avg`, `exp_avg_sq`; parameters numbered in
      the order of `flat.params`), so the moments and the step count resume from a checkpoint written by either optimiser;
    * every `step()` verifies on the host that each Parameter still aliases its slice of the flat buffer (`p.data = ...`, `module.to(dtype)`,
      `load_state_dict(assign=True)` break that silently: the kernel would update a buffer the model no longer reads) and re-adopts the parameters'
      current values when it does not; with `module=` it also verifies that the module still holds the very Parameter objects (assign=True
      replaces them — those cannot be re-adopted, the gradient hooks sit on the old ones: RuntimeError).  A replayed graph does not run this
      check: call `check_aliasing()` where parameters may have been replaced (after loading a checkpoint)."""

    def __init__(self, flat: "FlatGradients", lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, module=None):
        if not flat.flat.is_cuda:
            raise RuntimeError("FlatAdam runs on the GPU only")
        if not 0.0 <= float(lr) or not 0.0 <= float(eps) or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= float(weight_decay):
            raise ValueError("FlatAdam: invalid hyper-parameter")          # torch.optim.Adam's own checks
        self.flat = flat
        self._module = module
        dev = flat.flat.device
        n = flat.flat.numel()
        self._lr_dev = torch.full((), float(lr), dtype=torch.float64, device=dev)
        super().__init__(flat.params, dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=float(weight_decay),
                                           amsgrad=False, maximize=False, foreach=None, capturable=True, differentiable=False, fused=None))
        g = _LiveGroup(self, self.param_groups[0])
        dict.__setitem__(g, "lr", self._lr_dev)
        self.param_groups = [g]
        self.params = torch.empty(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self._order = None
        self._offs: dict = {}
        self._adopt()

    # hyper-parameters as attributes (the round-5 interface)
    lr = property(lambda self: float(self._lr_dev))
    betas = property(lambda self: self.param_groups[0]["betas"])
    eps = property(lambda self: self.param_groups[0]["eps"])
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"])

    def add_param_group(self, param_group):
        if getattr(self, "flat", None) is not None and self.param_groups:
            raise RuntimeError("FlatAdam has exactly one parameter group: the parameters of its FlatGradients")
        super().add_param_group(param_group)

    def _adopt(self, force: bool = False):
        """(Re-)lay the parameters and moments out in the gradient buffer's current order (it changes once, when the overlapped all-reduce
        calibrates its buckets) and point every Parameter at its slice.  `force`: the order is unchanged but a Parameter no longer aliases its
        slice — its current value is taken over (the model's tensors are the truth) and it is re-pointed."""
        order = list(self.flat._order_idx)
        if order == self._order and not force:
            return
        old = None if self._order is None or order == self._order else (self._order, self.exp_avg.clone(), self.exp_avg_sq.clone())
        new_p = torch.empty_like(self.params)
        off, offs = 0, {}
        with torch.no_grad():
            for i in order:
                p = self.flat.params[i]
                if p.dtype != torch.float32 or p.device != self.params.device:
                    raise RuntimeError(f"FlatAdam: parameter {i} is now {p.dtype} on {p.device}; the flat buffers are fp32 on {self.params.device} "
                                       "(module.to(dtype / device) after the optimiser was built is not supported)")
                offs[i] = off
                new_p[off:off + p.numel()].copy_(p.detach().reshape(-1))
                off += p.numel()
            if old is not None:                                   # carry the moments over to the new layout
                o_off, oo = {}, 0
                for i in old[0]:
                    o_off[i] = oo; oo += self.flat.params[i].numel()
                for i in order:
                    k = self.flat.params[i].numel()
                    self.exp_avg[offs[i]:offs[i] + k].copy_(old[1][o_off[i]:o_off[i] + k])
                    self.exp_avg_sq[offs[i]:offs[i] + k].copy_(old[2][o_off[i]:o_off[i] + k])
            if self._order is None or old is not None:
                self.params = new_p
            else:
                self.params.copy_(new_p)                          # same layout: keep the buffer's address (a captured graph holds it)
            for i in order:
                p = self.flat.params[i]
                p.data = self.params[offs[i]:offs[i] + p.numel()].view_as(p)
        self._order, self._offs = order, offs

    def check_aliasing(self, repair: bool = True) -> bool:
        """True when every Parameter still aliases its slice of the flat parameter buffer.  Otherwise: with `repair` the parameters' current
        values are copied into the buffer and the Parameters re-pointed (returns False once, True afterwards); without it RuntimeError."""
        if self._module is not None:
            seen, cur = set(), []
            for p in self._module.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p)); cur.append(p)
            if len(cur) != len(self.flat.params) or any(a is not b for a, b in zip(cur, self.flat.params)):
                raise RuntimeError("FlatAdam: the module no longer holds the Parameter objects this optimiser (and its FlatGradients) were built on "
                                   "(load_state_dict(assign=True) / parameter re-registration): rebuild FlatGradients and FlatAdam, or load "
                                   "with assign=False, which copies into the aliased tensors")
        base = self.params.data_ptr()
        ok = all(self.flat.params[i].data_ptr() == base + 4 * off and self.flat.params[i].dtype == torch.float32 for i, off in self._offs.items())
        if ok:
            return True
        if not repair:
            raise RuntimeError("FlatAdam: a Parameter no longer aliases the flat parameter buffer (p.data was replaced): the step would update a "
                               "buffer the model does not read")
        self._adopt(force=True)
        return False

    @torch.no_grad()
    def step(self, closure=None):
        from ._lib import call, ptr, stream_ptr
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._adopt()
        self.check_aliasing(repair=True)
        g = self.param_groups[0]
        if g["lr"] is not self._lr_dev:                              # someone replaced the entry around _LiveGroup (dict.update on a copy, ...)
            self._lr_dev.fill_(float(g["lr"]))
            dict.__setitem__(g, "lr", self._lr_dev)
        call("conan_adam_flat_step", ptr(self.params), ptr(self.flat.flat), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.step_dev),
             ptr(self._ticket), self.params.numel(), 0.0, g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], ptr(self._lr_dev), stream_ptr())
        return loss

    def zero_grad(self, set_to_none: bool = True):
        self.flat.zero()

    # ------------------------------------------------------------------------------------------------ checkpoint (torch.optim.Adam's format)
    def state_dict(self):
        self._adopt()
        state = {}
        steps = float(self.step_dev)
        if steps > 0:                                               # (torch's Adam has no state before its first step)
            for i, p in enumerate(self.flat.params):
                off, k = self._offs[i], p.numel()
                # `step` as torch's default (non-capturable) Adam keeps it: a CPU fp32 scalar — the checkpoint then loads into either optimiser
                state[i] = {"step": torch.tensor(steps, dtype=torch.float32), "exp_avg": self.exp_avg[off:off + k].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + k].view_as(p).clone()}
        grp = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        grp["lr"] = float(self._lr_dev)
        grp["capturable"] = False                                   # (a statement about torch's kernels, for a torch.optim.Adam that loads this)
        grp["params"] = list(range(len(self.flat.params)))
        return {"state": state, "param_groups": [grp]}

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.flat.params):
            raise ValueError("FlatAdam.load_state_dict: expected one parameter group of %d parameters" % len(self.flat.params))
        if groups[0].get("amsgrad", False) or groups[0].get("maximize", False):
            raise ValueError("FlatAdam.load_state_dict: amsgrad / maximize are not implemented")
        self._adopt()
        g = self.param_groups[0]
        for k in ("betas", "eps", "weight_decay"):
            if k in groups[0]:
                v = groups[0][k]
                dict.__setitem__(g, k, (float(v[0]), float(v[1])) if k == "betas" else float(v))
        self._lr_dev.fill_(float(groups[0]["lr"]))
        dict.__setitem__(g, "lr", self._lr_dev)
        ids = list(groups[0]["params"])
        state = state_dict["state"]
        self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.step_dev.zero_()
        steps = set()
        for pos, key in enumerate(ids):
            st = state.get(key, state.get(str(key)))
            if not st:
                continue
            p = self.flat.params[pos]
            off, k = self._offs[pos], p.numel()
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"FlatAdam.load_state_dict: moment of parameter {pos} has shape {tuple(st['exp_avg'].shape)}, expected {tuple(p.shape)}")
            self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1).to(self.exp_avg))
            self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1).to(self.exp_avg_sq))
            steps.add(float(st["step"]))
        if len(steps) > 1:
            raise ValueError(f"FlatAdam.load_state_dict: the parameters carry different step counts {sorted(steps)}; one counter serves them all")
        if steps:
            self.step_dev.fill_(steps.pop())
