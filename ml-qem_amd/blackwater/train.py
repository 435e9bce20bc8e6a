"""Training loop of the path: same semantics as the reference's ``train_gnn``
(docs/tutorials/__ml_models.py:100-187 == docs/tutorials/gnn.py:282-378): MSE loss on ``squeeze(y, 1)``,
Adam(lr=1e-3), ReduceLROnPlateau('min', factor 0.1, patience 15, min_lr 1e-5) stepped on the SUMMED validation
loss, batch 32 by default -- with three MI355X-side changes that do not alter the maths:

* the dataset is device-resident (:class:`GraphArena`) and batches are assembled on the GPU,
* the loss is accumulated on the device (the reference calls ``loss.item()`` every step: a host sync),
* under ``torch.distributed`` each rank trains on its shard and gradients are averaged with ONE all-reduce
  over a flat fp32 buffer (RCCL over xGMI on MI355X, gloo in the CPU tests).
"""
from __future__ import annotations

import math
import os
import pickle
import time
from typing import Callable, Iterable, List, Optional

import numpy as np
import torch
from torch import nn

from .native import ops as _ops


def flatten_parameters(model: nn.Module):
    """Re-homes every parameter as a view into one contiguous fp32 buffer and gives each a ``.grad`` view into a
    second one.  The optimizer then steps a single tensor and data-parallel training reduces a single buffer
    (13,645 ... 103,465 floats for the reference's models).  Returns (flat parameter, flat gradient buffer)."""
    params = [p for p in model.parameters()]
    total = sum(p.numel() for p in params)
    dev = params[0].device
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            p.grad = flat_grad[off:off + n].view(p.shape)
            off += n
    flat_param = nn.Parameter(flat)
    flat_param.grad = flat_grad
    return flat_param, flat_grad


class FlatAdam(torch.optim.Optimizer):
    """``torch.optim.Adam(lr, betas, eps)`` (no amsgrad, no weight decay: what the reference trains with,
    docs/tutorials/__ml_models.py:127) on ONE flat CUDA buffer, as one launch of ``mlqem_adam_step_f32``: the step count and the
    learning rate are device tensors (``state['step']``, ``param_groups[0]['lr']``), so the update is capturable in a
    hipGraph and ``ReduceLROnPlateau`` changes the rate in place.  torch's fused multi-tensor kernel gives a buffer of this
    path's size (1.8 k-180 k floats) to one workgroup: 35 us per step where this takes 3-4."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        params = list(params)
        if len(params) != 1 or not params[0].is_cuda or params[0].dtype != torch.float32 or not params[0].is_contiguous():
            raise ValueError("FlatAdam steps one contiguous fp32 CUDA buffer (train.flatten_parameters)")
        dev = params[0].device
        lr_t = lr.to(device=dev, dtype=torch.float32).clone() if torch.is_tensor(lr) else torch.tensor(float(lr), dtype=torch.float32, device=dev)
        super().__init__(params, dict(lr=lr_t, betas=tuple(betas), eps=float(eps)))
        p = params[0]
        self.state[p] = {"step": torch.zeros((), dtype=torch.float32, device=dev), "exp_avg": torch.zeros_like(p),
                         "exp_avg_sq": torch.zeros_like(p)}

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Copies a saved state IN PLACE into this optimizer's device tensors.  ``Optimizer.load_state_dict`` would leave a
        ``step`` loaded with ``map_location='cpu'`` on the host (this class declares neither ``capturable`` nor ``fused``) and
        would replace the learning-rate tensor by a deep copy -- captured graphs keep reading the old pointers, so a resumed
        run would step with stale moments and never see ``ReduceLROnPlateau``.  Hyper-parameters (betas, eps) are taken over."""
        groups = state_dict["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != 1:
            raise ValueError("FlatAdam state has one group of one flat parameter")
        saved = state_dict["state"].get(groups[0]["params"][0], None)
        group = self.param_groups[0]
        p = group["params"][0]
        st = self.state[p]
        if saved is not None:
            for key in ("exp_avg", "exp_avg_sq"):
                src = saved[key]
                if tuple(src.shape) != tuple(st[key].shape):
                    raise ValueError(f"FlatAdam.load_state_dict: {key} has {tuple(src.shape)}, this optimizer {tuple(st[key].shape)}")
                st[key].copy_(src.to(device=p.device, dtype=torch.float32))
            step = saved["step"]
            st["step"].fill_(float(step.item() if torch.is_tensor(step) else step))
        lr = groups[0]["lr"]
        group["lr"].fill_(float(lr.item() if torch.is_tensor(lr) else lr))
        for key, val in groups[0].items():
            if key not in ("params", "lr"):
                group[key] = tuple(val) if key == "betas" else val

    @torch.no_grad()
    def step(self, closure=None):
        from .native import ops

        group = self.param_groups[0]
        p = group["params"][0]
        st = self.state[p]
        if p.grad is None:
            return None
        ops.adam_step(p, p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"], group["lr"], group["betas"][0], group["betas"][1], group["eps"],
                      bump=getattr(self, "bump_counter", None))
        return None


class DataParallelShard:
    """Node-count-balanced shards of the graph ids, one per rank and all of one length, so that every rank runs the same
    number of steps per epoch (graphs on this path vary 13x in size; a rank with more nodes per step would stall the
    gradient all-reduce of all the others)."""

    @staticmethod
    def split(node_counts: np.ndarray, world_size: int) -> List[np.ndarray]:
        order = np.argsort(-node_counts, kind="stable")
        loads = np.zeros(world_size)
        buckets: List[List[int]] = [[] for _ in range(world_size)]
        for g in order:  # longest-processing-time greedy, then keep every shard the same length
            r = int(np.argmin(loads))
            buckets[r].append(int(g))
            loads[r] += node_counts[g]
        n = min(len(b) for b in buckets)
        return [np.sort(np.asarray(b[:n], dtype=np.int64)) for b in buckets]


class Trainer:
    def __init__(self, model: nn.Module, lr: float = 1e-3, distributed: bool = False, flat: bool = True,
                 capturable: bool = False):
        self.model = model
        self.distributed = distributed and torch.distributed.is_available() and torch.distributed.is_initialized()
        self.world = torch.distributed.get_world_size() if self.distributed else 1
        if flat:
            self.flat_param, self.flat_grad = flatten_parameters(model)
            opt_params = [self.flat_param]
            # gradient slots of the flat buffer, one view per parameter, in parameter order (see step())
            self._params = [p for p in model.parameters()]
            self._grad_slots, off = [], 0
            for p in self._params:
                self._grad_slots.append(self.flat_grad[off:off + p.numel()].view(p.shape))
                off += p.numel()
        else:
            self.flat_param = self.flat_grad = None
            opt_params = list(model.parameters())
        fused = opt_params[0].is_cuda
        if fused:
            from .native import ops

            ops.prepare_device(opt_params[0].device)      # per-device launch state exists before anything is captured
        if flat and fused:
            # one native launch over the flat buffer, step count and learning rate on the device (capturable either way)
            self.optimizer = FlatAdam(opt_params, lr=lr)
        elif capturable:   # step count and learning rate live on the device: the update can sit inside a captured hipGraph
            self.optimizer = torch.optim.Adam(opt_params, lr=torch.tensor(lr, dtype=torch.float32, device=opt_params[0].device),
                                              fused=True, capturable=True)
        else:
            self.optimizer = torch.optim.Adam(opt_params, lr=lr, fused=fused) if fused else torch.optim.Adam(opt_params, lr=lr)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, "min", factor=0.1, patience=15,
                                                                    min_lr=1e-5)
        self.criterion = nn.MSELoss()
        self._fused_loss = True                # optim.hip's one-launch MSE loss + gradient (a subclass with another criterion clears it)
        # The host enqueues a step several times faster than the device runs it.  Unbounded run-ahead makes torch's caching
        # allocator grow without end on the multi-stream path: blocks handed between the branch streams can only be reused
        # after their recorded events have completed, so every step queued ahead needs its own copy of the activations and
        # each growth is a hipMalloc that drains the queue (seen as 1536-circuit steps of 16-51 ms instead of 11.7).  A few steps
        # in flight keep the device busy and the pool at its steady size.
        self.max_steps_in_flight = int(os.environ.get("MLQEM_MAX_STEPS_IN_FLIGHT", "4"))   # 0 = unbounded; 4 measured as fast as unbounded (129-131 k circuits/s either way)
        self._inflight = []
        self._prescaled = False
        self.history = {"train_losses": [], "val_losses": []}
        if self.distributed:
            self.broadcast_parameters()

    def _capture_stream(self):
        """ONE side stream per trainer for the eager warm-up iterations AND every capture: the launch wrappers keep their
        per-stream state (workspaces, ticket slots) by stream handle, so what the warm-up allocated eagerly is what the
        captures use -- nothing is allocated, zeroed or first-touched inside a capture."""
        if getattr(self, "_cap_stream", None) is None:
            self._cap_stream = torch.cuda.Stream()
        return self._cap_stream

    def broadcast_parameters(self, src: int = 0):
        """Every replica starts from rank ``src``'s parameters and buffers (what DistributedDataParallel does at
        construction): a rank seeded differently, or restored from another checkpoint, cannot silently average gradients
        over diverging replicas."""
        with torch.no_grad():
            if self.flat_param is not None:
                torch.distributed.broadcast(self.flat_param.data, src=src)
            else:
                for p in self.model.parameters():
                    torch.distributed.broadcast(p.data, src=src)
            for b in self.model.buffers():
                torch.distributed.broadcast(b.data, src=src)

    def _agreed_min(self, n: int) -> int:
        """min over ranks of a host integer (1 collective); ``n`` itself outside torch.distributed."""
        if not self.distributed:
            return int(n)
        dev = self.flat_param.device if self.flat_param is not None else next(self.model.parameters()).device
        t = torch.tensor([int(n)], dtype=torch.int64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
        return int(t.item())

    # -- one optimisation step on an assembled batch -------------------------------------------------------
    def step(self, batch) -> torch.Tensor:
        """forward -> MSE -> backward -> (all-reduce) -> Adam.  Returns the loss as a device tensor (the host waits only for
        the step ``max_steps_in_flight`` steps back, see __init__)."""
        throttle = self.max_steps_in_flight and self.flat_grad is not None and self.flat_grad.is_cuda \
            and not torch.cuda.is_current_stream_capturing()
        if throttle and len(self._inflight) >= self.max_steps_in_flight:
            self._inflight.pop(0).synchronize()
        loss = self._forward_backward(batch)
        if self.distributed:
            self.all_reduce_gradients()
        self.optimizer.step()
        if throttle:
            ev = torch.cuda.Event()
            ev.record()
            self._inflight.append(ev)
        return loss

    def _step_counter(self, dev) -> torch.Tensor:
        """The device-resident step counter the dropout keys are derived from.  Every step sees the next value.  With FlatAdam the
        optimizer's launch increments it when the step is done (``mlqem_adam_step_f32`` ``bump_counter``: no launch of its own, which
        is 1.5-2.5 % of a launch-bound step); otherwise ``_tick`` does, before the step.  Either way the steps see 1, 2, 3, ..."""
        self._bumped_by_optimizer = isinstance(self.optimizer, FlatAdam)
        counter = torch.full((1,), 1 if self._bumped_by_optimizer else 0, dtype=torch.int64, device=dev)
        if self._bumped_by_optimizer:
            self.optimizer.bump_counter = counter
        return counter

    def _tick(self):
        if not self._bumped_by_optimizer:
            self.counter.add_(1)

    def _forward_backward(self, batch) -> torch.Tensor:
        """The local half of a step: forward, loss, backward, gradients filed into the flat buffer.  Returns the loss."""
        self.model.train()
        if self.flat_grad is not None:
            # With .grad pointing into the flat buffer autograd would ADD every parameter's gradient into its slot: one
            # tiny kernel per parameter (~30 per step here).  Cleared .grad fields make it hand the gradient tensors
            # over as they are; one multi-tensor copy then files them into the flat buffer.
            for p in self._params:
                p.grad = None
        else:
            self.optimizer.zero_grad(set_to_none=False)
        out = self.model(*batch.model_args())
        target = batch.y if batch.y.dim() == 2 else torch.squeeze(batch.y, 1)
        real = getattr(batch, "num_real", None)
        padded = real is not None and real < out.shape[0]   # a bucket-padded batch: the last row is the filler graph's
        rows = real if padded else out.shape[0]
        if (self._fused_loss and type(self.criterion) is nn.MSELoss and self.criterion.reduction == "mean" and out.is_cuda and out.dtype == torch.float32
                and out.dim() == 2 and target.dim() == 2 and target.shape[1] == out.shape[1] and target.shape[0] >= rows
                and target.dtype == torch.float32 and rows > 0 and out.stride(1) == 1 and target.stride(1) == 1):
            # loss and d loss / d out from one launch; the backward starts from that gradient (no ones fill, no mse kernels, and
            # for a padded batch no slice: the filler rows' gradient is written as zero by the same launch)
            from .native import ops

            loss, g_out = ops.mse_loss_grad(out.detach(), target, rows=rows)
            out.backward(g_out)
        else:
            if padded:
                out, target = out[:real], target[:real]
            loss = self.criterion(out, target)
            loss.backward()
        if self.flat_grad is not None:
            grads = [p.grad for p in self._params]
            if any(g is None for g in grads):   # a parameter the loss does not reach: its slot must read zero
                self.flat_grad.fill_(0.0)     # a fill kernel, not a memset node: the step may be replayed from a hipGraph (csrc/asap.hip)
                pairs = [(d, g) for d, g in zip(self._grad_slots, grads) if g is not None]
                if pairs:
                    torch._foreach_copy_([d for d, _ in pairs], [g for _, g in pairs])
            else:
                torch._foreach_copy_(self._grad_slots, grads)
            if self.world > 1:
                # the data-parallel mean: scale BEFORE the all-reduce, one kernel over the flat buffer that is part of the
                # captured half of a replayed step (a div_ behind the collective was one more eager launch per step)
                self.flat_grad.mul_(1.0 / self.world)
                self._prescaled = True
            for p, slot in zip(self._params, self._grad_slots):
                p.grad = slot                   # what callers (and the reference's loop shape) expect to find
        return loss.detach()

    def all_reduce_gradients(self):
        """Averages the gradients over the ranks: ONE collective over the flat buffer.  The 1/world factor is applied where
        the gradients are filed into the flat buffer (``_forward_backward``: inside the captured half of a replayed step), so
        the collective is a plain SUM and nothing runs behind it but Adam."""
        if self.flat_grad is not None:
            if not self._prescaled:       # gradients that did not come through _forward_backward (a caller's own backward)
                self.flat_grad.mul_(1.0 / self.world)
            torch.distributed.all_reduce(self.flat_grad, op=torch.distributed.ReduceOp.SUM)
            self._prescaled = False
        else:
            for p in self.model.parameters():
                p.grad.mul_(1.0 / self.world)
                torch.distributed.all_reduce(p.grad, op=torch.distributed.ReduceOp.SUM)

    @torch.no_grad()
    def evaluate(self, batches: Iterable) -> torch.Tensor:
        """Sum of per-batch MSE losses over ``batches`` (what the reference feeds the LR scheduler)."""
        self.model.eval()
        total = None
        for batch in batches:
            out = self.model(*batch.model_args())
            target = batch.y if batch.y.dim() == 2 else torch.squeeze(batch.y, 1)
            l = self.criterion(out, target)
            total = l if total is None else total + l
        return total

    def predict(self, arena, ids, batch_size: int = 256) -> torch.Tensor:
        """Model outputs (eval mode, no gradients) for the graphs ``ids`` of ``arena``, in that order -- the input of
        ``metrics.mitigation_report`` (the reference's post-training evaluation loop, __ml_models.py:207-230)."""
        was_training = self.model.training
        self.model.eval()
        outs = []
        with torch.no_grad():
            for i in range(0, len(ids), batch_size):
                outs.append(self.model(*arena.batch(ids[i:i + batch_size]).model_args()))
        self.model.train(was_training)
        return torch.cat(outs, dim=0)

    def save(self, path: str, history: Optional[dict] = None) -> str:
        """Writes what the reference's training cell writes (docs/tutorials/__ml_models.py:196-205): ``<path>.pth`` =
        ``torch.save(model.state_dict())`` -- plain CPU tensors under the reference's parameter names, loadable with
        ``strict=True`` by the reference's (and the oracle's) modules -- and ``<path>.pk`` = the pickled
        ``{'train_losses': [...], 'val_losses': [...]}`` of the last ``fit`` (or ``history``).  ``path`` may end in ``.pth``.
        Under data parallelism every rank holds the same parameters; rank 0 writes."""
        stem = path[:-4] if path.endswith(".pth") else path
        if self.distributed and torch.distributed.get_rank() != 0:
            return stem + ".pth"
        state = {k: v.detach().to("cpu", copy=True).contiguous() for k, v in self.model.state_dict().items()}
        os.makedirs(os.path.dirname(os.path.abspath(stem)), exist_ok=True)
        torch.save(state, stem + ".pth")
        hist = history if history is not None else self.history
        to_save = {"train_losses": [float(v) for v in hist["train_losses"]], "val_losses": [float(v) for v in hist["val_losses"]]}
        with open(stem + ".pk", "wb") as handle:
            pickle.dump(to_save, handle, protocol=pickle.HIGHEST_PROTOCOL)
        return stem + ".pth"

    def load(self, path: str) -> dict:
        """Restores ``save``'s files: parameters and buffers ``strict=True`` (into the flat buffer the optimizer steps), and
        returns the loss curves (empty lists when there is no ``.pk`` next to the checkpoint)."""
        stem = path[:-4] if path.endswith(".pth") else path
        state = torch.load(stem + ".pth", map_location="cpu", weights_only=True)
        missing = self.model.load_state_dict(state, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        if os.path.exists(stem + ".pk"):
            with open(stem + ".pk", "rb") as handle:
                self.history = pickle.load(handle)
        return self.history

    def _fit_step(self, arena, ids) -> torch.Tensor:
        """One training step of ``fit`` on the graphs ``ids`` of ``arena`` (BucketedTrainer replays a captured step here)."""
        return self.step(arena.batch(ids))

    def fit(self, arena, train_ids, val_ids, epochs: int, batch_size: int = 32, seed: int = 0, log: Callable = None):
        """Epoch loop with per-epoch shuffling (seed + epoch) and the reference's LR schedule.

        Under data parallelism every step ends in a gradient all-reduce, so all ranks must run the SAME number of steps:
        the per-epoch step count is agreed once (min over ranks of this rank's batch count); a rank whose shard holds a
        few graphs more -- round-robin shards differ by one -- leaves its surplus out of that epoch (a different surplus
        each epoch, the permutation is reseeded)."""
        history = self.history = {"train_losses": [], "val_losses": []}
        train_ids = np.asarray(train_ids)
        steps_per_epoch = self._agreed_min(-(-len(train_ids) // batch_size))
        for epoch in range(epochs):
            rng = np.random.RandomState(seed + epoch)
            order = rng.permutation(train_ids)
            running, n_batches = None, 0
            for i in range(0, steps_per_epoch * batch_size, batch_size):
                loss = self._fit_step(arena, order[i:i + batch_size])
                running = loss.clone() if running is None else running + loss     # (a captured step's loss tensor is rewritten by the next replay)
                n_batches += 1
            val_batches = [arena.batch(val_ids[i:i + batch_size]) for i in range(0, len(val_ids), batch_size)]
            val_total = self.evaluate(val_batches)
            if self.distributed:
                torch.distributed.all_reduce(val_total, op=torch.distributed.ReduceOp.SUM)
            self.scheduler.step(val_total.item())  # one sync per epoch
            _ops.check_overflow_flags()            # the epoch's capacity-bound launches (ASAPooling's list coarsening): raise, never truncate
            if epoch >= 1:  # the reference drops epoch 0 from its curves (__ml_models.py:182)
                history["train_losses"].append(running.item() / n_batches)
                history["val_losses"].append(val_total.item() / max(len(val_batches), 1) / self.world)
            if log:
                log(epoch, history)
        return history


class RowsTrainer(Trainer):
    """Training of a feature-matrix model (MLP1 / MLP2 / MLP3: docs/tutorials/mlp.py, trained in h10_mlp.ipynb cells [10]-[13]
    on fixed-size batches of ``encode_data`` rows) with the WHOLE step -- forward, MSE, backward, gradient filing, Adam --
    captured once per batch shape in a hipGraph and replayed: the host copies the batch into the graph's input buffers and
    launches one graph.  An MLP1 step on 262 144 rows is ~0.15-0.35 ms of device work in 6 kernels, and ~0.3 ms of Python when
    enqueued launch by launch.  Same mechanics as :class:`BucketedTrainer`: Adam with device-resident step count and learning
    rate, dropout masks keyed by a device-resident counter bumped inside the graph, three eager warm-up iterations that leave
    no trace.  ``graphs=False`` runs the identical step eagerly (bit-identical losses)."""

    def __init__(self, model: nn.Module, lr: float = 1e-3, graphs: bool = True, distributed: bool = False):
        super().__init__(model, lr=lr, distributed=distributed, flat=True, capturable=True)
        from .native import ops

        self.graphs = graphs
        self.fused_mlp1_step = True        # MLP1: image -> forward -> backward -> second stage -> Adam, nothing of autograd (False: the autograd path; tests compare the two)
        self.counter = self._step_counter(self.flat_param.device)
        ops.set_seed_counter(self.counter)
        model.static_dropout_key = True
        self._entries = {}
        self._warm = False

    class _Rows:
        def __init__(self, x, y):
            self.x, self.y = x, y

        def model_args(self):
            return (self.x,)

    def _step_on(self, batch):
        if self._mlp1_fused(batch):
            return self._step_mlp1(batch)
        self._tick()
        loss = self._forward_backward(batch)
        if self.distributed:
            self.all_reduce_gradients()
        self.optimizer.step()
        return loss

    def _mlp1_fused(self, batch) -> bool:
        """MLP1 with the MSE loss on one rank: the whole step is five launches with nothing of autograd in between."""
        from .native import functional as F
        from .nn.mlp import MLP1

        m = self.model
        return (type(m) is MLP1 and self._fused_loss and not self.distributed and self.flat_grad is not None
                and type(self.criterion) is nn.MSELoss and self.criterion.reduction == "mean"
                and self.fused_mlp1_step
                and torch.is_tensor(batch.x) and batch.y.dim() == 2 and batch.y.dtype == torch.float32 and batch.y.shape[0] == batch.x.shape[0]
                and batch.x.shape[0] > 0 and F.mlp1_fused_ok(batch.x, m.fc1.weight, m.fc2.weight))

    def _step_mlp1(self, batch):
        """image -> forward (+ loss sums, + d loss / d out) -> backward -> second stage (gradients straight into the flat buffer's
        slots, the loss) -> Adam: what ``loss = MSELoss()(model(x), y); loss.backward(); optimizer.step()`` computes for
        MLP1 (docs/tutorials/__ml_models.py:148-160, mlp.py:18-30), same gradients bit for bit as the autograd path."""
        from .native import ops

        m = self.model
        if getattr(self, "_mlp1_slots", None) is None:
            by_id = {id(p): s for p, s in zip(self._params, self._grad_slots)}
            self._mlp1_slots = tuple(by_id[id(p)] for p in (m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias))
        bf16 = getattr(m, "mfma", "f32") == "bf16"
        i, h = m.fc1.weight.shape[1], m.fc1.weight.shape[0]
        _, hs, xp, gout = ops.mlp1_forward(batch.x, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, bf16=bf16, stash=True, target=batch.y)
        *_, loss = ops.mlp1_backward(gout, xp, hs, m.fc2.weight, i, h, bf16=bf16, dst=self._mlp1_slots, want_loss=True)
        for p, slot in zip(self._params, self._grad_slots):
            p.grad = slot
        self.optimizer.step()
        return loss

    def step_rows(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """One optimisation step on the rows ``x`` [N, F] with targets ``y``; returns the loss (in graph mode a device
        tensor the next step of the same shape overwrites: read or clone it before stepping again)."""
        from .native import ops

        if not self.graphs or self.distributed:      # under data parallelism the collective sits inside the step: eager
            # the rows in the padded layout the captured step reads them in: the same kernels, the same sums, the same bits
            return self._step_on(self._Rows(ops._mlp1_x(x) if x.dtype == torch.float32 else x, y))

        key = (tuple(x.shape), tuple(y.shape))
        entry = self._entries.get(key)
        if entry is None:
            xs = ops.padded_empty(x.shape[0], x.shape[1], x.device)
            ys = torch.empty_like(y)
            xs.copy_(x)
            ys.copy_(y)
            batch = self._Rows(xs, ys)
            if not self._warm:
                dev = self.flat_param.device
                keep = (self.flat_param.detach().clone(), self.counter.clone(), torch.cuda.get_rng_state(dev))
                opt_keep = {id(st): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for st in self.optimizer.state.values()}
                buf_keep = [b.detach().clone() for b in self.model.buffers()]
                side = self._capture_stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        self._step_on(batch)
                torch.cuda.current_stream().wait_stream(side)
                with torch.no_grad():
                    self.flat_param.copy_(keep[0])
                    self.counter.copy_(keep[1])
                    for st in self.optimizer.state.values():
                        saved = opt_keep.get(id(st))
                        for k, v in st.items():
                            if torch.is_tensor(v):
                                if saved is not None and torch.is_tensor(saved.get(k)):
                                    v.copy_(saved[k])
                                else:
                                    v.zero_()
                    for b, kept in zip(self.model.buffers(), buf_keep):
                        b.copy_(kept)
                torch.cuda.synchronize()
                torch.cuda.set_rng_state(keep[2], dev)
                self._warm = True
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=self._capture_stream()):
                loss = self._step_on(batch)
            entry = self._entries[key] = {"graph": graph, "x": xs, "y": ys, "loss": loss}
        else:
            if x.data_ptr() != entry["x"].data_ptr():
                entry["x"].copy_(x)
            if y.data_ptr() != entry["y"].data_ptr():
                entry["y"].copy_(y)
        entry["graph"].replay()
        return entry["loss"]

    def input_buffers(self, x_shape, y_shape):
        """The graph's own input buffers for a batch shape (after its first step): a caller that assembles its batch directly
        in them (an index-select with ``out=``) saves the copy."""
        e = self._entries.get((tuple(x_shape), tuple(y_shape)))
        return (e["x"], e["y"]) if e else None


class StratifiedBatches:
    """Batches that all hold the same number of graphs of every SIZE (node count, edge count): a size-stratified sampler.

    The graphs of a circuit corpus come in a few sizes (a Trotter circuit's graph depends on its step count, not on its
    couplings), and a batch drawn uniformly has a different node total every time -- different launch shapes, a different
    allocation pattern, nothing to capture.  Here every batch takes ``quota[c]`` graphs of size class c -- the quotas are
    the classes' shares of ``batch``, remainders handed to the classes nearest the mean size so that the batch total stays
    near batch x mean -- walking each class in its own seeded epoch permutation.  All batches then have identical node and
    edge totals: one bucket for :class:`BucketedTrainer`, no padding."""

    def __init__(self, node_counts, edge_counts, batch: int, seed: int = 0):
        nodes, edges = np.asarray(node_counts, dtype=np.int64), np.asarray(edge_counts, dtype=np.int64)
        keys = nodes * (int(edges.max()) + 1 if len(edges) else 1) + edges
        uniq, inverse = np.unique(keys, return_inverse=True)
        self.classes = [np.flatnonzero(inverse == c) for c in range(len(uniq))]
        sizes = np.array([len(c) for c in self.classes], dtype=np.float64)
        share = sizes / sizes.sum() * batch
        quota = np.floor(share).astype(np.int64)
        class_nodes = np.array([nodes[c[0]] for c in self.classes], dtype=np.float64)
        mean = float((class_nodes * sizes).sum() / sizes.sum())
        spare = int(batch - quota.sum())
        # the remainder goes to classes with the largest fractional share, ties to the sizes nearest the mean
        order = sorted(range(len(uniq)), key=lambda c: (-(share[c] - quota[c]), abs(class_nodes[c] - mean)))
        for c in order[:spare]:
            quota[c] += 1
        if (quota > sizes).any():
            raise ValueError("StratifiedBatches: a size class holds fewer graphs than its share of one batch")
        self.quota, self.batch = quota, int(batch)
        self.rng = np.random.RandomState(seed)
        self._perm = [self.rng.permutation(c) for c in self.classes]
        self._pos = [0] * len(self.classes)
        self.nodes_per_batch = int((quota * class_nodes).sum())

    def draw(self) -> np.ndarray:
        out = []
        for c, q in enumerate(self.quota):
            if q == 0:
                continue
            if self._pos[c] + q > len(self._perm[c]):
                self._perm[c], self._pos[c] = self.rng.permutation(self.classes[c]), 0
            out.append(self._perm[c][self._pos[c]:self._pos[c] + q])
            self._pos[c] += q
        return np.concatenate(out)


def stable_padding(node_counts, node_quantum: int, n_fill: int):
    """Filler graphs that make a batch's node totals after EVERY ASAPooling level (ratio 0.5: k_g = ceil(n_g / 2), then
    ceil(k_g / 2) = ceil(n_g / 4)) functions of a bucket instead of the batch's size sequence.

    With N = sum n_g, sum ceil(n_g / 2) = (N + #odd) / 2 and sum ceil(n_g / 4) = (N + sum (-n_g mod 4)) / 4.  The batch is padded
    with exactly ``n_fill`` edgeless graphs (slices of the arena's filler) whose sizes bring the node total to a multiple of
    ``node_quantum``, the number of odd-sized graphs to a multiple of n_fill and sum (-n mod 4) to a multiple of 2 n_fill: the
    bucket is (n_pad, c1, c2) and K1 = (n_pad + c1) / 2, K2 = (n_pad + c2) / 4 whatever graphs were drawn.  A uniformly shuffled
    loader (the reference's: docs/tutorials/__ml_models.py:105, shuffle=True) then revisits a few dozen buckets, each captured once.
    Returns (n_pad, c1, c2, filler sizes [n_fill])."""
    n_of = np.asarray(node_counts, dtype=np.int64)
    f = int(n_fill)
    if f < 2 or f % 2 or node_quantum % 4:
        raise ValueError("stable_padding: an even number of fillers and a node quantum that is a multiple of 4")
    total, odd, pad4 = int(n_of.sum()), int((n_of & 1).sum()), int(((-n_of) % 4).sum())
    n_pad = -(-(total + 4 * f) // node_quantum) * node_quantum
    c1 = -(-odd // f) * f
    o = c1 - odd                                   # fillers of odd size: 0 .. f - 1
    c2 = -(-(pad4 + o) // (2 * f)) * (2 * f)
    m = (c2 - pad4 - o) // 2                       # 0 .. f - 1; (c2 - pad4 - o) is even: pad4 = odd (mod 2), c1 and c2 are even
    c_1 = min(o, m)                                # fillers of size 1 (mod 4), then 2, 3, 0: c_1 + c_3 = o, c_1 + c_2 = m
    c_2 = m - c_1
    c_3 = o - c_1
    c_0 = f - c_1 - c_2 - c_3
    assert c_0 >= 0 and (c2 - pad4 - o) % 2 == 0
    sizes = np.array([1] * c_1 + [2] * c_2 + [3] * c_3 + [4] * c_0, dtype=np.int64)
    extra = n_pad - total - int(sizes.sum())       # a non-negative multiple of 4 (n_pad >= total + 4 f)
    assert extra >= 0 and extra % 4 == 0
    units = extra // 4
    sizes += 4 * (units // f)
    sizes[: units % f] += 4
    return n_pad, c1, c2, sizes


class BucketedTrainer(Trainer):
    """The small-batch path (the reference's regime: batches of 32, docs/tutorials/__ml_models.py:105,148).

    At 32 four-qubit circuits a train step is ~8 k graph nodes: every kernel runs for microseconds and the step costs what
    the host needs to enqueue it (~2 ms of Python for ~100 launches).  Here the WHOLE step -- device batch assembly,
    forward, loss, backward, gradient filing, Adam -- is captured once per SIZE BUCKET in a hipGraph (torch.cuda.CUDAGraph)
    and replayed with one launch:

    * a batch is padded to its bucket's node count with a slice of the arena's edgeless filler graph (its output row is
      cut off before the loss) and the edge arrays are sized to the bucket's capacity, so every launch of the step has the
      same shapes for every selection of the bucket (``GraphArena.selection(ids, bucket)``);
    * what changes between replays lives in device memory: the packed selection (one small host->device copy per step) and
      the dropout step counter (``ops.set_seed_counter``; bumped inside the graph);
    * Adam runs with device-resident step / learning rate (``capturable=True``).

    ``graphs=False`` runs the very same bucketed step eagerly: the two modes launch identical kernels with identical
    arguments, so their loss trajectories agree bit for bit (tests/test_gpu_small_batch.py).

    The same trainer drives LARGE batches when their sizes repeat: with :class:`StratifiedBatches` every batch of the
    headline workload holds the same number of circuits of every size, so all of them fall into ONE bucket, nothing is
    padded (beyond the node quantum) and the step costs the host one selection upload and one replay -- it no longer
    matters whether the box's host enqueues a 70-launch step in 2 ms or in 6."""

    def __init__(self, model: nn.Module, arena, lr: float = 1e-3, graphs: bool = True, node_quantum: int = 1024,
                 edge_quantum: int = 2048, distributed: bool = False, split_update: bool = False, capture_collective: bool = False):
        super().__init__(model, lr=lr, distributed=distributed, flat=True, capturable=True)
        # Under data parallelism the gradient all-reduce sits between two captured halves -- (assembly, forward, backward)
        # and (Adam) -- and is enqueued eagerly: one collective call per step on the host instead of ~70 launches.
        # ``split_update`` forces the two-graph form without a process group (tests).
        # ``capture_collective``: try to capture the all-reduce INSIDE the step's graph (RCCL collectives are stream-ordered kernels
        # and can be captured; backend "nccl" only): the step is then ONE replay under data parallelism too.  A capture that raises
        # falls back to the two-graph form for the rest of the run; ``collective_in_graph`` says which form runs (None: not tried yet).
        self.split = bool(split_update) or self.distributed
        self.capture_collective = bool(capture_collective) and self.distributed and torch.distributed.get_backend() == "nccl"
        self.collective_in_graph, self.collective_capture_error = None, None
        if not arena.filler_nodes:
            raise ValueError("BucketedTrainer needs an arena built with filler_nodes > 0")
        if node_quantum > arena.filler_nodes:
            raise ValueError("node_quantum must not exceed the arena's filler_nodes")
        from .native import ops

        self.arena, self.graphs, self.nq, self.eq = arena, graphs, int(node_quantum), int(edge_quantum)
        dev = self.flat_param.device
        self.counter = self._step_counter(dev)
        ops.set_seed_counter(self.counter)
        model.static_dropout_key = True          # seeds = key(seed, rank) + device counter instead of a host call counter
        self._entries = {}
        self._pool = None
        self._update_graph = None
        self._warm = False
        # Models whose launch shapes follow every graph's size (Family B) replay a capture only for the same SEQUENCE of sizes.
        # With a size-stable sampler (StratifiedBatches) that is one pattern; with uniformly shuffled batches almost every
        # batch is a new one.  So such a pattern is run EAGERLY the first time it is seen (which also leaves its pooled graph
        # boundaries on the device: native/functional._device_ptr) and captured only when it comes back, and at most
        # ``max_pattern_captures`` patterns are ever captured (each holds a graph and a pinned ring): the rest stay eager.
        self.host_between_replays_s, self.host_between_replays_n = 0.0, 0
        self._pattern_model = bool(getattr(model, "needs_size_pattern", False))
        self.max_pattern_captures = int(os.environ.get("MLQEM_MAX_PATTERN_CAPTURES", "64"))
        self._seen = {}
        self.overflow_check_every = 256
        # SIZE-STABLE buckets for such models (default; MLQEM_STABLE_SHAPES=0: the size-pattern keys of rounds 3-5): every batch is
        # padded with a fixed number of small edgeless filler graphs sized so that the node totals after both poolings depend on
        # the bucket only (stable_padding), the pooled boundaries are computed on the device and the kernels that wanted the largest
        # graph get a bound -- a capture then serves every batch of its bucket, whatever the order or mix of graph sizes, and the
        # reference's shuffled loader replays captures instead of staying eager.  Needs every pooling at ratio 0.5 (the reference's).
        ratios = [float(getattr(m, "ratio")) for m in model.modules() if type(m).__name__ == "ASAPooling"]
        self.stable = (self._pattern_model and os.environ.get("MLQEM_STABLE_SHAPES", "1") != "0" and len(ratios) == 2
                       and all(r == 0.5 for r in ratios) and self.nq % 4 == 0)
        if self.stable:
            real_max = int(arena.node_counts[:len(arena)].max()) if len(arena) else 1
            self._nmax0 = max(real_max, 8 + 4 * (-(-self.nq // 16)))       # the largest filler (>= 4 of them share <= nq + 4 f nodes)
            self._kmax1, self._kmax2 = -(-self._nmax0 // 2), -(-(-(-self._nmax0 // 2)) // 2)
            # the structural capacity of the coarsened edge arrays matters to the list coarsening (large graphs) only
            self._cap_in_key = self._kmax1 > ops.asap_dense_max_k()

    def bucket_of(self, graph_ids):
        sel = np.asarray(graph_ids, dtype=np.int64)
        nb, eb = int(self.arena.node_counts[sel].sum()), int(self.arena.edge_counts[sel].sum())
        if self.stable:
            return self._stable_bucket(sel)[0]
        key = (-(-nb // self.nq) * self.nq, -(-max(eb, 1) // self.eq) * self.eq, len(sel))
        if getattr(self.model, "needs_size_pattern", False):
            # models whose launch shapes depend on every graph's size (Family B: ASAPooling keeps ceil(n_g / 2) clusters per
            # graph) replay a capture only for the same SEQUENCE of sizes -- what StratifiedBatches produces batch after batch --
            # and the same capacity of the coarsened edge arrays (a structural bound per graph: GraphArena.coarse_capacity)
            key += (self.arena.node_counts[sel].tobytes(), self.arena.coarse_capacity(sel))
        return key

    @staticmethod
    def fillers_for(batch: int) -> int:
        """Filler graphs of a size-stable batch: about half as many as circuits (even, 4 .. 64)."""
        return int(min(64, max(4, 2 * (-(-int(batch) // 4)))))

    def _stable_bucket(self, sel):
        """(bucket key, filler sizes, pooling plan, capacity) of a size-stable batch of the graphs ``sel``."""
        n_of, e_of = self.arena.node_counts[sel], self.arena.edge_counts[sel]
        f = self.fillers_for(len(sel))
        n_pad, c1, c2, fill = stable_padding(n_of, self.nq, f)
        e_pad = -(-max(int(e_of.sum()), 1) // self.eq) * self.eq
        cap = None
        if self._cap_in_key:
            cap = self.arena.coarse_capacity(np.concatenate([sel, np.full(f, len(self.arena), dtype=np.int64)]))
            if cap is not None:            # a bound: rounded up on a grid of 3 mantissa bits (<= 12.5 % over) so that batches share it
                e = max(int(cap).bit_length() - 4, 0)
                cap = -(-cap >> e) << e
        plan = (((n_pad + c1) // 2, self._nmax0, self._kmax1), ((n_pad + c2) // 4, self._kmax1, self._kmax2))
        return (n_pad, e_pad, len(sel) + f, "stable", c1, c2, cap), fill, plan, cap

    def _step_on(self, packed, b, n_pad, e_pad, sizes, num_real, cap=None, plan=None):
        loss = self._local_half(packed, b, n_pad, e_pad, sizes, num_real, cap, plan)
        if self.distributed:
            self.all_reduce_gradients()
        self.optimizer.step()
        return loss

    def _local_half(self, packed, b, n_pad, e_pad, sizes, num_real, cap=None, plan=None):
        self._tick()
        batch = self.arena.assemble(packed, b, n_pad, e_pad, None if plan else sizes, None, num_real, coarse_capacity=cap, pool_plan=plan)
        return self._forward_backward(batch)

    def _warm_up(self, ids, bucket):
        """torch asks for a few eager iterations on a side stream before the first capture (lazy initialisation of
        libraries, the autograd engine's threads).  They must not count as training: parameters, Adam state and the
        dropout counter are restored afterwards."""
        dev = self.flat_param.device
        keep = (self.flat_param.detach().clone(), self.counter.clone(), torch.cuda.get_rng_state(dev))
        # optimizer state (a checkpoint's, or what eager steps before the first capture left) and module buffers (BatchNorm
        # running statistics of an MLP2/3 head) are saved and put back: the three warm-up steps must leave no trace
        opt_keep = {id(st): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for st in self.optimizer.state.values()}
        buf_keep = [b.detach().clone() for b in self.model.buffers()]
        fill, plan, cap = None, None, bucket[4] if len(bucket) > 4 else None
        if self.stable:
            _, fill, plan, cap = self._stable_bucket(np.asarray(ids, dtype=np.int64))
        sel, nptr, eptr, nb, eb, real = self.arena.selection(ids, bucket[:2], filler_sizes=fill)
        packed = torch.from_numpy(np.concatenate([sel, nptr, eptr]).astype(np.int32)).to(self.flat_param.device)
        side = self._capture_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self._step_on(packed, len(sel), nb, eb, nptr[1:] - nptr[:-1], real, cap, plan)
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            self.flat_param.copy_(keep[0])
            self.counter.copy_(keep[1])
            for st in self.optimizer.state.values():
                saved = opt_keep.get(id(st))
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if saved is not None and torch.is_tensor(saved.get(k)):
                            v.copy_(saved[k])
                        else:
                            v.zero_()           # state the warm-up created: Adam's zero initial state
            for b, kept in zip(self.model.buffers(), buf_keep):
                b.copy_(kept)
        torch.cuda.synchronize()
        torch.cuda.set_rng_state(keep[2], dev)
        self._warm = True

    def _fit_step(self, arena, ids) -> torch.Tensor:
        # the reference's loop (shuffled epochs of 32, docs/tutorials/__ml_models.py:100-187) on this trainer's arena: every step is
        # the replay of a captured bucket -- size-stable buckets make that true for shuffled batches of Family B too
        if arena is self.arena:
            return self.step_ids(ids)
        return super()._fit_step(arena, ids)

    def step_ids(self, graph_ids) -> torch.Tensor:
        """One optimisation step on the graphs ``graph_ids`` of the arena; returns the loss (a device tensor that the
        NEXT step of the same bucket overwrites in graph mode: read or clone it before stepping again)."""
        self._steps_taken = getattr(self, "_steps_taken", 0) + 1
        if self._steps_taken % self.overflow_check_every == 0:
            # loops that call step_ids directly never reach fit()'s epoch-end check: the device's sticky capacity-overflow flag is read
            # here every so often (one 4-byte read; nothing at all unless a capacity-bound kernel was ever launched on this device)
            from .native import ops as _ops

            _ops.check_overflow_flags()
        fill, plan = None, None
        if self.stable:
            bucket, fill, plan, cap = self._stable_bucket(np.asarray(graph_ids, dtype=np.int64))
        else:
            bucket = self.bucket_of(graph_ids)
            cap = bucket[4] if len(bucket) > 4 else None
        sel, nptr, eptr, nb, eb, real = self.arena.selection(graph_ids, bucket[:2], filler_sizes=fill)
        host = np.concatenate([sel, nptr, eptr]).astype(np.int32)
        sizes = nptr[1:] - nptr[:-1]
        entry = self._entries.get(bucket) if self.graphs else None
        eager = not self.graphs
        if self.graphs and entry is None and self._pattern_model and not self.stable:
            first_sight = bucket not in self._seen
            if first_sight and len(self._seen) >= 4096:
                self._seen.pop(next(iter(self._seen)))       # forget the oldest pattern: it gets its eager pass again
            self._seen[bucket] = True
            eager = first_sight or len(self._entries) >= self.max_pattern_captures
        if self.graphs and entry is None and self.stable and len(self._entries) >= self.max_pattern_captures:
            eager = True                 # a rare bucket beyond the capture budget (each capture holds a graph and a pinned ring)
        if eager:
            packed = torch.from_numpy(host).to(self.flat_param.device, non_blocking=True)
            return self._step_on(packed, len(sel), nb, eb, sizes, real, cap, plan)
        if entry is None:
            if not self._warm:
                self._warm_up(graph_ids, bucket)
            # the selection travels through a small ring of pinned buffers: the host may run several steps ahead of the
            # device, so a buffer is rewritten only after the copy that read it has completed (its event)
            entry = {"ring": [[torch.empty(len(host), dtype=torch.int32).pin_memory(), None] for _ in range(4)], "turn": 0,
                     "packed": torch.empty(len(host), dtype=torch.int32, device=self.flat_param.device),
                     "graph": torch.cuda.CUDAGraph()}
            self._send(entry, host)
            torch.cuda.synchronize()
            kw = {"stream": self._capture_stream()}
            if self._pool is not None:
                kw["pool"] = self._pool
            whole = not self.split
            if self.split and self.capture_collective and self.collective_in_graph is not False:
                try:        # assembly, forward, backward, all-reduce, Adam as ONE graph
                    with torch.cuda.graph(entry["graph"], **kw):
                        entry["loss"] = self._step_on(entry["packed"], len(sel), nb, eb, sizes, real, cap, plan)
                    whole, self.collective_in_graph = True, True
                except Exception as exc:          # this RCCL / runtime does not capture it: the two-graph form from here on
                    self.collective_in_graph, self.collective_capture_error = False, f"{type(exc).__name__}: {exc}"
                    torch.cuda.synchronize()
                    entry["graph"] = torch.cuda.CUDAGraph()
            if whole:
                if "loss" not in entry:
                    with torch.cuda.graph(entry["graph"], **kw):
                        entry["loss"] = self._step_on(entry["packed"], len(sel), nb, eb, sizes, real, cap, plan)
            else:
                with torch.cuda.graph(entry["graph"], **kw):
                    entry["loss"] = self._local_half(entry["packed"], len(sel), nb, eb, sizes, real, cap, plan)
            entry["whole"] = whole
            if self._pool is None:
                self._pool = entry["graph"].pool()
            if self.split and not whole and self._update_graph is None:
                # Adam touches the flat buffers only: one graph serves all buckets (its own memory pool: it is replayed after
                # whichever bucket's graph ran, not in capture order)
                self._update_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._update_graph, stream=self._capture_stream()):
                    self.optimizer.step()
            self._entries[bucket] = entry
        else:
            self._send(entry, host)
        entry["graph"].replay()
        if self.split and not entry["whole"]:
            if self.distributed:
                self._prescaled = True      # the replayed half scaled the flat buffer by 1/world (its capture ran _forward_backward)
                t0 = time.perf_counter()
                self.all_reduce_gradients()
                self.host_between_replays_s += time.perf_counter() - t0      # host time of the eager collective between the two replays
                self.host_between_replays_n += 1
            self._update_graph.replay()
        return entry["loss"]

    @staticmethod
    def _send(entry, host: np.ndarray):
        slot = entry["ring"][entry["turn"] % len(entry["ring"])]
        entry["turn"] += 1
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0].copy_(torch.from_numpy(host))
        entry["packed"].copy_(slot[0], non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()


class BucketedPredictor:
    """Inference analogue of :class:`BucketedTrainer`: model outputs for selections of a device-resident arena, with the WHOLE forward
    -- device batch assembly and the model -- captured once per size bucket in a hipGraph and replayed.  The serial decorator path
    (blackwater/library/ngem/estimator.py:49-84: one model call per circuit) runs on it: a 4-qubit circuit is ~270 graph nodes, its
    ~45 launches take the host 0.6-0.8 ms to enqueue one by one and 0.1 ms to replay (the per-circuit device path was slower than
    the CPU oracle's loop: 1.2 k against 1.7 k circuits/s).  Models whose launch shapes follow every graph's size
    (``needs_size_pattern``: Family B) are run eagerly -- a per-circuit pattern never repeats.  The model is used in eval mode under
    ``torch.no_grad()``; outputs of a replayed bucket are overwritten by the next replay of the same bucket."""

    def __init__(self, model: nn.Module, arena, node_quantum: int = 256, edge_quantum: int = 512, graphs: bool = True):
        if not arena.filler_nodes:
            raise ValueError("BucketedPredictor needs an arena built with filler_nodes > 0")
        import weakref

        # the model through a WEAK reference: a predictor kept per model (library/ngem/estimator._predictors is keyed weakly by the
        # model) must not be what keeps the model -- and with it the captures, their pool and the arena -- alive
        self._model_ref = weakref.ref(model)
        self.nq, self.eq = int(min(node_quantum, arena.filler_nodes)), int(edge_quantum)
        self.graphs = graphs and not getattr(model, "needs_size_pattern", False)
        self._entries, self._pool, self._stream, self._warm = {}, None, None, False
        self.captures = 0
        self.arena = arena.with_capacity(2.0) if self.graphs else arena
        self._weights = self._weight_addresses()

    @property
    def model(self):
        model = self._model_ref()
        if model is None:
            raise RuntimeError("BucketedPredictor: its model has been garbage-collected")
        return model

    def _weight_addresses(self):
        return tuple(p.data_ptr() for p in self.model.parameters())

    def load(self, arena) -> None:
        """Another arena's graphs under the SAME captured forwards: its arrays are copied to the addresses the captures read (the
        circuits of the next run(); data/arena.py refill_from).  An arena that outgrew the allocation, or a model whose parameters
        moved (``.to()``, a replaced layer), drops the captures and starts again."""
        if not self.graphs:
            self.arena = arena
            return
        moved = self._weight_addresses() != self._weights
        if moved or not self.arena.refill_from(arena):
            self._entries.clear()
            self._pool = None               # the graphs' memory pool went with the last of them
            self.arena = arena.with_capacity(2.0)
            self._weights = self._weight_addresses()

    def _forward(self, packed, b, n_pad, e_pad, sizes, num_real):
        batch = self.arena.assemble(packed, b, n_pad, e_pad, sizes, None, num_real)
        return self.model(*batch.model_args())

    def predict_ids(self, graph_ids) -> torch.Tensor:
        """Outputs [len(graph_ids), out] of the model for these graphs (the filler row of a padded batch is cut off)."""
        sel0 = np.asarray(graph_ids, dtype=np.int64)
        nb, eb = int(self.arena.node_counts[sel0].sum()), int(self.arena.edge_counts[sel0].sum())
        bucket = (-(-nb // self.nq) * self.nq, -(-max(eb, 1) // self.eq) * self.eq, len(sel0))
        sel, nptr, eptr, n_pad, e_pad, real = self.arena.selection(sel0, bucket[:2])
        host = np.concatenate([sel, nptr, eptr]).astype(np.int32)
        sizes = nptr[1:] - nptr[:-1]
        dev = self.arena.device
        was_training = self.model.training
        if was_training:                # (Module.eval() walks every submodule: a third of a replayed call's host time when not needed)
            self.model.eval()
        try:
            with torch.no_grad():
                if not self.graphs:
                    packed = torch.from_numpy(host).to(dev, non_blocking=True)
                    return self._forward(packed, len(sel), n_pad, e_pad, sizes, real)[:real]
                entry = self._entries.get(bucket)
                if entry is None:
                    if self._stream is None:
                        self._stream = torch.cuda.Stream()
                    entry = {"ring": [[torch.empty(len(host), dtype=torch.int32).pin_memory(), None] for _ in range(4)], "turn": 0,
                             "packed": torch.empty(len(host), dtype=torch.int32, device=dev), "graph": torch.cuda.CUDAGraph()}
                    BucketedTrainer._send(entry, host)
                    if not self._warm:          # torch asks for eager iterations on the capture stream before the first capture
                        self._stream.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(self._stream):
                            for _ in range(3):
                                self._forward(entry["packed"], len(sel), n_pad, e_pad, sizes, real)
                        torch.cuda.current_stream().wait_stream(self._stream)
                        self._warm = True
                    torch.cuda.synchronize()
                    kw = {"stream": self._stream}
                    if self._pool is not None:
                        kw["pool"] = self._pool
                    with torch.cuda.graph(entry["graph"], **kw):
                        entry["out"] = self._forward(entry["packed"], len(sel), n_pad, e_pad, sizes, real)
                    if self._pool is None:
                        self._pool = entry["graph"].pool()
                    self._entries[bucket] = entry
                    self.captures += 1
                else:
                    BucketedTrainer._send(entry, host)
                entry["graph"].replay()
                return entry["out"][:real]
        finally:
            if was_training:
                self.model.train()

