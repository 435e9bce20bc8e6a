"""Thin typed wrappers over the C ABI (no autograd here): tensors in, raw pointers out.

Every function asserts what the kernel assumes (device, dtype, row-major with unit column stride, sizes
matching the CSR arrays) BEFORE launching -- a kernel that reads out of bounds can take the whole node down.
"""
from __future__ import annotations

import os

from typing import Optional

import torch

from . import _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """The HIP stream torch currently launches on, as the integer handle the C ABI takes."""
    if _raw_stream is not None:  # same handle as below without building a torch.cuda.Stream object per launch
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _mat(t: torch.Tensor, name: str, dtype=torch.float32) -> int:
    """Checks a 2-D row-major device matrix and returns its leading dimension."""
    if not t.is_cuda:
        raise _lib.NativeLibraryError(f"{name} must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != dtype or t.dim() != 2:
        raise ValueError(f"{name}: want 2-D {dtype}, got {t.dim()}-D {t.dtype}")
    if t.shape[0] > 1 and t.shape[1] > 0 and t.stride(1) != 1:
        raise ValueError(f"{name}: columns must be contiguous (stride {t.stride()})")
    return int(t.stride(0)) if t.shape[0] > 1 else max(int(t.shape[1]), 1)


class RowsOf:
    """Rows ``rows`` (int32 [n], device) of the padded matrix ``base`` [M, c]: a gathered matrix that is never
    materialised.  The dense entry points take it as their ``x`` operand (``x_rows`` of the C ABI), so the first layers
    of a model read their input rows straight from the device-resident dataset instead of from a per-batch copy."""

    requires_grad = False
    is_cuda = True

    def __init__(self, base: torch.Tensor, rows: torch.Tensor):
        if (not rows.is_cuda or rows.dtype != torch.int32 or rows.dim() != 1 or (rows.numel() > 1 and rows.stride(0) != 1)):
            raise ValueError("RowsOf: rows must be a contiguous 1-D int32 cuda tensor")
        _mat(base, "base")
        self.base, self.rows = base, rows
        self.shape = (int(rows.shape[0]), int(base.shape[1]))
        self.device, self.dtype = base.device, base.dtype

    def dim(self):
        return 2

    def materialize(self) -> torch.Tensor:
        """The gathered matrix in the padded row layout (one torch gather over whole padded rows)."""
        n, c = self.shape
        ld = int(self.base.stride(0)) if self.base.shape[0] > 1 else (c + 3) // 4 * 4
        whole = torch.as_strided(self.base, (self.base.shape[0], ld), (ld, 1))
        return whole.index_select(0, self.rows.long())[:, :c]


def _x_operand(x, name="x"):
    """(tensor to address, leading dimension, rows of the operand, row-map pointer or None) for a dense-kernel input."""
    if isinstance(x, RowsOf):
        return x.base, _mat(x.base, name), x.shape[0], x.rows.data_ptr()
    return x, _mat(x, name), x.shape[0], None


def padded_empty(n: int, c: int, device) -> torch.Tensor:
    """An [n, c] fp32 view of a buffer whose rows hold round_up(c, 4) floats (16-byte aligned rows): the kernels
    then move 16 bytes per lane, ~1.4x the rate of 8-byte accesses.  The pad columns are scratch."""
    c4 = (c + 3) // 4 * 4
    return torch.empty((max(n, 1), c4), dtype=torch.float32, device=device)[:n, :c]


def padded_copy(t: torch.Tensor) -> torch.Tensor:
    """``t`` copied into the padded row layout."""
    out = padded_empty(t.shape[0], t.shape[1], t.device)
    out.copy_(t)
    return out


def rowmajor(t: torch.Tensor) -> torch.Tensor:
    """``t`` itself when its columns are unit-stride (padded views stay padded), else a compact copy."""
    if isinstance(t, RowsOf):
        return t
    if t.dim() == 2 and (t.shape[1] <= 1 or t.stride(1) == 1) and (t.shape[0] <= 1 or t.stride(0) >= t.shape[1]):
        return t
    return t.contiguous()


def _vec(t: Optional[torch.Tensor], name: str, n: int, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda or t.dtype != dtype or t.dim() != 1 or t.shape[0] < n or (t.numel() > 1 and t.stride(0) != 1):
        raise ValueError(f"{name}: want contiguous 1-D {dtype} cuda tensor with >= {n} entries, got "
                         f"{tuple(t.shape)} {t.dtype} {t.device}")


def _owns_pad_columns(t, name):
    """The aggregation kernels move 16 bytes per lane whenever the operands' rows are 16-byte aligned, and then WRITE
    the columns between C and round_up(C, 4) of the output along with the rest.  In the padded row layout
    (``padded_empty``) those columns are scratch; in a column slice of a wider matrix they are a neighbour's data."""
    c = t.shape[1]
    c4 = (c + 3) // 4 * 4
    if c % 4 and t.shape[0] > 1 and t.stride(0) != c4 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0:
        raise ValueError(f"{name}: [N, {c}] view with row stride {t.stride(0)}: the kernel would overwrite columns {c}..{c4 - 1} "
                         f"of every row; write into a padded_empty(N, {c}) buffer instead")


_seed_counter = None   # optional device uint64 (an int64 tensor of one element) the dropout epilogues add to their seed


def set_seed_counter(t):
    """Installs (or, with None, removes) the device-resident step counter added to every dropout seed: launches captured
    in a hipGraph draw a fresh mask per replay as the owner bumps the counter (train.BucketedTrainer)."""
    global _seed_counter
    if t is not None and (not t.is_cuda or t.dtype != torch.int64 or t.numel() != 1):
        raise ValueError("seed counter: one int64 element on the device")
    _seed_counter = t


def csr_aggregate(x, ptr, idx, *, ell=None, cscale=None, rscale=None, dself=None, alpha=1.0, z=None, beta=0.0, bias=None,
                  relu=False, drop_p=0.0, seed=0, out=None, pool=None):
    """out = act(alpha * (rscale * sum_e cscale[idx[e]] x[idx[e]] + dself * x) + beta * z + bias).

    ``pool`` = a dict with ``graph_ptr``, ``num_graphs``, optional ``weights`` and the flags ``mean`` / ``wmean``: the pooled
    means of ``out`` come from the same launch (mlqem_csr_aggregate_pool_f32) and are left in ``pool["out_mean"]`` /
    ``pool["out_wmean"]`` ([B, c] tensors, None for a mean that was not asked for).  ``bits``: also ``pool["out_bits"]``, the
    sign bits of ``out`` (an opaque uint8 buffer in the launch's tiling -- ``pool_gate_unpack`` decodes it -- that
    ``segment_pool_bwd(gate_bits=)`` gates with); with ``bits`` and
    ``store=False`` the activation is not written at all and None is returned.  When the fused form is switched off or does
    not serve the shape, ``pool`` is left untouched, ``out`` is written and the caller pools it itself (``pooled_means``)."""
    n, c = x.shape
    ldx = _mat(x, "x")
    _vec(ptr, "ptr", n + 1, torch.int32)
    _vec(idx, "idx", 0, torch.int32)
    _ell(ell, n)
    for nm, v in (("cscale", cscale), ("rscale", rscale), ("dself", dself)):
        _vec(v, nm, n)
    _vec(bias, "bias", c)
    fusable = pool is not None and ell is not None and _POOL_FUSED and n > 0 and int(pool["num_graphs"]) > 0
    # the pooled form with gate bits and store=False never writes the activation: no buffer for it unless the launch is refused
    no_store = fusable and bool(pool.get("bits", False)) and not bool(pool.get("store", True))
    if no_store:
        out, ldo = None, (c + 3) // 4 * 4
    else:
        if out is None:
            out = padded_empty(n, c, x.device)
        else:
            _owns_pad_columns(out, "out")
        ldo = _mat(out, "out")
    ldz = 0
    if z is not None:
        if z.shape != x.shape:
            raise ValueError("z must have the shape of x")
        ldz = _mat(z, "z")
    lib = _lib.load()
    if fusable:
        gptr, nb, wts = pool["graph_ptr"], int(pool["num_graphs"]), pool.get("weights")
        want_mean, want_wmean = bool(pool.get("mean", True)), bool(pool.get("wmean", True))
        want_bits = bool(pool.get("bits", False))
        store = not no_store
        _vec(gptr, "graph_ptr", nb + 1, torch.int32)
        _vec(wts, "weights", n)
        mean = padded_empty(nb, c, x.device) if want_mean else None
        wmean = padded_empty(nb, c, x.device) if want_wmean else None
        bits = torch.empty(lib.mlqem_csr_aggregate_pool_gate_bytes(n, c), dtype=torch.uint8, device=x.device) if want_bits else None
        need = lib.mlqem_csr_aggregate_pool_workspace_bytes(n, nb, c)
        ws = _wgrad_workspace(x.device, need)
        code = lib.mlqem_csr_aggregate_pool_f32(
            _p(x), ldx, _p(ptr), _p(idx), _p(ell), _p(cscale), _p(rscale), _p(dself), float(alpha), float(beta), _p(z), ldz,
            _p(bias), 1 if relu else 0, float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(_seed_counter) if drop_p > 0 else None,
            _p(out) if store else None, ldo, n, c, _p(wts), _p(gptr), nb, _p(mean), _mat(mean, "mean") if want_mean else 0,
            _p(wmean), _mat(wmean, "wmean") if want_wmean else 0, _p(bits), _p(ws), need, _stream())
        if code != _lib.ERR_UNSUPPORTED:
            _lib.check(code, "mlqem_csr_aggregate_pool_f32")
            pool["out_mean"], pool["out_wmean"], pool["out_bits"] = mean, wmean, bits
            return out if store else None
        if out is None:                # refused (a shape the pooled form does not serve): the plain launch needs the buffer
            out = padded_empty(n, c, x.device)
            ldo = _mat(out, "out")
    common = (_p(x), ldx, _p(ptr), _p(idx), _p(ell), _p(cscale), _p(rscale), _p(dself), float(alpha), float(beta), _p(z), ldz,
              _p(bias), 1 if relu else 0, float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF,
              _p(_seed_counter) if drop_p > 0 else None, _p(out), ldo, n, c)
    code = lib.mlqem_csr_aggregate_f32(*common, _stream())
    _lib.check(code, "mlqem_csr_aggregate_f32")
    return out


def pooled_means(out, pool):
    """(mean, wmean) of a ``csr_aggregate(..., pool=pool)`` call: what the launch left in ``pool``, or ``segment_pool`` of ``out``."""
    if "out_wmean" in pool or "out_mean" in pool:
        return pool.get("out_mean"), pool.get("out_wmean")
    return segment_pool(out, pool["graph_ptr"], int(pool["num_graphs"]), weights=pool.get("weights"), mean=bool(pool.get("mean", True)),
                        wmean=bool(pool.get("wmean", True)))


# MLQEM_POOL_FUSED=0: aggregation, pool and a stored activation as separate steps (A/B).  Default: the pooled means AND the gate
# bits of a branch's last hidden activation come out of the aggregation launch that computes it (mlqem_csr_aggregate_pool_f32),
# and the activation itself is never written: 6.75-6.80 against 6.97-7.09 ms per bench step on one box.  (With the activation
# still stored the fused form was a wash: DESIGN.md section 3.)
_POOL_FUSED = __import__("os").environ.get("MLQEM_POOL_FUSED", "1") != "0"


def _ell(ell, n):
    if ell is None:
        return
    if (not ell.is_cuda or ell.dtype != torch.int32 or ell.dim() != 2 or ell.shape[1] != 2 or ell.shape[0] < n
            or not ell.is_contiguous()):
        raise ValueError("ell must be a contiguous [N,2] int32 cuda tensor")


def ell_from_csr(ptr, idx, num_nodes):
    ell = torch.empty((max(num_nodes, 1), 2), dtype=torch.int32, device=ptr.device)
    code = _lib.load().mlqem_ell_from_csr(_p(ptr), _p(idx), num_nodes, _p(ell), _stream())
    _lib.check(code, "mlqem_ell_from_csr")
    return ell


def csr_segment_max(x, ptr, idx, ell=None, out=None):
    n, c = x.shape
    ldx = _mat(x, "x")
    _vec(ptr, "ptr", n + 1, torch.int32)
    _vec(idx, "idx", 0, torch.int32)
    _ell(ell, n)
    if out is None:
        out = padded_empty(n, c, x.device)
    else:
        _owns_pad_columns(out, "out")
    code = _lib.load().mlqem_csr_segment_max_f32(_p(x), ldx, _p(ptr), _p(idx), _p(ell), _p(out), _mat(out, "out"), n, c,
                                                 _stream())
    _lib.check(code, "mlqem_csr_segment_max_f32")
    return out


_TICKET_SLOTS = 256
_ticket_pools = {}   # device -> (zeroed int32[_TICKET_SLOTS], {stream handle: slot})


def prepare_device(device):
    """Allocates, EAGERLY, the per-device state the launch wrappers hand to kernels: the pool of zero-initialised counters the
    "last workgroup done" kernels count on (they leave their counter at zero).  Trainers call this before any capture, so that
    the pool never becomes a node -- or a private-pool allocation -- of whichever hipGraph happened to be captured first; a
    stream gets its slot of the pool by host bookkeeping alone (kernels on one stream are serialised, streams that run
    concurrently never share a counter)."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    pool = _ticket_pools.get(device)
    if pool is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the ticket pool must exist before a capture begins: run a warm-up step or ops.prepare_device(device) first")
        # a fill, not torch.zeros (which may be recorded as a memset node): the form `loops` uses
        pool = _ticket_pools[device] = (torch.full((_TICKET_SLOTS,), 0, dtype=torch.int32, device=device), {})
        _overflow_flag(device)               # the sticky capacity-overflow flag: same rule, it must exist before any capture
    return pool


def _ticket(device):
    tickets, slots = prepare_device(device)
    key = _stream()
    i = slots.get(key)
    if i is None:
        i = len(slots)
        if i >= _TICKET_SLOTS:
            raise RuntimeError(f"more than {_TICKET_SLOTS} streams have launched ticketed kernels on {device}")
        slots[key] = i
    return tickets[i:i + 1]


def reset_tickets(device):
    """Re-zeroes the counters (after a launch that aborted half-way: a non-zero ticket would silently skip every later
    last-workgroup tail on its stream)."""
    device = torch.device(device)
    if device.index is None:            # the key prepare_device files the pool under ('cuda' alone found nothing: ADVICE r04)
        device = torch.device("cuda", torch.cuda.current_device())
    pool = _ticket_pools.get(device)
    if pool is not None:
        pool[0].zero_()


SEQ2_MAX_HIDDEN, SEQ2_MAX_OUT = 16, 8      # kSeqMaxH / kSeqMaxO of csrc/seq2.hip


def seq2_fits(x, w1, w2) -> bool:
    """Shapes and operands ``seq2_forward`` takes: a 2-D fp32 cuda matrix with contiguous columns, narrow hidden / output layers."""
    return (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0 and (x.shape[1] <= 1 or x.stride(1) == 1)
            and w1.dim() == 2 and w2.dim() == 2 and w1.shape[1] == x.shape[1] and w2.shape[1] == w1.shape[0]
            and w1.shape[0] <= SEQ2_MAX_HIDDEN and w2.shape[0] <= SEQ2_MAX_OUT)


def seq2_forward(x, w1, b1, w2, b2, drop_p=0.0, seed=0, keep=True):
    """(y [n, o], hidden [n, h] or None, mask [n] int32 or None) of Linear -> [Dropout(drop_p)] -> Linear in one launch
    (mlqem_seq2_forward_f32); ``keep``: also return what the backward needs."""
    n, i = x.shape
    h, o = w1.shape[0], w2.shape[0]
    dev = x.device
    y = torch.empty((n, o), dtype=torch.float32, device=dev)
    hidden = torch.empty((n, h), dtype=torch.float32, device=dev) if keep else None
    mask = torch.empty(n, dtype=torch.int32, device=dev) if (keep and drop_p > 0) else None
    code = _lib.load().mlqem_seq2_forward_f32(_p(x), int(x.stride(0)) if n > 1 else i, n, i, _p(w1), _p(b1), h, _p(w2), _p(b2), o, float(drop_p),
                                              int(seed) & 0xFFFFFFFFFFFFFFFF, _p(_seed_counter) if drop_p > 0 else None, _p(hidden),
                                              _p(mask), _p(y), o, _stream())
    _lib.check(code, "mlqem_seq2_forward_f32")
    return y, hidden, mask


def seq2_backward(gy, x, w1, w2, hidden, mask, drop_p, want_gx, want_b1=True, want_b2=True):
    """(gx or None, gw1, gb1, gw2, gb2) of the same block in one launch (mlqem_seq2_backward_f32)."""
    n, i = x.shape
    h, o = w1.shape[0], w2.shape[0]
    dev = x.device
    # an expanded gradient (`out.sum().backward()`) has stride 0; the strides of size-1 axes mean nothing (and .contiguous() keeps them)
    if gy.dim() != 2 or (o > 1 and gy.stride(1) != 1) or (n > 1 and gy.stride(0) < o):
        gy = gy.clone(memory_format=torch.contiguous_format)
    ld = lambda t: int(t.stride(0)) if t.shape[0] > 1 else int(t.shape[1])
    gx = torch.empty((n, i), dtype=torch.float32, device=dev) if want_gx else None
    gw1 = torch.empty((h, i), dtype=torch.float32, device=dev)
    gw2 = torch.empty((o, h), dtype=torch.float32, device=dev)
    gb1 = torch.empty(h, dtype=torch.float32, device=dev) if want_b1 else None
    gb2 = torch.empty(o, dtype=torch.float32, device=dev) if want_b2 else None
    lib = _lib.load()
    need = lib.mlqem_seq2_backward_workspace_bytes(n, i, h, o)
    ws = _wgrad_workspace(dev, need)
    code = lib.mlqem_seq2_backward_f32(_p(gy), ld(gy), _p(x), ld(x), n, i, _p(w1), h, _p(w2), o, _p(hidden), _p(mask),
                                       float(drop_p), _p(gx), i if want_gx else 0, _p(gw1), _p(gb1), _p(gw2), _p(gb2), _p(ws), need,
                                       _p(_ticket(dev)), _stream())
    _lib.check(code, "mlqem_seq2_backward_f32")
    return gx, gw1, gb1, gw2, gb2


def mse_loss_grad(out, target, want_grad=True, rows=None):
    """(loss, g): loss = mean((out - target)^2) as a 0-dim device tensor and g = 2 (out - target) / numel -- what
    ``MSELoss()(out, target).backward()`` hands to ``out`` -- from ONE launch (mlqem_mse_loss_grad_f32).  2-D fp32 operands
    with contiguous columns (row-strided views are fine).  ``rows``: only the first ``rows`` rows enter the loss (the rest are
    a padded batch's filler rows); g still has ``out``'s shape, zero beyond them -- what slicing the output and letting
    autograd pad the gradient back produces with a fill and a copy."""
    if out.dim() != 2 or target.dim() != 2 or out.shape[1] != target.shape[1] or out.numel() == 0:
        raise ValueError("mse_loss_grad: out and target must be non-empty 2-D tensors of one width")
    g_rows, c = out.shape
    n = g_rows if rows is None else int(rows)
    if not (1 <= n <= g_rows) or target.shape[0] < n:
        raise ValueError("mse_loss_grad: rows must lie in [1, out.shape[0]] and target must hold them")
    ldo, ldy = _mat(out, "out"), _mat(target, "target")
    g = torch.empty((g_rows, c), dtype=torch.float32, device=out.device) if want_grad else None
    loss = torch.empty((), dtype=torch.float32, device=out.device)
    lib = _lib.load()
    need = lib.mlqem_mse_loss_workspace_bytes()
    ws = _wgrad_workspace(out.device, need)
    code = lib.mlqem_mse_loss_grad_f32(_p(out), ldo, _p(target), ldy, _p(g), c, n, c, g_rows, _p(loss), _p(ws), need, _p(_ticket(out.device)),
                                       _stream())
    _lib.check(code, "mlqem_mse_loss_grad_f32")
    return loss, g


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, bump=None):
    """One Adam update of the flat fp32 buffer ``param`` in place (mlqem_adam_step_f32); ``step`` (0-dim fp32) and ``lr`` (0-dim
    fp32) live on the device: ``step`` is incremented by the launch, and so is ``bump`` (a one-element int64 device tensor: the
    trainers' dropout step counter) when given."""
    for name, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        if not t.is_cuda:
            raise _lib.NativeLibraryError(f"adam_step: {name} must live on the GPU; there is no CPU path")
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != param.numel():
            raise ValueError(f"adam_step: {name} must be contiguous fp32 of the parameters' size")
    if bump is not None and not (bump.is_cuda and bump.dtype == torch.int64 and bump.numel() == 1):
        raise ValueError("adam_step: bump must be a one-element int64 device tensor")
    if step.dtype != torch.float32 or lr.dtype != torch.float32 or step.numel() != 1 or lr.numel() != 1 or not (step.is_cuda and lr.is_cuda):
        raise ValueError("adam_step: step and lr must be one-element fp32 device tensors")
    code = _lib.load().mlqem_adam_step_f32(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), _p(lr), _p(step), float(beta1),
                                           float(beta2), float(eps), _p(_ticket(param.device)), _p(bump), _stream())
    _lib.check(code, "mlqem_adam_step_f32")


def relu_dropout_bwd(g, y, scale=1.0, out=None):
    """gx = (y > 0) ? g * scale : 0 on [N, C] matrices (any leading dimensions; ``out`` may alias ``g``)."""
    if g.shape != y.shape or g.dim() != 2:
        raise ValueError("relu_dropout_bwd: g and y must be 2-D of one shape")
    g, y = rowmajor(g), rowmajor(y)
    n, c = g.shape
    gx = padded_empty(n, c, g.device) if out is None else out
    if gx.shape != g.shape:
        raise ValueError("relu_dropout_bwd: bad out shape")
    code = _lib.load().mlqem_relu_dropout_bwd_f32(_p(g), _mat(g, "g"), _p(y), _mat(y, "y"), float(scale), _p(gx),
                                                  _mat(gx, "gx"), n, c, _stream())
    _lib.check(code, "mlqem_relu_dropout_bwd_f32")
    return gx


def relu_dropout(x, drop_p=0.0, seed=0, residual=None):
    """(y, s): y = dropout(relu(x)); s = y + residual (None without a residual) -- one launch (mlqem_relu_dropout_f32)."""
    x = rowmajor(x)
    n, c = x.shape
    y = padded_empty(n, c, x.device)
    s = None
    if residual is not None:
        residual = rowmajor(residual)
        if residual.shape != x.shape:
            raise ValueError("relu_dropout: residual must have the shape of x")
        s = padded_empty(n, c, x.device)
    code = _lib.load().mlqem_relu_dropout_f32(_p(x), _mat(x, "x"), float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                              _p(_seed_counter) if drop_p > 0 else None, _p(residual),
                                              _mat(residual, "residual") if residual is not None else 0, _p(y), _mat(y, "y"),
                                              _p(s), _mat(s, "s") if s is not None else 0, n, c, _stream())
    _lib.check(code, "mlqem_relu_dropout_f32")
    return y, s


def _gate(gate, n, o, padded: bool):
    """(pointer, leading dimension) of an optional [n, o] gate matrix (see mlqem_linear_f32)."""
    if gate is None:
        return None, 0
    if tuple(gate.shape) != (n, o):
        raise ValueError(f"gate must be [{n}, {o}], got {tuple(gate.shape)}")
    ld = _mat(gate, "gate")
    if n == 1:
        ld = int(gate.stride(0))
    if padded and (ld % 4 or ld < (o + 3) // 4 * 4 or gate.data_ptr() % 16):
        raise ValueError("gate: needs the padded row layout here")
    return gate.data_ptr(), ld


_MFMA_MAX_IN = 128    # input columns the matrix-core GEMM kernels take in one call (csrc/dense.hip: round_ks)


_WIDE_SPLIT_MIN_OUTPUTS = 1 << 16     # (rows x outputs) below which a wide-input product is one scalar-kernel launch


def linear(x, w, b=None, *, transposed=False, relu=False, rowscale=None, out=None, accumulate=False, drop_p=0.0,
           seed=0, rs_cols=-1, act_from=-1, gate=None, gate_scale=1.0):
    """y = act(x @ w.T + b) (``transposed=False``, w: [O,I]) or y = x @ w (``transposed=True``, w: [I,O]);
    ``gate``: y = gate > 0 ? y * gate_scale : 0 as the last step."""
    i = x.shape[1]
    # ... unless the product is tiny (the observable MLP of Family A: 1024 rows x 401 -> 10): then the split -- four column
    # slices, four copies, four launches -- costs more than the scalar kernel's one launch (a thread per output, 5 us)
    if (i > _MFMA_MAX_IN and torch.is_tensor(x) and w.dim() == 2 and w.shape[0 if transposed else 1] == i
            and x.shape[0] * w.shape[1 if transposed else 0] > _WIDE_SPLIT_MIN_OUTPUTS):
        # The matrix-core kernels hold a row's inputs in registers (<= 128 columns); wider inputs (MLP rows of 169-170
        # encode_data_v2_ecr features, docs/tutorials/mlp.py:148-194) used to fall to a scalar kernel (1.3 ms for 262 k x 170 -> 125
        # against 0.15 ms here).  Split the k range: y = x[:, :128] W[:, :128]^T + b, then y += x[:, 128:] W[:, 128:]^T with the
        # epilogue (row scale, ReLU, dropout, gate) on the last piece.  The data-gradient orientation (w: [I, O]) splits the same way,
        # over ROWS of w: the 192 padded head slots of a three-head TransformerConv's q | k | v | skip gradient (gnn.py:178-276: heads
        # 5 / 3) ran the scalar kernel, 245 us a step of the mixed corpus.
        if out is None:
            if accumulate:
                raise ValueError("linear: accumulate needs an existing out")
            out = padded_empty(x.shape[0], w.shape[1 if transposed else 0], x.device)
        for k0 in range(0, i, _MFMA_MAX_IN):
            k1 = min(k0 + _MFMA_MAX_IN, i)
            first, last = k0 == 0, k1 == i
            linear(x[:, k0:k1], w[k0:k1] if transposed else w[:, k0:k1].contiguous(), b if first else None, transposed=transposed, relu=relu and last,
                   rowscale=rowscale if last else None, out=out, accumulate=accumulate or not first,
                   drop_p=drop_p if last else 0.0, seed=seed, rs_cols=rs_cols if last else -1, act_from=act_from if last else -1,
                   gate=gate if last else None, gate_scale=gate_scale)
        return out
    xt, ldx, n, x_rows = _x_operand(x)
    if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 2 or not w.is_contiguous():
        raise ValueError("w must be a contiguous 2-D fp32 cuda tensor")
    o = w.shape[1] if transposed else w.shape[0]
    if (w.shape[0] if transposed else w.shape[1]) != i:
        raise ValueError(f"linear: x has {i} columns, w is {tuple(w.shape)} (transposed={transposed})")
    _vec(b, "b", o)
    _vec(rowscale, "rowscale", n)
    if out is None:
        if accumulate:
            raise ValueError("linear: accumulate needs an existing out")
        out = padded_empty(n, o, x.device)
    elif tuple(out.shape) != (n, o):
        raise ValueError("linear: bad out shape")
    code = _lib.load().mlqem_linear_f32(_p(xt), ldx, _p(w), 1 if transposed else 0, _p(b), _p(rowscale), _p(out), _mat(out, "out"),
                                        n, i, o, 1 if relu else 0, 1 if accumulate else 0, float(drop_p),
                                        int(seed) & 0xFFFFFFFFFFFFFFFF, int(rs_cols), int(act_from),
                                        *_gate(gate, n, o, False), float(gate_scale), x_rows, _stream())
    _lib.check(code, "mlqem_linear_f32")
    return out


def linear_bf16(x, w, b=None, *, relu=False, transposed=False, out=None):
    """y = act(x @ w.T + b) (w: [O, I]) or, ``transposed``, y = x @ w (w: [I, O]) on the bf16 matrix cores: operands rounded
    to bf16 in registers, fp32 accumulation, fp32 in and out (mlqem_linear_bf16_f32)."""
    n, i = x.shape
    ldx = _mat(x, "x")
    if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 2 or not w.is_contiguous() or w.shape[0 if transposed else 1] != i:
        raise ValueError(f"linear_bf16: w must be a contiguous fp32 cuda tensor matching {i} input columns, got {tuple(w.shape)}")
    o = w.shape[1] if transposed else w.shape[0]
    _vec(b, "b", o)
    if out is None:
        out = padded_empty(n, o, x.device)
    elif out.shape != (n, o):
        raise ValueError("linear_bf16: bad out shape")
    code = _lib.load().mlqem_linear_bf16_f32(_p(x), ldx, _p(w), 1 if transposed else 0, _p(b), _p(out), _mat(out, "out"), n, i, o,
                                             1 if relu else 0, _stream())
    _lib.check(code, "mlqem_linear_bf16_f32")
    return out


def linear_wgrad_bf16(gy, x, gw, gb=None, accumulate=False):
    """gw (+)= bf16(gy).T @ bf16(x); gb (+)= bf16(gy).sum(0) with fp32 accumulation (mlqem_linear_wgrad_bf16_f32)."""
    n, o = gy.shape
    i = x.shape[1]
    if x.shape[0] != n or gw.shape != (o, i) or not gw.is_contiguous() or gw.dtype != torch.float32:
        raise ValueError("linear_wgrad_bf16: shape mismatch")
    _vec(gb, "gb", o)
    lib = _lib.load()
    need = lib.mlqem_linear_wgrad_workspace_bytes(i, o)
    ws = _wgrad_workspace(gy.device, need)
    code = lib.mlqem_linear_wgrad_bf16_f32(_p(gy), _mat(gy, "gy"), _p(x), _mat(x, "x"), _p(gw), _p(gb), n, i, o,
                                           1 if accumulate else 0, _p(ws), need, _stream())
    _lib.check(code, "mlqem_linear_wgrad_bf16_f32")


def _col_parts(blocks, name: str, vector_rows: bool) -> "_lib.ColParts":
    """The C view of a list of [N, c] blocks (one shape, padded rows when ``vector_rows``)."""
    if not 1 <= len(blocks) <= _lib.MAX_COL_PARTS:
        raise ValueError(f"{name}: 1..{_lib.MAX_COL_PARTS} column blocks, got {len(blocks)}")
    n, c = blocks[0].shape
    width = (c + 3) // 4 * 4
    cp = _lib.ColParts()
    cp.count, cp.width, cp.cols, cp.reserved = len(blocks), width, c, 0
    for k, t in enumerate(blocks):
        if tuple(t.shape) != (n, c):
            raise ValueError(f"{name}: blocks must share one shape, got {tuple(t.shape)} vs {(n, c)}")
        _mat(t, f"{name}[{k}]")
        ld = int(t.stride(0)) if n >= 1 else width          # a one-row block still has to own its padding
        if vector_rows and (ld < width or ld % 4 or t.data_ptr() % 16):
            raise ValueError(f"{name}[{k}]: needs the padded row layout (ops.padded_empty), got stride {ld}")
        cp.ptr[k], cp.ld[k] = t.data_ptr(), ld
    return cp


def _ptr_array(tensors, count):
    import ctypes as _ct

    arr = (_ct.c_void_p * _lib.MAX_COL_PARTS)()
    for k in range(count):
        t = tensors[k] if tensors is not None else None
        arr[k] = None if t is None else t.data_ptr()
    return arr


def linear_parts(xs, ws, ys, *, w_minus=None, biases=None, rowscales=None, transposed=False, gate=None, gate_scale=1.0):
    """Projections over column blocks in separate (padded) buffers, the blocks on one side:
    fan-out (one x, ``transposed=False``): ys[k] = xs[0] @ (ws[k] - w_minus[k]).T + biases[k], ws[k]: [O, I];
    fan-in (one y, ``transposed=True``): ys[0] = sum_k xs[k] @ (ws[k] - w_minus[k]), ws[k]: [cols(xs[k]), cols(ys[0])].
    Weight blocks are the layers' own contiguous tensors -- nothing is concatenated or padded on the host."""
    import ctypes as _ct

    x_rows = None
    if isinstance(xs[0], RowsOf):       # the (single) input block read through a row map
        if len(xs) != 1:
            raise ValueError("linear_parts: a RowsOf input must be the only input block")
        x_rows, n = xs[0].rows.data_ptr(), xs[0].shape[0]
        xs = [xs[0].base]
    else:
        n = xs[0].shape[0]
    xp, yp = _col_parts(xs, "xs", True), _col_parts(ys, "ys", True)
    if ys[0].shape[0] != n:
        raise ValueError("linear_parts: xs and ys differ in rows")
    if (len(ys) if transposed else len(xs)) != 1:
        raise ValueError("linear_parts: the column blocks sit on one side (fan-out: one x; fan-in: one y)")
    nblk = len(xs) if transposed else len(ys)
    want = (xs[0].shape[1], ys[0].shape[1]) if transposed else (ys[0].shape[1], xs[0].shape[1])
    for name, group in (("ws", ws), ("w_minus", w_minus)):
        if group is None:
            continue
        if len(group) != nblk:
            raise ValueError(f"linear_parts: {name} needs {nblk} blocks")
        for w in group:
            if w is not None and (not w.is_cuda or w.dtype != torch.float32 or tuple(w.shape) != want or not w.is_contiguous()):
                raise ValueError(f"linear_parts: every {name} block must be a contiguous fp32 cuda tensor of shape {want}")
    if any(w is None for w in ws):
        raise ValueError("linear_parts: ws blocks cannot be None")
    if biases is not None:
        if transposed or len(biases) != nblk:
            raise ValueError("linear_parts: biases go with the fan-out form, one (or None) per block")
        for bvec in biases:
            _vec(bvec, "bias", want[0])
    if rowscales is not None:
        if transposed or len(rowscales) != nblk:
            raise ValueError("linear_parts: rowscales go with the fan-out form, one (or None) per block")
        for rvec in rowscales:
            _vec(rvec, "rowscale", n)
    if gate is not None and len(ys) != 1:
        raise ValueError("linear_parts: a gate needs a single output block")
    code = _lib.load().mlqem_linear_parts_f32(_ct.addressof(xp), _ptr_array(ws, nblk), _ptr_array(w_minus, nblk),
                                              1 if transposed else 0, _ptr_array(biases, nblk),
                                              _ptr_array(rowscales, nblk), _ct.addressof(yp), n,
                                              *_gate(gate, n, ys[0].shape[1], True), float(gate_scale), x_rows, _stream())
    _lib.check(code, "mlqem_linear_parts_f32")
    return ys


_wgrad_ws = {}   # (device, stream, bytes) -> partial-sum workspace: per stream, so concurrent streams never share one


def _wgrad_workspace(device, need: int):
    key = (device, _stream(), need)
    ws = _wgrad_ws.get(key)
    if ws is None:
        ws = _wgrad_ws[key] = torch.empty(need, dtype=torch.uint8, device=device)
    return ws


def linear_wgrad_parts(gys, x, gw, gb=None, accumulate=False):
    """gw[k*W + o, i] (+)= sum_n gys[k][n, o] x[n, i]; gb likewise (W = round_up(O, 4); padding rows come out 0)."""
    i = x.shape[1]
    xt, ldx, n, x_rows = _x_operand(x)
    gp = _col_parts(gys, "gys", False)
    o_tot = gp.count * gp.width
    if gys[0].shape[0] != n or tuple(gw.shape) != (o_tot, i) or not gw.is_contiguous() or gw.dtype != torch.float32:
        raise ValueError(f"linear_wgrad_parts: gw must be contiguous fp32 [{o_tot}, {i}]")
    _vec(gb, "gb", o_tot)
    lib = _lib.load()
    need = lib.mlqem_linear_wgrad_workspace_bytes(i, o_tot)
    ws = _wgrad_workspace(x.device, need)
    import ctypes as _ct

    code = lib.mlqem_linear_wgrad_parts_f32(_ct.addressof(gp), _p(xt), ldx, _p(gw), _p(gb), n, i,
                                            1 if accumulate else 0, _p(ws), need, x_rows, _stream())
    _lib.check(code, "mlqem_linear_wgrad_parts_f32")


def linear_wgrad(gy, x, gw, gb=None, accumulate=False):
    """gw (+)= gy.T @ x ; gb (+)= gy.sum(0)."""
    n, o = gy.shape
    i = x.shape[1]
    xt, ldx, xn, x_rows = _x_operand(x)
    if xn != n or gw.shape != (o, i) or not gw.is_contiguous() or gw.dtype != torch.float32:
        raise ValueError("linear_wgrad: shape mismatch")
    _vec(gb, "gb", o)
    lib = _lib.load()
    need = lib.mlqem_linear_wgrad_workspace_bytes(i, o)
    ws = _wgrad_workspace(gy.device, need)
    code = lib.mlqem_linear_wgrad_f32(_p(gy), _mat(gy, "gy"), _p(xt), ldx, _p(gw), _p(gb), n, i, o,
                                      1 if accumulate else 0, _p(ws), need, x_rows, _stream())
    _lib.check(code, "mlqem_linear_wgrad_f32")


MLP1_MAX_IN, MLP1_MAX_HIDDEN, MLP1_MAX_OUT = 175, 128, 4   # MLQEM_MLP1_MAX_IN / _HIDDEN_PAD / _MAX_OUT


def mlp1_fits(i: int, h: int, o2: int) -> bool:
    """Shapes the one-launch MLP head takes (csrc/mlp_head.hip); wider layers go through the per-layer GEMMs."""
    return 1 <= i <= MLP1_MAX_IN and 1 <= h <= MLP1_MAX_HIDDEN and 1 <= o2 <= MLP1_MAX_OUT


def _mlp1_x(x):
    """x in the padded row layout the head kernels read (16-byte aligned rows of round_up(I, 4) floats)."""
    if not x.is_cuda:
        raise _lib.NativeLibraryError(f"x must live on the GPU (got {x.device}); there is no CPU path")
    x = rowmajor(x)
    if x.shape[0] > 1 and (x.stride(0) % 4 or x.stride(0) < (x.shape[1] + 3) // 4 * 4 or x.data_ptr() % 16):
        x = padded_copy(x)
    elif x.shape[0] <= 1 and (x.data_ptr() % 16 or x.shape[1] % 4):
        x = padded_copy(x)
    return x


def mlp1_forward(x, w1, b1, w2, b2, bf16=False, stash=True, target=None):
    """(out [N, O2], stash, padded x[, gout]): out = relu(x w1^T + b1) w2^T + b2 in one launch (mlqem_mlp1_forward); ``stash`` =
    the hidden activation [N, 128] the backward reads, fp32 or -- ``bf16`` -- bfloat16 (None when ``stash`` is False).  With
    ``target`` [N, O2] the MSE loss is folded in: a fourth result ``gout`` = 2 (out - target) / numel, and the next
    ``mlp1_backward(..., want_loss=True)`` on this stream returns the loss."""
    x = _mlp1_x(x)
    n, i = x.shape
    h, o2 = w1.shape[0], w2.shape[0]
    for name, t, shape in (("w1", w1, (h, i)), ("b1", b1, (h,)), ("w2", w2, (o2, h)), ("b2", b2, (o2,))):
        if not t.is_cuda or t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous():
            raise ValueError(f"mlp1_forward: {name} must be a contiguous fp32 cuda tensor of shape {shape}, got {tuple(t.shape)}")
    out = torch.empty((n, o2), dtype=torch.float32, device=x.device)
    hs = torch.empty((max(n, 1), MLP1_MAX_HIDDEN), dtype=torch.bfloat16 if bf16 else torch.float32, device=x.device) if stash else None
    ldx = _mat(x, "x") if n > 1 else (i + 3) // 4 * 4
    lib = _lib.load()
    need = lib.mlqem_mlp1_workspace_bytes(i, o2)
    ws = _wgrad_workspace(x.device, need)
    gout, ldt = None, 0
    if target is not None:
        if tuple(target.shape) != (n, o2) or target.dtype != torch.float32 or not target.is_cuda or n == 0:
            raise ValueError("mlp1_forward: target must be a non-empty fp32 cuda tensor of the output's shape")
        ldt = _mat(target, "target") if n > 1 else o2
        gout = torch.empty((n, o2), dtype=torch.float32, device=x.device)
    code = lib.mlqem_mlp1_forward(_p(x), ldx, _p(w1), _p(b1), _p(w2), _p(b2), _p(hs), _p(out), o2, n, i, h, o2,
                                  1 if bf16 else 0, _p(target), ldt, _p(gout), o2, _p(ws), need, _stream())
    _lib.check(code, "mlqem_mlp1_forward")
    return (out, hs, x) if target is None else (out, hs, x, gout)


def mlp1_backward(gout, x, hs, w2, i, h, bf16=False, dst=None, want_loss=False):
    """(gw1 [H, I], gb1, gw2 [O2, H], gb2) from one pass over x and the stash (mlqem_mlp1_backward); ``x`` as returned by
    :func:`mlp1_forward` (padded rows).  ``dst`` = four contiguous fp32 tensors to write the gradients into (a trainer's flat
    gradient slots); ``want_loss`` appends the MSE loss of the preceding ``mlp1_forward(..., target=)`` (a 0-dim tensor)."""
    n, o2 = gout.shape
    gout = rowmajor(gout)
    if gout.dtype != torch.float32 or hs.shape != (max(n, 1), MLP1_MAX_HIDDEN) or hs.dtype != (torch.bfloat16 if bf16 else torch.float32):
        raise ValueError("mlp1_backward: gout / stash do not match the forward call")
    if tuple(x.shape) != (n, i) or tuple(w2.shape) != (o2, h) or not w2.is_contiguous():
        raise ValueError("mlp1_backward: shape mismatch")
    dev = x.device
    if dst is None:
        gw1, gb1 = torch.empty((h, i), dtype=torch.float32, device=dev), torch.empty(h, dtype=torch.float32, device=dev)
        gw2, gb2 = torch.empty((o2, h), dtype=torch.float32, device=dev), torch.empty(o2, dtype=torch.float32, device=dev)
    else:
        gw1, gb1, gw2, gb2 = dst
        for t, shape in ((gw1, (h, i)), (gb1, (h,)), (gw2, (o2, h)), (gb2, (o2,))):
            if tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                raise ValueError("mlp1_backward: dst tensors must be contiguous fp32 cuda tensors of the gradients' shapes")
    loss = torch.empty((), dtype=torch.float32, device=dev) if want_loss else None
    lib = _lib.load()
    need = lib.mlqem_mlp1_workspace_bytes(i, o2)
    ws = _wgrad_workspace(dev, need)
    ldx = _mat(x, "x") if n > 1 else (i + 3) // 4 * 4
    ldg = int(gout.stride(0)) if n > 1 else o2
    code = lib.mlqem_mlp1_backward(_p(gout), ldg, _p(x), ldx, _p(hs), _p(w2), _p(gw1), _p(gb1), _p(gw2), _p(gb2), n, i, h, o2,
                                   1 if bf16 else 0, _p(loss), _p(ws), need, _stream())
    _lib.check(code, "mlqem_mlp1_backward")
    return (gw1, gb1, gw2, gb2) if not want_loss else (gw1, gb1, gw2, gb2, loss)


# ---- the layers of MLP2 / MLP3 in training mode (csrc/mlp_layers.hip): every activation is a [N, 128] matrix, bfloat16 (mfma = "bf16":
# half the bytes of every pass) or float32 (mfma = "f32": the reference's own arithmetic).  The wrappers below take either -- the
# storage is the dtype of the activation they are handed -- and call the _bf16 / _f32 entry point of the same name.
LAYER_W = 128


def _layer_ws(device):
    need = _lib.load().mlqem_layer_workspace_bytes()
    return _wgrad_workspace(device, need), need


_layer_consts = {}   # device -> (ones [128], zeros [128]): the scale / shift of a block without BatchNorm


def layer_identity_vectors(device):
    c = _layer_consts.get(device)
    if c is None:
        c = _layer_consts[device] = (torch.ones(LAYER_W, dtype=torch.float32, device=device), torch.zeros(LAYER_W, dtype=torch.float32, device=device))
    return c


def _act(n, device, f32=False):
    return torch.empty((max(n, 1), LAYER_W), dtype=torch.float32 if f32 else torch.bfloat16, device=device)


def _is_act(t):
    return t.dim() == 2 and t.shape[1] == LAYER_W and t.is_contiguous()


def _sfx(f32):
    return "f32" if f32 else "bf16"


def _seed_args(drop_p, seed):
    return float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF, (_p(_seed_counter) if drop_p > 0 else None)


def layer_gemm_bf16(x, w, b=None, *, transposed=False, add=None, out_f32=False, relu=False, drop_p=0.0, seed=0):
    """Y = X W^T + b (W [U, K]) or, ``transposed``, Y = X W (+ add) (W [K, U]: the data gradient).  X: fp32 [N, K] (padded rows)
    or a bf16 activation [N, 128]; returns a bf16 activation [N, 128] or, ``out_f32``, fp32 [N, U].  ``relu`` / ``drop_p``: the
    result is dropout(relu(.)) -- a block without BatchNorm in one launch (its backward gates by result > 0), as layer_gemm_f32."""
    x_bf16 = x.dtype == torch.bfloat16
    n = x.shape[0]
    k, u = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    if x_bf16:
        if tuple(x.shape[1:]) != (LAYER_W,) or not x.is_contiguous():
            raise ValueError("layer_gemm_bf16: a bf16 input must be a contiguous [N, 128] activation")
        ldx = LAYER_W
    else:
        x = _mlp1_x(x)
        if x.shape[1] != k:
            raise ValueError(f"layer_gemm_bf16: x has {x.shape[1]} columns, w wants {k}")
        ldx = _mat(x, "x") if n > 1 else (k + 3) // 4 * 4
    if not w.is_cuda or w.dtype != torch.float32 or not w.is_contiguous():
        raise ValueError("layer_gemm_bf16: w must be a contiguous fp32 cuda tensor")
    y = torch.empty((n, u), dtype=torch.float32, device=w.device) if out_f32 else _act(n, w.device)
    ws, need = _layer_ws(w.device)
    code = _lib.load().mlqem_layer_gemm_bf16(_p(x), 1 if x_bf16 else 0, ldx, _p(w), 1 if transposed else 0, _p(b), _p(add), _p(y),
                                             1 if out_f32 else 0, u, 1 if relu else 0, *_seed_args(drop_p, seed), n, k, u, _p(ws), need,
                                             _stream())
    _lib.check(code, "mlqem_layer_gemm_bf16")
    return y


def layer_gemm_f32(x, w, b=None, *, transposed=False, add=None, narrow_out=False, relu=False, drop_p=0.0, seed=0):
    """The fp32-storage twin: X fp32 -- [N, K] in padded rows (the block's input) or an fp32 activation [N, 128] -- to an fp32
    activation [N, 128] or, ``narrow_out``, fp32 [N, U]; operands are not rounded.  ``relu`` / ``drop_p``: the result is
    dropout(relu(.)) -- a block without BatchNorm in one launch (its backward gates by result > 0)."""
    n = x.shape[0]
    k, u = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] < k:
        raise ValueError(f"layer_gemm_f32: x must be fp32 [N, >= {k}]")
    x = _mlp1_x(x)
    ldx = _mat(x, "x") if n > 1 else (x.shape[1] + 3) // 4 * 4
    if not w.is_cuda or w.dtype != torch.float32 or not w.is_contiguous():
        raise ValueError("layer_gemm_f32: w must be a contiguous fp32 cuda tensor")
    if add is not None and not (add.dtype == torch.float32 and _is_act(add)):
        raise ValueError("layer_gemm_f32: add must be an fp32 activation [N, 128]")
    y = torch.empty((n, u), dtype=torch.float32, device=w.device) if narrow_out else _act(n, w.device, True)
    ws, need = _layer_ws(w.device)
    code = _lib.load().mlqem_layer_gemm_f32(_p(x), ldx, _p(w), 1 if transposed else 0, _p(b), _p(add), _p(y), 0 if narrow_out else 1,
                                            u if narrow_out else LAYER_W, 1 if relu else 0, *_seed_args(drop_p, seed), n, k, u, _p(ws), need,
                                            _stream())
    _lib.check(code, "mlqem_layer_gemm_f32")
    return y


def layer_colstats_fwd(y, gamma, beta, eps, n, c, running=None):
    """(mean, var, invstd, scale, shift) of the first ``c`` columns of the activation ``y`` over ``n`` rows (each [128], zeros
    beyond ``c``).  ``running`` = (running_mean, running_var, momentum, num_batches_tracked | None): BatchNorm1d's buffer update,
    applied by the same launch."""
    dev = y.device
    o = list(torch.empty((5, LAYER_W), dtype=torch.float32, device=dev).unbind(0))      # the launch writes every entry
    ws, need = _layer_ws(dev)
    rm, rv, mo, nbt = running if running is not None else (None, None, 0.0, None)
    if rm is not None and not (rm.is_cuda and rv.is_cuda and rm.dtype == rv.dtype == torch.float32 and rm.numel() == rv.numel() == c
                               and rm.is_contiguous() and rv.is_contiguous() and (nbt is None or (nbt.is_cuda and nbt.dtype == torch.int64))):
        raise ValueError("layer_colstats_fwd: running statistics must be contiguous fp32 [c] device tensors (counter: int64)")
    name = "mlqem_layer_colstats_" + _sfx(y.dtype == torch.float32)
    code = getattr(_lib.load(), name)(0, _p(y), None, None, 0, None, None, None, None, _p(gamma), _p(beta), float(eps), 0, 0.0, 0,
                                      None, n, c, *[_p(t) for t in o], _p(rm), _p(rv), float(mo), _p(nbt), _p(ws), need, _stream())
    _lib.check(code, name)
    return o


def _grad_operand(g, y, n, c):
    """The incoming gradient as (activation in y's storage | None, narrow fp32 | None, its row pitch)."""
    if g.dtype == y.dtype and _is_act(g):
        return g, None, 0
    if g.dtype != torch.float32:
        raise ValueError("the incoming gradient must be an activation in the pipeline's storage or fp32 [N, c]")
    return None, g, (_mat(g, "g") if n > 1 else c)


def layer_colstats_bwd(g, y, scale, shift, mean, invstd, gamma, relu, drop_p, seed, n, c):
    """(dbeta, dgamma, gs, k1, k2) of a BatchNorm block from the incoming gradient ``g`` (an activation or fp32 [N, c]) and ``y``."""
    dev = y.device
    o = list(torch.empty((5, LAYER_W), dtype=torch.float32, device=dev).unbind(0))      # the launch writes every entry
    ws, need = _layer_ws(dev)
    ga, g32, ld = _grad_operand(g, y, n, c)
    name = "mlqem_layer_colstats_" + _sfx(y.dtype == torch.float32)
    code = getattr(_lib.load(), name)(1, _p(y), _p(ga), _p(g32), ld, _p(scale), _p(shift), _p(mean), _p(invstd), _p(gamma), None,
                                      0.0, 1 if relu else 0, *_seed_args(drop_p, seed), n, c, *[_p(t) for t in o], None, None, 0.0,
                                      None, _p(ws), need, _stream())
    _lib.check(code, name)
    return o


def layer_act_bf16(y, scale, shift, n, c, relu=True, drop_p=0.0, seed=0, res=None):
    """drop(relu(y scale + shift)) (+ res) as a new activation in y's storage."""
    f32 = y.dtype == torch.float32
    out = _act(n, y.device, f32)
    name = "mlqem_layer_pointwise_" + _sfx(f32)
    code = getattr(_lib.load(), name)(0, _p(y), None, None, 0, _p(res), _p(scale), _p(shift), None, None, None, None, None,
                                      1 if relu else 0, *_seed_args(drop_p, seed), _p(out), n, c, _stream())
    _lib.check(code, name)
    return out


def layer_bwd_apply_bf16(g, y, scale, shift, mean, invstd, gs, k1, k2, n, c, relu=True, drop_p=0.0, seed=0):
    """dy = gs (gu - k1 - xhat k2) as an activation in y's storage (gs = 1, k1 = k2 = 0: dy = gu, a block without BatchNorm)."""
    f32 = y.dtype == torch.float32
    out = _act(n, y.device, f32)
    ga, g32, ld = _grad_operand(g, y, n, c)
    name = "mlqem_layer_pointwise_" + _sfx(f32)
    code = getattr(_lib.load(), name)(1, _p(y), _p(ga), _p(g32), ld, None, _p(scale), _p(shift), _p(mean), _p(invstd), _p(gs),
                                      _p(k1), _p(k2), 1 if relu else 0, *_seed_args(drop_p, seed), _p(out), n, c, _stream())
    _lib.check(code, name)
    return out


def layer_wgrad_bf16(dy, x, u, k):
    """(gw [u, k], gb [u]) = (dy^T x, sum dy) with dy a bf16 activation and x fp32 [N, k] (padded rows) or a bf16 activation."""
    n = dy.shape[0] if x.dtype == torch.bfloat16 else x.shape[0]
    x_bf16 = x.dtype == torch.bfloat16
    ldx = LAYER_W if x_bf16 else (_mat(x, "x") if n > 1 else (k + 3) // 4 * 4)
    gw = torch.empty((u, k), dtype=torch.float32, device=dy.device)
    gb = torch.empty(u, dtype=torch.float32, device=dy.device)
    ws, need = _layer_ws(dy.device)
    code = _lib.load().mlqem_layer_wgrad_bf16(_p(dy), _p(x), 1 if x_bf16 else 0, ldx, _p(gw), _p(gb), n, k, u, _p(ws), need, _stream())
    _lib.check(code, "mlqem_layer_wgrad_bf16")
    return gw, gb


def layer_wgrad_f32(dy, x, u, k):
    """The fp32-storage twin: dy an fp32 activation, x fp32 [N, >= k] (padded rows or an activation)."""
    n = x.shape[0]
    ldx = _mat(x, "x") if n > 1 else (x.shape[1] + 3) // 4 * 4
    gw = torch.empty((u, k), dtype=torch.float32, device=dy.device)
    gb = torch.empty(u, dtype=torch.float32, device=dy.device)
    ws, need = _layer_ws(dy.device)
    code = _lib.load().mlqem_layer_wgrad_f32(_p(dy), _p(x), ldx, _p(gw), _p(gb), n, k, u, _p(ws), need, _stream())
    _lib.check(code, "mlqem_layer_wgrad_f32")
    return gw, gb


def layer_rowdot_bf16(h, w, b, n):
    o, c = w.shape
    out = torch.empty((n, o), dtype=torch.float32, device=h.device)
    name = "mlqem_layer_rowdot_" + _sfx(h.dtype == torch.float32)
    code = getattr(_lib.load(), name)(_p(h), _p(w), _p(b), _p(out), o, n, c, o, _stream())
    _lib.check(code, name)
    return out


def layer_rowdot_bwd_bf16(g, h, w, n, gate_scale=0.0):
    """(gh activation in h's storage, gw [O, C], gb [O]) of out = h w^T + b.  ``gate_scale`` > 0: gh comes out gated by h > 0 and
    scaled -- with h = dropout(relu(u)) that is the gradient at u (gate_scale = 1 / (1 - p))."""
    o, c = w.shape
    g = rowmajor(g)
    f32 = h.dtype == torch.float32
    gh = _act(n, h.device, f32)
    gw = torch.empty((o, c), dtype=torch.float32, device=h.device)
    gb = torch.empty(o, dtype=torch.float32, device=h.device)
    ws, need = _layer_ws(h.device)
    name = "mlqem_layer_rowdot_bwd_" + _sfx(f32)
    code = getattr(_lib.load(), name)(_p(g), int(g.stride(0)) if n > 1 else o, _p(h), _p(w), _p(gh), float(gate_scale), _p(gw), _p(gb), n, c, o,
                                      _p(ws), need, _stream())
    _lib.check(code, name)
    return gh, gw, gb


_pool_ws = {}   # (device, stream, bytes) -> partial-sum workspace of the pooling kernels (per stream, like _wgrad_ws)


def linear_bwd_fused(gy, x, w, gb_src=None, gate_scale=None):
    """(gx, gw, gb, gx_colsum) of a narrow hidden layer in one pass (mlqem_linear_bwd_fused_f32): gx = gate(x) * (gy @ w),
    gw = gy.T @ x, gb = gb_src.sum(0) (gb_src defaults to gy), gx_colsum = gx.sum(0) (the bias gradient of the layer below when
    an aggregation sits between the two).  All row operands in the padded layout, I, O <= 12."""
    n, o = gy.shape
    i = x.shape[1]
    if tuple(w.shape) != (o, i) or not w.is_contiguous() or x.shape[0] != n or (gb_src is not None and gb_src.shape != gy.shape):
        raise ValueError("linear_bwd_fused: shape mismatch")
    gx = padded_empty(n, i, gy.device)
    gw2 = torch.empty((25, i), dtype=torch.float32, device=gy.device)
    gb2 = torch.empty(25, dtype=torch.float32, device=gy.device)
    lib = _lib.load()
    need = lib.mlqem_linear_wgrad_workspace_bytes(i, 25)
    ws = _wgrad_workspace(gy.device, need)
    if not _fused_bwd_ok(gy, x, *([gb_src] if gb_src is not None else [])):
        raise ValueError("linear_bwd_fused: operands must be 2-D fp32 matrices of <= 12 columns in the padded row layout")
    ld = lambda t: int(t.stride(0))          # padded rows: the stride is meaningful (and a multiple of 4) for one row too
    code = lib.mlqem_linear_bwd_fused_f32(_p(gy), ld(gy), _p(gb_src), ld(gb_src) if gb_src is not None else 0, _p(x), ld(x), _p(w),
                                          0 if gate_scale is None else 1, float(gate_scale or 1.0), _p(gx), ld(gx), _p(gw2),
                                          _p(gb2), n, i, o, _p(ws), need, _stream())
    _lib.check(code, "mlqem_linear_bwd_fused_f32")
    return gx, gw2[:o], gb2[12:12 + o], gw2[24]


def _fused_bwd_ok(*mats):
    return all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] <= 12 and t.stride(1) == 1 and t.stride(0) % 4 == 0
               and t.stride(0) >= (t.shape[1] + 3) // 4 * 4 and t.data_ptr() % 16 == 0 for t in mats)


def _head_desc(terms, biases):
    """terms: [(P [B, C], w [C] or [1, C], column)], biases: per output column a 1-element tensor or None."""
    import ctypes as _ct

    d = _lib.HeadDesc()
    d.n_terms, d.n_cols = len(terms), len(biases)
    if not 1 <= d.n_terms <= _lib.MAX_HEAD_TERMS or not 1 <= d.n_cols <= _lib.MAX_HEAD_TERMS:
        raise ValueError("pooled head: 1..8 terms and columns")
    b, c = terms[0][0].shape
    for t, (pm, w, col) in enumerate(terms):
        if tuple(pm.shape) != (b, c) or w.numel() != c or not w.is_contiguous() or not w.is_cuda or w.dtype != torch.float32:
            raise ValueError("pooled head: every term needs P [B, C] and a contiguous fp32 cuda W of C entries")
        _mat(pm, f"P[{t}]")
        d.P[t], d.ldp[t], d.W[t], d.col[t] = pm.data_ptr(), (int(pm.stride(0)) if b > 1 else max(c, 1)), w.data_ptr(), int(col)
    for k, bias in enumerate(biases):
        d.bias[k] = None if bias is None else bias.data_ptr()
    return d, b, c


def pooled_head(terms, biases):
    """out[b, col] = bias[col] + sum over the terms of that column of P_t[b, :] . w_t  -> [B, n_cols] (one launch)."""
    import ctypes as _ct

    d, b, c = _head_desc(terms, biases)
    out = torch.empty((b, d.n_cols), dtype=torch.float32, device=terms[0][0].device)
    code = _lib.load().mlqem_pooled_head_f32(_ct.addressof(d), b, c, _p(out), d.n_cols, _stream())
    _lib.check(code, "mlqem_pooled_head_f32")
    return out


def pooled_head_bwd(terms, biases, gout):
    """([gP_t [B, C] padded], gW [n_terms, C], gb [n_cols]) for ``pooled_head`` (two launches)."""
    import ctypes as _ct

    d, b, c = _head_desc(terms, biases)
    gout = rowmajor(gout)
    if tuple(gout.shape) != (b, d.n_cols):
        raise ValueError("pooled_head_bwd: gout must be [B, n_cols]")
    dev = gout.device
    gps = [padded_empty(b, c, dev) for _ in terms]
    gw = torch.empty((len(terms), c), dtype=torch.float32, device=dev)
    gb = torch.empty(d.n_cols, dtype=torch.float32, device=dev)
    ptrs = (_ct.c_void_p * _lib.MAX_HEAD_TERMS)(*[g.data_ptr() for g in gps])
    lds = (_ct.c_int64 * _lib.MAX_HEAD_TERMS)(*[(int(g.stride(0)) if b > 1 else (c + 3) // 4 * 4) for g in gps])
    code = _lib.load().mlqem_pooled_head_bwd_f32(_ct.addressof(d), _p(gout), _mat(gout, "gout"), b, c, ptrs, lds, _p(gw), _p(gb),
                                                 _stream())
    _lib.check(code, "mlqem_pooled_head_bwd_f32")
    return gps, gw, gb


def segment_pool(x, graph_ptr, num_graphs, weights=None, mean=True, wmean=False):
    """(mean, wmean): mean[g] = (1/n_g) sum_{r in g} x[r], wmean[g] = (1/n_g) sum_r weights[r] x[r]; either may be skipped
    (None is returned in its place).  Outputs are [B, C] in the padded row layout."""
    x = rowmajor(x)
    n, c = x.shape
    if not (mean or wmean):
        raise ValueError("segment_pool: nothing to compute")
    _vec(graph_ptr, "graph_ptr", num_graphs + 1, torch.int32)
    if wmean:
        _vec(weights, "weights", n)
    lib = _lib.load()
    need = lib.mlqem_segment_pool_workspace_bytes(n, num_graphs, c)
    key = (x.device, _stream(), need)
    ws = _pool_ws.get(key)
    if ws is None:
        ws = _pool_ws[key] = torch.empty(max(need, 1), dtype=torch.uint8, device=x.device)
    o0 = padded_empty(num_graphs, c, x.device) if mean else None
    o1 = padded_empty(num_graphs, c, x.device) if wmean else None
    ld = lambda t: 0 if t is None else (int(t.stride(0)) if num_graphs > 1 else (c + 3) // 4 * 4)
    code = lib.mlqem_segment_pool_f32(_p(x), _mat(x, "x"), _p(weights) if wmean else None, _p(graph_ptr), n, num_graphs, c,
                                      _p(o0), ld(o0), _p(o1), ld(o1), _p(ws), need, _stream())
    _lib.check(code, "mlqem_segment_pool_f32")
    return o0, o1


def pool_gate_unpack(bits, n, c):
    """[n, ceil(c / 4)] int32 on the host: bit v of entry (row, slice) = (activation[row, 4 slice + v] > 0), decoded from the gate
    buffer a pooled aggregation leaves (include/mlqem_hip.h, mlqem_csr_aggregate_pool_f32: tile records, then 32 ballot words per
    tile).  For tests and debugging; the backward kernel reads the buffer as it is."""
    import numpy as np

    cv = (c + 3) // 4
    rows = max(1, 512 // cv)
    tiles = (max(n, 1) + rows - 1) // rows
    raw = bits.cpu().numpy()
    words = raw[16 * tiles:16 * tiles + 256 * tiles].view(np.uint64).reshape(tiles, 2, 4, 4)              # [tile][k][wave][v]
    item = np.arange(n * cv, dtype=np.int64)
    tile, local = item // (rows * cv), item % (rows * cv)
    k, wave, lane = local >> 8, (local >> 6) & 3, (local & 63).astype(np.uint64)
    out = np.zeros(n * cv, dtype=np.int32)
    for v in range(4):
        out |= ((words[tile, k, wave, v] >> lane) & np.uint64(1)).astype(np.int32) << v
    return out.reshape(n, cv)


def pool_node_gates(bits, n, c):
    """[n, ceil(c / 4)] int32 like ``pool_gate_unpack``, decoded from the per-node section of the gate buffer (c <= 16: 16 bits a node
    after the tile records and the ballots; what ``PooledGrad.aggregate`` gathers)."""
    import numpy as np

    cv = (c + 3) // 4
    rows = max(1, 512 // cv)
    tiles = (max(n, 1) + rows - 1) // rows
    per = bits.cpu().numpy()[(16 + 256) * tiles:].view(np.uint16)[:n].astype(np.int32)
    return np.stack([(per >> (4 * sl)) & 15 for sl in range(cv)], axis=1)


def segment_pool_bwd(g_mean, g_wmean, graph_ptr, num_nodes, weights=None, gate=None, gate_scale=1.0, out=None, gate_bits=None):
    """gx[r] = (g_mean[g] + weights[r] g_wmean[g]) / n_g (either gradient may be None), optionally gated by gate > 0."""
    ref = g_mean if g_mean is not None else g_wmean
    if ref is None:
        raise ValueError("segment_pool_bwd: no gradient given")
    b, c = ref.shape

    def padded(t):   # the [B, C] gradients are tiny: give them 16-byte rows so the kernel keeps one vector width
        if t is None:
            return None
        if tuple(t.shape) != (b, c):
            raise ValueError("segment_pool_bwd: gradients must share one shape")
        ok = t.is_cuda and t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and \
            t.stride(0) >= (c + 3) // 4 * 4 and t.data_ptr() % 16 == 0
        return t if ok else padded_copy(t)

    g_mean, g_wmean = padded(g_mean), padded(g_wmean)
    _vec(graph_ptr, "graph_ptr", b + 1, torch.int32)
    if g_wmean is not None:
        _vec(weights, "weights", num_nodes)
    gx = padded_empty(num_nodes, c, ref.device) if out is None else out
    if tuple(gx.shape) != (num_nodes, c):
        raise ValueError("segment_pool_bwd: bad out shape")
    gp, gld = _gate(gate, num_nodes, c, False)
    if gate_bits is not None:      # the gate as sign bits (csr_aggregate(..., pool=) leaves them: tile records + per-wave ballots)
        want = _lib.load().mlqem_csr_aggregate_pool_gate_bytes(num_nodes, c)
        if gate is not None or gate_bits.dtype != torch.uint8 or not gate_bits.is_cuda or gate_bits.numel() != want \
                or not gate_bits.is_contiguous():
            raise ValueError(f"segment_pool_bwd: gate_bits must be the contiguous uint8 cuda tensor of {want} bytes the pooled "
                             "aggregation of the same rows and columns left (and no gate)")
    ld = lambda t: 0 if t is None else (int(t.stride(0)) if b > 1 else (c + 3) // 4 * 4)
    code = _lib.load().mlqem_segment_pool_bwd_f32(_p(g_mean), ld(g_mean), _p(g_wmean), ld(g_wmean),
                                                  _p(weights) if g_wmean is not None else None, _p(graph_ptr), num_nodes, b, c,
                                                  gp, gld, float(gate_scale), _p(gate_bits), _p(gx), _mat(gx, "gx"), _stream())
    _lib.check(code, "mlqem_segment_pool_bwd_f32")
    return gx


def pooled_grad_supported(c):
    """Does ``pooled_grad_aggregate`` serve rows of ``c`` channels (the per-node gate of the pooled forward: c <= 16)?"""
    return bool(_lib.load().mlqem_pooled_grad_aggregate_supported(int(c)))


class PooledGrad:
    """The gradient of a Family A branch's last hidden activation, NOT written out: what ``segment_pool_bwd(gate_bits=)`` would
    compute it from.  A conv layer's backward hands it to ``aggregate`` -- its first transposed aggregation, which computes every
    source row from (gate bits, pool weight, the graph's two gradient rows) instead of gathering it, and leaves the matrix itself
    for the layer's dense consumers (csrc/pooled_grad.hip)."""

    def __init__(self, g_mean, g_wmean, graph_ptr, num_nodes, weights, gate_scale, gate_bits):
        self.g_mean, self.g_wmean, self.graph_ptr, self.num_nodes = g_mean, g_wmean, graph_ptr, num_nodes
        self.weights, self.gate_scale, self.gate_bits = weights, gate_scale, gate_bits

    def materialise(self):
        return segment_pool_bwd(self.g_mean, self.g_wmean, self.graph_ptr, self.num_nodes, weights=self.weights,
                                gate_scale=self.gate_scale, gate_bits=self.gate_bits)

    def _operands(self):
        ref = self.g_wmean if self.g_wmean is not None else self.g_mean
        b, c = ref.shape
        n = self.num_nodes

        def padded(t):
            if t is None:
                return None
            ok = t.is_cuda and t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and \
                t.stride(0) >= (c + 3) // 4 * 4 and t.data_ptr() % 16 == 0
            return t if ok else padded_copy(t)

        gm, gw = padded(self.g_mean), padded(self.g_wmean)
        _vec(self.graph_ptr, "graph_ptr", b + 1, torch.int32)
        _vec(self.weights, "weights", n)
        want = _lib.load().mlqem_csr_aggregate_pool_gate_bytes(n, c)
        bits = self.gate_bits
        if bits.dtype != torch.uint8 or not bits.is_cuda or bits.numel() != want or not bits.is_contiguous():
            raise ValueError(f"PooledGrad: gate_bits must be the {want}-byte buffer the pooled aggregation left")
        ld = lambda t: 0 if t is None else (int(t.stride(0)) if b > 1 else (c + 3) // 4 * 4)
        return ref, b, c, n, gm, ld(gm), gw, ld(gw)

    def colsum(self):
        """sum_j g[j, :] -> [C], from the same operands (six bytes a node): the bias gradient of a layer whose ``aggregate`` did not
        write g."""
        ref, b, c, n, gm, ldm, gw, ldw = self._operands()
        lib = _lib.load()
        part = torch.empty((lib.mlqem_pooled_grad_colsum_groups(n), (c + 3) // 4 * 4), dtype=torch.float32, device=ref.device)
        code = lib.mlqem_pooled_grad_colsum_f32(_p(self.gate_bits), _p(self.weights), _p(gm), ldm, _p(gw), ldw, _p(self.graph_ptr), b,
                                                float(self.gate_scale), n, c, _p(part), _stream())
        _lib.check(code, "mlqem_pooled_grad_colsum_f32")
        return part.sum(0)[:c]

    def aggregate(self, ptr, idx, ell, cscale, rscale=None, dself=None, alpha=1.0, want_g=True):
        """(alpha (rscale * sum_e cscale[idx[e]] g[idx[e]] + dself g), g or None)."""
        ref, b, c, n, gm, ldm, gw, ldw = self._operands()
        _vec(ptr, "ptr", n + 1, torch.int32)
        _vec(idx, "idx", 0, torch.int32)
        _ell(ell, n)
        for nm, v in (("cscale", cscale), ("rscale", rscale), ("dself", dself)):
            _vec(v, nm, n)
        if cscale is None:
            raise ValueError("PooledGrad.aggregate: cscale is required")
        out, g = padded_empty(n, c, ref.device), (padded_empty(n, c, ref.device) if want_g else None)
        code = _lib.load().mlqem_pooled_grad_aggregate_f32(_p(self.gate_bits), _p(self.weights), _p(cscale), _p(gm), ldm, _p(gw), ldw, _p(self.graph_ptr), b,
                                                           float(self.gate_scale), _p(ptr), _p(idx), _p(ell), _p(rscale), _p(dself),
                                                           float(alpha), _p(out), _mat(out, "out"), _p(g), _mat(g, "g") if want_g else 0,
                                                           n, c, _stream())
        _lib.check(code, "mlqem_pooled_grad_aggregate_f32")
        return out, g


def pooled_grad_aggregate(pooled, ptr, idx, ell, cscale, **kw):
    """``pooled.aggregate(...)`` as a module-level call (what the layers use: tools that wrap the entry points of this module by name --
    scripts/kernel_roofline.py -- see it)."""
    return pooled.aggregate(ptr, idx, ell, cscale, **kw)


def pooled_grad_colsum(pooled):
    return pooled.colsum()


def segment_mean(x, graph_ptr, num_graphs):
    return segment_pool(x, graph_ptr, num_graphs)[0]


def segment_mean_bwd(g, graph_ptr, num_nodes, out=None):
    return segment_pool_bwd(rowmajor(g), None, graph_ptr, num_nodes, out=out)


class CsrArrays(tuple):
    """(in_ptr, in_src, out_ptr, out_dst, loops) -- unpacks like the 5-tuple it always was -- plus ``.out_eid``."""

    def __new__(cls, in_ptr, in_src, out_ptr, out_dst, loops, out_eid):
        self = super().__new__(cls, (in_ptr, in_src, out_ptr, out_dst, loops))
        self.out_eid = out_eid
        return self


def csr_build(edge_index: torch.Tensor, num_nodes: int):
    """[2,E] int64 cuda edge list -> (in_ptr, in_src, out_ptr, out_dst, loops) int32, with ``.out_eid``."""
    if not edge_index.is_cuda or edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("edge_index must be a [2,E] int64 cuda tensor")
    edge_index = edge_index.contiguous()
    e = int(edge_index.shape[1])
    dev = edge_index.device
    lib = _lib.load()
    mk = lambda n: torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    in_ptr, out_ptr, loops = mk(num_nodes + 1), mk(num_nodes + 1), mk(num_nodes)
    in_src, out_dst, out_eid = mk(e), mk(e), mk(e)
    need = lib.mlqem_csr_build_workspace_bytes(num_nodes, e)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    code = lib.mlqem_csr_build(_p(edge_index), e, num_nodes, _p(in_ptr), _p(in_src), _p(out_ptr), _p(out_dst),
                               _p(out_eid), _p(loops), _p(ws), need, _stream())
    _lib.check(code, "mlqem_csr_build")
    return CsrArrays(in_ptr, in_src, out_ptr, out_dst, loops, out_eid)


def graph_norms(in_ptr, out_ptr, loops, num_nodes):
    dev = in_ptr.device
    mk = lambda: torch.empty(max(num_nodes, 1), dtype=torch.float32, device=dev)
    gcn, sage, cheb = mk(), mk(), mk()
    code = _lib.load().mlqem_graph_norms(_p(in_ptr), _p(out_ptr), _p(loops), num_nodes, _p(gcn), _p(sage), _p(cheb),
                                         _stream())
    _lib.check(code, "mlqem_graph_norms")
    return gcn, sage, cheb


# ------------------------------------------------------------------------------------------------- Family B
def transformer_attention(qkvs, in_ptr, in_src, loops, heads, channels):
    n = qkvs.shape[0]
    hc = heads * channels
    if qkvs.shape[1] != 4 * hc:
        raise ValueError(f"qkvs must be [N, {4 * hc}] (query|key|value|skip), got {tuple(qkvs.shape)}")
    ld = _mat(qkvs, "qkvs")
    _vec(in_ptr, "in_ptr", n + 1, torch.int32)
    _vec(loops, "loops", n, torch.int32)
    out = padded_empty(n, hc, qkvs.device)
    code = _lib.load().mlqem_transformer_attention_f32(_p(qkvs), ld, _p(in_ptr), _p(in_src), _p(loops), n, heads,
                                                       channels, _p(out), _mat(out, "out"), _stream())
    _lib.check(code, "mlqem_transformer_attention_f32")
    return out


def csr_softmax_aggregate(x, in_ptr, in_src, a_dst, c_src, negative_slope):
    n, c = x.shape
    ldx = _mat(x, "x")
    _vec(in_ptr, "in_ptr", n + 1, torch.int32)
    _vec(a_dst, "a_dst", n)
    _vec(c_src, "c_src", n)
    out = padded_empty(n, c, x.device)
    code = _lib.load().mlqem_csr_softmax_aggregate_f32(_p(x), ldx, _p(in_ptr), _p(in_src), _p(a_dst), _p(c_src),
                                                       float(negative_slope), n, c, _p(out), _mat(out, "out"),
                                                       _stream())
    _lib.check(code, "mlqem_csr_softmax_aggregate_f32")
    return out


def leconv_fitness(pqr, in_ptr, in_src, long_rows=False):
    """``long_rows``: a graph of long rows (a coarsened graph): a 16-lane group per row instead of a thread."""
    n = pqr.shape[0]
    if pqr.shape[1] != 3 or not pqr.is_contiguous() or pqr.dtype != torch.float32 or not pqr.is_cuda:
        raise ValueError("pqr must be a contiguous [N,3] fp32 cuda tensor")
    _vec(in_ptr, "in_ptr", n + 1, torch.int32)
    f = torch.empty(max(n, 1), dtype=torch.float32, device=pqr.device)[:n]
    code = _lib.load().mlqem_leconv_fitness_f32(_p(pqr), _p(in_ptr), _p(in_src), n, _p(f), 1 if long_rows else 0, _stream())
    _lib.check(code, "mlqem_leconv_fitness_f32")
    return f


def gather_scale_rows(x, perm, scale=None):
    k, c = perm.shape[0], x.shape[1]
    _vec(perm, "perm", k, torch.int32)
    _vec(scale, "scale", x.shape[0])
    out = padded_empty(k, c, x.device)
    code = _lib.load().mlqem_gather_scale_rows_f32(_p(x), _mat(x, "x"), _p(perm), _p(scale), k, c, _p(out),
                                                   _mat(out, "out"), _stream())
    _lib.check(code, "mlqem_gather_scale_rows_f32")
    return out


def asap_scores_fused(x, in_ptr, in_src, w_comp, b_comp, att_x, w3, b3, negative_slope):
    """(xmax, a_dst, c_src, x_new, pqr) of ASAPooling's forward in one pass over a graph of short rows (<= 64 channels)."""
    n, c = x.shape
    dev = x.device
    xmax, xnew = padded_empty(n, c, dev), padded_empty(n, c, dev)
    a_dst = torch.empty(max(n, 1), dtype=torch.float32, device=dev)[:n]
    c_src = torch.empty(max(n, 1), dtype=torch.float32, device=dev)[:n]
    pqr = torch.empty((n, 3), dtype=torch.float32, device=dev)
    code = _lib.load().mlqem_asap_scores_fused_f32(_p(x), _mat(x, "x"), _p(in_ptr), _p(in_src), _p(w_comp.contiguous()), _p(b_comp),
                                                   _p(att_x.contiguous()), _p(w3.contiguous()), _p(b3), float(negative_slope), n, c,
                                                   _p(xmax), _mat(xmax, "xmax"), _p(a_dst), _p(c_src), _p(xnew), _mat(xnew, "xnew"), _p(pqr),
                                                   _stream())
    _lib.check(code, "mlqem_asap_scores_fused_f32")
    return xmax, a_dst, c_src, xnew, pqr


def pad_head_rows(w, b, groups, channels, pitch):
    """(w, b) with every group of ``channels`` rows spread to ``pitch`` rows (zero rows between), in one launch."""
    cols = w.shape[1]
    wp = torch.empty((groups * pitch, cols), dtype=torch.float32, device=w.device)
    bp = torch.empty(groups * pitch, dtype=torch.float32, device=w.device) if b is not None else None
    code = _lib.load().mlqem_pad_head_rows_f32(_p(w.contiguous()), _p(b), groups, channels, pitch, cols, _p(wp), _p(bp), _stream())
    _lib.check(code, "mlqem_pad_head_rows_f32")
    return wp, bp


def pad_head_rows_parts(ws, bs, groups_per_part, channels, pitch):
    """(w, b) of ``pad_head_rows(torch.cat(ws), torch.cat(bs), len(ws) * groups_per_part, channels, pitch)`` in ONE launch, from the
    separate matrices (``pitch == channels``: the concatenation itself)."""
    import ctypes as _ct

    parts, cols = len(ws), ws[0].shape[1]
    if not 1 <= parts <= 4 or any(tuple(w.shape) != (groups_per_part * channels, cols) or w.dtype != torch.float32 or not w.is_cuda for w in ws):
        raise ValueError("pad_head_rows_parts: one to four fp32 cuda matrices of one shape [groups_per_part * channels, cols]")
    ws = [w.contiguous() for w in ws]
    bs = [None if b is None else b.contiguous() for b in bs]
    dev = ws[0].device
    wp = torch.empty((parts * groups_per_part * pitch, cols), dtype=torch.float32, device=dev)
    has_b = any(b is not None for b in bs)
    bp = torch.empty(parts * groups_per_part * pitch, dtype=torch.float32, device=dev) if has_b else None
    w_arr = (_ct.c_void_p * 4)(*[ws[k].data_ptr() if k < parts else None for k in range(4)])
    b_arr = (_ct.c_void_p * 4)(*[(bs[k].data_ptr() if bs[k] is not None else None) if k < parts else None for k in range(4)])
    code = _lib.load().mlqem_pad_head_rows_parts_f32(w_arr, b_arr, parts, groups_per_part, channels, pitch, cols, _p(wp), _p(bp), _stream())
    _lib.check(code, "mlqem_pad_head_rows_parts_f32")
    return wp, bp


def unpad_head_rows(gwp, gbp, groups, channels, pitch):
    """The real rows of gradients in the padded layout of ``pad_head_rows``: (gw [groups * channels, cols], gb), one launch."""
    cols = gwp.shape[1]
    gw = torch.empty((groups * channels, cols), dtype=torch.float32, device=gwp.device)
    gb = torch.empty(groups * channels, dtype=torch.float32, device=gwp.device) if gbp is not None else None
    code = _lib.load().mlqem_unpad_head_rows_f32(_p(gwp.contiguous()), _p(gbp), groups, channels, pitch, cols, _p(gw), _p(gb), _stream())
    _lib.check(code, "mlqem_unpad_head_rows_f32")
    return gw, gb


def asap_compose(lin_w, lin_b, att_w, att_b, l1_w, l1_b, l2_w, l3_w, l3_b):
    """(w_comp [1, D], b_comp [1], att_q [1, D], att_x [1, D], w3 [3, D], b3 [3]) of ASAPooling's parameters in one launch."""
    d = lin_w.shape[0]
    dev = lin_w.device
    ts = [t.contiguous() for t in (lin_w, lin_b, att_w, att_b, l1_w, l1_b, l2_w, l3_w, l3_b)]
    buf = torch.empty(6 * d + 4, dtype=torch.float32, device=dev)
    w_comp, att_q, att_x, w3 = buf[:d].view(1, d), buf[d:2 * d].view(1, d), buf[2 * d:3 * d].view(1, d), buf[3 * d:6 * d].view(3, d)
    b_comp, b3 = buf[6 * d:6 * d + 1], buf[6 * d + 1:6 * d + 4]
    code = _lib.load().mlqem_asap_compose_f32(*[_p(t) for t in ts], d, _p(w_comp), _p(b_comp), _p(att_q), _p(att_x), _p(w3), _p(b3), _stream())
    _lib.check(code, "mlqem_asap_compose_f32")
    return w_comp, b_comp, att_q, att_x, w3, b3


def asap_compose_bwd(g_w_comp, g_att_b, lin_w, lin_b, att_w, g_att_x):
    """(g_lin_w [D, D], g_lin_b [D], g_att_w [1, 2 D]): the chain rule of ``asap_compose`` in one launch."""
    d = lin_w.shape[0]
    dev = lin_w.device
    buf = torch.empty(d * d + 3 * d, dtype=torch.float32, device=dev)
    g_lin_w, g_lin_b, g_att_w = buf[:d * d].view(d, d), buf[d * d:d * d + d], buf[d * d + d:].view(1, 2 * d)
    code = _lib.load().mlqem_asap_compose_bwd_f32(_p(g_w_comp.contiguous()), _p(g_att_b), _p(lin_w.contiguous()), _p(lin_b.contiguous()),
                                                  _p(att_w.contiguous()), _p(g_att_x.contiguous()), d, _p(g_lin_w), _p(g_lin_b), _p(g_att_w),
                                                  _stream())
    _lib.check(code, "mlqem_asap_compose_bwd_f32")
    return g_lin_w, g_lin_b, g_att_w


def gather_rows(mats, sel):
    """[m[sel] for m in mats] for up to four contiguous fp32 matrices with one leading dimension G, in ONE launch
    (mlqem_gather_rows_f32): the per-graph inputs of a batch (labels, noisy values, depths, observables).  ``sel``: int32 [B]."""
    import ctypes as _ct

    if not 1 <= len(mats) <= 4:
        raise ValueError("gather_rows: one to four matrices")
    b = int(sel.shape[0])
    _vec(sel, "sel", b, torch.int32)
    outs, widths = [], []
    for m in mats:
        if not m.is_cuda or m.dtype != torch.float32 or not m.is_contiguous():
            raise ValueError("gather_rows: contiguous fp32 cuda tensors")
        w = int(m[0].numel()) if m.shape[0] else int(torch.Size(m.shape[1:]).numel())
        widths.append(w)
        outs.append(torch.empty((b,) + tuple(m.shape[1:]), dtype=torch.float32, device=m.device))
    n = len(mats)
    src = (_ct.c_void_p * 4)(*[mats[i].data_ptr() if i < n else None for i in range(4)])
    dst = (_ct.c_void_p * 4)(*[outs[i].data_ptr() if i < n else None for i in range(4)])
    wid = (_ct.c_int64 * 4)(*[widths[i] if i < n else 0 for i in range(4)])
    code = _lib.load().mlqem_gather_rows_f32(n, src, wid, _p(sel), b, dst, _stream())
    _lib.check(code, "mlqem_gather_rows_f32")
    return outs


def pool_keep_ptr(graph_ptr, num_graphs, ratio):
    """Boundaries [B + 1] int32 of the pooled batch, computed ON THE DEVICE from the batch's boundaries: k_g = ceil(float32(n_g) * ratio)
    per graph (mlqem_pool_keep_ptr).  No host value but B enters: the call can be captured and replayed for other size sequences."""
    _vec(graph_ptr, "graph_ptr", num_graphs + 1, torch.int32)
    out = torch.empty(num_graphs + 1, dtype=torch.int32, device=graph_ptr.device)
    code = _lib.load().mlqem_pool_keep_ptr(_p(graph_ptr), num_graphs, float(ratio), _p(out), _stream())
    _lib.check(code, "mlqem_pool_keep_ptr")
    return out


def segment_topk(fitness, graph_ptr, new_graph_ptr, num_nodes, num_graphs, k_total, max_graph_nodes=0, with_slot=False):
    """perm [k_total] -- and, ``with_slot``, (perm, slot) with ``asap_slot_map``'s slot[N] from the same launches."""
    _vec(fitness, "fitness", num_nodes)
    _vec(graph_ptr, "graph_ptr", num_graphs + 1, torch.int32)
    _vec(new_graph_ptr, "new_graph_ptr", num_graphs + 1, torch.int32)
    lib = _lib.load()
    need = lib.mlqem_segment_topk_workspace_bytes(num_nodes, num_graphs)
    ws = torch.empty(need, dtype=torch.uint8, device=fitness.device)
    perm = torch.empty(max(k_total, 1), dtype=torch.int32, device=fitness.device)[:k_total]
    slot = torch.empty(max(num_nodes, 1), dtype=torch.int32, device=fitness.device) if with_slot else None
    code = lib.mlqem_segment_topk(_p(fitness), _p(graph_ptr), _p(new_graph_ptr), num_nodes, num_graphs, k_total, int(max_graph_nodes),
                                  _p(perm), _p(slot), _p(ws), need, _stream())
    _lib.check(code, "mlqem_segment_topk")
    return (perm, slot) if with_slot else perm


def _sort_unique(keys, total):
    dev = keys.device
    lib = _lib.load()
    uniq = torch.empty(total, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    need = lib.mlqem_sort_unique_u64_workspace_bytes(total)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    code = lib.mlqem_sort_unique_u64(_p(keys), total, _p(uniq), _p(count), _p(ws), need, _stream())
    _lib.check(code, "mlqem_sort_unique_u64")
    return uniq, int(count.item())


def asap_coarsen(in_ptr, in_src, out_ptr, out_dst, perm, num_nodes, return_slot=False):
    """Edge list [2,E] int64 (cluster ids, row-major (src,dst) order, no diagonal) of the pooled graph.
    Two hops, each count -> fill -> sort-unique; four small device->host reads size the buffers."""
    dev = perm.device
    k = int(perm.shape[0])
    lib = _lib.load()
    slot = torch.empty(max(num_nodes, 1), dtype=torch.int32, device=dev)
    empty = torch.zeros((2, 0), dtype=torch.int64, device=dev)

    def scan_ws(n):
        need = lib.mlqem_asap_coarsen_workspace_bytes(n)
        return torch.empty(need, dtype=torch.uint8, device=dev), need

    offsets = torch.empty(k + 1, dtype=torch.int64, device=dev)
    ws, need = scan_ws(k)
    code = lib.mlqem_asap_hop1_count(_p(in_ptr), _p(in_src), _p(out_ptr), _p(out_dst), _p(perm), num_nodes, k, _p(slot),
                                     _p(offsets), _p(ws), need, _stream())
    _lib.check(code, "mlqem_asap_hop1_count")
    total = int(offsets[k].item())
    if total == 0:
        return (empty, slot) if return_slot else empty
    keys = torch.empty(total, dtype=torch.int64, device=dev)
    code = lib.mlqem_asap_hop1_fill(_p(in_ptr), _p(in_src), _p(out_ptr), _p(out_dst), _p(perm), _p(offsets), k, _p(keys),
                                    _stream())
    _lib.check(code, "mlqem_asap_hop1_fill")
    pairs, m = _sort_unique(keys, total)
    del keys
    offsets2 = torch.empty(m + 1, dtype=torch.int64, device=dev)
    ws, need = scan_ws(m)
    code = lib.mlqem_asap_hop2_count(_p(pairs), m, _p(out_ptr), _p(out_dst), _p(slot), _p(offsets2), _p(ws), need,
                                     _stream())
    _lib.check(code, "mlqem_asap_hop2_count")
    total2 = int(offsets2[m].item())
    if total2 == 0:
        return (empty, slot) if return_slot else empty
    keys2 = torch.empty(total2, dtype=torch.int64, device=dev)
    code = lib.mlqem_asap_hop2_fill(_p(pairs), m, _p(out_ptr), _p(out_dst), _p(slot), _p(offsets2), _p(keys2), _stream())
    _lib.check(code, "mlqem_asap_hop2_fill")
    uniq, e = _sort_unique(keys2, total2)
    del keys2
    if os.environ.get("MLQEM_ASAP_DEBUG"):
        print(f"asap_coarsen: N={num_nodes} k={k} hop1 candidates={total} distinct (p,v)={m} hop2 candidates={total2} edges={e}", flush=True)
    ei = torch.empty((2, e), dtype=torch.int64, device=dev)
    code = lib.mlqem_keys_to_edge_index(_p(uniq), e, _p(ei), _stream())
    _lib.check(code, "mlqem_keys_to_edge_index")
    return (ei, slot) if return_slot else ei


def _keep_info(keep_sizes):
    """(B, k_total, kmax, dense edge capacity) of a pooled batch from the host array of its k_g -- or from BOUNDS, a dict with
    ``b``, ``k`` (exact) and ``kmax`` (an upper bound on the largest k_g), when the per-graph sizes stay on the device (size-stable
    captured steps): sum k_g (k_g - 1) <= k (kmax - 1)."""
    import numpy as np

    if isinstance(keep_sizes, dict):
        b, k, kmax = int(keep_sizes["b"]), int(keep_sizes["k"]), int(keep_sizes["kmax"])
        return b, k, kmax, k * max(kmax - 1, 0)
    keep = np.asarray(keep_sizes, dtype=np.int64)
    b, k = int(keep.shape[0]), int(keep.sum())
    return b, k, (int(keep.max()) if b else 0), int((keep * (keep - 1)).sum())


def asap_coarsen_dense(s_in_ptr, s_in_src, s_out_ptr, s_out_dst, graph_ptr, new_graph_ptr, perm, num_nodes, keep_sizes, slot=None):
    """Pooled structure arrays (in_ptr, in_src, out_ptr, out_dst, out_eid, loops, slot) and the capacity of the edge
    arrays, with NO device->host copy (mlqem_asap_coarsen_dense).  ``keep_sizes``: host array of k_g per graph (or bounds:
    ``_keep_info``).  ``slot``: ``asap_slot_map(perm, num_nodes)`` when the caller has it already (two launches less)."""
    b, k, kmax, cap = _keep_info(keep_sizes)
    dev = perm.device
    lib = _lib.load()
    mk = lambda n: torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    ready = slot is not None
    if ready:
        _vec(slot, "slot", num_nodes, torch.int32)
    in_ptr, out_ptr, loops = mk(k + 1), mk(k + 1), mk(k)
    slot = slot if ready else mk(num_nodes)
    in_src, out_dst, out_eid = mk(cap), mk(cap), mk(cap)
    need = lib.mlqem_asap_coarsen_dense_workspace_bytes(b, k, kmax)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    code = lib.mlqem_asap_coarsen_dense(_p(s_in_ptr), _p(s_in_src), _p(s_out_ptr), _p(s_out_dst), _p(graph_ptr), _p(new_graph_ptr),
                                        _p(perm), num_nodes, k, b, kmax, _p(slot), 1 if ready else 0, _p(in_ptr), _p(in_src), _p(out_ptr),
                                        _p(out_dst), _p(out_eid), _p(loops), _p(ws), need, _stream())
    _lib.check(code, "mlqem_asap_coarsen_dense")
    return CsrArrays(in_ptr, in_src, out_ptr, out_dst, loops, out_eid), slot, cap


def asap_coarsen_rows(s_in_ptr, s_in_src, s_out_ptr, s_out_dst, graph_ptr, new_graph_ptr, perm, num_nodes, graph_sizes, keep_sizes,
                      capacity=None):
    """The same pooled structure arrays for LARGE graphs (mlqem_asap_coarsen_rows_count / _fill: one wave per cluster, bitsets
    in LDS, no sort): ONE 4-byte device->host read (the edge total) instead of the two-hop path's four reads and two
    64-bit sorts -- and NONE when the caller knows an upper bound ``capacity`` on the edge total (GraphArena.coarse_capacity):
    the edge arrays are then sized to it and the true total stays on the device (in_ptr[k]).  Returns (CsrArrays, slot,
    number of edges or the capacity)."""
    import numpy as np

    b, k, kmax, _ = _keep_info(keep_sizes)
    nmax = (int(keep_sizes["nmax"]) if isinstance(keep_sizes, dict) else int(np.asarray(graph_sizes).max())) if b else 0
    dev = perm.device
    lib = _lib.load()
    mk = lambda n: torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    slot, in_ptr, out_ptr = mk(num_nodes), mk(k + 1), mk(k + 1)
    need = lib.mlqem_asap_coarsen_rows_workspace_bytes(k, kmax)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    code = lib.mlqem_asap_coarsen_rows_count(_p(s_in_ptr), _p(s_in_src), _p(s_out_ptr), _p(s_out_dst), _p(graph_ptr),
                                             _p(new_graph_ptr), _p(perm), num_nodes, k, b, nmax, kmax, _p(slot), _p(in_ptr),
                                             _p(out_ptr), _p(ws), need, _stream())
    _lib.check(code, "mlqem_asap_coarsen_rows_count")
    if capacity is None:
        e = int(out_ptr[k].item()) if k > 0 else 0
    else:
        e = int(capacity) if k > 0 else 0
    in_src, out_dst, out_eid = mk(e), mk(e), mk(e)
    loops = torch.full((max(k, 1),), 0, dtype=torch.int32, device=dev)      # a fill kernel, not a memset node (the call may be captured)
    if e > 0:
        code = lib.mlqem_asap_coarsen_rows_fill(_p(new_graph_ptr), k, b, kmax, _p(in_ptr), _p(out_ptr), _p(in_src), _p(out_dst),
                                                _p(out_eid), _p(ws), need, _stream())
        _lib.check(code, "mlqem_asap_coarsen_rows_fill")
    return CsrArrays(in_ptr, in_src[:e], out_ptr, out_dst[:e], loops[:k], out_eid[:e]), slot, e


def asap_coarsen_lists(s_in_ptr, s_in_src, s_out_ptr, s_out_dst, graph_ptr, new_graph_ptr, perm, num_nodes, num_edges, keep_sizes,
                       capacity=None, link=True, slot=None):
    """The pooled structure arrays of ``asap_coarsen_rows`` from SORTED LISTS (mlqem_asap_coarsen_lists_*, round 4): per-node
    cluster lists built once, one walk per cluster by persistent waves, nothing dense in global memory, the twin links by binary
    search.  ``num_edges``: stored edges of the input structure (or a bound).  ``capacity``: a bound on the four list totals and on
    the edge total (GraphArena.coarse_capacity); without it the totals are read back (one 32-byte device->host copy, then the
    4-byte edge total).  Returns (CsrArrays, slot, edge capacity), or None when the candidate lists would exceed
    ``ASAP_LISTS_MAX_CAPACITY`` entries (the caller then takes another form).  ``slot``: ``asap_slot_map``'s result when the caller has
    it (ASAPooling's autograd node makes it for its backward)."""
    b, k, kmax, _ = _keep_info(keep_sizes)
    dev = perm.device
    lib = _lib.load()
    mk = lambda n: torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    slot_ready = slot is not None
    if not slot_ready:
        slot = mk(num_nodes)
    in_ptr, out_ptr = mk(k + 1), mk(k + 1)
    exact = capacity is None
    if exact:
        if k > 0:
            need = lib.mlqem_asap_coarsen_lists_workspace_bytes(num_nodes, k, 0, 0)
            ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
            totals = torch.empty(4, dtype=torch.int64, device=dev)
            code = lib.mlqem_asap_coarsen_lists_caps(_p(s_in_ptr), _p(s_in_src), _p(s_out_ptr), _p(s_out_dst), _p(new_graph_ptr), _p(perm),
                                                     num_nodes, k, b, _p(totals), _p(ws), need, _stream())
            _lib.check(code, "mlqem_asap_coarsen_lists_caps")
            cap = int(max(totals.tolist()))
        else:
            cap = 0
    else:
        cap = int(capacity) if k > 0 else 0
    if cap >= ASAP_LISTS_MAX_CAPACITY:
        # a graph whose clusters have thousands of candidates each (the coarsening of an already coarsened graph: rows of hundreds of
        # entries, three hops deep): the candidate lists would not fit 32-bit places -- the caller takes the bit-matrix form
        return None
    e = cap
    need = lib.mlqem_asap_coarsen_lists_workspace_bytes(num_nodes, k, int(num_edges), cap)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    loops = mk(k)                           # zeroed by the count pass (a kernel, not a memset node: the call may be captured)
    code = lib.mlqem_asap_coarsen_lists_count(_p(s_in_ptr), _p(s_in_src), _p(s_out_ptr), _p(s_out_dst), _p(graph_ptr), _p(new_graph_ptr),
                                              _p(perm), num_nodes, k, b, int(num_edges), kmax, cap, _p(slot), 1 if slot_ready else 0,
                                              _p(in_ptr), _p(out_ptr), _p(loops), _p(ws), need, _stream())
    _lib.check(code, "mlqem_asap_coarsen_lists_count")
    if exact and k > 0:
        e = int(out_ptr[k].item())          # the exact edge total (this path reads the device anyway)
    in_src, out_dst = mk(e), mk(e)
    out_eid = mk(e) if link else None       # link=False: no out_eid (the recomputed backward forms need none)
    if k > 0:
        # with a capacity bound nothing is read back: a list that would leave its buffer is dropped by the kernels, which OR this
        # device's STICKY flag -- read (and reset) lazily (check_overflow_flags: epoch ends, every 256 steps of step_ids,
        # MLQEM_SYNC_OPS, tests), so a truncated graph cannot pass unnoticed, in an eager step or in any replay of a captured one
        flag = _overflow_flag(dev)
        code = lib.mlqem_asap_coarsen_lists_fill(num_nodes, k, int(num_edges), cap, _p(in_ptr), _p(out_ptr), _p(in_src), _p(out_dst),
                                                 _p(out_eid), e, _p(flag), _p(ws), need, _stream())
        _lib.check(code, "mlqem_asap_coarsen_lists_fill")
        _overflow_what[flag.device] = f"mlqem_asap_coarsen_lists_fill: a list outgrew its capacity (last launch: capacity {cap}, k = {k} clusters)"
        if _lib._SYNC_OPS:
            check_overflow_flags()
    return CsrArrays(in_ptr, in_src[:e], out_ptr, out_dst[:e], loops[:k], None if out_eid is None else out_eid[:e]), slot, e


# One persistent int32 flag per device that the capacity-bound kernels atomicOr into.  It is allocated (zero) OUTSIDE any capture and
# never re-zeroed inside one: a flag raised by the 500th replay of a captured step is still set when the host looks (the
# per-launch flags of rounds 4-5 were registered at capture time only and capped at the 64 most recent launches).
_overflow_sticky = {}     # device -> int32[1]
_overflow_what = {}       # device -> what the last launch that could raise it was


def _overflow_flag(device):
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    flag = _overflow_sticky.get(device)
    if flag is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the overflow flag must exist before a capture begins: run the step eagerly once (or ops.prepare_device) first")
        flag = _overflow_sticky[device] = torch.full((1,), 0, dtype=torch.int32, device=device)
    return flag


def check_overflow_flags():
    """Reads (one host sync per device that launched a capacity-bound kernel) the sticky overflow flags, resets them and raises if one
    was set since the last call.  A no-op while a capture is in progress."""
    if not _overflow_sticky or torch.cuda.is_current_stream_capturing():
        return
    bad = []
    for device, flag in _overflow_sticky.items():
        if device not in _overflow_what:
            continue                                  # no capacity-bound launch was ever enqueued (or captured) on this device
        if int(flag.item()) != 0:                     # (the entry stays: replays of a captured step raise the flag without the host)
            flag.zero_()
            bad.append(_overflow_what[device])
    if bad:
        raise _lib.NativeLibraryError("capacity overflow on the device: " + "; ".join(sorted(set(bad))))


ASAP_LISTS_MAX_CAPACITY = 1 << 32      # entries per list buffer the list form addresses (32-bit places in its per-node records)


def asap_lists_max_k() -> int:
    return int(_lib.load().mlqem_asap_coarsen_lists_max_k())


def batch_norm_train(x, gamma, beta, eps):
    """y, mean, biased var, invstd of BatchNorm1d in training mode over the rows of x [N, C] (mlqem_batch_norm_train_f32)."""
    n, c = x.shape
    ldx = _mat(x, "x")
    dev = x.device
    y = padded_empty(n, c, dev)
    mean, var, invstd = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(3))
    lib = _lib.load()
    need = lib.mlqem_batch_norm_workspace_bytes(n, c)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    code = lib.mlqem_batch_norm_train_f32(_p(x), ldx, n, c, _p(gamma), _p(beta), float(eps), _p(y), _mat(y, "y"), _p(mean), _p(var),
                                          _p(invstd), _p(ws), need, _stream())
    _lib.check(code, "mlqem_batch_norm_train_f32")
    return y, mean, var, invstd


def batch_norm_train_bwd(dy, x, gamma, mean, invstd):
    """dx, dgamma, dbeta (mlqem_batch_norm_train_bwd_f32)."""
    n, c = x.shape
    dev = x.device
    dx = padded_empty(n, c, dev)
    dgamma, dbeta = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(2))
    lib = _lib.load()
    need = lib.mlqem_batch_norm_workspace_bytes(n, c)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    code = lib.mlqem_batch_norm_train_bwd_f32(_p(dy), _mat(dy, "dy"), _p(x), _mat(x, "x"), n, c, _p(gamma), _p(mean), _p(invstd),
                                              _p(dx), _mat(dx, "dx"), _p(dgamma), _p(dbeta), _p(ws), need, _stream())
    _lib.check(code, "mlqem_batch_norm_train_bwd_f32")
    return dx, dgamma, dbeta


def asap_slot_map(perm, num_nodes, graph_ptr=None, new_graph_ptr=None, num_graphs=0):
    """slot[N]: cluster id of every kept centre (slot[perm[p]] = p), -1 elsewhere.  With the graphs' node ranges before / after the
    pooling: one launch (a workgroup per graph) instead of a fill and a scatter."""
    k = int(perm.shape[0])
    slot = torch.empty(max(num_nodes, 1), dtype=torch.int32, device=perm.device)
    if graph_ptr is not None and new_graph_ptr is not None and num_graphs > 0:
        code = _lib.load().mlqem_asap_slot_map_graphs(_p(perm), _p(graph_ptr), _p(new_graph_ptr), int(num_graphs), num_nodes, k, _p(slot), _stream())
        _lib.check(code, "mlqem_asap_slot_map_graphs")
        return slot
    code = _lib.load().mlqem_asap_slot_map(_p(perm), num_nodes, k, _p(slot), _stream())
    _lib.check(code, "mlqem_asap_slot_map")
    return slot


def asap_rows_max_bits() -> int:
    return int(_lib.load().mlqem_asap_coarsen_rows_max_bits())


def asap_dense_max_k() -> int:
    return int(_lib.load().mlqem_asap_coarsen_dense_max_k())


# ------------------------------------------------------------------------------------------ Family B backward
def transformer_attention_train(qkvs, in_ptr, in_src, loops, num_edges, heads, channels, drop_p=0.0, seed=0, pair_key=False, ell=None,
                                head_pitch=0):
    """(out, attn_out, m, den).  ``ell``: the structure's in-edge side table (``GraphStructure.in_ell``) where most rows have at
    most two in-edges (circuit DAGs): those rows skip the ptr -> idx round trip.  ``attn_out`` is for the backward only (rows of at
    most four entries are not written)."""
    n, hc = qkvs.shape[0], heads * channels
    _ell(ell, n)
    cp = head_pitch or channels      # ``head_pitch``: channel pitch of a head inside qkvs' parts (pads zero; out stays compact)
    if qkvs.shape[1] != 4 * heads * cp:
        raise ValueError("qkvs must be [N, 4*H*head_pitch]")
    _vec(in_ptr, "in_ptr", n + 1, torch.int32)
    _vec(loops, "loops", n, torch.int32)
    dev = qkvs.device
    out, attn = padded_empty(n, hc, dev), padded_empty(n, hc, dev)
    m = torch.empty((max(n, 1), heads), dtype=torch.float32, device=dev)
    den = torch.empty_like(m)
    code = _lib.load().mlqem_transformer_attention_train_f32(
        _p(qkvs), _mat(qkvs, "qkvs"), _p(in_ptr), _p(in_src), _p(loops), n, num_edges, heads, channels, float(drop_p),
        int(seed) & 0xFFFFFFFFFFFFFFFF, _p(_seed_counter) if drop_p > 0 else None, 1 if pair_key else 0, _p(ell), int(head_pitch), _p(out), _mat(out, "out"),
        _p(attn), _mat(attn, "attn"), _p(m), _p(den), _stream())
    _lib.check(code, "mlqem_transformer_attention_train_f32")
    return out, attn, m, den


def transformer_attention_bwd(qkvs, g, attn, m, den, s, num_edges, heads, channels, drop_p=0.0, seed=0, pair_key=False, head_pitch=0):
    """Gradient of [query | key | value | skip].  A structure without ``out_eid`` (ASAPooling's coarsened graphs) takes the
    recomputed form: no per-edge buffers, the source side recomputes its weights from m / den / g . attn_out per (row, head)."""
    n = qkvs.shape[0]
    g = rowmajor(g)
    dev = qkvs.device
    gqkvs = padded_empty(n, 4 * heads * (head_pitch or channels), dev)      # the layout of qkvs (``head_pitch``: pads come out zero)
    if s.out_eid is None:          # the recomputed form: one 16-byte record {m, 1 / den, g . attn_out} per (row, head)
        al, gs = torch.empty(4 * max(n, 1) * heads, dtype=torch.float32, device=dev), None
    else:
        scratch = torch.empty((2, (num_edges + n) * heads + 1), dtype=torch.float32, device=dev)
        al, gs = scratch[0], scratch[1]
    code = _lib.load().mlqem_transformer_attention_bwd_f32(
        _p(qkvs), _mat(qkvs, "qkvs"), _p(g), _mat(g, "g"), _p(attn), _mat(attn, "attn"), _p(m), _p(den), _p(s.in_ptr),
        _p(s.in_src), _p(s.out_ptr), _p(s.out_dst), _p(s.out_eid), _p(s.loops), n, num_edges, heads, channels,
        float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(_seed_counter) if drop_p > 0 else None, 1 if pair_key else 0, int(head_pitch),
        _p(gqkvs), _mat(gqkvs, "gqkvs"), _p(al), _p(gs),
        _stream())
    _lib.check(code, "mlqem_transformer_attention_bwd_f32")
    return gqkvs


def csr_softmax_aggregate_bwd(x, xnew, gnew, s, num_edges, a_dst, c_src, negative_slope, xmax=None, gx_rank1=None, fuse_max_col=None):
    """(gx, g_a, g_c) -- and, given ``xmax`` (the segment max of x over the same entries, <= 128 channels), a fourth result: the
    per-channel tie counts ``csr_segment_max_bwd_`` would otherwise walk the in-edges for.  ``fuse_max_col`` (a [C] vector w; stored
    form, with ``xmax``): gx also receives the backward of that segment max for a gradient g_a (x) w, and ``csr_segment_max_bwd_``
    is not to be called."""
    n, c = x.shape
    dev = x.device
    gx = padded_empty(n, c, dev)
    ties = padded_empty(n, c, dev) if (xmax is not None and c <= 128) else None
    g_a = torch.empty(max(n, 1), dtype=torch.float32, device=dev)[:n]
    g_c = torch.empty(max(n, 1), dtype=torch.float32, device=dev)[:n]
    if s.out_eid is None:          # the recomputed form: one 16-byte record per row instead of two values per edge
        al, gp = torch.empty(4 * max(n, 1), dtype=torch.float32, device=dev), None
    else:
        scratch = torch.empty((2, num_edges + n + 1), dtype=torch.float32, device=dev)
        al, gp = scratch[0], scratch[1]
    code = _lib.load().mlqem_csr_softmax_aggregate_bwd_f32(
        _p(x), _mat(x, "x"), _p(xnew), _mat(xnew, "xnew"), _p(gnew), _mat(gnew, "gnew"), _p(s.in_ptr), _p(s.in_src),
        _p(s.out_ptr), _p(s.out_dst), _p(s.out_eid), _p(a_dst), _p(c_src), float(negative_slope), n, num_edges, c, 0,
        _p(gx), _mat(gx, "gx"), _p(g_a), _p(g_c), _p(al), _p(gp),
        _p(xmax) if ties is not None else None, _mat(xmax, "xmax") if ties is not None else 0,
        _p(ties), _mat(ties, "ties") if ties is not None else 0, _p(gx_rank1) if c <= 128 else None, _p(fuse_max_col), _stream())
    if gx_rank1 is not None and c > 128:         # wider than the kernels that fold it in: as its own pass
        linear(g_c.unsqueeze(1), gx_rank1.reshape(1, -1).contiguous(), transposed=True, out=gx, accumulate=True)
    _lib.check(code, "mlqem_csr_softmax_aggregate_bwd_f32")
    return (gx, g_a, g_c) if xmax is None else (gx, g_a, g_c, ties)


def csr_segment_max_bwd_(gx, x, xmax, gmax, s, ties=None, gmax_rank1=None):
    """gx += backward of the segment max over structure ``s`` (in place); ``ties``: the counts ``csr_softmax_aggregate_bwd`` left.
    ``gmax_rank1 = (row [N], col [C])``: the maximum's gradient is row (x) col and is never formed (needs ``ties``; ``gmax`` unused)."""
    n, c = x.shape
    share = padded_empty(n, c, x.device)
    row, col = gmax_rank1 if (gmax_rank1 is not None and ties is not None) else (None, None)
    if gmax_rank1 is not None and row is None:
        gmax = gmax_rank1[0].unsqueeze(1) * gmax_rank1[1].unsqueeze(0)
    code = _lib.load().mlqem_csr_segment_max_bwd_f32(_p(x), _mat(x, "x"), _p(xmax), _mat(xmax, "xmax"), _p(gmax) if row is None else None,
                                                     _mat(gmax, "gmax") if row is None else 0, _p(s.in_ptr), _p(s.in_src), _p(s.out_ptr),
                                                     _p(s.out_dst), n, c, _p(gx), _mat(gx, "gx"), _p(share),
                                                     _mat(share, "share"), _p(ties), _mat(ties, "ties") if ties is not None else 0,
                                                     _p(row), _p(col), _stream())
    _lib.check(code, "mlqem_csr_segment_max_bwd_f32")
    return gx


def rank_grad(terms):
    """Up to three weighted column sums over the same rows in ONE pass (mlqem_rank_grad_f32): ``terms`` = [(g, x), ...] with g [N]
    or [N, k <= 3] weights and x [N, D] in the padded row layout.  Returns [(g^T x [k, D], column sums of g [k]), ...] -- the
    weight and bias gradients of ASAPooling's one- and three-wide projections (what three ``linear_wgrad`` calls computed in six
    launches)."""
    import ctypes as _ct

    if not 1 <= len(terms) <= 3:
        raise ValueError("rank_grad: one to three terms")
    n, d = terms[0][1].shape
    gs, xs, ks = [], [], []
    for g, x in terms:
        if tuple(x.shape) != (n, d):
            raise ValueError("rank_grad: the matrices must share one shape")
        g2 = g.unsqueeze(1) if g.dim() == 1 else g
        if g2.shape[0] != n or g2.shape[1] > 3 or g2.dtype != torch.float32 or not g2.is_cuda or g2.stride(1) != 1:
            raise ValueError("rank_grad: weights must be fp32 [N] or [N, <= 3] on the device")
        x = rowmajor(x)
        _mat(x, "x")
        if n > 0 and (x.stride(0) < (d + 3) // 4 * 4 or x.stride(0) % 4 or x.data_ptr() % 16):
            x = padded_copy(x)
        gs.append(g2); xs.append(x); ks.append(int(g2.shape[1]))
    t = len(terms)
    dev = xs[0].device
    arr_p = lambda ts: (_ct.c_void_p * 3)(*[ts[i].data_ptr() if i < t else None for i in range(3)])
    arr_l = lambda vs: (_ct.c_int64 * 3)(*[int(vs[i]) if i < t else 0 for i in range(3)])
    ldx = [int(x.stride(0)) if n > 1 else (d + 3) // 4 * 4 for x in xs]
    ldg = [int(g.stride(0)) if n > 1 else int(g.shape[1]) for g in gs]
    k_arr = (_ct.c_int * 3)(*[ks[i] if i < t else 0 for i in range(3)])
    total = sum(ks)
    out = torch.empty((total, d), dtype=torch.float32, device=dev)
    bias = torch.empty(total, dtype=torch.float32, device=dev)
    lib = _lib.load()
    need = lib.mlqem_rank_grad_workspace_bytes(d)
    ws = _wgrad_workspace(dev, need)
    code = lib.mlqem_rank_grad_f32(t, arr_p(xs), arr_l(ldx), arr_p(gs), arr_l(ldg), k_arr, n, d, _p(out), _p(bias), _p(ws), need, _stream())
    if code == _lib.ERR_UNSUPPORTED:           # wider than the kernel's 64 column groups: the general weight-gradient kernel, term by term
        res, at = [], 0
        for g, x, k in zip(gs, xs, ks):
            gw, gb = torch.empty((k, d), dtype=torch.float32, device=dev), torch.empty(k, dtype=torch.float32, device=dev)
            linear_wgrad(g.contiguous(), x, gw, gb)
            res.append((gw, gb))
        return res
    _lib.check(code, "mlqem_rank_grad_f32")
    res, at = [], 0
    for k in ks:
        res.append((out[at:at + k], bias[at:at + k]))
        at += k
    return res


def gather_scale_rows_bwd(gout, xnew, fitness, slot):
    n, c = xnew.shape
    gout = rowmajor(gout)
    gxnew = padded_empty(n, c, xnew.device)
    gfit = torch.empty(max(n, 1), dtype=torch.float32, device=xnew.device)[:n]
    code = _lib.load().mlqem_gather_scale_rows_bwd_f32(_p(gout), _mat(gout, "gout") if gout.shape[0] else c, _p(xnew),
                                                       _mat(xnew, "xnew"), _p(fitness), _p(slot), n, c, _p(gxnew),
                                                       _mat(gxnew, "gxnew"), _p(gfit), _stream())
    _lib.check(code, "mlqem_gather_scale_rows_bwd_f32")
    return gxnew, gfit


def gather_rows_dot(gout, xnew, slot):
    """g_f[row] = g_out[slot[row]] . x'[row] (0 where slot < 0): the first half of ``gather_scale_rows_bwd``.  None when the layout is
    not the padded one (the caller then takes ``gather_scale_rows_bwd``)."""
    n, c = xnew.shape
    gout = rowmajor(gout)
    if gout.shape[0] == 0:
        return None
    gfit = torch.empty(max(n, 1), dtype=torch.float32, device=xnew.device)[:n]
    code = _lib.load().mlqem_gather_rows_dot_f32(_p(gout), _mat(gout, "gout"), _p(xnew), _mat(xnew, "xnew"), _p(slot), n, c, _p(gfit), _stream())
    if code == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(code, "mlqem_gather_rows_dot_f32")
    return gfit


def scatter_scale_rank(gout, fitness, slot, g3, w3, n, c):
    """g_x'[row] = (slot[row] >= 0 ? g_out[slot[row]] f[row] : 0) + g3[row] @ w3 in one store (g3 [N, K <= 3], w3 [K, C]): the second half
    of ``gather_scale_rows_bwd`` with the rank-K update that followed it.  Layout as ``gather_rows_dot`` (which the caller tried first)."""
    gout = rowmajor(gout)
    gxnew = padded_empty(n, c, fitness.device)
    w3 = w3.contiguous()
    code = _lib.load().mlqem_scatter_scale_rank_f32(_p(gout), _mat(gout, "gout"), _p(fitness), _p(slot), _p(g3), _mat(g3, "g3"), _p(w3),
                                                    int(g3.shape[1]), n, c, _p(gxnew), _mat(gxnew, "gxnew"), _stream())
    _lib.check(code, "mlqem_scatter_scale_rank_f32")
    return gxnew


def leconv_fitness_bwd(gfit, fitness, in_ptr, out_ptr, out_dst):
    n = fitness.shape[0]
    gpqr = torch.empty((max(n, 1), 3), dtype=torch.float32, device=fitness.device)[:n]
    code = _lib.load().mlqem_leconv_fitness_bwd_f32(_p(gfit), _p(fitness), _p(in_ptr), _p(out_ptr), _p(out_dst), n,
                                                    _p(gpqr), _stream())
    _lib.check(code, "mlqem_leconv_fitness_bwd_f32")
    return gpqr


# ------------------------------------------------------------------------------------------ row order of a coarsened graph
# The coarsened graphs ASAPooling makes of large circuits are unions of dense blocks: rows whose centres are close in program order
# share their sources.  The dense-block plans (DensePlan below) take a structure's long rows 16 at a time in that order.
ORDER_SPAN_SLACK = 32      # added to twice the largest pooled graph: a bound on the id range of one block's entries


def tile_order_by_position(slot, graph_ptr, new_graph_ptr, num_graphs, k_total):
    order = torch.empty(max(k_total, 1), dtype=torch.int32, device=slot.device)[:k_total]
    code = _lib.load().mlqem_tile_order_by_position(_p(slot), _p(graph_ptr), _p(new_graph_ptr), num_graphs, _p(order), _stream())
    _lib.check(code, "mlqem_tile_order_by_position")
    return order


class DensePlan:
    """A structure's long rows as dense blocks (csrc/dense_block.hpp): what the matrix-core forms of the edge walks read."""

    def __init__(self, records, counter, row_flag, lrows, max_blocks):
        self.records, self.counter, self.row_flag, self.lrows, self.max_blocks = records, counter, row_flag, lrows, int(max_blocks)

    def args(self):
        return _p(self.records), _p(self.counter), _p(self.row_flag), self.max_blocks


def dense_plan_build(ptr, idx, loops, num_rows, graph_ptr, num_graphs, order, max_span) -> DensePlan:
    """The dense blocks of the CSR (ptr, idx): rows of 32+ entries, 16 at a time in ``order`` (program position; None = row order)
    inside each graph (``graph_ptr``: the graphs' position ranges)."""
    lib = _lib.load()
    dev = ptr.device
    _vec(ptr, "ptr", num_rows + 1, torch.int32)
    max_blocks = max(int(lib.mlqem_dense_plan_max_blocks(num_rows, num_graphs)), 1)
    # the block counter and the row flags in ONE zeroed buffer (one fill launch instead of two: twice per pooled graph and step)
    zeroed = torch.zeros(16 + max(num_rows, 1), dtype=torch.uint8, device=dev)
    counter, row_flag = zeroed[:4].view(torch.int32), zeroed[16:]
    lrows = torch.empty(max_blocks * 17, dtype=torch.int32, device=dev)      # the blocks' rows, then one graph id per block
    records = torch.empty(max_blocks * lib.mlqem_dense_plan_record_ints(), dtype=torch.int32, device=dev)
    code = lib.mlqem_dense_plan_build(_p(ptr), _p(idx), _p(loops), _p(order), _p(graph_ptr), num_graphs, num_rows, int(max_span),
                                      _p(counter), _p(lrows), _p(records), _p(row_flag), _stream())
    _lib.check(code, "mlqem_dense_plan_build")
    return DensePlan(records, counter, row_flag, lrows, max_blocks)


def dense_attention_supported(heads, channels, head_pitch) -> bool:
    return bool(_lib.load().mlqem_dense_attention_supported(heads, channels, head_pitch or channels))


def _two_streams(first, second, side):
    """``first()`` on a side stream while ``second()`` runs on the caller's: two launches that touch disjoint rows.  Joined before
    returning (the caller's stream then holds both)."""
    if side is None:
        first()
        second()
        return
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        first()
    second()
    main.wait_stream(side)


def dense_attention_train(qkvs, in_ptr, in_src, loops, num_edges, heads, channels, plan: DensePlan, drop_p=0.0, seed=0, head_pitch=16,
                          side=None):
    """``transformer_attention_train`` (pair-keyed draws) with the plan's rows on the matrix cores: (out, attn_out, m, den).
    ``side``: a stream for the per-edge kernel over the rows outside the blocks, which then runs beside the block kernel."""
    n, hc = qkvs.shape[0], heads * channels
    dev = qkvs.device
    out, attn = padded_empty(n, hc, dev), padded_empty(n, hc, dev)
    m = torch.empty((max(n, 1), heads), dtype=torch.float32, device=dev)
    den = torch.empty_like(m)

    def call(parts):
        code = _lib.load().mlqem_dense_attention_train_f32(
            _p(qkvs), _mat(qkvs, "qkvs"), _p(in_ptr), _p(in_src), _p(loops), n, num_edges, heads, channels, float(drop_p),
            int(seed) & 0xFFFFFFFFFFFFFFFF, _p(_seed_counter) if drop_p > 0 else None, int(head_pitch), *plan.args(), parts,
            _p(out), _mat(out, "out"), _p(attn), _mat(attn, "attn"), _p(m), _p(den), _stream())
        _lib.check(code, "mlqem_dense_attention_train_f32")

    _two_streams(lambda: call(1), lambda: call(2), side)
    return out, attn, m, den


def dense_attention_bwd(qkvs, g, attn, m, den, s, num_edges, heads, channels, plan_in: DensePlan, plan_out: DensePlan, drop_p=0.0, seed=0,
                        head_pitch=16, side=None):
    """``transformer_attention_bwd`` (recomputing form) with the plans' rows on the matrix cores: the gradient of qkvs."""
    n = qkvs.shape[0]
    g = rowmajor(g)
    dev = qkvs.device
    gqkvs = padded_empty(n, 4 * heads * head_pitch, dev)
    al = torch.empty(4 * max(n, 1) * heads, dtype=torch.float32, device=dev)

    def call(parts):
        code = _lib.load().mlqem_dense_attention_bwd_f32(
            _p(qkvs), _mat(qkvs, "qkvs"), _p(g), _mat(g, "g"), _p(attn), _mat(attn, "attn"), _p(m), _p(den), _p(s.in_ptr), _p(s.in_src),
            _p(s.out_ptr), _p(s.out_dst), _p(s.loops), n, num_edges, heads, channels, float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF,
            _p(_seed_counter) if drop_p > 0 else None, int(head_pitch), *plan_in.args(), *plan_out.args(), parts, _p(gqkvs),
            _mat(gqkvs, "gqkvs"), _p(al), _stream())
        _lib.check(code, "mlqem_dense_attention_bwd_f32")

    # the source side reads the records BOTH destination-side launches file: joined in between
    _two_streams(lambda: call(1), lambda: call(2), side)
    _two_streams(lambda: call(4), lambda: call(8), side)
    return gqkvs


def dense_pool_supported(d) -> bool:
    return bool(_lib.load().mlqem_dense_pool_supported(int(d)))


def dense_softmax_aggregate(x, in_ptr, in_src, a_dst, c_src, negative_slope, plan: DensePlan, want_stat=True):
    """``csr_softmax_aggregate`` with the plan's rows on the matrix cores: (x', stat) -- stat [N, 2] = {maximum, 1 / denominator} of the
    block rows (other rows: not written), what ``dense_softmax_aggregate_bwd`` reads."""
    n, c = x.shape
    _vec(a_dst, "a_dst", n)
    _vec(c_src, "c_src", n)
    out = padded_empty(n, c, x.device)
    stat = torch.empty((max(n, 1), 2), dtype=torch.float32, device=x.device) if want_stat else None
    code = _lib.load().mlqem_dense_softmax_aggregate_f32(_p(x), _mat(x, "x"), _p(in_ptr), _p(in_src), _p(a_dst), _p(c_src),
                                                         float(negative_slope), n, c, *plan.args(), _p(out), _mat(out, "out"), _p(stat),
                                                         _stream())
    _lib.check(code, "mlqem_dense_softmax_aggregate_f32")
    return out, stat


def dense_pool_fits(*mats) -> bool:
    """The pooling's block kernels take matrices of 29-32 channels whose rows are exactly 32 floats, or of 45-48 channels in rows of 48 (the
    padded row layout of either: two or three channel tiles of 16; 45 = the heads-5/3 variants' second pooling, gnn.py:178-276)."""
    def ok(m):
        c = m.shape[1]
        ld = 32 if 29 <= c <= 32 else (48 if 45 <= c <= 48 else 0)
        return (ld and m.is_cuda and m.dtype == torch.float32 and (m.shape[0] <= 1 or m.stride(0) == ld) and m.stride(1) == 1
                and m.data_ptr() % 16 == 0)
    return all(ok(m) for m in mats)


def dense_leconv_fitness_bwd(gfit, fitness, in_ptr, out_ptr, out_dst, plan_out: DensePlan):
    """``leconv_fitness_bwd`` with the long rows of the out-structure's plan summed by a wave each."""
    n = fitness.shape[0]
    gpqr = torch.empty((max(n, 1), 3), dtype=torch.float32, device=fitness.device)[:n]
    code = _lib.load().mlqem_dense_leconv_fitness_bwd_f32(_p(gfit), _p(fitness), _p(in_ptr), _p(out_ptr), _p(out_dst), n, _p(plan_out.lrows),
                                                          _p(plan_out.counter), _p(plan_out.row_flag), plan_out.max_blocks, _p(gpqr), _stream())
    _lib.check(code, "mlqem_dense_leconv_fitness_bwd_f32")
    return gpqr


def dense_segment_max(x, in_ptr, in_src, plan: DensePlan):
    """``csr_segment_max`` (the row itself included) with the plan's rows walked as dense blocks."""
    n, c = x.shape
    out = padded_empty(n, c, x.device)
    code = _lib.load().mlqem_dense_segment_max_f32(_p(x), _mat(x, "x"), _p(in_ptr), _p(in_src), n, c, *plan.args(), _p(out), _mat(out, "out"),
                                                   _stream())
    _lib.check(code, "mlqem_dense_segment_max_f32")
    return out


def dense_softmax_aggregate_bwd(x, xnew, gnew, s, num_edges, a_dst, c_src, negative_slope, stat, plan_in: DensePlan, plan_out: DensePlan,
                                xmax, gx_rank1=None):
    """``csr_softmax_aggregate_bwd`` (recomputing form, with the tie counts of the segment max): (gx, g_a, g_c, ties)."""
    n, c = x.shape
    dev = x.device
    gx, ties = padded_empty(n, c, dev), padded_empty(n, c, dev)
    g_a = torch.empty(max(n, 1), dtype=torch.float32, device=dev)[:n]
    g_c = torch.empty(max(n, 1), dtype=torch.float32, device=dev)[:n]
    al = torch.empty(4 * max(n, 1), dtype=torch.float32, device=dev)
    code = _lib.load().mlqem_dense_softmax_aggregate_bwd_f32(
        _p(x), _mat(x, "x"), _p(xnew), _mat(xnew, "xnew"), _p(gnew), _mat(gnew, "gnew"), _p(s.in_ptr), _p(s.in_src), _p(s.out_ptr),
        _p(s.out_dst), _p(a_dst), _p(c_src), float(negative_slope), n, num_edges, c, _p(stat), *plan_in.args(), *plan_out.args(), _p(gx),
        _mat(gx, "gx"), _p(g_a), _p(g_c), _p(al), _p(xmax), _mat(xmax, "xmax"), _p(ties), _mat(ties, "ties"), _p(gx_rank1), _stream())
    _lib.check(code, "mlqem_dense_softmax_aggregate_bwd_f32")
    return gx, g_a, g_c, ties


def dense_segment_max_bwd_(gx, x, xmax, s, ties, gmax_rank1, plan_out: DensePlan):
    """``csr_segment_max_bwd_`` with tie counts and the maximum's gradient as row (x) col: gx += (in place)."""
    n, c = x.shape
    share = padded_empty(n, c, x.device)
    row, col = gmax_rank1
    code = _lib.load().mlqem_dense_segment_max_bwd_f32(_p(x), _mat(x, "x"), _p(xmax), _mat(xmax, "xmax"), _p(s.in_ptr), _p(s.in_src),
                                                       _p(s.out_ptr), _p(s.out_dst), n, c, _p(gx), _mat(gx, "gx"), _p(share),
                                                       _mat(share, "share"), _p(ties), _mat(ties, "ties"), _p(row), _p(col),
                                                       *plan_out.args(), _stream())
    _lib.check(code, "mlqem_dense_segment_max_bwd_f32")
    return gx

