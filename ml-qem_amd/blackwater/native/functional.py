"""torch.autograd nodes over the native kernels.

One node per conv layer (``_GCNLayer``, ``_ChebLayer``, ``_SAGELayer``, ``_TransformerConv``, ``_ASAPool``): its forward
and backward are short, fixed sequences of launches with the element-wise work folded into kernel epilogues; plus the
generic differentiable building blocks ``csr_aggregate``, ``linear`` and ``segment_mean``."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.autograd import Function

from . import ops
from .structure import GraphStructure


class _CsrAggregate(Function):
    """y = act(alpha * (R A C x + D x) + beta * z + bias);  backward runs the same kernel on the transposed CSR."""

    @staticmethod
    def forward(ctx, x, z, bias, struct: GraphStructure, cscale, rscale, dself, alpha, beta, relu, drop_p, seed):
        x = ops.rowmajor(x)
        y = ops.csr_aggregate(x, struct.in_ptr, struct.in_src, ell=struct.in_ell, cscale=cscale, rscale=rscale, dself=dself, alpha=alpha,
                              z=z, beta=beta, bias=bias, relu=relu, drop_p=drop_p, seed=seed)
        ctx.struct, ctx.scales = struct, (cscale, rscale, dself)
        ctx.alpha, ctx.beta, ctx.relu, ctx.drop_p = alpha, beta, relu, drop_p
        ctx.has_z, ctx.has_bias = z is not None, bias is not None
        ctx.save_for_backward(y if (relu or drop_p > 0) else None)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = ops.rowmajor(g)
        if y is not None:
            g = ops.relu_dropout_bwd(g, y, 1.0 / (1.0 - ctx.drop_p) if ctx.drop_p > 0 else 1.0)
        cscale, rscale, dself = ctx.scales
        s = ctx.struct
        gx = gz = gb = None
        if ctx.needs_input_grad[0]:
            gx = ops.csr_aggregate(g, s.out_ptr, s.out_dst, ell=s.out_ell, cscale=rscale, rscale=cscale, dself=dself, alpha=ctx.alpha)
        if ctx.has_z and ctx.needs_input_grad[1]:
            gz = g * ctx.beta
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gz, gb, None, None, None, None, None, None, None, None, None


def csr_aggregate(x, struct, *, cscale=None, rscale=None, dself=None, alpha=1.0, z=None, beta=0.0, bias=None,
                  relu=False, drop_p=0.0, seed=0):
    return _CsrAggregate.apply(x, z, bias, struct, cscale, rscale, dself, alpha, beta, relu, drop_p, seed)


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, b, relu, mfma):
        x = ops.rowmajor(x)
        if mfma == "bf16":   # operands rounded to bf16 on the matrix cores, forward and both gradients
            y = ops.linear_bf16(x, w.contiguous(), b, relu=relu)
        elif mfma == "f32":
            y = ops.linear(x, w.contiguous(), b, relu=relu)
        else:
            raise ValueError(f"mfma must be 'f32' or 'bf16', got {mfma!r}")
        ctx.relu, ctx.mfma = relu, mfma
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        g = ops.rowmajor(g)
        if ctx.relu:
            g = ops.relu_dropout_bwd(g, y, 1.0)
        gx = gw = gb = None
        bf16 = ctx.mfma == "bf16" and w.shape[0] <= 256
        if ctx.needs_input_grad[0]:
            gx = ops.linear_bf16(g, w.contiguous(), transposed=True) if bf16 else ops.linear(g, w.contiguous(), transposed=True)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty_like(w, memory_format=torch.contiguous_format)
            gb = torch.empty(w.shape[0], dtype=w.dtype, device=w.device) if ctx.has_bias else None
            (ops.linear_wgrad_bf16 if ctx.mfma == "bf16" else ops.linear_wgrad)(g, x, gw, gb)
        return gx, gw, gb, None, None


def linear(x, w, b=None, relu=False, mfma="f32"):
    lead = x.shape[:-1]
    y = _Linear.apply(x.reshape(-1, x.shape[-1]), w, b, relu, mfma)
    return y.reshape(*lead, w.shape[0])


class _Seq2(Function):
    """Linear -> [Dropout] -> Linear as one forward and one backward launch (csrc/seq2.hip): the dense heads of the graph models,
    which see one row per circuit."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, drop_p, seed):
        train = any(ctx.needs_input_grad[:5])
        w1c, w2c = w1.contiguous(), w2.contiguous()
        y, hidden, mask = ops.seq2_forward(x, w1c, None if b1 is None else b1.contiguous(), w2c, None if b2 is None else b2.contiguous(),
                                           drop_p=drop_p, seed=seed, keep=train)
        ctx.cfg = (drop_p, b1 is not None, b2 is not None)
        if train:
            ctx.save_for_backward(x, w1c, w2c, hidden, mask)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w1, w2, hidden, mask = ctx.saved_tensors
        drop_p, has_b1, has_b2 = ctx.cfg
        gx, gw1, gb1, gw2, gb2 = ops.seq2_backward(gy, x, w1, w2, hidden, mask, drop_p, want_gx=ctx.needs_input_grad[0],
                                                   want_b1=has_b1, want_b2=has_b2)
        return gx, gw1, gb1, gw2, gb2, None, None


def seq2_fused_ok(x, w1, w2) -> bool:
    return torch.is_tensor(x) and x.dim() in (2, 3) and ops.seq2_fits(x.reshape(-1, x.shape[-1]), w1, w2)


def seq2(x, w1, b1, w2, b2, drop_p=0.0, seed=0):
    """``(dropout(x @ w1.T + b1)) @ w2.T + b2`` over the last axis of a 2-D or 3-D x."""
    lead = x.shape[:-1]
    y = _Seq2.apply(x.reshape(-1, x.shape[-1]), w1, b1, w2, b2, float(drop_p), int(seed))
    return y.reshape(*lead, w2.shape[0])


class _MLP1(Function):
    """fc2(relu(fc1(x))) as one forward and one backward launch (csrc/mlp_head.hip); x gets no gradient."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, bf16):
        train = any(ctx.needs_input_grad[1:5])
        out, hs, xp = ops.mlp1_forward(x, w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous(), bf16=bf16, stash=train)
        ctx.bf16 = bf16
        ctx.save_for_backward(xp, hs, w2)
        ctx.dims = (w1.shape[1], w1.shape[0])
        return out

    @staticmethod
    def backward(ctx, g):
        xp, hs, w2 = ctx.saved_tensors
        gw1, gb1, gw2, gb2 = ops.mlp1_backward(g, xp, hs, w2.contiguous(), ctx.dims[0], ctx.dims[1], bf16=ctx.bf16)
        return None, gw1, gb1, gw2, gb2, None


def mlp1_fused_ok(x, w1, w2) -> bool:
    """Whether ``mlp1`` can take this call: a GPU feature matrix that needs no gradient, widths inside the kernel's limits."""
    return (torch.is_tensor(x) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and not x.requires_grad
            and ops.mlp1_fits(w1.shape[1], w1.shape[0], w2.shape[0]))


def mlp1(x, w1, b1, w2, b2, mfma="f32"):
    if mfma not in ("f32", "bf16"):
        raise ValueError(f"mfma must be 'f32' or 'bf16', got {mfma!r}")
    return _MLP1.apply(x, w1, b1, w2, b2, mfma == "bf16")


class _MLPTrunkBf16(Function):
    """MLP2 / MLP3 in training mode with bf16 storage (csrc/mlp_layers.hip): fc1 -> bn1 -> relu -> drop -> fc2 -> bn2 -> relu -> drop
    (+ residual) -> [fc3 -> relu -> drop -> fc4 | fc3], one autograd node.  Every activation it writes or saves is a [N, 128]
    bfloat16 matrix; the dropout masks are recomputed from their counters in the backward.  Returns (out, mean1, var1, mean2,
    var2) -- the batch statistics feed the running buffers."""

    @staticmethod
    def forward(ctx, x, cfg, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, w4, b4, run1=None, run2=None):
        eps1, eps2, p_trunk, p_tail, seeds = cfg[:5]
        f32 = ctx.f32 = len(cfg) > 5 and bool(cfg[5])      # fp32 storage (mfma = "f32"): the same graph on the _f32 entry points
        gemm = ctx.gemm = (ops.layer_gemm_f32 if f32 else ops.layer_gemm_bf16)
        n, i = x.shape
        h = w1.shape[0]
        xin = x if x.dtype == torch.bfloat16 else ops._mlp1_x(x)
        y1 = gemm(xin, w1.contiguous(), b1)
        m1, v1, is1, sc1, sh1 = ops.layer_colstats_fwd(y1, g1, be1, eps1, n, h, running=run1)
        x1 = ops.layer_act_bf16(y1, sc1, sh1, n, h, True, p_trunk, seeds[0])
        y2 = gemm(x1, w2.contiguous(), b2)
        m2, v2, is2, sc2, sh2 = ops.layer_colstats_fwd(y2, g2, be2, eps2, n, h, running=run2)
        s = ops.layer_act_bf16(y2, sc2, sh2, n, h, True, p_trunk, seeds[1], res=x1)
        if w4 is None:                      # MLP2: out = fc3(s)
            out = ops.layer_rowdot_bf16(s, w3.contiguous(), b3, n)
            y3 = h3 = None
        else:                               # MLP3: out = fc4(drop(relu(fc3(s))))
            h3w = w3.shape[0]
            ctx.one, ctx.zero = ops.layer_identity_vectors(x.device)
            # no BatchNorm between fc3 and its ReLU: the GEMM's epilogue applies ReLU + dropout, y3 is never stored (either storage)
            y3, h3 = None, gemm(s, w3.contiguous(), b3, relu=True, drop_p=p_tail, seed=seeds[2])
            out = ops.layer_rowdot_bf16(h3, w4.contiguous(), b4, n)
        ctx.cfg, ctx.dims = cfg, (n, i, h)
        ctx.x_needs_grad = ctx.needs_input_grad[0]
        ctx.save_for_backward(xin, y1, x1, y2, s, y3, h3, w1, w2, w3, w4, g1, g2, m1, is1, sc1, sh1, m2, is2, sc2, sh2)
        ctx.mark_non_differentiable(m1, v1, m2, v2)
        return out, m1[:h], v1[:h], m2[:h], v2[:h]

    @staticmethod
    def backward(ctx, gout, *_):
        xin, y1, x1, y2, s, y3, h3, w1, w2, w3, w4, g1, g2, m1, is1, sc1, sh1, m2, is2, sc2, sh2 = ctx.saved_tensors
        eps1, eps2, p_trunk, p_tail, seeds = ctx.cfg[:5]
        n, i, h = ctx.dims
        gemm, wgrad = ctx.gemm, (ops.layer_wgrad_f32 if ctx.f32 else ops.layer_wgrad_bf16)
        gout = ops.rowmajor(gout)
        if w4 is None:
            gs, gw3, gb3 = ops.layer_rowdot_bwd_bf16(gout, s, w3.contiguous(), n)
            gw4 = gb4 = None
        else:
            h3w = w3.shape[0]
            if y3 is None:    # h3 = dropout(relu(u)) is its own gate: the gradient at u in the launch that forms g w4
                dy3, gw4, gb4 = ops.layer_rowdot_bwd_bf16(gout, h3, w4.contiguous(), n, gate_scale=1.0 / (1.0 - p_tail))
            else:
                gh3, gw4, gb4 = ops.layer_rowdot_bwd_bf16(gout, h3, w4.contiguous(), n)
                dy3 = ops.layer_bwd_apply_bf16(gh3, y3, ctx.one, ctx.zero, ctx.zero, ctx.one, ctx.one, ctx.zero, ctx.zero, n, h3w, True,
                                               p_tail, seeds[2])
            gw3, gb3 = wgrad(dy3, s, h3w, h)
            gs = gemm(dy3, w3.contiguous(), transposed=True)
        # block 2: s = x1 + drop(relu(bn2(fc2 x1)))
        db2, dg2, gs2, k1, k2 = ops.layer_colstats_bwd(gs, y2, sc2, sh2, m2, is2, g2, True, p_trunk, seeds[1], n, h)
        dy2 = ops.layer_bwd_apply_bf16(gs, y2, sc2, sh2, m2, is2, gs2, k1, k2, n, h, True, p_trunk, seeds[1])
        gw2, gb2 = wgrad(dy2, x1, h, h)
        gx1 = gemm(dy2, w2.contiguous(), transposed=True, add=gs)       # + the residual path
        # block 1: x1 = drop(relu(bn1(fc1 x)))
        db1, dg1, gs1, k1, k2 = ops.layer_colstats_bwd(gx1, y1, sc1, sh1, m1, is1, g1, True, p_trunk, seeds[0], n, h)
        dy1 = ops.layer_bwd_apply_bf16(gx1, y1, sc1, sh1, m1, is1, gs1, k1, k2, n, h, True, p_trunk, seeds[0])
        gw1, gb1 = wgrad(dy1, xin, h, i)
        if not ctx.x_needs_grad:
            gx = None
        elif ctx.f32:
            gx = gemm(dy1, w1.contiguous(), transposed=True, narrow_out=True)
        else:
            gx = gemm(dy1, w1.contiguous(), transposed=True, out_f32=True)
        return (gx, None, gw1, gb1, dg1[:h], db1[:h], gw2, gb2, dg2[:h], db2[:h], gw3, gb3, gw4, gb4, None, None)


def mlp_trunk_bf16_ok(x, fc1, fc2, fc3, fc4, bn1, bn2) -> bool:
    """Whether the layer pipeline (csrc/mlp_layers.hip, either storage) takes this call: widths inside the kernels' limits, affine
    BatchNorm with a momentum."""
    if not (torch.is_tensor(x) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= 2):
        return False
    i, h = fc1.weight.shape[1], fc1.weight.shape[0]
    last = fc3 if fc4 is None else fc4
    if h > 128 or last.weight.shape[0] > ops.MLP1_MAX_OUT or i > (128 if x.requires_grad else ops.MLP1_MAX_IN):
        return False
    if fc4 is not None and fc3.weight.shape[0] > 128:
        return False
    for bn in (bn1, bn2):
        if bn.weight is None or bn.momentum is None:
            return False
    return True


def mlp_trunk_bf16(x, fc1, bn1, fc2, bn2, fc3, fc4, p_trunk, p_tail, seeds, f32=False):
    """Runs the block and updates the BatchNorm running statistics like torch (momentum, unbiased variance, batch counter).
    ``f32``: fp32 storage and unrounded operands (mfma = "f32") instead of bfloat16."""
    cfg = (float(bn1.eps), float(bn2.eps), float(p_trunk), float(p_tail), tuple(int(v) for v in seeds), bool(f32))
    w4, b4 = (None, None) if fc4 is None else (fc4.weight, fc4.bias)
    # BatchNorm1d's buffer update rides in the statistics launches (running mean / unbiased variance / batch counter)
    runs = [((bn.running_mean, bn.running_var, float(bn.momentum), bn.num_batches_tracked)
             if bn.track_running_stats and bn.running_mean is not None else None) for bn in (bn1, bn2)]
    out, _m1, _v1, _m2, _v2 = _MLPTrunkBf16.apply(x, cfg, fc1.weight, fc1.bias, bn1.weight, bn1.bias, fc2.weight, fc2.bias, bn2.weight,
                                                  bn2.bias, fc3.weight, fc3.bias, w4, b4, runs[0], runs[1])
    return out


class _ReluDropoutAdd(Function):
    """s = dropout(relu(u)) (+ residual) as one launch; the backward recovers the mask from the saved activation."""

    @staticmethod
    def forward(ctx, u, residual, drop_p, seed):
        y, s = ops.relu_dropout(u, drop_p, seed, residual)
        ctx.drop_p, ctx.has_res = drop_p, residual is not None
        ctx.save_for_backward(y)
        return y if s is None else s

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = ops.rowmajor(g)
        gu = ops.relu_dropout_bwd(g, y, 1.0 / (1.0 - ctx.drop_p) if ctx.drop_p > 0 else 1.0)
        return gu, (g if ctx.has_res else None), None, None


class _BatchNormTrain(Function):
    """BatchNorm1d in training mode on the native kernels (csrc/bn.hip): y, and the batch statistics for the running buffers."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = ops.rowmajor(x)
        y, mean, var, invstd = ops.batch_norm_train(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma, mean, invstd)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, gy, _gm, _gv):
        x, gamma, mean, invstd = ctx.saved_tensors
        dx, dgamma, dbeta = ops.batch_norm_train_bwd(ops.rowmajor(gy), x, gamma, mean, invstd)
        return dx, dgamma, dbeta, None


def batch_norm_train(x, bn):
    """``bn(x)`` for a ``torch.nn.BatchNorm1d`` in training mode over [N, C] rows, running statistics updated as torch does
    (momentum, unbiased variance, num_batches_tracked).  Falls back to the module itself for what the kernels do not cover
    (no affine parameters, cumulative-average momentum, C > 256, a single row)."""
    n, c = x.shape
    if bn.weight is None or bn.momentum is None or c > 256 or n < 2 or not x.is_cuda:
        return bn(x)
    y, mean, var = _BatchNormTrain.apply(x, bn.weight, bn.bias, bn.eps)
    if bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            m = bn.momentum
            bn.running_mean.mul_(1.0 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1.0 - m).add_(var, alpha=m * n / (n - 1))
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
    return y


def relu_dropout_add(u, residual=None, drop_p=0.0, seed=0):
    """dropout(relu(u)) + residual (the tail of an MLP2 / MLP3 trunk layer, docs/tutorials/mlp.py:60-66)."""
    return _ReluDropoutAdd.apply(u, residual, drop_p, seed)


# Mask hand-over between consecutive layers of one branch (a private contract of the model code, see nn/models.py):
# a layer called with ``defer_mask=True`` does not apply its own ReLU/dropout mask in backward and does not keep its
# output for it -- the NEXT layer, whose input x IS that output, is called with ``x_gate_scale = 1/(1-p)`` and returns
# gx already multiplied by (x > 0) * scale from the epilogue of its data-gradient GEMM.  One [N, C] read-modify-write
# pass per hidden layer disappears.  Only valid when the deferred layer's output feeds exactly that one consumer.


def _mask_grad(g, y, drop_p):
    """Gradient through the fused ReLU / dropout epilogue (the mask is read back from the saved output)."""
    g = ops.rowmajor(g)
    if y is None:
        return g
    return ops.relu_dropout_bwd(g, y, 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0)


# MLQEM_POOLED_GRAD=0: the pooled gradient of a Family A branch written out by ops.segment_pool_bwd and gathered by the transposed
# aggregation (A/B).  Default: computed inside that aggregation (ops.PooledGrad, csrc/pooled_grad.hip).
_POOLED_GRAD = os.environ.get("MLQEM_POOLED_GRAD", "1") != "0"
# ... for batches of at least this many nodes: below, a step is a chain of launches of a few microseconds each and the computed form
# has one more of them per branch than the written one (the reference's 32 four-qubit circuits per step: 0.222 -> 0.244 ms captured)
_POOLED_GRAD_MIN_NODES = 1 << 16

_PARTS_MAX_COLS = 64   # mlqem_linear_parts_f32 keeps the weight fragments of I <= 64 concatenated columns in registers


def _padded_rows(t):
    if isinstance(t, ops.RowsOf):     # rows of the (padded) arena, read through the batch's row map
        return t
    return t if (t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and t.stride(1) == 1) else ops.padded_copy(t)


def _fan_out(x, ws, biases, w_minus=None):
    """[x (W_0 - M_0)^T + b_0 | x W_1^T | ...] into separate padded buffers: one launch, x read once."""
    n, i = x.shape
    o = ws[0].shape[0]
    outs = [ops.padded_empty(n, o, x.device) for _ in ws]
    if i <= _PARTS_MAX_COLS:
        ops.linear_parts([x], ws, outs, w_minus=w_minus, biases=biases)
    else:   # wide inputs: one launch per block
        for j, w in enumerate(ws):
            wj = w if (w_minus is None or w_minus[j] is None) else w - w_minus[j]
            ops.linear(x, wj.contiguous(), None if biases is None else biases[j], out=outs[j])
    return outs


def _fan_in_t(gs, ws, i, w_minus=None, gate=None, gate_scale=1.0):
    """gx = sum_j gs[j] (W_j - M_j): one launch, gx written once; optional gate on the result."""
    n, o = gs[0].shape
    gx = ops.padded_empty(n, i, gs[0].device)
    if len(gs) * ((o + 3) // 4 * 4) <= _PARTS_MAX_COLS:
        ops.linear_parts(gs, ws, [gx], w_minus=w_minus, transposed=True, gate=gate, gate_scale=gate_scale)
    else:
        for j, g in enumerate(gs):
            last = j == len(gs) - 1
            wj = ws[j] if (w_minus is None or w_minus[j] is None) else ws[j] - w_minus[j]
            ops.linear(g, wj.contiguous(), transposed=True, out=gx, accumulate=j > 0, gate=gate if last else None,
                       gate_scale=gate_scale)
    return gx


class _ChebLayer(Function):
    """ChebConv (K = 2, 3) as ONE autograd node, evaluated PROJECT-FIRST (Clenshaw form).

    PyG computes T_0 = x, T_1 = L^x, T_2 = 2 L^T_1 - x and then sum_k T_k W_k^T (ChebConv.forward): every aggregation
    runs at the INPUT width and three [N, I] tensors are kept for the backward.  The same polynomial, re-associated:

        c_k = x W_k^T                      one GEMM over column blocks, x read once
        b_1 = c_1 + 2 L^ c_2               aggregation at the OUTPUT width, "+ c_1" in its epilogue
        y   = act((c_0 - c_2) + L^ b_1 + b)     c_0 - c_2 = x (W_0 - W_2)^T: folded into the weights

    so the aggregations move O instead of I columns (10 vs 22 in the first layer, 1 vs 10 in the second), and the
    backward needs only x: g_b1 = L^T g, g_c2 = 2 L^T g_b1, then ONE weight-gradient pass x^T [g | g_b1 | g_c2] and ONE
    GEMM gx = g (W_0 - W_2) + g_b1 W_1 + g_c2 W_2.  Same algebra as the reference, different fp32 rounding order
    (covered by the 1e-5 parity tests)."""

    @staticmethod
    def forward(ctx, x, bias, struct: GraphStructure, relu, drop_p, seed, defer_mask, x_gate_scale, *ws, pre=None, pool=None):
        s, k = struct, len(ws)
        x = _padded_rows(ops.rowmajor(x))
        o = ws[0].shape[0]
        ow = (o + 3) // 4 * 4
        ws = [w.contiguous() for w in ws]
        w_minus = [ws[2], None, None] if k == 3 else None          # block 0 multiplies by W_0 - W_2
        # pre: the projections c_k, already produced by a GEMM shared with other layers that read the same x
        c = list(pre) if pre is not None else _fan_out(x, ws, [bias] + [None] * (k - 1), w_minus)
        lap = dict(ell=s.in_ell, cscale=s.cheb_dinv, rscale=s.derived("cheb_neg"))
        act = dict(relu=relu, drop_p=drop_p, seed=seed)
        # pool: the pooled means of y from the launch that writes y (ops.csr_aggregate)
        if k == 2:
            y = ops.csr_aggregate(c[1], s.in_ptr, s.in_src, z=c[0], beta=1.0, out=c[0], pool=pool, **lap, **act)
        else:
            ops.csr_aggregate(c[2], s.in_ptr, s.in_src, alpha=2.0, z=c[1], beta=1.0, out=c[1], **lap)
            y = ops.csr_aggregate(c[1], s.in_ptr, s.in_src, z=c[0], beta=1.0, out=c[0], pool=pool, **lap, **act)
        ctx.struct, ctx.k, ctx.relu, ctx.drop_p, ctx.has_bias, ctx.dims = s, k, relu, drop_p, bias is not None, (o, ow)
        ctx.x_gate_scale = x_gate_scale
        ctx.save_for_backward(x, y if ((relu or drop_p > 0) and not defer_mask) else None, *ws)
        return y

    @staticmethod
    def backward(ctx, g, blocks_only=False):
        k, s, (o, ow) = ctx.k, ctx.struct, ctx.dims
        x, y, *ws = ctx.saved_tensors
        n, i = x.shape
        lap_t = dict(ell=s.out_ell, cscale=s.derived("cheb_neg"), rscale=s.cheb_dinv)
        if isinstance(g, ops.PooledGrad) and k >= 2 and y is None:      # the pooled gradient, computed inside its first aggregation
            g1, g = ops.pooled_grad_aggregate(g, s.out_ptr, s.out_dst, s.out_ell, s.derived("cheb_neg"), rscale=s.cheb_dinv)
            gs = [g, g1]
        else:
            if isinstance(g, ops.PooledGrad):
                g = g.materialise()
            g = _padded_rows(_mask_grad(g, y, ctx.drop_p))
            gs = [g]
            if k >= 2:
                gs.append(ops.csr_aggregate(g, s.out_ptr, s.out_dst, **lap_t))
        if k == 3:
            gs.append(ops.csr_aggregate(gs[1], s.out_ptr, s.out_dst, alpha=2.0, **lap_t))
        if blocks_only:     # the caller runs ONE weight-gradient pass over x for several layers (see cheb_grads_from_blocks)
            return gs
        gw = torch.empty((k * ow, i), dtype=torch.float32, device=x.device)
        gb = torch.empty(k * ow, dtype=torch.float32, device=x.device)
        ops.linear_wgrad_parts(gs, x, gw, gb)
        gw = gw.reshape(k, ow, i)[:, :o]
        gws = [gw[0], gw[1], gw[2] - gw[0]] if k == 3 else [gw[j] for j in range(k)]
        gx = None
        if ctx.needs_input_grad[0]:
            gated = ctx.x_gate_scale is not None
            gx = _fan_in_t(gs, ws, i, w_minus=[ws[2], None, None] if k == 3 else None, gate=x if gated else None,
                           gate_scale=ctx.x_gate_scale if gated else 1.0)
        return (gx, gb[:o] if ctx.has_bias else None, None, None, None, None, None, None, *gws)


class _ChebLayerRecurrence(Function):
    """ChebConv for any K (used for K = 1 and K > 3), in PyG's own order: T_0 = x, T_1 = L^x, T_k = 2 L^T_{k-1} - T_{k-2}; y = act(sum_k T_k W_k^T + b).
    Backward runs the recurrence in reverse with the transposed aggregation, folding every "+=" into the aggregation
    kernel's z/beta epilogue."""

    @staticmethod
    def forward(ctx, x, bias, struct: GraphStructure, relu, drop_p, seed, defer_mask, x_gate_scale, *ws):
        s = struct
        x = ops.rowmajor(x)
        k = len(ws)
        ws = [w.contiguous() for w in ws]
        lap = dict(cscale=s.cheb_dinv, rscale=s.derived("cheb_neg"))
        terms = [x]
        if k > 1:
            terms.append(ops.csr_aggregate(x, s.in_ptr, s.in_src, ell=s.in_ell, **lap))
        for _ in range(2, k):
            terms.append(ops.csr_aggregate(terms[-1], s.in_ptr, s.in_src, ell=s.in_ell, alpha=2.0, z=terms[-2],
                                           beta=-1.0, **lap))
        y = None
        for i in range(k):
            last = i == k - 1
            y = ops.linear(terms[i], ws[i], bias if i == 0 else None, out=y, accumulate=i > 0, relu=relu and last,
                           drop_p=drop_p if last else 0.0, seed=seed)
        ctx.struct, ctx.k, ctx.relu, ctx.drop_p, ctx.has_bias = s, k, relu, drop_p, bias is not None
        ctx.x_gate_scale = x_gate_scale
        ctx.save_for_backward(*terms, *ws, y if ((relu or drop_p > 0) and not defer_mask) else None)
        return y

    @staticmethod
    def backward(ctx, g):
        k, s = ctx.k, ctx.struct
        saved = ctx.saved_tensors
        terms, ws, y = saved[:k], saved[k:2 * k], saved[2 * k]
        g = _mask_grad(g, y, ctx.drop_p)
        gws, gb = [], None
        for i in range(k):
            gw = torch.empty_like(ws[i])
            want_b = i == 0 and ctx.has_bias
            if want_b:
                gb = torch.empty(ws[i].shape[0], dtype=g.dtype, device=g.device)
            ops.linear_wgrad(g, terms[i], gw, gb if want_b else None)
            gws.append(gw)
        gx = None
        if ctx.needs_input_grad[0]:
            # adjoint of the recurrence: a_k = g W_k; for k = K-1 .. 2: a_{k-1} += 2 L^T a_k, a_{k-2} -= a_k; gx = a_0 + L^T a_1
            lap_t = dict(cscale=s.derived("cheb_neg"), rscale=s.cheb_dinv)
            a = [ops.linear(g, ws[i], transposed=True) for i in range(k)]
            for i in range(k - 1, 1, -1):
                ops.csr_aggregate(a[i], s.out_ptr, s.out_dst, ell=s.out_ell, alpha=2.0, z=a[i - 1], beta=1.0,
                                  out=a[i - 1], **lap_t)
                a[i - 2] = a[i - 2] - a[i]
            gx = a[0] if k == 1 else ops.csr_aggregate(a[1], s.out_ptr, s.out_dst, ell=s.out_ell, z=a[0], beta=1.0,
                                                       out=a[0], **lap_t)
            if ctx.x_gate_scale is not None:
                gx = ops.relu_dropout_bwd(gx, terms[0], ctx.x_gate_scale)
        return (gx, gb, None, None, None, None, None, None, *gws)


def cheb_layer(x, ws, bias, struct, relu=False, drop_p=0.0, seed=0, defer_mask=False, x_gate_scale=None):
    node = _ChebLayer if 2 <= len(ws) <= 3 else _ChebLayerRecurrence
    return node.apply(x, bias, struct, relu, drop_p, seed, defer_mask, x_gate_scale, *ws)


class _SAGELayer(Function):
    """SAGEConv as ONE autograd node, PROJECT-FIRST: y = act(mean_in(x W_l^T) + x W_r^T + b) -- one GEMM produces both
    projections from one read of x, the mean runs at the output width with "+ x W_r^T + b" in its epilogue (PyG
    aggregates at the input width and projects afterwards: same algebra, different fp32 rounding order).
    Backward: g_p = mean_in^T(g); one weight-gradient pass x^T [g_p | g]; gx = g_p W_l + g W_r in one GEMM."""

    @staticmethod
    def forward(ctx, x, wl, bl, wr, struct: GraphStructure, relu, drop_p, seed, defer_mask, x_gate_scale, pre=None, pool=None):
        s = struct
        x = _padded_rows(ops.rowmajor(x))
        o = wl.shape[0]
        ow = (o + 3) // 4 * 4
        wl, wr = wl.contiguous(), wr.contiguous()
        p, r = pre if pre is not None else _fan_out(x, [wl, wr], [None, bl])   # the bias rides on the root term
        y = ops.csr_aggregate(p, s.in_ptr, s.in_src, ell=s.in_ell, rscale=s.sage_rinv, dself=s.derived("sage_dself"),
                              z=r, beta=1.0, relu=relu, drop_p=drop_p, seed=seed, out=r, pool=pool)
        ctx.struct, ctx.relu, ctx.drop_p, ctx.has_bias, ctx.dims = s, relu, drop_p, bl is not None, (o, ow)
        ctx.x_gate_scale = x_gate_scale
        ctx.save_for_backward(x, wl, wr, y if ((relu or drop_p > 0) and not defer_mask) else None)
        return y

    @staticmethod
    def backward(ctx, g, blocks_only=False):
        x, wl, wr, y = ctx.saved_tensors
        s, (o, ow) = ctx.struct, ctx.dims
        n, i = x.shape
        if isinstance(g, ops.PooledGrad) and y is None:
            gp, g = ops.pooled_grad_aggregate(g, s.out_ptr, s.out_dst, s.out_ell, s.sage_rinv, dself=s.derived("sage_dself"))
        else:
            if isinstance(g, ops.PooledGrad):
                g = g.materialise()
            g = _padded_rows(_mask_grad(g, y, ctx.drop_p))
            gp = ops.csr_aggregate(g, s.out_ptr, s.out_dst, ell=s.out_ell, cscale=s.sage_rinv, dself=s.derived("sage_dself"))
        if blocks_only:
            return [gp, g]
        gw = torch.empty((2 * ow, i), dtype=torch.float32, device=x.device)
        gb = torch.empty(2 * ow, dtype=torch.float32, device=x.device)
        ops.linear_wgrad_parts([gp, g], x, gw, gb)
        gx = None
        if ctx.needs_input_grad[0]:
            gated = ctx.x_gate_scale is not None
            gx = _fan_in_t([gp, g], [wl, wr], i, gate=x if gated else None, gate_scale=ctx.x_gate_scale if gated else 1.0)
        return gx, gw[:o], gb[ow:ow + o] if ctx.has_bias else None, gw[ow:ow + o], None, None, None, None, None, None


def sage_layer(x, wl, bl, wr, struct, relu=False, drop_p=0.0, seed=0, defer_mask=False, x_gate_scale=None):
    return _SAGELayer.apply(x, wl, bl, wr, struct, relu, drop_p, seed, defer_mask, x_gate_scale)


class _GCNLayer(Function):
    """y = act(D^-1/2 (A+I) D^-1/2 (x W^T) + b) as ONE autograd node.

    Forward: the projection writes h' = dinv * (x W^T) (row scale fused in the GEMM epilogue), so the aggregation
    needs no per-edge scalar: y = act(dinv * (sum_e h'[src_e] + h'[i]) + b).  Backward: ReLU/dropout mask from y,
    then the same symmetric-normalised aggregation on the transposed CSR, then the two GEMM gradients."""

    @staticmethod
    def forward(ctx, x, w, bias, struct: GraphStructure, relu, drop_p, seed, defer_mask, x_gate_scale, pre=None, pool=None):
        x = ops.rowmajor(x)
        dinv = struct.gcn_dinv
        h = pre if pre is not None else ops.linear(x, w.contiguous(), rowscale=dinv)
        y = ops.csr_aggregate(h, struct.in_ptr, struct.in_src, ell=struct.in_ell, rscale=dinv, dself=dinv, bias=bias, relu=relu,
                              drop_p=drop_p, seed=seed, pool=pool)
        ctx.struct, ctx.relu, ctx.drop_p, ctx.x_gate_scale = struct, relu, drop_p, x_gate_scale
        ctx.save_for_backward(x, w, y if ((relu or drop_p > 0) and not defer_mask) else None)
        return y

    @staticmethod
    def backward(ctx, g, blocks_only=False):
        x, w, y = ctx.saved_tensors
        s = ctx.struct
        fused = (not blocks_only and ctx.needs_input_grad[0] and ctx.needs_input_grad[2] and ctx.x_gate_scale is not None
                 and max(w.shape) <= 12)
        if isinstance(g, ops.PooledGrad) and y is None:
            # the fused backward below reads g only for the bias gradient: its column sums come from the bits, g is never written
            pooled = g
            gh, g = ops.pooled_grad_aggregate(pooled, s.out_ptr, s.out_dst, s.out_ell, s.gcn_dinv, rscale=s.gcn_dinv, dself=s.derived("gcn_dself"), want_g=not fused)
            if fused and ops._fused_bwd_ok(gh, x):
                gx, gw, _, ctx.gx_colsum = ops.linear_bwd_fused(gh, x, w.contiguous(), gate_scale=ctx.x_gate_scale)
                return gx, gw, ops.pooled_grad_colsum(pooled), None, None, None, None, None, None
            if g is None:
                g = pooled.materialise()
        else:
            if isinstance(g, ops.PooledGrad):
                g = g.materialise()
            g = _mask_grad(g, y, ctx.drop_p)
            gh = ops.csr_aggregate(g, s.out_ptr, s.out_dst, ell=s.out_ell, cscale=s.gcn_dinv, rscale=s.gcn_dinv,
                                   dself=s.derived("gcn_dself"))
        if blocks_only:
            return [gh, _padded_rows(g)]
        if (ctx.needs_input_grad[0] and ctx.needs_input_grad[2] and ctx.x_gate_scale is not None
                and max(w.shape) <= 12 and ops._fused_bwd_ok(gh, g, x)):
            # hidden layer of width <= 12 (conv2): gated data gradient, weight and bias gradient from ONE pass over gh, x, g
            gx, gw, gb, ctx.gx_colsum = ops.linear_bwd_fused(gh, x, w.contiguous(), gb_src=g, gate_scale=ctx.x_gate_scale)
            return gx, gw, gb, None, None, None, None, None, None
        gx = None
        if ctx.needs_input_grad[0]:
            gated = ctx.x_gate_scale is not None
            gx = ops.linear(gh, w.contiguous(), transposed=True, gate=x if gated else None,
                            gate_scale=ctx.x_gate_scale if gated else 1.0)
        gw = gb = None
        if ctx.needs_input_grad[2]:
            # the bias gradient sum_n g[n,:] is the ones-column of a weight-gradient pass: give that pass a second
            # column block [gh | g] instead of sweeping g with a separate reduction
            o, i = w.shape
            ow = (o + 3) // 4 * 4
            gw2 = torch.empty((2 * ow, i), dtype=torch.float32, device=x.device)
            gb2 = torch.empty(2 * ow, dtype=torch.float32, device=x.device)
            ops.linear_wgrad_parts([gh, g], x, gw2, gb2)
            gw, gb = gw2[:o], gb2[ow:ow + o]
        elif ctx.needs_input_grad[1]:
            gw = torch.empty_like(w, memory_format=torch.contiguous_format)
            ops.linear_wgrad(gh, x, gw, None)
        return gx, gw, gb, None, None, None, None, None, None


def gcn_layer(x, w, bias, struct, relu=False, drop_p=0.0, seed=0, defer_mask=False, x_gate_scale=None):
    return _GCNLayer.apply(x, w, bias, struct, relu, drop_p, seed, defer_mask, x_gate_scale)


class _SegmentMean(Function):
    @staticmethod
    def forward(ctx, x, struct: GraphStructure):
        ctx.struct = struct
        ctx.n = x.shape[0]
        return ops.segment_mean(ops.rowmajor(x), struct.graph_ptr, struct.num_graphs)

    @staticmethod
    def backward(ctx, g):
        return ops.segment_mean_bwd(ops.rowmajor(g), ctx.struct.graph_ptr, ctx.n), None


def segment_mean(x, struct):
    return _SegmentMean.apply(x, struct)


class _TransformerConv(Function):
    """out = attention(x W_qkvs^T + b) + skip as ONE autograd node: fused projection (one read of x), edge softmax
    with optional dropout on the attention weights, and on the way back the destination/source-side attention
    gradient kernels followed by the projection's data and weight gradients."""

    @staticmethod
    def forward(ctx, x, struct: GraphStructure, heads, channels, drop_p, seed, *wb):
        # wb = (w, b): the fused [4 H C, in] projection, or the reference's eight parameters (query, key, value, skip: weight, bias
        # each) as they are -- their concatenation and the per-head padding then come from ONE launch (ops.pad_head_rows_parts)
        # instead of two torch.cat and a padding launch
        x = ops.rowmajor(x)        # a RowsOf (rows of the device-resident dataset) stays one: the projection reads through its row map
        parts = len(wb) == 8
        ctx.parts = parts
        if parts:
            ws_, bs_ = list(wb[0::2]), list(wb[1::2])
            on_gpu = ws_[0].is_cuda
            fused = lambda pitch: (ops.pad_head_rows_parts(ws_, bs_, heads, channels, pitch) if on_gpu
                                   else _pad_heads(torch.cat(ws_, 0), torch.cat(bs_, 0), 4 * heads, channels, pitch))
            w = None
        else:
            w, b = wb
            w = w.contiguous()
            on_gpu = w.is_cuda
        e = struct.edge_count()
        # a structure whose rows share their sources (ASAPooling's coarsened graphs of large circuits): its long rows as dense blocks,
        # the edge softmax on the matrix cores (csrc/dense_block.hip)
        dense = (_DENSE_BLOCKS and struct.blocked and on_gpu and struct.out_eid is None and channels < 16
                 and ops.dense_attention_supported(heads, channels, 16))
        if not any(ctx.needs_input_grad) and drop_p == 0.0:  # inference: no statistics kept
            if parts:
                w, b = fused(channels)
            return ops.transformer_attention(ops.linear(x, w, b), struct.in_ptr, struct.in_src, struct.loops, heads, channels)
        # Training: a head's channels at a pitch of 16 inside q / k / v / skip (the reference's 15: every gathered segment becomes an
        # aligned 64-byte piece).  The projection writes that layout by itself when its weight and bias rows are padded the same way
        # (zero rows: the pads of qkvs are zeros, the gradient of a pad row is exactly zero); w itself stays [4 H C, in].
        cp = _ATTN_PITCH if (_ATTN_PITCH > channels and _ATTN_PITCH - channels < 4 and on_gpu) else 0
        if dense:
            cp = 16 if channels < 16 else 0
        ctx.cp = cp
        ctx.dense = dense
        if parts:
            w_used, b_used = fused(cp if cp else channels)
        else:
            w_used, b_used = _pad_heads(w, b, 4 * heads, channels, cp) if cp else (w, b)
        qkvs = ops.linear(x, w_used, b_used)
        # a structure without out_eid (ASAPooling's coarsened graphs: no parallel edges) takes the recomputed backward, whose dropout
        # draws are keyed by (destination, head, source)
        pair_key = struct.out_eid is None
        # the side table of a graph of short rows (circuit DAGs: the arena builds it with the batch); a coarsened graph's rows are long
        ell = struct.in_ell if struct.out_eid is not None else None
        if dense:
            out, attn, m, den = ops.dense_attention_train(qkvs, struct.in_ptr, struct.in_src, struct.loops, e, heads, channels,
                                                         struct.dense_plan("in"), drop_p=drop_p, seed=seed, head_pitch=cp,
                                                         side=_dense_side(w_used.device))
        else:
            out, attn, m, den = ops.transformer_attention_train(qkvs, struct.in_ptr, struct.in_src, struct.loops, e, heads,
                                                               channels, drop_p, seed, pair_key=pair_key, ell=ell, head_pitch=cp)
        ctx.struct, ctx.cfg = struct, (e, heads, channels, drop_p, seed, pair_key)
        ctx.x_rows_of = isinstance(x, ops.RowsOf)
        if ctx.x_rows_of:
            ctx.save_for_backward(x.base, x.rows, w_used, qkvs, attn, m, den)
        else:
            ctx.save_for_backward(x, w_used, qkvs, attn, m, den)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.x_rows_of:
            base, rows, w, qkvs, attn, m, den = ctx.saved_tensors
            x = ops.RowsOf(base, rows)
        else:
            x, w, qkvs, attn, m, den = ctx.saved_tensors
        e, heads, channels, drop_p, seed, pair_key = ctx.cfg
        cp = ctx.cp
        if ctx.dense:
            st = ctx.struct
            gqkvs = ops.dense_attention_bwd(qkvs, g, attn, m, den, st, e, heads, channels, st.dense_plan("in"), st.dense_plan("out"),
                                            drop_p=drop_p, seed=seed, head_pitch=cp, side=_dense_side(w.device))
        else:
            gqkvs = ops.transformer_attention_bwd(qkvs, g, attn, m, den, ctx.struct, e, heads, channels, drop_p, seed, pair_key=pair_key,
                                                  head_pitch=cp)
        gx = ops.linear(gqkvs, w, transposed=True) if ctx.needs_input_grad[0] else None       # w: the (padded) weight the forward used
        gw = torch.empty_like(w)
        gb = torch.empty(w.shape[0], dtype=w.dtype, device=w.device)
        ops.linear_wgrad(gqkvs, x, gw, gb)
        if cp:                     # the real rows of the padded gradients
            gw, gb = ops.unpad_head_rows(gw, gb, 4 * heads, channels, cp)
        if ctx.parts:              # the four parameters' gradients: row blocks of the fused one
            hc = heads * channels
            grads = []
            for k in range(4):
                grads += [gw[k * hc:(k + 1) * hc], gb[k * hc:(k + 1) * hc]]
            return (gx, None, None, None, None, None, *grads)
        return gx, None, None, None, None, None, gw, gb


def _pad_heads(w, b, groups, channels, cp):
    """Weight and bias of a projection whose ``groups`` blocks of ``channels`` rows each are spread to a pitch of ``cp`` rows (zero
    rows between: the projection then writes its output at that pitch, pads zero; the gradient of a pad row is exactly zero)."""
    if cp <= channels:
        return w, b
    if w.is_cuda:
        return ops.pad_head_rows(w, b, groups, channels, cp)
    wp = w.new_zeros((groups * cp, w.shape[1]))
    wp.view(groups, cp, -1)[:, :channels].copy_(w.view(groups, channels, -1))
    bp = None
    if b is not None:
        bp = b.new_zeros(groups * cp)
        bp.view(groups, cp)[:, :channels].copy_(b.view(groups, channels))
    return wp, bp


# the long rows of ASAPooling's coarsened graphs as dense blocks: TransformerConv's edge softmax over them on the f32 matrix cores
# (csrc/dense_block.hip); MLQEM_DENSE_BLOCKS=0: the per-edge kernels for every row (A/B runs, tests/test_gpu_dense_blocks.py)
_DENSE_BLOCKS = os.environ.get("MLQEM_DENSE_BLOCKS", "1") == "1"
# True: the per-edge kernel over the rows outside the blocks on a side stream, beside the block kernel (disjoint rows).  Measured on
# 64 100-qubit circuits: slower -- eagerly 117.7 us against 109.4 on one stream (forward), 305.5 against 290.7 (backward); inside the
# captured step 7.12 ms against 7.00 (scripts/family_b_step.py): the per-edge kernel's 11 k workgroups fill the chip either way, the
# fork and the join cost more than the overlap returns.  Kept as a switch of the ops (tests/test_gpu_dense_blocks.py runs both).
_DENSE_TWO_STREAMS = False


def _dense_side(device):
    return _branch_streams(device)[0] if _DENSE_TWO_STREAMS else None
# channel pitch of a head inside q / k / v / skip in training (0: compact heads, the layout of rounds 1-3; the parity test of the two
# layouts sets it)
_ATTN_PITCH = 16


def transformer_conv(x, w, b, struct, heads, channels, drop_p=0.0, seed=0):
    """``w`` / ``b``: the fused [4 H C, in] projection and its bias, or lists of the four parts (query, key, value, skip)."""
    if isinstance(w, (list, tuple)):
        wb = [t for pair in zip(w, b) for t in pair]
        return _TransformerConv.apply(x, struct, heads, channels, drop_p, seed, *wb)
    return _TransformerConv.apply(x, struct, heads, channels, drop_p, seed, w, b)


# Which form of the coarsening S^T (A S) a pooling takes, by the batch's graph sizes: the sync-free dense form when every graph pools to
# <= 512 clusters, the sorted-list form for larger graphs, the wave-per-cluster bit-matrix form when the lists would not fit 32-bit
# places, the general two-hop path (four device->host size reads) otherwise.  All forms yield identical arrays; these module
# attributes exist so that the tests can force one form and compare it with another (tests/test_gpu_family_b.py).
_ASAP_DENSE = True
# ASAPooling's forward on a graph of short rows as one fused pass (tests/test_gpu_family_b.py flips it: both forms, same results)
_ASAP_FUSED = True
_ASAP_ROWS = True
_ASAP_LISTS = True
# True: the list coarsening also links every out-entry to its in-CSR twin (out_eid; 0.55 ms for 64 100-qubit circuits) and the
# backward kernels on the coarsened graph take the stored form; default: no out_eid, recomputed form
_ASAP_LINK = False


# Graph boundaries of pooled batches on the device, by content.  Batches of a training run repeat their size patterns (the
# size-stratified batches of train.StratifiedBatches always do), and inside a hipGraph capture a host->device copy from
# pageable memory is not allowed: the eager pass that precedes a capture (train.BucketedTrainer runs a pattern eagerly the
# first time it sees it) leaves the array here.  Two tiers: an LRU of what eager calls made, and the arrays a CAPTURED graph
# reads -- their addresses are baked into the graph, so they are owned here for good and never evicted (evicting them handed
# their memory back to the allocator while replays still read it as graph boundaries).
from collections import OrderedDict

_ptr_cache = OrderedDict()
_ptr_pinned = {}
_PTR_CACHE_MAX = 256


def _device_ptr(host_i32, device):
    key = (host_i32.tobytes(), str(device))
    t = _ptr_pinned.get(key)
    if t is not None:
        return t
    capturing = torch.cuda.is_current_stream_capturing()
    t = _ptr_cache.get(key)
    if capturing:
        if t is None:
            raise RuntimeError("pooled graph boundaries of this size pattern are not on the device yet: run the batch eagerly once "
                               "before capturing it (a host->device copy cannot be part of a hipGraph capture)")
        _ptr_pinned[key] = _ptr_cache.pop(key)
        return t
    if t is None:
        while len(_ptr_cache) >= _PTR_CACHE_MAX:
            _ptr_cache.popitem(last=False)
        t = _ptr_cache[key] = torch.from_numpy(host_i32).to(device, non_blocking=True)
    else:
        _ptr_cache.move_to_end(key)
    return t


class _ASAPool(Function):
    """ASAPooling as ONE autograd node.  Differentiable output: x_out = x'[perm] * fitness[perm]; the pooled structure
    and ``perm`` are data-dependent side results handed back through ``holder``."""

    @staticmethod
    def forward(ctx, x, lin_w, lin_b, att_w, att_b, l1_w, l1_b, l2_w, l3_w, l3_b, struct: GraphStructure, ratio, slope,
                holder):
        s = struct
        x = ops.rowmajor(x)
        d, n = x.shape[1], s.num_nodes
        # the parameters' small-tensor algebra in one launch (csrc/asap.hip asap_compose_kernel): the halves of att_w, LEConv's three
        # one-wide projections as one [3, D] matrix (lin2 has no bias), and the query projection composed into the score projection
        w_comp, b_comp, att_q, att_x, w3, b3 = ops.asap_compose(lin_w, lin_b, att_w, att_b, l1_w, l1_b, l2_w, l3_w, l3_b)
        # the input graph's rows share their sources (it is itself a coarsened graph): its long rows as dense blocks
        # (csrc/dense_pool.hip), on the same plans TransformerConv's edge softmax built on this graph
        dense = _DENSE_BLOCKS and s.blocked and s.out_eid is None and ops.dense_pool_fits(x)
        ctx.dense = dense
        # ... or, a graph of short rows (the circuit DAGs: the arena hands their side table along): everything up to the fitness
        # projections in one pass over the rows (csrc/attn.hip asap_scores_fused_kernel)
        fused = _ASAP_FUSED and not dense and x.is_cuda and d <= 64 and s.in_ell is not None
        stat = None
        if fused:
            xq_raw, a_dst, c_src, x_new, pqr = ops.asap_scores_fused(x, s.in_ptr, s.in_src, w_comp, b_comp, att_x, w3, b3, slope)
            fitness = ops.leconv_fitness(pqr, s.in_ptr, s.in_src)
        elif dense:
            xq_raw = ops.dense_segment_max(x, s.in_ptr, s.in_src, s.dense_plan("in"))
        else:
            xq_raw = ops.csr_segment_max(x, s.in_ptr, s.in_src, ell=s.in_ell)
        if not fused:
            # ASAPooling's query x_q = lin(segmax) feeds ONLY the one-wide score a_i = att_q . x_q[i] + att_b (SURVEY appendix
            # B.2 steps 2-3): a_i = (att_q W) . segmax[i] + (att_q . b + att_b) -- one row dot of the segment max against a composed
            # 45-vector.  x_q [N, D] is never formed (a [N,D]x[D,D] GEMM forward; a data GEMM and a [D,D] weight-gradient pass
            # backward), the gradients of lin follow from the composed vector's by the chain rule on D x D tensors
            # (asap_compose above, asap_compose_bwd in the backward: deterministic, capturable).
            # (one-wide and three-wide projections into COMPACT outputs: a [N, 1] matrix with a row pitch of one float is the vector the
            # edge kernels take -- the padded default cost a strided copy per projection)
            a_dst = ops.linear(xq_raw, w_comp, b_comp, out=torch.empty((n, 1), dtype=torch.float32, device=x.device))[:, 0]
        if not fused:
            c_src = ops.linear(x, att_x, out=torch.empty((n, 1), dtype=torch.float32, device=x.device))[:, 0]
            if dense:
                x_new, stat = ops.dense_softmax_aggregate(x, s.in_ptr, s.in_src, a_dst, c_src, slope, s.dense_plan("in"))
            else:
                x_new = ops.csr_softmax_aggregate(x, s.in_ptr, s.in_src, a_dst, c_src, slope)
            fitness = ops.leconv_fitness(ops.linear(x_new, w3, b3, out=torch.empty((n, 3), dtype=torch.float32, device=x.device)), s.in_ptr, s.in_src,
                                         long_rows=dense)      # (the coarsened graph: a 16-lane group per row)
        plan = getattr(s, "pool_plan", None)
        if plan:
            # a size-stable batch (train.BucketedTrainer): the per-graph sizes stay on the device.  The pooled boundaries come from a
            # launch, the number of kept nodes is a function of the bucket (the batch's filler graphs are sized for that), and the
            # largest graph before / after pooling is known by a bound -- nothing here depends on the size SEQUENCE of the batch
            k_total, nmax, kmax = (int(v) for v in plan[0])
            new_ptr = ops.pool_keep_ptr(s.graph_ptr, s.num_graphs, ratio)
            keep = {"b": s.num_graphs, "k": k_total, "kmax": kmax, "nmax": nmax}
            sizes = None
            have = s.num_graphs > 0
        else:
            # k_g = ceil(ratio * n_g) evaluated in float32 like PyG's topk (float32 tensor times a python scalar)
            sizes = np.asarray(s.graph_sizes, dtype=np.int64)
            keep = np.ceil(sizes.astype(np.float32) * np.float32(ratio)).astype(np.int64)
            new_ptr_host = np.zeros(len(keep) + 1, dtype=np.int64)
            np.cumsum(keep, out=new_ptr_host[1:])
            k_total = int(new_ptr_host[-1])
            new_ptr = _device_ptr(new_ptr_host.astype(np.int32), x.device)
            have = len(keep) > 0
            nmax, kmax = (int(sizes.max()), int(keep.max())) if have else (0, 0)
        # ... with the backward's slot[] (cluster id of every kept centre, -1 elsewhere; the coarsenings read the same map) from the same launches
        perm, slot_fwd = ops.segment_topk(fitness, s.graph_ptr, new_ptr, n, s.num_graphs, k_total, max_graph_nodes=min(nmax, n), with_slot=True)
        x_out = ops.gather_scale_rows(x_new, perm, fitness)
        use_dense, use_rows, use_lists, link = _ASAP_DENSE, _ASAP_ROWS, _ASAP_LISTS, _ASAP_LINK   # the switches as they stand now: build() may run later

        def build():
            dense_ok = use_dense and have and kmax <= ops.asap_dense_max_k()
            if dense_ok:
                # small graphs: the pooled adjacency as per-graph bit matrices in LDS -- no device->host copy anywhere
                csr, slot, cap = ops.asap_coarsen_dense(s.in_ptr, s.in_src, s.out_ptr, s.out_dst, s.graph_ptr, new_ptr, perm, n, keep,
                                                        slot=slot_fwd)
                num_edges = cap     # an upper bound: the true count stays on the device (in_ptr[k_total])
            done = None
            if not dense_ok and use_rows and use_lists and have and kmax <= ops.asap_lists_max_k():
                # large graphs: per-node cluster lists, a thread per cluster gathers its candidates, persistent waves sort them through
                # LDS bitsets; no host read when the structure carries a capacity.  None: too many candidates for this form
                done = ops.asap_coarsen_lists(s.in_ptr, s.in_src, s.out_ptr, s.out_dst, s.graph_ptr, new_ptr, perm, n, s.edge_count(), keep,
                                              capacity=getattr(s, "coarse_capacity", None), link=link, slot=slot_fwd)
            if dense_ok:
                pass
            elif done is not None:
                csr, slot, num_edges = done
            elif (use_rows and have and nmax + 2 * kmax + 96 <= ops.asap_rows_max_bits()):
                # large graphs: one wave per cluster, bitsets in LDS, no sort; one 4-byte read (the edge total)
                csr, slot, num_edges = ops.asap_coarsen_rows(s.in_ptr, s.in_src, s.out_ptr, s.out_dst, s.graph_ptr, new_ptr, perm,
                                                             n, sizes, keep, capacity=getattr(s, "coarse_capacity", None))
            else:
                ei, slot = ops.asap_coarsen(s.in_ptr, s.in_src, s.out_ptr, s.out_dst, perm, n, return_slot=True)
                csr = ops.csr_build(ei, k_total)
                num_edges = int(ei.shape[1])
            return (csr[0], csr[1], csr[2], csr[3], csr[4], num_edges, csr.out_eid), slot

        # the coarsened connectivity S^T A S waits until a layer reads it (GraphStructure.deferred); the backward's slot[] does not
        # depend on it
        slot = slot_fwd
        holder["structure"] = GraphStructure.deferred(k_total, new_ptr, s.num_graphs, lambda: build()[0], graph_sizes=None if plan else keep)
        if plan:
            holder["structure"].pool_plan = plan[1:]             # the next pooling's level
            holder["structure"].num_real = s.num_real
        if (_DENSE_BLOCKS and use_rows and use_lists and not link and have and ops.asap_dense_max_k() < kmax):
            # large graphs (the list coarsening's): clusters whose centres are close in program order share their neighbours, so
            # the layers that read this graph take its long rows 16 at a time in the order of their centres' node index
            def block_order(slot=slot, gptr=s.graph_ptr, b=s.num_graphs):
                # (a bound on the id range of one block's entries: its rows may straddle two graphs)
                return ops.tile_order_by_position(slot, gptr, new_ptr, b, k_total), 2 * kmax + ops.ORDER_SPAN_SLACK

            holder["structure"].set_block_order(block_order)
        holder["perm"] = perm
        ctx.struct, ctx.slope, ctx.d = s, slope, d
        ctx.save_for_backward(x, xq_raw, w_comp, a_dst, c_src, x_new, fitness, slot, lin_w, att_w, w3, lin_b, stat if dense else None)
        return x_out

    @staticmethod
    def backward(ctx, g_out):
        x, xq_raw, w_comp, a_dst, c_src, x_new, fitness, slot, lin_w, att_w, w3, lin_b, dense_stat = ctx.saved_tensors
        s, d = ctx.struct, ctx.d
        e = s.edge_count()
        dev = x.device
        # x_out = x'[perm] * f[perm]: g_f = g_out . x' first (the fitness backward needs it), g_x' = g_out f + g_pqr W3 in one store after it
        # (padded rows of at most 64 channels; otherwise g_x' = g_out f is stored with g_f and a [N,3]x[3,D] GEMM adds g_pqr W3 to it)
        gfit = ops.gather_rows_dot(g_out, x_new, slot) if (x.is_cuda and _ASAP_FUSED) else None
        split = gfit is not None
        if not split:
            gxnew, gfit = ops.gather_scale_rows_bwd(g_out, x_new, fitness, slot)
        # f = sigmoid(LEConv(x')) on scalars pqr = x' W3^T + b3
        if ctx.dense:        # the long rows of the coarsened graph: a wave each (the plan of the out-structure lists them)
            gpqr = ops.dense_leconv_fitness_bwd(gfit, fitness, s.in_ptr, s.out_ptr, s.out_dst, s.dense_plan("out"))
        else:
            gpqr = ops.leconv_fitness_bwd(gfit, fitness, s.in_ptr, s.out_ptr, s.out_dst)
        if split:
            gxnew = ops.scatter_scale_rank(g_out, fitness, slot, gpqr, w3, x_new.shape[0], x_new.shape[1])
        else:
            ops.linear(gpqr, w3, transposed=True, out=gxnew, accumulate=True)
        # x' = sum_e softmax(LeakyReLU(a_i + c_j)) x_j
        # ... its destination-side walk also counts the ties of the segment max below (same x, same entries)
        att_x = att_w[:, d:]                 # (a view: its one row is contiguous)
        # c = x att_x^T: its gradient g_c (x) att_x rides in the source-side kernel's store of gx (it computes g_c itself) instead of
        # being a read-modify-write pass over gx
        dense = ctx.dense and ops.dense_pool_fits(gxnew, x_new, xq_raw)
        fuse_max = False
        if dense:
            gx, g_a, g_c, ties = ops.dense_softmax_aggregate_bwd(x, x_new, gxnew, s, e, a_dst, c_src, ctx.slope, dense_stat, s.dense_plan("in"),
                                                                 s.dense_plan("out"), xq_raw, gx_rank1=att_x[0])
        else:
            # the stored form (a graph with out_eid: the circuit DAGs) carries the segment max's backward in its source-side walk
            fuse_max = _ASAP_FUSED and s.out_eid is not None and d <= 128
            gx, g_a, g_c, ties = ops.csr_softmax_aggregate_bwd(x, x_new, gxnew, s, e, a_dst, c_src, ctx.slope, xmax=xq_raw, gx_rank1=att_x[0],
                                                               fuse_max_col=w_comp[0].contiguous() if fuse_max else None)
        # The three tiny weight gradients of the pooling in ONE pass over their operands (csrc/family_b_bwd.hip rank_grad_*: two
        # launches where three linear_wgrad calls were six): gw3 [3, D] = gpqr^T x' and gb3 = its column sums (LEConv's projections);
        # g_att_x [1, D] = g_c^T x (c = x att_x^T); g_w_comp [1, D] = g_a^T segmax and g_att_b = sum g_a -- a = xq_raw w_comp^T + b_comp
        # with w_comp = att_q W, b_comp = att_q . b + att_b: the gradients of lin and of att's query half follow by the chain rule on
        # D x D tensors (asap_compose_bwd); the segment max's gradient g_a (x) w_comp is formed inside its backward kernel
        if x.is_cuda:
            (gw3, gb3), (g_att_x, _), (g_w_comp, g_att_b) = ops.rank_grad([(gpqr, x_new), (g_c, x), (g_a, xq_raw)])
        else:
            gw3, gb3 = torch.empty_like(w3), torch.empty(3, dtype=torch.float32, device=dev)
            ops.linear_wgrad(gpqr, x_new, gw3, gb3)
            g_att_x = torch.empty((1, d), dtype=torch.float32, device=dev)
            ops.linear_wgrad(g_c.unsqueeze(1), x, g_att_x, None)
            g_w_comp, g_att_b = torch.empty_like(w_comp), torch.empty(1, dtype=torch.float32, device=dev)
            ops.linear_wgrad(g_a.unsqueeze(1), xq_raw, g_w_comp, g_att_b)
        if dense:
            ops.dense_segment_max_bwd_(gx, x, xq_raw, s, ties, (g_a, w_comp[0].contiguous()), s.dense_plan("out"))
        elif not fuse_max:
            ops.csr_segment_max_bwd_(gx, x, xq_raw, None, s, ties=ties, gmax_rank1=(g_a, w_comp[0].contiguous()))
        g_lin_w, g_lin_b, g_att_w = ops.asap_compose_bwd(g_w_comp, g_att_b, lin_w, lin_b, att_w, g_att_x)
        return (gx, g_lin_w, g_lin_b, g_att_w, g_att_b, gw3[0:1], gb3[0:1], gw3[1:2], gw3[2:3], gb3[2:3],
                None, None, None, None)


def asap_pool(x, mod, struct):
    """Runs ASAPooling with the parameters of ``mod``; returns (x_out, pooled structure, perm)."""
    holder = {}
    g = mod.gnn_score
    x_out = _ASAPool.apply(x, mod.lin.weight, mod.lin.bias, mod.att.weight, mod.att.bias, g.lin1.weight, g.lin1.bias,
                           g.lin2.weight, g.lin3.weight, g.lin3.bias, struct, mod.ratio, mod.negative_slope, holder)
    return x_out, holder["structure"], holder["perm"]


# ------------------------------------------------------------------------------------------------------------------
# Family A's graph part as ONE autograd node.  torch.autograd.Function.apply costs ~30 us of host time per node and
# direction; the model has 7 conv layers + 3 pools, which made the host enqueue a train step in 2.0 ms (a batch of 32
# small graphs is host-bound).  This node runs the very same layer code -- each layer's forward/backward static methods
# are called with a private context object -- so the arithmetic and the launch sequence do not change.
_side_streams = {}


_BRANCH_STREAMS_MIN_NODES = int(os.environ.get("MLQEM_BRANCH_STREAMS_MIN_NODES", "200000"))


def _branch_streams(device, num_nodes=None):
    """Two side streams per device for the Cheb and SAGE branches (created once).  ``MLQEM_SINGLE_STREAM=1`` keeps all
    three branches on the caller's stream: kernels then run one after the other, which is what a per-kernel profile
    needs (durations of overlapping kernels stretch each other; scripts/make_profiles.sh uses it for attribution).
    A SMALL batch stays on one stream too: its kernels are launch latency, and the forks and joins of a three-stream graph cost
    more than the branches' overlap returns (32 four-qubit circuits, captured: 0.245-0.279 ms on three streams, 0.199 on one;
    2.4 M nodes: 1.48 against 1.75 ms the other way round; 11.3 M nodes: 5.86-6.0 against 6.03-6.13)."""
    if os.environ.get("MLQEM_SINGLE_STREAM", "0") == "1" or (num_nodes is not None and num_nodes < _BRANCH_STREAMS_MIN_NODES):
        cur = torch.cuda.current_stream(device)
        return (cur, cur)
    key = torch.device(device).index
    if key not in _side_streams:
        _side_streams[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
    return _side_streams[key]


class _LayerCtx:
    """The part of a torch.autograd.Function context the layer nodes use."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class _FamilyAGraph(Function):
    """pooled [B, 3] = [GCN x3 | Cheb x2 | SAGE x2 branch, each mean-pooled] of the node features (01_ngem.ipynb cell [9]).

    Parameters, in order: conv1.W, conv1.b, conv2.W, conv2.b, conv3.W, conv3.b, cheb1.W0..W2, cheb1.b, cheb2.W0, cheb2.W1,
    cheb2.b, sage1.Wl, sage1.bl, sage1.Wr, sage2.Wl, sage2.bl, sage2.Wr (19 tensors).

    THE LAST CONV OF EVERY BRANCH IS FOLDED INTO ITS POOL.  The notebook's model puts no non-linearity between conv3 /
    cheb_conv2 / sage_conv2 and global_mean_pool, and both are linear, so with P the layer's propagation matrix and
    t = P^T 1 its column sums (a structural scalar per node, ``GraphStructure.colsum``):

        mean_pool(P (h W^T) + b)                 = wmean_t(h) W^T + b                    GCN   (P = D^-1/2 (A+I) D^-1/2)
        mean_pool(h W_0^T + (L^ h) W_1^T + b)    = mean(h) W_0^T + wmean_t(h) W_1^T + b   Cheb, K = 2
        mean_pool(M h W_l^T + b_l + h W_r^T)     = wmean_t(h) W_l^T + b_l + mean(h) W_r^T SAGE  (M = in-edge mean)

    with wmean_t(h)[g] = (1/n_g) sum_{j in g} t_j h_j.  One pass over h (``ops.segment_pool``) replaces a [N,10] -> [N,1]
    projection, a width-1 aggregation and a pool; the backward is one pass that writes gh = (g_mean + t g_wmean) / n_g
    with the hidden layer's ReLU/dropout mask applied (``ops.segment_pool_bwd``), and the weight gradients are [B, 10]
    matrix products.  Same algebra as the reference in a different fp32 summation order (1e-5 parity tests cover it);
    the width-1 kernels this removes ran at 20-35 % of the HBM peak and took 29 % of the step."""

    @staticmethod
    def forward(ctx, x, struct: GraphStructure, p1, p2, seed, *prm):
        (g1w, g1b, g2w, g2b, g3w, g3b, c1w0, c1w1, c1w2, c1b, c2w0, c2w1, c2b, s1l, s1b, s1r, s2l, s2b, s2r) = prm
        k1, k2 = 1.0 / (1.0 - p1), 1.0 / (1.0 - p2)
        T, Fa = True, False
        mk = lambda n_in, x_grad: _LayerCtx((x_grad,) + (T,) * (n_in - 1))
        L = ctx.layers = {}
        gptr, nb, n = struct.graph_ptr, struct.num_graphs, struct.num_nodes
        # The three branches are independent until the concatenation: each runs on its own HIP stream, so the tail of one
        # branch's kernels (hub-row waves keep a launch's last workgroups alive) is filled by the others' workgroups.
        main = torch.cuda.current_stream(x.device)
        side = ctx.side = _branch_streams(x.device, n)
        # everything the branches share must exist before they fork: the structure builds its side tables and derived
        # scalars on first use, and a table built by one branch's stream would be read by another's without an edge
        _ = (struct.in_ell, struct.out_ell, struct.gcn_dinv, struct.derived("gcn_dself"), struct.derived("sage_dself"),
             struct.derived("cheb_neg"))
        tg, tc, ts = struct.colsum("gcn"), struct.colsum("cheb"), struct.colsum("sage")
        # The first layer of every branch projects the SAME x: one GEMM over six output blocks reads x once instead of
        # three times (GCN: dinv * x W^T | Cheb: x (W_0 - W_2)^T + b, x W_1^T, x W_2^T | SAGE: x W_l^T, x W_r^T + b), and in the
        # backward ONE weight-gradient pass over x serves all seven gradient blocks (see backward).
        fuse = ctx.fuse = x.shape[1] <= _PARTS_MAX_COLS and 6 * ((g1w.shape[0] + 3) // 4 * 4) <= 96
        pre_g = pre_c = pre_s = None
        if fuse:
            xr = _padded_rows(ops.rowmajor(x))
            o = g1w.shape[0]
            blocks = [ops.padded_empty(xr.shape[0], o, x.device) for _ in range(6)]
            ws6 = [w.contiguous() for w in (g1w, c1w0, c1w1, c1w2, s1l, s1r)]
            ops.linear_parts([xr], ws6, blocks, w_minus=[None, ws6[3], None, None, None, None],
                             biases=[None, c1b, None, None, None, s1b],
                             rowscales=[struct.gcn_dinv, None, None, None, None, None])
            pre_g, pre_c, pre_s = blocks[0], blocks[1:4], blocks[4:6]
        for st in side:
            st.wait_stream(main)
        # GCN branch: args (x, w, bias, struct, relu, drop_p, seed, defer_mask, x_gate_scale)
        L["g1"] = mk(9, Fa); h = _GCNLayer.forward(L["g1"], x, g1w, g1b, struct, T, p1, seed + 1, T, None, pre=pre_g)
        # the pooled means of each branch's last hidden activation come out of the aggregation launch that writes it
        # (mlqem_csr_aggregate_pool_f32: the activation is not read a second time)
        # ... and since the backward reads that activation ONLY as the ReLU / dropout gate of the pool's gradient, the launch leaves
        # its sign bits (one byte per 16-byte slice) instead of the activation itself: h is never written, never read again
        pg = dict(graph_ptr=gptr, num_graphs=nb, weights=tg, mean=False, wmean=True, bits=True, store=False)
        L["g2"] = mk(9, T); hg = _GCNLayer.forward(L["g2"], h, g2w, g2b, struct, T, p1, seed + 2, T, k1, pool=pg)
        _, wg = ops.pooled_means(hg, pg)
        with torch.cuda.stream(side[0]):
            # Cheb branch: args (x, bias, struct, relu, drop_p, seed, defer_mask, x_gate_scale, *ws)
            pc = dict(graph_ptr=gptr, num_graphs=nb, weights=tc, mean=True, wmean=True, bits=True, store=False)
            L["c1"] = mk(11, Fa); hc = _ChebLayer.forward(L["c1"], x, c1b, struct, T, p2, seed + 3, T, None, c1w0, c1w1, c1w2,
                                                          pre=pre_c, pool=pc)
            mc, wc = ops.pooled_means(hc, pc)
        with torch.cuda.stream(side[1]):
            # SAGE branch: args (x, wl, bl, wr, struct, relu, drop_p, seed, defer_mask, x_gate_scale)
            ps = dict(graph_ptr=gptr, num_graphs=nb, weights=ts, mean=True, wmean=True, bits=True, store=False)
            L["s1"] = mk(10, Fa); hs = _SAGELayer.forward(L["s1"], x, s1l, s1b, s1r, struct, T, p2, seed + 4, T, None, pre=pre_s, pool=ps)
            ms, ws = ops.pooled_means(hs, ps)
        for st, ts_ in zip(side, ((mc, wc), (ms, ws))):
            main.wait_stream(st)
            for t in ts_:
                t.record_stream(main)
        if fuse:
            for t in blocks[1:]:          # made on the compute stream, consumed by the side streams
                t.record_stream(side[0] if t is not blocks[4] and t is not blocks[5] else side[1])
        # the three folded last convs: [wmean_g . W3 + b3 | mean_c . W0 + wmean_c . W1 + b | wmean_s . Wl + bl + mean_s . Wr]
        ctx.head = ([(wg, g3w, 0), (mc, c2w0, 1), (wc, c2w1, 1), (ws, s2l, 2), (ms, s2r, 2)], [g3b, c2b, s2b])
        out = ops.pooled_head(*ctx.head)
        # the gates of the three pooled activations: the activation itself, or its sign bits when the pooled launch left only those
        ctx.tail = (struct, k1, k2, hg, hc, hs)
        ctx.gate_bits = (pg.get("out_bits"), pc.get("out_bits"), ps.get("out_bits"))
        return out

    @staticmethod
    def backward(ctx, g):
        L = ctx.layers
        struct, k1, k2, hg, hc, hs = ctx.tail
        bg_, bc_, bs_ = ctx.gate_bits           # sign bits of the pooled activations (then hg / hc / hs are None: never written)
        gptr, n = struct.graph_ptr, struct.num_nodes
        main = torch.cuda.current_stream(g.device)
        side = ctx.side
        # the folded last convs first, on the compute stream: gradients of the five pooled matrices, five weight rows, three biases
        (ggw, gcm, gcw, gsw, gsm), gw5, gb3 = ops.pooled_head_bwd(ctx.head[0], ctx.head[1], g)
        g3wg, c2w0g, c2w1g, s2lg, s2rg = gw5[0:1], gw5[1:2], gw5[2:3], gw5[3:4], gw5[4:5]
        g3bg, c2bg, s2bg = gb3[0:1], gb3[1:2], gb3[2:3]
        c_ = ggw.shape[1]
        synth = (_POOLED_GRAD and n >= _POOLED_GRAD_MIN_NODES and ops.pooled_grad_supported(c_)
                 and struct.out_ell is not None)      # per branch: its gate bits exist
        for st in side:
            st.wait_stream(main)
        # GCN branch, last layer first: pooled = wmean(h) W^T + b

        def pooled_grad(gm, gw, kind, gate, scale, bits):
            if synth and bits is not None:
                return ops.PooledGrad(gm, gw, gptr, n, struct.colsum(kind), scale, bits)
            return ops.segment_pool_bwd(gm, gw, gptr, n, weights=struct.colsum(kind), gate=gate if bits is None else None, gate_scale=scale,
                                        gate_bits=bits)

        t = pooled_grad(None, ggw, "gcn", hg, k1, bg_)
        t, g2w, g2b = _GCNLayer.backward(L["g2"], t)[:3]
        # conv1's bias gradient is the column sum of the gradient conv2's backward just wrote: taken there, the first-layer
        # weight-gradient pass below reads six blocks instead of seven
        g1b_cs = getattr(L["g2"], "gx_colsum", None)
        fuse = ctx.fuse
        if fuse:
            bg = _GCNLayer.backward(L["g1"], t, blocks_only=True)            # [gh, g]
        else:
            _, g1w, g1b = _GCNLayer.backward(L["g1"], t)[:3]
        with torch.cuda.stream(side[0]):
            t = pooled_grad(gcm, gcw, "cheb", hc, k2, bc_)
            if fuse:
                bc = _ChebLayer.backward(L["c1"], t, blocks_only=True)       # [g, g_b1, g_c2]
            else:
                r = _ChebLayer.backward(L["c1"], t)
                c1b, c1w0, c1w1, c1w2 = r[1], r[8], r[9], r[10]
        with torch.cuda.stream(side[1]):
            t = pooled_grad(gsm, gsw, "sage", hs, k2, bs_)
            if fuse:
                bs = _SAGELayer.backward(L["s1"], t, blocks_only=True)       # [g_p, g]
            else:
                _, s1l, s1b, s1r = _SAGELayer.backward(L["s1"], t)[:4]
        for st in side:
            main.wait_stream(st)
        if fuse:
            # ONE pass over x for the weight gradients of all three first layers: x^T [gh | g || g | g_b1 | g_c2 || g_p | g];
            # the ones column of the pass yields the three bias gradients
            x0 = L["g1"].saved_tensors[0]
            o, i = L["g1"].saved_tensors[1].shape
            ow = (o + 3) // 4 * 4
            for blk in bc + bs:
                blk.record_stream(main)
            blks = ([bg[0]] if g1b_cs is not None else bg) + bc + bs
            nb, k = len(blks), len(blks) - 5             # k: index of the first Cheb block
            gwn = torch.empty((nb * ow, i), dtype=torch.float32, device=g.device)
            gbn = torch.empty(nb * ow, dtype=torch.float32, device=g.device)
            ops.linear_wgrad_parts(blks, x0, gwn, gbn)
            gwn = gwn.reshape(nb, ow, i)[:, :o]
            g1w, g1b = gwn[0], (g1b_cs[:o] if g1b_cs is not None else gbn[ow:ow + o])
            c1w0, c1w1, c1w2, c1b = gwn[k], gwn[k + 1], gwn[k + 2] - gwn[k], gbn[k * ow:k * ow + o]
            s1l, s1r, s1b = gwn[k + 3], gwn[k + 4], gbn[(k + 4) * ow:(k + 4) * ow + o]
        for t in (gcm, gcw, gsm, gsw):        # made on the compute stream, consumed by the side streams
            t.record_stream(side[0] if t is gcm or t is gcw else side[1])
        for t in (() if fuse else (c1b, c1w0, c1w1, c1w2, s1l, s1b, s1r)):
            t.record_stream(main)
        return (None, None, None, None, None, g1w, g1b, g2w, g2b, g3wg, g3bg, c1w0, c1w1, c1w2, c1b, c2w0g, c2w1g, c2bg,
                s1l, s1b, s1r, s2lg, s2bg, s2rg)


def family_a_graph(x, struct, p1, p2, seed, params):
    """One-node form of Family A's three conv branches + pools; ``params`` as listed in ``_FamilyAGraph``."""
    return _FamilyAGraph.apply(x, struct, p1, p2, seed, *params)
