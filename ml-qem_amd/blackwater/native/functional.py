"""torch.autograd wrappers: each forward/backward is one or two native kernel launches."""
from __future__ import annotations

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
    def forward(ctx, x, w, b, relu):
        x = ops.rowmajor(x)
        y = ops.linear(x, w.contiguous(), b, relu=relu)
        ctx.relu = relu
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
        if ctx.needs_input_grad[0]:
            gx = ops.linear(g, w.contiguous(), transposed=True)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty_like(w, memory_format=torch.contiguous_format)
            gb = torch.empty(w.shape[0], dtype=w.dtype, device=w.device) if ctx.has_bias else None
            ops.linear_wgrad(g, x, gw, gb)
        return gx, gw, gb, None


def linear(x, w, b=None, relu=False):
    lead = x.shape[:-1]
    y = _Linear.apply(x.reshape(-1, x.shape[-1]), w, b, relu)
    return y.reshape(*lead, w.shape[0])


class _MultiLinear(Function):
    """y = drop(act(sum_k x_k W_k^T + b)) -- ChebConv's sum over Chebyshev terms, SAGEConv's lin_l(mean) + lin_r(x) --
    as chained accumulating GEMM launches and ONE autograd node (no elementwise adds / relu / dropout passes)."""

    @staticmethod
    def forward(ctx, bias, relu, drop_p, seed, k, *xs_ws):
        xs = [ops.rowmajor(t) for t in xs_ws[:k]]
        ws = [t.contiguous() for t in xs_ws[k:]]
        y = None
        for i, (x, w) in enumerate(zip(xs, ws)):
            last = i == k - 1
            y = ops.linear(x, w, bias if i == 0 else None, out=y, accumulate=i > 0, relu=relu and last,
                           drop_p=drop_p if last else 0.0, seed=seed)
        ctx.k, ctx.relu, ctx.drop_p, ctx.has_bias = k, relu, drop_p, bias is not None
        ctx.save_for_backward(*xs, *ws, y if (relu or drop_p > 0) else None)
        return y

    @staticmethod
    def backward(ctx, g):
        k = ctx.k
        saved = ctx.saved_tensors
        xs, ws, y = saved[:k], saved[k:2 * k], saved[2 * k]
        g = ops.rowmajor(g)
        if y is not None:
            g = ops.relu_dropout_bwd(g, y, 1.0 / (1.0 - ctx.drop_p) if ctx.drop_p > 0 else 1.0)
        gb = None
        gxs, gws = [], []
        for i in range(k):
            gxs.append(ops.linear(g, ws[i], transposed=True) if ctx.needs_input_grad[5 + i] else None)
            gw = None
            if ctx.needs_input_grad[5 + k + i] or (i == 0 and ctx.has_bias and ctx.needs_input_grad[0]):
                gw = torch.empty_like(ws[i])
                want_b = i == 0 and ctx.has_bias
                if want_b:
                    gb = torch.empty(ws[i].shape[0], dtype=g.dtype, device=g.device)
                ops.linear_wgrad(g, xs[i], gw, gb if want_b else None)
            gws.append(gw)
        if ctx.has_bias and gb is None and ctx.needs_input_grad[0]:
            gb = g.sum(0)
        return (gb, None, None, None, None, *gxs, *gws)


def multi_linear(xs, ws, bias=None, relu=False, drop_p=0.0, seed=0):
    return _MultiLinear.apply(bias, relu, drop_p, seed, len(xs), *xs, *ws)


class _GCNLayer(Function):
    """y = act(D^-1/2 (A+I) D^-1/2 (x W^T) + b) as ONE autograd node.

    Forward: the projection writes h' = dinv * (x W^T) (row scale fused in the GEMM epilogue), so the aggregation
    needs no per-edge scalar: y = act(dinv * (sum_e h'[src_e] + h'[i]) + b).  Backward: ReLU/dropout mask from y,
    then the same symmetric-normalised aggregation on the transposed CSR, then the two GEMM gradients."""

    @staticmethod
    def forward(ctx, x, w, bias, struct: GraphStructure, relu, drop_p, seed):
        x = ops.rowmajor(x)
        dinv = struct.gcn_dinv
        h = ops.linear(x, w.contiguous(), rowscale=dinv)
        y = ops.csr_aggregate(h, struct.in_ptr, struct.in_src, ell=struct.in_ell, rscale=dinv, dself=dinv, bias=bias, relu=relu,
                              drop_p=drop_p, seed=seed)
        ctx.struct, ctx.relu, ctx.drop_p = struct, relu, drop_p
        ctx.save_for_backward(x, w, y if (relu or drop_p > 0) else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        s = ctx.struct
        g = ops.rowmajor(g)
        if y is not None:
            g = ops.relu_dropout_bwd(g, y, 1.0 / (1.0 - ctx.drop_p) if ctx.drop_p > 0 else 1.0)
        gb = g.sum(0) if ctx.needs_input_grad[2] else None
        gh = ops.csr_aggregate(g, s.out_ptr, s.out_dst, ell=s.out_ell, cscale=s.gcn_dinv, rscale=s.gcn_dinv,
                               dself=s.derived("gcn_dself"))
        gx = ops.linear(gh, w.contiguous(), transposed=True) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w, memory_format=torch.contiguous_format)
            ops.linear_wgrad(gh, x, gw, None)
        return gx, gw, gb, None, None, None, None


def gcn_layer(x, w, bias, struct, relu=False, drop_p=0.0, seed=0):
    return _GCNLayer.apply(x, w, bias, struct, relu, drop_p, seed)


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


class _ForwardOnly(Function):
    """Wraps a forward-only native op so that asking for its gradient fails loudly instead of silently
    cutting the graph (Family B's backward kernels are the next milestone)."""

    @staticmethod
    def forward(ctx, fn, name, *tensors):
        ctx.name = name
        return fn(*[t.detach() if torch.is_tensor(t) else t for t in tensors])

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError(f"{ctx.name}: backward kernel not implemented yet (forward/inference only)")


def forward_only(fn, name, *tensors):
    if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors):
        return _ForwardOnly.apply(fn, name, *tensors)
    return fn(*tensors)
