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
        x = x.contiguous()
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
        g = g.contiguous()
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
        x = x.contiguous()
        y = ops.linear(x, w.contiguous(), b, relu=relu)
        ctx.relu = relu
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        g = g.contiguous()
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


class _GCNLayer(Function):
    """y = act(D^-1/2 (A+I) D^-1/2 (x W^T) + b) as ONE autograd node.

    Forward: the projection writes h' = dinv * (x W^T) (row scale fused in the GEMM epilogue), so the aggregation
    needs no per-edge scalar: y = act(dinv * (sum_e h'[src_e] + h'[i]) + b).  Backward: ReLU/dropout mask from y,
    then the same symmetric-normalised aggregation on the transposed CSR, then the two GEMM gradients."""

    @staticmethod
    def forward(ctx, x, w, bias, struct: GraphStructure, relu, drop_p, seed):
        x = x.contiguous()
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
        g = g.contiguous()
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
        return ops.segment_mean(x.contiguous(), struct.graph_ptr, struct.num_graphs)

    @staticmethod
    def backward(ctx, g):
        return ops.segment_mean_bwd(g.contiguous(), ctx.struct.graph_ptr, ctx.n), None


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
