"""Family A's graph part as ONE autograd node over wide, slice-addressed activation buffers.

The three branches of the reference model (docs/tutorials/01_ngem.ipynb cell [9]: GCNx3 || Chebx2 || SAGEx2) read the
same node rows, so running them layer by layer re-reads x (and every gradient) once per ``Linear``.  Here each level
of the model is one buffer whose column slices belong to the branches:

    U  [N, 4*FW] = x | T1 = L^x | T2 = 2 L^T1 - x | mean_in(x)            (FW = F rounded up to 4 floats)
    Y1 [N, 3*HW] = dinv*(x Wg1^T) | c1 = act(Cheb1) | s1 = act(SAGE1)      one GEMM over U     (HW = hidden rounded up)
    Z  [N, 3*HW] = g1 = act(A^ Y1[0] + b) | L^ c1 | mean_in(s1)            three aggregations on slices of Y1
    Y2 [N, HW+4] = dinv*(g1 Wg2^T) | Cheb2 out, SAGE2 out                  one GEMM over [Z | c1 | s1]
    G2, h3, g3   = the rest of the GCN branch; pooled [B,3] = graph means of (g3, Cheb2, SAGE2)

so the first-layer weights of all three branches are ONE [3*HW, 4*FW] block matrix (zeros where branches do not
connect), their gradient is ONE pass over (dY1, U), and nothing flows back into x (it is data).  Aggregations write
straight into column slices through the kernels' leading-dimension arguments.  Every slice starts on a 4-float
boundary and owns the columns up to the next one: the aggregation kernel moves whole 16-byte groups, so a slice's pad
columns are scratch and must never hold another slice's data; pads stay zero, so the GEMMs may sweep over them.  Numerically this is the same sum of products in a different association; parity with the oracle is
checked by tests/test_gpu_family_a.py (forward <= 1e-5, every parameter gradient).
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from ..native import ops
from ..native.structure import GraphStructure


def _r4(n: int) -> int:
    return (n + 3) // 4 * 4


def wide_rows(num_nodes: int, num_features: int, device) -> torch.Tensor:
    """The U buffer; ``wide[:, :F]`` is where the batch's node features belong (pads of that slice zeroed)."""
    fw = _r4(num_features)
    wide = torch.empty((max(num_nodes, 1), 4 * fw), dtype=torch.float32, device=device)[:num_nodes]
    if fw > num_features:
        wide[:, num_features:fw].zero_()
    return wide


def _as_wide(nodes: torch.Tensor) -> torch.Tensor:
    n, f = nodes.shape
    fw = _r4(f)
    if nodes.dim() == 2 and nodes.stride(1) == 1 and nodes.stride(0) == 4 * fw and n > 1:
        # a view handed out by the device-resident dataset: the row already has room for the three other slices
        return torch.as_strided(nodes, (n, 4 * fw), (4 * fw, 1), nodes.storage_offset())
    wide = wide_rows(n, f, nodes.device)
    wide[:, :f].copy_(nodes)
    return wide


class _FamilyAGraph(Function):
    @staticmethod
    def forward(ctx, nodes, struct: GraphStructure, train, seed, c1w, c1b, c2w, c2b, c3w, c3b, ch1w0, ch1w1, ch1w2, ch1b,
                ch2w0, ch2w1, ch2b, sl1w, sl1b, sr1w, sl2w, sl2b, sr2w):
        s = struct
        n, f = nodes.shape
        hc = c1w.shape[0]
        fw, hw = _r4(f), _r4(hc)
        y2w, oc = hw + 4, hw          # Y2: GCN slice in [0, hw), the two scalar branch outputs at columns hw, hw + 1
        dev = nodes.device
        dinv = s.gcn_dinv
        p_g, p_o = (0.1, 0.2) if train else (0.0, 0.0)
        U = _as_wide(nodes)
        x, t1, t2, mx = (U[:, k * fw:k * fw + f] for k in range(4))
        lap = dict(cscale=s.cheb_dinv, rscale=s.derived("cheb_neg"))
        ops.csr_aggregate(x, s.in_ptr, s.in_src, ell=s.in_ell, out=t1, **lap)
        ops.csr_aggregate(t1, s.in_ptr, s.in_src, ell=s.in_ell, alpha=2.0, z=x, beta=-1.0, out=t2, **lap)
        ops.csr_aggregate(x, s.in_ptr, s.in_src, ell=s.in_ell, rscale=s.sage_rinv, dself=s.derived("sage_dself"), out=mx)

        w1 = torch.zeros((3 * hw, 4 * fw), dtype=torch.float32, device=dev)
        b1 = torch.zeros(3 * hw, dtype=torch.float32, device=dev)
        w1[0:hc, 0:f] = c1w
        w1[hw:hw + hc, 0:f], w1[hw:hw + hc, fw:fw + f], w1[hw:hw + hc, 2 * fw:2 * fw + f] = ch1w0, ch1w1, ch1w2
        w1[2 * hw:2 * hw + hc, 3 * fw:3 * fw + f], w1[2 * hw:2 * hw + hc, 0:f] = sl1w, sr1w
        b1[hw:hw + hc], b1[2 * hw:2 * hw + hc] = ch1b, sl1b
        Y1 = ops.linear(U, w1, b1, rowscale=dinv, rs_cols=hw, relu=True, act_from=hw, drop_p=p_o, seed=seed + 1)

        Z = ops.padded_empty(n, 3 * hw, dev)
        ops.csr_aggregate(Y1[:, 0:hc], s.in_ptr, s.in_src, ell=s.in_ell, rscale=dinv, dself=dinv, bias=c1b, relu=True,
                          drop_p=p_g, seed=seed + 2, out=Z[:, 0:hc])
        ops.csr_aggregate(Y1[:, hw:hw + hc], s.in_ptr, s.in_src, ell=s.in_ell, out=Z[:, hw:hw + hc], **lap)
        ops.csr_aggregate(Y1[:, 2 * hw:2 * hw + hc], s.in_ptr, s.in_src, ell=s.in_ell, rscale=s.sage_rinv,
                          dself=s.derived("sage_dself"), out=Z[:, 2 * hw:2 * hw + hc])

        w2a = torch.zeros((oc + 2, 3 * hw), dtype=torch.float32, device=dev)
        w2b = torch.zeros((oc + 2, 2 * hw), dtype=torch.float32, device=dev)
        b2 = torch.zeros(oc + 2, dtype=torch.float32, device=dev)
        w2a[0:hc, 0:hc], w2a[oc, hw:hw + hc], w2a[oc + 1, 2 * hw:2 * hw + hc] = c2w, ch2w1[0], sl2w[0]
        w2b[oc, 0:hc], w2b[oc + 1, hw:hw + hc] = ch2w0[0], sr2w[0]
        b2[oc], b2[oc + 1] = ch2b[0], sl2b[0]
        Y2 = torch.empty((max(n, 1), y2w), dtype=torch.float32, device=dev)[:n, :oc + 2]
        cs = Y1[:, hw:3 * hw]
        ops.linear(Z, w2a, b2, out=Y2)
        ops.linear(cs, w2b, out=Y2, accumulate=True, rowscale=dinv, rs_cols=hc)

        G2 = ops.csr_aggregate(Y2[:, 0:hc], s.in_ptr, s.in_src, ell=s.in_ell, rscale=dinv, dself=dinv, bias=c2b,
                               relu=True, drop_p=p_g, seed=seed + 3)
        h3 = ops.linear(G2, c3w.contiguous(), rowscale=dinv)
        g3 = ops.csr_aggregate(h3, s.in_ptr, s.in_src, ell=s.in_ell, rscale=dinv, dself=dinv, bias=c3b)
        pooled = torch.cat([ops.segment_mean(g3, s.graph_ptr, s.num_graphs),
                            ops.segment_mean(Y2[:, oc:oc + 2], s.graph_ptr, s.num_graphs)], dim=1)
        ctx.struct, ctx.dims, ctx.p = s, (n, f, hc, fw, hw), (p_g, p_o)
        ctx.save_for_backward(U, Y1, Z, Y2, G2, w2a, w2b, c3w)
        return pooled

    @staticmethod
    def backward(ctx, gp):
        U, Y1, Z, Y2, G2, w2a, w2b, c3w = ctx.saved_tensors
        s = ctx.struct
        n, f, hc, fw, hw = ctx.dims
        oc = hw
        p_g, p_o = ctx.p
        dev = U.device
        dinv, dself2 = s.gcn_dinv, s.derived("gcn_dself")
        gcn_t = dict(cscale=dinv, rscale=dinv, dself=dself2)                       # A^ is symmetric in form
        lap_t = dict(cscale=s.derived("cheb_neg"), rscale=s.cheb_dinv)             # transpose swaps the two factors
        sage_t = dict(cscale=s.sage_rinv, dself=s.derived("sage_dself"))
        agg_t = lambda g, **kw: ops.csr_aggregate(g, s.out_ptr, s.out_dst, ell=s.out_ell, **kw)
        gp = gp.contiguous()

        gY2 = torch.zeros((max(n, 1), Y2.stride(0)), dtype=torch.float32, device=dev)[:n, :oc + 2]
        gg3 = ops.segment_mean_bwd(gp[:, 0:1], s.graph_ptr, n)
        ops.segment_mean_bwd(gp[:, 1:3], s.graph_ptr, n, out=gY2[:, oc:oc + 2])
        g_c3b = gg3.sum(0)
        gh3 = agg_t(gg3, **gcn_t)
        gG2 = ops.linear(gh3, c3w.contiguous(), transposed=True)
        g_c3w = torch.empty_like(c3w)
        ops.linear_wgrad(gh3, G2, g_c3w, None)
        gz2 = ops.relu_dropout_bwd(gG2, G2, 1.0 / (1.0 - p_g))
        g_c2b = gz2.sum(0)
        agg_t(gz2, out=gY2[:, 0:hc], **gcn_t)

        gZ = ops.linear(gY2, w2a, transposed=True)                                 # [N, 3*HW]
        gY1 = ops.padded_empty(n, 3 * hw, dev)
        ops.linear(gY2, w2b, transposed=True, out=gY1[:, hw:3 * hw])
        gw2a, gb2 = torch.empty_like(w2a), torch.empty(oc + 2, dtype=torch.float32, device=dev)
        ops.linear_wgrad(gY2, Z, gw2a, gb2)
        gw2b = torch.empty_like(w2b)
        ops.linear_wgrad(gY2, Y1[:, hw:3 * hw], gw2b, None)

        gz1 = ops.relu_dropout_bwd(gZ[:, 0:hc], Z[:, 0:hc], 1.0 / (1.0 - p_g))
        g_c1b = gz1.sum(0)
        agg_t(gz1, out=gY1[:, 0:hc], **gcn_t)
        c_sl, s_sl = slice(hw, hw + hc), slice(2 * hw, 2 * hw + hc)
        agg_t(gZ[:, c_sl], z=gY1[:, c_sl], beta=1.0, out=gY1[:, c_sl], **lap_t)    # direct + through L^ c1
        agg_t(gZ[:, s_sl], z=gY1[:, s_sl], beta=1.0, out=gY1[:, s_sl], **sage_t)   # direct + through mean(s1)
        ops.relu_dropout_bwd(gY1[:, hw:3 * hw], Y1[:, hw:3 * hw], 1.0 / (1.0 - p_o), out=gY1[:, hw:3 * hw])

        gw1 = torch.empty((3 * hw, 4 * fw), dtype=torch.float32, device=dev)
        gb1 = torch.empty(3 * hw, dtype=torch.float32, device=dev)
        ops.linear_wgrad(gY1, U, gw1, gb1)

        ch, sg = slice(hw, hw + hc), slice(2 * hw, 2 * hw + hc)
        return (None, None, None, None,
                gw1[0:hc, 0:f], g_c1b, gw2a[0:hc, 0:hc], g_c2b, g_c3w, g_c3b,
                gw1[ch, 0:f], gw1[ch, fw:fw + f], gw1[ch, 2 * fw:2 * fw + f], gb1[ch],
                gw2b[oc:oc + 1, 0:hc], gw2a[oc:oc + 1, hw:hw + hc], gb2[oc:oc + 1],
                gw1[sg, 3 * fw:3 * fw + f], gb1[sg], gw1[sg, 0:f],
                gw2a[oc + 1:oc + 2, 2 * hw:2 * hw + hc], gb2[oc + 1:oc + 2], gw2b[oc + 1:oc + 2, hw:hw + hc])


def family_a_graph_part(model, nodes, struct: GraphStructure, train: bool, seed: int) -> torch.Tensor:
    """[B,3] = graph means of the (GCN, Cheb, SAGE) branch outputs of ``model`` (an ExpValCircuitGraphModelA)."""
    m = model
    return _FamilyAGraph.apply(
        nodes, struct, train, seed,
        m.conv1.lin.weight, m.conv1.bias, m.conv2.lin.weight, m.conv2.bias, m.conv3.lin.weight, m.conv3.bias,
        m.cheb_conv1.lins[0].weight, m.cheb_conv1.lins[1].weight, m.cheb_conv1.lins[2].weight, m.cheb_conv1.bias,
        m.cheb_conv2.lins[0].weight, m.cheb_conv2.lins[1].weight, m.cheb_conv2.bias,
        m.sage_conv1.lin_l.weight, m.sage_conv1.lin_l.bias, m.sage_conv1.lin_r.weight,
        m.sage_conv2.lin_l.weight, m.sage_conv2.lin_l.bias, m.sage_conv2.lin_r.weight)
