"""BASELINE.json's other configurations as parity cases (small instances): cfg1 MLP on circuit-level features,
cfg2 4-qubit TFIM GNN, cfg3 random 20-qubit depth-40 circuits, cfg5 mixed corpus with Pauli-twirled members.
Each: GPU forward vs the fp64 oracle within 1e-5 and, for the GNNs, one backward pass vs oracle autograd."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(key, value):
    """Realised gaps go to gpurun_out/parity_configs.json (copied to profiles/ by the builder), as tests/test_gpu_cfg4_parity.py
    does for cfg4: a loosened bound is only honest next to the number it was loosened for."""
    path = os.path.join(ROOT, "gpurun_out", "parity_configs.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = {}
    if os.path.exists(path):
        with open(path) as fh:
            data = json.load(fh)
    data[key] = value
    with open(path, "w") as fh:
        json.dump(data, fh, indent=1, sort_keys=True)


def _host_batch(corpus, sel, exp_3d=False):
    xs, eis, bs, off = [], [], [], 0
    for b, g in enumerate(sel):
        x = torch.from_numpy(corpus["x"][g])
        xs.append(x)
        eis.append(torch.from_numpy(corpus["edge_index"][g]) + off)
        bs.append(torch.full((x.shape[0],), b, dtype=torch.long))
        off += x.shape[0]
    t = lambda k: torch.from_numpy(corpus[k][sel])
    noisy, y = t("noisy"), t("y")
    if exp_3d:
        noisy = noisy.unsqueeze(1)
    return dict(noisy=noisy, observable=t("observable"), depth=t("depth"), x=torch.cat(xs),
                edge_index=torch.cat(eis, 1), batch=torch.cat(bs), y=y)


def _check_family_a(corpus, nq, sel, tag=None):
    from blackwater.nn import ExpValCircuitGraphModelA
    from oracle.models import FamilyA

    torch.manual_seed(11)
    model = ExpValCircuitGraphModelA(nq, 22, 10)
    ref = FamilyA(nq, 22, 10).double().eval()
    ref.load_state_dict(model.state_dict())
    model = model.to(DEV).eval()
    hb = _host_batch(corpus, sel)
    keys = ("noisy", "observable", "depth", "x", "edge_index", "batch")
    out = model(*[hb[k].to(DEV) for k in keys])
    want = ref(*[hb[k].double() if hb[k].is_floating_point() else hb[k] for k in keys])
    err = (out.detach().cpu().double() - want.detach()).abs().max().item()
    # north_star's literal bar: 1e-5 against the reference's fp32 CPU arithmetic (the same oracle module in fp32), next to the
    # criterion the repo states (1e-5 against the exact, fp64, value) and the CPU path's own distance from exact
    ref32 = FamilyA(nq, 22, 10).eval()
    ref32.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    with torch.no_grad():
        want32 = ref32(*[hb[k] for k in keys])
    err32 = (out.detach().cpu() - want32).abs().max().item()
    fp32_gap = (want32.double() - want.detach()).abs().max().item()
    if tag:
        _record(tag, {"max_abs_gpu_minus_fp64_oracle": err, "max_abs_fp32_cpu_minus_fp64_oracle": fp32_gap,
                      "max_abs_gpu_minus_fp32_cpu": err32, "bound_asserted": 1e-5, "north_star": 1e-5,
                      "prediction_scale": float(want.detach().abs().max()), "nodes": int(hb["x"].shape[0]), "graphs": int(len(sel))})
    assert err < 1e-5, (err, fp32_gap)
    assert err32 < 1e-5, (err32, fp32_gap)         # holds on every small config (cfg4's 100-qubit graphs: tests/test_gpu_cfg4_parity.py)
    torch.nn.functional.mse_loss(out, hb["y"].to(DEV)).backward()
    torch.nn.functional.mse_loss(want, hb["y"].double()).backward()
    grads = {k: p.grad for k, p in ref.named_parameters()}
    overall = max(g.abs().max().item() for g in grads.values())
    for name, p in model.named_parameters():
        scale = max(grads[name].abs().max().item(), 1e-2 * overall)  # fp32 noise floor ~1e-6 of the largest gradient
        assert (p.grad.cpu().double() - grads[name]).abs().max().item() / scale < 2e-4, name


def _check_family_b(corpus, sel, out_size, tag=None):
    """With the reference's trained weights (gnn1.pth).  Random-init weights saturate the pooling fitness sigmoid to
    exactly 1.0 in fp32 for activations in (17, 37) where fp64 still separates them, so the discrete top-k choice --
    and with it every downstream gradient -- would legitimately differ between an fp32 and an fp64 implementation."""
    import os

    from blackwater.nn import family_b_from_state_dict
    from oracle.models import family_b_from_state_dict as oracle_from_sd

    sd = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt", "gnn1.pth"),
                    weights_only=True)
    assert sd["body_seq.2.weight"].shape[0] == out_size
    model = family_b_from_state_dict(sd).to(DEV).eval()
    ref = oracle_from_sd(sd).double().eval()
    hb = _host_batch(corpus, sel, exp_3d=True)
    out = model(hb["noisy"].to(DEV), None, hb["depth"].to(DEV), hb["x"].to(DEV), hb["edge_index"].to(DEV),
                hb["batch"].to(DEV))
    want = ref(hb["noisy"].double(), None, hb["depth"].double(), hb["x"].double(), hb["edge_index"], hb["batch"])
    # the north_star bar is 1e-5 against the reference's fp32 CPU path; against the fp64 oracle allow for what fp32
    # arithmetic itself costs on these (larger, out-of-distribution) graphs: twice the oracle's own fp32-vs-fp64 gap
    with torch.no_grad():
        want32 = oracle_from_sd(sd).eval()(hb["noisy"], None, hb["depth"], hb["x"], hb["edge_index"], hb["batch"])
    fp32_gap = (want32.double() - want.detach()).abs().max().item()
    err = (out.detach().cpu().double() - want.detach()).abs().max().item()
    err32 = (out.detach().cpu() - want32).abs().max().item()
    if tag:
        _record(tag, {"max_abs_gpu_minus_fp64_oracle": err, "max_abs_fp32_cpu_minus_fp64_oracle": fp32_gap,
                      "max_abs_gpu_minus_fp32_cpu": err32, "bound_asserted": max(1e-5, 2 * fp32_gap), "north_star": 1e-5,
                      "prediction_scale": float(want.detach().abs().max()), "nodes": int(hb["x"].shape[0]), "graphs": int(len(sel))})
    assert err < max(1e-5, 2 * fp32_gap), (err, fp32_gap)
    # the north_star's literal bar: 1e-5 against the fp32 CPU path (realised: ~3e-8 on cfg3)
    assert err32 < 1e-5, err32
    torch.nn.functional.mse_loss(out, hb["y"].to(DEV)).backward()
    torch.nn.functional.mse_loss(want, hb["y"].double()).backward()
    grads = {k: p.grad for k, p in ref.named_parameters()}
    overall = max(g.abs().max().item() for g in grads.values())
    for name, p in model.named_parameters():
        scale = max(grads[name].abs().max().item(), 1e-2 * overall)  # fp32 noise floor ~1e-6 of the largest gradient
        assert (p.grad.cpu().double() - grads[name]).abs().max().item() / scale < 2e-3, name  # fp32 sums over ~7k nodes


def test_cfg2_tfim_4q_gnn_batch32():
    from blackwater.data.synthetic import tfim_corpus

    corpus = tfim_corpus(4, list(range(0, 15)), 3, two_q="cx", exp_value_size=1)
    sel = np.arange(0, 45)[:32]
    _check_family_a(corpus, 4, sel, tag="cfg2_family_a")
    corpus4 = tfim_corpus(4, list(range(0, 15)), 3, two_q="cx", exp_value_size=4)
    _check_family_b(corpus4, sel, 4, tag="cfg2_family_b_gnn1")


def test_cfg3_random_20q_depth40():
    from blackwater.data.synthetic import encode_corpus, random_circuit

    circs = [random_circuit(20, 40, seed=s) for s in range(12)]
    corpus = encode_corpus(circs, 20, exp_value_size=1)
    assert 400 < corpus["x"][0].shape[0] < 900
    _check_family_a(corpus, 20, np.arange(12), tag="cfg3_family_a")
    corpus4 = encode_corpus(circs, 20, exp_value_size=4)
    _check_family_b(corpus4, np.arange(12), 4, tag="cfg3_family_b_gnn1")


def test_cfg5_mixed_corpus_with_pauli_twirl():
    from blackwater.data.synthetic import encode_corpus, pauli_twirl, random_circuit, tfim_circuit

    circs = [tfim_circuit(12, s, J=0.3, two_q="cx") for s in (1, 3, 5)]
    circs += [random_circuit(12, 20, seed=s) for s in range(3)]
    circs += [pauli_twirl(tfim_circuit(12, s, J=0.7, two_q="cx"), seed=s) for s in (2, 4)]
    corpus = encode_corpus(circs, 12, exp_value_size=1)
    _check_family_a(corpus, 12, np.arange(len(circs)), tag="cfg5_family_a")


def test_cfg1_mlp_on_v2_features():
    """demo2 feature set: encode_data_v2_ecr(two_q_gate='cx') -> 169-d rows -> MLP1 / MLP3."""
    import blackwater.nn as bnn
    import oracle.models as om
    from blackwater.data.synthetic import tfim_circuit
    from blackwater.library.learning.features import encode_data_v2_ecr

    circs = [tfim_circuit(4, s, J=0.1 * s, two_q="cx") for s in range(0, 10) for _ in range(3)]
    rng = np.random.default_rng(0)
    noisy = rng.uniform(-1, 1, size=(len(circs), 4)).tolist()
    X, y = encode_data_v2_ecr(circs, rng.uniform(-1, 1, size=(len(circs), 4)).tolist(), noisy, 4, two_q_gate="cx")
    assert X.shape == (30, 169)
    for cls, hidden in (("MLP1", 64), ("MLP3", 128)):
        torch.manual_seed(3)
        gpu = getattr(bnn, cls)(169, hidden, 4)
        ref = getattr(om, cls)(169, hidden, 4).double().eval()
        ref.load_state_dict(gpu.state_dict())
        out = gpu.to(DEV).eval()(X.to(DEV))
        want = ref(X.double())
        assert (out.detach().cpu().double() - want.detach()).abs().max().item() < 1e-5
        out.square().mean().backward()
        want.square().mean().backward()
        for (name, p), (_, q) in zip(gpu.named_parameters(), ref.named_parameters()):
            scale = max(q.grad.abs().max().item(), 1e-9)
            assert (p.grad.cpu().double() - q.grad).abs().max().item() / scale < 2e-4, (cls, name)
    # TRAINING mode: MLP3's step on the fp32 layer kernels (csrc/mlp_layers.hip: batch statistics, one autograd
    # node) against the oracle in fp64, dropout off so that both evaluate the same function: 1e-5 of each result's scale
    torch.manual_seed(4)
    gpu = bnn.MLP3(169, 128, 4, dropout_rate=0.0)
    ref = om.MLP3(169, 128, 4, dropout_rate=0.0).double().train()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in gpu.state_dict().items()})
    gpu = gpu.to(DEV).train()
    out, want = gpu(X.to(DEV)), ref(X.double())
    assert (out.detach().cpu().double() - want.detach()).abs().max().item() <= 1e-5 * max(want.abs().max().item(), 1.0)
    out.square().mean().backward()
    want.square().mean().backward()
    top = max(q.grad.abs().max().item() for q in ref.parameters())
    for (name, p), (_, q) in zip(gpu.named_parameters(), ref.named_parameters()):
        # (a Linear's bias in front of a BatchNorm has an analytically zero gradient: rounding noise of a few fp32 ulps of the LARGEST gradient -- 1.2e-7 of it measured -- so the floor is 5e-7 of that)
        assert (p.grad.cpu().double() - q.grad).abs().max().item() <= 1e-5 * max(q.grad.abs().max().item(), 5e-2 * top), name
    assert (gpu.bn1.running_var.cpu().double() - ref.bn1.running_var).abs().max().item() <= 1e-5 * ref.bn1.running_var.abs().max().item()


def test_cfg5_bf16_mfma_mlp_head():
    """The "bf16 MFMA MLP head" of BASELINE.json's mixed-corpus configuration: MLP3 on 169-d rows with
    ``model.mfma = "bf16"``.  Stated tolerances: (1) the kernel equals "round both operands to bf16, multiply exactly,
    accumulate in fp32" to 1e-5 of the output scale on every layer shape of MLP1/MLP3; (2) the whole MLP3 differs from
    its fp32 path by < 3e-2 of the output scale; (3) the backward runs on the same matrix cores: data gradient and weight
    gradient equal the bf16-rounded-operand products to 1e-5 of their scale, and the MLP3's parameter gradients stay within
    10 % (Frobenius norm) of the fp32 path's."""
    import blackwater.nn as bnn
    from blackwater.native import ops

    g = torch.Generator().manual_seed(0)
    for n, i, o, relu in ((1, 169, 64, True), (1000, 169, 128, False), (777, 128, 42, True), (4099, 58, 4, False),
                          (300, 256, 20, False), (50, 42, 4, False)):
        x = torch.randn(n, i, generator=g)
        w = torch.randn(o, i, generator=g) / i ** 0.5
        b = torch.randn(o, generator=g)
        got = ops.linear_bf16(x.to(DEV), w.to(DEV), b.to(DEV), relu=relu).cpu().double()
        want = x.bfloat16().double() @ w.bfloat16().double().t() + b.double()
        want = want.relu() if relu else want
        assert (got - want).abs().max().item() < 1e-5 * max(want.abs().max().item(), 1.0), (n, i, o)
        full = x.double() @ w.double().t() + b.double()
        full = full.relu() if relu else full
        assert (got - full).abs().max().item() < 3e-2 * max(full.abs().max().item(), 1.0)
    with pytest.raises(Exception):
        ops.linear_bf16(torch.randn(4, 300, device=DEV), torch.randn(8, 300, device=DEV))      # I > 256: unsupported

    torch.manual_seed(3)
    model = bnn.MLP3(169, 128, 4).to(DEV).eval()
    x = torch.randn(512, 169, device=DEV)
    ref = model(x)
    model.mfma = "bf16"
    out = model(x)
    scale = ref.abs().max().item()
    err = (out - ref).abs().max().item()
    assert 0 < err < 3e-2 * scale                      # a different arithmetic, within the stated tolerance
    out.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    bf16_grads = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    model.mfma = "f32"
    model(x).square().mean().backward()
    for (name, p), gb in zip(model.named_parameters(), bf16_grads):
        # ReLU masks can flip for pre-activations within bf16 rounding of zero: compare in the Frobenius norm
        assert (gb - p.grad).norm().item() < 0.1 * (p.grad.norm().item() + 1e-12), name

    # the two backward kernels against "round both operands to bf16, multiply exactly"
    for n, i, o in ((1, 169, 64), (1000, 169, 128), (777, 128, 42), (4099, 58, 4), (33, 20, 256)):
        gy = torch.randn(n, o, generator=g)
        xx = torch.randn(n, i, generator=g)
        w = torch.randn(o, i, generator=g) / i ** 0.5
        gx = ops.linear_bf16(gy.to(DEV), w.to(DEV), transposed=True).cpu().double()
        want = gy.bfloat16().double() @ w.bfloat16().double()
        assert (gx - want).abs().max().item() < 1e-5 * max(want.abs().max().item(), 1.0), (n, i, o)
        gw = torch.empty(o, i, device=DEV)
        gbias = torch.empty(o, device=DEV)
        ops.linear_wgrad_bf16(gy.to(DEV), xx.to(DEV), gw, gbias)
        want_w = gy.bfloat16().double().t() @ xx.bfloat16().double()
        assert (gw.cpu().double() - want_w).abs().max().item() < 2e-5 * max(want_w.abs().max().item(), 1.0), (n, i, o)
        want_b = gy.bfloat16().double().sum(0)
        assert (gbias.cpu().double() - want_b).abs().max().item() < 2e-5 * max(want_b.abs().max().item(), 1.0)


def test_cfg5_family_b_with_bf16_mlp3_head_on_a_mixed_corpus(golden_dir):
    """cfg5 as ONE composition: the reference's GNN with an MLP3 head (ExpValCircuitGraphModel_3, docs/tutorials/gnn.py:178-224;
    weights: the reference's gnn3_ising.pth, 103 465 parameters) whose head runs ``mfma = "bf16"``, on a mixed mini-corpus --
    TFIM, random and Pauli-twirled circuits of 4, 12, 20 and 100 qubits in one batch.  Stated tolerances:
    * the graph part (what enters the head: pooled features | noisy values | depth) within 1e-5 of the fp64 oracle's, relative to
      its scale (depth is a raw ~1e2 count) -- or twice the fp32 CPU oracle's own gap where that is larger, recorded;
    * the whole fp32 model within the same bound of the fp64 oracle at the output;
    * the bf16 head equals "round both operands of every GEMM to bf16 (the eval-mode BatchNorm folded into the weights first),
      multiply exactly, add in fp32" to 1e-4 of the output scale GIVEN the device's own head input, and stays within 5e-2 of
      the fp32 head (a different arithmetic)."""
    from blackwater.data.synthetic import encode_corpus, pauli_twirl, random_circuit, tfim_circuit
    from blackwater.nn import family_b_from_state_dict
    from blackwater.nn.family_b import ExpValCircuitGraphModel_3
    from oracle.models import family_b_from_state_dict as oracle_from_sd

    parts = [
        encode_corpus([tfim_circuit(4, s, J=0.2 * s, two_q="cx") for s in (1, 4, 9)], 4, two_q="cx", exp_value_size=4),
        encode_corpus([random_circuit(20, 40, seed=s) for s in (1, 2)], 20, two_q="cx", exp_value_size=4),
        encode_corpus([pauli_twirl(tfim_circuit(12, s, J=0.7, two_q="cx"), seed=s) for s in (2, 4)], 12, two_q="cx", exp_value_size=4),
        encode_corpus([pauli_twirl(tfim_circuit(100, 1, J=0.4, two_q="ecr"), seed=9)], 100, two_q="ecr", exp_value_size=4),
    ]
    corpus = {"x": sum((p["x"] for p in parts), []), "edge_index": sum((p["edge_index"] for p in parts), [])}
    for k in ("y", "noisy", "depth"):
        corpus[k] = np.concatenate([p[k] for p in parts])
    corpus["observable"] = np.zeros((len(corpus["x"]), 1, 1), np.float32)     # Family B ignores it (gnn.py:100-122)
    sel = np.arange(len(corpus["x"]))
    assert corpus["x"][-1].shape[0] > 2000 and len(sel) == 8                   # the 100-qubit twirled member is in the batch
    sd = torch.load(os.path.join(golden_dir, "ckpt", "gnn3_ising.pth"), weights_only=True)
    model = family_b_from_state_dict(sd).to(DEV).eval()
    assert isinstance(model, ExpValCircuitGraphModel_3) and sum(p.numel() for p in model.parameters()) == 103465
    ref64, ref32 = oracle_from_sd(sd).double().eval(), oracle_from_sd(sd).eval()
    hb = _host_batch(corpus, sel, exp_3d=True)
    seen = {}
    model.body_seq.register_forward_pre_hook(lambda m, a: seen.__setitem__("gpu", a[0].detach().cpu().double()))
    ref64.body_seq.register_forward_pre_hook(lambda m, a: seen.__setitem__("f64", a[0].detach()))
    ref32.body_seq.register_forward_pre_hook(lambda m, a: seen.__setitem__("f32", a[0].detach().double()))
    args_dev = (hb["noisy"].to(DEV), None, hb["depth"].to(DEV), hb["x"].to(DEV), hb["edge_index"].to(DEV), hb["batch"].to(DEV))
    with torch.no_grad():
        out32 = model(*args_dev).cpu().double()
        want = ref64(hb["noisy"].double(), None, hb["depth"].double(), hb["x"].double(), hb["edge_index"], hb["batch"])
        want32 = ref32(hb["noisy"], None, hb["depth"], hb["x"], hb["edge_index"], hb["batch"]).double()
        head_in = seen["gpu"]
        model.body_seq.mfma = "bf16"
        out16 = model(*args_dev).cpu().double()
    scale_in = seen["f64"].abs().max().item()
    gap_in, cpu_gap_in = (head_in - seen["f64"]).abs().max().item(), (seen["f32"] - seen["f64"]).abs().max().item()
    scale = want.abs().max().item()
    gap_out, cpu_gap_out = (out32 - want).abs().max().item(), (want32 - want).abs().max().item()

    # the bf16 head, emulated exactly on the device's own head input
    bf = lambda t: t.to(torch.float32).to(torch.bfloat16).to(torch.float64)
    m = model.body_seq

    def folded(fc, bn):
        sc = bn.weight.detach().cpu().double() * torch.rsqrt(bn.running_var.cpu().double() + bn.eps)
        return (fc.weight.detach().cpu().double() * sc[:, None]).float(), \
               ((fc.bias.detach().cpu().double() - bn.running_mean.cpu().double()) * sc + bn.bias.detach().cpu().double()).float()

    def lin(x, w, b, relu):
        y = bf(x) @ bf(w).T + b.double()
        return y.relu().float().double() if relu else y.float().double()      # layer outputs are fp32 tensors on the device

    w1, b1 = folded(m.fc1, m.bn1)
    w2, b2 = folded(m.fc2, m.bn2)
    x1 = lin(head_in, w1, b1, True)
    x2 = lin(x1, w2, b2, True) + x1
    h3 = lin(x2.float().double(), m.fc3.weight.detach().cpu(), m.fc3.bias.detach().cpu(), True)
    emu = lin(h3, m.fc4.weight.detach().cpu(), m.fc4.bias.detach().cpu(), False)
    gap_bf16 = (out16 - emu).abs().max().item()
    _record("cfg5_family_b3_bf16_head", {
        "graphs": 8, "nodes": int(hb["x"].shape[0]), "head_input_scale": scale_in, "head_input_gap_gpu_vs_fp64": gap_in,
        "head_input_gap_fp32_cpu_vs_fp64": cpu_gap_in, "output_scale": scale, "output_gap_gpu_fp32_vs_fp64": gap_out,
        "output_gap_fp32_cpu_vs_fp64": cpu_gap_out, "bf16_head_vs_rounded_operand_emulation": gap_bf16,
        "bf16_head_vs_fp32_head": (out16 - out32).abs().max().item()})
    assert gap_in <= max(1e-5 * scale_in, 2 * cpu_gap_in), (gap_in, cpu_gap_in, scale_in)
    assert gap_out <= max(1e-5 * max(scale, 1.0), 2 * cpu_gap_out), (gap_out, cpu_gap_out)
    assert gap_bf16 <= 1e-4 * max(scale, 1.0), gap_bf16
    assert 0 < (out16 - out32).abs().max().item() <= 5e-2 * max(scale, 1.0)
