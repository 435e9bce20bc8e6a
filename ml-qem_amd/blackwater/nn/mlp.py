"""MLP regressors / heads (reference: docs/tutorials/mlp.py:18-108 == blackwater/library/learning/mlp.py:18-108).

Same constructor signatures and state-dict keys (``fc1.weight`` ... ``bn1.running_mean`` ...), so the reference's
21 MLP checkpoints load with ``strict=True``.  The GEMMs run on the MFMA dense kernels with the element-wise work in
their epilogues wherever the maths allows:

* eval mode: BatchNorm is an affine map of the GEMM output, so ``relu(bn(fc(x)))`` is ONE GEMM with bias + ReLU epilogue
  over weights scaled by ``gamma / sqrt(running_var + eps)`` (folded on the fly from the live parameters: four [H]-sized
  ops instead of three [batch, H] passes);
* train mode: BatchNorm needs the batch statistics of the GEMM output (``F.batch_norm_train``: the column reductions and
  element-wise passes of csrc/bn.hip -- torch's kernels take 27 ms per layer and direction at 262 k rows; per-rank statistics
  under data parallelism exactly like ``DistributedDataParallel`` without ``SyncBatchNorm``; running statistics are
  broadcast from rank 0 once by ``Trainer``); ReLU, dropout and the residual add of the trunk are one fused launch
  (``F.relu_dropout_add``), its mask keyed by torch's seed like every dropout of this build.

``model.mfma = "bf16"`` (default ``"f32"``) sends the GEMMs -- forward, data gradient and weight gradient -- to the bf16
matrix cores (v_mfma_f32_16x16x32_bf16): operands rounded to bf16, fp32 accumulation, fp32 master weights -- the "bf16 MFMA MLP
head" of BASELINE.json's mixed-corpus configuration.  In training mode the hidden activations are STORED as bf16 too (MLP1: the
one-launch head's stash, csrc/mlp_head.hip; MLP2 / MLP3: the bf16-storage pipeline of csrc/mlp_layers.hip with BatchNorm, ReLU,
dropout and the residual in one pass per block): these layers are bound by HBM, so half the bytes is what makes the mode
faster than fp32.  Results differ from the fp32 path at the 1e-2 level, so it is an opt-in.
"""
from __future__ import annotations

import torch
from torch import nn

from ..native import functional as F
from .conv import _kaiming_linear, _WeightOnly


def _fc(in_f, out_f):
    w, b = _kaiming_linear(out_f, in_f)
    return _WeightOnly(w, b)


class MLP1(nn.Module):
    def __init__(self, input_size, hidden_size, output_size):
        super().__init__()
        self.fc1, self.fc2 = _fc(input_size, hidden_size), _fc(hidden_size, output_size)
        self.mfma = "f32"

    def forward(self, x):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        if F.mlp1_fused_ok(x2, self.fc1.weight, self.fc2.weight):
            # one launch forward, one backward (csrc/mlp_head.hip): the hidden activation is never a GEMM operand in memory,
            # only a stash for the backward (bf16 when ``mfma == "bf16"``)
            return F.mlp1(x2, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.mfma).reshape(*lead, self.fc2.weight.shape[0])
        h = F.linear(x, self.fc1.weight, self.fc1.bias, relu=True, mfma=self.mfma)
        return F.linear(h, self.fc2.weight, self.fc2.bias, mfma=self.mfma)


class MLP2(nn.Module):
    def __init__(self, input_size, hidden_size, output_size, dropout_rate=0.5):
        super().__init__()
        self.fc1, self.bn1 = _fc(input_size, hidden_size), nn.BatchNorm1d(hidden_size)
        self.fc2, self.bn2 = _fc(hidden_size, hidden_size), nn.BatchNorm1d(hidden_size)
        self.fc3 = _fc(hidden_size, output_size)
        self.p = dropout_rate
        self.mfma = "f32"

    def _drop(self, t):
        return nn.functional.dropout(t, self.p, True) if (self.training and self.p > 0) else t

    @staticmethod
    def _folded(fc, bn):
        """(W', b') with relu(bn(fc(x))) == relu(x W'^T + b') in eval mode."""
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        return fc.weight * scale[:, None], (fc.bias - bn.running_mean) * scale + bn.bias

    def _layer(self, x, fc, bn, residual=None):
        if not self.training:
            w, b = self._folded(fc, bn)
            y = F.linear(x, w, b, relu=True, mfma=self.mfma)
            return y if residual is None else y + residual
        self._calls = getattr(self, "_calls", 0) + 1
        from .models import dropout_key

        # static_dropout_key: the step counter lives on the device (train.RowsTrainer replays the step from a hipGraph, where a
        # host counter would freeze the masks); the layer's position keeps the two masks of one step apart
        call = (1 if residual is None else 2) if getattr(self, "static_dropout_key", False) else self._calls
        u = F.batch_norm_train(F.linear(x, fc.weight, fc.bias, mfma=self.mfma), bn)
        return F.relu_dropout_add(u, residual, self.p, dropout_key(call, salt=0x4D4C50))

    def trunk(self, x):
        x1 = self._layer(x, self.fc1, self.bn1)
        return self._layer(x1, self.fc2, self.bn2, residual=x1)

    def _bf16_storage(self, x, fc4=None, p_tail=0.0):
        """Training mode: the whole block as ONE autograd node on the layer kernels (csrc/mlp_layers.hip) -- ``mfma == "bf16"``: layer
        outputs and everything kept for the backward as bfloat16, half the bytes of every pass; ``mfma == "f32"`` (the default, the
        reference's arithmetic): the same kernels with fp32 storage and unrounded operands.  None when the shapes are not theirs."""
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        f32 = self.mfma == "f32"
        if not (self.training and (self.mfma == "bf16" or f32) and F.mlp_trunk_bf16_ok(x2, self.fc1, self.fc2, self.fc3, fc4, self.bn1, self.bn2)):
            return None
        from .models import dropout_key

        self._calls = getattr(self, "_calls", 0) + 1
        static = getattr(self, "static_dropout_key", False)
        seeds = [dropout_key((k + 1) if static else 3 * self._calls + k, salt=0x4D4C50) for k in range(3)]
        out = F.mlp_trunk_bf16(x2, self.fc1, self.bn1, self.fc2, self.bn2, self.fc3, fc4, self.p, p_tail, seeds, f32=f32)
        return out.reshape(*lead, out.shape[-1])

    def forward(self, x):
        out = self._bf16_storage(x)
        if out is not None:
            return out
        return F.linear(self.trunk(x), self.fc3.weight, self.fc3.bias, mfma=self.mfma)


class MLP3(MLP2):
    def __init__(self, input_size, hidden_size, output_size, dropout_rate=0.3):
        super().__init__(input_size, hidden_size, output_size, dropout_rate)
        self.fc3 = _fc(hidden_size, hidden_size // 3)
        self.fc4 = _fc(hidden_size // 3, output_size)

    def forward(self, x):
        out = self._bf16_storage(x, self.fc4, self.p)
        if out is not None:
            return out
        t = self.trunk(x)
        if self.training and self.p > 0:
            # dropout(relu(fc3 t)) with the counter-based masks of this build (one launch, keyed like the trunk's: a captured
            # step replays with fresh masks, which torch's generator-driven nn.functional.dropout does not give bit for bit)
            from .models import dropout_key

            self._calls = getattr(self, "_calls", 0) + 1
            call = 3 if getattr(self, "static_dropout_key", False) else self._calls
            u = F.linear(t, self.fc3.weight, self.fc3.bias, mfma=self.mfma)
            h = F.relu_dropout_add(u, None, self.p, dropout_key(call, salt=0x4D4C50))
        else:
            h = F.linear(t, self.fc3.weight, self.fc3.bias, relu=True, mfma=self.mfma)
        return F.linear(h, self.fc4.weight, self.fc4.bias, mfma=self.mfma)
