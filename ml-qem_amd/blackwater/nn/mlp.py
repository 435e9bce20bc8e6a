"""MLP regressors / heads (reference: docs/tutorials/mlp.py:18-108 == blackwater/library/learning/mlp.py:18-108).

Same constructor signatures and state-dict keys (``fc1.weight`` ... ``bn1.running_mean`` ...), so the reference's
21 MLP checkpoints load with ``strict=True``.  The GEMMs (+bias, +ReLU where nothing sits in between) run on the
f32-MFMA dense kernel; BatchNorm/dropout/residual act on [batch, hidden] tensors and stay elementwise torch ops.

``model.mfma = "bf16"`` (default ``"f32"``) sends the forward GEMMs to the bf16 matrix cores -- operands rounded to bf16
in registers, fp32 accumulation, fp32 tensors in memory: the "bf16 MFMA MLP head" of BASELINE.json's mixed-corpus
configuration.  Outputs then differ from the fp32 path at the 1e-2 level, so it is an opt-in; gradients stay fp32.
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

    def trunk(self, x):
        x1 = self._drop(torch.relu(self.bn1(F.linear(x, self.fc1.weight, self.fc1.bias, mfma=self.mfma))))
        x2 = self._drop(torch.relu(self.bn2(F.linear(x1, self.fc2.weight, self.fc2.bias, mfma=self.mfma))))
        return x1 + x2

    def forward(self, x):
        return F.linear(self.trunk(x), self.fc3.weight, self.fc3.bias, mfma=self.mfma)


class MLP3(MLP2):
    def __init__(self, input_size, hidden_size, output_size, dropout_rate=0.3):
        super().__init__(input_size, hidden_size, output_size, dropout_rate)
        self.fc3 = _fc(hidden_size, hidden_size // 3)
        self.fc4 = _fc(hidden_size // 3, output_size)

    def forward(self, x):
        h = F.linear(self.trunk(x), self.fc3.weight, self.fc3.bias, relu=True, mfma=self.mfma)
        return F.linear(self._drop(h), self.fc4.weight, self.fc4.bias, mfma=self.mfma)
