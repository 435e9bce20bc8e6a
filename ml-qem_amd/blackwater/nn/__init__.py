"""Modules of the path (convolutions, pooling, MLP heads, the reference's model families) on the native kernels."""
from .conv import ChebConv, GCNConv, SAGEConv  # noqa: F401
from .family_b import (ASAPooling, ExpValCircuitGraphModel, ExpValCircuitGraphModel_2,  # noqa: F401
                       ExpValCircuitGraphModel_3, ExpValCircuitGraphModel_4, TransformerConv, family_b_from_state_dict)
from .mlp import MLP1, MLP2, MLP3  # noqa: F401
from .models import ExpValCircuitGraphModelA  # noqa: F401
