"""Modules of the path (convolutions, pooling, MLP heads, the reference's model families) on the native kernels."""
from .conv import ChebConv, GCNConv, SAGEConv  # noqa: F401
from .models import ExpValCircuitGraphModelA  # noqa: F401
