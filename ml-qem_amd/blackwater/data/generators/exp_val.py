"""Wire/disk record of one training example (reference: blackwater/data/generators/exp_val.py:31-89).

``exp_value_generator`` of that file runs Aer simulations and is out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Any, Dict, List

import torch

from ..graph import Data

OP_EDGE_KEY = "DAGOpNode_wire_DAGOpNode"


@dataclass
class ExpValueEntry:
    """Encoded circuit graph + observable + ideal / noisy expectation values (+ depth)."""

    circuit_graph: Dict[str, Any]
    observable: List[List[float]]
    ideal_exp_value: float
    noisy_exp_values: List[float]
    circuit_depth: int = 0

    def __repr__(self):
        return f"<ExpValueEntry (ideal: {self.ideal_exp_value}, noisy: {self.noisy_exp_values})>"

    def to_dict(self) -> Dict[str, Any]:
        return {
            "circuit_graph": self.circuit_graph,
            "observable": self.observable,
            "ideal_exp_value": self.ideal_exp_value,
            "noisy_exp_values": self.noisy_exp_values,
            "circuit_depth": self.circuit_depth,
        }

    @classmethod
    def from_json(cls, dictionary: Dict[str, Any]) -> "ExpValueEntry":
        return cls(**dictionary)

    def to_pyg_data(self) -> Data:
        """Tensor view used by the models: x[N,F] f32, edge_index[2,E] i64 over op->op wires (a graph
        without such edges raises ``KeyError`` -- the dataset relies on it to drop the entry),
        edge_attr[E,3], y[1,1(,k)], observable[1,T,1+4n], circuit_depth[1,1], noisy_i[1,1(,k)]."""
        wires = self.circuit_graph["edges"][OP_EDGE_KEY]
        fields = {
            "x": torch.tensor(self.circuit_graph["nodes"]["DAGOpNode"], dtype=torch.float),
            "edge_index": torch.tensor(wires["edge_index"], dtype=torch.long),
            "edge_attr": torch.tensor(wires["edge_attr"], dtype=torch.float),
            "y": torch.tensor([[self.ideal_exp_value]], dtype=torch.float),
            "observable": torch.tensor([self.observable], dtype=torch.float),
            "circuit_depth": torch.tensor([[self.circuit_depth]], dtype=torch.float),
        }
        for k, value in enumerate(self.noisy_exp_values):
            fields[f"noisy_{k}"] = torch.tensor([[value]], dtype=torch.float)
        return Data(**fields)
