"""Dataset of encoded circuit graphs with ideal / noisy expectation values
(reference: blackwater/data/loaders/exp_val.py:13-82)."""
from __future__ import annotations

import json
import pickle
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

from ..generators.exp_val import ExpValueEntry
from ..graph import AddSelfLoops, Data


class CircuitGraphExpValMitigationDataset:
    """Reads ``.json`` / ``.pk`` files holding a list of ``ExpValueEntry`` dicts.

    Per entry: ``circuit`` and ``metadata`` are discarded, the rest goes through
    ``ExpValueEntry.from_json(...).to_pyg_data()`` and the transforms (default: ``AddSelfLoops``).  An entry
    that raises ``KeyError`` on the way (e.g. a graph with no op->op edge) is silently skipped, as in the
    reference (:68-76), so ``len()`` can be smaller than the number of stored entries.
    """

    def __init__(
        self,
        path: Union[str, List[str]],
        transforms: Optional[List[Callable[[Data], Data]]] = None,
        num_samples: Optional[int] = None,
    ):
        self.paths = list(path) if isinstance(path, (list, tuple)) else [path]
        self.transforms = transforms or [AddSelfLoops()]
        self.entries: List[Data] = []
        for file_path in self.paths:
            for record in self._read(file_path, num_samples):
                graph = self._convert(record)
                if graph is not None:
                    self.entries.append(graph)

    @staticmethod
    def _read(file_path: str, num_samples: Optional[int]) -> List[Dict[str, Any]]:
        if file_path.endswith(".json"):
            with open(file_path, "r") as fh:
                records = json.load(fh)
        elif file_path.endswith(".pk"):
            with open(file_path, "rb") as fh:
                records = pickle.load(fh)
        else:
            raise ValueError(f"unsupported dataset file (want .json or .pk): {file_path}")
        return records if num_samples is None else records[:num_samples]

    def _convert(self, record: Dict[str, Any]) -> Optional[Data]:
        record = {k: v for k, v in record.items() if k not in ("circuit", "metadata")}
        try:
            graph = ExpValueEntry.from_json(record).to_pyg_data()
            for transform in self.transforms:
                graph = transform(graph)
            return graph
        except KeyError:
            return None

    def len(self) -> int:
        return len(self.entries)

    def get(self, idx: int) -> Data:
        return self.entries[idx]

    __len__ = len

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return [self.entries[i] for i in range(*idx.indices(len(self.entries)))]
        return self.entries[idx]

    def __iter__(self):
        return iter(self.entries)
