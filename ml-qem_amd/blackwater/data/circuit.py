"""Minimal circuit IR + OpenQASM-2 reader used by the encoders when qiskit is not installed.

The reference walks a qiskit ``QuantumCircuit`` (blackwater/data/utils.py:198-389 goes through
``circuit_to_dag``; docs/tutorials/mlp.py:124-133,180-186 use ``circuit.data`` / ``count_ops``).  All the
encoders need from a circuit is the ordered op list ``(name, qubits, clbits, params)``, so this module
provides exactly that, from three sources:

* an OpenQASM-2 string (the ``"circuit"`` field of the reference's ``.json`` datasets),
* a live qiskit ``QuantumCircuit`` (duck-typed: ``.data``, ``.qubits``, ``.clbits``; never imported here),
* an already built :class:`Circuit`.
"""
from __future__ import annotations

import ast
import math
import operator
import re
from dataclasses import dataclass, field
from typing import Any, Dict, List, Sequence, Tuple


@dataclass
class CircuitOp:
    """One instruction: gate name, flat qubit indices, flat clbit indices, float parameters."""

    name: str
    qubits: Tuple[int, ...]
    clbits: Tuple[int, ...] = ()
    params: Tuple[float, ...] = ()


@dataclass
class Circuit:
    """Ordered op list over ``num_qubits`` qubit wires and ``num_clbits`` classical wires."""

    num_qubits: int
    num_clbits: int = 0
    ops: List[CircuitOp] = field(default_factory=list)
    # index of each flat qubit inside its own register (what qiskit's deprecated ``Qubit.index`` returns);
    # identical to the flat index for the single-register circuits the reference datasets hold.
    qubit_reg_index: List[int] = field(default_factory=list)

    def __post_init__(self):
        if not self.qubit_reg_index:
            self.qubit_reg_index = list(range(self.num_qubits))

    # -- the three qiskit-circuit queries the reference's feature encoders make -----------------------
    def count_ops(self) -> Dict[str, int]:
        out: Dict[str, int] = {}
        for op in self.ops:
            out[op.name] = out.get(op.name, 0) + 1
        return out

    def depth(self) -> int:
        """Longest path over qubit+clbit wires; directives (barriers) do not count, like qiskit."""
        level = [0] * (self.num_qubits + self.num_clbits)
        for op in self.ops:
            if op.name == "barrier":
                continue
            wires = list(op.qubits) + [self.num_qubits + c for c in op.clbits]
            if not wires:
                continue
            new = max(level[w] for w in wires) + 1
            for w in wires:
                level[w] = new
        return max(level) if level else 0

    def bind_parameters(self, _params) -> "Circuit":
        """Circuits in this IR are always fully bound; kept so the estimator wrappers can call it."""
        return self

    assign_parameters = bind_parameters

    def __len__(self):
        return len(self.ops)

    # -- constructors ---------------------------------------------------------------------------------
    @staticmethod
    def from_qasm_str(text: str) -> "Circuit":
        return _QasmReader(text).read()

    @staticmethod
    def from_any(obj: Any) -> "Circuit":
        if isinstance(obj, Circuit):
            return obj
        if isinstance(obj, str):
            return Circuit.from_qasm_str(obj)
        if hasattr(obj, "data") and hasattr(obj, "qubits"):
            return _from_qiskit_like(obj)
        raise TypeError(f"cannot interpret {type(obj).__name__} as a circuit")


def _param_to_float(p: Any) -> float:
    """float(p) for numbers; bound ParameterExpressions via their symbolic value (utils.py:283-287)."""
    if isinstance(p, (int, float)):
        return float(p)
    if hasattr(p, "is_real") and p.is_real():
        return float(getattr(p, "_symbol_expr", p))
    return float(p)


def _from_qiskit_like(qc: Any) -> Circuit:
    qubits, clbits = list(qc.qubits), list(getattr(qc, "clbits", []))
    qpos = {id(q): i for i, q in enumerate(qubits)}
    cpos = {id(c): i for i, c in enumerate(clbits)}
    reg_index = []
    for i, q in enumerate(qubits):
        idx = getattr(q, "_index", None)
        reg_index.append(i if idx is None else int(idx))
    ops = []
    for inst in qc.data:
        operation = getattr(inst, "operation", None)
        if operation is None:  # legacy (instruction, qargs, cargs) tuples
            operation, qargs, cargs = inst
        else:
            qargs, cargs = inst.qubits, inst.clbits
        ops.append(
            CircuitOp(
                operation.name,
                tuple(qpos[id(q)] for q in qargs),
                tuple(cpos[id(c)] for c in cargs),
                tuple(_param_to_float(p) for p in operation.params),
            )
        )
    return Circuit(len(qubits), len(clbits), ops, reg_index)


# -------------------------------------------------------------------------------------------------
# OpenQASM 2 subset reader
_BINOPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
           ast.Pow: operator.pow}
_FUNCS = {"sin": math.sin, "cos": math.cos, "tan": math.tan, "exp": math.exp, "ln": math.log, "sqrt": math.sqrt,
          "asin": math.asin, "acos": math.acos, "atan": math.atan}


def eval_angle(expr: str, env: Dict[str, float] | None = None) -> float:
    """Evaluates a QASM parameter expression (numbers, ``pi``, + - * / ^, unary minus, math functions)."""
    tree = ast.parse(expr.replace("^", "**"), mode="eval").body

    def ev(node):
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            return float(node.value)
        if isinstance(node, ast.Name):
            if node.id == "pi":
                return math.pi
            if env and node.id in env:
                return env[node.id]
            raise ValueError(f"unknown identifier {node.id!r} in angle expression")
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            v = ev(node.operand)
            return -v if isinstance(node.op, ast.USub) else v
        if isinstance(node, ast.BinOp) and type(node.op) in _BINOPS:
            return _BINOPS[type(node.op)](ev(node.left), ev(node.right))
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in _FUNCS:
            return _FUNCS[node.func.id](*[ev(a) for a in node.args])
        raise ValueError(f"unsupported angle expression: {expr!r}")

    return ev(tree)


def _split_top_level(text: str, sep: str = ",") -> List[str]:
    parts, depth, cur = [], 0, []
    for ch in text:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    if cur:
        parts.append("".join(cur))
    return [p.strip() for p in parts if p.strip()]


class _QasmReader:
    _ARG = re.compile(r"^([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\s*(\d+)\s*\])?$")
    _STMT = re.compile(r"^([A-Za-z_][A-Za-z0-9_]*)\s*(?:\((.*)\))?\s*(.*)$", re.S)

    def __init__(self, text: str):
        text = re.sub(r"//[^\n]*", "", text)
        # gate definitions are kept opaque: an op that uses one carries the definition's NAME (what
        # QuantumCircuit.from_qasm_str yields as ``instruction.name``), it is not expanded.
        self.custom_gates = set(re.findall(r"\bgate\s+([A-Za-z_][A-Za-z0-9_]*)", text))
        text = re.sub(r"\b(gate|opaque)\b[^{;]*\{[^}]*\}", "", text)
        text = re.sub(r"\bopaque\b[^;]*;", "", text)
        self.statements = [s.strip() for s in text.split(";") if s.strip()]
        self.qregs: Dict[str, Tuple[int, int]] = {}
        self.cregs: Dict[str, Tuple[int, int]] = {}
        self.nq = self.nc = 0
        self.reg_index: List[int] = []

    def _bits(self, arg: str, regs: Dict[str, Tuple[int, int]]) -> List[int]:
        m = self._ARG.match(arg.strip())
        if not m or m.group(1) not in regs:
            raise ValueError(f"bad QASM argument {arg!r}")
        base, size = regs[m.group(1)]
        if m.group(2) is None:
            return [base + i for i in range(size)]
        idx = int(m.group(2))
        if idx >= size:
            raise ValueError(f"index out of range in {arg!r}")
        return [base + idx]

    def read(self) -> Circuit:
        ops: List[CircuitOp] = []
        for stmt in self.statements:
            if stmt.startswith("OPENQASM") or stmt.startswith("include"):
                continue
            m = re.match(r"^(qreg|creg)\s+([A-Za-z_][A-Za-z0-9_]*)\s*\[\s*(\d+)\s*\]$", stmt)
            if m:
                kind, name, size = m.group(1), m.group(2), int(m.group(3))
                if kind == "qreg":
                    self.qregs[name] = (self.nq, size)
                    self.nq += size
                    self.reg_index += list(range(size))
                else:
                    self.cregs[name] = (self.nc, size)
                    self.nc += size
                continue
            if stmt.startswith("measure"):
                src, dst = stmt[len("measure"):].split("->")
                qs, cs = self._bits(src, self.qregs), self._bits(dst, self.cregs)
                if len(qs) != len(cs):
                    raise ValueError(f"measure size mismatch: {stmt!r}")
                ops += [CircuitOp("measure", (q,), (c,)) for q, c in zip(qs, cs)]
                continue
            if stmt.startswith("barrier"):
                qs: List[int] = []
                for a in _split_top_level(stmt[len("barrier"):]):
                    qs += self._bits(a, self.qregs)
                ops.append(CircuitOp("barrier", tuple(qs)))
                continue
            if stmt.startswith("reset"):
                ops += [CircuitOp("reset", (q,)) for q in self._bits(stmt[len("reset"):], self.qregs)]
                continue
            m = self._STMT.match(stmt)
            if not m:
                raise ValueError(f"cannot parse QASM statement {stmt!r}")
            name, ptxt, atxt = m.group(1), m.group(2), m.group(3)
            params = tuple(eval_angle(p) for p in _split_top_level(ptxt)) if ptxt else ()
            arg_bits = [self._bits(a, self.qregs) for a in _split_top_level(atxt)]
            width = max(len(b) for b in arg_bits)
            for k in range(width):  # whole-register arguments broadcast
                ops.append(CircuitOp(name, tuple(b[k] if len(b) > 1 else b[0] for b in arg_bits), (), params))
        return Circuit(self.nq, self.nc, ops, self.reg_index)


def circuit_to_qasm(circ: Circuit, creg: str = "meas") -> str:
    """Writes the IR back as OpenQASM 2 (angles with full ``repr`` precision)."""
    lines = ["OPENQASM 2.0;", 'include "qelib1.inc";', f"qreg q[{circ.num_qubits}];"]
    if circ.num_clbits:
        lines.append(f"creg {creg}[{circ.num_clbits}];")
    for op in circ.ops:
        if op.name == "measure":
            lines.append(f"measure q[{op.qubits[0]}] -> {creg}[{op.clbits[0]}];")
            continue
        args = ",".join(f"q[{q}]" for q in op.qubits)
        head = f"{op.name}({','.join(repr(float(p)) for p in op.params)})" if op.params else op.name
        lines.append(f"{head} {args};")
    return "\n".join(lines) + "\n"
