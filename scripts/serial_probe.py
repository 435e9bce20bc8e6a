import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.backends import PauliObservable
from blackwater.data.circuit import circuit_to_qasm
from blackwater.data.synthetic import synthetic_backend, tfim_circuit
from blackwater.library.ngem.estimator import ngem
from blackwater.nn import ExpValCircuitGraphModelA
dev = "cuda:0"
nq = 4
backend = synthetic_backend(nq, "cx")
rng = np.random.RandomState(3)
texts = [circuit_to_qasm(tfim_circuit(nq, k % 15, float(rng.uniform(0, 2.0)), two_q="cx")) for k in range(60)]
obs1 = PauliObservable("IIIZ")
class _Job:
    def __init__(s, v): s.v = v
    def result(s):
        class R: pass
        r = R(); r.values = np.asarray(s.v); r.metadata = [{}] * len(s.v); return r
    def job_id(s): return "j"
    def status(s): return "DONE"
class Est:
    def run(self, c, o, p=None): return self._run(c, o, p or [()] * len(c))
    def _run(self, circuits, observables, parameter_values, **k): return _Job([0.1] * len(circuits))
model = ExpValCircuitGraphModelA(nq, 22, 10).to(dev).eval()
est = ngem(Est, model, backend)()
for n in (32, 256):
    qs = [texts[k % 60] for k in range(n)]
    est.run(qs, [obs1] * n).result()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); est.run(qs, [obs1] * n).result(); ts.append(time.perf_counter() - t0)
    print(n, "circuits: %.1f circuits/s  (%.2f ms per run)" % (n / sorted(ts)[2], sorted(ts)[2] * 1e3), flush=True)
import cProfile, pstats
qs = [texts[k % 60] for k in range(32)]
pr = cProfile.Profile(); pr.enable()
for _ in range(20): est.run(qs, [obs1] * 32).result()
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
