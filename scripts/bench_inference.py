"""Latency of estimator post-processing (the reference's VQE inner loop calls it once per energy evaluation):
NgemJob.result() circuit by circuit vs batched, on 4-qubit golden circuits."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from blackwater.data.backends import PauliObservable, StaticBackend
from blackwater.library.ngem.estimator import ngem
from blackwater.nn import ExpValCircuitGraphModelA
from test_estimators import FakeEstimator, _Job

qasm = json.load(open(os.path.join(ROOT, "tests/golden/g1_circuits.json")))[:64]
backend = StaticBackend.from_json(os.path.join(ROOT, "tests/golden/fake_lima_backend_props.json"))
obs = [PauliObservable("IIIIZ")] * len(qasm)

class Est(FakeEstimator):
    def _run(self, circuits, observables, parameter_values, **opts):
        return _Job([0.1] * len(circuits))

torch.manual_seed(0)
gpu_model = ExpValCircuitGraphModelA(5, 22, 10).to("cuda:0").eval()
for name, model, kw in (("gpu serial", gpu_model, {}), ("gpu batched", gpu_model, {"batched": True})):
    est = ngem(Est, model, backend, **kw)()
    est.run(qasm[:4], obs[:4]).result()  # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vals = est.run(qasm, obs).result().values
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:18s} {dt / len(qasm) * 1e3:7.2f} ms per circuit ({len(qasm)} circuits, {dt * 1e3:.0f} ms total)")
