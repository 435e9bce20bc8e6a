"""Times the batched estimator path on 1024 100-qubit TFIM circuits from OpenQASM text: the encoder alone
(host fill + upload vs op stream + device expansion) and a whole ngem(..., batched=True) run() with Family A.
    python scripts/encode_bench.py [count] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from blackwater.data.backends import PauliObservable
from blackwater.data.circuit import circuit_to_qasm
from blackwater.data.native_encoder import NativeEncoder
from blackwater.data.synthetic import synthetic_backend, tfim_circuit
from blackwater.data.utils import get_backend_properties_v1
import blackwater.library.ngem.estimator as mod
from blackwater.library.ngem.estimator import ngem
from blackwater.nn import ExpValCircuitGraphModelA
from test_estimators import FakeEstimator

count = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
nq = 100
backend = synthetic_backend(nq, "ecr")
props = get_backend_properties_v1(backend)
rng = np.random.RandomState(3)
steps = list(range(1, 11))
distinct = [circuit_to_qasm(tfim_circuit(nq, steps[k % 10], float(rng.uniform(0, 2.0)), two_q="ecr")) for k in range(20)]
texts = [distinct[k % 20] for k in range(count)]
print(f"{count} circuits, {sum(len(t) for t in texts) / 1e6:.0f} MB of text, host threads available: {os.cpu_count()}", flush=True)


def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, out


enc = NativeEncoder(props)
for name, fn in (("host fill + upload  ", lambda: enc.encode_batch_to_device(texts, dev)), ("op stream + expand  ", lambda: enc.encode_batch_expand(texts, dev))):
    fn()
    ts = [wall(fn)[0] for _ in range(reps)]
    print(f"encoder, {name}: {min(ts):7.1f} ms (runs: {', '.join(f'{t:.1f}' for t in ts)})", flush=True)
for thr in (8, 16, 32, 64):
    fn = lambda: enc.encode_batch_expand(texts, dev, threads=thr)
    fn()
    print(f"encoder, op stream + expand, {thr:3d} host threads: {min(wall(fn)[0] for _ in range(reps)):7.1f} ms", flush=True)
torch.manual_seed(0)
model = ExpValCircuitGraphModelA(nq, 22, 10).to(dev).eval()
obs = [PauliObservable("I" * (nq - 1) + "Z")] * count
own = [(t + " ")[:-1] for t in texts]          # every text its own buffer: nothing is scanned once for many
est = ngem(FakeEstimator, model, backend, batched=True)()      # (the batch is expanded on the device: the host fill is no longer a path of the decorator)
for what, qs in (("20 buffers named 51 times each", texts), ("every text its own buffer", own)):
    est.run(qs, obs).result()
    ts = [wall(lambda: est.run(qs, obs).result().values)[0] for _ in range(reps)]
    print(f"ngem batched run(), Family A, {what}: {min(ts):7.1f} ms = {count / min(ts) * 1e3:8.0f} circuits/s "
          f"(runs: {', '.join(f'{t:.1f}' for t in ts)})", flush=True)
# where the host time of a run() goes (expand path)
import cProfile, pstats
est = ngem(FakeEstimator, model, backend, batched=True)()
pr = cProfile.Profile(); pr.enable(); est.run(texts, obs).result(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
