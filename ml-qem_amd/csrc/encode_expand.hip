// Device-side expansion of the compact op stream of a batch of circuits into what the models consume: node features x[N, F],
// the op -> op edge list (edge_index) and the batch vector -- the arrays circuit_to_graph_data_json builds per circuit
// (blackwater/data/utils.py:198-389) and PyG's Batch.from_data_list collates (docs/tutorials/__ml_models.py:105-119).
// The host only scans the OpenQASM text (encode_qasm.cpp: mlqem_qasm_batch_parse / _stream_fill: 16 bytes per op + 2 per qubit
// argument); rows, edges and offsets are made here, so a 1024-circuit run() of 100-qubit circuits uploads 0.2 GB instead of 1.3 GB.
//
//   rows    a block stages 256 rows in LDS and writes them as one contiguous run (rows are 88 bytes for F = 22: a lane per row
//           would scatter 4-byte stores); the same kernel files, per qubit argument t ("incidence"), its op and the sort key
//           (circuit, wire);
//   wires   a stable radix sort of the incidences by (circuit, wire) lists every wire's ops in program order: neighbours in the
//           sorted order are the endpoints of an edge (utils.py:334-347: one edge per qubit wire from the previous op on it);
//   order   the reference lists a source's out-edges latest-inserted first (its DAG hands successors back in that order; SURVEY
//           section 8 row a2): sources in program order, each one's successors by descending incidence number of the destination.
//           An op has one successor per qubit argument at most: <= 3 values sorted in registers, a barrier's (one per wire) ranked
//           by its wave.
// Same arrays as mlqem_qasm_batch_fill, bit for bit (tests/test_gpu_device_encoder.py).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace mlqem {

constexpr uint32_t kNoInc = 0xFFFFFFFFu;
constexpr int kExpandRows = 256;       // rows staged per block
constexpr int kMaxF = 64;              // feature width the staging buffer is sized for (reference: 21, 22, 28)

struct ExpandArgs {
  const mlqem_op_rec* ops; const uint16_t* wires;
  const int64_t* node_ptr;             // [B + 1] (device)
  int64_t N, W; int B; int max_wires;
  const float* t1; const float* t2; const float* readout; int nq_cal;
  const int32_t* g1; const int32_t* g2; const float* gate_error; const float* gate_length;
  int n_slots, use_q, use_g, F;
  float* x; int64_t ldx; int64_t* batch;
  uint32_t* inc_op; uint32_t* inc_key; uint32_t* next_inc;
};

__device__ __forceinline__ int circuit_at(const int64_t* __restrict__ node_ptr, int B, int64_t node) {
  int lo = 0, hi = B;                  // node_ptr[lo] <= node < node_ptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (node_ptr[mid] <= node) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(kExpandRows) void expand_rows_kernel(const ExpandArgs a) {
  extern __shared__ float s_rows[];    // [kExpandRows][F]
  const int64_t node = (int64_t)blockIdx.x * kExpandRows + threadIdx.x;
  const int F = a.F;
  if (node < a.N) {
    const mlqem_op_rec r = a.ops[node];
    float* row = s_rows + threadIdx.x * F;
    for (int i = 0; i < F; ++i) row[i] = 0.f;
    const int qc = r.meta & 3, p_cnt = (r.meta >> 4) & 3;
    const bool barrier = (r.meta & 4) != 0;
    if (p_cnt > 0) row[0] = r.p0;
    row[3 + r.slot] = 1.f;
    int col = 3 + a.n_slots;
    if (a.use_q) {
      if (!barrier)
        for (int s = 0; s < qc; ++s) {
          const int qi = r.q[s];
          if (qi < a.nq_cal) { row[col + s] = a.t1[qi]; row[col + 3 + s] = a.t2[qi]; row[col + 6 + s] = a.readout[qi]; }
        }
      col += 9;
    }
    if (a.use_g && !barrier) {
      int g = -1;
      if (qc == 1 && r.q[0] < a.nq_cal) g = a.g1[(int64_t)r.slot * a.nq_cal + r.q[0]];
      else if (qc == 2 && r.q[0] < a.nq_cal && r.q[1] < a.nq_cal) g = a.g2[((int64_t)r.slot * a.nq_cal + r.q[0]) * a.nq_cal + r.q[1]];
      if (g >= 0) { row[col] = a.gate_error[g]; row[col + 1] = a.gate_length[g]; }      // three qubits: a patch from the host
    }
    const int circ = circuit_at(a.node_ptr, a.B, node);
    if (a.batch) a.batch[node] = circ;
    // this op's qubit arguments: their op, their (circuit, wire) key, no successor yet
    const uint32_t wb = r.winc, we = node + 1 < a.N ? a.ops[node + 1].winc : (uint32_t)a.W;
    const uint32_t key0 = (uint32_t)circ * (uint32_t)a.max_wires;
    for (uint32_t t = wb; t < we; ++t) {
      a.inc_op[t] = (uint32_t)node;
      a.inc_key[t] = key0 + a.wires[t];
      a.next_inc[t] = kNoInc;
    }
  }
  __syncthreads();
  // the block's rows leave as one contiguous run
  const int64_t first = (int64_t)blockIdx.x * kExpandRows;
  const int rows = (int)min((int64_t)kExpandRows, a.N - first);
  if (a.ldx == F) {
    float* __restrict__ dst = a.x + first * F;
    for (int i = threadIdx.x; i < rows * F; i += kExpandRows) dst[i] = s_rows[i];
  } else {
    for (int i = threadIdx.x; i < rows * F; i += kExpandRows) {
      const int rr = i / F, cc = i - rr * F;
      a.x[(first + rr) * a.ldx + cc] = s_rows[i];
    }
  }
}

// neighbours in the sorted order with the same key: the earlier one's successor is the later one
__global__ __launch_bounds__(kBlock) void expand_link_kernel(const uint32_t* __restrict__ key_s, const uint32_t* __restrict__ val_s, int64_t W,
                                                             uint32_t* __restrict__ next_inc) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < 1 || i >= W) return;
  if (key_s[i] == key_s[i - 1]) next_inc[val_s[i - 1]] = val_s[i];
}

__global__ __launch_bounds__(kBlock) void expand_outdeg_kernel(const mlqem_op_rec* __restrict__ ops, const uint32_t* __restrict__ next_inc,
                                                               int64_t N, int64_t W, int32_t* __restrict__ outdeg) {
  const int64_t node = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (node > N) return;
  if (node == N) { outdeg[N] = 0; return; }
  const uint32_t wb = ops[node].winc, we = node + 1 < N ? ops[node + 1].winc : (uint32_t)W;
  int n = 0;
  for (uint32_t t = wb; t < we; ++t) n += next_inc[t] != kNoInc ? 1 : 0;
  outdeg[node] = n;
}

// edges of every source op, successors by descending incidence number
__global__ __launch_bounds__(kBlock) void expand_edges_kernel(const mlqem_op_rec* __restrict__ ops, const uint32_t* __restrict__ next_inc,
                                                              const uint32_t* __restrict__ inc_op, const int32_t* __restrict__ eoff, int64_t N,
                                                              int64_t W, int64_t E, int64_t* __restrict__ edge_src, int64_t* __restrict__ edge_dst) {
  const int64_t node = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool has = node < N;
  uint32_t wb = 0, we = 0;
  if (has) { wb = ops[node].winc; we = node + 1 < N ? ops[node + 1].winc : (uint32_t)W; }
  const bool heavy = has && we - wb > 3;
  if (has && !heavy) {
    uint32_t v[3] = {0u, 0u, 0u};        // 0 stands for "none" here: incidence 0 is nobody's successor (it is the first of its wire)
    int n = 0;
    for (uint32_t t = wb; t < we; ++t) { const uint32_t nx = next_inc[t]; if (nx != kNoInc) v[n++] = nx; }
    if (v[0] < v[1]) { const uint32_t s = v[0]; v[0] = v[1]; v[1] = s; }
    if (v[1] < v[2]) { const uint32_t s = v[1]; v[1] = v[2]; v[2] = s; }
    if (v[0] < v[1]) { const uint32_t s = v[0]; v[0] = v[1]; v[1] = s; }
    const int64_t e0 = eoff[node];
    for (int j = 0; j < n; ++j)
      if (e0 + j < E) { edge_src[e0 + j] = node; edge_dst[e0 + j] = inc_op[v[j]]; }
  }
  unsigned long long todo = __ballot(heavy);
  while (todo) {                          // a barrier: its wave ranks the successors (all distinct) by counting the larger ones
    const int owner = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const uint32_t b = __shfl(wb, owner), e = __shfl(we, owner);
    const int64_t src = __shfl(node, owner);
    const int64_t e0 = eoff[src];
    for (uint32_t t = b + lane; t < e; t += 64) {
      const uint32_t mine = next_inc[t];
      if (mine == kNoInc) continue;
      int rank = 0;
      for (uint32_t u = b; u < e; ++u) { const uint32_t o = next_inc[u]; rank += (o != kNoInc && o > mine) ? 1 : 0; }
      if (e0 + rank < E) { edge_src[e0 + rank] = src; edge_dst[e0 + rank] = inc_op[mine]; }
    }
  }
}

__global__ __launch_bounds__(kBlock) void expand_patch_kernel(const mlqem_x_patch* __restrict__ patches, int64_t P, int64_t N, int F,
                                                              float* __restrict__ x, int64_t ldx) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= P) return;
  const mlqem_x_patch p = patches[i];
  if (p.node != 0xFFFFFFFFu && (int64_t)p.node < N && (int)p.col < F) x[(int64_t)p.node * ldx + p.col] = p.value;
}

static int bits_of_u64(uint64_t v) {
  int b = 1;
  while (b < 32 && (1ull << b) <= v) ++b;
  return b;
}

struct ExpandLayout { size_t inc, deg, sort, scan, total; };
static ExpandLayout expand_layout(int64_t N, int64_t W) {
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  ExpandLayout l;
  l.inc = up((size_t)std::max<int64_t>(W, 1) * sizeof(uint32_t));
  l.deg = up((size_t)(N + 1) * sizeof(int32_t));
  size_t t = 0;
  (void)rocprim::radix_sort_pairs(nullptr, t, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (size_t)std::max<int64_t>(W, 1), 0, 32, (hipStream_t)0);
  l.sort = up(t);
  size_t s2 = 0;
  (void)rocprim::exclusive_scan(nullptr, s2, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0, (size_t)(N + 1), rocprim::plus<int32_t>(),
                                (hipStream_t)0);
  l.scan = up(s2);
  l.total = 6 * l.inc + 2 * l.deg + l.sort + l.scan;
  return l;
}

}  // namespace mlqem

using namespace mlqem;

extern "C" size_t mlqem_encode_expand_workspace_bytes(int64_t N, int64_t W) {
  if (N < 0 || W < 0) return 0;
  return expand_layout(N, W).total;
}

extern "C" int mlqem_encode_expand(const mlqem_op_rec* ops, const uint16_t* wires, const mlqem_x_patch* patches, int64_t num_patches,
                                   const int64_t* node_ptr, int64_t N, int64_t W, int64_t E, int64_t B, int max_wires, const float* t1,
                                   const float* t2, const float* readout, int num_cal_qubits, const int32_t* g1, const int32_t* g2,
                                   const float* gate_error, const float* gate_length, int num_slots, int use_qubit_features,
                                   int use_gate_features, float* x, int64_t ldx, int64_t* edge_src, int64_t* edge_dst, int64_t* batch,
                                   void* workspace, size_t workspace_bytes, mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || W < 0 || E < 0 || B < 0 || num_patches < 0 || max_wires < 0 || num_slots < 2 || num_cal_qubits < 0) return MLQEM_ERR_BAD_ARG;
  const int F = 3 + num_slots + (use_qubit_features ? 9 : 0) + (use_gate_features ? 2 : 0);
  if (F > kMaxF || num_slots > 255) return MLQEM_ERR_UNSUPPORTED;
  if (N >= (1ll << 31) || W >= (1ll << 32) - 1 || E >= (1ll << 31)) return MLQEM_ERR_UNSUPPORTED;
  if ((uint64_t)std::max<int64_t>(B, 1) * (uint64_t)std::max(max_wires, 1) >= (1ull << 32)) return MLQEM_ERR_UNSUPPORTED;
  if (N == 0) return MLQEM_OK;
  if (!ops || !node_ptr || !x || ldx < F || B == 0 || (W > 0 && !wires) || (E > 0 && (!edge_src || !edge_dst))) return MLQEM_ERR_BAD_ARG;
  if (num_patches > 0 && !patches) return MLQEM_ERR_BAD_ARG;
  if (use_qubit_features && num_cal_qubits > 0 && (!t1 || !t2 || !readout)) return MLQEM_ERR_BAD_ARG;
  if (use_gate_features && num_cal_qubits > 0 && (!g1 || !g2)) return MLQEM_ERR_BAD_ARG;
  const ExpandLayout l = expand_layout(N, W);
  if (!workspace || workspace_bytes < l.total) return MLQEM_ERR_WORKSPACE;
  char* w = static_cast<char*>(workspace);
  auto take = [&](size_t bytes) { char* at = w; w += bytes; return at; };
  uint32_t* inc_op = reinterpret_cast<uint32_t*>(take(l.inc));
  uint32_t* inc_key = reinterpret_cast<uint32_t*>(take(l.inc));
  uint32_t* next_inc = reinterpret_cast<uint32_t*>(take(l.inc));
  uint32_t* key_s = reinterpret_cast<uint32_t*>(take(l.inc));
  uint32_t* val_s = reinterpret_cast<uint32_t*>(take(l.inc));
  uint32_t* iota = reinterpret_cast<uint32_t*>(take(l.inc));
  int32_t* outdeg = reinterpret_cast<int32_t*>(take(l.deg));
  int32_t* eoff = reinterpret_cast<int32_t*>(take(l.deg));
  void* sort_tmp = take(l.sort);
  void* scan_tmp = take(l.scan);
  (void)iota;
  ExpandArgs a{ops, wires, node_ptr, N, W, (int)B, std::max(max_wires, 1), t1, t2, readout, num_cal_qubits, g1, g2, gate_error, gate_length,
               num_slots, use_qubit_features, use_gate_features, F, x, ldx, batch, inc_op, inc_key, next_inc};
  hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)ceil_div(N, (int64_t)kExpandRows)), dim3(kExpandRows),
                     (size_t)kExpandRows * F * sizeof(float), stream, a);
  if (num_patches > 0)
    hipLaunchKernelGGL(expand_patch_kernel, dim3((unsigned)ceil_div(num_patches, (int64_t)kBlock)), dim3(kBlock), 0, stream, patches, num_patches,
                       N, F, x, ldx);
  if (W > 0 && E > 0) {
    size_t sort_bytes = l.sort;
    const int key_bits = bits_of_u64((uint64_t)B * (uint64_t)std::max(max_wires, 1));
    // values = the incidence numbers 0 .. W - 1 (a counting iterator: no iota array); the sort is stable, so a wire's ops stay in program order
    if (rocprim::radix_sort_pairs(sort_tmp, sort_bytes, inc_key, key_s, rocprim::counting_iterator<uint32_t>(0u), val_s, (size_t)W, 0,
                                  (unsigned)key_bits, stream) != hipSuccess)
      return MLQEM_ERR_LAUNCH;
    hipLaunchKernelGGL(expand_link_kernel, dim3((unsigned)ceil_div(W, (int64_t)kBlock)), dim3(kBlock), 0, stream, key_s, val_s, W, next_inc);
    hipLaunchKernelGGL(expand_outdeg_kernel, dim3((unsigned)ceil_div(N + 1, (int64_t)kBlock)), dim3(kBlock), 0, stream, ops, next_inc, N, W, outdeg);
    size_t scan_bytes = l.scan;
    if (rocprim::exclusive_scan(scan_tmp, scan_bytes, outdeg, eoff, (int32_t)0, (size_t)(N + 1), rocprim::plus<int32_t>(), stream) != hipSuccess)
      return MLQEM_ERR_LAUNCH;
    hipLaunchKernelGGL(expand_edges_kernel, dim3((unsigned)ceil_div(N, (int64_t)kBlock)), dim3(kBlock), 0, stream, ops, next_inc, inc_op, eoff, N, W,
                       E, edge_src, edge_dst);
  }
  return launch_status();
}
