// Row order of a coarsened graph and the per-device dynamic-LDS bookkeeping.
//
// mlqem_tile_order_by_position: the clusters of every graph in the PROGRAM ORDER of their centres.  ASAPooling's coarsened graphs of
// large circuits (docs/tutorials/gnn.py:85,92,104-112) have long rows around the circuit's barriers, and long rows that are
// neighbours in program order share nearly all of their sources: the dense blocks (dense_block.hpp) take 16 rows at a time in this
// order.  (Rounds 4-5 also carried LDS-staged "tile" kernels built on the same order -- tile_attn.hip, tile_pool.hip, a tile plan
// builder; measured slower than the per-edge kernels, superseded by the dense blocks and removed in round 6: git history.)
#include <mutex>

#include "common.hpp"

namespace mlqem {

bool ensure_dynamic_lds_impl(const void* kernel, size_t bytes) {
  constexpr int kMaxDev = 64, kMaxKernels = 64;
  struct Entry { const void* f; size_t have[kMaxDev]; };
  static Entry table[kMaxKernels] = {};
  static int used = 0;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return false;
  std::lock_guard<std::mutex> lock(mu);
  Entry* e = nullptr;
  for (int i = 0; i < used; ++i)
    if (table[i].f == kernel) { e = &table[i]; break; }
  if (!e) {
    if (used == kMaxKernels) return false;
    e = &table[used++];
    e->f = kernel;
  }
  if (e->have[dev] >= bytes || bytes <= 48 * 1024) return true;      // (the default limit serves small requests)
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  e->have[dev] = bytes;
  return true;
}

// exclusive prefix of one int per thread over a workgroup of WAVES waves; `tot` receives the total.  `tmp`: WAVES ints of LDS.
template <int WAVES>
__device__ __forceinline__ int block_exclusive_scan(int v, int* tmp, int& tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();                       // tmp may still be read from an earlier call
  if (lane == 63) tmp[wave] = inc;
  __syncthreads();
  int base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    const int t = tmp[w];
    if (w < wave) base += t;
    all += t;
  }
  tot = all;
  return base + inc - v;
}

// order[new_ptr[g] + r] = the cluster whose centre is the r-th kept node of graph g in NODE order (= program order: the encoder
// numbers a circuit's operations in instruction order, blackwater/data/utils.py:198-389).  slot[v] = cluster id of a kept centre v,
// -1 elsewhere (mlqem_asap_slot_map).  One workgroup per graph -- of 1024 threads, twelve nodes each: a 100-qubit circuit's 11 k nodes
// are ONE trip (load, scan, store); at 256 threads and eight nodes a trip the six trips, each waiting for its loads and two barriers,
// were 22 us whatever the batch (64 workgroups on 256 compute units).
constexpr int kOrderThreads = 1024;
__global__ __launch_bounds__(kOrderThreads) void tile_order_kernel(const int32_t* __restrict__ slot, const int32_t* __restrict__ gptr,
                                                                   const int32_t* __restrict__ new_gptr, int32_t* __restrict__ order) {
  __shared__ int tmp[kOrderThreads / 64];
  constexpr int kPer = 12;                                  // nodes per thread and trip: neighbours, so that the scan keeps their order
  const int g = blockIdx.x;
  const int n0 = gptr[g], n1 = gptr[g + 1];
  const int base = new_gptr[g], lim = new_gptr[g + 1];
  int run = 0;
  for (int v0 = n0; v0 < n1; v0 += kOrderThreads * kPer) {
    int s[kPer], mine = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int v = v0 + (int)threadIdx.x * kPer + k;
      s[k] = slot[min(v, n1 - 1)];
      s[k] = v < n1 ? s[k] : -1;
      mine += s[k] >= 0 ? 1 : 0;
    }
    int tot;
    int pos = base + run + block_exclusive_scan<kOrderThreads / 64>(mine, tmp, tot);
#pragma unroll
    for (int k = 0; k < kPer; ++k)
      if (s[k] >= 0) {
        if (pos < lim) order[pos] = s[k];
        ++pos;
      }
    run += tot;
  }
}

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_tile_order_by_position(const int32_t* slot, const int32_t* graph_ptr, const int32_t* new_graph_ptr,
                                            int64_t num_graphs, int32_t* order, mlqem_stream_t stream) {
  begin_launches();
  if (num_graphs < 0) return MLQEM_ERR_BAD_ARG;
  if (num_graphs == 0) return MLQEM_OK;
  if (!slot || !graph_ptr || !new_graph_ptr || !order) return MLQEM_ERR_BAD_ARG;
  if (num_graphs > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(tile_order_kernel, dim3((unsigned)num_graphs), dim3(kOrderThreads), 0, as_stream(stream), slot, graph_ptr, new_graph_ptr,
                     order);
  return launch_status();
}

