// ASAPooling's data-dependent steps: per-graph top-k by fitness and the
// coarsened connectivity pattern of S^T A S.  Sizes depend on the data, so every step is split into a
// count/size query and a fill, with the (tiny) totals read back by the host in between.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace mlqem {

// ------------------------------------------------------------------------------------------ per-graph top-k
// key = (order-preserving bits of fitness) << 32 | (0xFFFFFFFF - local index): a DESCENDING sort of these unique
// keys lists a graph's nodes by descending fitness with ties broken by the lower index, whatever algorithm the
// segmented sort picks for the segment size.
__global__ __launch_bounds__(kBlock) void topk_keys_kernel(const float* __restrict__ fitness,
                                                           const int32_t* __restrict__ gptr, int B, int64_t N,
                                                           uint64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  int lo = 0, hi = B;  // graph of node i: largest g with gptr[g] <= i
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gptr[mid] <= i) lo = mid; else hi = mid;
  }
  const uint32_t bits = __float_as_uint(fitness[i]);
  const uint32_t ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
  const uint32_t local = (uint32_t)(i - gptr[lo]);
  keys[i] = ((uint64_t)ord << 32) | (uint64_t)(0xFFFFFFFFu - local);
}

__global__ __launch_bounds__(kBlock) void topk_select_kernel(const uint64_t* __restrict__ sorted,
                                                             const int32_t* __restrict__ gptr,
                                                             const int32_t* __restrict__ new_gptr, int B,
                                                             int64_t K, int32_t* __restrict__ perm) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= K) return;
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (new_gptr[mid] <= p) lo = mid; else hi = mid;
  }
  const int64_t r = p - new_gptr[lo];
  const uint64_t key = sorted[gptr[lo] + r];
  perm[p] = gptr[lo] + (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
}

static size_t topk_temp_bytes(int64_t N, int64_t B) {
  size_t temp = 0;
  (void)rocprim::segmented_radix_sort_keys_desc(nullptr, temp, (uint64_t*)nullptr, (uint64_t*)nullptr, (unsigned)N,
                                                (unsigned)B, (const int32_t*)nullptr, (const int32_t*)nullptr, 0, 64,
                                                (hipStream_t)0);
  return (temp + 255) / 256 * 256;
}

// ------------------------------------------------------------------------------- coarsened connectivity
// New edge (p -> q), p != q, iff some member u of cluster p (u -> c_p or u == c_p) has an edge u -> v (or u == v) to
// some member v of cluster q (v -> c_q or v == c_q).  Per kept centre: in-neighbours -> their out-neighbours ->
// their kept out-neighbours, self included at every hop.
struct CoarsenArgs {
  const int32_t* in_ptr; const int32_t* in_src; const int32_t* out_ptr; const int32_t* out_dst;
  const int32_t* perm; const int32_t* slot; int64_t K;
};

template <bool FILL>
__global__ __launch_bounds__(kBlock) void coarsen_kernel(const CoarsenArgs a, int64_t* __restrict__ counts,
                                                         const int64_t* __restrict__ offsets,
                                                         uint64_t* __restrict__ keys) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.K) return;
  const int c = a.perm[p];
  int64_t n = 0;
  int64_t pos = FILL ? offsets[p] : 0;
  auto visit_w = [&](int w) {
    const int q = a.slot[w];
    if (q >= 0 && q != (int)p) {
      if (FILL) keys[pos++] = ((uint64_t)p << 32) | (uint32_t)q;
      else ++n;
    }
  };
  auto visit_v = [&](int v) {
    visit_w(v);
    for (int e = a.out_ptr[v]; e < a.out_ptr[v + 1]; ++e) visit_w(a.out_dst[e]);
  };
  auto visit_u = [&](int u) {
    visit_v(u);
    for (int e = a.out_ptr[u]; e < a.out_ptr[u + 1]; ++e) visit_v(a.out_dst[e]);
  };
  visit_u(c);
  for (int e = a.in_ptr[c]; e < a.in_ptr[c + 1]; ++e) visit_u(a.in_src[e]);
  if (!FILL) counts[p] = n;
}

__global__ __launch_bounds__(kBlock) void slot_map_kernel(const int32_t* __restrict__ perm, int64_t K,
                                                          int32_t* __restrict__ slot) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p < K) slot[perm[p]] = (int32_t)p;
}

__global__ __launch_bounds__(kBlock) void keys_to_edge_index_kernel(const uint64_t* __restrict__ keys, int64_t E,
                                                                    int64_t* __restrict__ ei) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  ei[e] = (int64_t)(keys[e] >> 32);
  ei[E + e] = (int64_t)(keys[e] & 0xFFFFFFFFull);
}

}  // namespace mlqem

using namespace mlqem;

extern "C" size_t mlqem_segment_topk_workspace_bytes(int64_t N, int64_t B) {
  if (N <= 0 || B <= 0) return 256;
  const size_t keys = ((size_t)N * sizeof(uint64_t) + 255) / 256 * 256;
  return 2 * keys + topk_temp_bytes(N, B);
}

extern "C" int mlqem_segment_topk(const float* fitness, const int32_t* graph_ptr, const int32_t* new_graph_ptr,
                                  int64_t N, int64_t B, int64_t K, int32_t* perm, void* workspace,
                                  size_t workspace_bytes, mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || B <= 0 || K < 0 || K > N || N >= 0x7fffffffLL) return MLQEM_ERR_BAD_ARG;
  if (N == 0 || K == 0) return MLQEM_OK;
  if (!fitness || !graph_ptr || !new_graph_ptr || !perm) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_segment_topk_workspace_bytes(N, B)) return MLQEM_ERR_WORKSPACE;
  const size_t kb = ((size_t)N * sizeof(uint64_t) + 255) / 256 * 256;
  char* ws = static_cast<char*>(workspace);
  uint64_t* keys = reinterpret_cast<uint64_t*>(ws);
  uint64_t* sorted = reinterpret_cast<uint64_t*>(ws + kb);
  void* temp = ws + 2 * kb;
  size_t temp_bytes = topk_temp_bytes(N, B);
  hipLaunchKernelGGL(topk_keys_kernel, dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, stream, fitness,
                     graph_ptr, (int)B, N, keys);
  if (rocprim::segmented_radix_sort_keys_desc(temp, temp_bytes, keys, sorted, (unsigned)N, (unsigned)B, graph_ptr,
                                              graph_ptr + 1, 0, 64, stream) != hipSuccess)
    return MLQEM_ERR_LAUNCH;
  hipLaunchKernelGGL(topk_select_kernel, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, stream, sorted,
                     graph_ptr, new_graph_ptr, (int)B, K, perm);
  return launch_status();
}

extern "C" size_t mlqem_asap_coarsen_workspace_bytes(int64_t K) {
  size_t temp = 0;
  (void)rocprim::exclusive_scan(nullptr, temp, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0,
                                (size_t)(K > 0 ? K + 1 : 1), rocprim::plus<int64_t>(), (hipStream_t)0);
  return (temp + 255) / 256 * 256 + ((size_t)(K + 1) * sizeof(int64_t) + 255) / 256 * 256;
}

// Pass 1: slot[N] (cluster id of every kept centre, -1 elsewhere) and offsets[K+1] = exclusive scan of the
// candidate pairs per cluster; offsets[K] is the total the caller reads back to size `keys`.
extern "C" int mlqem_asap_coarsen_count(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                        const int32_t* out_dst, const int32_t* perm, int64_t N, int64_t K,
                                        int32_t* slot, int64_t* offsets, void* workspace, size_t workspace_bytes,
                                        mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || K < 0 || K > N) return MLQEM_ERR_BAD_ARG;
  if (!slot || !offsets) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_asap_coarsen_workspace_bytes(K)) return MLQEM_ERR_WORKSPACE;
  if (N > 0 && hipMemsetAsync(slot, 0xFF, sizeof(int32_t) * (size_t)N, stream) != hipSuccess) return MLQEM_ERR_LAUNCH;
  const size_t cb = ((size_t)(K + 1) * sizeof(int64_t) + 255) / 256 * 256;
  int64_t* counts = static_cast<int64_t*>(workspace);
  void* temp = static_cast<char*>(workspace) + cb;
  size_t temp_bytes = workspace_bytes - cb;
  if (hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)(K + 1), stream) != hipSuccess) return MLQEM_ERR_LAUNCH;
  if (K > 0) {
    if (!in_ptr || !out_ptr || !perm) return MLQEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(slot_map_kernel, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, stream, perm, K, slot);
    CoarsenArgs a{in_ptr, in_src, out_ptr, out_dst, perm, slot, K};
    hipLaunchKernelGGL(coarsen_kernel<false>, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, stream, a, counts,
                       (const int64_t*)nullptr, (uint64_t*)nullptr);
  }
  if (rocprim::exclusive_scan(temp, temp_bytes, counts, offsets, (int64_t)0, (size_t)(K + 1), rocprim::plus<int64_t>(),
                              stream) != hipSuccess)
    return MLQEM_ERR_LAUNCH;
  return launch_status();
}

// Pass 2: keys[offsets[p] ...] = (p << 32 | q) for every candidate pair (duplicates included).
extern "C" int mlqem_asap_coarsen_fill(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                       const int32_t* out_dst, const int32_t* perm, const int32_t* slot,
                                       const int64_t* offsets, int64_t K, uint64_t* keys, mlqem_stream_t stream) {
  begin_launches();
  if (K < 0) return MLQEM_ERR_BAD_ARG;
  if (K == 0) return MLQEM_OK;
  if (!in_ptr || !out_ptr || !perm || !slot || !offsets || !keys) return MLQEM_ERR_BAD_ARG;
  CoarsenArgs a{in_ptr, in_src, out_ptr, out_dst, perm, slot, K};
  hipLaunchKernelGGL(coarsen_kernel<true>, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, as_stream(stream), a,
                     (int64_t*)nullptr, offsets, keys);
  return launch_status();
}

extern "C" size_t mlqem_sort_unique_u64_workspace_bytes(int64_t T) {
  if (T <= 0) return 256;
  size_t t1 = 0, t2 = 0;
  (void)rocprim::radix_sort_keys(nullptr, t1, (uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)T, 0, 64,
                                 (hipStream_t)0);
  (void)rocprim::unique(nullptr, t2, (uint64_t*)nullptr, (uint64_t*)nullptr, (int64_t*)nullptr, (size_t)T,
                        rocprim::equal_to<uint64_t>(), (hipStream_t)0);
  const size_t kb = ((size_t)T * sizeof(uint64_t) + 255) / 256 * 256;
  return kb + (std::max(t1, t2) + 255) / 256 * 256;
}

// Sorts `keys` ascending and writes the distinct values to out_keys; *out_count (device int64) = how many.
extern "C" int mlqem_sort_unique_u64(const uint64_t* keys, int64_t T, uint64_t* out_keys, int64_t* out_count,
                                     void* workspace, size_t workspace_bytes, mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (T < 0 || !out_count) return MLQEM_ERR_BAD_ARG;
  if (T == 0) {
    if (hipMemsetAsync(out_count, 0, sizeof(int64_t), stream) != hipSuccess) return MLQEM_ERR_LAUNCH;
    return MLQEM_OK;
  }
  if (!keys || !out_keys) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_sort_unique_u64_workspace_bytes(T)) return MLQEM_ERR_WORKSPACE;
  const size_t kb = ((size_t)T * sizeof(uint64_t) + 255) / 256 * 256;
  uint64_t* sorted = static_cast<uint64_t*>(workspace);
  void* temp = static_cast<char*>(workspace) + kb;
  size_t temp_bytes = workspace_bytes - kb;
  if (rocprim::radix_sort_keys(temp, temp_bytes, keys, sorted, (size_t)T, 0, 64, stream) != hipSuccess)
    return MLQEM_ERR_LAUNCH;
  if (rocprim::unique(temp, temp_bytes, sorted, out_keys, out_count, (size_t)T, rocprim::equal_to<uint64_t>(),
                      stream) != hipSuccess)
    return MLQEM_ERR_LAUNCH;
  return launch_status();
}

extern "C" int mlqem_keys_to_edge_index(const uint64_t* keys, int64_t E, int64_t* edge_index, mlqem_stream_t stream) {
  begin_launches();
  if (E < 0) return MLQEM_ERR_BAD_ARG;
  if (E == 0) return MLQEM_OK;
  if (!keys || !edge_index) return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(keys_to_edge_index_kernel, dim3((unsigned)ceil_div(E, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), keys, E, edge_index);
  return launch_status();
}
