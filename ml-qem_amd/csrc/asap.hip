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
// some member v of cluster q (v -> c_q or v == c_q).  Done in two hops with a sort-unique in between, because counting
// every (u, v, w) PATH explodes around hub nodes (a 100-wire barrier contributes ~10^6 paths but ~10^4 distinct pairs):
//   hop 1: (p, v) for v in N+[u], u in N-[c_p]      (N+/- include the node itself)   -> distinct (p, v)
//   hop 2: (p, slot[w]) for w in N+[v], w kept, slot[w] != p                           -> distinct (p, q)
struct Hop1Args {
  const int32_t* in_ptr; const int32_t* in_src; const int32_t* out_ptr; const int32_t* out_dst;
  const int32_t* perm; int64_t K;
};

template <bool FILL>
__global__ __launch_bounds__(kBlock) void coarsen_hop1_kernel(const Hop1Args a, int64_t* __restrict__ counts,
                                                              const int64_t* __restrict__ offsets,
                                                              uint64_t* __restrict__ keys) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.K) return;
  const int c = a.perm[p];
  int64_t n = 0;
  int64_t pos = FILL ? offsets[p] : 0;
  auto visit_u = [&](int u) {
    if (FILL) {
      keys[pos++] = ((uint64_t)p << 32) | (uint32_t)u;
      for (int e = a.out_ptr[u]; e < a.out_ptr[u + 1]; ++e) keys[pos++] = ((uint64_t)p << 32) | (uint32_t)a.out_dst[e];
    } else {
      n += 1 + (a.out_ptr[u + 1] - a.out_ptr[u]);
    }
  };
  visit_u(c);
  for (int e = a.in_ptr[c]; e < a.in_ptr[c + 1]; ++e) visit_u(a.in_src[e]);
  if (!FILL) counts[p] = n;
}

template <bool FILL>
__global__ __launch_bounds__(kBlock) void coarsen_hop2_kernel(const uint64_t* __restrict__ pairs, int64_t M,
                                                              const int32_t* __restrict__ out_ptr,
                                                              const int32_t* __restrict__ out_dst,
                                                              const int32_t* __restrict__ slot,
                                                              int64_t* __restrict__ counts,
                                                              const int64_t* __restrict__ offsets,
                                                              uint64_t* __restrict__ keys) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= M) return;
  const uint64_t pv = pairs[t];
  const int p = (int)(pv >> 32), v = (int)(pv & 0xFFFFFFFFull);
  int64_t n = 0;
  int64_t pos = FILL ? offsets[t] : 0;
  auto visit_w = [&](int w) {
    const int q = slot[w];
    if (q >= 0 && q != p) {
      if (FILL) keys[pos++] = ((uint64_t)p << 32) | (uint32_t)q;
      else ++n;
    }
  };
  visit_w(v);
  for (int e = out_ptr[v]; e < out_ptr[v + 1]; ++e) visit_w(out_dst[e]);
  if (!FILL) counts[t] = n;
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

static size_t scan_bytes(int64_t n) {
  size_t temp = 0;
  (void)rocprim::exclusive_scan(nullptr, temp, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0,
                                (size_t)(n > 0 ? n + 1 : 1), rocprim::plus<int64_t>(), (hipStream_t)0);
  return (temp + 255) / 256 * 256 + ((size_t)(n + 1) * sizeof(int64_t) + 255) / 256 * 256;
}

extern "C" size_t mlqem_asap_coarsen_workspace_bytes(int64_t K) { return scan_bytes(K); }

static int scan_counts(int64_t* counts, int64_t n, int64_t* offsets, void* temp, size_t temp_bytes, hipStream_t stream) {
  if (rocprim::exclusive_scan(temp, temp_bytes, counts, offsets, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(),
                              stream) != hipSuccess)
    return MLQEM_ERR_LAUNCH;
  return MLQEM_OK;
}

// Hop 1, pass 1: slot[N] (cluster id of every kept centre, -1 elsewhere) and offsets[K+1] = exclusive scan of the
// (p, v) candidates per cluster; offsets[K] is the total the caller reads back to size `keys`.
extern "C" int mlqem_asap_hop1_count(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                     const int32_t* out_dst, const int32_t* perm, int64_t N, int64_t K, int32_t* slot,
                                     int64_t* offsets, void* workspace, size_t workspace_bytes,
                                     mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || K < 0 || K > N) return MLQEM_ERR_BAD_ARG;
  if (!slot || !offsets) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < scan_bytes(K)) return MLQEM_ERR_WORKSPACE;
  if (N > 0 && hipMemsetAsync(slot, 0xFF, sizeof(int32_t) * (size_t)N, stream) != hipSuccess) return MLQEM_ERR_LAUNCH;
  const size_t cb = ((size_t)(K + 1) * sizeof(int64_t) + 255) / 256 * 256;
  int64_t* counts = static_cast<int64_t*>(workspace);
  if (hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)(K + 1), stream) != hipSuccess) return MLQEM_ERR_LAUNCH;
  if (K > 0) {
    if (!in_ptr || !out_ptr || !perm) return MLQEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(slot_map_kernel, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, stream, perm, K, slot);
    Hop1Args a{in_ptr, in_src, out_ptr, out_dst, perm, K};
    hipLaunchKernelGGL(coarsen_hop1_kernel<false>, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, stream, a,
                       counts, (const int64_t*)nullptr, (uint64_t*)nullptr);
  }
  const int rc = scan_counts(counts, K, offsets, static_cast<char*>(workspace) + cb, workspace_bytes - cb, stream);
  return rc != MLQEM_OK ? rc : launch_status();
}

extern "C" int mlqem_asap_hop1_fill(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                    const int32_t* out_dst, const int32_t* perm, const int64_t* offsets, int64_t K,
                                    uint64_t* keys, mlqem_stream_t stream) {
  begin_launches();
  if (K < 0) return MLQEM_ERR_BAD_ARG;
  if (K == 0) return MLQEM_OK;
  if (!in_ptr || !out_ptr || !perm || !offsets || !keys) return MLQEM_ERR_BAD_ARG;
  Hop1Args a{in_ptr, in_src, out_ptr, out_dst, perm, K};
  hipLaunchKernelGGL(coarsen_hop1_kernel<true>, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), a, (int64_t*)nullptr, offsets, keys);
  return launch_status();
}

// Hop 2 over the M distinct (p, v) pairs of hop 1: offsets[M+1] (count pass), then keys = (p << 32 | q).
extern "C" int mlqem_asap_hop2_count(const uint64_t* pairs, int64_t M, const int32_t* out_ptr, const int32_t* out_dst,
                                     const int32_t* slot, int64_t* offsets, void* workspace, size_t workspace_bytes,
                                     mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (M < 0 || !offsets) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < scan_bytes(M)) return MLQEM_ERR_WORKSPACE;
  const size_t cb = ((size_t)(M + 1) * sizeof(int64_t) + 255) / 256 * 256;
  int64_t* counts = static_cast<int64_t*>(workspace);
  if (hipMemsetAsync(counts + M, 0, sizeof(int64_t), stream) != hipSuccess) return MLQEM_ERR_LAUNCH;
  if (M > 0) {
    if (!pairs || !out_ptr || !slot) return MLQEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(coarsen_hop2_kernel<false>, dim3((unsigned)ceil_div(M, kBlock)), dim3(kBlock), 0, stream, pairs, M,
                       out_ptr, out_dst, slot, counts, (const int64_t*)nullptr, (uint64_t*)nullptr);
  }
  const int rc = scan_counts(counts, M, offsets, static_cast<char*>(workspace) + cb, workspace_bytes - cb, stream);
  return rc != MLQEM_OK ? rc : launch_status();
}

extern "C" int mlqem_asap_hop2_fill(const uint64_t* pairs, int64_t M, const int32_t* out_ptr, const int32_t* out_dst,
                                    const int32_t* slot, const int64_t* offsets, uint64_t* keys,
                                    mlqem_stream_t stream) {
  begin_launches();
  if (M < 0) return MLQEM_ERR_BAD_ARG;
  if (M == 0) return MLQEM_OK;
  if (!pairs || !out_ptr || !slot || !offsets || !keys) return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(coarsen_hop2_kernel<true>, dim3((unsigned)ceil_div(M, kBlock)), dim3(kBlock), 0, as_stream(stream),
                     pairs, M, out_ptr, out_dst, slot, (int64_t*)nullptr, offsets, keys);
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
