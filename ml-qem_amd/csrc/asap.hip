// ASAPooling's data-dependent steps: per-graph top-k by fitness and the
// coarsened connectivity pattern of S^T A S.  Sizes depend on the data, so every step is split into a
// count/size query and a fill, with the (tiny) totals read back by the host in between.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace mlqem {

// ------------------------------------------------------------------------------------------ per-graph top-k
// key = (order-preserving bits of fitness) << b | (2^b - 1 - local index), b = bits of the largest local index: a
// DESCENDING sort of these unique keys lists a graph's nodes by descending fitness with ties broken by the lower index,
// whatever algorithm the segmented sort picks for the segment size, and the radix sort walks 32 + b bits instead of 64.
// graph_bits > 0: the key also carries (B - 1 - graph) above the fitness bits, so that ONE device-wide descending sort lists the
// graphs in ascending order, each by descending fitness (batches of large graphs: see mlqem_segment_topk).
__global__ __launch_bounds__(kBlock) void topk_keys_kernel(const float* __restrict__ fitness,
                                                           const int32_t* __restrict__ gptr, int B, int64_t N,
                                                           int idx_bits, int graph_bits, uint64_t* __restrict__ keys) {
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
  uint64_t key = ((uint64_t)ord << idx_bits) | (uint64_t)(((1u << idx_bits) - 1u) - local);
  if (graph_bits > 0) key |= (uint64_t)(B - 1 - lo) << (32 + idx_bits);
  keys[i] = key;
}

__global__ __launch_bounds__(kBlock) void topk_select_kernel(const uint64_t* __restrict__ sorted,
                                                             const int32_t* __restrict__ gptr,
                                                             const int32_t* __restrict__ new_gptr, int B,
                                                             int64_t K, int idx_bits, int32_t* __restrict__ perm) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= K) return;
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (new_gptr[mid] <= p) lo = mid; else hi = mid;
  }
  const int64_t r = p - new_gptr[lo];
  const uint64_t key = sorted[gptr[lo] + r];
  const uint32_t mask = (1u << idx_bits) - 1u;
  perm[p] = gptr[lo] + (int32_t)(mask - ((uint32_t)key & mask));
}

// The small-tensor algebra around ASAPooling's composed score (native/functional.py: a_i = (att_q W) . segmax_i + (att_q . b + att_b),
// LEConv's three one-wide projections as one [3, D] matrix) as ONE launch per direction instead of ten element-wise / reduce / cat
// launches of the host framework each (a 5 us launch each: a fifth of a 32-circuit step of the reference's batch size).
//   forward:  w_comp[j] = sum_i att_q[i] lin_w[i, j];  b_comp = att_q . lin_b + att_b;  att_q, att_x = the halves of att_w;
//             w3 = [l1_w; l2_w; l3_w];  b3 = (l1_b, 0, l3_b)
//   backward: g_att_q[i] = sum_j g_w_comp[j] lin_w[i, j] + g_att_b lin_b[i];  g_lin_w[i, j] = att_q[i] g_w_comp[j];
//             g_lin_b[i] = g_att_b att_q[i];  g_att_w = (g_att_q | g_att_x)
constexpr int kComposeLds = 64;                            // lin_w of up to 64 x 64 is staged in LDS (the reference's poolings: 45, 30)
__global__ __launch_bounds__(kBlock) void asap_compose_kernel(const float* __restrict__ lin_w, const float* __restrict__ lin_b,
                                                              const float* __restrict__ att_w, const float* __restrict__ att_b,
                                                              const float* __restrict__ l1_w, const float* __restrict__ l1_b,
                                                              const float* __restrict__ l2_w, const float* __restrict__ l3_w,
                                                              const float* __restrict__ l3_b, int D, float* __restrict__ w_comp,
                                                              float* __restrict__ b_comp, float* __restrict__ att_q, float* __restrict__ att_x,
                                                              float* __restrict__ w3, float* __restrict__ b3) {
  __shared__ float lw[kComposeLds * kComposeLds], aq[kComposeLds], lb[kComposeLds];
  const bool staged = D <= kComposeLds;                  // one coalesced round trip instead of D dependent ones per thread
  if (staged) {
    for (int t = threadIdx.x; t < D * D; t += kBlock) lw[t] = lin_w[t];
    for (int t = threadIdx.x; t < D; t += kBlock) { aq[t] = att_w[t]; lb[t] = lin_b[t]; }
    __syncthreads();
  }
  for (int j = threadIdx.x; j < D; j += kBlock) {
    float acc = 0.f;
    if (staged) {
      for (int i = 0; i < D; ++i) acc = fmaf(aq[i], lw[i * D + j], acc);
    } else {
      for (int i = 0; i < D; ++i) acc = fmaf(att_w[i], lin_w[(int64_t)i * D + j], acc);
    }
    w_comp[j] = acc;
    att_q[j] = att_w[j];
    att_x[j] = att_w[D + j];
    w3[j] = l1_w[j];
    w3[D + j] = l2_w[j];
    w3[2 * D + j] = l3_w[j];
  }
  if (threadIdx.x == kBlock - 1) {                       // (a lane of the last wave: the first ones hold the columns)
    float acc = 0.f;
    if (staged) {
      for (int i = 0; i < D; ++i) acc = fmaf(aq[i], lb[i], acc);
    } else {
      for (int i = 0; i < D; ++i) acc = fmaf(att_w[i], lin_b[i], acc);
    }
    b_comp[0] = acc + att_b[0];
    b3[0] = l1_b[0];
    b3[1] = 0.f;
    b3[2] = l3_b[0];
  }
}

__global__ __launch_bounds__(kBlock) void asap_compose_bwd_kernel(const float* __restrict__ g_w_comp, const float* __restrict__ g_att_b,
                                                                  const float* __restrict__ lin_w, const float* __restrict__ lin_b,
                                                                  const float* __restrict__ att_w, const float* __restrict__ g_att_x, int D,
                                                                  float* __restrict__ g_lin_w, float* __restrict__ g_lin_b,
                                                                  float* __restrict__ g_att_w) {
  __shared__ float lw[kComposeLds * kComposeLds], gw[kComposeLds], aq[kComposeLds];
  const bool staged = D <= kComposeLds;
  if (staged) {
    for (int t = threadIdx.x; t < D * D; t += kBlock) lw[t] = lin_w[t];
    for (int t = threadIdx.x; t < D; t += kBlock) { gw[t] = g_w_comp[t]; aq[t] = att_w[t]; }
    __syncthreads();
  }
  const float gb = g_att_b[0];
  for (int i = threadIdx.x; i < D; i += kBlock) {
    float acc = 0.f;
    if (staged) {
      for (int j = 0; j < D; ++j) acc = fmaf(gw[j], lw[i * D + j], acc);
    } else {
      for (int j = 0; j < D; ++j) acc = fmaf(g_w_comp[j], lin_w[(int64_t)i * D + j], acc);
    }
    g_att_w[i] = acc + gb * lin_b[i];
    g_att_w[D + i] = g_att_x[i];
    g_lin_b[i] = gb * att_w[i];
  }
  if (staged) {
    for (int t = threadIdx.x; t < D * D; t += kBlock) g_lin_w[t] = aq[t / D] * gw[t % D];
  } else {
    for (int t = threadIdx.x; t < D * D; t += kBlock) g_lin_w[t] = att_w[t / D] * g_w_comp[t % D];
  }
}

// Top-k of LARGE graphs (thousands of nodes: 100-qubit circuits) in two launches (round 5; was one device-wide merge sort: 19-22
// launches of 5-7 us per pooling, 0.26 ms of a 64-circuit step).
//   topk_chunk_sort_kernel: a workgroup sorts one CHUNK of kTopkChunk consecutive nodes of one graph in LDS (bitonic, descending; the
//     key = ordered fitness bits above (2^32 - 1 - local index): unique inside a graph, equal fitness -> lower index first, the order
//     of the keys above) and writes it back in place of the chunk.  Workgroup w -> (graph, chunk) by cstart[g] = gptr[g] / chunk + g.
//   topk_rank_select_kernel: a thread per node: its rank in its graph = its place in its chunk + for every other chunk of the graph the
//     number of keys above it (a binary search in a sorted chunk); rank < k_g -> perm[new_gptr[g] + rank] = the node.  A thread
//     stops as soon as its rank reaches k_g.
template <int kTopkChunk>
__device__ __forceinline__ int topk_graph_of_chunk(const int32_t* __restrict__ gptr, int B, int w) {
  int lo = 0, hi = B;                                    // largest g with cstart[g] <= w
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gptr[mid] / kTopkChunk + mid <= w) lo = mid; else hi = mid;
  }
  return lo;
}

template <int kTopkChunk>
__global__ __launch_bounds__(kBlock) void topk_chunk_sort_kernel(const float* __restrict__ fitness, const int32_t* __restrict__ gptr, int B,
                                                                 uint64_t* __restrict__ sorted) {
  constexpr int kTopkPer = kTopkChunk / kBlock;
  __shared__ uint64_t key[kTopkChunk];
  const int w = blockIdx.x, tid = threadIdx.x;
  const int g = topk_graph_of_chunk<kTopkChunk>(gptr, B, w);
  const int g0 = gptr[g], n = gptr[g + 1] - g0;
  const int c = w - (g0 / kTopkChunk + g);
  const int first = c * kTopkChunk;
  if (first >= n) return;                                // (the whole workgroup)
  const int cnt = min(kTopkChunk, n - first);
#pragma unroll
  for (int k = 0; k < kTopkPer; ++k) {
    const int i = k * kBlock + tid;
    uint64_t v = 0;                                      // pads: below every key
    if (i < cnt) {
      const uint32_t bits = __float_as_uint(fitness[g0 + first + i]);
      const uint32_t ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
      v = ((uint64_t)ord << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(first + i));
    }
    key[i] = v;
  }
  __syncthreads();
  for (int size = 2; size <= kTopkChunk; size <<= 1) {
    for (int ls = 31 - __clz(size) - 1; ls >= 0; --ls) {  // stride = 1 << ls (shifts and masks: a variable division costs ~40 instructions)
      const int stride = 1 << ls;
#pragma unroll
      for (int k = 0; k < kTopkPer / 2; ++k) {
        const int t = k * kBlock + tid;                  // compare-exchange number t of this pass
        const int lo = ((t >> ls) << (ls + 1)) | (t & (stride - 1)), hi = lo + stride;
        const bool desc = ((lo & size) == 0);            // descending overall: the blocks with this bit clear sort downwards
        const uint64_t a = key[lo], b = key[hi];
        if ((a < b) == desc) { key[lo] = b; key[hi] = a; }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int k = 0; k < kTopkPer; ++k) {
    const int i = k * kBlock + tid;
    if (i < cnt) sorted[g0 + first + i] = key[i];
  }
}

// Top-k of SMALL graphs (at most SIZE <= 1024 nodes each: every dataset the reference ships and the 4- / 20-qubit corpora) in ONE
// launch: a workgroup per graph sorts the graph's keys in LDS with a bitonic network sized for the batch's largest graph and writes
// the first k_g of them.  Same keys as the other forms (fitness descending, equal fitness by index), so the same perm.  Before: a key
// launch, rocprim's segmented radix sort and a select launch -- 25-30 us of a 0.4 ms train step on 32 four-qubit circuits, twice.
template <int SIZE>
__global__ __launch_bounds__(kBlock) void topk_small_kernel(const float* __restrict__ fitness, const int32_t* __restrict__ gptr,
                                                            const int32_t* __restrict__ new_gptr, int32_t* __restrict__ perm,
                                                            int32_t* __restrict__ slot) {
  __shared__ uint64_t key[SIZE];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int g0 = gptr[g], n = min(SIZE, gptr[g + 1] - g0);
  const int k0 = new_gptr[g], keep = min(n, new_gptr[g + 1] - k0);
  // the network is sized for THIS graph (the next power of two from its node count), not for the batch's largest: SIZE is a bound the
  // caller vouches for, and a size-stable bucket's bound is its filler graphs' 1 024 nodes where a four-qubit circuit has ~50 -- a
  // 1 024-key network (55 passes of 512 exchanges) per 50-node graph was 20 us of a 0.4 ms train step
  int width = 2;
  while (width < n) width <<= 1;
  for (int i = tid; i < width; i += kBlock) {
    uint64_t v = 0;                                      // pads: below every key
    if (i < n) {
      const uint32_t bits = __float_as_uint(fitness[g0 + i]);
      const uint32_t ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
      v = ((uint64_t)ord << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
    }
    key[i] = v;
  }
  __syncthreads();
  for (int size = 2; size <= width; size <<= 1) {
    for (int ls = 31 - __clz(size) - 1; ls >= 0; --ls) {
      const int stride = 1 << ls;
      for (int t = tid; t < width / 2; t += kBlock) {
        const int lo = ((t >> ls) << (ls + 1)) | (t & (stride - 1)), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const uint64_t a = key[lo], b = key[hi];
        if ((a < b) == desc) { key[lo] = b; key[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int r = tid; r < keep; r += kBlock) perm[k0 + r] = g0 + (int32_t)(0xFFFFFFFFu - (uint32_t)key[r]);
  // slot[] (mlqem_asap_slot_map: the cluster of every kept centre, -1 elsewhere) from the same sorted keys: no launch of its own
  if (slot)
    for (int r = tid; r < n; r += kBlock) slot[g0 + (int32_t)(0xFFFFFFFFu - (uint32_t)key[r])] = r < keep ? k0 + r : -1;
}

// (Round 6 tried two other forms of this pass on the step's graphs of 2 000-20 000 nodes: a workgroup per chunk with the other chunks
// staged in LDS -- 107-115 us at chunks of 1 024, too few workgroups -- and this form with the searches of four chunks interleaved,
// branch-free: 103 against 101 us, the pass is bound by the number of loads, not by their latency.)
template <int kTopkChunk>
__global__ __launch_bounds__(kBlock) void topk_rank_select_kernel(const uint64_t* __restrict__ sorted, const int32_t* __restrict__ gptr,
                                                                  const int32_t* __restrict__ new_gptr, int B, int64_t N,
                                                                  int32_t* __restrict__ perm, int32_t* __restrict__ slot) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= N) return;
  int lo = 0, hi = B;                                    // graph of element e
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gptr[mid] <= e) lo = mid; else hi = mid;
  }
  const int g = lo, g0 = gptr[g], n = gptr[g + 1] - g0;
  const int keep = new_gptr[g + 1] - new_gptr[g];
  const int local = (int)(e - g0), c = local / kTopkChunk;
  int rank = local - c * kTopkChunk;                     // its place in its own (sorted) chunk
  if (rank >= keep && !slot) return;
  const uint64_t mine = sorted[e];
  const int node = g0 + (int32_t)(0xFFFFFFFFu - (uint32_t)mine);
  const int nch = (n + kTopkChunk - 1) / kTopkChunk;
  for (int o = 0; o < nch && rank < keep; ++o) {
    if (o == c) continue;
    const uint64_t* __restrict__ ch = sorted + g0 + (int64_t)o * kTopkChunk;
    int a = 0, b = min(kTopkChunk, n - o * kTopkChunk);  // the number of keys of chunk o above `mine` (the chunk descends)
    while (a < b) {
      const int mid = (a + b) >> 1;
      if (ch[mid] > mine) a = mid + 1; else b = mid;
    }
    rank += a;
  }
  const bool kept = rank < keep;
  if (kept) perm[new_gptr[g] + rank] = node;
  // slot[] (mlqem_asap_slot_map: the cluster of every kept centre, -1 elsewhere): every node is one sorted key -- no launch of its own
  if (slot) slot[node] = kept ? new_gptr[g] + rank : -1;
}

// The device-wide sort of the top-k always takes rocprim's MERGE sort: above 2^20 keys the default configuration switches to
// Onesweep, which clears its digit counters and look-back states with hipMemsetAsync -- memset nodes of a captured hipGraph, and a
// captured Family B step on 256 100-qubit circuits (2.8 M keys) died in its second replay inside
// radix_sort_onesweep_iteration (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION), the same failure hipMemsetAsync gave in this
// file's own fills (see fill_i32_kernel).  The merge sort launches kernels only.
using TopkSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, (size_t(1) << 40)>;

static size_t topk_temp_bytes(int64_t N, int64_t B) {     // enough for either sort
  size_t seg = 0, whole = 0;
  (void)rocprim::segmented_radix_sort_keys_desc(nullptr, seg, (uint64_t*)nullptr, (uint64_t*)nullptr, (unsigned)N,
                                                (unsigned)B, (const int32_t*)nullptr, (const int32_t*)nullptr, 0, 64,
                                                (hipStream_t)0);
  (void)rocprim::radix_sort_keys_desc<TopkSortConfig>(nullptr, whole, (uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)N, 0, 64, (hipStream_t)0);
  return (std::max(seg, whole) + 255) / 256 * 256;
}

static int bits_for(int64_t values) {     // bits that hold 0 .. values - 1
  int b = 1;
  while (b < 31 && ((int64_t)1 << b) < values) ++b;
  return b;
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

// Fills of device arrays are KERNELS here, not hipMemsetAsync: these entry points run inside captured hipGraphs
// (train.BucketedTrainer): with hipMemsetAsync here the second replay of a captured Family B step ended in a GPU memory
// fault (the first was bit-identical to the eager step); with kernels it does not -- consistent with a replayed memset
// node running out of order with its neighbours (slot[] wiped after slot_map_kernel had written it, then used as an index).
__global__ __launch_bounds__(kBlock) void fill_i32_kernel(int32_t* __restrict__ p, int32_t v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) p[i] = v;
}
static inline void fill_i32(int32_t* p, int32_t v, int64_t n, hipStream_t stream) {
  if (n > 0) hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)ceil_div(n, (int64_t)kBlock)), dim3(kBlock), 0, stream, p, v, n);
}

// (The caller's overflow flag is STICKY: a launch ORs its own flag into it and never clears it, so a flag raised by any replay of a
// captured step is still there when the host reads it -- and resets it -- at the end of an epoch: coarsen_lists_copy_kernel.)

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

// ------------------------------------------------------------- coarsened connectivity, small graphs, no host sync
// The two-hop path above sizes its buffers from totals the host reads back (four 8-byte device->host copies per
// pooling: each one drains the queue).  For graphs whose pooled size k_g is at most kDenseMaxK -- every dataset the
// reference ships (<= 97 nodes per circuit) and the 4- and 20-qubit synthetic corpora -- the pooled adjacency of a graph
// fits in LDS as a k_g x k_g BIT matrix, and nothing data-dependent has to be known on the host:
//   kernel 1 (one workgroup per graph): for every node u, every pair (p, q), p in cl(u), q in cl(v), v in N+[u]
//            (cl(x) = kept centres among N+[x]; N+ includes the node) sets bit (p, q), p != q -- an idempotent LDS
//            atomicOr, so the result does not depend on the order; row / column popcounts give both degree vectors;
//   device scans turn the degrees into in_ptr / out_ptr (clusters are numbered graph by graph, so the global prefix
//            sums ARE the CSR pointers);
//   kernel 2 scans rows (out_dst, ascending q) and columns (in_src, ascending p: the order of the reference's coalesced
//            COO list) and links the two (out_eid).
// The host only provides a capacity: sum_g k_g (k_g - 1) entries, never touched beyond the true edge count.
constexpr int kDenseMaxK = 512;

struct DenseArgs {
  const int32_t* in_ptr; const int32_t* in_src; const int32_t* out_ptr; const int32_t* out_dst;
  const int32_t* gptr; const int32_t* new_gptr; const int32_t* slot;
  int32_t* loops; int64_t K;     // new_loops [K] (zeroed by the bitmap kernel, a graph's range per workgroup); K: the scans' last element
  int W;                         // 32-bit words per bitmap row (uniform over the batch)
  uint32_t* bitmaps;             // [B][kmax][W]
  uint32_t* bitmaps_t;           // [B][kmax][W]: the transposes (row q = the sources p with p -> q)
  int kmax;
  int32_t* outdeg; int32_t* indeg;   // [K + 1]
};

// Small graphs are STAGED: the graph's out-CSR and its nodes' cluster ids in LDS (a 4-qubit circuit: 230 nodes, 300 edges), so that
// the three nested walks below run at LDS latency (from global memory each level was a dependent round trip: 41 us for 32 graphs).
constexpr int kDenseStageNodes = 2 * kDenseMaxK + 2, kDenseStageEdges = 6144;
__global__ __launch_bounds__(kBlock) void coarsen_dense_bitmap_kernel(const DenseArgs a) {
  extern __shared__ uint32_t s_bm[];   // [kg][W]
  __shared__ int s_ptr[kDenseStageNodes + 1], s_slot[kDenseStageNodes], s_dst[kDenseStageEdges];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int n0 = a.gptr[g], n1 = a.gptr[g + 1];
  const int k0 = a.new_gptr[g], kg = a.new_gptr[g + 1] - k0;
  const int W = a.W;
  for (int i = tid; i < kg * W; i += kBlock) s_bm[i] = 0u;
  // (what used to be three fill launches in front of this kernel: a pooled graph has no self-loops, and the degree arrays end in a
  // zero so that their exclusive scans end in the totals)
  for (int i = tid; i < kg; i += kBlock) a.loops[k0 + i] = 0;
  if (g == 0 && tid == 0) { a.outdeg[a.K] = 0; a.indeg[a.K] = 0; }
  const int ng = n1 - n0, e0 = a.out_ptr[n0], eg = a.out_ptr[n1] - e0;
  const bool staged = ng <= kDenseStageNodes && eg <= kDenseStageEdges;
  if (staged) {
    for (int i = tid; i <= ng; i += kBlock) s_ptr[i] = a.out_ptr[n0 + i] - e0;
    for (int i = tid; i < ng; i += kBlock) s_slot[i] = a.slot[n0 + i];
    for (int i = tid; i < eg; i += kBlock) s_dst[i] = a.out_dst[e0 + i] - n0;      // (a graph's edges stay inside the graph)
  }
  __syncthreads();
  if (staged) {
    for (int u = tid; u < ng; u += kBlock) {
      const int ub = s_ptr[u], ue = s_ptr[u + 1];
      for (int ci = ub - 1; ci < ue; ++ci) {               // c over {u} + out(u)
        const int c = ci < ub ? u : s_dst[ci];
        const int p = s_slot[c];
        if (p < 0) continue;
        uint32_t* row = s_bm + (p - k0) * W;
        for (int vi = ub - 1; vi < ue; ++vi) {             // v over {u} + out(u)
          const int v = vi < ub ? u : s_dst[vi];
          const int vb = s_ptr[v], ve = s_ptr[v + 1];
          for (int di = vb - 1; di < ve; ++di) {           // d over {v} + out(v)
            const int d = di < vb ? v : s_dst[di];
            const int q = s_slot[d];
            if (q >= 0 && q != p) atomicOr(row + ((q - k0) >> 5), 1u << ((q - k0) & 31));
          }
        }
      }
    }
  } else {
    for (int u = n0 + tid; u < n1; u += kBlock) {
      const int ub = a.out_ptr[u], ue = a.out_ptr[u + 1];
      for (int ci = ub - 1; ci < ue; ++ci) {               // c over {u} + out(u)
        const int c = ci < ub ? u : a.out_dst[ci];
        const int p = a.slot[c];
        if (p < 0) continue;
        uint32_t* row = s_bm + (p - k0) * W;
        for (int vi = ub - 1; vi < ue; ++vi) {             // v over {u} + out(u)
          const int v = vi < ub ? u : a.out_dst[vi];
          const int vb = a.out_ptr[v], ve = a.out_ptr[v + 1];
          for (int di = vb - 1; di < ve; ++di) {           // d over {v} + out(v)
            const int d = di < vb ? v : a.out_dst[di];
            const int q = a.slot[d];
            if (q >= 0 && q != p) atomicOr(row + ((q - k0) >> 5), 1u << ((q - k0) & 31));
          }
        }
      }
    }
  }
  __syncthreads();
  // The TRANSPOSE, in LDS behind the matrix: every row's owner sets bit p of the columns q it points to (a row has a few bits).  In-
  // degrees are then popcounts of its rows, and the fill kernel lists a column's sources by walking set bits -- both used to test
  // bit r of ALL k_g rows, a loop of k_g LDS reads per thread (36 + 50 us for 32 four-qubit circuits: a fifth of their train step).
  uint32_t* s_t = s_bm + kg * W;
  for (int i = tid; i < kg * W; i += kBlock) s_t[i] = 0u;
  __syncthreads();
  for (int r = tid; r < kg; r += kBlock)
    for (int w = 0; w < W; ++w) {
      uint32_t bits = s_bm[r * W + w];
      while (bits) {
        const int b = __ffs((int)bits) - 1;
        bits &= bits - 1;
        atomicOr(s_t + (w * 32 + b) * W + (r >> 5), 1u << (r & 31));
      }
    }
  __syncthreads();
  uint32_t* gb = a.bitmaps + (int64_t)g * a.kmax * W;
  uint32_t* gt = a.bitmaps_t + (int64_t)g * a.kmax * W;
  for (int i = tid; i < kg * W; i += kBlock) { gb[i] = s_bm[i]; gt[i] = s_t[i]; }
  for (int r = tid; r < kg; r += kBlock) {
    int od = 0, id = 0;
    for (int w = 0; w < W; ++w) { od += __popc(s_bm[r * W + w]); id += __popc(s_t[r * W + w]); }
    a.outdeg[k0 + r] = od;
    a.indeg[k0 + r] = id;
  }
}

__global__ __launch_bounds__(kBlock) void coarsen_dense_fill_kernel(const DenseArgs a, const int32_t* __restrict__ in_ptr_new,
                                                                    const int32_t* __restrict__ out_ptr_new,
                                                                    int32_t* __restrict__ in_src_new,
                                                                    int32_t* __restrict__ out_dst_new,
                                                                    int32_t* __restrict__ out_eid_new) {
  extern __shared__ uint32_t s_bm[];                       // the matrix [kg][W], its transpose [kg][W], prefix popcounts [kg][W]
  const int g = blockIdx.x, tid = threadIdx.x;
  const int k0 = a.new_gptr[g], kg = a.new_gptr[g + 1] - k0;
  const int W = a.W;
  const uint32_t* gb = a.bitmaps + (int64_t)g * a.kmax * W;
  const uint32_t* gt = a.bitmaps_t + (int64_t)g * a.kmax * W;
  uint32_t* s_t = s_bm + kg * W;
  uint32_t* s_pre = s_t + kg * W;                          // s_pre[r][w] = set bits of row r in the words before w
  __shared__ int s_optr[kDenseMaxK];                       // the rows' places in the out-CSR (read once per EDGE below)
  for (int i = tid; i < kg * W; i += kBlock) { s_bm[i] = gb[i]; s_t[i] = gt[i]; }
  for (int i = tid; i < kg; i += kBlock) s_optr[i] = out_ptr_new[k0 + i];
  __syncthreads();
  for (int r = tid; r < kg; r += kBlock) {
    uint32_t run = 0;
    for (int w = 0; w < W; ++w) { s_pre[r * W + w] = run; run += (uint32_t)__popc(s_bm[r * W + w]); }
  }
  __syncthreads();
  for (int r = tid; r < kg; r += kBlock) {
    int pos = s_optr[r];                                   // row r: destinations in ascending order
    for (int w = 0; w < W; ++w) {
      uint32_t bits = s_bm[r * W + w];
      while (bits) {
        const int b = __ffs((int)bits) - 1;
        bits &= bits - 1;
        out_dst_new[pos++] = k0 + w * 32 + b;
      }
    }
    int ipos = in_ptr_new[k0 + r];                         // column r: sources in ascending order = the set bits of row r of the transpose
    const int wq = r >> 5;
    const uint32_t below = (1u << (r & 31)) - 1u;
    for (int w = 0; w < W; ++w) {
      uint32_t bits = s_t[r * W + w];
      while (bits) {
        const int b = __ffs((int)bits) - 1;
        bits &= bits - 1;
        const int rr = w * 32 + b;
        in_src_new[ipos] = k0 + rr;
        // position of (rr -> r) inside row rr of the out-CSR
        const int rank = (int)s_pre[rr * W + wq] + __popc(s_bm[rr * W + wq] & below);
        out_eid_new[s_optr[rr] + rank] = ipos;
        ++ipos;
      }
    }
  }
}

// ------------------------------------------------------ coarsened connectivity, large graphs: one WAVE per cluster
// 100-qubit circuits pool to thousands of clusters per graph and their second pooling runs on graphs whose hub clusters
// have hundreds of neighbours: the two-hop path above then enumerates ~50 candidates per distinct pair (147 M hop-1 and
// 234 M hop-2 keys for eight circuits), sorts them as 64-bit keys (65 GB of buffers at 64 circuits) and reads four sizes
// back.  Here a wave owns cluster p and keeps BITSETS in LDS: X (the graph's nodes, n_g bits) and Y, Z (the graph's
// clusters, k_g bits each):
//     X  = N+[N-[c_p]],  Y = { slot[w] : w in N+[X]  }      -- the clusters p points to   (row p of the pooled adjacency)
//     X' = N-[N-[c_p]],  Z = { slot[w] : w in N+[X'] }      -- the clusters pointing to p (row p of its transpose)
// Duplicates collapse in the LDS atomicOr, nothing is sorted, and a bitset is enumerated in ascending order -- which is the
// order the CSR arrays want.  Both rows go to global bit matrices [K][Wk] with plain coalesced stores (the transpose used to
// be filled with one global atomicOr per distinct edge into a zeroed matrix: 9.7 M random read-modify-writes for 64 circuits,
// half of the kernel's time, plus the memset and a popcount pass).  Popcounts -> device scans -> the CSR pointers; a second
// kernel lists rows / columns and links them (out_eid).  Graphs up to ~65 k nodes (64 KB of LDS per workgroup of four waves).
struct RowsArgs {
  const int32_t* in_ptr; const int32_t* in_src; const int32_t* out_ptr; const int32_t* out_dst;
  const int32_t* gptr; const int32_t* new_gptr; const int32_t* perm; const int32_t* slot;
  int B; int64_t K;
  int Wn, Wk;                    // words of the node / cluster bitsets (batch maxima)
  uint32_t* bm;                  // [K][Wk]: row p = clusters q (local index) with p -> q
  uint64_t* bmT;                 // [K][Wk]: row q = sources p in the low word; high word = set bits of the row before this word
  int32_t* outdeg; int32_t* indeg;   // [K + 1]
};

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// exclusive prefix over the wave of a per-lane count
__device__ __forceinline__ int wave_excl_scan(int v, int lane) {
  int s = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(s, o);
    if (lane >= o) s += t;
  }
  return s - v;
}

// LDS traffic of ONE wave on its own region: the LDS pipeline serves a wave's instructions in order, so a fence that keeps the
// compiler from moving accesses across it (and waits for the returns) is all the synchronisation the phases below need --
// no workgroup barrier, the four waves of a workgroup never wait for each other's clusters.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int64_t wave_excl_scan64(int64_t v, int lane) {
  int64_t s = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int64_t t = __shfl_up(s, o);
    if (lane >= o) s += t;
  }
  return s - v;
}

// The same for a wave that has global loads in flight it does not want to wait for (a software pipeline): LDS traffic only.
__device__ __forceinline__ void wave_lds_only_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

constexpr int kRowsLight = 16;   // degree up to which a lane walks a node's edges alone

// Every lane brings one node (or none); f(w) is called for the node itself and for every w in its neighbourhood in the CSR
// (ptr, idx).  Nodes with few edges are walked by their lane; the others one after the other by the whole wave (coalesced reads
// of their adjacency) -- a hub cluster's reach is a few nodes with hundreds of edges each, which one lane would walk for
// hundreds of dependent round trips while 63 wait.
template <typename F>
__device__ __forceinline__ void visit_closed(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, bool has, int node,
                                             int lane, F f) {
  int eb = 0, ee = 0;
  if (has) { eb = ptr[node]; ee = ptr[node + 1]; f(node); }
  const bool heavy = has && ee - eb > kRowsLight;
  if (has && !heavy)
    for (int e = eb; e < ee; ++e) f(idx[e]);
  unsigned long long todo = __ballot(heavy);
  while (todo) {
    const int owner = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int b = __shfl(eb, owner), e = __shfl(ee, owner);
    for (int i = b + lane; i < e; i += 64) f(idx[i]);
  }
}

// S = { slot[w] >= 0 : w in N+[v], v in X } without `self` (cluster bits, local to the graph); X: Wn words of node bits
__device__ __forceinline__ void clusters_reached(const RowsArgs& a, const uint32_t* X, int Wn, uint32_t* S, int n0, int k0, int self,
                                                 int lane) {
  for (int w0 = 0; w0 < Wn; w0 += 64) {
    const int wi = w0 + lane;
    uint32_t bits = wi < Wn ? X[wi] : 0u;
    while (__ballot(bits != 0u)) {                   // every lane offers its next node, if it has one left
      const bool has = bits != 0u;
      int v = 0;
      if (has) { const int b = __ffs((int)bits) - 1; bits &= bits - 1; v = n0 + wi * 32 + b; }
      visit_closed(a.out_ptr, a.out_dst, has, v, lane, [&](int w) {
        const int q = a.slot[w];
        if (q >= 0 && q != self) atomicOr(&S[(q - k0) >> 5], 1u << ((q - k0) & 31));
      });
    }
  }
}

__global__ __launch_bounds__(kBlock) void coarsen_rows_kernel(const RowsArgs a) {
  extern __shared__ uint32_t s_bits[];              // per wave: X [Wn], Y [Wk], Z [Wk]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t* X = s_bits + (size_t)wid * (a.Wn + 2 * a.Wk);
  uint32_t* Y = X + a.Wn;
  uint32_t* Z = Y + a.Wk;
  const int64_t p = (int64_t)blockIdx.x * 4 + wid;
  if (p >= a.K) return;                              // wave-uniform
  const int c = a.perm[p];
  int lo = 0, hi = a.B;                              // graph of cluster p: largest g with new_gptr[g] <= p
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int64_t)a.new_gptr[mid] <= p) lo = mid; else hi = mid; }
  const int n0 = a.gptr[lo], k0 = a.new_gptr[lo];
  // words of THIS graph's bitsets (the LDS regions and the rows of the bit matrices are laid out for the batch's largest
  // graph; a batch mixes 2 k- and 20 k-node circuits, and every loop below runs over words, set or not)
  const int Wn = (a.gptr[lo + 1] - n0 + 31) >> 5, Wk = (a.new_gptr[lo + 1] - k0 + 31) >> 5;
  for (int i = lane; i < Wn; i += 64) X[i] = 0u;
  for (int i = lane; i < Wk; i += 64) { Y[i] = 0u; Z[i] = 0u; }
  wave_lds_sync();
  const int ib = a.in_ptr[c], ie = a.in_ptr[c + 1];
  // X = N+[N-[c]] (node bits, local to the graph)
  for (int i0 = ib - 1; i0 < ie; i0 += 64) {         // index ib - 1 stands for c itself
    const int i = i0 + lane;
    const bool has = i < ie;
    const int u = has ? (i < ib ? c : a.in_src[i]) : 0;
    visit_closed(a.out_ptr, a.out_dst, has, u, lane, [&](int v) { atomicOr(&X[(v - n0) >> 5], 1u << ((v - n0) & 31)); });
  }
  wave_lds_sync();
  clusters_reached(a, X, Wn, Y, n0, k0, (int)p, lane);
  wave_lds_sync();
  // X = N-[N-[c]]
  for (int i = lane; i < Wn; i += 64) X[i] = 0u;
  wave_lds_sync();
  for (int i0 = ib - 1; i0 < ie; i0 += 64) {
    const int i = i0 + lane;
    const bool has = i < ie;
    const int u = has ? (i < ib ? c : a.in_src[i]) : 0;
    visit_closed(a.in_ptr, a.in_src, has, u, lane, [&](int v) { atomicOr(&X[(v - n0) >> 5], 1u << ((v - n0) & 31)); });
  }
  wave_lds_sync();
  clusters_reached(a, X, Wn, Z, n0, k0, (int)p, lane);
  wave_lds_sync();
  int run_out = 0, run_in = 0;                         // set bits in the words before the current 64
  for (int w0 = 0; w0 < Wk; w0 += 64) {                // words past Wk of row p stay unwritten: nobody reads them
    const int wi = w0 + lane;
    const uint32_t yb = wi < Wk ? Y[wi] : 0u, zb = wi < Wk ? Z[wi] : 0u;
    const int cz = __popc(zb);
    const int ex = wave_excl_scan(cz, lane);
    if (wi < Wk) {
      a.bm[p * a.Wk + wi] = yb;
      a.bmT[p * a.Wk + wi] = (uint64_t)zb | ((uint64_t)(uint32_t)(run_in + ex) << 32);   // rank of the word's first bit inside the row
    }
    run_in += __shfl(ex + cz, 63);
    run_out += wave_sum(__popc(yb));
  }
  if (lane == 0) { a.outdeg[p] = run_out; a.indeg[p] = run_in; }
}

__global__ __launch_bounds__(kBlock) void coarsen_rows_fill_kernel(const RowsArgs a, const int32_t* __restrict__ in_ptr_new,
                                                                   const int32_t* __restrict__ out_ptr_new,
                                                                   int32_t* __restrict__ in_src_new,
                                                                   int32_t* __restrict__ out_dst_new,
                                                                   int32_t* __restrict__ out_eid_new) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.K) return;
  int lo = 0, hi = a.B;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int64_t)a.new_gptr[mid] <= r) lo = mid; else hi = mid; }
  const int k0 = a.new_gptr[lo];
  const int rl = (int)(r - k0);
  const int Wk = (a.new_gptr[lo + 1] - k0 + 31) >> 5;   // words of this graph's rows (the count pass wrote no more)
  // row r of the out-CSR: destinations ascending; each entry also learns where its twin lives in the in-CSR -- the rank of r in
  // row q of the transpose (bits before word wq: the high half of that word; bits below r inside it).  All stores of this kernel are
  // consecutive per row; what is random is one 8-byte and one 4-byte read per edge.
  const int wq = rl >> 5;
  const uint32_t below = (1u << (rl & 31)) - 1u;
  int base = out_ptr_new[r];
  for (int w0 = 0; w0 < Wk; w0 += 64) {
    const int wi = w0 + lane;
    uint32_t bits = wi < Wk ? a.bm[r * a.Wk + wi] : 0u;
    const int n = __popc(bits);
    const int ex = wave_excl_scan(n, lane);
    int pos = base + ex;
    while (bits) {
      const int b = __ffs((int)bits) - 1;
      bits &= bits - 1;
      const int64_t q = (int64_t)k0 + wi * 32 + b;
      const uint64_t t = a.bmT[q * a.Wk + wq];
      out_dst_new[pos] = (int32_t)q;
      out_eid_new[pos] = in_ptr_new[q] + (int)(t >> 32) + __popc((uint32_t)t & below);
      ++pos;
    }
    base += __shfl(ex + n, 63);
  }
  // row r of the in-CSR: sources ascending
  base = in_ptr_new[r];
  for (int w0 = 0; w0 < Wk; w0 += 64) {
    const int wi = w0 + lane;
    uint32_t bits = wi < Wk ? (uint32_t)a.bmT[r * a.Wk + wi] : 0u;
    const int n = __popc(bits);
    const int ex = wave_excl_scan(n, lane);
    int pos = base + ex;
    while (bits) {
      const int b = __ffs((int)bits) - 1;
      bits &= bits - 1;
      in_src_new[pos++] = k0 + wi * 32 + b;
    }
    base += __shfl(ex + n, 63);
  }
}

// ------------------------------------------------------ coarsened connectivity, large graphs: SORTED LISTS, one walk (round 4)
// The bit-matrix form above writes two dense [K][k_g / 32] matrices (0.93 GB for 64 100-qubit circuits, nearly all zeros) and
// reads them back (1.36 GB) to list 9.7 M edges, and every cluster's wave walks three hops of dependent gathers (in_src ->
// out_ptr -> out_dst -> slot: 112 M scattered cache accesses per 64 circuits, 72 % of its wave cycles parked on them).
// This form splits the three hops into per-NODE lists that are built once, so that a cluster only ORs a few contiguous lists
// into its LDS bitset -- and touches nothing dense in global memory:
//   C(v)  = { slot[w] >= 0 : w in N+[v] }                    the clusters node v belongs to            (<= 1 + outdeg v entries)
//   R(u)  = concat of C(v), v in N+[u];  R'(u) likewise over N-[u]  (closed neighbourhoods; duplicates stay: a bitset removes them)
//   row p       = ( union of R(u),  u in N-[c_p] ) \ {p};   row p of the transpose = ( union of R'(u), u in N-[c_p] ) \ {p}
// A hub's list (a barrier: ~300 entries) is formed once and read, coalesced, by each of the ~150 clusters that contain the hub,
// instead of being re-walked by every one of them.  Places: C(v) at out_ptr[v] + v; R / R' by exclusive scans of the structural
// sizes h_out / h_in; a row's sorted list in a scratch by scans of per-row bounds (sums of h over N-[c_p], clamped to k_g - 1).
//   walk  a PERSISTENT wave per cluster: two LDS bitsets (k_g bits each), read out in ascending order by lanes that own runs of
//         words (ONE wave scan of packed counts per cluster) and cleared on the way; degrees -> device scans -> CSR pointers;
//   copy  a 16-lane group per row moves its lists to their final place and notes every out-entry's row;
//   link  a thread per out-entry p -> q finds its twin in q's in-row by binary search of p in q's sorted list (out_eid).
// Same arrays as the bit-matrix form and the two-hop path (tests/test_gpu_family_b.py).
struct ListsArgs {
  const int32_t* in_ptr; const int32_t* in_src; const int32_t* out_ptr; const int32_t* out_dst;
  const int32_t* gptr; const int32_t* new_gptr; const int32_t* perm; const int32_t* slot;
  int B; int64_t N, K;
  int Wk;                        // LDS words of a cluster bitset (batch maximum)
  const uint32_t* rinfo;         // [N][4]: place and length of R(u), place and length of R'(u)
  const int32_t* r_o; const int32_t* r_i; int64_t r_cap;
  const int64_t* off_o; const int64_t* off_i;        // [K + 1]: places of the rows in the scratch lists
  int32_t* tmp_o; int32_t* tmp_i; int64_t tmp_cap;
  int32_t* outdeg; int32_t* indeg;   // [K + 1]
  int32_t* overflow;             // != 0: a list left its storage (the caller's capacity was no bound)
};

constexpr int kReachClamp = 1 << 28;   // two-hop degree sums saturate here (a row bound is clamped to k_g - 1 anyway)

// Per lane: g(node) + sum of g over the node's neighbourhood in (ptr, idx); hubs are summed by the whole wave.  Every lane of
// the wave must call it (has = false for lanes without a node).
template <typename G>
__device__ __forceinline__ int64_t sum_closed(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, bool has, int node,
                                              int lane, G g) {
  int eb = 0, ee = 0;
  int64_t acc = 0;
  if (has) { eb = ptr[node]; ee = ptr[node + 1]; acc = g(node); }
  const bool heavy = has && ee - eb > kRowsLight;
  if (has && !heavy)
    for (int e = eb; e < ee; ++e) acc += g(idx[e]);
  unsigned long long todo = __ballot(heavy);
  while (todo) {
    const int owner = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int b = __shfl(eb, owner), e = __shfl(ee, owner);
    int64_t part = 0;
    for (int i = b + lane; i < e; i += 64) part += g(idx[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == owner) acc += part;
  }
  return acc;
}

// h_out[u] = sum over v in N+[u] of (1 + outdeg v) >= |R(u)|; h_in[u] the same over N-[u] >= |R'(u)| (the last hop is an out-hop on
// both sides); entry N is 0 (the scans' last element is then the total)
__global__ __launch_bounds__(kBlock) void coarsen_reach_kernel(const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_src,
                                                               const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_dst,
                                                               int64_t N, int64_t* __restrict__ h_out, int64_t* __restrict__ h_in) {
  const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool has = u < N;
  auto od1 = [&](int v) { return (int64_t)(1 + out_ptr[v + 1] - out_ptr[v]); };
  const int64_t ho = sum_closed(out_ptr, out_dst, has, (int)u, lane, od1);
  const int64_t hi = sum_closed(in_ptr, in_src, has, (int)u, lane, od1);
  if (has) {
    h_out[u] = min(ho, (int64_t)kReachClamp);
    h_in[u] = min(hi, (int64_t)kReachClamp);
  } else if (u == N) {
    h_out[N] = 0;
    h_in[N] = 0;
  }
}

// cap_o[p] >= the candidates of row p, cap_i[p] >= those of row p of the transpose: sums of h over N-[c_p]; entry K is 0
__global__ __launch_bounds__(kBlock) void coarsen_row_caps_kernel(const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_src,
                                                                  const int32_t* __restrict__ perm, const int32_t* __restrict__ new_gptr,
                                                                  int B, int64_t K, const int64_t* __restrict__ h_out,
                                                                  const int64_t* __restrict__ h_in, int64_t* __restrict__ cap_o,
                                                                  int64_t* __restrict__ cap_i, int32_t* __restrict__ zero_rows) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool has = p < K;
  if (has && zero_rows) zero_rows[p] = 0;            // (the coarsened structure's self-loop counts: none -- a fill launch of its own before)
  const int c = has ? perm[p] : 0;
  const int64_t so = sum_closed(in_ptr, in_src, has, c, lane, [&](int u) { return h_out[u]; });
  const int64_t si = sum_closed(in_ptr, in_src, has, c, lane, [&](int u) { return h_in[u]; });
  if (has) {
    cap_o[p] = so;                 // room for the row's CANDIDATES (duplicates and all): the sorted row is written over them
    cap_i[p] = si;
  } else if (p == K) {
    cap_o[K] = 0;
    cap_i[K] = 0;
  }
}

// C(v): the kept centres among N+[v] (v itself included), as cluster ids, at clist[out_ptr[v] + v ...]; ccnt[v] of them
__global__ __launch_bounds__(kBlock) void coarsen_clists_kernel(const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_dst,
                                                                const int32_t* __restrict__ slot, int64_t N, int32_t* __restrict__ clist,
                                                                int32_t* __restrict__ ccnt, int32_t* __restrict__ zero_a,
                                                                int32_t* __restrict__ zero_b, int32_t* __restrict__ zero_c) {
  const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool has = v < N;
  if (v == 0) { *zero_a = 0; *zero_b = 0; *zero_c = 0; }      // three words the LATER launches of the coarsening count into
  int eb = 0, ee = 0, n = 0;
  int64_t base = 0;
  if (has) {
    eb = out_ptr[v]; ee = out_ptr[v + 1];
    base = (int64_t)eb + v;
    const int q = slot[v];
    if (q >= 0) clist[base + n++] = q;
  }
  const bool heavy = has && ee - eb > kRowsLight;
  // PARALLEL EDGES are walked once (here and in the two passes below: the coarsened graph is a set of edges, and the lists are sorted by
  // the structure's build, so a repeated neighbour is the one before it).  A hundred parallel edges between two consecutive barriers
  // of a 100-qubit circuit otherwise make |C| = 101, |R| = 10 201 and 1 020 201 candidates for ONE cluster, walked by one wave: 2 ms
  // in either list pass whenever the scores keep such a barrier (scripts/coarse_degree_probe.py).
  if (has && !heavy) {
    int prev = -1;
    for (int e = eb; e < ee; ++e) {
      const int d = out_dst[e];
      if (d == prev) continue;
      prev = d;
      const int q = slot[d];
      if (q >= 0) clist[base + n++] = q;
    }
  }
  unsigned long long todo = __ballot(heavy);
  while (todo) {                                       // a hub's out-list: the whole wave, kept entries compacted by ballot
    const int owner = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int b = __shfl(eb, owner), e = __shfl(ee, owner);
    const int64_t ob = __shfl(base, owner);
    int on = __shfl(n, owner);
    for (int i0 = b; i0 < e; i0 += 64) {
      const int i = i0 + lane;
      const int d = i < e ? out_dst[i] : -1;
      const bool again = i < e && i > b && out_dst[i - 1] == d;
      const int q = (i < e && !again) ? slot[d] : -1;
      const unsigned long long keep = __ballot(q >= 0);
      if (q >= 0) clist[ob + on + __popcll(keep & ((1ull << lane) - 1ull))] = q;
      on += __popcll(keep);
    }
    if (lane == owner) n = on;
  }
  if (has) ccnt[v] = n;
}

// R(u) (SIDE_IN = false: over N+[u]) or R'(u) (true: over N-[u]): the C lists of the closed neighbourhood, one after the other
template <bool SIDE_IN>
__global__ __launch_bounds__(kBlock) void coarsen_rlists_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                                const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ clist,
                                                                const int32_t* __restrict__ ccnt, int64_t N, const int64_t* __restrict__ roff,
                                                                int32_t* __restrict__ rlist, int64_t r_cap, uint32_t* __restrict__ rinfo,
                                                                int32_t* __restrict__ overflow) {
  const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool has = u < N;
  int eb = 0, ee = 0, n = 0;
  int64_t base = 0;
  bool fits = true;
  auto append = [&](int v, int64_t at) {              // C(v) to rlist[at ...]; returns its length
    const int m = ccnt[v];
    const int32_t* __restrict__ src = clist + ((int64_t)out_ptr[v] + v);
    for (int j = 0; j < m; ++j) rlist[at + j] = src[j];
    return m;
  };
  if (has) {
    eb = ptr[u]; ee = ptr[u + 1];
    base = roff[u];
    fits = roff[u + 1] <= r_cap;                       // the bound h(u) of this list lies inside the storage
    if (fits) n = append((int)u, base);
  }
  const bool heavy = has && fits && ee - eb > kRowsLight;
  if (has && fits && !heavy) {
    int prev = -1;
    for (int e = eb; e < ee; ++e) {
      const int v = idx[e];
      if (v == prev) continue;                         // a parallel edge
      prev = v;
      n += append(v, base + n);
    }
  }
  unsigned long long todo = __ballot(heavy);
  while (todo) {                                       // a hub: a lane per neighbour, places by a wave scan of the lists' lengths
    const int owner = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int b = __shfl(eb, owner), e = __shfl(ee, owner);
    const int64_t ob = __shfl(base, owner);
    int on = __shfl(n, owner);
    for (int i0 = b; i0 < e; i0 += 64) {
      const int i = i0 + lane;
      int v = i < e ? idx[i] : -1;
      if (i < e && i > b && idx[i - 1] == v) v = -1;    // a parallel edge
      const int m = v >= 0 ? ccnt[v] : 0;
      const int ex = wave_excl_scan(m, lane);
      if (v >= 0) append(v, ob + on + ex);
      on += __shfl(ex + m, 63);
    }
    if (lane == owner) n = on;
  }
  if (has) {
    rinfo[u * 4 + (SIDE_IN ? 2 : 0)] = (uint32_t)base;          // r_cap < 2^32 (checked by the host): places fit 32 bits
    rinfo[u * 4 + (SIDE_IN ? 3 : 1)] = (uint32_t)n;
    if (!fits) atomicOr(overflow, 1);
  }
}

// Per-node record: place and length of R(u), place and length of R'(u)
struct RInfo { uint32_t off_o, cnt_o, off_i, cnt_i; };
// Per-cluster record of the gather pass: place and length of the row's candidates, the same for the transposed row
struct CInfo { uint32_t off_o, cnt_o, off_i, cnt_i; };

// GATHER: the candidates of a cluster's two rows -- the lists R(u) / R'(u) of u in {c} + N-[c], one after the other, duplicates and
// the cluster itself included -- copied to the row's place in the scratch.  A THREAD per cluster: the pass is a chain of dependent
// gathers (perm -> in_ptr -> in_src -> record -> list), which a wave per cluster paid in full for every cluster (12 us each);
// here 363 k of them are in flight at once.  Lists longer than kRowsLight (a hub's) and clusters with many members (a hub's own)
// are copied by the whole wave.
__device__ __forceinline__ void gather_lists(const int32_t* __restrict__ rlist, uint32_t rb, uint32_t rn, bool has, int32_t* __restrict__ out,
                                             int64_t at, int64_t cap, int lane) {
  const bool light = rn <= (uint32_t)kRowsLight;
  if (has && light && at + rn <= cap) {
    int qs[kRowsLight];
    // (round 6: these sixteen loads made unconditional -- a place past the list's end reading its first entry -- cost 229 against 200 us:
    // most lists hold two or three entries)
#pragma unroll
    for (int j = 0; j < kRowsLight; ++j) qs[j] = j < (int)rn ? rlist[rb + j] : 0;
#pragma unroll
    for (int j = 0; j < kRowsLight; ++j) if (j < (int)rn) out[at + j] = qs[j];
  }
  // long lists: by all the lanes that are here together (the caller may be inside a loop only some lanes of the wave still run:
  // a lane's share is its rank among them)
  const unsigned long long here = __ballot(true);
  const int rank = __popcll(here & ((1ull << lane) - 1ull)), width = __popcll(here);
  unsigned long long todo = __ballot(has && !light);
  while (todo) {
    // FOUR lists a trip: their first pieces are loaded together, then stored (one list after the other every copy waited for its own
    // loads: the ~250 lists of a barrier's cluster were 250 round trips of one wave, 147 of the pass's 232 us on the mixed corpus)
    uint32_t b[4], m[4];
    int64_t o[4];
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int owner = todo ? __ffsll((long long)todo) - 1 : 0;
      const bool live = todo != 0ull;
      todo &= todo - 1;                                    // (0 stays 0)
      b[k] = __shfl(rb, owner); m[k] = live ? (uint32_t)__shfl((int)rn, owner) : 0u;
      o[k] = __shfl(at, owner);
      if (o[k] + m[k] > cap) m[k] = 0u;                    // a list that would leave its storage is dropped whole
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (uint32_t)rank < m[k] ? rlist[b[k] + rank] : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((uint32_t)rank < m[k]) out[o[k] + rank] = v[k];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      for (uint32_t j = rank + width; j < m[k]; j += width) out[o[k] + j] = rlist[b[k] + j];
  }
}

__global__ __launch_bounds__(kBlock) void coarsen_gather_kernel(const ListsArgs a, uint32_t* __restrict__ cinfo, int2* __restrict__ cgraph) {
  // Lane l of wave w takes cluster l R + w (R = waves in the grid): clusters that are neighbours in the top-k's order score alike, so
  // the clusters around a circuit's barriers -- a hundred members with lists of hundreds each, copied by their whole wave one after
  // the other -- sat in a handful of waves while the chip idled (2.1 ms for 110 k clusters when the scores pick the barriers'
  // neighbourhoods; the eval-mode forward of 20 100-qubit circuits).  Spread out, a wave meets one or two of them.
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = (int64_t)gridDim.x * (kBlock / kWave), wave = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const int64_t p = (int64_t)lane * n_waves + wave;
  const bool has = p < a.K;
  const RInfo* __restrict__ rinfo = reinterpret_cast<const RInfo*>(a.rinfo);
  int c = 0, ib = 0, ie = 0;
  int64_t bo = 0, bi = 0;
  if (has) { c = a.perm[p]; ib = a.in_ptr[c]; ie = a.in_ptr[c + 1]; bo = a.off_o[p]; bi = a.off_i[p]; }
  int64_t no = 0, ni = 0;
  const bool wide = has && ie - ib > kRowsLight;          // many members (the cluster of a hub): its wave walks them together
  if (has && !wide)
    for (int i = ib - 1, prev = -1; i < ie; ++i) {        // index ib - 1 stands for c itself
      const int u = i < ib ? c : a.in_src[i];
      if (i >= ib && u == prev) continue;                 // a parallel edge: the member is in the cluster once
      prev = i < ib ? -1 : u;
      const RInfo r = rinfo[u];
      gather_lists(a.r_o, r.off_o, r.cnt_o, true, a.tmp_o, bo + no, a.tmp_cap, lane);
      gather_lists(a.r_i, r.off_i, r.cnt_i, true, a.tmp_i, bi + ni, a.tmp_cap, lane);
      no += r.cnt_o; ni += r.cnt_i;
    }
  unsigned long long todo = __ballot(wide);
  while (todo) {
    const int owner = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int oc = __shfl(c, owner), ob = __shfl(ib, owner), oe = __shfl(ie, owner);
    int64_t wo = __shfl(bo, owner), wi = __shfl(bi, owner);
    int64_t so = 0, si = 0;
    for (int i0 = ob - 1; i0 < oe; i0 += 64) {
      const int i = i0 + lane;
      const int u = i < oe ? (i < ob ? oc : a.in_src[i]) : 0;
      const bool mine = i < oe && !(i > ob && a.in_src[i - 1] == u);      // (not a parallel edge)
      RInfo r{0u, 0u, 0u, 0u};
      if (mine) r = rinfo[u];
      // places of this lane's two lists: wave scans of the lengths
      const int64_t eo = wave_excl_scan64((int64_t)r.cnt_o, lane), ei2 = wave_excl_scan64((int64_t)r.cnt_i, lane);
      gather_lists(a.r_o, r.off_o, r.cnt_o, mine, a.tmp_o, wo + so + eo, a.tmp_cap, lane);
      gather_lists(a.r_i, r.off_i, r.cnt_i, mine, a.tmp_i, wi + si + ei2, a.tmp_cap, lane);
      so += __shfl(eo + (int64_t)r.cnt_o, 63);
      si += __shfl(ei2 + (int64_t)r.cnt_i, 63);
    }
    if (lane == owner) { no = so; ni = si; }
  }
  if (has) {
    const bool fits = bo + no <= a.tmp_cap && bi + ni <= a.tmp_cap && no < (1ll << 32) && ni < (1ll << 32);
    cinfo[p * 4 + 0] = (uint32_t)bo; cinfo[p * 4 + 1] = fits ? (uint32_t)no : 0u;
    cinfo[p * 4 + 2] = (uint32_t)bi; cinfo[p * 4 + 3] = fits ? (uint32_t)ni : 0u;
    const int g = graph_at(a.new_gptr, a.B, p);       // the cluster's graph: its first cluster and the words of its bitsets
    const int k0 = a.new_gptr[g];
    cgraph[p] = make_int2(k0, (a.new_gptr[g + 1] - k0 + 31) >> 5);
    if (!fits) atomicOr(a.overflow, 1);
  }
}

// UNIQUE: a persistent wave per cluster turns the two candidate lists into sorted lists without duplicates, IN PLACE: every
// candidate sets a bit of an LDS bitset (k_g bits), the bitsets are read out in ascending order by lanes that own runs of words
// (ONE wave scan of packed counts per cluster) and cleared on the way.  The pass streams: a cluster's record and its first 64
// candidates per side are fetched while the previous cluster is in the LDS (the records two clusters ahead), so the wave never
// waits on a chain of gathers.
struct UniqueMeta { CInfo c; int k0, Wk; };

__global__ __launch_bounds__(kBlock) void coarsen_unique_kernel(const ListsArgs a, const uint32_t* __restrict__ cinfo_raw,
                                                                const int2* __restrict__ cgraph) {
  extern __shared__ uint32_t s_bits[];              // per wave: Y [Wk], Z [Wk]
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t* Y = s_bits + (size_t)wid * 2 * a.Wk;
  uint32_t* Z = Y + a.Wk;
  for (int i = lane; i < 2 * a.Wk; i += 64) Y[i] = 0u;     // once per wave: every cluster leaves its region as it found it
  wave_lds_sync();
  const CInfo* __restrict__ cinfo = reinterpret_cast<const CInfo*>(cinfo_raw);
  const int64_t stride = (int64_t)gridDim.x * 4, lim = a.K;
  auto first_of = [&](const UniqueMeta& m, int& qo, int& qi) {       // the first 64 candidates of either side, -1 past the end
    qo = lane < (int)m.c.cnt_o ? a.tmp_o[(int64_t)m.c.off_o + lane] : -1;
    qi = lane < (int)m.c.cnt_i ? a.tmp_i[(int64_t)m.c.off_i + lane] : -1;
  };
  // The wave's clusters are first + j stride.  Their records come 64 at a time, ONE per lane (a vector load: a scalar load per
  // cluster -- and a search for its graph -- stood in every iteration's way: scalar loads return out of order, so every use waits
  // for all of them; measured 190 of 438 us for the search), and reach the scalar side by readlane when their turn comes.
  // What is left (330 us on 64 100-qubit circuits: 170 for the five light clusters of six, 45 for the marks and 115 for the
  // read-out of the heavy ones) is instruction issue, not latency: static chunks, deeper prefetch and batched loads moved nothing.
  for (int64_t first = (int64_t)blockIdx.x * 4 + wid; first < lim; first += stride * 64) {
    const int64_t mine = first + (int64_t)lane * stride;
    CInfo rc{0u, 0u, 0u, 0u};
    int2 kg = make_int2(0, 0);
    if (mine < lim) { rc = cinfo[mine]; kg = cgraph[mine]; }
    const int nb = (int)min((int64_t)64, (lim - first + stride - 1) / stride);      // clusters of this batch (wave-uniform)
    auto meta_of = [&](int j) {
      UniqueMeta m;
      m.c.off_o = (uint32_t)__builtin_amdgcn_readlane((int)rc.off_o, j); m.c.cnt_o = (uint32_t)__builtin_amdgcn_readlane((int)rc.cnt_o, j);
      m.c.off_i = (uint32_t)__builtin_amdgcn_readlane((int)rc.off_i, j); m.c.cnt_i = (uint32_t)__builtin_amdgcn_readlane((int)rc.cnt_i, j);
      m.k0 = __builtin_amdgcn_readlane(kg.x, j); m.Wk = __builtin_amdgcn_readlane(kg.y, j);
      return m;
    };
    // Clusters of at most 16 candidates on either side (most of the coarsened graph's rows: median degree 3) go FOUR at a time, one
    // per 16-lane row: a 16-wide bitonic network (10 exchange steps, 7 of them quad permutes) instead of the 64-wide one (21 steps of
    // ds_bpermute per side) that a wave ran for each of them alone -- 170 of this kernel's 410 us were those clusters.  Any four of
    // the batch's tiny clusters share a pass (their bits of `tiny`, lowest first); the others follow one by one.
    const unsigned long long valid = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
    const unsigned long long tiny = __ballot(mine < lim && rc.cnt_o <= 16u && rc.cnt_i <= 16u) & valid;
    for (unsigned long long t = tiny; t;) {
      int take[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        take[r] = t ? __ffsll((long long)t) - 1 : -1;
        if (t) t &= t - 1;
      }
      const int row = lane >> 4, sub = lane & 15;
      const int src = row == 0 ? take[0] : (row == 1 ? take[1] : (row == 2 ? take[2] : take[3]));
      const bool on = src >= 0;
      const int from = on ? src : 0;
      // (every shuffle OUTSIDE a condition on `on`: a ds_bpermute under a divergent branch reads nothing from the lanes the branch
      // switched off -- the first form of this pass had them inside one and emptied the clusters of rows whose source lane idled)
      const uint32_t co = (uint32_t)__shfl((int)rc.cnt_o, from), ci = (uint32_t)__shfl((int)rc.cnt_i, from);
      const uint32_t off_o = (uint32_t)__shfl((int)rc.off_o, from), cnt_o = on ? co : 0u;
      const uint32_t off_i = (uint32_t)__shfl((int)rc.off_i, from), cnt_i = on ? ci : 0u;
      const int64_t pr = first + (int64_t)from * stride;
      const int self = (int)pr;
      int vo = sub < (int)cnt_o ? a.tmp_o[(int64_t)off_o + sub] : -1, vi = sub < (int)cnt_i ? a.tmp_i[(int64_t)off_i + sub] : -1;
      vo = (vo >= 0 && vo != self) ? vo : INT32_MAX;
      vi = (vi >= 0 && vi != self) ? vi : INT32_MAX;
#pragma unroll
      for (int k = 2; k <= 16; k <<= 1) {
#pragma unroll
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
          int po2, pi2;
          if (jj == 1) { po2 = __builtin_amdgcn_update_dpp(0, vo, 0xB1, 0xF, 0xF, true); pi2 = __builtin_amdgcn_update_dpp(0, vi, 0xB1, 0xF, 0xF, true); }
          else if (jj == 2) { po2 = __builtin_amdgcn_update_dpp(0, vo, 0x4E, 0xF, 0xF, true); pi2 = __builtin_amdgcn_update_dpp(0, vi, 0x4E, 0xF, 0xF, true); }
          else { po2 = __shfl_xor(vo, jj); pi2 = __shfl_xor(vi, jj); }
          const bool take_min = ((sub & k) == 0) == ((sub & jj) == 0);
          vo = take_min ? min(vo, po2) : max(vo, po2);
          vi = take_min ? min(vi, pi2) : max(vi, pi2);
        }
      }
      const int lo = __shfl_up(vo, 1), li = __shfl_up(vi, 1);
      const bool ko = vo != INT32_MAX && (sub == 0 || vo != lo), ki = vi != INT32_MAX && (sub == 0 || vi != li);
      const unsigned mo = (unsigned)(__ballot(ko) >> (16 * row)) & 0xFFFFu, mi = (unsigned)(__ballot(ki) >> (16 * row)) & 0xFFFFu;
      const unsigned below = (1u << sub) - 1u;
      if (ko) a.tmp_o[(int64_t)off_o + __popc(mo & below)] = vo;
      if (ki) a.tmp_i[(int64_t)off_i + __popc(mi & below)] = vi;
      if (on && sub == 0) {
        a.outdeg[pr] = __popc(mo);
        a.indeg[pr] = __popc(mi);
      }
    }
    unsigned long long rest = valid & ~tiny;               // the other clusters, one per pass, the next one's candidates in flight
    UniqueMeta m0{{0u, 0u, 0u, 0u}, 0, 0};
    int qo0 = -1, qi0 = -1;
    if (rest) {
      m0 = meta_of(__ffsll((long long)rest) - 1);
      first_of(m0, qo0, qi0);
    }
    while (rest) {
      const int j = __ffsll((long long)rest) - 1;
      rest &= rest - 1;
      const int jn = rest ? __ffsll((long long)rest) - 1 : -1;
      const int64_t p = first + (int64_t)j * stride;
      UniqueMeta m1{{0u, 0u, 0u, 0u}, 0, 0};
      if (jn >= 0) m1 = meta_of(jn);
      int qo1, qi1;
      first_of(m1, qo1, qi1);                           // in flight while this cluster is sorted
      const int k0 = m0.k0, Wk = m0.Wk, self = (int)p;
      if (m0.c.cnt_o <= 64u && m0.c.cnt_i <= 64u) {
        // A LIGHT cluster (five of six): its candidates are already in registers, one per lane and side.  Sorted by a bitonic
        // network across the wave (21 exchange steps, both sides interleaved), duplicates dropped by a look at the left
        // neighbour, ranks from a ballot -- no LDS, no scan of k_g bits to find two dozen of them.
        int vo = (qo0 >= 0 && qo0 != self) ? qo0 : INT32_MAX, vi = (qi0 >= 0 && qi0 != self) ? qi0 : INT32_MAX;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
          for (int jj = k >> 1; jj > 0; jj >>= 1) {
            const int po2 = __shfl_xor(vo, jj), pi2 = __shfl_xor(vi, jj);
            const bool take_min = ((lane & k) == 0) == ((lane & jj) == 0);
            vo = take_min ? min(vo, po2) : max(vo, po2);
            vi = take_min ? min(vi, pi2) : max(vi, pi2);
          }
        }
        const int lo = __shfl_up(vo, 1), li = __shfl_up(vi, 1);
        const bool ko = vo != INT32_MAX && (lane == 0 || vo != lo), ki = vi != INT32_MAX && (lane == 0 || vi != li);
        const unsigned long long mo = __ballot(ko), mi = __ballot(ki), below = (1ull << lane) - 1ull;
        if (ko) a.tmp_o[(int64_t)m0.c.off_o + __popcll(mo & below)] = vo;       // over the candidates they came from (all in registers)
        if (ki) a.tmp_i[(int64_t)m0.c.off_i + __popcll(mi & below)] = vi;
        if (lane == 0) {
          a.outdeg[p] = __popcll(mo);
          a.indeg[p] = __popcll(mi);
        }
      } else {
        // A cluster with more than 64 candidates on a side (a hub's neighbourhood): every candidate sets a bit of an LDS bitset
        // (k_g bits), the bitsets are read out in ascending order by lanes that own runs of words (ONE wave scan of packed counts)
        // and cleared on the way.
        auto mark = [&](uint32_t* S, int q) { if (q >= 0 && q != self) atomicOr(&S[(q - k0) >> 5], 1u << ((q - k0) & 31)); };
        mark(Y, qo0);
        mark(Z, qi0);
        // the rest eight loads a side at a time, all issued before the first of them is marked (an LDS atomic behind every
        // load had the rounds wait for one another: ~2.5 us each)
        const uint32_t most = max(m0.c.cnt_o, m0.c.cnt_i);
        for (uint32_t t0 = 64; t0 < most; t0 += 512) {
          int ro[8], ri[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const uint32_t t = t0 + r * 64 + lane;
            ro[r] = t < m0.c.cnt_o ? a.tmp_o[(int64_t)m0.c.off_o + t] : -1;
            ri[r] = t < m0.c.cnt_i ? a.tmp_i[(int64_t)m0.c.off_i + t] : -1;
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) { mark(Y, ro[r]); mark(Z, ri[r]); }
        }
        wave_lds_only_sync();                             // the next cluster's candidates stay in flight
        const int cw = (Wk + 63) >> 6;                    // lane l owns words [l cw, (l + 1) cw); a row has fewer than 65 536 entries
        const int w_lo = min(Wk, lane * cw), w_hi = min(Wk, w_lo + cw);
        unsigned ny = 0, nz = 0;
        for (int w = w_lo; w < w_hi; ++w) { ny += __popc(Y[w]); nz += __popc(Z[w]); }
        const int packed = (int)(ny | (nz << 16));
        const int ex = wave_excl_scan(packed, lane);
        const unsigned tot = (unsigned)__shfl(ex + packed, 63);
        int64_t po = (int64_t)m0.c.off_o + (int)((unsigned)ex & 0xFFFFu), pi = (int64_t)m0.c.off_i + (int)((unsigned)ex >> 16);
        for (int w = w_lo; w < w_hi; ++w) {               // sorted, duplicate-free, over the candidates they came from (never longer)
          uint32_t yb = Y[w], zb = Z[w];
          Y[w] = 0u; Z[w] = 0u;
          const int first_q = k0 + w * 32;
          while (yb) { const int b = __ffs((int)yb) - 1; yb &= yb - 1; a.tmp_o[po++] = first_q + b; }
          while (zb) { const int b = __ffs((int)zb) - 1; zb &= zb - 1; a.tmp_i[pi++] = first_q + b; }
        }
        if (lane == 0) {
          a.outdeg[p] = (int)(tot & 0xFFFFu);
          a.indeg[p] = (int)(tot >> 16);
        }
        wave_lds_only_sync();
      }
      m0 = m1; qo0 = qo1; qi0 = qi1;
    }
  }
}

// a row's lists to their final place; out_row[e] = the row of out-entry e (the link pass is a thread per entry)
__global__ __launch_bounds__(kBlock) void coarsen_lists_copy_kernel(const ListsArgs a, const int32_t* __restrict__ in_ptr_new,
                                                                    const int32_t* __restrict__ out_ptr_new,
                                                                    int32_t* __restrict__ in_src_new, int32_t* __restrict__ out_dst_new,
                                                                    int32_t* __restrict__ out_row, int64_t edge_cap,
                                                                    int32_t* __restrict__ sticky) {
  const int64_t r = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x & (kGroup - 1);
  if (r >= a.K) return;
  // the caller's STICKY overflow flag takes pass 1's flag here (or_flag_kernel's job: a launch of its own before)
  if (r == 0 && l == 0 && sticky && *a.overflow != 0) atomicOr(sticky, *a.overflow);
  const int di = a.indeg[r], dout = a.outdeg[r];
  const int64_t bi = a.off_i[r], bo = a.off_o[r];
  const int64_t ip = in_ptr_new[r], op = out_ptr_new[r];
  // the first 32 entries of either list (most rows: 28 entries on average) as four loads issued together -- a place past a list's
  // end reads the list's first entry -- then the stores; longer lists finish in the loops below
  int vi[2], vo[2];
  const int64_t last = max(a.tmp_cap, (int64_t)1) - 1;      // (an empty list at the very end of the buffers)
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int i = l + c * kGroup;
    vi[c] = a.tmp_i[min(bi + (i < di ? i : 0), last)];
    vo[c] = a.tmp_o[min(bo + (i < dout ? i : 0), last)];
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int i = l + c * kGroup;
    if (i < di && ip + i < edge_cap) in_src_new[ip + i] = vi[c];
    if (i < dout && op + i < edge_cap) {
      out_dst_new[op + i] = vo[c];
      if (out_row) out_row[op + i] = (int32_t)r;
    }
  }
  for (int i = l + 2 * kGroup; i < di; i += kGroup)
    if (ip + i < edge_cap) in_src_new[ip + i] = a.tmp_i[bi + i];
  for (int i = l + 2 * kGroup; i < dout; i += kGroup)
    if (op + i < edge_cap) {
      out_dst_new[op + i] = a.tmp_o[bo + i];
      if (out_row) out_row[op + i] = (int32_t)r;
    }
}

// the twin of out-entry e = (r -> q) in q's in-row: the rank of r in q's sorted list of sources (r is in it: the two walks list
// the same edges)
__global__ __launch_bounds__(kBlock) void coarsen_lists_link_kernel(const int32_t* __restrict__ in_ptr_new, const int32_t* __restrict__ in_src_new,
                                                                    const int32_t* __restrict__ out_dst_new, const int32_t* __restrict__ out_row,
                                                                    const int32_t* __restrict__ edge_total, int64_t edge_cap,
                                                                    int32_t* __restrict__ out_eid_new) {
  const int64_t total = min(edge_cap, (int64_t)*edge_total);          // the grid is sized for the device, not for the capacity
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int q = out_dst_new[e], r = out_row[e];
    int lo = in_ptr_new[q], hi = in_ptr_new[q + 1];
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (in_src_new[mid] < r) lo = mid + 1; else hi = mid;
    }
    out_eid_new[e] = lo;
  }
}

// Exclusive scans of TWO short int32 arrays (the degree vectors of a dense coarsening: K + 1 <= kDualScanMax entries) by ONE
// workgroup in ONE launch -- rocprim's device scan is two launches per array, four of the eleven this coarsening made for a batch
// of 32 four-qubit circuits, where a launch costs what the whole kernel does.  Thread t scans its own run of ceil(n / 1024)
// consecutive entries, the runs' totals are scanned through LDS (wave scans, then the 16 wave totals).
constexpr int kDualScanThreads = 1024, kDualScanMax = 16 * kDualScanThreads;
__global__ __launch_bounds__(kDualScanThreads) void dual_scan_small_kernel(const int32_t* __restrict__ a0, const int32_t* __restrict__ a1,
                                                                           int n, int32_t* __restrict__ o0, int32_t* __restrict__ o1) {
  __shared__ int s_w[2][kDualScanThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n + kDualScanThreads - 1) / kDualScanThreads;        // <= 16
  const int lo = min(n, tid * per), hi = min(n, lo + per);
  int t0 = 0, t1 = 0;
  for (int i = lo; i < hi; ++i) { t0 += a0[i]; t1 += a1[i]; }
  int i0 = t0, i1 = t1;                                                   // inclusive scans inside the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int u0 = __shfl_up(i0, d), u1 = __shfl_up(i1, d);
    if (lane >= d) { i0 += u0; i1 += u1; }
  }
  if (lane == 63) { s_w[0][wave] = i0; s_w[1][wave] = i1; }
  __syncthreads();
  int b0 = 0, b1 = 0;
  for (int w = 0; w < wave; ++w) { b0 += s_w[0][w]; b1 += s_w[1][w]; }
  int r0 = b0 + i0 - t0, r1 = b1 + i1 - t1;                              // exclusive prefix of this thread's run
  for (int i = lo; i < hi; ++i) {
    const int v0 = a0[i], v1 = a1[i];
    o0[i] = r0; o1[i] = r1;
    r0 += v0; r1 += v1;
  }
}

static size_t dense_scan_bytes(int64_t K) {
  size_t temp = 0;
  (void)rocprim::exclusive_scan(nullptr, temp, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0, (size_t)(K + 1),
                                rocprim::plus<int32_t>(), (hipStream_t)0);
  return (temp + 255) / 256 * 256;
}

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_asap_compose_f32(const float* lin_w, const float* lin_b, const float* att_w, const float* att_b, const float* l1_w,
                                      const float* l1_b, const float* l2_w, const float* l3_w, const float* l3_b, int D, float* w_comp,
                                      float* b_comp, float* att_q, float* att_x, float* w3, float* b3, mlqem_stream_t stream) {
  begin_launches();
  if (D <= 0 || D > 4096) return MLQEM_ERR_BAD_ARG;
  if (!lin_w || !lin_b || !att_w || !att_b || !l1_w || !l1_b || !l2_w || !l3_w || !l3_b || !w_comp || !b_comp || !att_q || !att_x || !w3 || !b3)
    return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(asap_compose_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), lin_w, lin_b, att_w, att_b, l1_w, l1_b, l2_w, l3_w,
                     l3_b, D, w_comp, b_comp, att_q, att_x, w3, b3);
  return launch_status();
}

extern "C" int mlqem_asap_compose_bwd_f32(const float* g_w_comp, const float* g_att_b, const float* lin_w, const float* lin_b,
                                          const float* att_w, const float* g_att_x, int D, float* g_lin_w, float* g_lin_b, float* g_att_w,
                                          mlqem_stream_t stream) {
  begin_launches();
  if (D <= 0 || D > 4096) return MLQEM_ERR_BAD_ARG;
  if (!g_w_comp || !g_att_b || !lin_w || !lin_b || !att_w || !g_att_x || !g_lin_w || !g_lin_b || !g_att_w) return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(asap_compose_bwd_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), g_w_comp, g_att_b, lin_w, lin_b, att_w, g_att_x, D,
                     g_lin_w, g_lin_b, g_att_w);
  return launch_status();
}

// new_graph_ptr[g] = sum over h < g of ceil(float(n_h) * ratio), n_h = graph_ptr[h + 1] - graph_ptr[h]: the boundaries of the pooled
// batch (PyG topk keeps ceil(ratio * n) nodes per graph, evaluated in float32) from the DEVICE-resident boundaries of the batch.
// A captured step whose launch shapes do not depend on the per-graph sizes (train.BucketedTrainer's size-stable buckets) pools
// without a host-made boundary array.  One workgroup, B values in chunks of 1024 with a running carry.
__global__ __launch_bounds__(1024) void keep_ptr_kernel(const int32_t* __restrict__ graph_ptr, int64_t B, float ratio,
                                                        int32_t* __restrict__ new_ptr) {
  __shared__ int32_t part[1024];
  __shared__ int32_t carry;
  const int tid = threadIdx.x;
  if (tid == 0) { carry = 0; new_ptr[0] = 0; }
  __syncthreads();
  for (int64_t base = 0; base < B; base += 1024) {
    const int64_t g = base + tid;
    int32_t v = 0;
    if (g < B) v = (int32_t)ceilf((float)(graph_ptr[g + 1] - graph_ptr[g]) * ratio);
    part[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int32_t t = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += t;
      __syncthreads();
    }
    if (g < B) new_ptr[g + 1] = carry + part[tid];
    __syncthreads();
    if (tid == 1023) carry += part[1023];
    __syncthreads();
  }
}

extern "C" int mlqem_pool_keep_ptr(const int32_t* graph_ptr, int64_t B, float ratio, int32_t* new_graph_ptr, mlqem_stream_t stream_) {
  begin_launches();
  if (B < 0 || !(ratio > 0.0f) || ratio > 1.0f) return MLQEM_ERR_BAD_ARG;
  if (!graph_ptr || !new_graph_ptr) return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(keep_ptr_kernel, dim3(1), dim3(1024), 0, as_stream(stream_), graph_ptr, B, ratio, new_graph_ptr);
  return launch_status();
}

extern "C" size_t mlqem_segment_topk_workspace_bytes(int64_t N, int64_t B) {
  if (N <= 0 || B <= 0) return 256;
  const size_t keys = ((size_t)N * sizeof(uint64_t) + 255) / 256 * 256;
  return 2 * keys + topk_temp_bytes(N, B);
}

extern "C" int mlqem_segment_topk(const float* fitness, const int32_t* graph_ptr, const int32_t* new_graph_ptr,
                                  int64_t N, int64_t B, int64_t K, int64_t max_graph_nodes, int32_t* perm, int32_t* slot, void* workspace,
                                  size_t workspace_bytes, mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || B <= 0 || K < 0 || K > N || N >= 0x7fffffffLL) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (K == 0) {
    if (slot) fill_i32(slot, -1, N, stream);
    return launch_status();
  }
  if (!fitness || !graph_ptr || !new_graph_ptr || !perm) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_segment_topk_workspace_bytes(N, B)) return MLQEM_ERR_WORKSPACE;
  const size_t kb = ((size_t)N * sizeof(uint64_t) + 255) / 256 * 256;
  char* ws = static_cast<char*>(workspace);
  uint64_t* keys = reinterpret_cast<uint64_t*>(ws);
  uint64_t* sorted = reinterpret_cast<uint64_t*>(ws + kb);
  void* temp = ws + 2 * kb;
  size_t temp_bytes = topk_temp_bytes(N, B);
  // a graph has at most max_graph_nodes (the caller's bound; N when it has none) nodes: local indices fit idx_bits bits
  if (max_graph_nodes < 0 || max_graph_nodes > N) return MLQEM_ERR_BAD_ARG;
  const int idx_bits = bits_for(max_graph_nodes > 0 ? max_graph_nodes : N);
  // Graphs of thousands of nodes: the segmented sort gives a segment to ONE workgroup (64 workgroups on 256 CUs for a batch of
  // 64 100-qubit circuits: 0.40 ms for 0.7 M keys); with the graph index in the key's top bits one device-wide radix sort
  // does the same job with every CU.  Same keys below the graph bits, so the same permutation.
  if (max_graph_nodes > 0 && max_graph_nodes <= 1024 && B <= 0x7fffffffLL) {      // small graphs, and the caller vouches for the bound
    const dim3 grid((unsigned)B), block(kBlock);
    if (max_graph_nodes <= 64) hipLaunchKernelGGL(topk_small_kernel<64>, grid, block, 0, stream, fitness, graph_ptr, new_graph_ptr, perm, slot);
    else if (max_graph_nodes <= 256) hipLaunchKernelGGL(topk_small_kernel<256>, grid, block, 0, stream, fitness, graph_ptr, new_graph_ptr, perm, slot);
    else hipLaunchKernelGGL(topk_small_kernel<1024>, grid, block, 0, stream, fitness, graph_ptr, new_graph_ptr, perm, slot);
    return launch_status();
  }
  const int graph_bits = bits_for(B);
  const bool whole = N / B >= 1024 && graph_bits + 32 + idx_bits <= 64;
  if (N / B >= 1024) {
    // ... and since round 5 two launches of this file instead of the ~20 of a device-wide merge sort: chunks of a graph sorted in LDS,
    // then every node's rank among its graph's chunks.  Same order (fitness descending, equal fitness by index), so the same perm.
    // (chunk size, round 6 on a 64-circuit step, both poolings, sort + rank: 4 096 keys 117 + 35 us -- 64 graphs of 3 680 nodes are 64
    // workgroups on 256 CUs, 78 passes of 8 compare-exchanges per thread; 2 048: 64 + 57; 1 024: 42 + 101; 512: 28 + 182 -- the rank
    // pass searches every other chunk of the graph)
    constexpr int kTopkChunk = 2048;
    const int64_t chunks = N / kTopkChunk + B + 1;
    hipLaunchKernelGGL(topk_chunk_sort_kernel<kTopkChunk>, dim3((unsigned)chunks), dim3(kBlock), 0, stream, fitness, graph_ptr, (int)B, sorted);
    hipLaunchKernelGGL(topk_rank_select_kernel<kTopkChunk>, dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, stream, sorted, graph_ptr,
                       new_graph_ptr, (int)B, N, perm, slot);
    return launch_status();
  }
  hipLaunchKernelGGL(topk_keys_kernel, dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, stream, fitness,
                     graph_ptr, (int)B, N, idx_bits, whole ? graph_bits : 0, keys);
  if (whole) {
    if (rocprim::radix_sort_keys_desc<TopkSortConfig>(temp, temp_bytes, keys, sorted, (size_t)N, 0, (unsigned)(graph_bits + 32 + idx_bits), stream) != hipSuccess)
      return MLQEM_ERR_LAUNCH;
  } else if (rocprim::segmented_radix_sort_keys_desc(temp, temp_bytes, keys, sorted, (unsigned)N, (unsigned)B, graph_ptr,
                                                     graph_ptr + 1, 0, 32 + idx_bits, stream) != hipSuccess)
    return MLQEM_ERR_LAUNCH;
  hipLaunchKernelGGL(topk_select_kernel, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, stream, sorted,
                     graph_ptr, new_graph_ptr, (int)B, K, idx_bits, perm);
  if (slot) {                                          // (this form does not see every node: the map as launches of its own)
    fill_i32(slot, -1, N, stream);
    hipLaunchKernelGGL(slot_map_kernel, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, stream, perm, K, slot);
  }
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
  fill_i32(slot, -1, N, stream);
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

static void rows_layout(int64_t K, int kmax, size_t& bm, size_t& deg) {
  const size_t Wk = (size_t)(kmax + 31) / 32;
  bm = ((size_t)K * Wk * sizeof(uint32_t) + 255) / 256 * 256;      // the transpose (bits + ranks, 8 bytes per word) takes 2 * bm
  deg = ((size_t)(K + 1) * sizeof(int32_t) + 255) / 256 * 256;
}

extern "C" size_t mlqem_asap_coarsen_rows_workspace_bytes(int64_t K, int kmax) {
  if (K < 0 || kmax < 0) return 0;
  size_t bm, deg;
  rows_layout(K, kmax, bm, deg);
  return 3 * bm + 2 * deg + dense_scan_bytes(K);
}

// slot[] in ONE launch when the graphs' boundaries are at hand: workgroup g wipes its graph's nodes and, behind a barrier, files the
// graph's kept centres (a graph's centres are its own nodes) -- a fill and a scatter launch before.
__global__ __launch_bounds__(kBlock) void slot_map_graphs_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ gptr,
                                                                 const int32_t* __restrict__ new_gptr, int32_t* __restrict__ slot) {
  const int g = blockIdx.x;
  const int n0 = gptr[g], n1 = gptr[g + 1], k0 = new_gptr[g], k1 = new_gptr[g + 1];
  for (int i = n0 + threadIdx.x; i < n1; i += kBlock) slot[i] = -1;
  __syncthreads();
  for (int p = k0 + threadIdx.x; p < k1; p += kBlock) slot[perm[p]] = p;
}

extern "C" int mlqem_asap_slot_map_graphs(const int32_t* perm, const int32_t* graph_ptr, const int32_t* new_graph_ptr, int64_t B, int64_t N,
                                          int64_t K, int32_t* slot, mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || K < 0 || K > N || B < 0 || B > 0x7fffffffLL) return MLQEM_ERR_BAD_ARG;
  if (N == 0 || B == 0) return MLQEM_OK;
  if (!slot || !graph_ptr || !new_graph_ptr || (K > 0 && !perm)) return MLQEM_ERR_BAD_ARG;
  if (N / B > 65536) {          // a few huge graphs: a workgroup per graph would walk its nodes alone -- the two-launch form
    fill_i32(slot, -1, N, stream);
    if (K > 0) hipLaunchKernelGGL(slot_map_kernel, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, stream, perm, K, slot);
    return launch_status();
  }
  hipLaunchKernelGGL(slot_map_graphs_kernel, dim3((unsigned)B), dim3(kBlock), 0, stream, perm, graph_ptr, new_graph_ptr, slot);
  return launch_status();
}

extern "C" int mlqem_asap_slot_map(const int32_t* perm, int64_t N, int64_t K, int32_t* slot, mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || K < 0 || K > N) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!slot || (K > 0 && !perm)) return MLQEM_ERR_BAD_ARG;
  fill_i32(slot, -1, N, stream);
  if (K > 0) hipLaunchKernelGGL(slot_map_kernel, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, stream, perm, K, slot);
  return launch_status();
}

extern "C" int mlqem_asap_coarsen_rows_max_bits(void) { return 64 * 1024 * 8 / 4; }   // (n_g + 2 k_g) bits per wave: 64 KB of LDS, four waves

static bool rows_args(RowsArgs& a, const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                      const int32_t* graph_ptr, const int32_t* new_graph_ptr, const int32_t* perm, const int32_t* slot,
                      int64_t K, int64_t B, int nmax, int kmax, void* workspace) {
  size_t bm, deg;
  rows_layout(K, kmax, bm, deg);
  char* ws = static_cast<char*>(workspace);
  const int Wk = (kmax + 31) / 32;
  a = RowsArgs{in_ptr, in_src, out_ptr, out_dst, graph_ptr, new_graph_ptr, perm, slot, (int)B, K, (nmax + 31) / 32, Wk,
               reinterpret_cast<uint32_t*>(ws), reinterpret_cast<uint64_t*>(ws + bm), reinterpret_cast<int32_t*>(ws + 3 * bm),
               reinterpret_cast<int32_t*>(ws + 3 * bm + deg)};
  return true;
}

// Pass 1: slot[], the two bit matrices (kept in the workspace for pass 2) and both CSR pointer arrays; new_out_ptr[K] is the
// edge total the caller reads to size the edge arrays.  nmax / kmax: largest graph / largest pooled graph of the batch.
extern "C" int mlqem_asap_coarsen_rows_count(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                             const int32_t* out_dst, const int32_t* graph_ptr, const int32_t* new_graph_ptr,
                                             const int32_t* perm, int64_t N, int64_t K, int64_t B, int nmax, int kmax,
                                             int32_t* slot, int32_t* new_in_ptr, int32_t* new_out_ptr, void* workspace,
                                             size_t workspace_bytes, mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || K < 0 || K > N || B < 0 || kmax < 0 || nmax < 0 || N >= 0x7fffffffLL) return MLQEM_ERR_BAD_ARG;
  if (nmax + 2 * kmax + 96 > mlqem_asap_coarsen_rows_max_bits()) return MLQEM_ERR_UNSUPPORTED;
  if (!slot || !new_in_ptr || !new_out_ptr) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_asap_coarsen_rows_workspace_bytes(K, kmax)) return MLQEM_ERR_WORKSPACE;
  fill_i32(slot, -1, N, stream);
  if (K == 0 || B == 0) {
    fill_i32(new_in_ptr, 0, K + 1, stream);
    fill_i32(new_out_ptr, 0, K + 1, stream);
    return launch_status();
  }
  if (!in_ptr || !out_ptr || !graph_ptr || !new_graph_ptr || !perm) return MLQEM_ERR_BAD_ARG;
  RowsArgs a;
  rows_args(a, in_ptr, in_src, out_ptr, out_dst, graph_ptr, new_graph_ptr, perm, slot, K, B, nmax, kmax, workspace);
  size_t bm, deg;
  rows_layout(K, kmax, bm, deg);
  fill_i32(a.outdeg + K, 0, 1, stream);
  fill_i32(a.indeg + K, 0, 1, stream);
  hipLaunchKernelGGL(slot_map_kernel, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, stream, perm, K, slot);
  const size_t lds = (size_t)4 * (a.Wn + 2 * a.Wk) * sizeof(uint32_t);
  const unsigned grid = (unsigned)ceil_div(K, (int64_t)4);
  hipLaunchKernelGGL(coarsen_rows_kernel, dim3(grid), dim3(kBlock), lds, stream, a);
  void* temp = static_cast<char*>(workspace) + 3 * bm + 2 * deg;
  size_t temp_bytes = dense_scan_bytes(K);
  if (rocprim::exclusive_scan(temp, temp_bytes, a.outdeg, new_out_ptr, (int32_t)0, (size_t)(K + 1), rocprim::plus<int32_t>(),
                              stream) != hipSuccess)
    return MLQEM_ERR_LAUNCH;
  if (rocprim::exclusive_scan(temp, temp_bytes, a.indeg, new_in_ptr, (int32_t)0, (size_t)(K + 1), rocprim::plus<int32_t>(),
                              stream) != hipSuccess)
    return MLQEM_ERR_LAUNCH;
  return launch_status();
}

// Pass 2 (same workspace, untouched in between): the edge arrays, each sized new_out_ptr[K].
extern "C" int mlqem_asap_coarsen_rows_fill(const int32_t* new_graph_ptr, int64_t K, int64_t B, int kmax, const int32_t* new_in_ptr,
                                            const int32_t* new_out_ptr, int32_t* new_in_src, int32_t* new_out_dst,
                                            int32_t* new_out_eid, const void* workspace, size_t workspace_bytes,
                                            mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (K < 0 || B < 0 || kmax < 0) return MLQEM_ERR_BAD_ARG;
  if (K == 0 || B == 0) return MLQEM_OK;
  if (!new_graph_ptr || !new_in_ptr || !new_out_ptr || !new_in_src || !new_out_dst || !new_out_eid) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_asap_coarsen_rows_workspace_bytes(K, kmax)) return MLQEM_ERR_WORKSPACE;
  RowsArgs a;
  rows_args(a, nullptr, nullptr, nullptr, nullptr, nullptr, new_graph_ptr, nullptr, nullptr, K, B, 0, kmax,
            const_cast<void*>(workspace));
  hipLaunchKernelGGL(coarsen_rows_fill_kernel, dim3((unsigned)ceil_div(K, (int64_t)4)), dim3(kBlock), 0, stream, a, new_in_ptr,
                     new_out_ptr, new_in_src, new_out_dst, new_out_eid);
  return launch_status();
}

// ---- list form (round 4).  Workspace: h_out, h_in, roff_o, roff_i [N + 1] int64 | ccnt, rcnt_o, rcnt_i [N] | clist [E + N] |
//      cap_o, cap_i, off_o, off_i [K + 1] int64 | outdeg, indeg [K + 1] | flag | scan temp | r_o, r_i, tmp_o, tmp_i [capacity]
//      (rcnt_o + rcnt_i + ... : the per-node records rinfo [N][4] take the place of four count arrays)
//      (r_o is reused as out_row by the fill pass)
namespace {
struct ListsLayout {
  size_t h, ncnt, clist, caps, degs, flag, scan, lists, total;
};
// TWO arrays of one length scanned by ONE call (a zip of the two, added component-wise): rocprim's device scan is two launches per call,
// and the six scans of a list coarsening were twelve of a 64-circuit step's 115 launches
template <class T> struct PairPlus {
  __host__ __device__ rocprim::tuple<T, T> operator()(const rocprim::tuple<T, T>& a, const rocprim::tuple<T, T>& b) const {
    return rocprim::make_tuple(rocprim::get<0>(a) + rocprim::get<0>(b), rocprim::get<1>(a) + rocprim::get<1>(b));
  }
};
template <class T> hipError_t pair_scan(void* temp, size_t& bytes, T* a, T* b, T* oa, T* ob, size_t n, hipStream_t stream) {
  auto in = rocprim::make_zip_iterator(rocprim::make_tuple(a, b));
  auto out = rocprim::make_zip_iterator(rocprim::make_tuple(oa, ob));
  return rocprim::exclusive_scan(temp, bytes, in, out, rocprim::make_tuple(T(0), T(0)), n, PairPlus<T>(), stream);
}
size_t lists_scan_bytes(int64_t n) {
  size_t t64 = 0, t32 = 0;
  (void)pair_scan<int64_t>(nullptr, t64, nullptr, nullptr, nullptr, nullptr, (size_t)(n + 1), (hipStream_t)0);
  (void)pair_scan<int32_t>(nullptr, t32, nullptr, nullptr, nullptr, nullptr, (size_t)(n + 1), (hipStream_t)0);
  return (std::max(t64, t32) + 255) / 256 * 256;
}
ListsLayout lists_layout(int64_t N, int64_t K, int64_t E, int64_t capacity) {
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  ListsLayout l;
  l.h = up((size_t)(N + 1) * sizeof(int64_t));
  l.ncnt = up((size_t)std::max<int64_t>(N, 1) * sizeof(int32_t));
  l.clist = up((size_t)std::max<int64_t>(E + N, 1) * sizeof(int32_t));
  l.caps = up((size_t)(K + 1) * sizeof(int64_t));
  l.degs = up((size_t)(K + 1) * sizeof(int32_t));
  l.flag = 256;
  l.scan = lists_scan_bytes(std::max(N, K));
  l.lists = up((size_t)std::max<int64_t>(capacity, 1) * sizeof(int32_t));
  l.total = 4 * l.h + 5 * l.ncnt + l.clist + 4 * l.caps + 8 * l.degs + l.flag + l.scan + 4 * l.lists;
  return l;
}

struct ListsPointers {
  int64_t* h_out; int64_t* h_in; int64_t* roff_o; int64_t* roff_i; int32_t* ccnt; uint32_t* rinfo; int32_t* clist;
  int64_t* cap_o; int64_t* cap_i; int64_t* off_o; int64_t* off_i; int32_t* outdeg; int32_t* indeg; uint32_t* cinfo; int2* cgraph; int32_t* flag; void* scan;
  int32_t* r_o; int32_t* r_i; int32_t* tmp_o; int32_t* tmp_i;
};
ListsPointers lists_pointers(void* workspace, const ListsLayout& l) {
  char* w = static_cast<char*>(workspace);
  auto take = [&](size_t bytes) { char* at = w; w += bytes; return at; };
  ListsPointers q;
  q.h_out = reinterpret_cast<int64_t*>(take(l.h));
  q.h_in = reinterpret_cast<int64_t*>(take(l.h));
  q.roff_o = reinterpret_cast<int64_t*>(take(l.h));
  q.roff_i = reinterpret_cast<int64_t*>(take(l.h));
  q.ccnt = reinterpret_cast<int32_t*>(take(l.ncnt));
  q.rinfo = reinterpret_cast<uint32_t*>(take(4 * l.ncnt));
  q.clist = reinterpret_cast<int32_t*>(take(l.clist));
  q.cap_o = reinterpret_cast<int64_t*>(take(l.caps));
  q.cap_i = reinterpret_cast<int64_t*>(take(l.caps));
  q.off_o = reinterpret_cast<int64_t*>(take(l.caps));
  q.off_i = reinterpret_cast<int64_t*>(take(l.caps));
  q.outdeg = reinterpret_cast<int32_t*>(take(l.degs));
  q.indeg = reinterpret_cast<int32_t*>(take(l.degs));
  q.cinfo = reinterpret_cast<uint32_t*>(take(4 * l.degs));          // [K][4]: 16-byte records
  q.cgraph = reinterpret_cast<int2*>(take(2 * l.degs));             // [K]: first cluster and bitset words of the cluster's graph
  q.flag = reinterpret_cast<int32_t*>(take(l.flag));
  q.scan = take(l.scan);
  q.r_o = reinterpret_cast<int32_t*>(take(l.lists));
  q.r_i = reinterpret_cast<int32_t*>(take(l.lists));
  q.tmp_o = reinterpret_cast<int32_t*>(take(l.lists));
  q.tmp_i = reinterpret_cast<int32_t*>(take(l.lists));
  return q;
}

// structural sizes, row bounds and their scans: roff_o / roff_i [N + 1], off_o / off_i [K + 1] (the last entries = the totals)
int lists_caps(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst, const int32_t* new_graph_ptr,
               const int32_t* perm, int64_t N, int64_t K, int64_t B, const ListsPointers& q, const ListsLayout& l, hipStream_t stream,
               int32_t* zero_rows = nullptr) {
  hipLaunchKernelGGL(coarsen_reach_kernel, dim3((unsigned)ceil_div(N + 1, (int64_t)kBlock)), dim3(kBlock), 0, stream, in_ptr, in_src, out_ptr,
                     out_dst, N, q.h_out, q.h_in);
  hipLaunchKernelGGL(coarsen_row_caps_kernel, dim3((unsigned)ceil_div(K + 1, (int64_t)kBlock)), dim3(kBlock), 0, stream, in_ptr, in_src, perm,
                     new_graph_ptr, (int)B, K, q.h_out, q.h_in, q.cap_o, q.cap_i, zero_rows);
  size_t temp_bytes = l.scan;
  if (pair_scan<int64_t>(q.scan, temp_bytes, q.h_out, q.h_in, q.roff_o, q.roff_i, (size_t)(N + 1), stream) != hipSuccess) return MLQEM_ERR_LAUNCH;
  temp_bytes = l.scan;
  if (pair_scan<int64_t>(q.scan, temp_bytes, q.cap_o, q.cap_i, q.off_o, q.off_i, (size_t)(K + 1), stream) != hipSuccess) return MLQEM_ERR_LAUNCH;
  return MLQEM_OK;
}

__global__ void lists_totals_kernel(const int64_t* __restrict__ off_o, const int64_t* __restrict__ off_i, const int64_t* __restrict__ roff_o,
                                    const int64_t* __restrict__ roff_i, int64_t K, int64_t N, int64_t* __restrict__ totals) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { totals[0] = off_o[K]; totals[1] = off_i[K]; totals[2] = roff_o[N]; totals[3] = roff_i[N]; }
}

int device_cus() {
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return cus;
}
constexpr int kListsMaxLds = 160 * 1024;             // one workgroup may take a CU's whole LDS
constexpr int kListsMaxK = 65535;                    // a row's two lengths travel as one packed int through the wave scan
}  // namespace

extern "C" size_t mlqem_asap_coarsen_lists_workspace_bytes(int64_t N, int64_t K, int64_t E, int64_t capacity) {
  if (N < 0 || K < 0 || E < 0 || capacity < 0) return 0;
  return lists_layout(N, K, E, capacity).total;
}

extern "C" int mlqem_asap_coarsen_lists_max_k(void) { return kListsMaxK; }

extern "C" int mlqem_asap_coarsen_lists_caps(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                             const int32_t* new_graph_ptr, const int32_t* perm, int64_t N, int64_t K, int64_t B,
                                             int64_t* totals, void* workspace, size_t workspace_bytes, mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || K < 0 || K > N || B < 0 || N >= 0x7fffffffLL || !totals) return MLQEM_ERR_BAD_ARG;
  const ListsLayout l = lists_layout(N, K, 0, 0);
  if (!workspace || workspace_bytes < l.total) return MLQEM_ERR_WORKSPACE;
  if (N > 0 && (!in_ptr || !out_ptr)) return MLQEM_ERR_BAD_ARG;
  if (K > 0 && (!new_graph_ptr || !perm || B == 0)) return MLQEM_ERR_BAD_ARG;
  const ListsPointers q = lists_pointers(workspace, l);
  const int rc = lists_caps(in_ptr, in_src, out_ptr, out_dst, new_graph_ptr, perm, N, K, B, q, l, stream);
  if (rc != MLQEM_OK) return rc;
  hipLaunchKernelGGL(lists_totals_kernel, dim3(1), dim3(64), 0, stream, q.off_o, q.off_i, q.roff_o, q.roff_i, K, N, totals);
  return launch_status();
}

// Pass 1: slot[], the per-node lists, the rows as sorted lists in the workspace, both CSR pointer arrays.  E: the number of stored
// edges of the input structure (out_ptr[N]) or a bound on it.  `capacity`: entries each of the four list buffers holds -- a bound
// on all four totals mlqem_asap_coarsen_lists_caps reports (GraphArena knows a structural one); a list that would leave its
// buffer is dropped, nothing is written past one, and pass 2 reports it.
extern "C" int mlqem_asap_coarsen_lists_count(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                              const int32_t* out_dst, const int32_t* graph_ptr, const int32_t* new_graph_ptr,
                                              const int32_t* perm, int64_t N, int64_t K, int64_t B, int64_t E, int kmax, int64_t capacity,
                                              int32_t* slot, int slot_ready, int32_t* new_in_ptr, int32_t* new_out_ptr, int32_t* new_loops,
                                              void* workspace, size_t workspace_bytes, mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || K < 0 || K > N || B < 0 || E < 0 || kmax < 0 || capacity < 0 || N >= 0x7fffffffLL) return MLQEM_ERR_BAD_ARG;
  if (kmax > kListsMaxK || capacity >= (1ll << 32)) return MLQEM_ERR_UNSUPPORTED;      // 32-bit places in the per-node records
  if (!slot || !new_in_ptr || !new_out_ptr) return MLQEM_ERR_BAD_ARG;
  const ListsLayout l = lists_layout(N, K, E, capacity);
  if (!workspace || workspace_bytes < l.total) return MLQEM_ERR_WORKSPACE;
  if (!slot_ready) fill_i32(slot, -1, N, stream);
  if (K == 0 || B == 0) {
    fill_i32(new_in_ptr, 0, K + 1, stream);
    fill_i32(new_out_ptr, 0, K + 1, stream);
    return launch_status();
  }
  if (!in_ptr || !out_ptr || !graph_ptr || !new_graph_ptr || !perm || (E > 0 && (!in_src || !out_dst))) return MLQEM_ERR_BAD_ARG;
  const ListsPointers q = lists_pointers(workspace, l);
  if (!slot_ready) hipLaunchKernelGGL(slot_map_kernel, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, stream, perm, K, slot);
  const int rc = lists_caps(in_ptr, in_src, out_ptr, out_dst, new_graph_ptr, perm, N, K, B, q, l, stream, new_loops);
  if (rc != MLQEM_OK) return rc;
  const unsigned node_blocks = (unsigned)ceil_div(N, (int64_t)kBlock);
  // (the three words the later passes count into -- the scanned degree arrays' last entries and the overflow flag -- are zeroed by the
  // first thread of this launch: three one-word fill launches)
  hipLaunchKernelGGL(coarsen_clists_kernel, dim3(node_blocks), dim3(kBlock), 0, stream, out_ptr, out_dst, slot, N, q.clist, q.ccnt, q.outdeg + K,
                     q.indeg + K, q.flag);
  hipLaunchKernelGGL(coarsen_rlists_kernel<false>, dim3(node_blocks), dim3(kBlock), 0, stream, out_ptr, out_dst, out_ptr, q.clist, q.ccnt, N,
                     q.roff_o, q.r_o, capacity, q.rinfo, q.flag);
  hipLaunchKernelGGL(coarsen_rlists_kernel<true>, dim3(node_blocks), dim3(kBlock), 0, stream, in_ptr, in_src, out_ptr, q.clist, q.ccnt, N,
                     q.roff_i, q.r_i, capacity, q.rinfo, q.flag);
  ListsArgs a{in_ptr, in_src, out_ptr, out_dst, graph_ptr, new_graph_ptr, perm, slot, (int)B, N, K, (kmax + 31) / 32,
              q.rinfo, q.r_o, q.r_i, capacity, q.off_o, q.off_i, q.tmp_o, q.tmp_i, capacity, q.outdeg, q.indeg, q.flag};
  hipLaunchKernelGGL(coarsen_gather_kernel, dim3((unsigned)ceil_div(K, (int64_t)kBlock)), dim3(kBlock), 0, stream, a, q.cinfo, q.cgraph);
  const size_t lds = (size_t)4 * 2 * a.Wk * sizeof(uint32_t);
  if (!ensure_dynamic_lds(coarsen_unique_kernel, (size_t)kListsMaxLds)) return MLQEM_ERR_LAUNCH;      // per device (common.hpp)
  // persistent waves: as many workgroups as the LDS lets a CU hold (at most 8: 32 waves), never more than there are clusters
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)kListsMaxLds / std::max<size_t>(lds, 1)));
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(K, (int64_t)4), (int64_t)device_cus() * per_cu));
  hipLaunchKernelGGL(coarsen_unique_kernel, dim3(grid), dim3(kBlock), lds, stream, a, q.cinfo, q.cgraph);
  size_t temp_bytes = l.scan;
  if (pair_scan<int32_t>(q.scan, temp_bytes, q.outdeg, q.indeg, new_out_ptr, new_in_ptr, (size_t)(K + 1), stream) != hipSuccess)
    return MLQEM_ERR_LAUNCH;
  return launch_status();
}

// Pass 2 (same workspace, N, K, E and capacity, untouched in between): the edge arrays, each holding `edge_capacity` entries
// (new_out_ptr[K] <= edge_capacity <= capacity; nothing is written past it).  *overflow (device, optional) = 1 when pass 1
// dropped a list.
extern "C" int mlqem_asap_coarsen_lists_fill(int64_t N, int64_t K, int64_t E, int64_t capacity, const int32_t* new_in_ptr,
                                             const int32_t* new_out_ptr, int32_t* new_in_src, int32_t* new_out_dst, int32_t* new_out_eid,
                                             int64_t edge_capacity, int32_t* overflow, void* workspace, size_t workspace_bytes,
                                             mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || K < 0 || E < 0 || capacity < 0 || edge_capacity < 0 || edge_capacity > capacity) return MLQEM_ERR_BAD_ARG;
  if (K == 0) return MLQEM_OK;
  if (!new_in_ptr || !new_out_ptr || (edge_capacity > 0 && (!new_in_src || !new_out_dst))) return MLQEM_ERR_BAD_ARG;
  const ListsLayout l = lists_layout(N, K, E, capacity);
  if (!workspace || workspace_bytes < l.total) return MLQEM_ERR_WORKSPACE;
  const ListsPointers q = lists_pointers(workspace, l);
  ListsArgs a{};
  a.N = N; a.K = K;
  a.off_o = q.off_o; a.off_i = q.off_i; a.tmp_o = q.tmp_o; a.tmp_i = q.tmp_i; a.tmp_cap = capacity;
  a.outdeg = q.outdeg; a.indeg = q.indeg; a.overflow = q.flag;
  int32_t* out_row = new_out_eid ? q.r_o : nullptr;  // the R lists are dead after pass 1; no out_eid wanted: no link pass
  hipLaunchKernelGGL(coarsen_lists_copy_kernel, dim3((unsigned)ceil_div(K * kGroup, (int64_t)kBlock)), dim3(kBlock), 0, stream, a, new_in_ptr,
                     new_out_ptr, new_in_src, new_out_dst, out_row, edge_capacity, overflow);
  if (edge_capacity > 0 && new_out_eid)
    hipLaunchKernelGGL(coarsen_lists_link_kernel, dim3((unsigned)(device_cus() * 8)), dim3(kBlock), 0, stream, new_in_ptr, new_in_src,
                       new_out_dst, out_row, new_out_ptr + K, edge_capacity, new_out_eid);
  return launch_status();
}

extern "C" int mlqem_asap_coarsen_dense_max_k(void) { return kDenseMaxK; }

extern "C" size_t mlqem_asap_coarsen_dense_workspace_bytes(int64_t B, int64_t K, int kmax) {
  if (B < 0 || K < 0 || kmax < 0) return 0;
  const size_t W = (size_t)(kmax + 31) / 32;
  const size_t bm = ((size_t)B * kmax * W * sizeof(uint32_t) + 255) / 256 * 256;
  const size_t deg = ((size_t)(K + 1) * sizeof(int32_t) + 255) / 256 * 256;
  return 2 * bm + 2 * deg + dense_scan_bytes(K);      // the bit matrices and their transposes | both degree vectors | scan temp
}

extern "C" int mlqem_asap_coarsen_dense(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                        const int32_t* out_dst, const int32_t* graph_ptr, const int32_t* new_graph_ptr,
                                        const int32_t* perm, int64_t N, int64_t K, int64_t B, int kmax, int32_t* slot, int slot_ready,
                                        int32_t* new_in_ptr, int32_t* new_in_src, int32_t* new_out_ptr, int32_t* new_out_dst,
                                        int32_t* new_out_eid, int32_t* new_loops, void* workspace, size_t workspace_bytes,
                                        mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || K < 0 || K > N || B < 0 || kmax < 0 || N >= 0x7fffffffLL) return MLQEM_ERR_BAD_ARG;
  if (kmax > kDenseMaxK) return MLQEM_ERR_UNSUPPORTED;
  if (!slot || !new_in_ptr || !new_out_ptr) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_asap_coarsen_dense_workspace_bytes(B, K, kmax)) return MLQEM_ERR_WORKSPACE;
  // slot_ready: `slot` already holds mlqem_asap_slot_map(perm) (the pooling's forward made it for its backward): two launches less
  if (!slot_ready) fill_i32(slot, -1, N, stream);
  if (K == 0 || B == 0) {
    fill_i32(new_in_ptr, 0, K + 1, stream);
    fill_i32(new_out_ptr, 0, K + 1, stream);
    return launch_status();
  }
  if (!in_ptr || !out_ptr || !graph_ptr || !new_graph_ptr || !perm || !new_in_src || !new_out_dst || !new_loops)
    return MLQEM_ERR_BAD_ARG;
  const int W = (kmax + 31) / 32;
  const size_t bm = ((size_t)B * kmax * W * sizeof(uint32_t) + 255) / 256 * 256;
  const size_t deg = ((size_t)(K + 1) * sizeof(int32_t) + 255) / 256 * 256;
  char* ws = static_cast<char*>(workspace);
  DenseArgs a{in_ptr, in_src, out_ptr, out_dst, graph_ptr, new_graph_ptr, slot, new_loops, K, W, reinterpret_cast<uint32_t*>(ws),
              reinterpret_cast<uint32_t*>(ws + bm), kmax, reinterpret_cast<int32_t*>(ws + 2 * bm), reinterpret_cast<int32_t*>(ws + 2 * bm + deg)};
  void* temp = ws + 2 * bm + 2 * deg;
  size_t temp_bytes = dense_scan_bytes(K);
  if (!slot_ready) hipLaunchKernelGGL(slot_map_kernel, dim3((unsigned)ceil_div(K, kBlock)), dim3(kBlock), 0, stream, perm, K, slot);
  const size_t lds = (size_t)kmax * W * sizeof(uint32_t);      // one k x k bit matrix; the kernels hold two and three of them
  if (!ensure_dynamic_lds(coarsen_dense_bitmap_kernel, 2 * lds) || !ensure_dynamic_lds(coarsen_dense_fill_kernel, 3 * lds)) return MLQEM_ERR_LAUNCH;
  hipLaunchKernelGGL(coarsen_dense_bitmap_kernel, dim3((unsigned)B), dim3(kBlock), 2 * lds, stream, a);
  if (K + 1 <= kDualScanMax) {
    hipLaunchKernelGGL(dual_scan_small_kernel, dim3(1), dim3(kDualScanThreads), 0, stream, a.outdeg, a.indeg, (int)(K + 1), new_out_ptr, new_in_ptr);
  } else {
    if (rocprim::exclusive_scan(temp, temp_bytes, a.outdeg, new_out_ptr, (int32_t)0, (size_t)(K + 1), rocprim::plus<int32_t>(),
                                stream) != hipSuccess)
      return MLQEM_ERR_LAUNCH;
    if (rocprim::exclusive_scan(temp, temp_bytes, a.indeg, new_in_ptr, (int32_t)0, (size_t)(K + 1), rocprim::plus<int32_t>(),
                                stream) != hipSuccess)
      return MLQEM_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(coarsen_dense_fill_kernel, dim3((unsigned)B), dim3(kBlock), 3 * lds, stream, a, new_in_ptr, new_out_ptr,
                     new_in_src, new_out_dst, new_out_eid);
  return launch_status();
}
