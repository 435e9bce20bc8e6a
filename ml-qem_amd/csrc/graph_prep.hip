// Graph-structure preparation: COO edge list -> the two CSR structures + self-loop counts, the per-node
// normalisation scalars of the convolutions, and on-device batch assembly from a resident dataset arena.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace mlqem {

__global__ __launch_bounds__(kBlock) void split_edges_kernel(const int64_t* __restrict__ ei, int64_t E, int32_t N,
                                                             int32_t* __restrict__ key_d, int32_t* __restrict__ key_s,
                                                             int32_t* __restrict__ eid, int32_t* __restrict__ loops) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  const int32_t s = (int32_t)ei[e], d = (int32_t)ei[E + e];
  const bool loop = s == d;
  if (loop) atomicAdd(&loops[s], 1);
  // self-loops sort behind every real row (key N) and fall outside ptr[N]
  key_d[e] = loop ? N : d;
  key_s[e] = loop ? N : s;
  eid[e] = (int32_t)e;
}

// After the stable sort by destination: in_src[p] = src[eid_in[p]] and pos_in[eid_in[p]] = p.
__global__ __launch_bounds__(kBlock) void gather_in_kernel(const int64_t* __restrict__ ei, int64_t E,
                                                           const int32_t* __restrict__ eid_in,
                                                           int32_t* __restrict__ in_src, int32_t* __restrict__ pos_in) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= E) return;
  const int32_t e = eid_in[p];
  in_src[p] = (int32_t)ei[e];
  pos_in[e] = (int32_t)p;
}

// After the stable sort by source: out_dst[p] = dst[eid_out[p]], out_eid[p] = position of that edge in the in-CSR.
__global__ __launch_bounds__(kBlock) void gather_out_kernel(const int64_t* __restrict__ ei, int64_t E,
                                                            const int32_t* __restrict__ eid_out,
                                                            const int32_t* __restrict__ pos_in,
                                                            int32_t* __restrict__ out_dst, int32_t* __restrict__ out_eid) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= E) return;
  const int32_t e = eid_out[p];
  out_dst[p] = (int32_t)ei[E + e];
  if (out_eid) out_eid[p] = pos_in[e];
}

// ptr[i] = first position whose sorted key is >= i  (i = 0..N)
__global__ __launch_bounds__(kBlock) void lower_bound_kernel(const int32_t* __restrict__ sorted, int64_t E, int64_t N,
                                                             int32_t* __restrict__ ptr) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i > N) return;
  int64_t lo = 0, hi = E;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted[mid] < (int32_t)i) lo = mid + 1; else hi = mid;
  }
  ptr[i] = (int32_t)lo;
}

__global__ __launch_bounds__(kBlock) void graph_norms_kernel(const int32_t* __restrict__ in_ptr,
                                                             const int32_t* __restrict__ out_ptr,
                                                             const int32_t* __restrict__ loops, int64_t N,
                                                             float* __restrict__ gcn_dinv, float* __restrict__ sage_rinv,
                                                             float* __restrict__ cheb_dinv) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const int indeg = in_ptr[i + 1] - in_ptr[i];
  if (gcn_dinv) gcn_dinv[i] = 1.0f / sqrtf((float)(indeg + 1));
  if (sage_rinv) {
    const int cnt = indeg + (loops ? loops[i] : 0);
    sage_rinv[i] = 1.0f / (float)(cnt > 0 ? cnt : 1);
  }
  if (cheb_dinv) {
    const int outdeg = out_ptr[i + 1] - out_ptr[i];
    cheb_dinv[i] = outdeg > 0 ? 1.0f / sqrtf((float)outdeg) : 0.f;
  }
}

struct SortPlan {
  size_t keys_bytes, temp_bytes;
  int end_bit;
};

static SortPlan plan_sort(int64_t N, int64_t E) {
  SortPlan p{};
  p.keys_bytes = ((size_t)E * sizeof(int32_t) + 255) / 256 * 256;
  int bits = 1;
  while ((1ll << bits) <= N) ++bits;  // keys are in [0, N]
  p.end_bit = bits;
  size_t temp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, temp, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                  (int32_t*)nullptr, (size_t)E, 0, (unsigned)bits, (hipStream_t)0);
  p.temp_bytes = (temp + 255) / 256 * 256;
  return p;
}

// ------------------------------------------------------------------------------------- batch assembly
struct AssembleArgs {
  const float* x; int64_t ldx; int F;
  const float* nscal; int K;
  const int32_t* a_gptr; const int32_t* a_in_ptr; const int32_t* a_in_src;
  const int32_t* a_out_ptr; const int32_t* a_out_dst; const int32_t* a_out_eid; const int32_t* a_loops;
  const int32_t* sel; const int32_t* b_nptr; const int32_t* b_eptr;
  int B; int64_t Nb; int64_t Eb;
  float* xb; int64_t ldxb; float* nscal_b;
  float* derived_b;   // optional planar [3, Nb]: gcn_dinv^2, loops * sage_rinv, -cheb_dinv (needs K >= 3 and loops)
  int32_t* src_node;   // [Nb] arena row of every batch node (written by the nodes kernel, read by the rows kernel)
  int32_t* in_ptr_b; int32_t* in_src_b; int32_t* out_ptr_b; int32_t* out_dst_b; int32_t* out_eid_b; int32_t* loops_b;
  const int32_t* a_in_ell; const int32_t* a_out_ell;   // optional [N,2] side tables of the arena (global ids)
  int32_t* in_ell_b; int32_t* out_ell_b;
};

__device__ __forceinline__ int find_segment(const int32_t* __restrict__ ptr, int n_seg, int32_t v) {
  int lo = 0, hi = n_seg;  // largest b with ptr[b] <= v
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] <= v) lo = mid; else hi = mid;
  }
  return lo;
}

// rows of x (and of the packed per-node scalars): flattened (node, 16-byte chunk) so loads and stores stay coalesced.
// VEC4: both the arena's and the batch's feature rows are 16-byte aligned with F a multiple of 4 (the padded layout).
template <bool VEC4>
__global__ __launch_bounds__(kBlock) void assemble_rows_kernel(const AssembleArgs a) {
  const int FC = a.xb ? (VEC4 ? a.F / 4 : a.F) : 0;   // feature chunks per row (none: the caller reads x through src_node)
  const int W = FC + a.K;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= a.Nb * W) return;
  const int32_t i = (int32_t)(t / W);
  const int c = (int)(t - (int64_t)i * W);
  const int64_t gn = a.src_node[i];
  if (c < FC) {
    if (VEC4)
    {
      const float4 v = reinterpret_cast<const float4*>(a.x + gn * a.ldx)[c];
      const float r[4] = {v.x, v.y, v.z, v.w};
      vstore_nt<4>(a.xb + (int64_t)i * a.ldxb + 4 * c, r);
    }
    else
      a.xb[(int64_t)i * a.ldxb + c] = a.x[gn * a.ldx + c];
  } else {
    a.nscal_b[(int64_t)(c - FC) * a.Nb + i] = a.nscal[gn * a.K + (c - FC)];   // planar [K, Nb]: each scalar contiguous
  }
}

// ---- nodes and edges of the batch: a workgroup takes kChunk consecutive elements.
// Who owns an element (which selected graph) and where that graph sits in the arena is found ONCE per workgroup: two
// binary searches (first and last element), then the facts of the few graphs the chunk spans go to LDS.  Every thread
// then issues ALL loads of its kAsmPer elements before its first store: a thread of the one-element-per-thread form went
// through six dependent round trips (segment -> sel -> graph start -> pointers -> data -> store, and a load behind a
// store waits for that store on gfx9's single in-order vmcnt), which left the kernels latency-bound at 2.3 / 4.7 TB/s.
constexpr int kAsmPer = 4;
constexpr int kChunk = kAsmPer * kBlock;
constexpr int kSpanMax = 256;     // graphs per chunk held in LDS; chunks spanning more (tiny or empty graphs) use the slow path

struct ChunkGraphs {
  int b0, span;                   // first graph of the chunk, number of graphs it touches (0: slow path)
};

// start[j] = ptr[b0 + j] (j = 0..span), g0[j] = arena node id of graph b0 + j's first node, nb[j] = b_nptr, eb[j] = b_eptr
__device__ __forceinline__ ChunkGraphs chunk_graphs(const AssembleArgs& a, const int32_t* __restrict__ ptr, int64_t first,
                                                    int64_t total, int32_t* s_start, int32_t* s_g0, int32_t* s_nb,
                                                    int32_t* s_eb) {
  __shared__ int s_b0, s_b1;
  const int64_t last = min(first + kChunk, total) - 1;
  if (threadIdx.x == 0) s_b0 = find_segment(ptr, a.B, (int32_t)first);
  if (threadIdx.x == 64) s_b1 = find_segment(ptr, a.B, (int32_t)last);
  __syncthreads();
  ChunkGraphs c{s_b0, s_b1 - s_b0 + 1};
  if (c.span > kSpanMax) { c.span = 0; return c; }
  for (int j = threadIdx.x; j <= c.span; j += kBlock) {
    const int b = c.b0 + j;
    s_start[j] = ptr[b];                      // b <= B: ptr has B + 1 entries
    if (j < c.span) {
      s_g0[j] = a.a_gptr[a.sel[b]];
      s_nb[j] = a.b_nptr[b];
      s_eb[j] = a.b_eptr[b];
    }
  }
  __syncthreads();
  return c;
}

template <int KS>    // KS = 6: the six per-node scalars of the arena move through registers (no xb); -1: any K, one at a time
__global__ __launch_bounds__(kBlock) void assemble_nodes_kernel(const AssembleArgs a) {
  __shared__ int32_t s_start[kSpanMax + 1], s_g0[kSpanMax], s_nb[kSpanMax], s_eb[kSpanMax];
  const int64_t first = (int64_t)blockIdx.x * kChunk;
  if (first >= a.Nb) {            // the one workgroup past the nodes: the selection's own edge total closes both pointers
    if (first == ceil_div(a.Nb, (int64_t)kChunk) * kChunk && threadIdx.x == 0) {
      a.in_ptr_b[a.Nb] = a.b_eptr[a.B];
      a.out_ptr_b[a.Nb] = a.b_eptr[a.B];
    }
    return;
  }
  const ChunkGraphs c = chunk_graphs(a, a.b_nptr, first, a.Nb, s_start, s_g0, s_nb, s_eb);
  int64_t gn[kAsmPer]; int32_t g0[kAsmPer], eb[kAsmPer], shift[kAsmPer]; bool ok[kAsmPer];
  int lb = 0;
#pragma unroll
  for (int j = 0; j < kAsmPer; ++j) {
    const int64_t i = first + j * kBlock + threadIdx.x;
    ok[j] = i < a.Nb;
    const int32_t ii = (int32_t)min(i, a.Nb - 1);
    int32_t nb;
    if (c.span) {
      while (lb + 1 < c.span && ii >= s_start[lb + 1]) ++lb;
      g0[j] = s_g0[lb]; nb = s_nb[lb]; eb[j] = s_eb[lb];
    } else {
      const int b = find_segment(a.b_nptr, a.B, ii);
      g0[j] = a.a_gptr[a.sel[b]]; nb = a.b_nptr[b]; eb[j] = a.b_eptr[b];
    }
    gn[j] = (int64_t)g0[j] + (ii - nb);
    shift[j] = nb - g0[j];
  }
  // ---- loads
  int32_t ip[kAsmPer], ip0[kAsmPer], op[kAsmPer], op0[kAsmPer], lp[kAsmPer];
  int2 ie[kAsmPer], oe[kAsmPer];
  float sc[kAsmPer][KS > 0 ? KS : 1];
  const bool want_loops = a.loops_b || a.derived_b;
#pragma unroll
  for (int j = 0; j < kAsmPer; ++j) {
    ip[j] = a.a_in_ptr[gn[j]]; ip0[j] = a.a_in_ptr[g0[j]];
    op[j] = a.a_out_ptr[gn[j]]; op0[j] = a.a_out_ptr[g0[j]];
    lp[j] = want_loops ? a.a_loops[gn[j]] : 0;
    if (a.in_ell_b) ie[j] = reinterpret_cast<const int2*>(a.a_in_ell)[gn[j]];
    if (a.out_ell_b) oe[j] = reinterpret_cast<const int2*>(a.a_out_ell)[gn[j]];
    if (KS > 0) {
#pragma unroll
      for (int k = 0; k < KS; k += 2) {      // KS even, rows of KS floats: 8-byte aligned pairs
        const float2 v = *reinterpret_cast<const float2*>(a.nscal + gn[j] * KS + k);
        sc[j][k] = v.x; sc[j][k + 1] = v.y;
      }
    }
  }
  // ---- stores
#pragma unroll
  for (int j = 0; j < kAsmPer; ++j) {
    if (!ok[j]) continue;
    const int64_t i = first + j * kBlock + threadIdx.x;
    a.src_node[i] = (int32_t)gn[j];
    a.in_ptr_b[i] = ip[j] - ip0[j] + eb[j];
    a.out_ptr_b[i] = op[j] - op0[j] + eb[j];
    if (a.loops_b) a.loops_b[i] = lp[j];
    if (KS > 0) {
#pragma unroll
      for (int k = 0; k < KS; ++k) a.nscal_b[(int64_t)k * a.Nb + i] = sc[j][k];      // planar [K, Nb]
      if (a.derived_b) {    // the layers' derived per-node scalars, made here instead of by three element-wise passes per batch
        a.derived_b[i] = sc[j][0] * sc[j][0];
        a.derived_b[a.Nb + i] = (float)lp[j] * sc[j][1];
        a.derived_b[2 * a.Nb + i] = -sc[j][2];
      }
    } else if (KS < 0) {
      if (!a.xb)   // no feature rows to move: the per-node scalars ride along here instead of in a pass of their own
        for (int k = 0; k < a.K; ++k) a.nscal_b[(int64_t)k * a.Nb + i] = a.nscal[gn[j] * a.K + k];
      if (a.derived_b) {
        const float s0 = a.nscal[gn[j] * a.K], s1 = a.nscal[gn[j] * a.K + 1], s2 = a.nscal[gn[j] * a.K + 2];
        a.derived_b[i] = s0 * s0;
        a.derived_b[a.Nb + i] = (float)lp[j] * s1;
        a.derived_b[2 * a.Nb + i] = -s2;
      }
    }
    // ELL side tables: the arena's entries rebased to batch ids (-1 = no edge, bit 31 of .x = more than two edges)
    auto rebase = [&](int2 e) {
      if (e.x != -1) e.x = (((e.x & 0x7fffffff) + shift[j]) | (e.x & (int32_t)0x80000000));
      if (e.y != -1) e.y += shift[j];
      return e;
    };
    if (a.in_ell_b) reinterpret_cast<int2*>(a.in_ell_b)[i] = rebase(ie[j]);
    if (a.out_ell_b) reinterpret_cast<int2*>(a.out_ell_b)[i] = rebase(oe[j]);
  }
}

__global__ __launch_bounds__(kBlock) void assemble_edges_kernel(const AssembleArgs a) {
  __shared__ int32_t s_start[kSpanMax + 1], s_g0[kSpanMax], s_nb[kSpanMax], s_eb[kSpanMax];
  __shared__ int32_t s_in0[kSpanMax], s_out0[kSpanMax];
  const int64_t total = min(a.Eb, (int64_t)a.b_eptr[a.B]);   // Eb may be a capacity (fixed-shape launches): the real total is on the device
  const int64_t first = (int64_t)blockIdx.x * kChunk;
  if (first >= total) return;
  const ChunkGraphs c = chunk_graphs(a, a.b_eptr, first, total, s_start, s_g0, s_nb, s_eb);
  if (c.span) {
    for (int j = threadIdx.x; j < c.span; j += kBlock) { s_in0[j] = a.a_in_ptr[s_g0[j]]; s_out0[j] = a.a_out_ptr[s_g0[j]]; }
    __syncthreads();
  }
  int32_t shift[kAsmPer], eoff[kAsmPer]; int64_t in_at[kAsmPer], out_at[kAsmPer]; bool ok[kAsmPer];
  int lb = 0;
#pragma unroll
  for (int j = 0; j < kAsmPer; ++j) {
    const int64_t e = first + j * kBlock + threadIdx.x;
    ok[j] = e < total;
    const int32_t ee = (int32_t)min(e, total - 1);
    int32_t g0, nb, eb, in0, out0;
    if (c.span) {
      while (lb + 1 < c.span && ee >= s_start[lb + 1]) ++lb;       // empty graphs in between are stepped over
      g0 = s_g0[lb]; nb = s_nb[lb]; eb = s_eb[lb]; in0 = s_in0[lb]; out0 = s_out0[lb];
    } else {
      const int b = find_segment(a.b_eptr, a.B, ee);
      g0 = a.a_gptr[a.sel[b]]; nb = a.b_nptr[b]; eb = a.b_eptr[b]; in0 = a.a_in_ptr[g0]; out0 = a.a_out_ptr[g0];
    }
    shift[j] = nb - g0;
    in_at[j] = (int64_t)in0 + (ee - eb);
    out_at[j] = (int64_t)out0 + (ee - eb);
    eoff[j] = eb - in0;       // in-CSR position of an edge: rebased from the graph's slice of the arena to its slice of the batch
  }
  int32_t src[kAsmPer], dst[kAsmPer], eid[kAsmPer];
#pragma unroll
  for (int j = 0; j < kAsmPer; ++j) {
    src[j] = a.a_in_src[in_at[j]];
    dst[j] = a.a_out_dst[out_at[j]];
    eid[j] = a.out_eid_b ? a.a_out_eid[out_at[j]] : 0;
  }
#pragma unroll
  for (int j = 0; j < kAsmPer; ++j) {
    if (!ok[j]) continue;
    const int64_t e = first + j * kBlock + threadIdx.x;
    a.in_src_b[e] = src[j] + shift[j];
    a.out_dst_b[e] = dst[j] + shift[j];
    if (a.out_eid_b) a.out_eid_b[e] = eid[j] + eoff[j];
  }
}

}  // namespace mlqem

using namespace mlqem;

extern "C" size_t mlqem_csr_build_workspace_bytes(int64_t N, int64_t E) {
  if (N < 0 || E < 0) return 0;
  const SortPlan p = plan_sort(N, E > 0 ? E : 1);
  return 6 * p.keys_bytes + p.temp_bytes;
}

extern "C" int mlqem_csr_build(const int64_t* edge_index, int64_t E, int64_t N, int32_t* in_ptr, int32_t* in_src,
                               int32_t* out_ptr, int32_t* out_dst, int32_t* out_eid, int32_t* loops, void* workspace,
                               size_t workspace_bytes, mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N < 0 || E < 0 || N >= 0x7fffffffLL || E >= 0x7fffffffLL) return MLQEM_ERR_BAD_ARG;
  if (!in_ptr || !out_ptr || !loops) return MLQEM_ERR_BAD_ARG;
  if (hipMemsetAsync(loops, 0, sizeof(int32_t) * (size_t)N, stream) != hipSuccess && N > 0) return MLQEM_ERR_LAUNCH;
  if (E == 0) {
    (void)hipMemsetAsync(in_ptr, 0, sizeof(int32_t) * (size_t)(N + 1), stream);
    (void)hipMemsetAsync(out_ptr, 0, sizeof(int32_t) * (size_t)(N + 1), stream);
    return launch_status();
  }
  if (!edge_index || !in_src || !out_dst) return MLQEM_ERR_BAD_ARG;
  const SortPlan p = plan_sort(N, E);
  if (!workspace || workspace_bytes < 6 * p.keys_bytes + p.temp_bytes) return MLQEM_ERR_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  int32_t* key_d = reinterpret_cast<int32_t*>(ws);
  int32_t* key_s = reinterpret_cast<int32_t*>(ws + p.keys_bytes);
  int32_t* eid = reinterpret_cast<int32_t*>(ws + 2 * p.keys_bytes);
  int32_t* eid_sorted = reinterpret_cast<int32_t*>(ws + 3 * p.keys_bytes);
  int32_t* sorted = reinterpret_cast<int32_t*>(ws + 4 * p.keys_bytes);
  int32_t* pos_in = reinterpret_cast<int32_t*>(ws + 5 * p.keys_bytes);
  void* temp = ws + 6 * p.keys_bytes;
  size_t temp_bytes = p.temp_bytes;

  const unsigned eb = (unsigned)ceil_div(E, kBlock), nb = (unsigned)ceil_div(N + 1, kBlock);
  hipLaunchKernelGGL(split_edges_kernel, dim3(eb), dim3(kBlock), 0, stream, edge_index, E, (int32_t)N, key_d, key_s,
                     eid, loops);
  // LSD radix sort is stable: inside a row the caller's edge order survives
  if (rocprim::radix_sort_pairs(temp, temp_bytes, key_d, sorted, eid, eid_sorted, (size_t)E, 0, (unsigned)p.end_bit,
                                stream) != hipSuccess)
    return MLQEM_ERR_LAUNCH;
  hipLaunchKernelGGL(gather_in_kernel, dim3(eb), dim3(kBlock), 0, stream, edge_index, E, eid_sorted, in_src, pos_in);
  hipLaunchKernelGGL(lower_bound_kernel, dim3(nb), dim3(kBlock), 0, stream, sorted, E, N, in_ptr);
  if (rocprim::radix_sort_pairs(temp, temp_bytes, key_s, sorted, eid, eid_sorted, (size_t)E, 0, (unsigned)p.end_bit,
                                stream) != hipSuccess)
    return MLQEM_ERR_LAUNCH;
  hipLaunchKernelGGL(gather_out_kernel, dim3(eb), dim3(kBlock), 0, stream, edge_index, E, eid_sorted, pos_in, out_dst,
                     out_eid);
  hipLaunchKernelGGL(lower_bound_kernel, dim3(nb), dim3(kBlock), 0, stream, sorted, E, N, out_ptr);
  return launch_status();
}

extern "C" int mlqem_graph_norms(const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* loops, int64_t N,
                                 float* gcn_dinv, float* sage_rinv, float* cheb_dinv, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || (N > 0 && (!in_ptr || (cheb_dinv && !out_ptr)))) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  hipLaunchKernelGGL(graph_norms_kernel, dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, as_stream(stream),
                     in_ptr, out_ptr, loops, N, gcn_dinv, sage_rinv, cheb_dinv);
  return launch_status();
}

extern "C" int mlqem_batch_assemble(const float* x, int64_t ldx, int F, const float* nscal, int K,
                                    const int32_t* a_gptr, const int32_t* a_in_ptr, const int32_t* a_in_src,
                                    const int32_t* a_out_ptr, const int32_t* a_out_dst, const int32_t* a_out_eid,
                                    const int32_t* a_loops, const int32_t* a_in_ell, const int32_t* a_out_ell,
                                    const int32_t* sel, const int32_t* b_nptr, const int32_t* b_eptr, int64_t B,
                                    int64_t Nb, int64_t Eb, float* xb, int64_t ldxb, float* nscal_b, float* derived_b,
                                    int32_t* src_node, int32_t* in_ptr_b, int32_t* in_src_b, int32_t* out_ptr_b, int32_t* out_dst_b,
                                    int32_t* out_eid_b, int32_t* loops_b, int32_t* in_ell_b, int32_t* out_ell_b,
                                    mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (B <= 0 || Nb < 0 || Eb < 0 || F <= 0 || K < 0 || ldx < F || (xb && ldxb < F)) return MLQEM_ERR_BAD_ARG;
  if (B > 0x7fffffff || Nb >= 0x7fffffffLL || Eb >= 0x7fffffffLL) return MLQEM_ERR_UNSUPPORTED;
  if (!x || !a_gptr || !a_in_ptr || !a_out_ptr || !sel || !b_nptr || !b_eptr || !src_node || !in_ptr_b || !out_ptr_b)
    return MLQEM_ERR_BAD_ARG;
  if (K > 0 && (!nscal || !nscal_b)) return MLQEM_ERR_BAD_ARG;
  if (derived_b && (K < 3 || !a_loops)) return MLQEM_ERR_BAD_ARG;
  if (Eb > 0 && (!a_in_src || !a_out_dst || !in_src_b || !out_dst_b)) return MLQEM_ERR_BAD_ARG;
  if (loops_b && !a_loops) return MLQEM_ERR_BAD_ARG;
  if (out_eid_b && !a_out_eid) return MLQEM_ERR_BAD_ARG;
  if ((in_ell_b && !a_in_ell) || (out_ell_b && !a_out_ell)) return MLQEM_ERR_BAD_ARG;
  AssembleArgs a{x, ldx, F, nscal, K, a_gptr, a_in_ptr, a_in_src, a_out_ptr, a_out_dst, a_out_eid, a_loops, sel, b_nptr, b_eptr,
                 (int)B, Nb, Eb, xb, ldxb, nscal_b, derived_b, src_node, in_ptr_b, in_src_b, out_ptr_b, out_dst_b, out_eid_b, loops_b,
                 a_in_ell, a_out_ell, in_ell_b, out_ell_b};
  // one workgroup past the last node chunk writes the closing pointer entries
  const dim3 ngrid((unsigned)(ceil_div(Nb, (int64_t)kChunk) + 1));
  if (!xb && K == 6 && aligned_to(nscal, 8)) hipLaunchKernelGGL(assemble_nodes_kernel<6>, ngrid, dim3(kBlock), 0, stream, a);
  else hipLaunchKernelGGL(assemble_nodes_kernel<-1>, ngrid, dim3(kBlock), 0, stream, a);
  if (Nb > 0) {
    // xb == NULL: only the per-node scalars are gathered; the caller's first layers read x through src_node
    const bool vec4 = xb && F % 4 == 0 && ldx % 4 == 0 && ldxb % 4 == 0 && aligned_to(x, 16) && aligned_to(xb, 16);
    const int64_t per_row = xb ? (vec4 ? F / 4 : F) + K : 0;   // without xb the nodes kernel has moved the scalars
    if (per_row > 0) {
      if (vec4)
        hipLaunchKernelGGL(assemble_rows_kernel<true>, dim3((unsigned)ceil_div(Nb * per_row, kBlock)), dim3(kBlock), 0,
                           stream, a);
      else
        hipLaunchKernelGGL(assemble_rows_kernel<false>, dim3((unsigned)ceil_div(Nb * per_row, kBlock)), dim3(kBlock), 0,
                           stream, a);
    }
  }
  if (Eb > 0)
    hipLaunchKernelGGL(assemble_edges_kernel, dim3((unsigned)ceil_div(Eb, (int64_t)kChunk)), dim3(kBlock), 0, stream, a);
  return launch_status();
}

namespace mlqem {
struct GatherRowsArgs { const float* src[4]; float* dst[4]; int64_t width[4]; int64_t end[4]; int count; const int32_t* sel; int64_t B; };
// element e of the concatenation [B x width[0] | B x width[1] | ...]: which matrix, which row of the batch, which column
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const GatherRowsArgs a) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= a.end[a.count - 1]) return;
  int k = 0;
  while (e >= a.end[k]) ++k;                 // <= 3 steps
  const int64_t local = e - (k ? a.end[k - 1] : 0);
  const int64_t row = local / a.width[k], col = local - row * a.width[k];
  a.dst[k][local] = a.src[k][(int64_t)a.sel[row] * a.width[k] + col];
}
}  // namespace mlqem

extern "C" int mlqem_gather_rows_f32(int count, const float* const* src, const int64_t* width, const int32_t* sel, int64_t B,
                                     float* const* dst, mlqem_stream_t stream) {
  begin_launches();
  if (count < 1 || count > 4 || B < 0 || !src || !width || !dst) return MLQEM_ERR_BAD_ARG;
  if (B == 0) return MLQEM_OK;
  if (!sel) return MLQEM_ERR_BAD_ARG;
  mlqem::GatherRowsArgs a{};
  int64_t total = 0;
  for (int k = 0; k < count; ++k) {
    if (width[k] < 0 || (width[k] > 0 && (!src[k] || !dst[k]))) return MLQEM_ERR_BAD_ARG;
    a.src[k] = src[k]; a.dst[k] = dst[k]; a.width[k] = std::max<int64_t>(width[k], 1);
    total += B * width[k];
    a.end[k] = total;
  }
  if (total == 0) return MLQEM_OK;
  a.count = count; a.sel = sel; a.B = B;
  hipLaunchKernelGGL(mlqem::gather_rows_kernel, dim3((unsigned)mlqem::ceil_div(total, (int64_t)mlqem::kBlock)), dim3(mlqem::kBlock), 0,
                     mlqem::as_stream(stream), a);
  return mlqem::launch_status();
}
