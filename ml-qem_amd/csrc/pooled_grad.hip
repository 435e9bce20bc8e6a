// The first transposed aggregation of a Family A branch's backward with its SOURCE SYNTHESISED (round 5).
//
// The last conv of a branch is folded into its mean pool (01_ngem.ipynb cell [9]), so the gradient of the branch's last
// hidden activation is, per node j of graph b,
//     g[j, c] = gate(j, c) * gate_scale * (g_mean[b, c] + t_j g_wmean[b, c]) / n_b
// -- 12 sign bits, one structural scalar and two [B, C] rows.  pool.hip's pool_bwd_tiles_kernel wrote that out as an [N, C] matrix
// (48 bytes a node) for the transposed aggregation to gather 16 bytes at a time; here the aggregation computes the value of a source
// from what defines it: per source two structural scalars (t_j and the conv's column scale of j) and the 2-byte gate the pooled
// forward leaves per node (csr_aggregate.hip PoolFuse::node_gate), per row the two gradient rows of its graph (an edge never leaves a
// graph).  The row's own g is stored on the way for the dense consumers (weight and bias gradients), so the matrix is written once and
// never gathered.  Arithmetic and order of pool_bwd_tiles_kernel followed by csr_aggregate_ell_kernel<4, false, 2, false>: bit-equal
// (tests/test_gpu_family_a.py).
#include "common.hpp"

namespace mlqem {

struct PooledGradArgs {
  const uint16_t* gate;      // [N] PoolFuse::node_gate
  const float* wts;          // [N] pool weights t_j
  const float* cscale;       // [N] column scale of the aggregation
  const int4* tile_info;     // [tiles] (graph of the tile's first row, that graph's first row, the next graph's, 0): the forward's
  const float* g0; int64_t ldg0;      // gradient of the pooled mean [B, >= 4 CV] (may be NULL)
  const float* g1; int64_t ldg1;      // ... of the weighted mean
  const int32_t* gptr; int B; float gate_scale;
  const int32_t* ptr; const int32_t* idx; const int32_t* ell;
  const float* rscale; const float* dself; float alpha;
  float* out; int64_t ldo;   // alpha (rscale * sum_e cscale[j] g[j] + dself * g[row])
  float* g; int64_t ldg;     // the synthesised rows themselves (optional)
  int64_t N; int CV; int R;
};

constexpr int kGradHeavy = 32;         // csr_aggregate.hip kHeavyDegree

struct GradSrc { float2 rec; unsigned gate; };      // rec = (t_j, column scale of j)
__device__ __forceinline__ GradSrc grad_src(const PooledGradArgs& a, int j) {
  return GradSrc{make_float2(a.wts[j], a.cscale[j]), (unsigned)a.gate[j]};
}
// pool.hip pool_bwd_tiles_kernel's expression, for slice `ch / 4` of the source
__device__ __forceinline__ void grad_value(const PooledGradArgs& a, const GradSrc& s, int ch, const float (&a0)[4], const float (&a1)[4], float inv,
                                           float (&val)[4]) {
  const unsigned bits = s.gate >> ch;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const float u = fmaf(s.rec.x, a1[v], a0[v]) * inv;
    val[v] = (bits >> v & 1u) ? u * a.gate_scale : 0.f;
  }
}

// (72 registers, seven waves per SIMD: pinned at 64 like the gathering kernel it spilled 8 and ran 410-430 us instead of 330-360)
template <int IPT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(7, 8))) void pooled_grad_aggregate_kernel(const PooledGradArgs a) {
  constexpr int VEC = 4;
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)blk * a.R;
  const int nrows = (int)min((int64_t)a.R, a.N - r0);
  const int n_local = nrows * a.CV;
  const unsigned magic = ((1u << 20) + a.CV - 1) / a.CV;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const bool use_self = a.dself != nullptr;
  const int4 ti = a.tile_info[blk];
  const bool one_graph = r0 + nrows <= (int64_t)ti.z;      // workgroup-uniform
  const float inv0 = 1.f / (float)max(ti.z - ti.y, 1);

  int row[IPT], ch[IPT], gr[IPT];
  int2 e2[IPT];
  float rs[IPT], ds[IPT], inv[IPT], a0[IPT][VEC], a1[IPT][VEC];
  bool live[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int li = k * kBlock + tid;
    live[k] = li < n_local;
    const int lj = live[k] ? li : 0;
    const int lrow = (int)(((unsigned)lj * magic) >> 20);
    row[k] = (int)r0 + lrow;
    ch[k] = (lj - lrow * a.CV) * VEC;
    e2[k] = reinterpret_cast<const int2*>(a.ell)[row[k]];
    rs[k] = a.rscale ? a.rscale[row[k]] : 1.f;
    ds[k] = a.dself ? a.dself[row[k]] : 0.f;
    int g = ti.x;
    inv[k] = inv0;
    if (!one_graph && row[k] >= ti.z) {
      g = graph_at(a.gptr, a.B, row[k]);
      inv[k] = 1.f / (float)max(a.gptr[g + 1] - a.gptr[g], 1);
    }
    gr[k] = g;
#pragma unroll
    for (int v = 0; v < VEC; ++v) a0[k][v] = a1[k][v] = 0.f;
    if (a.g0) vload<VEC>(a.g0 + (int64_t)g * a.ldg0 + ch[k], a0[k]);
    if (a.g1) vload<VEC>(a.g1 + (int64_t)g * a.ldg1 + ch[k], a1[k]);
  }
  GradSrc q0[IPT], q1[IPT], qs[IPT];
  bool has0[IPT], has1[IPT], more[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    more[k] = (e2[k].x & kEllMore) != 0 && e2[k].x != -1;
    const int s0 = e2[k].x == -1 ? -1 : (e2[k].x & ~kEllMore), s1 = e2[k].y;
    has0[k] = s0 >= 0; has1[k] = s1 >= 0;
    q0[k] = grad_src(a, has0[k] ? s0 : row[k]);      // a missing edge reads the row itself, with weight 0
    q1[k] = grad_src(a, has1[k] ? s1 : row[k]);
    qs[k] = grad_src(a, row[k]);
  }
  float acc[IPT][VEC], self[IPT][VEC];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    float v0[VEC], v1[VEC];
    grad_value(a, q0[k], ch[k], a0[k], a1[k], inv[k], v0);
    grad_value(a, q1[k], ch[k], a0[k], a1[k], inv[k], v1);
    grad_value(a, qs[k], ch[k], a0[k], a1[k], inv[k], self[k]);
    const float w0 = has0[k] ? q0[k].rec.y : 0.f, w1 = has1[k] ? q1[k].rec.y : 0.f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[k][v] = fmaf(w1, v1[v], w0 * v0[v]);
  }
  bool heavy[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    heavy[k] = false;
    if (!live[k]) continue;
    if (a.g) vstore_nt<VEC>(a.g + (int64_t)row[k] * a.ldg + ch[k], self[k]);
    if (more[k]) {      // rows of more than two entries walk the CSR arrays from the third on
      const int beg = a.ptr[row[k]], end = a.ptr[row[k] + 1];
      if (end - beg > kGradHeavy) {      // a hub row: the wave-cooperative pass below
        heavy[k] = ch[k] == 0;
        continue;
      }
      for (int e = beg + 2; e < end; ++e) {
        const GradSrc q = grad_src(a, a.idx[e]);
        float r[VEC];
        grad_value(a, q, ch[k], a0[k], a1[k], inv[k], r);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[k][v] = fmaf(q.rec.y, r[v], acc[k][v]);
      }
    }
    float res[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) res[v] = a.alpha * fmaf(ds[k], use_self ? self[k][v] : 0.f, rs[k] * acc[k][v]);
    vstore_nt<VEC>(a.out + (int64_t)row[k] * a.ldo + ch[k], res);
  }
  // Hub rows (barrier nodes), one at a time, by the wave that owns the row's slice-0 item: csr_aggregate_ell_kernel's pass (lanes =
  // edge slot x slice, 64 source ids per coalesced read, a fixed shuffle tree over the slots)
  const int nslots = kWave / a.CV;                        // CV <= 4
  const int slot = lane / a.CV, hch = (lane - slot * a.CV) * VEC;
  const bool worker = slot < nslots;
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    unsigned long long todo = __ballot(heavy[k]);
    while (todo) {
      const int owner = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int r = __shfl(row[k], owner), g = __shfl(gr[k], owner);
      const float rs_r = __shfl(rs[k], owner), ds_r = __shfl(ds[k], owner), inv_r = __shfl(inv[k], owner);
      const int beg = a.ptr[r], end = a.ptr[r + 1];
      float b0[VEC], b1[VEC], part[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) b0[v] = b1[v] = part[v] = 0.f;
      if (worker) {
        if (a.g0) vload<VEC>(a.g0 + (int64_t)g * a.ldg0 + hch, b0);
        if (a.g1) vload<VEC>(a.g1 + (int64_t)g * a.ldg1 + hch, b1);
      }
      for (int base = beg; base < end; base += kWave) {
        const int my_src = base + lane < end ? a.idx[base + lane] : 0;
        const int in_chunk = min(kWave, end - base);
        const int rounds = (in_chunk + nslots - 1) / nslots;
        for (int rd = 0; rd < rounds; ++rd) {
          const int el = slot + rd * nslots;
          const int j = __shfl(my_src, el < kWave ? el : 0);
          if (worker && el < in_chunk) {
            const GradSrc q = grad_src(a, j);
            float qv[VEC];
            grad_value(a, q, hch, b0, b1, inv_r, qv);
#pragma unroll
            for (int v = 0; v < VEC; ++v) part[v] = fmaf(q.rec.y, qv[v], part[v]);
          }
        }
      }
      for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float other = __shfl_down(part[v], off * a.CV);
          if (worker && slot + off < nslots) part[v] += other;
        }
      }
      if (worker && slot == 0) {
        float sf[VEC], res[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) sf[v] = 0.f;
        if (use_self) grad_value(a, grad_src(a, r), hch, b0, b1, inv_r, sf);
#pragma unroll
        for (int v = 0; v < VEC; ++v) res[v] = a.alpha * fmaf(ds_r, sf[v], rs_r * part[v]);
        vstore_nt<VEC>(a.out + (int64_t)r * a.ldo + hch, res);
      }
    }
  }
}

// Column sums of the same gradient (the bias gradient of the layer whose aggregation computed it without writing it): a workgroup
// walks a contiguous range of rows, thread = (row lane, slice) with four running sums, six bytes read per node; the row lanes of a
// slice are added in order through LDS, the workgroups by the caller (fixed shapes: deterministic).
constexpr int kColsumGroups = 2048;
__global__ __launch_bounds__(kBlock) void pooled_grad_colsum_kernel(const uint16_t* __restrict__ gate, const float* __restrict__ wts,
                                                                    const float* __restrict__ g0, int64_t ldg0, const float* __restrict__ g1,
                                                                    int64_t ldg1, const int32_t* __restrict__ gptr, int B, float gate_scale,
                                                                    int64_t N, int CV, float* __restrict__ partial) {
  __shared__ float s_sum[kBlock][4];
  const int lanes = kBlock / CV;                            // rows in flight per pass
  const int tid = threadIdx.x, rl = tid / CV, ch = (tid - rl * CV) * 4;
  const int64_t per = ceil_div(N, (int64_t)gridDim.x);
  const int64_t beg = (int64_t)blockIdx.x * per, end = min(N, beg + per);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (rl < lanes && beg + rl < end) {
    int g = graph_at(gptr, B, beg + rl), gend = gptr[g + 1];
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, inv = 1.f / (float)max(gend - gptr[g], 1);
    if (g0) vload<4>(g0 + (int64_t)g * ldg0 + ch, a0);
    if (g1) vload<4>(g1 + (int64_t)g * ldg1 + ch, a1);
    constexpr int kAhead = 8;                                // rows in flight per thread: the loop is a chain of dependent loads otherwise
    for (int64_t r = beg + rl; r < end; r += (int64_t)kAhead * lanes) {
      unsigned gt[kAhead]; float tw[kAhead];
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const int64_t ru = r + (int64_t)u * lanes;
        gt[u] = ru < end ? (unsigned)gate[ru] : 0u;
        tw[u] = ru < end ? wts[ru] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const int64_t ru = r + (int64_t)u * lanes;
        if (ru >= end) break;
        if (ru >= gend) {                                    // the next graphs in turn: a thread's rows are `lanes` apart, a search from
          int gbeg = gend;                                   // scratch (ten dependent loads) per row was the kernel's time on small graphs
          do { ++g; gbeg = gend; gend = gptr[g + 1]; } while (ru >= gend && g + 1 < B);
          inv = 1.f / (float)max(gend - gbeg, 1);
          if (g0) vload<4>(g0 + (int64_t)g * ldg0 + ch, a0);
          if (g1) vload<4>(g1 + (int64_t)g * ldg1 + ch, a1);
        }
        const unsigned bits = gt[u] >> ch;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float uu = fmaf(tw[u], a1[v], a0[v]) * inv;
          acc[v] += (bits >> v & 1u) ? uu * gate_scale : 0.f;
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) s_sum[tid][v] = acc[v];
  __syncthreads();
  // the row lanes of a slice, halved until one is left: a fixed tree
  for (int n = lanes; n > 1;) {
    const int h = (n + 1) >> 1;
    if (rl < n - h) {
#pragma unroll
      for (int v = 0; v < 4; ++v) s_sum[tid][v] += s_sum[tid + h * CV][v];
    }
    __syncthreads();
    n = h;
  }
  if (tid < CV * 4) partial[(int64_t)blockIdx.x * (CV * 4) + tid] = s_sum[tid >> 2][tid & 3];
}

int aggregate_pool_rows_per_tile(int C);
int aggregate_pool_mask_words();

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_pooled_grad_aggregate_supported(int C) { return C > 0 && (C + 3) / 4 <= 4 ? 1 : 0; }

extern "C" int mlqem_pooled_grad_aggregate_f32(const uint8_t* gate_bits, const float* weights, const float* cscale, const float* g_mean, int64_t ld_gmean,
                                               const float* g_wmean, int64_t ld_gwmean, const int32_t* graph_ptr, int64_t B,
                                               float gate_scale, const int32_t* ptr, const int32_t* idx, const int32_t* ell,
                                               const float* rscale, const float* dself, float alpha, float* out, int64_t ldo, float* g,
                                               int64_t ldg, int64_t N, int C, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || B < 0 || C <= 0 || B > 0x7fffffffLL || N > 0x7fffffffLL) return MLQEM_ERR_BAD_ARG;
  if (!mlqem_pooled_grad_aggregate_supported(C)) return MLQEM_ERR_UNSUPPORTED;
  if (N == 0) return MLQEM_OK;
  const int c4 = (C + 3) / 4 * 4, cv = c4 / 4;
  auto rows_ok = [&](const float* p, int64_t ld) { return !p || (ld >= c4 && ld % 4 == 0 && aligned_to(p, 16)); };
  if (!gate_bits || !weights || !cscale || !graph_ptr || !ptr || !idx || !ell || !out || B == 0 || (!g_mean && !g_wmean)) return MLQEM_ERR_BAD_ARG;
  if (!rows_ok(out, ldo) || !rows_ok(g, ldg) || !rows_ok(g_mean, ld_gmean) || !rows_ok(g_wmean, ld_gwmean) || !aligned_to(gate_bits, 16))
    return MLQEM_ERR_BAD_ARG;
  const int R = aggregate_pool_rows_per_tile(C);
  const int64_t tiles = ceil_div(N, (int64_t)R);
  const int4* info = reinterpret_cast<const int4*>(gate_bits);
  const unsigned long long* words = reinterpret_cast<const unsigned long long*>(info + tiles);
  const uint16_t* node_gate = reinterpret_cast<const uint16_t*>(words + tiles * aggregate_pool_mask_words());
  const PooledGradArgs a{node_gate, weights, cscale, info, g_mean, ld_gmean, g_wmean, ld_gwmean, graph_ptr, (int)B,
                         gate_scale, ptr, idx, ell, rscale, dself, alpha, out, ldo, g, ldg, N, cv, R};
  hipLaunchKernelGGL(pooled_grad_aggregate_kernel<2>, dim3((unsigned)tiles), dim3(kBlock), 0, as_stream(stream), a);
  return launch_status();
}

// partial: [mlqem_pooled_grad_colsum_groups(N)][round_up(C, 4)] floats, to be added over the groups by the caller -> sum_j g[j, :]
extern "C" int mlqem_pooled_grad_colsum_groups(int64_t N) {      // a trip or more of eight rows per thread, at most kColsumGroups workgroups
  return (int)std::min<int64_t>(kColsumGroups, std::max<int64_t>(1, ceil_div(std::max<int64_t>(N, 1), (int64_t)512)));
}
extern "C" int mlqem_pooled_grad_colsum_f32(const uint8_t* gate_bits, const float* weights, const float* g_mean, int64_t ld_gmean,
                                            const float* g_wmean, int64_t ld_gwmean, const int32_t* graph_ptr, int64_t B, float gate_scale,
                                            int64_t N, int C, float* partial, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || B <= 0 || C <= 0 || B > 0x7fffffffLL || N > 0x7fffffffLL) return MLQEM_ERR_BAD_ARG;
  if (!mlqem_pooled_grad_aggregate_supported(C)) return MLQEM_ERR_UNSUPPORTED;
  const int c4 = (C + 3) / 4 * 4, cv = c4 / 4;
  auto rows_ok = [&](const float* p, int64_t ld) { return !p || (ld >= c4 && ld % 4 == 0 && aligned_to(p, 16)); };
  if (!gate_bits || !weights || !graph_ptr || !partial || (!g_mean && !g_wmean) || !rows_ok(g_mean, ld_gmean) || !rows_ok(g_wmean, ld_gwmean) ||
      !aligned_to(gate_bits, 16))
    return MLQEM_ERR_BAD_ARG;
  const int R = aggregate_pool_rows_per_tile(C);
  const int64_t tiles = ceil_div(std::max<int64_t>(N, 1), (int64_t)R);
  const int4* info = reinterpret_cast<const int4*>(gate_bits);
  const unsigned long long* words = reinterpret_cast<const unsigned long long*>(info + tiles);
  const uint16_t* node_gate = reinterpret_cast<const uint16_t*>(words + tiles * aggregate_pool_mask_words());
  hipLaunchKernelGGL(pooled_grad_colsum_kernel, dim3((unsigned)mlqem_pooled_grad_colsum_groups(N)), dim3(kBlock), 0, as_stream(stream), node_gate, weights, g_mean, ld_gmean, g_wmean, ld_gwmean, graph_ptr, (int)B, gate_scale, N, cv, partial);
  return launch_status();
}
