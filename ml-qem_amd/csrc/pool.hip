// Pooling over the contiguous node range of each graph (global_mean_pool, docs/tutorials/gnn.py:114; 01_ngem.ipynb
// cell [9]) and its backward -- with an optional per-node weight, which is what the last conv layer of a Family A
// branch collapses to:  mean_pool(P (h W^T)) = (1/n_g) sum_j t_j (h_j W^T)  with  t = P^T 1  (column sums of the
// propagation matrix P: a structural per-node scalar), because nothing non-linear sits between that conv and the pool.
//
// Work decomposition.  Graphs on this path have 7 ... 20,711 nodes, so a workgroup per graph is either starved or
// alone with 20k rows (round 1: 5 % of the HBM peak).  Here the ROWS are tiled: a 256-thread workgroup owns kPoolRows
// consecutive rows whatever graphs they belong to, walks the (tile, graph) segments inside its tile -- one or two for
// 100-qubit graphs, a handful for 4-qubit graphs -- and reduces each with 16-byte row loads, several rows in flight per
// thread, then an LDS tree in a fixed order.  A segment's partial sum goes to slot (tile + graph): both indices are
// monotone along the row axis, so the slot is unique and < tiles + graphs.  A second tiny kernel adds a graph's slots in
// tile order and divides by the node count: deterministic, no atomics.
#include "common.hpp"

namespace mlqem {

constexpr int kPoolRows = 1024;   // rows per workgroup (graphs of hundreds to thousands of rows)
constexpr int kPoolRowsSmall = 64;   // ... when the batch's graphs average fewer than 128 rows: a tile then holds a graph or two
                                     // instead of dozens, which one workgroup would walk one after the other (three barriers each:
                                     // 40 us for the 33 pooled graphs of a 32-circuit Family B batch, whatever their size)
constexpr int kPoolUnroll = 4;    // rows a thread has in flight

struct PoolArgs {
  const float* x; int64_t ldx;
  const float* wts;          // optional [N]
  const int32_t* gptr;       // [B+1]
  int64_t N; int B; int C; int CV;   // CV = ceil(C / VEC) channel slices per row
  float* partial;            // [(tiles + B)][2][CV * VEC]
  int rows;                  // rows per tile: kPoolRows or kPoolRowsSmall
};

// LDS traffic of ONE wave on its own region (asap.hip's wave_lds_sync): the LDS pipeline serves a wave's instructions in order
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void pool_partial_kernel(const PoolArgs a) {
  __shared__ float s_red[kBlock][2 * VEC];
  const int tid = threadIdx.x;
  const int64_t tile = blockIdx.x;
  const int64_t r0 = tile * a.rows, r1 = min(a.N, r0 + a.rows);
  const int lanes_r = kBlock / a.CV;            // row lanes (>= 1: CV <= kBlock checked on the host)
  const int cs = tid % a.CV, rl = tid / a.CV;
  const bool worker = rl < lanes_r;
  const int ch = cs * VEC;
  const int q_lanes = min(16, lanes_r);         // second-level width of the LDS tree
  const int slot_w = a.CV * VEC;
  const int g_first = graph_at(a.gptr, a.B, r0);
  int g_end = g_first;
  while (g_end < a.B && (int64_t)a.gptr[g_end] < r1) ++g_end;      // graphs [g_first, g_end) touch the tile (workgroup-uniform)
  if (g_end - g_first > 4 && a.CV <= kWave) {
    // A tile of MANY short segments (the filler graphs behind a size-stable batch, batches of small circuits): a WAVE per segment, the
    // four waves side by side, each on its own quarter of the LDS with wave-level fences -- the workgroup-wide tree below pays three
    // barriers a segment, one segment after the other (55 us for the tile that holds the 32 fillers of a 64-circuit batch; the
    // whole pool takes 11 us without them).
    const int lane = tid & (kWave - 1), wv = tid / kWave;
    const int lanes_w = kWave / a.CV, cs_w = lane % a.CV, rl_w = lane / a.CV, ch_w = cs_w * VEC;
    const bool worker_w = rl_w < lanes_w;
    float (*red)[2 * VEC] = s_red + wv * kWave;
    for (int g = g_first + wv; g < g_end; g += kBlock / kWave) {
      const int64_t s0 = max(r0, (int64_t)a.gptr[g]), s1 = min(r1, (int64_t)a.gptr[g + 1]);
      if (s1 <= s0) continue;                   // an empty graph (wave-uniform)
      float acc0[VEC], acc1[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc0[v] = acc1[v] = 0.f;
      if (worker_w)
        for (int64_t r = s0 + rl_w; r < s1; r += (int64_t)lanes_w * kPoolUnroll) {
          float xv[kPoolUnroll][VEC], w[kPoolUnroll];
#pragma unroll
          for (int u = 0; u < kPoolUnroll; ++u) {
            const int64_t ru = r + (int64_t)u * lanes_w;
            const bool ok = ru < s1;
            const int64_t rr = ok ? ru : s0;
            float raw[VEC];
            vload<VEC>(a.x + rr * a.ldx + ch_w, raw);
            const float wr = a.wts ? a.wts[rr] : 1.f;
            w[u] = ok ? wr : 0.f;
#pragma unroll
            for (int v = 0; v < VEC; ++v) xv[u][v] = ok ? raw[v] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < kPoolUnroll; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              acc0[v] += xv[u][v];
              acc1[v] = fmaf(w[u], xv[u][v], acc1[v]);
            }
        }
#pragma unroll
      for (int v = 0; v < VEC; ++v) { red[lane][v] = acc0[v]; red[lane][VEC + v] = acc1[v]; }
      wave_lds_fence();
      if (worker_w && rl_w == 0) {              // the row lanes in order
        float t[2 * VEC];
#pragma unroll
        for (int v = 0; v < 2 * VEC; ++v) t[v] = 0.f;
        for (int k = 0; k < lanes_w; ++k)
#pragma unroll
          for (int v = 0; v < 2 * VEC; ++v) t[v] += red[k * a.CV + cs_w][v];
        float* dst = a.partial + ((tile + g) * 2) * slot_w + ch_w;
#pragma unroll
        for (int v = 0; v < VEC; ++v) { dst[v] = t[v]; dst[slot_w + v] = t[VEC + v]; }
      }
      wave_lds_fence();                         // the region is reused by the wave's next segment
    }
    return;
  }
  for (int g = g_first; g < g_end; ++g) {   // workgroup-uniform loop
    const int64_t s0 = max(r0, (int64_t)a.gptr[g]), s1 = min(r1, (int64_t)a.gptr[g + 1]);
    if (s1 <= s0) continue;                     // an empty graph
    float acc0[VEC], acc1[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc0[v] = acc1[v] = 0.f;
    if (worker) {
      for (int64_t r = s0 + rl; r < s1; r += (int64_t)lanes_r * kPoolUnroll) {
        float xv[kPoolUnroll][VEC], w[kPoolUnroll];
#pragma unroll
        for (int u = 0; u < kPoolUnroll; ++u) {
          const int64_t ru = r + (int64_t)u * lanes_r;
          const bool ok = ru < s1;
          const int64_t rr = ok ? ru : s0;      // a valid address; its contribution is zeroed below
          vload<VEC>(a.x + rr * a.ldx + ch, xv[u]);
          w[u] = a.wts ? a.wts[rr] : 1.f;
          if (!ok) {
            w[u] = 0.f;
#pragma unroll
            for (int v = 0; v < VEC; ++v) xv[u][v] = 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < kPoolUnroll; ++u)
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            acc0[v] += xv[u][v];
            acc1[v] = fmaf(w[u], xv[u][v], acc1[v]);
          }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) { s_red[tid][v] = acc0[v]; s_red[tid][VEC + v] = acc1[v]; }
    __syncthreads();
    // second level: row lane q < Q adds lanes q, q + Q, q + 2Q, ... in that order, then lane 0 adds the Q sums
    const bool second = worker && rl < q_lanes;
    float t2[2 * VEC];
#pragma unroll
    for (int v = 0; v < 2 * VEC; ++v) t2[v] = 0.f;
    if (second)
      for (int k = rl; k < lanes_r; k += q_lanes)
#pragma unroll
        for (int v = 0; v < 2 * VEC; ++v) t2[v] += s_red[k * a.CV + cs][v];
    __syncthreads();                            // every read of the first level is done
    if (second)
#pragma unroll
      for (int v = 0; v < 2 * VEC; ++v) s_red[tid][v] = t2[v];
    __syncthreads();
    if (rl == 0) {
      float t[2 * VEC];
#pragma unroll
      for (int v = 0; v < 2 * VEC; ++v) t[v] = 0.f;
      for (int k = 0; k < q_lanes; ++k)
#pragma unroll
        for (int v = 0; v < 2 * VEC; ++v) t[v] += s_red[k * a.CV + cs][v];
      float* dst = a.partial + ((tile + g) * 2) * slot_w + ch;
#pragma unroll
      for (int v = 0; v < VEC; ++v) { dst[v] = t[v]; dst[slot_w + v] = t[VEC + v]; }
    }
    __syncthreads();                            // s_red is reused by the next segment
  }
}

__global__ __launch_bounds__(kBlock) void pool_finish_kernel(const float* __restrict__ partial, const int32_t* __restrict__ gptr,
                                                             int B, int C, int slot_w, int rows, float* __restrict__ out_mean,
                                                             int64_t ld0, float* __restrict__ out_wmean, int64_t ld1) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= (int64_t)B * C) return;
  const int g = (int)(t / C), c = (int)(t - (int64_t)g * C);
  const int beg = gptr[g], end = gptr[g + 1];
  float s0 = 0.f, s1 = 0.f;
  if (end > beg) {
    const int64_t t_first = beg / rows, t_last = (end - 1) / rows;
    for (int64_t tile = t_first; tile <= t_last; ++tile) {
      const float* p = partial + ((tile + g) * 2) * slot_w + c;
      s0 += p[0];
      s1 += p[slot_w];
    }
    const float inv = 1.f / (float)(end - beg);
    s0 *= inv; s1 *= inv;
  }
  if (out_mean) out_mean[(int64_t)g * ld0 + c] = s0;
  if (out_wmean) out_wmean[(int64_t)g * ld1 + c] = s1;
}

// The same second stage for the pooled aggregation's tiles (170 rows each: a 100-qubit circuit spans ~65 of them, and one thread
// walking 65 strided partials per (graph, channel) took 59 us): a workgroup per graph, eight tile lanes x 32 value lanes (the
// 2 x slot_w sums of a partial), every lane adds every eighth tile in order, the eight lanes are added in order.
__global__ __launch_bounds__(kBlock) void pool_finish_tiles_kernel(const float* __restrict__ partial, const int32_t* __restrict__ gptr,
                                                                   int C, int slot_w, int rows, float* __restrict__ out_mean,
                                                                   int64_t ld0, float* __restrict__ out_wmean, int64_t ld1) {
  __shared__ float s_t[8][32];
  const int g = blockIdx.x, vl = threadIdx.x & 31, tl = threadIdx.x >> 5;
  const int beg = gptr[g], end = gptr[g + 1];
  const int which = vl / slot_w, c = vl - which * slot_w;         // 2 slot_w <= 32: checked on the host
  float t = 0.f;
  if (end > beg && vl < 2 * slot_w) {
    const int64_t t_first = beg / rows, t_last = (end - 1) / rows;
    for (int64_t tile = t_first + tl; tile <= t_last; tile += 8) t += partial[((tile + g) * 2 + which) * slot_w + c];
  }
  s_t[tl][vl] = t;
  __syncthreads();
  if (tl != 0 || vl >= 2 * slot_w || c >= C) return;
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) tot += s_t[k][vl];
  tot = end > beg ? tot / (float)(end - beg) : 0.f;
  if (which == 0) { if (out_mean) out_mean[(int64_t)g * ld0 + c] = tot; }
  else if (out_wmean) out_wmean[(int64_t)g * ld1 + c] = tot;
}

// Backward: gx[r,:] = (g_mean[g,:] + wts[r] * g_wmean[g,:]) / n_g, g = graph of row r, optionally gated by the ReLU/dropout
// mask of the activation that was pooled (gx = gate[r,:] > 0 ? gx * gate_scale : 0 -- the mask hand-over of the layer
// nodes).  Items = (row, 16-byte channel slice) numbered row-major, kPoolBwdItems per thread; the workgroup finds the
// graph of its first row once, keeps the next graph boundaries in LDS and every item walks forward from there.
constexpr int kPoolBwdItems = 4;
constexpr int kPoolPtrCache = 64;

// GATE: 0 none, 1 the activation itself (gate > 0).  (The sign-bit gate the pooled aggregation leaves has its own kernel below.)
template <int VEC, int GATE>
__global__ __launch_bounds__(kBlock) void pool_bwd_kernel(const float* __restrict__ g0, int64_t ldg0, const float* __restrict__ g1,
                                                          int64_t ldg1, const float* __restrict__ wts,
                                                          const int32_t* __restrict__ gptr, int64_t N, int B, int CV,
                                                          const float* __restrict__ gate, int64_t ldgate, float gate_scale,
                                                          float* __restrict__ gx, int64_t ldgx) {
  __shared__ int s_ptr[kPoolPtrCache + 1];
  __shared__ int s_g0;
  const int tid = threadIdx.x;
  const int64_t item0 = (int64_t)blockIdx.x * kBlock * kPoolBwdItems;
  const int64_t n_items = N * CV;
  const int64_t row_first = item0 / CV;
  if (tid == 0) s_g0 = graph_at(gptr, B, row_first);
  __syncthreads();
  const int gbase = s_g0;
  if (tid <= kPoolPtrCache) s_ptr[tid] = gptr[min(gbase + tid, B)];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPoolBwdItems; ++k) {
    const int64_t it = item0 + (int64_t)k * kBlock + tid;
    if (it >= n_items) continue;
    const int64_t r = it / CV;
    const int ch = (int)(it - r * CV) * VEC;
    int j = 0;                                   // graph = gbase + j: s_ptr[j] <= r < s_ptr[j + 1]
    while (j < kPoolPtrCache && (int64_t)s_ptr[j + 1] <= r) ++j;
    int g = gbase + j, beg, end;
    if (j < kPoolPtrCache) { beg = s_ptr[j]; end = s_ptr[j + 1]; }
    else { g = graph_at(gptr, B, r); beg = gptr[g]; end = gptr[g + 1]; }     // more than 64 graphs inside one workgroup
    const float inv = 1.f / (float)(end - beg);
    float a0[VEC], a1[VEC], gv[VEC], o[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) a0[v] = a1[v] = 0.f;
    // (the two [B, C] gradient rows staged in LDS per workgroup instead: 174 -> 211 us, measured and reverted)
    if (g0) vload<VEC>(g0 + (int64_t)g * ldg0 + ch, a0);
    if (g1) vload<VEC>(g1 + (int64_t)g * ldg1 + ch, a1);
    const float w = (g1 && wts) ? wts[r] : 1.f;
    if (GATE == 1) vload<VEC>(gate + r * ldgate + ch, gv);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float u = fmaf(w, a1[v], a0[v]) * inv;
      if (GATE == 1) u = gv[v] > 0.f ? u * gate_scale : 0.f;
      o[v] = u;
    }
    vstore_nt<VEC>(gx + r * ldgx + ch, o);
  }
}

// The same gradient gated by the SIGN BITS the pooled aggregation leaves (csr_aggregate.hip PoolFuse::mask): a workgroup takes the
// tile of rows a workgroup of that launch produced, thread t the items k 256 + t as there, so a wave's gate is four 8-byte words
// per k at a wave-uniform address (scalar loads) instead of a byte per lane, the item -> (row, slice) split is the tile-local
// multiply-shift instead of a 64-bit division per item, and the tile's graph comes from the 16-byte record the forward launch left
// next to the bits (r03: 178 us for 0.59 GB, bound by its request rate: five memory instructions per 16 bytes written).
// A tile inside ONE graph (all but one in 65 at 100 qubits) reads the graph's two gradient rows at uniform addresses as well.
#ifndef MLQEM_POOL_BWD_PLAIN
#define MLQEM_POOL_BWD_PLAIN 0     // 1: plain instead of non-temporal stores (A/B builds)
#endif
template <int CVT>
__global__ __launch_bounds__(kBlock) void pool_bwd_tiles_kernel(const float* __restrict__ g0, int64_t ldg0, const float* __restrict__ g1,
                                                                int64_t ldg1, const float* __restrict__ wts,
                                                                const int32_t* __restrict__ gptr, int64_t N, int B, int CV, int R,
                                                                float gate_scale, const int4* __restrict__ tile_info,
                                                                const unsigned long long* __restrict__ words, int words_per_tile,
                                                                float* __restrict__ gx, int64_t ldgx) {
  constexpr int VEC = 4;
  // (several tiles per workgroup with every load issued up front: 167 -> 185 us at one tile, 190 at two, 244 at four -- measured, dropped)
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)blk * R;
  const int nrows = (int)min((int64_t)R, N - r0);
  const int cv = CVT ? CVT : CV;
  const int n_local = nrows * cv;
  const unsigned magic = ((1u << 20) + cv - 1) / cv;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int4 ti = tile_info[blk];                      // (graph, its first row, the next graph's first row, -)
  const bool one_graph = r0 + nrows <= (int64_t)ti.z;  // workgroup-uniform
  const float inv0 = 1.f / (float)max(ti.z - ti.y, 1);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int li = k * kBlock + tid;
    if (li >= n_local) continue;
    const int lrow = (int)(((unsigned)li * magic) >> 20);
    const int sl = li - lrow * cv, ch = sl * VEC;
    const int64_t r = r0 + lrow;
    const unsigned long long* __restrict__ w4 = words + ((int64_t)blk * words_per_tile + (k * 4 + wid) * VEC);
    float a0[VEC], a1[VEC], o[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) a0[v] = a1[v] = 0.f;
    float inv = inv0;
    if (one_graph) {
      if (g0) vload<VEC>(g0 + (int64_t)ti.x * ldg0 + ch, a0);
      if (g1) vload<VEC>(g1 + (int64_t)ti.x * ldg1 + ch, a1);
    } else {
      int g = ti.x, beg = ti.y, end = ti.z;
      if (r >= (int64_t)end) { g = graph_at(gptr, B, r); beg = gptr[g]; end = gptr[g + 1]; }
      inv = 1.f / (float)(end - beg);
      if (g0) vload<VEC>(g0 + (int64_t)g * ldg0 + ch, a0);
      if (g1) vload<VEC>(g1 + (int64_t)g * ldg1 + ch, a1);
    }
    const float w = (g1 && wts) ? wts[r] : 1.f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float u = fmaf(w, a1[v], a0[v]) * inv;
      o[v] = ((w4[v] >> lane) & 1ull) ? u * gate_scale : 0.f;
    }
#if MLQEM_POOL_BWD_PLAIN
    vstore<VEC>(gx + r * ldgx + ch, o);
#else
    vstore_nt<VEC>(gx + r * ldgx + ch, o);
#endif
  }
}

// ---------------------------------------------------------------------------------------------- pooled head
// out[b, col(t)] += P_t[b, :] . w_t  (+ bias_col): the [B, C] x [C] products that remain of a branch's last conv once it
// has been folded into its pool (three branches, five terms in Family A).  As torch ops these are fifteen tiny launches
// per step (rocBLAS picks a 28 us kernel for a [1, B] x [B, 10] product); here the forward is one launch and the
// backward two: gP_t[b, :] = gout[b, col(t)] w_t, then gw_t = sum_b gout[b, col(t)] P_t[b, :] and gb_k = sum_b gout[b, k]
// by one workgroup per term / column with a fixed-order tree.
struct HeadArgs {
  int n_terms, n_cols, C;
  const float* P[MLQEM_HEAD_MAX_TERMS]; int64_t ldp[MLQEM_HEAD_MAX_TERMS];
  const float* W[MLQEM_HEAD_MAX_TERMS]; int col[MLQEM_HEAD_MAX_TERMS];
  const float* bias[MLQEM_HEAD_MAX_TERMS];   // per output column (may be NULL)
  int64_t B;
};

__global__ __launch_bounds__(kBlock) void pooled_head_kernel(const HeadArgs a, float* __restrict__ out, int64_t ldo) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= a.B * a.n_cols) return;
  const int64_t b = t / a.n_cols;
  const int k = (int)(t - b * a.n_cols);
  float acc = a.bias[k] ? a.bias[k][0] : 0.f;
  for (int term = 0; term < a.n_terms; ++term) {
    if (a.col[term] != k) continue;
    const float* __restrict__ p = a.P[term] + b * a.ldp[term];
    float s = 0.f;
    for (int c = 0; c < a.C; ++c) s = fmaf(p[c], a.W[term][c], s);
    acc += s;
  }
  out[b * ldo + k] = acc;
}

struct HeadGradRows { float* gP[MLQEM_HEAD_MAX_TERMS]; int64_t ld[MLQEM_HEAD_MAX_TERMS]; };

__global__ __launch_bounds__(kBlock) void pooled_head_bwd_rows_kernel(const HeadArgs a, const float* __restrict__ gout, int64_t ldg,
                                                                      const HeadGradRows o) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t per = a.B * a.C;
  if (t >= per * a.n_terms) return;
  const int term = (int)(t / per);
  const int64_t r = t - (int64_t)term * per;
  const int64_t b = r / a.C;
  const int c = (int)(r - b * a.C);
  o.gP[term][b * o.ld[term] + c] = gout[b * ldg + a.col[term]] * a.W[term][c];
}

// blockIdx.x < n_terms: gw of that term; otherwise gb of column blockIdx.x - n_terms
__global__ __launch_bounds__(kBlock) void pooled_head_bwd_sums_kernel(const HeadArgs a, const float* __restrict__ gout, int64_t ldg,
                                                                      float* __restrict__ gW, float* __restrict__ gb) {
  __shared__ float s_red[kBlock];
  const int tid = threadIdx.x;
  const bool is_w = (int)blockIdx.x < a.n_terms;
  const int term = is_w ? blockIdx.x : 0, k = is_w ? a.col[term] : (int)blockIdx.x - a.n_terms;
  const int n_out = is_w ? a.C : 1;
  for (int c = 0; c < n_out; ++c) {
    float s = 0.f;
    for (int64_t b = tid; b < a.B; b += kBlock) {
      const float g = gout[b * ldg + k];
      s += is_w ? g * a.P[term][b * a.ldp[term] + c] : g;
    }
    s_red[tid] = s;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {       // fixed tree: deterministic
      if (tid < off) s_red[tid] += s_red[tid + off];
      __syncthreads();
    }
    if (tid == 0) { if (is_w) gW[term * a.C + c] = s_red[0]; else gb[k] = s_red[0]; }
    __syncthreads();
  }
}

static int pool_rows(int64_t N, int64_t B) { return (B > 0 && N / B < 128) ? kPoolRowsSmall : kPoolRows; }

}  // namespace mlqem

using namespace mlqem;

extern "C" size_t mlqem_segment_pool_workspace_bytes(int64_t N, int64_t B, int C) {
  if (N < 0 || B < 0 || C <= 0) return 0;
  const int64_t tiles = ceil_div(std::max<int64_t>(N, 1), pool_rows(N, B));
  return (size_t)(tiles + B) * 2 * ((C + 3) / 4 * 4) * sizeof(float);
}

extern "C" int mlqem_segment_pool_f32(const float* x, int64_t ldx, const float* weights, const int32_t* graph_ptr, int64_t N,
                                      int64_t B, int C, float* out_mean, int64_t ld_mean, float* out_wmean, int64_t ld_wmean,
                                      void* workspace, size_t workspace_bytes, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || B < 0 || C <= 0 || ldx < C || B > 0x7fffffffLL || N > 0x7fffffffLL) return MLQEM_ERR_BAD_ARG;
  if ((!out_mean && !out_wmean) || (out_mean && ld_mean < C) || (out_wmean && ld_wmean < C)) return MLQEM_ERR_BAD_ARG;
  if (B == 0) return MLQEM_OK;
  if (!graph_ptr || (N > 0 && !x)) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_segment_pool_workspace_bytes(N, B, C)) return MLQEM_ERR_WORKSPACE;
  hipStream_t s = as_stream(stream);
  const int c4 = (C + 3) / 4 * 4;
  // 16-byte row accesses when the rows own round_up(C, 4) columns (the padded activation layout): the pad columns are
  // summed along and never read back
  const bool wide = ldx >= c4 && ldx % 4 == 0 && aligned_to(x, 16);
  PoolArgs a{x, ldx, weights, graph_ptr, N, (int)B, C, wide ? c4 / 4 : C, static_cast<float*>(workspace), pool_rows(N, B)};
  if (a.CV > kBlock) return MLQEM_ERR_UNSUPPORTED;
  if (N > 0) {
    const unsigned tiles = (unsigned)ceil_div(N, (int64_t)a.rows);
    if (wide) hipLaunchKernelGGL(pool_partial_kernel<4>, dim3(tiles), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL(pool_partial_kernel<1>, dim3(tiles), dim3(kBlock), 0, s, a);
  }
  hipLaunchKernelGGL(pool_finish_kernel, dim3((unsigned)ceil_div(B * C, kBlock)), dim3(kBlock), 0, s, a.partial, graph_ptr, (int)B,
                     C, wide ? c4 : C, a.rows, out_mean, ld_mean, out_wmean, ld_wmean);
  return launch_status();
}

namespace mlqem {
int launch_aggregate_with_pool(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* idx, const int32_t* ell, const float* cscale,
                               const float* rscale, const float* dself, float alpha, float beta, const float* z, int64_t ldz,
                               const float* bias, int act, float drop_p, uint64_t seed, const uint64_t* seed_counter, float* out,
                               int64_t ldo, int64_t N, int C, const float* pool_weights, const int32_t* graph_ptr, int B,
                               float* partial, int2* tile_graph, uint8_t* gate_bits, int* rows_per_tile, hipStream_t stream);
int aggregate_pool_rows_per_tile(int C);
int aggregate_pool_mask_words();
}  // namespace mlqem

// Bytes of the gate buffer the pooled aggregation leaves for mlqem_segment_pool_bwd_f32: per tile of rows a 16-byte record
// (graph, its boundaries) and the sign bits of the tile's items as per-wave ballots (csr_aggregate.hip PoolFuse::mask).
extern "C" size_t mlqem_csr_aggregate_pool_gate_bytes(int64_t N, int C) {
  if (N < 0 || C <= 0) return 0;
  const int64_t tiles = ceil_div(std::max<int64_t>(N, 1), aggregate_pool_rows_per_tile(C));
  // ... and, for rows of at most four 16-byte slices, the same bits per node (16 each; csr_aggregate.hip PoolFuse::node_gate)
  const size_t per_node = (C + 3) / 4 <= 4 ? ((size_t)(N + 1) / 2 * 4 + 15) / 16 * 16 : 0;
  return (size_t)tiles * (sizeof(int4) + (size_t)aggregate_pool_mask_words() * sizeof(unsigned long long)) + per_node;
}

extern "C" size_t mlqem_csr_aggregate_pool_workspace_bytes(int64_t N, int64_t B, int C) {
  if (N < 0 || B < 0 || C <= 0) return 0;
  const int64_t tiles = ceil_div(std::max<int64_t>(N, 1), aggregate_pool_rows_per_tile(C));
  return (size_t)(tiles + B) * 2 * ((C + 3) / 4 * 4) * sizeof(float) + (size_t)tiles * sizeof(int2);     // partial sums | tile -> (graph, next boundary)
}

// mlqem_csr_aggregate_f32 and mlqem_segment_pool_f32 of its output in one pass over the rows: the aggregation's workgroups
// keep their tile of the output in LDS and leave per-(tile, graph) partial sums, the finish kernel is the pool's own.
extern "C" int mlqem_csr_aggregate_pool_f32(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* idx, const int32_t* ell,
                                            const float* cscale, const float* rscale, const float* dself, float alpha, float beta,
                                            const float* z, int64_t ldz, const float* bias, int act, float drop_p, uint64_t seed,
                                            const uint64_t* seed_counter, float* out, int64_t ldo, int64_t N, int C,
                                            const float* pool_weights, const int32_t* graph_ptr, int64_t B, float* out_mean,
                                            int64_t ld_mean, float* out_wmean, int64_t ld_wmean, uint8_t* gate_bits,
                                            void* workspace, size_t workspace_bytes, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || B < 0 || C <= 0 || drop_p < 0.f || drop_p >= 1.f || B > 0x7fffffffLL || N > 0x7fffffffLL) return MLQEM_ERR_BAD_ARG;
  if ((!out_mean && !out_wmean) || (out_mean && ld_mean < C) || (out_wmean && ld_wmean < C)) return MLQEM_ERR_BAD_ARG;
  if (!ell) return MLQEM_ERR_UNSUPPORTED;
  if (!out && !gate_bits) return MLQEM_ERR_BAD_ARG;      // the activation may stay unwritten only when its gate bits are kept
  if (B == 0) return MLQEM_OK;
  if (!graph_ptr) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_csr_aggregate_pool_workspace_bytes(N, B, C)) return MLQEM_ERR_WORKSPACE;
  hipStream_t s = as_stream(stream);
  int rows = 0;
  if (N > 0) {
    const int64_t tiles = ceil_div(N, aggregate_pool_rows_per_tile(C));
    float* partial = static_cast<float*>(workspace);
    int2* tile_graph = reinterpret_cast<int2*>(partial + (tiles + B) * 2 * ((C + 3) / 4 * 4));
    const int rc = launch_aggregate_with_pool(x, ldx, ptr, idx, ell, cscale, rscale, dself, alpha, beta, z, ldz, bias, act, drop_p, seed,
                                              seed_counter, out, ldo, N, C, pool_weights, graph_ptr, (int)B, partial, tile_graph, gate_bits, &rows, s);
    if (rc != MLQEM_OK) return rc;
    if (rows != aggregate_pool_rows_per_tile(C)) return MLQEM_ERR_LAUNCH;      // the workspace was sized for another tiling
  } else {
    rows = aggregate_pool_rows_per_tile(C);
  }
  const int c4 = (C + 3) / 4 * 4;
  if (2 * c4 <= 32)
    hipLaunchKernelGGL(pool_finish_tiles_kernel, dim3((unsigned)B), dim3(kBlock), 0, s, static_cast<const float*>(workspace), graph_ptr, C, c4,
                       rows, out_mean, ld_mean, out_wmean, ld_wmean);
  else
    hipLaunchKernelGGL(pool_finish_kernel, dim3((unsigned)ceil_div(B * C, kBlock)), dim3(kBlock), 0, s, static_cast<const float*>(workspace),
                       graph_ptr, (int)B, C, c4, rows, out_mean, ld_mean, out_wmean, ld_wmean);
  return launch_status();
}

extern "C" int mlqem_segment_pool_bwd_f32(const float* g_mean, int64_t ld_gmean, const float* g_wmean, int64_t ld_gwmean,
                                          const float* weights, const int32_t* graph_ptr, int64_t N, int64_t B, int C,
                                          const float* gate, int64_t ldgate, float gate_scale, const uint8_t* gate_bits, float* gx,
                                          int64_t ldgx, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || B < 0 || C <= 0 || ldgx < C || B > 0x7fffffffLL || N > 0x7fffffffLL) return MLQEM_ERR_BAD_ARG;
  if ((!g_mean && !g_wmean) || (g_mean && ld_gmean < C) || (g_wmean && ld_gwmean < C) || (gate && ldgate < C)) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!graph_ptr || !gx || B == 0) return MLQEM_ERR_BAD_ARG;
  const int c4 = (C + 3) / 4 * 4;
  auto rows_ok = [&](const float* p, int64_t ld) { return !p || (ld >= c4 && ld % 4 == 0 && aligned_to(p, 16)); };
  // the [B, C] gradient rows are read with the same vector width as the [N, C] rows: they must own their padding too
  const bool wide = rows_ok(gx, ldgx) && rows_ok(gate, ldgate) && rows_ok(g_mean, ld_gmean) && rows_ok(g_wmean, ld_gwmean);
  const int cv = wide ? c4 / 4 : C;
  const int64_t blocks = ceil_div(N * cv, (int64_t)kBlock * kPoolBwdItems);
  if (blocks > 0x7fffffffLL) return MLQEM_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
#define MLQEM_POOL_BWD(V, G) hipLaunchKernelGGL((pool_bwd_kernel<V, G>), dim3((unsigned)blocks), dim3(kBlock), 0, s, g_mean, ld_gmean, \
                                                g_wmean, ld_gwmean, weights, graph_ptr, N, (int)B, cv, gate, ldgate, gate_scale, gx, ldgx)
  if (gate_bits) {      // the pooled aggregation's tile records + ballots: the layout of the 16-byte form only
    if (gate || !wide || cv * 4 > kWave || !aligned_to(gate_bits, 16)) return MLQEM_ERR_BAD_ARG;
    const int R = aggregate_pool_rows_per_tile(C);
    const int64_t tiles = ceil_div(N, (int64_t)R);
    const int4* info = reinterpret_cast<const int4*>(gate_bits);
    const unsigned long long* words = reinterpret_cast<const unsigned long long*>(info + tiles);
#define MLQEM_POOL_BWD_TILES(T) hipLaunchKernelGGL((pool_bwd_tiles_kernel<T>), dim3((unsigned)tiles), dim3(kBlock), 0, s, g_mean, ld_gmean, \
                                                   g_wmean, ld_gwmean, weights, graph_ptr, N, (int)B, cv, R, gate_scale, info, words,        \
                                                   aggregate_pool_mask_words(), gx, ldgx)
    if (cv == 3) MLQEM_POOL_BWD_TILES(3); else MLQEM_POOL_BWD_TILES(0);
#undef MLQEM_POOL_BWD_TILES
  } else if (wide) { if (gate) MLQEM_POOL_BWD(4, 1); else MLQEM_POOL_BWD(4, 0); }
  else { if (gate) MLQEM_POOL_BWD(1, 1); else MLQEM_POOL_BWD(1, 0); }
#undef MLQEM_POOL_BWD
  return launch_status();
}

static int head_args(const mlqem_head_desc* d, int64_t B, int C, HeadArgs& a) {
  if (!d || B < 0 || C <= 0 || d->n_terms < 1 || d->n_terms > MLQEM_HEAD_MAX_TERMS || d->n_cols < 1 || d->n_cols > MLQEM_HEAD_MAX_TERMS)
    return MLQEM_ERR_BAD_ARG;
  a.n_terms = d->n_terms; a.n_cols = d->n_cols; a.C = C; a.B = B;
  for (int t = 0; t < d->n_terms; ++t) {
    if (!d->P[t] || !d->W[t] || d->ldp[t] < C || d->col[t] < 0 || d->col[t] >= d->n_cols) return MLQEM_ERR_BAD_ARG;
    a.P[t] = static_cast<const float*>(d->P[t]); a.ldp[t] = d->ldp[t]; a.W[t] = static_cast<const float*>(d->W[t]); a.col[t] = d->col[t];
  }
  for (int k = 0; k < MLQEM_HEAD_MAX_TERMS; ++k) a.bias[k] = k < d->n_cols ? static_cast<const float*>(d->bias[k]) : nullptr;
  return MLQEM_OK;
}

extern "C" int mlqem_pooled_head_f32(const mlqem_head_desc* desc, int64_t B, int C, float* out, int64_t ldo, mlqem_stream_t stream) {
  begin_launches();
  HeadArgs a{};
  const int rc = head_args(desc, B, C, a);
  if (rc != MLQEM_OK) return rc;
  if (B == 0) return MLQEM_OK;
  if (!out || ldo < a.n_cols) return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(pooled_head_kernel, dim3((unsigned)ceil_div(B * a.n_cols, kBlock)), dim3(kBlock), 0, as_stream(stream), a, out, ldo);
  return launch_status();
}

extern "C" int mlqem_pooled_head_bwd_f32(const mlqem_head_desc* desc, const float* gout, int64_t ldg, int64_t B, int C,
                                         float* const* gP, const int64_t* ldgp, float* gW, float* gb, mlqem_stream_t stream) {
  begin_launches();
  HeadArgs a{};
  const int rc = head_args(desc, B, C, a);
  if (rc != MLQEM_OK) return rc;
  if (!gout || ldg < a.n_cols || !gP || !ldgp || !gW || !gb) return MLQEM_ERR_BAD_ARG;
  HeadGradRows o{};
  for (int t = 0; t < a.n_terms; ++t) {
    if (!gP[t] || ldgp[t] < C) return MLQEM_ERR_BAD_ARG;
    o.gP[t] = gP[t]; o.ld[t] = ldgp[t];
  }
  hipStream_t s = as_stream(stream);
  if (B > 0)
    hipLaunchKernelGGL(pooled_head_bwd_rows_kernel, dim3((unsigned)ceil_div(B * C * a.n_terms, kBlock)), dim3(kBlock), 0, s, a, gout, ldg,
                       o);
  hipLaunchKernelGGL(pooled_head_bwd_sums_kernel, dim3((unsigned)(a.n_terms + a.n_cols)), dim3(kBlock), 0, s, a, gout, ldg, gW, gb);
  return launch_status();
}
