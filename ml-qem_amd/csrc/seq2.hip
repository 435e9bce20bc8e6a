// nn.Sequential(Linear(I, H), [Dropout(p)], Linear(H, O)) -- the dense heads of the reference's graph models: `obs_seq`
// (n_qubits * 4 + 1 -> hidden -> 1 with Dropout(0.2)) and `body_seq` (6 -> hidden -> 1) of 01_ngem.ipynb cell [9], `body_seq`
// (pooled + 1 + k -> hidden -> k with Dropout) of docs/tutorials/gnn.py:94-98 -- as ONE launch per direction.
//
// These layers see one row per CIRCUIT (1024 rows per step, 32 at the reference's batch size), so they are nothing but launches:
// as separate GEMMs, bias / dropout kernels, weight-gradient passes and their second stages the two heads of Family A were 17
// launches of ~5 us each way out of 50 per step (a third of the replayed 32-circuit step).
//
// Nothing here is bound by arithmetic or bandwidth; what a launch costs is the number of DEPENDENT memory round trips on its
// longest thread.  So every loop below is short and its loads independent (unrolled: in flight together):
//   forward   a wave per row: lane = (hidden unit j, one of four column slices); the four slices are added by two shuffles, then
//             bias, dropout (mask bits kept per row: a hidden value that is exactly zero is not a dropped one), and the second
//             layer as a DPP sum over the sixteen hidden units per output.
//   backward  workgroup = 32 input columns x 32 row slices (1024 threads): a thread owns a column and every 32nd row, rebuilds
//             gh = (gy W2) * mask / (1 - p) for the row from uniform loads, accumulates its column of gW1 and writes its element
//             of gx when the input needs a gradient; the slices are added up in slice order (a shuffle, then LDS).  One more
//             workgroup adds up gb1, gW2, gb2 the same way.  No partial sums in memory, no second stage: deterministic, one launch.
#include "common.hpp"

namespace mlqem {

constexpr int kSeqMaxH = 16, kSeqMaxO = 8;

struct Seq2Fwd {
  const float* x; int64_t ldx; int64_t N; int I;
  const float* w1; const float* b1; int H;
  const float* w2; const float* b2; int O;
  float drop_p; uint64_t seed; const uint64_t* seed_counter;
  float* hidden; uint32_t* mask; float* y; int64_t ldy;
};

__global__ __launch_bounds__(kBlock) void seq2_forward_kernel(const Seq2Fwd a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int j = lane & (kGroup - 1), sl = lane >> 4;        // hidden unit, column slice (columns sl, sl + 4, ...)
  const int64_t row = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (row >= a.N) return;                                   // wave-uniform
  const bool live = j < a.H;
  const float* __restrict__ xr = a.x + row * a.ldx;
  const float* __restrict__ wr = a.w1 + (int64_t)(live ? j : 0) * a.I;
  float acc0 = 0.f, acc1 = 0.f;
  int c = sl;
#pragma unroll 4
  for (; c + 4 < a.I; c += 8) {                              // two chains, eight loads in flight per unrolled step
    acc0 = fmaf(xr[c], wr[c], acc0);
    acc1 = fmaf(xr[c + 4], wr[c + 4], acc1);
  }
  if (c < a.I) acc0 = fmaf(xr[c], wr[c], acc0);
  float v = acc0 + acc1;
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);                                   // every lane of unit j holds the sum over all columns
  const uint64_t seed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  bool keep = live;
  float h = 0.f;
  if (live) {
    v += a.b1 ? a.b1[j] : 0.f;
    if (a.drop_p > 0.f) {
      keep = uniform01(seed, (uint64_t)(row * a.H + j)) >= a.drop_p;
      v *= 1.f / (1.f - a.drop_p);
    }
    h = keep ? v : 0.f;
    if (sl == 0 && a.hidden) a.hidden[row * a.H + j] = h;
  }
  if (a.mask) {                                             // bit j of the row's mask: disjoint bits added up over the sixteen units (DPP)
    int m = keep ? (1 << j) : 0;
    m += __builtin_amdgcn_update_dpp(0, m, 0xB1, 0xF, 0xF, true);
    m += __builtin_amdgcn_update_dpp(0, m, 0x4E, 0xF, 0xF, true);
    m += __builtin_amdgcn_update_dpp(0, m, 0x141, 0xF, 0xF, true);
    m += __builtin_amdgcn_update_dpp(0, m, 0x140, 0xF, 0xF, true);
    if (lane == 0) a.mask[row] = (uint32_t)m;
  }
  for (int o = 0; o < a.O; ++o) {
    const float y = group16_sum(live ? h * a.w2[o * a.H + j] : 0.f);
    if (lane == 0) a.y[row * a.ldy + o] = y + (a.b2 ? a.b2[o] : 0.f);
  }
}

struct Seq2Bwd {
  const float* gy; int64_t ldgy; const float* x; int64_t ldx; int64_t N; int I;
  const float* w1; int H; const float* w2; int O;
  const float* hidden; const uint32_t* mask; float drop_p;
  float* gx; int64_t ldgx;
  float* gw1; float* gb1; float* gw2; float* gb2;
  int col_blocks;
};

constexpr int kSeqBlock = 1024, kSeqCols = 32, kSeqSlices = kSeqBlock / kSeqCols;     // 32 columns x 32 row slices per workgroup

template <bool ONE_OUT, bool WANT_GX>
__global__ __launch_bounds__(kSeqBlock) void seq2_backward_kernel(const Seq2Bwd a) {
  __shared__ float s_acc[kSeqSlices / 2][kSeqMaxH][kSeqCols];    // 32 KB: a wave holds two slices and adds them by a shuffle first
  __shared__ float s_w2[kSeqMaxO][kSeqMaxH];                     // W2, zero-padded: the row loop reads it at constant bounds, and no
  const int tid = threadIdx.x;                                   // global load of the loop sits behind the loop's gx stores
  const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  if (tid < kSeqMaxO * kSeqMaxH) {
    const int o = tid / kSeqMaxH, j = tid % kSeqMaxH;
    s_w2[o][j] = (o < a.O && j < a.H) ? a.w2[o * a.H + j] : 0.f;
  }
  __syncthreads();
  const float* __restrict__ gyp = a.gy;
  const float* __restrict__ xp = a.x;
  const uint32_t* __restrict__ maskp = a.mask;
  float* __restrict__ gxp = WANT_GX ? a.gx : nullptr;
  if ((int)blockIdx.x < a.col_blocks) {
    const int cl = tid % kSeqCols, sl = tid / kSeqCols;
    const int c = blockIdx.x * kSeqCols + cl;
    const bool col_ok = c < a.I;
    float w1c[kSeqMaxH], acc[kSeqMaxH];
#pragma unroll
    for (int j = 0; j < kSeqMaxH; ++j) {
      acc[j] = 0.f;
      w1c[j] = (WANT_GX && col_ok && j < a.H) ? a.w1[(int64_t)j * a.I + c] : 0.f;
    }
    if constexpr (ONE_OUT) {
      // one output (the heads of Family A): four rows' loads in flight, then the arithmetic -- gh[j] = gy w2[j] mask needs no array
      constexpr int U = WANT_GX ? 2 : 8;
      for (int64_t r0 = sl; r0 < a.N; r0 += (int64_t)U * kSeqSlices) {
        float xv[U], gv[U], gxv[U];
        uint32_t mv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t r = r0 + (int64_t)u * kSeqSlices;
          const bool ok = r < a.N;
          xv[u] = (ok && col_ok) ? xp[r * a.ldx + c] : 0.f;
          gv[u] = ok ? gyp[r * a.ldgy] * keep_scale : 0.f;
          mv[u] = (ok && maskp) ? maskp[r] : 0xFFFFFFFFu;
          gxv[u] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < kSeqMaxH; ++j) {
          const float w2j = s_w2[0][j];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const float g = ((mv[u] >> j) & 1u) ? gv[u] * w2j : 0.f;
            acc[j] = fmaf(g, xv[u], acc[j]);
            if (WANT_GX) gxv[u] = fmaf(g, w1c[j], gxv[u]);
          }
        }
        if (gxp && col_ok) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + (int64_t)u * kSeqSlices;
            if (r < a.N) gxp[r * a.ldgx + c] = gxv[u];
          }
        }
      }
    } else {
#pragma unroll 2
    for (int64_t r = sl; r < a.N; r += kSeqSlices) {
      const float xv = col_ok ? xp[r * a.ldx + c] : 0.f;
      const uint32_t m = maskp ? maskp[r] : 0xFFFFFFFFu;
      float gh[kSeqMaxH];
#pragma unroll
      for (int j = 0; j < kSeqMaxH; ++j) gh[j] = 0.f;
      for (int o = 0; o < a.O; ++o) {
        const float g = gyp[r * a.ldgy + o];
#pragma unroll
        for (int j = 0; j < kSeqMaxH; ++j) gh[j] = fmaf(g, s_w2[o][j], gh[j]);
      }
      float gxv = 0.f;
#pragma unroll
      for (int j = 0; j < kSeqMaxH; ++j) {
        const float g = ((m >> j) & 1u) ? gh[j] * keep_scale : 0.f;
        acc[j] = fmaf(g, xv, acc[j]);
        if (WANT_GX) gxv = fmaf(g, w1c[j], gxv);
      }
      if (gxp && col_ok) gxp[r * a.ldgx + c] = gxv;
    }
    }
#pragma unroll
    for (int j = 0; j < kSeqMaxH; ++j) {
      const float other = __shfl_down(acc[j], 32);            // the wave's odd slice onto its even one
      if ((tid & 32) == 0) s_acc[sl >> 1][j][cl] = acc[j] + other;
    }
    __syncthreads();
    for (int e = tid; e < kSeqMaxH * kSeqCols; e += kSeqBlock) {
      const int j = e / kSeqCols, cc = e % kSeqCols;
      const int col = blockIdx.x * kSeqCols + cc;
      if (j < a.H && col < a.I) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < kSeqSlices / 2; ++k) t += s_acc[k][j][cc];
        a.gw1[(int64_t)j * a.I + col] = t;
      }
    }
    return;
  }
  // the last workgroup: three families of sums over the rows, every one of the form sum_r gy[r][o] * f(r) with
  //   S0[o][j]: f = mask bit j of row r   (gb1[j] = sum_o W2[o][j] S0[o][j] / (1 - p))
  //   S1[o][j]: f = hidden[r][j]          (gW2[o][j])
  //   S2[o]:    f = 1                     (gb2[o])
  // thread = (sum, row slice); a loop per family, branch-free inside and unrolled: its loads are in flight together.
  float* s_flat = &s_acc[0][0][0];                               // [slices][sums]: at most 1024 floats
  const int oh = a.O * a.H, sums = 2 * oh + a.O;                 // <= 264
  const int slices = kSeqBlock / sums;                           // >= 3
  const int q = tid % sums, sl = tid / sums;
  if (sl < slices) {
    float t = 0.f;
    if (q < oh) {
      const int o = q / a.H, j = q % a.H;
      if (a.mask) {
#pragma unroll 8
        for (int64_t r = sl; r < a.N; r += slices) t += ((a.mask[r] >> j) & 1u) ? a.gy[r * a.ldgy + o] : 0.f;
      } else {
#pragma unroll 8
        for (int64_t r = sl; r < a.N; r += slices) t += a.gy[r * a.ldgy + o];
      }
    } else if (q < 2 * oh) {
      const int o = (q - oh) / a.H, j = (q - oh) % a.H;
#pragma unroll 8
      for (int64_t r = sl; r < a.N; r += slices) t = fmaf(a.gy[r * a.ldgy + o], a.hidden[r * a.H + j], t);
    } else {
      const int o = q - 2 * oh;
#pragma unroll 8
      for (int64_t r = sl; r < a.N; r += slices) t += a.gy[r * a.ldgy + o];
    }
    s_flat[sl * sums + q] = t;
  }
  __syncthreads();
  if (tid < sums) {
    float t = 0.f;
    for (int k = 0; k < slices; ++k) t += s_flat[k * sums + tid];
    s_flat[tid] = t;                                             // slice 0's slot: read below by the thread that wrote it, or after the barrier
  }
  __syncthreads();
  if (tid < a.H) {
    if (a.gb1) {
      float g = 0.f;
      for (int o = 0; o < a.O; ++o) g = fmaf(a.w2[o * a.H + tid], s_flat[o * a.H + tid], g);
      a.gb1[tid] = g * keep_scale;
    }
  } else if (tid < a.H + oh) {
    a.gw2[tid - a.H] = s_flat[oh + (tid - a.H)];
  } else if (tid < a.H + oh + a.O) {
    if (a.gb2) a.gb2[tid - a.H - oh] = s_flat[2 * oh + (tid - a.H - oh)];
  }
}

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_seq2_forward_f32(const float* x, int64_t ldx, int64_t N, int I, const float* w1, const float* b1, int H,
                                      const float* w2, const float* b2, int O, float drop_p, uint64_t seed, const uint64_t* seed_counter,
                                      float* hidden, uint32_t* mask, float* y, int64_t ldy, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || I <= 0 || H <= 0 || O <= 0 || ldx < I || ldy < O || drop_p < 0.f || drop_p >= 1.f) return MLQEM_ERR_BAD_ARG;
  if (H > kSeqMaxH || O > kSeqMaxO) return MLQEM_ERR_UNSUPPORTED;
  if (N == 0) return MLQEM_OK;
  if (!x || !w1 || !w2 || !y) return MLQEM_ERR_BAD_ARG;
  if (drop_p > 0.f && hidden && !mask) return MLQEM_ERR_BAD_ARG;       // a training forward keeps its mask
  const Seq2Fwd a{x, ldx, N, I, w1, b1, H, w2, b2, O, drop_p, seed, seed_counter, hidden, mask, y, ldy};
  hipLaunchKernelGGL(seq2_forward_kernel, dim3((unsigned)ceil_div(N, (int64_t)(kBlock / kWave))), dim3(kBlock), 0, as_stream(stream), a);
  return launch_status();
}

extern "C" size_t mlqem_seq2_backward_workspace_bytes(int64_t N, int I, int H, int O) {
  (void)N; (void)I; (void)H; (void)O;
  return 0;                                                   // sums meet in LDS: no workspace (kept in the ABI for a larger form)
}

extern "C" int mlqem_seq2_backward_f32(const float* gy, int64_t ldgy, const float* x, int64_t ldx, int64_t N, int I, const float* w1,
                                       int H, const float* w2, int O, const float* hidden, const uint32_t* mask, float drop_p,
                                       float* gx, int64_t ldgx, float* gw1, float* gb1, float* gw2, float* gb2, void* workspace,
                                       size_t workspace_bytes, unsigned* ticket, mlqem_stream_t stream) {
  begin_launches();
  (void)workspace; (void)workspace_bytes; (void)ticket;
  if (N <= 0 || I <= 0 || H <= 0 || O <= 0 || ldx < I || ldgy < O || (gx && ldgx < I) || drop_p < 0.f || drop_p >= 1.f) return MLQEM_ERR_BAD_ARG;
  if (H > kSeqMaxH || O > kSeqMaxO) return MLQEM_ERR_UNSUPPORTED;
  if (!gy || !x || !w1 || !w2 || !hidden || !gw1 || !gw2) return MLQEM_ERR_BAD_ARG;
  if (drop_p > 0.f && !mask) return MLQEM_ERR_BAD_ARG;
  const int col_blocks = (int)ceil_div((int64_t)I, (int64_t)kSeqCols);
  const Seq2Bwd a{gy, ldgy, x, ldx, N, I, w1, H, w2, O, hidden, drop_p > 0.f ? mask : nullptr, drop_p, gx, ldgx, gw1, gb1, gw2, gb2, col_blocks};
  const dim3 grid((unsigned)(col_blocks + 1));
  if (O == 1 && gx) hipLaunchKernelGGL((seq2_backward_kernel<true, true>), grid, dim3(kSeqBlock), 0, as_stream(stream), a);
  else if (O == 1) hipLaunchKernelGGL((seq2_backward_kernel<true, false>), grid, dim3(kSeqBlock), 0, as_stream(stream), a);
  else if (gx) hipLaunchKernelGGL((seq2_backward_kernel<false, true>), grid, dim3(kSeqBlock), 0, as_stream(stream), a);
  else hipLaunchKernelGGL((seq2_backward_kernel<false, false>), grid, dim3(kSeqBlock), 0, as_stream(stream), a);
  return launch_status();
}
