// Dense layers of the path (per-node projections inside the convs, the MLP heads).  All shapes here are
// "tall and skinny": N rows (nodes or graphs) by I, O <= a few hundred, so the kernels are bandwidth-bound
// on the N x (I + O) activations and the weights live in LDS.
#include "common.hpp"

namespace mlqem {

// y[n,o] = act(sum_k x[n,k] * W(o,k) + b[o]).  Thread t of the flattened (row, output) space; the x row is
// broadcast to the lanes that share it, W is read from LDS.  TRANSPOSED selects W stored [I,O] (gx = gy @ W).
template <bool TRANSPOSED>
__global__ __launch_bounds__(kBlock) void linear_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float* __restrict__ y, int64_t ldy, int64_t N, int I, int O,
                                                        int act, int accumulate) {
  extern __shared__ float w_lds[];  // [O][I+pad], pad makes the row stride odd -> conflict-free across o
  const int stride = I | 1;
  for (int p = threadIdx.x; p < O * I; p += kBlock) {
    const int o = TRANSPOSED ? p % O : p / I;
    const int k = TRANSPOSED ? p / O : p % I;
    w_lds[o * stride + k] = w[p];
  }
  __syncthreads();
  const int64_t total = N * O;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t row = t / O;
    const int o = (int)(t - row * O);
    const float* __restrict__ xr = x + row * ldx;
    const float* wr = w_lds + o * stride;
    float acc0 = b ? b[o] : 0.f, acc1 = 0.f;
    int k = 0;
    for (; k + 1 < I; k += 2) {
      acc0 = fmaf(xr[k], wr[k], acc0);
      acc1 = fmaf(xr[k + 1], wr[k + 1], acc1);
    }
    if (k < I) acc0 = fmaf(xr[k], wr[k], acc0);
    float r = acc0 + acc1;
    if (act & 1) r = fmaxf(r, 0.f);
    float* dst = y + row * ldy + o;
    *dst = accumulate ? *dst + r : r;
  }
}

// Stage 1 of the weight gradient: block b owns the row tiles b, b+G, ... and keeps one running sum per (o,i)
// pair (and per o for the bias) in registers; tiles of gy and x are staged through LDS with coalesced loads.
constexpr int kWgradTile = 64;      // rows per LDS tile
constexpr int kWgradMaxPairsPerThread = 16;

__global__ __launch_bounds__(kBlock) void wgrad_partial_kernel(const float* __restrict__ gy, int64_t ldgy,
                                                               const float* __restrict__ x, int64_t ldx,
                                                               float* __restrict__ partial, int64_t N, int I, int O) {
  extern __shared__ float tile[];  // gy tile [T][O] then x tile [T][I+1] (last column = 1 for the bias)
  float* gy_t = tile;
  float* x_t = tile + kWgradTile * O;
  const int I1 = I + 1;
  const int pairs = O * I1;
  const int pair0 = blockIdx.y * kBlock * kWgradMaxPairsPerThread;  // this block's slice of the (o,i) pairs
  float acc[kWgradMaxPairsPerThread];
#pragma unroll
  for (int q = 0; q < kWgradMaxPairsPerThread; ++q) acc[q] = 0.f;

  const int64_t n_tiles = ceil_div(N, kWgradTile);
  for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const int64_t r0 = tl * kWgradTile;
    const int rows = (int)min((int64_t)kWgradTile, N - r0);
    __syncthreads();
    for (int p = threadIdx.x; p < rows * O; p += kBlock) gy_t[p] = gy[(r0 + p / O) * ldgy + p % O];
    for (int p = threadIdx.x; p < rows * I1; p += kBlock) {
      const int r = p / I1, k = p % I1;
      x_t[p] = k < I ? x[(r0 + r) * ldx + k] : 1.f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kWgradMaxPairsPerThread; ++q) {
      const int p = pair0 + threadIdx.x + q * kBlock;
      if (p < pairs) {
        const int o = p / I1, k = p % I1;
        float s = acc[q];
        for (int r = 0; r < rows; ++r) s = fmaf(gy_t[r * O + o], x_t[r * I1 + k], s);
        acc[q] = s;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kWgradMaxPairsPerThread; ++q) {
    const int p = pair0 + threadIdx.x + q * kBlock;
    if (p < pairs) partial[(int64_t)blockIdx.x * pairs + p] = acc[q];
  }
}

// Stage 2: fixed-order sum over the G partials (deterministic), scattered to gw / gb.
__global__ __launch_bounds__(kBlock) void wgrad_reduce_kernel(const float* __restrict__ partial, int G, int I, int O,
                                                              float* __restrict__ gw, float* __restrict__ gb,
                                                              int accumulate) {
  const int I1 = I + 1;
  const int p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= O * I1) return;
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += partial[(int64_t)g * O * I1 + p];
  const int o = p / I1, k = p % I1;
  if (k < I) {
    float* d = gw + o * I + k;
    *d = accumulate ? *d + s : s;
  } else if (gb) {
    gb[o] = accumulate ? gb[o] + s : s;
  }
}

constexpr int kWgradBlocks = 512;

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_linear_f32(const float* x, int64_t ldx, const float* w, int transposed, const float* b, float* y,
                                int64_t ldy, int64_t N, int I, int O, int act, int accumulate, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || I <= 0 || O <= 0 || ldx < I || ldy < O) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!x || !w || !y) return MLQEM_ERR_BAD_ARG;
  const size_t lds = (size_t)O * (I | 1) * sizeof(float);
  if (lds > 64 * 1024) return MLQEM_ERR_UNSUPPORTED;
  const int64_t blocks = std::min<int64_t>(ceil_div(N * O, kBlock), 256 * 16);
  if (transposed)
    hipLaunchKernelGGL(linear_kernel<true>, dim3((unsigned)blocks), dim3(kBlock), lds, as_stream(stream), x, ldx, w, b,
                       y, ldy, N, I, O, act, accumulate);
  else
    hipLaunchKernelGGL(linear_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), lds, as_stream(stream), x, ldx, w, b,
                       y, ldy, N, I, O, act, accumulate);
  return launch_status();
}

extern "C" size_t mlqem_linear_wgrad_workspace_bytes(int I, int O) {
  return (size_t)kWgradBlocks * O * (I + 1) * sizeof(float);
}

extern "C" int mlqem_linear_wgrad_f32(const float* gy, int64_t ldgy, const float* x, int64_t ldx, float* gw, float* gb,
                                      int64_t N, int I, int O, int accumulate, void* workspace, size_t workspace_bytes,
                                      mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || I <= 0 || O <= 0 || !gw || ldgy < O || ldx < I) return MLQEM_ERR_BAD_ARG;
  const size_t lds = (size_t)kWgradTile * (O + I + 1) * sizeof(float);
  if (lds > 64 * 1024) return MLQEM_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < mlqem_linear_wgrad_workspace_bytes(I, O)) return MLQEM_ERR_WORKSPACE;
  if (N > 0 && (!gy || !x)) return MLQEM_ERR_BAD_ARG;
  const int G = (int)std::max<int64_t>(1, std::min<int64_t>(kWgradBlocks, ceil_div(N, kWgradTile)));
  float* partial = static_cast<float*>(workspace);
  const int slices = (int)ceil_div(O * (I + 1), kBlock * kWgradMaxPairsPerThread);
  hipLaunchKernelGGL(wgrad_partial_kernel, dim3(G, slices), dim3(kBlock), lds, as_stream(stream), gy, ldgy, x, ldx,
                     partial, N, I, O);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(O * (I + 1), kBlock)), dim3(kBlock), 0,
                     as_stream(stream), partial, G, I, O, gw, gb, accumulate);
  return launch_status();
}
