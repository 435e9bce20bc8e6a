// Mean pooling over the contiguous node range of each graph (global_mean_pool) and its backward.
#include "common.hpp"

namespace mlqem {

// One block per graph: threads stride over the graph's rows, then the per-thread sums are added in a fixed order.
// Graphs on this path have 7 ... 20,711 nodes and C <= 125 channels; a batch has a few hundred graphs, so the block is
// made as large as the hardware allows (1024 threads = 16 waves) -- with 256 threads the 256-graph benchmark batch kept
// one wave per SIMD busy and took 32 us per call for 45 MB.
constexpr int kPoolBlock = 1024;
__global__ __launch_bounds__(kPoolBlock) void segment_mean_kernel(const float* __restrict__ x, int64_t ldx,
                                                                  const int32_t* __restrict__ gptr,
                                                                  float* __restrict__ out, int64_t ldo, int C) {
  constexpr int kBlock = kPoolBlock;   // shadows the library-wide block size inside this kernel
  __shared__ float red[kBlock];
  const int g = blockIdx.x;
  const int beg = gptr[g], end = gptr[g + 1];
  // lane layout: cl consecutive threads cover the channels of one row, kBlock/cl rows in flight
  const int cl = C >= kBlock ? kBlock : C;
  const int rows_par = kBlock / cl;
  const int c_lane = threadIdx.x % cl, r_lane = threadIdx.x / cl;
  for (int c0 = 0; c0 < C; c0 += cl) {
    const int c = c0 + c_lane;
    float s = 0.f;
    if (r_lane < rows_par && c < C)
      for (int r = beg + r_lane; r < end; r += rows_par) s += x[(int64_t)r * ldx + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (r_lane == 0 && c < C) {
      float tot = 0.f;
      for (int k = 0; k < rows_par; ++k) tot += red[k * cl + c_lane];
      const int n = end - beg;
      out[(int64_t)g * ldo + c] = n > 0 ? tot / (float)n : 0.f;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void segment_mean_bwd_kernel(const float* __restrict__ g, int64_t ldg,
                                                                  const int32_t* __restrict__ gptr,
                                                                  float* __restrict__ gx, int64_t ldgx, int C) {
  const int gi = blockIdx.x;
  const int beg = gptr[gi], end = gptr[gi + 1];
  const int n = end - beg;
  if (n <= 0) return;
  const float inv = 1.f / (float)n;
  const int64_t total = (int64_t)n * C;
  for (int64_t t = (int64_t)blockIdx.y * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.y * kBlock) {
    const int r = (int)(t / C), c = (int)(t % C);
    gx[(int64_t)(beg + r) * ldgx + c] = g[(int64_t)gi * ldg + c] * inv;
  }
}

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_segment_mean_f32(const float* x, int64_t ldx, const int32_t* graph_ptr, float* out, int64_t ldo,
                                      int64_t B, int C, mlqem_stream_t stream) {
  begin_launches();
  if (B < 0 || C <= 0 || ldx < C || ldo < C) return MLQEM_ERR_BAD_ARG;
  if (B == 0) return MLQEM_OK;
  if (!x || !graph_ptr || !out) return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(segment_mean_kernel, dim3((unsigned)B), dim3(kPoolBlock), 0, as_stream(stream), x, ldx, graph_ptr,
                     out, ldo, C);
  return launch_status();
}

extern "C" int mlqem_segment_mean_bwd_f32(const float* g, int64_t ldg, const int32_t* graph_ptr, float* gx,
                                          int64_t ldgx, int64_t B, int C, mlqem_stream_t stream) {
  begin_launches();
  if (B < 0 || C <= 0 || ldg < C || ldgx < C) return MLQEM_ERR_BAD_ARG;
  if (B == 0) return MLQEM_OK;
  if (!g || !graph_ptr || !gx) return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(segment_mean_bwd_kernel, dim3((unsigned)B, 8), dim3(kBlock), 0, as_stream(stream), g, ldg,
                     graph_ptr, gx, ldgx, C);
  return launch_status();
}
