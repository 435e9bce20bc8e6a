// BatchNorm1d in training mode over [N, C] activations (the bn1 / bn2 of MLP2 / MLP3, docs/tutorials/mlp.py:45-66,
// 87-108), forward and backward.
//
//   forward:   mean_c = E[x_c], var_c = E[(x_c - s_c)^2] - (mean_c - s_c)^2 (biased; s_c = x[0, c], a shift that keeps the two
//              terms at the size of the variance: E[x^2] - mean^2 on raw values cancels catastrophically for columns with
//              |mean| >> std, where torch's Welford reduction does not), y = (x - mean) * rsqrt(var + eps) * gamma + beta
//   backward:  dbeta = sum dy, dgamma = sum dy * xhat, dx = gamma * invstd * (dy - dbeta / N - xhat * dgamma / N)
//
// Two passes per direction: a column reduction (every workgroup takes a contiguous range of rows, lanes along the
// columns so that a wave reads whole rows, eight row lanes per workgroup added in a fixed order; one partial per workgroup,
// summed in a fixed order in double precision by a finish kernel, one wave per column: deterministic, no atomics) and an
// element-wise pass.  torch's kernels for this shape (N = 262 144, C = 125) take 7 ms (Welford reduction) and 20 ms (backward)
// on an MI355X -- 94 % of an MLP3 train step; these take 0.1-0.2 ms each, the time of their 2-3 passes over the matrix.
#include "common.hpp"

namespace mlqem {

constexpr int kBnLanes = 32;                      // lanes along the columns
constexpr int kBnRowLanes = kBlock / kBnLanes;    // 8 row lanes
constexpr int kBnMaxChunks = 8;                   // C <= 256
constexpr int kBnMaxBlocks = 2048;

// MODE 0: s1 = sum (x - s), s2 = sum (x - s)^2 with the per-column shift s = x[0, :].
// MODE 1: s1 = sum dy, s2 = sum dy * xhat, xhat = (x - mean) * invstd.
template <int MODE> __global__ __launch_bounds__(kBlock) void bn_column_sums_kernel(
    const float* __restrict__ a, int64_t lda, const float* __restrict__ x, int64_t ldx, const float* __restrict__ mean,
    const float* __restrict__ invstd, int64_t N, int C, int64_t rows_per_block, float* __restrict__ partial) {
  __shared__ float s_red[2][kBnRowLanes][kBnMaxChunks * kBnLanes];
  const int tx = threadIdx.x % kBnLanes, ty = threadIdx.x / kBnLanes;
  const int chunks = (C + kBnLanes - 1) / kBnLanes;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
  float s1[kBnMaxChunks], s2[kBnMaxChunks], mu[kBnMaxChunks], is[kBnMaxChunks];
#pragma unroll
  for (int k = 0; k < kBnMaxChunks; ++k) {
    s1[k] = s2[k] = 0.f;
    const int c = k * kBnLanes + tx;
    mu[k] = (k < chunks && c < C) ? (MODE == 1 ? mean[c] : a[c]) : 0.f;     // MODE 0: the shift (row 0 of the matrix)
    is[k] = (MODE == 1 && k < chunks && c < C) ? invstd[c] : 0.f;
  }
  for (int64_t r = r0 + ty; r < r1; r += kBnRowLanes) {
#pragma unroll
    for (int k = 0; k < kBnMaxChunks; ++k) {
      const int c = k * kBnLanes + tx;
      if (k < chunks && c < C) {
        const float v = a[r * lda + c];
        if (MODE == 0) {
          const float d = v - mu[k];
          s1[k] += d;
          s2[k] = fmaf(d, d, s2[k]);
        } else {
          s1[k] += v;
          s2[k] = fmaf(v, (x[r * ldx + c] - mu[k]) * is[k], s2[k]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kBnMaxChunks; ++k) {
    s_red[0][ty][k * kBnLanes + tx] = s1[k];
    s_red[1][ty][k * kBnLanes + tx] = s2[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += kBlock) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int j = 0; j < kBnRowLanes; ++j) { t1 += s_red[0][j][c]; t2 += s_red[1][j][c]; }
    partial[((int64_t)blockIdx.x * 2 + 0) * C + c] = t1;
    partial[((int64_t)blockIdx.x * 2 + 1) * C + c] = t2;
  }
}

// One 64-lane workgroup per column: lane j adds the partials of workgroups j, j + 64, ... in double, lane 0 adds the 64 lane
// sums in lane order (a fixed order: deterministic).  (One thread per column walking all 2048 partials took 0.5 ms.)
// MODE 0: mean, biased variance, invstd, and the affine map of the forward (scale = gamma * invstd, shift = beta - mean * scale).
// MODE 1: dbeta = s1, dgamma = s2, and the per-column constants of the backward's element-wise pass.
template <int MODE> __global__ __launch_bounds__(kWave) void bn_finish_kernel(
    const float* __restrict__ partial, int nblocks, int64_t N, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ invstd_in /* MODE 0: the shift, row 0 of x */, float eps, float* __restrict__ o1,
    float* __restrict__ o2, float* __restrict__ o3, float* __restrict__ o4, float* __restrict__ o5) {
  __shared__ double s_t[2][kWave];
  const int c = blockIdx.x, j = threadIdx.x;
  double t1 = 0.0, t2 = 0.0;
  for (int b = j; b < nblocks; b += kWave) {
    t1 += (double)partial[((int64_t)b * 2 + 0) * C + c];
    t2 += (double)partial[((int64_t)b * 2 + 1) * C + c];
  }
  s_t[0][j] = t1;
  s_t[1][j] = t2;
  __syncthreads();
  if (j != 0) return;
  t1 = t2 = 0.0;
  for (int k = 0; k < kWave; ++k) { t1 += s_t[0][k]; t2 += s_t[1][k]; }
  if (MODE == 0) {
    const double ms = t1 / (double)N;                   // mean of the shifted values
    double var = t2 / (double)N - ms * ms;
    if (var < 0.0) var = 0.0;
    const double m = (double)invstd_in[c] + ms;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    o1[c] = (float)m;                         // mean
    o2[c] = (float)var;                       // biased variance
    o3[c] = is;                               // invstd
    o4[c] = sc;                               // scale
    o5[c] = (beta ? beta[c] : 0.f) - (float)m * sc;   // shift
  } else {
    o1[c] = (float)t1;                        // dbeta
    o2[c] = (float)t2;                        // dgamma
    o3[c] = (gamma ? gamma[c] : 1.f) * invstd_in[c];  // gamma * invstd
    o4[c] = (float)(t1 / (double)N);          // dbeta / N
    o5[c] = (float)(t2 / (double)N);          // dgamma / N
  }
}

// y = x * scale[c] + shift[c]
__global__ __launch_bounds__(kBlock) void bn_affine_kernel(const float* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int64_t N, int C,
                                                           float* __restrict__ y, int64_t ldy) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= N * C) return;
  const int64_t r = t / C;
  const int c = (int)(t - r * C);
  y[r * ldy + c] = fmaf(x[r * ldx + c], scale[c], shift[c]);
}

// dx = gs[c] * (dy - k1[c] - xhat * k2[c]),  xhat = (x - mean[c]) * invstd[c]
__global__ __launch_bounds__(kBlock) void bn_bwd_apply_kernel(const float* __restrict__ dy, int64_t ldg,
                                                              const float* __restrict__ x, int64_t ldx,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gs, const float* __restrict__ k1,
                                                              const float* __restrict__ k2, int64_t N, int C,
                                                              float* __restrict__ dx, int64_t lddx) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= N * C) return;
  const int64_t r = t / C;
  const int c = (int)(t - r * C);
  const float xh = (x[r * ldx + c] - mean[c]) * invstd[c];
  dx[r * lddx + c] = gs[c] * (dy[r * ldg + c] - k1[c] - xh * k2[c]);
}

static int bn_blocks(int64_t N) {
  const int64_t want = ceil_div(N, (int64_t)(4 * kBnRowLanes));   // at least four passes of the row lanes per workgroup
  return (int)(want < 1 ? 1 : (want > kBnMaxBlocks ? kBnMaxBlocks : want));
}

}  // namespace mlqem

using namespace mlqem;

// workspace: partial sums [blocks][2][C] + five per-column vectors of the finish kernels
extern "C" size_t mlqem_batch_norm_workspace_bytes(int64_t N, int C) {
  if (N < 0 || C <= 0) return 0;
  return ((size_t)bn_blocks(N) * 2 * C + 5 * (size_t)C) * sizeof(float) + 256;
}

extern "C" int mlqem_batch_norm_train_f32(const float* x, int64_t ldx, int64_t N, int C, const float* gamma,
                                          const float* beta, float eps, float* y, int64_t ldy, float* mean, float* var,
                                          float* invstd, void* workspace, size_t workspace_bytes, mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N <= 0 || C <= 0 || ldx < C || ldy < C) return MLQEM_ERR_BAD_ARG;
  if (C > kBnMaxChunks * kBnLanes) return MLQEM_ERR_UNSUPPORTED;
  if (!x || !y || !mean || !var || !invstd) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_batch_norm_workspace_bytes(N, C)) return MLQEM_ERR_WORKSPACE;
  const int nb = bn_blocks(N);
  float* partial = static_cast<float*>(workspace);
  float* scale = partial + (size_t)nb * 2 * C;
  float* shift = scale + C;
  hipLaunchKernelGGL(bn_column_sums_kernel<0>, dim3((unsigned)nb), dim3(kBlock), 0, stream, x, ldx, (const float*)nullptr,
                     (int64_t)0, (const float*)nullptr, (const float*)nullptr, N, C, ceil_div(N, (int64_t)nb), partial);
  hipLaunchKernelGGL(bn_finish_kernel<0>, dim3((unsigned)C), dim3(kWave), 0, stream, partial, nb, N, C, gamma, beta,
                     x /* the shift: row 0 */, eps, mean, var, invstd, scale, shift);
  hipLaunchKernelGGL(bn_affine_kernel, dim3((unsigned)ceil_div(N * C, (int64_t)kBlock)), dim3(kBlock), 0, stream, x, ldx,
                     scale, shift, N, C, y, ldy);
  return launch_status();
}

extern "C" int mlqem_batch_norm_train_bwd_f32(const float* dy, int64_t ldg, const float* x, int64_t ldx, int64_t N, int C,
                                              const float* gamma, const float* mean, const float* invstd, float* dx,
                                              int64_t lddx, float* dgamma, float* dbeta, void* workspace,
                                              size_t workspace_bytes, mlqem_stream_t stream_) {
  begin_launches();
  hipStream_t stream = as_stream(stream_);
  if (N <= 0 || C <= 0 || ldg < C || ldx < C || lddx < C) return MLQEM_ERR_BAD_ARG;
  if (C > kBnMaxChunks * kBnLanes) return MLQEM_ERR_UNSUPPORTED;
  if (!dy || !x || !mean || !invstd || !dx || !dgamma || !dbeta) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_batch_norm_workspace_bytes(N, C)) return MLQEM_ERR_WORKSPACE;
  const int nb = bn_blocks(N);
  float* partial = static_cast<float*>(workspace);
  float* gs = partial + (size_t)nb * 2 * C;
  float* k1 = gs + C;
  float* k2 = k1 + C;
  hipLaunchKernelGGL(bn_column_sums_kernel<1>, dim3((unsigned)nb), dim3(kBlock), 0, stream, dy, ldg, x, ldx, mean, invstd, N,
                     C, ceil_div(N, (int64_t)nb), partial);
  hipLaunchKernelGGL(bn_finish_kernel<1>, dim3((unsigned)C), dim3(kWave), 0, stream, partial, nb, N, C, gamma, (const float*)nullptr,
                     invstd, 0.f, dbeta, dgamma, gs, k1, k2);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)ceil_div(N * C, (int64_t)kBlock)), dim3(kBlock), 0, stream, dy, ldg,
                     x, ldx, mean, invstd, gs, k1, k2, N, C, dx, lddx);
  return launch_status();
}
