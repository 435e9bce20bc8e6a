// The two ends of a train step that are not model layers (docs/tutorials/__ml_models.py:100-187: `loss = criterion(out, y);
// loss.backward(); optimizer.step()` with torch.nn.MSELoss and torch.optim.Adam), as ONE launch each.
//
//   mlqem_mse_loss_grad_f32   loss = mean (out - y)^2 and g = 2 (out - y) / (N C) from one pass over the outputs.  torch spends
//                             a subtract-square kernel, a mean reduction, a ones fill for the backward root and a gradient
//                             kernel on it: four launches of ~4-8 us each on rows the MLP head kernels process in 50-180 us.
//   mlqem_adam_step_f32       Adam (no amsgrad, no weight decay) on one flat parameter buffer, step count and learning rate
//                             resident on the device (capturable in a hipGraph).  torch's fused multi-tensor kernel hands a
//                             22 k-float buffer to ONE workgroup: 35 us per step, as long as the bf16 MLP1 forward GEMM.
//
// Both finish inside their own launch with the "last workgroup done" pattern: every workgroup publishes its partial (or, for
// Adam, has read the step count) before it takes a ticket, and the workgroup that draws the last ticket does the serial tail --
// the partials summed in INDEX order (deterministic whichever workgroup comes last), the step count written back -- and
// returns the ticket counter to zero for the next launch.
#include "common.hpp"

namespace mlqem {

constexpr int kLossMaxBlocks = 256;

__global__ __launch_bounds__(kBlock) void mse_loss_grad_kernel(const float* __restrict__ out, int64_t ldo, const float* __restrict__ y,
                                                               int64_t ldy, float* __restrict__ g, int64_t ldg, int64_t N, int C,
                                                               int64_t g_rows, float grad_scale, float* __restrict__ loss,
                                                               float* __restrict__ partial, unsigned* __restrict__ ticket) {
  __shared__ float s_red[kBlock / kWave];
  __shared__ bool s_last;
  const int64_t total = N * C;
  const int64_t per = ceil_div(total, (int64_t)gridDim.x);
  const int64_t e0 = (int64_t)blockIdx.x * per, e1 = e0 + per < total ? e0 + per : total;
  float acc = 0.f;
  if (g && g_rows > N) {       // rows of the output that are not part of the loss (a padded batch's filler rows) get a zero gradient
    const int64_t extra = (g_rows - N) * C;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < extra; e += (int64_t)gridDim.x * kBlock) {
      const int64_t r = N + (C == 1 ? e : e / C);
      const int c = C == 1 ? 0 : (int)(e % C);
      g[r * ldg + c] = 0.f;
    }
  }
  for (int64_t e = e0 + threadIdx.x; e < e1; e += kBlock) {
    const int64_t r = C == 1 ? e : e / C;
    const int c = C == 1 ? 0 : (int)(e - r * C);
    const float d = out[r * ldo + c] - y[r * ldy + c];
    acc = fmaf(d, d, acc);
    if (g) g[r * ldg + c] = d * grad_scale;
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & (kWave - 1)) == 0) s_red[threadIdx.x / kWave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) t += s_red[w];
    if (gridDim.x == 1) {                              // a batch of a few rows: nothing to hand over, no fence (a release
      *loss = t / (float)total;                        // fence writes the whole L2's dirty lines back: microseconds)
      s_last = false;
    } else {
      partial[blockIdx.x] = t;
      __threadfence();                                 // the partial is visible before the ticket is
      s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // the last workgroup: partials in index order, one wave, a fixed tree
  if (threadIdx.x < kWave) {
    float t = 0.f;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += kWave) t += __builtin_nontemporal_load(partial + b);
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) t += __shfl_xor(t, off);
    if (threadIdx.x == 0) {
      *loss = t / (float)total;
      *ticket = 0u;
    }
  }
}

// One thread per four parameters.  s = step + 1 is what every thread computes with; the last workgroup writes it back.
__global__ __launch_bounds__(kBlock) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, int64_t n, const float* __restrict__ lr_ptr,
                                                           float* __restrict__ step_ptr, double beta1_d, double beta2_d, float eps,
                                                           unsigned* __restrict__ ticket, unsigned long long* __restrict__ bump) {
  const float s = *step_ptr + 1.0f;
  const float lr = *lr_ptr;
  // torch.optim.Adam: bias corrections in double from the double betas (fused_adam_utils.cuh), the update in fp32
  const double bc1 = 1.0 - pow(beta1_d, (double)s);
  const double bc2 = 1.0 - pow(beta2_d, (double)s);
  // the weights as torch forms them: 1 - beta in DOUBLE, then rounded (1.0f - 0.999f is off by 1.3e-5 of itself)
  const float beta2 = (float)beta2_d, w1 = (float)(1.0 - beta1_d), w2 = (float)(1.0 - beta2_d);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
  auto update = [&](float& pi, float gi, float& mi, float& vi) {
    mi = mi + (gi - mi) * w1;                                    // exp_avg.lerp_(grad, 1 - beta1)
    vi = beta2 * vi + w2 * gi * gi;                              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
  };
  if (i0 + 3 < n) {
    float4 pp = *reinterpret_cast<float4*>(p + i0), mm = *reinterpret_cast<float4*>(m + i0), vv = *reinterpret_cast<float4*>(v + i0);
    const float4 gg = *reinterpret_cast<const float4*>(g + i0);
    update(pp.x, gg.x, mm.x, vv.x); update(pp.y, gg.y, mm.y, vv.y); update(pp.z, gg.z, mm.z, vv.z); update(pp.w, gg.w, mm.w, vv.w);
    *reinterpret_cast<float4*>(p + i0) = pp; *reinterpret_cast<float4*>(m + i0) = mm; *reinterpret_cast<float4*>(v + i0) = vv;
  } else {
    for (int64_t i = i0; i < n; ++i) update(p[i], g[i], m[i], v[i]);
  }
  __syncthreads();                                     // every thread of this workgroup has read the step count
  if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1) {
    *step_ptr = s;
    *ticket = 0u;
    if (bump) *bump += 1ull;                           // the caller's step counter (dropout keys): one torch launch a step less
  }
}

}  // namespace mlqem

using namespace mlqem;

extern "C" size_t mlqem_mse_loss_workspace_bytes(void) { return (size_t)kLossMaxBlocks * sizeof(float); }

extern "C" int mlqem_mse_loss_grad_f32(const float* out, int64_t ldo, const float* y, int64_t ldy, float* g, int64_t ldg, int64_t N,
                                       int C, int64_t g_rows, float* loss, void* workspace, size_t workspace_bytes, unsigned* ticket,
                                       mlqem_stream_t stream) {
  begin_launches();
  if (N < 1 || C < 1 || !out || !y || !loss || !ticket || ldo < C || ldy < C || (g && (ldg < C || g_rows < N))) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_mse_loss_workspace_bytes()) return MLQEM_ERR_WORKSPACE;
  const int64_t total = N * C;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(kLossMaxBlocks, ceil_div(total, 4 * kBlock)));
  hipLaunchKernelGGL(mse_loss_grad_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), out, ldo, y, ldy, g, ldg, N, C, g_rows,
                     2.0f / (float)total, loss, static_cast<float*>(workspace), ticket);
  return launch_status();
}

extern "C" int mlqem_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr,
                                   float* step, double beta1, double beta2, double eps, unsigned* ticket, uint64_t* bump_counter,
                                   mlqem_stream_t stream) {
  begin_launches();
  if (n < 0 || !lr || !step || !ticket || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return MLQEM_ERR_BAD_ARG;
  if (!aligned_to(param, 16) || !aligned_to(grad, 16) || !aligned_to(exp_avg, 16) || !aligned_to(exp_avg_sq, 16)) return MLQEM_ERR_BAD_ARG;
  const unsigned grid = (unsigned)std::max<int64_t>(1, ceil_div(n, 4 * kBlock));
  hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n, lr, step,
                     beta1, beta2, (float)eps, ticket, reinterpret_cast<unsigned long long*>(bump_counter));
  return launch_status();
}
