// What the fp32 matrix cores of THIS box sustain: v_mfma_f32_16x16x4_f32 back to back on random operands, one wave per SIMD
// and two, 8 independent accumulators (the shape of the MLP head's inner loop), for ~10 ms per launch; the in-kernel clock is
// read as delta(s_memtime) / delta(s_memrealtime) x 100 MHz (guide: MI355X_MICROARCH.md, DVFS give-back item 6).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_f32_rate.hip -o scripts/micro/mfma_f32_rate && scripts/micro/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, unsigned long long* clk, int iters) {
  const int lane = threadIdx.x;
  float a[8], b[4];
  for (int i = 0; i < 8; ++i) a[i] = in[(blockIdx.x * 256 + lane) * 12 + i];
  for (int i = 0; i < 4; ++i) b[i] = in[(blockIdx.x * 256 + lane) * 12 + 8 + i];
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[s], acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + lane] = s;
  if (lane == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  const int blocks_max = 512, iters = 20000;
  std::vector<float> h(blocks_max * 256 * 12);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *in, *out; unsigned long long* clk;
  hipMalloc(&in, h.size() * 4); hipMalloc(&out, blocks_max * 256 * 4); hipMalloc(&clk, blocks_max * 16);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int blocks : {256, 512}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> c(2 * blocks);
      hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
      const double ghz = (double)c[0] / (double)c[1] * 0.1;
      const double flops = (double)blocks * 4 * iters * 32 * 2048.0;
      printf("%d workgroups (%d wave/SIMD): %.2f ms, %.1f TFLOP/s, in-kernel clock %.2f GHz, %.1f cycles per MFMA per SIMD\n", blocks,
             blocks / 256, ms, flops / ms / 1e9, ghz, (double)c[0] / (iters * 32.0) / (blocks / 256));
    }
  }
  return 0;
}
