// How many 256-thread workgroups with a given dynamic-LDS request are resident per CU on this device?  (API answer and a measured one:
// every workgroup spins for a fixed number of clock ticks; the launch's duration tells how many ran side by side.)
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/lds_occupancy.hip -o /tmp/lds_occupancy && /tmp/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void spin(long long ticks, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* f = reinterpret_cast<float*>(smem);
  f[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  float acc = 0.f;
  while (wall_clock64() - t0 < ticks) acc += f[(threadIdx.x * 7) & 255];
  if (acc == -1.f) out[0] = acc;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("%s: CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerBlockOptin %zu\n", prop.name,
         prop.multiProcessorCount, prop.sharedMemPerBlock, prop.maxSharedMemoryPerMultiProcessor, prop.sharedMemPerBlockOptin);
  float* out;
  hipMalloc(&out, 4);
  const int cus = prop.multiProcessorCount;
  for (int kb : {8, 32, 40, 48, 52, 56, 64, 72, 78, 80, 96, 120, 160}) {
    const size_t lds = (size_t)kb * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      printf("%3d KB: attribute refused\n", kb);
      (void)hipGetLastError();
      continue;
    }
    int api = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, spin, 256, lds);
    const int wgs = cus * 8;                       // eight workgroups per CU's worth of work
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), lds, 0, 1000LL, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), lds, 0, 2000LL, out);      // 2000 ticks of the 100 MHz wall clock = 20 us per workgroup
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    printf("%3d KB: API says %d per CU; 8 per CU of 20 us each took %.1f us => %.1f side by side\n", kb, api, ms * 1e3, 8.0 * 20.0 / (ms * 1e3));
  }
  return 0;
}
