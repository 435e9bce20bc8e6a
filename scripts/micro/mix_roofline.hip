// What can the memory system deliver for the ACCESS MIX of the first-layer kernels, with no arithmetic at all?
// Reads an [N, 24] fp32 matrix and writes K blocks of [N, 12] (K = 1..6; 6 = the fan-out GEMM's mix, 1 : 3 read : write),
// and the reverse (reads K blocks, writes nothing: the weight gradient's mix).  Build: hipcc -O3 --offload-arch=gfx950
// scripts/micro/mix_roofline.hip -o /tmp/mix_roofline ; run: /tmp/mix_roofline [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../include/mlqem_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Ptrs { float* y[8]; };
typedef float f4 __attribute__((ext_vector_type(4)));

template <int K, bool NT, bool TILE>
__global__ __launch_bounds__(256) void fan_kernel(const float* __restrict__ x, Ptrs p, long n) {
  // TILE: a wave owns 16 consecutive rows at a time, like the MFMA kernel (lane -> row l%16, 16-byte piece l/16);
  // else: flat float4 items.
  const long wave = (blockIdx.x * 256L + threadIdx.x) >> 6, n_waves = (gridDim.x * 256L) >> 6;
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const long tiles = (n + 15) / 16;
  for (long t = wave; t < tiles; t += n_waves) {
    const long row = t * 16 + lr;
    if (row >= n) continue;
    float4 a = *reinterpret_cast<const float4*>(x + row * 24 + 4 * lq);
    float4 b = make_float4(0, 0, 0, 0);
    if (lq < 2) b = *reinterpret_cast<const float4*>(x + row * 24 + 16 + 4 * lq);
    float4 v = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    if (lq < 3) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        f4* dst = reinterpret_cast<f4*>(p.y[k] + row * 12 + 4 * lq);
        const f4 vv = {v.x, v.y, v.z, v.w};
        if (NT) __builtin_nontemporal_store(vv, dst); else *dst = vv;
        v.x += 1.f;
      }
    }
  }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same traffic plus MF dependent-chain MFMAs (16x16x4 fp32) per block and tile; PF: next tile's loads issued before them
template <int K, int MF, bool PF>
__global__ __launch_bounds__(256) void fan_mfma_kernel(const float* __restrict__ x, Ptrs p, long n) {
  const long wave = (blockIdx.x * 256L + threadIdx.x) >> 6, n_waves = (gridDim.x * 256L) >> 6;
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const long tiles = (n + 15) / 16;
  auto load = [&](long t, float4& a, float4& b) {
    const long row = t * 16 + lr;
    a = make_float4(0, 0, 0, 0); b = a;
    if (t < tiles && row < n) {
      a = *reinterpret_cast<const float4*>(x + row * 24 + 4 * lq);
      if (lq < 2) b = *reinterpret_cast<const float4*>(x + row * 24 + 16 + 4 * lq);
    }
  };
  float4 a, b, an, bn;
  if (PF) load(wave, an, bn);
  for (long t = wave; t < tiles; t += n_waves) {
    const long row = t * 16 + lr;
    if (PF) { a = an; b = bn; load(t + n_waves, an, bn); } else load(t, a, b);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float av[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int m = 0; m < MF; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f + k, av[m & 7], acc, 0, 0, 0);
      if (MF == 0) acc = f32x4{a.x + b.x + k, a.y, a.z, a.w};
      if (lq < 3 && row < n) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(p.y[k] + row * 12 + 4 * lq));
    }
  }
}

struct RealArgs { const float* x; float* y[8]; const float* w[8]; const float* b[8]; const float* rs[8]; long ldy[8]; long n; int yn, xc, yc, yw; long ldx; };
// steps from the synthetic kernel towards the product kernel: STAGE 1 = weights from LDS; 2 = + bias and row scale;
// 3 = + run-time block count / pointers / leading dimensions / live-column masks
template <int STAGE>
__global__ __launch_bounds__(256) void fan_real_kernel(const RealArgs a) {
  constexpr int G = 2;
  __shared__ float4 s_w[8][G][64];
  __shared__ float s_b[8][16];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < a.yn * G * 64; idx += 256) {
    const int blk = idx / (G * 64), g = (idx / 64) % G, l = idx % 64;
    const int o = l & 15, q = l >> 4;
    float v[4];
    for (int s4 = 0; s4 < 4; ++s4) { const int k = 16 * g + 4 * q + s4; v[s4] = (o < a.yc && k < a.xc) ? a.w[blk][o * a.xc + k] : 0.f; }
    s_w[blk][g][l] = make_float4(v[0], v[1], v[2], v[3]);
  }
  for (int idx = tid; idx < a.yn * 16; idx += 256) s_b[idx >> 4][idx & 15] = (a.b[idx >> 4] && (idx & 15) < a.yc) ? a.b[idx >> 4][idx & 15] : 0.f;
  __syncthreads();
  const long wave = (blockIdx.x * 256L + tid) >> 6, n_waves = (gridDim.x * 256L) >> 6;
  const int lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const long n = a.n, tiles = (n + 15) / 16;
  for (long t = wave; t < tiles; t += n_waves) {
    const long row = t * 16 + lr;
    const bool row_ok = row < n;
    float4 av[2];
    if (STAGE >= 3) {
      for (int g = 0; g < G; ++g) {
        const int k0 = 16 * g + 4 * lq; const int live = k0 < a.xc ? min(4, a.xc - k0) : 0;
        av[g] = make_float4(0, 0, 0, 0);
        if (row_ok && live) av[g] = *reinterpret_cast<const float4*>(a.x + row * a.ldx + k0);
        if (live < 2) av[g].y = 0.f; if (live < 3) av[g].z = 0.f; if (live < 4) av[g].w = 0.f; if (live < 1) av[g].x = 0.f;
      }
    } else {
      av[0] = av[1] = make_float4(0, 0, 0, 0);
      if (row_ok) { av[0] = *reinterpret_cast<const float4*>(a.x + row * 24 + 4 * lq); if (lq < 2) av[1] = *reinterpret_cast<const float4*>(a.x + row * 24 + 16 + 4 * lq); }
    }
#pragma unroll
    for (int blk = 0; blk < (STAGE >= 3 ? 8 : 6); ++blk) {
      if (STAGE >= 3 && blk >= a.yn) break;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float4 w4 = s_w[blk][g][lane];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, av[g].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, av[g].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, av[g].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, av[g].w, acc, 0, 0, 0);
      }
      if (!row_ok || lq >= 3) continue;
      if (STAGE >= 2) {
        const float rs = a.rs[blk] ? a.rs[blk][row] : 1.f;
        for (int r = 0; r < 4; ++r) { acc[r] += s_b[blk][4 * lq + r]; if (a.rs[blk]) acc[r] *= rs; }
      }
      const long ld = STAGE >= 3 ? a.ldy[blk] : 12;
      __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(a.y[blk] + row * ld + 4 * lq));
    }
  }
}

template <int K>
__global__ __launch_bounds__(256) void fanin_kernel(const float* __restrict__ x, Ptrs p, long n, float* out) {
  const long wave = (blockIdx.x * 256L + threadIdx.x) >> 6, n_waves = (gridDim.x * 256L) >> 6;
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const long tiles = (n + 15) / 16;
  float acc = 0.f;
  for (long t = wave; t < tiles; t += n_waves) {
    const long row = t * 16 + lr;
    if (row >= n) continue;
    float4 a = *reinterpret_cast<const float4*>(x + row * 24 + 4 * lq);
    acc += a.x + a.y + a.z + a.w;
    if (lq < 2) { float4 b = *reinterpret_cast<const float4*>(x + row * 24 + 16 + 4 * lq); acc += b.x + b.w; }
    if (lq < 3) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float4 g = *reinterpret_cast<const float4*>(p.y[k] + row * 12 + 4 * lq);
        acc += g.x + g.y + g.z + g.w;
      }
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

template <typename F>
static double time_us(F launch, float* flush, long flush_n) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ts;
  for (int r = 0; r < 7; ++r) {
    CK(hipMemsetAsync(flush, r, flush_n * 4, 0));
    CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

// weight-gradient mix in the PRODUCT's lane layout: per k-step (4 rows) six scalar loads of gy (column 16*ob + lr of row 4u + lq,
// blocks of 12 columns in separate buffers) and two of x; MODE 0: add them up; 1: 48 MFMAs per 16 rows; 2: + next slab's loads first
template <int MODE>
__global__ __launch_bounds__(256) void wgrad_like_kernel(const float* __restrict__ x, Ptrs p, long n, float* out) {
  const long wave = (blockIdx.x * 256L + threadIdx.x) >> 6, n_waves = (gridDim.x * 256L) >> 6;
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const float* gcol[6];
  for (int ob = 0; ob < 6; ++ob) { const int o = 16 * ob + lr, part = o / 12, lc = o - 12 * part; gcol[ob] = part < 7 ? p.y[part] + lc : nullptr; }
  f32x4 acc[6][2];
  for (int ob = 0; ob < 6; ++ob) for (int ib = 0; ib < 2; ++ib) acc[ob][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  float sum = 0.f;
  const long iters = (n + 15) / 16;
  auto issue = [&](long it, float (&A)[4][6], float (&B)[4][2]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long r = it * 16 + 4 * u + lq; const bool ok = it < iters && r < n;
#pragma unroll
      for (int ob = 0; ob < 6; ++ob) A[u][ob] = (ok && gcol[ob]) ? gcol[ob][r * 12] : 0.f;
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) { const int i = 16 * ib + lr; B[u][ib] = (ok && i < 24) ? x[r * 24 + i] : 0.f; }
    }
  };
  auto use = [&](const float (&A)[4][6], const float (&B)[4][2]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int ob = 0; ob < 6; ++ob)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
          if (MODE == 0) sum += A[u][ob] * B[u][ib];
          else acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[u][ob], B[u][ib], acc[ob][ib], 0, 0, 0);
        }
  };
  if (MODE == 2) {
    float a0[4][6], b0[4][2];
    issue(wave, a0, b0);
    for (long it = wave; it < iters; it += n_waves) {
      float a1[4][6], b1[4][2];
      issue(it + n_waves, a1, b1);
      use(a0, b0);
      for (int u = 0; u < 4; ++u) { for (int ob = 0; ob < 6; ++ob) a0[u][ob] = a1[u][ob]; for (int ib = 0; ib < 2; ++ib) b0[u][ib] = b1[u][ib]; }
    }
  } else {
    for (long it = wave; it < iters; it += n_waves) { float a0[4][6], b0[4][2]; issue(it, a0, b0); use(a0, b0); }
  }
  for (int ob = 0; ob < 6; ++ob) for (int ib = 0; ib < 2; ++ib) sum += acc[ob][ib][0] + acc[ob][ib][1] + acc[ob][ib][2] + acc[ob][ib][3];
  if (sum == 123.456f) out[0] = sum;
}

// the same with KU k-steps (4 rows each) per iteration and DEPTH iterations of loads in flight ahead of the one being multiplied
template <int KU, int DEPTH>
__global__ __launch_bounds__(256) void wgrad_pipe_kernel(const float* __restrict__ x, Ptrs p, long n, float* out) {
  const long wave = (blockIdx.x * 256L + threadIdx.x) >> 6, n_waves = (gridDim.x * 256L) >> 6;
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const float* gcol[6];
  for (int ob = 0; ob < 6; ++ob) { const int o = 16 * ob + lr, part = o / 12, lc = o - 12 * part; gcol[ob] = part < 7 ? p.y[part] + lc : nullptr; }
  f32x4 acc[6][2];
  for (int ob = 0; ob < 6; ++ob) for (int ib = 0; ib < 2; ++ib) acc[ob][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long iters = (n + 4 * KU - 1) / (4 * KU);
  float A[DEPTH + 1][KU][6], B[DEPTH + 1][KU][2];
  auto issue = [&](long it, float (&a)[KU][6], float (&b)[KU][2]) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const long r = it * 4 * KU + 4 * u + lq; const bool ok = it < iters && r < n;
#pragma unroll
      for (int ob = 0; ob < 6; ++ob) a[u][ob] = (ok && gcol[ob]) ? gcol[ob][r * 12] : 0.f;
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) { const int i = 16 * ib + lr; b[u][ib] = (ok && i < 24) ? x[r * 24 + i] : 0.f; }
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(wave + d * n_waves, A[d], B[d]);
  for (long it = wave; it < iters; it += n_waves) {
    issue(it + DEPTH * n_waves, A[DEPTH], B[DEPTH]);
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
      for (int ob = 0; ob < 6; ++ob)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0][u][ob], B[0][u][ib], acc[ob][ib], 0, 0, 0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int u = 0; u < KU; ++u) {
#pragma unroll
        for (int ob = 0; ob < 6; ++ob) A[d][u][ob] = A[d + 1][u][ob];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) B[d][u][ib] = B[d + 1][u][ib];
      }
  }
  float sum = 0.f;
  for (int ob = 0; ob < 6; ++ob) for (int ib = 0; ib < 2; ++ib) sum += acc[ob][ib][0] + acc[ob][ib][1] + acc[ob][ib][2] + acc[ob][ib][3];
  if (sum == 123.456f) out[0] = sum;
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 11291888L;
  float* x; CK(hipMalloc(&x, n * 24 * 4)); CK(hipMemset(x, 0, n * 24 * 4));
  Ptrs p;
  for (int k = 0; k < 8; ++k) { CK(hipMalloc(&p.y[k], n * 12 * 4 + (k * 449376L))); CK(hipMemset(p.y[k], 0, n * 12 * 4)); }
  float* flush; const long fn = 1L << 28; CK(hipMalloc(&flush, fn * 4));
  float* out; CK(hipMalloc(&out, 4));
  for (int wgs : {5, 16}) {
    const int grid = 256 * wgs;
#define FAN(K, NT) { double us = time_us([&] { hipLaunchKernelGGL((fan_kernel<K, NT, true>), dim3(grid), dim3(256), 0, 0, x, p, n); }, flush, fn); \
      double gb = n * (96.0 + K * 48.0) / 1e9; printf("fan-out  K=%d nt=%d wg/CU=%2d: %7.1f us  %5.2f TB/s (%.2f GB)\n", K, NT, wgs, us, gb / us * 1e3, gb); }
    FAN(6, true)
#define FIN(K) { double us = time_us([&] { hipLaunchKernelGGL((fanin_kernel<K>), dim3(grid), dim3(256), 0, 0, x, p, n, out); }, flush, fn); \
      double gb = n * (96.0 + K * 48.0) / 1e9; printf("fan-in   K=%d      wg/CU=%2d: %7.1f us  %5.2f TB/s (%.2f GB)\n", K, wgs, us, gb / us * 1e3, gb); }
    FIN(7)
#define FANM(MF, PF) { double us = time_us([&] { hipLaunchKernelGGL((fan_mfma_kernel<6, MF, PF>), dim3(grid), dim3(256), 0, 0, x, p, n); }, flush, fn); \
      printf("fan-out  K=6 mfma/block=%d prefetch=%d wg/CU=%2d: %7.1f us\n", MF, PF, wgs, us); }
    FANM(0, false) FANM(8, false)
  }
  {
    float *w, *bias, *rs;
    CK(hipMalloc(&w, 6 * 10 * 22 * 4)); CK(hipMemset(w, 0, 6 * 10 * 22 * 4));
    CK(hipMalloc(&bias, 64)); CK(hipMemset(bias, 0, 64));
    CK(hipMalloc(&rs, n * 4)); CK(hipMemset(rs, 0, n * 4));
    RealArgs a{}; a.x = x; a.n = n; a.yn = 6; a.xc = 22; a.yc = 10; a.yw = 12; a.ldx = 24;
    for (int k = 0; k < 6; ++k) { a.y[k] = p.y[k]; a.w[k] = w + 220 * k; a.ldy[k] = 12; }
    a.b[1] = bias; a.b[5] = bias; a.rs[0] = rs;
    for (int wgs : {5, 16}) {
      const int grid = 256 * wgs;
#define REAL(S) { double us = time_us([&] { hipLaunchKernelGGL((fan_real_kernel<S>), dim3(grid), dim3(256), 0, 0, a); }, flush, fn); printf("fan-out towards product, stage %d wg/CU=%2d: %7.1f us\n", S, wgs, us); }
      REAL(1) REAL(2) REAL(3)
    }
  }
  for (int wgs : {8}) {
    const int grid = 256 * wgs;
#define WGL(M) { double us = time_us([&] { hipLaunchKernelGGL((wgrad_like_kernel<M>), dim3(grid), dim3(256), 0, 0, x, p, n, out); }, flush, fn); printf("wgrad-like mode %d wg/CU=%2d: %7.1f us\n", M, wgs, us); }
    WGL(0) WGL(1) WGL(2)
  }
  for (int wgs : {2, 3, 4, 5, 6, 8, 12, 16, 32}) {
    const int grid = 256 * wgs;
#define WGP(KU, D) { double us = time_us([&] { hipLaunchKernelGGL((wgrad_pipe_kernel<KU, D>), dim3(grid), dim3(256), 0, 0, x, p, n, out); }, flush, fn); printf("wgrad-pipe KU=%d depth=%d wg/CU=%2d: %7.1f us\n", KU, D, wgs, us); }
    WGP(4, 1) WGP(4, 2) WGP(2, 1) WGP(2, 2) WGP(2, 3) WGP(1, 2) WGP(1, 4)
  }
  // the product kernels on the same buffers, same box, same harness (libmlqem_hip.so through its C ABI)
  {
    float *w, *bias, *rs, *gw, *gb; void* wsp;
    CK(hipMalloc(&w, 6 * 10 * 22 * 4)); CK(hipMemset(w, 0, 6 * 10 * 22 * 4));
    CK(hipMalloc(&bias, 64)); CK(hipMemset(bias, 0, 64));
    CK(hipMalloc(&rs, n * 4)); CK(hipMemset(rs, 0, n * 4));
    CK(hipMalloc(&gw, 84 * 22 * 4)); CK(hipMalloc(&gb, 84 * 4));
    const size_t wsb = mlqem_linear_wgrad_workspace_bytes(22, 84); CK(hipMalloc(&wsp, wsb));
    int32_t* rows; CK(hipMalloc(&rows, n * 4));
    { std::vector<int32_t> h(n); for (long i = 0; i < n; ++i) h[i] = (int32_t)i; CK(hipMemcpy(rows, h.data(), n * 4, hipMemcpyHostToDevice)); }
    mlqem_col_parts xp{1, 24, 22, 0, {x}, {24}}, yp{6, 12, 10, 0, {}, {}}, gp{7, 12, 10, 0, {}, {}};
    for (int k = 0; k < 7; ++k) { if (k < 6) { yp.ptr[k] = p.y[k]; yp.ld[k] = 12; } gp.ptr[k] = p.y[k]; gp.ld[k] = 12; }
    const float* wb[8] = {w, w + 220, w + 440, w + 660, w + 880, w + 1100};
    const float* bb[8] = {nullptr, bias, nullptr, nullptr, nullptr, bias};
    const float* rb[8] = {rs, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int map = 0; map < 2; ++map) {
      double us = time_us([&] { int rc = mlqem_linear_parts_f32(&xp, wb, nullptr, 0, bb, rb, &yp, n, nullptr, 0, 1.f, map ? rows : nullptr, nullptr); if (rc) { printf("rc %d\n", rc); exit(1); } }, flush, fn);
      printf("PRODUCT fan-out 22 -> 6x10 rowmap=%d: %7.1f us\n", map, us);
      us = time_us([&] { int rc = mlqem_linear_wgrad_parts_f32(&gp, x, 24, gw, gb, n, 22, 0, wsp, wsb, map ? rows : nullptr, nullptr); if (rc) { printf("rc %d\n", rc); exit(1); } }, flush, fn);
      printf("PRODUCT wgrad 7x10 x 22 rowmap=%d: %7.1f us\n", map, us);
    }
  }
  return 0;
}
