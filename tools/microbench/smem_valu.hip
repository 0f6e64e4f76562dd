// Microbenchmark (gfx950): wave-uniform records fetched with scalar loads (s_load_dwordx4 from random 16-byte
// records of a large array, as the blend kernels would fetch survivor records) + fp32 VALU work with SGPR operands,
// 16 waves per CU.  Build: hipcc --offload-arch=gfx950 -O3 -o smem_valu smem_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int R, int F, int AHEAD>
__global__ void __launch_bounds__(64) k(const float4* __restrict__ rec, float* out, int iters, unsigned mask) {
  float acc[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  const float m = 1.0000001f + threadIdx.x * 1e-9f;
  unsigned h = __builtin_amdgcn_readfirstlane(blockIdx.x * 2654435761u + 12345u);
  float4 cur[R], nxt[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    h = h * 1664525u + 1013904223u;
    cur[r] = rec[(h >> 4) & mask];
  }
  for (int it = 0; it < iters; ++it) {
    if (AHEAD) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        h = h * 1664525u + 1013904223u;
        nxt[r] = rec[(h >> 4) & mask];
      }
    }
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s4[0] += cur[r].x; s4[1] += cur[r].y; s4[2] += cur[r].z; s4[3] += cur[r].w;
    }
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f & 7] = __builtin_fmaf(acc[f & 7], m, R ? cur[f % R].x : 1.0f);
    acc[0] += s4[0] + s4[1] + s4[2] + s4[3];
    if (AHEAD) {
#pragma unroll
      for (int r = 0; r < R; ++r) cur[r] = nxt[r];
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        h = h * 1664525u + 1013904223u;
        cur[r] = rec[(h >> 4) & mask];
      }
    }
  }
  float t = 0.f;
  for (int i = 0; i < 8; ++i) t += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = t;
}

template <int R, int F, int AHEAD>
static void run(const char* name, const float4* rec, float* out, int waves_per_cu, unsigned mask) {
  const int cus = 256, iters = 5000;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  k<R, F, AHEAD><<<cus * waves_per_cu, 64>>>(rec, out, 50, mask);
  (void)hipEventRecord(a);
  k<R, F, AHEAD><<<cus * waves_per_cu, 64>>>(rec, out, iters, mask);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  printf("%-34s waves/CU %2d R=%2d F=%3d records %8u: %.3f ms -> %.0f cycles per iteration per wave\n", name, waves_per_cu, R, F,
         mask + 1, ms, ms * 1e-3 * 2.4e9 / iters);
}

int main() {
  const unsigned n = 1u << 22;  // 4M records x 16 B = 64 MB
  float4* rec;
  float* out;
  (void)hipMalloc(&rec, (size_t)n * sizeof(float4));
  (void)hipMemset(rec, 0, (size_t)n * sizeof(float4));
  (void)hipMalloc(&out, 256 * 32 * 64 * sizeof(float));
  run<12, 136, 0>("smem(12)+valu, no prefetch, 64MB", rec, out, 16, n - 1);
  run<12, 136, 1>("smem(12)+valu, prefetch 1, 64MB", rec, out, 16, n - 1);
  run<6, 68, 1>("smem(6)+valu(68), prefetch 1, 64MB", rec, out, 16, n - 1);
  run<12, 136, 1>("smem(12)+valu, prefetch 1, 1MB", rec, out, 16, (1u << 16) - 1);
  run<12, 136, 1>("smem(12)+valu, prefetch 1, 16KB", rec, out, 16, (1u << 10) - 1);
  run<12, 0, 1>("smem(12) only, prefetch 1, 64MB", rec, out, 16, n - 1);
  run<12, 136, 1>("smem(12)+valu, prefetch 1, 64MB, 8w", rec, out, 8, n - 1);
  return 0;
}
