// Microbenchmark (gfx950): cost of wave-uniform (broadcast) ds_read_b128 versus fp32 VALU work, at the occupancy of
// the blend kernels (16 waves per CU).  Build: hipcc --offload-arch=gfx950 -O3 -o lds_valu lds_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int R, int F, bool PACKED>
__global__ void __launch_bounds__(64) k(float* out, int iters, int stride) {
  __shared__ __attribute__((aligned(16))) float s[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) s[i] = (float)i * 1e-3f;
  __syncthreads();
  float acc[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  v4f pacc[4] = {{1.f, 2.f, 3.f, 4.f}, {5.f, 6.f, 7.f, 8.f}, {1.5f, 2.5f, 3.5f, 4.5f}, {5.5f, 6.5f, 7.5f, 8.5f}};
  const float m = 1.0000001f + threadIdx.x * 1e-9f;
  int j = 0;
  for (int it = 0; it < iters; ++it) {
    v4f sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R; ++r) sum += *reinterpret_cast<const v4f*>(&s[r * 64 + j]);  // same address in all lanes
    j = (j + stride) & 60;
    if (PACKED) {
#pragma unroll
      for (int f = 0; f < F / 2; ++f) pacc[f & 3] = __builtin_elementwise_fma(pacc[f & 3], (v4f)m, sum);  // 2 v_pk_fma each
    } else {
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f & 7] = __builtin_fmaf(acc[f & 7], m, sum[f & 3]);
    }
    if (R > 0 && F == 0) acc[0] += sum[0] + sum[1] + sum[2] + sum[3];
  }
  float t = 0.f;
  for (int i = 0; i < 8; ++i) t += acc[i];
  for (int i = 0; i < 4; ++i) t += pacc[i][0] + pacc[i][1] + pacc[i][2] + pacc[i][3];
  out[blockIdx.x * 64 + threadIdx.x] = t;
}

template <int R, int F, bool PACKED>
static void run(const char* name, float* out, int waves_per_cu) {
  const int cus = 256, iters = 20000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<R, F, PACKED><<<cus * waves_per_cu, 64>>>(out, 100, 4);
  hipEventRecord(a);
  k<R, F, PACKED><<<cus * waves_per_cu, 64>>>(out, iters, 4);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  // cycles per iteration per wave at 2.4 GHz, and per-CU cycles per (iteration of all its waves)
  const double cyc = ms * 1e-3 * 2.4e9 / iters;
  printf("%-28s waves/CU %2d  R=%2d F=%3d  %.3f ms  -> %.0f cycles per iteration per wave (%.1f per LDS read, %.2f per fma-instr per SIMD)\n",
         name, waves_per_cu, R, F, ms, cyc, R ? cyc / R : 0.0, F ? cyc / (F * (waves_per_cu / 4.0)) : 0.0);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 32 * 64 * sizeof(float));
  for (int w : {4, 8, 16}) {
    if (w == 4) { run<11, 0, false>("lds only", out, 4); run<0, 136, false>("valu only", out, 4); run<0, 136, true>("valu packed", out, 4); run<11, 136, false>("lds+valu", out, 4); }
    if (w == 8) { run<11, 0, false>("lds only", out, 8); run<0, 136, false>("valu only", out, 8); run<0, 136, true>("valu packed", out, 8); run<11, 136, false>("lds+valu", out, 8); }
    if (w == 16) { run<11, 0, false>("lds only", out, 16); run<0, 136, false>("valu only", out, 16); run<0, 136, true>("valu packed", out, 16); run<11, 136, false>("lds+valu", out, 16); run<11, 136, true>("lds+valu packed", out, 16); run<4, 136, false>("lds(4)+valu", out, 16); }
  }
  return 0;
}
