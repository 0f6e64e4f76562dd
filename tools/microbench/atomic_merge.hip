// Development tool: do memory-side float atomics of ONE wave instruction that fall into the same 32- / 64-byte piece of memory
// travel as one request?  (Round 6: K7's global atomics cost 47 of 215 us on the headline view and 200 of 757 us on synth-v2,
// GSR_BWD_ABLATE=2; its nine accumulators per Gaussian live in four arrays, so every atomic is a request of its own.)
//   hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics tools/microbench/atomic_merge.hip -o build_variants/atomic_merge
// Every variant adds nine values to each of M random rows (the same pseudo-random row sequence), rows of a table of N rows:
//   soa     nine instructions per 64 rows, lane = row, four arrays (3 + 4 + 1 + 3 floats per row): K7's flush today
//   aos16   rows of 16 floats; one instruction covers 4 rows x 16 lanes, lanes 0..8 of each 16 active
//   aos12   rows of 12 floats; one instruction covers 5 rows x 12 lanes (60 lanes), lanes 0..8 of each 12 active
//   aos16x  as aos16, but all 16 lanes active (16 values per row): is it the request or the dword that costs?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t row_of(uint32_t i, uint32_t N) {  // the i-th row of the sequence
  uint32_t x = i * 2654435761u + 12345u;
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  return x % N;
}

__global__ void __launch_bounds__(256) soa(float* m2, float* conic, float* op, float* col, uint32_t N, uint32_t M) {
  const uint32_t stride = gridDim.x * 256u;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < M; i += stride) {
    const size_t r = row_of(i, N);
    const float v = 1.0f;
    unsafeAtomicAdd(&m2[3 * r], v); unsafeAtomicAdd(&m2[3 * r + 1], v);
    unsafeAtomicAdd(&conic[4 * r], v); unsafeAtomicAdd(&conic[4 * r + 1], v); unsafeAtomicAdd(&conic[4 * r + 3], v);
    unsafeAtomicAdd(&op[r], v);
    unsafeAtomicAdd(&col[3 * r], v); unsafeAtomicAdd(&col[3 * r + 1], v); unsafeAtomicAdd(&col[3 * r + 2], v);
  }
}
template <int ROW, int ACTIVE>
__global__ void __launch_bounds__(256) aos(float* acc, uint32_t N, uint32_t M) {
  constexpr uint32_t PER = 64 / ROW;  // rows per wave instruction
  const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u;
  const uint32_t sub = lane / ROW, k = lane % ROW;
  for (uint32_t i0 = wave * PER; i0 < M; i0 += nwaves * PER) {
    const uint32_t i = i0 + sub;
    if (sub < PER && i < M && k < (uint32_t)ACTIVE) unsafeAtomicAdd(&acc[(size_t)row_of(i, N) * ROW + k], 1.0f);
  }
}

int main() {
  const uint32_t N = 1000000, M = 1u << 22;  // 4 M rows updated per launch (K7 headline: ~0.36 M (tile, instance) flushes x 9)
  float *m2, *conic, *op, *col, *acc;
  CHECK(hipMalloc(&m2, sizeof(float) * 3 * N)); CHECK(hipMalloc(&conic, sizeof(float) * 4 * N));
  CHECK(hipMalloc(&op, sizeof(float) * N)); CHECK(hipMalloc(&col, sizeof(float) * 3 * N));
  CHECK(hipMalloc(&acc, sizeof(float) * 16 * (size_t)N));
  CHECK(hipMemset(m2, 0, sizeof(float) * 3 * N)); CHECK(hipMemset(conic, 0, sizeof(float) * 4 * N));
  CHECK(hipMemset(op, 0, sizeof(float) * N)); CHECK(hipMemset(col, 0, sizeof(float) * 3 * N));
  CHECK(hipMemset(acc, 0, sizeof(float) * 16 * (size_t)N));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    launch();
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0, 0);
      launch();
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%-8s %8.1f us for %u rows x 9 values = %6.2f ns per row, %5.1f M values / ms\n", name, best * 1e3, M, best * 1e6 / M,
           9.0 * M / best / 1e6);
  };
  const dim3 g(2048), b(256);
  timeit("soa", [&] { hipLaunchKernelGGL(soa, g, b, 0, 0, m2, conic, op, col, N, M); });
  timeit("aos16", [&] { hipLaunchKernelGGL((aos<16, 9>), g, b, 0, 0, acc, N, M); });
  timeit("aos12", [&] { hipLaunchKernelGGL((aos<12, 9>), g, b, 0, 0, acc, N, M); });
  timeit("aos16x", [&] { hipLaunchKernelGGL((aos<16, 16>), g, b, 0, 0, acc, N, M); });
  timeit("aos8", [&] { hipLaunchKernelGGL((aos<8, 8>), g, b, 0, 0, acc, N, M); });
  // sanity: every variant added the same total
  std::vector<float> h(16 * (size_t)N);
  CHECK(hipMemcpy(h.data(), acc, sizeof(float) * 16 * (size_t)N, hipMemcpyDeviceToHost));
  double s = 0;
  for (float v : h) s += v;
  printf("sum over the aos table: %.0f\n", s);
  return 0;
}
