// Where do persistent workgroups land?  Launches `grid` workgroups of `T` threads that all stay resident (they spin
// until every workgroup has checked in) and prints how blockIdx maps to XCD / CU / SIMD / wave slot.
// Build: hipcc --offload-arch=gfx950 -O3 -o placement placement.hip ; run: ./placement <threads> <grid>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k(unsigned* rec, unsigned* arrived, unsigned total) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = 0.f;
  const unsigned wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
    rec[(blockIdx.x * nw + wave) * 2] = hw;
    rec[(blockIdx.x * nw + wave) * 2 + 1] = xcc;
  }
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total && spins < 20000000) ++spins;
  }
  __syncthreads();
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 256, grid = argc > 2 ? atoi(argv[2]) : 1024, lds = argc > 3 ? atoi(argv[3]) : 16384;
  const int nw = T / 64;
  unsigned *rec, *arrived;
  (void)hipMalloc(&rec, sizeof(unsigned) * 2 * grid * nw);
  (void)hipMalloc(&arrived, 4);
  (void)hipMemset(arrived, 0, 4);
  hipLaunchKernelGGL(k, dim3(grid), dim3(T), lds, 0, rec, arrived, (unsigned)grid);
  (void)hipDeviceSynchronize();
  std::vector<unsigned> h(2 * grid * nw);
  (void)hipMemcpy(h.data(), rec, h.size() * 4, hipMemcpyDeviceToHost);
  auto cu = [&](int b, int w) { unsigned hw = h[(b * nw + w) * 2], x = h[(b * nw + w) * 2 + 1] & 15; return (int)(x * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15) * 16); };
  auto simd = [&](int b, int w) { return (int)((h[(b * nw + w) * 2] >> 4) & 3); };
  auto slot = [&](int b, int w) { return (int)(h[(b * nw + w) * 2] & 15); };
  printf("threads %d grid %d lds %d\n", T, grid, lds);
  for (int b = 0; b < 12 && b < grid; ++b) {
    printf("  block %d: xcc %u cu %d waves(simd/slot):", b, h[(b * nw) * 2 + 1] & 15, cu(b, 0) % 4096);
    for (int w = 0; w < nw; ++w) printf(" %d/%d", simd(b, w), slot(b, w));
    printf("\n");
  }
  for (int stride : {8, 64, 128, 256, 512, 1024, 2048}) {
    if (stride >= grid) continue;
    int same = 0, n = 0;
    for (int b = 0; b + stride < grid; ++b, ++n) same += cu(b, 0) == cu(b + stride, 0);
    printf("  fraction of blocks with cu(b) == cu(b + %d): %.3f\n", stride, (double)same / n);
  }
  int xok = 0;
  for (int b = 0; b < grid; ++b) xok += (int)(h[(b * nw) * 2 + 1] & 15) == b % 8;
  printf("  xcc == blockIdx %% 8 for %.3f of the blocks\n", (double)xok / grid);
  return 0;
}
