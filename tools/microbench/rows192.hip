// Microbenchmark (gfx950): one thread per 192-byte row (the SH record of a Gaussian), twelve dwordx4 loads and twelve
// dwordx4 stores per thread at a 192-byte lane stride, as K1 / K8+K9 access the SH arrays -- at different occupancies.
// Build: hipcc --offload-arch=gfx950 -O3 -o rows192 rows192.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WAVES_PER_EU, bool COALESCED>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU)))
k(const float4* __restrict__ in, float4* __restrict__ out, int P) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  float4 v[12];
  if (COALESCED) {  // lane l of a wave touches float4 k*64 + l of the wave's 12 KB block (what a transposed layout would give)
    const size_t base = (size_t)(idx & ~63) * 12 + (idx & 63);
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = in[base + (size_t)k * 64];
#pragma unroll
    for (int k = 0; k < 12; ++k) out[base + (size_t)k * 64] = make_float4(v[k].x * 2.f, v[k].y + 1.f, v[k].z, v[k].w);
  } else {
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = in[(size_t)idx * 12 + k];
#pragma unroll
    for (int k = 0; k < 12; ++k) out[(size_t)idx * 12 + k] = make_float4(v[k].x * 2.f, v[k].y + 1.f, v[k].z, v[k].w);
  }
}

// Third variant: the global accesses of the coalesced variant, but every thread ends up with ITS OWN row in registers
// (what K1 / K8+K9 need): the wave's 12 KB go through LDS, rows padded to 13 float4 so that both the transposing
// writes and the per-row reads are conflict-free; the results return the same way.
template <int WAVES_PER_EU>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU)))
k_lds(const float4* __restrict__ in, float4* __restrict__ out, int P) {
  __shared__ float4 tile[4][64 * 13];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;  // (P is a multiple of 256 here)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t base = (size_t)(idx & ~63) * 12;
  float4 v[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) v[k] = in[base + (size_t)k * 64 + lane];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int e = k * 64 + lane, r = e / 12, c = e - 12 * r;
    tile[w][r * 13 + c] = v[k];
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): single wave per tile, no barrier needed
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < 12; ++k) v[k] = tile[w][lane * 13 + k];
#pragma unroll
  for (int k = 0; k < 12; ++k) v[k] = make_float4(v[k].x * 2.f, v[k].y + 1.f, v[k].z, v[k].w);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < 12; ++k) tile[w][lane * 13 + k] = v[k];
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int e = k * 64 + lane, r = e / 12, c = e - 12 * r;
    out[base + (size_t)k * 64 + lane] = tile[w][r * 13 + c];
  }
}
template <int W>
static void run_lds(const float4* in, float4* out, int P) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  k_lds<W><<<(P + 255) / 256, 256>>>(in, out, P);
  (void)hipEventRecord(a);
  for (int i = 0; i < 10; ++i) k_lds<W><<<(P + 255) / 256, 256>>>(in, out, P);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  ms /= 10;
  printf("waves/SIMD %d %-10s: %.1f us, %.2f TB/s (384 B per row)\n", W, "via LDS", ms * 1e3, 384.0 * P / ms / 1e9);
}

template <int W, bool C>
static void run(const float4* in, float4* out, int P) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  k<W, C><<<(P + 255) / 256, 256>>>(in, out, P);
  (void)hipEventRecord(a);
  for (int i = 0; i < 10; ++i) k<W, C><<<(P + 255) / 256, 256>>>(in, out, P);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  ms /= 10;
  printf("waves/SIMD %d %-10s: %.1f us, %.2f TB/s (384 B per row)\n", W, C ? "coalesced" : "row/thread", ms * 1e3, 384.0 * P / ms / 1e9);
}

int main() {
  const int P = 1 << 20;
  float4 *in, *out;
  (void)hipMalloc(&in, (size_t)P * 192);
  (void)hipMalloc(&out, (size_t)P * 192);
  (void)hipMemset(in, 0, (size_t)P * 192);
  run<2, false>(in, out, P); run<3, false>(in, out, P); run<4, false>(in, out, P); run<6, false>(in, out, P); run<8, false>(in, out, P);
  run<2, true>(in, out, P); run<4, true>(in, out, P); run<8, true>(in, out, P);
  run_lds<2>(in, out, P); run_lds<3>(in, out, P);
  return 0;
}
