// Development tool: single-wave instruction issue / latency microbenchmark for gfx950 (MI355X).
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/issue_rate.hip -o build_variants/issue_rate && build_variants/issue_rate
// Every test runs ONE wave (or W waves on one SIMD / CU) executing an unrolled block of inline-asm instructions REP times,
// timed with s_memtime; prints shader cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP 2000

#define R4(x) x x x x
#define R16(x) R4(R4(x))

template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, uint64_t* cyc, float seed) {
  float a = seed + threadIdx.x, b = seed * 2.f, c = seed * 3.f, d = seed * 4.f, e = 0.5f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pa = {a, b}, pb = {c, d}, pc = {e, e}, pd = {a, c};
  uint64_t m0 = 0, m1 = 0;
  __shared__ float lds[1024];
  lds[threadIdx.x] = a;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REP; ++r) {
    if (MODE == 0) {  // 16 dependent v_fma_f32
      asm volatile(R16("v_fma_f32 %0, %0, %1, %1\n") : "+v"(a) : "v"(e));
    } else if (MODE == 1) {  // 16 v_fma_f32, 4 independent chains interleaved
      asm volatile(R4("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n")
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
    } else if (MODE == 2) {  // 16 dependent v_pk_fma_f32
      asm volatile(R16("v_pk_fma_f32 %0, %0, %1, %1\n") : "+v"(pa) : "v"(pc));
    } else if (MODE == 3) {  // 16 v_pk_fma_f32, 4 independent chains
      asm volatile(R4("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n")
                   : "+v"(pa), "+v"(pb), "+v"(pd), "+v"(pc) : "v"(pc));
    } else if (MODE == 4) {  // 16 dependent s_add_u32
      uint32_t s = r;
      asm volatile(R16("s_add_u32 %0, %0, 1\n") : "+s"(s) : : "scc");
      m0 += s;
    } else if (MODE == 5) {  // 16 x 2 chains: v_fma / s_add alternating (independent of each other)
      uint32_t s = r;
      asm volatile(R16("v_fma_f32 %0, %0, %2, %2\n s_add_u32 %1, %1, 1\n") : "+v"(a), "+s"(s) : "v"(e) : "scc");
      m0 += s;
    } else if (MODE == 6) {  // v_cmp -> s_and -> s_and_saveexec -> v_mov -> restore exec: the blend chain's round trip, x4
      asm volatile(R4("v_mul_f32 %1, %0, %2\n v_cmp_gt_f32 vcc, %1, %3\n s_and_b64 %4, vcc, exec\n s_and_saveexec_b64 %5, %4\n v_mov_b32 %0, %1\n s_or_b64 exec, exec, %5\n")
                   : "+v"(a), "+v"(b), "+v"(e), "+v"(c), "+s"(m0), "+s"(m1) : : "vcc", "scc");
    } else if (MODE == 7) {  // the same dependency carried by a select: v_mul, v_cmp, v_cndmask  x4
      asm volatile(R4("v_mul_f32 %1, %0, %2\n v_cmp_gt_f32 vcc, %1, %3\n v_cndmask_b32 %0, %0, %1, vcc\n")
                   : "+v"(a), "+v"(b) : "v"(e), "v"(c) : "vcc");
    } else if (MODE == 8) {  // 16 dependent v_mul_f32 (VOP2)
      asm volatile(R16("v_mul_f32 %0, %0, %1\n") : "+v"(a) : "v"(e));
    } else if (MODE == 9) {  // uniform ds_read_b32, wait, use as next address, x4 (LDS round trip)
      uint32_t addr = 0;
      float q;
      asm volatile(R4("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_and_b32 %1, 0xfc, %0\n") : "=&v"(q), "+v"(addr));
      a += q;
    } else if (MODE == 14) {  // 12 uniform ds_read_b128 back to back, one wait (the staging reads of a group)
      uint32_t addr = 0;
      float4 q0, q1, q2, q3;
      asm volatile(R4("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n") "s_waitcnt lgkmcnt(0)\n"
                   : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(addr));
      a += q0.x + q1.x + q2.x;
    } else if (MODE == 10) {  // 16 dependent v_ldexp / v_rndne / v_cvt mix (the unpackable part of exp)
      asm volatile(R4("v_rndne_f32 %0, %0\n v_cvt_i32_f32 %1, %0\n v_ldexp_f32 %0, %0, %1\n v_min_f32 %0, %0, %2\n") : "+v"(a), "+v"(b) : "v"(e));
    } else if (MODE == 11) {  // 16 independent v_fma (8 chains)
      float x0 = a, x1 = b, x2 = c, x3 = d, x4 = a + 1, x5 = b + 1, x6 = c + 1, x7 = d + 1;
      asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(e));
      a = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else if (MODE == 12) {  // v_cmp writing an SGPR pair, then s_and reading it, then v_cndmask reading the result  x4
      asm volatile(R4("v_cmp_gt_f32 %2, %0, %1\n s_and_b64 %2, %2, exec\n v_cndmask_b32 %0, %0, %1, %2\n")
                   : "+v"(a), "+v"(b), "+s"(m0) : : "scc");
    } else if (MODE == 13) {  // v_exp_f32 dependent x16 (quarter rate)
      asm volatile(R16("v_exp_f32 %0, %0\n") : "+v"(a));
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + pa.x + pa.y + pb.x + pd.y + (float)m0 + (float)m1;
}

template <int MODE>
static void run(const char* name, int instr_per_rep, int grid) {
  float* out;
  uint64_t* cyc;
  hipMalloc(&out, sizeof(float) * 64 * grid);
  hipMalloc(&cyc, sizeof(uint64_t) * grid);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, cyc, 1.0f);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, cyc, 1.0f);
  hipDeviceSynchronize();
  std::vector<uint64_t> h(grid);
  hipMemcpy(h.data(), cyc, sizeof(uint64_t) * grid, hipMemcpyDeviceToHost);
  uint64_t mx = 0;
  for (auto v : h) mx = v > mx ? v : mx;
  printf("%-62s grid %5d: %7.2f cycles / instruction (%d per block)\n", name, grid, (double)mx / REP / instr_per_rep, instr_per_rep);
  fflush(stdout);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int grid : {1, 4096}) {  // one wave alone; 4 waves per SIMD on the whole chip (1 wave workgroups, 1024 SIMDs)
    run<0>("v_fma_f32 dependent chain", 16, grid);
    run<8>("v_mul_f32 dependent chain", 16, grid);
    run<1>("v_fma_f32 4 independent chains", 16, grid);
    run<11>("v_fma_f32 8 independent chains", 16, grid);
    run<2>("v_pk_fma_f32 dependent chain", 16, grid);
    run<3>("v_pk_fma_f32 4 independent chains", 16, grid);
    run<4>("s_add_u32 dependent chain", 16, grid);
    run<5>("v_fma / s_add alternating (two chains)", 32, grid);
    run<6>("v_mul,v_cmp,s_and,s_and_saveexec,v_mov,s_or exec (serial)", 24, grid);
    run<7>("v_mul,v_cmp,v_cndmask (serial)", 12, grid);
    run<12>("v_cmp->sgpr, s_and, v_cndmask (serial)", 12, grid);
    run<9>("ds_read_b128 + waitcnt + v_and (serial)", 12, grid);
    run<14>("12 x ds_read_b128 uniform + one waitcnt", 13, grid);
    run<10>("v_rndne,v_cvt,v_ldexp,v_min (serial)", 16, grid);
    run<13>("v_exp_f32 dependent chain", 16, grid);
  }
  return 0;
}
