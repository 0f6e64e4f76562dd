// What does one grid-wide barrier cost on this chip, next to what one small dependent kernel launch costs?
// (The 19 kernels between K1 and K6 average 8 us; the smallest -- a 1 KB-per-bin scan -- take 4.5-5 us whatever their body
//  does.  A single persistent kernel with grid barriers between its phases pays off only if a barrier + one dependent
//  memory round trip is well below that.)
//
//   phase k of every workgroup: write T words of "its" slice, BARRIER, read the slice of workgroup (b + 17) % grid and check
//   it (so that the barrier's release / acquire really has to make other XCDs' writes visible), repeat.
//
// Barrier forms:  0 flat     one agent-scope fetch-add per workgroup on ONE word, everybody polls that word
//                 1 per-XCD  arrive on the word of the workgroup's XCD (XCC_ID), the XCD's last arriver adds to the global
//                            word and, when it is the last of all, releases eight per-XCD flags; poll the own XCD's flag
// Spins are bounded (a workgroup that is not resident must not hang the box): `err` counts give-ups and wrong values.
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip ; run: ./grid_barrier [phases]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr long SPIN_CAP = 4000000;

struct Bar {
  unsigned* global;    // [0] arrivals of all workgroups (flat) / of XCD leaders (per-XCD)
  unsigned* xcd;       // [8 * 32] per-XCD arrival words, 128 bytes apart
  unsigned* flag;      // [8 * 32] per-XCD generation flags, 128 bytes apart
  unsigned* xcd_size;  // [8] workgroups resident on each XCD (counted by the kernel's first phase)
};

__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u; }

template <int MODE>
__device__ inline bool grid_barrier(const Bar& b, unsigned gen, unsigned grid, unsigned xcc, unsigned nxcd_active, unsigned mine) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    long spins = 0;
    if (MODE == 0) {
      __hip_atomic_fetch_add(b.global, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = gen * grid;
      while (__hip_atomic_load(b.global, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && spins < SPIN_CAP) ++spins;
    } else {
      const unsigned prev = __hip_atomic_fetch_add(b.xcd + xcc * 32, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (prev + 1u == gen * mine) {  // last of this XCD
        const unsigned p2 = __hip_atomic_fetch_add(b.global, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (p2 + 1u == gen * nxcd_active) {  // last of all
          for (int x = 0; x < 8; ++x) __hip_atomic_store(b.flag + x * 32, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      while (__hip_atomic_load(b.flag + xcc * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen && spins < SPIN_CAP) ++spins;
    }
    ok = spins < SPIN_CAP;
  }
  __syncthreads();
  return ok;
}

// WORK: 0 barrier only, 1 write a slice / barrier / read another workgroup's slice
template <int MODE, int WORK>
__global__ void __launch_bounds__(256) phases_kernel(Bar b, unsigned* data, unsigned phases, unsigned* err) {
  const unsigned grid = gridDim.x, T = blockDim.x, xcc = xcc_id();
  unsigned gen = 0, nx = 0, mine = 0;
  if (MODE == 1) {  // count the workgroups of every XCD first (flat barrier on a separate word: b.global + 32)
    if (threadIdx.x == 0) {
      atomicAdd(b.xcd_size + xcc, 1u);
      __hip_atomic_fetch_add(b.global + 32, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(b.global + 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < grid && spins < SPIN_CAP) ++spins;
      if (spins >= SPIN_CAP) atomicAdd(err, 1u);
    }
    __syncthreads();
    for (int x = 0; x < 8; ++x) nx += __hip_atomic_load(b.xcd_size + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    mine = __hip_atomic_load(b.xcd_size + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned bad = 0;
  for (unsigned k = 0; k < phases; ++k) {
    unsigned* slice = data + (size_t)(k & 1u) * grid * T;
    if (WORK) slice[blockIdx.x * T + threadIdx.x] = k * 131u + blockIdx.x;
    if (!grid_barrier<MODE>(b, ++gen, grid, xcc, nx, mine)) { bad |= 1u; break; }
    if (WORK) {
      const unsigned o = (blockIdx.x + 17u) % grid;
      const unsigned v = __builtin_nontemporal_load(slice + o * T + threadIdx.x);
      bad |= (v != k * 131u + o) ? 2u : 0u;
    }
  }
  if (bad) atomicAdd(err + (bad & 2u ? 1 : 0), 1u);
}

__global__ void __launch_bounds__(256) small_kernel(unsigned* data, unsigned k, unsigned grid_prev) {
  // the same phase as a kernel of its own: read what the previous launch's workgroup (b + 17) wrote, write the own slice
  const unsigned T = blockDim.x, grid = gridDim.x;
  const unsigned* prev = data + (size_t)((k + 1u) & 1u) * grid * T;
  unsigned* slice = data + (size_t)(k & 1u) * grid * T;
  const unsigned o = (blockIdx.x + 17u) % grid;
  slice[blockIdx.x * T + threadIdx.x] = prev[o * T + threadIdx.x] + 1u + grid_prev * 0u;
}

__global__ void empty_kernel() {}

template <int MODE, int WORK>
static int run(const char* name, int grid, unsigned phases, Bar b, unsigned* data, unsigned* err, unsigned* raw, size_t raw_bytes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  unsigned herr[2] = {0, 0};
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(raw, 0, raw_bytes));
    CK(hipMemset(err, 0, 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((phases_kernel<MODE, WORK>), dim3(grid), dim3(256), 0, 0, b, data, phases, err);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
    unsigned h[2]; CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
    herr[0] += h[0]; herr[1] += h[1];
  }
  printf("  %-28s grid %4d: %7.3f us per phase  (%u phases, give-ups %u, wrong values %u)\n", name, grid, best * 1e3f / phases, phases, herr[0], herr[1]);
  fflush(stdout);
  return herr[0] ? 2 : 0;
}

int main(int argc, char** argv) {
  const unsigned phases = argc > 1 ? (unsigned)atoi(argv[1]) : 200u;
  const int grids[] = {256, 512, 1024};
  const int max_grid = 1024;
  unsigned *raw, *data, *err;
  const size_t raw_words = 64 + 8 * 32 + 8 * 32 + 8;
  CK(hipMalloc(&raw, raw_words * 4));
  CK(hipMalloc(&data, (size_t)2 * max_grid * 256 * 4));
  CK(hipMalloc(&err, 8));
  CK(hipMemset(data, 0, (size_t)2 * max_grid * 256 * 4));
  Bar b = {raw, raw + 64, raw + 64 + 256, raw + 64 + 512};
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("%s, %d CUs\n", prop.name, prop.multiProcessorCount);

  const bool barriers = argc > 2 ? atoi(argv[2]) != 0 : true;
  if (barriers) printf("one persistent kernel, grid barrier between phases:\n");
  for (int g : grids) {
    if (!barriers) break;
    if (run<0, 0>("flat, barrier only", g, phases, b, data, err, raw, raw_words * 4) == 2) break;
    run<0, 1>("flat, write/barrier/read", g, phases, b, data, err, raw, raw_words * 4);
    if (run<1, 0>("per-XCD, barrier only", g, phases, b, data, err, raw, raw_words * 4) == 2) break;
    run<1, 1>("per-XCD, write/barrier/read", g, phases, b, data, err, raw, raw_words * 4);
  }

  printf("one kernel launch per phase (same stream):\n");
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int g : grids) {
    for (int form = 0; form < 3; ++form) {  // 0 empty kernels, 1 read-previous / write-own kernels, 2 the same through a hipGraph
      float best = 1e30f;
      hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
      if (form == 2) {
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (unsigned k = 0; k < phases; ++k) hipLaunchKernelGGL(small_kernel, dim3(g), dim3(256), 0, s, data, k, 0u);
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      }
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, s));
        if (form == 2) CK(hipGraphLaunch(exec, s));
        else for (unsigned k = 0; k < phases; ++k) {
          if (form == 0) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), 0, s);
          else hipLaunchKernelGGL(small_kernel, dim3(g), dim3(256), 0, s, data, k, 0u);
        }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      if (exec) { CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); }
      printf("  %-28s grid %4d: %7.3f us per launch\n", form == 0 ? "empty kernels" : form == 1 ? "read/write kernels" : "read/write kernels, hipGraph", g, best * 1e3f / phases);
      fflush(stdout);
    }
  }
  return 0;
}
