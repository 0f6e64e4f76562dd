// gsr_binning.hip -- K3 (emit keys), K4 (stable LSD radix sort of 64-bit keys with 32-bit
// values), K5 (tile ranges).
//
// K4 replaces cub::DeviceRadixSort::SortPairs (rasterizer_impl.cu:256-261).  Every pass is
// three kernels (histogram -> per-bin scan -> scatter); there is no decoupled look-back,
// so no inter-workgroup hand-off inside a launch (per-XCD L2s are not coherent; a kernel
// boundary is the cheapest correct fence, ~1.5 us each).  Stability comes from ranking
// with wave64 ballots in key order, never from atomics.
#include "gsr_kernels.h"

namespace gsr {

__device__ __forceinline__ int f2i_sat(float v) {
  if (!(v == v)) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int)v;
}

// ----------------------------------------------------------------------------------
// K3: duplicateWithKeys, rasterizer_impl.cu:67-100.  The per-Gaussian write offset is
// block_offs[block] + (exclusive scan of tiles_touched inside the block).
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(GAUSS_BLOCK) emit_keys_kernel(int P, int gx, int gy, const int32_t* __restrict__ radii,
                                                               const Geom g, uint64_t* __restrict__ keys,
                                                               uint32_t* __restrict__ vals) {
  __shared__ uint32_t smem[GAUSS_BLOCK / 64 + 1];
  const int idx = (int)(blockIdx.x * GAUSS_BLOCK + threadIdx.x);
  const uint32_t n = idx < P ? g.tiles[idx] : 0u;
  uint32_t total;
  uint32_t off = g.block_offs[blockIdx.x] + block_excl_scan_u32<GAUSS_BLOCK>(n, &total, smem);
  if (n == 0) return;
  const float4 r1 = g.rec1[idx];
  const int radius = radii[idx];
  const float r = (float)radius;
  // getRect, auxiliary.h:46-56 (same inputs as in K1 => same rectangle)
  const uint32_t minx = (uint32_t)min(gx, max(0, f2i_sat((r1.x - r) / (float)TILE)));
  const uint32_t miny = (uint32_t)min(gy, max(0, f2i_sat((r1.y - r) / (float)TILE)));
  const uint32_t maxx = (uint32_t)min(gx, max(0, f2i_sat((r1.x + r + (float)TILE - 1.0f) / (float)TILE)));
  const uint32_t maxy = (uint32_t)min(gy, max(0, f2i_sat((r1.y + r + (float)TILE - 1.0f) / (float)TILE)));
  const uint64_t dbits = (uint64_t)__float_as_uint(r1.z);
  for (uint32_t y = miny; y < maxy; y++)
    for (uint32_t x = minx; x < maxx; x++) {
      const uint64_t key = ((uint64_t)(y * (uint32_t)gx + x) << 32) | dbits;
      keys[off] = key;
      vals[off] = (uint32_t)idx;
      off++;
    }
}

// ----------------------------------------------------------------------------------
// K4: stable LSD radix sort, `BITS`-bit digits (8 or 9), SORT_KPB keys per block.
// Pass = histogram -> per-bin scan -> scatter.  hist is bin-major: hist[bin*nblocks+block].
// ----------------------------------------------------------------------------------
template <int BITS>
__global__ void __launch_bounds__(SORT_THREADS) sort_hist_kernel(const uint64_t* __restrict__ keys, int64_t R, int shift,
                                                                uint32_t* __restrict__ hist, uint32_t nblocks) {
  constexpr int BINS = 1 << BITS;
  __shared__ uint32_t h[BINS];
  for (int i = threadIdx.x; i < BINS; i += SORT_THREADS) h[i] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_KPB;
#pragma unroll 4
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int64_t k = base + (int64_t)i * SORT_THREADS + threadIdx.x;
    if (k < R) atomicAdd(&h[(uint32_t)(keys[k] >> shift) & (BINS - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BINS; i += SORT_THREADS) hist[(size_t)i * nblocks + blockIdx.x] = h[i];
}

// One block per bin; exclusive scan of that bin's nblocks counts in place.
__global__ void __launch_bounds__(SORT_THREADS) sort_scan_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ bin_total,
                                                                uint32_t nblocks) {
  __shared__ uint32_t smem[SORT_THREADS / 64 + 1];
  uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nblocks; base += SORT_THREADS) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nblocks ? row[i] : 0u;
    uint32_t chunk;
    const uint32_t ex = block_excl_scan_u32<SORT_THREADS>(v, &chunk, smem);
    if (i < nblocks) row[i] = carry + ex;
    carry += chunk;
  }
  if (threadIdx.x == 0) bin_total[blockIdx.x] = carry;
}

// Stable scatter.  Wave w of a block owns 1024 consecutive keys and walks them 64 at a time in
// memory order; a key's rank among equal digits of its wave is (count of that digit in earlier
// iterations) + (lower lanes with the same digit in this iteration, from BITS ballots).  The
// block's keys/values are then permuted into digit order IN LDS and written out with consecutive
// threads covering consecutive sorted slots, so every digit's run is one contiguous global store
// stream (a direct scatter issues 64 isolated 8-byte stores per instruction on the low-entropy-free
// mantissa digits).
template <int BITS>
__global__ void __launch_bounds__(SORT_THREADS) sort_scatter_kernel(const uint64_t* __restrict__ keys_in,
                                                                   const uint32_t* __restrict__ vals_in,
                                                                   uint64_t* __restrict__ keys_out,
                                                                   uint32_t* __restrict__ vals_out, int64_t R, int shift,
                                                                   const uint32_t* __restrict__ hist,
                                                                   const uint32_t* __restrict__ bin_total,
                                                                   uint32_t nblocks) {
  constexpr int BINS = 1 << BITS;
  constexpr int NW = SORT_THREADS / 64;
  constexpr int BPT = BINS / SORT_THREADS;  // bins handled per thread (1 or 2)
  __shared__ uint32_t cnt[NW][BINS];        // per-wave digit counts -> per-wave local bases
  __shared__ uint32_t gbase[BINS];          // global position of the block's first key of each digit
  __shared__ uint32_t lexcl[BINS];          // position of each digit's run inside the block-sorted order
  __shared__ uint32_t smem[SORT_THREADS / 64 + 1];
  __shared__ uint64_t skey[SORT_KPB];
  __shared__ uint32_t sval[SORT_KPB];
  const int w = (int)(threadIdx.x >> 6), l = lane_id();
#pragma unroll
  for (int i = 0; i < NW; ++i)
#pragma unroll
    for (int b = 0; b < BPT; ++b) cnt[i][threadIdx.x * BPT + b] = 0;
  {
    // global base of every bin: exclusive scan of bin_total over bins, plus this block's offset inside the bin
    uint32_t t[BPT], tsum = 0;
#pragma unroll
    for (int b = 0; b < BPT; ++b) {
      t[b] = bin_total[threadIdx.x * BPT + b];
      tsum += t[b];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan_u32<SORT_THREADS>(tsum, &tot, smem);
#pragma unroll
    for (int b = 0; b < BPT; ++b) {
      const int bin = threadIdx.x * BPT + b;
      gbase[bin] = run + hist[(size_t)bin * nblocks + blockIdx.x];
      run += t[b];
    }
  }
  __syncthreads();

  const int64_t bbase = (int64_t)blockIdx.x * SORT_KPB;
  const int64_t wbase = bbase + (int64_t)w * (SORT_ITEMS * 64);
  uint64_t key[SORT_ITEMS];
  uint32_t val[SORT_ITEMS];
  uint16_t rank[SORT_ITEMS];
  const uint64_t lt_mask = (1ull << l) - 1ull;
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int64_t k = wbase + (int64_t)i * 64 + l;
    const int64_t kc = k < R ? k : R - 1;  // unconditional loads (clamped), validity handled below
    key[i] = keys_in[kc];
    val[i] = vals_in[kc];
  }
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int64_t k = wbase + (int64_t)i * 64 + l;
    const bool valid = k < R;
    const uint32_t d = (uint32_t)(key[i] >> shift) & (BINS - 1);
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      const uint64_t bb = __ballot((d >> b) & 1u);
      m &= ((d >> b) & 1u) ? bb : ~bb;
    }
    const uint32_t before = (uint32_t)__popcll(m & lt_mask);
    uint32_t old = 0;
    if (valid) old = cnt[w][d];
    // all reads of this iteration precede the leader's write (one wave, program order)
    if (valid && before == 0) cnt[w][d] = old + (uint32_t)__popcll(m);
    rank[i] = (uint16_t)(old + before);
  }
  __syncthreads();
  {
    // per digit: block total, exclusive prefix over the waves, and the digit's offset in block-sorted order
    uint32_t tsum = 0, tb[BPT];
#pragma unroll
    for (int b = 0; b < BPT; ++b) {
      const int bin = threadIdx.x * BPT + b;
      uint32_t run = 0;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const uint32_t c = cnt[i][bin];
        cnt[i][bin] = run;
        run += c;
      }
      tb[b] = run;
      tsum += run;
    }
    uint32_t tot;
    uint32_t run = block_excl_scan_u32<SORT_THREADS>(tsum, &tot, smem);
#pragma unroll
    for (int b = 0; b < BPT; ++b) {
      lexcl[threadIdx.x * BPT + b] = run;
      run += tb[b];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int64_t k = wbase + (int64_t)i * 64 + l;
    if (k < R) {
      const uint32_t d = (uint32_t)(key[i] >> shift) & (BINS - 1);
      const uint32_t lp = lexcl[d] + cnt[w][d] + rank[i];
      skey[lp] = key[i];
      sval[lp] = val[i];
    }
  }
  __syncthreads();
  const int nvalid = (int)min((int64_t)SORT_KPB, R - bbase);
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int j = i * SORT_THREADS + (int)threadIdx.x;
    if (j < nvalid) {
      const uint64_t kk = skey[j];
      const uint32_t d = (uint32_t)(kk >> shift) & (BINS - 1);
      const uint32_t pos = gbase[d] + ((uint32_t)j - lexcl[d]);
      keys_out[pos] = kk;
      vals_out[pos] = sval[j];
    }
  }
}

// ----------------------------------------------------------------------------------
// K5: identifyTileRanges, rasterizer_impl.cu:105-125 (ranges zeroed beforehand, :263-265).
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tile_ranges_kernel(int64_t L, const uint64_t* __restrict__ keys,
                                                         uint2* __restrict__ ranges) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= L) return;
  const uint32_t currtile = (uint32_t)(keys[idx] >> 32);
  if (idx == 0)
    ranges[currtile].x = 0;
  else {
    const uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
    if (currtile != prevtile) {
      ranges[prevtile].y = (uint32_t)idx;
      ranges[currtile].x = (uint32_t)idx;
    }
  }
  if (idx == L - 1) ranges[currtile].y = (uint32_t)L;
}

// ----------------------------------------------------------------------------------
// Work list for the blend kernels: tile ids ordered longest-list-first (bucketed by
// ceil(len/64)), empty tiles last.  One 1024-thread block; T is a few thousand.
// The order inside a bucket depends on LDS-atomic timing: it only affects scheduling,
// never results.
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) tile_worklist_kernel(int T, const uint2* __restrict__ ranges,
                                                            uint32_t* __restrict__ order, uint32_t* __restrict__ meta) {
  __shared__ uint32_t hist[WORK_BUCKETS + 1], cursor[WORK_BUCKETS + 1];
  for (int i = threadIdx.x; i <= WORK_BUCKETS; i += 1024) hist[i] = 0;
  __syncthreads();
  auto bucket_of = [](uint32_t len) -> uint32_t {
    if (len == 0) return WORK_BUCKETS;  // empty tiles: last
    const uint32_t c = (len + 63u) / 64u;
    return (uint32_t)(WORK_BUCKETS - 1) - min(c - 1u, (uint32_t)(WORK_BUCKETS - 1));
  };
  for (int t = threadIdx.x; t < T; t += 1024) {
    const uint2 r = ranges[t];
    atomicAdd(&hist[bucket_of(r.y - r.x)], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 0; i <= WORK_BUCKETS; ++i) {
      cursor[i] = run;
      run += hist[i];
    }
    meta[0] = cursor[WORK_BUCKETS];  // number of non-empty tiles
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 1024) {
    const uint2 r = ranges[t];
    const uint32_t pos = atomicAdd(&cursor[bucket_of(r.y - r.x)], 1u);
    order[pos] = (uint32_t)t;
  }
}

hipError_t launch_binning(hipStream_t s, int P, int64_t R, int W, int H, const int32_t* radii, const Geom& g,
                          const Binning& b, const Image& im) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  hipError_t e = hipMemsetAsync(im.ranges, 0, sizeof(uint2) * (size_t)gx * gy, s);
  if (e != hipSuccess) return e;
  if (R <= 0) {
    hipLaunchKernelGGL(tile_worklist_kernel, dim3(1), dim3(1024), 0, s, gx * gy, im.ranges, im.work_order, im.work_meta);
    return hipGetLastError();
  }
  const int nbg = (P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  hipLaunchKernelGGL(emit_keys_kernel, dim3(nbg), dim3(GAUSS_BLOCK), 0, s, P, gx, gy, radii, g, b.keys[0], b.vals[0]);
  int cur = 0;
  for (int p = 0; p < b.passes; ++p) {
    const int shift = p * b.digit_bits;
    if (b.digit_bits == 9) {
      hipLaunchKernelGGL(sort_hist_kernel<9>, dim3(b.nblocks), dim3(SORT_THREADS), 0, s, b.keys[cur], R, shift, b.hist,
                         b.nblocks);
      hipLaunchKernelGGL(sort_scan_kernel, dim3(512), dim3(SORT_THREADS), 0, s, b.hist, b.bin_total, b.nblocks);
      hipLaunchKernelGGL(sort_scatter_kernel<9>, dim3(b.nblocks), dim3(SORT_THREADS), 0, s, b.keys[cur], b.vals[cur],
                         b.keys[cur ^ 1], b.vals[cur ^ 1], R, shift, b.hist, b.bin_total, b.nblocks);
    } else {
      hipLaunchKernelGGL(sort_hist_kernel<8>, dim3(b.nblocks), dim3(SORT_THREADS), 0, s, b.keys[cur], R, shift, b.hist,
                         b.nblocks);
      hipLaunchKernelGGL(sort_scan_kernel, dim3(256), dim3(SORT_THREADS), 0, s, b.hist, b.bin_total, b.nblocks);
      hipLaunchKernelGGL(sort_scatter_kernel<8>, dim3(b.nblocks), dim3(SORT_THREADS), 0, s, b.keys[cur], b.vals[cur],
                         b.keys[cur ^ 1], b.vals[cur ^ 1], R, shift, b.hist, b.bin_total, b.nblocks);
    }
    cur ^= 1;
  }
  const int64_t nbr = (R + 255) / 256;
  hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)nbr), dim3(256), 0, s, R, b.keys[cur], im.ranges);
  hipLaunchKernelGGL(tile_worklist_kernel, dim3(1), dim3(1024), 0, s, gx * gy, im.ranges, im.work_order, im.work_meta);
  return hipGetLastError();
}

}  // namespace gsr
