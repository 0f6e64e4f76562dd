// gsr_binning.hip -- K3 (emit), K4 (sort), K5 (tile ranges) and the blend work list.
//
// The reference builds one 64-bit key (tile | depth bits) per (Gaussian, tile) instance and runs a
// stable radix sort over all R instances on 32 + getHigherMsb(T) key bits (rasterizer_impl.cu:67-100,
// 253-261: 45 bits at 1080p).  Every instance of a Gaussian carries the SAME depth, so the identical
// order is obtained far cheaper as two stable sorts (a stable sort by the minor key followed by a stable
// sort by the major key):
//
//   1. stable LSD radix sort of the P Gaussians by their 32 depth bits (value = index, culled ones
//      last).  P is ~5x smaller than R, and it runs inside gsr_preprocess, i.e. under the host's
//      blocking readback of num_rendered (launch_depth_order);
//   2. instances are emitted in that Gaussian order, so the instance array is already depth-ordered,
//      ties in emission order = ascending Gaussian index exactly like the reference's;
//   3. stable LSD radix sort of the R instances by tile id only: 13 bits at 1080p = 2 passes over
//      8-byte (tile, index) pairs instead of 6 passes over 12-byte pairs.
//
// A radix pass is three kernels (histogram -> per-bin scan -> scatter); there is no decoupled
// look-back, so no inter-workgroup hand-off inside a launch (per-XCD L2s are not coherent; a kernel
// boundary is the cheapest correct fence).  Stability comes from ranking with wave64 ballots in key
// order, never from atomics.
#include "gsr_kernels.h"

namespace gsr {

__device__ __forceinline__ int f2i_sat(float v) {
  if (!(v == v)) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int)v;
}

// ----------------------------------------------------------------------------------
// Stable LSD radix sort pass on 32-bit keys with 32-bit values, 8-bit digits (256 bins; a pass may
// use fewer significant bits through `mask`), SORT_KPB keys per block.
// hist is bin-major: hist[bin*nblocks+block].  IOTA: the input values are the element indices.
// ----------------------------------------------------------------------------------
constexpr int RBITS = 8;
constexpr int RBINS = 1 << RBITS;

// `publish_dst` (first depth pass only): block 0 also reduces the per-block partials K1 left (num_rendered, range of the
// depth keys) and writes the four words into the caller's pinned, device-mapped host buffer -- the readback of
// gsr_preprocess without a copy command of its own (a 4 us blit kernel plus a 6 us bubble behind it before) and without a
// header to clear in front of K1.
template <class K>
__global__ void __launch_bounds__(SORT_THREADS) sort_hist_kernel(const K* __restrict__ keys, int64_t n, int shift,
                                                                uint32_t mask, uint32_t* __restrict__ hist,
                                                                uint32_t nblocks, const uint4* __restrict__ publish_src,
                                                                uint32_t publish_count, uint32_t* __restrict__ publish_dst,
                                                                uint32_t publish_seq) {
  __shared__ uint32_t h[RBINS];
  if (publish_dst != nullptr && blockIdx.x == 0) {
    __shared__ unsigned long long psum[SORT_THREADS / 64];
    __shared__ uint32_t pmax[SORT_THREADS / 64], pinv[SORT_THREADS / 64];
    unsigned long long sum = 0;
    uint32_t kmax = 0, kinv = 0;
    for (uint32_t i = threadIdx.x; i < publish_count; i += SORT_THREADS) {
      const uint4 v = publish_src[i];
      sum += v.x;
      kmax = max(kmax, v.y);
      kinv = max(kinv, v.z);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      sum += __shfl_xor(sum, d, 64);
      kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d, 64));
      kinv = max(kinv, (uint32_t)__shfl_xor((int)kinv, d, 64));
    }
    if (lane_id() == 0) {
      psum[threadIdx.x >> 6] = sum;
      pmax[threadIdx.x >> 6] = kmax;
      pinv[threadIdx.x >> 6] = kinv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < SORT_THREADS / 64; ++w) {
        sum += psum[w];
        kmax = max(kmax, pmax[w]);
        kinv = max(kinv, pinv[w]);
      }
      publish_dst[0] = (uint32_t)sum;
      publish_dst[1] = (uint32_t)(sum >> 32);
      publish_dst[GEOM_HDR_KEYMAX] = kmax;
      publish_dst[GEOM_HDR_KEYINVMAX] = kinv;
      __threadfence_system();
      // the host spins on this word (fine-grained pinned memory): no event, hence no barrier packet in the stream
      __hip_atomic_store(publish_dst + GEOM_HDR_FINAL, publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_KPB;
  // all 16 loads of a thread are issued back to back (unconditional, index clamped): the kernel runs one 4-wave block
  // per CU and is bound by memory latency, not by bandwidth or the LDS atomics
  uint32_t kv[SORT_ITEMS];
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int64_t k = base + (int64_t)i * SORT_THREADS + threadIdx.x;
    kv[i] = (uint32_t)keys[k < n ? k : n - 1];
  }
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int64_t k = base + (int64_t)i * SORT_THREADS + threadIdx.x;
    if (k < n) atomicAdd(&h[(kv[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// One block per bin; exclusive scan of that bin's nblocks counts in place.  Every thread takes SCAN_PER consecutive counts
// per round (the rows of a deep scene hold > 10^4 counts: one count per thread and round made this kernel as long as a
// histogram pass).
constexpr int SCAN_PER = 8;
__global__ void __launch_bounds__(SORT_THREADS) sort_scan_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ bin_total,
                                                                uint32_t nblocks) {
  __shared__ uint32_t smem[SORT_THREADS / 64 + 1];
  uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nblocks; base += SORT_THREADS * SCAN_PER) {
    const uint32_t i0 = base + threadIdx.x * SCAN_PER;
    uint32_t v[SCAN_PER], sum = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER; ++j) {
      v[j] = i0 + j < nblocks ? row[i0 + j] : 0u;
      sum += v[j];
    }
    uint32_t chunk;
    uint32_t run = carry + block_excl_scan_u32<SORT_THREADS>(sum, &chunk, smem);
#pragma unroll
    for (int j = 0; j < SCAN_PER; ++j) {
      if (i0 + j < nblocks) row[i0 + j] = run;
      run += v[j];
    }
    carry += chunk;
  }
  if (threadIdx.x == 0) bin_total[blockIdx.x] = carry;
}

// Stable scatter.  Wave w of a block owns 1024 consecutive keys and walks them 64 at a time in memory
// order; a key's rank among equal digits of its wave is (count of that digit in earlier iterations) +
// (lower lanes with the same digit in this iteration, from 8 ballots).  The block's pairs are then
// permuted into digit order IN LDS and written out with consecutive threads covering consecutive
// sorted slots, so every digit's run is one contiguous global store stream.
template <bool IOTA, class K>
__global__ void __launch_bounds__(SORT_THREADS) sort_scatter_kernel(const K* __restrict__ keys_in,
                                                                   const uint32_t* __restrict__ vals_in,
                                                                   K* __restrict__ keys_out,
                                                                   uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                                   uint32_t mask, const uint32_t* __restrict__ hist,
                                                                   const uint32_t* __restrict__ bin_total,
                                                                   uint32_t nblocks) {
  constexpr int NW = SORT_THREADS / 64;
  __shared__ uint32_t cnt[NW][RBINS];   // per-wave digit counts -> per-wave local bases
  __shared__ uint32_t gbase[RBINS];     // global position of the block's first key of each digit
  __shared__ uint32_t lexcl[RBINS];     // position of each digit's run inside the block-sorted order
  __shared__ uint32_t smem[SORT_THREADS / 64 + 1];
  __shared__ uint32_t skey[SORT_KPB];
  __shared__ uint32_t sval[SORT_KPB];
  const int w = (int)(threadIdx.x >> 6), l = lane_id();
  const int64_t bbase = (int64_t)blockIdx.x * SORT_KPB;
  const int64_t wbase = bbase + (int64_t)w * (SORT_ITEMS * 64);
  uint32_t key[SORT_ITEMS], val[SORT_ITEMS];
  uint16_t rank[SORT_ITEMS];
  const uint64_t lt_mask = (1ull << l) - 1ull;
  // the block's pairs are requested first, so that they travel while the digit bases below are loaded and scanned
  // (no load crosses a barrier on its own)
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int64_t k = wbase + (int64_t)i * 64 + l;
    const int64_t kc = k < n ? k : n - 1;  // unconditional loads (clamped), validity handled below
    key[i] = (uint32_t)keys_in[kc];
    val[i] = IOTA ? (uint32_t)kc : vals_in[kc];
  }
  const uint32_t my_hist = hist[(size_t)threadIdx.x * nblocks + blockIdx.x];
#pragma unroll
  for (int i = 0; i < NW; ++i) cnt[i][threadIdx.x] = 0;
  {
    uint32_t tot;
    const uint32_t run = block_excl_scan_u32<SORT_THREADS>(bin_total[threadIdx.x], &tot, smem);
    gbase[threadIdx.x] = run + my_hist;
  }
  __syncthreads();

#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int64_t k = wbase + (int64_t)i * 64 + l;
    const bool valid = k < n;
    const uint32_t d = (key[i] >> shift) & mask;
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RBITS; ++b) {
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
    uint32_t run = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const uint32_t c = cnt[i][threadIdx.x];
      cnt[i][threadIdx.x] = run;
      run += c;
    }
    uint32_t tot;
    lexcl[threadIdx.x] = block_excl_scan_u32<SORT_THREADS>(run, &tot, smem);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int64_t k = wbase + (int64_t)i * 64 + l;
    if (k < n) {
      const uint32_t d = (key[i] >> shift) & mask;
      const uint32_t lp = lexcl[d] + cnt[w][d] + rank[i];
      skey[lp] = key[i];
      sval[lp] = val[i];
    }
  }
  __syncthreads();
  const int nvalid = (int)min((int64_t)SORT_KPB, n - bbase);
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; ++i) {
    const int j = i * SORT_THREADS + (int)threadIdx.x;
    if (j < nvalid) {
      const uint32_t kk = skey[j];
      const uint32_t d = (kk >> shift) & mask;
      const uint32_t pos = gbase[d] + ((uint32_t)j - lexcl[d]);
      keys_out[pos] = (K)kk;
      vals_out[pos] = sval[j];
    }
  }
}

// Sorts (keys[0], vals[0]) by key bits [0, sum(digits)); result in buffer (npass & 1).  Passes [p0, npass) of the
// sequence are launched (pass p reads buffer p & 1), so a caller can enqueue a prefix of the passes, decide how
// many more are needed and continue.
template <class K>
static void radix_sort_pairs(hipStream_t s, K* const keys[2], uint32_t* const vals[2], int64_t n, int npass,
                             const int* digit_bits, uint32_t* hist, uint32_t* bin_total, bool iota_first, int p0 = 0,
                             const uint4* publish_src = nullptr, uint32_t publish_count = 0, uint32_t* publish_dst = nullptr,
                             uint32_t publish_seq = 0) {
  const uint32_t nblocks = (uint32_t)((n + SORT_KPB - 1) / SORT_KPB);
  int cur = p0 & 1, shift = 0;
  for (int p = 0; p < p0; ++p) shift += digit_bits[p];
  for (int p = p0; p < npass; ++p) {
    const uint32_t mask = (1u << digit_bits[p]) - 1u;
    const bool pub = p == p0 && publish_dst != nullptr;
    hipLaunchKernelGGL(sort_hist_kernel<K>, dim3(nblocks), dim3(SORT_THREADS), 0, s, (const K*)keys[cur], n, shift, mask, hist, nblocks,
                       pub ? publish_src : nullptr, publish_count, pub ? publish_dst : nullptr, publish_seq);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(RBINS), dim3(SORT_THREADS), 0, s, hist, bin_total, nblocks);
    if (p == 0 && iota_first)
      hipLaunchKernelGGL((sort_scatter_kernel<true, K>), dim3(nblocks), dim3(SORT_THREADS), 0, s, (const K*)keys[cur],
                         (const uint32_t*)vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, mask, hist, bin_total, nblocks);
    else
      hipLaunchKernelGGL((sort_scatter_kernel<false, K>), dim3(nblocks), dim3(SORT_THREADS), 0, s, (const K*)keys[cur],
                         (const uint32_t*)vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, mask, hist, bin_total, nblocks);
    shift += digit_bits[p];
    cur ^= 1;
  }
}

// ----------------------------------------------------------------------------------
// Depth order of the Gaussians + prefix of tiles_touched in that order.
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(GAUSS_BLOCK) sorted_block_sums_kernel(int P, int gx, const uint32_t* __restrict__ order,
                                                                       const uint2* __restrict__ rect,
                                                                       uint32_t* __restrict__ wh_sorted,
                                                                       uint32_t* __restrict__ org_sorted,
                                                                       uint32_t* __restrict__ block_sums,
                                                                       uint32_t* __restrict__ hdr, uint32_t final_buf) {
  __shared__ uint32_t smem[GAUSS_BLOCK / 64 + 1];
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr[GEOM_HDR_FINAL] = final_buf;  // for emit_keys_kernel (gsr_bin)
  const int i = (int)(blockIdx.x * GAUSS_BLOCK + threadIdx.x);
  // THE gather of the binning: the tile rectangle of the i-th Gaussian in depth order (8 bytes; its area is the tile count).
  // It is left in depth order for the emit kernel, which then reads everything coalesced.
  const uint2 rc = i < P ? rect[order[i]] : make_uint2(0u, 0u);
  const uint32_t n = (rc.y & 0xffffu) * (rc.y >> 16);
  if (i < P) {
    wh_sorted[i] = rc.y;                                            // width | height << 16
    org_sorted[i] = (rc.x >> 16) * (uint32_t)gx + (rc.x & 0xffffu);  // tile id of the rectangle's first tile
  }
  uint32_t total;
  (void)block_excl_scan_u32<GAUSS_BLOCK>(n, &total, smem);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024) scan_blocks_kernel(const uint32_t* __restrict__ sums, uint32_t* __restrict__ offs,
                                                          int nb) {
  __shared__ uint32_t smem[1024 / 64 + 1];
  uint32_t carry = 0;
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const uint32_t v = i < nb ? sums[i] : 0u;
    uint32_t chunk_total;
    const uint32_t ex = block_excl_scan_u32<1024>(v, &chunk_total, smem);
    if (i < nb) offs[i] = carry + ex;
    carry += chunk_total;
  }
}

// exported for the other translation units (gsr_knn.hip)
void radix_sort_pairs_u32(hipStream_t s, uint32_t* const keys[2], uint32_t* const vals[2], int64_t n, int npass,
                          const int* digit_bits, uint32_t* hist, uint32_t* bin_total, bool iota_first) {
  radix_sort_pairs<uint32_t>(s, keys, vals, n, npass, digit_bits, hist, bin_total, iota_first);
}

// Depth order of the Gaussians: passes [p0, p1) of the 8-bit LSD sort on the depth bits.  Only the bits in which
// the smallest and the largest key differ need sorting (the rest is a common prefix), so the caller enqueues the
// first passes, learns the key range from K1 and adds what is missing.
static const int kDepthDigits[4] = {8, 8, 8, 8};
hipError_t launch_depth_passes(hipStream_t s, int P, const Geom& g, int p0, int p1, uint32_t* publish_dst, uint32_t publish_seq) {
  radix_sort_pairs<uint32_t>(s, g.dkey, g.dval, P, p1, kDepthDigits, g.ghist, g.gbin_total, true, p0, g.k1_partials,
                   (uint32_t)((P + GAUSS_BLOCK - 1) / GAUSS_BLOCK), publish_dst, publish_seq);
  return hipGetLastError();
}
// After `passes` passes: tile counts gathered into depth order (+ their per-block sums and the prefix of those).
hipError_t launch_depth_finish(hipStream_t s, int P, const Geom& g, int passes, int gx) {
  const int fin = passes & 1;
  const int nb = (P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  // dkey[fin ^ 1] / dval[fin ^ 1] (the input of the last pass) are dead: reuse them for the depth-ordered rectangles
  hipLaunchKernelGGL(sorted_block_sums_kernel, dim3(nb), dim3(GAUSS_BLOCK), 0, s, P, gx, g.dval[fin], g.rect, g.dkey[fin ^ 1],
                     g.dval[fin ^ 1], g.block_sums, reinterpret_cast<uint32_t*>(g.total), (uint32_t)fin);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, s, g.block_sums, g.block_offs, nb);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// K3: duplicateWithKeys, rasterizer_impl.cu:67-100, in depth order of the Gaussians.  A block
// takes 256 consecutive Gaussians of that order; their instances occupy one contiguous run of the
// output (block_offs[block] + in-block exclusive scan).  The threads then walk the run SLOT BY SLOT
// (thread t writes slots t, t + 256, ...: coalesced stores however uneven the rectangles are): a
// binary search in the block's offset table finds the Gaussian a slot belongs to, the slot's position
// inside its rectangle gives the tile, rows first as the reference's loops do.
// Only the tile id is written as key (the depth is implied by the position).
// ----------------------------------------------------------------------------------
template <class K>
__global__ void __launch_bounds__(GAUSS_BLOCK) emit_keys_kernel(int P, int gx, int gy, const int32_t* __restrict__ radii,
                                                               const Geom g, K* __restrict__ tkeys,
                                                               uint32_t* __restrict__ vals, uint2* __restrict__ ranges) {
  __shared__ uint32_t smem[GAUSS_BLOCK / 64 + 1];
  __shared__ uint32_t s_off[GAUSS_BLOCK + 1], s_idx[GAUSS_BLOCK], s_org[GAUSS_BLOCK], s_w[GAUSS_BLOCK];
  const int i = (int)(blockIdx.x * GAUSS_BLOCK + threadIdx.x);
  // the ranges of tiles no instance falls into stay (0,0) (reference: cudaMemset, rasterizer_impl.cu:263-265);
  // cleared here, two launches ahead of tile_ranges_kernel, instead of by a memset of its own
  if (i < gx * gy) ranges[i] = make_uint2(0u, 0u);
  if ((int)blockIdx.x * GAUSS_BLOCK >= P) return;  // (extra blocks only clear ranges)
  const uint32_t fin = reinterpret_cast<const uint32_t*>(g.total)[GEOM_HDR_FINAL];  // side holding the depth order
  const uint32_t idx = i < P ? g.dval[fin][i] : 0u;
  // the rectangle of the Gaussian in depth order, left there by sorted_block_sums_kernel: no gather in this kernel
  const uint32_t wh = i < P ? g.dkey[fin ^ 1u][i] : 0u;
  const uint32_t w = max(wh & 0xffffu, 1u), n = (wh & 0xffffu) * (wh >> 16);  // n = tiles_touched
  const uint32_t org = i < P ? g.dval[fin ^ 1u][i] : 0u;
  uint32_t total;
  const uint32_t boff = g.block_offs[blockIdx.x];
  const uint32_t off = block_excl_scan_u32<GAUSS_BLOCK>(n, &total, smem);
  s_off[threadIdx.x] = off;
  s_idx[threadIdx.x] = idx;
  s_org[threadIdx.x] = org;
  s_w[threadIdx.x] = w;
  if (threadIdx.x == 0) s_off[GAUSS_BLOCK] = total;
  __syncthreads();
  for (uint32_t s = threadIdx.x; s < total; s += GAUSS_BLOCK) {
    // the last j with s_off[j] <= s: s_off[j + 1] > s, so Gaussian j owns at least one slot
    uint32_t lo = 0, hi = GAUSS_BLOCK;
#pragma unroll
    for (int step = 0; step < 8; ++step) {  // GAUSS_BLOCK == 256
      const uint32_t mid = (lo + hi) >> 1;
      if (s_off[mid] <= s) lo = mid; else hi = mid;
    }
    const uint32_t k = s - s_off[lo], wj = s_w[lo];
    // row = k / wj without an integer division: k, wj < 2^24 (at most gx * gy tiles), one correction step each way
    uint32_t row = (uint32_t)((float)k * __builtin_amdgcn_rcpf((float)wj));
    if (row * wj > k) row--;
    if ((row + 1u) * wj <= k) row++;
    const uint32_t col = k - row * wj;
    tkeys[boff + s] = (K)(s_org[lo] + row * (uint32_t)gx + col);
    vals[boff + s] = s_idx[lo];
  }
}

// ----------------------------------------------------------------------------------
// K5: identifyTileRanges, rasterizer_impl.cu:105-125 (ranges zeroed beforehand, :263-265).
// ----------------------------------------------------------------------------------
// RANGE_PER consecutive instances per thread, read with one or two 16-byte loads (the sorted ids are a plain stream: with one
// instance per thread the kernel ran at 1.5 TB/s on the deep-tile scene, 67 us for 98 MB).
constexpr int RANGE_PER = 8;
template <class K>
__global__ void __launch_bounds__(256) tile_ranges_kernel(int64_t L, const K* __restrict__ tkeys,
                                                         uint2* __restrict__ ranges) {
  const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * RANGE_PER;
  if (base >= L) return;
  K k[RANGE_PER];
  if (base + RANGE_PER <= L) {  // (the key buffers are 256-byte aligned and base is a multiple of 8: 16-byte aligned loads)
    constexpr int NV = (int)(sizeof(K) * RANGE_PER / sizeof(uint4));
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(tkeys + base);
    uint4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = src[i];
    __builtin_memcpy(k, v, sizeof(k));
  } else {
#pragma unroll
    for (int i = 0; i < RANGE_PER; ++i) k[i] = base + i < L ? tkeys[base + i] : (K)0;
  }
  uint32_t prev = base ? (uint32_t)tkeys[base - 1] : 0u;
#pragma unroll
  for (int i = 0; i < RANGE_PER; ++i) {
    const int64_t idx = base + i;
    if (idx >= L) break;
    const uint32_t cur = (uint32_t)k[i];
    if (idx == 0) {
      ranges[cur].x = 0;
    } else if (cur != prev) {
      ranges[prev].y = (uint32_t)idx;
      ranges[cur].x = (uint32_t)idx;
    }
    if (idx == L - 1) ranges[cur].y = (uint32_t)L;
    prev = cur;
  }
}

// ----------------------------------------------------------------------------------
// Work list for the blend kernels: tile ids ordered longest-list-first (bucketed by
// ceil(len/64)), empty tiles last.  One 1024-thread block; T is a few thousand.
// The order inside a bucket depends on LDS-atomic timing: it only affects scheduling,
// never results.
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) tile_worklist_kernel(int T, const uint2* __restrict__ ranges,
                                                            uint32_t* __restrict__ order, uint32_t* __restrict__ meta,
                                                            uint32_t* __restrict__ queues, uint32_t* __restrict__ est) {
  // Counting sort of the tiles by bucket.  Most tiles of an image fall into a handful of buckets, and LDS atomics on one
  // address serialise, so every bucket has WORK_SUB counters (chosen by the thread's lane): the order inside a bucket is
  // free anyway, and the sort's time stops growing with the number of tiles per bucket.
  constexpr int WORK_SUB = 16, NCNT = (WORK_BUCKETS + 1) * WORK_SUB;
  __shared__ uint32_t cnt[NCNT];
  __shared__ uint32_t smem[1024 / 64 + 1];
  // per-quadrant work counters of the forward blend (the items of a quadrant combine their counts with atomicMax)
  for (int i = threadIdx.x; i < 4 * T; i += 1024) est[i] = 0u;
  // work-queue cursors and retire counters of the three blend kernels start at zero; each blend launch leaves its
  // own zeroed again (gsr_blend.hip: retire_queue), so this is the only place that clears them
  for (int i = threadIdx.x; i < QUEUE_KINDS * QUEUE_LINES; i += 1024) {
    queues[(size_t)i * QUEUE_STRIDE] = 0u;      // taken from the front / counter
    queues[(size_t)i * QUEUE_STRIDE + 1] = 0u;  // (second word of the line: spare)
  }
  for (int i = threadIdx.x; i < NCNT; i += 1024) cnt[i] = 0;
  __syncthreads();
  const uint32_t sub = threadIdx.x & (WORK_SUB - 1);
  auto bucket_of = [](uint32_t len) -> uint32_t {
    if (len == 0) return WORK_BUCKETS;  // empty tiles: last
    const uint32_t c = (len + 63u) / 64u;
    return (uint32_t)(WORK_BUCKETS - 1) - min(c - 1u, (uint32_t)(WORK_BUCKETS - 1));
  };
  for (int t = threadIdx.x; t < T; t += 1024) {
    const uint2 r = ranges[t];
    atomicAdd(&cnt[bucket_of(r.y - r.x) * WORK_SUB + sub], 1u);
  }
  __syncthreads();
  {  // exclusive scan over the NCNT counters in (bucket, sub) order: counts -> cursors
    uint32_t carry = 0;
    for (int base = 0; base < NCNT; base += 1024) {
      const int i = base + (int)threadIdx.x;
      const uint32_t v = i < NCNT ? cnt[i] : 0u;
      uint32_t chunk;
      const uint32_t ex = block_excl_scan_u32<1024>(v, &chunk, smem);
      if (i < NCNT) cnt[i] = carry + ex;
      if (i == WORK_BUCKETS * WORK_SUB) meta[0] = carry + ex;  // number of non-empty tiles
      carry += chunk;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 1024) {
    const uint2 r = ranges[t];
    const uint32_t pos = atomicAdd(&cnt[bucket_of(r.y - r.x) * WORK_SUB + sub], 1u);
    order[pos] = (uint32_t)t;
  }
}

hipError_t launch_binning(hipStream_t s, int P, int64_t R, int W, int H, const int32_t* radii, const Geom& g,
                          const Binning& b, const Image& im) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  if (R <= 0) {
    hipError_t e = hipMemsetAsync(im.ranges, 0, sizeof(uint2) * (size_t)gx * gy, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tile_worklist_kernel, dim3(1), dim3(1024), 0, s, gx * gy, im.ranges, im.work_order, im.work_meta,
                       im.queue_heads, im.work_est);
    return hipGetLastError();
  }
  const int nbg = (P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  const dim3 ge(max(nbg, (gx * gy + GAUSS_BLOCK - 1) / GAUSS_BLOCK)), gr((unsigned)((R + 256 * RANGE_PER - 1) / (256 * RANGE_PER)));
  if (b.key_bytes == 2) {  // tile ids fit 16 bits: 6 instead of 8 bytes per sorted pair
    uint16_t* const tk[2] = {(uint16_t*)b.tkey[0], (uint16_t*)b.tkey[1]};
    hipLaunchKernelGGL(emit_keys_kernel<uint16_t>, ge, dim3(GAUSS_BLOCK), 0, s, P, gx, gy, radii, g, tk[0], b.vals[0], im.ranges);
    radix_sort_pairs<uint16_t>(s, tk, b.vals, R, b.passes, b.digit_bits, b.hist, b.bin_total, false);
    hipLaunchKernelGGL(tile_ranges_kernel<uint16_t>, gr, dim3(256), 0, s, R, (const uint16_t*)tk[b.final_buf], im.ranges);
  } else {
    uint32_t* const tk[2] = {(uint32_t*)b.tkey[0], (uint32_t*)b.tkey[1]};
    hipLaunchKernelGGL(emit_keys_kernel<uint32_t>, ge, dim3(GAUSS_BLOCK), 0, s, P, gx, gy, radii, g, tk[0], b.vals[0], im.ranges);
    radix_sort_pairs<uint32_t>(s, tk, b.vals, R, b.passes, b.digit_bits, b.hist, b.bin_total, false);
    hipLaunchKernelGGL(tile_ranges_kernel<uint32_t>, gr, dim3(256), 0, s, R, (const uint32_t*)tk[b.final_buf], im.ranges);
  }
  hipLaunchKernelGGL(tile_worklist_kernel, dim3(1), dim3(1024), 0, s, gx * gy, im.ranges, im.work_order, im.work_meta,
                     im.queue_heads, im.work_est);
  return hipGetLastError();
}

// Test-only: rebuild the reference's 64-bit sorted keys, (tile << 32) | depth bits.
template <class K>
__global__ void __launch_bounds__(256) export_keys_kernel(int64_t R, const K* __restrict__ tkeys,
                                                         const uint32_t* __restrict__ vals, const float4* __restrict__ rec1,
                                                         uint64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  keys[i] = ((uint64_t)tkeys[i] << 32) | (uint64_t)__float_as_uint(rec1[vals[i]].z);
}
hipError_t launch_export_keys(hipStream_t s, int64_t R, const Binning& b, const Geom& g, uint64_t* keys) {
  if (b.key_bytes == 2)
    hipLaunchKernelGGL(export_keys_kernel<uint16_t>, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, R,
                       (const uint16_t*)b.tkey[b.final_buf], b.vals[b.final_buf], g.rec1, keys);
  else
    hipLaunchKernelGGL(export_keys_kernel<uint32_t>, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, R,
                       (const uint32_t*)b.tkey[b.final_buf], b.vals[b.final_buf], g.rec1, keys);
  return hipGetLastError();
}

}  // namespace gsr
