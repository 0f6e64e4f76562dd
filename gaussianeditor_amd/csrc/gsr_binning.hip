// gsr_binning.hip -- K3 (emit), K4 (sort), K5 (tile ranges) and the blend work list.
//
// The reference builds one 64-bit key (tile | depth bits) per (Gaussian, tile) instance and runs a
// stable radix sort over all R instances on 32 + getHigherMsb(T) key bits (rasterizer_impl.cu:67-100,
// 253-261: 45 bits at 1080p).  Every instance of a Gaussian carries the SAME depth, so the identical
// order is obtained far cheaper by ordering the Gaussians first and the instances by tile afterwards
// (a stable sort by the minor key followed by a stable sort by the major key):
//
//   1. stable LSD radix sort of the P Gaussians by their 32 depth bits (value = index, culled ones
//      last).  P is ~5x smaller than R, and it runs inside gsr_preprocess, i.e. under the host's
//      blocking readback of num_rendered (launch_depth_passes);
//   2. GROUP INSTANCES: tiles are taken in groups of 8 x 8 (128 x 128 pixels).  One (group, Gaussian)
//      pair per group a Gaussian's tile rectangle reaches is emitted in depth order -- ~1.3 per visible
//      Gaussian where the reference emits ~5 (tile, Gaussian) pairs, ~3 instead of ~30 on deep-tile scenes --
//      and sorted stably by group id in ONE radix pass (at most GROUP_MAX = 2048 groups);
//   3. inside a group a Gaussian's tiles are a 64-bit mask.  One wave takes 64 consecutive group
//      instances, transposes the 64 x 64 bit matrix (lane j: mask of instance j -> lane t: which of the
//      64 instances touch tile t, in order) and lane t appends those Gaussian indices to ITS tile's list:
//      the per-tile order is the instance order = depth order, ties in ascending Gaussian index exactly
//      like the reference's.  Where a wave's run of a tile starts comes from a count pass with the same
//      transposes and a column scan over the chunks of a group; the per-tile ranges (K5) are the prefix
//      of the tile totals -- no (tile, index) pair is ever written, sorted or re-read: 4 bytes per
//      instance leave the chip once.
//
// Images with more than GROUP_MAX groups (> 131 072 tiles, 33 MPix) keep the round-2 path: (tile id,
// Gaussian) pairs emitted in depth order and radix-sorted on the tile id in ceil(bits / 8) passes.
//
// A radix pass is three kernels (histogram -> per-bin scan -> scatter); there is no decoupled
// look-back, so no inter-workgroup hand-off inside a launch (per-XCD L2s are not coherent; a kernel
// boundary is the cheapest correct fence).  Stability comes from ranking with wave64 ballots in key
// order, never from atomics.
#include <stdlib.h>

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

// `publish_dst` (first depth pass only): one EXTRA block reduces the per-block partials K1 left (num_rendered, range of the
// depth keys) and writes the four words into the caller's pinned, device-mapped host buffer -- the readback of
// gsr_preprocess without a copy command of its own (a 4 us blit kernel plus a 6 us bubble behind it before) and without a
// header to clear in front of K1.
// TH threads per block (256, or 1024 for sorts of few blocks: see radix_sort_pairs), SORT_KPB / TH keys per thread.
// WEIGHTED (the LAST depth pass of the grouped path): next to the digit counts the block leaves, in rows RBINS .. 2 RBINS - 1
// of `hist`, the sum of the group counts of its keys per digit (rect32: the packed tile rectangles in the order of `keys`):
// scanned by the same scan launch, they give the last scatter kernel every Gaussian's position in the emission order.
template <class K, int TH, bool WEIGHTED = false>
__global__ void __launch_bounds__(TH) sort_hist_kernel(const K* __restrict__ keys, int64_t n, int shift,
                                                                uint32_t mask, uint32_t* __restrict__ hist,
                                                                uint32_t nblocks, const uint4* __restrict__ publish_src,
                                                                uint32_t publish_count, uint32_t* __restrict__ publish_dst,
                                                                uint32_t publish_seq, const uint32_t* __restrict__ rect32) {
  __shared__ uint32_t h[RBINS];
  __shared__ uint32_t hw[WEIGHTED ? RBINS : 1];
  if (publish_dst != nullptr && blockIdx.x == nblocks) {  // the EXTRA block of a publishing launch: it does nothing else
    __shared__ unsigned long long psum[TH / 64];
    __shared__ uint32_t pmax[TH / 64], pinv[TH / 64];
    __shared__ unsigned long long pgrp[TH / 64];
    unsigned long long sum = 0, gsum = 0;
    uint32_t kmax = 0, kinv = 0;
    // PUB_UNROLL partials per thread and round, their loads issued together (index clamped): at 6 M Gaussians a 256-thread
    // block walks 23 438 partials, and one dependent load per round made this block 40 us -- the whole launch waited for it
    constexpr int PUB_UNROLL = 8;
    for (uint32_t base = threadIdx.x; base < publish_count; base += TH * PUB_UNROLL) {
      uint4 v[PUB_UNROLL];
#pragma unroll
      for (int u = 0; u < PUB_UNROLL; ++u) v[u] = publish_src[min(base + (uint32_t)u * TH, publish_count - 1u)];
#pragma unroll
      for (int u = 0; u < PUB_UNROLL; ++u) {
        if (base + (uint32_t)u * TH < publish_count) {
          sum += v[u].x;
          gsum += v[u].w;
          kmax = max(kmax, v[u].y);
          kinv = max(kinv, v[u].z);
        }
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      sum += __shfl_xor(sum, d, 64);
      gsum += __shfl_xor(gsum, d, 64);
      kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d, 64));
      kinv = max(kinv, (uint32_t)__shfl_xor((int)kinv, d, 64));
    }
    if (lane_id() == 0) {
      psum[threadIdx.x >> 6] = sum;
      pgrp[threadIdx.x >> 6] = gsum;
      pmax[threadIdx.x >> 6] = kmax;
      pinv[threadIdx.x >> 6] = kinv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < TH / 64; ++w) {
        sum += psum[w];
        gsum += pgrp[w];
        kmax = max(kmax, pmax[w]);
        kinv = max(kinv, pinv[w]);
      }
      publish_dst[0] = (uint32_t)sum;
      publish_dst[1] = (uint32_t)(sum >> 32);
      publish_dst[GEOM_HDR_KEYMAX] = kmax;
      publish_dst[GEOM_HDR_KEYINVMAX] = kinv;
      publish_dst[GEOM_HDR_GROUPS] = (uint32_t)gsum;
      publish_dst[GEOM_HDR_GROUPS + 1] = (uint32_t)(gsum >> 32);
      __threadfence_system();
      // the host spins on this word (fine-grained pinned memory): no event, hence no barrier packet in the stream
      __hip_atomic_store(publish_dst + GEOM_HDR_FINAL, publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  if (threadIdx.x < RBINS) {
    h[threadIdx.x] = 0;
    if (WEIGHTED) hw[threadIdx.x] = 0;
  }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_KPB;
  // all loads of a thread are issued back to back (unconditional, index clamped): the kernel is bound by memory latency,
  // not by bandwidth or the LDS atomics
  uint32_t kv[(SORT_KPB / TH)], rv[WEIGHTED ? (SORT_KPB / TH) : 1];
#pragma unroll
  for (int i = 0; i < (SORT_KPB / TH); ++i) {
    const int64_t k = base + (int64_t)i * TH + threadIdx.x;
    kv[i] = (uint32_t)keys[k < n ? k : n - 1];
    if (WEIGHTED) rv[i] = rect32[k < n ? k : n - 1];
  }
#pragma unroll
  for (int i = 0; i < (SORT_KPB / TH); ++i) {
    const int64_t k = base + (int64_t)i * TH + threadIdx.x;
    if (k < n) {
      const uint32_t d = (kv[i] >> shift) & mask;
      atomicAdd(&h[d], 1u);
      if (WEIGHTED) {
        const uint32_t g = rect32_groups(rv[i]);
        if (g != 0u) atomicAdd(&hw[d], g);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < RBINS) {
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
    if (WEIGHTED) hist[(size_t)(RBINS + threadIdx.x) * nblocks + blockIdx.x] = hw[threadIdx.x];
  }
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
template <bool IOTA, class K, int TH>
__global__ void __launch_bounds__(TH) sort_scatter_kernel(const K* __restrict__ keys_in,
                                                                   const uint32_t* __restrict__ vals_in,
                                                                   K* __restrict__ keys_out,
                                                                   uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                                   uint32_t mask, const uint32_t* __restrict__ hist,
                                                                   const uint32_t* __restrict__ bin_total,
                                                                   uint32_t nblocks) {
  constexpr int NW = TH / 64, ITEMS = SORT_KPB / TH;  // waves per block, keys per thread (a wave owns 64 * ITEMS consecutive keys)
  __shared__ uint32_t cnt[NW][RBINS];   // per-wave digit counts -> per-wave local bases
  __shared__ uint32_t gbase[RBINS];     // global position of the block's first key of each digit
  __shared__ uint32_t lexcl[RBINS];     // position of each digit's run inside the block-sorted order
  __shared__ uint32_t smem[TH / 64 + 1];
  __shared__ uint32_t skey[SORT_KPB];
  __shared__ uint32_t sval[SORT_KPB];
  const int w = (int)(threadIdx.x >> 6), l = lane_id();
  const int64_t bbase = (int64_t)blockIdx.x * SORT_KPB;
  const int64_t wbase = bbase + (int64_t)w * (ITEMS * 64);
  uint32_t key[ITEMS], val[ITEMS];
  uint16_t rank[ITEMS];
  const uint64_t lt_mask = (1ull << l) - 1ull;
  // the block's pairs are requested first, so that they travel while the digit bases below are loaded and scanned
  // (no load crosses a barrier on its own)
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int64_t k = wbase + (int64_t)i * 64 + l;
    const int64_t kc = k < n ? k : n - 1;  // unconditional loads (clamped), validity handled below
    key[i] = (uint32_t)keys_in[kc];
    val[i] = IOTA ? (uint32_t)kc : vals_in[kc];
  }
  const bool owns_bin = threadIdx.x < RBINS;  // thread d < 256 owns digit d
  const uint32_t my_hist = owns_bin ? hist[(size_t)threadIdx.x * nblocks + blockIdx.x] : 0u;
  for (int i = threadIdx.x; i < NW * RBINS; i += TH) (&cnt[0][0])[i] = 0;
  {
    uint32_t tot;
    const uint32_t run = block_excl_scan_u32<TH>(owns_bin ? bin_total[threadIdx.x] : 0u, &tot, smem);
    if (owns_bin) gbase[threadIdx.x] = run + my_hist;
  }
  __syncthreads();

#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
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
    if (owns_bin) {
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const uint32_t c = cnt[i][threadIdx.x];
        cnt[i][threadIdx.x] = run;
        run += c;
      }
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan_u32<TH>(run, &tot, smem);
    if (owns_bin) lexcl[threadIdx.x] = ex;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
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
  for (int i = 0; i < ITEMS; ++i) {
    const int j = i * TH + (int)threadIdx.x;
    if (j < nvalid) {
      const uint32_t kk = skey[j];
      const uint32_t d = (kk >> shift) & mask;
      const uint32_t pos = gbase[d] + ((uint32_t)j - lexcl[d]);
      keys_out[pos] = (K)kk;
      vals_out[pos] = sval[j];
    }
  }
}

// A sort of few blocks (the depth order of ~10^6 Gaussians is 245 blocks on 256 CUs) is bound by the latency of ONE block:
// with 1024 threads a block's 4096 keys are ranked by 16 waves in 4 rounds instead of by 4 waves in 16.  Many-block sorts
// keep 256 threads (more blocks resident per CU).
constexpr uint32_t SORT_WIDE_MAX_BLOCKS = 1024;
template <class K, int TH>
static void radix_sort_pairs_th(hipStream_t s, K* const keys[2], uint32_t* const vals[2], int64_t n, int npass,
                                const int* digit_bits, uint32_t* hist, uint32_t* bin_total, bool iota_first, int p0,
                                const uint4* publish_src, uint32_t publish_count, uint32_t* publish_dst, uint32_t publish_seq) {
  const uint32_t nblocks = (uint32_t)((n + SORT_KPB - 1) / SORT_KPB);
  int cur = p0 & 1, shift = 0;
  for (int p = 0; p < p0; ++p) shift += digit_bits[p];
  for (int p = p0; p < npass; ++p) {
    const uint32_t mask = (1u << digit_bits[p]) - 1u;
    const bool pub = p == p0 && publish_dst != nullptr;
    // (a publishing launch has one block more: the reduction of K1's partials + the stores into the host's slot used to sit
    // in front of block 0's histogram and made the first pass 2.5 us longer than the others)
    hipLaunchKernelGGL((sort_hist_kernel<K, TH>), dim3(nblocks + (pub ? 1u : 0u)), dim3(TH), 0, s, (const K*)keys[cur], n, shift, mask,
                       hist, nblocks, pub ? publish_src : nullptr, publish_count, pub ? publish_dst : nullptr, publish_seq,
                       (const uint32_t*)nullptr);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(RBINS), dim3(SORT_THREADS), 0, s, hist, bin_total, nblocks);
    if (p == 0 && iota_first)
      hipLaunchKernelGGL((sort_scatter_kernel<true, K, TH>), dim3(nblocks), dim3(TH), 0, s, (const K*)keys[cur],
                         (const uint32_t*)vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, mask, hist, bin_total, nblocks);
    else
      hipLaunchKernelGGL((sort_scatter_kernel<false, K, TH>), dim3(nblocks), dim3(TH), 0, s, (const K*)keys[cur],
                         (const uint32_t*)vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, mask, hist, bin_total, nblocks);
    shift += digit_bits[p];
    cur ^= 1;
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
  if ((uint32_t)((n + SORT_KPB - 1) / SORT_KPB) <= SORT_WIDE_MAX_BLOCKS)
    radix_sort_pairs_th<K, 1024>(s, keys, vals, n, npass, digit_bits, hist, bin_total, iota_first, p0, publish_src, publish_count,
                                 publish_dst, publish_seq);
  else
    radix_sort_pairs_th<K, SORT_THREADS>(s, keys, vals, n, npass, digit_bits, hist, bin_total, iota_first, p0, publish_src,
                                         publish_count, publish_dst, publish_seq);
}

// ----------------------------------------------------------------------------------
// Grouped path (round 4): the depth sort carries the tile rectangle as a second payload, and its LAST pass leaves
// everything the emission needs in depth order -- nothing is gathered or scanned afterwards (rounds 1-3: a gather kernel
// of the 8-byte rectangles, 14 us at 1 M Gaussians and 116 us at 6 M, + a block-sum scan).
//   LAST = false: sort_scatter_kernel + the payload (IOTA: packed from Geom::rect, read in index order);
//   LAST = true:  additionally every Gaussian's position in the emission order, i.e. the exclusive prefix of the group
//                 counts in depth order = (groups of all smaller digits) + (groups of this digit in earlier blocks: both
//                 from the WEIGHTED histogram rows, scanned) + (groups of this digit ahead of it inside the block: a block
//                 scan over the LDS-sorted slots); it goes where the sorted keys would (nobody reads them), and the block
//                 stamps the header with the side that holds the order.
// ----------------------------------------------------------------------------------
template <bool LAST, bool IOTA, int TH>
__global__ void __launch_bounds__(TH) depth_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                          const uint32_t* __restrict__ vals_in,
                                                          const uint32_t* __restrict__ rect_in,
                                                          const uint2* __restrict__ rect8,
                                                          uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                          uint32_t* __restrict__ rect_out, int64_t n, int shift, uint32_t mask,
                                                          const uint32_t* __restrict__ hist,
                                                          const uint32_t* __restrict__ bin_total, uint32_t nblocks,
                                                          uint32_t* __restrict__ hdr, uint32_t final_buf) {
  constexpr int NW = TH / 64, ITEMS = SORT_KPB / TH;
  __shared__ uint32_t cnt[NW][RBINS];
  __shared__ uint32_t gbase[RBINS], lexcl[RBINS], wgbase[LAST ? RBINS : 1];
  __shared__ uint32_t smem[TH / 64 + 1];
  __shared__ uint32_t skey[LAST ? 4 : SORT_KPB];  // LAST: the sorted keys are not written, only their digit is needed (sdig)
  __shared__ uint8_t sdig[LAST ? SORT_KPB : 4];   // (12 KB less: three blocks of the 256-thread form per CU instead of two)
  __shared__ uint32_t sval[SORT_KPB];
  __shared__ __align__(16) uint32_t srect[SORT_KPB];
  __shared__ __align__(16) uint32_t spre[LAST ? SORT_KPB : 4];  // exclusive prefix of the group counts over the sorted slots
  const int w = (int)(threadIdx.x >> 6), l = lane_id();
  const int64_t bbase = (int64_t)blockIdx.x * SORT_KPB;
  const int64_t wbase = bbase + (int64_t)w * (ITEMS * 64);
  uint32_t key[ITEMS], val[ITEMS], rc[ITEMS];
  uint16_t rank[ITEMS];
  const uint64_t lt_mask = (1ull << l) - 1ull;
  if (LAST && blockIdx.x == 0 && threadIdx.x == 0) hdr[GEOM_HDR_FINAL] = final_buf;  // for the emit kernel (gsr_bin)
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int64_t k = wbase + (int64_t)i * 64 + l;
    const int64_t kc = k < n ? k : n - 1;  // unconditional loads (clamped), validity handled below
    key[i] = keys_in[kc];
    if (!IOTA) val[i] = vals_in[kc];
    if (IOTA) {
      // ONE 8-byte load per rectangle, consumed after all loads of the thread have been issued (a first version read
      // .y, waited, and read .x only for a non-empty rectangle: ITEMS dependent round trips, 87 us at 6 M Gaussians)
      const unsigned long long r8 = reinterpret_cast<const unsigned long long*>(rect8)[kc];
      rc[i] = (uint32_t)r8;
      val[i] = (uint32_t)(r8 >> 32);  // (parked: val = kc is recomputed below)
    } else {
      rc[i] = rect_in[kc];
    }
  }
  if (IOTA) {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int64_t k = wbase + (int64_t)i * 64 + l;
      rc[i] = pack_rect32(rc[i], val[i]);
      val[i] = (uint32_t)(k < n ? k : n - 1);
    }
  }
  const bool owns_bin = threadIdx.x < RBINS;  // thread d < 256 owns digit d
  const uint32_t my_hist = owns_bin ? hist[(size_t)threadIdx.x * nblocks + blockIdx.x] : 0u;
  const uint32_t my_whist = (LAST && owns_bin) ? hist[(size_t)(RBINS + threadIdx.x) * nblocks + blockIdx.x] : 0u;
  const uint32_t my_total = owns_bin ? bin_total[threadIdx.x] : 0u;
  const uint32_t my_wtotal = (LAST && owns_bin) ? bin_total[RBINS + threadIdx.x] : 0u;
  for (int i = threadIdx.x; i < NW * RBINS; i += TH) (&cnt[0][0])[i] = 0;
  {
    uint32_t tot;
    const uint32_t run = block_excl_scan_u32<TH>(my_total, &tot, smem);
    if (owns_bin) gbase[threadIdx.x] = run + my_hist;
    if (LAST) {
      const uint32_t wrun = block_excl_scan_u32<TH>(my_wtotal, &tot, smem);
      if (owns_bin) wgbase[threadIdx.x] = wrun + my_whist;
    }
  }
  __syncthreads();

#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
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
    uint32_t run = 0;
    if (owns_bin) {
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const uint32_t c = cnt[i][threadIdx.x];
        cnt[i][threadIdx.x] = run;
        run += c;
      }
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan_u32<TH>(run, &tot, smem);
    if (owns_bin) lexcl[threadIdx.x] = ex;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int64_t k = wbase + (int64_t)i * 64 + l;
    if (k < n) {
      const uint32_t d = (key[i] >> shift) & mask;
      const uint32_t lp = lexcl[d] + cnt[w][d] + rank[i];
      if (LAST) sdig[lp] = (uint8_t)d;
      else skey[lp] = key[i];
      sval[lp] = val[i];
      srect[lp] = rc[i];
    }
  }
  __syncthreads();
  const int nvalid = (int)min((int64_t)SORT_KPB, n - bbase);
  if (LAST) {
    // exclusive prefix of the group counts over the block-sorted slots: thread t takes slots [t ITEMS, (t + 1) ITEMS)
    uint32_t g[ITEMS], sum = 0;
    const int j0 = (int)threadIdx.x * ITEMS;
#pragma unroll
    for (int q = 0; q < ITEMS / 4; ++q) {
      const uint4 r = reinterpret_cast<const uint4*>(srect + j0)[q];
      g[4 * q] = r.x; g[4 * q + 1] = r.y; g[4 * q + 2] = r.z; g[4 * q + 3] = r.w;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      g[i] = j0 + i < nvalid ? rect32_groups(g[i]) : 0u;  // (slots behind the last key hold garbage)
      sum += g[i];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan_u32<TH>(sum, &tot, smem);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      spre[j0 + i] = run;
      run += g[i];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int j = i * TH + (int)threadIdx.x;
    if (j < nvalid) {
      const uint32_t kk = LAST ? 0u : skey[j];
      const uint32_t d = LAST ? (uint32_t)sdig[j] : (kk >> shift) & mask;
      const uint32_t first = lexcl[d];  // the digit's first slot in the block-sorted order (a digit that occurs: first <= j)
      const uint32_t pos = gbase[d] + ((uint32_t)j - first);
      keys_out[pos] = LAST ? wgbase[d] + (spre[j] - spre[first]) : kk;
      vals_out[pos] = sval[j];
      rect_out[pos] = srect[j];
    }
  }
}

// The depth sort of the grouped path: passes [p0, p1) (pass p reads side p & 1); `last`: pass p1 - 1 is the final one.
template <int TH>
static void depth_sort_grouped_th(hipStream_t s, int P, const Geom& g, int p0, int p1, bool last, uint32_t* publish_dst,
                                  uint32_t publish_seq) {
  const int64_t n = P;
  const uint32_t nblocks = (uint32_t)((n + SORT_KPB - 1) / SORT_KPB);
  const uint32_t npart = (uint32_t)((P + GAUSS_BLOCK - 1) / GAUSS_BLOCK);
  uint32_t* const hdr = reinterpret_cast<uint32_t*>(g.total);
  for (int p = p0; p < p1; ++p) {
    const int cur = p & 1, shift = 8 * p;
    const bool pub = p == p0 && publish_dst != nullptr, fin = last && p == p1 - 1;
    // (the histogram kernel always runs 1024 threads wide: it holds 1 KB of LDS, and its extra publishing block walks
    //  ceil(P / 256) partials -- 23 438 at 6 M Gaussians, 40 us with 256 threads and one load in flight per thread)
    const dim3 gh(nblocks + (pub ? 1u : 0u)), gs(nblocks), bt(TH), bh(1024);
    if (fin)
      hipLaunchKernelGGL((sort_hist_kernel<uint32_t, 1024, true>), gh, bh, 0, s, (const uint32_t*)g.dkey[cur], n, shift, 255u, g.ghist,
                         nblocks, pub ? g.k1_partials : nullptr, npart, pub ? publish_dst : nullptr, publish_seq,
                         (const uint32_t*)g.drect[cur]);
    else
      hipLaunchKernelGGL((sort_hist_kernel<uint32_t, 1024, false>), gh, bh, 0, s, (const uint32_t*)g.dkey[cur], n, shift, 255u, g.ghist,
                         nblocks, pub ? g.k1_partials : nullptr, npart, pub ? publish_dst : nullptr, publish_seq,
                         (const uint32_t*)nullptr);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(fin ? 2 * RBINS : RBINS), dim3(SORT_THREADS), 0, s, g.ghist, g.gbin_total, nblocks);
#define GSR_DEPTH_SCATTER(LASTP, IOTAP)                                                                                          \
  hipLaunchKernelGGL((depth_scatter_kernel<LASTP, IOTAP, TH>), gs, bt, 0, s, (const uint32_t*)g.dkey[cur],                       \
                     (const uint32_t*)g.dval[cur], (const uint32_t*)g.drect[cur], (const uint2*)g.rect, g.dkey[cur ^ 1],         \
                     g.dval[cur ^ 1], g.drect[cur ^ 1], n, shift, 255u, (const uint32_t*)g.ghist, (const uint32_t*)g.gbin_total, \
                     nblocks, hdr, (uint32_t)(cur ^ 1))
    if (fin && p == 0) GSR_DEPTH_SCATTER(true, true);
    else if (fin) GSR_DEPTH_SCATTER(true, false);
    else if (p == 0) GSR_DEPTH_SCATTER(false, true);
    else GSR_DEPTH_SCATTER(false, false);
#undef GSR_DEPTH_SCATTER
  }
}
hipError_t launch_depth_passes_grouped(hipStream_t s, int P, const Geom& g, int p0, int p1, bool last, uint32_t* publish_dst,
                                       uint32_t publish_seq) {
  if ((uint32_t)(((int64_t)P + SORT_KPB - 1) / SORT_KPB) <= SORT_WIDE_MAX_BLOCKS)
    depth_sort_grouped_th<1024>(s, P, g, p0, p1, last, publish_dst, publish_seq);
  else
    depth_sort_grouped_th<SORT_THREADS>(s, P, g, p0, p1, last, publish_dst, publish_seq);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// Depth order of the Gaussians + prefix of tiles_touched in that order.
// ----------------------------------------------------------------------------------
// `sgx` != 0 (grouped path): block sums of the number of 8x8-tile GROUPS a rectangle reaches; sgx == 0 (legacy pair sort):
// of its tiles.
__global__ void __launch_bounds__(GAUSS_BLOCK) sorted_block_sums_kernel(int P, int gx, int sgx, const uint32_t* __restrict__ order,
                                                                       const uint2* __restrict__ rect,
                                                                       uint32_t* __restrict__ wh_sorted,
                                                                       uint32_t* __restrict__ org_sorted,
                                                                       uint32_t* __restrict__ block_sums,
                                                                       uint32_t* __restrict__ hdr, uint32_t final_buf) {
  __shared__ uint32_t smem[GAUSS_BLOCK / 64 + 1];
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr[GEOM_HDR_FINAL] = final_buf;  // for the emit kernels (gsr_bin)
  const int i = (int)(blockIdx.x * GAUSS_BLOCK + threadIdx.x);
  // THE gather of the binning's first half: the tile rectangle of the i-th Gaussian in depth order (8 bytes).
  // What the emit kernel needs of it is left in depth order, so that it reads everything coalesced.
  const uint2 rc = i < P ? rect[order[i]] : make_uint2(0u, 0u);
  const uint32_t w = rc.y & 0xffffu, h = rc.y >> 16, x0 = rc.x & 0xffffu, y0 = rc.x >> 16;
  uint32_t n = w * h, org = y0 * (uint32_t)gx + x0;
  if (sgx != 0 && n != 0) {  // grouped path: the emit kernel gets the tile rectangle as it is and counts GROUPS
    n = (((x0 + w - 1u) >> GROUP_SHIFT) - (x0 >> GROUP_SHIFT) + 1u) * (((y0 + h - 1u) >> GROUP_SHIFT) - (y0 >> GROUP_SHIFT) + 1u);
    org = rc.x;
  }
  if (i < P) {
    wh_sorted[i] = rc.y;  // width | height << 16 of the tile rectangle
    org_sorted[i] = org;  // legacy path: id of its first tile; grouped path: x | y << 16 of its first tile
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
hipError_t launch_depth_finish(hipStream_t s, int P, const Geom& g, int passes, int gx, int sgx) {
  const int fin = passes & 1;
  const int nb = (P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  // dkey[fin ^ 1] / dval[fin ^ 1] (the input of the last pass) are dead: reuse them for the depth-ordered rectangles
  hipLaunchKernelGGL(sorted_block_sums_kernel, dim3(nb), dim3(GAUSS_BLOCK), 0, s, P, gx, sgx, g.dval[fin], g.rect,
                     g.dkey[fin ^ 1], g.dval[fin ^ 1], g.block_sums, reinterpret_cast<uint32_t*>(g.total), (uint32_t)fin);
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
__global__ void __launch_bounds__(GAUSS_BLOCK) emit_keys_kernel(int P, int gx, int nclear, const Geom g, K* __restrict__ tkeys,
                                                               uint32_t* __restrict__ vals, uint2* __restrict__ ranges) {
  __shared__ uint32_t smem[GAUSS_BLOCK / 64 + 1];
  __shared__ uint32_t s_off[GAUSS_BLOCK + 1], s_idx[GAUSS_BLOCK], s_org[GAUSS_BLOCK], s_w[GAUSS_BLOCK];
  const int i = (int)(blockIdx.x * GAUSS_BLOCK + threadIdx.x);
  // legacy path: the ranges of tiles no instance falls into stay (0,0) (reference: cudaMemset, rasterizer_impl.cu:263-265);
  // cleared here, two launches ahead of tile_ranges_kernel, instead of by a memset of its own
  if (ranges != nullptr && i < nclear) ranges[i] = make_uint2(0u, 0u);
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

// Grouped path: one GROUP INSTANCE per (Gaussian, 8x8-tile group) pair, in depth order, slot-parallel exactly like
// emit_keys_kernel.  An instance is (key, Gaussian index) with key = group id | local rectangle << 16: the part of the
// Gaussian's tile rectangle inside the group as x0 | (x1 - 1) << 3 | y0 << 6 | (y1 - 1) << 9 (tile coordinates relative
// to the group, 0..7) -- all the chunk kernels need to rebuild the 64-bit tile mask, so they never gather anything.
__device__ __forceinline__ uint32_t group_local_rect(uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t gxi, uint32_t gyi) {
  const uint32_t bx = gxi << GROUP_SHIFT, by = gyi << GROUP_SHIFT;
  const uint32_t lx0 = max(x0, bx) - bx, lx1 = min(x0 + w, bx + GROUP_EDGE) - bx;  // [lx0, lx1) within 0..8, non-empty
  const uint32_t ly0 = max(y0, by) - by, ly1 = min(y0 + h, by + GROUP_EDGE) - by;
  return lx0 | ((lx1 - 1u) << 3) | (ly0 << 6) | ((ly1 - 1u) << 9);
}
// Blocks behind the last Gaussian block (`nbg` on) write the padding value into every KEY slot of the radix pass's output side
// (the scatter overwrites the real ones: what remains are the padding slots between the segments and behind the last one; a real
// key never equals GROUP_PAD).  It is independent of everything this kernel does and used to sit in front of the histogram
// kernel's own work (+1.5 us there).
__global__ void __launch_bounds__(GAUSS_BLOCK) emit_groups_kernel(int P, int sgx, const Geom g, uint32_t* __restrict__ gkeys,
                                                                 uint32_t* __restrict__ vals, int nbg,
                                                                 uint32_t* __restrict__ fill_dst, int64_t fill_n,
                                                                 uint4* __restrict__ zero_dst, int64_t zero_n4) {
  if ((int)blockIdx.x >= nbg) {
    const int64_t nb = (int64_t)gridDim.x - nbg, b = (int64_t)blockIdx.x - nbg;
    const int64_t n4 = fill_n / 4;  // (the key arrays are 256-byte aligned)
    uint4* const q = reinterpret_cast<uint4*>(fill_dst);
    // (the forward blend's per-checkpoint work counts and the tiles' walk depths start every view at zero)
    for (int64_t k = b * GAUSS_BLOCK + threadIdx.x; k < zero_n4; k += nb * GAUSS_BLOCK) zero_dst[k] = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t k = b * GAUSS_BLOCK + threadIdx.x; k < n4; k += nb * GAUSS_BLOCK) q[k] = make_uint4(GROUP_PAD, GROUP_PAD, GROUP_PAD, GROUP_PAD);
    if (b == 0)
      for (int64_t k = 4 * n4 + threadIdx.x; k < fill_n; k += GAUSS_BLOCK) fill_dst[k] = GROUP_PAD;
    return;
  }
  __shared__ uint32_t s_off[GAUSS_BLOCK + 1], s_idx[GAUSS_BLOCK], s_rect[GAUSS_BLOCK];
  __shared__ uint32_t s_first, s_end;
  const int i = (int)(blockIdx.x * GAUSS_BLOCK + threadIdx.x);
  const uint32_t fin = reinterpret_cast<const uint32_t*>(g.total)[GEOM_HDR_FINAL];  // side holding the depth order
  // Everything in depth order, left by the last pass of the depth sort (depth_scatter_kernel<true>): the Gaussian, its
  // packed tile rectangle and its first slot in the emission order -- three coalesced loads, no gather, no scan.
  const uint32_t idx = i < P ? g.dval[fin][i] : 0u;
  const uint32_t r32 = i < P ? g.drect[fin][i] : RECT32_NONE;
  const uint32_t pre = i < P ? g.dkey[fin][i] : 0u;
  const uint32_t n = rect32_groups(r32);
  const int last = min(P - 1 - (int)blockIdx.x * GAUSS_BLOCK, GAUSS_BLOCK - 1);
  if (threadIdx.x == 0) s_first = pre;
  if ((int)threadIdx.x == last) s_end = pre + n;
  __syncthreads();
  const uint32_t boff = s_first, total = s_end - boff;
  s_off[threadIdx.x] = i < P ? pre - boff : total;
  s_idx[threadIdx.x] = idx;
  s_rect[threadIdx.x] = r32;
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
    const uint32_t k = s - s_off[lo], jr = s_rect[lo];
    const uint32_t jx = jr & 0xffu, jy = (jr >> 8) & 0xffu, jw = ((jr >> 16) & 0xffu) + 1u, jh = (jr >> 24) + 1u;
    const uint32_t g0x = jx >> GROUP_SHIFT, nsx = ((jx + jw - 1u) >> GROUP_SHIFT) - g0x + 1u;
    // row = k / nsx without an integer division (k, nsx < 2^24), one correction step each way
    uint32_t row = (uint32_t)((float)k * __builtin_amdgcn_rcpf((float)nsx));
    if (row * nsx > k) row--;
    if ((row + 1u) * nsx <= k) row++;
    const uint32_t gxi = g0x + (k - row * nsx), gyi = (jy >> GROUP_SHIFT) + row;
    gkeys[boff + s] = (gyi * (uint32_t)sgx + gxi) | (group_local_rect(jx, jy, jw, jh, gxi, gyi) << 16);
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
// Grouped path (`tile_total` != null): K5 happens here first -- the exclusive prefix of the tiles' instance counts in tile-id
// order gives every tile's [begin, end) (identifyTileRanges, rasterizer_impl.cu:105-125; (0, 0) for a tile without
// instances, as the reference's memset leaves it, :263-265) and `tile_start`, where the chunk waves start their runs.
__global__ void __launch_bounds__(1024) tile_worklist_kernel(int T, uint2* __restrict__ ranges,
                                                            uint32_t* __restrict__ order, uint32_t* __restrict__ meta,
                                                            uint32_t* __restrict__ queues, uint32_t* __restrict__ est,
                                                            const uint32_t* __restrict__ tile_total,
                                                            uint32_t* __restrict__ tile_start, uint32_t* __restrict__ ck_table,
                                                            uint32_t n_ck_tiles) {
  // Counting sort of the tiles by bucket.  Most tiles of an image fall into a handful of buckets, and LDS atomics on one
  // address serialise, so every bucket has WORK_SUB counters (chosen by the thread's lane): the order inside a bucket is
  // free anyway, and the sort's time stops growing with the number of tiles per bucket.
  // The kernel is one block on the critical path between the binning and the forward blend, i.e. a chain of dependent
  // memory round trips: the list lengths are fetched ONCE and kept in LDS (images of up to WORK_LDS_TILES tiles), and
  // everything nothing waits for (clearing the queue cursors and the forward's work counters) is stored last.
  constexpr int WORK_SUB = 16, NCNT = (WORK_BUCKETS + 1) * WORK_SUB, WORK_LDS_TILES = 8192;
  __shared__ uint32_t cnt[NCNT];
  __shared__ uint32_t smem[1024 / 64 + 1];
  __shared__ uint32_t s_len[WORK_LDS_TILES];
  const bool cached = T <= WORK_LDS_TILES;
  // Two independent halves: the prefix of the tile totals (ranges, tile_start: what the list-append kernel waits for) and the
  // counting sort of the tiles by list length (the work list: what only the blend kernels read).  Launched as TWO blocks they
  // run side by side, each through its own chain of dependent round trips (10.7 -> ~7 us on the path); as one block (legacy
  // path: the ranges exist already) one after the other.
  const bool split = gridDim.x == 2;
  const bool do_prefix = tile_total != nullptr && (!split || blockIdx.x == 0);
  const bool do_sort = !split || blockIdx.x == 1;
  for (int i = threadIdx.x; i < NCNT; i += 1024) cnt[i] = 0;
  if (do_prefix) {
    constexpr int PER = 8;  // consecutive tiles per thread and round: two 16-byte loads, six 16-byte stores
    uint32_t carry = 0;
    for (int base = 0; base < T; base += 1024 * PER) {
      const int i0 = base + (int)threadIdx.x * PER;
      const bool full = i0 + PER <= T;  // (tile_total / tile_start / ranges are 256-byte aligned, i0 a multiple of 8)
      uint32_t v[PER], sum = 0;
      if (full) {
        const uint4 a0 = reinterpret_cast<const uint4*>(tile_total + i0)[0], a1 = reinterpret_cast<const uint4*>(tile_total + i0)[1];
        v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
      } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) v[j] = i0 + j < T ? tile_total[i0 + j] : 0u;
      }
#pragma unroll
      for (int j = 0; j < PER; ++j) sum += v[j];
      uint32_t chunk;
      uint32_t run = carry + block_excl_scan_u32<1024>(sum, &chunk, smem);
      uint32_t st[PER];
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        st[j] = run;
        run += v[j];
      }
      if (full) {
        reinterpret_cast<uint4*>(tile_start + i0)[0] = make_uint4(st[0], st[1], st[2], st[3]);
        reinterpret_cast<uint4*>(tile_start + i0)[1] = make_uint4(st[4], st[5], st[6], st[7]);
#pragma unroll
        for (int j = 0; j < PER; j += 2)
          reinterpret_cast<uint4*>(ranges + i0)[j / 2] = make_uint4(v[j] ? st[j] : 0u, v[j] ? st[j] + v[j] : 0u,
                                                                   v[j + 1] ? st[j + 1] : 0u, v[j + 1] ? st[j + 1] + v[j + 1] : 0u);
      }
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        if (i0 + j < T) {
          if (!full) {
            tile_start[i0 + j] = st[j];
            ranges[i0 + j] = v[j] ? make_uint2(st[j], st[j] + v[j]) : make_uint2(0u, 0u);
          }
          if (cached) s_len[i0 + j] = v[j];
        }
      }
      carry += chunk;
    }
    if (threadIdx.x == 0) tile_start[T] = carry;
  } else if (cached && tile_total != nullptr) {  // (the sorting block of a split launch)
    for (int t = threadIdx.x; t < T; t += 1024) s_len[t] = tile_total[t];
  } else if (cached) {
    for (int t = threadIdx.x; t < T; t += 1024) {
      const uint2 r = ranges[t];
      s_len[t] = r.y - r.x;
    }
  }
  if (!do_sort) return;
  __syncthreads();  // (s_len; without the cache the block reads its own `ranges` stores below)
  auto len_of = [&](int t) -> uint32_t {
    if (cached) return s_len[t];
    if (tile_total != nullptr) return tile_total[t];
    const uint2 r = ranges[t];
    return r.y - r.x;
  };
  const uint32_t sub = threadIdx.x & (WORK_SUB - 1);
  auto bucket_of = [](uint32_t len) -> uint32_t {
    if (len == 0) return WORK_BUCKETS;  // empty tiles: last
    const uint32_t c = (len + 63u) / 64u;
    return (uint32_t)(WORK_BUCKETS - 1) - min(c - 1u, (uint32_t)(WORK_BUCKETS - 1));
  };
  for (int t = threadIdx.x; t < T; t += 1024) atomicAdd(&cnt[bucket_of(len_of(t)) * WORK_SUB + sub], 1u);
  __syncthreads();
  {  // exclusive scan over the NCNT counters in (bucket, sub) order: counts -> cursors (one block scan, 3 counters a thread)
    constexpr int CPT = (NCNT + 1023) / 1024;
    const int i0 = (int)threadIdx.x * CPT;
    uint32_t v[CPT], sum = 0;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      v[j] = i0 + j < NCNT ? cnt[i0 + j] : 0u;
      sum += v[j];
    }
    uint32_t all;
    uint32_t run = block_excl_scan_u32<1024>(sum, &all, smem);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      if (i0 + j < NCNT) cnt[i0 + j] = run;
      if (i0 + j == WORK_BUCKETS * WORK_SUB) meta[0] = run;  // number of non-empty tiles
      run += v[j];
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 1024) {
    const uint32_t len = len_of(t);
    const uint32_t pos = atomicAdd(&cnt[bucket_of(len) * WORK_SUB + sub], 1u);
    order[pos] = (uint32_t)t;
    // the tiles with the longest lists own checkpoint slots (Image::ck_table): the forward blend writes its state there
    // every few hundred list positions, the backward walks such a tile's list as independent segments
    ck_table[t] = (len != 0u && pos < n_ck_tiles) ? pos : CK_NONE;
  }
  // per-quadrant work counters and walk depths of the forward blend (Image::work_est, work_maxc: contiguous; quadrants
  // outside the image are never written, the items of a cut quadrant combine their values with atomicMax)
  for (int i = threadIdx.x; i < 8 * T; i += 1024) est[i] = 0u;
  // work-queue cursors and retire counters of the three blend kernels start at zero; each blend launch leaves its
  // own zeroed again (gsr_blend.hip: retire_queue), so this is the only place that clears them
  for (int i = threadIdx.x; i < QUEUE_KINDS * QUEUE_LINES; i += 1024) {
    queues[(size_t)i * QUEUE_STRIDE] = 0u;      // taken from the front / counter
    queues[(size_t)i * QUEUE_STRIDE + 1] = 0u;  // (second word of the line: spare)
  }
}

// ----------------------------------------------------------------------------------
// Grouped path, step 2: ONE stable radix pass over the group instances (key = group id, < 2^BITS), whose output leaves
// each group's segment padded to whole chunks: segment s starts at chunk * (number of chunks of the groups before it).
// ----------------------------------------------------------------------------------
// histogram of the pass + the padding value into every KEY slot of the output side (the scatter overwrites the real ones:
// what remains are the padding slots between the segments and behind the last one; a real key never equals GROUP_PAD,
// its local rectangle has 12 bits)
template <int BITS, int TH>
__global__ void __launch_bounds__(TH) group_hist_kernel(const uint32_t* __restrict__ keys, int64_t n,
                                                                 uint32_t* __restrict__ hist, uint32_t nblocks,
                                                                 uint32_t* __restrict__ fill_dst, int64_t fill_n) {
  constexpr int NB = 1 << BITS;
  __shared__ uint32_t h[NB];
  for (int i = threadIdx.x; i < NB; i += TH) h[i] = 0;
  (void)fill_dst;  // (the padding of the sorted side is written by extra blocks of emit_groups_kernel since round 3)
  (void)fill_n;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_KPB;
  uint32_t kv[(SORT_KPB / TH)];
#pragma unroll
  for (int i = 0; i < (SORT_KPB / TH); ++i) {
    const int64_t k = base + (int64_t)i * TH + threadIdx.x;
    kv[i] = (uint32_t)keys[k < n ? k : n - 1];
  }
#pragma unroll
  for (int i = 0; i < (SORT_KPB / TH); ++i) {
    const int64_t k = base + (int64_t)i * TH + threadIdx.x;
    if (k < n) atomicAdd(&h[kv[i] & (NB - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NB; i += TH) hist[(size_t)i * nblocks + blockIdx.x] = h[i];
}

// sort_scatter_kernel with 2^BITS bins and padded segment starts; ranks from BITS ballots per key (stable).
template <int BITS, int TH>
__global__ void __launch_bounds__(TH) group_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                    const uint32_t* __restrict__ vals_in,
                                                                    uint32_t* __restrict__ keys_out,
                                                                    uint32_t* __restrict__ vals_out, int64_t n,
                                                                    const uint32_t* __restrict__ hist,
                                                                    const uint32_t* __restrict__ bin_total,
                                                                    uint32_t nblocks, uint32_t chunk,
                                                                    uint32_t* __restrict__ group_first) {
  constexpr int NB = 1 << BITS, NW = TH / 64, ITEMS = SORT_KPB / TH;
  constexpr int BT = NB < TH ? NB : TH, BPT = NB / BT;  // the first BT threads own BPT consecutive bins each
  const bool owns = (int)threadIdx.x < BT;
  __shared__ uint16_t cnt[NW][NB];     // per-wave digit counts -> per-wave local bases
  __shared__ uint32_t gbase[NB];       // global position of the block's first key of each digit
  __shared__ uint16_t lexcl[NB];       // position of each digit's run inside the block-sorted order
  __shared__ uint32_t smem[TH / 64 + 1];
  __shared__ uint32_t skey[SORT_KPB];
  __shared__ uint32_t sval[SORT_KPB];
  const int w = (int)(threadIdx.x >> 6), l = lane_id();
  const int64_t bbase = (int64_t)blockIdx.x * SORT_KPB;
  const int64_t wbase = bbase + (int64_t)w * (ITEMS * 64);
  uint32_t key[ITEMS], val[ITEMS];  // key: group id (the digit) | local rectangle << 16 (payload)
  uint16_t rank[ITEMS];
  const uint64_t lt_mask = (1ull << l) - 1ull;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int64_t k = wbase + (int64_t)i * 64 + l;
    const int64_t kc = k < n ? k : n - 1;  // unconditional loads (clamped), validity handled below
    key[i] = keys_in[kc];
    val[i] = vals_in[kc];
  }
  {
    // padded start of every digit's segment: exclusive prefix of the bin totals rounded up to whole chunks
    uint32_t padded[BPT], mine[BPT], sum = 0;
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const uint32_t d = owns ? threadIdx.x * BPT + j : 0u;
      const uint32_t tot = owns ? bin_total[d] : 0u;
      padded[j] = (tot + chunk - 1u) / chunk * chunk;
      mine[j] = owns ? hist[(size_t)d * nblocks + blockIdx.x] : 0u;
      sum += padded[j];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan_u32<TH>(sum, &tot, smem);
    if (owns) {
#pragma unroll
      for (int j = 0; j < BPT; ++j) {
        gbase[threadIdx.x * BPT + j] = run + mine[j];
        // row of the group's first chunk in the per-chunk tables (+ one entry behind the last group), for group_colscan_kernel
        if (blockIdx.x == 0) group_first[threadIdx.x * BPT + j] = run / chunk;
        run += padded[j];
      }
      if (blockIdx.x == 0 && (int)threadIdx.x == BT - 1) group_first[NB] = run / chunk;
    }
  }
  for (int i = threadIdx.x; i < NW * NB; i += TH) (&cnt[0][0])[i] = 0;
  __syncthreads();

#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int64_t k = wbase + (int64_t)i * 64 + l;
    const bool valid = k < n;
    const uint32_t d = key[i] & (NB - 1);
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
    if (valid && before == 0) cnt[w][d] = (uint16_t)(old + (uint32_t)__popcll(m));
    rank[i] = (uint16_t)(old + before);
  }
  __syncthreads();
  {
    // per digit: block total, exclusive prefix over the waves, and the digit's offset in block-sorted order
    uint32_t tot_d[BPT], sum = 0;
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const uint32_t d = owns ? threadIdx.x * BPT + j : 0u;
      uint32_t run = 0;
      if (owns) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
          const uint32_t c = cnt[i][d];
          cnt[i][d] = (uint16_t)run;
          run += c;
        }
      }
      tot_d[j] = run;
      sum += run;
    }
    uint32_t tot;
    uint32_t run = block_excl_scan_u32<TH>(sum, &tot, smem);
    if (owns) {
#pragma unroll
      for (int j = 0; j < BPT; ++j) {
        lexcl[threadIdx.x * BPT + j] = (uint16_t)run;
        run += tot_d[j];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int64_t k = wbase + (int64_t)i * 64 + l;
    if (k < n) {
      const uint32_t d = key[i] & (NB - 1);
      const uint32_t lp = (uint32_t)lexcl[d] + (uint32_t)cnt[w][d] + rank[i];
      skey[lp] = key[i];
      sval[lp] = val[i];
    }
  }
  __syncthreads();
  const int nvalid = (int)min((int64_t)SORT_KPB, n - bbase);
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int j = i * TH + (int)threadIdx.x;
    if (j < nvalid) {
      const uint32_t kk = skey[j], d = kk & (NB - 1);
      const uint32_t pos = gbase[d] + ((uint32_t)j - (uint32_t)lexcl[d]);
      keys_out[pos] = kk;
      vals_out[pos] = sval[j];
    }
  }
}

// ----------------------------------------------------------------------------------
// Grouped path, step 3: chunks.  A chunk is `chunk` consecutive slots of the sorted, padded group-instance array; all its
// items belong to one group (its first slot is always a real item, padding only follows the last item of a group).  One
// wave per chunk, lane t = tile t of the group.
// ----------------------------------------------------------------------------------
struct GroupArgs {
  int gx, gy, sgx;
  size_t chunks;             // upper bound of the number of chunks (Binning::chunks)
  uint32_t stage_cap;        // scatter pass: entries of the LDS stage (1024 or 2048; + 1 dump slot)
  const uint32_t* gkey;      // sorted + padded keys: group id | local rectangle << 16; GROUP_PAD in the padding slots
  const uint32_t* gval;      // sorted + padded Gaussian indices
  uint16_t* chunk_cnt;       // (chunks, 64)
  const uint32_t* chunk_pre; // (chunks, 64)
  const uint32_t* tile_start;
  uint32_t* point_list;
};

// The tiles of its group a group instance covers, from the local rectangle group_local_rect packed:
// bit (ty & 7) * 8 + (tx & 7), rows 0..3 in `lo`, rows 4..7 in `hi`.
__device__ __forceinline__ void group_mask(uint32_t lr, uint32_t& lo, uint32_t& hi) {
  const uint32_t x0 = lr & 7u, x1 = (lr >> 3) & 7u, y0 = (lr >> 6) & 7u, y1 = (lr >> 9) & 7u;  // inclusive bounds
  const uint32_t cols = ((2u << (x1 - x0)) - 1u) << x0;                                      // the covered columns, 8 bits
  // one 0x01 byte per covered row, times the column bits (no carries between the bytes)
  const uint64_t rows = ((~0ull >> (56u - 8u * (y1 - y0))) << (8u * y0)) & 0x0101010101010101ull;
  lo = (uint32_t)rows * cols;
  hi = (uint32_t)(rows >> 32) * cols;
}

// 64 x 64 bit transpose across the wave: in: lane j holds the tile mask of batch item j (lo: tiles 0..31, hi: 32..63);
// out: lane t holds the set of items that touch tile t.  Recursive block transpose in six butterfly stages: at distance d
// (32, 16, ..., 1) lane l and lane l ^ d exchange the off-diagonal d x d blocks -- the lane with bit d clear keeps its bits
// whose index has bit d clear and receives its partner's such bits d places higher; the other lane the mirror image.
//   d = 32  one v_permlane32_swap of the two halves;
//   d < 32  per 32-bit half: the partner's word (ds_swizzle xor 16 / 4, DPP row_ror:8 / quad_perm for 8 / 2 / 1), rotated
//           into place (v_alignbit_b32) and merged under a per-lane mask (v_bfi_b32);
// 31 instructions for the 4096 bits.  (The first version formed one ballot per tile and wrote it into lane t with
// v_writelane_b32: 272 instructions, and v_writelane_b32 turned out NOT to interlock on an SGPR a VALU instruction has
// just written -- three tile columns came out wrong exactly where hipcc had scheduled nothing in between.)
struct TransposeLane {
  uint32_t keep[5], rot[5];  // stages d = 16, 8, 4, 2, 1
};
__device__ __forceinline__ TransposeLane transpose_lane_consts(int lane) {
  constexpr uint32_t LOW[5] = {0x0000ffffu, 0x00ff00ffu, 0x0f0f0f0fu, 0x33333333u, 0x55555555u};
  TransposeLane c;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int d = 16 >> i;
    const bool up = (lane & d) != 0;
    c.keep[i] = up ? ~LOW[i] : LOW[i];
    c.rot[i] = up ? (uint32_t)d : (uint32_t)(32 - d);  // rotate right: the partner's bits move d places down / up
  }
  return c;
}
template <int STAGE>
__device__ __forceinline__ uint32_t transpose_partner(uint32_t x) {
  const int v = (int)x;
  if constexpr (STAGE == 0) return (uint32_t)__builtin_amdgcn_ds_swizzle(v, 0x401f);              // lane ^ 16
  else if constexpr (STAGE == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);  // row_ror:8 = lane ^ 8
  else if constexpr (STAGE == 2) return (uint32_t)__builtin_amdgcn_ds_swizzle(v, 0x101f);         // lane ^ 4
  else if constexpr (STAGE == 3) return (uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  else return (uint32_t)__builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false);                  // quad_perm [1,0,3,2]
}
template <int STAGE>
__device__ __forceinline__ void transpose_stage(const TransposeLane& c, uint32_t& lo, uint32_t& hi) {
  const uint32_t pl = transpose_partner<STAGE>(lo), ph = transpose_partner<STAGE>(hi);
  const uint32_t rl = __builtin_amdgcn_alignbit(pl, pl, c.rot[STAGE]), rh = __builtin_amdgcn_alignbit(ph, ph, c.rot[STAGE]);
  lo = (c.keep[STAGE] & lo) | (~c.keep[STAGE] & rl);  // v_bfi_b32
  hi = (c.keep[STAGE] & hi) | (~c.keep[STAGE] & rh);
}
__device__ __forceinline__ void transpose_masks(const TransposeLane& c, uint32_t lo, uint32_t hi, uint32_t& wlo, uint32_t& whi) {
  {
    const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);  // [lo.lower | hi.lower], [lo.upper | hi.upper]
    lo = r[0];
    hi = r[1];
  }
  transpose_stage<0>(c, lo, hi);
  transpose_stage<1>(c, lo, hi);
  transpose_stage<2>(c, lo, hi);
  transpose_stage<3>(c, lo, hi);
  transpose_stage<4>(c, lo, hi);
  wlo = lo;
  whi = hi;
}

// SCATTER = false: count pass (chunk_cnt[c][t] = instances of tile t in chunk c).
// SCATTER = true:  lane t appends the Gaussians of its tile, batch by batch in item order, at
//                  tile_start[tile] + chunk_pre[c][t]: the reference's point_list.
// NB = 64-item batches per chunk; all of a chunk's loads are issued before anything is used.
// The appended indices do not leave the wave one by one -- 64 lanes storing 4 bytes each to 64 different lists is one
// L2 request per element (measured: 47 us for 4.8 M elements, 10x the time of everything else in the kernel) -- but
// through an LDS stage of (index, destination) pairs: lane t deposits its tile's new entries in consecutive slots, and
// when the stage is full (or the chunk done) the wave writes the slots out with consecutive lanes on consecutive slots,
// so a tile's run leaves as one or two requests.  The stage (dynamic LDS) holds `stage_cap` entries + one dump slot;
// a batch with more entries than that (64 items x up to 64 tiles) is taken in halves or quarters of its items (a
// quarter, 16 items, never exceeds 1024 entries <= stage_cap).
template <bool SCATTER, int NB>
__global__ void __launch_bounds__(64) group_chunk_kernel(const GroupArgs a) {
  extern __shared__ __align__(8) unsigned char stage_raw[];
  uint2* const stage = reinterpret_cast<uint2*>(stage_raw);
  // Workgroup b runs on XCD b % 8 (tools/microbench/placement.hip), and the XCDs' L2s are separate: the chunks are dealt so
  // that every XCD gets a CONTIGUOUS eighth of them.  Consecutive chunks append to the same tiles' lists; written through
  // one L2 the pieces of a list merge into whole lines there instead of reaching memory as eight XCDs' partial lines.
  const size_t c = (size_t)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // (the grid is a multiple of 8)
  if (c >= a.chunks) return;
  const int lane = (int)threadIdx.x;
  const size_t base = c * (size_t)(64 * NB);
  uint32_t key[NB], idx[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    key[b] = a.gkey[base + (size_t)b * 64 + lane];
    idx[b] = SCATTER ? a.gval[base + (size_t)b * 64 + lane] : 0u;
  }
  const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)key[0]);
  if (first == GROUP_PAD) return;  // behind the last chunk (the grid is an upper bound)
  const TransposeLane tc = transpose_lane_consts(lane);
  const uint32_t cap = a.stage_cap;
  uint32_t cnt = 0, g = 0, staged = 0;  // g: where in point_list this lane's tile continues
  if (SCATTER) {
    const uint32_t s = first & 0xffffu;
    const uint32_t tx = ((s % (uint32_t)a.sgx) << GROUP_SHIFT) + (uint32_t)(lane & 7);
    const uint32_t ty = ((s / (uint32_t)a.sgx) << GROUP_SHIFT) + (uint32_t)(lane >> 3);
    const bool tile_ok = tx < (uint32_t)a.gx && ty < (uint32_t)a.gy;  // (a lane without a tile never sees a set bit)
    g = tile_ok ? a.tile_start[ty * (uint32_t)a.gx + tx] + a.chunk_pre[c * GROUP_TILES + lane] : 0u;
  }
  auto flush = [&]() {
    __syncthreads();  // (one wave: orders the LDS writes of all lanes before the reads)
    for (uint32_t e = (uint32_t)lane; e < staged; e += 64u) {
      const uint2 x = stage[e];
      a.point_list[x.y] = x.x;
    }
    __syncthreads();
    staged = 0;
  };
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const bool valid = key[b] != GROUP_PAD;
    if (b > 0 && __ballot(valid) == 0ull) break;  // padding only follows the items
    uint32_t lo = 0u, hi = 0u;
    if (valid) group_mask(key[b] >> 16, lo, hi);
    uint32_t wlo, whi;  // lane t: the batch items (bit j = item j) that touch tile t
    transpose_masks(tc, lo, hi, wlo, whi);
    if (!SCATTER) {
      cnt += (uint32_t)__popc(wlo) + (uint32_t)__popc(whi);
    } else {
      const uint32_t nl = (uint32_t)__popc(wlo), nh = (uint32_t)__popc(whi);
      const uint32_t tot_l = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp(nl), 63);
      const uint32_t tot_h = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp(nh), 63);
      const int nparts = tot_l + tot_h <= cap ? 1 : (max(tot_l, tot_h) <= cap ? 2 : 4);  // (uniform)
      for (int part = 0; part < nparts; ++part) {
        // this part's items as two 32-bit streams: A = items 0..31, B = items 32..63; A's entries precede B's
        uint32_t wa = wlo, wb = whi;
        if (nparts == 2) {
          wa = part == 0 ? wlo : 0u;
          wb = part == 0 ? 0u : whi;
        } else if (nparts == 4) {
          const uint32_t m = 0xffffu << (16 * (part & 1));
          wa = part < 2 ? (wlo & m) : 0u;
          wb = part < 2 ? 0u : (whi & m);
        }
        const uint32_t na = (uint32_t)__popc(wa), np = na + (uint32_t)__popc(wb);
        const uint32_t incl = wave_incl_scan_dpp(np);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (staged + total > cap) flush();
        uint32_t curA = staged + incl - np, curB = curA + na, gA = g, gB = g + na;
        // one entry of each stream per round: the two cross-lane reads (ds_bpermute) travel together; a lane whose
        // stream is exhausted writes into the dump slot behind the stage instead of branching
        while (__ballot((wa | wb) != 0u) != 0ull) {
          const bool va = wa != 0u, vb = wb != 0u;
          const int ja = va ? __builtin_ctz(wa) : 0, jb = vb ? 32 + __builtin_ctz(wb) : 0;
          const uint32_t xa = (uint32_t)__shfl((int)idx[b], ja, 64), xb = (uint32_t)__shfl((int)idx[b], jb, 64);
          stage[va ? curA : cap] = make_uint2(xa, gA);
          stage[vb ? curB : cap] = make_uint2(xb, gB);
          curA += va ? 1u : 0u;
          gA += va ? 1u : 0u;
          curB += vb ? 1u : 0u;
          gB += vb ? 1u : 0u;
          wa &= wa - 1u;
          wb &= wb - 1u;
        }
        staged += total;
        g += np;
      }
    }
  }
  if (SCATTER) flush();
  if (!SCATTER) a.chunk_cnt[c * GROUP_TILES + lane] = (uint16_t)cnt;
}

// Column scan: one workgroup per group; its chunks are consecutive rows [first, first + n) of the count table.  The
// waves take contiguous slabs of rows: slab sums -> exclusive prefix over the slabs (LDS) -> the rows' prefixes.  A wave
// keeps up to COLSCAN_REG rows in registers between the two steps (all loads of a slab in flight together; longer slabs
// are read twice).
constexpr int COLSCAN_WAVES = 16, COLSCAN_REG = 16;
__global__ void __launch_bounds__(COLSCAN_WAVES * 64) group_colscan_kernel(int gx, int gy, int sgx,
                                                                          const uint32_t* __restrict__ group_first,
                                                                          const uint16_t* __restrict__ chunk_cnt,
                                                                          uint32_t* __restrict__ chunk_pre,
                                                                          uint32_t* __restrict__ tile_total) {
  __shared__ uint32_t slab[COLSCAN_WAVES][GROUP_TILES];
  const uint32_t s = blockIdx.x;
  const int w = (int)(threadIdx.x >> 6), lane = lane_id();
  const uint32_t first = group_first[s], n = group_first[s + 1] - first;  // (left by the group sort's scatter kernel)
  const uint32_t per = (n + COLSCAN_WAVES - 1) / COLSCAN_WAVES;
  const uint32_t r0 = min(n, (uint32_t)w * per), r1 = min(n, r0 + per);
  const uint16_t* __restrict__ src = chunk_cnt + (size_t)(first + r0) * GROUP_TILES + lane;
  uint32_t* __restrict__ out = chunk_pre + (size_t)(first + r0) * GROUP_TILES + lane;
  const uint32_t rows = r1 - r0;
  // a slab is read COLSCAN_REG rows at a time, all loads of a round issued together (unconditionally, row index clamped: a
  // predicated or a dependent load costs one memory round trip each -- the central groups of an image have several times
  // the average number of chunks, and a row-by-row loop over their slabs made this kernel 14 us)
  auto load_rows = [&](uint32_t first_row, uint32_t (&v)[COLSCAN_REG]) {
    const uint32_t last = rows - 1u;
    uint32_t raw[COLSCAN_REG];
#pragma unroll
    for (int i = 0; i < COLSCAN_REG; ++i) raw[i] = (uint32_t)src[(size_t)min(first_row + (uint32_t)i, last) * GROUP_TILES];
#pragma unroll
    for (int i = 0; i < COLSCAN_REG; ++i) v[i] = first_row + (uint32_t)i < rows ? raw[i] : 0u;
  };
  uint32_t v0[COLSCAN_REG];  // the first round stays in registers between the two steps
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < COLSCAN_REG; ++i) v0[i] = 0u;
  if (rows != 0u) {
    load_rows(0u, v0);
#pragma unroll
    for (int i = 0; i < COLSCAN_REG; ++i) sum += v0[i];
    for (uint32_t r = COLSCAN_REG; r < rows; r += COLSCAN_REG) {
      uint32_t v[COLSCAN_REG];
      load_rows(r, v);
#pragma unroll
      for (int i = 0; i < COLSCAN_REG; ++i) sum += v[i];
    }
  }
  slab[w][lane] = sum;
  __syncthreads();
  uint32_t run = 0, total = 0;
#pragma unroll
  for (int i = 0; i < COLSCAN_WAVES; ++i) {
    const uint32_t x = slab[i][lane];
    run += i < w ? x : 0u;
    total += x;
  }
#pragma unroll
  for (int i = 0; i < COLSCAN_REG; ++i) {
    if ((uint32_t)i < rows) out[(size_t)i * GROUP_TILES] = run;
    run += v0[i];
  }
  for (uint32_t r = COLSCAN_REG; r < rows; r += COLSCAN_REG) {
    uint32_t v[COLSCAN_REG];
    load_rows(r, v);
#pragma unroll
    for (int i = 0; i < COLSCAN_REG; ++i) {
      if (r + (uint32_t)i < rows) out[(size_t)(r + (uint32_t)i) * GROUP_TILES] = run;
      run += v[i];
    }
  }
  if (w == 0) {
    const uint32_t tx = ((s % (uint32_t)sgx) << GROUP_SHIFT) + (uint32_t)(lane & 7);
    const uint32_t ty = ((s / (uint32_t)sgx) << GROUP_SHIFT) + (uint32_t)(lane >> 3);
    if (tx < (uint32_t)gx && ty < (uint32_t)gy) tile_total[ty * (uint32_t)gx + tx] = total;
  }
}

template <bool SCATTER>
static void launch_chunks(hipStream_t s, const Binning& b, const GroupArgs& a) {
  const dim3 grid((unsigned)((b.chunks + 7) / 8 * 8)), block(64);  // a multiple of 8: see the chunk -> XCD dealing in the kernel
  const size_t lds = SCATTER ? sizeof(uint2) * ((size_t)a.stage_cap + 1) : 0;
  switch (b.chunk / 64) {
    case 1: hipLaunchKernelGGL((group_chunk_kernel<SCATTER, 1>), grid, block, lds, s, a); break;
    case 2: hipLaunchKernelGGL((group_chunk_kernel<SCATTER, 2>), grid, block, lds, s, a); break;
    case 4: hipLaunchKernelGGL((group_chunk_kernel<SCATTER, 4>), grid, block, lds, s, a); break;
    default: hipLaunchKernelGGL((group_chunk_kernel<SCATTER, 8>), grid, block, lds, s, a); break;
  }
}

static int64_t checkpoint_state_words(const Image& im);
template <int BITS>
static void launch_grouped(hipStream_t s, int P, int64_t R, int gx, int gy, const Geom& g, const Binning& b, const Image& im) {
  const int nbg = (P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  const int64_t padded = b.chunks * (int64_t)b.chunk;
  const int fill_blocks = 512;  // (two per CU: 5 MB of padding at 1 M Gaussians)
  hipLaunchKernelGGL(emit_groups_kernel, dim3(nbg + fill_blocks), dim3(GAUSS_BLOCK), 0, s, P, b.sgx, g, b.gkey[0], b.gval[0], nbg,
                     b.gkey[1], padded, reinterpret_cast<uint4*>(im.ck_work), checkpoint_state_words(im) / 4);
  const bool wide = b.sort_blocks <= SORT_WIDE_MAX_BLOCKS;  // few blocks: 1024 threads per block (see radix_sort_pairs)
  if (wide)
    hipLaunchKernelGGL((group_hist_kernel<BITS, 1024>), dim3(b.sort_blocks), dim3(1024), 0, s, (const uint32_t*)b.gkey[0], b.G,
                       b.ghist, b.sort_blocks, b.gkey[1], padded);
  else
    hipLaunchKernelGGL((group_hist_kernel<BITS, SORT_THREADS>), dim3(b.sort_blocks), dim3(SORT_THREADS), 0, s,
                       (const uint32_t*)b.gkey[0], b.G, b.ghist, b.sort_blocks, b.gkey[1], padded);
  hipLaunchKernelGGL(sort_scan_kernel, dim3(1 << BITS), dim3(SORT_THREADS), 0, s, b.ghist, b.gbin_total, b.sort_blocks);
  if (wide)
    hipLaunchKernelGGL((group_scatter_kernel<BITS, 1024>), dim3(b.sort_blocks), dim3(1024), 0, s, (const uint32_t*)b.gkey[0],
                       (const uint32_t*)b.gval[0], b.gkey[1], b.gval[1], b.G, (const uint32_t*)b.ghist,
                       (const uint32_t*)b.gbin_total, b.sort_blocks, (uint32_t)b.chunk, b.group_first);
  else
    hipLaunchKernelGGL((group_scatter_kernel<BITS, SORT_THREADS>), dim3(b.sort_blocks), dim3(SORT_THREADS), 0, s,
                       (const uint32_t*)b.gkey[0], (const uint32_t*)b.gval[0], b.gkey[1], b.gval[1], b.G, (const uint32_t*)b.ghist,
                       (const uint32_t*)b.gbin_total, b.sort_blocks, (uint32_t)b.chunk, b.group_first);
  GroupArgs a;
  a.gx = gx; a.gy = gy; a.sgx = b.sgx; a.chunks = (size_t)b.chunks;
  // stage of the scatter pass: a whole chunk's entries when they fit 1024 (8 KB of LDS: 20 waves per CU), else 2048
  a.stage_cap = (R / b.G + 1) * (int64_t)b.chunk <= 820 ? 1024u : 2048u;
  a.gkey = b.gkey[1]; a.gval = b.gval[1];
  a.chunk_cnt = b.chunk_cnt; a.chunk_pre = b.chunk_pre; a.tile_start = b.tile_start; a.point_list = b.point_list;
  launch_chunks<false>(s, b, a);
  hipLaunchKernelGGL(group_colscan_kernel, dim3(b.groups), dim3(COLSCAN_WAVES * 64), 0, s, gx, gy, b.sgx,
                     (const uint32_t*)b.group_first, (const uint16_t*)b.chunk_cnt, b.chunk_pre, b.tile_total);
  hipLaunchKernelGGL(tile_worklist_kernel, dim3(2), dim3(1024), 0, s, gx * gy, im.ranges, im.work_order, im.work_meta,
                     im.queue_heads, im.work_est, (const uint32_t*)b.tile_total, b.tile_start, im.ck_table,
                     (uint32_t)ck_tiles((size_t)gx * gy));
  launch_chunks<true>(s, b, a);
}

// The words between Image::ck_work and the end of Image::tile_maxc (work split, walk depths: contiguous sections).
static int64_t checkpoint_state_words(const Image& im) {
  return (int64_t)((reinterpret_cast<const char*>(im.ck_pool) - reinterpret_cast<const char*>(im.ck_work)) / 4);
}

hipError_t launch_binning(hipStream_t s, int P, int64_t R, int W, int H, const Geom& g, const Binning& b, const Image& im) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  if (R <= 0 || b.legacy) {  // (the grouped path clears the checkpoint state in its emit launch)
    hipError_t e = hipMemsetAsync(im.ck_work, 0, sizeof(uint32_t) * (size_t)checkpoint_state_words(im), s);
    if (e != hipSuccess) return e;
  }
  if (R <= 0) {
    hipError_t e = hipMemsetAsync(im.ranges, 0, sizeof(uint2) * (size_t)gx * gy, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tile_worklist_kernel, dim3(1), dim3(1024), 0, s, gx * gy, im.ranges, im.work_order, im.work_meta,
                       im.queue_heads, im.work_est, (const uint32_t*)nullptr, (uint32_t*)nullptr, im.ck_table,
                       (uint32_t)ck_tiles((size_t)gx * gy));
    return hipGetLastError();
  }
  if (!b.legacy) {
    if (b.group_bits == 8) launch_grouped<8>(s, P, R, gx, gy, g, b, im);
    else launch_grouped<11>(s, P, R, gx, gy, g, b, im);
    return hipGetLastError();
  }
  const int nbg = (P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  const dim3 ge(max(nbg, (gx * gy + GAUSS_BLOCK - 1) / GAUSS_BLOCK)), gr((unsigned)((R + 256 * RANGE_PER - 1) / (256 * RANGE_PER)));
  if (b.key_bytes == 2) {  // tile ids fit 16 bits: 6 instead of 8 bytes per sorted pair
    uint16_t* const tk[2] = {(uint16_t*)b.tkey[0], (uint16_t*)b.tkey[1]};
    hipLaunchKernelGGL(emit_keys_kernel<uint16_t>, ge, dim3(GAUSS_BLOCK), 0, s, P, gx, gx * gy, g, tk[0], b.vals[0], im.ranges);
    radix_sort_pairs<uint16_t>(s, tk, b.vals, R, b.passes, b.digit_bits, b.hist, b.bin_total, false);
    hipLaunchKernelGGL(tile_ranges_kernel<uint16_t>, gr, dim3(256), 0, s, R, (const uint16_t*)tk[b.final_buf], im.ranges);
  } else {
    uint32_t* const tk[2] = {(uint32_t*)b.tkey[0], (uint32_t*)b.tkey[1]};
    hipLaunchKernelGGL(emit_keys_kernel<uint32_t>, ge, dim3(GAUSS_BLOCK), 0, s, P, gx, gx * gy, g, tk[0], b.vals[0], im.ranges);
    radix_sort_pairs<uint32_t>(s, tk, b.vals, R, b.passes, b.digit_bits, b.hist, b.bin_total, false);
    hipLaunchKernelGGL(tile_ranges_kernel<uint32_t>, gr, dim3(256), 0, s, R, (const uint32_t*)tk[b.final_buf], im.ranges);
  }
  hipLaunchKernelGGL(tile_worklist_kernel, dim3(1), dim3(1024), 0, s, gx * gy, im.ranges, im.work_order, im.work_meta,
                     im.queue_heads, im.work_est, (const uint32_t*)nullptr, (uint32_t*)nullptr, im.ck_table,
                     (uint32_t)ck_tiles((size_t)gx * gy));
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
// grouped path: the tile of instance i is the last one whose run starts at or before i (tile_start is monotone)
__global__ void __launch_bounds__(256) export_keys_grouped_kernel(int64_t R, uint32_t T, const uint32_t* __restrict__ tile_start,
                                                                 const uint32_t* __restrict__ vals,
                                                                 const float4* __restrict__ rec1, uint64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  uint32_t lo = 0, hi = T;  // invariant: tile_start[lo] <= i < tile_start[hi]
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((int64_t)tile_start[mid] <= i) lo = mid; else hi = mid;
  }
  keys[i] = ((uint64_t)lo << 32) | (uint64_t)__float_as_uint(rec1[vals[i]].z);
}
hipError_t launch_export_keys(hipStream_t s, int64_t R, int W, int H, const Binning& b, const Geom& g, uint64_t* keys) {
  const dim3 grid((unsigned)((R + 255) / 256));
  if (!b.legacy) {
    const uint32_t T = (uint32_t)(((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE));
    hipLaunchKernelGGL(export_keys_grouped_kernel, grid, dim3(256), 0, s, R, T, (const uint32_t*)b.tile_start,
                       (const uint32_t*)b.point_list, g.rec1, keys);
  } else if (b.key_bytes == 2) {
    hipLaunchKernelGGL(export_keys_kernel<uint16_t>, grid, dim3(256), 0, s, R, (const uint16_t*)b.tkey[b.final_buf],
                       b.vals[b.final_buf], g.rec1, keys);
  } else {
    hipLaunchKernelGGL(export_keys_kernel<uint32_t>, grid, dim3(256), 0, s, R, (const uint32_t*)b.tkey[b.final_buf],
                       b.vals[b.final_buf], g.rec1, keys);
  }
  return hipGetLastError();
}

}  // namespace gsr
