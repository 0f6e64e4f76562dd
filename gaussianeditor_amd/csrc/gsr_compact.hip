// gsr_compact.hip -- stable stream compaction of the rows of MANY tensors by one keep mask (SURVEY.md section 8(f)
// rank 4).
//
// The reference prunes its model with boolean-mask indexing, one tensor at a time: six parameters, their twelve Adam
// moment tensors and five bookkeeping tensors (gaussiansplatting/scene/gaussian_model.py:568-609, prune_points /
// _prune_optimizer): 23 x (nonzero + host sync + gather).  Here the mask is scanned once (one host readback for the
// number of survivors, needed to size the outputs) and one launch moves the surviving rows of every tensor; the order
// of the survivors is the original one, so the result is what `tensor[mask]` returns, bit for bit.
#include <string.h>

#include "gsr_kernels.h"

namespace gsr {

constexpr int CMP_ROWS = VIEW_MSG_ROWS;  // rows per block (1024; the view messages of the multi-GPU exchange carry block_off)
constexpr int CMP_MAX_TENSORS = 32;

struct CompactWork {
  uint32_t* block_count;  // (nb)
  uint32_t* block_off;    // (nb) exclusive prefix
  uint64_t* total;        // (1)
  size_t bytes;
};
__host__ __device__ inline CompactWork carve_compact(void* base, int64_t P) {
  char* p = (char*)base;
  const size_t nb = ((size_t)P + CMP_ROWS - 1) / CMP_ROWS;
  CompactWork w;
  size_t off = 0;
  w.total = (uint64_t*)(p + off);        off += 256;
  w.block_count = (uint32_t*)(p + off);  off += align_up(sizeof(uint32_t) * nb);
  w.block_off = (uint32_t*)(p + off);    off += align_up(sizeof(uint32_t) * nb);
  w.bytes = off;
  return w;
}
size_t compact_workspace_bytes(int64_t P) { return carve_compact(nullptr, P).bytes; }

__global__ void __launch_bounds__(CMP_ROWS) compact_count_kernel(int64_t P, const uint8_t* __restrict__ keep,
                                                                uint32_t* __restrict__ block_count) {
  __shared__ uint32_t smem[CMP_ROWS / 64 + 1];
  const int64_t r = (int64_t)blockIdx.x * CMP_ROWS + threadIdx.x;
  const uint32_t k = (r < P && keep[r] != 0) ? 1u : 0u;
  uint32_t total;
  (void)block_excl_scan_u32<CMP_ROWS>(k, &total, smem);
  if (threadIdx.x == 0) block_count[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024) compact_scan_kernel(int64_t nb, const uint32_t* __restrict__ cnt,
                                                           uint32_t* __restrict__ off, uint64_t* __restrict__ total) {
  __shared__ uint32_t smem[1024 / 64 + 1];
  uint64_t carry = 0;
  for (int64_t base = 0; base < nb; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const uint32_t v = i < nb ? cnt[i] : 0u;
    uint32_t chunk;
    const uint32_t ex = block_excl_scan_u32<1024>(v, &chunk, smem);
    if (i < nb) off[i] = (uint32_t)carry + ex;
    carry += chunk;
  }
  if (threadIdx.x == 0) *total = carry;
}

struct CompactTensorDev {
  const uint8_t* src;
  uint8_t* dst;
  uint32_t row_bytes;
};
struct CompactArgs {
  CompactTensorDev t[CMP_MAX_TENSORS];
  int nt;
  int64_t P;
  const uint8_t* keep;
  const uint32_t* block_off;
  uint32_t limit;  // surviving rows beyond this many are dropped (a destination that holds only `limit` rows)
};

constexpr int CMP_TENSORS_PER_BLOCK = 8;  // tensors a block moves after scanning its part of the mask once
// grid = (row blocks, ceil(tensors / CMP_TENSORS_PER_BLOCK))
__global__ void __launch_bounds__(CMP_ROWS) compact_apply_kernel(const CompactArgs a) {
  __shared__ uint32_t smem[CMP_ROWS / 64 + 1];
  __shared__ uint32_t pos[CMP_ROWS];
  const int64_t row0 = (int64_t)blockIdx.x * CMP_ROWS;
  const int64_t r = row0 + threadIdx.x;
  const uint32_t k = (r < a.P && a.keep[r] != 0) ? 1u : 0u;
  uint32_t total;
  const uint32_t ex = block_excl_scan_u32<CMP_ROWS>(k, &total, smem);
  const uint32_t my_pos = a.block_off[blockIdx.x] + ex;
  pos[threadIdx.x] = (k && my_pos < a.limit) ? my_pos : 0xffffffffu;
  __syncthreads();
  if (total == 0) return;
  const uint32_t nrows = (uint32_t)min((int64_t)CMP_ROWS, a.P - row0);
  const int t_end = min(a.nt, ((int)blockIdx.y + 1) * CMP_TENSORS_PER_BLOCK);
  for (int ti = (int)blockIdx.y * CMP_TENSORS_PER_BLOCK; ti < t_end; ++ti) {
  const CompactTensorDev t = a.t[ti];
  if (t.src == nullptr) {  // "iota" source: the surviving row numbers themselves (int32), internal callers only
    const uint32_t p = pos[threadIdx.x];
    if (p != 0xffffffffu) reinterpret_cast<int32_t*>(t.dst)[p] = (int32_t)r;
    continue;
  }
  if ((t.row_bytes & 3u) == 0 && t.row_bytes <= 16u && (((uintptr_t)t.src | (uintptr_t)t.dst) & 3u) == 0) {
    // short rows (1..4 words: positions, scales, quaternions, opacities, ...): each thread moves its own row
    const uint32_t W = t.row_bytes >> 2, p = pos[threadIdx.x];
    if (p != 0xffffffffu) {
      const uint32_t* __restrict__ s = reinterpret_cast<const uint32_t*>(t.src) + (size_t)r * W;
      uint32_t* __restrict__ d = reinterpret_cast<uint32_t*>(t.dst) + (size_t)p * W;
      uint32_t v[4];
#pragma unroll
      for (uint32_t c = 0; c < 4; ++c) v[c] = c < W ? s[c] : 0u;
#pragma unroll
      for (uint32_t c = 0; c < 4; ++c)
        if (c < W) d[c] = v[c];
    }
  } else if ((t.row_bytes & 3u) == 0 && (((uintptr_t)t.src | (uintptr_t)t.dst) & 3u) == 0) {
    const uint32_t W = t.row_bytes >> 2;
    const uint32_t* __restrict__ s = reinterpret_cast<const uint32_t*>(t.src) + (size_t)row0 * W;
    uint32_t* __restrict__ d = reinterpret_cast<uint32_t*>(t.dst);
    const uint32_t n = nrows * W;
    for (uint32_t e = threadIdx.x; e < n; e += CMP_ROWS) {
      const uint32_t rr = e / W, c = e - rr * W;
      const uint32_t p = pos[rr];
      if (p != 0xffffffffu) d[(size_t)p * W + c] = s[e];
    }
  } else {
    const uint32_t W = t.row_bytes;
    const uint8_t* __restrict__ s = t.src + (size_t)row0 * W;
    const uint32_t n = nrows * W;
    for (uint32_t e = threadIdx.x; e < n; e += CMP_ROWS) {
      const uint32_t rr = e / W, c = e - rr * W;
      const uint32_t p = pos[rr];
      if (p != 0xffffffffu) t.dst[(size_t)p * W + c] = s[e];
    }
  }
  }
}

hipError_t launch_compact_plan(hipStream_t s, int64_t P, const uint8_t* keep, void* workspace) {
  const CompactWork w = carve_compact(workspace, P);
  const int64_t nb = (P + CMP_ROWS - 1) / CMP_ROWS;
  hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)nb), dim3(CMP_ROWS), 0, s, P, keep, w.block_count);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, nb, w.block_count, w.block_off, w.total);
  return hipGetLastError();
}
const uint64_t* compact_total_ptr(void* workspace, int64_t P) { return carve_compact(workspace, P).total; }
const uint32_t* compact_block_off_ptr(void* workspace, int64_t P) { return carve_compact(workspace, P).block_off; }

hipError_t launch_compact_apply(hipStream_t s, int64_t P, const uint8_t* keep, void* workspace, int nt,
                                const gsr_compact_tensor* tensors, int64_t limit) {
  if (nt <= 0 || P <= 0) return hipSuccess;
  if (nt > CMP_MAX_TENSORS) return hipErrorInvalidValue;
  const CompactWork w = carve_compact(workspace, P);
  CompactArgs a;
  memset(&a, 0, sizeof(a));
  a.P = P;
  a.nt = nt;
  a.keep = keep;
  a.block_off = w.block_off;
  a.limit = (limit < 0 || limit > 0xfffffffell) ? 0xffffffffu : (uint32_t)limit;
  for (int i = 0; i < nt; ++i) {
    a.t[i].src = (const uint8_t*)tensors[i].src;
    a.t[i].dst = (uint8_t*)tensors[i].dst;
    a.t[i].row_bytes = (uint32_t)tensors[i].row_bytes;
  }
  const int64_t nb = (P + CMP_ROWS - 1) / CMP_ROWS;
  hipLaunchKernelGGL(compact_apply_kernel, dim3((unsigned)nb, (unsigned)((nt + CMP_TENSORS_PER_BLOCK - 1) / CMP_TENSORS_PER_BLOCK)),
                     dim3(CMP_ROWS), 0, s, a);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// Row append for MANY tensors in one launch (densification): dst = [src (P rows) ; ext (n rows) or zeros].
// Replaces the torch.cat + torch.zeros_like pairs of cat_tensors_to_optimizer
// (gaussiansplatting/scene/gaussian_model.py:609-641: per parameter group the parameter itself and its two Adam
// moments, the moments extended by zeros).  Pure streaming: the P old rows move as 16-byte vectors.
// ----------------------------------------------------------------------------------
struct AppendTensorDev {
  const uint8_t* src;
  const uint8_t* ext;  // null: the appended rows are zero
  uint8_t* dst;
  uint64_t old_bytes;  // P * row_bytes
  uint64_t new_bytes;  // n * row_bytes
};
struct AppendArgs {
  AppendTensorDev t[CMP_MAX_TENSORS];
};
// grid = (blocks per tensor, tensors)
__global__ void __launch_bounds__(256) append_rows_kernel(const AppendArgs a) {
  const AppendTensorDev t = a.t[blockIdx.y];
  const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, stride = (uint64_t)gridDim.x * 256;
  // the old rows: 16-byte vectors where source and destination allow it, then the odd bytes
  uint64_t done = 0;
  if ((((uintptr_t)t.src | (uintptr_t)t.dst) & 15u) == 0) {
    const uint64_t nv = t.old_bytes >> 4;
    const uint4* __restrict__ s = reinterpret_cast<const uint4*>(t.src);
    uint4* __restrict__ d = reinterpret_cast<uint4*>(t.dst);
    for (uint64_t i = tid; i < nv; i += stride) d[i] = s[i];
    done = nv << 4;
  }
  for (uint64_t i = done + tid; i < t.old_bytes; i += stride) t.dst[i] = t.src[i];
  // the appended rows (few): 4-byte words when everything is word aligned, bytes otherwise
  uint8_t* __restrict__ d2 = t.dst + t.old_bytes;
  if ((((uintptr_t)d2 | (uintptr_t)t.ext | t.new_bytes) & 3u) == 0) {
    const uint64_t nw = t.new_bytes >> 2;
    const uint32_t* __restrict__ e = reinterpret_cast<const uint32_t*>(t.ext);
    uint32_t* __restrict__ d = reinterpret_cast<uint32_t*>(d2);
    for (uint64_t i = tid; i < nw; i += stride) d[i] = e ? e[i] : 0u;
  } else {
    for (uint64_t i = tid; i < t.new_bytes; i += stride) d2[i] = t.ext ? t.ext[i] : (uint8_t)0;
  }
}

hipError_t launch_append_rows(hipStream_t s, int64_t P, int64_t n, int nt, const gsr_append_tensor* tensors) {
  if (nt <= 0 || P + n <= 0) return hipSuccess;
  if (nt > CMP_MAX_TENSORS) return hipErrorInvalidValue;
  AppendArgs a;
  memset(&a, 0, sizeof(a));
  uint64_t most = 0;
  for (int i = 0; i < nt; ++i) {
    a.t[i].src = (const uint8_t*)tensors[i].src;
    a.t[i].ext = (const uint8_t*)tensors[i].ext;
    a.t[i].dst = (uint8_t*)tensors[i].dst;
    a.t[i].old_bytes = (uint64_t)P * (uint64_t)tensors[i].row_bytes;
    a.t[i].new_bytes = (uint64_t)n * (uint64_t)tensors[i].row_bytes;
    const uint64_t b = a.t[i].old_bytes + a.t[i].new_bytes;
    if (b > most) most = b;
  }
  // enough blocks for the largest tensor at 64 bytes per thread, capped (the loops are grid-stride)
  const uint64_t want = (most + 256ull * 64ull - 1) / (256ull * 64ull);
  const unsigned bx = (unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(append_rows_kernel, dim3(bx, (unsigned)nt), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace gsr
