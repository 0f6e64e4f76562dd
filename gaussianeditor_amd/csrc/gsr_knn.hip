// gsr_knn.hip -- mean squared distance to the 3 nearest neighbours of every point: the `distCUDA2` of the
// reference's simple-knn submodule (gaussiansplatting/submodules/simple-knn/simple_knn.cu:185-221,
// spatial.cu:15-25), used once per scene to initialise the Gaussian scales
// (gaussiansplatting/scene/gaussian_model.py:288-291).  SURVEY.md section 8(f) rank 1: without it the reference's
// GaussianModel cannot even be imported on ROCm.
//
// The reference's result is the EXACT 3-NN mean (its Morton ordering and 1024-point boxes only prune the
// search, simple_knn.cu:147-183), so any exact search that forms the squared distance as
// dx*dx + dy*dy + dz*dz in float and averages (b0 + b1 + b2) / 3.0f reproduces it bit for bit.
//
// Pipeline (no host round trips; the reference does two blocking D2H copies for the bounds):
//   bounds (two-level min/max) -> 30-bit Morton codes -> stable radix sort (the rasterizer's sort) ->
//   points gathered into Morton order (float4, coalesced) -> per-1024-point boxes -> pruned exact search.
#include <float.h>

#include "gsr_kernels.h"

namespace gsr {

constexpr int KNN_BOX = 1024;  // BOX_SIZE, simple_knn.cu:12

struct KnnWork {
  float* bounds;        // [nb][6] per-block min xyz, max xyz; final result in bounds[0..5]
  uint32_t* key[2];     // Morton codes (ping-pong)
  uint32_t* val[2];     // point indices (ping-pong); val[0] = Morton order after 4 passes
  uint32_t* hist;       // 256 * nsb
  uint32_t* bin_total;  // 512
  float4* sorted;       // points in Morton order
  float* boxes;         // [nboxes][6]
  size_t bytes;
};

__host__ __device__ inline KnnWork carve_knn(void* base, int P) {
  char* p = (char*)base;
  KnnWork w;
  size_t off = 0;
  const size_t nb = ((size_t)P + 255) / 256, nsb = ((size_t)P + SORT_KPB - 1) / SORT_KPB, nboxes = ((size_t)P + KNN_BOX - 1) / KNN_BOX;
  w.bounds = (float*)(p + off);       off += align_up(sizeof(float) * 6 * (nb + 1));
  for (int i = 0; i < 2; ++i) { w.key[i] = (uint32_t*)(p + off); off += align_up(sizeof(uint32_t) * (size_t)P); }
  for (int i = 0; i < 2; ++i) { w.val[i] = (uint32_t*)(p + off); off += align_up(sizeof(uint32_t) * (size_t)P); }
  w.hist = (uint32_t*)(p + off);      off += align_up(sizeof(uint32_t) * 256 * nsb);
  w.bin_total = (uint32_t*)(p + off); off += align_up(sizeof(uint32_t) * 512);
  w.sorted = (float4*)(p + off);      off += align_up(sizeof(float4) * (size_t)P);
  w.boxes = (float*)(p + off);        off += align_up(sizeof(float) * 6 * nboxes);
  w.bytes = off;
  return w;
}

template <int NT>
__device__ __forceinline__ void block_minmax(float (&mn)[3], float (&mx)[3], float* smem /* NT/64 * 6 */) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      mn[c] = fminf(mn[c], __shfl_xor(mn[c], d, 64));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], d, 64));
    }
  }
  const int w = (int)(threadIdx.x >> 6), l = lane_id();
  __syncthreads();
  if (l == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      smem[w * 6 + c] = mn[c];
      smem[w * 6 + 3 + c] = mx[c];
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    mn[c] = smem[c];
    mx[c] = smem[3 + c];
    for (int i = 1; i < NT / 64; ++i) {
      mn[c] = fminf(mn[c], smem[i * 6 + c]);
      mx[c] = fmaxf(mx[c], smem[i * 6 + 3 + c]);
    }
  }
}

__global__ void __launch_bounds__(256) knn_bounds_kernel(int P, const float* __restrict__ pts, float* __restrict__ bounds) {
  __shared__ float smem[4 * 6];
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (i < P) {
#pragma unroll
    for (int c = 0; c < 3; ++c) mn[c] = mx[c] = pts[3 * (size_t)i + c];
  }
  block_minmax<256>(mn, mx, smem);
  if (threadIdx.x == 0) {
    float* o = bounds + 6 * (size_t)(blockIdx.x + 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      o[c] = mn[c];
      o[3 + c] = mx[c];
    }
  }
}
__global__ void __launch_bounds__(1024) knn_bounds_final_kernel(int nb, float* __restrict__ bounds) {
  __shared__ float smem[16 * 6];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int b = (int)threadIdx.x; b < nb; b += 1024) {
    const float* o = bounds + 6 * (size_t)(b + 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mn[c] = fminf(mn[c], o[c]);
      mx[c] = fmaxf(mx[c], o[3 + c]);
    }
  }
  block_minmax<1024>(mn, mx, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      bounds[c] = mn[c];
      bounds[3 + c] = mx[c];
    }
  }
}

// simple_knn.cu:45-61.  The code only orders the search; it never changes the result.
__device__ __forceinline__ uint32_t prep_morton(uint32_t x) {
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}
__global__ void __launch_bounds__(256) knn_morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ bounds,
                                                        uint32_t* __restrict__ codes) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= P) return;
  uint32_t q[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float lo = bounds[c], hi = bounds[3 + c];
    float t = ((pts[3 * (size_t)i + c] - lo) / (hi - lo)) * (float)((1 << 10) - 1);
    t = (t == t) ? fminf(fmaxf(t, 0.f), 1023.f) : 0.f;  // degenerate extents: any code is fine
    q[c] = prep_morton((uint32_t)t);
  }
  codes[i] = q[0] | (q[1] << 1) | (q[2] << 2);
}

__global__ void __launch_bounds__(256) knn_gather_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                        float4* __restrict__ sorted) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= P) return;
  const size_t j = order[i];
  sorted[i] = make_float4(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], 0.f);
}

// boxMinMax, simple_knn.cu:78-117: bounds of every run of KNN_BOX Morton-consecutive points.
__global__ void __launch_bounds__(256) knn_box_kernel(int P, const float4* __restrict__ sorted, float* __restrict__ boxes) {
  __shared__ float smem[4 * 6];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int k = 0; k < KNN_BOX / 256; ++k) {
    const int i = (int)blockIdx.x * KNN_BOX + k * 256 + (int)threadIdx.x;
    if (i < P) {
      const float4 p = sorted[i];
      mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
      mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
  }
  block_minmax<256>(mn, mx, smem);
  if (threadIdx.x == 0) {
    float* o = boxes + 6 * (size_t)blockIdx.x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      o[c] = mn[c];
      o[3 + c] = mx[c];
    }
  }
}

// updateKBest<3>, simple_knn.cu:131-145
__device__ __forceinline__ void update3(const float4& ref, const float4& p, float (&knn)[3]) {
  const float dx = p.x - ref.x, dy = p.y - ref.y, dz = p.z - ref.z;
  float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    if (knn[j] > dist) {
      const float t = knn[j];
      knn[j] = dist;
      dist = t;
    }
  }
}
// distBoxPoint, simple_knn.cu:119-129
__device__ __forceinline__ float dist_box_point(const float* __restrict__ b, const float4& p) {
  float dx = 0.f, dy = 0.f, dz = 0.f;
  if (p.x < b[0] || p.x > b[3]) dx = fminf(fabsf(p.x - b[0]), fabsf(p.x - b[3]));
  if (p.y < b[1] || p.y > b[4]) dy = fminf(fabsf(p.y - b[1]), fabsf(p.y - b[4]));
  if (p.z < b[2] || p.z > b[5]) dz = fminf(fabsf(p.z - b[2]), fabsf(p.z - b[5]));
  return dx * dx + dy * dy + dz * dz;
}

// boxMeanDist, simple_knn.cu:147-183
__global__ void __launch_bounds__(256) knn_search_kernel(int P, const float4* __restrict__ sorted, const uint32_t* __restrict__ order,
                                                        const float* __restrict__ boxes, float* __restrict__ dists) {
  const int idx = (int)(blockIdx.x * 256 + threadIdx.x);
  if (idx >= P) return;
  const float4 point = sorted[idx];
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
    if (i == idx) continue;
    update3(point, sorted[i], best);
  }
  const float reject = best[2];
  best[0] = best[1] = best[2] = FLT_MAX;
  const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
  for (int b = 0; b < nboxes; b++) {
    const float d = dist_box_point(boxes + 6 * (size_t)b, point);
    if (d > reject || d > best[2]) continue;
    const int e = min(P, (b + 1) * KNN_BOX);
    for (int i = b * KNN_BOX; i < e; i++) {
      if (i == idx) continue;
      update3(point, sorted[i], best);
    }
  }
  dists[order[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
}

void radix_sort_pairs_u32(hipStream_t s, uint32_t* const keys[2], uint32_t* const vals[2], int64_t n, int npass,
                          const int* digit_bits, uint32_t* hist, uint32_t* bin_total, bool iota_first);  // gsr_binning.hip

size_t knn_workspace_bytes(int P) { return carve_knn(nullptr, P).bytes; }

// bounds -> Morton codes -> sort -> gather -> boxes over `points`; leaves w.sorted / w.val[0] / w.boxes
static void build_knn_index(hipStream_t s, int P, const float* points, const KnnWork& w) {
  const int nb = (P + 255) / 256;
  hipLaunchKernelGGL(knn_bounds_kernel, dim3(nb), dim3(256), 0, s, P, points, w.bounds);
  hipLaunchKernelGGL(knn_bounds_final_kernel, dim3(1), dim3(1024), 0, s, nb, w.bounds);
  hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(256), 0, s, P, points, w.bounds, w.key[0]);
  static const int digits[4] = {8, 8, 8, 8};  // 30-bit codes; 4 passes => result back in buffer 0
  radix_sort_pairs_u32(s, w.key, w.val, P, 4, digits, w.hist, w.bin_total, true);
  hipLaunchKernelGGL(knn_gather_kernel, dim3(nb), dim3(256), 0, s, P, points, w.val[0], w.sorted);
  hipLaunchKernelGGL(knn_box_kernel, dim3((P + KNN_BOX - 1) / KNN_BOX), dim3(256), 0, s, P, w.sorted, w.boxes);
}

hipError_t launch_knn(hipStream_t s, int P, const float* points, void* workspace, float* out) {
  const KnnWork w = carve_knn(workspace, P);
  build_knn_index(s, P, points, w);
  hipLaunchKernelGGL(knn_search_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, w.sorted, w.val[0], w.boxes, out);
  return hipGetLastError();
}

// ---- the Delete path's neighbour query: which query points have a reference point within `thresh` ----
// Reference: GaussianModel.get_near_gaussians_by_mask (gaussiansplatting/scene/gaussian_model.py:865-898) asks
// K_nearest_neighbors(object_xyz, 1, query=in_box_remaining_xyz, return_dist=True) (gaussiansplatting/knn.py: a
// scipy.spatial.KDTree built on the float32 points widened to float64; the 1-NN Euclidean distance comes back as
// float64 and is cast to the points' dtype) and keeps `nn_dist <= dist_thresh`.
// Here: the Morton boxes of the reference set prune in float with a safety margin; every surviving candidate's distance
// is formed in double from the widened coordinates, sqrt in double, rounded to float once -- the same value the KDTree
// path compares.  With `nn_dist` null a query stops at its first hit.
__global__ void __launch_bounds__(256) near_search_kernel(int n_ref, const float4* __restrict__ sorted, const float* __restrict__ boxes,
                                                         int n_query, const float* __restrict__ query, float thresh,
                                                         uint8_t* __restrict__ near, float* __restrict__ nn_dist) {
  const int q = (int)(blockIdx.x * 256 + threadIdx.x);
  if (q >= n_query) return;
  const float4 point = make_float4(query[3 * (size_t)q], query[3 * (size_t)q + 1], query[3 * (size_t)q + 2], 0.f);
  const float prune = thresh * thresh * 1.0001f + FLT_MIN;  // float candidates that the double test could still accept
  const double px = (double)point.x, py = (double)point.y, pz = (double)point.z;
  double best = HUGE_VAL;  // squared, double
  const bool first_hit = nn_dist == nullptr;
  bool hit = false;
  const int nboxes = (n_ref + KNN_BOX - 1) / KNN_BOX;
  for (int b = 0; b < nboxes && !(first_hit && hit); b++) {
    if (!(dist_box_point(boxes + 6 * (size_t)b, point) <= prune)) continue;
    const int e = min(n_ref, (b + 1) * KNN_BOX);
    for (int i = b * KNN_BOX; i < e; i++) {
      const float4 r = sorted[i];
      const float fx = point.x - r.x, fy = point.y - r.y, fz = point.z - r.z;
      if (!(fx * fx + fy * fy + fz * fz <= prune)) continue;
      const double dx = px - (double)r.x, dy = py - (double)r.y, dz = pz - (double)r.z;
      const double d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < best) {
        best = d2;
        hit = (float)sqrt(d2) <= thresh;
        if (first_hit && hit) break;
      }
    }
  }
  const float d = (float)sqrt(best);  // +inf when nothing lay inside the pruning radius
  near[q] = (d <= thresh) ? 1 : 0;
  if (nn_dist) nn_dist[q] = d;
}

hipError_t launch_near_points(hipStream_t s, int n_ref, const float* ref, int n_query, const float* query, float thresh,
                              void* workspace, uint8_t* near, float* nn_dist) {
  const KnnWork w = carve_knn(workspace, n_ref);
  build_knn_index(s, n_ref, ref, w);
  hipLaunchKernelGGL(near_search_kernel, dim3((n_query + 255) / 256), dim3(256), 0, s, n_ref, w.sorted, w.boxes, n_query, query,
                     thresh, near, nn_dist);
  return hipGetLastError();
}

}  // namespace gsr
