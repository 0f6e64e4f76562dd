// gsr_blend.hip -- the per-tile alpha-compositing kernels: K6 (forward), K7 (backward),
// K12 (semantic tracing / apply_weights).
//
// CDNA4 mapping.  The reference runs one 256-thread block per 16x16 tile (8 warps of 32)
// with block-wide barriers around a shared-memory staging buffer.  Here the unit of work is
// ONE WAVE64 = one 8x8 pixel quadrant of a tile, launched as a single-wave workgroup:
//   * no workgroup barrier exists anywhere in these kernels (a one-wave workgroup's
//     s_barrier is free), each quadrant terminates as soon as ITS 64 pixels are opaque,
//     and the CU's wave slots are refilled at wave granularity;
//   * 8x8 is the most compact 64-pixel footprint, so the exec mask stays coherent and the
//     "no lane contributes" early-outs (scalar branches on a 64-bit ballot) fire often;
//   * each wave stages 64 sorted instances at a time in LDS (3 x float4 per instance, read
//     back as uniform-address ds_read_b128 broadcasts);
//   * workgroup ids are remapped so that the 64 waves of a 4x4-tile super-tile are
//     consecutive on ONE XCD (blockIdx % 8): the 4 quadrants of a tile and its neighbours
//     gather the same Gaussians from the same 4 MiB L2, while super-tiles are dealt
//     round-robin to the 8 XCDs for load balance;
//   * the backward reduces the 9 per-Gaussian gradient terms across the 64 lanes with DPP
//     row shifts / row broadcasts (6 v_add_f32_dpp per value) and issues ONE vectorised
//     atomic per (quadrant, instance, term) instead of the reference's one per pixel.
#include "gsr_kernels.h"

namespace gsr {


struct PixelWave {
  int tile, px, py;
  bool inside;
};

// XCD-aware workgroup -> (tile, quadrant) map.  Returns false if this wave has no pixels.
__device__ __forceinline__ bool map_wave(const BlendArgs& a, PixelWave& pw) {
  const uint32_t b = blockIdx.x;
  const uint32_t xcd = b & 7u, k = b >> 3;
  const uint32_t s = xcd + 8u * (k >> 6), w = k & 63u;
  if (s >= (uint32_t)a.NS) return false;
  const uint32_t sx = s % (uint32_t)a.SX, sy = s / (uint32_t)a.SX;
  const uint32_t t = w >> 2, quad = w & 3u;
  const int tx = (int)(sx * 4 + (t & 3u)), ty = (int)(sy * 4 + (t >> 2));
  if (tx >= a.gx || ty >= a.gy) return false;
  const int lane = lane_id();
  pw.tile = ty * a.gx + tx;
  pw.px = tx * TILE + (int)(quad & 1u) * QUAD + (lane & 7);
  pw.py = ty * TILE + (int)(quad >> 1) * QUAD + (lane >> 3);
  pw.inside = pw.px < a.W && pw.py < a.H;
  return __any(pw.inside) != 0;
}

__host__ inline unsigned blend_grid(int gx, int gy, int* SX, int* NS) {
  *SX = (gx + 3) / 4;
  const int SY = (gy + 3) / 4;
  *NS = *SX * SY;
  return 8u * 64u * (unsigned)((*NS + 7) / 8);
}

// ----------------------------------------------------------------------------------
// K6: renderCUDA (forward), DGR/cuda_rasterizer/forward.cu:261-379.
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE) blend_forward_kernel(const BlendArgs a) {
  PixelWave pw;
  if (!map_wave(a, pw)) return;
  const int lane = lane_id();
  const uint2 range = a.ranges[pw.tile];
  const float pfx = (float)pw.px, pfy = (float)pw.py;
  bool done = !pw.inside;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
  uint32_t last_contributor = 0;

  __shared__ float4 s0[WAVE], s1[WAVE], s2[WAVE];
  for (uint32_t base = range.x; base < range.y; base += WAVE) {
    if (__all(done)) break;
    const uint32_t n = min((uint32_t)WAVE, range.y - base);
    __syncthreads();
    if ((uint32_t)lane < n) {
      const uint32_t id = a.point_list[base + lane];
      s0[lane] = a.rec0[id];
      s1[lane] = a.rec1[id];
      s2[lane] = a.rec2[id];
    }
    __syncthreads();
    const uint32_t cbase = base - range.x;
    for (uint32_t j = 0; !done && j < n; ++j) {
      const float4 g = s1[j];
      const float4 co = s0[j];
      const float dx = g.x - pfx, dy = g.y - pfy;
      const float power = blend_power(co.x, co.y, co.z, dx, dy);
      if (power > 0.0f) continue;
      const float alpha = fminf(0.99f, co.w * gsr_expf(power));
      if (alpha < 1.0f / 255.0f) continue;
      const float test_T = T * (1.0f - alpha);
      if (test_T < 0.0001f) {
        done = true;
        continue;
      }
      const float4 col = s2[j];
      const float w = alpha * T;
      C0 = __builtin_fmaf(col.x, w, C0);
      C1 = __builtin_fmaf(col.y, w, C1);
      C2 = __builtin_fmaf(col.z, w, C2);
      D = __builtin_fmaf(g.z, w, D);
      T = test_T;
      last_contributor = cbase + j + 1;
    }
  }
  if (pw.inside) {
    const size_t pix = (size_t)pw.py * a.W + pw.px, HW = (size_t)a.H * a.W;
    a.final_T[pix] = T;
    a.n_contrib[pix] = last_contributor;
    a.out_color[pix] = __builtin_fmaf(T, a.bg[0], C0);
    a.out_color[HW + pix] = __builtin_fmaf(T, a.bg[1], C1);
    a.out_color[2 * HW + pix] = __builtin_fmaf(T, a.bg[2], C2);
    a.out_depth[pix] = D;
  }
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
  return v;
}

// ----------------------------------------------------------------------------------
// K7: renderCUDA (backward), DGR/cuda_rasterizer/backward.cu:399-557.
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE) blend_backward_kernel(const BlendArgs a) {
  PixelWave pw;
  if (!map_wave(a, pw)) return;
  const int lane = lane_id();
  const uint2 range = a.ranges[pw.tile];
  if (range.y <= range.x) return;
  const float pfx = (float)pw.px, pfy = (float)pw.py;
  const size_t pix = (size_t)pw.py * a.W + pw.px, HW = (size_t)a.H * a.W;

  const float T_final = pw.inside ? a.final_T[pix] : 0.f;
  float T = T_final;
  const uint32_t last_contributor = pw.inside ? a.n_contrib[pix] : 0u;
  const uint32_t maxc = wave_max_u32(last_contributor);
  if (maxc == 0) return;

  float dpx[3] = {0.f, 0.f, 0.f};
  if (pw.inside) {
    dpx[0] = a.dL_dpix[pix];
    dpx[1] = a.dL_dpix[HW + pix];
    dpx[2] = a.dL_dpix[2 * HW + pix];
  }
  float bg_dot_dpixel = 0.f;
#pragma unroll
  for (int i = 0; i < 3; i++) bg_dot_dpixel += a.bg[i] * dpx[i];
  const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;

  float accum_rec[3] = {0.f, 0.f, 0.f}, last_color[3] = {0.f, 0.f, 0.f};
  float last_alpha = 0.f;

  __shared__ float4 s0[WAVE], s1[WAVE], s2[WAVE];
  __shared__ uint32_t sid[WAVE];
  __shared__ float sacc[9][WAVE];

  // back to front over positions [0, maxc) of the tile's list, 64 at a time
  for (uint32_t end = maxc; end > 0; end -= min(end, (uint32_t)WAVE)) {
    const uint32_t n = min((uint32_t)WAVE, end);
    __syncthreads();
    if ((uint32_t)lane < n) {
      const uint32_t id = a.point_list[range.x + end - 1 - lane];
      sid[lane] = id;
      s0[lane] = a.rec0[id];
      s1[lane] = a.rec1[id];
      s2[lane] = a.rec2[id];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) sacc[k][lane] = 0.f;
    __syncthreads();

    for (uint32_t j = 0; j < n; ++j) {
      const uint32_t c = end - 1 - j;  // 0-based position of this instance in the tile list
      const float4 g = s1[j];
      const float4 co = s0[j];
      const float dx = g.x - pfx, dy = g.y - pfy;
      const float power = blend_power(co.x, co.y, co.z, dx, dy);
      bool contrib = (c < last_contributor) && !(power > 0.0f);
      float G = 0.f, alpha = 0.f;
      if (contrib) {
        G = gsr_expf(power);
        alpha = fminf(0.99f, co.w * G);
        contrib = !(alpha < 1.0f / 255.0f);
      }
      if (!__any(contrib)) continue;  // wave-uniform

      float v[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) v[k] = 0.f;
      if (contrib) {
        const float4 col = s2[j];
        const float one_m = 1.f - alpha;
        T = T * __builtin_amdgcn_rcpf(one_m);  // T / (1 - alpha), backward.cu:503
        const float dchannel_dcolor = alpha * T;
        float dL_dalpha = 0.0f;
        const float cc[3] = {col.x, col.y, col.z};
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
          accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
          last_color[ch] = cc[ch];
          dL_dalpha += (cc[ch] - accum_rec[ch]) * dpx[ch];
          v[6 + ch] = dchannel_dcolor * dpx[ch];
        }
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-T_final * __builtin_amdgcn_rcpf(one_m)) * bg_dot_dpixel;
        const float dL_dG = co.w * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        const float dG_ddelx = -gdx * co.x - gdy * co.y;
        const float dG_ddely = -gdy * co.z - gdx * co.y;
        v[0] = dL_dG * dG_ddelx * ddelx_dx;
        v[1] = dL_dG * dG_ddely * ddely_dy;
        v[2] = -0.5f * gdx * dx * dL_dG;
        v[3] = -0.5f * gdx * dy * dL_dG;
        v[4] = -0.5f * gdy * dy * dL_dG;
        v[5] = G * dL_dalpha;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) v[k] = wave_sum_to_lane63(v[k]);
      if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 9; ++k) sacc[k][j] = v[k];
      }
    }
    __syncthreads();
    if ((uint32_t)lane < n) {
      float r[9];
      bool any = false;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        r[k] = sacc[k][lane];
        any |= (r[k] != 0.f);
      }
      if (any) {
        const size_t id = sid[lane];
        unsafeAtomicAdd(&a.dL_dmean2D[3 * id + 0], r[0]);
        unsafeAtomicAdd(&a.dL_dmean2D[3 * id + 1], r[1]);
        unsafeAtomicAdd(&a.dL_dconic[4 * id + 0], r[2]);
        unsafeAtomicAdd(&a.dL_dconic[4 * id + 1], r[3]);
        unsafeAtomicAdd(&a.dL_dconic[4 * id + 3], r[4]);
        unsafeAtomicAdd(&a.dL_dopacity[id], r[5]);
        unsafeAtomicAdd(&a.dL_dcolors[3 * id + 0], r[6]);
        unsafeAtomicAdd(&a.dL_dcolors[3 * id + 1], r[7]);
        unsafeAtomicAdd(&a.dL_dcolors[3 * id + 2], r[8]);
      }
    }
  }
}

// ----------------------------------------------------------------------------------
// K12: renderCUDA_apply_weights, DGR/cuda_rasterizer/apply_weights.cu:239-356.
// Same traversal / termination as K6; every blended (pixel, instance) adds the pixel's
// mask value(s) to weights[id] and C to cnt[id] (the reference increments cnt inside its
// channel loop, apply_weights.cu:331-339).
// ----------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(WAVE) trace_weights_kernel(const BlendArgs a) {
  PixelWave pw;
  if (!map_wave(a, pw)) return;
  const int lane = lane_id();
  const uint2 range = a.ranges[pw.tile];
  if (range.y <= range.x) return;
  const float pfx = (float)pw.px, pfy = (float)pw.py;
  const size_t pix = (size_t)pw.py * a.W + pw.px, HW = (size_t)a.H * a.W;
  float Cw[C];
#pragma unroll
  for (int ch = 0; ch < C; ++ch) Cw[ch] = pw.inside ? a.image_weights[(size_t)ch * HW + pix] : 0.f;
  bool done = !pw.inside;
  float T = 1.0f;

  __shared__ float4 s0[WAVE], s1[WAVE];
  __shared__ uint32_t sid[WAVE];
  __shared__ float sacc[C][WAVE];
  __shared__ int scnt[WAVE];
  for (uint32_t base = range.x; base < range.y; base += WAVE) {
    if (__all(done)) break;
    const uint32_t n = min((uint32_t)WAVE, range.y - base);
    __syncthreads();
    if ((uint32_t)lane < n) {
      const uint32_t id = a.point_list[base + lane];
      sid[lane] = id;
      s0[lane] = a.rec0[id];
      s1[lane] = a.rec1[id];
    }
#pragma unroll
    for (int ch = 0; ch < C; ++ch) sacc[ch][lane] = 0.f;
    scnt[lane] = 0;
    __syncthreads();
    for (uint32_t j = 0; j < n; ++j) {
      if (__all(done)) break;
      const float4 g = s1[j];
      const float4 co = s0[j];
      const float dx = g.x - pfx, dy = g.y - pfy;
      const float power = blend_power(co.x, co.y, co.z, dx, dy);
      bool hit = !done && !(power > 0.0f);
      float test_T = T;
      if (hit) {
        const float alpha = fminf(0.99f, co.w * gsr_expf(power));
        hit = !(alpha < 1.0f / 255.0f);
        if (hit) {
          test_T = T * (1.0f - alpha);
          if (test_T < 0.0001f) {
            done = true;
            hit = false;
          }
        }
      }
      const uint64_t m = __ballot(hit);
      if (m == 0) continue;
      if (hit) T = test_T;
      float v[C];
#pragma unroll
      for (int ch = 0; ch < C; ++ch) v[ch] = wave_sum_to_lane63(hit ? Cw[ch] : 0.f);
      if (lane == 63) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) sacc[ch][j] = v[ch];
        scnt[j] = (int)__popcll(m) * C;
      }
    }
    __syncthreads();
    if ((uint32_t)lane < n) {
      const int cn = scnt[lane];
      if (cn != 0) {
        const size_t id = sid[lane];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) unsafeAtomicAdd(&a.weights[id * C + ch], sacc[ch][lane]);
        atomicAdd(&a.cnt[id], cn);
      }
    }
  }
}

hipError_t launch_blend_forward(hipStream_t s, BlendArgs a) {
  const unsigned grid = blend_grid(a.gx, a.gy, &a.SX, &a.NS);
  hipLaunchKernelGGL(blend_forward_kernel, dim3(grid), dim3(WAVE), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_blend_backward(hipStream_t s, BlendArgs a) {
  const unsigned grid = blend_grid(a.gx, a.gy, &a.SX, &a.NS);
  hipLaunchKernelGGL(blend_backward_kernel, dim3(grid), dim3(WAVE), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_trace_weights(hipStream_t s, BlendArgs a) {
  const unsigned grid = blend_grid(a.gx, a.gy, &a.SX, &a.NS);
  switch (a.C) {
    case 1: hipLaunchKernelGGL(trace_weights_kernel<1>, dim3(grid), dim3(WAVE), 0, s, a); break;
    case 2: hipLaunchKernelGGL(trace_weights_kernel<2>, dim3(grid), dim3(WAVE), 0, s, a); break;
    case 3: hipLaunchKernelGGL(trace_weights_kernel<3>, dim3(grid), dim3(WAVE), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace gsr
