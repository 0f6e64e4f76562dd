// gsr_preprocess.hip -- per-Gaussian kernels: K1 (forward preprocess, incl. num_rendered), K8+K9 fused
// (backward preprocess), K10 (frustum mark).  K2 (the tile-count prefix) lives in gsr_binning.hip.
//
// All of them are HBM-streaming kernels: one thread per Gaussian, 256 Gaussians per
// block, every input byte read once and every output byte written once.  Arithmetic
// follows the reference's statement order with no FMA contraction so that the
// integer outputs (radii, tiles_touched) are bit-identical to the CPU oracle.
#include "gsr_kernels.h"

#ifndef GSR_K1_PREFETCH
#define GSR_K1_PREFETCH 1  // (A/B builds: 0 = the loads where the reference has them)
#endif

namespace gsr {

// glm::mat3 semantics (column-major m[col][row]; product evaluated left to right),
// DGR/third_party/glm/glm/detail/type_mat3x3.inl:486-519.
struct M3 {
  float m[3][3];
};
__device__ __forceinline__ M3 mk(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
  M3 r;
  r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
  r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
  r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
  return r;
}
__device__ __forceinline__ M3 mul(const M3& A, const M3& B) {
  M3 R;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
  return R;
}
__device__ __forceinline__ M3 tr(const M3& A) {
  M3 R;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) R.m[c][r] = A.m[r][c];
  return R;
}

__device__ __forceinline__ int f2i_sat(float v) {
  if (!(v == v)) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int)v;
}

struct Cam {
  float view[16];
  float proj[16];
  float campos[3];
};

__device__ __forceinline__ void load_cam(Cam& c, const float* view, const float* proj, const float* campos) {
  // 35 uniform floats: the compiler turns these into scalar (s_load) loads.
#pragma unroll
  for (int i = 0; i < 16; ++i) c.view[i] = view[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) c.proj[i] = proj[i];
  if (campos) {
#pragma unroll
    for (int i = 0; i < 3; ++i) c.campos[i] = campos[i];
  }
}

__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy, uint32_t& minx,
                                         uint32_t& miny, uint32_t& maxx, uint32_t& maxy) {
  // DGR/cuda_rasterizer/auxiliary.h:46-56
  const float r = (float)max_radius;
  minx = (uint32_t)min(gx, max(0, f2i_sat((px - r) / (float)TILE)));
  miny = (uint32_t)min(gy, max(0, f2i_sat((py - r) / (float)TILE)));
  maxx = (uint32_t)min(gx, max(0, f2i_sat((px + r + (float)TILE - 1.0f) / (float)TILE)));
  maxy = (uint32_t)min(gy, max(0, f2i_sat((py + r + (float)TILE - 1.0f) / (float)TILE)));
}

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f,  -0.4570457994644658f,
                                       0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Loads the SH coefficients of one Gaussian into registers.  M == 16 (the standard 3DGS ply) is a
// 192-byte, 16-byte-aligned record read with dwordx4 loads.  All loads of a degree are issued
// UNCONDITIONALLY back to back (the switch is on the wave-uniform active degree): a per-load guard
// makes hipcc wait for each load before the next branch, i.e. one memory round trip per load.
template <int NV>
__device__ __forceinline__ void load_sh_vec(const float4* __restrict__ q, V3 (&sh)[16]) {
  float f[48];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    if (i < NV) {
      const float4 v = q[i];  // (a nontemporal load here: K1 +54 us, K8+K9 +80 us -- measured, profiles/r03_b)
      f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
    } else {
      f[4 * i] = f[4 * i + 1] = f[4 * i + 2] = f[4 * i + 3] = 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) sh[k] = {f[3 * k], f[3 * k + 1], f[3 * k + 2]};
}

__device__ __forceinline__ void load_sh(const float* __restrict__ shs, size_t idx, int M, int D, V3 (&sh)[16]) {
  const float* p = shs + idx * (size_t)M * 3;
  if (M == 16) {
    const float4* q = reinterpret_cast<const float4*>(p);
    switch (D) {  // floats needed: 3 (D+1)^2 = 3, 12, 27, 48
      case 0: load_sh_vec<1>(q, sh); break;
      case 1: load_sh_vec<3>(q, sh); break;
      case 2: load_sh_vec<7>(q, sh); break;
      default: load_sh_vec<12>(q, sh); break;
    }
  } else {
    // generic record length: clamp the index instead of guarding the load (same reason)
    const int ncoef = (D + 1) * (D + 1);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int kk = k < ncoef ? k : ncoef - 1;
      const V3 v = {p[3 * kk], p[3 * kk + 1], p[3 * kk + 2]};
      sh[k] = k < ncoef ? v : V3{0.f, 0.f, 0.f};
    }
  }
}

// computeCov3D, forward.cu:118-152: Sigma = (S R)^T (S R), upper triangle.  Used by K1 and again by K8+K9 -- the reference
// keeps the six floats in its geometry buffer between the passes (rasterizer_impl.cu:225, 388); recomputing them from
// the 28 bytes of scale and rotation the backward reads anyway saves a 24-byte store and a 24-byte load per Gaussian.
__device__ __forceinline__ void cov3d_from_values(float s0, float s1, float s2, float scale_modifier, const float4& q,
                                                  float (&c3)[6]) {
  M3 S = mk(1, 0, 0, 0, 1, 0, 0, 0, 1);
  S.m[0][0] = scale_modifier * s0;
  S.m[1][1] = scale_modifier * s1;
  S.m[2][2] = scale_modifier * s2;
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  const M3 R = mk(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                  2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                  2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
  const M3 Mm = mul(S, R);
  const M3 Sigma = mul(tr(Mm), Mm);
  c3[0] = Sigma.m[0][0]; c3[1] = Sigma.m[0][1]; c3[2] = Sigma.m[0][2];
  c3[3] = Sigma.m[1][1]; c3[4] = Sigma.m[1][2]; c3[5] = Sigma.m[2][2];
}
__device__ __forceinline__ void cov3d_from_scale_rot(const float* __restrict__ scales, float scale_modifier,
                                                     const float* __restrict__ rotations, int idx, float (&c3)[6]) {
  cov3d_from_values(scales[3 * idx + 0], scales[3 * idx + 1], scales[3 * idx + 2], scale_modifier,
                    reinterpret_cast<const float4*>(rotations)[idx], c3);
}

// d(RGB)/d(dir) of computeColorFromSH, backward.cu:88-136 (the part of its backward that needs the SH coefficients):
// evaluated by K1, which holds them, and handed to K8+K9 through the geometry scratch.
__device__ __forceinline__ void sh_color_dir_derivatives(int D, float x, float y, float z, const V3 (&sh)[16], V3& dRGBdx,
                                                         V3& dRGBdy, V3& dRGBdz) {
  dRGBdx = {0, 0, 0};
  dRGBdy = {0, 0, 0};
  dRGBdz = {0, 0, 0};
  if (D > 0) {
    dRGBdx = (-SH_C1) * sh[3];
    dRGBdy = (-SH_C1) * sh[1];
    dRGBdz = SH_C1 * sh[2];
    if (D > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      dRGBdx = dRGBdx + ((SH_C2[0] * y) * sh[4] + (SH_C2[2] * 2.f * -x) * sh[6] + (SH_C2[3] * z) * sh[7] +
                         (SH_C2[4] * 2.f * x) * sh[8]);
      dRGBdy = dRGBdy + ((SH_C2[0] * x) * sh[4] + (SH_C2[1] * z) * sh[5] + (SH_C2[2] * 2.f * -y) * sh[6] +
                         (SH_C2[4] * 2.f * -y) * sh[8]);
      dRGBdz = dRGBdz + ((SH_C2[1] * y) * sh[5] + (SH_C2[2] * 2.f * 2.f * z) * sh[6] + (SH_C2[3] * x) * sh[7]);
      if (D > 2) {
        // `SH_C3[k] * sh[n] * s1 * s2` parses as (((SH_C3[k]*sh[n])*s1)*s2)
        dRGBdx = dRGBdx + ((SH_C3[0] * sh[9]) * 3.f * 2.f * xy + (SH_C3[1] * sh[10]) * yz +
                           (SH_C3[2] * sh[11]) * -2.f * xy + (SH_C3[3] * sh[12]) * -3.f * 2.f * xz +
                           (SH_C3[4] * sh[13]) * (-3.f * xx + 4.f * zz - yy) + (SH_C3[5] * sh[14]) * 2.f * xz +
                           (SH_C3[6] * sh[15]) * 3.f * (xx - yy));
        dRGBdy = dRGBdy + ((SH_C3[0] * sh[9]) * 3.f * (xx - yy) + (SH_C3[1] * sh[10]) * xz +
                           (SH_C3[2] * sh[11]) * (-3.f * yy + 4.f * zz - xx) +
                           (SH_C3[3] * sh[12]) * -3.f * 2.f * yz + (SH_C3[4] * sh[13]) * -2.f * xy +
                           (SH_C3[5] * sh[14]) * -2.f * yz + (SH_C3[6] * sh[15]) * -3.f * 2.f * xy);
        dRGBdz = dRGBdz + ((SH_C3[1] * sh[10]) * xy + (SH_C3[2] * sh[11]) * 4.f * 2.f * yz +
                           (SH_C3[3] * sh[12]) * 3.f * (2.f * zz - xx - yy) +
                           (SH_C3[4] * sh[13]) * 4.f * 2.f * xz + (SH_C3[5] * sh[14]) * (xx - yy));
      }
    }
  }
}

// ----------------------------------------------------------------------------------
// K1: preprocessCUDA, DGR/cuda_rasterizer/forward.cu:155-256 (+ apply_weights.cu:148-234).
// Additionally produces the per-block sum of tiles_touched (first level of K2).
// ----------------------------------------------------------------------------------

// (70 VGPRs would allow 7 waves per SIMD; with one 192-byte SH row per thread that many waves thrash the caches --
//  tools/microbench/rows192.hip: 4.55 TB/s at 2-4 waves per SIMD, 4.16 at 6, 3.24 at 8 -- so the kernel is held at 4: -3 us.
//  Round 4 measured the alternative K8+K9 uses for its stores -- the wave's 64 records loaded cooperatively, 12 coalesced
//  1 KB loads into an LDS tile, every lane then reading its row -- on the same box: preprocess stage 139 -> 161 us at 1 M
//  Gaussians, 602 -> 703 us at 6 M (the 53 KB of tiles leave three blocks per CU, and the rows cross LDS on top of the
//  memory round trip): commit 4aadae3 holds it, profiles/r04_b_k1_and_chain.md the table.)
// MAIN: the training configuration -- covariance from scales / rotations, colours from SH records of 16 coefficients at degree
// 3 -- as an instantiation of its own: with the other input modes compiled out there is no second definition of any register
// at a join, and the waits hipcc inserts sit in front of the first USE of the early SH request instead of right behind it.
template <bool MAIN>
__global__ void __launch_bounds__(GAUSS_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) preprocess_kernel(const PreArgs a) {
  const float* const cov3D_precomp = MAIN ? nullptr : a.cov3D_precomp;
  const float* const colors_precomp = MAIN ? nullptr : a.colors_precomp;
  const bool skip_color = MAIN ? false : (a.skip_color != 0);
  const int D = MAIN ? 3 : a.D, M = MAIN ? 16 : a.M;
  __shared__ uint32_t smem[GAUSS_BLOCK / 64 + 1], ssum[GAUSS_BLOCK / 64];
  __shared__ uint32_t skmax[2];
  if (threadIdx.x < 2) skmax[threadIdx.x] = 0u;
  const int idx = (int)(blockIdx.x * GAUSS_BLOCK + threadIdx.x);
  uint32_t my_tiles = 0, my_groups = 0;  // tiles touched; 8x8-tile groups touched (gsr_binning.hip: group instances)
  uint32_t kmax = 0u, kinv = 0u;  // max of the depth key / of its complement over the visible Gaussians of the wave
  Cam cam;
  load_cam(cam, a.viewmatrix, a.projmatrix, skip_color || colors_precomp ? nullptr : a.campos);
  const float* view = cam.view;
  const float* proj = cam.proj;
  // ---- geometry (forward.cu:182-237): everything up to the colour.  `live` = the Gaussian survives every cull.
  bool live = false;
  int my_radius_i = 0;
  float my_depth = 0.f, my_radius = 0.f, conx = 0.f, cony = 0.f, conz = 0.f, pix = 0.f, piy = 0.f;
  V3 p = {0.f, 0.f, 0.f};
  uint2 my_rect = make_uint2(0u, 0u);  // tile rectangle (origin, width | height << 16); width * height == tiles_touched
  // Loads: K1 is a streaming kernel at 4 waves per SIMD, so every DEPENDENT memory round trip of a wave shows.  Position, scale,
  // rotation and opacity are fetched together, unconditionally (28 wasted bytes for a Gaussian behind the camera); the SH record
  // -- 192 of the ~300 bytes -- is requested as soon as the projected centre says the Gaussian is probably on screen, in front
  // of the ~800 instructions of covariance arithmetic instead of behind them (a Gaussian that turns out live without the early
  // request fetches it where it always did; one that is culled after it wasted the fetch: the 1.5x screen margin keeps both rare).
  V3 sh[16];
  bool sh_loaded = false;
  const bool want_sh = GSR_K1_PREFETCH && !skip_color && colors_precomp == nullptr;
  float sc0 = 0.f, sc1 = 0.f, sc2 = 0.f, my_opacity = 0.f;
  float4 quat = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < a.P) {
    if (GSR_K1_PREFETCH) {
      if (cov3D_precomp == nullptr) {
        sc0 = a.scales[3 * idx + 0];
        sc1 = a.scales[3 * idx + 1];
        sc2 = a.scales[3 * idx + 2];
        quat = reinterpret_cast<const float4*>(a.rotations)[idx];
      }
      my_opacity = a.opacities[idx];
    }
    do {
      p = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
      // in_frustum, auxiliary.h:139-164: only the near test survives
      const V3 p_view = {view[0] * p.x + view[4] * p.y + view[8] * p.z + view[12],
                         view[1] * p.x + view[5] * p.y + view[9] * p.z + view[13],
                         view[2] * p.x + view[6] * p.y + view[10] * p.z + view[14]};
      if (p_view.z <= 0.2f) break;
      const float hx = proj[0] * p.x + proj[4] * p.y + proj[8] * p.z + proj[12];
      const float hy = proj[1] * p.x + proj[5] * p.y + proj[9] * p.z + proj[13];
      const float hw = proj[3] * p.x + proj[7] * p.y + proj[11] * p.z + proj[15];
      const float p_w = 1.0f / (hw + 0.0000001f);
      const float projx = hx * p_w, projy = hy * p_w;
      if (want_sh) {
        // (no branch on `likely` around the loads: behind a divergent region hipcc waits for everything in flight.  The other
        //  lanes re-read the block's first record -- one row for all of them, from the cache)
        const bool likely = fabsf(projx) < 1.5f && fabsf(projy) < 1.5f;
        load_sh(a.shs, likely ? (size_t)idx : (size_t)blockIdx.x * GAUSS_BLOCK, M, D, sh);
        sh_loaded = likely;
      }

      // cov3D: forward.cu:118-152, or the precomputed one
      float c3[6];
      if (cov3D_precomp != nullptr) {
#pragma unroll
        for (int i = 0; i < 6; ++i) c3[i] = cov3D_precomp[6 * (size_t)idx + i];
      } else {
        if (GSR_K1_PREFETCH) cov3d_from_values(sc0, sc1, sc2, a.scale_modifier, quat, c3);
        else cov3d_from_scale_rot(a.scales, a.scale_modifier, a.rotations, idx, c3);
      }
      // (not stored: the backward recomputes it from the same inputs, bit for bit, instead of reading 24 B back)

      // cov2D: forward.cu:74-113
      V3 t = p_view;
      const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
      const float txtz = t.x / t.z, tytz = t.y / t.z;
      t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
      t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
      const M3 J = mk(a.focal_x / t.z, 0.0f, -(a.focal_x * t.x) / (t.z * t.z), 0.0f, a.focal_y / t.z,
                      -(a.focal_y * t.y) / (t.z * t.z), 0, 0, 0);
      const M3 Wm = mk(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
      const M3 T = mul(Wm, J);
      const M3 Vrk = mk(c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]);
      M3 cov = mul(mul(tr(T), tr(Vrk)), T);
      const float cx = cov.m[0][0] + 0.3f, cy = cov.m[0][1], cz = cov.m[1][1] + 0.3f;

      // forward.cu:219-237
      const float det = (cx * cz - cy * cy);
      if (det == 0.0f) break;
      const float det_inv = 1.f / det;
      conx = cz * det_inv;
      cony = -cy * det_inv;
      conz = cx * det_inv;
      const float mid = 0.5f * (cx + cz);
      const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
      const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
      my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
      // ndc2Pix in double, auxiliary.h:41-44
      pix = (float)((((double)projx + 1.0) * (double)a.W - 1.0) * 0.5);
      piy = (float)((((double)projy + 1.0) * (double)a.H - 1.0) * 0.5);
      uint32_t minx, miny, maxx, maxy;
      const int radius_i = f2i_sat(my_radius);
      get_rect(pix, piy, radius_i, a.gx, a.gy, minx, miny, maxx, maxy);
      const uint32_t area = (maxx - minx) * (maxy - miny);
      if (area == 0) break;
      // A NaN radius (non-finite scale / rotation / covariance) converts to 0 and still spans one tile.  The reference
      // then counts that tile in num_rendered but never writes the instance (duplicateWithKeys skips radii <= 0,
      // rasterizer_impl.cu:93), so its sort reads an uninitialised key.  Here such a Gaussian touches no tile.
      if (radius_i <= 0) break;
      uint32_t ntiles = area;
      if (a.tile_bounds) {
        // Opt-in (GSR_FLAG_TILE_BOUNDS_ALPHA): bin the Gaussian only into the tiles its alpha >= 1/255 level set
        // can reach -- the axis-aligned bounding box of that ellipse, with the margins of the blend kernels' own cull test
        // (gsr_blend.hip: can_touch_quad), intersected with the reference's square.  An instance dropped here passes
        // `alpha < 1/255 -> continue` (forward.cu:340-344) at every pixel of its tile, so images, radii and gradients
        // do not change; num_rendered, the lists and n_contrib do.  radii keeps the reference's value.
        const float o = GSR_K1_PREFETCH ? my_opacity : a.opacities[idx];
        float hx = (float)radius_i, hy = (float)radius_i;
        if (o < 1.0f / 255.0f) {
          ntiles = 0;  // never reaches the threshold (a NaN opacity fails this test and keeps the reference rectangle)
        } else if (o == o) {
          const float xz = conx * conz;
          const float detc = xz - cony * cony;
          if (detc > 0.0f && detc >= 1e-3f * xz) {  // (an ill-conditioned conic keeps the reference rectangle: can_touch_quad)
            const float tau2 = 2.0f * (__logf(255.0f * o) * 1.001f + 0.01f);
            const float inv = __builtin_amdgcn_rcpf(detc) * 1.001f;
            const float ex = __builtin_sqrtf(tau2 * conz * inv) + 0.01f, ey = __builtin_sqrtf(tau2 * conx * inv) + 0.01f;
            if (ex == ex && ey == ey) {
              hx = fminf(hx, ex);
              hy = fminf(hy, ey);
            }
          }
        }
        if (ntiles != 0) {
          // tile t (pixels 16 t .. 16 t + 15) can be reached iff 16 t <= p + h and 16 t + 15 >= p - h.  (The reference's
          // own upper bound, floor((p + r + 15) / 16), stops one pixel short of p + r; hence "+ TILE" here and the
          // intersection with the reference's rectangle, which keeps the result a subset of it.)
          minx = max(minx, (uint32_t)min(a.gx, max(0, f2i_sat((pix - hx) / (float)TILE))));
          miny = max(miny, (uint32_t)min(a.gy, max(0, f2i_sat((piy - hy) / (float)TILE))));
          maxx = min(maxx, (uint32_t)min(a.gx, max(0, f2i_sat((pix + hx + (float)TILE) / (float)TILE))));
          maxy = min(maxy, (uint32_t)min(a.gy, max(0, f2i_sat((piy + hy + (float)TILE) / (float)TILE))));
          ntiles = (maxx > minx && maxy > miny) ? (maxx - minx) * (maxy - miny) : 0u;
        }
      }
      if (ntiles != 0) {
        my_rect = make_uint2(minx | (miny << 16), (maxx - minx) | ((maxy - miny) << 16));
        my_groups = (((maxx - 1u) >> GROUP_SHIFT) - (minx >> GROUP_SHIFT) + 1u) * (((maxy - 1u) >> GROUP_SHIFT) - (miny >> GROUP_SHIFT) + 1u);
      }
      my_radius_i = radius_i;
      my_depth = p_view.z;
      my_tiles = ntiles;
      live = true;
    } while (false);
  }
  // ---- colour: forward.cu:20-71, or a copy of colors_precomp into the gather record
  if (live) {
      float4 col = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!skip_color) {
        if (colors_precomp != nullptr) {
          col.x = colors_precomp[3 * (size_t)idx];
          col.y = colors_precomp[3 * (size_t)idx + 1];
          col.z = colors_precomp[3 * (size_t)idx + 2];
        } else {
          V3 dir = {p.x - cam.campos[0], p.y - cam.campos[1], p.z - cam.campos[2]};
          const float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
          dir = {dir.x / len, dir.y / len, dir.z / len};
          if (!sh_loaded) load_sh(a.shs, (size_t)idx, M, D, sh);
          V3 result = SH_C0 * sh[0];
          if (D > 0) {
            const float x = dir.x, y = dir.y, z = dir.z;
            result = result - (SH_C1 * y) * sh[1] + (SH_C1 * z) * sh[2] - (SH_C1 * x) * sh[3];
            if (D > 1) {
              const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
              result = result + (SH_C2[0] * xy) * sh[4] + (SH_C2[1] * yz) * sh[5] +
                       (SH_C2[2] * (2.0f * zz - xx - yy)) * sh[6] + (SH_C2[3] * xz) * sh[7] +
                       (SH_C2[4] * (xx - yy)) * sh[8];
              if (D > 2) {
                result = result + (SH_C3[0] * y * (3.0f * xx - yy)) * sh[9] + (SH_C3[1] * xy * z) * sh[10] +
                         (SH_C3[2] * y * (4.0f * zz - xx - yy)) * sh[11] +
                         (SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * sh[12] +
                         (SH_C3[4] * x * (4.0f * zz - xx - yy)) * sh[13] + (SH_C3[5] * z * (xx - yy)) * sh[14] +
                         (SH_C3[6] * x * (xx - 3.0f * yy)) * sh[15];
              }
            }
          }
          if (!a.forward_only && D > 0) {  // what the backward needs of the SH record (see Geom::dcol)
            V3 ddx, ddy, ddz;
            sh_color_dir_derivatives(D, dir.x, dir.y, dir.z, sh, ddx, ddy, ddz);
            *reinterpret_cast<V3*>(a.g.dcol[0] + 3 * (size_t)idx) = ddx;  // (12-byte records: one dwordx3 store each)
            *reinterpret_cast<V3*>(a.g.dcol[1] + 3 * (size_t)idx) = ddy;
            *reinterpret_cast<V3*>(a.g.dcol[2] + 3 * (size_t)idx) = ddz;
          }
          result = {result.x + 0.5f, result.y + 0.5f, result.z + 0.5f};
          a.g.clamped[idx] = (uint8_t)((result.x < 0 ? 1 : 0) | (result.y < 0 ? 2 : 0) | (result.z < 0 ? 4 : 0));
          // glm::max(result, 0.0f) is (x < y) ? y : x (forward.cu:70): a NaN colour stays NaN, unlike fmaxf
          col.x = result.x < 0.0f ? 0.0f : result.x;
          col.y = result.y < 0.0f ? 0.0f : result.y;
          col.z = result.z < 0.0f ? 0.0f : result.z;
        }
      }
      // forward.cu:250-255
      a.g.rec0[idx] = make_float4(conx, cony, conz, GSR_K1_PREFETCH ? my_opacity : a.opacities[idx]);
      a.g.rec1[idx] = make_float4(pix, piy, my_depth, my_radius);
      a.g.rec2[idx] = col;
  }
  if (idx < a.P) {
    a.radii[idx] = my_radius_i;
    a.g.rect[idx] = my_tiles ? my_rect : make_uint2(0u, 0u);  // written for every Gaussian: the binning reads nothing else of it
    // key of the depth ordering (gsr_binning.hip): depth bits, culled Gaussians after every live one
    a.g.dkey[0][idx] = my_tiles ? __float_as_uint(my_depth) : 0xffffffffu;
    if (my_tiles) {
      kmax = __float_as_uint(my_depth);
      kinv = ~kmax;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d, 64));
    kinv = max(kinv, (uint32_t)__shfl_xor((int)kinv, d, 64));
  }
  // num_rendered = sum of tiles_touched; the number of group instances = sum of the groups touched
  uint32_t tsum = my_tiles, gsum = my_groups;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    tsum += (uint32_t)__shfl_xor((int)tsum, d, 64);
    gsum += (uint32_t)__shfl_xor((int)gsum, d, 64);
  }
  if (lane_id() == 0) {
    smem[threadIdx.x >> 6] = tsum;
    ssum[threadIdx.x >> 6] = gsum;
  }
  __syncthreads();  // (also orders the skmax init)
  if (lane_id() == 0 && kinv) {
    atomicMax(&skmax[0], kmax);
    atomicMax(&skmax[1], kinv);
  }
  __syncthreads();
  // this block's share of num_rendered, of the group instances and of the range of the depth keys (max, and max of the
  // complement = ~min: the host sizes the depth sort with it); reduced by the first depth-sort kernel (sort_hist_kernel, publish)
  if (threadIdx.x == 0) {
    uint32_t total = 0, gtotal = 0;
#pragma unroll
    for (int w = 0; w < GAUSS_BLOCK / 64; ++w) {
      total += smem[w];
      gtotal += ssum[w];
    }
    a.g.k1_partials[blockIdx.x] = make_uint4(total, skmax[0], skmax[1], gtotal);
  }
}

// ----------------------------------------------------------------------------------
// K10: checkFrustum, rasterizer_impl.cu:53-63.
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(GAUSS_BLOCK) mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                                  const float* __restrict__ view,
                                                                  uint8_t* __restrict__ present) {
  const int idx = (int)(blockIdx.x * GAUSS_BLOCK + threadIdx.x);
  if (idx >= P) return;
  const float x = means3D[3 * idx], y = means3D[3 * idx + 1], z = means3D[3 * idx + 2];
  const float vz = view[2] * x + view[6] * y + view[10] * z + view[14];
  present[idx] = vz > 0.2f ? 1 : 0;
}

// ----------------------------------------------------------------------------------
// K8 + K9 fused: computeCov2DCUDA (backward.cu:144-274) and the backward preprocessCUDA
// (backward.cu:346-396) with the SH backward (backward.cu:20-139) and the cov3D backward
// (backward.cu:278-341).  One thread per Gaussian; dL_dcov3D stays in registers between
// the two reference kernels.  Every element of every output is written (zeros for
// Gaussians with radii <= 0), so the caller does not have to zero-fill them.
// ----------------------------------------------------------------------------------

__device__ __forceinline__ void store_sh_grad(float* __restrict__ dst, int M, const V3* g, int ncoef) {
  // dst -> (M,3) row of this Gaussian; coefficients >= ncoef are zero.
  if (M == 16) {
    float f[48];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const bool on = k < ncoef;
      f[3 * k] = on ? g[k].x : 0.f;
      f[3 * k + 1] = on ? g[k].y : 0.f;
      f[3 * k + 2] = on ? g[k].z : 0.f;
    }
    float4* q = reinterpret_cast<float4*>(dst);
#pragma unroll
    for (int i = 0; i < 12; ++i) q[i] = make_float4(f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]);
  } else {
    for (int k = 0; k < M; ++k) {
      const bool on = k < ncoef && k < 16;
      V3 v = {0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
        if (kk == k && on) v = g[kk];
      dst[3 * k] = v.x; dst[3 * k + 1] = v.y; dst[3 * k + 2] = v.z;
    }
  }
}

// Wave-cooperative store of 64 consecutive 192-byte SH-gradient rows (M = 16): every lane puts its row into a private
// LDS tile whose rows are padded to 13 float4 (bank-conflict free both ways) and the wave's 12 KB block leaves as
// twelve fully coalesced 1 KB stores (tools/microbench/rows192.hip: 5.55 TB/s against 4.55 TB/s for one thread per
// row).  One wave owns a tile: no workgroup barrier.  K8+K9: 122 -> 110 us.  The same route for the SH LOADS changes
// nothing (measured again with this tile: 110 us either way), so they stay one row per thread.
constexpr int SH_TILE_F4 = 64 * 13;
__device__ __forceinline__ void sh_tile_store_rows(float* __restrict__ dst_rows, int nrows, float4* __restrict__ tile, int lane,
                                                   const V3* g, int ncoef) {
  float f[48];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const bool on = k < ncoef;  // coefficients >= ncoef are zero
    f[3 * k] = on ? g[k].x : 0.f;
    f[3 * k + 1] = on ? g[k].y : 0.f;
    f[3 * k + 2] = on ? g[k].z : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) tile[lane * 13 + i] = make_float4(f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]);
  __builtin_amdgcn_wave_barrier();
  float4* __restrict__ q = reinterpret_cast<float4*>(dst_rows);
  const int nf4 = nrows * 12;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int e = k * 64 + lane, r = e / 12, c = e - 12 * r;
    if (e < nf4) {
      // streaming store: the gradient is written once and next read by the optimizer, long after; written through the
      // caches it evicts the parameters K1 streams again at the start of the next iteration (same-box A/B: preprocess
      // stage 156 -> 146 us, profiles/r03_b)
      typedef float nt_f4 __attribute__((ext_vector_type(4)));
      const float4 tv = tile[r * 13 + c];
      const nt_f4 nv = {tv.x, tv.y, tv.z, tv.w};
      __builtin_nontemporal_store(nv, reinterpret_cast<nt_f4*>(q) + e);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// (Three waves per SIMD: the 52 KB of store tiles per workgroup allow three workgroups per CU, ~140 VGPRs.)
// ROWS (PreBwdArgs::row_state, gsr_preprocess_backward_rows): the gradient arrays belong to the caller ACROSS calls.  Nine
// Gaussians of ten get all-zero gradients from a view (culled, or no pixel blended them); their rows are rewritten only if
// they do not hold this kernel's zeros already (row_state[g] != 0), so what leaves the chip per view is the rows that are
// non-zero now or were the last time -- 248 B x ~20 % instead of 248 B x P.  dL_dcov3D is always written.
template <bool ROWS>
__global__ void __launch_bounds__(GAUSS_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3)))
preprocess_backward_kernel(const PreBwdArgs a) {
  __shared__ float4 sh_tile[GAUSS_BLOCK / 64][SH_TILE_F4];
  // Blocks take the Gaussians in DESCENDING order: K8+K9 is the last kernel of an iteration and K1 the first of the next,
  // both stream the same 236 B of parameters per Gaussian -- what this kernel touches last is what K1 asks for first, and
  // finds in the 256 MB memory-side cache (same-box A/B: K1 + depth sort 144.8 -> 141.1 us, this kernel 106.9 -> 102.0)
  const int idx_raw = (int)((gridDim.x - 1u - blockIdx.x) * GAUSS_BLOCK + threadIdx.x);
  const bool live = idx_raw < a.P;  // (no early return: the SH-gradient rows leave wave-cooperatively)
  const int idx = live ? idx_raw : a.P - 1;
  const int ncoef = (a.D + 1) * (a.D + 1);
  V3 dmean = {0.f, 0.f, 0.f};
  float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  V3 dsh[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) dsh[k] = {0.f, 0.f, 0.f};
  V3 dscale = {0.f, 0.f, 0.f};
  float4 drot = make_float4(0.f, 0.f, 0.f, 0.f);
  V3 drgb = {0.f, 0.f, 0.f};  // dL_dRGB with the clamped channels zeroed (output of the "rgb" mode)
  bool nonzero_in = false;    // ROWS: some entry of the Gaussian's accumulator row is not zero (NaN != 0: counts)
  // The Gaussian's accumulator row (round 6: K7 adds into ONE 64-byte row per Gaussian, gsr_common.h ACC_*): requested for
  // every thread, visible or not -- the screen-space and opacity gradients leave through this kernel now (K7 used to add
  // into the caller's arrays directly), and a Gaussian without a pixel has an all-zero row.
  // (GSR_K9_NT_ACC=1, A/B builds: streaming loads of the rows and streaming stores of the two copied-out gradients, to keep
  //  them out of the memory-side cache that holds the parameters K1 streams again next -- measured: K1 does not notice either
  //  way once the table is no longer cleared per backward, and this kernel is 3-4 us SLOWER with them, 10 us on synth-v2,
  //  profiles/r06_e_accumulator_rows.md)
#ifndef GSR_K9_NT_ACC
#define GSR_K9_NT_ACC 0
#endif
  typedef float acc_f4 __attribute__((ext_vector_type(4)));
  const acc_f4* const acc_row = reinterpret_cast<const acc_f4*>(a.acc + (size_t)idx * ACC_ROW);
#if GSR_K9_NT_ACC
#define K9_ACC_LD(i) __builtin_nontemporal_load(acc_row + (i))
#else
#define K9_ACC_LD(i) acc_row[i]
#endif
  const acc_f4 acc_m2d_ = K9_ACC_LD(ACC_MEAN2D / 4), acc_col_ = K9_ACC_LD(ACC_COLOR / 4), acc_con_ = K9_ACC_LD(ACC_CONIC / 4);
#undef K9_ACC_LD
  const float4 acc_m2d = make_float4(acc_m2d_.x, acc_m2d_.y, acc_m2d_.z, acc_m2d_.w);  // dL_dmean2D.x, .y, (0), dL_dopacity
  const float4 acc_col = make_float4(acc_col_.x, acc_col_.y, acc_col_.z, acc_col_.w);  // dL_dcolor r, g, b, (0)
  const float4 acc_con = make_float4(acc_con_.x, acc_con_.y, acc_con_.z, acc_con_.w);  // dL_dconic x, y, (0), w
  if (a.acc_clean != nullptr && live) {
    // GSR_FLAG_ACC_SELF_CLEAN: the table is the caller's across backwards -- a row K7 touched (one in ten on the benchmark
    // view) goes back to zero here, whole 64-byte lines, so that the next backward starts from a zero table without a clear
    const bool dirty = !(acc_m2d.x == 0.f) || !(acc_m2d.y == 0.f) || !(acc_m2d.w == 0.f) || !(acc_con.x == 0.f) ||
                       !(acc_con.y == 0.f) || !(acc_con.w == 0.f) || !(acc_col.x == 0.f) || !(acc_col.y == 0.f) || !(acc_col.z == 0.f);
    if (dirty) {
      float4* const w = reinterpret_cast<float4*>(a.acc_clean + (size_t)idx * ACC_ROW);
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      w[0] = z; w[1] = z; w[2] = z; w[3] = z;
    }
  }

  if (live && a.radii[idx] > 0) {
    Cam cam;
    load_cam(cam, a.viewmatrix, a.projmatrix, a.shs ? a.campos : nullptr);
    const float* view = cam.view;
    const float* proj = cam.proj;
    // Every input of a visible Gaussian is requested HERE, together: in source order (each load in front of its use) a wave
    // went through five dependent memory round trips -- position / scale / rotation / conic gradient, the screen-space
    // gradient, the colour hand-off, scale / rotation again -- with the kernel's arithmetic in between.
    const V3 mean = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
    float sc0 = 0.f, sc1 = 0.f, sc2 = 0.f;
    float4 quat = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.scales != nullptr) {
      sc0 = a.scales[3 * idx];
      sc1 = a.scales[3 * idx + 1];
      sc2 = a.scales[3 * idx + 2];
      quat = reinterpret_cast<const float4*>(a.rotations)[idx];
    }
    const float4 gc = acc_con;  // (the row's three float4 were requested above, in front of the branch)
    const float g2x = acc_m2d.x, g2y = acc_m2d.y;
    V3 dRGBdx = {0.f, 0.f, 0.f}, dRGBdy = {0.f, 0.f, 0.f}, dRGBdz = {0.f, 0.f, 0.f}, dL_dRGB = {0.f, 0.f, 0.f};
    uint8_t cl = 0;
    if (a.shs != nullptr) {
      // d(RGB)/d(dir): K1 evaluated it from the SH record it held (sh_color_dir_derivatives) -- the record itself is not read
      // again (192 B per Gaussian at M = 16 against these 36); degree 0: the colour does not depend on the direction, K1
      // wrote nothing
      if (a.D > 0) {
        dRGBdx = *reinterpret_cast<const V3*>(a.dcol[0] + 3 * (size_t)idx);
        dRGBdy = *reinterpret_cast<const V3*>(a.dcol[1] + 3 * (size_t)idx);
        dRGBdz = *reinterpret_cast<const V3*>(a.dcol[2] + 3 * (size_t)idx);
      }
      cl = a.clamped[idx];
      dL_dRGB = {acc_col.x, acc_col.y, acc_col.z};
    }
    float c3[6];
    if (a.cov3D_precomp != nullptr) {
#pragma unroll
      for (int i = 0; i < 6; ++i) c3[i] = a.cov3D_precomp[6 * (size_t)idx + i];
    } else {
      cov3d_from_values(sc0, sc1, sc2, a.scale_modifier, quat, c3);  // what K1 computed, bit for bit
    }
    const V3 dL_dcon = {gc.x, gc.y, gc.w};
    nonzero_in = gc.x != 0.f || gc.y != 0.f || gc.w != 0.f || acc_m2d.w != 0.f || acc_col.x != 0.f || acc_col.y != 0.f || acc_col.z != 0.f;

    // ---- computeCov2DCUDA, backward.cu:159-273 ----
    V3 t = {view[0] * mean.x + view[4] * mean.y + view[8] * mean.z + view[12],
            view[1] * mean.x + view[5] * mean.y + view[9] * mean.z + view[13],
            view[2] * mean.x + view[6] * mean.y + view[10] * mean.z + view[14]};
    const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    const float x_grad_mul = txtz < -limx || txtz > limx ? 0.f : 1.f;
    const float y_grad_mul = tytz < -limy || tytz > limy ? 0.f : 1.f;
    const float h_x = a.h_x, h_y = a.h_y;
    const M3 J = mk(h_x / t.z, 0.0f, -(h_x * t.x) / (t.z * t.z), 0.0f, h_y / t.z, -(h_y * t.y) / (t.z * t.z), 0, 0, 0);
    const M3 Wm = mk(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    const M3 Vrk = mk(c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]);
    const M3 T = mul(Wm, J);
    const M3 cov2D = mul(mul(tr(T), tr(Vrk)), T);
    const float ca = cov2D.m[0][0] + 0.3f, cb = cov2D.m[0][1], cc = cov2D.m[1][1] + 0.3f;
    const float denom = ca * cc - cb * cb;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    const auto& Tm = T.m;
    if (denom2inv != 0) {
      dL_da = denom2inv * (-cc * cc * dL_dcon.x + 2 * cb * cc * dL_dcon.y + (denom - ca * cc) * dL_dcon.z);
      dL_dc = denom2inv * (-ca * ca * dL_dcon.z + 2 * ca * cb * dL_dcon.y + (denom - ca * cc) * dL_dcon.x);
      dL_db = denom2inv * 2 * (cb * cc * dL_dcon.x - (denom + 2 * cb * cb) * dL_dcon.y + ca * cb * dL_dcon.z);
      dcov[0] = (Tm[0][0] * Tm[0][0] * dL_da + Tm[0][0] * Tm[1][0] * dL_db + Tm[1][0] * Tm[1][0] * dL_dc);
      dcov[3] = (Tm[0][1] * Tm[0][1] * dL_da + Tm[0][1] * Tm[1][1] * dL_db + Tm[1][1] * Tm[1][1] * dL_dc);
      dcov[5] = (Tm[0][2] * Tm[0][2] * dL_da + Tm[0][2] * Tm[1][2] * dL_db + Tm[1][2] * Tm[1][2] * dL_dc);
      dcov[1] = 2 * Tm[0][0] * Tm[0][1] * dL_da + (Tm[0][0] * Tm[1][1] + Tm[0][1] * Tm[1][0]) * dL_db +
                2 * Tm[1][0] * Tm[1][1] * dL_dc;
      dcov[2] = 2 * Tm[0][0] * Tm[0][2] * dL_da + (Tm[0][0] * Tm[1][2] + Tm[0][2] * Tm[1][0]) * dL_db +
                2 * Tm[1][0] * Tm[1][2] * dL_dc;
      dcov[4] = 2 * Tm[0][2] * Tm[0][1] * dL_da + (Tm[0][1] * Tm[1][2] + Tm[0][2] * Tm[1][1]) * dL_db +
                2 * Tm[1][1] * Tm[1][2] * dL_dc;
    }
    const auto& V = Vrk.m;
    const float dL_dT00 = 2 * (Tm[0][0] * V[0][0] + Tm[0][1] * V[0][1] + Tm[0][2] * V[0][2]) * dL_da +
                          (Tm[1][0] * V[0][0] + Tm[1][1] * V[0][1] + Tm[1][2] * V[0][2]) * dL_db;
    const float dL_dT01 = 2 * (Tm[0][0] * V[1][0] + Tm[0][1] * V[1][1] + Tm[0][2] * V[1][2]) * dL_da +
                          (Tm[1][0] * V[1][0] + Tm[1][1] * V[1][1] + Tm[1][2] * V[1][2]) * dL_db;
    const float dL_dT02 = 2 * (Tm[0][0] * V[2][0] + Tm[0][1] * V[2][1] + Tm[0][2] * V[2][2]) * dL_da +
                          (Tm[1][0] * V[2][0] + Tm[1][1] * V[2][1] + Tm[1][2] * V[2][2]) * dL_db;
    const float dL_dT10 = 2 * (Tm[1][0] * V[0][0] + Tm[1][1] * V[0][1] + Tm[1][2] * V[0][2]) * dL_dc +
                          (Tm[0][0] * V[0][0] + Tm[0][1] * V[0][1] + Tm[0][2] * V[0][2]) * dL_db;
    const float dL_dT11 = 2 * (Tm[1][0] * V[1][0] + Tm[1][1] * V[1][1] + Tm[1][2] * V[1][2]) * dL_dc +
                          (Tm[0][0] * V[1][0] + Tm[0][1] * V[1][1] + Tm[0][2] * V[1][2]) * dL_db;
    const float dL_dT12 = 2 * (Tm[1][0] * V[2][0] + Tm[1][1] * V[2][1] + Tm[1][2] * V[2][2]) * dL_dc +
                          (Tm[0][0] * V[2][0] + Tm[0][1] * V[2][1] + Tm[0][2] * V[2][2]) * dL_db;
    const auto& Wx = Wm.m;
    const float dL_dJ00 = Wx[0][0] * dL_dT00 + Wx[0][1] * dL_dT01 + Wx[0][2] * dL_dT02;
    const float dL_dJ02 = Wx[2][0] * dL_dT00 + Wx[2][1] * dL_dT01 + Wx[2][2] * dL_dT02;
    const float dL_dJ11 = Wx[1][0] * dL_dT10 + Wx[1][1] * dL_dT11 + Wx[1][2] * dL_dT12;
    const float dL_dJ12 = Wx[2][0] * dL_dT10 + Wx[2][1] * dL_dT11 + Wx[2][2] * dL_dT12;
    const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 +
                         (2 * h_y * t.y) * tz3 * dL_dJ12;
    // transformVec4x3Transpose, auxiliary.h:89-97 (assignment, backward.cu:273)
    dmean = {view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz,
             view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz,
             view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz};

    // ---- preprocessCUDA (backward), backward.cu:370-395 ----
    const V3 m = mean;
    const float m_hw = proj[3] * m.x + proj[7] * m.y + proj[11] * m.z + proj[15];
    const float m_w = 1.0f / (m_hw + 0.0000001f);
    const float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    nonzero_in = nonzero_in || g2x != 0.f || g2y != 0.f;
    V3 dL_dmean;
    dL_dmean.x = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    dL_dmean.y = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    dL_dmean.z = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    dmean = dmean + dL_dmean;

    if (a.shs != nullptr) {
      // computeColorFromSH (backward), backward.cu:20-139
      const V3 dir_orig = {m.x - cam.campos[0], m.y - cam.campos[1], m.z - cam.campos[2]};
      const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
      const V3 dir = {dir_orig.x / len, dir_orig.y / len, dir_orig.z / len};
      nonzero_in = nonzero_in || dL_dRGB.x != 0.f || dL_dRGB.y != 0.f || dL_dRGB.z != 0.f;
      dL_dRGB.x *= (cl & 1) ? 0.f : 1.f;
      dL_dRGB.y *= (cl & 2) ? 0.f : 1.f;
      dL_dRGB.z *= (cl & 4) ? 0.f : 1.f;
      drgb = dL_dRGB;
      const float x = dir.x, y = dir.y, z = dir.z;
      dsh[0] = SH_C0 * dL_dRGB;
      if (a.D > 0) {
        dsh[1] = (-SH_C1 * y) * dL_dRGB;
        dsh[2] = (SH_C1 * z) * dL_dRGB;
        dsh[3] = (-SH_C1 * x) * dL_dRGB;
        if (a.D > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          dsh[4] = (SH_C2[0] * xy) * dL_dRGB;
          dsh[5] = (SH_C2[1] * yz) * dL_dRGB;
          dsh[6] = (SH_C2[2] * (2.f * zz - xx - yy)) * dL_dRGB;
          dsh[7] = (SH_C2[3] * xz) * dL_dRGB;
          dsh[8] = (SH_C2[4] * (xx - yy)) * dL_dRGB;
          if (a.D > 2) {
            dsh[9] = (SH_C3[0] * y * (3.f * xx - yy)) * dL_dRGB;
            dsh[10] = (SH_C3[1] * xy * z) * dL_dRGB;
            dsh[11] = (SH_C3[2] * y * (4.f * zz - xx - yy)) * dL_dRGB;
            dsh[12] = (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dL_dRGB;
            dsh[13] = (SH_C3[4] * x * (4.f * zz - xx - yy)) * dL_dRGB;
            dsh[14] = (SH_C3[5] * z * (xx - yy)) * dL_dRGB;
            dsh[15] = (SH_C3[6] * x * (xx - 3.f * yy)) * dL_dRGB;
          }
        }
      }
      const V3 dL_ddir = {dot3(dRGBdx, dL_dRGB), dot3(dRGBdy, dL_dRGB), dot3(dRGBdz, dL_dRGB)};
      // dnormvdv, auxiliary.h:107-117
      const V3 v = dir_orig, dv = dL_ddir;
      const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
      const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
      V3 dm;
      dm.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
      dm.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
      dm.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
      dmean = dmean + dm;
    }

    if (a.scales != nullptr) {
      // computeCov3D (backward), backward.cu:278-341
      const float4 q = quat;
      const float r = q.x, x = q.y, y = q.z, z = q.w;
      const M3 R = mk(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                      2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                      2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
      M3 S = mk(1, 0, 0, 0, 1, 0, 0, 0, 1);
      const V3 s = {a.scale_modifier * sc0, a.scale_modifier * sc1, a.scale_modifier * sc2};
      S.m[0][0] = s.x; S.m[1][1] = s.y; S.m[2][2] = s.z;
      const M3 Mm = mul(S, R);
      const M3 dL_dSigma = mk(dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
                              0.5f * dcov[2], 0.5f * dcov[4], dcov[5]);
      M3 M2;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) M2.m[c][rr] = Mm.m[c][rr] * 2.0f;
      const M3 dL_dM = mul(M2, dL_dSigma);
      const M3 Rt = tr(R);
      M3 Q = tr(dL_dM);  // dL_dMt
      dscale.x = Rt.m[0][0] * Q.m[0][0] + Rt.m[0][1] * Q.m[0][1] + Rt.m[0][2] * Q.m[0][2];
      dscale.y = Rt.m[1][0] * Q.m[1][0] + Rt.m[1][1] * Q.m[1][1] + Rt.m[1][2] * Q.m[1][2];
      dscale.z = Rt.m[2][0] * Q.m[2][0] + Rt.m[2][1] * Q.m[2][1] + Rt.m[2][2] * Q.m[2][2];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        Q.m[0][k] *= s.x;
        Q.m[1][k] *= s.y;
        Q.m[2][k] *= s.z;
      }
      const auto& D_ = Q.m;
      drot.x = 2 * z * (D_[0][1] - D_[1][0]) + 2 * y * (D_[2][0] - D_[0][2]) + 2 * x * (D_[1][2] - D_[2][1]);
      drot.y = 2 * y * (D_[1][0] + D_[0][1]) + 2 * z * (D_[2][0] + D_[0][2]) + 2 * r * (D_[1][2] - D_[2][1]) -
               4 * x * (D_[2][2] + D_[1][1]);
      drot.z = 2 * x * (D_[1][0] + D_[0][1]) + 2 * r * (D_[2][0] - D_[0][2]) + 2 * z * (D_[1][2] + D_[2][1]) -
               4 * y * (D_[2][2] + D_[0][0]);
      drot.w = 2 * r * (D_[0][1] - D_[1][0]) + 2 * x * (D_[2][0] + D_[0][2]) + 2 * y * (D_[1][2] + D_[2][1]) -
               4 * z * (D_[1][1] + D_[0][0]);
    }
  }

  if (ROWS) {
    if (!live) return;
    // the inputs of a Gaussian decide: all-zero accumulator rows give all-zero gradients (with precomputed colours the
    // colour accumulator is not this kernel's input)
    const bool nonzero = nonzero_in;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (a.dL_dcov3D != nullptr) a.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
    const uint8_t was = a.row_state[idx];
    if (!nonzero && was == 0) return;  // the rows hold the zeros written when this Gaussian last went from non-zero to zero
    if ((nonzero ? 1 : 0) != was) a.row_state[idx] = nonzero ? 1 : 0;
    if (a.dL_dsh != nullptr) store_sh_grad(a.dL_dsh + (size_t)idx * a.M * 3, a.M, dsh, ncoef);
  } else if (a.dL_dsh != nullptr) {
    if (a.M == 16 && (reinterpret_cast<uintptr_t>(a.dL_dsh) & 15u) == 0) {
      const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6);
      const int row0 = idx_raw - lane;  // first row of this wave
      if (row0 < a.P) sh_tile_store_rows(a.dL_dsh + (size_t)row0 * 48, min(64, a.P - row0), sh_tile[wv], lane, dsh, ncoef);
    } else if (live) {
      store_sh_grad(a.dL_dsh + (size_t)idx * a.M * 3, a.M, dsh, ncoef);
    }
  }
  if (!live) return;
#if defined(GSR_K9_NT_GRADS) && GSR_K9_NT_GRADS
#define K9_ST(p, v) __builtin_nontemporal_store((v), (p))  // (A/B build)
#else
#define K9_ST(p, v) (*(p) = (v))
#endif
  K9_ST(&a.dL_dmeans3D[3 * (size_t)idx], dmean.x);
  K9_ST(&a.dL_dmeans3D[3 * (size_t)idx + 1], dmean.y);
  K9_ST(&a.dL_dmeans3D[3 * (size_t)idx + 2], dmean.z);
  // the two gradients K7 accumulates for the caller (backward.cu:545-546, 554) and, with precomputed colours, the colour's
  // (:523): copied out of the accumulator row (a Gaussian that is not visible has a zero row: K7 never touched it)
#if GSR_K9_NT_ACC
#define K9_ST2(p, v) __builtin_nontemporal_store((v), (p))
#else
#define K9_ST2(p, v) K9_ST(p, v)
#endif
  K9_ST2(&a.dL_dmean2D[3 * (size_t)idx], acc_m2d.x);
  K9_ST2(&a.dL_dmean2D[3 * (size_t)idx + 1], acc_m2d.y);
  K9_ST2(&a.dL_dmean2D[3 * (size_t)idx + 2], 0.f);
  K9_ST2(&a.dL_dopacity[idx], acc_m2d.w);
#undef K9_ST2
  if (a.dL_dcolor != nullptr) {
    K9_ST(&a.dL_dcolor[3 * (size_t)idx], acc_col.x);
    K9_ST(&a.dL_dcolor[3 * (size_t)idx + 1], acc_col.y);
    K9_ST(&a.dL_dcolor[3 * (size_t)idx + 2], acc_col.z);
  }
  if (!ROWS) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (a.dL_dcov3D != nullptr) K9_ST(&a.dL_dcov3D[6 * (size_t)idx + i], dcov[i]);  // (null: no precomputed covariance to take a gradient)
  }
  if (a.dL_drgb != nullptr) {
    a.dL_drgb[3 * (size_t)idx] = drgb.x;
    a.dL_drgb[3 * (size_t)idx + 1] = drgb.y;
    a.dL_drgb[3 * (size_t)idx + 2] = drgb.z;
  }
  if (a.dL_dscale != nullptr) {
    K9_ST(&a.dL_dscale[3 * (size_t)idx], dscale.x);
    K9_ST(&a.dL_dscale[3 * (size_t)idx + 1], dscale.y);
    K9_ST(&a.dL_dscale[3 * (size_t)idx + 2], dscale.z);
  }
  if (a.dL_drot != nullptr) reinterpret_cast<float4*>(a.dL_drot)[idx] = drot;
#undef K9_ST
}

// ----------------------------------------------------------------------------------
// SH gradient of a BATCH of views from their colour gradients (multi-GPU exchange, gaussianeditor_amd/multiview.py).
// Per view the SH gradient is rank one: dL_dsh[k] = c_k(dir) * dL_dRGB (backward.cu:44-48, 59-61, 73-77, 92-98), and
// dir depends only on the Gaussian's position and the view's camera centre.  So N ranks exchange 3 floats per
// Gaussian and view (all-gather) instead of summing 3M (all-reduce of 192 B per Gaussian at M = 16), and every rank
// rebuilds sum_v c_k(dir_v) * dL_dRGB_v itself, views in ascending order: the operations and the order of a single
// process that accumulates the views' gradients one after the other, so all replicas hold bit-identical sums.
// The c_k are spelled exactly as in preprocess_backward_kernel.
// ----------------------------------------------------------------------------------
// One view's SH-gradient terms of a Gaussian: t[k] = c_k(dir) * dL_dRGB for the coefficients of degree <= D, zero above.
__device__ __forceinline__ void sh_grad_terms(int D, V3 m, const float* __restrict__ cam, V3 dL_dRGB, V3* t) {
  const V3 dir_orig = {m.x - cam[0], m.y - cam[1], m.z - cam[2]};
  const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
  const V3 dir = {dir_orig.x / len, dir_orig.y / len, dir_orig.z / len};
  const float x = dir.x, y = dir.y, z = dir.z;
#pragma unroll
  for (int k = 0; k < 16; ++k) t[k] = {0.f, 0.f, 0.f};
  t[0] = SH_C0 * dL_dRGB;
  if (D > 0) {
    t[1] = (-SH_C1 * y) * dL_dRGB;
    t[2] = (SH_C1 * z) * dL_dRGB;
    t[3] = (-SH_C1 * x) * dL_dRGB;
    if (D > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      t[4] = (SH_C2[0] * xy) * dL_dRGB;
      t[5] = (SH_C2[1] * yz) * dL_dRGB;
      t[6] = (SH_C2[2] * (2.f * zz - xx - yy)) * dL_dRGB;
      t[7] = (SH_C2[3] * xz) * dL_dRGB;
      t[8] = (SH_C2[4] * (xx - yy)) * dL_dRGB;
      if (D > 2) {
        t[9] = (SH_C3[0] * y * (3.f * xx - yy)) * dL_dRGB;
        t[10] = (SH_C3[1] * xy * z) * dL_dRGB;
        t[11] = (SH_C3[2] * y * (4.f * zz - xx - yy)) * dL_dRGB;
        t[12] = (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dL_dRGB;
        t[13] = (SH_C3[4] * x * (4.f * zz - xx - yy)) * dL_dRGB;
        t[14] = (SH_C3[5] * z * (xx - yy)) * dL_dRGB;
        t[15] = (SH_C3[6] * x * (xx - 3.f * yy)) * dL_dRGB;
      }
    }
  }
}

__global__ void __launch_bounds__(GAUSS_BLOCK) sh_grad_compose_kernel(int P, int D, int M, int N,
                                                                     const float* __restrict__ means3D,
                                                                     const float* __restrict__ campos,  // (N,3)
                                                                     const float* __restrict__ dL_drgb,  // (N,P,3)
                                                                     float* __restrict__ dL_dsh) {      // (P,M,3)
  __shared__ float4 sh_tile[GAUSS_BLOCK / 64][SH_TILE_F4];
  const int idx_raw = (int)(blockIdx.x * GAUSS_BLOCK + threadIdx.x);
  const bool live = idx_raw < P;  // (no early return: the rows leave wave-cooperatively)
  const int idx = live ? idx_raw : P - 1;
  const int ncoef = (D + 1) * (D + 1);
  V3 dsh[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) dsh[k] = {0.f, 0.f, 0.f};
  const V3 m = {means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]};
  for (int v = 0; v < (live ? N : 0); ++v) {
    const float* g = dL_drgb + ((size_t)v * P + idx) * 3;
    const V3 dL_dRGB = {g[0], g[1], g[2]};
    // a view that does not see the Gaussian (or whose colour was clamped in all channels) contributes exact zeros
    if (dL_dRGB.x == 0.f && dL_dRGB.y == 0.f && dL_dRGB.z == 0.f) continue;
    V3 t[16];
    sh_grad_terms(D, m, campos + 3 * v, dL_dRGB, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) dsh[k] = dsh[k] + t[k];
  }
  if (M == 16 && (reinterpret_cast<uintptr_t>(dL_dsh) & 15u) == 0) {
    const int lane = (int)(threadIdx.x & 63u), row0 = idx_raw - lane;
    if (row0 < P) sh_tile_store_rows(dL_dsh + (size_t)row0 * 48, min(64, P - row0), sh_tile[threadIdx.x >> 6], lane, dsh, ncoef);
  } else if (live) {
    store_sh_grad(dL_dsh + (size_t)idx * M * 3, M, dsh, ncoef);
  }
}

// ----------------------------------------------------------------------------------
// "Touched rows" exchange of the multi-GPU step (gaussianeditor_amd/multiview.py).  A view only produces gradients for
// the Gaussians it blends, so a rank sends the rows that are not entirely zero: index + 14 floats (means3D, scales,
// rotations, means2D, opacities) + its 3-float colour gradient.  Every rank then adds the gathered rows of view 0, 1, ...
// in that order to zeroed dense arrays -- the operations and the order of one process accumulating the views -- and
// rebuilds the SH gradient from the colour gradients with the terms of sh_grad_compose_kernel.
// ----------------------------------------------------------------------------------
struct RowSet {
  const float* data[8];
  int row_len[8];
  int n;
};
__global__ void __launch_bounds__(256) touched_rows_kernel(int64_t P, RowSet rs, uint8_t* __restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  bool any = false;
  for (int t = 0; t < rs.n; ++t) {
    const float* r = rs.data[t] + (size_t)i * rs.row_len[t];
    if (rs.row_len[t] == (int)ACC_ROW && (reinterpret_cast<uintptr_t>(rs.data[t]) & 15u) == 0) {  // the backward's accumulator rows
      const float4* r4 = reinterpret_cast<const float4*>(r);
      const float4 v0 = r4[0], v1 = r4[1], v2 = r4[2];  // (columns 12..15 are never written)
      any = any || !(v0.x == 0.f) || !(v0.y == 0.f) || !(v0.z == 0.f) || !(v0.w == 0.f) || !(v1.x == 0.f) || !(v1.y == 0.f) ||
            !(v1.z == 0.f) || !(v1.w == 0.f) || !(v2.x == 0.f) || !(v2.y == 0.f) || !(v2.z == 0.f) || !(v2.w == 0.f);
      continue;
    }
    for (int k = 0; k < rs.row_len[t]; ++k) any = any || !(r[k] == 0.0f);  // NaN counts as touched
  }
  mask[i] = any ? 1 : 0;
}

struct DenseGrads {
  float *means3D, *scales, *rotations, *means2D, *opacities, *sh;  // (P,3|3|4|3|1|M*3); sh may be null
};
// One view's message (all 32-bit words; `nb` = ceil(P / 1024) row blocks, `cap` >= count rows of capacity):
//   [0..2] camera centre, [3] count (int), [4 .. 4+nb) number of touched rows before row block b (the compaction's
//   block offsets), then idx (cap ints), means3D (3 cap), scales (3 cap), rotations (4 cap), means2D (3 cap),
//   opacities (cap), rgb (3 cap).
struct ViewMsg {
  const float* cam;
  const uint32_t* boff;
  const int32_t* idx;
  const float *means3D, *scales, *rotations, *means2D, *opacities, *rgb;
  uint32_t count;
};
__host__ __device__ inline int64_t view_message_words(int64_t P, int64_t cap) {
  return 4 + (P + VIEW_MSG_ROWS - 1) / VIEW_MSG_ROWS + 18 * cap;
}
__device__ __forceinline__ ViewMsg carve_view_message(const float* m, int64_t P, int64_t cap) {
  ViewMsg v;
  const int64_t nb = (P + VIEW_MSG_ROWS - 1) / VIEW_MSG_ROWS;
  v.cam = m;
  v.count = __float_as_uint(m[3]);
  v.boff = reinterpret_cast<const uint32_t*>(m + 4);
  const float* r = m + 4 + nb;
  v.idx = reinterpret_cast<const int32_t*>(r);
  v.means3D = r + cap;
  v.scales = r + 4 * cap;
  v.rotations = r + 7 * cap;
  v.means2D = r + 11 * cap;
  v.opacities = r + 14 * cap;
  v.rgb = r + 15 * cap;
  return v;
}

__global__ void __launch_bounds__(256) view_message_header_kernel(int64_t P, const float* __restrict__ campos,
                                                                 const uint32_t* __restrict__ block_off,
                                                                 const uint64_t* __restrict__ total, float* __restrict__ msg) {
  const int64_t nb = (P + VIEW_MSG_ROWS - 1) / VIEW_MSG_ROWS;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < 3) msg[i] = campos[i];
  if (i == 3) msg[3] = __uint_as_float((uint32_t)*total);
  if (i < nb) msg[4 + i] = __uint_as_float(block_off[i]);
}

// Adds the messages of all views, view 0 first, per Gaussian in registers (the operations and the order of one process
// that accumulates the views one after the other; a Gaussian no view touched gets zeros) and writes the dense gradients.
// The rows a view sends for the 1024-row message block b are the contiguous slice [boff[b], boff[b+1]) of its packed rows.
// A workgroup of 256 threads owns 256 consecutive Gaussians, a quarter of a message block: it walks the slice of every
// view, notes in an LDS position map where the rows of its quarter are, and the thread of Gaussian g then adds them in
// view order.  The SH rows leave wave-cooperatively (sh_tile_store_rows).
constexpr int VIEW_BATCH = 8;  // views whose position maps are resident in LDS at a time
// SPARSE (row_valid != null): a Gaussian no view sent a row for gets row_valid[g] = 0 and NOTHING is written to its gradient
// rows (they keep whatever they held); the others get row_valid[g] = 1 and their sums.  The consumer treats invalid rows as
// zero gradients (gsr_adam_step_rows with grad_valid = row_valid: the moments still decay).  At 2 / 8 views 81 / 43 % of the rows are
// invalid, and the dense write of 248 B per Gaussian is most of this kernel's time.
template <bool SPARSE>
__global__ void __launch_bounds__(GAUSS_BLOCK) view_messages_accumulate_kernel(int64_t P, int D, int M, int n_views,
                                                                              const float* __restrict__ messages,
                                                                              int64_t stride_words, int64_t cap,
                                                                              const float* __restrict__ means3D_param,
                                                                              DenseGrads d, uint8_t* __restrict__ row_valid) {
  __shared__ uint16_t posmap[VIEW_BATCH][GAUSS_BLOCK];
  __shared__ float4 sh_tile[GAUSS_BLOCK / 64][SH_TILE_F4];
  constexpr int QUARTERS = VIEW_MSG_ROWS / GAUSS_BLOCK;
  const int64_t nb = (P + VIEW_MSG_ROWS - 1) / VIEW_MSG_ROWS;
  const int64_t b = blockIdx.x / QUARTERS;                       // message block
  const int64_t g0 = (int64_t)blockIdx.x * GAUSS_BLOCK, g_raw = g0 + threadIdx.x;
  const bool live = g_raw < P;
  const int64_t g = live ? g_raw : P - 1;
  const int ncoef = (D + 1) * (D + 1);
  V3 am = {0.f, 0.f, 0.f}, as = {0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f};
  float4 ar = make_float4(0.f, 0.f, 0.f, 0.f);
  float ao = 0.f;
  V3 dsh[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) dsh[k] = {0.f, 0.f, 0.f};
  V3 m = {0.f, 0.f, 0.f};
  if (d.sh != nullptr) m = {means3D_param[3 * g], means3D_param[3 * g + 1], means3D_param[3 * g + 2]};
  bool touched = false;
  // Every step below issues the loads of ALL views of the batch before it uses any of them: the kernel is a chain of
  // dependent round trips (slice bounds -> row indices -> rows), and walked view by view it paid that chain once per view
  // (0.011 ms per view and launch at 1 M Gaussians; the arithmetic is a fraction of that).
  for (int vb = 0; vb < n_views; vb += VIEW_BATCH) {
    const int nv = min(VIEW_BATCH, n_views - vb);
    uint32_t lo[VIEW_BATCH], hi[VIEW_BATCH];
#pragma unroll
    for (int u = 0; u < VIEW_BATCH; ++u) {
      lo[u] = hi[u] = 0u;
      if (u < nv) {
        const ViewMsg v = carve_view_message(messages + (size_t)(vb + u) * stride_words, P, cap);
        lo[u] = v.boff[b];
        hi[u] = b + 1 < nb ? v.boff[b + 1] : v.count;
      }
      posmap[u][threadIdx.x] = 0xffffu;
    }
    __syncthreads();
    int32_t first[VIEW_BATCH];
#pragma unroll
    for (int u = 0; u < VIEW_BATCH; ++u) {  // the first 256 rows of every slice (a view sends ~100 rows per message block)
      first[u] = -1;
      if (u < nv && hi[u] > lo[u]) {
        const ViewMsg v = carve_view_message(messages + (size_t)(vb + u) * stride_words, P, cap);
        first[u] = v.idx[min(lo[u] + threadIdx.x, hi[u] - 1u)];  // (clamped: an unconditional load)
      }
    }
#pragma unroll
    for (int u = 0; u < VIEW_BATCH; ++u) {
      if (u < nv && lo[u] + threadIdx.x < hi[u]) {
        const int64_t q = (int64_t)first[u] - g0;
        if (q >= 0 && q < GAUSS_BLOCK) posmap[u][q] = (uint16_t)threadIdx.x;
      }
    }
    for (int u = 0; u < nv; ++u) {  // (rare: more than 256 rows of a view in this message block)
      if (hi[u] - lo[u] <= (uint32_t)GAUSS_BLOCK) continue;
      const ViewMsg v = carve_view_message(messages + (size_t)(vb + u) * stride_words, P, cap);
      for (uint32_t j = lo[u] + GAUSS_BLOCK + threadIdx.x; j < hi[u]; j += GAUSS_BLOCK) {
        const int64_t q = (int64_t)v.idx[j] - g0;
        if (q >= 0 && q < GAUSS_BLOCK) posmap[u][q] = (uint16_t)(j - lo[u]);
      }
    }
    __syncthreads();
    constexpr int VG = 4;  // views whose rows are in flight together (17 registers each)
#pragma unroll
    for (int u0 = 0; u0 < VIEW_BATCH; u0 += VG) {
      if (u0 >= nv || cap <= 0) break;
      float r[VG][17];
      bool hit[VG];
#pragma unroll
      for (int k = 0; k < VG; ++k) {
        const int u = u0 + k;
        hit[k] = false;
        if (u < nv) {
          const uint32_t o = posmap[u][threadIdx.x];
          hit[k] = o != 0xffffu && live;
          const ViewMsg v = carve_view_message(messages + (size_t)(vb + u) * stride_words, P, cap);
          const size_t j = min((size_t)lo[u] + (hit[k] ? o : 0u), (size_t)cap - 1);  // (any valid row where there is none)
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            r[k][i] = v.means3D[3 * j + i];
            r[k][3 + i] = v.scales[3 * j + i];
            r[k][10 + i] = v.means2D[3 * j + i];
            r[k][14 + i] = v.rgb[3 * j + i];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) r[k][6 + i] = v.rotations[4 * j + i];
          r[k][13] = v.opacities[j];
        }
      }
#pragma unroll
      for (int k = 0; k < VG; ++k) {  // view order
        const int u = u0 + k;
        if (u >= nv || !hit[k]) continue;
        touched = true;
        am = am + V3{r[k][0], r[k][1], r[k][2]};
        as = as + V3{r[k][3], r[k][4], r[k][5]};
        a2 = a2 + V3{r[k][10], r[k][11], r[k][12]};
        ar.x += r[k][6];
        ar.y += r[k][7];
        ar.z += r[k][8];
        ar.w += r[k][9];
        ao += r[k][13];
        if (d.sh == nullptr) continue;
        const V3 dL_dRGB = {r[k][14], r[k][15], r[k][16]};
        if (dL_dRGB.x == 0.f && dL_dRGB.y == 0.f && dL_dRGB.z == 0.f) continue;  // (as sh_grad_compose_kernel)
        const ViewMsg v = carve_view_message(messages + (size_t)(vb + u) * stride_words, P, cap);
        V3 t[16];
        sh_grad_terms(D, m, v.cam, dL_dRGB, t);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) dsh[kk] = dsh[kk] + t[kk];
      }
    }
    __syncthreads();
  }
  if (SPARSE) {  // only the rows some view sent: one row per thread (no tile: the rows are not consecutive)
    if (!live) return;
    row_valid[g] = touched ? 1 : 0;
    if (!touched) return;
    if (d.sh != nullptr) store_sh_grad(d.sh + (size_t)g * M * 3, M, dsh, ncoef);
  } else if (d.sh != nullptr) {
    if (M == 16 && (reinterpret_cast<uintptr_t>(d.sh) & 15u) == 0) {
      const int lane = (int)(threadIdx.x & 63u);
      const int64_t row0 = g_raw - lane;
      if (row0 < P)
        sh_tile_store_rows(d.sh + (size_t)row0 * 48, (int)min((int64_t)64, P - row0), sh_tile[threadIdx.x >> 6], lane, dsh, ncoef);
    } else if (live) {
      store_sh_grad(d.sh + (size_t)g * M * 3, M, dsh, ncoef);
    }
  }
  if (!live) return;
  d.means3D[3 * g] = am.x; d.means3D[3 * g + 1] = am.y; d.means3D[3 * g + 2] = am.z;
  d.scales[3 * g] = as.x; d.scales[3 * g + 1] = as.y; d.scales[3 * g + 2] = as.z;
  d.means2D[3 * g] = a2.x; d.means2D[3 * g + 1] = a2.y; d.means2D[3 * g + 2] = a2.z;
  d.rotations[4 * g] = ar.x; d.rotations[4 * g + 1] = ar.y; d.rotations[4 * g + 2] = ar.z; d.rotations[4 * g + 3] = ar.w;
  d.opacities[g] = ao;
}

// ----------------------------------------------------------------------------------
// Host-side launchers (called from gsr_capi.hip).
// ----------------------------------------------------------------------------------
hipError_t launch_preprocess(hipStream_t s, const PreArgs& a) {
  const int nb = (a.P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  const bool main_mode = a.cov3D_precomp == nullptr && a.colors_precomp == nullptr && !a.skip_color && a.M == 16 && a.D == 3;
  if (main_mode) hipLaunchKernelGGL(preprocess_kernel<true>, dim3(nb), dim3(GAUSS_BLOCK), 0, s, a);
  else hipLaunchKernelGGL(preprocess_kernel<false>, dim3(nb), dim3(GAUSS_BLOCK), 0, s, a);
  return hipGetLastError();
}
// Test-only introspection: unpack the gather records into the reference's separate arrays.
__global__ void __launch_bounds__(GAUSS_BLOCK) export_geom_kernel(int P, const Geom g, float* means2D, float* depths,
                                                                 float* rgb, float* conic_opacity, uint32_t* tiles_touched,
                                                                 uint8_t* clamped) {
  const int idx = (int)(blockIdx.x * GAUSS_BLOCK + threadIdx.x);
  if (idx >= P) return;
  const uint32_t wh = g.rect[idx].y;  // width | height << 16 of the tile rectangle: its area is tiles_touched
  const bool live = wh != 0;          // records of culled Gaussians are uninitialised
  if (tiles_touched) tiles_touched[idx] = (wh & 0xffffu) * (wh >> 16);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 r0 = live ? g.rec0[idx] : z, r1 = live ? g.rec1[idx] : z, r2 = live ? g.rec2[idx] : z;
  if (means2D) { means2D[2 * idx] = r1.x; means2D[2 * idx + 1] = r1.y; }
  if (depths) depths[idx] = r1.z;
  if (rgb) { rgb[3 * idx] = r2.x; rgb[3 * idx + 1] = r2.y; rgb[3 * idx + 2] = r2.z; }
  if (conic_opacity) reinterpret_cast<float4*>(conic_opacity)[idx] = r0;
  if (clamped) {
    const uint8_t c = live ? g.clamped[idx] : 0;
    clamped[3 * idx] = c & 1; clamped[3 * idx + 1] = (c >> 1) & 1; clamped[3 * idx + 2] = (c >> 2) & 1;
  }
}
__global__ void __launch_bounds__(GAUSS_BLOCK) export_cov3d_kernel(int P, const float* scales, float scale_modifier,
                                                                  const float* rotations, float* cov3D) {
  const int idx = (int)(blockIdx.x * GAUSS_BLOCK + threadIdx.x);
  if (idx >= P) return;
  float c3[6];
  cov3d_from_scale_rot(scales, scale_modifier, rotations, idx, c3);
#pragma unroll
  for (int i = 0; i < 6; ++i) cov3D[6 * (size_t)idx + i] = c3[i];
}
hipError_t launch_export_cov3d(hipStream_t s, int P, const float* scales, float scale_modifier, const float* rotations,
                               float* cov3D) {
  hipLaunchKernelGGL(export_cov3d_kernel, dim3((P + GAUSS_BLOCK - 1) / GAUSS_BLOCK), dim3(GAUSS_BLOCK), 0, s, P, scales, scale_modifier,
                     rotations, cov3D);
  return hipGetLastError();
}
hipError_t launch_export_geom(hipStream_t s, int P, const Geom& g, float* means2D, float* depths, float* rgb,
                              float* conic_opacity, uint32_t* tiles_touched, uint8_t* clamped) {
  const int nb = (P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  hipLaunchKernelGGL(export_geom_kernel, dim3(nb), dim3(GAUSS_BLOCK), 0, s, P, g, means2D, depths, rgb, conic_opacity,
                     tiles_touched, clamped);
  return hipGetLastError();
}
hipError_t launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* view, uint8_t* present) {
  const int nb = (P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  hipLaunchKernelGGL(mark_visible_kernel, dim3(nb), dim3(GAUSS_BLOCK), 0, s, P, means3D, view, present);
  return hipGetLastError();
}
hipError_t launch_sh_grad_compose(hipStream_t s, int P, int D, int M, int N, const float* means3D, const float* campos,
                                  const float* dL_drgb, float* dL_dsh) {
  const int nb = (P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  hipLaunchKernelGGL(sh_grad_compose_kernel, dim3(nb), dim3(GAUSS_BLOCK), 0, s, P, D, M, N, means3D, campos, dL_drgb, dL_dsh);
  return hipGetLastError();
}
hipError_t launch_touched_rows(hipStream_t s, int64_t P, int nt, const float* const* data, const int* row_len, uint8_t* mask) {
  RowSet rs;
  rs.n = nt;
  for (int t = 0; t < nt; ++t) {
    rs.data[t] = data[t];
    rs.row_len[t] = row_len[t];
  }
  hipLaunchKernelGGL(touched_rows_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, P, rs, mask);
  return hipGetLastError();
}
int64_t view_message_words_host(int64_t P, int64_t cap) { return view_message_words(P, cap); }
// message slices for the compaction (host-side carve of the same layout)
hipError_t launch_view_message_header(hipStream_t s, int64_t P, const float* campos, const uint32_t* block_off,
                                      const uint64_t* total, float* msg) {
  const int64_t nb = (P + VIEW_MSG_ROWS - 1) / VIEW_MSG_ROWS;
  const int64_t n = nb > 4 ? nb : 4;
  hipLaunchKernelGGL(view_message_header_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, P, campos, block_off, total, msg);
  return hipGetLastError();
}
hipError_t launch_view_messages_accumulate(hipStream_t s, int64_t P, int D, int M, int n_views, const float* messages,
                                           int64_t stride_words, int64_t cap, const float* means3D, float* const dense[6],
                                           uint8_t* row_valid) {
  DenseGrads d = {dense[0], dense[1], dense[2], dense[3], dense[4], dense[5]};
  const int64_t nblk = (P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  if (row_valid != nullptr)
    hipLaunchKernelGGL(view_messages_accumulate_kernel<true>, dim3((unsigned)nblk), dim3(GAUSS_BLOCK), 0, s, P, D, M, n_views,
                       messages, stride_words, cap, means3D, d, row_valid);
  else
    hipLaunchKernelGGL(view_messages_accumulate_kernel<false>, dim3((unsigned)nblk), dim3(GAUSS_BLOCK), 0, s, P, D, M, n_views,
                       messages, stride_words, cap, means3D, d, row_valid);
  return hipGetLastError();
}
hipError_t launch_preprocess_backward(hipStream_t s, const PreBwdArgs& a) {
  const int nb = (a.P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  if (a.row_state != nullptr) hipLaunchKernelGGL(preprocess_backward_kernel<true>, dim3(nb), dim3(GAUSS_BLOCK), 0, s, a);
  else hipLaunchKernelGGL(preprocess_backward_kernel<false>, dim3(nb), dim3(GAUSS_BLOCK), 0, s, a);
  return hipGetLastError();
}

}  // namespace gsr
