// ref_driver.hip -- TEST INFRASTRUCTURE.  A C interface around the REFERENCE's own rasterizer
// (CudaRasterizer::Rasterizer, DGR/cuda_rasterizer/rasterizer.h:22-113), whose unmodified .cu files
// are compiled for gfx950 by the Makefile next to this file.  It plays the role of the reference's
// torch glue (DGR/rasterize_points.cu) without torch: device pointers in, device pointers out, and
// the three scratch chunks are kept in a context so that tests can read every intermediate
// (GeometryState / BinningState / ImageState, rasterizer_impl.h:30-61) and run the backward.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>

#include "rasterizer.h"
#include "rasterizer_impl.h"

namespace {
struct Chunk {
  char* ptr = nullptr;
  size_t cap = 0;
  char* get(size_t n) {
    if (n > cap) {
      if (ptr) (void)hipFree(ptr);
      if (hipMalloc(&ptr, n) != hipSuccess) throw std::runtime_error("hipMalloc failed");
      cap = n;
    }
    return ptr;
  }
  ~Chunk() {
    if (ptr) (void)hipFree(ptr);
  }
};
struct Ctx {
  Chunk geom, binning, img;
  int P = 0, R = 0, W = 0, H = 0;
};
template <typename T>
void d2d(T* dst, const T* src, size_t n) {
  if (dst && n) (void)hipMemcpy(dst, src, sizeof(T) * n, hipMemcpyDeviceToDevice);
}
}  // namespace

extern "C" {

void* gsrref_create() { return new Ctx(); }
void gsrref_destroy(void* h) { delete static_cast<Ctx*>(h); }

// Rasterizer::forward, rasterizer_impl.cu:179-285.  Returns num_rendered, or -1 on exception.
int gsrref_forward(void* h, int P, int D, int M, const float* bg, int W, int H, const float* means3D, const float* shs,
                   const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                   const float* rotations, const float* cov3D_precomp, const float* view, const float* proj,
                   const float* campos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                   float* out_depth, int* radii) {
  Ctx* c = static_cast<Ctx*>(h);
  try {
    std::function<char*(size_t)> g = [c](size_t n) { return c->geom.get(n); };
    std::function<char*(size_t)> b = [c](size_t n) { return c->binning.get(n); };
    std::function<char*(size_t)> i = [c](size_t n) { return c->img.get(n); };
    (void)hipMemset(out_color, 0, sizeof(float) * 3 * (size_t)W * H);  // torch::full(0), rasterize_points.cu:57-58
    (void)hipMemset(out_depth, 0, sizeof(float) * (size_t)W * H);
    (void)hipMemset(radii, 0, sizeof(int) * (size_t)P);
    c->P = P; c->W = W; c->H = H;
    c->R = CudaRasterizer::Rasterizer::forward(g, b, i, P, D, M, bg, W, H, means3D, shs, colors_precomp, opacities, scales,
                                               scale_modifier, rotations, cov3D_precomp, view, proj, campos, tan_fovx,
                                               tan_fovy, prefiltered != 0, out_color, out_depth, radii, false);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return c->R;
  } catch (...) {
    return -1;
  }
}

// Copy the reference's intermediate state out of its chunks (any destination may be NULL).
int gsrref_export(void* h, float* means2D, float* depths, float* cov3D, float* rgb, float* conic_opacity,
                  uint32_t* tiles_touched, uint8_t* clamped, uint32_t* point_offsets, uint64_t* keys_sorted,
                  uint32_t* point_list, uint32_t* ranges, float* final_T, uint32_t* n_contrib) {
  Ctx* c = static_cast<Ctx*>(h);
  const size_t P = c->P, R = c->R, N = (size_t)c->W * c->H;
  const size_t T = (size_t)((c->W + 15) / 16) * ((c->H + 15) / 16);
  char* gp = c->geom.ptr;
  CudaRasterizer::GeometryState gs = CudaRasterizer::GeometryState::fromChunk(gp, P);
  d2d(means2D, (const float*)gs.means2D, 2 * P);
  d2d(depths, (const float*)gs.depths, P);
  d2d(cov3D, (const float*)gs.cov3D, 6 * P);
  d2d(rgb, (const float*)gs.rgb, 3 * P);
  d2d(conic_opacity, (const float*)gs.conic_opacity, 4 * P);
  d2d(tiles_touched, (const uint32_t*)gs.tiles_touched, P);
  d2d(clamped, (const uint8_t*)gs.clamped, 3 * P);
  d2d(point_offsets, (const uint32_t*)gs.point_offsets, P);
  if (R > 0) {
    char* bp = c->binning.ptr;
    CudaRasterizer::BinningState bs = CudaRasterizer::BinningState::fromChunk(bp, R);
    d2d(keys_sorted, (const uint64_t*)bs.point_list_keys, R);
    d2d(point_list, (const uint32_t*)bs.point_list, R);
  }
  char* ip = c->img.ptr;
  CudaRasterizer::ImageState is = CudaRasterizer::ImageState::fromChunk(ip, N);
  d2d(ranges, (const uint32_t*)is.ranges, 2 * T);
  d2d(final_T, (const float*)is.accum_alpha, N);
  d2d(n_contrib, (const uint32_t*)is.n_contrib, N);
  return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}

// Rasterizer::backward, rasterizer_impl.cu:289-341.  All nine gradient buffers are zeroed here as
// the reference's glue does (rasterize_points.cu:120-128).
int gsrref_backward(void* h, int D, int M, const float* bg, const float* means3D, const float* shs,
                    const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                    const float* cov3D_precomp, const float* view, const float* proj, const float* campos, float tan_fovx,
                    float tan_fovy, const int* radii, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                    float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                    float* dL_dscale, float* dL_drot) {
  Ctx* c = static_cast<Ctx*>(h);
  const size_t P = c->P;
  try {
    (void)hipMemset(dL_dmean2D, 0, sizeof(float) * 3 * P);
    (void)hipMemset(dL_dconic, 0, sizeof(float) * 4 * P);
    (void)hipMemset(dL_dopacity, 0, sizeof(float) * P);
    (void)hipMemset(dL_dcolor, 0, sizeof(float) * 3 * P);
    (void)hipMemset(dL_dmean3D, 0, sizeof(float) * 3 * P);
    (void)hipMemset(dL_dcov3D, 0, sizeof(float) * 6 * P);
    if (dL_dsh) (void)hipMemset(dL_dsh, 0, sizeof(float) * 3 * (size_t)M * P);
    (void)hipMemset(dL_dscale, 0, sizeof(float) * 3 * P);
    (void)hipMemset(dL_drot, 0, sizeof(float) * 4 * P);
    CudaRasterizer::Rasterizer::backward(c->P, D, M, c->R, bg, c->W, c->H, means3D, shs, colors_precomp, scales,
                                         scale_modifier, rotations, cov3D_precomp, view, proj, campos, tan_fovx, tan_fovy,
                                         radii, c->geom.ptr, c->binning.ptr, c->img.ptr, dL_dpix, dL_dmean2D, dL_dconic,
                                         dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, false);
    return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
  } catch (...) {
    return -1;
  }
}

// Rasterizer::apply_weights, rasterizer_impl.cu:343-447 (weights, cnt accumulated in place).
int gsrref_apply_weights(void* h, int P, int W, int H, const float* means3D, float* weights, const float* opacities,
                         const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* view, const float* proj, const float* campos, float tan_fovx, float tan_fovy,
                         const float* image_weights, int* radii, int* cnt, int num_channels) {
  Ctx* c = static_cast<Ctx*>(h);
  try {
    std::function<char*(size_t)> g = [c](size_t n) { return c->geom.get(n); };
    std::function<char*(size_t)> b = [c](size_t n) { return c->binning.get(n); };
    std::function<char*(size_t)> i = [c](size_t n) { return c->img.get(n); };
    float bg[3] = {0, 0, 0};
    float* dbg = nullptr;
    (void)hipMalloc(&dbg, sizeof(bg));
    (void)hipMemcpy(dbg, bg, sizeof(bg), hipMemcpyHostToDevice);
    (void)hipMemset(radii, 0, sizeof(int) * (size_t)P);
    CudaRasterizer::Rasterizer::apply_weights(g, b, i, P, 0, 0, dbg, W, H, means3D, nullptr, weights, opacities, scales,
                                              scale_modifier, rotations, cov3D_precomp, view, proj, campos, tan_fovx,
                                              tan_fovy, false, image_weights, radii, cnt, num_channels, false);
    const bool ok = hipDeviceSynchronize() == hipSuccess;
    (void)hipFree(dbg);
    return ok ? 0 : -1;
  } catch (...) {
    return -1;
  }
}

// Rasterizer::markVisible, rasterizer_impl.cu:128-133.
int gsrref_mark_visible(int P, float* means3D, float* view, float* proj, bool* present) {
  CudaRasterizer::Rasterizer::markVisible(P, means3D, view, proj, present);
  return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}

}  // extern "C"
