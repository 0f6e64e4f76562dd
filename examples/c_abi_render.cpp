// c_abi_render.cpp -- a host that uses libgsr_hip.so WITHOUT torch or Python: plain HIP allocations, the C ABI of
// include/gsr.h, one forward render.  It is what a compiled re-implementation of the reference's torch glue
// (DGR/rasterize_points.cu:35-95) reduces to, and it doubles as a test that no torch type hides behind the boundary
// (tests/test_gpu_parity.py::test_c_abi_host_without_torch compares its image bit for bit with the Python binding's).
//
// Build:  hipcc -std=c++17 -O2 -I include examples/c_abi_render.cpp -L gaussianeditor_amd -lgsr_hip \
//               -Wl,-rpath,'$ORIGIN/../gaussianeditor_amd' -o examples/c_abi_render
// Usage:  c_abi_render scene.bin out.bin
//   scene.bin: int32 P, D, M, W, H; float32 tanfovx, tanfovy, scale_modifier; then float32 arrays
//              bg[3], means3D[P*3], scales[P*3], rotations[P*4], opacities[P], shs[P*M*3], viewmatrix[16], projmatrix[16], campos[3]
//   out.bin:   int64 num_rendered; float32 color[3*H*W], depth[H*W]; int32 radii[P]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gsr.h"

#define HIP_OK(x)                                                          \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));         \
      return 2;                                                            \
    }                                                                      \
  } while (0)
#define GSR_OK_(x)                                                         \
  do {                                                                     \
    int s_ = (x);                                                          \
    if (s_ != GSR_OK) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, gsr_status_string(s_));         \
      return 3;                                                            \
    }                                                                      \
  } while (0)

template <typename T>
static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (h.empty()) return nullptr;
  if (hipMalloc(&d, sizeof(T) * h.size()) != hipSuccess) return nullptr;
  (void)hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]);
    return 1;
  }
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t hdr[5];
  float fl[3];
  if (std::fread(hdr, 4, 5, f) != 5 || std::fread(fl, 4, 3, f) != 3) return 1;
  const int P = hdr[0], D = hdr[1], M = hdr[2], W = hdr[3], H = hdr[4];
  auto rd = [&](size_t n) {
    std::vector<float> v(n);
    if (std::fread(v.data(), 4, n, f) != n) std::exit(1);
    return v;
  };
  const auto bg = rd(3), means = rd((size_t)P * 3), scales = rd((size_t)P * 3), rots = rd((size_t)P * 4), opac = rd(P),
             shs = rd((size_t)P * M * 3), view = rd(16), proj = rd(16), campos = rd(3);
  std::fclose(f);

  if (gsr_abi_version() != GSR_ABI_VERSION) return 4;
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  float *d_bg = to_device(bg), *d_means = to_device(means), *d_scales = to_device(scales), *d_rots = to_device(rots),
        *d_opac = to_device(opac), *d_shs = to_device(shs), *d_view = to_device(view), *d_proj = to_device(proj),
        *d_campos = to_device(campos);
  size_t sizes[3];
  GSR_OK_(gsr_scratch_sizes(P, 0, 0, W, H, sizes));
  void *geom = nullptr, *image = nullptr, *binning = nullptr;
  int32_t* d_radii = nullptr;
  float *d_color = nullptr, *d_depth = nullptr;
  HIP_OK(hipMalloc(&geom, sizes[0]));
  HIP_OK(hipMalloc(&image, sizes[2]));
  HIP_OK(hipMalloc(&d_radii, sizeof(int32_t) * (size_t)P));
  HIP_OK(hipMalloc(&d_color, sizeof(float) * 3 * (size_t)W * H));
  HIP_OK(hipMalloc(&d_depth, sizeof(float) * (size_t)W * H));

  int64_t counts[2] = {0, 0};  // the one blocking readback: {tile instances, tile-group instances} size the binning scratch
  GSR_OK_(gsr_preprocess(stream, P, D, M, d_means, d_scales, fl[2], d_rots, d_opac, d_shs, nullptr, nullptr, d_view, d_proj,
                         d_campos, W, H, fl[0], fl[1], 0, 0, /*flags=*/0u, d_radii, geom, counts));
  const int64_t R = counts[0], G = counts[1];
  GSR_OK_(gsr_scratch_sizes(P, R, G, W, H, sizes));
  if (sizes[1]) HIP_OK(hipMalloc(&binning, sizes[1]));
  GSR_OK_(gsr_bin(stream, P, R, G, W, H, geom, binning, image));
  GSR_OK_(gsr_blend_forward(stream, P, R, W, H, d_bg, geom, binning, image, d_color, d_depth, /*flags=*/0u));
  HIP_OK(hipStreamSynchronize(stream));

  std::vector<float> color(3 * (size_t)W * H), depth((size_t)W * H);
  std::vector<int32_t> radii(P);
  HIP_OK(hipMemcpy(color.data(), d_color, sizeof(float) * color.size(), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(depth.data(), d_depth, sizeof(float) * depth.size(), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(radii.data(), d_radii, sizeof(int32_t) * radii.size(), hipMemcpyDeviceToHost));
  std::FILE* o = std::fopen(argv[2], "wb");
  if (!o) return 1;
  std::fwrite(&R, 8, 1, o);
  std::fwrite(color.data(), 4, color.size(), o);
  std::fwrite(depth.data(), 4, depth.size(), o);
  std::fwrite(radii.data(), 4, radii.size(), o);
  std::fclose(o);
  std::printf("rendered %dx%d, %d Gaussians, %lld instances\n", W, H, P, (long long)R);
  return 0;
}
