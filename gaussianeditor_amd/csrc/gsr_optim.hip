// gsr_optim.hip -- fused, row-masked Adam step over all parameter groups of a Gaussian model in ONE launch
// (SURVEY.md section 8(f) rank 3).
//
// The reference trains with torch.optim.Adam(l, lr=0.0, eps=1e-15) over six parameter groups
// (gaussiansplatting/scene/gaussian_model.py:336-380), masks the gradients of five of them row-wise with tensor hooks
// (apply_grad_mask, :841-856: grad * mask[:, None]) and adds an anchor (MSE-to-snapshot) loss whose gradient is
// 2 w_row (p - anchor) / N (anchor_loss, :152-184).  Per step that is a dozen elementwise passes over 59 scalars per
// Gaussian at M = 16.  Here every scalar is read and written exactly once: 16 B in (p, g, m, v; + 4 B anchor when
// used), 12 B out = the 28 B per scalar of SURVEY.md section 8(f) -- 1.65 GB per step at 1 M Gaussians, an HBM stream.
//
// Arithmetic = torch's single-tensor Adam (torch/optim/adam.py, _single_tensor_adam, no amsgrad / weight decay /
// maximize), one IEEE binary32 operation per line, scalars prepared on the host in double like torch does:
//   m  = m + (1 - beta1) * (g - m)                    exp_avg.lerp_(grad, 1 - beta1)
//   v  = v * beta2 + ((1 - beta2) * g) * g            exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
//   d  = sqrt(v) / sqrt(1 - beta2^t) + eps            (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
//   p  = p + (-(lr / (1 - beta1^t))) * (m / d)        param.addcdiv_(exp_avg, denom, value=-step_size)
// with g = mask[row] ? grad (+ anchor_scale * w[row] * (p - anchor)) : 0 for masked groups.
#include <math.h>
#include <string.h>

#include "gsr_kernels.h"

namespace gsr {

constexpr int ADAM_MAX_TENSORS = 8;
constexpr int ADAM_THREADS = 256;
constexpr int ADAM_PER_THREAD = 4;
constexpr int ADAM_PER_BLOCK = ADAM_THREADS * ADAM_PER_THREAD;

struct AdamTensorDev {
  float* p;
  const float* g;
  float* m;
  float* v;
  const float* anchor;
  long long n;
  int row_len;
  int masked;
  float neg_step_size;  // -(lr / bias_correction1)
  float anchor_scale;
  int vec4;             // all pointers 16-byte aligned
};
struct AdamArgs {
  AdamTensorDev t[ADAM_MAX_TENSORS];
  unsigned block_start[ADAM_MAX_TENSORS + 1];
  int nt;
  const uint8_t* row_mask;
  const float* row_weight;
  const uint8_t* grad_valid;  // null, or per row: 0 = the gradient row was not written, take zeros and do not read it
  float beta2, one_minus_b1, one_minus_b2, bc2_sqrt, eps;
};

__device__ __forceinline__ void adam_scalar(const AdamArgs& a, const AdamTensorDev& t, long long e, float& p, float g,
                                            float& m, float& v) {
  if (t.masked || t.anchor != nullptr) {
    const long long row = e / t.row_len;
    if (t.anchor != nullptr) {
      const float w = a.row_weight != nullptr ? a.row_weight[row] : 1.0f;
      g = g + (t.anchor_scale * w) * (p - t.anchor[e]);
    }
    if (t.masked && a.row_mask != nullptr && a.row_mask[row] == 0) g = 0.0f;
  }
  m = m + a.one_minus_b1 * (g - m);
  v = v * a.beta2 + (a.one_minus_b2 * g) * g;
  const float d = sqrtf(v) / a.bc2_sqrt + a.eps;
  p = p + t.neg_step_size * (m / d);
}

__global__ void __launch_bounds__(ADAM_THREADS) adam_step_kernel(const AdamArgs a) {
  int ti = 0;
#pragma unroll
  for (int i = 1; i < ADAM_MAX_TENSORS; ++i)
    if (i < a.nt && blockIdx.x >= a.block_start[i]) ti = i;
  const AdamTensorDev& t = a.t[ti];
  const long long e0 = ((long long)(blockIdx.x - a.block_start[ti]) * ADAM_THREADS + threadIdx.x) * ADAM_PER_THREAD;
  if (e0 >= t.n) return;
  if (t.vec4 && e0 + ADAM_PER_THREAD <= t.n) {
    float4 p = *reinterpret_cast<const float4*>(t.p + e0);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.grad_valid == nullptr) {
      g = *reinterpret_cast<const float4*>(t.g + e0);
    } else {  // gradient rows that were never written (gsr_view_messages_accumulate_rows) are zeros and are not read
      const bool v0 = a.grad_valid[e0 / t.row_len] != 0, v1 = a.grad_valid[(e0 + 1) / t.row_len] != 0;
      const bool v2 = a.grad_valid[(e0 + 2) / t.row_len] != 0, v3 = a.grad_valid[(e0 + 3) / t.row_len] != 0;
      if (v0 || v1 || v2 || v3) {
        const float4 x = *reinterpret_cast<const float4*>(t.g + e0);
        g = make_float4(v0 ? x.x : 0.f, v1 ? x.y : 0.f, v2 ? x.z : 0.f, v3 ? x.w : 0.f);
      }
    }
    float4 m = *reinterpret_cast<const float4*>(t.m + e0);
    float4 v = *reinterpret_cast<const float4*>(t.v + e0);
    adam_scalar(a, t, e0, p.x, g.x, m.x, v.x);
    adam_scalar(a, t, e0 + 1, p.y, g.y, m.y, v.y);
    adam_scalar(a, t, e0 + 2, p.z, g.z, m.z, v.z);
    adam_scalar(a, t, e0 + 3, p.w, g.w, m.w, v.w);
    *reinterpret_cast<float4*>(t.p + e0) = p;
    *reinterpret_cast<float4*>(t.m + e0) = m;
    *reinterpret_cast<float4*>(t.v + e0) = v;
  } else {
    for (long long e = e0; e < min(e0 + (long long)ADAM_PER_THREAD, t.n); ++e) {
      float p = t.p[e], m = t.m[e], v = t.v[e];
      const float ge = (a.grad_valid == nullptr || a.grad_valid[e / t.row_len] != 0) ? t.g[e] : 0.f;
      adam_scalar(a, t, e, p, ge, m, v);
      t.p[e] = p;
      t.m[e] = m;
      t.v[e] = v;
    }
  }
}

hipError_t launch_adam_step(hipStream_t s, int nt, const gsr_adam_tensor* tensors, long long step, double beta1,
                            double beta2, double eps, const uint8_t* row_mask, const float* row_weight,
                            const uint8_t* grad_valid) {
  if (nt <= 0) return hipSuccess;
  if (nt > ADAM_MAX_TENSORS) return hipErrorInvalidValue;
  AdamArgs a;
  memset(&a, 0, sizeof(a));
  a.nt = nt;
  a.row_mask = row_mask;
  a.row_weight = row_weight;
  a.grad_valid = grad_valid;
  // scalars in double, as torch computes them from Python floats (torch/optim/adam.py: bias_correction1/2, step_size)
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  a.beta2 = (float)beta2;
  a.one_minus_b1 = (float)(1.0 - beta1);
  a.one_minus_b2 = (float)(1.0 - beta2);
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = (float)eps;
  unsigned blocks = 0;
  for (int i = 0; i < nt; ++i) {
    const gsr_adam_tensor& h = tensors[i];
    AdamTensorDev& d = a.t[i];
    d.p = h.param; d.g = h.grad; d.m = h.exp_avg; d.v = h.exp_avg_sq; d.anchor = h.anchor;
    d.n = h.numel;
    d.row_len = h.row_len > 0 ? h.row_len : 1;
    d.masked = h.masked;
    d.neg_step_size = (float)(-(h.lr / bc1));
    d.anchor_scale = h.anchor_scale;
    const uintptr_t bits = (uintptr_t)h.param | (uintptr_t)h.grad | (uintptr_t)h.exp_avg | (uintptr_t)h.exp_avg_sq;
    d.vec4 = (bits & 15u) == 0 ? 1 : 0;
    a.block_start[i] = blocks;
    blocks += (unsigned)((h.numel + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK);
  }
  a.block_start[nt] = blocks;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(ADAM_THREADS), 0, s, a);
  return hipGetLastError();
}

}  // namespace gsr
