// gsr_blend.hip -- the per-tile alpha-compositing kernels: K6 (forward), K7 (backward),
// K12 (semantic tracing / apply_weights).
//
// CDNA4 mapping.  The reference runs one 256-thread block per 16x16 tile (8 warps of 32)
// with block-wide barriers around a shared-memory staging buffer.  Here the unit of work is
// ONE WAVE64 = one 8x8 pixel quadrant of a tile, launched as a single-wave workgroup:
//   * the forward and the tracing kernel have no workgroup barrier (a one-wave workgroup's
//     s_barrier is free), each quadrant terminates as soon as ITS 64 pixels are opaque,
//     and the CU's wave slots are refilled at wave granularity;
//   * 8x8 is the most compact 64-pixel footprint, so the exec mask stays coherent and the
//     "no lane contributes" early-outs (scalar branches on a 64-bit ballot) fire often;
//   * each wave stages 64 sorted instances at a time in LDS (3 x float4 per instance, read
//     back as uniform-address ds_read_b128 broadcasts);
//   * the kernels are PERSISTENT: a fixed number of single-wave workgroups (a few per SIMD)
//     pull (tile, quadrant) items from eight per-XCD queues (one relaxed agent-scope atomic
//     per item).  Tiles are queued longest-list-first (gsr_binning.hip: tile_worklist_kernel)
//     and dealt round-robin to the XCD queues, so every XCD gets the same amount of work, the
//     heavy quadrants start first and the light ones fill the gaps (the instance lists are
//     heavy-tailed: with one workgroup per quadrant most SIMDs idle through a long tail);
//     the four quadrants of a tile are consecutive items of ONE queue, so they gather the same
//     Gaussians through the same 4 MiB L2.  Which wave runs an item never affects results;
//   * the backward runs one 4-wave workgroup per tile: the quadrant waves reduce the 9 per-Gaussian gradient terms of
//     four entries at a time across their 64 lanes (two v_permlane32_swap, one v_permlane16_swap, four row_shr DPP adds),
//     add their totals into a per-chunk LDS accumulator (double-buffered by chunk parity: one workgroup barrier per chunk),
//     and ONE global atomic per (tile, instance, term) leaves the CU -- the reference issues one per pixel.
#include <stdlib.h>

#include <type_traits>

#include "gsr_kernels.h"

namespace gsr {

struct PixelWave {
  int tile, px, py;
  bool inside;
};

// (tile, quadrant) -> this lane's pixel.  Returns false if the quadrant lies outside the image.
__device__ __forceinline__ bool setup_wave(const BlendArgs& a, uint32_t tile, uint32_t quad, PixelWave& pw) {
  const int tx = (int)(tile % (uint32_t)a.gx), ty = (int)(tile / (uint32_t)a.gx);
  const int lane = lane_id();
  pw.tile = (int)tile;
  pw.px = tx * TILE + (int)(quad & 1u) * QUAD + (lane & 7);
  pw.py = ty * TILE + (int)(quad >> 1) * QUAD + (lane >> 3);
  pw.inside = pw.px < a.W && pw.py < a.H;
  return __any(pw.inside) != 0;
}

// The pixels of one forward / trace item and its pixel rectangle (the cull test of the blend loops starts from the pixels
// that are live, live_pixel_box, which lie inside it).  `code` = quad | sub << 2 from
// run_work_queue.  The item covers its quadrant, or -- when the image is small (SPLIT, image-wide) -- rows [4 s, 4 s + 4) /
// the 4x4 block s of it: lanes outside the part are switched off (pw.inside), the box shrinks with it.  Returns false if
// no pixel is inside the image.
struct ItemBox {
  float qx0, qy0, qw, qh;
};
template <int SPLIT>
__device__ __forceinline__ bool setup_item(const BlendArgs& a, uint32_t tile, uint32_t code, PixelWave& pw, ItemBox& b) {
  constexpr uint32_t split = (uint32_t)SPLIT;
  const uint32_t sub = (code >> 2) & 3u;
  if (!setup_wave(a, tile, code & 3u, pw)) return false;
  const int lane = lane_id();
  b.qx0 = (float)(pw.px - (lane & 7));
  b.qy0 = (float)(pw.py - (lane >> 3));
  b.qw = b.qh = (float)(QUAD - 1);
  if (split == 2u) {  // rows [4 sub, 4 sub + 4)
    pw.inside = pw.inside && ((uint32_t)(lane >> 5) == sub);
    b.qy0 += 4.0f * (float)sub;
    b.qh = 3.0f;
  } else if (split == 4u) {  // the 4x4 block (sub & 1, sub >> 1)
    pw.inside = pw.inside && ((uint32_t)((lane >> 2) & 1) == (sub & 1u)) && ((uint32_t)(lane >> 5) == (sub >> 1));
    b.qx0 += 4.0f * (float)(sub & 1u);
    b.qy0 += 4.0f * (float)(sub >> 1);
    b.qw = b.qh = 3.0f;
  }
  return true;
}

// Persistent work loop.  Queue x (one per XCD) owns entries x, x+8, x+16, ... of work_order;
// its cursor counts quadrant items (4 per non-empty tile) and then, if `with_empty`, one item
// per empty tile.  A wave serves the queue of the XCD it happens to run on (HW_REG_XCC_ID, a
// placement hint only: any wave may run any item).
// Called by ONE thread of every workgroup after its last (failed) pop: the workgroup that retires last puts the
// cursors of this queue kind back to zero, so the next launch of the kind finds them cleared without a memset.
// (Every pop of every other workgroup has returned before that workgroup's retire increment is issued.)
// Two levels, one counter per 64 workgroups and one on top: memory-side atomics on ONE line serialise at ~6 ns
// each, which for 4096 workgroups retiring together measured 20 us on the tail of the forward kernel.
__device__ __forceinline__ void retire_queue(uint32_t* queue) {
  const uint32_t nwg = gridDim.x, grp = blockIdx.x >> 6, ngrp = (nwg + 63u) >> 6;
  if (ngrp > (uint32_t)QUEUE_GROUPS) return;  // (launchers fall back to a memset for such grids)
  const uint32_t gsize = min(64u, nwg - (grp << 6));
  uint32_t* top = queue + 8 * QUEUE_STRIDE;
  uint32_t* gcnt = queue + 9 * QUEUE_STRIDE;
  if (__hip_atomic_fetch_add(gcnt + grp * QUEUE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gsize - 1u) return;
  if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ngrp - 1u) return;
#pragma unroll
  for (int x = 0; x < 8; ++x) __hip_atomic_store(queue + x * QUEUE_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t i = 0; i < ngrp; ++i)
    __hip_atomic_store(gcnt + i * QUEUE_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Work distribution.  The persistent workgroups are placed deterministically (measured with the per-wave profile,
// tools/wave_profile.py): with `units` = number of SIMDs (single-wave workgroups) or CUs (4-wave workgroups), block b
// runs on unit b % units in residency slot b / units, and on XCD b % 8.  A kernel ends when its busiest SIMD does, and
// with a plain longest-first queue every wave's FIRST item is one of the few very long lists (the per-item work is
// extremely skewed: mean 80 evaluated entries per quadrant, maximum 800), so the four long items that happen to share
// a SIMD decide the run time: the busiest SIMD carried 1.83x the mean load.
// Hence the first item of every workgroup is assigned, not popped: slot 0 of unit u takes item u of its XCD's queue,
// slot 1 item 2n-1-u, slot 2 item 2n+u, slot 3 item 4n-1-u, ... (n = units per XCD): every unit gets one item from
// each stratum of the sorted list, folded so that the sums even out.  Everything after the first W n items is popped
// dynamically.  The assignment is a permutation of [0, W n) whatever the real placement is, so results never depend on it.
__device__ __forceinline__ uint32_t first_item_of_block(uint32_t units, uint32_t& queue_x, uint32_t& first_dynamic) {
  if (units == 0u) {  // no assignment: everything is popped
    queue_x = 0u;
    first_dynamic = 0u;
    return 0xffffffffu;
  }
  const uint32_t n = units / 8u, W = gridDim.x / units;  // launchers guarantee units % 8 == 0, gridDim.x % units == 0
  const uint32_t b = blockIdx.x, slot = b / units, u = (b % units) / 8u;
  queue_x = b % 8u;
  first_dynamic = W * n;
  return (slot & 1u) ? (slot + 1u) * n - 1u - u : slot * n + u;
}

template <int SPLIT, class F>
__device__ __forceinline__ void run_work_queue(const BlendArgs& a, bool with_empty, F&& item) {
  const uint32_t nwork = a.work_meta[0];
  const uint32_t T = (uint32_t)(a.gx * a.gy);
  const uint32_t nempty = with_empty ? T - nwork : 0u;
  // queue x (one per XCD) owns entries x, x+8, ... of work_order: 4 quadrant items per non-empty tile, then one item
  // per empty tile
  // Granularity: a quadrant is one item, unless that leaves fewer than two items per persistent wave (small images,
  // e.g. the editor's 512x512 views: 1024 tiles for 4096 waves): then every quadrant is cut into 2 (8x4 pixels) or 4
  // (4x4) items, which trades lane utilisation the chip is not using anyway for shorter serial chains and finer
  // culling.  The item code passed on is quad | sub << 2, decoded by the item function.
  constexpr uint32_t split = (uint32_t)SPLIT;  // chosen by the launcher from the number of tiles of the image
  constexpr uint32_t per_tile = 4u * split;
  auto run_item = [&](uint32_t x, uint32_t q) __attribute__((always_inline)) -> bool {
    const uint32_t n_x = nwork > x ? (nwork - x + 7u) / 8u : 0u;
    const uint32_t e_x = nempty > x ? (nempty - x + 7u) / 8u : 0u;
    if (q >= per_tile * n_x + e_x) return false;
    if (q < per_tile * n_x) {
      const uint32_t t = q / per_tile, r = q - t * per_tile;  // r = quad * split + sub
      item(a.work_order[x + 8u * t], split == 1u ? r : ((r / split) | ((r % split) << 2)), false);
    } else {
      item(a.work_order[nwork + x + 8u * (q - per_tile * n_x)], 0u, true);
    }
    return true;
  };
  uint32_t x0, base;
  const uint32_t q0 = first_item_of_block((uint32_t)a.units, x0, base);
  (void)run_item(x0, q0);
  // then serve the queue of the XCD the wave really runs on (HW_REG_XCC_ID; equal to x0 on this chip)
  const uint32_t x = (uint32_t)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;
  uint32_t* head = a.queue + x * QUEUE_STRIDE;
  for (;;) {
    uint32_t q = 0;
    if (lane_id() == 0) q = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
    if (!run_item(x, base + q)) break;
  }
  if (a.self_reset && lane_id() == 0) retire_queue(a.queue);
}

// Read-only tables of an earlier kernel, read through the SCALAR cache at a wave-uniform index (s_load: its counter,
// lgkmcnt, is independent of the vector memory counter, so picking such a value up never waits for stores or gathers).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t scalar_load(const uint32_t* p, uint32_t i) {
  return ((__attribute__((address_space(4))) const uint32_t*)(uintptr_t)p)[i];
}
__device__ __forceinline__ uint2 scalar_load(const uint2* p, uint32_t i) {
  const u32x2 v = ((__attribute__((address_space(4))) const u32x2*)(uintptr_t)p)[i];
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint4 scalar_load(const uint4* p, uint32_t i) {
  const u32x4 v = ((__attribute__((address_space(4))) const u32x4*)(uintptr_t)p)[i];
  return make_uint4(v.x, v.y, v.z, v.w);
}

// max over the 64 lanes of a wave, in every lane: 6 v_max_u32_dpp (row_shr 1, 2, 4, 8, row_bcast 15 / 31) + one v_readlane
__device__ __forceinline__ uint32_t wave_max_u32_dpp(uint32_t v) {
  auto step = [](uint32_t w, auto ctrl, auto rows) __attribute__((always_inline)) {
    return max(w, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, decltype(ctrl)::value, decltype(rows)::value, 0xf, false));
  };
  v = step(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
  v = step(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
  v = step(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
  v = step(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
  v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
  v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

constexpr int GROUP = 4;  // survivors processed per inner-loop iteration
#ifndef GSR_FWD_SELECT_CHAIN
#define GSR_FWD_SELECT_CHAIN 1  // 0: the masked (exec-region) serial part for every chunk (A/B builds)
#endif
// T / (1 - alpha) of the backward's transmittance chain (backward.cu:503): 0 = T * v_rcp_f32 (1 ulp), 2 = the reciprocal
// refined by one Newton step (the product since round 5: along a list of thousands of translucent entries the raw
// reciprocal's error accumulated to 1e-5 of a gradient, tools/fuzz_v2.py seeds 110 / 256), 1 = IEEE division (A/B builds)
#ifndef GSR_BWD_DIV
#define GSR_BWD_DIV 2
#endif
#ifndef GSR_BWD_HYBRID_EXP
#define GSR_BWD_HYBRID_EXP 0  // 1 (A/B builds, round 6): hardware 2^x + a rare polynomial fallback next to the alpha threshold -- 34 VALU
                              // instructions fewer per group and NOT faster: the replay of the group body (tools/microbench/k7_group_replay.py)
                              // takes 835 cycles per group and SIMD at 4 waves either way (a v_exp_f32 costs a SIMD ~7.6 cycles), K7 219.6 vs
                              // 216.9-217.4 us same box (profiles/r06_a_k7_limiter.md)
#endif
#ifndef GSR_BWD_SLOT_REG
#define GSR_BWD_SLOT_REG 0
#endif
#ifndef GSR_BWD_DEFER_FLUSH
#define GSR_BWD_DEFER_FLUSH 1  // 0: round 4's loop -- flush at the end of its own chunk, two barriers per chunk (A/B builds)
#endif

// Sums each of four per-lane values over the 64 lanes of the wave, 10 instructions for all four
// instead of 4 x 6: two v_permlane32_swap + adds fold the half-waves (a,c | b,d), one
// v_permlane16_swap + add folds row pairs so that row r of the wave holds 16 partial sums of
// value r, then 4 row_shr DPP adds finish each row.  The total of value r ends up in lane 16 r + 15.
__device__ __forceinline__ float wave_sum4_to_rows(float a, float b, float c, float d) {
  unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
  unsigned uc = __builtin_bit_cast(unsigned, c), ud = __builtin_bit_cast(unsigned, d);
  {
    auto r = __builtin_amdgcn_permlane32_swap(ua, uc, false, false);  // ua = [a_lo|c_lo], uc = [a_hi|c_hi]
    ua = r[0];
    uc = r[1];
  }
  {
    auto r = __builtin_amdgcn_permlane32_swap(ub, ud, false, false);
    ub = r[0];
    ud = r[1];
  }
  const float x = __builtin_bit_cast(float, ua) + __builtin_bit_cast(float, uc);  // lanes 0-31: a, 32-63: c
  const float y = __builtin_bit_cast(float, ub) + __builtin_bit_cast(float, ud);  // lanes 0-31: b, 32-63: d
  unsigned ux = __builtin_bit_cast(unsigned, x), uy = __builtin_bit_cast(unsigned, y);
  {
    auto r = __builtin_amdgcn_permlane16_swap(ux, uy, false, false);  // ux rows [x0,y0,x2,y2], uy rows [x1,y1,x3,y3]
    ux = r[0];
    uy = r[1];
  }
  float z = __builtin_bit_cast(float, ux) + __builtin_bit_cast(float, uy);  // rows: a, b, c, d
  z = dpp_add<0x111>(z);
  z = dpp_add<0x112>(z);
  z = dpp_add<0x114>(z);
  z = dpp_add<0x118>(z);
  return z;
}

struct Entry {
  uint32_t id;
  float4 r0, r1, r2;
};

// Can the alpha >= 1/255 level set of an entry reach the pixel rectangle [qx0, qx0 + qw] x [qy0, qy0 + qh]?  Conservative:
// `false` only if every pixel of the rectangle would take `alpha < 1/255 -> continue` (forward.cu:340-344) anyway.
// The level set is the ellipse {d : d^T Q d <= tau2}, tau2 = 2 ln(255 o), d = pixel - mean.  The test is exact up to its
// margins: the minimum of the convex form d^T Q d over the rectangle is attained on one of the two sides that face the
// mean (or is 0 when the mean lies inside), so with (cx, cy) the mean clamped into the rectangle it is
//     min( min_x q(x, cy), min_y q(cx, y) ),   min_x q(x, cy) at x = clamp(-Q_xy cy / Q_xx)  (and likewise for y).
// A misplaced minimiser only makes the value larger by a second-order amount.  (Round 1/2 bounded the ellipse by its
// axis-aligned box instead; on the headline scene 14 % of the (quadrant, entry) pairs that passed that box are rejected
// here, tools/cull_study.py, and each rejected pair saves ~40 wave instructions in the forward and ~90 in the backward.)
// Rounding: det Q = Q_xx Q_zz - Q_xy^2 cancels for a needle that is not axis aligned, and both this function's and the
// blend loops' evaluation of the form carry errors of relative size ~1e-7 against terms that are up to 2/rho times larger
// than the result (rho = det Q / (Q_xx Q_zz)).  So the test is trusted only for rho >= 1e-3, where these stay below the
// margins applied (tau2 x 1.001 + 0.02 for the exponent the loops evaluate, x 1.001 again for the value computed here,
// the rectangle widened by 0.01 pixel); worse-conditioned entries are never culled (tools/cull_replay.py replays this
// arithmetic in binary32 against the reference's per-pixel evaluation).
__device__ __forceinline__ bool can_touch_quad(const float4& r0, const float4& r1, float qx0, float qy0,
                                               float qw = (float)(QUAD - 1), float qh = (float)(QUAD - 1)) {
  // r0 = conic.x, conic.y, conic.z, opacity; r1 = mean.x, mean.y, depth, radius
  const float o = r0.w;
  if (!(o >= 1.0f / 255.0f)) return !(o == o) ? true : false;  // o < 1/255: alpha < 1/255 everywhere (NaN: keep)
  const float A = r0.x, B = r0.y, C = r0.z;
  const float xz = A * C;
  const float det = xz - B * B;
  if (!(det >= 1e-3f * xz) || !(det > 0.0f) || !(A > 0.0f)) return true;  // ill-conditioned / degenerate / NaN: never cull
  const float tau2 = 2.0f * (__logf(255.0f * o) * 1.001f + 0.01f) * 1.001f;
  const float xl = (qx0 - 0.01f) - r1.x, xh = (qx0 + qw + 0.01f) - r1.x;
  const float yl = (qy0 - 0.01f) - r1.y, yh = (qy0 + qh + 0.01f) - r1.y;
  const float cx = __builtin_amdgcn_fmed3f(0.0f, xl, xh), cy = __builtin_amdgcn_fmed3f(0.0f, yl, yh);
  const float x1 = __builtin_amdgcn_fmed3f(-(B * cy) * __builtin_amdgcn_rcpf(A), xl, xh);
  const float y2 = __builtin_amdgcn_fmed3f(-(B * cx) * __builtin_amdgcn_rcpf(C), yl, yh);
  const float q1 = (A * x1) * x1 + ((2.0f * B) * x1) * cy + (C * cy) * cy;
  const float q2 = (A * cx) * cx + ((2.0f * B) * cx) * y2 + (C * y2) * y2;
  return !(q1 > tau2) || !(q2 > tau2);  // culled only if both candidates are outside (NaN anywhere: keep)
}

// The cull rectangle of a chunk: the bounding box of the pixels of the wave that can still use an entry (`live`: not
// saturated / not yet past their last contributor), in pixel coordinates, from the 8x8 lane grid whose origin is (ox, oy).
// A long item typically ends with a few stragglers among its 64 pixels; entries that cannot reach THEM are skipped by every
// lane that still matters (on the headline scene the twenty longest forward items evaluate a third fewer entries,
// tools/cull_study.py).  Scalar arithmetic on the ballot mask: row r of the grid is byte r of the mask.
struct LiveBox {
  float x0, y0, w, h;
};
__device__ __forceinline__ LiveBox live_pixel_box(uint64_t live_mask, float ox, float oy) {
  const uint32_t ymin = (uint32_t)__builtin_ctzll(live_mask) >> 3, ymax = (63u - (uint32_t)__builtin_clzll(live_mask)) >> 3;
  uint32_t c = (uint32_t)live_mask | (uint32_t)(live_mask >> 32);
  c |= c >> 16;
  c |= c >> 8;
  c &= 0xffu;
  const uint32_t xmin = (uint32_t)__builtin_ctz(c), xmax = 31u - (uint32_t)__builtin_clz(c);
  return {ox + (float)xmin, oy + (float)ymin, (float)(xmax - xmin), (float)(ymax - ymin)};
}

// exp(power) of the blend loops: the exactly specified polynomial (bit-identical to the CPU oracle), or -- per-call
// opt-in GSR_FLAG_FAST_EXP -- the hardware's 2^x on power * log2(e) (one multiply + one quarter-rate v_exp_f32 instead
// of 12 full-rate instructions).
template <bool FAST>
__device__ __forceinline__ float blend_exp(float power) {
  if (FAST) return __builtin_amdgcn_exp2f(power * 0x1.715476p+0f);
  return gsr_expf_noclamp(power);
}

// The same for two entries at once (packed binary32: v_pk_mul / v_pk_add / v_pk_fma_f32 round each half exactly as the
// scalar instructions do; v_rndne, v_cvt and v_ldexp have no packed form and run per half).
template <bool FAST>
__device__ __forceinline__ f32x2 blend_exp2(f32x2 power) {
  const f32x2 t = power * f32x2{0x1.715476p+0f, 0x1.715476p+0f};
  if (FAST) return f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  const f32x2 n = {__builtin_rintf(t.x), __builtin_rintf(t.y)};
  const f32x2 f = t - n;
  f32x2 p = {0x1.44138ap-13f, 0x1.44138ap-13f};
  p = __builtin_elementwise_fma(p, f, f32x2{0x1.5f0890p-10f, 0x1.5f0890p-10f});
  p = __builtin_elementwise_fma(p, f, f32x2{0x1.3b2a54p-7f, 0x1.3b2a54p-7f});
  p = __builtin_elementwise_fma(p, f, f32x2{0x1.c6af6cp-5f, 0x1.c6af6cp-5f});
  p = __builtin_elementwise_fma(p, f, f32x2{0x1.ebfbe0p-3f, 0x1.ebfbe0p-3f});
  p = __builtin_elementwise_fma(p, f, f32x2{0x1.62e430p-1f, 0x1.62e430p-1f});
  p = __builtin_elementwise_fma(p, f, f32x2{1.0f, 1.0f});
  return f32x2{__builtin_amdgcn_ldexpf(p.x, (int)n.x), __builtin_amdgcn_ldexpf(p.y, (int)n.y)};
}

// Walks list positions in chunks.  FORWARD: positions [0,len) ascending, lane l of chunk c holds
// position 64c + l.  Otherwise descending from `top`: lane l of chunk c holds top-1-(64c+l).
template <bool FORWARD, bool AUX = false>  // AUX: colours come from BlendArgs::colors3 (auxiliary forward render)
struct ChunkWalker {
  const BlendArgs& a;
  uint32_t list_base;  // range.x
  uint32_t count;      // number of positions to walk
  uint32_t id_next, id_next2;
  Entry cur, nxt;
  uint32_t chunk;  // index of the chunk held in `cur`

  __device__ __forceinline__ uint32_t pos_of(uint32_t k) const {  // k = 64*chunk + lane
    return FORWARD ? k : (count - 1 - k);
  }
  // All loads are UNCONDITIONAL (out-of-range lanes re-read the last entry): a predicated load would
  // make hipcc merge old and new register values right after the issue and wait for the data there,
  // which defeats the prefetch.
  __device__ __forceinline__ uint32_t load_id(uint32_t k) const {
    return a.point_list[list_base + pos_of(min(k, count - 1u))];
  }
  __device__ __forceinline__ void gather(Entry& e, uint32_t id) const {
    e.id = id;
    e.r0 = a.rec0[id];
    e.r1 = a.rec1[id];
    if (AUX) {  // a compile-time switch: a run-time test here costs the main kernel 8 % (measured)
      const float* __restrict__ c = a.colors3 + 3 * (size_t)id;
      e.r2 = make_float4(c[0], c[1], c[2], 0.f);
    } else {
      e.r2 = a.rec2[id];
    }
  }
  __device__ __forceinline__ ChunkWalker(const BlendArgs& a_, uint32_t base, uint32_t n) : a(a_), list_base(base), count(n) {
    const uint32_t l = (uint32_t)lane_id();
    const uint32_t id0 = load_id(l);
    id_next = load_id(WAVE + l);
    id_next2 = load_id(2 * WAVE + l);
    gather(cur, id0);
    gather(nxt, id_next);
    chunk = 0;
  }
  // number of list positions covered by the current chunk
  __device__ __forceinline__ uint32_t chunk_size() const { return min((uint32_t)WAVE, count - chunk * WAVE); }
  __device__ __forceinline__ bool valid() const { return chunk * WAVE < count; }
  // list position (0-based, in tile order) held by this lane in the current chunk
  __device__ __forceinline__ uint32_t lane_pos() const { return pos_of(chunk * WAVE + (uint32_t)lane_id()); }
  // advance: cur <- nxt, start the loads for the chunk after next
  __device__ __forceinline__ void advance() {
    const uint32_t l = (uint32_t)lane_id();
    cur = nxt;
    chunk++;
    gather(nxt, id_next2);
    id_next = id_next2;
    id_next2 = load_id((chunk + 2) * WAVE + l);
  }
};

// ----------------------------------------------------------------------------------
// K6: renderCUDA (forward), DGR/cuda_rasterizer/forward.cu:261-379.
// ----------------------------------------------------------------------------------
// One forward item: a quadrant of `tile` (or, on small images, a part of one: SPLIT).  `bg0..2`: the background, read once
// per kernel into scalar registers (see blend_forward_kernel).
template <bool PROFILE, bool AUX, int SPLIT, bool FAST, bool CKPT>
__device__ __forceinline__ void forward_item(const BlendArgs& a, uint32_t tile, uint32_t quad, const float bg0, const float bg1,
                                             const float bg2, uint32_t& prof_visited, uint64_t* prof_cyc) {
  PixelWave pw;
  ItemBox box;
  if (!setup_item<SPLIT>(a, tile, quad, pw, box)) return;  // (work_est / work_maxc were cleared by tile_worklist_kernel)
  quad &= 3u;
  const int lane = lane_id();
  const uint2 range = a.ranges[pw.tile];
  const float pfx = (float)pw.px, pfy = (float)pw.py;
  (void)box;  // (the cull rectangle follows the pixels that are still live, live_pixel_box)
  bool done = !pw.inside;
  float T = 1.0f;
  uint32_t last_contributor = 0;
  uint32_t evaluated = 0;  // entries this quadrant evaluated (wave-uniform): the backward's work estimate for the tile

  // Staging layout: the survivors of a chunk are stored in PAIRS, the two entries' values of each footprint input next
  // to each other, so that one uniform ds_read_b128 puts them into adjacent registers and the footprint arithmetic of
  // two entries runs as packed binary32 instructions (v_pk_add / v_pk_mul / v_pk_fma_f32: two IEEE operations per lane
  // and issue slot, the same roundings as the scalar forms).  A wave issues at most one instruction every four cycles,
  // and the forward ends with its longest item, so the instruction count of an entry is what the kernel's length follows.
  //   pair p (entries 2p, 2p+1), FWD_PAIR floats: [0] hA0 hA1 nB0 nB1 | [4] hC0 hC1 op0 op1 | [8] x0 x1 y0 y1 |
  //                                               [12] r0 g0 b0 z0 | [16] r1 g1 b1 z1 | [20] pos0 pos1 - -
  constexpr int FWD_PAIR = 24;
  constexpr bool SELECT_CHAIN = GSR_FWD_SELECT_CHAIN != 0;
  __shared__ __attribute__((aligned(16))) float sp[(WAVE / 2) * FWD_PAIR];
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const f32x2 pfx2 = {pfx, pfx}, pfy2 = {pfy, pfy};
  f32x2 C01 = {0.f, 0.f}, C2D = {0.f, 0.f};  // (C0, C1), (C2, D)
  // CKPT: the colour accumulated SINCE THE LAST CHECKPOINT, in accumulators of its own (round 5).  The backward needs, at a
  // segment's upper end, the colour of everything behind it; as C_final - C(checkpoint) that is a difference of two sums
  // near 1 whose increments were rounded to 6e-8 each, divided by a transmittance that may be 0.01 -- the gradients of a
  // Gaussian in front of a dense segment came out 1e-5 .. 2e-5 off (tools/fuzz_v2.py, seed 365).  A segment's own sum
  // carries its own magnitude's rounding only.
  f32x2 S01 = {0.f, 0.f};
  float S2 = 0.f;
  // Checkpoints for the backward's list segments (Image::ck_*): in front of list position pos(k) (CkTable) the state of every
  // pixel that is still live -- transmittance and accumulated colour -- goes into the tile's slot k (one 16-byte store per
  // lane and one atomic per wave: the slots are assigned by tile_worklist_kernel, nothing is allocated here).
  // (CKPT: a template switch, chosen per view by the host -- views with short lists run the kernel without any of this)
  constexpr bool ckpt = CKPT && !AUX;
  const uint32_t pidx = (uint32_t)((pw.py & (TILE - 1)) * TILE + (pw.px & (TILE - 1)));  // pixel inside its tile
  const uint32_t ck_base = ckpt ? a.ck_table[tile] : CK_NONE;  // rank of the tile among the checkpointed ones
  uint32_t ck_next = (ckpt && ck_base != CK_NONE) ? (uint32_t)a.ck_pos.chunk[1] : 0xffffffffu, ck_k = 1u;
  auto stage = [&](uint32_t slot, float hA, float nB, float hC, float op, float x, float y, float z, float r, float g,
                   float b, float pos) {
    float* q = sp + (slot >> 1) * FWD_PAIR + (slot & 1u);
    q[0] = hA;
    q[2] = nB;
    q[4] = hC;
    q[6] = op;
    q[8] = x;
    q[10] = y;
    q[20] = pos;
    *reinterpret_cast<float4*>(sp + (slot >> 1) * FWD_PAIR + 12 + 4 * (slot & 1u)) = make_float4(r, g, b, z);
  };
  if (range.y > range.x) {
    ChunkWalker<true, AUX> walk(a, range.x, range.y - range.x);
    for (; walk.valid(); walk.advance()) {
      const uint64_t live_m = __ballot(!done);
      if (live_m == 0) break;
      if (ckpt && walk.chunk == ck_next) {  // (uniform; never for walks shorter than the stride)
        if (ck_k < (uint32_t)a.ck_slots) {
          // slot k: T in front of position pos(k), and the colour of segment k - 1.  Written by every pixel of the
          // quadrant (a saturated one writes zeros: the backward never reads them, but it may read the slots of a pixel that
          // is live and simply found nothing to blend)
          if (pw.inside) a.ck_pool[((size_t)ck_base * (uint32_t)a.ck_slots + ck_k) * (TILE * TILE) + pidx] = make_float4(T, S01.x, S01.y, S2);
          S01 = f32x2{0.f, 0.f};
          S2 = 0.f;
          // (what this wave has evaluated so far: the backward's work list splits the tile's estimate with it)
          if (lane == 0) atomicAdd(&a.ck_work[(size_t)tile * (uint32_t)a.ck_slots + ck_k], evaluated);
        }
        ck_k++;
        ck_next = ck_k < (uint32_t)a.ck_slots ? (uint32_t)a.ck_pos.chunk[ck_k] : 0xffffffffu;  // (the table is ascending)
      }
      uint64_t tc0 = 0;
      if (PROFILE) {
        tc0 = __builtin_amdgcn_s_memtime();
        prof_cyc[2]++;  // chunks walked
      }
      const LiveBox lb = live_pixel_box(live_m, pfx - (float)(lane & 7), pfy - (float)(lane >> 3));
      const bool keep = ((uint32_t)lane < walk.chunk_size()) && can_touch_quad(walk.cur.r0, walk.cur.r1, lb.x0, lb.y0, lb.w, lb.h);
      const uint64_t m = __ballot(keep);
      if (PROFILE) prof_cyc[3] += __builtin_amdgcn_s_memtime() - tc0;  // wait for the chunk's records + cull
      if (m == 0) continue;
      const uint32_t cnt = (uint32_t)__popcll(m);
      const uint32_t cnt4 = (cnt + GROUP - 1) & ~(uint32_t)(GROUP - 1);
      __syncthreads();
      // a staged colour or depth that is NaN / Inf (0 * c would no longer be 0): this chunk takes the masked form
      const float csum = (walk.cur.r2.x + walk.cur.r2.y) + (walk.cur.r2.z + walk.cur.r1.z);
      const bool bad_value = keep && !(csum - csum == 0.0f);
      if (keep)  // conic pre-scaled: (-0.5 A, -B, -0.5 C), exact
        stage((uint32_t)__popcll(m & lt_mask), -0.5f * walk.cur.r0.x, -walk.cur.r0.y, -0.5f * walk.cur.r0.z, walk.cur.r0.w,
              walk.cur.r1.x, walk.cur.r1.y, walk.cur.r1.z, walk.cur.r2.x, walk.cur.r2.y, walk.cur.r2.z,
              __uint_as_float(walk.lane_pos() + 1u));
      if ((uint32_t)lane >= cnt && (uint32_t)lane < cnt4)
        // null entries pad the survivors to a multiple of GROUP: opacity 0 => alpha 0 => never a hit
        stage((uint32_t)lane, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
      __syncthreads();
      // GROUP entries per iteration: the footprint / exp evaluations of a group are independent
      // straight-line code (ILP for a wave that is alone on its SIMD); only the short
      // transmittance chain below is serial.
      uint64_t tc1 = 0;
      if (PROFILE) {
        tc1 = __builtin_amdgcn_s_memtime();
        prof_cyc[0] += tc1 - tc0;  // chunk start -> staged
      }
      if (SELECT_CHAIN && !__any(bad_value)) {
        // Fast path (every staged colour / depth of the chunk is finite): the serial part runs on selects only, in one
        // basic block without a single scalar instruction.  A wave issues one instruction about every 4.7 cycles whatever
        // its type, and a v_cmp -> SALU -> exec -> VALU turn costs ~40 cycles against ~13 for v_mul -> v_cmp -> v_cndmask
        // (tools/microbench/issue_rate.hip), so the length of the longest item follows the instruction count and the
        // number of such turns.  State: Ts = T for a live lane, -T for one that is done (out of the image, or saturated).
        //   entry skipped for this pixel (power > 0 or alpha < 1/255): alpha := 0, so 1 - alpha = 1, test_T = T exactly,
        //     w = 0 and the accumulators take c * 0 = 0 (c finite);
        //   T (1 - alpha) < 0.0001: A is false, w := 0, Ts := -T (final_T stays the T before this entry, forward.cu:349-354);
        //   lane done: test_T <= 0 < 0.0001, A false, nothing changes.
        // Same operations and roundings as the masked form below wherever a value is kept.
        float Ts = done ? -T : T;
        for (uint32_t j = 0; j < cnt4; j += GROUP) {
          if (__all(Ts < 0.0f)) break;
          evaluated += GROUP;
          if (PROFILE) prof_visited += GROUP;
          float ae[GROUP], om[GROUP], posf[GROUP];
          f32x2 rg[GROUP], bz[GROUP];
#pragma unroll
          for (int pp = 0; pp < GROUP / 2; ++pp) {
            const float* q = sp + ((j >> 1) + pp) * FWD_PAIR;
            const float4 q0 = *reinterpret_cast<const float4*>(q), q1 = *reinterpret_cast<const float4*>(q + 4),
                         q2 = *reinterpret_cast<const float4*>(q + 8), c0 = *reinterpret_cast<const float4*>(q + 12),
                         c1 = *reinterpret_cast<const float4*>(q + 16);
            const float2 ps = *reinterpret_cast<const float2*>(q + 20);
            const f32x2 hA = {q0.x, q0.y}, nB = {q0.z, q0.w}, hC = {q1.x, q1.y}, op = {q1.z, q1.w};
            const f32x2 dx = f32x2{q2.x, q2.y} - pfx2, dy = f32x2{q2.z, q2.w} - pfy2;
            const f32x2 a_ = (hA * dx) * dx;
            const f32x2 s_ = __builtin_elementwise_fma(hC * dy, dy, a_);
            const f32x2 power = __builtin_elementwise_fma(nB * dx, dy, s_);
            const f32x2 ao = op * blend_exp2<FAST>(power);
            const float a0 = fminf(0.99f, ao.x), a1 = fminf(0.99f, ao.y);
            const float g0 = (power.x > 0.0f) ? 0.0f : a0, g1 = (power.y > 0.0f) ? 0.0f : a1;
            const f32x2 a2 = {(g0 < 1.0f / 255.0f) ? 0.0f : g0, (g1 < 1.0f / 255.0f) ? 0.0f : g1};
            const f32x2 o2 = f32x2{1.0f, 1.0f} - a2;
            ae[2 * pp] = a2.x;
            ae[2 * pp + 1] = a2.y;
            om[2 * pp] = o2.x;
            om[2 * pp + 1] = o2.y;
            rg[2 * pp] = f32x2{c0.x, c0.y};
            bz[2 * pp] = f32x2{c0.z, c0.w};
            rg[2 * pp + 1] = f32x2{c1.x, c1.y};
            bz[2 * pp + 1] = f32x2{c1.z, c1.w};
            posf[2 * pp] = ps.x;
            posf[2 * pp + 1] = ps.y;
          }
#pragma unroll
          for (int u = 0; u < GROUP; ++u) asm volatile("" : "+v"(ae[u]), "+v"(om[u]));
#pragma unroll
          for (int u = 0; u < GROUP; ++u) {
            const float test_T = Ts * om[u];
            const bool A = !(test_T < 0.0001f);
            const float w = ae[u] * Ts;
            const float wz = A ? w : 0.0f;
            const f32x2 w2 = {wz, wz};
            C01 = __builtin_elementwise_fma(rg[u], w2, C01);
            C2D = __builtin_elementwise_fma(bz[u], w2, C2D);
            if (ckpt) {
              S01 = __builtin_elementwise_fma(rg[u], w2, S01);
              S2 = __builtin_fmaf(bz[u].x, wz, S2);
            }
            Ts = A ? test_T : -__builtin_fabsf(Ts);
            last_contributor = (wz > 0.0f) ? __float_as_uint(posf[u]) : last_contributor;
          }
        }
        done = Ts < 0.0f;
        T = __builtin_fabsf(Ts);
      } else {
        for (uint32_t j = 0; j < cnt4; j += GROUP) {
          if (__all(done)) break;
          evaluated += GROUP;
          if (PROFILE) prof_visited += GROUP;
          float al[GROUP], om[GROUP], posf[GROUP];
          bool ok[GROUP];
          f32x2 rg[GROUP], bz[GROUP];
  #pragma unroll
          for (int pp = 0; pp < GROUP / 2; ++pp) {
            const float* q = sp + ((j >> 1) + pp) * FWD_PAIR;
            const float4 q0 = *reinterpret_cast<const float4*>(q), q1 = *reinterpret_cast<const float4*>(q + 4),
                         q2 = *reinterpret_cast<const float4*>(q + 8), c0 = *reinterpret_cast<const float4*>(q + 12),
                         c1 = *reinterpret_cast<const float4*>(q + 16);
            const float2 ps = *reinterpret_cast<const float2*>(q + 20);
            const f32x2 hA = {q0.x, q0.y}, nB = {q0.z, q0.w}, hC = {q1.x, q1.y}, op = {q1.z, q1.w};
            const f32x2 dx = f32x2{q2.x, q2.y} - pfx2, dy = f32x2{q2.z, q2.w} - pfy2;
            // blend_power_prescaled on both entries
            const f32x2 a_ = (hA * dx) * dx;
            const f32x2 s_ = __builtin_elementwise_fma(hC * dy, dy, a_);
            const f32x2 power = __builtin_elementwise_fma(nB * dx, dy, s_);
            const f32x2 e = blend_exp2<FAST>(power);
            const f32x2 ao = op * e;
            const f32x2 a2 = {fminf(0.99f, ao.x), fminf(0.99f, ao.y)};
            const f32x2 o2 = f32x2{1.0f, 1.0f} - a2;
            al[2 * pp] = a2.x;
            al[2 * pp + 1] = a2.y;
            om[2 * pp] = o2.x;
            om[2 * pp + 1] = o2.y;
            ok[2 * pp] = !(power.x > 0.0f) && !(a2.x < 1.0f / 255.0f);
            ok[2 * pp + 1] = !(power.y > 0.0f) && !(a2.y < 1.0f / 255.0f);
            rg[2 * pp] = f32x2{c0.x, c0.y};
            bz[2 * pp] = f32x2{c0.z, c0.w};
            rg[2 * pp + 1] = f32x2{c1.x, c1.y};
            bz[2 * pp + 1] = f32x2{c1.z, c1.w};
            posf[2 * pp] = ps.x;
            posf[2 * pp + 1] = ps.y;
          }
          // pin the four footprints ahead of the serial part: otherwise the optimiser sinks each one into the masked
          // region that consumes it and the independent exp chains no longer overlap (forward blend -2 %)
  #pragma unroll
          for (int u = 0; u < GROUP; ++u) asm volatile("" : "+v"(al[u]), "+v"(om[u]));
  #pragma unroll
          for (int u = 0; u < GROUP; ++u) {
            const bool hit = ok[u] && !done;
            const float test_T = T * om[u];
            const bool term = hit && (test_T < 0.0001f);
            done = done || term;
            if (hit && !term) {  // kept as a lane-masked region: fewer instructions under exec than selects
              const float w = al[u] * T;
              const f32x2 w2 = {w, w};
              C01 = __builtin_elementwise_fma(rg[u], w2, C01);
              C2D = __builtin_elementwise_fma(bz[u], w2, C2D);
              if (ckpt) {
                S01 = __builtin_elementwise_fma(rg[u], w2, S01);
                S2 = __builtin_fmaf(bz[u].x, w, S2);
              }
              T = test_T;
              last_contributor = __float_as_uint(posf[u]);
            }
          }
        }
      }
      if (PROFILE) prof_cyc[1] += __builtin_amdgcn_s_memtime() - tc1;  // group loop
    }
  }
  const float C0 = C01.x, C1 = C01.y, C2 = C2D.x, D = C2D.y;
  // how deep the backward will walk this quadrant's part of the list (a lane outside the image never contributes)
  const uint32_t deep = (a.work_est != nullptr || ckpt) ? wave_max_u32_dpp(last_contributor) : 0u;
  if (ckpt) {
    // ... for the whole tile, and -- for the pixels some checkpoint was written for -- the colour of the segment the walk
    // ended in, as if its closing checkpoint had been reached: slot ck_k, or slot 0 once the tile's slots are used up (that
    // "segment" then reaches to the end of the list).  With it the colour of segment k is ALWAYS in slot k + 1 (k + 1 < ck_slots)
    // or in slot 0 (k = ck_slots - 1), whether the wave went on beyond it or not.
    if (deep > a.ck_pos.pos(1u) && lane == 0) atomicMax(&a.tile_maxc[tile], deep);
    if (ck_k > 1u && pw.inside)
      a.ck_pool[((size_t)ck_base * (uint32_t)a.ck_slots + (ck_k < (uint32_t)a.ck_slots ? ck_k : 0u)) * (TILE * TILE) + pidx] = make_float4(T, S01.x, S01.y, S2);
  }
  // the backward's work estimate and walk depth for the quadrant.  One item per quadrant: plain stores; the sub-items of a
  // cut quadrant report the largest of their values (they walk the same list; their SUM as the estimate was measured and is
  // worse, profiles/r06_s_forward_split.md).
  if (a.work_est != nullptr && lane == 0) {
    if (SPLIT == 1) {
      a.work_est[4u * tile + quad] = evaluated;
      a.work_maxc[4u * tile + quad] = deep;
    } else {
      atomicMax(&a.work_est[4u * tile + quad], evaluated);
      atomicMax(&a.work_maxc[4u * tile + quad], deep);
    }
  }
  if (pw.inside) {
    const size_t pix = (size_t)pw.py * a.W + pw.px, HW = (size_t)a.H * a.W;
    if (a.final_T != nullptr) {  // (null in an auxiliary render: the state the backward needs stays that of the main one)
      a.final_T[pix] = T;
      a.n_contrib[pix] = last_contributor;
    }
    a.out_color[pix] = __builtin_fmaf(T, bg0, C0);
    a.out_color[HW + pix] = __builtin_fmaf(T, bg1, C1);
    a.out_color[2 * HW + pix] = __builtin_fmaf(T, bg2, C2);
    if (a.out_depth != nullptr) a.out_depth[pix] = D;
  }
}

template <bool PROFILE, bool AUX, int SPLIT, bool FAST, bool CKPT>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(4, 4))) blend_forward_kernel(const BlendArgs a) {
  uint64_t t_start = 0;
  uint32_t prof_visited = 0, prof_items = 0;
  uint64_t prof_cyc[4] = {0, 0, 0, 0};
  if (PROFILE) t_start = __builtin_amdgcn_s_memtime();
  // `bg` once per kernel, in scalar registers.  (Read at its uses, hipcc cannot prove that the image stores do not alias it:
  // it re-loaded each component between two stores and waited for everything in flight each time.)
  const float bg0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bg[0])));
  const float bg1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bg[1])));
  const float bg2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bg[2])));
  // (always_inline: past a size threshold hipcc stops inlining the item function into the work loop's two call sites and
  //  CALLS it -- the 300-byte argument block then travels through scratch memory)
  run_work_queue<SPLIT>(a, true, [&](uint32_t tile, uint32_t quad, bool empty) __attribute__((always_inline)) {
    if (PROFILE) prof_items++;
    if (!empty) {
      forward_item<PROFILE, AUX, SPLIT, FAST, CKPT>(a, tile, quad, bg0, bg1, bg2, prof_visited, prof_cyc);
    } else {
      // a tile no Gaussian touches: background only (forward.cu:371-378 with an empty range)
      for (uint32_t q = 0; q < 4; ++q) {
        PixelWave pw;
        if (!setup_wave(a, tile, q, pw)) continue;
        if (pw.inside) {
          const size_t pix = (size_t)pw.py * a.W + pw.px, HW = (size_t)a.H * a.W;
          if (a.final_T != nullptr) {
            a.final_T[pix] = 1.0f;
            a.n_contrib[pix] = 0u;
          }
          a.out_color[pix] = __builtin_fmaf(1.0f, bg0, 0.f);
          a.out_color[HW + pix] = __builtin_fmaf(1.0f, bg1, 0.f);
          a.out_color[2 * HW + pix] = __builtin_fmaf(1.0f, bg2, 0.f);
          if (a.out_depth != nullptr) a.out_depth[pix] = 0.f;
        }
      }
    }
  });
  if (PROFILE) {
    // debug record per persistent wave (8 x u64): start, end (s_memtime ticks), XCC_ID<<32 | HW_ID,
    // items<<32 | visited entries, then the cycle split below
    const uint64_t t_end = __builtin_amdgcn_s_memtime();
    if (lane_id() == 0) {
      uint64_t* rec = a.profile + (size_t)blockIdx.x * 8;
      rec[4] = prof_cyc[0];  // cycles: chunk start -> survivors staged
      rec[5] = prof_cyc[1];  // cycles: group loops
      rec[6] = prof_cyc[2];  // chunks walked
      rec[7] = prof_cyc[3];  // cycles: chunk start -> cull ballot (includes the wait for the gathered records)
      rec[0] = t_start;
      rec[1] = t_end;
      rec[2] = ((uint64_t)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32) |  // HW_REG_XCC_ID[3:0]
               (uint64_t)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);              // HW_REG_HW_ID
      rec[3] = ((uint64_t)prof_items << 32) | (uint64_t)prof_visited;
    }
  }
}

// ----------------------------------------------------------------------------------
// K7: renderCUDA (backward), DGR/cuda_rasterizer/backward.cu:399-557.
// ----------------------------------------------------------------------------------
// Wave-level ordering of LDS traffic inside a multi-wave workgroup: a wave runs in lockstep, so all
// that is needed is that the compiler keeps program order and waits for the LDS queue.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

constexpr int BWD_WAVES = 4;  // one workgroup = the four quadrants of one tile (or two quadrants of a heavy one)
constexpr int ACC_LDS_ROW = 9;  // floats per chunk slot of the backward's LDS accumulators: the nine raw moments
constexpr uint32_t BWD_ITEM_HALF = 0x80000000u;   // item code: the workgroup handles one half of the tile ...
constexpr uint32_t BWD_ITEM_PART = 0x40000000u;   // ... quadrants {2,3} instead of {0,1}
// ... or (round 4) A RUN OF LIST SEGMENTS of a deep tile: the segments lo .. hi between its checkpoints, i.e. positions
// [pos(lo), pos(hi + 1)) of its list (CkTable: where a view's checkpoints sit), a run that ends with the tile's last segment
// up to the tile's deepest contributor.  The
// pixels that still contribute behind the run start from the forward's checkpoint in front of its upper end instead of from
// their final state (backward_tile).  The work list makes one item per stride (lo == hi): runs of several strides, merged
// to about equal measured work, were measured in round 6 and bought nothing (profiles/r06_m_fine_checkpoints.md).
constexpr uint32_t BWD_ITEM_SEG = 0x10000000u;
constexpr int BWD_SEG_SHIFT = 20, BWD_NSEG_SHIFT = 24;  // 4 bits each: first stride of the run, last stride of the run
static_assert(CK_MAX <= 16, "a stride index must fit the item code's 4 bits");
constexpr uint32_t BWD_ITEM_TILE = 0x000fffffu;   // (images of up to 2^20 tiles: GSR_MAX_TILES, checked by gsr_blend_backward)
static_assert(BWD_ITEM_TILE + 1u == (uint32_t)GSR_MAX_TILES, "include/gsr.h states the backward's tile limit");


// One tile, processed by a 4-wave workgroup (wave w = quadrant w).  The four waves walk the tile's list
// back to front in the SAME 64-entry chunks; each wave culls / compacts / evaluates the chunk for its own
// 64 pixels exactly as before, but the wave totals of the nine gradient terms are summed over the four
// quadrants in LDS (ds_add_f32), and only then does one global atomic per (tile, instance, term) leave
// the CU.  The global float atomics execute at the memory side on this chip and were the largest single
// cost of the backward (about 200 of 530 us with one atomic per quadrant); a Gaussian typically touches
// 2-3 of a tile's 4 quadrants.
template <int ABLATE, bool FAST, bool SEG>  // ABLATE: 0 = product; 1..6 = timing experiments only (wrong results), see launch_blend_backward
__device__ __forceinline__ uint32_t backward_tile(const BlendArgs& a, const uint4 item, const float bg0, const float bg1,
                                              const float bg2, float4 (*s0)[WAVE], float4 (*s1)[WAVE], float4 (*s2)[WAVE],
                                              uint32_t (*sid)[WAVE], float4 (*sco)[WAVE], float (*sacc)[WAVE][ACC_LDS_ROW]) {
  const int w = (int)(threadIdx.x >> 6), lane = lane_id();
  // The item's descriptor (assembled by the caller from three scalar loads): x = code -- a whole tile, wave w = quadrant w; or
  // half a tile, waves (0,1) and (2,3) = the upper / lower 8x4 pixels of its two quadrants; or one list segment --, y = first
  // position of the tile's list, z = how deep the item's pixels reach into it (the forward's work_maxc).  Until round 4 an
  // item's start was a chain of dependent round trips: work list -> ranges, final_T, n_contrib -> maximum over the
  // workgroup (two barriers) -> dL_dpixel, bg -> ids -> records.  Now everything per pixel and the first ids are requested
  // together, right behind the descriptor.
  uint32_t tile = item.x;
  const bool half_item = (tile & BWD_ITEM_HALF) != 0u;
  const uint32_t part = (tile & BWD_ITEM_PART) ? 1u : 0u;
  const bool seg_item = SEG && (tile & BWD_ITEM_SEG) != 0u;  // (SEG: a template switch -- the work list of a view without checkpoints holds no such item)
  const uint32_t seg_first = (tile >> BWD_SEG_SHIFT) & 15u, seg = (tile >> BWD_NSEG_SHIFT) & 15u;  // the run's first / last stride
  tile &= BWD_ITEM_TILE;
  const uint32_t list_base = item.y, tile_max = item.z;
  if (tile_max == 0) return 0u;  // (uniform: nothing of the item's pixels ever contributed)
  // the list positions this item walks: all of [0, tile_max), or a run of strides of them (the stride of the last slot
  // reaches to the end of the list, and so does whatever stride holds the tile's deepest contributor)
  const uint32_t seg_lo = seg_item ? a.ck_pos.pos(seg_first) : 0u;
  const uint32_t nslots = (uint32_t)a.ck_slots;  // slots in use per tile (<= CK_MAX), the stride of the pool and of ck_work
  const uint32_t seg_hi = (seg_item && seg != nslots - 1u) ? min(tile_max, a.ck_pos.pos(seg + 1u)) : tile_max;
  if (seg_lo >= seg_hi) return 0u;  // (uniform)
  PixelWave pw;
  const bool has_pixels = setup_wave(a, tile, half_item ? 2u * part + (uint32_t)(w >> 1) : (uint32_t)w, pw);
  const float pfx = (float)pw.px, pfy = (float)pw.py;
  const float qx0 = (float)(pw.px - (lane & 7)), qy0 = (float)(pw.py - (lane >> 3));  // origin of the wave's 8x8 lane grid
  if (half_item) pw.inside = pw.inside && ((lane >> 5) == (w & 1));  // (lanes of the other half never become live)
  const size_t HW = (size_t)a.H * a.W;
  const bool live = has_pixels && pw.inside;
  // per-pixel inputs: UNCONDITIONAL loads (a lane without a pixel reads pixel 0 and drops the values below; a predicated
  // load makes hipcc wait for the data at the issue point), in front of the walker's first ids
  const size_t pix = live ? (size_t)pw.py * a.W + pw.px : (size_t)0;
  const float T_final_ld = a.final_T[pix];
  const uint32_t n_contrib_ld = a.n_contrib[pix];
  const float dpx_ld[3] = {a.dL_dpix[pix], a.dL_dpix[HW + pix], a.dL_dpix[2 * HW + pix]};
  // back to front over the item's positions [seg_lo, seg_hi) of the tile's list, all four waves in the same chunks
  ChunkWalker<false> walk(a, list_base + seg_lo, seg_hi - seg_lo);

  const float T_final = live ? T_final_ld : 0.f;
  float T = T_final;
  const uint32_t last_contributor = live ? n_contrib_ld : 0u;
  const uint32_t maxc = wave_max_u32_dpp(last_contributor);  // (this wave's pixels; the item's is tile_max)
  const float dpx[3] = {live ? dpx_ld[0] : 0.f, live ? dpx_ld[1] : 0.f, live ? dpx_ld[2] : 0.f};
  const float bg_dot_dpixel = (0.f + bg0 * dpx[0]) + bg1 * dpx[1] + bg2 * dpx[2];
  const float neg_Tfinal_bg = -T_final * bg_dot_dpixel;  // the background's share of dL/dalpha is this times 1 / (1 - alpha)
  const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;

  float B_acc = 0.f, last_cdot = 0.f;  // sum_ch accum_rec[ch]*dL_dpixel[ch], sum_ch last_color[ch]*dL_dpixel[ch]
  float last_alpha = 0.f;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  if (seg_item && last_contributor > seg_hi) {
    // This pixel still contributes BEHIND the segment: it enters at position seg_hi - 1 with the state the sequential walk
    // would have there.  T in front of position seg_hi is the forward's checkpoint; the colour recurrence (accum_rec,
    // backward.cu:515) at that point is the colour of everything behind, over that transmittance:
    //   accum_rec = (C_final - C(seg_hi)) / T(seg_hi),  with nothing pending (last_alpha = 0).
    // Round 5: the colour behind is the SUM of the later segments' own colours (the forward accumulates every segment from
    // zero: slot k + 1 holds segment k's, slot 0 that of segment CK_MAX - 1 and everything after it), smallest first -- as
    // C_final - C(seg_hi) it was a difference of two numbers near 1 divided by a small transmittance.
    const uint32_t pidx = (uint32_t)((pw.py & (TILE - 1)) * TILE + (pw.px & (TILE - 1)));
    const size_t slot0 = (size_t)a.ck_table[tile] * nslots;  // (slot k >= 1: T in front of position pos(k) + segment k - 1's colour)
    uint32_t m = 0u;  // the segment of the pixel's last contributor: the checkpoints at or in front of its position
#pragma unroll
    for (uint32_t k = 1u; k < (uint32_t)CK_MAX; ++k) m += (k < nslots && last_contributor - 1u >= a.ck_pos.pos(k)) ? 1u : 0u;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f;
    // (wave-uniform bounds; a lane takes part from its own m down.  The two slot counts the library itself chooses get a
    //  loop with a constant upper end -- unrolled, as when CK_MAX was the only count: with both ends variable every
    //  iteration waits for its own load, 3-4 % of the whole kernel on the deep-tile views)
    auto add_slot = [&](uint32_t k, uint32_t top) __attribute__((always_inline)) {
      if (k > seg && k <= m) {
        const float4 sk = a.ck_pool[(slot0 + (k + 1u < top ? k + 1u : 0u)) * (TILE * TILE) + pidx];
        b0 += sk.y;
        b1 += sk.z;
        b2 += sk.w;
      }
    };
    auto sum_behind = [&](auto slots_c) __attribute__((always_inline)) {
      constexpr uint32_t S = decltype(slots_c)::value;
#pragma unroll
      for (uint32_t i = 1u; i < S; ++i) add_slot(S - i, S);  // k = S - 1 down to 1
    };
    if (nslots == (uint32_t)CK_MAX / 2u) sum_behind(std::integral_constant<uint32_t, (uint32_t)CK_MAX / 2u>{});
    else if (nslots == (uint32_t)CK_MAX) sum_behind(std::integral_constant<uint32_t, (uint32_t)CK_MAX>{});
    else
      for (uint32_t k = nslots - 1u; k > seg; --k) add_slot(k, nslots);  // (GSR_CK_SLOTS: any other count)
    T = a.ck_pool[(slot0 + seg + 1u) * (TILE * TILE) + pidx].x;
    B_acc = (b0 * dpx[0] + b1 * dpx[1] + b2 * dpx[2]) / T;
  }

  // Flush of one chunk's accumulators (round 6: ONE memory-side request per (tile, instance) instead of up to nine).
  // Device-scope float atomics execute at the memory side on this chip, and what they cost is the REQUEST: a wave instruction's
  // lanes that hit the same 64-byte line travel as one (tools/microbench/atomic_merge.hip: 9 values into each of 4 M random
  // rows -- 1 819 us from four arrays, 37.7 M requests; 206 us into 64-byte rows, 4.2 M requests).  With the nine accumulators
  // of a Gaussian in four arrays every atomic was a request of its own: 47 of K7's 215 us on the headline view, 200 of
  // 757 us on synth-v2 (GSR_BWD_ABLATE=2).  Now a Gaussian's accumulators are ONE 64-byte row of `acc` (ACC_* columns,
  // include/gsr.h), and the flush puts the sixteen columns of a row on sixteen adjacent lanes: instruction j of wave w
  // covers the chunk slots 16 w + 4 j .. + 3.  Lane (slot, column) forms its column's term from the slot's raw moments and
  // the entry's conic / opacity (sco) -- backward.cu:545-554 --, skips what is exactly zero, and the lanes of the raw
  // moments' indices put the LDS accumulator back to zero for the buffer's next chunk.
  auto flush = [&](uint32_t fb, uint32_t fsize) __attribute__((always_inline)) {
    constexpr bool emit = ABLATE != 2 && ABLATE != 3 && ABLATE != 6;  // (experiments: no global atomics)
    const uint32_t col = (uint32_t)lane & 15u, sub = (uint32_t)lane >> 4;
    // raw moments a column reads: dL_dmean2D needs (1, 2); conic x / y / w: 3 / 4 / 5; opacity: 0; colour: 6 / 7 / 8
    const uint32_t ia = col == ACC_MEAN2D || col == ACC_MEAN2D + 1u ? 1u
                        : col == ACC_OPACITY ? 0u
                        : col == ACC_CONIC ? 3u : col == ACC_CONIC + 1u ? 4u : col == ACC_CONIC + 3u ? 5u
                        : col >= ACC_COLOR && col < ACC_COLOR + 3u ? 6u + (col - ACC_COLOR) : 9u;  // 9: the column is not used
#pragma unroll
    for (uint32_t jj = 0; jj < 4u; ++jj) {
      const uint32_t p = 16u * (uint32_t)w + 4u * jj + sub;
      if (16u * (uint32_t)w + 4u * jj >= fsize) break;  // (wave-uniform: slots >= fsize are never added to)
      float* const row = sacc[fb][p];
      const bool used = p < fsize && ia < 9u;
      const float ma = used ? row[ia] : 0.f;
      const float mb = used && ia == 1u ? row[2] : 0.f;
      const float4 co = sco[fb][p];  // conic.x, conic.y, conic.z, opacity
      float val;
      bool nz;
      if (ia == 1u) {
        // backward.cu:545-546: dL_dmean2D = -o (A t1 + B t2, B t1 + C t2) * (0.5 W, 0.5 H)
        const bool xcol = col == ACC_MEAN2D;
        const float scale = xcol ? ddelx_dx : ddely_dy, c1 = xcol ? co.x : co.y, c2 = xcol ? co.y : co.z;
        val = -(scale * co.w) * (c1 * ma + c2 * mb);
        nz = ma != 0.f || mb != 0.f;
      } else {
        const bool conic = ia >= 3u && ia <= 5u;  // backward.cu:549-551: -0.5 o t; opacity (:554) and colour (:523): the moment itself
        val = conic ? (-0.5f * co.w) * ma : ma;
        nz = ma != 0.f;
      }
      if (used && nz && emit) unsafeAtomicAdd(&a.acc[(size_t)sid[fb][p] * ACC_ROW + col], val);
      if (a.touched != nullptr && emit) {  // (wave-uniform) the exchange's row mask, without a pass over the table: lane 15 of a
        const uint64_t nzm = __ballot(used && nz);  // group marks the group's Gaussian if any of its columns was added to
        if (col == 15u && p < fsize && ((nzm >> (16u * sub)) & 0xffffull) != 0ull) a.touched[sid[fb][p]] = 1;
      }
      if (p < fsize && col < 9u) row[col] = 0.f;  // (every read of this instruction precedes it: one wave, program order)
    }
  };

  // The flush of chunk k is DEFERRED into iteration k + 1 (round 5), behind the walker's prefetch loads: a wave's vector
  // memory operations retire in order, so the wait for the next chunk's records at the top of an iteration also waited for
  // the memory-side float atomics the flush had issued a few instructions earlier (ISA: s_waitcnt vmcnt(0) in the loop
  // header) -- one atomic round trip per chunk on every wave's critical path.  Now the atomics of chunk k have the whole
  // group loop of chunk k + 1 to complete.  The accumulators, sid and sco are double-buffered by chunk parity, which also
  // removes one of the two workgroup barriers per chunk: chunk k + 1 writes buffer (k + 1) & 1, whose last flush (chunk
  // k - 1) ran in iteration k, in front of barrier (B) of that iteration.
  uint32_t pend_size = 0u;  // slots of the previous chunk still to be flushed (0: none)
  for (; walk.valid(); walk.advance()) {
    const uint32_t csize = walk.chunk_size();
    const uint32_t cb = walk.chunk & 1u;
#if !GSR_BWD_DEFER_FLUSH
    __syncthreads();  // (A) the previous chunk's flush is complete: sacc is zero again, sid is free
#endif
    if (w == 0 && (uint32_t)lane < csize) {
      sid[cb][lane] = walk.cur.id;
      sco[cb][lane] = walk.cur.r0;
    }
    const uint32_t pos = seg_lo + walk.lane_pos();
    // pixels that can still use an entry of this chunk: their last contributor lies above the chunk's lowest position;
    // the cull rectangle is their bounding box (the first chunks of a long list are walked for a few stragglers only)
    const uint32_t chunk_lo = seg_hi - min(seg_hi - seg_lo, (walk.chunk + 1u) * (uint32_t)WAVE);
    const uint64_t live_m = __ballot(last_contributor > chunk_lo);
    bool keep = false;
    if (live_m != 0) {
      const LiveBox lb = live_pixel_box(live_m, qx0, qy0);
      keep = ((uint32_t)lane < csize) && (pos < maxc) && can_touch_quad(walk.cur.r0, walk.cur.r1, lb.x0, lb.y0, lb.w, lb.h);
    }
    const uint64_t m = __ballot(keep);
    const uint32_t cnt = (uint32_t)__popcll(m);
    const uint32_t cnt4 = (cnt + GROUP - 1) & ~(uint32_t)(GROUP - 1);
    if (keep) {
      const uint32_t slot = (uint32_t)__popcll(m & lt_mask);
      // conic pre-scaled as in the forward: (-0.5 A, -B, -0.5 C), exact
      s0[w][slot] = make_float4(-0.5f * walk.cur.r0.x, -walk.cur.r0.y, -0.5f * walk.cur.r0.z, walk.cur.r0.w);
      s1[w][slot] = make_float4(walk.cur.r1.x, walk.cur.r1.y, walk.cur.r1.z, __uint_as_float(pos));
      // .w carries the entry's index inside the chunk (= the lane that holds it): the LDS accumulator slot
      s2[w][slot] = make_float4(walk.cur.r2.x, walk.cur.r2.y, walk.cur.r2.z, __uint_as_float((uint32_t)lane));
    }
    if ((uint32_t)lane >= cnt && (uint32_t)lane < cnt4) {
      s0[w][lane] = make_float4(0.f, 0.f, 0.f, 0.f);  // null entry: alpha 0 => never contributes
      s1[w][lane] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xffffffffu));
      s2[w][lane] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
    }
    wave_lds_sync();
#if GSR_BWD_DEFER_FLUSH
    if (pend_size != 0u) flush(cb ^ 1u, pend_size);  // (behind barrier (B) of the previous iteration)
    pend_size = csize;
#endif

    for (uint32_t j = 0; j < cnt4; j += GROUP) {
      float G[GROUP], al[GROUP], dxs[GROUP], dys[GROUP];
      bool contrib[GROUP];
      float4 cos_[GROUP], cols[GROUP];
      bool any = false;
#if GSR_BWD_HYBRID_EXP
      // The exponential of the backward (round 6).  The FORWARD's colours are compared bit for bit, so it evaluates the
      // exactly specified polynomial (12 instructions); here exp(power) enters two things: the decision `alpha < 1/255`,
      // which must be the forward's, and G / alpha as factors of gradients that are compared at 1e-5.  So: the hardware's
      // 2^x (v_exp_f32, 1 ulp; with the rounding of power * log2(e) at most 4e-7 relative for the powers that matter,
      // |power| < 5.6) gives G and alpha, and ONLY a group in which some lane's o * G lies within 2^-18 relative of 1/255
      // (about one group in 10^4) is evaluated again with the polynomial, wave-uniformly.  A lane further from the
      // threshold than the two results can differ takes the forward's decision by construction.
      float pw_[GROUP];
      bool near = false;
#pragma unroll
      for (int u = 0; u < GROUP; ++u) {
        const float4 g = s1[w][j + u];
        cos_[u] = s0[w][j + u];
        cols[u] = s2[w][j + u];
        dxs[u] = g.x - pfx;
        dys[u] = g.y - pfy;
        pw_[u] = blend_power_prescaled(cos_[u].x, cos_[u].y, cos_[u].z, dxs[u], dys[u]);
        G[u] = __builtin_amdgcn_exp2f(pw_[u] * 0x1.715476p+0f);
        const float ao = cos_[u].w * G[u];
        al[u] = fminf(0.99f, ao);
        near = near || (!FAST && __builtin_fabsf(ao - 1.0f / 255.0f) <= (1.0f / 255.0f) * 0x1p-18f);
      }
      if (!FAST && __any(near)) {  // wave-uniform, rare
        asm volatile("; polynomial exp: a lane within 2^-18 of the alpha threshold");  // (keeps hipcc from flattening the branch into selects)
#pragma unroll
        for (int u = 0; u < GROUP; ++u) {
          G[u] = gsr_expf_noclamp(pw_[u]);
          al[u] = fminf(0.99f, cos_[u].w * G[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < GROUP; ++u) {
        const uint32_t c = __float_as_uint(s1[w][j + u].w);  // 0-based position of this instance in the tile list
        contrib[u] = (c < last_contributor) && !(pw_[u] > 0.0f) && !(al[u] < 1.0f / 255.0f);
        any = any || contrib[u];
      }
#else
#pragma unroll
      for (int u = 0; u < GROUP; ++u) {
        const float4 g = s1[w][j + u];
        cos_[u] = s0[w][j + u];
        cols[u] = s2[w][j + u];
        const uint32_t c = __float_as_uint(g.w);  // 0-based position of this instance in the tile list
        dxs[u] = g.x - pfx;
        dys[u] = g.y - pfy;
        const float power = blend_power_prescaled(cos_[u].x, cos_[u].y, cos_[u].z, dxs[u], dys[u]);
        G[u] = blend_exp<FAST>(power);  // only used where alpha >= 1/255, i.e. far above the clamp
        al[u] = fminf(0.99f, cos_[u].w * G[u]);
        contrib[u] = (c < last_contributor) && !(power > 0.0f) && !(al[u] < 1.0f / 255.0f);
        any = any || contrib[u];
      }
#endif
      if (!__any(any)) continue;  // wave-uniform
      if (ABLATE == 4) continue;  // experiment: footprint + exp only

      float v[9][GROUP];
      {
#pragma clang fp contract(fast)  // gradients are tolerance-checked (<= 1e-5), not bit-compared: let a*b+c fuse here
#pragma unroll
        for (int u = 0; u < GROUP; ++u) {
          // backward.cu:503-534 with the colour recurrence collapsed: the reference keeps accum_rec[ch] and
          // last_color[ch] per channel and forms sum_ch (c[ch] - accum_rec[ch]) * dL_dpixel[ch]; the recurrence is
          // linear and dL_dpixel is constant for the pixel, so B = sum_ch accum_rec[ch] * dL_dpixel[ch] obeys the
          // same recurrence with the scalar cdot = sum_ch c[ch] * dL_dpixel[ch] in place of the colour.  Lanes that
          // do not contribute keep their state through selects and hand zeros on.
          // A lane that does not contribute runs the same arithmetic with alpha = 0: its transmittance is multiplied
          // by rcp(1) = 1 exactly, the pending fold of (last_alpha, last_cdot) into B happens now instead of at the
          // lane's next contributor (same operands, same value; afterwards last_alpha = 0 makes the fold a no-op),
          // and its moments are zero -- three selects per entry instead of six.
          const bool on = contrib[u];
          const float alpha = on ? al[u] : 0.f;
#if GSR_BWD_DIV == 1  // (A/B builds) IEEE division
          const float inv_one_m = 1.0f / (1.f - alpha);
#elif GSR_BWD_DIV == 2  // rcp + one Newton step
          const float one_m = 1.f - alpha;
          const float r0 = __builtin_amdgcn_rcpf(one_m);
          const float inv_one_m = __builtin_fmaf(r0, __builtin_fmaf(-one_m, r0, 1.0f), r0);
#else
          const float inv_one_m = __builtin_amdgcn_rcpf(1.f - alpha);
#endif
          const float cdot = cols[u].x * dpx[0] + cols[u].y * dpx[1] + cols[u].z * dpx[2];
          T = T * inv_one_m;                                      // T / (1 - alpha), backward.cu:503
          B_acc = B_acc + last_alpha * (last_cdot - B_acc);       // accum_rec update, backward.cu:515
          const float dL_dalpha = (cdot - B_acc) * T + neg_Tfinal_bg * inv_one_m;
          last_alpha = alpha;
          last_cdot = on ? cdot : last_cdot;  // (kept by a select: a NaN colour of a skipped entry must not enter B as 0 * NaN)
          // Per lane only the MOMENTS of q = G * dL/dalpha are formed; every per-entry constant of
          // backward.cu:538-554 (conic, opacity, 0.5*W, 0.5*H, -0.5) is applied once per (tile, entry) by the flush.
          const float q = on ? G[u] * dL_dalpha : 0.f;
          const float mD = alpha * T;  // dchannel_dcolor (0 for a skipped lane)
          const float dx = dxs[u], dy = dys[u];
          const float qx = q * dx, qy = q * dy;
          v[0][u] = q;
          v[1][u] = qx;
          v[2][u] = qy;
          v[3][u] = qx * dx;
          v[4][u] = qx * dy;
          v[5][u] = qy * dy;
          v[6][u] = mD * dpx[0];
          v[7][u] = mD * dpx[1];
          v[8][u] = mD * dpx[2];
        }
      }
      // 4-entry transposed wave reduction: afterwards lane 15 of row r holds the wave totals of entry j + r and adds the
      // nine RAW moments to the tile-level accumulator of the entry's chunk slot (they are relative to the entry's own
      // mean, so the quadrants' sums simply add; the flush turns them into the reference's terms).
      float tot[9];
#pragma unroll
      for (int k = 0; k < 9; ++k)
        tot[k] = (ABLATE == 1 || ABLATE == 3) ? (v[k][0] + v[k][1]) + (v[k][2] + v[k][3])  // experiment: no reduction
                                              : wave_sum4_to_rows(v[k][0], v[k][1], v[k][2], v[k][3]);
      // keep the reductions whole in front of the 4-lane tail: otherwise the last row_shr step is sunk into the masked
      // region as v_mov 0 + v_mov_dpp + v_add (3 instructions per term instead of one v_add_f32_dpp)
#pragma unroll
      for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(tot[k]));
      if ((lane & 15) == 15) {
#if GSR_BWD_SLOT_REG  // (A/B) the row's accumulator slot from the colour records already in registers: no LDS round trip in the tail
        const uint32_t row = (uint32_t)lane >> 4;
        const float sw = row == 0u ? cols[0].w : row == 1u ? cols[1].w : row == 2u ? cols[2].w : cols[3].w;
        const uint32_t my_slot = __float_as_uint(sw);
#else
        const uint32_t my_slot = __float_as_uint(s2[w][j + (uint32_t)(lane >> 4)].w);
#endif
#pragma unroll
        for (int k = 0; k < 9; ++k) atomicAdd(&sacc[cb][my_slot][k], tot[k]);
      }
    }
    if (ABLATE != 5 && ABLATE != 6) __syncthreads();  // (B) every quadrant's contribution to this chunk is in sacc[cb]
#if !GSR_BWD_DEFER_FLUSH
    flush(cb, csize);
#endif
  }
#if GSR_BWD_DEFER_FLUSH
  // the last chunk (the next item's first LDS write lies behind barriers).  The walker's last prefetch -- chunks beyond the
  // end, never used -- is drained first: the flush re-uses its registers, and behind the first atomic every such wait
  // would be one for the atomic.
  asm volatile("" ::"v"(walk.nxt.r0.x), "v"(walk.nxt.r1.x), "v"(walk.nxt.r2.x), "v"(walk.id_next), "v"(walk.id_next2));
  if (pend_size != 0u) flush((walk.chunk - 1u) & 1u, pend_size);
#endif
  return seg_hi - seg_lo;
}

#ifndef GSR_BWD_WAVES_PER_EU
#define GSR_BWD_WAVES_PER_EU 4  // (A/B builds: 5 with GSR_BLEND_WAVES_PER_SIMD=5)
#endif
template <int ABLATE, bool FAST, bool SEG>
__global__ void __launch_bounds__(WAVE* BWD_WAVES) __attribute__((amdgpu_waves_per_eu(GSR_BWD_WAVES_PER_EU, GSR_BWD_WAVES_PER_EU)))
blend_backward_kernel(const BlendArgs a) {
  __shared__ float4 s0[BWD_WAVES][WAVE], s1[BWD_WAVES][WAVE], s2[BWD_WAVES][WAVE];
  __shared__ uint32_t sid[2][WAVE];  // (sid, sco, sacc: double-buffered by chunk parity, see backward_tile)
  __shared__ float4 sco[2][WAVE];
  __shared__ float sacc[2][WAVE][ACC_LDS_ROW];  // raw moments 0..8 of every chunk slot (row stride 9: odd, the four rows of a flush instruction spread over the banks)
  __shared__ uint32_t s_item;
  for (int i = threadIdx.x; i < 2 * WAVE * ACC_LDS_ROW; i += WAVE * BWD_WAVES) (&sacc[0][0][0])[i] = 0.f;
  // (element i was zeroed by thread i % 256, i.e. by any wave: with the deferred flush the placement-assigned first item
  // reaches its first ds_add_f32 without passing a workgroup barrier otherwise.  Once per kernel, not per item.)
  __syncthreads();
  // `bg` once per kernel, in scalar registers
  const float bg0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bg[0])));
  const float bg1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bg[1])));
  const float bg2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bg[2])));
  // item-granular queue: queue x (one per XCD) owns items x, x+8, ... of the backward's work list
  const uint32_t x = (uint32_t)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;  // HW_REG_XCC_ID
  const uint32_t nwork = a.bwd_meta[0];
  const uint32_t n_x = nwork > x ? (nwork - x + 7u) / 8u : 0u;
  uint32_t* head = a.queue + x * QUEUE_STRIDE;
  // debug timing (gsr_debug_blend_backward_profile): per workgroup start / end, tiles, longest and first tile
  const bool prof = a.profile != nullptr;
  uint64_t t_begin = 0, t_tile = 0, longest = 0, first = 0;
  uint32_t ntiles = 0, west = 0;
  if (prof) t_begin = __builtin_amdgcn_s_memtime();
  auto run_tile = [&](uint32_t index) {
    if (prof) t_tile = __builtin_amdgcn_s_memtime();
    // The item's descriptor -- code, first position of the tile's list, how deep the item's pixels reach into it -- from three
    // SCALAR loads (tables of earlier kernels: the work list, the ranges, the walk depths the forward left per quadrant; the
    // scalar path does not queue behind the previous item's last atomics): code first, the other two together.  (Round 5's
    // first form had backward_worklist_kernel write whole descriptors: that one-block kernel is a chain of dependent round
    // trips between the forward and the backward -- 14 -> 16-18 us for what costs this kernel, whose waves wait behind three
    // others, nothing measurable.)
    const uint32_t code = scalar_load(a.bwd_order, index);
    const uint32_t ctile = code & BWD_ITEM_TILE;
    const uint2 crange = scalar_load(a.ranges, ctile);
    const uint4 cmax = scalar_load(reinterpret_cast<const uint4*>(a.work_maxc), ctile);
    const uint32_t r01 = max(cmax.x, cmax.y), r23 = max(cmax.z, cmax.w);
    const uint4 item = make_uint4(code, crange.x, (code & BWD_ITEM_HALF) ? ((code & BWD_ITEM_PART) ? r23 : r01) : max(r01, r23), 0u);
    const uint32_t tile = item.x;
    const uint32_t tmax = backward_tile<ABLATE, FAST, SEG>(a, item, bg0, bg1, bg2, s0, s1, s2, sid, sco, sacc);
    if (prof) {
      const uint64_t d = __builtin_amdgcn_s_memtime() - t_tile;
      if (a.profile_items != nullptr && threadIdx.x == 0 && a.work_est != nullptr) {
        // per item: cycles, the forward's per-quadrant counts of the tile, positions walked, item code
        const uint32_t tt = tile & BWD_ITEM_TILE;
        uint64_t* r = a.profile_items + ((size_t)tt * 2 + ((tile & BWD_ITEM_PART) ? 1 : 0)) * 4;
        const uint4 e = reinterpret_cast<const uint4*>(a.work_est)[tt];
        r[0] = d;
        r[1] = ((uint64_t)e.x) | ((uint64_t)e.y << 16) | ((uint64_t)e.z << 32) | ((uint64_t)e.w << 48);
        r[2] = tmax;
        r[3] = tile;
      }
      if (ntiles == 0) first = d;
      longest = d > longest ? d : longest;
      ntiles++;
      if (a.work_est != nullptr) {
        const uint32_t tt = tile & BWD_ITEM_TILE;
        if (tile & BWD_ITEM_HALF) {
          const uint32_t q0 = (tile & BWD_ITEM_PART) ? 2u : 0u;
          west += a.work_est[4u * tt + q0] + a.work_est[4u * tt + q0 + 1];
        } else {
          west += a.work_est[4u * tt] + a.work_est[4u * tt + 1] + a.work_est[4u * tt + 2] + a.work_est[4u * tt + 3];
        }
      }
    }
  };
  // first item assigned by placement (see first_item_of_block), the rest popped
  uint32_t x0, base;
  const uint32_t q0 = first_item_of_block((uint32_t)a.units, x0, base);
  {
    const uint32_t n0 = nwork > x0 ? (nwork - x0 + 7u) / 8u : 0u;
    if (q0 < n0) run_tile(x0 + 8u * q0);
  }
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_item = base + __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t q = s_item;
    if (q >= n_x) break;
    run_tile(x + 8u * q);
  }
  if (prof && threadIdx.x == 0) {
    uint64_t* rec = a.profile + (size_t)blockIdx.x * 8;
    rec[0] = t_begin;
    rec[1] = __builtin_amdgcn_s_memtime();
    rec[2] = ((uint64_t)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32) |
             (uint64_t)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    rec[3] = ((uint64_t)ntiles << 32) | (uint64_t)west;
    rec[4] = longest;
    rec[5] = first;
  }
  if (a.self_reset && threadIdx.x == 0) retire_queue(a.queue);
}

// ----------------------------------------------------------------------------------
// K12: renderCUDA_apply_weights, DGR/cuda_rasterizer/apply_weights.cu:239-356.
// Same traversal / termination as K6; every blended (pixel, instance) adds the pixel's
// mask value(s) to weights[id] and C to cnt[id] (the reference increments cnt inside its
// channel loop, apply_weights.cu:331-339).
// ----------------------------------------------------------------------------------
#ifndef GSR_TRACE_ABLATE
#define GSR_TRACE_ABLATE 0  // A/B builds, timing only (WRONG results): 1 no atomics, 2 no cross-lane reduction, 3 neither
#endif
// The inner loop is K6's (round 6; rounds 1-5 walked one survivor per iteration with a ballot, a branch and a 64-lane reduction
// per entry: 182 -> 122 us per 512 x 512 view at 10^6 Gaussians, profiles/r06_j_trace_weights.md): survivors are staged in pairs
// and evaluated two per instruction (packed binary32, same roundings), four per iteration; the serial part -- transmittance,
// saturation -- runs on selects (Ts = T for a live pixel, -T for one that is done), and what a pixel adds for the four entries
// (its mask value per channel, and 1 for the count) is summed over the wave four values at a time (wave_sum4_to_rows: 10
// instructions instead of 4 x 8).
template <int C, int SPLIT, bool FAST>
__device__ __forceinline__ void trace_item(const BlendArgs& a, uint32_t tile, uint32_t quad) {
  PixelWave pw;
  ItemBox box;
  if (!setup_item<SPLIT>(a, tile, quad, pw, box)) return;
  const int lane = lane_id();
  const uint2 range = a.ranges[pw.tile];
  if (range.y <= range.x) return;
  const float pfx = (float)pw.px, pfy = (float)pw.py;
  const f32x2 pfx2 = {pfx, pfx}, pfy2 = {pfy, pfy};
  (void)box;  // (the cull rectangle follows the pixels that are still live, live_pixel_box)
  const size_t pix = (size_t)pw.py * a.W + pw.px, HW = (size_t)a.H * a.W;
  float Cw[C];
#pragma unroll
  for (int ch = 0; ch < C; ++ch) Cw[ch] = pw.inside ? a.image_weights[(size_t)ch * HW + pix] : 0.f;
  float Ts = pw.inside ? 1.0f : -1.0f;

  // a pair of survivors: {hA0,hA1,nB0,nB1} {hC0,hC1,op0,op1} {x0,x1,y0,y1} (conic pre-scaled: -0.5 A, -B, -0.5 C, exact)
  __shared__ __align__(16) float spair[WAVE / 2][12];
  __shared__ uint32_t sid[WAVE];
  __shared__ float sacc[C][WAVE];
  __shared__ float scnt[WAVE];
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const bool row_end = (lane & 15) == 15;
  ChunkWalker<true> walk(a, range.x, range.y - range.x);
  for (; walk.valid(); walk.advance()) {
    const uint64_t live_m = __ballot(Ts > 0.0f);
    if (live_m == 0) break;
    const LiveBox lb = live_pixel_box(live_m, pfx - (float)(lane & 7), pfy - (float)(lane >> 3));
    const bool keep = ((uint32_t)lane < walk.chunk_size()) && can_touch_quad(walk.cur.r0, walk.cur.r1, lb.x0, lb.y0, lb.w, lb.h);
    const uint64_t km = __ballot(keep);
    if (km == 0) continue;
    const uint32_t n = (uint32_t)__popcll(km);
    const uint32_t n4 = (n + GROUP - 1) & ~(uint32_t)(GROUP - 1);
    __syncthreads();
    if (keep) {
      const uint32_t slot = (uint32_t)__popcll(km & lt_mask);
      float* q = &spair[slot >> 1][slot & 1u];
      q[0] = -0.5f * walk.cur.r0.x;
      q[2] = -walk.cur.r0.y;
      q[4] = -0.5f * walk.cur.r0.z;
      q[6] = walk.cur.r0.w;
      q[8] = walk.cur.r1.x;
      q[10] = walk.cur.r1.y;
      sid[slot] = walk.cur.id;
    }
    if ((uint32_t)lane >= n && (uint32_t)lane < n4) {
      // null entries pad the survivors to a multiple of GROUP: opacity 0 => alpha 0 => never a hit
      float* q = &spair[lane >> 1][lane & 1];
      q[0] = q[2] = q[4] = q[6] = q[8] = q[10] = 0.f;
    }
    __syncthreads();
    uint32_t flushed = n;  // entries whose totals this chunk wrote (all of them unless every pixel saturated on the way)
    for (uint32_t j = 0; j < n4; j += GROUP) {
      if (__all(Ts < 0.0f)) {
        flushed = j;
        break;
      }
      float ae[GROUP], om[GROUP];
#pragma unroll
      for (int pp = 0; pp < GROUP / 2; ++pp) {
        const float* q = &spair[(j >> 1) + pp][0];
        const float4 q0 = *reinterpret_cast<const float4*>(q), q1 = *reinterpret_cast<const float4*>(q + 4),
                     q2 = *reinterpret_cast<const float4*>(q + 8);
        const f32x2 hA = {q0.x, q0.y}, nB = {q0.z, q0.w}, hC = {q1.x, q1.y}, op = {q1.z, q1.w};
        const f32x2 dx = f32x2{q2.x, q2.y} - pfx2, dy = f32x2{q2.z, q2.w} - pfy2;
        const f32x2 a_ = (hA * dx) * dx;  // blend_power_prescaled on both entries
        const f32x2 s_ = __builtin_elementwise_fma(hC * dy, dy, a_);
        const f32x2 power = __builtin_elementwise_fma(nB * dx, dy, s_);
        const f32x2 ao = op * blend_exp2<FAST>(power);
        const float a0 = fminf(0.99f, ao.x), a1 = fminf(0.99f, ao.y);
        const float g0 = (power.x > 0.0f) ? 0.0f : a0, g1 = (power.y > 0.0f) ? 0.0f : a1;
        const f32x2 a2 = {(g0 < 1.0f / 255.0f) ? 0.0f : g0, (g1 < 1.0f / 255.0f) ? 0.0f : g1};
        const f32x2 o2 = f32x2{1.0f, 1.0f} - a2;
        ae[2 * pp] = a2.x;
        ae[2 * pp + 1] = a2.y;
        om[2 * pp] = o2.x;
        om[2 * pp + 1] = o2.y;
      }
#pragma unroll
      for (int u = 0; u < GROUP; ++u) asm volatile("" : "+v"(ae[u]), "+v"(om[u]));
      // entry skipped for this pixel (power > 0 or alpha < 1/255): alpha := 0, test_T = T exactly, no hit;
      // T (1 - alpha) < 0.0001: the pixel is done BEFORE this entry counts (apply_weights.cu:318-323), Ts := -T;
      // pixel done: test_T <= 0, nothing changes.
      float h[GROUP];
#pragma unroll
      for (int u = 0; u < GROUP; ++u) {
        const float test_T = Ts * om[u];
        const bool A = !(test_T < 0.0001f);
        h[u] = (A && ae[u] > 0.0f) ? 1.0f : 0.0f;
        Ts = A ? test_T : -__builtin_fabsf(Ts);
      }
      const uint32_t e = j + ((uint32_t)lane >> 4);  // the entry whose totals end up in this lane's row
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        const float z = (GSR_TRACE_ABLATE & 2) ? h[0] * Cw[ch]
                                                : wave_sum4_to_rows(h[0] > 0.f ? Cw[ch] : 0.f, h[1] > 0.f ? Cw[ch] : 0.f,
                                                                    h[2] > 0.f ? Cw[ch] : 0.f, h[3] > 0.f ? Cw[ch] : 0.f);
        if (row_end) sacc[ch][e] = z;
      }
      const float zc = (GSR_TRACE_ABLATE & 2) ? h[0] : wave_sum4_to_rows(h[0], h[1], h[2], h[3]);
      if (row_end) scnt[e] = zc;
    }
    __syncthreads();
    // (the groups behind a break wrote nothing: their slots hold an earlier chunk's values)
    if ((uint32_t)lane < min(n, flushed)) {
      const int cn = (int)scnt[lane] * C;
      if (cn != 0 && (!(GSR_TRACE_ABLATE & 1) || sacc[0][lane] == 12345.678f)) {
        const size_t id = sid[lane];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) unsafeAtomicAdd(&a.weights[id * C + ch], sacc[ch][lane]);
        atomicAdd(&a.cnt[id], cn);
      }
    }
  }
}
template <int C, int SPLIT, bool FAST>
__global__ void __launch_bounds__(WAVE) trace_weights_kernel(const BlendArgs a) {
  run_work_queue<SPLIT>(a, false, [&](uint32_t tile, uint32_t quad, bool) __attribute__((always_inline)) { trace_item<C, SPLIT, FAST>(a, tile, quad); });
}

// Work list of the backward blend: tiles ordered by the work the FORWARD blend measured for them (entries evaluated,
// summed over the four quadrants; the backward revisits the same (pixel, entry) pairs), heaviest first, in buckets of
// 16 entries; tiles in which the forward evaluated nothing have no contributor anywhere and are dropped.  The list
// length, which orders the forward's own work list, is a poor predictor because of early termination.
// One 1024-thread block (T is a few thousand); the order inside a bucket depends on LDS-atomic timing: scheduling
// only, never results.
constexpr int BWD_BUCKETS = 256;
// A tile whose work comes close to a workgroup's fair share of the whole launch would decide the run time on its own
// (the kernel ends with its longest tile: measured 505 K of 509 K cycles).  Such tiles are cut into two items, each
// a workgroup that covers TWO quadrants with two waves per quadrant (8x4 pixels per wave): half the tile per
// workgroup, and the finer culling shortens the waves' lists as well.  `workgroups` = size of the launch.
// GSR_FLAG_CLEAR_GRADS: the four arrays the blend backward accumulates into are cleared by EXTRA blocks of this launch
// (blocks 1 .. gridDim.x - 1; block 0 is the work list): the list is one block of dependent round trips during which the
// rest of the chip idles.  Measured: work list 8.2 us + a separate fill of 44 bytes per Gaussian 8.9 us -> 14 us together
// (the fill runs at 3.1 TB/s here against 4.9 TB/s alone), and one launch less between the forward and the backward.
struct ClearArgs {
  float* ptr[4];
  long long n[4];  // floats
};
template <bool SEG>
__global__ void __launch_bounds__(1024) backward_worklist_kernel(int T, const uint32_t* __restrict__ est,
                                                                uint32_t* __restrict__ order, uint32_t* __restrict__ meta,
                                                                uint32_t workgroups, int allow_halves,  // allow_halves: 0, or the threshold in 1/8 of a fair share
                                                                const ClearArgs clear, const uint32_t* __restrict__ tile_maxc,
                                                                const uint32_t* __restrict__ ck_table,
                                                                const uint32_t* __restrict__ ck_work, const CkTable ckp,
                                                                uint32_t slots,   // checkpoint slots in use per tile (<= CK_MAX)
                                                                int seg_share) {  // 0, or the threshold in 1/8 of a fair share
  if (blockIdx.x != 0) {
    const long long nb = (long long)gridDim.x - 1, b = (long long)blockIdx.x - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float* const p = clear.ptr[k];
      const long long n = clear.n[k];
      if (p == nullptr || n <= 0) continue;
      // 16-byte stores over the aligned middle, scalar stores at the two ends
      const long long head = min(n, (long long)(((16u - (unsigned)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u) / 4u));
      const long long n4 = (n - head) / 4;
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4* const q = reinterpret_cast<f4*>(p + head);
      const f4 z = {0.f, 0.f, 0.f, 0.f};
      // consecutive blocks take consecutive 64 KB pieces (4 x 16 KB per thread block and round), four stores in flight per thread
      for (long long i0 = b * 4096; i0 < n4; i0 += nb * 4096) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long i = i0 + u * 1024 + threadIdx.x;
          if (i < n4) q[i] = z;
        }
      }
      if (b == 0) {
        for (long long i = threadIdx.x; i < head; i += 1024) p[i] = 0.f;
        for (long long i = head + 4 * n4 + threadIdx.x; i < n; i += 1024) p[i] = 0.f;
      }
    }
    return;
  }
  // (BWD_SUB counters per bucket, chosen by the lane: see tile_worklist_kernel in gsr_binning.hip)
  constexpr int BWD_SUB = 16, NCNT = (BWD_BUCKETS + 1) * BWD_SUB;
  constexpr int CK_TILES_MAX = CK_TILES_CAP;  // (ck_tiles(T) at most: the ranks of Image::ck_table)
  static_assert(BWD_BUCKETS <= 256, "a segment item's bucket is cached in one byte");
  __shared__ uint32_t cnt[NCNT];
  __shared__ uint32_t smem[1024 / 64 + 1];
  __shared__ uint32_t s_threshold, s_seg_threshold, s_ncand;
  // List segments (SEG): what the three passes below need of a cut tile, by the tile's checkpoint rank -- filled ONCE
  // (round 6; rounds 4-5 re-read the tile's row of ck_work in every pass, one dependent round trip per tile and pass in a
  // kernel that is ONE block between two blend kernels: 21 us where the kernel without segments takes 9).
  constexpr uint32_t SEG_CUT = 0x80000000u;
  __shared__ uint32_t s_tile[SEG ? CK_TILES_MAX : 1];         // tile | strides << 20 | SEG_CUT; 0 = the rank's tile is no candidate
  __shared__ uint16_t s_queue[SEG ? CK_TILES_MAX : 1];        // ranks of the candidates
  __shared__ uint8_t s_bucket[SEG ? CK_TILES_MAX : 1][CK_MAX];  // bucket of every stride's item
  const uint32_t sub = threadIdx.x & (BWD_SUB - 1);
  for (int i = threadIdx.x; i < NCNT; i += 1024) cnt[i] = 0;
  if (SEG) {
    for (int i = threadIdx.x; i < CK_TILES_MAX; i += 1024) s_tile[i] = 0u;
    if (threadIdx.x == 0) s_ncand = 0u;
  }
  // The kernel is one block between the forward and the backward blend, i.e. a chain of dependent round trips: the
  // forward's counts are fetched ONCE (all loads of a thread in flight together; images of up to 1024 * EST_REG tiles keep
  // them in registers, larger ones re-read) and the counters are scanned by one block scan.
  constexpr int EST_REG = 8;
  const bool in_regs = T <= 1024 * EST_REG;
  uint4 er[EST_REG];
  uint32_t deep[EST_REG];  // how far the backward walks the tile (0 unless deeper than one checkpoint stride)
  uint32_t rank[EST_REG];  // the tile's checkpoint rank (CK_NONE: it owns no slots)
  const uint32_t stride = slots >= 2u ? ckp.pos(1u) : 0u;  // the first checkpoint: a tile the backward walks no deeper is never cut
  const bool segments = SEG && ck_table != nullptr && tile_maxc != nullptr && ck_work != nullptr && stride != 0u && slots >= 2u && seg_share > 0;
#pragma unroll
  for (int j = 0; j < EST_REG; ++j) {
    const int t = (int)threadIdx.x + 1024 * j;
    er[j] = (in_regs && T > 0) ? reinterpret_cast<const uint4*>(est)[t < T ? t : T - 1] : make_uint4(0u, 0u, 0u, 0u);
    deep[j] = (in_regs && segments && T > 0) ? tile_maxc[t < T ? t : T - 1] : 0u;
    rank[j] = (in_regs && segments && T > 0) ? ck_table[t < T ? t : T - 1] : CK_NONE;
    if (t >= T) {
      er[j] = make_uint4(0u, 0u, 0u, 0u);
      deep[j] = 0u;
      rank[j] = CK_NONE;
    }
  }
  // total work -> the weight above which a tile is cut
  uint32_t mine = 0;
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < EST_REG; ++j) mine += er[j].x + er[j].y + er[j].z + er[j].w;
  } else {
    for (int t = threadIdx.x; t < T; t += 1024) {
      const uint4 e = reinterpret_cast<const uint4*>(est)[t];
      mine += e.x + e.y + e.z + e.w;
    }
  }
  uint32_t total;
  (void)block_excl_scan_u32<1024>(mine, &total, smem);
  if (threadIdx.x == 0) {
    s_threshold = allow_halves ? max(64u, (uint32_t)((uint64_t)total * (uint32_t)allow_halves / (8u * max(workgroups, 1u)))) : 0xffffffffu;
    s_seg_threshold = segments ? max(64u, (uint32_t)((uint64_t)total * (uint32_t)seg_share / (8u * max(workgroups, 1u)))) : 0xffffffffu;
  }
  __syncthreads();
  const uint32_t threshold = s_threshold, seg_threshold = s_seg_threshold;
  auto bucket_of = [](uint32_t w) -> uint32_t {
    if (w == 0) return BWD_BUCKETS;  // nothing to do: after the end of the list
    return (uint32_t)(BWD_BUCKETS - 1) - min((w - 1u) / 16u, (uint32_t)(BWD_BUCKETS - 1));
  };
  // every tile of this thread with what the passes need of it: fn(tile, its four counts, its checkpoint rank, its depth)
  auto for_each_tile = [&](auto&& fn) {
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < EST_REG; ++j) {
        const int t = (int)threadIdx.x + 1024 * j;
        if (t < T) fn(t, er[j], rank[j], deep[j]);
      }
    } else {
      for (int t = threadIdx.x; t < T; t += 1024)
        fn(t, reinterpret_cast<const uint4*>(est)[t], segments ? ck_table[t] : CK_NONE, segments ? tile_maxc[t] : 0u);
    }
  };
  if (segments) {
    // List segments: a tile the backward walks deeper than one checkpoint stride AND whose work is a sizeable share of a
    // workgroup's becomes one item per stride, provided it owns checkpoint slots (the forward then left the state every
    // segment starts from); the items run in different workgroups at the same time.  (a) the candidates are queued; (b) ONE
    // thread per candidate fetches the tile's counts and its whole row of ck_work at once -- entry k: what the forward had
    // evaluated in the tile when it reached checkpoint k, so stride j holds entries k = j .. j + 1 of it: the front strides
    // hold most of the work, the deep positions of a list are walked for a few stragglers -- and decides: a cut only pays
    // if it really divides the work (a tile whose largest stride holds more than 4/5 of it stays whole).
    for_each_tile([&](int t, const uint4 e, uint32_t rk, uint32_t dp) {
      const uint32_t w = e.x + e.y + e.z + e.w;
      if (rk == CK_NONE || rk >= (uint32_t)CK_TILES_MAX || w < seg_threshold || dp <= stride) return;
      uint32_t ns = 1u;  // segments the backward's walk reaches into: one more than the checkpoints in front of its end
#pragma unroll
      for (uint32_t k = 1u; k < (uint32_t)CK_MAX; ++k) ns += (k < slots && ckp.pos(k) < dp) ? 1u : 0u;
      s_tile[rk] = (uint32_t)t | (ns << 20);
      s_queue[atomicAdd(&s_ncand, 1u)] = (uint16_t)rk;
    });
    __syncthreads();
    const uint32_t ncand = s_ncand;
    for (uint32_t c = threadIdx.x; c < ncand; c += 1024) {
      const uint32_t rk = s_queue[c], tt = s_tile[rk];
      const uint32_t t = tt & BWD_ITEM_TILE, ns = (tt >> 20) & 31u;
      const uint4 e = reinterpret_cast<const uint4*>(est)[t];
      const uint32_t* row = ck_work + (size_t)t * slots;
      uint32_t at[CK_MAX + 1];  // at[j]: entries evaluated in front of stride j (at[0] = 0, at[j >= ns] = the tile's total)
#pragma unroll
      for (uint32_t j = 1; j < (uint32_t)CK_MAX; ++j) at[j] = row[min(j, slots - 1u)];  // (unconditional: all in flight)
      const uint32_t w = e.x + e.y + e.z + e.w;
      at[0] = 0u;
      at[CK_MAX] = w;
#pragma unroll
      for (uint32_t j = 1; j < (uint32_t)CK_MAX; ++j) at[j] = j < ns ? min(at[j], w) : w;
      uint32_t largest = 0u;
#pragma unroll
      for (uint32_t j = 0; j < (uint32_t)CK_MAX; ++j) {
        const uint32_t sw = max(at[j + 1] > at[j] ? at[j + 1] - at[j] : 0u, 1u);
        if (j < ns) {
          largest = max(largest, sw);
          s_bucket[rk][j] = (uint8_t)bucket_of(sw);
        }
      }
      if (ns > 1u && (uint64_t)largest * 5u <= (uint64_t)w * 4u) s_tile[rk] = tt | SEG_CUT;
    }
    __syncthreads();
  }
  // strides of a tile that is cut (0: it is not)
  auto strides_of = [&](uint32_t rk) -> uint32_t {
    if (!segments || rk == CK_NONE || rk >= (uint32_t)CK_TILES_MAX) return 0u;
    const uint32_t tt = s_tile[rk];
    return (tt & SEG_CUT) ? ((tt >> 20) & 31u) : 0u;
  };
  for_each_tile([&](int t, const uint4 e, uint32_t rk, uint32_t) {
    const uint32_t w = e.x + e.y + e.z + e.w;
    const uint32_t ns = strides_of(rk);
    if (ns != 0u) {
      for (uint32_t j = 0; j < ns; ++j) atomicAdd(&cnt[(uint32_t)s_bucket[rk][j] * BWD_SUB + sub], 1u);
    } else if (w >= threshold) {
      atomicAdd(&cnt[bucket_of(e.x + e.y) * BWD_SUB + sub], 1u);
      atomicAdd(&cnt[bucket_of(e.z + e.w) * BWD_SUB + sub], 1u);
    } else {
      atomicAdd(&cnt[bucket_of(w) * BWD_SUB + sub], 1u);
    }
  });
  __syncthreads();
  {  // exclusive scan over the counters in (bucket, sub) order: counts -> cursors (one block scan, CPT counters a thread)
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
      if (i0 + j == BWD_BUCKETS * BWD_SUB) meta[0] = run;  // number of items with work
      run += v[j];
    }
  }
  __syncthreads();
  for_each_tile([&](int t, const uint4 e, uint32_t rk, uint32_t) {
    const uint32_t w = e.x + e.y + e.z + e.w;
    const uint32_t ns = strides_of(rk);
    if (ns != 0u) {
      for (uint32_t j = 0; j < ns; ++j)  // (item code: first and last stride of the run -- here one stride)
        order[atomicAdd(&cnt[(uint32_t)s_bucket[rk][j] * BWD_SUB + sub], 1u)] =
            (uint32_t)t | BWD_ITEM_SEG | (j << BWD_SEG_SHIFT) | (j << BWD_NSEG_SHIFT);
    } else if (w >= threshold) {
      order[atomicAdd(&cnt[bucket_of(e.x + e.y) * BWD_SUB + sub], 1u)] = (uint32_t)t | BWD_ITEM_HALF;
      order[atomicAdd(&cnt[bucket_of(e.z + e.w) * BWD_SUB + sub], 1u)] = (uint32_t)t | BWD_ITEM_HALF | BWD_ITEM_PART;
    } else {
      order[atomicAdd(&cnt[bucket_of(w) * BWD_SUB + sub], 1u)] = (uint32_t)t;
    }
  });
}

// Compute units of the device a launch goes to: the device of the STREAM (a C-ABI caller may hand over a stream of another
// device than the thread's current one), looked up per call, the attribute cached per device ordinal.
static int cus_of_stream(hipStream_t s) {
  static int cache[64] = {};  // 0 = not asked yet (a benign race: every writer stores the same value)
  int dev = 0;
  hipDevice_t sdev = 0;
  if (s != nullptr && hipStreamGetDevice(s, &sdev) == hipSuccess) dev = (int)sdev;
  else (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (cache[dev] == 0) {
    int c = 0;
    (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
    cache[dev] = c > 0 ? c : 256;
  }
  return cache[dev];
}
// Number of persistent waves of a launch: (SIMDs on the device) x (waves per SIMD), 4 by default.
// GSR_BLEND_WAVES_PER_SIMD overrides the default for all blend kernels, GSR_FWD_WAVES_PER_SIMD for the forward / trace
// kernels only (the backward is built for exactly 4: amdgpu_waves_per_eu) -- tuning knobs, read once.
static int waves_per_simd(bool backward, bool shared_simds) {
  static const int all = [] { const char* e = getenv("GSR_BLEND_WAVES_PER_SIMD"); return e ? atoi(e) : 0; }();
  static const int fwd = [] { const char* e = getenv("GSR_FWD_WAVES_PER_SIMD"); return e ? atoi(e) : 0; }();
  // (GSR_FLAG_SHARED_SIMDS: a second stream's kernels run alongside -- 2 waves per SIMD; the environment knobs win)
  const int v = (!backward && fwd > 0) ? fwd : (all > 0 ? all : (shared_simds ? 2 : 4));
  return v < 1 ? 1 : (v > 8 ? 8 : v);
}
unsigned blend_grid_size(bool backward, hipStream_t s, bool shared_simds) {
  // (GSR_FWD_GRID: development knob, any number of persistent forward waves -- tools/microbench, profiles/r02_e)
  static const unsigned fwd_fixed = [] { const char* e = getenv("GSR_FWD_GRID"); return e && atoi(e) > 0 ? (unsigned)atoi(e) : 0u; }();
  if (!backward && fwd_fixed != 0u) return fwd_fixed;
  return (unsigned)cus_of_stream(s) * 4u * (unsigned)waves_per_simd(backward, shared_simds);
}
// Placement units of a launch with `waves_per_wg`-wave workgroups (see first_item_of_block): SIMDs or CUs; 0 turns the
// assigned first items off (GSR_BLEND_FOLD=0, or a CU count the fold does not divide).
static unsigned blend_units(unsigned waves_per_wg, hipStream_t s, bool shared_simds) {
  static const bool fold = [] { const char* e = getenv("GSR_BLEND_FOLD"); return !e || atoi(e) != 0; }();
  const unsigned cus = (unsigned)cus_of_stream(s);
  const unsigned units = waves_per_wg == 1 ? cus * 4u : cus;
  const unsigned grid = blend_grid_size(waves_per_wg != 1, s, shared_simds) / waves_per_wg;
  if (!fold || units % 8u != 0u || grid % units != 0u) return 0u;
  return units;
}
// The cursors are zero on entry (cleared by tile_worklist_kernel, then by every launch's last workgroup).
// GSR_QUEUE_MEMSET=1 (timing experiments), or a grid too large for the retire counters: clear them with a memset
// before the launch instead.
static hipError_t prepare_queue(hipStream_t s, BlendArgs& a, unsigned grid) {
  static const bool env_memset = [] { const char* e = getenv("GSR_QUEUE_MEMSET"); return e && atoi(e) != 0; }();
  const bool use_memset = env_memset || (grid + 63u) / 64u > (unsigned)QUEUE_GROUPS;
  a.self_reset = use_memset ? 0 : 1;
  return use_memset ? hipMemsetAsync(a.queue, 0, sizeof(uint32_t) * QUEUE_STRIDE * QUEUE_LINES, s) : hipSuccess;
}
hipError_t launch_blend_forward(hipStream_t s, BlendArgs a) {
  const bool sh = a.shared_simds != 0;
  hipError_t e = prepare_queue(s, a, blend_grid_size(false, s, sh));
  if (e != hipSuccess) return e;
  a.units = (int)blend_units(1, s, sh);
  static const bool split_ok = [] { const char* e = getenv("GSR_FWD_SPLIT"); return !e || atoi(e) != 0; }();
  a.allow_split = split_ok ? 1 : 0;
  // fewer than two quadrant items per persistent wave (bounded by the tile count of the image): cut the quadrants
  const unsigned grid = blend_grid_size(false, s, sh), quads = 4u * (unsigned)(a.gx * a.gy);
  // GSR_FWD_SPLIT_FORCE=1|2|4 (sweeps): the cut of the forward's quadrants whatever the image
  static const int split_force = [] { const char* e = getenv("GSR_FWD_SPLIT_FORCE"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 0; }();
  // A render that will see a backward cuts later (round 6, profiles/r06_s_forward_split.md): the cut costs the BACKWARD --
  // a quadrant's work estimate is then the largest count among its sub-items, the tile's checkpoint row their sum, and
  // the backward's work list orders and cuts by both -- 5-35 % of K7 on images of 320 x 320 .. 720 x 720, more than the
  // forward gains.  Forward-only renders keep the cut that is best for the forward alone.
  const int split_alone = quads >= 2u * grid ? 1 : (2u * quads >= 2u * grid ? 2 : 4);
  const int split_train = quads > grid ? 1 : (4u * quads > grid ? 2 : 4);
  const int split = split_force ? split_force : (!a.allow_split ? 1 : (a.for_backward ? split_train : split_alone));
  const dim3 g(grid), b(WAVE);
#define GSR_FWD_LAUNCH(AUXV, FASTV, CKV)                                                                        \
  do {                                                                                                          \
    if (split == 1) hipLaunchKernelGGL((blend_forward_kernel<false, AUXV, 1, FASTV, CKV>), g, b, 0, s, a);       \
    else if (split == 2) hipLaunchKernelGGL((blend_forward_kernel<false, AUXV, 2, FASTV, CKV>), g, b, 0, s, a);  \
    else hipLaunchKernelGGL((blend_forward_kernel<false, AUXV, 4, FASTV, CKV>), g, b, 0, s, a);                  \
  } while (0)
  // checkpoints for the backward's list segments: where the caller handed the tables over (gsr_capi.hip: checkpoint_chunks)
  const bool ck = a.ck_table != nullptr && a.ck_chunks > 0 && a.colors3 == nullptr && !a.profile;
  if (a.profile)
    hipLaunchKernelGGL((blend_forward_kernel<true, false, 1, false, false>), g, b, 0, s, a);
  else if (a.colors3 != nullptr) {
    if (a.fast_exp) GSR_FWD_LAUNCH(true, true, false); else GSR_FWD_LAUNCH(true, false, false);
  } else if (ck) {
    if (a.fast_exp) GSR_FWD_LAUNCH(false, true, true); else GSR_FWD_LAUNCH(false, false, true);
  } else {
    if (a.fast_exp) GSR_FWD_LAUNCH(false, true, false); else GSR_FWD_LAUNCH(false, false, false);
  }
#undef GSR_FWD_LAUNCH
  return hipGetLastError();
}
hipError_t launch_blend_backward(hipStream_t s, BlendArgs a) {
  const bool sh = a.shared_simds != 0;
  hipError_t e = prepare_queue(s, a, blend_grid_size(true, s, sh) / BWD_WAVES);
  if (e != hipSuccess) return e;
  a.units = (int)blend_units(BWD_WAVES, s, sh);
  // GSR_BWD_ABLATE (debug, timing experiments only): 1 no wave reduction, 2 no atomics, 3 neither, 4 footprint only,
  // 5 no workgroup barrier per chunk (the quadrant waves drift apart), 6 = 5 without the global atomics
  static const int ablate = [] { const char* e = getenv("GSR_BWD_ABLATE"); return e ? atoi(e) : 0; }();
  bool seg_items = false;  // the work list holds list-segment items (views whose forward left checkpoints)
  ClearArgs clear = {{nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}};
  if (a.clear_grads) {
    const long long P = a.P;
    clear.ptr[0] = a.acc; clear.n[0] = (long long)ACC_ROW * P;
  }
  if (a.touched != nullptr) {  // the row mask always starts from zero (P bytes rounded up to whole floats: gsr.h asks for the padding)
    clear.ptr[1] = reinterpret_cast<float*>(a.touched); clear.n[1] = ((long long)a.P + 3) / 4;
  }
  if (a.work_est == nullptr || a.work_maxc == nullptr || a.bwd_order == nullptr) return hipErrorInvalidValue;
  {
    // its own work list, ordered by the work the forward measured
    static const int halves = [] { const char* e = getenv("GSR_BWD_HALVES"); return e ? atoi(e) : 10; }();  // tiles above 1.25 fair shares: measured best (sweep 6..16)
    const unsigned fill_blocks = a.clear_grads ? 2u * (unsigned)cus_of_stream(s) : (a.touched != nullptr ? 16u : 0u);  // (1 .. 8 per CU: the same 14 us)
    // GSR_BWD_SEG: tiles above this many eighths of a fair share are cut into list segments where the forward left
    // checkpoints (0: never; tests use 1)
    static const int seg_share = [] { const char* e = getenv("GSR_BWD_SEG"); return e ? atoi(e) : 5; }();
    // (Merging consecutive strides of a cut tile into items of about equal measured work -- the plan of
    //  profiles/r06_k -- was built and swept over eleven workloads at 2 .. 12 sixteenths of a fair share per item: equal at
    //  best, 8-40 % slower where pixels walk deep; removed: profiles/r06_m_fine_checkpoints.md.)
    seg_items = a.ck_table != nullptr && a.ck_chunks > 0 && seg_share > 0 && ablate == 0;
    if (seg_items)
      hipLaunchKernelGGL(backward_worklist_kernel<true>, dim3(1u + fill_blocks), dim3(1024), 0, s, a.gx * a.gy, a.work_est,
                         a.bwd_order, a.bwd_meta, blend_grid_size(true, s, sh) / BWD_WAVES, halves, clear, (const uint32_t*)a.tile_maxc,
                         (const uint32_t*)a.ck_table, (const uint32_t*)a.ck_work, a.ck_pos, (uint32_t)a.ck_slots, seg_share);
    else
      hipLaunchKernelGGL(backward_worklist_kernel<false>, dim3(1u + fill_blocks), dim3(1024), 0, s, a.gx * a.gy, a.work_est,
                         a.bwd_order, a.bwd_meta, blend_grid_size(true, s, sh) / BWD_WAVES, halves, clear, (const uint32_t*)nullptr,
                         (const uint32_t*)nullptr, (const uint32_t*)nullptr, CkTable{}, 0u, 0);
  }
  // #CUs x 4 workgroups of 4 waves: the same 4 waves per SIMD as the forward
  const dim3 g(blend_grid_size(true, s, sh) / BWD_WAVES), b(WAVE * BWD_WAVES);
  switch (ablate) {
    case 1: hipLaunchKernelGGL((blend_backward_kernel<1, false, false>), g, b, 0, s, a); break;
    case 2: hipLaunchKernelGGL((blend_backward_kernel<2, false, false>), g, b, 0, s, a); break;
    case 3: hipLaunchKernelGGL((blend_backward_kernel<3, false, false>), g, b, 0, s, a); break;
    case 4: hipLaunchKernelGGL((blend_backward_kernel<4, false, false>), g, b, 0, s, a); break;
    case 5: hipLaunchKernelGGL((blend_backward_kernel<5, false, false>), g, b, 0, s, a); break;
    case 6: hipLaunchKernelGGL((blend_backward_kernel<6, false, false>), g, b, 0, s, a); break;
    default:
      if (seg_items) {
        if (a.fast_exp) hipLaunchKernelGGL((blend_backward_kernel<0, true, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((blend_backward_kernel<0, false, true>), g, b, 0, s, a);
      } else {
        if (a.fast_exp) hipLaunchKernelGGL((blend_backward_kernel<0, true, false>), g, b, 0, s, a);
        else hipLaunchKernelGGL((blend_backward_kernel<0, false, false>), g, b, 0, s, a);
      }
      break;
  }
  return hipGetLastError();
}
hipError_t launch_trace_weights(hipStream_t s, BlendArgs a) {
  const bool sh = a.shared_simds != 0;
  hipError_t e = prepare_queue(s, a, blend_grid_size(false, s, sh));
  if (e != hipSuccess) return e;
  a.units = (int)blend_units(1, s, sh);
  const unsigned grid = blend_grid_size(false, s, sh), quads = 4u * (unsigned)(a.gx * a.gy);
  static const bool split_ok = [] { const char* e = getenv("GSR_FWD_SPLIT"); return !e || atoi(e) != 0; }();
  const int split = !split_ok || quads >= 2u * grid ? 1 : (2u * quads >= 2u * grid ? 2 : 4);  // as the forward
  const dim3 g(grid), b(WAVE);
#define GSR_TRACE_LAUNCH2(CC, FASTV)                                                                    \
  do {                                                                                                  \
    if (split == 1) hipLaunchKernelGGL((trace_weights_kernel<CC, 1, FASTV>), g, b, 0, s, a);             \
    else if (split == 2) hipLaunchKernelGGL((trace_weights_kernel<CC, 2, FASTV>), g, b, 0, s, a);        \
    else hipLaunchKernelGGL((trace_weights_kernel<CC, 4, FASTV>), g, b, 0, s, a);                        \
  } while (0)
#define GSR_TRACE_LAUNCH(CC)                                                         \
  do {                                                                               \
    if (a.fast_exp) GSR_TRACE_LAUNCH2(CC, true); else GSR_TRACE_LAUNCH2(CC, false);  \
  } while (0)
  switch (a.C) {
    case 1: GSR_TRACE_LAUNCH(1); break;
    case 2: GSR_TRACE_LAUNCH(2); break;
    case 3: GSR_TRACE_LAUNCH(3); break;
    default: return hipErrorInvalidValue;
  }
#undef GSR_TRACE_LAUNCH2
#undef GSR_TRACE_LAUNCH
  return hipGetLastError();
}

}  // namespace gsr
