// gsr_common.h -- shared definitions for the gfx950 Gaussian-splatting rasterizer kernels.
//
// Target: AMD Instinct MI355X (CDNA4, gfx950) only.  wave = 64 lanes; one
// rasterizer "pixel wave" covers an 8x8 pixel quadrant of a 16x16 tile.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/gsr.h"

namespace gsr {

constexpr int TILE = 16;        // tile edge in pixels (reference: BLOCK_X/BLOCK_Y, config.h:16-17)
constexpr int QUAD = 8;         // one wave = 8x8 pixels
constexpr int WAVE = 64;
constexpr int GAUSS_BLOCK = 256;  // Gaussians per block in the per-Gaussian kernels
// The blend backward's accumulators: ONE 64-byte row per Gaussian, so that the (up to nine) float atomics of a (tile,
// Gaussian) pair leave as one memory-side request (gsr_blend.hip: flush).  Columns as include/gsr.h states them:
// [0] [1] dL_dmean2D.xy  [3] dL_dopacity  [4] [5] [7] dL_dconic.x .y .w  [8..10] dL_dcolor; 2, 6, 11..15 stay zero.
constexpr uint32_t ACC_ROW = GSR_ACC_ROW, ACC_MEAN2D = GSR_ACC_MEAN2D, ACC_OPACITY = GSR_ACC_OPACITY, ACC_CONIC = GSR_ACC_CONIC,
                   ACC_COLOR = GSR_ACC_COLOR;
static_assert(ACC_ROW == 16 && ACC_MEAN2D == 0 && ACC_OPACITY == 3 && ACC_CONIC == 4 && ACC_COLOR == 8, "float4-aligned column groups");
// Binning works on GROUPS of 8 x 8 tiles (128 x 128 pixels): a Gaussian's tile rectangle inside one group is a 64-bit
// mask, one bit per tile, bit = (tile_y & 7) * 8 + (tile_x & 7) (gsr_binning.hip).
constexpr int GROUP_SHIFT = 3;
constexpr int GROUP_EDGE = 1 << GROUP_SHIFT;
constexpr int GROUP_TILES = GROUP_EDGE * GROUP_EDGE;  // 64 = one wave: lane t owns tile t of the group
constexpr int GROUP_MAX = 2048;                // images with more groups (> 131 072 tiles) take the two-pass tile sort
constexpr uint32_t GROUP_PAD = 0xffffffffu;    // key held by the padding slots between the groups' segments

// ----------------------------------------------------------------------------------
// Opaque scratch layouts.  Every section is 256-byte aligned.
// ----------------------------------------------------------------------------------
__host__ __device__ inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Per-Gaussian state ("geomBuffer").  The three float4 records are what the blend
// kernels gather per instance (48 B, 16-byte aligned vector loads):
//   rec0 = conic.x, conic.y, conic.z, opacity     (reference: conic_opacity, forward.cu:254)
//   rec1 = mean2D.x, mean2D.y, depth, radius(float) (reference: points_xy_image, depths, radii)
//   rec2 = r, g, b, 0                              (reference: rgb, or a copy of colors_precomp)
struct Geom {
  float4* rec0;
  float4* rec1;
  float4* rec2;
  // d(colour)/d(view direction) of the SH colour, three floats per direction component (x, y, z): K1 has the
  // Gaussian's SH record in registers anyway, K8+K9 needs nothing else of it (backward.cu:88-136) -- 36 bytes stored and
  // re-read instead of the 12 M-byte record streamed a second time (192 B at M = 16).  Written only for Gaussians that
  // survive every cull, only when the backward will run (GSR_FLAG_FORWARD_ONLY) and only for D > 0 (degree 0 has no
  // direction dependence: K8+K9 takes zeros without reading).  Round 3 kept float4s (48 B).
  float* dcol[3];
  uint2* rect;           // (P)     tile rectangle, written for every Gaussian: x = minx | miny << 16, y = width | height << 16 (0 if culled)
  uint8_t* clamped;      // (P)     bit ch set <=> SH colour channel ch was clamped at 0
  uint32_t* block_sums;  // (nb)    sum of tiles_touched per 256-Gaussian block, in DEPTH-SORTED Gaussian order
  uint32_t* block_offs;  // (nb)    exclusive prefix of block_sums
  uint64_t* total;       // header (GEOM_HDR_BYTES): word GEOM_HDR_FINAL only, written before it is read -- never cleared
  uint4* k1_partials;    // (nb)    per K1 block: sum of tiles_touched, max depth key, max complemented key, sum of groups touched
  // depth ordering of the Gaussians (stable LSD radix sort of (depth bits, index); culled Gaussians last)
  uint32_t* dkey[2];     // (P)     ping-pong depth keys
  uint32_t* dval[2];     // (P)     ping-pong Gaussian indices; dval[header FINAL] holds the final order
  // grouped path (round 4): the tile rectangle travels with the sort as a second 32-bit payload (pack_rect32), so that
  // nothing is gathered afterwards; after the LAST pass drect[FINAL] holds the rectangles in depth order and dkey[FINAL]
  // -- whose sorted keys nobody reads -- the exclusive prefix of the Gaussians' group counts in depth order
  uint32_t* drect[2];    // (P)     ping-pong packed tile rectangles
  uint32_t* ghist;       // (256 * ceil(P/4096)) per-block digit counts of the depth sort
  uint32_t* gbin_total;  // (512)
  size_t bytes;
};

// Geom header.  K1 leaves one partial per block in k1_partials (plain stores: nothing to clear beforehand, no atomics --
// memory-side atomics on ONE line serialise at ~6 ns each, 3 x 4096 of them cost 80 us); block 0 of the first depth-sort
// kernel reduces them and writes num_rendered and the key range into the host's pinned slot: u32 words there
constexpr int GEOM_HDR_SLOTS = 64;
constexpr int GEOM_HDR_SLOT_WORDS = 32;
constexpr int GEOM_HDR_KEYMAX = 2;     // [0..1] num_rendered (u64); max depth key over the visible Gaussians
constexpr int GEOM_HDR_KEYINVMAX = 3;  // max of the complemented key (= ~min key)
constexpr int GEOM_HDR_FINAL = 4;      // device header only: which ping-pong side (dval[]) holds the depth order
constexpr int GEOM_HDR_GROUPS = 5;     // host slot only: [5..6] number of group instances (u64)
constexpr int GEOM_HDR_BYTES = GEOM_HDR_SLOTS * GEOM_HDR_SLOT_WORDS * 4;

__host__ __device__ inline Geom carve_geom(void* base, int P) {
  char* p = (char*)base;
  const size_t nb = ((size_t)P + GAUSS_BLOCK - 1) / GAUSS_BLOCK;
  Geom g;
  size_t off = 0;
  g.total = (uint64_t*)(p + off);      off += align_up(GEOM_HDR_BYTES);
  g.k1_partials = (uint4*)(p + off);   off += align_up(sizeof(uint4) * nb);
  g.rec0 = (float4*)(p + off);         off += align_up(sizeof(float4) * (size_t)P);
  g.rec1 = (float4*)(p + off);         off += align_up(sizeof(float4) * (size_t)P);
  g.rec2 = (float4*)(p + off);         off += align_up(sizeof(float4) * (size_t)P);
  for (int i = 0; i < 3; ++i) { g.dcol[i] = (float*)(p + off); off += align_up(sizeof(float) * 3 * (size_t)P); }
  g.rect = (uint2*)(p + off);          off += align_up(sizeof(uint2) * (size_t)P);
  g.clamped = (uint8_t*)(p + off);     off += align_up((size_t)P);
  g.block_sums = (uint32_t*)(p + off); off += align_up(sizeof(uint32_t) * nb);
  g.block_offs = (uint32_t*)(p + off); off += align_up(sizeof(uint32_t) * nb);
  const size_t nsb = ((size_t)P + 4096 - 1) / 4096;
  for (int i = 0; i < 2; ++i) { g.dkey[i] = (uint32_t*)(p + off); off += align_up(sizeof(uint32_t) * (size_t)P); }
  for (int i = 0; i < 2; ++i) { g.dval[i] = (uint32_t*)(p + off); off += align_up(sizeof(uint32_t) * (size_t)P); }
  for (int i = 0; i < 2; ++i) { g.drect[i] = (uint32_t*)(p + off); off += align_up(sizeof(uint32_t) * (size_t)P); }
  g.ghist = (uint32_t*)(p + off);      off += align_up(sizeof(uint32_t) * 512 * nsb);
  g.gbin_total = (uint32_t*)(p + off); off += align_up(sizeof(uint32_t) * 512);
  g.bytes = off;
  return g;
}

// Per-pixel / per-tile state ("imgBuffer").
constexpr int WORK_BUCKETS = 128;     // tiles are bucketed by ceil(len/64), longest first
constexpr int QUEUE_STRIDE = 32;      // u32 words between queue heads (128 B: one head per cache line)
constexpr int QUEUE_KINDS = 3;        // forward, backward, trace
constexpr int QUEUE_GROUPS = 128;     // retire counters: one per 64 workgroups of a launch (grids of up to 8192)
constexpr int QUEUE_LINES = 8 + 1 + QUEUE_GROUPS;  // per kind: 8 per-XCD heads, the top retire counter, the group counters
struct Image {
  uint2* ranges;        // (T)  [begin,end) into point_list
  float* final_T;       // (N)
  uint32_t* n_contrib;  // (N)
  uint32_t* work_order; // (T)  tile ids: non-empty tiles, longest list first (bucketed), then the empty tiles
  uint32_t* work_meta;  // [0] = number of non-empty tiles
  uint32_t* work_est;   // (T,4) entries the forward blend evaluated per (tile, quadrant): the backward's work estimate
  uint32_t* work_maxc;  // (T,4) right behind work_est (one clear): the largest last-contributor position + 1 over the quadrant's
                        //      pixels, left by the forward blend -- how deep the backward walks the tile's list, known to an item
                        //      from one scalar load instead of n_contrib -> maximum over the workgroup (two barriers)
  uint32_t* bwd_order;  // (2T + CK_MAX * CK_TILES(T)) items of the backward blend (a tile, half of a heavy tile, or a list segment of a deep
                        //      one): most forward work first, tiles without any work dropped
  uint32_t* bwd_meta;   // [0] = number of items in bwd_order
  uint32_t* queue_heads;// (QUEUE_KINDS x QUEUE_LINES) work-queue cursors + retire counters, QUEUE_STRIDE words apart
  // Checkpoints of the forward blend (round 4): the backward can then walk a tile's list as independent SEGMENTS in
  // separate work items.  A checkpoint of a pixel is one float4, a checkpoint of a tile 256 of them: slot k >= 1 holds the
  // transmittance in front of list position pos(k) (CkTable below) and (round 5) the colour segment k - 1 -- positions
  // [pos(k - 1), pos(k)) -- contributed, accumulated from zero; slot 0 the colour of segment CK_MAX - 1 and everything behind it.  The
  // CK_TILES(T) tiles with the longest lists own CK_MAX consecutive 4 KB slots each, assigned by tile_worklist_kernel -- no
  // allocation, no atomics in the forward's loop.
  uint32_t* ck_table;   // (T)  the tile's rank among the checkpointed tiles (its slots: rank * S + k, S = the slots the view uses, <= CK_MAX), or CK_NONE
  uint32_t* ck_work;    // (T x S, room for S = CK_MAX) entry k: entries the forward had evaluated in the tile when it reached checkpoint k (summed
                        //      over its quadrants): how the tile's backward work splits over its list segments.  Zero per view.
  uint32_t* tile_maxc;  // (T)  largest last-contributor position + 1 over the tile's pixels (what the backward walks).  Zero per view.
  float4* ck_pool;      // (CK_TILES(T) x S x 256, room for S = CK_MAX)
  size_t bytes;
};
// Round 6 (second half): sixteen slots per tile instead of eight, at positions that are fine in front and coarse behind
// (CkTable below; gsr_capi.hip: checkpoint_table).  profiles/r06_m_fine_checkpoints.md, r06_n_checkpoint_table.md.
constexpr int CK_MAX = 16;           // slots per tile: 15 checkpoints + the tail (beyond the last one the last segment is longer)
constexpr uint32_t CK_NONE = 0xffffffffu;
constexpr int CK_CHUNKS_DEFAULT = 4;  // the first checkpoint, in 64-entry chunks (256 list positions), where checkpoints are on
// Where a view's checkpoints sit: checkpoint k (1 <= k < slots) lies in front of list position 64 * chunk[k]; chunk[0] = 0,
// ascending.  Uniform (k * stride) or -- the small-image class, round 6 -- FINE IN FRONT, COARSE BEHIND: the work of a tile
// sits where its pixels are still live, in the front of its list, and the deep positions are walked for a few stragglers;
// how deep "the front" is depends on the scene, which only the forward measures (a 512 x 512 view of 1 M Gaussians wants
// strides of 256 and 3 000 positions of reach, the same image of 3 M or a 256 x 256 one wants 12 000 of reach:
// profiles/r06_n).  Passed by value to the three kernels that read it.
struct CkTable {
  uint16_t chunk[CK_MAX];
  __host__ __device__ inline uint32_t pos(uint32_t k) const { return (uint32_t)chunk[k] * 64u; }
};
constexpr int CK_TILES_CAP = 2048;    // tiles of a view that can own checkpoint slots: the ones with the longest lists (128 MB of slots at most)
__host__ __device__ inline size_t ck_tiles(size_t T) { return T < (size_t)CK_TILES_CAP ? T : (size_t)CK_TILES_CAP; }
// with_ck_pool: false leaves the checkpoint pool out of `bytes` (it is the last section, so nothing else moves); the
// pointer is still set and must not be used then (gsr_scratch_sizes / the blend entry points share one predicate)
__host__ __device__ inline Image carve_image(void* base, int W, int H, bool with_ck_pool = true) {
  char* p = (char*)base;
  const size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
  const size_t N = (size_t)W * H;
  Image im;
  size_t off = 0;
  im.ranges = (uint2*)(p + off);        off += align_up(sizeof(uint2) * T);
  im.final_T = (float*)(p + off);       off += align_up(sizeof(float) * N);
  im.n_contrib = (uint32_t*)(p + off);  off += align_up(sizeof(uint32_t) * N);
  im.work_order = (uint32_t*)(p + off); off += align_up(sizeof(uint32_t) * T);
  im.work_meta = (uint32_t*)(p + off);  off += 256;
  im.work_est = (uint32_t*)(p + off);   off += sizeof(uint32_t) * 4 * T;  // (work_maxc follows without a gap)
  im.work_maxc = (uint32_t*)(p + off);  off = align_up(off + sizeof(uint32_t) * 4 * T);
  im.bwd_order = (uint32_t*)(p + off);  off += align_up(sizeof(uint32_t) * (2 * T + CK_MAX * ck_tiles(T)));
  im.bwd_meta = (uint32_t*)(p + off);   off += 256;
  im.queue_heads = (uint32_t*)(p + off); off += align_up(sizeof(uint32_t) * QUEUE_STRIDE * QUEUE_LINES * QUEUE_KINDS);
  im.ck_table = (uint32_t*)(p + off);   off += align_up(sizeof(uint32_t) * T);
  im.ck_work = (uint32_t*)(p + off);    off += align_up(sizeof(uint32_t) * CK_MAX * T);
  im.tile_maxc = (uint32_t*)(p + off);  off += align_up(sizeof(uint32_t) * T);
  im.ck_pool = (float4*)(p + off);      off += with_ck_pool ? align_up(sizeof(float4) * 256 * CK_MAX * ck_tiles(T)) : 0;
  im.bytes = off;
  return im;
}

// Radix sort geometry.
constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 16;                          // keys per thread
constexpr int SORT_KPB = SORT_THREADS * SORT_ITEMS;     // keys per block (4096)
constexpr int SORT_MAX_BINS = 512;                      // 9-bit digits at most

// Per-instance state ("binningBuffer").  Everything the blend kernels read -- the reference's point_list -- comes FIRST,
// so its address depends on nothing but the buffer's base.
//
// Grouped path (images of up to GROUP_MAX groups): one GROUP INSTANCE per (Gaussian, 8x8-tile group) pair is emitted in
// depth order, stably sorted by group id in ONE radix pass whose output leaves every group's segment padded to whole
// CHUNKS, and each chunk (one wave) then turns its 64-item batches into per-tile lists with a 64 x 64 bit transpose:
// lane t ends up with the set of batch items that touch tile t of the group, in order.
// Legacy path (more groups): (tile id, Gaussian) pairs, stable radix sort on the tile id in ceil(bits / 8) passes.
struct Binning {
  uint32_t* point_list;   // (R) Gaussian indices, tile by tile, each tile's in depth order: the reference's point_list
  int legacy;
  // --- grouped path
  int sgx, sgy, groups;   // groups per row / column, S = sgx * sgy
  int group_bits;         // 8 (S <= 256) or 11: digit width of the one radix pass
  int chunk;              // group instances per chunk (a multiple of 64)
  int64_t G;              // group instances
  int64_t chunks;         // upper bound of the number of chunks: floor(G / chunk) + S
  uint32_t sort_blocks;   // ceil(G / SORT_KPB)
  uint32_t* gkey[2];      // [0]: (G) group id | local rectangle << 16, in emission (= depth) order; [1]: (chunks * chunk)
                          //      sorted by group and padded: padding slots hold GROUP_PAD
  uint32_t* gval[2];      // the Gaussian index of each group instance, same shapes
  uint32_t* ghist;        // (bins * sort_blocks), bin-major
  uint32_t* gbin_total;   // (bins)
  uint32_t* group_first;  // (bins + 1) row of each group's first chunk in the per-chunk tables
  uint16_t* chunk_cnt;    // (chunks, 64) instances of tile t in chunk c
  uint32_t* chunk_pre;    // (chunks, 64) the same summed over the earlier chunks of the chunk's group
  uint32_t* tile_total;   // (T) instances per tile
  uint32_t* tile_start;   // (T + 1) exclusive prefix of tile_total in tile-id order
  // --- legacy path
  void* tkey[2];        // (R) tile ids, uint16 when they fit (key_bytes == 2: images of up to 65535 tiles), else uint32
  int key_bytes;
  uint32_t* vals[2];    // (R) Gaussian indices; vals[final_buf] is point_list
  uint32_t* hist;       // (bins * nblocks), bin-major
  uint32_t* bin_total;  // (bins)
  uint32_t nblocks;
  int tile_bits;        // getHigherMsb(T)
  int passes;           // ceil(tile_bits / 8)
  int digit_bits[4];    // bits sorted by each pass (balanced split, <= 8)
  int final_buf;        // which ping-pong side holds the sorted result (= passes & 1)
  size_t bytes;
};
__host__ __device__ inline uint32_t higher_msb(uint32_t n) {  // rasterizer_impl.cu:36-49
  uint32_t msb = sizeof(n) * 4, step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step; else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}
__host__ __device__ inline int sort_key_bits(int W, int H) {
  const uint32_t T = (uint32_t)((W + TILE - 1) / TILE) * (uint32_t)((H + TILE - 1) / TILE);
  return 32 + (int)higher_msb(T);
}
__host__ __device__ inline int64_t group_count(int W, int H) {
  const int64_t gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  return ((gx + GROUP_EDGE - 1) >> GROUP_SHIFT) * ((gy + GROUP_EDGE - 1) >> GROUP_SHIFT);
}
// A tile rectangle in 32 bits, for images of up to RECT32_EDGE tiles per side (4096 pixels): x0 | y0 << 8 | (w - 1) << 16 |
// (h - 1) << 24; RECT32_NONE = no tile (a real rectangle never has x0 = 255 AND w = 256).  It is the second payload of the
// depth sort on the grouped path; larger images take the legacy path, which gathers the 8-byte rectangles.
constexpr int RECT32_EDGE = 256;
constexpr uint32_t RECT32_NONE = 0xffffffffu;
__host__ __device__ inline uint32_t pack_rect32(uint32_t xy, uint32_t wh) {  // Geom::rect's two words
  const uint32_t w = wh & 0xffffu, h = wh >> 16;  // (branch-free: a select, so that callers' loads are not made conditional)
  const uint32_t r = (xy & 0xffu) | (((xy >> 16) & 0xffu) << 8) | (((w - 1u) & 0xffu) << 16) | (((h - 1u) & 0xffu) << 24);
  return w * h == 0u ? RECT32_NONE : r;
}
// number of 8x8-tile groups the rectangle reaches (0 for RECT32_NONE)
__host__ __device__ inline uint32_t rect32_groups(uint32_t r) {
  if (r == RECT32_NONE) return 0u;
  const uint32_t x0 = r & 0xffu, y0 = (r >> 8) & 0xffu, x1 = x0 + ((r >> 16) & 0xffu), y1 = y0 + (r >> 24);  // inclusive
  return ((x1 >> GROUP_SHIFT) - (x0 >> GROUP_SHIFT) + 1u) * ((y1 >> GROUP_SHIFT) - (y0 >> GROUP_SHIFT) + 1u);
}
// Group instances per chunk: 1, 2, 4 or 8 batches of 64.  A chunk is one wave and one row of the per-chunk count tables.
// A chunk wave's time is the sum of its memory and LDS round trips (measured: the kernels are latency bound at any
// occupancy they reach), so chunks are short while the launch stays within ~16 waves per SIMD.
#ifndef GSR_GROUP_CHUNKS_TARGET
#define GSR_GROUP_CHUNKS_TARGET 16384  // (A/B builds override it)
#endif
constexpr int GROUP_CHUNKS_TARGET = GSR_GROUP_CHUNKS_TARGET;
__host__ __device__ inline int group_chunk_items(int64_t G) {
  const int64_t batches = (G + 63) / 64;
  int per = 1;
  while (per < 8 && batches > (int64_t)GROUP_CHUNKS_TARGET * per) per *= 2;
  return 64 * per;
}
// `legacy`: the caller's choice of path (capi: image size, GSR_BIN_LEGACY); G is ignored on the legacy path.
__host__ __device__ inline Binning carve_binning(void* base, int64_t R, int64_t G, int W, int H, int legacy) {
  char* p = (char*)base;
  Binning b;
  const int64_t gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, T = gx * gy;
  b.legacy = legacy;
  size_t off = 0;
  // legacy fields
  b.nblocks = (uint32_t)((R + SORT_KPB - 1) / SORT_KPB);
  b.tile_bits = sort_key_bits(W, H) - 32;
  b.passes = (b.tile_bits + 7) / 8;
  int left = b.tile_bits;
  for (int i = 0; i < 4; ++i) {
    const int remaining_passes = b.passes - i;
    b.digit_bits[i] = remaining_passes > 0 ? (left + remaining_passes - 1) / remaining_passes : 0;
    left -= b.digit_bits[i];
  }
  b.final_buf = b.passes & 1;
  b.key_bytes = b.tile_bits <= 16 ? 2 : 4;  // a third less traffic per sorted pair, half for the histogram / range kernels
  // grouped fields
  b.sgx = (int)((gx + GROUP_EDGE - 1) >> GROUP_SHIFT);
  b.sgy = (int)((gy + GROUP_EDGE - 1) >> GROUP_SHIFT);
  b.groups = b.sgx * b.sgy;
  b.group_bits = b.groups <= 256 ? 8 : 11;
  b.G = legacy ? 0 : G;
  b.chunk = group_chunk_items(b.G);
  b.chunks = b.G / b.chunk + b.groups;
  b.sort_blocks = (uint32_t)((b.G + SORT_KPB - 1) / SORT_KPB);
  if (legacy) {
    // point_list = vals[final_buf]: the side is a function of the image size only
    b.vals[b.final_buf] = (uint32_t*)(p + off);      off += align_up(sizeof(uint32_t) * (size_t)R);
    b.vals[b.final_buf ^ 1] = (uint32_t*)(p + off);  off += align_up(sizeof(uint32_t) * (size_t)R);
    b.point_list = b.vals[b.final_buf];
    for (int i = 0; i < 2; ++i) { b.tkey[i] = (void*)(p + off); off += align_up((size_t)b.key_bytes * (size_t)R); }
    b.hist = (uint32_t*)(p + off);       off += align_up(sizeof(uint32_t) * SORT_MAX_BINS * (size_t)b.nblocks);
    b.bin_total = (uint32_t*)(p + off);  off += align_up(sizeof(uint32_t) * SORT_MAX_BINS);
    b.gkey[0] = b.gkey[1] = nullptr; b.gval[0] = b.gval[1] = nullptr;
    b.ghist = b.gbin_total = b.group_first = b.chunk_pre = b.tile_total = b.tile_start = nullptr;
    b.chunk_cnt = nullptr;
  } else {
    const size_t bins = (size_t)1 << b.group_bits, padded = (size_t)b.chunks * (size_t)b.chunk;
    // (everything that does not depend on G first: the blend kernels and the debug exports carve with G = 0)
    b.point_list = (uint32_t*)(p + off);  off += align_up(sizeof(uint32_t) * (size_t)R);
    b.tile_total = (uint32_t*)(p + off);  off += align_up(sizeof(uint32_t) * (size_t)T);
    b.tile_start = (uint32_t*)(p + off);  off += align_up(sizeof(uint32_t) * (size_t)(T + 1));
    b.gval[0] = (uint32_t*)(p + off);     off += align_up(sizeof(uint32_t) * (size_t)b.G);
    b.gval[1] = (uint32_t*)(p + off);     off += align_up(sizeof(uint32_t) * padded);
    b.gkey[0] = (uint32_t*)(p + off);     off += align_up(sizeof(uint32_t) * (size_t)b.G);
    b.gkey[1] = (uint32_t*)(p + off);     off += align_up(sizeof(uint32_t) * padded);
    b.ghist = (uint32_t*)(p + off);       off += align_up(sizeof(uint32_t) * bins * (size_t)b.sort_blocks);
    b.gbin_total = (uint32_t*)(p + off);  off += align_up(sizeof(uint32_t) * bins);
    b.group_first = (uint32_t*)(p + off); off += align_up(sizeof(uint32_t) * (bins + 1));
    b.chunk_cnt = (uint16_t*)(p + off);   off += align_up(sizeof(uint16_t) * GROUP_TILES * (size_t)b.chunks);
    b.chunk_pre = (uint32_t*)(p + off);   off += align_up(sizeof(uint32_t) * GROUP_TILES * (size_t)b.chunks);
    b.tkey[0] = b.tkey[1] = nullptr; b.vals[0] = b.vals[1] = nullptr; b.hist = b.bin_total = nullptr;
  }
  b.bytes = R > 0 ? off : 0;
  return b;
}

#if defined(__HIPCC__)
// ----------------------------------------------------------------------------------
// Device maths.  This translation unit set is compiled with -ffp-contract=off:
// every float op below is one IEEE binary32 operation unless fmaf is spelled out.
// ----------------------------------------------------------------------------------

// two binary32 values in one 64-bit register pair: the operand type of the packed instructions v_pk_{add,mul,fma}_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The exactly specified exponential (see oracle/gsr_oracle.cpp and DESIGN.md section 4):
// exp(x) = 2^n p(f), t = max(x log2 e, -125), n = rint(t), f = t - n, p = degree-6
// Horner polynomial in fmaf.  9 full-rate VALU ops + v_rndne + v_cvt + v_ldexp; no
// transcendental unit, bit-reproducible on any IEEE machine.
__device__ __forceinline__ float gsr_expf(float x) {
  float t = x * 0x1.715476p+0f;
  t = fmaxf(t, -125.0f);
  const float n = __builtin_rintf(t);
  const float f = t - n;
  float p = 0x1.44138ap-13f;
  p = __builtin_fmaf(p, f, 0x1.5f0890p-10f);
  p = __builtin_fmaf(p, f, 0x1.3b2a54p-7f);
  p = __builtin_fmaf(p, f, 0x1.c6af6cp-5f);
  p = __builtin_fmaf(p, f, 0x1.ebfbe0p-3f);
  p = __builtin_fmaf(p, f, 0x1.62e430p-1f);
  p = __builtin_fmaf(p, f, 1.0f);
  return __builtin_amdgcn_ldexpf(p, (int)n);
}

// gsr_expf without the lower clamp, for uses where any result below ~1e-37 is equivalent to 0 because it
// is only compared against the 1/255 threshold (v_ldexp_f32 handles every exponent; for t < -126 the
// result is subnormal or zero either way).  Bit-identical to gsr_expf for x >= -86.6.
__device__ __forceinline__ float gsr_expf_noclamp(float x) {
  const float t = x * 0x1.715476p+0f;
  const float n = __builtin_rintf(t);
  const float f = t - n;
  float p = 0x1.44138ap-13f;
  p = __builtin_fmaf(p, f, 0x1.5f0890p-10f);
  p = __builtin_fmaf(p, f, 0x1.3b2a54p-7f);
  p = __builtin_fmaf(p, f, 0x1.c6af6cp-5f);
  p = __builtin_fmaf(p, f, 0x1.ebfbe0p-3f);
  p = __builtin_fmaf(p, f, 0x1.62e430p-1f);
  p = __builtin_fmaf(p, f, 1.0f);
  return __builtin_amdgcn_ldexpf(p, (int)n);
}

// blend_power with the conic pre-scaled at staging time: (hA, B_, hC) = (-0.5 A, -B, -0.5 C).  Scaling by
// -0.5 / -1 commutes with every rounding, so the result is bit-identical to blend_power(A, B, C, dx, dy).
__device__ __forceinline__ float blend_power_prescaled(float hA, float nB, float hC, float dx, float dy) {
  const float a = (hA * dx) * dx;
  const float s = __builtin_fmaf(hC * dy, dy, a);
  return __builtin_fmaf(nB * dx, dy, s);
}

// Gaussian footprint exponent, forward.cu:335-338 with the fma placement of the spec.
__device__ __forceinline__ float blend_power(float cx, float cy, float cz, float dx, float dy) {
  const float a = (cx * dx) * dx;
  const float s = __builtin_fmaf(cz * dy, dy, a);
  const float h = -0.5f * s;
  return __builtin_fmaf(-(cy * dx), dy, h);
}

// float -> int with v_cvt_i32_f32 semantics (saturating, NaN -> 0).
__device__ __forceinline__ int f2i(float v) { return (int)v; }

// --- wave64 primitives -------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {
  // v + (v moved by DPP control CTRL); lanes without a source add 0.
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, false);
  return v + __builtin_bit_cast(float, moved);
}
// Sum over the 64 lanes of a wave; the total is valid in lane 63 only.
// 6 v_add_f32_dpp: row_shr 1,2,4,8 then row_bcast15 (rows 1,3) and row_bcast31 (rows 2,3).
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0x111>(v);
  v = dpp_add<0x112>(v);
  v = dpp_add<0x114>(v);
  v = dpp_add<0x118>(v);
  v = dpp_add<0x142, 0xa>(v);
  v = dpp_add<0x143, 0xc>(v);
  return v;
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  const int l = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d, 64);
    if (l >= d) v += o;
  }
  return v;
}

// Exclusive scan of one value per thread across a block of NT threads (NT multiple of 64, <= 1024).
// Returns the exclusive prefix; *block_total receives the block sum (valid in all threads).
// The same with DPP row shifts / row broadcasts: 6 VALU instructions, no LDS round trip.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);  // lanes without a source add 0
}
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
  v = dpp_add_u32<0x111>(v);       // row_shr:1
  v = dpp_add_u32<0x112>(v);       // row_shr:2
  v = dpp_add_u32<0x114>(v);       // row_shr:4
  v = dpp_add_u32<0x118>(v);       // row_shr:8  -> inclusive scan inside each row of 16
  v = dpp_add_u32<0x142, 0xa>(v);  // row_bcast:15 into rows 1, 3
  v = dpp_add_u32<0x143, 0xc>(v);  // row_bcast:31 into rows 2, 3
  return v;
}

template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* block_total, uint32_t* smem /* NT/64 + 1 */) {
  constexpr int NW = NT / 64;
  const int w = (int)(threadIdx.x >> 6), l = lane_id();
  const uint32_t incl = wave_incl_scan_dpp(v);
  __syncthreads();  // protect smem reuse across calls
  if (l == 63) smem[w] = incl;
  __syncthreads();
  if (w == 0) {
    uint32_t t = (l < NW) ? smem[l] : 0u;
    const uint32_t ti = wave_incl_scan_dpp(t);
    if (l < NW) smem[l] = ti - t;
    if (l == NW - 1) smem[NW] = ti;
  }
  __syncthreads();
  *block_total = smem[NW];
  return smem[w] + incl - v;
}
#endif  // __HIPCC__

}  // namespace gsr
