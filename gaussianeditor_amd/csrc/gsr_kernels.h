// gsr_kernels.h -- kernel argument blocks and host-side launchers shared by the translation units.
#pragma once
#include "gsr_common.h"

namespace gsr {

// K1 arguments (preprocess_kernel, gsr_preprocess.hip)
struct PreArgs {
  int P, D, M;
  const float* means3D;
  const float* scales;
  float scale_modifier;
  const float* rotations;
  const float* opacities;
  const float* shs;
  const float* cov3D_precomp;
  const float* colors_precomp;
  const float* viewmatrix;
  const float* projmatrix;
  const float* campos;
  int W, H;
  float tan_fovx, tan_fovy, focal_x, focal_y;
  int gx, gy;
  int skip_color;
  int forward_only;  // GSR_FLAG_FORWARD_ONLY: Geom::dcol (read by the backward only) is not written
  int tile_bounds;  // 0: the reference's square of side 2 ceil(3 sigma_max); 1: its intersection with the alpha >= 1/255 box (gsr_set_option)
  int32_t* radii;
  Geom g;
};

// K8+K9 arguments (preprocess_backward_kernel, gsr_preprocess.hip)
struct PreBwdArgs {
  int P, D, M;
  const float* means3D;
  const int32_t* radii;
  const float* shs;
  const float* scales;
  const float* rotations;
  float scale_modifier;
  const float* cov3D_precomp;  // (P,6) or null: then recomputed from scales / rotations as K1 did (never stored)
  const uint8_t* clamped;
  const float* dcol[3];        // Geom::dcol: d(RGB)/d(dir) left by K1, 3 floats per Gaussian each (read instead of the SH record)
  const float* viewmatrix;
  const float* projmatrix;
  const float* campos;
  float h_x, h_y, tan_fovx, tan_fovy;
  const float* acc;         // (P, ACC_ROW): the blend backward's accumulator rows (dL_dmean2D, dL_dopacity, dL_dconic, dL_dcolor)
  float* acc_clean;         // == acc (GSR_FLAG_ACC_SELF_CLEAN: rows that are not zero are put back to zero) | null
  float* dL_dmean2D;        // (P,3) out: columns ACC_MEAN2D .. + 1 of the rows, z = 0
  float* dL_dopacity;       // (P)   out: column ACC_OPACITY
  float* dL_dcolor;         // (P,3) out | null: columns ACC_COLOR .. + 2 (the gradient of colors_precomp)
  float* dL_dmeans3D;       // (P,3)
  float* dL_dcov3D;         // (P,6)
  float* dL_dsh;            // (P,M,3) | null
  float* dL_drgb;           // (P,3)   | null: the clamp-masked colour gradient, for gsr_sh_grad_compose
  float* dL_dscale;         // (P,3)   | null
  float* dL_drot;           // (P,4)   | null
  uint8_t* row_state;       // (P) | null: gsr_preprocess_backward_rows -- rows that still hold this kernel's zeros are not rewritten
};

// K6 / K7 / K12 arguments (gsr_blend.hip)
struct BlendArgs {
  int W, H, gx, gy;
  const uint32_t* work_order;  // tile ids, longest first, empty tiles last
  const uint32_t* work_meta;   // [0] = number of non-empty tiles
  uint32_t* work_est;          // forward: out, (T,4) evaluated entries per quadrant; null = not recorded
  uint32_t* work_maxc;         // forward: out, (T,4) deepest contributing list position + 1 per quadrant (with work_est)
  uint32_t* bwd_order;         // backward launch: scratch for its own work list (gsr_blend.hip: backward_worklist_kernel)
  uint32_t* bwd_meta;
  uint32_t* queue;             // 8 per-XCD cursors + retire counters of this kind, QUEUE_STRIDE words apart; zero on
                               // entry, and left zero again by the launch's last workgroup
  const uint2* ranges;
  const uint32_t* point_list;
  const float4* rec0;
  const float4* rec1;
  const float4* rec2;
  const float* colors3;        // auxiliary forward render: (P,3) colours blended instead of rec2; null otherwise
  const float* bg;
  float* final_T;
  uint32_t* n_contrib;
  // forward outputs
  float* out_color;
  float* out_depth;
  // backward
  const float* dL_dpix;
  float* acc;      // (P, ACC_ROW): one 64-byte row of accumulators per Gaussian (ACC_* columns, gsr_common.h / include/gsr.h)
  uint8_t* touched;  // (P) | null: 1 for every Gaussian whose row this launch adds to (cleared by the launch itself)
  // tracing
  int C;
  const float* image_weights;
  float* weights;
  int32_t* cnt;
  int P;           // number of Gaussians (rows of the backward's accumulator table)
  int clear_grads; // GSR_FLAG_CLEAR_GRADS: the backward clears its accumulator rows itself (launch_blend_backward)
  int fast_exp;    // GSR_FLAG_FAST_EXP: hardware 2^x instead of the specified polynomial (gsr_blend.hip: blend_exp)
  int shared_simds;  // GSR_FLAG_SHARED_SIMDS: 2 persistent waves per SIMD instead of 4 (another stream's kernels run alongside)
  int self_reset;  // the last workgroup to retire clears the queue cursors (default)
  int allow_split; // forward: quadrants may be cut into 2 or 4 items when the image has few tiles (run_work_queue)
  int for_backward; // forward: the render's state will be read by a backward (not GSR_FLAG_FORWARD_ONLY, not an auxiliary render)
  int units;       // placement units (SIMDs or CUs) for the assigned first items, 0 = none; gsr_blend.hip: first_item_of_block
  // forward checkpoints / backward list segments (Image::ck_*); ck_table == null: none (auxiliary render, tracing)
  uint32_t* ck_table;
  uint32_t* ck_work;
  uint32_t* tile_maxc;
  float4* ck_pool;
  int ck_chunks;   // checkpoint stride in 64-entry chunks
  int ck_slots;    // checkpoint slots in use per tile (<= CK_MAX: the stride of the pool's layout)
  CkTable ck_pos;  // list position of every checkpoint, in 64-entry chunks (gsr_common.h; pos[1] == ck_chunks)
  // debug: per-workgroup timing records (4 x u64 each), or null
  uint64_t* profile_items;  // debug (backward): 4 x u64 per (tile, half) after the workgroup records, or null
  uint64_t* profile;
};

hipError_t launch_preprocess(hipStream_t s, const PreArgs& a);
hipError_t launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* view, uint8_t* present);
hipError_t launch_preprocess_backward(hipStream_t s, const PreBwdArgs& a);
hipError_t launch_sh_grad_compose(hipStream_t s, int P, int D, int M, int N, const float* means3D, const float* campos,
                                  const float* dL_drgb, float* dL_dsh);
// "touched rows" exchange (gsr_preprocess.hip): mask of rows with a non-zero entry in any of up to 8 row-major tensors;
// a view's message = header (camera centre, count, compaction block offsets) + packed rows; all views' messages added
// per Gaussian in view order into the dense gradients (dense: means3D, scales, rotations, means2D, opacities, sh | null)
constexpr int VIEW_MSG_ROWS = 1024;  // rows per block of the message's offset table == the compaction's block (gsr_compact.hip)
hipError_t launch_touched_rows(hipStream_t s, int64_t P, int nt, const float* const* data, const int* row_len, uint8_t* mask);
int64_t view_message_words_host(int64_t P, int64_t cap);
hipError_t launch_view_message_header(hipStream_t s, int64_t P, const float* campos, const uint32_t* block_off,
                                      const uint64_t* total, float* msg);
hipError_t launch_view_messages_accumulate(hipStream_t s, int64_t P, int D, int M, int n_views, const float* messages,
                                           int64_t stride_words, int64_t cap, const float* means3D, float* const dense[6],
                                           uint8_t* row_valid);
const uint32_t* compact_block_off_ptr(void* workspace, int64_t P);
hipError_t launch_export_cov3d(hipStream_t s, int P, const float* scales, float scale_modifier, const float* rotations,
                               float* cov3D);
hipError_t launch_export_geom(hipStream_t s, int P, const Geom& g, float* means2D, float* depths, float* rgb,
                              float* conic_opacity, uint32_t* tiles_touched, uint8_t* clamped);
hipError_t launch_depth_passes(hipStream_t s, int P, const Geom& g, int p0, int p1, uint32_t* publish_dst = nullptr,
                               uint32_t publish_seq = 0);
// sgx: groups per row when the grouped binning path follows (the depth-ordered rectangles are then group rectangles), 0 for
// the legacy pair sort
hipError_t launch_depth_finish(hipStream_t s, int P, const Geom& g, int passes, int gx, int sgx);
// Grouped path: the same passes carrying the packed tile rectangle; `last`: pass p1 - 1 is the final one and leaves the
// depth order, the rectangles in that order and every Gaussian's first slot of the emission (no launch_depth_finish).
hipError_t launch_depth_passes_grouped(hipStream_t s, int P, const Geom& g, int p0, int p1, bool last,
                                       uint32_t* publish_dst = nullptr, uint32_t publish_seq = 0);
hipError_t launch_export_keys(hipStream_t s, int64_t R, int W, int H, const Binning& b, const Geom& g, uint64_t* keys);
hipError_t launch_binning(hipStream_t s, int P, int64_t R, int W, int H, const Geom& g, const Binning& b, const Image& im);
size_t knn_workspace_bytes(int P);
hipError_t launch_knn(hipStream_t s, int P, const float* points, void* workspace, float* out);
hipError_t launch_near_points(hipStream_t s, int n_ref, const float* ref, int n_query, const float* query, float thresh,
                              void* workspace, uint8_t* near, float* nn_dist);
size_t compact_workspace_bytes(int64_t P);
hipError_t launch_compact_plan(hipStream_t s, int64_t P, const uint8_t* keep, void* workspace);
const uint64_t* compact_total_ptr(void* workspace, int64_t P);
// limit: rows the destinations hold (survivors beyond it are dropped); < 0 = unlimited
hipError_t launch_compact_apply(hipStream_t s, int64_t P, const uint8_t* keep, void* workspace, int nt,
                                const gsr_compact_tensor* tensors, int64_t limit = -1);
hipError_t launch_append_rows(hipStream_t s, int64_t P, int64_t n, int nt, const gsr_append_tensor* tensors);
hipError_t launch_adam_step(hipStream_t s, int nt, const gsr_adam_tensor* tensors, long long step, double beta1,
                            double beta2, double eps, const uint8_t* row_mask, const float* row_weight,
                            const uint8_t* grad_valid = nullptr);
hipError_t launch_blend_forward(hipStream_t s, BlendArgs a);
unsigned blend_grid_size(bool backward, hipStream_t s, bool shared_simds = false);  // persistent waves of a blend launch on the device of stream s
hipError_t launch_blend_backward(hipStream_t s, BlendArgs a);
hipError_t launch_trace_weights(hipStream_t s, BlendArgs a);

}  // namespace gsr
