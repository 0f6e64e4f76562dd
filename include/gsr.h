/* gsr.h -- C ABI of libgsr_hip.so, the MI355X (gfx950) Gaussian-splatting rasterizer.
 *
 * This is the drop-in boundary for the reference's native extension
 * `diff_gaussian_rasterization._C` (reference paths relative to
 * /root/reference/gaussiansplatting/submodules/diff-gaussian-rasterization):
 *
 *   _C.rasterize_gaussians           ext.cpp:16, rasterize_points.cu:35-95
 *       -> gsr_preprocess + gsr_bin + gsr_blend_forward
 *   _C.rasterize_gaussians_backward  ext.cpp:17, rasterize_points.cu:97-157
 *       -> gsr_backward
 *   _C.mark_visible                  ext.cpp:18, rasterize_points.cu:159-175
 *       -> gsr_mark_visible
 *   _C.apply_weights                 ext.cpp:19, rasterize_points.cu:177-234
 *       -> gsr_preprocess + gsr_bin + gsr_trace_weights
 *
 * Rules of the boundary
 *   - plain C: pointers, sizes, scalars; no torch / pybind types.
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`.
 *   - the library never allocates device memory: the caller (torch) owns every
 *     output and scratch buffer; sizes come from gsr_scratch_sizes().  Scratch
 *     buffers are opaque bytes that the caller keeps alive between forward and
 *     backward (the reference's geomBuffer / binningBuffer / imgBuffer,
 *     rasterize_points.cu:64-69, diff_gaussian_rasterization/__init__.py:122-133).
 *   - scratch buffers (geom / binning / image) must be 256-byte aligned (torch's allocations are): their sections are
 *     read with 16-byte vector loads.  A misaligned one is rejected with GSR_ERR_BAD_ARGUMENT.
 *   - `stream` is a hipStream_t (passed as void*); all work is enqueued on it.
 *     The library is re-entrant and keeps no BEHAVIOUR state (the switches that can change a result travel with every
 *     call as `flags`); the only thing it owns is a small pool of 8 KiB pinned host buffers, one checked out per host
 *     thread and device, through which gsr_preprocess receives its counts (the device writes them, the host polls).
 *   - tuning / experiment knobs are environment variables read ONCE per process, none of which changes any result
 *     (launch geometry of the persistent blend kernels, work-list ordering, ablation switches, and GSR_BIN_LEGACY=1,
 *     which sends every image down the tile-pair sort that otherwise only images beyond 131 072 tiles take); they are
 *     listed in DESIGN.md section 3.3 and are not part of this ABI.
 *   - an absent optional input is a NULL pointer (the reference uses empty
 *     tensors for the same purpose, diff_gaussian_rasterization/__init__.py:285-295).
 *   - return value: GSR_OK (0) or a negative gsr_status; never exit()/abort().
 *   - all floating point is IEEE binary32; images are CHW; matrices are the
 *     16-float row-major tensors holding the TRANSPOSED 4x4 matrices exactly as
 *     the reference passes them.
 */
#ifndef GSR_H_INCLUDED
#define GSR_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* Everything declared here has default visibility; the library itself is compiled with -fvisibility=hidden, so these
 * entry points are ALL it exports (tests/test_cpu_abi.py checks both directions). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* 4 (round 5): adds gsr_near_workspace_size / gsr_near_points (round 4 had left the number at 3), the scratch layouts
 * changed again (sizes come from gsr_scratch_sizes: rebuild nothing, re-query), images of more than GSR_MAX_TILES tiles
 * are refused by the backward entry points instead of being walked wrongly. */
/* 5 (round 6): adds gsr_arrays_equal; the blend backward accumulates into ONE table of 64-byte rows (`acc`, GSR_ACC_*)
 * instead of four arrays, and gsr_preprocess_backward* copies dL_dmeans2D / dL_dopacity (/ dL_dcolors) out of it: the
 * signatures of gsr_backward, gsr_blend_backward, gsr_preprocess_backward{,_rgb,_rows}, gsr_view_message_plan_blend and
 * gsr_debug_blend_backward_profile changed. */
/* 6 (round 6): adds gsr_preprocess_begin / gsr_preprocess_end (nothing else changed). */
#define GSR_ABI_VERSION 6
/* The blend backward's accumulator table: GSR_ACC_ROW floats (one 64-byte line) per Gaussian, 64-byte aligned.  Columns:
 *   [GSR_ACC_MEAN2D] .x [+1] .y of dL_dmean2D      (backward.cu:545-546)
 *   [GSR_ACC_OPACITY] dL_dopacity                  (backward.cu:554)
 *   [GSR_ACC_CONIC] .x [+1] .y [+3] .w of dL_dconic (backward.cu:549-551; the reference's float4 layout, .z unused)
 *   [GSR_ACC_COLOR .. +2] dL_dcolor                (backward.cu:523)
 * every other column stays zero.  Why one row: float atomics execute at the memory side on this chip and cost per REQUEST;
 * the lanes of a wave instruction that fall into one 64-byte line travel as one request, so the (up to nine) adds of a
 * (tile, Gaussian) pair leave as one instead of nine (tools/microbench/atomic_merge.hip: 8.8x). */
#define GSR_ACC_ROW 16
#define GSR_ACC_MEAN2D 0
#define GSR_ACC_OPACITY 3
#define GSR_ACC_CONIC 4
#define GSR_ACC_COLOR 8
/* Largest image the blend BACKWARD accepts, in 16 x 16 tiles (its work items carry the tile id in 20 bits). */
#define GSR_MAX_TILES (1 << 20)

typedef enum gsr_status {
  GSR_OK = 0,
  GSR_ERR_BAD_ARGUMENT = -1,   /* NULL where a pointer is required, negative sizes, ... */
  GSR_ERR_BAD_CHANNELS = -2,   /* apply_weights with C outside {1,2,3} (reference: exit(-1), apply_weights.cu:377-380) */
  GSR_ERR_TOO_MANY = -3,       /* num_rendered does not fit the 31-bit index space */
  GSR_ERR_HIP = -4             /* a HIP runtime call failed; see gsr_last_hip_error() */
} gsr_status;

/* Version of this ABI (== GSR_ABI_VERSION of the header the library was built from). */
int gsr_abi_version(void);
/* Static description of a status code. */
const char* gsr_status_string(int status);
/* hipError_t of the most recent failing HIP call on the calling thread (0 if none). */
int gsr_last_hip_error(void);

/* Byte sizes of the three opaque scratch buffers for P Gaussians, R rendered
 * instances, G group instances (both returned by gsr_preprocess) and a W x H image.  Pass R = G = 0 before they are
 * known (sizes[1] is then 0).
 * sizes[0] = geometry (per Gaussian), sizes[1] = binning (per instance),
 * sizes[2] = image (per pixel / tile).  sizes[2] with the real R may be smaller than with R = 0 (the forward's checkpoint
 * pool -- 64 KB per tile, 128 MB at most -- is only needed for views with long lists): a caller that allocates the image
 * scratch after gsr_preprocess saves it, one that sized it with R = 0 beforehand is always large enough.
 * REUSE: the entry points cannot see how large a scratch buffer is.  An image scratch that is kept and reused for later
 * views (or for a view rendered under the GSR_CK_CHUNKS knob) MUST have been sized with R = 0: one sized with a short-list
 * R lacks the pool, and a later view with long lists would have its forward write checkpoints behind the buffer's end.
 * (The Python binding allocates the image scratch per view, with that view's R.)
 * Replaces required<GeometryState/BinningState/ImageState>() (rasterizer_impl.h:63-72). */
int gsr_scratch_sizes(int P, int64_t R, int64_t G, int W, int H, size_t sizes[3]);

/* Per-call behaviour flags (ABI 2; ABI 1 had a process-wide gsr_set_option instead).  The library keeps no option
 * state: every entry point below that takes `flags` receives them with the call, and the calls that belong to one view
 * (gsr_preprocess ... gsr_backward / gsr_trace_weights) must be given the same value.  0 reproduces the reference's
 * internal state bit for bit.
 *   GSR_FLAG_TILE_BOUNDS_ALPHA  (read by gsr_preprocess) a Gaussian is binned only into the tiles its alpha >= 1/255
 *                        level set can reach (bounding box of that ellipse with conservative margins, intersected with
 *                        the reference's square of side 2 ceil(3 sigma_max), auxiliary.h:46-56, forward.cu:226-230).
 *                        Every dropped (tile, Gaussian) instance would have been skipped at each pixel by
 *                        forward.cu:340-344, so images, depths, radii, traced weights and gradients are unchanged --
 *                        but num_rendered, the instance lists in the binning scratch and n_contrib differ from the
 *                        reference's.
 *   GSR_FLAG_CLEAR_GRADS (read by gsr_blend_backward / gsr_backward) the accumulator table `acc`
 *       need not be zero on entry: the call clears it before it accumulates (inside the launch that builds the backward's
 *       work list, while that one workgroup runs: 14 us instead of 8 + 9 at 10^6 Gaussians, and one launch less);
 *   GSR_FLAG_FORWARD_ONLY (read by gsr_preprocess) no gsr_preprocess_backward / gsr_backward will be called on the geometry
 *       state this call leaves: the 48 bytes per Gaussian only the backward reads (the colour's derivative by the view
 *       direction) are neither computed nor written.  A backward on such a state would read uninitialised memory; it is
 *       refused with GSR_ERR_BAD_ARGUMENT where the library can tell: when the flag is passed on to gsr_blend_backward /
 *       gsr_backward, and when `geom` is the buffer the most recent gsr_preprocess on it left forward-only.  That second
 *       check is BEST EFFORT by construction: a host-side table of 1 024 hashed slots keyed by the `geom` pointer, no device
 *       read -- a note can be evicted by a gsr_preprocess on another buffer that hashes to its slot, and a state copied to
 *       another buffer is not recognised; do not rely on it to catch the misuse.  The Python
 *       binding sets the flag exactly for renders none of whose inputs requires a gradient;
 *   GSR_FLAG_FAST_EXP    (read by the blend / trace entry points) exp(power) is evaluated with the hardware's
 *                        v_exp_f32 (2^x, 1 ulp) on power * log2(e) instead of the exactly specified polynomial
 *                        gsr_expf.  Colours / depths / gradients stay within the 1e-5 parity
 *                        bar, but a pixel whose alpha or transmittance sits within a rounding of a threshold
 *                        (1/255, 1e-4) may take the other branch than the CPU oracle, so n_contrib / final_T are no
 *                        longer bit-identical to it (they are not bit-identical to the reference's libm build either).
 *   GSR_FLAG_ACC_SELF_CLEAN (ABI 5; read by gsr_backward / gsr_preprocess_backward / _rgb) the accumulator table `acc` is one the caller KEEPS between
 *       backwards: it is all zero on entry (the caller's promise -- hipMemset it once) and all zero again when the call has
 *       run: K8+K9, which reads every row anyway, writes zeros over the rows K7 touched (one Gaussian in ten on the
 *       benchmark view: 6 MB instead of a 64 MB clear in front of every backward).  Not combined with GSR_FLAG_CLEAR_GRADS
 *       (which it makes unnecessary).  With the two halves called separately: gsr_blend_backward WITHOUT GSR_FLAG_CLEAR_GRADS
 *       on the zero table, then gsr_preprocess_backward (or _rgb) with this flag.  Nothing else may read `acc` while K8+K9
 *       runs (the exchange plans its messages from K7's `touched` mask, not from the table);
 *   GSR_FLAG_SHARED_SIMDS (ABI 4; read by the blend / trace entry points) the caller overlaps this view's kernels with
 *                        another view's on a second stream: the persistent blend kernels are launched with 2 waves per
 *                        SIMD instead of 4, which leaves wave slots and registers for the other stream's kernels (a rank
 *                        that renders several views of a batch: -15 % per view, profiles/r04_c_pipelining.md).  Never
 *                        changes a result.  The environment knob GSR_BLEND_WAVES_PER_SIMD, where set, takes precedence.
 * Unknown bits are rejected with GSR_ERR_BAD_ARGUMENT. */
#define GSR_FLAG_TILE_BOUNDS_ALPHA 1u
#define GSR_FLAG_FAST_EXP 2u
#define GSR_FLAG_CLEAR_GRADS 4u
#define GSR_FLAG_FORWARD_ONLY 8u
#define GSR_FLAG_SHARED_SIMDS 16u
#define GSR_FLAG_ACC_SELF_CLEAN 32u
#define GSR_FLAG_ALL 63u

/* Number of sort-key bits, 32 + getHigherMsb(tiles) (rasterizer_impl.cu:36-49, 253). */
int gsr_sort_key_bits(int W, int H);

/* K1 + K2: per-Gaussian preprocessing (SH -> RGB, 3D -> 2D covariance, conic,
 * radius, tile rectangle), the depth ordering of the Gaussians (first half of K4, see gsr_bin) and the
 * prefix sums the emission in that order needs; then the ONE
 * blocking device->host readback of the path: counts_host[0] = num_rendered, the total number
 * of (Gaussian, tile) instances, counts_host[1] = the number of (Gaussian, 8x8-tile group) instances gsr_bin works on
 * (0 for images that take the tile-pair sort).  Both size the binning scratch.  Reference: FORWARD::preprocess +
 * InclusiveSum + cudaMemcpy, rasterizer_impl.cu:217-239 (and :381-403 for apply_weights).
 * The host normally waits ~80 us for the counts, spinning on a pinned word the device writes; if nothing arrives within
 * 5 ms (earlier work queued on the stream, or a failed launch) the call blocks in hipStreamSynchronize instead and
 * reports what that returns.  `prefiltered` is accepted for signature compatibility and has no effect (the reference
 * uses it only to trap the device when a point it was promised to be visible is culled, auxiliary.h:156-160).
 *
 *   P, D, M          #Gaussians, active SH degree (0..3), SH coefficients per Gaussian (0 if shs == NULL)
 *   means3D (P,3); scales (P,3)|NULL; rotations (P,4)|NULL; opacities (P); shs (P,M,3)|NULL;
 *   cov3D_precomp (P,6)|NULL; colors_precomp (P,3)|NULL  (exactly one of shs/colors, one of scales+rot/cov3D,
 *   unless skip_color != 0, in which case neither colour source is read: the apply_weights path)
 *   viewmatrix, projmatrix (16 floats each); campos (3)
 *   radii (P) int32 out; geom: scratch, sizes[0] bytes. */
int gsr_preprocess(void* stream, int P, int D, int M, const float* means3D, const float* scales,
                   float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                   const float* cov3D_precomp, const float* colors_precomp, const float* viewmatrix,
                   const float* projmatrix, const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                   int prefiltered, int skip_color, unsigned flags, int32_t* radii, void* geom,
                   int64_t counts_host[2]);

/* gsr_preprocess in two halves (ABI 6), for a caller that keeps several views in flight: _begin checks the arguments,
 * enqueues K1 and the depth-sort passes that do not depend on the counts, and returns AT ONCE with a ticket; _end (same
 * stream, same P / W / H / geom; any host thread) waits for that view's counts exactly as gsr_preprocess does, enqueues the
 * rest of the depth ordering and fills counts_host.  gsr_preprocess(...) == _begin(...) followed by _end(...): same
 * kernels, same results.  Between the two calls the launch thread may enqueue other views' work -- a rank that renders a
 * batch of views issues view v + 1's _begin before view v's backward and never waits for a readback at all
 * (gaussianeditor_amd/multiview.py; the reference's loop over batch["camera"], threestudio/systems/GassuianEditor.py:
 * 165-207, blocks in cudaMemcpy once per view).  A ticket is spent by _end whatever _end returns; one that is never
 * passed to _end leaks one 64-byte pinned slot.  P == 0 is refused (GSR_ERR_BAD_ARGUMENT): nothing to wait for. */
int gsr_preprocess_begin(void* stream, int P, int D, int M, const float* means3D, const float* scales,
                         float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                         const float* cov3D_precomp, const float* colors_precomp, const float* viewmatrix,
                         const float* projmatrix, const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                         int prefiltered, int skip_color, unsigned flags, int32_t* radii, void* geom, void** ticket);
int gsr_preprocess_end(void* stream, int P, int W, int H, void* geom, void* ticket, int64_t counts_host[2]);

/* Are n <= 8 pairs of device arrays identical bit for bit?  a[i] / b[i]: device pointers (4-byte aligned; a pair with
 * a[i] == b[i] or bytes[i] == 0 is equal without being read), bytes[i]: their size, a multiple of 4.  One compare launch
 * over all pairs (16-byte loads where both sides are 16-byte aligned) plus a one-thread launch that publishes the answer
 * into the calling thread's pinned slot; like gsr_preprocess the call BLOCKS until the answer is there (it polls; ~10 us
 * + 64 MB per 12 us).  *equal_host = 1 (all pairs equal) or 0.
 * No reference counterpart: the Python layer uses it to PROVE that a second render() of a view -- the reference renders
 * every training view and GUI frame twice, with override_color the second time (threestudio/systems/GassuianEditor.py:
 * 166-191, webui.py:693-713) -- got the opacities / scales / rotations of the first (fresh activation tensors each time),
 * before it reuses the first render's preprocessing, sort and tile lists ("view reuse"). */
int gsr_arrays_equal(void* stream, int n, const void* const* a, const void* const* b, const size_t* bytes, int* equal_host);

/* K3 + K4 + K5: the per-tile instance lists (the reference's point_list) and their [begin,end) ranges, in exactly the
 * order of the reference's stable sort on the low gsr_sort_key_bits() bits of (tile << 32 | depth bits): the Gaussians
 * were depth-ordered by gsr_preprocess; here one (group, Gaussian) pair per 8x8-tile group a Gaussian reaches is emitted
 * in that order, stably sorted by group in one radix pass, and expanded group by group into the tiles' lists
 * (gsr_binning.hip).  R and G are the two counts gsr_preprocess returned.
 * Reference: duplicateWithKeys + cub::DeviceRadixSort::SortPairs + identifyTileRanges,
 * rasterizer_impl.cu:248-271.  `binning` holds sizes[1] bytes for this (R, G). */
int gsr_bin(void* stream, int P, int64_t R, int64_t G, int W, int H, const void* geom, void* binning, void* image);

/* K6: per-tile front-to-back alpha compositing.  Reference: FORWARD::render,
 * forward.cu:261-409.  out_color (3,H,W), out_depth (1,H,W) are fully written
 * (background where nothing is blended); final_T / n_contrib go to `image`. */
int gsr_blend_forward(void* stream, int P, int64_t R, int W, int H, const float* bg, const void* geom,
                      const void* binning, void* image, float* out_color, float* out_depth, unsigned flags);

/* Auxiliary forward render of the SAME view with other per-Gaussian colours: blends
 * `colors` (P,3) through the geometry / binning state an earlier gsr_preprocess + gsr_bin left in the scratch buffers,
 * i.e. K6 only.  GaussianEditor renders every training view and every GUI frame twice, the second time with
 * `override_color` = the semantic mask (threestudio/systems/GassuianEditor.py:166-191, webui.py:693-713); K1-K5 of that
 * second render are identical to the first one's.  The image equals what gsr_preprocess(colors_precomp = colors) ...
 * gsr_blend_forward would produce.  final_T / n_contrib in `image` are NOT touched, so the backward of the main render
 * is unaffected; no backward exists for the auxiliary image.  out_depth may be NULL. */
int gsr_blend_forward_aux(void* stream, int P, int64_t R, int W, int H, const float* bg, const void* geom,
                          const void* binning, void* image, const float* colors, float* out_color, float* out_depth,
                          unsigned flags);

/* K7 + K8 + K9: the whole backward.  Reference: Rasterizer::backward,
 * rasterizer_impl.cu:289-341 (BACKWARD::render then BACKWARD::preprocess).
 *   dL_dpix (3,H,W) in.
 *   acc (P, GSR_ACC_ROW) workspace: the accumulator table K7 adds into with atomics and K8+K9 reads.  MUST be zero on entry
 *       (or is cleared by the call: GSR_FLAG_CLEAR_GRADS), 64-byte aligned, and MUST live in ordinary (coarse-grained)
 *       device memory -- hipMalloc, torch's caching allocator --, not in fine-grained / host-coherent allocations
 *       (hipMallocManaged, hipHostMalloc, hipExtMallocWithFlags(... hipDeviceMallocFinegrained)): the library is built with
 *       -munsafe-fp-atomics, i.e. its float adds are the hardware's global_atomic_add_f32, which is only defined on
 *       coarse-grained memory (on fine-grained memory the adds are silently lost).  The same holds for `weights` / `cnt` of
 *       gsr_trace_weights.
 *   Fully written by the library (no need to zero): dL_dmeans2D (P,3) [z = 0], dL_dopacity (P), dL_dcolors (P,3) [may be
 *       NULL: only a caller with precomputed colours needs it], dL_dmeans3D (P,3), dL_dcov3D (P,6) [may be NULL when
 *       cov3D_precomp is NULL: it is the gradient of that input; 24 B per Gaussian nobody reads otherwise],
 *       dL_dsh (P,M,3) [NULL if shs == NULL], dL_dscales (P,3) and dL_drots (P,4) [NULL if scales == NULL].
 *   (The reference zero-fills all nine of its outputs, rasterize_points.cu:120-128, and accumulates into four of them.) */
int gsr_backward(void* stream, int P, int D, int M, int64_t R, int W, int H, const float* bg, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                 const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int32_t* radii,
                 const void* geom, const void* binning, const void* image, const float* dL_dpix, float* acc,
                 float* dL_dmeans2D, float* dL_dopacity, float* dL_dcolors, float* dL_dmeans3D,
                 float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drots, unsigned flags);

/* The two halves of gsr_backward as separate entry points (same argument meaning), so a caller
 * can time or overlap them: K7 = BACKWARD::render (backward.cu:399-557), K8+K9 = BACKWARD::preprocess
 * (backward.cu:559-622).  gsr_backward == gsr_blend_backward followed by gsr_preprocess_backward.
 * After gsr_blend_backward alone `acc` holds the sums in the GSR_ACC_* columns; gsr_preprocess_backward reads them and
 * writes dL_dmeans2D / dL_dopacity (/ dL_dcolors) next to its own outputs. */
/* `touched` (P bytes rounded up to a multiple of 16, 16-byte aligned, device; may be NULL): cleared by the call, then 1 for
 * every Gaussian whose accumulator row K7 adds to -- the row mask of the multi-GPU exchange (gsr_view_message_plan_blend)
 * without a pass over the 64 P bytes of the table. */
int gsr_blend_backward(void* stream, int P, int64_t R, int W, int H, const float* bg, const void* geom,
                       const void* binning, const void* image, const float* dL_dpix, float* acc, uint8_t* touched,
                       unsigned flags);
int gsr_preprocess_backward(void* stream, int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                            const float* scales, float scale_modifier, const float* rotations,
                            const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                            const float* campos, float tan_fovx, float tan_fovy, const int32_t* radii,
                            const void* geom, float* acc, float* dL_dmeans2D, float* dL_dopacity,
                            float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                            float* dL_dscales, float* dL_drots, unsigned flags /* 0 | GSR_FLAG_ACC_SELF_CLEAN */);

/* Multi-GPU exchange support (gaussianeditor_amd/multiview.py).  Per view the SH gradient is
 * rank one, dL_dsh[k] = c_k(dir) * dL_dRGB with dir = normalize(mean - campos) (backward.cu:44-98), so ranks exchange
 * the 3-float colour gradient per Gaussian and view instead of the 3M-float SH gradient:
 *   gsr_preprocess_backward_rgb = gsr_preprocess_backward, but instead of dL_dsh it writes dL_drgb (P,3): dL_dcolors
 *       with the channels the forward clamped at 0 zeroed (zeros for Gaussians the view does not see);
 *   gsr_sh_grad_compose rebuilds dL_dsh (P,M,3) = sum over v = 0..num_views-1 (ascending, binary32) of
 *       c_k(dir_v) * dL_drgb[v] from campos (num_views,3) and dL_drgb (num_views,P,3), all device arrays: bit for bit
 *       what accumulating the views' gsr_preprocess_backward outputs one after the other gives. */
int gsr_preprocess_backward_rgb(void* stream, int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                                const float* scales, float scale_modifier, const float* rotations,
                                const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                const float* campos, float tan_fovx, float tan_fovy, const int32_t* radii,
                                const void* geom, float* acc, float* dL_dmeans2D, float* dL_dopacity,
                                float* dL_dmeans3D, float* dL_dcov3D, float* dL_drgb,
                                float* dL_dscales, float* dL_drots, unsigned flags /* 0 | GSR_FLAG_ACC_SELF_CLEAN */);

/* The same kernel for gradient arrays the caller keeps ACROSS calls (a training loop's gradient bucket).  A view leaves
 * nine Gaussians of ten with all-zero gradients (culled, or blended by no pixel), and rewriting those zeros is most of
 * this kernel's traffic (248 B per Gaussian at M = 16).  row_state (P bytes, device) travels with the arrays
 * dL_dmeans2D, dL_dopacity, dL_dcolors (if given), dL_dmeans3D, dL_dsh or dL_drgb (exactly one of the two, the other NULL),
 * dL_dscales, dL_drots:
 *   row_state[g] != 0  the rows of g may hold anything: they are written (values, or zeros) -- initialise to 1;
 *   row_state[g] == 0  the rows of g hold the zeros this function wrote before: if g's gradients are zero again, nothing
 *                      is written.
 * On return row_state[g] = 1 iff g's accumulator row (`acc`) had a non-zero entry.  After the call every row holds what gsr_preprocess_backward(_rgb) would have written, for finite
 * parameters (all-zero accumulator rows give all-zero gradients; with a non-finite parameter that kernel computes 0 x inf
 * on every call, this one only when it writes the row).  Whoever else writes to these arrays must set row_state to 1 for
 * the rows it touched.  dL_dcov3D is written for every Gaussian. */
int gsr_preprocess_backward_rows(void* stream, int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                                 const float* scales, float scale_modifier, const float* rotations,
                                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                 const float* campos, float tan_fovx, float tan_fovy, const int32_t* radii,
                                 const void* geom, const float* acc, float* dL_dmeans2D, float* dL_dopacity,
                                 float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                                 float* dL_drgb, float* dL_dscales, float* dL_drots, uint8_t* row_state);
int gsr_sh_grad_compose(void* stream, int P, int D, int M, int num_views, const float* means3D, const float* campos,
                        const float* dL_drgb, float* dL_dsh);

/* K10: present[i] = (view-space z of point i) > 0.2.  Reference: checkFrustum,
 * rasterizer_impl.cu:53-63, 128-133.  `present` is one byte per Gaussian (torch.bool). */
int gsr_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present);

/* K12: semantic tracing -- every pixel scatters its C mask values to every Gaussian
 * it would blend.  weights (P,C) float and cnt (P) int32 are accumulated IN PLACE.
 * Reference: APPLY_WEIGHTS::render, apply_weights.cu:239-381.  Unlike the reference,
 * image_weights is only read for pixels inside the image (hazard a). */
int gsr_trace_weights(void* stream, int P, int64_t R, int W, int H, int C, const void* geom, const void* binning,
                      const void* image, const float* image_weights, float* weights, int32_t* cnt, unsigned flags);

/* ---- the simple-knn submodule -------------------------------------------------
 * mean_dist2[i] = mean of the squared distances from point i to its 3 nearest other points (exact), i.e.
 * `simple_knn._C.distCUDA2(points)` (gaussiansplatting/submodules/simple-knn/spatial.cu:15-25, simple_knn.cu:185-221),
 * used by GaussianModel.create_from_pcd (gaussiansplatting/scene/gaussian_model.py:288-291).
 * points (P,3) f32, mean_dist2 (P) f32, workspace: gsr_knn_workspace_size(P) bytes of device scratch. */
int gsr_knn_workspace_size(int P, size_t* bytes);
int gsr_knn_mean_dist2(void* stream, int P, const float* points, void* workspace, float* mean_dist2);

/* The Delete path's neighbour query: near[q] = 1 iff some reference point lies within dist_thresh of query point q.
 * Replaces the CPU KD-tree call inside GaussianModel.get_near_gaussians_by_mask
 * (gaussiansplatting/scene/gaussian_model.py:865-898: K_nearest_neighbors(object_xyz, 1, query=..., return_dist=True)
 * of gaussiansplatting/knn.py, a scipy.spatial.KDTree, followed by `nn_dist <= dist_thresh`).  The compared value is
 * the reference's: the Euclidean distance of the float64-widened coordinates, rounded to float32 once.
 * ref_points (n_ref,3) f32, query_points (n_query,3) f32, near (n_query) u8, nn_dist (n_query) f32 | NULL: the 1-NN
 * distance where it is <= dist_thresh * (1 + 1e-4), +inf where no reference point lies that close (the mask never needs
 * more; NULL lets a query stop at its first hit).  workspace: gsr_near_workspace_size(n_ref) bytes of device scratch.
 * n_query == 0 is a no-op; n_ref == 0 clears `near` (and fills nn_dist with +inf). */
int gsr_near_workspace_size(int n_ref, size_t* bytes);
int gsr_near_points(void* stream, int n_ref, const float* ref_points, int n_query, const float* query_points,
                    float dist_thresh, void* workspace, uint8_t* near, float* nn_dist);

/* ---- fused, row-masked Adam step over all parameter groups in one launch ----
 * Replaces, per training step, torch.optim.Adam.step() over the six groups of GaussianModel.training_setup
 * (gaussiansplatting/scene/gaussian_model.py:336-380; lr per group, betas (0.9, 0.999), eps 1e-15, no weight decay, no
 * amsgrad), the gradient-mask hooks of apply_grad_mask (:841-856) and, optionally, the gradient of anchor_loss
 * (:152-184).  Arithmetic = torch/optim/adam.py::_single_tensor_adam in binary32 (see gsr_optim.hip).
 * Every tensor is a dense float32 device array; `numel` need not be a multiple of the row length.
 *   row_len      floats per Gaussian in this tensor (3, 45, 1, 3, 4, ...): element e belongs to row e / row_len
 *   masked       != 0: the gradient of rows with row_mask[row] == 0 is replaced by 0 (the moments still decay and the
 *                parameter still moves, exactly as with the reference's hooks)
 *   anchor       NULL, or a snapshot of the parameter: anchor_scale * row_weight[row] * (param - anchor) is added to
 *                the gradient first (anchor_scale = lambda * 2 / N of the reference's mean-reduced MSE term)
 *   lr           this group's learning rate for this step
 * step = the step count AFTER this update (1 for the first), shared by all tensors; row_mask (P bytes, 0/1) and
 * row_weight (P floats) are device pointers or NULL.  At most 8 tensors per call. */
typedef struct gsr_adam_tensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  const float* anchor;
  int64_t numel;
  int32_t row_len;
  int32_t masked;
  double lr;           /* doubles where torch holds Python floats: the derived scalars are rounded to binary32 once */
  float anchor_scale;
} gsr_adam_tensor;
int gsr_adam_step(void* stream, int num_tensors, const gsr_adam_tensor* tensors, int64_t step, double beta1, double beta2,
                  double eps, const uint8_t* row_mask, const float* row_weight);
/* The same step for gradients of which only some rows were written: grad_valid (one byte per row of every tensor, device;
 * NULL = gsr_adam_step) marks them, the gradient of a row with grad_valid[row] == 0 is taken as zeros WITHOUT being read
 * (what gsr_view_messages_accumulate_rows leaves: the sums of the rows some view touched, and their mask).  Everything
 * else -- anchors, row_mask, the decay of the moments of every row -- is unchanged, so the result equals gsr_adam_step on
 * the same gradients with the invalid rows zero-filled, bit for bit. */
int gsr_adam_step_rows(void* stream, int num_tensors, const gsr_adam_tensor* tensors, int64_t step, double beta1, double beta2,
                       double eps, const uint8_t* row_mask, const float* row_weight, const uint8_t* grad_valid);

/* ---- multi-GPU exchange in "touched rows" form (new; SURVEY.md section 8(e), gaussianeditor_amd/multiview.py) ----
 * A view only produces gradients for the Gaussians it blends.  Instead of all-reducing dense (P, 14 + 3M) buffers a
 * rank packs the rows that are not entirely zero into a MESSAGE, the ranks all-gather the messages, and each rank adds
 * them per Gaussian, view 0 first: the operations and the order of one process accumulating the views one after the
 * other, bit for bit and identical on every replica (a ring all-reduce guarantees no order).
 * Message layout, 32-bit words, nb = ceil(P / 1024):
 *   [0..2] camera centre  [3] count (int32)  [4 .. 4+nb) touched rows before row block b
 *   then idx (cap x int32, ascending), means3D (cap x 3), scales (cap x 3), rotations (cap x 4), means2D (cap x 3),
 *   opacities (cap), rgb (cap x 3: the clamp-masked colour gradient gsr_preprocess_backward_rgb emits); rows >= count
 *   are padding.  All ranks use one `cap` >= every rank's count (they exchange the counts first).
 *   gsr_view_message_words        number of words of a message;
 *   gsr_view_message_plan         marks the touched rows of this view's gradients (mask: P bytes of scratch; workspace:
 *                                 gsr_compact_workspace_size(P) bytes) and returns their number in *count_host
 *                                 (one blocking readback).  With count_host == NULL nothing is read back and the call
 *                                 does not block: the count is then the uint64 at the start of `workspace`, in
 *                                 stream order (multiview.py all-gathers it from there: one host sync for all ranks'
 *                                 counts instead of two);
 *   gsr_view_message_plan_blend   the same plan from what gsr_blend_backward ALONE leaves behind -- the `touched` mask it
 *                                 writes next to its accumulator table (which then IS the plan's mask): a Gaussian no pixel
 *                                 blended has an all-zero row in the table and a 0 there, and gsr_preprocess_backward turns all-zero rows into
 *                                 all-zero gradients, so this mask is a superset of gsr_view_message_plan's (a row it
 *                                 adds carries zeros: the sums do not change).  It never blocks; the count is the uint64 at
 *                                 the start of `workspace`.  Purpose: the ranks can exchange their counts and the host can
 *                                 size the messages WHILE gsr_preprocess_backward (K8+K9, ~100 us) still runs, so the one
 *                                 host synchronisation of the exchange costs the GPU nothing;
 *   gsr_view_message_pack         writes the message (same mask / workspace).  If the view has more touched rows than
 *                                 `cap`, only the first `cap` are written (no overrun) while the header keeps the true
 *                                 count: such a message must not be accumulated -- a sender that sized its messages
 *                                 before the counts were known (multiview.py speculates on the previous steps' counts)
 *                                 sends it again with a sufficient cap;
 *   gsr_view_messages_accumulate  num_views messages, `stride_words` apart, -> the dense sums in `out` (every row of
 *                                 every array is written: Gaussians no view touched get zeros); out->sh, if not NULL,
 *                                 is rebuilt from the colour gradients exactly as gsr_sh_grad_compose does (means3D,
 *                                 active degree D, M coefficients).
 * local->sh is ignored by plan / pack. */
typedef struct gsr_dense_grads {
  float* means3D;   /* (P,3) */
  float* scales;    /* (P,3) */
  float* rotations; /* (P,4) */
  float* means2D;   /* (P,3) */
  float* opacities; /* (P,1) */
  float* sh;        /* (P,M,3) or NULL */
} gsr_dense_grads;
int gsr_view_message_words(int64_t P, int64_t cap, int64_t* words);
int gsr_view_message_plan(void* stream, int64_t P, const gsr_dense_grads* local, const float* rgb, uint8_t* mask,
                          void* workspace, int64_t* count_host);
int gsr_view_message_plan_blend(void* stream, int64_t P, const uint8_t* touched, void* workspace);
int gsr_view_message_pack(void* stream, int64_t P, const gsr_dense_grads* local, const float* rgb, const float* campos,
                          const uint8_t* mask, void* workspace, int64_t cap, float* message);
int gsr_view_messages_accumulate(void* stream, int64_t P, int D, int M, int num_views, const float* messages,
                                 int64_t stride_words, int64_t cap, const float* means3D, const gsr_dense_grads* out);
/* The same sums, written only where a view sent a row: row_valid (P bytes, device; NULL = gsr_view_messages_accumulate)
 * receives 1 for the Gaussians some view touched and 0 for the others, whose rows of `out` are NOT written (they keep their
 * previous contents).  The consumer treats those rows as zero gradients: gsr_adam_step_rows with grad_valid = row_valid takes
 * an invalid row's gradient as zeros WITHOUT reading it and still decays that row's moments, bit for bit what gsr_adam_step
 * does on zero-filled gradients.  (NOT gsr_adam_step's `row_mask`: a masked row keeps its moments from decaying, which is the
 * reference's apply_grad_mask semantics, not a zero gradient.)  A view touches ~10 % of the benchmark scene, so with 2 / 8
 * views 81 / 43 % of the 248 B per Gaussian are neither written here nor read there. */
int gsr_view_messages_accumulate_rows(void* stream, int64_t P, int D, int M, int num_views, const float* messages,
                                      int64_t stride_words, int64_t cap, const float* means3D, const gsr_dense_grads* out,
                                      uint8_t* row_valid);

/* ---- prune = stable compaction of the rows of many tensors by one mask ----
 * Replaces the per-tensor boolean-mask indexing of GaussianModel.prune_points / _prune_optimizer
 * (gaussiansplatting/scene/gaussian_model.py:568-609: parameters, Adam moments, bookkeeping tensors).
 *   gsr_compact_plan   scans keep (P bytes, 0/1, device) into `workspace` (gsr_compact_workspace_size(P) bytes of device
 *                      scratch) and returns the number of surviving rows in *kept_host (one blocking readback: the
 *                      caller sizes the outputs with it);
 *   gsr_compact_apply  one launch: for every tensor copies the surviving rows, in their original order, from src
 *                      (P rows of row_bytes) to dst (kept rows) -- what `tensor[mask]` returns.  At most 32 tensors per
 *                      call; src and dst must not overlap. */
typedef struct gsr_compact_tensor {
  const void* src;
  void* dst;
  int64_t row_bytes;
} gsr_compact_tensor;
int gsr_compact_workspace_size(int64_t P, size_t* bytes);
int gsr_compact_plan(void* stream, int64_t P, const uint8_t* keep, void* workspace, int64_t* kept_host);
int gsr_compact_apply(void* stream, int64_t P, const uint8_t* keep, void* workspace, int num_tensors,
                      const gsr_compact_tensor* tensors);

/* ---- SURVEY.md section 8(f) rank 4, second half: densification = rows APPENDED to many tensors in one launch ----
 * Replaces cat_tensors_to_optimizer (gaussiansplatting/scene/gaussian_model.py:609-641), which per parameter group runs
 * torch.cat on the parameter and on its two Adam moments (the moments extended by torch.zeros_like): 18 cats + 12 fills
 * per densification step (densify_and_clone :730-766 and densify_and_split :673-728 both end in it through
 * densification_postfix :643-671).  For every tensor: dst (P + n rows) = src (P rows) followed by ext (n rows), or by n
 * zero rows when ext == NULL -- what torch.cat((src, ext)) / torch.cat((src, zeros_like(ext))) return, bit for bit.
 * The caller allocates dst; src / ext / dst must not overlap; at most 32 tensors per call.  Which rows are appended (the
 * clone mask, the split samples) stays with the caller: a clone's ext is gsr_compact_apply's output for that mask. */
typedef struct gsr_append_tensor {
  const void* src;   /* P rows */
  const void* ext;   /* n rows, or NULL for zeros */
  void* dst;         /* P + n rows */
  int64_t row_bytes;
} gsr_append_tensor;
int gsr_append_rows(void* stream, int64_t P, int64_t n, int num_tensors, const gsr_append_tensor* tensors);

/* ---- introspection used by the parity tests (not needed by the drop-in) ----
 * Copy internal per-Gaussian / per-instance / per-pixel state out of the opaque
 * scratch buffers into caller-provided DEVICE arrays (any may be NULL):
 *   means2D (P,2) f32, depths (P) f32, rgb (P,3) f32, conic_opacity (P,4) f32,
 *   tiles_touched (P) u32, clamped (P,3) u8;
 *   keys (R) u64 sorted, point_list (R) u32 sorted; ranges (T,2) u32; final_T (N) f32; n_contrib (N) u32.
 * The 3D covariance (reference: geomState.cov3D, rasterizer_impl.cu:225) is not kept in `geom`: forward and backward
 * both compute it from scales / rotations with the same operations.  gsr_debug_cov3d evaluates that function for all
 * P Gaussians into cov3D (P,6). */
int gsr_debug_export_geom(void* stream, int P, const void* geom, float* means2D, float* depths, float* rgb,
                          float* conic_opacity, uint32_t* tiles_touched, uint8_t* clamped);
int gsr_debug_cov3d(void* stream, int P, const float* scales, float scale_modifier, const float* rotations, float* cov3D);
int gsr_debug_export_binning(void* stream, int P, int64_t R, int W, int H, const void* geom, const void* binning,
                             uint64_t* keys, uint32_t* point_list);
int gsr_debug_export_image(void* stream, int W, int H, const void* image, uint32_t* ranges, float* final_T,
                           uint32_t* n_contrib);

/* Performance introspection (tools/wave_profile.py): run K6 with per-wave instrumentation.  One record
 * of 8 x u64 per launched workgroup: {s_memtime at start, at end, XCC_ID<<32 | HW_ID,
 * items<<32 | entries evaluated, cycles chunk start -> survivors staged, cycles in the group loops, chunks walked,
 * cycles chunk start -> cull ballot}; records of workgroups that exit before doing any work stay zero.  *n_records_host = number of workgroups (call with max_records = 0
 * to query; returns GSR_ERR_BAD_ARGUMENT in that case after setting it). */
/* The same for K7: one record of 8 x u64 per 4-wave workgroup: {start, end, XCC_ID<<32 | HW_ID, tiles<<32 | forward work
 * estimate of those tiles, longest tile (ticks), first tile (ticks), 0, 0}.  Arguments as gsr_blend_backward. */
int gsr_debug_blend_backward_profile(void* stream, int P, int64_t R, int W, int H, const float* bg, const void* geom,
                                     const void* binning, const void* image, const float* dL_dpix, float* acc,
                                     uint64_t* records, int64_t max_records, int64_t* n_records_host);
int gsr_debug_blend_forward_profile(void* stream, int P, int64_t R, int W, int H, const float* bg, const void* geom,
                                    const void* binning, void* image, float* out_color, float* out_depth,
                                    uint64_t* records, int64_t max_records, int64_t* n_records_host);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GSR_H_INCLUDED */
