// gsr_capi.hip -- the C ABI declared in include/gsr.h: argument checking, scratch carving and
// kernel sequencing.  No device memory is allocated here and no option state is kept (behaviour flags arrive with every
// call); the only state is the pool of pinned readback slots (host_slot).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>


#include "gsr_kernels.h"

using namespace gsr;

namespace {
thread_local int g_last_hip_error = 0;
inline int hip_fail(hipError_t e) {
  g_last_hip_error = (int)e;
  return GSR_ERR_HIP;
}
#define GSR_HIP(expr)                          \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) return hip_fail(_e); \
  } while (0)

// Pinned landing slots of the num_rendered readback.  A host thread checks one slot per device out of a process-wide
// pool on its first gsr_preprocess for that device and hands it back when the thread exits (no HIP call is made in a
// thread destructor: the runtime may already be gone); slots are reused by later threads and never freed, so the
// pool is bounded by the largest number of threads that were inside the library at once.  This is the only state the
// library keeps; it owns no device memory.
struct HostSlot {
  uint32_t* words = nullptr;      // GEOM_HDR_BYTES, pinned (fine-grained) and mapped into the device's address space
  uint32_t* dev_words = nullptr;  // the same memory as the device sees it
  uint32_t seq = 0;               // generation of the last readback; the device stores it behind the data
  int device = -1;
  HostSlot* next_free = nullptr;
};
constexpr int MAX_DEV = 64;
std::mutex g_pool_mutex;
HostSlot* g_pool_free[MAX_DEV] = {};  // per device: slots no thread holds at the moment

struct ThreadSlots {
  HostSlot* held[MAX_DEV] = {};
  ~ThreadSlots() {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (int d = 0; d < MAX_DEV; ++d)
      if (held[d] != nullptr) {
        held[d]->next_free = g_pool_free[d];
        g_pool_free[d] = held[d];
      }
  }
};

// A fresh slot for device `dev` (pinned, device-mapped, zeroed).
HostSlot* new_slot(int dev, hipError_t* err) {
  int cur = -1;
  const bool switched = hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess;
  void* p = nullptr;
  void* dp = nullptr;
  HostSlot* h = nullptr;
  if ((*err = hipHostMalloc(&p, GEOM_HDR_BYTES, hipHostMallocPortable | hipHostMallocMapped)) == hipSuccess) {
    if ((*err = hipHostGetDevicePointer(&dp, p, 0)) == hipSuccess) {
      memset(p, 0, GEOM_HDR_BYTES);
      h = new HostSlot();
      h->words = (uint32_t*)p;
      h->dev_words = (uint32_t*)dp;
      h->device = dev;
    } else {
      (void)hipHostFree(p);
    }
  }
  if (switched) (void)hipSetDevice(cur);
  return h;
}

// The slot of the calling thread for the device `stream` belongs to (NOT the thread's current device: a C-ABI caller
// may hand over a stream of another device).
// On failure *err tells why: hipSuccess = the device ordinal is outside the pool (an argument problem), otherwise the
// failing runtime call's error.
HostSlot* host_slot(hipStream_t stream, hipError_t* err) {
  static thread_local ThreadSlots mine;
  int dev = 0;
  hipDevice_t sdev = 0;
  *err = hipSuccess;
  if (stream != nullptr && hipStreamGetDevice(stream, &sdev) == hipSuccess)
    dev = (int)sdev;  // (hipDevice_t is the ordinal)
  else if ((*err = hipGetDevice(&dev)) != hipSuccess)
    return nullptr;
  if (dev < 0 || dev >= MAX_DEV) return nullptr;
  if (mine.held[dev] != nullptr) return mine.held[dev];
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (g_pool_free[dev] != nullptr) {
      HostSlot* h = g_pool_free[dev];
      g_pool_free[dev] = h->next_free;
      h->next_free = nullptr;
      mine.held[dev] = h;
      return h;
    }
  }
  HostSlot* h = new_slot(dev, err);
  mine.held[dev] = h;
  return h;
}

// Host side of a device-written answer: poll the slot's generation word until the device has stored `seq` behind its data.
// The spin is BOUNDED in time: after 5 ms (a long queue of earlier work on the stream, or a failed launch) the thread stops
// burning a core and blocks in hipStreamSynchronize, which also surfaces any launch error; the word is then either there or
// the call fails.
hipError_t wait_for_slot(const HostSlot* slot, uint32_t seq, hipStream_t s) {
  volatile const uint32_t* flag = slot->words + GEOM_HDR_FINAL;
  bool seen = false;
  const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(5);
  for (uint32_t spin = 1;; ++spin) {
    if (*flag == seq) {
      seen = true;
      break;
    }
    __builtin_ia32_pause();
    if ((spin & 0x3ffu) == 0 && std::chrono::steady_clock::now() >= give_up) break;
  }
  if (!seen) {
    const hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    if (*flag != seq) return hipErrorUnknown;  // the stream is idle and nothing was published
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return hipSuccess;
}

// gsr_arrays_equal: up to 8 pairs of device arrays compared bit for bit in one launch.  A mismatch is reported straight
// into the caller thread's pinned slot (word GEOM_HDR_DIFFER: at most one store per wave, none when the arrays agree).
constexpr int EQ_MAX_PAIRS = 8;
constexpr int GEOM_HDR_DIFFER = 8;  // host slot only (GEOM_HDR_* of gsr_common.h end at 6)
struct EqualArgs {
  const uint32_t* a[EQ_MAX_PAIRS];
  const uint32_t* b[EQ_MAX_PAIRS];
  unsigned long long words[EQ_MAX_PAIRS];
  int n;
};
__global__ void __launch_bounds__(256) arrays_equal_kernel(const EqualArgs q, uint32_t* __restrict__ slot) {
  bool differ = false;
  const unsigned long long stride = (unsigned long long)gridDim.x * 256ull, t0 = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  for (int i = 0; i < q.n; ++i) {
    const uint32_t* a = q.a[i];
    const uint32_t* b = q.b[i];
    const unsigned long long nw = q.words[i];
    if (a == b) continue;
    if ((((uintptr_t)a | (uintptr_t)b) & 15u) == 0) {
      const uint4* a4 = reinterpret_cast<const uint4*>(a);
      const uint4* b4 = reinterpret_cast<const uint4*>(b);
      const unsigned long long nv = nw / 4;
      unsigned long long v = t0;
      for (; v + 3 * stride < nv; v += 4 * stride) {  // four independent 16-byte loads per array in flight
        uint4 x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          x[u] = a4[v + (unsigned long long)u * stride];
          y[u] = b4[v + (unsigned long long)u * stride];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) differ = differ || x[u].x != y[u].x || x[u].y != y[u].y || x[u].z != y[u].z || x[u].w != y[u].w;
      }
      for (; v < nv; v += stride) {
        const uint4 x = a4[v], y = b4[v];
        differ = differ || x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w;
      }
      for (unsigned long long w = nv * 4 + t0; w < nw; w += stride) differ = differ || a[w] != b[w];
    } else {
      for (unsigned long long w = t0; w < nw; w += stride) differ = differ || a[w] != b[w];
    }
  }
  if (__any(differ) && lane_id() == 0) __hip_atomic_store(slot + GEOM_HDR_DIFFER, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (a launch of its own: in stream order behind every store of the compare kernel)
__global__ void publish_seq_kernel(uint32_t* __restrict__ slot, uint32_t seq) {
  __threadfence_system();
  __hip_atomic_store(slot + GEOM_HDR_FINAL, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Guard against the one misuse of GSR_FLAG_FORWARD_ONLY that nothing on the device can report: a backward on a geometry
// state whose gsr_preprocess skipped what only the backward reads.  gsr_preprocess notes (geom pointer -> forward-only?) in
// a small fixed-size table, the backward entry points look their `geom` up and refuse with GSR_ERR_BAD_ARGUMENT instead of
// reading uninitialised memory.  Best effort by construction -- an entry is overwritten by the next gsr_preprocess on the
// same buffer and may be evicted by one on another buffer that hashes to its slot (then the misuse goes unnoticed, as it did
// before); a state that was COPIED elsewhere is unknown.  No device work, no synchronisation.
struct GeomNote {
  const void* geom = nullptr;
  bool forward_only = false;
};
constexpr size_t GEOM_NOTES = 1024;
std::mutex g_notes_mutex;
GeomNote g_notes[GEOM_NOTES];
inline size_t note_slot(const void* p) { return (size_t)(((uintptr_t)p >> 8) * 0x9E3779B97F4A7C15ull >> 54) % GEOM_NOTES; }
inline void note_geom(const void* geom, bool forward_only) {
  std::lock_guard<std::mutex> lock(g_notes_mutex);
  GeomNote& n = g_notes[note_slot(geom)];
  n.geom = geom;
  n.forward_only = forward_only;
}
inline bool geom_is_forward_only(const void* geom) {
  std::lock_guard<std::mutex> lock(g_notes_mutex);
  const GeomNote& n = g_notes[note_slot(geom)];
  return n.geom == geom && n.forward_only;
}

// Which binning path an image takes (gsr_binning.hip): a function of the image size only, so that gsr_scratch_sizes,
// gsr_preprocess, gsr_bin and the blend entry points of one view agree.  GSR_BIN_LEGACY=1 (tests) forces the pair sort.
inline int bin_legacy(int W, int H) {
  static const bool forced = [] {
    const char* e = getenv("GSR_BIN_LEGACY");
    return e != nullptr && e[0] == '1';
  }();
  // (the grouped path packs a tile rectangle into 32 bits: images of up to RECT32_EDGE tiles per side)
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  return (forced || group_count(W, H) > GROUP_MAX || gx > RECT32_EDGE || gy > RECT32_EDGE) ? 1 : 0;
}
// Whether a view's forward leaves checkpoints, and where the first one sits (in 64-entry chunks): where lists are LONG
// (deep-tile scenes, 6 M Gaussians: the backward walks up to 7 000 positions of a tile's list; blend backward 438 -> 339 us
// and 528 -> 314 us when introduced at a uniform 512 positions), none where they are short.  On the benchmark view (mean list 593 entries per tile) the heaviest backward
// items are dense tiles walked 350-450 positions deep, the hundred longest within 13 % of each other and of a workgroup's
// mean load: cutting them (stride 128 / 192 / 256, any work threshold) adds the per-item start-up of a few hundred more
// items and buys no balance -- blend backward 219 -> 226-232 us in four same-box A/Bs (profiles/r04_d_segments.md) -- and at
// strides below 256 the segments' start state shows in the per-row parity bar.  A function of what gsr_blend_forward
// and gsr_blend_backward of a view both receive (R, W, H), so the two agree.  GSR_CK_CHUNKS overrides it (tests use 1 or 2
// to segment small scenes; 0 = never).  Never changes a result beyond the backward's summation order.
// Strides below 4 chunks (256 positions) are never a gain, and in round 4 -- when a list segment's start state was a
// difference of two binary32 accumulators of the forward -- they cost precision (profiles/r04_d_segments.md; since round 5
// the forward accumulates every segment's colour separately, gsr_blend.hip): the override is clamped to >= 4 unless
// GSR_CK_DEBUG=1 says the caller knows (the tolerance tests segment small scenes at stride 64).
inline int checkpoint_chunks(int64_t R, int W, int H) {
  static const int env = [] {
    const char* e = getenv("GSR_CK_CHUNKS");
    const char* d = getenv("GSR_CK_DEBUG");
    int c = e != nullptr ? atoi(e) : -1;
    if (c > 0 && c < 4 && !(d != nullptr && d[0] == '1')) c = 4;
    return c > 1024 ? 1024 : c;
  }();
  if (env >= 0) return env;
  // Round 6, second half (profiles/r06_m_fine_checkpoints.md, r06_n_checkpoint_table.md): what the stride sweeps of r06_k had
  // charged to "fine strides" was the REACH of 8 slots; and what a tile needs is fine cuts where its work is -- in front,
  // where its pixels are still live -- and reach for the stragglers behind.  Every view with checkpoints now takes all
  // CK_MAX slots at the positions of `checkpoint_table`: 256 apart in front, 4 096 apart at the end, 16 384 of reach.
  // Images of up to 4 096 tiles (few tiles per workgroup of the backward: a tile must be cut to fill the chip) turn
  // checkpoints on from a mean list of 1 200 entries, larger ones from 2 048 (the 1080p synth-v2 view at 1 143 loses
  // 18 % of its backward with them).  GSR_CK_MIN_LIST overrides the threshold (tuning knob).
  static const int64_t env_min_list = [] {
    const char* e = getenv("GSR_CK_MIN_LIST");
    return (int64_t)(e != nullptr && atoi(e) > 0 ? atoi(e) : 0);
  }();
  const int64_t T = (int64_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
  const int64_t min_list = env_min_list > 0 ? env_min_list : (T <= 4096 ? 1200 : 2048);
  return R < min_list * T ? 0 : CK_CHUNKS_DEFAULT;
}
// Checkpoint slots in use per tile: all CK_MAX (the pool and ck_work are laid out with the count in use as their stride).
// GSR_CK_SLOTS overrides (2 .. CK_MAX; tests).
inline int checkpoint_slots(int64_t R, int W, int H) {
  static const int env = [] {
    const char* e = getenv("GSR_CK_SLOTS");
    const int v = e != nullptr ? atoi(e) : 0;
    return v < 2 ? 0 : (v > CK_MAX ? CK_MAX : v);
  }();
  (void)R; (void)W; (void)H;
  return env > 0 ? env : CK_MAX;
}
// The positions of a view's checkpoints (CkTable, gsr_common.h): 256 positions apart in front, then 512, 1 024, 2 048 and
// 4 096 -- 16 384 of reach with 16 slots, the first 1 536 positions cut as finely as a uniform stride of 256 cuts them.
// Measured against uniform tables of every stride on twenty views (profiles/r06_n_checkpoint_table.md): equal where lists
// are short, 1.2-3x faster backward where they are long (a 512 x 512 view of 6 M Gaussians: K7 778 -> 247 us).
// GSR_CK_CHUNKS (tests, sweeps) or GSR_CK_GEOM=0: uniform, k * stride.
inline CkTable checkpoint_table(int64_t R, int W, int H) {
  static const bool env_chunks = getenv("GSR_CK_CHUNKS") != nullptr;
  static const bool geom = [] { const char* e = getenv("GSR_CK_GEOM"); return !(e && e[0] == '0'); }();
  static const uint16_t fine[CK_MAX] = {0, 4, 8, 12, 16, 20, 24, 32, 40, 48, 64, 80, 96, 128, 192, 256};
  static_assert(CK_MAX == 16 && CK_CHUNKS_DEFAULT == 4, "the table above is written for 16 slots, the first 4 chunks in");
  // GSR_CK_TABLE="c1,c2,...,c15" (sweeps): the checkpoints' positions in 64-entry chunks, ascending
  static const CkTable env_table = [] {
    CkTable e;
    memset(&e, 0, sizeof(e));
    const char* p = getenv("GSR_CK_TABLE");
    for (int k = 1; p != nullptr && *p && k < CK_MAX; ++k) {
      const long v = strtol(p, const_cast<char**>(&p), 10);
      e.chunk[k] = (uint16_t)(v > e.chunk[k - 1] ? (v > 65535 ? 65535 : v) : e.chunk[k - 1] + 1);
      if (*p == ',') ++p;
      if (k == CK_MAX - 1) e.chunk[0] = 1;  // (marks a complete table; put back to 0 below)
    }
    return e;
  }();
  CkTable t;
  const int c = checkpoint_chunks(R, W, H);
  const bool use_fine = geom && !env_chunks;
  for (int k = 0; k < CK_MAX; ++k) {
    const int u = k * c;
    t.chunk[k] = use_fine ? fine[k] : (uint16_t)(u > 65535 ? 65535 : u);
  }
  if (use_fine && env_table.chunk[0] == 1) {
    t = env_table;
    t.chunk[0] = 0;
  }
  return t;
}
// What the blend / export entry points need of the binning scratch sits in front of everything sized by the number of
// group instances, so they carve with G = 0.
inline Binning carve_binning_view(const void* binning, int64_t R, int W, int H) {
  return carve_binning(const_cast<void*>(binning), R, 0, W, H, bin_legacy(W, H));
}
// Scratch buffers are accessed with 16-byte vector loads at 256-byte aligned section offsets.
inline bool misaligned(const void* p) { return ((uintptr_t)p & 255u) != 0; }

inline BlendArgs make_blend_args(int W, int H, const Geom& g, const Binning& b, const Image& im, const float* bg,
                                 int queue_kind, int64_t R) {
  BlendArgs a;
  memset(&a, 0, sizeof(a));
  a.W = W;
  a.H = H;
  a.gx = (W + TILE - 1) / TILE;
  a.gy = (H + TILE - 1) / TILE;
  a.work_order = im.work_order;
  a.work_meta = im.work_meta;
  a.work_est = im.work_est;
  a.work_maxc = im.work_maxc;
  a.bwd_order = im.bwd_order;
  a.bwd_meta = im.bwd_meta;
  a.ranges = im.ranges;
  a.point_list = b.point_list;
  a.rec0 = g.rec0;
  a.rec1 = g.rec1;
  a.rec2 = g.rec2;
  a.bg = bg;
  a.final_T = im.final_T;
  a.n_contrib = im.n_contrib;
  a.queue = im.queue_heads + (size_t)queue_kind * QUEUE_LINES * QUEUE_STRIDE;
  a.ck_chunks = checkpoint_chunks(R, W, H);
  a.ck_slots = checkpoint_slots(R, W, H);
  a.ck_pos = checkpoint_table(R, W, H);
  if (a.ck_chunks > 0) {
    a.ck_table = im.ck_table;
    a.ck_work = im.ck_work;
    a.tile_maxc = im.tile_maxc;
    a.ck_pool = im.ck_pool;
  }
  return a;
}
}  // namespace

extern "C" {

int gsr_abi_version(void) { return GSR_ABI_VERSION; }

const char* gsr_status_string(int status) {
  switch (status) {
    case GSR_OK: return "ok";
    case GSR_ERR_BAD_ARGUMENT: return "bad argument";
    case GSR_ERR_BAD_CHANNELS: return "unsupported number of channels (apply_weights supports 1, 2 or 3)";
    case GSR_ERR_TOO_MANY: return "number of rendered instances exceeds the 31-bit index space";
    case GSR_ERR_HIP: return "HIP runtime error";
    default: return "unknown status";
  }
}

int gsr_last_hip_error(void) { return g_last_hip_error; }

int gsr_sort_key_bits(int W, int H) { return sort_key_bits(W, H); }

int gsr_scratch_sizes(int P, int64_t R, int64_t G, int W, int H, size_t sizes[3]) {
  if (P < 0 || R < 0 || G < 0 || W <= 0 || H <= 0 || sizes == nullptr) return GSR_ERR_BAD_ARGUMENT;
  sizes[0] = carve_geom(nullptr, P).bytes;
  sizes[1] = carve_binning(nullptr, R, G, W, H, bin_legacy(W, H)).bytes;
  // (the checkpoint pool -- up to 64 MB, the LAST section of the image scratch -- only for views whose forward writes
  //  checkpoints: the same predicate of (R, W, H) the blend entry points use.  R = 0, "not known yet": the size WITH the
  //  pool, so that a buffer sized before gsr_preprocess is never too small)
  sizes[2] = carve_image(nullptr, W, H, R == 0 || checkpoint_chunks(R, W, H) > 0).bytes;
  return GSR_OK;
}

namespace {
// A slot of its own for a split readback (gsr_preprocess_begin .. _end): taken from the pool, not the thread's held one, so
// that any number of views may be in flight on one thread and _end may run on another.
HostSlot* checkout_slot(hipStream_t stream, hipError_t* err) {
  int dev = 0;
  hipDevice_t sdev = 0;
  *err = hipSuccess;
  if (stream != nullptr && hipStreamGetDevice(stream, &sdev) == hipSuccess)
    dev = (int)sdev;
  else if ((*err = hipGetDevice(&dev)) != hipSuccess)
    return nullptr;
  if (dev < 0 || dev >= MAX_DEV) return nullptr;
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (g_pool_free[dev] != nullptr) {
      HostSlot* h = g_pool_free[dev];
      g_pool_free[dev] = h->next_free;
      h->next_free = nullptr;
      return h;
    }
  }
  return new_slot(dev, err);
}
void return_slot(HostSlot* h) {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  h->next_free = g_pool_free[h->device];
  g_pool_free[h->device] = h;
}

// The argument checks of gsr_preprocess / gsr_preprocess_begin (before anything touches the runtime: they answer on a machine
// without a GPU too).
int preprocess_args_ok(int P, int D, int M, const float* means3D, const float* scales, const float* rotations,
                       const float* opacities, const float* shs, const float* cov3D_precomp, const float* colors_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* campos, int W, int H, int skip_color,
                       unsigned flags, const int32_t* radii, const void* geom) {
  if (P < 0 || W <= 0 || H <= 0 || D < 0 || D > 3 || (flags & ~GSR_FLAG_ALL)) return GSR_ERR_BAD_ARGUMENT;
  if (!means3D || !opacities || !viewmatrix || !projmatrix || !radii || !geom || misaligned(geom)) return GSR_ERR_BAD_ARGUMENT;
  if (cov3D_precomp == nullptr && (scales == nullptr || rotations == nullptr)) return GSR_ERR_BAD_ARGUMENT;
  if (!skip_color) {
    // the reference throws std::runtime_error for "non-RGB without precomputed colours"
    // (rasterizer_impl.cu:210-213); with NUM_CHANNELS == 3 the only failure left is "no colour source".
    if (colors_precomp == nullptr && (shs == nullptr || campos == nullptr)) return GSR_ERR_BAD_ARGUMENT;
    if (colors_precomp == nullptr && M < (D + 1) * (D + 1)) return GSR_ERR_BAD_ARGUMENT;
  }
  return GSR_OK;
}

// First half of gsr_preprocess: K1 and the two depth-sort passes that never depend on the key range; the
// first of them publishes the counts into `slot` behind generation `*seq`.
int preprocess_begin_impl(hipStream_t s, int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                          const float* rotations, const float* opacities, const float* shs, const float* cov3D_precomp,
                          const float* colors_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                          int W, int H, float tan_fovx, float tan_fovy, int skip_color, unsigned flags, int32_t* radii,
                          void* geom, HostSlot* slot, uint32_t* seq_out) {
  PreArgs a;
  a.P = P; a.D = D; a.M = M;
  a.means3D = means3D; a.scales = scales; a.scale_modifier = scale_modifier; a.rotations = rotations;
  a.opacities = opacities; a.shs = shs; a.cov3D_precomp = cov3D_precomp; a.colors_precomp = colors_precomp;
  a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.campos = campos;
  a.W = W; a.H = H; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
  a.focal_y = H / (2.0f * tan_fovy);  // rasterizer_impl.cu:190-191
  a.focal_x = W / (2.0f * tan_fovx);
  a.gx = (W + TILE - 1) / TILE; a.gy = (H + TILE - 1) / TILE;
  a.skip_color = skip_color;
  a.forward_only = (flags & GSR_FLAG_FORWARD_ONLY) ? 1 : 0;
  a.tile_bounds = (flags & GSR_FLAG_TILE_BOUNDS_ALPHA) ? 1 : 0;
  a.radii = radii;
  a.g = carve_geom(geom, P);
  note_geom(geom, a.forward_only != 0);
  GSR_HIP(launch_preprocess(s, a));
  // The one blocking readback of the path (reference: cudaMemcpy, rasterizer_impl.cu:236-239).  K1 already holds
  // num_rendered (and the range of the depth keys); the first kernel behind it -- the histogram of the first depth-sort
  // pass -- writes those words into pinned, device-mapped host memory, a generation word last.  The first two passes of
  // the depth sort are enqueued at once and run while the host waits.
  const uint32_t seq = ++slot->seq ? slot->seq : ++slot->seq;  // (never 0: that is what a fresh slot holds)
  *seq_out = seq;
  if (bin_legacy(W, H)) GSR_HIP(launch_depth_passes(s, P, a.g, 0, 2, slot->dev_words, seq));
  else GSR_HIP(launch_depth_passes_grouped(s, P, a.g, 0, 2, false, slot->dev_words, seq));
  return GSR_OK;
}

// Second half: wait for the counts, launch the depth-sort passes that needed the key range.
int preprocess_end_impl(hipStream_t s, int P, int W, int H, void* geom, HostSlot* slot, uint32_t seq, int64_t counts_host[2]) {
  // Poll the generation word (an event would put a barrier packet into the stream -- a 6 us bubble -- and sleeping on
  // an interrupt costs far more than the ~80 us normally waited for).  The spin is BOUNDED in time: after 5 ms (a long
  // queue of earlier work on the stream, or a failed launch) the thread stops burning a core and blocks in
  // hipStreamSynchronize, which also surfaces any launch error; the word is then either there or the call fails.
  GSR_HIP(wait_for_slot(slot, seq, s));
  const Geom g = carve_geom(geom, P);
  const int gx = (W + TILE - 1) / TILE;
  const int legacy = bin_legacy(W, H);
  const uint32_t* w = slot->words;
  const uint64_t total = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
  const uint32_t kmax = w[GEOM_HDR_KEYMAX], kinv = w[GEOM_HDR_KEYINVMAX];
  // bits above the highest one in which min and max key differ are a common prefix of every visible key
  const uint32_t kmin = ~kinv;
  int nbits = 0;
  for (uint32_t d = total ? (kmax ^ kmin) : 0u; d; d >>= 1) ++nbits;
  if (legacy) {
    const int passes = nbits <= 16 ? 2 : (nbits + 7) / 8;
    if (passes > 2) GSR_HIP(launch_depth_passes(s, P, g, 2, passes));
    GSR_HIP(launch_depth_finish(s, P, g, passes, gx, 0));
  } else {
    // The LAST pass also leaves what the emission of gsr_bin needs (depth_scatter_kernel<true>), and which pass that is
    // is only known now: at least one pass follows the two that ran under the wait (a key range of <= 16 bits: its
    // digit is the same for every key -- a stable pass that moves nothing).
    const int passes = nbits <= 24 ? 3 : 4;
    GSR_HIP(launch_depth_passes_grouped(s, P, g, 2, passes, true));
  }
  if (total >= (1ull << 31)) return GSR_ERR_TOO_MANY;
  counts_host[0] = (int64_t)total;
  counts_host[1] = legacy ? 0 : (int64_t)((uint64_t)w[GEOM_HDR_GROUPS] | ((uint64_t)w[GEOM_HDR_GROUPS + 1] << 32));
  return GSR_OK;
}
}  // namespace

int gsr_preprocess(void* stream, int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                   const float* rotations, const float* opacities, const float* shs, const float* cov3D_precomp,
                   const float* colors_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                   int W, int H, float tan_fovx, float tan_fovy, int prefiltered, int skip_color, unsigned flags,
                   int32_t* radii, void* geom, int64_t counts_host[2]) {
  // `prefiltered`: the reference promises with it that no point fails the frustum test and TRAPS the device when one
  // does (auxiliary.h:156-160).  No caller sets it (render() passes False, gaussian_renderer/__init__.py:83); a culled
  // point is simply culled here, which is what the flag's absence does.
  (void)prefiltered;
  if (counts_host == nullptr) return GSR_ERR_BAD_ARGUMENT;
  counts_host[0] = counts_host[1] = 0;
  if (P == 0) return GSR_OK;
  const int ok = preprocess_args_ok(P, D, M, means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp,
                                    viewmatrix, projmatrix, campos, W, H, skip_color, flags, radii, geom);
  if (ok != GSR_OK) return ok;
  hipStream_t s = (hipStream_t)stream;
  hipError_t slot_err = hipSuccess;
  HostSlot* slot = host_slot(s, &slot_err);
  if (slot == nullptr) return slot_err == hipSuccess ? GSR_ERR_BAD_ARGUMENT /* device ordinal beyond the pool */ : hip_fail(slot_err);
  uint32_t seq = 0;
  const int st = preprocess_begin_impl(s, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp,
                                       colors_precomp, viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, skip_color,
                                       flags, radii, geom, slot, &seq);
  if (st != GSR_OK) return st;
  return preprocess_end_impl(s, P, W, H, geom, slot, seq, counts_host);
}

int gsr_preprocess_begin(void* stream, int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                         const float* rotations, const float* opacities, const float* shs, const float* cov3D_precomp,
                         const float* colors_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                         int W, int H, float tan_fovx, float tan_fovy, int prefiltered, int skip_color, unsigned flags,
                         int32_t* radii, void* geom, void** ticket) {
  (void)prefiltered;
  if (ticket == nullptr) return GSR_ERR_BAD_ARGUMENT;
  *ticket = nullptr;
  if (P <= 0) return GSR_ERR_BAD_ARGUMENT;  // (nothing to wait for: an empty scene takes gsr_preprocess)
  const int ok = preprocess_args_ok(P, D, M, means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp,
                                    viewmatrix, projmatrix, campos, W, H, skip_color, flags, radii, geom);
  if (ok != GSR_OK) return ok;
  hipStream_t s = (hipStream_t)stream;
  hipError_t slot_err = hipSuccess;
  HostSlot* slot = checkout_slot(s, &slot_err);
  if (slot == nullptr) return slot_err == hipSuccess ? GSR_ERR_BAD_ARGUMENT : hip_fail(slot_err);
  uint32_t seq = 0;
  const int st = preprocess_begin_impl(s, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp,
                                       colors_precomp, viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, skip_color,
                                       flags, radii, geom, slot, &seq);
  if (st != GSR_OK) {
    return_slot(slot);
    return st;
  }
  *ticket = slot;  // (its `seq` is the generation to wait for: a slot serves one view at a time)
  return GSR_OK;
}

int gsr_preprocess_end(void* stream, int P, int W, int H, void* geom, void* ticket, int64_t counts_host[2]) {
  if (ticket == nullptr || counts_host == nullptr || P <= 0 || W <= 0 || H <= 0 || !geom || misaligned(geom))
    return GSR_ERR_BAD_ARGUMENT;
  counts_host[0] = counts_host[1] = 0;
  HostSlot* slot = (HostSlot*)ticket;
  const int st = preprocess_end_impl((hipStream_t)stream, P, W, H, geom, slot, slot->seq, counts_host);
  return_slot(slot);  // (also after a failure: the ticket is spent either way)
  return st;
}

int gsr_arrays_equal(void* stream, int n, const void* const* a, const void* const* b, const size_t* bytes, int* equal_host) {
  if (equal_host == nullptr || n < 0 || n > EQ_MAX_PAIRS || (n > 0 && (!a || !b || !bytes))) return GSR_ERR_BAD_ARGUMENT;
  *equal_host = 1;
  EqualArgs q;
  memset(&q, 0, sizeof(q));
  unsigned long long total = 0;
  for (int i = 0; i < n; ++i) {
    if (bytes[i] == 0 || a[i] == b[i]) continue;
    if (!a[i] || !b[i] || (bytes[i] & 3u) || (((uintptr_t)a[i] | (uintptr_t)b[i]) & 3u)) return GSR_ERR_BAD_ARGUMENT;
    q.a[q.n] = (const uint32_t*)a[i];
    q.b[q.n] = (const uint32_t*)b[i];
    q.words[q.n] = bytes[i] / 4;
    total += q.words[q.n];
    q.n++;
  }
  if (q.n == 0) return GSR_OK;
  hipStream_t s = (hipStream_t)stream;
  hipError_t slot_err = hipSuccess;
  HostSlot* slot = host_slot(s, &slot_err);
  if (slot == nullptr) return slot_err == hipSuccess ? GSR_ERR_BAD_ARGUMENT : hip_fail(slot_err);
  const uint32_t seq = ++slot->seq ? slot->seq : ++slot->seq;
  slot->words[GEOM_HDR_DIFFER] = 0;  // (pinned host memory: in place before the launch below is submitted)
  __atomic_thread_fence(__ATOMIC_RELEASE);
  const unsigned long long per_block = 256ull * 16ull;  // words: four 16-byte vectors per thread
  const unsigned blocks = (unsigned)(total / per_block < 1 ? 1 : (total / per_block > 2048 ? 2048 : total / per_block));
  hipLaunchKernelGGL(arrays_equal_kernel, dim3(blocks), dim3(256), 0, s, q, slot->dev_words);
  hipLaunchKernelGGL(publish_seq_kernel, dim3(1), dim3(1), 0, s, slot->dev_words, seq);
  GSR_HIP(hipGetLastError());
  GSR_HIP(wait_for_slot(slot, seq, s));
  *equal_host = slot->words[GEOM_HDR_DIFFER] == 0 ? 1 : 0;
  return GSR_OK;
}

int gsr_bin(void* stream, int P, int64_t R, int64_t G, int W, int H, const void* geom, void* binning, void* image) {
  if (P < 0 || R < 0 || G < 0 || W <= 0 || H <= 0 || !image || misaligned(image)) return GSR_ERR_BAD_ARGUMENT;
  if (R >= (1ll << 31) || G >= (1ll << 31)) return GSR_ERR_TOO_MANY;
  if (R > 0 && (!geom || !binning || misaligned(geom) || misaligned(binning))) return GSR_ERR_BAD_ARGUMENT;
  const int legacy = bin_legacy(W, H);
  if (R > 0 && !legacy && (G <= 0 || G > R)) return GSR_ERR_BAD_ARGUMENT;  // every group instance holds >= 1 tile instance
  const Geom g = carve_geom(const_cast<void*>(geom), P);
  const Binning b = carve_binning(binning, R, G, W, H, legacy);
  const Image im = carve_image(image, W, H);
  GSR_HIP(launch_binning((hipStream_t)stream, P, R, W, H, g, b, im));
  return GSR_OK;
}

int gsr_blend_forward(void* stream, int P, int64_t R, int W, int H, const float* bg, const void* geom,
                      const void* binning, void* image, float* out_color, float* out_depth, unsigned flags) {
  if (P < 0 || R < 0 || W <= 0 || H <= 0 || !bg || !image || !out_color || !out_depth || (flags & ~GSR_FLAG_ALL))
    return GSR_ERR_BAD_ARGUMENT;
  if (R > 0 && (!geom || !binning)) return GSR_ERR_BAD_ARGUMENT;
  const Geom g = carve_geom(const_cast<void*>(geom), P);
  const Binning b = carve_binning_view(binning, R, W, H);
  const Image im = carve_image(image, W, H);
  BlendArgs a = make_blend_args(W, H, g, b, im, bg, 0, R);
  a.out_color = out_color;
  a.out_depth = out_depth;
  a.fast_exp = (flags & GSR_FLAG_FAST_EXP) ? 1 : 0;
  a.shared_simds = (flags & GSR_FLAG_SHARED_SIMDS) ? 1 : 0;
  a.for_backward = (flags & GSR_FLAG_FORWARD_ONLY) ? 0 : 1;
  GSR_HIP(launch_blend_forward((hipStream_t)stream, a));
  return GSR_OK;
}

int gsr_blend_forward_aux(void* stream, int P, int64_t R, int W, int H, const float* bg, const void* geom,
                          const void* binning, void* image, const float* colors, float* out_color, float* out_depth,
                          unsigned flags) {
  if (P < 0 || R < 0 || W <= 0 || H <= 0 || !bg || !image || !out_color || (flags & ~GSR_FLAG_ALL)) return GSR_ERR_BAD_ARGUMENT;
  if (R > 0 && (!geom || !binning || !colors)) return GSR_ERR_BAD_ARGUMENT;
  const Geom g = carve_geom(const_cast<void*>(geom), P);
  const Binning b = carve_binning_view(binning, R, W, H);
  const Image im = carve_image(image, W, H);
  BlendArgs a = make_blend_args(W, H, g, b, im, bg, 0, R);
  a.colors3 = colors;
  a.out_color = out_color;
  a.out_depth = out_depth;
  // leave everything the backward of the MAIN render reads untouched
  a.final_T = nullptr;
  a.n_contrib = nullptr;
  a.work_est = nullptr;
  a.work_maxc = nullptr;
  a.ck_table = nullptr;
  a.fast_exp = (flags & GSR_FLAG_FAST_EXP) ? 1 : 0;
  a.shared_simds = (flags & GSR_FLAG_SHARED_SIMDS) ? 1 : 0;
  GSR_HIP(launch_blend_forward((hipStream_t)stream, a));
  return GSR_OK;
}

int gsr_debug_blend_forward_profile(void* stream, int P, int64_t R, int W, int H, const float* bg, const void* geom,
                                    const void* binning, void* image, float* out_color, float* out_depth,
                                    uint64_t* records, int64_t max_records, int64_t* n_records_host) {
  if (!records || !n_records_host) return GSR_ERR_BAD_ARGUMENT;
  const int64_t n = (int64_t)blend_grid_size(false, (hipStream_t)stream);
  *n_records_host = n;
  if (max_records < n) return GSR_ERR_BAD_ARGUMENT;
  if (P < 0 || R < 0 || W <= 0 || H <= 0 || !bg || !image || !out_color || !out_depth) return GSR_ERR_BAD_ARGUMENT;
  if (R > 0 && (!geom || !binning)) return GSR_ERR_BAD_ARGUMENT;
  const Geom g = carve_geom(const_cast<void*>(geom), P);
  const Binning b = carve_binning_view(binning, R, W, H);
  const Image im = carve_image(image, W, H);
  BlendArgs a = make_blend_args(W, H, g, b, im, bg, 0, R);
  a.out_color = out_color;
  a.out_depth = out_depth;
  a.profile = records;
  GSR_HIP(hipMemsetAsync(records, 0, sizeof(uint64_t) * 8 * (size_t)n, (hipStream_t)stream));
  GSR_HIP(launch_blend_forward((hipStream_t)stream, a));
  return GSR_OK;
}

int gsr_blend_backward(void* stream, int P, int64_t R, int W, int H, const float* bg, const void* geom,
                       const void* binning, const void* image, const float* dL_dpix, float* acc, uint8_t* touched,
                       unsigned flags) {
  // (a backward of a view whose forward was declared forward-only: the flags of one view travel together)
  if ((flags & ~GSR_FLAG_ALL) || (flags & (GSR_FLAG_FORWARD_ONLY | GSR_FLAG_ACC_SELF_CLEAN))) return GSR_ERR_BAD_ARGUMENT;
  if (P == 0) return GSR_OK;
  if (P < 0 || !acc || ((uintptr_t)acc & 63u)) return GSR_ERR_BAD_ARGUMENT;  // (a row must not straddle two 64-byte lines)
  if (touched != nullptr && ((uintptr_t)touched & 15u)) return GSR_ERR_BAD_ARGUMENT;
  if (R == 0) {  // nothing to blend: the accumulator rows stay zero -- or become zero
    if (flags & GSR_FLAG_CLEAR_GRADS) GSR_HIP(hipMemsetAsync(acc, 0, sizeof(float) * ACC_ROW * (size_t)P, (hipStream_t)stream));
    if (touched != nullptr) GSR_HIP(hipMemsetAsync(touched, 0, (size_t)P, (hipStream_t)stream));
    return GSR_OK;
  }
  if (R < 0 || W <= 0 || H <= 0) return GSR_ERR_BAD_ARGUMENT;
  // (the backward's work items carry the tile id in 20 bits, next to the half / segment fields: gsr_blend.hip BWD_ITEM_TILE)
  if ((int64_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE) > (int64_t)GSR_MAX_TILES) return GSR_ERR_BAD_ARGUMENT;
  if (!bg || !geom || !binning || !image || !dL_dpix) return GSR_ERR_BAD_ARGUMENT;
  const Geom g = carve_geom(const_cast<void*>(geom), P);
  const Image im = carve_image(const_cast<void*>(image), W, H);
  const Binning b = carve_binning_view(binning, R, W, H);
  BlendArgs a = make_blend_args(W, H, g, b, im, bg, 1, R);
  a.dL_dpix = dL_dpix;
  a.acc = acc;
  a.touched = touched;
  a.fast_exp = (flags & GSR_FLAG_FAST_EXP) ? 1 : 0;
  a.shared_simds = (flags & GSR_FLAG_SHARED_SIMDS) ? 1 : 0;
  a.P = P;
  a.clear_grads = (flags & GSR_FLAG_CLEAR_GRADS) ? 1 : 0;
  GSR_HIP(launch_blend_backward((hipStream_t)stream, a));
  return GSR_OK;
}

static int preprocess_backward_impl(void* stream, int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                                    const float* scales, float scale_modifier, const float* rotations,
                                    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                    const float* campos, float tan_fovx, float tan_fovy, const int32_t* radii,
                                    const void* geom, const float* acc, float* dL_dmeans2D, float* dL_dopacity,
                                    float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                                    float* dL_drgb, float* dL_dscales, float* dL_drots, uint8_t* row_state = nullptr,
                                    bool self_clean = false) {
  if (P == 0) return GSR_OK;
  if (P < 0 || W <= 0 || H <= 0 || D < 0 || D > 3) return GSR_ERR_BAD_ARGUMENT;
  if (!means3D || !viewmatrix || !projmatrix || !radii || !geom) return GSR_ERR_BAD_ARGUMENT;
  // this geometry state was left by gsr_preprocess(GSR_FLAG_FORWARD_ONLY): what K8+K9 reads of it was never written
  if (shs && geom_is_forward_only(geom)) return GSR_ERR_BAD_ARGUMENT;
  if (!acc || ((uintptr_t)acc & 63u) || !dL_dmeans2D || !dL_dopacity || !dL_dmeans3D) return GSR_ERR_BAD_ARGUMENT;
  if (cov3D_precomp && !dL_dcov3D) return GSR_ERR_BAD_ARGUMENT;  // (without a precomputed covariance its gradient is optional)
  if (shs && ((!dL_dsh && !dL_drgb) || !campos)) return GSR_ERR_BAD_ARGUMENT;
  if (scales && (!rotations || !dL_dscales || !dL_drots)) return GSR_ERR_BAD_ARGUMENT;
  if (!cov3D_precomp && !scales) return GSR_ERR_BAD_ARGUMENT;  // the 3D covariance is recomputed, not read from `geom`
  const Geom g = carve_geom(const_cast<void*>(geom), P);
  PreBwdArgs pa;
  pa.P = P; pa.D = D; pa.M = shs ? M : 0;
  pa.means3D = means3D; pa.radii = radii; pa.shs = shs; pa.scales = scales; pa.rotations = rotations;
  pa.scale_modifier = scale_modifier;
  pa.cov3D_precomp = cov3D_precomp;
  pa.clamped = g.clamped;
  for (int i = 0; i < 3; ++i) pa.dcol[i] = g.dcol[i];
  pa.viewmatrix = viewmatrix; pa.projmatrix = projmatrix; pa.campos = campos;
  pa.h_y = H / (2.0f * tan_fovy);  // rasterizer_impl.cu:308-309
  pa.h_x = W / (2.0f * tan_fovx);
  pa.tan_fovx = tan_fovx; pa.tan_fovy = tan_fovy;
  pa.acc = acc; pa.dL_dmean2D = dL_dmeans2D; pa.dL_dopacity = dL_dopacity; pa.dL_dcolor = dL_dcolors;
  pa.dL_dmeans3D = dL_dmeans3D; pa.dL_dcov3D = dL_dcov3D;
  pa.dL_dsh = shs ? dL_dsh : nullptr;
  pa.dL_drgb = shs ? dL_drgb : nullptr;
  pa.dL_dscale = scales ? dL_dscales : nullptr;
  pa.dL_drot = scales ? dL_drots : nullptr;
  pa.row_state = row_state;
  pa.acc_clean = self_clean ? const_cast<float*>(acc) : nullptr;  // (GSR_FLAG_ACC_SELF_CLEAN: the caller's table, writable by contract)
  GSR_HIP(launch_preprocess_backward((hipStream_t)stream, pa));
  return GSR_OK;
}

int gsr_debug_blend_backward_profile(void* stream, int P, int64_t R, int W, int H, const float* bg, const void* geom,
                                     const void* binning, const void* image, const float* dL_dpix, float* acc,
                                     uint64_t* records, int64_t max_records, int64_t* n_records_host) {
  if (!records || !n_records_host) return GSR_ERR_BAD_ARGUMENT;
  const int64_t n = (int64_t)blend_grid_size(true, (hipStream_t)stream) / 4;
  *n_records_host = n;
  if (max_records < n) return GSR_ERR_BAD_ARGUMENT;
  if (P < 0 || R <= 0 || W <= 0 || H <= 0 || !bg || !geom || !binning || !image || !dL_dpix) return GSR_ERR_BAD_ARGUMENT;
  if (!acc || ((uintptr_t)acc & 63u)) return GSR_ERR_BAD_ARGUMENT;
  const Geom g = carve_geom(const_cast<void*>(geom), P);
  const Binning b = carve_binning_view(binning, R, W, H);
  const Image im = carve_image(const_cast<void*>(image), W, H);
  BlendArgs a = make_blend_args(W, H, g, b, im, bg, 1, R);
  a.dL_dpix = dL_dpix;
  a.acc = acc;
  a.profile = records;
  // optional extension: with room for T more records, 4 x u64 per (tile, half) follow the workgroup records:
  // {cycles, the forward's four per-quadrant counts (16 bits each), positions the backward walked, item code}
  const int64_t T = (int64_t)a.gx * a.gy;
  const bool items = max_records >= n + T;
  a.profile_items = items ? records + 8 * n : nullptr;
  GSR_HIP(hipMemsetAsync(records, 0, sizeof(uint64_t) * 8 * (size_t)(items ? n + T : n), (hipStream_t)stream));
  GSR_HIP(launch_blend_backward((hipStream_t)stream, a));
  return GSR_OK;
}

int gsr_preprocess_backward(void* stream, int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                            const float* scales, float scale_modifier, const float* rotations,
                            const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                            const float* campos, float tan_fovx, float tan_fovy, const int32_t* radii,
                            const void* geom, float* acc, float* dL_dmeans2D, float* dL_dopacity,
                            float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                            float* dL_dscales, float* dL_drots, unsigned flags) {
  if (shs && !dL_dsh) return GSR_ERR_BAD_ARGUMENT;
  if (flags & ~GSR_FLAG_ACC_SELF_CLEAN) return GSR_ERR_BAD_ARGUMENT;  // (the one flag this half reads)
  return preprocess_backward_impl(stream, P, D, M, W, H, means3D, shs, scales, scale_modifier, rotations, cov3D_precomp,
                                  viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom, acc, dL_dmeans2D, dL_dopacity,
                                  dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dsh, nullptr, dL_dscales, dL_drots, nullptr,
                                  (flags & GSR_FLAG_ACC_SELF_CLEAN) != 0);
}

int gsr_preprocess_backward_rgb(void* stream, int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                                const float* scales, float scale_modifier, const float* rotations,
                                const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                const float* campos, float tan_fovx, float tan_fovy, const int32_t* radii,
                                const void* geom, float* acc, float* dL_dmeans2D, float* dL_dopacity,
                                float* dL_dmeans3D, float* dL_dcov3D, float* dL_drgb,
                                float* dL_dscales, float* dL_drots, unsigned flags) {
  if (!shs || !dL_drgb) return GSR_ERR_BAD_ARGUMENT;
  if (flags & ~GSR_FLAG_ACC_SELF_CLEAN) return GSR_ERR_BAD_ARGUMENT;
  return preprocess_backward_impl(stream, P, D, M, W, H, means3D, shs, scales, scale_modifier, rotations, cov3D_precomp,
                                  viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom, acc, dL_dmeans2D, dL_dopacity,
                                  nullptr, dL_dmeans3D, dL_dcov3D, nullptr, dL_drgb, dL_dscales, dL_drots, nullptr,
                                  (flags & GSR_FLAG_ACC_SELF_CLEAN) != 0);
}

int gsr_preprocess_backward_rows(void* stream, int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                                 const float* scales, float scale_modifier, const float* rotations,
                                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                 const float* campos, float tan_fovx, float tan_fovy, const int32_t* radii,
                                 const void* geom, const float* acc, float* dL_dmeans2D, float* dL_dopacity,
                                 float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                                 float* dL_drgb, float* dL_dscales, float* dL_drots, uint8_t* row_state) {
  if (!row_state || (dL_dsh && dL_drgb)) return GSR_ERR_BAD_ARGUMENT;
  if (shs && !dL_dsh && !dL_drgb) return GSR_ERR_BAD_ARGUMENT;
  return preprocess_backward_impl(stream, P, D, M, W, H, means3D, shs, scales, scale_modifier, rotations, cov3D_precomp,
                                  viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom, acc, dL_dmeans2D, dL_dopacity,
                                  dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_drgb, dL_dscales, dL_drots, row_state);
}

int gsr_sh_grad_compose(void* stream, int P, int D, int M, int num_views, const float* means3D, const float* campos,
                        const float* dL_drgb, float* dL_dsh) {
  if (P == 0) return GSR_OK;
  if (P < 0 || D < 0 || D > 3 || M < (D + 1) * (D + 1) || num_views < 1) return GSR_ERR_BAD_ARGUMENT;
  if (!means3D || !campos || !dL_drgb || !dL_dsh) return GSR_ERR_BAD_ARGUMENT;
  GSR_HIP(launch_sh_grad_compose((hipStream_t)stream, P, D, M, num_views, means3D, campos, dL_drgb, dL_dsh));
  return GSR_OK;
}

int gsr_view_message_words(int64_t P, int64_t cap, int64_t* words) {
  if (P < 0 || cap < 0 || !words) return GSR_ERR_BAD_ARGUMENT;
  *words = view_message_words_host(P, cap);
  return GSR_OK;
}

static bool dense_grads_complete(const gsr_dense_grads* g) {
  return g && g->means3D && g->scales && g->rotations && g->means2D && g->opacities;
}

int gsr_view_message_plan(void* stream, int64_t P, const gsr_dense_grads* local, const float* rgb, uint8_t* mask,
                          void* workspace, int64_t* count_host) {
  if (count_host) *count_host = 0;
  if (P == 0) return GSR_OK;
  if (P < 0 || !dense_grads_complete(local) || !rgb || !mask || !workspace) return GSR_ERR_BAD_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  const float* data[6] = {local->means3D, local->scales, local->rotations, local->means2D, local->opacities, rgb};
  const int row_len[6] = {3, 3, 4, 3, 1, 3};
  GSR_HIP(launch_touched_rows(s, P, 6, data, row_len, mask));
  GSR_HIP(launch_compact_plan(s, P, mask, workspace));
  if (!count_host) return GSR_OK;  // the caller reads the count from the workspace itself (first 8 bytes)
  uint64_t total = 0;
  GSR_HIP(hipMemcpyAsync(&total, compact_total_ptr(workspace, P), sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  GSR_HIP(hipStreamSynchronize(s));
  *count_host = (int64_t)total;
  return GSR_OK;
}

int gsr_view_message_plan_blend(void* stream, int64_t P, const uint8_t* touched, void* workspace) {
  if (P == 0) return GSR_OK;
  if (P < 0 || !touched || !workspace) return GSR_ERR_BAD_ARGUMENT;
  GSR_HIP(launch_compact_plan((hipStream_t)stream, P, touched, workspace));
  return GSR_OK;
}

int gsr_view_message_pack(void* stream, int64_t P, const gsr_dense_grads* local, const float* rgb, const float* campos,
                          const uint8_t* mask, void* workspace, int64_t cap, float* message) {
  if (P < 0 || cap < 0 || !message || !campos) return GSR_ERR_BAD_ARGUMENT;
  if (P > 0 && (!dense_grads_complete(local) || !rgb || !mask || !workspace)) return GSR_ERR_BAD_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  if (P == 0) {  // header only: camera centre, count 0
    GSR_HIP(hipMemcpyAsync(message, campos, 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    GSR_HIP(hipMemsetAsync(message + 3, 0, sizeof(float), s));
    return GSR_OK;
  }
  const int64_t nb = (P + VIEW_MSG_ROWS - 1) / VIEW_MSG_ROWS;
  float* rows = message + 4 + nb;
  const gsr_compact_tensor t[7] = {{nullptr, rows, 4},  // the row numbers themselves
                                   {local->means3D, rows + cap, 12},
                                   {local->scales, rows + 4 * cap, 12},
                                   {local->rotations, rows + 7 * cap, 16},
                                   {local->means2D, rows + 11 * cap, 12},
                                   {local->opacities, rows + 14 * cap, 4},
                                   {rgb, rows + 15 * cap, 12}};
  // (a message speculatively sized below this view's row count -- multiview.py -- keeps its first `cap` rows: its header
  //  still carries the true count, the receivers' host code sees the overflow and has the message sent again)
  GSR_HIP(launch_compact_apply(s, P, mask, workspace, 7, t, cap));
  GSR_HIP(launch_view_message_header(s, P, campos, compact_block_off_ptr(workspace, P), compact_total_ptr(workspace, P), message));
  return GSR_OK;
}

int gsr_view_messages_accumulate(void* stream, int64_t P, int D, int M, int num_views, const float* messages,
                                 int64_t stride_words, int64_t cap, const float* means3D, const gsr_dense_grads* out) {
  return gsr_view_messages_accumulate_rows(stream, P, D, M, num_views, messages, stride_words, cap, means3D, out, nullptr);
}

int gsr_view_messages_accumulate_rows(void* stream, int64_t P, int D, int M, int num_views, const float* messages,
                                      int64_t stride_words, int64_t cap, const float* means3D, const gsr_dense_grads* out,
                                      uint8_t* row_valid) {
  if (P == 0) return GSR_OK;
  if (P < 0 || num_views < 1 || !messages || cap < 0 || stride_words < view_message_words_host(P, cap)) return GSR_ERR_BAD_ARGUMENT;
  if (!dense_grads_complete(out)) return GSR_ERR_BAD_ARGUMENT;
  if (out->sh && (!means3D || D < 0 || D > 3 || M < (D + 1) * (D + 1))) return GSR_ERR_BAD_ARGUMENT;
  float* const d[6] = {out->means3D, out->scales, out->rotations, out->means2D, out->opacities, out->sh};
  GSR_HIP(launch_view_messages_accumulate((hipStream_t)stream, P, D, M, num_views, messages, stride_words, cap, means3D, d, row_valid));
  return GSR_OK;
}

int gsr_backward(void* stream, int P, int D, int M, int64_t R, int W, int H, const float* bg, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* campos, float tan_fovx, float tan_fovy, const int32_t* radii, const void* geom,
                 const void* binning, const void* image, const float* dL_dpix, float* acc, float* dL_dmeans2D,
                 float* dL_dopacity, float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscales, float* dL_drots, unsigned flags) {
  (void)colors_precomp;  // the blend kernels read the colour copy held in the geometry records
  if (P == 0) return GSR_OK;
  if (R > 0 && !binning) return GSR_ERR_BAD_ARGUMENT;
  const bool self_clean = (flags & GSR_FLAG_ACC_SELF_CLEAN) != 0;
  if (self_clean && (flags & GSR_FLAG_CLEAR_GRADS)) return GSR_ERR_BAD_ARGUMENT;
  int st = gsr_blend_backward(stream, P, R, W, H, bg, geom, binning, image, dL_dpix, acc, nullptr, flags & ~GSR_FLAG_ACC_SELF_CLEAN);
  if (st != GSR_OK) return st;
  return gsr_preprocess_backward(stream, P, D, M, W, H, means3D, shs, scales, scale_modifier, rotations, cov3D_precomp,
                                 viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom, acc, dL_dmeans2D, dL_dopacity,
                                 dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drots,
                                 self_clean ? GSR_FLAG_ACC_SELF_CLEAN : 0u);
}

int gsr_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present) {
  (void)projmatrix;  // computed but unused by the reference as well (auxiliary.h:149-154)
  if (P == 0) return GSR_OK;
  if (P < 0 || !means3D || !viewmatrix || !present) return GSR_ERR_BAD_ARGUMENT;
  GSR_HIP(launch_mark_visible((hipStream_t)stream, P, means3D, viewmatrix, present));
  return GSR_OK;
}

int gsr_trace_weights(void* stream, int P, int64_t R, int W, int H, int C, const void* geom, const void* binning,
                      const void* image, const float* image_weights, float* weights, int32_t* cnt, unsigned flags) {
  if (C < 1 || C > 3) return GSR_ERR_BAD_CHANNELS;
  if (flags & ~GSR_FLAG_ALL) return GSR_ERR_BAD_ARGUMENT;
  if (P < 0 || R < 0 || W <= 0 || H <= 0 || !image || !image_weights || !weights || !cnt) return GSR_ERR_BAD_ARGUMENT;
  if (R == 0) return GSR_OK;
  if (!geom || !binning) return GSR_ERR_BAD_ARGUMENT;
  const Geom g = carve_geom(const_cast<void*>(geom), P);
  const Binning b = carve_binning_view(binning, R, W, H);
  const Image im = carve_image(const_cast<void*>(image), W, H);
  BlendArgs a = make_blend_args(W, H, g, b, im, nullptr, 2, R);
  a.C = C;
  a.image_weights = image_weights;
  a.weights = weights;
  a.cnt = cnt;
  a.fast_exp = (flags & GSR_FLAG_FAST_EXP) ? 1 : 0;
  a.shared_simds = (flags & GSR_FLAG_SHARED_SIMDS) ? 1 : 0;
  GSR_HIP(launch_trace_weights((hipStream_t)stream, a));
  return GSR_OK;
}

int gsr_knn_workspace_size(int P, size_t* bytes) {
  if (P < 0 || !bytes) return GSR_ERR_BAD_ARGUMENT;
  *bytes = knn_workspace_bytes(P);
  return GSR_OK;
}

int gsr_knn_mean_dist2(void* stream, int P, const float* points, void* workspace, float* mean_dist2) {
  if (P == 0) return GSR_OK;
  if (P < 0 || !points || !workspace || !mean_dist2) return GSR_ERR_BAD_ARGUMENT;
  GSR_HIP(launch_knn((hipStream_t)stream, P, points, workspace, mean_dist2));
  return GSR_OK;
}

int gsr_near_workspace_size(int n_ref, size_t* bytes) { return gsr_knn_workspace_size(n_ref, bytes); }

int gsr_near_points(void* stream, int n_ref, const float* ref_points, int n_query, const float* query_points,
                    float dist_thresh, void* workspace, uint8_t* near, float* nn_dist) {
  if (n_query == 0) return GSR_OK;
  if (n_ref < 0 || n_query < 0 || !query_points || !near || !(dist_thresh >= 0.f)) return GSR_ERR_BAD_ARGUMENT;
  if (n_ref == 0) {  // nothing to be near to
    GSR_HIP(hipMemsetAsync(near, 0, (size_t)n_query, (hipStream_t)stream));
    if (nn_dist) GSR_HIP(hipMemsetD32Async((hipDeviceptr_t)nn_dist, 0x7f800000, (size_t)n_query, (hipStream_t)stream));
    return GSR_OK;
  }
  if (!ref_points || !workspace) return GSR_ERR_BAD_ARGUMENT;
  GSR_HIP(launch_near_points((hipStream_t)stream, n_ref, ref_points, n_query, query_points, dist_thresh, workspace, near, nn_dist));
  return GSR_OK;
}

int gsr_adam_step(void* stream, int num_tensors, const gsr_adam_tensor* tensors, int64_t step, double beta1, double beta2,
                  double eps, const uint8_t* row_mask, const float* row_weight) {
  return gsr_adam_step_rows(stream, num_tensors, tensors, step, beta1, beta2, eps, row_mask, row_weight, nullptr);
}

int gsr_adam_step_rows(void* stream, int num_tensors, const gsr_adam_tensor* tensors, int64_t step, double beta1, double beta2,
                       double eps, const uint8_t* row_mask, const float* row_weight, const uint8_t* grad_valid) {
  if (num_tensors == 0) return GSR_OK;
  if (num_tensors < 0 || num_tensors > 8 || !tensors || step < 1) return GSR_ERR_BAD_ARGUMENT;
  if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return GSR_ERR_BAD_ARGUMENT;
  for (int i = 0; i < num_tensors; ++i) {
    const gsr_adam_tensor& t = tensors[i];
    if (t.numel < 0 || t.row_len < 1) return GSR_ERR_BAD_ARGUMENT;
    if (t.numel > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq)) return GSR_ERR_BAD_ARGUMENT;
  }
  GSR_HIP(launch_adam_step((hipStream_t)stream, num_tensors, tensors, (long long)step, beta1, beta2, eps, row_mask, row_weight,
                           grad_valid));
  return GSR_OK;
}

int gsr_compact_workspace_size(int64_t P, size_t* bytes) {
  if (P < 0 || !bytes) return GSR_ERR_BAD_ARGUMENT;
  *bytes = compact_workspace_bytes(P);
  return GSR_OK;
}

int gsr_compact_plan(void* stream, int64_t P, const uint8_t* keep, void* workspace, int64_t* kept_host) {
  if (!kept_host) return GSR_ERR_BAD_ARGUMENT;
  *kept_host = 0;
  if (P == 0) return GSR_OK;
  if (P < 0 || P >= (1ll << 32) - 1024 || !keep || !workspace) return GSR_ERR_BAD_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  GSR_HIP(launch_compact_plan(s, P, keep, workspace));
  uint64_t total = 0;
  GSR_HIP(hipMemcpyAsync(&total, compact_total_ptr(workspace, P), sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  GSR_HIP(hipStreamSynchronize(s));
  *kept_host = (int64_t)total;
  return GSR_OK;
}

int gsr_compact_apply(void* stream, int64_t P, const uint8_t* keep, void* workspace, int num_tensors,
                      const gsr_compact_tensor* tensors) {
  if (P == 0 || num_tensors == 0) return GSR_OK;
  if (P < 0 || num_tensors < 0 || num_tensors > 32 || !keep || !workspace || !tensors) return GSR_ERR_BAD_ARGUMENT;
  for (int i = 0; i < num_tensors; ++i)
    if (!tensors[i].src || tensors[i].row_bytes < 1 || tensors[i].row_bytes > (1 << 20)) return GSR_ERR_BAD_ARGUMENT;
  // dst may be NULL only if nothing survives; the kernel then writes nothing
  GSR_HIP(launch_compact_apply((hipStream_t)stream, P, keep, workspace, num_tensors, tensors));
  return GSR_OK;
}

int gsr_append_rows(void* stream, int64_t P, int64_t n, int num_tensors, const gsr_append_tensor* tensors) {
  if (P < 0 || n < 0 || num_tensors < 0 || num_tensors > 32) return GSR_ERR_BAD_ARGUMENT;
  if (num_tensors == 0 || P + n == 0) return GSR_OK;
  if (!tensors) return GSR_ERR_BAD_ARGUMENT;
  for (int i = 0; i < num_tensors; ++i) {
    const gsr_append_tensor& t = tensors[i];
    if (t.row_bytes < 1 || t.row_bytes > (1 << 20) || !t.dst || (P > 0 && !t.src)) return GSR_ERR_BAD_ARGUMENT;
  }
  GSR_HIP(launch_append_rows((hipStream_t)stream, P, n, num_tensors, tensors));
  return GSR_OK;
}

int gsr_debug_cov3d(void* stream, int P, const float* scales, float scale_modifier, const float* rotations, float* cov3D) {
  if (P <= 0) return GSR_OK;
  if (!scales || !rotations || !cov3D) return GSR_ERR_BAD_ARGUMENT;
  GSR_HIP(launch_export_cov3d((hipStream_t)stream, P, scales, scale_modifier, rotations, cov3D));
  return GSR_OK;
}

int gsr_debug_export_geom(void* stream, int P, const void* geom, float* means2D, float* depths, float* rgb,
                          float* conic_opacity, uint32_t* tiles_touched, uint8_t* clamped) {
  if (P <= 0) return GSR_OK;
  if (!geom) return GSR_ERR_BAD_ARGUMENT;
  const Geom g = carve_geom(const_cast<void*>(geom), P);
  GSR_HIP(launch_export_geom((hipStream_t)stream, P, g, means2D, depths, rgb, conic_opacity, tiles_touched, clamped));
  return GSR_OK;
}

int gsr_debug_export_binning(void* stream, int P, int64_t R, int W, int H, const void* geom, const void* binning,
                             uint64_t* keys, uint32_t* point_list) {
  if (R <= 0) return GSR_OK;
  if (!binning || !geom) return GSR_ERR_BAD_ARGUMENT;
  const Geom g = carve_geom(const_cast<void*>(geom), P);
  const Binning b = carve_binning_view(binning, R, W, H);
  hipStream_t s = (hipStream_t)stream;
  if (keys) GSR_HIP(launch_export_keys(s, R, W, H, b, g, keys));
  if (point_list)
    GSR_HIP(hipMemcpyAsync(point_list, b.point_list, sizeof(uint32_t) * (size_t)R, hipMemcpyDeviceToDevice, s));
  return GSR_OK;
}

int gsr_debug_export_image(void* stream, int W, int H, const void* image, uint32_t* ranges, float* final_T,
                           uint32_t* n_contrib) {
  if (!image || W <= 0 || H <= 0) return GSR_ERR_BAD_ARGUMENT;
  const Image im = carve_image(const_cast<void*>(image), W, H);
  hipStream_t s = (hipStream_t)stream;
  const size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE), N = (size_t)W * H;
  if (ranges) GSR_HIP(hipMemcpyAsync(ranges, im.ranges, sizeof(uint2) * T, hipMemcpyDeviceToDevice, s));
  if (final_T) GSR_HIP(hipMemcpyAsync(final_T, im.final_T, sizeof(float) * N, hipMemcpyDeviceToDevice, s));
  if (n_contrib) GSR_HIP(hipMemcpyAsync(n_contrib, im.n_contrib, sizeof(uint32_t) * N, hipMemcpyDeviceToDevice, s));
  return GSR_OK;
}

}  // extern "C"
