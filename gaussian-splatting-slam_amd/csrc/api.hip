// api.hip - extern "C" entry points of libgsr_hip.so (declared in include/gsr.h) and host orchestration.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <atomic>

#include "gsr_common.h"

// launchers defined in the kernel files
void gsr_launch_preprocess_fwd(const gsr_settings*, const gsr_gaussians*, int32_t*, char*, const GsrGeomLayout&, bool, bool, uint32_t*, int,
                               const uint32_t*, uint32_t*, uint32_t, hipStream_t);
void gsr_launch_shade(const gsr_settings*, const gsr_gaussians*, char*, const GsrGeomLayout&, bool, hipStream_t);
void gsr_launch_adam_culled_rows(int, int, const char*, const GsrGeomLayout&, const GsrAdamArgs&, uint32_t, hipStream_t);
int gsr_launch_preprocess_bwd(const gsr_settings*, const gsr_gaussians*, const int32_t*, const char*,
                              const GsrGeomLayout&, const float4*, uint32_t, const gsr_grads*, const GsrAdamArgs*, int,
                              hipStream_t);
void gsr_launch_mark_visible(int, const float*, const float*, uint8_t*, hipStream_t);
void gsr_launch_emit(int, int, int, char*, const GsrGeomLayout&, char*, const GsrBinLayout&, uint32_t, bool, unsigned long long*,
                     int, const uint32_t*, hipStream_t);
void gsr_launch_tile_depth_sort(int, bool, uint2*, const uint2*, uint32_t*, uint32_t*, const uint32_t*, uint32_t*, uint32_t*,
                                uint32_t*, uint32_t*, uint32_t*, hipStream_t);
void gsr_launch_finalize(uint32_t, const uint32_t*, const uint32_t*, char*, const GsrBinLayout&, hipStream_t);
void gsr_launch_sum_tiles(int, const char*, const GsrGeomLayout&, uint32_t*, hipStream_t);
void gsr_launch_render_fwd(const gsr_settings*, int, int, const uint2*, const uint32_t*, const float4*, float*,
                           float*, float*, uint32_t*, const uint32_t*, uint32_t*, uint32_t*, const uint32_t*, const uint32_t*,
                           uint32_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*, hipStream_t);
void gsr_launch_count_pairs(const gsr_settings*, int, int, const uint2*, const uint32_t*, const float4*, uint32_t*,
                            hipStream_t);
void gsr_launch_render_bwd(const gsr_settings*, int, int, const uint2*, const uint32_t*, const float4*,
                           const float*, const uint32_t*, const float*, const float*, const uint32_t*, float4*,
                           const uint32_t*, uint32_t, const uint32_t*, const uint32_t*, const uint32_t*, hipStream_t);

// ---------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------
static thread_local char t_err[1024] = "";

void gsr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_err, sizeof(t_err), fmt, ap);
  va_end(ap);
}

int gsr_check(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  gsr_set_error("%s: %s", what, hipGetErrorString(e));
  return GSR_ERR_HIP;
}

static thread_local const char* t_fail_stage = nullptr;
static thread_local hipError_t t_fail_err = hipSuccess;

void gsr_note_launch_failure(const char* stage, hipError_t e) {
  if (!t_fail_stage) { t_fail_stage = stage; t_fail_err = e; }
}

// status of the launches since the last call on this host thread: the first stage whose launch was rejected, if any
int gsr_launch_status(const char* what) {
  if (t_fail_stage) {
    gsr_set_error("%s: launch of stage '%s' failed: %s", what, t_fail_stage, hipGetErrorString(t_fail_err));
    t_fail_stage = nullptr;
    t_fail_err = hipSuccess;
    return GSR_ERR_HIP;
  }
  return gsr_check(hipGetLastError(), what);
}

// status0: the frame's first status word on the device (meta[0]); with debug = 1 a stage that left GSR_STATUS_SORT_TIMEOUT there
// fails the call (reference README.md:168-171: with --debug a failing rasterizer call raises and dumps its inputs).
static int debug_sync(const gsr_settings* s, hipStream_t st, const char* stage, const uint32_t* status0 = nullptr) {
  static const bool trace = getenv("GSR_TRACE") != nullptr;   // GSR_TRACE=1: name every stage on stderr as it completes
  if (trace) {
    fprintf(stderr, "[gsr] launched: %s\n", stage);
    fflush(stderr);
    hipError_t e = hipStreamSynchronize(st);
    fprintf(stderr, "[gsr] done    : %s (%s)\n", stage, hipGetErrorString(e));
    fflush(stderr);
  }
  if (!s->debug) return 0;
  hipError_t e = hipStreamSynchronize(st);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) {
    gsr_set_error("debug: failure after stage '%s': %s", stage, hipGetErrorString(e));
    return GSR_ERR_HIP;
  }
  if (status0) {
    uint32_t w = 0;
    if (hipMemcpy(&w, status0, 4, hipMemcpyDeviceToHost) == hipSuccess && (w & GSR_STATUS_SORT_TIMEOUT)) {
      gsr_set_error("debug: stage '%s': a radix-sort look-back wait timed out (inter-workgroup hand-off broken): the frame is "
                    "mis-sorted", stage);
      return GSR_ERR_HIP;
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// per-kernel profiling (HIP events on the launch stream)
// ---------------------------------------------------------------------------------------------------
int g_gsr_profile_on = 0;
// frames with at least this many instances flag their gradient records (gsr_common.h); GSR_FLAGS_MIN_R in the environment of the
// process sets the starting value (0 = always: how the seeded sweeps run their small scenes through the flagged form)
static unsigned flags_min_r_from_env() {
  const char* e = getenv("GSR_FLAGS_MIN_R");
  return e && *e ? (unsigned)strtoul(e, nullptr, 10) : GSR_FLAGS_MIN_R;
}
unsigned g_gsr_flags_min_r = flags_min_r_from_env();
namespace {
struct Pending { const char* name; hipEvent_t a, b; };
std::mutex g_prof_mu;
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_free_events;
std::map<std::string, std::pair<double, int64_t>> g_totals;
std::vector<std::string> g_names_keepalive;
thread_local Pending t_open = {nullptr, nullptr, nullptr};

hipEvent_t get_event() {
  if (!g_free_events.empty()) {
    hipEvent_t e = g_free_events.back();
    g_free_events.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

void collect_locked() {
  for (auto& p : g_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      auto& t = g_totals[p.name];
      t.first += ms;
      t.second += 1;
    }
    g_free_events.push_back(p.a);
    g_free_events.push_back(p.b);
  }
  g_pending.clear();
}
}  // namespace

void gsr_prof_begin(const char* name, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  t_open.name = name;
  t_open.a = get_event();
  t_open.b = get_event();
  (void)hipEventRecord(t_open.a, st);
}

void gsr_prof_end(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(t_open.b, st);
  g_pending.push_back(t_open);
  if (g_pending.size() > 4096) collect_locked();
}

// ---------------------------------------------------------------------------------------------------
// Per host thread AND per device: the pinned read-back slot, its event, and the side stream of the colour pass.  Created on first
// use with that device current (one process per GPU is the design, but one thread driving two devices gets two sets: a stream or
// an event made on device 0 is never handed to a launch on device 1).
// ---------------------------------------------------------------------------------------------------
struct SideShade { hipStream_t stream; hipEvent_t fork, join; bool ok, tried; };
struct DeviceLocal {
  uint32_t* pinned = nullptr;        // 64 B of pinned host memory: [0..3] the blocking path's copy of meta[0..3]; [8..9] the early count
  uint32_t* pinned_dev = nullptr;    // the same allocation as the device sees it
  hipEvent_t readback = nullptr;
  SideShade shade = {nullptr, nullptr, nullptr, false, false};
};
static DeviceLocal* device_local() {
  static thread_local std::map<int, DeviceLocal> per_device;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  return &per_device[dev];
}

static uint32_t* pinned_slot() {
  DeviceLocal* d = device_local();
  if (!d) return nullptr;
  if (!d->pinned) {
    if (hipHostMalloc((void**)&d->pinned, 64, hipHostMallocDefault) != hipSuccess) { d->pinned = nullptr; return nullptr; }
    memset(d->pinned, 0, 64);
    void* alias = nullptr;
    if (hipHostGetDevicePointer(&alias, d->pinned, 0) == hipSuccess) d->pinned_dev = (uint32_t*)alias;
    else (void)hipGetLastError();
  }
  return d->pinned;
}

static hipEvent_t readback_event() {
  DeviceLocal* d = device_local();
  if (!d) return nullptr;
  if (!d->readback && hipEventCreateWithFlags(&d->readback, hipEventDisableTiming) != hipSuccess) d->readback = nullptr;
  return d->readback;
}

static int tile_bits(int tiles) {
  int b = 1;
  while ((1 << b) < tiles) b++;
  return b;
}
// which ping-pong buffer holds the tile-sorted (key, slot) arrays: one swap per 8-bit pass
static int tile_sort_result_buffer(int tiles) { return gsr_radix_passes(tile_bits(tiles)) & 1; }
// the Gaussian-id list rides through the tile sort as a second payload, ping-ponging gauss_of_slot <-> point_list
static size_t point_list_offset(const GsrBinLayout& BL, int tiles) {
  return tile_sort_result_buffer(tiles) ? BL.point_list : BL.gauss_of_slot;
}

static int validate(const gsr_settings* s, const gsr_gaussians* g) {
  if (!s || !g) { gsr_set_error("null settings/gaussians"); return GSR_ERR_INVALID_ARGUMENT; }
  if (g->P < 0 || s->image_width <= 0 || s->image_height <= 0) {
    gsr_set_error("bad sizes P=%d W=%d H=%d", g->P, s->image_width, s->image_height);
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (g->P == 0) return 0;  // nothing to rasterize: background only
  if ((g->shs == nullptr && g->dc == nullptr) == (g->colors_precomp == nullptr)) {
    gsr_set_error("Please provide excatly one of either SHs or precomputed colors!");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  const bool sr = g->scales != nullptr || g->rotations != nullptr;
  if ((sr && g->cov3D_precomp) || (!sr && !g->cov3D_precomp) || (sr && (!g->scales || !g->rotations))) {
    gsr_set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (!g->colors_precomp) {
    if (s->sh_degree < 0 || s->sh_degree > 3) {
      gsr_set_error("sh_degree %d unsupported (0..3)", s->sh_degree);
      return GSR_ERR_INVALID_ARGUMENT;
    }
    const int need = (s->sh_degree + 1) * (s->sh_degree + 1);
    const int have = g->sh_coeffs + (g->dc ? 1 : 0);
    if (have < need || (need > 1 && !g->shs)) {
      gsr_set_error("sh_degree %d needs %d coefficients, %d stored", s->sh_degree, need, have);
      return GSR_ERR_INVALID_ARGUMENT;
    }
  }
  // tile ids are sort keys and 16-bit rect coordinates: at most 65535 tiles per side and 2^24 tiles in all
  const long long gx = ((long long)s->image_width + GSR_TILE - 1) / GSR_TILE, gy = ((long long)s->image_height + GSR_TILE - 1) / GSR_TILE;
  if (gx > 65535 || gy > 65535 || gx * gy > (1ll << 24)) {
    gsr_set_error("image too large: %lld x %lld tiles (limit 65535 per side, 2^24 in all)", gx, gy);
    return GSR_ERR_INVALID_ARGUMENT;
  }
  return 0;
}

extern "C" {

int gsr_abi_version(void) { return GSR_ABI_VERSION; }
const char* gsr_last_error(void) { return t_err; }

size_t gsr_geometry_state_bytes(int32_t P) { return gsr_geom_layout((size_t)(P < 0 ? 0 : P)).total; }
size_t gsr_image_state_bytes(int32_t W, int32_t H) { return gsr_img_layout(W, H).total; }
size_t gsr_binning_state_bytes(int32_t P, int32_t W, int32_t H, int64_t R) {
  (void)P;
  const size_t tiles = (size_t)((W + GSR_TILE - 1) / GSR_TILE) * (size_t)((H + GSR_TILE - 1) / GSR_TILE);
  return gsr_bin_layout((size_t)(R < 0 ? 0 : R), tiles).total;
}
size_t gsr_backward_scratch_bytes(int32_t P, int64_t R) {
  (void)P;
  const size_t cap = (size_t)(R < 1 ? 1 : R);
  return gsr_igrad_bytes(cap) + gsr_align(cap + 16); // records + one validity byte per emission slot (gsr_common.h; + the reader's over-read)
}

// Colour pass (SH -> RGB, the HBM-heavy half of the projection) on a library-owned side stream, concurrent with the depth sort /
// scan / emission / tile sort, which are latency-bound and leave most of the machine idle.  Used by gsr_forward_async from
// 200 k Gaussians up (below that the extra launch and two event operations cost the host more than the overlap saves);
// GSR_SHADE_STREAM=0 keeps everything on the caller's stream, =1 forces the side stream at any size.  Same results.
static SideShade* side_shade() {
  DeviceLocal* d = device_local();
  if (!d) return nullptr;
  SideShade& ss = d->shade;
  if (!ss.tried) {
    ss.tried = true;
    ss.ok = hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess;
  }
  return ss.ok ? &ss : nullptr;
}

// Device-side address of a pinned host allocation (hipHostMalloc / torch's pin_memory), nullptr for anything else.  Queried on
// every call (~1 us): a cached answer would go stale if the caller freed the slot and the address came back as pageable memory.
static uint32_t* device_alias_of_pinned(uint32_t* host) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, host) == hipSuccess && a.type == hipMemoryTypeHost && a.devicePointer)
    return (uint32_t*)a.devicePointer;
  (void)hipGetLastError();
  return nullptr;
}

// Geometry stages of the forward: projection, num_rendered (kept on the device in meta[2..3] and copied to `host_status`),
// depth order, tile-count prefix sum.  Never waits for the device.
static int forward_geometry(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, size_t geometry_bytes,
                            int32_t* radii, hipStream_t st, bool defer_color, uint32_t* host_status,
                            hipEvent_t copied /* recorded right behind the status copy, or nullptr */,
                            SideShade* shade_aside = nullptr, bool tile_local = false,
                            unsigned long long** early_word = nullptr,
                            uint32_t* tile_sort_head = nullptr /* tile-local form: cleared by the projection kernel */,
                            const uint32_t* tile_cutoff = nullptr /* lists truncated by depth (gsr_forward_async_culled) */,
                            uint32_t* culled_any = nullptr, uint32_t frame_tag = 0) {
  const int P = g->P;
  const GsrGeomLayout L = gsr_geom_layout(P);
  if (!geometry_state || geometry_bytes < L.total) {
    gsr_set_error("geometry state too small: %zu < %zu", geometry_bytes, L.total);
    return GSR_ERR_STATE_TOO_SMALL;
  }
  int rc;
  char* geom = (char*)geometry_state;
  uint32_t* meta = (uint32_t*)(geom + L.meta);
  // meta (num_rendered, flags) and, right behind it, the depth sort's digit histograms + pass tickets: cleared here for the
  // global-order form.  The tile-local form has no depth sort and its emission kernel writes every meta word itself
  // (k_emit_instances): no memset launch in front of the projection
  static_assert(sizeof(uint32_t) == 4, "");
  if (!tile_local && (rc = gsr_check(hipMemsetAsync(meta, 0, 256 + GSR_RADIX_HEAD_WORDS * 4, st), "memset meta"))) return rc;

  gsr_launch_preprocess_fwd(s, g, radii, geom, L, defer_color, /*block_sums=*/tile_local, tile_local ? tile_sort_head : nullptr,
                            GSR_RADIX_HEAD_WORDS, tile_cutoff, culled_any, frame_tag, st);
  if ((rc = debug_sync(s, st, "preprocess"))) return rc;
  // (tile-local binning form: the colour pass is forked behind the emission instead - forward_render_impl - because the
  // two are both HBM-bound and slowed each other down (emission 36 -> 55 us); the tile sort and the per-tile ordering that
  // follow are latency- / issue-bound and share the machine well: -12 us per frame at C3)
  if (shade_aside && !tile_local) {
    if ((rc = gsr_check(hipEventRecord(shade_aside->fork, st), "fork shade"))) return rc;
    if ((rc = gsr_check(hipStreamWaitEvent(shade_aside->stream, shade_aside->fork, 0), "fork shade"))) return rc;
    gsr_launch_shade(s, g, geom, L, /*beside_other_work=*/true, shade_aside->stream);
    if ((rc = gsr_check(hipEventRecord(shade_aside->join, shade_aside->stream), "join shade"))) return rc;
  }

  // num_rendered (meta[2..3]) and the error flags go back to the host NOW, ahead of the depth sort and the offset scan:
  // a blocking caller waits for the two words while the GPU still has work queued (no idle gap at the read-back), a
  // non-blocking caller looks at them whenever it likes.
  if (tile_local) {
    // second form of the binning stage (binning.hip, k_tile_depth_sort): no global depth order; the instances are emitted in
    // index order, the projection kernel has left per-workgroup instance totals and k_emit_instances (forward_render_impl)
    // takes the prefix sum and num_rendered from them itself.  A caller that waits for the count - gsr_forward_async(
    // num_rendered_out) - gets it from that kernel: one 8-byte store into this thread's pinned slot, word pair [8..9],
    // GSR_COUNT_VALID | flag << 62 | count; here only the slot is armed.
    unsigned long long* early = nullptr;
    if (host_status) {
      DeviceLocal* d = device_local();
      if (d && d->pinned == host_status && d->pinned_dev) {
        ((volatile unsigned long long*)(host_status + 8))[0] = 0ull;
        early = (unsigned long long*)(d->pinned_dev + 8);
      }
    }
    if (early_word) *early_word = early;
    return debug_sync(s, st, "projection");
  }
  gsr_launch_sum_tiles(P, geom, L, meta, st);
  if (host_status &&
      (rc = gsr_check(hipMemcpyAsync(host_status, meta, 16, hipMemcpyDeviceToHost, st), "read num_rendered")))
    return rc;
  if (copied && (rc = gsr_check(hipEventRecord(copied, st), "record read-back event"))) return rc;

  // depth order of the Gaussians (stable, so equal depths keep ascending id); 4 passes -> result in (depth_key, order)
  const int where = gsr_radix_sort_pairs((uint32_t*)(geom + L.depth_key), (uint32_t*)(geom + L.order),
                                         (uint32_t*)(geom + L.key_tmp), (uint32_t*)(geom + L.val_tmp),
                                         /*vals_iota=*/true, (size_t)P, 32, (uint32_t*)(geom + L.radix_tmp), st, nullptr,
                                         nullptr, nullptr, /*head_zeroed=*/true, /*fail_flags=*/meta);
  if (where != 0) { gsr_set_error("internal: depth sort ended in the wrong buffer"); return GSR_ERR_HIP; }
  if ((rc = debug_sync(s, st, "depth sort", meta))) return rc;

  // inclusive prefix sum of tiles_touched in depth order (its last element equals num_rendered)
  gsr_scan_u32((const uint32_t*)(geom + L.tiles_touched), (const uint32_t*)(geom + L.order),
               (uint32_t*)(geom + L.offsets), (size_t)P, 1, (uint32_t*)(geom + L.scan_tmp), st);
  if ((rc = debug_sync(s, st, "tile-count scan"))) return rc;
  return 0;
}

// Waits until the count of the forward just enqueued by THIS host thread has reached its pinned slot (`ev` was recorded right
// behind the kernel / copy that delivers it - the rest of the forward may still be queued behind it) and decodes it.
#define GSR_COUNT_VALID (1ull << 63)
// Never a wait without a bound: the host POLLS - the count word the scan kernel stores into pinned memory, or the event behind
// the 16-byte copy - first spinning (the count is ~50 us of device time away once the frame has started), then with short sleeps,
// and gives up loudly after GSR_COUNT_TIMEOUT_S seconds (default 120; a device fault aborts the process long before).
static int64_t wait_for_count(uint32_t* host, hipEvent_t ev, bool early_word) {
  static const double limit_s = getenv("GSR_COUNT_TIMEOUT_S") ? atof(getenv("GSR_COUNT_TIMEOUT_S")) : 120.0;
  volatile unsigned long long* w = (volatile unsigned long long*)(host + 8);
  unsigned long long v = 0;
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned long it = 0;; it++) {
    if (early_word) {
      v = *w;
      if (v & GSR_COUNT_VALID) break;
    } else {
      const hipError_t e = hipEventQuery(ev);
      if (e == hipSuccess) break;
      if (e != hipErrorNotReady) return gsr_check(e, "wait for num_rendered");
      (void)hipGetLastError();
    }
    if (it >= (early_word ? 20000ul : 200ul)) {
      struct timespec nap = {0, 2000};
      nanosleep(&nap, nullptr);
      if ((it & 255ul) == 0) {
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > limit_s) {
          gsr_set_error("num_rendered did not arrive within %.0f s (device hung, or the stream is blocked behind an event that "
                        "never completes)", limit_s);
          return GSR_ERR_HIP;
        }
      }
    }
  }
  unsigned long long total;
  bool culled;
  if (early_word) {
    culled = (v >> 62) & 1ull;
    total = v & ((1ull << 62) - 1ull);
  } else {
    total = (unsigned long long)host[2] | ((unsigned long long)host[3] << 32);
    culled = host[1] & 1u;
  }
  if (culled) {
    gsr_set_error("Point is filtered although prefiltered is set. This shouldn't happen!");
    return GSR_ERR_PREFILTERED_CULLED;
  }
  if (total > 0x3FFFFFFFull) {
    gsr_set_error("num_rendered %llu does not fit 30 bits", total);
    return GSR_ERR_TOO_MANY_INSTANCES;
  }
  return (int64_t)total;
}

static int64_t forward_prepare_impl(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state,
                                    size_t geometry_bytes, int32_t* radii, void* stream, bool defer_color) {
  int rc = validate(s, g);
  if (rc) return rc;
  if (g->P == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  uint32_t* host = pinned_slot();
  hipEvent_t ev = readback_event();
  if (!host || !ev) { gsr_set_error("hipHostMalloc / hipEventCreate failed"); return GSR_ERR_HIP; }
  // the host waits on an event recorded right behind the 16-byte copy, i.e. while the depth sort and the offset scan are still
  // queued: the GPU has ~0.1 ms of work left when the host goes on to size the binning state and enqueue the rest
  if ((rc = forward_geometry(s, g, geometry_state, geometry_bytes, radii, st, defer_color, host, ev, nullptr, false, nullptr)))
    return rc;
  return wait_for_count(host, ev, false);
}

int64_t gsr_forward_prepare(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state,
                            size_t geometry_bytes, int32_t* radii, void* stream) {
  return forward_prepare_impl(s, g, geometry_state, geometry_bytes, radii, stream, false);
}

int64_t gsr_forward_prepare_geometry(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state,
                                     size_t geometry_bytes, int32_t* radii, void* stream) {
  return forward_prepare_impl(s, g, geometry_state, geometry_bytes, radii, stream, true);
}

int gsr_forward_shade(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, void* stream) {
  int rc = validate(s, g);
  if (rc) return rc;
  if (g->P == 0) return 0;
  const GsrGeomLayout L = gsr_geom_layout(g->P);
  gsr_launch_shade(s, g, (char*)geometry_state, L, false, (hipStream_t)stream);
  if ((rc = debug_sync(s, (hipStream_t)stream, "shade"))) return rc;
  return gsr_launch_status("shade");
}

// `num_rendered` sizes the binning state and the grids (the CAPACITY); the number of instances really present is read by the
// kernels from the geometry state (min(meta num_rendered, capacity), gsr_eff_n).  On the blocking path the two are equal.
static int forward_render_impl(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, void* binning_state,
                               size_t binning_bytes, int64_t num_rendered, void* image_state, size_t image_bytes,
                               float* out_color, float* out_invdepth, bool for_backward, bool shade_late,
                               hipEvent_t sh_ready, void* stream, bool tile_local = false,
                               uint32_t* host_status_late = nullptr, unsigned long long* early = nullptr,
                               uint32_t* host_count = nullptr, hipEvent_t count_copied = nullptr,
                               bool sort_head_clean = false /* this call's projection kernel cleared the tile sort's head */,
                               uint32_t* tile_cutoff = nullptr /* per-tile depth cut-off: updated by the compositing kernel */,
                               bool cull_applied = false /* ... and the projection counted with it: emit with it too */,
                               uint32_t frame_tag = 0) {
  int rc = validate(s, g);
  if (rc) return rc;
  if (num_rendered < 0 || num_rendered > 0x3FFFFFFFll) {   // the tile sort counts keys in 30-bit fields (sort_scan.hip)
    gsr_set_error("num_rendered / capacity %lld out of range (limit 2^30 - 1)", (long long)num_rendered);
    return num_rendered < 0 ? GSR_ERR_INVALID_ARGUMENT : GSR_ERR_TOO_MANY_INSTANCES;
  }
  hipStream_t st = (hipStream_t)stream;
  const int W = s->image_width, H = s->image_height;
  const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
  const int tiles = gx * gy;
  const size_t R = (size_t)num_rendered;
  const GsrGeomLayout GL = gsr_geom_layout(g->P);
  const GsrBinLayout BL = gsr_bin_layout(R, tiles);
  const GsrImgLayout IL = gsr_img_layout(W, H);
  if (!binning_state || binning_bytes < BL.total || !image_state || image_bytes < IL.total) {
    gsr_set_error("binning/image state too small: %zu < %zu or %zu < %zu", binning_bytes, BL.total, image_bytes,
                  IL.total);
    return GSR_ERR_STATE_TOO_SMALL;
  }
  char* geom = (char*)geometry_state;
  char* bin = (char*)binning_state;
  char* img = (char*)image_state;
  const uint32_t* n_dev = g->P > 0 ? (const uint32_t*)(geom + GL.meta) + 2 : nullptr;
  // (round 4) walk classes: with a backward to follow, the compositing kernel files every tile under the class of its walk length
  // (image state), and the backward takes the classes longest first.  The 64 counters are cleared by the first workgroup of the
  // kernel in front of the compositing (tile-local form: the per-tile ordering), by a memset where there is none.
  uint32_t* walk_cnt = for_backward ? (uint32_t*)(img + IL.walk_cnt) : nullptr;
  bool walk_cnt_clear = false;
  if (R == 0 || g->P == 0) {   // nothing to emit: every tile range is empty (otherwise the emit kernel clears them on its way)
    if ((rc = gsr_check(hipMemsetAsync(bin + BL.ranges, 0, (size_t)tiles * 8, st), "memset ranges"))) return rc;
  } else {
    // (round 4) tile-local form: the emission counts the tile sort's digit histograms and clears its look-back table, the sort's
    // last pass leaves the tile ranges, the first per-tile kernel decodes them: no k_radix_hist_all, no k_finalize_bins
    // (GSR_TILE_HIST=0: the round-3 chain, for the A/B).  The sort's head must be clear BEFORE the emission's workgroups add
    // to it: the projection kernel of this call did that, or (re-render on another binning state) a memset here.
    const char* th = getenv("GSR_TILE_HIST");
    const bool fused_bins = tile_local && !(th && th[0] == '0');
    if (fused_bins && !sort_head_clean &&
        (rc = gsr_check(hipMemsetAsync(bin + BL.radix_tmp, 0, GSR_RADIX_HEAD_WORDS * 4, st), "memset sort head")))
      return rc;
    gsr_launch_emit(g->P, gx, tiles, geom, GL, bin, BL, (uint32_t)R, tile_local, early, fused_bins ? tile_bits(tiles) : 0,
                    cull_applied ? tile_cutoff : nullptr, st);
    // tile-local form: the emission kernel is where num_rendered comes into being.  A waiting caller whose pinned slot has no
    // device alias (early == nullptr) gets the status words by a copy, marked by an event
    if (tile_local && host_count && !early) {
      if ((rc = gsr_check(hipMemcpyAsync(host_count, geom + GL.meta, 16, hipMemcpyDeviceToHost, st), "read num_rendered"))) return rc;
      if (count_copied && (rc = gsr_check(hipEventRecord(count_copied, st), "record read-back event"))) return rc;
    }
    {
      SideShade* a = (tile_local && !shade_late && sh_ready) ? side_shade() : nullptr;
      if (a && sh_ready == a->join) {      // colour pass on the side stream, beside the tile sort (see forward_geometry)
        if ((rc = gsr_check(hipEventRecord(a->fork, st), "fork shade"))) return rc;
        if ((rc = gsr_check(hipStreamWaitEvent(a->stream, a->fork, 0), "fork shade"))) return rc;
        gsr_launch_shade(s, g, (char*)geom, GL, /*beside_other_work=*/true, a->stream);
        if ((rc = gsr_check(hipEventRecord(a->join, a->stream), "join shade"))) return rc;
      }
    }
    if ((rc = debug_sync(s, st, "emit instances"))) return rc;
    // With a backward to follow, the sort carries (emission slot, Gaussian id): the slot of every list position is where the
    // backward stores that instance's gradient record.  A forward-only render (torch.no_grad) needs the ids alone: they
    // become the one sorted value (same ping-pong parity, so the list ends up in the same place), 8 B less per key and pass.
    const int where =
        for_backward
            ? gsr_radix_sort_pairs((uint32_t*)(bin + BL.key_a), (uint32_t*)(bin + BL.val_a), (uint32_t*)(bin + BL.key_b),
                                   (uint32_t*)(bin + BL.val_b), /*vals_iota=*/true, R, tile_bits(tiles),
                                   (uint32_t*)(bin + BL.radix_tmp), st, (uint32_t*)(bin + BL.gauss_of_slot),
                                   (uint32_t*)(bin + BL.point_list), n_dev, /*head_zeroed (by the emit kernel)=*/true,
                                   /*fail_flags=*/(uint32_t*)(geom + GL.meta), /*hist_counted=*/fused_bins,
                                   fused_bins ? (uint2*)(bin + BL.ranges_enc) : nullptr)
            : gsr_radix_sort_pairs((uint32_t*)(bin + BL.key_a), (uint32_t*)(bin + BL.gauss_of_slot),
                                   (uint32_t*)(bin + BL.key_b), (uint32_t*)(bin + BL.point_list), /*vals_iota=*/false, R,
                                   tile_bits(tiles), (uint32_t*)(bin + BL.radix_tmp), st, nullptr, nullptr, n_dev, true,
                                   (uint32_t*)(geom + GL.meta), fused_bins, fused_bins ? (uint2*)(bin + BL.ranges_enc) : nullptr);
    if (where != tile_sort_result_buffer(tiles)) { gsr_set_error("internal: tile sort buffer parity"); return GSR_ERR_HIP; }
    if ((rc = debug_sync(s, st, "tile sort", (const uint32_t*)(geom + GL.meta)))) return rc;
    if (!fused_bins) {
      const uint32_t* ks = (const uint32_t*)(bin + (where ? BL.key_b : BL.key_a));
      gsr_launch_finalize((uint32_t)R, n_dev, ks, bin, BL, st);
      if ((rc = debug_sync(s, st, "finalize bins"))) return rc;
    }
    if (tile_local) {
      // every tile orders its own list by (depth bits, id); the free halves of the tile sort's ping-pong buffers serve the
      // (slow) path for lists beyond the LDS capacity
      uint32_t* free_k = (uint32_t*)(bin + (where ? BL.key_a : BL.key_b));
      uint32_t* free_w = (uint32_t*)(bin + (where ? BL.gauss_of_slot : BL.point_list));
      uint32_t* free_v = (uint32_t*)(bin + (for_backward ? (where ? BL.val_a : BL.val_b) : BL.val_a));
      uint32_t* slots = for_backward ? (uint32_t*)(bin + (where ? BL.val_b : BL.val_a)) : nullptr;
      gsr_launch_tile_depth_sort(tiles, for_backward, (uint2*)(bin + BL.ranges),
                                 fused_bins ? (const uint2*)(bin + BL.ranges_enc) : nullptr,
                                 (uint32_t*)(bin + point_list_offset(BL, tiles)), slots,
                                 (const uint32_t*)(geom + GL.depth_key), free_k, free_v, free_w,
                                 (uint32_t*)(geom + GL.meta), walk_cnt, st);
      walk_cnt_clear = true;
      if ((rc = debug_sync(s, st, "tile depth sort"))) return rc;
    }
  }
  if (walk_cnt && !walk_cnt_clear &&
      (rc = gsr_check(hipMemsetAsync(walk_cnt, 0, GSR_WALK_CLASSES * 4, st), "memset walk classes")))
    return rc;
  // status words of the non-blocking forward: written by the compositing kernel itself when the caller's slot is pinned
  // host memory (it is mapped into the device's address space), copied otherwise
  uint32_t* status_dev = (host_status_late && g->P > 0) ? device_alias_of_pinned(host_status_late) : nullptr;
  if (host_status_late && g->P > 0 && !status_dev &&
      (rc = gsr_check(hipMemcpyAsync(host_status_late, geom + GL.meta, 32, hipMemcpyDeviceToHost, st), "read status")))
    return rc;
  if (!shade_late && sh_ready && g->P > 0) {   // colours were evaluated on a side stream: join it
    if ((rc = gsr_check(hipStreamWaitEvent(st, sh_ready, 0), "join shade"))) return rc;
  }
  if (shade_late && g->P > 0) {
    // the colours are the LAST thing the compositing needs: everything above ran without the SH coefficients, which may
    // still be receiving their update on another stream
    if (sh_ready && (rc = gsr_check(hipStreamWaitEvent(st, sh_ready, 0), "wait for the SH coefficients"))) return rc;
    gsr_launch_shade(s, g, geom, GL, false, st);
    if ((rc = debug_sync(s, st, "shade"))) return rc;
  }
  gsr_launch_render_fwd(s, tiles, gx, (const uint2*)(bin + BL.ranges), (const uint32_t*)(bin + point_list_offset(BL, tiles)),
                        (const float4*)(geom + GL.rec), out_color, out_invdepth, (float*)(img + IL.final_T),
                        (uint32_t*)(img + IL.n_contrib), status_dev ? (const uint32_t*)(geom + GL.meta) : nullptr, status_dev,
                        (g->P > 0 && R > 0) ? tile_cutoff : nullptr, (const uint32_t*)(geom + GL.depth_key),
                        cull_applied ? (const uint32_t*)(bin + BL.culled_any) : nullptr, frame_tag, (uint32_t*)(geom + GL.meta),
                        walk_cnt, walk_cnt ? (uint32_t*)(img + IL.walk_list) : nullptr,
                        walk_cnt ? (uint32_t*)(img + IL.walk_of_tile) : nullptr, st);
  if ((rc = debug_sync(s, st, "render forward"))) return rc;
  return gsr_launch_status("forward");
}

int gsr_forward_render(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, void* binning_state,
                       size_t binning_bytes, int64_t num_rendered, void* image_state, size_t image_bytes,
                       float* out_color, float* out_invdepth, int32_t for_backward, void* stream) {
  return forward_render_impl(s, g, geometry_state, binning_state, binning_bytes, num_rendered, image_state, image_bytes,
                             out_color, out_invdepth, for_backward != 0, false, nullptr, stream);
}

int gsr_forward_render_shade(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, void* binning_state,
                             size_t binning_bytes, int64_t num_rendered, void* image_state, size_t image_bytes,
                             float* out_color, float* out_invdepth, int32_t for_backward, void* sh_ready_event,
                             void* stream) {
  return forward_render_impl(s, g, geometry_state, binning_state, binning_bytes, num_rendered, image_state, image_bytes,
                             out_color, out_invdepth, for_backward != 0, true, (hipEvent_t)sh_ready_event, stream);
}

static int forward_async_impl(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, size_t geometry_bytes,
                      int32_t* radii, void* binning_state, size_t binning_bytes, int64_t capacity, void* image_state,
                      size_t image_bytes, float* out_color, float* out_invdepth, int32_t for_backward,
                      int32_t defer_color, void* sh_ready_event, uint32_t* host_status, int32_t tile_local_sort,
                      void* stream, int64_t* num_rendered_out, uint32_t* tile_cutoff, bool cull_apply) {
  int rc = validate(s, g);
  if (rc) return rc;
  const bool tlo = tile_local_sort != 0;
  if (num_rendered_out) *num_rendered_out = 0;
  if (g->P > 0) {
    const bool late = defer_color != 0 && !g->colors_precomp;
    const char* aside_str = getenv("GSR_SHADE_STREAM");      // (read per call: tests switch it inside one process)
    const int aside_env = aside_str ? atoi(aside_str) : -1;
    const bool want_aside = aside_env < 0 ? g->P >= 200000 : aside_env != 0;
    SideShade* aside = (want_aside && !late && !g->colors_precomp && (g->shs || g->dc) && !s->debug) ? side_shade() : nullptr;
    // verified speculation (num_rendered_out): the count goes to this thread's pinned slot as EARLY as the kernels know it,
    // the whole frame is enqueued for `capacity`, and only then does the host wait for the count - the device still has the
    // binning and compositing stages queued, so it never idles while the host looks
    uint32_t* host = nullptr;
    hipEvent_t ev = nullptr;
    unsigned long long* early = nullptr;
    if (num_rendered_out) {
      host = pinned_slot();
      ev = readback_event();
      if (!host || !ev) { gsr_set_error("hipHostMalloc / hipEventCreate failed"); return GSR_ERR_HIP; }
    }
    // (tile-local form: the projection kernel clears the head of the tile sort's scratch in the caller's binning state, so that
    // the emission's workgroups can add their digit counts to it)
    uint32_t* sort_head = nullptr;
    uint32_t* culled_any = nullptr;
    static std::atomic<uint32_t> frame_counter{0};
    const uint32_t frame_tag = ++frame_counter | 0x40000000u;     // (never 0: what a cleared buffer holds)
    // lists truncated by the per-tile depth cut-off: only on the path whose frames are looked at AFTERWARDS (unverified: a frame
    // whose truncation turns out too tight flags itself, its backward is a no-op, the caller renders it again), in the tile-local
    // binning form (the projection kernel clears the per-tile "lost an instance" flags next to the sort's head)
    bool cull = cull_apply && tile_cutoff != nullptr && tlo && num_rendered_out == nullptr && !s->debug;
    if (tlo && binning_state && capacity > 0) {
      const int gx0 = (s->image_width + GSR_TILE - 1) / GSR_TILE, gy0 = (s->image_height + GSR_TILE - 1) / GSR_TILE;
      const GsrBinLayout BL0 = gsr_bin_layout((size_t)capacity, (size_t)gx0 * gy0);
      if (binning_bytes >= BL0.total) {
        sort_head = (uint32_t*)((char*)binning_state + BL0.radix_tmp);
        culled_any = (uint32_t*)((char*)binning_state + BL0.culled_any);
      }
    }
    if (!culled_any) cull = false;
    if ((rc = forward_geometry(s, g, geometry_state, geometry_bytes, radii, (hipStream_t)stream, late || aside != nullptr,
                               host, ev, aside, tlo, &early, sort_head, cull ? tile_cutoff : nullptr, cull ? culled_any : nullptr,
                               frame_tag)))
      return rc;
    // (the caller's status words - flags, num_rendered and, in the tile-local form, the longest tile list meta[4] - leave at the END)
    rc = forward_render_impl(s, g, geometry_state, binning_state, binning_bytes, capacity, image_state, image_bytes,
                             out_color, out_invdepth, for_backward != 0, aside ? false : late,
                             aside ? aside->join : (hipEvent_t)sh_ready_event, stream, tlo, host_status, early, tlo ? host : nullptr,
                             ev, /*sort_head_clean=*/sort_head != nullptr, tile_cutoff, cull, frame_tag);
    if (rc) return rc;
    if (num_rendered_out) {
      const int64_t n = wait_for_count(host, ev, early != nullptr);
      if (n < 0) return (int)n;
      *num_rendered_out = n;
    }
    return 0;
  }
  return forward_render_impl(s, g, geometry_state, binning_state, binning_bytes, 0, image_state, image_bytes, out_color,
                             out_invdepth, for_backward != 0, false, nullptr, stream);
}

int gsr_forward_async(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, size_t geometry_bytes,
                      int32_t* radii, void* binning_state, size_t binning_bytes, int64_t capacity, void* image_state,
                      size_t image_bytes, float* out_color, float* out_invdepth, int32_t for_backward,
                      int32_t defer_color, void* sh_ready_event, uint32_t* host_status, int32_t tile_local_sort,
                      void* stream, int64_t* num_rendered_out) {
  return forward_async_impl(s, g, geometry_state, geometry_bytes, radii, binning_state, binning_bytes, capacity, image_state,
                            image_bytes, out_color, out_invdepth, for_backward, defer_color, sh_ready_event, host_status,
                            tile_local_sort, stream, num_rendered_out, nullptr, false);
}

int gsr_forward_async_culled(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, size_t geometry_bytes,
                             int32_t* radii, void* binning_state, size_t binning_bytes, int64_t capacity, void* image_state,
                             size_t image_bytes, float* out_color, float* out_invdepth, int32_t for_backward,
                             int32_t defer_color, void* sh_ready_event, uint32_t* host_status, int32_t tile_local_sort,
                             void* stream, int64_t* num_rendered_out, uint32_t* tile_depth_cutoff, int32_t apply) {
  return forward_async_impl(s, g, geometry_state, geometry_bytes, radii, binning_state, binning_bytes, capacity, image_state,
                            image_bytes, out_color, out_invdepth, for_backward, defer_color, sh_ready_event, host_status,
                            tile_local_sort, stream, num_rendered_out, tile_depth_cutoff, apply != 0);
}

int gsr_forward_rerender(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, void* binning_state,
                         size_t binning_bytes, int64_t capacity, void* image_state, size_t image_bytes, float* out_color,
                         float* out_invdepth, int32_t for_backward, int32_t tile_local_sort, uint32_t* host_status,
                         void* stream) {
  // everything phase 2 reads of the geometry state - records with their colours, tile counts, the count itself, depth order
  // or per-workgroup start slots - was left complete by the gsr_forward_async call this one repairs
  return forward_render_impl(s, g, geometry_state, binning_state, binning_bytes, capacity, image_state, image_bytes,
                             out_color, out_invdepth, for_backward != 0, false, nullptr, stream, tile_local_sort != 0,
                             host_status);
}

static int adam_args(const gsr_gaussians* g, const gsr_fused_adam* opt, GsrAdamArgs& A) {
  float* params[6] = {(float*)g->means3D, (float*)g->dc, (float*)g->shs, (float*)g->opacities, (float*)g->scales,
                      (float*)g->rotations};
  for (int i = 0; i < 6; i++) {
    const bool present = params[i] != nullptr;
    if (present && (!opt->exp_avg[i] || !opt->exp_avg_sq[i])) {
      gsr_set_error("fused adam: no moments for parameter group %d", i);
      return GSR_ERR_INVALID_ARGUMENT;
    }
    A.p[i] = params[i]; A.m[i] = opt->exp_avg[i]; A.v[i] = opt->exp_avg_sq[i];
    A.lr[i] = opt->lr[i];
    const double bc1 = 1.0 - pow(opt->beta1, (double)opt->step[i]);
    const double bc2 = 1.0 - pow(opt->beta2, (double)opt->step[i]);
    A.step_size[i] = opt->sparse == 1 ? 0.f : (float)((double)opt->lr[i] / bc1);
    A.inv_bc2_sqrt[i] = opt->sparse == 1 ? 0.f : (float)(1.0 / sqrt(bc2));
  }
  A.beta1 = (float)opt->beta1; A.beta2 = (float)opt->beta2;
  A.omb1 = (float)(1.0 - opt->beta1); A.omb2 = (float)(1.0 - opt->beta2);
  A.eps = (float)opt->eps;
  A.dyn = opt->dynamic;
  return 0;
}

__global__ void k_adam_set_dynamic(GsrAdamArgs A, float* __restrict__ dyn) {
  const int i = threadIdx.x;
  if (i < 6) {
    dyn[i] = A.lr[i];
    dyn[6 + i] = A.step_size[i];
    dyn[12 + i] = A.inv_bc2_sqrt[i];
  }
}

extern "C" int gsr_adam_set_dynamic(const gsr_fused_adam* opt, float* dynamic_dev, void* stream) {
  if (!opt || !dynamic_dev) {
    gsr_set_error("adam_set_dynamic: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  // (the same arithmetic as adam_args: lr / (1 - beta1^step) and 1 / sqrt(1 - beta2^step) formed in double)
  GsrAdamArgs A;
  for (int i = 0; i < 6; i++) {
    A.p[i] = A.m[i] = A.v[i] = nullptr;
    A.lr[i] = opt->lr[i];
    const double bc1 = 1.0 - pow(opt->beta1, (double)opt->step[i]);
    const double bc2 = 1.0 - pow(opt->beta2, (double)opt->step[i]);
    A.step_size[i] = opt->sparse == 1 ? 0.f : (float)((double)opt->lr[i] / bc1);
    A.inv_bc2_sqrt[i] = opt->sparse == 1 ? 0.f : (float)(1.0 / sqrt(bc2));
  }
  A.beta1 = A.beta2 = A.omb1 = A.omb2 = A.eps = 0.f;
  A.dyn = nullptr;
  hipLaunchKernelGGL(k_adam_set_dynamic, dim3(1), dim3(64), 0, (hipStream_t)stream, A, dynamic_dev);
  return gsr_launch_status("adam set dynamic");
}

static int backward_impl(const gsr_settings* s, const gsr_gaussians* g, const int32_t* radii, const void* geometry_state,
                         const void* binning_state, const void* image_state, int64_t num_rendered, const float* dL_dcolor,
                         const float* dL_dinvdepth, void* scratch, size_t scratch_bytes, const gsr_grads* grads,
                         const gsr_fused_adam* opt, void* stream) {
  int rc = validate(s, g);
  if (rc) return rc;
  if (!grads || !grads->dL_dmeans2D || !dL_dcolor || (!opt && (!grads->dL_dmeans3D || !grads->dL_dopacities))) {
    gsr_set_error("backward: missing mandatory gradient buffers");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if ((grads->xyz_gradient_accum != nullptr) != (grads->denom != nullptr) ||
      (grads->denom != nullptr) != (grads->max_radii2D != nullptr)) {
    gsr_set_error("backward: densification statistics need all three arrays (or none)");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (g->P == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int W = s->image_width, H = s->image_height;
  const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
  const int tiles = gx * gy;
  const size_t R = (size_t)num_rendered;
  if (scratch_bytes < gsr_backward_scratch_bytes(g->P, num_rendered) || !scratch) {
    gsr_set_error("backward scratch too small");
    return GSR_ERR_STATE_TOO_SMALL;
  }
  const GsrGeomLayout GL = gsr_geom_layout(g->P);
  const GsrBinLayout BL = gsr_bin_layout(R, tiles);
  const GsrImgLayout IL = gsr_img_layout(W, H);
  const char* geom = (const char*)geometry_state;
  const char* bin = (const char*)binning_state;
  const char* img = (const char*)image_state;
  float4* igrad = (float4*)scratch;
  if (R > 0) {
    gsr_launch_render_bwd(s, tiles, gx, (const uint2*)(bin + BL.ranges), (const uint32_t*)(bin + point_list_offset(BL, tiles)),
                          (const float4*)(geom + GL.rec), (const float*)(img + IL.final_T),
                          (const uint32_t*)(img + IL.n_contrib), dL_dcolor, dL_dinvdepth,
                          (const uint32_t*)(bin + (tile_sort_result_buffer(tiles) ? BL.val_b : BL.val_a)), igrad,
                          (const uint32_t*)(geom + GL.meta) + 2, (uint32_t)R, (const uint32_t*)(img + IL.walk_cnt),
                          (const uint32_t*)(img + IL.walk_list), (const uint32_t*)(img + IL.walk_of_tile), st);
    if ((rc = debug_sync(s, st, "render backward"))) return rc;
  }
  GsrAdamArgs A;
  if (opt && (rc = adam_args(g, opt, A))) return rc;
  // rows: 0 every row (dense) / 1 rows with radii > 0, no bias correction (sparse) / 2 dense, rows with instances only
  const int mode = !opt ? 0 : (opt->sparse == 1 ? 2 : (opt->sparse == 2 ? 3 : 1));
  if (gsr_launch_preprocess_bwd(s, g, radii, geom, GL, igrad, (uint32_t)R, grads, opt ? &A : nullptr, mode, st)) {
    gsr_set_error("backward_adam needs the raw-parameter call form with dc / shs passed separately (raw_activations = 1, "
                  "dc != NULL, no colors_precomp / cov3D_precomp, every stored SH coefficient active)");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if ((rc = debug_sync(s, st, "preprocess backward"))) return rc;
  return gsr_launch_status("backward");
}

int gsr_backward(const gsr_settings* s, const gsr_gaussians* g, const int32_t* radii, const void* geometry_state,
                 const void* binning_state, const void* image_state, int64_t num_rendered, const float* dL_dcolor,
                 const float* dL_dinvdepth, void* scratch, size_t scratch_bytes, const gsr_grads* grads,
                 void* stream) {
  return backward_impl(s, g, radii, geometry_state, binning_state, image_state, num_rendered, dL_dcolor, dL_dinvdepth,
                       scratch, scratch_bytes, grads, nullptr, stream);
}

int gsr_backward_adam(const gsr_settings* s, const gsr_gaussians* g, const int32_t* radii, const void* geometry_state,
                      const void* binning_state, const void* image_state, int64_t num_rendered, const float* dL_dcolor,
                      const float* dL_dinvdepth, void* scratch, size_t scratch_bytes, const gsr_grads* grads,
                      const gsr_fused_adam* opt, void* stream) {
  if (!opt) { gsr_set_error("backward_adam: null optimizer arguments"); return GSR_ERR_INVALID_ARGUMENT; }
  return backward_impl(s, g, radii, geometry_state, binning_state, image_state, num_rendered, dL_dcolor, dL_dinvdepth,
                       scratch, scratch_bytes, grads, opt, stream);
}

int gsr_adam_step_culled_rows(const gsr_gaussians* g, const void* geometry_state, int64_t num_rendered,
                              const gsr_fused_adam* opt, void* stream) {
  if (!g || !geometry_state || !opt || opt->sparse == 1 || num_rendered < 0 || num_rendered > 0x3FFFFFFFll) {
    gsr_set_error("adam_step_culled_rows: bad arguments (dense Adam only)");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (g->P == 0) return 0;
  if (!g->raw_activations || !g->dc || g->colors_precomp || g->cov3D_precomp || !g->scales || !g->rotations ||
      (g->shs == nullptr) != (g->sh_coeffs == 0)) {
    gsr_set_error("adam_step_culled_rows needs the raw-parameter call form with dc / shs passed separately");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  GsrAdamArgs A;
  int rc = adam_args(g, opt, A);
  if (rc) return rc;
  gsr_launch_adam_culled_rows(g->P, g->shs ? g->sh_coeffs : 0, (const char*)geometry_state, gsr_geom_layout(g->P), A,
                              (uint32_t)num_rendered, (hipStream_t)stream);
  return gsr_launch_status("adam culled rows");
}

int gsr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, uint8_t* present, void* stream) {
  if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) {
    gsr_set_error("mark_visible: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (P == 0) return 0;
  gsr_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
  return gsr_launch_status("mark_visible");
}

int gsr_debug_geometry_views(const void* geometry_state, int32_t P, const float** rec48,
                             const uint32_t** depth_keys_sorted, const uint32_t** order, const uint32_t** tiles_touched,
                             const uint16_t** rect, const uint32_t** offsets, const uint8_t** clamped) {
  const GsrGeomLayout L = gsr_geom_layout((size_t)(P < 0 ? 0 : P));
  const char* geom = (const char*)geometry_state;
  if (rec48) *rec48 = (const float*)(geom + L.rec);
  if (depth_keys_sorted) *depth_keys_sorted = (const uint32_t*)(geom + L.depth_key);
  if (order) *order = (const uint32_t*)(geom + L.order);
  if (tiles_touched) *tiles_touched = (const uint32_t*)(geom + L.tiles_touched);
  if (rect) *rect = (const uint16_t*)(geom + L.rect);
  if (offsets) *offsets = (const uint32_t*)(geom + L.offsets);
  if (clamped) *clamped = (const uint8_t*)(geom + L.clamped);
  return 0;
}

int gsr_debug_binning_views(const void* binning_state, int32_t W, int32_t H, int64_t R, const uint32_t** point_list,
                            const uint32_t** ranges) {
  const size_t tiles = (size_t)((W + GSR_TILE - 1) / GSR_TILE) * (size_t)((H + GSR_TILE - 1) / GSR_TILE);
  const GsrBinLayout BL = gsr_bin_layout((size_t)R, tiles);
  const char* bin = (const char*)binning_state;
  if (point_list) *point_list = (const uint32_t*)(bin + point_list_offset(BL, tiles));
  if (ranges) *ranges = (const uint32_t*)(bin + BL.ranges);
  return 0;
}

int gsr_debug_count_pairs(const gsr_settings* s, int32_t P, const void* geometry_state, const void* binning_state,
                          int64_t num_rendered, uint32_t* pairs, void* stream) {
  if (!s || !geometry_state || !binning_state || !pairs || P <= 0) {
    gsr_set_error("count_pairs: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  const int W = s->image_width, H = s->image_height;
  const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
  const int tiles = gx * gy;
  const GsrGeomLayout GL = gsr_geom_layout(P);
  const GsrBinLayout BL = gsr_bin_layout((size_t)num_rendered, tiles);
  const char* geom = (const char*)geometry_state;
  const char* bin = (const char*)binning_state;
  gsr_launch_count_pairs(s, tiles, gx, (const uint2*)(bin + BL.ranges), (const uint32_t*)(bin + point_list_offset(BL, tiles)),
                         (const float4*)(geom + GL.rec), pairs, (hipStream_t)stream);
  return gsr_launch_status("count_pairs");
}

size_t gsr_debug_radix_tmp_bytes(int64_t n) { return gsr_align(gsr_radix_tmp_elems((size_t)(n < 1 ? 1 : n)) * 4); }

int gsr_debug_radix_sort(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, uint32_t* w0, uint32_t* w1, int64_t n,
                         int32_t bits, int32_t vals_iota, const uint32_t* n_dev, void* tmp, void* stream) {
  if (n < 0 || n > 0x3FFFFFFF || bits < 1 || bits > 32 || !k0 || !v0 || !k1 || !v1 || !tmp || (w0 == nullptr) != (w1 == nullptr)) {
    gsr_set_error("debug_radix_sort: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  const int where = gsr_radix_sort_pairs(k0, v0, k1, v1, vals_iota != 0, (size_t)n, bits, (uint32_t*)tmp, (hipStream_t)stream,
                                         w0, w1, n_dev, false);
  const int rc = gsr_launch_status("debug radix sort");
  return rc ? rc : where;
}

int gsr_debug_image_views(const void* image_state, int32_t W, int32_t H, const float** final_T,
                          const uint32_t** n_contrib) {
  const GsrImgLayout IL = gsr_img_layout(W, H);
  const char* img = (const char*)image_state;
  if (final_T) *final_T = (const float*)(img + IL.final_T);
  if (n_contrib) *n_contrib = (const uint32_t*)(img + IL.n_contrib);
  return 0;
}

int gsr_debug_walk_views(const void* image_state, int32_t W, int32_t H, const uint32_t** walk_cnt, const uint32_t** walk_list,
                         const uint32_t** walk_of_tile) {
  const GsrImgLayout IL = gsr_img_layout(W, H);
  const char* img = (const char*)image_state;
  if (walk_cnt) *walk_cnt = (const uint32_t*)(img + IL.walk_cnt);
  if (walk_list) *walk_list = (const uint32_t*)(img + IL.walk_list);
  if (walk_of_tile) *walk_of_tile = (const uint32_t*)(img + IL.walk_of_tile);
  return GSR_WALK_CLASSES;
}

void gsr_profile_enable(int32_t on) { g_gsr_profile_on = on ? 1 : 0; }

int64_t gsr_debug_set_flags_min_r(int64_t min_instances) {
  const int64_t old = (int64_t)g_gsr_flags_min_r;
  if (min_instances >= 0) g_gsr_flags_min_r = min_instances > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)min_instances;
  return old;
}

void gsr_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  collect_locked();
  g_totals.clear();
}

int32_t gsr_profile_read(const char** names, double* total_ms, int64_t* calls, int32_t max) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  collect_locked();
  g_names_keepalive.clear();
  for (auto& kv : g_totals) g_names_keepalive.push_back(kv.first);
  int32_t i = 0;
  for (auto& kv : g_totals) {
    if (i < max) {
      if (names) names[i] = g_names_keepalive[i].c_str();
      if (total_ms) total_ms[i] = kv.second.first;
      if (calls) calls[i] = kv.second.second;
    }
    i++;
  }
  return i;
}

}  // extern "C"
