// smk_internal.h -- context and kernel-parameter structures of the HIP ray-marcher.
// Product code: never includes anything from oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/smk.h"

#define SMK_MAX_RANKS 8
#define SMK_TIMING_RING 64
#define SMK_BRICK_LOG2 3   // bricks of 8x8x8 cells (smk_bricks.hip)
#define SMK_SHADOW_BOX_EPS 0.0009765625f  // voxels: frames with shadows test an eye sample against the box widened by this (smk_api.hip)
#define SMK_TUNE_SETTLE 6  // auto mode: untimed slice-ring frames before a new configuration's timed trial (smk_api.hip)
#define SMK_STATUS_RING 8  // frames whose slice-ring status stays readable (smk_frame_failed)

// Eye rays of a frame with shadows (half-angle slicing, smk_shadow.hip).  The slice planes are not perpendicular to the
// view axis, so a ray's coefficients are not affine in the pixel coordinate; they are (smk_ray_AB, smk_device.h)
//   D_a = fma(px, Dx_a, fma(py, Dy_a, Dc_a)),  nD = fma(px, nDx, fma(py, nDy, nDc)),
//   tauA = numA / nD, dtau = dB / nD,  A_a = fma(tauA, D_a, Ec_a),  B_a = dtau * D_a
// and plane m = 0..nplanes-1, counted FROM THE EYE, holds the sample fma(m, B, A) where fma(m, dtau, tauA) is positive and
// finite (a sample behind the eye does not exist).  Plane m is slice k = k0 + dk m of the light's order (k = 1..nslices away
// from the light); its sample is shaded under the light buffer as slices < k left it: hist[k - 1].
struct SmkShadowRays {
  int on;  // 0: the affine coefficients of smk_raycoef
  float Ec[3], Dc[3], Dx[3], Dy[3], nDc, nDx, nDy;
  float numA, dB;
  float llo[3], lhi[3];  // the box a LIGHT ray's sample must lie in (closed): the volume, or what an orthogonal clip plane leaves of it
  int k0, dk, LB;
  float Xm[4], Ym[4], Wm[4], lscale, lbias;
  const float4 *hist;  // [nslices + 1] buffers of [LB][LB] texels, `hstride` texels apart
  long long hstride;   // (LB * LB + a pad: buffers a power of two apart would meet in the same memory channels)
};

// Everything a render kernel needs, passed by value as the kernarg (wave-uniform => SGPRs).
struct RenderParams {
  // ---- volume, packed layout (DESIGN.md "HBM layout")
  //   u8 : uint2  {c0|c1<<8|c2<<16|c3<<24, n0|n1<<8|n2<<16}                 8 B/voxel
  //   f32: float4 {c0,c1,c2, c3 or normal bits}                            16 B/voxel
  const void *vox;
  const uint32_t *nrm;  // separate packed normals (only f32 with 4 channels), else null
  int N[3];             // whole-volume dims
  int O[3];             // global index of stored voxel (0,0,0)
  int D[3];             // stored dims (region + halo)
  float lo[3], hi[3];   // region in voxel coordinates: [g0-.5, g1-.5)
  int top[3];           // region touches the volume's top face on this axis (inclusive)
  int cplane_on;        // free clip plane: a sample stays when fma-chain(cplane . (p,1)) >= 0 (voxel coordinates)
  float cplane[4];
  float hin[3];         // largest coordinate that is inside: hi on a top face, else the float below hi
  float invN[3];
  int nelts;
  int n_in_w;  // f32: normal bits live in .w
  // ---- classification
  const float4 *tlut;  // premultiplied (r*a,g*a,b*a,a)
  int tlut_size;
  const uint32_t *tf_vg;  // [sg][sv] RGBA8
  const uint32_t *tf_h;   // [sg][sv] RGBA8 (alpha used) or null
  const uint32_t *tf_occ;  // [sg][occ_roww] bit s of row t: the bilinear lookup based at texel (s,t) can be non-transparent
  int occ_roww;
  int sv, sg, third_axis;
  const uint32_t *tf3d;  // [s3h][s3g][s3v]
  int s3v, s3g, s3h;
  // ---- brick flags (smk_bricks.hip): [nbr[2]][nbr[1]][nbr[0]] bytes over the stored box, 0 = no sample whose cell lies in
  // the brick can be visible under the current table; null = not available (1-D colour table, option "bricks" 0)
  const unsigned char *bricks;
  int nbr[3];
  const unsigned char *bricks_dil;  // the same, each flag spread over the bricks a perturbed fetch can reach from there, or null
  // ---- camera / sample placement
  smk_raycoef rc;
  SmkShadowRays sh;
  int W, H;
  float znear;
  // ---- shading
  int use_spec;
  float L[3], Hv[3];
  float R[9];  // rows of rinfo.xform's rotation: Nw = R * n
  float intens;
  int blend;  // smk_blend
  // ---- perturbation
  const uint32_t *noise;  // [nn][nn][nn] RGBA8
  int nn, pert_on;
  int nn_log2;  // log2(nn) when the noise texture's edge is a power of two, else -1
  float pw[2], ps[2];
  // ---- output
  float4 *out;
  float *depth;
  // ---- tile mapping
  int ntx, nty, tiles_per_xcd;
  int wave_w, blk_w, lockstep;  // gather-kernel tiling knobs (smk_set_option)
};

// side buffers of the slice-ring kernel, owned by the context
struct SlabAux {
  int *h_status = nullptr;      // pinned, device-visible: SMK_STATUS_RING error words (0 = ok), one per frame in turn
  int status_slot = 0;          // the word of the frame being launched
  int status_tag = 0;           // ... and that frame's id << 8, which the kernel writes with its status
  hipEvent_t frame_ev0 = nullptr;  // recorded by the launcher right before its first stream operation
  float *d_diag = nullptr;      // [16] diagnostic counters (option lockstep bit 16)
  int2 *d_order = nullptr;      // workgroup schedule of the current camera
  int2 *h_order[4] = {nullptr, nullptr, nullptr, nullptr};  // pinned staging copies, used in turn
  hipEvent_t order_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // completion of each one's last copy
  int order_next = 0;
  int order_cap = 0;
  std::vector<int2> order_host;  // what d_order holds
  struct Scan {  // the per-tile geometry scan of the last frame, per workgroup configuration tried (key = camera + region + tile shape)
    std::vector<unsigned char> key;
    double v[4] = {0, 0, 0, 0};
    std::vector<int> work;
  } scan[4];
  unsigned scan_next = 0;
  std::vector<unsigned char> shape_key;  // the small-workgroup shape chosen for this view class (see the launcher's probing pass)
  int shape_choice = -1, shape_age = 0, shape_chunks = 0;  // (... its age in frames, and the DMA chunks per slice it had when chosen)
  std::vector<int> plan_work;   // the weights the last schedule was built from,
  std::vector<unsigned char> plan_cuts;  // the cuts,
  std::vector<int2> plan_order;  // and that schedule
  int plan_slots = 0;
  // per-tile workgroup durations of an earlier frame: the schedule's weights
  unsigned *d_ticks = nullptr, *h_ticks = nullptr;  // device buffer the kernel writes; pinned copy in flight
  int ticks_cap = 0, ticks_pending_n = 0, ticks_age = 0, ticks_adopted = 0;
  int ticks_n_last = 0;  // tiles of the latest slice-ring launch (d_ticks holds [3][that many] words)
  bool ticks_pending = false;
  long long ticks_pending_sig = 0, ticks_good_sig = -1;
  hipEvent_t ticks_ev = nullptr;
  std::vector<unsigned> ticks_good;
  // DEPTH SEGMENTS (smk_slab.hip): partial frames of the pieces 1.. of split tiles, [maxseg - 1][W * H] float4
  void *d_seg = nullptr;
  size_t seg_cap = 0;
  int opt_split = 0;            // option "slab_split": 0 auto, 1 off, 2.. forced
  int nsplit_last = 0, nblocks_last = 0;
  std::vector<unsigned char> ksplit_last;  // pieces per tile of the latest launch
  unsigned *d_pticks = nullptr, *h_pticks = nullptr;  // [ntiles][8] durations of the pieces of split tiles (device; pinned copy)
  std::vector<unsigned> pticks_good;                   // the latest that came back ...
  std::vector<unsigned char> cuts, cuts_pending, cuts_good;  // [ntiles][10] {K, cut_0..cut_K}: current | of the copy in flight | of pticks_good
  int cuts_split = -1;
  long long cuts_sig = -1;
  bool recut = true, cuts_engaged = false, merge_warm = false;
  unsigned *d_trace = nullptr;  // [trace_n][8] workgroup timeline of the last traced frame (option lockstep bit 32)
  int trace_cap = 0, trace_n = 0;
};

// the column-stream kernel's side buffers (smk_cols.hip), owned by the context
struct ColLayout {  // the stored box re-laid out for one principal axis: [cv][cu][s][(CH+1)][(CW+1)] voxels
  void *d = nullptr;
  size_t bytes = 0;
  int CW = 0, CH = 0, ncu = 0, ncv = 0, Du = 0, Dv = 0, Ds = 0, slice_bytes = 0, vb = 0;
  const void *src = nullptr;  // the native volume it was built from
};
struct ColsAux {
  ColLayout lay[3];             // per principal axis (perm 0: S = z, 1: S = y, 2: S = x), built on first use
  void *d_layers = nullptr;     // [nkeys][npix] float4: a ray's partial composites, one per job it crosses
  size_t layers_cap = 0;
  void *d_masks = nullptr;      // [npix][mask_words] bit per key written
  size_t masks_cap = 0;
  bool masks_dirty = true;
  int mask_words_last = 0;
  unsigned *d_ticks = nullptr;  // [njobs] workgroup durations of the latest frame (100 MHz ticks)
  int ticks_cap = 0, njobs_last = 0;
  unsigned long long *d_counts = nullptr;  // [8] developer statistics of the latest frame (ColParams::counts)
  int want_counts = 0;
  int opt_fill = 0, opt_take_min = 0, opt_take_wait = 0, opt_fly = 0;  // developer knobs (smk_set_option cols_fill / cols_take_min / cols_take_wait)
  int builds = 0;               // layouts built so far
  int last = 0;                 // CW | CH << 8 | nslots << 16 | shape << 24 of the latest launch
  double last_stream_bytes = 0; // bytes the loaders of the latest launch had to stream
  hipEvent_t frame_ev0 = nullptr;
  int status_tag = 0;           // the frame's id << 8 (written with a status word)
};
void smk_cols_free(ColsAux *aux);
void smk_cols_drop_layouts(ColsAux *aux);

hipError_t smk_bricks_dilate(const unsigned char *flags, const int nb[3], const int r[3], unsigned char *out, hipStream_t s);
// the flags of one table (version): buffers, and how many bricks came out flagged (copied back behind the kernel)
struct BrickSet {
  unsigned char *flags = nullptr;
  uint32_t *sat = nullptr;
  unsigned *d_count = nullptr, *h_count = nullptr;  // device word; pinned copy
  hipEvent_t counted = nullptr;
  size_t flags_cap = 0, sat_cap = 0;
  unsigned char *dil = nullptr;  // flags dilated by dil_r bricks per axis (perturbed fetch), made on demand
  size_t dil_cap = 0;
  int dil_r[3] = {-1, -1, -1};
  bool valid = false;
  float fill = -1.f;  // share of the bricks that are flagged; < 0: not known yet
};

struct smk_ctx {
  int device = 0;
  std::string err;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // ring of HIP-event pairs bracketing the render kernel of the last SMK_TIMING_RING frames,
  // recorded on the stream the kernel is launched on (bench.py's roofline leg reads them)
  std::vector<hipEvent_t> tev0, tev1;
  long long tcount = 0;

  // volume
  bool have_volume = false;
  int dtype = 0, nelts = 0, dmode = 0;
  int N[3] = {0, 0, 0};
  float fsize[3] = {1, 1, 1};
  int g0[3] = {0, 0, 0}, g1[3] = {0, 0, 0};  // this context's region
  int O[3] = {0, 0, 0}, D[3] = {0, 0, 0};
  int halo = 1;
  int region_on = 0;  // sub-box of renderVolume(.., xext, yext, zext): volume space
  float region_lo[3] = {0, 0, 0}, region_hi[3] = {0, 0, 0};
  int clip_axis = 0;  // orthogonal clip plane: 0 off, 1..6 = X+ X- Y+ Y- Z+ Z-
  float clip_vpos[3] = {0, 0, 0};
  int cplane_on = 0;  // free clip plane (glClipPlane), eye space
  double cplane_eye[4] = {0, 0, 0, 0};
  void *d_vox = nullptr;
  void *d_vox_x = nullptr;  // x-major copy [x][z][y] for views whose principal axis is x (lazy)
  float4 *d_brick_mm = nullptr;  // per brick of the stored box: range of the first two channels (smk_bricks.hip)
  int nbr[3] = {0, 0, 0};
  BrickSet br3;                        // brick flags under the dense 3-D table (the 2-D table's live in its versions)
  bool bricks3_dirty = true;
  std::string slab_why;     // why the last frame fell back to the gather kernel ("" if it did not)
  uint32_t *d_nrm = nullptr;
  bool have_normals = false;
  size_t vox_bytes = 0;

  // sharding
  int rank = 0, nranks = 1;

  // classification
  int tf_mode = -1;  // 0 1-D, 1 2-D, 2 3-D
  float4 *d_tlut = nullptr;
  int tlut_size = 0;
  std::vector<unsigned char> h_tf_vg, h_tf_h;
  unsigned char *d_tf_raw = nullptr;  // the table as it was set (the effective one = a 256-entry alpha map applied by a kernel)
  size_t tf_raw_cap = 0;
  bool tf_raw_stale = true, tf_raw_ev_valid = false;
  hipEvent_t tf_raw_ev = nullptr;     // the last kernel that read d_tf_raw
  hipStream_t tf_stream = nullptr;    // the table refresh runs here, beside the previous frame's ray-march
  hipEvent_t tf_ready = nullptr;      // ... and a frame's stream waits for this
  bool tf_ready_pending = false;
  uint32_t *d_tf_vg = nullptr, *d_tf_h = nullptr, *d_tf3d = nullptr;
  uint32_t *d_tf3d_occ = nullptr;  // occupancy of the dense 3-D table folded over its third axis (smk_set_tf3d)
  int tf3d_occ_roww = 0;
  uint32_t *d_tf_occ = nullptr;  // occupancy bitmap of the effective (V,G) table, tf_occ_roww words per row
  int tf_occ_roww = 0;
  // versions of the effective table + bitmap (d_tf_vg / d_tf_occ point into the current one): rebuilt without
  // stalling the frames in flight when the correction rate moves with the camera (smk_api.hip refresh_tf2d)
  struct TfVersion {
    unsigned char *d = nullptr, *h = nullptr;  // device copy; pinned staging
    BrickSet br;                                // this version's brick flags (smk_bricks.hip)
    size_t cap = 0;
    hipEvent_t copied = nullptr, used = nullptr;
    bool used_valid = false;
  } tfv[4];
  int tf_cur = -1;
  int sv = 0, sg = 0, s3v = 0, s3g = 0, s3h = 0;
  bool tf_dirty = true;
  float tf_rate_applied = -1.f;
  unsigned char tf_map_applied[256] = {0};  // the alpha-byte map the current effective table was made with
  bool tf_map_valid = false;

  // camera
  bool have_camera = false;
  double mv[16];
  float frustum[4], clip[2];
  int W = 0, H = 0;

  // sampling
  float sample_rate = 2.5f, gamma = 1.f;
  int steps = 0, scale_alphas = 1;

  // shading
  int blend = 0;  // smk_blend
  int shade = 0;
  float light_pos[3] = {0, 0, -5}, eye[3] = {0, 0, -7}, at[3] = {0, 0, 0};
  float xform[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float intens = .75f, amb = .05f;

  // half-angle-slicing shadows (smk_set_shadow)
  int shadow_on = 0, shadow_px = 1024;
  float shadow_q = .5f;
  float4 *d_light[2] = {nullptr, nullptr};  // ping-pong light buffers, light_cap texels each
  size_t light_cap = 0;
  int light_lb = 0;                         // the light buffer's edge in the last frame with shadows
  float4 *d_light_hist = nullptr;           // [nslices + 1][LB][LB]: every slice's light buffer (the two marches, smk_shadow.hip)
  size_t light_hist_cap = 0;                // texels
  const float4 *d_light_last = nullptr;     // the light buffer the last frame with shadows left (in d_light[] or the history)
  int opt_shadow_march = 1;                 // option "shadow_march": 1 = two marches (default), 0 = a launch per slice
  unsigned *d_shadow_barrier = nullptr;     // the fused shadow launch's grid-barrier counter

  // perturbation
  uint32_t *d_noise = nullptr;
  std::vector<unsigned char> h_noise;  // what d_noise holds (smk_set_perturb replaces the device copy only when the bytes change)
  int nn = 0;
  float pw[4] = {0, 0, 0, 0}, ps[4] = {0, 0, 0, 0};

  // scratch output for host-pointer renders
  float4 *d_out = nullptr;
  float *d_depth = nullptr;
  size_t out_cap = 0;

  // options / stats
  int opt_kernel = 0, opt_slab_T = 0, opt_tf_raw = 0, opt_tile = 0;
  int opt_slab_fly = 0;  // slices a loader keeps in flight (0 = default)
  int opt_slab_ns = 0;   // cap on the ring's slots (0 = as many as fit)
  int opt_bricks = 1;      // brick flags on (0: every slice is streamed and sampled, as before round 2's last step)
  int opt_slab_sched = 0;  // order of an XCD's tiles: 0 longest first, 1.. coarse weight classes + spatial blocks
  int opt_inject_status = 0;  // (test hook) the next slice-ring frame reports this status word
  int opt_wave_w = 8, opt_blk_w = 2, opt_lockstep = 1;
  SlabAux slab;  // slice-ring kernel side buffers
  ColsAux cols;  // column-stream kernel side buffers
  int opt_cols = 0;  // developer knobs of the column-stream kernel: shape | slots << 8 | chunk << 16 | (wstep + 1) << 28
  // auto mode (option kernel = 0) picks the ray-marcher by measurement: the first frames of a
  // new configuration run the slice-ring kernel, then the gather kernel (bit-identical frames),
  // and the faster one is kept for that configuration
  long long frame_id = 0;                  // frames enqueued so far (smk_last_frame_id)
  long long slab_failures = 0, slab_retries = 0;  // slice-ring frames flagged invalid / re-rendered by smk_render
  long long slab_lost = 0;  // ... of which flagged too late for anybody to be told (their status slot had been handed on)
  struct TuneEntry { int kernel; long long expires; };  // (a measured or forced choice is re-examined after a while)
  std::map<unsigned long long, TuneEntry> tune_choice;
  unsigned long long last_slab_sig = 0;  // configuration of the latest slice-ring launch (a failed one is not tried again)
  unsigned long long tune_sig = 0;
  int tune_state = 0, tune_slot[2] = {0, 0};
  long long tune_tcount = 0;  // frame count when the second timed trial was enqueued
  int last_kernel = 0;
  float last_ms = 0;
  double last_alg_bytes = 0;
};

// brick flags (smk_bricks.hip)
hipError_t smk_bricks_minmax(const void *vox, int dtype, const int D[3], const int nb[3], float4 *mm, hipStream_t s);
hipError_t smk_bricks_flags(const float4 *mm, const int nb[3], const uint32_t *occ, int roww, int sv, int sg, uint32_t *sat,
                            unsigned char *flags, unsigned *count, hipStream_t s);

// launchers (one translation unit per kernel family)
hipError_t smk_launch_gather(const RenderParams &P, int dtype, int tf_mode, int shade_kind,
                             hipStream_t s);
hipError_t smk_launch_count_inside(const RenderParams &P, unsigned long long *d_count, hipStream_t s);
// one launch per slice (smk_shadow.hip); L0 cleared by the caller
hipError_t smk_launch_shadow(const RenderParams &P, const smk_shadowcoef &sc, int dtype, int tf_mode, int shade_kind,
                             float4 *L0, float4 *L1, unsigned *barrier /* one device word for the fused launch's grid barrier, or null */, hipStream_t s);
hipError_t smk_launch_shadow_march(const RenderParams &P, const smk_shadowcoef &sc, int dtype, int tf_mode, float4 *hist, long long hstride,
                                   hipStream_t s);
// returns hipErrorNotSupported (and *why) when the frame must use the gather kernel
hipError_t smk_launch_slab(RenderParams P, int dtype, int tf_mode, int shade_kind, int opt_T, int opt_tile, int forced,
                           const void *vox_native, const void *vox_xmajor, SlabAux *aux, const char **why,
                           hipStream_t s);
// the column-stream kernel (smk_cols.hip); same convention as smk_launch_slab
hipError_t smk_launch_cols(RenderParams P, int dtype, int tf_mode, int shade_kind, int knobs, const void *vox_native, ColsAux *aux,
                           int *status_word, const char **why, hipStream_t s);

// a few host threads for the per-frame planning (smk_api.hip): run(n, f) calls f(0) .. f(n - 1), f(0) on the caller
#include <functional>
int smk_host_pool_size();
void smk_host_pool_run(int n, const std::function<void(int)> &f);
