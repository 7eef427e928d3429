// smk_api.hip -- the C ABI (include/smk.h): context, HBM layout, host-side setup, launches.
// Product code: no CPU rendering path exists here; without a HIP device every entry fails.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "smk_device.h"

static std::string g_create_err;

// ------------------------------------------------------------------------------- host thread pool
// Planning a frame (the slice-ring launcher's scan of every tile's ray bundle) is double arithmetic on the host's thread:
// a pool of a few workers shares it.  One pool per process; contexts are used from one thread at a time (smk.h), a mutex
// keeps two contexts on different threads from interleaving jobs.
#include <condition_variable>
#include <mutex>
#include <thread>
namespace {
struct HostPool {
  std::vector<std::thread> workers;
  std::mutex m, use;
  std::condition_variable cv, done_cv;
  const std::function<void(int)> *job = nullptr;
  int njobs = 0, next = 0, pending = 0;
  unsigned long long gen = 0;
  bool quit = false;
  explicit HostPool(int n) {
    for (int i = 0; i < n; ++i)
      workers.emplace_back([this] {
        unsigned long long seen = 0;
        for (;;) {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [&] { return quit || (gen != seen && next < njobs); });
          if (quit) return;
          while (next < njobs) {
            const int k = next++;
            lk.unlock();
            (*job)(k);
            lk.lock();
            if (--pending == 0) done_cv.notify_all();
          }
          seen = gen;
        }
      });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
    }
    cv.notify_all();
    for (std::thread &t : workers) t.join();
  }
};
HostPool *host_pool() {
  static HostPool *pool = [] {
    const char *e = getenv("SMK_HOST_THREADS");
    int n = e ? atoi(e) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 2));
    n = std::max(1, std::min(n, 32));
    return new HostPool(n - 1);  // (the caller is the n-th; never destroyed: workers must not outlive a static's destructor order)
  }();
  return pool;
}
}  // namespace

int smk_host_pool_size() { return (int)host_pool()->workers.size() + 1; }

void smk_host_pool_run(int n, const std::function<void(int)> &f) {
  HostPool *p = host_pool();
  if (n <= 1 || p->workers.empty()) {
    for (int k = 0; k < n; ++k) f(k);
    return;
  }
  std::lock_guard<std::mutex> one(p->use);
  {
    std::lock_guard<std::mutex> lk(p->m);
    p->job = &f;
    p->njobs = n;
    p->next = 1;        // job 0 is the caller's
    p->pending = n - 1;
    ++p->gen;
  }
  p->cv.notify_all();
  f(0);
  std::unique_lock<std::mutex> lk(p->m);
  // (help with what is left rather than wait for a worker to wake up)
  while (p->next < p->njobs) {
    const int k = p->next++;
    lk.unlock();
    f(k);
    lk.lock();
    --p->pending;
  }
  p->done_cv.wait(lk, [&] { return p->pending == 0; });
  p->job = nullptr;
  p->njobs = 0;
}

#define HIPCHK(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      char b_[512];                                                                    \
      snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      (ctx)->err = b_;                                                                 \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

#define FAIL(ctx, ...)                     \
  do {                                     \
    char b_[512];                          \
    snprintf(b_, sizeof b_, __VA_ARGS__);  \
    (ctx)->err = b_;                       \
    return 1;                              \
  } while (0)

// ------------------------------------------------------------------------------- lifecycle

extern "C" smk_ctx *smk_create(int device_ordinal, int *err) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_err = "smk_create: no HIP device available (this library has no CPU fallback)";
    if (err) *err = 1;
    return nullptr;
  }
  if (device_ordinal < 0 || device_ordinal >= n) {
    g_create_err = "smk_create: device ordinal out of range";
    if (err) *err = 2;
    return nullptr;
  }
  if (hipSetDevice(device_ordinal) != hipSuccess) {
    g_create_err = "smk_create: hipSetDevice failed";
    if (err) *err = 3;
    return nullptr;
  }
  smk_ctx *c = new smk_ctx();
  c->device = device_ordinal;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      false) {
    g_create_err = "smk_create: stream/event creation failed";
    delete c;
    if (err) *err = 4;
    return nullptr;
  }
  if (err) *err = 0;
  return c;
}

static void free_brick_set(BrickSet &B) {
  if (B.flags) (void)hipFree(B.flags);
  if (B.dil) (void)hipFree(B.dil);
  if (B.sat) (void)hipFree(B.sat);
  if (B.d_count) (void)hipFree(B.d_count);
  if (B.h_count) (void)hipHostFree(B.h_count);
  if (B.counted) (void)hipEventDestroy(B.counted);
  B = BrickSet();
}

static void free_volume(smk_ctx *c) {
  if (c->d_vox) (void)hipFree(c->d_vox);
  if (c->d_nrm) (void)hipFree(c->d_nrm);
  if (c->d_vox_x) (void)hipFree(c->d_vox_x);
  if (c->d_brick_mm) (void)hipFree(c->d_brick_mm);
  smk_cols_drop_layouts(&c->cols);  // (built from this volume)
  c->d_brick_mm = nullptr;
  c->d_vox_x = nullptr;
  c->d_vox = nullptr;
  c->d_nrm = nullptr;
  c->have_volume = false;
}

extern "C" void smk_destroy(smk_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  free_volume(c);
  for (smk_ctx::TfVersion &T : c->tfv) {
    if (T.d) (void)hipFree(T.d);
    if (T.h) (void)hipHostFree(T.h);
    free_brick_set(T.br);
    if (T.copied) (void)hipEventDestroy(T.copied);
    if (T.used) (void)hipEventDestroy(T.used);
  }
  if (c->tf_raw_ev) (void)hipEventDestroy(c->tf_raw_ev);
  if (c->tf_ready) (void)hipEventDestroy(c->tf_ready);
  if (c->tf_stream) { (void)hipStreamSynchronize(c->tf_stream); (void)hipStreamDestroy(c->tf_stream); }
  free_brick_set(c->br3);
  smk_cols_free(&c->cols);
  void *ptrs[] = {c->d_light_hist, c->d_shadow_barrier, c->d_tf_raw, c->d_tlut, c->d_tf_h, c->d_tf3d, c->d_tf3d_occ, c->d_noise, c->d_out, c->d_depth, c->d_light[0], c->d_light[1]};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (c->slab.h_status) (void)hipHostFree(c->slab.h_status);
  if (c->slab.d_diag) (void)hipFree(c->slab.d_diag);
  if (c->slab.d_order) (void)hipFree(c->slab.d_order);
  if (c->slab.d_pticks) (void)hipFree(c->slab.d_pticks);
  if (c->slab.h_pticks) (void)hipHostFree(c->slab.h_pticks);
  if (c->slab.d_seg) (void)hipFree(c->slab.d_seg);
  for (int k = 0; k < 4; ++k) {
    if (c->slab.h_order[k]) (void)hipHostFree(c->slab.h_order[k]);
    if (c->slab.order_ev[k]) (void)hipEventDestroy(c->slab.order_ev[k]);
  }
  if (c->slab.d_trace) (void)hipFree(c->slab.d_trace);
  if (c->slab.d_ticks) (void)hipFree(c->slab.d_ticks);
  if (c->slab.h_ticks) (void)hipHostFree(c->slab.h_ticks);
  if (c->slab.ticks_ev) (void)hipEventDestroy(c->slab.ticks_ev);
  for (hipEvent_t e : c->tev0) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->tev1) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" const char *smk_last_error(smk_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

// ------------------------------------------------------------------------------- volume upload

// source voxel (brick-local, caller layout [z][y][x][nelts] + optional [z][y][x][3] normals)
// -> packed voxel in the dense region+halo box.  One thread per source voxel of the chunk.
template <int DT>
__global__ void smk_k_pack(const void *src, const unsigned char *grad, int bx, int by, int cz, int nelts,
                           int gx0, int gy0, int gz0,  // global index of the chunk's first voxel
                           int Ox, int Oy, int Oz, int Dx, int Dy, int Dz, void *dst, uint32_t *nrm, int n_in_w) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)bx * by * cz;
  if (t >= total) return;
  int x = (int)(t % bx);
  int y = (int)((t / bx) % by);
  int z = (int)(t / ((size_t)bx * by));
  int X = gx0 + x - Ox, Y = gy0 + y - Oy, Z = gz0 + z - Oz;
  if (X < 0 || X >= Dx || Y < 0 || Y >= Dy || Z < 0 || Z >= Dz) return;
  size_t o = ((size_t)Z * Dy + Y) * Dx + X;
  uint32_t nb = 0x00808080u;
  if (grad) nb = (uint32_t)grad[t * 3] | ((uint32_t)grad[t * 3 + 1] << 8) | ((uint32_t)grad[t * 3 + 2] << 16);
  if (DT == 0) {
    const unsigned char *s = (const unsigned char *)src + t * nelts;
    uint32_t d = 0;
    for (int e = 0; e < nelts; ++e) d |= (uint32_t)s[e] << (8 * e);
    ((uint2 *)dst)[o] = make_uint2(d, nb);
  } else {
    const float *s = (const float *)src + t * nelts;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    v.x = s[0];
    if (nelts > 1) v.y = s[1];
    if (nelts > 2) v.z = s[2];
    if (n_in_w) v.w = __uint_as_float(nb);
    else {
      v.w = s[3];
      if (nrm) nrm[o] = nb;
    }
    ((float4 *)dst)[o] = v;
  }
}

static void shard_region(const smk_ctx *c, int g0[3], int g1[3]) {
  for (int a = 0; a < 3; ++a) {
    g0[a] = 0;
    g1[a] = c->N[a];
  }
  int bit = 0;
  for (int n = c->nranks; n > 1; n >>= 1, ++bit) {
    int a = bit % 3, half = c->N[a] / 2;  // bit 0 splits x, bit 1 y, bit 2 z
    if ((c->rank >> bit) & 1) g0[a] = std::max(g0[a], half);
    else g1[a] = std::min(g1[a], half);
  }
}

static int upload_impl(smk_ctx *c, const smk_volume_desc *b, int nb, int nelts, smk_dtype dtype, smk_datamode dmode,
                       bool on_device) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (!b || nb <= 0) FAIL(c, "smk_upload_volume: no bricks");
  if (nelts < 1 || nelts > 4) FAIL(c, "smk_upload_volume: nelts %d not in 1..4", nelts);
  if (dtype != SMK_U8 && dtype != SMK_F32) FAIL(c, "smk_upload_volume: bad dtype");
  // whole-volume dims and extent from the brick list (MetaVolume::brick keeps iPos/fPos)
  int N[3] = {0, 0, 0};
  float fs[3] = {0, 0, 0};
  bool any_grad = false, all_grad = true;
  for (int i = 0; i < nb; ++i) {
    if (!b[i].data) FAIL(c, "smk_upload_volume: brick %d has no data", i);
    if (b[i].xiSize <= 0 || b[i].yiSize <= 0 || b[i].ziSize <= 0) FAIL(c, "smk_upload_volume: brick %d empty", i);
    if (b[i].xiPos < 0 || b[i].yiPos < 0 || b[i].ziPos < 0) FAIL(c, "smk_upload_volume: brick %d negative origin", i);
    N[0] = std::max(N[0], b[i].xiPos + b[i].xiSize);
    N[1] = std::max(N[1], b[i].yiPos + b[i].yiSize);
    N[2] = std::max(N[2], b[i].ziPos + b[i].ziSize);
    fs[0] = std::max(fs[0], b[i].xfPos + b[i].xfSize);
    fs[1] = std::max(fs[1], b[i].yfPos + b[i].yfSize);
    fs[2] = std::max(fs[2], b[i].zfPos + b[i].zfSize);
    any_grad |= b[i].grad != nullptr;
    all_grad &= b[i].grad != nullptr;
  }
  if (any_grad && !all_grad) FAIL(c, "smk_upload_volume: some bricks have normals and some do not");
  size_t vox = 0;
  for (int i = 0; i < nb; ++i) vox += (size_t)b[i].xiSize * b[i].yiSize * b[i].ziSize;
  if (vox != (size_t)N[0] * N[1] * N[2]) FAIL(c, "smk_upload_volume: bricks do not tile the %dx%dx%d volume", N[0], N[1], N[2]);
  if (!(fs[0] > 0 && fs[1] > 0 && fs[2] > 0)) FAIL(c, "smk_upload_volume: non-positive extent");

  free_volume(c);
  c->dtype = dtype;
  c->nelts = nelts;
  c->dmode = dmode;
  for (int a = 0; a < 3; ++a) {
    c->N[a] = N[a];
    c->fsize[a] = fs[a];
  }
  if (c->nranks > 1)
    for (int a = 0; a < 3; ++a)
      if (N[a] < 2) FAIL(c, "smk_upload_volume: volume too thin to shard");
  shard_region(c, c->g0, c->g1);
  for (int a = 0; a < 3; ++a) {
    int lo = std::max(c->g0[a] - c->halo, 0), hi = std::min(c->g1[a] + c->halo, N[a]);
    c->O[a] = lo;
    c->D[a] = hi - lo;
  }
  // 8-byte voxels travel to LDS in 16-byte units: the slice-ring kernel wants even row lengths
  // along both axes that can be its contiguous one (x, and y in the x-major copy) -- one more halo
  // voxel where the volume has one, else a pad column nobody samples (index N: the texel pair of a
  // clamped coordinate ends at N-1)
  bool padded = false;
  if (dtype == SMK_U8)
    for (int a = 0; a < 2; ++a)
      if (c->D[a] & 1) {
        if (c->O[a] + c->D[a] < N[a]) ++c->D[a];
        else if (c->O[a] > 0) { --c->O[a]; ++c->D[a]; }
        else { ++c->D[a]; padded = true; }
      }
  size_t nst = (size_t)c->D[0] * c->D[1] * c->D[2];
  size_t vb = dtype == SMK_U8 ? 8 : 16;
  HIPCHK(c, hipMalloc(&c->d_vox, nst * vb + 16));  // (+16: the gather kernel reads u8 voxels in pairs, smk_load_pair_u8)
  if (padded) HIPCHK(c, hipMemset(c->d_vox, 0, nst * vb));
  c->vox_bytes = nst * vb;
  bool n_in_w = dtype == SMK_F32 && nelts <= 3;
  if (dtype == SMK_F32 && nelts == 4 && any_grad) HIPCHK(c, hipMalloc((void **)&c->d_nrm, nst * 4));
  c->have_normals = any_grad;

  const size_t esz = dtype == SMK_U8 ? 1 : 4;
  void *d_stage = nullptr;
  unsigned char *d_gstage = nullptr;
  const size_t chunk_budget = (size_t)256 << 20;
  for (int i = 0; i < nb; ++i) {
    const smk_volume_desc &k = b[i];
    // skip bricks that cannot touch this context's stored box
    if (k.xiPos >= c->O[0] + c->D[0] || k.xiPos + k.xiSize <= c->O[0] || k.yiPos >= c->O[1] + c->D[1] ||
        k.yiPos + k.yiSize <= c->O[1] || k.ziPos >= c->O[2] + c->D[2] || k.ziPos + k.ziSize <= c->O[2])
      continue;
    size_t slice = (size_t)k.xiSize * k.yiSize;
    int zper = on_device ? k.ziSize : (int)std::max<size_t>(1, chunk_budget / (slice * nelts * esz));
    for (int z0 = 0; z0 < k.ziSize; z0 += zper) {
      int cz = std::min(zper, k.ziSize - z0);
      // only the z range that intersects the stored box
      if (k.ziPos + z0 >= c->O[2] + c->D[2] || k.ziPos + z0 + cz <= c->O[2]) continue;
      const void *src;
      const unsigned char *gsrc = nullptr;
      size_t off = (size_t)z0 * slice;
      if (on_device) {
        src = (const char *)k.data + off * nelts * esz;
        if (k.grad) gsrc = k.grad + off * 3;
      } else {
        if (!d_stage) HIPCHK(c, hipMalloc(&d_stage, std::min(chunk_budget + slice * nelts * esz, (size_t)-1)));
        size_t bytes = (size_t)cz * slice * nelts * esz;
        if (bytes > chunk_budget + slice * nelts * esz) FAIL(c, "smk_upload_volume: internal staging overflow");
        HIPCHK(c, hipMemcpy(d_stage, (const char *)k.data + off * nelts * esz, bytes, hipMemcpyHostToDevice));
        src = d_stage;
        if (k.grad) {
          if (!d_gstage) HIPCHK(c, hipMalloc((void **)&d_gstage, ((size_t)zper + 1) * slice * 3));
          HIPCHK(c, hipMemcpy(d_gstage, k.grad + off * 3, (size_t)cz * slice * 3, hipMemcpyHostToDevice));
          gsrc = d_gstage;
        }
      }
      size_t total = (size_t)cz * slice;
      unsigned blocks = (unsigned)((total + 255) / 256);
      if (dtype == SMK_U8)
        hipLaunchKernelGGL(smk_k_pack<0>, dim3(blocks), dim3(256), 0, 0, src, gsrc, k.xiSize, k.yiSize, cz, nelts,
                           k.xiPos, k.yiPos, k.ziPos + z0, c->O[0], c->O[1], c->O[2], c->D[0], c->D[1], c->D[2],
                           c->d_vox, c->d_nrm, 0);
      else
        hipLaunchKernelGGL(smk_k_pack<1>, dim3(blocks), dim3(256), 0, 0, src, gsrc, k.xiSize, k.yiSize, cz, nelts,
                           k.xiPos, k.yiPos, k.ziPos + z0, c->O[0], c->O[1], c->O[2], c->D[0], c->D[1], c->D[2],
                           c->d_vox, c->d_nrm, n_in_w ? 1 : 0);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipDeviceSynchronize());
    }
  }
  if (d_stage) (void)hipFree(d_stage);
  if (d_gstage) (void)hipFree(d_gstage);
  // value ranges of the 8x8x8-cell bricks (smk_bricks.hip): what the brick flags of every later table are made from
  for (int a = 0; a < 3; ++a) c->nbr[a] = (c->D[a] - 1) / (1 << SMK_BRICK_LOG2) + 1;
  HIPCHK(c, hipMalloc((void **)&c->d_brick_mm, (size_t)c->nbr[0] * c->nbr[1] * c->nbr[2] * sizeof(float4)));
  HIPCHK(c, smk_bricks_minmax(c->d_vox, dtype == SMK_U8 ? 0 : 1, c->D, c->nbr, c->d_brick_mm, 0));
  HIPCHK(c, hipDeviceSynchronize());
  c->bricks3_dirty = true;
  c->have_volume = true;
  c->tune_choice.clear();  // a new volume: the kernels' relative speed is measured afresh
  c->tune_sig = 0;
  c->tf_dirty = true;
  return 0;
}

extern "C" int smk_upload_volume(smk_ctx *c, const smk_volume_desc *b, int nb, int nelts, smk_dtype dt,
                                 smk_datamode dm) {
  return upload_impl(c, b, nb, nelts, dt, dm, false);
}
extern "C" int smk_upload_volume_device(smk_ctx *c, const smk_volume_desc *b, int nb, int nelts, smk_dtype dt,
                                        smk_datamode dm) {
  return upload_impl(c, b, nb, nelts, dt, dm, true);
}

extern "C" int smk_set_clip(smk_ctx *c, int on, int oaxis, const float *vpos) {
  if (!c) return 1;
  if (!on) {
    c->clip_axis = 0;
    return 0;
  }
  if (oaxis < 1 || oaxis > 6 || !vpos) FAIL(c, "smk_set_clip: axis must be 1..6 (X+ X- Y+ Y- Z+ Z-) and vpos given");
  c->clip_axis = oaxis;
  for (int a = 0; a < 3; ++a) c->clip_vpos[a] = vpos[a];
  return 0;
}

extern "C" int smk_set_region(smk_ctx *c, int on, const float *lo, const float *hi) {
  if (!c) return 1;
  if (!on) {
    c->region_on = 0;
    return 0;
  }
  if (!lo || !hi) FAIL(c, "smk_set_region: null extents");
  for (int a = 0; a < 3; ++a) {
    c->region_lo[a] = lo[a] < hi[a] ? lo[a] : hi[a];
    c->region_hi[a] = lo[a] < hi[a] ? hi[a] : lo[a];
  }
  c->region_on = 1;
  return 0;
}

extern "C" int smk_set_clip_plane(smk_ctx *c, int on, const double *plane_eye) {
  if (!c) return 1;
  if (!on) {
    c->cplane_on = 0;
    return 0;
  }
  if (!plane_eye) FAIL(c, "smk_set_clip_plane: plane missing");
  c->cplane_on = 1;
  for (int k = 0; k < 4; ++k) c->cplane_eye[k] = plane_eye[k];
  return 0;
}

extern "C" int smk_set_shard(smk_ctx *c, int rank, int nranks) {
  if (!c) return 1;
  if (!(nranks == 1 || nranks == 2 || nranks == 4 || nranks == 8)) FAIL(c, "smk_set_shard: nranks must be 1,2,4 or 8");
  if (rank < 0 || rank >= nranks) FAIL(c, "smk_set_shard: rank out of range");
  if (c->have_volume) FAIL(c, "smk_set_shard: must be called before smk_upload_volume");
  c->rank = rank;
  c->nranks = nranks;
  return 0;
}

// ------------------------------------------------------------------------------- classification

template <class T>
static int dev_replace(smk_ctx *c, T **dptr, const void *host, size_t bytes) {
  if (*dptr) (void)hipFree(*dptr);
  *dptr = nullptr;
  HIPCHK(c, hipMalloc((void **)dptr, bytes));
  HIPCHK(c, hipMemcpy(*dptr, host, bytes, hipMemcpyHostToDevice));
  return 0;
}

extern "C" int smk_set_tlut1d(smk_ctx *c, const float *rgba, int size) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (!rgba || size < 2) FAIL(c, "smk_set_tlut1d: bad table");
  // TLUT::loadTransferTableRGBA (TLUT.cpp:65-71): theTable = (r*a, g*a, b*a, a)
  std::vector<float> t((size_t)size * 4);
  for (int n = 0; n < size; ++n) {
    float a = rgba[n * 4 + 3];
    t[n * 4 + 0] = rgba[n * 4 + 0] * a;
    t[n * 4 + 1] = rgba[n * 4 + 1] * a;
    t[n * 4 + 2] = rgba[n * 4 + 2] * a;
    t[n * 4 + 3] = a;
  }
  if (dev_replace(c, &c->d_tlut, t.data(), t.size() * 4)) return 1;
  c->tlut_size = size;
  c->tf_mode = 0;
  return 0;
}

extern "C" int smk_set_tf2d(smk_ctx *c, const unsigned char *deptex, const unsigned char *deptex2, int sv, int sg) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (!deptex || sv < 2 || sg < 1) FAIL(c, "smk_set_tf2d: bad table");
  size_t bytes = (size_t)sv * sg * 4;
  c->h_tf_vg.assign(deptex, deptex + bytes);
  c->tf_raw_stale = true;
  if (deptex2) {
    c->h_tf_h.assign(deptex2, deptex2 + bytes);
    if (dev_replace(c, &c->d_tf_h, deptex2, bytes)) return 1;
  } else {
    c->h_tf_h.clear();
    if (c->d_tf_h) (void)hipFree(c->d_tf_h);
    c->d_tf_h = nullptr;
  }
  c->sv = sv;
  c->sg = sg;
  c->tf_mode = 1;
  c->tf_dirty = true;
  return 0;
}

extern "C" int smk_set_tf3d(smk_ctx *c, const unsigned char *ptex, int sv, int sg, int sh) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (!ptex || sv < 1 || sg < 1 || sh < 1) FAIL(c, "smk_set_tf3d: bad table");
  if (dev_replace(c, &c->d_tf3d, ptex, (size_t)sv * sg * sh * 4)) return 1;
  // occupancy of the dense table, folded over its third axis: bit (t, s) is set when ANY sheet has a non-zero alpha in
  // the 2 x 2 texel quad based at (s, t).  A clear bit means the trilinear lookup of every sample whose (v, g) base texel
  // is (s, t) returns alpha == 0 exactly, whatever its third coordinate -- the kernels skip the eight-texel gather then
  // (the reference's widgets paint a few regions of a 256 x 256 x 4 table: most of it is transparent)
  {
    const int roww = (sv + 31) / 32;
    std::vector<uint32_t> occ((size_t)roww * sg, 0u);
    for (int t = 0; t < sg; ++t)
      for (int sx = 0; sx < sv; ++sx) {
        const int s1 = std::min(sx + 1, sv - 1), t1 = std::min(t + 1, sg - 1);
        unsigned any = 0;
        for (int h = 0; h < sh && !any; ++h) {
          const unsigned char *e = ptex + (size_t)h * sg * sv * 4;
          any = e[((size_t)t * sv + sx) * 4 + 3] | e[((size_t)t * sv + s1) * 4 + 3] | e[((size_t)t1 * sv + sx) * 4 + 3] | e[((size_t)t1 * sv + s1) * 4 + 3];
        }
        if (any) occ[(size_t)t * roww + (sx >> 5)] |= 1u << (sx & 31);
      }
    if (dev_replace(c, &c->d_tf3d_occ, occ.data(), occ.size() * 4)) return 1;
    c->tf3d_occ_roww = roww;
  }
  c->s3v = sv;
  c->s3g = sg;
  c->s3h = sh;
  c->tf_mode = 2;
  c->bricks3_dirty = true;
  return 0;
}

// ------------------------------------------------------------------------------- camera etc.

extern "C" int smk_set_camera(smk_ctx *c, const double mv[16], const float fr[4], const float clip[2], int w, int h) {
  if (!c) return 1;
  if (!mv || !fr || !clip) FAIL(c, "smk_set_camera: null argument");
  if (w <= 0 || h <= 0) FAIL(c, "smk_set_camera: bad window %dx%d", w, h);
  if (!(clip[0] > 0)) FAIL(c, "smk_set_camera: near plane must be > 0");
  if (!(fr[1] > fr[0]) || !(fr[3] > fr[2])) FAIL(c, "smk_set_camera: degenerate frustum");
  memcpy(c->mv, mv, sizeof c->mv);
  memcpy(c->frustum, fr, sizeof c->frustum);
  memcpy(c->clip, clip, sizeof c->clip);
  c->W = w;
  c->H = h;
  c->have_camera = true;  // (steps-mode opacity correction follows the view-depth extent: refresh_tf2d compares the rate itself)
  return 0;
}

extern "C" int smk_set_shading(smk_ctx *c, smk_shade mode, const float lp[3], const float eye[3], const float at[3],
                               const float xf[16], float intens, float amb) {
  if (!c) return 1;
  if (mode < SMK_SHADE_NONE || mode > SMK_SHADE_NV20_DSPEC) FAIL(c, "smk_set_shading: bad mode");
  if (!lp || !eye || !at || !xf) FAIL(c, "smk_set_shading: null argument");
  c->shade = mode;
  memcpy(c->light_pos, lp, 12);
  memcpy(c->eye, eye, 12);
  memcpy(c->at, at, 12);
  memcpy(c->xform, xf, 64);
  c->intens = intens;
  c->amb = amb;
  return 0;
}

extern "C" int smk_set_sampling(smk_ctx *c, float rate, int steps, float gamma, int scale_alphas) {
  if (!c) return 1;
  if (steps <= 0 && !(rate > 0)) FAIL(c, "smk_set_sampling: need sample_rate > 0 or steps > 0");
  if (!(gamma > 0)) FAIL(c, "smk_set_sampling: gamma must be > 0");
  c->sample_rate = rate;
  c->steps = steps > 0 ? steps : 0;
  c->gamma = gamma;
  c->scale_alphas = scale_alphas;
  c->tf_dirty = true;
  return 0;
}

extern "C" int smk_set_blend(smk_ctx *c, smk_blend mode) {
  if (!c) return 1;
  if (mode < SMK_BLEND_FRONT_TO_BACK || mode > SMK_BLEND_MAX) FAIL(c, "smk_set_blend: bad mode");
  c->blend = mode;
  return 0;
}

extern "C" int smk_set_perturb(smk_ctx *c, const unsigned char *noise, int n, const float w[4], const float s[4]) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (!noise || !w || !s || (w[0] == 0 && w[1] == 0)) {
    c->pw[0] = c->pw[1] = 0;
    return 0;
  }
  if (n < 1) FAIL(c, "smk_set_perturb: bad noise size");
  // a caller that sets the perturbation every frame (the adapter's draw()) hands over the same texture each time: the
  // device copy is replaced -- hipFree, hipMalloc and a synchronous copy, i.e. a device synchronisation -- only when the
  // bytes differ; weights and scales alone cost nothing
  const size_t nbytes = (size_t)n * n * n * 4;
  if (!(c->d_noise && c->nn == n && c->h_noise.size() == nbytes && !memcmp(c->h_noise.data(), noise, nbytes))) {
    if (dev_replace(c, &c->d_noise, noise, nbytes)) return 1;
    c->h_noise.assign(noise, noise + nbytes);
  }
  c->nn = n;
  memcpy(c->pw, w, 16);
  memcpy(c->ps, s, 16);
  return 0;
}

// ------------------------------------------------------------------------------- host setup

// VolumeRenderer::inverseMatrix (VolumeRenderer.cpp:1096-1131), affine, double
static void inverse_affine(double inv[16], const double m[16]) {
  double det = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[1] * m[4] * m[10] + m[1] * m[6] * m[8] +
               m[2] * m[4] * m[9] - m[2] * m[5] * m[8];
  inv[0] = (m[5] * m[10] - m[6] * m[9]) / det;
  inv[1] = (-m[1] * m[10] + m[2] * m[9]) / det;
  inv[2] = (m[1] * m[6] - m[2] * m[5]) / det;
  inv[3] = 0.0;
  inv[4] = (-m[4] * m[10] + m[6] * m[8]) / det;
  inv[5] = (m[0] * m[10] - m[2] * m[8]) / det;
  inv[6] = (-m[0] * m[6] + m[2] * m[4]) / det;
  inv[7] = 0.0;
  inv[8] = (m[4] * m[9] - m[5] * m[8]) / det;
  inv[9] = (-m[0] * m[9] + m[1] * m[8]) / det;
  inv[10] = (m[0] * m[5] - m[1] * m[4]) / det;
  inv[11] = 0.0;
  inv[12] = -(inv[0] * m[12] + inv[4] * m[13] + inv[8] * m[14]);
  inv[13] = -(inv[1] * m[12] + inv[5] * m[13] + inv[9] * m[14]);
  inv[14] = -(inv[2] * m[12] + inv[6] * m[13] + inv[10] * m[14]);
  inv[15] = 1.0;
}

// Sample placement of VolumeRenderer::render3DVA (VolumeRenderer.cpp:521-611) as ray
// coefficients: planes z_k = zmin + k*dis (k=1..S) perpendicular to view z, global for the whole
// volume (R8kVolRen3D.cpp:1331-1351); voxel coordinate of plane m (front to back) on the ray
// through pixel (i,j) is fma(m, B_a, A_a) with A,B affine in the pixel's frustum coordinates.
static int compute_raycoef(smk_ctx *c, smk_raycoef *o, double inv[16]) {
  inverse_affine(inv, c->mv);
  const double *M = c->mv;
  double f[3] = {c->fsize[0], c->fsize[1], c->fsize[2]};
  double zmin = 1e300, zmax = -1e300;
  for (int i = 0; i < 8; ++i) {
    double x = (i & 1) ? f[0] : 0, y = (i & 2) ? f[1] : 0, z = (i & 4) ? f[2] : 0;
    double zz = M[2] * x + M[6] * y + M[10] * z + M[14];
    if (zz < zmin) zmin = zz;
    if (zz > zmax) zmax = zz;
  }
  double dist = zmax - zmin, dis;
  int S;
  if (c->steps > 0) {
    S = c->steps;
    dis = dist / S;
  } else {
    float disf = c->fsize[0] / ((float)c->N[0] * c->sample_rate);  // VolumeRenderer.cpp:595
    dis = disf;
    S = (int)(dist / dis);  // :598
  }
  if (S < 0) S = 0;
  double n = c->clip[0];
  double z0 = zmin + S * dis;
  double tau0 = -z0 / n, dtau = dis / n;
  double l = c->frustum[0], r = c->frustum[1], b = c->frustum[2], t = c->frustum[3];
  o->pxs = (float)((r - l) / c->W);
  o->pxl = (float)l;
  o->pys = (float)((t - b) / c->H);
  o->pyl = (float)b;
  double N[3] = {(double)c->N[0], (double)c->N[1], (double)c->N[2]};
  for (int a = 0; a < 3; ++a) {
    double s = N[a] / f[a];
    double R0 = inv[0 + a], R1 = inv[4 + a], R2 = inv[8 + a], e = inv[12 + a];
    o->Ac[a] = (float)((e - tau0 * n * R2) * s - 0.5);
    o->Ax[a] = (float)(tau0 * R0 * s);
    o->Ay[a] = (float)(tau0 * R1 * s);
    o->Bc[a] = (float)(-dtau * n * R2 * s);
    o->Bx[a] = (float)(dtau * R0 * s);
    o->By[a] = (float)(dtau * R1 * s);
  }
  o->nplanes = S;
  o->tau0 = (float)tau0;
  o->dtau = (float)dtau;
  o->zmin = (float)zmin;
  o->zmax = (float)zmax;
  o->dis = (float)dis;
  return 0;
}

extern "C" int smk_get_raycoef(smk_ctx *c, smk_raycoef *out) {
  if (!c || !out) return 1;
  if (!c->have_volume || !c->have_camera) FAIL(c, "smk_get_raycoef: volume and camera must be set");
  double inv[16];
  return compute_raycoef(c, out, inv);
}

static double dot3d(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// Half-angle slicing set-up (R8kVolRen3D.cpp:296-326; light transform LTWidgetRen.cpp:231-291; light-buffer
// coordinates R8kVolRen3D.cpp:1664-1676), everything in double, rounded once: slice planes sn . X = tmin + k dc
// in model space, eye rays X = e + tau (R0 px + R1 py - n R2), light rays from the apex of the light's
// projection, and the model -> light-buffer map.  The CPU checker (orc_shadow_setup) does the same steps.
static int compute_shadowcoef(smk_ctx *c, smk_shadowcoef *o) {
  memset(o, 0, sizeof *o);
  const double f[3] = {c->fsize[0], c->fsize[1], c->fsize[2]}, N[3] = {(double)c->N[0], (double)c->N[1], (double)c->N[2]};
  double vd[3] = {(double)c->at[0] - c->eye[0], (double)c->at[1] - c->eye[1], (double)c->at[2] - c->eye[2]};
  double ld[3] = {-(double)c->light_pos[0], -(double)c->light_pos[1], -(double)c->light_pos[2]};
  const double vl = sqrt(dot3d(vd, vd)), d0 = sqrt(dot3d(ld, ld));
  if (!(vl > 0) || !(d0 > 0)) FAIL(c, "smk_render: shadows need eye != at and a light away from the origin");
  for (int k = 0; k < 3; ++k) {
    vd[k] /= vl;
    ld[k] /= d0;
  }
  const double vdl = dot3d(vd, ld);
  if (vdl <= 0)
    for (int k = 0; k < 3; ++k) vd[k] = -vd[k];
  double h[3];
  for (int k = 0; k < 3; ++k) h[k] = (vd[k] - ld[k]) * .5 + ld[k];
  o->front_to_back = vdl > 0;
  double xf[16], xinv[16];
  for (int i = 0; i < 16; ++i) xf[i] = c->xform[i];
  inverse_affine(xinv, xf);
  double sn[3];
  for (int a = 0; a < 3; ++a) sn[a] = xinv[0 + a] * h[0] + xinv[4 + a] * h[1] + xinv[8 + a] * h[2];
  const double snl = sqrt(dot3d(sn, sn));
  if (!(snl > 0)) FAIL(c, "smk_render: shadows: degenerate half-way vector");
  for (int a = 0; a < 3; ++a) sn[a] /= snl;
  double tmin = 1e300, tmax = -1e300;
  for (int i = 0; i < 8; ++i) {
    const double X[3] = {(i & 1) ? f[0] : 0, (i & 2) ? f[1] : 0, (i & 4) ? f[2] : 0};
    const double t = dot3d(sn, X);
    if (t < tmin) tmin = t;
    if (t > tmax) tmax = t;
  }
  double dc;
  int S;
  if (c->steps > 0) {
    S = c->steps;
    dc = (tmax - tmin) / S;
  } else {
    const float disf = c->fsize[0] / ((float)c->N[0] * c->sample_rate);  // R8kVolRen3D.cpp:1330
    dc = disf;
    S = (int)((tmax - tmin) / dc);
  }
  if (S < 0) S = 0;
  o->nslices = S;
  double inv[16];
  inverse_affine(inv, c->mv);
  const double n = c->clip[0];
  const double l = c->frustum[0], r = c->frustum[1], b = c->frustum[2], t = c->frustum[3];
  o->pxs = (float)((r - l) / c->W);
  o->pxl = (float)l;
  o->pys = (float)((t - b) / c->H);
  o->pyl = (float)b;
  const double R0[3] = {inv[0], inv[1], inv[2]}, R1[3] = {inv[4], inv[5], inv[6]}, R2[3] = {inv[8], inv[9], inv[10]};
  const double e[3] = {inv[12], inv[13], inv[14]};
  for (int a = 0; a < 3; ++a) {
    const double s = N[a] / f[a];
    o->Ec[a] = (float)(e[a] * s - 0.5);
    o->Dx[a] = (float)(R0[a] * s);
    o->Dy[a] = (float)(R1[a] * s);
    o->Dc[a] = (float)(-n * R2[a] * s);
  }
  o->nDx = (float)dot3d(sn, R0);
  o->nDy = (float)dot3d(sn, R1);
  o->nDc = (float)(-n * dot3d(sn, R2));
  o->num0 = (float)(tmin - dot3d(sn, e));
  o->dnum = (float)dc;
  // light view: x' = s.q, y' = u.q, z' = 1 - F.q, w = 1 + z'/d0 for a world point q = xform (X - f/2)
  const double F[3] = {-ld[0], -ld[1], -ld[2]};
  double sv[3] = {F[1] * 0 - F[2] * 1, F[2] * 0 - F[0] * 0, F[0] * 1 - F[1] * 0};
  const double sl = sqrt(dot3d(sv, sv));
  if (!(sl > 1e-12)) FAIL(c, "smk_render: shadows: a light on the y axis has no light transform (gluLookAt with up = y, LTWidgetRen.cpp:262-276)");
  for (int k = 0; k < 3; ++k) sv[k] /= sl;
  const double uv[3] = {sv[1] * F[2] - sv[2] * F[1], sv[2] * F[0] - sv[0] * F[2], sv[0] * F[1] - sv[1] * F[0]};
  double rowx[4], rowy[4], roww[4];
  for (int a = 0; a < 3; ++a) {
    const double col[3] = {xf[4 * a + 0], xf[4 * a + 1], xf[4 * a + 2]};
    rowx[a] = dot3d(sv, col);
    rowy[a] = dot3d(uv, col);
    roww[a] = -dot3d(F, col) / d0;
  }
  {
    const double tcol[3] = {xf[12], xf[13], xf[14]};
    rowx[3] = dot3d(sv, tcol);
    rowy[3] = dot3d(uv, tcol);
    roww[3] = 1.0 + (1.0 - dot3d(F, tcol)) / d0;
    for (int a = 0; a < 3; ++a) {
      rowx[3] -= rowx[a] * f[a] * .5;
      rowy[3] -= rowy[a] * f[a] * .5;
      roww[3] -= roww[a] * f[a] * .5;
    }
  }
  double cx = rowx[3], cy = rowy[3], cw = roww[3];
  for (int a = 0; a < 3; ++a) {
    const double sc = f[a] / N[a];
    o->Xm[a] = (float)(rowx[a] * sc);
    o->Ym[a] = (float)(rowy[a] * sc);
    o->Wm[a] = (float)(roww[a] * sc);
    cx += rowx[a] * sc * .5;
    cy += rowy[a] * sc * .5;
    cw += roww[a] * sc * .5;
  }
  o->Xm[3] = (float)cx;
  o->Ym[3] = (float)cy;
  o->Wm[3] = (float)cw;
  const double LBf = (double)c->shadow_q * (double)c->shadow_px;
  o->LB = (int)ceil(LBf);
  if (o->LB < 1) FAIL(c, "smk_render: shadows: empty light buffer");
  o->lscale = (float)(.85 * LBf);
  o->lbias = (float)(.5 * LBf);
  o->las = (float)(1.0 / (.85 * LBf));
  o->lal = (float)(-.5 / .85);
  double apex[3], gx[3], gy[3], gc[3];
  for (int a = 0; a < 3; ++a) {
    apex[a] = xinv[12 + a] + f[a] * .5;
    gx[a] = gy[a] = gc[a] = 0;
    for (int k = 0; k < 3; ++k) {
      apex[a] += xinv[4 * k + a] * F[k] * (1.0 + d0);
      gx[a] += xinv[4 * k + a] * sv[k];
      gy[a] += xinv[4 * k + a] * uv[k];
      gc[a] += xinv[4 * k + a] * F[k] * -d0;
    }
  }
  for (int a = 0; a < 3; ++a) {
    const double s = N[a] / f[a];
    o->Lc[a] = (float)(apex[a] * s - 0.5);
    o->Gx[a] = (float)(gx[a] * s);
    o->Gy[a] = (float)(gy[a] * s);
    o->Gc[a] = (float)(gc[a] * s);
  }
  o->nGx = (float)dot3d(sn, gx);
  o->nGy = (float)dot3d(sn, gy);
  o->nGc = (float)dot3d(sn, gc);
  o->lnum0 = (float)(tmin - dot3d(sn, apex));
  o->ldnum = (float)dc;
  return 0;
}

extern "C" int smk_set_shadow(smk_ctx *c, int on, int buffer_px, float quality) {
  if (!c) return 1;
  if (on && (buffer_px < 1 || buffer_px > 8192 || !(quality > 0.0f) || quality > 1.0f))
    FAIL(c, "smk_set_shadow: buffer_px in 1..8192 and quality in (0,1] (gluvvui.cpp:156-167 clamps the qualities to [.1,1])");
  c->shadow_on = on ? 1 : 0;
  if (on) {
    c->shadow_px = buffer_px;
    c->shadow_q = quality;
  }
  return 0;
}

extern "C" int smk_get_shadowcoef(smk_ctx *c, smk_shadowcoef *out) {
  if (!c || !out) return 1;
  if (!c->have_volume || !c->have_camera) FAIL(c, "smk_get_shadowcoef: volume and camera must be set");
  return compute_shadowcoef(c, out);
}

extern "C" int smk_get_light_buffer(smk_ctx *c, float *rgba_out, int *lb_out) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->light_lb || !c->d_light_last) FAIL(c, "smk_get_light_buffer: no frame with shadows has been rendered");
  if (lb_out) *lb_out = c->light_lb;
  if (rgba_out) {
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(rgba_out, c->d_light_last, (size_t)c->light_lb * c->light_lb * 16, hipMemcpyDeviceToHost));
  }
  return 0;
}

static void normalize3(float v[3]) {
  float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (l > 0) {
    v[0] /= l;
    v[1] /= l;
    v[2] /= l;
  }
}

// light / half vectors: R8kVolRen3D::loadCubeTex (:2625-2640) or NV20VolRen3D::setupRegComb (:637-668)
static void shading_vectors(const smk_ctx *c, RenderParams &P) {
  const float *r = c->xform;
  // Nw = rows of xform . n (R8kVolRen3D.cpp:333-339)
  P.R[0] = r[0]; P.R[1] = r[4]; P.R[2] = r[8];
  P.R[3] = r[1]; P.R[4] = r[5]; P.R[5] = r[9];
  P.R[6] = r[2]; P.R[7] = r[6]; P.R[8] = r[10];
  P.intens = c->intens;
  P.use_spec = (c->shade == SMK_SHADE_R8K_DSPEC || c->shade == SMK_SHADE_NV20_DSPEC) ? 1 : 0;
  if (c->shade == SMK_SHADE_R8K_DIFF || c->shade == SMK_SHADE_R8K_DSPEC) {
    float l[3] = {-c->light_pos[0], -c->light_pos[1], -c->light_pos[2]};
    normalize3(l);
    float vd[3] = {-(c->eye[0] - c->at[0]), -(c->eye[1] - c->at[1]), -(c->eye[2] - c->at[2])};
    normalize3(vd);
    float h[3];
    for (int k = 0; k < 3; ++k) h[k] = l[k] + 0.5f * (vd[k] - l[k]);
    normalize3(h);
    memcpy(P.L, l, 12);
    memcpy(P.Hv, h, 12);
  } else {
    float vd[3] = {c->eye[0] - c->at[0], c->eye[1] - c->at[1], c->eye[2] - c->at[2]};
    normalize3(vd);
    float lt[3] = {c->light_pos[0] - c->at[0], c->light_pos[1] - c->at[1], c->light_pos[2] - c->at[2]};
    normalize3(lt);
    float h[3];
    for (int k = 0; k < 3; ++k) h[k] = lt[k] + 0.5f * (vd[k] - lt[k]);
    // through inverse(xform) (= transpose of the rotation), negated, normalised
    float hv[3] = {r[0] * h[0] + r[1] * h[1] + r[2] * h[2], r[4] * h[0] + r[5] * h[1] + r[6] * h[2],
                   r[8] * h[0] + r[9] * h[1] + r[10] * h[2]};
    float lv[3] = {r[0] * lt[0] + r[1] * lt[1] + r[2] * lt[2], r[4] * lt[0] + r[5] * lt[1] + r[6] * lt[2],
                   r[8] * lt[0] + r[9] * lt[1] + r[10] * lt[2]};
    for (int k = 0; k < 3; ++k) {
      hv[k] = -hv[k];
      lv[k] = -lv[k];
    }
    normalize3(hv);
    normalize3(lv);
    memcpy(P.L, lv, 12);
    memcpy(P.Hv, hv, 12);
  }
}

// brick flags of one table (smk_bricks.hip): buffers sized on demand, the two launches, the count copied back behind them
static int make_brick_set(smk_ctx *c, BrickSet &B, const uint32_t *occ, int roww, int sv, int sg, hipStream_t s) {
  const size_t nbricks = (size_t)c->nbr[0] * c->nbr[1] * c->nbr[2], sat_words = (size_t)(sv + 1) * (sg + 1);
  if (B.flags_cap < nbricks) {  // (a new volume size: rare; hipFree waits for the frames in flight)
    if (B.flags) (void)hipFree(B.flags);
    B.flags = nullptr;
    B.flags_cap = 0;
    HIPCHK(c, hipMalloc((void **)&B.flags, nbricks));
    B.flags_cap = nbricks;
  }
  if (B.sat_cap < sat_words) {
    if (B.sat) (void)hipFree(B.sat);
    B.sat = nullptr;
    B.sat_cap = 0;
    HIPCHK(c, hipMalloc((void **)&B.sat, sat_words * 4));
    B.sat_cap = sat_words;
  }
  if (!B.d_count) HIPCHK(c, hipMalloc((void **)&B.d_count, 4));
  if (!B.h_count) HIPCHK(c, hipHostMalloc((void **)&B.h_count, 4, hipHostMallocDefault));
  if (!B.counted) HIPCHK(c, hipEventCreateWithFlags(&B.counted, hipEventDisableTiming));
  else HIPCHK(c, hipEventSynchronize(B.counted));  // (the pinned word's previous copy: long done)
  HIPCHK(c, smk_bricks_flags(c->d_brick_mm, c->nbr, occ, roww, sv, sg, B.sat, B.flags, B.d_count, s));
  HIPCHK(c, hipMemcpyAsync(B.h_count, B.d_count, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipEventRecord(B.counted, s));
  B.valid = true;
  B.fill = -1.f;
  B.dil_r[0] = -1;  // (the dilated copy belongs to the old flags)
  return 0;
}

// the flags a frame should use, or null: none made, or (known by now) nearly every brick is flagged -- a table that
// leaves nothing to skip would only pay the flags' set-up in every workgroup
static const unsigned char *brick_flags_to_use(smk_ctx *c, BrickSet &B) {
  if (!B.valid) return nullptr;
  if (B.fill < 0.f && hipEventQuery(B.counted) == hipSuccess)
    B.fill = (float)((double)*B.h_count / ((double)c->nbr[0] * c->nbr[1] * c->nbr[2]));
  (void)hipGetLastError();
  return B.fill > 0.9f ? nullptr : B.flags;
}

// NV20VolRen3D::copyScale (:1645-1660) with the rate the renderer would pass (:94-98, :117)
// The effective (V,G) table of a frame: opacity correction as copyScale does it, plus the occupancy bitmap.  In steps
// mode the correction rate follows the view-depth extent, i.e. it changes with every camera move: the table is then
// rebuilt per frame, so the rebuild must neither stall the pipeline nor cost the host much.  The correction is a function
// of the alpha byte alone (a 256-entry map, same arithmetic as copyScale), so the host computes that map and a kernel
// applies it to the raw table kept on the device, writing the effective table and its bitmap into one of four versions
// (a version is rewritten only after the last frame that read it: an event wait ON THE STREAM, not on the host).
// Round 2 before: hipFree + hipMalloc + hipMemcpy per refresh = a device synchronisation and 1.6 ms of pow() per frame;
// then 65536 map look-ups + the bitmap on the host (0.4 ms per moving-camera frame, more than the launcher's planning).
__global__ __launch_bounds__(256) void smk_k_tf_effective(const uint32_t *raw, const unsigned char *map, int sv, int sg, int roww,
                                                           uint32_t *eff, uint32_t *occ) {
  // a wave = 64 consecutive texels of one row: its two bitmap words come from one ballot
  const int wave = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  const int wpr = (sv + 63) / 64;  // waves per row
  const int t = wave / wpr, s = (wave - t * wpr) * 64 + lane;
  if (t >= sg) return;
  bool any = false;
  if (s < sv) {
    const int s1 = min(s + 1, sv - 1), t1 = min(t + 1, sg - 1);
    const uint32_t a = raw[(size_t)t * sv + s];
    const unsigned ma = map[a >> 24];
    eff[(size_t)t * sv + s] = (a & 0x00ffffffu) | (ma << 24);
    any = (ma | map[raw[(size_t)t * sv + s1] >> 24] | map[raw[(size_t)t1 * sv + s] >> 24] | map[raw[(size_t)t1 * sv + s1] >> 24]) != 0;
  }
  const unsigned long long m = __ballot(any);
  const int w0 = (s - lane) >> 5;
  if (lane == 0 && w0 < roww) occ[(size_t)t * roww + w0] = (uint32_t)m;
  if (lane == 32 && w0 + 1 < roww) occ[(size_t)t * roww + w0 + 1] = (uint32_t)(m >> 32);
}

static float tf2d_rate(const smk_ctx *c, const smk_raycoef &rc) {
  if (c->scale_alphas) {
    float rate = c->steps > 0 ? (float)(c->fsize[0] / ((double)c->N[0] * (double)rc.dis)) : c->sample_rate;
    return rate * 1 / c->gamma;
  }
  return 1 / c->gamma;
}

// NV20VolRen3D::copyScale (:1645-1660) per possible alpha byte
static void tf2d_alpha_map(const smk_ctx *c, float sr, unsigned char map[256]) {
  if (c->opt_tf_raw) {
    for (int a = 0; a < 256; ++a) map[a] = (unsigned char)a;
    return;
  }
  const float alphaScale = (float)(1.0 / sr);
  for (int a = 0; a < 256; ++a) map[a] = (unsigned char)(int)((1.0 - pow((1.0 - (a / 255.0)), alphaScale)) * 255);
}

static int refresh_tf2d(smk_ctx *c, const smk_raycoef &rc, hipStream_t frame_stream) {
  if (c->tf_mode != 1) return 0;
  const float sr = tf2d_rate(c, rc);
  if (!c->tf_dirty && sr == c->tf_rate_applied && c->d_tf_vg) {
    if (c->tf_ready_pending) HIPCHK(c, hipStreamWaitEvent(frame_stream, c->tf_ready, 0));  // (a no-op once the refresh has run)
    return 0;
  }
  // The correction is a function of the alpha BYTE: when the new rate maps every byte where the old one did -- a camera
  // that turns a little changes the view-depth extent in its fifth digit -- the effective table, its occupancy bitmap and
  // the brick flags would all come out byte for byte the same: nothing to refresh (a moving camera paid ~0.15 ms of
  // stream time per frame for these launches).
  unsigned char newmap[256];
  tf2d_alpha_map(c, sr, newmap);
  if (!c->tf_dirty && c->d_tf_vg && c->tf_map_valid && !memcmp(newmap, c->tf_map_applied, 256)) {
    c->tf_rate_applied = sr;
    if (c->tf_ready_pending) HIPCHK(c, hipStreamWaitEvent(frame_stream, c->tf_ready, 0));
    return 0;
  }
  // The refresh (alpha map copy, effective table + bitmap, summed-area table, brick flags: four small launches) runs on a
  // stream of its own and the frame's stream waits for its end: when the camera moves every frame the host enqueues frame
  // i + 1's refresh while frame i is still ray-marching, and the two overlap instead of queueing up behind one another.
  if (!c->tf_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->tf_stream, hipStreamNonBlocking));
  if (!c->tf_ready) HIPCHK(c, hipEventCreateWithFlags(&c->tf_ready, hipEventDisableTiming));
  const hipStream_t s = c->tf_stream;
  const size_t n = (size_t)c->sv * c->sg;
  const int sv = c->sv, sg = c->sg, roww = (sv + 31) / 32;
  const size_t occ_words = (size_t)roww * sg, bytes = n * 4 + occ_words * 4 + 256;  // table | bitmap | alpha map
  const int v = (c->tf_cur + 1) & 3;
  smk_ctx::TfVersion &T = c->tfv[v];
  if (T.cap < bytes) {  // (a new table size: rare)
    if (T.copied) (void)hipEventSynchronize(T.copied);
    if (T.used_valid) (void)hipEventSynchronize(T.used);
    if (T.d) (void)hipFree(T.d);
    if (T.h) (void)hipHostFree(T.h);
    T.d = nullptr;
    T.h = nullptr;
    T.cap = 0;
    HIPCHK(c, hipMalloc((void **)&T.d, bytes));
    HIPCHK(c, hipHostMalloc((void **)&T.h, bytes, hipHostMallocDefault));
    T.cap = bytes;
    if (!T.copied) HIPCHK(c, hipEventCreateWithFlags(&T.copied, hipEventDisableTiming));
    if (!T.used) HIPCHK(c, hipEventCreateWithFlags(&T.used, hipEventDisableTiming));
    T.used_valid = false;
  } else if (T.copied) {
    HIPCHK(c, hipEventSynchronize(T.copied));  // the staging buffer's previous copies (four refreshes ago) have run
  }
  // the raw table lives on the device; a new one travels through this version's staging buffer
  if (c->tf_raw_cap < n * 4) {
    HIPCHK(c, hipDeviceSynchronize());
    if (c->d_tf_raw) (void)hipFree(c->d_tf_raw);
    c->d_tf_raw = nullptr;
    c->tf_raw_cap = 0;
    HIPCHK(c, hipMalloc((void **)&c->d_tf_raw, n * 4));
    c->tf_raw_cap = n * 4;
    c->tf_raw_stale = true;
  }
  if (!c->tf_raw_ev) HIPCHK(c, hipEventCreateWithFlags(&c->tf_raw_ev, hipEventDisableTiming));
  if (c->tf_raw_stale) {
    memcpy(T.h, c->h_tf_vg.data(), n * 4);
    if (c->tf_raw_ev_valid) HIPCHK(c, hipStreamWaitEvent(s, c->tf_raw_ev, 0));  // the last kernel that read the old raw table is done
    HIPCHK(c, hipMemcpyAsync(c->d_tf_raw, T.h, n * 4, hipMemcpyHostToDevice, s));
    c->tf_raw_stale = false;
  }
  unsigned char *map = T.h + n * 4 + occ_words * 4;
  memcpy(map, newmap, 256);
  memcpy(c->tf_map_applied, newmap, 256);
  c->tf_map_valid = true;
  if (T.used_valid) HIPCHK(c, hipStreamWaitEvent(s, T.used, 0));  // the last frame that read this version is done
  HIPCHK(c, hipMemcpyAsync(T.d + n * 4 + occ_words * 4, map, 256, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipEventRecord(T.copied, s));
  // effective table + occupancy bitmap: bit (t, s) is set when any of the four texels a bilinear lookup with base texel
  // (s, t) touches has alpha != 0 after the correction.  A clear bit means the lookup returns alpha == 0 EXACTLY (lerps
  // of zeros), so a kernel may skip the fetch without changing a single bit of the frame.
  {
    const int wpr = (sv + 63) / 64;
    const unsigned blocks = (unsigned)(((size_t)wpr * sg + 3) / 4);
    hipLaunchKernelGGL(smk_k_tf_effective, dim3(blocks), dim3(256), 0, s, (const uint32_t *)c->d_tf_raw, T.d + n * 4 + occ_words * 4, sv,
                       sg, roww, (uint32_t *)T.d, (uint32_t *)(T.d + n * 4));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->tf_raw_ev, s));
    c->tf_raw_ev_valid = true;
  }
  c->tf_cur = v;
  c->d_tf_vg = reinterpret_cast<uint32_t *>(T.d);
  c->d_tf_occ = reinterpret_cast<uint32_t *>(T.d + n * 4);
  // this version's brick flags (smk_bricks.hip), behind the table on the same stream: two small launches per refresh
  T.br.valid = false;
  if (c->opt_bricks && c->d_brick_mm && make_brick_set(c, T.br, c->d_tf_occ, roww, sv, sg, s)) return 1;
  c->tf_occ_roww = roww;
  c->tf_rate_applied = sr;
  c->tf_dirty = false;
  HIPCHK(c, hipEventRecord(c->tf_ready, s));
  HIPCHK(c, hipStreamWaitEvent(frame_stream, c->tf_ready, 0));
  c->tf_ready_pending = true;
  return 0;
}

extern "C" int smk_get_tf2d_effective(smk_ctx *c, unsigned char *out, float *rate) {
  if (!c) return 1;
  if (c->tf_mode != 1 || !c->have_volume || !c->have_camera) FAIL(c, "smk_get_tf2d_effective: 2-D TF, volume and camera required");
  smk_raycoef rc;
  double inv[16];
  compute_raycoef(c, &rc, inv);
  if (refresh_tf2d(c, rc, c->stream)) return 1;
  if (out) {  // (the same map applied on the host: this call is for checkers, not for frames)
    unsigned char map[256];
    tf2d_alpha_map(c, c->tf_rate_applied, map);
    memcpy(out, c->h_tf_vg.data(), c->h_tf_vg.size());
    for (size_t i = 0; i < (size_t)c->sv * c->sg; ++i) out[i * 4 + 3] = map[c->h_tf_vg[i * 4 + 3]];
  }
  if (rate) *rate = c->tf_rate_applied;
  return 0;
}

extern "C" int smk_shard_order(smk_ctx *c, int *order) {
  if (!c || !order) return 1;
  if (!c->have_volume || !c->have_camera) FAIL(c, "smk_shard_order: volume and camera must be set");
  double inv[16];
  inverse_affine(inv, c->mv);
  // eye in voxel index space; per split axis the half holding the eye is in front (BSP order)
  int nearbit[3];
  for (int a = 0; a < 3; ++a) {
    double e = inv[12 + a] * c->N[a] / c->fsize[a] - 0.5;
    nearbit[a] = e >= (double)(c->N[a] / 2) - 0.5 ? 1 : 0;
  }
  int nbits = 0;
  for (int n = c->nranks; n > 1; n >>= 1) ++nbits;
  std::vector<std::pair<int, int>> keyed;
  for (int r = 0; r < c->nranks; ++r) {
    int key = 0;
    for (int bit = 0; bit < nbits; ++bit)
      if (((r >> bit) & 1) != nearbit[bit % 3]) key |= 1 << bit;
    keyed.push_back({key, r});
  }
  std::sort(keyed.begin(), keyed.end());
  for (int r = 0; r < c->nranks; ++r) order[r] = keyed[r].second;
  return 0;
}

// ------------------------------------------------------------------------------- render

extern "C" int smk_set_option(smk_ctx *c, const char *key, int value) {
  if (!c || !key) return 1;
  if (!strcmp(key, "kernel")) c->opt_kernel = value;
  else if (!strcmp(key, "slab_T")) c->opt_slab_T = value;
  else if (!strcmp(key, "inject_slab_status")) c->opt_inject_status = value;
  else if (!strcmp(key, "slab_fly")) c->opt_slab_fly = value < 0 ? 0 : (value > 63 ? 63 : value);
  else if (!strcmp(key, "slab_sched")) c->opt_slab_sched = value < 0 ? 0 : (value > 15 ? 15 : value);
  else if (!strcmp(key, "slab_ns")) c->opt_slab_ns = value < 0 ? 0 : (value > 63 ? 63 : value);
  else if (!strcmp(key, "tile")) c->opt_tile = value;
  else if (!strcmp(key, "shadow_march")) c->opt_shadow_march = value ? 1 : 0;  // (0: a launch per slice, the form of rounds 1-2)
  else if (!strcmp(key, "shadow_fused")) c->opt_lockstep = value ? (c->opt_lockstep | 256) : (c->opt_lockstep & ~256);  // (developer: all slices in one cooperative launch)
  else if (!strcmp(key, "slab_split")) c->slab.opt_split = value < 0 ? 0 : (value > 8 ? 8 : value);
  else if (!strcmp(key, "cols_shape")) c->opt_cols = (c->opt_cols & ~0xff) | (value & 0xff);
  else if (!strcmp(key, "cols_ns")) c->opt_cols = (c->opt_cols & ~0xff00) | ((value & 0xff) << 8);
  else if (!strcmp(key, "cols_chunk")) c->opt_cols = (c->opt_cols & ~0xfff0000) | ((value & 0xfff) << 16);
  else if (!strcmp(key, "cols_wstep")) c->opt_cols = (c->opt_cols & 0x0fffffff) | ((value & 7) << 28);
  else if (!strcmp(key, "cols_counts")) c->cols.want_counts = value ? 1 : 0;
  else if (!strcmp(key, "cols_fill")) { c->cols.opt_fill = value; smk_cols_drop_layouts(&c->cols); }
  else if (!strcmp(key, "cols_take_min")) c->cols.opt_take_min = value;
  else if (!strcmp(key, "cols_take_wait")) c->cols.opt_take_wait = value;
  else if (!strcmp(key, "cols_fly")) c->cols.opt_fly = value;
  else if (!strcmp(key, "bricks")) {  // 0: the slice-ring kernel streams and samples every slice (smk_bricks.hip off)
    c->opt_bricks = value ? 1 : 0;
    c->tf_dirty = true;
    c->bricks3_dirty = true;
  }
  else if (!strcmp(key, "wave_w")) {
    if (!(value == 1 || value == 2 || value == 4 || value == 8 || value == 16 || value == 32 || value == 64)) FAIL(c, "smk_set_option: wave_w must be a power of two <= 64");
    c->opt_wave_w = value;
  } else if (!strcmp(key, "blk_w")) {
    if (!(value == 1 || value == 2 || value == 4)) FAIL(c, "smk_set_option: blk_w must be 1, 2 or 4");
    c->opt_blk_w = value;
  } else if (!strcmp(key, "lockstep")) c->opt_lockstep = value;
  else if (!strcmp(key, "tf_raw")) {  // the 2-D TF handed over is already opacity-corrected
    c->opt_tf_raw = value;
    c->tf_dirty = true;
  }
  else if (!strcmp(key, "halo")) {
    if (c->have_volume) FAIL(c, "smk_set_option: halo must be set before smk_upload_volume");
    if (value < 1) FAIL(c, "smk_set_option: halo must be >= 1");
    c->halo = value;
  } else
    FAIL(c, "smk_set_option: unknown key '%s'", key);
  return 0;
}

extern "C" int smk_timing_reset(smk_ctx *c) {
  if (!c) return 1;
  c->tcount = 0;
  return 0;
}

// average render-kernel duration over the frames recorded since smk_timing_reset (at most the
// last SMK_TIMING_RING); synchronises the device
extern "C" int smk_timing_read(smk_ctx *c, float *avg_ms, int *nframes) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  int n = (int)std::min<long long>(c->tcount, SMK_TIMING_RING);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->tev0[i], c->tev1[i]));
    sum += ms;
  }
  if (avg_ms) *avg_ms = n ? (float)(sum / n) : 0.f;
  if (nframes) *nframes = n;
  return 0;
}

// Slice-ring kernel error words -> failed frames (the kernel never hangs and never returns a frame
// built from unloaded data silently).  Every frame has its own word, SMK_STATUS_RING of them in turn.
// A caller that keeps frames in flight asks per frame (smk_frame_failed, after synchronising with it)
// and renders a flagged frame again; a flag nobody asked about fails the NEXT render call loudly.
// The status word of frame `id` (slot id % SMK_STATUS_RING), consumed: 0 = none.  Words carry the id of the frame that wrote
// them (a slice-ring frame may write late, into a slot that has since been handed to a younger frame): one of ANOTHER frame
// is counted as that frame's failure and left alone for whoever owns it -- or dropped when that frame is out of the ring.
static int take_status(smk_ctx *c, long long id) {
  if (!c->slab.h_status || id <= 0) return 0;
  volatile int *w = (volatile int *)c->slab.h_status + (int)(id % SMK_STATUS_RING);
  const int st = *w;
  if (!st) return 0;
  const long long tag = (st >> 8) & 0x7fffff;
  if (tag != (id & 0x7fffff)) {
    // a late word of an older frame of this slot: nobody can be told about that frame any more
    if (((id - tag) & 0x7fffff) % SMK_STATUS_RING == 0 && tag != 0) {
      *w = 0;
      ++c->slab_failures;
      ++c->slab_lost;
    }
    return 0;
  }
  *w = 0;
  ++c->slab_failures;
  // not again soon: in auto mode that configuration is the gather kernel's for a while
  if (c->opt_kernel == 0 && c->last_slab_sig) c->tune_choice[c->last_slab_sig] = {1, c->frame_id + 256};
  return st & 0xff;
}

static int build_params(smk_ctx *c, RenderParams &P, hipStream_t s);
extern "C" int smk_get_brick_flags(smk_ctx *c, unsigned char *flags_out, int *nb_out, int *in_use_out) {
  if (!c || !nb_out) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  RenderParams P;
  if (build_params(c, P, c->stream)) return 1;
  HIPCHK(c, hipDeviceSynchronize());
  for (int a = 0; a < 3; ++a) nb_out[a] = c->nbr[a];
  const BrickSet *B = c->tf_mode == 1 && c->tf_cur >= 0 ? &c->tfv[c->tf_cur].br : c->tf_mode == 2 ? &c->br3 : nullptr;
  if (in_use_out) *in_use_out = P.bricks != nullptr ? 1 : 0;
  if (flags_out) {
    if (!B || !B->valid || !B->flags) FAIL(c, "smk_get_brick_flags: no flags for this table (1-D colour table, or option 'bricks' 0)");
    HIPCHK(c, hipMemcpy(flags_out, B->flags, (size_t)c->nbr[0] * c->nbr[1] * c->nbr[2], hipMemcpyDeviceToHost));
  }
  return 0;
}

static const char *status_text(int st) {
  return st == 1 ? "a streaming kernel reported a producer/consumer time-out" : st == 2 ? "the slice-ring kernel reported a window outside its host bound"
         : st == 3 ? "the column-stream kernel reported a job whose rays do not fit its lanes or its list" : "the column-stream kernel reported a ray it cannot list";
}

// frame `id` was flagged and nobody has asked about it: the call fails
static int check_frame_status(smk_ctx *c, long long id) {
  const int st = take_status(c, id);
  if (st) FAIL(c, "%s (status %d, frame %lld); frame invalid", status_text(st), st, id);
  return 0;
}

extern "C" long long smk_last_frame_id(smk_ctx *c) { return c ? c->frame_id : 0; }

extern "C" int smk_frame_failed(smk_ctx *c, long long frame_id) {
  if (!c) return 1;
  if (frame_id <= 0 || frame_id > c->frame_id || frame_id + SMK_STATUS_RING <= c->frame_id) return -1;  // never enqueued, or out of the ring: unknown
  return take_status(c, frame_id) ? 1 : 0;
}

extern "C" int smk_get_stat(smk_ctx *c, const char *name, double *value) {
  if (!c || !name || !value) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  static const char *diag_names[16] = {"slab_iters", "slab_active_lanes", "slab_inside_lanes", "slab_hit_lanes",
                                      "slab_loader_issue_kcyc", "slab_loader_wait_kcyc", "slab_loader_blocked_kcyc", "slab_loader_total_kcyc",
                                      "slab_iters_with_hit", "slab_lead_sum", "slab_waits", "slab_wstep_sum", "slab_dead_tail_sum", "slab_waves",
                                      "slab_iters_sampling", "slab_iters_own_brick"};
  for (int k = 0; k < 16; ++k)
    if (!strcmp(name, diag_names[k])) {
      float v = 0.f;
      if (c->slab.d_diag) {
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, hipMemcpy(&v, c->slab.d_diag + k, 4, hipMemcpyDeviceToHost));
      }
      *value = v;
      return 0;
    }
  if (!strcmp(name, "slab_status")) {  // status word of the latest frame (0 = ok); synchronises
    HIPCHK(c, hipDeviceSynchronize());
    *value = c->slab.h_status ? (((volatile int *)c->slab.h_status)[c->slab.status_slot] & 0xff) : 0;
    return 0;
  }
  // frames the slice-ring kernel flagged invalid, as far as the host has looked (no synchronisation:
  // call after the frames of interest have completed); frames smk_render rendered a second time
  if (!strcmp(name, "slab_failures")) {
    long long n = c->slab_failures;
    if (c->slab.h_status)
      for (int k = 0; k < SMK_STATUS_RING; ++k) n += ((volatile int *)c->slab.h_status)[k] != 0;
    *value = (double)n;
    return 0;
  }
  // share of the slices in the tiles' ranges that the loaders of the latest slice-ring frame really
  // streamed (they stop when every ray of a tile is saturated); synchronises
  if (!strcmp(name, "slab_streamed_fraction")) {
    *value = 1.0;
    const int nt = c->slab.ticks_n_last;
    if (c->last_kernel == 2 && c->slab.d_ticks && nt > 0) {
      HIPCHK(c, hipDeviceSynchronize());
      std::vector<unsigned> h((size_t)2 * nt);
      HIPCHK(c, hipMemcpy(h.data(), c->slab.d_ticks + nt, (size_t)2 * nt * 4, hipMemcpyDeviceToHost));
      double st = 0, pl = 0;
      for (int t = 0; t < nt; ++t) {
        st += h[t];
        pl += h[(size_t)nt + t];
      }
      if (pl > 0) *value = st / pl;
    }
    return 0;
  }
  // longest and summed workgroup durations of the latest slice-ring frame, in ms (the kernel's per-tile ticks); synchronises
  if (!strcmp(name, "slab_tile_ms_max") || !strcmp(name, "slab_tile_ms_sum")) {
    *value = 0.0;
    const int nt = c->slab.ticks_n_last;
    if (c->last_kernel == 2 && c->slab.d_ticks && nt > 0) {
      HIPCHK(c, hipDeviceSynchronize());
      std::vector<unsigned> h((size_t)nt);
      HIPCHK(c, hipMemcpy(h.data(), c->slab.d_ticks, (size_t)nt * 4, hipMemcpyDeviceToHost));
      double mx = 0, sum = 0;
      for (int t = 0; t < nt; ++t) {
        // (a split tile's word is the sum over its pieces: the longest workgroup is taken as an equal share)
        const double k = t < (int)c->slab.ksplit_last.size() ? std::max<int>(c->slab.ksplit_last[t], 1) : 1;
        mx = std::max(mx, (double)h[t] / k);
        sum += h[t];
      }
      *value = (name[13] == 'm' ? mx : sum) * 1e-5;  // 100 MHz ticks
    }
    return 0;
  }
  // column-stream kernel (smk_cols.hip), latest frame; these synchronise
  if (!strncmp(name, "cols_", 5)) {
    *value = 0.0;
    if (!strcmp(name, "cols_builds")) { *value = c->cols.builds; return 0; }
    if (!strcmp(name, "cols_config")) { *value = c->cols.last; return 0; }
    if (!strcmp(name, "cols_jobs")) { *value = c->cols.njobs_last; return 0; }
    if (!strcmp(name, "cols_stream_bytes")) { *value = c->cols.last_stream_bytes; return 0; }
    if (!strcmp(name, "cols_setup_ms_sum") || !strcmp(name, "cols_rays")) {
      const int nj = c->cols.njobs_last;
      if (c->last_kernel == 4 && c->cols.d_ticks && nj > 0) {
        HIPCHK(c, hipDeviceSynchronize());
        std::vector<unsigned> h((size_t)nj);
        HIPCHK(c, hipMemcpy(h.data(), c->cols.d_ticks + (size_t)nj * (name[5] == 's' ? 1 : 2), (size_t)nj * 4, hipMemcpyDeviceToHost));
        double sum = 0;
        for (int t = 0; t < nj; ++t) sum += h[t];
        *value = name[5] == 's' ? sum * 1e-5 : sum;
      }
      return 0;
    }
    if (!strcmp(name, "cols_job_ms_max") || !strcmp(name, "cols_job_ms_sum")) {
      const int nj = c->cols.njobs_last;
      if (c->last_kernel == 4 && c->cols.d_ticks && nj > 0) {
        HIPCHK(c, hipDeviceSynchronize());
        std::vector<unsigned> h((size_t)nj);
        HIPCHK(c, hipMemcpy(h.data(), c->cols.d_ticks, (size_t)nj * 4, hipMemcpyDeviceToHost));
        double mx = 0, sum = 0;
        for (int t = 0; t < nj; ++t) {
          mx = std::max(mx, (double)h[t]);
          sum += h[t];
        }
        *value = (name[12] == 'm' ? mx : sum) * 1e-5;
      }
      return 0;
    }
    static const char *cn[8] = {"cols_samples", "cols_visible", "cols_slices", "cols_segments", "cols_iters", "cols_active_lanes", "cols_switch_iters", "cols_switch_lanes"};
    for (int k = 0; k < 8; ++k)
      if (!strcmp(name, cn[k])) {
        if (c->cols.d_counts && c->cols.want_counts) {
          HIPCHK(c, hipDeviceSynchronize());
          unsigned long long v = 0;
          HIPCHK(c, hipMemcpy(&v, c->cols.d_counts + k, 8, hipMemcpyDeviceToHost));
          *value = (double)v;
        }
        return 0;
      }
    FAIL(c, "smk_get_stat: unknown name '%s'", name);
  }
  if (!strcmp(name, "slab_split_tiles")) { *value = c->slab.nsplit_last; return 0; }
  if (!strcmp(name, "slab_workgroups")) { *value = c->slab.nblocks_last; return 0; }
  if (!strcmp(name, "slab_retries")) {
    *value = (double)c->slab_retries;
    return 0;
  }
  FAIL(c, "smk_get_stat: unknown name '%s'", name);
}

extern "C" int smk_get_trace(smk_ctx *c, unsigned *out, int cap_records, int *nrecords) {
  if (!c || !nrecords) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (!(c->opt_lockstep & 32)) {
    // no diagnostic instance ran: the product kernel's own words per tile (start, duration, where) in the same record shape
    const int nt = c->last_kernel == 2 && c->slab.d_ticks ? c->slab.ticks_n_last : 0;
    *nrecords = nt;
    if (out && nt > 0) {
      HIPCHK(c, hipDeviceSynchronize());
      std::vector<unsigned> h((size_t)nt * 5);
      HIPCHK(c, hipMemcpy(h.data(), c->slab.d_ticks, (size_t)nt * 20, hipMemcpyDeviceToHost));
      for (int t = 0; t < std::min(cap_records, nt); ++t) {
        unsigned *r = out + (size_t)t * 8;
        r[0] = h[(size_t)3 * nt + t];
        r[1] = h[(size_t)3 * nt + t] ? r[0] + h[t] : 0u;   // (a tile cut in depth segments has no start word: record left empty)
        r[2] = h[(size_t)4 * nt + t] & 0xff00u;
        r[3] = (h[(size_t)4 * nt + t] & 0xfu) | ((unsigned)t << 8) | (h[(size_t)2 * nt + t] << 20);
        r[4] = r[5] = r[6] = r[7] = 0;
      }
    }
    return 0;
  }
  *nrecords = c->slab.d_trace ? c->slab.trace_n : 0;
  if (out && *nrecords > 0) {
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(out, c->slab.d_trace, (size_t)std::min(cap_records, *nrecords) * 32, hipMemcpyDeviceToHost));
  }
  return 0;
}

extern "C" int smk_count_samples(smk_ctx *c, double *in_volume) {
  if (!c || !in_volume) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  RenderParams P;
  if (build_params(c, P, c->stream)) return 1;
  unsigned long long *d = nullptr, h = 0;
  HIPCHK(c, hipMalloc((void **)&d, 8));
  hipError_t e = hipMemsetAsync(d, 0, 8, c->stream);
  if (e == hipSuccess) e = smk_launch_count_inside(P, d, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  HIPCHK(c, e);
  *in_volume = (double)h;
  return 0;
}

extern "C" int smk_last_frame_info(smk_ctx *c, int *kernel, float *ms, double *alg_bytes) {
  if (!c) return 1;
  if (kernel) *kernel = c->last_kernel;
  if (ms) *ms = c->last_ms;
  if (alg_bytes) *alg_bytes = c->last_alg_bytes;
  return 0;
}

static int build_params(smk_ctx *c, RenderParams &P, hipStream_t s) {
  if (!c->have_volume) FAIL(c, "smk_render: no volume uploaded");
  if (!c->have_camera) FAIL(c, "smk_render: no camera set");
  if (c->tf_mode < 0) FAIL(c, "smk_render: no transfer function set");
  memset(&P, 0, sizeof P);
  double inv[16];
  compute_raycoef(c, &P.rc, inv);
  if (refresh_tf2d(c, P.rc, s)) return 1;
  P.vox = c->d_vox;
  P.nrm = c->d_nrm;
  for (int a = 0; a < 3; ++a) {
    P.N[a] = c->N[a];
    P.O[a] = c->O[a];
    P.D[a] = c->D[a];
    P.lo[a] = (float)c->g0[a] - 0.5f;
    P.hi[a] = (float)c->g1[a] - 0.5f;
    P.top[a] = c->g1[a] == c->N[a];
    P.invN[a] = 1.0f / (float)c->N[a];
  }
  // orthogonal clip plane (NV20VolRen3D::setupClips, NV20VolRen3D.cpp:251-327): the sliced box ends
  // at the plane, i.e. the region shrinks along one axis; the new face is an outer (inclusive) one
  if (c->clip_axis >= 1 && c->clip_axis <= 6) {
    const int a = (c->clip_axis - 1) / 2;
    const float fs = c->fsize[a];
    const float cp = c->clip_vpos[a] > 0.0f ? (c->clip_vpos[a] < fs ? c->clip_vpos[a] : fs) : 0.0f;
    const float face = (float)((double)cp / (double)fs * (double)c->N[a] - 0.5);
    if ((c->clip_axis - 1) % 2 == 0) {
      if (face < P.hi[a] || (face == P.hi[a] && !P.top[a])) { P.hi[a] = face; P.top[a] = 1; }
    } else if (face > P.lo[a]) {
      P.lo[a] = face;
    }
  }
  // sub-box of renderVolume(.., xext, yext, zext) (VolumeRenderer.cpp:428-505): the region shrinks on every axis; the new
  // upper faces are half-open like a shard's inner ones (a face that coincides with a voxel plane belongs to one side)
  if (c->region_on)
    for (int a = 0; a < 3; ++a) {
      const double fs = c->fsize[a];
      const double l = c->region_lo[a] > 0 ? (c->region_lo[a] < fs ? c->region_lo[a] : fs) : 0.0;
      const double h = c->region_hi[a] > 0 ? (c->region_hi[a] < fs ? c->region_hi[a] : fs) : 0.0;
      const float flo = (float)(l / fs * (double)c->N[a] - 0.5), fhi = (float)(h / fs * (double)c->N[a] - 0.5);
      if (flo > P.lo[a]) P.lo[a] = flo;
      if (fhi < P.hi[a]) { P.hi[a] = fhi; P.top[a] = 0; }
    }
  for (int a = 0; a < 3; ++a) P.hin[a] = P.top[a] ? P.hi[a] : nextafterf(P.hi[a], -INFINITY);
  // free clip plane: eye-space plane -> voxel coordinates.  eye = MV * model, model = (p + 1/2)/N * fSize
  // (same operations in the same order as the CPU checker's orc_clip_plane_voxel)
  P.cplane_on = c->cplane_on;
  for (int k = 0; k < 4; ++k) P.cplane[k] = 0.0f;
  if (c->cplane_on) {
    double pm[4];  // plane in model space: row vector times MV (column-major)
    for (int k = 0; k < 4; ++k)
      pm[k] = c->cplane_eye[0] * c->mv[4 * k + 0] + c->cplane_eye[1] * c->mv[4 * k + 1] + c->cplane_eye[2] * c->mv[4 * k + 2] +
              c->cplane_eye[3] * c->mv[4 * k + 3];
    double w = pm[3];
    for (int a = 0; a < 3; ++a) {
      const double sc = (double)c->fsize[a] / (double)c->N[a];
      P.cplane[a] = (float)(pm[a] * sc);
      w += pm[a] * sc * 0.5;
    }
    P.cplane[3] = (float)w;
  }
  P.nelts = c->nelts;
  P.n_in_w = (c->dtype == SMK_F32 && c->nelts <= 3) ? 1 : 0;
  P.tlut = c->d_tlut;
  P.tlut_size = c->tlut_size;
  P.tf_vg = c->d_tf_vg;
  P.tf_h = c->d_tf_h;
  P.tf_occ = c->tf_mode == 1 ? c->d_tf_occ : c->tf_mode == 2 ? c->d_tf3d_occ : nullptr;
  P.occ_roww = c->tf_mode == 2 ? c->tf3d_occ_roww : c->tf_occ_roww;
  P.sv = c->sv;
  P.sg = c->sg;
  // third-axis data modes (NV20VolRen3D.cpp:686-693, 813-819)
  bool third = c->dmode == SMK_GDM_VGH || c->dmode == SMK_GDM_V1GH || c->dmode == SMK_GDM_V2G ||
               c->dmode == SMK_GDM_V2GH || c->dmode == SMK_GDM_V3 || c->dmode == SMK_GDM_V3G || c->dmode == SMK_GDM_V4;
  P.third_axis = (third && c->d_tf_h && c->nelts >= 3) ? 1 : 0;
  P.tf3d = c->d_tf3d;
  P.s3v = c->s3v;
  P.s3g = c->s3g;
  P.s3h = c->s3h;
  // brick flags (smk_bricks.hip): the 2-D table's come with its version; the dense 3-D table's are made here when stale
  P.bricks = nullptr;
  for (int a = 0; a < 3; ++a) P.nbr[a] = c->nbr[a];
  BrickSet *bset = nullptr;
  P.bricks_dil = nullptr;
  if (c->opt_bricks && c->d_brick_mm) {
    if (c->tf_mode == 1 && c->tf_cur >= 0) {
      P.bricks = brick_flags_to_use(c, c->tfv[c->tf_cur].br);
      bset = &c->tfv[c->tf_cur].br;
    }
    if (c->tf_mode == 2 && c->d_tf3d_occ) {
      if (c->bricks3_dirty || !c->br3.valid) {
        HIPCHK(c, hipDeviceSynchronize());  // (a new table or volume: rare; frames in flight may still read the old flags)
        if (make_brick_set(c, c->br3, c->d_tf3d_occ, c->tf3d_occ_roww, c->s3v, c->s3g, s)) return 1;
        c->bricks3_dirty = false;
      }
      P.bricks = brick_flags_to_use(c, c->br3);
      bset = &c->br3;
    }
  }
  P.W = c->W;
  P.H = c->H;
  P.znear = c->clip[0];
  shading_vectors(c, P);
  P.blend = c->blend;
  P.noise = c->d_noise;
  P.nn = c->nn;
  P.nn_log2 = -1;
  for (int l = 0; l < 12; ++l)
    if (c->nn == (1 << l)) P.nn_log2 = l;
  P.pw[0] = c->pw[0];
  P.pw[1] = c->pw[1];
  P.ps[0] = c->ps[0];
  P.ps[1] = c->ps[1];
  P.pert_on = (c->d_noise && (c->pw[0] != 0 || c->pw[1] != 0)) ? 1 : 0;
  if (P.pert_on && c->nranks > 1) {
    // a displaced fetch must stay inside region + halo
    for (int a = 0; a < 3; ++a) {
      int need = 1 + (int)ceil(0.5 * (fabs(c->pw[0]) + fabs(c->pw[1])) * c->N[a]);
      if (c->halo < need && c->D[a] < c->N[a])
        FAIL(c, "smk_render: perturbation needs halo >= %d voxels on a sharded volume (have %d); set option 'halo' before upload", need, c->halo);
    }
  }
  // a perturbed fetch lands within 0.5 (|w0| + |w1|) N voxels of the undisplaced position on every axis: when that is a
  // brick or two, flags spread that far tell BEFORE the noise lookups that nothing visible can be reached
  if (P.pert_on && P.bricks && bset) {
    int r[3];
    bool near = true;
    for (int a = 0; a < 3; ++a) {
      const int dv = (int)ceil(0.5 * (fabs(c->pw[0]) + fabs(c->pw[1])) * c->N[a] + 1e-3);
      r[a] = (dv + (1 << SMK_BRICK_LOG2) - 1) >> SMK_BRICK_LOG2;
      near = near && r[a] <= 2;
    }
    if (near) {
      const size_t nbricks = (size_t)c->nbr[0] * c->nbr[1] * c->nbr[2];
      if (bset->dil_cap < nbricks) {
        if (bset->dil) (void)hipFree(bset->dil);
        bset->dil = nullptr;
        bset->dil_cap = 0;
        HIPCHK(c, hipMalloc((void **)&bset->dil, nbricks));
        bset->dil_cap = nbricks;
        bset->dil_r[0] = -1;
      }
      if (bset->dil_r[0] != r[0] || bset->dil_r[1] != r[1] || bset->dil_r[2] != r[2]) {
        HIPCHK(c, smk_bricks_dilate(bset->flags, c->nbr, r, bset->dil, s));
        for (int a = 0; a < 3; ++a) bset->dil_r[a] = r[a];
      }
      P.bricks_dil = bset->dil;
    }
  }
  P.wave_w = c->opt_wave_w;
  P.blk_w = c->opt_blk_w;
  P.lockstep = c->opt_lockstep;
  int tw = P.wave_w * P.blk_w, th = (64 / P.wave_w) * (4 / P.blk_w);
  P.ntx = (c->W + tw - 1) / tw;
  P.nty = (c->H + th - 1) / th;
  P.tiles_per_xcd = (P.ntx * P.nty + 7) / 8;
  return 0;
}

// [z][y][x] -> [x][z][y] so that views along x also read contiguous rows (288 GB of HBM buys a
// second layout; only one copy is read per frame).  32x32 tiles through LDS keep both sides
// coalesced in 16-byte (f32: 2 x 8-byte) units.
template <class V>
__global__ __launch_bounds__(256) void smk_k_xmajor(const V *src, V *dst, int Dx, int Dy, int Dz) {
  __shared__ V tile[32][33];
  int z = blockIdx.z, x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    int x = x0 + tx, y = y0 + r;
    if (x < Dx && y < Dy) tile[r][tx] = src[((size_t)z * Dy + y) * Dx + x];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int x = x0 + r, y = y0 + tx;
    if (x < Dx && y < Dy) dst[((size_t)x * Dz + z) * Dy + y] = tile[tx][r];
  }
}

static int make_xmajor_copy(smk_ctx *c) {
  if (c->d_vox_x) return 0;
  HIPCHK(c, hipMalloc(&c->d_vox_x, c->vox_bytes));
  dim3 grid((c->D[0] + 31) / 32, (c->D[1] + 31) / 32, c->D[2]);
  if (c->dtype == SMK_U8)
    hipLaunchKernelGGL(smk_k_xmajor<uint2>, grid, dim3(256), 0, c->stream, (const uint2 *)c->d_vox, (uint2 *)c->d_vox_x,
                       c->D[0], c->D[1], c->D[2]);
  else
    hipLaunchKernelGGL(smk_k_xmajor<float4>, grid, dim3(256), 0, c->stream, (const float4 *)c->d_vox,
                       (float4 *)c->d_vox_x, c->D[0], c->D[1], c->D[2]);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

static int shade_kind_of(const smk_ctx *c) {
  if (c->tf_mode == 0 || !c->have_normals) return 0;
  if (c->shade == SMK_SHADE_R8K_DIFF || c->shade == SMK_SHADE_R8K_DSPEC) return 1;
  if (c->shade == SMK_SHADE_NV20_DIFF || c->shade == SMK_SHADE_NV20_DSPEC) return 2;
  return 0;
}

extern "C" int smk_render_device(smk_ctx *c, void *d_rgba, void *d_depth, void *stream) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (!d_rgba) FAIL(c, "smk_render_device: null output");
  RenderParams P;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  if (build_params(c, P, s)) return 1;
  P.out = (float4 *)d_rgba;
  P.depth = (float *)d_depth;
  // algorithmic bytes: every stored voxel once + TF + RGBA f32 frame
  size_t nst = (size_t)c->D[0] * c->D[1] * c->D[2];
  double bv = c->dtype == SMK_U8 ? (double)c->nelts : 4.0 * c->nelts;
  if (shade_kind_of(c)) bv += 3.0;
  double tfb = c->tf_mode == 0 ? 16.0 * c->tlut_size
               : c->tf_mode == 1 ? 4.0 * c->sv * c->sg * (P.third_axis ? 2 : 1)
                                 : 4.0 * c->s3v * c->s3g * c->s3h;
  c->last_alg_bytes = (double)nst * bv + tfb + 16.0 * c->W * c->H;
  if (c->tev0.empty()) {
    c->tev0.resize(SMK_TIMING_RING);
    c->tev1.resize(SMK_TIMING_RING);
    for (int i = 0; i < SMK_TIMING_RING; ++i) {
      HIPCHK(c, hipEventCreate(&c->tev0[i]));
      HIPCHK(c, hipEventCreate(&c->tev1[i]));
    }
  }
  int slot = (int)(c->tcount % SMK_TIMING_RING);
  c->ev0 = c->tev0[slot];
  c->ev1 = c->tev1[slot];
  // (ev0 is recorded by the launcher right before the kernel: host-side planning between the two
  //  events would otherwise count as kernel time whenever the stream is idle)
  // kernel choice: the slice-ring kernel when it applies (2-D / separable classification,
  // no perturbation, rays sharing one principal axis), the generic gather kernel otherwise
  c->last_kernel = 1;
  c->slab_why.clear();
  // The status word this frame takes over belongs to frame id - SMK_STATUS_RING: flagged, and nobody has asked about it
  // (smk_frame_failed) while they could -- the call fails.  Younger frames' words are left for their owners: a host that
  // pipelines frames asks about frame i AFTER enqueuing frame i + 1 (sortlast.Pipeline), and must find the word there.
  if (check_frame_status(c, c->frame_id + 1 - SMK_STATUS_RING)) return 1;
  ++c->frame_id;
  c->slab.status_slot = (int)(c->frame_id % SMK_STATUS_RING);
  c->slab.status_tag = c->cols.status_tag = (int)((c->frame_id & 0x7fffff) << 8);
  if (c->slab.h_status) ((volatile int *)c->slab.h_status)[c->slab.status_slot] = 0;
  bool ev0_recorded = false;  // (frames with shadows open the kernel-time bracket before their light march)
  if (c->shadow_on) {
    // ---- half-angle slicing (smk_shadow.hip): the light march, then the eye pass as an ordinary frame of the ray-marchers
    // below over the half-angle slices (SmkShadowRays) -- or, option shadow_march 0, a launch per slice
    const int sk = shade_kind_of(c);
    if (c->tf_mode == 0) FAIL(c, "smk_render: shadows need a 2-D or 3-D transfer function (the 1-D table renderer has no shadow mode)");
    if (sk == 2) FAIL(c, "smk_render: shadows are implemented for R8k shading or none (NV20 combiners: no shadow mode in NV20VolRen3D)");
    if (c->nranks > 1) FAIL(c, "smk_render: shadows need the whole volume on one GPU (the light buffer couples every slice of every brick)");
    if (P.pert_on || d_depth || c->region_on)
      FAIL(c, "smk_render: shadows cannot be combined with perturbation, a sub-box or depth output");
    if (c->opt_kernel == 3) FAIL(c, "smk_render: the column-stream kernel has no shadow mode");
    smk_shadowcoef sc;
    if (compute_shadowcoef(c, &sc)) return 1;
    // the eye rays over the half-angle slices, planes counted from the eye (smk_internal.h SmkShadowRays)
    SmkShadowRays &h = P.sh;
    memset(&h, 0, sizeof h);
    h.on = 1;
    for (int a = 0; a < 3; ++a) { h.Ec[a] = sc.Ec[a]; h.Dc[a] = sc.Dc[a]; h.Dx[a] = sc.Dx[a]; h.Dy[a] = sc.Dy[a]; }
    h.nDc = sc.nDc; h.nDx = sc.nDx; h.nDy = sc.nDy;
    if (sc.front_to_back) { h.numA = fmaf(1.0f, sc.dnum, sc.num0); h.dB = sc.dnum; h.k0 = 1; h.dk = 1; }
    else { h.numA = fmaf((float)sc.nslices, sc.dnum, sc.num0); h.dB = -sc.dnum; h.k0 = sc.nslices; h.dk = -1; }
    h.LB = sc.LB;
    for (int q = 0; q < 4; ++q) { h.Xm[q] = sc.Xm[q]; h.Ym[q] = sc.Ym[q]; h.Wm[q] = sc.Wm[q]; }
    h.lscale = sc.lscale; h.lbias = sc.lbias;
    {
      smk_raycoef &rc = P.rc;
      memset(&rc, 0, sizeof rc);
      rc.pxs = sc.pxs; rc.pxl = sc.pxl; rc.pys = sc.pys; rc.pyl = sc.pyl;
      rc.nplanes = sc.nslices;
      // (Bc: the central ray's step, which the kernel choice below keys its measurements on)
      const double nDc = (double)sc.nDc != 0.0 ? (double)sc.nDc : 1.0;
      for (int a = 0; a < 3; ++a) rc.Bc[a] = (float)((double)h.dB / nDc * (double)sc.Dc[a]);
    }
    // The last slice lies ON the volume's far corner -- on a whole face when the half-way vector is a volume axis (a light at
    // the eye) -- where a sample's coordinate, the end of an fma chain, lands on either side of the face by rounding.  The
    // reference draws that slice (a polygon clipped against the box keeps its boundary); the eye pass's membership test is
    // therefore 2^-10 voxels wide of the box (clamp-to-edge fetches: the value at the face).  The CPU checker does the same.
    // Clip planes (round 3): both passes draw the same clipped slice polygons in the reference (volShadow slices the box
    // setupClips left; glClipPlane stays enabled), so a light ray's sample must lie in the same box (closed, no slack: its
    // last slice gets no special treatment in rounds 1-2 either) and on the kept side of the free plane.
    for (int a = 0; a < 3; ++a) {
      h.llo[a] = P.lo[a];
      h.lhi[a] = P.hi[a];
      P.lo[a] -= SMK_SHADOW_BOX_EPS;
      P.hi[a] += SMK_SHADOW_BOX_EPS;
      P.hin[a] = P.hi[a];
      P.top[a] = 1;
    }
    P.blend = SMK_BLEND_FRONT_TO_BACK;  // (a light that faces the viewer: the per-slice form blends back to front, the marchers
                                        //  composite the same samples front to back -- the association of the blend differs)
    const size_t nl = (size_t)sc.LB * sc.LB;
    // The light march keeps every slice's light buffer: (nslices + 1) buffers.  Where that does not fit (more than a quarter
    // of the device's free memory, or 32 GB) the frame is a launch per slice, as with the option off.
    // (buffers 4 KiB + 256 B further apart than their size: 512^2 texels are exactly 4 MiB, and a wave of the light march
    //  stores to 8 consecutive buffers at once -- a power-of-two stride could put them all into the same memory channels;
    //  measured: 1.00 ms with the pad, 1.03 without, i.e. the fabric's address hash already spreads them)
    const size_t hstride = nl + 272;
    const size_t nhist = hstride * ((size_t)sc.nslices + 1);
    bool march = c->opt_shadow_march && !(c->opt_lockstep & 256) && sc.nslices > 0;
    if (march && nhist > c->light_hist_cap) {
      size_t fr = 0, tot = 0;
      if (c->d_light_hist) (void)hipFree(c->d_light_hist);
      c->d_light_hist = nullptr;
      c->light_hist_cap = 0;
      c->d_light_last = nullptr;
      if (hipMemGetInfo(&fr, &tot) != hipSuccess || nhist * 16 > fr / 4 || nhist * 16 > ((size_t)32 << 30) ||
          hipMalloc((void **)&c->d_light_hist, nhist * 16) != hipSuccess) {
        (void)hipGetLastError();
        c->d_light_hist = nullptr;
        march = false;
      } else c->light_hist_cap = nhist;
    }
    c->light_lb = sc.LB;
    if (march) {
      HIPCHK(c, hipEventRecord(c->ev0, s));
      ev0_recorded = true;
      hipError_t e = smk_launch_shadow_march(P, sc, c->dtype, c->tf_mode, c->d_light_hist, (long long)hstride, s);
      if (e == hipErrorNotSupported) FAIL(c, "smk_render: no shadow kernel instance for this configuration");
      HIPCHK(c, e);
      h.hist = c->d_light_hist;
      h.hstride = (long long)hstride;
      c->d_light_last = c->d_light_hist + (size_t)sc.nslices * hstride;
      // the history is written once (16 B per texel and slice)
      c->last_alg_bytes += (double)sc.nslices * (16.0 * (double)nl);
      // ... and the eye pass is the frame the code below renders
    } else {
      if (nl > c->light_cap) {
        for (int k = 0; k < 2; ++k) {
          if (c->d_light[k]) (void)hipFree(c->d_light[k]);
          c->d_light[k] = nullptr;
          HIPCHK(c, hipMalloc((void **)&c->d_light[k], nl * 16));
        }
        c->light_cap = nl;
        c->d_light_last = nullptr;
      }
      HIPCHK(c, hipEventRecord(c->ev0, s));
      HIPCHK(c, hipMemsetAsync(c->d_light[0], 0, nl * 16, s));
      HIPCHK(c, hipMemsetAsync(d_rgba, 0, (size_t)c->W * c->H * 16, s));
      if (!c->d_shadow_barrier) HIPCHK(c, hipMalloc((void **)&c->d_shadow_barrier, 16 * 9 * 4));  // (the common word + one per XCD, a cache line apart)
      hipError_t e = smk_launch_shadow(P, sc, c->dtype, c->tf_mode, sk, c->d_light[0], c->d_light[1], c->d_shadow_barrier, s);
      if (e == hipErrorNotSupported) FAIL(c, "smk_render: no shadow kernel instance for this configuration");
      HIPCHK(c, e);
      HIPCHK(c, hipEventRecord(c->ev1, s));
      c->d_light_last = c->d_light[sc.nslices & 1];
      // per slice the frame buffer (read + write where the slice covers it) and both light buffers move again
      c->last_alg_bytes += (double)sc.nslices * (32.0 * (double)nl);
      c->last_kernel = 3;
      if (c->tf_mode == 1 && c->tf_cur >= 0) {  // this frame read the current table version (refresh_tf2d waits for this before rewriting it)
        HIPCHK(c, hipEventRecord(c->tfv[c->tf_cur].used, s));
        c->tfv[c->tf_cur].used_valid = true;
      }
      c->tcount++;
      return 0;
    }
  }
  // ---- auto mode: which kernel for this configuration?
  bool try_slab = c->opt_kernel != 1;
  unsigned long long sig = 0;
  int trial = -1;  // 0 / 1: this frame is the slice-ring / gather trial of a new configuration
  if (c->opt_kernel == 0) {
    int as = 0;
    for (int a = 1; a < 3; ++a)
      if (fabsf(P.rc.Bc[a]) > fabsf(P.rc.Bc[as])) as = a;
    const unsigned long long f[] = {(unsigned long long)c->dtype, (unsigned long long)c->nelts, (unsigned long long)c->D[0],
                                    (unsigned long long)c->D[1], (unsigned long long)c->D[2], (unsigned long long)c->W,
                                    (unsigned long long)c->H, (unsigned long long)P.rc.nplanes, (unsigned long long)shade_kind_of(c),
                                    (unsigned long long)P.third_axis, (unsigned long long)(as * 2 + (P.rc.Bc[as] > 0)),
                                    (unsigned long long)c->sv, (unsigned long long)c->sg, (unsigned long long)(d_depth != nullptr),
                                    (unsigned long long)P.pert_on, (unsigned long long)c->tf_mode, (unsigned long long)c->s3v,
                                    (unsigned long long)c->s3g, (unsigned long long)c->s3h, (unsigned long long)c->blend,
                                    (unsigned long long)P.sh.on};
    sig = 1469598103934665603ull;
    for (unsigned long long v : f) sig = (sig ^ v) * 1099511628211ull;
    auto it = c->tune_choice.find(sig);
    if (it != c->tune_choice.end() && it->second.expires <= c->frame_id) {  // measured long ago: measure again
      c->tune_choice.erase(it);
      it = c->tune_choice.end();
      c->tune_sig = 0;
    }
    if (it != c->tune_choice.end()) {
      try_slab = it->second.kernel == 2;
    } else {
      if (c->tune_sig != sig) {
        c->tune_sig = sig;
        c->tune_state = 0;
      }
      // Trial frames of a new configuration: SMK_TUNE_SETTLE untimed slice-ring frames -- its schedule and depth cuts come
      // from the workgroup times of earlier frames, so each waits for the one before it (a one-time stall; without it, and
      // with a single untimed frame, the timed trial was the first frame with cuts, 1.14 ms on a 1/8 shard that settles at
      // 0.15, and auto mode kept the 0.69 ms gather kernel for the shard) --, one untimed gather frame, then the timed pair.
      if (c->tune_state == SMK_TUNE_SETTLE + 3) {  // both timed trials issued: decide once their events have completed
        float ms_s = 0, ms_g = 0;
        // (a host that enqueues frames far ahead of the GPU would recycle the trials' event pairs -- the ring holds the
        //  last 64 frames -- before they complete, and the comparison would then be between two later frames of the
        //  same kernel: wait for the trials rather than let their slots go)
        if (c->tcount - c->tune_tcount >= SMK_TIMING_RING - 8) (void)hipEventSynchronize(c->tev1[c->tune_slot[1]]);
        if (hipEventQuery(c->tev1[c->tune_slot[0]]) == hipSuccess && hipEventQuery(c->tev1[c->tune_slot[1]]) == hipSuccess &&
            hipEventElapsedTime(&ms_s, c->tev0[c->tune_slot[0]], c->tev1[c->tune_slot[0]]) == hipSuccess &&
            hipEventElapsedTime(&ms_g, c->tev0[c->tune_slot[1]], c->tev1[c->tune_slot[1]]) == hipSuccess) {
          c->tune_choice[sig] = {ms_s <= ms_g ? 2 : 1, c->frame_id + 1024};
          try_slab = ms_s <= ms_g;
          if (getenv("SMK_DEBUG")) fprintf(stderr, "[smk] auto mode: slice-ring %.3f ms, gather %.3f ms (frame %lld)\n", ms_s, ms_g, c->frame_id);
        }  // else: keep the slice-ring kernel for this frame and ask again
        (void)hipGetLastError();
      } else {
        trial = c->tune_state;
        try_slab = trial != SMK_TUNE_SETTLE && trial != SMK_TUNE_SETTLE + 2;
        if (try_slab && trial > 0) (void)hipStreamSynchronize(s);  // (the previous settle frame's workgroup times are back)
      }
    }
  }
  if (c->opt_kernel == 3) {
    // ---- the column-stream kernel, forced (smk_cols.hip)
    try_slab = false;
    if (!c->slab.h_status) {
      HIPCHK(c, hipHostMalloc((void **)&c->slab.h_status, SMK_STATUS_RING * sizeof(int), hipHostMallocMapped));
      for (int k = 0; k < SMK_STATUS_RING; ++k) c->slab.h_status[k] = 0;
      HIPCHK(c, hipMalloc((void **)&c->slab.d_diag, 16 * sizeof(float)));
    }
    const char *why = nullptr;
    c->cols.frame_ev0 = c->ev0;
    hipError_t e = smk_launch_cols(P, c->dtype, c->tf_mode, shade_kind_of(c), c->opt_cols, c->d_vox, &c->cols,
                                   c->slab.h_status + c->slab.status_slot, &why, s);
    if (e == hipErrorNotSupported) {
      c->slab_why = why ? why : "?";
      FAIL(c, "smk_render: column-stream kernel forced but not applicable: %s", c->slab_why.c_str());
    }
    HIPCHK(c, e);
    c->last_kernel = 4;
    if (c->opt_inject_status && c->slab.h_status) {
      ((volatile int *)c->slab.h_status)[c->slab.status_slot] = c->slab.status_tag | c->opt_inject_status;
      c->opt_inject_status = 0;
    }
  }
  if (try_slab) {
    if (!c->slab.h_status) {
      HIPCHK(c, hipHostMalloc((void **)&c->slab.h_status, SMK_STATUS_RING * sizeof(int), hipHostMallocMapped));
      for (int k = 0; k < SMK_STATUS_RING; ++k) c->slab.h_status[k] = 0;
      HIPCHK(c, hipMalloc((void **)&c->slab.d_diag, 16 * sizeof(float)));
    }
    if (c->opt_lockstep & 16) HIPCHK(c, hipMemsetAsync(c->slab.d_diag, 0, 16 * sizeof(float), s));
    const char *why = nullptr;
    const int forced = c->opt_kernel == 2;
    c->slab.frame_ev0 = ev0_recorded ? nullptr : c->ev0;
    const int knobs = c->opt_slab_T | (c->opt_slab_fly << 8) | (c->opt_slab_ns << 16) | (c->opt_slab_sched << 24);
    hipError_t e = smk_launch_slab(P, c->dtype, c->tf_mode, shade_kind_of(c), knobs, c->opt_tile, forced, c->d_vox, c->d_vox_x, &c->slab, &why, s);
    if (e == hipErrorNotSupported && why && !strcmp(why, "x-major copy unavailable")) {
      if (make_xmajor_copy(c)) return 1;
      e = smk_launch_slab(P, c->dtype, c->tf_mode, shade_kind_of(c), knobs, c->opt_tile, forced, c->d_vox, c->d_vox_x, &c->slab, &why, s);
    }
    if (e == hipSuccess) {
      c->last_kernel = 2;
      c->last_slab_sig = sig;
      if (c->opt_inject_status && c->slab.h_status) {  // (test hook: what a failed frame leaves behind)
        ((volatile int *)c->slab.h_status)[c->slab.status_slot] = c->slab.status_tag | c->opt_inject_status;
        c->opt_inject_status = 0;
      }
    }
    else if (e == hipErrorNotSupported) {
      c->slab_why = why ? why : "?";
      if (c->opt_kernel == 2) FAIL(c, "smk_render: slab kernel forced but not applicable: %s", c->slab_why.c_str());
      // Not remembered: most reasons depend on the pose (too oblique, window does not fit LDS, ...)
      // and the signature does not; the next frame is planned afresh (planning runs per frame anyway)
      // and returns to the slice-ring kernel as soon as the view allows.
      if (c->opt_kernel == 0) trial = -1;
    } else
      HIPCHK(c, e);
  } else if (c->opt_kernel == 2) {
    FAIL(c, "smk_render: slab kernel forced but classification mode %d is gather-only", c->tf_mode);
  }
  if (c->last_kernel == 1) {
    if (!ev0_recorded) HIPCHK(c, hipEventRecord(c->ev0, s));
    HIPCHK(c, smk_launch_gather(P, c->dtype, c->tf_mode, shade_kind_of(c), s));
  }
  HIPCHK(c, hipEventRecord(c->ev1, s));
  if (trial >= 0) {
    if (trial >= SMK_TUNE_SETTLE + 1) {
      c->tune_slot[trial - (SMK_TUNE_SETTLE + 1)] = slot;
      c->tune_tcount = c->tcount;
    }
    c->tune_state = trial + 1;
  }
  if (c->tf_mode == 1 && c->tf_cur >= 0) {  // this frame read the current table version (refresh_tf2d waits for this before rewriting it)
    HIPCHK(c, hipEventRecord(c->tfv[c->tf_cur].used, s));
    c->tfv[c->tf_cur].used_valid = true;
  }
  c->tcount++;
  return 0;
}

extern "C" int smk_render(smk_ctx *c, float *rgba, float *depth) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (!rgba) FAIL(c, "smk_render: null output");
  if (!c->have_camera) FAIL(c, "smk_render: no camera set");
  size_t npix = (size_t)c->W * c->H;
  if (npix > c->out_cap) {
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_depth) (void)hipFree(c->d_depth);
    c->d_out = nullptr;
    c->d_depth = nullptr;
    HIPCHK(c, hipMalloc((void **)&c->d_out, npix * 16));
    HIPCHK(c, hipMalloc((void **)&c->d_depth, npix * 4));
    c->out_cap = npix;
  }
  if (smk_render_device(c, c->d_out, depth ? c->d_depth : nullptr, c->stream)) return 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
  if (check_frame_status(c, c->frame_id)) {
    // the synchronous entry still owes its caller this frame: in auto mode it is rendered again, by
    // the gather kernel (take_status has just retired the slice-ring kernel for this configuration)
    if (c->opt_kernel != 0) return 1;
    const std::string first = c->err;
    ++c->slab_retries;
    if (smk_render_device(c, c->d_out, depth ? c->d_depth : nullptr, c->stream)) return 1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->last_kernel != 1 || check_frame_status(c, c->frame_id)) {
      c->err = first;
      return 1;
    }
    fprintf(stderr, "[smk] %s -- frame rendered again by the gather kernel\n", first.c_str());
  }
  HIPCHK(c, hipMemcpy(rgba, c->d_out, npix * 16, hipMemcpyDeviceToHost));
  if (depth) HIPCHK(c, hipMemcpy(depth, c->d_depth, npix * 4, hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------------------------------------------------------- renderSlice

// VolumeRenderer::render3dSliceEXT (VolumeRenderer.cpp:762-807): one textured quad blended into the frame.  A thread per
// pixel: the ray through the pixel centre against the quad's two triangles (vertices in GL's order 1, 0, 2, 3 -> (a, b, c),
// (a, c, d)); texture coordinates are vertex / fSize, i.e. linear in the model-space position, so the perspective-correct
// interpolation GL does IS the hit point.  Sampling as the ray-marchers do: GL_LINEAR, clamp to edge, channel 0.
struct SliceArg {
  float v[4][3];   // quad vertices in voxel coordinates (x / fSize * N - 0.5), order a, b, c, d
  float alpha;
};

template <int DT>
__global__ __launch_bounds__(256) void smk_k_render_slice(const RenderParams P, const SliceArg Q, float4 *fb) {
  const int i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= P.W || j >= P.H) return;
  const smk_raycoef &rc = P.rc;
  const float px = __fmaf_rn((float)i + 0.5f, rc.pxs, rc.pxl), py = __fmaf_rn((float)j + 0.5f, rc.pys, rc.pyl);
  float A[3], B[3];
  for (int a = 0; a < 3; ++a) {
    A[a] = __fmaf_rn(px, rc.Ax[a], __fmaf_rn(py, rc.Ay[a], rc.Ac[a]));
    B[a] = __fmaf_rn(px, rc.Bx[a], __fmaf_rn(py, rc.By[a], rc.Bc[a]));
  }
  // eye = A - (tau0 / dtau) B (every ray starts there); direction B
  const float k = rc.tau0 / rc.dtau;
  const float O[3] = {A[0] - k * B[0], A[1] - k * B[1], A[2] - k * B[2]};
  bool hit = false;
  float X[3] = {0, 0, 0};
  for (int t = 0; t < 2 && !hit; ++t) {
    const float *v0 = Q.v[0], *v1 = Q.v[t == 0 ? 1 : 2], *v2 = Q.v[t == 0 ? 2 : 3];
    const float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]}, e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    const float pv[3] = {B[1] * e2[2] - B[2] * e2[1], B[2] * e2[0] - B[0] * e2[2], B[0] * e2[1] - B[1] * e2[0]};
    const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    if (fabsf(det) < 1e-20f) continue;
    const float inv = 1.0f / det;
    const float tv[3] = {O[0] - v0[0], O[1] - v0[1], O[2] - v0[2]};
    const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
    const float qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
    const float w = (B[0] * qv[0] + B[1] * qv[1] + B[2] * qv[2]) * inv;
    const float tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
    if (u < 0.0f || w < 0.0f || u + w > 1.0f || !(tt * rc.dtau > 0.0f)) continue;  // (in front of the eye)
    hit = true;
    for (int a = 0; a < 3; ++a) X[a] = v0[a] + u * e1[a] + w * e2[a];
  }
  if (!hit) return;
  int x0, x1, y0, y1, z0, z1;
  float fx, fy, fz;
  smk_lin_clamp(X[0], P.N[0], x0, x1, fx);
  smk_lin_clamp(X[1], P.N[1], y0, y1, fy);
  smk_lin_clamp(X[2], P.N[2], z0, z1, fz);
  x0 = min(max(x0 - P.O[0], 0), P.D[0] - 1); x1 = min(max(x1 - P.O[0], 0), P.D[0] - 1);
  y0 = min(max(y0 - P.O[1], 0), P.D[1] - 1); y1 = min(max(y1 - P.O[1], 0), P.D[1] - 1);
  z0 = min(max(z0 - P.O[2], 0), P.D[2] - 1); z1 = min(max(z1 - P.O[2], 0), P.D[2] - 1);
  const size_t Dx = P.D[0], Dy = P.D[1];
  auto at = [&](int x, int y, int z) -> float { return smk_load_corner<DT>(P, ((size_t)z * Dy + y) * Dx + x).c0; };
  float I = smk_lerp(smk_lerp(smk_lerp(at(x0, y0, z0), at(x1, y0, z0), fx), smk_lerp(at(x0, y1, z0), at(x1, y1, z0), fx), fy),
                     smk_lerp(smk_lerp(at(x0, y0, z1), at(x1, y0, z1), fx), smk_lerp(at(x0, y1, z1), at(x1, y1, z1), fx), fy), fz);
  if (DT == 0) I *= SMK_INV255;
  I = smk_sat(I);
  const float sa = smk_sat(I * Q.alpha);
  float4 D = fb[(size_t)j * P.W + i];
  const float w1 = 1.0f - sa;
  D.x = __fmaf_rn(w1, D.x, I);
  D.y = __fmaf_rn(w1, D.y, I);
  D.z = __fmaf_rn(w1, D.z, I);
  D.w = __fmaf_rn(w1, D.w, sa);
  fb[(size_t)j * P.W + i] = D;
}

extern "C" int smk_render_slice_device(smk_ctx *c, const float quad[4][3], float alpha, void *d_rgba, void *stream) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (!quad || !d_rgba) FAIL(c, "smk_render_slice: null argument");
  if (!c->have_volume) FAIL(c, "smk_render_slice: no volume uploaded");
  if (!c->have_camera) FAIL(c, "smk_render_slice: no camera set");
  if (c->nranks > 1) FAIL(c, "smk_render_slice: the slice quad is drawn from the whole volume (unsharded context)");
  RenderParams P;
  memset(&P, 0, sizeof P);
  double inv[16];
  compute_raycoef(c, &P.rc, inv);
  P.vox = c->d_vox;
  P.nrm = c->d_nrm;
  for (int a = 0; a < 3; ++a) { P.N[a] = c->N[a]; P.O[a] = c->O[a]; P.D[a] = c->D[a]; }
  P.nelts = c->nelts;
  P.n_in_w = (c->dtype == SMK_F32 && c->nelts <= 3) ? 1 : 0;
  P.W = c->W;
  P.H = c->H;
  SliceArg Q;
  const int ord[4] = {1, 0, 2, 3};  // glVertex order of render3dSliceEXT (:776-792)
  for (int k = 0; k < 4; ++k)
    for (int a = 0; a < 3; ++a) Q.v[k][a] = (float)((double)quad[ord[k]][a] / (double)c->fsize[a] * (double)c->N[a] - 0.5);
  Q.alpha = alpha;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  dim3 grid((c->W + 15) / 16, (c->H + 15) / 16);
  if (c->dtype == SMK_U8) hipLaunchKernelGGL(smk_k_render_slice<0>, grid, dim3(256), 0, s, P, Q, (float4 *)d_rgba);
  else hipLaunchKernelGGL(smk_k_render_slice<1>, grid, dim3(256), 0, s, P, Q, (float4 *)d_rgba);
  HIPCHK(c, hipGetLastError());
  return 0;
}

extern "C" int smk_render_slice(smk_ctx *c, const float quad[4][3], float alpha, float *rgba) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (!rgba) FAIL(c, "smk_render_slice: null frame");
  if (!c->have_camera) FAIL(c, "smk_render_slice: no camera set");
  const size_t npix = (size_t)c->W * c->H;
  if (npix > c->out_cap) {
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_depth) (void)hipFree(c->d_depth);
    c->d_out = nullptr;
    c->d_depth = nullptr;
    HIPCHK(c, hipMalloc((void **)&c->d_out, npix * 16));
    HIPCHK(c, hipMalloc((void **)&c->d_depth, npix * 4));
    c->out_cap = npix;
  }
  HIPCHK(c, hipMemcpy(c->d_out, rgba, npix * 16, hipMemcpyHostToDevice));
  if (smk_render_slice_device(c, quad, alpha, c->d_out, c->stream)) return 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(rgba, c->d_out, npix * 16, hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------------------------------------------------------- sort-last merge

struct OrderArg {
  int o[SMK_MAX_RANKS];
};

__global__ void smk_k_over(const float4 *layers, int nlayers, OrderArg ord, int npix, float4 *out, int use_max) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  float4 C = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l = 0; l < nlayers; ++l) {
    float4 s = layers[(size_t)ord.o[l] * npix + p];
    if (use_max) {  // GL_MAX layers merge by maximum, in any order
      C = make_float4(fmaxf(C.x, s.x), fmaxf(C.y, s.y), fmaxf(C.z, s.z), fmaxf(C.w, s.w));
      continue;
    }
    float w = 1.0f - C.w;
    C.x = __fmaf_rn(w, s.x, C.x);
    C.y = __fmaf_rn(w, s.y, C.y);
    C.z = __fmaf_rn(w, s.z, C.z);
    C.w = __fmaf_rn(w, s.w, C.w);
  }
  out[p] = C;
}

extern "C" int smk_composite_over_device(smk_ctx *c, const void *d_layers, int nlayers, const int *order, int npix,
                                         void *d_out, void *stream) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (!d_layers || !d_out || !order) FAIL(c, "smk_composite_over_device: null argument");
  if (nlayers < 1 || nlayers > SMK_MAX_RANKS) FAIL(c, "smk_composite_over_device: 1..%d layers", SMK_MAX_RANKS);
  OrderArg oa;
  for (int l = 0; l < nlayers; ++l) {
    if (order[l] < 0 || order[l] >= nlayers) FAIL(c, "smk_composite_over_device: order[%d]=%d out of range", l, order[l]);
    oa.o[l] = order[l];
  }
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  hipLaunchKernelGGL(smk_k_over, dim3((npix + 255) / 256), dim3(256), 0, s, (const float4 *)d_layers, nlayers, oa,
                     npix, (float4 *)d_out, c->blend == SMK_BLEND_MAX ? 1 : 0);
  HIPCHK(c, hipGetLastError());
  return 0;
}
