// smk_exchange.hip -- sort-last merge of the ranks' layers behind the C ABI: direct
// send of 1/P image tiles, ordered "over" of the P layers, gather of the finished tiles on rank 0.
//
// The reference has no distributed code (its only decomposition is MetaVolume::brick's grid drawn
// serially in visibility order on one GPU, MetaVolume.cpp:1369-1452, NV20VolRen3D.cpp:190-231); this
// is the MI355X-native equivalent for one node of 8 GPUs on an xGMI full mesh.  "over" is associative
// but not commutative, so the merge is NOT an all-reduce: every rank sends tile r of its layer to rank
// r (grouped ncclSend/ncclRecv: all 7 links of a GPU busy at once), composites the P layers of its own
// tile in the BSP visibility order of the shards (smk_shard_order) and sends the finished tile to rank 0.
//
// Two transports behind the same entry points:
//   RCCL        one process per GPU (the host hands the 128-byte communicator id from rank 0 to the
//               others by whatever channel it has).  librccl is opened at run time (dlopen), so the
//               library itself loads where RCCL is not installed.
//   in-process  the ranks are contexts of ONE host process (a C++ host that owns several GPUs, or the
//               tests' two contexts on one GPU): tiles move by peer copies, events order the streams.
// Product code: nothing here renders or falls back to the CPU.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "smk_internal.h"

namespace {

// the handful of RCCL entry points used, declared here with the published NCCL signatures
// (rccl.h: ncclResult_t is an int enum with ncclSuccess = 0, ncclFloat32 = 7, ncclUniqueId = 128 bytes by value)
struct NcclId { char internal[SMK_EXCHANGE_ID_BYTES]; };
struct Rccl {
  void *handle = nullptr;
  int (*GetUniqueId)(NcclId *) = nullptr;
  int (*CommInitRank)(void **, int, NcclId, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string why;
};
const int kNcclFloat = 7;

Rccl *rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.handle ? &r : nullptr;
  tried = true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names)
    if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!r.handle) {
    r.why = "librccl.so.1 could not be opened";
    return nullptr;
  }
#define SYM(field, name)                                        \
  *(void **)(&r.field) = dlsym(r.handle, name);                 \
  if (!r.field) {                                               \
    r.why = std::string("librccl lacks ") + name;               \
    r.handle = nullptr;                                         \
    return nullptr;                                             \
  }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
  SYM(Send, "ncclSend")
  SYM(Recv, "ncclRecv")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  return &r;
}

}  // namespace

struct smk_exchange {
  smk_ctx *ctx = nullptr;
  int rank = 0, nranks = 1, npix = 0, tp = 0;  // tp = pixels per tile = ceil(npix / nranks)
  bool local = true;
  void *comm = nullptr;                        // ncclComm_t
  std::vector<smk_exchange *> peers;           // in-process transport: every rank's object, by rank
  hipStream_t xs = nullptr;                    // the exchange's own stream: frame i's merge overlaps frame i+1's ray-marching
  float4 *partial[2] = {nullptr, nullptr};     // [nranks * tp] this rank's layer, one per frame slot
  float4 *recv[2] = {nullptr, nullptr};        // [nranks][tp] tile `rank` of every rank's layer
  float4 *tile[2] = {nullptr, nullptr};        // [tp] the finished tile
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_sent[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
  bool used[2] = {false, false};
  // visibility order of the ranks for the frame in each slot, taken when the frame was RENDERED (smk_exchange_rendered): the
  // merge is enqueued a frame later, when the context's camera is already the next frame's
  int order[2][SMK_MAX_RANKS] = {};
  bool order_valid[2] = {false, false};
  std::string err;
  int cnt(int r) const {  // pixels of tile r that exist (the last tile may be short)
    const long long left = (long long)npix - (long long)r * tp;
    return left <= 0 ? 0 : (int)(left < tp ? left : tp);
  }
};

static std::string g_exchange_err;

#define XFAIL(x, ...)                       \
  do {                                      \
    char b_[512];                           \
    snprintf(b_, sizeof b_, __VA_ARGS__);   \
    (x)->err = b_;                          \
    (x)->ctx->err = b_;                     \
    return 1;                               \
  } while (0)
#define XHIP(x, call)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) XFAIL(x, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define XNCCL(x, call)                                                                   \
  do {                                                                                   \
    int e_ = (call);                                                                     \
    if (e_ != 0) XFAIL(x, "%s failed: %s", #call, rccl()->GetErrorString(e_));           \
  } while (0)

extern "C" int smk_exchange_unique_id(unsigned char *id) {
  if (!id) return 1;
  Rccl *r = rccl();
  if (!r) return 1;
  NcclId nid;
  if (r->GetUniqueId(&nid) != 0) return 1;
  memcpy(id, nid.internal, SMK_EXCHANGE_ID_BYTES);
  return 0;
}

extern "C" const char *smk_exchange_last_error(smk_exchange *x) { return x ? x->err.c_str() : g_exchange_err.c_str(); }

extern "C" void smk_exchange_destroy(smk_exchange *x) {
  if (!x) return;
  (void)hipSetDevice(x->ctx->device);
  if (x->xs) (void)hipStreamSynchronize(x->xs);
  if (x->comm && rccl()) (void)rccl()->CommDestroy(x->comm);
  for (int s = 0; s < 2; ++s) {
    if (x->partial[s]) (void)hipFree(x->partial[s]);
    if (x->recv[s]) (void)hipFree(x->recv[s]);
    if (x->tile[s]) (void)hipFree(x->tile[s]);
    if (x->ev_in[s]) (void)hipEventDestroy(x->ev_in[s]);
    if (x->ev_sent[s]) (void)hipEventDestroy(x->ev_sent[s]);
    if (x->ev_done[s]) (void)hipEventDestroy(x->ev_done[s]);
  }
  if (x->xs) (void)hipStreamDestroy(x->xs);
  delete x;
}

extern "C" smk_exchange *smk_exchange_create(smk_ctx *ctx, int rank, int nranks, const unsigned char *id, int npix, int *err) {
  if (err) *err = 1;
  if (!ctx) {
    g_exchange_err = "smk_exchange_create: no context";
    return nullptr;
  }
  if (nranks < 1 || nranks > SMK_MAX_RANKS || rank < 0 || rank >= nranks || npix <= 0) {
    g_exchange_err = ctx->err = "smk_exchange_create: 1..8 ranks, 0 <= rank < nranks, npix > 0";
    return nullptr;
  }
  if (ctx->nranks != nranks || ctx->rank != rank) {
    g_exchange_err = ctx->err = "smk_exchange_create: rank / nranks differ from the context's shard (smk_set_shard)";
    return nullptr;
  }
  if (hipSetDevice(ctx->device) != hipSuccess) {
    g_exchange_err = ctx->err = "smk_exchange_create: hipSetDevice failed";
    return nullptr;
  }
  smk_exchange *x = new smk_exchange();
  x->ctx = ctx;
  x->rank = rank;
  x->nranks = nranks;
  x->npix = npix;
  x->tp = (npix + nranks - 1) / nranks;
  x->local = id == nullptr;
  const size_t layer = (size_t)x->tp * nranks * sizeof(float4);
  bool ok = hipStreamCreateWithFlags(&x->xs, hipStreamNonBlocking) == hipSuccess;
  for (int s = 0; s < 2 && ok; ++s) {
    ok = hipMalloc((void **)&x->partial[s], layer) == hipSuccess && hipMalloc((void **)&x->recv[s], layer) == hipSuccess &&
         hipMalloc((void **)&x->tile[s], (size_t)x->tp * sizeof(float4)) == hipSuccess &&
         hipMemset(x->partial[s], 0, layer) == hipSuccess && hipMemset(x->recv[s], 0, layer) == hipSuccess &&
         hipEventCreateWithFlags(&x->ev_in[s], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&x->ev_sent[s], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&x->ev_done[s], hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) {
    g_exchange_err = ctx->err = "smk_exchange_create: device allocation failed";
    smk_exchange_destroy(x);
    return nullptr;
  }
  if (!x->local) {
    Rccl *r = rccl();
    if (!r) {
      g_exchange_err = ctx->err = "smk_exchange_create: RCCL transport unavailable (librccl.so.1 not loadable)";
      smk_exchange_destroy(x);
      return nullptr;
    }
    NcclId nid;
    memcpy(nid.internal, id, SMK_EXCHANGE_ID_BYTES);
    const int e = r->CommInitRank(&x->comm, nranks, nid, rank);  // collective: every rank calls it
    if (e != 0) {
      g_exchange_err = ctx->err = std::string("smk_exchange_create: ncclCommInitRank failed: ") + r->GetErrorString(e);
      x->comm = nullptr;
      smk_exchange_destroy(x);
      return nullptr;
    }
  }
  if (err) *err = 0;
  return x;
}

extern "C" int smk_exchange_connect_local(smk_exchange *const *all, int nranks) {
  if (!all || nranks < 1) return 1;
  for (int r = 0; r < nranks; ++r) {
    if (!all[r]) return 1;
    if (!all[r]->local || all[r]->nranks != nranks || all[r]->rank != r || all[r]->npix != all[0]->npix)
      XFAIL(all[r], "smk_exchange_connect_local: object %d is not rank %d of %d in-process ranks of one frame size", r, r, nranks);
  }
  for (int r = 0; r < nranks; ++r) {
    all[r]->peers.assign(all, all + nranks);
    // peer copies between different devices want peer access where the platform offers it (xGMI)
    for (int p = 0; p < nranks; ++p)
      if (all[p]->ctx->device != all[r]->ctx->device) {
        int can = 0;
        (void)hipSetDevice(all[r]->ctx->device);
        if (hipDeviceCanAccessPeer(&can, all[r]->ctx->device, all[p]->ctx->device) == hipSuccess && can)
          (void)hipDeviceEnablePeerAccess(all[p]->ctx->device, 0);
        (void)hipGetLastError();  // (already enabled is fine)
      }
  }
  return 0;
}

extern "C" void *smk_exchange_partial(smk_exchange *x, int slot) { return x && slot >= 0 && slot < 2 ? x->partial[slot] : nullptr; }

// before rendering into partial(slot) again: the frame that used it two frames ago must have been sent
extern "C" int smk_exchange_acquire(smk_exchange *x, int slot, void *render_stream) {
  if (!x || slot < 0 || slot > 1) return 1;
  x->order_valid[slot] = false;  // a new frame goes into the slot: its order is taken at smk_exchange_rendered
  if (!x->used[slot]) return 0;
  XHIP(x, hipSetDevice(x->ctx->device));
  XHIP(x, hipStreamWaitEvent(render_stream ? (hipStream_t)render_stream : x->ctx->stream, x->ev_sent[slot], 0));
  return 0;
}

extern "C" int smk_exchange_set_order(smk_exchange *x, int slot, const int *order) {
  if (!x || slot < 0 || slot > 1 || !order) return 1;
  for (int r = 0; r < x->nranks; ++r) {
    if (order[r] < 0 || order[r] >= x->nranks) XFAIL(x, "smk_exchange_set_order: order[%d] = %d out of range", r, order[r]);
    x->order[slot][r] = order[r];
  }
  x->order_valid[slot] = true;
  return 0;
}

// the ray-marcher's work for frame `slot` is in `render_stream` up to here: the merge waits for exactly
// that (and not for a later frame's ray-marching enqueued before the merge is)
extern "C" int smk_exchange_rendered(smk_exchange *x, int slot, void *render_stream) {
  if (!x || slot < 0 || slot > 1) return 1;
  XHIP(x, hipSetDevice(x->ctx->device));
  // (NULL = the context's own stream, as in smk_render_device)
  XHIP(x, hipEventRecord(x->ev_in[slot], render_stream ? (hipStream_t)render_stream : x->ctx->stream));
  // the shards' visibility order under the camera the frame was rendered with -- once per frame: a second mark of the
  // same slot (the host re-rendered a flagged frame, possibly after setting the next pose) keeps the first one's
  if (!x->order_valid[slot]) {
    if (smk_shard_order(x->ctx, x->order[slot])) XFAIL(x, "smk_exchange_rendered: %s", x->ctx->err.c_str());
    x->order_valid[slot] = true;
  }
  return 0;
}

// this rank's layers of its own tile, in visibility order, -> the finished tile
static int merge_tile(smk_exchange *x, int slot) {
  int order[SMK_MAX_RANKS];
  if (x->order_valid[slot]) memcpy(order, x->order[slot], sizeof order);
  else if (smk_shard_order(x->ctx, order)) XFAIL(x, "smk_exchange: %s", x->ctx->err.c_str());  // (a frame nobody marked: the current camera)
  if (smk_composite_over_device(x->ctx, x->recv[slot], x->nranks, order, x->tp, x->tile[slot], x->xs))
    XFAIL(x, "smk_exchange: %s", x->ctx->err.c_str());
  return 0;
}

extern "C" int smk_exchange_frame(smk_exchange *x, int slot, void *d_frame) {
  if (!x || slot < 0 || slot > 1) return 1;
  if (x->local && x->nranks > 1) XFAIL(x, "smk_exchange_frame: in-process ranks exchange through smk_exchange_frame_local");
  if (x->rank == 0 && !d_frame) XFAIL(x, "smk_exchange_frame: rank 0 needs the frame buffer");
  XHIP(x, hipSetDevice(x->ctx->device));
  const int P = x->nranks, me = x->rank, tp = x->tp;
  XHIP(x, hipStreamWaitEvent(x->xs, x->ev_in[slot], 0));  // (smk_exchange_rendered)
  // ---- direct send: tile p of my layer goes to rank p, tile `me` of everyone's comes here
  if (P > 1) {
    Rccl *r = rccl();
    XNCCL(x, r->GroupStart());
    for (int p = 0; p < P; ++p) {
      if (p == me) continue;
      XNCCL(x, r->Send(x->partial[slot] + (size_t)p * tp, (size_t)tp * 4, kNcclFloat, p, x->comm, x->xs));
      XNCCL(x, r->Recv(x->recv[slot] + (size_t)p * tp, (size_t)tp * 4, kNcclFloat, p, x->comm, x->xs));
    }
    XNCCL(x, r->GroupEnd());
  }
  XHIP(x, hipMemcpyAsync(x->recv[slot] + (size_t)me * tp, x->partial[slot] + (size_t)me * tp, (size_t)tp * sizeof(float4), hipMemcpyDeviceToDevice, x->xs));
  XHIP(x, hipEventRecord(x->ev_sent[slot], x->xs));
  x->used[slot] = true;
  // ---- ordered over of the P layers of my tile
  if (merge_tile(x, slot)) return 1;
  // ---- finished tiles to rank 0
  if (me == 0)
    XHIP(x, hipMemcpyAsync((float4 *)d_frame, x->tile[slot], (size_t)x->cnt(0) * sizeof(float4), hipMemcpyDeviceToDevice, x->xs));
  if (P > 1) {
    Rccl *r = rccl();
    XNCCL(x, r->GroupStart());
    if (me != 0) {
      if (x->cnt(me) > 0) XNCCL(x, r->Send(x->tile[slot], (size_t)x->cnt(me) * 4, kNcclFloat, 0, x->comm, x->xs));
    } else {
      for (int p = 1; p < P; ++p)
        if (x->cnt(p) > 0) XNCCL(x, r->Recv((float4 *)d_frame + (size_t)p * tp, (size_t)x->cnt(p) * 4, kNcclFloat, p, x->comm, x->xs));
    }
    XNCCL(x, r->GroupEnd());
  }
  XHIP(x, hipEventRecord(x->ev_done[slot], x->xs));
  return 0;
}

extern "C" int smk_exchange_frame_local(smk_exchange *const *all, int nranks, int slot, void *d_frame) {
  if (!all || nranks < 1 || !all[0] || slot < 0 || slot > 1) return 1;
  smk_exchange *x0 = all[0];
  if (!d_frame) XFAIL(x0, "smk_exchange_frame_local: no frame buffer");
  for (int r = 0; r < nranks; ++r)
    if (!all[r] || !all[r]->local || (int)all[r]->peers.size() != nranks || all[r]->peers[r] != all[r])
      XFAIL(x0, "smk_exchange_frame_local: ranks not connected (smk_exchange_connect_local)");
  const int tp = x0->tp;
  // ---- every rank puts tile p of its layer into rank p's receive buffer
  for (int r = 0; r < nranks; ++r) {
    smk_exchange *x = all[r];
    XHIP(x, hipSetDevice(x->ctx->device));
    XHIP(x, hipStreamWaitEvent(x->xs, x->ev_in[slot], 0));  // (smk_exchange_rendered)
    // (the receive buffers of this slot are free once the merge of two frames ago has run)
    for (int p = 0; p < nranks; ++p)
      if (all[p]->used[slot]) XHIP(x, hipStreamWaitEvent(x->xs, all[p]->ev_done[slot], 0));
    for (int p = 0; p < nranks; ++p) {
      smk_exchange *y = all[p];
      float4 *dst = y->recv[slot] + (size_t)r * tp;
      const float4 *src = x->partial[slot] + (size_t)p * tp;
      if (y->ctx->device == x->ctx->device)
        XHIP(x, hipMemcpyAsync(dst, src, (size_t)tp * sizeof(float4), hipMemcpyDeviceToDevice, x->xs));
      else
        XHIP(x, hipMemcpyPeerAsync(dst, y->ctx->device, src, x->ctx->device, (size_t)tp * sizeof(float4), x->xs));
    }
    XHIP(x, hipEventRecord(x->ev_sent[slot], x->xs));
  }
  // ---- every rank merges its tile once all layers of it have arrived, and hands it to rank 0's frame
  for (int r = 0; r < nranks; ++r) {
    smk_exchange *x = all[r];
    XHIP(x, hipSetDevice(x->ctx->device));
    for (int p = 0; p < nranks; ++p)
      if (p != r) XHIP(x, hipStreamWaitEvent(x->xs, all[p]->ev_sent[slot], 0));
    if (merge_tile(x, slot)) return 1;
    if (x->cnt(r) > 0) {
      float4 *dst = (float4 *)d_frame + (size_t)r * tp;
      if (x->ctx->device == x0->ctx->device)
        XHIP(x, hipMemcpyAsync(dst, x->tile[slot], (size_t)x->cnt(r) * sizeof(float4), hipMemcpyDeviceToDevice, x->xs));
      else
        XHIP(x, hipMemcpyPeerAsync(dst, x0->ctx->device, x->tile[slot], x->ctx->device, (size_t)x->cnt(r) * sizeof(float4), x->xs));
    }
    XHIP(x, hipEventRecord(x->ev_done[slot], x->xs));
    x->used[slot] = true;
  }
  // rank 0's exchange stream is the one a caller waits on (smk_exchange_wait)
  XHIP(x0, hipSetDevice(x0->ctx->device));
  for (int r = 1; r < nranks; ++r) XHIP(x0, hipStreamWaitEvent(x0->xs, all[r]->ev_done[slot], 0));
  return 0;
}

extern "C" int smk_exchange_wait(smk_exchange *x, void *stream) {
  if (!x) return 1;
  XHIP(x, hipSetDevice(x->ctx->device));
  hipEvent_t ev;
  XHIP(x, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, x->xs);
  if (e == hipSuccess) e = hipStreamWaitEvent(stream ? (hipStream_t)stream : x->ctx->stream, ev, 0);
  (void)hipEventDestroy(ev);  // (destroyed once the recorded work has completed)
  XHIP(x, e);
  return 0;
}
