/*
 * smk.h -- C ABI of the MI355X-native volume ray-marcher that takes over Simian's renderer slot.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  A
 * gluvvPrimitive subclass (INTEGRATION.md shows it; a buildable mirror lives in
 * simian-spacemonkey_amd/host/) calls these from init()/draw() instead of issuing OpenGL.
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference tree, zzmuxi/simian-spacemonkey).
 *
 * Conventions (VolumeRenderer.cpp:101-126, glUE.cpp:194-207): int returns are 0 = ok,
 * non-zero = error, message via smk_last_error(); nothing throws.  Inputs are copied at the
 * call (the reference renderers copy into texture memory at init(), NV20VolRen3D.cpp:1315-1330);
 * the caller keeps ownership of every pointer it passes.  One context per GPU, not thread-safe
 * (the reference is single-threaded GLUT).  There is NO CPU fallback: without a HIP device
 * smk_create() fails.
 */
#ifndef SMK_H
#define SMK_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smk_ctx smk_ctx;

/* voxel storage handed over by the caller */
typedef enum { SMK_U8 = 0, SMK_F32 = 1 } smk_dtype;

/* gluvvDataMode, same order and meaning (gluvv.h:221-235) */
typedef enum {
  SMK_GDM_V1, SMK_GDM_V1G, SMK_GDM_V1GH, SMK_GDM_V2, SMK_GDM_V2G, SMK_GDM_V2GH, SMK_GDM_V3,
  SMK_GDM_V3G, SMK_GDM_V4, SMK_GDM_VGH, SMK_GDM_VGH_VG, SMK_GDM_VGH_V, SMK_GDM_UNKNOWN
} smk_datamode;

/* shading: gluvvShade (gluvv.h:199-207) collapsed to what the two VGH renderers implement */
typedef enum {
  SMK_SHADE_NONE = 0,      /* gluvvShadeAmb / Faux: colour * alpha only                        */
  SMK_SHADE_R8K_DIFF = 1,  /* R8kVolRen3D cube-map Phong, diffuse only (gluvvShadeDiff)         */
  SMK_SHADE_R8K_DSPEC = 2, /* R8kVolRen3D diffuse + specular^30 (gluvvShadeDSpec) -- canonical  */
  SMK_SHADE_NV20_DIFF = 3, /* NV20VolRen3D register-combiner Phong, diffuse                     */
  SMK_SHADE_NV20_DSPEC = 4 /* NV20VolRen3D diffuse + specular^16                                */
} smk_shade;

/* framebuffer blend of the slice polygons (SURVEY 2.1 "Framebuffer blend") */
typedef enum {
  SMK_BLEND_FRONT_TO_BACK = 0, /* GL_ONE_MINUS_DST_ALPHA, GL_ONE (R8kVolRen3D.cpp:1441-1449); exact early termination */
  SMK_BLEND_BACK_TO_FRONT = 1, /* GL_ONE, GL_ONE_MINUS_SRC_ALPHA, far plane first (VolumeRenderer.cpp:590, NV20VolRen3D.cpp:930) */
  SMK_BLEND_MAX = 2            /* glBlendEquationEXT(GL_MAX): gluvvShadeMIP (NV20VolRen3D.cpp:158-163) */
} smk_blend;

/* Exactly the fields of `class Volume` a renderer reads (MetaVolume.h:18-61): one brick as
 * produced by MetaVolume::brick (MetaVolume.cpp:1369-1452). */
typedef struct {
  int xiSize, yiSize, ziSize;   /* voxels                                                  */
  float xfSize, yfSize, zfSize; /* extent in volume space                                  */
  int xiPos, yiPos, ziPos;      /* voxel origin inside the whole volume                    */
  float xfPos, yfPos, zfPos;    /* origin in volume space                                  */
  const void *data;             /* currentData: [z][y][x][nelts], u8 or f32                */
  const unsigned char *grad;    /* currentGrad: [z][y][x][3] scale-biased normals, or NULL */
} smk_volume_desc;

/* replaces `new VolumeRenderer(gluvv.mv,0)` / the renderer constructors (VolumeRenderable.cpp:66,
 * gluvv.cpp:141-199).  device_ordinal >= 0.  *err (may be NULL) gets 0 or an error code. */
smk_ctx *smk_create(int device_ordinal, int *err);
void smk_destroy(smk_ctx *ctx);
const char *smk_last_error(smk_ctx *ctx); /* ctx may be NULL: last smk_create failure */

/* replaces VolumeRenderer::createVolume x2 (VolumeRenderer.cpp:101-212) and
 * NV20VolRen3D/R8kVolRen3D::createBricks (NV20VolRen3D.cpp:1255-1369, R8kVolRen3D.cpp:1926-2055):
 * copies the bricks of one MetaVolume into HBM.  Bricks must tile the volume (MetaVolume::brick
 * output, or a single whole volume).  They are re-assembled into one dense volume with global
 * voxel addressing, which removes the reference's seam artefacts. */
int smk_upload_volume(smk_ctx *ctx, const smk_volume_desc *bricks, int n_bricks, int nelts,
                      smk_dtype dtype, smk_datamode dmode);
/* same, but data/grad are DEVICE pointers on this context's GPU (volumes produced on the GPU,
 * e.g. by smk_synth_volume/smk_make_vgh_device; no PCIe copy) */
int smk_upload_volume_device(smk_ctx *ctx, const smk_volume_desc *bricks, int n_bricks, int nelts,
                             smk_dtype dtype, smk_datamode dmode);

/* Sort-last sharding (no reference equivalent: the reference draws bricks serially on one GPU,
 * NV20VolRen3D.cpp:190-231).  Must be called BEFORE smk_upload_volume: the context then keeps
 * only its convex sub-box (+1 voxel halo) of the volume.  nranks in {1,2,4,8}: split x, then y,
 * then z at the midpoint -- the MetaVolume::brick grid for a 2x2x2 bricking. */
int smk_set_shard(smk_ctx *ctx, int rank, int nranks);
/* front-to-back order of the ranks for the current camera (BSP rule per split axis) */
int smk_shard_order(smk_ctx *ctx, int *order_out /* nranks */);

/* replaces TLUT::loadTransferTableRGBA (TLUT.cpp:48-81): straight-colour float RGBA[size] as
 * held in TLUT::_rgba (already opacity-corrected by TLUT::scaleAlpha, which the caller keeps
 * doing exactly as VolumeRenderable::draw does, VolumeRenderable.cpp:50); premultiplied here. */
int smk_set_tlut1d(smk_ctx *ctx, const float *rgba, int size);
/* replaces NV20VolRen3D::loadDepTex x2 (NV20VolRen3D.cpp:1579-1622): deptex[sg][sv][RGBA8]
 * (gluvv.volren.deptex) and the optional third-axis table deptex2 (gluvv.volren.deptex2). */
int smk_set_tf2d(smk_ctx *ctx, const unsigned char *deptex, const unsigned char *deptex2_or_null,
                 int sv, int sg);
/* replaces TFWidgetRen::loadPtex for the dense table ptex[sh][sg][sv][RGBA8]
 * (TFWidgetRen.cpp:779-845) */
int smk_set_tf3d(smk_ctx *ctx, const unsigned char *ptex, int sv, int sg, int sh);

/* replaces the glGetDoublev(GL_MODELVIEW_MATRIX) + glFrustum state a renderer reads
 * (VolumeRenderable.cpp:40-49, gluvv.cpp:531-552): column-major modelview, frustum
 * {left,right,bottom,top} at clip[0] (near), window size (gluvv.win). */
int smk_set_camera(smk_ctx *ctx, const double modelview[16], const float frustum[4],
                   const float clip[2], int width, int height);
/* replaces R8kVolRen3D::loadCubeTex (R8kVolRen3D.cpp:2620-2679) / NV20VolRen3D::setupRegComb's
 * host half (NV20VolRen3D.cpp:637-668): gluvv.light.pos, gluvv.env.eye/at, gluvv.rinfo.xform,
 * gluvv.light.intens, gluvv.light.amb (amb is only consumed by the shadow passes; stored) */
int smk_set_shading(smk_ctx *ctx, smk_shade mode, const float light_pos[3], const float eye[3],
                    const float at[3], const float xform[16], float intens, float amb);
/* gluvv.volren.sampleRate / gamma / scaleAlphas (NV20VolRen3D.cpp:87-122).  steps > 0 fixes the
 * plane count instead (dis = view-depth extent / steps).  With scale_alphas the 2-D TF alpha is
 * corrected as copyScale does (NV20VolRen3D.cpp:1645-1660) with rate/gamma. */
int smk_set_sampling(smk_ctx *ctx, float sample_rate, int steps, float gamma, int scale_alphas);
/* replaces the orthogonal mode of the clip-plane widget (gluvv.clip.{on,ortho,oaxis,vpos},
 * gluvv.h:163-175; NV20VolRen3D::setupClips, NV20VolRen3D.cpp:251-327): the volume is drawn only
 * on one side of an axis-aligned plane through vpos (volume space, the units of fPos/fSize).
 * oaxis = VolRenMajorAxis: 1 X+ (x <= vpos.x stays), 2 X-, 3 Y+, 4 Y-, 5 Z+, 6 Z-.  on = 0: off. */
int smk_set_clip(smk_ctx *ctx, int on, int oaxis, const float vpos[3]);
/* replaces the extents of VolumeRenderer::renderVolume(sampleRate, mv, xext, yext, zext) (VolumeRenderer.h:103-108,
 * VolumeRenderer.cpp:333-384, render3DVolumeEXTSV :428-505): only the axis-aligned sub-box lo..hi of the volume is drawn
 * (volume space, the units of fPos / fSize; clamped to the volume as :452-457 do; the reference's `x[1] -= origf[1]` slip,
 * :459, is not reproduced).  Planes stay the whole volume's.  on = 0: off. */
int smk_set_region(smk_ctx *ctx, int on, const float lo[3], const float hi[3]);
/* replaces the clip widget's free mode: glClipPlane(GL_CLIP_PLANE5, {0,0,-1,0}) specified under the
 * modelview wmv * T(clip.pos) * clip.xform (NV20VolRen3D.cpp:346-357; R8kVolRen3D.cpp:780-794).
 * plane_eye = the eye-space plane OpenGL stores for that call ({0,0,-1,0} times the inverse of that
 * matrix); a sample stays when plane_eye . (x_eye, 1) >= 0.  Both ray-marchers take it (the slice-ring kernel folds it
 * into each ray's plane interval: nothing per sample). */
int smk_set_clip_plane(smk_ctx *ctx, int on, const double plane_eye[4]);
/* replaces R8kVolRen3D_cpy::createNoiseTex + gluvv.pert (R8kVolRen3D_cpy.cpp:2392-2480,
 * 1590-1595): n^3 RGBA8 noise (GL_REPEAT), weights/scales of the two live octaves. noise NULL
 * or all weights 0 turns perturbation off. */
int smk_set_perturb(smk_ctx *ctx, const unsigned char *noise_rgba, int n, const float w[4],
                    const float s[4]);
/* replaces R8kVolRen3D's shadow mode (gluvv.light.shadow; gluvv.cpp:287-300 buffer size and qualities):
 * half-angle slicing.  The slice axis is the half-way vector of view and light direction
 * (R8kVolRen3D.cpp:296-326), every slice is drawn into the frame with its colour scaled by 1 - light
 * buffer (:1651-1740, shader :2928-2934) and then composited into the light buffer (:1760-1860, shader
 * :2991-3180) under the light's projection (LTWidgetRen::genXForm, LTWidgetRen.cpp:231-291).  Uses the
 * light position, eye, at and xform of smk_set_shading.  buffer_px = gluvv.light.buffsz[0], quality =
 * gluvv.light.gShadowQual or iShadowQual: the light buffer has ceil(quality * buffer_px)^2 texels.
 * Applies to 2-D / 3-D classification with no or R8k shading on an unsharded context, with or without the clip-plane
 * widget's planes (smk_set_clip, smk_set_clip_plane: both passes leave out what lies beyond them, as the reference's
 * clipped slice polygons do); other configurations (1-D table, NV20 combiners, perturbation, a sub-box, depth output,
 * shards) make smk_render fail with the reason.  The blend order follows the light (under when
 * the slices run away from the eye, over otherwise), smk_set_blend is not consulted.
 * How it is rendered: a light-buffer texel depends on itself alone from slice to slice, so the light pass is
 * ONE march per texel that keeps every slice's buffer (nslices + 1 buffers in device memory), and the eye pass is an
 * ordinary frame of the ray-marchers over the half-angle slices that looks each sample's slice up -- two launches instead
 * of one per slice.  Option "shadow_march" 0 (or a history that does not fit a quarter of the free device memory) renders
 * a launch per slice as the reference draws them: the same samples, bit-identical light buffers.  The light buffers belong
 * to the context: its frames with shadows must be enqueued on ONE stream (they order themselves there). */
int smk_set_shadow(smk_ctx *ctx, int on, int buffer_px, float quality);
/* replaces the glBlendFunc / glBlendEquationEXT state of the slice loop (VolumeRenderer.cpp:589-590,
 * NV20VolRen3D.cpp:158-163, 930; R8kVolRen3D.cpp:1436-1449).  Default: front to back.  The two
 * "over" orders are the same operator evaluated from opposite ends (equal up to fp32 rounding):
 * the gather kernel (option "kernel" 1) walks the planes of a back-to-front frame from the far one, as
 * the reference does; the slice-ring kernel composites such a frame front to back. */
int smk_set_blend(smk_ctx *ctx, smk_blend mode);

/* replaces gluvvPrimitive::draw() -> renderVolume (VolumeRenderer.cpp:280-328,
 * NV20VolRen3D.cpp:87-185): one frame.  rgba_out: [height][width][4] float, premultiplied,
 * row 0 = bottom (GL window order).  depth_out (may be NULL): view-space depth of the first
 * contributing sample, +inf where none. */
int smk_render(smk_ctx *ctx, float *rgba_out, float *depth_out);
/* same with DEVICE output pointers; asynchronous on `stream` (a hipStream_t, NULL = default).  (Option "kernel" 0: the
 * first nine frames of a configuration the context has not measured yet are trials of the two ray-marchers -- identical
 * frames -- and six of them wait for the frame before them on `stream`: a one-time stall per configuration.) */
int smk_render_device(smk_ctx *ctx, void *d_rgba, void *d_depth, void *stream);

/* replaces VolumeRenderer::renderSlice(quad, alpha) (VolumeRenderer.h:114, VolumeRenderer.cpp:748-807): ONE quad (model
 * space, the units of fPos / fSize; drawn as glBegin(GL_QUADS) with the vertices in the order 1, 0, 2, 3) textured with the
 * scalar volume -- GL_INTENSITY8, GL_LINEAR, no colour table (:768), texture coordinates = vertex / fSize -- modulated by
 * glColor4f(1, 1, 1, alpha) and blended GL_ONE, GL_ONE_MINUS_SRC_ALPHA into the frame:
 *   src = (I, I, I, I * alpha), frame = src + (1 - src.a) * frame,   I = the first data channel in [0, 1].
 * Uses the camera of smk_set_camera.  rgba_inout: [height][width][4] float, read and written. */
int smk_render_slice(smk_ctx *ctx, const float quad[4][3], float alpha, float *rgba_inout);
int smk_render_slice_device(smk_ctx *ctx, const float quad[4][3], float alpha, void *d_rgba_inout, void *stream);

/* Frames in flight (no reference equivalent: the reference renders synchronously).  smk_render_device
 * only enqueues; the slice-ring kernel reports a protocol time-out or a window outside its host bound
 * through a per-frame status word.  smk_last_frame_id: id of the frame the last smk_render_device call
 * enqueued (1, 2, ...).  smk_frame_failed: call after synchronising with that frame's stream; 1 = the
 * frame was flagged and must be rendered again (option "kernel" = 1 renders it on the gather kernel),
 * 0 = valid, -1 = unknown (never enqueued, or older than the last 8 frames).  Asking consumes the answer.
 * The host may enqueue frame i + 1 before it asks about frame i (the pipelined protocol of the sort-last
 * merge): a render call only looks at the status slot it takes over -- a flagged frame nobody asked
 * about while it could be asked about makes the render call 8 frames later fail.  The synchronous
 * smk_render re-renders a flagged frame itself.  Status words carry their frame's id, so a word that
 * arrives after its slot was handed on is not blamed on the younger frame.  All of it is counted
 * (smk_get_stat "slab_failures" / "slab_retries") so a test or a benchmark can require zero. */
long long smk_last_frame_id(smk_ctx *ctx);
int smk_frame_failed(smk_ctx *ctx, long long frame_id);

/* sort-last merge: out = layer[order[0]] over layer[order[1]] over ...; layers are
 * premultiplied RGBA tiles of npix pixels, DEVICE pointers, layer l at layers + l*npix*4 floats. */
int smk_composite_over_device(smk_ctx *ctx, const void *d_layers, int nlayers, const int *order,
                              int npix, void *d_out, void *stream);

/* ---- the sort-last merge in C (no reference equivalent): one object per rank = per
 * context / GPU.  Direct send of the 1/P image tiles (grouped ncclSend/ncclRecv over xGMI), ordered
 * over of the P layers in smk_shard_order's order, finished tiles gathered on rank 0.
 *   RCCL transport (id != NULL): one process per GPU.  Rank 0 makes the 128-byte communicator id
 *     (smk_exchange_unique_id = ncclGetUniqueId) and hands it to the other ranks by whatever channel
 *     the host has; smk_exchange_create is then collective (ncclCommInitRank).  librccl is opened at
 *     run time.
 *   in-process transport (id == NULL): the ranks are contexts of ONE process (a C++ host that owns
 *     several GPUs): create all, smk_exchange_connect_local, then smk_exchange_frame_local per frame.
 * A rank renders frame i into smk_exchange_partial(x, i & 1) -- [npix][4] floats, device -- after
 * smk_exchange_acquire(x, i & 1, render_stream), and marks it with smk_exchange_rendered(x, i & 1,
 * render_stream); smk_exchange_frame enqueues that frame's merge on the exchange's own stream behind
 * that mark and returns at once, so frame i's merge overlaps frame i+1's ray-marching (which may
 * already be enqueued: a host checks smk_frame_failed for frame i in between).  d_frame ([npix][4], rank 0 only) receives the merged frame;
 * smk_exchange_wait makes a stream wait for everything enqueued so far.  A NULL stream means the
 * context's own stream, as in smk_render_device.  The context must have been sharded with the same
 * rank / nranks (smk_set_shard). */
typedef struct smk_exchange smk_exchange;
#define SMK_EXCHANGE_ID_BYTES 128
int smk_exchange_unique_id(unsigned char id[SMK_EXCHANGE_ID_BYTES]);
smk_exchange *smk_exchange_create(smk_ctx *ctx, int rank, int nranks, const unsigned char *id, int npix, int *err);
int smk_exchange_connect_local(smk_exchange *const *all, int nranks);
void smk_exchange_destroy(smk_exchange *x);
const char *smk_exchange_last_error(smk_exchange *x); /* x may be NULL: last smk_exchange_create failure */
void *smk_exchange_partial(smk_exchange *x, int slot);
int smk_exchange_acquire(smk_exchange *x, int slot, void *render_stream);
int smk_exchange_rendered(smk_exchange *x, int slot, void *render_stream);
/* The shards' visibility order of the frame in `slot` is taken from the context's camera at the first
 * smk_exchange_rendered after smk_exchange_acquire -- i.e. under the pose the frame was rendered with, not the one current
 * when the merge is enqueued; a host that knows better (a frame re-rendered later, a replayed sequence) sets it itself:
 * order[nranks], front to back (smk_shard_order's convention). */
int smk_exchange_set_order(smk_exchange *x, int slot, const int *order);
int smk_exchange_frame(smk_exchange *x, int slot, void *d_frame);
int smk_exchange_frame_local(smk_exchange *const *all, int nranks, int slot, void *d_frame);
int smk_exchange_wait(smk_exchange *x, void *stream);

/* data prep on the GPU (genVGH/main.cpp:56-182, VectorMath.h:874-899,
 * 1133-1148, 1217-1281).  All pointers are DEVICE pointers.
 *   scalar  [z][y][x] u8 or f32      -> vgh_u8 [z][y][x][3] (quantised as makeVGH) and/or
 *                                       vgh_f32 [z][y][x][3] in [0,1] (unquantised variant)
 *   vgh_u8                           -> normals [z][y][x][3] (derivative3DVGH+[blurV3D]+scalebiasN) */
int smk_make_vgh_device(smk_ctx *ctx, const void *d_scalar, smk_dtype dtype, int sx, int sy,
                        int sz, int compat, void *d_vgh_u8_or_null, void *d_vgh_f32_or_null);
int smk_normals_vgh_device(smk_ctx *ctx, const void *d_vgh_u8, int nelts, int sx, int sy, int sz,
                           int blur, void *d_normals);
/* replaces MetaVolume::mergeMV with addG (MetaVolume.cpp:1109-1268; AGradArb VectorMath.h:945-1004,
 * GMag :1010-1030, scalebiasN :1133-1148): nf (1..3) co-registered scalar fields, interleaved
 * [z][y][x][nf] u8 -> d_out [z][y][x][nf+1] with the magnitude of the SUMMED per-field gradient as
 * last element, and optionally the normal bytes [z][y][x][3] of that gradient.  Device pointers;
 * bytes identical to the reference arithmetic. */
int smk_merge_fields_device(smk_ctx *ctx, const void *d_fields_u8, int nf, int sx, int sy, int sz,
                            void *d_out, void *d_normals_or_null);
/* replaces MetaVolume::hist2D (MetaVolume.cpp:1650-1688; caller TFWidgetRen::loadHist,
 * TFWidgetRen1.cpp:660-700): the log-scaled joint histogram of the (value, gradient) bytes,
 * hist[g*256 + v], 65536 bytes in HOST memory; bit-identical to the reference's, including its
 * float bins that stop counting at 2^24.  nelts < 2 is refused as there.  _device: the volume
 * [z][y][x][nelts] is already in device memory. */
int smk_hist2d(smk_ctx *ctx, const smk_volume_desc *bricks, int n_bricks, int nelts, unsigned char *hist);
int smk_hist2d_device(smk_ctx *ctx, const void *d_vol_u8, int nelts, int sx, int sy, int sz, unsigned char *hist);
/* synthetic scalar test volumes generated on the GPU (u8, [z][y][x]):
 *   kind 1 = the reference's own generator: `genvol -spheres 4 -p 10 -pscale .7 -pwrap 3 3 3 -pabs -blur
 *            -bw 1 1 1 .7` with srand(seed) (genvol/main.cpp:153-165, 212-256, 334-430; perlin.c; script
 *            genvol/scripts/make64.bat:1) -- bytes identical to a CPU run of that tool on glibc
 *   kind 0 = smooth noisy concentric shells (analytic, no staircase; round-1 bench input) */
int smk_synth_volume_device(smk_ctx *ctx, int kind, unsigned seed, int sx, int sy, int sz,
                            void *d_scalar_u8);

/* introspection used by tests and bench (no reference equivalent) */
typedef struct {
  float pxs, pxl, pys, pyl;
  float Ac[3], Ax[3], Ay[3], Bc[3], Bx[3], By[3];
  int nplanes;
  float tau0, dtau, zmin, zmax, dis;
} smk_raycoef;
int smk_get_raycoef(smk_ctx *ctx, smk_raycoef *out);
/* sample placement of a frame with shadows: slice k = 1..nslices in the light's order.  A light-buffer texel's sample is
 * fma(w, G, Lc) with G_a = fma(a, Gx_a, fma(b, Gy_a, Gc_a)), (a, b) = fma(texel + .5, las, lal),
 * w = fma(k, ldnum, lnum0) / fma(a, nGx, fma(b, nGy, nGc)).  An eye ray's planes are counted FROM THE EYE, m = k - 1 when the
 * slices run away from the viewer (front_to_back), nslices - k otherwise: with D_a = fma(px, Dx_a, fma(py, Dy_a, Dc_a)),
 * nD = fma(px, nDx, fma(py, nDy, nDc)), numA = plane 0's numerator (fma(1, dnum, num0) or fma(nslices, dnum, num0)) and
 * dB = +-dnum:  tauA = numA / nD, dtau = dB / nD, A_a = fma(tauA, D_a, Ec_a), B_a = dtau * D_a, sample = fma(m, B, A), which
 * exists where fma(m, dtau, tauA) is positive and finite and lies within 2^-10 voxels of the volume's box.  X/Y/Wm map a
 * voxel coordinate to light space, its light-buffer position is fma(x'/w, lscale, lbias) (DESIGN.md "Shadows") */
typedef struct {
  float pxs, pxl, pys, pyl;
  float Ec[3], Dc[3], Dx[3], Dy[3];
  float nDc, nDx, nDy, num0, dnum;
  float las, lal;
  float Lc[3], Gc[3], Gx[3], Gy[3];
  float nGc, nGx, nGy, lnum0, ldnum;
  float Xm[4], Ym[4], Wm[4];
  float lscale, lbias;
  int nslices, LB, front_to_back;
} smk_shadowcoef;
int smk_get_shadowcoef(smk_ctx *ctx, smk_shadowcoef *out);
/* the light buffer as the last frame with shadows left it: [LB][LB][4] floats to HOST memory (synchronises) */
int smk_get_light_buffer(smk_ctx *ctx, float *rgba_out, int *lb_out);
/* Empty-space skipping (option "bricks", no reference counterpart: the reference draws every slice and lets the blend
 * unit discard what the table made transparent, VolumeRenderer.cpp:507-741).  The flags the NEXT frame would use, for
 * checkers: one byte per brick of 8x8x8 cells of this context's stored box, x fastest, 1 = some sample whose cell lies
 * in the brick may be visible under the current table.  nb_out[3] receives the brick counts; flags_out may be NULL to
 * ask for the counts alone; *in_use_out (may be NULL) tells whether frames use them (not for a 1-D colour table, with
 * the option off, or once > 90 % of the bricks turned out flagged).  Needs volume, table and camera; synchronises. */
int smk_get_brick_flags(smk_ctx *ctx, unsigned char *flags_out, int *nb_out, int *in_use_out);
/* options (all optional; defaults in brackets):
 *   "kernel"   [0] 0 auto: both ray-marchers produce bit-identical frames, the first frames of a new
 *              configuration time one and the other and the faster is kept; 1 gather kernel (every
 *              mode); 2 slice-ring kernel (fails where it does not apply, with the reason); 3 column-stream
 *              kernel (smk_cols.hip: the volume re-laid out in columns with their halo, streamed sequentially,
 *              a ray's per-column partial composites merged in order -- the same samples, the blend
 *              re-associated: <= 2e-5 from the other two; fails where it does not apply)
 *   "slab_split" [0] depth segments of the slice-ring kernel: 0 = tiles measured long are rendered by several
 *              workgroups where the longest tile stands above the mean load of a workgroup slot (sharded
 *              contexts), the partial frames merged in order (<= 2e-5 from the unsplit frame); 1 = never
 *              (bit-identical to the gather kernel everywhere); 2..8 = every tile in that many.  Frames with a
 *              depth output are never cut (the merge pass knows colours only)
 *   "tf_raw"   [0] 1: the 2-D table handed to smk_set_tf2d is already opacity-corrected (copyScale off)
 *   "halo"     [1] voxels of halo kept around a shard's region (before smk_upload_volume)
 *   "bricks"   [1] empty-space skipping: 8x8x8-cell bricks in which no sample can be visible under the current
 *              table are neither streamed nor sampled (the skipped samples are exactly transparent: frames are
 *              bit-identical with 0 and 1); 0 = every sample is fetched and classified
 *   "shadow_march" [1] frames with shadows as two marches (light-buffer texels, then eye pixels on the ray-marchers);
 *              0 = a launch per slice
 *   developer knobs: "tile" (slice-ring workgroup shape id), "slab_T" (band wait + 1), "slab_fly"
 *   (slices a loader keeps in flight), "slab_ns" (cap on the ring's slots), "lockstep" (bit 0 gather lockstep; bits 1..6 slice-ring
 *   diagnostics, see tools/kbench.py), "wave_w"/"blk_w" (gather tile shape), "inject_slab_status"
 *   (test hook: the next slice-ring frame reports this status word) */
int smk_set_option(smk_ctx *ctx, const char *key, int value);
/* samples of the current frame set-up that lie inside the volume (region and clip planes counted in): the renderers'
 * membership test for every plane of every ray, nothing fetched.  SURVEY 8(d)'s "in-volume sample count", to be read
 * beside the nominal width x height x planes.  Synchronises. */
int smk_count_samples(smk_ctx *ctx, double *in_volume);
/* last frame: which kernel ran (1 gather, 2 slice-ring, 3 the per-slice shadow passes of option shadow_march 0, 4 column-
 * stream; a frame with shadows reports the kernel of its eye pass, its time covers the light march too), its HIP-event time
 * in ms, algorithmic bytes */
int smk_last_frame_info(smk_ctx *ctx, int *kernel, float *ms, double *alg_bytes);
/* HIP-event timing of the render kernel on its launch stream: reset, render N frames, read the
 * average (ms) over the last min(N,64) frames.  smk_timing_read synchronises the device. */
int smk_timing_reset(smk_ctx *ctx);
int smk_timing_read(smk_ctx *ctx, float *avg_ms, int *nframes);
/* named counters of the last frame (developer statistics, no reference counterpart):
 * "slab_iters", "slab_active_lanes", "slab_inside_lanes" (lanes that interpolate a sample: not skipped as part of an
 * empty layer), "slab_hit_lanes" (collected when option
 * lockstep has bit 16 set), "slab_status" (these synchronise the device); "slab_failures",
 * "slab_retries" (host-side counters, no synchronisation). */
int smk_get_stat(smk_ctx *ctx, const char *name, double *value);
/* workgroup timeline of the last slice-ring frame (developer tool):
 * records of 8 x uint32 {start, end (100 MHz ticks), HW_ID, XCC_ID | tile<<8 | slices<<20, loader 0's
 * issue / vmcnt-wait / ring-blocked time in units of 64 cycles, sum of the consumer waves' iterations};
 * with option lockstep bit 32 set the records come from the diagnostic kernel instances, one per workgroup;
 * without it from the product kernel itself, one per tile (the first four words; a tile cut in depth segments
 * leaves its record empty).
 * *nrecords = records available; copies min(cap_records, *nrecords) when out != NULL. */
int smk_get_trace(smk_ctx *ctx, unsigned *out, int cap_records, int *nrecords);
/* effective 2-D TF after opacity correction, as uploaded (sg*sv*4 bytes) */
int smk_get_tf2d_effective(smk_ctx *ctx, unsigned char *out, float *rate_out);

#ifdef __cplusplus
}
#endif
#endif
