// HipVolumeRenderer.cpp -- see the header.  Every GL call of the reference path is replaced by
// one smk_* call; everything else (who owns what, when tables are re-sent, error style) follows
// the reference files cited inline.
#include "HipVolumeRenderer.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <cstdio>
#include <iostream>

static void build_modelview(double mv[16]);  // LookAt * T(trans) * R(xform) * T(-size/2)
static void build_world_modelview(double mv[16]);
static void mul(double o[16], const double a[16], const double b[16]);
static void translate(double m[16], double x, double y, double z);

HipVolumeRenderer::HipVolumeRenderer(MetaVolume *vm, int, int device) : ctx(nullptr), m_vol(vm), tlut(nullptr), failed(0), m_bb(0), m_bbb(0) {
  int err = 0;
  ctx = smk_create(device, &err);
  if (!ctx) {
    std::cerr << "ERROR: HipVolumeRenderer: " << smk_last_error(nullptr) << std::endl;
    failed = 1;
  }
}

HipVolumeRenderer::~HipVolumeRenderer() {
  if (tlut) delete tlut;  // VolumeRenderer.h:89
  smk_destroy(ctx);
}

int HipVolumeRenderer::upload(Volume *v, int n) {
  if (!ctx) return 1;
  std::vector<smk_volume_desc> d(n);
  for (int i = 0; i < n; ++i) {
    d[i].xiSize = v[i].xiSize; d[i].yiSize = v[i].yiSize; d[i].ziSize = v[i].ziSize;
    d[i].xfSize = v[i].xfSize; d[i].yfSize = v[i].yfSize; d[i].zfSize = v[i].zfSize;
    d[i].xiPos = v[i].xiPos; d[i].yiPos = v[i].yiPos; d[i].ziPos = v[i].ziPos;
    d[i].xfPos = v[i].xfPos; d[i].yfPos = v[i].yfPos; d[i].zfPos = v[i].zfPos;
    d[i].data = v[i].currentData;
    d[i].grad = v[i].currentGrad;
  }
  if (smk_upload_volume(ctx, d.data(), n, m_vol->nelts, SMK_U8, (smk_datamode)gluvv.dmode)) {
    std::cerr << "ERROR: HipVolumeRenderer::createVolume: " << smk_last_error(ctx) << std::endl;
    failed = 1;
    return 1;
  }
  return 0;
}

int HipVolumeRenderer::hist2D(unsigned char *hist) {
  if (!ctx || !m_vol || !m_vol->volumes) return 0;
  std::vector<smk_volume_desc> d((size_t)m_vol->numSubVols);
  for (int i = 0; i < m_vol->numSubVols; ++i) {
    const Volume &v = m_vol->volumes[i];
    memset(&d[(size_t)i], 0, sizeof(smk_volume_desc));
    d[(size_t)i].xiSize = v.xiSize; d[(size_t)i].yiSize = v.yiSize; d[(size_t)i].ziSize = v.ziSize;
    d[(size_t)i].data = v.currentData;
  }
  if (smk_hist2d(ctx, d.data(), m_vol->numSubVols, m_vol->nelts, hist)) {
    std::cerr << "MetaVolume::hist2D, " << smk_last_error(ctx) << std::endl;
    return 0;  // the reference's "not implemented" return
  }
  return 1;
}

int HipVolumeRenderer::createVolume(int type, Volume *v) {
  if (type != VolRen3DExt) {  // VolumeRenderer.cpp:103-112: the other modes "not implemented"
    std::cerr << "texture mapping method is not implemented" << std::endl;
    return 1;
  }
  return upload(v, 1);
}

int HipVolumeRenderer::createVolume(int type, Volume *v, int nVols) {
  if (type != VolRen3DExt) {
    std::cerr << "texture mapping method is not implemented" << std::endl;
    return 1;
  }
  return upload(v, nVols);
}

int HipVolumeRenderer::createTLUT() {
  tlut = new HipTLUT;
  return 1;
}

// TLUT::scaleAlpha (TLUT.cpp:138-154) up to, not including, its loadTransferTableRGBA() call
int HipTLUT::scaleAlphaNoUpload(float sampleRate) {
  if (lastSampleRate == sampleRate) return 0;
  float alphaScale = lastSampleRate / sampleRate;
  lastSampleRate = sampleRate;
  for (int i = 0; i < _size; ++i) _rgba[i * _numElts + 3] = 1 - pow((1 - _rgba[i * _numElts + 3]), alphaScale);
  return 1;
}

void HipVolumeRenderer::loadTransferTableRGBA() {
  if (!ctx || !tlut) return;
  if (smk_set_tlut1d(ctx, tlut->GetRGBA(0), tlut->GetSize())) {
    std::cerr << "ERROR: HipVolumeRenderer::loadTransferTableRGBA: " << smk_last_error(ctx) << std::endl;
    failed = 1;
  }
}

void HipVolumeRenderer::renderVolume(float sampleRate, double mv[16]) {
  if (!ok()) return;
  const int W = (int)gluvv.win.width, H = (int)gluvv.win.height;
  fb.assign((size_t)W * H * 4, 0.0f);
  int rc = smk_set_camera(ctx, mv, gluvv.env.frustum, gluvv.env.clip, W, H);
  rc |= smk_set_sampling(ctx, sampleRate, 0, gluvv.volren.gamma, gluvv.volren.scaleAlphas);
  if (!rc) rc = smk_render(ctx, fb.data(), nullptr);
  if (rc) {
    std::cerr << "ERROR: HipVolumeRenderer::renderVolume: " << smk_last_error(ctx) << std::endl;
    failed = 1;
  }
}

void HipVolumeRenderer::renderVolume(float sampleRate, double mv[16], float xext[2], float yext[2], float zext[2]) {
  if (!ok()) return;
  // (render3DVolumeEXTSV clamps the extents to the volume and returns when nothing is left, VolumeRenderer.cpp:452-463)
  const float lo[3] = {xext[0], yext[0], zext[0]}, hi[3] = {xext[1], yext[1], zext[1]};
  if (smk_set_region(ctx, 1, lo, hi)) {
    std::cerr << "ERROR: HipVolumeRenderer::renderVolume: " << smk_last_error(ctx) << std::endl;
    failed = 1;
    return;
  }
  renderVolume(sampleRate, mv);
  smk_set_region(ctx, 0, nullptr, nullptr);
}

void HipVolumeRenderer::renderSlice(float quad[4][3], float alpha) {
  if (!ok()) return;
  const size_t npix = (size_t)gluvv.win.width * (size_t)gluvv.win.height;
  if (fb.size() != npix * 4) fb.assign(npix * 4, 0.0f);  // (no frame yet: the slice goes onto a cleared one)
  if (smk_render_slice(ctx, quad, alpha, fb.data())) {
    std::cerr << "ERROR: HipVolumeRenderer::renderSlice: " << smk_last_error(ctx) << std::endl;
    failed = 1;
  }
}

// -------------------------------------------------------------------------------------------

void HipVolumeRenderable::init() {
  if (!gluvv.mv) {  // VolumeRenderable.cpp:62-65
    std::cerr << "ERROR: HipVolumeRenderable::init(), no MetaVolume defined" << std::endl;
    return;
  }
  volren = new HipVolumeRenderer(gluvv.mv, 0, device);
  int bad;
  if (gluvv.mv->numSubVols == 1) bad = volren->createVolume(VolRen3DExt, gluvv.mv->volumes);
  else bad = volren->createVolume(VolRen3DExt, gluvv.mv->volumes, gluvv.mv->numSubVols);
  if (bad || !volren->ok()) return;  // go stays 0: draw() is a no-op (NV20VolRen3D.cpp:44-65)
  if (gluvv.dmode == GDM_V1 && !gluvv.volren.deptex) {
    // scalar path: 1-D TLUT owned by the renderer, published in gluvv.volren.tlut
    // (VolumeRenderable.cpp:72-78; the preset colormaps are the caller's to choose)
    volren->createTLUT();
    gluvv.volren.tlut = volren->getColorMap();
    volren->loadTransferTableRGBA();
  } else {
    gluvv.volren.loadTLUT = 1;  // first draw() sends deptex/deptex2 (NV20VolRen3D.cpp:91-122)
  }
  createNoiseTex(32, 32, 32);  // R8kVolRen3D_cpy.cpp:61 (constructor) -> :300-304
  go = 1;
}

// R8kVolRen3D_cpy::createNoiseTex (:2392-2436): srand(1), four draws per texel, no blur pass (its
// loop runs zero times); the GL texture it fills is GL_REPEAT + GL_LINEAR -- smk_set_perturb's fetch
void HipVolumeRenderable::createNoiseTex(int sx, int sy, int sz) {
  noise.resize((size_t)sx * sy * sz * 4);
  srand(1);
  for (int i = 0; i < sz; ++i)
    for (int j = 0; j < sy; ++j)
      for (int k = 0; k < sx; ++k)
        for (int e = 0; e < 4; ++e)
          noise[(size_t)i * sx * sy * 4 + (size_t)j * sx * 4 + k * 4 + e] = (unsigned char)(((rand() / (float)RAND_MAX * .5) + .5 + 1.0 / 512) * 255);
}

void HipVolumeRenderable::modelview(double mv[16]) { build_modelview(mv); }

void HipVolumeRenderable::draw() {
  if (!go || !volren) return;
  if (gluvv.reblend) return;  // R8kVolRen3D.cpp:177
  if (gluvv.picking) return;  // :179
  double mv[16];
  build_modelview(mv);
  smk_ctx *c = volren->context();
  if (gluvv.volren.tlut && gluvv.dmode == GDM_V1 && !gluvv.volren.deptex) {
    // VolumeRenderable::draw (:50-54): opacity-correct the TLUT for the sample rate, re-send
    // (TLUT::scaleAlpha itself re-uploads through GL and returns nothing, TLUT.h:45: the same
    //  arithmetic without the upload, then one smk_set_tlut1d when anything changed)
    if (volren->colorMap()->scaleAlphaNoUpload(gluvv.volren.sampleRate) | gluvv.volren.loadTLUT) {
      gluvv.volren.loadTLUT = 0;
      volren->loadTransferTableRGBA();
    }
  } else if (gluvv.volren.loadTLUT) {
    // TFWidgetRen raised loadTLUT after rasterising into deptex (TFWidgetRen1.cpp:232-242)
    // a table with several sheets along the third axis (gluvv.tf.ptexsz[2] > 1: what LevWidget::rasterize
    // fills as tex[h][g][v][4], LevWidget.cpp:691-693, and the old widget called ptex,
    // TFWidgetRen.cpp:779-845) is the dense 3-D transfer function; one sheet is the 2-D one x deptex2
    int bad;
    if (gluvv.tf.ptexsz[2] > 1) bad = smk_set_tf3d(c, gluvv.volren.deptex, gluvv.tf.ptexsz[0], gluvv.tf.ptexsz[1], gluvv.tf.ptexsz[2]);
    else bad = smk_set_tf2d(c, gluvv.volren.deptex, gluvv.volren.deptex2, gluvv.tf.ptexsz[0], gluvv.tf.ptexsz[1]);
    if (bad) std::cerr << "ERROR: HipVolumeRenderable::draw: " << smk_last_error(c) << std::endl;
    gluvv.volren.loadTLUT = 0;
  }
  // noise-perturbed fetch (R8kVolRen3D_cpy.cpp:300-304 constants, :1590-1595 coordinates): the _cpy
  // renderer, chosen at compile time in gluvv.cpp:164-198, always perturbs; as one renderer among the
  // others this one follows gluvv.pert.on
  if (gluvv.pert.on) smk_set_perturb(c, noise.data(), 32, gluvv.pert.weights, gluvv.pert.scales);
  else smk_set_perturb(c, nullptr, 0, nullptr, nullptr);
  // clip-plane widget, orthogonal mode (NV20VolRen3D::setupClips, NV20VolRen3D.cpp:251-327)
  smk_set_clip(c, gluvv.clip.on && gluvv.clip.ortho, (int)gluvv.clip.oaxis, gluvv.clip.vpos);
  // free mode: glClipPlane(GL_CLIP_PLANE5, {0,0,-1,0}) under wmv * T(clip.pos) * clip.xform (:346-357).
  // That matrix K is rigid, so the eye-space plane GL stores, zup * K^-1, is (-z_K, z_K . t_K): normal
  // against the widget's local z axis, through its origin.
  if (gluvv.clip.on && !gluvv.clip.ortho) {
    double wmv[16], t[16], x[16], k[16];
    build_world_modelview(wmv);
    translate(t, gluvv.clip.pos[0], gluvv.clip.pos[1], gluvv.clip.pos[2]);
    for (int i = 0; i < 16; ++i) x[i] = gluvv.clip.xform[i];
    mul(k, wmv, t);
    mul(k, k, x);
    const double plane[4] = {-k[8], -k[9], -k[10], k[8] * k[12] + k[9] * k[13] + k[10] * k[14]};
    smk_set_clip_plane(c, 1, plane);
  } else {
    smk_set_clip_plane(c, 0, nullptr);
  }
  // Phong of the platform's renderer: register combiners on the GeForce3 targets (NV20VolRen3D.cpp:634-806),
  // the cube-map shader everywhere else (R8kVolRen3D.cpp:2620-2679, 2886-2902; renderer choice gluvv.cpp:141-199)
  const bool nv20 = gluvv.plat == GPNV20 || gluvv.plat == GPNV202D;
  smk_shade sm = SMK_SHADE_NONE;
  if (gluvv.shade == gluvvShadeDiff) sm = nv20 ? SMK_SHADE_NV20_DIFF : SMK_SHADE_R8K_DIFF;
  if (gluvv.shade == gluvvShadeDSpec) sm = nv20 ? SMK_SHADE_NV20_DSPEC : SMK_SHADE_R8K_DSPEC;
  // gluvvShadeMIP: glBlendEquationEXT(GL_MAX) around renderBricks (NV20VolRen3D.cpp:158-163)
  smk_set_blend(c, gluvv.shade == gluvvShadeMIP ? SMK_BLEND_MAX : SMK_BLEND_FRONT_TO_BACK);
  smk_set_shading(c, sm, gluvv.light.pos, gluvv.env.eye, gluvv.env.at, gluvv.rinfo.xform, gluvv.light.intens, gluvv.light.amb);
  // shadow mode (R8kVolRen3D.cpp:296-326, 1651-1868): light buffer of buffsz texels scaled by the good or the
  // interactive quality, whichever sampling rate is in force (R8kVolRen3D::setupPBuff, :1114-1123)
  smk_set_shadow(c, gluvv.light.shadow, gluvv.light.buffsz[0],
                 gluvv.volren.sampleRate == gluvv.volren.goodSamp ? gluvv.light.gShadowQual : gluvv.light.iShadowQual);
  volren->renderVolume(gluvv.volren.sampleRate, mv);
  if (!volren->ok()) go = 0;
}

// gluLookAt(eye,at,up) * T(trans) * R(xform) * T(-size/2)   (gluvv.cpp:531-540, VolumeRenderable.cpp:40-46)
static void mul(double o[16], const double a[16], const double b[16]) {
  double t[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[k * 4 + r] * b[c * 4 + k];
      t[c * 4 + r] = s;
    }
  memcpy(o, t, sizeof t);
}
static void translate(double m[16], double x, double y, double z) {
  memset(m, 0, 16 * sizeof(double));
  m[0] = m[5] = m[10] = m[15] = 1;
  m[12] = x; m[13] = y; m[14] = z;
}
static void build_world_modelview(double mv[16]) {  // gluLookAt(eye, at, up): the "wmv" the clip widget lives in
  const float *e = gluvv.env.eye, *a = gluvv.env.at, *u = gluvv.env.up;
  double f[3] = {a[0] - e[0], a[1] - e[1], a[2] - e[2]};
  double fl = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  for (double &x : f) x /= fl;
  double s[3] = {f[1] * u[2] - f[2] * u[1], f[2] * u[0] - f[0] * u[2], f[0] * u[1] - f[1] * u[0]};
  double sl = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
  for (double &x : s) x /= sl;
  double uu[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
  double la[16] = {s[0], uu[0], -f[0], 0, s[1], uu[1], -f[1], 0, s[2], uu[2], -f[2], 0, 0, 0, 0, 1};
  double t[16];
  translate(t, -e[0], -e[1], -e[2]);
  mul(mv, la, t);
}
static void build_modelview(double mv[16]) {
  const float *e = gluvv.env.eye, *a = gluvv.env.at, *u = gluvv.env.up;
  double f[3] = {a[0] - e[0], a[1] - e[1], a[2] - e[2]};
  double fl = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  for (double &x : f) x /= fl;
  double s[3] = {f[1] * u[2] - f[2] * u[1], f[2] * u[0] - f[0] * u[2], f[0] * u[1] - f[1] * u[0]};
  double sl = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
  for (double &x : s) x /= sl;
  double uu[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
  double la[16] = {s[0], uu[0], -f[0], 0, s[1], uu[1], -f[1], 0, s[2], uu[2], -f[2], 0, 0, 0, 0, 1};
  double t[16], r[16];
  translate(t, -e[0], -e[1], -e[2]);
  mul(mv, la, t);
  translate(t, gluvv.rinfo.trans[0], gluvv.rinfo.trans[1], gluvv.rinfo.trans[2]);
  mul(mv, mv, t);
  for (int i = 0; i < 16; ++i) r[i] = gluvv.rinfo.xform[i];
  mul(mv, mv, r);
  translate(t, -gluvv.mv->xfSize / 2, -gluvv.mv->yfSize / 2, -gluvv.mv->zfSize / 2);
  mul(mv, mv, t);
}
