// HipVolumeRenderer.h -- host-side mirror of the reference renderer interface over the C ABI.
//
//   HipVolumeRenderer   same public surface as VolumeRenderer (VolumeRenderer.h:86-118):
//                       createVolume x2, createTLUT, getColorMap, renderVolume(sampleRate, mv[, x/y/zext]),
//                       renderSlice(quad, alpha), useBBox / useBBoxBrackets
//   HipVolumeRenderable a gluvvPrimitive whose init()/draw() do what VolumeRenderable (scalar,
//                       1-D TLUT; VolumeRenderable.cpp:36-82) and NV20VolRen3D/R8kVolRen3D (VGH,
//                       deptex/deptex2, Phong; NV20VolRen3D.cpp:44-185) do, minus OpenGL: the
//                       frame lands in a float RGBA buffer (premultiplied) the app can blit.
//
// Error convention of the reference: int returns 0 = ok / 1 = error, messages on cerr, a failed
// renderer turns draw() into a no-op (`go = 0`, NV20VolRen3D.cpp:44-65).  Nothing here renders
// on the CPU: without libsmk_hip.so + a HIP device init() fails and says so.
#pragma once
#include <vector>

#include "../../include/smk.h"
#ifdef SMK_USE_REFERENCE_HEADERS
#include "gluvv.h"  // the reference's own: pulls in gluvvPrimitive.h, MetaVolume.h, TLUT.h
#else
#include "gluvv_compat.h"
#endif

enum { VolRenUnkown, VolRen2DTexture, VolRen3DTexture, VolRen3DExt };  // VolumeRenderer.h:64-69

// The renderer's colour map.  TLUT::scaleAlpha (TLUT.cpp:138-154) ends in loadTransferTableRGBA(), a
// GL colour-table upload, and returns nothing; this subclass reaches TLUT's protected members to do
// the same opacity correction on the same fields without the GL call and says whether the table
// changed (so the adapter knows when to send it through smk_set_tlut1d).  It adds no data member:
// what gluvv.volren.tlut publishes is a plain TLUT to everyone else.
class HipTLUT : public TLUT {
 public:
  HipTLUT() : TLUT() {}
  int scaleAlphaNoUpload(float sampleRate);  // 1 = alphas were rescaled
};

class HipVolumeRenderer {
 public:
  HipVolumeRenderer(MetaVolume *vm, int subVolNum, int device = 0);
  ~HipVolumeRenderer();
  int createVolume(int type, Volume *v);             // VolumeRenderer.cpp:101
  int createVolume(int type, Volume *v, int nVols);  // VolumeRenderer.cpp:128
  int createTLUT();                                  // always returns 1, as the reference (:214-219)
  TLUT *getColorMap() { return tlut; }
  HipTLUT *colorMap() { return tlut; }  // the same object with the GL-free opacity correction
  // one frame; mv = column-major modelview as glGetDoublev returns it (VolumeRenderable.cpp:47-48)
  void renderVolume(float sampleRate, double mv[16]);
  // a smaller axis-aligned box of the volume (VolumeRenderer.h:103-108): extents in volume space
  void renderVolume(float sampleRate, double mv[16], float xext[2], float yext[2], float zext[2]);
  // one textured quad blended into the frame of the last renderVolume (VolumeRenderer.h:114; camera of that call)
  void renderSlice(float quad[4][3], float alpha);
  // bounding box and brackets (VolumeRenderer.h:122-123): GL line drawing in the reference; kept so that callers
  // compile, and readable by a host that draws its own overlay
  void useBBox(int on_off) { m_bb = on_off; }
  void useBBoxBrackets(int on_off) { m_bbb = on_off; }
  int bbox() const { return m_bb; }
  int bboxBrackets() const { return m_bbb; }
  // the frame of the last renderVolume: [height][width][4] premultiplied float RGBA
  const float *framebuffer() const { return fb.data(); }
  int ok() const { return ctx != nullptr && !failed; }
  smk_ctx *context() { return ctx; }
  void loadTransferTableRGBA();  // what TLUT::loadTransferTableRGBA did with the GL color table
  // MetaVolume::hist2D (MetaVolume.cpp:1650-1688) for the TF window's background: 256x256 bytes, 1 = ok
  int hist2D(unsigned char *hist);

 private:
  int upload(Volume *v, int n);
  smk_ctx *ctx;
  MetaVolume *m_vol;
  HipTLUT *tlut;
  std::vector<float> fb;
  int failed;
  int m_bb;
  int m_bbb;
};

class HipVolumeRenderable final : public gluvvPrimitive {
 public:
  explicit HipVolumeRenderable(int device = 0) : volren(nullptr), go(0), device(device) {}
  ~HipVolumeRenderable() { delete volren; }  // (gluvvPrimitive's destructor is not virtual, gluvvPrimitive.h:26: delete through this type)
  void init();  // virtual in gluvvPrimitive (gluvvPrimitive.h:29-30)
  void draw();
  const float *framebuffer() const { return volren ? volren->framebuffer() : nullptr; }
  int running() const { return go; }
  HipVolumeRenderer *renderer() { return volren; }  // the inner interface (VolumeRenderable keeps it private; a host that
                                                    // draws sub-boxes or slice quads needs it)
  static void modelview(double mv[16]);             // what draw() hands renderVolume: LookAt * T(trans) * R(xform) * T(-fSize/2)

 private:
  void createNoiseTex(int sx, int sy, int sz);  // R8kVolRen3D_cpy::createNoiseTex (:2392-2436)
  HipVolumeRenderer *volren;
  int go;
  int device;
  std::vector<unsigned char> noise;  // [sz][sy][sx][4]
};
