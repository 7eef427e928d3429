// HipVolumeRenderer.h -- host-side mirror of the reference renderer interface over the C ABI.
//
//   HipVolumeRenderer   same public surface as VolumeRenderer (VolumeRenderer.h:86-118):
//                       createVolume x2, createTLUT, getColorMap, renderVolume(sampleRate, mv)
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
#ifndef SMK_USE_REFERENCE_HEADERS
#include "gluvv_compat.h"
#endif

enum { VolRenUnkown, VolRen2DTexture, VolRen3DTexture, VolRen3DExt };  // VolumeRenderer.h:64-69

class HipVolumeRenderer {
 public:
  HipVolumeRenderer(MetaVolume *vm, int subVolNum, int device = 0);
  ~HipVolumeRenderer();
  int createVolume(int type, Volume *v);             // VolumeRenderer.cpp:101
  int createVolume(int type, Volume *v, int nVols);  // VolumeRenderer.cpp:128
  int createTLUT();                                  // always returns 1, as the reference (:214-219)
  TLUT *getColorMap() { return tlut; }
  // one frame; mv = column-major modelview as glGetDoublev returns it (VolumeRenderable.cpp:47-48)
  void renderVolume(float sampleRate, double mv[16]);
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
  TLUT *tlut;
  std::vector<float> fb;
  int failed;
};

class HipVolumeRenderable : public gluvvPrimitive {
 public:
  explicit HipVolumeRenderable(int device = 0) : volren(nullptr), go(0), device(device) {}
  ~HipVolumeRenderable() override { delete volren; }
  void init() override;
  void draw() override;
  const float *framebuffer() const { return volren ? volren->framebuffer() : nullptr; }
  int running() const { return go; }

 private:
  HipVolumeRenderer *volren;
  int go;
  int device;
};
