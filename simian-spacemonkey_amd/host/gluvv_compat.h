// gluvv_compat.h -- the slice of Simian's data model a renderer touches, declared with the SAME
// names, types and member signatures as the reference so the adapter can be compiled and tested
// without the reference tree (whose gluvv.h pulls in GLUT/GLUI through TFWindow.h/LTWidgetRen.h).
// In a real integration this header is NOT used: the adapter is built with
// -DSMK_USE_REFERENCE_HEADERS and includes the reference's own MetaVolume.h / TLUT.h / gluvv.h /
// gluvvPrimitive.h instead.
//
//   Volume, MetaVolume ........ MetaVolume.h:18-170
//   TLUT ...................... TLUT.h:16-116 (inline accessors :121-200), TLUT.cpp:26-42, 125-154
//   gluvv structs + enums ..... gluvv.h:29-268
//   gluvvPrimitive ............ gluvvPrimitive.h:23-55, gluvvPrimitive.cpp:101-167
//
// Rules this file keeps (tests/test_compat_signatures.py checks them against the reference text,
// and compiles the adapter against the reference's real headers):
//   * every class / struct / enum here exists in the reference under the same name;
//   * every data member has the reference's type; every member function the reference's return
//     type, parameter types and constness -- a SUBSET of the reference's members, never an addition;
//   * enumerators keep the reference's order (their values cross the C ABI as integers).
// What differs, and why it is harmless: member functions have inline bodies here (the reference
// defines them in .cpp files that need OpenGL); TLUT::loadTransferTableRGBA uploads nothing (the
// adapter sends the table through smk_set_tlut1d instead of the GL colour table).
#pragma once
#include <math.h>
#include <string.h>

class Volume {
 public:
  Volume();
  ~Volume();

  int subVolNum;

  int xiSize, yiSize, ziSize;
  int xiVSize, yiVSize, ziVSize;
  float xfSize, yfSize, zfSize;
  float xfVSize, yfVSize, zfVSize;

  int xiPos, yiPos, ziPos;
  float xfPos, yfPos, zfPos;

  unsigned char *dbData[2];
  unsigned char *dbGrad[2];
  unsigned char *currentData;  // [z][y][x][nelts]
  unsigned char *currentGrad;  // [z][y][x][3] or null

  int extradivs;

  void *nativeData;

  int timestep;
};

// (the reference's constructor also sets up its time-step cache; MetaVolume.cpp:36-75)
inline Volume::Volume() { memset(this, 0, sizeof *this); }
inline Volume::~Volume() {}  // voxel arrays belong to whoever loaded them (VolumeFiles.h: LoadedVolume)

class MetaVolume {
 public:
  MetaVolume();
  ~MetaVolume();

  int isValid(void) { return valid; };

  int valid;

  int dataType;
  int dataEndian;

  int tsteps;
  int tstart;
  int tstop;
  int currentTStep;
  int tstepCache;

  int xiSize, yiSize, ziSize;
  int xiVSize, yiVSize, ziVSize;
  float xfSize, yfSize, zfSize;
  float xfVSize, yfVSize, zfVSize;
  float xSpc, ySpc, zSpc;
  int numSubVols;
  int nelts;
  int nVelts;
  int dims;

  Volume *volumes;
  Volume *wholeVol;
};

inline MetaVolume::MetaVolume() {
  memset(this, 0, sizeof *this);
  nelts = 1;  // "usualy 1" (MetaVolume.h:150)
}
inline MetaVolume::~MetaVolume() {}

class TLUT {
 public:
  typedef enum {
    _numElts = 4
  } TLUTEnums;

  TLUT(const int size = 256);
  ~TLUT();

  void loadTransferTableRGBA();

  void channelConstant(const int channelIndex, const float alpha);
  void channelRamp(const int channelIndex, const int startIndex, const int stopIndex, const float startValue, const float stopValue);

  void scaleAlpha(float sampleRate);

  void alphaConstant(const float alpha) { channelConstant(3, alpha); }
  void alphaRamp(const int startIndex, const int stopIndex, const float startValue, const float stopValue) { channelRamp(3, startIndex, stopIndex, startValue, stopValue); }

  void rgbGrayScaleRamp();

  inline int GetSize() const;
  inline float *GetRGBA(const float t) const;
  inline float *GetRGBA(const int n) const;
  inline void SetRGBA(const int n, float r, float g, float b, float a);
  inline void SetRGB(const int n, float r, float g, float b);
  inline void SetAlpha(const int n, float a);

 protected:
  float *theTable;
  float *_rgba;
  int _size;
  float _alpha;
  float lastSampleRate;
  float lastAlphaScale;
};

inline TLUT::TLUT(const int size) {  // TLUT.cpp:26-36
  _size = size;
  _rgba = new float[_numElts * size];
  theTable = new float[_numElts * size];
  _alpha = (float)(1.0 / _size);
  rgbGrayScaleRamp();
  alphaConstant(_alpha);
  lastSampleRate = 1.0f;
  lastAlphaScale = 1.0f;
}
inline TLUT::~TLUT() {
  delete[] _rgba;
  delete[] theTable;
}
inline void TLUT::loadTransferTableRGBA() {  // TLUT.cpp:65-71 without the glColorTable call
  for (int n = 0; n < _size; ++n) {
    float *rgba = this->GetRGBA(n);
    theTable[n * _numElts + 0] = rgba[0] * rgba[3];
    theTable[n * _numElts + 1] = rgba[1] * rgba[3];
    theTable[n * _numElts + 2] = rgba[2] * rgba[3];
    theTable[n * _numElts + 3] = rgba[3];
  }
}
inline void TLUT::channelConstant(const int channelIndex, const float alpha) {  // TLUT.cpp:112-123
  for (int n = 0; n < _size; ++n) _rgba[n * _numElts + channelIndex] = alpha;
}
inline void TLUT::channelRamp(const int channelIndex, const int startIndex, const int stopIndex, const float startValue, const float stopValue) {  // :125-136
  float denom = stopIndex - startIndex;
  float range = stopValue - startValue;
  for (int n = startIndex; n <= stopIndex; ++n) _rgba[n * _numElts + channelIndex] = startValue + range * (n - startIndex) / denom;
}
inline void TLUT::scaleAlpha(float sampleRate) {  // TLUT.cpp:138-154
  if (lastSampleRate == sampleRate) return;
  float alphaScale = lastSampleRate / sampleRate;
  lastSampleRate = sampleRate;
  for (int i = 0; i < _size; ++i) _rgba[i * _numElts + 3] = 1 - pow((1 - _rgba[i * _numElts + 3]), alphaScale);
  loadTransferTableRGBA();
}
inline void TLUT::rgbGrayScaleRamp() {  // TLUT.cpp:192-199
  for (int n = 0; n < _size; ++n) _rgba[n * _numElts] = _rgba[n * _numElts + 1] = _rgba[n * _numElts + 2] = n / (float)(_size - 1.0);
}
inline int TLUT::GetSize() const { return _size; }
inline float *TLUT::GetRGBA(const float t) const {
  int offset = (int)(t * (_size - 1));
  return &(_rgba[offset * _numElts]);
}
inline float *TLUT::GetRGBA(const int n) const { return &(_rgba[n * _numElts]); }
inline void TLUT::SetRGBA(const int n, float r, float g, float b, float a) {
  SetRGB(n, r, g, b);
  SetAlpha(n, a);
}
inline void TLUT::SetRGB(const int n, float r, float g, float b) {
  int offset = n * _numElts;
  _rgba[offset] = r;
  _rgba[offset + 1] = g;
  _rgba[offset + 2] = b;
}
inline void TLUT::SetAlpha(const int n, float a) {
  int offset = n * _numElts;
  _rgba[offset + 3] = a;
}

class gluvvPrimitive {
 public:
  gluvvPrimitive();
  ~gluvvPrimitive();

  virtual void init();
  virtual void draw();
  virtual int key(unsigned char k, int x, int y);
  virtual int special(int k, int x, int y);
  virtual int pick(int data1, int data2, int data3, float x, float y, float z);
  virtual int pick();
  virtual int mouse(int button, int state, int x, int y);
  virtual int move(int x, int y);
  virtual int release();

  void setNext(gluvvPrimitive *p);
  gluvvPrimitive *getNext();

  void setName(const char *PName);

 private:
  gluvvPrimitive *next;
  char *name;
};

// defaults of gluvvPrimitive.cpp:101-167
inline gluvvPrimitive::gluvvPrimitive() : next(0), name(0) {}
inline gluvvPrimitive::~gluvvPrimitive() {}  // NOT virtual in the reference either (gluvvPrimitive.h:26)
inline void gluvvPrimitive::init() {}
inline void gluvvPrimitive::draw() {}
inline int gluvvPrimitive::key(unsigned char, int, int) { return 0; }
inline int gluvvPrimitive::special(int, int, int) { return 0; }
inline int gluvvPrimitive::pick(int, int, int, float, float, float) { return 0; }
inline int gluvvPrimitive::pick() { return 0; }
inline int gluvvPrimitive::mouse(int, int, int, int) { return 1; }
inline int gluvvPrimitive::move(int, int) { return 1; }
inline int gluvvPrimitive::release() { return 0; }
inline void gluvvPrimitive::setNext(gluvvPrimitive *p) {  // LIFO insert (gluvvPrimitive.cpp:163-167)
  p->next = next;
  next = p;
}
inline gluvvPrimitive *gluvvPrimitive::getNext() { return next; }
inline void gluvvPrimitive::setName(const char *) {}

struct gluvvWindow {
  unsigned int width;
  unsigned int height;
  unsigned int xPos;
  unsigned int yPos;
};

struct gluvvEnv {
  float eye[3];
  float at[3];
  float up[3];
  float frustum[4];
  float clip[2];
  float diff[4];
  float spec[4];
  int bgColor;
};

struct gluvvLight {
  float pos[3];
  float startpos[3];
  float color[3];
  float amb;
  float intens;
  float mv[16];
  float pj[16];
  float xf[16];
  int buffsz[2];
  int shadow;
  int softShadow;
  int showView;
  int shadowTF;
  int showShadowTF;
  int sill;
  float gShadowQual;
  float iShadowQual;
  unsigned int cubeName;
  float *cubeKey[6];
  int csz;
  int load;
  int gload;
  int fog;
  float fogColor[3];
  float fogThresh;
  float fogLimits[2];
  int latt;
  float lattThresh;
  float lattLimits[2];
};

struct gluvvRInfo {
  float xform[16];
  float scale;
  float trans[3];
};

struct gluvvVolRen {
  float sampleRate;
  float interactSamp;
  float goodSamp;
  int shade;
  TLUT *tlut;
  unsigned char *deptex;
  unsigned int deptexName;
  unsigned char *deptex2;
  unsigned int deptex2Name;
  unsigned char *deptex3;
  unsigned int deptex3Name;
  int loadTLUT;
  int scaleAlphas;
  float gamma;
  int timestep;
};

typedef enum {
  VolRenAxisUnknown,
  VolRenAxisXPos,
  VolRenAxisXNeg,
  VolRenAxisYPos,
  VolRenAxisYNeg,
  VolRenAxisZPos,
  VolRenAxisZNeg
} VolRenMajorAxis;

struct gluvvTF {
  int loadme;
  int paintme;
  int dropme;
  int clearpaint;
  int ptexsz[3];
  int numelts;
  int brushon;
  float slider1;
  float slider1hi;
  float slider1lo;
  float slider2;
  int histOn;
};

struct gluvvClip {
  int on;
  int ortho;
  VolRenMajorAxis oaxis;
  float xform[16];
  unsigned int pname;
  float alpha;
  float pos[3];
  float vpos[3];
  float dir[3];
};

struct gluvvPert {
  int on;
  int numHarm;
  float weights[10];
  float scales[10];
};

typedef enum {
  NoBrush,
  EllipseBrush,
  AutoEllipseBrush,
  TriangleBrush,
  OneDBrush,
  AutoOneDBrush
} gluvvBrush;

struct gluvvProbe {
  float vpos[3];
  float slider;
  gluvvBrush brush;
};

typedef enum {
  gluvvShadeUnknown,
  gluvvShadeAmb,
  gluvvShadeDiff,
  gluvvShadeDSpec,
  gluvvShadeFaux,
  gluvvShadeArb,
  gluvvShadeMIP
} gluvvShade;

typedef enum {
  GPGineric,
  GPOctane,
  GPOctane2,
  GPInfinite,
  GPNV15,
  GPNV20,
  GPNV202D,
  GPWildcat,
  GPATI8K
} gluvvPlatform;

typedef enum {
  GDM_V1,
  GDM_V1G,
  GDM_V1GH,
  GDM_V2,
  GDM_V2G,
  GDM_V2GH,
  GDM_V3,
  GDM_V3G,
  GDM_V4,
  GDM_VGH,
  GDM_VGH_VG,
  GDM_VGH_V,
  GDM_UNKNOWN
} gluvvDataMode;

typedef enum {
  GB_NONE,
  GB_UNDER,
  GB_OVER,
  GB_ZERO
} gluvvBlend;

struct gluvvGlobal {
  int debug;
  gluvvWindow win;
  gluvvEnv env;
  gluvvLight light;
  gluvvRInfo rinfo;
  gluvvPlatform plat;
  int picking;
  gluvvVolRen volren;
  gluvvTF tf;
  gluvvClip clip;
  gluvvProbe probe;
  int mprobe;
  MetaVolume *mv;
  MetaVolume *mv1;
  MetaVolume *mv2;
  MetaVolume *mv3;
  int mainWindow;
  gluvvShade shade;
  gluvvDataMode dmode;
  gluvvBlend reblend;
  gluvvPert pert;
};

extern gluvvGlobal gluvv;  // "This needs to be declared in the "main" function!" (gluvv.h:270-273)

// initGluvv()'s defaults for the fields a renderer reads (gluvv.cpp:240-368; gluvvui.cpp:213-267 for
// the perturbation block).  A stand-alone driver calls this once; inside Simian initGluvv does.
inline void gluvvCompatDefaults(gluvvGlobal &g) {
  memset(&g, 0, sizeof g);
  g.win.width = g.win.height = 512;
  g.env.eye[2] = -7;
  g.env.up[1] = 1;
  g.env.frustum[0] = g.env.frustum[2] = -.2f;
  g.env.frustum[1] = g.env.frustum[3] = .2f;
  g.env.clip[0] = 1;
  g.env.clip[1] = 20;
  g.light.pos[2] = -5;
  g.light.amb = .05f;
  g.light.intens = .75f;
  g.light.buffsz[0] = g.light.buffsz[1] = 1024;
  g.light.gShadowQual = .5f;   // gluvv.cpp:299-300
  g.light.iShadowQual = .2f;
  for (int i = 0; i < 4; ++i) g.rinfo.xform[5 * i] = g.clip.xform[5 * i] = 1;
  g.rinfo.scale = 1;
  g.volren.sampleRate = g.volren.goodSamp = 2.5f;
  g.volren.interactSamp = .6f;
  g.volren.scaleAlphas = 1;
  g.volren.gamma = 1;
  g.tf.ptexsz[0] = g.tf.ptexsz[1] = 256;
  g.tf.ptexsz[2] = 1;
  g.tf.numelts = 4;
  g.tf.slider1 = g.tf.slider1hi = 1;
  g.clip.ortho = 1;
  g.clip.oaxis = VolRenAxisXPos;
  g.pert.weights[0] = .2f;
  g.pert.scales[0] = .2f;
  g.pert.scales[1] = 2.1f;
  g.pert.scales[2] = 4.5f;
  g.pert.scales[3] = 8.7f;
  g.plat = GPGineric;
  g.shade = gluvvShadeFaux;
  g.dmode = GDM_V1;
  g.reblend = GB_NONE;
}
