// gluvv_compat.h -- the slice of Simian's data model a renderer touches, declared with the SAME
// names and meaning as the reference so the adapter below reads like a reference renderer and
// can be compiled and tested without the reference tree (which needs GLUT/GLUI/WGL).
// In a real integration this header is NOT used: the adapter includes the reference's own
// MetaVolume.h / TLUT.h / gluvv.h / gluvvPrimitive.h instead.
//
//   Volume, MetaVolume ........ MetaVolume.h:18-170 (fields a renderer reads)
//   TLUT ...................... TLUT.h:16-116, TLUT.cpp:26-36, 138-154
//   gluvvGlobal (subset) ...... gluvv.h:29-275
//   gluvvPrimitive ............ gluvvPrimitive.h:23-55
#pragma once
#include <cmath>
#include <cstring>

class Volume {
 public:
  int xiSize = 0, yiSize = 0, ziSize = 0;
  float xfSize = 0, yfSize = 0, zfSize = 0;
  int xiPos = 0, yiPos = 0, ziPos = 0;
  float xfPos = 0, yfPos = 0, zfPos = 0;
  unsigned char *currentData = nullptr;  // [z][y][x][nelts]
  unsigned char *currentGrad = nullptr;  // [z][y][x][3] or null
};

class MetaVolume {
 public:
  Volume *volumes = nullptr;  // numSubVols bricks
  int numSubVols = 0;
  int nelts = 1;
  int xiSize = 0, yiSize = 0, ziSize = 0;
  float xfSize = 0, yfSize = 0, zfSize = 0;
};

class TLUT {
 public:
  explicit TLUT(int size = 256) : _size(size), _rgba(new float[4 * size]), lastSampleRate(1.0f) {
    for (int n = 0; n < size; ++n) {
      _rgba[4 * n] = _rgba[4 * n + 1] = _rgba[4 * n + 2] = n / (float)(size - 1.0);
      _rgba[4 * n + 3] = (float)(1.0 / size);
    }
  }
  ~TLUT() { delete[] _rgba; }
  int GetSize() const { return _size; }
  float *GetRGBA(int n) const { return &_rgba[4 * n]; }
  void SetRGBA(int n, float r, float g, float b, float a) {
    float *p = GetRGBA(n);
    p[0] = r; p[1] = g; p[2] = b; p[3] = a;
  }
  // a <- 1-(1-a)^(lastSR/SR); returns true when the table changed (the reference re-uploads then)
  bool scaleAlpha(float sampleRate) {
    if (lastSampleRate == sampleRate) return false;
    float alphaScale = lastSampleRate / sampleRate;
    lastSampleRate = sampleRate;
    for (int i = 0; i < _size; ++i) _rgba[4 * i + 3] = (float)(1 - pow((1 - _rgba[4 * i + 3]), alphaScale));
    return true;
  }

 private:
  int _size;
  float *_rgba;
  float lastSampleRate;
};

typedef enum { VolRenAxisUnknown, VolRenAxisXPos, VolRenAxisXNeg, VolRenAxisYPos, VolRenAxisYNeg, VolRenAxisZPos, VolRenAxisZNeg } VolRenMajorAxis;  // gluvv.h:136-144
typedef enum { gluvvShadeUnknown, gluvvShadeAmb, gluvvShadeDiff, gluvvShadeDSpec, gluvvShadeFaux, gluvvShadeArb, gluvvShadeMIP } gluvvShade;
typedef enum {
  GDM_V1, GDM_V1G, GDM_V1GH, GDM_V2, GDM_V2G, GDM_V2GH, GDM_V3, GDM_V3G, GDM_V4, GDM_VGH, GDM_VGH_VG, GDM_VGH_V, GDM_UNKNOWN
} gluvvDataMode;

struct gluvvGlobal {
  struct { unsigned int width = 512, height = 512; } win;
  struct {
    float eye[3] = {0, 0, -7}, at[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    float frustum[4] = {-.2f, .2f, -.2f, .2f};
    float clip[2] = {1, 20};
    int bgColor = 0;
  } env;
  struct { float pos[3] = {0, 0, -5}; float amb = .05f, intens = .75f; } light;
  struct { float xform[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; float scale = 1; float trans[3] = {0, 0, 0}; } rinfo;
  struct {
    float sampleRate = 2.5f, interactSamp = .6f, goodSamp = 2.5f;
    TLUT *tlut = nullptr;
    unsigned char *deptex = nullptr, *deptex2 = nullptr;
    int loadTLUT = 0, scaleAlphas = 1;
    float gamma = 1;
  } volren;
  struct { int ptexsz[3] = {256, 256, 1}; int numelts = 4; } tf;
  struct {
    int on = 0, ortho = 1;
    VolRenMajorAxis oaxis = VolRenAxisXPos;
    float vpos[3] = {0, 0, 0};   // plane position in volume space (orthogonal mode)
    float pos[3] = {0, 0, 0};    // plane position in world space (free mode)
    float xform[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // its orientation
  } clip;  // gluvvClip (gluvv.h:163-175), the fields the renderer reads
  struct { int on = 0; float weights[10] = {.2f, 0, 0, 0}, scales[10] = {.2f, 2.1f, 4.5f, 8.7f}; } pert;
  int picking = 0;
  int reblend = 0;
  MetaVolume *mv = nullptr;
  gluvvShade shade = gluvvShadeFaux;
  gluvvDataMode dmode = GDM_V1;
};
extern gluvvGlobal gluvv;  // "This needs to be declared in the main function!" (gluvv.h:270)

class gluvvPrimitive {
 public:
  gluvvPrimitive() : next(nullptr) {}
  virtual ~gluvvPrimitive() {}
  virtual void init() {}
  virtual void draw() {}
  virtual int key(unsigned char, int, int) { return 0; }
  virtual int special(int, int, int) { return 0; }
  virtual int pick(int, int, int, float, float, float) { return 0; }
  virtual int pick() { return 0; }
  virtual int mouse(int, int, int, int) { return 1; }
  virtual int move(int, int) { return 1; }
  virtual int release() { return 0; }
  void setNext(gluvvPrimitive *p) {  // LIFO insert (gluvvPrimitive.cpp:163-167)
    p->next = next;
    next = p;
  }
  gluvvPrimitive *getNext() { return next; }

 private:
  gluvvPrimitive *next;
};
