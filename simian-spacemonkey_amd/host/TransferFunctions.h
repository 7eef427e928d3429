// TransferFunctions.h -- the classification widgets' rasterisers as headless functions: what
// TFWidgetRen / LevWidget paint into gluvv.volren.deptex / deptex2 / the dense 3-D table, without
// GLUT, picking or drawing.  The renderer consumes the resulting byte tables through
// smk_set_tf2d / smk_set_tf3d; nothing here touches the GPU.
//
//   LevWidget::rasterize      LevWidget.cpp:674-1074 (triangle :704-761, ellipse :764-900,
//                             1-D style :903-1019, default style :1022-1072), setPos :1098-1125
//   HSLPicker::getColor       HSLPicker.cpp:52-93, updateHL :44-48
//   TFWidgetRen::rasterizevgH TFWidgetRen1.cpp:1035-1083 (VGH / V1GH branch)
//   TFWidgetRen frame logic   TFWidgetRen1.cpp:194-242 (clear paint / paint / drop / regenerate), :625-660 (init)
//   TFWidgetRen::drawProbe    TFWidgetRen1.cpp:309-595 (voxel under the probe -> transfer-function domain,
//                             brush placement), triLerpV3 :600-621; DPWidgetRen::update_pos DPWidgetRen.cpp:278-317
#pragma once
#include <vector>

namespace smktf {

enum WidgetShape { LWtriangle = 0, LWsquare = 1, LW1d = 2, LWdef = 3 };  // LevWidget.h:117-120

struct LevWidgetState {
  WidgetShape type = LWtriangle;
  float bottom[2] = {.5f, 0}, left[2] = {.3f, .7f}, right[2] = {.7f, .7f};  // verts[0..2]
  float thresh[2] = {.5f, .35f};
  float color[3] = {1, 0, 0};  // HSL (0, 1, .5)
  float alpha = .5f;
  float boundary_emphasis = 1;  // `be`: alpha scale of every sheet but the second
  bool faux_shading = false;    // gluvv.shade == gluvvShadeFaux: the colour weight follows the ramp
};

// HSLPicker::getColor
void hsl_to_rgb(float h, float s, float l, float rgb[3]);

// LevWidget::setPos: vertices clamped to [0,1]; tw / th = -10 selects the default thresholds
void set_positions(LevWidgetState *w, const float b[2], const float l[2], const float r[2], float tw = -10, float th = -10);

// paint one widget into tex[sh][sg][sv][4] (straight colour, alpha = opacity), blending with what
// is already there exactly as the reference does
void rasterize(const LevWidgetState &w, unsigned char *tex, int sv, int sg, int sh);

// third-axis (second derivative) alpha ramp into the alpha bytes of ptex[sy][sx][4]
void rasterize_vgh(unsigned char *ptex, int sx, int sy, float slider1hi);

// ---- the data probe: where a point of the volume sits in the transfer-function domain ("dual-domain"
// interaction), and the paint brush that follows it

enum Brush { NoBrush, EllipseBrush, AutoEllipseBrush, TriangleBrush, OneDBrush, AutoOneDBrush };  // gluvvBrush, gluvv.h:185-192
// gluvvDataMode values the probe distinguishes (gluvv.h:221-235)
enum { DM_V1 = 0, DM_V1G = 1, DM_V1GH = 2, DM_V2 = 3, DM_V2G = 4, DM_V2GH = 5, DM_V3 = 6, DM_V3G = 7, DM_V4 = 8, DM_VGH = 9, DM_VGH_VG = 10, DM_VGH_V = 11 };

struct ProbeSample {
  bool inside = false;   // the probe's voxel cell lies inside the volume (1 .. size-2, as the reference tests)
  int cell[3] = {0, 0, 0};
  float corners[8][3];   // the cell's eight voxels as transfer-function coordinates in [0,1] (index z*4 + y*2 + x)
  float hessian_pos[8];  // VGH modes: where the corner sits on the third axis' display, (sign(h) sqrt|h| + 1) / 2
  float value[3];        // the trilinear value at the probe = its position in the transfer-function domain
};

// DPWidgetRen::update_pos: widget tip in world space -> volume coordinates in [0,1]^3 (gluvv.probe.vpos):
// inverse of T(trans) R(xform) S(scale) T(-size/2) S(size) applied to the point, float arithmetic as there
void probe_world_to_volume(const float pos[3], const float trans[3], const float xform[16], float scale, const float fsize[3], float vpos[3]);

// TFWidgetRen::drawProbe's data half: data = [z][y][x][nelts] bytes of the (unbricked) volume
void probe_sample(const unsigned char *data, int nelts, int sx, int sy, int sz, int dmode, const float vpos[3], ProbeSample *out);

// ... and its brush half (:497-560): where the brush widget goes for this sample; `slider` = gluvv.probe.slider
void place_brush(LevWidgetState *brush, Brush kind, int dmode, const ProbeSample &s, float slider);

// ---- the transfer-function frame: the state TFWidgetRen owns between frames (paint layer, widget list,
// brush) and the three things it does with it when a flag is raised
struct TFFrame {
  int sv, sg, sh, dmode;
  bool faux_shading = true;                // gluvv.shade == gluvvShadeFaux (the start-up default)
  std::vector<unsigned char> paintex;      // what has been painted so far
  std::vector<LevWidgetState> widgets;     // newest first, as LevWidget::insert links them behind the root
  LevWidgetState brush;                    // alpha .7, ellipse, at TFWidgetRen::init's position
  bool brushon = false;                    // gluvv.tf.brushon
  Brush brush_kind = NoBrush;              // gluvv.probe.brush
  TFFrame(int sv, int sg, int sh, int dmode);
  void clear_paint();                      // gluvv.tf.clearpaint
  void paint();                            // gluvv.tf.paintme: the brush into the paint layer, or (triangle brush) a new widget
  void drop();                             // gluvv.tf.dropme: the brush becomes a widget
  // gluvv.tf.loadme (:232-242): deptex3 cleared, deptex <- paint layer, widgets oldest first, then the brush if on
  void regenerate(unsigned char *deptex, unsigned char *deptex3_or_null) const;
};

}  // namespace smktf
