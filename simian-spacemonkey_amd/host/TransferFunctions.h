// TransferFunctions.h -- the classification widgets' rasterisers as headless functions: what
// TFWidgetRen / LevWidget paint into gluvv.volren.deptex / deptex2 / the dense 3-D table, without
// GLUT, picking or drawing.  The renderer consumes the resulting byte tables through
// smk_set_tf2d / smk_set_tf3d; nothing here touches the GPU.
//
//   LevWidget::rasterize      LevWidget.cpp:674-1074 (triangle :704-761, ellipse :764-900,
//                             1-D style :903-1019, default style :1022-1072), setPos :1098-1125
//   HSLPicker::getColor       HSLPicker.cpp:52-93, updateHL :44-48
//   TFWidgetRen::rasterizevgH TFWidgetRen1.cpp:1035-1083 (VGH / V1GH branch)
#pragma once

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

}  // namespace smktf
