// TransferFunctions.cpp -- see TransferFunctions.h.  The arithmetic types matter (bytes must come
// out as the reference's): where the reference mixes float variables with double literals the
// expression is evaluated in double and truncated to the byte, and that is spelled out here.
#include "TransferFunctions.h"

#include <cmath>
#include <cstddef>
#include <algorithm>
#include <cstring>

namespace smktf {
namespace {

inline unsigned char to_byte(double x) {  // C conversion double -> unsigned char through int; out of range -> 0
  if (!(x > -2147483649.0 && x < 2147483648.0)) return 0;
  return (unsigned char)(int)x;
}

inline double ramp(double i0, double x, double i1, double o0, double o1) {  // VectorMath.h affine()
  return (o1 - o0) * (x - i0) / (i1 - i0) + o0;
}

struct Texel {
  unsigned char *p;
  // colour = alpha-weighted average of what is there and the widget's; `weight` is the widget's
  // alpha, or alpha times the ramp under faux shading
  void blend_colour(const float rgb[3], float weight, float new_alpha) const {
    const float old_alpha = p[3] / 255.0f;
    for (int e = 0; e < 3; ++e) p[e] = to_byte((old_alpha * p[e] / 255.0 + weight * rgb[e]) / (old_alpha + new_alpha) * 255);
  }
  void alpha_max(float a, float scale) const {  // triangle: the larger of old and new
    const float n = a * 255;
    p[3] = to_byte((n > p[3] ? n : (float)p[3]) * scale);
  }
  void alpha_over(float a, float scale) const {  // the other shapes: new over old
    p[3] = to_byte((a * 255 + (1.0 - a) * p[3]) * scale);
  }
};

struct Canvas {
  unsigned char *tex;
  int sv, sg;
  Texel at(int sheet, int line, int col) const {
    return Texel{tex + ((size_t)sheet * sg + line) * (size_t)sv * 4 + (size_t)col * 4};
  }
};

}  // namespace

void hsl_to_rgb(float H, float S, float L, float col[3]) {
  if (S == 0) {
    col[0] = col[1] = col[2] = L;
    return;
  }
  const float m2 = L <= 0.5 ? L * (1 + S) : L + S - L * S;
  const float m1 = 2 * L - m2;
  if (1.0 == H) H = 0;
  H *= 6;
  const int sextant = (int)floor(H);
  const float fract = H - sextant;
  const float mid1 = m1 + fract * (m2 - m1), mid2 = m2 + fract * (m1 - m2);
  switch (sextant) {
    case 0: col[0] = m2; col[1] = mid1; col[2] = m1; break;
    case 1: col[0] = mid2; col[1] = m2; col[2] = m1; break;
    case 2: col[0] = m1; col[1] = m2; col[2] = mid1; break;
    case 3: col[0] = m1; col[1] = mid2; col[2] = m2; break;
    case 4: col[0] = mid1; col[1] = m1; col[2] = m2; break;
    case 5: col[0] = m2; col[1] = m1; col[2] = mid2; break;
    default: break;  // (H outside [0,1]: the reference leaves col untouched)
  }
}

void set_positions(LevWidgetState *w, const float b[2], const float l[2], const float r[2], float tw, float th) {
  auto unit = [](float x) { return x > 0 ? (x < 1 ? x : 1.0f) : 0.0f; };
  for (int a = 0; a < 2; ++a) {
    w->bottom[a] = unit(b[a]);
    w->left[a] = unit(l[a]);
    w->right[a] = unit(r[a]);
  }
  if (th == -10) w->thresh[1] = w->bottom[1] + (w->left[1] - w->bottom[1]) / 2;
  else w->thresh[1] = th > w->bottom[1] ? (th < w->left[1] ? th : w->left[1]) : w->bottom[1];
  if (tw == -10) w->thresh[0] = w->left[0] + (w->right[0] - w->left[0]) / 2;
  else w->thresh[0] = tw < w->right[0] ? (tw > w->left[0] ? tw : w->left[0]) : w->right[0];
}

void rasterize(const LevWidgetState &w, unsigned char *tex, int sv, int sg, int sh) {
  const Canvas cv{tex, sv, sg};
  const int top_line = (int)(w.left[1] * sg) - 1;  // H
  const int first_line = (int)(w.bottom[1] * sg);
  auto sheet_scale = [&](int k) { return k != 1 ? w.boundary_emphasis : 1.0f; };
  auto weight = [&](float a, float ramp_value) { return w.faux_shading ? a * ramp_value : a; };

  switch (w.type) {
    case LWtriangle: {
      const int base = (int)(w.thresh[1] * sg);
      for (int k = 0; k < sh; ++k)
        for (int i = base; i <= top_line; ++i) {
          const float g = i / (float)sg;
          const int start = (int)((w.bottom[0] + g * (w.left[0] - w.bottom[0]) / w.left[1]) * sv);
          const int width = (int)((w.bottom[0] + g * (w.right[0] - w.bottom[0]) / w.left[1]) * sv) + 1 - start;
          for (int j = 0; j < width; ++j) {
            float t = (float)ramp(-1, j, width, -1, 1);  // tent across the scan line
            t = t < 0 ? 1.0f + t : 1.0f - t;
            const float a = t * w.alpha;
            const Texel px = cv.at(k, i, start + j);
            px.blend_colour(w.color, weight(a, t), a);
            px.alpha_max(a, sheet_scale(k));
          }
        }
      break;
    }
    case LWsquare: {
      const int W = (int)((w.right[0] - w.left[0]) * sv);
      const int h = (int)((w.bottom[1] - w.left[1]) * sg);
      const int hc = (int)((w.thresh[0] - w.left[0]) * sv);
      const int vc = (int)(w.thresh[1] * sg);
      float maxd, scaleh, scalew;
      if (W * W < h * h) {
        maxd = (float)((W / 2) * (W / 2));
        scalew = 1.0f;
        scaleh = (W / 2 * W / 2) / (float)(h / 2 * h / 2);
      } else {
        maxd = (float)((h / 2) * (h / 2));
        scaleh = 1.0f;
        scalew = (h / 2 * h / 2) / (float)(W / 2 * W / 2);
      }
      const int start = (int)(w.left[0] * sv);
      for (int k = 0; k < sh; ++k)
        for (int i = first_line; i <= top_line; ++i)
          for (int j = 0; j < W; ++j) {
            const float d = ((i - vc) * (i - vc) * scaleh) + ((j - hc) * (j - hc) * scalew);
            const float t = (float)ramp(0, d, maxd, 1, 0);
            const float a = t > 0 ? (t < 1 ? t * t * w.alpha : w.alpha) : 0;
            const Texel px = cv.at(k, i, start + j);
            px.blend_colour(w.color, weight(a, t), a);
            px.alpha_over(a, sheet_scale(k));
          }
      break;
    }
    case LW1d: {
      const int hc = (int)((w.thresh[0] - w.left[0]) * sv);
      const float vthresh = (w.thresh[1] - w.bottom[1]) / (w.left[1] - w.bottom[1]);
      const int start = (int)(w.left[0] * sv);
      const int dist = (int)(w.right[0] * sv - start);
      const int rise_end = (int)(hc * (1.0 - vthresh) + 1);
      const int fall_begin = (int)(dist - (dist - hc) * (1.0 - vthresh) + 1);
      for (int k = 0; k < sh; ++k)
        for (int i = first_line; i <= top_line; ++i)
          for (int j = 0; j < dist; ++j) {
            float t;
            if (j < rise_end) t = (float)ramp(0, j, rise_end, 0, 1);
            else if (j < fall_begin) t = 1;
            else t = (float)ramp(fall_begin, j, dist, 1, 0);
            const float a = t * w.alpha;
            const Texel px = cv.at(k, i, start + j);
            px.blend_colour(w.color, weight(a, t), a);
            px.alpha_over(a, 1.0f);  // (this shape ignores the boundary emphasis)
          }
      break;
    }
    case LWdef: {
      const float m = (w.thresh[1] - w.bottom[1]) / (w.left[1] - w.bottom[1]);
      const int start = (int)(w.left[0] * sv);
      const int width = (int)(w.right[0] * sv) - start;
      const float hue_step = 1 / ((w.left[0] - w.right[0]) * sv - 1);
      for (int k = 0; k < sh; ++k)
        for (int i = first_line; i <= top_line; ++i) {
          float a = (float)(((i / 255.0) - w.bottom[1]) / (m + (i / 255.0) - w.bottom[1]));  // (255.0 as in the reference, not sg)
          a *= w.alpha;
          a = a > 1 ? 1 : (a < 0 ? 0 : a);
          float hue = 0;  // HSLPicker::reset(0, 1, .5), then one updateHL(hue_step, 0) per pixel
          for (int j = 0; j < width; ++j) {
            hue = (float)((hue + hue_step) > 1.0 ? (hue + hue_step - 1.0) : ((hue + hue_step) < 0 ? (hue + hue_step + 1.0) : (hue + hue_step)));
            float rgb[3];
            hsl_to_rgb(hue, 1, .5f, rgb);
            const Texel px = cv.at(k, i, start + j);
            px.blend_colour(rgb, a, a);
            px.alpha_over(a, sheet_scale(k));
          }
        }
      break;
    }
  }
}

void rasterize_vgh(unsigned char *ptex, int sx, int sy, float slider1hi) {
  const int cent = (int)(sx / 3.0);
  float b = 255 - 20 * cent * (1 - slider1hi);
  const float d = 255 - b;
  float m = (d < 0 ? -d : d) / (float)cent;
  auto clamp255 = [](float x) { return x < 0 ? 0.0f : (x > 255 ? 255.0f : x); };
  for (int i = 0; i < sy; ++i)
    for (int j = 0; j < cent; ++j) ptex[((size_t)i * sx + j) * 4 + 3] = to_byte(clamp255(j * m + b));
  b = 255;
  m = -m;
  for (int i = 0; i < sy; ++i)
    for (int j = 1; j <= cent; ++j) ptex[((size_t)i * sx + j + cent) * 4 + 3] = to_byte(clamp255(j * m + b));
}

// ------------------------------------------------------------------------------------ probe

void probe_world_to_volume(const float pos[3], const float trans[3], const float xform[16], float scale, const float fsize[3], float vpos[3]) {
  // volm = T(trans) * xform * S(scale) * T(-size/2) * S(size)   (DPWidgetRen.cpp:281-306; matrixMult VectorMath.h:424-445)
  auto mul = [](float o[16], const float a[16], const float b[16]) {
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 4; ++r) o[c * 4 + r] = a[r] * b[c * 4] + a[4 + r] * b[c * 4 + 1] + a[8 + r] * b[c * 4 + 2] + a[12 + r] * b[c * 4 + 3];
  };
  auto ident = [](float m[16]) {
    memset(m, 0, 16 * sizeof(float));
    m[0] = m[5] = m[10] = m[15] = 1;
  };
  float m1[16], m2[16], sc[16], m3[16], m4[16], m5[16], volm[16];
  ident(m1);
  m1[12] = trans[0]; m1[13] = trans[1]; m1[14] = trans[2];
  mul(m2, m1, xform);
  ident(sc);
  sc[0] = sc[5] = sc[10] = scale;
  mul(m3, m2, sc);
  ident(m4);
  m4[12] = (float)(-.5 * fsize[0]); m4[13] = (float)(-.5 * fsize[1]); m4[14] = (float)(-.5 * fsize[2]);
  mul(m5, m3, m4);
  float m6[16];
  ident(m6);
  m6[0] = fsize[0]; m6[5] = fsize[1]; m6[10] = fsize[2];
  mul(volm, m5, m6);
  // affine inverse (VolumeRenderer::inverseMatrix's formula, VectorMath.h inverseMatrix), double inside
  const float *m = volm;
  const double det = (double)m[0] * m[5] * m[10] - (double)m[0] * m[6] * m[9] - (double)m[1] * m[4] * m[10] + (double)m[1] * m[6] * m[8] +
                     (double)m[2] * m[4] * m[9] - (double)m[2] * m[5] * m[8];
  double inv[16];
  inv[0] = ((double)m[5] * m[10] - (double)m[6] * m[9]) / det;
  inv[1] = (-(double)m[1] * m[10] + (double)m[2] * m[9]) / det;
  inv[2] = ((double)m[1] * m[6] - (double)m[2] * m[5]) / det;
  inv[4] = (-(double)m[4] * m[10] + (double)m[6] * m[8]) / det;
  inv[5] = ((double)m[0] * m[10] - (double)m[2] * m[8]) / det;
  inv[6] = (-(double)m[0] * m[6] + (double)m[2] * m[4]) / det;
  inv[8] = ((double)m[4] * m[9] - (double)m[5] * m[8]) / det;
  inv[9] = (-(double)m[0] * m[9] + (double)m[1] * m[8]) / det;
  inv[10] = ((double)m[0] * m[5] - (double)m[1] * m[4]) / det;
  inv[12] = -(inv[0] * m[12] + inv[4] * m[13] + inv[8] * m[14]);
  inv[13] = -(inv[1] * m[12] + inv[5] * m[13] + inv[9] * m[14]);
  inv[14] = -(inv[2] * m[12] + inv[6] * m[13] + inv[10] * m[14]);
  for (int a = 0; a < 3; ++a) vpos[a] = (float)(inv[a] * pos[0] + inv[4 + a] * pos[1] + inv[8 + a] * pos[2] + inv[12 + a]);
}

void probe_sample(const unsigned char *dp, int ne, int sx, int sy, int sz, int dmode, const float vpos[3], ProbeSample *out) {
  ProbeSample &s = *out;
  const int px = (int)(vpos[0] * sx), py = (int)(vpos[1] * sy), pz = (int)(vpos[2] * sz);
  const float fpos[3] = {vpos[0] * sx, vpos[1] * sy, vpos[2] * sz};
  s.cell[0] = px; s.cell[1] = py; s.cell[2] = pz;
  memset(s.corners, 0, sizeof s.corners);
  memset(s.hessian_pos, 0, sizeof s.hessian_pos);
  s.value[0] = s.value[1] = s.value[2] = 0;
  s.inside = !((px < 1) || (px > (sx - 2)) || (py < 1) || (py > (sy - 2)) || (pz < 1) || (pz > (sz - 2)));
  if (!s.inside) return;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 2; ++k) {
        const unsigned char *v = dp + ((size_t)(pz + i) * sx * sy * ne) + ((size_t)(py + j) * sx * ne) + ((size_t)(px + k) * ne);
        float *c = s.corners[i * 4 + j * 2 + k];
        switch (dmode) {
          case DM_V1:
            c[0] = (float)(v[0] / 255.0);
            break;
          case DM_V1G: case DM_V2: case DM_V2G:
            c[0] = (float)(v[0] / 255.0);
            c[1] = (float)(v[1] / 255.0);
            break;
          case DM_V2GH: case DM_V3: case DM_V3G: case DM_V4:  // (falls through into the VGH case in the reference)
          case DM_VGH: case DM_V1GH: case DM_VGH_VG: case DM_VGH_V: {
            c[0] = (float)(v[0] / 255.0);
            c[1] = (float)(v[1] / 255.0);
            float h = (float)(v[2] / 85.0 - 1);
            h = h > 0 ? (float)sqrt(h) : (float)-sqrt(-h);
            s.hessian_pos[i * 4 + j * 2 + k] = (float)((h + 1) / 2.0);
            c[2] = (float)(v[2] / 169.0);
            break;
          }
          default: break;
        }
      }
  // triLerpV3 (:600-621): fractions of the floating-point position, x then y then z
  const float fx = fpos[0] - (int)fpos[0], fy = fpos[1] - (int)fpos[1], fz = fpos[2] - (int)fpos[2];
  for (int e = 0; e < 3; ++e) {
    const float x1 = s.corners[0][e] + (float)(s.corners[1][e] - s.corners[0][e]) * fx;
    const float x2 = s.corners[2][e] + (float)(s.corners[3][e] - s.corners[2][e]) * fx;
    const float x3 = s.corners[4][e] + (float)(s.corners[5][e] - s.corners[4][e]) * fx;
    const float x4 = s.corners[6][e] + (float)(s.corners[7][e] - s.corners[6][e]) * fx;
    const float xy1 = x1 + (x2 - x1) * fy, xy2 = x3 + (x4 - x3) * fy;
    s.value[e] = (xy1 + (xy2 - xy1) * fz);
  }
}

void place_brush(LevWidgetState *brush, Brush kind, int dmode, const ProbeSample &s, float slider) {
  if (!s.inside) {  // outside: the brush collapses into the origin (:348-352)
    const float z[2] = {0, 0};
    set_positions(brush, z, z, z);
    return;
  }
  const float *val = s.value;
  auto maxf = [](float a, float b) { return a > b ? a : b; };
  auto minf = [](float a, float b) { return a < b ? a : b; };
  if (kind == EllipseBrush) {
    const float bsz = (float)((1.0 - slider) / 4.0);
    const float l[2] = {val[0] - bsz, val[1] + bsz}, r[2] = {val[0] + bsz, val[1] + bsz}, b[2] = {val[0] + bsz, val[1] - bsz};
    set_positions(brush, b, l, r, val[0], val[1]);
    brush->type = LWsquare;
  } else if (kind == TriangleBrush || kind == AutoEllipseBrush) {
    float maxx = -1000, maxy = -1000, minx = 1000, miny = 1000;
    for (int i = 0; i < 8; ++i) {
      maxx = maxf(maxx, s.corners[i][0]); maxy = maxf(maxy, s.corners[i][1]);
      minx = minf(minx, s.corners[i][0]); miny = minf(miny, s.corners[i][1]);
    }
    const float bsz = (float)((1.0 - slider) * 2);
    const float w = (float)(((maxx - minx) / 2.0) > .01 ? ((maxx - minx) / 2.0) : .01);
    const float h = (float)(((maxy - miny) / 2.0) > .01 ? ((maxy - miny) / 2.0) : .01);
    const float l[2] = {val[0] - w * bsz, val[1] + h * bsz}, r[2] = {val[0] + w * bsz, val[1] + h * bsz};
    float b[2] = {val[0], 0};
    if (kind == TriangleBrush) {
      set_positions(brush, b, l, r, 0, val[1] - h * bsz);
      brush->type = LWtriangle;
    } else {
      b[0] = val[0] + w * bsz;
      b[1] = val[1] - h * bsz;
      set_positions(brush, b, l, r, val[0], val[1]);
      brush->type = LWsquare;
    }
  }
  if (dmode == DM_V1 || dmode == DM_VGH_V || kind == OneDBrush || kind == AutoOneDBrush) {
    float maxx = -1000, minx = 1000;
    for (int i = 0; i < 8; ++i) {
      maxx = maxf(maxx, s.corners[i][0]);
      minx = minf(minx, s.corners[i][0]);
    }
    const float bsz = (float)((1.0 - slider) * 2);
    const float w = (float)(((maxx - minx) / 2.0) > .01 ? ((maxx - minx) / 2.0) : .01);
    float l[2] = {val[0] - w * bsz, 1}, r[2] = {val[0] + w * bsz, 1};
    const float b[2] = {val[0], 0};
    if (kind == OneDBrush) {
      l[0] = (float)(val[0] + (1.0 - slider) / 4.0);
      r[0] = (float)(val[0] - (1.0 - slider) / 4.0);
    }
    set_positions(brush, b, l, r, val[0], val[1]);
    brush->type = LW1d;
  }
}

// ------------------------------------------------------------------------------------ frame

TFFrame::TFFrame(int sv_, int sg_, int sh_, int dmode_) : sv(sv_), sg(sg_), sh(sh_), dmode(dmode_), paintex((size_t)sv_ * sg_ * sh_ * 4, 0) {
  // TFWidgetRen::init (:648-656): position of the root widget, alpha .7, ellipse
  const float b[2] = {.5f, 0}, l[2] = {.3f, .7f}, r[2] = {.7f, .7f};
  set_positions(&brush, b, l, r);
  brush.alpha = .7f;
  brush.type = LWsquare;
}

void TFFrame::clear_paint() { std::fill(paintex.begin(), paintex.end(), (unsigned char)0); }

namespace {
// LevWidget::rasterize's head (:677-682): scalar data modes turn every widget into the 1-D style over the full height
LevWidgetState for_mode(LevWidgetState w, int dmode, bool faux) {
  if (dmode == DM_V1 || dmode == DM_VGH_V) {
    w.bottom[1] = 0;
    w.left[1] = 1;
    w.right[1] = 1;
    w.type = LW1d;
  }
  w.faux_shading = faux;
  return w;
}
}  // namespace

void TFFrame::paint() {
  switch (brush_kind) {
    case EllipseBrush: case AutoEllipseBrush: case OneDBrush: case AutoOneDBrush:
      rasterize(for_mode(brush, dmode, faux_shading), paintex.data(), sv, sg, sh);
      break;
    case TriangleBrush:
      widgets.insert(widgets.begin(), brush);
      break;
    default: break;
  }
}

void TFFrame::drop() { widgets.insert(widgets.begin(), brush); }

void TFFrame::regenerate(unsigned char *deptex, unsigned char *deptex3) const {
  const size_t n = (size_t)sv * sg * sh * 4;
  if (deptex3) memset(deptex3, 0, n);
  memcpy(deptex, paintex.data(), n);
  // the list is root -> newest -> ... -> oldest and every widget rasterises its successors before itself
  for (size_t k = widgets.size(); k-- > 0;) rasterize(for_mode(widgets[k], dmode, faux_shading), deptex, sv, sg, sh);
  if (brushon) rasterize(for_mode(brush, dmode, faux_shading), deptex, sv, sg, sh);
}

}  // namespace smktf
