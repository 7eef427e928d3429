// TransferFunctions.cpp -- see TransferFunctions.h.  The arithmetic types matter (bytes must come
// out as the reference's): where the reference mixes float variables with double literals the
// expression is evaluated in double and truncated to the byte, and that is spelled out here.
#include "TransferFunctions.h"

#include <cmath>
#include <cstddef>

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

}  // namespace smktf
