/*
 * smk_prep.c -- CPU restatement of the data-preparation and classification-table code on
 * either side of the Simian hot path (inputs the renderer consumes).
 *
 * TEST INFRASTRUCTURE ONLY (see smk_oracle.h): used to build synthetic inputs and golden
 * fixtures and as the checker for the GPU VGH/normal kernels.  PARITY UNPINNED by the
 * reference except for the Perlin functions, which are checked against the reference's own
 * genvol/perlin.c compiled into oracle/_ref/libperlin_ref.so (tests/test_perlin_ref.py).
 *
 * Citations are relative to /root/reference.
 */
#include "smk_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXF(x, y) (((x) > (y)) ? (x) : (y))
#define MINF(x, y) (((x) < (y)) ? (x) : (y))

/* VectorMath.h:70-74 */
static inline double affine(double i, double x, double I, double o, double O) {
  return ((O) - (o)) * ((x) - (i)) / ((I) - (i)) + (o);
}
/* VectorMath.h:60 */
static inline double clamp01(double x) { return x > 0 ? (x < 1 ? x : 1) : 0; }
/* VectorMath.h:65-68 */
static inline double clamp_arb(double c, double x, double C) { return x > c ? (x < C ? x : C) : c; }

/* float -> unsigned char the way x86 does it for the reference's C casts: truncate toward
 * zero through a 32-bit int, keep the low byte; NaN / out-of-int-range give 0 (cvttss2si
 * "integer indefinite" 0x80000000).  Makes (unsigned char)256.0f == 0 explicit. */
static inline unsigned char uc_cast(double x) {
  if (!(x > -2147483649.0 && x < 2147483648.0)) return 0;
  return (unsigned char)(int32_t)x;
}

/* ------------------------------------------------------------ rand(): glibc TYPE_3 clone */
static int32_t rnd_r[34 + 310 + 8];
static uint32_t rnd_ring[34];
static int rnd_pos = 0;

void orc_srand(unsigned seed) {
  int32_t r[344];
  if (seed == 0) seed = 1;
  r[0] = (int32_t)seed;
  for (int i = 1; i < 31; ++i) {
    int64_t w = (16807LL * r[i - 1]) % 2147483647LL;
    if (w < 0) w += 2147483647LL;
    r[i] = (int32_t)w;
  }
  for (int i = 31; i < 34; ++i) r[i] = r[i - 31];
  for (int i = 34; i < 344; ++i) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
  for (int i = 0; i < 34; ++i) rnd_ring[i] = (uint32_t)r[310 + i];
  rnd_pos = 0;
  (void)rnd_r;
}

int orc_rand(void) {
  /* o_k = o_{k-31} + o_{k-3}; ring of the last 34 values */
  uint32_t v = rnd_ring[(rnd_pos + 34 - 31) % 34] + rnd_ring[(rnd_pos + 34 - 3) % 34];
  rnd_ring[rnd_pos] = v;
  rnd_pos = (rnd_pos + 1) % 34;
  return (int)(v >> 1);
}
#define ORC_RAND_MAX 2147483647

/* ------------------------------------------------------------ Perlin (genvol/perlin.c) */
#define PB 0x100
#define PBM 0xff
#define PN 0x1000

static int pp[PB + PB + 2];
static double pg3[PB + PB + 2][3];
static double pg2[PB + PB + 2][2];
static double pg1[PB + PB + 2];
/* perlin.c:13 `static int start = 1`: the first noise call re-runs init() even when the caller
 * already did (genvol main.cpp:118-120 calls init() itself), so genvol's tables are the SECOND
 * set of rand() draws.  orc_perlin_reset() re-arms the flag (a fresh process in the reference). */
static int pstart = 1;
void orc_perlin_reset(void) { pstart = 1; }

static void pnorm2(double v[2]) {
  double s = sqrt(v[0] * v[0] + v[1] * v[1]);
  v[0] /= s;
  v[1] /= s;
}
static void pnorm3(double v[3]) {
  double s = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  v[0] /= s;
  v[1] /= s;
  v[2] /= s;
}

/* perlin.c:145-176 -- the draw order from rand() (g1, g2 x2, g3 x3 per entry, then the
 * permutation shuffle) is what fixes the tables */
void orc_perlin_init(void) {
  int i, j, k;
  for (i = 0; i < PB; i++) {
    pp[i] = i;
    pg1[i] = (double)((orc_rand() % (PB + PB)) - PB) / PB;
    for (j = 0; j < 2; j++) pg2[i][j] = (double)((orc_rand() % (PB + PB)) - PB) / PB;
    pnorm2(pg2[i]);
    for (j = 0; j < 3; j++) pg3[i][j] = (double)((orc_rand() % (PB + PB)) - PB) / PB;
    pnorm3(pg3[i]);
  }
  while (--i) {
    k = pp[i];
    pp[i] = pp[j = orc_rand() % PB];
    pp[j] = k;
  }
  for (i = 0; i < PB + 2; i++) {
    pp[PB + i] = pp[i];
    pg1[PB + i] = pg1[i];
    for (j = 0; j < 2; j++) pg2[PB + i][j] = pg2[i][j];
    for (j = 0; j < 3; j++) pg3[PB + i][j] = pg3[i][j];
  }
}

/* perlin.c:76-124 with perlin.h's setup/s_curve/lerp/at3 macros written out */
double orc_noise3(const double vec[3]) {
  if (pstart) {
    pstart = 0;
    orc_perlin_init();
  }
  int b0[3], b1[3];
  double r0[3], r1[3];
  for (int a = 0; a < 3; ++a) {
    double t = vec[a] + PN;
    b0[a] = ((int)t) & PBM;
    b1[a] = (b0[a] + 1) & PBM;
    r0[a] = t - (int)t;
    r1[a] = r0[a] - 1.;
  }
  int i = pp[b0[0]], j = pp[b1[0]];
  int b00 = pp[i + b0[1]], b10 = pp[j + b0[1]], b01 = pp[i + b1[1]], b11 = pp[j + b1[1]];
  double t = r0[0] * r0[0] * (3. - 2. * r0[0]);
  double sy = r0[1] * r0[1] * (3. - 2. * r0[1]);
  double sz = r0[2] * r0[2] * (3. - 2. * r0[2]);
  double *q, u, v, a, b, c, d;
#define AT3(rx, ry, rz) (rx * q[0] + ry * q[1] + rz * q[2])
#define LERP(t, a, b) (a + t * (b - a))
  q = pg3[b00 + b0[2]]; u = AT3(r0[0], r0[1], r0[2]);
  q = pg3[b10 + b0[2]]; v = AT3(r1[0], r0[1], r0[2]);
  a = LERP(t, u, v);
  q = pg3[b01 + b0[2]]; u = AT3(r0[0], r1[1], r0[2]);
  q = pg3[b11 + b0[2]]; v = AT3(r1[0], r1[1], r0[2]);
  b = LERP(t, u, v);
  c = LERP(sy, a, b);
  q = pg3[b00 + b1[2]]; u = AT3(r0[0], r0[1], r1[2]);
  q = pg3[b10 + b1[2]]; v = AT3(r1[0], r0[1], r1[2]);
  a = LERP(t, u, v);
  q = pg3[b01 + b1[2]]; u = AT3(r0[0], r1[1], r1[2]);
  q = pg3[b11 + b1[2]]; v = AT3(r1[0], r1[1], r1[2]);
  b = LERP(t, u, v);
  d = LERP(sy, a, b);
  return LERP(sz, c, d);
#undef AT3
#undef LERP
}

/* perlin.c:220-240 */
double orc_perlin3d(double x, double y, double z, double alpha, double beta, int n) {
  double val, sum = 0, p[3] = {x, y, z}, scale = 1;
  for (int i = 0; i < n; i++) {
    val = orc_noise3(p);
    sum += val / scale;
    scale *= alpha;
    p[0] *= beta;
    p[1] *= beta;
    p[2] *= beta;
  }
  return sum;
}

/* perlin.c:246-263 */
double orc_perlin3d_abs(double x, double y, double z, double alpha, double beta, int n) {
  double val, sum = 0, p[3] = {x, y, z}, scale = 1;
  for (int i = 0; i < n; i++) {
    val = orc_noise3(p);
    val = val < 0 ? -val : val;
    sum += val / scale;
    scale *= alpha;
    p[0] *= beta;
    p[1] *= beta;
    p[2] *= beta;
  }
  return sum;
}

/* ------------------------------------------------------------ genvol */

/* genvol/main.cpp:153-165 perl(), ntype 0 signed / 1 unsigned / 2 inverse-unsigned */
static double gv_perl(double x, double y, double z, int ntype, int pharm, double pscale,
                      const float pwrap[3], float palpha, float pbeta) {
  switch (ntype) {
    case 0: return orc_perlin3d(x * pwrap[0], y * pwrap[1], z * pwrap[2], palpha, pbeta, pharm) * pscale;
    case 1: return orc_perlin3d_abs(x * pwrap[0], y * pwrap[1], z * pwrap[2], palpha, pbeta, pharm) * pscale;
    case 2: return 1.0 - orc_perlin3d_abs(x * pwrap[0], y * pwrap[1], z * pwrap[2], palpha, pbeta, pharm) * pscale;
  }
  return 0;
}

/* genvol/main.cpp:212-256 */
void orc_genvol_spheres(unsigned char *d, int sx, int sy, int sz, int nspheres, int use_perl,
                        int ntype, int pharm, double pscale, const float pwrap[3],
                        float palpha, float pbeta) {
  float dd = 255 / (float)nspheres;
  float c[3] = {.5f, .5f, .5f};
  float dx = 1 / (float)sx, dy = 1 / (float)sy, dz = 1 / (float)sz;
  for (int i = 0; i < sz; ++i)
    for (int j = 0; j < sy; ++j)
      for (int k = 0; k < sx; ++k) {
        float p[3] = {k * dx, j * dy, i * dz};
        float v[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
        float nv = (float)sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); /* normV3 */
        float r = (float)MINF(nv, .48);
        if (use_perl) {
          r += (float)gv_perl(p[0], p[1], p[2], ntype, pharm, pscale, pwrap, palpha, pbeta);
          r = (float)clamp_arb(0, r, .5);
        }
        int val = (int)(r * 2 * nspheres);
        d[((size_t)i * sy + j) * sx + k] = uc_cast((nspheres - val) * dd);
      }
}

/* genvol/main.cpp:306-332 */
void orc_genvol_perl(unsigned char *d, int sx, int sy, int sz, int param, int pharm,
                     const float pwrap[3], float palpha, float pbeta) {
  float dd = 255 / (float)param;
  float dx = 1 / (float)sx, dy = 1 / (float)sy, dz = 1 / (float)sz;
  for (int i = 0; i < sz; ++i)
    for (int j = 0; j < sy; ++j)
      for (int k = 0; k < sx; ++k) {
        float p[3] = {k * dx, j * dy, i * dz};
        double pn = orc_perlin3d(p[0] * pwrap[0], p[1] * pwrap[1], p[2] * pwrap[2], palpha, pbeta, pharm);
        float r = (float)(pn < 0 ? -pn : pn);
        int val = (int)(r * param);
        d[((size_t)i * sy + j) * sx + k] = uc_cast((param - val) * dd);
      }
}

/* genvol/main.cpp:334-430: 27-tap scatter blur, weights bw0, bw1/6, bw2/12, bw3/8, divisor
 * bw0+bw1+bw2+bw3; scatter order preserved so float sums round identically */
void orc_genvol_blur(unsigned char *dataV, int sx, int sy, int sz, const float bw[4]) {
  size_t n = (size_t)sx * sy * sz;
  float *tmp = (float *)calloc(n, sizeof(float));
  const int sxy = sx * sy;
  for (int i = 1; i < sz - 1; ++i)
    for (int j = 1; j < sy - 1; ++j)
      for (int k = 1; k < sx - 1; ++k) {
        size_t index = (size_t)i * sxy + (size_t)j * sx + k;
        float v0 = (float)(dataV[index] / 255.0 * bw[0]);
        float v1 = (float)(dataV[index] / 255.0 * bw[1] / 6.0);
        float v2 = (float)(dataV[index] / 255.0 * bw[2] / 12.0);
        float v3 = (float)(dataV[index] / 255.0 * bw[3] / 8.0);
#define TMP(di, dj, dk) tmp[(size_t)(i + (di)) * sxy + (size_t)(j + (dj)) * sx + (k + (dk))]
        TMP(0, 0, 0) += v0;
        TMP(1, 0, 0) += v1;
        TMP(1, 1, 0) += v2;
        TMP(1, 1, 1) += v3;
        TMP(1, 1, -1) += v3;
        TMP(1, -1, 0) += v2;
        TMP(1, -1, 1) += v3;
        TMP(1, -1, -1) += v3;
        TMP(1, 0, 1) += v2;
        TMP(1, 0, -1) += v2;
        TMP(-1, 0, 0) += v1;
        TMP(-1, 1, 0) += v2;
        TMP(-1, 1, 1) += v3;
        TMP(-1, 1, -1) += v3;
        TMP(-1, -1, 0) += v2;
        TMP(-1, -1, 1) += v3;
        TMP(-1, -1, -1) += v3;
        TMP(-1, 0, 1) += v2;
        TMP(-1, 0, -1) += v2;
        TMP(0, 1, 0) += v1;
        TMP(0, 1, 1) += v2;
        TMP(0, 1, -1) += v2;
        TMP(0, -1, 0) += v1;
        TMP(0, -1, 1) += v2;
        TMP(0, -1, -1) += v2;
        TMP(0, 0, 1) += v1;
        TMP(0, 0, -1) += v1;
#undef TMP
      }
  float div = bw[0] + bw[1] + bw[2] + bw[3];
  for (size_t q = 0; q < n; ++q) dataV[q] = uc_cast(clamp01(tmp[q] / div) * 255);
  free(tmp);
}

/* ------------------------------------------------------------ genVGH makeVGH<T> */

void orc_make_vgh(const void *in, int in_dtype, int sx, int sy, int sz, int compat,
                  unsigned char *out_u8, float *out_f32) {
  size_t n = (size_t)sx * sy * sz;
  float *gradV3 = (float *)malloc(n * 3 * sizeof(float));
  float *gmag = (float *)malloc(n * sizeof(float));
  float *hess = (float *)malloc(n * sizeof(float));
  float gmmax = -100000000, gmmin = 100000000, dmax = -100000000, dmin = 100000000;
  const unsigned char *d8 = (const unsigned char *)in;
  const float *df = (const float *)in;
#define D(I, J, K) (in_dtype == 0 ? (float)d8[((size_t)(I)*sy + (J)) * sx + (K)] : df[((size_t)(I)*sy + (J)) * sx + (K)])
#define BORDER(I, J, K) (((K) < 1) || ((K) > sx - 2) || ((J) < 1) || ((J) > sy - 2) || ((I) < 1) || ((I) > sz - 2))
  /* 1st derivative, un-normalised central differences, 0 on the border (:75-101) */
  for (int i = 0; i < sz; ++i)
    for (int j = 0; j < sy; ++j)
      for (int k = 0; k < sx; ++k) {
        size_t o = ((size_t)i * sy + j) * sx + k;
        if (BORDER(i, j, k)) {
          gradV3[o * 3 + 0] = gradV3[o * 3 + 1] = gradV3[o * 3 + 2] = 0;
          gmag[o] = 0;
        } else {
          float dx = (float)(D(i, j, k + 1) - D(i, j, k - 1));
          float dy = (float)(D(i, j + 1, k) - D(i, j - 1, k));
          float dz = (float)(D(i + 1, j, k) - D(i - 1, j, k));
          gradV3[o * 3 + 0] = dx;
          gradV3[o * 3 + 1] = dy;
          gradV3[o * 3 + 2] = dz;
          gmag[o] = (float)sqrt(dx * dx + dy * dy + dz * dz);
          gmmax = MAXF(gmag[o], gmmax);
          gmmin = MINF(gmag[o], gmmin);
          dmax = MAXF(D(i, j, k), dmax);
          dmin = MINF(D(i, j, k), dmin);
        }
      }
  float hmax = -100000000, hmin = 100000000;
  /* 2nd derivative along the gradient (:108-152) */
  for (int i = 0; i < sz; ++i)
    for (int j = 0; j < sy; ++j)
      for (int k = 0; k < sx; ++k) {
        size_t o = ((size_t)i * sy + j) * sx + k;
        if (BORDER(i, j, k)) {
          hess[o] = 0;
          continue;
        }
#define GR(I, J, K, E) gradV3[(((size_t)(I)*sy + (J)) * sx + (K)) * 3 + (E)]
        float h[9];
        h[0] = GR(i, j, k + 1, 0) - GR(i, j, k - 1, 0);
        h[1] = GR(i, j + 1, k, 0) - GR(i, j - 1, k, 0);
        h[2] = GR(i + 1, j, k, 0) - GR(i - 1, j, k, 0);
        h[3] = GR(i, j, k + 1, 1) - GR(i, j, k - 1, 1);
        h[4] = GR(i, j + 1, k, 1) - GR(i, j - 1, k, 1);
        h[5] = GR(i + 1, j, k, 1) - GR(i - 1, j, k, 1);
        h[6] = GR(i, j, k + 1, 2) - GR(i, j, k - 1, 2);
        h[7] = GR(i, j + 1, k, 2) - GR(i, j - 1, k, 2);
        h[8] = GR(i + 1, j, k, 2) - GR(i - 1, j, k, 2);
        float tg[3] = {GR(i, j, k, 0) / gmag[o], GR(i, j, k, 1) / gmag[o], GR(i, j, k, 2) / gmag[o]};
        float tv[3];
        tv[0] = tg[0] * h[0] + tg[1] * h[1] + tg[2] * h[2];
        /* reference typo: tg[1] is not multiplied by h[4] (:135-137) */
        tv[1] = compat ? tg[0] * h[3] + tg[1] + tg[2] * h[5]
                       : tg[0] * h[3] + tg[1] * h[4] + tg[2] * h[5];
        tv[2] = tg[0] * h[6] + tg[1] * h[7] + tg[2] * h[8];
        hess[o] = tg[0] * tv[0] + tg[1] * tv[1] + tg[2] * tv[2];
        hmax = MAXF(hess[o], hmax); /* NaN (zero gradient) never wins the macro compare */
        hmin = MINF(hess[o], hmin);
#undef GR
      }
  /* quantise (:155-179): V,G min/max -> 0..255; H<0 -> [0,85], H>=0 -> [85,170] */
  for (int i = 0; i < sz; ++i)
    for (int j = 0; j < sy; ++j)
      for (int k = 0; k < sx; ++k) {
        size_t o = ((size_t)i * sy + j) * sx + k;
        unsigned char q[3] = {0, 0, 0};
        float f3[3] = {0, 0, 0};
        if (!BORDER(i, j, k)) {
          double vq = affine(dmin, D(i, j, k), dmax, 0, 255);
          double gq = affine(gmmin, gmag[o], gmmax, 0, 255);
          double hq;
          if (hess[o] < 0) {
            float th = (float)affine(hmin, hess[o], 0, 0, 1);
            hq = affine(0, th, 1, 0, 255 / 3);
          } else {
            float th = (float)affine(0, hess[o], hmax, 0, 1);
            hq = affine(0, th, 1, 255 / 3, 255 / 3 * 2);
          }
          q[0] = uc_cast(vq);
          q[1] = uc_cast(gq);
          q[2] = uc_cast(hq); /* NaN -> 0, as the x86 cast does */
          f3[0] = (float)(vq / 255.0);
          f3[1] = (float)(gq / 255.0);
          f3[2] = (hq == hq) ? (float)(hq / 255.0) : 0.0f;
        }
        if (out_u8) memcpy(out_u8 + o * 3, q, 3);
        if (out_f32) memcpy(out_f32 + o * 3, f3, sizeof f3);
      }
#undef D
#undef BORDER
  free(gradV3);
  free(gmag);
  free(hess);
}

/* ------------------------------------------------------------ normals */

/* VectorMath.h:1217-1281 blurV3D: 27-tap scatter of weights (w0,w1,w2,w3) / (w0+6w1+12w2+8w3) */
static void blur_v3d(float *g, float w0, float w1, float w2, float w3, int sx, int sy, int sz) {
  size_t n = (size_t)sx * sy * sz * 3;
  float *tmp = (float *)calloc(n, sizeof(float));
  const size_t stx = 3, sty = (size_t)sx * 3, stz = (size_t)sy * sx * 3;
  static const int order[27][4] = {
      /* {di, dj, dk, weight-index}, exactly the csaddV3 sequence of :1235-1265 */
      {0, 0, 0, 0},  {1, 0, 0, 1},   {1, 1, 0, 2},   {1, 1, 1, 3},   {1, 1, -1, 3},  {1, -1, 0, 2},
      {1, -1, 1, 3}, {1, -1, -1, 3}, {1, 0, 1, 2},   {1, 0, -1, 2},  {-1, 0, 0, 1},  {-1, 1, 0, 2},
      {-1, 1, 1, 3}, {-1, 1, -1, 3}, {-1, -1, 0, 2}, {-1, -1, 1, 3}, {-1, -1, -1, 3}, {-1, 0, 1, 2},
      {-1, 0, -1, 2}, {0, 1, 0, 1},  {0, 1, 1, 2},   {0, 1, -1, 2},  {0, -1, 0, 1},  {0, -1, 1, 2},
      {0, -1, -1, 2}, {0, 0, 1, 1},  {0, 0, -1, 1}};
  const float w[4] = {w0, w1, w2, w3};
  for (int i = 1; i < sz - 1; ++i)
    for (int j = 1; j < sy - 1; ++j)
      for (int k = 1; k < sx - 1; ++k) {
        const float *src = g + i * stz + j * sty + k * stx;
        for (int t = 0; t < 27; ++t) {
          float *dst = tmp + (i + order[t][0]) * stz + (j + order[t][1]) * sty + (k + order[t][2]) * stx;
          float ww = w[order[t][3]];
          dst[0] += ww * src[0]; /* csaddV3: out += s*in */
          dst[1] += ww * src[1];
          dst[2] += ww * src[2];
        }
      }
  float div = w0 + 6 * w1 + 12 * w2 + 8 * w3;
  for (size_t q = 0; q < n; ++q) g[q] = tmp[q] / div;
  free(tmp);
}

/* scalebiasN (VectorMath.h:1133-1148): normalise (skip zero length, :359-367) then
 * (uchar)(n*128+128).  Deviation: n=+1 would wrap 256 -> 0; clamp to 255. */
static void scalebias_n(unsigned char *out, float *g, size_t nvox) {
  for (size_t q = 0; q < nvox; ++q) {
    float *v = g + q * 3;
    float len = (float)sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (len > 0) {
      v[0] /= len;
      v[1] /= len;
      v[2] /= len;
    }
    for (int e = 0; e < 3; ++e) {
      float s = v[e] * 128 + 128;
      out[q * 3 + e] = s >= 255.0f ? 255 : uc_cast(s);
    }
  }
}

void orc_normals_vgh(const unsigned char *dat, int nelts, int sx, int sy, int sz, int blur,
                     unsigned char *out) {
  size_t n = (size_t)sx * sy * sz;
  float *g = (float *)malloc(n * 3 * sizeof(float));
  /* derivative3DVGH (VectorMath.h:874-899): int central differences of channel 0 */
  for (int i = 0; i < sz; ++i)
    for (int j = 0; j < sy; ++j)
      for (int k = 0; k < sx; ++k) {
        size_t o = ((size_t)i * sy + j) * sx + k;
        if ((k < 1) || (k > sx - 2) || (j < 1) || (j > sy - 2) || (i < 1) || (i > sz - 2)) {
          g[o * 3] = g[o * 3 + 1] = g[o * 3 + 2] = 0;
        } else {
#define DT(I, J, K) ((int)dat[(((size_t)(I)*sy + (J)) * sx + (K)) * nelts])
          g[o * 3 + 0] = (float)(DT(i, j, k + 1) - DT(i, j, k - 1));
          g[o * 3 + 1] = (float)(DT(i, j + 1, k) - DT(i, j - 1, k));
          g[o * 3 + 2] = (float)(DT(i + 1, j, k) - DT(i - 1, j, k));
#undef DT
        }
      }
  if (blur) blur_v3d(g, 1.0f, .3f, .2f, .1f, sx, sy, sz); /* MetaVolume.cpp:1316 */
  scalebias_n(out, g, n);
  free(g);
}

void orc_merge_addg(const unsigned char *in, int nf, int sx, int sy, int sz, unsigned char *out,
                    unsigned char *grad_out) {
  size_t n = (size_t)sx * sy * sz;
  int ne = nf + 1;
  float *g = (float *)malloc(n * 3 * sizeof(float));
  float *mag = (float *)malloc(n * sizeof(float));
  /* AGradArb over the first ne-1 channels: SUM of per-field central differences */
  for (int i = 0; i < sz; ++i)
    for (int j = 0; j < sy; ++j)
      for (int k = 0; k < sx; ++k) {
        size_t o = ((size_t)i * sy + j) * sx + k;
        g[o * 3] = g[o * 3 + 1] = g[o * 3 + 2] = 0;
        for (int e = 0; e < nf; ++e) out[o * ne + e] = in[o * nf + e];
        if ((k < 1) || (k > sx - 2) || (j < 1) || (j > sy - 2) || (i < 1) || (i > sz - 2)) continue;
#define DI(I, J, K, E) ((float)in[(((size_t)(I)*sy + (J)) * sx + (K)) * nf + (E)])
        for (int e = 0; e < nf; ++e) {
          g[o * 3 + 0] += DI(i, j, k + 1, e) - DI(i, j, k - 1, e);
          g[o * 3 + 1] += DI(i, j + 1, k, e) - DI(i, j - 1, k, e);
          g[o * 3 + 2] += DI(i + 1, j, k, e) - DI(i - 1, j, k, e);
        }
#undef DI
      }
  /* GMag (VectorMath.h:1010-1030): |g|/max*255 truncated */
  float maxm = 0;
  for (size_t o = 0; o < n; ++o) {
    mag[o] = (float)sqrt(g[o * 3] * g[o * 3] + g[o * 3 + 1] * g[o * 3 + 1] + g[o * 3 + 2] * g[o * 3 + 2]);
    maxm = MAXF(mag[o], maxm);
  }
  for (size_t o = 0; o < n; ++o) out[o * ne + nf] = uc_cast(mag[o] / maxm * 255.0);
  if (grad_out) scalebias_n(grad_out, g, n);
  free(g);
  free(mag);
}

/* MetaVolume::brick(int) split order z, y, x (MetaVolume.cpp:1379-1390) */
void orc_brick_grid(int sx, int sy, int sz, int maxsz, int dims[3]) {
  int xd = 1, yd = 1, zd = 1;
  if ((long long)sx * sy * sz > maxsz) {
    while (((sx / (float)xd) * (sy / (float)yd) * (sz / (float)zd)) > maxsz) {
      if (zd > yd) {
        if (yd > xd) xd *= 2;
        else yd *= 2;
      } else
        zd *= 2;
    }
  }
  dims[0] = xd;
  dims[1] = yd;
  dims[2] = zd;
}

/* ------------------------------------------------------------ TLUT (TLUT.cpp) */

void orc_tlut_default(float *rgba, int size) {
  /* TLUT::TLUT (:26-36): grey ramp n/(size-1), alpha 1/size */
  for (int n = 0; n < size; ++n) {
    rgba[n * 4] = rgba[n * 4 + 1] = rgba[n * 4 + 2] = n / (float)(size - 1.0);
    rgba[n * 4 + 3] = (float)(1.0 / size);
  }
}

static void seg(float *rgba, int min, int max, const float c1[3], const float c2[3]) {
  for (int n = min; n < max; ++n) {
    float t = (float)((n - min) / (max - min - 1.0));
    for (int e = 0; e < 3; ++e) rgba[n * 4 + e] = (1 - t) * c1[e] + t * c2[e];
  }
}

void orc_tlut_spectral(float *rgba, int size) {
  /* TLUT::rgbSpectral (:201-298): 7 linear segments of 1,3,3,3,1,2,3 sixteenths */
  static const float key[8][3] = {{238 / 255.0f, 138 / 255.0f, 238 / 255.0f},
                                  {25 / 255.0f, 25 / 255.0f, 112 / 255.0f},
                                  {0, 0, 1},
                                  {0, 1, 0},
                                  {173 / 255.0f, 252 / 255.0f, 0},
                                  {1, 1, 0},
                                  {1, 165 / 255.0f, 0},
                                  {1, 0, 0}};
  static const int w16[7] = {1, 3, 3, 3, 1, 2, 3};
  int min = 0, max = size / 16;
  for (int s = 0; s < 7; ++s) {
    if (s > 0) {
      min = max;
      max += w16[s] * size / 16;
    }
    seg(rgba, min, max, key[s], key[s + 1]);
  }
}

void orc_tlut_cyanmagenta(float *rgba, int size) {
  for (int n = 0; n < size; ++n) {
    float t = n / (float)(size - 1.0);
    rgba[n * 4 + 0] = (1 - t) * 1.0f + t * 0.0f;
    rgba[n * 4 + 1] = (1 - t) * 0.0f + t * 1.0f;
    rgba[n * 4 + 2] = (1 - t) * 1.0f + t * 1.0f;
  }
}

void orc_tlut_blackbody(float *rgba, int size) {
  /* TLUT::rgbBlackBody (:454-472): the (unsigned char) cast of a [0,1] ramp is reproduced
   *  -- every ramp entry comes out 0 or 1 */
  int i;
  for (i = 0; i < size / 3; ++i) {
    rgba[i * 4 + 0] = uc_cast(affine(0, i, size / 3, 0, 1));
    rgba[i * 4 + 1] = 0;
    rgba[i * 4 + 2] = uc_cast(affine(0, i, size / 3, .4, 0));
  }
  for (i = size / 3; i < 2 * size / 3; ++i) {
    rgba[i * 4 + 0] = 1;
    rgba[i * 4 + 1] = uc_cast(affine(size / 3, i, 2 * size / 3, 0, 1));
    rgba[i * 4 + 2] = 0;
  }
  for (i = 2 * size / 3; i < size; ++i) {
    rgba[i * 4 + 0] = 1;
    rgba[i * 4 + 1] = 1;
    rgba[i * 4 + 2] = uc_cast(affine(2 * size / 3, i, size, 0, 1.05));
  }
}

void orc_tlut_channel_ramp(float *rgba, int ch, int i0, int i1, float v0, float v1) {
  float denom = (float)(i1 - i0), range = v1 - v0;
  for (int n = i0; n <= i1; ++n) rgba[n * 4 + ch] = v0 + range * (n - i0) / denom;
}

void orc_tlut_scale_alpha(float *rgba, int size, float last_rate, float rate) {
  /* TLUT::scaleAlpha (:138-154): a <- 1-(1-a)^(lastSR/SR), pow in double like libm's */
  if (last_rate == rate) return;
  float alphaScale = last_rate / rate;
  for (int i = 0; i < size; ++i) rgba[i * 4 + 3] = (float)(1 - pow((1 - rgba[i * 4 + 3]), alphaScale));
}

void orc_tlut_premultiply(const float *rgba, int size, float *table) {
  for (int n = 0; n < size; ++n) {
    table[n * 4 + 0] = rgba[n * 4 + 0] * rgba[n * 4 + 3];
    table[n * 4 + 1] = rgba[n * 4 + 1] * rgba[n * 4 + 3];
    table[n * 4 + 2] = rgba[n * 4 + 2] * rgba[n * 4 + 3];
    table[n * 4 + 3] = rgba[n * 4 + 3];
  }
}

/* ------------------------------------------------------------ 2-D TF tables */

void orc_deptex_default(unsigned char *deptex, unsigned char *deptex2, int dsx, int dsy) {
  for (int j = 0; j < dsy; ++j)
    for (int k = 0; k < dsx; ++k) {
      size_t o = ((size_t)j * dsx + k) * 4;
      if (deptex) { /* NV20VolRen3D.cpp:1479-1486 */
        deptex[o + 0] = uc_cast(k / (float)dsx * 255);
        deptex[o + 1] = uc_cast(j / (float)dsy * 255);
        deptex[o + 2] = uc_cast(255 - j / (float)dsy * 255);
        deptex[o + 3] = (unsigned char)(j / (float)dsy * 255 / (float)2);
      }
      if (deptex2) { /* :1523-1530 */
        deptex2[o + 0] = uc_cast(k / (float)dsx * 255);
        deptex2[o + 1] = uc_cast(j / (float)dsy * 255);
        deptex2[o + 2] = uc_cast(255 - j / (float)dsy * 255);
        deptex2[o + 3] = 255;
      }
    }
}

void orc_copy_scale(const unsigned char *in, unsigned char *out, int dsx, int dsy, float sr) {
  /* NV20VolRen3D::copyScale (:1645-1660) */
  float alphaScale = (float)(1.0 / sr);
  for (int i = 0; i < dsy; ++i)
    for (int j = 0; j < dsx; ++j) {
      size_t o = ((size_t)i * dsx + j) * 4;
      out[o + 0] = in[o + 0];
      out[o + 1] = in[o + 1];
      out[o + 2] = in[o + 2];
      out[o + 3] = uc_cast((1.0 - pow((1.0 - (in[o + 3] / 255.0)), alphaScale)) * 255);
    }
}

/* ------------------------------------------------------------ LevWidget */

void orc_hsl_color(float H, float S, float L, float col[3]) {
  /* HSLPicker::getColor (HSLPicker.cpp:33-68), Foley/van Dam sextant form */
  float m1, m2, fract, mid1, mid2;
  int sextant;
  if (S == 0) {
    col[0] = col[1] = col[2] = L;
    return;
  }
  if (L <= 0.5f) m2 = L * (1 + S);
  else m2 = L + S - L * S;
  m1 = 2 * L - m2;
  if (H == 1) H = 0;
  H *= 6;
  sextant = (int)floor(H);
  fract = H - sextant;
  mid1 = m1 + fract * (m2 - m1);
  mid2 = m2 + fract * (m1 - m2);
  switch (sextant) {
    case 0: col[0] = m2; col[1] = mid1; col[2] = m1; break;
    case 1: col[0] = mid2; col[1] = m2; col[2] = m1; break;
    case 2: col[0] = m1; col[1] = m2; col[2] = mid1; break;
    case 3: col[0] = m1; col[1] = mid2; col[2] = m2; break;
    case 4: col[0] = mid1; col[1] = m1; col[2] = m2; break;
    default: col[0] = m2; col[1] = m1; col[2] = mid2; break;
  }
}

void orc_lev_setpos(orc_levwidget *w, const float b[2], const float l[2], const float r[2],
                    float tw, float th) {
  /* LevWidget::setPos (LevWidget.cpp:1098-1125); -10 = "default" */
#define C01(x) ((x) > 0 ? ((x) < 1 ? (x) : 1) : 0)
  w->verts[0][0] = C01(b[0]);
  w->verts[0][1] = C01(b[1]);
  w->verts[1][0] = C01(l[0]);
  w->verts[1][1] = C01(l[1]);
  w->verts[2][0] = C01(r[0]);
  w->verts[2][1] = C01(r[1]);
#undef C01
  if (th == -10) w->thresh[1] = w->verts[0][1] + (w->verts[1][1] - w->verts[0][1]) / 2;
  else w->thresh[1] = th > w->verts[0][1] ? (th < w->verts[1][1] ? th : w->verts[1][1]) : w->verts[0][1];
  if (tw == -10) w->thresh[0] = w->verts[1][0] + (w->verts[2][0] - w->verts[1][0]) / 2;
  else w->thresh[0] = tw < w->verts[2][0] ? (tw > w->verts[1][0] ? tw : w->verts[1][0]) : w->verts[2][0];
}

/* colour blend + alpha write shared by every shape (e.g. LevWidget.cpp:737-759, 788-806) */
static inline void lev_pixel_col(const orc_levwidget *w, const float *color, int use_faux, unsigned char *tex, size_t offset,
                                 float tmpa, float cs, float alphaScale, int max_rule) {
  float cw = (use_faux && w->faux) ? tmpa * cs : tmpa;
  float tmpta = tex[offset + 3] / 255.0f;
  for (int e = 0; e < 3; ++e)
    tex[offset + e] = uc_cast((tmpta * tex[offset + e] / 255.0 + cw * color[e]) / (tmpta + tmpa) * 255);
  if (max_rule) tex[offset + 3] = uc_cast(MAXF(tmpa * 255, tex[offset + 3]) * alphaScale);
  else tex[offset + 3] = uc_cast((tmpa * 255 + (1.0 - tmpa) * tex[offset + 3]) * alphaScale);
}
static inline void lev_pixel(const orc_levwidget *w, unsigned char *tex, size_t offset, float tmpa,
                             float cs, float alphaScale, int max_rule) {
  lev_pixel_col(w, w->color, 1, tex, offset, tmpa, cs, alphaScale, max_rule);
}

void orc_lev_rasterize(const orc_levwidget *w, unsigned char *tex, int sv, int sg, int sh) {
  const float(*verts)[2] = w->verts;
  int H = (int)(verts[1][1] * sg) - 1;
  int sth = sg * sv * 4, stg = sv * 4, stv = 4;
  int base = (int)(w->thresh[1] * sg);
  if (w->type == 0) {
    /* triangle (:704-761) */
    for (int k = 0; k < sh; ++k) {
      float alphaScale = (k != 1) ? w->be : 1;
      for (int i = base; i < H + 1; ++i) {
        int start = (int)((verts[0][0] + (i / (float)sg) * (verts[1][0] - verts[0][0]) / verts[1][1]) * sv);
        int fin = (int)((verts[0][0] + (i / (float)sg) * (verts[2][0] - verts[0][0]) / verts[1][1]) * sv) + 1;
        fin -= start;
        for (int j = 0; j < fin; ++j) {
          size_t offset = (size_t)k * sth + (size_t)i * stg + (size_t)(start + j) * stv;
          float tmpa = (float)affine(-1, j, fin, -1, 1);
          tmpa = tmpa < 0 ? (1.0f + tmpa) : (1.0f - tmpa);
          float cs = tmpa;
          tmpa *= w->alpha;
          lev_pixel(w, tex, offset, tmpa, cs, alphaScale, 1);
        }
      }
    }
  } else if (w->type == 2) {
    /* 1-D style (:903-1019): a plateau of full alpha with linear ramps either side, every scan
     * line of the widget's height the same; no boundary-emphasis scale on this shape */
    int hc = (int)((w->thresh[0] - verts[1][0]) * sv);
    float vthresh = (w->thresh[1] - verts[0][1]) / (verts[1][1] - verts[0][1]);
    for (int k = 0; k < sh; ++k)
      for (int i = (int)(verts[0][1] * sg); i < H + 1; ++i) {
        int start = (int)(verts[1][0] * sv);
        int dist = (int)(verts[2][0] * sv - start);
        int hc0 = (int)(hc * (1.0 - vthresh) + 1);
        int hc1 = (int)(dist - (dist - hc) * (1.0 - vthresh) + 1);
        int j = 0;
        for (; j < hc0; ++j) {
          float tmpa = (float)affine(0, j, hc0, 0, 1);
          float cs = tmpa;
          tmpa *= w->alpha;
          lev_pixel(w, tex, (size_t)k * sth + (size_t)i * stg + (size_t)(start + j) * stv, tmpa, cs, 1.0f, 0);
        }
        for (j = hc0; j < hc1; ++j) {
          float tmpa = 1, cs = 1;
          tmpa *= w->alpha;
          lev_pixel(w, tex, (size_t)k * sth + (size_t)i * stg + (size_t)(start + j) * stv, tmpa, cs, 1.0f, 0);
        }
        for (j = hc1; j < dist; ++j) {
          float tmpa = (float)affine(hc1, j, dist, 1, 0);
          float cs = tmpa;
          tmpa *= w->alpha;
          lev_pixel(w, tex, (size_t)k * sth + (size_t)i * stg + (size_t)(start + j) * stv, tmpa, cs, 1.0f, 0);
        }
      }
  } else if (w->type == 3) {
    /* default style (:1022-1072): alpha rises along g as x/(m+x) (note the literal 255.0 where the
     * other shapes use sg), colour walks once around the hue circle across the widget's width
     * (HSLPicker::reset(0,1,.5) then updateHL(dc,0) per pixel, HSLPicker.cpp:44-48) */
    float m = (w->thresh[1] - verts[0][1]) / (verts[1][1] - verts[0][1]);
    for (int k = 0; k < sh; ++k) {
      float alphaScale = (k != 1) ? w->be : 1;
      int start = (int)(verts[1][0] * sv);
      int fin = (int)(verts[2][0] * sv);
      fin -= start;
      float dc = 1 / ((verts[1][0] - verts[2][0]) * sv - 1);
      for (int i = (int)(verts[0][1] * sg); i < H + 1; ++i) {
        float tmpa = (float)(((i / 255.0) - verts[0][1]) / (m + (i / 255.0) - verts[0][1]));
        tmpa *= w->alpha;
        tmpa = tmpa > 1 ? 1 : (tmpa < 0 ? 0 : tmpa);
        float hue = 0, lev = .5f;
        for (int j = 0; j < fin; ++j) {
          float cl[3];
          hue = (float)((hue + dc) > 1.0 ? (hue + dc - 1.0) : ((hue + dc) < 0 ? (hue + dc + 1.0) : (hue + dc)));
          lev = (lev + 0) > 1 ? 1 : (lev + 0) < 0 ? 0 : (lev + 0);
          orc_hsl_color(hue, 1, lev, cl);
          lev_pixel_col(w, cl, 0, tex, (size_t)k * sth + (size_t)i * stg + (size_t)(start + j) * stv, tmpa, tmpa, alphaScale, 0);
        }
      }
    }
  } else {
    /* ellipse inside the "square" widget (:764-900); the four quadrant loops of the
     * reference run the same body over j in [0,W) and i in [verts[0][1]*sg, H] */
    int W = (int)((verts[2][0] - verts[1][0]) * sv);
    int h = (int)((verts[0][1] - verts[1][1]) * sg);
    int hc = (int)((w->thresh[0] - verts[1][0]) * sv);
    int VC = (int)((w->thresh[1]) * sg);
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
    for (int k = 0; k < sh; ++k) {
      float alphaScale = (k != 1) ? w->be : 1;
      for (int i = (int)(verts[0][1] * sg); i < H + 1; ++i)
        for (int j = 0; j < W; ++j) {
          size_t offset = (size_t)k * sth + (size_t)i * stg + (size_t)((int)(verts[1][0] * sv) + j) * stv;
          float d = ((i - VC) * (i - VC) * scaleh) + ((j - hc) * (j - hc) * scalew);
          float tmpa = (float)affine(0, d, maxd, 1, 0);
          float cs = tmpa;
          tmpa = tmpa > 0 ? (tmpa < 1 ? tmpa * tmpa * w->alpha : w->alpha) : 0;
          lev_pixel(w, tex, offset, tmpa, cs, alphaScale, 0);
        }
    }
  }
}

void orc_rasterize_vgh(unsigned char *ptex, int sx, int sy, float slider1hi) {
  /* TFWidgetRen::rasterizevgH, VGH/V1GH branch (TFWidgetRen1.cpp:1040-1062) */
  int cent = (int)(sx / 3.0);
  int sti = sx * 4, stj = 4;
  float b = 255 - 20 * cent * (1 - slider1hi);
  float fb = 255 - b;
  float m = (fb < 0 ? -fb : fb) / (float)cent;
  for (int i = 0; i < sy; ++i)
    for (int j = 0; j < cent; ++j) ptex[i * sti + j * stj + 3] = uc_cast(clamp_arb(0, j * m + b, 255));
  b = 255;
  m = -m;
  for (int i = 0; i < sy; ++i)
    for (int j = 1; j < cent + 1; ++j)
      ptex[i * sti + (j + cent) * stj + 3] = uc_cast(clamp_arb(0, j * m + b, 255));
}

void orc_noise_tex(unsigned char *out, int n) {
  /* R8kVolRen3D_cpy::createNoiseTex (:2421-2433): srand(1), 4 draws per texel */
  orc_srand(1);
  size_t cnt = (size_t)n * n * n * 4;
  for (size_t q = 0; q < cnt; ++q)
    out[q] = uc_cast(((orc_rand() / (float)ORC_RAND_MAX * .5) + .5 + 1.0 / 512) * 255);
}


/* MetaVolume::hist2D (MetaVolume.cpp:1650-1688): joint histogram of the first two voxel bytes
 * (value, gradient magnitude) over all sub-volumes, counted in FLOAT bins (so a bin stops growing
 * at 2^24), then log-scaled: h = (float)log(count), max over h starting from 0,
 * hist = (uchar)(h / max * 255).  Empty bins (log 0 = -inf) and the degenerate max = 0 case cast
 * an infinity / NaN to uchar in the reference; here both give 0.  Returns 0 when nelts < 2
 * ("this type of histogram is not implemented"). */
int orc_hist2d(const unsigned char *data, int nelts, long long nvox, unsigned char *hist) {
  if (nelts < 2 || !data) return 0;
  float *ih = (float *)calloc(65536, sizeof(float));
  for (long long i = 0; i < nvox; ++i) {
    const unsigned char *d = data + i * nelts;
    ih[d[0] + d[1] * 256] += 1;
  }
  float max = 0;
  for (int i = 0; i < 65536; ++i) {
    ih[i] = (float)log(ih[i]);
    max = ih[i] > max ? ih[i] : max;
  }
  for (int i = 0; i < 65536; ++i) {
    if (!(max > 0) || !isfinite(ih[i])) hist[i] = 0;
    else hist[i] = (unsigned char)(ih[i] / (float)max * 255);
  }
  free(ih);
  return 1;
}
