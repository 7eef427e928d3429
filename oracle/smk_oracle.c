/*
 * smk_oracle.c -- CPU reference ray-marcher (fp32) for the Simian volume-rendering hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see smk_oracle.h).  PARITY UNPINNED by the reference: it ships no
 * CPU renderer / tests / golden images; pinned by the KATs in tests/test_oracle_kat.py.
 *
 * What is restated, line by line (paths relative to /root/reference):
 *   sample placement .... VolumeRenderer::render3DVA            VolumeRenderer.cpp:507-611
 *                         (view-aligned planes, spacing dis, far corner start, S=(int)(dist/dis))
 *                         + brick-consistent global planes       R8kVolRen3D.cpp:1331-1351
 *   texel addressing .... edge-to-edge texcoords 0..1            VolumeRenderer.cpp:410-418
 *                         GL_LINEAR + clamp                      NV20VolRen3D.cpp:1379-1383
 *   1-D classification .. color table after filtering            VolumeRenderer.cpp:576-587, TLUT.cpp:65-80
 *   2-D/3rd axis ........ dependent lookups (V,G) x (H,4th)      NV20VolRen3D.cpp:544-596, 810-838
 *   3-D dense ........... ptex[h][g][v]                          TFWidgetRen.cpp:779-845
 *   Phong R8k ........... cube-map LUT + fragment shader         R8kVolRen3D.cpp:2620-2679, 2831-2977
 *   Phong NV20 .......... register combiners                     NV20VolRen3D.cpp:634-806
 *   blending ............ BTF ONE,1-SRC_ALPHA / FTB 1-DST_ALPHA,ONE   VolumeRenderer.cpp:590, R8kVolRen3D.cpp:1441-1449
 *   perturbation ........ tc += sum w*(noise(tc*s)-.5)           R8kVolRen3D_cpy.cpp:1590-1595, 3424-3505
 *
 * Arithmetic contract (DESIGN.md "sample placement"): every float expression that decides
 * WHERE a sample is (ray coefficients -> voxel coordinate -> inside test -> base index) is
 * written as an explicit fmaf chain so that any implementation reproducing the same chain in
 * IEEE fp32 gets bit-identical positions.  Build with -ffp-contract=off.
 */
#include "smk_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define INV255 (1.0f / 255.0f)

static long long g_inside_samples = 0;
long long orc_last_inside_samples(void) { return g_inside_samples; }

/* ------------------------------------------------------------ matrices (double, GL col-major) */

/* VolumeRenderer::inverseMatrix (VolumeRenderer.cpp:1096-1131): affine inverse, det in the
 * reference is a GLfloat; we keep double throughout (host-side setup, not a parity surface). */
static void affine_inverse(double inv[16], const double m[16]) {
  double det = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[1] * m[4] * m[10] +
               m[1] * m[6] * m[8] + m[2] * m[4] * m[9] - m[2] * m[5] * m[8];
  inv[0] = (m[5] * m[10] - m[6] * m[9]) / det;
  inv[1] = (-m[1] * m[10] + m[2] * m[9]) / det;
  inv[2] = (m[1] * m[6] - m[2] * m[5]) / det;
  inv[3] = 0.0;
  inv[4] = (-m[4] * m[10] + m[6] * m[8]) / det;
  inv[5] = (m[0] * m[10] - m[2] * m[8]) / det;
  inv[6] = (-m[0] * m[6] + m[2] * m[4]) / det;
  inv[7] = 0.0;
  inv[8] = (m[4] * m[9] - m[5] * m[8]) / det;
  inv[9] = (-m[0] * m[9] + m[1] * m[8]) / det;
  inv[10] = (m[0] * m[5] - m[1] * m[4]) / det;
  inv[11] = 0.0;
  inv[12] = -(inv[0] * m[12] + inv[4] * m[13] + inv[8] * m[14]);
  inv[13] = -(inv[1] * m[12] + inv[5] * m[13] + inv[9] * m[14]);
  inv[14] = -(inv[2] * m[12] + inv[6] * m[13] + inv[10] * m[14]);
  inv[15] = 1.0;
}

static void mat_mul(double out[16], const double a[16], const double b[16]) {
  double t[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[k * 4 + r] * b[c * 4 + k];
      t[c * 4 + r] = s;
    }
  memcpy(out, t, sizeof t);
}

static void mat_translate(double m[16], double x, double y, double z) {
  memset(m, 0, 16 * sizeof(double));
  m[0] = m[5] = m[10] = m[15] = 1.0;
  m[12] = x;
  m[13] = y;
  m[14] = z;
}

/* gluLookAt by its GL definition (the reference's own buildLookAt is broken) */
static void mat_lookat(double m[16], const float eye[3], const float at[3], const float up[3]) {
  double f[3] = {at[0] - eye[0], at[1] - eye[1], at[2] - eye[2]};
  double fl = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  for (int i = 0; i < 3; ++i) f[i] /= fl;
  double u[3] = {up[0], up[1], up[2]};
  double s[3] = {f[1] * u[2] - f[2] * u[1], f[2] * u[0] - f[0] * u[2], f[0] * u[1] - f[1] * u[0]};
  double sl = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
  for (int i = 0; i < 3; ++i) s[i] /= sl;
  double uu[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
  double r[16] = {s[0], uu[0], -f[0], 0, s[1], uu[1], -f[1], 0, s[2], uu[2], -f[2], 0, 0, 0, 0, 1};
  double t[16];
  mat_translate(t, -eye[0], -eye[1], -eye[2]);
  mat_mul(m, r, t);
}

void orc_modelview(const float eye[3], const float at[3], const float up[3],
                   const float trans[3], const float xform[16], const float fsize[3],
                   double mv[16]) {
  double la[16], t1[16], r[16], t2[16];
  mat_lookat(la, eye, at, up);
  mat_translate(t1, trans[0], trans[1], trans[2]);
  for (int i = 0; i < 16; ++i) r[i] = xform[i];
  mat_translate(t2, -fsize[0] / 2.0f, -fsize[1] / 2.0f, -fsize[2] / 2.0f);
  mat_mul(mv, la, t1);
  mat_mul(mv, mv, r);
  mat_mul(mv, mv, t2);
}

/* ------------------------------------------------------------ ray coefficients */

int orc_ray_setup(const orc_volume *v, const orc_camera *c, orc_raycoef *o) {
  double inv[16];
  affine_inverse(inv, c->mv);
  const double *M = c->mv;
  double f[3] = {v->fx, v->fy, v->fz};
  /* view-space z of the 8 corners of the WHOLE volume (VolumeRenderer.cpp:521-531): the
   * plane set is global so every brick samples the same planes (R8kVolRen3D.cpp:1331-1351) */
  double zmin = 1e300, zmax = -1e300;
  for (int i = 0; i < 8; ++i) {
    double x = (i & 1) ? f[0] : 0, y = (i & 2) ? f[1] : 0, z = (i & 4) ? f[2] : 0;
    double zz = M[2] * x + M[6] * y + M[10] * z + M[14];
    if (zz < zmin) zmin = zz;
    if (zz > zmax) zmax = zz;
  }
  double dist = zmax - zmin, dis;
  int S;
  if (c->steps > 0) {
    S = c->steps;
    dis = dist / S;
  } else {
    /* dis = sizef[0] / (size[0] * sampleFrequency), float (VolumeRenderer.cpp:595) */
    float disf = v->fx / ((float)v->nx * c->sample_rate);
    dis = disf;
    S = (int)(dist / dis); /* :598 */
  }
  if (S < 0) S = 0;
  double n = c->znear;
  /* planes z_k = zmin + k*dis, k = 1..S (sp starts at the far corner and is advanced before
   * the first slice, :611-616).  Front-to-back index m = S-k, m = 0..S-1. */
  double z0 = zmin + S * dis;
  double tau0 = -z0 / n, dtau = dis / n;
  double l = c->frustum[0], r = c->frustum[1], b = c->frustum[2], t = c->frustum[3];
  o->pxs = (float)((r - l) / c->width);
  o->pxl = (float)l;
  o->pys = (float)((t - b) / c->height);
  o->pyl = (float)b;
  double N[3] = {v->nx, v->ny, v->nz};
  for (int a = 0; a < 3; ++a) {
    double s = N[a] / f[a];
    double R0 = inv[0 + a], R1 = inv[4 + a], R2 = inv[8 + a], e = inv[12 + a];
    o->Ac[a] = (float)((e - tau0 * n * R2) * s - 0.5);
    o->Ax[a] = (float)(tau0 * R0 * s);
    o->Ay[a] = (float)(tau0 * R1 * s);
    o->Bc[a] = (float)(-dtau * n * R2 * s);
    o->Bx[a] = (float)(dtau * R0 * s);
    o->By[a] = (float)(dtau * R1 * s);
  }
  o->nplanes = S;
  o->tau0 = (float)tau0;
  o->dtau = (float)dtau;
  o->zmin = (float)zmin;
  o->zmax = (float)zmax;
  o->dis = (float)dis;
  return 0;
}

/* ------------------------------------------------------------ filtering helpers */

/* GL_LINEAR + clamp-to-edge along one axis: x already in texel units (u*N - 0.5) */
static inline void lin_clamp(float x, int n, int *i0, int *i1, float *f) {
  float hi = (float)(n - 1);
  float xc = fminf(fmaxf(x, 0.0f), hi);
  int i = (int)xc; /* xc >= 0 so truncation == floor */
  int imax = n >= 2 ? n - 2 : 0;
  if (i > imax) i = imax;
  *i0 = i;
  *i1 = (i + 1 < n) ? i + 1 : n - 1;
  *f = xc - (float)i;
}

/* GL_LINEAR + GL_REPEAT (noise texture, R8kVolRen3D_cpy.cpp:2463-2466) */
static inline void lin_repeat(float x, int n, int *i0, int *i1, float *f) {
  float fl = floorf(x);
  *f = x - fl;
  int i = (int)fl % n;
  if (i < 0) i += n;
  *i0 = i;
  *i1 = (i + 1) % n;
}

static inline float lerpf(float a, float b, float f) { return fmaf(f, b - a, a); }
static inline float sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

static inline float vox(const orc_volume *v, int x, int y, int z, int e) {
  size_t idx = (((size_t)z * v->ny + y) * v->nx + x) * v->nelts + e;
  return v->dtype == 0 ? (float)((const unsigned char *)v->data)[idx]
                       : ((const float *)v->data)[idx];
}

/* trilinear fetch of all channels at voxel coordinates (x,y,z) (already u*N-0.5) */
static void fetch_voxel(const orc_volume *v, float x, float y, float z, float ch[4]) {
  int x0, x1, y0, y1, z0, z1;
  float fx, fy, fz;
  lin_clamp(x, v->nx, &x0, &x1, &fx);
  lin_clamp(y, v->ny, &y0, &y1, &fy);
  lin_clamp(z, v->nz, &z0, &z1, &fz);
  for (int e = 0; e < 4; ++e) {
    if (e >= v->nelts) {
      ch[e] = 0.0f;
      continue;
    }
    float c00 = lerpf(vox(v, x0, y0, z0, e), vox(v, x1, y0, z0, e), fx);
    float c10 = lerpf(vox(v, x0, y1, z0, e), vox(v, x1, y1, z0, e), fx);
    float c01 = lerpf(vox(v, x0, y0, z1, e), vox(v, x1, y0, z1, e), fx);
    float c11 = lerpf(vox(v, x0, y1, z1, e), vox(v, x1, y1, z1, e), fx);
    float c0 = lerpf(c00, c10, fy), c1 = lerpf(c01, c11, fy);
    float c = lerpf(c0, c1, fz);
    ch[e] = v->dtype == 0 ? c * INV255 : c;
  }
}

static void fetch_normal(const orc_volume *v, float x, float y, float z, float n[3]) {
  int x0, x1, y0, y1, z0, z1;
  float fx, fy, fz;
  lin_clamp(x, v->nx, &x0, &x1, &fx);
  lin_clamp(y, v->ny, &y0, &y1, &fy);
  lin_clamp(z, v->nz, &z0, &z1, &fz);
#define G(X, Y, Z, E) ((float)v->grad[(((size_t)(Z)*v->ny + (Y)) * v->nx + (X)) * 3 + (E)])
  for (int e = 0; e < 3; ++e) {
    float c00 = lerpf(G(x0, y0, z0, e), G(x1, y0, z0, e), fx);
    float c10 = lerpf(G(x0, y1, z0, e), G(x1, y1, z0, e), fx);
    float c01 = lerpf(G(x0, y0, z1, e), G(x1, y0, z1, e), fx);
    float c11 = lerpf(G(x0, y1, z1, e), G(x1, y1, z1, e), fx);
    float c = lerpf(lerpf(c00, c10, fy), lerpf(c01, c11, fy), fz);
    /* hardware decode 2*(b/255)-1 (NV EXPAND_NORMAL / ATI BIAS|2X, SURVEY KAT 6) */
    n[e] = fmaf(c, 2.0f * INV255, -1.0f);
  }
#undef G
}

/* bilinear RGBA8 lookup: tex[t][s][4], s,t in [0,1] (GL_LINEAR, clamp) -> 0..1 floats */
static void tex2d(const unsigned char *tex, int ss, int st, float s, float t, float out[4]) {
  int s0, s1, t0, t1;
  float fs, ft;
  lin_clamp(fmaf(s, (float)ss, -0.5f), ss, &s0, &s1, &fs);
  lin_clamp(fmaf(t, (float)st, -0.5f), st, &t0, &t1, &ft);
  for (int e = 0; e < 4; ++e) {
    float a = lerpf((float)tex[((size_t)t0 * ss + s0) * 4 + e], (float)tex[((size_t)t0 * ss + s1) * 4 + e], fs);
    float b = lerpf((float)tex[((size_t)t1 * ss + s0) * 4 + e], (float)tex[((size_t)t1 * ss + s1) * 4 + e], fs);
    out[e] = lerpf(a, b, ft) * INV255;
  }
}

static void tex3d(const unsigned char *tex, int ss, int st, int sr, float s, float t, float r,
                  float out[4]) {
  int s0, s1, t0, t1, r0, r1;
  float fs, ft, fr;
  lin_clamp(fmaf(s, (float)ss, -0.5f), ss, &s0, &s1, &fs);
  lin_clamp(fmaf(t, (float)st, -0.5f), st, &t0, &t1, &ft);
  lin_clamp(fmaf(r, (float)sr, -0.5f), sr, &r0, &r1, &fr);
#define T(S, TT, R, E) ((float)tex[((((size_t)(R)*st) + (TT)) * ss + (S)) * 4 + (E)])
  for (int e = 0; e < 4; ++e) {
    float c00 = lerpf(T(s0, t0, r0, e), T(s1, t0, r0, e), fs);
    float c10 = lerpf(T(s0, t1, r0, e), T(s1, t1, r0, e), fs);
    float c01 = lerpf(T(s0, t0, r1, e), T(s1, t0, r1, e), fs);
    float c11 = lerpf(T(s0, t1, r1, e), T(s1, t1, r1, e), fs);
    out[e] = lerpf(lerpf(c00, c10, ft), lerpf(c01, c11, ft), fr) * INV255;
  }
#undef T
}

static void noise_fetch(const orc_perturb *p, float s, float t, float r, float out[3]) {
  int n = p->n, s0, s1, t0, t1, r0, r1;
  float fs, ft, fr;
  lin_repeat(fmaf(s, (float)n, -0.5f), n, &s0, &s1, &fs);
  lin_repeat(fmaf(t, (float)n, -0.5f), n, &t0, &t1, &ft);
  lin_repeat(fmaf(r, (float)n, -0.5f), n, &r0, &r1, &fr);
#define T(S, TT, R, E) ((float)p->noise[((((size_t)(R)*n) + (TT)) * n + (S)) * 4 + (E)])
  for (int e = 0; e < 3; ++e) {
    float c00 = lerpf(T(s0, t0, r0, e), T(s1, t0, r0, e), fs);
    float c10 = lerpf(T(s0, t1, r0, e), T(s1, t1, r0, e), fs);
    float c01 = lerpf(T(s0, t0, r1, e), T(s1, t0, r1, e), fs);
    float c11 = lerpf(T(s0, t1, r1, e), T(s1, t1, r1, e), fs);
    out[e] = lerpf(lerpf(c00, c10, ft), lerpf(c01, c11, ft), fr) * INV255;
  }
#undef T
}

/* ------------------------------------------------------------ one sample -> premultiplied src */

static inline float pow30(float x) {
  float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
  return ((x16 * x8) * x4) * x2;
}

/* returns 0 if the sample contributes nothing (alpha == 0) */
static int shade_sample_s(const orc_volume *v, const orc_classify *tf, const orc_shade *sh,
                          const orc_perturb *pt, float x, float y, float z, const float *shadow, float src[4]) {
  if (pt && pt->on) {
    /* tc' = tc + sum_m w_m*(noise(tc*s_m).rgb - 0.5), two live octaves
     * (R8kVolRen3D_cpy.cpp:1590-1595, 3462-3490); tc = (vc+0.5)/N */
    float tc[3] = {(x + 0.5f) * (1.0f / (float)v->nx), (y + 0.5f) * (1.0f / (float)v->ny),
                   (z + 0.5f) * (1.0f / (float)v->nz)};
    float off[3] = {0, 0, 0};
    for (int m = 0; m < 2; ++m) {
      if (pt->w[m] == 0.0f) continue;
      float nz[3];
      noise_fetch(pt, tc[0] * pt->s[m], tc[1] * pt->s[m], tc[2] * pt->s[m], nz);
      for (int a = 0; a < 3; ++a) off[a] = fmaf(pt->w[m], nz[a] - 0.5f, off[a]);
    }
    x = fmaf(tc[0] + off[0], (float)v->nx, -0.5f);
    y = fmaf(tc[1] + off[1], (float)v->ny, -0.5f);
    z = fmaf(tc[2] + off[2], (float)v->nz, -0.5f);
  }
  float ch[4];
  fetch_voxel(v, x, y, z, ch);

  float col[4];
  if (tf->mode == ORC_TF_1D) {
    /* color table: index = round(v*(size-1)), entries premultiplied (TLUT.cpp:65-71) */
    int idx = (int)(fmaf(ch[0], (float)(tf->tlut_size - 1), 0.5f));
    if (idx < 0) idx = 0;
    if (idx > tf->tlut_size - 1) idx = tf->tlut_size - 1;
    const float *e = tf->tlut + 4 * idx;
    if (e[3] == 0.0f) return 0;
    src[0] = e[0] * e[3];
    src[1] = e[1] * e[3];
    src[2] = e[2] * e[3];
    src[3] = e[3];
    return 1;
  } else if (tf->mode == ORC_TF_2D) {
    tex2d(tf->tf_vg, tf->sv, tf->sg, ch[0], ch[1], col);
    if (tf->third_axis && tf->tf_h) {
      float h[4];
      tex2d(tf->tf_h, tf->sv, tf->sg, ch[2], ch[3], h);
      col[3] *= h[3];
    }
  } else {
    tex3d(tf->tf3d, tf->s3v, tf->s3g, tf->s3h, ch[0], ch[1], ch[2], col);
  }
  float a = sat(col[3]); /* R8kVolRen3D.cpp:2914-2917 */
  if (a == 0.0f) return 0;

  float c[3] = {col[0], col[1], col[2]};
  if (sh && sh->mode != ORC_SHADE_NONE && v->grad) {
    float n[3];
    fetch_normal(v, x, y, z, n);
    if (sh->mode == ORC_SHADE_R8K) {
      /* Nw = rows of rinfo.xform . n (R8kVolRen3D.cpp:333-339, 2831-2846) */
      const float *r = sh->xform;
      float w[3] = {fmaf(r[0], n[0], fmaf(r[4], n[1], r[8] * n[2])),
                    fmaf(r[1], n[0], fmaf(r[5], n[1], r[9] * n[2])),
                    fmaf(r[2], n[0], fmaf(r[6], n[1], r[10] * n[2]))};
      /* the cube map is addressed by direction: implicit normalisation */
      float l2 = fmaf(w[0], w[0], fmaf(w[1], w[1], w[2] * w[2]));
      float il = l2 > 0.0f ? 1.0f / sqrtf(l2) : 0.0f;
      w[0] *= il;
      w[1] *= il;
      w[2] *= il;
      float dl = fabsf(fmaf(sh->L[0], w[0], fmaf(sh->L[1], w[1], sh->L[2] * w[2])));
      float dh = fabsf(fmaf(sh->Hv[0], w[0], fmaf(sh->Hv[1], w[1], sh->Hv[2] * w[2])));
      /* diff = clamp(max(|L.n|,.2))*I ; spec = clamp(|H.n|^30)*I  (:2654-2669) */
      float kd = sat(fmaxf(sat(dl), 0.2f)) * sh->intens;
      float ks = sh->use_spec ? sat(pow30(sat(dh))) * sh->intens : 0.0f;
      float g = ch[1]; /* r0.green = second data channel (:2898-2902) */
      for (int k = 0; k < 3; ++k) {
        float shaded = fmaf(c[k], kd, ks);        /* MAD col*diff + spec (:2886-2890) */
        c[k] = fmaf(g, shaded - c[k], c[k]);      /* LERP by G                        */
        if (shadow) c[k] *= 1.0f - shadow[k];     /* MUL r0, r0, 1-r5 (:2928-2934)    */
      }
      src[0] = sat(c[0] * a); /* :2974-2977 */
      src[1] = sat(c[1] * a);
      src[2] = sat(c[2] * a);
      src[3] = a;
      return 1;
    } else {
      /* NV20 combiners (NV20VolRen3D.cpp:673-806): volume-space L/H, un-normalised N,
       * two-sided diffuse, spec = (N.H)^16 by squaring, ambient .3 */
      float dl = fabsf(fmaf(sh->L[0], n[0], fmaf(sh->L[1], n[1], sh->L[2] * n[2])));
      float dh = fmaf(sh->Hv[0], n[0], fmaf(sh->Hv[1], n[1], sh->Hv[2] * n[2]));
      float s2 = sat(dh * dh), s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8;
      float spec = sh->use_spec ? s16 * sh->intens * a : 0.0f;
      float ia = sh->intens * a, aa = 0.3f * a;
      for (int k = 0; k < 3; ++k) {
        float cc = sat(fmaf(c[k] * sat(dl), ia, c[k] * aa));
        src[k] = sat(fmaf(spec, 1.0f - cc, cc));
      }
      src[3] = a;
      return 1;
    }
  }
  if (shadow)
    for (int k = 0; k < 3; ++k) c[k] *= 1.0f - shadow[k];
  src[0] = sat(c[0] * a);
  src[1] = sat(c[1] * a);
  src[2] = sat(c[2] * a);
  src[3] = a;
  return 1;
}

static int shade_sample(const orc_volume *v, const orc_classify *tf, const orc_shade *sh,
                        const orc_perturb *pt, float x, float y, float z, float src[4]) {
  return shade_sample_s(v, tf, sh, pt, x, y, z, NULL, src);
}

/* classification alone (the light-buffer pass, R8kVolRen3D.cpp:3028-3100): straight colour, alpha =
 * sat(a_VG * a_H); returns 0 when alpha == 0 */
static int classify_sample(const orc_volume *v, const orc_classify *tf, float x, float y, float z, float col[4]) {
  float ch[4];
  fetch_voxel(v, x, y, z, ch);
  if (tf->mode == ORC_TF_2D) {
    tex2d(tf->tf_vg, tf->sv, tf->sg, ch[0], ch[1], col);
    if (tf->third_axis && tf->tf_h) {
      float h[4];
      tex2d(tf->tf_h, tf->sv, tf->sg, ch[2], ch[3], h);
      col[3] *= h[3];
    }
  } else if (tf->mode == ORC_TF_3D) {
    tex3d(tf->tf3d, tf->s3v, tf->s3g, tf->s3h, ch[0], ch[1], ch[2], col);
  } else {
    return 0;
  }
  col[3] = sat(col[3]);
  return col[3] != 0.0f;
}

/* Free clip plane (NV20VolRen3D.cpp:346-357): glClipPlane keeps eye-space points with
 * plane_eye . (x,1) >= 0.  With eye = MV * model and model = (p + 1/2)/N * fSize the same test in
 * voxel coordinates has the coefficients below (double arithmetic, rounded once to float). */
void orc_clip_plane_voxel(const double plane_eye[4], const double mv[16], const float fsize[3], const int N[3], float out[4]) {
  double pm[4];
  for (int k = 0; k < 4; ++k)
    pm[k] = plane_eye[0] * mv[4 * k + 0] + plane_eye[1] * mv[4 * k + 1] + plane_eye[2] * mv[4 * k + 2] + plane_eye[3] * mv[4 * k + 3];
  double w = pm[3];
  for (int a = 0; a < 3; ++a) {
    const double sc = (double)fsize[a] / (double)N[a];
    out[a] = (float)(pm[a] * sc);
    w += pm[a] * sc * 0.5;
  }
  out[3] = (float)w;
}

/* the box samples must lie in: the region [g0, g1) in voxel coordinates, cut by the orthogonal clip plane */
static void region_box(const orc_volume *v, float lo[3], float hi[3], int top[3]) {
  const int N[3] = {v->nx, v->ny, v->nz};
  for (int a = 0; a < 3; ++a) {
    lo[a] = (float)v->g0[a] - 0.5f;
    hi[a] = (float)v->g1[a] - 0.5f;
    top[a] = v->g1[a] == N[a]; /* outer face inclusive, interior faces half-open */
  }
  /* Orthogonal clip plane (NV20VolRen3D::setupClips, NV20VolRen3D.cpp:251-327): the box a brick is
   * sliced in ends at the plane -- the corners on the far side of it move to cp = vpos clamped into
   * the box -- so samples beyond it do not exist.  In voxel coordinates the face sits at
   * cp/fSize * N - 1/2 (texture coordinate cp/fSize, edge-to-edge mapping); it is an outer face of
   * what is drawn, hence inclusive. */
  if (v->clip_axis >= 1 && v->clip_axis <= 6) {
    const int a = (v->clip_axis - 1) / 2;
    const float fs = a == 0 ? v->fx : (a == 1 ? v->fy : v->fz);
    float cp = v->clip_vpos[a] > 0.0f ? (v->clip_vpos[a] < fs ? v->clip_vpos[a] : fs) : 0.0f;
    const float face = (float)((double)cp / (double)fs * (double)N[a] - 0.5);
    if ((v->clip_axis - 1) % 2 == 0) {  /* X+ : what lies below the plane stays */
      if (face < hi[a] || (face == hi[a] && !top[a])) { hi[a] = face; top[a] = 1; }
    } else if (face > lo[a]) {
      lo[a] = face;
    }
  }
}

/* ------------------------------------------------------------ one ray */

static long long march_pixel(const orc_volume *v, const orc_classify *tf, const orc_shade *sh,
                             const orc_perturb *pt, const orc_raycoef *rc, int blend, int i, int j,
                             float out[4], float *depth, float znear) {
  float px = fmaf((float)i + 0.5f, rc->pxs, rc->pxl);
  float py = fmaf((float)j + 0.5f, rc->pys, rc->pyl);
  float A[3], B[3];
  for (int a = 0; a < 3; ++a) {
    A[a] = fmaf(px, rc->Ax[a], fmaf(py, rc->Ay[a], rc->Ac[a]));
    B[a] = fmaf(px, rc->Bx[a], fmaf(py, rc->By[a], rc->Bc[a]));
  }
  float lo[3], hi[3];
  int top[3];
  region_box(v, lo, hi, top);
  float C[4] = {0, 0, 0, 0};
  float first = INFINITY;
  long long inside = 0;
  int S = rc->nplanes;
  for (int mm = 0; mm < S; ++mm) {
    int m = blend == 1 ? S - 1 - mm : mm;
    float p[3];
    int in = 1;
    for (int a = 0; a < 3; ++a) {
      p[a] = fmaf((float)m, B[a], A[a]);
      if (!(p[a] >= lo[a] && (p[a] < hi[a] || (top[a] && p[a] <= hi[a])))) in = 0;
    }
    if (!in) continue;
    if (v->cplane_on && !(fmaf(p[0], v->cplane[0], fmaf(p[1], v->cplane[1], fmaf(p[2], v->cplane[2], v->cplane[3]))) >= 0.0f)) continue;
    ++inside;
    float src[4];
    if (!shade_sample(v, tf, sh, pt, p[0], p[1], p[2], src)) continue;
    if (blend == 0) {
      /* C += (1-A)*src (GL_ONE_MINUS_DST_ALPHA, GL_ONE) */
      float w = 1.0f - C[3];
      if (first == INFINITY) first = fmaf((float)m, rc->dtau, rc->tau0) * znear;
      C[0] = fmaf(w, src[0], C[0]);
      C[1] = fmaf(w, src[1], C[1]);
      C[2] = fmaf(w, src[2], C[2]);
      C[3] = fmaf(w, src[3], C[3]);
    } else if (blend == 2) {
      /* D = max(S, D) per component: glBlendEquationEXT(GL_MAX), gluvvShadeMIP (NV20VolRen3D.cpp:158-163);
       * GL_MAX ignores the blend factors */
      if (first == INFINITY) first = fmaf((float)m, rc->dtau, rc->tau0) * znear;
      for (int k = 0; k < 4; ++k) C[k] = src[k] > C[k] ? src[k] : C[k];
    } else {
      /* D = S + (1-S.a)*D (GL_ONE, GL_ONE_MINUS_SRC_ALPHA), far plane first */
      float w = 1.0f - src[3];
      first = fmaf((float)m, rc->dtau, rc->tau0) * znear;
      C[0] = fmaf(w, C[0], src[0]);
      C[1] = fmaf(w, C[1], src[1]);
      C[2] = fmaf(w, C[2], src[2]);
      C[3] = fmaf(w, C[3], src[3]);
    }
  }
  memcpy(out, C, sizeof C);
  if (depth) *depth = first;
  return inside;
}

int orc_render(const orc_volume *v, const orc_classify *tf, const orc_camera *cam,
               const orc_shade *sh, const orc_perturb *pt, int blend, float *rgba, float *depth,
               int row0, int row1, int nthreads) {
  orc_raycoef rc;
  if (orc_ray_setup(v, cam, &rc)) return 1;
  long long total = 0;
  int W = cam->width;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : total)
#endif
  for (int j = row0; j < row1; ++j)
    for (int i = 0; i < W; ++i)
      total += march_pixel(v, tf, sh, pt, &rc, blend, i, j, rgba + 4 * ((size_t)j * W + i),
                           depth ? depth + (size_t)j * W + i : NULL, cam->znear);
  g_inside_samples = total;
  return 0;
}

int orc_render_pixels(const orc_volume *v, const orc_classify *tf, const orc_camera *cam,
                      const orc_shade *sh, const orc_perturb *pt, int blend, const int *pix,
                      int npix, float *out) {
  orc_raycoef rc;
  if (orc_ray_setup(v, cam, &rc)) return 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
  for (int k = 0; k < npix; ++k)
    march_pixel(v, tf, sh, pt, &rc, blend, pix[2 * k], pix[2 * k + 1], out + 4 * k, NULL, cam->znear);
  return 0;
}

/* ------------------------------------------------------------ half-angle-slicing shadows */

static double dot3d(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

int orc_shadow_setup(const orc_volume *v, const orc_camera *cam, const float light_pos[3], const float eye[3],
                     const float at[3], const float xform[16], int buffer_px, float quality, orc_shadowcoef *o) {
  memset(o, 0, sizeof *o);
  const double f[3] = {v->fx, v->fy, v->fz}, N[3] = {v->nx, v->ny, v->nz};
  /* view and light direction in world space (R8kVolRen3D.cpp:296-305: axis (0,0,1) IS the world view
   * direction of gluvv's camera; ldir = -norm(light.pos)) */
  double vd[3] = {(double)at[0] - eye[0], (double)at[1] - eye[1], (double)at[2] - eye[2]};
  double ld[3] = {-(double)light_pos[0], -(double)light_pos[1], -(double)light_pos[2]};
  double vl = sqrt(dot3d(vd, vd)), d0 = sqrt(dot3d(ld, ld));
  if (!(vl > 0) || !(d0 > 0)) return 1;
  for (int k = 0; k < 3; ++k) { vd[k] /= vl; ld[k] /= d0; }
  const double vdl = dot3d(vd, ld);
  if (vdl <= 0) for (int k = 0; k < 3; ++k) vd[k] = -vd[k];   /* :307-311 */
  double h[3];
  for (int k = 0; k < 3; ++k) h[k] = (vd[k] - ld[k]) * .5 + ld[k];   /* (v - l)/2 + l, :316-320 */
  o->front_to_back = vdl > 0;
  /* model <- world for directions: inverse of rinfo.xform's linear part (mvinv . axis, :1321-1324) */
  double xf[16], xinv[16];
  for (int i = 0; i < 16; ++i) xf[i] = xform[i];
  affine_inverse(xinv, xf);
  double sn[3];
  for (int a = 0; a < 3; ++a) sn[a] = xinv[0 + a] * h[0] + xinv[4 + a] * h[1] + xinv[8 + a] * h[2];
  const double snl = sqrt(dot3d(sn, sn));
  if (!(snl > 0)) return 1;
  for (int a = 0; a < 3; ++a) sn[a] /= snl;
  /* slice planes sn . X = c_k = tmin + k dc, k = 1..S, over the whole volume's corners */
  double tmin = 1e300, tmax = -1e300;
  for (int i = 0; i < 8; ++i) {
    const double X[3] = {(i & 1) ? f[0] : 0, (i & 2) ? f[1] : 0, (i & 4) ? f[2] : 0};
    const double t = dot3d(sn, X);
    if (t < tmin) tmin = t;
    if (t > tmax) tmax = t;
  }
  double dc;
  int S;
  if (cam->steps > 0) {
    S = cam->steps;
    dc = (tmax - tmin) / S;
  } else {
    const float disf = v->fx / ((float)v->nx * cam->sample_rate);  /* :1330 */
    dc = disf;
    S = (int)((tmax - tmin) / dc);
  }
  if (S < 0) S = 0;
  o->nslices = S;
  /* eye rays: X = e + tau (R0 px + R1 py - n R2) (see orc_ray_setup) */
  double inv[16];
  affine_inverse(inv, cam->mv);
  const double n = cam->znear;
  const double l = cam->frustum[0], r = cam->frustum[1], b = cam->frustum[2], t = cam->frustum[3];
  o->pxs = (float)((r - l) / cam->width);
  o->pxl = (float)l;
  o->pys = (float)((t - b) / cam->height);
  o->pyl = (float)b;
  const double R0[3] = {inv[0], inv[1], inv[2]}, R1[3] = {inv[4], inv[5], inv[6]}, R2[3] = {inv[8], inv[9], inv[10]};
  const double e[3] = {inv[12], inv[13], inv[14]};
  for (int a = 0; a < 3; ++a) {
    const double s = N[a] / f[a];
    o->Ec[a] = (float)(e[a] * s - 0.5);
    o->Dx[a] = (float)(R0[a] * s);
    o->Dy[a] = (float)(R1[a] * s);
    o->Dc[a] = (float)(-n * R2[a] * s);
  }
  o->nDx = (float)dot3d(sn, R0);
  o->nDy = (float)dot3d(sn, R1);
  o->nDc = (float)(-n * dot3d(sn, R2));
  o->num0 = (float)(tmin - dot3d(sn, e));
  o->dnum = (float)dc;
  /* light transform (LTWidgetRen::genXForm, LTWidgetRen.cpp:231-291): gluLookAt(eye = -norm(light.pos),
   * at 0, up y) with its z translation negated, then w = 1 + z'/d0: light-view coordinates of a world
   * point q are x' = s.q, y' = u.q, z' = 1 - F.q (F = norm(light.pos)), w = 1 + z'/d0 */
  const double F[3] = {-ld[0], -ld[1], -ld[2]};
  double sv[3] = {F[1] * 0 - F[2] * 1, F[2] * 0 - F[0] * 0, F[0] * 1 - F[1] * 0};   /* F x (0,1,0) */
  const double sl = sqrt(dot3d(sv, sv));
  if (!(sl > 1e-12)) return 1;
  for (int k = 0; k < 3; ++k) sv[k] /= sl;
  const double uv[3] = {sv[1] * F[2] - sv[2] * F[1], sv[2] * F[0] - sv[0] * F[2], sv[0] * F[1] - sv[1] * F[0]};  /* s x F */
  /* world q = xform (X - f/2) (ltxf = light.xf . xform . tb, R8kVolRen3D.cpp:1280-1290) */
  double rowx[4], rowy[4], roww[4];   /* affine forms of the model point X */
  for (int a = 0; a < 3; ++a) {
    const double col[3] = {xf[4 * a + 0], xf[4 * a + 1], xf[4 * a + 2]};   /* xform's column a */
    rowx[a] = dot3d(sv, col);
    rowy[a] = dot3d(uv, col);
    roww[a] = -dot3d(F, col) / d0;
  }
  {
    const double tcol[3] = {xf[12], xf[13], xf[14]};
    rowx[3] = dot3d(sv, tcol);
    rowy[3] = dot3d(uv, tcol);
    roww[3] = 1.0 + (1.0 - dot3d(F, tcol)) / d0;
    for (int a = 0; a < 3; ++a) {
      rowx[3] -= rowx[a] * f[a] * .5;
      rowy[3] -= rowy[a] * f[a] * .5;
      roww[3] -= roww[a] * f[a] * .5;
    }
  }
  /* ... in voxel coordinates: X_a = (p_a + 1/2) f_a / N_a */
  double cx = rowx[3], cy = rowy[3], cw = roww[3];
  for (int a = 0; a < 3; ++a) {
    const double sc = f[a] / N[a];
    o->Xm[a] = (float)(rowx[a] * sc);
    o->Ym[a] = (float)(rowy[a] * sc);
    o->Wm[a] = (float)(roww[a] * sc);
    cx += rowx[a] * sc * .5;
    cy += rowy[a] * sc * .5;
    cw += roww[a] * sc * .5;
  }
  o->Xm[3] = (float)cx;
  o->Ym[3] = (float)cy;
  o->Wm[3] = (float)cw;
  /* light-buffer coordinates lc = (x'/w * .85 + .5) * quality, in texels of a buffer_px^2 buffer (:1673-1674) */
  const double LBf = (double)quality * (double)buffer_px;
  o->LB = (int)ceil(LBf);
  if (o->LB < 1) return 1;
  o->lscale = (float)(.85 * LBf);
  o->lbias = (float)(.5 * LBf);
  o->las = (float)(1.0 / (.85 * LBf));
  o->lal = (float)(-.5 / .85);
  /* light rays, the inverse of the above: q(w) = F (1 + d0) + w (a s + b u - d0 F); X = xform^-1 q + f/2 */
  double apex[3], gx[3], gy[3], gc[3];
  for (int a = 0; a < 3; ++a) {
    apex[a] = xinv[12 + a] + f[a] * .5;
    gx[a] = gy[a] = gc[a] = 0;
    for (int k = 0; k < 3; ++k) {
      apex[a] += xinv[4 * k + a] * F[k] * (1.0 + d0);
      gx[a] += xinv[4 * k + a] * sv[k];
      gy[a] += xinv[4 * k + a] * uv[k];
      gc[a] += xinv[4 * k + a] * F[k] * -d0;
    }
  }
  for (int a = 0; a < 3; ++a) {
    const double s = N[a] / f[a];
    o->Lc[a] = (float)(apex[a] * s - 0.5);
    o->Gx[a] = (float)(gx[a] * s);
    o->Gy[a] = (float)(gy[a] * s);
    o->Gc[a] = (float)(gc[a] * s);
  }
  o->nGx = (float)dot3d(sn, gx);
  o->nGy = (float)dot3d(sn, gy);
  o->nGc = (float)dot3d(sn, gc);
  o->lnum0 = (float)(tmin - dot3d(sn, apex));
  o->ldnum = (float)dc;
  return 0;
}

/* bilinear lookup of the light buffer, texels outside it are 0 (the rest of the pbuffer stays cleared) */
static void light_lookup(const float *L, int LB, float lx, float ly, float out[3]) {
  const float fx0 = floorf(lx - 0.5f), fy0 = floorf(ly - 0.5f);
  const float fx = (lx - 0.5f) - fx0, fy = (ly - 0.5f) - fy0;
  out[0] = out[1] = out[2] = 0.0f;
  if (!(fx0 >= -1.0f && fx0 < (float)LB && fy0 >= -1.0f && fy0 < (float)LB)) return;   /* (also NaN) */
  const int x0 = (int)fx0, y0 = (int)fy0;
  for (int k = 0; k < 3; ++k) {
    float t[4];
    for (int q = 0; q < 4; ++q) {
      const int x = x0 + (q & 1), y = y0 + (q >> 1);
      t[q] = (x >= 0 && x < LB && y >= 0 && y < LB) ? L[4 * ((size_t)y * LB + x) + k] : 0.0f;
    }
    out[k] = lerpf(lerpf(t[0], t[1], fx), lerpf(t[2], t[3], fx), fy);
  }
}

int orc_render_shadow(const orc_volume *v, const orc_classify *tf, const orc_camera *cam, const orc_shade *sh,
                      const orc_shadowcoef *sc, float *rgba, float *light_out, int nthreads) {
  if (tf->mode == ORC_TF_1D || (sh && sh->mode == ORC_SHADE_NV20)) return 2;
  const int W = cam->width, H = cam->height, LB = sc->LB;
  /* the box both passes sample in: the volume, or what an orthogonal clip plane leaves of it (both passes draw the same
   * clipped slice polygons, and glClipPlane stays enabled through both); closed */
  float blo[3], bhi[3];
  int btop[3];
  region_box(v, blo, bhi, btop);
  float *L0 = (float *)calloc((size_t)LB * LB * 4, sizeof(float));
  float *L1 = (float *)calloc((size_t)LB * LB * 4, sizeof(float));
  if (!L0 || !L1) { free(L0); free(L1); return 1; }
  memset(rgba, 0, sizeof(float) * 4 * (size_t)W * H);
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#endif
  for (int k = 1; k <= sc->nslices; ++k) {
    const float lnum = fmaf((float)k, sc->ldnum, sc->lnum0);
    /* The eye sample of slice k is plane m of the pixel's ray, planes counted FROM THE EYE (m = k - 1 when the slices run away
     * from the viewer, nslices - k when they run towards it): with D_a = fma(px, Dx_a, fma(py, Dy_a, Dc_a)),
     * nD = fma(px, nDx, fma(py, nDy, nDc)), numA = the numerator of plane 0 (fma(1, dnum, num0) or fma(nslices, dnum, num0)),
     * dB = +-dnum:  tauA = numA / nD, dtau = dB / nD, A_a = fma(tauA, D_a, Ec_a), B_a = dtau * D_a, sample = fma(m, B, A),
     * which exists where fma(m, dtau, tauA) is positive and finite.  (Round 3: the product's ray-marchers render the eye
     * pass as one march per pixel over these planes; the chain is theirs -- smk_ray_AB, smk_device.h.) */
    const float numA = sc->front_to_back ? fmaf(1.0f, sc->dnum, sc->num0) : fmaf((float)sc->nslices, sc->dnum, sc->num0);
    const float dB = sc->front_to_back ? sc->dnum : -sc->dnum;
    const int m = sc->front_to_back ? k - 1 : sc->nslices - k;
    /* ---- eye pass: reads L0 (the buffer as the previous slices left it) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
    for (int j = 0; j < H; ++j)
      for (int i = 0; i < W; ++i) {
        const float px = fmaf((float)i + 0.5f, sc->pxs, sc->pxl), py = fmaf((float)j + 0.5f, sc->pys, sc->pyl);
        const float nD = fmaf(px, sc->nDx, fmaf(py, sc->nDy, sc->nDc));
        const float tauA = numA / nD, dtau = dB / nD;
        if (!(fabsf(nD) > 0.0f) || !(fabsf(tauA) < INFINITY) || !(fabsf(dtau) < INFINITY)) continue;
        const float tau = fmaf((float)m, dtau, tauA);
        if (!(tau > 0.0f) || !(tau < INFINITY)) continue;
        float p[3];
        int in = 1;
        for (int a = 0; a < 3; ++a) {
          const float D = fmaf(px, sc->Dx[a], fmaf(py, sc->Dy[a], sc->Dc[a]));
          const float A = fmaf(tauA, D, sc->Ec[a]), B = dtau * D;
          p[a] = fmaf((float)m, B, A);
          /* (the box 2^-10 voxels wide: the last slice lies ON the far corner / face, where the chain's rounding would decide) */
          if (!(p[a] >= blo[a] - 0.0009765625f && p[a] <= bhi[a] + 0.0009765625f)) in = 0;
        }
        if (!in) continue;
        if (v->cplane_on && !(fmaf(p[0], v->cplane[0], fmaf(p[1], v->cplane[1], fmaf(p[2], v->cplane[2], v->cplane[3]))) >= 0.0f)) continue;
        float *C = rgba + 4 * ((size_t)j * W + i);
        if (sc->front_to_back && C[3] == 1.0f) continue;   /* (no later sample can change the pixel) */
        const float lw = fmaf(p[0], sc->Wm[0], fmaf(p[1], sc->Wm[1], fmaf(p[2], sc->Wm[2], sc->Wm[3])));
        const float lx = fmaf(fmaf(p[0], sc->Xm[0], fmaf(p[1], sc->Xm[1], fmaf(p[2], sc->Xm[2], sc->Xm[3]))) / lw, sc->lscale, sc->lbias);
        const float ly = fmaf(fmaf(p[0], sc->Ym[0], fmaf(p[1], sc->Ym[1], fmaf(p[2], sc->Ym[2], sc->Ym[3]))) / lw, sc->lscale, sc->lbias);
        float shadow[3], src[4];
        light_lookup(L0, LB, lx, ly, shadow);
        if (!shade_sample_s(v, tf, sh, NULL, p[0], p[1], p[2], shadow, src)) continue;
        if (sc->front_to_back) {
          const float w = 1.0f - C[3];
          for (int q = 0; q < 4; ++q) C[q] = fmaf(w, src[q], C[q]);
        } else {
          const float w = 1.0f - src[3];
          for (int q = 0; q < 4; ++q) C[q] = fmaf(w, C[q], src[q]);
        }
      }
    /* ---- light pass: L1 = this slice's classification composited onto L0 (LERP by alpha, saturated;
     * alpha = sat((1-a) L.a + a), R8kVolRen3D.cpp:3150-3165) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
    for (int y = 0; y < LB; ++y)
      for (int x = 0; x < LB; ++x) {
        const float *Lo = L0 + 4 * ((size_t)y * LB + x);
        float *Ln = L1 + 4 * ((size_t)y * LB + x);
        memcpy(Ln, Lo, 4 * sizeof(float));
        const float a = fmaf((float)x + 0.5f, sc->las, sc->lal), b = fmaf((float)y + 0.5f, sc->las, sc->lal);
        const float nG = fmaf(a, sc->nGx, fmaf(b, sc->nGy, sc->nGc));
        const float w = lnum / nG;
        if (!(w > 0.0f) || isinf(w)) continue;
        float p[3];
        int in = 1;
        for (int q = 0; q < 3; ++q) {
          const float G = fmaf(a, sc->Gx[q], fmaf(b, sc->Gy[q], sc->Gc[q]));
          p[q] = fmaf(w, G, sc->Lc[q]);
          if (!(p[q] >= blo[q] && p[q] <= bhi[q])) in = 0;
        }
        if (!in) continue;
        if (v->cplane_on && !(fmaf(p[0], v->cplane[0], fmaf(p[1], v->cplane[1], fmaf(p[2], v->cplane[2], v->cplane[3]))) >= 0.0f)) continue;
        float col[4];
        if (!classify_sample(v, tf, p[0], p[1], p[2], col)) continue;
        for (int q = 0; q < 3; ++q) Ln[q] = sat(fmaf(col[3], sat(col[q]) - Lo[q], Lo[q]));
        Ln[3] = sat(fmaf(1.0f - col[3], Lo[3], col[3]));
      }
    float *t = L0;
    L0 = L1;
    L1 = t;
  }
  if (light_out) memcpy(light_out, L0, sizeof(float) * 4 * (size_t)LB * LB);
  free(L0);
  free(L1);
  return 0;
}

void orc_composite_over(const float *layers, int nlayers, int npix, float *out) {
  for (int p = 0; p < npix; ++p) {
    float C[4] = {0, 0, 0, 0};
    for (int l = 0; l < nlayers; ++l) {
      const float *s = layers + ((size_t)l * npix + p) * 4;
      float w = 1.0f - C[3];
      for (int k = 0; k < 4; ++k) C[k] = fmaf(w, s[k], C[k]);
    }
    memcpy(out + 4 * (size_t)p, C, sizeof C);
  }
}

/* ------------------------------------------------------------ shading vectors */

static void norm3(float v[3]) {
  float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (l > 0) {
    v[0] /= l;
    v[1] /= l;
    v[2] /= l;
  }
}

void orc_shade_setup(int mode, int use_spec, const float light_pos[3], const float eye[3],
                     const float at[3], const float xform[16], float intens, orc_shade *o) {
  memset(o, 0, sizeof *o);
  o->mode = mode;
  o->use_spec = use_spec;
  o->intens = intens;
  memcpy(o->xform, xform, 16 * sizeof(float));
  if (mode == ORC_SHADE_R8K) {
    /* ldir = -norm(light.pos); vdir = -norm(eye-at); half = norm(ldir + (vdir-ldir)/2)
     * (R8kVolRen3D.cpp:2625-2640) */
    float l[3] = {-light_pos[0], -light_pos[1], -light_pos[2]};
    norm3(l);
    float vd[3] = {-(eye[0] - at[0]), -(eye[1] - at[1]), -(eye[2] - at[2])};
    norm3(vd);
    float h[3];
    for (int k = 0; k < 3; ++k) h[k] = l[k] + 0.5f * (vd[k] - l[k]);
    norm3(h);
    memcpy(o->L, l, sizeof l);
    memcpy(o->Hv, h, sizeof h);
  } else if (mode == ORC_SHADE_NV20) {
    /* vdir = norm(eye-at); ltdir = norm(light.pos-at); half = ltdir + (vdir-ltdir)/2;
     * both through inverse(rinfo.xform), negated, normalised (NV20VolRen3D.cpp:637-668).
     * xform is a rotation: inverse = transpose. */
    float vd[3] = {eye[0] - at[0], eye[1] - at[1], eye[2] - at[2]};
    norm3(vd);
    float lt[3] = {light_pos[0] - at[0], light_pos[1] - at[1], light_pos[2] - at[2]};
    norm3(lt);
    float h[3];
    for (int k = 0; k < 3; ++k) h[k] = lt[k] + 0.5f * (vd[k] - lt[k]);
    const float *r = xform;
    float hv[3] = {r[0] * h[0] + r[1] * h[1] + r[2] * h[2], r[4] * h[0] + r[5] * h[1] + r[6] * h[2],
                   r[8] * h[0] + r[9] * h[1] + r[10] * h[2]};
    float lv[3] = {r[0] * lt[0] + r[1] * lt[1] + r[2] * lt[2], r[4] * lt[0] + r[5] * lt[1] + r[6] * lt[2],
                   r[8] * lt[0] + r[9] * lt[1] + r[10] * lt[2]};
    for (int k = 0; k < 3; ++k) {
      hv[k] = -hv[k];
      lv[k] = -lv[k];
    }
    norm3(hv);
    norm3(lv);
    memcpy(o->L, lv, sizeof lv);
    memcpy(o->Hv, hv, sizeof hv);
  }
}
