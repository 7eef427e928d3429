/*
 * smk_oracle.h -- CPU restatement of the Simian/spaceMonkey volume-rendering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The HIP
 * product (simian-spacemonkey_amd/csrc) never includes, links or calls it.
 *
 * PARITY UNPINNED (renderer): the reference (/root/reference, zzmuxi/simian-spacemonkey) has
 * no CPU renderer, no tests and no golden images (its renderers are fixed-function OpenGL
 * state for 2001 GPUs and cannot be built or run here), so the ray-marcher below is pinned
 * only by closed-form known-answer tests derived from the cited reference lines
 * (tests/test_oracle_kat.py).  The one reference source that does build here,
 * genvol/perlin.c, pins the Perlin restatement through oracle/_ref/libperlin_ref.so
 * (see oracle/Makefile and tests/test_perlin_ref.py).
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef SMK_ORACLE_H
#define SMK_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ renderer */

/* One dense volume (the unbricked parity target) + the sub-box this call renders
 * (g0/g1 = whole volume for single-GPU parity; a brick region for the sort-last tests). */
typedef struct {
  int nx, ny, nz;            /* voxels (Volume::x/y/ziSize, MetaVolume.h:18-61)            */
  int nelts;                 /* interleaved channels per voxel, x fastest (MetaVolume.cpp:1175) */
  int dtype;                 /* 0 = u8 (decoded /255), 1 = f32 already in [0,1]            */
  const void *data;          /* [nz][ny][nx][nelts]                                        */
  const unsigned char *grad; /* [nz][ny][nx][3] scale-biased normals or NULL               */
  float fx, fy, fz;          /* extent in model space (Volume::x/y/zfSize)                 */
  int g0[3], g1[3];          /* region [g0,g1) in voxel indices (x,y,z)                    */
  int clip_axis;             /* orthogonal clip plane: 0 off, 1..6 = X+ X- Y+ Y- Z+ Z- (VolRenMajorAxis, gluvv.h:136-144) */
  float clip_vpos[3];        /* its position in volume space (gluvv.clip.vpos)             */
  int cplane_on;             /* free clip plane (glClipPlane): keep fma-chain(cplane . (p,1)) >= 0, voxel coordinates */
  float cplane[4];
} orc_volume;

enum { ORC_TF_1D = 0, ORC_TF_2D = 1, ORC_TF_3D = 2 };

typedef struct {
  int mode;
  /* 1-D: straight (non-premultiplied) float RGBA, TLUT::_rgba (TLUT.h:16-116) */
  const float *tlut;
  int tlut_size;
  /* 2-D: deptex[g][v][RGBA8] (+ optional deptex2[t][s] alpha-only third axis),
   * NV20VolRen3D.cpp:1466-1574; alpha already opacity-corrected by the caller. */
  const unsigned char *tf_vg;
  int sv, sg;
  const unsigned char *tf_h; /* NULL or same dims as tf_vg */
  int third_axis;            /* alpha *= tf_h(s=ch2,t=ch3).a  (NV20VolRen3D.cpp:821-826) */
  /* 3-D dense: ptex[h][g][v][RGBA8], TFWidgetRen.cpp:98-124, 779-845 */
  const unsigned char *tf3d;
  int s3v, s3g, s3h;
} orc_classify;

typedef struct {
  double mv[16];       /* column-major modelview (VolumeRenderable.cpp:40-49)         */
  float frustum[4];    /* left,right,bottom,top at the near plane (gluvv.cpp:544-549) */
  float znear;         /* gluvv.env.clip[0]                                           */
  int width, height;
  float sample_rate;   /* planes per voxel (VolumeRenderer.cpp:595), used if steps==0 */
  int steps;           /* fixed plane count, 0 = use sample_rate          */
} orc_camera;

enum { ORC_SHADE_NONE = 0, ORC_SHADE_R8K = 1, ORC_SHADE_NV20 = 2 };

typedef struct {
  int mode;
  float L[3];       /* R8k: world-space light dir; NV20: volume-space light dir */
  float Hv[3];      /* half vector, same space as L                              */
  float xform[16];  /* gluvv.rinfo.xform (column-major rotation)                 */
  float intens;     /* gluvv.light.intens                                        */
  int use_spec;     /* gluvvShadeDSpec vs gluvvShadeDiff                         */
} orc_shade;

typedef struct {
  int on;
  const unsigned char *noise; /* [n][n][n][4] RGBA8, GL_REPEAT */
  int n;
  float w[4], s[4];           /* gluvv.pert.weights / scales; octaves 0..1 live */
} orc_perturb;

/* ray coefficients shared by every implementation (see DESIGN.md "sample placement") */
typedef struct {
  float pxs, pxl, pys, pyl;       /* px = fma(i+.5, pxs, pxl)                    */
  float Ac[3], Ax[3], Ay[3];      /* A_a = fma(px,Ax,fma(py,Ay,Ac))              */
  float Bc[3], Bx[3], By[3];      /* B_a likewise; vc_a(m) = fma(m, B_a, A_a)    */
  int nplanes;                    /* S                                           */
  float tau0, dtau;               /* ray parameter of plane m: fma(m,dtau,tau0)  */
  float zmin, zmax, dis;
} orc_raycoef;

int orc_ray_setup(const orc_volume *v, const orc_camera *c, orc_raycoef *out);

/* blend: 2 = maximum per component (GL_MAX, gluvvShadeMIP, NV20VolRen3D.cpp:158-163),
 * 0 = front-to-back (R8kVolRen3D.cpp:1441-1449), 1 = back-to-front
 * (VolumeRenderer.cpp:589-590).  rgba: [height][width][4] premultiplied; depth may be NULL.
 * Rows [row0,row1) only (others untouched).  Returns 0 ok. */
int orc_render(const orc_volume *v, const orc_classify *tf, const orc_camera *cam,
               const orc_shade *sh, const orc_perturb *pt, int blend,
               float *rgba, float *depth, int row0, int row1, int nthreads);

/* Sparse variant for full-size parity: pix = npix (i,j) pairs, out = npix*4. */
int orc_render_pixels(const orc_volume *v, const orc_classify *tf, const orc_camera *cam,
                      const orc_shade *sh, const orc_perturb *pt, int blend,
                      const int *pix, int npix, float *out);

/* ---- half-angle-slicing shadows (R8kVolRen3D.cpp:296-326 slice axis, :1651-1868 volShadow,
 * :2991-3180 light-buffer shader, :2928-2934 the eye shader's 1 - shadow term; light transform
 * LTWidgetRen.cpp:231-291).  INTENDED behaviour, stated in DESIGN.md: slices perpendicular to the
 * half-way vector of view and light direction (view direction flipped when the light faces the
 * viewer), marched away from the light; per slice the eye pass shades every pixel's sample by
 * 1 - L(light-buffer position of the sample) and blends it (under when the slices run away from the
 * eye, over otherwise), then the light pass composites the slice's classification into L.
 * Both passes place samples with the fma chains below (shared by every implementation). */
typedef struct {
  float pxs, pxl, pys, pyl;     /* eye rays: px = fma(i+.5, pxs, pxl), py likewise                        */
  float Ec[3];                  /* eye point, voxel coordinates                                           */
  float Dc[3], Dx[3], Dy[3];    /* D_a = fma(px, Dx_a, fma(py, Dy_a, Dc_a)); sample = fma(tau, D_a, Ec_a) */
  float nDc, nDx, nDy;          /* nD = fma(px, nDx, fma(py, nDy, nDc)); tau = fma(k, dnum, num0) / nD    */
  float num0, dnum;
  float las, lal;               /* light rays: a = fma(u+.5, las, lal), b = fma(v+.5, las, lal)           */
  float Lc[3];                  /* light apex, voxel coordinates                                          */
  float Gc[3], Gx[3], Gy[3];    /* G_a = fma(a, Gx_a, fma(b, Gy_a, Gc_a)); sample = fma(w, G_a, Lc_a)     */
  float nGc, nGx, nGy;          /* nG likewise; w = fma(k, ldnum, lnum0) / nG                             */
  float lnum0, ldnum;
  float Xm[4], Ym[4], Wm[4];    /* voxel -> light space: x' = fma(p0,Xm0,fma(p1,Xm1,fma(p2,Xm2,Xm3))), y', w */
  float lscale, lbias;          /* light-buffer pixel coordinate = fma(x'/w, lscale, lbias)               */
  int nslices;                  /* k = 1..nslices                                                         */
  int LB;                       /* light buffer is LB x LB texels                                         */
  int front_to_back;            /* 1: slices run away from the eye (blend under), 0: towards it (over)    */
} orc_shadowcoef;

/* buffer_px = gluvv.light.buffsz[0] (gluvv.cpp:287), quality = g/iShadowQual (:299-300).  Returns 0 ok,
 * 1 when the light sits on the y axis (the reference's gluLookAt(up = y) degenerates there). */
int orc_shadow_setup(const orc_volume *v, const orc_camera *cam, const float light_pos[3], const float eye[3],
                     const float at[3], const float xform[16], int buffer_px, float quality, orc_shadowcoef *out);

/* one frame with shadows.  rgba [height][width][4] premultiplied; light_out (may be NULL) [LB][LB][4] =
 * the light buffer after the last slice.  sh->mode NV20 and the 1-D table are refused (return 2). */
int orc_render_shadow(const orc_volume *v, const orc_classify *tf, const orc_camera *cam, const orc_shade *sh,
                      const orc_shadowcoef *sc, float *rgba, float *light_out, int nthreads);

/* in-volume samples (alpha-independent) visited by the last orc_render call in this thread
 * group; used by bench.py to print nominal vs in-volume sample counts */
long long orc_last_inside_samples(void);

/* ordered "over" of P premultiplied layers, front first (sort-last composite) */
void orc_composite_over(const float *layers, int nlayers, int npix, float *out);

/* shading vectors: R8kVolRen3D::loadCubeTex (:2620-2640), NV20VolRen3D::setupRegComb (:637-668) */
void orc_shade_setup(int mode, int use_spec, const float light_pos[3], const float eye[3],
                     const float at[3], const float xform[16], float intens, orc_shade *out);

/* camera helper: LookAt * T(trans) * R(xform) * T(-fSize/2)  (VolumeRenderable.cpp:40-49,
 * gluvv.cpp:531-540); GL definition of gluLookAt. */
void orc_modelview(const float eye[3], const float at[3], const float up[3],
                   const float trans[3], const float xform[16], const float fsize[3],
                   double mv[16]);

/* ------------------------------------------------------------------ data prep */
/* glibc TYPE_3 rand() clone so synthetic inputs do not depend on the libc in use */
void orc_srand(unsigned seed);
int orc_rand(void);

void orc_perlin_init(void); /* genvol/perlin.c:145-176 (tables from orc_rand) */
void orc_perlin_reset(void); /* re-arm perlin.c's `start` flag (= a fresh process) */
double orc_noise3(const double vec[3]);
double orc_perlin3d(double x, double y, double z, double alpha, double beta, int n);
double orc_perlin3d_abs(double x, double y, double z, double alpha, double beta, int n);

/* genvol/main.cpp:212-256 (spheres), :306-332 (perl), :334-430 (blur) */
void orc_genvol_spheres(unsigned char *d, int sx, int sy, int sz, int nspheres, int use_perl,
                        int ntype, int pharm, double pscale, const float pwrap[3],
                        float palpha, float pbeta);
void orc_genvol_perl(unsigned char *d, int sx, int sy, int sz, int param, int pharm,
                     const float pwrap[3], float palpha, float pbeta);
void orc_genvol_blur(unsigned char *d, int sx, int sy, int sz, const float bw[4]);

/* genVGH/main.cpp:56-182.  compat=1 keeps the tv[1] typo.  in_dtype 0=u8 1=f32.
 * out_u8: [sz][sy][sx][3] or NULL; out_f32 (unquantised, scaled to [0,1]) or NULL */
void orc_make_vgh(const void *in, int in_dtype, int sx, int sy, int sz, int compat,
                  unsigned char *out_u8, float *out_f32);

/* MetaVolume::normalsVGH (MetaVolume.cpp:1274-1324): derivative3DVGH + blurV3D + scalebiasN.
 * data = u8 [..][nelts] (channel 0 differenced); out [..][3] */
void orc_clip_plane_voxel(const double plane_eye[4], const double mv[16], const float fsize[3], const int N[3], float out[4]);
int orc_hist2d(const unsigned char *data, int nelts, long long nvox, unsigned char *hist);
void orc_normals_vgh(const unsigned char *data, int nelts, int sx, int sy, int sz, int blur,
                     unsigned char *out);

/* MetaVolume::mergeMV-style G append for multi-field data (MetaVolume.cpp:1109-1268,
 * AGradArb VectorMath.h:945-1004, GMag :1010): in [..][nfields] u8 -> out [..][nfields+1] */
void orc_merge_addg(const unsigned char *in, int nfields, int sx, int sy, int sz,
                    unsigned char *out, unsigned char *grad_out);

/* MetaVolume::brick(maxsz) (MetaVolume.cpp:1369-1417): grid dims */
void orc_brick_grid(int sx, int sy, int sz, int maxsz, int dims[3]);

/* TLUT (TLUT.cpp) on a float[size*4] straight-colour table */
void orc_tlut_default(float *rgba, int size);                 /* :26-36  */
void orc_tlut_spectral(float *rgba, int size);                /* :201-298 */
void orc_tlut_blackbody(float *rgba, int size);               /* :454-472 (quirk q5) */
void orc_tlut_cyanmagenta(float *rgba, int size);             /* :300-316 */
void orc_tlut_channel_ramp(float *rgba, int ch, int i0, int i1, float v0, float v1); /* :125 */
void orc_tlut_scale_alpha(float *rgba, int size, float last_rate, float rate);       /* :138 */
void orc_tlut_premultiply(const float *rgba, int size, float *table);                /* :65-71 */

/* NV20VolRen3D::create2DDepTex defaults (:1479-1486, :1523-1530) and copyScale (:1645-1660) */
void orc_deptex_default(unsigned char *deptex, unsigned char *deptex2, int sx, int sy);
void orc_copy_scale(const unsigned char *in, unsigned char *out, int sx, int sy, float sr);

/* LevWidget::rasterize (LevWidget.cpp:674-1074), types as LevWidget.h:117-120: 0 triangle,
 * 1 ellipse ("square"), 2 1-D style, 3 default style */
typedef struct {
  int type;
  float verts[3][2];  /* bottom, left, right (setPos, :1098-1125) */
  float thresh[2];
  float color[3];
  float alpha;
  float be;           /* boundary emphasis */
  int faux;           /* gluvv.shade == gluvvShadeFaux */
} orc_levwidget;
void orc_lev_setpos(orc_levwidget *w, const float b[2], const float l[2], const float r[2],
                    float tw, float th);
void orc_lev_rasterize(const orc_levwidget *w, unsigned char *tex, int sv, int sg, int sh);
void orc_hsl_color(float h, float s, float l, float col[3]);  /* HSLPicker.cpp:33-68 */
/* TFWidgetRen::rasterizevgH (TFWidgetRen1.cpp:1035-1062), VGH branch */
void orc_rasterize_vgh(unsigned char *ptex, int sx, int sy, float slider1hi);

/* R8kVolRen3D_cpy::createNoiseTex (:2392-2436): n^3 RGBA8 from srand(1) */
void orc_noise_tex(unsigned char *out, int n);

#ifdef __cplusplus
}
#endif
#endif
