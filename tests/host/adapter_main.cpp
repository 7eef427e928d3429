// Drives the host-side mirror the way Simian's main() drives a renderer (gluvv.cpp:141-199,
// 518-525, 593-597): fill `gluvv`, new the primitive, link it, init() once, draw() per frame.
// usage: adapter_main <vol.u8 nx ny nz nelts> <grad.u8|-> <deptex.rgba|-> <W> <H> <rate> <shade 0|3> <xform16...> <out.f32>
//        ... [key=value ...] after <out.f32>: dmode=<gluvvDataMode>, plat=<gluvvPlatform>, tfsize=<sv>,<sg>,<sh> (a table of
//        several sheets is the dense 3-D transfer function), deptex2=<file>, pert=<w0>,<w1>,<s0>,<s1> (gluvv.pert, on)
//        adapter_main <dataset.trex 0 0 0 1> ...   the volume comes from disk the way `gluvv data.trex` loads it
//                                                   (MetaVolume(file) + readAll(tstart), gluvv.cpp:160-176)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "HipVolumeRenderer.h"
#include "VolumeFiles.h"

gluvvGlobal gluvv;

static std::vector<unsigned char> slurp(const char *p) {
  std::vector<unsigned char> v;
  FILE *f = fopen(p, "rb");
  if (!f) return v;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize(n);
  if (fread(v.data(), 1, n, f) != (size_t)n) v.clear();
  fclose(f);
  return v;
}

int main(int argc, char **argv) {
  if (argc < 12 + 16) {
    fprintf(stderr, "bad usage\n");
    return 2;
  }
  gluvvCompatDefaults(gluvv);  // what initGluvv() does (gluvv.cpp:240-368)
  int a = 1;
  const std::string first = argv[1];
  const bool from_trex = first.size() > 5 && first.substr(first.size() - 5) == ".trex";
  smkfiles::LoadedVolume loaded;
  if (from_trex) {
    std::string err;
    smkfiles::TrexHeader peek;
    if (smkfiles::parse_trex(argv[1], &peek, &err) != 1 || !smkfiles::load_trex(argv[1], peek.tstart, &loaded, &err)) {
      fprintf(stderr, "%s\n", err.c_str());
      return 5;
    }
  }
  auto vol = from_trex ? std::vector<unsigned char>() : slurp(argv[a]);
  ++a;
  int nx = atoi(argv[a++]), ny = atoi(argv[a++]), nz = atoi(argv[a++]), ne = atoi(argv[a++]);
  auto grad = slurp(argv[a++]);
  auto dep = slurp(argv[a++]);
  gluvv.win.width = atoi(argv[a++]);
  gluvv.win.height = atoi(argv[a++]);
  gluvv.volren.sampleRate = (float)atof(argv[a++]);
  gluvv.shade = (gluvvShade)atoi(argv[a++]);
  for (int i = 0; i < 16; ++i) gluvv.rinfo.xform[i] = (float)atof(argv[a++]);
  const char *out = argv[a++];
  std::vector<unsigned char> dep2;
  int dmode_arg = -1;
  bool have_sub = false, have_slice = false;  // the inner interface: renderVolume(.., xext, yext, zext), renderSlice(quad, alpha)
  float sub[6] = {0, 0, 0, 0, 0, 0}, quad[4][3] = {{0}}, slice_alpha = 1.f;
  for (; a < argc; ++a) {  // optional state a GUI session would have set
    const std::string kv = argv[a];
    const size_t eq = kv.find('=');
    if (eq == std::string::npos) continue;
    const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
    if (k == "dmode") dmode_arg = atoi(v.c_str());
    else if (k == "plat") gluvv.plat = (gluvvPlatform)atoi(v.c_str());
    else if (k == "tfsize") sscanf(v.c_str(), "%d,%d,%d", &gluvv.tf.ptexsz[0], &gluvv.tf.ptexsz[1], &gluvv.tf.ptexsz[2]);
    else if (k == "deptex2") dep2 = slurp(v.c_str());
    else if (k == "light") sscanf(v.c_str(), "%f,%f,%f", &gluvv.light.pos[0], &gluvv.light.pos[1], &gluvv.light.pos[2]);
    else if (k == "shadow") {  // shadow=<buffer>,<good quality>: the GUI's shadow check box and quality spinners (gluvvui.cpp:150-167)
      gluvv.light.shadow = 1;
      sscanf(v.c_str(), "%d,%f", &gluvv.light.buffsz[0], &gluvv.light.gShadowQual);
      gluvv.light.buffsz[1] = gluvv.light.buffsz[0];
    }
    else if (k == "subbox") { have_sub = true; sscanf(v.c_str(), "%f,%f,%f,%f,%f,%f", &sub[0], &sub[1], &sub[2], &sub[3], &sub[4], &sub[5]); }
    else if (k == "slice") {
      have_slice = true;
      sscanf(v.c_str(), "%f,%f,%f,%f,%f,%f,%f,%f,%f,%f,%f,%f,%f", &slice_alpha, &quad[0][0], &quad[0][1], &quad[0][2], &quad[1][0], &quad[1][1], &quad[1][2],
             &quad[2][0], &quad[2][1], &quad[2][2], &quad[3][0], &quad[3][1], &quad[3][2]);
    }
    else if (k == "pert") {
      gluvv.pert.on = 1;
      sscanf(v.c_str(), "%f,%f,%f,%f", &gluvv.pert.weights[0], &gluvv.pert.weights[1], &gluvv.pert.scales[0], &gluvv.pert.scales[1]);
    }
  }
  if (!from_trex && vol.size() != (size_t)nx * ny * nz * ne) {
    fprintf(stderr, "volume size mismatch\n");
    return 2;
  }
  // MetaVolume as the loader leaves it: one brick, largest dimension normalised to 1
  MetaVolume mv;
  Volume v;
  int mx = nx > ny ? (nx > nz ? nx : nz) : (ny > nz ? ny : nz);
  mv.xiSize = v.xiSize = nx; mv.yiSize = v.yiSize = ny; mv.ziSize = v.ziSize = nz;
  mv.xfSize = v.xfSize = nx / (float)mx; mv.yfSize = v.yfSize = ny / (float)mx; mv.zfSize = v.zfSize = nz / (float)mx;
  v.currentData = vol.data();
  v.currentGrad = grad.empty() ? nullptr : grad.data();
  mv.volumes = &v;
  mv.numSubVols = 1;
  mv.nelts = ne;
  gluvv.mv = from_trex ? &loaded.mv : &mv;
  gluvv.dmode = dmode_arg >= 0 ? (gluvvDataMode)dmode_arg : (ne == 1 ? GDM_V1 : GDM_VGH);
  const float fr = 0.5f / 7;
  gluvv.env.frustum[0] = -fr; gluvv.env.frustum[1] = fr; gluvv.env.frustum[2] = -fr; gluvv.env.frustum[3] = fr;
  if (!dep.empty()) gluvv.volren.deptex = dep.data();
  if (!dep2.empty()) gluvv.volren.deptex2 = dep2.data();

  gluvvPrimitive renderables;                 // "Dummy Node" list head (gluvv.cpp:252)
  HipVolumeRenderable *r = new HipVolumeRenderable(0);
  renderables.setNext(r);
  for (gluvvPrimitive *p = renderables.getNext(); p; p = p->getNext()) p->init();   // initRenderables
  if (!r->running()) {
    fprintf(stderr, "renderer did not start (no HIP device?)\n");
    return 3;
  }
  if (ne == 1) {  // VolumeRenderable::init's colour map: here a plain alpha ramp 0 -> .1
    TLUT *t = gluvv.volren.tlut;
    for (int n = 0; n < t->GetSize(); ++n) t->GetRGBA(n)[3] = 0.1f * n / (t->GetSize() - 1);
    gluvv.volren.loadTLUT = 1;
  }
  for (gluvvPrimitive *p = renderables.getNext(); p; p = p->getNext()) p->draw();   // display()
  if (!r->running()) return 4;
  if (have_sub || have_slice) {
    // VolumeRenderer's inner interface, used the way a caller of the reference class would (VolumeRenderer.h:86-123)
    HipVolumeRenderer *vr = r->renderer();
    vr->useBBox(1);
    vr->useBBoxBrackets(0);
    double mvm[16];
    HipVolumeRenderable::modelview(mvm);
    if (have_sub) {
      float xe[2] = {sub[0], sub[1]}, ye[2] = {sub[2], sub[3]}, ze[2] = {sub[4], sub[5]};
      vr->renderVolume(gluvv.volren.sampleRate, mvm, xe, ye, ze);
    }
    if (have_slice) vr->renderSlice(quad, slice_alpha);
    if (!vr->ok()) return 5;
  }
  FILE *f = fopen(out, "wb");
  fwrite(r->framebuffer(), 4, (size_t)gluvv.win.width * gluvv.win.height * 4, f);
  fclose(f);
  delete r;
  return 0;
}
