// tf_main.cpp -- drives simian-spacemonkey_amd/host/TransferFunctions.{h,cpp} for tests/test_transfer_functions.py
//   tf_main lev <shape 0..3> <faux 0|1> <sv> <sg> <sh> bx by lx ly rx ry tw th H S L alpha be <in.tex|-> <out.tex>
//   tf_main vgh <sx> <sy> <slider1hi> <out.tex>
//   tf_main session <script>     a TF-window session without the window: one command per line
//        frame sv sg sh dmode faux | volume file sx sy sz nelts | widget shape bx by lx ly rx ry tw th H S L alpha be
//        brush kind on | probe vx vy vz slider | world px py pz tx ty tz scale fx fy fz m0..m15 | paint | drop | clear | regen out.tex
//      "probe" prints: inside cx cy cz v0 v1 v2 ; "world" prints: vx vy vz (%.9g)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "TransferFunctions.h"

static int session(const char *path) {
  std::ifstream in(path);
  if (!in) return 3;
  std::unique_ptr<smktf::TFFrame> fr;
  std::vector<unsigned char> vol;
  int vx = 0, vy = 0, vz = 0, vne = 1;
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string cmd;
    if (!(ss >> cmd)) continue;
    if (cmd == "frame") {
      int sv, sg, sh, dm, faux;
      ss >> sv >> sg >> sh >> dm >> faux;
      fr.reset(new smktf::TFFrame(sv, sg, sh, dm));
      fr->faux_shading = faux != 0;
    } else if (cmd == "volume") {
      std::string f;
      ss >> f >> vx >> vy >> vz >> vne;
      vol.resize((size_t)vx * vy * vz * vne);
      FILE *fp = fopen(f.c_str(), "rb");
      if (!fp || fread(vol.data(), 1, vol.size(), fp) != vol.size()) return 3;
      fclose(fp);
    } else if (cmd == "widget" && fr) {
      smktf::LevWidgetState w;
      int shape;
      float b[2], l[2], r[2], tw, th, H, S, L;
      ss >> shape >> b[0] >> b[1] >> l[0] >> l[1] >> r[0] >> r[1] >> tw >> th >> H >> S >> L >> w.alpha >> w.boundary_emphasis;
      w.type = (smktf::WidgetShape)shape;
      smktf::set_positions(&w, b, l, r, tw, th);
      smktf::hsl_to_rgb(H, S, L, w.color);
      fr->widgets.insert(fr->widgets.begin(), w);
    } else if (cmd == "brush" && fr) {
      int kind, on;
      ss >> kind >> on;
      fr->brush_kind = (smktf::Brush)kind;
      fr->brushon = on != 0;
    } else if (cmd == "probe" && fr) {
      float v[3], slider;
      ss >> v[0] >> v[1] >> v[2] >> slider;
      smktf::ProbeSample s;
      smktf::probe_sample(vol.data(), vne, vx, vy, vz, fr->dmode, v, &s);
      if (fr->brushon || !s.inside) smktf::place_brush(&fr->brush, fr->brush_kind, fr->dmode, s, slider);
      printf("%d %d %d %d %.9g %.9g %.9g\n", (int)s.inside, s.cell[0], s.cell[1], s.cell[2], s.value[0], s.value[1], s.value[2]);
    } else if (cmd == "world") {
      float p[3], t[3], scale, fs[3], m[16], out[3];
      ss >> p[0] >> p[1] >> p[2] >> t[0] >> t[1] >> t[2] >> scale >> fs[0] >> fs[1] >> fs[2];
      for (float &x : m) ss >> x;
      smktf::probe_world_to_volume(p, t, m, scale, fs, out);
      printf("%.9g %.9g %.9g\n", out[0], out[1], out[2]);
    } else if (cmd == "paint" && fr) fr->paint();
    else if (cmd == "drop" && fr) fr->drop();
    else if (cmd == "clear" && fr) fr->clear_paint();
    else if (cmd == "regen" && fr) {
      std::string f;
      ss >> f;
      std::vector<unsigned char> dep(fr->paintex.size()), dep3(fr->paintex.size(), 7);
      fr->regenerate(dep.data(), dep3.data());
      for (unsigned char c : dep3)
        if (c) return 4;  // deptex3 is cleared
      FILE *fp = fopen(f.c_str(), "wb");
      if (!fp) return 3;
      fwrite(dep.data(), 1, dep.size(), fp);
      fclose(fp);
    } else {
      fprintf(stderr, "session: bad line '%s'\n", line.c_str());
      return 2;
    }
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc == 3 && !strcmp(argv[1], "session")) return session(argv[2]);
  if (argc >= 6 && !strcmp(argv[1], "vgh")) {
    const int sx = atoi(argv[2]), sy = atoi(argv[3]);
    std::vector<unsigned char> t((size_t)sx * sy * 4, 0);
    smktf::rasterize_vgh(t.data(), sx, sy, (float)atof(argv[4]));
    FILE *f = fopen(argv[5], "wb");
    if (!f) return 3;
    fwrite(t.data(), 1, t.size(), f);
    fclose(f);
    return 0;
  }
  if (argc == 22 && !strcmp(argv[1], "lev")) {
    smktf::LevWidgetState w;
    int a = 2;
    w.type = (smktf::WidgetShape)atoi(argv[a++]);
    w.faux_shading = atoi(argv[a++]) != 0;
    const int sv = atoi(argv[a++]), sg = atoi(argv[a++]), sh = atoi(argv[a++]);
    float b[2], l[2], r[2];
    b[0] = (float)atof(argv[a++]); b[1] = (float)atof(argv[a++]);
    l[0] = (float)atof(argv[a++]); l[1] = (float)atof(argv[a++]);
    r[0] = (float)atof(argv[a++]); r[1] = (float)atof(argv[a++]);
    const float tw = (float)atof(argv[a++]), th = (float)atof(argv[a++]);
    smktf::set_positions(&w, b, l, r, tw, th);
    const float H = (float)atof(argv[a++]), S = (float)atof(argv[a++]), L = (float)atof(argv[a++]);
    smktf::hsl_to_rgb(H, S, L, w.color);
    w.alpha = (float)atof(argv[a++]);
    w.boundary_emphasis = (float)atof(argv[a++]);
    std::vector<unsigned char> t((size_t)sv * sg * sh * 4, 0);
    if (strcmp(argv[a], "-")) {
      FILE *f = fopen(argv[a], "rb");
      if (!f || fread(t.data(), 1, t.size(), f) != t.size()) return 3;
      fclose(f);
    }
    ++a;
    smktf::rasterize(w, t.data(), sv, sg, sh);
    FILE *f = fopen(argv[a], "wb");
    if (!f) return 3;
    fwrite(t.data(), 1, t.size(), f);
    fclose(f);
    return 0;
  }
  fprintf(stderr, "bad usage (%d args)\n", argc);
  return 2;
}
