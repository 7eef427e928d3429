// tf_main.cpp -- drives simian-spacemonkey_amd/host/TransferFunctions.{h,cpp} for tests/test_transfer_functions.py
//   tf_main lev <shape 0..3> <faux 0|1> <sv> <sg> <sh> bx by lx ly rx ry tw th H S L alpha be <in.tex|-> <out.tex>
//   tf_main vgh <sx> <sy> <slider1hi> <out.tex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "TransferFunctions.h"

int main(int argc, char **argv) {
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
