// files_main.cpp -- drives simian-spacemonkey_amd/host/VolumeFiles.{h,cpp} for tests/test_volume_files.py
// the way Simian's main() drives MetaVolume (gluvv.cpp:141-199: parse the .trex, readAll(tstart)).
//
//   files_main parse  <file.trex>                      header fields as key=value lines
//   files_main load   <file.trex> <timestep> <out.u8>  all bricks, quantised, concatenated in brick order
//   files_main write  <prefix> nx ny nz fx fy fz gx gy gz append <in.u8>   brick like MetaVolume::brick(gx,gy,gz), write
//   files_main nrrd-read  <file.nrrd> <out.u8>
//   files_main nrrd-write <file.nrrd> nelts nx ny nz fx fy fz <in.u8>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "VolumeFiles.h"

using namespace smkfiles;

static std::vector<unsigned char> slurp(const char *path) {
  std::vector<unsigned char> v;
  FILE *f = fopen(path, "rb");
  if (!f) return v;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize((size_t)n);
  if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
  fclose(f);
  return v;
}

static int spit(const char *path, const unsigned char *p, size_t n) {
  FILE *f = fopen(path, "wb");
  if (!f) return 1;
  size_t w = fwrite(p, 1, n, f);
  fclose(f);
  return w == n ? 0 : 1;
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  std::string err;
  const std::string cmd = argv[1];
  if (cmd == "parse") {
    TrexHeader h;
    int rc = parse_trex(argv[2], &h, &err);
    if (rc != 1) {
      fprintf(stderr, "%s\n", err.c_str());
      return rc == -1 ? 3 : 4;
    }
    static const char *tn[] = {"uchar", "short", "ushort", "int", "uint", "float", "double"};
    printf("name=%s\nnative_name=%s\nfiles=%s\ntlut=%s\nbane=%s\nnrrd=%s\n", h.name.c_str(), h.native_name.c_str(),
           h.files.c_str(), h.tlut_file.c_str(), h.bane_file.c_str(), h.nrrd_file.c_str());
    printf("tsteps=%d %d %d\ncache=%d\nisize=%d %d %d\nfsize=%.9g %.9g %.9g\n", h.tsteps, h.tstart, h.tstop, h.tstep_cache,
           h.isize[0], h.isize[1], h.isize[2], h.fsize[0], h.fsize[1], h.fsize[2]);
    printf("type=%s\nbig_endian=%d\nappend=%d\nbricks=%zu\n", tn[h.type], h.big_endian ? 1 : 0, h.append_numbers ? 1 : 0, h.bricks.size());
    for (size_t i = 0; i < h.bricks.size(); ++i) {
      const TrexBrick &b = h.bricks[i];
      printf("brick%zu=%d %d %d | %.9g %.9g %.9g | %d %d %d | %.9g %.9g %.9g\n", i, b.isize[0], b.isize[1], b.isize[2], b.fsize[0],
             b.fsize[1], b.fsize[2], b.ipos[0], b.ipos[1], b.ipos[2], b.fpos[0], b.fpos[1], b.fpos[2]);
    }
    for (const std::string &d : h.displays) printf("display=%s\n", d.c_str());
    for (const std::string &w : h.warnings) printf("warning=%s\n", w.c_str());
    printf("file0=%s\n", h.bricks.empty() ? "" : brick_file(h, h.tstart, 0).c_str());
    return 0;
  }
  if (cmd == "load" && argc == 5) {
    LoadedVolume lv;
    size_t n = load_trex(argv[2], atoi(argv[3]), &lv, &err);
    if (!n) {
      fprintf(stderr, "%s\n", err.c_str());
      return 4;
    }
    std::vector<unsigned char> all;
    for (auto &d : lv.data) all.insert(all.end(), d.begin(), d.end());
    printf("bytes=%zu\nbricks=%d\nisize=%d %d %d\nfsize=%.9g %.9g %.9g\n", n, lv.mv.numSubVols, lv.mv.xiSize, lv.mv.yiSize,
           lv.mv.ziSize, lv.mv.xfSize, lv.mv.yfSize, lv.mv.zfSize);
    return spit(argv[4], all.data(), all.size());
  }
  if (cmd == "write" && argc == 14) {
    const int nx = atoi(argv[3]), ny = atoi(argv[4]), nz = atoi(argv[5]);
    const float fs[3] = {(float)atof(argv[6]), (float)atof(argv[7]), (float)atof(argv[8])};
    const int gx = atoi(argv[9]), gy = atoi(argv[10]), gz = atoi(argv[11]);
    const bool append = atoi(argv[12]) != 0;
    std::vector<unsigned char> in = slurp(argv[13]);
    if (in.size() != (size_t)nx * ny * nz) return 5;
    // MetaVolume::brick geometry (MetaVolume.cpp:1394-1417): equal bricks, remainder voxels dropped
    const int bx = nx / gx, by = ny / gy, bz = nz / gz;
    std::vector<Volume> vols((size_t)gx * gy * gz);
    std::vector<std::vector<unsigned char>> data(vols.size());
    for (int i = 0; i < gz; ++i)
      for (int j = 0; j < gy; ++j)
        for (int k = 0; k < gx; ++k) {
          const size_t cv = ((size_t)i * gy + j) * gx + k;
          Volume &v = vols[cv];
          v.xiSize = bx; v.yiSize = by; v.ziSize = bz;
          v.xfSize = fs[0] * (bx / (float)nx); v.yfSize = fs[1] * (by / (float)ny); v.zfSize = fs[2] * (bz / (float)nz);
          v.xfPos = v.xfSize * k; v.yfPos = v.yfSize * j; v.zfPos = v.zfSize * i;
          v.xiPos = bx * k; v.yiPos = by * j; v.ziPos = bz * i;
          data[cv].resize((size_t)bx * by * bz);
          for (int z = 0; z < bz; ++z)
            for (int y = 0; y < by; ++y)
              memcpy(&data[cv][((size_t)z * by + y) * bx], &in[((size_t)(i * bz + z) * ny + (j * by + y)) * nx + k * bx], (size_t)bx);
          v.currentData = data[cv].data();
        }
    MetaVolume mv;
    mv.volumes = vols.data();
    mv.numSubVols = (int)vols.size();
    mv.xiSize = nx; mv.yiSize = ny; mv.ziSize = nz;
    mv.xfSize = fs[0]; mv.yfSize = fs[1]; mv.zfSize = fs[2];
    size_t n = write_trex(argv[2], mv, append, &err);
    if (!n) {
      fprintf(stderr, "%s\n", err.c_str());
      return 4;
    }
    printf("bytes=%zu\n", n);
    return 0;
  }
  if (cmd == "nrrd-read" && argc == 4) {
    NrrdVolume v;
    size_t n = read_nrrd(argv[2], &v, &err);
    if (!n) {
      fprintf(stderr, "%s\n", err.c_str());
      return 4;
    }
    printf("elements=%zu\nnelts=%d\nisize=%d %d %d\nspacing=%.9g %.9g %.9g\nfsize=%.9g %.9g %.9g\ntype=%s\n", n, v.nelts, v.isize[0],
           v.isize[1], v.isize[2], v.spacing[0], v.spacing[1], v.spacing[2], v.fsize[0], v.fsize[1], v.fsize[2],
           v.type == T_USHORT ? "ushort" : "uchar");
    return spit(argv[3], v.data.data(), v.data.size());
  }
  if (cmd == "nrrd-write" && argc == 11) {
    const int nelts = atoi(argv[3]);
    const int is[3] = {atoi(argv[4]), atoi(argv[5]), atoi(argv[6])};
    const float fs[3] = {(float)atof(argv[7]), (float)atof(argv[8]), (float)atof(argv[9])};
    std::vector<unsigned char> in = slurp(argv[10]);
    if (in.size() != (size_t)nelts * is[0] * is[1] * is[2]) return 5;
    size_t n = write_nrrd(argv[2], in.data(), nelts, is, fs, nelts == 3 ? "vgh" : "v", &err);
    if (!n) {
      fprintf(stderr, "%s\n", err.c_str());
      return 4;
    }
    printf("bytes=%zu\n", n);
    return 0;
  }
  return 2;
}
