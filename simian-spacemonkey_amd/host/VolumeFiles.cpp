// VolumeFiles.cpp -- see VolumeFiles.h.  Plain C++17, no GPU, no reference code: the formats are
// restated from the cited lines.  Deliberate differences from the reference's loaders, all on
// the tolerant side: CR/LF line ends are accepted, lines may be longer than 127 characters, the
// closing brace of a SubVolume block may be indented, int/uint volumes ARE quantised (the
// reference reads and byte-swaps them but forgets the quantize call, MetaVolume.cpp:826-848),
// and 8-byte volumes are quantised instead of being read into the 1-byte buffer (:850-868).
#include "VolumeFiles.h"

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace smkfiles {
namespace {

std::string trim(const std::string &s) {  // MetaVolume::wtspc (:1692-1717), plus '\r'
  size_t a = 0, b = s.size();
  while (b > 0 && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\n' || s[b - 1] == '\r')) --b;
  while (a < b && (s[a] == ' ' || s[a] == '\t')) ++a;
  return s.substr(a, b - a);
}

bool read_line(FILE *f, std::string *line) {
  line->clear();
  int c;
  bool any = false;
  while ((c = fgetc(f)) != EOF) {
    any = true;
    if (c == '\n') break;
    line->push_back((char)c);
  }
  return any;
}

// "a, b, c" -> up to three trimmed fields (strtok(NULL, ",") x2 then strtok(NULL, "\n"))
int split3(const std::string &v, std::string out[3]) {
  int n = 0;
  size_t pos = 0;
  while (n < 3) {
    size_t c = n < 2 ? v.find(',', pos) : std::string::npos;
    std::string field = trim(v.substr(pos, c == std::string::npos ? std::string::npos : c - pos));
    if (field.empty()) break;
    out[n++] = field;
    if (c == std::string::npos) break;
    pos = c + 1;
  }
  return n;
}

bool any_of(const std::string &v, const char *a, const char *b, const char *c) { return v == a || v == b || v == c; }

void int3(const std::string &value, const char *what, int out[3], TrexHeader *h) {
  std::string f[3];
  int n = split3(value, f);
  static const char *axis[3] = {"x", "y", "z"};
  for (int a = 0; a < 3; ++a) {
    if (a < n) out[a] = atoi(f[a].c_str());
    else h->warnings.push_back(std::string(what) + " (" + axis[a] + ") not read, syntax error");
  }
}

void float3(const std::string &value, const char *what, float out[3], TrexHeader *h) {
  std::string f[3];
  int n = split3(value, f);
  static const char *axis[3] = {"x", "y", "z"};
  for (int a = 0; a < 3; ++a) {
    if (a < n) out[a] = (float)atof(f[a].c_str());
    else h->warnings.push_back(std::string(what) + " (" + axis[a] + ") not read, syntax error");
  }
}

size_t type_size(DataType t) {
  switch (t) {
    case T_UCHAR: return 1;
    case T_SHORT: case T_USHORT: return 2;
    case T_INT: case T_UINT: case T_FLOAT: return 4;
    default: return 8;
  }
}

void swap_bytes(unsigned char *p, size_t n, size_t width) {  // _nrrdSwapShortEndian / _nrrdSwapWordEndian
  for (size_t i = 0; i < n; ++i, p += width)
    for (size_t a = 0, b = width - 1; a < b; ++a, --b) {
      unsigned char t = p[a];
      p[a] = p[b];
      p[b] = t;
    }
}

template <typename T, typename Acc>
void quantize_typed(const T *in, size_t n, unsigned char *out, Acc lo0, Acc hi0) {
  Acc hi = hi0, lo = lo0;  // the reference's starting values (max <- smallest, min <- largest)
  for (size_t i = 0; i < n; ++i) {
    hi = hi > (Acc)in[i] ? hi : (Acc)in[i];
    lo = lo < (Acc)in[i] ? lo : (Acc)in[i];
  }
  const double i0 = (double)lo, i1 = (double)hi;
  for (size_t i = 0; i < n; ++i) {
    if (i1 == i0) { out[i] = 0; continue; }  // (the reference divides by zero here)
    const double q = (255.0 - 0.0) * ((double)in[i] - i0) / (i1 - i0) + 0.0;  // affine(min, x, max, 0, 255)
    out[i] = (unsigned char)q;
  }
}

}  // namespace

int parse_trex(const char *filename, TrexHeader *h, std::string *err) {
  FILE *f = fopen(filename, "r");
  if (!f) {
    if (err) *err = std::string("MetaVolume() : Could not open '") + filename + "' for reading.";
    return -1;
  }
  *h = TrexHeader();
  std::string line;
  int subv = 0;
  auto fail = [&](const std::string &m) {
    if (err) *err = m;
    fclose(f);
    return 0;
  };
  while (read_line(f, &line)) {
    // key = text before the first of ":{" (strtok(str, ":{\n")), trimmed
    size_t cut = line.find_first_of(":{");
    std::string key = trim(line.substr(0, cut));
    std::string value = cut == std::string::npos ? std::string() : trim(line.substr(cut + 1));
    if (key.empty()) continue;
    if (key == "Data Type") {
      if (value.empty()) h->warnings.push_back("Data Type not read, syntax error");
      else if (any_of(value, "float", "FLOAT", "Float")) h->type = T_FLOAT;
      else if (any_of(value, "double", "DOUBLE", "Double")) h->type = T_DOUBLE;
      else if (any_of(value, "int", "INT", "Int")) h->type = T_INT;
      else if (any_of(value, "uint", "UINT", "UInt")) h->type = T_UINT;
      else if (any_of(value, "short", "SHORT", "Short")) h->type = T_SHORT;
      else if (any_of(value, "ushort", "USHORT", "UShort")) h->type = T_USHORT;
      else if (any_of(value, "uchar", "UCHAR", "Uchar")) h->type = T_UCHAR;
    } else if (any_of(key, "Time Step Cache", "Time step cache", "time step cache")) {
      if (value.empty()) h->warnings.push_back("Time Step Cache not read, syntax error");
      else h->tstep_cache = atoi(value.c_str());
    } else if (any_of(key, "ENDIAN", "Endian", "endian")) {
      if (value.empty()) h->warnings.push_back("Endian not read, syntax error");
      else if (any_of(value, "BIG", "big", "Big")) h->big_endian = true;
      else if (any_of(value, "LITTLE", "little", "Little")) h->big_endian = false;
    } else if (key == "Displays") {
      size_t pos = 0;
      while (pos <= value.size()) {
        size_t c = value.find(',', pos);
        std::string d = trim(value.substr(pos, c == std::string::npos ? std::string::npos : c - pos));
        if (!d.empty()) h->displays.push_back(d);
        if (c == std::string::npos) break;
        pos = c + 1;
      }
    } else if (key == "Don't append numbers") {
      h->append_numbers = false;
    } else if (key == "Data Set Name") {
      if (value.empty()) h->warnings.push_back("Data Set Name not read, syntax error");
      else h->name = value;
    } else if (key == "Native Data Set Name") {
      if (value.empty()) h->warnings.push_back("Native Data Set Name not read, syntax error");
      else h->native_name = value;
    } else if (key == "Data Set Files") {
      if (value.empty()) h->warnings.push_back("Data Set Files not read, syntax error");
      else h->files = value;
    } else if (key == "Number of Time Steps") {
      int t[3] = {h->tsteps, h->tstart, h->tstop};
      std::string fld[3];
      int n = split3(value, fld);
      static const char *what[3] = {"Number of Time Steps", "Time Step Start", "Time Step Stop"};
      for (int a = 0; a < 3; ++a) {
        if (a < n) t[a] = atoi(fld[a].c_str());
        else h->warnings.push_back(std::string(what[a]) + " not read, syntax error");
      }
      h->tsteps = t[0]; h->tstart = t[1]; h->tstop = t[2];
    } else if (key == "TLUT File") {
      if (value.empty()) h->warnings.push_back("TLUT file not read, syntax error");
      else h->tlut_file = value;
    } else if (key == "Bane File") {
      if (value.empty()) h->warnings.push_back("Bane file not read, syntax error");
      else h->bane_file = value;
    } else if (key == "Nrrd File") {
      if (value.empty()) h->warnings.push_back("Nrrd file not read, syntax error");
      else h->nrrd_file = value;
    } else if (key == "Volume Size int") {
      int3(value, "Volume size int", h->isize, h);
    } else if (key == "Volume Size float") {
      float3(value, "Volume size float", h->fsize, h);
    } else if (key == "Number of Sub Volumes") {
      if (value.empty()) h->warnings.push_back("Number of Sub Volumes not read, syntax error");
      else {
        h->declared_bricks = atoi(value.c_str());
        if (h->declared_bricks < 0 || h->declared_bricks > 100000) return fail("Number of Sub Volumes out of range");
        h->bricks.assign((size_t)h->declared_bricks, TrexBrick());
      }
    } else if (key == "SubVolume") {
      if (h->bricks.empty()) return fail("Error: Number of subvolumes not known");
      if (subv >= (int)h->bricks.size()) return fail("more SubVolume blocks than 'Number of Sub Volumes' declares");
      TrexBrick &b = h->bricks[(size_t)subv];
      bool closed = false;
      while (read_line(f, &line)) {
        if (trim(line) == "}") { closed = true; break; }
        size_t c2 = line.find(':');
        std::string k2 = trim(line.substr(0, c2));
        std::string v2 = c2 == std::string::npos ? std::string() : trim(line.substr(c2 + 1));
        if (k2 == "Size int") int3(v2, "Size int", b.isize, h);
        else if (k2 == "Size float") float3(v2, "Size float", b.fsize, h);
        else if (k2 == "Pos int") int3(v2, "Pos int", b.ipos, h);
        else if (k2 == "Pos float") float3(v2, "Pos int", b.fpos, h);
        else if (!(k2.empty() || line[0] == '#'))
          h->warnings.push_back("MetaVolume::parse() : unknown argument : '" + k2 + "'");
      }
      if (!closed) return fail(std::string("Error parsing ") + filename + ": SubVolume{");
      ++subv;
    } else if (line[0] != '#') {
      h->warnings.push_back("MetaVolume::parse() : unknown argument : '" + key + "'");
    }
  }
  fclose(f);
  return 1;
}

std::string brick_file(const TrexHeader &h, int timestep, int brick) {
  if (!h.append_numbers) return h.files;
  char tail[64];
  snprintf(tail, sizeof tail, ".%04d.%02d", timestep, brick);
  return h.files + tail;
}

void quantize_to_u8(const void *native, DataType t, size_t n, unsigned char *out) {
  switch (t) {
    case T_UCHAR: memcpy(out, native, n); break;
    case T_USHORT: quantize_typed<unsigned short, unsigned short>((const unsigned short *)native, n, out, USHRT_MAX, 0); break;
    case T_SHORT: quantize_typed<short, short>((const short *)native, n, out, SHRT_MAX, SHRT_MIN); break;
    case T_INT: quantize_typed<int, int>((const int *)native, n, out, INT_MAX, INT_MIN); break;
    case T_UINT: quantize_typed<unsigned, unsigned>((const unsigned *)native, n, out, UINT_MAX, 0u); break;
    case T_FLOAT: quantize_typed<float, float>((const float *)native, n, out, 10000000000.0f, -10000000000.0f); break;
    case T_DOUBLE: quantize_typed<double, double>((const double *)native, n, out, 1e300, -1e300); break;
  }
}

size_t read_brick(const TrexHeader &h, int timestep, int brick, std::vector<unsigned char> *u8,
                  std::vector<float> *native_f32, std::string *err) {
  if (brick < 0 || brick >= (int)h.bricks.size()) {
    if (err) *err = "read_brick: no such sub-volume";
    return 0;
  }
  const TrexBrick &b = h.bricks[(size_t)brick];
  if (b.isize[0] <= 0 || b.isize[1] <= 0 || b.isize[2] <= 0) {
    if (err) *err = "read_brick: sub-volume has no size";
    return 0;
  }
  const size_t n = (size_t)b.isize[0] * b.isize[1] * b.isize[2];
  const size_t bytes = n * type_size(h.type);
  const std::string file = brick_file(h, timestep, brick);
  FILE *f = fopen(file.c_str(), "rb");
  if (!f) {
    if (err) *err = "Reader::readVolume, failed to open " + file + " for reading";
    return 0;
  }
  std::vector<unsigned char> raw(bytes);
  const size_t got = fread(raw.data(), 1, bytes, f);
  fclose(f);
  if (got != bytes) {
    char m[160];
    snprintf(m, sizeof m, "Reader::readVolume, read failed: n = %zu of %zu", got, bytes);
    if (err) *err = m;
    return 0;
  }
  // files are swapped when they are declared big-endian (the reference assumes a little-endian host)
  if (h.big_endian && type_size(h.type) > 1) swap_bytes(raw.data(), n, type_size(h.type));
  u8->resize(n);
  quantize_to_u8(raw.data(), h.type, n, u8->data());
  if (native_f32) {
    native_f32->resize(n);
    for (size_t i = 0; i < n; ++i) {
      const unsigned char *p = raw.data() + i * type_size(h.type);
      double v = 0;
      switch (h.type) {
        case T_UCHAR: v = *p; break;
        case T_SHORT: { short s; memcpy(&s, p, 2); v = s; } break;
        case T_USHORT: { unsigned short s; memcpy(&s, p, 2); v = s; } break;
        case T_INT: { int s; memcpy(&s, p, 4); v = s; } break;
        case T_UINT: { unsigned s; memcpy(&s, p, 4); v = s; } break;
        case T_FLOAT: { float s; memcpy(&s, p, 4); v = s; } break;
        case T_DOUBLE: memcpy(&v, p, 8); break;
      }
      (*native_f32)[i] = (float)v;
    }
  }
  return got;
}

size_t load_trex(const char *filename, int timestep, LoadedVolume *out, std::string *err) {
  if (parse_trex(filename, &out->header, err) != 1) return 0;
  const TrexHeader &h = out->header;
  if (h.bricks.empty()) {
    if (err) *err = "no sub-volumes declared";
    return 0;
  }
  out->vols.assign(h.bricks.size(), Volume());
  out->data.assign(h.bricks.size(), std::vector<unsigned char>());
  size_t total = 0;
  for (size_t i = 0; i < h.bricks.size(); ++i) {
    size_t n = read_brick(h, timestep, (int)i, &out->data[i], nullptr, err);
    if (!n) return 0;  // MetaVolume::readAll stops at the first failure (:891-899)
    total += n;
    Volume &v = out->vols[i];
    const TrexBrick &b = h.bricks[i];
    v.xiSize = b.isize[0]; v.yiSize = b.isize[1]; v.ziSize = b.isize[2];
    v.xfSize = b.fsize[0]; v.yfSize = b.fsize[1]; v.zfSize = b.fsize[2];
    v.xiPos = b.ipos[0]; v.yiPos = b.ipos[1]; v.ziPos = b.ipos[2];
    v.xfPos = b.fpos[0]; v.yfPos = b.fpos[1]; v.zfPos = b.fpos[2];
    v.currentData = out->data[i].data();
    v.currentGrad = nullptr;
  }
  MetaVolume &mv = out->mv;
  mv.volumes = out->vols.data();
  mv.numSubVols = (int)out->vols.size();
  mv.nelts = 1;
  mv.xiSize = h.isize[0]; mv.yiSize = h.isize[1]; mv.ziSize = h.isize[2];
  mv.xfSize = h.fsize[0]; mv.yfSize = h.fsize[1]; mv.zfSize = h.fsize[2];
  return total;
}

size_t write_trex(const char *prefix, const MetaVolume &mv, bool append_numbers, std::string *err) {
  const std::string name = std::string(prefix) + ".trex";
  FILE *f = fopen(name.c_str(), "w");
  if (!f) {
    if (err) *err = "MetaVolume::writeAll, failed to open " + name + " for writing";
    return 0;
  }
  fprintf(f, "#  Meta Volume  #\n");
  fprintf(f, "\n\n#    Global info\n");
  fprintf(f, "Data Set Name:         %s\n", prefix);
  fprintf(f, "#  NOTE: Data Set Files uses an implicit extension: filename.timestep.subvol\n");
  fprintf(f, "Data Set Files:        %s\n", prefix);
  fprintf(f, "Number of Time Steps:  1, 0, 0\n");
  fprintf(f, "Volume Size int:       %d, %d, %d\n", mv.xiSize, mv.yiSize, mv.ziSize);
  fprintf(f, "Volume Size float:     %f, %f, %f\n", mv.xfSize, mv.yfSize, mv.zfSize);
  if (!append_numbers) fprintf(f, "Don't append numbers\n");  // (the reference's writer cannot say this; its reader needs it)
  fprintf(f, "\n\n#    Subvolume info\n");
  fprintf(f, "Number of Sub Volumes: %d\n", mv.numSubVols);
  size_t total = 0;
  for (int i = 0; i < mv.numSubVols; ++i) {
    const Volume &v = mv.volumes[i];
    fprintf(f, "SubVolume {\n");
    fprintf(f, "\tSize int:        %d, %d, %d\n", v.xiSize, v.yiSize, v.ziSize);
    fprintf(f, "\tSize float:      %f, %f, %f\n", v.xfSize, v.yfSize, v.zfSize);
    fprintf(f, "\tPos int:         %d, %d, %d\n", v.xiPos, v.yiPos, v.ziPos);
    fprintf(f, "\tPos float:       %f, %f, %f\n", v.xfPos, v.yfPos, v.zfPos);
    fprintf(f, "}\n");
    std::string sub = prefix;
    if (append_numbers) {
      char tail[64];
      snprintf(tail, sizeof tail, ".%04d.%02d", 0, i);
      sub += tail;
    }
    FILE *g = fopen(sub.c_str(), "wb");
    if (!g) {
      if (err) *err = "MetaVolume::writeAll, failed to open " + sub + " for writing";
      fclose(f);
      return 0;
    }
    const size_t n = (size_t)v.xiSize * v.yiSize * v.ziSize;
    const size_t w = fwrite(v.currentData, 1, n, g);
    fclose(g);
    if (w != n) {
      if (err) *err = "MetaVolume:writeVol, write failed";
      fclose(f);
      return 0;
    }
    total += w;
  }
  fclose(f);
  return total;
}

size_t read_nrrd(const char *filename, NrrdVolume *out, std::string *err) {
  FILE *f = fopen(filename, "rb");
  if (!f) {
    if (err) *err = std::string("Reader::readNrrd, failed to open ") + filename + " for reading";
    return 0;
  }
  *out = NrrdVolume();
  int dims = 3;
  std::string line;
  auto fail = [&](const std::string &m) {
    if (err) *err = m;
    fclose(f);
    return (size_t)0;
  };
  // header: "key: value" lines up to the first line of length <= 1 (MetaVolume.cpp:1019-1024)
  while (read_line(f, &line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty()) break;
    size_t c = line.find(':');
    const std::string key = line.substr(0, c);
    const std::string value = c == std::string::npos ? std::string() : line.substr(c + 1);
    if (key == "dimension") {
      dims = atoi(value.c_str());
    } else if (key == "sizes") {
      if (dims == 3) {
        if (sscanf(value.c_str(), "%d %d %d", &out->isize[0], &out->isize[1], &out->isize[2]) != 3) return fail("Error parsing nrrd file: sizes");
      } else if (dims == 4) {
        if (sscanf(value.c_str(), "%d %d %d %d", &out->nelts, &out->isize[0], &out->isize[1], &out->isize[2]) != 4) return fail("Error parsing nrrd file: sizes");
      } else {
        return fail("Error parsing nrrd file: incorrect dimension");
      }
    } else if (key == "spacings") {
      if (dims == 3) {
        sscanf(value.c_str(), "%f %f %f", &out->spacing[0], &out->spacing[1], &out->spacing[2]);
      } else if (dims == 4) {
        // the element axis' entry (a NaN token) is skipped, the other three are read (:1545-1549)
        const std::string t = trim(value);
        size_t sp = t.find(' ');
        if (sp != std::string::npos) sscanf(t.c_str() + sp + 1, "%f %f %f", &out->spacing[0], &out->spacing[1], &out->spacing[2]);
      } else {
        return fail("Error parsing nrrd file: incorrect dimension");
      }
    } else if (key == "type") {
      const std::string t = trim(value);
      if (t == "unsigned char") out->type = T_UCHAR;
      else if (t == "unsigned short") out->type = T_USHORT;
      else return fail("Error parsing nrrd file: only unsigned char/short data type supported");
    }
  }
  if (out->isize[0] <= 0 || out->isize[1] <= 0 || out->isize[2] <= 0 || out->nelts <= 0) return fail("Error parsing nrrd file: sizes missing");
  const size_t n = (size_t)out->isize[0] * out->isize[1] * out->isize[2] * out->nelts;
  out->data.resize(n);
  size_t got;
  if (out->type == T_USHORT) {
    std::vector<unsigned short> raw(n);
    got = fread(raw.data(), 2, n, f);
    if (got != n) return fail("Reader::readNrrd, read failed");
    quantize_to_u8(raw.data(), T_USHORT, n, out->data.data());  // ("only supports scalar data": one min/max for all)
  } else {
    got = fread(out->data.data(), 1, n, f);
    if (got != n) return fail("Reader::readNrrd, UC read failed");
  }
  fclose(f);
  float m = 0;
  for (int a = 0; a < 3; ++a) {
    out->fsize[a] = out->spacing[a] * out->isize[a];
    m = out->fsize[a] > m ? out->fsize[a] : m;
  }
  for (int a = 0; a < 3; ++a) out->fsize[a] /= m;
  return got;
}

size_t write_nrrd(const char *filename, const unsigned char *data, int nelts, const int isize[3],
                  const float fsize[3], const char *element_label, std::string *err) {
  FILE *f = fopen(filename, "wb");
  if (!f) {
    if (err) *err = std::string("writeData, failed to open ") + filename + " for writing";
    return 0;
  }
  const size_t n = (size_t)isize[0] * isize[1] * isize[2] * nelts;
  fprintf(f, "NRRD00.01\n");
  fprintf(f, "number: %zu\n", n);
  fprintf(f, "type: unsigned char\n");
  fprintf(f, "dimension: 4\n");
  fprintf(f, "encoding: raw\n");
  fprintf(f, "endian: big\n");
  fprintf(f, "sizes: %d %d %d %d\n", nelts, isize[0], isize[1], isize[2]);
  fprintf(f, "spacings: nan0x7fffffff %f %f %f\n", fsize[0] / isize[0], fsize[1] / isize[1], fsize[2] / isize[2]);
  fprintf(f, "labels: \"%s\" \"x\" \"y\" \"z\"\n", element_label);
  fprintf(f, "\n");
  const size_t w = fwrite(data, 1, n, f);
  fclose(f);
  if (w != n) {
    if (err) *err = "writeData, write failed";
    return 0;
  }
  return w;
}

}  // namespace smkfiles
