// VolumeFiles.h -- the reference's on-disk volume formats, read and written without the reference:
//
//   .trex        text description of a (possibly pre-bricked, possibly time-varying) data set
//                (MetaVolume::parse, MetaVolume.cpp:233-627; writers :632-661 and :963-1000)
//   raw bricks   "<Data Set Files>.<TTTT>.<BB>" (or the bare path after "Don't append numbers"),
//                x-fastest, one scalar per voxel in the declared type and endianness, quantised to
//                8 bits by min/max on load (MetaVolume::readVol :709-889, quantize<T>
//                VectorMath.h:1441-1552, _nrrdSwap*Endian :1559-1590)
//   NRRD00.01    the subset MetaVolume::readNrrd / parseNrrd accept (:1006-1105, :1518-1566):
//                raw unsigned char / unsigned short, 3 axes (scalar) or 4 (element axis first);
//                writer = genVGH's (genVGH/main.cpp:418-456)
//
// Everything lands in the same MetaVolume / Volume fields the renderers read (gluvv_compat.h or
// the reference's own MetaVolume.h), so HipVolumeRenderer::createVolume takes it as it is.
// Return convention of the reference loaders: bytes (elements) read, 0 = failure; text on `err`.
#pragma once
#include <string>
#include <vector>

#ifndef SMK_USE_REFERENCE_HEADERS
#include "gluvv_compat.h"
#endif

namespace smkfiles {

enum DataType { T_UCHAR, T_SHORT, T_USHORT, T_INT, T_UINT, T_FLOAT, T_DOUBLE };

struct TrexBrick {
  int isize[3] = {0, 0, 0}, ipos[3] = {0, 0, 0};
  float fsize[3] = {0, 0, 0}, fpos[3] = {0, 0, 0};
};

struct TrexHeader {
  std::string name, native_name, files, tlut_file, bane_file, nrrd_file;
  std::vector<std::string> displays;
  int tsteps = 0, tstart = 0, tstop = 0, tstep_cache = 0;
  int isize[3] = {0, 0, 0};
  float fsize[3] = {0, 0, 0};
  DataType type = T_UCHAR;  // "default data type" (MetaVolume.cpp:246)
  bool big_endian = false;
  bool append_numbers = true;
  int declared_bricks = 0;  // "Number of Sub Volumes"
  std::vector<TrexBrick> bricks;
  std::vector<std::string> warnings;  // what the reference prints on cerr and carries on from
};

// MetaVolume::parse: 1 = ok, 0 = structural error inside a SubVolume block, -1 = cannot open
int parse_trex(const char *filename, TrexHeader *h, std::string *err);

// "<files>.<timestep %04d>.<brick %02d>" or the bare path (MetaVolume.cpp:756-760)
std::string brick_file(const TrexHeader &h, int timestep, int brick);

// 8-bit quantisation by the volume's own min/max: (uchar) affine(min, x, max, 0, 255), evaluated
// in double and truncated (VectorMath.h:70-74, 1441-1552).  A constant volume maps to 0.
void quantize_to_u8(const void *native, DataType t, size_t n, unsigned char *out);

// One brick of one time step as 8-bit voxels (MetaVolume::readVol).  `native_f32`, when given,
// receives the values before quantisation (the reference keeps them in Volume::nativeData).
// Returns the bytes read from the file, 0 on failure.
size_t read_brick(const TrexHeader &h, int timestep, int brick, std::vector<unsigned char> *u8,
                  std::vector<float> *native_f32, std::string *err);

// Owns the voxel arrays a MetaVolume points into.
struct LoadedVolume {
  MetaVolume mv;
  std::vector<Volume> vols;
  std::vector<std::vector<unsigned char>> data;
  TrexHeader header;
};

// parse + readAll(timestep): every brick read and quantised, MetaVolume fields filled
// (MetaVolume.cpp:233-627, 891-899).  Returns total bytes read, 0 on failure.
size_t load_trex(const char *filename, int timestep, LoadedVolume *out, std::string *err);

// MetaVolume::writeAll + Volume::writeVol (:963-1000, :99-126): "<prefix>.trex" and one raw
// 8-bit file per brick; scalar volumes only, as in the reference.  Returns bytes written.
size_t write_trex(const char *prefix, const MetaVolume &mv, bool append_numbers, std::string *err);

struct NrrdVolume {
  int nelts = 1, isize[3] = {0, 0, 0};
  float spacing[3] = {1, 1, 1};
  float fsize[3] = {0, 0, 0};  // spacing * size, normalised so the largest is 1 (:1048-1055)
  DataType type = T_UCHAR;
  std::vector<unsigned char> data;  // [z][y][x][nelts], quantised when the file held shorts
};

// MetaVolume::readNrrd: header lines up to the first blank line, then raw data.  Returns the
// elements read, 0 on failure.
size_t read_nrrd(const char *filename, NrrdVolume *out, std::string *err);

// genVGH's writer: "NRRD00.01", 4 axes, the element axis first, spacings with a NaN for it.
size_t write_nrrd(const char *filename, const unsigned char *data, int nelts, const int isize[3],
                  const float fsize[3], const char *element_label, std::string *err);

}  // namespace smkfiles
