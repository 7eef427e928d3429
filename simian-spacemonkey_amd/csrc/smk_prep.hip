// smk_prep.hip -- data preparation on the GPU, filled in below.
#include "smk_internal.h"

extern "C" int smk_make_vgh_device(smk_ctx *c, const void *, smk_dtype, int, int, int, int, void *, void *) {
  if (c) c->err = "smk_make_vgh_device: not implemented yet";
  return 1;
}
extern "C" int smk_normals_vgh_device(smk_ctx *c, const void *, int, int, int, int, int, void *) {
  if (c) c->err = "smk_normals_vgh_device: not implemented yet";
  return 1;
}
extern "C" int smk_synth_volume_device(smk_ctx *c, int, unsigned, int, int, int, void *) {
  if (c) c->err = "smk_synth_volume_device: not implemented yet";
  return 1;
}
