// smk_slab_shadow.hip -- the slice-ring kernel's instances for the eye pass of frames with shadows (half-angle slices,
// SHD = true; smk_slab.hip), compiled as their own translation unit: the instances are most of the library's build time.
#define SLAB_PART 2
#include "smk_slab.hip"
