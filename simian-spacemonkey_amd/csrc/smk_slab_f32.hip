// smk_slab_f32.hip -- the float-voxel instances of the slice-ring kernel (smk_slab.hip), compiled as their own
// translation unit beside the byte-voxel ones: the instances are most of the library's build time.
#define SLAB_PART 1
#include "smk_slab.hip"
