/* shadow/grid/msh/msh_hash_grid.h — put shadow/grid BEFORE lib on the include path; optional: redirects msh_hash_grid_init_3d,
 * msh_hash_grid_term and msh_hash_grid_radius_search to librescan_dropin.so by suppressing the
 * implementation section of lib/msh/msh_hash_grid.h:300-1654. */
#pragma once
#ifdef MSH_HASH_GRID_IMPLEMENTATION
#undef MSH_HASH_GRID_IMPLEMENTATION
#define RESCAN_HIP_SUPPRESSED_HASH_GRID_IMPLEMENTATION 1
#endif
#include_next "msh/msh_hash_grid.h"
