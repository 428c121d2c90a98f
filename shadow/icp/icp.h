/* shadow/icp/icp.h — put shadow/icp BEFORE lib/rs on the include path of the Rescan apps.
 * Keeps every declaration of lib/rs/icp.h and suppresses its implementation section
 * (lib/rs/icp.h:123-553); icp_align / icp_find_corrs / icp_estimate_rigid_xform_pt2pl then
 * resolve to librescan_dropin.so at link time.  See INTEGRATION.md §1. */
#pragma once
#ifdef ICP_IMPLEMENTATION
#undef ICP_IMPLEMENTATION
#define RESCAN_HIP_SUPPRESSED_ICP_IMPLEMENTATION 1
#endif
#include_next "icp.h"
