// buildid.cpp — radhip_build_id(): a hash of the library's source tree, set by the Makefile at build time
#include "../../include/rad_hip.h"
#ifndef RADHIP_BUILD_ID
#define RADHIP_BUILD_ID "unknown"
#endif
#ifndef RADHIP_TRAVERSE_ID
#define RADHIP_TRAVERSE_ID "unknown"
#endif
extern "C" const char *radhip_build_id(void) { return RADHIP_BUILD_ID; }
extern "C" const char *radhip_traverse_build_id(void) { return RADHIP_TRAVERSE_ID; }
