// version.cc -- gaml_hip_version(): library version + the hash of the sources it was built from (Makefile: SRC_HASH).
#include "../../include/gaml_hip.h"

#ifndef GAML_SRC_HASH
#define GAML_SRC_HASH "unknown"
#endif

#ifdef GAML_HIP_DEV
#define GAML_FLAVOUR " dev"
#else
#define GAML_FLAVOUR ""
#endif
extern "C" const char* gaml_hip_version(void) { return "gaml_hip 0.3" GAML_FLAVOUR " (gfx950) src " GAML_SRC_HASH; }
