// Build identification of libngp_hip.so: build.py passes the digest of every source, this header and the
// compiler flags as NGP_BUILD_ID; _lib.load() compares it with the digest of the files it finds next to it.
#include "../../include/ngp_hip.h"

#ifndef NGP_BUILD_ID
#define NGP_BUILD_ID "unknown"
#endif

extern "C" const char* ngp_build_id(void) { return NGP_BUILD_ID; }
