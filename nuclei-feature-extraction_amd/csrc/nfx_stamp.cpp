// nfx_stamp.cpp -- nfx_build_stamp(): the hash of the sources this binary was built from (csrc/stamp.py), handed in by the Makefile.
// This object depends on every source file, so it is recompiled -- with the new hash -- whenever any of them changes.
#include "../../include/nfx.h"

#ifndef NFX_CSRC_STAMP
#define NFX_CSRC_STAMP "unstamped"
#endif

// "NFX_BUILD_STAMP=<stamp>" lies in the binary as one string: a tool can read the stamp out of the FILE (bench.py: library_is_fresh) without
// mapping a possibly stale library into its process; nfx_build_stamp() returns the part behind the '='
static const char kStampMarker[] = "NFX_BUILD_STAMP=" NFX_CSRC_STAMP;

extern "C" const char *nfx_build_stamp(void) { return kStampMarker + 16; }
