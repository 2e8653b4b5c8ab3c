// nfx_stamp.cpp -- nfx_build_stamp(): the hash of the sources this binary was built from (csrc/stamp.py), handed in by the Makefile.
// This object depends on every source file, so it is recompiled -- with the new hash -- whenever any of them changes.
#include "../../include/nfx.h"

#ifndef NFX_CSRC_STAMP
#define NFX_CSRC_STAMP "unstamped"
#endif

extern "C" const char *nfx_build_stamp(void) { return NFX_CSRC_STAMP; }
