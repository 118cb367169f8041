// version.hip -- library identity.
#include "common.h"
extern "C" int d3_version(void) {
    D3_CLEAR(); return 100; }
extern "C" const char *d3_arch(void) { return "gfx950"; }
