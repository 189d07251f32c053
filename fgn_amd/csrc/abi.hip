// ABI version of libfgn_hip.so (bumped whenever include/fgn_hip.h changes incompatibly).
#include "common.h"
extern "C" int fgn_abi_version(void) { return 8; }
