// ABI version of librato_saa.so (include/rato_saa.h).
#include "rato_common.h"

extern "C" int rato_abi_version(void) { return RATO_ABI_VERSION; }
