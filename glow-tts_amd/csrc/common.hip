// Library identification for the C-ABI (include/glowtts_hip.h).
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/glowtts_hip.h"

extern "C" const char* gt_version(void) { return "glowtts_hip 0.1.0 gfx950"; }
