// Build identification of libgsm_hip.so.  GSM_SRC_HASH is the first 16 hex digits of the SHA-256 over the library's
// sources (every csrc/*.hip, csrc/*.h and include/gsm.h, sorted by name) at build time; mcmc_gpu_amd/_lib.py recomputes
// it at load time and refuses (or rebuilds) a binary built from other sources.
#include "../../include/gsm.h"

#ifndef GSM_SRC_HASH
#define GSM_SRC_HASH "unknown"
#endif

extern "C" const char* gsm_version(void) { return "gsm-hip 0.2 gfx950 src:" GSM_SRC_HASH; }
