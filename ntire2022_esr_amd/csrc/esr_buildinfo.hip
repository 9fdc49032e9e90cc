// esr_buildinfo.hip -- esr_source_hash(): which sources this libesr_hip.so was built from (include/esr_hip.h, ABI v9).
// __graft_entry__.build() writes esr_source_hash.gen.h (git-ignored) next to this file before compiling it.
#include "esr_hip.h"
#if __has_include("esr_source_hash.gen.h")
#include "esr_source_hash.gen.h"
#endif
#ifndef ESR_SOURCE_HASH
#define ESR_SOURCE_HASH "unknown"
#endif

extern "C" const char* esr_source_hash(void) { return ESR_SOURCE_HASH; }
