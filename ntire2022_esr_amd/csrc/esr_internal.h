// shared by the translation units of libesr_hip.so (not part of the public ABI)
#pragma once
#include <hip/hip_runtime.h>
#include "esr_hip.h"

void esr_set_err(const char* what, hipError_t e);
int esr_check_launch(const char* what);
static inline int esr_round_up(int v, int m) { return (v + m - 1) / m * m; }
