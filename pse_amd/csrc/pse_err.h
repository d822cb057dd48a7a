// Error state of the C-ABI (include/pse_amd.h: every entry point returns a status, the message is read with
// pse_last_error()).  Defined in pse_host_api.cpp, which holds everything of the C-ABI that needs no device -- so that the
// host-only entry points, the parameter rule and the tridiagonal solver can also be built for the CPU sanitizers
// (python -m pse_amd.build --asan).
#pragma once
#include <string>

#include "../../include/pse_amd.h"
#include "pse_host.h"

namespace pse {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));   // records the message, returns code
std::string &error_text();                                                          // thread-local
void fill_info(const Derived &d, pse_info *o);

}  // namespace pse
