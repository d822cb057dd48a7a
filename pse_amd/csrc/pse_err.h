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
// 0, or PSE_ERR_INVALID with a message: the spreading Gaussian of these parameters leaves the double range over its support on the
// coarsest of the three grid spacings (an override the reference's rule cannot produce) -- shared by pse_create, pse_set_box,
// pse_host_select_params and the sanitizer build's stand-in, so that all agree on which configurations are valid
int gaussian_fits(const Derived &d, double hx, double hy, double hz);

}  // namespace pse
