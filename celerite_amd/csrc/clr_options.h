// celerite_amd/csrc/clr_options.h -- the library's tuning / cross-check switches (clr_set_option, include/celerite_hip.h).
// ONE table per process; `clr::option(key)` is what every translation unit asks instead of getenv: the value set through
// clr_set_option, else -- only when the process was started with CLR_ALLOW_ENV=1 -- the environment variable of the same
// name, else null.  (Round 5 read 19 environment variables directly: a stray variable silently changed kernel selection.)
#pragma once

namespace clr {
// null when the option is not set; the pointer is valid until the calling thread's next call
const char* option(const char* key);
// the output check of a chunked replay whose end states missed the scanned start states (BatchParams::head_check):
// largest state mismatch it is attempted for (CLR_OUTPUT_CHECK_CAP, default 1e-6; 0: route off) and largest output
// mismatch that settles the problem (CLR_OUTPUT_CHECK_TOL, default 2e-11)
double output_check_cap();
double output_check_tol();
}  // namespace clr
