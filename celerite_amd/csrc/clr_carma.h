// celerite_amd/csrc/clr_carma.h -- internal interface of carma.hip (model set-up on the host, filter on the device)
#pragma once
#include <hip/hip_runtime.h>

#include <complex>
#include <string>
#include <vector>

namespace clr {

struct CarmaModel {
  int p = 0, q = 0;
  double sigma = 0.0;
  std::vector<std::complex<double>> arroots, beta, b, V, loglam;  // V row-major p x p
};

// carma.h:54-72,141-165; returns a clr_status, message in `err`
int carma_setup(double log_sigma, int p, const double* ar, int q, const double* ma, CarmaModel& M, std::string& err);
// carma.h:74-139: a_real, c_real, a_comp, b_comp, c_comp, d_comp
void carma_celerite_coeffs(const CarmaModel& M, std::vector<double> out[6]);
// carma.h:221-239; model = [b | V | loglam] as (re, im) pairs, device memory
void launch_carma_filter(int n, int p, const double* model, const double* t, const double* y, const double* yerr,
                         double* out, int* status, hipStream_t s);

}  // namespace clr
