// celerite_amd/csrc/api_misc.hip -- C ABI: library / device entries and the CARMA handle (clr_carma_*).
#include "api_internal.h"

#include <map>
#include <mutex>

thread_local std::string clr_api_last_error;
thread_local int clr_api_device = 0;

// ---- the option table (clr_options.h) ---------------------------------------------------------------------------
namespace {
std::mutex g_option_mutex;
std::map<std::string, std::string>& option_table() {
  static std::map<std::string, std::string> table;
  return table;
}
bool env_allowed() {
  static const bool allowed = [] { const char* e = getenv("CLR_ALLOW_ENV"); return e && e[0] == '1'; }();
  return allowed;
}
}  // namespace

namespace clr {
const char* option(const char* key) {
  thread_local std::string value;
  {
    std::lock_guard<std::mutex> lock(g_option_mutex);
    auto it = option_table().find(key);
    if (it != option_table().end()) {
      value = it->second;
      return value.c_str();
    }
  }
  return env_allowed() ? getenv(key) : nullptr;
}
double output_check_cap() {
  if (const char* e = option("CLR_OUTPUT_CHECK_CAP")) return atof(e);
  return 1e-6;
}
double output_check_tol() {
  if (const char* e = option("CLR_OUTPUT_CHECK_TOL")) return atof(e);
  return 2e-11;
}
}  // namespace clr

namespace {
// The fp64 FMA rate the vector ALUs of THIS device sustain when every SIMD issues v_fma_f64 back to back, and the shader
// clock they do it at: each wave times its own stream of 64 x iters FMAs (8 independent chains) with s_memtime (shader
// cycles) and s_memrealtime (100 MHz).  The datasheet's 78.6 TFLOP/s is 4 cycles per wave-instruction at 2.4 GHz; under
// this load the chip clocks ~1.9 GHz and a SIMD issues one FMA per ~4.45 cycles (profiles/r05c_clock_under_fp64_load.txt).
__global__ void __launch_bounds__(64) fp64_load_kernel(double* out, unsigned long long* rec, int iters, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  unsigned long long c0, r0, c1, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0));
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                   "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1));
  if (threadIdx.x == 0) { rec[2 * blockIdx.x] = c1 - c0; rec[2 * blockIdx.x + 1] = r1 - r0; }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
}  // namespace

extern "C" {

/* ---- library / device ------------------------------------------------------ */
const char* clr_version(void) { return "0.3.0"; }
const char* clr_last_error(void) { return g_last_error.c_str(); }

const char* clr_status_string(int status) {
  switch (status) {
    case CLR_OK: return "ok";
    case CLR_DIMENSION_MISMATCH: return "dimension mismatch";
    case CLR_NOT_POSITIVE_DEFINITE: return "failed to factorize or solve matrix";
    case CLR_NOT_COMPUTED: return "you must call 'compute' first";
    case CLR_NO_DEVICE: return "no gfx950 device available (libcelerite_hip has no CPU path)";
    case CLR_HIP_ERROR: return "HIP runtime error";
    case CLR_INVALID_ARGUMENT: return "invalid argument";
    case CLR_UNSUPPORTED: return "unsupported configuration";
    case CLR_CARMA_INSTABILITY: return "CARMA model encountered an instability";
    default: return "unknown status";
  }
}

int clr_set_option(const char* key, const char* value) {
  if (!key || strncmp(key, "CLR_", 4) != 0) return fail(CLR_INVALID_ARGUMENT, "clr_set_option: keys start with CLR_");
  std::lock_guard<std::mutex> lock(g_option_mutex);
  if (value) option_table()[key] = value;
  else option_table().erase(key);
  return CLR_OK;
}

const char* clr_get_option(const char* key) { return key ? clr::option(key) : nullptr; }

int clr_device_count(void) { return visible_gfx950(); }

int clr_set_device(int device) {
  int st = require_device(device);
  if (st == CLR_OK) g_device = device;
  return st;
}

int clr_get_device(int* device) {
  *device = g_device;
  return CLR_OK;
}

int clr_device_synchronize(void) {
  int st = require_device(g_device);
  if (st != CLR_OK) return st;
  HIP_TRY(hipDeviceSynchronize());
  return CLR_OK;
}

int clr_device_info(char* name, size_t name_len, int* compute_units, size_t* hbm_bytes) {
  int st = require_device(g_device);
  if (st != CLR_OK) return st;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, g_device));
  if (name && name_len) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return CLR_OK;
}

int clr_device_memory(size_t* free_bytes, size_t* total_bytes) {
  int st = require_device(g_device);
  if (st != CLR_OK) return st;
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return CLR_OK;
}

int clr_device_measure_fp64(int waves_per_simd, int iters, double* tflops, double* clock_mhz, double* cycles_per_fma) {
  int st = require_device(g_device);
  if (st != CLR_OK) return st;
  if (waves_per_simd < 1 || waves_per_simd > 8 || iters < 16) return fail(CLR_INVALID_ARGUMENT, "measure_fp64: 1..8 waves per SIMD, iters >= 16");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, g_device));
  const int waves = prop.multiProcessorCount * 4 * waves_per_simd;
  double* out = nullptr;
  unsigned long long* rec = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&out), (size_t)waves * 64 * sizeof(double)));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&rec), (size_t)waves * 2 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  hipLaunchKernelGGL(fp64_load_kernel, dim3(waves), dim3(64), 0, 0, out, rec, iters / 4, 1.0);  // (clocks settle)
  HIP_TRY(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(fp64_load_kernel, dim3(waves), dim3(64), 0, 0, out, rec, iters, 1.0);
  HIP_TRY(hipEventRecord(e1, 0));
  HIP_TRY(hipDeviceSynchronize());
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)waves * 2);
  HIP_TRY(hipMemcpy(h.data(), rec, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> mhz((size_t)waves), cpi((size_t)waves);
  for (int w = 0; w < waves; ++w) {
    mhz[(size_t)w] = (double)h[2 * (size_t)w] / (double)h[2 * (size_t)w + 1] * 100.0;
    // cycles the SIMD spends per FMA it issues: the wave's own cycles per instruction over the waves sharing the SIMD
    cpi[(size_t)w] = (double)h[2 * (size_t)w] / (64.0 * iters) / waves_per_simd;
  }
  std::sort(mhz.begin(), mhz.end());
  std::sort(cpi.begin(), cpi.end());
  if (tflops) *tflops = 2.0 * 64.0 * 64.0 * iters * waves / (ms * 1e-3) / 1e12;
  if (clock_mhz) *clock_mhz = mhz[mhz.size() / 2];
  if (cycles_per_fma) *cycles_per_fma = cpi[cpi.size() / 2];
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(out);
  (void)hipFree(rec);
  return CLR_OK;
}


/* ======================================================================== */
/* CARMASolver (carma.h): host model + one-wave Kalman filter (carma.hip)    */
/* ======================================================================== */
struct clr_carma {
  clr::CarmaModel model;
  int device = 0;
  hipStream_t stream = nullptr;
  DevBuf dmodel, dt, dy, dyerr, dout;  // dout: [ll | status as int bits]
  bool model_resident = false;
};

clr_carma* clr_carma_create(double log_sigma, int p, const double* arparams, int q, const double* maparams,
                            int* status) {
  int st = CLR_OK;
  clr_carma* h = nullptr;
  if (p < 0 || q < 0 || (p > 0 && !arparams) || (q > 0 && !maparams)) {
    st = fail(CLR_INVALID_ARGUMENT, "bad CARMA parameter arrays");
  } else {
    h = new clr_carma();
    std::string err;
    st = clr::carma_setup(log_sigma, p, arparams, q, maparams, h->model, err);
    if (st != CLR_OK) {
      fail(st, err);
      delete h;
      h = nullptr;
    } else {
      h->device = g_device;
    }
  }
  if (status) *status = st;
  return h;
}

void clr_carma_destroy(clr_carma* h) {
  if (!h) return;
  if (h->stream || h->dmodel.p) {
    (void)hipSetDevice(h->device);
    for (DevBuf* b : {&h->dmodel, &h->dt, &h->dy, &h->dyerr, &h->dout}) b->release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
  }
  delete h;
}

int clr_carma_get_celerite_coeffs(const clr_carma* h, int* n_real, int* n_comp, double* a_real, double* c_real,
                                  double* a_comp, double* b_comp, double* c_comp, double* d_comp) {
  std::vector<double> v[6];
  clr::carma_celerite_coeffs(h->model, v);
  if (n_real) *n_real = (int)v[0].size();
  if (n_comp) *n_comp = (int)v[2].size();
  double* dst[6] = {a_real, c_real, a_comp, b_comp, c_comp, d_comp};
  for (int i = 0; i < 6; ++i)
    if (dst[i]) std::copy(v[i].begin(), v[i].end(), dst[i]);
  return CLR_OK;
}

int clr_carma_log_likelihood(clr_carma* h, int n_t, const double* t, int n_y, const double* y, int n_yerr,
                             const double* yerr, double* out) {
  if (n_y != n_t || n_yerr != n_t) return fail(CLR_DIMENSION_MISMATCH, "dimension mismatch");  // carma.h:223
  int st = require_device(h->device);
  if (st != CLR_OK) return st;
  if (!h->stream) HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const int p = h->model.p, n = n_t;
  if (n == 0 || p == 0) {
    // (an empty series: the filter loop does not run; p = 0 cannot happen, q < p)
    *out = -0.5 * (double)n * 1.8378770664093453;
    return CLR_OK;
  }
  if (!h->model_resident) {
    std::vector<double> pk;
    auto push = [&](const std::vector<std::complex<double>>& a) {
      for (const std::complex<double>& c : a) { pk.push_back(c.real()); pk.push_back(c.imag()); }
    };
    push(h->model.b); push(h->model.V); push(h->model.loglam);
    if ((st = upload(h->dmodel, pk.data(), pk.size(), h->stream)) != CLR_OK) return st;
    HIP_TRY(hipStreamSynchronize(h->stream));  // (pk is a local)
    h->model_resident = true;
  }
  if ((st = upload(h->dt, t, (size_t)n, h->stream)) != CLR_OK) return st;
  if ((st = upload(h->dy, y, (size_t)n, h->stream)) != CLR_OK) return st;
  if ((st = upload(h->dyerr, yerr, (size_t)n, h->stream)) != CLR_OK) return st;
  if ((st = h->dout.reserve(2)) != CLR_OK) return st;
  clr::launch_carma_filter(n, p, h->dmodel.p, h->dt.p, h->dy.p, h->dyerr.p, h->dout.p,
                           reinterpret_cast<int*>(h->dout.p + 1), h->stream);
  HIP_TRY(hipGetLastError());
  double host[2] = {0.0, 0.0};
  HIP_TRY(hipMemcpyAsync(host, h->dout.p, sizeof(host), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  int bad = 0;
  memcpy(&bad, &host[1], sizeof(int));
  if (bad) return fail(CLR_CARMA_INSTABILITY, "CARMA model encountered an instability");  // exceptions.h:8-12
  *out = host[0];
  return CLR_OK;
}

}  // extern "C"
