// celerite_amd/csrc/sharded.cpp -- the batch axis over several GPUs (BASELINE
// config 4).  Problems are independent (all state of the reference solver is per object,
// cholesky.h:703-706), so a sharded plan is nothing but S single-device plans (clr_batch_*)
// over contiguous slices of the batch axis: no collective, no device-to-device traffic.
//
// Kernel selection is resolved ONCE for the whole batch: the maxima the single-device plans look at
// (max |t|, largest time step, largest decay rate and frequency) are taken over all problems and handed
// to every shard (clr_batch_set_selection_bounds), and all shards use the first shard's chunk count, so
// a batch gives bit-identical results under any sharding with the default settings -- as long as the
// warm-started recurrence (series that forget their past) does not come into play.  That path is
// ADAPTIVE per plan: it is switched on when at least half of the plan's problems are eligible, its chunking looks at
// the plan's batch size and its warm-up lengths grow with the plan's own history of fallbacks.  A batch in which
// about half of the problems forget can therefore take the warm recurrence in one sharding and the scan in another:
// two certified evaluations of the same numbers, equal to the rounding of the scan (<= 1e-11 relative, statuses
// identical; tests/test_gpu_batch.py::test_sharding_a_batch_with_mixed_warm_eligibility), not bit for bit.
// clr_batch_set_warm_start(plan, 0, 0) on every shard (ShardedBatchedGP.set_warm_start(0)) restores bit-identity.
//
// Each shard has its own host worker thread (which owns the shard's HIP device binding,
// stream and pinned staging through its clr_batch handle); an API call posts one job to every
// worker and waits for all of them, so the shards' uploads, launches and downloads overlap.
// A device may be listed more than once: the shards then share that GPU.  That is how the
// sharding is tested on a one-GPU box (tests/test_gpu_batch.py: 1, 2, 3, 8 shards must give
// bit-identical results).
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/celerite_hip.h"

namespace {

struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = true, quit = false;
  int status = CLR_OK;
  std::string error;

  void loop() {
    for (;;) {
      std::function<int()> j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return has_job || quit; });
        if (quit) return;
        j = std::move(job);
        has_job = false;
      }
      const int st = j();
      std::string msg;
      if (st != CLR_OK) msg = clr_last_error();  // (thread-local in the library: read it here)
      {
        std::lock_guard<std::mutex> lk(mu);
        status = st;
        error = std::move(msg);
        done = true;
      }
      cv.notify_all();
    }
  }
  void post(std::function<int()> j) {
    {
      std::lock_guard<std::mutex> lk(mu);
      job = std::move(j);
      has_job = true;
      done = false;
    }
    cv.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
    return status;
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

thread_local std::string g_sharded_error;

}  // namespace

struct clr_sharded {
  int B = 0, N = 0, J_real = 0, J_comp = 0;
  std::vector<int> device, lo, hi;
  std::vector<clr_batch*> plan;
  std::vector<Worker*> worker;

  // run f(shard) on every shard's worker; first non-OK status wins
  int all(const std::function<int(int)>& f) {
    const int S = (int)plan.size();
    for (int s = 0; s < S; ++s) worker[s]->post([=] { return f(s); });
    int st = CLR_OK;
    for (int s = 0; s < S; ++s) {
      const int r = worker[s]->wait();
      if (r != CLR_OK && st == CLR_OK) {
        st = r;
        g_sharded_error = "shard " + std::to_string(s) + " (device " + std::to_string(device[s]) +
                          "): " + worker[s]->error;
      }
    }
    return st;
  }
};

extern "C" {

int clr_shard_bounds(int total, int nshards, int shard, int* lo, int* hi) {
  if (total < 0 || nshards < 1 || shard < 0 || shard >= nshards) return CLR_INVALID_ARGUMENT;
  const int base = total / nshards, extra = total % nshards;
  const int l = shard * base + (shard < extra ? shard : extra);
  if (lo) *lo = l;
  if (hi) *hi = l + base + (shard < extra ? 1 : 0);
  return CLR_OK;
}

const char* clr_sharded_last_error(void) { return g_sharded_error.c_str(); }

void clr_sharded_destroy(clr_sharded* h) {
  if (!h) return;
  for (size_t s = 0; s < h->worker.size(); ++s) {
    clr_batch* p = s < h->plan.size() ? h->plan[s] : nullptr;
    if (p) {
      h->worker[s]->post([p] { clr_batch_destroy(p); return (int)CLR_OK; });
      h->worker[s]->wait();
    }
    h->worker[s]->stop();
    delete h->worker[s];
  }
  delete h;
}

clr_sharded* clr_sharded_create(int B, int N, int J_real, int J_comp, const int* devices, int nshards) {
  g_sharded_error.clear();
  if (B < 1 || N < 1 || nshards < 1 || !devices) {
    g_sharded_error = "clr_sharded_create: bad sizes";
    return nullptr;
  }
  if (nshards > B) nshards = B;  // (no empty shards)
  const int ndev = clr_device_count();
  if (ndev < 1) {
    g_sharded_error = "no gfx950 (MI355X) device is visible; libcelerite_hip has no CPU path";
    return nullptr;
  }
  for (int s = 0; s < nshards; ++s)
    if (devices[s] < 0 || devices[s] >= ndev) {
      g_sharded_error = "clr_sharded_create: device index " + std::to_string(devices[s]) + " out of range (" +
                        std::to_string(ndev) + " visible)";
      return nullptr;
    }
  clr_sharded* h = new clr_sharded();
  h->B = B; h->N = N; h->J_real = J_real; h->J_comp = J_comp;
  h->plan.assign(nshards, nullptr);
  for (int s = 0; s < nshards; ++s) {
    int lo = 0, hi = 0;
    clr_shard_bounds(B, nshards, s, &lo, &hi);
    h->device.push_back(devices[s]);
    h->lo.push_back(lo);
    h->hi.push_back(hi);
    Worker* w = new Worker();
    w->th = std::thread([w] { w->loop(); });
    h->worker.push_back(w);
  }
  const int st = h->all([h, N, J_real, J_comp](int s) {
    h->plan[s] = clr_batch_create(h->hi[s] - h->lo[s], N, J_real, J_comp, h->device[s]);
    return h->plan[s] ? (int)CLR_OK : (int)CLR_HIP_ERROR;
  });
  if (st != CLR_OK) {
    const std::string keep = g_sharded_error;
    clr_sharded_destroy(h);
    g_sharded_error = keep;
    return nullptr;
  }
  // one chunk count for all shards (the automatic choice looks at the shard's own batch size, which differs by
  // one between shards when B is not a multiple of the shard count)
  int nchunk = 0;
  clr_batch_get_chunks(h->plan[0], &nchunk, nullptr);
  if (nshards > 1) (void)clr_sharded_set_chunks(h, nchunk);
  return h;
}

int clr_sharded_num_shards(const clr_sharded* h) { return (int)h->plan.size(); }

int clr_sharded_get_shard(const clr_sharded* h, int shard, int* device, int* lo, int* hi) {
  if (shard < 0 || shard >= (int)h->plan.size()) return CLR_INVALID_ARGUMENT;
  if (device) *device = h->device[shard];
  if (lo) *lo = h->lo[shard];
  if (hi) *hi = h->hi[shard];
  return CLR_OK;
}

int clr_sharded_set_chunks(clr_sharded* h, int nchunk) {
  return h->all([=](int s) { return clr_batch_set_chunks(h->plan[s], nchunk); });
}

int clr_sharded_set_warm_start(clr_sharded* h, int mode, int forced_warmup) {
  return h->all([=](int s) { return clr_batch_set_warm_start(h->plan[s], mode, forced_warmup); });
}

int clr_sharded_set_rescue(clr_sharded* h, int mode) {
  return h->all([=](int s) { return clr_batch_set_rescue(h->plan[s], mode); });
}

int clr_sharded_get_rescue(const clr_sharded* h, int* last_count) {
  // problems of the last fetched evaluation that were re-planned, summed over the shards (inline replays count negative)
  int total = 0;
  for (clr_batch* p : h->plan) {
    int n = 0;
    const int st = clr_batch_get_rescue(p, &n, nullptr, nullptr, nullptr);
    if (st != CLR_OK) return st;
    total += n < 0 ? -n : n;
  }
  if (last_count) *last_count = total;
  return CLR_OK;
}

int clr_sharded_set_summarize_mode(clr_sharded* h, int mode) {
  return h->all([=](int s) { return clr_batch_set_summarize_mode(h->plan[s], mode); });
}

int clr_sharded_get_chunks(const clr_sharded* h, int shard, int* nchunk, int* chunk_len) {
  if (shard < 0 || shard >= (int)h->plan.size()) return CLR_INVALID_ARGUMENT;
  return clr_batch_get_chunks(h->plan[shard], nchunk, chunk_len);
}

int clr_sharded_set_series(clr_sharded* h, const double* t, long t_stride, const double* diag,
                           long diag_stride, const double* y, long y_stride) {
  int st = h->all([=](int s) {
    const long lo = h->lo[s];
    return clr_batch_set_series(h->plan[s], t + lo * t_stride, t_stride, diag + lo * diag_stride,
                                diag_stride, y + lo * y_stride, y_stride);
  });
  if (st != CLR_OK) return st;
  // the series' maxima over the WHOLE batch (every shard scanned its own slice during the upload)
  double tmax = 0.0, dxmax = 0.0;
  for (clr_batch* p : h->plan) {
    double a = 0.0, b = 0.0;
    clr_batch_get_selection_bounds(p, &a, &b, nullptr, nullptr, nullptr);
    if (!(a <= tmax)) tmax = a;
    if (!(b <= dxmax)) dxmax = b;
  }
  for (clr_batch* p : h->plan) clr_batch_set_selection_bounds(p, tmax, dxmax, -1.0, -1.0);
  return CLR_OK;
}

int clr_sharded_get_series_order(const clr_sharded* h, double* dtmin) {
  // the smallest step over the shards' finite values; a negative one ("not sorted") wins over a NaN time in any
  // shard, and only a batch without a negative step reports NaN (clr_batch_get_series_order has the same rule)
  double m = 1.0 / 0.0;
  bool nan = false;
  for (clr_batch* p : h->plan) {
    double d = 0.0;
    const int st = clr_batch_get_series_order(p, &d);
    if (st != CLR_OK) return st;
    if (d != d) nan = true;
    else if (d < m) m = d;
  }
  if (dtmin) *dtmin = (m < 0.0) ? m : (nan ? 0.0 / 0.0 : m);
  return CLR_OK;
}

int clr_sharded_clear_series(clr_sharded* h) {
  return h->all([=](int s) { return clr_batch_clear_series(h->plan[s]); });
}

// largest |d_comp| and decay rate over the whole batch -> every shard (before it takes its slice)
static void global_coefficient_bounds(clr_sharded* h, const double* c_real, const double* c_comp, const double* d_comp) {
  double dmax = 0.0, cmax = 0.0;
  const size_t nr = (size_t)h->B * h->J_real, nc = (size_t)h->B * h->J_comp;
  for (size_t i = 0; i < nc; ++i) {
    const double m = d_comp[i] < 0 ? -d_comp[i] : d_comp[i], c = c_comp[i] < 0 ? -c_comp[i] : c_comp[i];
    if (!(m <= dmax)) dmax = m;
    if (!(c <= cmax)) cmax = c;
  }
  for (size_t i = 0; i < nr; ++i) {
    const double c = c_real[i] < 0 ? -c_real[i] : c_real[i];
    if (!(c <= cmax)) cmax = c;
  }
  for (clr_batch* p : h->plan) clr_batch_set_selection_bounds(p, -1.0, -1.0, dmax, cmax);
}

int clr_sharded_get_summarize_kernel(const clr_sharded* h, int* kind) {
  if (!kind) return CLR_INVALID_ARGUMENT;
  int k0 = 0;
  int st = clr_batch_get_summarize_kernel(h->plan[0], &k0);
  for (size_t s = 1; s < h->plan.size() && st == CLR_OK; ++s) {
    int k = 0;
    st = clr_batch_get_summarize_kernel(h->plan[s], &k);
    if (k != k0) k0 = -1;  // (cannot happen once series and coefficients went through this layer)
  }
  *kind = k0;
  return st;
}

int clr_sharded_set_coefficients(clr_sharded* h, const double* jitter, const double* a_real,
                                 const double* c_real, const double* a_comp, const double* b_comp,
                                 const double* c_comp, const double* d_comp) {
  const long JR = h->J_real, JC = h->J_comp;
  global_coefficient_bounds(h, c_real, c_comp, d_comp);
  return h->all([=](int s) {
    const long lo = h->lo[s];
    return clr_batch_set_coefficients(h->plan[s], jitter ? jitter + lo : nullptr, a_real + lo * JR, c_real + lo * JR,
                                      a_comp + lo * JC, b_comp + lo * JC, c_comp + lo * JC,
                                      d_comp + lo * JC);
  });
}

int clr_sharded_enqueue(clr_sharded* h) {
  return h->all([=](int s) { return clr_batch_enqueue(h->plan[s], 0); });
}

int clr_sharded_synchronize(clr_sharded* h) {
  return h->all([=](int s) { return clr_batch_synchronize(h->plan[s]); });
}

int clr_sharded_get_results(clr_sharded* h, double* loglike, double* logdet, double* quad, int* status) {
  return h->all([=](int s) {
    const long lo = h->lo[s];
    return clr_batch_get_results(h->plan[s], loglike ? loglike + lo : nullptr, logdet ? logdet + lo : nullptr,
                                 quad ? quad + lo : nullptr, status ? status + lo : nullptr);
  });
}

// value + gradient of every problem at the coefficients in force: clr_batch_grad on every shard concurrently
int clr_sharded_grad(clr_sharded* h, double* value, double* grad, int* status) {
  const long NG = 1 + 2 * (long)h->J_real + 4 * (long)h->J_comp;
  return h->all([=](int s) {
    const long lo = h->lo[s];
    return clr_batch_grad(h->plan[s], value ? value + lo : nullptr, grad ? grad + lo * NG : nullptr,
                          status ? status + lo : nullptr);
  });
}

int clr_sharded_evaluate(clr_sharded* h, const double* jitter, const double* a_real, const double* c_real,
                         const double* a_comp, const double* b_comp, const double* c_comp,
                         const double* d_comp, double* loglike, double* logdet, double* quad, int* status) {
  const long JR = h->J_real, JC = h->J_comp;
  global_coefficient_bounds(h, c_real, c_comp, d_comp);
  return h->all([=](int s) {
    const long lo = h->lo[s];
    clr_batch* p = h->plan[s];
    int st = clr_batch_set_coefficients(p, jitter ? jitter + lo : nullptr, a_real + lo * JR, c_real + lo * JR, a_comp + lo * JC,
                                        b_comp + lo * JC, c_comp + lo * JC, d_comp + lo * JC);
    if (st == CLR_OK) st = clr_batch_enqueue(p, 0);
    if (st == CLR_OK)
      st = clr_batch_get_results(p, loglike ? loglike + lo : nullptr, logdet ? logdet + lo : nullptr,
                                 quad ? quad + lo : nullptr, status ? status + lo : nullptr);
    return st;
  });
}

int clr_sharded_run_timed(clr_sharded* h, int steps, double* shard_ms /* [nshards] or NULL */) {
  return h->all([=](int s) {
    double total = 0.0, k[6];
    const int st = clr_batch_run_timed(h->plan[s], 0, steps, 0, &total, k);
    if (shard_ms) shard_ms[s] = total;
    return st;
  });
}

int clr_batch_log_likelihood_sharded(int B, int N, int J_real, int J_comp, const double* jitter,
                                     const double* a_real, const double* c_real, const double* a_comp,
                                     const double* b_comp, const double* c_comp, const double* d_comp,
                                     const double* t, long t_stride, const double* diag, long diag_stride,
                                     const double* y, long y_stride, double* loglike, double* logdet,
                                     double* quad, int* status, const int* devices, int ndevices) {
  clr_sharded* h = clr_sharded_create(B, N, J_real, J_comp, devices, ndevices);
  if (!h) return clr_device_count() > 0 ? CLR_INVALID_ARGUMENT : CLR_NO_DEVICE;
  int st = clr_sharded_set_series(h, t, t_stride, diag, diag_stride, y, y_stride);
  if (st == CLR_OK)
    st = clr_sharded_evaluate(h, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp, loglike, logdet,
                              quad, status);
  const std::string keep = g_sharded_error;
  clr_sharded_destroy(h);
  g_sharded_error = keep;
  return st;
}

}  // extern "C"
