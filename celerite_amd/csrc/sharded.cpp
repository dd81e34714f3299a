// celerite_amd/csrc/sharded.cpp -- the batch axis over several GPUs (BASELINE
// config 4).  Problems are independent (all state of the reference solver is per object,
// cholesky.h:703-706), so a sharded plan is nothing but S single-device plans (clr_batch_*)
// over contiguous slices of the batch axis: no collective, no device-to-device traffic.
//
// EVERY decision a plan takes from a quantity over "its" problems is taken ONCE for the whole batch, so that a batch
// gives bit-identical results under any sharding with the default settings (given one chunk count, see below):
//  * kernel selection: the maxima the single-device plans look at (max |t|, largest time step, largest decay rate and
//    frequency) are taken over all problems and handed to every shard (clr_batch_set_selection_bounds);
//  * the prefix plan's time model, the one-launch path of short narrow problems and the deferral of level-1 problems
//    look at the size of the WHOLE batch (clr_group::set_batch_context);
//  * the warm-started recurrence (series that forget their past) is switched on when at least half
//    of the problems OF THE BATCH are eligible, and its warm-up lengths adapt to the fallbacks OF THE BATCH: the shards'
//    counts are added up here and handed back (reselect_all, resolve_all);
//  * level-1 problems left pending by an evaluation are re-planned as a side plan whose chunk count follows their number
//    IN THE BATCH (or replayed inline when the batch holds too many of them): resolve_all adds the pending counts up
//    between the two halves of every shard's resolve (clr_group_hooks.h).
// Round 5 left the last two per plan (bit-identity only with the warm start and the side plans switched off).
// The chunk count: all shards use the first shard's, i.e. the automatic choice looks at a SHARD's batch size -- what
// fills one GPU -- and may differ from the unsharded plan's; clr_sharded_set_chunks pins it (a pinned count also pins the
// warm path's chunking), and with equal chunk counts the bits are equal.
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
#include "clr_group_hooks.h"

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

namespace {

// the warm path's activation: eligible problems of the whole batch -> every shard (host fields only; the workers idle)
void reselect_all(clr_sharded* h) {
  if (h->plan.size() < 2) return;
  long total = 0;
  for (clr_batch* p : h->plan) total += clr_group::warm_eligible(p);
  for (clr_batch* p : h->plan) clr_group::set_warm_eligible_total(p, total);
}

// the two halves of the shards' resolve around the batch-wide sums (pending problems, warm-eligible problems)
int resolve_finish_all(clr_sharded* h, const std::vector<long>& pend, const std::vector<long>& elig,
                       const std::function<int(int)>& then) {
  long P = 0, E = 0;
  for (long v : pend) P += v;
  for (long v : elig) E += v;
  return h->all([=](int s) {
    const int st = clr_group::resolve_finish(h->plan[s], P, E);
    return (st == CLR_OK && then) ? then(s) : st;
  });
}

int resolve_all(clr_sharded* h) {
  if (h->plan.size() < 2) return CLR_OK;  // (a single shard is the whole batch: its plan resolves by itself)
  bool any = false;
  for (clr_batch* p : h->plan) any = any || clr_group::in_flight(p);
  if (!any) return CLR_OK;
  const size_t S = h->plan.size();
  std::vector<long> pend(S, 0), elig(S, 0);
  long* pp = pend.data();
  long* ee = elig.data();
  const int st = h->all([=](int s) { return clr_group::resolve_begin(h->plan[s], pp + s, ee + s); });
  if (st != CLR_OK) return st;
  return resolve_finish_all(h, pend, elig, nullptr);
}

}  // namespace

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
  for (clr_batch* p : h->plan) clr_group::set_batch_context(p, nshards > 1 ? B : 0);
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
  int st = resolve_all(h);
  if (st != CLR_OK) return st;
  if (nchunk <= 0 && h->plan.size() > 1) {  // automatic: the first shard's choice for all
    clr_batch* p0 = h->plan[0];
    h->worker[0]->post([p0] { return clr_batch_set_chunks(p0, 0); });  // (on the shard's own thread: its device binding)
    if ((st = h->worker[0]->wait()) != CLR_OK) return st;
    clr_batch_get_chunks(p0, &nchunk, nullptr);
  }
  st = h->all([=](int s) { return clr_batch_set_chunks(h->plan[s], nchunk); });
  reselect_all(h);
  return st;
}

int clr_sharded_set_warm_start(clr_sharded* h, int mode, int forced_warmup) {
  int st = resolve_all(h);
  if (st != CLR_OK) return st;
  st = h->all([=](int s) { return clr_batch_set_warm_start(h->plan[s], mode, forced_warmup); });
  reselect_all(h);
  return st;
}

int clr_sharded_set_rescue(clr_sharded* h, int mode) {
  const int st = resolve_all(h);
  if (st != CLR_OK) return st;
  return h->all([=](int s) { return clr_batch_set_rescue(h->plan[s], mode); });
}

int clr_sharded_set_certificate(clr_sharded* h, double max_gamma_over_mu, double max_residual, double max_gamma,
                                double max_gamma_times_error) {
  const int st0 = resolve_all(h);
  if (st0 != CLR_OK) return st0;
  return h->all([=](int s) {
    int st = clr_batch_set_certificate(h->plan[s], max_gamma_over_mu, max_residual);
    if (st == CLR_OK && !(max_gamma < 0.0) && !(max_gamma_times_error < 0.0))
      st = clr_batch_set_certificate_gamma(h->plan[s], max_gamma, max_gamma_times_error);
    return st;
  });
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
  const int st0 = resolve_all(h);
  if (st0 != CLR_OK) return st0;
  return h->all([=](int s) { return clr_batch_set_summarize_mode(h->plan[s], mode); });
}

int clr_sharded_get_chunks(const clr_sharded* h, int shard, int* nchunk, int* chunk_len) {
  if (shard < 0 || shard >= (int)h->plan.size()) return CLR_INVALID_ARGUMENT;
  return clr_batch_get_chunks(h->plan[shard], nchunk, chunk_len);
}

int clr_sharded_set_series(clr_sharded* h, const double* t, long t_stride, const double* diag,
                           long diag_stride, const double* y, long y_stride) {
  int st = resolve_all(h);  // (an evaluation in flight is settled on ITS series, by the counts of the whole batch)
  if (st != CLR_OK) return st;
  st = h->all([=](int s) {
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
  reselect_all(h);
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
  const int st0 = resolve_all(h);
  if (st0 != CLR_OK) return st0;
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
  int st = resolve_all(h);  // (pending problems of an evaluation in flight: at ITS coefficients and selection bounds)
  if (st != CLR_OK) return st;
  global_coefficient_bounds(h, c_real, c_comp, d_comp);
  st = h->all([=](int s) {
    const long lo = h->lo[s];
    return clr_batch_set_coefficients(h->plan[s], jitter ? jitter + lo : nullptr, a_real + lo * JR, c_real + lo * JR,
                                      a_comp + lo * JC, b_comp + lo * JC, c_comp + lo * JC,
                                      d_comp + lo * JC);
  });
  reselect_all(h);
  return st;
}

int clr_sharded_enqueue(clr_sharded* h) {
  return h->all([=](int s) { return clr_batch_enqueue(h->plan[s], 0); });
}

int clr_sharded_synchronize(clr_sharded* h) {
  const int st0 = resolve_all(h);
  if (st0 != CLR_OK) return st0;
  return h->all([=](int s) { return clr_batch_synchronize(h->plan[s]); });
}

int clr_sharded_get_results(clr_sharded* h, double* loglike, double* logdet, double* quad, int* status) {
  const int st0 = resolve_all(h);
  if (st0 != CLR_OK) return st0;
  return h->all([=](int s) {
    const long lo = h->lo[s];
    return clr_batch_get_results(h->plan[s], loglike ? loglike + lo : nullptr, logdet ? logdet + lo : nullptr,
                                 quad ? quad + lo : nullptr, status ? status + lo : nullptr);
  });
}

// value + gradient of every problem at the coefficients in force: clr_batch_grad on every shard concurrently
int clr_sharded_grad(clr_sharded* h, double* value, double* grad, int* status) {
  const long NG = 1 + 2 * (long)h->J_real + 4 * (long)h->J_comp;
  const int st0 = resolve_all(h);
  if (st0 != CLR_OK) return st0;
  return h->all([=](int s) {
    const long lo = h->lo[s];
    return clr_batch_grad(h->plan[s], value ? value + lo : nullptr, grad ? grad + lo * NG : nullptr,
                          status ? status + lo : nullptr);
  });
}

int clr_sharded_evaluate(clr_sharded* h, const double* jitter, const double* a_real, const double* c_real,
                         const double* a_comp, const double* b_comp, const double* c_comp,
                         const double* d_comp, double* loglike, double* logdet, double* quad, int* status) {
  if (h->plan.size() < 2) {  // one shard: the plan is the whole batch, one job
    global_coefficient_bounds(h, c_real, c_comp, d_comp);
    return h->all([=](int s) {
      clr_batch* p = h->plan[s];
      int st = clr_batch_set_coefficients(p, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp);
      if (st == CLR_OK) st = clr_batch_enqueue(p, 0);
      if (st == CLR_OK) st = clr_batch_get_results(p, loglike, logdet, quad, status);
      return st;
    });
  }
  // several shards, three rounds of jobs with the batch-wide sums between them: coefficients in (then the warm path's
  // activation over the whole batch); launches + the first half of the resolve (then the pending / eligible counts of
  // the whole batch); the second half + the results
  int st = clr_sharded_set_coefficients(h, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp);
  if (st != CLR_OK) return st;
  const size_t S = h->plan.size();
  std::vector<long> pend(S, 0), elig(S, 0);
  long* pp = pend.data();
  long* ee = elig.data();
  st = h->all([=](int s) {
    const int e = clr_batch_enqueue(h->plan[s], 0);
    return e != CLR_OK ? e : clr_group::resolve_begin(h->plan[s], pp + s, ee + s);
  });
  if (st != CLR_OK) return st;
  return resolve_finish_all(h, pend, elig, [=](int s) {
    const long lo = h->lo[s];
    return clr_batch_get_results(h->plan[s], loglike ? loglike + lo : nullptr, logdet ? logdet + lo : nullptr,
                                 quad ? quad + lo : nullptr, status ? status + lo : nullptr);
  });
}

// a materialising evaluation of every shard (clr_batch_enqueue(plan, 1)) settled with the batch-wide counts: what
// clr_sharded_solve / _dot_L / _predict read
int clr_sharded_materialize(clr_sharded* h, double* loglike, double* logdet, double* quad, int* status) {
  const int st0 = resolve_all(h);
  if (st0 != CLR_OK) return st0;
  auto results = [=](int s) {
    const long lo = h->lo[s];
    return clr_batch_get_results(h->plan[s], loglike ? loglike + lo : nullptr, logdet ? logdet + lo : nullptr,
                                 quad ? quad + lo : nullptr, status ? status + lo : nullptr);
  };
  if (h->plan.size() < 2)
    return h->all([=](int s) {
      const int e = clr_batch_enqueue(h->plan[s], 1);
      return e != CLR_OK ? e : results(s);
    });
  const size_t S = h->plan.size();
  std::vector<long> pend(S, 0), elig(S, 0);
  long* pp = pend.data();
  long* ee = elig.data();
  const int st = h->all([=](int s) {
    const int e = clr_batch_enqueue(h->plan[s], 1);
    return e != CLR_OK ? e : clr_group::resolve_begin(h->plan[s], pp + s, ee + s);
  });
  if (st != CLR_OK) return st;
  return resolve_finish_all(h, pend, elig, results);
}

// clr_batch_solve / clr_batch_dot_L / clr_batch_predict on every shard concurrently, each on its slice of the host arrays
int clr_sharded_solve(clr_sharded* h, int nrhs, const double* b, double* x) {
  if (nrhs < 1 || !x) return CLR_INVALID_ARGUMENT;
  const int st0 = resolve_all(h);
  if (st0 != CLR_OK) return st0;
  const long per = (long)nrhs * h->N;
  return h->all([=](int s) {
    const long lo = h->lo[s];
    return clr_batch_solve(h->plan[s], nrhs, b ? b + lo * per : nullptr, x + lo * per);
  });
}

int clr_sharded_dot_L(clr_sharded* h, int nrhs, const double* z, double* y) {
  if (nrhs < 1 || !z || !y) return CLR_INVALID_ARGUMENT;
  const int st0 = resolve_all(h);
  if (st0 != CLR_OK) return st0;
  const long per = (long)nrhs * h->N;
  return h->all([=](int s) {
    const long lo = h->lo[s];
    return clr_batch_dot_L(h->plan[s], nrhs, z + lo * per, y + lo * per);
  });
}

int clr_sharded_dot(clr_sharded* h, int nrhs, const double* z, double* y) {
  if (nrhs < 1 || !z || !y) return CLR_INVALID_ARGUMENT;
  const int st0 = resolve_all(h);
  if (st0 != CLR_OK) return st0;
  const long per = (long)nrhs * h->N;
  return h->all([=](int s) {
    const long lo = h->lo[s];
    return clr_batch_dot(h->plan[s], nrhs, z + lo * per, y + lo * per);
  });
}

int clr_sharded_predict(clr_sharded* h, int M, const double* xs, long xs_stride, double* pred) {
  if (M < 0 || (M > 0 && (!xs || !pred)) || (xs_stride != 0 && xs_stride != M)) return CLR_INVALID_ARGUMENT;
  const int st0 = resolve_all(h);
  if (st0 != CLR_OK) return st0;
  return h->all([=](int s) {
    const long lo = h->lo[s];
    return clr_batch_predict(h->plan[s], M, xs + lo * xs_stride, xs_stride, pred + lo * (long)M);
  });
}

int clr_sharded_run_timed(clr_sharded* h, int steps, double* shard_ms /* [nshards] or NULL */) {
  // (a timing tool: every shard times its own steps and settles them by its own counts)
  const int st0 = resolve_all(h);
  if (st0 != CLR_OK) return st0;
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
