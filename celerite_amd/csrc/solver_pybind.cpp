// celerite_amd/csrc/solver_pybind.cpp -- the compiled module `celerite_amd.solver`.
//
// Same names, arity, argument order, return shapes and exception types as the
// reference's pybind11 module `celerite.solver` (celerite/solver.cpp:64-664),
// implemented as a thin host-C++ layer over the C ABI of
// include/celerite_hip.h.  No Eigen: NumPy buffers go straight to the ABI.
//
//   status CLR_NOT_POSITIVE_DEFINITE -> solver.LinAlgError   (solver.cpp:87)
//   every other non-zero status      -> RuntimeError(what()) (pybind11's default
//       for std::exception, which the reference's tests rely on:
//       tests/test_celerite.py:97-100, 341-344)
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/celerite_hip.h"
#include "../../include/celerite_hip_debug.h"

namespace py = pybind11;

namespace {

struct linalg_error : public std::exception {
  const char* what() const noexcept override { return "failed to factorize or solve matrix"; }
};

void check(int status) {
  if (status == CLR_OK) return;
  if (status == CLR_NOT_POSITIVE_DEFINITE) throw linalg_error();
  if (status == CLR_DIMENSION_MISMATCH || status == CLR_NOT_COMPUTED || status == CLR_CARMA_INSTABILITY)
    throw std::runtime_error(clr_status_string(status));
  std::string msg = clr_status_string(status);
  const char* detail = clr_last_error();
  if (detail && *detail) msg += std::string(": ") + detail;
  throw std::runtime_error(msg);
}

typedef py::array_t<double, py::array::c_style | py::array::forcecast> darray;

// 1-D view of a vector argument (accepts (n,), (n,1), (1,n) like the Eigen caster).
struct Vec {
  darray a;
  explicit Vec(const darray& in) : a(in) {
    if (a.ndim() > 2 || (a.ndim() == 2 && a.shape(0) != 1 && a.shape(1) != 1 && a.size() != 0))
      throw py::type_error("expected a one-dimensional float array");
  }
  int n() const { return (int)a.size(); }
  const double* p() const { return a.data(); }
};

// Row-major (rows, cols) view of U / V.
struct Mat {
  darray a;
  int rows = 0, cols = 0;
  explicit Mat(const darray& in) : a(in) {
    if (a.ndim() == 2) {
      rows = (int)a.shape(0);
      cols = (int)a.shape(1);
    } else if (a.ndim() == 1) {  // a lone vector is a column (Eigen's convention)
      rows = (int)a.shape(0);
      cols = rows ? 1 : 0;
    } else if (a.ndim() != 0 || a.size() != 0) {
      throw py::type_error("expected a two-dimensional float array");
    }
  }
  const double* p() const { return a.data(); }
};

// (N,) or (N, nrhs) -> column-major copy.
struct Rhs {
  std::vector<double> data;
  int rows = 0, cols = 0;
  explicit Rhs(const darray& in) {
    if (in.ndim() == 1) {
      rows = (int)in.shape(0);
      cols = 1;
      data.assign(in.data(), in.data() + rows);
    } else if (in.ndim() == 2) {
      rows = (int)in.shape(0);
      cols = (int)in.shape(1);
      data.resize((size_t)rows * cols);
      auto r = in.unchecked<2>();
      for (int k = 0; k < cols; ++k)
        for (int n = 0; n < rows; ++n) data[(size_t)k * rows + n] = r(n, k);
    } else {
      throw py::type_error("expected a 1-D or 2-D float array");
    }
  }
};

py::array_t<double> to_numpy_colmajor(const std::vector<double>& v, int rows, int cols) {
  // shape (rows, cols), Fortran-ordered like the Eigen matrices the reference returns
  py::array_t<double> out({(py::ssize_t)rows, (py::ssize_t)cols},
                          {(py::ssize_t)sizeof(double), (py::ssize_t)(sizeof(double) * rows)});
  if (!v.empty()) std::memcpy(out.mutable_data(), v.data(), sizeof(double) * v.size());
  return out;
}

struct Carma {
  clr_carma* h = nullptr;
  Carma(double log_sigma, const Vec& ar, const Vec& ma) {
    int st = CLR_OK;
    h = clr_carma_create(log_sigma, ar.n(), ar.p(), ma.n(), ma.p(), &st);
    if (!h) check(st == CLR_OK ? (int)CLR_INVALID_ARGUMENT : st);
  }
  ~Carma() { clr_carma_destroy(h); }
  Carma(const Carma&) = delete;
  Carma& operator=(const Carma&) = delete;
};

class Solver {
 public:
  Solver() : h_(clr_solver_create()) {}
  ~Solver() { clr_solver_destroy(h_); }
  Solver(const Solver&) = delete;
  Solver& operator=(const Solver&) = delete;

  void reset() {
    clr_solver_destroy(h_);
    h_ = clr_solver_create();
  }
  clr_solver* h() const { return h_; }

 private:
  clr_solver* h_;
};

struct CoeffArgs {
  Vec a_real, c_real, a_comp, b_comp, c_comp, d_comp, A;
  Mat U, V;
  CoeffArgs(const darray& ar, const darray& cr, const darray& ac, const darray& bc,
            const darray& cc, const darray& dc, const darray& A_, const darray& U_,
            const darray& V_)
      : a_real(ar), c_real(cr), a_comp(ac), b_comp(bc), c_comp(cc), d_comp(dc), A(A_), U(U_),
        V(V_) {}
};

}  // namespace

PYBIND11_MODULE(solver, m) {
  m.doc() =
      "Low-level interface to the MI355X (gfx950) implementation of the celerite solver.\n"
      "Drop-in for the reference's compiled module `celerite.solver`; every method of\n"
      "CholeskySolver runs as HIP kernels through libcelerite_hip.so (no CPU path).";

  m.def("get_library_version", []() { return std::string(clr_version()); },
        "The version of the linked library");
  // forward-mode gradients (solver.cpp:246-463) run as one wave per partial on the GPU
  // (csrc/grad_kernels.hip): report it like a reference build with autodiff (solver.cpp:79-85)
  m.def("has_autodiff", []() { return true; },
        "Returns True if the module was compiled with autodiff support");
  m.def("device_count", []() { return clr_device_count(); },
        "Number of visible gfx950 devices");

  py::register_exception<linalg_error>(m, "LinAlgError");

  auto scalar_map = [](double (*fn)(int, const double*, const double*, int, const double*,
                                    const double*, const double*, const double*, double),
                       const darray& ar, const darray& cr, const darray& ac, const darray& bc,
                       const darray& cc, const darray& dc, const darray& x) {
    Vec a_real(ar), c_real(cr), a_comp(ac), b_comp(bc), c_comp(cc), d_comp(dc);
    if (a_real.n() != c_real.n() || a_comp.n() != b_comp.n() || a_comp.n() != c_comp.n() ||
        a_comp.n() != d_comp.n())
      throw std::runtime_error("dimension mismatch");
    std::vector<py::ssize_t> shape(x.shape(), x.shape() + x.ndim());
    py::array_t<double> out(shape);
    const double* in = x.data();
    double* o = out.mutable_data();
    const py::ssize_t n = x.size();
    for (py::ssize_t i = 0; i < n; ++i)
      o[i] = fn(a_real.n(), a_real.p(), c_real.p(), a_comp.n(), a_comp.p(), b_comp.p(),
                c_comp.p(), d_comp.p(), in[i]);
    return out;
  };

  m.def("get_kernel_value",
        [scalar_map](const darray& ar, const darray& cr, const darray& ac, const darray& bc,
                     const darray& cc, const darray& dc, const darray& tau) {
          return scalar_map(&clr_kernel_value, ar, cr, ac, bc, cc, dc, tau);
        },
        "Value of the kernel at the lags `tau` (any shape); solver.cpp:89-105");
  m.def("get_psd_value",
        [scalar_map](const darray& ar, const darray& cr, const darray& ac, const darray& bc,
                     const darray& cc, const darray& dc, const darray& omega) {
          return scalar_map(&clr_psd_value, ar, cr, ac, bc, cc, dc, omega);
        },
        "PSD of the kernel at the angular frequencies `omega`; solver.cpp:127-143");
  m.def("check_coefficients",
        [](const darray& ar, const darray& cr, const darray& ac, const darray& bc,
           const darray& cc, const darray& dc) {
          Vec a_real(ar), c_real(cr), a_comp(ac), b_comp(bc), c_comp(cc), d_comp(dc);
          return clr_check_coefficients(a_real.n(), a_real.p(), c_real.n(), c_real.p(),
                                        a_comp.n(), a_comp.p(), b_comp.n(), b_comp.p(),
                                        c_comp.n(), c_comp.p(), d_comp.n(), d_comp.p()) != 0;
        },
        "Sturm's-theorem check that the PSD is everywhere positive; solver.cpp:165-175");

  // celerite::carma::CARMASolver (carma.h; solver.cpp:200-235): host model algebra + a one-wave Kalman filter
  // on the device (csrc/carma.hip) behind clr_carma_*
  py::class_<Carma>(m, "CARMASolver",
                    "CARMA(p, q) model in carma_pack's parameterisation (reference: celerite.solver.CARMASolver)")
      .def(py::init([](double log_sigma, const darray& ar, const darray& ma) {
        Vec a(ar), b(ma);
        return std::unique_ptr<Carma>(new Carma(log_sigma, a, b));
      }))
      .def("log_likelihood",
           [](Carma& c, const darray& t, const darray& y, const darray& yerr) {
             Vec vt(t), vy(y), ve(yerr);
             double out = 0.0;
             check(clr_carma_log_likelihood(c.h, vt.n(), vt.p(), vy.n(), vy.p(), ve.n(), ve.p(), &out));
             return out;
           },
           "Compute the log likelihood using a Kalman filter (carma.h:221-239)")
      .def("get_celerite_coeffs",
           [](Carma& c) {
             int nr = 0, nc = 0;
             check(clr_carma_get_celerite_coeffs(c.h, &nr, &nc, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
             py::array_t<double> ar(nr), cr(nr), a(nc), b(nc), cc(nc), d(nc);
             check(clr_carma_get_celerite_coeffs(c.h, nullptr, nullptr, ar.mutable_data(), cr.mutable_data(),
                                                 a.mutable_data(), b.mutable_data(), cc.mutable_data(), d.mutable_data()));
             return py::make_tuple(ar, cr, a, b, cc, d);
           },
           "Compute the coefficients of the celerite model for the given CARMA model (carma.h:74-139)");

  py::class_<Solver> cls(m, "CholeskySolver",
                         "Device-resident semiseparable Cholesky factorisation "
                         "(reference: celerite.solver.CholeskySolver)");
  cls.def(py::init<>());

  cls.def("compute",
          [](Solver& s, double jitter, const darray& ar, const darray& cr, const darray& ac,
             const darray& bc, const darray& cc, const darray& dc, const darray& A,
             const darray& U, const darray& V, const darray& x, const darray& diag) {
            CoeffArgs c(ar, cr, ac, bc, cc, dc, A, U, V);
            Vec xs(x), dg(diag);
            int st;
            {
              py::gil_scoped_release nogil;
              st = clr_solver_compute(
                  s.h(), jitter, c.a_real.n(), c.a_real.p(), c.c_real.n(), c.c_real.p(),
                  c.a_comp.n(), c.a_comp.p(), c.b_comp.n(), c.b_comp.p(), c.c_comp.n(),
                  c.c_comp.p(), c.d_comp.n(), c.d_comp.p(), c.A.n(), c.A.p(), c.U.rows, c.U.cols,
                  c.U.p(), c.V.rows, c.V.cols, c.V.p(), xs.n(), xs.p(), dg.n(), dg.p());
            }
            check(st);
          },
          "compute(jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp, A, U, V, x, diag)\n"
          "Factorise the celerite covariance matrix (solver.cpp:467-483).");

  cls.def("solve",
          [](Solver& s, const darray& b) {
            Rhs rhs(b);
            std::vector<double> x((size_t)rhs.rows * rhs.cols);
            int st;
            {
              py::gil_scoped_release nogil;
              st = clr_solver_solve(s.h(), rhs.rows, rhs.cols, rhs.data.data(), x.data());
            }
            check(st);
            return to_numpy_colmajor(x, rhs.rows, rhs.cols);
          },
          "K^-1 b for b of shape (n,) or (n, nrhs); always returns (n, nrhs) (solver.cpp:507-509)");

  cls.def("dot_solve",
          [](Solver& s, const darray& b) {
            Vec v(b);
            double out = 0.0;
            int st;
            {
              py::gil_scoped_release nogil;
              st = clr_solver_dot_solve(s.h(), v.n(), v.p(), &out);
            }
            check(st);
            return out;
          },
          "b^T K^-1 b (solver.cpp:527-529)");

  cls.def("_hint_rhs",
          [](Solver& s, const darray& b) {
            Vec v(b);
            check(clr_solver_hint_rhs(s.h(), v.n(), v.p()));
          },
          "not in the reference: announce the vector of the coming dot_solve so that the next compute folds "
          "its quadratic form into the factorisation pass (clr_solver_hint_rhs)");

  cls.def("_route",
          [](Solver& s) {
            int level = -1, nchunk = 0;
            double residual = 0.0;
            check(clr_solver_debug_route(s.h(), &level, &nchunk, &residual));
            return std::make_tuple(level, nchunk, residual);
          },
          "not in the reference: (level, chunks, residual) of the last compute's chunked flow (clr_solver_debug_route, "
          "include/celerite_hip_debug.h); level -1: the flow was not taken");

  cls.def("dot_L",
          [](Solver& s, const darray& z) {
            Rhs rhs(z);
            std::vector<double> y((size_t)rhs.rows * rhs.cols);
            check(clr_solver_dot_L(s.h(), rhs.rows, rhs.cols, rhs.data.data(), y.data()));
            return to_numpy_colmajor(y, rhs.rows, rhs.cols);
          },
          "L z with K = L L^T (solver.cpp:547-549)");

  cls.def("dot",
          [](Solver& s, double jitter, const darray& ar, const darray& cr, const darray& ac,
             const darray& bc, const darray& cc, const darray& dc, const darray& A,
             const darray& U, const darray& V, const darray& x, const darray& b) {
            CoeffArgs c(ar, cr, ac, bc, cc, dc, A, U, V);
            Vec xs(x);
            Rhs rhs(b);
            std::vector<double> y((size_t)rhs.rows * rhs.cols);
            check(clr_solver_dot(s.h(), jitter, c.a_real.n(), c.a_real.p(), c.c_real.n(),
                                 c.c_real.p(), c.a_comp.n(), c.a_comp.p(), c.b_comp.n(),
                                 c.b_comp.p(), c.c_comp.n(), c.c_comp.p(), c.d_comp.n(),
                                 c.d_comp.p(), c.A.n(), c.A.p(), c.U.rows, c.U.cols, c.U.p(),
                                 c.V.rows, c.V.cols, c.V.p(), xs.n(), xs.p(), rhs.rows, rhs.cols,
                                 rhs.data.data(), y.data()));
            return to_numpy_colmajor(y, rhs.rows, rhs.cols);
          },
          "K b without factorising (solver.cpp:567-581)");

  cls.def("predict",
          [](Solver& s, const darray& y, const darray& x) {
            Vec yv(y), xv(x);
            py::array_t<double> out((py::ssize_t)xv.n());
            check(clr_solver_predict(s.h(), yv.n(), yv.p(), xv.n(), xv.p(), out.mutable_data()));
            return out;
          },
          "Conditional mean at x given y in O(N + M) (solver.cpp:611-615)");

  cls.def("grad_log_likelihood",
          [](Solver& s, double jitter, const darray& ar, const darray& cr, const darray& ac,
             const darray& bc, const darray& cc, const darray& dc, const darray& A,
             const darray& U, const darray& V, const darray& x, const darray& y,
             const darray& diag) {
            CoeffArgs c(ar, cr, ac, bc, cc, dc, A, U, V);
            Vec xs(x), ys(y), dg(diag);
            const int G = 1 + 2 * c.a_real.n() + 4 * c.a_comp.n();
            py::array_t<double> grad((py::ssize_t)G);
            double value = 0.0;
            double* gp = grad.mutable_data();
            int st;
            {
              py::gil_scoped_release nogil;
              st = clr_solver_grad_log_likelihood(
                  s.h(), jitter, c.a_real.n(), c.a_real.p(), c.c_real.n(), c.c_real.p(),
                  c.a_comp.n(), c.a_comp.p(), c.b_comp.n(), c.b_comp.p(), c.c_comp.n(),
                  c.c_comp.p(), c.d_comp.n(), c.d_comp.p(), c.A.n(), c.A.p(), c.U.rows, c.U.cols,
                  c.U.p(), c.V.rows, c.V.cols, c.V.p(), xs.n(), xs.p(), ys.n(), ys.p(), dg.n(),
                  dg.p(), &value, G, gp);
            }
            check(st);
            return py::make_tuple(value, grad);
          },
          "grad_log_likelihood(jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp, A, U, V, x, y, diag)\n"
          "(value, gradient with respect to the jitter and the coefficients), solver.cpp:347-463.");

  cls.def("log_determinant",
          [](Solver& s) {
            double out = 0.0;
            check(clr_solver_log_determinant(s.h(), &out));
            return out;
          },
          "log det K (solver.cpp:620-622)");

  cls.def("computed", [](Solver& s) { return clr_solver_computed(s.h()) != 0; },
          "True once compute() has succeeded (solver.cpp:632-634)");

  // (computed, N, J, log_det, phi[J x (N-1)], u[J x (N-1)], W[J x N], D[N]) as in
  // PicklableCholeskySolver::serialize / deserialize (solver.cpp:36-58, bound at
  // :644-663): the factor is copied out of / back into HBM on demand.
  cls.def(py::pickle(
      [](const Solver& s) -> py::tuple {
        int computed = 0, N = 0, J = 0;
        double log_det = 0.0;
        clr_solver_get_dims(s.h(), &computed, &N, &J, &log_det);
        if (!computed) {
          return py::make_tuple(false, N, J, log_det, to_numpy_colmajor({}, 0, 0),
                                to_numpy_colmajor({}, 0, 0), to_numpy_colmajor({}, 0, 0),
                                py::array_t<double>((py::ssize_t)0));
        }
        const int Nm1 = N - 1;
        std::vector<double> phi((size_t)J * Nm1), u((size_t)J * Nm1), W((size_t)J * N);
        py::array_t<double> D((py::ssize_t)N);
        check(clr_solver_get_state(s.h(), phi.data(), u.data(), W.data(), D.mutable_data()));
        return py::make_tuple(true, N, J, log_det, to_numpy_colmajor(phi, J, Nm1),
                              to_numpy_colmajor(u, J, Nm1), to_numpy_colmajor(W, J, N), D);
      },
      [](py::tuple t) -> std::unique_ptr<Solver> {
        if (t.size() != 8) throw std::runtime_error("Invalid state!");  // solver.cpp:649
        std::unique_ptr<Solver> s(new Solver());
        const bool computed = t[0].cast<bool>();
        const int N = t[1].cast<int>(), J = t[2].cast<int>();
        const double log_det = t[3].cast<double>();
        if (!computed) {
          check(clr_solver_set_state(s->h(), 0, N, J, log_det, nullptr, nullptr, nullptr,
                                     nullptr));
          return s;
        }
        typedef py::array_t<double, py::array::f_style | py::array::forcecast> farray;
        farray phi = t[4].cast<farray>(), u = t[5].cast<farray>(), W = t[6].cast<farray>();
        darray D = t[7].cast<darray>();
        if (phi.size() != (py::ssize_t)J * (N - 1) || u.size() != phi.size() ||
            W.size() != (py::ssize_t)J * N || D.size() != N)
          throw std::runtime_error("Invalid state!");
        check(clr_solver_set_state(s->h(), 1, N, J, log_det, phi.data(), u.data(), W.data(),
                                   D.data()));
        return s;
      }));
}
