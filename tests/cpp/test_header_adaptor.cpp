// Exercises the header-only C++ adaptor (include/tinyopt_amd/tinyopt.hpp) the way a tinyopt user would:
// build a cost model, call Optimize(x, cost, options), read the Output.  Reads like the reference's
// tests (tests/sqrt2.cpp:30-56: Succeeded && Converged && answer within a margin).  Needs a GPU.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "tinyopt_amd/tinyopt.hpp"

using namespace tinyopt_amd;

static int fails = 0;
#define REQUIRE(c) do { if (!(c)) { std::printf("REQUIRE failed %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

template <typename T>
static void run(int P, int n, int m, double tol) {
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1, 1);
  std::vector<T> A(size_t(P) * m * n), b(size_t(P) * m), x(size_t(P) * n);
  std::vector<double> xs(size_t(P) * n);
  for (auto& v : A) v = T(U(rng));
  for (int p = 0; p < P; ++p) {
    for (int j = 0; j < n; ++j) { xs[p * n + j] = U(rng); x[p * n + j] = T(xs[p * n + j] + 0.5 * U(rng)); }
    for (int i = 0; i < m; ++i) {
      double t = 0;
      for (int j = 0; j < n; ++j) t += double(A[(size_t(p) * m + i) * n + j]) * xs[p * n + j];
      b[size_t(p) * m + i] = T(t + 0.1 * std::sin(t));
    }
  }
  Context ctx(0);
  DenseRow<T> cost(ctx, P, n, m, A.data(), b.data());
  Options options;                       // defaults, as `Optimize(x, loss)` in the reference
  const auto out = Optimize(x, cost, options, /*history=*/true);
  for (int p = 0; p < P; ++p) {
    REQUIRE(out.Succeeded(p));
    REQUIRE(out.num_iters[p] >= 2 && out.num_iters[p] <= options.max_iters + 1);
    for (int j = 0; j < n; ++j) REQUIRE(std::abs(double(x[p * n + j]) - xs[p * n + j]) < tol);
    REQUIRE(out.final_hessian[size_t(p) * n * n] > 0);  // tests/basic.cpp:35 H(0,0) > 0
    REQUIRE(out.errs[size_t(p) * out.hist_stride] > out.final_cost[p]);
  }
  // Accumulate seam at the solution: gradient ~ 0, cost ~ 0
  std::vector<T> g, H;
  std::vector<double> c;
  Accumulate(cost, x, &g, &H, c);
  for (int p = 0; p < P; ++p) REQUIRE(c[p] < 1e-6 * m);
  // misuse -> std::invalid_argument (reference: optimize.h:47,55,75)
  std::vector<T> bad(3);
  bool threw = false;
  try { Optimize(bad, cost, options); } catch (const std::invalid_argument&) { threw = true; }
  REQUIRE(threw);
}

int main() {
  run<double>(5, 12, 200, 1e-7);
  run<float>(3, 50, 600, 2e-3);
  std::printf("test_header_adaptor: %s\n", fails ? "FAILED" : "ok");
  return fails ? 1 : 0;
}
