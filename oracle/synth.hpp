// TEST INFRASTRUCTURE ONLY.  CPU definition of the synthetic inputs of SURVEY.md §8(d) /
// BASELINE.md §2 (the reference has no such generator: its benchmarks use unseeded
// Eigen::Random, benchmarks/dense.cpp:54-55,72).  Counter-based so any problem / element can be
// generated independently on any rank; the device generator (tinyopt_amd/csrc) follows the same
// recipe and tests/ compare the two.
#pragma once
#include <cmath>
#include <cstdint>

namespace oracle {
namespace synth {

constexpr uint64_t kDefaultSeed = 0x71940917ull;

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// stream ids: 0 = A entries, 1 = x*, 2 = noise on b, 3 = x0 perturbation, 4 = sigma, 5 = y
inline uint64_t key(uint64_t seed, uint64_t problem, uint64_t stream) {
  return splitmix64(splitmix64(seed + problem) ^ (stream * 0xD6E8FEB86659FD93ull));
}
// U(-1, 1), 53-bit
inline double u11(uint64_t k, uint64_t idx) {
  return double(splitmix64(k + idx) >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

// DenseRow problem p: A ~ U(-1,1) (m×n row-major, rounded to T), x* ~ U(-1,1)^n,
// b_i = f(a_i.x*) + 1e-3 U(-1,1) with f(t) = t + 0.1 sin t, x0 = x* + 0.5 U(-1,1).
// The planted dot product uses the T-rounded A, accumulated in double in ascending j.
template <typename T>
inline void dense_row_problem(uint64_t seed, uint64_t p, int n, int m, T* A, T* b, T* x0, double* xstar_out) {
  const uint64_t kA = key(seed, p, 0), kx = key(seed, p, 1), kn = key(seed, p, 2), k0 = key(seed, p, 3);
  for (int i = 0; i < m; ++i) {
    double t = 0;
    for (int j = 0; j < n; ++j) {
      const T a = T(u11(kA, uint64_t(i) * n + j));
      if (A) A[size_t(i) * n + j] = a;
      t += double(a) * u11(kx, j);
    }
    if (b) b[i] = T(t + 0.1 * std::sin(t) + 1e-3 * u11(kn, i));
  }
  for (int j = 0; j < n; ++j) {
    const double xs = u11(kx, j);
    if (xstar_out) xstar_out[j] = xs;
    if (x0) x0[j] = T(xs + 0.5 * u11(k0, j));
  }
}

// GaussianPrior problem p: y ~ U(-1,1)^n, sigma ~ U(0.5,1.5), x0 ~ U(-1,1)^n.
template <typename T>
inline void gaussian_prior_problem(uint64_t seed, uint64_t p, int n, T* y, T* sigma, T* x0) {
  const uint64_t ks = key(seed, p, 4), ky = key(seed, p, 5), k0 = key(seed, p, 3);
  for (int j = 0; j < n; ++j) {
    if (y) y[j] = T(u11(ky, j));
    if (sigma) sigma[j] = T(1.0 + 0.5 * u11(ks, j));
    if (x0) x0[j] = T(u11(k0, j));
  }
}

}  // namespace synth
}  // namespace oracle
