// K3 — LDS-resident pivoted LDL^T for one small dense symmetric system per wavefront (n <= 64).
//
// Replaces tinyopt::SolveLDLT (include/tinyopt/math.h:232-240), i.e. Eigen 3.4's
// `A.selfadjointView<Upper>().ldlt()` + `info()==Success && isPositive()` + `solve(b)`.
// The factorisation follows Eigen's algorithm (diagonal pivoting on max |d_ii|, left-looking
// column update, zero-pivot and sign bookkeeping, pseudo-inverse of D in the solve) so that the
// ACCEPTANCE RULE is identical to the reference: fail iff a pivot is negative after a positive
// one (indefinite), all pivots negative, or a non-zero pivot follows a zero pivot; semi-definite
// systems pass with the zero-pivot components zeroed.  Arithmetic is re-associated for 64 lanes
// (lane i owns row i), so results equal Eigen's to rounding, not bitwise.
//
// LDS image: M[i*LD + j], LD odd (conflict-free row-per-lane access for ds_read_b32/b64).
#pragma once
#include "wave_utils.hpp"

namespace toa {

template <typename T>
struct NumLimits;
template <>
struct NumLimits<float> {
  static __device__ __forceinline__ float min_normal() { return 1.17549435e-38f; }
  static __device__ __forceinline__ float max() { return 3.402823466e+38f; }
};
template <>
struct NumLimits<double> {
  static __device__ __forceinline__ double min_normal() { return 2.2250738585072014e-308; }
  static __device__ __forceinline__ double max() { return 1.7976931348623157e+308; }
};

// Factorises the symmetric matrix held (at least in its lower triangle) in M, in place:
// strict lower = L, diagonal = D.  perm[] receives the composed pivot permutation
// ((P b)[i] = b[perm[i]]).  temp: n scratch elements.  Returns info()==Success && isPositive().
template <typename T>
__device__ __forceinline__ bool ldlt_factor_wave(T* __restrict__ M, const int LD, int* __restrict__ perm,
                                                 T* __restrict__ temp, const int n, const int lane) {
  if (lane < n) perm[lane] = lane;
  int sign = 0;  // 0 ZeroSign, 1 PositiveSemiDef, -1 NegativeSemiDef, 2 Indefinite (Eigen internal::SignMatrix)
  bool ok = true;
  if (n == 1) {
    wave_sync();
    const T d = M[0];
    return !(d < T(0));  // isPositive(): PositiveSemiDef || ZeroSign
  }
  bool found_zero_pivot = false;
  const bool in_n = lane < n;
  for (int k = 0; k < n; ++k) {
    wave_sync();
    // --- pivot: first index of the largest |diagonal| in the trailing corner
    const bool cand = in_n && lane >= k;
    const T a = cand ? fabs(M[lane * LD + lane]) : T(-1);
    const T mx = wave_allreduce_max(a);
    const unsigned long long mask = __ballot(cand && a == mx);
    const int p = mask ? int(__builtin_ctzll(mask)) : k;
    if (p != k) {
      // symmetric row/column interchange k <-> p touching only the lower triangle (Eigen LDLT.h)
      int ia, ib;  // element indices to swap for this lane
      bool doswap = in_n;
      if (lane < k) { ia = k * LD + lane; ib = p * LD + lane; }
      else if (lane == k) { ia = k * LD + k; ib = p * LD + p; }
      else if (lane < p) { ia = lane * LD + k; ib = p * LD + lane; }
      else if (lane > p) { ia = lane * LD + k; ib = lane * LD + p; }
      else { ia = ib = 0; doswap = false; }
      if (doswap) {
        const T va = M[ia], vb = M[ib];
        M[ia] = vb;
        M[ib] = va;
      }
      if (lane == 0) {
        const int t = perm[k];
        perm[k] = perm[p];
        perm[p] = t;
      }
      wave_sync();
    }
    // --- temp[j] = D_j * L_kj ; column k -= L[:, 0:k] * temp
    if (k > 0) {
      if (lane < k) temp[lane] = M[lane * LD + lane] * M[k * LD + lane];
      wave_sync();
      if (cand) {
        const T* row = M + lane * LD;
        T acc = 0;
        for (int j = 0; j < k; ++j) acc += row[j] * temp[j];
        M[lane * LD + k] -= acc;
      }
      wave_sync();
    }
    const T akk = M[k * LD + k];  // broadcast read
    const bool pivot_is_valid = fabs(akk) > T(0);
    if (k == 0 && !pivot_is_valid) {
      // entire diagonal is zero: success iff every off-diagonal entry is zero too
      bool nz = false;
      if (in_n)
        for (int j = 0; j < lane; ++j) nz = nz || (M[lane * LD + j] != T(0));
      return !__any(nz);
    }
    const bool below = in_n && lane > k;
    if (k < n - 1) {
      if (pivot_is_valid) {
        if (below) M[lane * LD + k] /= akk;
      } else {
        const bool nz = below && (M[lane * LD + k] != T(0));
        ok = ok && !__any(nz);
      }
    }
    if (found_zero_pivot && pivot_is_valid) ok = false;
    else if (!pivot_is_valid) found_zero_pivot = true;
    if (sign == 1) { if (akk < T(0)) sign = 2; }
    else if (sign == -1) { if (akk > T(0)) sign = 2; }
    else if (sign == 0) { if (akk > T(0)) sign = 1; else if (akk < T(0)) sign = -1; }
  }
  wave_sync();
  return ok && (sign == 1 || sign == 0);
}

// x = P^T L^-T D^+ L^-1 P b  (Eigen LDLT::_solve_impl; D^+ zeroes |d| <= numeric_limits::min()).
// b_lane / return value: element `lane` of b / x (lanes >= n: ignored / 0).  vec: n scratch elements.
template <typename T>
__device__ __forceinline__ T ldlt_solve_wave(const T* __restrict__ M, const int LD, const int* __restrict__ perm,
                                             T* __restrict__ vec, const int n, const int lane, const T b_lane) {
  const bool in_n = lane < n;
  wave_sync();
  if (in_n) vec[lane] = b_lane;
  wave_sync();
  T y = in_n ? vec[perm[lane]] : T(0);
  const T* row = M + (in_n ? lane : 0) * LD;
  for (int j = 0; j < n - 1; ++j) {  // L^-1 (unit lower), column sweep
    const T yj = wave_bcast(y, j);
    if (in_n && lane > j) y -= row[j] * yj;
  }
  const T d = in_n ? row[lane] : T(1);
  y = (fabs(d) > NumLimits<T>::min_normal()) ? y / d : T(0);
  for (int j = n - 1; j > 0; --j) {  // L^-T, column sweep of the transpose (row j of L)
    const T xj = wave_bcast(y, j);
    if (lane < j) y -= M[j * LD + lane] * xj;
  }
  wave_sync();
  if (in_n) vec[perm[lane]] = y;
  wave_sync();
  const T x = in_n ? vec[lane] : T(0);
  wave_sync();
  return x;
}

}  // namespace toa
