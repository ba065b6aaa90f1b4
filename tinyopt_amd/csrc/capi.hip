// libtinyopt_amd.so — the C-ABI of include/tinyopt_amd.h: argument checking, dispatch to the
// per-(dtype, block count) kernel instantiations (inst.hip), and the data-format callers either side
// of the path (pack / synth kernels).  gfx950 (MI355X, CDNA4) only.
#include "kernels.hpp"

namespace toa {

// STREAM-like read ceiling ("a measured copy ceiling reported next to the nominal one"): every lane
// streams 16-byte loads, 4 in flight, and folds them into one word that is (practically never) written.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) hbm_read_probe_kernel(const u32x4* __restrict__ src, size_t n16, uint32_t* __restrict__ sink) {
  const size_t stride = size_t(gridDim.x) * 256;
  size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
    const u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n16; i += stride) { const u32x4 a = src[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
  if (acc == 0x9E3779B9u) *sink = acc;
}

// On-die (Infinity Cache) read ceiling: the access pattern of a problem re-read by its own waves — `slices` contiguous slices,
// each streamed start to end by `wps` waves, `passes` times inside one launch (tools/ubench/llc_probe.hip is the standalone form).
__global__ void __launch_bounds__(256) llc_read_probe_kernel(const u32x4* __restrict__ src, size_t per16, int slices, int wps, int passes,
                                                             uint32_t* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const int gw = int((size_t(blockIdx.x) * 256 + threadIdx.x) >> 6);
  const int s = gw / wps, part = gw % wps;
  if (s >= slices) return;
  const size_t lo = per16 * size_t(part) / size_t(wps), hi = per16 * size_t(part + 1) / size_t(wps);
  const u32x4* base = src + size_t(s) * per16;
  uint32_t acc = 0;
  for (int r = 0; r < passes; ++r) {
    size_t i = lo + lane;
    for (; i + 192 < hi; i += 256) {
      const u32x4 a = base[i], b = base[i + 64], c = base[i + 128], d = base[i + 192];
      acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < hi; i += 64) { const u32x4 a = base[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    asm volatile("" : "+v"(acc));
  }
  if (acc == 0x9E3779B9u) *sink = acc;
}

template <typename T>
__global__ void __launch_bounds__(256) robust_norm_kernel(int kind, long long count, const T* __restrict__ n2, T th2,
                                                          T* __restrict__ loss, T* __restrict__ scale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  T l, s;
  robust_norm(kind, n2[i], th2, l, s);
  loss[i] = l;
  scale[i] = s;
}

// ceres::Jet on the device, function by function (tests/test_gpu_jet.py pins every one against closed-form
// derivatives): out[i] = (f, df/da, df/db) of function `fn` at (a[i], b[i]) through Jet<T, 2> seeded on (a, b).
template <typename T>
__global__ void __launch_bounds__(256) jet_eval_kernel(int fn, long long count, const T* __restrict__ a, const T* __restrict__ b,
                                                       T* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  using J = Jet<T, 2>;
  const J x(a[i], 0), y(b[i], 1);
  J r;
  switch (fn) {
    case 0: r = x + y; break;           case 1: r = x - y; break;            case 2: r = x * y; break;
    case 3: r = x / y; break;           case 4: r = abs(x); break;           case 5: r = log(x); break;
    case 6: r = exp(x); break;          case 7: r = sqrt(x); break;          case 8: r = cos(x); break;
    case 9: r = sin(x); break;          case 10: r = tan(x); break;          case 11: r = atan(x); break;
    case 12: r = tanh(x); break;        case 13: r = atan2(x, y); break;     case 14: r = pow(x, 2.5); break;
    case 15: r = acos(x); break;        case 16: r = asin(x); break;         case 17: r = sinh(x); break;
    case 18: r = cosh(x); break;        case 19: r = cbrt(x); break;         case 20: r = exp2(x); break;
    case 21: r = log2(x); break;        case 22: r = log10(x); break;        case 23: r = log1p(x); break;
    case 24: r = expm1(x); break;       case 25: r = hypot(x, y); break;     case 26: r = fmax(x, y); break;
    case 27: r = fmin(x, y); break;     case 28: r = erf(x); break;          case 29: r = erfc(x); break;
    case 30: r = pow(x, y); break;      case 31: r = pow(a[i], y); break;    case 32: r = fma(x, y, x); break;
    case 33: r = fdim(x, y); break;     case 34: r = floor(x); break;        case 35: r = ceil(x); break;
    case 36: r = norm(x); break;        case 37: r = copysign(x, y); break;  case 38: r = T(2) / x + x / T(4) - T(3) * y; break;
    case 39: r = hypot(x, y, x * y); break;
    case 40: r = BesselJ0(x); break;    case 41: r = BesselJ1(x); break;     case 42: r = BesselJn(3, x); break;
    case 43: r = cyl_bessel_j(0, x) + cyl_bessel_j(2, y); break;
    case 44: r = lerp(x, y, x * y); break;                                    case 45: r = midpoint(x, y); break;
    case 46:   // classification / comparison on the scalar part: a bit mask in the value, no derivative
      r = J(T((isfinite(x) ? 1 : 0) | (isinf(x) ? 2 : 0) | (isnan(x) ? 4 : 0) | (isnormal(x) ? 8 : 0) | (signbit(x) ? 16 : 0) |
              (isless(x, y) ? 32 : 0) | (isgreater(x, y) ? 64 : 0) | (islessequal(x, y) ? 128 : 0) | (isgreaterequal(x, y) ? 256 : 0) |
              (islessgreater(x, y) ? 512 : 0) | (isunordered(x, y) ? 1024 : 0) | (fpclassify(x) << 11)));
      break;
    default: r = J(T(NAN)); break;
  }
  out[3 * i] = r.a; out[3 * i + 1] = r.v[0]; out[3 * i + 2] = r.v[1];
}

// Natural (A [P][m][n], b [P][m]) -> packed [P][m4][RS] (layout: DenseRowLayout).
template <typename T>
__global__ void dense_row_pack_kernel(const T* __restrict__ A, const T* __restrict__ b, T* __restrict__ out,
                                      long long P, int n, int m, DenseRowLayout lay) {
  const int RS = lay.rs, m4 = lay.m4;
  const long long total = P * (long long)m4 * RS;
  const int pb = lay.pos_b();
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int q = int(e % RS);
    const long long rowg = e / RS;
    const int i = int(rowg % m4);
    const long long p = rowg / m4;
    T v = 0;
    if (i < m) {
      if (q == pb) v = b[p * m + i];
      else if (q < lay.nmr) v = A[(p * m + i) * n + q];
      else if (q >= lay.rsm && lay.nmr + (q - lay.rsm) < n) v = A[(p * m + i) * n + lay.nmr + (q - lay.rsm)];
    }
    out[e] = v;
  }
}

// ---- synthetic inputs (same recipe as oracle/synth.hpp) ----
__host__ __device__ inline unsigned long long sm64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__host__ __device__ inline unsigned long long skey(unsigned long long seed, unsigned long long p, unsigned long long s) {
  return sm64(sm64(seed + p) ^ (s * 0xD6E8FEB86659FD93ull));
}
__host__ __device__ inline double u11(unsigned long long k, unsigned long long idx) {
  return double(sm64(k + idx) >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

template <typename T>
__global__ void dense_row_synth_kernel(T* __restrict__ out, T* __restrict__ x0, T* __restrict__ xstar,
                                       long long P, int n, int m, DenseRowLayout lay, unsigned long long seed,
                                       long long problem0) {
  const int RS = lay.rs, m4 = lay.m4;
  const long long rows = P * (long long)m4;
  for (long long rg = (long long)blockIdx.x * blockDim.x + threadIdx.x; rg < rows; rg += (long long)gridDim.x * blockDim.x) {
    const long long p = rg / m4;
    const int i = int(rg % m4);
    const unsigned long long pid = (unsigned long long)(problem0 + p);
    T* row = out + rg * RS;
    for (int q = 0; q < RS; ++q) row[q] = T(0);
    if (i < m) {
      const unsigned long long kA = skey(seed, pid, 0), kx = skey(seed, pid, 1), kn = skey(seed, pid, 2);
      double t = 0;
      for (int j = 0; j < n; ++j) {
        const T a = T(u11(kA, (unsigned long long)i * n + j));
        row[lay.pos_col(j)] = a;
        t += double(a) * u11(kx, j);
      }
      row[lay.pos_b()] = T(t + 0.1 * sin(t) + 1e-3 * u11(kn, i));
    }
    if (i == 0) {
      const unsigned long long kx = skey(seed, pid, 1), k0 = skey(seed, pid, 3);
      for (int j = 0; j < n; ++j) {
        const double xs = u11(kx, j);
        if (xstar) xstar[p * n + j] = T(xs);
        if (x0) x0[p * n + j] = T(xs + 0.5 * u11(k0, j));
      }
    }
  }
}

}  // namespace toa

using namespace toa;

int toa_inst_solve_0_0(int npad, toa_handle h, int n, int64_t P, const void* H, const void* g, double scale, void* dx, int32_t* ok);
int toa_inst_fused_0_1(int thin, toa_handle h, const toa::FusedParams& prm);
int toa_inst_accumulate_0_1(int thin, toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_fused_0_2(int thin, toa_handle h, const toa::FusedParams& prm);
int toa_inst_accumulate_0_2(int thin, toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_fused_0_3(int thin, toa_handle h, const toa::FusedParams& prm);
int toa_inst_accumulate_0_3(int thin, toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_fused_0_4(int thin, toa_handle h, const toa::FusedParams& prm);
int toa_inst_accumulate_0_4(int thin, toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_solve_1_0(int npad, toa_handle h, int n, int64_t P, const void* H, const void* g, double scale, void* dx, int32_t* ok);
int toa_inst_fused_1_1(int thin, toa_handle h, const toa::FusedParams& prm);
int toa_inst_accumulate_1_1(int thin, toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_fused_1_2(int thin, toa_handle h, const toa::FusedParams& prm);
int toa_inst_accumulate_1_2(int thin, toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_fused_1_3(int thin, toa_handle h, const toa::FusedParams& prm);
int toa_inst_accumulate_1_3(int thin, toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_fused_1_4(int thin, toa_handle h, const toa::FusedParams& prm);
int toa_inst_accumulate_1_4(int thin, toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);

int toa_inst_misc_fused_0_0(int model, int npad, toa_handle h, const toa::FusedParams& prm);
int toa_inst_misc_fused_1_0(int model, int npad, toa_handle h, const toa::FusedParams& prm);
int toa_inst_misc_accumulate_0_0(int model, int npad, toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_misc_accumulate_1_0(int model, int npad, toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);

int toa_inst_misc_wide_0_0(int model, toa_handle h, const toa::FusedParams& prm, int splits);
int toa_inst_wide_0_1(int thin, toa_handle h, const toa::FusedParams& prm, int splits);
int toa_inst_wide_0_2(int thin, toa_handle h, const toa::FusedParams& prm, int splits);
int toa_inst_wide_0_3(int thin, toa_handle h, const toa::FusedParams& prm, int splits);
int toa_inst_wide_0_4(int thin, toa_handle h, const toa::FusedParams& prm, int splits);
int toa_inst_misc_wide_1_0(int model, toa_handle h, const toa::FusedParams& prm, int splits);
int toa_inst_wide_1_1(int thin, toa_handle h, const toa::FusedParams& prm, int splits);
int toa_inst_wide_1_2(int thin, toa_handle h, const toa::FusedParams& prm, int splits);
int toa_inst_wide_1_3(int thin, toa_handle h, const toa::FusedParams& prm, int splits);
int toa_inst_wide_1_4(int thin, toa_handle h, const toa::FusedParams& prm, int splits);

int toa_inst_narrow_fused_0_0(int n, toa_handle h, const toa::FusedParams& prm);
int toa_inst_narrow_fused_1_0(int n, toa_handle h, const toa::FusedParams& prm);
int toa_inst_narrow_accumulate_1_0(toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
// the instances of inst.hip's narrow routes of TOA_MODEL_DENSE_ROW (JetModel / RowModel over the packed rows)
static bool dense_row_lane_route(int dtag, int n, bool robust) {
  if (n >= 1 && n <= (dtag == 0 ? 11 : 5)) return true;      // narrow blocks, with or without an M-estimator
  if (dtag == 1 && n == 6) return true;                      // (fp64 n = 6: JetModel without the estimator branch for L2, RowModel with a loss)
  if (!robust) return false;
  return n == 12 || n == 50 || (dtag == 1 && n == 6);         // the BASELINE shapes with an M-estimator on the handle
}
int toa_inst_narrow_accumulate_0_0(toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_jetrow_fused_0_0(int n, toa_handle h, const toa::FusedParams& prm);
int toa_inst_jetrow_fused_1_0(int n, toa_handle h, const toa::FusedParams& prm);
int toa_inst_jetrow_wide_0_0(int n, toa_handle h, const toa::FusedParams& prm);
int toa_inst_jetrow_wide_1_0(int n, toa_handle h, const toa::FusedParams& prm);
int toa_inst_jetrow_accumulate_0_0(toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_jetrow_accumulate_1_0(toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_inv_cov_0_0(int npad, toa_handle h, int n, int64_t P, const void* H, void* C, int32_t* ok);
int toa_inst_inv_cov_1_0(int npad, toa_handle h, int n, int64_t P, const void* H, void* C, int32_t* ok);

static thread_local std::string g_err;
int toa_fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
static int fail(int code, const std::string& msg) { return toa_fail(code, msg); }

int toa_inst_fused(int dtag, int nbm, int thin, toa_handle h, const FusedParams& prm) {
  switch (dtag * 8 + nbm) {
    case 1: return toa_inst_fused_0_1(thin, h, prm); case 2: return toa_inst_fused_0_2(thin, h, prm);
    case 3: return toa_inst_fused_0_3(thin, h, prm); case 4: return toa_inst_fused_0_4(thin, h, prm);
    case 9: return toa_inst_fused_1_1(thin, h, prm); case 10: return toa_inst_fused_1_2(thin, h, prm);
    case 11: return toa_inst_fused_1_3(thin, h, prm); case 12: return toa_inst_fused_1_4(thin, h, prm);
  }
  return fail(TOA_E_ARG, "bad block count");
}
int toa_inst_accumulate(int dtag, int nbm, int thin, toa_handle h, int n, int m, int64_t P, const void* data,
                        const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres) {
#define TOA_A(f) return f(thin, h, n, m, P, data, x, want_grad, g, H, cost, nres)
  switch (dtag * 8 + nbm) {
    case 1: TOA_A(toa_inst_accumulate_0_1); case 2: TOA_A(toa_inst_accumulate_0_2);
    case 3: TOA_A(toa_inst_accumulate_0_3); case 4: TOA_A(toa_inst_accumulate_0_4);
    case 9: TOA_A(toa_inst_accumulate_1_1); case 10: TOA_A(toa_inst_accumulate_1_2);
    case 11: TOA_A(toa_inst_accumulate_1_3); case 12: TOA_A(toa_inst_accumulate_1_4);
  }
#undef TOA_A
  return fail(TOA_E_ARG, "bad block count");
}
int toa_inst_wide(int dtag, int model, int nbm, int thin, toa_handle h, const FusedParams& prm, int splits) {
  if (model != TOA_MODEL_DENSE_ROW)
    return dtag == 0 ? toa_inst_misc_wide_0_0(model, h, prm, splits) : toa_inst_misc_wide_1_0(model, h, prm, splits);
  switch (dtag * 8 + nbm) {
    case 1: return toa_inst_wide_0_1(thin, h, prm, splits); case 2: return toa_inst_wide_0_2(thin, h, prm, splits);
    case 3: return toa_inst_wide_0_3(thin, h, prm, splits); case 4: return toa_inst_wide_0_4(thin, h, prm, splits);
    case 9: return toa_inst_wide_1_1(thin, h, prm, splits); case 10: return toa_inst_wide_1_2(thin, h, prm, splits);
    case 11: return toa_inst_wide_1_3(thin, h, prm, splits); case 12: return toa_inst_wide_1_4(thin, h, prm, splits);
  }
  return fail(TOA_E_ARG, "bad block count");
}
int toa_inst_misc_fused(int dtag, int model, int npad, toa_handle h, const FusedParams& prm) {
  return dtag == 0 ? toa_inst_misc_fused_0_0(model, npad, h, prm) : toa_inst_misc_fused_1_0(model, npad, h, prm);
}
int toa_inst_misc_accumulate(int dtag, int model, int npad, toa_handle h, int n, int m, int64_t P, const void* data,
                             const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres) {
  return dtag == 0 ? toa_inst_misc_accumulate_0_0(model, npad, h, n, m, P, data, x, want_grad, g, H, cost, nres)
                   : toa_inst_misc_accumulate_1_0(model, npad, h, n, m, P, data, x, want_grad, g, H, cost, nres);
}
int toa_inst_inv_cov(int dtag, int npad, toa_handle h, int n, int64_t P, const void* H, void* C, int32_t* ok) {
  return dtag == 0 ? toa_inst_inv_cov_0_0(npad, h, n, P, H, C, ok) : toa_inst_inv_cov_1_0(npad, h, n, P, H, C, ok);
}
int toa_large_solve(toa_handle h, int dtype, int n, int64_t P, const void* H, const void* g, double scale, void* dx, int32_t* ok);
int toa_large_inv_cov(toa_handle h, int dtype, int n, int64_t P, const void* H, void* C, int32_t* ok);
int toa_large_lm_run(toa_handle h, int dtype, int n, int m, int64_t P, const void* data, void* x, const toa_options* options,
                     const toa_results* results, uint64_t* counters);
size_t toa_large_state_bytes(int dtype, int n, int64_t P);
int toa_large_lm_step(toa_handle h, int dtype, int n, int m, int64_t P, const void* data, void* x, const toa_options* options,
                      const toa_results* results, uint64_t* counters, int mode, void* state, int32_t* active_dev,
                      const int32_t* stop_request);
int toa_large_step_log(toa_handle h, int dtype, int n, int64_t P, const void* state, double* lambda, int32_t* nres, int32_t* ninl);
int toa_large_step_info(toa_handle h, int dtype, int n, int64_t P, const void* state, double* err, double* dx2, double* g2,
                        void* dx_out, void* g_out);
int toa_inst_solve(int dtag, int npad, toa_handle h, int n, int64_t P, const void* H, const void* g, double scale,
                   void* dx, int32_t* ok) {
  return dtag == 0 ? toa_inst_solve_0_0(npad, h, n, P, H, g, scale, dx, ok)
                   : toa_inst_solve_1_0(npad, h, n, P, H, g, scale, dx, ok);
}

extern "C" {

const char* toa_last_error(void) { return g_err.c_str(); }

void toa_options_default(toa_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->solver_type = 0;
  o->max_iters = 50;
  o->min_error = 1e-12f;
  o->min_rerr_dec = 1e-10f;
  o->min_step_norm2 = 1e-14f;
  o->min_grad_norm2 = 1e-18f;
  o->max_total_failures = 0;
  o->max_consec_failures = 5;
  o->damping_init = 1e-4f;
  o->damping_min = 1e-9f;
  o->damping_max = 1e9f;
  o->good_factor = 1.0f / 3.0f;
  o->bad_factor = 2.0f;
  o->grad_clipping = 0;
  o->check_min_H_diag = 0;
  o->check_final_cost = 0;
  o->use_step_quality_approx = 0;
  o->use_ldlt = 1;
  o->H_is_full = 1;
  o->save_last = 1;
  o->use_squared_norm = 1;
  o->downscale_by_2 = 0;
  o->normalize = 0;
}

void toa_options_benchmark(toa_options* o) {
  toa_options_default(o);
  o->max_iters = 10;
  o->min_error = 0;
  o->min_rerr_dec = 1e-12f;
  o->min_step_norm2 = 1e-16f;
  o->max_consec_failures = 3;
  o->save_last = 0;
}

int toa_create(toa_handle* out, int device, void* stream) {
  if (!out) return fail(TOA_E_ARG, "toa_create: out is null");
  int count = 0;
  HIP_TRY(hipGetDeviceCount(&count));
  if (device < 0 || device >= count) return fail(TOA_E_ARG, "toa_create: no such device");
  TOA_ON_DEVICE(device);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(TOA_E_UNSUPPORTED, std::string("toa_create: this library is built for gfx950 only, device is ") + prop.gcnArchName);
  toa_context* c = new (std::nothrow) toa_context();
  if (!c) return fail(TOA_E_NOMEM, "toa_create: host allocation failed");
  c->device = device;
  c->stream = static_cast<hipStream_t>(stream);
  c->num_cus = prop.multiProcessorCount;
  c->clock_khz = prop.clockRate;
  c->max_lds = int(prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : prop.sharedMemPerBlock);
  std::strncpy(c->name, prop.name, sizeof(c->name) - 1);
  hipError_t e = hipMalloc(&c->queue, 256);
  if (e != hipSuccess) { delete c; return fail(TOA_E_NOMEM, "toa_create: hipMalloc(queue) failed"); }
  e = hipMalloc(&c->params_dev, 1024);
  if (e != hipSuccess) { (void)hipFree(c->queue); delete c; return fail(TOA_E_NOMEM, "toa_create: hipMalloc(params) failed"); }
  *out = c;
  return TOA_OK;
}

int toa_destroy(toa_handle h) {
  if (!h) return TOA_OK;
  toa::DeviceGuard guard_(h->device);
  if (h->queue) (void)hipFree(h->queue);
  if (h->params_dev) (void)hipFree(h->params_dev);
  if (h->scratch) (void)hipFree(h->scratch);
  if (h->memo) (void)hipFree(h->memo);
  if (h->aux) (void)hipFree(h->aux);
  for (void* b : h->retired_blocks) (void)hipFree(b);   // workspaces outgrown after a capture (toa_release_workspace)
  if (h->pass_flags) (void)hipHostFree(h->pass_flags);
  for (hipEvent_t e : h->pass_done) if (e) (void)hipEventDestroy(e);
  if (h->lane_fork) (void)hipEventDestroy(h->lane_fork);
  for (hipEvent_t e : h->lane_gram) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->lane_join) if (e) (void)hipEventDestroy(e);
  for (hipStream_t s : h->lane_stream) if (s) (void)hipStreamDestroy(s);
  for (int i = 0; i < h->nside; ++i) {
    if (h->side_blas[i] && h->blas_destroy) (void)h->blas_destroy(h->side_blas[i]);
    if (h->side_done[i]) (void)hipEventDestroy(h->side_done[i]);
    if (h->side_stream[i]) (void)hipStreamDestroy(h->side_stream[i]);
  }
  if (h->side_fork) (void)hipEventDestroy(h->side_fork);
  if (h->blas && h->blas_destroy) (void)h->blas_destroy(h->blas);
  for (int i = 0; i < h->nwgraphs; ++i) (void)hipGraphExecDestroy(h->wgraphs[i].exec);
  delete h;
  return TOA_OK;
}

int toa_device_count(int* count) {
  if (!count) return fail(TOA_E_ARG, "toa_device_count: null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;   // (no device, no driver: zero GPUs, not an error)
  *count = n;
  return TOA_OK;
}

int toa_device_info(toa_handle h, int* num_cus, int* clock_khz, char* name, size_t name_len) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (num_cus) *num_cus = h->num_cus;
  if (clock_khz) *clock_khz = h->clock_khz;
  if (name && name_len) { std::strncpy(name, h->name, name_len - 1); name[name_len - 1] = 0; }
  return TOA_OK;
}

int toa_malloc(toa_handle h, void** dev_ptr, size_t bytes) {
  if (!h || !dev_ptr) return fail(TOA_E_ARG, "toa_malloc: null argument");
  TOA_ON_DEVICE(h->device);
  HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 1));
  return TOA_OK;
}
int toa_free(toa_handle h, void* dev_ptr) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  TOA_ON_DEVICE(h->device);
  HIP_TRY(hipFree(dev_ptr));
  return TOA_OK;
}
int toa_memcpy_h2d(toa_handle h, void* dst, const void* src, size_t bytes) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  TOA_ON_DEVICE(h->device);
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return TOA_OK;
}
int toa_memcpy_d2h(toa_handle h, void* dst, const void* src, size_t bytes) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  TOA_ON_DEVICE(h->device);
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return TOA_OK;
}
int toa_memset(toa_handle h, void* dst, int value, size_t bytes) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  TOA_ON_DEVICE(h->device);
  HIP_TRY(hipMemsetAsync(dst, value, bytes, h->stream));
  return TOA_OK;
}
int toa_hbm_read_probe(toa_handle h, const void* src_dev, size_t bytes, int reps, double* gb_per_s) {
  if (!h || !src_dev || !gb_per_s || reps < 1 || bytes < 16) return fail(TOA_E_ARG, "toa_hbm_read_probe: bad argument");
  if (reinterpret_cast<uintptr_t>(src_dev) & 15) return fail(TOA_E_ARG, "toa_hbm_read_probe: src_dev must be 16-byte aligned");
  TOA_ON_DEVICE(h->device);
  const size_t n16 = bytes / 16;
  const int grid = h->num_cus * 16;
  struct Events {  // destroyed on every exit path
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
  } ev;
  HIP_TRY(hipEventCreate(&ev.e0));
  HIP_TRY(hipEventCreate(&ev.e1));
  uint32_t* sink = reinterpret_cast<uint32_t*>(h->queue) + 60;  // unused tail of the 256-byte queue block
  hipLaunchKernelGGL(toa::hbm_read_probe_kernel, dim3(grid), dim3(256), 0, h->stream, (const toa::u32x4*)src_dev, n16, sink);  // warm
  HIP_TRY(hipEventRecord(ev.e0, h->stream));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(toa::hbm_read_probe_kernel, dim3(grid), dim3(256), 0, h->stream, (const toa::u32x4*)src_dev, n16, sink);
  HIP_TRY(hipEventRecord(ev.e1, h->stream));
  HIP_TRY(hipEventSynchronize(ev.e1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  HIP_TRY(hipGetLastError());
  *gb_per_s = double(n16) * 16.0 * reps / (double(ms) * 1e-3) * 1e-9;
  return TOA_OK;
}

int toa_llc_read_probe(toa_handle h, const void* src_dev, size_t bytes, double* gb_per_s) {
  if (!h || !src_dev || !gb_per_s || bytes < (size_t(1) << 20)) return fail(TOA_E_ARG, "toa_llc_read_probe: bad argument (at least 1 MiB)");
  if (reinterpret_cast<uintptr_t>(src_dev) & 15) return fail(TOA_E_ARG, "toa_llc_read_probe: src_dev must be 16-byte aligned");
  TOA_ON_DEVICE(h->device);
  // two slices per compute unit, six waves per slice: twelve waves per CU, the fused kernel's occupancy
  const int slices = h->num_cus * 2, wps = 6;
  const size_t per16 = bytes / 16 / size_t(slices);
  const int grid = (slices * wps + 3) / 4;
  struct Events {
    hipEvent_t e[3] = {nullptr, nullptr, nullptr};
    ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
  } ev;
  for (hipEvent_t& x : ev.e) HIP_TRY(hipEventCreate(&x));
  uint32_t* sink = reinterpret_cast<uint32_t*>(h->queue) + 60;
  auto launch = [&](int passes) {
    hipLaunchKernelGGL(toa::llc_read_probe_kernel, dim3(grid), dim3(256), 0, h->stream, (const toa::u32x4*)src_dev, per16, slices, wps, passes, sink);
  };
  // t(9 passes) - t(3 passes) = six re-reads of a working set that the first passes have brought on-die (whatever the
  // first pass cost — HBM or cache — cancels); best of three
  double best = 0;
  launch(1);
  for (int t = 0; t < 3; ++t) {
    HIP_TRY(hipEventRecord(ev.e[0], h->stream));
    launch(3);
    HIP_TRY(hipEventRecord(ev.e[1], h->stream));
    launch(9);
    HIP_TRY(hipEventRecord(ev.e[2], h->stream));
    HIP_TRY(hipEventSynchronize(ev.e[2]));
    float t3 = 0, t9 = 0;
    HIP_TRY(hipEventElapsedTime(&t3, ev.e[0], ev.e[1]));
    HIP_TRY(hipEventElapsedTime(&t9, ev.e[1], ev.e[2]));
    if (t9 > t3) best = std::max(best, double(per16) * 16.0 * slices * 6.0 / (double(t9 - t3) * 1e-3) * 1e-9);
  }
  HIP_TRY(hipGetLastError());
  *gb_per_s = best;
  return TOA_OK;
}

int toa_robust_norm(toa_handle h, int kind, int dtype, int64_t count, const void* n2, double th2, void* loss, void* scale) {
  if (!h || !n2 || !loss || !scale || count < 0) return fail(TOA_E_ARG, "toa_robust_norm: null argument");
  if (kind < TOA_LOSS_L2 || kind > TOA_LOSS_BLAKE_ZISSERMAN) return fail(TOA_E_ARG, "toa_robust_norm: unknown loss kind");
  if (dtype != TOA_F32 && dtype != TOA_F64) return fail(TOA_E_ARG, "toa_robust_norm: dtype must be TOA_F32 or TOA_F64");
  if (count == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  const unsigned grid = unsigned((count + 255) / 256);
  if (dtype == TOA_F32)
    hipLaunchKernelGGL(toa::robust_norm_kernel<float>, dim3(grid), dim3(256), 0, h->stream, kind, (long long)count,
                       (const float*)n2, float(th2), (float*)loss, (float*)scale);
  else
    hipLaunchKernelGGL(toa::robust_norm_kernel<double>, dim3(grid), dim3(256), 0, h->stream, kind, (long long)count,
                       (const double*)n2, th2, (double*)loss, (double*)scale);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

int toa_jet_eval(toa_handle h, int fn, int dtype, int64_t count, const void* a, const void* b, void* out) {
  if (!h || !a || !b || !out || count < 0) return fail(TOA_E_ARG, "toa_jet_eval: null argument");
  if (fn < 0 || fn > 46) return fail(TOA_E_ARG, "toa_jet_eval: unknown function id");
  if (dtype != TOA_F32 && dtype != TOA_F64) return fail(TOA_E_ARG, "dtype must be TOA_F32 or TOA_F64");
  if (count == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  const unsigned grid = unsigned((count + 255) / 256);
  if (dtype == TOA_F32)
    hipLaunchKernelGGL(toa::jet_eval_kernel<float>, dim3(grid), dim3(256), 0, h->stream, fn, (long long)count, (const float*)a,
                       (const float*)b, (float*)out);
  else
    hipLaunchKernelGGL(toa::jet_eval_kernel<double>, dim3(grid), dim3(256), 0, h->stream, fn, (long long)count, (const double*)a,
                       (const double*)b, (double*)out);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

int toa_abi_version(void) { return TOA_ABI_VERSION; }

int toa_set_tuning(toa_handle h, const toa_tuning* t) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (!t) { h->tune = toa_tuning{}; return TOA_OK; }
  if (t->coop_chunks != 0 && (t->coop_chunks < 2 || t->coop_chunks > 64)) return fail(TOA_E_ARG, "toa_set_tuning: coop_chunks must be 0 or in [2, 64]");
  if (t->max_workgroups < 0 || t->wide_team_max_per_cu < 0) return fail(TOA_E_ARG, "toa_set_tuning: negative count");
  h->tune = *t;
  return TOA_OK;
}
int toa_get_tuning(toa_handle h, toa_tuning* out) {
  if (!h || !out) return fail(TOA_E_ARG, "toa_get_tuning: null argument");
  *out = h->tune;
  return TOA_OK;
}
int toa_debug_timeline(toa_handle h, const char* path) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  h->timeline_path = path ? path : "";
  return TOA_OK;
}

int toa_set_loss(toa_handle h, int kind, double th2) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (kind < TOA_LOSS_L2 || kind > TOA_LOSS_BLAKE_ZISSERMAN) return fail(TOA_E_ARG, "toa_set_loss: unknown loss kind");
  if (kind != TOA_LOSS_L2 && !(th2 > 0)) return fail(TOA_E_ARG, "toa_set_loss: the squared threshold must be positive");
  h->loss = kind;
  h->loss_th2 = th2;
  return TOA_OK;
}

int toa_synchronize(toa_handle h) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  TOA_ON_DEVICE(h->device);
  HIP_TRY(hipStreamSynchronize(h->stream));
  return TOA_OK;
}

// per-model shape contract (reference: std::invalid_argument on dimension mismatch, gn.h:51,66)
static int check_model(int model, int n, int m, const void* data) {
  switch (model) {
    case TOA_MODEL_DENSE_ROW:
      if (!data) return fail(TOA_E_ARG, "DenseRow: data pointer is null");
      return TOA_OK;
    case TOA_MODEL_GAUSSIAN_PRIOR:
      if (m != n) return fail(TOA_E_ARG, "GaussianPrior: m must equal n (one residual per parameter)");
      if (!data) return fail(TOA_E_ARG, "GaussianPrior: data pointer ([P][2][n]: y, sigma) is null");
      return TOA_OK;
    case TOA_MODEL_SQRT2:
      if (n != 1 || m != 1) return fail(TOA_E_ARG, "Sqrt2: n and m must be 1");
      return TOA_OK;
    case TOA_MODEL_SE3_PRIOR:
      if (n != 6 || m != 6) return fail(TOA_E_ARG, "SE3Prior: n and m must be 6");
      if (!data) return fail(TOA_E_ARG, "SE3Prior: data pointer ([P][12] = prior_inv) is null");
      return TOA_OK;
    case TOA_MODEL_MAHA_PRIOR:
      if (m != n) return fail(TOA_E_ARG, "MahaPrior: m must equal n (one whitened residual per parameter)");
      if (!data) return fail(TOA_E_ARG, "MahaPrior: data pointer ([P][n + n*n]: y, U) is null");
      return TOA_OK;
    case TOA_MODEL_TESTFN:
      if (n != 1 && n != 2 && n != 4) return fail(TOA_E_ARG, "TestFn: n must be 1 (x - 2), 2 (Rosenbrock, plateau, Beale, Himmelblau) or 4 (Powell)");
      if (!data) return fail(TOA_E_ARG, "TestFn: data pointer ([1] = function id) is null");
      return TOA_OK;
    case TOA_MODEL_CIRCLE_FIT:
      if (n != 3) return fail(TOA_E_ARG, "CircleFit: n must be 3 (cx, cy, radius)");
      if (!data) return fail(TOA_E_ARG, "CircleFit: data pointer ([P][m][2] observed points) is null");
      return TOA_OK;
    case TOA_MODEL_DENSE_ROW_AD6:
      if (n != 6) return fail(TOA_E_ARG, "DenseRowAD6: n must be 6");
      if (!data) return fail(TOA_E_ARG, "DenseRowAD6: data pointer ([P][m][7] = a_i, b_i) is null");
      return TOA_OK;
    case TOA_MODEL_DENSE_ROW_AD:
      if (n != 12 && n != 50) return fail(TOA_E_UNSUPPORTED, "DenseRowAD: instantiated for n = 12 and n = 50 (a functor's parameter count is a compile-time constant)");
      if (!data) return fail(TOA_E_ARG, "DenseRowAD: data pointer ([P][m][n + 1] = a_i, b_i) is null");
      return TOA_OK;
    case TOA_MODEL_SE3_REPROJ:
      if (n != 6 || m < 2 || (m & 1)) return fail(TOA_E_ARG, "SE3Reproj: n must be 6 and m an even count of residuals");
      if (!data) return fail(TOA_E_ARG, "SE3Reproj: data pointer ([P][8 + 5*m/2]) is null");
      return TOA_OK;
    default:
      return fail(TOA_E_UNSUPPORTED, "model not available on this path");
  }
}

static int check_shape(int dtype, int n, int m, int64_t P) {
  if (dtype != TOA_F32 && dtype != TOA_F64) return fail(TOA_E_ARG, "dtype must be TOA_F32 or TOA_F64");
  if (n < 1 || n > 63) return fail(TOA_E_ARG, "n must be in [1, 63] on the LDS-resident path");
  if (m < 1) return fail(TOA_E_ARG, "m must be >= 1");
  if (P < 0 || P > 0x7fffffff) return fail(TOA_E_ARG, "P out of range");
  return TOA_OK;
}

int toa_dense_row_layout(int dtype, int n, int m, int* nb, int* thin, int* row_stride, int* rows_padded,
                         size_t* bytes_per_problem) {
  if (int rc = check_shape(dtype, n, m, 0)) return rc;
  const DenseRowLayout L = DenseRowLayout::make(n, m);
  if (nb) *nb = L.nbm;
  if (thin) *thin = L.thin;
  if (row_stride) *row_stride = L.rs;
  if (rows_padded) *rows_padded = L.m4;
  if (bytes_per_problem) *bytes_per_problem = L.elems_per_problem() * (dtype == TOA_F32 ? 4 : 8);
  return TOA_OK;
}

int toa_dense_row_pack(toa_handle h, int dtype, int n, int m, int64_t P, const void* A, const void* b, void* packed) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (int rc = check_shape(dtype, n, m, P)) return rc;
  if (P == 0) return TOA_OK;
  if (!A || !b || !packed) return fail(TOA_E_ARG, "toa_dense_row_pack: null pointer");
  TOA_ON_DEVICE(h->device);
  const DenseRowLayout L = DenseRowLayout::make(n, m);
  const int grid = h->num_cus * 8;
  if (dtype == TOA_F32)
    hipLaunchKernelGGL(dense_row_pack_kernel<float>, dim3(grid), dim3(256), 0, h->stream, (const float*)A, (const float*)b,
                       (float*)packed, (long long)P, n, m, L);
  else
    hipLaunchKernelGGL(dense_row_pack_kernel<double>, dim3(grid), dim3(256), 0, h->stream, (const double*)A, (const double*)b,
                       (double*)packed, (long long)P, n, m, L);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

int toa_dense_row_synth(toa_handle h, int dtype, int n, int m, int64_t P, uint64_t seed, int64_t problem0,
                        void* packed, void* x0, void* xstar) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (int rc = check_shape(dtype, n, m, P)) return rc;
  if (P == 0) return TOA_OK;
  if (!packed) return fail(TOA_E_ARG, "toa_dense_row_synth: packed_dev is null");
  TOA_ON_DEVICE(h->device);
  const DenseRowLayout L = DenseRowLayout::make(n, m);
  const int grid = h->num_cus * 16;
  if (dtype == TOA_F32)
    hipLaunchKernelGGL(dense_row_synth_kernel<float>, dim3(grid), dim3(256), 0, h->stream, (float*)packed, (float*)x0,
                       (float*)xstar, (long long)P, n, m, L, (unsigned long long)seed, (long long)problem0);
  else
    hipLaunchKernelGGL(dense_row_synth_kernel<double>, dim3(grid), dim3(256), 0, h->stream, (double*)packed, (double*)x0,
                       (double*)xstar, (long long)P, n, m, L, (unsigned long long)seed, (long long)problem0);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

static int check_loss_supported(toa_handle h, int model, const char* who);
int toa_accumulate(toa_handle h, int model, int dtype, int n, int m, int64_t P, const void* data, const void* x,
                   int want_grad, void* g, void* H, double* cost, int32_t* nres) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (model == TOA_MODEL_DENSE_ROW_NATURAL) {  // the seam beyond one wavefront (SolverGN::Accumulate / Evaluate at any Dims, gn.h:97-113)
    if (dtype != TOA_F32 && dtype != TOA_F64) return fail(TOA_E_ARG, "dtype must be TOA_F32 or TOA_F64");
    if (n < 1 || n > 4096) return fail(TOA_E_ARG, "TOA_MODEL_DENSE_ROW_NATURAL: n must be in [1, 4096]");
    if (m < 1 || P < 0 || !data) return fail(TOA_E_ARG, "toa_accumulate: bad shape or null data pointer");
    if (!x || !cost || (want_grad && (!g || !H))) return fail(TOA_E_ARG, "toa_accumulate: null pointer");
    if (P == 0) return TOA_OK;
    TOA_ON_DEVICE(h->device);
    // 64 <= n <= 128 without an M-estimator: one launch of the workgroup-per-problem kernel (large_fused.hip).  Everything else the
    // natural layout takes — n > 128 up to 4096, n < 64, toa_set_loss at any n — is ONE data pass of the launch-per-stage pipeline
    // (rows kernel + Gram, ours or the library's; round 6).
    if (h->loss == TOA_LOSS_L2 && toa_large_fused_eligible(h, dtype, n, m))
      return toa_large_accumulate(h, dtype, n, m, P, data, x, want_grad, g, H, cost, nres);
    return toa_large_accumulate_pipeline(h, dtype, n, m, P, data, x, want_grad, g, H, cost, nres);
  }
  if (int rc = check_shape(dtype, n, m, P)) return rc;
  if (int rc = check_model(model, n, m, data)) return rc;
  if (int rc = check_loss_supported(h, model, "toa_accumulate")) return rc;
  if (!x || !cost || (want_grad && (!g || !H))) return fail(TOA_E_ARG, "toa_accumulate: null pointer");
  if (P == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  const int dtag = dtype == TOA_F32 ? 0 : 1;
  if (model == TOA_MODEL_DENSE_ROW_AD)
    return dtag == 0 ? toa_inst_jetrow_accumulate_0_0(h, n, m, P, data, x, want_grad, g, H, cost, nres)
                     : toa_inst_jetrow_accumulate_1_0(h, n, m, P, data, x, want_grad, g, H, cost, nres);
  if (model != TOA_MODEL_DENSE_ROW)
    return toa_inst_misc_accumulate(dtag, model, 16 * ((n + 15) / 16), h, n, m, P, data, x, want_grad, g, H, cost, nres);
  const DenseRowLayout lay_ = DenseRowLayout::make(n, m);
  if (h->loss == TOA_LOSS_L2 && !h->tune.narrow_mfma_pass && dense_row_lane_route(dtag, n, false))
    return dtag == 0 ? toa_inst_narrow_accumulate_0_0(h, n, m, P, data, x, want_grad, g, H, cost, nres)
                     : toa_inst_narrow_accumulate_1_0(h, n, m, P, data, x, want_grad, g, H, cost, nres);
  return toa_inst_accumulate(dtag, lay_.nbm, lay_.thin, h, n, m, P, data, x, want_grad, g, H, cost, nres);
}

int toa_solve_damped(toa_handle h, int dtype, int n, int64_t P, const void* H, const void* g, double scale, void* dx,
                     int32_t* ok) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  // n <= 63: one wavefront per matrix (register / LDS LDL^T); 64 <= n <= 4096: rocSOLVER batched Cholesky (large_n.hip).
  // toa_tuning::large_library_solver sends small matrices down the library path too (tools/k3_crossover.py measures both).
  const bool force_lib = h->tune.large_library_solver != 0;
  const bool large = n > 63 || force_lib;
  if (large) {
    if (dtype != TOA_F32 && dtype != TOA_F64) return fail(TOA_E_ARG, "dtype must be TOA_F32 or TOA_F64");
    if (n < 1 || n > 4096) return fail(TOA_E_ARG, "toa_solve_damped: n must be in [1, 4096]");
    if (P < 0 || P > 65535) return fail(TOA_E_ARG, "toa_solve_damped: P must be in [0, 65535] on the library path");
  } else if (int rc = check_shape(dtype, n, 1, P)) {
    return rc;
  }
  if (!H || !g || !dx || !ok) return fail(TOA_E_ARG, "toa_solve_damped: null pointer");
  if (P == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  if (large) return toa_large_solve(h, dtype, n, P, H, g, scale, dx, ok);
  return toa_inst_solve(dtype == TOA_F32 ? 0 : 1, 16 * ((n + 15) / 16), h, n, P, H, g, scale, dx, ok);
}

int toa_inv_cov(toa_handle h, int dtype, int n, int64_t P, const void* H, void* C, int32_t* ok) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (n > 63) {  // beyond one wavefront: Cholesky against the identity through rocSOLVER (large_n.hip)
    if (dtype != TOA_F32 && dtype != TOA_F64) return fail(TOA_E_ARG, "dtype must be TOA_F32 or TOA_F64");
    if (n > 4096) return fail(TOA_E_ARG, "toa_inv_cov: n must be in [1, 4096]");
    if (P < 0 || P > 65535) return fail(TOA_E_ARG, "toa_inv_cov: P must be in [0, 65535] for n > 63");
    if (!H || !C || !ok) return fail(TOA_E_ARG, "toa_inv_cov: null pointer");
    if (P == 0) return TOA_OK;
    TOA_ON_DEVICE(h->device);
    return toa_large_inv_cov(h, dtype, n, P, H, C, ok);
  }
  if (int rc = check_shape(dtype, n, 1, P)) return rc;
  if (!H || !C || !ok) return fail(TOA_E_ARG, "toa_inv_cov: null pointer");
  if (P == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  return toa_inst_inv_cov(dtype == TOA_F32 ? 0 : 1, 16 * ((n + 15) / 16), h, n, P, H, C, ok);
}

// toa_set_loss is sticky handle state: a launch of a family that has no M-estimator while a loss is set would silently be a
// plain L2 solve with inlier_ratio = 1.  Refused instead (the header promises "refused, never ignored").
static int check_loss_supported(toa_handle h, int model, const char* who) {
  if (h->loss == TOA_LOSS_L2) return TOA_OK;
  switch (model) {
    case TOA_MODEL_DENSE_ROW: case TOA_MODEL_CIRCLE_FIT: case TOA_MODEL_DENSE_ROW_AD6: case TOA_MODEL_DENSE_ROW_NATURAL:
      return TOA_OK;   // (NATURAL: its own range check follows in the callers)
    case TOA_MODEL_SE3_REPROJ:
      return fail(TOA_E_UNSUPPORTED, std::string(who) + ": TOA_MODEL_SE3_REPROJ takes its loss from its data header; clear the handle's (toa_set_loss(h, TOA_LOSS_L2, 0))");
    default:
      return fail(TOA_E_UNSUPPORTED, std::string(who) + ": this model family has no M-estimator and a loss is set on the handle (toa_set_loss); clear it first");
  }
}

static int lm_run_impl(toa_handle h, int model, int dtype, int n, int m, int64_t P, const void* data, void* x,
                       const toa_options* options, const toa_results* results, uint64_t* counters, int splits,
                       int mode = 0, void* state = nullptr, int32_t* active = nullptr, const int32_t* stop_request = nullptr) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  const bool natural = model == TOA_MODEL_DENSE_ROW_NATURAL;  // n beyond one wavefront (large_fused.hip / large_n.hip)
  if (natural) {
    if (dtype != TOA_F32 && dtype != TOA_F64) return fail(TOA_E_ARG, "dtype must be TOA_F32 or TOA_F64");
    // (beyond 1024 unknowns every stage of a pass is the library's — rocBLAS GEMM / GEMV, rocSOLVER potrf / potrs or LU — like
    //  toa_solve_damped, which takes n up to 4096 the same way; the reference's Dims == Dynamic is unbounded, optimizer.h:61-92)
    if (n < 1 || n > 4096) return fail(TOA_E_ARG, "TOA_MODEL_DENSE_ROW_NATURAL: n must be in [1, 4096]");
    if (m < 1) return fail(TOA_E_ARG, "m must be >= 1");
    if (P < 0) return fail(TOA_E_ARG, "P must be >= 0");   // (no upper limit: the launch-per-stage pipeline takes 65 535 problems per slice)
    if (!data) return fail(TOA_E_ARG, "null data pointer");
    if (splits >= 0) return fail(TOA_E_UNSUPPORTED, "TOA_MODEL_DENSE_ROW_NATURAL: no row-split form");
    if (mode != 0 && n < 64) return fail(TOA_E_UNSUPPORTED, "TOA_MODEL_DENSE_ROW_NATURAL: the stepping form starts at n = 64 (use TOA_MODEL_DENSE_ROW below)");
  } else {
    if (int rc = check_shape(dtype, n, m, P)) return rc;
    if (int rc = check_model(model, n, m, data)) return rc;
  }
  if (int rc = check_loss_supported(h, model, "toa_lm_run")) return rc;
  if (!x || !options || !results) return fail(TOA_E_ARG, "toa_lm_run: null pointer");
  if (!results->stop_reason || !results->num_iters || !results->final_cost)
    return fail(TOA_E_ARG, "toa_lm_run: stop_reason, num_iters and final_cost outputs are required");
  if (options->solver_type != 0 && options->solver_type != 1)
    return fail(TOA_E_ARG, "toa_lm_run: solver_type must be 0 (LM) or 1 (GN) on this path");  // optimize.h:75
  if ((results->errs || results->deltas2 || results->successes) && results->hist_stride < options->max_iters + 2)
    return fail(TOA_E_ARG, "toa_lm_run: hist_stride must be >= max_iters + 2");
  if (options->max_iters < 0 || options->max_iters > 65535) return fail(TOA_E_ARG, "max_iters out of range");
  if (P == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  if (natural) {
    // (toa_set_loss: every form of this family applies it since round 5 — the one-kernel form at 64 <= n <= 128, the
    //  launch-per-stage pipeline beyond, for fp64 rows above n = 96, and under the stepping form)
    if (mode != 0) {   // the stepping form runs on the launch-per-stage pipeline for every n >= 64 (large_n.hip)
      if (!state) return fail(TOA_E_ARG, "toa_lm_begin / toa_lm_step: state_dev is null");
      return toa_large_lm_step(h, dtype, n, m, P, data, x, options, results, counters, mode, state, active, stop_request);
    }
    return toa_large_lm_run(h, dtype, n, m, P, data, x, options, results, counters);
  }
  FusedParams prm;
  std::memset(&prm, 0, sizeof(prm));
  prm.data = data;
  prm.x = x;
  prm.P = P;
  prm.n = n;
  prm.m = m;
  prm.opt = *options;
  prm.res = *results;
  prm.counters = reinterpret_cast<unsigned long long*>(counters);
  prm.mode = mode;
  prm.state = state;
  prm.active = active;
  prm.stop_request = stop_request;
  prm.loss = h->loss;
  prm.loss_th2 = h->loss_th2;
  if (mode != 0) {
    if (!state) return fail(TOA_E_ARG, "toa_lm_begin / toa_lm_step: state_dev is null");
    // the stepping form runs on the launch-per-iteration kernels with one chunk per problem (launch_stepping)
    const DenseRowLayout lay_s = DenseRowLayout::make(n, m);
    if (model == TOA_MODEL_DENSE_ROW_AD) return dtype == TOA_F32 ? toa_inst_jetrow_wide_0_0(n, h, prm) : toa_inst_jetrow_wide_1_0(n, h, prm);
    return toa_inst_wide(dtype == TOA_F32 ? 0 : 1, model, lay_s.nbm, lay_s.thin, h, prm, 1);
  }
  const int dtag = dtype == TOA_F32 ? 0 : 1;
  const DenseRowLayout lay_ = DenseRowLayout::make(n, m);
  const bool splittable = model == TOA_MODEL_DENSE_ROW || model == TOA_MODEL_SE3_REPROJ;
  // splits < 0: automatic — row-split when one-wave-per-problem would leave most of the chip idle
  // (fewer problems than CUs and enough rows to give every chunk >= 256 of them)
  //   or, for small problems (n <= 15, 512..4096 rows), the team form: one workgroup per problem has no co-residency
  //   requirement, so it also pays for whole batches of them — measured (tests/tools/team_probe.py, C2-sized problems):
  //   89-100 us for 1..256 problems against 131-144 us with one wavefront per problem; the crossover is one problem per
  //   compute unit at n = 6 x 1000 rows and two at n = 12 x 2000.  toa_tuning::wide_team_max_per_cu overrides, toa_tuning::wide_no_autosplit disables.
  if (splits == -1) {
    const bool no_auto = h->tune.wide_no_autosplit != 0;
    const long long team_env = h->tune.wide_team_max_per_cu;
    const long long team_per_cu = team_env > 0 ? team_env : ((long long)m * (n + 1) >= 20000 ? 2 : 1);
    const bool few = P * 4 <= h->num_cus && m >= 512;
    const bool team = n <= 15 && m >= 512 && m <= 4096 && P <= team_per_cu * h->num_cus;
    splits = (splittable && !no_auto && (few || team)) ? 0 : -1;
  }
  // DenseRow with an M-estimator on the handle: the robust data pass lives in the launch-per-iteration form (kernels.hpp
  // RobustOf): chunked automatically for a few huge problems, one chunk per problem for a batch
  // (round 6: where a row-per-lane instance exists — inst.hip — a BATCH runs the loss inside the fused kernel instead)
  if (model == TOA_MODEL_DENSE_ROW && h->loss != TOA_LOSS_L2 && splits < 0) {
    const bool few = P * 4 <= h->num_cus && m >= 512;
    if (!few && !h->tune.narrow_mfma_pass && dense_row_lane_route(dtag, n, true))
      return dtag == 0 ? toa_inst_narrow_fused_0_0(n, h, prm) : toa_inst_narrow_fused_1_0(n, h, prm);
    splits = few ? 0 : 1;
  }
  if (splits >= 0) {
    if (!splittable) return fail(TOA_E_UNSUPPORTED, "row-split execution is available for DenseRow and SE3Reproj");
    return toa_inst_wide(dtag, model, lay_.nbm, lay_.thin, h, prm, splits);
  }
  if (model == TOA_MODEL_DENSE_ROW_AD) return dtag == 0 ? toa_inst_jetrow_fused_0_0(n, h, prm) : toa_inst_jetrow_fused_1_0(n, h, prm);
  if (model != TOA_MODEL_DENSE_ROW) return toa_inst_misc_fused(dtag, model, 16 * ((n + 15) / 16), h, prm);
  // narrow fp32 blocks: a row per lane (RowModel) instead of sixteen lanes per row (toa_tuning::narrow_mfma_pass: the old route)
  if (!h->tune.narrow_mfma_pass && dense_row_lane_route(dtag, n, false)) return dtag == 0 ? toa_inst_narrow_fused_0_0(n, h, prm) : toa_inst_narrow_fused_1_0(n, h, prm);
  return toa_inst_fused(dtag, lay_.nbm, lay_.thin, h, prm);
  return fail(TOA_E_ARG, "toa_lm_run: bad block count");
}

// The system (Hessians, work matrices) cannot be allocated: the reference does not throw — ResizeIfNeeded catches
// std::bad_alloc and the solve returns with StopReason::kOutOfMemory, x untouched, nothing iterated (optimizer.h:75-86,
// stop_reasons.h:20).  Same here: every problem's stop_reason becomes TOA_STOP_OUT_OF_MEMORY, num_iters 0, and the call
// succeeds; toa_last_error() still names the allocation.
static int oom_as_stop_reason(toa_handle h, int rc, int64_t P, const toa_results* results) {
  if (rc != TOA_E_NOMEM || !results || !results->stop_reason || P <= 0) return rc;
  if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(results->stop_reason), int(TOA_STOP_OUT_OF_MEMORY), size_t(P), h->stream) != hipSuccess) return rc;
  if (results->num_iters) (void)hipMemsetAsync(results->num_iters, 0, size_t(P) * sizeof(int32_t), h->stream);
  if (results->num_failures) (void)hipMemsetAsync(results->num_failures, 0, size_t(P) * sizeof(int32_t), h->stream);
  if (results->num_consec_failures) (void)hipMemsetAsync(results->num_consec_failures, 0, size_t(P) * sizeof(int32_t), h->stream);
  return TOA_OK;
}

int toa_lm_run(toa_handle h, int model, int dtype, int n, int m, int64_t P, const void* data, void* x,
               const toa_options* options, const toa_results* results, uint64_t* counters) {
  return oom_as_stop_reason(h, lm_run_impl(h, model, dtype, n, m, P, data, x, options, results, counters, -1), P, results);
}

size_t toa_lm_state_bytes(int dtype, int n, int64_t P) {
  if (P < 0 || n < 1) return 0;
  if (n >= 64) return toa_large_state_bytes(dtype, n, P);   // TOA_MODEL_DENSE_ROW_NATURAL (the only family that wide)
  return dtype == TOA_F32 ? toa::stepping_state_bytes<float>(n, P) : toa::stepping_state_bytes<double>(n, P);
}

int toa_lm_begin(toa_handle h, int model, int dtype, int n, int m, int64_t P, const void* data, void* x,
                 const toa_options* options, const toa_results* results, void* state_dev) {
  return lm_run_impl(h, model, dtype, n, m, P, data, x, options, results, nullptr, -1, 1, state_dev, nullptr);
}

int toa_lm_step(toa_handle h, int model, int dtype, int n, int m, int64_t P, const void* data, void* x,
                const toa_options* options, const toa_results* results, uint64_t* counters, void* state_dev,
                int32_t* active_dev) {
  return lm_run_impl(h, model, dtype, n, m, P, data, x, options, results, counters, -1, 2, state_dev, active_dev);
}

int toa_lm_stop(toa_handle h, int model, int dtype, int n, int m, int64_t P, const void* data, void* x,
                const toa_options* options, const toa_results* results, uint64_t* counters, void* state_dev,
                const int32_t* stop_request_dev) {
  if (!stop_request_dev) return fail(TOA_E_ARG, "toa_lm_stop: stop_request_dev is null");
  return lm_run_impl(h, model, dtype, n, m, P, data, x, options, results, counters, -1, 3, state_dev, nullptr, stop_request_dev);
}

int toa_lm_step_info(toa_handle h, int dtype, int n, int64_t P, const void* state_dev, double* err_dev, double* dx_norm2_dev,
                     double* grad_norm2_dev, void* dx_dev, void* g_dev) {
  if (!h || !state_dev) return fail(TOA_E_ARG, "toa_lm_step_info: null argument");
  if (n >= 64) {
    if ((dtype != TOA_F32 && dtype != TOA_F64) || n > 4096 || P < 0 || P > 65535) return fail(TOA_E_ARG, "toa_lm_step_info: bad shape");
    if (P == 0) return TOA_OK;
    TOA_ON_DEVICE(h->device);
    return toa_large_step_info(h, dtype, n, P, state_dev, err_dev, dx_norm2_dev, grad_norm2_dev, dx_dev, g_dev);
  }
  if (int rc = check_shape(dtype, n, 1, P)) return rc;
  if (P == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  const unsigned grid = unsigned((P + 3) / 4);
  if (dtype == TOA_F32)
    hipLaunchKernelGGL(toa::step_info_kernel<float>, dim3(grid), dim3(256), 0, h->stream, state_dev, (long long)P, n, err_dev,
                       dx_norm2_dev, grad_norm2_dev, (float*)dx_dev, (float*)g_dev);
  else
    hipLaunchKernelGGL(toa::step_info_kernel<double>, dim3(grid), dim3(256), 0, h->stream, state_dev, (long long)P, n, err_dev,
                       dx_norm2_dev, grad_norm2_dev, (double*)dx_dev, (double*)g_dev);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

int toa_lm_step_log(toa_handle h, int dtype, int n, int64_t P, const void* state_dev, double* lambda_dev, int32_t* num_residuals_dev,
                    int32_t* num_inliers_dev) {
  if (!h || !state_dev) return fail(TOA_E_ARG, "toa_lm_step_log: null argument");
  if (n >= 64) {
    if ((dtype != TOA_F32 && dtype != TOA_F64) || n > 4096 || P < 0 || P > 65535) return fail(TOA_E_ARG, "toa_lm_step_log: bad shape");
    if (P == 0) return TOA_OK;
    TOA_ON_DEVICE(h->device);
    return toa_large_step_log(h, dtype, n, P, state_dev, lambda_dev, num_residuals_dev, num_inliers_dev);
  }
  if (int rc = check_shape(dtype, n, 1, P)) return rc;
  if (P == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  const unsigned grid = unsigned((P + 255) / 256);
  if (dtype == TOA_F32)
    hipLaunchKernelGGL(toa::step_log_kernel<float>, dim3(grid), dim3(256), 0, h->stream, state_dev, (long long)P, lambda_dev, num_residuals_dev, num_inliers_dev);
  else
    hipLaunchKernelGGL(toa::step_log_kernel<double>, dim3(grid), dim3(256), 0, h->stream, state_dev, (long long)P, lambda_dev, num_residuals_dev, num_inliers_dev);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

int toa_lm_run_split(toa_handle h, int model, int dtype, int n, int m, int64_t P, const void* data, void* x,
                     const toa_options* options, const toa_results* results, uint64_t* counters, int splits) {
  if (splits < 0) return fail(TOA_E_ARG, "toa_lm_run_split: splits must be >= 0 (0 = choose automatically)");
  return lm_run_impl(h, model, dtype, n, m, P, data, x, options, results, counters, splits);
}

}  // extern "C"
