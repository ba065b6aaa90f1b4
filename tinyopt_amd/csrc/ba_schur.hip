// Bundle adjustment with the points eliminated ("block-sparse / Schur"): C SE3 cameras x N 3-D points,
// reprojection residuals, the SAME Levenberg-Marquardt state machine as every other path (LmState, lm_judge_core,
// lm_good_step / lm_bad_step of lm_device.hpp).  The reference has no such path: it would run `Optimize(x, acc)` on the
// full (6C + 3N)^2 system with a dense LDL^T, or Eigen's SimplicialLDLT on the sparse one (include/tinyopt/math.h:232-240,
// 266-277; README.md:30,165-167 "sparse is slow") — which is what the oracle does (oracle/ba.hpp).  Here the block
// structure of H = [[U, W], [W^T, V]] (U: 6x6 per camera, V: 3x3 per point, W: 6x3 per observation) is used:
//
//   (U - W V^-1 W^T) dc = -g_c + W V^-1 g_p            reduced camera system, 6C <= 60 unknowns: blocked LDL^T by the four waves
//   dp_j = -V_j^-1 (g_pj + W_j^T dc)                   back-substitution, a 3x3 solve per point
//
// with Marquardt's multiplicative damping on EVERY diagonal entry of H (lm.h:108-117), cameras and points alike.
// The Schur complement is accumulated on the matrix cores: with V_j = R_j R_j^T (Cholesky) and Y_j = [W_1j; ...; W_Cj],
//   W V^-1 W^T = sum_j Z_j Z_j^T,  W V^-1 g_p = sum_j Z_j q_j,   Z_j = Y_j R_j^-T (6C x 3),  q_j = R_j^-1 g_pj
// i.e. the Gram of the (3N) x (6C + 1) matrix of rows [Z_j[:, a]^T | q_j[a]] — exactly the [J | r] Gram DenseRowGram
// computes, fed through add_step (one point = 3 rows of a 4-row MFMA step).
//
// One workgroup (4 waves) per scene runs the whole solve in one launch; scenes are independent (grid = P).  Every reduction
// has a fixed order (per-lane partial sums, wave butterflies, waves folded 0..3), so results are bit-reproducible.
//   data: [f cx cy 0 0 0 0 0 | uv: C x N x 2 | vis: C x N]     x: [12 C poses (R row-major, t) | 3 N points], in place
#include "kernels.hpp"
#include "ldlt_wg.hpp"

namespace toa {

struct BaParams {
  const void* data;
  void* x;
  void* work;                 // per scene: (W blocks,) V, g_p, R^-1, q, dp, last dp, points of the last build (see BaWork)
  long long P;
  int C, N;
  toa_options opt;
  toa_results res;
  unsigned long long* counters;
  int lds_wave;               // bytes of the WaveLds carve (wave 0's LDL^T workspace + vectors + LmState)
  int lds_gpart;              // byte offset of the per-wave camera sums
  int lds_part2;              // byte offset of a second partial-Gram buffer (n*n + 128 elements), 0 = none (LDS budget)
};

template <typename T>
struct BaWork {  // element offsets into a scene's scratch block
  size_t W, Voff, hdp, gp, Rinv, q, dp, ldp, ptsb, total;
  int nb64;
  __host__ __device__ BaWork(int C, int N) {
    size_t o = 0;
    // W_cj = J_c^T J_p (6 x 3, element e = 3 dof + b): points in tiles of 64 (one wave's worth), element-major inside a tile —
    // block (c, j) element e at ((c nb64 + j / 64) 18 + e) 64 + j % 64.  The lanes of a wave (consecutive points) then touch
    // consecutive addresses and the 18 elements are compile-time offsets.  (Block-major [c][j][18] made every store of a wave
    // 64 separate cache lines: 18 such stores per observation kept the accumulate phase on the address unit.  The per-point
    // arrays below stay point-major: component-major costs more registers than the kernel has — 52 spills, slower.)
    nb64 = (N + 63) / 64;
    W = o; o += size_t(C) * nb64 * 18 * 64;
    Voff = o; o += size_t(N) * 3;     // V_j off-diagonals (0,1) (0,2) (1,2)
    hdp = o; o += size_t(N) * 3;      // CURRENT (damped) diagonal of V_j
    gp = o; o += size_t(N) * 3;
    Rinv = o; o += size_t(N) * 6;     // R_j^-1, lower: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2)
    q = o; o += size_t(N) * 3;
    dp = o; o += size_t(N) * 3;
    ldp = o; o += size_t(N) * 3;
    ptsb = o; o += size_t(N) * 3;     // the points of the last BUILD (eval-only iterations solve with that system while x sits at a trial point)
    total = (o + 63) & ~size_t(63);
  }
};

// fixed-order sum over the 256 threads of the workgroup: wave butterflies, then the four wave totals in index order
template <typename T>
__device__ __forceinline__ T ba_block_sum(T v, T* red4) {
  v = wave_allreduce_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

// pose <- pose * exp(sign * d)  (sophus.h:24-26), one thread, pose = R row-major | t
template <typename T>
__device__ __forceinline__ void ba_se3_plus(T* P, const T* dv, T sign) {
  T d[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) d[i] = sign * dv[i];
  const T wx = d[3], wy = d[4], wz = d[5];
  const T t2 = wx * wx + wy * wy + wz * wz;
  const T th = sqrt(t2);
  T A, B, Cc;
  if (t2 < T(1e-10)) { A = T(1) - t2 / T(6); B = T(0.5) - t2 / T(24); Cc = T(1) / T(6) - t2 / T(120); }
  else { T sn, cs; sincos_t(th, &sn, &cs); A = sn / th; B = (T(1) - cs) / t2; Cc = (th - sn) / (t2 * th); }
  T Rd[9];
  Rd[0] = T(1) - B * (wy * wy + wz * wz); Rd[1] = -A * wz + B * wx * wy;          Rd[2] = A * wy + B * wx * wz;
  Rd[3] = A * wz + B * wx * wy;          Rd[4] = T(1) - B * (wx * wx + wz * wz); Rd[5] = -A * wx + B * wy * wz;
  Rd[6] = -A * wy + B * wx * wz;         Rd[7] = A * wx + B * wy * wz;          Rd[8] = T(1) - B * (wx * wx + wy * wy);
  const T c1[3] = {wy * d[2] - wz * d[1], wz * d[0] - wx * d[2], wx * d[1] - wy * d[0]};
  const T c2[3] = {wy * c1[2] - wz * c1[1], wz * c1[0] - wx * c1[2], wx * c1[1] - wy * c1[0]};
  T td[3], x[12];
#pragma unroll
  for (int i = 0; i < 3; ++i) td[i] = d[i] + B * c1[i] + Cc * c2[i];
#pragma unroll
  for (int i = 0; i < 12; ++i) x[i] = P[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) P[3 * i + j] = x[3 * i] * Rd[j] + x[3 * i + 1] * Rd[3 + j] + x[3 * i + 2] * Rd[6 + j];
    P[9 + i] = x[3 * i] * td[0] + x[3 * i + 1] * td[1] + x[3 * i + 2] * td[2] + x[9 + i];
  }
}

// residual and Jacobians of observation (camera P, point q): r (2), Jc (2 x 6), Jp (2 x 3)
template <typename T, bool WANT_J>
__device__ __forceinline__ void ba_obs(const T* P, const T* q, const T f, const T cx, const T cy, const T u, const T v, T* r, T (*Jc)[6],
                                       T (*Jp)[3]) {
  const T X = P[0] * q[0] + P[1] * q[1] + P[2] * q[2] + P[9];
  const T Y = P[3] * q[0] + P[4] * q[1] + P[5] * q[2] + P[10];
  const T Z = P[6] * q[0] + P[7] * q[1] + P[8] * q[2] + P[11];
  const T iz = T(1) / Z;
  r[0] = f * X * iz + cx - u;
  r[1] = f * Y * iz + cy - v;
  if constexpr (WANT_J) {
    const T du0 = f * iz, du2 = -f * X * iz * iz, dv1 = f * iz, dv2 = -f * Y * iz * iz;
    T D[3][6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      D[a][0] = P[3 * a]; D[a][1] = P[3 * a + 1]; D[a][2] = P[3 * a + 2];
      D[a][3] = -(P[3 * a + 1] * q[2] - P[3 * a + 2] * q[1]);
      D[a][4] = -(-P[3 * a] * q[2] + P[3 * a + 2] * q[0]);
      D[a][5] = -(P[3 * a] * q[1] - P[3 * a + 1] * q[0]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) { Jc[0][k] = fma(du0, D[0][k], du2 * D[2][k]); Jc[1][k] = fma(dv1, D[1][k], dv2 * D[2][k]); }
#pragma unroll
    for (int k = 0; k < 3; ++k) { Jp[0][k] = fma(du0, P[k], du2 * P[6 + k]); Jp[1][k] = fma(dv1, P[3 + k], dv2 * P[6 + k]); }
  }
}

// W_cj[dof][b] = (J_c^T J_p)[dof][b]: ONE spelling for every site that forms it (the contraction into fma must not differ
// between the accumulate phase, the Schur loop and the back-substitution)
template <typename T>
__device__ __forceinline__ T ba_w(const T jc0, const T jc1, const T jp0, const T jp1) { return fma(jc0, jp0, jc1 * jp1); }
// element d of a 6-vector held in registers, d known only at run time (a dynamic index would go through scratch)
template <typename T>
__device__ __forceinline__ T ba_pick6(const T (&a)[6], const int d) {
  T v = a[0];
#pragma unroll
  for (int k = 1; k < 6; ++k) {
    v = d == k ? a[k] : v;
    asm volatile("" : "+v"(v));   // keeps it a chain of selects: hipcc otherwise turns it back into an indexed load from scratch
  }
  return v;
}

// Rows dof0 .. dof0 + NR - 1 of W_cj = J_c^T J_p for observation (camera P, point q): the same expressions as ba_obs + ba_w, but
// only the NR camera columns a lane of the Schur loop owns (dof0 is a multiple of NR, known at run time only).
template <typename T, int NR>
__device__ __forceinline__ void ba_obs_wrows(const T* P, const T* q, const T f, const int dof0, T (&w)[NR][3]) {
  const T X = P[0] * q[0] + P[1] * q[1] + P[2] * q[2] + P[9];
  const T Y = P[3] * q[0] + P[4] * q[1] + P[5] * q[2] + P[10];
  const T Z = P[6] * q[0] + P[7] * q[1] + P[8] * q[2] + P[11];
  const T iz = T(1) / Z;
  const T du0 = f * iz, du2 = -f * X * iz * iz, dv1 = f * iz, dv2 = -f * Y * iz * iz;
  T Jp[2][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { Jp[0][k] = fma(du0, P[k], du2 * P[6 + k]); Jp[1][k] = fma(dv1, P[3 + k], dv2 * P[6 + k]); }
  T D[3][6];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    D[a][0] = P[3 * a]; D[a][1] = P[3 * a + 1]; D[a][2] = P[3 * a + 2];
    D[a][3] = -(P[3 * a + 1] * q[2] - P[3 * a + 2] * q[1]);
    D[a][4] = -(-P[3 * a] * q[2] + P[3 * a + 2] * q[0]);
    D[a][5] = -(P[3 * a] * q[1] - P[3 * a + 1] * q[0]);
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    T d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if constexpr (NR == 3) d[a] = dof0 != 0 ? D[a][3 + i] : D[a][i];   // translation or rotation half
      else d[a] = ba_pick6<T>(D[a], dof0 + i);
    }
    const T jc0 = fma(du0, d[0], du2 * d[2]), jc1 = fma(dv1, d[1], dv2 * d[2]);
#pragma unroll
    for (int b = 0; b < 3; ++b) w[i][b] = ba_w<T>(jc0, jc1, Jp[0][b], Jp[1][b]);
  }
}

// Two workgroups per CU (256 registers per lane; the fp64 instances spill ~25 of them): uncapped, hipcc takes 340 registers
// for fp64 and ONE workgroup — four waves — is all a CU runs, with nothing to cover the latency of the phase-to-phase L2
// round trips.  Measured (bench.py --workload ba, 1024 scenes x 8 cameras x 256 points fp64): 3.76 ms uncapped, 2.46 ms
// with 2, 3.8 ms with 3 (118 spills).
#ifndef TOA_BA_WGS
#define TOA_BA_WGS 2
#endif
#ifdef TOA_BA_TIMING   // workgroup 0 prints where its time went (constant 100 MHz ticks -> us)
#define BA_TICK_START unsigned long long tkp_ = wall_clock64();
#define BA_TICK(i) { const unsigned long long now_ = wall_clock64(); tk_[i] += now_ - tkp_; tkp_ = now_; }
#else
#define BA_TICK_START
#define BA_TICK(i)
#endif
template <typename T, int NBM, int THIN>
__global__ void __launch_bounds__(256, TOA_BA_WGS) ba_schur_kernel(const BaParams* __restrict__ prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = prm->C, N = prm->N, n = 6 * C;
#ifdef TOA_BA_TIMING
  unsigned long long tk_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const long long p = blockIdx.x;
  const BaWork<T> wk(C, N);
  const int nb64 = wk.nb64;
  // W is 18 values per observation: written once and read twice per iteration it was 3/4 of the kernel's HBM traffic (the 151 MB
  // of the resident scenes' W do not fit the L2s; PMC: 6.0 GB per launch against 0.28 GB of algorithmic bytes).  Where all the
  // Schur columns of a lane belong to ONE camera (NBM divides 6, no thin Z columns) W is never stored: the Schur loop and the
  // back-substitution re-form it from the observation — pose and point of the last build (poses_b, ptsb), u, v, visibility:
  // 24 bytes per observation instead of 144, ~110 flops.
  constexpr bool kRecomputeW = (6 % NBM == 0) && THIN <= 1;
  // ---- LDS carve: [ WaveLds of wave 0 | part: n*n | pvec, phd: 64 + 64 | poses, poses_b: 12 C + 12 C | U: 36 C | gc, rhs: 64 + 64 | red | flags ]
  WaveLds<T> L = WaveLds<T>::carve(smem, n);
  T* part = reinterpret_cast<T*>(smem + prm->lds_wave);
  T* pvec = part + size_t(n) * n;
  T* phd = pvec + 64;
  T* poses = phd + 64;
  T* poses_b = poses + 12 * C;  // the poses of the last build (see kRecomputeW)
  T* U = poses_b + 12 * C;        // undamped 6 x 6 blocks (row-major); the damped diagonal lives in L.hd
  T* gc = U + 36 * C;
  T* red = gc + 64;             // [8]
  int* flags = reinterpret_cast<int*>(red + 8);   // [0] continue, [1] do_acc, [2] action, [3] build ok, [4] solve ok
  T* gpart = reinterpret_cast<T*>(smem + prm->lds_gpart);   // [4 waves][C][32]: per-wave totals of the cameras' 28 sums
  // Every HBM array is addressed through address_space(1) pointers.  The parameter block is read from memory, so hipcc
  // cannot prove that the pointers in it are global and emits flat_load / flat_store for plain `T*` (528 + 474 of them in
  // this kernel): a flat access counts on BOTH memory counters, so every LDS wait also drains the outstanding HBM accesses
  // and the other way round — the phases of an iteration were serialised on memory latency.
  using GT = __attribute__((address_space(1))) T;
  using GCT = const __attribute__((address_space(1))) T;
  GCT* data = (GCT*)(static_cast<const T*>(prm->data) + size_t(p) * (8 + size_t(3) * C * N));
  const T f = data[0], cx = data[1], cy = data[2];
  GCT* uv = data + 8;
  GCT* vis = uv + size_t(2) * C * N;
  GT* X = (GT*)(static_cast<T*>(prm->x) + size_t(p) * (size_t(12) * C + size_t(3) * N));
  GT* pts = X + 12 * C;
  GT* work = (GT*)(static_cast<T*>(prm->work) + size_t(p) * wk.total);
  GT* Wb = work + wk.W; GT* Voff = work + wk.Voff; GT* hdp = work + wk.hdp; GT* gp = work + wk.gp;
  GT* Rinv = work + wk.Rinv; GT* qv = work + wk.q; GT* dp = work + wk.dp; GT* ldp = work + wk.ldp; GT* ptsb = work + wk.ptsb;
  const DenseRowLayout lay = DenseRowLayout::make(n, 4);

  if (wave == 0) {
    const int* src_o = reinterpret_cast<const int*>(&prm->opt);
    int* dst_o = reinterpret_cast<int*>(L.opt);
    for (int i = lane; i < int(sizeof(toa_options) / 4); i += 64) dst_o[i] = src_o[i];
    const int* src_r = reinterpret_cast<const int*>(&prm->res);
    int* dst_r = reinterpret_cast<int*>(L.res);
    for (int i = lane; i < int(sizeof(toa_results) / 4); i += 64) dst_r[i] = src_r[i];
    L.st->acc_passes = 0; L.st->eval_passes = 0; L.st->solves = 0; L.st->problems = 0;
    wave_sync();
    lm_init<T>(L, lane);
  }
  for (int i = tid; i < 12 * C; i += 256) poses[i] = X[i];
  __syncthreads();
  LmState<T>& S = *L.st;
  const toa_options& opt = *L.opt;
  const bool is_lm = opt.solver_type == 0;

  for (;;) {  // one pass = Build (+ Solve) of the loop at optimizer.h:358; a failed solve retries without advancing the iteration
    const bool do_acc = !is_lm || S.rebuild;
    BA_TICK_START
    // ================= Accumulate / Evaluate (gn.h:97-113) =================
    T csum = 0, nvis = 0;
    if (do_acc)
      for (int i = tid; i < 12 * C; i += 256) poses_b[i] = poses[i];   // read again only behind later barriers
    for (int c = 0; c < C; ++c) {
      T Pm[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) Pm[i] = poses[12 * c + i];
      T G[28];
#pragma unroll
      for (int i = 0; i < 28; ++i) G[i] = T(0);
      for (int j = tid; j < N; j += 256) {
        const bool seen = vis[size_t(c) * N + j] != T(0);
        const T q[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
        T r[2] = {T(0), T(0)};
        if (do_acc) {
          T Jc[2][6], Jp[2][3];
#pragma unroll
          for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int k = 0; k < 6; ++k) Jc[a][k] = T(0);
#pragma unroll
            for (int k = 0; k < 3; ++k) Jp[a][k] = T(0);
          }
          if (seen) ba_obs<T, true>(Pm, q, f, cx, cy, uv[(size_t(c) * N + j) * 2], uv[(size_t(c) * N + j) * 2 + 1], r, Jc, Jp);
          // camera block: upper Gram of [Jc | r] (7 x 7), as Se3ReprojModel
#pragma unroll
          for (int row = 0; row < 2; ++row) {
            T w7[7];
#pragma unroll
            for (int k = 0; k < 6; ++k) w7[k] = Jc[row][k];
            w7[6] = r[row];
            int t = 0;
#pragma unroll
            for (int a = 0; a < 7; ++a)
#pragma unroll
              for (int b = a; b < 7; ++b) G[t++] += w7[a] * w7[b];
          }
          // point block V_j += Jp^T Jp, g_pj += Jp^T r (the thread owns point j for every camera: no races), W_cj = Jc^T Jp
          T v6[6], g3[3];
          if (c == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { v6[k] = T(0); v6[3 + k] = T(0); g3[k] = T(0); }
          } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { v6[k] = hdp[3 * j + k]; v6[3 + k] = Voff[3 * j + k]; g3[k] = gp[3 * j + k]; }
          }
#pragma unroll
          for (int row = 0; row < 2; ++row) {
            v6[0] += Jp[row][0] * Jp[row][0]; v6[1] += Jp[row][1] * Jp[row][1]; v6[2] += Jp[row][2] * Jp[row][2];
            v6[3] += Jp[row][0] * Jp[row][1]; v6[4] += Jp[row][0] * Jp[row][2]; v6[5] += Jp[row][1] * Jp[row][2];
#pragma unroll
            for (int k = 0; k < 3; ++k) g3[k] += Jp[row][k] * r[row];
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) { hdp[3 * j + k] = v6[k]; Voff[3 * j + k] = v6[3 + k]; gp[3 * j + k] = g3[k]; }
          if constexpr (kRecomputeW) {
            if (c == 0) { ptsb[3 * j] = q[0]; ptsb[3 * j + 1] = q[1]; ptsb[3 * j + 2] = q[2]; }
          } else {
            GT* Wd = Wb + (size_t(c) * nb64 + (j >> 6)) * (18 * 64) + (j & 63);
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
              for (int b = 0; b < 3; ++b) Wd[(3 * k + b) * 64] = ba_w(Jc[0][k], Jc[1][k], Jp[0][b], Jp[1][b]);
          }
        } else if (seen) {
          ba_obs<T, false>(Pm, q, f, cx, cy, uv[(size_t(c) * N + j) * 2], uv[(size_t(c) * N + j) * 2 + 1], r, nullptr, nullptr);
        }
        csum += r[0] * r[0] + r[1] * r[1];
        nvis += seen ? T(2) : T(0);
      }
      if (do_acc) {  // this wave's totals of the camera's 28 sums: one transposed reduction, parked in LDS until every camera is done
        T G32[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) G32[i] = i < 28 ? G[i] : T(0);
        const T tot = wave_transposed_reduce32(G32, lane);
        if (lane < 28) gpart[(wave * C + c) * 32 + lane] = tot;
      }
    }
    if (do_acc) {  // fold the four waves' totals in fixed order and scatter: 6 x 6 block of camera c, g_c
      __syncthreads();
      for (int idx = tid; idx < 28 * C; idx += 256) {
        const int c = idx / 28, t = idx - 28 * c;
        const T tot = (gpart[(0 * C + c) * 32 + t] + gpart[(1 * C + c) * 32 + t]) + (gpart[(2 * C + c) * 32 + t] + gpart[(3 * C + c) * 32 + t]);
        int a = 0, rem = t;  // tt(a, b) = a * 7 - a (a - 1) / 2 + (b - a)
        while (rem >= 7 - a) { rem -= 7 - a; ++a; }
        const int b = a + rem;
        if (b < 6) { U[36 * c + 6 * a + b] = tot; U[36 * c + 6 * b + a] = tot; }
        else if (a < 6) gc[6 * c + a] = tot;
      }
      __syncthreads();
    }
    BA_TICK(0)
    const T cost_raw = ba_block_sum<T>(csum, red);
    const int nres = int(ba_block_sum<T>(nvis, red));
    __syncthreads();
    // ================= rest of Build (lm.h:59-120): validity, clipping, diagonal check, Marquardt damping =================
    if (tid == 0) {
      if (do_acc) S.acc_passes++; else S.eval_passes++;
      S.cost_val = normalize_cost(double(cost_raw), nres, opt);
      S.cost_nres = nres;
      S.cost_ninl = nres;
      flags[3] = (nres > 0 && S.cost_val != kDblMax) ? 1 : 0;
    }
    __syncthreads();
    bool built = flags[3] != 0;
    if (built && do_acc) {
      if (opt.grad_clipping != 0) {  // base.h:29-38
        const T mm = opt.grad_clipping;
        for (int i = tid; i < n; i += 256) gc[i] = fmin(fmax(gc[i], -mm), mm);
        for (int i = tid; i < 3 * N; i += 256) gp[i] = fmin(fmax(gp[i], -mm), mm);
      }
      for (int i = tid; i < n; i += 256) L.hd[i] = U[36 * (i / 6) + 7 * (i % 6)];   // undamped camera diagonal
      __syncthreads();
      if (opt.check_min_H_diag > 0) {  // lm.h:82-86, every diagonal entry of H
        T low = 0;
        for (int i = tid; i < n; i += 256) low += fabs(L.hd[i]) < T(opt.check_min_H_diag) ? T(1) : T(0);
        for (int i = tid; i < 3 * N; i += 256) low += fabs(hdp[i]) < T(opt.check_min_H_diag) ? T(1) : T(0);
        if (ba_block_sum<T>(low, red) > T(0)) built = false;
      }
    }
    __syncthreads();
    if (built && is_lm && S.lambda > T(0)) {  // lm.h:108-117, s in double
      const double s = S.rebuild ? 1.0 + double(S.lambda) : (1.0 + double(S.lambda)) / (1.0 + double(S.prev_lambda));
      for (int i = tid; i < n; i += 256) L.hd[i] = T(double(L.hd[i]) * s);
      for (int i = tid; i < 3 * N; i += 256) hdp[i] = T(double(hdp[i]) * s);
    }
    __syncthreads();
    BA_TICK(1)
    // ================= Solve (gn.h:150-171) through the Schur complement =================
    T bad = 0;
    if (built) {
      // per point: Cholesky of the damped V_j, R^-1, q = R^-1 g_p
      for (int j = tid; j < N; j += 256) {
        const T a00 = hdp[3 * j], a11 = hdp[3 * j + 1], a22 = hdp[3 * j + 2];
        const T a01 = Voff[3 * j], a02 = Voff[3 * j + 1], a12 = Voff[3 * j + 2];
        // a point no camera sees has V = 0: like the dense LDL^T's zero pivots it takes a zero step (pseudo-inverse)
        const bool empty = a00 == T(0) && a11 == T(0) && a22 == T(0) && a01 == T(0) && a02 == T(0) && a12 == T(0);
        T r00 = 0, r10 = 0, r11 = 0, r20 = 0, r21 = 0, r22 = 0;   // R^-1 (lower)
        if (!empty) {
          const T l00s = a00;
          const T l00 = sqrt(l00s);
          const T l10 = a01 / l00, l20 = a02 / l00;
          const T l11s = a11 - l10 * l10;
          const T l11 = sqrt(l11s);
          const T l21 = (a12 - l20 * l10) / l11;
          const T l22s = a22 - l20 * l20 - l21 * l21;
          const T l22 = sqrt(l22s);
          if (!(l00s > T(0)) || !(l11s > T(0)) || !(l22s > T(0))) bad += T(1);   // not positive definite: the solve fails
          r00 = T(1) / l00; r11 = T(1) / l11; r22 = T(1) / l22;
          r10 = -l10 * r00 * r11;
          r21 = -l21 * r11 * r22;
          r20 = -(l20 * r00 + l21 * r10) * r22;
        }
        GT* Rj = Rinv + 6 * j;
        Rj[0] = r00; Rj[1] = r10; Rj[2] = r11; Rj[3] = r20; Rj[4] = r21; Rj[5] = r22;
        const T g0 = gp[3 * j], g1 = gp[3 * j + 1], g2 = gp[3 * j + 2];
        qv[3 * j] = r00 * g0;
        qv[3 * j + 1] = r10 * g0 + r11 * g1;
        qv[3 * j + 2] = r20 * g0 + r21 * g1 + r22 * g2;
      }
      bad = ba_block_sum<T>(bad, red);
      __syncthreads();
      BA_TICK(2)
      // M = U (block diagonal, damped diagonal), rhs = -g_c
      for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e % n;
        T v = (i / 6 == j / 6) ? U[36 * (i / 6) + 6 * (i % 6) + (j % 6)] : T(0);
        if (i == j) v = L.hd[i];
        L.M[i * L.LD + j] = v;
      }
      for (int i = tid; i < 64; i += 256) L.vec[i] = i < n ? -gc[i] : T(0);
      __syncthreads();
      // the Schur Gram on the matrix cores.  Points go in groups of four consecutive ones, wave w takes groups w, w + 4, ...;
      // a 4-row MFMA step carries row a of [Z_j | q_j] for the four points of a group (one point per row group), so the three
      // steps of a group share ONE fetch of the group's W / R^-1 / q values.
      {
        DenseRowGram<T, NBM, THIN> gram;
        gram.clear();
        const int k = lane >> 4, c16 = lane & 15;
        const bool isB = (THIN == 0) && ((c16 + 1) * NBM == lay.rsm);
        const int ngroups = (N + 3) >> 2;
        const int ngr_w = ngroups > wave ? (ngroups - wave + 3) / 4 : 0;
        // The values a lane needs of its point — the W entries of its NBM (+ thin) columns, R^-1, q — are fetched one group
        // AHEAD of the Gram steps that consume them: they are L2 hits (~0.7 us) in front of 0.2 us of arithmetic.  (First form
        // of this loop: one point per step in row groups 0..2 — the fourth idle, every W value fetched by three row groups:
        // 13 loads per lane and step instead of 6, 64 steps per wave of an 8 x 256 scene instead of 48.)
        constexpr int NCOL = NBM + (THIN > 0 ? THIN - 1 : 0);
        // stored-W form: the lane's W entries; recomputed form: what W is made of (point of the last build, u, v, visibility)
        struct PtVals { T wv[kRecomputeW ? 1 : NCOL][3]; T pt[3]; T vs; T r[6]; T q[3]; };
        // column of Z handled by slot i of this lane: its NBM main columns, then the thin ones
        auto slot_col = [&](const int i) -> int { return i < NBM ? NBM * c16 + i : lay.nmr + (i - NBM); };
        const int col0 = NBM * c16;                       // kRecomputeW: the lane's columns col0 .. col0 + NBM - 1 are
        const int cam_l = col0 / 6, dof0 = col0 % 6;      // degrees of freedom dof0 .. of camera cam_l
        const bool cam_ok = cam_l < C;
        const T* Pb = poses_b + 12 * (cam_ok ? cam_l : 0);
        auto load_grp = [&](const int sidx) __attribute__((always_inline)) {
          PtVals pv;
          const int j = 4 * (wave + 4 * sidx) + k;
          const bool live = sidx < ngr_w && j < N;
          const int jj = live ? j : 0;          // every load is unconditional (a valid address), its value masked
          GCT* Rj = Rinv + 6 * jj;
#pragma unroll
          for (int b2 = 0; b2 < 6; ++b2) { const T v0 = Rj[b2]; pv.r[b2] = live ? v0 : T(0); }
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) { const T v0 = qv[3 * jj + b2]; pv.q[b2] = live ? v0 : T(0); }
          if constexpr (kRecomputeW) {
            const size_t ob = size_t(cam_ok ? cam_l : 0) * N + jj;
#pragma unroll
            for (int b2 = 0; b2 < 3; ++b2) pv.pt[b2] = ptsb[3 * jj + b2];
            const T vs = vis[ob];
            pv.vs = (live && cam_ok) ? vs : T(0);
          } else {
#pragma unroll
            for (int i = 0; i < NCOL; ++i) {
              const int col = slot_col(i);
              const bool use = live && col < n && (i >= NBM || col < lay.nmr);
              const int ju = use ? j : 0;
              GCT* Wd = Wb + ((size_t(use ? col / 6 : 0) * nb64 + (ju >> 6)) * 18 + 3 * (use ? col % 6 : 0)) * 64 + (ju & 63);
#pragma unroll
              for (int b2 = 0; b2 < 3; ++b2) { const T v0 = Wd[b2 * 64]; pv.wv[i][b2] = use ? v0 : T(0); }
            }
          }
          return pv;
        };
        PtVals cur = load_grp(0);
        for (int s = 0; s < ngr_w; ++s) {
          const PtVals nxt = load_grp(s + 1);   // in flight during this group's arithmetic
          T wrec[NBM][3];   // kRecomputeW: the lane's W rows, re-formed from the observation (cam_l, point of row group k)
          if constexpr (kRecomputeW) {
            T Pm[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) Pm[i] = Pb[i];
            ba_obs_wrows<T, NBM>(Pm, cur.pt, f, dof0, wrec);
            const bool seen = cur.vs != T(0);
#pragma unroll
            for (int i = 0; i < NBM; ++i) {
              const bool use = seen && col0 + i < n && col0 + i < lay.nmr;
#pragma unroll
              for (int b2 = 0; b2 < 3; ++b2) wrec[i][b2] = use ? wrec[i][b2] : T(0);
            }
          }
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            // Z_j[col][a] = sum_b W_{cam,j}[dof][b] R^-1[a][b]   (R^-1 lower triangular: r = {00, 10, 11, 20, 21, 22})
            auto zslot = [&](const int i) -> T {
              T w3[3];
#pragma unroll
              for (int b2 = 0; b2 < 3; ++b2) {
                if constexpr (kRecomputeW) w3[b2] = wrec[i < NBM ? i : 0][b2];
                else w3[b2] = cur.wv[i][b2];
              }
              if (a == 0) return w3[0] * cur.r[0];
              if (a == 1) return w3[0] * cur.r[1] + w3[1] * cur.r[2];
              return w3[0] * cur.r[3] + w3[1] * cur.r[4] + w3[2] * cur.r[5];
            };
            T w[NBM], v[THIN ? THIN : 1];
#pragma unroll
            for (int cb = 0; cb < NBM; ++cb) w[cb] = zslot(cb);
            if constexpr (THIN == 0) {
              if (isB) w[NBM - 1] = cur.q[a];
            } else {
#pragma unroll
              for (int jt = 0; jt + 1 < THIN; ++jt) v[jt] = zslot(NBM + jt);
              v[THIN - 1] = cur.q[a];
            }
            gram.add_step(w, v, __builtin_amdgcn_readfirstlane(int(s + 1 == ngr_w && a == 2)));
          }
          cur = nxt;
        }
        BA_TICK(7)
        gram.finish_steps();
        // fold the four partial Grams in wave order: M - G0 - G1 - G2 - G3, rhs + p0 + p1 + p2 + p3.  With a second LDS buffer
        // (where two workgroups per CU still fit) two waves publish at a time: half the barriers, the same arithmetic.
        if (prm->lds_part2 != 0) {
          T* part2 = reinterpret_cast<T*>(smem + prm->lds_part2);
          T* pvec2 = part2 + size_t(n) * n;
          T* phd2 = pvec2 + 64;
          for (int h2 = 0; h2 < 2; ++h2) {
            if (wave == 2 * h2) {
              gram.write_sym(part, n, lay, n, lane);
              (void)gram.extract_g_diag_cost(pvec, phd, lay, n, lane, &red[4]);
            } else if (wave == 2 * h2 + 1) {
              gram.write_sym(part2, n, lay, n, lane);
              (void)gram.extract_g_diag_cost(pvec2, phd2, lay, n, lane, &red[5]);
            }
            __syncthreads();
            for (int e = tid; e < n * n; e += 256) {
              const int i = e / n, j = e % n;
              T m = L.M[i * L.LD + j];
              m -= (i == j) ? phd[i] : part[e];
              m -= (i == j) ? phd2[i] : part2[e];
              L.M[i * L.LD + j] = m;
            }
            for (int i = tid; i < n; i += 256) L.vec[i] = (L.vec[i] + pvec[i]) + pvec2[i];
            __syncthreads();
          }
        } else {
          for (int wv = 0; wv < 4; ++wv) {
            if (wave == wv) {
              gram.write_sym(part, n, lay, n, lane);
              (void)gram.extract_g_diag_cost(pvec, phd, lay, n, lane, &red[4]);
            }
            __syncthreads();
            for (int e = tid; e < n * n; e += 256) {
              const int i = e / n, j = e % n;
              L.M[i * L.LD + j] -= (i == j) ? phd[i] : part[e];
            }
            for (int i = tid; i < n; i += 256) L.vec[i] += pvec[i];
            __syncthreads();
          }
        }
      }
      BA_TICK(3)
      // reduced camera system (6C unknowns): blocked LDL^T of the LDS image by the four waves, trailing updates on the matrix
      // cores (ldlt_wg.hpp; the one-wavefront register factorisation used here before took 33 us of a 145 us iteration at
      // C = 8 in fp64).  It factors in place, so the image is parked in `part` (free since the fold) first: a pivot that is
      // not safely positive falls back to the pivoted factorisation with the reference's acceptance rule on the restored image.
      for (int e = tid; e < n * n; e += 256) part[e] = L.M[(e / n) * L.LD + (e % n)];
      __syncthreads();
      constexpr int NBS = NBM + (THIN > 1 ? 1 : 0);   // 16-column panels covering n = 16 NBM + max(THIN - 1, 0) (- 1 without a thin tail)
      const bool wg_ok = WgLdlt<T, NBS>::factor(L.M, L.LD, n, L.tmp, tid);
      if (wave == 0) {
        bool ok = wg_ok;
        if (ok) {
          L.dx[lane] = lane < n ? L.vec[lane] : T(0);
          wave_sync();
          WgLdlt<T, NBS>::solve(L.M, L.LD, n, L.tmp, L.dx, lane);
        } else {
          const T rhs = lane < n ? L.vec[lane] : T(0);
          for (int e = lane; e < n * n; e += 64) L.M[(e / n) * L.LD + (e % n)] = part[e];
          wave_sync();
          ok = ldlt_factor_wave<T>(L.M, L.LD, L.perm, L.tmp, n, lane);
          if (ok) L.dx[lane] = ldlt_solve_wave<T>(L.M, L.LD, L.perm, L.vec, n, lane, rhs);
        }
        wave_sync();
        if (lane == 0) { flags[4] = (ok && bad == T(0)) ? 1 : 0; S.solves++; }
      }
      __syncthreads();
    }
    BA_TICK(4)
    const bool solved = built && flags[4] != 0;
    // back-substitution, |dx|^2, |g|^2 (optimizer.h:412-415)
    T d2 = 0, g2 = 0;
    if (solved) {
      for (int j = tid; j < N; j += 256) {
        T u3[3] = {gp[3 * j], gp[3 * j + 1], gp[3 * j + 2]};
        g2 += u3[0] * u3[0] + u3[1] * u3[1] + u3[2] * u3[2];
        T qb[3] = {T(0), T(0), T(0)};
        if constexpr (kRecomputeW) { qb[0] = ptsb[3 * j]; qb[1] = ptsb[3 * j + 1]; qb[2] = ptsb[3 * j + 2]; }
        for (int c = 0; c < C; ++c) {
          T wl[18];
          if constexpr (kRecomputeW) {   // W_cj re-formed from the observation at the point of the last build
            T Pm[12], Jc[2][6], Jp[2][3], rr[2];
#pragma unroll
            for (int i = 0; i < 12; ++i) Pm[i] = poses_b[12 * c + i];
            const bool seen = vis[size_t(c) * N + j] != T(0);
            ba_obs<T, true>(Pm, qb, f, T(0), T(0), T(0), T(0), rr, Jc, Jp);   // the Jacobians do not depend on cx, cy, u, v
#pragma unroll
            for (int kk = 0; kk < 6; ++kk)
#pragma unroll
              for (int b = 0; b < 3; ++b) { const T w0 = ba_w<T>(Jc[0][kk], Jc[1][kk], Jp[0][b], Jp[1][b]); wl[3 * kk + b] = seen ? w0 : T(0); }
          } else {
            GCT* Wd = Wb + (size_t(c) * nb64 + (j >> 6)) * (18 * 64) + (j & 63);
            // the block's 18 loads go out together (left to itself hipcc reuses ONE register pair: 18 round trips)
#pragma unroll
            for (int e = 0; e < 18; ++e) wl[e] = Wd[e * 64];
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int kk = 0; kk < 6; ++kk) {
            const T dck = L.dx[6 * c + kk];
            u3[0] += wl[3 * kk] * dck; u3[1] += wl[3 * kk + 1] * dck; u3[2] += wl[3 * kk + 2] * dck;
          }
        }
        GCT* Rj = Rinv + 6 * j;
        const T y0 = Rj[0] * u3[0], y1 = Rj[1] * u3[0] + Rj[2] * u3[1], y2 = Rj[3] * u3[0] + Rj[4] * u3[1] + Rj[5] * u3[2];
        const T e0 = -(Rj[0] * y0 + Rj[1] * y1 + Rj[3] * y2), e1 = -(Rj[2] * y1 + Rj[4] * y2), e2 = -(Rj[5] * y2);
        dp[3 * j] = e0; dp[3 * j + 1] = e1; dp[3 * j + 2] = e2;
        d2 += e0 * e0 + e1 * e1 + e2 * e2;
      }
      for (int i = tid; i < n; i += 256) { d2 += L.dx[i] * L.dx[i]; g2 += gc[i] * gc[i]; }
    }
    BA_TICK(5)
    const T d2s = ba_block_sum<T>(d2, red);
    const T g2s = ba_block_sum<T>(g2, red);
    __syncthreads();
    // ================= Step bookkeeping + the loop body of OptimizeAcc (optimizer.h:266-310, 370-399), one thread =================
    if (tid == 0) {
      const unsigned max_tries = opt.max_consec_failures > 0 ? (opt.max_consec_failures > 1 ? opt.max_consec_failures : 1) : 255;
      int rc;  // 0 step, 1 solver failed for good, 2 early stop, -1 retry (same iteration)
      if (solved) {
        rc = 0;
      } else {
        S.num_consec = (S.num_consec + 1) & 0xff;
        S.num_failures = (S.num_failures + 1) & 0xff;
        if (S.cost_nres == 0) { S.stop = TOA_STOP_SKIPPED; rc = 2; }
        else if (isnan(S.cost_val) || isinf(S.cost_val)) { S.stop = TOA_STOP_NAN_OR_INF; rc = 2; }
        else if (opt.max_consec_failures > 0 && S.num_consec >= unsigned(opt.max_consec_failures)) {
          if (S.final_cost < double(NumLimits<T>::max())) S.stop = TOA_STOP_MAX_CONSEC_NO_DECR;
          rc = 1;
        } else {
          lm_bad_step(S, opt);
          rc = (S.num_consec <= max_tries) ? -1 : 1;
        }
      }
      int action = 0, cont = 1;
      if (rc >= 0) {
        int status = 0;
        if (rc == 1) S.stop = TOA_STOP_SOLVER_FAILED;
        if (rc == 0) status = lm_judge_core<T>(S, opt, *L.res, p, double(d2s), opt.min_grad_norm2 > 0.0f ? double(g2s) : 0.0, true);
        bool eval_only = false;
        if (status & 1) {
          action = 1; S.has_last_dx = 1; S.last_was_success = 1;
          if (opt.check_final_cost && S.iter + 1 == S.max_iters) eval_only = true;
        } else {
          if (S.has_last_dx) { action = 2; S.has_last_dx = 0; }
          else if (status & 2) { action = 1; S.has_last_dx = 1; }
          eval_only = (S.last_was_success == 0);
          S.last_was_success = 0;
        }
        if (is_lm) S.rebuild = eval_only ? 0 : 1;
        S.num_iters = S.num_iters + 1;
        S.iter = S.iter + 1;
        cont = (S.stop == TOA_STOP_NONE && S.iter < S.max_iters) ? 1 : 0;
      }
      flags[0] = cont;
      flags[2] = action;
    }
    __syncthreads();
    const int action = flags[2];
    if (action == 1) {  // x (+)= dx: SE3 on the cameras (sophus.h:24-26), Euclidean on the points (traits.h:184-190)
      if (tid < C) ba_se3_plus<T>(poses + 12 * tid, L.dx + 6 * tid, T(1));
      if (tid >= 64 && tid < 128) L.ldx[tid - 64] = L.dx[tid - 64];
      for (int i = tid; i < 3 * N; i += 256) { const T d = dp[i]; pts[i] += d; ldp[i] = d; }
    } else if (action == 2) {  // roll back the last step
      if (tid < C) ba_se3_plus<T>(poses + 12 * tid, L.ldx + 6 * tid, T(-1));
      for (int i = tid; i < 3 * N; i += 256) pts[i] -= ldp[i];
    }
    __syncthreads();
    BA_TICK(6)
    if (!flags[0]) break;
  }
#ifdef TOA_BA_TIMING
  if (tid == 0 && blockIdx.x == 0)
    printf("ba wg0: accumulate %.1f us  build %.1f us  point Cholesky %.1f us  Schur Gram %.1f + fold %.1f us  camera solve %.1f us  back-substitution %.1f us  step %.1f us\n",
           tk_[0] * 0.01, tk_[1] * 0.01, tk_[2] * 0.01, tk_[7] * 0.01, tk_[3] * 0.01, tk_[4] * 0.01, tk_[5] * 0.01, tk_[6] * 0.01);
#endif
  // ---- optimizer.h:313-327
  for (int i = tid; i < 12 * C; i += 256) X[i] = poses[i];
  if (tid == 0) {
    const toa_results& res = *L.res;
    if (S.stop == TOA_STOP_NONE && S.num_iters >= S.max_iters) S.stop = TOA_STOP_MAX_ITERS;
    res.stop_reason[p] = S.stop;
    res.num_iters[p] = S.num_iters;
    res.final_cost[p] = S.final_cost;
    if (res.num_failures) res.num_failures[p] = int(S.num_failures);
    if (res.num_consec_failures) res.num_consec_failures[p] = int(S.num_consec);
    if (res.final_num_residuals) res.final_num_residuals[p] = S.final_nres;
    if (res.final_rerr_dec) res.final_rerr_dec[p] = S.final_rerr;
    if (res.final_inlier_ratio) res.final_inlier_ratio[p] = 1.0f;
    if (prm->counters) {
      atomicAdd(&prm->counters[0], S.acc_passes);
      atomicAdd(&prm->counters[1], S.eval_passes);
      atomicAdd(&prm->counters[2], S.solves);
      atomicAdd(&prm->counters[3], 1ull);
    }
  }
}

template <typename T, int NBM, int THIN>
int launch_ba(toa_handle h, BaParams& prm) {
  const int n = 6 * prm.C;
  size_t pw = WaveLds<T>::bytes(n);
  pw = (pw + 15) & ~size_t(15);
  prm.lds_wave = int(pw);
  size_t lds = pw + (size_t(n) * n + 64 + 64 + size_t(24) * prm.C + size_t(36) * prm.C + 64 + 8) * sizeof(T) + 64;
  lds = (lds + 15) & ~size_t(15);
  prm.lds_gpart = int(lds);
  lds += size_t(4) * prm.C * 32 * sizeof(T);
  lds = (lds + 15) & ~size_t(15);
  prm.lds_part2 = 0;
  {  // second partial-Gram buffer for the fold, if two workgroups per CU still fit with it
    const size_t with2 = lds + (size_t(n) * n + 128) * sizeof(T);
#ifndef TOA_BA_ONE_PART   // A/B switch: the four-round fold
    if (with2 <= size_t(h->max_lds) / TOA_BA_WGS) { prm.lds_part2 = int(lds); lds = with2; }
#endif
  }
  if (lds > size_t(160 * 1024)) return toa_fail(TOA_E_UNSUPPORTED, "toa_ba_run: LDS footprint exceeds 160 KiB");
  const BaWork<T> wk(prm.C, prm.N);
  const size_t need = size_t(prm.P) * wk.total * sizeof(T);
  if (need > h->scratch_bytes) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->scratch) (void)hipFree(h->scratch);
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&h->scratch, need));
    h->scratch_bytes = need;
  }
  prm.work = h->scratch;
  static_assert(sizeof(BaParams) <= 1024, "parameter block too large");
  if (int rc = upload_params(h, &prm, sizeof(prm))) return rc;
  auto kern = ba_schur_kernel<T, NBM, THIN>;
  if (int rc = ensure_lds_attr(h, (const void*)kern, lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(unsigned(prm.P)), dim3(256), lds, h->stream, (const BaParams*)h->params_dev);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

template <typename T>
int launch_ba_any(toa_handle h, BaParams& prm) {
  const DenseRowLayout lay = DenseRowLayout::make(6 * prm.C, 4);
  switch (lay.nbm * 8 + lay.thin) {
    case 8 + 0: return launch_ba<T, 1, 0>(h, prm);    // C = 1, 2
    case 8 + 3: return launch_ba<T, 1, 3>(h, prm);    // C = 3
    case 16 + 0: return launch_ba<T, 2, 0>(h, prm);   // C = 4, 5
    case 24 + 0: return launch_ba<T, 3, 0>(h, prm);   // C = 6, 7
    case 24 + 1: return launch_ba<T, 3, 1>(h, prm);   // C = 8
    case 32 + 0: return launch_ba<T, 4, 0>(h, prm);   // C = 9, 10
  }
  return toa_fail(TOA_E_UNSUPPORTED, "toa_ba_run: no kernel for this camera count");
}

}  // namespace toa

using namespace toa;

extern "C" int toa_ba_run(toa_handle h, int dtype, int num_cameras, int num_points, int64_t P, const void* data_dev, void* x_dev,
                          const toa_options* options, const toa_results* results, uint64_t* counters_dev) {
  if (!h) return toa_fail(TOA_E_ARG, "null handle");
  if (dtype != TOA_F32 && dtype != TOA_F64) return toa_fail(TOA_E_ARG, "dtype must be TOA_F32 or TOA_F64");
  if (num_cameras < 1 || num_cameras > 10) return toa_fail(TOA_E_ARG, "toa_ba_run: 1 <= num_cameras <= 10 (reduced camera system of one wavefront)");
  if (num_points < 1 || num_points > (1 << 22)) return toa_fail(TOA_E_ARG, "toa_ba_run: num_points out of range");
  if (P < 0 || P > 65535) return toa_fail(TOA_E_ARG, "toa_ba_run: P must be in [0, 65535]");
  if (!data_dev || !x_dev || !options || !results) return toa_fail(TOA_E_ARG, "toa_ba_run: null pointer");
  if (!results->stop_reason || !results->num_iters || !results->final_cost)
    return toa_fail(TOA_E_ARG, "toa_ba_run: stop_reason, num_iters and final_cost outputs are required");
  if (options->solver_type != 0 && options->solver_type != 1) return toa_fail(TOA_E_ARG, "toa_ba_run: solver_type must be 0 (LM) or 1 (GN)");
  if (!options->use_ldlt) return toa_fail(TOA_E_UNSUPPORTED, "toa_ba_run: use_ldlt=false is not available");
  if ((results->errs || results->deltas2 || results->successes) && results->hist_stride < options->max_iters + 2)
    return toa_fail(TOA_E_ARG, "toa_ba_run: hist_stride must be >= max_iters + 2");
  if (options->max_iters < 0 || options->max_iters > 65535) return toa_fail(TOA_E_ARG, "max_iters out of range");
  if (h->loss != TOA_LOSS_L2)   // sticky handle state must not be ignored silently (include/tinyopt_amd.h, toa_set_loss)
    return toa_fail(TOA_E_UNSUPPORTED, "toa_ba_run: bundle adjustment has no M-estimator and a loss is set on the handle (toa_set_loss); clear it first");
  if (P == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  BaParams prm;
  std::memset(&prm, 0, sizeof(prm));
  prm.data = data_dev; prm.x = x_dev; prm.P = P; prm.C = num_cameras; prm.N = num_points;
  prm.opt = *options; prm.res = *results;
  prm.res.final_hessian = nullptr;   // the block Hessian is not exported
  prm.counters = reinterpret_cast<unsigned long long*>(counters_dev);
  return dtype == TOA_F32 ? launch_ba_any<float>(h, prm) : launch_ba_any<double>(h, prm);
}
