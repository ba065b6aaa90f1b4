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
  int loss;                   // TOA_LOSS_* of the handle (toa_set_loss; the ROBUST instantiation of the kernel only)
  double loss_th2;
};

template <typename T>
struct BaWork {  // element offsets into a scene's scratch block
  size_t W, Voff, hdp, gp, Rinv, q, dp, ldp, ptsb, wgt, total;
  int nb64;
  __host__ __device__ BaWork(int C, int N, bool robust = false) {
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
    wgt = o; if (robust) o += size_t(C) * N;   // M-estimator scale s of every observation at the last build (0 = not seen): W is re-formed, not stored
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
// ROBUST: every observation's |r|^2 goes through the handle's M-estimator (toa_set_loss; losses/robust_norms.h:20-26 "JtJ * dx =
// Jt*res*s"): cost += l, the observation's rows of [J_c | J_p | r] scaled by sqrt(s) before the Grams, W scaled by s where it
// is re-formed, inliers = observations with |r|^2 <= th^2 (cost.h:84-95).  A separate instantiation (its own translation unit,
// -DTOA_BA_ROBUST_TU): the estimators' exp / log / atan2 would cost the plain kernel registers it does not have.
template <typename T, int NBM, int THIN, bool ROBUST>
__global__ void __launch_bounds__(256, TOA_BA_WGS) ba_schur_kernel(const BaParams* __restrict__ prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = prm->C, N = prm->N, n = 6 * C;
#ifdef TOA_BA_TIMING
  unsigned long long tk_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const long long p = blockIdx.x;
  const BaWork<T> wk(C, N, ROBUST);
  const int nb64 = wk.nb64;
  const int loss = ROBUST ? prm->loss : TOA_LOSS_L2;
  const T th2 = T(prm->loss_th2);
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
  GT* wgt = work + wk.wgt;
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
    T csum = 0, nvis = 0, ninl = 0;
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
          T lobs = T(0);
          if constexpr (ROBUST) {
            const T n2 = r[0] * r[0] + r[1] * r[1];
            T l, sw;
            robust_norm(loss, n2, th2, l, sw);
            lobs = seen ? l : T(0);                         // (l(0) is not 0 for every estimator: an unseen pair has no cost)
            ninl += (seen && n2 <= th2) ? T(2) : T(0);
            const T sq = r_sqrt(sw);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
              for (int k = 0; k < 6; ++k) Jc[a][k] *= sq;
#pragma unroll
              for (int k = 0; k < 3; ++k) Jp[a][k] *= sq;
            }
            const T rs0 = r[0] * sq, rs1 = r[1] * sq;
            if constexpr (kRecomputeW) wgt[size_t(c) * N + j] = seen ? sw : T(0);
            r[0] = rs0; r[1] = rs1;
            csum += lobs;
          }
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
          if constexpr (ROBUST) {
            const T n2 = r[0] * r[0] + r[1] * r[1];
            T l, sw;
            robust_norm(loss, n2, th2, l, sw);
            csum += l;
            ninl += n2 <= th2 ? T(2) : T(0);
          }
        }
        if constexpr (!ROBUST) csum += r[0] * r[0] + r[1] * r[1];
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
    int ninl_all = nres;
    if constexpr (ROBUST) ninl_all = int(ba_block_sum<T>(ninl, red));
    __syncthreads();
    // ================= rest of Build (lm.h:59-120): validity, clipping, diagonal check, Marquardt damping =================
    if (tid == 0) {
      if (do_acc) S.acc_passes++; else S.eval_passes++;
      S.cost_val = normalize_cost(double(cost_raw), nres, opt);
      S.cost_nres = nres;
      S.cost_ninl = ninl_all;
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
          // not positive definite: the solve fails — unless use_ldlt = false, "dx = -H.inverse() * g without any checks"
          // (gn.h:157-162, options.h:59): no verdict there; a step that comes out non-finite is refused further down as everywhere
          if (opt.use_ldlt && (!(l00s > T(0)) || !(l11s > T(0)) || !(l22s > T(0)))) bad += T(1);
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
            const T vs = ROBUST ? wgt[ob] : vis[ob];   // ROBUST: the observation's scale s at the build (0 = not seen)
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
              for (int b2 = 0; b2 < 3; ++b2) wrec[i][b2] = use ? (ROBUST ? wrec[i][b2] * cur.vs : wrec[i][b2]) : T(0);
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
      // use_ldlt = false: the reduced system by the PIVOTED factorisation with its verdict ignored — the solution of every
      // non-singular system, definite or not, to rounding (the n <= 63 paths do the same, lm_device.hpp)
      const bool unchecked = !opt.use_ldlt;
      const bool wg_ok = unchecked ? false : WgLdlt<T, NBS>::factor(L.M, L.LD, n, L.tmp, tid);
      if (wave == 0) {
        bool ok = wg_ok;
        if (unchecked) {
          const T rhs = lane < n ? L.vec[lane] : T(0);
          (void)ldlt_factor_wave<T>(L.M, L.LD, L.perm, L.tmp, n, lane);
          L.dx[lane] = ldlt_solve_wave<T>(L.M, L.LD, L.perm, L.vec, n, lane, rhs);
          ok = true;
        } else if (ok) {
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
            const T sobs = ROBUST ? wgt[size_t(c) * N + j] : vis[size_t(c) * N + j];
            const bool seen = sobs != T(0);
            ba_obs<T, true>(Pm, qb, f, T(0), T(0), T(0), T(0), rr, Jc, Jp);   // the Jacobians do not depend on cx, cy, u, v
#pragma unroll
            for (int kk = 0; kk < 6; ++kk)
#pragma unroll
              for (int b = 0; b < 3; ++b) { const T w0 = ba_w<T>(Jc[0][kk], Jc[1][kk], Jp[0][b], Jp[1][b]); wl[3 * kk + b] = seen ? (ROBUST ? w0 * sobs : w0) : T(0); }
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
    if (res.final_inlier_ratio) res.final_inlier_ratio[p] = S.final_nres > 0 ? float(S.final_ninl) / float(S.final_nres) : 1.0f;
    if (prm->counters) {
      atomicAdd(&prm->counters[0], S.acc_passes);
      atomicAdd(&prm->counters[1], S.eval_passes);
      atomicAdd(&prm->counters[2], S.solves);
      atomicAdd(&prm->counters[3], 1ull);
    }
  }
}

template <typename T, int NBM, int THIN, bool ROBUST>
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
  const BaWork<T> wk(prm.C, prm.N, ROBUST);
  const size_t need = size_t(prm.P) * wk.total * sizeof(T);
  if (need > h->scratch_bytes) {
    if (int rc = grow_sync(h, "device workspace")) return rc;
    toa_release_workspace(h, h->scratch);
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&h->scratch, need));
    h->scratch_bytes = need;
  }
  prm.work = h->scratch;
  static_assert(sizeof(BaParams) <= 1024, "parameter block too large");
  if (int rc = upload_params(h, &prm, sizeof(prm))) return rc;
  auto kern = ba_schur_kernel<T, NBM, THIN, ROBUST>;
  if (int rc = ensure_lds_attr(h, (const void*)kern, lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(unsigned(prm.P)), dim3(256), lds, h->stream, (const BaParams*)h->params_dev);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

template <typename T, bool ROBUST>
int launch_ba_any(toa_handle h, BaParams& prm) {
  const DenseRowLayout lay = DenseRowLayout::make(6 * prm.C, 4);
  switch (lay.nbm * 8 + lay.thin) {
    case 8 + 0: return launch_ba<T, 1, 0, ROBUST>(h, prm);    // C = 1, 2
    case 8 + 3: return launch_ba<T, 1, 3, ROBUST>(h, prm);    // C = 3
    case 16 + 0: return launch_ba<T, 2, 0, ROBUST>(h, prm);   // C = 4, 5
    case 24 + 0: return launch_ba<T, 3, 0, ROBUST>(h, prm);   // C = 6, 7
    case 24 + 1: return launch_ba<T, 3, 1, ROBUST>(h, prm);   // C = 8
    case 32 + 0: return launch_ba<T, 4, 0, ROBUST>(h, prm);   // C = 9, 10
  }
  return toa_fail(TOA_E_UNSUPPORTED, "toa_ba_run: no kernel for this camera count");
}
// the ROBUST instantiations live in their own translation unit (the same source with -DTOA_BA_ROBUST_TU)
int toa_ba_launch_robust(toa_handle h, int dtype, BaParams& prm);
#ifdef TOA_BA_ROBUST_TU
int toa_ba_launch_robust(toa_handle h, int dtype, BaParams& prm) {
  return dtype == TOA_F32 ? launch_ba_any<float, true>(h, prm) : launch_ba_any<double, true>(h, prm);
}
#endif


#ifndef TOA_BA_ROBUST_TU   // (everything below exists once, in the plain translation unit)
// =====================================================================================================================
// Bundle adjustment with VISIBILITY LISTS (round 3): tens to hundreds of cameras, each point seen by a few of them.
//
// The one-workgroup kernel above keeps the whole reduced camera system (6C <= 60 unknowns) of a scene in LDS and walks a
// dense C x N visibility mask.  A real multi-pose problem is the opposite shape — C in the tens or hundreds, N in the
// thousands, every point observed by a handful of cameras (the structure Eigen's SimplicialLDLT exists for, math.h:266-277;
// README.md:30,165-167) — so this path takes the observations as a LIST and is a pipeline of small kernels per Build + Solve
// attempt, with the reduced camera system S (6C x 6C) in HBM:
//
//   bl_obs      thread / observation   r, J_c (2 x 6), J_p (2 x 3) at the current x (build), r^2 (always)
//   bl_point    thread / point         V_j = sum J_p^T J_p, g_pj = sum J_p^T r over the point's observations, in list order
//   bl_cam      workgroup / camera     U_c = sum J_c^T J_c, g_c = sum J_c^T r over the camera's observations, fixed-order fold
//   bl_build    workgroup / scene      cost, validity, clipping, diagonal check, Marquardt scale (lm.h:59-120)
//   bl_psolve   thread / point         damp V_j, V_j^-1 (Cholesky), q_j = V_j^-1 g_pj
//   bl_schur    workgroup / camera c   block row c of  S = U - sum_j W_cj V_j^-1 W_c'j^T  and of  g_c - W V^-1 g_p : block
//                                      products staged in LDS, lane (component, observer) accumulates into the LDS row of the
//                                      observer's camera; the camera's observations are walked in order (no atomics)
//   solve       toa_large_solve_each   S dc = -(g_c - W V^-1 g_p): workgroup LDL^T up to 128 unknowns, the one-workgroup blocked
//                                      Cholesky up to 512 (fp64) / 1024 (fp32), rocSOLVER potrf beyond
//   bl_back     thread / point         dp_j = -V_j^-1 (g_pj + sum W_ij^T dc), |dx|^2, |g|^2 partials
//   bl_step     workgroup / scene      the rest of Step + the loop body of OptimizeAcc (optimizer.h:266-310, 370-539): the same
//                                      scalar state machine as every other path (lm_judge_core, lm_good_step / lm_bad_step)
//   bl_update   thread / pose, point   x (+)= dx or the roll-back (sophus.h:24-26, traits.h:184-190)
//
// W_ij = J_c^T J_p is never stored: every site contracts through the 2-vector J_c dc / the 2 x 3 J_p instead (18 values
// per observation less HBM traffic).  Every sum has a fixed order: results are bit-reproducible and a scene solved alone
// equals its row in a batch.  The host reads ONE integer back per pass (is any scene still running), which is also where
// max_duration_ms is honoured (kTimedOut, optimizer.h:302-305).
// Observations: sorted by (point, camera), each pair at most once.
}  // namespace toa
int toa_large_solve(toa_handle h, int dtype, int n, int64_t P, const void* H, const void* g, double scale, void* dx, int32_t* ok);
int toa_large_solve_inplace(toa_handle h, int dtype, int n, int64_t P, void* H, const void* g, void* dx, int32_t* ok);
int toa_large_solve_each(toa_handle h, int dtype, int n, int64_t P, const void* H, const void* g, double scale, void* dx, int32_t* ok);
int toa_large_solve_unchecked(toa_handle h, int dtype, int n, int64_t P, const void* H, const void* g, double scale, void* dx, int32_t* ok);
namespace toa {
struct BlParams {
  const void* intr;           // [P][4]: f cx cy 0
  const int* obs_cam;         // [P][M]
  const int* obs_pt;          // [P][M]
  const void* obs_uv;         // [P][M][2]
  void* x;                    // [P][12 C + 3 N]
  void* work;                 // per scene: see BlWork
  int* iwork;                 // per scene: pt_start [N + 1], cam_start [C + 1], cam_order [M], flags [16]
  long long P;
  int C, N, M;
  toa_options opt;
  toa_results res;
  unsigned long long* counters;
  int* any_active;            // one int for the host
  void* Sall;                 // [P][6C][6C] reduced camera systems, [P][6C] right-hand sides and solutions: contiguous over the
  void* rhsall;               // batch so that ONE call of the batched solver serves every scene
  void* dcall;
  int loss;                   // TOA_LOSS_* of the handle (toa_set_loss): each observation's |r|^2 through the M-estimator
  double loss_th2;
};

template <typename T>
struct BlWork {  // element offsets into a scene's scratch block (T)
  size_t Jc, Jp, r, r2, inl, ptcost, ptinl, Vd, Voff, gp, Vinv, q, dp, ldp, U, gc, Ud, ldc, pd2, pg2, state, total;
  __host__ __device__ BlWork(int C, int N, int M) {
    size_t o = 0;
    const size_t n = size_t(6) * C;
    auto take = [&](size_t k) { const size_t at = o; o += (k + 7) & ~size_t(7); return at; };
    // per observation ONE 128-byte record [J_c (2 x 6) | r (2) | pad] and ONE 64-byte record [J_p (2 x 3) | pad]: the camera-wise
    // kernels (bl_cam, bl_schur) GATHER observations, and component-major arrays cost them a 64-byte HBM fetch per 8-byte
    // component (PMC: 374 + 244 MB per solve in those two kernels for ~6 MB of records per scene and pass)
    Jc = take(size_t(16) * M); Jp = take(size_t(8) * M); r = Jc; r2 = take(M); inl = take(M);
    ptcost = take(N); ptinl = take(N); Vd = take(size_t(3) * N); Voff = take(size_t(3) * N); gp = take(size_t(3) * N); Vinv = take(size_t(6) * N);
    q = take(size_t(3) * N); dp = take(size_t(3) * N); ldp = take(size_t(3) * N);
    U = take(size_t(36) * C); gc = take(n); Ud = take(n); ldc = take(n);
    pd2 = take(N); pg2 = take(N);
    state = take((sizeof(LmState<T>) + sizeof(T) - 1) / sizeof(T) + 8);
    total = o;
  }
};
struct BlIdx {  // int offsets into a scene's index block
  size_t pt_start, cam_start, cam_order, flags, total;
  __host__ __device__ BlIdx(int C, int N, int M) {
    size_t o = 0;
    pt_start = o; o += size_t(N) + 1;
    cam_start = o; o += size_t(C) + 1;
    cam_order = o; o += size_t(M);
    flags = o; o += 16;   // [0] continue, [1] do_acc of the NEXT pass, [2] action, [3] built, [4] bad points, [5] solve ok (int32 from the solver)
    total = (o + 15) & ~size_t(15);
  }
};

// Index structures of a scene (setup, once per solve): pt_start (CSR over the point-sorted list), the cameras' observation
// counts, and — a wave per camera, stable — each camera's observations in increasing point order.  (The first version did all
// of it with ONE thread: 24.5 ms per call at 30 000 observations, 45 % of a 53 ms solve — profiles/r03_ab_log.md.)
__global__ void __launch_bounds__(256) bl_index_a_kernel(const BlParams* __restrict__ prm) {
  const long long p = blockIdx.y;
  const int C = prm->C, N = prm->N, M = prm->M;
  const BlIdx ix(C, N, M);
  int* iw = prm->iwork + size_t(p) * ix.total;
  const int* oc = prm->obs_cam + size_t(p) * M;
  const int* op = prm->obs_pt + size_t(p) * M;
  int* ps = iw + ix.pt_start;
  int* fl = iw + ix.flags;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const int pt = op[i], cm = oc[i];
  if (pt < 0 || pt >= N || cm < 0 || cm >= C) { atomicOr(&fl[6], 1); return; }
  if (i > 0) {
    const int pp = op[i - 1];
    if (pp > pt || (pp == pt && oc[i - 1] >= cm)) { atomicOr(&fl[6], 1); return; }   // sorted by (point, camera), no duplicates
    if (pp >= 0 && pp < pt) for (int j = pp + 1; j <= pt; ++j) ps[j] = i;          // the first observation of point pt (and of the empty points before it)
  } else {
    for (int j = 0; j <= pt; ++j) ps[j] = 0;
  }
  if (i == M - 1) for (int j = pt + 1; j <= N; ++j) ps[j] = M;
  atomicAdd(&iw[ix.cam_start + cm + 1], 1);
}
__global__ void bl_index_b_kernel(const BlParams* __restrict__ prm) {   // exclusive scan of the cameras' counts: C <= 682 values
  if (threadIdx.x != 0) return;
  const long long p = blockIdx.x;
  const BlIdx ix(prm->C, prm->N, prm->M);
  int* cs = prm->iwork + size_t(p) * ix.total + ix.cam_start;
  cs[0] = 0;
  for (int c = 0; c < prm->C; ++c) cs[c + 1] += cs[c];
}
__global__ void __launch_bounds__(64) bl_index_c_kernel(const BlParams* __restrict__ prm) {
  const long long p = blockIdx.y;
  const int c = blockIdx.x, lane = threadIdx.x;
  const int M = prm->M;
  const BlIdx ix(prm->C, prm->N, M);
  int* iw = prm->iwork + size_t(p) * ix.total;
  if (iw[ix.flags + 6]) return;
  const int* oc = prm->obs_cam + size_t(p) * M;
  int* co = iw + ix.cam_order;
  int pos = iw[ix.cam_start + c];
  // eight trips' worth of camera ids in flight at a time (a dependent load per trip was 0.28 us x 469 trips = 130 us per call
  // at 30 000 observations — 2 % of a four-scene solve spent building an index); the list order is unchanged
  for (int i0 = 0; i0 < M; i0 += 64 * 8) {
    int cam[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u + lane; cam[u] = i < M ? oc[i] : -1; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool mine = cam[u] == c;
      const unsigned long long mask = __ballot(mine);
      if (mine) co[pos + __popcll(mask & ((1ull << lane) - 1ull))] = i0 + 64 * u + lane;
      pos += __popcll(mask);
    }
  }
}

template <typename T>
__device__ __forceinline__ LmState<T>* bl_state(const BlParams* prm, const BlWork<T>& wk, long long p) {
  return reinterpret_cast<LmState<T>*>(static_cast<T*>(prm->work) + size_t(p) * wk.total + wk.state);
}

template <typename T>
__global__ void __launch_bounds__(64) bl_init_kernel(const BlParams* __restrict__ prm) {
  const long long p = blockIdx.x;
  const BlWork<T> wk(prm->C, prm->N, prm->M);
  const BlIdx ix(prm->C, prm->N, prm->M);
  LmState<T>& S = *bl_state<T>(prm, wk, p);
  if (threadIdx.x == 0) {
    const toa_options& opt = prm->opt;
    S.lambda = opt.damping_init; S.prev_lambda = 0; S.bad_factor = opt.bad_factor; S.rebuild = 1;
    S.final_cost = kDblMax; S.final_nres = 0; S.final_ninl = 0; S.cost_ninl = 0; S.final_rerr = kDblMax;
    S.stop = TOA_STOP_NONE; S.num_iters = 0; S.num_failures = 0; S.num_consec = 0;
    S.cost_val = 0; S.cost_nres = 0;
    S.max_iters = opt.max_iters + 1 + (opt.check_final_cost ? 1 : 0);
    S.has_last_dx = 0; S.last_was_success = 1; S.iter = 0;
    S.acc_passes = 0; S.eval_passes = 0; S.solves = 0; S.problems = 0;
    int* fl = prm->iwork + size_t(p) * ix.total + ix.flags;
    fl[0] = 1; fl[1] = 1;
    if (fl[6]) {   // malformed observation list: nothing is optimised, the Output of an untouched problem with kSkipped
      S.stop = TOA_STOP_SKIPPED;
      fl[0] = 0;
      const toa_results& res = prm->res;
      res.stop_reason[p] = TOA_STOP_SKIPPED;
      res.num_iters[p] = 0;
      res.final_cost[p] = kDblMax;
      if (res.num_failures) res.num_failures[p] = 0;
      if (res.num_consec_failures) res.num_consec_failures[p] = 0;
      if (res.final_num_residuals) res.final_num_residuals[p] = 0;
      if (res.final_rerr_dec) res.final_rerr_dec[p] = kDblMax;
      if (res.final_inlier_ratio) res.final_inlier_ratio[p] = 1.0f;
      if (prm->counters) atomicAdd(&prm->counters[3], 1ull);
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) bl_obs_kernel(const BlParams* __restrict__ prm) {
  const long long p = blockIdx.y;
  const int C = prm->C, N = prm->N, M = prm->M;
  const BlWork<T> wk(C, N, M);
  const BlIdx ix(C, N, M);
  const int* fl = prm->iwork + size_t(p) * ix.total + ix.flags;
  if (!fl[0]) return;
  const bool do_acc = fl[1] != 0;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const T* intr = static_cast<const T*>(prm->intr) + size_t(p) * 4;
  const T* X = static_cast<const T*>(prm->x) + size_t(p) * (size_t(12) * C + size_t(3) * N);
  T* w = static_cast<T*>(prm->work) + size_t(p) * wk.total;
  const int cm = prm->obs_cam[size_t(p) * M + i], pt = prm->obs_pt[size_t(p) * M + i];
  const T* uv = static_cast<const T*>(prm->obs_uv) + (size_t(p) * M + i) * 2;
  T Pm[12], q[3], r[2];
#pragma unroll
  for (int k = 0; k < 12; ++k) Pm[k] = X[12 * cm + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) q[k] = X[size_t(12) * C + size_t(3) * pt + k];
  // M-estimator (toa_set_loss; losses/robust_norms.h:20-26 "JtJ * dx = Jt*res*s"): the observation's cost is l(|r|^2) and its
  // rows of [J_c | J_p | r] are stored scaled by sqrt(s), s = dl/dn2 — every later kernel (bl_point, bl_cam, bl_schur, bl_back)
  // contracts the stored rows, so J^T J, J^T r and W all come out scaled by s without knowing about the loss
  const int loss = prm->loss;
  const T th2 = T(prm->loss_th2);
  if (do_acc) {
    T Jc[2][6], Jp[2][3];
    ba_obs<T, true>(Pm, q, intr[0], intr[1], intr[2], uv[0], uv[1], r, Jc, Jp);
    const T n2 = r[0] * r[0] + r[1] * r[1];
    T l = n2, sq = T(1);
    if (loss != TOA_LOSS_L2) {
      T sw;
      robust_norm(loss, n2, th2, l, sw);
      sq = r_sqrt(sw);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int k = 0; k < 6; ++k) w[wk.Jc + size_t(i) * 16 + (6 * a + k)] = loss != TOA_LOSS_L2 ? sq * Jc[a][k] : Jc[a][k];
#pragma unroll
      for (int k = 0; k < 3; ++k) w[wk.Jp + size_t(i) * 8 + (3 * a + k)] = loss != TOA_LOSS_L2 ? sq * Jp[a][k] : Jp[a][k];
      w[wk.Jc + size_t(i) * 16 + 12 + (a)] = loss != TOA_LOSS_L2 ? sq * r[a] : r[a];
    }
    w[wk.r2 + i] = l;
    w[wk.inl + i] = (loss == TOA_LOSS_L2 || n2 <= th2) ? T(2) : T(0);   // cost.h:84-95: both residuals of an inlier observation
  } else {
    ba_obs<T, false>(Pm, q, intr[0], intr[1], intr[2], uv[0], uv[1], r, nullptr, nullptr);
    const T n2 = r[0] * r[0] + r[1] * r[1];
    T l = n2;
    if (loss != TOA_LOSS_L2) { T sw; robust_norm(loss, n2, th2, l, sw); }
    w[wk.r2 + i] = l;
    w[wk.inl + i] = (loss == TOA_LOSS_L2 || n2 <= th2) ? T(2) : T(0);
  }
}

template <typename T>
__device__ __forceinline__ void bl_point_body(const BlParams* __restrict__ prm, const int bx) {   // block bx of the scene's points
  const long long p = blockIdx.y;
  const int C = prm->C, N = prm->N, M = prm->M;
  const BlWork<T> wk(C, N, M);
  const BlIdx ix(C, N, M);
  const int* iw = prm->iwork + size_t(p) * ix.total;
  if (!iw[ix.flags + 0]) return;
  const bool do_acc = iw[ix.flags + 1] != 0;
  const int j = bx * 256 + threadIdx.x;
  if (j >= N) return;
  T* w = static_cast<T*>(prm->work) + size_t(p) * wk.total;
  const int i0 = iw[ix.pt_start + j], i1 = iw[ix.pt_start + j + 1];
  T cost = 0, inl = 0;
  T v[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  for (int i = i0; i < i1; ++i) {
    cost += w[wk.r2 + i];
    inl += w[wk.inl + i];
    if (do_acc) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const T j0 = w[wk.Jp + size_t(i) * 8 + (3 * a)], j1 = w[wk.Jp + size_t(i) * 8 + (3 * a + 1)], j2 = w[wk.Jp + size_t(i) * 8 + (3 * a + 2)];
        const T ra = w[wk.Jc + size_t(i) * 16 + 12 + (a)];
        v[0] += j0 * j0; v[1] += j1 * j1; v[2] += j2 * j2; v[3] += j0 * j1; v[4] += j0 * j2; v[5] += j1 * j2;
        g[0] += j0 * ra; g[1] += j1 * ra; g[2] += j2 * ra;
      }
    }
  }
  w[wk.ptcost + j] = cost;
  w[wk.ptinl + j] = inl;
  if (do_acc) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { w[wk.Vd + 3 * j + k] = v[k]; w[wk.Voff + 3 * j + k] = v[3 + k]; w[wk.gp + 3 * j + k] = g[k]; }
  }
}

template <typename T>
__device__ __forceinline__ void bl_cam_body(const BlParams* __restrict__ prm, const int c, T (&red)[4][32]) {
  const long long p = blockIdx.y;
  const int C = prm->C, N = prm->N, M = prm->M;
  const BlWork<T> wk(C, N, M);
  const BlIdx ix(C, N, M);
  const int* iw = prm->iwork + size_t(p) * ix.total;
  if (!iw[ix.flags + 0] || !iw[ix.flags + 1]) return;   // build passes only
  T* w = static_cast<T*>(prm->work) + size_t(p) * wk.total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = iw[ix.cam_start + c], k1 = iw[ix.cam_start + c + 1];
  T G[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) G[t] = T(0);
  for (int k = k0 + tid; k < k1; k += 256) {
    const int i = iw[ix.cam_order + k];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      T w7[7];
#pragma unroll
      for (int d = 0; d < 6; ++d) w7[d] = w[wk.Jc + size_t(i) * 16 + (6 * a + d)];
      w7[6] = w[wk.Jc + size_t(i) * 16 + 12 + (a)];
      int t = 0;
#pragma unroll
      for (int x = 0; x < 7; ++x)
#pragma unroll
        for (int y = x; y < 7; ++y) G[t++] += w7[x] * w7[y];
    }
  }
  const T tot = wave_transposed_reduce32(G, lane);   // lane t < 28 holds this wave's total of sum t
  if (lane < 32) red[wave][lane] = tot;
  __syncthreads();
  if (tid < 28) {
    const T s = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    int a = 0, rem = tid;  // tt(a, b) = a * 7 - a (a - 1) / 2 + (b - a)
    while (rem >= 7 - a) { rem -= 7 - a; ++a; }
    const int b = a + rem;
    if (b < 6) { w[wk.U + 36 * c + 6 * a + b] = s; w[wk.U + 36 * c + 6 * b + a] = s; if (a == b) w[wk.Ud + 6 * c + a] = s; }
    else if (a < 6) w[wk.gc + 6 * c + a] = s;
  }
}

// The points' 3 x 3 blocks and the cameras' 6 x 6 blocks read the observation records and nothing of each other: ONE launch (round 5;
// two launches of 9 and 7 us, each mostly launch latency): blocks [0, gN) are bl_point_body's, blocks [gN, gN + C) a camera each.
template <typename T>
__global__ void __launch_bounds__(256) bl_point_cam_kernel(const BlParams* __restrict__ prm, const int gN) {
  __shared__ T red[4][32];
  if (int(blockIdx.x) < gN) bl_point_body<T>(prm, int(blockIdx.x));
  else bl_cam_body<T>(prm, int(blockIdx.x) - gN, red);
}

// fixed-order sum of arr[0 .. count) by the 256 threads of a block
template <typename T>
__device__ __forceinline__ T bl_block_sum(const T* arr, int count, T* red4) {
  T v = 0;
  for (int i = threadIdx.x; i < count; i += 256) v += arr[i];
  return ba_block_sum<T>(v, red4);
}

template <typename T>
__global__ void __launch_bounds__(256) bl_build_kernel(const BlParams* __restrict__ prm) {
  __shared__ T red[8];
  __shared__ int sflag;
  const long long p = blockIdx.x;
  const int C = prm->C, N = prm->N, M = prm->M, n = 6 * C;
  const BlWork<T> wk(C, N, M);
  const BlIdx ix(C, N, M);
  int* fl = prm->iwork + size_t(p) * ix.total + ix.flags;
  if (!fl[0]) return;
  const bool do_acc = fl[1] != 0;
  T* w = static_cast<T*>(prm->work) + size_t(p) * wk.total;
  LmState<T>& S = *bl_state<T>(prm, wk, p);
  const toa_options& opt = prm->opt;
  const bool is_lm = opt.solver_type == 0;
  const int tid = threadIdx.x;
  const T cost_raw = bl_block_sum<T>(w + wk.ptcost, N, red);
  __syncthreads();
  const T inl_sum = prm->loss != TOA_LOSS_L2 ? bl_block_sum<T>(w + wk.ptinl, N, red) : T(2 * M);   // (exact in T: small integers)
  __syncthreads();
  if (tid == 0) {
    if (do_acc) S.acc_passes++; else S.eval_passes++;
    S.cost_val = normalize_cost(double(cost_raw), 2 * M, opt);
    S.cost_nres = 2 * M;
    S.cost_ninl = int(inl_sum);
    sflag = (M > 0 && S.cost_val != kDblMax) ? 1 : 0;
  }
  __syncthreads();
  bool built = sflag != 0;
  if (built && do_acc) {
    if (opt.grad_clipping != 0) {  // base.h:29-38
      const T mm = opt.grad_clipping;
      for (int i = tid; i < n; i += 256) w[wk.gc + i] = fmin(fmax(w[wk.gc + i], -mm), mm);
      for (int i = tid; i < 3 * N; i += 256) w[wk.gp + i] = fmin(fmax(w[wk.gp + i], -mm), mm);
    }
    if (opt.check_min_H_diag > 0) {  // lm.h:82-86, every diagonal entry of H
      T low = 0;
      for (int i = tid; i < n; i += 256) low += fabs(w[wk.Ud + i]) < T(opt.check_min_H_diag) ? T(1) : T(0);
      for (int i = tid; i < 3 * N; i += 256) low += fabs(w[wk.Vd + i]) < T(opt.check_min_H_diag) ? T(1) : T(0);
      if (ba_block_sum<T>(low, red) > T(0)) built = false;
    }
  }
  __syncthreads();
  if (built && is_lm && S.lambda > T(0)) {  // lm.h:108-117, s in double; the points' diagonal is scaled in bl_psolve
    const double s = S.rebuild ? 1.0 + double(S.lambda) : (1.0 + double(S.lambda)) / (1.0 + double(S.prev_lambda));
    for (int i = tid; i < n; i += 256) w[wk.Ud + i] = T(double(w[wk.Ud + i]) * s);
  }
  if (tid == 0) { fl[3] = built ? 1 : 0; fl[4] = 0; fl[5] = 0; }
}

template <typename T>
__global__ void __launch_bounds__(256) bl_psolve_kernel(const BlParams* __restrict__ prm) {
  const long long p = blockIdx.y;
  const int C = prm->C, N = prm->N, M = prm->M;
  const BlWork<T> wk(C, N, M);
  const BlIdx ix(C, N, M);
  int* fl = prm->iwork + size_t(p) * ix.total + ix.flags;
  if (!fl[0] || !fl[3]) return;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  T* w = static_cast<T*>(prm->work) + size_t(p) * wk.total;
  const LmState<T>& S = *bl_state<T>(prm, wk, p);
  const toa_options& opt = prm->opt;
  T a00 = w[wk.Vd + 3 * j], a11 = w[wk.Vd + 3 * j + 1], a22 = w[wk.Vd + 3 * j + 2];
  if (opt.solver_type == 0 && S.lambda > T(0)) {
    const double s = S.rebuild ? 1.0 + double(S.lambda) : (1.0 + double(S.lambda)) / (1.0 + double(S.prev_lambda));
    a00 = T(double(a00) * s); a11 = T(double(a11) * s); a22 = T(double(a22) * s);
    w[wk.Vd + 3 * j] = a00; w[wk.Vd + 3 * j + 1] = a11; w[wk.Vd + 3 * j + 2] = a22;
  }
  const T a01 = w[wk.Voff + 3 * j], a02 = w[wk.Voff + 3 * j + 1], a12 = w[wk.Voff + 3 * j + 2];
  // a point no camera sees has V = 0: like the dense LDL^T's zero pivots it takes a zero step (pseudo-inverse)
  const bool empty = a00 == T(0) && a11 == T(0) && a22 == T(0) && a01 == T(0) && a02 == T(0) && a12 == T(0);
  T r00 = 0, r10 = 0, r11 = 0, r20 = 0, r21 = 0, r22 = 0;   // R^-1 (lower), V = R R^T... V = L L^T, R^-1 = L^-1
  if (!empty) {
    const T l00s = a00;
    const T l00 = sqrt(l00s);
    const T l10 = a01 / l00, l20 = a02 / l00;
    const T l11s = a11 - l10 * l10;
    const T l11 = sqrt(l11s);
    const T l21 = (a12 - l20 * l10) / l11;
    const T l22s = a22 - l20 * l20 - l21 * l21;
    const T l22 = sqrt(l22s);
    if (prm->opt.use_ldlt && (!(l00s > T(0)) || !(l11s > T(0)) || !(l22s > T(0)))) atomicAdd(&fl[4], 1);   // not positive definite: the solve fails (use_ldlt = false: no verdict, gn.h:157-162)
    r00 = T(1) / l00; r11 = T(1) / l11; r22 = T(1) / l22;
    r10 = -l10 * r00 * r11;
    r21 = -l21 * r11 * r22;
    r20 = -(l20 * r00 + l21 * r10) * r22;
  }
  T* Rj = w + wk.Vinv + 6 * j;
  Rj[0] = r00; Rj[1] = r10; Rj[2] = r11; Rj[3] = r20; Rj[4] = r21; Rj[5] = r22;
  const T g0 = w[wk.gp + 3 * j], g1 = w[wk.gp + 3 * j + 1], g2 = w[wk.gp + 3 * j + 2];
  // q = V^-1 g_p = L^-T (L^-1 g_p)
  const T y0 = r00 * g0, y1 = r10 * g0 + r11 * g1, y2 = r20 * g0 + r21 * g1 + r22 * g2;
  w[wk.q + 3 * j] = r00 * y0 + r10 * y1 + r20 * y2;
  w[wk.q + 3 * j + 1] = r11 * y1 + r21 * y2;
  w[wk.q + 3 * j + 2] = r22 * y2;
}

// *p += v on an LDS word nobody else touches concurrently: ds_add_f32 / ds_add_f64 without return (IEEE add, as v_add would)
template <typename T>
__device__ __forceinline__ void bl_lds_add(T* p, const T v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// K consecutive elements of a 64-byte aligned observation record as 16-byte loads (round 5: bl_schur_kernel's B stage issued 42
// eight-byte loads per thread for the 18 values of a record pair: 8.3 of a 23.5 us chunk)
template <int K, typename T>
__device__ __forceinline__ void bl_load_rec(const T* __restrict__ src, T (&out)[K]) {
  constexpr int W = 16 / int(sizeof(T));
  using V = T __attribute__((ext_vector_type(W)));
  static_assert(K % W == 0, "whole 16-byte words");
#pragma unroll
  for (int q = 0; q < K / W; ++q) {
    const V v = *reinterpret_cast<const V*>(src + W * q);
#pragma unroll
    for (int k = 0; k < W; ++k) out[W * q + k] = v[k];
  }
}

// Block row c of the reduced camera system.  Thread c' (< C, in tiles of 256) owns block (c, c'); the camera's observations
// are walked in list order in chunks of 64 whose per-observation factors T_i = J_c,i^T (J_p,i V_j^-1) (6 x 3) are staged
// in LDS by the first 64 threads.  For observation i of point j, camera c' contributes iff it sees j: a binary search of
// the point's (camera-sorted) list.  S(c, c') -= T_i (J_p,i'^T J_c,i'),   red_c -= T_i g_pj  [red = g_c - W V^-1 g_p].
template <typename T>
__global__ void __launch_bounds__(256) bl_schur_kernel(const BlParams* __restrict__ prm, const int split, T* __restrict__ spart,
                                                       T* __restrict__ rpart) {
  // Round 4: with split > 1 (scenes whose cameras see >= 256 observations each, C <= 128 — a property of the scene, never of the
  // batch, so that a scene solved alone still gives the bits of its row in a batch) the observations of camera c are dealt to
  // `split` workgroups in contiguous runs; each leaves its partial block row (and partial right-hand side) in spart / rpart
  // and bl_schur_reduce_kernel sums them in run order and finishes the row.  Four scenes x 64 cameras are 256 workgroups of
  // ~470 observations each — one per compute unit, 0.255 ms per pass; dealt three ways (round 5; four in round 4) 768 workgroups are one round of three per compute unit.
  // One workgroup per camera c forms block row c of S (lower block triangle, mirrored) and of the reduced right-hand side.
  // Per staged chunk of 32 of the camera's observations (point j each):
  //   T_s  = J_c,s^T (J_p,s V_j^-1)                 6 x 3   thread s
  //   B_se = J_p,i2^T J_c,i2 for the e-th observer i2 of point j   3 x 6   thread (s, e), the first kFast observers of a point
  //   S(c, c2) -= T_s B_se  with c2 = camera of i2: lane (a d, e) adds component (a, d) of the e-th observer's product into the
  //   LDS row of that observer's camera. The six lanes of one (a, d) sit in ONE wave (9 components x 6 entries = 54 lanes per wave),
  //   a point's observers are distinct cameras, and a wave's LDS operations are in order — so no two lanes ever race on an address,
  //   every sum runs over s in the same fixed order, there are no atomics, and 216 of 256 lanes do 3 FMAs per observation.
  // (A thread per block (c, c2) with the product in registers — round 3's first version — executed the 108-FMA block product
  //  for ~6 of its wave's 64 lanes per observation: 0.66 ms per pass at 64 cameras, divergence, not memory.)
  constexpr int kS = 32, kFast = 6, kListCap = 12, kTile = 64;
  __shared__ T Tl[kS][18];
  __shared__ T Bl[kS][kFast][18];
  __shared__ T Srow[kTile][36];
  __shared__ T gl[kS][3];
  __shared__ short cl[kS][kListCap];
  __shared__ int jl[kS], l0[kS], ln[kS];
  const long long p = blockIdx.y;
  const int c = blockIdx.x / split, part = blockIdx.x % split;
  const int C = prm->C, N = prm->N, M = prm->M, n = 6 * C;
  const BlWork<T> wk(C, N, M);
  const BlIdx ix(C, N, M);
  const int* iw = prm->iwork + size_t(p) * ix.total;
  if (!iw[ix.flags + 0] || !iw[ix.flags + 3]) return;
  T* w = static_cast<T*>(prm->work) + size_t(p) * wk.total;
  T* Sg = static_cast<T*>(prm->Sall) + size_t(p) * n * n;
  T* rg = static_cast<T*>(prm->rhsall) + size_t(p) * n;
  const int* oc = prm->obs_cam + size_t(p) * M;
  const int tid = threadIdx.x;
  int k0 = iw[ix.cam_start + c], k1 = iw[ix.cam_start + c + 1];
  if (split > 1) {   // this workgroup's run of the camera's observations: whole staged chunks
    const int per = (((k1 - k0 + split - 1) / split) + 31) / 32 * 32;
    k0 = min(k1, k0 + part * per);
    k1 = min(k1, k0 + per);
  }
  T* sp = split > 1 ? spart + ((size_t(p) * C + c) * split + part) * size_t(C) * 36 : nullptr;
  const int lane = tid & 63, ent = lane % kFast;
  const bool live = lane < 54;
  const int ad = live ? 9 * (tid >> 6) + lane / kFast : 0, a = ad / 6, d = ad % 6;   // component (a, d), list entries ent, ent + 6, ...
  T rva = T(0);   // lanes (a, d = 0, entry 0): row a of W V^-1 g_p
#ifdef TOA_BL_TIMING
  unsigned long long tq[6] = {0, 0, 0, 0, 0, 0}, tp = wall_clock64();
#define BL_TICK(i) { __syncthreads(); const unsigned long long now_ = wall_clock64(); tq[i] += now_ - tp; tp = now_; }
#else
#define BL_TICK(i)
#endif
  for (int cbase = 0; cbase <= c; cbase += kTile) {                  // camera tiles of the lower block triangle
    for (int e = tid; e < kTile * 36; e += 256) (&Srow[0][0])[e] = T(0);
    BL_TICK(0)
    for (int kk = k0; kk < k1; kk += kS) {
      __syncthreads();
      if (tid < kS && kk + tid < k1) {
        const int i = iw[ix.cam_order + kk + tid];
        const int j = prm->obs_pt[size_t(p) * M + i];
        const int a0 = iw[ix.pt_start + j], a1 = iw[ix.pt_start + j + 1];
        jl[tid] = j;
        l0[tid] = a0;
        ln[tid] = a1 - a0;
        gl[tid][0] = w[wk.gp + 3 * j]; gl[tid][1] = w[wk.gp + 3 * j + 1]; gl[tid][2] = w[wk.gp + 3 * j + 2];
#pragma unroll
        for (int e = 0; e < kListCap; ++e) cl[tid][e] = short(a0 + e < a1 ? oc[a0 + e] : 32767);
        const T* Rj = w + wk.Vinv + 6 * j;
        const T r00 = Rj[0], r10 = Rj[1], r11 = Rj[2], r20 = Rj[3], r21 = Rj[4], r22 = Rj[5];
        // V^-1 = R^-T R^-1 with R^-1 lower: rows
        const T v00 = r00 * r00 + r10 * r10 + r20 * r20, v01 = r10 * r11 + r20 * r21, v02 = r20 * r22;
        const T v11 = r11 * r11 + r21 * r21, v12 = r21 * r22, v22 = r22 * r22;
        T jpi[8], jci[12];
        bl_load_rec<8>(w + wk.Jp + size_t(i) * 8, jpi);
        bl_load_rec<12>(w + wk.Jc + size_t(i) * 16, jci);
        T A[2][3];   // J_p,i V^-1 (2 x 3)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const T p0 = jpi[3 * q], p1 = jpi[3 * q + 1], p2 = jpi[3 * q + 2];
          A[q][0] = p0 * v00 + p1 * v01 + p2 * v02;
          A[q][1] = p0 * v01 + p1 * v11 + p2 * v12;
          A[q][2] = p0 * v02 + p1 * v12 + p2 * v22;
        }
#pragma unroll
        for (int dd = 0; dd < 6; ++dd) {
          const T jc0 = jci[dd], jc1 = jci[6 + dd];
#pragma unroll
          for (int b = 0; b < 3; ++b) Tl[tid][3 * dd + b] = jc0 * A[0][b] + jc1 * A[1][b];
        }
      }
      __syncthreads();
      BL_TICK(1)
      const int cnt = min(kS, k1 - kk);
      if (tid < kS * kFast) {   // B_se for the first kFast observers of every staged point
        const int s_ = tid / kFast, e_ = tid % kFast;
        if (s_ < cnt && e_ < ln[s_]) {
          const int i2 = l0[s_] + e_;
          T jp[8], jc[12];
          bl_load_rec<8>(w + wk.Jp + size_t(i2) * 8, jp);
          bl_load_rec<12>(w + wk.Jc + size_t(i2) * 16, jc);
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            const T p0 = jp[b], p1 = jp[3 + b];
#pragma unroll
            for (int dd = 0; dd < 6; ++dd) Bl[s_][e_][6 * b + dd] = p0 * jc[dd] + p1 * jc[6 + dd];
          }
        }
      }
      __syncthreads();
      BL_TICK(2)
      if (live) {
        // round 5: the operands of observation s + 1 (list length, its first observer's camera and product, T_s) are on their way
        // from LDS while observation s is added — the loop was one chain of ~4 dependent LDS round trips per observation (length ->
        // camera -> product / row -> write: 880 cycles, 11.7 of a 23.5 us chunk); what stays serial is the read-modify-write of
        // the row itself.  Same additions in the same order.
        struct Op { int len, c2; T t0, t1, t2, b0, b1, b2; };
        auto fetch = [&](const int s_) __attribute__((always_inline)) {
          Op o;
          o.len = ln[s_];
          o.t0 = Tl[s_][3 * a]; o.t1 = Tl[s_][3 * a + 1]; o.t2 = Tl[s_][3 * a + 2];
          o.c2 = int(cl[s_][ent]);                       // (ent < kFast <= kListCap; 32767 past the end of the list)
          o.b0 = Bl[s_][ent][d]; o.b1 = Bl[s_][ent][6 + d]; o.b2 = Bl[s_][ent][12 + d];   // (unused when ent >= len)
          return o;
        };
        T* const Sflat = &Srow[0][0];
        auto add_obs = [&](const int s_, const Op& cur) __attribute__((always_inline)) {
          const T t0 = cur.t0, t1 = cur.t1, t2 = cur.t2;
          {
            // (an LDS add without return instead of read - subtract - write: nothing of the row comes back into the loop.  A wave's LDS
            //  operations execute in order and no two lanes of the workgroup share an address, so every sum still runs in order s.
            //  Past the end of the point's list the staged camera is 32767 > c: no separate test of the list length.)
            const int c2 = cur.c2;
            if (!(c2 > c || c2 < cbase || c2 >= cbase + kTile)) bl_lds_add(Sflat + (unsigned(c2 - cbase) * 36u + unsigned(ad)), -(t0 * cur.b0 + t1 * cur.b1 + t2 * cur.b2));
          }
          for (int e_ = ent + kFast; e_ < cur.len; e_ += kFast) {   // a point seen by more cameras than the fast path stages
            int c2;
            if (e_ < kListCap) c2 = int(cl[s_][e_]); else c2 = oc[l0[s_] + e_];
            if (c2 > c || c2 < cbase || c2 >= cbase + kTile) continue;
            const int i2 = l0[s_] + e_;                  // its entry straight from the records
            const T jc0 = w[wk.Jc + size_t(i2) * 16 + (d)], jc1 = w[wk.Jc + size_t(i2) * 16 + (6 + d)];
            const T b0 = w[wk.Jp + size_t(i2) * 8 + 0] * jc0 + w[wk.Jp + size_t(i2) * 8 + 3] * jc1;
            const T b1 = w[wk.Jp + size_t(i2) * 8 + 1] * jc0 + w[wk.Jp + size_t(i2) * 8 + 4] * jc1;
            const T b2 = w[wk.Jp + size_t(i2) * 8 + 2] * jc0 + w[wk.Jp + size_t(i2) * 8 + 5] * jc1;
            bl_lds_add(Sflat + (unsigned(c2 - cbase) * 36u + unsigned(ad)), -(t0 * b0 + t1 * b1 + t2 * b2));
          }
          if (cbase == 0 && ent == 0 && d == 0) rva -= t0 * gl[s_][0] + t1 * gl[s_][1] + t2 * gl[s_][2];
        };
        // two observations per trip, each one's operands fetched while the other is added (no register copies)
        Op oa = fetch(0), ob;
        int s_ = 0;
        for (; s_ + 1 < cnt; s_ += 2) {
          ob = fetch(s_ + 1);
          add_obs(s_, oa);
          if (s_ + 2 < cnt) oa = fetch(s_ + 2);
          add_obs(s_ + 1, ob);
        }
        if (s_ < cnt) add_obs(s_, oa);
      }
      BL_TICK(3)
    }
    __syncthreads();
    if (split > 1) {   // the partial block row of this run, tile by tile
      for (int e = tid; e < kTile * 36; e += 256) {
        const int c2 = cbase + e / 36;
        if (c2 <= c && c2 < C) sp[size_t(c2) * 36 + e % 36] = Srow[c2 - cbase][e % 36];
      }
      __syncthreads();
      continue;
    }
    for (int e = tid; e < kTile * 36; e += 256) {
      const int c2 = cbase + e / 36, q = e % 36, qa = q / 6, qd = q % 6;
      if (c2 > c || c2 >= C) continue;
      T v = Srow[c2 - cbase][q];
      if (c2 == c) {   // the diagonal block: symmetrise the sum of T_i B_i (equal in exact arithmetic) + U_c with its damped diagonal
        v = (qd <= qa) ? Srow[c2 - cbase][qa * 6 + qd] : Srow[c2 - cbase][qd * 6 + qa];
        v += (qa == qd) ? w[wk.Ud + 6 * c + qa] : w[wk.U + 36 * c + 6 * qa + qd];
      }
      Sg[size_t(6 * c + qa) * n + 6 * c2 + qd] = v;
      if (c2 != c) Sg[size_t(6 * c2 + qd) * n + 6 * c + qa] = v;
    }
    __syncthreads();
  }
  BL_TICK(4)
#ifdef TOA_BL_TIMING
  if (tid == 0 && p == 0 && (blockIdx.x == 5 || blockIdx.x == 250)) printf("bl_schur wg %d: %d obs: zero %.1f stage %.1f B %.1f accumulate %.1f epilogue %.1f us\n", int(blockIdx.x), k1 - k0, tq[0] * 0.01, tq[1] * 0.01, tq[2] * 0.01, tq[3] * 0.01, tq[4] * 0.01);
#endif
  if (live && ent == 0 && d == 0) {
    if (split > 1) rpart[((size_t(p) * C + c) * split + part) * 6 + a] = rva;
    else rg[6 * c + a] = w[wk.gc + 6 * c + a] + rva;
  }
}

// split > 1: block row c of S and of the reduced right-hand side from the `split` partial rows, summed in run order
template <typename T>
__global__ void __launch_bounds__(256) bl_schur_reduce_kernel(const BlParams* __restrict__ prm, const int split, const T* __restrict__ spart,
                                                              const T* __restrict__ rpart) {
  constexpr int kTile = 64;
  __shared__ T Srow[kTile][36];
  const long long p = blockIdx.y;
  const int c = blockIdx.x, tid = threadIdx.x;
  const int C = prm->C, N = prm->N, M = prm->M, n = 6 * C;
  const BlWork<T> wk(C, N, M);
  const BlIdx ix(C, N, M);
  const int* iw = prm->iwork + size_t(p) * ix.total;
  if (!iw[ix.flags + 0] || !iw[ix.flags + 3]) return;
  const T* w = static_cast<const T*>(prm->work) + size_t(p) * wk.total;
  T* Sg = static_cast<T*>(prm->Sall) + size_t(p) * n * n;
  T* rg = static_cast<T*>(prm->rhsall) + size_t(p) * n;
  const T* sp = spart + (size_t(p) * C + c) * split * size_t(C) * 36;
  for (int cbase = 0; cbase <= c; cbase += kTile) {
    for (int e = tid; e < kTile * 36; e += 256) {
      const int c2 = cbase + e / 36;
      T v = T(0);
      if (c2 <= c && c2 < C)
        for (int k = 0; k < split; ++k) v += sp[(size_t(k) * C + c2) * 36 + e % 36];
      Srow[e / 36][e % 36] = v;
    }
    __syncthreads();
    for (int e = tid; e < kTile * 36; e += 256) {
      const int c2 = cbase + e / 36, q = e % 36, qa = q / 6, qd = q % 6;
      if (c2 > c || c2 >= C) continue;
      T v = Srow[c2 - cbase][q];
      if (c2 == c) {   // the diagonal block: symmetrise + U_c with its damped diagonal (as bl_schur_kernel does for split = 1)
        v = (qd <= qa) ? Srow[c2 - cbase][qa * 6 + qd] : Srow[c2 - cbase][qd * 6 + qa];
        v += (qa == qd) ? w[wk.Ud + 6 * c + qa] : w[wk.U + 36 * c + 6 * qa + qd];
      }
      Sg[size_t(6 * c + qa) * n + 6 * c2 + qd] = v;
      if (c2 != c) Sg[size_t(6 * c2 + qd) * n + 6 * c + qa] = v;
    }
    __syncthreads();
  }
  if (tid < 6) {
    T v = T(0);
    for (int k = 0; k < split; ++k) v += rpart[((size_t(p) * C + c) * split + k) * 6 + tid];
    rg[6 * c + tid] = w[wk.gc + 6 * c + tid] + v;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) bl_back_kernel(const BlParams* __restrict__ prm, const int32_t* __restrict__ solve_ok) {
  const long long p = blockIdx.y;
  const int C = prm->C, N = prm->N, M = prm->M;
  const BlWork<T> wk(C, N, M);
  const BlIdx ix(C, N, M);
  const int* iw = prm->iwork + size_t(p) * ix.total;
  if (!iw[ix.flags + 0] || !iw[ix.flags + 3] || iw[ix.flags + 4] != 0 || !solve_ok[p]) return;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  T* w = static_cast<T*>(prm->work) + size_t(p) * wk.total;
  const int* oc = prm->obs_cam + size_t(p) * M;
  T u3[3] = {w[wk.gp + 3 * j], w[wk.gp + 3 * j + 1], w[wk.gp + 3 * j + 2]};
  const T g2 = u3[0] * u3[0] + u3[1] * u3[1] + u3[2] * u3[2];
  for (int i = iw[ix.pt_start + j]; i < iw[ix.pt_start + j + 1]; ++i) {
    const T* dcc = static_cast<const T*>(prm->dcall) + size_t(p) * 6 * C + 6 * oc[i];
    // W_i^T dc = J_p^T (J_c dc)
    T e[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      T s = 0;
#pragma unroll
      for (int d = 0; d < 6; ++d) s += w[wk.Jc + size_t(i) * 16 + (6 * a + d)] * dcc[d];
      e[a] = s;
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) u3[b] += w[wk.Jp + size_t(i) * 8 + (b)] * e[0] + w[wk.Jp + size_t(i) * 8 + (3 + b)] * e[1];
  }
  const T* Rj = w + wk.Vinv + 6 * j;
  const T y0 = Rj[0] * u3[0], y1 = Rj[1] * u3[0] + Rj[2] * u3[1], y2 = Rj[3] * u3[0] + Rj[4] * u3[1] + Rj[5] * u3[2];
  const T e0 = -(Rj[0] * y0 + Rj[1] * y1 + Rj[3] * y2), e1 = -(Rj[2] * y1 + Rj[4] * y2), e2 = -(Rj[5] * y2);
  w[wk.dp + 3 * j] = e0; w[wk.dp + 3 * j + 1] = e1; w[wk.dp + 3 * j + 2] = e2;
  w[wk.pd2 + j] = e0 * e0 + e1 * e1 + e2 * e2;
  w[wk.pg2 + j] = g2;
}

template <typename T>
__global__ void __launch_bounds__(256) bl_step_kernel(const BlParams* __restrict__ prm, const int32_t* __restrict__ solve_ok, int timed_out,
                                                      int* __restrict__ any_active) {
  __shared__ T red[8];
  const long long p = blockIdx.x;
  const int C = prm->C, N = prm->N, M = prm->M, n = 6 * C;
  const BlWork<T> wk(C, N, M);
  const BlIdx ix(C, N, M);
  int* fl = prm->iwork + size_t(p) * ix.total + ix.flags;
  if (!fl[0]) {   // a finished scene: its last action (applied by the pass that finished it) must not be applied again — round 5: here instead of a launch of its own behind bl_update
    if (threadIdx.x == 0) fl[2] = 0;
    return;
  }
  T* w = static_cast<T*>(prm->work) + size_t(p) * wk.total;
  LmState<T>& S = *bl_state<T>(prm, wk, p);
  const toa_options& opt = prm->opt;
  const bool is_lm = opt.solver_type == 0;
  const int tid = threadIdx.x;
  const bool solved = fl[3] != 0 && fl[4] == 0 && solve_ok[p] != 0;
  T d2s = 0, g2s = 0;
  if (solved) {
    T d2 = 0, g2 = 0;
    for (int i = tid; i < N; i += 256) { d2 += w[wk.pd2 + i]; g2 += w[wk.pg2 + i]; }
    const T* dcg = static_cast<const T*>(prm->dcall) + size_t(p) * n;
    for (int i = tid; i < n; i += 256) { d2 += dcg[i] * dcg[i]; g2 += w[wk.gc + i] * w[wk.gc + i]; }
    d2s = ba_block_sum<T>(d2, red);
    g2s = ba_block_sum<T>(g2, red);
  }
  __syncthreads();
  if (tid == 0) {
    if (fl[3]) S.solves++;
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
      if (rc == 0) status = lm_judge_core<T>(S, opt, prm->res, p, double(d2s), opt.min_grad_norm2 > 0.0f ? double(g2s) : 0.0, true);
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
      if (timed_out && S.stop == TOA_STOP_NONE) S.stop = TOA_STOP_TIMED_OUT;   // optimizer.h:302-305 (after the step was applied)
      S.num_iters = S.num_iters + 1;
      S.iter = S.iter + 1;
      cont = (S.stop == TOA_STOP_NONE && S.iter < S.max_iters) ? 1 : 0;
    }
    fl[1] = (!is_lm || S.rebuild) ? 1 : 0;
    fl[2] = action;
    fl[0] = cont;
    if (cont) atomicAdd(any_active, 1);   // this pass's slot of the host's ring
    if (!cont) {   // optimizer.h:313-327
      const toa_results& res = prm->res;
      if (S.stop == TOA_STOP_NONE && S.num_iters >= S.max_iters) S.stop = TOA_STOP_MAX_ITERS;
      res.stop_reason[p] = S.stop;
      res.num_iters[p] = S.num_iters;
      res.final_cost[p] = S.final_cost;
      if (res.num_failures) res.num_failures[p] = int(S.num_failures);
      if (res.num_consec_failures) res.num_consec_failures[p] = int(S.num_consec);
      if (res.final_num_residuals) res.final_num_residuals[p] = S.final_nres;
      if (res.final_rerr_dec) res.final_rerr_dec[p] = S.final_rerr;
      if (res.final_inlier_ratio) res.final_inlier_ratio[p] = S.final_nres > 0 ? float(S.final_ninl) / float(S.final_nres) : 1.0f;
      if (prm->counters) {
        atomicAdd(&prm->counters[0], S.acc_passes);
        atomicAdd(&prm->counters[1], S.eval_passes);
        atomicAdd(&prm->counters[2], S.solves);
        atomicAdd(&prm->counters[3], 1ull);
      }
    }
  }
}

// x (+)= dx: SE3 on the cameras (sophus.h:24-26), Euclidean on the points (traits.h:184-190); action 2 rolls the last step back
template <typename T>
__global__ void __launch_bounds__(256) bl_update_kernel(const BlParams* __restrict__ prm) {
  const long long p = blockIdx.y;
  const int C = prm->C, N = prm->N, M = prm->M;
  const BlWork<T> wk(C, N, M);
  const BlIdx ix(C, N, M);
  const int* fl = prm->iwork + size_t(p) * ix.total + ix.flags;
  const int action = fl[2];
  if (action == 0) return;
  T* w = static_cast<T*>(prm->work) + size_t(p) * wk.total;
  T* X = static_cast<T*>(prm->x) + size_t(p) * (size_t(12) * C + size_t(3) * N);
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < C) {
    if (action == 1) {
      T d[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) { d[k] = static_cast<const T*>(prm->dcall)[size_t(p) * 6 * C + 6 * t + k]; w[wk.ldc + 6 * t + k] = d[k]; }
      ba_se3_plus<T>(X + 12 * t, d, T(1));
    } else {
      ba_se3_plus<T>(X + 12 * t, w + wk.ldc + 6 * t, T(-1));
    }
  }
  for (int i = t; i < 3 * N; i += gridDim.x * 256) {
    if (action == 1) { const T d = w[wk.dp + i]; X[size_t(12) * C + i] += d; w[wk.ldp + i] = d; }
    else X[size_t(12) * C + i] -= w[wk.ldp + i];
  }
}

// a scene that is still running when the host's pass budget is exhausted: finalised as kMaxIters (optimizer.h:320-321)
template <typename T>
__global__ void __launch_bounds__(64) bl_force_stop_kernel(const BlParams* __restrict__ prm) {
  const long long p = blockIdx.x;
  const BlWork<T> wk(prm->C, prm->N, prm->M);
  const BlIdx ix(prm->C, prm->N, prm->M);
  int* fl = prm->iwork + size_t(p) * ix.total + ix.flags;
  if (threadIdx.x != 0 || !fl[0]) return;
  LmState<T>& S = *bl_state<T>(prm, wk, p);
  const toa_results& res = prm->res;
  fl[0] = 0;
  if (S.stop == TOA_STOP_NONE) S.stop = TOA_STOP_MAX_ITERS;
  res.stop_reason[p] = S.stop;
  res.num_iters[p] = S.num_iters;
  res.final_cost[p] = S.final_cost;
  if (res.num_failures) res.num_failures[p] = int(S.num_failures);
  if (res.num_consec_failures) res.num_consec_failures[p] = int(S.num_consec);
  if (res.final_num_residuals) res.final_num_residuals[p] = S.final_nres;
  if (res.final_rerr_dec) res.final_rerr_dec[p] = S.final_rerr;
  if (res.final_inlier_ratio) res.final_inlier_ratio[p] = S.final_nres > 0 ? float(S.final_ninl) / float(S.final_nres) : 1.0f;
  if (prm->counters) {
    atomicAdd(&prm->counters[0], S.acc_passes);
    atomicAdd(&prm->counters[1], S.eval_passes);
    atomicAdd(&prm->counters[2], S.solves);
    atomicAdd(&prm->counters[3], 1ull);
  }
}

template <typename T>
int ba_lists_run_t(toa_handle h, int dtype, BlParams prm, double max_duration_ms) {
  // Under stream capture (hipGraph; round 5) the host can look at no stop flag: the whole pass budget of the options is recorded — every
  // kernel of a pass returns at once for the scenes that have finished, the solver takes its per-scene mask — and the scenes still
  // running at the end of it are finalised as kMaxIters, as the eager loop does.  Like the n > 128 pipeline's captured form this needs
  // a bounded budget (max_consec_failures > 0), every stage a kernel of this library (use_ldlt, the one-workgroup factorisations) and
  // workspaces that exist (a first eager call of the shape makes them); max_duration_ms needs the host's clock and is refused.
  bool capturing = false;
  {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    capturing = hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  }
  const int C = prm.C, N = prm.N, M = prm.M, n = 6 * C;
  const long long P = prm.P;
  const BlWork<T> wk(C, N, M);
  const BlIdx ix(C, N, M);
  // every iteration is at most max_consec retries + 1 passes; bounded like the n > 128 pipeline's host loop
  long long max_passes = (long long)(prm.opt.max_iters + 3) * 260;
  if (capturing) {   // (refusals come BEFORE the first stream operation: a capture that handles the error is not left half-recorded)
    constexpr long long kMaxCapturedPasses = 256;
    const long long tries = prm.opt.max_consec_failures > 0 ? (long long)prm.opt.max_consec_failures + 1 : 256;
    max_passes = (long long)(prm.opt.max_iters + 2) * tries;
    const size_t chol2_lds = (size_t(32) * 36 + size_t(n) * 37 + 96) * sizeof(T) + 64;
    const bool own_solver = n <= 128 || (P <= 65535 && chol2_lds + 2048 <= size_t(h->max_lds));
    if (max_duration_ms > 0 || !prm.opt.use_ldlt || h->tune.large_library_solver != 0 || !own_solver)
      return toa_fail(TOA_E_UNSUPPORTED, "toa_ba_lists_run under stream capture: only solves whose every stage is a kernel of this library can be captured "
                                         "(use_ldlt, the one-workgroup factorisation of the reduced camera system, no max_duration_ms)");
    if (max_passes > kMaxCapturedPasses)
      return toa_fail(TOA_E_UNSUPPORTED, "toa_ba_lists_run under stream capture: the pass budget of these options is " + std::to_string(max_passes) +
                                         " passes (~" + std::to_string(max_passes * 12) + " graph nodes); at most " + std::to_string(kMaxCapturedPasses) +
                                         " are recorded — set max_consec_failures > 0 (the retry bound per iteration) or lower max_iters");
  }
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t b_work = al(size_t(P) * wk.total * sizeof(T)), b_iwork = al(size_t(P) * ix.total * sizeof(int)), b_ok = al(size_t(P) * sizeof(int32_t));
  const size_t b_S = al(size_t(P) * n * n * sizeof(T)), b_v = al(size_t(P) * n * sizeof(T));
  // bl_schur split (see the kernel): a property of the scene's shape only
  const int split = (C <= 128 && M / std::max(C, 1) >= 256) ? 3 : 1;   // (round 5, four scenes x 64 cameras, it/s: 2: 6759, 3: 7001, 4: 6730, 5: 6701, 6: 6865, 8: 6764; round 4, 8 against 4: 5.02 / 5.10 ms at four scenes, 10.43 / 10.28 at 32)
  const size_t b_sp = split > 1 ? al(size_t(P) * C * split * C * 36 * sizeof(T)) : 0, b_rp = split > 1 ? al(size_t(P) * C * split * 6 * sizeof(T)) : 0;
  const size_t need = b_work + b_iwork + b_ok + 256 + b_S + 2 * b_v + b_sp + b_rp;
  if (need > h->aux_bytes) {   // (h->scratch belongs to toa_large_solve, which this pipeline calls)
    if (int rc = grow_sync(h, "bundle adjustment workspace")) return rc;   // (refused under capture with a message that says why)
    toa_release_workspace(h, h->aux);
    h->aux = nullptr;
    h->aux_bytes = 0;
    HIP_TRY(hipMalloc(&h->aux, need));
    h->aux_bytes = need;
  }
  char* base = static_cast<char*>(h->aux);
  prm.work = base;
  prm.iwork = reinterpret_cast<int*>(base + b_work);
  int32_t* ok = reinterpret_cast<int32_t*>(base + b_work + b_iwork);
  prm.any_active = reinterpret_cast<int*>(base + b_work + b_iwork + b_ok);
  prm.Sall = base + b_work + b_iwork + b_ok + 256;
  prm.rhsall = static_cast<char*>(prm.Sall) + b_S;
  prm.dcall = static_cast<char*>(prm.rhsall) + b_v;
  T* spart = reinterpret_cast<T*>(static_cast<char*>(prm.dcall) + b_v);
  T* rpart = reinterpret_cast<T*>(static_cast<char*>(prm.dcall) + b_v + b_sp);
  HIP_TRY(hipMemsetAsync(prm.dcall, 0, b_v, h->stream));
  static_assert(sizeof(BlParams) <= 1024, "parameter block too large");
  if (int rc = upload_params(h, &prm, sizeof(prm))) return rc;
  const BlParams* dev = static_cast<const BlParams*>(h->params_dev);
  hipStream_t st = h->stream;
  const unsigned gM = unsigned((M + 255) / 256), gN = unsigned((N + 255) / 256), gX = unsigned((std::max(N * 3 / 8, C) + 255) / 256);
  HIP_TRY(hipMemsetAsync(prm.iwork, 0, b_iwork, st));   // flags, camera counts
  hipLaunchKernelGGL(bl_index_a_kernel, dim3(gM, unsigned(P)), dim3(256), 0, st, dev);
  hipLaunchKernelGGL(bl_index_b_kernel, dim3(unsigned(P)), dim3(64), 0, st, dev);
  hipLaunchKernelGGL(bl_index_c_kernel, dim3(unsigned(C), unsigned(P)), dim3(64), 0, st, dev);
  hipLaunchKernelGGL(bl_init_kernel<T>, dim3(unsigned(P)), dim3(64), 0, st, dev);
  HIP_TRY(hipGetLastError());
  if (capturing) h->shadow_retired = true;   // (a graph of this handle now exists: its workspaces are never freed under it, toa_release_workspace)
  // Round 4: the host no longer waits for a pass before it enqueues the next one.  Every pass leaves "is any scene still
  // running" in its own slot of a small ring; the slot is copied to pinned host memory behind the pass and the host looks at
  // pass k's answer only before it enqueues pass k + kAhead — the GPU always has the next pass queued (round 3: a
  // hipStreamSynchronize + 4-byte read-back per pass, ~40 us of idle GPU each).  The at most kAhead surplus passes that are
  // enqueued after the last scene has finished find fl[0] == 0 everywhere and return at once (every kernel of the pipeline
  // checks it first).  max_duration_ms needs the elapsed device time BEFORE each pass is enqueued (optimizer.h:302-305):
  // that form keeps the pass-by-pass hand-shake.
  constexpr int kAhead = 2, kRing = toa_context::kPassRing;
  struct Events {   // (RAII: every early return below used to leak the timing events — ADVICE r03)
    hipEvent_t t0 = nullptr, t1 = nullptr;
    hipEvent_t* done = nullptr;     // the handle's ring (kept across calls)
    int* host_flags = nullptr;
    ~Events() {
      if (t0) (void)hipEventDestroy(t0);
      if (t1) (void)hipEventDestroy(t1);
    }
  } ev;
  const bool timed = max_duration_ms > 0;
  if (timed) {
    HIP_TRY(hipEventCreate(&ev.t0));
    HIP_TRY(hipEventCreate(&ev.t1));
    HIP_TRY(hipEventRecord(ev.t0, st));
  }
  if (!capturing) {
    if (int rc = ensure_pass_ring(h)) return rc;
    ev.done = h->pass_done;
    ev.host_flags = h->pass_flags;
  }
  int* any_ring = prm.any_active;   // kRing ints (the block reserves 256 bytes)
  int rc_all = TOA_OK;
  bool finished = false;
  long long pass = 0;
  for (; pass < max_passes; ++pass) {
    const int slot = int(pass % kRing);
    if (!capturing && (pass >= kAhead || timed)) {   // the answer of pass - kAhead (timed form: of the previous pass)
      const long long look = timed ? pass - 1 : pass - kAhead;
      if (look >= 0) {
        HIP_TRY(hipEventSynchronize(ev.done[look % kRing]));
        if (ev.host_flags[look % kRing] == 0) { finished = true; break; }
      }
    }
    int timed_out = 0;
    if (timed) {   // the reference adds up the iterations' wall time (optimizer.h:302-305): device time here
      HIP_TRY(hipEventRecord(ev.t1, st));
      HIP_TRY(hipEventSynchronize(ev.t1));
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, ev.t0, ev.t1));
      timed_out = double(ms) > max_duration_ms ? 1 : 0;
    }
    int* const any_slot = any_ring + slot;
    HIP_TRY(hipMemsetAsync(any_slot, 0, sizeof(int), st));
    hipLaunchKernelGGL(bl_obs_kernel<T>, dim3(gM, unsigned(P)), dim3(256), 0, st, dev);
    hipLaunchKernelGGL(bl_point_cam_kernel<T>, dim3(gN + unsigned(C), unsigned(P)), dim3(256), 0, st, dev, int(gN));
    hipLaunchKernelGGL(bl_build_kernel<T>, dim3(unsigned(P)), dim3(256), 0, st, dev);
    hipLaunchKernelGGL(bl_psolve_kernel<T>, dim3(gN, unsigned(P)), dim3(256), 0, st, dev);
    hipLaunchKernelGGL(bl_schur_kernel<T>, dim3(unsigned(C * split), unsigned(P)), dim3(256), 0, st, dev, split, spart, rpart);
    if (split > 1) hipLaunchKernelGGL(bl_schur_reduce_kernel<T>, dim3(unsigned(C), unsigned(P)), dim3(256), 0, st, dev, split, (const T*)spart, (const T*)rpart);
    HIP_TRY(hipGetLastError());
    // reduced camera system: S dc = -red  (toa_large_solve: dx = -H^-1 g; scale 1: S is already damped).  A scene that is not
    // running (or whose Build failed) still goes through the solver on whatever its S holds: its verdict is ignored.
    // (the solver's own kernels skip the scenes that have stopped: flags [0]; with passes enqueued ahead of the stop flag the
    // surplus passes then cost launches, not factorisations)
    struct MaskScope {
      toa_handle h;
      MaskScope(toa_handle hh, const int32_t* m, int64_t s) : h(hh) { h->solve_mask = m; h->solve_mask_stride = s; }
      ~MaskScope() { h->solve_mask = nullptr; h->solve_mask_stride = 0; }
    } mask_scope(h, prm.iwork + ix.flags, int64_t(ix.total));
    if (!prm.opt.use_ldlt) {   // gn.h:157-162: "-H.inverse() * g without any checks": the library's general LU, verdict ignored
      if (int rc = toa_large_solve_unchecked(h, dtype, n, P, prm.Sall, prm.rhsall, 1.0, prm.dcall, ok)) { rc_all = rc; break; }
    } else if (n <= 128) {   // the workgroup LDL^T: one workgroup per matrix, the same arithmetic whatever the batch
      if (int rc = toa_large_solve(h, dtype, n, P, prm.Sall, prm.rhsall, 1.0, prm.dcall, ok)) { rc_all = rc; break; }
    } else {
      // rocSOLVER: ONE matrix per call — its batched Cholesky picks its blocking by batch size, and a scene solved alone must
      // give the bits of its row in a batch (tests/test_gpu_ba_lists.py); scenes of this size are few per call
      // (toa_large_solve_each: the calls go out over side streams, so the scenes' factorisations overlap)
      // (round 5: the one-workgroup Cholesky runs IN PLACE on S — every pass rebuilds S — and writes step and verdict itself: no mask /
      //  copy / finish launches around it; toa_large_solve_each where it does not apply)
      rc_all = toa_large_solve_inplace(h, dtype, n, P, prm.Sall, prm.rhsall, prm.dcall, ok);
      if (rc_all != TOA_OK) break;
    }
    hipLaunchKernelGGL(bl_back_kernel<T>, dim3(gN, unsigned(P)), dim3(256), 0, st, dev, (const int32_t*)ok);
    hipLaunchKernelGGL(bl_step_kernel<T>, dim3(unsigned(P)), dim3(256), 0, st, dev, (const int32_t*)ok, timed_out, any_slot);
    hipLaunchKernelGGL(bl_update_kernel<T>, dim3(gX, unsigned(P)), dim3(256), 0, st, dev);
    HIP_TRY(hipGetLastError());
    if (!capturing) {
      HIP_TRY(hipMemcpyAsync(&ev.host_flags[slot], any_slot, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipEventRecord(ev.done[slot], st));
    }
  }
  if (rc_all != TOA_OK) return rc_all;
  if (capturing) {   // whoever still runs when the recorded budget ends: kMaxIters (the kernel looks at the scenes' own flags)
    hipLaunchKernelGGL(bl_force_stop_kernel<T>, dim3(unsigned(P)), dim3(64), 0, st, dev);
    HIP_TRY(hipGetLastError());
    return TOA_OK;
  }
  if (!finished) {
    // the passes enqueued last have not been looked at yet; and a loop that ran out of passes must not hand back scenes that
    // are still running with whatever the caller's result arrays held: they end with kMaxIters, like
    // optimizer.h:320-321 ends a loop whose iteration count is exhausted
    HIP_TRY(hipStreamSynchronize(st));
    bool any = false;
    for (long long k = std::max(0ll, pass - kAhead); k < pass; ++k) any = ev.host_flags[k % kRing] != 0;   // (the LAST pass decides)
    if (any) {
      hipLaunchKernelGGL(bl_force_stop_kernel<T>, dim3(unsigned(P)), dim3(64), 0, st, dev);
      HIP_TRY(hipGetLastError());
    }
  }
  return TOA_OK;
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
  if ((results->errs || results->deltas2 || results->successes) && results->hist_stride < options->max_iters + 2)
    return toa_fail(TOA_E_ARG, "toa_ba_run: hist_stride must be >= max_iters + 2");
  if (options->max_iters < 0 || options->max_iters > 65535) return toa_fail(TOA_E_ARG, "max_iters out of range");
  if (P == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  BaParams prm;
  std::memset(&prm, 0, sizeof(prm));
  prm.data = data_dev; prm.x = x_dev; prm.P = P; prm.C = num_cameras; prm.N = num_points;
  prm.opt = *options; prm.res = *results;
  prm.res.final_hessian = nullptr;   // the block Hessian is not exported
  prm.counters = reinterpret_cast<unsigned long long*>(counters_dev);
  prm.loss = h->loss;
  prm.loss_th2 = h->loss_th2;
  if (h->loss != TOA_LOSS_L2) return toa_ba_launch_robust(h, dtype, prm);   // the handle's M-estimator (toa_set_loss)
  return dtype == TOA_F32 ? launch_ba_any<float, false>(h, prm) : launch_ba_any<double, false>(h, prm);
}

extern "C" int toa_ba_lists_run(toa_handle h, int dtype, int num_cameras, int num_points, int num_obs, int64_t P, const void* intr_dev,
                                const int32_t* obs_cam_dev, const int32_t* obs_pt_dev, const void* obs_uv_dev, void* x_dev,
                                const toa_options* options, const toa_results* results, uint64_t* counters_dev, double max_duration_ms) {
  if (!h) return toa_fail(TOA_E_ARG, "null handle");
  if (dtype != TOA_F32 && dtype != TOA_F64) return toa_fail(TOA_E_ARG, "dtype must be TOA_F32 or TOA_F64");
  if (num_cameras < 1 || num_cameras > 682) return toa_fail(TOA_E_ARG, "toa_ba_lists_run: 1 <= num_cameras <= 682 (reduced camera system of at most 4092 unknowns)");
  if (num_points < 1 || num_points > (1 << 22) || num_obs < 1 || num_obs > (1 << 26)) return toa_fail(TOA_E_ARG, "toa_ba_lists_run: num_points / num_obs out of range");
  if (P < 0 || P > 65535) return toa_fail(TOA_E_ARG, "toa_ba_lists_run: P must be in [0, 65535]");
  if (!intr_dev || !obs_cam_dev || !obs_pt_dev || !obs_uv_dev || !x_dev || !options || !results) return toa_fail(TOA_E_ARG, "toa_ba_lists_run: null pointer");
  if (!results->stop_reason || !results->num_iters || !results->final_cost)
    return toa_fail(TOA_E_ARG, "toa_ba_lists_run: stop_reason, num_iters and final_cost outputs are required");
  if (options->solver_type != 0 && options->solver_type != 1) return toa_fail(TOA_E_ARG, "toa_ba_lists_run: solver_type must be 0 (LM) or 1 (GN)");
  if ((results->errs || results->deltas2 || results->successes) && results->hist_stride < options->max_iters + 2)
    return toa_fail(TOA_E_ARG, "toa_ba_lists_run: hist_stride must be >= max_iters + 2");
  if (options->max_iters < 0 || options->max_iters > 65535) return toa_fail(TOA_E_ARG, "max_iters out of range");
  if (P == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  BlParams prm;
  std::memset(&prm, 0, sizeof(prm));
  prm.intr = intr_dev; prm.obs_cam = obs_cam_dev; prm.obs_pt = obs_pt_dev; prm.obs_uv = obs_uv_dev; prm.x = x_dev;
  prm.P = P; prm.C = num_cameras; prm.N = num_points; prm.M = num_obs;
  prm.opt = *options; prm.res = *results;
  prm.res.final_hessian = nullptr;
  prm.counters = reinterpret_cast<unsigned long long*>(counters_dev);
  prm.loss = h->loss;
  prm.loss_th2 = h->loss_th2;
  return dtype == TOA_F32 ? ba_lists_run_t<float>(h, dtype, prm, max_duration_ms) : ba_lists_run_t<double>(h, dtype, prm, max_duration_ms);
}
#else
}  // namespace toa
#endif  // TOA_BA_ROBUST_TU
