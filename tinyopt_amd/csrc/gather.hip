// C1 — the single collective of the path (SURVEY §8(b) export list `gather(handle_group...)`, §8(e)): problems shard
// embarrassingly across the GPUs of a node (one process per GPU, rank g owns the contiguous id block
// shard_range(P_total, g, G)); nothing is exchanged during the solve, and ONE RCCL gather over xGMI brings
// (x, stop_reason, num_iters, final_cost) of every shard to the root, in problem-id order and in their native types
// (C4: 50 floats + 2 int32 + 1 double = 216 B per problem, 2.7 MB per GPU).
//
//   pack kernel      the rank's P_local records -> one byte buffer [P_max][rec]   (P_max = largest shard: equal counts)
//   ncclGather       ONE collective (RCCL's gather extension, rccl.h:745), root receives [G][P_max][rec]
//   unpack kernel    root only: records -> x_all [P_total][xd], stop_reason / num_iters / final_cost [P_total]
//
// RCCL is opened with dlopen on first use, like rocBLAS / rocSOLVER in large_n.hip: it is not a load-time dependency of
// the single-GPU product path; inside a torch process the name resolves to the copy torch has already loaded.
#include <dlfcn.h>
// RCCL is a RUN-time option of this library (dlopen below), so its development headers must not be a BUILD-time requirement
// of the single-GPU product either: without them the handful of types / enumerators used here are declared locally, with
// the values of nccl.h 2.x (rccl.h carries the same ABI).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
}
#endif

#include <mutex>

#include "kernels.hpp"

struct toa_comm_s {
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0, device = 0;
  void* send = nullptr;   // [P_max][rec] bytes
  void* recv = nullptr;   // root: [nranks][P_max][rec] bytes
  size_t send_bytes = 0, recv_bytes = 0;
};

namespace {

struct RcclApi {
  ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
  ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
  ncclResult_t (*gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*error_string)(ncclResult_t) = nullptr;
  std::string err;
  bool ok = false;
};

RcclApi& rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) {
      const char* e = dlerror();
      api.err = std::string("cannot open RCCL (librccl.so): ") + (e ? e : "?");
      return;
    }
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(lib, name);
      if (!p && api.err.empty()) api.err = std::string("RCCL: missing symbol ") + name;
      return p;
    };
    api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(sym("ncclGetUniqueId"));
    api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(sym("ncclCommInitRank"));
    api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(sym("ncclCommDestroy"));
    api.gather = reinterpret_cast<decltype(api.gather)>(sym("ncclGather"));
    api.error_string = reinterpret_cast<decltype(api.error_string)>(sym("ncclGetErrorString"));
    api.ok = api.err.empty();
  });
  return api;
}

#define RCCL_TRY(api, expr)                                                                                   \
  do {                                                                                                        \
    const ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) return toa_fail(TOA_E_RCCL, std::string(#expr) + ": " + (api).error_string(r_));   \
  } while (0)

// contiguous block partition of problem ids (tinyopt_amd/dist.py shard_range): sizes differ by at most one
__host__ __device__ inline long long shard_lo(long long P, int r, int G) {
  const long long base = P / G, rem = P % G;
  return r * base + (r < rem ? r : rem);
}

// record layout: [ x : xd * sizeof(T) | stop_reason i32 | num_iters i32 | final_cost f64 ], 8-byte aligned
__host__ __device__ inline size_t rec_bytes(int xd, int tsize) { return ((size_t(xd) * tsize + 7) & ~size_t(7)) + 16; }

// One thread per record ELEMENT (xd values of x, then one slot for the three trailer fields): consecutive threads read
// consecutive elements of x and write consecutive bytes of a record, so both sides coalesce for any xd (a thread per record
// walked its record with stride rec: 64 separate cache lines per wave access).
template <typename T>
__global__ void gather_pack_kernel(const T* __restrict__ x, const int* __restrict__ stop, const int* __restrict__ iters,
                                   const double* __restrict__ cost, long long P_local, int xd, char* __restrict__ out) {
  const size_t rec = rec_bytes(xd, sizeof(T));
  const size_t xb = (size_t(xd) * sizeof(T) + 7) & ~size_t(7);
  const long long per = (long long)xd + 1;
  const long long total = P_local * per;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long p = e / per;
    const int j = int(e - p * per);
    char* r = out + size_t(p) * rec;
    if (j < xd) {
      reinterpret_cast<T*>(r)[j] = x[size_t(p) * xd + j];
    } else {
      reinterpret_cast<int*>(r + xb)[0] = stop[p];
      reinterpret_cast<int*>(r + xb)[1] = iters[p];
      *reinterpret_cast<double*>(r + xb + 8) = cost[p];
    }
  }
}

template <typename T>
__global__ void gather_unpack_kernel(const char* __restrict__ in, long long P_total, long long P_max, int G, int xd,
                                     T* __restrict__ x, int* __restrict__ stop, int* __restrict__ iters, double* __restrict__ cost) {
  const size_t rec = rec_bytes(xd, sizeof(T));
  const size_t xb = (size_t(xd) * sizeof(T) + 7) & ~size_t(7);
  const long long per = (long long)xd + 1;
  const long long total = P_total * per;
  const long long base = P_total / G, rem = P_total % G;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long q = e / per;
    const int j = int(e - q * per);
    // owner of problem id q under the block partition
    const int r = (q < rem * (base + 1)) ? int(q / (base + 1)) : int(rem + (q - rem * (base + 1)) / (base > 0 ? base : 1));
    const long long local = q - shard_lo(P_total, r, G);
    const char* src = in + (size_t(r) * P_max + size_t(local)) * rec;
    if (j < xd) {
      if (x) x[size_t(q) * xd + j] = reinterpret_cast<const T*>(src)[j];
    } else {
      if (stop) stop[q] = reinterpret_cast<const int*>(src + xb)[0];
      if (iters) iters[q] = reinterpret_cast<const int*>(src + xb)[1];
      if (cost) cost[q] = *reinterpret_cast<const double*>(src + xb + 8);
    }
  }
}

int ensure(void** buf, size_t* have, size_t need) {
  if (need <= *have) return TOA_OK;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr;
  *have = 0;
  HIP_TRY(hipMalloc(buf, need));
  *have = need;
  return TOA_OK;
}

}  // namespace

extern "C" {

int toa_comm_unique_id(void* id_bytes) {
  if (!id_bytes) return toa_fail(TOA_E_ARG, "toa_comm_unique_id: null output");
  RcclApi& api = rccl_api();
  if (!api.ok) return toa_fail(TOA_E_UNSUPPORTED, api.err);
  static_assert(sizeof(ncclUniqueId) == TOA_COMM_ID_BYTES, "TOA_COMM_ID_BYTES must match ncclUniqueId");
  ncclUniqueId id;
  RCCL_TRY(api, api.get_unique_id(&id));
  std::memcpy(id_bytes, &id, sizeof(id));
  return TOA_OK;
}

int toa_comm_init_rank(toa_handle h, const void* id_bytes, int nranks, int rank, toa_comm* out) {
  if (!h || !id_bytes || !out) return toa_fail(TOA_E_ARG, "toa_comm_init_rank: null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return toa_fail(TOA_E_ARG, "toa_comm_init_rank: rank out of range");
  RcclApi& api = rccl_api();
  if (!api.ok) return toa_fail(TOA_E_UNSUPPORTED, api.err);
  TOA_ON_DEVICE(h->device);
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof(id));
  toa_comm_s* c = new (std::nothrow) toa_comm_s();
  if (!c) return toa_fail(TOA_E_NOMEM, "toa_comm_init_rank: host allocation failed");
  const ncclResult_t r = api.comm_init_rank(&c->comm, nranks, id, rank);   // collective over the ranks of the id
  if (r != ncclSuccess) {
    delete c;
    return toa_fail(TOA_E_RCCL, std::string("ncclCommInitRank: ") + api.error_string(r));
  }
  c->nranks = nranks;
  c->rank = rank;
  c->device = h->device;
  *out = c;
  return TOA_OK;
}

int toa_comm_destroy(toa_comm c) {
  if (!c) return TOA_OK;
  toa::DeviceGuard guard_(c->device);
  RcclApi& api = rccl_api();
  if (c->send) (void)hipFree(c->send);
  if (c->recv) (void)hipFree(c->recv);
  if (api.ok && c->comm) (void)api.comm_destroy(c->comm);
  delete c;
  return TOA_OK;
}

int toa_shard_range(int64_t P_total, int rank, int nranks, int64_t* lo, int64_t* hi) {
  if (P_total < 0 || nranks < 1 || rank < 0 || rank >= nranks || !lo || !hi) return toa_fail(TOA_E_ARG, "toa_shard_range: bad argument");
  *lo = shard_lo(P_total, rank, nranks);
  *hi = shard_lo(P_total, rank + 1, nranks);
  return TOA_OK;
}

int toa_gather(toa_handle h, toa_comm c, int dtype, int xdim, int64_t P_total, const void* x_dev, const toa_results* local,
               int root, void* x_all_dev, const toa_results* all) {
  if (!h || !c) return toa_fail(TOA_E_ARG, "toa_gather: null handle / communicator");
  if (dtype != TOA_F32 && dtype != TOA_F64) return toa_fail(TOA_E_ARG, "dtype must be TOA_F32 or TOA_F64");
  // any parameter-block width (DenseRowNatural: n up to 1024; bundle adjustment: 12 C + 3 N); only the byte counts are bounded
  if (xdim < 1 || xdim > (1 << 24) || P_total < 0 || P_total > 0x7fffffff) return toa_fail(TOA_E_ARG, "toa_gather: bad shape");
  if (root < 0 || root >= c->nranks) return toa_fail(TOA_E_ARG, "toa_gather: root out of range");
  if (c->device != h->device) return toa_fail(TOA_E_ARG, "toa_gather: communicator belongs to another device");
  const bool is_root = c->rank == root;
  const long long lo = shard_lo(P_total, c->rank, c->nranks), hi = shard_lo(P_total, c->rank + 1, c->nranks);
  const long long P_local = hi - lo;
  if (P_local > 0 && (!x_dev || !local || !local->stop_reason || !local->num_iters || !local->final_cost))
    return toa_fail(TOA_E_ARG, "toa_gather: x / stop_reason / num_iters / final_cost of the local shard are required");
  if (is_root && !all) return toa_fail(TOA_E_ARG, "toa_gather: the root needs the destination arrays");
  RcclApi& api = rccl_api();
  if (!api.ok) return toa_fail(TOA_E_UNSUPPORTED, api.err);
  if (P_total == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  const int tsize = dtype == TOA_F32 ? 4 : 8;
  const size_t rec = rec_bytes(xdim, tsize);
  const long long P_max = (P_total + c->nranks - 1) / c->nranks;
  const size_t send_bytes = size_t(P_max) * rec;
  if (send_bytes / rec != size_t(P_max) || send_bytes > (size_t(1) << 40) / size_t(c->nranks))
    return toa_fail(TOA_E_ARG, "toa_gather: P_total * record size exceeds 1 TiB");
  if (int rc = ensure(&c->send, &c->send_bytes, send_bytes)) return rc;
  if (is_root)
    if (int rc = ensure(&c->recv, &c->recv_bytes, send_bytes * size_t(c->nranks))) return rc;
  const int grid = h->num_cus * 4;
  if (P_local > 0) {
    if (dtype == TOA_F32)
      hipLaunchKernelGGL(gather_pack_kernel<float>, dim3(grid), dim3(256), 0, h->stream, (const float*)x_dev, local->stop_reason,
                         local->num_iters, local->final_cost, P_local, xdim, (char*)c->send);
    else
      hipLaunchKernelGGL(gather_pack_kernel<double>, dim3(grid), dim3(256), 0, h->stream, (const double*)x_dev, local->stop_reason,
                         local->num_iters, local->final_cost, P_local, xdim, (char*)c->send);
    HIP_TRY(hipGetLastError());
  }
  RCCL_TRY(api, api.gather(c->send, is_root ? c->recv : nullptr, send_bytes, ncclUint8, root, c->comm, h->stream));
  if (is_root) {
    if (dtype == TOA_F32)
      hipLaunchKernelGGL(gather_unpack_kernel<float>, dim3(grid), dim3(256), 0, h->stream, (const char*)c->recv, (long long)P_total, P_max,
                         c->nranks, xdim, (float*)x_all_dev, all->stop_reason, all->num_iters, all->final_cost);
    else
      hipLaunchKernelGGL(gather_unpack_kernel<double>, dim3(grid), dim3(256), 0, h->stream, (const char*)c->recv, (long long)P_total, P_max,
                         c->nranks, xdim, (double*)x_all_dev, all->stop_reason, all->num_iters, all->final_cost);
    HIP_TRY(hipGetLastError());
  }
  return TOA_OK;
}

}  // extern "C"
