// Run-time user functors: `Optimize(x, cost)` with a residual the LIBRARY has never seen, without rebuilding it.
//
// The reference's whole API is "pass any callable": tinyopt::Optimize(x, [](const auto& x) { return r(x); })
// (include/tinyopt/optimize.h:16-33, optimizers/optimizer.h:145-160, docs/API.md:21-35) — the residual is a C++ template,
// differentiated by ceres::Jet at COMPILE time of the user's program.  A device path cannot take a host callable, and until
// round 3 a new residual meant editing csrc/inst.hip and rebuilding libtinyopt_amd.so.  This file is the MI355X-native
// equivalent of "any callable": the user hands over the BODY of the residual as C++ source text, written exactly like the
// lambda they would hand to tinyopt (generic in the scalar type S: it is instantiated on toa::Jet<T, N> for Accumulate and on
// plain T for the cost-only form, optimize_autodiff.h:91-166), and
//
//     hiprtc  ->  a code object holding lm_fused_kernel<JetModel<T, UserFunctor>> and accumulate_kernel<...>  ->  hipModule
//
// i.e. the same kernels, state machine, LDL^T and forward-mode AD as every built-in family (csrc/kernels.hpp is compiled
// as is: it guards its host half with __HIPCC_RTC__), specialised for the user's residual at run time (~2-3 s, once).
// hiprtc is opened with dlopen on first use, like rocBLAS / rocSOLVER / RCCL: not a load-time dependency of the product.
#include <dlfcn.h>
#include <glob.h>

#include <mutex>
#include <string>
#include <vector>

#include "kernels.hpp"

struct toa_jit_model_s {
  hipModule_t module = nullptr;
  hipFunction_t fused = nullptr, accumulate = nullptr;
  int dtype = 0, kN = 0, kR = 0, kD = 0, kH = 0, device = 0;
  int wg_per_cu = 0;
  size_t lds_wg = 0;
};

namespace {

// the few hiprtc entry points used, with hiprtc.h's own signatures (hiprtcResult is an int-sized enum, 0 = success)
struct RtcApi {
  int (*create)(void**, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
  int (*add_name)(void*, const char*) = nullptr;
  int (*compile)(void*, int, const char* const*) = nullptr;
  int (*log_size)(void*, size_t*) = nullptr;
  int (*get_log)(void*, char*) = nullptr;
  int (*code_size)(void*, size_t*) = nullptr;
  int (*get_code)(void*, char*) = nullptr;
  int (*lowered)(void*, const char*, const char**) = nullptr;
  int (*destroy)(void**) = nullptr;
  std::string err;
  bool ok = false;
};

RtcApi& rtc_api() {
  static RtcApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* lib = nullptr;
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) {
      const char* e = dlerror();
      api.err = std::string("cannot open hiprtc (libhiprtc.so): ") + (e ? e : "?");
      return;
    }
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(lib, name);
      if (!p && api.err.empty()) api.err = std::string("hiprtc: missing symbol ") + name;
      return p;
    };
    api.create = reinterpret_cast<decltype(api.create)>(sym("hiprtcCreateProgram"));
    api.add_name = reinterpret_cast<decltype(api.add_name)>(sym("hiprtcAddNameExpression"));
    api.compile = reinterpret_cast<decltype(api.compile)>(sym("hiprtcCompileProgram"));
    api.log_size = reinterpret_cast<decltype(api.log_size)>(sym("hiprtcGetProgramLogSize"));
    api.get_log = reinterpret_cast<decltype(api.get_log)>(sym("hiprtcGetProgramLog"));
    api.code_size = reinterpret_cast<decltype(api.code_size)>(sym("hiprtcGetCodeSize"));
    api.get_code = reinterpret_cast<decltype(api.get_code)>(sym("hiprtcGetCode"));
    api.lowered = reinterpret_cast<decltype(api.lowered)>(sym("hiprtcGetLoweredName"));
    api.destroy = reinterpret_cast<decltype(api.destroy)>(sym("hiprtcDestroyProgram"));
    api.ok = api.err.empty();
  });
  return api;
}

// Where the kernel sources live: next to this shared library (tinyopt_amd/libtinyopt_amd.so -> tinyopt_amd/csrc), or
// TOA_JIT_CSRC.  They ship with the package: the run-time model is compiled from the very headers the library was built from.
std::string csrc_dir() {
  if (const char* e = std::getenv("TOA_JIT_CSRC")) return e;
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&rtc_api), &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t s = p.find_last_of('/');
    return (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/csrc";
  }
  return "tinyopt_amd/csrc";
}

// hiprtc finds libstdc++ on its own but not clang's builtin headers (stddef.h, ...): the resource directory of the ROCm LLVM
std::vector<std::string> extra_includes() {
  std::vector<std::string> v;
  if (const char* e = std::getenv("TOA_JIT_INCLUDE")) v.push_back(e);
  for (const char* pat : {"/opt/rocm/lib/llvm/lib/clang/*/include", "/opt/rocm-*/lib/llvm/lib/clang/*/include"}) {
    glob_t g;
    if (glob(pat, 0, nullptr, &g) == 0) {
      for (size_t i = 0; i < g.gl_pathc; ++i) v.push_back(g.gl_pathv[i]);
      globfree(&g);
    }
    if (!v.empty()) break;
  }
  v.push_back("/opt/rocm/include");
  return v;
}

}  // namespace

extern "C" {

int toa_model_compile(toa_handle h, int dtype, int num_params, int residuals_per_item, int scalars_per_item, int header_scalars,
                      const char* residual_body, toa_jit_model* out, char* log_out, size_t log_cap) {
  if (log_out && log_cap) log_out[0] = 0;
  if (!h || !residual_body || !out) return toa_fail(TOA_E_ARG, "toa_model_compile: null argument");
  if (dtype != TOA_F32 && dtype != TOA_F64) return toa_fail(TOA_E_ARG, "dtype must be TOA_F32 or TOA_F64");
  if (num_params < 1 || num_params > 12)
    return toa_fail(TOA_E_UNSUPPORTED, "toa_model_compile: 1 <= num_params <= 12 (the register Gram of JetModel; wider blocks: TOA_MODEL_DENSE_ROW_AD's chunked Jets are built in only)");
  if (residuals_per_item < 1 || residuals_per_item > 8 || scalars_per_item < 0 || scalars_per_item > 64 || header_scalars < 0 || header_scalars > 4096)
    return toa_fail(TOA_E_ARG, "toa_model_compile: residuals_per_item in [1, 8], scalars_per_item in [0, 64], header_scalars in [0, 4096]");
  RtcApi& api = rtc_api();
  if (!api.ok) return toa_fail(TOA_E_UNSUPPORTED, api.err);
  TOA_ON_DEVICE(h->device);
  const char* tname = dtype == TOA_F32 ? "float" : "double";
  // The functor concept of JetModel (kernels.hpp): all static; eval is generic in the scalar type S (Jet or T) and in the
  // parameter accessor X (x[j] -> S).  `h` = the problem's header scalars, `p` = the item's scalars, `r` = its residuals.
  std::string src;
  src += "#include \"kernels.hpp\"\n";
  src += "namespace toa {\ntemplate <typename T>\nstruct UserFunctor {\n";
  src += "  static constexpr int kN = " + std::to_string(num_params) + ", kR = " + std::to_string(residuals_per_item) +
         ", kD = " + std::to_string(scalars_per_item) + ", kH = " + std::to_string(header_scalars) + ";\n";
  src += "  template <class S, class X>\n  static __device__ __forceinline__ void eval(const X& x, const T* h, const T* p, S* r) {\n";
  src += "    (void)h; (void)p;\n#line 1 \"residual_body\"\n";
  src += residual_body;
  src += "\n  }\n};\n}  // namespace toa\n";
  const std::string model = std::string("toa::JetModel<") + tname + ", toa::UserFunctor<" + tname + ">>";
  const std::string k_fused = "toa::lm_fused_kernel<" + model + ">";
  const std::string k_acc = "toa::accumulate_kernel<" + model + ">";
  src += "template __global__ void " + k_fused + "(const toa::FusedParams*);\n";
  src += "template __global__ void " + k_acc + "(const void*, const void*, long long, int, int, int, void*, void*, double*, int*, int, int, double);\n";
  void* prog = nullptr;
  if (api.create(&prog, src.c_str(), "toa_user_model.hip", 0, nullptr, nullptr) != 0) return toa_fail(TOA_E_HIP, "hiprtcCreateProgram failed");
  api.add_name(prog, k_fused.c_str());
  api.add_name(prog, k_acc.c_str());
  std::vector<std::string> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I" + csrc_dir()};
  for (const std::string& inc : extra_includes()) opts.push_back("-I" + inc);
  std::vector<const char*> optp;
  for (const std::string& o : opts) optp.push_back(o.c_str());
  const int rc = api.compile(prog, int(optp.size()), optp.data());
  size_t ls = 0;
  api.log_size(prog, &ls);
  std::string log(ls, 0);
  if (ls) api.get_log(prog, &log[0]);
  if (log_out && log_cap) {
    std::snprintf(log_out, log_cap, "%s", log.c_str());
  }
  if (rc != 0) {
    api.destroy(&prog);
    return toa_fail(TOA_E_ARG, "toa_model_compile: the residual does not compile:\n" + log.substr(0, 4000));
  }
  size_t cs = 0;
  api.code_size(prog, &cs);
  std::vector<char> code(cs);
  api.get_code(prog, code.data());
  const char *n_fused = nullptr, *n_acc = nullptr;
  api.lowered(prog, k_fused.c_str(), &n_fused);
  api.lowered(prog, k_acc.c_str(), &n_acc);
  if (!n_fused || !n_acc) {
    api.destroy(&prog);
    return toa_fail(TOA_E_HIP, "toa_model_compile: hiprtc did not report the kernels' lowered names");
  }
  toa_jit_model_s* m = new (std::nothrow) toa_jit_model_s;
  if (!m) {
    api.destroy(&prog);
    return toa_fail(TOA_E_NOMEM, "out of host memory");
  }
  m->dtype = dtype; m->kN = num_params; m->kR = residuals_per_item; m->kD = scalars_per_item; m->kH = header_scalars;
  m->device = h->device;
  hipError_t e = hipModuleLoadData(&m->module, code.data());
  if (e == hipSuccess) e = hipModuleGetFunction(&m->fused, m->module, n_fused);
  if (e == hipSuccess) e = hipModuleGetFunction(&m->accumulate, m->module, n_acc);
  api.destroy(&prog);
  if (e != hipSuccess) {
    if (m->module) (void)hipModuleUnload(m->module);
    delete m;
    return toa_fail(TOA_E_HIP, std::string("toa_model_compile: loading the code object: ") + hipGetErrorString(e));
  }
  // launch geometry of the fused kernel, once
  m->lds_wg = 4 * (((dtype == TOA_F32 ? toa::WaveLds<float>::bytes(num_params) : toa::WaveLds<double>::bytes(num_params)) + 15) & ~size_t(15));
  int w = 0;
  if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&w, m->fused, 256, m->lds_wg) != hipSuccess || w < 1) w = 1;
  m->wg_per_cu = w;
  *out = m;
  return TOA_OK;
}

int toa_model_destroy(toa_jit_model m) {
  if (!m) return TOA_OK;
  if (m->module) (void)hipModuleUnload(m->module);
  delete m;
  return TOA_OK;
}

// `Optimize(x, cost)` with the run-time model: toa_lm_run's contract (include/tinyopt_amd.h) for data_dev = [P][kH + items * kD],
// m = items * kR residuals per problem, x_dev = [P][kN].
int toa_jit_lm_run(toa_handle h, toa_jit_model model, int num_items, int64_t P, const void* data_dev, void* x_dev,
                   const toa_options* options, const toa_results* results, uint64_t* counters_dev) {
  if (!h || !model) return toa_fail(TOA_E_ARG, "toa_jit_lm_run: null handle / model");
  if (model->device != h->device) return toa_fail(TOA_E_ARG, "toa_jit_lm_run: the model was compiled for another device's context");
  if (num_items < 1 || P < 0 || P > 0x7fffffff) return toa_fail(TOA_E_ARG, "toa_jit_lm_run: bad shape");
  if (!data_dev || !x_dev || !options || !results) return toa_fail(TOA_E_ARG, "toa_jit_lm_run: null pointer");
  if (!results->stop_reason || !results->num_iters || !results->final_cost)
    return toa_fail(TOA_E_ARG, "toa_jit_lm_run: stop_reason, num_iters and final_cost outputs are required");
  if (options->solver_type != 0 && options->solver_type != 1) return toa_fail(TOA_E_ARG, "toa_jit_lm_run: solver_type must be 0 (LM) or 1 (GN)");
  if ((results->errs || results->deltas2 || results->successes) && results->hist_stride < options->max_iters + 2)
    return toa_fail(TOA_E_ARG, "toa_jit_lm_run: hist_stride must be >= max_iters + 2");
  if (options->max_iters < 0 || options->max_iters > 65535) return toa_fail(TOA_E_ARG, "max_iters out of range");
  if (P == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  toa::FusedParams prm;
  std::memset(&prm, 0, sizeof(prm));
  prm.data = data_dev;
  prm.x = x_dev;
  prm.P = P;
  prm.n = model->kN;
  prm.m = num_items * model->kR;
  prm.opt = *options;
  prm.res = *results;
  prm.counters = reinterpret_cast<unsigned long long*>(counters_dev);
  prm.loss = h->loss;            // the handle's M-estimator applies to each item's squared norm (toa_set_loss)
  prm.loss_th2 = h->loss_th2;
  prm.lds_per_wave = int(model->lds_wg / 4);
  prm.queue = h->queue;
  if (h->queue_dirty) {
    HIP_TRY(hipMemsetAsync(h->queue, 0, 48 * sizeof(int), h->stream));
    h->queue_dirty = false;
  }
  long long grid = (long long)h->num_cus * model->wg_per_cu;
  const long long need = (P + 3) / 4;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  if (int rc = toa::upload_params(h, &prm, sizeof(prm))) return rc;
  const toa::FusedParams* dev = static_cast<const toa::FusedParams*>(h->params_dev);
  void* args[] = {&dev};
  const hipError_t e = hipModuleLaunchKernel(model->fused, unsigned(grid), 1, 1, 256, 1, 1, unsigned(model->lds_wg), h->stream, args, nullptr);
  if (e != hipSuccess) {
    h->queue_dirty = true;
    return toa_fail(TOA_E_HIP, std::string("toa_jit_lm_run: launch: ") + hipGetErrorString(e));
  }
  return TOA_OK;
}

// The Accumulate seam of the run-time model (toa_accumulate's contract).
int toa_jit_accumulate(toa_handle h, toa_jit_model model, int num_items, int64_t P, const void* data_dev, const void* x_dev,
                       int want_grad, void* g_dev, void* H_dev, double* cost_dev, int32_t* nres_dev) {
  if (!h || !model) return toa_fail(TOA_E_ARG, "toa_jit_accumulate: null handle / model");
  if (num_items < 1 || P < 0 || !data_dev || !x_dev || !cost_dev || (want_grad && (!g_dev || !H_dev)))
    return toa_fail(TOA_E_ARG, "toa_jit_accumulate: bad shape or null pointer");
  if (P == 0) return TOA_OK;
  TOA_ON_DEVICE(h->device);
  long long grid = (P + 3) / 4;
  const long long cap = (long long)h->num_cus * 8;
  if (grid > cap) grid = cap;
  const void* d = data_dev;
  const void* x = x_dev;
  long long Pll = P;
  int n = model->kN, m = num_items * model->kR, wg = want_grad, pw = int(model->lds_wg / 4), loss = h->loss;
  double th2 = h->loss_th2;
  void* args[] = {&d, &x, &Pll, &n, &m, &wg, &g_dev, &H_dev, &cost_dev, &nres_dev, &pw, &loss, &th2};
  const hipError_t e = hipModuleLaunchKernel(model->accumulate, unsigned(grid), 1, 1, 256, 1, 1, unsigned(model->lds_wg), h->stream, args, nullptr);
  if (e != hipSuccess) return toa_fail(TOA_E_HIP, std::string("toa_jit_accumulate: launch: ") + hipGetErrorString(e));
  return TOA_OK;
}

}  // extern "C"
