// gemm_lt.hip -- coda_gemm_f32 / coda_gemm_ex: hipBLASLt GEMMs (fp32, or fp16 operands with fp32 accumulation)
// with per-shape cached plans (host code only).
//
// hipBLASLt is column-major.  A row-major X (r x c, row stride ld) is the column-major matrix X^T
// (c x r, leading dimension ld), so  C = op(A) op(B)  is computed as  C^T = op(B)^T op(A)^T :
// the library's first operand is B (transposed in its view iff transb), the second is A.
#include "coda_gemm.h"
#include "common.hip.h"

#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>
#include <unordered_map>

namespace coda {
namespace {

constexpr size_t kWorkspaceBytes = 32u << 20;

struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t workspace = 0;
  int status = CODA_OK;  // != CODA_OK: the library refused this shape (cached verdict)
};

// (transa, transb, m, n, k, lda, ldb, ldc, epilogue, dtype); epilogue: 0 none, 1 +bias, 2 swish(. + bias)
using Key = std::tuple<int, int, int, int, int, long long, long long, long long, int, int>;

struct State {
  std::mutex mu;
  hipblasLtHandle_t handle = nullptr;
  hipblasLtMatmulPreference_t pref = nullptr;
  std::map<Key, Plan> plans;
  std::unordered_map<hipStream_t, void *> workspaces;
};

// one State per device: the hipBLASLt handle, the plans' heuristics and the workspaces belong to it
State &state() {
  static std::mutex mu;
  static std::map<int, State> per_device;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  return per_device[dev];
}

int lt_error(hipblasStatus_t st) { return -(2000 + static_cast<int>(st)); }

#define LT_CHECK(expr)                                   \
  do {                                                   \
    hipblasStatus_t st_ = (expr);                        \
    if (st_ != HIPBLAS_STATUS_SUCCESS) return lt_error(st_); \
  } while (0)

int make_plan_checked(State &s, const Key &key, Plan &p);

// Plan creation: any failure means "the library has nothing for this problem" and is reported as
// -(3000 + hipblasStatus) -- the caller may route that ONE shape elsewhere -- with the descriptors
// released and the verdict cached (Plan::status), so a refused shape is not re-queried on every call.
int make_plan(State &s, const Key &key, Plan &p) {
  const int st = make_plan_checked(s, key, p);
  if (st != CODA_OK) {
    if (p.lc) (void)hipblasLtMatrixLayoutDestroy(p.lc);
    if (p.lb) (void)hipblasLtMatrixLayoutDestroy(p.lb);
    if (p.la) (void)hipblasLtMatrixLayoutDestroy(p.la);
    if (p.desc) (void)hipblasLtMatmulDescDestroy(p.desc);
    p = Plan{};
    p.status = st <= -2000 ? st - 1000 : st;
  }
  return p.status;
}

int make_plan_checked(State &s, const Key &key, Plan &p) {
  const auto [transa, transb, m, n, k, lda, ldb, ldc, epilogue, dtype] = key;
  const hipDataType ty = dtype == CODA_DTYPE_F16 ? HIP_R_16F : HIP_R_32F;
  LT_CHECK(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  // library operand 1 = B, operand 2 = A (see the header comment)
  const hipblasOperation_t op1 = transb ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  const hipblasOperation_t op2 = transa ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op1, sizeof(op1)));
  LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op2, sizeof(op2)));
  if (epilogue) {
    const hipblasLtEpilogue_t epi = epilogue == 2 ? HIPBLASLT_EPILOGUE_SWISH_BIAS_EXT : HIPBLASLT_EPILOGUE_BIAS;
    LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
    const int32_t bias_ty = HIP_R_32F;  // bias vectors stay fp32 whatever the operand type
    LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bias_ty, sizeof(bias_ty)));
  }
  // column-major shapes as stored: B row-major (k x n) -> (n x k); transb: stored (n x k) -> (k x n)
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.la, ty, transb ? k : n, transb ? n : k, ldb));
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.lb, ty, transa ? m : k, transa ? k : m, lda));
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.lc, ty, n, m, ldc));
  hipblasLtMatmulHeuristicResult_t res;
  int found = 0;
  LT_CHECK(hipblasLtMatmulAlgoGetHeuristic(s.handle, p.desc, p.la, p.lb, p.lc, p.lc, s.pref, 1, &res, &found));
  if (found < 1) return lt_error(HIPBLAS_STATUS_NOT_SUPPORTED);
  p.algo = res.algo;
  p.workspace = res.workspaceSize;
  return CODA_OK;
}

}  // namespace
}  // namespace coda

CODA_API int coda_gemm_ex(int dtype, int epilogue, int transa, int transb, int m, int n, int k, const void *a,
                          long long lda, const void *b, long long ldb, void *c, long long ldc, const float *bias,
                          float alpha, float beta, void *stream) {
  using namespace coda;
  if (m < 0 || n < 0 || k < 0) return CODA_EINVAL;
  if (dtype != CODA_DTYPE_F32 && dtype != CODA_DTYPE_F16) return CODA_EINVAL;
  if (epilogue < 0 || epilogue > 2 || (epilogue != 0) != (bias != nullptr)) return CODA_EINVAL;
  if (m == 0 || n == 0) return CODA_OK;
  if (k == 0 || !a || !b || !c) return CODA_EINVAL;
  if (lda < (transa ? m : k) || ldb < (transb ? k : n) || ldc < n) return CODA_EINVAL;
  State &s = state();
  std::lock_guard<std::mutex> lock(s.mu);
  if (!s.handle) {
    LT_CHECK(hipblasLtCreate(&s.handle));
    LT_CHECK(hipblasLtMatmulPreferenceCreate(&s.pref));
    const uint64_t ws = kWorkspaceBytes;
    LT_CHECK(hipblasLtMatmulPreferenceSetAttribute(s.pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws,
                                                   sizeof(ws)));
  }
  const Key key{transa != 0, transb != 0, m, n, k, lda, ldb, ldc, epilogue, dtype};
  auto it = s.plans.find(key);
  if (it == s.plans.end()) {
    Plan p;
    (void)make_plan(s, key, p);
    it = s.plans.emplace(key, p).first;
  }
  Plan &p = it->second;
  if (p.status != CODA_OK) return p.status;
  hipStream_t hs = static_cast<hipStream_t>(stream);
  void *ws = nullptr;
  if (p.workspace) {
    auto w = s.workspaces.find(hs);
    if (w == s.workspaces.end()) {
      void *buf = nullptr;
      const hipError_t e = hipMalloc(&buf, kWorkspaceBytes);
      if (e != hipSuccess) return static_cast<int>(e);
      w = s.workspaces.emplace(hs, buf).first;
    }
    ws = w->second;
  }
  if (bias) LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
  LT_CHECK(hipblasLtMatmul(s.handle, p.desc, &alpha, b, p.la, a, p.lb, &beta, c, p.lc, c, p.lc, &p.algo, ws,
                           p.workspace, hs));
  return CODA_OK;
}

CODA_API int coda_gemm_f32(int transa, int transb, int m, int n, int k, const float *a, long long lda,
                           const float *b, long long ldb, float *c, long long ldc, const float *bias,
                           int accumulate, void *stream) {
  return coda_gemm_ex(CODA_DTYPE_F32, bias ? 1 : 0, transa, transb, m, n, k, a, lda, b, ldb, c, ldc, bias, 1.0f,
                      accumulate ? 1.0f : 0.0f, stream);
}
