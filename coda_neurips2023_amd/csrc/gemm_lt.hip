// gemm_lt.hip -- coda_gemm_f32 / coda_gemm_ex: hipBLASLt GEMMs (fp32, or fp16 operands with fp32 accumulation)
// with per-shape cached plans (host code only).
//
// hipBLASLt is column-major.  A row-major X (r x c, row stride ld) is the column-major matrix X^T
// (c x r, leading dimension ld), so  C = op(A) op(B)  is computed as  C^T = op(B)^T op(A)^T :
// the library's first operand is B (transposed in its view iff transb), the second is A.
#include "coda_gemm.h"
#include "common.hip.h"

#include <hipblaslt/hipblaslt.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace coda {
namespace {

// Upper bound on an algorithm's scratch.  The heuristic does not always honour the preference's limit (the fp16
// problems of the image tower come back with 60+ MB stream-K scratch), so results are filtered against it and the
// per-stream buffer grows to what the chosen algorithms actually ask for.
constexpr size_t kWorkspaceBytes = 256u << 20;
constexpr size_t kWorkspaceInitial = 32u << 20;

struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t workspace = 0;
  std::vector<hipblasLtMatmulHeuristicResult_t> candidates;  // the heuristic's list, for the first-use timing
  bool tuned = false;
  int status = CODA_OK;  // != CODA_OK: the library refused this shape (cached verdict)
};

// (transa, transb, m, n, k, lda, ldb, ldc, epilogue, dtype); epilogue: 0 none, 1 +bias, 2 swish(. + bias)
using Key = std::tuple<int, int, int, int, int, long long, long long, long long, int, int>;

struct State {
  std::mutex mu;
  hipblasLtHandle_t handle = nullptr;
  hipblasLtMatmulPreference_t pref = nullptr;
  std::map<Key, Plan> plans;
  std::unordered_map<hipStream_t, std::pair<void *, size_t>> workspaces;
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
  constexpr int kAsk = 16;
  hipblasLtMatmulHeuristicResult_t res[kAsk];
  int found = 0;
  LT_CHECK(hipblasLtMatmulAlgoGetHeuristic(s.handle, p.desc, p.la, p.lb, p.lc, p.lc, s.pref, kAsk, res, &found));
  for (int i = 0; i < found; ++i)
    if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= kWorkspaceBytes) p.candidates.push_back(res[i]);
  if (getenv("CODA_LT_DEBUG"))
    fprintf(stderr, "[coda_gemm] m=%d n=%d k=%d epi=%d dtype=%d: %d candidates (%zu usable), first wants %zu B\n", m, n, k,
            epilogue, dtype, found, p.candidates.size(), p.candidates.empty() ? 0 : p.candidates[0].workspaceSize);
  if (p.candidates.empty()) return lt_error(HIPBLAS_STATUS_NOT_SUPPORTED);
  p.algo = p.candidates[0].algo;
  p.workspace = p.candidates[0].workspaceSize;
  return CODA_OK;
}

int ensure_workspace(State &s, hipStream_t hs, size_t bytes, void **out) {
  *out = nullptr;
  if (!bytes) return CODA_OK;
  auto &slot = s.workspaces[hs];
  if (slot.second < bytes) {  // grow: hipFree waits for the work that may still use the old buffer
    const size_t want = bytes > kWorkspaceInitial ? bytes : kWorkspaceInitial;
    void *buf = nullptr;
    const hipError_t e = hipMalloc(&buf, want);
    if (e != hipSuccess) return static_cast<int>(e);
    if (slot.first) (void)hipFree(slot.first);
    slot = {buf, want};
  }
  *out = slot.first;
  return CODA_OK;
}

// per-dtype default (fp16: on, fp32: off -- the fp32 path keeps the heuristic's first answer so that a process always
// runs the same kernels); CODA_GEMM_TUNE=0/1, read once, forces it for both (no setter: the library keeps no mutable
// process-wide state)
bool tuning_enabled(int dtype) {
  static const int env = [] { const char *e = getenv("CODA_GEMM_TUNE"); return e ? atoi(e) != 0 : -1; }();
  return env >= 0 ? env != 0 : dtype == CODA_DTYPE_F16;
}

// First use of a shape: time the heuristic's candidates on the caller's operands (output to a scratch matrix)
// and keep the fastest.  Skipped while the stream is being captured.
void tune(State &s, Plan &p, const void *a, const void *b, size_t c_bytes, const float *alpha, hipStream_t hs) {
  p.tuned = true;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(hs, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
  size_t need = 0;
  for (const auto &c : p.candidates) need = c.workspaceSize > need ? c.workspaceSize : need;
  void *ws = nullptr, *scratch = nullptr;
  if (ensure_workspace(s, hs, need, &ws) != CODA_OK) return;
  if (hipMalloc(&scratch, c_bytes) != hipSuccess) return;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const float beta = 0.0f;
  float best = 1e30f;
  int best_i = 0;
  for (size_t i = 0; i < p.candidates.size(); ++i) {
    const auto &c = p.candidates[i];
    auto once = [&] {
      return hipblasLtMatmul(s.handle, p.desc, alpha, b, p.la, a, p.lb, &beta, scratch, p.lc, scratch, p.lc, &c.algo, ws,
                             c.workspaceSize, hs);
    };
    if (once() != HIPBLAS_STATUS_SUCCESS) continue;  // warm-up, and "does it run"
    (void)hipEventRecord(e0, hs);
    for (int r = 0; r < 3; ++r) (void)once();
    (void)hipEventRecord(e1, hs);
    if (hipEventSynchronize(e1) != hipSuccess) continue;
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (getenv("CODA_LT_DEBUG")) fprintf(stderr, "[coda_gemm]   candidate %zu: %.1f us, workspace %zu B\n", i, ms * 1000.0f / 3, c.workspaceSize);
    if (ms < best) { best = ms; best_i = static_cast<int>(i); }
  }
  (void)hipGetLastError();
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(scratch);
  p.algo = p.candidates[best_i].algo;
  p.workspace = p.candidates[best_i].workspaceSize;
  if (getenv("CODA_LT_DEBUG")) fprintf(stderr, "[coda_gemm]   kept candidate %d\n", best_i);
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
    const uint64_t ws = kWorkspaceInitial;  // the hint; see kWorkspaceBytes
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
  if (bias) LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
  if (!p.tuned && p.candidates.size() > 1 && tuning_enabled(dtype))
    tune(s, p, a, b, static_cast<size_t>(m) * ldc * (dtype == CODA_DTYPE_F16 ? 2 : 4), &alpha, hs);
  void *ws = nullptr;
  {
    const int wst = ensure_workspace(s, hs, p.workspace, &ws);
    if (wst != CODA_OK) return wst;
  }
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
