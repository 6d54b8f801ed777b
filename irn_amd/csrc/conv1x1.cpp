// 1x1 convolutions of the channels-last ResNet-50 trunk as hipBLASLt GEMMs whose epilogue carries everything the
// reference runs as separate elementwise kernels behind the convolution (net/resnet50.py:34-54): the inference batch
// norm (the caller folds its scale into the weights and hands its shift over as the bias), the residual add (the GEMM's
// C operand, beta = 1) and the ReLU.  A channels-last activation [n, c, h, w] IS the row-major matrix [n*h*w, c], so the
// layer is out = act(x . w^T + bias (+ residual)) with no layout change on either side.
//
// hipBLASLt is column-major: D[cout x m] = W[cout x cin] . X[cin x m], i.e. A = w read transposed (ld = cin),
// B = x (ld = cin), C = residual, D = out (ld = cout); the bias runs along D's rows (= output channels).
//
// Which kernel runs is a pure function of the problem: the k-th entry of hipBLASLt's heuristic list (k = 0 unless the
// caller names another rank from a table measured once per device, irn_amd/data/gemm/), never a timing made in this
// process — two processes given the same activations produce the same bits.
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <memory>
#include <mutex>
#include <tuple>

#include "common.hpp"

namespace irn {
namespace {

constexpr int kMaxAlgos = 16;
constexpr size_t kWorkspaceBytes = 64u << 20;

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
    hipblasLtMatmulHeuristicResult_t algo[kMaxAlgos];
    int n_algos = 0;
};

// (device, m, cin, cout, bias, residual, relu, workspace, operand type: 0 fp32 / 1 fp16, leading dimension of x: 0 = cin)
using Key = std::tuple<int, int64_t, int, int, int, int, int, size_t, int, int64_t>;

std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;
std::map<Key, Plan *> g_plans;
// VOC has hundreds of image sizes, each with its own m per stage and scale: the cache is emptied when it reaches this many
// plans (a plan is four small descriptors and 16 heuristic records; rebuilding one costs a heuristic query, ~1 ms)
constexpr size_t kMaxPlans = 8192;

void destroy_plan(Plan *p) {
    if (p->desc) (void)hipblasLtMatmulDescDestroy(p->desc);
    for (hipblasLtMatrixLayout_t l : {p->a, p->b, p->c, p->d})
        if (l) (void)hipblasLtMatrixLayoutDestroy(l);
    delete p;
}

#define IRN_LT_TRY(expr)                                                                                         \
    do {                                                                                                         \
        hipblasStatus_t _s = (expr);                                                                             \
        if (_s != HIPBLAS_STATUS_SUCCESS)                                                                        \
            return ::irn::fail(IRN_ERR_HIP, "%s failed: hipblas status %d (%s:%d)", #expr, (int)_s, __FILE__, __LINE__); \
    } while (0)

int get_handle(int dev, hipblasLtHandle_t *out) {
    auto it = g_handles.find(dev);
    if (it == g_handles.end()) {
        hipblasLtHandle_t h = nullptr;
        IRN_LT_TRY(hipblasLtCreate(&h));
        it = g_handles.emplace(dev, h).first;
    }
    *out = it->second;
    return IRN_OK;
}

int epilogue_of(bool bias, bool relu) {
    if (bias) return relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS;
    return relu ? HIPBLASLT_EPILOGUE_RELU : HIPBLASLT_EPILOGUE_DEFAULT;
}

// caller holds g_mu
int get_plan(hipblasLtHandle_t handle, int dev, int64_t m, int cin, int cout, bool bias, bool residual, bool relu,
             size_t workspace_bytes, Plan **out, int f16 = 0, int64_t ldx = 0) {
    Key key(dev, m, cin, cout, bias, residual, relu, workspace_bytes, f16, ldx);
    const hipDataType ab_type = f16 ? HIP_R_16F : HIP_R_32F;      // operands; C / D / bias / scale stay fp32, fp32 accumulation
    auto it = g_plans.find(key);
    if (it != g_plans.end()) {
        *out = it->second;
        return IRN_OK;
    }
    if (g_plans.size() >= kMaxPlans) {        // nothing is in flight with a plan: the caller holds g_mu for the whole enqueue
        for (auto &kv : g_plans) destroy_plan(kv.second);
        g_plans.clear();
    }
    // owned here until it is in the cache: a failing call below frees the descriptors made so far
    std::unique_ptr<Plan, void (*)(Plan *)> owner(new Plan(), destroy_plan);
    Plan *p = owner.get();
    IRN_LT_TRY(hipblasLtMatmulDescCreate(&p->desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    const int32_t op_t = HIPBLAS_OP_T, op_n = HIPBLAS_OP_N;
    IRN_LT_TRY(hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op_t, sizeof op_t));
    IRN_LT_TRY(hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op_n, sizeof op_n));
    const uint32_t epi = (uint32_t)epilogue_of(bias, relu);
    IRN_LT_TRY(hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof epi));
    if (bias) {
        const int32_t bt = HIP_R_32F;
        IRN_LT_TRY(hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof bt));
    }
    // stored shapes (column-major): A = w [cin x cout], B = x [cin x m], C / D = [cout x m]
    IRN_LT_TRY(hipblasLtMatrixLayoutCreate(&p->a, ab_type, (uint64_t)cin, (uint64_t)cout, (int64_t)cin));
    // ldx < cin: consecutive rows of x OVERLAP (row r = memory rows r .. r + cin/ldx - 1 of a narrower matrix): the row-fused 3x3
    IRN_LT_TRY(hipblasLtMatrixLayoutCreate(&p->b, ab_type, (uint64_t)cin, (uint64_t)m, ldx ? ldx : (int64_t)cin));
    IRN_LT_TRY(hipblasLtMatrixLayoutCreate(&p->c, HIP_R_32F, (uint64_t)cout, (uint64_t)m, (int64_t)cout));
    IRN_LT_TRY(hipblasLtMatrixLayoutCreate(&p->d, HIP_R_32F, (uint64_t)cout, (uint64_t)m, (int64_t)cout));
    hipblasLtMatmulPreference_t pref = nullptr;
    IRN_LT_TRY(hipblasLtMatmulPreferenceCreate(&pref));
    const uint64_t ws = workspace_bytes;
    IRN_LT_TRY(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof ws));
    int found = 0;
    hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(handle, p->desc, p->a, p->b, p->c, p->d, pref, kMaxAlgos, p->algo, &found);
    (void)hipblasLtMatmulPreferenceDestroy(pref);
    if (st != HIPBLAS_STATUS_SUCCESS || found < 1) {
        return fail(IRN_ERR_STATE, "hipBLASLt has no %s kernel for the 1x1 convolution m=%lld cin=%d cout=%d (status %d, %d found)",
                    f16 ? "fp16 -> fp32" : "fp32", (long long)m, cin, cout, (int)st, found);
    }
    p->n_algos = found;
    g_plans[key] = owner.release();
    *out = p;
    return IRN_OK;
}

int check_shape(int64_t m, int cin, int cout) {
    if (m < 1 || cin < 1 || cout < 1) return fail(IRN_ERR_ARG, "conv1x1: m, cin, cout must be positive (%lld, %d, %d)", (long long)m, cin, cout);
    if (m > INT32_MAX) return fail(IRN_ERR_ARG, "conv1x1: m = %lld exceeds 2^31 - 1 pixels per call", (long long)m);
    return IRN_OK;
}

}  // namespace
}  // namespace irn

extern "C" {

size_t irn_conv1x1_workspace_bytes(void) { return irn::kWorkspaceBytes; }

int irn_conv1x1_algo_count(int64_t m, int cin, int cout, int has_bias, int has_residual, int relu, size_t workspace_bytes,
                           int *count_out) {
    using namespace irn;
    if (!count_out) return fail(IRN_ERR_ARG, "conv1x1_algo_count: count_out is NULL");
    if (int rc = check_shape(m, cin, cout)) return rc;
    int dev = 0;
    IRN_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    hipblasLtHandle_t handle;
    if (int rc = get_handle(dev, &handle)) return rc;
    Plan *p = nullptr;
    if (int rc = get_plan(handle, dev, m, cin, cout, has_bias != 0, has_residual != 0, relu != 0, workspace_bytes, &p)) return rc;
    *count_out = p->n_algos;
    return IRN_OK;
}

int irn_conv1x1_nhwc(const float *x_dev, const float *w_dev, const float *bias_dev, const float *residual_dev, float *out_dev,
                     int64_t m, int cin, int cout, int relu, int algo_rank, void *workspace_dev, size_t workspace_bytes,
                     void *stream) {
    using namespace irn;
    if (!x_dev || !w_dev || !out_dev) return fail(IRN_ERR_ARG, "conv1x1: x, w and out must not be NULL");
    if (int rc = check_shape(m, cin, cout)) return rc;
    if (workspace_bytes && !workspace_dev) return fail(IRN_ERR_ARG, "conv1x1: workspace_bytes > 0 with a NULL workspace");
    int dev = 0;
    IRN_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);      // the descriptor's bias pointer is per call: one enqueue at a time
    hipblasLtHandle_t handle;
    if (int rc = get_handle(dev, &handle)) return rc;
    Plan *p = nullptr;
    if (int rc = get_plan(handle, dev, m, cin, cout, bias_dev != nullptr, residual_dev != nullptr, relu != 0, workspace_bytes, &p)) return rc;
    if (algo_rank < 0 || algo_rank >= p->n_algos)
        return fail(IRN_ERR_ARG, "conv1x1: algo_rank %d outside hipBLASLt's list of %d for m=%lld cin=%d cout=%d", algo_rank, p->n_algos,
                    (long long)m, cin, cout);
    if (bias_dev) IRN_LT_TRY(hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias_dev, sizeof bias_dev));
    const float alpha = 1.0f, beta = residual_dev ? 1.0f : 0.0f;
    const float *c = residual_dev ? residual_dev : out_dev;
    IRN_LT_TRY(hipblasLtMatmul(handle, p->desc, &alpha, w_dev, p->a, x_dev, p->b, &beta, c, p->c, out_dev, p->d, &p->algo[algo_rank].algo,
                               workspace_dev, workspace_bytes, (hipStream_t)stream));
    return IRN_OK;
}

// The split-precision form: a16 = [x_hi | x_hi | x_lo'] fp16 [m, k] (irn_split16, k = 3 cin), b16 = [w_hi | w_lo | w_hi 2^-11]
// fp16 [cout, k]; one fp16 MFMA GEMM with fp32 accumulation, the same epilogue.
int irn_gemm16_algo_count(int64_t m, int k, int cout, int has_bias, int has_residual, int relu, size_t workspace_bytes, int *count_out) {
    using namespace irn;
    if (!count_out) return fail(IRN_ERR_ARG, "gemm16_algo_count: count_out is NULL");
    if (int rc = check_shape(m, k, cout)) return rc;
    int dev = 0;
    IRN_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    hipblasLtHandle_t handle;
    if (int rc = get_handle(dev, &handle)) return rc;
    Plan *p = nullptr;
    if (int rc = get_plan(handle, dev, m, k, cout, has_bias != 0, has_residual != 0, relu != 0, workspace_bytes, &p, 1)) return rc;
    *count_out = p->n_algos;
    return IRN_OK;
}

int irn_gemm16_nhwc(const void *a16_dev, const void *b16_dev, const float *bias_dev, const float *residual_dev, float *out_dev,
                    int64_t m, int k, int cout, int relu, float alpha, int algo_rank, void *workspace_dev, size_t workspace_bytes,
                    void *stream) {
    using namespace irn;
    if (!a16_dev || !b16_dev || !out_dev) return fail(IRN_ERR_ARG, "gemm16: a, b and out must not be NULL");
    if (int rc = check_shape(m, k, cout)) return rc;
    if (k & 7) return fail(IRN_ERR_ARG, "gemm16: k = %d must be a multiple of 8 (16-byte rows)", k);
    if (workspace_bytes && !workspace_dev) return fail(IRN_ERR_ARG, "gemm16: workspace_bytes > 0 with a NULL workspace");
    int dev = 0;
    IRN_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    hipblasLtHandle_t handle;
    if (int rc = get_handle(dev, &handle)) return rc;
    Plan *p = nullptr;
    if (int rc = get_plan(handle, dev, m, k, cout, bias_dev != nullptr, residual_dev != nullptr, relu != 0, workspace_bytes, &p, 1)) return rc;
    if (algo_rank < 0 || algo_rank >= p->n_algos)
        return fail(IRN_ERR_ARG, "gemm16: algo_rank %d outside hipBLASLt's list of %d for m=%lld k=%d cout=%d", algo_rank, p->n_algos,
                    (long long)m, k, cout);
    if (bias_dev) IRN_LT_TRY(hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias_dev, sizeof bias_dev));
    const float beta = residual_dev ? 1.0f : 0.0f;
    const float *c = residual_dev ? residual_dev : out_dev;
    IRN_LT_TRY(hipblasLtMatmul(handle, p->desc, &alpha, b16_dev, p->a, a16_dev, p->b, &beta, c, p->c, out_dev, p->d, &p->algo[algo_rank].algo,
                               workspace_dev, workspace_bytes, (hipStream_t)stream));
    return IRN_OK;
}

// 3x3 / pad 1 / stride 1 convolution on the zero-bordered split operand (irn_split16_pad), one call.
//   row_fused = 0: nine accumulating GEMMs over K = 3 cin, w16 = [9, cout, 3 cin];
//   row_fused = 1: THREE accumulating GEMMs over K = 9 cin, w16 = [3, cout, 9 cin] (the three taps of a kernel row side by side):
//       the taps (ky, 0..2) of pixel r are the memory rows r + (ky-1)(w+2) - 1, +0, +1 — contiguous — so the operand of a kernel
//       row is the SAME buffer read with leading dimension 3 cin and 9 cin columns (overlapping rows).  A third of the
//       read-modify-write passes over the fp32 result (which were 40 % of the nine-GEMM form's time at 512 planes).
int irn_conv3x3_split_gemm(const void *a16_dev, const void *w16_dev, float *out_dev, int64_t n_images, int h, int w, int cin, int cout,
                           float alpha, int row_fused, int algo_rank, void *workspace_dev, size_t workspace_bytes, void *stream) {
    using namespace irn;
    if (!a16_dev || !w16_dev || !out_dev) return fail(IRN_ERR_ARG, "conv3x3_split_gemm: a16, w16 and out must not be NULL");
    if (n_images < 1 || h < 1 || w < 1) return fail(IRN_ERR_ARG, "conv3x3_split_gemm: n_images, h, w must be positive");
    const int64_t m = n_images * (h + 2) * (int64_t)(w + 2);
    const int k1 = 3 * cin, k = row_fused ? 3 * k1 : k1, n_gemms = row_fused ? 3 : 9;
    if (int rc = check_shape(m, k, cout)) return rc;
    if (k1 & 7) return fail(IRN_ERR_ARG, "conv3x3_split_gemm: cin = %d must be a multiple of 8", cin);
    if (workspace_bytes && !workspace_dev) return fail(IRN_ERR_ARG, "conv3x3_split_gemm: workspace_bytes > 0 with a NULL workspace");
    int dev = 0;
    IRN_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    hipblasLtHandle_t handle;
    if (int rc = get_handle(dev, &handle)) return rc;
    const int64_t ldx = row_fused ? k1 : 0;
    Plan *first = nullptr, *rest = nullptr;          // beta = 0 for the first GEMM, 1 (C = D = out) for the others
    if (int rc = get_plan(handle, dev, m, k, cout, false, false, false, workspace_bytes, &first, 1, ldx)) return rc;
    if (int rc = get_plan(handle, dev, m, k, cout, false, true, false, workspace_bytes, &rest, 1, ldx)) return rc;
    if (int rc = get_plan(handle, dev, m, k, cout, false, false, false, workspace_bytes, &first, 1, ldx)) return rc;      // (the cache may have been emptied by the call above)
    const int r0 = algo_rank >= 0 && algo_rank < first->n_algos ? algo_rank : 0, r1 = algo_rank >= 0 && algo_rank < rest->n_algos ? algo_rank : 0;
    const char *a = (const char *)a16_dev, *wt = (const char *)w16_dev;
    const size_t row_bytes = (size_t)k1 * 2u, w_bytes = (size_t)cout * k * 2u;
    for (int t = 0; t < n_gemms; ++t) {
        const int ky = row_fused ? t : t / 3, kx = row_fused ? 0 : t % 3;
        const int64_t off = (int64_t)(ky - 1) * (w + 2) + (kx - 1);
        const float beta = t ? 1.0f : 0.0f;
        Plan *p = t ? rest : first;
        IRN_LT_TRY(hipblasLtMatmul(handle, p->desc, &alpha, wt + t * w_bytes, p->a, a + off * (int64_t)row_bytes, p->b, &beta, out_dev, p->c,
                                   out_dev, p->d, &p->algo[t ? r1 : r0].algo, workspace_dev, workspace_bytes, (hipStream_t)stream));
    }
    return IRN_OK;
}

}  // extern "C"
