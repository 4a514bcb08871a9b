// Does hipBLASLt on this box offer GELU_AUX_BIAS / DGELU_BGRAD / BGRADB plans for the FFN shapes (bf16 in/out, fp32
// accumulate), and which GELU is it (erf or tanh)?   hipcc --offload-arch=gfx950 lt_epilogue_probe.hip -lhipblaslt
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned short bf16;
static bf16 f2b(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16)(u >> 16); }
static float b2f(bf16 b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
#define CK(x) do { auto s_ = (x); if (s_ != 0) { printf("FAIL %s -> %d\n", #x, (int)s_); return 1; } } while (0)
int main() {
  const int M = 512, N = 3072, K = 768;   // row-major C[M][N] = A[M][K] W[N][K]^T  (+ bias[N]); issued column-major
  std::vector<bf16> hA(M * K), hW(N * K), hB(N);
  srand(1);
  for (auto& v : hA) v = f2b((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  for (auto& v : hW) v = f2b((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
  for (auto& v : hB) v = f2b((rand() / (float)RAND_MAX - 0.5f));
  bf16 *dA, *dW, *dB, *dC, *dAux;
  CK(hipMalloc(&dA, M * K * 2)); CK(hipMalloc(&dW, N * K * 2)); CK(hipMalloc(&dB, N * 2));
  CK(hipMalloc(&dC, M * N * 2)); CK(hipMalloc(&dAux, M * N * 2));
  CK(hipMemcpy(dA, hA.data(), M * K * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), N * K * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), N * 2, hipMemcpyHostToDevice));
  hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
  void* ws; CK(hipMalloc(&ws, 32 << 20));
  for (int ep_i = 0; ep_i < 3; ++ep_i) {
    const hipblasLtEpilogue_t eps[3] = {HIPBLASLT_EPILOGUE_GELU_AUX_BIAS, HIPBLASLT_EPILOGUE_DGELU_BGRAD, HIPBLASLT_EPILOGUE_BGRADB};
    const char* names[3] = {"GELU_AUX_BIAS", "DGELU_BGRAD", "BGRADB"};
    hipblasLtMatmulDesc_t d; CK(hipblasLtMatmulDescCreate(&d, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;   // C^T[N][M] = W[N][K] (stored K-fast => op T on (K x N) view) . A^T
    CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    hipblasLtEpilogue_t ep = eps[ep_i];
    CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
    hipDataType bt = HIP_R_16BF;
    CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    void* bp = dB; CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)));
    if (ep_i < 2) {
      void* ap = dAux; int64_t ald = N;
      CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_POINTER, &ap, sizeof(ap)));
      CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_LD, &ald, sizeof(ald)));
    }
    hipblasLtMatrixLayout_t la, lb, lc;
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, K, N, K));   // W as (K x N) column-major, ld K, used transposed
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, K, M, K));   // A^T as (K x M) column-major, ld K
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, N, M, N));
    hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
    uint64_t wsb = 32 << 20; CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb)));
    hipblasLtMatmulHeuristicResult_t res[8]; int found = 0;
    auto st = hipblasLtMatmulAlgoGetHeuristic(h, d, la, lb, lc, lc, pref, 8, res, &found);
    printf("%s: heuristic status %d, %d candidates\n", names[ep_i], (int)st, found);
    if (st != 0 || found == 0) continue;
    float alpha = 1.f, beta = 0.f;
    st = hipblasLtMatmul(h, d, &alpha, dW, la, dA, lb, &beta, dC, lc, dC, lc, &res[0].algo, ws, 32 << 20, 0);
    CK(hipDeviceSynchronize());
    printf("  matmul status %d\n", (int)st);
    if (ep_i == 0 && st == 0) {
      std::vector<bf16> hC(M * N), hX(M * N);
      CK(hipMemcpy(hC.data(), dC, M * N * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hX.data(), dAux, M * N * 2, hipMemcpyDeviceToHost));
      double e_erf = 0, e_tanh = 0, e_pre = 0;
      for (int i = 0; i < 64; ++i) for (int j = 0; j < N; j += 7) {
        double acc = b2f(hB[j]);
        for (int k = 0; k < K; ++k) acc += (double)b2f(hA[i * K + k]) * b2f(hW[j * K + k]);
        const double g_erf = 0.5 * acc * (1 + erf(acc / sqrt(2.0)));
        const double g_tanh = 0.5 * acc * (1 + tanh(0.7978845608 * (acc + 0.044715 * acc * acc * acc)));
        e_erf = fmax(e_erf, fabs(b2f(hC[i * N + j]) - g_erf)); e_tanh = fmax(e_tanh, fabs(b2f(hC[i * N + j]) - g_tanh));
        e_pre = fmax(e_pre, fabs(b2f(hX[i * N + j]) - acc));
      }
      printf("  max |out - erf-gelu| %.5f   max |out - tanh-gelu| %.5f   max |aux - preactivation| %.5f\n", e_erf, e_tanh, e_pre);
    }
  }
  printf("done\n");
  return 0;
}
