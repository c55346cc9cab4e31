/* A host that is NOT Python: plain C (gcc), device memory from the HIP runtime's C API, the fusion library through nothing but
 * include/tfusion.h.  Runs one nn.Linear (tf_gemm_fwd, bias epilogue) and one LayerNorm (tf_layernorm_fwd) and checks both against a
 * CPU computation written here.  Built and run by tests/test_c_abi_consumer.py; exits 0 and prints "c_abi_consumer: OK" on success.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include c_abi_consumer.c -L transfusion_amd/lib -ltfusion_hip \
 *       -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,<lib dirs>
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "tfusion.h"

static uint16_t f2bf(float f) {               /* round to nearest even */
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static float frand(uint32_t* s) {             /* LCG in [-1, 1) */
  *s = *s * 1664525u + 1013904223u;
  return (float)((*s >> 8) & 0xffffff) / 8388608.0f - 1.0f;
}
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_TF(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s: rc %d (%s)\n", #x, rc_, tf_last_error()); return 3; } } while (0)

int main(void) {
  const int M = 300, K = 128, N = 72, LDC = 72;
  if (tf_version() != TF_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
  uint16_t* hA = malloc(sizeof(uint16_t) * M * K);
  uint16_t* hW = malloc(sizeof(uint16_t) * N * K);
  uint16_t* hC = malloc(sizeof(uint16_t) * M * LDC);
  float* hb = malloc(sizeof(float) * N);
  float* hg = malloc(sizeof(float) * N);
  float* hbe = malloc(sizeof(float) * N);
  float* hy = malloc(sizeof(float) * M * N);
  uint32_t seed = 12345u;
  for (int i = 0; i < M * K; ++i) hA[i] = f2bf(frand(&seed));
  for (int i = 0; i < N * K; ++i) hW[i] = f2bf(frand(&seed) * 0.1f);
  for (int i = 0; i < N; ++i) { hb[i] = frand(&seed); hg[i] = 1.0f + 0.1f * frand(&seed); hbe[i] = 0.1f * frand(&seed); }

  void *dA, *dW, *dC, *db, *dg, *dbe, *dy, *dmean, *drstd;
  CHECK_HIP(hipMalloc(&dA, sizeof(uint16_t) * M * K));
  CHECK_HIP(hipMalloc(&dW, sizeof(uint16_t) * N * K));
  CHECK_HIP(hipMalloc(&dC, sizeof(uint16_t) * M * LDC));
  CHECK_HIP(hipMalloc(&db, sizeof(float) * N));
  CHECK_HIP(hipMalloc(&dg, sizeof(float) * N));
  CHECK_HIP(hipMalloc(&dbe, sizeof(float) * N));
  CHECK_HIP(hipMalloc(&dy, sizeof(float) * M * N));
  CHECK_HIP(hipMalloc(&dmean, sizeof(float) * M));
  CHECK_HIP(hipMalloc(&drstd, sizeof(float) * M));
  CHECK_HIP(hipMemcpy(dA, hA, sizeof(uint16_t) * M * K, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dW, hW, sizeof(uint16_t) * N * K, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(db, hb, sizeof(float) * N, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dg, hg, sizeof(float) * N, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dbe, hbe, sizeof(float) * N, hipMemcpyHostToDevice));
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));

  /* y = x W^T + b  (torch18_adapters.py:683-685 and friends) */
  TfGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = dA; g.lda = K; g.W = dW; g.ldw = K; g.C = dC; g.ldc = LDC; g.bias = (const float*)db;
  g.M = M; g.N = N; g.K = K; g.epilogue = TF_EPI_BIAS; g.drop_scale = 1.0f;
  CHECK_TF(tf_gemm_fwd(&g, (tf_stream_t)st));
  /* LayerNorm over the N columns of every row (torch18_adapters.py:110), fp32 out */
  TfLnArgs n;
  memset(&n, 0, sizeof(n));
  n.x = dC; n.ldx = LDC; n.y = dy; n.ldy = N; n.y_is_f32 = 1; n.gamma = (const float*)dg; n.beta = (const float*)dbe;
  n.mean = (float*)dmean; n.rstd = (float*)drstd; n.rows = M; n.d = N; n.rows_per_group = M; n.x_group_stride = M; n.y_group_stride = M;
  n.eps = 1e-5f;
  CHECK_TF(tf_layernorm_fwd(&n, (tf_stream_t)st));
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_HIP(hipMemcpy(hC, dC, sizeof(uint16_t) * M * LDC, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(hy, dy, sizeof(float) * M * N, hipMemcpyDeviceToHost));

  double num = 0, den = 0, lnum = 0, lden = 0;
  for (int m = 0; m < M; ++m) {
    float row[72];
    for (int c = 0; c < N; ++c) {
      double acc = hb[c];
      for (int k = 0; k < K; ++k) acc += (double)bf2f(hA[m * K + k]) * (double)bf2f(hW[c * K + k]);
      const double got = bf2f(hC[m * LDC + c]);
      num += (got - acc) * (got - acc); den += acc * acc;
      row[c] = bf2f(hC[m * LDC + c]);                  /* LayerNorm reference on what the GEMM stored */
    }
    double mean = 0, var = 0;
    for (int c = 0; c < N; ++c) mean += row[c];
    mean /= N;
    for (int c = 0; c < N; ++c) var += (row[c] - mean) * (row[c] - mean);
    var /= N;
    const double rstd = 1.0 / sqrt(var + 1e-5);
    for (int c = 0; c < N; ++c) {
      const double ref = (row[c] - mean) * rstd * hg[c] + hbe[c];
      const double got = hy[m * N + c];
      lnum += (got - ref) * (got - ref); lden += ref * ref;
    }
  }
  const double e_gemm = sqrt(num / den), e_ln = sqrt(lnum / lden);
  printf("gemm rel err %.3e (bf16 output rounding), layernorm rel err %.3e\n", e_gemm, e_ln);
  if (!(e_gemm < 4e-3) || !(e_ln < 1e-5)) { fprintf(stderr, "c_abi_consumer: MISMATCH\n"); return 4; }
  /* an argument error comes back as a negative code with a message, not a crash */
  g.K = 100;                                            /* not a multiple of 64 */
  if (tf_gemm_fwd(&g, (tf_stream_t)st) >= 0 || strlen(tf_last_error()) == 0) { fprintf(stderr, "expected an argument error\n"); return 5; }
  printf("c_abi_consumer: OK\n");
  return 0;
}
