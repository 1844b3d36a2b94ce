// Times u2_kmeans_assign (screening E step) through the C ABI on N x D random points; ablation switches through U2_KM_ABL.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "u2seg_hip.h"
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)
int main(int argc, char** argv) {
  const int N = 1000000, D = 768, K = 300;
  float *x, *c, *ws; long long* lab;
  HIPCHK(hipMalloc(&x, (size_t)N * D * 4)); HIPCHK(hipMalloc(&c, (size_t)K * D * 4));
  const long long wsf = u2_kmeans_assign_workspace_floats(N, D, K);
  HIPCHK(hipMalloc(&ws, wsf * 4)); HIPCHK(hipMalloc(&lab, (size_t)N * 8));
  std::vector<float> h((size_t)N * D);
  unsigned r = 1;
  for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xffff) / 32768.0f - 1.0f; }
  HIPCHK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c, h.data(), (size_t)K * D * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  for (int it = 0; it < 2; ++it) u2_kmeans_assign(x, c, ws, lab, N, D, K, 0, nullptr);
  HIPCHK(hipEventRecord(e0));
  const int reps = 5;
  for (int it = 0; it < reps; ++it) u2_kmeans_assign(x, c, ws, lab, N, D, K, 0, nullptr);
  HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  printf("abl=%s  %.3f ms per assign\n", getenv("U2_KM_ABL") ? getenv("U2_KM_ABL") : "-", ms / reps);
  return 0;
}
