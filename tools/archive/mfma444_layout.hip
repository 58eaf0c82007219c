// Empirical lane layout of v_mfma_f64_4x4x4_4b_f64: for every (la, lb) put a single 1.0 in A at lane la and
// in B at lane lb and record which lane of D becomes non-zero.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
  const int l = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      unsigned long long m = __ballot(d != 0.0);
      if (l == 0) out[la * 64 + lb] = m ? (__ffsll((long long)m) - 1) + 100 * __popcll(m) : -1;
    }
}
int main() {
  int* d; hipMalloc(&d, 4096 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  static int h[4096]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  // For each A lane: which B lanes pair with it (same block, same k) and where the product lands
  for (int la = 0; la < 64; la++) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb] >= 0) printf("  B%02d->D%02d(x%d)", lb, h[la * 64 + lb] % 100, h[la * 64 + lb] / 100);
    printf("\n");
  }
  return 0;
}
