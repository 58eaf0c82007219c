// Trailing-update micro-benchmark: syrk2_kernel (VGPR staging, two barriers per chunk) vs syrk4_kernel (LDS-DMA ring,
// one barrier per chunk) on one launch shape of the LML path, with a result comparison, interleaved timing rounds,
// the load-only / MFMA-only ablations of syrk4 and an in-kernel timeline.  Build: tools/build_syrk4_bench.sh.
//   syrk4_bench [B n K colmode]      defaults: 128 2048 256 0   (the largest launch of BASELINE config C)
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" void bgp_debug_launch_syrk2(hipStream_t st, int B8, int ntile, double* dK, const int* dstatus, int ld,
                                       size_t mstride, int nblk, int kp, int K, int jstart, int colmode, int B);
extern "C" int bgp_debug_launch_syrk4(int T, int var, hipStream_t st, int B8, double* dK, const int* dstatus,
                                      int ld, size_t mstride, int nblk, int kp, int K, int jstart, int colmode, int B,
                                      unsigned long long* trace);
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 128, n = argc > 2 ? atoi(argv[2]) : 2048;
  const int K = argc > 3 ? atoi(argv[3]) : 256, colmode = argc > 4 ? atoi(argv[4]) : 0;
  const int nblk = n / 128, np = argc > 5 ? nblk - atoi(argv[5]) : K / 128;  // [nt]: trailing block rows (default: all)
  const int quick = argc > 6 ? atoi(argv[6]) : 0;                            // 1: timing table only
  const size_t ms = (size_t)n * n;
  double *dK, *dK2;
  int* dst;
  hipMalloc(&dK, ms * B * 8);
  hipMalloc(&dK2, ms * B * 8);
  hipMalloc(&dst, B * 4);
  hipMemset(dst, 0, B * 4);
  std::vector<double> h(ms);
  srand(1);
  for (size_t i = 0; i < ms; i++) h[i] = ((double)rand() / RAND_MAX - 0.5);
  for (int b = 0; b < B; b++) {
    h[(size_t)b * 17 % ms] += 1e-3 * b;  // matrices differ
    hipMemcpy(dK + b * ms, h.data(), ms * 8, hipMemcpyHostToDevice);
  }
  double* dK3;  // pristine copy
  hipMalloc(&dK3, ms * B * 8);
  hipMemcpy(dK3, dK, ms * B * 8, hipMemcpyDeviceToDevice);
  hipMemcpy(dK2, dK, ms * B * 8, hipMemcpyDeviceToDevice);
  const int nt = nblk - np, ntile = colmode ? nt : nt * (nt + 1) / 2, B8 = 8 * ((B + 7) / 8);
  int grid = B8 * ntile;
  unsigned long long* dtrace;
  hipMalloc(&dtrace, (size_t)grid * 4 * 8 * 8);  // (T = 64 has up to 4x the tiles)

  // ---- results: one launch each from identical inputs, compare the matrices b = 0, 1, B-1 completely
  bgp_debug_launch_syrk2(0, B8, ntile, dK, dst, n, ms, nblk, 0, K, np, colmode, B);
  std::vector<double> r2(ms), r4(ms);
  const int chk[3] = {0, 1 % B, B - 1};
  const int cT[2] = {128, 64};
  for (int cv = 0; cv < 2; cv++) {
  hipMemcpy(dK2, dK3, ms * B * 8, hipMemcpyDeviceToDevice);
  bgp_debug_launch_syrk4(cT[cv], 0, 0, B8, dK2, dst, n, ms, nblk, 0, K, np, colmode, B, nullptr);
  hipError_t e = hipDeviceSynchronize();
  printf("syrk4 T=%d: launch status: %s\n", cT[cv], hipGetErrorString(e));
  for (int c = 0; c < 3; c++) {
    hipMemcpy(r2.data(), dK + (size_t)chk[c] * ms, ms * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r4.data(), dK2 + (size_t)chk[c] * ms, ms * 8, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    size_t nbad = 0, nchanged = 0;
    for (size_t i = 0; i < ms; i++) {
      const double d = fabs(r2[i] - r4[i]);
      if (d > md) md = d;
      if (fabs(r2[i]) > mx) mx = fabs(r2[i]);
      if (d > 1e-9) nbad++;
      if (r2[i] != h[i]) nchanged++;
    }
    printf("  matrix %3d: max |syrk2 - syrk4| = %.3e (max |value| %.3e), %zu elements differ by > 1e-9, %zu changed by syrk2\n",
           chk[c], md, mx, nbad, nchanged);
  }
  }

  // ---- timing: interleaved rounds (variants: -1 = syrk2, 0 = syrk4, 1 = syrk4 without LDS-DMA, 2 = syrk4 without
  // MFMAs, 3 = neither)
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int NV = 6;
  const int vT[NV] = {0, 128, 64, 64, 64, 64}, vV[NV] = {0, 0, 0, 1, 2, 3};
  const char* names[NV] = {"syrk2 (VGPR staging)", "syrk4 T128", "syrk4 T64", "syrk4 T64, no LDS-DMA", "syrk4 T64, no MFMA",
                           "syrk4 T64, neither"};
  std::vector<float> tms[NV];
  const int reps = 5, rounds = 5;
  for (int r = 0; r < rounds + 1; r++)
    for (int v = 0; v < (quick ? 3 : NV); v++) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < reps; i++) {
        if (vT[v] == 0)
          bgp_debug_launch_syrk2(0, B8, ntile, dK, dst, n, ms, nblk, 0, K, np, colmode, B);
        else
          bgp_debug_launch_syrk4(vT[v], vV[v], 0, B8, dK2, dst, n, ms, nblk, 0, K, np, colmode, B, nullptr);
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float msec;
      hipEventElapsedTime(&msec, e0, e1);
      if (r > 0) tms[v].push_back(msec / reps);
    }
  const double m = (double)(n - 128 * np);
  const double alg = colmode ? 2.0 * K * 128.0 * m * B : (double)K * m * (m + 1) * B;
  printf("B=%d n=%d K=%d colmode=%d tiles/matrix=%d grid=%d\n", B, n, K, colmode, ntile, grid);
  for (int v = 0; v < (quick ? 3 : NV); v++) {
    std::sort(tms[v].begin(), tms[v].end());
    const double med = tms[v][tms[v].size() / 2], mn = tms[v][0];
    printf("  %-24s median %.4f ms  min %.4f ms  -> %.1f TF algorithmic (median)\n", names[v], med, mn,
           alg / (med * 1e-3) / 1e12);
  }

  if (quick) return 0;
  // ---- timeline of syrk4 (variant 4): per tile, cycles from start to [C + chunk 0 landed] to [loop done] to [end]
  for (int pv = 0; pv < 2; pv++) {
  const int TT = pv ? 64 : 128;
  printf("timeline of syrk4 T=%d\n", TT);
  hipMemset(dtrace, 0, (size_t)grid * 4 * 8 * 8);
  grid = bgp_debug_launch_syrk4(TT, 4, 0, B8, dK2, dst, n, ms, nblk, 0, K, np, colmode, B, dtrace);
  hipDeviceSynchronize();
  std::vector<unsigned long long> tr((size_t)grid * 8);
  hipMemcpy(tr.data(), dtrace, tr.size() * 8, hipMemcpyDeviceToHost);
  double s01 = 0, s12 = 0, s23 = 0;
  double d01 = 0, d12 = 0, d23 = 0, cyc = 0, wall = 0;
  size_t cnt = 0, cntd = 0;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int g = 0; g < grid; g++) {
    const unsigned long long* t = &tr[(size_t)g * 8];
    if (t[3] == 0) continue;
    const bool diag = (t[6] / 1000) == (t[6] % 1000);
    (diag ? d01 : s01) += (double)(t[1] - t[0]);
    (diag ? d12 : s12) += (double)(t[2] - t[1]);
    (diag ? d23 : s23) += (double)(t[3] - t[2]);
    (diag ? cntd : cnt)++;
    cyc += (double)(t[3] - t[0]);
    wall += (double)(t[7] - t[4]);
    tmin = std::min(tmin, t[0]);
    tmax = std::max(tmax, t[3]);
  }
  printf("timeline (s_memtime ticks = shader cycles):\n");
  if (cnt) printf("  off-diagonal tiles (%zu): prologue %.0f  main loop %.0f  epilogue %.0f  (per chunk %.1f)\n", cnt, s01 / cnt,
                  s12 / cnt, s23 / cnt, s12 / cnt / (K / 16));
  if (cntd) printf("  diagonal tiles     (%zu): prologue %.0f  main loop %.0f  epilogue %.0f\n", cntd, d01 / cntd, d12 / cntd,
                   d23 / cntd);
  printf("  all tiles: start->end %.0f cycles on average\n", cyc / (cnt + cntd));
  printf("  shader clock while the tiles ran: %.3f GHz (s_memtime ticks per 100 MHz wall tick, summed over tiles)\n",
         cyc / wall * 0.1);
  // first 12 workgroups in detail
  for (int g = 0; g < 12 && g < grid; g++) {
    const unsigned long long* t = &tr[(size_t)g * 8];
    printf("  wg %4d tile %6llu xcc %llu: +%llu, +%llu, +%llu\n", g, t[6], t[5], t[1] - t[0], t[2] - t[1], t[3] - t[2]);
  }
  }
  return 0;
}
