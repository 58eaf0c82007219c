// What bounds the panel solve (trsm4_kernel: 3.7 TB/s on 4 GB read-modify-written in place)?  A pure data-movement probe of ITS
// access pattern next to the contiguous one: every workgroup reads a 64-row x 128-column fp64 block (1 KB per row), adds 1 and
// writes it back in place,
//   strided:    rows 1 KB long at a stride of `ld` doubles (the row-major n_pad x n_pad working matrices: ld = 2048 -> 16 KB)
//   contiguous: the same 64 KB as one contiguous run (what a tile-major storage of the working matrices would give)
// with 16 x 16-byte loads per lane in flight (the register-operand variant's pattern) over `blocks` workgroups per launch.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/archive/stride_rmw_probe.hip -o /tmp/srp && /tmp/srp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));

// 256 threads: thread t handles 16-byte column pair (t & 63) of rows (t >> 6) + 4 i, i < 16  (64 rows x 128 columns)
__global__ void __launch_bounds__(256) rmw(double* base, size_t block_stride, size_t row_stride) {
  double* p = base + (size_t)blockIdx.x * block_stride + (size_t)(threadIdx.x >> 6) * row_stride + 2 * (threadIdx.x & 63);
  d2 v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = *reinterpret_cast<const d2*>(p + (size_t)(4 * i) * row_stride);
#pragma unroll
  for (int i = 0; i < 16; i++) {
    v[i][0] += 1.0;
    v[i][1] += 1.0;
    *reinterpret_cast<d2*>(p + (size_t)(4 * i) * row_stride) = v[i];
  }
}

static double run(double* d, int blocks, size_t block_stride, size_t row_stride, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(rmw, dim3(blocks), dim3(256), 0, 0, d, block_stride, row_stride);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(rmw, dim3(blocks), dim3(256), 0, 0, d, block_stride, row_stride);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)blocks * 64 * 1024 * 2.0 * reps / (ms * 1e-3) / 1e12;  // TB/s read + written
}

int main() {
  // config C's first panel solve: 128 matrices x 15 row blocks x 2 halves = 3840 workgroups of 64 x 128 doubles
  const int n = 2048, B = 128;
  const size_t words = (size_t)B * n * n;  // 4 GiB
  double* d;
  if (hipMalloc(&d, words * 8) != hipSuccess) return 1;
  hipMemset(d, 0, words * 8);
  for (int ld : {2048, 2048 + 16, 2048 + 128}) {
    // strided: block q = (matrix b, half-row-block h): rows 128 + 64 h .. of matrix b, columns 0 .. 127; the padded leading
    // dimensions shift consecutive rows over the memory channels
    // (blocks laid out so that a launch covers 3840 distinct blocks: b * (ld * n) + h * 64 * ld)
    if ((size_t)B * ld * n > words) continue;
    const int blocks = 3840;
    // emulate (b, h) by two launches' worth of strides: 30 half blocks per matrix
    // block_stride cannot express both: use 30 launches-in-one via gridDim = 3840 with block_stride = 64 * ld and matrices packed
    // back to back every 30 blocks is not expressible either -- walk down ONE tall column instead: 3840 blocks x 64 rows = 245 760
    // rows of 1 KB at stride ld (the same row stride, the same bytes in flight; rows beyond a matrix simply continue)
    if ((size_t)blocks * 64 * ld + 128 > words) continue;
    printf("strided  ld = %5d doubles (row stride %6zu B): %.2f TB/s\n", ld, (size_t)ld * 8, run(d, blocks, (size_t)64 * ld, ld, 20));
  }
  printf("contiguous 64 KB blocks (row stride 1 KB):        %.2f TB/s\n", run(d, 3840, 64 * 128, 128, 20));
  printf("contiguous, 16 x more blocks (4 GB):              %.2f TB/s\n", run(d, 61440, 64 * 128, 128, 5));
  hipFree(d);
  return 0;
}
