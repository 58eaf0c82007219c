// LDS-DMA primitives shared by the ring kernels (bgp_syrk4.hip, bgp_kbuild.hip).
#pragma once
#include "bgp_common.h"

// One LDS-DMA instruction: 64 lanes x 16 B from (wave-uniform base + per-lane byte offset) to LDS
// [lds_addr, lds_addr + 1 KB).  hipcc neither counts it in its vmcnt bookkeeping nor waits for it: the kernel
// places its own waits.  (M0 is compiler-reserved: saved and restored inside the statement,
// cdna_hip_programming.md 5.7.)
static __device__ __forceinline__ void s4_glds(const double* gbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(gbase), "s"(lds_addr)
      : "memory");
}
#define S4_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

