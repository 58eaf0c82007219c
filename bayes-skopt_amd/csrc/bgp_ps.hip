// The launch-free factorisation of small batches: ONE persistent kernel per batch -- ps_kernel -- whose first
// workgroups are the chain (one or two per matrix: the diagonal blocks, bgp_pf.h) and whose other workgroups are tile workers
// (this file) that draw left-looking block tasks from ticket counters.  Same arithmetic, operand order and summation order as
// the launch schedule (potrf_kernel / trsm4_kernel / syrk4_kernel): bit-identical factors and log-likelihoods.
// Replaces nothing in the reference: a scheduling choice behind cholesky() of sklearn/_gpr.py:587.
#include "bgp_s4.h"
#include "bgp_kb.h"
#include "bgp_pf.h"

// ------------------------------------------------------------------------------------------
// Tile worker of the launch-free factorisation (see ps_chain_kernel, bgp_chol.hip, for the scheme).  Left-looking by
// blocks: a task owns one 128 x 128 block (I, Jc) of matrix b, one 512-thread workgroup (8 waves), and
//   1. loads it once and applies the finished panels to its left,  C -= X_I,p X_Jc,p^T, on a FOUR-stage LDS-DMA ring
//      (three 16-wide chunks in flight: a two-stage ring spent an L2 round trip of 2-5 us on every 0.4 us chunk --
//      tools/persist_trace.py), as far as the panels are final: it waits on xready only when it has caught up with the
//      factorisation;
//   2. S(I, J), I >= J+2: stores the block, waits for W_JJ (wready) and runs the panel solve X = C W_JJ^T in place with the
//      fused right-hand-side update y_I -= X z_J (the arithmetic of trsm4_kernel), then raises xready[I][J];
//      P(I) = block (I, I-1) and Dg(I) = block (I, I), I >= 2: the PRE-updates with the panels 0 .. I-2: they store the block
//      and raise subrdy[I] / diagrdy[I] -- the chain workgroup applies the last panel, solves and factorises them itself.
// Tasks are drawn from ticket counters.  Order, per block column J = 0 .. nblk-3 and across the matrices of the batch: the
// panel solve S(J+2, J) -- the block both pre-updates of the column wait for --, then P(J+2), Dg(J+2), then S(J+3 .., J): a
// topological order of the dependency graph (every task only waits for tasks with smaller tickets and for the chain), so
// the earliest unfinished task always belongs to a running workgroup: no deadlock whatever the number of resident
// workgroups.  With PsArgs::ncrit > 0 the three tasks at the head of every column have ticket lists and workgroups of
// their own (the first ncrit of the launch): a task the chain is going to wait for never queues behind a long update
// (each list is in topological order and together they hold every task: still no deadlock; a workgroup whose pool is
// exhausted helps the other one).
// Chain pairs (PsArgs::psplit == 4; bgp_pf.h: pf_pair_helper): the per-column cycle of the chain runs THROUGH the critical tasks, and
// one CU applies a panel to a 128 x 128 block in 14 us (0.307 TF of fp64 MFMA per CU).  They go out in 64 x 64 QUADRANTS on
// workgroups of their own (ps_ll_update_quad) -- P(J+2) in four, Dg(J+2) in three, and four quadrants Q of block (J+2, J) AHEAD
// of the critical solve S(J+2, J), which then only waits for them (s2rdy) and solves -- and every quadrant consumes its LAST
// panel chunk by chunk behind the blocks that feed it: the chain helper's X_{J+1,J} and the streamed solves (pf_stream_S)
// publish a count of 16-column blocks that are complete in memory (xcol; write-through stores), chunk c of the panel is column
// block c.  Order inside a column's critical group: Q, S, P, Dg (still topological: Q waits for solves of column J-1 only).
// Per C element the operations and their order are those of syrk4_kernel / trsm4_kernel (accumulator = C, MFMA k-steps
// ascending, A-negate): bit-identical factors.
// ------------------------------------------------------------------------------------------
// critical tasks per block column and matrix: S(J+2, J), the np parts of P(J+2) (np = PsArgs::psplit: 2 column slices or 4
// quadrants), the PS_ND(np) parts of Dg(J+2) and the PS_NQ(np) quadrants Q ahead of S(J+2, J)
#define PS_ND(np) ((np) == 4 ? 3 : 1)
#define PS_NQ(np) ((np) == 4 ? 4 : 0)
static __host__ __device__ __forceinline__ int ps_crit_per_matrix(int nblk, int np) {
  return nblk > 2 ? (np + 1 + PS_ND(np) + PS_NQ(np)) * (nblk - 2) : 0;
}
static __host__ __device__ __forceinline__ int ps_bulk_per_matrix(int nblk) { return nblk > 3 ? (nblk - 3) * (nblk - 2) / 2 : 0; }
static __host__ __device__ __forceinline__ int ps_tasks_per_matrix(int nblk, int np) { return ps_crit_per_matrix(nblk, np) + ps_bulk_per_matrix(nblk); }

// vmcnt(N) with a compile-time N
template <int N>
static __device__ __forceinline__ void s4_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// eight-wave staging of a 128-row operand chunk: wave w stages rows [16 w, 16 w + 16) = two instructions of 8 rows
static __device__ __forceinline__ void s8_src(unsigned (&voff)[2], int ld, int w, int lane) {
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = 16 * w + 8 * i + (lane >> 3);
    const int g = (lane & 7) ^ ((row >> 1) & 7);
    voff[i] = (unsigned)(row * ld + g * 2) * 8u;
  }
}
static __device__ __forceinline__ void s8_issue(const double* X, const unsigned (&voff)[2], int k0, unsigned lds_op_base, int w) {
#pragma unroll
  for (int i = 0; i < 2; i++) s4_glds(X + k0, voff[i], lds_op_base + (unsigned)((16 * w + 8 * i) * S4_ROWB));
}

// vmcnt(2 n) / vmcnt(4 n) for a run-time n in 0 .. 2 (the counted waits of the four-stage ring)
static __device__ __forceinline__ void s4_wait_vm_n2(int n) {
  if (n >= 2) s4_wait_vm<4>(); else if (n == 1) s4_wait_vm<2>(); else s4_wait_vm<0>();
}
static __device__ __forceinline__ void s4_wait_vm_n4(int n) {
  if (n >= 2) s4_wait_vm<8>(); else if (n == 1) s4_wait_vm<4>(); else s4_wait_vm<0>();
}

// acc (-)= A B^T over `nch` 16-wide chunks on a four-stage LDS-DMA ring, eight waves: 128 rows of A at XA, 128 rows of B at
// XB.  Waits are counted: the DMA returns in order, so "at most r younger chunks outstanding" = vmcnt(4 r) (four
// instructions per wave and chunk).  `tri`: B is lower triangular (panel solve: chunk c only reaches the column blocks >= c).
template <int NST, int NR, int NC, int NEGA>
static __device__ __forceinline__ void s8_ring_run(const double* XA, const unsigned (&voffA)[2], const double* XB,
                                                   const unsigned (&voffB)[2], int nch, unsigned lds0,
                                                   const unsigned (&pa)[4], const unsigned (&pb)[4], d4 (&acc)[NR][NC],
                                                   int w, int tri) {
  static_assert(NST == 4, "the counted waits are written for four stages");
  constexpr unsigned AOPB = 128 * S4_ROWB, STAGEB = 256 * S4_ROWB;
#pragma unroll
  for (int s = 0; s < NST - 1; s++) {
    if (s < nch) {
      s8_issue(XA, voffA, s * S4_KC, lds0 + s * STAGEB, w);
      s8_issue(XB, voffB, s * S4_KC, lds0 + s * STAGEB + AOPB, w);
    }
  }
  for (int c = 0; c < nch; c += NST) {
#pragma unroll
    for (int s = 0; s < NST; s++) {
      if (c + s >= nch) break;            // (wave- and workgroup-uniform)
      const int rem = nch - (c + s) - 1;  // chunks behind this one
      s4_wait_vm_n4(rem < NST - 2 ? rem : NST - 2);
      __builtin_amdgcn_s_barrier();  // everybody's share of this chunk has landed; everybody is done with the previous one
      if (c + s + NST - 1 < nch) {
        const unsigned nb = lds0 + (unsigned)(((s + NST - 1) % NST) * STAGEB);
        s8_issue(XA, voffA, (c + s + NST - 1) * S4_KC, nb, w);
        s8_issue(XB, voffB, (c + s + NST - 1) * S4_KC, nb + AOPB, w);
      }
      s4_mma<NR, NC, -64, 0, NEGA>(pa, pb, s * STAGEB, acc, tri ? c + s : 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// The same ring for a DIAGONAL block's update, dt[u] -= X_ti X_tj^T on this wave's lower 16 x 16 tiles (t = w, w + 8, ... < 36
// in row-major order of the triangle: waves 0-3 five, 4-7 four): only what the factorisation reads is computed -- 36 tiles
// instead of the 48 that a 4 x 2 arrangement of 32 x 64 wave blocks covers with six busy waves of eight tiles each -- and all
// eight waves share it: 5 instead of 8 tiles on the busiest wave.  One image of the 128 rows per chunk (X_I is both operands).
// oa / ob: byte offsets of the tile's row blocks inside the image; per element the k order of every other update path.
template <int NST, int NT>
static __device__ __forceinline__ void s8_ring_run_diag(const double* XA, const unsigned (&voffA)[2], int nch, unsigned lds0,
                                                        const unsigned (&p0)[4], const unsigned (&oa)[5], const unsigned (&ob)[5],
                                                        d4 (&dt)[5], int w) {
  static_assert(NST == 4, "the counted waits are written for four stages");
  typedef __attribute__((address_space(3))) const double* lds_cdp;
  constexpr unsigned STAGEB = 256 * S4_ROWB;
#pragma unroll
  for (int s = 0; s < NST - 1; s++)
    if (s < nch) s8_issue(XA, voffA, s * S4_KC, lds0 + s * STAGEB, w);
  for (int c = 0; c < nch; c += NST) {
#pragma unroll
    for (int s = 0; s < NST; s++) {
      if (c + s >= nch) break;            // (wave- and workgroup-uniform)
      const int rem = nch - (c + s) - 1;  // chunks behind this one
      s4_wait_vm_n2(rem < NST - 2 ? rem : NST - 2);
      __builtin_amdgcn_s_barrier();  // everybody's share of this chunk has landed; everybody is done with the previous one
      if (c + s + NST - 1 < nch) s8_issue(XA, voffA, (c + s + NST - 1) * S4_KC, lds0 + (unsigned)(((s + NST - 1) % NST) * STAGEB), w);
      double a[2][NT], b[2][NT];
#pragma unroll
      for (int u = 0; u < NT; u++) {
        a[0][u] = *(lds_cdp)(uintptr_t)(p0[0] + s * STAGEB + oa[u]);
        b[0][u] = *(lds_cdp)(uintptr_t)(p0[0] + s * STAGEB + ob[u]);
      }
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk < 3) {
#pragma unroll
          for (int u = 0; u < NT; u++) {
            a[nxt][u] = *(lds_cdp)(uintptr_t)(p0[kk + 1] + s * STAGEB + oa[u]);
            b[nxt][u] = *(lds_cdp)(uintptr_t)(p0[kk + 1] + s * STAGEB + ob[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < NT; u++) dt[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][u], b[cur][u], dt[u], 0, 0, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

#define PS_NST 4
// Left-looking update of a 128 x (16 NC x 2) slice of a block with the panels 0 .. npan-1 as far as they are final (it waits on
// xready only when it has caught up with the factorisation): C -= X_I,p X_J,p^T, waves as 4 x 2, each 32 rows x 16 NC columns.
// XB = the rows of block row Jc that belong to the slice's columns (the ring stages 128 rows from there: the rows behind a
// narrower slice are staged and not read).  Returns 0, or -1 when a wait was abandoned.
template <int NC>
static __device__ __forceinline__ int ps_ll_update(const PsArgs& a, const double* XA, const double* XB, double* C, int npan,
                                                   unsigned* xrI, unsigned* xrJ, unsigned* err, int* sh_q, unsigned lds0,
                                                   const unsigned (&voffX)[2], int ld, int w, int lane, int tid, int I,
                                                   unsigned long long* tr) {
  constexpr unsigned AOPB = 128 * S4_ROWB;
  const int wr = w >> 1, wc = w & 1;
  unsigned pa[4], pb[4];
  d4 acc[2][NC];
  s4_frag_addr(pa, lds0, wr * 32, lane);
  s4_frag_addr(pb, lds0 + AOPB, wc * 16 * NC, lane);
  gk_load_c<2, NC, -64>(C, (size_t)ld, acc, wr * 32, wc * 16 * NC, lane);
  int q = 0;
  while (q < npan) {
    if (tid == 0) {
      int qq = q;
      bool ok = true;
#define PS_READY(p) (ps_ld(xrI + (p)) >= 1u && ps_ld(xrJ + (p)) >= 1u)
      while (qq < npan && PS_READY(qq)) qq++;
      if (qq == q) {  // caught up with the factorisation: wait for the next panel
        ok = ps_wait_ge2(xrI + q, 1u, xrJ + q, 1u, err, a.spin_limit);
        qq = q + 1;
        while (ok && qq < npan && PS_READY(qq)) qq++;
      }
#undef PS_READY
      ps_acquire();
      *sh_q = ok ? qq : -1;
      if (tr && q == 0) tr[1] = wall_clock64();
      if (tr && qq == npan) tr[2] = wall_clock64();  // (the last panels are ready: what follows is pure work)
    }
    __syncthreads();
    const int qq = *sh_q;
    if (qq < 0) return -1;  // abandoned
    int nch_run = (qq - q) * 8;
#ifdef BGP_FAULT_INJECT  // (see s4_tile: the same fault in the launch-free tile tasks)
    if ((BGP_FAULT_INJECT & 2) && I >= 12 && qq == npan) nch_run -= 1;
#endif
    s8_ring_run<PS_NST, 2, NC, 1>(XA + (size_t)q * 128, voffX, XB + (size_t)q * 128, voffX, nch_run, lds0, pa, pb, acc, w, 0);
    __syncthreads();  // (the ring and sh_q are free again)
    q = qq;
  }
  gk_store_c<2, NC, -64>(C, (size_t)ld, acc, wr * 32, wc * 16 * NC, lane);
  return 0;
}

// One QUADRANT (64 x 64) of a P block, the four-way split of the chain pairs' critical pre-update: 64 rows of X_I and 64 rows of
// X_Jc per chunk -- half the bytes of a 128 x 32 column slice, which stages 128 + 128 rows and reads 160 of them -- on an
// EIGHT-stage ring of 128-row images (the same 128 KB of LDS): seven chunks in flight instead of three.  The last panel's term is
// pure latency (its 8 chunks, written a microsecond ago by two other workgroups, arrive at the hand-off rate of
// MI355X_MICROARCH.md "handoff-payload"): 6.3 us -> see DESIGN.md section 4 with three 32 KB chunks in flight.  Waves as 4 x 2, each
// 16 rows x 32 columns; wave w stages rows 16 w .. 16 w + 15 of the image (waves 0-3: X_I, 4-7: X_Jc): two instructions per
// wave and chunk, "at most r younger chunks outstanding" = vmcnt(2 r).  Per element the k order of every other update path.
static __device__ __forceinline__ void q8_wait_vm(int r) {
  switch (r) {
    case 0: s4_wait_vm<0>(); break;
    case 1: s4_wait_vm<2>(); break;
    case 2: s4_wait_vm<4>(); break;
    case 3: s4_wait_vm<6>(); break;
    case 4: s4_wait_vm<8>(); break;
    case 5: s4_wait_vm<10>(); break;
    case 6: s4_wait_vm<12>(); break;
    default: s4_wait_vm<14>(); break;
  }
}
static __device__ __forceinline__ int ps_ll_update_quad(const PsArgs& a, const double* XA, const double* XB, double* C, int npan,
                                                        unsigned* xrI, unsigned* xrJ, unsigned* err, int* sh_q, unsigned lds0,
                                                        const unsigned (&voffX)[2], int ld, int w, int lane, int tid, int I,
                                                        unsigned long long* tr, const unsigned* xcA, const unsigned* xcB, int* sh_p) {
  constexpr unsigned QST = 128 * S4_ROWB;  // one stage: 64 rows of each operand
  constexpr int NST = 8;
  static_assert(NST * QST <= PF_LDS_BYTES, "the quadrant ring lives in the chain role's LDS array");
  const int wr = w >> 1, wc = w & 1;
  unsigned pa[4], pb[4];
  d4 acc[1][2];
  s4_frag_addr(pa, lds0, wr * 16, lane);
  s4_frag_addr(pb, lds0 + 64 * S4_ROWB, wc * 32, lane);
  gk_load_c<1, 2, -64>(C, (size_t)ld, acc, wr * 16, wc * 32, lane);
  // this wave's source: image row R = 16 w + ... is row R of X_I's 64 (waves 0-3) or row R - 64 of X_Jc's 64 (waves 4-7)
  const double* const Xsrc = w < 4 ? XA : XB - (size_t)64 * ld;
  int q = 0;
  while (q < npan) {
    if (tid == 0) {
      int qq = q;
      bool ok = true;
#define PS_READY(p) (ps_ld(xrI + (p)) >= 1u && ps_ld(xrJ + (p)) >= 1u)
      while (qq < npan && PS_READY(qq)) qq++;
      if (qq == q && q == npan - 1) {
        qq = -2;  // caught up at the LAST panel: follow its two blocks column block by column block (below)
      } else {
        if (qq == q) {  // caught up with the factorisation: wait for the next panel
          ok = ps_wait_ge2(xrI + q, 1u, xrJ + q, 1u, err, a.spin_limit);
          qq = q + 1;
          while (ok && qq < npan && PS_READY(qq)) qq++;
        }
        ps_acquire();
        if (!ok) qq = -1;
      }
#undef PS_READY
      *sh_q = qq;
      if (tr && q == 0) tr[1] = wall_clock64();
      if (tr && qq == npan) tr[2] = wall_clock64();  // (the last panels are ready: what follows is pure work)
    }
    __syncthreads();
    const int qq = *sh_q;
    if (qq == -1) return -1;  // abandoned
    const double* const X0 = Xsrc + (size_t)q * 128;
    if (qq == -2) {
      // ---- the last panel, streamed: chunk c = column block c of X_{I,q} (the streamed solve S(I, q)) and of X_{Jc,q} (the chain
      // helper), each handed over through xcol as its stores complete (or whole, through xready: solves that are not
      // streamed, failed matrices).  Stage c of the ring holds chunk c: nothing is reused inside the panel.
      __syncthreads();  // (sh_q is free again)
      if (tr) tr[1] = wall_clock64();  // (the streamed panel begins)
      int have = 0, issued = 0;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        if (have <= c) {
          __syncthreads();  // (everybody has read the previous round's count)
          if (tid == 0) {
            bool ok = true;
            int h = 0;
            const unsigned long long t0 = wall_clock64();
            for (unsigned it = 0;; it++) {
              const unsigned fa = ps_ld(xrI + q), fb = ps_ld(xrJ + q), ca = ps_ld(xcA), cb = ps_ld(xcB);  // (four loads in flight together)
              const int ha = fa >= 1u ? 8 : (int)ca, hb = fb >= 1u ? 8 : (int)cb;
              h = ha < hb ? ha : hb;
              if (h > c) break;
              __builtin_amdgcn_s_sleep(1);
              if ((it & 15) == 15) {
                if (ps_ld(err) != 0) {
                  ok = false;
                  break;
                }
                if (wall_clock64() - t0 > a.spin_limit) {
                  ps_st(err, 1u);
                  ok = false;
                  break;
                }
              }
            }
            ps_acquire();
            *sh_q = ok ? h : -1;
            if (tr && h == 8) tr[2] = wall_clock64();  // (both blocks are complete: what follows is pure work)
          }
          __syncthreads();
          have = *sh_q;
          __syncthreads();
          if (have < 0) return -1;  // abandoned
        }
#pragma unroll
        for (int s = 0; s < 8; s++)
          if (s >= issued && s < have) s8_issue(X0, voffX, s * S4_KC, lds0 + s * QST, w);
        issued = have;
        // a LOOK for further column blocks rides along with the wait for chunk c (its flag loads return behind this wave's chunk
        // loads, which it waits for anyway): their loads go out behind this chunk's barrier instead of after a poll of their own
        // (the count goes through one of two LDS words by the parity of c: a wave that reads late still reads ITS round's value)
        int* const slot = (c & 1) ? sh_p : sh_q;
        if (tid == 0) {
          int h = have;
          if (have < 8) {
            const unsigned fa = ps_ld(xrI + q), fb = ps_ld(xrJ + q), ca = ps_ld(xcA), cb = ps_ld(xcB);
            const int ha = fa >= 1u ? 8 : (int)ca, hb = fb >= 1u ? 8 : (int)cb;
            h = ha < hb ? ha : hb;
            if (h > have) ps_acquire();
            else h = have;
            if (tr && h == 8) tr[2] = wall_clock64();
          }
          *slot = h;
        }
        q8_wait_vm(issued - c - 1);  // (the chunks behind this one)
        pf_lds_barrier();            // (LDS only: the younger chunks stay in flight)
        have = *slot;
#pragma unroll
        for (int s = 0; s < 8; s++)
          if (s >= issued && s < have) s8_issue(X0, voffX, s * S4_KC, lds0 + s * QST, w);
        issued = have;
#ifdef BGP_FAULT_INJECT  // (see s4_tile: the same fault in the launch-free tile tasks)
        if ((BGP_FAULT_INJECT & 2) && I >= 12 && c == 7) continue;
#endif
        s4_mma<1, 2, -64, 0, 1>(pa, pb, c * QST, acc, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      break;
    }
    int nch = (qq - q) * 8;
#ifdef BGP_FAULT_INJECT  // (see s4_tile: the same fault in the launch-free tile tasks)
    if ((BGP_FAULT_INJECT & 2) && I >= 12 && qq == npan) nch -= 1;
#endif
#pragma unroll
    for (int s = 0; s < NST - 1; s++)
      if (s < nch) s8_issue(X0, voffX, s * S4_KC, lds0 + s * QST, w);
    for (int c = 0; c < nch; c += NST) {
#pragma unroll
      for (int s = 0; s < NST; s++) {
        if (c + s >= nch) break;            // (wave- and workgroup-uniform)
        const int rem = nch - (c + s) - 1;  // chunks behind this one
        q8_wait_vm(rem < NST - 2 ? rem : NST - 2);
        __builtin_amdgcn_s_barrier();  // everybody's share of this chunk has landed; everybody is done with the previous one
        if (c + s + NST - 1 < nch) s8_issue(X0, voffX, (c + s + NST - 1) * S4_KC, lds0 + (unsigned)(((s + NST - 1) % NST) * QST), w);
        s4_mma<1, 2, -64, 0, 1>(pa, pb, s * QST, acc, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();  // (the ring and sh_q are free again)
    q = qq;
  }
  gk_store_c<1, 2, -64>(C, (size_t)ld, acc, wr * 16, wc * 32, lane);
  return 0;
}

// PsArgs::gen: one Gram block (I, J) of matrix b, generated where the Gram kernel in front of the launch would have written it
// (same arithmetic per element: kb_gram_tile512); block (I, 0) also copies row block I of y into the working right-hand side.
static __device__ __forceinline__ void ps_gen_task(const PsArgs& a, int b, int I, int J) {
  const int tid = threadIdx.x;
  double* const lds = reinterpret_cast<double*>(pf_lds_raw());
  double* const xi = lds;
  double* const xj = xi + KB_DK * BGP_TILE_LD;
  double* const ell = xj + KB_DK * BGP_TILE_LD;
  if (J == 0 && tid < 128) a.yw[(size_t)b * a.ystride + I * 128 + tid] = a.y[I * 128 + tid];
  kb_gram_tile512<BGP_MATERN52, BGP_FORM_PRODUCT>(a.X, a.n, a.d, a.H + (size_t)b * (a.d + 2), a.alpha, I * 128, J * 128,
                                                 a.K + (size_t)b * a.mstride, (size_t)a.ld, a.ld, xi, xj, ell);
  ps_publish_barrier();
  if (tid == 0) ps_signal_add(a.flags + PS_GEN(a.B, a.nblk) + ((size_t)b * a.nblk + I) * a.nblk + J);
  __syncthreads();
}

// wg = this workgroup's index among the tile workgroups of the launch
template <int PAIR>
static __device__ __forceinline__ void ps_tile_role(const PsArgs& a, int wg) {
  constexpr unsigned AOPB = 128 * S4_ROWB, STAGEB = 256 * S4_ROWB;
  static_assert(PS_NST * STAGEB <= PF_LDS_BYTES, "the operand ring lives in the chain role's LDS array");
  __shared__ int sh_t, sh_q;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)pf_lds_raw();
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = a.nblk, B = a.B, ld = a.ld;
  unsigned* const flags = a.flags;
  unsigned* const err = flags + PS_ERROR;
  unsigned voffX[2], voffW[2];
  s8_src(voffX, ld, w, lane0);
  s8_src(voffW, 128, w, lane0);
  // XCD affinity (placement only): matrix b belongs to the ticket lists of XCD b % 8 -- where its chain workgroup runs
  // (block b of the chain kernel is dispatched to XCD b % 8) -- so a matrix's panels, W blocks and flags stay in ONE
  // XCD's L2 and the hand-offs are same-XCD; a workgroup whose own list is exhausted helps the next lists.
  const int xcc = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u);
  const bool pools = a.ncrit > 0;
  int pool = (pools && wg < a.ncrit) ? 0 : 1, pools_done = 0;
  int list = 0;  // lists tried so far (own first)
  for (;;) {
    const int x = (xcc + list) & 7;
    const int Bx = (B - x + 7) / 8;  // matrices b = x, x + 8, ... < B
    // parts of a P task, of a Dg task, quadrants Q ahead of the critical solve; critical tasks per column and matrix
    const int NP = a.psplit, ND = PS_ND(NP), NQ = PS_NQ(NP), NK = NQ + 1 + NP + ND;
    const int ngen = a.gen ? nblk * (nblk + 1) / 2 : 0;  // (gen: one list, the Gram blocks lead it)
    const int per_matrix = !pools ? ngen + ps_tasks_per_matrix(nblk, NP) : (pool == 0 ? ps_crit_per_matrix(nblk, NP) : ps_bulk_per_matrix(nblk));
    if (tid == 0) {
      int tt = -1;
      if (Bx > 0 && per_matrix > 0) {
        tt = (int)__hip_atomic_fetch_add(flags + PS_TICKET + 2 + 8 * pool + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tt >= Bx * per_matrix) tt = -1;
      }
      sh_t = tt;
    }
    __syncthreads();
    int t = sh_t;
    __syncthreads();
    if (t < 0) {  // this list is finished: next one, then (pools) the other pool, or done
      if (++list == 8) {
        if (!pools || ++pools_done == 2) return;
        pool ^= 1;
        list = 0;
      }
      continue;
    }
    // (trace slot: unique per (pool, list, ticket) while B % 8 == 0)
    const int tglobal = (pools && pool == 1 ? B * ps_crit_per_matrix(nblk, NP) : 0) + (int)(((long long)t * 8 + x) % ((long long)B * per_matrix));
    // ---- ticket -> (column J, matrix b, kind, block row I).  kind 0: S(I, J); 1: P(I), part `part` of NP; 2: Dg(I), part of ND;
    // 3: Q(I), quadrant `part` of block (I, J) with I = J + 2.  Order inside a column's critical group: Q, S, P, Dg -- a task only
    // ever waits for tasks with EARLIER tickets (or for the chain): Q for the bulk solves of column J - 1, S(J+2, J) for its Q.
    int J = 0, kq = NQ, I;
    if (ngen) {
      // one list: G0 G1 G2 T0 G3 T1 G4 T2 ... -- G_j = the nblk - j blocks of block column j, T_J = the tasks of column J, which touch
      // blocks of the columns J (solves), J + 1 (P) and J + 2 (Dg): still topological, and only three columns of blocks (not the
      // whole triangle) stand between the start of the launch and the first solves
      int Jg = 0;
      bool hit = false;
      for (; Jg < 3 && Jg < nblk; Jg++) {
        const int cg = (nblk - Jg) * Bx;
        if (t < cg) {
          hit = true;
          break;
        }
        t -= cg;
      }
      if (!hit)
        for (;; J++) {
          const int c = (NK + nblk - J - 3) * Bx;
          if (t < c) break;
          t -= c;
          if (J + 3 < nblk) {
            const int cg = (nblk - J - 3) * Bx;
            if (t < cg) {
              hit = true;
              Jg = J + 3;
              break;
            }
            t -= cg;
          }
        }
      if (hit) {
        ps_gen_task(a, x + 8 * (t % Bx), Jg + t / Bx, Jg);
        continue;
      }
    }
    if (pools && pool == 0) {  // NK critical tasks per column and matrix
      J = t / (NK * Bx);
      t -= J * NK * Bx;
      kq = t / Bx;
      I = J + 2;
    } else {
      const int head = pools ? 0 : NK;  // (one list: the critical tasks lead their column)
      if (!ngen)  // (gen: J and the position inside the column were found above)
        for (;;) {
          const int c = (head + nblk - J - 3) * Bx;
          if (t < c) break;
          t -= c;
          J++;
        }
      const int q0 = t / Bx;
      kq = q0 < head ? q0 : NQ;  // (a bulk task is a solve)
      I = q0 < head ? J + 2 : J + 3 + (q0 - head);
    }
    const int kind = kq < NQ ? 3 : (kq == NQ ? 0 : (kq <= NQ + NP ? 1 : 2));
    const int part = kind == 3 ? kq : (kind == 1 ? kq - NQ - 1 : kq - NQ - NP - 1);
    const int b = x + 8 * (t % Bx);
    const bool presub = kind == 1, diag = kind == 2, qpre = kind == 3;
    const int Jc = qpre ? J : J + kind;                  // block column of the task's block
    const int npan = (kind == 0 || qpre) ? J : J + 1;    // panels 0 .. npan-1 are applied here
    const bool qsolve = kind == 0 && NQ > 0 && I == J + 2;  // the critical solve: its block's pre-update came in quadrants
    int lane = lane0;
    asm volatile("" : "+v"(lane));  // (per-lane addresses of a task are formed in the task: hoisted out of this loop they spilled)
    unsigned* const wready = flags + PS_HDR + (size_t)b * nblk;
    unsigned* const diagrdy = flags + PS_HDR + (size_t)B * nblk + (size_t)b * nblk;
    unsigned* const xrI = flags + PS_HDR + (size_t)2 * B * nblk + ((size_t)b * nblk + I) * nblk;
    unsigned* const xrJ = flags + PS_HDR + (size_t)2 * B * nblk + ((size_t)b * nblk + Jc) * nblk;
    unsigned* const subrdy = flags + PS_HDR + (size_t)B * nblk * (2 + nblk) + (size_t)b * nblk;
    unsigned* const s2rdy = flags + PS_S2RDY(B, nblk) + (size_t)b * nblk;
    const int* const stat = a.status + b;
    double* const M = a.K + (size_t)b * a.mstride;
    double* const C = M + (size_t)I * 128 * ld + Jc * 128;
    unsigned long long* const tr = (a.trace && tid == 0) ? a.trace + (size_t)B * nblk * 8 + (size_t)tglobal * 8 : nullptr;
    if (tr) {
      tr[0] = wall_clock64();
      tr[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | ((unsigned long long)kind << 28) |
              ((unsigned long long)Jc << 20) | ((unsigned long long)I << 12) | (unsigned long long)b;
    }
    if (tid == 0) {
      bool ok = true;
      if (a.gen) {  // the task's own block comes from a tile worker of this launch
        ok = ps_wait_ge(flags + PS_GEN(B, nblk) + ((size_t)b * nblk + I) * nblk + Jc, 1u, err, a.spin_limit);
        ps_acquire();
      }
      sh_q = !ok ? -1 : ((__hip_atomic_load(stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0);
    }
    __syncthreads();
    if (sh_q < 0) return;   // abandoned
    bool dead = sh_q != 0;  // the matrix has failed: nothing to compute, the task only passes its flag on
    __syncthreads();
    if (qpre) {
      // ---- 1p. a quadrant of block (J+2, J) with the panels 0 .. J-1, the last one (X_{J+2,J-1}: a solve of the previous column,
      // streamed with BGP_PS_STREAM >= 3; X_{J,J-1}: the chain helper's block) chunk by chunk: the 14 us of MFMA that the last
      // panel's term costs one CU sat between the helper's block and the start of the column's critical solve
      if (npan > 0 && !dead) {
        const double* const XI = M + (size_t)I * 128 * ld;
        if (ps_ll_update_quad(a, XI + (size_t)(64 * (part >> 1)) * ld, M + ((size_t)Jc * 128 + 64 * (part & 1)) * ld,
                              C + (size_t)(64 * (part >> 1)) * ld + 64 * (part & 1), npan, xrI, xrJ, err, &sh_q, lds0, voffX, ld, w, lane, tid, I,
                              tr, flags + PS_XCOL(B, nblk) + ((size_t)b * nblk + I) * 3 + 2, flags + PS_XCOL(B, nblk) + ((size_t)b * nblk + Jc) * 3,
                              &sh_t) < 0)
          return;  // abandoned
      }
      if (tr) tr[3] = wall_clock64();
      ps_publish_barrier();
      if (tid == 0) ps_signal_add(s2rdy + I);
      if (tr) tr[6] = wall_clock64();
      __syncthreads();
      continue;
    }
    if (qsolve) {
      // ---- 1s. the critical solve's block arrives pre-updated: wait for its four quadrants
      if (tid == 0) {
        const bool ok = ps_wait_ge(s2rdy + I, 4u, err, a.spin_limit);
        ps_acquire();
        sh_q = !ok ? -1 : (__hip_atomic_load(stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 0 : 1);
        if (tr) tr[1] = tr[2] = wall_clock64();
      }
      __syncthreads();
      const int r = sh_q;
      __syncthreads();
      if (r < 0) return;
      dead = dead || r == 0;
    } else if (npan > 0 && !dead && diag && ND == 3) {
      // ---- 1q. a diagonal block in quadrants (0,0), (1,0), (1,1), each the quadrant update of a P block with X_I on both sides (the
      // diagonal quadrants compute their upper tiles too -- nobody reads those): a third of the 8 us of MFMA a whole diagonal
      // block's last panel costs one CU, and the last panel streamed behind the solve S(I, I-2) like the P quadrants
      const double* const XI = M + (size_t)I * 128 * ld;
      const int qr = part == 0 ? 0 : 1, qc = part == 2 ? 1 : 0;
      const unsigned* const xc = flags + PS_XCOL(B, nblk) + ((size_t)b * nblk + I) * 3 + 1;
      if (ps_ll_update_quad(a, XI + (size_t)(64 * qr) * ld, XI + (size_t)(64 * qc) * ld, C + (size_t)(64 * qr) * ld + 64 * qc, npan, xrI, xrI, err,
                            &sh_q, lds0, voffX, ld, w, lane, tid, I, tr, xc, xc, &sh_t) < 0)
        return;  // abandoned
    } else if (npan > 0 && !dead && diag) {
      // ---- 1d. a diagonal block: its 36 lower 16 x 16 tiles, dealt to the eight waves
      const double* const XA = M + (size_t)I * 128 * ld;
      const int lr = lane & 15, lk = lane >> 4;
      unsigned p0[4], oa[5], ob[5];
      s4_frag_addr(p0, lds0, 0, lane);
      d4 dt[5];
      int offc[5];
#pragma unroll
      for (int u = 0; u < 5; u++) {
        const int tt = w + 8 * u;
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= tt) ti++;
        const int tj = tt - ti * (ti + 1) / 2;
        oa[u] = (unsigned)(ti * 16 * S4_ROWB);
        ob[u] = (unsigned)(tj * 16 * S4_ROWB);
        offc[u] = (ti * 16 + lk) * ld + tj * 16 + lr;
        if (tt < 36) {
#pragma unroll
          for (int r = 0; r < 4; r++) dt[u][r] = C[(size_t)offc[u] + (size_t)(4 * r) * ld];
        } else {
          dt[u] = (d4){0.0, 0.0, 0.0, 0.0};
        }
      }
      int q = 0;
      while (q < npan) {
        if (tid == 0) {
          int qq = q;
          bool ok = true;
          while (qq < npan && ps_ld(xrI + qq) >= 1u) qq++;
          if (qq == q) {  // caught up with the factorisation: wait for the next panel
            ok = ps_wait_ge(xrI + q, 1u, err, a.spin_limit);
            qq = q + 1;
            while (ok && qq < npan && ps_ld(xrI + qq) >= 1u) qq++;
          }
          ps_acquire();
          sh_q = ok ? qq : -1;
          if (tr && q == 0) tr[1] = wall_clock64();
          if (tr && qq == npan) tr[2] = wall_clock64();  // (the last panels are ready: what follows is pure work)
        }
        __syncthreads();
        const int qq = sh_q;
        if (qq < 0) return;  // abandoned
        if (w < 4)
          s8_ring_run_diag<PS_NST, 5>(XA + (size_t)q * 128, voffX, (qq - q) * 8, lds0, p0, oa, ob, dt, w);
        else
          s8_ring_run_diag<PS_NST, 4>(XA + (size_t)q * 128, voffX, (qq - q) * 8, lds0, p0, oa, ob, dt, w);
        __syncthreads();  // (the ring and sh_q are free again)
        q = qq;
      }
#pragma unroll
      for (int u = 0; u < 5; u++) {
        if (w + 8 * u < 36) {
#pragma unroll
          for (int r = 0; r < 4; r++) C[(size_t)offc[u] + (size_t)(4 * r) * ld] = dt[u][r];
        }
      }
    } else if (npan > 0 && !dead) {
      // ---- 1. left-looking update with the panels 0 .. npan-1: waves as 4 x 2, each 32 rows x 64 columns (2 x 4 MFMA tiles);
      // a P task split NP ways owns 128 / NP columns of its block (each wave 32 rows x 32 or 16 columns)
      const double* const XA = M + (size_t)I * 128 * ld;
      int rc;
      if (presub && NP == 2)
        rc = ps_ll_update<2>(a, XA, M + ((size_t)Jc * 128 + 64 * part) * ld, C + 64 * part, npan, xrI, xrJ, err, &sh_q, lds0, voffX, ld, w,
                             lane, tid, I, tr);
      else if (presub && NP == 4)  // quadrants: rows 64 (part >> 1) .., columns 64 (part & 1) ..
        rc = ps_ll_update_quad(a, XA + (size_t)(64 * (part >> 1)) * ld, M + ((size_t)Jc * 128 + 64 * (part & 1)) * ld,
                               C + (size_t)(64 * (part >> 1)) * ld + 64 * (part & 1), npan, xrI, xrJ, err, &sh_q, lds0, voffX, ld, w, lane,
                               tid, I, tr, flags + PS_XCOL(B, nblk) + ((size_t)b * nblk + I) * 3 + 1,
                               flags + PS_XCOL(B, nblk) + ((size_t)b * nblk + Jc) * 3, &sh_t);
      else
        rc = ps_ll_update<4>(a, XA, M + (size_t)Jc * 128 * ld, C, npan, xrI, xrJ, err, &sh_q, lds0, voffX, ld, w, lane, tid, I, tr);
      if (rc < 0) return;  // abandoned
    }
    if (tr) tr[3] = wall_clock64();
    if (diag || presub) {
      ps_publish_barrier();
      if (tid == 0) ps_signal_add((diag ? diagrdy : subrdy) + I);
      if (tr) tr[6] = wall_clock64();
      __syncthreads();
      continue;
    }
    // ---- 2. panel solve against W_JJ: waves stacked along the rows, 16 rows x 128 columns each
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this block's updated values have left the wave
    if (PAIR && I <= J + a.ncrit_stream) {  // chain pairs: the critical solve of the column follows pf_block(J) row block by row block
      __syncthreads();  // every wave's part of the block is in memory (the solve re-reads it as A fragments)
      const int r = dead ? -2 : pf_stream_S(a, b, J, I, &sh_q, &sh_t, tr);
      if (r == -1) return;
      if (tr) tr[5] = wall_clock64();
      ps_publish_barrier();
      if (tid == 0) ps_signal_add(xrI + J);
      if (tr) tr[6] = wall_clock64();
      __syncthreads();
      continue;
    }
    if (tid == 0) {
      const bool ok = ps_wait_ge(wready + J, 1u, err, a.spin_limit);
      ps_acquire();
      sh_q = !ok ? -1 : (__hip_atomic_load(stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 0 : 1);
      if (tr) tr[4] = wall_clock64();
    }
    __syncthreads();
    if (sh_q < 0) return;
    dead = dead || sh_q == 0;
    if (!dead) {
      const double* const Wm = a.W + ((size_t)b * nblk + J) * (128 * 128);
      const int r0 = w * 16;
      unsigned pa[4], pb[4];
      s4_frag_addr(pa, lds0, r0, lane);
      s4_frag_addr(pb, lds0 + AOPB, 0, lane);
      d4 acc[1][8];
#pragma unroll
      for (int j = 0; j < 8; j++) acc[0][j] = (d4){0.0, 0.0, 0.0, 0.0};
      // (W_JJ is lower triangular: chunk c only reaches the column blocks j >= c -- `tri`; its never-written upper
      // blocks are staged all the same, which keeps the instruction count per chunk fixed for the counted waits)
      // the right-hand-side operands are fetched under the solve (z_J: this wave's 8 columns; y_I: this lane's rows)
      const double* const zk = a.yw + (size_t)b * a.ystride + J * 128;
      double zc[8], yv[4];
#pragma unroll
      for (int j = 0; j < 8; j++) zc[j] = zk[GK_COLB(0, j, lane)];
      double* const yi = a.yw + (size_t)b * a.ystride + I * 128;
#pragma unroll
      for (int r = 0; r < 4; r++) yv[r] = yi[GK_ROWB(r0, 0, lane, r)];
      s8_ring_run<PS_NST, 1, 8, 0>(C, voffX, Wm, voffW, 8, lds0, pa, pb, acc, w, 1);
      // in place (every read of these rows was staged before the last barrier), right-hand side in the same pass:
      // one wave per row, fixed shuffle order (as trsm4_kernel)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = GK_ROWB(r0, 0, lane, r);
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const double x = acc[0][j][r];
          C[(size_t)row * ld + GK_COLB(0, j, lane)] = x;
          part += x * zc[j];
        }
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 4);
        part += __shfl_xor(part, 8);
        if ((lane & 15) == 0) yi[row] = yv[r] - part;
      }
    }
    if (tr) tr[5] = wall_clock64();
    ps_publish_barrier();
    if (tid == 0) ps_signal_add(xrI + J);
    if (tr) tr[6] = wall_clock64();
    __syncthreads();
  }
}

// The launch-free factorisation: workgroups 0 .. B-1 are the chain (one per matrix, bgp_pf.h), the others the tile workers.
// One LDS array serves both roles (157 KB: one workgroup per CU); workgroups are placed in index order, so the chain is
// resident before any tile worker starts to spin.
template <int PAIR>
__global__ void __launch_bounds__(512, 1) ps_kernel(PsArgs a) {
  const int id = (int)blockIdx.x;
  if (id >= a.nchain) {
    ps_tile_role<PAIR>(a, id - a.nchain);
  } else if (!PAIR) {
    ps_chain_role<0>(a, id, 0);
  } else {
    const int p = id >= a.Bpad ? 1 : 0, b = id - p * a.Bpad;
    if (b < a.B) ps_chain_role<1>(a, b, p);  // (the padding slots of a pair group exit at once: their CUs go to tile workers)
  }
}

void bgp_launch_ps(hipStream_t st, const PsArgs& a, int nwg) {
  if (a.pair)
    hipLaunchKernelGGL(ps_kernel<1>, dim3(nwg), dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL(ps_kernel<0>, dim3(nwg), dim3(512), 0, st, a);
}
int bgp_ps_total_tasks(int B, int nblk, int np, int gen) { return B * (ps_tasks_per_matrix(nblk, np) + (gen ? nblk * (nblk + 1) / 2 : 0)); }

