// gf_tiles.hpp -- register-resident 16 x 16 blocks for the substitution kernels (csrc/gp_mll_grad_fused.hip, csrc/gp_fit_blocked.hip).
#pragma once
#include "scaml_common.hpp"

namespace scaml {

// ---- hand-managed block registers ------------------------------------------------------------------------------
// The strips' blocks live in AGPRs a[8 S : 8 S + 7] (slot S) that the COMPILER NEVER SEES as values: with the blocks as
// C++ values hipcc shuffled all 136 registers through copies at every merge of the unrolled step code (and, like in
// the fused fit, spilled).  Every access is an asm statement naming the physical registers (csrc/tile_regs.inc); the
// build sets "amdgpu-agpr-alloc"="0" so the compiler keeps out of the AGPR half, and caps the arch VGPRs so that
// VGPRs + block AGPRs <= 256 (two waves per SIMD).  Hazards hipcc cannot see inside asm: GF_DRAIN (19 wait states)
// separates the last MFMA writing a register from any non-accumulating read of it (as MFMA A/B operand,
// v_accvgpr_read, VALU); dependent accumulation into the same registers issues back to back (interlocked).
#include "tile_regs.inc"
#ifdef GF_NO_DRAIN   // (timing experiments only: the results are wrong without the wait states)
#define GF_DRAIN() asm volatile("" ::: "memory")
#else
#define GF_DRAIN() asm volatile("s_nop 15\n\ts_nop 2" ::: "memory")
#endif
#ifdef GF_NO_MFMA    // (timing experiments only: everything but the matrix instructions)
#define GF_MFMA(txt) "; " txt
#else
#define GF_MFMA(txt) txt
#endif

// The accumulator of a strip step is a block of its own, slot 17 = a[136:143] ("ACC"): compiler-visible VGPR
// accumulators were copied around by VALU moves right behind the asm MFMAs (stale reads of the last result pair).
#define GF_ACC "a[136:143]"
__device__ __forceinline__ void gf_acc_zero() {
  const int z = 0;
  asm volatile("v_accvgpr_write_b32 a136, %0\n\tv_accvgpr_write_b32 a137, %0\n\tv_accvgpr_write_b32 a138, %0\n\t"
               "v_accvgpr_write_b32 a139, %0\n\tv_accvgpr_write_b32 a140, %0\n\tv_accvgpr_write_b32 a141, %0\n\t"
               "v_accvgpr_write_b32 a142, %0\n\tv_accvgpr_write_b32 a143, %0"
               :
               : "v"(z)
               : "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");
}

template <int S>
struct GfTile;
#define GF_DEF_TILE(S, r0, r1, r2, r3, r4, r5, r6, r7)                                                             \
  template <>                                                                                                      \
  struct GfTile<S> {                                                                                               \
    /* ACC -= A * block, the block as B operand (k-step m reads register pair m) */                                \
    static __device__ __forceinline__ void chain_neg(double a0, double a1, double a2, double a3) {                 \
      asm volatile("s_nop 1\n\t"                                                                                   \
                   GF_MFMA("v_mfma_f64_16x16x4_f64 " GF_ACC ", %0, a[" #r0 ":" #r1 "], " GF_ACC " neg:[1,0,0]\n\t") \
                   GF_MFMA("v_mfma_f64_16x16x4_f64 " GF_ACC ", %1, a[" #r2 ":" #r3 "], " GF_ACC " neg:[1,0,0]\n\t") \
                   GF_MFMA("v_mfma_f64_16x16x4_f64 " GF_ACC ", %2, a[" #r4 ":" #r5 "], " GF_ACC " neg:[1,0,0]\n\t") \
                   GF_MFMA("v_mfma_f64_16x16x4_f64 " GF_ACC ", %3, a[" #r6 ":" #r7 "], " GF_ACC " neg:[1,0,0]\n\t") \
                   "s_nop 0"                                                                                       \
                   :                                                                                               \
                   : "v"(a0), "v"(a1), "v"(a2), "v"(a3)                                                            \
                   : "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");                              \
    }                                                                                                              \
    /* block = A * ACC (the caller has drained the MFMAs that wrote ACC) */                                        \
    static __device__ __forceinline__ void set_prod_acc(double a0, double a1, double a2, double a3) {              \
      asm volatile("s_nop 1\n\t"                                                                                   \
                   GF_MFMA("v_mfma_f64_16x16x4_f64 a[" #r0 ":" #r7 "], %0, a[136:137], 0\n\t")                     \
                   GF_MFMA("v_mfma_f64_16x16x4_f64 a[" #r0 ":" #r7 "], %1, a[138:139], a[" #r0 ":" #r7 "]\n\t")    \
                   GF_MFMA("v_mfma_f64_16x16x4_f64 a[" #r0 ":" #r7 "], %2, a[140:141], a[" #r0 ":" #r7 "]\n\t")    \
                   GF_MFMA("v_mfma_f64_16x16x4_f64 a[" #r0 ":" #r7 "], %3, a[142:143], a[" #r0 ":" #r7 "]\n\t")    \
                   "s_nop 7"                                                                                       \
                   :                                                                                               \
                   : "v"(a0), "v"(a1), "v"(a2), "v"(a3)                                                            \
                   : "a" #r0, "a" #r1, "a" #r2, "a" #r3, "a" #r4, "a" #r5, "a" #r6, "a" #r7);                      \
    }                                                                                                              \
    /* ACC <- block */                                                                                             \
    static __device__ __forceinline__ void copy_to_acc() {                                                         \
      asm volatile("v_accvgpr_mov_b32 a136, a" #r0 "\n\tv_accvgpr_mov_b32 a137, a" #r1 "\n\t"                     \
                   "v_accvgpr_mov_b32 a138, a" #r2 "\n\tv_accvgpr_mov_b32 a139, a" #r3 "\n\t"                     \
                   "v_accvgpr_mov_b32 a140, a" #r4 "\n\tv_accvgpr_mov_b32 a141, a" #r5 "\n\t"                     \
                   "v_accvgpr_mov_b32 a142, a" #r6 "\n\tv_accvgpr_mov_b32 a143, a" #r7                             \
                   :                                                                                               \
                   :                                                                                               \
                   : "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");                              \
    }                                                                                                              \
    /* block <- four doubles (element g = row lq + 4 g, column lc) */                                              \
    static __device__ __forceinline__ void set(double d0, double d1, double d2, double d3) {                       \
      asm volatile("v_accvgpr_write_b32 a" #r0 ", %0\n\tv_accvgpr_write_b32 a" #r1 ", %1\n\t"                     \
                   "v_accvgpr_write_b32 a" #r2 ", %2\n\tv_accvgpr_write_b32 a" #r3 ", %3\n\t"                     \
                   "v_accvgpr_write_b32 a" #r4 ", %4\n\tv_accvgpr_write_b32 a" #r5 ", %5\n\t"                     \
                   "v_accvgpr_write_b32 a" #r6 ", %6\n\tv_accvgpr_write_b32 a" #r7 ", %7"                          \
                   :                                                                                               \
                   : "v"(__double2loint(d0)), "v"(__double2hiint(d0)), "v"(__double2loint(d1)),                    \
                     "v"(__double2hiint(d1)), "v"(__double2loint(d2)), "v"(__double2hiint(d2)),                    \
                     "v"(__double2loint(d3)), "v"(__double2hiint(d3))                                              \
                   : "a" #r0, "a" #r1, "a" #r2, "a" #r3, "a" #r4, "a" #r5, "a" #r6, "a" #r7);                      \
    }                                                                                                              \
    /* four doubles <- block (the caller has drained the MFMAs that wrote it) */                                   \
    static __device__ __forceinline__ d4_t get() {                                                                 \
      int l0, h0, l1, h1, l2, h2, l3, h3;                                                                          \
      asm volatile("v_accvgpr_read_b32 %0, a" #r0 "\n\tv_accvgpr_read_b32 %1, a" #r1 "\n\t"                       \
                   "v_accvgpr_read_b32 %2, a" #r2 "\n\tv_accvgpr_read_b32 %3, a" #r3 "\n\t"                       \
                   "v_accvgpr_read_b32 %4, a" #r4 "\n\tv_accvgpr_read_b32 %5, a" #r5 "\n\t"                       \
                   "v_accvgpr_read_b32 %6, a" #r6 "\n\tv_accvgpr_read_b32 %7, a" #r7                               \
                   : "=v"(l0), "=v"(h0), "=v"(l1), "=v"(h1), "=v"(l2), "=v"(h2), "=v"(l3), "=v"(h3));              \
      d4_t r = {__hiloint2double(h0, l0), __hiloint2double(h1, l1), __hiloint2double(h2, l2),                      \
                __hiloint2double(h3, l3)};                                                                         \
      return r;                                                                                                    \
    }                                                                                                              \
  };
SCAML_TILE_LIST(GF_DEF_TILE)
#undef GF_DEF_TILE

// register slot of block j of a wave's first (SB = false) / second (SB = true) strip
template <int NBT, bool SB>
__device__ __forceinline__ constexpr int gf_slot(int j) { return SB ? j + 1 : NBT - 1 - j; }

// A operands of one block product (four k-steps)
struct GfOps {
  double a0, a1, a2, a3;
};
// (volatile: four ds_read_b64.  Left to itself hipcc merges the pairs into ds_read2_b64 / ds_read2st64_b64, which the LDS serves at
//  128 B/clk in 16-lane groups banked mod 32 dwords -- the pitches of the staged row blocks (258, 16) are laid out for
//  ds_read_b64's two 32-lane groups banked mod 64: merged, lanes lc and lc + 8 collide, 2-way, on top of the halved rate.
//  MI355X_MICROARCH.md, LDS table.  PMC before: 45 % of the strip solve's and 29 % of the fused gradient's LDS cycles were conflicts.)
__device__ __forceinline__ GfOps gf_load_ops(const double* q, int stride) {
#ifdef GF_MERGED_OPS   // (A/B: the compiler's merged reads)
  GfOps o = {q[0], q[stride], q[2 * stride], q[3 * stride]};
#else
  // (LDS-typed: a volatile access through a generic pointer keeps its flat_load)
  const volatile __attribute__((address_space(3))) double* v = (const volatile __attribute__((address_space(3))) double*)q;
  GfOps o = {v[0], v[stride], v[2 * stride], v[3 * stride]};
#endif
  return o;
}

// forward: ACC -= sum_{j = c}^{KB-1} L[KB][j] V_j, entered at the run-time first block c and falling through.  The
// operands of product j + 1 are read from LDS before the MFMAs of product j are issued (every case is a basic block
// of its own: without this each product would expose an LDS round trip); after the last product `cur` holds the
// operands at block column KB -- W_KB, what the closing product needs.
#define GF_FWD_CASE(J)                                                                                  \
  case J:                                                                                               \
    if constexpr (J < KB && J < NBT) {                                                                  \
      const GfOps nxt = gf_load_ops(pa + 16 * (J + 1), 4);                                              \
      GfTile<gf_slot<NBT, SB>(J < NBT ? J : 0)>::chain_neg(cur.a0, cur.a1, cur.a2, cur.a3);             \
      cur = nxt;                                                                                        \
    }                                                                                                   \
    [[fallthrough]];
template <int KB, int NBT, bool SB>
__device__ __forceinline__ void gf_fwd_chain(GfOps& cur, int c, const double* pa) {
  switch (c) {
    GF_FWD_CASE(0) GF_FWD_CASE(1) GF_FWD_CASE(2) GF_FWD_CASE(3) GF_FWD_CASE(4) GF_FWD_CASE(5) GF_FWD_CASE(6) GF_FWD_CASE(7)
    GF_FWD_CASE(8) GF_FWD_CASE(9) GF_FWD_CASE(10) GF_FWD_CASE(11) GF_FWD_CASE(12) GF_FWD_CASE(13) GF_FWD_CASE(14)
    default: break;
  }
}
#undef GF_FWD_CASE

// backward: ACC -= sum_{j = KB+1}^{NB-1} L[j][KB]^T Z_j, entered at the run-time last block NB-1 and falling through
// (descending; the read-ahead ends on block KB: W_KB^T)
#define GF_BWD_CASE(J)                                                                                  \
  case J:                                                                                               \
    if constexpr (J > KB && J < NBT) {                                                                  \
      const GfOps nxt = gf_load_ops(pb + 16 * (J - 1) * 16, 64);                                        \
      GfTile<gf_slot<NBT, SB>(J < NBT ? J : 0)>::chain_neg(cur.a0, cur.a1, cur.a2, cur.a3);             \
      cur = nxt;                                                                                        \
    }                                                                                                   \
    [[fallthrough]];
template <int KB, int NBT, bool SB>
__device__ __forceinline__ void gf_bwd_chain(GfOps& cur, int last, const double* pb) {
  switch (last) {
    GF_BWD_CASE(15) GF_BWD_CASE(14) GF_BWD_CASE(13) GF_BWD_CASE(12) GF_BWD_CASE(11) GF_BWD_CASE(10) GF_BWD_CASE(9)
    GF_BWD_CASE(8) GF_BWD_CASE(7) GF_BWD_CASE(6) GF_BWD_CASE(5) GF_BWD_CASE(4) GF_BWD_CASE(3) GF_BWD_CASE(2) GF_BWD_CASE(1)
    default: break;
  }
}
#undef GF_BWD_CASE

// value of an MFMA result for the VALU: 19 wait states behind the instruction (hipcc pads for the 8-pass gfx942
// instruction; on gfx950 the last result pair is not interlocked), and the data dependency keeps the uses behind it
__device__ __forceinline__ d4_t gf_settle(d4_t v) {
  asm volatile("s_nop 15\n\ts_nop 2" : "+v"(v));
  return v;
}


// ACC <- four doubles (element g = row lq + 4 g, column lc)
__device__ __forceinline__ void gf_acc_set(double d0, double d1, double d2, double d3) {
  asm volatile("v_accvgpr_write_b32 a136, %0\n\tv_accvgpr_write_b32 a137, %1\n\tv_accvgpr_write_b32 a138, %2\n\t"
               "v_accvgpr_write_b32 a139, %3\n\tv_accvgpr_write_b32 a140, %4\n\tv_accvgpr_write_b32 a141, %5\n\t"
               "v_accvgpr_write_b32 a142, %6\n\tv_accvgpr_write_b32 a143, %7"
               :
               : "v"(__double2loint(d0)), "v"(__double2hiint(d0)), "v"(__double2loint(d1)), "v"(__double2hiint(d1)),
                 "v"(__double2loint(d2)), "v"(__double2hiint(d2)), "v"(__double2loint(d3)), "v"(__double2hiint(d3))
               : "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");
}

}  // namespace scaml
