// valu_issue.hip -- developer probe (not part of the library): cycles per wave64 instruction per SIMD on gfx950.
//
// Every wave runs ITER x UNROLL copies of ONE instruction on NACC independent registers (inline asm, so the compiler neither
// removes nor re-schedules them) between two s_memtime reads.  W waves per SIMD run that loop side by side (a workgroup is
// 4*W waves; one workgroup per CU, forced by its LDS allocation), so  cycles / (W * instructions)  is the SIMD's issue cost of
// the instruction at that occupancy.  NACC = 1 gives the dependent-chain latency instead.  s_memtime ticks at the shader clock
// (checked against the wall time of the launch).  Output = the table kept as profiles/r3_valu_issue.txt; bench.py's
// roofline_valu reads the constants from it (tools/make_valu_table.py).
//
//   build: make -C rnnoise_amd/csrc tools ; run: rnnoise_amd/csrc/build/valu_issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum Op { FMA32, ADD32, MUL32, ADD32_DPP_ROWSHR, ADD32_DPP_QUAD, MOV_DPP_ROWROR, PKADD32, PKFMA32, PKMUL32, FMA64, ADD64, MUL64, DOT4, ADDU32, MADU24,
          MED3, CVTPKU8, RCP32, EXP32, PERM, CNDMASK, CMP32, DSREAD32, DSREAD64, DSREAD128, DSSWIZZLE, DSBPERMUTE, PERMLANE32SWAP, FMA32_DSREAD, FMA32_SALU, NOPS,
          MOV32, AND32, LSHL32, XOR32, MAX32, SUB32, CNDMASK_S, CMP_S, BFE32, LSHLADD, READLANE, MULLO, ADD32_SGPR, MUL32_LIT, DSWRITE32, DSWRITE128, DSREADU16, DPP_WAVESHR, ADD3, CNDMASK_E64VCC, CMP_CND_VCC, CMP_CND_S, CNDMASK_VCC_INIT, CMP_CND2, CMP_CND4, CND_ADD, CMP_ADD_CND2, DPP_FMA, F64_FMA, LSHL_FMA, MAX_FMA };
static const char *OPN[] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_add_f32 dpp row_shr:1", "v_add_f32 dpp quad_perm", "v_mov_b32 dpp row_ror:8", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mul_f32",
                            "v_fma_f64", "v_add_f64", "v_mul_f64", "v_dot4_i32_i8", "v_add_u32", "v_mad_u32_u24", "v_med3_f32", "v_cvt_pk_u8_f32", "v_rcp_f32", "v_exp_f32", "v_perm_b32",
                            "v_cndmask_b32", "v_cmp_lt_f32", "ds_read_b32", "ds_read_b64", "ds_read_b128", "ds_swizzle_b32", "ds_bpermute_b32", "v_permlane32_swap", "v_fma_f32 + ds_read_b32 (1:1)",
                            "v_fma_f32 + s_add_u32 (1:1)", "s_nop 0",
                            "v_mov_b32", "v_and_b32", "v_lshlrev_b32", "v_xor_b32", "v_max_f32", "v_sub_f32", "v_cndmask_b32 (sgpr mask)", "v_cmp_lt_f32 -> sgpr pair", "v_bfe_u32", "v_lshl_add_u32",
                            "v_readlane_b32", "v_mul_lo_u32", "v_add_f32 (sgpr operand)", "v_mul_f32 (literal)", "ds_write_b32", "ds_write_b128", "ds_read_u16", "v_add_f32 dpp wave_shr:1", "v_add3_u32",
                            "v_cndmask_b32_e64 (vcc spelled)", "v_cmp->vcc + v_cndmask vcc (pair)", "v_cmp->sgpr + v_cndmask sgpr (pair)", "v_cndmask_b32 vcc (vcc set before)",
                            "v_cmp->vcc + 2 x v_cndmask vcc", "v_cmp->vcc + 4 x v_cndmask vcc", "v_cndmask vcc + v_add_f32 (1:1)", "v_cmp->vcc, add, cnd, add, cnd",
                            "v_add_f32 dpp + v_fma_f32 (1:1)", "v_fma_f64 + v_fma_f32 (1:1)", "v_lshlrev_b32 + v_fma_f32 (1:1)", "v_max_f32 + v_fma_f32 (1:1)"};
constexpr int UNROLL = 64;

// UNROLL = 64 instructions on 8 (or 1, NACC == 1) accumulators, as ONE asm statement (hipcc puts an s_nop after every
// inline-asm statement that it cannot see into)
#define G8_(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define G1_(T) T(0) T(0) T(0) T(0) T(0) T(0) T(0) T(0)
#define G8(T) G8_(T) G8_(T) G8_(T) G8_(T) G8_(T) G8_(T) G8_(T) G8_(T)
#define G1(T) G1_(T) G1_(T) G1_(T) G1_(T) G1_(T) G1_(T) G1_(T) G1_(T)
#define EMIT(T, r)                                                                                                               \
  do {                                                                                                                           \
    if constexpr (NACC == 8)                                                                                                     \
      asm volatile(G8(T) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])        \
                   : [c0] "v"(c0), [c1] "v"(c1), [ad] "v"(lds_addr), [ad2] "v"(lds_addr * 2) : "vcc", "s20", "s22", "s23", "s24", "scc");  \
    else                                                                                                                         \
      asm volatile(G1(T) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])        \
                   : [c0] "v"(c0), [c1] "v"(c1), [ad] "v"(lds_addr), [ad2] "v"(lds_addr * 2) : "vcc", "s20", "s22", "s23", "s24", "scc");  \
  } while (0)
#define T_FMA32(n) "v_fma_f32 %" #n ", %" #n ", %[c0], %[c1]\n"
#define T_ADD32(n) "v_add_f32 %" #n ", %" #n ", %[c0]\n"
#define T_MUL32(n) "v_mul_f32 %" #n ", %" #n ", %[c0]\n"
#define T_DPPSHR(n) "v_add_f32_dpp %" #n ", %" #n ", %[c0] row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_DPPQUAD(n) "v_add_f32_dpp %" #n ", %" #n ", %[c0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_DPPROR(n) "v_mov_b32_dpp %" #n ", %" #n " row_ror:8 row_mask:0xf bank_mask:0xf\n"
#define T_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %7\n"
#define T_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %7, %7\n"
#define T_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %7\n"
#define T_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %7, %7\n"
#define T_ADD64(n) "v_add_f64 %" #n ", %" #n ", %7\n"
#define T_MUL64(n) "v_mul_f64 %" #n ", %" #n ", %7\n"
#define T_DOT4(n) "v_dot4_i32_i8 %" #n ", %[ad], %[ad], %" #n "\n"
#define T_ADDU32(n) "v_add_u32 %" #n ", %" #n ", %[ad]\n"
#define T_MADU24(n) "v_mad_u32_u24 %" #n ", %" #n ", %[ad], %[ad]\n"
#define T_MED3(n) "v_med3_f32 %" #n ", %" #n ", %[c0], %[c1]\n"
#define T_CVTPKU8(n) "v_cvt_pk_u8_f32 %" #n ", %[c0], 1, %" #n "\n"
#define T_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define T_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define T_PERM(n) "v_perm_b32 %" #n ", %" #n ", %[ad], %[ad]\n"
#define T_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %[c0], vcc\n"
#define T_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %[c0]\n"
#define T_DSR32(n) "ds_read_b32 %" #n ", %[ad]\n"
#define T_DSR64(n) "ds_read_b64 %" #n ", %[ad2]\n"
#define T_DSSWZ(n) "ds_swizzle_b32 %" #n ", %" #n " offset:0x401f\n"
#define T_DSBPERM(n) "ds_bpermute_b32 %" #n ", %[ad], %" #n "\n"
#define T_PL32(n) "v_permlane32_swap_b32 %" #n ", %7\n"
#define T_FMADS(n) "v_fma_f32 %" #n ", %" #n ", %[c0], %[c1]\n ds_read_b32 %7, %[ad]\n"
#define T_FMASALU(n) "v_fma_f32 %" #n ", %" #n ", %[c0], %[c1]\n s_add_u32 s20, s20, 1\n"
#define T_NOP(n) "s_nop 0\n"
#define T_MOV(n) "v_mov_b32 %" #n ", %[c0]\n"
#define T_AND(n) "v_and_b32 %" #n ", %" #n ", %[ad]\n"
#define T_LSHL(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define T_XOR(n) "v_xor_b32 %" #n ", %" #n ", %[ad]\n"
#define T_MAX(n) "v_max_f32 %" #n ", %" #n ", %[c0]\n"
#define T_SUB(n) "v_sub_f32 %" #n ", %" #n ", %[c0]\n"
#define T_CNDS(n) "v_cndmask_b32 %" #n ", %" #n ", %[c0], s[22:23]\n"
#define T_CMPS(n) "v_cmp_lt_f32 s[22:23], %" #n ", %[c0]\n"
#define T_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 12\n"
#define T_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %[ad]\n"
#define T_READLANE(n) "v_readlane_b32 s20, %" #n ", 5\n"
#define T_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %[ad]\n"
#define T_ADDS(n) "v_add_f32 %" #n ", s24, %" #n "\n"
#define T_MULLIT(n) "v_mul_f32 %" #n ", 0x3f8ccccd, %" #n "\n"
#define T_DSW32(n) "ds_write_b32 %[ad], %" #n "\n"
#define T_DSRU16(n) "ds_read_u16 %" #n ", %[ad]\n"
#define T_WAVESHR(n) "v_add_f32_dpp %" #n ", %" #n ", %[c0] wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_CND64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %[c0], vcc\n"
#define T_CMPCNDV(n) "v_cmp_lt_f32 vcc, %" #n ", %[c0]\n v_cndmask_b32 %" #n ", %" #n ", %[c0], vcc\n"
#define T_CMPCNDS(n) "v_cmp_lt_f32 s[22:23], %" #n ", %[c0]\n v_cndmask_b32 %" #n ", %" #n ", %[c0], s[22:23]\n"
#define T_CMPCND2(n) "v_cmp_lt_f32 vcc, %" #n ", %[c0]\n v_cndmask_b32 %" #n ", %" #n ", %[c0], vcc\n v_cndmask_b32 %" #n ", %" #n ", %[c1], vcc\n"
#define T_CMPCND4(n) "v_cmp_lt_f32 vcc, %" #n ", %[c0]\n v_cndmask_b32 %" #n ", %" #n ", %[c0], vcc\n v_cndmask_b32 %" #n ", %" #n ", %[c1], vcc\n v_cndmask_b32 %" #n ", %" #n ", %[c0], vcc\n v_cndmask_b32 %" #n ", %" #n ", %[c1], vcc\n"
#define T_CNDADD(n) "v_cndmask_b32 %" #n ", %" #n ", %[c0], vcc\n v_add_f32 %" #n ", %" #n ", %[c1]\n"
#define T_CMPADDCND2(n) "v_cmp_lt_f32 vcc, %" #n ", %[c0]\n v_add_f32 %" #n ", %" #n ", %[c1]\n v_cndmask_b32 %" #n ", %" #n ", %[c0], vcc\n v_add_f32 %" #n ", %" #n ", %[c1]\n v_cndmask_b32 %" #n ", %" #n ", %[c1], vcc\n"
#define T_DPPFMA(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n v_fma_f32 %" #n ", %" #n ", %[c0], %[c1]\n"
#define T_MAXFMA(n) "v_max_f32 %" #n ", %" #n ", %[c0]\n v_fma_f32 %" #n ", %" #n ", %[c0], %[c1]\n"
#define T_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %[ad], %[ad]\n"
template <int OP, int NACC>
__device__ __forceinline__ void body(float (&a)[8], v2f (&p)[8], double (&d)[8], int (&n)[8], v4f (&q)[8], float c0, float c1, int lds_addr) {
  if constexpr (OP == FMA32) EMIT(T_FMA32, a);
  if constexpr (OP == ADD32) EMIT(T_ADD32, a);
  if constexpr (OP == MUL32) EMIT(T_MUL32, a);
  if constexpr (OP == ADD32_DPP_ROWSHR) EMIT(T_DPPSHR, a);
  if constexpr (OP == ADD32_DPP_QUAD) EMIT(T_DPPQUAD, a);
  if constexpr (OP == MOV_DPP_ROWROR) EMIT(T_DPPROR, a);
  if constexpr (OP == PKADD32) EMIT(T_PKADD, p);      // (the 8th register is the common second operand: 7 accumulators + itself)
  if constexpr (OP == PKFMA32) EMIT(T_PKFMA, p);
  if constexpr (OP == PKMUL32) EMIT(T_PKMUL, p);
  if constexpr (OP == FMA64) EMIT(T_FMA64, d);
  if constexpr (OP == ADD64) EMIT(T_ADD64, d);
  if constexpr (OP == MUL64) EMIT(T_MUL64, d);
  if constexpr (OP == DOT4) EMIT(T_DOT4, n);
  if constexpr (OP == ADDU32) EMIT(T_ADDU32, n);
  if constexpr (OP == MADU24) EMIT(T_MADU24, n);
  if constexpr (OP == MED3) EMIT(T_MED3, a);
  if constexpr (OP == CVTPKU8) EMIT(T_CVTPKU8, n);
  if constexpr (OP == RCP32) EMIT(T_RCP, a);
  if constexpr (OP == EXP32) EMIT(T_EXP, a);
  if constexpr (OP == PERM) EMIT(T_PERM, n);
  if constexpr (OP == CNDMASK) EMIT(T_CNDMASK, a);
  if constexpr (OP == CMP32) EMIT(T_CMP, a);
  if constexpr (OP == DSREAD32) EMIT(T_DSR32, a);
  if constexpr (OP == DSREAD64) EMIT(T_DSR64, p);
  if constexpr (OP == DSREAD128) {
    for (int u = 0; u < 8; u++) asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8\n ds_read_b128 %2, %8\n ds_read_b128 %3, %8\n ds_read_b128 %4, %8\n ds_read_b128 %5, %8\n ds_read_b128 %6, %8\n ds_read_b128 %7, %8\n"
                 : "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3]), "=v"(q[4]), "=v"(q[5]), "=v"(q[6]), "=v"(q[7]) : "v"(lds_addr * 4));
  }
  if constexpr (OP == DSSWIZZLE) EMIT(T_DSSWZ, a);
  if constexpr (OP == DSBPERMUTE) EMIT(T_DSBPERM, a);
  if constexpr (OP == PERMLANE32SWAP) EMIT(T_PL32, a);
  if constexpr (OP == FMA32_DSREAD) EMIT(T_FMADS, a);
  if constexpr (OP == FMA32_SALU) EMIT(T_FMASALU, a);
  if constexpr (OP == NOPS) EMIT(T_NOP, a);
  if constexpr (OP == MOV32) EMIT(T_MOV, a);
  if constexpr (OP == AND32) EMIT(T_AND, n);
  if constexpr (OP == LSHL32) EMIT(T_LSHL, n);
  if constexpr (OP == XOR32) EMIT(T_XOR, n);
  if constexpr (OP == MAX32) EMIT(T_MAX, a);
  if constexpr (OP == SUB32) EMIT(T_SUB, a);
  if constexpr (OP == CNDMASK_S) EMIT(T_CNDS, a);
  if constexpr (OP == CMP_S) EMIT(T_CMPS, a);
  if constexpr (OP == BFE32) EMIT(T_BFE, n);
  if constexpr (OP == LSHLADD) EMIT(T_LSHLADD, n);
  if constexpr (OP == READLANE) EMIT(T_READLANE, a);
  if constexpr (OP == MULLO) EMIT(T_MULLO, n);
  if constexpr (OP == ADD32_SGPR) EMIT(T_ADDS, a);
  if constexpr (OP == MUL32_LIT) EMIT(T_MULLIT, a);
  if constexpr (OP == DSWRITE32) EMIT(T_DSW32, a);
  if constexpr (OP == DSWRITE128) {
    for (int u = 0; u < 8; u++) asm volatile("ds_write_b128 %8, %0\n ds_write_b128 %8, %1\n ds_write_b128 %8, %2\n ds_write_b128 %8, %3\n ds_write_b128 %8, %4\n ds_write_b128 %8, %5\n ds_write_b128 %8, %6\n ds_write_b128 %8, %7\n"
                 : : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]), "v"(lds_addr * 4) : "memory");
  }
  if constexpr (OP == DSREADU16) EMIT(T_DSRU16, n);
  if constexpr (OP == DPP_WAVESHR) EMIT(T_WAVESHR, a);
  if constexpr (OP == ADD3) EMIT(T_ADD3, n);
  if constexpr (OP == CNDMASK_E64VCC) EMIT(T_CND64, a);
  if constexpr (OP == CMP_CND_VCC) EMIT(T_CMPCNDV, a);
  if constexpr (OP == CMP_CND_S) EMIT(T_CMPCNDS, a);
  if constexpr (OP == CNDMASK_VCC_INIT) EMIT(T_CNDMASK, a);
  if constexpr (OP == DPP_FMA) EMIT(T_DPPFMA, a);
  if constexpr (OP == MAX_FMA) EMIT(T_MAXFMA, a);
  if constexpr (OP == F64_FMA) { EMIT(T_FMA64, d); EMIT(T_FMA32, a); }
  if constexpr (OP == LSHL_FMA) { EMIT(T_LSHL, n); EMIT(T_FMA32, a); }
  if constexpr (OP == CMP_CND2) EMIT(T_CMPCND2, a);
  if constexpr (OP == CMP_CND4) EMIT(T_CMPCND4, a);
  if constexpr (OP == CND_ADD) EMIT(T_CNDADD, a);
  if constexpr (OP == CMP_ADD_CND2) EMIT(T_CMPADDCND2, a);
  if constexpr (OP == DSREAD32 || OP == DSREAD64 || OP == DSREAD128 || OP == DSSWIZZLE || OP == DSBPERMUTE || OP == FMA32_DSREAD || OP == DSWRITE32 || OP == DSWRITE128 || OP == DSREADU16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int OP, int NACC>
__global__ void __launch_bounds__(1024) probe(uint64_t *times, float *sink, int iters, float c0, float c1) {
  extern __shared__ float lds[];
  float a[8];
  v2f p[8];
  double d[8];
  int n[8];
  v4f q[8];
  for (int i = 0; i < 8; i++) { a[i] = 1.0f + threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] * 0.5f}; d[i] = a[i]; n[i] = threadIdx.x + i; q[i] = v4f{0, 0, 0, 0}; }
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const int lds_addr = (threadIdx.x & 63) * 4;  // conflict-free b32
  uint64_t t0, t1;
  if constexpr (OP == CNDMASK_VCC_INIT) asm volatile("s_mov_b64 vcc, 0x5555\n s_nop 7" ::: "vcc");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_barrier\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; it++) {
    body<OP, NACC>(a, p, d, n, q, c0, c1, lds_addr);
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)d[i] + n[i] + q[i].x + q[i].w;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
  if ((threadIdx.x & 63) == 0) { times[3 * wave] = t0; times[3 * wave + 1] = t1; times[3 * wave + 2] = (uint64_t(xcc & 0xf) << 32) | hw; }
  if (s == 12345.678f) sink[0] = s;
}

struct Res { double cpi_wave, cpi_simd, ticks_per_us; int wmin, wmax; };

// one launch: per-wave ticks, and per-SIMD windows (waves grouped by the hardware id of the SIMD they ran on)
template <int OP, int NACC>
static void launch(int blocks, int threads, int ldsb, int iters, uint64_t *dT, float *dS, std::vector<uint64_t> &h, float &ms) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<OP, NACC><<<blocks, threads, ldsb>>>(dT, dS, iters, 1.0000001f, 1e-9f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), dT, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int OP, int NACC>
static Res run(int W, int B, int iters, uint64_t *dT, float *dS) {
  const int blocks = 256 * B, threads = 256 * W;        // 4*W waves per workgroup, B workgroups per CU wanted (LDS allocation)
  const int ldsb = (B == 1 ? 96 : 64) * 1024;
  const int waves = blocks * threads / 64;
  std::vector<uint64_t> h1(3 * waves), h2(3 * waves);
  hipFuncSetAttribute((const void *)probe<OP, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  float ms0, ms1, ms2;
  launch<OP, NACC>(blocks, threads, ldsb, iters / 8 + 1, dT, dS, h1, ms0);  // warm-up (clocks, code)
  launch<OP, NACC>(blocks, threads, ldsb, iters, dT, dS, h1, ms1);
  launch<OP, NACC>(blocks, threads, ldsb, 2 * iters, dT, dS, h2, ms2);
  const double per = (OP == FMA32_DSREAD || OP == FMA32_SALU || OP == CMP_CND_VCC || OP == CMP_CND_S || OP == CND_ADD || OP == DPP_FMA || OP == F64_FMA || OP == LSHL_FMA || OP == MAX_FMA) ? 2 : (OP == CMP_CND2 ? 3 : ((OP == CMP_CND4 || OP == CMP_ADD_CND2) ? 5 : 1)), ninst = double(iters) * UNROLL * per;
  // (1) slope of the mean wave time between the two launches = ticks for `ninst` more instructions per wave
  double s1 = 0, s2 = 0;
  for (int w = 0; w < waves; w++) { s1 += double(h1[3 * w + 1] - h1[3 * w]); s2 += double(h2[3 * w + 1] - h2[3 * w]); }
  // (2) per SIMD: how many waves ran on it (hardware ids), and the window from its first start to its last end
  struct Acc { uint64_t lo = ~0ull, hi = 0; int n = 0; };
  std::map<uint64_t, Acc> simd;
  for (int w = 0; w < waves; w++) {
    const uint64_t id = h2[3 * w + 2], hw = id & 0xffffffffu;
    // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; + xcc in bits 35:32
    const uint64_t key = (id >> 32 << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 11) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3);
    Acc &a = simd[key];
    a.lo = std::min(a.lo, h2[3 * w]); a.hi = std::max(a.hi, h2[3 * w + 1]); a.n++;
  }
  Res r;
  r.wmin = 1 << 30; r.wmax = 0;
  double win = 0, cnt = 0;
  for (auto &kv : simd) { r.wmin = std::min(r.wmin, kv.second.n); r.wmax = std::max(r.wmax, kv.second.n); win += double(kv.second.hi - kv.second.lo); cnt += kv.second.n; }
  r.cpi_wave = (s2 - s1) / waves / (ninst * W * B);
  r.cpi_simd = win / (cnt * 2 * ninst);                // sum of SIMD windows / instructions issued on those SIMDs
  r.ticks_per_us = (s2 - s1) / waves / ((ms2 - ms1) * 1e3);
  return r;
}

template <int OP>
static void row(uint64_t *dT, float *dS, int iters) {
  printf("%-30s", OPN[OP]);
  const int cfg[4][2] = {{1, 1}, {2, 1}, {4, 1}, {4, 2}};
  double tpu = 0; int wmin = 0, wmax = 0;
  for (auto &c : cfg) { Res r = run<OP, 8>(c[0], c[1], iters, dT, dS); printf(" %6.2f/%-6.2f", r.cpi_wave, r.cpi_simd); tpu = r.ticks_per_us; wmin = r.wmin; wmax = r.wmax; }
  Res d = run<OP, 1>(1, 1, iters, dT, dS);
  printf(" | dep. chain %6.2f | 8w: %d..%d waves/SIMD seen, %.0f ticks/us\n", d.cpi_wave, wmin, wmax, tpu);
}

int main(int argc, char **argv) {
  uint64_t *dT; float *dS;
  hipMalloc(&dT, sizeof(uint64_t) * 3 * 256 * 32);
  hipMalloc(&dS, 64);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("# valu_issue: %s, %d CUs, clockRate %.2f GHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate * 1e-6);
  printf("# cycles (s_memtime ticks) per wave64 instruction per SIMD = wave time / (waves on the SIMD x instructions per wave); 8 independent registers\n");
  printf("# two figures per column: A/B.  A = slope of the mean wave time between launches of N and 2N iterations / (waves per SIMD x instructions);\n"
         "# B = sum over SIMDs (grouped by hardware id) of [last end - first start] / instructions issued on them.  ticks/us: s_memtime slope / wall-time slope.\n");
  printf("%-30s %13s %13s %13s %13s\n", "# instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "8 (2 wg/CU)");
  const int it = 1000;
  if (argc > 1 && !strcmp(argv[1], "cnd")) {  // only the select forms
    row<CNDMASK>(dT, dS, it); row<CNDMASK_VCC_INIT>(dT, dS, it); row<CNDMASK_E64VCC>(dT, dS, it); row<CNDMASK_S>(dT, dS, it); row<CMP_CND_VCC>(dT, dS, it); row<CMP_CND_S>(dT, dS, it);
    row<CMP_CND2>(dT, dS, it); row<CMP_CND4>(dT, dS, it); row<CND_ADD>(dT, dS, it); row<CMP_ADD_CND2>(dT, dS, it);
    row<DPP_FMA>(dT, dS, it); row<MAX_FMA>(dT, dS, it); row<F64_FMA>(dT, dS, it); row<LSHL_FMA>(dT, dS, it);
    return 0;
  }
  row<FMA32>(dT, dS, it); row<ADD32>(dT, dS, it); row<MUL32>(dT, dS, it); row<ADD32_DPP_ROWSHR>(dT, dS, it); row<ADD32_DPP_QUAD>(dT, dS, it); row<MOV_DPP_ROWROR>(dT, dS, it);
  row<PKADD32>(dT, dS, it); row<PKFMA32>(dT, dS, it); row<PKMUL32>(dT, dS, it); row<FMA64>(dT, dS, it); row<ADD64>(dT, dS, it); row<MUL64>(dT, dS, it);
  row<DOT4>(dT, dS, it); row<ADDU32>(dT, dS, it); row<MADU24>(dT, dS, it); row<MED3>(dT, dS, it); row<CVTPKU8>(dT, dS, it); row<RCP32>(dT, dS, it); row<EXP32>(dT, dS, it);
  row<PERM>(dT, dS, it); row<CNDMASK>(dT, dS, it); row<CMP32>(dT, dS, it);
  row<DSREAD32>(dT, dS, it); row<DSREAD64>(dT, dS, it); row<DSREAD128>(dT, dS, it); row<DSSWIZZLE>(dT, dS, it); row<DSBPERMUTE>(dT, dS, it); row<PERMLANE32SWAP>(dT, dS, it);
  row<FMA32_DSREAD>(dT, dS, it); row<FMA32_SALU>(dT, dS, it); row<NOPS>(dT, dS, it);
  row<MOV32>(dT, dS, it); row<AND32>(dT, dS, it); row<LSHL32>(dT, dS, it); row<XOR32>(dT, dS, it); row<MAX32>(dT, dS, it); row<SUB32>(dT, dS, it); row<CNDMASK_S>(dT, dS, it);
  row<CMP_S>(dT, dS, it); row<BFE32>(dT, dS, it); row<LSHLADD>(dT, dS, it); row<ADD3>(dT, dS, it); row<READLANE>(dT, dS, it); row<MULLO>(dT, dS, it); row<ADD32_SGPR>(dT, dS, it);
  row<MUL32_LIT>(dT, dS, it); row<DPP_WAVESHR>(dT, dS, it); row<DSWRITE32>(dT, dS, it); row<DSWRITE128>(dT, dS, it); row<DSREADU16>(dT, dS, it);
  row<CNDMASK_VCC_INIT>(dT, dS, it); row<CNDMASK_E64VCC>(dT, dS, it); row<CMP_CND_VCC>(dT, dS, it); row<CMP_CND_S>(dT, dS, it);
  row<CMP_CND2>(dT, dS, it); row<CMP_CND4>(dT, dS, it); row<CND_ADD>(dT, dS, it); row<CMP_ADD_CND2>(dT, dS, it);
  row<DPP_FMA>(dT, dS, it); row<MAX_FMA>(dT, dS, it); row<F64_FMA>(dT, dS, it); row<LSHL_FMA>(dT, dS, it);
  return 0;
}
