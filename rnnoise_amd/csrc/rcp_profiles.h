// rcp_profiles.h -- the host-specific part of the reference's arithmetic: `rcpps`.
//
// The reference's tanh / sigmoid (src/vec_avx.h:413,442; SSE twins :484,505) divide through _mm256_rcp_ps / _mm_rcp_ps, an
// approximate reciprocal whose low bits differ between CPU families, so "the reference's output" is a function of the
// host it runs on.  On every host measured (oracle/rcp_capture.c --analyze) the instruction is a pure function of the top
// 12 mantissa bits (11 on Intel), exponent-invariant, low 11 result bits zero: 4096 16-bit entries describe it exactly.
//
// Profiles:  "intel"     captured on the Intel Xeon build host (the committed goldens were produced there)
//            "amd-zen5"  captured on the AMD EPYC 9575F hosts of the MI355X boxes
//            "host"      captured at load time from the CPU the library runs on (tables.cpp: rcp_select_locked) -- the default:
//                        a process that swaps librnnoise.so.0 gets the bits the reference produced on that same machine.
// Both the product (tables.cpp) and the oracle (oracle/rn_oracle.c) include this header; neither includes the other.
#pragma once
#include "rcp_profile_intel.h"
#include "rcp_profile_amd_zen5.h"

#define RN_RCP_ENTRIES 4096
// bits(rcp(x)) of a positive normal x from a profile table (the device twin is rn_dev.h: rn_rcp_x86)
static inline unsigned int rn_rcp_bits_from(const unsigned short *t, unsigned int b) {
  return ((unsigned int)t[(b >> 11) & 0xfff] << 11) + (0x7e800000u - (b & 0x7f800000u));
}

#if defined(__x86_64__) || defined(__i386__)
#include <xmmintrin.h>
// This CPU's rcpps as a profile table.  Returns 0 when the 4096 captured entries have the 16-bit form AND reproduce the
// instruction on a strided sample of ~1.3 million other inputs (all binades whose reciprocal is normal; the exhaustive
// proof is oracle/rcp_capture.c's job, 4 s per host), -1 otherwise.
static inline int rn_rcp_capture_host(unsigned short *t) {
  union { float f; unsigned int u; } x, y;
  for (int i = 0; i < RN_RCP_ENTRIES; i++) {
    x.u = 0x3f800000u | ((unsigned int)i << 11);
    y.f = _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(x.f)));
    if ((y.u & 0x7ff) || y.u < 0x3f000000u || y.u > 0x3f800000u) return -1;
    t[i] = (unsigned short)((y.u - 0x3f000000u) >> 11);
  }
  for (unsigned int e = 2; e <= 252; e += 5)
    for (unsigned int m = e; m < (1u << 23); m += 331) {
      x.u = (e << 23) | m;
      y.f = _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(x.f)));
      if (y.u != rn_rcp_bits_from(t, x.u)) return -1;
    }
  return 0;
}
#else
static inline int rn_rcp_capture_host(unsigned short *t) { (void)t; return -1; }
#endif
