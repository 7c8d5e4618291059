// hp_lean.hip -- hp_kernel.hip's lane = stream kernel once more with 16-sample blocks and a register budget of 64: a wave of it fits
// a SIMD BESIDE four waves of rn_analysis_kernel (4 x 112 + 64 = 512 VGPRs).  The standard build (107 VGPRs) takes the place of an
// analysis wave on every SIMD it runs on -- and, being 1,024 waves, that is every SIMD of the machine: while it runs beside the
// analysis kernel of another frame (the pipelined schedule) each CU holds three analysis workgroups instead of four.
#define RN_HP_KERNEL_NAME rn_hp_lean_kernel
#define RN_HP_VARIANT_ONLY 1
#define RN_HP_BLK 4
#define RN_HP_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#include "hp_kernel.hip"
