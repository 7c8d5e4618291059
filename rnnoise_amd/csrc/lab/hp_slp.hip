// hp_slp.hip -- INSTRUMENTED BUILD ONLY: hp_kernel.hip's lane = stream kernel compiled WITH the SLP vectoriser, as it shipped in
// rounds 1-4.  The vectoriser turns the kernel's autocorrelation chains into v_pk_mul_f32 / v_pk_add_f32, 60 of them with an
// op_sel bit -- and on gfx950 such an instruction takes the wrong half of its operand in lanes 48..63 while another wave of the
// SIMD issues v_mfma_i32_16x16x64_i8 (profiles/r5_gru_race.txt, tools/pk_coissue_probe.hip).  Kept so that the A/B that found it
// stays reproducible: $RNNOISE_AMD_HP_AB=2048 on librnnoise_amd_instr.so (tools/gru_race.py).  The product library never has it.
#define RN_HP_KERNEL_NAME rn_hp_slp_kernel
#define RN_HP_VARIANT_ONLY 1
#include "../hp_kernel.hip"
