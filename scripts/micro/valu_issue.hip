// VALU issue-rate micro-benchmark for gfx950: cycles per wave64 instruction per SIMD for the integer
// instructions the scan kernel is made of, at 1 / 2 / 4 wavefronts per SIMD (the scan kernel runs 4).
// Every wavefront executes ITER x 64 instructions of one kind on eight independent registers (no
// dependent chain shorter than eight instructions), so the figure is the issue rate, not the latency.
// Printed: cycles per wave-instruction per SIMD = kernel time x shader clock x SIMDs / (waves x instructions).
// The shader clock is measured in the same launch (s_memtime ticks of one wavefront over the kernel's wall time
// are the constant 100 MHz reference, so the clock is taken from v_fma_f32 = 2 cycles as a cross-check only;
// the primary figure uses hipDeviceProp.clockRate and is printed beside the ratio to v_fma_f32).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP64(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)

enum {
    K_FMA_F32, K_ADD_U32, K_ADD3_U32, K_MAD_U24, K_MUL_U24, K_MUL_LO, K_CMP_SAND, K_CMP_ONLY, K_ADDC, K_DOT2, K_DOT2_I16,
    K_ADD_SDWA, K_PK_ADD_U16, K_PK_SUB_CLAMP, K_ALIGNBIT, K_PERM, K_BFE, K_LSHL_ADD, K_AND_OR, K_CNDMASK, K_SQRT, K_READLANE, K_WRITELANE,
    K_DPP_ADD, K_LSHRREV, K_XOR, K_MBCNT, K_SAD_U16, K_MAX3, K_PK_MAX_U16, K_PK_MAD_U16, K_SUB_U32, K_AND, K_AND_LIT, K_OR, K_LSHLREV, K_ASHRREV, K_MIN_U32, K_MAX_I32, K_MOV, K_ADD_CO, K_ADDC_E32, K_CNDMASK_E32, K_CMP_E32, K_CMP_F32_E32, K_CMP_I32_E32, K_CMP_U16_E32, K_CMP_ADDC_E32, K_ADD_F32, K_SUB_F32, K_MUL_F32, K_FMAC_F32, K_MAX_F32, K_MED3_F32, K_CVT_F32_U32, K_CVT_U32_F32, K_CVT_F32_UB1, K_CVT_F32_F16, K_PK_FMA_F32, K_PK_ADD_F32, K_ADD_E64, K_ADD_SGPR, K_XOR_E64, K_FMA_F32_SG, K_MAD_I32_I24, K_ADD_LSHL, K_OR3, K_BFI, K_BCNT, K_MUL_HI, K_MAD_U64, K_FMA_F16, K_ADD_F16, K_ADD_U16, K_SUB_U16, K_MAD_U16, K_LSHL_OR, K_SUBB, K_FFBH, K_SUBREV_SDWA, K_NKINDS
};
static const char *const NAMES[K_NKINDS] = {
    "v_fma_f32", "v_add_u32", "v_add3_u32", "v_mad_u32_u24", "v_mul_u32_u24", "v_mul_lo_u32", "v_cmp_gt_u32 + s_and_b64", "v_cmp_gt_u32 (sgpr dst)",
    "v_addc_co_u32 (sgpr carry-in)", "v_dot2_u32_u16", "v_dot2_i32_i16", "v_add_u32 sdwa WORD_1", "v_pk_add_u16", "v_pk_sub_u16 clamp", "v_alignbit_b32", "v_perm_b32", "v_bfe_u32",
    "v_lshl_add_u32", "v_and_or_b32", "v_cndmask_b32 (sgpr mask)", "v_sqrt_f32", "v_readlane_b32", "v_writelane_b32", "v_add_u32 dpp row_shr:1",
    "v_lshrrev_b32", "v_xor_b32", "v_mbcnt_lo_u32_b32", "v_sad_u16", "v_max3_u32", "v_pk_max_u16", "v_pk_mad_u16", "v_sub_u32", "v_and_b32", "v_and_b32 literal 0xffff", "v_or_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_min_u32", "v_max_i32", "v_mov_b32", "v_add_co_u32 e32 (vcc out)", "v_addc_co_u32 e32 (vcc in/out)", "v_cndmask_b32 e32 (vcc)", "v_cmp_gt_u32 e32 (vcc)", "v_cmp_ge_f32 e32 (vcc)", "v_cmp_ge_i32 e32 (vcc)", "v_cmp_gt_u16 e32 (vcc)", "v_cmp_gt_u32 vcc + v_addc e32 (2 insts)", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fmac_f32", "v_max_f32", "v_med3_f32", "v_cvt_f32_u32", "v_cvt_u32_f32", "v_cvt_f32_ubyte1", "v_cvt_f32_f16", "v_pk_fma_f32 (2 values)", "v_pk_add_f32 (2 values)", "v_add_u32 e64", "v_add_u32 sgpr operand", "v_xor_b32 e64", "v_fma_f32 sgpr multiplier", "v_mad_i32_i24", "v_add_lshl_u32", "v_or3_b32", "v_bfi_b32", "v_bcnt_u32_b32", "v_mul_hi_u32", "v_mad_u64_u32", "v_pk_fma_f16", "v_add_f16", "v_add_u16", "v_sub_u16", "v_mad_u16", "v_lshl_or_b32", "v_subb_co_u32 e32", "v_ffbh_u32", "v_sub_u32 sdwa W0-W1"};

template <int KIND>
__global__ void __launch_bounds__(256) bench(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t r0 = threadIdx.x * 3u + seed, r1 = r0 ^ 0x1234u, r2 = r0 + 77u, r3 = r0 * 5u, r4 = r0 + 9u, r5 = r0 ^ 0xffu, r6 = r0 + 1u, r7 = r0 + 2u;
    uint32_t a = threadIdx.x + 3u, b = seed | 1u;
    uint64_t m = 0x5555aaaa3333ccccull ^ seed, m2 = 0;
    uint32_t s = seed;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define R(k) "%" #k
        if (KIND == K_FMA_F32) {
#define S(k) "v_fma_f32 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_ADD_U32) {
#define S(k) "v_add_u32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_ADD3_U32) {
#define S(k) "v_add3_u32 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MAD_U24) {
#define S(k) "v_mad_u32_u24 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MUL_U24) {
#define S(k) "v_mul_u32_u24 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MUL_LO) {
#define S(k) "v_mul_lo_u32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_CMP_SAND) {
            /* counted as ONE vector instruction each: the s_and goes to the scalar unit */
#define S(k) "v_cmp_gt_u32 %8, " R(k) ", %10\n s_and_b64 %9, %9, %8\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+s"(m), "+s"(m2) : "v"(a) : "scc");
#undef S
        } else if (KIND == K_CMP_ONLY) {
#define S(k) "v_cmp_gt_u32 %8, " R(k) ", %9\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+s"(m) : "v"(a));
#undef S
        } else if (KIND == K_ADDC) {
#define S(k) "v_addc_co_u32 " R(k) ", vcc, " R(k) ", " R(k) ", %8\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(m) : "vcc");
#undef S
        } else if (KIND == K_DOT2) {
#define S(k) "v_dot2_u32_u16 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_DOT2_I16) {
#define S(k) "v_dot2_i32_i16 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_ADD_SDWA) {
#define S(k) "v_add_u32_sdwa " R(k) ", %8, " R(k) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_PK_ADD_U16) {
#define S(k) "v_pk_add_u16 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_PK_SUB_CLAMP) {
#define S(k) "v_pk_sub_u16 " R(k) ", %8, " R(k) " clamp\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_ALIGNBIT) {
#define S(k) "v_alignbit_b32 " R(k) ", %8, " R(k) ", 16\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_PERM) {
#define S(k) "v_perm_b32 " R(k) ", %8, " R(k) ", %9\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_BFE) {
#define S(k) "v_bfe_u32 " R(k) ", " R(k) ", 3, 17\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_LSHL_ADD) {
#define S(k) "v_lshl_add_u32 " R(k) ", " R(k) ", 1, %8\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_AND_OR) {
#define S(k) "v_and_or_b32 " R(k) ", " R(k) ", %8, %9\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_CNDMASK) {
#define S(k) "v_cndmask_b32 " R(k) ", " R(k) ", %8, %9\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "s"(m));
#undef S
        } else if (KIND == K_SQRT) {
#define S(k) "v_sqrt_f32 " R(k) ", " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_READLANE) {
#define S(k) "v_readlane_b32 %8, " R(k) ", 5\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+s"(s) : "v"(a));
#undef S
        } else if (KIND == K_WRITELANE) {
#define S(k) "v_writelane_b32 " R(k) ", %8, 7\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(s));
#undef S
        } else if (KIND == K_DPP_ADD) {
#define S(k) "v_add_u32_dpp " R(k) ", %8, " R(k) " row_shr:1 row_mask:0xf bank_mask:0xf\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_LSHRREV) {
#define S(k) "v_lshrrev_b32 " R(k) ", 1, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_XOR) {
#define S(k) "v_xor_b32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MBCNT) {
#define S(k) "v_mbcnt_lo_u32_b32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_SAD_U16) {
#define S(k) "v_sad_u16 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MAX3) {
#define S(k) "v_max3_u32 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_PK_MAX_U16) {
#define S(k) "v_pk_max_u16 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_PK_MAD_U16) {
#define S(k) "v_pk_mad_u16 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_SUB_U32) {
#define S(k) "v_sub_u32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_AND) {
#define S(k) "v_and_b32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_AND_LIT) {
#define S(k) "v_and_b32 " R(k) ", 0xffff, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_OR) {
#define S(k) "v_or_b32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_LSHLREV) {
#define S(k) "v_lshlrev_b32 " R(k) ", 1, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_ASHRREV) {
#define S(k) "v_ashrrev_i32 " R(k) ", 1, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MIN_U32) {
#define S(k) "v_min_u32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MAX_I32) {
#define S(k) "v_max_i32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MOV) {
#define S(k) "v_mov_b32 " R(k) ", %8\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_ADD_CO) {
#define S(k) "v_add_co_u32 " R(k) ", vcc, %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
#undef S
        } else if (KIND == K_ADDC_E32) {
#define S(k) "v_addc_co_u32 " R(k) ", vcc, %8, " R(k) ", vcc\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
#undef S
        } else if (KIND == K_CNDMASK_E32) {
#define S(k) "v_cndmask_b32 " R(k) ", %8, " R(k) ", vcc\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
#undef S
        } else if (KIND == K_CMP_E32) {
#define S(k) "v_cmp_gt_u32 vcc, %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
#undef S
        } else if (KIND == K_CMP_F32_E32) {
#define S(k) "v_cmp_ge_f32 vcc, %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
#undef S
        } else if (KIND == K_CMP_I32_E32) {
#define S(k) "v_cmp_ge_i32 vcc, %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
#undef S
        } else if (KIND == K_CMP_U16_E32) {
#define S(k) "v_cmp_gt_u16 vcc, %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
#undef S
        } else if (KIND == K_CMP_ADDC_E32) {
#define S(k) "v_cmp_gt_u32 vcc, %8, " R(k) "\n v_addc_co_u32 " R(k) ", vcc, " R(k) ", " R(k) ", vcc\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
#undef S
        } else if (KIND == K_ADD_F32) {
#define S(k) "v_add_f32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_SUB_F32) {
#define S(k) "v_sub_f32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MUL_F32) {
#define S(k) "v_mul_f32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_FMAC_F32) {
#define S(k) "v_fmac_f32 " R(k) ", %8, %9\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MAX_F32) {
#define S(k) "v_max_f32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MED3_F32) {
#define S(k) "v_med3_f32 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_CVT_F32_U32) {
#define S(k) "v_cvt_f32_u32 " R(k) ", " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_CVT_U32_F32) {
#define S(k) "v_cvt_u32_f32 " R(k) ", " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_CVT_F32_UB1) {
#define S(k) "v_cvt_f32_ubyte1 " R(k) ", " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_CVT_F32_F16) {
#define S(k) "v_cvt_f32_f16 " R(k) ", " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_ADD_E64) {
#define S(k) "v_add_u32_e64 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_ADD_SGPR) {
#define S(k) "v_add_u32 " R(k) ", %10, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "s"(s));
#undef S
        } else if (KIND == K_XOR_E64) {
#define S(k) "v_xor_b32_e64 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_FMA_F32_SG) {
#define S(k) "v_fma_f32 " R(k) ", %8, %10, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "s"(s));
#undef S
        } else if (KIND == K_MAD_I32_I24) {
#define S(k) "v_mad_i32_i24 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_ADD_LSHL) {
#define S(k) "v_add_lshl_u32 " R(k) ", " R(k) ", %8, 1\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_OR3) {
#define S(k) "v_or3_b32 " R(k) ", " R(k) ", %8, %9\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_BFI) {
#define S(k) "v_bfi_b32 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_BCNT) {
#define S(k) "v_bcnt_u32_b32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MUL_HI) {
#define S(k) "v_mul_hi_u32 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_FMA_F16) {
#define S(k) "v_pk_fma_f16 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_ADD_F16) {
#define S(k) "v_add_f16 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_ADD_U16) {
#define S(k) "v_add_u16 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_SUB_U16) {
#define S(k) "v_sub_u16 " R(k) ", %8, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_MAD_U16) {
#define S(k) "v_mad_u16 " R(k) ", %8, %9, " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_LSHL_OR) {
#define S(k) "v_lshl_or_b32 " R(k) ", " R(k) ", 16, %8\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_SUBB) {
#define S(k) "v_subb_co_u32 " R(k) ", vcc, %8, " R(k) ", vcc\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
#undef S
        } else if (KIND == K_FFBH) {
#define S(k) "v_ffbh_u32 " R(k) ", " R(k) "\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_SUBREV_SDWA) {
#define S(k) "v_sub_u32_sdwa " R(k) ", %8, " R(k) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
            asm volatile(REP64(S) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef S
        } else if (KIND == K_PK_FMA_F32 || KIND == K_PK_ADD_F32 || KIND == K_MAD_U64) {
            static_assert(sizeof(uint64_t) == 8, "");
            uint64_t q0 = r0 | ((uint64_t)r1 << 32), q1 = r2 | ((uint64_t)r3 << 32), q2 = r4 | ((uint64_t)r5 << 32), q3 = r6 | ((uint64_t)r7 << 32);
            uint64_t q4 = q0 + 1, q5 = q1 + 1, q6 = q2 + 1, q7 = q3 + 1, pa = a | ((uint64_t)b << 32);
            for (int j = 0; j < 1; ++j) {
                if (KIND == K_PK_FMA_F32) {
#define S(k) "v_pk_fma_f32 " R(k) ", %8, %8, " R(k) "\n"
                    asm volatile(REP64(S) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(pa));
#undef S
                } else if (KIND == K_PK_ADD_F32) {
#define S(k) "v_pk_add_f32 " R(k) ", %8, " R(k) "\n"
                    asm volatile(REP64(S) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(pa));
#undef S
                } else {
#define S(k) "v_mad_u64_u32 " R(k) ", vcc, %9, %10, " R(k) "\n"
                    asm volatile(REP64(S) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(pa), "v"(a), "v"(b) : "vcc");
#undef S
                }
            }
            r0 ^= (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) >> 32);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (uint32_t)m ^ (uint32_t)m2 ^ s;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        reinterpret_cast<unsigned long long *>(out + (1 << 22))[0] = t1 - t0;
}

static int g_cus;
static double g_khz;

template <int KIND>
int run(uint32_t *d_out, FILE *csv)
{
    const int iters = 400;
    printf("%-32s", NAMES[KIND]);
    for (int wps = 1; wps <= 8; wps *= 2) { /* wavefronts per SIMD: one 256-thread workgroup = one per SIMD */
        const int grid = g_cus * wps;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        bench<KIND><<<grid, 256>>>(d_out, 10, 1);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        unsigned long long ticks = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            bench<KIND><<<grid, 256>>>(d_out, iters, 1);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) {
                best = ms;
                CK(hipMemcpy(&ticks, d_out + (1 << 22), 8, hipMemcpyDeviceToHost));
            }
        }
        const double insts = (double)iters * 64.0 * wps; /* per SIMD */
        const double cyc = best * 1e-3 * g_khz * 1e3 / insts;
        const double cyc_tick = (double)ticks / ((double)iters * 64.0) / wps; /* s_memtime ticks of wavefront 0 per instruction-slot */
        printf("  %d w/SIMD: %5.2f cyc (%6.3f ms; %5.2f ticks)", wps, cyc, best, cyc_tick);
        if (csv)
            fprintf(csv, "%s,%d,%.3f,%.4f,%.3f\n", NAMES[KIND], wps, cyc, best, cyc_tick);
        CK(hipEventDestroy(e0));
        CK(hipEventDestroy(e1));
    }
    printf("\n");
    return 0;
}

template <int K>
struct RunAll {
    static int go(uint32_t *d, FILE *csv) { return run<K>(d, csv) || RunAll<K + 1>::go(d, csv); }
};
template <>
struct RunAll<K_NKINDS> {
    static int go(uint32_t *, FILE *) { return 0; }
};

int main(int argc, char **argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    g_khz = prop.clockRate;
    printf("%s: %d CUs, clockRate %.0f MHz (cycles below = time x clockRate; 'ticks' = s_memtime of one wavefront)\n", prop.gcnArchName, g_cus, g_khz / 1e3);
    printf("cycles per wave64 instruction per SIMD, 25600 instructions per wavefront, eight independent registers\n");
    uint32_t *d_out;
    CK(hipMalloc(&d_out, (size_t)((1 << 22) + 16) * 4));
    FILE *csv = argc > 1 ? fopen(argv[1], "w") : NULL;
    if (csv)
        fprintf(csv, "instruction,waves_per_simd,cycles_per_inst_per_simd,ms,ticks_per_slot\n");
    const int rc = RunAll<0>::go(d_out, csv);
    if (csv)
        fclose(csv);
    return rc;
}
