// VALU issue rates of the instructions the strip DP cell is made of (gfx950), 4 waves per SIMD, chain-free streams over 8 registers.
//   hipcc -O3 --offload-arch=gfx950 scripts/valu_microbench2.hip -o /tmp/valu2 && /tmp/valu2
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP 4096
#define OP8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define KERNEL(NAME, TEXT)                                                                                                     \
    __global__ __launch_bounds__(64) void NAME(int *out, int seed) {                                                           \
        int a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        const int b = seed | 0x00030001, c = 0x05040100;                                                                       \
        for (int i = 0; i < REP; ++i)                                                                                          \
            asm volatile(TEXT : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "v40", "v41", "v42", "v43", "v44", "v45", "s20", "s21"); \
        out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                            \
    }
#define T8(F) F("0") F("1") F("2") F("3") F("4") F("5") F("6") F("7")
#define I_ADD(r) "v_add_u32 %" r ", %" r ", %8\n"
#define I_SUB(r) "v_sub_u32 %" r ", %" r ", %8\n"
#define I_AND(r) "v_and_b32 %" r ", %" r ", %8\n"
#define I_LSHLOR(r) "v_lshl_or_b32 %" r ", %" r ", 1, %8\n"
#define I_ANDOR(r) "v_and_or_b32 %" r ", %" r ", 7, %8\n"
#define I_PERM(r) "v_perm_b32 %" r ", %" r ", %8, %9\n"
#define I_PKADD(r) "v_pk_add_u16 %" r ", %" r ", %8\n"
#define I_PKSUBSAT(r) "v_pk_sub_u16 %" r ", %" r ", %8 clamp\n"
#define I_PKMINU(r) "v_pk_min_u16 %" r ", %" r ", %8\n"
#define I_PKMAXI(r) "v_pk_max_i16 %" r ", %" r ", %8\n"
#define I_PKMINSEL(r) "v_pk_min_i16 %" r ", %" r ", %8 op_sel_hi:[0,1]\n"
#define I_MAX3(r) "v_max3_i16 %" r ", %" r ", %" r ", %8 op_sel:[0,1,0,0]\n"
#define I_DPP(r) "v_mov_b32_dpp %" r ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_CNDMASK(r) "v_cndmask_b32 %" r ", %" r ", %8, vcc\n"
#define I_MOV64(r) "v_lshl_add_u32 %" r ", %" r ", 2, %8\n"
#define I_MAD16(r) "v_mad_u32_u16 %" r ", %" r ", %8, %9\n"
#define I_BITOP(r) "v_bfi_b32 %" r ", %" r ", %8, %9\n"
#define I_OR(r) "v_or_b32 %" r ", %" r ", %8\n"
#define I_XOR(r) "v_xor_b32 %" r ", %" r ", %8\n"
#define I_LSHL(r) "v_lshlrev_b32 %" r ", 1, %" r "\n"
#define I_LSHR(r) "v_lshrrev_b32 %" r ", 1, %" r "\n"
#define I_MINU(r) "v_min_u32 %" r ", %" r ", %8\n"
#define I_MAXI(r) "v_max_i32 %" r ", %" r ", %8\n"
#define I_MAX3P(r) "v_max3_i16 %" r ", %" r ", %8, %9\n"
#define I_MAX3I32(r) "v_max3_i32 %" r ", %" r ", %8, %9\n"
#define I_MAXI16(r) "v_max_i16 %" r ", %" r ", %8\n"
#define I_PKMAXSEL(r) "v_pk_max_i16 %" r ", %" r ", %" r " op_sel:[0,1] op_sel_hi:[1,0]\n"
#define I_MOV(r) "v_mov_b32 %" r ", %8\n"
#define I_ADD3(r) "v_add3_u32 %" r ", %" r ", %8, %9\n"
#define I_MAD24(r) "v_mad_u32_u24 %" r ", %" r ", %8, %9\n"
#define I_MULLO(r) "v_mul_lo_u32 %" r ", %" r ", %8\n"
#define I_MULHI(r) "v_mul_hi_u32 %" r ", %" r ", %8\n"
#define I_MUL24(r) "v_mul_u32_u24 %" r ", %" r ", %8\n"
#define I_BFE(r) "v_bfe_u32 %" r ", %" r ", 3, 5\n"
#define I_BCNT(r) "v_bcnt_u32_b32 %" r ", %" r ", %8\n"
#define I_OR3(r) "v_or3_b32 %" r ", %" r ", %8, %9\n"
#define I_SUBSDWA(r) "v_add_u32_sdwa %" r ", %" r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
#define I_PKLSHL(r) "v_pk_lshlrev_b16 %" r ", 1, %" r "\n"
KERNEL(k_add, T8(I_ADD)) KERNEL(k_sub, T8(I_SUB)) KERNEL(k_and, T8(I_AND)) KERNEL(k_lshlor, T8(I_LSHLOR)) KERNEL(k_andor, T8(I_ANDOR))
KERNEL(k_perm, T8(I_PERM)) KERNEL(k_pkadd, T8(I_PKADD)) KERNEL(k_pksubsat, T8(I_PKSUBSAT)) KERNEL(k_pkminu, T8(I_PKMINU))
KERNEL(k_pkmaxi, T8(I_PKMAXI)) KERNEL(k_pkminsel, T8(I_PKMINSEL)) KERNEL(k_max3, T8(I_MAX3)) KERNEL(k_dpp, T8(I_DPP))
KERNEL(k_cnd, "v_cmp_gt_u32 vcc, %8, %9\n" T8(I_CNDMASK)) KERNEL(k_cnd2, "s_mov_b64 vcc, 0x5555\n" T8(I_CNDMASK))
#define I_CMP(r) "v_cmp_gt_u32 vcc, %" r ", %8\n"
#define I_CMPS(r) "v_cmp_gt_u32 s[20:21], %" r ", %8\n"
KERNEL(k_cmp, T8(I_CMP))
#define I_MINI16(r) "v_min_i16 %" r ", %" r ", %8\n"
#define I_MINU16(r) "v_min_u16 %" r ", %" r ", %8\n"
#define I_ADDU16(r) "v_add_u16 %" r ", %" r ", %8\n"
#define I_MOV64B(r) "v_mov_b64 v[40:41], v[42:43]\n"
#define I_MAD64(r) "v_mad_u64_u32 v[40:41], s[20:21], %" r ", %8, 0\n"
#define I_LSHLADD64(r) "v_lshl_add_u64 v[40:41], v[42:43], 3, v[44:45]\n"
#define I_LSHR64(r) "v_lshrrev_b64 v[40:41], 5, v[42:43]\n"
#define I_MOV64X(r) "v_mov_b64 v[40:41], v[42:43]\n"
#define I_CVT(r) "v_cvt_f64_i32 v[40:41], %" r "\n"
#define I_MULF64(r) "v_mul_f64 v[40:41], v[42:43], v[44:45]\n"
KERNEL(k_mad64, T8(I_MAD64)) KERNEL(k_lshladd64, T8(I_LSHLADD64)) KERNEL(k_lshr64, T8(I_LSHR64)) KERNEL(k_mov64, T8(I_MOV64X)) KERNEL(k_cvt, T8(I_CVT)) KERNEL(k_mulf64, T8(I_MULF64))
KERNEL(k_mini16, T8(I_MINI16)) KERNEL(k_minu16, T8(I_MINU16)) KERNEL(k_addu16, T8(I_ADDU16)) KERNEL(k_lshladd, T8(I_MOV64)) KERNEL(k_mad16, T8(I_MAD16)) KERNEL(k_bfi, T8(I_BITOP))
KERNEL(k_or, T8(I_OR)) KERNEL(k_xor, T8(I_XOR)) KERNEL(k_lshl, T8(I_LSHL)) KERNEL(k_lshr, T8(I_LSHR)) KERNEL(k_minu, T8(I_MINU)) KERNEL(k_maxi, T8(I_MAXI))
KERNEL(k_max3p, T8(I_MAX3P)) KERNEL(k_max3i32, T8(I_MAX3I32)) KERNEL(k_maxi16, T8(I_MAXI16)) KERNEL(k_pkmaxsel, T8(I_PKMAXSEL)) KERNEL(k_mov, T8(I_MOV))
KERNEL(k_add3, T8(I_ADD3)) KERNEL(k_mad24, T8(I_MAD24)) KERNEL(k_mullo, T8(I_MULLO)) KERNEL(k_mulhi, T8(I_MULHI)) KERNEL(k_mul24, T8(I_MUL24))
KERNEL(k_bfe, T8(I_BFE)) KERNEL(k_bcnt, T8(I_BCNT)) KERNEL(k_or3, T8(I_OR3)) KERNEL(k_sdwa, T8(I_SUBSDWA)) KERNEL(k_pklshl, T8(I_PKLSHL))

static void run(const char *name, void (*kern)(int *, int), int *d_out) {
    const int w = 4, blocks = 256 * 4 * w;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_out, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_out, r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double rate = 4.0 * blocks * (double)REP * 8.0 / (ms * 1e-3);
    printf("%-28s %8.1f G wave-instr/s  -> %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, rate / 1e9, 1024.0 * 2.4e9 / rate);
}

int main() {
    int *d_out;
    hipMalloc(&d_out, 256 * 4 * 8 * 64 * sizeof(int));
    run("v_add_u32", k_add, d_out); run("v_sub_u32", k_sub, d_out); run("v_and_b32", k_and, d_out); run("v_lshl_or_b32", k_lshlor, d_out);
    run("v_and_or_b32", k_andor, d_out); run("v_lshl_add_u32", k_lshladd, d_out); run("v_bfi_b32", k_bfi, d_out); run("v_cndmask_b32", k_cnd, d_out);
    run("v_perm_b32", k_perm, d_out); run("v_mad_u32_u16", k_mad16, d_out); run("v_mov_b32_dpp wave_shr", k_dpp, d_out);
    run("v_pk_add_u16", k_pkadd, d_out); run("v_pk_sub_u16 clamp", k_pksubsat, d_out); run("v_pk_min_u16", k_pkminu, d_out);
    run("v_pk_max_i16", k_pkmaxi, d_out); run("v_pk_min_i16 op_sel_hi", k_pkminsel, d_out); run("v_max3_i16 op_sel", k_max3, d_out);
    run("v_max3_i16 (no op_sel)", k_max3p, d_out); run("v_max3_i32", k_max3i32, d_out); run("v_max_i16", k_maxi16, d_out); run("v_pk_max_i16 op_sel swap", k_pkmaxsel, d_out);
    run("v_or_b32", k_or, d_out); run("v_xor_b32", k_xor, d_out); run("v_lshlrev_b32", k_lshl, d_out); run("v_lshrrev_b32", k_lshr, d_out);
    run("v_min_u32", k_minu, d_out); run("v_max_i32", k_maxi, d_out); run("v_mov_b32", k_mov, d_out); run("v_add3_u32", k_add3, d_out); run("v_or3_b32", k_or3, d_out);
    run("v_mad_u32_u24", k_mad24, d_out); run("v_mul_u32_u24", k_mul24, d_out); run("v_mul_lo_u32", k_mullo, d_out); run("v_mul_hi_u32", k_mulhi, d_out);
    run("v_cndmask_b32 (vcc by s_mov)", k_cnd2, d_out); run("v_cmp_gt_u32 -> vcc", k_cmp, d_out);
    run("v_min_i16", k_mini16, d_out); run("v_min_u16", k_minu16, d_out); run("v_add_u16", k_addu16, d_out);
    run("v_mad_u64_u32", k_mad64, d_out); run("v_lshl_add_u64", k_lshladd64, d_out); run("v_lshrrev_b64", k_lshr64, d_out); run("v_mov_b64", k_mov64, d_out);
    run("v_cvt_f64_i32", k_cvt, d_out); run("v_mul_f64", k_mulf64, d_out);
    run("v_bfe_u32", k_bfe, d_out); run("v_bcnt_u32_b32", k_bcnt, d_out); run("v_add_u32_sdwa", k_sdwa, d_out); run("v_pk_lshlrev_b16", k_pklshl, d_out);
    return 0;
}
