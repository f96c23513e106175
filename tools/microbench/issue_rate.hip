// issue_rate.hip -- cycles per instruction of ONE wave per SIMD for different
// encodings (s_memtime around an unrolled loop of 16 independent instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define BODY16(STR)                                                                                                   \
  asm volatile(STR(0) STR(1) STR(2) STR(3) STR(4) STR(5) STR(6) STR(7) STR(8) STR(9) STR(10) STR(11) STR(12) STR(13) \
                   STR(14) STR(15)                                                                                    \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),      \
                 "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
               : "v"(b), "v"(c), "s"(sb) : "s20", "s21", "vcc", "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15")

#define FMA_VOP3(i) "v_fma_f32 %" #i ", %16, %17, %" #i "\n"
#define FMAC_E32(i) "v_fmac_f32_e32 %" #i ", %16, %17\n"
#define MUL_E32(i) "v_mul_f32_e32 %" #i ", %16, %" #i "\n"
#define MUL_SGPR(i) "v_mul_f32_e32 %" #i ", %18, %" #i "\n"
#define FMA_SGPR(i) "v_fma_f32 %" #i ", %18, %17, %" #i "\n"
#define MOV_DPP(i) "v_mov_b32_dpp %" #i ", %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define ADD_DPP(i) "v_add_f32_dpp %" #i ", %16, %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define CNDMASK(i) "v_cndmask_b32_e32 %" #i ", %16, %17, vcc\n"
#define RCP(i) "v_rcp_f32_e32 %" #i ", %16\n"
#define MOV(i) "v_mov_b32_e32 %" #i ", %16\n"
#define ACCW(i) "v_accvgpr_write_b32 a" #i ", %16\n"
#define DEP_FMAC(i) "v_fmac_f32_e32 %0, %16, %17\n"
#define DEP_FMA3(i) "v_fma_f32 %0, %16, %17, %0\n"
#define DEP_MUL_ALT(i) "v_mul_f32_e32 %0, %16, %0\n"
#define NOP_ONLY(i) "s_nop 0\n"
#define SALU(i) "s_add_u32 s20, s20, 1\n"


#define CND_E64(i) "v_cndmask_b32_e64 %" #i ", %16, %17, s[20:21]\n"
#define CMP_CND(i) "v_cmp_lt_f32_e32 vcc, %16, %" #i "\nv_cndmask_b32_e32 %" #i ", %16, %17, vcc\n"
#define CMP_ONLY(i) "v_cmp_lt_f32_e32 vcc, %16, %" #i "\n"
#define CMP_E64(i) "v_cmp_lt_f32_e64 s[20:21], %16, %" #i "\n"
#define MAXF(i) "v_max_f32_e32 %" #i ", %16, %" #i "\n"
#define XORB(i) "v_xor_b32_e32 %" #i ", %16, %" #i "\n"
#define MULLO(i) "v_mul_lo_u32 %" #i ", %16, %17\n"
#define MULHI(i) "v_mul_hi_u32 %" #i ", %16, %17\n"
#define SQRT(i) "v_sqrt_f32_e32 %" #i ", %16\n"
#define SINF(i) "v_sin_f32_e32 %" #i ", %16\n"
#define ACCR(i) "v_accvgpr_read_b32 %" #i ", a" #i "\n"
#define ACCWR(i) "v_accvgpr_write_b32 a" #i ", %16\n"
#define RDLANE(i) "v_readlane_b32 s20, %16, 3\n"
#define WRLANE(i) "v_writelane_b32 %" #i ", s20, 3\n"
#define PKFMA(i) "v_pk_mul_f32 %16, %16, %16\n"
#define FMA_NEG(i) "v_fma_f32 %" #i ", -%16, %17, %" #i "\n"
#define SUB_E32(i) "v_sub_f32_e32 %" #i ", %16, %" #i "\n"
#define MUL_E64ABS(i) "v_mul_f32_e64 %" #i ", |%16|, %" #i "\n"
#define SMOV(i) "s_mov_b32 s20, s21\n"
#define MIX1(i) "v_fmac_f32_e32 %" #i ", %16, %17\ns_mov_b32 s20, s21\n"

#define FMAC_DPP(i) "v_fmac_f32_dpp %" #i ", %16, %17 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define FMAC_DPP_NEG(i) "v_fmac_f32_dpp %" #i ", -%16, %17 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define MUL_DPP(i) "v_mul_f32_dpp %" #i ", %16, %17 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define ADD_ROR8(i) "v_add_f32_dpp %" #i ", %16, %" #i " row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define DEP_FMAC_DPP(i) "v_fmac_f32_dpp %0, %16, %17 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define DPP_THEN_FMA(i) "v_fmac_f32_dpp %" #i ", %16, %17 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_fmac_f32_e32 %" #i ", %16, %17\n"
#define NOP1(i) "s_nop 1\n"
#define MFMA4(i) "v_mfma_f32_4x4x1_16b_f32 a[0:3], %16, %17, a[0:3]\n"
#define MFMA4_IND(i) "v_mfma_f32_4x4x1_16b_f32 a[" "4*(" #i "%4)" ":" "4*(" #i "%4)+3" "], %16, %17, a[0:3]\n"
// branches (exec is never zero here): not taken, taken to the next instruction, taken over 8 skipped instructions
#define BR_NOT_TAKEN(i) "s_cbranch_execz .Lnt%=_" #i "\n.Lnt%=_" #i ":\n"
#define BR_TAKEN(i) "s_cbranch_execnz .Ltk%=_" #i "\n.Ltk%=_" #i ":\n"
#define BR_TAKEN_FAR(i) "s_cbranch_execnz .Ltf%=_" #i "\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\n.Ltf%=_" #i ":\n"
#define BR_NT_VALU(i) "v_fmac_f32_e32 %" #i ", %16, %17\ns_cbranch_execz .Lnv%=_" #i "\n.Lnv%=_" #i ":\n"
#define BR_TK_VALU(i) "v_fmac_f32_e32 %" #i ", %16, %17\ns_cbranch_execnz .Ltv%=_" #i "\n.Ltv%=_" #i ":\n"
// selects as the compiler writes them: one compare, then one or three v_cndmask on its mask -- through VCC (e32) or through an SGPR pair (e64)
#define CMP64_CND64(i) "v_cmp_lt_f32_e64 s[20:21], %16, %" #i "\nv_cndmask_b32_e64 %" #i ", %16, %17, s[20:21]\n"
#define CMP32_3CND32(i) "v_cmp_lt_f32_e32 vcc, %16, %" #i "\nv_cndmask_b32_e32 %" #i ", %16, %17, vcc\nv_cndmask_b32_e32 %" #i ", %17, %16, vcc\nv_cndmask_b32_e32 %" #i ", %16, %17, vcc\n"
#define CMP64_3CND64(i) "v_cmp_lt_f32_e64 s[20:21], %16, %" #i "\nv_cndmask_b32_e64 %" #i ", %16, %17, s[20:21]\nv_cndmask_b32_e64 %" #i ", %17, %16, s[20:21]\nv_cndmask_b32_e64 %" #i ", %16, %17, s[20:21]\n"
#define CMP32_FMA_CND32(i) "v_cmp_lt_f32_e32 vcc, %16, %" #i "\nv_fmac_f32_e32 %" #i ", %16, %17\nv_fmac_f32_e32 %" #i ", %16, %17\nv_cndmask_b32_e32 %" #i ", %16, %17, vcc\n"
#define CND32_FMA3(i) "v_cndmask_b32_e32 %" #i ", %16, %17, vcc\nv_fmac_f32_e32 %" #i ", %16, %17\nv_fmac_f32_e32 %" #i ", %16, %17\nv_fmac_f32_e32 %" #i ", %16, %17\n"
#define CND64_FMA3(i) "v_cndmask_b32_e64 %" #i ", %16, %17, s[20:21]\nv_fmac_f32_e32 %" #i ", %16, %17\nv_fmac_f32_e32 %" #i ", %16, %17\nv_fmac_f32_e32 %" #i ", %16, %17\n"
#define CND_VCC_E64(i) "v_cndmask_b32_e64 %" #i ", %16, %17, vcc\n"
#define KERNEL(NAME, STR)                                                          \
  __global__ void NAME(float* out, long long* cyc, int iters, float b, float c) { \
    float a[16];                                                                   \
    float sb = __builtin_amdgcn_readfirstlane(b);                                  \
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;                           \
    long long t0 = __builtin_amdgcn_s_memtime();                                   \
    for (int it = 0; it < iters; ++it) {                                           \
      BODY16(STR);                                                                 \
      BODY16(STR);                                                                 \
      BODY16(STR);                                                                 \
      BODY16(STR);                                                                 \
    }                                                                              \
    long long t1 = __builtin_amdgcn_s_memtime();                                   \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                               \
    float s = 0.f;                                                                 \
    for (int i = 0; i < 16; ++i) s += a[i];                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
  }

KERNEL(k_fma_vop3, FMA_VOP3)
KERNEL(k_fmac_e32, FMAC_E32)
KERNEL(k_mul_e32, MUL_E32)
KERNEL(k_mul_sgpr, MUL_SGPR)
KERNEL(k_fma_sgpr, FMA_SGPR)
KERNEL(k_mov_dpp, MOV_DPP)
KERNEL(k_add_dpp, ADD_DPP)
KERNEL(k_cndmask, CNDMASK)
KERNEL(k_rcp, RCP)
KERNEL(k_mov, MOV)
KERNEL(k_dep_fmac, DEP_FMAC)
KERNEL(k_dep_fma3, DEP_FMA3)
KERNEL(k_dep_mul, DEP_MUL_ALT)
KERNEL(k_nop, NOP_ONLY)

KERNEL(k_cnd_e64, CND_E64)
KERNEL(k_cmp_cnd, CMP_CND)
KERNEL(k_cmp_only, CMP_ONLY)
KERNEL(k_cmp_e64, CMP_E64)
KERNEL(k_max, MAXF)
KERNEL(k_xor, XORB)
KERNEL(k_mullo, MULLO)
KERNEL(k_mulhi, MULHI)
KERNEL(k_sqrt, SQRT)
KERNEL(k_sin, SINF)
KERNEL(k_accr, ACCR)
KERNEL(k_accw, ACCWR)
KERNEL(k_rdlane, RDLANE)
KERNEL(k_wrlane, WRLANE)
KERNEL(k_fmaneg, FMA_NEG)
KERNEL(k_sub, SUB_E32)
KERNEL(k_mulabs, MUL_E64ABS)
KERNEL(k_smov, SMOV)
KERNEL(k_mix1, MIX1)

KERNEL(k_fmac_dpp, FMAC_DPP)
KERNEL(k_fmac_dpp_neg, FMAC_DPP_NEG)
KERNEL(k_mul_dpp, MUL_DPP)
KERNEL(k_add_ror8, ADD_ROR8)
KERNEL(k_dep_fmac_dpp, DEP_FMAC_DPP)
KERNEL(k_dpp_then_fma, DPP_THEN_FMA)
KERNEL(k_nop1, NOP1)
KERNEL(k_mfma4, MFMA4)

KERNEL(k_br_nt, BR_NOT_TAKEN)
KERNEL(k_br_tk, BR_TAKEN)
KERNEL(k_br_tf, BR_TAKEN_FAR)
KERNEL(k_br_nt_valu, BR_NT_VALU)
KERNEL(k_br_tk_valu, BR_TK_VALU)

KERNEL(k_cmp64_cnd64, CMP64_CND64)
KERNEL(k_cmp32_3cnd32, CMP32_3CND32)
KERNEL(k_cmp64_3cnd64, CMP64_3CND64)
KERNEL(k_cmp32_fma_cnd32, CMP32_FMA_CND32)
KERNEL(k_cnd32_fma3, CND32_FMA3)
KERNEL(k_cnd64_fma3, CND64_FMA3)
KERNEL(k_cnd_vcc_e64, CND_VCC_E64)

typedef void (*kern_t)(float*, long long*, int, float, float);

int main() {
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * 1024 * 1024);
  (void)hipMalloc(&cyc, sizeof(long long) * 4096);
  const int iters = 20000;
  struct { const char* name; kern_t k; } ks[] = {
      {"v_fma_f32 (VOP3, 8 B) independent", k_fma_vop3}, {"v_fmac_f32_e32 (4 B) independent", k_fmac_e32},
      {"v_mul_f32_e32 independent", k_mul_e32},          {"v_mul_f32_e32 with SGPR src", k_mul_sgpr},
      {"v_fma_f32 with SGPR src", k_fma_sgpr},           {"v_mov_b32_dpp quad_perm", k_mov_dpp},
      {"v_add_f32_dpp quad_perm", k_add_dpp},            {"v_cndmask_b32_e32", k_cndmask},
      {"v_rcp_f32", k_rcp},                              {"v_mov_b32", k_mov},
      {"v_fmac_f32_e32 dependent chain", k_dep_fmac},    {"v_fma_f32 VOP3 dependent chain", k_dep_fma3},
      {"v_mul_f32_e32 dependent chain", k_dep_mul},      {"s_nop 0", k_nop},
      {"v_cndmask_b32_e64 sgpr-pair mask", k_cnd_e64},
      {"v_cmp + v_cndmask pair (per 2 instr)", k_cmp_cnd},
      {"v_cmp_lt_f32_e32 -> vcc", k_cmp_only},
      {"v_cmp_lt_f32_e64 -> sgpr pair", k_cmp_e64},
      {"v_max_f32_e32", k_max},
      {"v_xor_b32", k_xor},
      {"v_mul_lo_u32", k_mullo},
      {"v_mul_hi_u32", k_mulhi},
      {"v_sqrt_f32", k_sqrt},
      {"v_sin_f32", k_sin},
      {"v_accvgpr_read_b32", k_accr},
      {"v_accvgpr_write_b32", k_accw},
      {"v_readlane_b32", k_rdlane},
      {"v_writelane_b32", k_wrlane},
      {"v_fma_f32 with neg modifier", k_fmaneg},
      {"v_sub_f32_e32", k_sub},
      {"v_mul_f32_e64 |abs|", k_mulabs},
      {"s_mov_b32", k_smov},
      {"v_fmac + s_mov interleaved (per pair)", k_mix1},
      {"v_fmac_f32_dpp quad_perm broadcast", k_fmac_dpp},
      {"v_fmac_f32_dpp with neg modifier", k_fmac_dpp_neg},
      {"v_mul_f32_dpp quad_perm broadcast", k_mul_dpp},
      {"v_add_f32_dpp row_ror:8", k_add_ror8},
      {"v_fmac_f32_dpp dependent accumulator", k_dep_fmac_dpp},
      {"v_fmac_f32_dpp + v_fmac_f32_e32 (per pair)", k_dpp_then_fma},
      {"s_nop 1", k_nop1},
      {"v_mfma_f32_4x4x1_16b_f32 same accumulator", k_mfma4},
      {"s_cbranch_execz not taken", k_br_nt},
      {"s_cbranch_execnz taken (to the next instruction)", k_br_tk},
      {"s_cbranch_execnz taken over 8 skipped instructions", k_br_tf},
      {"v_fmac + s_cbranch_execz not taken (per pair)", k_br_nt_valu},
      {"v_fmac + s_cbranch_execnz taken (per pair)", k_br_tk_valu},
      {"v_cmp_e64 + v_cndmask_e64 sgpr pair (per pair)", k_cmp64_cnd64},
      {"v_cmp_e32 + 3 v_cndmask_e32 vcc (per 4)", k_cmp32_3cnd32},
      {"v_cmp_e64 + 3 v_cndmask_e64 sgpr pair (per 4)", k_cmp64_3cnd64},
      {"v_cmp_e32, 2 v_fmac, v_cndmask_e32 vcc (per 4)", k_cmp32_fma_cnd32},
      {"v_cndmask_e32 vcc + 3 v_fmac (per 4)", k_cnd32_fma3},
      {"v_cndmask_e64 sgpr pair + 3 v_fmac (per 4)", k_cnd64_fma3},
      {"v_cndmask_b32_e64 with vcc as the mask", k_cnd_vcc_e64}};
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(k_fma_vop3, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
  (void)hipDeviceSynchronize();
  // two and four waves per SIMD for the instruction classes of the step kernel: what the vector ALU sustains when it is shared
  for (int wps : {2, 4}) {
    const char* shared[] = {"v_fma_f32 (VOP3, 8 B) independent", "v_fmac_f32_e32 (4 B) independent", "v_mul_f32_e32 independent", "v_sub_f32_e32",
                            "v_fmac_f32_dpp quad_perm broadcast", "v_add_f32_dpp row_ror:8", "v_mov_b32_dpp quad_perm", "v_cndmask_b32_e64 sgpr-pair mask",
                            "v_mov_b32", "v_max_f32_e32", "v_rcp_f32", "v_fmac_f32_e32 dependent chain", "s_mov_b32"};
    for (auto& k : ks) {
      bool wanted = false;
      for (const char* name : shared) wanted = wanted || std::string(name) == k.name;
      if (!wanted) continue;
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k.k, dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
      (void)hipEventRecord(e1, 0);
      (void)hipDeviceSynchronize();
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(256);
      (void)hipMemcpy(h.data(), cyc, sizeof(long long) * 256, hipMemcpyDeviceToHost);
      long long longest = 0;
      for (long long v : h) longest = v > longest ? v : longest;
      // (wall clock beside the wave's own counter: the counter is a constant-rate timer, the wall clock is what a launch costs)
      printf("%d wave(s)/SIMD  %-40s %.2f ticks per instruction per wave (slowest block %.2f), kernel %.3f ms = %.2f ns per instruction per wave\n", wps, k.name,
             h[0] / (64.0 * iters), longest / (64.0 * iters), ms, ms * 1e6 / (64.0 * iters));
    }
  }
  for (int wps : {1}) {
    for (auto& k : ks) {
      hipLaunchKernelGGL(k.k, dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
      (void)hipDeviceSynchronize();
      std::vector<long long> h(256);
      (void)hipMemcpy(h.data(), cyc, sizeof(long long) * 256, hipMemcpyDeviceToHost);
      printf("%d wave(s)/SIMD  %-40s %.2f cycles per instruction per wave\n", wps, k.name, h[0] / (64.0 * iters));
    }
  }
  return 0;
}
