// Single-wave co-issue microbenchmark for gfx950: one wave per SIMD runs {v_mfma_f32_32x32x16_bf16 ; K filler instructions} x 8 per loop
// trip (8 different accumulators) and reports shader cycles per MFMA.  Which fillers hide under an MFMA, and which MFMA operand forms
// (accumulator in AGPRs or VGPRs, B operand from AGPRs) change that?   hipcc --offload-arch=gfx950 -O2 mfma_fill.hip -o mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define STR(x) #x
#define XSTR(x) STR(x)
// accumulator j of the trip: AGPR form a[16j : 16j+15], VGPR form v[64+16j : ...]
#define MFMA_A(j) "v_mfma_f32_32x32x16_bf16 a[" XSTR(j) "*16:" XSTR(j) "*16+15], v[0:3], v[4:7], a[" XSTR(j) "*16:" XSTR(j) "*16+15]\n"

template <int V> __device__ __forceinline__ void body();

// the asm text needs literal register numbers: generate the eight groups by macro
#define G_A(lo, hi) "v_mfma_f32_32x32x16_bf16 a[" #lo ":" #hi "], v[0:3], v[4:7], a[" #lo ":" #hi "]\n"
#define G_AB(lo, hi) "v_mfma_f32_32x32x16_bf16 a[" #lo ":" #hi "], v[0:3], a[128:131], a[" #lo ":" #hi "]\n"
#define G_V(lo, hi) "v_mfma_f32_32x32x16_bf16 v[" #lo ":" #hi "], v[0:3], v[4:7], v[" #lo ":" #hi "]\n"
#define G_VB(lo, hi) "v_mfma_f32_32x32x16_bf16 v[" #lo ":" #hi "], v[0:3], a[128:131], v[" #lo ":" #hi "]\n"
#define FMA "v_fma_f32 v8, v9, v10, v11\n"
#define FMA2 "v_fma_f32 v12, v13, v10, v11\n"
#define ADD "v_add_f32 v14, v15, v16\n"
#define ADD2 "v_add_f32 v17, v18, v16\n"
#define EXP "v_exp_f32 v19, v20\n"
#define CVT "v_cvt_pk_bf16_f32 v21, v22, v23\n"
#define MOV "v_mov_b32 v24, v25\n"
#define NOPI "s_nop 0\n"
#define ACCR "v_accvgpr_read_b32 v26, a140\n"
#define PKADD "v_pk_add_f32 v[28:29], v[30:31], v[32:33]\n"
#define DSR "ds_read_b128 v[36:39], v40\n"
// reads of the VGPR-form accumulators by the fillers (the softmax reads the score registers)
#define FMA_S "v_fma_f32 v8, v200, v10, v11\n"
#define FMA_R(r) "v_fma_f32 v8, v" #r ", v10, v11\n"
#define FMA_R2(r) "v_fma_f32 v12, v" #r ", v10, v11\n"

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v28","v29","v30","v31","v32","v33","v36","v37","v38","v39","v40"

#define TRIP_A(F) G_A(0,15) F G_A(16,31) F G_A(32,47) F G_A(48,63) F G_A(64,79) F G_A(80,95) F G_A(96,111) F G_A(112,127) F
#define TRIP_AB(F) G_AB(0,15) F G_AB(16,31) F G_AB(32,47) F G_AB(48,63) F G_AB(64,79) F G_AB(80,95) F G_AB(96,111) F G_AB(112,127) F
#define TRIP_V(F) G_V(64,79) F G_V(80,95) F G_V(96,111) F G_V(112,127) F G_V(128,143) F G_V(144,159) F G_V(160,175) F G_V(176,191) F
#define TRIP_VB(F) G_VB(64,79) F G_VB(80,95) F G_VB(96,111) F G_VB(112,127) F G_VB(128,143) F G_VB(144,159) F G_VB(160,175) F G_VB(176,191) F
// two alternating accumulators only (what QK^T does: S_A, S_B)
#define TRIP_V2(F) G_V(64,79) F G_V(80,95) F G_V(64,79) F G_V(80,95) F G_V(64,79) F G_V(80,95) F G_V(64,79) F G_V(80,95) F

struct Case { const char* name; void (*launch)(unsigned long long*, int, hipStream_t); };

#define KERNEL(NAME, TEXT)                                                                                    \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned long long* out, int trips) {                       \
        asm volatile("v_mov_b32 v40, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v16, 0\n v_mov_b32 v20, 0\n" ::: CLOB); \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                 \
        for (int i = 0; i < trips; ++i) {                                                                     \
            asm volatile(TEXT ::: CLOB, "a0", "a127", "a128", "a131", "a140", "v64", "v191", "v200", "v203", "v32", "v35", "v48", "v51");         \
        }                                                                                                     \
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");                                                     \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                 \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                            \
    }                                                                                                         \
    static void l_##NAME(unsigned long long* o, int trips, hipStream_t s) { hipLaunchKernelGGL(k_##NAME, dim3(256), dim3(256), 0, s, o, trips); }

KERNEL(a_f0, TRIP_A(""))
KERNEL(a_fma2, TRIP_A(FMA FMA2))
KERNEL(a_fma4, TRIP_A(FMA FMA2 FMA FMA2))
KERNEL(a_fma5, TRIP_A(FMA FMA2 FMA FMA2 FMA))
KERNEL(a_fma6, TRIP_A(FMA FMA2 FMA FMA2 FMA FMA2))
KERNEL(a_fma8, TRIP_A(FMA FMA2 FMA FMA2 FMA FMA2 FMA FMA2))
KERNEL(a_add6, TRIP_A(ADD ADD2 ADD ADD2 ADD ADD2))
KERNEL(a_exp2, TRIP_A(EXP EXP))
KERNEL(a_exp4, TRIP_A(EXP EXP EXP EXP))
KERNEL(a_mix, TRIP_A(FMA EXP ADD FMA2 EXP ADD2 CVT))          /* one softmax pair: 7 instructions */
KERNEL(a_mixh, TRIP_A(FMA EXP ADD CVT))                        /* about half of it */
KERNEL(a_cvt4, TRIP_A(CVT CVT CVT CVT))
KERNEL(a_mov6, TRIP_A(MOV MOV MOV MOV MOV MOV))
KERNEL(a_nop6, TRIP_A(NOPI NOPI NOPI NOPI NOPI NOPI))
KERNEL(a_acc4, TRIP_A(ACCR ACCR ACCR ACCR))
KERNEL(a_pk2, TRIP_A(PKADD PKADD))
KERNEL(a_ds1, TRIP_A(DSR))
KERNEL(ab_f0, TRIP_AB(""))
KERNEL(ab_fma5, TRIP_AB(FMA FMA2 FMA FMA2 FMA))
KERNEL(v_f0, TRIP_V(""))
KERNEL(v_fma5, TRIP_V(FMA FMA2 FMA FMA2 FMA))
KERNEL(v_mix, TRIP_V(FMA EXP ADD FMA2 EXP ADD2 CVT))
KERNEL(vb_f0, TRIP_VB(""))
KERNEL(vb_fma5, TRIP_VB(FMA FMA2 FMA FMA2 FMA))
KERNEL(vb_mix, TRIP_VB(FMA EXP ADD FMA2 EXP ADD2 CVT))
KERNEL(v2_f0, TRIP_V2(""))
KERNEL(v2_fma5, TRIP_V2(FMA FMA2 FMA FMA2 FMA))
KERNEL(v2_mix, TRIP_V2(FMA EXP ADD FMA2 EXP ADD2 CVT))
KERNEL(v_fmas5, TRIP_V(FMA_S FMA2 FMA_S FMA2 FMA_S))           /* fillers that READ a (not currently written) score register */
/* two accumulators v[64:95] written in turn; fillers read registers at various distances from them (never written in the loop) */
KERNEL(v2_r96, TRIP_V2(FMA_R(96) FMA_R2(97) FMA_R(98) FMA_R2(99)))
KERNEL(v2_r112, TRIP_V2(FMA_R(112) FMA_R2(113) FMA_R(114) FMA_R2(115)))
KERNEL(v2_r128, TRIP_V2(FMA_R(128) FMA_R2(129) FMA_R(130) FMA_R2(131)))
KERNEL(v2_r48, TRIP_V2(FMA_R(48) FMA_R2(49) FMA_R(50) FMA_R2(51)))
KERNEL(v2_r32, TRIP_V2(FMA_R(32) FMA_R2(33) FMA_R(34) FMA_R2(35)))
KERNEL(v2_r200, TRIP_V2(FMA_R(200) FMA_R2(201) FMA_R(202) FMA_R2(203)))
KERNEL(v2_r9, TRIP_V2(FMA FMA2 FMA FMA2))
/* AGPR-form accumulators (what PV does), fillers reading VGPRs that an EARLIER VGPR-form MFMA wrote: no MFMA touches VGPRs in the loop */
KERNEL(a_r96, TRIP_A(FMA_R(96) FMA_R2(97) FMA_R(98) FMA_R2(99)))

#define TEXT_flow_read \
"v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" \
"v_fma_f32 v8, v64, v10, v11\n" "v_fma_f32 v12, v65, v10, v11\n" "v_fma_f32 v14, v80, v10, v11\n" "v_fma_f32 v17, v81, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n" \
"v_fma_f32 v8, v66, v10, v11\n" "v_fma_f32 v12, v67, v10, v11\n" "v_fma_f32 v14, v82, v10, v11\n" "v_fma_f32 v17, v83, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n" \
"v_fma_f32 v8, v68, v10, v11\n" "v_fma_f32 v12, v69, v10, v11\n" "v_fma_f32 v14, v84, v10, v11\n" "v_fma_f32 v17, v85, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n" \
"v_fma_f32 v8, v70, v10, v11\n" "v_fma_f32 v12, v71, v10, v11\n" "v_fma_f32 v14, v86, v10, v11\n" "v_fma_f32 v17, v87, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n" \
"v_fma_f32 v8, v72, v10, v11\n" "v_fma_f32 v12, v73, v10, v11\n" "v_fma_f32 v14, v88, v10, v11\n" "v_fma_f32 v17, v89, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[80:95], v[0:3], v[4:7], a[80:95]\n" \
"v_fma_f32 v8, v74, v10, v11\n" "v_fma_f32 v12, v75, v10, v11\n" "v_fma_f32 v14, v90, v10, v11\n" "v_fma_f32 v17, v91, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[96:111], v[0:3], v[4:7], a[96:111]\n" \
"v_fma_f32 v8, v76, v10, v11\n" "v_fma_f32 v12, v77, v10, v11\n" "v_fma_f32 v14, v92, v10, v11\n" "v_fma_f32 v17, v93, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[112:127], v[0:3], v[4:7], a[112:127]\n" \
"v_fma_f32 v8, v78, v10, v11\n" "v_fma_f32 v12, v79, v10, v11\n" "v_fma_f32 v14, v94, v10, v11\n" "v_fma_f32 v17, v95, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" \
"v_fma_f32 v8, v64, v10, v11\n" "v_fma_f32 v12, v65, v10, v11\n" "v_fma_f32 v14, v80, v10, v11\n" "v_fma_f32 v17, v81, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n" \
"v_fma_f32 v8, v66, v10, v11\n" "v_fma_f32 v12, v67, v10, v11\n" "v_fma_f32 v14, v82, v10, v11\n" "v_fma_f32 v17, v83, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n" \
"v_fma_f32 v8, v68, v10, v11\n" "v_fma_f32 v12, v69, v10, v11\n" "v_fma_f32 v14, v84, v10, v11\n" "v_fma_f32 v17, v85, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n" \
"v_fma_f32 v8, v70, v10, v11\n" "v_fma_f32 v12, v71, v10, v11\n" "v_fma_f32 v14, v86, v10, v11\n" "v_fma_f32 v17, v87, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n" \
"v_fma_f32 v8, v72, v10, v11\n" "v_fma_f32 v12, v73, v10, v11\n" "v_fma_f32 v14, v88, v10, v11\n" "v_fma_f32 v17, v89, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[80:95], v[0:3], v[4:7], a[80:95]\n" \
"v_fma_f32 v8, v74, v10, v11\n" "v_fma_f32 v12, v75, v10, v11\n" "v_fma_f32 v14, v90, v10, v11\n" "v_fma_f32 v17, v91, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[96:111], v[0:3], v[4:7], a[96:111]\n" \
"v_fma_f32 v8, v76, v10, v11\n" "v_fma_f32 v12, v77, v10, v11\n" "v_fma_f32 v14, v92, v10, v11\n" "v_fma_f32 v17, v93, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[112:127], v[0:3], v[4:7], a[112:127]\n" \
"v_fma_f32 v8, v78, v10, v11\n" "v_fma_f32 v12, v79, v10, v11\n" "v_fma_f32 v14, v94, v10, v11\n" "v_fma_f32 v17, v95, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v64, v10, v11\n" "v_fma_f32 v12, v65, v10, v11\n" "v_fma_f32 v14, v80, v10, v11\n" "v_fma_f32 v17, v81, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v66, v10, v11\n" "v_fma_f32 v12, v67, v10, v11\n" "v_fma_f32 v14, v82, v10, v11\n" "v_fma_f32 v17, v83, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v68, v10, v11\n" "v_fma_f32 v12, v69, v10, v11\n" "v_fma_f32 v14, v84, v10, v11\n" "v_fma_f32 v17, v85, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v70, v10, v11\n" "v_fma_f32 v12, v71, v10, v11\n" "v_fma_f32 v14, v86, v10, v11\n" "v_fma_f32 v17, v87, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v72, v10, v11\n" "v_fma_f32 v12, v73, v10, v11\n" "v_fma_f32 v14, v88, v10, v11\n" "v_fma_f32 v17, v89, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v74, v10, v11\n" "v_fma_f32 v12, v75, v10, v11\n" "v_fma_f32 v14, v90, v10, v11\n" "v_fma_f32 v17, v91, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v76, v10, v11\n" "v_fma_f32 v12, v77, v10, v11\n" "v_fma_f32 v14, v92, v10, v11\n" "v_fma_f32 v17, v93, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v78, v10, v11\n" "v_fma_f32 v12, v79, v10, v11\n" "v_fma_f32 v14, v94, v10, v11\n" "v_fma_f32 v17, v95, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v64, v10, v11\n" "v_fma_f32 v12, v65, v10, v11\n" "v_fma_f32 v14, v80, v10, v11\n" "v_fma_f32 v17, v81, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v66, v10, v11\n" "v_fma_f32 v12, v67, v10, v11\n" "v_fma_f32 v14, v82, v10, v11\n" "v_fma_f32 v17, v83, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v68, v10, v11\n" "v_fma_f32 v12, v69, v10, v11\n" "v_fma_f32 v14, v84, v10, v11\n" "v_fma_f32 v17, v85, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v70, v10, v11\n" "v_fma_f32 v12, v71, v10, v11\n" "v_fma_f32 v14, v86, v10, v11\n" "v_fma_f32 v17, v87, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v72, v10, v11\n" "v_fma_f32 v12, v73, v10, v11\n" "v_fma_f32 v14, v88, v10, v11\n" "v_fma_f32 v17, v89, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v74, v10, v11\n" "v_fma_f32 v12, v75, v10, v11\n" "v_fma_f32 v14, v90, v10, v11\n" "v_fma_f32 v17, v91, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v76, v10, v11\n" "v_fma_f32 v12, v77, v10, v11\n" "v_fma_f32 v14, v92, v10, v11\n" "v_fma_f32 v17, v93, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v78, v10, v11\n" "v_fma_f32 v12, v79, v10, v11\n" "v_fma_f32 v14, v94, v10, v11\n" "v_fma_f32 v17, v95, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" \
"v_fma_f32 v8, v96, v10, v11\n" "v_fma_f32 v12, v97, v10, v11\n" "v_fma_f32 v14, v112, v10, v11\n" "v_fma_f32 v17, v113, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n" \
"v_fma_f32 v8, v98, v10, v11\n" "v_fma_f32 v12, v99, v10, v11\n" "v_fma_f32 v14, v114, v10, v11\n" "v_fma_f32 v17, v115, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n" \
"v_fma_f32 v8, v100, v10, v11\n" "v_fma_f32 v12, v101, v10, v11\n" "v_fma_f32 v14, v116, v10, v11\n" "v_fma_f32 v17, v117, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n" \
"v_fma_f32 v8, v102, v10, v11\n" "v_fma_f32 v12, v103, v10, v11\n" "v_fma_f32 v14, v118, v10, v11\n" "v_fma_f32 v17, v119, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n" \
"v_fma_f32 v8, v104, v10, v11\n" "v_fma_f32 v12, v105, v10, v11\n" "v_fma_f32 v14, v120, v10, v11\n" "v_fma_f32 v17, v121, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[80:95], v[0:3], v[4:7], a[80:95]\n" \
"v_fma_f32 v8, v106, v10, v11\n" "v_fma_f32 v12, v107, v10, v11\n" "v_fma_f32 v14, v122, v10, v11\n" "v_fma_f32 v17, v123, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[96:111], v[0:3], v[4:7], a[96:111]\n" \
"v_fma_f32 v8, v108, v10, v11\n" "v_fma_f32 v12, v109, v10, v11\n" "v_fma_f32 v14, v124, v10, v11\n" "v_fma_f32 v17, v125, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[112:127], v[0:3], v[4:7], a[112:127]\n" \
"v_fma_f32 v8, v110, v10, v11\n" "v_fma_f32 v12, v111, v10, v11\n" "v_fma_f32 v14, v126, v10, v11\n" "v_fma_f32 v17, v127, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" \
"v_fma_f32 v8, v96, v10, v11\n" "v_fma_f32 v12, v97, v10, v11\n" "v_fma_f32 v14, v112, v10, v11\n" "v_fma_f32 v17, v113, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n" \
"v_fma_f32 v8, v98, v10, v11\n" "v_fma_f32 v12, v99, v10, v11\n" "v_fma_f32 v14, v114, v10, v11\n" "v_fma_f32 v17, v115, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n" \
"v_fma_f32 v8, v100, v10, v11\n" "v_fma_f32 v12, v101, v10, v11\n" "v_fma_f32 v14, v116, v10, v11\n" "v_fma_f32 v17, v117, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n" \
"v_fma_f32 v8, v102, v10, v11\n" "v_fma_f32 v12, v103, v10, v11\n" "v_fma_f32 v14, v118, v10, v11\n" "v_fma_f32 v17, v119, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n" \
"v_fma_f32 v8, v104, v10, v11\n" "v_fma_f32 v12, v105, v10, v11\n" "v_fma_f32 v14, v120, v10, v11\n" "v_fma_f32 v17, v121, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[80:95], v[0:3], v[4:7], a[80:95]\n" \
"v_fma_f32 v8, v106, v10, v11\n" "v_fma_f32 v12, v107, v10, v11\n" "v_fma_f32 v14, v122, v10, v11\n" "v_fma_f32 v17, v123, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[96:111], v[0:3], v[4:7], a[96:111]\n" \
"v_fma_f32 v8, v108, v10, v11\n" "v_fma_f32 v12, v109, v10, v11\n" "v_fma_f32 v14, v124, v10, v11\n" "v_fma_f32 v17, v125, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[112:127], v[0:3], v[4:7], a[112:127]\n" \
"v_fma_f32 v8, v110, v10, v11\n" "v_fma_f32 v12, v111, v10, v11\n" "v_fma_f32 v14, v126, v10, v11\n" "v_fma_f32 v17, v127, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v96, v10, v11\n" "v_fma_f32 v12, v97, v10, v11\n" "v_fma_f32 v14, v112, v10, v11\n" "v_fma_f32 v17, v113, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v98, v10, v11\n" "v_fma_f32 v12, v99, v10, v11\n" "v_fma_f32 v14, v114, v10, v11\n" "v_fma_f32 v17, v115, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v100, v10, v11\n" "v_fma_f32 v12, v101, v10, v11\n" "v_fma_f32 v14, v116, v10, v11\n" "v_fma_f32 v17, v117, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v102, v10, v11\n" "v_fma_f32 v12, v103, v10, v11\n" "v_fma_f32 v14, v118, v10, v11\n" "v_fma_f32 v17, v119, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v104, v10, v11\n" "v_fma_f32 v12, v105, v10, v11\n" "v_fma_f32 v14, v120, v10, v11\n" "v_fma_f32 v17, v121, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v106, v10, v11\n" "v_fma_f32 v12, v107, v10, v11\n" "v_fma_f32 v14, v122, v10, v11\n" "v_fma_f32 v17, v123, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v108, v10, v11\n" "v_fma_f32 v12, v109, v10, v11\n" "v_fma_f32 v14, v124, v10, v11\n" "v_fma_f32 v17, v125, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v110, v10, v11\n" "v_fma_f32 v12, v111, v10, v11\n" "v_fma_f32 v14, v126, v10, v11\n" "v_fma_f32 v17, v127, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v96, v10, v11\n" "v_fma_f32 v12, v97, v10, v11\n" "v_fma_f32 v14, v112, v10, v11\n" "v_fma_f32 v17, v113, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v98, v10, v11\n" "v_fma_f32 v12, v99, v10, v11\n" "v_fma_f32 v14, v114, v10, v11\n" "v_fma_f32 v17, v115, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v100, v10, v11\n" "v_fma_f32 v12, v101, v10, v11\n" "v_fma_f32 v14, v116, v10, v11\n" "v_fma_f32 v17, v117, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v102, v10, v11\n" "v_fma_f32 v12, v103, v10, v11\n" "v_fma_f32 v14, v118, v10, v11\n" "v_fma_f32 v17, v119, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v104, v10, v11\n" "v_fma_f32 v12, v105, v10, v11\n" "v_fma_f32 v14, v120, v10, v11\n" "v_fma_f32 v17, v121, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v106, v10, v11\n" "v_fma_f32 v12, v107, v10, v11\n" "v_fma_f32 v14, v122, v10, v11\n" "v_fma_f32 v17, v123, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v108, v10, v11\n" "v_fma_f32 v12, v109, v10, v11\n" "v_fma_f32 v14, v124, v10, v11\n" "v_fma_f32 v17, v125, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v110, v10, v11\n" "v_fma_f32 v12, v111, v10, v11\n" "v_fma_f32 v14, v126, v10, v11\n" "v_fma_f32 v17, v127, v10, v11\n" \


__global__ __launch_bounds__(256) void k_flow_read(unsigned long long* out, int trips) {
    asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n" ::: CLOB);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < trips; ++i) { asm volatile(TEXT_flow_read ::: CLOB, "a0", "a127", "a128", "a131", "v64", "v127"); }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (t1 - t0) / 8;      /* 64 MFMAs per trip: main() divides by 8 */
}
static void l_flow_read(unsigned long long* o, int trips, hipStream_t s) { hipLaunchKernelGGL(k_flow_read, dim3(256), dim3(256), 0, s, o, trips); }
#define TEXT_flow_const \
"v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[80:95], v[0:3], v[4:7], a[80:95]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[96:111], v[0:3], v[4:7], a[96:111]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[112:127], v[0:3], v[4:7], a[112:127]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[80:95], v[0:3], v[4:7], a[80:95]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[96:111], v[0:3], v[4:7], a[96:111]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[112:127], v[0:3], v[4:7], a[112:127]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[80:95], v[0:3], v[4:7], a[80:95]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[96:111], v[0:3], v[4:7], a[96:111]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[112:127], v[0:3], v[4:7], a[112:127]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[80:95], v[0:3], v[4:7], a[80:95]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[96:111], v[0:3], v[4:7], a[96:111]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 a[112:127], v[0:3], v[4:7], a[112:127]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_fma_f32 v8, v9, v10, v11\n" "v_fma_f32 v12, v13, v10, v11\n" "v_fma_f32 v14, v15, v10, v11\n" "v_fma_f32 v17, v18, v10, v11\n" \


__global__ __launch_bounds__(256) void k_flow_const(unsigned long long* out, int trips) {
    asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n" ::: CLOB);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < trips; ++i) { asm volatile(TEXT_flow_const ::: CLOB, "a0", "a127", "a128", "a131", "v64", "v127"); }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (t1 - t0) / 8;      /* 64 MFMAs per trip: main() divides by 8 */
}
static void l_flow_const(unsigned long long* o, int trips, hipStream_t s) { hipLaunchKernelGGL(k_flow_const, dim3(256), dim3(256), 0, s, o, trips); }
#define TEXT_flow_none \
"v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" \
"v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n" \
"v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n" \
"v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n" \
"v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n" \
"v_mfma_f32_32x32x16_bf16 a[80:95], v[0:3], v[4:7], a[80:95]\n" \
"v_mfma_f32_32x32x16_bf16 a[96:111], v[0:3], v[4:7], a[96:111]\n" \
"v_mfma_f32_32x32x16_bf16 a[112:127], v[0:3], v[4:7], a[112:127]\n" \
"v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" \
"v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n" \
"v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n" \
"v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n" \
"v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n" \
"v_mfma_f32_32x32x16_bf16 a[80:95], v[0:3], v[4:7], a[80:95]\n" \
"v_mfma_f32_32x32x16_bf16 a[96:111], v[0:3], v[4:7], a[96:111]\n" \
"v_mfma_f32_32x32x16_bf16 a[112:127], v[0:3], v[4:7], a[112:127]\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_mfma_f32_32x32x16_bf16 v[96:111], v[0:3], a[128:131], v[96:111]\n" \
"v_mfma_f32_32x32x16_bf16 v[112:127], v[0:3], a[128:131], v[112:127]\n" \
"v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" \
"v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n" \
"v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n" \
"v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n" \
"v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n" \
"v_mfma_f32_32x32x16_bf16 a[80:95], v[0:3], v[4:7], a[80:95]\n" \
"v_mfma_f32_32x32x16_bf16 a[96:111], v[0:3], v[4:7], a[96:111]\n" \
"v_mfma_f32_32x32x16_bf16 a[112:127], v[0:3], v[4:7], a[112:127]\n" \
"v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" \
"v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n" \
"v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n" \
"v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n" \
"v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n" \
"v_mfma_f32_32x32x16_bf16 a[80:95], v[0:3], v[4:7], a[80:95]\n" \
"v_mfma_f32_32x32x16_bf16 a[96:111], v[0:3], v[4:7], a[96:111]\n" \
"v_mfma_f32_32x32x16_bf16 a[112:127], v[0:3], v[4:7], a[112:127]\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \
"v_mfma_f32_32x32x16_bf16 v[64:79], v[0:3], a[128:131], v[64:79]\n" \
"v_mfma_f32_32x32x16_bf16 v[80:95], v[0:3], a[128:131], v[80:95]\n" \


__global__ __launch_bounds__(256) void k_flow_none(unsigned long long* out, int trips) {
    asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n" ::: CLOB);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < trips; ++i) { asm volatile(TEXT_flow_none ::: CLOB, "a0", "a127", "a128", "a131", "v64", "v127"); }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (t1 - t0) / 8;      /* 64 MFMAs per trip: main() divides by 8 */
}
static void l_flow_none(unsigned long long* o, int trips, hipStream_t s) { hipLaunchKernelGGL(k_flow_none, dim3(256), dim3(256), 0, s, o, trips); }
int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    std::vector<Case> cases = {
#define C(N) {#N, l_##N}
        C(a_f0), C(a_fma2), C(a_fma4), C(a_fma5), C(a_fma6), C(a_fma8), C(a_add6), C(a_exp2), C(a_exp4), C(a_mix), C(a_mixh), C(a_cvt4),
        C(a_mov6), C(a_nop6), C(a_acc4), C(a_pk2), C(a_ds1), C(ab_f0), C(ab_fma5), C(v_f0), C(v_fma5), C(v_mix), C(vb_f0), C(vb_fma5),
        C(vb_mix), C(v2_f0), C(v2_fma5), C(v2_mix), C(v_fmas5), C(v2_r96), C(v2_r112), C(v2_r128), C(v2_r48), C(v2_r32), C(v2_r200), C(v2_r9), C(a_r96), C(flow_none), C(flow_const), C(flow_read)};
    const int trips = 2000;
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) c.launch(d, trips, 0);
        hipDeviceSynchronize();
        unsigned long long h = 0;
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%-10s %7.2f cycles per MFMA\n", c.name, (double)h / (trips * 8.0));
    }
    return 0;
}
