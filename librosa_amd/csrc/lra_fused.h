// lra_fused.h -- __global__ entry points of the fused power-of-two kernels and the list of their
// instances.
//
// The library holds ~370 kernel instances (n_fft x dtype x epilogue x power mode x ring/row alignment); hipcc
// compiles device code of one translation unit on one core, so the instances are explicitly instantiated in
// lra_inst.hip, which the build compiles once per LRA_INST_GROUP in parallel, and lra_api.hip only sees
// `extern template` declarations (LRA_FUSED_EXTERN).  Both sides expand the same X-macro list below.
#pragma once

#include <hip/hip_runtime.h>

#include "lra_dispatch.h"
#include "lra_kernels_pc.h"

// The PCM input and the output travel as separate __restrict__ kernel parameters (not only inside the
// argument struct): without the noalias guarantee hipcc must assume that the next frame's sample
// loads may read what the previous frame's spectrum stores wrote, and -- because loads may overtake
// stores in the vector memory pipeline -- it then parks the wave on s_waitcnt vmcnt(0) at the top of
// every frame until all of its stores have landed in L2, serialising FFT and store traffic.
// RA: the hop is a whole number of ring rows (lra_kernels.h, ring_rows_aligned) -- the fast ring addressing.
template <class Cfg, int MODE, int PM, int RA>
__global__ __launch_bounds__(Cfg::NT, Cfg::MIN_WAVES) void stft_kernel(lra::StftArgs<typename Cfg::real> a, const typename Cfg::real* __restrict__ y,
                                                                     void* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char lra_smem[];
    lra::Lds lds;
    lds.base = lra_smem;
    a.y = y;
    a.D = static_cast<typename Cfg::cplx*>(out);
    a.S = static_cast<typename Cfg::real*>(out);
    a.Mel = static_cast<typename Cfg::real*>(out);
    const int blk = lra::xcd_block((int)blockIdx.x, a.xcd_chunk);
    if (blk >= a.n_blocks) return;  // grid padded to a multiple of 8 (uniform exit, before any barrier)
    lra::stft_block<Cfg, MODE, PM, RA>(a, blk, lds);
}

// Second-generation forward kernel (lra_kernels2.h): PCM ring in registers, mirrored last pass with the split step in
// registers.  HD = n_fft / hop.  The register budget is sized for 3 waves per SIMD (12 slots of 8.7 KB per CU) where the
// kernel fits it without spilling (n_fft = 2048 with hop <= n_fft/4, i.e. at most 4 sample pairs in flight per thread).
#ifndef LRA_V3_WAVES
#define LRA_V3_WAVES 3  // waves per SIMD the radix 16-16-4 form is compiled for
#endif
template <class Cfg, int HD, int MODE, int PM>
__global__ __launch_bounds__(Cfg::NT, (Cfg::NT > 256 ? 1 : (Cfg::TF == 64 && HD >= 4 && PM != lra::POW_GENERAL && MODE != lra::OUT_MELR ? (Cfg::PLAN == 1 ? LRA_V3_WAVES : 3) : 2))) void stft2_kernel(lra::StftArgs<typename Cfg::real> a, const typename Cfg::real* __restrict__ y,
                                                                                  void* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char lra_smem[];
    lra::Lds lds;
    lds.base = lra_smem;
    a.y = y;
    a.D = static_cast<typename Cfg::cplx*>(out);
    a.S = static_cast<typename Cfg::real*>(out);
    a.Mel = static_cast<typename Cfg::real*>(out);
    const int blk = lra::xcd_block((int)blockIdx.x, a.xcd_chunk);
    if (blk >= a.n_blocks) return;
    lra::stft_block2<Cfg, HD, MODE, PM>(a, blk, lds);
}

// Producer / consumer form of the fused mel kernel (lra_kernels_pc.h): 192-thread workgroups [P, P, C], three waves per SIMD.
template <class Cfg, int HD, int PM>
__global__ __launch_bounds__(lra::PcLayout<Cfg>::NT, 3) void stft_pc_kernel(lra::StftArgs<typename Cfg::real> a, const typename Cfg::real* __restrict__ y, void* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char lra_smem[];
    lra::Lds lds;
    lds.base = lra_smem;
    a.y = y;
    a.Mel = static_cast<typename Cfg::real*>(out);
    const int blk = lra::xcd_block((int)blockIdx.x, a.xcd_chunk);
    if (blk >= a.n_blocks) return;
    lra::stft_pc_block<Cfg, HD, PM>(a, blk, lds);
}

template <class Cfg, int HC>
__global__ __launch_bounds__(Cfg::NT, Cfg::MIN_WAVES) void istft_kernel(lra::IstftArgs<typename Cfg::real> a, const typename Cfg::cplx* __restrict__ D,
                                                                      const typename Cfg::real* __restrict__ wss, typename Cfg::real* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) char lra_smem[];
    lra::Lds lds;
    lds.base = lra_smem;
    a.D = D;
    a.wss = wss;
    a.y = y;
    const int blk = lra::xcd_block((int)blockIdx.x, a.xcd_chunk);
    if (blk >= a.n_blocks) return;
    lra::istft_block<Cfg, HC>(a, blk, lds);
}


namespace lra {

// ---- named configurations (macro arguments cannot carry the commas of template argument lists) ----------
#define LRA_CFG_ALIAS(NAME, ...) using NAME = typename CfgSel<__VA_ARGS__>::type; using NAME##_mel = typename MelCfgOf<NAME>::type; using NAME##_melmany = typename MelManyCfgOf<NAME>::type;
LRA_CFG_ALIAS(cfg_f32_4, float, 4, 0)
LRA_CFG_ALIAS(cfg_f32_5, float, 5, 0)
LRA_CFG_ALIAS(cfg_f32_6, float, 6, 0)
LRA_CFG_ALIAS(cfg_f32_7, float, 7, 0)
LRA_CFG_ALIAS(cfg_f32_8, float, 8, 0)
LRA_CFG_ALIAS(cfg_f32_9, float, 9, 0)
LRA_CFG_ALIAS(cfg_f32_10, float, 10, 0)
LRA_CFG_ALIAS(cfg_f32_10v1, float, 10, 1)
LRA_CFG_ALIAS(cfg_f32_10v4, float, 10, 4)
LRA_CFG_ALIAS(cfg_f32_10r, float, 10, 5)
LRA_CFG_ALIAS(cfg_f32_10g, float, 10, 6)
LRA_CFG_ALIAS(cfg_f32_10q, float, 10, 7)
LRA_CFG_ALIAS(cfg_f32_11, float, 11, 0)
LRA_CFG_ALIAS(cfg_f32_12, float, 12, 0)
LRA_CFG_ALIAS(cfg_f32_13, float, 13, 0)
LRA_CFG_ALIAS(cfg_f64_4, double, 4, 0)
LRA_CFG_ALIAS(cfg_f64_5, double, 5, 0)
LRA_CFG_ALIAS(cfg_f64_6, double, 6, 0)
LRA_CFG_ALIAS(cfg_f64_7, double, 7, 0)
LRA_CFG_ALIAS(cfg_f64_8, double, 8, 0)
LRA_CFG_ALIAS(cfg_f64_9, double, 9, 0)
LRA_CFG_ALIAS(cfg_f64_10, double, 10, 0)
LRA_CFG_ALIAS(cfg_f64_11, double, 11, 0)
LRA_CFG_ALIAS(cfg_f64_12, double, 12, 0)
#undef LRA_CFG_ALIAS

}  // namespace lra

// ---- instance lists ----------------------------------------------------------------------------------------
// S(CFG, MODE, PM, RA): one stft_kernel instance; I(CFG, HC): one istft_kernel instance.
#define LRA_STFT_SET(S, C, RA)                                                                       \
    S(lra::C, 0, 2, RA)                                                                              \
    S(lra::C, 1, 1, RA) S(lra::C, 1, 2, RA) S(lra::C, 1, 3, RA)                                      \
    S(lra::C, 2, 1, RA) S(lra::C, 2, 2, RA) S(lra::C, 2, 3, RA)                                      \
    S(lra::C##_mel, 3, 1, RA) S(lra::C##_mel, 3, 2, RA) S(lra::C##_mel, 3, 3, RA)                   \
    S(lra::C##_mel, 4, 1, RA) S(lra::C##_mel, 4, 2, RA) S(lra::C##_mel, 4, 3, RA)
// direct framing (RA = 2: hop >= n_fft, no ring): complex and power epilogues
#define LRA_STFT_DIRECT(S, C) S(lra::C, 0, 2, 2) S(lra::C, 1, 1, 2) S(lra::C, 1, 2, 2) S(lra::C, 1, 3, 2)
// register ring (RA = 3 .. 6: hop = n_fft / 2, 4, 8, 16) for the large frames the second-generation kernel does not cover
#define LRA_STFT_REGRING(S, C) LRA_STFT_REGRING1(S, C, 3) LRA_STFT_REGRING1(S, C, 4) LRA_STFT_REGRING1(S, C, 5) LRA_STFT_REGRING1(S, C, 6)
#define LRA_STFT_REGRING1(S, C, RA) S(lra::C, 0, 2, RA) S(lra::C, 1, 1, RA) S(lra::C, 1, 2, RA) S(lra::C, 1, 3, RA)
// f32: both ring addressings; the overlap-add row counts HC = R/2, R/4, R/8 (8, 4, 2 at 16 points per thread)
// (HCS: rows per hop at hop = n_fft / 16 -- 1 for the 16-point configurations, which are the only ones with that instance)
#define LRA_F32_CFG(S, I, C, HCQ, HCE) LRA_STFT_SET(S, C, 0) LRA_STFT_SET(S, C, 1) LRA_STFT_DIRECT(S, C) I(lra::C, 0) I(lra::C, (2 * HCQ)) I(lra::C, HCQ) I(lra::C, HCE) LRA_ISTFT_HCS_##HCE(I, C)
#define LRA_ISTFT_HCS_2(I, C) I(lra::C, 1)
#define LRA_ISTFT_HCS_1(I, C)
#define LRA_F64_CFG(S, I, C) LRA_STFT_SET(S, C, 0) LRA_STFT_DIRECT(S, C) I(lra::C, 0)

#define LRA_INST_GROUP_0(S, I) LRA_F32_CFG(S, I, cfg_f32_10, 4, 2)
#define LRA_INST_GROUP_1(S, I) LRA_F32_CFG(S, I, cfg_f32_10v4, 2, 1)
#define LRA_INST_GROUP_2(S, I) LRA_F32_CFG(S, I, cfg_f32_10v1, 2, 1)
#define LRA_INST_GROUP_3(S, I) LRA_F32_CFG(S, I, cfg_f32_4, 4, 2) LRA_F32_CFG(S, I, cfg_f32_5, 4, 2) LRA_F32_CFG(S, I, cfg_f32_6, 4, 2)
// (the many-bands shape of the run-ordered mel epilogue, n_fft = 512: lra_dispatch.h, MelManyCfgOf)
#define LRA_STFT_MELMANY(S, C) S(lra::C##_melmany, 4, 1, 0) S(lra::C##_melmany, 4, 2, 0) S(lra::C##_melmany, 4, 3, 0) S(lra::C##_melmany, 4, 1, 1) S(lra::C##_melmany, 4, 2, 1) S(lra::C##_melmany, 4, 3, 1)
#define LRA_INST_GROUP_4(S, I) LRA_F32_CFG(S, I, cfg_f32_7, 4, 2) LRA_F32_CFG(S, I, cfg_f32_8, 4, 2) LRA_STFT_REGRING(S, cfg_f32_7) LRA_STFT_REGRING(S, cfg_f32_8) LRA_STFT_MELMANY(S, cfg_f32_8)
#define LRA_INST_GROUP_5(S, I) LRA_F32_CFG(S, I, cfg_f32_9, 4, 2) LRA_F32_CFG(S, I, cfg_f32_11, 4, 2)
#define LRA_INST_GROUP_6(S, I) LRA_F32_CFG(S, I, cfg_f32_12, 4, 2) LRA_F32_CFG(S, I, cfg_f32_13, 4, 2) LRA_STFT_REGRING(S, cfg_f32_12) LRA_STFT_REGRING(S, cfg_f32_13)
#define LRA_INST_GROUP_7(S, I) LRA_F64_CFG(S, I, cfg_f64_4) LRA_F64_CFG(S, I, cfg_f64_5) LRA_F64_CFG(S, I, cfg_f64_6) LRA_F64_CFG(S, I, cfg_f64_7) LRA_F64_CFG(S, I, cfg_f64_8)
#define LRA_INST_GROUP_8(S, I) LRA_F64_CFG(S, I, cfg_f64_9) LRA_F64_CFG(S, I, cfg_f64_10) LRA_F64_CFG(S, I, cfg_f64_11) LRA_F64_CFG(S, I, cfg_f64_12)
// second-generation forward kernels: T(CFG, HD, MODE, PM)
#define LRA_STFT2_HD(T, C, HD) T(lra::C, HD, 0, 2) T(lra::C, HD, 1, 1) T(lra::C, HD, 1, 2) T(lra::C, HD, 1, 3)
#define LRA_STFT2_CFG(T, C) LRA_STFT2_HD(T, C, 1) LRA_STFT2_HD(T, C, 2) LRA_STFT2_HD(T, C, 4) LRA_STFT2_HD(T, C, 8)
#define LRA_STFT2_MEL_HD(T, C, HD) T(lra::C, HD, 4, 1) T(lra::C, HD, 4, 2) T(lra::C, HD, 4, 3)
#define LRA_STFT2_MEL(T, C) LRA_STFT2_MEL_HD(T, C, 1) LRA_STFT2_MEL_HD(T, C, 2) LRA_STFT2_MEL_HD(T, C, 4) LRA_STFT2_MEL_HD(T, C, 8)
#define LRA_INST2_GROUP_9(T, I) LRA_STFT2_CFG(T, cfg_f32_10) LRA_STFT2_MEL(T, cfg_f32_10_mel)
// ... and the inverse kernel on the ascending-radix configuration (Hermitian step fused into the first pass)
#define LRA_INST2_GROUP_10(T, I) LRA_STFT2_CFG(T, cfg_f32_9) LRA_STFT2_CFG(T, cfg_f32_11) I(lra::cfg_f32_10r, 8) I(lra::cfg_f32_10r, 4) I(lra::cfg_f32_10r, 2) I(lra::cfg_f32_10r, 1)
#define LRA_INST2_ALL(T, I) LRA_INST2_GROUP_9(T, I) LRA_INST2_GROUP_10(T, I) LRA_INST2_GROUP_12(T, I)
// producer / consumer mel kernels: P(CFG, HD, PM)
#define LRA_PC_HD(P, C, HD) P(lra::C, HD, 1) P(lra::C, HD, 2) P(lra::C, HD, 3)
#define LRA_INST3_GROUP_11(P) LRA_PC_HD(P, cfg_f32_10, 4) LRA_PC_HD(P, cfg_f32_10, 8) LRA_PC_HD(P, cfg_f32_10g, 4) LRA_PC_HD(P, cfg_f32_10g, 8)
#define LRA_INST3_ALL(P) LRA_INST3_GROUP_11(P)
// the radix 16-16-4 form of the second-generation forward kernels (variant 6)
#define LRA_INST2_GROUP_12(T, I) LRA_STFT2_CFG(T, cfg_f32_10g) I(lra::cfg_f32_10q, 8) I(lra::cfg_f32_10q, 4) I(lra::cfg_f32_10q, 2) I(lra::cfg_f32_10q, 1)
#define LRA_INST_NUM_GROUPS 13
#define LRA_INST_ALL(S, I)                                                                                                   \
    LRA_INST_GROUP_0(S, I) LRA_INST_GROUP_1(S, I) LRA_INST_GROUP_2(S, I) LRA_INST_GROUP_3(S, I) LRA_INST_GROUP_4(S, I)       \
    LRA_INST_GROUP_5(S, I) LRA_INST_GROUP_6(S, I) LRA_INST_GROUP_7(S, I) LRA_INST_GROUP_8(S, I)

#define LRA_STFT_SIG(C) lra::StftArgs<typename C::real>, const typename C::real*, void*
#define LRA_ISTFT_SIG(C) lra::IstftArgs<typename C::real>, const typename C::cplx*, const typename C::real*, typename C::real*
#define LRA_S_EXTERN(C, MODE, PM, RA) extern template __global__ void stft_kernel<C, MODE, PM, RA>(LRA_STFT_SIG(C));
#define LRA_I_EXTERN(C, HC) extern template __global__ void istft_kernel<C, HC>(LRA_ISTFT_SIG(C));
#define LRA_S_DEFINE(C, MODE, PM, RA) template __global__ void stft_kernel<C, MODE, PM, RA>(LRA_STFT_SIG(C));
#define LRA_I_DEFINE(C, HC) template __global__ void istft_kernel<C, HC>(LRA_ISTFT_SIG(C));

#define LRA_T_EXTERN(C, HD, MODE, PM) extern template __global__ void stft2_kernel<C, HD, MODE, PM>(LRA_STFT_SIG(C));
#define LRA_T_DEFINE(C, HD, MODE, PM) template __global__ void stft2_kernel<C, HD, MODE, PM>(LRA_STFT_SIG(C));

#define LRA_P_EXTERN(C, HD, PM) extern template __global__ void stft_pc_kernel<C, HD, PM>(LRA_STFT_SIG(C));
#define LRA_P_DEFINE(C, HD, PM) template __global__ void stft_pc_kernel<C, HD, PM>(LRA_STFT_SIG(C));

#if defined(LRA_FUSED_EXTERN) && !defined(LRA_PROBE_ONLY)
LRA_INST_ALL(LRA_S_EXTERN, LRA_I_EXTERN)
LRA_INST2_ALL(LRA_T_EXTERN, LRA_I_EXTERN)
LRA_INST3_ALL(LRA_P_EXTERN)
#endif
