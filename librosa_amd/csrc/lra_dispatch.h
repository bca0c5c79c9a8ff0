// lra_dispatch.h -- n_fft -> compile-time FFT configuration, shared by the gfx950 library
// (lra_api.hip) and the CPU thread simulator (tests/hostsim).
#pragma once

#include "lra_kernels.h"
#include "lra_kernels2.h"

namespace lra {

constexpr int kMinLogM = 4;     // n_fft = 32
constexpr int kMaxLogM = 13;    // n_fft = 16384 (M = 8192 complex: 68 KiB of LDS per frame in f32)
constexpr int kMaxLogM64 = 12;  // f64: n_fft <= 8192
constexpr int kNumVariants = 8; // tuning variants exist for f32 n_fft = 2048 only

// Compile-time configuration per (dtype, log2 M, variant).  Variant 0 is the default:
//   f32: 16 complex points per thread, 256-VGPR budget (2 waves/SIMD), window/twiddle values kept
//        in registers across the frame loop for n_fft <= 2048 (no table traffic in steady state);
//   f64:  8 complex points per thread, 256-VGPR budget, tables re-read from L1/L2 every frame.
// A workgroup is ONE wave64 (several frame slots when n_fft < 2048) unless a frame needs more threads:
// slots are private pipelines, so small workgroups only add scheduling freedom and keep the LDS
// footprint (frame area + PCM ring + mel tile, ~17-19 KiB per slot at n_fft = 2048) granular.
// Variants 1 and 4 (f32, n_fft = 2048) split a frame over two waves (8 points per thread) for higher
// occupancy; variants 2 and 3 were retired (always slower) and now alias variant 0; variant 5 is the inverse kernel's
// ascending-radix configuration (the forward launchers reject it).
template <class T, int L, int VAR> struct CfgSel {
    using type = FftCfg<L, 4, T, 64, 2, true>;
};
template <int L, int VAR> struct CfgSel<double, L, VAR> {
    using type = FftCfg<L, 3, double, 64, 2, false>;
};
template <> struct CfgSel<float, 10, 1> {
    using type = FftCfg<10, 3, float, 128, 4, false>;
};
template <> struct CfgSel<float, 10, 4> {
    using type = FftCfg<10, 3, float, 128, 3, true>;  // two waves per frame, tables in registers, 3 waves/SIMD
};
template <> struct CfgSel<float, 10, 5> {
    using type = FftCfg<10, 4, float, 64, 2, true, true>;  // inverse kernel only: radices 8, 8, 16 (Hermitian step fused into the first pass)
};

template <> struct CfgSel<float, 10, 6> {
    using type = FftCfg<10, 4, float, 64, 2, true, false, 1>;  // radices 16, 16, 4: adjacent bins per thread (lra_kernels2.h, third form); second-generation forward kernels only
};

template <> struct CfgSel<float, 10, 7> {
    using type = FftCfg<10, 4, float, 64, 2, true, true, 1>;  // inverse kernel only: radices 4, 16, 16 -- four first-pass butterflies per thread, 16-byte spectrum loads (round 6)
};

// true when n_fft is a power of two handled by the fused LDS kernels
inline bool pow2_supported(int n_fft, bool f64) {
    if (n_fft <= 0 || (n_fft & (n_fft - 1))) return false;
    int logn = 0;
    while ((1 << logn) < n_fft) ++logn;
    const int logm = logn - 1;
    return logm >= kMinLogM && logm <= (f64 ? kMaxLogM64 : kMaxLogM);
}

inline int log2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// Calls f.template operator()<Cfg>() for the configuration of (T, logm, variant); returns false if
// unsupported.
template <class T, class F> inline bool dispatch_logm(int logm, int variant, F&& f) {
#ifdef LRA_PROBE_ONLY  // development builds (scripts/probe_build.sh): one f32 size only (n_fft = 2048 unless -DLRA_PROBE_LOGM=..), compiles in seconds
#ifndef LRA_PROBE_LOGM
#define LRA_PROBE_LOGM 10
#endif
    if (logm != LRA_PROBE_LOGM || sizeof(T) != 4) return false;
#endif
#if defined(LRA_PROBE_ONLY) && defined(LRA_PROBE_CFG)  // e.g. -DLRA_PROBE_LOGM=12 "-DLRA_PROBE_CFG=FftCfg<12,4,T,64,4,false>"
    (void)variant;
    f.template operator()<LRA_PROBE_CFG>();
    return true;
#elif defined(LRA_PROBE_ONLY) && LRA_PROBE_LOGM != 10
    (void)variant;
    f.template operator()<typename CfgSel<T, LRA_PROBE_LOGM, 0>::type>();
    return true;
#else
    switch (logm) {
#define LRA_CASE(L) \
    case L: f.template operator()<typename CfgSel<T, L, 0>::type>(); return true;
#ifndef LRA_PROBE_ONLY
        LRA_CASE(4) LRA_CASE(5) LRA_CASE(6) LRA_CASE(7) LRA_CASE(8) LRA_CASE(9) LRA_CASE(11) LRA_CASE(12)
#endif
#undef LRA_CASE
        case 10:
            if constexpr (sizeof(T) == 4) {
                if (variant == 1) { f.template operator()<typename CfgSel<T, 10, 1>::type>(); return true; }
                if (variant == 4) { f.template operator()<typename CfgSel<T, 10, 4>::type>(); return true; }
                if (variant == 5) { f.template operator()<typename CfgSel<T, 10, 5>::type>(); return true; }
                if (variant == 6) { f.template operator()<typename CfgSel<T, 10, 6>::type>(); return true; }
                if (variant == 7) { f.template operator()<typename CfgSel<T, 10, 7>::type>(); return true; }
            }
            f.template operator()<typename CfgSel<T, 10, 0>::type>();
            return true;
#ifndef LRA_PROBE_ONLY
        case 13:
            if constexpr (sizeof(T) == 4) { f.template operator()<typename CfgSel<T, 13, 0>::type>(); return true; }
            return false;
#endif
        default: return false;
    }
#endif
}

// workgroup configuration of the two-slope mel kernel: its filter tables are shared across the slots of a
// larger workgroup.  When a frame fits one wave: 256 threads = 4 slots (two such workgroups share a CU and
// drift apart; measured +2 % over one 512-thread workgroup whose 8 waves start every phase together)
template <class Cfg> struct MelCfgOf {
#ifndef LRA_MELNT
#define LRA_MELNT 256
#endif
    static constexpr int MELNT = Cfg::TF <= 64 ? LRA_MELNT : (2 * Cfg::TF <= 1024 ? 2 * Cfg::TF : Cfg::TF);
    using type = typename Cfg::template with_nt<MELNT>;
};
// ... and the shape for MANY bands per frame thread (round 5): 128 threads = eight slots of 16 threads.  The thread count doubles as the tag by which
// lra_kernels.h (melr_nb / melr_ph) sizes the per-thread band state: eight bands and one hoisted piece per list instead of four and three -- n_fft = 512 with more
// than 100 bands, whose pair segments are a couple of bins wide (same box: 128 bands 1.54 -> 1.04 ms; with 80 bands or fewer the four-band form is faster by a third).
constexpr int MELNT_MANY = 128;
template <class Cfg> struct MelManyCfgOf {
    using type = typename Cfg::template with_nt<MELNT_MANY>;
};
template <class Cfg> constexpr bool mel_many_applies() { return Cfg::TF == 16 && sizeof(typename Cfg::real) == 4; }


}  // namespace lra
