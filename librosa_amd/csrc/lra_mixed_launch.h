// lra_mixed_launch.h -- what lra_api.hip sees of the mixed-radix kernels: the list of frame lengths they are instantiated for and the
// launcher (defined in lra_mixed_inst.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "lra_mixed.h"

// frame lengths n_fft = 2 M, M = 2^a 3^b 5^c 7^d, served by one fused launch (everything else that is not a power of two keeps the rocFFT
// path): the 15 / 20 / 25 / 30 / 40 / 50 / 60 ms frames of 8, 16, 22.05 (rounded), 32 and 48 kHz front ends and their doubles; round 6: radix 7 for the
// 20 / 40 / 60 / 80 ms frames of 44.1 kHz (882 = 2 3^2 7^2, 1764, 2646, 3528 samples)
#define LRA_MIXED_SIZES(X) X(160) X(200) X(240) X(320) X(400) X(480) X(600) X(640) X(720) X(800) X(882) X(960) X(1000) X(1200) X(1280) X(1440) X(1600) X(1764) X(1920) X(2000) X(2400) X(2646) X(3200) X(3528) X(4800)

// INVERSE only: power-of-two frame lengths whose hop does not divide them the way the register-tiled inverse wants (n_fft / {2, 4, 8, 16}) --
// n_fft = 512 with hop 160 (25 ms windows padded to 512 at 16 kHz), 1024 with 441, ... -- take the fused gather kernel of this file instead of
// spec_pack + rocFFT + overlap-add (larger frames hold too few frames per workgroup for it: lra_api.hip's own / halo rule)
#define LRA_MIXED_INV_POW2(X) X(256) X(512) X(1024)

// FORWARD, fused mel only: powers of two whose register-tiled mel kernel is slower than this file's (n_fft 256: eight-thread frames leave the band combine to too
// few lanes -- 256 / 64 / 80 bands over 256 x 30 s: 2.02 ms against 1.04 for the bare transform; ctx option mixed_pow2_mel)
#define LRA_MIXED_FWD_POW2(X) X(128) X(256)

// the constant-Q octave kernel's frame lengths (filters.wavelet pads every octave's filters to a power of two)
#define LRA_CQT_SIZES(X) X(32) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096)

namespace lra {
namespace mixed {
constexpr bool in_cqt_size_list(int n_fft) {
#define LRA_MIXED_CASE(N) \
    if (n_fft == N) return true;
    LRA_CQT_SIZES(LRA_MIXED_CASE)
#undef LRA_MIXED_CASE
    return false;
}
constexpr bool in_size_list(int n_fft) {
#define LRA_MIXED_CASE(N) \
    if (n_fft == N) return true;
    LRA_MIXED_SIZES(LRA_MIXED_CASE)
#undef LRA_MIXED_CASE
    return false;
}
constexpr bool in_fwd_pow2_list(int n_fft) {
#define LRA_MIXED_CASE(N) \
    if (n_fft == N) return true;
    LRA_MIXED_FWD_POW2(LRA_MIXED_CASE)
#undef LRA_MIXED_CASE
    return false;
}
constexpr bool in_inv_pow2_list(int n_fft) {
#define LRA_MIXED_CASE(N) \
    if (n_fft == N) return true;
    LRA_MIXED_INV_POW2(LRA_MIXED_CASE)
#undef LRA_MIXED_CASE
    return false;
}
#define LRA_MIXED_CHECK(N) static_assert(supported(N), "n_fft = " #N ": n_fft / 2 must factor into 2, 3, 5 and 7");
LRA_MIXED_SIZES(LRA_MIXED_CHECK)
LRA_MIXED_INV_POW2(LRA_MIXED_CHECK)
LRA_MIXED_FWD_POW2(LRA_MIXED_CHECK)
#undef LRA_MIXED_CHECK
int frames_per_group_of(int n_fft, int elem_bytes);
hipError_t launch_f32(int n_fft, int mode, const Args<float>& a, long long batch, hipStream_t stream);
hipError_t launch_f64(int n_fft, int mode, const Args<double>& a, long long batch, hipStream_t stream);
int cqt_frames_per_group_of(int n_fft, int elem_bytes);
hipError_t launch_cqt_f32(int n_fft, const CqtArgs<float>& a, long long batch, hipStream_t stream);
hipError_t launch_cqt_f64(int n_fft, const CqtArgs<double>& a, long long batch, hipStream_t stream);
hipError_t launch_cqt_multi_f32(int n_fft, const CqtMultiArgs<float>& m, hipStream_t stream);
hipError_t launch_cqt_multi_f64(int n_fft, const CqtMultiArgs<double>& m, hipStream_t stream);
hipError_t launch_irfft_f32(int n_fft, const IrArgs<float>& a, long long clips, hipStream_t stream);   // the inverse real transform alone (frames -> HBM), forward geometry
hipError_t launch_irfft_f64(int n_fft, const IrArgs<double>& a, long long clips, hipStream_t stream);
int inv_frames_max_of(int n_fft, int elem_bytes);  // frames (own + halo) the inverse kernel holds per workgroup
hipError_t launch_inv_f32(int n_fft, const InvArgs<float>& a, long long batch, hipStream_t stream);
hipError_t launch_inv_f64(int n_fft, const InvArgs<double>& a, long long batch, hipStream_t stream);
}  // namespace mixed
}  // namespace lra
