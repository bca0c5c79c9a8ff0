// lra_mixed_inst.hip -- instances and launcher of the mixed-radix fused forward kernels (lra_mixed.h), a translation unit of its own so
// that it compiles side by side with lra_api.hip and the power-of-two instance groups (librosa_amd/build.py).
#include "lra_mixed.h"

#include "lra_mixed_launch.h"

namespace lra {
namespace mixed {
namespace {

template <class T, int N> hipError_t launch_n(int mode, const Args<T>& a, long long batch, hipStream_t stream) {
    const int lds = lds_bytes<T, N>() + (mode == MIXED_MEL && a.mel_nnz > 0 ? mel_lds_bytes<T>(a.n_mels, a.mel_nnz) : 0);
    const long long grid = batch * a.groups_per_clip;
    if (grid <= 0) return hipSuccess;
    if (grid > 0x7ffffff0LL) return hipErrorInvalidConfiguration;
    void (*kern)(Args<T>) = mode == MIXED_COMPLEX ? mixed_stft_kernel<T, N, MIXED_COMPLEX> : (mode == MIXED_POWER ? mixed_stft_kernel<T, N, MIXED_POWER> : mixed_stft_kernel<T, N, MIXED_MEL>);
    if (lds > 65536) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, stream, a);
    return hipGetLastError();
}

template <class T> hipError_t launch_t(int n_fft, int mode, const Args<T>& a, long long batch, hipStream_t stream) {
    switch (n_fft) {
#define LRA_MIXED_CASE(N) \
    case N: return launch_n<T, N>(mode, a, batch, stream);
        LRA_MIXED_SIZES(LRA_MIXED_CASE)
        LRA_MIXED_FWD_POW2(LRA_MIXED_CASE)
#undef LRA_MIXED_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

template <class T, int N> hipError_t launch_inv_n(const InvArgs<T>& a, long long batch, hipStream_t stream) {
    constexpr int lds = inv_lds_bytes<T, N>();
    const long long grid = batch * a.groups_per_clip;
    if (grid <= 0) return hipSuccess;
    if (grid > 0x7ffffff0LL) return hipErrorInvalidConfiguration;
    void (*kern)(InvArgs<T>) = mixed_istft_kernel<T, N>;
    if (lds > 65536) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, stream, a);
    return hipGetLastError();
}
template <class T> hipError_t launch_inv_t(int n_fft, const InvArgs<T>& a, long long batch, hipStream_t stream) {
    switch (n_fft) {
#define LRA_MIXED_CASE(N) \
    case N: return launch_inv_n<T, N>(a, batch, stream);
        LRA_MIXED_SIZES(LRA_MIXED_CASE)
        LRA_MIXED_INV_POW2(LRA_MIXED_CASE)
#undef LRA_MIXED_CASE
        default: return hipErrorInvalidValue;
    }
}

template <class T, int N> hipError_t launch_irfft_n(const IrArgs<T>& a, long long clips, hipStream_t stream) {
    constexpr int lds = lds_bytes<T, N>();
    const long long grid = clips * a.groups_per_clip;
    if (grid <= 0) return hipSuccess;
    if (grid > 0x7ffffff0LL) return hipErrorInvalidConfiguration;
    void (*kern)(IrArgs<T>) = mixed_irfft_kernel<T, N>;
    if (lds > 65536) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, stream, a);
    return hipGetLastError();
}
template <class T> hipError_t launch_irfft_t(int n_fft, const IrArgs<T>& a, long long clips, hipStream_t stream) {
    switch (n_fft) {
#define LRA_MIXED_CASE(N) \
    case N: return launch_irfft_n<T, N>(a, clips, stream);
        LRA_MIXED_SIZES(LRA_MIXED_CASE)
#undef LRA_MIXED_CASE
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_irfft_f32(int n_fft, const IrArgs<float>& a, long long clips, hipStream_t stream) { return launch_irfft_t<float>(n_fft, a, clips, stream); }
hipError_t launch_irfft_f64(int n_fft, const IrArgs<double>& a, long long clips, hipStream_t stream) { return launch_irfft_t<double>(n_fft, a, clips, stream); }

template <class T, int N> hipError_t launch_cqt_n(const CqtArgs<T>& a, long long batch, hipStream_t stream) {
    constexpr int lds = cqt_lds_bytes<T, N>();
    const long long grid = batch * a.groups_per_clip;
    if (grid <= 0) return hipSuccess;
    if (grid > 0x7ffffff0LL) return hipErrorInvalidConfiguration;
    void (*kern)(CqtArgs<T>) = mixed_cqt_kernel<T, N>;
    if (lds > 65536) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, stream, a);
    return hipGetLastError();
}
template <class T> hipError_t launch_cqt_t(int n_fft, const CqtArgs<T>& a, long long batch, hipStream_t stream) {
    switch (n_fft) {
#define LRA_MIXED_CASE(N) \
    case N: return launch_cqt_n<T, N>(a, batch, stream);
        LRA_CQT_SIZES(LRA_MIXED_CASE)
#undef LRA_MIXED_CASE
        default: return hipErrorInvalidValue;
    }
}
template <class T, int N> hipError_t launch_cqt_multi_n(const CqtMultiArgs<T>& m, hipStream_t stream) {
    constexpr int lds = cqt_lds_bytes<T, N>();
    const long long grid = (long long)m.blocks_per_octave * m.n_oct;
    if (grid <= 0) return hipSuccess;
    if (grid > 0x7ffffff0LL) return hipErrorInvalidConfiguration;
    void (*kern)(CqtMultiArgs<T>) = mixed_cqt_multi_kernel<T, N>;
    if (lds > 65536) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, stream, m);
    return hipGetLastError();
}
template <class T> hipError_t launch_cqt_multi_t(int n_fft, const CqtMultiArgs<T>& m, hipStream_t stream) {
    switch (n_fft) {
#define LRA_MIXED_CASE(N) \
    case N: return launch_cqt_multi_n<T, N>(m, stream);
        LRA_CQT_SIZES(LRA_MIXED_CASE)
#undef LRA_MIXED_CASE
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_cqt_multi_f32(int n_fft, const CqtMultiArgs<float>& m, hipStream_t stream) { return launch_cqt_multi_t<float>(n_fft, m, stream); }
hipError_t launch_cqt_multi_f64(int n_fft, const CqtMultiArgs<double>& m, hipStream_t stream) { return launch_cqt_multi_t<double>(n_fft, m, stream); }

int cqt_frames_per_group_of(int n_fft, int elem_bytes) {
    switch (n_fft) {
#define LRA_MIXED_CASE(N) \
    case N: return elem_bytes == 8 ? cqt_frames_per_group<double, N>() : cqt_frames_per_group<float, N>();
        LRA_CQT_SIZES(LRA_MIXED_CASE)
#undef LRA_MIXED_CASE
        default: return 0;
    }
}
hipError_t launch_cqt_f32(int n_fft, const CqtArgs<float>& a, long long batch, hipStream_t stream) { return launch_cqt_t<float>(n_fft, a, batch, stream); }
hipError_t launch_cqt_f64(int n_fft, const CqtArgs<double>& a, long long batch, hipStream_t stream) { return launch_cqt_t<double>(n_fft, a, batch, stream); }

int inv_frames_max_of(int n_fft, int elem_bytes) {
    switch (n_fft) {
#define LRA_MIXED_CASE(N) \
    case N: return elem_bytes == 8 ? inv_frames_max<double, N>() : inv_frames_max<float, N>();
        LRA_MIXED_SIZES(LRA_MIXED_CASE)
        LRA_MIXED_INV_POW2(LRA_MIXED_CASE)
#undef LRA_MIXED_CASE
        default: return 0;
    }
}
hipError_t launch_inv_f32(int n_fft, const InvArgs<float>& a, long long batch, hipStream_t stream) { return launch_inv_t<float>(n_fft, a, batch, stream); }
hipError_t launch_inv_f64(int n_fft, const InvArgs<double>& a, long long batch, hipStream_t stream) { return launch_inv_t<double>(n_fft, a, batch, stream); }

int frames_per_group_of(int n_fft, int elem_bytes) {
    switch (n_fft) {
#define LRA_MIXED_CASE(N) \
    case N: return elem_bytes == 8 ? frames_per_group<double, N>() : frames_per_group<float, N>();
        LRA_MIXED_SIZES(LRA_MIXED_CASE)
        LRA_MIXED_FWD_POW2(LRA_MIXED_CASE)
#undef LRA_MIXED_CASE
        default: return 0;
    }
}

hipError_t launch_f32(int n_fft, int mode, const Args<float>& a, long long batch, hipStream_t stream) { return launch_t<float>(n_fft, mode, a, batch, stream); }
hipError_t launch_f64(int n_fft, int mode, const Args<double>& a, long long batch, hipStream_t stream) { return launch_t<double>(n_fft, mode, a, batch, stream); }

}  // namespace mixed
}  // namespace lra
