// postsim.cpp -- runs the PCEN / band max-filter kernel bodies of librosa_amd/csrc/lra_pcen.h and the constant-Q kernels of lra_cqt.h on
// host threads.
//
// TEST INFRASTRUCTURE ONLY.  Built by tests/test_hostsim.py (g++ -DLRA_POSTSIM -pthread) into tests/hostsim/_postsim.so.  One OS
// thread per lane of a workgroup, __syncthreads() is a barrier across them, __shared__ is a static the lanes share; workgroups run
// one after the other.  Never linked into, imported by, or used as a fallback for the product library.
#define LRA_POSTSIM 1
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

struct SimIdx { unsigned x = 0, y = 0, z = 0; };
static thread_local SimIdx threadIdx;
static thread_local SimIdx blockIdx;

namespace {
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int n = 0, waiting = 0;
    unsigned long long gen = 0;
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        const unsigned long long g = gen;
        if (++waiting == n) {
            waiting = 0;
            ++gen;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g; });
        }
    }
};
Barrier g_barrier;
}  // namespace
static inline void __syncthreads() { g_barrier.wait(); }

#define __global__
#define __device__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
using std::exp;
using std::expm1;
using std::log;
using std::log1p;
using std::hypot;
using std::pow;
using std::sqrt;

alignas(16) static unsigned char g_postsim_dyn_lds[160 * 1024];  // the dynamic LDS of the workgroup being run

#include "../../librosa_amd/csrc/lra_pcen.h"
#include "../../librosa_amd/csrc/lra_cqt.h"
#include "../../librosa_amd/csrc/lra_hpss.h"
#include "../../librosa_amd/csrc/lra_mixed.h"
#include "../../librosa_amd/csrc/lra_rng.h"

namespace {
template <class F> void run_grid(unsigned grid, unsigned block, F body) {
    g_barrier.n = (int)block;
    for (unsigned b = 0; b < grid; ++b) {
        std::vector<std::thread> lanes;
        for (unsigned t = 0; t < block; ++t)
            lanes.emplace_back([=] {
                threadIdx.x = t;
                blockIdx.x = b;
                body();
            });
        for (auto& l : lanes) l.join();
    }
}
// kernels without __syncthreads: the lanes of a workgroup one after the other on the calling thread
template <class F> void run_grid_serial(unsigned grid, unsigned block, F body) {
    for (unsigned b = 0; b < grid; ++b)
        for (unsigned t = 0; t < block; ++t) {
            threadIdx.x = t;
            blockIdx.x = b;
            body();
        }
}
}  // namespace

namespace {
template <class T> void sim_hpss(const void* mag, const void* D, void* out_h, void* out_p, const lra::HpssArgs& a, unsigned grid) {
    const lra::HpssPlan plan = lra::hpss_plan(a, (int)sizeof(T));  // the same selection as hpss_launch (lra_api.hip)
    const unsigned tgrid = (unsigned)((lra::hpss_tiles(a) + 255) / 256);
    if (plan.tile_slots == 32 && plan.fixed_win) {
        run_grid(tgrid, 256, [=] { lra::hpss_tile_kernel<T, 32, lra::kHpssFixedWin>((const T*)mag, (const lra::HpssCplx<T>*)D, out_h, out_p, a); });
        return;
    }
    if (plan.tile_slots == 32) {
        run_grid(tgrid, 256, [=] { lra::hpss_tile_kernel<T, 32, 0>((const T*)mag, (const lra::HpssCplx<T>*)D, out_h, out_p, a); });
        return;
    }
    if constexpr (sizeof(T) == 4) {
        if (plan.tile_slots == 64) {
            run_grid(tgrid, 256, [=] { lra::hpss_tile_kernel<T, 64, 0>((const T*)mag, (const lra::HpssCplx<T>*)D, out_h, out_p, a); });
            return;
        }
    }
    if (plan.element_slots == 32) {
        run_grid_serial(grid, 256, [=] { lra::hpss_kernel<T, 32>((const T*)mag, (const lra::HpssCplx<T>*)D, out_h, out_p, a); });
        return;
    }
    if constexpr (sizeof(T) == 4) {
        if (plan.element_slots == 64) {
            run_grid_serial(grid, 256, [=] { lra::hpss_kernel<T, 64>((const T*)mag, (const lra::HpssCplx<T>*)D, out_h, out_p, a); });
            return;
        }
    }
    run_grid_serial(grid, 256, [=] { lra::hpss_kernel<T, 0>((const T*)mag, (const lra::HpssCplx<T>*)D, out_h, out_p, a); });
}
}  // namespace

extern "C" {
// the launch of lra_pcen_exec (lra_api.hip), same argument preparation
int postsim_pcen(const void* S, const void* ref, double* out, long long rows, long long n_frames, int is_f64, double b, double gain, double bias, double power, double eps, const double* zi,
                 double zi_scalar, double* zf) {
    lra::PcenArgs p;
    p.b = b;
    p.a1 = b - 1.0;
    p.zi_scalar = zi_scalar;
    p.neg_gain = -gain;
    p.log_eps = std::log(eps);
    p.eps = eps;
    p.power = power;
    p.bias = bias;
    p.bias_pow = std::pow(bias, power);
    p.mode = power == 0 ? 0 : (bias == 0 ? 1 : 2);
    const unsigned grid = (unsigned)((rows + lra::kPcenRows - 1) / lra::kPcenRows);
    if (is_f64)
        run_grid(grid, 64, [=] { lra::pcen_kernel<double>((const double*)S, (const double*)(ref ? ref : S), out, rows, n_frames, p, zi, zf); });
    else
        run_grid(grid, 64, [=] { lra::pcen_kernel<float>((const float*)S, (const float*)(ref ? ref : S), out, rows, n_frames, p, zi, zf); });
    return 0;
}

int postsim_maxfilter(const void* S, void* out, long long outer, int n_bands, long long inner, int size, int is_f64) {
    const long long count = outer * n_bands * inner;
    const unsigned grid = (unsigned)((count + 255) / 256);
    if (is_f64)
        run_grid_serial(grid, 256, [=] { lra::maxfilter_bands_kernel<double>((const double*)S, (double*)out, outer, n_bands, inner, size); });
    else
        run_grid_serial(grid, 256, [=] { lra::maxfilter_bands_kernel<float>((const float*)S, (float*)out, outer, n_bands, inner, size); });
    return 0;
}

// the launch of lra_resample_poly_exec (lra_api.hip); up == 1 goes to the decimators below, as there
int postsim_fir_decimate(const void* x, void* out, long long batch, long long n_in, long long n_out, const void* taps, int n_taps, int down, int first, double div, double mul, int is_f64);
int postsim_resample_poly(const void* x, void* out, long long batch, long long n_in, long long n_out, const void* taps, int n_taps, int up, int down, int first, double div, double mul, int is_f64) {
    if (up == 1) return postsim_fir_decimate(x, out, batch, n_in, n_out, taps, n_taps, down, first, div, mul, is_f64);
    const unsigned grid = (unsigned)((batch * n_out + 255) / 256);
    if (is_f64)
        run_grid_serial(grid, 256, [=] { lra::resample_poly_kernel<double>((const double*)x, (double*)out, (const double*)taps, batch, n_in, n_out, n_taps, up, down, first, div, mul); });
    else
        run_grid_serial(grid, 256, [=] { lra::resample_poly_kernel<float>((const float*)x, (float*)out, (const float*)taps, batch, n_in, n_out, n_taps, up, down, first, div, mul); });
    return 0;
}

// the launches of lra_fir_decimate_exec / lra_cqt_project_exec (lra_api.hip)
int postsim_fir_decimate(const void* x, void* out, long long batch, long long n_in, long long n_out, const void* taps, int n_taps, int down, int first, double div, double mul, int is_f64) {
    // (as lra_fir_decimate_exec, lra_api.hip: four outputs per thread for big jobs whose span fits, else one; the direct kernel for spans beyond the LDS)
    const size_t elem = is_f64 ? 8 : 4;
    const bool four = n_out >= 8192 && ((size_t)(1023 * (long long)down + n_taps)) * elem <= 32 * 1024;
    const int opt = four ? 4 : 1;
    const int blocks_per_clip = (int)((n_out + 256 * opt - 1) / (256 * opt));
    const size_t span2 = (size_t)1023 * 2 + n_taps;
    if (four && down == 2 && (span2 + span2 / 8 + 1) * 2 * elem <= 64 * 1024) {
        const unsigned grid2 = (unsigned)(blocks_per_clip * batch);
        if (is_f64) run_grid(grid2, 256, [=] { lra::fir_halve4_kernel<double>((const double*)x, (double*)out, (const double*)taps, n_in, n_out, blocks_per_clip, n_taps, first, div, mul); });
        else run_grid(grid2, 256, [=] { lra::fir_halve4_kernel<float>((const float*)x, (float*)out, (const float*)taps, n_in, n_out, blocks_per_clip, n_taps, first, div, mul); });
        return 0;
    }
    if ((size_t)((256 * opt - 1) * down + n_taps) * elem > 64 * 1024) {
        const unsigned dgrid = (unsigned)((batch * n_out + 255) / 256);
        if (is_f64)
            run_grid_serial(dgrid, 256, [=] { lra::fir_decimate_direct_kernel<double>((const double*)x, (double*)out, (const double*)taps, batch, n_in, n_out, n_taps, down, first, div, mul); });
        else
            run_grid_serial(dgrid, 256, [=] { lra::fir_decimate_direct_kernel<float>((const float*)x, (float*)out, (const float*)taps, batch, n_in, n_out, n_taps, down, first, div, mul); });
        return 0;
    }
    const unsigned grid = (unsigned)(blocks_per_clip * batch);
#define SIM_FIR(T, OPT) run_grid(grid, 256, [=] { lra::fir_decimate_kernel<T, OPT>((const T*)x, (T*)out, (const T*)taps, n_in, n_out, blocks_per_clip, n_taps, down, first, div, mul); })
    if (is_f64) { if (four) SIM_FIR(double, 4); else SIM_FIR(double, 1); }
    else { if (four) SIM_FIR(float, 4); else SIM_FIR(float, 1); }
#undef SIM_FIR
    return 0;
}

int postsim_cqt_project(const void* D, void* out, const int* row_ptr, const int* col, const void* val, const double* sqrt_len, long long batch, long long frames_in, int n_bins,
                        long long n_frames, int n_total, int bin0, int row0, int n_rows, int is_f64) {
    const unsigned grid = (unsigned)((batch * n_frames * n_rows + 255) / 256);
    if (is_f64)
        run_grid_serial(grid, 256, [=] {
            lra::cqt_project_kernel<double>((const lra::CqtCplx<double>*)D, (lra::CqtCplx<double>*)out, row_ptr, col, (const lra::CqtCplx<double>*)val, sqrt_len, batch, frames_in, n_bins,
                                            n_frames, n_total, bin0, row0, n_rows);
        });
    else
        run_grid_serial(grid, 256, [=] {
            lra::cqt_project_kernel<float>((const lra::CqtCplx<float>*)D, (lra::CqtCplx<float>*)out, row_ptr, col, (const lra::CqtCplx<float>*)val, sqrt_len, batch, frames_in, n_bins, n_frames,
                                           n_total, bin0, row0, n_rows);
        });
    return 0;
}

// the launches of lra_magnitude_exec / lra_hpss_exec (lra_api.hip)
int postsim_magnitude(const void* D, void* mag, long long count, int is_f64) {
    const unsigned grid = (unsigned)((count + 255) / 256);
    if (is_f64) run_grid_serial(grid, 256, [=] { lra::magnitude_kernel<double>((const lra::HpssCplx<double>*)D, (double*)mag, count); });
    else run_grid_serial(grid, 256, [=] { lra::magnitude_kernel<float>((const lra::HpssCplx<float>*)D, (float*)mag, count); });
    return 0;
}

// the launches of lra_pcg64_random_exec / lra_griffinlim_init_pcg64 (lra_api.hip), same argument preparation
int postsim_pcg64_random(const unsigned long long* state4, unsigned long long offset, double* out, long long count) {
    const lra::rng::Pcg64 g{state4[0], state4[1], state4[2], state4[3]};
    const long long runs = (count + lra::rng::kRunLength - 1) / lra::rng::kRunLength;
    run_grid_serial((unsigned)((runs + 255) / 256), 256, [=] { lra::rng::pcg64_uniform_kernel(g, offset, out, count); });
    return 0;
}

int postsim_griffinlim_init_pcg64(const unsigned long long* state4, const void* S, void* angles, long long batch, int n_bins, long long n_frames, int seg, int is_f64) {
    const lra::rng::Pcg64 g{state4[0], state4[1], state4[2], state4[3]};
    const int bin_blocks = (n_bins + 255) / 256;
    const unsigned grid = (unsigned)(batch * bin_blocks * ((n_frames + seg - 1) / seg));
    if (is_f64) run_grid_serial(grid, 256, [=] { lra::rng::griffinlim_init_pcg64_kernel<double>(g, (const double*)S, (lra::rng::RngCplx<double>*)angles, batch, n_bins, n_frames, seg, bin_blocks); });
    else run_grid_serial(grid, 256, [=] { lra::rng::griffinlim_init_pcg64_kernel<float>(g, (const float*)S, (lra::rng::RngCplx<float>*)angles, batch, n_bins, n_frames, seg, bin_blocks); });
    return 0;
}

int postsim_magphase(const void* D, int is_complex, void* mag, void* phase, long long count, double power, int is_f64) {
    const unsigned grid = (unsigned)((count + 255) / 256);
    if (is_f64) run_grid_serial(grid, 256, [=] { lra::magphase_kernel<double>(D, is_complex, (double*)mag, (lra::HpssCplx<double>*)phase, count, power); });
    else run_grid_serial(grid, 256, [=] { lra::magphase_kernel<float>(D, is_complex, (float*)mag, (lra::HpssCplx<float>*)phase, count, (float)power); });
    return 0;
}

int postsim_hpss(const void* mag, const void* D, void* out_h, void* out_p, long long batch, long long n_frames, int n_bins, int win_harm, int win_perc, double power, double margin_harm,
                 double margin_perc, int want_mask, int is_f64) {
    lra::HpssArgs a;
    a.batch = batch;
    a.n_frames = n_frames;
    a.n_bins = n_bins;
    a.win_harm = win_harm;
    a.win_perc = win_perc;
    a.hard = std::isinf(power) ? 1 : 0;
    a.power = a.hard ? 1.0 : power;
    a.margin_harm = margin_harm;
    a.margin_perc = margin_perc;
    a.want_mask = want_mask ? 1 : 0;
    const unsigned grid = (unsigned)((batch * n_frames * n_bins + 255) / 256);
    if (is_f64) sim_hpss<double>(mag, D, out_h, out_p, a, grid);
    else sim_hpss<float>(mag, D, out_h, out_p, a, grid);
    return 0;
}
}

// ---- mixed-radix fused forward kernel (librosa_amd/csrc/lra_mixed.h), the launch of stft_run's mixed branch (lra_api.hip) ----------------------
namespace {
template <class T, int N> int sim_mixed(int mode, lra::mixed::Args<T> a, long long batch) {
    using namespace lra::mixed;
    constexpr int F = frames_per_group<T, N>();
    static_assert(lds_bytes<T, N>() <= (int)sizeof(g_postsim_dyn_lds), "simulated LDS too small");
    a.groups_per_clip = (a.n_frames + F - 1) / F;
    const unsigned grid = (unsigned)(batch * a.groups_per_clip);
    if (mode == MIXED_COMPLEX) run_grid(grid, NT, [=] { mixed_stft_kernel<T, N, MIXED_COMPLEX>(a); });
    else if (mode == MIXED_POWER) run_grid(grid, NT, [=] { mixed_stft_kernel<T, N, MIXED_POWER>(a); });
    else run_grid(grid, NT, [=] { mixed_stft_kernel<T, N, MIXED_MEL>(a); });
    return 0;
}
template <class T> int sim_mixed_n(int n_fft, int mode, const lra::mixed::Args<T>& a, long long batch) {
    switch (n_fft) {  // (a few of the product's sizes: every radix, one and several passes of each)
        case 160: return sim_mixed<T, 160>(mode, a, batch);
        case 240: return sim_mixed<T, 240>(mode, a, batch);
        case 400: return sim_mixed<T, 400>(mode, a, batch);
        case 480: return sim_mixed<T, 480>(mode, a, batch);
        case 1000: return sim_mixed<T, 1000>(mode, a, batch);
        case 1200: return sim_mixed<T, 1200>(mode, a, batch);
        case 1280: return sim_mixed<T, 1280>(mode, a, batch);
        case 882: return sim_mixed<T, 882>(mode, a, batch);    // 3 x 3 x 7 x 7 (round 6: radix 7)
        default: return 1;
    }
}
}  // namespace

extern "C" int postsim_mixed_stft(int n_fft, int mode, int is_f64, const void* y, long long batch, long long n, int n_frames, int hop, int pad, int pad_mode, const void* win,
                                  const void* tw_m, const void* tw_n, void* out, int power_mode, double power, const int* mel_c0, const int* mel_len, const int* mel_off,
                                  const void* mel_val, int n_mels) {
    auto fill = [&](auto& a, auto tag) {
        using T = decltype(tag);
        a.y = (const T*)y; a.y_stride = n; a.n = n; a.n_frames = n_frames; a.hop = hop; a.pad = pad; a.pad_mode = pad_mode;
        a.win = (const T*)win; a.tw_m = (const lra::mixed::cpx<T>*)tw_m; a.tw_n = (const lra::mixed::cpx<T>*)tw_n;
        a.D = (lra::mixed::cpx<T>*)out; a.S = (T*)out; a.Mel = (T*)out;
        a.power_mode = power_mode; a.power = (T)power;
        a.mel_c0 = mel_c0; a.mel_len = mel_len; a.mel_off = mel_off; a.mel_val = (const T*)mel_val; a.n_mels = n_mels;
        int nnz = 0;  // as stft_run (lra_api.hip): the band table is staged in LDS where it fits
        for (int m = 0; mel_off && m < n_mels; ++m) nnz = mel_off[m] + mel_len[m] > nnz ? mel_off[m] + mel_len[m] : nnz;
        a.mel_nnz = lra::mixed::mel_lds_fits(n_mels, nnz) ? nnz : 0;
    };
    if (is_f64) {
        lra::mixed::Args<double> a = lra::mixed::Args<double>();
        fill(a, double());
        return sim_mixed_n<double>(n_fft, mode, a, batch);
    }
    lra::mixed::Args<float> a = lra::mixed::Args<float>();
    fill(a, float());
    return sim_mixed_n<float>(n_fft, mode, a, batch);
}

namespace {
template <class T, int N> int sim_mixed_inv(lra::mixed::InvArgs<T> a, long long batch) {
    using namespace lra::mixed;
    static_assert(inv_lds_bytes<T, N>() <= (int)sizeof(g_postsim_dyn_lds), "simulated LDS too small");
    const int fmax = inv_frames_max<T, N>();
    a.halo = (N + a.hop - 1) / a.hop - 1;
    if (fmax - a.halo < 1) return 2;  // (the product also asks for own >= 2 halo: a performance rule; the simulator takes every shape that fits)
    a.group_hops = fmax - a.halo;
    a.groups_per_clip = (a.n_used + a.group_hops - 1) / a.group_hops;
    run_grid((unsigned)(batch * a.groups_per_clip), NT, [=] { mixed_istft_kernel<T, N>(a); });
    return 0;
}
template <class T> int sim_mixed_inv_n(int n_fft, const lra::mixed::InvArgs<T>& a, long long batch) {
    switch (n_fft) {
        case 160: return sim_mixed_inv<T, 160>(a, batch);
        case 240: return sim_mixed_inv<T, 240>(a, batch);
        case 400: return sim_mixed_inv<T, 400>(a, batch);
        case 480: return sim_mixed_inv<T, 480>(a, batch);
        case 1000: return sim_mixed_inv<T, 1000>(a, batch);
        case 1200: return sim_mixed_inv<T, 1200>(a, batch);
        case 882: return sim_mixed_inv<T, 882>(a, batch);
        case 256: return sim_mixed_inv<T, 256>(a, batch);   // LRA_MIXED_INV_POW2: powers of two with a hop the register-tiled inverse does not take
        case 512: return sim_mixed_inv<T, 512>(a, batch);
        case 1024: return sim_mixed_inv<T, 1024>(a, batch);
        default: return 1;
    }
}
}  // namespace

// D: [batch][n_frames_total][M + 1]; norm: the normalisation factors (1 / wss where wss > tiny, else 1); y arrives as the caller filled it
extern "C" int postsim_mixed_istft(int n_fft, int is_f64, const void* D, long long batch, int n_frames_total, int n_used, int hop, int drop, const void* win_scaled, const void* tw_m,
                                   const void* tw_n, const void* norm, void* y, long long out_len) {
    auto fill = [&](auto& a, auto tag) {
        using T = decltype(tag);
        const int M = n_fft / 2;
        a.D = (const lra::mixed::cpx<T>*)D; a.d_frame_stride = M + 1; a.d_batch_stride = (long long)n_frames_total * (M + 1);
        a.n_used = n_used; a.hop = hop; a.drop = drop; a.win_scaled = (const T*)win_scaled;
        a.tw_m = (const lra::mixed::cpx<T>*)tw_m; a.tw_n = (const lra::mixed::cpx<T>*)tw_n; a.norm = (const T*)norm;
        a.y = (T*)y; a.y_stride = out_len; a.out_len = out_len;
    };
    if (is_f64) {
        lra::mixed::InvArgs<double> a = lra::mixed::InvArgs<double>();
        fill(a, double());
        return sim_mixed_inv_n<double>(n_fft, a, batch);
    }
    lra::mixed::InvArgs<float> a = lra::mixed::InvArgs<float>();
    fill(a, float());
    return sim_mixed_inv_n<float>(n_fft, a, batch);
}

// the inverse real transform alone (mixed_irfft_kernel: what lra_api.hip's general inverse path launches for listed lengths too long for the gather kernel)
namespace {
template <class T, int N> int sim_mixed_irfft(lra::mixed::IrArgs<T> a, long long batch) {
    using namespace lra::mixed;
    constexpr int F = frames_per_group<T, N>();
    a.groups_per_clip = (a.n_used + F - 1) / F;
    run_grid((unsigned)(batch * a.groups_per_clip), NT, [=] { mixed_irfft_kernel<T, N>(a); });
    return 0;
}
template <class T> int sim_mixed_irfft_n(int n_fft, const lra::mixed::IrArgs<T>& a, long long batch) {
    switch (n_fft) {
        case 400: return sim_mixed_irfft<T, 400>(a, batch);
        case 882: return sim_mixed_irfft<T, 882>(a, batch);
        case 1200: return sim_mixed_irfft<T, 1200>(a, batch);
        case 1280: return sim_mixed_irfft<T, 1280>(a, batch);
        default: return 1;
    }
}
}  // namespace
extern "C" int postsim_mixed_irfft(int n_fft, int is_f64, const void* D, long long batch, int n_used, const void* tw_m, const void* tw_n, void* frames) {
    auto fill = [&](auto& a, auto tag) {
        using T = decltype(tag);
        const int M = n_fft / 2;
        a.D = (const lra::mixed::cpx<T>*)D; a.d_frame_stride = M + 1; a.d_batch_stride = (long long)n_used * (M + 1); a.n_used = n_used;
        a.tw_m = (const lra::mixed::cpx<T>*)tw_m; a.tw_n = (const lra::mixed::cpx<T>*)tw_n; a.frames = (T*)frames;
    };
    if (is_f64) {
        lra::mixed::IrArgs<double> a = lra::mixed::IrArgs<double>();
        fill(a, double());
        return sim_mixed_irfft_n<double>(n_fft, a, batch);
    }
    lra::mixed::IrArgs<float> a = lra::mixed::IrArgs<float>();
    fill(a, float());
    return sim_mixed_irfft_n<float>(n_fft, a, batch);
}

namespace {
template <class T, int N> int sim_cqt_octave(lra::mixed::CqtArgs<T> a, long long batch) {
    using namespace lra::mixed;
    static_assert(cqt_lds_bytes<T, N>() <= (int)sizeof(g_postsim_dyn_lds), "simulated LDS too small");
    constexpr int F = cqt_frames_per_group<T, N>();
    a.groups_per_clip = (a.n_frames + F - 1) / F;
    run_grid((unsigned)(batch * a.groups_per_clip), NT, [=] { mixed_cqt_kernel<T, N>(a); });
    return 0;
}
template <class T> int sim_cqt_octave_n(int n_fft, const lra::mixed::CqtArgs<T>& a, long long batch) {
    switch (n_fft) {
        case 32: return sim_cqt_octave<T, 32>(a, batch);
        case 64: return sim_cqt_octave<T, 64>(a, batch);
        case 128: return sim_cqt_octave<T, 128>(a, batch);
        case 256: return sim_cqt_octave<T, 256>(a, batch);
        case 512: return sim_cqt_octave<T, 512>(a, batch);
        case 1024: return sim_cqt_octave<T, 1024>(a, batch);
        case 2048: return sim_cqt_octave<T, 2048>(a, batch);
        default: return 1;
    }
}
}  // namespace

// the launch of lra_cqt_octave_exec (lra_api.hip); tw_m / tw_n: the twiddle tables the context builds per frame length
extern "C" int postsim_cqt_octave(int n_fft, int is_f64, const void* y, long long batch, long long n, long long y_stride, int hop, int pad_mode, const void* tw_m, const void* tw_n,
                                  const int* row_ptr, const int* col, const void* val, const double* sqrt_len, void* out, long long n_frames, int n_total, int bin0, int row0, int n_rows) {
    auto fill = [&](auto& a, auto tag) {
        using T = decltype(tag);
        a.y = (const T*)y; a.y_stride = y_stride; a.n = n; a.hop = hop; a.pad = n_fft / 2; a.pad_mode = pad_mode;
        a.tw_m = (const lra::mixed::cpx<T>*)tw_m; a.tw_n = (const lra::mixed::cpx<T>*)tw_n;
        a.row_ptr = row_ptr; a.col = col; a.val = (const lra::mixed::cpx<T>*)val; a.sqrt_len = sqrt_len;
        a.out = (lra::mixed::cpx<T>*)out; a.n_frames = (int)n_frames; a.n_total = n_total; a.bin0 = bin0; a.row0 = row0; a.n_rows = n_rows;
        a.nonfinite_flag = nullptr;
    };
    if (is_f64) {
        lra::mixed::CqtArgs<double> a = lra::mixed::CqtArgs<double>();
        fill(a, double());
        return sim_cqt_octave_n<double>(n_fft, a, batch);
    }
    lra::mixed::CqtArgs<float> a = lra::mixed::CqtArgs<float>();
    fill(a, float());
    return sim_cqt_octave_n<float>(n_fft, a, batch);
}

