// postsim.cpp -- runs the PCEN / band max-filter kernel bodies of librosa_amd/csrc/lra_pcen.h on host threads.
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

#include "../../librosa_amd/csrc/lra_pcen.h"

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
        run_grid(grid, 256, [=] { lra::maxfilter_bands_kernel<double>((const double*)S, (double*)out, outer, n_bands, inner, size); });
    else
        run_grid(grid, 256, [=] { lra::maxfilter_bands_kernel<float>((const float*)S, (float*)out, outer, n_bands, inner, size); });
    return 0;
}
}
