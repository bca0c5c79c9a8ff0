// Build: hipcc --offload-arch=gfx950 -O3 -o probe/vmm_placement scripts/vmm_placement.hip ; run on the GPU box.
// Development probe (round 5): does HOW the 2.7 GB spectrum is allocated decide which of the placement levels of profiles/r05_pitch.md it gets?
// The STFT's forward access stream (strips of 162 consecutive 8 200-byte rows per wave, 16 x dwordx2 per row in butterfly order + the middle bin, 2 048 B of
// PCM read per row, next row's loads ahead of this row's stores: csrc/lra_probe.h) on buffers from hipMalloc and from the virtual-memory API (hipMemCreate +
// hipMemMap) as ONE physical handle, as 64 MiB handles and as 2 MiB handles mapped back to back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s (%d) at line %d\n", hipGetErrorString(e_), (int)e_, __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ROWS = 1292, CLIPS = 256, BINS = 1025, M = 1024, STRIP = 162;
constexpr long long CLIP_IN = 661500LL * 4, ROWB = 8200;

__global__ __launch_bounds__(64) void forward_stream(const char* __restrict__ in, char* __restrict__ out, int xcd_chunk) {
    extern __shared__ char pad[];
    const int lane = threadIdx.x;
    long long b = blockIdx.x;
    if (xcd_chunk > 0) b = (b % 8) * (long long)xcd_chunk + b / 8;
    const int spc = (ROWS + STRIP - 1) / STRIP;
    const long long clip = b / spc;
    const int part = (int)(b % spc);
    if (clip >= CLIPS) return;
    f2 v = {(float)lane, (float)b};
    f2 cur[4], nx[4];
    for (int c = 0; c < 4; ++c) cur[c] = v;
    for (int it = 0; it < STRIP; ++it) {
        const int row = part * STRIP + it;
        if (row >= ROWS) break;
        for (int c = 0; c < 4; ++c) nx[c] = cur[c];
        if (row + 1 < ROWS - 1) {
            const f2* src = reinterpret_cast<const f2*>(in + clip * CLIP_IN + (long long)(row + 1) * 2048) + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c) nx[c] = src[c * 64];
        }
        v.x += cur[0].x + cur[3].y;
        f2* rp = reinterpret_cast<f2*>(out + (clip * ROWS + row) * ROWB);
        f2* pk = rp + lane;
        f2* pm = rp + (M - lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) { pk[i * 64] = v; pm[-i * 64] = v; }
        if (lane == 0) rp[M / 2] = v;
        for (int c = 0; c < 4; ++c) cur[c] = nx[c];
    }
}

static float time_stream(const char* in, char* out) {
    const int spc = (ROWS + STRIP - 1) / STRIP;
    const long long strips = (long long)CLIPS * spc, grid = (strips + 7) / 8 * 8;
    const int lds = (160 * 1024 / 12) & ~255;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 30; ++r) hipLaunchKernelGGL(forward_stream, dim3((unsigned)grid), dim3(64), lds, 0, in, out, (int)(grid / 8));
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(forward_stream, dim3((unsigned)grid), dim3(64), lds, 0, in, out, (int)(grid / 8));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / 10);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best;
}

// order: 0 = handles mapped in creation order, 1 = a pseudo-random permutation (the virtual neighbours of a chunk are physically elsewhere), 2 = a stride permutation
static char* vmm_alloc(size_t total, size_t chunk, std::vector<hipMemGenericAllocationHandle_t>& handles, int order = 0, unsigned seed = 12345) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (chunk == 0) chunk = total;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t padded = (total + chunk - 1) / chunk * chunk;
    void* ptr = nullptr;
    CK(hipMemAddressReserve(&ptr, padded, 0, nullptr, 0));
    const size_t n = padded / chunk;
    std::vector<size_t> slot(n);
    for (size_t i = 0; i < n; ++i) slot[i] = i;
    if (order == 1) {
        unsigned long long st = seed;
        for (size_t i = n - 1; i > 0; --i) {
            st = st * 6364136223846793005ULL + 1442695040888963407ULL;
            const size_t j = (size_t)((st >> 33) % (i + 1));
            std::swap(slot[i], slot[j]);
        }
    } else if (order == 2) {
        size_t stride = 1;
        while (stride * stride < n) ++stride;  // ~sqrt(n): consecutive creations land sqrt(n) slots apart
        size_t k = 0;
        for (size_t r = 0; r < stride; ++r)
            for (size_t i = r; i < n; i += stride) slot[k++] = i;
    }
    for (size_t i = 0; i < n; ++i) {   // created in order (physically next to each other, presumably), mapped at slot[i]
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char*)ptr + slot[i] * chunk, chunk, 0, h, 0));
        handles.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(ptr, padded, &acc, 1));
    return (char*)ptr;
}

int main() {
    const size_t total = (size_t)CLIPS * ROWS * ROWB;
    char* in;
    CK(hipMalloc(&in, (size_t)CLIPS * CLIP_IN + 4096));
    CK(hipMemset(in, 0, (size_t)CLIPS * CLIP_IN));
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gmin = 0, grec = 0;
        CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
        CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
        printf("vmm granularity: minimum %zu, recommended %zu bytes\n", gmin, grec);
    }
    std::vector<hipMemGenericAllocationHandle_t> keep;
    {
        const char* on[3] = {"in order", "shuffled", "strided"};
        for (int order = 0; order < 3; ++order)
            for (size_t mb : {2, 8, 32, 128})
                for (int a = 0; a < 2; ++a)
                    printf("vmm %3zu MiB handles %s #%d: %.3f ms\n", mb, on[order], a, time_stream(in, vmm_alloc(total, mb << 20, keep, order, 777u + 13u * a)));
        fflush(stdout);
    }
    for (int round = 0; round < 1; ++round) {
        for (int a = 0; a < 4; ++a) {
            char* d; CK(hipMalloc(&d, total + (size_t)a * (3 << 20)));
            printf("hipMalloc #%d: %.3f ms\n", a, time_stream(in, d));
        }
        for (int a = 0; a < 3; ++a) printf("vmm one handle #%d: %.3f ms\n", a, time_stream(in, vmm_alloc(total, 0, keep)));
        for (int a = 0; a < 3; ++a) printf("vmm 64 MiB handles #%d: %.3f ms\n", a, time_stream(in, vmm_alloc(total, 64 << 20, keep)));
        for (int a = 0; a < 2; ++a) printf("vmm 2 MiB handles #%d: %.3f ms\n", a, time_stream(in, vmm_alloc(total, 2 << 20, keep)));
        fflush(stdout);
    }
    return 0;
}
