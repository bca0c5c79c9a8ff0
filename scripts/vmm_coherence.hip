// Build: hipcc --offload-arch=gfx950 -O3 -o probe/vmm_coherence scripts/vmm_coherence.hip ; run on the GPU box.
// Development probe (round 6): a buffer from hipMemCreate / hipMemMap is written by kernel A (stream s1), then by kernel B (stream 0), then read by kernel C and by hipMemcpy.
// Do B's writes always win?  Variants: candidates created and released in between (the lra_malloc_placed sequence), address ranges re-reserved, hipMalloc as the control.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s (%d) at line %d\n", hipGetErrorString(e_), (int)e_, __LINE__); exit(1); } } while (0)
__global__ void fillk(float* p, size_t n, float v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void countk(const float* p, size_t n, float v, unsigned long long* bad) { unsigned long long c = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != v; if (c) atomicAdd(bad, c); }
struct VB { void* p; size_t padded, chunk; std::vector<hipMemGenericAllocationHandle_t> h; };
// mode 4: ONE reserved arena, ranges handed out first-fit and re-used after their chunks were unmapped; the arena is never freed
static char* g_arena = nullptr; static size_t g_arena_bytes = (size_t)64 << 30; static std::vector<std::pair<size_t, size_t>> g_used;  // (offset, size)
static void* arena_take(size_t bytes) {
    if (!g_arena) { void* p; CK(hipMemAddressReserve(&p, g_arena_bytes, 0, nullptr, 0)); g_arena = (char*)p; }
    size_t off = 0;
    for (;;) { bool hit = false; for (auto& u : g_used) if (off < u.first + u.second && u.first < off + bytes) { off = u.first + u.second; hit = true; } if (!hit) break; }
    if (off + bytes > g_arena_bytes) { printf("arena full\n"); exit(1); }
    g_used.push_back({off, bytes});
    return g_arena + off;
}
static void arena_give(void* p) { for (size_t i = 0; i < g_used.size(); ++i) if (g_arena + g_used[i].first == (char*)p) { g_used.erase(g_used.begin() + i); return; } }
static int g_mode = 0;
static VB vmm(size_t bytes, size_t chunk) {
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    VB b; b.chunk = chunk; b.padded = (bytes + chunk - 1) / chunk * chunk;
    if (g_mode == 4) b.p = arena_take(b.padded); else
    CK(hipMemAddressReserve(&b.p, b.padded, 0, nullptr, 0));
    for (size_t o = 0; o < b.padded; o += chunk) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); CK(hipMemMap((char*)b.p + o, chunk, 0, h, 0)); b.h.push_back(h); }
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(b.p, b.padded, &acc, 1));
    return b;
}
// g_mode 0: unmap + release per chunk, free the range; 1: one unmap of the whole range, then releases; 2: as 0 but the address range is never freed; 3: as 0 with hipDeviceSynchronize around
static void rel(VB& b) {
    if (g_mode == 3) CK(hipDeviceSynchronize());
    if (g_mode == 1) { CK(hipMemUnmap(b.p, b.padded)); for (auto h : b.h) CK(hipMemRelease(h)); }
    else for (size_t i = 0; i < b.h.size(); ++i) { CK(hipMemUnmap((char*)b.p + i * b.chunk, b.chunk)); CK(hipMemRelease(b.h[i])); }
    if (g_mode == 4) arena_give(b.p); else
    if (g_mode != 2) CK(hipMemAddressFree(b.p, b.padded));
    if (g_mode == 3) CK(hipDeviceSynchronize());
}
int main(int argc, char** argv) {
    g_mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t chunk_mb = argc > 2 ? (size_t)atoi(argv[2]) : 64;
    printf("== release mode %d, %zu MiB handles\n", g_mode, chunk_mb);
    const size_t bytes = (size_t)1017062400, n = bytes / 4;
    hipStream_t s1; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    unsigned long long* bad; CK(hipMalloc(&bad, 8));
    std::vector<float> host(1 << 20);
    auto check = [&](float* p, const char* what) {
        hipLaunchKernelGGL(fillk, dim3(2048), dim3(256), 0, s1, p, n, 7.0f);   // "probe" on the side stream
        CK(hipStreamSynchronize(s1));
        hipLaunchKernelGGL(fillk, dim3(4096), dim3(256), 0, 0, p, n, 4.0f);   // the real writer on the default stream
        CK(hipDeviceSynchronize());
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(countk, dim3(4096), dim3(256), 0, 0, p, n, 4.0f, bad);
        unsigned long long hb = 0; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(host.data(), p, host.size() * 4, hipMemcpyDeviceToHost));
        size_t hbad = 0; for (float v : host) hbad += v != 4.0f;
        printf("%-46s kernel sees %llu stale floats, hipMemcpy of the first 4 MB %zu\n", what, hb, hbad);
    };
    float* d; CK(hipMalloc(&d, bytes)); check(d, "hipMalloc");
    std::vector<VB> live;
    for (int i = 0; i < 4; ++i) { live.push_back(vmm(bytes, chunk_mb << 20)); char w[64]; snprintf(w, 64, "vmm #%d (all alive)", i); check((float*)live.back().p, w); }
    for (int i = 0; i < 4; ++i) {   // the lra_malloc_placed sequence: three candidates probed and released, one kept
        std::vector<VB> c;
        for (int k = 0; k < 4; ++k) { c.push_back(vmm(bytes, chunk_mb << 20)); hipLaunchKernelGGL(fillk, dim3(2048), dim3(256), 0, s1, (float*)c.back().p, n, 9.0f); CK(hipStreamSynchronize(s1)); }
        for (int k = 0; k < 4; ++k) if (k != (i % 4)) rel(c[k]);
        char w[64]; snprintf(w, 64, "vmm kept candidate %d of 4, others released", i % 4); check((float*)c[i % 4].p, w);
        live.push_back(c[i % 4]);
    }
    for (auto& b : live) check((float*)b.p, "re-check of a live buffer");
    // churn: release every second live buffer, allocate new ones (mode 4: into the freed ranges), re-check everything
    for (int round = 0; round < 3; ++round) {
        for (size_t i = round % 2; i < live.size(); i += 2) { rel(live[i]); live[i] = vmm(bytes, chunk_mb << 20); }
        for (auto& b : live) check((float*)b.p, "churn re-check");
    }
    return 0;
}
