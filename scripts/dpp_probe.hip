// Build: hipcc --offload-arch=gfx950 -O2 -o probe/dpp_probe scripts/dpp_probe.hip ; run on the GPU box.
// Development probe (round 5): what the DPP wavefront shifts deliver on gfx950 (lane i <- lane i -/+ 1 across the 16- and 32-lane boundaries).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ int dpp(int x) { return __builtin_amdgcn_update_dpp(-1, x, CTRL, 0xf, 0xf, true); }
__global__ void k(int* out) {
    const int l = threadIdx.x;
    out[l] = dpp<0x138>(l + 100);        // wave_shr:1
    out[64 + l] = dpp<0x130>(l + 100);   // wave_shl:1
    out[128 + l] = dpp<0xB1>(l + 100);   // quad_perm [1,0,3,2]
    out[192 + l] = dpp<0x13C>(l + 100);  // wave_ror:1
    out[256 + l] = dpp<0x134>(l + 100);  // wave_rol:1
    out[320 + l] = dpp<0x111>(l + 100);  // row_shr:1
}
int main() {
    int* d; hipMalloc(&d, 384 * 4);
    k<<<1, 64>>>(d);
    int h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"wave_shr:1", "wave_shl:1", "quad_perm[1,0,3,2]", "wave_ror:1", "wave_rol:1", "row_shr:1"};
    for (int t = 0; t < 6; ++t) { printf("%-20s", names[t]); for (int l = 0; l < 64; ++l) printf(" %d", h[t * 64 + l] - 100); printf("\n"); }
    return 0;
}
