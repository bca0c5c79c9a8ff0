// Development probe (GPU box): semantics of v_permlane32_swap_b32 on gfx950 -- operand roles, same register for both operands, EXEC.
//   hipcc --offload-arch=gfx950 -O2 scripts/permlane_probe.hip -o probe/permlane_probe && probe/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int l = threadIdx.x;
    int a = 100 + l, b = 200 + l;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // two registers
    out[l] = a; out[64 + l] = b;
    int c = 300 + l;
    asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(c));             // one register on both sides
    out[128 + l] = c;
    int d = 400 + l;
    if (l != 0 && l != 32 && l != 5) asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(d));  // lanes 0, 32 (a pair) and 5 (half a pair) masked off
    out[192 + l] = d;
}
int main() {
    int* d; hipMalloc(&d, 256 * sizeof(int));
    k<<<1, 64>>>(d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"a (vdst)", "b (src)", "c (same reg)", "d (exec-masked, same reg)"};
    for (int r = 0; r < 4; ++r) {
        printf("%s:", names[r]);
        for (int l = 0; l < 64; ++l) if (l < 7 || (l >= 30 && l < 39) || l == 63) printf(" [%d]=%d", l, h[r * 64 + l]);
        printf("\n");
    }
    return 0;
}
