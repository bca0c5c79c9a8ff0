// Microbenchmark 2 (GPU box): select / lane-exchange forms.  hipcc --offload-arch=gfx950 -O3 scripts/valu_probe2.hip -o probe/valu_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define OP8(fmt) asm volatile(fmt("%0", "%0", "%1") "\n" fmt("%1", "%1", "%2") "\n" fmt("%2", "%2", "%3") "\n" fmt("%3", "%3", "%4") "\n" fmt("%4", "%4", "%5") "\n" fmt("%5", "%5", "%6") "\n" fmt("%6", "%6", "%7") "\n" fmt("%7", "%7", "%0") \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "s"(sm) : "vcc");
#define F_CND_VCC(d, a, b) "v_cndmask_b32 " d ", " a ", " b ", vcc"
#define F_CND_S(d, a, b) "v_cndmask_b32_e64 " d ", " a ", " b ", %9"
#define F_BFI(d, a, b) "v_bfi_b32 " d ", %8, " a ", " b
#define F_AND(d, a, b) "v_and_b32 " d ", " a ", " b
#define F_DPP_SHR(d, a, b) "v_mov_b32_dpp " d ", " b " row_shr:1 row_mask:0xf bank_mask:0xf"
#define F_DPP_QP(d, a, b) "v_mov_b32_dpp " d ", " b " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define F_DPP_WSHR(d, a, b) "v_mov_b32_dpp " d ", " b " wave_shr:1 row_mask:0xf bank_mask:0xf"
#define F_ADD_DPP(d, a, b) "v_add_f32_dpp " d ", " b ", " a " row_shr:1 row_mask:0xf bank_mask:0xf"
#define F_SWZ(d, a, b) "ds_swizzle_b32 " d ", " b " offset:0x041f\n s_waitcnt lgkmcnt(0)"
#define F_PERML32(d, a, b) "v_permlane32_swap_b32 " d ", " b
#define F_MAD24(d, a, b) "v_mad_u32_u24 " d ", " a ", " b ", %8"
#define F_LSHLADD(d, a, b) "v_lshl_add_u32 " d ", " a ", 3, " b
#define F_FMAC(d, a, b) "v_fmac_f32 " d ", " a ", " b
#define F_MUL(d, a, b) "v_mul_f32 " d ", " a ", " b
#define F_CMP(d, a, b) "v_cmp_eq_u32 vcc, " a ", " b
#define F_CND_E64VCC(d, a, b) "v_cndmask_b32_e64 " d ", " a ", " b ", vcc"
#define F_CND_MIX(d, a, b) "v_add_u32 " a ", " a ", " b "\n v_cndmask_b32 " d ", " a ", " b ", vcc"
#define F_CMP_CND(d, a, b) "v_cmp_eq_u32 vcc, " a ", %8\n v_cndmask_b32 " d ", " a ", " b ", vcc"
template <int MODE> __global__ __launch_bounds__(64) void k(float* out, int iters, long long* cyc) {
    unsigned a0 = threadIdx.x, a1 = 3, a2 = 5, a3 = 7, a4 = 11, a5 = 13, a6 = 17, a7 = 19;
    const unsigned m = threadIdx.x == 0 ? 0xffffffffu : 0u;
    const unsigned long long sm = 1ull;  // lane 0
    asm volatile("v_cmp_eq_u32 vcc, 0, %0" :: "v"(threadIdx.x) : "vcc");
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP8(OP8(F_CND_VCC)) }
        else if (MODE == 1) { REP8(OP8(F_CND_S)) }
        else if (MODE == 2) { REP8(OP8(F_BFI)) }
        else if (MODE == 3) { REP8(OP8(F_AND)) }
        else if (MODE == 4) { REP8(OP8(F_DPP_SHR)) }
        else if (MODE == 5) { REP8(OP8(F_DPP_QP)) }
        else if (MODE == 6) { REP8(OP8(F_DPP_WSHR)) }
        else if (MODE == 7) { REP8(OP8(F_ADD_DPP)) }
        else if (MODE == 8) { REP8(OP8(F_SWZ)) }
        else if (MODE == 9) { REP8(OP8(F_PERML32)) }
        else if (MODE == 10) { REP8(OP8(F_MAD24)) }
        else if (MODE == 11) { REP8(OP8(F_LSHLADD)) }
        else if (MODE == 12) { REP8(OP8(F_FMAC)) }
        else if (MODE == 13) { REP8(OP8(F_MUL)) }
        else if (MODE == 14) { REP8(OP8(F_CMP)) }
        else if (MODE == 15) { REP8(OP8(F_CMP_CND)) }
        else if (MODE == 16) { REP8(OP8(F_CND_E64VCC)) }
        else if (MODE == 17) { REP8(OP8(F_CND_MIX)) }
        else if (MODE == 18) { REP8(asm volatile("v_cmp_eq_u32 vcc, 0, %0" :: "v"(threadIdx.x) : "vcc"); OP8(F_CND_VCC)) }
        else if (MODE == 19) { asm volatile("v_cmp_eq_u32 vcc, 0, %0" :: "v"(threadIdx.x) : "vcc"); REP8(OP8(F_CND_VCC)) }
        else if (MODE == 20) { asm volatile("s_mov_b64 vcc, 1" ::: "vcc"); REP8(OP8(F_CND_VCC)) }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    unsigned s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 0x12345678u) out[threadIdx.x] = (float)s;
}
template <int MODE> void run(const char* name) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 8 * 4096);
    for (int wps : {1, 2}) {
        const int iters = 1000, grid = 256 * 4 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, 10, cyc);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, cyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long h[8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        const double n = 64.0 * iters;
        printf("%-34s %d wave/SIMD: %.3f ms  ticks/instr/wave %.2f   wall %.2f ns/instr/SIMD\n", name, wps, ms, h[0] / n, ms * 1e6 / (n * wps));
    }
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0>("v_cndmask_b32 vcc"); run<1>("v_cndmask_b32_e64 sgpr"); run<2>("v_bfi_b32 (vgpr mask)"); run<3>("v_and_b32"); run<4>("v_mov_b32_dpp row_shr:1"); run<5>("v_mov_b32_dpp quad_perm");
    run<6>("v_mov_b32_dpp wave_shr:1"); run<7>("v_add_f32_dpp row_shr:1"); run<8>("ds_swizzle + wait"); run<9>("v_permlane32_swap"); run<10>("v_mad_u32_u24"); run<11>("v_lshl_add_u32");
    run<12>("v_fmac_f32"); run<13>("v_mul_f32"); run<14>("v_cmp_eq_u32 -> vcc"); run<15>("v_cmp + v_cndmask (2 instr)");
    run<16>("v_cndmask_b32_e64 vcc"); run<17>("v_add_u32 + v_cndmask vcc (2 instr)"); run<18>("v_cmp then 8 cndmask (9 instr/8)"); run<19>("v_cmp then 64 cndmask"); run<20>("s_mov vcc then 64 cndmask");
    return 0;
}
