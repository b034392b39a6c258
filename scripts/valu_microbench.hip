// VALU issue-rate microbenchmark for gfx950 (settles the "2 or 4 cycles per wave64 VALU instruction" question of VERDICT r01).
// Each wave runs a long chain-free stream of one instruction type over 8 independent registers; the grid puts W waves on
// every SIMD (256 CUs x 4 SIMDs x W single-wave workgroups).  Reported: wave-instructions per second for the whole chip
// and the implied cycles per instruction per SIMD at the clock measured with s_memtime-free wall time (HIP events).
//   hipcc -O3 --offload-arch=gfx950 scripts/valu_microbench.hip -o /tmp/valu && /tmp/valu
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP 4096
template <int KIND>
__global__ __launch_bounds__(64) void k(int *out, int seed) {
    int a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const int b = seed | 1;
    for (int i = 0; i < REP; ++i) {
        if (KIND == 0) {
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (KIND == 1) {
            asm volatile("v_pk_add_i16 %0, %0, %8\n v_pk_add_i16 %1, %1, %8\n v_pk_add_i16 %2, %2, %8\n v_pk_add_i16 %3, %3, %8\n"
                         "v_pk_add_i16 %4, %4, %8\n v_pk_add_i16 %5, %5, %8\n v_pk_add_i16 %6, %6, %8\n v_pk_add_i16 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (KIND == 2) {
            asm volatile("v_bfe_i32 %0, %0, %8, 6\n v_bfe_i32 %1, %1, %8, 6\n v_bfe_i32 %2, %2, %8, 6\n v_bfe_i32 %3, %3, %8, 6\n"
                         "v_bfe_i32 %4, %4, %8, 6\n v_bfe_i32 %5, %5, %8, 6\n v_bfe_i32 %6, %6, %8, 6\n v_bfe_i32 %7, %7, %8, 6\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else {
            asm volatile("v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n"
                         "v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int KIND>
static void run(const char *name, int *d_out) {
    for (int w = 1; w <= 8; w *= 2) {
        const int blocks = 256 * 4 * w;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, 1);  // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, r);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr = 4.0 * blocks * (double)REP * 8.0;
        const double rate = instr / (ms * 1e-3);
        printf("%-14s %d wave(s)/SIMD: %8.1f G wave-instr/s  -> %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, w, rate / 1e9,
               1024.0 * 2.4e9 / rate);
    }
}

int main() {
    int *d_out;
    hipMalloc(&d_out, 256 * 4 * 8 * 64 * sizeof(int));
    run<0>("v_add_u32", d_out);
    run<1>("v_pk_add_i16", d_out);
    run<2>("v_bfe_i32", d_out);
    run<3>("v_max_i32", d_out);
    return 0;
}
