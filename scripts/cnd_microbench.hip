// Is v_cndmask_b32 slow on gfx950?  Chain-free streams of selects against a lane mask held in VCC / in an SGPR pair, compiled
// from C++ (the compiler's own v_cndmask) and from inline asm, beside v_bfi_b32 and v_and_b32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 4096
__global__ __launch_bounds__(64) void k_c(int *out, int seed, int thr) {
    int a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const bool m = (int)threadIdx.x < thr;
    const int b = seed | 5;
    for (int i = 0; i < REP; ++i) {
        a0 = m ? a0 ^ b : a0 + 1; a1 = m ? a1 ^ b : a1 + 1; a2 = m ? a2 ^ b : a2 + 1; a3 = m ? a3 ^ b : a3 + 1;
        a4 = m ? a4 ^ b : a4 + 1; a5 = m ? a5 ^ b : a5 + 1; a6 = m ? a6 ^ b : a6 + 1; a7 = m ? a7 ^ b : a7 + 1;
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ __launch_bounds__(64) void k_asm(int *out, int seed, int thr) {
    int a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const int b = seed | 5;
    asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(thr), "v"((int)threadIdx.x) : "vcc");
    for (int i = 0; i < REP; ++i)
        asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
                     "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
static void run(const char *name, void (*kern)(int *, int, int), int *d_out, double per_iter) {
    const int blocks = 256 * 4 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_out, 1, 33);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_out, r, 33);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double iters = 4.0 * blocks * (double)REP;
    printf("%-40s %.1f cycles per loop iteration per SIMD (4 waves/SIMD) = %.2f per instruction if %g instructions\n", name,
           1024.0 * 2.4e9 * ms * 1e-3 / iters, 1024.0 * 2.4e9 * ms * 1e-3 / iters / per_iter, per_iter);
}
int main() {
    int *d_out;
    (void)hipMalloc(&d_out, 256 * 4 * 8 * 64 * sizeof(int));
    run("C++ select of (xor, add) x 8", k_c, d_out, 24);
    run("asm v_cndmask_b32_e32 x 8 (vcc set once)", k_asm, d_out, 8);
    return 0;
}
