// Issue rate of packed against single fp32 fused multiply-adds on one GPU: the same number of instructions per wave, so
// equal times mean a packed instruction costs one issue slot (twice the flops), double time means two.
// Measured on an MI355X: v_fma_f32 480, v_pk_fma_f32 456, v_pk_mul_f32 551, v_pk_add_f32 582 G wave-instructions/s
// (1,024 SIMDs x 2.4 GHz / 4 cycles = 614 G/s): packed instructions issue at the full rate.
//   hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "HIP error at line %d\n", __LINE__); std::exit(1); } } while (0)
typedef float fx2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    fx2 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = fx2{a + i + threadIdx.x, b + i};
    const fx2 m = {a, a}, c = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i].x = __builtin_fmaf(acc[i].x, m.x, c.x);                       // 8 v_fma_f32
            else if (MODE == 1) acc[i] = __builtin_elementwise_fma(acc[i], m, c);              // 8 v_pk_fma_f32
            else if (MODE == 2) acc[i] = acc[i] * m;                                            // 8 v_pk_mul_f32
            else acc[i] = acc[i] + c;                                                           // 8 v_pk_add_f32
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
float run(float* d, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(4096), dim3(256), 0, 0, d, 16, 1.0001f, 0.5f);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(4096), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}
int main() {
    float* d;
    CK(hipMalloc(&d, 4096 * 256 * 4));
    const int iters = 20000;
    const double insts = 4096.0 * 4 * iters * 8;   // wave instructions
    const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters);
    printf("v_fma_f32     %.2f ms  %.1f G wave-inst/s\n", t0, insts / t0 * 1e-6);
    printf("v_pk_fma_f32  %.2f ms  %.1f G wave-inst/s\n", t1, insts / t1 * 1e-6);
    printf("v_pk_mul_f32  %.2f ms  %.1f G wave-inst/s\n", t2, insts / t2 * 1e-6);
    printf("v_pk_add_f32  %.2f ms  %.1f G wave-inst/s\n", t3, insts / t3 * 1e-6);
    return 0;
}
