// Issue rates of the instructions a near-field SDF evaluation is made of, against v_fma_f32: the same number of
// instructions per wave in every mode, eight independent chains per lane, so time ratios are issue-slot ratios.
//   hipcc --offload-arch=gfx950 -O3 -o trans_rate trans_rate.hip && ./trans_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "HIP error at line %d\n", __LINE__); std::exit(1); } } while (0)
enum { FMA, EXP2, LOG2, RCP, RSQ, SQRT, FRACT, BPERMUTE, READLANE_DPP, N_MODES };
const char* kNames[N_MODES] = {"v_fma_f32", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_fract_f32", "ds_bpermute_b32", "v_mov_dpp"};
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = a + 0.001f * i + 1e-6f * threadIdx.x;
    const int addr = ((threadIdx.x * 7 + 3) & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == FMA) acc[i] = __builtin_fmaf(acc[i], a, b);
            else if (MODE == EXP2) acc[i] = __builtin_amdgcn_exp2f(acc[i]);
            else if (MODE == LOG2) acc[i] = __builtin_amdgcn_logf(acc[i]);
            else if (MODE == RCP) acc[i] = __builtin_amdgcn_rcpf(acc[i]);
            else if (MODE == RSQ) acc[i] = __builtin_amdgcn_rsqf(acc[i]);
            else if (MODE == SQRT) acc[i] = __builtin_amdgcn_sqrtf(acc[i]);
            else if (MODE == FRACT) acc[i] = __builtin_amdgcn_fractf(acc[i]) + 0.0f * acc[i];
            else if (MODE == BPERMUTE) acc[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(acc[i])));
            else acc[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[i]), 0xB1, 0xf, 0xf, false));
            asm volatile("" : "+v"(acc[i]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
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
    const int iters = 4000;
    const double insts = 4096.0 * 4 * iters * 8;   // wave instructions
    float t[N_MODES] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters), run<5>(d, iters), run<6>(d, iters), run<7>(d, iters), run<8>(d, iters)};
    for (int m = 0; m < N_MODES; ++m) printf("%-16s %8.2f ms  %7.1f G wave-inst/s  %.2f x v_fma_f32\n", kNames[m], t[m], insts / t[m] * 1e-6, t[m] / t[0]);
    return 0;
}
