// valu_probe.hip -- issue rate of the vector ALU on gfx950 for the instructions the loss kernels are made of: v_fma_f32 against the
// transcendental unit (v_exp_f32, v_log_f32, v_rcp_f32).  Every lane keeps 8 independent chains in registers (enough to cover
// the instruction latency at 4+ waves per SIMD), the grid fills every SIMD with 8 waves, nothing touches memory.  Prints lane
// operations per second and cycles per wave64 instruction at the measured clock-independent rate (ops / s / (1024 SIMDs)).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/build/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void probe(float* sink, int iters, float seed) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = seed + 0.001f * (threadIdx.x + k);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (OP == 0) v[k] = __builtin_fmaf(v[k], 0.999f, 0.001f);
            else if (OP == 1) v[k] = __builtin_amdgcn_exp2f(v[k] * 0.5f) ;          // v_mul + v_exp
            else if (OP == 2) v[k] = __builtin_amdgcn_logf(v[k] + 2.0f);           // v_add + v_log
            else if (OP == 3) v[k] = __builtin_amdgcn_rcpf(v[k] + 1.5f);           // v_add + v_rcp
            else if (OP == 4) v[k] = __builtin_fmaf(v[k], 0.5f, 0.25f) + 0.125f;   // v_fma + v_add (2 full-rate)
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    if (s == 12345.678f) sink[0] = s;
}

template <int OP>
static double run(const char* name, int per_iter_instr, float* sink) {
    const int iters = 4096, grid = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(256), 0, 0, sink, iters, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(256), 0, 0, sink, iters, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = 5.0 * grid * 4 * (double)iters * 8 * per_iter_instr;   // wave64 instructions issued
    const double per_simd_per_s = wave_instr / (ms * 1e-3) / 1024.0;
    printf("%-34s %8.3f ms   %7.2f G wave-instr/s per SIMD-second^-1 x1024 = %6.2f T lane-ops/s   %5.2f ns per wave64 instr per SIMD\n", name, ms / 5,
           per_simd_per_s / 1e9 * 1024, wave_instr * 64 / (ms * 1e-3) / 1e12, 1e9 / per_simd_per_s);
    return 1e9 / per_simd_per_s;
}

int main() {
    float* sink; CK(hipMalloc(&sink, 64));
    const double fma = run<0>("v_fma_f32", 1, sink);
    const double two = run<4>("v_fma_f32 + v_add_f32", 2, sink) / 1.0;
    const double ex = run<1>("v_mul_f32 + v_exp_f32", 2, sink);
    const double lg = run<2>("v_add_f32 + v_log_f32", 2, sink);
    const double rc = run<3>("v_add_f32 + v_rcp_f32", 2, sink);
    printf("cost in units of one full-rate instruction: exp %.2f  log %.2f  rcp %.2f   (pair time x 2 / fma time - 1; fma+add pair = %.2f)\n",
           2 * ex / fma - 1, 2 * lg / fma - 1, 2 * rc / fma - 1, 2 * two / fma);
    return 0;
}
