// bw_probe.hip -- HBM ceiling probe for MI355X: read-only, copy and 8-stream-reduce kernels with explicit
// loads-in-flight, default vs non-temporal.  Build: hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o tools/build/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
#define float4 v4f
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ in, float* __restrict__ sink, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    float acc = 0.f;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], &out[i + u * stride]); else out[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) out[i] = in[i];
}

// 8 streams (spaced n4 apart) -> 1 output: the access shape of the TTA reduce, without any index games
template <bool NT>
__global__ __launch_bounds__(256) void reduce8_kernel(const float4* __restrict__ in, float4* __restrict__ out, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = NT ? __builtin_nontemporal_load(&in[i + k * n4]) : in[i + k * n4];
        float4 s = v[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
        out[i] = s;
    }
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const long long bytes = 1LL << 30;  // 1 GiB per buffer, 4 buffers rotate (>> 256 MiB Infinity Cache)
    const long long n4 = bytes / 16;
    const int NB = 4;
    float4* buf[NB]; float4* out; float* sink;
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&buf[i], bytes)); CK(hipMemset(buf[i], i + 1, bytes)); }
    CK(hipMalloc(&out, bytes)); CK(hipMalloc(&sink, 4));
    const int reps = 12;
    for (int blocks_per_cu : {4, 8, 16, 32}) {
        const int grid = 256 * blocks_per_cu;
#define RD(U, NT) { double ms = time_ms([&](int i) { hipLaunchKernelGGL((read_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, buf[i % NB], sink, n4); }, reps); \
        printf("read   U=%d nt=%d grid=%5d : %8.1f GB/s\n", U, NT, grid, bytes / ms / 1e6); }
        RD(4, false) RD(8, false) RD(8, true) RD(16, false)
#define CP(U, NT) { double ms = time_ms([&](int i) { hipLaunchKernelGGL((copy_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, buf[i % NB], out, n4); }, reps); \
        printf("copy   U=%d nt=%d grid=%5d : %8.1f GB/s (r+w)\n", U, NT, grid, 2.0 * bytes / ms / 1e6); }
        CP(4, false) CP(8, false) CP(8, true)
        { double ms = time_ms([&](int i) { hipLaunchKernelGGL((reduce8_kernel<false>), dim3(grid), dim3(256), 0, 0, buf[i % NB], out, n4 / 8); }, reps);
          printf("reduce8      nt=0 grid=%5d : %8.1f GB/s read (+1/8 write)\n", grid, bytes / ms / 1e6); }
        { double ms = time_ms([&](int i) { hipLaunchKernelGGL((reduce8_kernel<true>), dim3(grid), dim3(256), 0, 0, buf[i % NB], out, n4 / 8); }, reps);
          printf("reduce8      nt=1 grid=%5d : %8.1f GB/s read (+1/8 write)\n", grid, bytes / ms / 1e6); }
    }
    { double ms = time_ms([&](int i) { CK(hipMemcpyAsync(out, buf[i % NB], bytes, hipMemcpyDeviceToDevice, 0)); }, reps);
      printf("hipMemcpy D2D            : %8.1f GB/s (r+w)\n", 2.0 * bytes / ms / 1e6); }
    { double ms = time_ms([&](int i) { CK(hipMemsetAsync(out, 0, bytes, 0)); }, reps);
      printf("hipMemset                : %8.1f GB/s (w)\n", bytes / ms / 1e6); }
    return 0;
}
