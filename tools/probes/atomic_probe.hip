// Probe: cost of 256 workgroups x 1024 threads each adding one int64 into replica (blockIdx % R) of a
// P = 1024-column accumulator (the cross-workgroup reduction pattern considered for the one-launch step).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void __launch_bounds__(1024) k_atomic(unsigned long long* acc, int R, int mode)
{
    __shared__ float lds[16 * 64];
    // a little LDS work so the kernel is not empty in mode 0
    lds[threadIdx.x] = threadIdx.x * 0.5f;
    __syncthreads();
    const float v = lds[(threadIdx.x * 7) & 1023];
    if (mode == 0) { if (v < -1.f) acc[0] = 1; return; }
    const long long fx = __float2ll_rn(v * 1099511627776.0f);
    atomicAdd(acc + (size_t)(blockIdx.x % R) * 1024 + threadIdx.x, (unsigned long long)fx);
}
__global__ void __launch_bounds__(1024) k_read(const unsigned long long* acc, int R, float* out)
{
    long long s = 0;
    for (int r = 0; r < R; ++r) s += (long long)acc[(size_t)r * 1024 + threadIdx.x];
    if (blockIdx.x == 0) out[threadIdx.x] = (float)s * (1.0f / 1099511627776.0f);
}
int main()
{
    unsigned long long* acc; float* out;
    hipMalloc(&acc, 64 * 1024 * 8); hipMalloc(&out, 4096);
    hipMemset(acc, 0, 64 * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int R : {1, 4, 8, 16, 32, 64}) for (int mode : {0, 1}) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_atomic, dim3(256), dim3(1024), 0, 0, acc, R, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_atomic, dim3(256), dim3(1024), 0, 0, acc, R, mode);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("R=%2d mode=%d  %.2f us per launch (back-to-back)\n", R, mode, ms * 1000 / 200);
    }
    for (int R : {16}) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_read, dim3(256), dim3(1024), 0, 0, acc, R, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read R=%d: %.2f us per launch\n", R, ms * 1000 / 200);
    }
    return 0;
}
