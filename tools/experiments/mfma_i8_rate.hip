// Cycles per v_mfma_i32_16x16x64_i8 / v_mfma_i32_32x32x32_i8 / v_mfma_f32_16x16x32_bf16 on one SIMD: one wave, four independent accumulators,
// back-to-back issue, s_memtime around the loop.  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_i8_rate tools/experiments/mfma_i8_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int WHICH>
__global__ void k(int iters, long long *out, int *sink)
{
    v4i a = {(int)threadIdx.x + 1, 2, 3, 4}, b = {5, 6, 7, (int)threadIdx.x};
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    v16i d0, d1;
    for (int i = 0; i < 16; ++i) { d0[i] = 0; d1[i] = 0; }
    v8s ha = {1, 2, 3, 4, 5, 6, 7, 8}, hb = {8, 7, 6, 5, 4, 3, 2, 1};
    v4f f0 = {0, 0, 0, 0}, f1 = f0, f2 = f0, f3 = f0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (WHICH == 0) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
        } else if (WHICH == 1) {
            d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, d1, 0, 0, 0);
        } else {
            f0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, f0, 0, 0, 0);
            f1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, f1, 0, 0, 0);
            f2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, f2, 0, 0, 0);
            f3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, f3, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5] + (int)(f0[0] + f1[1] + f2[2] + f3[3]);
}

int main()
{
    long long *out; int *sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 4096);
    const int iters = 20000;
    const char *names[3] = {"v_mfma_i32_16x16x64_i8", "v_mfma_i32_32x32x32_i8", "v_mfma_f32_16x16x32_bf16"};
    for (int w = 0; w < 3; ++w) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (w == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, iters, out, sink);
            else if (w == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, iters, out, sink);
            else hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, iters, out, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long t; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
            if (rep) printf("%s: %.1f s_memtime ticks per MFMA, %.2f ns per MFMA (event time; at 2.4 GHz = %.1f cycles)\n", names[w], (double)t / (4.0 * iters),
                            ms * 1e6 / (4.0 * iters), ms * 1e6 / (4.0 * iters) * 2.4);
        }
    }
    return 0;
}
