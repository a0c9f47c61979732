// How much matrix-pipe time do the instructions beside the MFMAs cost on gfx950?
// One k-step-like body: 20 independent v_mfma_f32_16x16x32_bf16, each followed by V plain VALU instructions (and,
// optionally, one ds_read_b128 per 4 MFMAs), issued from 1 or 2 waves per SIMD.  Prints shader cycles per MFMA
// (s_memtime inside the kernel, one workgroup on one CU) and the wall-clock rate with the whole chip busy.
// build: hipcc -O3 --offload-arch=gfx950 mfma_valu_mix.hip -o mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

template <int V, int LDSR>
__global__ __launch_bounds__(512) void k_mix(int iters, unsigned long long *cycles, float *sink)
{
    __shared__ uint4 lds[1024];
    f32x4 acc[20];
    for (int j = 0; j < 20; j++)
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int j = 0; j < 8; j++) {
        a[j] = (__bf16)(float)(threadIdx.x & 3);
        b[j] = (__bf16)1.0f;
    }
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    uint4 ld = make_uint4(0, 0, 0, 0);
    lds[threadIdx.x & 1023] = make_uint4(1, 2, 3, 4);
    __syncthreads();
    const unsigned laddr = (threadIdx.x & 63) * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 20; j++) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            if (V >= 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(x1));
            if (V >= 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x1) : "v"(x2));
            if (V >= 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x2) : "v"(x3));
            if (V >= 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x3) : "v"(x0));
            if (LDSR && (j & 3) == 0) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(laddr));
            }
        }
        if (LDSR)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 20; j++)
        s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    s += (float)(x0 + x1 + x2 + x3 + ld.x);
    if (threadIdx.x % 64 == 0)
        cycles[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V, int LDSR> static void run(int waves_per_simd)
{
    const int threads = 256 * waves_per_simd, iters = 2000;
    unsigned long long *d_c;
    float *d_s;
    CK(hipMalloc((void **)&d_c, 512 * 8 * 8));
    CK(hipMalloc((void **)&d_s, 512 * 512 * 4));
    // one workgroup: cycles per MFMA on one CU at an unloaded clock
    hipEvent_t s0, s1;
    CK(hipEventCreate(&s0));
    CK(hipEventCreate(&s1));
    hipLaunchKernelGGL((k_mix<V, LDSR>), dim3(1), dim3(threads), 0, 0, iters, d_c, d_s);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(s0));
    hipLaunchKernelGGL((k_mix<V, LDSR>), dim3(1), dim3(threads), 0, 0, 10 * iters, d_c, d_s);
    CK(hipEventRecord(s1));
    CK(hipEventSynchronize(s1));
    float one_ms;
    CK(hipEventElapsedTime(&one_ms, s0, s1));
    const double ns_per_mfma_wave = one_ms * 1e6 / (10.0 * iters * 20.0);
    hipLaunchKernelGGL((k_mix<V, LDSR>), dim3(1), dim3(threads), 0, 0, iters, d_c, d_s);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> c(8);
    CK(hipMemcpy(c.data(), d_c, 64, hipMemcpyDeviceToHost));
    const double cyc = (double)c[0] / (iters * 20.0);          // cycles per MFMA of one wave
    // whole chip: 256 workgroups, wall clock
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mix<V, LDSR>), dim3(256), dim3(threads), 0, 0, iters, d_c, d_s);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++)
        hipLaunchKernelGGL((k_mix<V, LDSR>), dim3(256), dim3(threads), 0, 0, iters, d_c, d_s);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfmas = 5.0 * 256 * (threads / 64) * iters * 20.0;
    const double tf = mfmas * 16384.0 / (ms * 1e-3) / 1e12;
    CK(hipMemcpy(c.data(), d_c, 64, hipMemcpyDeviceToHost));
    const double cyc_loaded = (double)c[0] / (iters * 20.0);
    printf("VALU/MFMA %d  ds_read/4 MFMA %d  waves/SIMD %d : %.1f ticks = %.2f ns per MFMA per wave (one CU: %.2f GHz) -> pipe busy %.0f %%;  whole chip %.0f TFLOP/s (%.0f %% of 2500), %.1f cycles per MFMA per wave\n",
           V, LDSR, waves_per_simd, cyc, ns_per_mfma_wave, cyc / ns_per_mfma_wave, 100.0 * 16.0 * waves_per_simd / cyc, tf, 100.0 * tf / 2500.0, cyc_loaded);
    (void)hipFree(d_c);
    (void)hipFree(d_s);
}

int main()
{
    for (int w = 1; w <= 2; w++) {
        run<0, 0>(w);
        run<1, 0>(w);
        run<2, 0>(w);
        run<3, 0>(w);
        run<4, 0>(w);
        run<2, 1>(w);
    }
    return 0;
}
