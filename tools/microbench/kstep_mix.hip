// The tower's k-step as an instruction mix, without its data flow: per 20 v_mfma_f32_16x16x32_bf16 (independent
// accumulators) NDS ds_read_b128, NVM global_load_dwordx4 (L2-resident window) and NV plain VALU instructions, all
// consumed one body later (waits at the top of the body: whatever was issued a body ago has had ~320 cycles to land).
// One workgroup of 4 or 8 waves on one CU; wall clock per MFMA per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 kstep_mix.hip -o kstep_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NDS, int NVM, int NV, int AHEAD>
__global__ __launch_bounds__(512) void k_mix(int iters, const uint4 *wts, unsigned long long *cycles, float *sink)
{
    __shared__ uint4 lds[4096];
    f32x4 acc[20];
    for (int j = 0; j < 20; j++)
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int j = 0; j < 8; j++) {
        a[j] = (__bf16)(float)(threadIdx.x & 3);
        b[j] = (__bf16)1.0f;
    }
    unsigned x0 = threadIdx.x, x1 = 1;
    u32x4 ld[5], vm[2][4];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x)
        lds[i] = make_uint4(i, 2, 3, 4);
    __syncthreads();
    const unsigned laddr = (threadIdx.x & 63) * 16 + (threadIdx.x / 64) * 1024;
    const int wave = threadIdx.x / 64;
    const uint4 *wp = wts + (size_t)(blockIdx.x * 8 + wave) * 16384 + (threadIdx.x & 63);   // 256 KB window per wave
    for (int k = 0; k < 5; k++)
        ld[k] = u32x4{0, 0, 0, 0};
    for (int k = 0; k < 4; k++)
        vm[0][k] = vm[1][k] = u32x4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // A load's destination stays live (it is consumed after the wait, a body later): the compiler must not hand the
    // register to anything else while the load is in flight.
#define BODY(SET, IT)                                                                                                  \
    {                                                                                                                  \
        if (NDS) {                                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
            _Pragma("unroll") for (int k = 0; k < NDS; k++)                                                            \
                asm volatile("; use %0" :: "v"(ld[k]));                                                                \
        }                                                                                                              \
        if (NVM) {                                                                                                     \
            if (AHEAD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NVM) : "memory");                                           \
            _Pragma("unroll") for (int k = 0; k < NVM; k++)                                                            \
                asm volatile("; use %0" :: "v"(vm[SET][k]));                                                           \
        }                                                                                                              \
        const uint4 *p = wp + (size_t)(((IT) * 4) & 127) * 64; /* + 3 * 64 + 63 < 16384: inside the wave's window */   \
        _Pragma("unroll") for (int j = 0; j < 20; j++) {                                                               \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));                   \
            if (j < NV) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(x1));                                     \
            if ((j & 3) == 1 && j / 4 < NDS)                                                                           \
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[j / 4]) : "v"(laddr), "n"((j / 4) * 4096));     \
            if ((j & 3) == 3 && j / 4 < NVM)                                                                           \
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vm[SET][j / 4]) : "v"(p + (j / 4) * 64));        \
        }                                                                                                              \
    }
    for (int it = 0; it < iters; it += 2) {
        BODY(0, it)
        BODY(1, it + 1)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 20; j++)
        s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    s += (float)(x0 + x1 + ld[0].x + ld[1].x + ld[2].x + ld[3].x + ld[4].x + vm[0][0].x + vm[0][1].x + vm[0][2].x + vm[0][3].x + vm[1][0].x + vm[1][1].x + vm[1][2].x + vm[1][3].x);
    if (threadIdx.x % 64 == 0)
        cycles[blockIdx.x * 8 + wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static uint4 *d_w;
static unsigned long long *d_c;
static float *d_s;

template <int NDS, int NVM, int NV, int AHEAD> static void run(int waves_per_simd, int grid)
{
    const int threads = 256 * waves_per_simd, iters = 4000;
    hipEvent_t s0, s1;
    CK(hipEventCreate(&s0));
    CK(hipEventCreate(&s1));
    hipLaunchKernelGGL((k_mix<NDS, NVM, NV, AHEAD>), dim3(grid), dim3(threads), 0, 0, iters, d_w, d_c, d_s);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(s0));
    hipLaunchKernelGGL((k_mix<NDS, NVM, NV, AHEAD>), dim3(grid), dim3(threads), 0, 0, iters, d_w, d_c, d_s);
    CK(hipEventRecord(s1));
    CK(hipEventSynchronize(s1));
    float ms;
    CK(hipEventElapsedTime(&ms, s0, s1));
    std::vector<unsigned long long> c(8);
    CK(hipMemcpy(c.data(), d_c, 64, hipMemcpyDeviceToHost));
    unsigned long long cmax = 0;
    for (int i = 0; i < 4 * waves_per_simd; i++)
        cmax = c[i] > cmax ? c[i] : cmax;
    const double per_simd = (double)iters * 20.0 * waves_per_simd;   // MFMAs per SIMD
    printf("ds_read %d  vmem %d (%d ahead)  valu %d per 20 MFMA, %d wave(s)/SIMD, %3d workgroups: %.2f ns per MFMA per SIMD, slowest wave %.1f ticks per own MFMA",
           NDS, NVM, AHEAD, NV, waves_per_simd, grid, ms * 1e6 / per_simd, (double)cmax / (iters * 20.0));
    if (grid > 1)
        printf("  -> %.0f TFLOP/s", (double)grid * 4 * per_simd * 16384.0 / (ms * 1e-3) / 1e12);
    printf("\n");
}

int main()
{
    CK(hipMalloc((void **)&d_w, (size_t)256 * 8 * 16384 * 16));
    CK(hipMemset(d_w, 1, (size_t)256 * 8 * 16384 * 16));
    CK(hipMalloc((void **)&d_c, 512 * 8 * 8));
    CK(hipMalloc((void **)&d_s, 512 * 512 * 4));
    for (int w = 1; w <= 2; w++)
        for (int grid : {1, 256}) {
            run<0, 0, 0, 1>(w, grid);
            run<5, 0, 0, 1>(w, grid);
            run<0, 4, 0, 1>(w, grid);
            run<0, 4, 0, 2>(w, grid);
            run<5, 4, 0, 2>(w, grid);
            run<5, 4, 10, 2>(w, grid);
            run<5, 4, 20, 2>(w, grid);
        }
    return 0;
}
