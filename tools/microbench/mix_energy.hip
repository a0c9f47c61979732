// What do the operand streams beside the MFMAs cost inside the package power budget?  Whole chip, two waves per SIMD,
// v_mfma_f32_16x16x32_bf16 on random operands, plus per 20 MFMAs: NDS ds_read_b128 (1 KiB per wave each) and NVM
// 16-byte global loads (1 KiB per wave each) from a 288 KiB window that ALL workgroups share and walk together, like
// the tower's packed weights (L2-resident).  Loads are consumed one body later; the loaded values are not used by the
// MFMAs (the operand registers stay random).  Sustained TFLOP/s per mix; the drop against the MFMA-only line is the
// energy the stream takes out of the budget.
// build: hipcc -O3 --offload-arch=gfx950 mix_energy.hip -o mix_energy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline bf16x8 rnd_frag(unsigned &s, float scale, int zero_every)
{
    bf16x8 v;
    for (int j = 0; j < 8; j++) {
        s = s * 1664525u + 1013904223u;
        float f = ((float)(s >> 8) / 8388608.0f - 1.0f) * scale;
        if (zero_every && ((s >> 3) % zero_every) == 0)
            f = 0.0f;
        v[j] = (__bf16)f;
    }
    return v;
}

constexpr int WINDOW = 18432;   // uint4 elements = 288 KiB, one tower layer's weights

template <int NDS, int NVM, int NV>
__global__ __launch_bounds__(512) void k_mix(int iters, const u32x4 *wts, float *sink)
{
    __shared__ u32x4 lds[4096];   // 64 KiB
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; i++) {
        a[i] = rnd_frag(s, 0.05f, 0);
        b[i] = rnd_frag(s, 1.0f, 2);
    }
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) {
        s = s * 1664525u + 1013904223u;
        lds[i] = u32x4{s, s * 3u, s * 5u, s * 7u};
    }
    __syncthreads();
    f32x4 acc[20];
    for (int j = 0; j < 20; j++)
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 ld[8], vm[2][8];
    for (int k = 0; k < 8; k++)
        ld[k] = vm[0][k] = vm[1][k] = u32x4{0, 0, 0, 0};
    unsigned x0 = threadIdx.x, x1 = 1;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned laddr = lane * 16 + (wave & 3) * 1024;
    // the lane's 16 bytes inside the 1 KiB fragment; waves 0/1, 2/3, ... read the same fragments (as the tower's two cell halves do)
    const u32x4 *wp = wts + lane + (wave >> 1 & 1) * 64 * 4;
#define BODY(SET, IT)                                                                                                  \
    {                                                                                                                  \
        if (NDS) {                                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
            _Pragma("unroll") for (int k = 0; k < NDS; k++) asm volatile("; use %0" ::"v"(ld[k]));                     \
        }                                                                                                              \
        if (NVM) {                                                                                                     \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NVM) : "memory");                                                 \
            _Pragma("unroll") for (int k = 0; k < NVM; k++) asm volatile("; use %0" ::"v"(vm[SET][k]));                \
        }                                                                                                              \
        const u32x4 *p = wp + (size_t)(((IT) * 8 * 64) % (WINDOW - 8 * 64 - 64 * 8)); /* stays inside the window */    \
        _Pragma("unroll") for (int j = 0; j < 20; j++) {                                                               \
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[(j >> 2) & 3], acc[j], 0, 0, 0);              \
            if (j < NV) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(x1));                                     \
            if ((j & 1) == 1 && j / 2 < NDS)                                                                           \
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[j / 2]) : "v"(laddr), "n"((j / 2) * 4096));     \
            if ((j & 1) == 0 && j / 2 < NVM)                                                                           \
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vm[SET][j / 2]) : "v"(p + (j / 2) * 64));        \
        }                                                                                                              \
    }
    for (int it = 0; it < iters; it += 2) {
        BODY(0, it)
        BODY(1, it + 1)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float out = (float)(x0 + x1);
    for (int j = 0; j < 20; j++)
        out += acc[j][0] + acc[j][3];
    for (int k = 0; k < 8; k++)
        out += (float)(ld[k].x + vm[0][k].y + vm[1][k].z);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = out;
}

static u32x4 *d_w;
static float *d_s;

template <int NDS, int NVM, int NV> static void run()
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, last = 0.f;
    const int launches = 16;   // ~0.4 s per repetition: long enough for the power controller to settle
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        for (int l = 0; l < launches; l++)
            hipLaunchKernelGGL((k_mix<NDS, NVM, NV>), dim3(256), dim3(512), 0, 0, iters, d_w, d_s);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&last, e0, e1));
        best = last < best ? last : best;
    }
    const double flop = (double)launches * 256 * 8 * (double)iters * 20 * 16384.0;
    printf("per 20 MFMA: %d ds_read_b128, %d global_load_dwordx4, %2d valu : %7.1f ms (last of 4), %5.0f TFLOP/s;  LDS %.1f TB/s, L2->VGPR %.1f TB/s\n",
           NDS, NVM, NV, last, flop / (last * 1e-3) / 1e12, (double)launches * 256 * 8 * (double)iters * NDS * 1024 / (last * 1e-3) / 1e12,
           (double)launches * 256 * 8 * (double)iters * NVM * 1024 / (last * 1e-3) / 1e12);
}

int main()
{
    CK(hipMalloc((void **)&d_w, (size_t)WINDOW * 16));
    CK(hipMemset(d_w, 0x3c, (size_t)WINDOW * 16));
    CK(hipMalloc((void **)&d_s, 256 * 512 * 4));
    run<0, 0, 0>();
    run<5, 0, 0>();
    run<8, 0, 0>();
    run<0, 4, 0>();
    run<0, 8, 0>();
    run<5, 4, 0>();
    run<5, 4, 10>();
    run<5, 4, 20>();
    run<8, 2, 10>();
    run<0, 0, 0>();
    return 0;
}
