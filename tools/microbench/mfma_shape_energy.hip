// Which MFMA shape does more work inside the package power budget?  Whole chip, two waves per SIMD, nothing but MFMAs
// on random bf16 operands (four operand sets in rotation, 16 independent accumulators), long enough for the clock to
// settle.  Prints sustained TFLOP/s per shape; the faster one spends less energy per FLOP.
// build: hipcc -O3 --offload-arch=gfx950 mfma_shape_energy.hip -o mfma_shape_energy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline bf16x8 rnd_frag(unsigned &s, float scale, int zero_every)
{
    bf16x8 v;
    for (int j = 0; j < 8; j++) {
        s = s * 1664525u + 1013904223u;
        float f = ((float)(s >> 8) / 8388608.0f - 1.0f) * scale;
        if (zero_every && ((s >> 3) % zero_every) == 0)
            f = 0.0f;  // post-relu activations: about half zeros
        v[j] = (__bf16)f;
    }
    return v;
}

template <int SHAPE> __global__ __launch_bounds__(512) void k_mfma(int iters, float *sink)
{
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; i++) {
        a[i] = rnd_frag(s, 0.05f, 0);   // weights
        b[i] = rnd_frag(s, 1.0f, 2);    // activations, half of them zero
    }
    float out = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[16];
        for (int j = 0; j < 16; j++)
            acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 16; j++)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[(j >> 2) & 3], acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 16; j++)
            out += acc[j][0] + acc[j][3];
    } else {
        f32x16 acc[4];
        for (int j = 0; j < 4; j++)
            for (int i = 0; i < 16; i++)
                acc[j][i] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 2; r++)   // 8 MFMAs of 32 K flop = the 16 of 16 K above
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(j + r) & 3], b[j], acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; j++)
            out += acc[j][0] + acc[j][15];
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = out;
}

template <int SHAPE> static void run(const char *name, float *d_s)
{
    const int iters = 40000;   // x 16 x 16 K flop per wave
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = getenv("LONG") ? 24 : 3, launches = getenv("LONG") ? 32 : 4;   // LONG=1: ~8 s per shape, to sample rocm-smi beside it
    for (int rep = 0; rep < reps; rep++) {
        CK(hipEventRecord(e0));
        for (int l = 0; l < launches; l++)
            hipLaunchKernelGGL((k_mfma<SHAPE>), dim3(256), dim3(512), 0, 0, iters, d_s);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = (double)launches * 256 * 8 * (double)iters * 16 * 16384.0;
        printf("%s: %.1f ms, %.0f TFLOP/s\n", name, ms, flop / (ms * 1e-3) / 1e12);
    }
}

int main()
{
    float *d_s;
    CK(hipMalloc((void **)&d_s, 256 * 512 * 4));
    run<16>("v_mfma_f32_16x16x32_bf16", d_s);
    run<32>("v_mfma_f32_32x32x16_bf16", d_s);
    if (!getenv("LONG")) {
        run<16>("v_mfma_f32_16x16x32_bf16", d_s);
        run<32>("v_mfma_f32_32x32x16_bf16", d_s);
    }
    return 0;
}
