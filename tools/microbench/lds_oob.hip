// Does an LDS read beyond the workgroup's allocation return zero on gfx950 (and nothing else happens)?
// One workgroup, 4 KiB of dynamic LDS filled with ones; every lane reads 16 bytes at several byte addresses past
// the end (up to bit 28 set) and, as a control, inside it.
// build: hipcc -O3 --offload-arch=gfx950 lds_oob.hip -o lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_oob(unsigned *out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x)
        reinterpret_cast<unsigned *>(smem)[i] = 0x01010101u;
    __syncthreads();
    const unsigned addrs[6] = {16u * threadIdx.x, 4096u + 16u * threadIdx.x, 81920u, 163840u + 16u * threadIdx.x, (1u << 28) + 16u * threadIdx.x,
                               (1u << 28) + 2000u};
    for (int k = 0; k < 6; k++) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:61440\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addrs[k]) : "memory");
        u32x4 w;
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addrs[k]) : "memory");
        out[(k * 64 + threadIdx.x) * 2 + 0] = v.x | v.y | v.z | v.w;
        out[(k * 64 + threadIdx.x) * 2 + 1] = w.x | w.y | w.z | w.w;
    }
}

int main()
{
    unsigned *d, h[6 * 64 * 2];
    CK(hipMalloc((void **)&d, sizeof(h)));
    CK(hipMemset(d, 0xFF, sizeof(h)));
    hipLaunchKernelGGL(k_oob, dim3(1), dim3(64), 4096, 0, d);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const char *what[6] = {"inside (0..1023)", "just past the end (4096..)", "at 80 KiB", "at 160 KiB", "bit 28 set", "bit 28 set + 2000"};
    for (int k = 0; k < 6; k++) {
        unsigned any_off = 0, any = 0;
        for (int l = 0; l < 64; l++) {
            any_off |= h[(k * 64 + l) * 2];
            any |= h[(k * 64 + l) * 2 + 1];
        }
        printf("%-28s: with offset:61440 -> %08x   without offset -> %08x\n", what[k], any_off, any);
    }
    return 0;
}
