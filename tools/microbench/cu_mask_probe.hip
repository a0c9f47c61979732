// Which CUs does bit i of a hipExtStreamCreateWithCUMask mask stand for on this device?  For every single-bit mask a stream is
// created and a small grid launched on it; every workgroup records HW_REG_XCC_ID and HW_REG_HW_ID (shader engine, CU).  Then, for a
// candidate split (the first `reserve` bits for one stream, the rest for another), a grid of LDS-heavy workgroups (80 KB each, like
// the tower's) is launched on the large side and the workgroups per XCD are counted: a split that takes the same number of CUs
// out of every XCD keeps the round-robin dispatch balanced.
// build: hipcc -O3 --offload-arch=gfx950 cu_mask_probe.hip -o cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_where(unsigned *out, int spin)
{
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
    // stay a while so that the grid spreads over everything the mask allows
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
    if (spin < 0)
        lds[threadIdx.x] = 1;
}

static void decode(unsigned xcc, unsigned hw, int &x, int &se, int &sh, int &cu)
{
    x = xcc & 15;
    cu = (hw >> 8) & 15;
    sh = (hw >> 12) & 1;
    se = (hw >> 13) & 7;
}

int main(int argc, char **argv)
{
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("multiProcessorCount %d\n", cus);
    const int words = (cus + 31) / 32;
    unsigned *d;
    CK(hipMalloc((void **)&d, 2 * 8192 * 4));
    std::vector<unsigned> h(2 * 8192);
    CK(hipFuncSetAttribute((const void *)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    // 1. single bits
    for (int bit = 0; bit < cus; bit++) {
        std::vector<uint32_t> mask(words, 0u);
        mask[bit / 32] = 1u << (bit % 32);
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, words, mask.data()));
        hipLaunchKernelGGL(k_where, dim3(16), dim3(64), 0, s, d, 200);   // 2 us each
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, 2 * 16 * 4, hipMemcpyDeviceToHost));
        std::set<unsigned> seen;
        for (int i = 0; i < 16; i++) {
            int x, se, sh, cu;
            decode(h[2 * i], h[2 * i + 1], x, se, sh, cu);
            seen.insert((unsigned)(x << 12 | se << 8 | sh << 4 | cu));
        }
        printf("bit %3d ->", bit);
        for (unsigned v : seen)
            printf(" xcc %u se %u sh %u cu %u;", v >> 12, (v >> 8) & 15, (v >> 4) & 15, v & 15);
        printf("\n");
        CK(hipStreamDestroy(s));
    }
    // 2. splits: the first `reserve` bits against the rest, tower-like workgroups (80 KB of LDS, 256 threads) on the rest
    for (int reserve : {0, 8, 16, 32}) {
        std::vector<uint32_t> big(words, 0u), small(words, 0u);
        for (int b = 0; b < cus; b++)
            (b < reserve ? small : big)[b / 32] |= 1u << (b % 32);
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, words, big.data()));
        const int grid = 4096;
        hipLaunchKernelGGL(k_where, dim3(grid), dim3(256), 81920, s, d, 2000);  // 20 us each
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, 2 * grid * 4, hipMemcpyDeviceToHost));
        std::map<int, int> per_xcc;
        std::set<unsigned> cu_seen;
        for (int i = 0; i < grid; i++) {
            int x, se, sh, cu;
            decode(h[2 * i], h[2 * i + 1], x, se, sh, cu);
            per_xcc[x]++;
            cu_seen.insert((unsigned)(x << 12 | se << 8 | sh << 4 | cu));
        }
        printf("all but the first %d bits: %zu distinct CUs used; workgroups per XCD:", reserve, cu_seen.size());
        for (auto &kv : per_xcc)
            printf(" %d:%d", kv.first, kv.second);
        printf("\n");
        CK(hipStreamDestroy(s));
        if (reserve) {
            CK(hipExtStreamCreateWithCUMask(&s, words, small.data()));
            hipLaunchKernelGGL(k_where, dim3(1024), dim3(64), 0, s, d, 500);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h.data(), d, 2 * 1024 * 4, hipMemcpyDeviceToHost));
            per_xcc.clear();
            cu_seen.clear();
            for (int i = 0; i < 1024; i++) {
                int x, se, sh, cu;
                decode(h[2 * i], h[2 * i + 1], x, se, sh, cu);
                per_xcc[x]++;
                cu_seen.insert((unsigned)(x << 12 | se << 8 | sh << 4 | cu));
            }
            printf("the first %d bits: %zu distinct CUs used; workgroups per XCD:", reserve, cu_seen.size());
            for (auto &kv : per_xcc)
                printf(" %d:%d", kv.first, kv.second);
            printf("\n");
            CK(hipStreamDestroy(s));
        }
    }
    return 0;
}
