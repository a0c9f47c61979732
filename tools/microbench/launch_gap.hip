// Time per dependent kernel boundary on one stream: plain launches against the same chain replayed from a hipGraph.
// build: hipcc -O3 --offload-arch=gfx950 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_small(int *p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] += 1;
}

int main()
{
    int *d;
    const int n = 4096 * 64;
    CK(hipMalloc((void **)&d, n * 4));
    CK(hipMemset(d, 0, n * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int K = 300;
    for (int grid : {1, 4096}) {
        // plain launches
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < K; i++)
                hipLaunchKernelGGL(k_small, dim3(grid), dim3(64), 0, s, d, n);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("grid %4d: plain launches  %.2f us per kernel\n", grid, ms * 1e3 / K);
        // graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < K; i++)
            hipLaunchKernelGGL(k_small, dim3(grid), dim3(64), 0, s, d, n);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
        }
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("grid %4d: graph replay    %.2f us per kernel\n", grid, ms * 1e3 / K);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
