// The roofline a batched tree descent can be held against: dependent chains of scattered reads, one chain per wave,
// each step one load of `bytes_per_lane` bytes in `lanes` lanes (a node's children: 16 B x ~42 lanes in the engine) from a
// position that depends on the data just read (a PUCT level: the next node is only known once this node's children are
// scored).  Reports, for W chains in flight over a buffer of a few GiB, the time per dependent step and the bytes per
// second of the whole chip — i.e. what HBM3E and the fabric deliver for THIS access pattern, to put beside the 8 TB/s of
// streaming reads.  build: hipcc -O3 --offload-arch=gfx950 random_chase.hip -o random_chase ; run: ./random_chase [GiB [reads per step: 1 | 2 [lanes,lanes,...]]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one record = 1 KiB (64 x 16 B): a node's child list; a region = `region_records` consecutive records (a game's arena)
// `reads` = 2: every step issues a second, independent read of the same shape (another record of the chain's region,
// its position known when the first is issued) — do two requests of a wave in flight cost what one costs?
__global__ __launch_bounds__(256) void k_chase(const uint4 *__restrict__ buf, unsigned long long n_regions, unsigned region_records,
                                              int steps, int lanes, int waves_total, unsigned seed, unsigned *sink, int reads)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= waves_total)
        return;
    // every chain stays inside its own region, like a game inside its arena (regions are spread over the whole buffer)
    const unsigned long long region = ((unsigned long long)wave * 2654435761ull) % n_regions;
    unsigned rec = ((unsigned)wave * 40503u + seed * 977u) % region_records;
    unsigned acc = 0;
    for (int s = 0; s < steps; s++) {
        // (lanes beyond `lanes` re-read lane 0's 16 bytes: no request of their own, and no branch around the loads — a
        // load under its own branch is waited for before the branch ends, which would serialise the two reads)
        const int l = lane < lanes ? lane : 0;
        const unsigned rec2 = reads == 2 ? (rec * 7u + 13u + (unsigned)s) % region_records : rec;
        const uint4 v = buf[(region * region_records + rec) * 64ull + l];
        const uint4 v2 = buf[(region * region_records + rec2) * 64ull + l];
        acc += v2.x ^ v2.y ^ v2.z ^ v2.w;
        // a little arithmetic on what was read, then the next position from it (lane 0's word, like the chosen child's id)
        acc += v.x ^ (v.y >> 3) ^ v.w;
        // (the step number and the launch's seed go into the hash: a pure function of the record read would be a random
        // map on 612 elements, whose chains fall into cycles of ~25 records within ~30 steps and then live in the caches)
        unsigned nxt = (unsigned)__builtin_amdgcn_readfirstlane((int)(v.x + v.z)) + (unsigned)s * 0x9E3779B9u + seed * 0x85EBCA6Bu;
        nxt ^= nxt >> 15; nxt *= 0x2C1B3C6Du; nxt ^= nxt >> 12;
        rec = nxt % region_records;
    }
    if (acc == 0xDEADBEEFu)
        *sink = acc;
}

__global__ void k_fill(uint4 *buf, unsigned long long n)
{
    unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long x = i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        buf[i] = make_uint4((unsigned)x, (unsigned)(x >> 32), (unsigned)(x * 31), (unsigned)(x >> 17));
    }
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const int reads = argc > 2 ? atoi(argv[2]) : 1;
    const unsigned region_records = 612;                    // 612 KiB: one game's edge arena at 400 sims (39,168 edges x 16 B)
    const unsigned long long n_regions = (unsigned long long)(gib * 1024.0 * 1024.0 / region_records);
    const unsigned long long n = n_regions * region_records * 64ull;
    uint4 *buf;
    unsigned *sink;
    CK(hipMalloc((void **)&buf, n * 16));
    CK(hipMalloc((void **)&sink, 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, n);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("dependent scattered reads over %.1f GiB in %llu regions of %u KiB, one chain per wave, 48 steps per chain, %d read(s) per step\n",
           gib, n_regions, region_records, reads);
    printf("%6s %6s %8s | %10s %12s %12s\n", "waves", "lanes", "B/lane", "us/step", "GB/s (req.)", "GB/s (64B sectors)");
    const int steps = 48;
    // lanes per request: the engine's 42 (a node's children) and a full wave; `./random_chase GiB reads 4,8,16` measures
    // other request sizes (what a level would cost if it read only a node's VISITED children: DESIGN.md section 8)
    std::vector<int> lane_list = {42, 64};
    if (argc > 3) {
        lane_list.clear();
        for (const char *p = argv[3]; *p;) {
            lane_list.push_back(atoi(p));
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
    }
    for (int lanes : lane_list) {
        for (int waves : {256, 1024, 2048, 4096, 8192, 16384, 32768}) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_chase, dim3((waves + 3) / 4), dim3(256), 0, 0, (const uint4 *)buf, n_regions, region_records, steps,
                                   lanes, waves, (unsigned)(rep * 7 + lanes), sink, reads);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best)
                    best = ms;
            }
            const double bytes = (double)waves * steps * lanes * 16.0 * reads;
            const double sectors = (double)waves * steps * ((lanes * 16 + 63) / 64) * 64.0 * reads;
            // a launch of more waves than fit (8192) runs in rounds: us/step is per chain, over the rounds it needs
            const double rounds = waves > 8192 ? waves / 8192.0 : 1.0;
            printf("%6d %6d %8d | %10.3f %12.0f %12.0f\n", waves, lanes, 16, best * 1e3 / steps / rounds, bytes / (best * 1e-3) / 1e9,
                   sectors / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
