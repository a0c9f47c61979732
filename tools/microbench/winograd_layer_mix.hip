// What would a Winograd F(2x2, 3x3) tower layer cost on gfx950?  A stand-in with the instruction mix of such a layer, priced
// BEFORE any kernel is written (round-3 review, item 4: the arithmetic reduction is 1,536 v_mfma_f32_16x16x32 per layer and
// 3-board workgroup against about 2,400 for the direct tower; what it adds is the input / output transforms' vector adds,
// their LDS traffic, 1.78x the weight stream, eight barriers per layer instead of one, and an LDS footprint that leaves room
// for ONE workgroup per CU).
//
// One workgroup = 4 waves = 3 boards = 48 tiles of 2x2 outputs, as the design on paper ("Winograd, priced"):
//   LDS   activations [2][16 units of 8 channels][160 slots] f16 (80 KB, the direct tower's image) + V [4 positions][16 units]
//         [48 tiles] f16 (48 KB): the transformed input of ONE row group of the 4x4 positions — all sixteen are 196 KB.
//         (Lay-outs chosen so that a B fragment's 16 lanes read 256 contiguous bytes and the transform writes contiguous
//         tiles; with the tile-major lay-out of a first attempt the same mix ran 16-way bank-conflicted.)
//   per layer, four passes (row group a = 0..3 of the 4x4 transform):
//     transform  every lane takes 3 (tile, unit) items: 8 ds_read_b128 (two rows of the 4x4 patch), 32 v_pk_add_f16
//                (t_j = d[i1][j] +- d[i2][j]; V_b = t_j1 +- t_j2), 4 ds_write_b128 (positions (a, 0..3))
//     barrier
//     products   a wave owns 2 output-channel tiles x 3 tile tiles = 6 accumulator tiles per position: per position and
//                k-step 3 ds_read_b128 (B fragments), 2 global_load_dwordx4 (A fragments of U, L2-resident), 6 MFMAs;
//                4 positions x 4 k-steps = 96 MFMAs
//     fold       A^T M A for this row group into the 2x2 output accumulators: 28 v_add_f32 per accumulator register group
//     barrier
//   epilogue     shift, relu, convert, 24 ds_write_b64 into the other activation image (as the direct tower's)
// Printed: microseconds per layer and workgroup, and for the whole "tower" of 24 such layers over N boards — to be read
// against the real direct tower on the same box in the same call (tools/net_bench.py).  Random f16 data (the chip is power
// limited: constant operands flatter it).  This is a mix, not a convolution: the addresses are realistic in shape and bank
// behaviour, the numbers that come out mean nothing.
// build: hipcc -O3 --offload-arch=gfx950 winograd_layer_mix.hip -o winograd_layer_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int CELLS = 147, UNITS = 16, TILES = 48;
constexpr int CS = 160;                        // slots per unit of an activation image ([unit][cell], as the direct tower's image)
constexpr int ACT_U4 = CS * UNITS;             // uint4 per activation image
constexpr int V_U4 = 4 * UNITS * TILES;        // uint4 in the V buffer: [position][unit][tile] — a B fragment's 16 lanes read 16
                                               // consecutive tiles (256 contiguous bytes: no bank conflict)
constexpr int LDS_BYTES = (2 * ACT_U4 + V_U4) * 16;

union U4H {
    uint4 u;
    h8 h;
    h2 p[4];
};

__global__ __launch_bounds__(256) void k_wino(const uint4 *__restrict__ U, int layers, int boards, float *sink, int phases)
{
    extern __shared__ uint4 lds[];
    uint4 *act = lds;                // [2][16 units][160 slots]
    uint4 *V = lds + 2 * ACT_U4;     // [4][16 units][48 tiles]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // pseudo-random activations
    for (int i = t; i < 2 * ACT_U4; i += 256) {
        unsigned s = (unsigned)(i * 2654435761u) ^ (blockIdx.x * 40503u);
        U4H x;
        for (int k = 0; k < 8; k++) {
            s = s * 1664525u + 1013904223u;
            x.h[k] = (_Float16)((float)((s >> 9) & 0xFF) * (1.0f / 256.0f) - 0.4f);
        }
        act[i] = x.u;
    }
    __syncthreads();
    f4 Y[6][4];
    int cur = 0;
    for (int layer = 0; layer < layers; layer++) {
#pragma unroll
        for (int u = 0; u < 6; u++)
#pragma unroll
            for (int o = 0; o < 4; o++)
                Y[u][o] = f4{0.01f, 0.02f, 0.03f, 0.04f};
        const uint4 *Ul = U + (size_t)layer * (16 * 8 * 4 * 64);
        const uint4 *src = act + cur * ACT_U4;
        for (int a = 0; a < 4; a++) {
            // ---- input transform of row group a: 3 (tile, unit) items per lane
            if (phases & 1)
#pragma unroll
            for (int it = 0; it < 3; it++) {
                const int item = t + 256 * it;          // 0..767: consecutive lanes take consecutive tiles of one unit
                const int tile = item % TILES, unit = item / TILES;
                const int board = tile >> 4, ty = (tile >> 2) & 3, tx = tile & 3;
                // rows of B^T for row group a: (0: d0 - d2), (1: d1 + d2), (2: d2 - d1), (3: d1 - d3)
                const int i1 = a == 0 ? 0 : 1, i2 = a == 3 ? 3 : 2;
                U4H d1[4], d2[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int y1 = 2 * ty - 1 + i1, y2 = 2 * ty - 1 + i2, x = 2 * tx - 1 + j;
                    y1 = y1 < 0 ? 0 : (y1 > 6 ? 6 : y1);
                    y2 = y2 < 0 ? 0 : (y2 > 6 ? 6 : y2);
                    x = x < 0 ? 0 : (x > 6 ? 6 : x);
                    d1[j].u = src[unit * CS + 21 * y1 + 7 * board + x];
                    d2[j].u = src[unit * CS + 21 * y2 + 7 * board + x];
                }
                U4H tr[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        tr[j].p[q] = (a == 1) ? d1[j].p[q] + d2[j].p[q] : d1[j].p[q] - d2[j].p[q];
                U4H vb[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    vb[0].p[q] = tr[0].p[q] - tr[2].p[q];
                    vb[1].p[q] = tr[1].p[q] + tr[2].p[q];
                    vb[2].p[q] = tr[2].p[q] - tr[1].p[q];
                    vb[3].p[q] = tr[1].p[q] - tr[3].p[q];
                }
#pragma unroll
                for (int b = 0; b < 4; b++)
                    V[(b * UNITS + unit) * TILES + tile] = vb[b].u;
            }
            __syncthreads();
            // ---- products: 4 positions x 4 k-steps x (2 oc tiles x 3 tile tiles); the A fragments (U, from L2) are requested
            // two steps ahead in a register ring, the B fragments (V, from LDS) one step ahead — as the direct tower's k-loop does
            f4 M[4][6];
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int u = 0; u < 6; u++)
                    M[b][u] = f4{0.f, 0.f, 0.f, 0.f};
            U4H A[3][2], B[2][3];
            auto load_a = [&](int step, U4H *dst) {
                const int pos = a * 4 + (step >> 2), ks = step & 3;
                dst[0].u = Ul[((pos * 8 + 2 * w) * 4 + ks) * 64 + lane];
                dst[1].u = Ul[((pos * 8 + 2 * w + 1) * 4 + ks) * 64 + lane];
            };
            auto load_b = [&](int step, U4H *dst) {
                const int b = step >> 2, ks = step & 3;
#pragma unroll
                for (int tt = 0; tt < 3; tt++)
                    dst[tt].u = V[(b * UNITS + ks * 4 + (lane >> 4)) * TILES + tt * 16 + (lane & 15)];
            };
            if (phases & 2) {
            load_a(0, A[0]);
            load_a(1, A[1]);
            load_b(0, B[0]);
#pragma unroll
            for (int step = 0; step < 16; step++) {
                if (step + 2 < 16)
                    load_a(step + 2, A[(step + 2) % 3]);
                if (step + 1 < 16)
                    load_b(step + 1, B[(step + 1) & 1]);
                const int b = step >> 2;
#pragma unroll
                for (int tt = 0; tt < 3; tt++) {
                    M[b][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[step % 3][0].h, B[step & 1][tt].h, M[b][tt], 0, 0, 0);
                    M[b][3 + tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[step % 3][1].h, B[step & 1][tt].h, M[b][3 + tt], 0, 0, 0);
                }
            }
            }
            // ---- fold row group a of A^T M A into the 2x2 outputs
            if (phases & 4)
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const f4 r0 = M[0][u] + M[1][u] + M[2][u];
                const f4 r1 = M[1][u] - M[2][u] - M[3][u];
                if (a <= 2) {
                    Y[u][0] += r0;
                    Y[u][1] += r1;
                }
                if (a == 1) {
                    Y[u][2] += r0;
                    Y[u][3] += r1;
                } else if (a >= 2) {
                    Y[u][2] -= r0;
                    Y[u][3] -= r1;
                }
            }
            __syncthreads();
        }
        // ---- epilogue: relu, convert, write the other activation image (8-byte runs of four channels, as the direct tower)
        uint4 *dst = act + (1 - cur) * ACT_U4;
        if (phases & 8)
#pragma unroll
        for (int u = 0; u < 6; u++)
#pragma unroll
            for (int o = 0; o < 4; o++) {
                f4 y = Y[u][o];
                y.x = y.x > 0.f ? y.x : 0.f;
                y.y = y.y > 0.f ? y.y : 0.f;
                y.z = y.z > 0.f ? y.z : 0.f;
                y.w = y.w > 0.f ? y.w : 0.f;
                h2 lo = {(_Float16)(y.x * 0.01f), (_Float16)(y.y * 0.01f)}, hi = {(_Float16)(y.z * 0.01f), (_Float16)(y.w * 0.01f)};
                const int cell = ((u % 3) * 16 + (lane & 15)) * 3 + o;   // some cell of the image (< 147)
                const int unit = (2 * w + u / 3) * 2 + ((lane >> 4) >> 1);
                uint2 *p = reinterpret_cast<uint2 *>(dst + unit * CS + cell % CELLS) + ((lane >> 4) & 1);
                uint2 val;
                __builtin_memcpy(&val.x, &lo, 4);
                __builtin_memcpy(&val.y, &hi, 4);
                *p = val;
            }
        cur = 1 - cur;
        __syncthreads();
    }
    float s = 0.f;
    U4H x;
    x.u = act[cur * ACT_U4 + t];
    for (int k = 0; k < 8; k++)
        s += (float)x.h[k];
    if (blockIdx.x * 3 < boards)
        sink[blockIdx.x * 256 + t] = s;
}

int main(int argc, char **argv)
{
    const int boards = argc > 1 ? atoi(argv[1]) : 3600;
    const int layers = argc > 2 ? atoi(argv[2]) : 24;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    const int wgs = (boards + 2) / 3;
    const size_t u_count = (size_t)layers * 16 * 8 * 4 * 64;   // uint4
    std::vector<uint4> hU(u_count);
    unsigned s = 12345u;
    for (size_t i = 0; i < u_count; i++) {
        U4H x;
        for (int k = 0; k < 8; k++) {
            s = s * 1664525u + 1013904223u;
            x.h[k] = (_Float16)(((float)((s >> 9) & 0xFF) * (1.0f / 256.0f) - 0.5f) * 0.05f);
        }
        hU[i] = x.u;
    }
    uint4 *dU;
    float *dS;
    CK(hipMalloc((void **)&dU, u_count * 16));
    CK(hipMemcpy(dU, hU.data(), u_count * 16, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&dS, (size_t)wgs * 256 * 4));
    CK(hipFuncSetAttribute((const void *)k_wino, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct { int mask; const char *what; } runs[] = {{15, "whole layer"}, {1, "input transform only"}, {2, "products only (fragment loads + MFMAs)"},
                                                     {4, "output fold only"}, {8, "epilogue only"}, {0, "barriers and loop only"}};
    for (auto &run : runs) {
        for (int r = 0; r < 2; r++)
            hipLaunchKernelGGL(k_wino, dim3(wgs), dim3(256), LDS_BYTES, 0, (const uint4 *)dU, layers, boards, dS, run.mask);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++)
            hipLaunchKernelGGL(k_wino, dim3(wgs), dim3(256), LDS_BYTES, 0, (const uint4 *)dU, layers, boards, dS, run.mask);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double rounds = (double)wgs / 256.0;   // one workgroup per CU (128 KB of LDS)
        // one workgroup alone: the per-layer time without the other CUs' L2 traffic
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++)
            hipLaunchKernelGGL(k_wino, dim3(1), dim3(256), LDS_BYTES, 0, (const uint4 *)dU, layers, 3, dS, run.mask);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms1;
        CK(hipEventElapsedTime(&ms1, e0, e1));
        printf("%-42s %d boards = %d workgroups (%.2f rounds of 256), %d layers: %.3f ms per launch, %.2f us per layer and workgroup "
               "round; one workgroup alone %.2f us per layer\n", run.what, boards, wgs, rounds, layers, ms,
               1e3 * ms / layers / (rounds < 1 ? 1 : rounds), 1e3 * ms1 / reps / layers);
    }
    printf("(U stream %.1f MB, LDS %d bytes per workgroup)\n", u_count * 16 / 1e6, LDS_BYTES);
    return 0;
}
