// Select-only prototype of SPARSE CHILD RECORDS (the round-5 review's item 3): the PUCT descent of the
// tree kernels replayed over real trees of bench.py's steady state (tools/sparse_select_dump.py) in two layouts —
//
//   A  today's: a node's children are M consecutive 16-byte records (prior | mark, W, visits | child, the child's range);
//      a level loads all of them, one per lane (two per lane for 65-128 moves): ~11 cache lines for a mid-game node
//   B  sparse: a node is a block [header][v records of VISITED children, 16 B each][... free ...][u (prior, move) pairs of
//      the unvisited children, 8 B each, sorted by prior, best first from the block's end backwards].  Among a node's unvisited
//      children only the one with the largest prior can win (score c P sqrt(1 + N), Q = 0), and the header carries a copy of
//      it: a level loads header + visited records — ONE request of 8 lanes x 16 B (2 cache lines) for nodes with at most
//      7 visited children, 64 lanes for the others (a "wide" bit in the parent's record says which before the request is
//      made).  Capacity (M + 1) x 16 B per node: 16 + 16 v + 8 u never exceeds it, a first visit moves nothing (the new
//      record goes where the free space is, the best pair leaves the far end), arenas grow by one record per node.
//
// Both kernels select with the engine's arithmetic (puct_score below is engine.hip's, bit for bit) and the engine's tie rule
// (last maximal edge in movegen order), make the early request of the marked child, and must choose the same edges: the
// host compares a hash of every game's path.  Nothing is written: this prices the LEVEL (the descent is 60-70 % of the
// tree launch's span), not the rewrite of backup, expansion, re-root and export that a real build needs.
//
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I ataxxzero_amd/csrc tools/microbench/sparse_select.hip -o sparse_select
// run:   ./sparse_select /tmp/azh_trees.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "azh_device.h"

using namespace azh;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr u32 ENONE = 0xFFFFu;
constexpr u32 PRIOR_MASK = 0x7FFFFFFFu;
constexpr u32 NOBLK = 0x7FFFFFu;
constexpr int BUDGET = 48;       // bench.py's select budget
constexpr int NARROW = 8;        // lanes of a narrow request: header + 7 visited records = 128 bytes

__device__ inline u32 kid_first(u32 w) { return w & 0x7FFFFFu; }
__device__ inline int kid_count(u32 w) { return (int)((w >> 23) & 0xFFu); }
__device__ inline bool kid_finished(u32 w) { return (w >> 31) != 0u; }

__device__ inline float puct_score(float prior, float W, u32 n, float sq, float c_puct)
{
    prior = __builtin_fabsf(prior);
    const float q = n ? W / (float)n : 0.0f;
    const float u = (sq / (1.0f + (float)n)) * (c_puct * prior);
    return u + q;
}

struct Game {
    u64 base_a, base_b;    // the game's region in the two buffers (16-byte units)
    u32 root_a, root_b;    // the root's packed range (A) / block | wide << 31 (B)
    u32 root_visits, pad;
};

struct Out {
    u32 levels, hash;
    u64 ticks;             // s_memrealtime ticks (100 MHz) from the wave's start to the end of its descent
};

__device__ inline float sqrt_1p(u32 n)
{
    float r = sqrtf((float)(1u + n));
    asm volatile("" : "+v"(r));
    return r;
}

// ------------------------------------------------------------------ A: dense 16-byte records (engine.hip's fast level, read-only)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_select_dense(const uint4 *__restrict__ buf, const Game *games,
                                                                                                        int n_games, Out *out)
{
    const int g = blockIdx.x * WAVES + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (g >= n_games)
        return;
    const int lane = lane_id();
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    const Game G = games[g];
    const uint4 *ed = buf + G.base_a;
    u32 kid = G.root_a, n_node = G.root_visits, levels = 0, hash = 0;
    float sq = sqrt_1p(n_node);
    uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0, p0 = c0, p1 = c0;
    bool cur_loaded = false;
    auto load_children = [&](u32 k, uint4 &r0, uint4 &r1) {
        const int cnt = kid_count(k);
        const u32 f = kid_first(k), last = (u32)cnt - 1u;
        r0 = ed[f + min((u32)lane, last)];
        if (cnt > 64)
            r1 = ed[f + min((u32)(lane + 64), last)];
    };
    for (;;) {
        if (levels == BUDGET)
            break;
        const int M = kid_count(kid);
        if (kid_finished(kid) || M == 0)
            break;
        levels++;
        if (!cur_loaded)
            load_children(kid, c0, c1);
        cur_loaded = false;
        const bool two = M > 64;
        const bool live0 = lane < M, live1 = two && lane + 64 < M;
        int pv = -1;
        const u64 mk0 = __ballot((int)c0.x < 0);
        const u64 mk1 = two ? __ballot((int)c1.x < 0) : 0ull;
        if (mk0 | mk1) {
            pv = mk0 ? __ffsll((long long)mk0) - 1 : 64 + __ffsll((long long)mk1) - 1;
            const u32 pk = pv < 64 ? (u32)read_lane((int)c0.w, pv) : (u32)read_lane((int)c1.w, pv - 64);
            load_children(pk, p0, p1);
        }
        const u32 n0 = c0.z & 0xFFFFu, n1 = c1.z & 0xFFFFu;
        u32 bits0, bits1 = 0u;
        bool valid0, valid1 = false;
        {
            const float s = puct_score(u2f(c0.x), u2f(c0.y), n0, sq, 1.0f);
            valid0 = live0 && s >= 0.0f;
            bits0 = valid0 ? f2u(s + 0.0f) : 0u;
        }
        if (two) {
            const float s = puct_score(u2f(c1.x), u2f(c1.y), n1, sq, 1.0f);
            valid1 = live1 && s >= 0.0f;
            bits1 = valid1 ? f2u(s + 0.0f) : 0u;
        }
        const u32 top = wave_max_u32(bits0 > bits1 ? bits0 : bits1);
        const u64 cand0 = __ballot(valid0 && bits0 == top);
        const u64 cand1 = two ? __ballot(valid1 && bits1 == top) : 0ull;
        int bj = 0;
        if (cand1)
            bj = 64 + 63 - __clzll((long long)cand1);
        else if (cand0)
            bj = 63 - __clzll((long long)cand0);
        hash = hash * 31u + (u32)bj + 1u;
        u32 zsel, wsel;
        if (bj < 64) {
            zsel = (u32)read_lane((int)c0.z, bj);
            wsel = (u32)read_lane((int)c0.w, bj);
        } else {
            zsel = (u32)read_lane((int)c1.z, bj - 64);
            wsel = (u32)read_lane((int)c1.w, bj - 64);
        }
        if ((zsel >> 16) == ENONE || (zsel & 0xFFFFu) == 0u)
            break;   // an edge without a child: expand here (or the leaf the dumped select has just created)
        kid = wsel;
        n_node = (zsel & 0xFFFFu) - 1u;
        sq = sqrt_1p(n_node);
        if (pv == bj) {
            c0 = p0;
            c1 = p1;
            cur_loaded = true;
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        out[g].levels = levels;
        out[g].hash = hash;
        out[g].ticks = t1 - t0;
    }
}

// ------------------------------------------------------------------ B: header + visited records
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_select_sparse(const uint4 *__restrict__ buf, const Game *games,
                                                                                                         int n_games, Out *out)
{
    const int g = blockIdx.x * WAVES + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (g >= n_games)
        return;
    const int lane = lane_id();
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    const Game G = games[g];
    const uint4 *bl = buf + G.base_b;
    u32 kid = G.root_b, n_node = G.root_visits, levels = 0, hash = 0;
    float sq = sqrt_1p(n_node);
    uint4 c = make_uint4(0, 0, 0, 0), p = c;
    bool cur_loaded = false;
    // a node's window: header + records.  narrow: 8 lanes (the others re-read lane 7's record: same line, no request of their own)
    auto load_block = [&](u32 k, uint4 &r) {
        const u32 lim = (k >> 31) ? 63u : (u32)(NARROW - 1);
        r = bl[(k & 0x7FFFFFu) + min((u32)lane, lim)];
    };
    for (;;) {
        if (levels == BUDGET)
            break;
        if ((kid & 0x7FFFFFu) == NOBLK)
            break;   // a finished position (or a node this prototype does not model)
        levels++;
        if (!cur_loaded)
            load_block(kid, c);
        cur_loaded = false;
        const u32 hx = (u32)__builtin_amdgcn_readfirstlane((int)c.x);   // lane 0 holds the header
        const int v = (int)(hx & 0xFFu), u = (int)((hx >> 8) & 0xFFu);
        const bool rec = lane >= 1 && lane <= v;
        int pv = -1;
        const u64 mk = __ballot(rec && (int)c.x < 0);
        if (mk) {
            pv = __ffsll((long long)mk) - 1;
            const u32 pk = (u32)read_lane((int)c.w, pv);
            if ((pk & 0x7FFFFFu) != NOBLK)
                load_block(pk, p);
            else
                pv = -1;
        }
        // lane 0 scores the best unvisited child (header: y = its prior, z = move | index << 16), lanes 1..v their records
        const u32 n = rec ? (c.z & 0xFFFFu) : 0u;
        const float W = rec ? u2f(c.y) : 0.0f;
        const float P = rec ? u2f(c.x) : u2f(c.y);
        const float s = puct_score(P, W, n, sq, 1.0f);
        const bool valid = (rec || (lane == 0 && u > 0)) && s >= 0.0f;
        const u32 bits = valid ? f2u(s + 0.0f) : 0u;
        const u32 idx = rec ? ((c.w >> 23) & 0xFFu) : (c.z >> 16);
        const u32 top = wave_max_u32(bits);
        const u64 cand = __ballot(valid && bits == top);
        int bl_lane = 0;
        if (cand & (cand - 1ull)) {
            // several edges share the top score: the last in movegen order wins (cpp/self_play_client.cpp:354)
            const u32 best = wave_max_u32(((cand >> lane) & 1ull) ? idx + 1u : 0u);
            bl_lane = __ffsll((long long)__ballot(((cand >> lane) & 1ull) && idx + 1u == best)) - 1;
        } else if (cand) {
            bl_lane = __ffsll((long long)cand) - 1;
        }   // (no valid score at all — NaN evaluations — does not occur with a real net: the engine's fallback to edge 0 is not modelled)
        const u32 isel = (u32)read_lane((int)idx, bl_lane);
        hash = hash * 31u + isel + 1u;
        if (bl_lane == 0)
            break;   // the best unvisited child: expand it
        const u32 zsel = (u32)read_lane((int)c.z, bl_lane), wsel = (u32)read_lane((int)c.w, bl_lane);
        if ((zsel & 0xFFFFu) == 0u)
            break;   // (the leaf the dumped select has just created)
        kid = wsel;
        n_node = (zsel & 0xFFFFu) - 1u;
        sq = sqrt_1p(n_node);
        if (pv == bl_lane) {
            c = p;
            cur_loaded = true;
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        out[g].levels = levels;
        out[g].hash = hash;
        out[g].ticks = t1 - t0;
    }
}

// ------------------------------------------------------------------ host
struct Tree {
    int n_nodes, n_edges, root_visits;
    std::vector<u32> info, edges;   // [n_nodes][4], [n_edges][4] (device records)
};

int main(int argc, char **argv)
{
    if (argc < 2) {
        printf("usage: %s trees.bin [block alignment in 16-byte units: 1 (default) | 8 = a narrow request is ONE 128-byte line]\n", argv[0]);
        return 2;
    }
    const u32 ALIGN = argc > 2 ? (u32)atoi(argv[2]) : 1u;
    FILE *f = fopen(argv[1], "rb");
    if (!f) {
        printf("cannot open %s\n", argv[1]);
        return 2;
    }
    int G0 = 0;
    if (fread(&G0, 4, 1, f) != 1)
        return 2;
    std::vector<Tree> trees(G0);
    size_t tot_nodes = 0, tot_edges = 0;
    for (auto &t : trees) {
        int h[3];
        if (fread(h, 4, 3, f) != 3)
            return 2;
        t.n_nodes = h[0]; t.n_edges = h[1]; t.root_visits = h[2];
        t.info.resize((size_t)t.n_nodes * 4);
        t.edges.resize((size_t)t.n_edges * 4);
        if (fread(t.info.data(), 16, t.n_nodes, f) != (size_t)t.n_nodes || fread(t.edges.data(), 16, t.n_edges, f) != (size_t)t.n_edges)
            return 2;
        tot_nodes += t.n_nodes;
        tot_edges += t.n_edges;
    }
    fclose(f);
    printf("%d trees of bench.py's steady state: %zu nodes, %zu edges (%.1f per node)\n", G0, tot_nodes, tot_edges, tot_edges / (double)tot_nodes);

    // ---- build both layouts once (per game a region), statistics on the way
    std::vector<uint4> host_a, host_b;
    std::vector<Game> games0(G0);
    size_t visited_total = 0, wide_nodes = 0, live_nodes = 0, unmodelled = 0, hist[10] = {};
    for (int g = 0; g < G0; g++) {
        Tree &t = trees[g];
        Game &gm = games0[g];
        gm.base_a = host_a.size();
        gm.base_b = host_b.size();
        gm.root_visits = (u32)t.root_visits;
        // per node: v (children with a child node), the block's offset in B
        std::vector<int> v(t.n_nodes, 0), M(t.n_nodes, 0), live(t.n_nodes, 0);
        std::vector<u32> blk(t.n_nodes, NOBLK);
        u32 next = 0;
        for (int n = 0; n < t.n_nodes; n++) {
            const u32 first = t.info[4 * n], m = t.info[4 * n + 1] & 0xFFFFu, res = t.info[4 * n + 1] >> 16;
            M[n] = (int)m;
            live[n] = res == 0 && m > 0;
            for (u32 j = 0; j < m && live[n]; j++)
                v[n] += (t.edges[4 * (size_t)(first + j) + 2] >> 16) != ENONE;
            if (live[n] && (m > 128 || v[n] > 63)) {   // (the engine's general path / a fully visited 64-move node: not modelled, both kernels stop there)
                live[n] = 0;
                unmodelled++;
            }
            if (live[n]) {
                next = (next + ALIGN - 1) / ALIGN * ALIGN;
                blk[n] = next;
                next += m + 1;
                live_nodes++;
                visited_total += v[n];
                wide_nodes += v[n] > NARROW - 1;
                hist[std::min(v[n], 9)]++;
            }
        }
        // A: the records as dumped, with a node that is not modelled shown as finished to its parent
        const size_t a0 = host_a.size();
        host_a.resize(a0 + t.n_edges + 128, make_uint4(0, 0, 0, 0));   // (+128: the early request of a child shown as finished reads records 0..127)
        memcpy(&host_a[a0], t.edges.data(), (size_t)t.n_edges * 16);
        for (int e = 0; e < t.n_edges; e++) {
            const u32 child = host_a[a0 + e].z >> 16;
            if (child != ENONE && !live[child])
                host_a[a0 + e].w = 0x80000000u;
        }
        gm.root_a = live[0] ? (t.info[0] | ((u32)M[0] << 23)) : 0x80000000u;
        // B: blocks
        const size_t b0 = host_b.size();
        host_b.resize(b0 + (next + 64 + 7) / 8 * 8, make_uint4(0, 0, 0, 0));   // (+64: a wide request of the last block reads past it; regions stay 128-byte aligned)
        for (int n = 0; n < t.n_nodes; n++) {
            if (!live[n])
                continue;
            const u32 first = t.info[4 * n];
            uint4 *b = &host_b[b0 + blk[n]];
            int nv = 0;
            u32 best_p = 0, best_i = 0;
            bool have = false;
            std::vector<std::pair<u32, u32>> pairs;   // (prior bits, index) of the unvisited children
            for (int j = 0; j < M[n]; j++) {
                const u32 *e = &t.edges[4 * (size_t)(first + j)];
                const u32 child = e[2] >> 16;
                if (child != ENONE) {
                    const u32 kidw = (live[child] ? blk[child] : NOBLK) | ((u32)j << 23) | (live[child] && v[child] > NARROW - 1 ? 0x80000000u : 0u);
                    b[1 + nv++] = make_uint4(e[0], e[1], e[2], kidw);
                } else {
                    const u32 pb = e[0] & PRIOR_MASK;
                    pairs.emplace_back(pb, (u32)j);
                    if (!have || pb > best_p || (pb == best_p && (u32)j > best_i)) {
                        best_p = pb; best_i = (u32)j; have = true;
                    }
                }
            }
            b[0] = make_uint4((u32)nv | ((u32)pairs.size() << 8) | ((u32)M[n] << 16), best_p, best_i << 16, 0u);
            // the pairs, best first from the block's end backwards (never read by the descent: there for the footprint)
            std::sort(pairs.begin(), pairs.end(), [](auto &x, auto &y) { return x.first != y.first ? x.first > y.first : x.second > y.second; });
            u32 *w = reinterpret_cast<u32 *>(b + M[n] + 1);
            for (size_t k = 0; k < pairs.size(); k++) {
                w[-2 * (long)(k + 1)] = pairs[k].first;
                w[-2 * (long)(k + 1) + 1] = pairs[k].second << 16;
            }
        }
        gm.root_b = live[0] ? (blk[0] | (v[0] > NARROW - 1 ? 0x80000000u : 0u)) : NOBLK;
    }
    printf("nodes with children: %zu, visited children per node %.2f (of %.1f), nodes with more than %d visited children (wide request): %.2f %%, not modelled: %zu\n",
           live_nodes, visited_total / (double)live_nodes, tot_edges / (double)live_nodes, NARROW - 1, 100.0 * wide_nodes / live_nodes, unmodelled);
    printf("visited children per node, histogram 0..8, 9+:");
    for (int k = 0; k < 10; k++)
        printf(" %.1f%%", 100.0 * hist[k] / live_nodes);
    printf("\nfootprint per copy: A %.2f GB, B %.2f GB (blocks aligned to %u bytes)\n", host_a.size() * 16 / 1e9, host_b.size() * 16 / 1e9, ALIGN * 16);

    const int COPIES = 4;
    uint4 *d_a, *d_b;
    CK(hipMalloc((void **)&d_a, host_a.size() * 16 * COPIES));
    CK(hipMalloc((void **)&d_b, host_b.size() * 16 * COPIES));
    std::vector<Game> games((size_t)G0 * COPIES);
    for (int c = 0; c < COPIES; c++) {
        CK(hipMemcpy(d_a + host_a.size() * c, host_a.data(), host_a.size() * 16, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_b + host_b.size() * c, host_b.data(), host_b.size() * 16, hipMemcpyHostToDevice));
        for (int g = 0; g < G0; g++) {
            Game gm = games0[g];
            gm.base_a += host_a.size() * c;
            gm.base_b += host_b.size() * c;
            games[(size_t)c * G0 + g] = gm;
        }
    }
    Game *d_games;
    Out *d_out;
    CK(hipMalloc((void **)&d_games, games.size() * sizeof(Game)));
    CK(hipMemcpy(d_games, games.data(), games.size() * sizeof(Game), hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_out, games.size() * sizeof(Out)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    std::vector<Out> out_a(games.size()), out_b(games.size());
    auto launch = [&](int layout, int W) {
        const bool small = W <= 8192;   // engine.hip: four games per workgroup up to 8192 games, one beyond
        const dim3 grid(small ? (W + 3) / 4 : W), block(small ? 256 : 64);
        if (layout == 0) {
            if (small) hipLaunchKernelGGL(k_select_dense<4>, grid, block, 0, 0, (const uint4 *)d_a, (const Game *)d_games, W, d_out);
            else hipLaunchKernelGGL(k_select_dense<1>, grid, block, 0, 0, (const uint4 *)d_a, (const Game *)d_games, W, d_out);
        } else {
            if (small) hipLaunchKernelGGL(k_select_sparse<4>, grid, block, 0, 0, (const uint4 *)d_b, (const Game *)d_games, W, d_out);
            else hipLaunchKernelGGL(k_select_sparse<1>, grid, block, 0, 0, (const uint4 *)d_b, (const Game *)d_games, W, d_out);
        }
    };
    printf("\nselect-only descents (budget %d levels), launches of A and B alternating, 6 timed launches each after 2 untimed:\n", BUDGET);
    printf("%7s | %-34s | %-34s | %s\n", "games", "A dense: launch us, us/level, levels", "B sparse: launch us, us/level, levels", "B/A level");
    for (int W : {1024, 4096, 8192, 16384}) {
        if (W > G0 * COPIES)
            continue;
        double ms[2] = {0, 0}, per_level[2] = {0, 0}, lv[2] = {0, 0};
        for (int rep = 0; rep < 8; rep++) {
            for (int layout = 0; layout < 2; layout++) {
                CK(hipEventRecord(e0, 0));
                launch(layout, W);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                std::vector<Out> &o = layout ? out_b : out_a;
                CK(hipMemcpy(o.data(), d_out, (size_t)W * sizeof(Out), hipMemcpyDeviceToHost));
                if (rep >= 2) {
                    ms[layout] += t / 6.0;
                    double ticks = 0, levels = 0;
                    for (int g = 0; g < W; g++) {
                        ticks += (double)o[g].ticks;
                        levels += o[g].levels;
                    }
                    per_level[layout] += ticks / 100.0 / levels / 6.0;   // 100 MHz ticks -> us; every game's descent time over its levels
                    lv[layout] = levels / W;
                }
            }
            // the same edges in both layouts
            for (int g = 0; g < W; g++)
                if (out_a[g].levels != out_b[g].levels || out_a[g].hash != out_b[g].hash) {
                    printf("MISMATCH game %d: A levels %u hash %08x, B levels %u hash %08x\n", g, out_a[g].levels, out_a[g].hash,
                           out_b[g].levels, out_b[g].hash);
                    return 1;
                }
        }
        printf("%7d | %10.1f %10.3f %10.1f | %10.1f %10.3f %10.1f | %.3f\n", W, ms[0] * 1e3, per_level[0], lv[0], ms[1] * 1e3, per_level[1],
               lv[1], per_level[1] / per_level[0]);
    }
    printf("(every game's path hash equal in both layouts at every size: the two kernels select the same edges)\n");
    return 0;
}
