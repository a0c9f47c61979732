// Rules entry points of the C ABI: GPU perft, batched movegen / makemove /
// adjudication / feature rows, and uniform-random playouts.
//
// Reference behaviour restated (file:line under /root/reference):
//   perft                 perft.py:5-26 (a position without a move has one pass child)
//   movegen order         cpp/movegen.cpp:10-79
//   makemove              cpp/makemove.cpp:56-76
//   adjudication          cpp/self_play_client.cpp:109-144, ataxx_rules.py:159-179
//   feature rows          cpp/self_play_client.cpp:174-202
//   random play           generate_games.py:16-75 (--random-play)
#include <utility>

#include "azh_device.h"
#include "azh_host.h"

namespace azh {

// ---- per-thread (one lane = one position) rules, used by perft and playouts

__device__ inline int count_moves(const Board &b, u64 blockers)
{
    const u64 own = b.turn ? b.o : b.x;
    const u64 empty = BOARD_MASK & ~(b.x | b.o | blockers);
    int n = __popcll(single_jump_bb(own) & empty);
    u64 c = own;
    while (c) {
        const int from = __ffsll((long long)c) - 1;
        n += __popcll(double_jump_bb(1ULL << from) & empty);
        c &= c - 1;
    }
    return n;
}

// k-th move (0-based) in the reference's movegen order.
__device__ inline u32 kth_move(const Board &b, u64 blockers, int k)
{
    const u64 own = b.turn ? b.o : b.x;
    const u64 empty = BOARD_MASK & ~(b.x | b.o | blockers);
    u64 c = own;
    while (c) {
        const int from = __ffsll((long long)c) - 1;
        u64 t = double_jump_bb(1ULL << from) & empty;
        const int n = __popcll(t);
        if (k < n) {
            for (int i = 0; i < k; i++)
                t &= t - 1;
            const int to = __ffsll((long long)t) - 1;
            return (u32)(from | (to << 8));
        }
        k -= n;
        c &= c - 1;
    }
    u64 cl = single_jump_bb(own) & empty;
    for (int i = 0; i < k; i++)
        cl &= cl - 1;
    const int to = __ffsll((long long)cl) - 1;
    return (u32)(to | (to << 8));
}

__device__ inline int board_result(const Board &b, u64 blockers, int n_moves)
{
    int p1 = __popcll(b.x), p2 = __popcll(b.o);
    const int bl = __popcll(blockers);
    const int emp = 49 - p1 - p2 - bl;
    if (p1 == 0) return 2;
    if (p2 == 0) return 1;
    if (n_moves == 0) {
        if (b.turn == 0) p2 += emp;
        else p1 += emp;
    }
    if (p1 + p2 + bl == 49)
        return p1 < p2 ? 2 : 1;
    return 0;
}

// perft level: every thread owns one frontier position.  `leaf` levels only
// count; inner levels append all children to `next`.
__global__ void k_perft_level(const ulonglong2 *cur, u64 n, u64 blockers, ulonglong2 *next,
                              unsigned long long *next_count, unsigned long long *leaf_total, int leaf)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    int cnt = 0;
    Board b;
    if (i < n) {
        const ulonglong2 w = cur[i];
        b = unpack_board(w.x, w.y);
        cnt = count_moves(b, blockers);
    }
    const int children = i < n ? (cnt ? cnt : 1) : 0;
    // wave-aggregated reservation: one atomic per wave
    const int incl = wave_incl_scan(children);
    const int total = bcast_last(incl);
    if (leaf) {
        if (lane_id() == 63 && total)
            atomicAdd(leaf_total, (unsigned long long)total);
        return;
    }
    unsigned long long base = 0;
    if (lane_id() == 63 && total)
        base = atomicAdd(next_count, (unsigned long long)total);
    base = ((u64)(u32)__shfl((int)(base >> 32), 63, 64) << 32) | (u64)(u32)__shfl((int)base, 63, 64);
    if (i >= n)
        return;
    u64 pos = base + (u64)(incl - children);
    if (cnt == 0) {
        Board c = b;
        c.turn ^= 1;  // "pass" (ataxx_rules.py:112-114)
        next[pos] = make_ulonglong2(pack_word0(c), c.o);
        return;
    }
    const u64 own = b.turn ? b.o : b.x;
    const u64 empty = BOARD_MASK & ~(b.x | b.o | blockers);
    u64 c = own;
    while (c) {
        const int from = __ffsll((long long)c) - 1;
        u64 t = double_jump_bb(1ULL << from) & empty;
        while (t) {
            const int to = __ffsll((long long)t) - 1;
            const Board nb = make_move(b, from, to);
            next[pos++] = make_ulonglong2(pack_word0(nb), nb.o);
            t &= t - 1;
        }
        c &= c - 1;
    }
    u64 cl = single_jump_bb(own) & empty;
    while (cl) {
        const int to = __ffsll((long long)cl) - 1;
        const Board nb = make_move(b, to, to);
        next[pos++] = make_ulonglong2(pack_word0(nb), nb.o);
        cl &= cl - 1;
    }
}

// One wave per board: the wave-cooperative movegen the search kernels use.
__global__ __launch_bounds__(WAVE) void k_rules_batch(const ulonglong2 *boards, int n, u64 blockers, u16 *moves,
                                                      int *counts, int *results)
{
    __shared__ u16 s_moves[MAX_MOVES];
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n)
        return;
    const ulonglong2 w = boards[i];
    const Board b = unpack_board(w.x, w.y);
    int res;
    const int M = wave_movegen(b, blockers, s_moves, &res);
    __syncthreads();
    if (moves)
        for (int j = lane; j < M; j += WAVE)
            moves[(size_t)i * MAX_MOVES + j] = s_moves[j];
    if (lane == 0) {
        if (counts) counts[i] = M;
        if (results) results[i] = res;
    }
}

__global__ void k_makemove_batch(const ulonglong2 *boards, const u16 *moves, int n, ulonglong2 *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const ulonglong2 w = boards[i];
    Board b = unpack_board(w.x, w.y);
    const u32 mv = moves[i];
    if (mv == 0xFFFFu)
        b.turn ^= 1;
    else
        b = make_move(b, (int)(mv & 0xFF), (int)(mv >> 8));
    out[i] = make_ulonglong2(pack_word0(b), b.o);
}

__global__ void k_features_batch(const ulonglong2 *boards, int n, u64 blockers, float *out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * 49)
        return;
    const int row = idx / 49, c = idx % 49;
    const int x = c / 7, y = c % 7, sq = x + 7 * (6 - y);
    const ulonglong2 b = boards[row];
    float4 f;
    f.x = 1.0f;
    f.y = (float)((b.x >> sq) & 1ULL);
    f.z = (float)((b.y >> sq) & 1ULL);
    f.w = (float)((blockers >> sq) & 1ULL);
    reinterpret_cast<float4 *>(out)[idx] = f;
}

// One lane per game: uniform-random legal move each ply (generate_games.py:24),
// stop at a result (ataxx_rules.py:159-179) or at max_plies.
__global__ void k_random_play(int n_games, u32 k0, u32 k1, u64 sx, u64 so, u64 blockers, int turn, int max_plies,
                              int *plies, int *results, ulonglong2 *trace_boards, u16 *trace_moves)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_games)
        return;
    Board b;
    b.x = sx;
    b.o = so;
    b.turn = turn;
    int ply = 0, res = 0;
    for (; ply < max_plies; ply++) {
        const int n = count_moves(b, blockers);
        res = board_result(b, blockers, n);
        if (res != 0)
            break;
        const Philox4 r = philox(k0, k1, (u32)g, (u32)ply, STREAM_RANDOM_PLAY, 0u);
        const int k = (int)(((u64)r.v[0] * (u64)n) >> 32);
        const u32 mv = kth_move(b, blockers, k);
        if (trace_boards)
            trace_boards[(size_t)g * max_plies + ply] = make_ulonglong2(b.x, b.o);
        if (trace_moves)
            trace_moves[(size_t)g * max_plies + ply] = (u16)mv;
        b = make_move(b, (int)(mv & 0xFF), (int)(mv >> 8));
    }
    if (res == 0 && ply == max_plies)
        res = board_result(b, blockers, count_moves(b, blockers));
    if (plies) plies[g] = ply;
    if (results) results[g] = res;
}

// Test hook: evaluate the deterministic math on the device so tests can compare it
// bit for bit with the oracle's restatement (oracle/detmath.h).
__global__ void k_probe(int kind, int n, const float *in, const u32 *aux, u32 k0, u32 k1, u32 *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    if (kind == 0) {
        out[i] = f2u(det_expf(in[i]));
    } else if (kind == 1) {
        out[i] = f2u(det_logf(in[i]));
    } else if (kind == 2) {
        out[i] = f2u(det_gamma(in[0], k0, k1, aux[3 * i], aux[3 * i + 1], aux[3 * i + 2]));
    } else {
        const Philox4 r = philox(k0, k1, aux[4 * i], aux[4 * i + 1], aux[4 * i + 2], aux[4 * i + 3]);
        for (int j = 0; j < 4; j++)
            out[4 * i + j] = r.v[j];
    }
}

}  // namespace azh

using namespace azh;

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { AZH_HIP(hipMalloc(&p, bytes ? bytes : 1)); return 0; }
    template <typename T> T *as() { return (T *)p; }
};
}  // namespace

extern "C" int azh_perft(uint64_t x, uint64_t o, uint64_t blockers, int turn, int depth, uint64_t *nodes_out)
{
    if (!nodes_out || depth < 0 || depth > 7)
        return azh_fail(-1, "azh_perft: bad argument (depth 0..7)");
    if (azh_require_device())
        return -3;
    if (depth == 0) {
        *nodes_out = 1;
        return 0;
    }
    // Level-synchronous expansion: the frontier of depth d-1 lives in `cur`; a
    // counting pass sizes the next frontier exactly before it is generated.
    DevBuf cur, next, counters;
    size_t cap_cur = 1 << 16, cap_next = 1 << 16;
    if (cur.alloc(cap_cur * 16) || next.alloc(cap_next * 16) || counters.alloc(16))
        return -4;
    Board b;
    b.x = x;
    b.o = o;
    b.turn = turn;
    const ulonglong2 root = make_ulonglong2(pack_word0(b), b.o);
    AZH_HIP(hipMemcpy(cur.p, &root, 16, hipMemcpyHostToDevice));
    u64 n = 1;
    for (int d = 1; d <= depth; d++) {
        const unsigned grid = (unsigned)((n + 255) / 256);
        AZH_HIP(hipMemset(counters.p, 0, 16));
        hipLaunchKernelGGL(k_perft_level, dim3(grid), dim3(256), 0, 0, cur.as<ulonglong2>(), n, blockers,
                           (ulonglong2 *)nullptr, (unsigned long long *)nullptr,
                           counters.as<unsigned long long>() + 1, 1);
        AZH_HIP(hipGetLastError());
        u64 children = 0;
        AZH_HIP(hipMemcpy(&children, counters.as<unsigned long long>() + 1, 8, hipMemcpyDeviceToHost));
        if (d == depth) {
            *nodes_out = children;
            return 0;
        }
        if (children > cap_next) {
            DevBuf bigger;
            if (bigger.alloc((size_t)children * 16))
                return -4;
            std::swap(next.p, bigger.p);
            cap_next = (size_t)children;
        }
        hipLaunchKernelGGL(k_perft_level, dim3(grid), dim3(256), 0, 0, cur.as<ulonglong2>(), n, blockers,
                           next.as<ulonglong2>(), counters.as<unsigned long long>(),
                           counters.as<unsigned long long>() + 1, 0);
        AZH_HIP(hipGetLastError());
        u64 made = 0;
        AZH_HIP(hipMemcpy(&made, counters.p, 8, hipMemcpyDeviceToHost));
        if (made != children)
            return azh_fail(-7, "azh_perft: level %d generated %llu children, counted %llu", d,
                            (unsigned long long)made, (unsigned long long)children);
        n = children;
        std::swap(cur.p, next.p);
        std::swap(cap_cur, cap_next);
    }
    return 0;
}

extern "C" int azh_rules_batch(int n, const uint64_t *boards, uint64_t blockers, uint16_t *moves_out,
                               int32_t *counts_out, int32_t *results_out)
{
    if (n < 0 || !boards)
        return azh_fail(-1, "azh_rules_batch: bad argument");
    if (azh_require_device())
        return -3;
    if (n == 0)
        return 0;
    DevBuf db, dm, dc, dr;
    if (db.alloc((size_t)n * 16) || dm.alloc((size_t)n * MAX_MOVES * 2) || dc.alloc((size_t)n * 4) || dr.alloc((size_t)n * 4))
        return -4;
    AZH_HIP(hipMemcpy(db.p, boards, (size_t)n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rules_batch, dim3(n), dim3(WAVE), 0, 0, db.as<ulonglong2>(), n, blockers, dm.as<u16>(),
                       dc.as<int>(), dr.as<int>());
    AZH_HIP(hipGetLastError());
    AZH_HIP(hipDeviceSynchronize());
    if (moves_out) AZH_HIP(hipMemcpy(moves_out, dm.p, (size_t)n * MAX_MOVES * 2, hipMemcpyDeviceToHost));
    if (counts_out) AZH_HIP(hipMemcpy(counts_out, dc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (results_out) AZH_HIP(hipMemcpy(results_out, dr.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int azh_makemove_batch(int n, const uint64_t *boards, const uint16_t *moves, uint64_t *boards_out)
{
    if (n < 0 || !boards || !moves || !boards_out)
        return azh_fail(-1, "azh_makemove_batch: bad argument");
    if (azh_require_device())
        return -3;
    if (n == 0)
        return 0;
    DevBuf db, dm, dout;
    if (db.alloc((size_t)n * 16) || dm.alloc((size_t)n * 2) || dout.alloc((size_t)n * 16))
        return -4;
    AZH_HIP(hipMemcpy(db.p, boards, (size_t)n * 16, hipMemcpyHostToDevice));
    AZH_HIP(hipMemcpy(dm.p, moves, (size_t)n * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_makemove_batch, dim3((n + 255) / 256), dim3(256), 0, 0, db.as<ulonglong2>(), dm.as<u16>(), n,
                       dout.as<ulonglong2>());
    AZH_HIP(hipGetLastError());
    AZH_HIP(hipMemcpy(boards_out, dout.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int azh_features_batch(int n, const uint64_t *leaf_boards, uint64_t blockers, float *out)
{
    if (n < 0 || !leaf_boards || !out)
        return azh_fail(-1, "azh_features_batch: bad argument");
    if (azh_require_device())
        return -3;
    if (n == 0)
        return 0;
    DevBuf db, df;
    if (db.alloc((size_t)n * 16) || df.alloc((size_t)n * AZH_FEATURE_SIZE * 4))
        return -4;
    AZH_HIP(hipMemcpy(db.p, leaf_boards, (size_t)n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_features_batch, dim3((n * 49 + 255) / 256), dim3(256), 0, 0, db.as<ulonglong2>(), n, blockers,
                       df.as<float>());
    AZH_HIP(hipGetLastError());
    AZH_HIP(hipMemcpy(out, df.p, (size_t)n * AZH_FEATURE_SIZE * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int azh_random_play(int n_games, uint64_t seed, uint64_t x, uint64_t o, uint64_t blockers, int turn,
                               int max_plies, int32_t *plies, int32_t *results, uint64_t *boards_out,
                               uint16_t *moves_out)
{
    if (n_games < 0 || max_plies <= 0)
        return azh_fail(-1, "azh_random_play: bad argument");
    if (azh_require_device())
        return -3;
    if (n_games == 0)
        return 0;
    DevBuf dp, dr, db, dm;
    const size_t cells = (size_t)n_games * max_plies;
    if (dp.alloc((size_t)n_games * 4) || dr.alloc((size_t)n_games * 4))
        return -4;
    if (boards_out && db.alloc(cells * 16))
        return -4;
    if (moves_out && dm.alloc(cells * 2))
        return -4;
    hipLaunchKernelGGL(k_random_play, dim3((n_games + 63) / 64), dim3(64), 0, 0, n_games, (u32)seed, (u32)(seed >> 32), x, o,
                       blockers, turn, max_plies, dp.as<int>(), dr.as<int>(), boards_out ? db.as<ulonglong2>() : nullptr,
                       moves_out ? dm.as<u16>() : nullptr);
    AZH_HIP(hipGetLastError());
    AZH_HIP(hipDeviceSynchronize());
    if (plies) AZH_HIP(hipMemcpy(plies, dp.p, (size_t)n_games * 4, hipMemcpyDeviceToHost));
    if (results) AZH_HIP(hipMemcpy(results, dr.p, (size_t)n_games * 4, hipMemcpyDeviceToHost));
    if (boards_out) AZH_HIP(hipMemcpy(boards_out, db.p, cells * 16, hipMemcpyDeviceToHost));
    if (moves_out) AZH_HIP(hipMemcpy(moves_out, dm.p, cells * 2, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int azh_probe_detmath(int kind, int n, const float *in, const uint32_t *aux, uint64_t seed, uint32_t *out)
{
    if (kind < 0 || kind > 3 || n <= 0 || !out)
        return azh_fail(-1, "azh_probe_detmath: bad argument");
    if (azh_require_device())
        return -3;
    const size_t n_in = kind == 2 ? 1 : (size_t)n, n_aux = kind == 2 ? 3 * (size_t)n : 4 * (size_t)n;
    const size_t n_out = kind == 3 ? 4 * (size_t)n : (size_t)n;
    DevBuf din, daux, dout;
    if (din.alloc(n_in * 4) || daux.alloc(n_aux * 4) || dout.alloc(n_out * 4))
        return -4;
    if (in) AZH_HIP(hipMemcpy(din.p, in, n_in * 4, hipMemcpyHostToDevice));
    if (aux) AZH_HIP(hipMemcpy(daux.p, aux, n_aux * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe, dim3((n + 255) / 256), dim3(256), 0, 0, kind, n, din.as<float>(), daux.as<u32>(),
                       (u32)seed, (u32)(seed >> 32), dout.as<u32>());
    AZH_HIP(hipGetLastError());
    AZH_HIP(hipMemcpy(out, dout.p, n_out * 4, hipMemcpyDeviceToHost));
    return 0;
}
