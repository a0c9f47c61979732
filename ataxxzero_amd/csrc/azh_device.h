// Device building blocks for the gfx950 tree kernels: 7x7 bitboards, wave64
// helpers, deterministic f32 math and Philox4x32-10.
//
// Rules follow the reference's C++ rules (file:line under /root/reference):
//   neighbourhood masks  cpp/bitboards.hpp:32-33, cpp/bitboards.cpp:6-39
//   makemove             cpp/makemove.cpp:56-76
//   movegen order        cpp/movegen.cpp:10-79
//   adjudication         cpp/self_play_client.cpp:109-144
// written as shift-and-mask arithmetic derived from the board geometry
// (bit = file + 7*rank), one wavefront cooperating on a position.
//
// Everything here is compiled with -ffp-contract=off: the f32 results of the
// search (priors, scores) are part of the engine's bit-exact contract with the
// CPU oracle, so every operation is a single IEEE basic op in a fixed order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace azh {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;

constexpr u64 BOARD_MASK = 0x1FFFFFFFFFFFFULL;
constexpr u64 TURN_BIT = 1ULL << 63;
constexpr u32 NONE = 0xFFFFFFFFu;
constexpr int MAX_MOVES = 256;
constexpr int WAVE = 64;

// ---------------------------------------------------------------- bitboards

// Mask of the files a stone may land on after moving df files sideways.
__host__ __device__ constexpr u64 file_mask(int f)
{
    u64 m = 0;
    for (int r = 0; r < 7; r++)
        m |= 1ULL << (f + 7 * r);
    return m;
}

__host__ __device__ constexpr u64 landing_files(int df)
{
    u64 m = 0;
    for (int f = 0; f < 7; f++)
        if (f - df >= 0 && f - df < 7)
            m |= file_mask(f);
    return m;
}

__host__ __device__ constexpr u64 shift_board(u64 bb, int s)
{
    return s >= 0 ? (bb << s) : (bb >> (-s));
}

// Union over the set squares of the ring at Chebyshev distance D.
template <int D>
__host__ __device__ constexpr u64 ring_bb(u64 bb)
{
    u64 out = 0;
    for (int dr = -D; dr <= D; dr++) {
        for (int df = -D; df <= D; df++) {
            int adf = df < 0 ? -df : df, adr = dr < 0 ? -dr : dr;
            if ((adf > adr ? adf : adr) != D)
                continue;
            out |= shift_board(bb, df + 7 * dr) & landing_files(df);
        }
    }
    return out & BOARD_MASK;
}

__host__ __device__ inline u64 single_jump_bb(u64 bb) { return ring_bb<1>(bb); }
__host__ __device__ inline u64 double_jump_bb(u64 bb) { return ring_bb<2>(bb); }

struct Board {
    u64 x;  // x stones (side to move is kept separately)
    u64 o;
    int turn;
};

__host__ __device__ inline Board unpack_board(u64 w0, u64 w1)
{
    Board b;
    b.x = w0 & ~TURN_BIT;
    b.o = w1;
    b.turn = (int)(w0 >> 63);
    return b;
}

__host__ __device__ inline u64 pack_word0(const Board &b) { return b.x | ((u64)b.turn << 63); }

// cpp/makemove.cpp:56-76
__host__ __device__ inline Board make_move(Board b, int from, int to)
{
    u64 own = b.turn ? b.o : b.x;
    u64 opp = b.turn ? b.x : b.o;
    u64 to_bb = 1ULL << to, from_bb = 1ULL << from;
    u64 captured = single_jump_bb(to_bb) & opp;
    own &= ~from_bb;
    own ^= to_bb;
    own ^= captured;
    opp ^= captured;
    Board r;
    r.x = b.turn ? opp : own;
    r.o = b.turn ? own : opp;
    r.turn = b.turn ^ 1;
    return r;
}

// cpp/self_play_client.cpp:220-237, engine.py:75,98-110: flat index into (7,7,17).
__host__ __device__ inline int policy_index(u32 move)
{
    int from = move & 0xFF, to = (move >> 8) & 0xFF;
    int fx = from % 7, fy = 6 - from / 7;
    int tx = to % 7, ty = 6 - to / 7;
    int layer;
    if (from == to) {
        layer = 16;
    } else {
        int dx = tx - fx, dy = ty - fy;
        if (dx == -2) layer = dy + 2;
        else if (dx == 2) layer = 13 + dy;
        else layer = 5 + 2 * (dx + 1) + (dy > 0 ? 1 : 0);
    }
    return 119 * tx + 17 * ty + layer;
}

// ---------------------------------------------------------------- wave64 helpers

__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }

// One wave owns a game; several such waves share a workgroup and take different paths, so nothing inside a game's
// code may wait for the other waves.  What the game's 64 lanes hand each other through LDS or global memory needs only
// this: earlier accesses of the wave are complete (workgroup-scope fence = the waitcnt a __syncthreads() would issue) and
// the compiler keeps later ones behind it.  LDS and the CU's vector L1 serve a wave's accesses in issue order.
// (-DAZH_WAVE_SYNC_WAVEFRONT=1: the same at WAVEFRONT scope — no waitcnt at all, the hand-offs no longer wait for the
// acknowledgement of the wave's global stores.  Every lock-step test passes with it and nothing gets faster — the re-root's
// passes are bound by their loads, not by its store acknowledgements: profiles/round6_wave_sync_scope_ab.txt.  Not the default.)
__device__ inline void wave_sync()
{
#ifdef AZH_WAVE_SYNC_WAVEFRONT
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}

// Cross-lane traffic goes through DPP (VALU speed), not ds_bpermute (an LDS round trip per
// hop): the PUCT descent is a dependent chain of ~15 reductions per tree level.
// gfx9 DPP controls: quad_perm 0x00-0xFF, row_shr:n 0x110+n, row_mirror 0x140,
// row_half_mirror 0x141, row_bcast15 0x142, row_bcast31 0x143.
template <int CTRL, int ROW_MASK = 0xF>
__device__ inline int dpp_i32(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xF, false);
}

__device__ inline int bcast_last(int v) { return __builtin_amdgcn_readlane(v, 63); }
// value of lane `src`, where src is the same in every lane (a scalar): v_readlane, no LDS hop
__device__ inline int read_lane(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }

// Inclusive prefix sum over the 64 lanes.
__device__ inline int wave_incl_scan(int v)
{
    v += dpp_i32<0x111>(0, v);          // row_shr:1 (lanes without a source add 0)
    v += dpp_i32<0x112>(0, v);          // row_shr:2
    v += dpp_i32<0x114>(0, v);          // row_shr:4
    v += dpp_i32<0x118>(0, v);          // row_shr:8  -> inclusive within each row of 16
    v += dpp_i32<0x142, 0xA>(0, v);     // row_bcast15 into rows 1 and 3
    v += dpp_i32<0x143, 0xC>(0, v);     // row_bcast31 into rows 2 and 3
    return v;
}

// Reductions leave the result in lane 63 and broadcast it through a scalar register.
__device__ inline int wave_sum_int(int v)
{
    v += dpp_i32<0xB1>(0, v);           // quad_perm [1,0,3,2]
    v += dpp_i32<0x4E>(0, v);           // quad_perm [2,3,0,1]
    v += dpp_i32<0x141>(0, v);          // row_half_mirror
    v += dpp_i32<0x140>(0, v);          // row_mirror
    v += dpp_i32<0x142, 0xA>(0, v);
    v += dpp_i32<0x143, 0xC>(0, v);
    return bcast_last(v);
}

__device__ inline u32 wave_sum_u32(u32 v) { return (u32)wave_sum_int((int)v); }

// f32 sum in the canonical order of the engine/oracle contract: the xor butterfly with
// offsets 1, 2, 4, 8, 16, 32 (pairs, quads, ... halves); lane 63 carries exactly that
// value (every add below pairs the two operands the butterfly pairs).
__device__ inline float wave_sum_f32(float v)
{
    v = v + __int_as_float(dpp_i32<0xB1>(0, __float_as_int(v)));
    v = v + __int_as_float(dpp_i32<0x4E>(0, __float_as_int(v)));
    v = v + __int_as_float(dpp_i32<0x141>(0, __float_as_int(v)));
    v = v + __int_as_float(dpp_i32<0x140>(0, __float_as_int(v)));
    v = v + __int_as_float(dpp_i32<0x142, 0xA>(0, __float_as_int(v)));
    v = v + __int_as_float(dpp_i32<0x143, 0xC>(0, __float_as_int(v)));
    return __int_as_float(bcast_last(__float_as_int(v)));
}

__device__ inline float wave_max_f32(float v)
{
    const auto mx = [](float a, float b) { return b > a ? b : a; };
    v = mx(v, __int_as_float(dpp_i32<0xB1>(__float_as_int(v), __float_as_int(v))));
    v = mx(v, __int_as_float(dpp_i32<0x4E>(__float_as_int(v), __float_as_int(v))));
    v = mx(v, __int_as_float(dpp_i32<0x141>(__float_as_int(v), __float_as_int(v))));
    v = mx(v, __int_as_float(dpp_i32<0x140>(__float_as_int(v), __float_as_int(v))));
    v = mx(v, __int_as_float(dpp_i32<0x142, 0xA>(__float_as_int(v), __float_as_int(v))));
    v = mx(v, __int_as_float(dpp_i32<0x143, 0xC>(__float_as_int(v), __float_as_int(v))));
    return __int_as_float(bcast_last(__float_as_int(v)));
}

__device__ inline u32 wave_max_u32(u32 v)
{
    const auto mx = [](u32 a, u32 b) { return b > a ? b : a; };
    v = mx(v, (u32)dpp_i32<0xB1>((int)v, (int)v));
    v = mx(v, (u32)dpp_i32<0x4E>((int)v, (int)v));
    v = mx(v, (u32)dpp_i32<0x141>((int)v, (int)v));
    v = mx(v, (u32)dpp_i32<0x140>((int)v, (int)v));
    v = mx(v, (u32)dpp_i32<0x142, 0xA>((int)v, (int)v));
    v = mx(v, (u32)dpp_i32<0x143, 0xC>((int)v, (int)v));
    return (u32)bcast_last((int)v);
}

// max of a 64-bit key (used for arg-max as (score bits << 32) | index)
template <int CTRL, int ROW_MASK = 0xF>
__device__ inline u64 dpp_max_u64(u64 v)
{
    const u32 lo = (u32)dpp_i32<CTRL, ROW_MASK>((int)(u32)v, (int)(u32)v);
    const u32 hi = (u32)dpp_i32<CTRL, ROW_MASK>((int)(u32)(v >> 32), (int)(u32)(v >> 32));
    const u64 o = ((u64)hi << 32) | lo;
    return o > v ? o : v;
}

__device__ inline u64 wave_max_u64(u64 v)
{
    v = dpp_max_u64<0xB1>(v);
    v = dpp_max_u64<0x4E>(v);
    v = dpp_max_u64<0x141>(v);
    v = dpp_max_u64<0x140>(v);
    v = dpp_max_u64<0x142, 0xA>(v);
    v = dpp_max_u64<0x143, 0xC>(v);
    const u32 lo = (u32)bcast_last((int)(u32)v), hi = (u32)bcast_last((int)(u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}

// ---------------------------------------------------------------- wave movegen

// Wave-cooperative move generation + adjudication for one position.
// Lane sq (< 49) owns square sq.  Moves are written to `moves` (if non-null) in
// the reference's order: jumps by ascending (from, to), then clones ascending
// (cpp/movegen.cpp:10-79).  Returns the move count (wave-uniform); *result gets
// get_board_result (cpp/self_play_client.cpp:109-144).
__device__ inline int wave_movegen(const Board &b, u64 blockers, u16 *moves, int *result)
{
    int lane = lane_id();
    u64 own = b.turn ? b.o : b.x;
    u64 opp = b.turn ? b.x : b.o;
    u64 empty = BOARD_MASK & ~(b.x | b.o | blockers);
    u64 targets = 0;
    if (lane < 49 && ((own >> lane) & 1ULL))
        targets = double_jump_bb(1ULL << lane) & empty;
    int cnt = __popcll(targets);
    int incl = wave_incl_scan(cnt);
    int jumps = bcast_last(incl);
    u64 clones = single_jump_bb(own) & empty;
    int n_clones = __popcll(clones);
    int total = jumps + n_clones;
    if (moves) {
        int pos = incl - cnt;
        while (targets) {
            int to = __ffsll((long long)targets) - 1;
            moves[pos++] = (u16)(lane | (to << 8));
            targets &= targets - 1;
        }
        if (lane < 49 && ((clones >> lane) & 1ULL)) {
            int idx = jumps + __popcll(clones & ((1ULL << lane) - 1ULL));
            moves[idx] = (u16)(lane | (lane << 8));
        }
    }
    if (result) {
        int p1 = __popcll(b.x), p2 = __popcll(b.o), bl = __popcll(blockers);
        int emp = 49 - p1 - p2 - bl;
        int res = 0;
        if (p1 == 0) res = 2;
        else if (p2 == 0) res = 1;
        else {
            if (total == 0) {
                if (b.turn == 0) p2 += emp;
                else p1 += emp;
            }
            if (p1 + p2 + bl == 49)
                res = p1 < p2 ? 2 : 1;
        }
        *result = res;
        if (p1 == 0 || p2 == 0)
            total = 0;  // the reference adjudicates before generating moves
    }
    (void)opp;
    return total;
}

// ---------------------------------------------------------------- deterministic math

__host__ __device__ inline float u2f(u32 u)
{
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f; __builtin_memcpy(&f, &u, 4); return f;
#endif
}
__host__ __device__ inline u32 f2u(float f)
{
#ifdef __HIP_DEVICE_COMPILE__
    return __float_as_uint(f);
#else
    u32 u; __builtin_memcpy(&u, &f, 4); return u;
#endif
}

// exp(x): 0 below -87, clamped above 88; ~2e-7 relative error.
__host__ __device__ inline float det_expf(float x)
{
    if (!(x >= -87.0f))
        return 0.0f;
    if (x > 88.0f)
        x = 88.0f;
    float t = x * 1.44269504f;
    float n = (t + 12582912.0f) - 12582912.0f;
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float rr = r * r;
    float y = __builtin_fmaf(p, rr, r);
    y = y + 1.0f;
    int ni = (int)n;
    return y * u2f((u32)(ni + 127) << 23);
}

// log(x) for normal x > 0; ~2e-7 relative error.
__host__ __device__ inline float det_logf(float x)
{
    if (!(x >= 1.17549435e-38f))
        return -87.33654475f;
    u32 b = f2u(x);
    int e = (int)((b >> 23) & 255u) - 126;
    float m = u2f((b & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.70710678f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    float y = (m * z) * p;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = m + y;
    return __builtin_fmaf(fe, 0.693359375f, r);
}

struct Philox4 {
    u32 v[4];
};

// Philox4x32-10 (Salmon et al., SC'11).
__host__ __device__ inline Philox4 philox(u32 k0, u32 k1, u32 c0, u32 c1, u32 c2, u32 c3)
{
    for (int round = 0; round < 10; round++) {
        u64 p0 = (u64)0xD2511F53u * c0;
        u64 p1 = (u64)0xCD9E8D57u * c2;
        u32 n0 = (u32)(p1 >> 32) ^ c1 ^ k0;
        u32 n1 = (u32)p1;
        u32 n2 = (u32)(p0 >> 32) ^ c3 ^ k1;
        u32 n3 = (u32)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 r;
    r.v[0] = c0; r.v[1] = c1; r.v[2] = c2; r.v[3] = c3;
    return r;
}

constexpr u32 STREAM_SAMPLE = 1u;
constexpr u32 STREAM_RANDOM_PLAY = 2u;
constexpr u32 STREAM_RANDOM_PLY = 3u;
constexpr u32 STREAM_GAMMA = 0x10000u;

// Gamma(alpha, 1), alpha < 1: Marsaglia-Tsang on alpha + 1 with a polar normal,
// boosted by U^(1/alpha); one Philox block per attempt.  Stands in for
// std::gamma_distribution at cpp/self_play_client.cpp:252-255.
__host__ __device__ inline float det_gamma(float alpha, u32 k0, u32 k1, u32 uid, u32 ply, u32 edge)
{
    float d = (alpha + 1.0f) - 0.333333343f;
    float c = 1.0f / sqrtf(9.0f * d);
    for (u32 attempt = 0; attempt < 64u; attempt++) {
        Philox4 r = philox(k0, k1, uid, ply, STREAM_GAMMA + edge, attempt);
        float u1 = (float)(r.v[0] >> 8) * 1.1920929e-7f - 1.0f;
        float u2 = (float)(r.v[1] >> 8) * 1.1920929e-7f - 1.0f;
        float s = u1 * u1 + u2 * u2;
        if (!(s < 1.0f) || s == 0.0f)
            continue;
        float x = u1 * sqrtf((-2.0f * det_logf(s)) / s);
        float v = 1.0f + c * x;
        if (!(v > 0.0f))
            continue;
        v = (v * v) * v;
        float U = (float)((r.v[2] >> 8) + 1u) * 5.9604645e-8f;
        float lhs = det_logf(U);
        float rhs = ((0.5f * x) * x + d) - d * v + d * det_logf(v);
        if (!(lhs < rhs))
            continue;
        float U2 = (float)((r.v[3] >> 8) + 1u) * 5.9604645e-8f;
        float boost = det_expf(det_logf(U2) / alpha);
        return (d * v) * boost;
    }
    return 0.0f;
}

}  // namespace azh
