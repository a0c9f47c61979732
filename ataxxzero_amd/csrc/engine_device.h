// Device code of the search engine that more than one translation unit compiles: the engine's parameter block, the HBM
// layout helpers of the tree arenas, game start, and the ply advance (sample the move, record the ply, re-root, finish or
// restart the game).  engine.hip owns the tree kernels; net_kernels.hip compiles advance_game too, because the device-resident
// loop plays the queued moves INSIDE the tower launch (advance_worker below, round 6): the tower leaves no wave slot and no
// LDS beside itself, so a re-root launch on a side stream waited most of its life for the tower's first workgroups to retire,
// and its cross-stream events cost every iteration 7-10 us whether or not a move was due (profiles/round6_reroots_in_the_tower_launch.txt).
// Nothing here does floating-point arithmetic whose result could depend on the translation unit's contraction mode.
#pragma once

#include "azh_device.h"
#include "azh_host.h"

namespace azh {

constexpr int REC_HDR_WORDS = 8;
constexpr int REC_MAXD = 256;
constexpr int REC_STRIDE_WORDS = REC_HDR_WORDS + REC_MAXD;
constexpr u32 RING_MAGIC = 0x415A4847u;  // "AZHG"
constexpr int NSTAT = AZH_STAT_COUNT;
constexpr int BFS_QL = 384;  // re-root frontier entries kept in LDS; later ones spill to bfs_spill in HBM

struct EngineParams {
    int G, visits, node_cap, edge_cap, path_cap, max_plies;
    float c_puct, alpha, noise_w;
    u32 k0, k1;
    u64 start_x, start_o, blockers;
    int start_turn;
    u32 flags;
    int select_budget;  // tree levels per select launch and game (0 = unlimited), azh_config.select_budget
    u32 uid_limit;       // azh_engine_set_game_limit: games with uid >= this are not started (0 = no limit); a slot whose
                         // next game would be one of them goes idle (phase 3: no leaf, no move, nothing to back up)
    u32 *tt;             // AZH_FLAG_EVAL_CACHE: [2][G][tt_size] open-addressed table of evaluated nodes, keyed by the board
    int tt_size;         // power of two >= 4 * node_cap
    int *no_emit;        // [G] start ply + 1 when the slot's current game was started from a loaded position
                         // (azh_engine_set_positions), else 0: such a game is played, counted and its record assembled and
                         // handed to the host like any other, but it lacks the plies before the start, so no line is written
    azh_game_state *gs;
    int *force;
    int *adv_list;   // games whose move is due (phase 2), appended by mark_game, consumed by k_advance_list
    int *adv_count;
    int *adv_done;   // tickets of k_advance_list's workgroups: the last one to finish empties the queue
    int *path;
    ulonglong2 *node_board;
    uint4 *node_info;
    uint4 *edge;
    u16 *edge_move;
    ulonglong2 *leaf_board;
    int *need_eval;
    int *leaf_list;
    int *leaf_count;
    int *leaf_list2;   // arena: leaves of the games whose mover is net B
    int *leaf_count2;
    int *tree_done;    // tickets of the workgroups of a k_tree launch (the last one compacts the leaf list): TICKET_SHARDS
                       // counters, TICKET_STRIDE ints apart, and one on top of them (a single word takes ~88 atomics per
                       // microsecond: thousands of workgroups finishing together would queue on it)
    u32 *need_mask;    // [2][mask_words] one bit per game: its leaf goes to the net (row 1: to net B, arena); set by
    int mask_words;    // k_tree's workgroups with one atomic OR each, read and cleared by the workgroup that finishes last
    u64 *stamps;       // diagnostic instantiation of k_tree only: [G][TREE_STAMPS] s_memrealtime readings (100 MHz)
    float *logits;
    float *values;
    u32 *rec;
    u32 *ring;
    u64 ring_cap_words;
    u64 *ring_head;
    u64 *stats;
    u32 *bfs_spill;  // [G][3][node_cap] frontier entries beyond BFS_QL
};

// Per-block (= per-game wave) LDS scratch shared by the tree phases.
struct TreeLds {
    u16 moves[MAX_MOVES];
    u32 old[WAVE], pref[WAVE + 1];
    union {              // the sampling weights are dead before the re-root copy starts
        u64 w[MAX_MOVES];
        u32 q[3][BFS_QL];  // frontier queue: old node id, the node's packed edge range (old arena), parent edge (new arena)
    };
};
// All G waves of a launch must be resident at once (the kernel lasts as long as its deepest descent):
// 16 games per CU at G = 4096, so the scratch has to stay under 160 KiB / 16.
static_assert(sizeof(TreeLds) <= 8192, "TreeLds: keep >= 20 game waves per CU");

struct Arena {
    ulonglong2 *nb;
    uint4 *ni;
    uint4 *ed;
    u16 *em;
};

__device__ inline Arena arena_of(const EngineParams &P, int a, int g)
{
    const size_t slot = (size_t)a * P.G + g;
    Arena A;
    A.nb = P.node_board + slot * P.node_cap;
    A.ni = P.node_info + slot * P.node_cap;
    A.ed = P.edge + slot * P.edge_cap;
    A.em = P.edge_move + slot * P.edge_cap;
    return A;
}

// The 16-byte edge record.  Everything a PUCT level needs about a child sits in it: 16 B per child instead of the 24
// of a separate (first_edge, n_edges) array — the descents of thousands of games are in flight together and their
// level loads share the memory system (tools/tree_stamps.py: a level costs 0.93 us at 1024 games, 1.37 at 4096).
//   x  prior (f32 bits; >= 0, so bit 31 is free: it MARKS the child the last descent through this node chose —
//      select_game's early request of the next level; never read by anything that decides, masked out of the documented
//      tree, PRIOR_MASK)                     y  total score W (f32 bits)
//   z  visits (bits 0-15) | child node (bits 16-31, ENONE = not expanded)
//   w  the child's first edge (bits 0-22) | its edge count (bits 23-30) | finished position (bit 31); 0 in an edge without
//      a child (rounds 3-4 kept the descent's hint in the first such edge's word: AZH_HINT_SIGN=0)
// visits <= 60000 and nodes <= visits + 8 (azh_engine_create), edges per game < 2^23, moves per position <= 255.
constexpr u32 ENONE = 0xFFFFu;
__device__ inline u32 edge_visits(const uint4 &e) { return e.z & 0xFFFFu; }
__device__ inline u32 edge_child(const uint4 &e) { return e.z >> 16; }
__device__ inline uint4 fresh_edge(u32 prior_bits) { return make_uint4(prior_bits, 0u, ENONE << 16, 0u); }
__device__ inline u32 pack_kid(u32 first, u32 n_edges, u32 finished) { return first | (n_edges << 23) | (finished << 31); }
__device__ inline u32 kid_first(u32 w) { return w & 0x7FFFFFu; }
__device__ inline int kid_count(u32 w) { return (int)((w >> 23) & 0xFFu); }
__device__ inline bool kid_finished(u32 w) { return (w >> 31) != 0u; }

// ONE_RANDOM_MOVE (:515-518): ply of the uniformly random move, uniform on 0..119, a pure function of
// (seed, game uid).
__device__ inline int random_ply_of(const EngineParams &P, u32 uid)
{
    const Philox4 r = philox(P.k0, P.k1, uid, 0u, STREAM_RANDOM_PLY, 0u);
    return (int)(((u64)r.v[0] * 120ull) >> 32);
}

__device__ inline void add_stat(const EngineParams &P, int g, int k, u64 v)
{
    if (v)
        P.stats[(size_t)g * NSTAT + k] += v;
}

// ------------------------------------------------------------------ evaluation cache (AZH_FLAG_EVAL_CACHE)
// A position the search of this game has already evaluated is not sent to the net again: engine.py's NNEvaluator.cache
// (engine.py:127-234; the C++ generator has none).  Transpositions and re-visited positions are about a quarter of all
// leaves at 400 sims/move.  Per game and arena an open-addressed table maps a board to a node that carries its
// evaluation (priors on its edges, value in node_info.w); the net is deterministic, so copying that evaluation is what
// evaluating again would give.  The root is never a source: its priors carry this ply's noise.
__host__ __device__ inline u32 tt_hash(u64 w0, u64 w1, u32 mask)
{
    const u64 k = (w0 * 0x9E3779B97F4A7C15ULL) ^ ((w1 + 0x7F4A7C15ULL) * 0xC2B2AE3D27D4EB4FULL);
    return (u32)(k >> 40) & mask;
}

__device__ inline u32 *tt_of(const EngineParams &P, int a, int g)
{
    return P.tt + ((size_t)a * P.G + g) * (size_t)P.tt_size;
}

// Node holding the evaluation of board (w0, w1), or NONE.  Wave-wide: lane l looks at slot h + l (the chain up to the
// first empty slot is searched in one round trip for the slots and one for the candidates' boards).
__device__ inline u32 tt_lookup(const u32 *tt, u32 mask, const Arena &A, u64 w0, u64 w1)
{
    const int lane = lane_id();
    const u32 id = tt[(tt_hash(w0, w1, mask) + (u32)lane) & mask];
    const u64 empties = __ballot(id == NONE);
    const u64 before = empties ? (empties & (0ULL - empties)) - 1ULL : ~0ULL;  // lanes ahead of the first empty slot
    bool match = false;
    if (((before >> lane) & 1ULL) && id != NONE) {
        const ulonglong2 b = A.nb[id];
        match = b.x == w0 && b.y == w1;
    }
    const u64 hits = __ballot(match);
    if (!hits)
        return NONE;
    return (u32)read_lane((int)id, __ffsll((long long)hits) - 1);
}

// One lane enters `id` under its board (linear probing; several lanes of the wave may insert at once).
__device__ inline void tt_insert(u32 *tt, u32 mask, u64 w0, u64 w1, u32 id)
{
    u32 slot = tt_hash(w0, w1, mask);
    for (u32 tries = 0; tries <= mask; tries++) {
        if (atomicCAS(&tt[slot], NONE, id) == NONE)
            return;
        slot = (slot + 1) & mask;
    }
}

__device__ inline void tt_clear(u32 *tt, int size)
{
    for (int i = lane_id(); i < size; i += WAVE)
        tt[i] = NONE;
}

// Fresh tree at the start position in arena 0 (generate_game :510-512,
// MCTS::init_from_scratch :380-383).  Wave-cooperative; s_moves is LDS scratch.
__device__ inline void init_game_at(const EngineParams &P, int g, u32 uid, azh_game_state &s, u16 *s_moves, Board b, int ply,
                                    int loaded)
{
    const int lane = lane_id();
    Arena A = arena_of(P, 0, g);
    int res;
    const int M = wave_movegen(b, P.blockers, s_moves, &res);
    wave_sync();
    for (int j = lane; j < M; j += WAVE) {
        A.ed[j] = fresh_edge(0u);
        A.em[j] = s_moves[j];
    }
    if (lane == 0) {
        A.nb[0] = make_ulonglong2(pack_word0(b), b.o);
        A.ni[0] = make_uint4(0u, (u32)M | ((u32)res << 16), 0u, 0u);
        P.force[g] = 0;
        P.no_emit[g] = loaded ? ply + 1 : 0;
    }
    if (P.flags & AZH_FLAG_EVAL_CACHE)
        tt_clear(tt_of(P, 0, g), P.tt_size);
    wave_sync();
    s.phase = 0;
    s.arena = 0;
    s.n_nodes = 1;
    s.n_edges = M;
    s.ply = ply;
    s.root_visits = 0;
    s.leaf_kind = AZH_LEAF_NONE;
    s.leaf_node = 0;
    s.path_len = 0;
    s.uid = uid;
}

// Fresh game at the configured start position — or, past the game limit, no game: the slot goes idle.
__device__ inline void init_game(const EngineParams &P, int g, u32 uid, azh_game_state &s, u16 *s_moves)
{
    if (P.uid_limit != 0u && uid >= P.uid_limit) {
        s.phase = 3;
        s.uid = uid;
        s.leaf_kind = AZH_LEAF_NONE;
        s.path_len = 0;
        s.root_visits = 0;
        if (lane_id() == 0)
            P.force[g] = 0;
        return;
    }
    Board b;
    b.x = P.start_x;
    b.o = P.start_o;
    b.turn = P.start_turn;
    init_game_at(P, g, uid, s, s_moves, b, 0, 0);
}

constexpr u32 PRIOR_MASK = 0x7FFFFFFFu;  // the prior proper (AZH_HINT_SIGN: bit 31 marks the remembered child)

// ------------------------------------------------------------------ ply advance

__device__ inline void advance_game(const EngineParams &P, int g, TreeLds &L)
{
    u16 *s_moves = L.moves;
    u32 *s_old = L.old, *s_pref = L.pref;
    u64 *s_w = L.w;
    u32 *spill = P.bfs_spill + (size_t)g * 3 * P.node_cap;
    const int node_cap = P.node_cap;
    // frontier queue accessors: LDS for the first BFS_QL nodes, HBM beyond
    auto q_put = [&](u32 i, u32 a, u32 b, u32 c) {
        if (i < (u32)BFS_QL) {
            L.q[0][i] = a; L.q[1][i] = b; L.q[2][i] = c;
        } else {
            spill[i] = a; spill[node_cap + i] = b; spill[2 * node_cap + i] = c;
        }
    };
    auto q_get = [&](u32 i, int f) -> u32 { return i < (u32)BFS_QL ? L.q[f][i] : spill[(size_t)f * node_cap + i]; };
    const int lane = lane_id();
    azh_game_state s = P.gs[g];
    // while (root.all_edge_visits < global_visits) step();  (:522-525)
    if (s.phase != 2)
        return;
    Arena A = arena_of(P, s.arena, g);
    Arena B = arena_of(P, 1 - s.arena, g);
    const uint4 rinfo = A.ni[0];
    const u32 first = rinfo.x;
    const int M = (int)(rinfo.y & 0xFFFFu);

    // sample_proportionally_to_visits (:495-506) on integer visit counts
    const Philox4 rr = philox(P.k0, P.k1, s.uid, (u32)s.ply, STREAM_SAMPLE, 0u);
    const u32 N = (u32)s.root_visits;
    const u32 r = (u32)(((u64)rr.v[0] * (u64)N) >> 32);
    u32 nv[4], ch[4], mvs[4];  // visits, child node, move of this lane's root edges
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = lane + 64 * k;
        uint4 ev = fresh_edge(0u);
        mvs[k] = 0;
        if (j < M) {
            ev = A.ed[first + j];
            mvs[k] = A.em[first + j];
        }
        nv[k] = edge_visits(ev);
        ch[k] = edge_child(ev);
    }
    int chosen = -1;
    if (P.flags & AZH_FLAG_SAMPLE_POW5) {
        // sample_with_exponential_weight (engine.py:532-548), exponent 5: weights (n/N)^5 over
        // edges with n >= max/2; the common 1/N^5 cancels, so the integers n^5 are exact.
        u64 mk = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            mk = (u64)nv[k] > mk ? (u64)nv[k] : mk;
        const u64 maxn = wave_max_u64(mk);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = lane + 64 * k;
            if (j < M) {
                const u64 n = nv[k];
                s_w[j] = (2 * n >= maxn) ? n * n * n * n * n : 0ull;
            }
        }
        wave_sync();
        if (lane == 0) {
            u64 T = 0;
            for (int j = 0; j < M; j++)
                T += s_w[j];
            const u64 R = ((u64)rr.v[0] << 32) | (u64)rr.v[1];
            const u64 rq = __umul64hi(R, T);
            u64 cum = 0;
            int c = -1;
            for (int j = 0; j < M; j++) {
                cum += s_w[j];
                if (c < 0 && cum > rq)
                    c = j;
            }
            s_pref[0] = (u32)c;
        }
        wave_sync();
        chosen = (int)s_pref[0];
        wave_sync();
    } else {
        u32 run = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = lane + 64 * k;
            const int incl = wave_incl_scan((int)nv[k]);
            const u32 cum = run + (u32)incl;
            const u64 mask = __ballot(j < M && cum > r);
            if (chosen < 0 && mask)
                chosen = 64 * k + (__ffsll((long long)mask) - 1);
            run += (u32)bcast_last(incl);
        }
    }
    if (chosen < 0)
        chosen = 0;
    if (P.flags & AZH_FLAG_ONE_RANDOM_MOVE) {
        const int rp = random_ply_of(P, s.uid);
        if (s.ply == rp) {
            // AT the randomization point: a uniformly random legal move (:531-540)
            chosen = (int)(((u64)rr.v[1] * (u64)(u32)M) >> 32);
        } else if (s.ply > rp) {
            // AFTER it: the most visited move (:543-551), first maximum in movegen order
            u64 key = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = lane + 64 * k;
                if (j < M) {
                    const u64 kk = ((u64)nv[k] << 32) | (u64)(0xFFFFFFFFu - (u32)j);
                    key = kk > key ? kk : key;
                }
            }
            key = wave_max_u64(key);
            chosen = (int)(0xFFFFFFFFu - (u32)key);
        }
    }

    // record the ply (:565-572): board, move, visit distribution over expanded edges
    u32 *rec = P.rec + ((size_t)g * P.max_plies + s.ply) * REC_STRIDE_WORDS;
    const u64 lt = (1ULL << lane) - 1ULL;
    int nd = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = lane + 64 * k;
        const bool has = j < M && ch[k] != ENONE;
        const u64 mask = __ballot(has);
        if (has)
            rec[REC_HDR_WORDS + nd + __popcll(mask & lt)] = mvs[k] | (nv[k] << 16);
        nd += __popcll(mask);
    }
    const int ck = chosen >> 6, cl = chosen & 63;
    const u32 my_mv = ck == 0 ? mvs[0] : (ck == 1 ? mvs[1] : (ck == 2 ? mvs[2] : mvs[3]));
    const u32 my_ch = ck == 0 ? ch[0] : (ck == 1 ? ch[1] : (ck == 2 ? ch[2] : ch[3]));
    const u32 mv = (u32)read_lane((int)my_mv, cl);
    const u32 c = (P.flags & AZH_FLAG_NO_REUSE) ? ENONE : (u32)read_lane((int)my_ch, cl);  // arena engines rebuild the tree every ply
    const ulonglong2 rootw = A.nb[0];
    if (lane == 0) {
        const u64 bx = rootw.x & ~TURN_BIT;
        rec[0] = (u32)bx;
        rec[1] = (u32)(bx >> 32);
        rec[2] = (u32)rootw.y;
        rec[3] = (u32)(rootw.y >> 32);
        rec[4] = (mv & 0xFFFFu) | ((u32)nd << 16);
        rec[5] = 0;
        rec[6] = 0;
        rec[7] = 0;
    }

    // MCTS::play (:475-492): keep the chosen child's subtree, compacted breadth-first
    // into the other arena (children keep their edge order).
    int result;
    u64 st_nodes = 0, st_edges = 0, st_spill = 0;
    if (c == ENONE) {
        // miss: fresh tree from the position after the move (:479-483)
        const Board nbrd = make_move(unpack_board(rootw.x, rootw.y), (int)(mv & 0xFF), (int)(mv >> 8));
        const int Mn = wave_movegen(nbrd, P.blockers, s_moves, &result);
        wave_sync();
        const int Mw = result != 0 ? 0 : Mn;
        for (int j = lane; j < Mw; j += WAVE) {
            B.ed[j] = fresh_edge(0u);
            B.em[j] = s_moves[j];
        }
        if (lane == 0) {
            float tv = result == 1 ? 1.0f : -1.0f;
            if (nbrd.turn == 1)
                tv = -tv;
            B.nb[0] = make_ulonglong2(pack_word0(nbrd), nbrd.o);
            B.ni[0] = result != 0 ? make_uint4(0u, (u32)result << 16, 0u, f2u(tv)) : make_uint4(0u, (u32)Mw, 0u, 0u);
        }
        s.n_nodes = 1;
        s.n_edges = Mw;
        s.root_visits = 0;
    } else {
        const uint4 cinfo = A.ni[c];
        result = (int)(cinfo.y >> 16);
        // Breadth-first copy, up to 64 frontier nodes per pass.  Nodes are numbered in
        // (parent order, edge order) and a node's edges land at the running edge count,
        // exactly as the node-at-a-time loop of the oracle does, so the compacted arena is
        // bit-identical.  The frontier (old node id, old edge range, parent edge) is queued in
        // LDS when a child is discovered — its edge range is part of the edge that leads to it —
        // so one pass costs ONE dependent memory round trip.
        u32 t = 1, eb = 0, rv = 0, qs = 0;
        if (lane == 0)
            q_put(0u, c, pack_kid(cinfo.x, cinfo.y & 0xFFFFu, (cinfo.y >> 16) != 0u), 0u);
        wave_sync();
        while (qs < t) {
            const u32 nchunk = min(t - qs, (u32)WAVE);
            u32 of = 0, Mq = 0, kw = 0, old = 0, pe = 0;
            if ((u32)lane < nchunk) {
                old = q_get(qs + lane, 0);
                kw = q_get(qs + lane, 1);
                pe = q_get(qs + lane, 2);
                of = kid_first(kw);
                Mq = (u32)kid_count(kw);
            }
            const u32 incl = (u32)wave_incl_scan((int)Mq);
            const u32 Ef = (u32)bcast_last((int)incl);
            if ((u32)lane < nchunk) {
                const u32 nf = Mq ? eb + incl - Mq : 0u;
                s_old[lane] = of;
                s_pref[lane] = incl - Mq;
                // node copy: not on the dependent chain (nothing below waits for these loads)
                B.nb[qs + lane] = A.nb[old];
                const uint4 oinfo = A.ni[old];
                B.ni[qs + lane] = make_uint4(nf, oinfo.y, 0u, oinfo.w);
                if (qs + lane > 0)  // the edge that leads here was copied in an earlier pass: now it learns the new range
                    reinterpret_cast<u32 *>(&B.ed[pe])[3] = pack_kid(nf, Mq, kid_finished(kw));
            }
            if (lane == 0)
                s_pref[nchunk] = Ef;
            wave_sync();
            for (u32 e0 = 0; e0 < Ef; e0 += WAVE) {
                const u32 e = e0 + (u32)lane;
                const bool valid = e < Ef;
                uint4 ed = fresh_edge(0u);
                u16 m = 0;
                if (valid) {
                    u32 lo = 0, hi = nchunk;  // largest i with s_pref[i] <= e
                    while (hi - lo > 1) {
                        const u32 mid = (lo + hi) >> 1;
                        if (s_pref[mid] <= e) lo = mid;
                        else hi = mid;
                    }
                    const u32 src = s_old[lo] + (e - s_pref[lo]);
                    ed = A.ed[src];
                    m = A.em[src];
                    if (qs == 0 && lo == 0)
                        rv += edge_visits(ed);
                }
                const bool has = valid && edge_child(ed) != ENONE;
                const u64 mask = __ballot(has);
                const u32 dst = eb + e;
                if (has) {
                    const u32 nc = t + (u32)__popcll(mask & lt);
                    q_put(nc, edge_child(ed), ed.w, dst);
                    ed.z = (ed.z & 0xFFFFu) | (nc << 16);  // (ed.w still names the OLD range: rewritten when the child is copied)
                }
                if (valid) {
                    B.ed[dst] = ed;
                    B.em[dst] = m;
                }
                t += (u32)__popcll(mask);
            }
            eb += Ef;
            qs += nchunk;
            wave_sync();
        }
        s.n_nodes = (int)t;
        s.n_edges = (int)eb;
        s.root_visits = (int)wave_sum_u32(rv);
        st_nodes = t;
        st_edges = eb;
        st_spill = t > (u32)BFS_QL ? 1 : 0;
    }
    if (P.flags & AZH_FLAG_EVAL_CACHE) {
        // the kept subtree's evaluations stay usable: rebuild the table of the new arena from its nodes (all but the
        // root, whose priors are about to get this ply's noise; finished positions carry no priors)
        u32 *tt = tt_of(P, 1 - s.arena, g);
        tt_clear(tt, P.tt_size);
        __threadfence();
        wave_sync();
        for (u32 n = 1u + (u32)lane; n < (u32)s.n_nodes; n += WAVE) {
            const uint4 info = B.ni[n];
            if ((info.y >> 16) == 0u && (info.y & 0xFFFFu) != 0u) {
                const ulonglong2 b = B.nb[n];
                tt_insert(tt, (u32)P.tt_size - 1u, b.x, b.y, n);
            }
        }
    }
    s.arena = 1 - s.arena;
    s.ply += 1;
    wave_sync();

    u64 st_games = 0, st_dropped = 0, st_ring = 0;
    const bool cut = result == 0 && s.ply >= P.max_plies;
    // ONE_RANDOM_MOVE: "Skipping game with no board state just after the uniformly random move" (:632-637)
    const bool no_sample = result != 0 && (P.flags & AZH_FLAG_ONE_RANDOM_MOVE) && random_ply_of(P, s.uid) + 1 >= s.ply;
    // A game that is dropped leaves an 8-word marker in the ring, so the host knows this uid will never come
    // (uid-ordered emission, azh_engine_set_emit_order).
    auto drop_marker = [&]() {
        u64 off = 0;
        if (lane == 0)
            off = atomicAdd((unsigned long long *)P.ring_head, 8ull);
        off = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((int)off);
        if (off + 8 > P.ring_cap_words) {
            st_ring = 1;
        } else if (lane == 0) {
            u32 *out = P.ring + off;
            out[0] = RING_MAGIC; out[1] = (u32)g; out[2] = s.uid; out[3] = (u32)s.ply;
            out[4] = 0; out[5] = 8; out[6] = 0; out[7] = 1;  // word 7: dropped
        }
    };
    const int loaded = P.no_emit[g];            // start ply + 1 of a game that began at a loaded position, else 0
    const int p0 = loaded ? loaded - 1 : 0;     // first ply this game recorded
    if (no_sample) {
        st_dropped = 1;
        drop_marker();
        init_game(P, g, s.uid + (u32)P.G, s, s_moves);
    } else if (result != 0 || (cut && (P.flags & AZH_FLAG_KEEP_UNFINISHED))) {
        // finished: emit the packed record (generate_game :577-578, Worker :637-642)
        const u32 *recg = P.rec + (size_t)g * P.max_plies * REC_STRIDE_WORDS;
        int words = 0;
        for (int p = p0 + lane; p < s.ply; p += WAVE)
            words += 6 + (int)(recg[(size_t)p * REC_STRIDE_WORDS + 4] >> 16);
        words = wave_sum_int(words) + 8;
        u64 off = 0;
        if (lane == 0)
            off = atomicAdd((unsigned long long *)P.ring_head, (unsigned long long)words);
        off = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((int)off);
        if (off + (u64)words <= P.ring_cap_words) {
            u32 *out = P.ring + off;
            if (lane == 0) {
                out[0] = RING_MAGIC;
                out[1] = (u32)g;
                out[2] = s.uid;
                out[3] = (u32)(s.ply - p0);   // plies in the record
                out[4] = (u32)result;
                out[5] = (u32)words;
                out[6] = (P.flags & AZH_FLAG_ONE_RANDOM_MOVE) ? (u32)random_ply_of(P, s.uid) + 1u : 0u;
                out[7] = loaded ? 2u : 0u;    // 2: partial game (begins at a loaded position): formatted by the host, not written
            }
            u32 pos = 8;
            for (int p = p0; p < s.ply; p++) {
                const u32 *rp = recg + (size_t)p * REC_STRIDE_WORDS;
                const u32 ndp = rp[4] >> 16;
                if (lane < 6)
                    out[pos + lane] = rp[lane];
                for (u32 j = lane; j < ndp; j += WAVE)
                    out[pos + 6 + j] = rp[REC_HDR_WORDS + j];
                pos += 6 + ndp;
            }
            st_games = cut ? 0 : 1;
        } else {
            st_ring = 1;
        }
        st_dropped = cut ? 1 : 0;  // arena: "invalid" -> annulled (uai_ringmaster.py:147-150)
        init_game(P, g, s.uid + (u32)P.G, s, s_moves);
    } else if (cut) {
        st_dropped = 1;  // null-result games are skipped (:628-631)
        drop_marker();
        init_game(P, g, s.uid + (u32)P.G, s, s_moves);
    } else {
        s.phase = 0;
    }
    if (lane == 0) {
        P.force[g] = 0;
        P.gs[g] = s;
    }
    {
        const u64 inc = lane == AZH_STAT_PLIES ? 1ull
                      : lane == AZH_STAT_GAMES ? (u64)st_games
                      : lane == AZH_STAT_DROPPED ? (u64)st_dropped
                      : lane == AZH_STAT_RING_OVERFLOW ? (u64)st_ring
                      : lane == AZH_STAT_REROOT_NODES ? st_nodes
                      : lane == AZH_STAT_REROOT_EDGES ? st_edges
                      : lane == AZH_STAT_REROOT_SPILLS ? st_spill : 0ull;
        if (lane < NSTAT)
            add_stat(P, g, lane, inc);
    }
}

// The queued moves of one search iteration, played by `workers` workgroups of the tower launch (its first ones: at_head) that follows the
// tree launch which queued them (mark_game): worker w takes the games adv_list[w], adv_list[w + workers], ... (one wave; the
// workgroup's other waves return at once), the last worker to finish empties the queue.  `smem`: the workgroup's LDS (the
// tower's image: not in use by a worker).  The tree launch behind the tower finds every move played — same stream, kernel
// order — exactly as it did behind the side stream's k_advance_list.
struct AdvanceHook {
    int workers;      // 0: the launch plays no moves (evaluations outside the device loop)
    int at_head;      // 1: the workers are the launch's FIRST workgroups — a launch of more workgroups than the chip has slots,
                      // whose last ones wait for earlier ones to retire; 0: its last workgroups — a launch that fits the chip at
                      // once, where every workgroup starts at once anyway and workers in front would only push tiles onto CUs
                      // that already hold one (set per launch by the tower's launch functions)
    EngineParams P;
};

__device__ inline void advance_worker(const EngineParams &P, unsigned char *smem, int worker, int workers)
{
    TreeLds &L = *reinterpret_cast<TreeLds *>(smem);
    const int n = *P.adv_count;
    for (int i = worker; i < n; i += workers) {
        advance_game(P, P.adv_list[i], L);
        wave_sync();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane_id() == 0 &&
        __hip_atomic_fetch_add(P.adv_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == workers - 1) {
        __hip_atomic_store(P.adv_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(P.adv_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace azh
