// Batched MCTS self-play engine for gfx950: thousands of concurrent search trees
// resident in HBM as structure-of-arrays, one wavefront per game.
//
// Replaces the worker threads of the reference's cpp/self_play_client.cpp
// (file:line under /root/reference):
//   select_game   PUCT descent + expansion      MCTS::select_principal_variation :386-417,
//                                               MCTSNode::select_action :333-366,
//                                               total_action_score :310-324, step() :419-447
//   backup_game   priors, root noise, backup    Evaluations::populate :204-271, step() :449-459
//   mark_game     "is the move due?"            generate_game :522-525
//   advance_game  sample, record, re-root       sample_proportionally_to_visits :495-506,
//                                               generate_game :526-578, MCTS::play :475-492
// Each game contributes exactly one leaf per search iteration, so the per-iteration evaluation
// batch is the number of concurrent games.  The device-resident loop (run_loop) runs an
// iteration as TWO launches on one stream — the fused tower, then k_tree = backup + mark + the
// next select + the leaf-list compaction, one wave per game, four games per workgroup — with the
// queued moves (advance_game, engine_device.h) played by the first workgroups of the tower launch.  The step-wise API
// (azh_engine_select / _backup, used by the lock-step parity tests and the reference ABI) runs
// the same device functions as separate kernels: k_select -> k_compact -> k_advance_list ->
// (evaluator) -> k_backup -> k_mark.
//
// HBM layout (per game, two ping-pong arenas so re-rooting compacts by copying):
//   node_board [2][G][node_cap]  16 B  x stones | turn<<63, o stones
//   node_info  [2][G][node_cap]  16 B  first_edge, n_edges | result<<16, -, terminal value
//   edge       [2][G][edge_cap]  16 B  prior f32 | total score f32 | visits u16, child node u16 | the child's edge
//                                       range: first_edge (23 bits), n_edges (8 bits), finished (1 bit) — a derived
//                                       copy of the child's node_info, so a PUCT level is ONE 16-byte load per child
//   edge_move  [2][G][edge_cap]   2 B  from | to<<8
// A node's children are one contiguous edge range, so PUCT reads them with a
// single coalesced 16-byte-per-lane load; children are bump-allocated by the one
// wave that owns the game (no atomics inside a tree).
//
// Determinism contract: given the same config and the same evaluator outputs the
// engine reproduces oracle/mcts_oracle.c bit for bit (ties -> last maximal edge,
// Philox-keyed randomness, fixed f32 operation order; compiled -ffp-contract=off).
#include <algorithm>
#include <map>

#include <hip/hip_ext.h>

#include "azh_device.h"
#include "azh_host.h"
#include "engine_device.h"

namespace azh {

// Games (one wave each) per workgroup of the fused tree kernel.  Up to 8192 games every game wave is resident at once (8
// waves per SIMD) and the number hardly matters (tree phase 0.084-0.086 ms for 1 / 2 / 4 at 4096 games); beyond that the
// waves come in rounds, and a workgroup gives its slots back only when its slowest game is done: at 16384 games 0.183 ms
// with one game per workgroup, 0.203 with 2, 0.222 with 4, 0.233 with 8, 0.311 with 16 (tools/tree_waves_sweep.sh,
// profiles/round3_tree_waves_sweep.txt; one game per workgroup needs the sharded tickets below: with a single ticket word
// it was the slowest at 0.253).  -DAZH_TREE_WAVES=n forces one value.
#ifdef AZH_TREE_WAVES
constexpr int TREE_WAVES_SMALL = AZH_TREE_WAVES, TREE_WAVES_LARGE = AZH_TREE_WAVES;
#else
constexpr int TREE_WAVES_SMALL = 4, TREE_WAVES_LARGE = 1;
#endif
constexpr int TREE_ONE_ROUND_GAMES = 8192;  // 8 waves x 4 SIMDs x 256 CUs
constexpr int TICKET_SHARDS = 64, TICKET_STRIDE = 32;
// stamps of the diagnostic k_tree<true> (azh_engine_tree_stamps): wave start, state loaded, backup done, mark done,
// descent done (leaf edge chosen / parked / terminal), expansion done, state stored, workgroup done (all four games)
constexpr int TREE_STAMPS = 10;  // + [8] levels descended, [9] children scanned in this launch

__device__ inline u64 tree_stamp()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the phase's memory operations are part of the phase
    const u64 t = __builtin_amdgcn_s_memrealtime();  // 100 MHz, the same counter on every XCD (s_memtime is per XCD, shader clock)
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// azh_engine_set_positions: every slot restarts at a given position and ply (fresh tree, uid = slot)
__global__ __launch_bounds__(WAVE) void k_init_positions(EngineParams P, const ulonglong2 *boards, const int *plies)
{
    __shared__ u16 s_moves[MAX_MOVES];
    const int g = blockIdx.x;
    azh_game_state s;
    const ulonglong2 w = boards[g];
    init_game_at(P, g, (u32)g, s, s_moves, unpack_board(w.x, w.y), plies[g], 1);
    if (threadIdx.x == 0)
        P.gs[g] = s;
}

// azh_engine_set_game_limit: a slot whose game is past the limit and has not begun goes idle; an idle slot whose game
// is below a (raised) limit starts it
__global__ __launch_bounds__(WAVE) void k_limit_slots(EngineParams P)
{
    __shared__ u16 s_moves[MAX_MOVES];
    const int g = blockIdx.x;
    azh_game_state s = P.gs[g];
    if (s.phase == 3 && s.uid < P.uid_limit) {
        init_game(P, g, s.uid, s, s_moves);
        if (threadIdx.x == 0)
            P.gs[g] = s;
    } else if (s.phase == 0 && s.leaf_kind == AZH_LEAF_NONE && s.ply == 0 && s.n_nodes == 1 && s.root_visits == 0 &&
               s.uid >= P.uid_limit) {  // a game that has not begun (and has no root evaluation in flight)
        if (threadIdx.x == 0)
            P.gs[g].phase = 3;
    }
}

__global__ __launch_bounds__(WAVE) void k_init(EngineParams P)
{
    __shared__ u16 s_moves[MAX_MOVES];
    const int g = blockIdx.x;
    azh_game_state s;
    init_game(P, g, (u32)g, s, s_moves);
    if (threadIdx.x == 0)
        P.gs[g] = s;
    for (int k = threadIdx.x; k < NSTAT; k += WAVE)
        P.stats[(size_t)g * NSTAT + k] = 0;
}

// ------------------------------------------------------------------ select + expand

// Q + U of one edge (total_action_score :310-324) — IEEE division and square root in the fixed order the oracle restates.
// -DAZH_FAST_SCORE=1 is a MEASUREMENT build only (never shipped, not bit-exact with the oracle): the two divisions and the
// square root become single approximate instructions (v_rcp_f32 / v_sqrt_f32), i.e. the level's dependent chain without
// the ~25 instructions that precomputing q = W/n and r = cP/(1+n) at backup time would remove — an upper bound of what
// that restructuring could buy (profiles/round5_puct_chain_ab.txt).
#ifndef AZH_FAST_SCORE
#define AZH_FAST_SCORE 0
#endif
#ifndef AZH_SQRT_EARLY
#define AZH_SQRT_EARLY 1
#endif
// Where a node keeps "the child the last descent chose here" (select_game's early request): 1 = the SIGN BIT of that child's
// prior (priors are >= 0, so the bit is free in every edge: found with one ballot, and present in nodes whose children have
// all been visited — the tree's upper levels); 0 = the spare word of the node's first unvisited edge (rounds 3-4: two ballots,
// a find-first-set and a lane read to find, and nothing to find in a fully visited node).
#ifndef AZH_HINT_SIGN
#define AZH_HINT_SIGN 1
#endif
__device__ inline float puct_sqrt(float x)
{
#if AZH_FAST_SCORE
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
__device__ inline float puct_score(float prior, float W, u32 n, float sq, float c_puct)
{
    prior = __builtin_fabsf(prior);  // (the sign bit may carry the descent's mark: a source modifier, no instruction)
#if AZH_FAST_SCORE
    const float q = n ? W * __builtin_amdgcn_rcpf((float)n) : 0.0f;
    const float u = (sq * __builtin_amdgcn_rcpf(1.0f + (float)n)) * (c_puct * prior);
#else
    const float q = n ? W / (float)n : 0.0f;
    const float u = (sq / (1.0f + (float)n)) * (c_puct * prior);
#endif
    return u + q;
}

// `s` is the game's state, held in registers by the caller (the same values in all 64 lanes); written back here.
// Returns what the leaf needs: 0 nothing, 1 an evaluation (by net A), 2 an evaluation by net B (arena).
template <bool STAMP = false>
__device__ inline int select_game(const EngineParams &P, int g, azh_game_state &s, u16 *s_moves, u64 *st = nullptr)
{
    const int lane = lane_id();
    Arena A = arena_of(P, s.arena, g);
    int *path = P.path + (size_t)g * P.path_cap;

    int kind = AZH_LEAF_NONE, leaf_node = 0, depth = 0, over = 0;
    u64 st_steps = 0, st_evals = 0, st_levels = 0, st_children = 0, st_newmoves = 0, st_cached = 0, st_parked = 0;
    u64 leaf_mover = 0, leaf_opp = 0;

    if (s.phase >= 2) {
        // 2: the move of this game is due: its re-root runs after this select (the next tower launch's first workgroups), no leaf now
        // 3: the slot is idle (azh_engine_set_game_limit: every game it was to play has been played)
        kind = AZH_LEAF_NONE;
    } else if (s.phase == 0 && (P.flags & AZH_FLAG_TWO_NETS) && (A.ni[0].y & 0xFFFFu) == 1u) {
        // arena: a single legal move is played without search (uai_ringmaster.py:114-116)
        kind = AZH_LEAF_NONE;
        s.phase = 1;
        over = 2;  // sets the force flag without counting an overflow
    } else if (s.phase == 0) {
        // the root's priors are (re)computed with noise (:380-383, :485-490)
        kind = AZH_LEAF_ROOT;
        st_evals = 1;
        const ulonglong2 w = A.nb[0];
        const Board b = unpack_board(w.x, w.y);
        leaf_mover = b.turn ? b.o : b.x;
        leaf_opp = b.turn ? b.x : b.o;
    } else {
        // a descent parked by the level budget resumes at the node it stopped at (same tree: nothing of this
        // game was touched in between)
        const bool resume = s.leaf_kind == AZH_LEAF_DESCENT;
        st_steps = resume ? 0 : 1;
        u32 node = resume ? (u32)s.leaf_node : 0u;
        depth = resume ? s.path_len : 0;
        const uint4 rinfo = A.ni[node];
        // edge range of the node being scanned, in the packed form the edges carry for their children
        u32 kid = pack_kid(rinfo.x, rinfo.y & 0xFFFFu, (rinfo.y >> 16) != 0u);
        int levels_done = 0;
        // N of the node being scanned (the sum of its children's visits) without a wave reduction on the level's critical
        // path: for the root it is the game's root_visits; below, every visit of the edge into a node but the first —
        // the one that created the node — went on into one of its children, so N = that edge's visits - 1 (the
        // bookkeeping identity tests/test_gpu_engine.py checks at full size).  A resumed descent does not have the edge it
        // came through at hand and sums once.
        bool have_n = !resume;
        u32 n_node = (u32)s.root_visits;
        // sqrt(1 + N) of the node about to be scanned, computed as soon as N is known — at the END of the level above, while
        // the node's records are still on their way — instead of after their arrival: the IEEE square root (a dozen
        // instructions) leaves the chain between the arrival of a level's records and the request for the next level's.
        // Same function of the same argument: nothing the oracle could see.  (AZH_SQRT_EARLY=0: where it used to be.)
        auto sqrt_1p = [](u32 n) {
            float r = puct_sqrt((float)(1u + n));
            asm volatile("" : "+v"(r));  // computed HERE (the optimiser would otherwise sink it to its use behind the wait)
            return r;
        };
        float sq_node = sqrt_1p(n_node);
        // The path of the descent: lane k keeps the k-th entry added in this launch and the entries are stored together
        // afterwards (every 64 levels when there is no budget): a store per level would have to be acknowledged before
        // the next level's records count as arrived (vmcnt counts stores in order with the loads).
        u32 path_buf = 0;
        int path_base = depth;
        auto push_path = [&](u32 eidx) {
            if (lane == depth - path_base)
                path_buf = eidx;
            depth++;
            if (depth - path_base == WAVE) {
                path[path_base + lane] = (int)path_buf;
                path_base = depth;
            }
        };
        // Early request of the next level (never changes what is selected): a node remembers which child the last descent
        // through it chose — the sign bit of that child's prior (priors are >= 0; one ballot finds it, and every node has
        // room for it, the fully visited nodes of the upper levels included).  When a node's records arrive, the
        // remembered child's own records — its range is in this node's records — are requested at once and the scores are
        // computed while they are in flight; if the scores pick that child, the next level starts with its records
        // already on the way: a level then costs max(memory latency, its instructions) instead of their sum.
        // (two records per lane: nodes of up to 128 moves take this path — a third of the nodes of a mid-game position have
        // more than 64)
        // Two sets of record registers take turns without a register-to-register copy: a level that scans set a requests
        // the remembered child's records into set b; if the scores pick that child, the next level — a second copy of the
        // level's code — scans set b and requests into set a.  Only set a is ever live across the loop's back edge.
        uint4 ea0 = fresh_edge(0u), ea1 = fresh_edge(0u), eb0 = fresh_edge(0u), eb1 = fresh_edge(0u);
        bool cur_loaded = false;  // the records of the node about to be scanned have been requested (set a at the loop's head)
        // Every lane loads: lanes past the node's last edge read that last edge again (same cache line, no traffic) — no
        // lane mask around the load, so no register has to be cleared or merged per request; every use below is masked by
        // live0 / live1 or reads a lane the node's edge count has been checked against.  (cnt >= 1 at both call sites.)
        auto load_children = [&](u32 k, uint4 &r0, uint4 &r1) {
            const int cnt = kid_count(k);
            const u32 f = kid_first(k), last = (u32)cnt - 1u;
            r0 = A.ed[f + min((u32)lane, last)];
            if (cnt > WAVE)
                r1 = A.ed[f + min((u32)(lane + WAVE), last)];
        };
        u32 sel_eidx = 0;
        // One level of the fast path (all but a handful of nodes): one edge per lane, two beyond 64 moves.  Same arithmetic
        // as the general path below; the arg-max is a 32-bit max of the score bits plus ballots for the tie rule, instead of
        // a 64-bit (score, index) key reduction — this loop is a latency chain, instructions count.  Scans (c0, c1),
        // requests into (p0, p1).  true: descended into a child; false: the chosen edge has no child (sel_eidx: expand it).
        auto fast_level = [&](uint4 &c0, uint4 &c1, uint4 &p0, uint4 &p1, int M, u32 first) -> bool {
            if (!cur_loaded)
                load_children(kid, c0, c1);
            cur_loaded = false;
            const uint4 &e0 = c0, &e1 = c1;
            const bool two = M > WAVE;
            const bool live0 = lane < M, live1 = two && lane + WAVE < M;
            // the remembered child, requested before anything is scored
            int pv = -1, pred = -1;
#if !AZH_HINT_SIGN
            int u0 = -1;
#endif
#if !AZH_HINT_SIGN
            auto request_child = [&](int idx) {
                u32 pz, pk;
                if (idx < WAVE) {
                    pz = (u32)read_lane((int)e0.z, idx);
                    pk = (u32)read_lane((int)e0.w, idx);
                } else {
                    pz = (u32)read_lane((int)e1.z, idx - WAVE);
                    pk = (u32)read_lane((int)e1.w, idx - WAVE);
                }
                if ((pz >> 16) != ENONE && !kid_finished(pk) && kid_count(pk) > 0 && kid_count(pk) <= 2 * WAVE) {
                    load_children(pk, p0, p1);
                    pred = idx;
                }
            };
#endif
#if AZH_HINT_SIGN
            {
                // (a mark is only ever set on an edge whose child exists, is not a finished position and has 1 .. 128 moves
                // — below — and none of that changes while the edge lives: nothing to check here, one lane read)
                // (no live mask: a lane past the node's last edge holds a copy of that last edge, so the FIRST set bit is
                // always the marked edge itself — and a bare compare writes the lane mask without a detour through a register)
                const u64 mk0 = __ballot((int)e0.x < 0);
                const u64 mk1 = two ? __ballot((int)e1.x < 0) : 0ull;
                if (mk0 | mk1) {
                    pv = mk0 ? __ffsll((long long)mk0) - 1 : WAVE + __ffsll((long long)mk1) - 1;
                    const u32 pk = pv < WAVE ? (u32)read_lane((int)e0.w, pv) : (u32)read_lane((int)e1.w, pv - WAVE);
                    load_children(pk, p0, p1);
                    pred = pv;
                }
            }
#else
            const u64 unv0 = __ballot(live0 && edge_child(e0) == ENONE);
            const u64 unv1 = two ? __ballot(live1 && edge_child(e1) == ENONE) : 0ull;
            if (unv0 | unv1) {
                u32 hw;
                if (unv0) {
                    u0 = __ffsll((long long)unv0) - 1;
                    hw = (u32)read_lane((int)e0.w, u0);
                } else {
                    u0 = __ffsll((long long)unv1) - 1;
                    hw = (u32)read_lane((int)e1.w, u0);
                    u0 += WAVE;
                }
                if ((hw >> 31) && (int)(hw & 0xFFu) < M) {
                    pv = (int)(hw & 0xFFu);
                    request_child(pv);
                }
            }
#endif
            const u32 n0 = edge_visits(e0), n1 = edge_visits(e1);
            float sq1;
            if (AZH_SQRT_EARLY && have_n)
                sq1 = sq_node;
            else
                sq1 = puct_sqrt((float)(1u + (have_n ? n_node : wave_sum_u32((live0 ? n0 : 0u) + (live1 ? n1 : 0u)))));
            u32 bits0, bits1 = 0u;
            bool valid0, valid1 = false;
            {
                const float score = puct_score(u2f(e0.x), u2f(e0.y), n0, sq1, P.c_puct);
                valid0 = live0 && score >= 0.0f;  // NaN scores are never selected by either reference
                bits0 = valid0 ? f2u(score + 0.0f) : 0u;
            }
            if (two) {
                const float score = puct_score(u2f(e1.x), u2f(e1.y), n1, sq1, P.c_puct);
                valid1 = live1 && score >= 0.0f;
                bits1 = valid1 ? f2u(score + 0.0f) : 0u;
            }
            const u32 top = wave_max_u32(bits0 > bits1 ? bits0 : bits1);
            const u64 cand0 = __ballot(valid0 && bits0 == top);
            const u64 cand1 = two ? __ballot(valid1 && bits1 == top) : 0ull;
            int bj = 0;
            if (P.flags & AZH_FLAG_TIE_FIRST) {
                if (cand0)
                    bj = __ffsll((long long)cand0) - 1;
                else if (cand1)
                    bj = WAVE + __ffsll((long long)cand1) - 1;
            } else {
                if (cand1)
                    bj = WAVE + 63 - __clzll((long long)cand1);
                else if (cand0)
                    bj = 63 - __clzll((long long)cand0);
            }
            const u32 eidx = first + (u32)bj;
            push_path(eidx);
            // the chosen edge: visits | child << 16, and the child's range
            u32 zsel, wsel;
            if (bj < WAVE) {
                zsel = (u32)read_lane((int)e0.z, bj);
                wsel = (u32)read_lane((int)e0.w, bj);
            } else {
                zsel = (u32)read_lane((int)e1.z, bj - WAVE);
                wsel = (u32)read_lane((int)e1.w, bj - WAVE);
            }
            const u32 child = zsel >> 16;
            if (child == ENONE) {
                sel_eidx = eidx;
                return false;
            }
            // remember the choice (stored only when it changes)
#if AZH_HINT_SIGN
            if (bj != pv) {
                // the mark moves: the lane that holds the edge chosen last time clears its prior's sign bit, the lane that
                // holds the newly chosen edge sets it.  Two predicated stores: bj and pv may be 64 apart, i.e. the two
                // edges of ONE lane — as one store with a per-lane choice of the edge that lane dropped its clear, and the
                // node kept two marks for good (found by the round-5 review; tests/test_gpu_engine.py reads the raw marks).
                // The mark moves in about one level of a hundred: the second store is not on the level's usual path.
                const bool markable = !kid_finished(wsel) && kid_count(wsel) > 0 && kid_count(wsel) <= 2 * WAVE;
                if (pv >= 0 && lane == (pv & 63))
                    reinterpret_cast<u32 *>(&A.ed[first + (u32)pv])[0] = (pv >= WAVE ? e1.x : e0.x) & PRIOR_MASK;
                if (markable && lane == (bj & 63))
                    reinterpret_cast<u32 *>(&A.ed[first + (u32)bj])[0] = (bj >= WAVE ? e1.x : e0.x) | ~PRIOR_MASK;
            }
#else
            if (u0 >= 0 && bj != pv && lane == 0)
                reinterpret_cast<u32 *>(&A.ed[first + (u32)u0])[3] = 0x80000000u | (u32)bj;
#endif
            node = child;
            kid = wsel;
            n_node = (zsel & 0xFFFFu) - 1u;
            have_n = true;
            if (AZH_SQRT_EARLY)
                sq_node = sqrt_1p(n_node);  // (while the requested records are on their way)
            if (pred == bj)
                cur_loaded = true;  // the records requested before the scores were computed are the next level's
            return true;
        };
        // what every level starts with: the level budget, finished positions, the counters.  true: the descent ends here.
        int M = 0;
        u32 first = 0;
        auto level_begin = [&]() -> bool {
            if (P.select_budget != 0 && levels_done == P.select_budget) {
                kind = AZH_LEAF_DESCENT;  // park: no leaf for the evaluator from this game this iteration
                leaf_node = (int)node;
                st_parked = 1;
                return true;
            }
            levels_done++;
            M = kid_count(kid);
            first = kid_first(kid);
            if (kid_finished(kid) || M == 0) {
                kind = AZH_LEAF_TERMINAL;  // select_action -> NO_MOVE (:336-340)
                leaf_node = (int)node;
                return true;
            }
            st_levels += 1;
            st_children += (u64)M;
            return false;
        };
        bool expand = false;
        for (;;) {
            if (level_begin())
                break;
            if (M <= 2 * WAVE) {
                if (!fast_level(ea0, ea1, eb0, eb1, M, first)) {
                    expand = true;
                    break;
                }
                if (!cur_loaded)
                    continue;  // (the next level requests its records itself, into set a)
                // the remembered child was chosen: its records are on their way into set b (it has 1 .. 128 moves)
                if (level_begin())
                    break;
                if (!fast_level(eb0, eb1, ea0, ea1, M, first)) {
                    expand = true;
                    break;
                }
                continue;  // (cur_loaded: set a holds the next level's records)
            } else {
            const int rounds = (M + 63) >> 6;
            uint4 *const evp[4] = {&ea0, &ea1, &eb0, &eb1};  // the fast path's record registers (whatever they held is dead:
#define ev(k) (*evp[k])                                        // cur_loaded is cleared below)
            u32 nsum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = lane + 64 * k;
                if (k < rounds && j < M) {
                    ev(k) = A.ed[first + j];
                    nsum += edge_visits(ev(k));
                }
            }
            const u32 ntot = have_n ? n_node : wave_sum_u32(nsum);
            const float sq = (AZH_SQRT_EARLY && have_n) ? sq_node : puct_sqrt((float)(1u + ntot));
            // arg-max with ties to the LAST maximal edge (:354) — or the FIRST, python's max()
            // (engine.py:291), in the arena: scores are >= 0, so their bit patterns order like
            // the floats and (bits << 32 | index or ~index) is a total order; NaN scores (never
            // selected by either reference) and empty lanes map to key 0.
            const u32 tie_flip = (P.flags & AZH_FLAG_TIE_FIRST) ? 0xFFFFFFFFu : 0u;
            // every lane keeps the child data of its own best edge: the winning lane's best IS the winner, so
            // no array is indexed by the (run-time) round of the winner (that put the arrays in scratch memory
            // and a scratch round trip into every level)
            u64 key = 0;
            u32 mine = ev(0).z, mkid = ev(0).w;  // (lane 0 always holds edge 0: the key-0 default below)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = lane + 64 * k;
                if (k < rounds && j < M) {
                    const float score = puct_score(u2f(ev(k).x), u2f(ev(k).y), edge_visits(ev(k)), sq, P.c_puct);
                    const u64 kj = score >= 0.0f ? (((u64)f2u(score + 0.0f)) << 32) | (u64)((u32)j ^ tie_flip) : 0ull;
                    if (kj > key) {
                        key = kj;
                        mine = ev(k).z;
                        mkid = ev(k).w;
                    }
                }
            }
#undef ev
            key = wave_max_u64(key);
            const int bj = key ? (int)((u32)key ^ tie_flip) : 0;  // key 0: edge 0 = lane 0's round-0 default
            const u32 eidx = first + (u32)bj;
            push_path(eidx);
            const u32 zsel = (u32)read_lane((int)mine, bj & 63);
            const u32 child = zsel >> 16;
            if (child != ENONE) {
                node = child;
                kid = (u32)read_lane((int)mkid, bj & 63);
                n_node = (zsel & 0xFFFFu) - 1u;
                have_n = true;
                if (AZH_SQRT_EARLY)
                    sq_node = sqrt_1p(n_node);
                cur_loaded = false;
                continue;
            }
            sel_eidx = eidx;
            expand = true;
            break;
            }
        }
        if (expand) {
            // expand (:429-439)
            if constexpr (STAMP) st[4] = tree_stamp();
            const u32 eidx = sel_eidx;
            const u32 mv = A.em[eidx];
            const ulonglong2 pw = A.nb[node];
            const Board cb = make_move(unpack_board(pw.x, pw.y), (int)(mv & 0xFF), (int)(mv >> 8));
            int res2;
            const int M2 = wave_movegen(cb, P.blockers, s_moves, &res2);
            wave_sync();
            if (s.n_nodes >= P.node_cap || (res2 == 0 && (s.n_edges + M2 > P.edge_cap || M2 > 255))) {
                over = 1;
                kind = AZH_LEAF_NONE;
                leaf_node = 0;
                depth = 0;
            } else {
            const u32 cid = (u32)s.n_nodes;
            s.n_nodes += 1;
            u32 known = NONE;  // a node of this tree that already carries the evaluation of this position
            if ((P.flags & AZH_FLAG_EVAL_CACHE) && res2 == 0)
                known = tt_lookup(tt_of(P, s.arena, g), (u32)P.tt_size - 1u, A, pack_word0(cb), cb.o);
            if (res2 != 0) {
                float tv = res2 == 1 ? 1.0f : -1.0f;
                if (cb.turn == 1)
                    tv = -tv;
                if (lane == 0)
                    A.ni[cid] = make_uint4(0u, (u32)res2 << 16, 0u, f2u(tv));
                kind = AZH_LEAF_TERMINAL;
            } else if (known != NONE) {
                // same position, same moves in the same order: take the priors and the value, no evaluation
                const uint4 kinfo = A.ni[known];
                const u32 nf = (u32)s.n_edges;
                for (int j = lane; j < M2; j += WAVE) {
                    A.ed[nf + j] = fresh_edge(A.ed[kinfo.x + j].x & PRIOR_MASK);  // (without the other node's mark)
                    A.em[nf + j] = s_moves[j];
                }
                s.n_edges += M2;
                if (lane == 0)
                    A.ni[cid] = make_uint4(nf, (u32)M2, 0u, kinfo.w);
                kind = AZH_LEAF_TERMINAL;  // "value known": backed up from node_info.w like a finished position
                st_cached = 1;
                st_newmoves = (u64)M2;
            } else {
                const u32 nf = (u32)s.n_edges;
                for (int j = lane; j < M2; j += WAVE) {
                    A.ed[nf + j] = fresh_edge(0u);
                    A.em[nf + j] = s_moves[j];
                }
                s.n_edges += M2;
                if (lane == 0)
                    A.ni[cid] = make_uint4(nf, (u32)M2, 0u, 0u);
                kind = AZH_LEAF_EVAL;
                st_evals = 1;
                st_newmoves = (u64)M2;
            }
            if (lane == 0) {
                A.nb[cid] = make_ulonglong2(pack_word0(cb), cb.o);
                // the edge gets its child (an edge without a child has no visits: the first one creates it) and the child's range
                reinterpret_cast<uint2 *>(&A.ed[eidx])[1] =
                    make_uint2(cid << 16, res2 != 0 ? pack_kid(0u, 0u, 1u) : pack_kid((u32)(s.n_edges - M2), (u32)M2, 0u));
            }
            leaf_node = (int)cid;
            leaf_mover = cb.turn ? cb.o : cb.x;
            leaf_opp = cb.turn ? cb.x : cb.o;
            }
        }
        // the path entries of this launch, one store for all of them (none after an overflow: depth is 0 then)
        if (lane < depth - path_base)
            path[path_base + lane] = (int)path_buf;
    }

    s.leaf_kind = kind;
    s.leaf_node = leaf_node;
    s.path_len = depth;
    if constexpr (STAMP) {
        st[5] = tree_stamp();
        if (st[4] == 0)
            st[4] = st[5];  // no expansion: the descent ended at a finished position, parked, or there was none
        st[8] = st_levels;
        st[9] = st_children;
    }
    int need = (kind == AZH_LEAF_EVAL || kind == AZH_LEAF_ROOT) ? 1 : 0;
    if (need && (P.flags & AZH_FLAG_TWO_NETS))
        need = 1 + ((s.ply + g) & 1);  // net A (1) / net B (2) is to move; even slots give x to A
    if (lane == 0) {
        P.gs[g] = s;
        P.need_eval[g] = need;
        P.leaf_board[g] = make_ulonglong2(leaf_mover, leaf_opp);
        if (over)
            P.force[g] = 1;
    }
    {   // counters: lane k owns counter k, one read-modify-write round trip for all of them
        const u64 inc = lane == AZH_STAT_STEPS ? st_steps
                      : lane == AZH_STAT_NN_EVALS ? st_evals
                      : lane == AZH_STAT_LEVELS ? st_levels
                      : lane == AZH_STAT_CHILDREN ? st_children
                      : lane == AZH_STAT_NEW_MOVES ? st_newmoves
                      : lane == AZH_STAT_EDGE_OVERFLOW ? (u64)(over == 1)
                      : lane == AZH_STAT_CACHE_HITS ? st_cached
                      : lane == AZH_STAT_PARKED ? st_parked : 0ull;
        if (lane < NSTAT)
            add_stat(P, g, lane, inc);
    }
    return need;
}

// Dense, game-ordered list of the games whose leaf needs the evaluator.
__global__ __launch_bounds__(1024) void k_compact(const int *need, int G, int *list, int *count, int cls, int any)
{
    __shared__ int s_sum[1024];
    const int t = threadIdx.x;
    const int c = (G + 1023) / 1024;
    const int lo = t * c, hi = min(G, lo + c);
    int cnt = 0;
    for (int i = lo; i < hi; i++)
        cnt += any ? need[i] != 0 : need[i] == cls;
    s_sum[t] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? s_sum[t - off] : 0;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    int base = s_sum[t] - cnt;
    for (int i = lo; i < hi; i++)
        if (any ? need[i] != 0 : need[i] == cls)
            list[base++] = i;
    if (t == 1023)
        *count = s_sum[1023];
}

// ------------------------------------------------------------------ priors + backup

// `s`: the game's state in the caller's registers (uniform over the lanes); updated, not stored.
__device__ inline void backup_game(const EngineParams &P, int g, azh_game_state &s)
{
    const int lane = lane_id();
    const int kind = s.leaf_kind;
    if (kind == AZH_LEAF_NONE || kind == AZH_LEAF_DESCENT)
        return;  // nothing evaluated; a parked descent keeps its state
    Arena A = arena_of(P, s.arena, g);

    if (kind == AZH_LEAF_EVAL || kind == AZH_LEAF_ROOT) {
        // Evaluations::populate (:204-271): softmax over the legal moves' logits —
        // identical to the reference's 833-way softmax renormalised over the legal
        // moves — then the Dirichlet mix at the root.
        const uint4 info = A.ni[s.leaf_node];
        const u32 first = info.x;
        const int M = (int)(info.y & 0xFFFFu);
        const int rounds = (M + 63) >> 6;
        const float *row = P.logits + (size_t)g * AZH_POLICY_SIZE;
        float l[4], ex[4];
        float pr[4];
        if (P.flags & AZH_FLAG_PY_POSTERIOR) {
            // engine.py:197-203: softmax over all 833 logits, gather the legal moves, divide by
            // (their sum + 1e-6)
            float la[14];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 14; t++) {
                const int i = lane + 64 * t;
                la[t] = i < AZH_POLICY_SIZE ? row[i] : -INFINITY;
                if (la[t] > mx)
                    mx = la[t];
            }
            mx = wave_max_f32(mx);
            float part = 0.0f;
#pragma unroll
            for (int t = 0; t < 14; t++)
                if (lane + 64 * t < AZH_POLICY_SIZE)
                    part = part + det_expf(la[t] - mx);
            const float S = wave_sum_f32(part);
            float lpart = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = lane + 64 * k;
                ex[k] = 0.0f;
                if (k < rounds && j < M) {
                    ex[k] = det_expf(row[policy_index(A.em[first + j])] - mx) / S;
                    lpart = lpart + ex[k];
                }
            }
            const float den = wave_sum_f32(lpart) + 1e-6f;
#pragma unroll
            for (int k = 0; k < 4; k++)
                pr[k] = ex[k] / den;
        } else {
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = lane + 64 * k;
                l[k] = -INFINITY;
                if (k < rounds && j < M) {
                    l[k] = row[policy_index(A.em[first + j])];
                    if (l[k] > mx)
                        mx = l[k];
                }
            }
            mx = wave_max_f32(mx);
            float part = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = lane + 64 * k;
                ex[k] = 0.0f;
                if (k < rounds && j < M) {
                    ex[k] = det_expf(l[k] - mx);
                    part = part + ex[k];
                }
            }
            const float S = wave_sum_f32(part);
#pragma unroll
            for (int k = 0; k < 4; k++)
                pr[k] = S > 0.0f ? ex[k] / S : ex[k];
        }
        if (kind == AZH_LEAF_ROOT && P.noise_w > 0.0f) {
            float gm[4];
            float gpart = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = lane + 64 * k;
                gm[k] = 0.0f;
                if (k < rounds && j < M) {
                    gm[k] = det_gamma(P.alpha, P.k0, P.k1, s.uid, (u32)s.ply, (u32)j);
                    gpart = gpart + gm[k];
                }
            }
            const float T = wave_sum_f32(gpart);
            const float w = P.noise_w, omw = 1.0f - w;
            if (T > 0.0f) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float d = gm[k] / T;
                    const float t1 = w * d;
                    const float t2 = omw * pr[k];
                    pr[k] = t1 + t2;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = lane + 64 * k;
            if (k < rounds && j < M)  // (stored without a sign: bit 31 is the descent's mark.  Priors are >= 0 and never NaN —
                                      // det_expf is 0 for a NaN argument — so the mask changes nothing: belt and braces)
                reinterpret_cast<u32 *>(&A.ed[first + j])[0] = f2u(pr[k]) & PRIOR_MASK;
        }
    }

    if ((P.flags & AZH_FLAG_EVAL_CACHE) && kind == AZH_LEAF_EVAL) {
        // the leaf now carries an evaluation: remember its value and enter it in the table
        if (lane == 0) {
            reinterpret_cast<u32 *>(&A.ni[s.leaf_node])[3] = f2u(P.values[g]);
            const ulonglong2 b = A.nb[s.leaf_node];
            tt_insert(tt_of(P, s.arena, g), (u32)P.tt_size - 1u, b.x, b.y, (u32)s.leaf_node);
        }
    }
    if (kind == AZH_LEAF_EVAL || kind == AZH_LEAF_TERMINAL) {
        // step() part 4 (:449-459): flip the score at every edge on the way up.
        const float v = kind == AZH_LEAF_EVAL ? P.values[g] : u2f(A.ni[s.leaf_node].w);
        const float sc0 = (v + 1.0f) * 0.5f;
        const float fa = 1.0f - sc0, fb = 1.0f - fa, fc = 1.0f - fb;
        const int *path = P.path + (size_t)g * P.path_cap;
        for (int i = lane; i < s.path_len; i += WAVE) {
            const int flips = s.path_len - i;
            const float val = flips == 1 ? fa : ((flips & 1) ? fc : fb);
            u32 *e = reinterpret_cast<u32 *>(&A.ed[path[i]]);
            e[1] = f2u(u2f(e[1]) + val);
            e[2] += 1u;  // visits: the low half of the word (<= 60000, never carries into the child id)
        }
        if (s.path_len > 0)
            s.root_visits += 1;
    }
    if (kind == AZH_LEAF_ROOT)
        s.phase = 1;
    s.leaf_kind = AZH_LEAF_NONE;
}

// The tree phases as kernels (step-wise API) and fused (device-resident loop): one wave owns a game through
// backup -> mark -> select, so a step costs one launch instead of three.
__global__ __launch_bounds__(WAVE) void k_select(EngineParams P)
{
    __shared__ u16 s_moves[MAX_MOVES];  // the move list of the node being expanded: all the scratch a descent needs
    azh_game_state s = P.gs[blockIdx.x];
    select_game(P, blockIdx.x, s, s_moves);
}

__global__ __launch_bounds__(WAVE) void k_backup(EngineParams P)
{
    const int g = blockIdx.x;
    azh_game_state s = P.gs[g];
    const int kind = s.leaf_kind;
    backup_game(P, g, s);
    if (kind != AZH_LEAF_NONE && kind != AZH_LEAF_DESCENT && threadIdx.x == 0)
        P.gs[g] = s;
}

// while (root.all_edge_visits < global_visits) step();  (:522-525): once the threshold is reached the move is due.
// The game is only MARKED here (phase 2) and queued; the next select gives it no leaf, and the re-root
// (advance_game) then runs from the queue in its own launch, beside the tower of the other games — a deep
// subtree copy (one dependent round trip per tree level) no longer sits on every iteration's critical path.
// `forced`: force[g], loaded by the caller beside the state.
__device__ inline void mark_game(const EngineParams &P, int g, azh_game_state &s, int forced)
{
    if (s.phase == 1 && s.leaf_kind != AZH_LEAF_DESCENT && (s.root_visits >= P.visits || forced != 0)) {
        s.phase = 2;
        if (lane_id() == 0)
            P.adv_list[atomicAdd(P.adv_count, 1)] = g;
    }
}

__global__ __launch_bounds__(WAVE) void k_mark(EngineParams P)
{
    const int g = blockIdx.x;
    azh_game_state s = P.gs[g];
    const int phase = s.phase;
    mark_game(P, g, s, P.force[g]);
    if (s.phase != phase && threadIdx.x == 0)
        P.gs[g] = s;
}

__global__ __launch_bounds__(WAVE) void k_advance_list(EngineParams P)
{
    __shared__ TreeLds L;
    const int n = *P.adv_count;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        advance_game(P, P.adv_list[i], L);
        wave_sync();
    }
    // the workgroup that finishes last empties the queue (every workgroup has read the count by then): no memset
    // command behind the kernel, and the kernel's own completion is the event the next tree launch waits for
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(P.adv_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
        __hip_atomic_store(P.adv_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(P.adv_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The dense, game-ordered leaf list(s) of k_compact, written inside the tree launch by the workgroup that finishes
// last.  Every workgroup ORs its games' need bits into a bit mask (one returning agent-scope atomic per workgroup,
// performed before it draws its ticket from an agent-scope counter); the workgroup whose ticket is the last reads the
// mask — 16 bytes per 128 games — expands it in game order and clears it for the next launch.  Atomics and L1-bypassing
// loads on both sides, so no fence is needed (MI355X_MICROARCH.md, hand-off forms).  One kernel and one kernel boundary
// less per search iteration, and a tail of about a microsecond whatever the number of games.
template <int TREE_WAVES>
__device__ inline void compact_leaves(const EngineParams &P, int two, int *s_cnt /* [2][TREE_WAVES] */)
{
    constexpr int TREE_THREADS = TREE_WAVES * WAVE;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int nw = P.mask_words;
    const int c = (nw + TREE_THREADS - 1) / TREE_THREADS;  // consecutive mask words per thread (1 up to 8192 games)
    const int lo = min(nw, t * c), hi = min(nw, lo + c);
    u32 *m1p = P.need_mask, *m2p = P.need_mask + nw;
    int n1 = 0, n2 = 0;
    for (int k = lo; k < hi; k++) {
        n1 += __popc(__hip_atomic_load(&m1p[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (two)
            n2 += __popc(__hip_atomic_load(&m2p[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    const int i1 = wave_incl_scan(n1), i2 = wave_incl_scan(n2);
    if (lane == WAVE - 1) {
        s_cnt[w] = i1;
        s_cnt[TREE_WAVES + w] = i2;
    }
    __syncthreads();
    int b1 = i1 - n1, b2 = i2 - n2, t1 = 0, t2 = 0;
#pragma unroll
    for (int k = 0; k < TREE_WAVES; k++) {
        b1 += k < w ? s_cnt[k] : 0;
        b2 += k < w ? s_cnt[TREE_WAVES + k] : 0;
        t1 += s_cnt[k];
        t2 += s_cnt[TREE_WAVES + k];
    }
    for (int k = lo; k < hi; k++) {
        u32 m = __hip_atomic_load(&m1p[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m)
            __hip_atomic_store(&m1p[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (; m; m &= m - 1)
            P.leaf_list[b1++] = 32 * k + (__ffs((int)m) - 1);
        if (two) {
            u32 m2 = __hip_atomic_load(&m2p[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (m2)
                __hip_atomic_store(&m2p[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (; m2; m2 &= m2 - 1)
                P.leaf_list2[b2++] = 32 * k + (__ffs((int)m2) - 1);
        }
    }
    if (t == 0) {
        *P.leaf_count = t1;
        if (two)
            *P.leaf_count2 = t2;
        __hip_atomic_store(&P.tree_done[TICKET_SHARDS * TICKET_STRIDE], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The tree phase of a search iteration as ONE launch: a wave owns a game through backup -> "is the move due?" ->
// the next select, with the game's state in registers throughout; TREE_WAVES games share a workgroup (each wave on
// its own: wave_sync, never a workgroup barrier, inside a game), and the last workgroup to finish compacts the leaf
// list.  mode bit 0: backup + mark, bit 1: select (+ compaction).
template <bool STAMP, int TREE_WAVES>
__global__ __launch_bounds__(TREE_WAVES * WAVE) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_tree(EngineParams P, int mode, int two)
{
    __shared__ u16 s_moves[TREE_WAVES][MAX_MOVES];  // per game: the move list of the node being expanded
    __shared__ int s_cnt[2 * TREE_WAVES];
    __shared__ int s_need[TREE_WAVES];
    __shared__ int s_last;
    static_assert(32 % TREE_WAVES == 0, "a workgroup's need bits must lie in one word of the mask");
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: the game's addresses are scalars
    const int g = blockIdx.x * TREE_WAVES + w;
    u64 st[TREE_STAMPS] = {};
    int need = 0;
    if constexpr (STAMP) st[0] = tree_stamp();
    if (g < P.G) {
        azh_game_state s = P.gs[g];
        const int forced = P.force[g];  // (same round trip as the state)
        if constexpr (STAMP) st[1] = tree_stamp();
        if (mode & 1) {
            backup_game(P, g, s);
            if constexpr (STAMP) st[2] = tree_stamp();
            mark_game(P, g, s, forced);
        }
        if constexpr (STAMP) {
            st[3] = tree_stamp();
            if (!(mode & 1)) st[2] = st[3];
        }
        if (mode & 2)
            need = select_game<STAMP>(P, g, s, s_moves[w], st);  // stores the state
        else if (lane_id() == 0)
            P.gs[g] = s;
    }
    if (!(mode & 2))
        return;
    if (lane_id() == 0)
        s_need[w] = need;
    if constexpr (STAMP) st[6] = tree_stamp();
    __syncthreads();
    if constexpr (STAMP) {
        st[7] = tree_stamp();
        if (g < P.G && lane_id() < TREE_STAMPS) {
            u64 v = st[0];
#pragma unroll
            for (int k = 1; k < TREE_STAMPS; k++)
                v = lane_id() == k ? st[k] : v;
            P.stamps[(size_t)g * TREE_STAMPS + lane_id()] = v;
        }
    }
    if (threadIdx.x == 0) {
        u32 m1 = 0, m2 = 0;
#pragma unroll
        for (int k = 0; k < TREE_WAVES; k++) {
            const int v = s_need[k];
            m1 |= (u32)(two ? v == 1 : v != 0) << k;
            m2 |= (u32)(two && v == 2) << k;
        }
        const int g0 = blockIdx.x * TREE_WAVES;
        u32 seen = 0;
        if (m1)
            seen |= __hip_atomic_fetch_or(&P.need_mask[g0 >> 5], m1 << (g0 & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m2)
            seen |= __hip_atomic_fetch_or(&P.need_mask[P.mask_words + (g0 >> 5)], m2 << (g0 & 31), __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
        // the ORs have returned, i.e. have been performed, before the ticket is drawn
        asm volatile("s_waitcnt vmcnt(0)" : : "v"(seen) : "memory");
        // two-level ticket: the workgroup that completes its shard draws a ticket of the top counter
        const int shard = (int)(blockIdx.x % TICKET_SHARDS);
        const int in_shard = ((int)gridDim.x - 1 - shard) / TICKET_SHARDS + 1;
        int last = 0;
        if (__hip_atomic_fetch_add(&P.tree_done[shard * TICKET_STRIDE], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
            __hip_atomic_store(&P.tree_done[shard * TICKET_STRIDE], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int shards = min((int)gridDim.x, TICKET_SHARDS);
            last = __hip_atomic_fetch_add(&P.tree_done[TICKET_SHARDS * TICKET_STRIDE], 1, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) == shards - 1;
        }
        s_last = last;
    }
    __syncthreads();
    if (s_last)
        compact_leaves<TREE_WAVES>(P, two, s_cnt);
}

// Reference feature rows for the dense leaf list (cpp/self_play_client.cpp:174-202).
__global__ void k_features(const ulonglong2 *boards, const int *list, int n, u64 blockers, float *out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * 49)
        return;
    const int row = idx / 49, c = idx % 49;
    const int x = c / 7, y = c % 7, sq = x + 7 * (6 - y);
    const ulonglong2 b = boards[list ? list[row] : row];
    float4 f;
    f.x = 1.0f;
    f.y = (float)((b.x >> sq) & 1ULL);
    f.z = (float)((b.y >> sq) & 1ULL);
    f.w = (float)((blockers >> sq) & 1ULL);
    reinterpret_cast<float4 *>(out)[idx] = f;
}

__global__ void k_reduce_stats(const u64 *stats, int G, u64 *out)
{
    const int k = blockIdx.x;
    u64 acc = 0;
    for (int g = threadIdx.x; g < G; g += blockDim.x)
        acc += stats[(size_t)g * NSTAT + k];
    __shared__ u64 s_acc[256];
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off)
            s_acc[threadIdx.x] += s_acc[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[k] = s_acc[0];
}

}  // namespace azh

using namespace azh;

// ------------------------------------------------------------------ host

std::string azh_format_game_json(const uint32_t *rec, size_t words, bool with_ids);  // json.cpp
bool azh_record_well_formed(const uint32_t *rec, size_t avail, uint32_t max_plies, const char **why);

struct azh_engine {
    azh_config cfg;
    EngineParams P;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // AZH_REROOT_SIDE_STREAM=1 only: queued re-roots (k_advance_list) beside the tower, rounds 3-5's loop
    hipEvent_t ev_sel = nullptr, ev_adv = nullptr;
    std::vector<void *> allocs;
    float *d_feat = nullptr;
    float *d_sym_logits = nullptr, *d_sym_values = nullptr;  // AZH_FLAG_SYMMETRY_AVG scratch
    u64 *d_stat_out = nullptr;
    // finished games formatted but not yet handed out
    std::vector<std::string> pending;
    size_t pending_pos = 0;
    // records taken off the device (azh_engine_fetch) and not yet formatted
    std::vector<uint32_t> staged;
    // an explicit azh_engine_fetch covers the drain sequence that follows it: until that sequence has ended (a drain call
    // that hands out no game) a drain with nothing staged returns empty instead of fetching — the caller has enqueued
    // its next run in between, and a fetch would wait for that run
    bool fetch_covers_drain = false;
    // work has been enqueued since the last fetch (its finished games are still on the device): a drain by a caller that
    // never fetches fetches when this is set — once per round, whether it drains in a loop until nothing comes or once
    bool unfetched_work = true;
    long long implicit_fetches = 0;  // fetches made by azh_engine_drain_json itself (azh_engine_implicit_fetches)
    // uid-ordered emission: finished games wait here until every game with a smaller uid has been handed out or
    // is known to have been dropped ("" = dropped)
    bool emit_by_uid = false;
    std::map<uint32_t, std::string> held;
    uint32_t next_uid = 0;
    bool order_broken = false;  // records were lost (ring overflow): some uid will never come, see azh_engine_drain_json
    // timing: every `timing_stride`-th iteration of the device loop is bracketed by events on the engine's stream
    // (tower start, tower end, start of the next tower = end of the tree phase).  Sampling keeps the event packets —
    // each a barrier in the queue, a few microseconds — out of most iterations of the region being measured.
    int timing_stride = 0;  // 0 = off
    long long loop_iter = 0;
    std::vector<hipEvent_t> events;  // 3 per sample: tower start, tower end, tree phase end
    size_t samples = 0;
    std::vector<int> close_of;       // per sample: index of the event that closes its tree phase
    bool close_pending = false;      // the last sample's tree phase is closed at the start of the next iteration
    int *h_count = nullptr;  // pinned
    // pinned staging of azh_engine_fetch: the ring's head and its records are copied here (a copy into pageable memory is
    // staged by the runtime through buffers of its own and may serialise with other streams' copies), then into `staged`
    u64 *h_head = nullptr;
    u32 *h_stage = nullptr;
    size_t h_stage_words = 0;
    std::vector<u32 *> h_stage_retired;
    bool selected = false;
    bool arena_lists = false;  // run_arena: one leaf list per net
    bool stamp_next = false;   // azh_engine_tree_stamps: the next fused tree launch of the loop is the stamped instantiation
    int thin_mode = -1;        // azh_engine_set_thin_batches: 0 the 3-board tower, 1 one board per workgroup, -1 by the engine's size
    int adv_workers = 8;       // workgroups at the head of every tower launch of the device loop that play the queued moves
};

// the tower for this engine's leaf batches: one board per workgroup for engines (or batches the host says are) small
static int thin_batches(const azh_engine *e)
{
    return e->thin_mode >= 0 ? e->thin_mode : (e->P.G <= AZH_THIN_MAX_GAMES ? 1 : 0);
}

static const size_t MAX_TIMED_SAMPLES = 8192;

template <typename T> static int dev_alloc(azh_engine *e, T **p, size_t count)
{
    void *q = nullptr;
    AZH_HIP(hipMalloc(&q, count * sizeof(T)));
    AZH_HIP(hipMemset(q, 0, count * sizeof(T)));
    e->allocs.push_back(q);
    *p = (T *)q;
    return 0;
}

extern "C" int azh_engine_create(const azh_config *cfg, azh_engine **out)
{
    if (!cfg || !out)
        return azh_fail(-1, "azh_engine_create: null argument");
    if (cfg->games <= 0 || cfg->visits <= 0 || cfg->visits > 60000 || cfg->max_plies <= 0 ||
        cfg->edges_per_node < 8)
        return azh_fail(-2, "azh_engine_create: bad config (games %d visits %d max_plies %d edges_per_node %d)",
                        cfg->games, cfg->visits, cfg->max_plies, cfg->edges_per_node);
    if ((long long)(cfg->visits + 8) * cfg->edges_per_node >= (1 << 23))
        return azh_fail(-2, "azh_engine_create: (visits + 8) * edges_per_node = %lld edges per game do not fit the 23-bit edge "
                            "index of the tree's edge records", (long long)(cfg->visits + 8) * cfg->edges_per_node);
    if (azh_require_device())
        return -3;
    azh_engine *e = new azh_engine();
    e->cfg = *cfg;
    EngineParams &P = e->P;
    memset(&P, 0, sizeof(P));
    P.G = cfg->games;
    P.visits = cfg->visits;
    P.node_cap = cfg->visits + 8;
    P.edge_cap = P.node_cap * cfg->edges_per_node;
    P.path_cap = P.node_cap;
    P.max_plies = cfg->max_plies;
    P.c_puct = cfg->c_puct;
    P.alpha = cfg->dirichlet_alpha;
    P.noise_w = cfg->dirichlet_weight;
    P.k0 = (u32)cfg->seed;
    P.k1 = (u32)(cfg->seed >> 32);
    P.start_x = cfg->start_x;
    P.start_o = cfg->start_o;
    P.blockers = cfg->blockers;
    P.start_turn = cfg->start_turn;
    P.flags = cfg->flags;
    P.select_budget = (int)cfg->select_budget;

    const size_t G = (size_t)P.G;
    // ring of finished-game records (64 KiB per slot between two drains)
    P.ring_cap_words = std::max<size_t>((size_t)1 << 22, G * 16384);
    int rc = 0;
    rc |= dev_alloc(e, &P.gs, G);
    rc |= dev_alloc(e, &P.force, G);
    rc |= dev_alloc(e, &P.no_emit, G);
    P.tt_size = 1;
    while (P.tt_size < 4 * P.node_cap)
        P.tt_size <<= 1;
    if (cfg->flags & AZH_FLAG_EVAL_CACHE)
        rc |= dev_alloc(e, &P.tt, 2 * G * (size_t)P.tt_size);
    rc |= dev_alloc(e, &P.adv_list, G);
    rc |= dev_alloc(e, &P.adv_count, 1);
    rc |= dev_alloc(e, &P.adv_done, 1);
    rc |= dev_alloc(e, &P.path, G * P.path_cap);
    rc |= dev_alloc(e, &P.node_board, 2 * G * P.node_cap);
    rc |= dev_alloc(e, &P.node_info, 2 * G * P.node_cap);
    rc |= dev_alloc(e, &P.edge, 2 * G * P.edge_cap);
    rc |= dev_alloc(e, &P.edge_move, 2 * G * P.edge_cap);
    rc |= dev_alloc(e, &P.leaf_board, G);
    rc |= dev_alloc(e, &P.need_eval, G);
    rc |= dev_alloc(e, &P.leaf_list, G);
    rc |= dev_alloc(e, &P.leaf_count, 1);
    rc |= dev_alloc(e, &P.leaf_list2, G);
    rc |= dev_alloc(e, &P.leaf_count2, 1);
    rc |= dev_alloc(e, &P.tree_done, (size_t)TICKET_SHARDS * TICKET_STRIDE + 1);
    P.mask_words = (P.G + 31) / 32;
    rc |= dev_alloc(e, &P.need_mask, 2 * (size_t)P.mask_words);
    rc |= dev_alloc(e, &P.logits, G * AZH_POLICY_SIZE);
    rc |= dev_alloc(e, &P.values, G);
    rc |= dev_alloc(e, &P.rec, G * P.max_plies * REC_STRIDE_WORDS);
    rc |= dev_alloc(e, &P.ring, P.ring_cap_words);
    rc |= dev_alloc(e, &P.ring_head, 1);
    rc |= dev_alloc(e, &P.stats, G * NSTAT);
    rc |= dev_alloc(e, &P.bfs_spill, G * 3 * P.node_cap);
    rc |= dev_alloc(e, &e->d_stat_out, NSTAT);
    if (rc) {
        azh_engine_destroy(e);
        return rc;
    }
    // the side stream of the re-roots gets the highest priority the device offers: its few waves must not queue
    // behind the tower's 1,200 workgroups for the slots those free
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    if (hipStreamCreate(&e->stream) != hipSuccess ||
        hipStreamCreateWithPriority(&e->stream2, hipStreamDefault, prio_high) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_sel, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_adv, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void **)&e->h_count, sizeof(int)) != hipSuccess ||
        hipHostMalloc((void **)&e->h_head, sizeof(u64)) != hipSuccess) {
        azh_engine_destroy(e);
        return azh_fail(-4, "azh_engine_create: stream / pinned allocation failed");
    }
    // one move-playing workgroup per 128 game slots (an iteration queues about one game in 400 at 400 sims/move, one in 200
    // at 200), at least 8, at most 64, a multiple of 8 (the tiles behind them keep their XCD)
    e->adv_workers = std::min(64, std::max(8, (P.G / 128 + 7) / 8 * 8));
    hipLaunchKernelGGL(k_init, dim3(P.G), dim3(WAVE), 0, e->stream, P);
    if (hipStreamSynchronize(e->stream) != hipSuccess) {
        azh_engine_destroy(e);
        return azh_fail(-5, "azh_engine_create: init kernel failed");
    }
    *out = e;
    return 0;
}

extern "C" void azh_engine_destroy(azh_engine *e)
{
    if (!e)
        return;
    if (e->stream)
        (void)hipStreamSynchronize(e->stream);
    if (e->stream2)
        (void)hipStreamSynchronize(e->stream2);  // (a re-root launch may still be running under an abandoned run)
    for (auto ev : e->events)
        (void)hipEventDestroy(ev);
    for (void *p : e->allocs)
        (void)hipFree(p);
    if (e->d_feat)
        (void)hipFree(e->d_feat);
    if (e->d_sym_logits)
        (void)hipFree(e->d_sym_logits);
    if (e->d_sym_values)
        (void)hipFree(e->d_sym_values);
    if (e->h_count)
        (void)hipHostFree(e->h_count);
    if (e->h_head)
        (void)hipHostFree(e->h_head);
    if (e->h_stage)
        (void)hipHostFree(e->h_stage);
    for (u32 *q : e->h_stage_retired)
        (void)hipHostFree(q);
    if (e->ev_sel)
        (void)hipEventDestroy(e->ev_sel);
    if (e->ev_adv)
        (void)hipEventDestroy(e->ev_adv);
    if (e->stream2)
        (void)hipStreamDestroy(e->stream2);
    if (e->stream)
        (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int azh_engine_node_cap(const azh_engine *e) { return e ? e->P.node_cap : -1; }
extern "C" int azh_engine_edge_cap(const azh_engine *e) { return e ? e->P.edge_cap : -1; }

static int enqueue_compact(azh_engine *e);

constexpr int ADV_GRID = 64;  // one wave each; a search iteration queues G * (1 / visits + ...) re-roots: about 11 at 4096 games

// the queued re-roots, on `stream` (always after a select has passed over the queued games); `done`, if given, is
// signalled by the kernel's own completion (no separate event packet in the queue)
static int enqueue_advance(azh_engine *e, hipStream_t stream, hipEvent_t done = nullptr)
{
    e->unfetched_work = true;  // (every path that can finish a game goes through here)
    hipExtLaunchKernelGGL(k_advance_list, dim3(ADV_GRID), dim3(WAVE), 0, stream, nullptr, done, 0, e->P);
    AZH_HIP(hipGetLastError());
    return 0;
}

static int enqueue_select(azh_engine *e)
{
    hipLaunchKernelGGL(k_select, dim3(e->P.G), dim3(WAVE), 0, e->stream, e->P);
    if (enqueue_compact(e))
        return -1;
    return enqueue_advance(e, e->stream);
}

static int enqueue_backup(azh_engine *e)
{
    hipLaunchKernelGGL(k_backup, dim3(e->P.G), dim3(WAVE), 0, e->stream, e->P);
    hipLaunchKernelGGL(k_mark, dim3(e->P.G), dim3(WAVE), 0, e->stream, e->P);
    AZH_HIP(hipGetLastError());
    return 0;
}

extern "C" int azh_engine_select(azh_engine *e, int32_t *n_leaves_out)
{
    if (!e)
        return azh_fail(-1, "azh_engine_select: null engine");
    if (enqueue_select(e))
        return -1;
    AZH_HIP(hipMemcpyAsync(e->h_count, e->P.leaf_count, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    AZH_HIP(hipStreamSynchronize(e->stream));
    e->selected = true;
    if (n_leaves_out)
        *n_leaves_out = *e->h_count;
    return 0;
}

extern "C" int azh_engine_leaves(azh_engine *e, int32_t *need_eval, uint64_t *leaf_boards)
{
    if (!e)
        return azh_fail(-1, "azh_engine_leaves: null engine");
    AZH_HIP(hipStreamSynchronize(e->stream));
    if (need_eval)
        AZH_HIP(hipMemcpy(need_eval, e->P.need_eval, (size_t)e->P.G * 4, hipMemcpyDeviceToHost));
    if (leaf_boards)
        AZH_HIP(hipMemcpy(leaf_boards, e->P.leaf_board, (size_t)e->P.G * 16, hipMemcpyDeviceToHost));
    return 0;
}

// Dense feature rows of the current leaf batch, in leaf-list (game) order.
extern "C" int azh_engine_leaf_features(azh_engine *e, float *out, int32_t *games_out)
{
    if (!e || !out)
        return azh_fail(-1, "azh_engine_leaf_features: null argument");
    AZH_HIP(hipStreamSynchronize(e->stream));
    int n = 0;
    AZH_HIP(hipMemcpy(&n, e->P.leaf_count, 4, hipMemcpyDeviceToHost));
    if (n <= 0)
        return 0;
    if (!e->d_feat)
        AZH_HIP(hipMalloc((void **)&e->d_feat, (size_t)e->P.G * AZH_FEATURE_SIZE * 4));
    hipLaunchKernelGGL(k_features, dim3((n * 49 + 255) / 256), dim3(256), 0, e->stream,
                       (const ulonglong2 *)e->P.leaf_board, (const int *)e->P.leaf_list, n, e->P.blockers, e->d_feat);
    AZH_HIP(hipGetLastError());
    AZH_HIP(hipStreamSynchronize(e->stream));
    AZH_HIP(hipMemcpy(out, e->d_feat, (size_t)n * AZH_FEATURE_SIZE * 4, hipMemcpyDeviceToHost));
    if (games_out)
        AZH_HIP(hipMemcpy(games_out, e->P.leaf_list, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

// one evaluation of the listed leaves by `net`; with AZH_FLAG_SYMMETRY_AVG the 8-way averaged one
static int launch_eval(azh_engine *e, azh_net *net, int dtype, const int *list, const int *count, const AdvanceHook *hook = nullptr)
{
    if (!(e->P.flags & AZH_FLAG_SYMMETRY_AVG))
        return azh_net_launch(net, dtype, (const unsigned long long *)e->P.leaf_board, list, count, e->P.G,
                              e->P.blockers, e->P.logits, e->P.values, e->stream, nullptr, thin_batches(e), hook);
    if (!e->d_sym_logits) {
        AZH_HIP(hipMalloc((void **)&e->d_sym_logits, (size_t)e->P.G * 8 * AZH_POLICY_SIZE * 4));
        AZH_HIP(hipMalloc((void **)&e->d_sym_values, (size_t)e->P.G * 8 * 4));
    }
    return azh_net_launch_sym(net, dtype, (const unsigned long long *)e->P.leaf_board, list, count, e->P.G,
                              e->P.blockers, e->d_sym_logits, e->d_sym_values, e->P.logits, e->P.values, e->stream,
                              e->thin_mode >= 0 ? e->thin_mode : (e->P.G * 8 <= AZH_THIN_MAX_GAMES ? 1 : 0),  // (8 virtual boards per leaf)
                              hook);
}

extern "C" int azh_engine_eval(azh_engine *e, azh_net *net, int dtype)
{
    if (!e || !net)
        return azh_fail(-1, "azh_engine_eval: null argument");
    return launch_eval(e, net, dtype, e->P.leaf_list, e->P.leaf_count);
}

extern "C" int azh_engine_set_evals(azh_engine *e, const float *logits, const float *values)
{
    if (!e || !logits || !values)
        return azh_fail(-1, "azh_engine_set_evals: null argument");
    AZH_HIP(hipMemcpyAsync(e->P.logits, logits, (size_t)e->P.G * AZH_POLICY_SIZE * 4, hipMemcpyHostToDevice, e->stream));
    AZH_HIP(hipMemcpyAsync(e->P.values, values, (size_t)e->P.G * 4, hipMemcpyHostToDevice, e->stream));
    AZH_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int azh_engine_backup(azh_engine *e)
{
    if (!e)
        return azh_fail(-1, "azh_engine_backup: null engine");
    if (enqueue_backup(e))
        return -1;
    AZH_HIP(hipStreamSynchronize(e->stream));
    e->selected = false;
    return 0;
}

static int enqueue_compact(azh_engine *e)
{
    const int two = (e->P.flags & AZH_FLAG_TWO_NETS) && e->arena_lists;
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, e->stream, (const int *)e->P.need_eval, e->P.G, e->P.leaf_list,
                       e->P.leaf_count, 1, two ? 0 : 1);
    if (two)
        hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, e->stream, (const int *)e->P.need_eval, e->P.G,
                           e->P.leaf_list2, e->P.leaf_count2, 2, 0);
    AZH_HIP(hipGetLastError());
    return 0;
}

// Device-resident loop.  Iteration = select -> tower -> backup -> "is the move due?"; consecutive iterations run
// backup + mark + the next select of one game in a single fused launch (k_tree), and the queued re-roots
// inside the tower launch that follows (its first workgroups).
//
// The queued moves (sample, record, re-root: advance_game) are played by the first workgroups of the tower launch that
// follows the tree launch which queued them — dispatched before any tile's workgroup, done after ~0.1 ms, and the next
// tree launch, behind the tower on the same stream, finds every move played: one stream, two launches per iteration.
// Rounds 3-5 ran them as k_advance_list on a high-priority side stream beside the tower, with the tree launches' and
// its own completion signals as cross-stream events: the tower leaves no wave slot and no LDS beside itself, so that
// launch spent most of its 0.2-0.5 ms waiting for the tower's first workgroups to retire (and outlasted the tower in
// 5-9 % of the iterations), and the event machinery cost an iteration 7-10 us with nothing queued
// (profiles/round6_reroots_in_the_tower_launch.txt).  AZH_REROOT_SIDE_STREAM=1 runs that loop (A/B runs; read per call).
//
// A run is enqueued as begin() + `iterations` x iteration(): azh_engine_run does that for one engine, azh_engines_run for
// several with their iterations interleaved, so that the half-batches of a GPU all start with the first launches enqueued
// instead of one after the other's whole run.
struct RunLoop {
    azh_engine *e;
    azh_net *net_a, *net_b;
    int dtype, iterations;
    int two = 0;
    bool pair = false, side = false;
    AdvanceHook hook;

    // one fused tree launch; ev (or nullptr) is signalled by the kernel's own completion
    void launch_tree(bool stamped, int mode, hipEvent_t ev)
    {
        const bool small = e->P.G <= TREE_ONE_ROUND_GAMES;
        const int waves = small ? TREE_WAVES_SMALL : TREE_WAVES_LARGE;
        const dim3 grid((e->P.G + waves - 1) / waves), block(waves * WAVE);
        if (stamped && small)
            hipExtLaunchKernelGGL((k_tree<true, TREE_WAVES_SMALL>), grid, block, 0, e->stream, nullptr, ev, 0, e->P, mode, two);
        else if (stamped)
            hipExtLaunchKernelGGL((k_tree<true, TREE_WAVES_LARGE>), grid, block, 0, e->stream, nullptr, ev, 0, e->P, mode, two);
        else if (small)
            hipExtLaunchKernelGGL((k_tree<false, TREE_WAVES_SMALL>), grid, block, 0, e->stream, nullptr, ev, 0, e->P, mode, two);
        else
            hipExtLaunchKernelGGL((k_tree<false, TREE_WAVES_LARGE>), grid, block, 0, e->stream, nullptr, ev, 0, e->P, mode, two);
    }
    // side-stream mode: ev_sel is signalled by the tree launch itself, ev_adv by the re-root launch (hipExtLaunchKernelGGL's
    // stop event: no event-record packets at the kernel boundaries of the main stream)
    int side_advance()
    {
        if (!side)
            return 0;
        AZH_HIP(hipStreamWaitEvent(e->stream2, e->ev_sel, 0));
        return enqueue_advance(e, e->stream2, e->ev_adv);
    }

    int begin()
    {
        two = (e->P.flags & AZH_FLAG_TWO_NETS) && e->arena_lists;  // one leaf list per net
        // arena: the two nets' towers as one launch (AZH_ARENA_PAIR=0: two launches back to back, for A/B runs)
        const char *pair_s = getenv("AZH_ARENA_PAIR");  // (read per call: a test switches it inside one process)
        const bool pair_env = !(pair_s && atoi(pair_s) == 0);
        pair = pair_env && two && !(e->P.flags & AZH_FLAG_SYMMETRY_AVG);
        const char *side_s = getenv("AZH_REROOT_SIDE_STREAM");
        side = side_s && atoi(side_s) != 0;
        hook.workers = e->adv_workers;
        hook.at_head = 1;   // (decided per launch by the tower's launch functions: in front only where workgroups queue for slots)
        hook.P = e->P;
        e->unfetched_work = true;  // (every path that can finish a game goes through a re-root)
        launch_tree(false, 2, side ? e->ev_sel : nullptr);  // select + leaf list
        return side_advance();
    }

    int iteration(int it)
    {
        const AdvanceHook *moves = side ? nullptr : &hook;
        const bool rec = e->timing_stride > 0 && e->loop_iter % e->timing_stride == 0 && e->samples < MAX_TIMED_SAMPLES;
        hipEvent_t *ev = rec ? &e->events[3 * e->samples] : nullptr;
        if (e->close_pending) {
            // the tree phase of the previous sample ends where this tower starts
            if (rec) {
                e->close_of[e->samples - 1] = (int)(3 * e->samples);  // this sample's start event doubles as the end
            } else {
                AZH_HIP(hipEventRecord(e->events[3 * (e->samples - 1) + 2], e->stream));
                e->close_of[e->samples - 1] = (int)(3 * (e->samples - 1) + 2);
            }
            e->close_pending = false;
        }
        if (rec) AZH_HIP(hipEventRecord(ev[0], e->stream));
        int rc = 1;
        if (net_b && pair)  // both nets' leaf lists in ONE tower launch (1: not applicable -> one after the other)
            rc = azh_net_launch_pair(net_a, net_b, dtype, (const unsigned long long *)e->P.leaf_board, e->P.leaf_list,
                                     e->P.leaf_count, e->P.leaf_list2, e->P.leaf_count2, e->P.G, e->P.blockers, e->P.logits,
                                     e->P.values, e->stream, thin_batches(e), moves);
        if (rc == 1) {
            rc = launch_eval(e, net_a, dtype, e->P.leaf_list, e->P.leaf_count, moves);   // (the first launch plays the moves)
            if (rc == 0 && net_b)
                rc = launch_eval(e, net_b, dtype, e->P.leaf_list2, e->P.leaf_count2);
        }
        if (rc) return rc;
        if (rec) AZH_HIP(hipEventRecord(ev[1], e->stream));
        const int last = it + 1 == iterations;
        if (side)
            AZH_HIP(hipStreamWaitEvent(e->stream, e->ev_adv, 0));
        if (e->stamp_next && !last) {
            launch_tree(true, 3, side ? e->ev_sel : nullptr);
            e->stamp_next = false;
        } else {
            launch_tree(false, last ? 1 : 3, last || !side ? nullptr : e->ev_sel);
        }
        AZH_HIP(hipGetLastError());
        if (!last && side_advance()) return -1;
        if (rec) {
            e->samples++;
            if (last) {
                AZH_HIP(hipEventRecord(ev[2], e->stream));
                e->close_of[e->samples - 1] = (int)(3 * (e->samples - 1) + 2);
            } else {
                e->close_pending = true;
            }
        }
        e->loop_iter++;
        return 0;
    }
};

static int run_loop(azh_engine *e, azh_net *net_a, azh_net *net_b, int dtype, int iterations)
{
    if (iterations <= 0)
        return 0;
    RunLoop L{e, net_a, net_b, dtype, iterations};
    int rc = L.begin();
    for (int it = 0; it < iterations && rc == 0; it++)
        rc = L.iteration(it);
    return rc;
}

// The same run for several engines of one GPU (the half-batches of selfplay.SelfPlay), their iterations enqueued in turn:
// enqueueing engine after engine lets the second one start only when the first one's whole run — two launches per iteration,
// a few milliseconds of host time per 250 iterations — has been enqueued, and finish that much later, alone on the chip.
extern "C" int azh_engines_run(azh_engine *const *engines, int n, azh_net *net, int dtype, int iterations)
{
    if (!engines || n < 1 || !net || iterations < 0)
        return azh_fail(-1, "azh_engines_run: bad argument");
    for (int i = 0; i < n; i++)
        if (!engines[i])
            return azh_fail(-1, "azh_engines_run: null engine");
    if (iterations == 0)
        return 0;
    std::vector<RunLoop> loops;
    for (int i = 0; i < n; i++)
        loops.push_back(RunLoop{engines[i], net, nullptr, dtype, iterations});
    int rc = 0;
    for (int i = 0; i < n && rc == 0; i++)
        rc = loops[i].begin();
    for (int it = 0; it < iterations && rc == 0; it++)
        for (int i = 0; i < n && rc == 0; i++)
            rc = loops[i].iteration(it);
    return rc;
}

extern "C" int azh_engine_run(azh_engine *e, azh_net *net, int dtype, int iterations)
{
    if (!e || !net || iterations < 0)
        return azh_fail(-1, "azh_engine_run: bad argument");
    return run_loop(e, net, nullptr, dtype, iterations);
}

extern "C" int azh_engine_run_arena(azh_engine *e, azh_net *net_a, azh_net *net_b, int dtype, int iterations)
{
    if (!e || !net_a || !net_b || iterations < 0)
        return azh_fail(-1, "azh_engine_run_arena: bad argument");
    if (!(e->P.flags & AZH_FLAG_TWO_NETS))
        return azh_fail(-2, "azh_engine_run_arena: engine was not created with AZH_FLAG_TWO_NETS");
    e->arena_lists = true;
    const int rc = run_loop(e, net_a, net_b, dtype, iterations);
    e->arena_lists = false;
    return rc;
}

// Diagnostic: two search iterations of the device loop, the tree launch between the two towers being the stamped
// instantiation of k_tree; out [games][10] u64 = s_memrealtime readings (100 MHz) per game: wave start, state loaded,
// backup done, mark done, descent done, expansion done, state stored, workgroup done; then the levels descended and the
// children scanned by that descent.
extern "C" int azh_engine_tree_stamps(azh_engine *e, azh_net *net, int dtype, uint64_t *out)
{
    if (!e || !net || !out)
        return azh_fail(-1, "azh_engine_tree_stamps: null argument");
    if (!e->P.stamps) {
        void *q = nullptr;
        AZH_HIP(hipMalloc(&q, (size_t)e->P.G * TREE_STAMPS * 8));
        e->allocs.push_back(q);
        e->P.stamps = (u64 *)q;
    }
    AZH_HIP(hipMemsetAsync(e->P.stamps, 0, (size_t)e->P.G * TREE_STAMPS * 8, e->stream));
    e->stamp_next = true;
    const int rc = run_loop(e, net, nullptr, dtype, 2);
    e->stamp_next = false;
    if (rc)
        return rc;
    AZH_HIP(hipStreamSynchronize(e->stream));
    AZH_HIP(hipMemcpy(out, e->P.stamps, (size_t)e->P.G * TREE_STAMPS * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int azh_engine_set_thin_batches(azh_engine *e, int mode)
{
    if (!e || mode < -1 || mode > 1)
        return azh_fail(-1, "azh_engine_set_thin_batches: mode is -1 (by size), 0 or 1");
    e->thin_mode = mode;  // (host state only: takes effect with the next launches enqueued)
    return 0;
}

// Root-visit threshold for the coming moves (1 .. the value the engine was created with; the
// arenas are sized for that).  Takes effect at the next k_advance.
extern "C" int azh_engine_set_visits(azh_engine *e, int visits)
{
    if (!e || visits < 1 || visits > e->cfg.visits)
        return azh_fail(-1, "azh_engine_set_visits: need 1 <= visits <= %d", e ? e->cfg.visits : 0);
    AZH_HIP(hipStreamSynchronize(e->stream));
    e->P.visits = visits;
    return 0;
}

// At most `games` games are played: uids 0 .. games - 1 (slot g plays uids g, g + G, ...).  A slot whose next game
// would be past the limit goes idle, so the batch thins out as the last games end and no search is spent on games
// nobody asked for — what a generator given a target count (accelerated_generate_games.py --game-count) wants in
// uid order, where line N only appears once the slowest of the first N games has ended.  The limit may be RAISED later
// (games that were dropped leave the caller short of lines): idle slots whose next game is now below it start it.
// Games that have begun are never stopped.
extern "C" int azh_engine_set_game_limit(azh_engine *e, int64_t games)
{
    if (!e || games < 1 || games > 0xFFFFFFFFll)
        return azh_fail(-1, "azh_engine_set_game_limit: bad argument");
    AZH_HIP(hipStreamSynchronize(e->stream));
    AZH_HIP(hipStreamSynchronize(e->stream2));
    e->P.uid_limit = (u32)games;
    hipLaunchKernelGGL(k_limit_slots, dim3(e->P.G), dim3(WAVE), 0, e->stream, e->P);
    AZH_HIP(hipGetLastError());
    AZH_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

// Order in which finished games are handed out.  0 (default): as they finish.  1: by game uid — a game waits until
// every game with a smaller uid has been written or dropped.  Games that finish first are the SHORT ones, so a
// consumer that stops reading after N lines (looper.py:51-64 kills the generator at --game-count lines) gets a
// length-biased sample in finish order; in uid order the first N lines are the first N games started, whatever
// their length.
extern "C" int azh_engine_set_emit_order(azh_engine *e, int by_uid)
{
    if (!e)
        return azh_fail(-1, "azh_engine_set_emit_order: null engine");
    if (e->next_uid != 0 || !e->held.empty())
        return azh_fail(-2, "azh_engine_set_emit_order: games have already been handed out in uid order");
    e->emit_by_uid = by_uid != 0;
    return 0;
}

// Every slot restarts at a given position: boards [G][2] packed (x | turn << 63, o), plies [G] (the ply the position is at;
// < max_plies).  Fresh trees, uids = slot numbers.  Games started this way are played and counted like any other, but
// their records would lack the plies before the start, so they are not written (the slot's NEXT game is a normal one).
// A measurement set-up hook: bench.py loads the positions a long-running generator was found at
// (profiles/round2_steady_state_positions.npz) instead of waiting a game generation for the steady state to form.
extern "C" int azh_engine_set_positions(azh_engine *e, const uint64_t *boards, const int32_t *plies)
{
    if (!e || !boards || !plies)
        return azh_fail(-1, "azh_engine_set_positions: null argument");
    AZH_HIP(hipStreamSynchronize(e->stream));
    AZH_HIP(hipStreamSynchronize(e->stream2));
    const size_t G = (size_t)e->P.G;
    // uids restart at the slot numbers: whatever the previous games left behind (undrained records, games held back
    // for uid order) belongs to uids that are about to be reused, and is discarded with them
    e->pending.clear();
    e->pending_pos = 0;
    e->staged.clear();
    e->held.clear();
    e->next_uid = 0;
    e->order_broken = false;
    AZH_HIP(hipMemsetAsync(e->P.ring_head, 0, 8, e->stream));
    for (size_t g = 0; g < G; g++) {
        const uint64_t x = boards[2 * g] & ~TURN_BIT, o = boards[2 * g + 1];
        if (plies[g] < 0 || plies[g] >= e->P.max_plies || (x & o) || ((x | o) & (e->P.blockers | ~BOARD_MASK)) || !x || !o)
            return azh_fail(-2, "azh_engine_set_positions: slot %zu: bad position or ply %d", g, plies[g]);
    }
    ulonglong2 *d_b = nullptr;
    int *d_p = nullptr;
    hipError_t rc = hipMalloc((void **)&d_b, G * 16);
    if (rc == hipSuccess) rc = hipMalloc((void **)&d_p, G * 4);
    if (rc == hipSuccess) rc = hipMemcpy(d_b, boards, G * 16, hipMemcpyHostToDevice);
    if (rc == hipSuccess) rc = hipMemcpy(d_p, plies, G * 4, hipMemcpyHostToDevice);
    if (rc == hipSuccess) rc = hipMemsetAsync(e->P.adv_count, 0, sizeof(int), e->stream);
    if (rc == hipSuccess) {
        hipLaunchKernelGGL(k_init_positions, dim3(e->P.G), dim3(WAVE), 0, e->stream, e->P, (const ulonglong2 *)d_b, (const int *)d_p);
        rc = hipStreamSynchronize(e->stream);
    }
    (void)hipFree(d_b);
    (void)hipFree(d_p);
    AZH_HIP(rc);
    return 0;
}

extern "C" int azh_engine_sync(azh_engine *e)
{
    if (!e)
        return azh_fail(-1, "azh_engine_sync: null engine");
    AZH_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int azh_engine_game_state(azh_engine *e, int game, azh_game_state *out)
{
    if (!e || !out || game < 0 || game >= e->P.G)
        return azh_fail(-1, "azh_engine_game_state: bad argument");
    AZH_HIP(hipStreamSynchronize(e->stream));
    AZH_HIP(hipMemcpy(out, e->P.gs + game, sizeof(azh_game_state), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int azh_engine_tree(azh_engine *e, int game, uint64_t *boards, uint32_t *info, uint32_t *edges,
                               uint16_t *moves)
{
    if (!e || game < 0 || game >= e->P.G)
        return azh_fail(-1, "azh_engine_tree: bad argument");
    azh_game_state s;
    if (azh_engine_game_state(e, game, &s))
        return -1;
    const size_t slot = (size_t)s.arena * e->P.G + game;
    if (boards) AZH_HIP(hipMemcpy(boards, e->P.node_board + slot * e->P.node_cap, (size_t)s.n_nodes * 16, hipMemcpyDeviceToHost));
    if (info) AZH_HIP(hipMemcpy(info, e->P.node_info + slot * e->P.node_cap, (size_t)s.n_nodes * 16, hipMemcpyDeviceToHost));
    if (edges) {
        AZH_HIP(hipMemcpy(edges, e->P.edge + slot * e->P.edge_cap, (size_t)s.n_edges * 16, hipMemcpyDeviceToHost));
        for (int j = 0; j < s.n_edges; j++) {  // device record (prior, W, visits | child << 16, child's range) -> documented row
            uint32_t *r = edges + 4 * (size_t)j;
            const uint32_t w = r[1], z = r[2];
            r[0] &= PRIOR_MASK;  // (the sign bit is the descent's mark, not part of the tree)
            r[1] = z & 0xFFFFu;
            r[2] = w;
            r[3] = (z >> 16) == 0xFFFFu ? 0xFFFFFFFFu : (z >> 16);
        }
    }
    if (moves) AZH_HIP(hipMemcpy(moves, e->P.edge_move + slot * e->P.edge_cap, (size_t)s.n_edges * 2, hipMemcpyDeviceToHost));
    return 0;
}

// Diagnostic: the game's edge records as they lie in HBM (the 16-byte device record described at the top of this file,
// the descent's mark in bit 31 of word 0 included) — what azh_engine_tree turns into the documented rows.  For the test
// of the mark's invariants: at most one marked edge per node, and only on an edge whose child exists, is unfinished and
// has 1 .. 128 moves.
extern "C" int azh_engine_tree_raw(azh_engine *e, int game, uint32_t *edges)
{
    if (!e || !edges || game < 0 || game >= e->P.G)
        return azh_fail(-1, "azh_engine_tree_raw: bad argument");
    azh_game_state s;
    if (azh_engine_game_state(e, game, &s))
        return -1;
    const size_t slot = (size_t)s.arena * e->P.G + game;
    AZH_HIP(hipMemcpy(edges, e->P.edge + slot * e->P.edge_cap, (size_t)s.n_edges * 16, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int azh_engine_stats(azh_engine *e, uint64_t *out)
{
    if (!e || !out)
        return azh_fail(-1, "azh_engine_stats: bad argument");
    hipLaunchKernelGGL(k_reduce_stats, dim3(NSTAT), dim3(256), 0, e->stream, (const u64 *)e->P.stats, e->P.G, e->d_stat_out);
    AZH_HIP(hipGetLastError());
    AZH_HIP(hipStreamSynchronize(e->stream));
    AZH_HIP(hipMemcpy(out, e->d_stat_out, NSTAT * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int azh_engine_timing_reset(azh_engine *e, int enable)
{
    if (!e)
        return azh_fail(-1, "azh_engine_timing_reset: null engine");
    AZH_HIP(hipStreamSynchronize(e->stream));
    e->timing_stride = enable > 0 ? enable : 0;
    e->samples = 0;
    e->close_pending = false;
    e->loop_iter = 0;
    if (e->timing_stride && e->events.empty()) {
        e->events.resize(3 * MAX_TIMED_SAMPLES);
        for (auto &ev : e->events)
            AZH_HIP(hipEventCreate(&ev));
        e->close_of.assign(MAX_TIMED_SAMPLES, -1);
    }
    return 0;
}

extern "C" int azh_engine_timing(azh_engine *e, azh_timing *out)
{
    if (!e || !out)
        return azh_fail(-1, "azh_engine_timing: bad argument");
    AZH_HIP(hipStreamSynchronize(e->stream));
    memset(out, 0, sizeof(*out));
    for (size_t i = 0; i < e->samples; i++) {
        if (e->close_of[i] < 0)
            continue;  // the run ended with this sample's tree phase still open (never: the last iteration closes it)
        float b = 0, c = 0;
        AZH_HIP(hipEventElapsedTime(&b, e->events[3 * i], e->events[3 * i + 1]));
        AZH_HIP(hipEventElapsedTime(&c, e->events[3 * i + 1], e->events[(size_t)e->close_of[i]]));
        out->net_ms += b;
        out->select_ms += c;  // the fused tree launch (backup + mark + select), the compaction and the gaps around them
        out->iterations += 1;
    }
    return 0;
}

// Takes the finished-game records off the device: waits for the work enqueued so far, copies the record ring to the host
// and empties it.  azh_engine_drain_json formats what was fetched without touching the device, so a caller may start its
// next run between the two calls and the formatting (a quarter of a millisecond per 150-ply game) and its own file
// writes happen under that run instead of in front of it.  Calling azh_engine_fetch is optional: a drain with nothing
// fetched fetches — but never inside the drain sequence that follows an explicit fetch (up to and including the first
// drain call that hands out no game): there "nothing staged" means "no game finished", and a fetch would wait for the run
// the caller has enqueued in between (round 3's loop did exactly that and ran sequentially without saying so).
static int fetch_records(azh_engine *e)
{
    // Everything on the engine's OWN streams: a fetch of one half-batch must not wait for the other half's run (the
    // legacy null stream would: the engines' streams are blocking streams).
    AZH_HIP(hipStreamSynchronize(e->stream));
    AZH_HIP(hipStreamSynchronize(e->stream2));  // (idle by now: the last tree launch of a run waits for the last re-roots)
    e->unfetched_work = false;
    AZH_HIP(hipMemcpyAsync(e->h_head, e->P.ring_head, 8, hipMemcpyDeviceToHost, e->stream));
    AZH_HIP(hipStreamSynchronize(e->stream));
    u64 head = *e->h_head;
    if (head > e->P.ring_cap_words) {
        // A record or a drop marker did not fit (AZH_STAT_RING_OVERFLOW counts them): its uid will never come, and
        // uid order would wait for it for ever.  From here on the order is given up instead of the games: what is
        // held is handed out, and later games are handed out as they arrive.
        head = e->P.ring_cap_words;
        e->order_broken = true;
    }
    if (head > 0) {
        if ((size_t)head > e->h_stage_words) {
            // (grows to the largest fetch seen, in powers of two from 4 MiB: a round's records, not the ring's capacity)
            size_t want = (size_t)1 << 20;
            while (want < (size_t)head)
                want <<= 1;
            // (the outgrown buffer is kept until the engine is destroyed: freeing pinned memory waits for the whole device,
            // i.e. for the other half-batch's run)
            if (e->h_stage)
                e->h_stage_retired.push_back(e->h_stage);
            e->h_stage = nullptr;
            e->h_stage_words = 0;
            AZH_HIP(hipHostMalloc((void **)&e->h_stage, want * 4));
            e->h_stage_words = want;
        }
        AZH_HIP(hipMemcpyAsync(e->h_stage, e->P.ring, (size_t)head * 4, hipMemcpyDeviceToHost, e->stream));
        AZH_HIP(hipMemsetAsync(e->P.ring, 0, (size_t)head * 4, e->stream));
        AZH_HIP(hipMemsetAsync(e->P.ring_head, 0, 8, e->stream));
        AZH_HIP(hipStreamSynchronize(e->stream));
        e->staged.insert(e->staged.end(), e->h_stage, e->h_stage + (size_t)head);
    }
    return 0;
}

// 1 while work enqueued on the engine is still in flight, 0 when it is idle: a query, never a wait.  (The main stream is
// the one asked: everything a run enqueues is on it — in side-stream mode the last tree launch of a run waits for the last
// re-roots — so an idle main stream means an idle engine.)
extern "C" int azh_engine_query(azh_engine *e)
{
    if (!e)
        return azh_fail(-1, "azh_engine_query: null engine");
    const hipError_t rc = hipStreamQuery(e->stream);
    if (rc == hipErrorNotReady)
        return 1;
    AZH_HIP(rc);
    return 0;
}

extern "C" int azh_engine_fetch(azh_engine *e)
{
    if (!e)
        return azh_fail(-1, "azh_engine_fetch: null engine");
    const int rc = fetch_records(e);
    if (rc == 0)
        e->fetch_covers_drain = true;
    return rc;
}

extern "C" long long azh_engine_implicit_fetches(const azh_engine *e) { return e ? e->implicit_fetches : -1; }

// the fetched records -> lines (in uid order where that is asked for); host work only
static void format_staged(azh_engine *e)
{
    std::vector<uint32_t> host;
    host.swap(e->staged);
    std::vector<std::pair<uint32_t, size_t>> order;  // (uid, offset)
    size_t pos = 0;
    // (two fetches may sit behind each other in the staging buffer: a fetch that ended in a record cut off by a full
    // ring is followed by the next one's first header, which the scan finds again at the next magic word)
    while (pos + 8 <= host.size()) {
        if (host[pos] != RING_MAGIC) {
            pos++;
            continue;
        }
        // (a payload word may equal the magic — board halves are arbitrary bit patterns: a header counts only if the
        // record it announces is whole and its ply structure walks to its end exactly)
        if (!azh_record_well_formed(host.data() + pos, host.size() - pos, (uint32_t)e->P.max_plies, nullptr)) {
            pos++;
            continue;
        }
        const size_t words = host[pos + 5];
        order.emplace_back(host[pos + 2], pos);
        pos += words;
    }
    std::sort(order.begin(), order.end());
    const bool ids = (e->P.flags & AZH_FLAG_TWO_NETS) != 0;
    for (auto &o : order) {
        const uint32_t *rec = host.data() + o.second;
        const bool dropped = rec[7] == 1;
        std::string line = dropped ? std::string() : azh_format_game_json(rec, rec[5], ids);
        if (rec[7] == 2 && !ids)
            line.clear();  // a game that began at a loaded position: record drained and formatted like any other
                           // (the measured path does the same work per finished ply), but it is not a whole game.
                           // (The arena hands such games out: a match from given openings — uai_ringmaster.py:185-196 —
                           // loads the position after the opening; the caller knows the opening moves.)
        if (e->emit_by_uid && !e->order_broken)
            e->held[o.first] = std::move(line);
        else if (!line.empty())
            e->pending.push_back(std::move(line));
    }
    // (belt and braces: the longest possible game, 400 plies, outlives a few generations of short ones in the
    // other slots — not sixteen)
    if (e->held.size() > 16 * (size_t)e->P.G + 64)
        e->order_broken = true;
    if (e->emit_by_uid) {
        for (auto it = e->held.begin(); it != e->held.end() && (e->order_broken || it->first == e->next_uid);
             it = e->held.erase(it)) {
            if (!it->second.empty())
                e->pending.push_back(std::move(it->second));
            e->next_uid = it->first + 1;
        }
    }
}

extern "C" int azh_engine_drain_json(azh_engine *e, char *buf, int64_t cap, int64_t *used, int32_t *n_games)
{
    if (!e || !buf || !used || !n_games)
        return azh_fail(-1, "azh_engine_drain_json: null argument");
    *used = 0;
    *n_games = 0;
    if (e->pending_pos >= e->pending.size()) {
        e->pending.clear();
        e->pending_pos = 0;
        if (e->staged.empty() && !e->fetch_covers_drain && e->unfetched_work) {
            // a caller that never fetches: the drain does (and waits for whatever is enqueued) — once per round: the call
            // that ends a drain loop finds no work enqueued since this fetch and returns empty without another one, and
            // a caller that drains with ONE call per round gets a fetch in every round
            e->implicit_fetches++;
            const int rc = fetch_records(e);
            if (rc)
                return rc;
        }
        if (!e->staged.empty())
            format_staged(e);
        if (e->pending.empty())
            e->fetch_covers_drain = false;  // the sequence the explicit fetch covered ends with this (empty) call
    }
    while (e->pending_pos < e->pending.size()) {
        const std::string &line = e->pending[e->pending_pos];
        if (*used + (int64_t)line.size() + 1 > cap)
            break;
        memcpy(buf + *used, line.data(), line.size());
        buf[*used + line.size()] = '\n';
        *used += (int64_t)line.size() + 1;
        *n_games += 1;
        e->pending_pos++;
    }
    if (*n_games == 0 && e->pending_pos < e->pending.size())
        return azh_fail(-6, "azh_engine_drain_json: buffer of %lld bytes cannot hold one game line (%zu bytes)",
                        (long long)cap, e->pending[e->pending_pos].size() + 1);
    return 0;
}
