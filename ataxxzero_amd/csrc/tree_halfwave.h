// The tree phase with TWO games per wave, 32 lanes each (included by engine.hip inside namespace azh, after the
// one-wave-per-game functions whose arithmetic it repeats).
//
// Why: a CU holds 32 waves, the chip 8192; beyond that many games the one-wave-per-game launch runs in rounds, and the
// launch lasts as long as the deepest descent of its last round (16384 games: waves of the second round start 47 us late
// on average, profiles/round3_tree_stamps.txt).  With 32 lanes per game 16384 descents fit the wave slots at once.
//
// Same functions, same reference lines (select_game / backup_game / mark_game above), same results bit for bit:
//   * every per-game value is an ordinary per-lane value that is equal across the game's 32 lanes; the two games of a
//     wave take their own paths under the EXEC mask; cross-lane traffic never leaves a game (azh_device.h, namespace hw);
//   * element j of a node's move list lives in lane j % 32; where the engine/oracle contract fixes the order of an f32
//     sum (softmax denominator, Dirichlet total) the lane keeps two partial sums — elements with (j / 32) even and odd,
//     i.e. virtual lanes l and l + 32 of the 64-lane order — and hw::sum_f32 combines them exactly as the 64-lane
//     butterfly does.  The oracle does not change.
//   * the two games of a wave run in lockstep, and a branch only one of them takes is time the other one waits: whatever
//     waits for memory is therefore written as ONE instruction stream for both — every level of a descent loads up to four
//     records per lane (128 children) under lane predicates instead of choosing between a short and a general path, and the
//     backup requests the leaf's moves, the logits and the path's edges for both games together — so that a tree level
//     costs one memory round trip per WAVE.  (The first version kept select_game's branches: a level cost 1.4 us against
//     0.97 with 4096 games and the launch was slower than one wave per game, profiles/round4_halfwave_first_version_*.txt.)
// Not provided on 32 lanes (the engine keeps one wave per game for them): the arena's flags (AZH_FLAG_PY_POSTERIOR's
// 833-way softmax, two nets) — arena batches are a thousand games, far from the 8192 where this matters.

// Node holding the evaluation of board (w0, w1), or NONE: the same 64 probe slots as tt_lookup, two per lane.
__device__ inline u32 tt_lookup_h(const u32 *tt, u32 mask, const Arena &A, u64 w0, u64 w1)
{
    const int l = hw::lane();
    const u32 h = tt_hash(w0, w1, mask);
    const u32 id0 = tt[(h + (u32)l) & mask], id1 = tt[(h + (u32)l + 32u) & mask];
    const u64 empties = (u64)hw::ballot(id0 == NONE) | ((u64)hw::ballot(id1 == NONE) << 32);
    const u64 before = empties ? (empties & (0ULL - empties)) - 1ULL : ~0ULL;  // slots ahead of the first empty one
    bool m0 = false, m1 = false;
    if (((before >> l) & 1ULL) && id0 != NONE) {
        const ulonglong2 b = A.nb[id0];
        m0 = b.x == w0 && b.y == w1;
    }
    if (((before >> (l + 32)) & 1ULL) && id1 != NONE) {
        const ulonglong2 b = A.nb[id1];
        m1 = b.x == w0 && b.y == w1;
    }
    const u64 hits = (u64)hw::ballot(m0) | ((u64)hw::ballot(m1) << 32);
    if (!hits)
        return NONE;
    const int first = __ffsll((long long)hits) - 1;
    return (u32)hw::read_lane((int)(first < 32 ? id0 : id1), first & 31);
}

// select_game on 32 lanes.  `s`: the game's state after backup / mark (equal in the game's lanes).
//
// Written for a small register footprint — two games share a wave, so what select_game keeps in scalar registers is
// per-lane data here, and 16384 resident games need 8 waves per SIMD, i.e. at most 64 vector registers: the state is
// stored at once (what backup and mark changed) and only the fields select itself changes are stored again at the end;
// the node and edge counts are re-read at the expansion, beside the loads the expansion waits for anyway; of the arena
// only the edge base is kept as a pointer; the leaf board is stored where it is computed.
// The fields of azh_game_state the tree phase reads (the node / edge counts are read where an expansion needs them, ply
// and uid where a root's noise is drawn): six registers per lane instead of ten.
struct HState {
    int phase, arena, root_visits, leaf_kind, leaf_node, path_len;
};

__device__ inline HState load_hstate(const azh_game_state *gs)
{
    HState h;
    h.phase = gs->phase;
    h.arena = gs->arena;
    h.root_visits = gs->root_visits;
    h.leaf_kind = gs->leaf_kind;
    h.leaf_node = gs->leaf_node;
    h.path_len = gs->path_len;
    return h;
}

template <bool STAMP = false>
__device__ inline int select_game_h(const EngineParams &P, int g, const HState &s, u16 *s_moves, u64 *st = nullptr)
{
    constexpr int HL = 32;
    const int l = hw::lane();
    const u32 slot = (u32)s.arena * (u32)P.G + (u32)g;
    uint4 *const ed = P.edge + (size_t)slot * P.edge_cap;
    azh_game_state *const gs = P.gs + g;
    if (l == 0) {   // what backup and mark may have changed (select's own fields follow at the end)
        gs->phase = s.phase;
        gs->root_visits = s.root_visits;
    }

    int kind = AZH_LEAF_NONE, leaf_node = 0, depth = 0;
    // bit 0: an MCTS step begins, 1: evaluation found in the tree, 2: edge arena overflow
    u32 flags = 0, st_levels = 0, st_children = 0, st_newmoves = 0;
    u32 node = 0, sel_eidx = 0;
    bool expand = false;
    int path_base = 0;
    u32 path_buf0 = 0, path_buf1 = 0;
    auto path_of = [&]() { return P.path + (size_t)g * P.path_cap; };

    if (s.phase >= 2) {
        kind = AZH_LEAF_NONE;  // the move is due (its re-root runs after this select) or the slot is idle
        if (l == 0)
            P.leaf_board[g] = make_ulonglong2(0ull, 0ull);
    } else if (s.phase == 0) {
        kind = AZH_LEAF_ROOT;  // the root's priors are (re)computed with noise (:380-383, :485-490)
        const ulonglong2 w = P.node_board[(size_t)slot * P.node_cap];
        const Board b = unpack_board(w.x, w.y);
        if (l == 0)
            P.leaf_board[g] = b.turn ? make_ulonglong2(b.o, b.x) : make_ulonglong2(b.x, b.o);
    } else {
        const bool resume = s.leaf_kind == AZH_LEAF_DESCENT;
        flags = resume ? 0u : 1u;
        node = resume ? (u32)s.leaf_node : 0u;
        depth = resume ? s.path_len : 0;
        path_base = depth;
        u32 kid;
        {
            const uint4 rinfo = P.node_info[(size_t)slot * P.node_cap + node];
            kid = pack_kid(rinfo.x, rinfo.y & 0xFFFFu, (rinfo.y >> 16) != 0u);
        }
        int levels_done = 0;
        bool have_n = !resume;
        u32 n_node = (u32)s.root_visits;
        // the path entries of this launch: lane k keeps entries k and k + 32, stored together afterwards (every 64 levels
        // when there is no budget)
        auto push_path = [&](u32 eidx) {
            const int k = depth - path_base;
            if (l == (k & 31)) {
                if (k < HL) path_buf0 = eidx;
                else path_buf1 = eidx;
            }
            depth++;
            if (depth - path_base == 2 * HL) {
                int *path = path_of();
                path[path_base + l] = (int)path_buf0;
                path[path_base + HL + l] = (int)path_buf1;
                path_base = depth;
            }
        };
        // the records of children base + l + 32 k (k = 0..3) of the node whose range is `k`: up to 128 children in one go
        uint4 ev[4];
        auto load_level = [&](u32 range, int base) {
            const int cnt = kid_count(range);
            const uint4 *f = ed + kid_first(range) + (u32)(base + l);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                ev[k] = fresh_edge(0u);
                if (base + l + HL * k < cnt)
                    ev[k] = f[HL * k];
            }
        };
        if (!kid_finished(kid))
            load_level(kid, 0);
        for (;;) {
            if (P.select_budget != 0 && levels_done == P.select_budget) {
                kind = AZH_LEAF_DESCENT;  // park: no leaf for the evaluator from this game this iteration
                leaf_node = (int)node;
                break;
            }
            levels_done++;
            const int M = kid_count(kid);
            const u32 first = kid_first(kid);
            if (kid_finished(kid) || M == 0) {
                kind = AZH_LEAF_TERMINAL;  // select_action -> NO_MOVE (:336-340)
                leaf_node = (int)node;
                break;
            }
            st_levels += 1;
            st_children += (u32)M;
            // N of the node (see select_game): known from the edge that led here, except on a resumed descent
            u32 ntot = n_node;
            if (!have_n) {
                u32 nsum = 0;
                if (M <= 4 * HL) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        nsum += edge_visits(ev[k]);   // (records that were not loaded hold 0 visits)
                } else {
                    for (int j = l; j < M; j += HL)
                        nsum += reinterpret_cast<const u32 *>(ed + first + (u32)j)[2] & 0xFFFFu;
                }
                ntot = hw::sum_u32(nsum);
            }
            const float sq = sqrtf((float)(1u + ntot));
            // arg-max of the score bits (scores are >= 0, their bit patterns order like the floats), ties to the LAST maximal
            // edge (:354) or the first (the arena's python max(), engine.py:291); NaN scores and empty lanes never win
            u32 best_bits = 0, best_z = 0, best_w = 0;
            int best_j = -1;
            for (int base = 0;;) {
                u32 bits[4];
                bool valid[4];
                u32 top = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float prior = u2f(ev[k].x);
                    const u32 n = edge_visits(ev[k]);
                    const float W = u2f(ev[k].y);
                    const float q = n ? W / (float)n : 0.0f;
                    const float u = (sq / (1.0f + (float)n)) * (P.c_puct * prior);
                    const float score = u + q;
                    valid[k] = base + l + HL * k < M && score >= 0.0f;
                    bits[k] = valid[k] ? f2u(score + 0.0f) : 0u;
                    top = bits[k] > top ? bits[k] : top;
                }
                top = hw::max_u32(top);
                // who holds `top` in this chunk: one ballot per row of records; the rows are visited so that the LAST hit is
                // the tie rule's choice (highest index for the last-max rule, lowest for the first-max rule)
                const bool tie_first = (P.flags & AZH_FLAG_TIE_FIRST) != 0;
                u32 cand[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    cand[k] = hw::ballot(valid[k] && bits[k] == top);
                int ck = -1, cl = 0;
                if (!tie_first) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (cand[k]) {
                            ck = k;
                            cl = 31 - __clz((int)cand[k]);
                        }
                } else {
#pragma unroll
                    for (int k = 3; k >= 0; k--)
                        if (cand[k]) {
                            ck = k;
                            cl = __ffs((int)cand[k]) - 1;
                        }
                }
                // across chunks (more than 128 moves): a later chunk wins ties under the last-max rule, loses them under the first
                if (ck >= 0 && (best_j < 0 || top > best_bits || (top == best_bits && !tie_first))) {
                    const u32 vz = ck == 0 ? ev[0].z : (ck == 1 ? ev[1].z : (ck == 2 ? ev[2].z : ev[3].z));
                    const u32 vw = ck == 0 ? ev[0].w : (ck == 1 ? ev[1].w : (ck == 2 ? ev[2].w : ev[3].w));
                    best_bits = top;
                    best_j = base + HL * ck + cl;
                    best_z = (u32)hw::read_lane((int)vz, cl);
                    best_w = (u32)hw::read_lane((int)vw, cl);
                }
                base += 4 * HL;
                if (base >= M)
                    break;
                load_level(kid, base);   // (rare: a second round trip for the children beyond 128)
            }
            u32 zsel = best_z, wsel = best_w;
            int bj = best_j;
            if (bj < 0) {   // no valid score anywhere (NaN priors): edge 0, as select_game's key 0 does
                bj = 0;
                const uint4 e = ed[first];
                zsel = e.z;
                wsel = e.w;
            }
            const u32 eidx = first + (u32)bj;
            push_path(eidx);
            if ((zsel >> 16) == ENONE) {
                sel_eidx = eidx;
                expand = true;
                break;
            }
            node = zsel >> 16;
            kid = wsel;
            n_node = (zsel & 0xFFFFu) - 1u;
            have_n = true;
            // the next level's records, requested at once (none for a finished position, none when the budget parks the descent)
            if (!kid_finished(kid) && !(P.select_budget != 0 && levels_done == P.select_budget))
                load_level(kid, 0);
        }
    }
    if (expand) {
        // expand (:429-439)
        if constexpr (STAMP) st[4] = tree_stamp();
        const u32 eidx = sel_eidx;
        const size_t nbase = (size_t)slot * P.node_cap, ebase = (size_t)slot * P.edge_cap;
        const u32 mv = P.edge_move[ebase + eidx];
        const ulonglong2 pw = P.node_board[nbase + node];
        const int2 counts = *reinterpret_cast<const int2 *>(&gs->n_nodes);   // (n_nodes, n_edges): not kept over the descent
        const Board cb = make_move(unpack_board(pw.x, pw.y), (int)(mv & 0xFF), (int)(mv >> 8));
        int res2;
        const int M2 = hw::movegen(cb, P.blockers, s_moves, &res2);
        hw::sync();
        if (counts.x >= P.node_cap || (res2 == 0 && (counts.y + M2 > P.edge_cap || M2 > 255))) {
            flags |= 4u;
            kind = AZH_LEAF_NONE;
            leaf_node = 0;
            depth = 0;
            path_base = 0;
            if (l == 0)
                P.leaf_board[g] = make_ulonglong2(0ull, 0ull);
        } else {
            const u32 cid = (u32)counts.x;
            const u32 nf = (u32)counts.y;
            u32 known = NONE;
            if ((P.flags & AZH_FLAG_EVAL_CACHE) && res2 == 0) {
                Arena A = arena_of(P, s.arena, g);
                known = tt_lookup_h(tt_of(P, s.arena, g), (u32)P.tt_size - 1u, A, pack_word0(cb), cb.o);
            }
            int added = 0;
            if (res2 != 0) {
                float tv = res2 == 1 ? 1.0f : -1.0f;
                if (cb.turn == 1)
                    tv = -tv;
                if (l == 0)
                    P.node_info[nbase + cid] = make_uint4(0u, (u32)res2 << 16, 0u, f2u(tv));
                kind = AZH_LEAF_TERMINAL;
            } else if (known != NONE) {
                // same position, same moves in the same order: take the priors and the value, no evaluation
                const uint4 kinfo = P.node_info[nbase + known];
                for (int j = l; j < M2; j += HL) {
                    ed[nf + j] = fresh_edge(ed[kinfo.x + j].x);
                    P.edge_move[ebase + nf + j] = s_moves[j];
                }
                added = M2;
                if (l == 0)
                    P.node_info[nbase + cid] = make_uint4(nf, (u32)M2, 0u, kinfo.w);
                kind = AZH_LEAF_TERMINAL;  // "value known": backed up from node_info.w like a finished position
                flags |= 2u;
                st_newmoves = (u32)M2;
            } else {
                for (int j = l; j < M2; j += HL) {
                    ed[nf + j] = fresh_edge(0u);
                    P.edge_move[ebase + nf + j] = s_moves[j];
                }
                added = M2;
                if (l == 0)
                    P.node_info[nbase + cid] = make_uint4(nf, (u32)M2, 0u, 0u);
                kind = AZH_LEAF_EVAL;
                st_newmoves = (u32)M2;
            }
            if (l == 0) {
                P.node_board[nbase + cid] = make_ulonglong2(pack_word0(cb), cb.o);
                // the edge gets its child and the child's range
                reinterpret_cast<uint2 *>(ed + eidx)[1] =
                    make_uint2(cid << 16, res2 != 0 ? pack_kid(0u, 0u, 1u) : pack_kid(nf, (u32)M2, 0u));
                *reinterpret_cast<int2 *>(&gs->n_nodes) = make_int2(counts.x + 1, counts.y + added);
                P.leaf_board[g] = cb.turn ? make_ulonglong2(cb.o, cb.x) : make_ulonglong2(cb.x, cb.o);
            }
            leaf_node = (int)cid;
        }
    } else if (s.phase == 1 && l == 0) {
        P.leaf_board[g] = make_ulonglong2(0ull, 0ull);   // parked, or ended at a finished position
    }
    if (s.phase == 1) {
        // the path entries of this launch (none after an overflow: depth is 0 then)
        const int left = depth - path_base;
        if (l < left)
            path_of()[path_base + l] = (int)path_buf0;
        if (l + HL < left)
            path_of()[path_base + HL + l] = (int)path_buf1;
    }

    if constexpr (STAMP) {
        st[5] = tree_stamp();
        if (st[4] == 0)
            st[4] = st[5];
        st[8] = st_levels;
        st[9] = st_children;
    }
    const int need = (kind == AZH_LEAF_EVAL || kind == AZH_LEAF_ROOT) ? 1 : 0;
    if (l == 0) {
        gs->leaf_kind = kind;
        gs->leaf_node = leaf_node;
        gs->path_len = depth;
        P.need_eval[g] = need;
        if (flags & 4u)
            P.force[g] = 1;
    }
    {   // counters: lane k owns counter k
        const u32 inc = l == AZH_STAT_STEPS ? (flags & 1u)
                      : l == AZH_STAT_NN_EVALS ? (u32)need
                      : l == AZH_STAT_LEVELS ? st_levels
                      : l == AZH_STAT_CHILDREN ? st_children
                      : l == AZH_STAT_NEW_MOVES ? st_newmoves
                      : l == AZH_STAT_EDGE_OVERFLOW ? (flags >> 2) & 1u
                      : l == AZH_STAT_CACHE_HITS ? (flags >> 1) & 1u
                      : l == AZH_STAT_PARKED ? (u32)(kind == AZH_LEAF_DESCENT) : 0u;
        if (l < NSTAT)
            add_stat(P, g, l, (u64)inc);
    }
    return need;
}

// backup_game on 32 lanes (cpp/self_play_client.cpp:204-271 priors + noise, :449-459 backup).  `scratch`: 256 floats of LDS
// of this game's own (the root's gamma draws wait there for their total: eight interleaved Philox chains in registers cost
// the whole kernel a wave per SIMD).
//
// One instruction stream for both games of the wave, whatever their leaves are: three memory round trips — (1) the leaf's
// header, the net's value, the path; (2) the leaf's moves and the path's edges; (3) the logits, while the path update is
// stored — under lane predicates, instead of a branch per kind of leaf that the other game would sit out.
__device__ inline void backup_game_h(const EngineParams &P, int g, HState &s, float *scratch)
{
    constexpr int HL = 32, R = MAX_MOVES / HL;  // 8 elements per lane; element j = l + 32 r: virtual lane l + 32 (r & 1)
    const int l = hw::lane();
    const int kind = s.leaf_kind;
    if (kind == AZH_LEAF_NONE || kind == AZH_LEAF_DESCENT)
        return;
    const u32 slot = (u32)s.arena * (u32)P.G + (u32)g;
    uint4 *const ed = P.edge + (size_t)slot * P.edge_cap;
    uint4 *const ni = P.node_info + (size_t)slot * P.node_cap;
    const bool priors = kind == AZH_LEAF_EVAL || kind == AZH_LEAF_ROOT;
    const bool walk = kind == AZH_LEAF_EVAL || kind == AZH_LEAF_TERMINAL;
    const bool remember = (P.flags & AZH_FLAG_EVAL_CACHE) && kind == AZH_LEAF_EVAL && l == 0;

    // (1)
    const uint4 info = ni[s.leaf_node];
    const float net_value = P.values[g];
    const int *path = P.path + (size_t)g * P.path_cap;
    const int plen = walk ? s.path_len : 0;
    int pe0 = -1, pe1 = -1;
    if (l < plen)
        pe0 = path[l];
    if (l + HL < plen)
        pe1 = path[l + HL];
    ulonglong2 leaf_b = make_ulonglong2(0ull, 0ull);
    if (remember)
        leaf_b = P.node_board[(size_t)slot * P.node_cap + s.leaf_node];

    // (2)
    const u32 first = info.x;
    const int M = priors ? (int)(info.y & 0xFFFFu) : 0;
    const u16 *em = P.edge_move + (size_t)slot * P.edge_cap + first;
    u32 mv[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int j = l + HL * r;
        mv[r] = 0;
        if (j < M)
            mv[r] = em[j];
    }
    uint2 q0 = make_uint2(0u, 0u), q1 = make_uint2(0u, 0u);   // (total score, visits | child) of this lane's path edges
    if (pe0 >= 0) {
        const u32 *e = reinterpret_cast<const u32 *>(ed + pe0);
        q0 = make_uint2(e[1], e[2]);
    }
    if (pe1 >= 0) {
        const u32 *e = reinterpret_cast<const u32 *>(ed + pe1);
        q1 = make_uint2(e[1], e[2]);
    }

    // (3)
    const float *row = P.logits + (size_t)g * AZH_POLICY_SIZE;
    float ex[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int j = l + HL * r;
        ex[r] = -INFINITY;
        if (j < M)
            ex[r] = row[policy_index(mv[r])];
    }
    // step() part 4 (:449-459): flip the score at every edge on the way up
    const float v = kind == AZH_LEAF_EVAL ? net_value : u2f(info.w);
    const float sc0 = (v + 1.0f) * 0.5f;
    const float fa = 1.0f - sc0, fb = 1.0f - fa, fc = 1.0f - fb;
    if (pe0 >= 0) {
        const int flips = plen - l;
        const float val = flips == 1 ? fa : ((flips & 1) ? fc : fb);
        u32 *e = reinterpret_cast<u32 *>(ed + pe0);
        e[1] = f2u(u2f(q0.x) + val);
        e[2] = q0.y + 1u;  // visits: the low half of the word (<= 60000, never carries into the child id)
    }
    if (pe1 >= 0) {
        const int flips = plen - (l + HL);
        const float val = flips == 1 ? fa : ((flips & 1) ? fc : fb);
        u32 *e = reinterpret_cast<u32 *>(ed + pe1);
        e[1] = f2u(u2f(q1.x) + val);
        e[2] = q1.y + 1u;
    }
    for (int i = 2 * HL + l; i < plen; i += HL) {   // (paths longer than 64 edges: the rest, a round trip per 32)
        const int flips = plen - i;
        const float val = flips == 1 ? fa : ((flips & 1) ? fc : fb);
        u32 *e = reinterpret_cast<u32 *>(ed + path[i]);
        e[1] = f2u(u2f(e[1]) + val);
        e[2] += 1u;  // visits: the low half of the word (<= 60000, never carries into the child id)
    }
    if (remember) {
        // the leaf now carries an evaluation: remember its value and enter it in the table
        reinterpret_cast<u32 *>(ni + s.leaf_node)[3] = f2u(net_value);
        tt_insert(tt_of(P, s.arena, g), (u32)P.tt_size - 1u, leaf_b.x, leaf_b.y, (u32)s.leaf_node);
    }

    if (priors) {
        // Evaluations::populate (:204-271): softmax over the legal moves' logits, then the Dirichlet mix at the root
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < R; r++)
            if (l + HL * r < M && ex[r] > mx)
                mx = ex[r];
        mx = hw::max_f32(mx);
        float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int j = l + HL * r;
            const float lg = ex[r];
            ex[r] = 0.0f;
            if (j < M) {
                ex[r] = det_expf(lg - mx);
                if (r & 1) p1 = p1 + ex[r];
                else p0 = p0 + ex[r];
            }
        }
        const float S = hw::sum_f32(p0, p1);
#pragma unroll
        for (int r = 0; r < R; r++)
            ex[r] = S > 0.0f ? ex[r] / S : ex[r];
        if (kind == AZH_LEAF_ROOT && P.noise_w > 0.0f) {
            const u32 uid = P.gs[g].uid, ply = (u32)P.gs[g].ply;
            float g0 = 0.0f, g1 = 0.0f;
#pragma unroll 1
            for (int j = l; j < M; j += 2 * HL) {   // one draw at a time; this lane's virtual lanes in turn, rounds in order
                const float a = det_gamma(P.alpha, P.k0, P.k1, uid, ply, (u32)j);
                scratch[j] = a;
                g0 = g0 + a;
                if (j + HL < M) {
                    const float b = det_gamma(P.alpha, P.k0, P.k1, uid, ply, (u32)(j + HL));
                    scratch[j + HL] = b;
                    g1 = g1 + b;
                }
            }
            const float T = hw::sum_f32(g0, g1);
            const float w = P.noise_w, omw = 1.0f - w;
            if (T > 0.0f) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int j = l + HL * r;
                    const float d = (j < M ? scratch[j] : 0.0f) / T;   // (this lane's own stores)
                    const float t1 = w * d;
                    const float t2 = omw * ex[r];
                    ex[r] = t1 + t2;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int j = l + HL * r;
            if (j < M)
                reinterpret_cast<u32 *>(ed + first + j)[0] = f2u(ex[r]);
        }
    }
    if (walk && s.path_len > 0)
        s.root_visits += 1;
    if (kind == AZH_LEAF_ROOT)
        s.phase = 1;
    s.leaf_kind = AZH_LEAF_NONE;
}

__device__ inline void mark_game_h(const EngineParams &P, int g, HState &s, int forced)
{
    if (s.phase == 1 && s.leaf_kind != AZH_LEAF_DESCENT && (s.root_visits >= P.visits || forced != 0)) {
        s.phase = 2;
        if (hw::lane() == 0)
            P.adv_list[atomicAdd(P.adv_count, 1)] = g;
    }
}

// per game 1 KiB of LDS: the move list of the node being expanded (select) / the root's gamma draws (backup)
union HScratch {
    u16 moves[MAX_MOVES];
    float f[MAX_MOVES];
};

// the step-wise API's kernels: two games per 64-thread workgroup
__global__ __launch_bounds__(WAVE) void k_select_h(EngineParams P)
{
    __shared__ HScratch scr[2];
    const int g = 2 * (int)blockIdx.x + hw::half();
    if (g < P.G)
        select_game_h(P, g, load_hstate(P.gs + g), scr[hw::half()].moves);
}

__global__ __launch_bounds__(WAVE) void k_backup_h(EngineParams P)
{
    __shared__ HScratch scr[2];
    const int g = 2 * (int)blockIdx.x + hw::half();
    if (g < P.G) {
        HState s = load_hstate(P.gs + g);
        const int kind = s.leaf_kind;
        backup_game_h(P, g, s, scr[hw::half()].f);
        if (kind != AZH_LEAF_NONE && kind != AZH_LEAF_DESCENT && hw::lane() == 0) {
            azh_game_state *gs = P.gs + g;
            gs->phase = s.phase;
            gs->root_visits = s.root_visits;
            gs->leaf_kind = s.leaf_kind;
        }
    }
}

__global__ __launch_bounds__(WAVE) void k_mark_h(EngineParams P)
{
    const int g = 2 * (int)blockIdx.x + hw::half();
    if (g < P.G) {
        HState s = load_hstate(P.gs + g);
        const int phase = s.phase;
        mark_game_h(P, g, s, P.force[g]);
        if (s.phase != phase && hw::lane() == 0)
            P.gs[g].phase = s.phase;
    }
}

// k_tree with two games per wave: TREE_WAVES waves = 2 * TREE_WAVES games per workgroup; the leaf-list compaction by
// the last workgroup is k_tree's (compact_leaves), with one net.
// AZH_TREE_H_OCC: waves per SIMD the compiler must leave room for (register budget 512 / n per lane, the rest spills to
// scratch); 0 = no bound.  16384 games are 8192 waves: 8 per SIMD keep them all resident, 7 keep 14336 games.
#ifndef AZH_TREE_H_OCC
#define AZH_TREE_H_OCC 0
#endif
#if AZH_TREE_H_OCC
#define AZH_TREE_H_BOUNDS(threads) __launch_bounds__(threads, AZH_TREE_H_OCC)
#else
#define AZH_TREE_H_BOUNDS(threads) __launch_bounds__(threads)
#endif
template <bool STAMP, int TREE_WAVES>
__global__ AZH_TREE_H_BOUNDS(TREE_WAVES * WAVE) void k_tree_h(EngineParams P, int mode)
{
    constexpr int GAMES = 2 * TREE_WAVES;
    __shared__ HScratch scr[GAMES];
    __shared__ int s_cnt[2 * TREE_WAVES];
    __shared__ int s_need[GAMES];
    __shared__ int s_last;
    static_assert(32 % GAMES == 0, "a workgroup's need bits must lie in one word of the mask");
    const int slot = (int)(threadIdx.x >> 5);
    const int g = (int)blockIdx.x * GAMES + slot;
    u64 st[TREE_STAMPS] = {};
    int need = 0;
    if constexpr (STAMP) st[0] = tree_stamp();
    if (g < P.G) {
        HState s = load_hstate(P.gs + g);
        const int forced = P.force[g];
        if constexpr (STAMP) st[1] = tree_stamp();
        if (mode & 1) {
            backup_game_h(P, g, s, scr[slot].f);
            if constexpr (STAMP) st[2] = tree_stamp();
            mark_game_h(P, g, s, forced);
        }
        if constexpr (STAMP) {
            st[3] = tree_stamp();
            if (!(mode & 1)) st[2] = st[3];
        }
        if (mode & 2) {
            need = select_game_h<STAMP>(P, g, s, scr[slot].moves, st);
        } else if (hw::lane() == 0) {
            azh_game_state *gs = P.gs + g;
            gs->phase = s.phase;
            gs->root_visits = s.root_visits;
            gs->leaf_kind = s.leaf_kind;
        }
    }
    if (!(mode & 2))
        return;
    if (hw::lane() == 0)
        s_need[slot] = need;
    if constexpr (STAMP) st[6] = tree_stamp();
    __syncthreads();
    if constexpr (STAMP) {
        st[7] = tree_stamp();
        if (g < P.G && hw::lane() < TREE_STAMPS) {
            u64 v = st[0];
#pragma unroll
            for (int k = 1; k < TREE_STAMPS; k++)
                v = hw::lane() == k ? st[k] : v;
            P.stamps[(size_t)g * TREE_STAMPS + hw::lane()] = v;
        }
    }
    if (threadIdx.x == 0) {
        u32 m1 = 0;
#pragma unroll
        for (int k = 0; k < GAMES; k++)
            m1 |= (u32)(s_need[k] != 0) << k;
        const int g0 = (int)blockIdx.x * GAMES;
        u32 seen = 0;
        if (m1)
            seen |= __hip_atomic_fetch_or(&P.need_mask[g0 >> 5], m1 << (g0 & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" : : "v"(seen) : "memory");   // the OR has been performed before the ticket is drawn
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
        compact_leaves<TREE_WAVES>(P, 0, s_cnt);
}
