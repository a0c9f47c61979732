/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see mcts_oracle.h for scope and the
 * reference lines each step restates).
 *
 * One game at a time, plain loops.  Floating-point work is f32 in the exact
 * operation order the HIP engine uses (lane-striped partial sums followed by
 * an xor-butterfly over 64 lanes), so stored priors / scores / visit counts
 * can be compared bitwise.
 */
#include "mcts_oracle.h"

#include <stdlib.h>
#include <string.h>

#include "ataxx_rules_oracle.h"
#include "detmath.h"

#define NONE 0xFFFFFFFFu
#define REC_HDR 32
#define REC_MAXD 256
#define REC_STRIDE (REC_HDR + 4 * REC_MAXD)
#define TURN_BIT (1ULL << 63)

typedef struct blob {
    struct blob *next;
    int64_t size;
    uint8_t data[];
} blob;

struct orc_engine {
    orc_config cfg;
    int G, node_cap, edge_cap, path_cap;
    orc_game_state *gs;
    int32_t *force;
    int32_t *path;
    uint64_t *node_board;
    uint32_t *node_info;
    uint32_t *edge;
    uint16_t *edge_move;
    uint8_t *rec;
    uint64_t *stats;
    blob *done_head, *done_tail;
    int done_count;
    int32_t *no_emit;  /* [G] start ply + 1 while the slot's game is one started by orc_engine_set_positions (its record is partial), else 0 */
    uint32_t uid_limit; /* orc_engine_set_game_limit: games with uid >= this are not started (0 = no limit) */
    uint32_t *tt;      /* ORC_FLAG_EVAL_CACHE: [2][G][tt_size] open-addressed table of evaluated nodes, keyed by the board */
    int tt_size;
};

static inline uint64_t *NB(const orc_engine *e, int a, int g) { return e->node_board + ((size_t)(a * e->G + g) * e->node_cap) * 2; }
static inline uint32_t *NI(const orc_engine *e, int a, int g) { return e->node_info + ((size_t)(a * e->G + g) * e->node_cap) * 4; }
static inline uint32_t *ED(const orc_engine *e, int a, int g) { return e->edge + ((size_t)(a * e->G + g) * e->edge_cap) * 4; }
static inline uint16_t *EM(const orc_engine *e, int a, int g) { return e->edge_move + (size_t)(a * e->G + g) * e->edge_cap; }
static inline uint8_t *REC(const orc_engine *e, int g, int ply) { return e->rec + ((size_t)g * e->cfg.max_plies + ply) * REC_STRIDE; }

static void unpack(const orc_engine *e, const uint64_t *b, orc_pos *p)
{
    p->pieces[0] = b[0] & ~TURN_BIT;
    p->pieces[1] = b[1];
    p->blockers = e->cfg.blockers;
    p->turn = (int)(b[0] >> 63);
    p->ply = 0;
}

static void pack(const orc_pos *p, uint64_t *b)
{
    b[0] = p->pieces[0] | ((uint64_t)p->turn << 63);
    b[1] = p->pieces[1];
}

/* ---- evaluation cache (ORC_FLAG_EVAL_CACHE): engine.py's NNEvaluator.cache (engine.py:127-234) per game.  A position
 * the game's search has already evaluated takes its priors and value from the node that carries them.  Same hash and
 * same open addressing as the HIP engine; the engine enters several nodes at once when it rebuilds the table after a
 * move, so two nodes with the same board may sit in a different order there — they carry the same evaluation. */
static inline uint32_t *TT(const orc_engine *e, int a, int g) { return e->tt + ((size_t)a * e->G + g) * (size_t)e->tt_size; }

static uint32_t tt_hash(uint64_t w0, uint64_t w1, uint32_t mask)
{
    uint64_t k = (w0 * 0x9E3779B97F4A7C15ULL) ^ ((w1 + 0x7F4A7C15ULL) * 0xC2B2AE3D27D4EB4FULL);
    return (uint32_t)(k >> 40) & mask;
}

static uint32_t tt_lookup(const orc_engine *e, int a, int g, uint64_t w0, uint64_t w1)
{
    const uint32_t *tt = TT(e, a, g);
    uint32_t mask = (uint32_t)e->tt_size - 1u;
    const uint64_t *nb = NB(e, a, g);
    uint32_t slot = tt_hash(w0, w1, mask);
    for (int tries = 0; tries < 64 && tt[slot] != NONE; tries++, slot = (slot + 1) & mask)
        if (nb[2 * (size_t)tt[slot]] == w0 && nb[2 * (size_t)tt[slot] + 1] == w1)
            return tt[slot];
    return NONE;
}

static void tt_insert(orc_engine *e, int a, int g, uint32_t id)
{
    uint32_t *tt = TT(e, a, g);
    uint32_t mask = (uint32_t)e->tt_size - 1u;
    const uint64_t *b = NB(e, a, g) + 2 * (size_t)id;
    uint32_t slot = tt_hash(b[0], b[1], mask);
    while (tt[slot] != NONE)
        slot = (slot + 1) & mask;
    tt[slot] = id;
}

static void tt_clear(orc_engine *e, int a, int g)
{
    if (e->cfg.flags & ORC_FLAG_EVAL_CACHE)
        memset(TT(e, a, g), 0xFF, sizeof(uint32_t) * (size_t)e->tt_size);
}

/* 64-lane emulation: sum v[0..n) as lane-striped partials + xor butterfly with offsets
 * 1, 2, 4, 8, 16, 32 (the order the HIP engine's DPP reduction produces). */
static float wave_sum(const float *v, int n)
{
    float lane[64];
    for (int l = 0; l < 64; l++) {
        float p = 0.0f;
        for (int j = l; j < n; j += 64)
            p = p + v[j];
        lane[l] = p;
    }
    for (int off = 1; off <= 32; off <<= 1) {
        float t[64];
        for (int l = 0; l < 64; l++)
            t[l] = lane[l] + lane[l ^ off];
        memcpy(lane, t, sizeof(lane));
    }
    return lane[0];
}

/* Create node `id` of game g (arena a) from position p: adjudicate, generate
 * edges.  Returns 0, or -1 on edge-arena overflow (nothing written). */
static int make_node(orc_engine *e, int g, int a, int id, const orc_pos *p, int *out_moves)
{
    orc_game_state *s = &e->gs[g];
    uint16_t moves[ORC_MAX_MOVES];
    int M = 0;
    int result = orc_result(p, moves, &M);
    uint32_t *info = NI(e, a, g) + 4 * (size_t)id;
    if (result != 0) {
        float tv = result == 1 ? 1.0f : -1.0f;
        if (p->turn == 1)
            tv = -tv;
        pack(p, NB(e, a, g) + 2 * (size_t)id);
        info[0] = 0;
        info[1] = (uint32_t)result << 16;
        info[2] = 0;
        info[3] = orc_f2u(tv);
        if (out_moves) *out_moves = 0;
        return 0;
    }
    if (s->n_edges + M > e->edge_cap)
        return -1;
    pack(p, NB(e, a, g) + 2 * (size_t)id);
    uint32_t first = (uint32_t)s->n_edges;
    uint32_t *ed = ED(e, a, g) + 4 * (size_t)first;
    uint16_t *em = EM(e, a, g) + first;
    for (int j = 0; j < M; j++) {
        ed[4 * j + 0] = 0;
        ed[4 * j + 1] = 0;
        ed[4 * j + 2] = 0;
        ed[4 * j + 3] = NONE;
        em[j] = moves[j];
    }
    s->n_edges += M;
    info[0] = first;
    info[1] = (uint32_t)M;
    info[2] = 0;
    info[3] = 0;
    if (out_moves) *out_moves = M;
    return 0;
}

static void init_game(orc_engine *e, int g, uint32_t uid)
{
    orc_game_state *s = &e->gs[g];
    if (e->uid_limit != 0 && uid >= e->uid_limit) {
        /* past the game limit (azh_engine_set_game_limit): no game, the slot goes idle; the arena stays as the last move left it */
        s->phase = ORC_PHASE_IDLE;
        s->uid = uid;
        s->leaf_kind = ORC_LEAF_NONE;
        s->path_len = 0;
        s->root_visits = 0;
        e->force[g] = 0;
        return;
    }
    memset(s, 0, sizeof(*s));
    s->uid = uid;
    s->phase = ORC_PHASE_ROOT_EVAL;
    e->force[g] = 0;
    e->no_emit[g] = 0;
    orc_pos p;
    p.pieces[0] = e->cfg.start_x;
    p.pieces[1] = e->cfg.start_o;
    p.blockers = e->cfg.blockers;
    p.turn = e->cfg.start_turn;
    p.ply = 0;
    make_node(e, g, 0, 0, &p, NULL);
    s->n_nodes = 1;
    tt_clear(e, 0, g);
}

orc_engine *orc_engine_create(const orc_config *cfg)
{
    orc_engine *e = (orc_engine *)calloc(1, sizeof(*e));
    e->cfg = *cfg;
    e->G = cfg->games;
    e->node_cap = cfg->visits + 8;
    e->edge_cap = e->node_cap * cfg->edges_per_node;
    e->path_cap = e->node_cap;
    size_t G = (size_t)e->G;
    e->gs = (orc_game_state *)calloc(G, sizeof(orc_game_state));
    e->force = (int32_t *)calloc(G, sizeof(int32_t));
    e->no_emit = (int32_t *)calloc(G, sizeof(int32_t));
    e->path = (int32_t *)calloc(G * e->path_cap, sizeof(int32_t));
    e->node_board = (uint64_t *)calloc(2 * G * e->node_cap * 2, sizeof(uint64_t));
    e->node_info = (uint32_t *)calloc(2 * G * e->node_cap * 4, sizeof(uint32_t));
    e->edge = (uint32_t *)calloc(2 * G * e->edge_cap * 4, sizeof(uint32_t));
    e->edge_move = (uint16_t *)calloc(2 * G * e->edge_cap, sizeof(uint16_t));
    e->rec = (uint8_t *)calloc(G * cfg->max_plies, REC_STRIDE);
    e->stats = (uint64_t *)calloc(G * ORC_STAT_COUNT, sizeof(uint64_t));
    e->tt_size = 1;
    while (e->tt_size < 4 * e->node_cap)
        e->tt_size <<= 1;
    if (cfg->flags & ORC_FLAG_EVAL_CACHE)
        e->tt = (uint32_t *)malloc(sizeof(uint32_t) * 2 * G * (size_t)e->tt_size);
    for (int g = 0; g < e->G; g++)
        init_game(e, g, (uint32_t)g);
    return e;
}

void orc_engine_destroy(orc_engine *e)
{
    if (!e) return;
    while (e->done_head) { blob *n = e->done_head->next; free(e->done_head); e->done_head = n; }
    free(e->gs); free(e->force); free(e->path); free(e->node_board); free(e->node_info);
    free(e->edge); free(e->edge_move); free(e->rec); free(e->stats); free(e->no_emit); free(e->tt); free(e);
}

int orc_engine_node_cap(const orc_engine *e) { return e->node_cap; }
void orc_engine_set_visits(orc_engine *e, int visits) { e->cfg.visits = visits; }
/* mirror of azh_engine_set_game_limit: uids 0 .. games - 1 are played; a slot whose (next) game is past that goes idle */
void orc_engine_set_game_limit(orc_engine *e, int64_t games)
{
    e->uid_limit = (uint32_t)games;
    for (int g = 0; g < e->G; g++) {
        orc_game_state *s = &e->gs[g];
        if (s->phase == ORC_PHASE_IDLE && s->uid < e->uid_limit)
            init_game(e, g, s->uid);   /* the limit was raised: the slot plays the game it was waiting with */
        else if (s->phase == ORC_PHASE_ROOT_EVAL && s->leaf_kind == ORC_LEAF_NONE && s->ply == 0 && s->n_nodes == 1 &&
                 s->root_visits == 0 && s->uid >= e->uid_limit)
            s->phase = ORC_PHASE_IDLE; /* a game that has not begun */
    }
}
/* mirror of azh_engine_set_positions: every slot restarts at boards[g] (x | turn << 63, o) / plies[g], fresh tree, uid = g */
void orc_engine_set_positions(orc_engine *e, const uint64_t *boards, const int32_t *plies)
{
    for (int g = 0; g < e->G; g++) {
        orc_game_state *s = &e->gs[g];
        memset(s, 0, sizeof(*s));
        s->uid = (uint32_t)g;
        s->phase = ORC_PHASE_ROOT_EVAL;
        s->ply = plies[g];
        e->force[g] = 0;
        e->no_emit[g] = plies[g] + 1;   /* start ply + 1 (the HIP engine's no_emit): the record begins there */
        orc_pos p;
        unpack(e, boards + 2 * (size_t)g, &p);
        make_node(e, g, 0, 0, &p, NULL);
        s->n_nodes = 1;
        tt_clear(e, 0, g);
    }
}

int orc_engine_edge_cap(const orc_engine *e) { return e->edge_cap; }

/* cpp/self_play_client.cpp:386-447: descend by PUCT, expand one node. */
static void select_game(orc_engine *e, int g)
{
    orc_game_state *s = &e->gs[g];
    uint64_t *st = e->stats + (size_t)g * ORC_STAT_COUNT;
    int a = s->arena;
    const int resume = s->leaf_kind == ORC_LEAF_DESCENT;
    if (!resume)
        s->path_len = 0;
    if (s->phase == ORC_PHASE_ADVANCING || s->phase == ORC_PHASE_IDLE) {
        s->leaf_kind = ORC_LEAF_NONE;
        s->leaf_node = 0;
        return; /* ADVANCING: orc_engine_select plays the move after the leaf pass; IDLE: nothing to search */
    }
    if (s->phase == ORC_PHASE_ROOT_EVAL) {
        if ((e->cfg.flags & ORC_FLAG_TWO_NETS) && (NI(e, a, g)[1] & 0xFFFFu) == 1) {
            /* arena: a single legal move is played without search (uai_ringmaster.py:114-116) */
            s->leaf_kind = ORC_LEAF_NONE;
            s->leaf_node = 0;
            s->phase = ORC_PHASE_SEARCH;
            e->force[g] = 1;
            return;
        }
        s->leaf_kind = ORC_LEAF_ROOT;
        s->leaf_node = 0;
        st[ORC_STAT_NN_EVALS]++;
        return;
    }
    uint32_t *ni = NI(e, a, g);
    uint32_t *ed = ED(e, a, g);
    int32_t *path = e->path + (size_t)g * e->path_cap;
    uint32_t node = resume ? (uint32_t)s->leaf_node : 0;
    int depth = resume ? s->path_len : 0;
    int levels_done = 0;
    if (!resume)
        st[ORC_STAT_STEPS]++;
    for (;;) {
        if (e->cfg.select_budget && levels_done == (int)e->cfg.select_budget) {
            /* park the descent at this node; the evaluator gets no leaf from this game this iteration */
            s->leaf_kind = ORC_LEAF_DESCENT;
            s->leaf_node = (int)node;
            s->path_len = depth;
            return;
        }
        levels_done++;
        uint32_t first = ni[4 * node + 0];
        int M = (int)(ni[4 * node + 1] & 0xFFFFu);
        int result = (int)(ni[4 * node + 1] >> 16);
        if (result != 0 || M == 0) {
            /* select_action -> NO_MOVE at a finished position (:336-340) */
            s->leaf_kind = ORC_LEAF_TERMINAL;
            s->leaf_node = (int)node;
            s->path_len = depth;
            return;
        }
        st[ORC_STAT_LEVELS]++;
        st[ORC_STAT_CHILDREN] += (uint64_t)M;
        uint32_t ntot = 0;
        for (int j = 0; j < M; j++)
            ntot += ed[4 * (first + j) + 1];
        /* total_action_score (:310-324) in f32 */
        float sq = sqrtf((float)(1u + ntot));
        float best = -INFINITY;
        int bj = -1;
        for (int j = 0; j < M; j++) {
            const uint32_t *x = ed + 4 * (size_t)(first + j);
            float P = orc_u2f(x[0]);
            uint32_t n = x[1];
            float W = orc_u2f(x[2]);
            float q = n ? W / (float)n : 0.0f;
            float u = (sq / (1.0f + (float)n)) * (e->cfg.c_puct * P);
            float score = u + q;
            /* ties: last maximal (:354); python's max() keeps the first (engine.py:291) */
            if (score > best || (score == best && j > bj && !(e->cfg.flags & ORC_FLAG_TIE_FIRST))) {
                best = score;
                bj = j;
            }
        }
        if (bj < 0)
            bj = 0;
        uint32_t eidx = first + (uint32_t)bj;
        path[depth++] = (int32_t)eidx;
        uint32_t child = ed[4 * (size_t)eidx + 3];
        if (child != NONE) {
            node = child;
            continue;
        }
        /* expand (:429-439) */
        uint16_t mv = EM(e, a, g)[eidx];
        orc_pos p;
        unpack(e, NB(e, a, g) + 2 * (size_t)node, &p);
        orc_makemove(&p, mv & 0xFF, mv >> 8);
        int M2 = 0;
        if (s->n_nodes >= e->node_cap || make_node(e, g, a, s->n_nodes, &p, &M2) != 0) {
            st[ORC_STAT_EDGE_OVERFLOW]++;
            e->force[g] = 1;
            s->leaf_kind = ORC_LEAF_NONE;
            s->leaf_node = 0;
            s->path_len = 0;
            return;
        }
        uint32_t cid = (uint32_t)s->n_nodes++;
        ed[4 * (size_t)eidx + 3] = cid;
        st[ORC_STAT_NEW_MOVES] += (uint64_t)M2;
        int result2 = (int)(ni[4 * cid + 1] >> 16);
        s->leaf_node = (int)cid;
        s->path_len = depth;
        uint32_t known = NONE;
        if ((e->cfg.flags & ORC_FLAG_EVAL_CACHE) && result2 == 0) {
            const uint64_t *cb = NB(e, a, g) + 2 * (size_t)cid;
            known = tt_lookup(e, a, g, cb[0], cb[1]);
        }
        if (result2 != 0) {
            s->leaf_kind = ORC_LEAF_TERMINAL;
        } else if (known != NONE) {
            /* this game's search has evaluated the position before: same moves in the same order, take the priors and
             * the value; backed up from node_info[3] like a finished position, no evaluation */
            uint32_t kf = ni[4 * known + 0], nf = ni[4 * cid + 0];
            for (int j = 0; j < M2; j++)
                ed[4 * (size_t)(nf + j) + 0] = ed[4 * (size_t)(kf + j) + 0];
            ni[4 * cid + 3] = ni[4 * known + 3];
            s->leaf_kind = ORC_LEAF_TERMINAL;
            st[ORC_STAT_CACHE_HITS]++;
        } else {
            s->leaf_kind = ORC_LEAF_EVAL;
            st[ORC_STAT_NN_EVALS]++;
        }
        return;
    }
}

static void advance_game(orc_engine *e, int g);

int orc_engine_select(orc_engine *e, int32_t *need_eval)
{
    int count = 0;
    for (int g = 0; g < e->G; g++) {
        select_game(e, g);
        int k = e->gs[g].leaf_kind;
        int need = (k == ORC_LEAF_EVAL || k == ORC_LEAF_ROOT);
        /* arena: the side to move alternates between net A (1) and net B (2); even slots give x to A */
        if (need && (e->cfg.flags & ORC_FLAG_TWO_NETS))
            need = 1 + ((e->gs[g].ply + g) & 1);
        if (need_eval) need_eval[g] = need;
        count += need != 0;
    }
    for (int g = 0; g < e->G; g++)
        if (e->gs[g].phase == ORC_PHASE_ADVANCING)
            advance_game(e, g);
    return count;
}

static void leaf_pos(const orc_engine *e, int g, orc_pos *p)
{
    const orc_game_state *s = &e->gs[g];
    unpack(e, NB(e, s->arena, g) + 2 * (size_t)s->leaf_node, p);
}

void orc_engine_leaf_boards(const orc_engine *e, uint64_t *out)
{
    for (int g = 0; g < e->G; g++) {
        orc_pos p;
        leaf_pos(e, g, &p);
        out[2 * g + 0] = p.pieces[p.turn];
        out[2 * g + 1] = p.pieces[!p.turn];
    }
}

void orc_engine_leaf_features(const orc_engine *e, int g, float *out196)
{
    orc_pos p;
    leaf_pos(e, g, &p);
    orc_features(&p, out196);
}

/* Evaluations::populate (:204-271): softmax over the legal moves' logits
 * (mathematically the reference's full softmax renormalised over legal moves),
 * optional Dirichlet mix at the root. */
static void apply_priors(orc_engine *e, int g, const float *logits, int root)
{
    orc_game_state *s = &e->gs[g];
    int a = s->arena;
    uint32_t *info = NI(e, a, g) + 4 * (size_t)s->leaf_node;
    uint32_t first = info[0];
    int M = (int)(info[1] & 0xFFFFu);
    uint32_t *ed = ED(e, a, g) + 4 * (size_t)first;
    const uint16_t *em = EM(e, a, g) + first;
    float l[ORC_MAX_MOVES], ex[ORC_MAX_MOVES];
    float P[ORC_MAX_MOVES];
    if (e->cfg.flags & ORC_FLAG_PY_POSTERIOR) {
        /* engine.py:197-203: softmax over all 833 logits, gather the legal moves, divide by
         * (their sum + 1e-6) */
        float all[833];
        float mx = -INFINITY;
        for (int i = 0; i < 833; i++)
            if (logits[i] > mx) mx = logits[i];
        for (int i = 0; i < 833; i++)
            all[i] = orc_det_expf(logits[i] - mx);
        float S = wave_sum(all, 833);
        for (int j = 0; j < M; j++)
            ex[j] = orc_det_expf(logits[orc_policy_index(em[j])] - mx) / S;
        float den = wave_sum(ex, M) + 1e-6f;
        for (int j = 0; j < M; j++)
            P[j] = ex[j] / den;
    } else {
        float mx = -INFINITY;
        for (int j = 0; j < M; j++) {
            l[j] = logits[orc_policy_index(em[j])];
            if (l[j] > mx) mx = l[j];
        }
        for (int j = 0; j < M; j++)
            ex[j] = orc_det_expf(l[j] - mx);
        float S = wave_sum(ex, M);
        for (int j = 0; j < M; j++)
            P[j] = S > 0.0f ? ex[j] / S : ex[j];
    }
    if (root && e->cfg.dirichlet_weight > 0.0f) {
        float gm[ORC_MAX_MOVES];
        uint32_t k0 = (uint32_t)e->cfg.seed, k1 = (uint32_t)(e->cfg.seed >> 32);
        for (int j = 0; j < M; j++)
            gm[j] = orc_det_gamma(e->cfg.dirichlet_alpha, k0, k1, s->uid, (uint32_t)s->ply, (uint32_t)j);
        float T = wave_sum(gm, M);
        float w = e->cfg.dirichlet_weight, omw = 1.0f - w;
        if (T > 0.0f) {
            for (int j = 0; j < M; j++) {
                float d = gm[j] / T;
                float t1 = w * d;
                float t2 = omw * P[j];
                P[j] = t1 + t2;
            }
        }
    }
    /* priors are >= 0 (and never NaN: orc_det_expf is 0 for a NaN argument); stored without a sign like the HIP engine's,
     * which keeps bit 31 for a mark of its own — the mask changes no value */
    for (int j = 0; j < M; j++)
        ed[4 * j + 0] = orc_f2u(P[j]) & 0x7FFFFFFFu;
}

/* step() part 4 (:449-459). */
static void backup_path(orc_engine *e, int g, float value)
{
    orc_game_state *s = &e->gs[g];
    uint32_t *ed = ED(e, s->arena, g);
    const int32_t *path = e->path + (size_t)g * e->path_cap;
    float sc = (value + 1.0f) * 0.5f;
    for (int i = s->path_len - 1; i >= 0; i--) {
        sc = 1.0f - sc;
        uint32_t *x = ed + 4 * (size_t)path[i];
        x[2] = orc_f2u(orc_u2f(x[2]) + sc);
        x[1] += 1;
    }
    if (s->path_len > 0)
        s->root_visits += 1;
}

/* ONE_RANDOM_MOVE (:515-518): the ply of the uniformly random move, uniform on 0..119, a pure
 * function of (seed, game uid). */
static int random_ply_of(const orc_engine *e, uint32_t uid)
{
    uint32_t r[4];
    orc_philox((uint32_t)e->cfg.seed, (uint32_t)(e->cfg.seed >> 32), uid, 0, ORC_STREAM_RANDOM_PLY, 0, r);
    return (int)(((uint64_t)r[0] * 120u) >> 32);
}

/* p0: the first ply the game recorded (0, or the ply a loaded position was at); partial: the game began at a loaded
 * position (orc_engine_set_positions) — its record lacks the plies before, bit 30 of the header's result word says so */
static void finish_game(orc_engine *e, int g, int result, int p0, int partial)
{
    orc_game_state *s = &e->gs[g];
    int64_t size = 16;
    for (int p = p0; p < s->ply; p++) {
        const uint8_t *r = REC(e, g, p);
        uint16_t nd;
        memcpy(&nd, r + 18, 2);
        size += REC_HDR - 8 + 4 * (int64_t)nd;
    }
    blob *b = (blob *)malloc(sizeof(blob) + (size_t)size);
    b->next = NULL;
    b->size = size;
    int32_t hdr[4] = {g, (int32_t)s->uid, s->ply - p0, result};
    if (e->cfg.flags & ORC_FLAG_ONE_RANDOM_MOVE)
        hdr[3] |= (random_ply_of(e, s->uid) + 1) << 8;
    if (partial)
        hdr[3] |= 1 << 30;
    memcpy(b->data, hdr, 16);
    uint8_t *w = b->data + 16;
    for (int p = p0; p < s->ply; p++) {
        const uint8_t *r = REC(e, g, p);
        uint16_t nd;
        memcpy(&nd, r + 18, 2);
        memcpy(w, r, 24);
        w += 24;
        memcpy(w, r + REC_HDR, 4 * (size_t)nd);
        w += 4 * (size_t)nd;
    }
    if (e->done_tail) e->done_tail->next = b; else e->done_head = b;
    e->done_tail = b;
    e->done_count++;
}

/* generate_game ply body (:526-574) + MCTS::play (:475-492). */
static void advance_game(orc_engine *e, int g)
{
    orc_game_state *s = &e->gs[g];
    uint64_t *st = e->stats + (size_t)g * ORC_STAT_COUNT;
    int a = s->arena, b = 1 - a;
    uint32_t *ni = NI(e, a, g), *ed = ED(e, a, g);
    uint16_t *em = EM(e, a, g);
    uint64_t *nb = NB(e, a, g);
    uint32_t first = ni[0];
    int M = (int)(ni[1] & 0xFFFFu);
    /* sample_proportionally_to_visits (:495-506), integer form */
    uint32_t rnd[4];
    orc_philox((uint32_t)e->cfg.seed, (uint32_t)(e->cfg.seed >> 32), s->uid, (uint32_t)s->ply,
               ORC_STREAM_SAMPLE, 0, rnd);
    uint32_t N = (uint32_t)s->root_visits;
    int chosen = -1;
    if (e->cfg.flags & ORC_FLAG_SAMPLE_POW5) {
        /* sample_with_exponential_weight (engine.py:532-548), exponent 5: weights (n/N)^5 over
         * edges with n >= max/2; the common 1/N^5 cancels, so integers n^5 are exact */
        uint32_t maxn = 0;
        for (int j = 0; j < M; j++)
            if (ed[4 * (first + j) + 1] > maxn) maxn = ed[4 * (first + j) + 1];
        uint64_t T = 0, w[ORC_MAX_MOVES];
        for (int j = 0; j < M; j++) {
            uint64_t n = ed[4 * (first + j) + 1];
            w[j] = (2 * n >= maxn) ? n * n * n * n * n : 0;
            T += w[j];
        }
        uint64_t R = ((uint64_t)rnd[0] << 32) | rnd[1];
        uint64_t r = (uint64_t)(((unsigned __int128)R * T) >> 64);
        uint64_t cum = 0;
        for (int j = 0; j < M; j++) {
            cum += w[j];
            if (chosen < 0 && cum > r)
                chosen = j;
        }
    } else {
        uint32_t r = (uint32_t)(((uint64_t)rnd[0] * N) >> 32);
        uint32_t cum = 0;
        for (int j = 0; j < M; j++) {
            cum += ed[4 * (first + j) + 1];
            if (chosen < 0 && cum > r)
                chosen = j;
        }
    }
    if (chosen < 0)
        chosen = 0;
    if (e->cfg.flags & ORC_FLAG_ONE_RANDOM_MOVE) {
        int rp = random_ply_of(e, s->uid);
        if (s->ply == rp) {
            /* AT the randomization point: a uniformly random legal move (:531-540) */
            chosen = (int)(((uint64_t)rnd[1] * (uint32_t)M) >> 32);
        } else if (s->ply > rp) {
            /* AFTER it: the most visited move (:543-551); the reference's strict '>' over an
             * unordered_map leaves ties to hash order: fixed here as the first maximum */
            uint32_t best = 0;
            chosen = 0;
            for (int j = 0; j < M; j++)
                if (ed[4 * (first + j) + 1] > best) {
                    best = ed[4 * (first + j) + 1];
                    chosen = j;
                }
        }
    }
    /* record (:565-572) */
    uint8_t *rec = REC(e, g, s->ply);
    uint64_t bx = nb[0] & ~TURN_BIT, bo = nb[1];
    uint16_t mv = em[first + chosen];
    uint16_t nd = 0;
    for (int j = 0; j < M; j++) {
        const uint32_t *x = ed + 4 * (size_t)(first + j);
        if (x[3] != NONE && nd < REC_MAXD) {
            uint32_t ent = (uint32_t)em[first + j] | ((x[1] & 0xFFFFu) << 16);
            memcpy(rec + REC_HDR + 4 * (size_t)nd, &ent, 4);
            nd++;
        }
    }
    memset(rec, 0, REC_HDR);
    memcpy(rec + 0, &bx, 8);
    memcpy(rec + 8, &bo, 8);
    memcpy(rec + 16, &mv, 2);
    memcpy(rec + 18, &nd, 2);
    st[ORC_STAT_PLIES]++;

    /* play (:475-492): keep the chosen child's subtree, copied breadth-first
     * into the other arena. */
    uint32_t c = ed[4 * (size_t)(first + chosen) + 3];
    if (e->cfg.flags & ORC_FLAG_NO_REUSE)
        c = NONE;  /* the arena engines rebuild their tree every ply (engine.py:452-472) */
    uint32_t *ni2 = NI(e, b, g), *ed2 = ED(e, b, g);
    uint16_t *em2 = EM(e, b, g);
    uint64_t *nb2 = NB(e, b, g);
    s->arena = b;
    if (c == NONE) {
        /* miss: fresh tree from the position after the move (:479-483) */
        orc_pos p;
        unpack(e, nb, &p);
        orc_makemove(&p, mv & 0xFF, mv >> 8);
        s->n_nodes = 0;
        s->n_edges = 0;
        make_node(e, g, b, 0, &p, NULL);
        s->n_nodes = 1;
        s->root_visits = 0;
    } else {
        memcpy(nb2, nb + 2 * (size_t)c, 16);
        memcpy(ni2, ni + 4 * (size_t)c, 16);
        uint32_t t = 1, eb = 0;
        for (uint32_t q = 0; q < t; q++) {
            uint32_t of = ni2[4 * q + 0];
            uint32_t Mq = ni2[4 * q + 1] & 0xFFFFu;
            uint32_t nf = eb;
            eb += Mq;
            for (uint32_t j = 0; j < Mq; j++) {
                const uint32_t *src = ed + 4 * (size_t)(of + j);
                uint32_t *dst = ed2 + 4 * (size_t)(nf + j);
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
                em2[nf + j] = em[of + j];
                uint32_t oc = src[3];
                if (oc != NONE) {
                    uint32_t nc = t++;
                    memcpy(nb2 + 2 * (size_t)nc, nb + 2 * (size_t)oc, 16);
                    memcpy(ni2 + 4 * (size_t)nc, ni + 4 * (size_t)oc, 16);
                    dst[3] = nc;
                } else {
                    dst[3] = NONE;
                }
            }
            ni2[4 * q + 0] = Mq ? nf : 0;
        }
        s->n_nodes = (int32_t)t;
        s->n_edges = (int32_t)eb;
        uint32_t rv = 0;
        uint32_t Mr = ni2[1] & 0xFFFFu;
        for (uint32_t j = 0; j < Mr; j++)
            rv += ed2[4 * (size_t)(ni2[0] + j) + 1];
        s->root_visits = (int32_t)rv;
        st[ORC_STAT_REROOT_NODES] += t;
        st[ORC_STAT_REROOT_EDGES] += eb;
    }
    if (e->cfg.flags & ORC_FLAG_EVAL_CACHE) {
        /* the kept subtree's evaluations stay usable (all nodes but the root, whose priors get this ply's noise) */
        tt_clear(e, b, g);
        for (uint32_t n = 1; n < (uint32_t)s->n_nodes; n++)
            if ((ni2[4 * n + 1] >> 16) == 0 && (ni2[4 * n + 1] & 0xFFFFu) != 0)
                tt_insert(e, b, g, n);
    }
    s->ply += 1;
    e->force[g] = 0;
    int result = (int)(ni2[1] >> 16);
    /* a game that began at a loaded position is played and counted like any other; its record (from the loaded ply on)
     * is handed out marked partial — the self-play callers do not write it, the arena's do (a match from openings) */
    const int loaded = e->no_emit[g];
    const int p0 = loaded ? loaded - 1 : 0;
    if (result != 0 && (e->cfg.flags & ORC_FLAG_ONE_RANDOM_MOVE) && random_ply_of(e, s->uid) + 1 >= s->ply) {
        /* "Skipping game with no board state just after the uniformly random move" (:632-637) */
        st[ORC_STAT_DROPPED]++;
        init_game(e, g, s->uid + (uint32_t)e->G);
    } else if (result != 0) {
        finish_game(e, g, result, p0, loaded != 0);
        st[ORC_STAT_GAMES]++;
        init_game(e, g, s->uid + (uint32_t)e->G);
    } else if (s->ply >= e->cfg.max_plies) {
        st[ORC_STAT_DROPPED]++; /* null-result games are skipped (:628-631) */
        if (e->cfg.flags & ORC_FLAG_KEEP_UNFINISHED)
            finish_game(e, g, 0, p0, loaded != 0); /* arena: "invalid" -> annulled (uai_ringmaster.py:147-150) */
        init_game(e, g, s->uid + (uint32_t)e->G);
    } else {
        s->phase = ORC_PHASE_ROOT_EVAL;
    }
}

void orc_engine_backup(orc_engine *e, const float *logits, const float *values)
{
    for (int g = 0; g < e->G; g++) {
        orc_game_state *s = &e->gs[g];
        const float *lg = logits + (size_t)g * 833;
        switch (s->leaf_kind) {
        case ORC_LEAF_ROOT:
            apply_priors(e, g, lg, 1);
            s->phase = ORC_PHASE_SEARCH;
            break;
        case ORC_LEAF_EVAL:
            apply_priors(e, g, lg, 0);
            if (e->cfg.flags & ORC_FLAG_EVAL_CACHE) {
                NI(e, s->arena, g)[4 * (size_t)s->leaf_node + 3] = orc_f2u(values[g]);
                tt_insert(e, s->arena, g, (uint32_t)s->leaf_node);
            }
            backup_path(e, g, values[g]);
            break;
        case ORC_LEAF_TERMINAL: {
            uint32_t tv = NI(e, s->arena, g)[4 * (size_t)s->leaf_node + 3];
            backup_path(e, g, orc_u2f(tv));
            break;
        }
        default:
            break;
        }
        if (s->leaf_kind == ORC_LEAF_DESCENT)
            continue; /* parked descent: nothing to back up, and the tree must stay as it is */
        s->leaf_kind = ORC_LEAF_NONE;
        /* while (root.all_edge_visits < global_visits) step(); (:522-525): the move is due */
        if (s->phase == ORC_PHASE_SEARCH && (s->root_visits >= e->cfg.visits || e->force[g]))
            s->phase = ORC_PHASE_ADVANCING;
    }
}

void orc_engine_game_state(const orc_engine *e, int g, orc_game_state *out) { *out = e->gs[g]; }

void orc_engine_tree(const orc_engine *e, int g, uint64_t *boards, uint32_t *info, uint32_t *edges,
                     uint16_t *moves)
{
    const orc_game_state *s = &e->gs[g];
    memcpy(boards, NB(e, s->arena, g), 16 * (size_t)s->n_nodes);
    memcpy(info, NI(e, s->arena, g), 16 * (size_t)s->n_nodes);
    memcpy(edges, ED(e, s->arena, g), 16 * (size_t)s->n_edges);
    memcpy(moves, EM(e, s->arena, g), 2 * (size_t)s->n_edges);
}

void orc_engine_stats(const orc_engine *e, uint64_t *out)
{
    memset(out, 0, sizeof(uint64_t) * ORC_STAT_COUNT);
    for (int g = 0; g < e->G; g++)
        for (int k = 0; k < ORC_STAT_COUNT; k++)
            out[k] += e->stats[(size_t)g * ORC_STAT_COUNT + k];
}

int64_t orc_engine_pop_game(orc_engine *e, uint8_t *buf, int64_t cap)
{
    blob *b = e->done_head;
    if (!b || b->size > cap)
        return 0;
    memcpy(buf, b->data, (size_t)b->size);
    int64_t n = b->size;
    e->done_head = b->next;
    if (!e->done_head) e->done_tail = NULL;
    e->done_count--;
    free(b);
    return n;
}

int orc_engine_pending_games(const orc_engine *e) { return e->done_count; }

float orc_probe_expf(float x) { return orc_det_expf(x); }
float orc_probe_logf(float x) { return orc_det_logf(x); }
float orc_probe_gamma(float alpha, uint64_t seed, uint32_t uid, uint32_t ply, uint32_t edge)
{
    return orc_det_gamma(alpha, (uint32_t)seed, (uint32_t)(seed >> 32), uid, ply, edge);
}
void orc_probe_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t *out4)
{
    orc_philox((uint32_t)seed, (uint32_t)(seed >> 32), c0, c1, c2, c3, out4);
}
