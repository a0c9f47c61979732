/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * Sequential CPU restatement of the reference's batched self-play search
 * (cpp/self_play_client.cpp:153-582): PUCT select (:310-366,:386-417), expand +
 * evaluate + backup (:419-473), visit-proportional move sampling (:495-506),
 * tree reuse on play (:475-492), Dirichlet root noise (:250-271), game records
 * (:508-582).  The reference is irreproducible by construction (unseeded shared
 * RNG, hash-order tie-breaks: SURVEY.md §7 "hard parts"), so this oracle fixes
 * the two free choices — ties go to the LAST maximal edge in movegen order
 * (the reference's `>=`, :354) and all randomness is Philox keyed by
 * (seed, game uid, ply, stream) — and the HIP engine must then match it bit
 * for bit on every integer and every f32 it stores.
 *
 * Used by tests/ as the checker and by bench.py's cpu_baseline leg only.
 */
#ifndef MCTS_ORACLE_H
#define MCTS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t games;            /* concurrent game slots */
    int32_t visits;           /* root visit threshold per move (global_visits, :46,:522) */
    int32_t max_plies;        /* maximum_game_plies = 400 (:34) */
    int32_t edges_per_node;   /* edge arena = node_cap * edges_per_node */
    float c_puct;             /* exploration_parameter = 1.0 (:31) */
    float dirichlet_alpha;    /* 0.15 (:32) */
    float dirichlet_weight;   /* 0.25 (:33) */
    int32_t start_turn;
    uint64_t seed;
    uint64_t start_x, start_o, blockers; /* STARTING_GAME_POSITION (:23) */
    uint32_t flags;           /* ORC_FLAG_*: the arena variants of the search (engine.py) */
    uint32_t select_budget;   /* 0 = a descent always finishes within one orc_engine_select; k > 0 = at most k tree
                                 levels per call: a longer descent parks (ORC_LEAF_DESCENT, no leaf this
                                 iteration) and resumes at the next call from the node it stopped at.  The tree of
                                 a parked game does not change in between, so every game plays bit for bit what it
                                 plays with budget 0: only the iteration in which a leaf reaches the evaluator moves */
} orc_config;

enum {
    ORC_FLAG_NO_REUSE = 1, ORC_FLAG_TIE_FIRST = 2, ORC_FLAG_PY_POSTERIOR = 4, ORC_FLAG_SAMPLE_POW5 = 8,
    ORC_FLAG_KEEP_UNFINISHED = 16, ORC_FLAG_TWO_NETS = 32, ORC_FLAG_ARENA = 63,
    ORC_FLAG_ONE_RANDOM_MOVE = 64, /* the ONE_RANDOM_MOVE build of the client (cpp/self_play_client.cpp:515-552) */
    ORC_FLAG_EVAL_CACHE = 256      /* engine.py's NNEvaluator.cache (engine.py:127-234), per game: mirror of AZH_FLAG_EVAL_CACHE */
};

enum { ORC_LEAF_NONE = 0, ORC_LEAF_EVAL = 1, ORC_LEAF_TERMINAL = 2, ORC_LEAF_ROOT = 3, ORC_LEAF_DESCENT = 4 };
/* ORC_PHASE_ADVANCING: the move is due (root visits reached the threshold at the last backup).  The next select
 * gives the game no leaf and then plays the move (sampling, record, re-root): in the HIP engine the re-root runs
 * in its own launch beside the evaluator, so the iteration structure is part of the engine/oracle contract. */
enum { ORC_PHASE_ROOT_EVAL = 0, ORC_PHASE_SEARCH = 1, ORC_PHASE_ADVANCING = 2,
       ORC_PHASE_IDLE = 3 /* the slot's next game is past the game limit: no leaf, no move */ };

/* per-game scalar state, same field order as the HIP engine's snapshot */
typedef struct {
    int32_t phase, arena, n_nodes, n_edges, ply, root_visits, leaf_kind, leaf_node, path_len;
    uint32_t uid;
} orc_game_state;

enum {
    ORC_STAT_STEPS = 0,      /* MCTS steps (node-evals incl. terminal re-hits) */
    ORC_STAT_NN_EVALS,       /* leaves sent to the evaluator (incl. root refresh) */
    ORC_STAT_LEVELS,         /* L: levels descended */
    ORC_STAT_CHILDREN,       /* C: children scanned */
    ORC_STAT_NEW_MOVES,      /* M: legal moves of new nodes */
    ORC_STAT_PLIES,
    ORC_STAT_GAMES,
    ORC_STAT_DROPPED,
    ORC_STAT_EDGE_OVERFLOW,
    ORC_STAT_REROOT_NODES,
    ORC_STAT_REROOT_EDGES,
    ORC_STAT_CACHE_HITS,     /* expansions served by the evaluation cache */
    ORC_STAT_COUNT = 16
};

typedef struct orc_engine orc_engine;

orc_engine *orc_engine_create(const orc_config *cfg);
void orc_engine_destroy(orc_engine *e);
int orc_engine_node_cap(const orc_engine *e);
int orc_engine_edge_cap(const orc_engine *e);
/* root-visit threshold for the coming moves, 1 .. the value the engine was created with (the arenas are sized for
 * that); mirror of azh_engine_set_visits */
void orc_engine_set_visits(orc_engine *e, int visits);
/* mirror of azh_engine_set_game_limit (may be raised later: idle slots below the new limit start their game) */
void orc_engine_set_game_limit(orc_engine *e, int64_t games);
/* mirror of azh_engine_set_positions: every slot restarts at boards[g] (x | turn << 63, o) / plies[g] with a fresh tree;
 * such games are counted, not written */
void orc_engine_set_positions(orc_engine *e, const uint64_t *boards, const int32_t *plies);

/* phase 1 of an iteration: select/expand in every game; returns #leaves needing
 * the evaluator.  need_eval[g] (optional) = 1 for those games. */
int orc_engine_select(orc_engine *e, int32_t *need_eval);
/* with ORC_FLAG_TWO_NETS need_eval[g] is 1 (net A) or 2 (net B) for the side to move */
/* (mover, opponent) bitboards of every game's leaf, [G][2] */
void orc_engine_leaf_boards(const orc_engine *e, uint64_t *out);
/* reference feature rows (cpp/self_play_client.cpp:174-202) for game g's leaf */
void orc_engine_leaf_features(const orc_engine *e, int g, float *out196);
/* phase 2: priors / backup / ply advance. logits [G][833], values [G], indexed by game */
void orc_engine_backup(orc_engine *e, const float *logits, const float *values);

void orc_engine_game_state(const orc_engine *e, int g, orc_game_state *out);
/* arena dump of game g: boards [n][2] u64, info [n][4] u32, edges [m][4] u32, moves [m] u16 */
void orc_engine_tree(const orc_engine *e, int g, uint64_t *boards, uint32_t *info, uint32_t *edges,
                     uint16_t *moves);
void orc_engine_stats(const orc_engine *e, uint64_t *out /* ORC_STAT_COUNT */);

/* finished games, oldest first. Record = int32 header {slot, uid, plies, result}
 * then per ply {x u64, o u64, move u16, ndist u16, pad u32, ndist * u32 (move | n << 16)}.
 * With ORC_FLAG_ONE_RANDOM_MOVE the header's result word carries (random_ply + 1) << 8.
 * Returns bytes written (0 when none pending or cap too small for the next game). */
int64_t orc_engine_pop_game(orc_engine *e, uint8_t *buf, int64_t cap);
int orc_engine_pending_games(const orc_engine *e);

/* detmath probes for tests */
float orc_probe_expf(float x);
float orc_probe_logf(float x);
float orc_probe_gamma(float alpha, uint64_t seed, uint32_t uid, uint32_t ply, uint32_t edge);
void orc_probe_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t *out4);

#ifdef __cplusplus
}
#endif
#endif
