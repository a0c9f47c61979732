/*
 * ataxxzero_hip.h — C ABI of the MI355X-native batched Ataxx self-play engine.
 *
 * Drop-in boundary for the reference's self-play hot path.  Plain C types only
 * (pointers, sizes, ints); no torch / Python objects cross this line.  All
 * `azh_*` functions return 0 on success and a negative code on failure, with a
 * message available from azh_last_error().  Host pointers are caller-owned and
 * are fully consumed (copied) before the call returns unless stated otherwise.
 *
 * Each entry point names the reference interface it replaces (file:line under
 * /root/reference).  The binding a maintainer of the reference would add is in
 * INTEGRATION.md; the Python side of this repo binds it in ataxxzero_amd/link.py.
 *
 * Bit conventions (cpp/bitboards.hpp:9-25, cpp/ataxx.hpp:28-34): square =
 * file + 7*rank0, a1 = bit 0.  A packed board is two u64: word0 = x stones with
 * the side to move in bit 63 (0 = x, 1 = o), word1 = o stones.  A "leaf board"
 * is (mover stones, opponent stones).  A move is u16 = from | to << 8, clone
 * <=> from == to (cpp/move.hpp:9-33).  Policy rows are the reference's
 * (7,7,17) f32 logits, flat index 119*to_x + 17*to_y + layer
 * (cpp/self_play_client.cpp:220-237); feature rows are (7,7,4) f32
 * (cpp/self_play_client.cpp:174-202).
 */
#ifndef ATAXXZERO_HIP_H
#define ATAXXZERO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZH_POLICY_SIZE 833
#define AZH_FEATURE_SIZE 196
#define AZH_MAX_MOVES 256
#define AZH_STAT_COUNT 16

enum { AZH_DTYPE_F32 = 0, AZH_DTYPE_BF16 = 1, AZH_DTYPE_F16 = 2 };
enum { AZH_LEAF_NONE = 0, AZH_LEAF_EVAL = 1, AZH_LEAF_TERMINAL = 2, AZH_LEAF_ROOT = 3, AZH_LEAF_DESCENT = 4 };

/* ------------------------------------------------------------------ errors */
const char *azh_last_error(void);
/* number of visible HIP devices, or a negative code when the runtime is unusable */
int azh_device_count(void);
/* select the device used by every later call from this process (one process per GPU) */
int azh_set_device(int device);
/* PCI bus id ("0000:05:00.0") of HIP device `device`: bench.py --gpus N prints it per rank, so that two ranks on one card
 * (or a straggling card) can be told from the aggregate line. */
int azh_device_pci_bus_id(int device, char *buf, int cap);

/* ------------------------------------------------------------------ rules
 * Replaces cpp/movegen.cpp:10-79 (movegen), cpp/makemove.cpp:56-76 (makemove),
 * cpp/self_play_client.cpp:109-144 (get_board_result) and perft.py:5-26. */

/* perft node count on the GPU with perft.py's "a position without a move has one
 * pass child" semantics. */
int azh_perft(uint64_t x, uint64_t o, uint64_t blockers, int turn, int depth, uint64_t *nodes_out);

/* For n packed boards: legal moves in the reference's movegen order
 * (moves_out [n][AZH_MAX_MOVES]), their count, and the adjudication
 * 0 / 1 / 2.  Any output pointer may be NULL. */
int azh_rules_batch(int n, const uint64_t *boards, uint64_t blockers, uint16_t *moves_out,
                    int32_t *counts_out, int32_t *results_out);

/* boards_out[i] = makemove(boards[i], moves[i]); a move of 0xFFFF passes
 * (ataxx_rules.py:112-114). */
int azh_makemove_batch(int n, const uint64_t *boards, const uint16_t *moves, uint64_t *boards_out);

/* Reference feature rows (cpp/self_play_client.cpp:174-202, engine.py:53-73) for n
 * leaf boards (mover, opponent): out [n][7][7][4] f32. */
int azh_features_batch(int n, const uint64_t *leaf_boards, uint64_t blockers, float *out);

/* Uniform-random playouts (generate_games.py:16-75 with --random-play): plays
 * n_games games from the given start to the end or max_plies.  Outputs, all
 * optional: plies[n], results[n] (0 = unfinished), and the per-ply trace
 * boards_out [n][max_plies][2] (x, o), moves_out [n][max_plies]. */
int azh_random_play(int n_games, uint64_t seed, uint64_t x, uint64_t o, uint64_t blockers, int turn,
                    int max_plies, int32_t *plies, int32_t *results, uint64_t *boards_out,
                    uint16_t *moves_out);

/* Test hook for the engine's deterministic f32 math / Philox4x32-10 (the arithmetic
 * the search's bit-exact contract with the CPU oracle rests on).  kind 0: exp(in[i]);
 * 1: log(in[i]); 2: gamma(alpha = in[0]) keyed (seed, aux[3i..3i+2] = uid, ply, edge);
 * 3: philox(seed; aux[4i..4i+3]) -> out[4i..4i+3].  out holds raw 32-bit patterns. */
int azh_probe_detmath(int kind, int n, const float *in, const uint32_t *aux, uint64_t seed, uint32_t *out);

/* ------------------------------------------------------------------ network
 * Replaces model.Network's forward pass (model.py:38-79,116-142) and
 * model.load_model (model.py:186-196). */
typedef struct azh_net azh_net;

/* conv_flat: the 2*blocks+5 parameter arrays of the .npy file's first list,
 * concatenated in file order: (3,3,4,F), 2*blocks x (3,3,F,F), (1,1,F,17),
 * (1,1,F,1), fc_w (49,1), fc_b (1,).  bn_flat: the 2*(2*blocks+1) arrays of the
 * second list, (moving_mean, moving_variance) per batch-norm in creation order.
 * gamma = 1, beta = 0 as after model.load_model;
 * bn_eps is TensorFlow's default 1e-3.  filters: 128 (model.py:16; the tuned towers), 64 or 256 (model.Network.FILTERS
 * is a class attribute that uai_interface.py:92-93 patches: these run on the width-templated 32x32 tower). */
int azh_net_create(int blocks, int filters, const float *conv_flat, const float *bn_flat,
                   float bn_eps, azh_net **out);
void azh_net_destroy(azh_net *net);

/* logits_out [n][833] f32 (raw policy logits, model.py:66-69), values_out [n]
 * (tanh value, model.py:71-79), for n leaf boards (mover, opponent). */
int azh_net_forward(azh_net *net, int dtype, int n, const uint64_t *leaf_boards, uint64_t blockers,
                    float *logits_out, float *values_out);
/* The same with test-time symmetry averaging, nn_evals.evaluate (nn_evals.py:48-62): the tower runs the 8
 * dihedral images of every board in one launch; logits come back through the inverse symmetry
 * (spatially) and are averaged, values are averaged. */
int azh_net_forward_sym(azh_net *net, int dtype, int n, const uint64_t *leaf_boards, uint64_t blockers,
                        float *logits_out, float *values_out);

/* Measurement hook: average HIP-event milliseconds per launch of the tower kernel over
 * n synthetic boards (iters launches on one stream, 3 untimed warm-up launches). */
int azh_net_bench(azh_net *net, int dtype, int n, int iters, float *ms_out);
/* THIN batches (a match's last games, a UAI engine's single position): the 16-bit towers with ONE board per workgroup — the
 * latency of a launch of a handful of boards is one workgroup's time for the 25 layers, and a board alone in its workgroup
 * needs 288 MFMAs per wave and layer instead of the 3-board workgroup's 720.  Same net, same boards -> the same logits and
 * values up to the summation order (the last bits of the 16-bit towers differ from the 3-board kernel's); f32 and other
 * widths run the ordinary tower.  azh_net_forward_thin / azh_net_bench_thin: azh_net_forward / azh_net_bench through that
 * kernel. */
int azh_net_forward_thin(azh_net *net, int dtype, int n, const uint64_t *leaf_boards, uint64_t blockers,
                         float *logits_out, float *values_out);
int azh_net_bench_thin(azh_net *net, int dtype, int n, int iters, float *ms_out);
/* Diagnostic build of the bf16 tower with s_memtime stamps per layer phase: copies the
 * stamps of the first `wgs` workgroups, [wg][wave][128] u64, to out (layout: net_kernels.hip). */
int azh_net_stamps(azh_net *net, int n, int wgs, uint64_t *out);

/* ------------------------------------------------------------------ engine
 * Replaces the worker threads of cpp/self_play_client.cpp: MCTS::step (:419-473),
 * Evaluations::populate (:153-272), MCTS::play (:475-492),
 * sample_proportionally_to_visits (:495-506), generate_game (:508-582). */
typedef struct azh_engine azh_engine;

typedef struct {
    int32_t games;           /* concurrent game slots ("threads" in the reference) */
    int32_t visits;          /* global_visits (:46,:522) */
    int32_t max_plies;       /* maximum_game_plies (:34) */
    int32_t edges_per_node;  /* edge arena per game = (visits + 8) * edges_per_node */
    float c_puct;            /* exploration_parameter (:31) */
    float dirichlet_alpha;   /* (:32) */
    float dirichlet_weight;  /* (:33) */
    int32_t start_turn;
    uint64_t seed;
    uint64_t start_x, start_o, blockers; /* STARTING_GAME_POSITION (:23) */
    uint32_t flags;          /* AZH_FLAG_*; 0 = the C++ self-play generator's behaviour */
    uint32_t select_budget;  /* 0 = every descent finishes inside one select; k > 0 = at most k tree levels per
                                select: a deeper descent parks (AZH_LEAF_DESCENT, no leaf this iteration) and
                                resumes next iteration where it stopped.  A parked game's tree does not change in
                                between, so each game plays bit for bit what it plays with budget 0; the launch no
                                longer lasts as long as the deepest descent of the batch */
} azh_config;

/* Behaviour switches that turn the self-play search into the arena search, i.e. the
 * Python engine the reference's uai_ringmaster.py drives (engine.py, uai_interface.py): */
enum {
    AZH_FLAG_NO_REUSE = 1,        /* fresh tree every ply: with per-ply "moves" messages engine.set_state
                                     (engine.py:452-472) never finds its grand-child and rebuilds the tree */
    AZH_FLAG_TIE_FIRST = 2,       /* python max(): first maximal move (engine.py:291), not the C++ last (:354) */
    AZH_FLAG_PY_POSTERIOR = 4,    /* 833-way softmax, gather, / (sum_legal + 1e-6) (engine.py:197-203) */
    AZH_FLAG_SAMPLE_POW5 = 8,     /* move ~ (n/N)^5 over edges with n >= max/2 (engine.py:532-548, exponent 5
                                     at uai_interface.py:76-79) */
    AZH_FLAG_KEEP_UNFINISHED = 16,/* games cut at max_plies are reported with result 0 ("invalid" ->
                                     annulled, uai_ringmaster.py:147-150) instead of dropped */
    AZH_FLAG_TWO_NETS = 32,       /* arena: the side to move alternates between two nets; slot parity picks
                                     which net plays x; records carry "slot" and "uid" */
    AZH_FLAG_ARENA = 1 | 2 | 4 | 8 | 16 | 32,
    AZH_FLAG_SYMMETRY_AVG = 128,  /* every evaluation is nn_evals.evaluate (nn_evals.py:48-62): the mean over the 8
                                     dihedral symmetries of the board, logits brought back spatially (move-type
                                     layers not permuted, as the reference), values averaged; 8x the tower work */
    AZH_FLAG_EVAL_CACHE = 256,    /* engine.py's NNEvaluator.cache (engine.py:127-234) for the device loop: a position this
                                     game's search has already evaluated (a transposition, a re-visited position: about
                                     a quarter of the leaves at 400 sims/move) takes its priors and value from the node
                                     that carries them instead of going to the net again.  Off by default: the C++
                                     generator evaluates every new node.  The net being deterministic, the trees are the
                                     ones the uncached search builds; AZH_STAT_NN_EVALS falls, AZH_STAT_CACHE_HITS counts */
    AZH_FLAG_ONE_RANDOM_MOVE = 64 /* the ONE_RANDOM_MOVE build of the client (cpp/self_play_client.cpp:515-552):
                                     per game one ply in 0..119 plays a uniformly random legal move, every later
                                     ply the most visited move; the entry gains "random_ply" (train.py:47-49) */
};

typedef struct {
    int32_t phase, arena, n_nodes, n_edges, ply, root_visits, leaf_kind, leaf_node, path_len;
    uint32_t uid;
} azh_game_state;

enum {
    AZH_STAT_STEPS = 0, AZH_STAT_NN_EVALS, AZH_STAT_LEVELS, AZH_STAT_CHILDREN, AZH_STAT_NEW_MOVES,
    AZH_STAT_PLIES, AZH_STAT_GAMES, AZH_STAT_DROPPED, AZH_STAT_EDGE_OVERFLOW, AZH_STAT_REROOT_NODES,
    AZH_STAT_REROOT_EDGES, AZH_STAT_RING_OVERFLOW, AZH_STAT_CACHE_HITS,
    AZH_STAT_PARKED,        /* (game, iteration) pairs in which a descent was parked by select_budget */
    AZH_STAT_REROOT_SPILLS  /* re-roots whose breadth-first frontier outgrew its LDS queue (the rest went through HBM) */
};

typedef struct {
    /* sums of HIP-event elapsed milliseconds over the iterations SAMPLED since the last azh_engine_timing_reset
     * (every `enable`-th iteration of the device loop, events on the engine's stream), and the number of samples */
    double select_ms, net_ms, backup_ms; /* select_ms: the fused tree launch of the run loop (backup + advance +
                                            select + compaction); backup_ms: 0 there */
    int64_t iterations;
    int64_t net_evals; /* leaves evaluated by the net in those iterations */
} azh_timing;

int azh_engine_create(const azh_config *cfg, azh_engine **out);
void azh_engine_destroy(azh_engine *e);
int azh_engine_node_cap(const azh_engine *e);
int azh_engine_edge_cap(const azh_engine *e);

/* One search iteration = select -> evaluate -> backup.
 * select: PUCT descent + expansion in every game (one new leaf per game). */
int azh_engine_select(azh_engine *e, int32_t *n_leaves_out);
/* need_eval [G], leaf_boards [G][2] (mover, opponent); either may be NULL */
int azh_engine_leaves(azh_engine *e, int32_t *need_eval, uint64_t *leaf_boards);
/* the reference's request_evaluation role (:648-681): feature rows of the current leaf
 * batch, dense, in game order: out [n_leaves][7][7][4] f32; games_out [n_leaves]
 * (optional) = the game each row belongs to */
int azh_engine_leaf_features(azh_engine *e, float *out, int32_t *games_out);
/* evaluate the leaf batch on the device with the built-in net */
int azh_engine_eval(azh_engine *e, azh_net *net, int dtype);
/* or supply evaluations from outside (the reference's complete_workload role,
 * :723-738): host arrays indexed by game, logits [G][833], values [G] */
int azh_engine_set_evals(azh_engine *e, const float *logits, const float *values);
/* priors (+ root Dirichlet), backup, and — once the root has `visits` visits —
 * sample the move, record the ply, re-root, finish/restart games */
int azh_engine_backup(azh_engine *e);

/* `iterations` full iterations with the built-in net, enqueued asynchronously */
int azh_engine_run(azh_engine *e, azh_net *net, int dtype, int iterations);
/* the same run for n engines of one GPU (half-batches: the reference's double buffer, cpp/self_play_client.cpp:593-600),
 * their iterations enqueued in turn, so that all of them start with the first launches enqueued */
int azh_engines_run(azh_engine *const *engines, int n, azh_net *net, int dtype, int iterations);
/* arena (AZH_FLAG_TWO_NETS): net_a plays x in even slots and o in odd slots, net_b the
 * other way round (uai_ringmaster.py:241-247 queues every pairing both ways).  Per iteration the leaves whose mover is net_a and
 * those whose mover is net_b are evaluated by ONE launch of the 16-bit tower (each workgroup picks its weight set from the list
 * it serves; the nets may differ in depth; f32, other widths and AZH_FLAG_SYMMETRY_AVG: one launch per net), with results bit
 * for bit those of separate launches.  A match is a fixed cohort under azh_engine_set_game_limit, and its last games want
 * azh_engine_set_thin_batches (below). */
int azh_engine_run_arena(azh_engine *e, azh_net *net_a, azh_net *net_b, int dtype, int iterations);
int azh_engine_sync(azh_engine *e);
/* change the root-visit threshold (global_visits) for the coming moves; 1 <= visits <= the
 * value the engine was created with */
int azh_engine_set_visits(azh_engine *e, int visits);
/* Which tower the device-resident loop evaluates its leaves with: 0 the 3-board workgroups (throughput), 1 one board per
 * workgroup (latency: azh_net_forward_thin's kernel), -1 (default) by the engine's size — thin for engines of at most
 * AZH_THIN_MAX_GAMES game slots.  A host that knows its batch has thinned out (a match under a game limit whose last games
 * are running, uai_ringmaster.py:221-262) switches at a drain; results of the 16-bit towers differ in the last bits between
 * the two kernels, so switch at points that do not depend on timing. */
#define AZH_THIN_MAX_GAMES 512
int azh_engine_set_thin_batches(azh_engine *e, int mode);

/* Measurement set-up hook: every slot restarts at a given position — boards [games][2] packed (x | turn << 63, o),
 * plies [games] — with a fresh tree.  Such games are played, counted (AZH_STAT_GAMES / _DROPPED), and their records are
 * assembled, drained and formatted like any other (the measured path does the same work per finished game), but no line is
 * handed out: the record lacks the plies before the start.  bench.py loads the positions a long-running generator would
 * be found at instead of waiting a game generation (about 70 s at 400 sims/move) for the steady state to form. */
int azh_engine_set_positions(azh_engine *e, const uint64_t *boards, const int32_t *plies);

int azh_engine_game_state(azh_engine *e, int game, azh_game_state *out);
/* arena dump: boards [n_nodes][2] u64, info [n_nodes][4] u32
 * (first_edge, n_edges | result << 16, 0, terminal value bits), edges
 * [n_edges][4] u32 (prior bits, visits, total score bits, child), moves [n_edges] */
int azh_engine_tree(azh_engine *e, int game, uint64_t *boards, uint32_t *info, uint32_t *edges,
                    uint16_t *moves);
/* Diagnostic: the same edges as the 16-byte records the tree kernels read — edges [n_edges][4] u32 = prior bits (bit 31:
 * the mark of the child the last descent through the node chose — the early request of the next level, never part of a
 * decision), total score bits, visits | child << 16 (0xFFFF: none), the child's edge range (first | count << 23 |
 * finished << 31).  tests/test_gpu_engine.py checks the mark's invariants on it. */
int azh_engine_tree_raw(azh_engine *e, int game, uint32_t *edges);
int azh_engine_stats(azh_engine *e, uint64_t *out /* [AZH_STAT_COUNT] */);
/* enable = 0: off; n > 0: bracket every n-th iteration of the device loop with events (tower start / tower end /
 * next tower start); at most 8192 samples are kept */
/* Diagnostic: runs two iterations of the device loop with the tree launch between their towers stamped by
 * s_memrealtime (100 MHz): out [games][10] u64 = wave start, state loaded, backup done, move-due mark done, descent done,
 * expansion done, state stored, workgroup (its four games, one beyond 8192) done, then two counts: levels descended, children scanned —
 * tools/tree_stamps.py turns them into a breakdown. */
int azh_engine_tree_stamps(azh_engine *e, azh_net *net, int dtype, uint64_t *out);
int azh_engine_timing_reset(azh_engine *e, int enable);
int azh_engine_timing(azh_engine *e, azh_timing *out);

/* Finished games as JSON lines in the reference's format
 * (cpp/self_play_client.cpp:512,565-578,639-641: keys boards, dists, moves,
 * result; one compact object per line).  Writes whole lines only; *used = bytes
 * written, *n_games = lines written; call again while *n_games > 0. */
int azh_engine_drain_json(azh_engine *e, char *buf, int64_t cap, int64_t *used, int32_t *n_games);
/* Takes the finished games off the device without formatting them: waits for the work enqueued so far, copies the record
 * ring to the host and empties it; the next azh_engine_drain_json formats what was fetched and does not touch the device.
 * A host loop that calls fetch, enqueues its next azh_engine_run and only then drains has the formatting and its own file
 * writes running under that run instead of in front of it (the reference's workers write their games from their own
 * threads, cpp/self_play_client.cpp:637-642: there, too, nobody waits for a game to be written).  Optional: a drain with
 * nothing fetched fetches by itself whenever work has been enqueued since the last fetch (so a caller that never fetches gets
 * one fetch per round, whether it drains in a loop until *n_games == 0 or with one call per round) — except inside the
 * drain sequence that follows an explicit fetch (the calls up to and including the first one that returns *n_games == 0):
 * those never touch the device, whatever was enqueued meanwhile.  A fetch touches only its own engine's streams: with
 * several engines on one GPU (half-batches) fetching one does not wait for the others' runs. */
int azh_engine_fetch(azh_engine *e);
/* 1 while work enqueued on this engine's streams is still in flight, 0 when they are idle (< 0: error).  Never waits. */
int azh_engine_query(azh_engine *e);
/* How many times azh_engine_drain_json had to fetch by itself (and so waited for the device) since the engine was created:
 * 0 for a host loop that fetches explicitly before every drain sequence. */
long long azh_engine_implicit_fetches(const azh_engine *e);
/* The line of ONE finished-game record (the words between two ring headers, as the device loop leaves them: 8-word header
 * {magic, slot, uid, plies, result, words, random_ply + 1, kind}, then per ply {x lo, x hi, o lo, o hi, move | nd << 16, 0,
 * nd x (move | visits << 16)}) exactly as azh_engine_drain_json writes it, without the newline: what the reference's
 * `entry.dump()` gives (cpp/self_play_client.cpp:565-578,639-641: nlohmann::json — sorted keys, no whitespace, floats as
 * the digits its Grisu2 finds, plain decimals from 1e-4 up, d.ddde-XX below).  Host code only, usable without a device.
 * *used = bytes the line has; -6 if `cap` is smaller (nothing written), -2 if the words are not a well-formed record.
 * with_ids: the arena's two extra keys (slot, uid). */
int azh_format_record_json(const uint32_t *rec, int64_t words, int32_t with_ids, char *buf, int64_t cap, int64_t *used);
/* Order in which azh_engine_drain_json hands games out: 0 (default) as they finish; 1 by game uid (slot g plays
 * uids g, g + games, g + 2 games, ...): a finished game is held back until every game with a smaller uid has been
 * handed out or dropped.  The reference's workers write games as they finish (Worker::thread_main :637-642) and
 * looper.py:51-64 stops the generator at --game-count lines, i.e. keeps the games that finished FIRST — the short
 * ones; with thousands of games in flight that bias is no longer slight, and uid order removes it (the first N
 * lines are the first N games started).  Call before the first drain. */
int azh_engine_set_emit_order(azh_engine *e, int by_uid);
/* Play at most `games` games (uids 0 .. games - 1) and then stop searching: a slot whose next game would be past the
 * limit goes idle, the batch thins out as the last games end.  For a generator that was given a target count: in uid
 * order line N appears once the slowest of the first N games has ended, and without a limit every other slot meanwhile
 * plays games nobody will read.  The limit may be raised later (dropped games leave the caller short of lines): idle
 * slots whose next game is now below it start it; games that have begun are never stopped.  (The reference's client has
 * no such notion: it is stopped from outside, looper.py:51-64.) */
int azh_engine_set_game_limit(azh_engine *e, int64_t games);

/* ------------------------------------------------------------------ reference ABI
 * The four symbols link.py:6-32 binds (cpp/self_play_client.cpp:683-749), with
 * the worker threads replaced by GPU game slots: `thread_count` concurrent
 * games are split into two halves of `buffer_entries` leaf rows; the host
 * evaluates a filled (buffer_entries,7,7,4) f32 buffer and hands back
 * (buffer_entries,7,7,17) posteriors (raw logits) + (buffer_entries,1) values.
 * (`shutdown` shadows the libc socket call of the same name exactly as the
 * reference's library does; define AZH_NO_REFERENCE_ABI to hide these.) */
#ifndef AZH_NO_REFERENCE_ABI
void launch_threads(char *output_path, int visits, float *fill_buffer1, float *fill_buffer2,
                    int buffer_entries, int thread_count);
int get_workload(void);
void complete_workload(int workload, float *posteriors, float *values);
void shutdown(void);
#endif

#ifdef __cplusplus
}
#endif
#endif
