/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's 7x7 Ataxx rules, used solely as the
 * checker for the HIP path (tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg).  Nothing under ataxxzero_amd/ may include, link or call
 * this file.
 *
 * Each function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  Parity pin: tests/golden/rules_*.json.gz and
 * tests/golden/perft.json, generated from the reference's own Python rules
 * (ataxx_rules.py, perft.py) by tests/golden/gen_rules_fixtures.py.
 *
 * Conventions:
 *   bit index   = file + 7*rank0, a1 = bit 0 (cpp/bitboards.hpp:9-25)
 *   python x,y  = file, 6 - rank0              (ataxx_rules.py:74-80)
 *   move code   = from | to << 8 ; clone <=> from == to (cpp/move.hpp:9-33)
 */
#ifndef ATAXX_RULES_ORACLE_H
#define ATAXX_RULES_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BOARD_MASK 0x1FFFFFFFFFFFFULL
#define ORC_MAX_MOVES 256
#define ORC_NO_MOVE 0xFFFFu

typedef struct {
    uint64_t pieces[2]; /* [0] = x / CROSS (moves first), [1] = o / NOUGHT */
    uint64_t blockers;
    int32_t turn;       /* 0 = x to move, 1 = o to move */
    int32_t ply;
} orc_pos;

uint64_t orc_singles(int sq);                 /* cpp/bitboards.hpp:32 (table), derived */
uint64_t orc_doubles(int sq);                 /* cpp/bitboards.hpp:33 (table), derived */
uint64_t orc_single_jump_bb(uint64_t bb);     /* cpp/bitboards.cpp:6-16 */
int orc_set_fen(orc_pos *pos, const char *fen);            /* cpp/ataxx.cpp:14-90 */
int orc_fen(const orc_pos *pos, char *out, int cap);       /* ataxx_rules.py:95-106 */
int orc_movegen(const orc_pos *pos, uint16_t *moves);      /* cpp/movegen.cpp:10-79 */
void orc_makemove(orc_pos *pos, int from, int to);         /* cpp/makemove.cpp:56-76 */
void orc_pass(orc_pos *pos);                               /* ataxx_rules.py:112-114 */
int orc_result(const orc_pos *pos, uint16_t *moves, int *num_moves); /* cpp/self_play_client.cpp:109-144 */
uint64_t orc_perft(const orc_pos *pos, int depth);         /* perft.py:5-16 */
int orc_move_string(uint16_t move, char *out);             /* cpp/move.cpp:11-21 */
int orc_policy_index(uint16_t move);                       /* cpp/self_play_client.cpp:220-237 */
void orc_features(const orc_pos *pos, float *out196);      /* cpp/self_play_client.cpp:174-202 */
void orc_board_cells(const orc_pos *pos, int32_t *out49);  /* cpp/self_play_client.cpp:88-107 */

#ifdef __cplusplus
}
#endif
#endif
