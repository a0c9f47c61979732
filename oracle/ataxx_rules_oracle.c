/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ataxx_rules_oracle.h).
 *
 * Plain-C restatement of the reference rules.  Written from the behaviour of
 * the cited reference lines; the neighbourhood masks are derived from board
 * geometry (Chebyshev distance 1 / 2 on a 7-wide board), not transcribed.
 */
#include "ataxx_rules_oracle.h"

#include <string.h>

static int popcount64(uint64_t v) { return __builtin_popcountll(v); }
static int lsb64(uint64_t v) { return __builtin_ctzll(v); }

/* Squares at Chebyshev distance exactly d from sq (cpp/bitboards.hpp:32-33). */
static uint64_t ring_mask(int sq, int d)
{
    int f = sq % 7, r = sq / 7;
    uint64_t m = 0;
    for (int rr = r - d; rr <= r + d; rr++) {
        for (int ff = f - d; ff <= f + d; ff++) {
            if (rr < 0 || rr >= 7 || ff < 0 || ff >= 7)
                continue;
            int df = ff > f ? ff - f : f - ff;
            int dr = rr > r ? rr - r : r - rr;
            if ((df > dr ? df : dr) != d)
                continue;
            m |= 1ULL << (ff + 7 * rr);
        }
    }
    return m;
}

uint64_t orc_singles(int sq) { return ring_mask(sq, 1); }
uint64_t orc_doubles(int sq) { return ring_mask(sq, 2); }

/* cpp/bitboards.cpp:6-16: union of the distance-1 rings of every set square. */
uint64_t orc_single_jump_bb(uint64_t bb)
{
    uint64_t out = 0;
    while (bb) {
        out |= ring_mask(lsb64(bb), 1);
        bb &= bb - 1;
    }
    return out & ORC_BOARD_MASK;
}

/* cpp/ataxx.cpp:14-90: FEN rows run from rank 7 down to rank 1; 'x','o','-',
 * digits skip; optional side token. Returns 0 on success. */
int orc_set_fen(orc_pos *pos, const char *fen)
{
    memset(pos, 0, sizeof(*pos));
    int sq = 42; /* a7 */
    const char *p = fen;
    for (; *p && *p != ' '; p++) {
        char c = *p;
        if (c == 'x' || c == 'X') {
            pos->pieces[0] |= 1ULL << sq++;
        } else if (c == 'o' || c == 'O') {
            pos->pieces[1] |= 1ULL << sq++;
        } else if (c == '-') {
            pos->blockers |= 1ULL << sq++;
        } else if (c >= '1' && c <= '7') {
            sq += c - '0';
        } else if (c == '/') {
            sq -= 14;
        } else {
            return 5;
        }
    }
    while (*p == ' ')
        p++;
    if (*p == 'o' || *p == 'O')
        pos->turn = 1;
    return 0;
}

/* ataxx_rules.py:95-106: rows top (rank 7) to bottom, runs of empties as digits. */
int orc_fen(const orc_pos *pos, char *out, int cap)
{
    int n = 0;
    for (int y = 0; y < 7; y++) {
        int run = 0;
        for (int x = 0; x < 7; x++) {
            uint64_t m = 1ULL << (x + 7 * (6 - y));
            char c = 0;
            if (pos->pieces[0] & m) c = 'x';
            else if (pos->pieces[1] & m) c = 'o';
            else if (pos->blockers & m) c = '-';
            if (!c) { run++; continue; }
            if (run) { if (n < cap - 1) out[n++] = (char)('0' + run); run = 0; }
            if (n < cap - 1) out[n++] = c;
        }
        if (run && n < cap - 1) out[n++] = (char)('0' + run);
        if (y != 6 && n < cap - 1) out[n++] = '/';
    }
    if (n < cap - 2) { out[n++] = ' '; out[n++] = pos->turn ? 'o' : 'x'; }
    out[n] = 0;
    return n;
}

/* cpp/movegen.cpp:10-79: jumps for every own stone (ascending from, then
 * ascending to), then one clone per reachable empty destination (ascending).
 * No pass is generated when there is no move. */
int orc_movegen(const orc_pos *pos, uint16_t *moves)
{
    uint64_t own = pos->pieces[pos->turn];
    uint64_t empty = ORC_BOARD_MASK & ~(pos->pieces[0] | pos->pieces[1] | pos->blockers);
    int n = 0;
    uint64_t copy = own;
    while (copy) {
        int from = lsb64(copy);
        uint64_t to_bb = orc_doubles(from) & empty;
        while (to_bb) {
            int to = lsb64(to_bb);
            moves[n++] = (uint16_t)(from | (to << 8));
            to_bb &= to_bb - 1;
        }
        copy &= copy - 1;
    }
    uint64_t clones = orc_single_jump_bb(own) & empty;
    while (clones) {
        int to = lsb64(clones);
        moves[n++] = (uint16_t)(to | (to << 8));
        clones &= clones - 1;
    }
    return n;
}

/* cpp/makemove.cpp:56-76. */
void orc_makemove(orc_pos *pos, int from, int to)
{
    uint64_t to_bb = 1ULL << to, from_bb = 1ULL << from;
    int us = pos->turn, them = !pos->turn;
    uint64_t captured = orc_singles(to) & pos->pieces[them];
    pos->pieces[us] &= ~from_bb;
    pos->pieces[us] ^= to_bb;
    pos->pieces[us] ^= captured;
    pos->pieces[them] ^= captured;
    pos->turn = them;
    pos->ply++;
}

/* ataxx_rules.py:112-114: a pass only flips the side to move. */
void orc_pass(orc_pos *pos) { pos->turn = !pos->turn; }

/* cpp/self_play_client.cpp:109-144 (twin: ataxx_rules.py:159-179):
 * 0 ongoing, 1 = x wins, 2 = o wins. */
int orc_result(const orc_pos *pos, uint16_t *moves, int *num_moves)
{
    uint16_t local[ORC_MAX_MOVES];
    int p1 = popcount64(pos->pieces[0]);
    int p2 = popcount64(pos->pieces[1]);
    int bl = popcount64(pos->blockers);
    int empty = 49 - p1 - p2 - bl;
    if (!moves)
        moves = local;
    if (p1 == 0) { if (num_moves) *num_moves = 0; return 2; }
    if (p2 == 0) { if (num_moves) *num_moves = 0; return 1; }
    int n = orc_movegen(pos, moves);
    if (num_moves)
        *num_moves = n;
    if (n == 0) {
        if (pos->turn == 0) p2 += empty;
        else p1 += empty;
    }
    if (p1 + p2 + bl == 49)
        return p1 < p2 ? 2 : 1;
    return 0;
}

/* perft.py:5-16: every position expands into its legal moves; a position with
 * no move contributes exactly one "pass" child (ataxx_rules.py:154-156). */
uint64_t orc_perft(const orc_pos *pos, int depth)
{
    if (depth == 0)
        return 1;
    uint16_t moves[ORC_MAX_MOVES];
    int n = orc_movegen(pos, moves);
    if (n == 0) {
        orc_pos c = *pos;
        orc_pass(&c);
        return orc_perft(&c, depth - 1);
    }
    if (depth == 1)
        return (uint64_t)n;
    uint64_t total = 0;
    for (int i = 0; i < n; i++) {
        orc_pos c = *pos;
        orc_makemove(&c, moves[i] & 0xFF, moves[i] >> 8);
        total += orc_perft(&c, depth - 1);
    }
    return total;
}

/* cpp/move.cpp:11-21 + cpp/ataxx.hpp:21-27: "b7" for a clone, "a7c6" for a jump. */
int orc_move_string(uint16_t move, char *out)
{
    int from = move & 0xFF, to = move >> 8, n = 0;
    if (from != to) {
        out[n++] = (char)('a' + from % 7);
        out[n++] = (char)('1' + from / 7);
    }
    out[n++] = (char)('a' + to % 7);
    out[n++] = (char)('1' + to / 7);
    out[n] = 0;
    return n;
}

/* cpp/self_play_client.cpp:77-86,220-237 (twin engine.py:75,98-110): flat index
 * into the (7,7,17) policy tensor = 119*to_x + 17*to_y + layer; clone -> layer
 * 16; jump -> layer = rank of (dx,dy) in the reference's enumeration of the 16
 * distance-2 offsets (ataxx_rules.py:17-20: dx major, dy minor, skipping the
 * 3x3 centre). */
int orc_policy_index(uint16_t move)
{
    int from = move & 0xFF, to = move >> 8;
    int fx = from % 7, fy = 6 - from / 7;
    int tx = to % 7, ty = 6 - to / 7;
    int layer;
    if (from == to) {
        layer = 16;
    } else {
        int dx = tx - fx, dy = ty - fy;
        layer = 0;
        for (int a = -2; a <= 2; a++) {
            for (int b = -2; b <= 2; b++) {
                int near = (a >= -1 && a <= 1 && b >= -1 && b <= 1);
                if (near)
                    continue;
                if (a == dx && b == dy)
                    return 119 * tx + 17 * ty + layer;
                layer++;
            }
        }
        return -1;
    }
    return 119 * tx + 17 * ty + layer;
}

/* cpp/self_play_client.cpp:174-202: [x][y][c], plane 0 ones, 1 mover, 2 opponent,
 * 3 blockers; x = file, y = 6 - rank0. */
void orc_features(const orc_pos *pos, float *out196)
{
    memset(out196, 0, 196 * sizeof(float));
    for (int y = 0; y < 7; y++) {
        for (int x = 0; x < 7; x++) {
            uint64_t m = 1ULL << (x + 7 * (6 - y));
            float *f = out196 + 28 * x + 4 * y;
            f[0] = 1.0f;
            if (pos->pieces[pos->turn] & m) f[1] = 1.0f;
            else if (pos->pieces[!pos->turn] & m) f[2] = 1.0f;
            if (pos->blockers & m) f[3] = 1.0f;
        }
    }
}

/* cpp/self_play_client.cpp:88-107: index x + 7*y with y = 0 at rank 7;
 * 1 = x stone, 2 = o stone, blockers and empties are 0. */
void orc_board_cells(const orc_pos *pos, int32_t *out49)
{
    for (int y = 0; y < 7; y++) {
        for (int x = 0; x < 7; x++) {
            uint64_t m = 1ULL << (x + 7 * (6 - y));
            out49[x + 7 * y] = (pos->pieces[0] & m) ? 1 : ((pos->pieces[1] & m) ? 2 : 0);
        }
    }
}
