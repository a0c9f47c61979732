"""Batched head-to-head match between two nets on one GPU (BASELINE config 5).

Restates what uai_ringmaster.py does with two `uai_interface.py --network-path X --visits V`
engine subprocesses (uai_ringmaster.py:75-160,198-265), as one device-resident batch:
  * no-blocker start (ataxx_rules.py:44-50), engine 1 ("white") moves first
  * each side searches exactly `visits` MCTS steps per move from a fresh tree (the
    per-ply `moves` message makes engine.set_state rebuild it, engine.py:452-472), no
    Dirichlet noise, python posterior and tie-break (engine.py:197-203,291)
  * the move is sampled ~ (n/N)^5 over edges with n >= max/2 (engine.py:532-548)
  * a single legal move is played without search (uai_ringmaster.py:114-116)
  * every pairing is played both ways (uai_ringmaster.py:241-247): even game slots give x to
    net A, odd slots to net B
  * result 1/2 scores a win for that colour's engine; a game cut at --max-plies is
    "invalid": half a point each and counted as annulled (uai_ringmaster.py:251-257)
"""
import datetime
import json

from . import link, model, selfplay


def random_openings(pairs, depth, seed):
    """`pairs` openings of `depth` uniformly random plies from the start position — uai_ringmaster.get_opening
    (uai_ringmaster.py:185-196: OPENING_DEPTH random.choice(board.legal_moves()) moves; the constant is 0 in the reference's
    file and is edited there) — played by the HIP playout kernel.  -> (positions (pairs, 2) u64 packed x | turn << 63, o;
    moves: per pairing the list of UAI strings).  Games that end inside the opening are not used."""
    import numpy as np
    x, o, bl, turn = selfplay.parse_fen(selfplay.START_FEN_PLAIN)
    boards, moves = [], []
    batch = 0
    while len(boards) < pairs:
        plies, results, trace, mv = link.random_play(2 * pairs + 16, (int(seed) << 8) + batch, x, o, bl, turn, max(depth + 1, 2))
        batch += 1
        for g in range(len(plies)):
            # still running after `depth` plies, and no side had to pass on the way (a pass inside the opening would
            # shift the mover's parity under the PGN replay; the reference's get_opening plays legal_moves(), which at
            # these depths never holds a pass either)
            if len(boards) < pairs and plies[g] > depth and not (mv[g, :depth] == 0xFFFF).any():
                bx, bo = int(trace[g, depth, 0]) & ((1 << 63) - 1), int(trace[g, depth, 1])
                boards.append([bx | ((depth & 1) << 63), bo])
                moves.append([uai_move(int(m)) for m in mv[g, :depth]])
    return np.array(boards, dtype=np.uint64), moves


def uai_move(mv):
    """u16 (from | to << 8, square = file + 7 * rank) -> UAI text (uai_interface.py:11-22)"""
    def sq(s):
        return "abcdefg"[s % 7] + str(s // 7 + 1)
    if mv == 0xFFFF:
        return "0000"          # a pass (uai_interface.py:12-13); the playout kernel's marker
    frm, to = mv & 0xFF, mv >> 8
    return sq(to) if frm == to else sq(frm) + sq(to)


class Match:
    def __init__(self, weights_a, weights_b, visits, games=1024, dtype="f16", seed=selfplay.DEFAULT_SEED,
                 max_plies=400, opening_depth=0):
        if games % 2:
            raise ValueError("the number of concurrent games must be even (each pairing is played both ways)")
        if not 0 <= opening_depth < max_plies:
            raise ValueError("opening_depth must be in 0 .. max_plies - 1 (%d), got %d" % (max_plies - 1, opening_depth))
        self.net_a = link.Net(*weights_a, model.BN_EPSILON)
        self.net_b = link.Net(*weights_b, model.BN_EPSILON)
        self.dtype = link.DTYPES[dtype]
        cfg = selfplay.make_config(games, visits, seed=seed, fen=selfplay.START_FEN_PLAIN, max_plies=max_plies,
                                   dirichlet_weight=0.0, flags=link.FLAG_ARENA)
        self.engine = link.Engine(cfg)
        self.games = games
        self.opening_depth = opening_depth
        self.openings = None
        self.opening_boards = None   # (pairs, 2) u64: the position after each pairing's opening
        self.limit = None        # set_game_limit: the size of the match
        self.finished = 0        # games handed out so far
        self.thin = False        # the tower runs one board per workgroup (the match's last games)
        if opening_depth > 0:
            # every pairing gets its own random opening, played both ways (uai_ringmaster.py:242-246): slots 2k and 2k + 1
            # start from the same position.  Only the FIRST game of a slot starts from it (uid < games): a match from
            # openings is a cohort of at most `games` games.
            import numpy as np
            boards, self.openings = random_openings(games // 2, opening_depth, seed)
            self.opening_boards = boards
            self.engine.set_positions(np.repeat(boards, 2, axis=0), np.full(games, opening_depth, dtype=np.int32))

    def set_game_limit(self, games):
        """The match is the games with uid < `games` (slot g plays uids g, g + concurrent, ...): a slot whose next game would be
        past the limit goes idle instead of starting a game nobody scores, so the batch thins out as the match ends and its
        last, longest games run at the latency of a nearly empty batch (azh_engine_set_game_limit)."""
        self.engine.set_game_limit(games)
        self.limit = games
        self._pick_tower()

    def _pick_tower(self):
        """The match's last games — at most link.THIN_MAX_GAMES left of a match that started with more in flight — are a
        handful of leaves per iteration, whose cost is one workgroup's time for the whole net: the tower then runs one board
        per workgroup (azh_engine_set_thin_batches).  Decided by the count of games handed out, at a drain or when the limit
        moves: the same point in every run of the same match (the two 16-bit kernels differ in the last bits).  A limit that
        is RAISED past that point (uai_ringmaster.py keeps a branch for callers that do) brings the throughput kernel back."""
        thin = self.limit is not None and self.limit - self.finished <= link.THIN_MAX_GAMES < self.games
        if thin != self.thin:
            self.engine.set_thin_batches(1 if thin else -1)
            self.thin = thin

    def run(self, iterations):
        self.engine.run_arena(self.net_a, self.net_b, iterations, self.dtype)

    def fetch(self):
        """Waits for the iterations enqueued so far and takes their finished games off the device; a `run` enqueued between
        this and `drain` executes while the host parses and scores them."""
        self.engine.fetch()

    def drain(self):
        """Finished games: dicts with moves (UAI strings), result (1, 2 or 0 = cut), white ('a'/'b'),
        final_score (x stones, o stones)."""
        out = []
        for line in self.engine.drain_json():
            e = json.loads(line)
            white = "a" if e["slot"] % 2 == 0 else "b"
            opening = []
            if self.openings is not None and e["uid"] < self.games:
                opening = self.openings[e["uid"] // 2]
                e["moves"] = opening + e["moves"]          # (boards[] starts after the opening; only its last entry is used)
            out.append({"moves": e["moves"], "result": e["result"], "white": white, "uid": e["uid"],
                        "final_score": replay_final_score(e), "boards": e["boards"], "opening": opening})
        self.finished += len(out)
        self._pick_tower()
        return out

    def lost_games(self):
        """Why the match can no longer be completed, or None.  A cohort under a game limit ends when every one of its games
        has been handed out; a record that did not fit the device's ring (the host did not drain for far too long) never
        comes, and a loop `while handed_out < cohort` would then enqueue search for an idle engine for ever.  Call right
        after fetch(): the engine's streams are idle there (the call costs one small kernel and waits for nothing), and
        every game the counters know of has been taken off the device — so after the drain that follows, a game the engine
        has ended and the host has not seen is a lost one."""
        st = self.engine.stats()
        if st["ring_overflow"] > 0:
            return "%d finished games did not fit the device's record ring and are lost" % st["ring_overflow"]
        if self.limit is not None and st["games"] + st["dropped"] >= self.limit:
            return "the engine has ended all %d games of the match" % self.limit   # (a reason only if some were not handed out)
        return None

    def close(self):
        self.engine.close()
        self.net_a.close()
        self.net_b.close()


def replay_final_score(entry):
    """Stone counts after the last move: boards[] holds the position BEFORE each move, so the last
    move is applied to the last board (clone/jump + capture of the 8 neighbours)."""
    if not entry["boards"]:
        return (2, 2)
    cells = list(entry["boards"][-1])
    mv = entry["moves"][-1]
    mover = 1 if (len(entry["moves"]) - 1) % 2 == 0 else 2

    def sq(s):
        return (ord(s[0]) - 97, 7 - int(s[1]))

    if len(mv) == 4:
        fx, fy = sq(mv[:2])
        cells[fx + 7 * fy] = 0
        tx, ty = sq(mv[2:])
    else:
        tx, ty = sq(mv)
    cells[tx + 7 * ty] = mover
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            x, y = tx + dx, ty + dy
            if 0 <= x < 7 and 0 <= y < 7 and cells[x + 7 * y] not in (0, mover):
                cells[x + 7 * y] = mover
    return (cells.count(1), cells.count(2))


def write_game_to_pgn(path, game, white_name, black_name, round_index, tc):
    """uai_ringmaster.write_game_to_pgn (:162-180), same tags and layout."""
    now = datetime.datetime.now()
    with open(path, "a+") as f:
        print('[Event "?"]', file=f)
        print('[Site "?"]', file=f)
        print('[Date "%s"]' % (now.strftime("%Y.%m.%d"),), file=f)
        print('[Round "%i"]' % (round_index,), file=f)
        print('[White "%s"]' % (white_name,), file=f)
        print('[Black "%s"]' % (black_name,), file=f)
        print('[Opening "%s"]' % (", ".join(game.get("opening", [])),), file=f)
        print('[GameStartTime "%s"]' % (now.isoformat(),), file=f)
        print('[GameEndTime "%s"]' % (now.isoformat(),), file=f)
        print('[Plycount "%i"]' % (len(game["moves"]),), file=f)
        result_string = {1: "1-0", 2: "0-1", 0: "1/2-1/2"}[game["result"]]
        print('[Result "%s"]' % (result_string,), file=f)
        print('[FinalScore "%i-%i"]' % tuple(game["final_score"]), file=f)
        print('[TimeControl "+%r"]' % (tc,), file=f)
        print(file=f)
        print(" ".join(game["moves"]), file=f)
        print(file=f)
