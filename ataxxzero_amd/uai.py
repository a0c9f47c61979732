"""Single-position search for the UAI front-end: what engine.MCTSEngine.genmove does
(engine.py:474-530) with the tree, the search and the net on the GPU.

One game slot, the Python engine's search semantics (first-max ties, 833-way posterior, no
Dirichlet: AZH_FLAG_TIE_FIRST | AZH_FLAG_PY_POSTERIOR), a fresh tree per call — which is what
the reference engine ends up with when a UAI master sends one `moves` message per ply
(engine.set_state, engine.py:452-472).  With `visits` the search is exactly `visits` MCTS steps
(uai_interface.py:44-46); with a time budget it runs until the time is used.  The move is then
sampled on the host exactly like sample_with_exponential_weight (engine.py:532-548).
"""
import random
import time

import numpy as np

from . import link, model, selfplay

FILES = "abcdefg"


def encode_square(sq):
    return "%s%i" % (FILES[sq % 7], sq // 7 + 1)


def encode_move(mv):
    """u16 from | to << 8 -> UAI string (uai_interface.py:11-17); 0xFFFF -> "0000"."""
    if mv == 0xFFFF:
        return "0000"
    frm, to = mv & 0xFF, mv >> 8
    return encode_square(to) if frm == to else encode_square(frm) + encode_square(to)


def decode_move(s):
    """UAI string -> u16 (uai_interface.py:24-32)."""
    if s in ("pass", "none", "0000"):
        return 0xFFFF
    sq = lambda t: FILES.index(t[0].lower()) + 7 * (int(t[1]) - 1)
    if len(s) == 2:
        return sq(s) | (sq(s) << 8)
    if len(s) == 4:
        return sq(s[:2]) | (sq(s[2:]) << 8)
    raise Exception("Bad UAI move string: %r" % s)


class Position:
    """Board bookkeeping of the front-end, with the GPU rules behind it."""

    def __init__(self, x, o, turn):
        self.x, self.o, self.turn = x, o, turn

    @staticmethod
    def initial():
        x, o, _, turn = selfplay.parse_fen(selfplay.START_FEN_PLAIN)
        return Position(x, o, turn)

    @staticmethod
    def from_fen(fen):
        x, o, bl, turn = selfplay.parse_fen(fen)
        return Position(x, o, turn)

    def fen(self):
        """FEN in the reference's dialect (ataxx_rules.py fen(): ranks 7..1, x/o/digits, side to move)."""
        rows = []
        for r in range(6, -1, -1):
            row, run = "", 0
            for f in range(7):
                bit = 1 << (f + 7 * r)
                ch = "x" if self.x & bit else ("o" if self.o & bit else None)
                if ch is None:
                    run += 1
                else:
                    row += (str(run) if run else "") + ch
                    run = 0
            rows.append(row + (str(run) if run else ""))
        return "/".join(rows) + (" x" if self.turn == 0 else " o")

    def packed(self):
        return link.pack_board(self.x, self.o, self.turn)

    def legal_moves(self):
        moves, counts, results = link.rules_batch(self.packed().reshape(1, 2), 0)
        return [int(m) for m in moves[0, :counts[0]]], int(results[0])

    def move(self, mv):
        out = link.makemove_batch(self.packed().reshape(1, 2), np.array([mv], dtype=np.uint16))[0]
        self.x, self.o, self.turn = int(out[0]) & ~(1 << 63), int(out[1]), int(out[0]) >> 63

    def __str__(self):
        rows = []
        for r in range(6, -1, -1):
            rows.append(" ".join("X" if (self.x >> (f + 7 * r)) & 1 else ("O" if (self.o >> (f + 7 * r)) & 1 else ".")
                                 for f in range(7)))
        return "\n".join(rows)


class Searcher:
    TIME_CAP_VISITS = 20000  # arena size of a time-controlled search

    def __init__(self, network_path, dtype="f16", symmetry_average=False):
        # symmetry_average: every evaluation is nn_evals.evaluate (nn_evals.py:48-62); with one game the
        # eight images ride in the same tower launch, so it costs no time
        self.extra_flags = link.FLAG_SYMMETRY_AVG if symmetry_average else 0
        conv, bn = model.load_model(network_path)
        self.net = link.Net(conv, bn, model.BN_EPSILON)
        self.dtype = link.DTYPES[dtype]
        self.last_steps = 0
        self.last_seconds = 0.0

    def root_visits(self, pos, visits=None, seconds=None):
        """-> [(move u16, visits)] over the expanded root edges after the search."""
        cap = visits if visits is not None else self.TIME_CAP_VISITS
        cfg = link.Config(games=1, visits=cap + 1, max_plies=400, edges_per_node=96, c_puct=1.0, dirichlet_alpha=0.15,
                          dirichlet_weight=0.0, start_turn=pos.turn, seed=random.getrandbits(63), start_x=pos.x,
                          start_o=pos.o, blockers=0,
                          flags=link.FLAG_TIE_FIRST | link.FLAG_PY_POSTERIOR | self.extra_flags)
        eng = link.Engine(cfg)
        start = time.time()
        try:
            if visits is not None:
                eng.run(self.net, 1 + visits, self.dtype)  # root evaluation + `visits` steps
                steps = visits
            else:
                eng.run(self.net, 2, self.dtype)
                steps = 1
                while time.time() - start < seconds and steps + 64 < cap:
                    eng.run(self.net, 64, self.dtype)
                    eng.sync()
                    steps += 64
            eng.sync()
            boards, info, edges, moves = eng.tree(0)
        finally:
            eng.close()
        self.last_steps, self.last_seconds = steps, max(time.time() - start, 1e-9)
        first, n = int(info[0, 0]), int(info[0, 1] & 0xFFFF)
        return [(int(moves[first + j]), int(edges[first + j, 1])) for j in range(n) if int(edges[first + j, 3]) != 0xFFFFFFFF]

    def genmove(self, pos, visits=None, seconds=None, exponent=5.0):
        legal, result = pos.legal_moves()
        if not legal:
            return 0xFFFF  # the reference answers "pass" when no edge was visited (engine.py:495-496)
        edges = self.root_visits(pos, visits=visits, seconds=seconds)
        if not edges:
            return legal[0]
        # sample_with_exponential_weight (engine.py:532-548)
        total = float(sum(n for _, n in edges))
        max_visits = max(n for _, n in edges)
        weights = {mv: (n / total) ** exponent for mv, n in edges if n >= max_visits * 0.5}
        norm = 1.0 / sum(weights.values())
        x = random.random()
        for mv, w in weights.items():  # sample_by_weight (engine.py:38-47)
            if x <= w * norm:
                return mv
            x -= w * norm
        return next(iter(weights))


# ---------------------------------------------------------------- text protocol

def xy_to_square(xy):
    """(x, y) with y = 0 at rank 7 (the reference's board coordinates) -> "a7"-style square."""
    return FILES[xy[0]] + str(7 - xy[1])


def square_to_xy(text):
    return FILES.index(text[0].lower()), 7 - int(text[1])


def xy_move_to_text(move):
    """Reference move value ("pass" | ("c", (x, y)) | ((x0, y0), (x1, y1))) -> UAI text."""
    if move == "pass":
        return "0000"
    return "".join(xy_to_square(part) for part in move if part != "c")


def text_to_xy_move(text):
    if text in ("pass", "none", "0000"):
        return "pass"
    if len(text) not in (2, 4):
        raise Exception("Bad UAI move string: %r" % (text,))
    squares = [square_to_xy(text[k:k + 2]) for k in range(0, len(text), 2)]
    return ("c", squares[0]) if len(squares) == 1 else (squares[0], squares[1])


class Session:
    """One UAI dialogue (the command set of uai_interface.py:41-88) over a Searcher.  Commands are looked up in a
    table of (prefix, handler) pairs; every handler returns the lines to print."""

    def __init__(self, searcher, visits=None, safety_ms=0, show_game=False, log=None):
        self.searcher, self.visits, self.safety_ms, self.show_game, self.log = searcher, visits, safety_ms, show_game, log
        self.position = Position.initial()
        self.table = [
            ("uainewgame", self.on_newgame),
            ("uai", self.on_hello),
            ("isready", lambda rest: ["readyok"]),
            ("moves ", self.on_moves),
            ("position fen ", self.on_fen),
            ("go movetime ", self.on_go),
            ("showboard", lambda rest: str(self.position).split("\n") + ["boardok"]),
        ]

    def on_hello(self, rest):
        return ["id name AtaxxZero-MI355X", "id author ataxxzero_amd", "uaiok"]

    def on_newgame(self, rest):
        self.position = Position.initial()
        return []

    def on_moves(self, rest):
        for text in rest.split():
            self.position.move(decode_move(text))
        return []

    def on_fen(self, rest):
        self.position = Position.from_fen(rest)
        if self.show_game and self.log is not None:
            print("===\n%s" % (self.position,), file=self.log)
        return []

    def on_go(self, rest):
        if self.visits is not None:
            move = self.searcher.genmove(self.position, visits=self.visits)
        else:
            budget_ms = max(int(rest) - self.safety_ms, 1)
            move = self.searcher.genmove(self.position, seconds=budget_ms * 1e-3)
        speed = self.searcher.last_steps / self.searcher.last_seconds
        return ["info speed %f nps" % (speed,), "bestmove %s" % (encode_move(move),)]

    def handle(self, line):
        """-> (lines to print, keep going)."""
        if line == "quit":
            return [], False
        for prefix, handler in self.table:
            exact = not prefix.endswith(" ")
            if (line == prefix) if exact else line.startswith(prefix):
                return handler(line[len(prefix):]), True
        return [], True  # unknown commands are ignored, as the reference does

    def serve(self, source, sink):
        for raw in source:
            out, more = self.handle(raw.rstrip("\n"))
            for text in out:
                print(text, file=sink)
            sink.flush()
            if not more:
                break
