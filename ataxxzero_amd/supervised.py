"""Teacher ("supervised") game generation: generate_games.py --supervised CMD (generate_games.py:26-34,
45-49,56-57): an external UAI engine picks the training move of every ply; the move actually played is
replaced by a uniformly random legal one with the opening-randomisation schedule's probability
(generate_games.py:11-14).  Entries carry boards + moves (engine move format, no "dists"), which
train.py turns into one-hot policy targets (train.py:64-65).

The rules run on the GPU (rules entry points of the C ABI); the teacher is any UAI program — for
instance this repo's own `uai_interface.py`, which searches on the GPU.
"""
import random
import subprocess

from . import selfplay, uai

MAXIMUM_GAME_PLIES = 400
OPENING_RANDOMIZATION_SCHEDULE = [0.2 * (0.5 ** (i / 2)) for i in range(10)]  # generate_games.py:11-14


class Teacher:
    """The master side of a UAI dialogue with an engine subprocess (what uai_ringmaster.py:9-60 does for the
    reference): handshake on start, `position fen` + `go movetime` per move, `quit` on exit."""

    HANDSHAKE = ("uai", "setoption name Hash value 1024", "isready", "uainewgame")

    def __init__(self, command):
        self.command = command
        self.process = subprocess.Popen(command, shell=isinstance(command, str), stdin=subprocess.PIPE,
                                        stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        self.tell(*self.HANDSHAKE)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.quit()

    def tell(self, *lines):
        self.process.stdin.write(("\n".join(lines) + "\n").encode("utf8"))
        self.process.stdin.flush()

    def quit(self):
        try:
            self.tell("quit")
        except (BrokenPipeError, OSError):
            pass
        try:
            self.process.wait(timeout=5)
        except subprocess.TimeoutExpired:
            self.process.kill()
            self.process.wait()

    def set_state(self, position):
        self.tell("position fen " + position.fen())

    def genmove(self, ms=1000):
        """-> u16 move; raises when the engine closes its output without a bestmove."""
        self.tell("go movetime %i" % ms)
        for raw in iter(self.process.stdout.readline, b""):
            words = raw.decode("utf8").split()
            if words and words[0] == "bestmove":
                return uai.decode_move(words[1])
        raise Exception("Bad UAI!")


UAIPlayer = Teacher  # the reference's name for it


def generate_game(teacher, ms, rng=random):
    """One teacher game -> entry {"boards", "moves", "result"} (result None when cut at 400 plies)."""
    board = uai.Position.initial()
    entry = {"boards": [], "moves": []}
    result = 0
    for ply in range(MAXIMUM_GAME_PLIES):
        legal, result = board.legal_moves()
        if result != 0:
            break
        teacher.set_state(board)
        training_move = selected_move = teacher.genmove(ms)
        if training_move != 0xFFFF and training_move not in legal:
            raise ValueError("teacher played an illegal move %s in %s" % (uai.encode_move(training_move), board.fen()))
        probability = OPENING_RANDOMIZATION_SCHEDULE[ply] if ply < len(OPENING_RANDOMIZATION_SCHEDULE) else 0.0
        if rng.random() < probability:
            selected_move = rng.choice(legal) if legal else 0xFFFF
        entry["boards"].append(selfplay.board_cells(board.x, board.o))
        entry["moves"].append(selfplay.python_move(training_move))
        board.move(selected_move)
    else:
        _, result = board.legal_moves()
    entry["result"] = result if result else None
    return entry
