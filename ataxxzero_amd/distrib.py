"""One process per GPU, no data-path collective: self-play games are independent, so the
N ranks of a node each own `games_per_gpu` game slots (weak scaling) and only meet for
the timing barrier and the max-over-ranks reduction of the measured time.

torch.distributed is plumbing here (backend "nccl" = RCCL when the ranks hold GPUs, "gloo"
otherwise); with a single rank nothing is initialised at all.
"""
import os


def world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


class Group:
    def __init__(self, backend=None):
        self.rank, self.local_rank, self.world = world()
        self.dist = None
        self.device = None
        self.backend = None
        # AZH_DIST_FORCE=1: build the process group even for one rank, to rehearse the N > 1 order of initialisation
        # (torch's HIP runtime and RCCL first, then this package's library) on a one-GPU box
        if self.world > 1 or os.environ.get("AZH_DIST_FORCE"):
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            if backend is None:
                # AZH_DIST_BACKEND=gloo: rehearse several ranks on one GPU (RCCL refuses duplicate devices)
                backend = os.environ.get("AZH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            self.backend = backend
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                # rendezvous on the device right away: a lazily initialised communicator would
                # otherwise be built inside the first timed barrier
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world,
                                        device_id=self.device)
                dist.barrier()
                self.dist = dist
                self.torch = torch
                return
            else:
                self.device = torch.device("cpu")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.dist = dist
            self.torch = torch

    def barrier(self):
        if self.dist is not None:
            if self.device.type == "cuda":
                self.torch.cuda.synchronize()
            self.dist.barrier()

    def reduce(self, value, op):
        """op in {"max", "sum"} over ranks; returns a python float on every rank."""
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def device_for(local_rank, device_count):
    """One process per GPU: local rank r drives GPU r (AZH_DEVICE_MOD=1 folds the ranks onto the
    available devices, to rehearse N ranks on a smaller box)."""
    if os.environ.get("AZH_DEVICE_MOD") and device_count > 0:
        return local_rank % device_count
    return local_rank


def shard_seed(base_seed, rank):
    """Distinct Philox stream per rank."""
    return int(base_seed) + int(rank)


def aggregate(group, units_local, seconds_local):
    """Whole-job throughput: units all ranks processed / max-over-ranks time."""
    total = group.reduce(units_local, "sum")
    t = group.reduce(seconds_local, "max")
    return total, t, (total / t if t > 0 else 0.0)
