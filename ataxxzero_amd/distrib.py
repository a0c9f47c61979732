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

    def gather(self, values):
        """every rank's list of floats -> [[rank 0's], [rank 1's], ...] on every rank (a few scalars per rank: what the
        aggregate line says about each shard)"""
        values = [float(v) for v in values]
        if self.dist is None:
            return [values]
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.device)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [[float(x) for x in o.tolist()] for o in out]

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


def local_world():
    """ranks on this node (torchrun exports LOCAL_WORLD_SIZE; bench.py's own launcher is single-node)"""
    return int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or "1")


def core_share(local_rank, n_local, cores=None):
    """The host cores of local rank r: the r-th of n_local contiguous, disjoint, equal shares of the cores this process may
    run on (contiguous ids are one socket / L3 neighbourhood on the boxes seen).  With fewer cores than ranks every rank
    keeps them all."""
    cores = sorted(os.sched_getaffinity(0) if cores is None else cores)
    if n_local <= 1 or len(cores) < n_local:
        return cores
    per = len(cores) // n_local
    return cores[local_rank * per:(local_rank + 1) * per]


def pin_host_threads(local_rank, n_local):
    """Confine this rank — and every thread it starts later: the HIP runtime's, the drain's — to its core share, so that N
    ranks' host loops do not migrate over each other (the host side is the only thing the ranks of a node share besides
    the file system: SURVEY.md §8e).  Returns the share.  AZH_NO_PIN=1 leaves the affinity alone."""
    share = core_share(local_rank, n_local)
    if n_local > 1 and not os.environ.get("AZH_NO_PIN"):
        os.sched_setaffinity(0, set(share))
    return share


def pci_to_numbers(bus_id):
    """ "0000:05:00.0" -> [domain, bus, device, function] (numbers travel through the all-gather) """
    try:
        dom, bus, rest = bus_id.strip().split(":")
        dev, fn = rest.split(".")
        return [int(dom, 16), int(bus, 16), int(dev, 16), int(fn, 16)]
    except Exception:
        return [-1, -1, -1, -1]


def numbers_to_pci(v):
    v = [int(x) for x in v]
    return None if v[0] < 0 else "%04x:%02x:%02x.%x" % tuple(v)


def per_rank_report(group, value, ms_per_step, device, bus_id, cores):
    """What the aggregate line carries about each rank: its own rate and step time (a straggler is then visible beside the
    max-over-ranks figure), the device ordinal it drove and that device's PCI bus id (two ranks on one card are then
    visible), the host cores it was confined to."""
    rows = group.gather([value, ms_per_step, device] + pci_to_numbers(bus_id) + [len(cores), cores[0] if cores else -1,
                                                                                 cores[-1] if cores else -1])
    values = [r[0] for r in rows]
    ms = [r[1] for r in rows]
    ids = [numbers_to_pci(r[3:7]) for r in rows]
    real = [i for i in ids if i is not None]
    return {"value": values, "ms_per_step": ms, "device": [int(r[2]) for r in rows], "pci_bus_id": ids,
            "host_cores": ["%d-%d (%d)" % (int(r[8]), int(r[9]), int(r[7])) for r in rows],
            "value_min_max": [min(values), max(values)], "ms_per_step_min_max": [min(ms), max(ms)],
            "ranks_sharing_a_device": len(real) - len(set(real))}
