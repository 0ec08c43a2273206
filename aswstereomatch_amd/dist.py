"""Process-group plumbing for the multi-GPU bench (one process per GPU).

Frames are independent, so ranks never exchange image data: the process group carries only the
barrier around the timed region and the MAX over ranks of the elapsed time.  backend "nccl" is RCCL on
ROCm; "gloo" is used by the CPU tests.
"""
import os


def env_ranks():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def frames_for_rank(n_frames, rank, world):
    """Global frame indices owned by `rank`: frame i -> rank i % world (same rule as asw_stereo_match_batch)."""
    return list(range(rank, n_frames, world))


class Group:
    def __init__(self, backend=None, device=None):
        self.rank, self.local_rank, self.world = env_ranks()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend=backend or "gloo", rank=self.rank, world_size=self.world, **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def gather_ints(self, value):
        """One integer per rank, in rank order, on every rank (e.g. the ranks' PIDs)."""
        if self.dist is None:
            return [int(value)]
        import torch

        t = torch.zeros(self.world, dtype=torch.int64, device=self.device if self.device is not None else "cpu")
        t[self.rank] = int(value)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [int(v) for v in t.tolist()]

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
