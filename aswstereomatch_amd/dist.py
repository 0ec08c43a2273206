"""Process-group plumbing for the multi-GPU bench (one process per GPU).

Frames are independent, so ranks never exchange image data: the process group carries only the
barrier around the timed region and the MAX over ranks of the elapsed time.  backend "nccl" is RCCL on
ROCm; "gloo" is used by the CPU tests.
"""
import contextlib
import os
import sys


@contextlib.contextmanager
def _stdout_to_stderr():
    """File descriptor 1 -> 2 for the duration: the communication libraries print banners on stdout while a process group
    is built (RCCL's version block under NCCL_DEBUG=VERSION, gloo's "Rank 0 is connected ..."), and rank 0's stdout is
    the bench's one JSON line.  C stdio is flushed inside the redirection so that nothing surfaces later."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


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
        self.backend = "none"
        self.requested_backend = backend or "gloo"
        self.rccl_error = None  # first line of the error that made an "nccl" group fall back to gloo
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            with _stdout_to_stderr():
                self.backend = backend or "gloo"
                kw = {}
                if self.backend == "nccl" and device is not None:
                    kw["device_id"] = device  # eager communicator on this rank's own GPU: a failure shows here, on every rank
                try:
                    dist.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world, **kw)
                except Exception as e:  # noqa: BLE001 -- RCCL could not build the communicator (it fails on all ranks alike)
                    if self.backend != "nccl":
                        raise
                    # The group carries a barrier and two scalar all-reduces, never image data: gloo over the loopback does
                    # that as well.  A second rendezvous needs its own port (rank 0's first store may still hold the old one).
                    self.rccl_error = (str(e).splitlines() or [type(e).__name__])[0][:300]
                    print("dist: RCCL process group failed (%s); falling back to gloo for the barrier / timing reductions" % self.rccl_error,
                          file=sys.stderr, flush=True)
                    try:
                        dist.destroy_process_group()
                    except Exception:  # noqa: BLE001
                        pass
                    os.environ["MASTER_PORT"] = str(int(os.environ["MASTER_PORT"]) + 1)
                    self.backend = "gloo"
                    self.device = None
                    dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
                dist.barrier()
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
