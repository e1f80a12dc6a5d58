"""Multi-GPU plumbing of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in
the CPU tests).  Two modes:

  replicas : the path shards by independent request batches -- every rank owns a full copy of the tables (1.4 / 15 / 63 GB
             all fit in 288 GB) and its own request stream; NO data-path collective.  This is the reference's own model
             (private weights and buffers per stream, cuda_server.c:136-183).  Only the timing barrier and the max-over-ranks
             reduction touch the network.
  sharded  : tables partitioned by table-ID (contiguous runs of record segments) over the ranks; each rank gathers its
             slice [B x F] and ONE RCCL all-gather over xGMI rebuilds the records; rank r then runs the FC chain on its
             B/G items.  Used when the tables exceed one GPU's HBM (BASELINE configs 4/5).
"""
import os

import numpy as np


class DistEnv:
    def __init__(self, backend=None, force_group=False):
        """force_group: build the process group even for ONE rank (a one-GPU rehearsal of the RCCL code path: every collective of the
        N > 1 line really goes through torch.distributed / RCCL on device tensors, degenerate as it is)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = backend
        self.dist = None
        self.torch = None
        if self.world > 1 or (force_group and backend):
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", str(self.rank))            # (a forced one-rank group: the env:// rendezvous still wants them)
            os.environ.setdefault("WORLD_SIZE", str(self.world))
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            self.backend = backend
            if backend == "nccl":
                # librccl prints a version banner ("RCCL version : ...", five lines) to STDOUT when its first communicator comes up; rank 0's
                # stdout is the bench contract's ONE JSON line.  The communicator is brought up here, with fd 1 pointed at stderr meanwhile.
                import sys
                sys.stdout.flush()
                saved = os.dup(1)
                os.dup2(2, 1)
                try:
                    torch.cuda.set_device(self.local_rank)
                    dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
                    dist.barrier()
                    torch.cuda.synchronize()
                finally:
                    sys.stdout.flush()
                    os.dup2(saved, 1)
                    os.close(saved)
            else:
                dist.init_process_group(backend)

    @property
    def device(self):
        if self.torch is not None and self.backend == "nccl":
            return self.torch.device("cuda", self.local_rank)
        return "cpu"

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        """MAX-reduce a python float over all ranks (the bench contract's timing rule)."""
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def all_gather_slices(self, local, out=None):
        """local: torch tensor [B, F] (same shape on every rank) -> [world, B, F]: the ONE exchange step of the sharded
        mode (ncclAllGather over xGMI when the backend is nccl)."""
        if self.dist is None:  # one rank: the "exchange" is the identity -- but the caller's buffer must still receive it
            return self._single_rank(local, out)
        if out is None:
            out = self.torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        # the concatenated form [world*B, F] is what both the nccl and the gloo back-ends accept
        self.dist.all_gather_into_tensor(out.view((self.world * local.shape[0],) + tuple(local.shape[1:])), local.contiguous())
        return out

    def all_to_all_slices(self, local, out=None):
        """local: torch tensor [B, F], B divisible by world.  Rank r keeps only the items it will run the FC chain on:
        it sends rows [j*B/G, (j+1)*B/G) of its slice to rank j and receives every shard's slice of ITS items
        -> [world, B/world, F].  1/world of the all-gather's traffic."""
        B = local.shape[0]
        if self.dist is None:
            return self._single_rank(local, out)
        assert B % self.world == 0, "all-to-all exchange needs the batch divisible by the number of shards"
        if out is None:
            out = self.torch.empty((self.world, B // self.world) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        self.dist.all_to_all_single(out.view((B,) + tuple(local.shape[1:])), local.contiguous())
        return out

    @staticmethod
    def _single_rank(local, out):
        """world == 1: [B, F] -> [1, B, F]; when the caller supplies `out`, the slice is copied INTO it (the caller keeps
        reading its own buffer), otherwise a view of `local` is returned."""
        shape = (1,) + tuple(local.shape)
        if out is None:
            return local.reshape(shape)
        dst = out.reshape(shape)
        if hasattr(dst, "copy_"):
            dst.copy_(local.reshape(shape))      # torch
        else:
            dst[...] = local.reshape(shape)      # numpy
        return out

    def min_over_ranks_int(self, values):
        """Element-wise MIN of a short list of python ints over all ranks (fp8 chain: every rank must quantise X, R1, R2, R3
        with the SAME power-of-two exponents, or slices encoded by one rank are decoded with another rank's scale)."""
        if self.dist is None:
            return [int(v) for v in values]
        t = self.torch.tensor([int(v) for v in values], dtype=self.torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return [int(v) for v in t.tolist()]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def replica_seed(base_seed, rank):
    """Every replica draws its own request stream."""
    return base_seed + 1000003 * rank


def aggregate_throughput(items_per_rank, elapsed_max_s, world):
    """value = the units ALL ranks processed / the max-over-ranks time (bench contract)."""
    return world * items_per_rank / elapsed_max_s


def item_range(rank, world, batch):
    """Sharded mode: the items of a batch whose FC chain rank r computes (contiguous, near-equal)."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def assemble_records(gathered, slice_offsets, slice_lens, record_len):
    """[G][B][F] all-gathered padded slices -> item-major records [B][K] (numpy; the checker-side restatement of what
    the device's slice transpose does)."""
    g = np.asarray(gathered)
    G, B, F = g.shape
    out = np.zeros((B, record_len), dtype=g.dtype)
    for s in range(G):
        out[:, slice_offsets[s]:slice_offsets[s] + slice_lens[s]] = g[s, :, :slice_lens[s]]
    return out
