"""Data-parallel plumbing for the training step: one process per GPU, every rank holds a
full replica and a shard of the global batch; the flat gradient (unscaled batch sums, run.rs:130) and the loss
sum are all-reduced (RCCL over xGMI on GPUs: torch.distributed backend "nccl"; "gloo" on CPU for tests), then
every rank applies the same AdamW step with adj = 1 / global_batch, so replicas stay bit-identical.
The reference itself is single-device (no collective anywhere in the tree); this is new functionality."""
import os


def env_world():
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when not launched by it."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n positions for `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_step_buffers(dist, step_buffer):
    """Sum the step buffer over ranks (in place). The step buffer is the flat gradient with the loss sum stored
    right behind it (`Net.step_buffer`), so a training step needs ONE collective: config 2 is a single 3.2 MB
    message; a second 4-byte all-reduce would cost a full ring latency for nothing."""
    if dist is None:
        return
    if step_buffer.is_cuda and dist.get_backend() == "gloo":  # CPU collective (tests / rehearsal): through the host
        host = step_buffer.cpu()
        dist.all_reduce(host)
        step_buffer.copy_(host)
        return
    dist.all_reduce(step_buffer)


def adj_factor(global_batch):
    """gradient_factor of run.rs:130 for the GLOBAL batch"""
    return 1.0 / float(global_batch)
