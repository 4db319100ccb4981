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


class _DeviceFloats:
    """n floats at a raw device address, for torch.as_tensor (device memory plumbing only)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def device_view(ptr, n):
    import torch
    return torch.as_tensor(_DeviceFloats(ptr, n), device="cuda")


class Comm:
    """Owner of a bt_comm (section 7 of include/bullet_amd.h). `Comm.rccl(dist)` is the product path: a native RCCL
    communicator, one rank per process and GPU -- torch.distributed only carries the 128-byte id from rank 0 to the
    others. `Comm.from_torch(dist)` wraps torch.distributed's own collectives as hooks (gloo staged through the host:
    the two-ranks-on-one-GPU tests, where RCCL cannot run)."""

    def __init__(self, handle, rank, world, keep=()):
        self.h, self.rank, self.world, self._keep, self.errors, self._token = handle, rank, world, keep, [], None

    @classmethod
    def rccl(cls, dist=None):
        import ctypes as C
        from . import lib
        L = lib.load()
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
        # every rank says whether it can load librccl BEFORE anyone enters ncclCommInitRank, which blocks until all
        # ranks have joined: one rank without the library would otherwise leave the others waiting there
        ok = bool(L.bt_comm_available())
        if world > 1:
            flags = [None] * world
            dist.all_gather_object(flags, ok)
            ok = all(flags)
        if not ok:
            raise RuntimeError("the native RCCL communicator is not available on every rank: " + L.bt_last_error().decode())
        ident = C.create_string_buffer(128)
        payload, err = None, None
        if rank == 0:
            try:
                lib.check(L.bt_comm_unique_id(ident))
                payload = ident.raw
            except Exception as e:  # the other ranks are waiting in the broadcast: tell them, then raise everywhere
                err = e
        if world > 1:
            box = [payload]
            dist.broadcast_object_list(box, src=0)
            payload = box[0]
        if payload is None:
            raise err or RuntimeError("rank 0 could not create the RCCL unique id")
        ident = C.create_string_buffer(payload, 128)
        h = C.c_void_p()
        try:
            lib.check(L.bt_comm_init_rank(ident, rank, world, C.byref(h)))
        except lib.BtError as e:  # the other ranks are inside ncclCommInitRank: no collective fallback from here
            raise SystemExit(f"rank {rank}: bt_comm_init_rank failed after every rank had agreed to use RCCL: {e}")
        return cls(h, rank, world)

    @classmethod
    def from_torch(cls, dist, own_stream=False):
        import ctypes as C
        import torch
        from . import lib
        L = lib.load()
        rank, world = dist.get_rank(), dist.get_world_size()
        staged = dist.get_backend() == "gloo"
        self = cls(None, rank, world)

        def guarded(fn):
            def call(*a):
                try:
                    # (the NULL stream arrives as None)
                    with torch.cuda.stream(torch.cuda.ExternalStream(a[-1]) if a[-1] else torch.cuda.default_stream()):
                        fn(*a[1:-1])
                    return 0
                except Exception as e:  # an exception must not unwind through the C frames
                    self.errors.append(e)
                    return 1
            return call

        def all_reduce(buf, n):
            t = device_view(buf, n)
            if staged:
                host = t.cpu()
                dist.all_reduce(host)
                t.copy_(host)
            else:
                dist.all_reduce(t)

        def reduce_scatter(send, recv, n):
            s, r = device_view(send, n * world), device_view(recv, n)
            if staged:
                host = s.cpu()
                dist.all_reduce(host)
                r.copy_(host[rank * n:(rank + 1) * n])
            else:
                dist.reduce_scatter_tensor(r, s)

        def all_gather(send, recv, n):
            s, r = device_view(send, n), device_view(recv, n * world)
            if staged:
                parts = [torch.empty(n) for _ in range(world)]
                dist.all_gather(parts, s.cpu())
                r.copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(r, s)

        hooks = lib.CommHooks(lib.COMM_ALLREDUCE_CB(guarded(all_reduce)), lib.COMM_SCATTER_CB(guarded(reduce_scatter)),
                              lib.COMM_SCATTER_CB(guarded(all_gather)), None)
        h = C.c_void_p()
        lib.check(L.bt_comm_from_hooks(C.byref(hooks), rank, world, C.byref(h)))
        if own_stream:  # the hooks then run on the communicator's stream, event-fenced against the step's (bt_net_dp_step)
            lib.check(L.bt_comm_use_own_stream(h))
        self.h, self._keep = h, (hooks,)
        return self

    def all_reduce(self, tensor, stream=None):
        """in-place sum of a float32 device tensor over the ranks (bt_allreduce_sum_f32)"""
        import torch
        from . import lib
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        lib.check(lib.load().bt_allreduce_sum_f32(self.h, s, tensor.data_ptr(), tensor.numel()))

    def barrier(self, stream=None):
        """every rank has reached this point (a 1-float all-reduce on `stream`, waited for)"""
        import torch
        from . import lib
        if self._token is None:
            self._token = torch.zeros(1, device="cuda")
            torch.cuda.synchronize()
        self.all_reduce(self._token, stream)
        if stream is None:
            torch.cuda.current_stream().synchronize()
        else:
            lib.check(lib.load().bt_stream_sync(stream))

    def close(self):
        if self.h:
            from . import lib
            lib.load().bt_comm_destroy(self.h)
            self.h = None
