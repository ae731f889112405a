"""The reference's `distributed` package surface (/root/reference/distributed/__init__.py:1-13) on RCCL.

Same eleven public names with the same argument meaning: get_rank, get_local_rank, is_primary, synchronize,
get_world_size, all_reduce, all_gather, reduce_dict, data_sampler, LOCAL_PROCESS_GROUP, launch.  On ROCm the
torch.distributed backend string "nccl" IS RCCL; one process per GPU.

Bring-up is environment-first: a process started by `python -m torch.distributed.run` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT) calls bringup() -- or launch(), which notices the environment -- and gets
device binding + the RCCL group in one place.  launch() without such an environment starts the per-GPU processes
itself (the reference's behaviour, distributed/launch.py:22-49) by exporting the same variables to each child
and sending it through the same bringup().
"""
import os
import socket

import torch
from torch import distributed as dist

LOCAL_PROCESS_GROUP = None   # kept for name parity (launch.py:85-90); get_local_rank() reads LOCAL_RANK first

_ENV_KEYS = ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def _group_up():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if _group_up() else 0


def get_world_size():
    return dist.get_world_size() if _group_up() else 1


def is_primary():
    return get_rank() == 0


def get_local_rank():
    if not _group_up():
        return 0
    if "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"])
    if LOCAL_PROCESS_GROUP is None:
        raise ValueError("no LOCAL_RANK in the environment and no LOCAL_PROCESS_GROUP")   # distributed.py:33-34
    return dist.get_rank(group=LOCAL_PROCESS_GROUP)


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def all_reduce(tensor, op=dist.ReduceOp.SUM):
    """distributed.py:64-72: identity at world size 1, else in-place all-reduce; returns the tensor."""
    if get_world_size() > 1:
        dist.all_reduce(tensor, op=op)
    return tensor


def all_gather(data):
    """distributed.py:75-107: one picklable object per rank -> list ordered by rank."""
    world = get_world_size()
    if world == 1:
        return [data]
    gathered = [None] * world
    dist.all_gather_object(gathered, data)
    return gathered


def reduce_dict(input_dict, average=True):
    """distributed.py:110-132: dict of same-shaped tensors summed onto rank 0 (mean if `average`)."""
    world = get_world_size()
    if world < 2:
        return input_dict
    names = sorted(input_dict)
    with torch.no_grad():
        stacked = torch.stack([input_dict[n] for n in names])
        dist.reduce(stacked, dst=0)
        if average and dist.get_rank() == 0:
            stacked /= world
    return dict(zip(names, stacked))


def data_sampler(dataset, shuffle, distributed):
    """distributed.py:135-143."""
    from torch.utils import data
    if distributed:
        return data.distributed.DistributedSampler(dataset, shuffle=shuffle)
    return data.RandomSampler(dataset) if shuffle else data.SequentialSampler(dataset)


# ---------------------------------------------------------------------------------------------- the data path
class TorchComm:
    """Gradient / EMA-sum exchange through the torch.distributed process group ("nccl" = RCCL)."""
    name = "torch.distributed"

    def all_reduce(self, tensor):
        dist.all_reduce(tensor)

    def broadcast(self, tensor, src=0):
        dist.broadcast(tensor, src)

    def world(self):
        return dist.get_world_size()


class NativeComm:
    """The same exchange through libvq2's own RCCL communicator (include/vq2.h vq2_comm_*): what a host that is
    not PyTorch binds.  Collectives are enqueued on the CURRENT stream, like the process-group calls."""
    name = "libvq2 vq2_comm"

    @staticmethod
    def _args(tensor):
        if not (tensor.is_cuda and tensor.dtype == torch.float32 and tensor.is_contiguous()):
            raise RuntimeError("NativeComm moves contiguous float32 device tensors")
        import ctypes
        return ctypes.c_void_p(tensor.data_ptr()), tensor.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def all_reduce(self, tensor):
        from ._lib import lib, check
        check(lib.vq2_comm_allreduce_sum(*self._args(tensor)), "comm_allreduce_sum")

    def broadcast(self, tensor, src=0):
        from ._lib import lib, check
        ptr, n, stream = self._args(tensor)
        check(lib.vq2_comm_broadcast(ptr, n, src, stream), "comm_broadcast")

    def world(self):
        from ._lib import lib
        return int(lib.vq2_comm_world())


_native = None


def native_comm_init():
    """Create libvq2's communicator for this job: rank 0 draws the RCCL unique id and publishes it through the
    rendezvous store (the process group's, or a TCPStore on MASTER_ADDR:MASTER_PORT when no group exists)."""
    global _native
    if _native is not None:
        return _native
    import ctypes
    from ._lib import lib, check
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if _group_up():
        store = dist.distributed_c10d._get_default_store()
    else:
        store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, is_master=(rank == 0))
    if rank == 0:
        buf = ctypes.create_string_buffer(128)
        check(lib.vq2_comm_unique_id(buf), "comm_unique_id")
        store.set("vq2_comm_id", buf.raw)
    uid = store.get("vq2_comm_id")
    check(lib.vq2_comm_init(ctypes.create_string_buffer(bytes(uid), 128), rank, world), "comm_init")
    _native = NativeComm()
    return _native


def data_comm():
    """The communicator Stage1Trainer moves gradients and EMA sums with: libvq2's own when it was brought up
    (VQ2_COMM=capi), else the torch.distributed group."""
    return _native if _native is not None else TorchComm()


# ---------------------------------------------------------------------------------------------- bring-up
def bringup(backend="nccl"):
    """Join the job described by the launcher environment: bind this process to its GPU, then create the process
    group (RCCL for "nccl").  Returns (rank, local_rank, world_size).  A no-op at WORLD_SIZE 1 unless
    VQ2_DP_FORCE=1 asks for a one-rank group (used to exercise the RCCL path on a single GPU)."""
    # dmabuf IPC is the only mode the host driver supports; must be set before the HIP runtime starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    gpu = backend == "nccl"
    if gpu:
        if not torch.cuda.is_available():
            raise OSError("CUDA is not available. Please check your environments")   # launch.py:55-56
        if local_rank >= torch.cuda.device_count():
            raise ValueError(f"local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
        torch.cuda.set_device(local_rank)
    force = os.environ.get("VQ2_DP_FORCE", "0") != "0"
    if (world > 1 or force) and not _group_up():
        missing = [k for k in _ENV_KEYS if k not in os.environ]
        if missing:
            raise OSError(f"failed to initialize NCCL groups: {missing} not set")          # launch.py:68-69
        kw = {"device_id": torch.device("cuda", local_rank)} if gpu else {}
        try:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        except Exception as exc:
            raise OSError("failed to initialize NCCL groups") from exc
        synchronize()
    if gpu and (world > 1 or force) and os.environ.get("VQ2_COMM", "torch") == "capi":
        native_comm_init()
    return rank, local_rank, world


def find_free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _child(local_rank, fn, world, per_machine, machine_rank, addr, port, backend, args):
    os.environ.update(RANK=str(machine_rank * per_machine + local_rank), LOCAL_RANK=str(local_rank),
                      WORLD_SIZE=str(world), MASTER_ADDR=addr, MASTER_PORT=str(port))
    bringup(backend)
    try:
        fn(*args)
    finally:
        if _group_up():
            dist.destroy_process_group()


def launch(fn, n_gpu_per_machine, n_machine=1, machine_rank=0, dist_url=None, args=(), backend="nccl"):
    """launch.py:22-49: run fn(*args) on n_machine x n_gpu_per_machine ranks, one process per GPU.  Under torchrun
    (the launcher already made the processes) it only joins the group; otherwise it spawns the local ranks."""
    world = n_machine * n_gpu_per_machine
    if all(k in os.environ for k in _ENV_KEYS) and int(os.environ["WORLD_SIZE"]) == world:
        bringup(backend)
        return fn(*args)
    if world == 1:
        return fn(*args)
    os.environ.setdefault("OMP_NUM_THREADS", "1")                    # launch.py:26-27
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # inherited by the children, before their HIP init
    if dist_url in (None, "auto"):
        if n_machine != 1:
            raise ValueError('dist_url="auto" not supported in multi-machine jobs')       # launch.py:30-33
        addr, port = "127.0.0.1", find_free_port()
    else:
        if not dist_url.startswith("tcp://"):
            raise ValueError(f"dist_url must look like tcp://host:port, got {dist_url!r}")
        addr, _, port = dist_url[len("tcp://"):].rpartition(":")
    import torch.multiprocessing as mp
    mp.spawn(_child, nprocs=n_gpu_per_machine, daemon=False,
             args=(fn, world, n_gpu_per_machine, machine_rank, addr, int(port), backend, args))
