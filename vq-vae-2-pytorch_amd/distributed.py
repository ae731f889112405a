"""Mirror of the reference's distributed/ package (distributed/__init__.py:1-13) on RCCL.

Same names and argument meaning as /root/reference/distributed/distributed.py and launch.py.
On ROCm the torch.distributed backend string "nccl" IS RCCL; one process per GPU.
"""
import os
import socket

import torch
from torch import distributed as dist
from torch import multiprocessing as mp
from torch.utils import data

LOCAL_PROCESS_GROUP = None


def is_primary():
    return get_rank() == 0


def get_rank():
    if not dist.is_available() or not dist.is_initialized():
        return 0
    return dist.get_rank()


def get_local_rank():
    if not dist.is_available() or not dist.is_initialized():
        return 0
    if LOCAL_PROCESS_GROUP is None:
        # torchrun-style launches carry the local rank in the environment
        if "LOCAL_RANK" in os.environ:
            return int(os.environ["LOCAL_RANK"])
        raise ValueError("tensorfn.distributed.LOCAL_PROCESS_GROUP is None")  # distributed.py:33-34
    return dist.get_rank(group=LOCAL_PROCESS_GROUP)


def synchronize():
    if not dist.is_available() or not dist.is_initialized():
        return
    if dist.get_world_size() == 1:
        return
    dist.barrier()


def get_world_size():
    if not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size()


def all_reduce(tensor, op=dist.ReduceOp.SUM):
    """distributed.py:64-72: identity at world size 1, else in-place all-reduce; returns tensor."""
    if get_world_size() == 1:
        return tensor
    dist.all_reduce(tensor, op=op)
    return tensor


def all_gather(data):
    """distributed.py:75-107: gather arbitrary picklable objects from every rank into a list."""
    world_size = get_world_size()
    if world_size == 1:
        return [data]
    out = [None] * world_size
    dist.all_gather_object(out, data)
    return out


def reduce_dict(input_dict, average=True):
    """distributed.py:110-132: reduce a dict of 0-dim tensors to rank 0 (averaged by default)."""
    world_size = get_world_size()
    if world_size < 2:
        return input_dict
    with torch.no_grad():
        keys = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in keys], 0)
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0 and average:
            values /= world_size
        return {k: v for k, v in zip(keys, values)}


def data_sampler(dataset, shuffle, distributed):
    if distributed:
        return data.distributed.DistributedSampler(dataset, shuffle=shuffle)
    if shuffle:
        return data.RandomSampler(dataset)
    return data.SequentialSampler(dataset)


# ------------------------------------------------------------------ launch.py:10-92
def find_free_port():
    sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    sock.bind(("", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def launch(fn, n_gpu_per_machine, n_machine=1, machine_rank=0, dist_url=None, args=()):
    world_size = n_machine * n_gpu_per_machine
    if world_size > 1:
        if "OMP_NUM_THREADS" not in os.environ:
            os.environ["OMP_NUM_THREADS"] = "1"
        if dist_url == "auto":
            if n_machine != 1:
                raise ValueError('dist_url="auto" not supported in multi-machine jobs')
            dist_url = f"tcp://127.0.0.1:{find_free_port()}"
        mp.spawn(distributed_worker, nprocs=n_gpu_per_machine,
                 args=(fn, world_size, n_gpu_per_machine, machine_rank, dist_url, args), daemon=False)
    else:
        fn(*args)


def distributed_worker(local_rank, fn, world_size, n_gpu_per_machine, machine_rank, dist_url, args,
                       backend="nccl"):
    if backend == "nccl" and not torch.cuda.is_available():
        raise OSError("CUDA is not available. Please check your environments")  # launch.py:55-56
    global_rank = machine_rank * n_gpu_per_machine + local_rank
    # keep dmabuf IPC for RCCL on this driver
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        dist.init_process_group(backend=backend, init_method=dist_url, world_size=world_size, rank=global_rank)
    except Exception:
        raise OSError("failed to initialize NCCL groups")  # launch.py:68-69
    synchronize()
    if backend == "nccl":
        if n_gpu_per_machine > torch.cuda.device_count():
            raise ValueError(f"specified n_gpu_per_machine larger than available device "
                             f"({torch.cuda.device_count()})")
        torch.cuda.set_device(local_rank)
    global LOCAL_PROCESS_GROUP
    if LOCAL_PROCESS_GROUP is not None:
        raise ValueError("torch.distributed.LOCAL_PROCESS_GROUP is not None")
    n_machine = world_size // n_gpu_per_machine
    for i in range(n_machine):
        ranks_on_i = list(range(i * n_gpu_per_machine, (i + 1) * n_gpu_per_machine))
        pg = dist.new_group(ranks_on_i)
        if i == machine_rank:
            LOCAL_PROCESS_GROUP = pg
    fn(*args)
