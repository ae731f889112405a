"""Child of tests/test_gpu_rccl.py: started by `python -m torch.distributed.run --nproc-per-node 1` (the launcher
runs before anything touches the GPU) with VQ2_DP_FORCE=1, so Stage1Trainer takes its data-parallel path over
the real "nccl" (= RCCL) backend at world size 1: group init with device binding, the initial broadcast, both
gradient buckets on the side stream, events against the compute stream."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import vqvae2_amd  # noqa: E402
from oracle import vqvae_oracle as O  # noqa: E402
from tests._train_cases import run_case, run_dropin_case  # noqa: E402


def main():
    out = sys.argv[1]
    rank, local_rank, world = vqvae2_amd.distributed.bringup("nccl")
    assert world == 1 and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
    want_native = os.environ.get("VQ2_COMM") == "capi"
    assert (vqvae2_amd._lib.lib.vq2_comm_world() == 1) == want_native
    info = {"data_path": vqvae2_amd.distributed.data_comm().name}
    for case in ("tiny", "default64"):
        sd, losses, tr = run_case(vqvae2_amd, case)
        assert tr.dp and tr.comm_stream is not None and len(tr.buckets) == 3
        assert isinstance(tr.comm, vqvae2_amd.distributed.NativeComm) == want_native
        info[case] = {"early": tr.early_buckets, "losses": losses}
        np.savez(os.path.join(out, f"{case}.npz"), **sd)
    # the reference's own wrapping (train_vqvae.py:166-171): DistributedDataParallel around the drop-in module
    sd, losses = run_dropin_case(vqvae2_amd, wrap_ddp=True)
    assert all(k.startswith("module.") for k in sd)            # the key prefix the reference's checkpoints carry
    info["ddp"] = {"losses": losses}
    np.savez(os.path.join(out, "ddp.npz"), **sd)
    # the mirrored helper over RCCL (distributed.py:64-72 returns the tensor untouched at world size 1)
    t = torch.ones(4, device="cuda")
    assert vqvae2_amd.distributed.all_reduce(t) is t and float(t.sum()) == 4.0
    torch.distributed.all_reduce(t)      # the collective itself, on the current stream
    torch.cuda.synchronize()
    assert float(t.sum()) == 4.0
    with open(os.path.join(out, "info.json"), "w") as f:
        json.dump(info, f)
    if want_native:
        vqvae2_amd.distributed.data_comm().all_reduce(t)   # vq2_comm_allreduce_sum on the current stream
        torch.cuda.synchronize()
        assert float(t.sum()) == 4.0
        assert vqvae2_amd._lib.lib.vq2_comm_destroy() == 0 and vqvae2_amd._lib.lib.vq2_comm_world() == 0
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
