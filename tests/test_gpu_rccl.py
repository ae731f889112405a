"""GPU: the RCCL path on the one GPU of the test box.  A fresh child under torch.distributed.run (--nproc-per-node 1,
backend "nccl") runs Stage1Trainer with VQ2_DP_FORCE=1 -- group init + device binding, initial broadcast, the tail
bucket from the backward hook and the head bucket on the side stream -- and must reproduce the plain
single-process run BIT FOR BIT (a one-rank all-reduce is the identity, and the step has no float atomics).
bench.py goes through the same launcher once."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(script_args, extra_env, nproc=1):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("comm", ["torch", "capi"])
def test_trainer_over_rccl_world1_equals_plain_run(tmp_path, comm):
    """comm = "torch": collectives through the torch.distributed "nccl" group; "capi": through libvq2's own
    communicator (vq2_comm_unique_id / init / allreduce_sum / broadcast / destroy, include/vq2.h)."""
    import vqvae2_amd
    from tests._train_cases import CASES, run_case
    r = _torchrun([os.path.join(ROOT, "tests", "_rccl_child.py"), str(tmp_path)], {"VQ2_DP_FORCE": "1", "VQ2_COMM": comm})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    info = json.load(open(tmp_path / "info.json"))
    assert info["data_path"] == ("libvq2 vq2_comm" if comm == "capi" else "torch.distributed")
    for case, (cfg, size, batch, steps, seed) in CASES.items():
        sd, losses, tr = run_case(vqvae2_amd, case)
        assert not tr.dp
        assert info[case]["early"] == 2 * steps, "tail and middle buckets must go out from backward hooks every step"
        assert info[case]["losses"] == losses
        got = np.load(tmp_path / f"{case}.npz")
        for k, v in sd.items():
            assert np.array_equal(got[k], v), f"{case}: {k} differs between the RCCL path and the plain path"


def test_ddp_wrapped_dropin_module_equals_plain(tmp_path):
    """the drop-in VQVAE inside nn.parallel.DistributedDataParallel over RCCL (world size 1)
    trains exactly like the bare module (a one-rank mean is the identity); checkpoints carry DDP's "module." prefix."""
    import vqvae2_amd
    from tests._train_cases import run_dropin_case
    r = _torchrun([os.path.join(ROOT, "tests", "_rccl_child.py"), str(tmp_path)], {"VQ2_DP_FORCE": "1"})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    info = json.load(open(tmp_path / "info.json"))
    sd, losses = run_dropin_case(vqvae2_amd, wrap_ddp=False)
    assert info["ddp"]["losses"] == losses
    got = np.load(tmp_path / "ddp.npz")
    for k, v in sd.items():
        assert np.array_equal(got["module." + k], v), k


def test_bench_under_torchrun_with_rccl_group():
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "tiny",
                   "--no-cpu-baseline"], {"VQ2_DP_FORCE": "1"})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    c = line["collectives"]
    assert c["backend"] == "nccl" and c["world"] == 1 and c["early_buckets"] == 8 and c["buckets_per_step"] == 3
    assert sorted(c["bucket_bytes"]) == ["head", "middle", "tail"] and all(v > 0 for v in c["bucket_bytes"].values())


def test_bench_two_ranks_rehearsal_over_gloo():
    """The driver's multi-GPU command line (torch.distributed.run --nproc-per-node N bench.py --gpus N) with N = 2
    ranks sharing the one GPU of the test box over gloo: rank-dependent inputs, the initial broadcast, both buckets,
    barriers, the MAX-over-ranks timing and the single JSON line of rank 0."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "tiny",
                   "--no-cpu-baseline"], {"VQ2_BENCH_BACKEND": "gloo", "VQ2_SHARE_GPU": "1"}, nproc=2)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["config"]["parallelism"] == "dp2"
    c = line["collectives"]
    assert line["scaling"] == "weak" and line["value"] > 0 and c["early_buckets"] == 8 and c["world"] == 2
    assert c["buckets_per_step"] == 3 and c["bucket_bytes"]["head"] < c["bucket_bytes"]["middle"] + c["bucket_bytes"]["tail"]
