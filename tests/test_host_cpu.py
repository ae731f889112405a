"""CPU-side checks: the C-ABI library loads and exports every symbol include/vq2.h declares (no
compute calls without a GPU), host logic (module layout, fusion plan, scheduler) behaves."""
import ctypes
import os
import re

import pytest
import torch

from oracle import vqvae_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def amd():
    import vqvae2_amd
    return vqvae2_amd


def test_library_exports_every_declared_symbol(amd):
    hdr = open(os.path.join(ROOT, "include", "vq2.h")).read()
    declared = set(re.findall(r"\b(vq2_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = ctypes.CDLL(amd._lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in vq2.h but not exported"
    assert declared == set(amd._lib.EXPORTS), declared ^ set(amd._lib.EXPORTS)
    assert lib.vq2_version() >= 1


def test_invalid_arguments_are_rejected_without_gpu(amd):
    lib = amd._lib.lib
    d = amd._lib.ConvDesc()
    assert lib.vq2_conv_fwd(ctypes.byref(d), 0, None, None, None, None, 0, None, None) == 1
    assert b"non-positive" in lib.vq2_last_error()
    d.N, d.H, d.W, d.Ci, d.Co, d.KH, d.KW, d.stride, d.pad, d.ldx, d.ldy = 1, 8, 8, 6, 8, 3, 3, 1, 1, 8, 8
    assert lib.vq2_conv_fwd(ctypes.byref(d), 0, None, None, None, None, 0, None, None) == 1
    assert b"multiples of 4" in lib.vq2_last_error()
    assert lib.vq2_vq_fwd(None, 0, None, None, None, 0, 0, 0, None, None, 0, None, None, None, None) == 1
    assert lib.vq2_adam_step(None, None, None, None, 0, 1e-3, .9, .999, 1e-8, 1, 1.0, None) == 1
    with pytest.raises(RuntimeError):
        amd._lib.check(1, "x")


def test_state_dict_layout_matches_reference(amd):
    m = amd.VQVAE()
    spec = O.state_spec(O.DEFAULT)
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
    assert m.embed_dim == 128
    live = dict(m.live_named_parameters())
    assert sum(p.numel() for p in live.values()) == 1388867
    assert sum(p.numel() for p in m.parameters()) == 1833092
    m.load_state_dict(O.make_state(O.DEFAULT))      # round-trips a reference-layout checkpoint
    t = amd.VQVAE(channel=32, n_res_block=1, n_res_channel=8, embed_dim=16, n_embed=64)
    assert list(t.state_dict().keys()) == list(O.state_spec(O.TINY).keys())


def test_no_cpu_fallback(amd):
    m = amd.Conv2d(8, 8, 3, padding=1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 8, 4, 4))
    with pytest.raises(NotImplementedError):
        amd.ConvTranspose2d(8, 8, 3, stride=1)
    with pytest.raises(ValueError):
        amd.Encoder(3, 32, 1, 8, stride=3)


def test_cycle_scheduler_matches_oracle(amd):
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=3e-4)
    s = amd.CycleScheduler(opt, 3e-4, n_iter=200, momentum=None, warmup_proportion=0.05)
    o = O.CycleSchedule(3e-4, 200, warmup_proportion=0.05)
    for _ in range(450):   # crosses two restarts
        lr, mom = s.step()
        assert mom is None and abs(lr - o.step()) < 1e-15
        assert opt.param_groups[0]["lr"] == lr
    s2 = amd.CycleScheduler(opt, 1e-3, n_iter=10)
    lrs = [s2.step() for _ in range(10)]
    assert abs(lrs[2][0] - 1e-3) < 1e-12 and abs(lrs[2][1] - 0.85) < 1e-12


def test_distributed_helpers_single_process(amd):
    d = amd.distributed
    assert d.get_world_size() == 1 and d.get_rank() == 0 and d.is_primary()
    x = torch.ones(3)
    assert d.all_reduce(x) is x
    assert d.all_gather({"a": 1}) == [{"a": 1}]
    d.synchronize()
