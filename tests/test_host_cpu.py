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
    assert lib.vq2_version() == amd._lib.API_VERSION == int(re.search(r"#define\s+VQ2_API_VERSION\s+(\d+)", hdr).group(1))


def test_hot_kernels_do_not_spill():
    """The compiler's resource report written by csrc/build.sh: none of the kernels the train step runs may spill a
    register or use scratch memory (round 2: ONE spilled VGPR in the four-per-CU conv tile = 3 % of the whole step,
    invisible to every functional test)."""
    import glob
    files = glob.glob(os.path.join(ROOT, "vq-vae-2-pytorch_amd", "csrc", "_obj", "*.res"))
    if not files:
        pytest.skip("no resource reports: run __graft_entry__.build() first")
    hot = re.compile(r"conv_gemm_fast_kernel|wino3_kernel|wino_k4s2_kernel|wino_subpixel_kernel|rbw_fwd_kernel|conv1x1_k64_kernel|wgrad_fast_kernel|resblock_|subpixel_conv|vq_fwd_kernel|conv_k4s2_c4|"
                     r"convT_small_mfma|vq_stats_|wgrad_reduce_batched|adam_kernel|mse_")
    seen = 0
    for f in files:
        name = None
        for line in open(f):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                seen += bool(hot.search(name))
                continue
            m = re.search(r"(VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", line)
            if m and name and hot.search(name):
                assert int(m.group(2)) == 0 or m.group(1) == "SGPRs Spill" and int(m.group(2)) <= 16, \
                    f"{name}: {m.group(1)} = {m.group(2)}"
    assert seen >= 40, f"only {seen} hot kernels found in the reports"


def test_weight_gradient_plans_and_slab_layouts_without_gpu(amd):
    """Host side of the weight-gradient launches (pure planning code, no device): which shapes take the Winograd-domain
    kernels, how large their slab workspaces are, and what the reduction job says about the slab layout
    (vq2_wgrad_job.swapped: bit 0 exchanged roles, bits 1-2 the Winograd mode)."""
    lib = amd._lib.lib
    lib.vq2_conv_wgrad_workspace_bytes.restype = ctypes.c_size_t

    def plan(cin, cout, k, stride, pad, transposed, n, h, w):
        d = amd._lib.ConvDesc()
        d.N, d.H, d.W, d.Ci, d.Co, d.KH, d.KW, d.stride, d.pad = n, h, w, cin, cout, k, k, stride, pad
        d.transposed, d.ldx, d.ldy, d.Cir, d.Cor = int(transposed), cin, cout, cin, cout
        job = amd._lib.WgradJob()
        dummy = ctypes.c_void_p(16)
        rc = lib.vq2_wgrad_job_init(ctypes.byref(d), dummy, dummy, dummy, ctypes.byref(job))
        assert rc == 0, lib.vq2_last_error()
        return lib.vq2_conv_wgrad_workspace_bytes(ctypes.byref(d)), job

    # 3x3 128 -> 128 at 64x64: F(2,3) over column pairs -- K axis 12 * I, one thread column per (o, kh, i) in the reduction
    ws, job = plan(128, 128, 3, 1, 1, False, 32, 64, 64)
    assert (job.swapped >> 1) & 3 == 1 and job.swapped & 1 == 0 and job.taps == 9
    assert job.n_units_w == 128 * 3 * 128 // 32 and job.n_units_b == 128 // 32
    assert ws == (job.S * 128 * 12 * 128 + job.S * 128) * 4
    # the same layer at 32x32: rows of 16 pairs, two image rows per 32-pair chunk -- still the Winograd layout
    ws, job = plan(128, 128, 3, 1, 1, False, 32, 32, 32)
    assert (job.swapped >> 1) & 3 == 1 and ws == (job.S * 128 * 12 * 128 + job.S * 128) * 4
    # ... and at 16x16 (neither whole 64- nor 32-pixel rows): direct form, K axis 9 * I
    ws, job = plan(128, 128, 3, 1, 1, False, 32, 16, 16)
    assert job.swapped == 0 and job.n_units_w == 128 * 9 * 128 // 32 and ws == (job.S * 128 * 9 * 128 + job.S * 128) * 4
    # 4x4 stride-2 64 -> 128 from 128x128, and the conv-transpose 128 -> 64 to 128x128: F(2,2) by column parity, K axis 24 * I
    ws, job = plan(64, 128, 4, 2, 1, False, 32, 128, 128)
    assert (job.swapped >> 1) & 3 == 2 and job.n_units_w == 128 * 8 * 64 // 32 and job.bias_splits == 0
    assert ws == (job.S * 128 * 24 * 64 + job.S * 128) * 4
    ws, job = plan(128, 64, 4, 2, 1, True, 32, 64, 64)
    assert (job.swapped >> 1) & 3 == 2 and job.O == 128 and job.I == 64 and job.bias_splits == 4 * job.S
    assert ws == (job.S * 128 * 24 * 64 + job.S * 4 * 64) * 4
    # the ResBlock 3x3 (128 -> 32): exchanged roles + F(2,3), four 128 x 96 tiles (also at 32x32); direct exchanged roles at 16x16
    ws, job = plan(128, 32, 3, 1, 1, False, 32, 64, 64)
    assert job.swapped == 1 | (3 << 1) and job.O == 128 and job.I == 32 and job.n_units_w == 128 * 3 * 32 // 32
    assert ws == (job.S * 128 * 384 + job.S * 32) * 4
    ws, job = plan(128, 32, 3, 1, 1, False, 32, 32, 32)
    assert job.swapped == 1 | (3 << 1) and ws == (job.S * 128 * 384 + job.S * 32) * 4
    ws, job = plan(128, 32, 3, 1, 1, False, 32, 16, 16)
    assert job.swapped == 1 and ws == (job.S * 128 * 288 + job.S * 32) * 4
    # every plan keeps at least eight 32-row chunks per split and fills the resident slots
    for args in ((128, 128, 3, 1, 1, False, 32, 64, 64), (64, 128, 4, 2, 1, False, 32, 128, 128), (128, 32, 3, 1, 1, False, 8, 128, 128)):
        _, job = plan(*args)
        assert 1 <= job.S <= 256


def test_invalid_arguments_are_rejected_without_gpu(amd):
    lib = amd._lib.lib
    d = amd._lib.ConvDesc()
    assert lib.vq2_conv_fwd(ctypes.byref(d), 0, None, None, None, None, 0, None, None) == 1
    assert b"non-positive" in lib.vq2_last_error()
    d.N, d.H, d.W, d.Ci, d.Co, d.KH, d.KW, d.stride, d.pad, d.ldx, d.ldy = 1, 8, 8, 6, 8, 3, 3, 1, 1, 8, 8
    assert lib.vq2_conv_fwd(ctypes.byref(d), 0, None, None, None, None, 0, None, None) == 1
    assert b"multiples of 4" in lib.vq2_last_error()
    assert lib.vq2_vq_fwd(None, 0, None, None, None, 0, 0, 0, None, None, 0, None, None) == 1
    assert lib.vq2_vq_stats(None, 0, None, 0, 0, 0, None, None, None, 0, None) == 1
    assert lib.vq2_vq_stats_workspace_bytes(131072, 64, 8192) > 131072 * 8
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


def test_cycle_scheduler_matches_reference_golden(amd, golden):
    """Trajectories captured from the reference's scheduler.py (oracle/make_golden.py gen_scheduler), for the
    product scheduler AND the oracle's restatement; then an exact resume from state_dict()."""
    from oracle.make_golden_cases import SCHED_CASES
    import numpy as np
    g = golden("scheduler")
    for tag, kw, steps in SCHED_CASES:
        opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        s = amd.CycleScheduler(opt, **kw)
        lrs, moms = [], []
        for i in range(steps):
            lr, mom = s.step()
            lrs.append(lr)
            moms.append(np.nan if mom is None else mom)
            assert opt.param_groups[0]["lr"] == lr
            assert opt.param_groups[0]["betas"][0] == g[f"{tag}.group_beta1"][i]
        # bit parity: every segment is evaluated with the reference's own expressions (scheduler.py:221-228)
        assert np.array_equal(np.asarray(lrs), g[f"{tag}.lr"]), tag
        assert np.array_equal(np.asarray(moms), g[f"{tag}.momentum"], equal_nan=True), tag
    tag, kw, steps = SCHED_CASES[0]
    o = O.CycleSchedule(kw["lr_max"], kw["n_iter"], warmup_proportion=kw["warmup_proportion"])
    np.testing.assert_allclose([o.step() for _ in range(steps)], g[f"{tag}.lr"], rtol=1e-12, atol=0)
    # resume: 123 steps, save, fresh scheduler, load, continue == uninterrupted
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    a = amd.CycleScheduler(opt, **kw)
    for _ in range(123):
        a.step()
    b = amd.CycleScheduler(opt, **kw)
    b.load_state_dict(a.state_dict())
    assert np.array_equal(np.asarray([b.step()[0] for _ in range(400)]), g[f"{tag}.lr"][123:523])
    with pytest.raises(ZeroDivisionError):      # the reference divides by an empty warm-up phase on its first step
        amd.CycleScheduler(opt, 1e-3, n_iter=10, warmup_proportion=0.05).step()


def test_code_rows_are_the_reference_pickles(amd, tmp_path):
    """extract_code.py:27-33 / dataset.py:11,36-51: rows are pickle.dumps(CodeRow(top, bottom, filename)) under
    str(index) keys plus a 'length' key.  The expected bytes are built here from the field spec alone."""
    import collections
    import pickle
    import sys
    import types
    import numpy as np
    from vqvae2_amd import codes
    top = np.arange(12, dtype=np.int64).reshape(3, 4)
    bottom = np.arange(48, dtype=np.int64).reshape(6, 8) * 3
    mod = types.ModuleType("dataset")
    mod.CodeRow = collections.namedtuple("CodeRow", ["top", "bottom", "filename"])
    mod.CodeRow.__module__ = "dataset"
    assert "dataset" not in sys.modules
    sys.modules["dataset"] = mod
    try:
        want = pickle.dumps(mod.CodeRow(top=top, bottom=bottom, filename="cls/img_0.png"))
    finally:
        del sys.modules["dataset"]
    got = codes.code_row_bytes(top, bottom, "cls/img_0.png")
    assert got == want and "dataset" not in sys.modules
    path = str(tmp_path / "codes.db")
    with codes.CodeStore(path, "w", backend="sqlite") as st:
        codes.write_code_rows(st, [top, top + 1], [bottom, bottom + 1], ["a/0.png", "a/1.png"], start=0)
        codes.write_code_rows(st, [top + 2], [bottom + 2], ["b/2.png"], start=2)
        st.put(b"length", b"3")
    ds = codes.CodeDataset(path)
    assert len(ds) == 3
    t2, b2, name = ds[2]
    assert name == "b/2.png" and torch.equal(t2, torch.from_numpy(top + 2)) and torch.equal(b2, torch.from_numpy(bottom + 2))
    with codes.CodeStore(path, "r") as st:
        assert st.get(b"1") == codes.code_row_bytes(top + 1, bottom + 1, "a/1.png") and st.get(b"length") == b"3"


def test_code_row_loader_refuses_foreign_globals(amd, tmp_path):
    """A row blob that names anything but dataset.CodeRow / numpy's array rebuild helpers must raise, not run."""
    import pickle
    import numpy as np
    from vqvae2_amd import codes
    marker = tmp_path / "ran"

    class Evil:
        def __reduce__(self):
            return (open, (str(marker), "w"))
    with pytest.raises(pickle.UnpicklingError):
        codes.load_code_row(pickle.dumps(Evil()))
    assert not marker.exists()
    with pytest.raises(pickle.UnpicklingError):      # right container, wrong payload types
        codes.load_code_row(pickle.dumps(("top", "bottom", "name")))
    with pytest.raises(pickle.UnpicklingError):      # object arrays could smuggle arbitrary pickles
        codes.load_code_row(codes.code_row_bytes(np.array([{"a": 1}], dtype=object), np.zeros(2, np.int64), "x"))
    t, b, n = codes.load_code_row(codes.code_row_bytes(np.arange(4), np.arange(6), "k/x.png"))
    assert n == "k/x.png" and t.tolist() == [0, 1, 2, 3] and b.dtype == np.int64


def test_extract_code_preprocessing_uses_torchvision_integer_rules(amd):
    """transforms.Resize(size) truncates the long side (int(size * long / short)) and transforms.CenterCrop(size)
    offsets are int(round((dim - size) / 2.0)) -- hand-computed boxes for odd-sized images (extract_code.py:46-51;
    torchvision itself is not importable here, so these are the formulas' values, not a captured run)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("extract_code_example", os.path.join(ROOT, "examples", "extract_code.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    box = mod.resize_crop_box
    assert box(301, 200, 64) == ((96, 64), (16, 0, 80, 64))           # 64*301/200 = 96.32 -> 96
    assert box(203, 100, 33) == ((66, 33), (16, 0, 49, 33))           # 66.99 truncates to 66 (rounding would say 67)
    assert box(207, 100, 33) == ((68, 33), (18, 0, 51, 33))           # (68-33)/2 = 17.5 -> round -> 18 (floor: 17)
    assert box(100, 206, 33) == ((33, 67), (0, 17, 33, 50))           # 67.98 truncates to 67
    assert box(256, 256, 256) == ((256, 256), (0, 0, 256, 256))


def test_unsent_bucket_slices_cover_the_gradient_buffer(amd):
    """Stage1Trainer's fallback after backward: whatever slices the backward hooks did NOT hand to the communicator go
    out then, as maximal contiguous ranges -- never a gap, never a slice twice (no silent un-reduced gradient)."""
    import types
    fake = types.SimpleNamespace(arena=types.SimpleNamespace(flat_g=torch.zeros(100)), _sent=[])
    unsent = lambda: amd.Stage1Trainer._unsent_slices(fake)
    assert unsent() == [(0, 100)]                       # no hook fired: one collective over everything
    fake._sent = [(60, 90)]                             # tail only
    assert unsent() == [(0, 60), (90, 100)]
    fake._sent = [(60, 90), (0, 60)]                    # tail + middle: enc_b's slice is what stays exposed
    assert unsent() == [(90, 100)]
    fake._sent = [(0, 60), (60, 100)]
    assert unsent() == []


def test_launch_spawns_ranks_and_joins_the_group(amd, tmp_path):
    """distributed.launch (launch.py:22-49) with the gloo backend: two child ranks, environment-first bring-up."""
    out = str(tmp_path / "ranks")
    os.makedirs(out)
    amd.distributed.launch(_launch_probe, 2, 1, 0, "auto", args=(out,), backend="gloo")
    assert sorted(os.listdir(out)) == ["0_of_2_local0", "1_of_2_local1"]


def _launch_probe(out):
    import vqvae2_amd.distributed as d
    x = torch.ones(2) * (d.get_rank() + 1)
    d.all_reduce(x)
    assert float(x[0]) == 3.0
    open(os.path.join(out, f"{d.get_rank()}_of_{d.get_world_size()}_local{d.get_local_rank()}"), "w").close()
    d.synchronize()


def test_distributed_helpers_single_process(amd):
    d = amd.distributed
    assert d.get_world_size() == 1 and d.get_rank() == 0 and d.is_primary()
    x = torch.ones(3)
    assert d.all_reduce(x) is x
    assert d.all_gather({"a": 1}) == [{"a": 1}]
    d.synchronize()
