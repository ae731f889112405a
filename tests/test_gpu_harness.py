"""GPU: the harness around the step: exact resume of a training run from
Stage1Trainer.state_dict(), and the two example scripts end to end (train -> reference-format checkpoint ->
extract codes -> CodeRow rows)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import vqvae_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trainer(amd, seed):
    cfg = O.TINY
    m = amd.VQVAE(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
                  embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    m.load_state_dict(O.make_state(cfg, seed))
    m.cuda()
    return m, amd.Stage1Trainer(m, lr=1e-3, sched="cycle", n_iter=40)


def test_resume_continues_the_uninterrupted_trajectory(tmp_path):
    """5 steps -> save -> fresh model / optimizer / scheduler -> load -> 5 steps == 10 uninterrupted steps, bit for
    bit: parameters, EMA buffers, Adam moments and the CycleScheduler position all come back (the reference saves
    only model.state_dict(), train_vqvae.py:205-206; its --resume restarts Adam and the schedule)."""
    import vqvae2_amd as amd
    imgs = [O.make_images(4, 32, 900 + s).cuda() for s in range(10)]
    m_ref, tr_ref = _trainer(amd, 5)
    ref_losses = [float(tr_ref.step(x)["loss"]) for x in imgs]
    m_a, tr_a = _trainer(amd, 5)
    first = [float(tr_a.step(x)["loss"]) for x in imgs[:5]]
    path = str(tmp_path / "ckpt.pt")
    torch.save(tr_a.state_dict(), path)
    del m_a, tr_a
    m_b, tr_b = _trainer(amd, 777)                              # different initial state: everything must come from the file
    tr_b.load_state_dict(torch.load(path, map_location="cuda", weights_only=True))
    second = [float(tr_b.step(x)["loss"]) for x in imgs[5:]]
    assert first + second == ref_losses
    for (k, a), (_, b) in zip(m_ref.state_dict().items(), m_b.state_dict().items()):
        if not k.startswith("dec_ir."):
            assert torch.equal(a, b), k
    assert torch.equal(tr_ref.optimizer._m, tr_b.optimizer._m) and torch.equal(tr_ref.optimizer._v, tr_b.optimizer._v)
    assert tr_ref.optimizer._t == tr_b.optimizer._t == 10
    assert tr_ref.optimizer.param_groups[0]["lr"] == tr_b.optimizer.param_groups[0]["lr"]
    # the reference's own resume file (bare model state_dict, DDP "module." prefix) loads too
    m_c, tr_c = _trainer(amd, 778)
    tr_c.load_state_dict({"module." + k: v for k, v in m_ref.state_dict().items()})
    assert torch.equal(m_c.state_dict()["enc_b.blocks.0.weight"], m_ref.state_dict()["enc_b.blocks.0.weight"])
    assert tr_c.optimizer._t == 0


def test_two_models_in_one_process_do_not_share_state():
    """No process-global mutable state on the autograd path: interleaving the steps of two trainers
    gives each the trajectory it has alone."""
    import vqvae2_amd as amd
    imgs = [O.make_images(2, 32, 950 + s).cuda() for s in range(3)]
    m1, t1 = _trainer(amd, 11)
    m2, t2 = _trainer(amd, 12)
    inter = [(float(t1.step(x)["loss"]), float(t2.step(x)["loss"])) for x in imgs]
    m3, t3 = _trainer(amd, 11)
    alone1 = [float(t3.step(x)["loss"]) for x in imgs]
    m4, t4 = _trainer(amd, 12)
    alone2 = [float(t4.step(x)["loss"]) for x in imgs]
    assert [a for a, _ in inter] == alone1 and [b for _, b in inter] == alone2
    assert all(torch.equal(a, b) for a, b in zip(m1.state_dict().values(), m3.state_dict().values()))


def test_example_scripts_train_then_extract(tmp_path):
    """examples/train_stage1.py (reference CLI, reference-format checkpoint) and examples/extract_code.py
    (CodeRow rows + 'length' key), run once as a user would; the stored rows decode to model.encode()'s indices."""
    import vqvae2_amd as amd
    from vqvae2_amd import codes
    data = tmp_path / "data"
    data.mkdir()
    rng = np.random.default_rng(0)
    for i in range(2):
        np.save(data / f"batch{i}.npy", rng.standard_normal((4, 3, 64, 64)).astype(np.float32))
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = tmp_path / "ckpt"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_stage1.py"), "--size", "64", "--batch_size",
                        "1", "--epoch", "3", "--sched", "cycle", "--path", str(data), "--out", str(out)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "avg mse" in r.stdout
    ckpt = out / "vqvae_003.pt"      # 8 steps per epoch x 3 epochs = 24 schedule steps (1 warm-up step at 5 %)
    assert ckpt.exists() and (out / "trainer_003.pt").exists() and (out / "vqvae_001.pt").exists()
    sd = torch.load(ckpt, map_location="cpu", weights_only=True)
    assert list(sd.keys()) == list(O.state_spec(O.DEFAULT).keys())      # the reference's state_dict layout
    # resume from the trainer checkpoint of epoch 1 for epochs 2 and 3
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_stage1.py"), "--size", "64",
                         "--batch_size", "1", "--epoch", "3", "--sched", "cycle", "--path", str(data), "--out",
                         str(tmp_path / "ckpt2"), "--resume", str(out / "trainer_001.pt")], env=env,
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    sd2 = torch.load(tmp_path / "ckpt2" / "vqvae_003.pt", map_location="cpu", weights_only=True)
    assert all(torch.equal(sd[k], sd2[k]) for k in sd if not k.startswith("dec_ir."))    # exact resume
    name = str(tmp_path / "codes.db")
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "extract_code.py"), "--size", "64", "--ckpt",
                         str(ckpt), "--name", name, str(data)], env=env, capture_output=True, text=True, timeout=600)
    assert r3.returncode == 0, r3.stdout[-2000:] + r3.stderr[-2000:]
    ds = codes.CodeDataset(name)
    assert len(ds) == 8
    m = amd.VQVAE()
    m.load_state_dict(sd)
    m.cuda().eval()
    with torch.no_grad():
        _, _, _, id_t, id_b = m.encode(torch.from_numpy(np.load(data / "batch1.npy")).cuda())
    top, bottom, fname = ds[6]
    assert fname == "batch1.npy:2" and top.dtype == torch.int64 and tuple(top.shape) == (8, 8) and tuple(bottom.shape) == (16, 16)
    assert torch.equal(top, id_t[2].cpu()) and torch.equal(bottom, id_b[2].cpu())


def test_gradient_accumulation_and_arena_guards():
    """With a ParamArena active, a second backward without zero_grad must ACCUMULATE (old + new), not overwrite the
    slot that p.grad aliases; and FusedAdam refuses to fork its state when a gradient missed the arena."""
    import vqvae2_amd as amd
    m, tr = _trainer(amd, 21)
    img = O.make_images(2, 32, 970).cuda()
    crit = torch.nn.MSELoss()

    def backward_once():
        dec, diff = m(img)
        (crit(dec, img) + 0.25 * diff.mean()).backward()

    m.train()
    tr.arena.zero_grad()
    backward_once()
    g1 = {k: p.grad.clone() for k, p in m.live_named_parameters()}
    backward_once()                                   # accumulates into the same .grad tensors
    for k, p in m.live_named_parameters():
        # the trainer defers the EMA update, so both passes see the same codebooks: the sum is exactly twice the first
        assert torch.equal(p.grad, 2 * g1[k]), k
    # a gradient that does not live in the arena: the fused optimizer must raise instead of silently re-creating m/v
    tr.arena.zero_grad()
    backward_once()
    p0 = tr.arena.params[0]
    p0.grad = p0.grad.clone()
    with pytest.raises(RuntimeError, match="did not land in the flat arena"):
        tr.optimizer.step()
