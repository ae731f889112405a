"""GPU parity: libvq2 (through the C ABI, via vqvae2_amd) vs the reference's golden vectors and vs
the CPU oracle on identical seeded inputs.  Indices must be bit-exact; floating point within the
fp32 tolerances stated per test (different but valid fp32 accumulation orders)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import rng
from oracle import vqvae_oracle as O
from oracle.make_golden_cases import (BLOCK_CASES, CONV_FLAVOURS, SEED, block_state, conv_inputs,
                                      quantize_inputs)

pytestmark = pytest.mark.gpu

RT, AT = 2e-4, 2e-5   # fp32 tolerance for activations/gradients (sums of up to ~2k products)


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, rtol=RT, atol=AT, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


@pytest.fixture(scope="module")
def amd():
    import vqvae2_amd
    return vqvae2_amd


def test_native_library_is_loaded(amd):
    import ctypes
    assert isinstance(amd._lib.lib, ctypes.CDLL) and amd._lib.lib.vq2_version() == amd._lib.API_VERSION
    with open("/proc/self/maps") as f:
        assert "libvq2.so" in f.read()


def test_conv_flavours(amd, golden):
    g = golden("convs")
    for tag, kind, ws, stride, pad, hw in CONV_FLAVOURS:
        x, w, b = conv_inputs(tag, kind, ws, hw)
        if kind == "conv":
            m = amd.Conv2d(ws[1], ws[0], ws[2], stride=stride, padding=pad)
        else:
            m = amd.ConvTranspose2d(ws[0], ws[1], ws[2], stride=stride, padding=pad)
        m.load_state_dict({"weight": t(w), "bias": t(b)})
        m.to(dev())
        xt = t(x).to(dev()).requires_grad_(True)
        y = m(xt)
        assert tuple(y.shape) == tuple(g[f"{tag}.y"].shape), tag
        close(y, g[f"{tag}.y"], what=f"{tag}.y")
        y.backward(t(rng.normal(SEED, f"{tag}.gy", tuple(y.shape))).to(dev()))
        close(xt.grad, g[f"{tag}.gx"], what=f"{tag}.gx")
        close(m.weight.grad, g[f"{tag}.gw"], rtol=5e-4, atol=1e-4, what=f"{tag}.gw")
        close(m.bias.grad, g[f"{tag}.gb"], rtol=5e-4, atol=1e-4, what=f"{tag}.gb")


def _run_quantize(amd, tag, D, K, xshape, training, tie=False):
    x, embed, cs0, gw = quantize_inputs(tag, D, K, xshape, tie)
    q = amd.Quantize(D, K)
    q.load_state_dict({"embed": t(embed), "cluster_size": t(cs0), "embed_avg": t(embed) * t(cs0)[None, :]})
    q.to(dev()).train(training)
    xt = t(x).to(dev()).requires_grad_(True)
    out, diff, idx = q(xt)
    ((out * t(gw).to(dev())).sum() + 0.25 * diff).backward()
    return q, out, diff, idx, xt.grad


def test_quantize_golden(amd, golden):
    g = golden("quantize")
    for tag, D, K, xs, tr, tie in [("q512_train", 64, 512, (2, 8, 8, 64), True, False),
                                   ("q512_eval", 64, 512, (2, 8, 8, 64), False, False),
                                   ("q512_tie", 64, 512, (2, 8, 8, 64), True, True),
                                   ("q64_train", 16, 64, (2, 4, 4, 16), True, False)]:
        q, out, diff, idx, gx = _run_quantize(amd, tag, D, K, xs, tr, tie)
        assert idx.dtype == torch.int64 and tuple(idx.shape) == xs[:-1]
        assert np.array_equal(idx.cpu().numpy().astype(np.int32), g[f"{tag}.idx"]), f"{tag}: indices differ"
        close(out, g[f"{tag}.out"], rtol=1e-6, atol=1e-6, what=tag)
        close(diff, g[f"{tag}.diff"], rtol=1e-5, atol=1e-7, what=tag)
        close(gx, g[f"{tag}.xgrad"], rtol=1e-5, atol=1e-7, what=tag)
        close(q.cluster_size, g[f"{tag}.cluster_size_after"], rtol=1e-5, atol=1e-6, what=tag)
        close(q.embed_avg, g[f"{tag}.embed_avg_after"], rtol=1e-5, atol=1e-5, what=tag)
        close(q.embed, g[f"{tag}.embed_after"], rtol=1e-4, atol=1e-5, what=tag)
    gi = g["q512_tie.idx"].reshape(-1)
    assert gi[0] == 5 and gi[1] == 64 and gi[2] == 5


def test_quantize_8192_golden(amd, golden):
    g = golden("quantize")
    q, out, diff, idx, gx = _run_quantize(amd, "q8192_train", 64, 8192, (2, 16, 16, 64), True)
    assert np.array_equal(idx.cpu().numpy().astype(np.int32), g["q8192_train.idx"])
    close(out.reshape(-1, 64)[::16], g["q8192_train.out_rows"], rtol=1e-6, atol=1e-6)
    close(q.cluster_size, g["q8192_train.cluster_size_after"], rtol=1e-5, atol=1e-6)


def test_quantize_ragged_and_embed_code(amd):
    # M not a multiple of the 128-row workgroup tile, several tiny shapes, eval mode
    # (the K >= 1024 cases take the K-split search: code ranges searched by separate workgroups + vq_merge_kernel,
    #  including a last split that starts past K)
    for M, D, K in [(1, 64, 512), (37, 64, 512), (129, 16, 64), (1000, 32, 128), (300, 8, 36), (5, 4, 4),
                    (300, 64, 1540), (77, 16, 2052), (129, 32, 4096), (3000, 64, 1024)]:
        x = t(rng.normal(7, f"r.x{M}", (M, 1, 1, D)))
        e = t(rng.normal(7, f"r.e{M}", (D, K)))
        q = amd.Quantize(D, K)
        q.load_state_dict({"embed": e, "cluster_size": torch.zeros(K), "embed_avg": e.clone()})
        q.to(dev()).eval()
        out, diff, idx = q(x.to(dev()))
        ro, rd, ri = O.quantize_forward(x, e.clone(), torch.zeros(K), e.clone(), False)
        margin, _ = O.quantize_margin(x, e)
        bad = (idx.cpu() != ri).reshape(-1)
        assert not bool((bad & (margin > 1e-4)).any()), (M, D, K)
        if not bool(bad.any()):
            close(out, ro, rtol=1e-6, atol=1e-6)
            close(diff, rd, rtol=1e-5, atol=1e-7)
        code = q.embed_code(idx)
        close(code, F.embedding(idx.cpu(), e.t()), rtol=0, atol=0)


def test_quantize_large_properties(amd):
    # BASELINE config sizes (B=32 bottom level: M = 131072) through size-independent properties
    D, K, M = 64, 512, 32 * 64 * 64
    x = t(rng.normal(11, "big.x", (32, 64, 64, D))).to(dev())
    e = t(rng.normal(11, "big.e", (D, K)))
    q = amd.Quantize(D, K)
    q.load_state_dict({"embed": e, "cluster_size": torch.zeros(K), "embed_avg": e.clone()})
    q.to(dev()).train()
    e0 = q.embed.clone()
    out, diff, idx = q(x)
    # (1) every index is the fp64 argmin wherever the fp64 margin is not a near-tie
    margin, ref_idx = O.quantize_margin(x.cpu(), e)
    bad = idx.cpu().reshape(-1) != ref_idx
    assert int(bad.sum()) <= 8 and not bool((bad & (margin > 1e-3)).any())
    # (2) cluster_size EMA: counts sum to M  ->  sum(cluster_size) = 0.01 * M
    close(q.cluster_size.sum(), 0.01 * M, rtol=1e-5)
    # (3) idempotence on the pre-update codebook: quantising the quantised output returns the same codes
    q2 = amd.Quantize(D, K)
    q2.load_state_dict({"embed": e0.cpu(), "cluster_size": torch.zeros(K), "embed_avg": e0.cpu()})
    q2.to(dev()).eval()
    code = q2.embed_code(idx)
    _, d2, idx2 = q2(code)
    assert torch.equal(idx2, idx) and float(d2) < 1e-12
    # (4) diff equals the mean squared distance to the selected codes
    close(diff, (code - x).pow(2).mean(), rtol=1e-4)
    # (5) a ragged row count on the 512-vector-workgroup path (M >= 131072, not a multiple of 128): indices,
    #     loss and the complete EMA update against the oracle
    xr = torch.cat([x.reshape(-1, D), x.reshape(-1, D)[:200] * 0.5], 0).contiguous()
    q3 = amd.Quantize(D, K)
    q3.load_state_dict({"embed": e, "cluster_size": torch.zeros(K), "embed_avg": e.clone()})
    q3.to(dev()).train()
    o3, d3, i3 = q3(xr)
    emb, cs, ea = e.clone(), torch.zeros(K), e.clone()
    ro, rd, ri = O.quantize_forward(xr.cpu(), emb, cs, ea, True)
    badr = i3.cpu() != ri
    mr, _ = O.quantize_margin(xr.cpu(), e)
    assert int(badr.sum()) <= 8 and not bool((badr & (mr > 1e-3)).any())
    close(d3, rd, rtol=1e-4)
    if not bool(badr.any()):
        close(q3.cluster_size, cs, rtol=1e-5, atol=1e-6)
        close(q3.embed_avg, ea, rtol=1e-4, atol=1e-4)
        close(q3.embed, emb, rtol=1e-3, atol=1e-4)


def test_blocks(amd, golden):
    g = golden("blocks")
    for tag, kind, args, xs in BLOCK_CASES:
        st = block_state(tag, kind, args)
        if kind == "resblock":
            m = amd.ResBlock(*args)
        elif kind == "encoder":
            m = amd.Encoder(*args[:4], stride=args[4])
        else:
            m = amd.Decoder(*args[:5], stride=args[5])
        m.load_state_dict(st)
        m.to(dev())
        x = t(rng.normal(SEED, f"{tag}.x", xs)).to(dev()).requires_grad_(True)
        y = m(x)
        close(y, g[f"{tag}.y"], what=f"{tag}.y")
        y.backward(t(rng.normal(SEED, f"{tag}.gy", tuple(y.shape))).to(dev()))
        close(x.grad, g[f"{tag}.gx"], what=f"{tag}.gx")
        for k, p in m.named_parameters():
            close(p.grad, g[f"{tag}.g.{k}"], rtol=5e-4, atol=1e-4, what=f"{tag}.g.{k}")


def _tiny(amd):
    cfg = O.TINY
    m = amd.VQVAE(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
                  embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    m.load_state_dict(O.make_state(cfg, SEED))
    return m.to(dev())


def test_tiny_vqvae_dropin_api_three_adam_steps(amd, golden):
    """The reference's own loop (train_vqvae.py:83-91) with stock nn.MSELoss + torch.optim.Adam."""
    g = golden("tiny_vqvae")
    m = _tiny(amd)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)
    crit = torch.nn.MSELoss()
    for step in range(3):
        img = O.make_images(2, 32, SEED + 100 * step).to(dev())
        opt.zero_grad()
        dec, diff = m(img)
        assert tuple(dec.shape) == (2, 3, 32, 32) and tuple(diff.shape) == (1,)
        recon = crit(dec, img)
        latent = diff.mean()
        loss = recon + 0.25 * latent
        loss.backward()
        if step == 0:
            close(dec, g["s0.dec"], what="dec")
            close(diff, g["s0.diff"], rtol=1e-5, atol=1e-7)
            for k, p in m.named_parameters():
                if k.startswith("dec_ir."):
                    assert p.grad is None
                else:
                    close(p.grad, g[f"s0.g.{k}"], rtol=1e-3, atol=2e-6, what=k)
        close(loss, g[f"s{step}.loss"], rtol=1e-4)
        opt.step()
        if step in (0, 2):
            for k, v in m.state_dict().items():
                if not k.startswith("dec_ir."):
                    close(v, g[f"s{step}.after.{k}"], rtol=1e-3, atol=1e-4 if step else 2e-5, what=k)


def test_tiny_vqvae_encode_decode(amd, golden):
    g = golden("tiny_vqvae")
    m = _tiny(amd).eval()
    img = O.make_images(2, 32, SEED).to(dev())
    with torch.no_grad():
        qt, qb, diff, id_t, id_b = m.encode(img)
        assert np.array_equal(id_t.cpu().numpy().astype(np.int32), g["s0.id_t"])
        assert np.array_equal(id_b.cpu().numpy().astype(np.int32), g["s0.id_b"])
        close(qt, g["s0.quant_t"], rtol=1e-5, atol=1e-5)
        close(qb, g["s0.quant_b"], rtol=1e-5, atol=1e-5)
        close(diff, g["eval.diff"], rtol=1e-5)
        dec, _ = m(img)
        dec2 = m.decode_code(id_t, id_b)          # upstream semantics of vqvae.py:251-259
        st = O.make_state(O.TINY, SEED)
        ref = O.vqvae_decode_code(st, O.TINY, id_t.cpu(), id_b.cpu())
        close(dec2, ref)
        close(dec2, dec, rtol=1e-4, atol=1e-5)      # STE value x+(q-x) vs q differ by one rounding
        assert m.embed_dim == 2 * O.TINY.embed_dim


def test_trainer_matches_oracle_three_steps(amd, golden):
    g = golden("tiny_vqvae")
    m = _tiny(amd)
    tr = amd.Stage1Trainer(m, lr=3e-4)
    for step in range(3):
        out = tr.step(O.make_images(2, 32, SEED + 100 * step).to(dev()))
        close(out["loss"], g[f"s{step}.loss"], rtol=1e-4)
        close(out["recon"], g[f"s{step}.recon"], rtol=1e-4)
        close(out["latent"], g[f"s{step}.latent"], rtol=1e-4)
        assert tr.arena.grads_ready(), "gradients must land in the flat arena (one-launch Adam)"
        if step in (0, 2):
            for k, v in m.state_dict().items():
                if not k.startswith("dec_ir."):
                    close(v, g[f"s{step}.after.{k}"], rtol=1e-3, atol=1e-4 if step else 2e-5, what=k)


def test_single_level_config1(amd, golden):
    """BASELINE config 1: Encoder(s=4) -> 1x1 -> Quantize -> Decoder(s=4), 32x32, batch 16."""
    g = golden("single_level")
    cfg = O.DEFAULT
    st = O.make_single_level_state(cfg, SEED)
    enc = amd.Encoder(3, 128, 2, 32, stride=4)
    qconv = amd.Conv2d(128, 64, 1)
    quant = amd.Quantize(64, 512)
    dec = amd.Decoder(64, 3, 128, 2, 32, stride=4)
    enc.load_state_dict({k[4:]: v for k, v in st.items() if k.startswith("enc.")})
    qconv.load_state_dict({k[len("quantize_conv."):]: v for k, v in st.items() if k.startswith("quantize_conv.")})
    quant.load_state_dict({k[len("quantize."):]: v for k, v in st.items() if k.startswith("quantize.")})
    dec.load_state_dict({k[4:]: v for k, v in st.items() if k.startswith("dec.")})
    for mod in (enc, qconv, quant, dec):
        mod.to(dev())
    img = O.make_images(16, 32, SEED).to(dev())
    h = qconv(enc(img)).permute(0, 2, 3, 1)
    q, diff, idx = quant(h)
    out = dec(q.permute(0, 3, 1, 2))
    recon = F.mse_loss(out, img)
    loss = recon + 0.25 * diff
    loss.backward()
    assert np.array_equal(idx.cpu().numpy().astype(np.int32), g["idx"])
    close(out[:2], g["dec"])
    close(loss, g["loss"], rtol=1e-4)
    for name, mod in (("enc", enc), ("quantize_conv", qconv), ("dec", dec)):
        for k, p in mod.named_parameters():
            close(p.grad.norm(), g[f"gnorm.{name}.{k}"], rtol=1e-3)
    close(quant.cluster_size, g["cluster_size_after"], rtol=1e-5, atol=1e-6)


def test_full256_default_model(amd, golden):
    """BASELINE config 2 geometry (256x256, default VQVAE) at batch 2 against the reference's outputs."""
    g = golden("full256")
    cfg = O.DEFAULT
    st = O.make_state(cfg, SEED)
    m = amd.VQVAE()
    m.load_state_dict(st)
    m.to(dev()).train()
    img = O.make_images(2, 256, SEED).to(dev())
    ids = {}
    m.quantize_t.register_forward_hook(lambda mod, i, o: ids.__setitem__("t", o[2]))
    m.quantize_b.register_forward_hook(lambda mod, i, o: ids.__setitem__("b", o[2]))
    tr = amd.Stage1Trainer(m, lr=3e-4)
    out = tr.step(img, return_dec=True)
    nt = int((ids["t"].cpu().numpy().astype(np.int16) != g["id_t"]).sum())
    nb = int((ids["b"].cpu().numpy().astype(np.int16) != g["id_b"]).sum())
    assert nt == 0 and nb == 0, f"index mismatches top={nt} bottom={nb}"
    close(out["recon"], g["recon"], rtol=1e-4)
    close(out["loss"], g["loss"], rtol=1e-4)
    close(out["dec"][:, :, ::16, ::16], g["dec_sample"], rtol=1e-3, atol=1e-4)
    for k, p in m.named_parameters():
        if p.grad is not None:
            close(p.grad.norm(), g[f"gnorm.{k}"], rtol=2e-3, what=k)
    close(m.quantize_t.cluster_size, g["cluster_size_t_after"], rtol=1e-5, atol=1e-6)
    close(m.quantize_b.cluster_size, g["cluster_size_b_after"], rtol=1e-5, atol=1e-6)


def test_baseline_batch32_properties(amd):
    """BASELINE configs[1] at its full size (256x256, batch 32): size-independent properties."""
    cfg = O.DEFAULT
    m = amd.VQVAE()
    m.load_state_dict(O.make_state(cfg, SEED))
    m.to(dev())
    img = O.make_images(32, 256, SEED).to(dev())
    tr = amd.Stage1Trainer(m, lr=3e-4)
    out = tr.step(img)
    assert torch.isfinite(out["loss"]).item()
    # EMA counts: one vote per latent vector (top 32*32*32, bottom 32*64*64)
    close(m.quantize_t.cluster_size.sum(), 0.01 * 32 * 32 * 32, rtol=1e-5)
    close(m.quantize_b.cluster_size.sum(), 0.01 * 32 * 64 * 64, rtol=1e-5)
    # batch linearity: the first 2 images reproduce the batch-2 reconstruction (no cross-image coupling)
    m2 = amd.VQVAE()
    m2.load_state_dict(O.make_state(cfg, SEED))
    m2.to(dev()).eval()
    with torch.no_grad():
        d32, _ = m2(img)
        d2, _ = m2(img[:2].contiguous())
        close(d32[:2], d2, rtol=0, atol=0)
        # encode -> decode_code round trip equals forward
        _, _, _, id_t, id_b = m2.encode(img[:4].contiguous())
        close(m2.decode_code(id_t, id_b), d32[:4], rtol=1e-4, atol=1e-5)


def test_error_behaviour(amd):
    with pytest.raises(RuntimeError):
        amd.Conv2d(8, 8, 3, padding=1).to(dev())(torch.zeros(1, 8, 4, 4))         # CPU tensor: no fallback
    with pytest.raises(NotImplementedError):
        amd.Conv2d(8, 8, 3, stride=3)
    with pytest.raises(RuntimeError):
        amd.Quantize(16, 64).to(dev())(torch.zeros(1, 2, 2, 8, device=dev()))
    d = amd._lib.ConvDesc()
    assert amd._lib.lib.vq2_conv_fwd(d, 0, None, None, None, None, 0, None, None) != 0
    assert b"conv" in amd._lib.lib.vq2_last_error()


def _thirty_steps(amd, check_indices):
    cfg = O.TINY
    st = O.make_state(cfg, 99)
    m = amd.VQVAE(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
                  embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    m.load_state_dict(st)
    m.to(dev())
    seen = {}

    def grab(key):
        def hook(mod, i, o):
            seen[key] = (i[0].detach().cpu(), o[2].cpu())
        return hook
    if check_indices:
        m.quantize_t.register_forward_hook(grab("t"))
        m.quantize_b.register_forward_hook(grab("b"))
    tr = amd.Stage1Trainer(m, lr=1e-3, sched="cycle", n_iter=100)
    adam = O.AdamState({k: v for k, v in st.items() if not O.is_buffer(k)})
    sched = O.CycleSchedule(1e-3, 100, warmup_proportion=0.05)
    got, ref, flips = [], [], 0
    for step in range(30):
        img = O.make_images(4, 32, 500 + step)
        pre = {k: getattr(m, f"quantize_{k}").embed.detach().cpu().clone() for k in ("t", "b")} if check_indices else None
        out = tr.step(img.to(dev()))
        got.append(float(out["loss"]))
        if check_indices:
            # EVERY step: each index is the fp64 argmin over the codebook this step searched (the GPU's own EMA
            # trajectory), except at fp32 near-ties
            for k in ("t", "b"):
                x, ids = seen[k]
                margin, want = O.quantize_margin(x, pre[k])
                bad = ids.reshape(-1) != want
                if bool(bad.any()):
                    scale = x.reshape(bad.numel(), -1).double().pow(2).sum(-1) + 1.0
                    assert float((margin / scale)[bad].max()) < 2e-6, f"step {step} {k}: index away from a near-tie"
                    flips += int(bad.sum())
            lr = sched.step()
            r = O.train_step(st, cfg, img, adam, lr=lr)
            ref.append(float(r["loss"]))
    torch.cuda.synchronize()
    return m, st, got, ref, flips


def test_thirty_step_trajectory_tracks_oracle(amd):
    """30 consecutive train steps (EMA codebook dynamics, Adam state, CycleScheduler): per-step indices are the exact
    argmin of the running codebook, the loss stays on the CPU oracle's trajectory, and a second run from the same
    state is bit-identical (deterministic EMA sums and split-K reductions)."""
    m, st, got, ref, flips = _thirty_steps(amd, True)
    np.testing.assert_allclose(got, ref, rtol=2e-3)
    assert got[-1] < got[0]            # it trains
    assert flips <= 4
    close(m.quantize_b.cluster_size, st["quantize_b.cluster_size"], rtol=5e-2, atol=0.05)
    m2, _, got2, _, _ = _thirty_steps(amd, False)
    assert got == got2
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), f"{k}: two identical 30-step runs differ"


def _step_vs_oracle(amd, cfg, size, batch, seed, all_elementwise=False):
    st = O.make_state(cfg, seed)
    m = amd.VQVAE(in_channel=cfg.in_channel, channel=cfg.channel, n_res_block=cfg.n_res_block,
                  n_res_channel=cfg.n_res_channel, embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    m.load_state_dict(st)
    m.to(dev())
    ids, qin = {}, {}
    embed0 = {"t": st["quantize_t.embed"].clone(), "b": st["quantize_b.embed"].clone()}

    def grab(key):
        def hook(mod, i, o):
            ids[key], qin[key] = o[2], i[0].detach().clone()
        return hook
    m.quantize_t.register_forward_hook(grab("t"))
    m.quantize_b.register_forward_hook(grab("b"))
    tr = amd.Stage1Trainer(m, lr=3e-4)
    img = O.make_images(batch, size, seed)
    out = tr.step(img.to(dev()), return_dec=True)
    adam = O.AdamState({k: v for k, v in st.items() if not O.is_buffer(k)})
    ref = O.train_step(st, cfg, img, adam)
    for key in ("t", "b"):
        got = ids[key].cpu()
        want = ref["ids"][0 if key == "t" else 1]
        bad = (got != want).reshape(-1)
        if bool(bad.any()):
            # an index may differ from the CPU path only at a genuine fp32 near-tie: the fp64 gap between the
            # two best codes (on the GPU's own quantizer input) must be below fp32 resolution of the distance
            margin, _ = O.quantize_margin(qin[key].cpu(), embed0[key])
            scale = qin[key].cpu().double().pow(2).sum(-1).reshape(-1) + 1.0
            rel = (margin / scale)[bad]
            assert int(bad.sum()) <= max(2, bad.numel() // 2000), f"{key}: {int(bad.sum())} index mismatches"
            assert float(rel.max()) < 2e-6, f"{key}: mismatch at relative fp64 margin {float(rel.max()):.3e}"
    close(out["loss"], ref["loss"], rtol=1e-4)
    exact = all(torch.equal(ids[k].cpu(), ref["ids"][0 if k == "t" else 1]) for k in ("t", "b"))
    if exact:   # a flipped near-tie changes one latent vector's code: downstream values then differ locally
        close(out["dec"], ref["dec"], rtol=1e-3, atol=1e-4)
    for k, p in m.named_parameters():
        if p.grad is not None:
            close(p.grad.norm(), ref["grads"][k].norm(), rtol=2e-3 if exact else 2e-2, what=k)
            # element-wise for the ResBlock convs (their gradients come out of the fused backward kernel and its
            # per-workgroup 1x1 slabs) and the first / last layers
            if (exact and (".conv." in k or k.startswith("enc_b.blocks.0") or k.startswith("dec.blocks.6"))) or all_elementwise:
                gref = ref["grads"][k]
                # (first-layer gradients are sums of ~65k cancelling products: their fp32 noise floor is a few 1e-4 of
                # the largest element whatever the summation order; a flipped near-tie moves ONE latent vector of the
                # batch to the neighbouring code, which perturbs a batch-mean weight gradient by ~1/M)
                close(p.grad, gref, rtol=2e-3, atol=(5e-4 if exact else 2e-3) * float(gref.abs().max()) + 1e-9,
                      what=k + " (element-wise)")
    sd = m.state_dict()
    last_bias = [k for k in st if k.startswith("dec.blocks.") and k.endswith(".bias")][-1]
    for k in ("quantize_t.cluster_size", "quantize_b.cluster_size", "quantize_b.embed_avg", "enc_b.blocks.0.weight",
              last_bias):
        got, want = sd[k].cpu(), st[k]
        if exact and k in ref["grads"]:
            # first Adam step: the update is lr * g / (|g| + 1e-8), so an element whose gradient sits at the fp32 noise
            # floor of its sum (see above) may legitimately move anywhere within +-lr; compare those within 2 * lr
            gref = ref["grads"][k]
            noisy = gref.abs() < 1e-3 * gref.abs().max()
            close(got[noisy], want[noisy], rtol=0, atol=6e-4, what=k + " (noise-floor gradients)")
            got, want = got[~noisy], want[~noisy]
        close(got, want, rtol=1e-3 if exact else 5e-2, atol=2e-5 if exact else 2e-2, what=k)


def test_config4_large_codebook_step_vs_oracle(amd):
    """BASELINE configs[3]: 256x256 with n_embed = 8192 (distance GEMM + argmin over 8192 codes)."""
    _step_vs_oracle(amd, O.VQVAEConfig(n_embed=8192), 256, 1, 31)


def test_config5_512px_step_vs_oracle(amd):
    """BASELINE configs[4] geometry: 512x512 images through the default two-level model."""
    _step_vs_oracle(amd, O.DEFAULT, 512, 1, 32)


def test_128px_step_vs_oracle_all_gradients(amd):
    """128x128 images, batch 3: the bottom level is 32x32 (rows of 32 pixels: the 4-row x 16-pair geometry and the
    64-channel tiles of the Winograd 3x3 kernel), the top level 16x16 (direct kernels only); every gradient element-wise."""
    _step_vs_oracle(amd, O.DEFAULT, 128, 3, 67, all_elementwise=True)


def test_config2_full_batch_step_vs_oracle(amd):
    """BASELINE configs[1] at the BENCH batch (256x256, default model, batch 32) against the CPU oracle, backward
    included: this is the launch geometry bench.py times -- the four-per-CU 128x128x16 dgrad instance (513..1,024
    tiles), the S = 56 / 64 / 164 split-K weight gradients and the 1,024-tile fused ResBlock backward launch -- none of
    which a batch-1..3 step selects.  (vqvae.py:81-166, train_vqvae.py:83-87.)"""
    _step_vs_oracle(amd, O.DEFAULT, 256, 32, 61, all_elementwise=True)


def _grad_batch_linearity(amd, cfg, size, batch, seed, sub=2):
    """Where the oracle at the full batch is too heavy (configs[3], configs[4]): the gradients of ONE full-batch
    Stage1Trainer.step (big-tile / split-K launch geometry) must equal the mean of the gradients of its `sub`-image
    sub-batches pushed through the small-launch path (already pinned against the oracle at batch 1..3) from the same
    state.  Legal because the loss is a mean over images (train_vqvae.py:83-85, vqvae.py:72) and the trainer defers the
    EMA update, so both sides search the same codebooks; the sub-batches run in eval mode (no EMA side effect)."""
    st = O.make_state(cfg, seed)

    def build():
        m = amd.VQVAE(n_embed=cfg.n_embed)
        m.load_state_dict(st)
        return m.to(dev())
    img = O.make_images(batch, size, seed).to(dev())
    big = build()
    tr = amd.Stage1Trainer(big, lr=3e-4)
    ids_big = {}
    big.quantize_t.register_forward_hook(lambda mod, i, o: ids_big.__setitem__("t", o[2].clone()))
    big.quantize_b.register_forward_hook(lambda mod, i, o: ids_big.__setitem__("b", o[2].clone()))
    tr.step(img)
    got = {k: p.grad.detach().clone() for k, p in big.named_parameters() if p.grad is not None}
    small = build().eval()
    acc = {k: torch.zeros_like(p, dtype=torch.float64) for k, p in small.named_parameters() if not k.startswith("dec_ir.")}
    ids_small = {"t": [], "b": []}
    small.quantize_t.register_forward_hook(lambda mod, i, o: ids_small["t"].append(o[2].clone()))
    small.quantize_b.register_forward_hook(lambda mod, i, o: ids_small["b"].append(o[2].clone()))
    for s in range(0, batch, sub):
        small.zero_grad(set_to_none=True)
        x = img[s:s + sub].contiguous()
        dec, diff = small(x)
        loss, _, _ = amd.stage1_loss(dec, diff, x)
        loss.backward()
        for k, p in small.named_parameters():
            if p.grad is not None:
                acc[k] += p.grad.double()
    # same codes on both sides (the search is per vector and does not depend on the launch size)
    for key in ("t", "b"):
        assert torch.equal(ids_big[key], torch.cat(ids_small[key], 0)), f"{key}: indices depend on the batch size"
    assert set(got) == set(acc)
    for k, g in got.items():
        want = (acc[k] / (batch // sub)).float()
        # different split-K trees over up to 2^19 rows: element error ~ sqrt(rows) * eps of the largest partial sum
        close(g, want, rtol=1e-3, atol=2e-4 * float(want.abs().max()) + 1e-10, what=k + " (batch linearity)")


def test_config4_gradient_batch_linearity(amd):
    """BASELINE configs[3] (n_embed 8192) at its bench batch 32: full-batch gradients == mean of 2-image gradients."""
    _grad_batch_linearity(amd, O.VQVAEConfig(n_embed=8192), 256, 32, 53)


def test_config5_gradient_batch_linearity(amd):
    """BASELINE configs[4] (512x512) at its per-GPU batch 8."""
    _grad_batch_linearity(amd, O.DEFAULT, 512, 8, 54)


def test_quantize_8192_full_size(amd):
    """The kernel instance behind configs[3]: 512-vector workgroups looping over 16 LDS tiles of 512 codes
    (M = 131,072 >= 512*256, K = 8192).  Checked against the fp64 argmin in row blocks, the oracle's
    scatter statistics, count conservation and a ragged tail."""
    D, K = 64, 8192
    x = t(rng.normal(21, "c4.x", (32, 64, 64, D)))
    x = torch.cat([x.reshape(-1, D), x.reshape(-1, D)[:77] * 0.25], 0).contiguous()      # M = 131,149 (ragged)
    M = x.shape[0]
    e = t(rng.normal(21, "c4.e", (D, K)))
    q = amd.Quantize(D, K)
    q.load_state_dict({"embed": e, "cluster_size": torch.zeros(K), "embed_avg": e.clone()})
    q.to(dev()).train()
    out, diff, idx = q(x.to(dev()).reshape(M, 1, 1, D))
    idx = idx.cpu().reshape(-1)
    margin, ref_idx = O.quantize_margin_chunked(x, e)
    bad = idx != ref_idx
    scale = x.double().pow(2).sum(-1) + 1.0
    assert int(bad.sum()) <= max(2, M // 2000), f"{int(bad.sum())} index mismatches vs the fp64 argmin"
    assert not bool(bad.any()) or float((margin / scale)[bad].max()) < 2e-6
    # gather / STE output / loss on the GPU's own indices
    code = F.embedding(idx, e.t())
    close(out.reshape(M, D), x + (code - x), rtol=1e-6, atol=1e-6)
    close(diff, (code - x).pow(2).mean(), rtol=1e-4)
    # statistics: exact counts, sums vs the oracle's index_add, full EMA update
    counts, sums = O.quantize_stats(x, idx, K)
    assert float(counts.sum()) == M
    close(q.cluster_size, 0.01 * counts, rtol=1e-6, atol=0)
    cs, ea, emb = torch.zeros(K), e.clone(), e.clone()
    O.ema_update_(emb, cs, ea, counts, sums)
    close(q.embed_avg, ea, rtol=5e-5, atol=1e-5)
    close(q.embed, emb, rtol=1e-4, atol=1e-5)


def _fullsize_step_checks(amd, cfg, size, batch, seed, sub=2):
    """One Stage1Trainer.step at a BASELINE config's full per-GPU batch: every index against the fp64 argmin of
    the GPU's own quantizer inputs (row blocks), exact EMA counts + ordered sums against the oracle's scatter,
    loss terms recomputed on the host, and the first `sub` images against the CPU oracle (images are independent)."""
    st = O.make_state(cfg, seed)
    m = amd.VQVAE(n_embed=cfg.n_embed)
    m.load_state_dict(st)
    m.to(dev())
    ids, qin = {}, {}

    def grab(key):
        def hook(mod, i, o):
            ids[key], qin[key] = o[2].cpu(), i[0].detach().cpu()
        return hook
    m.quantize_t.register_forward_hook(grab("t"))
    m.quantize_b.register_forward_hook(grab("b"))
    tr = amd.Stage1Trainer(m, lr=3e-4)
    img = O.make_images(batch, size, seed)
    out = tr.step(img.to(dev()), return_dec=True)
    dec = out["dec"].cpu()
    latent = 0.0
    for key in ("t", "b"):
        e0 = st[f"quantize_{key}.embed"]
        x, got = qin[key], ids[key].reshape(-1)
        M = got.numel()
        margin, ref_idx = O.quantize_margin_chunked(x, e0)
        bad = got != ref_idx
        if bool(bad.any()):
            scale = x.reshape(M, -1).double().pow(2).sum(-1) + 1.0
            assert int(bad.sum()) <= max(2, M // 2000), f"{key}: {int(bad.sum())} index mismatches"
            assert float((margin / scale)[bad].max()) < 2e-6, f"{key}: mismatch away from an fp32 near-tie"
        counts, sums = O.quantize_stats(x, got, cfg.n_embed)
        cs, ea, emb = torch.zeros(cfg.n_embed), e0.clone(), e0.clone()
        O.ema_update_(emb, cs, ea, counts, sums)
        sd = m.state_dict()
        assert float(counts.sum()) == M
        close(sd[f"quantize_{key}.cluster_size"], cs, rtol=1e-6, atol=0, what=f"{key}.cluster_size")
        # embed_avg: exact to 1e-5 wherever a code collects a modest number of rows; a code that sums >= 1e4 rows may
        # differ by reassociation (fixed summation tree here, index_add's sequential order in the oracle: error
        # ~ sqrt(rows) * eps of the sum) -- asserted as exactly that: every element beyond 1e-5 belongs to such a code
        got_ea = sd[f"quantize_{key}.embed_avg"].cpu()
        err = (got_ea - ea).abs()
        loose = err > 1e-5 * ea.abs() + 1e-5
        if bool(loose.any()):
            heavy = counts >= 1e4
            assert bool(heavy[loose.any(0)].all()), f"{key}.embed_avg: an element of a lightly used code is off"
        close(got_ea, ea, rtol=5e-5, atol=1e-5, what=f"{key}.embed_avg")
        close(sd[f"quantize_{key}.embed"], emb, rtol=1e-4, atol=1e-5, what=f"{key}.embed")
        code = F.embedding(got.reshape(x.shape[:-1]), e0.t())
        latent = latent + float((code - x).double().pow(2).mean())
    close(out["latent"], latent, rtol=1e-4)
    close(out["recon"], float((dec.double() - img.double()).pow(2).mean()), rtol=1e-4)
    close(out["loss"], float(out["recon"]) + 0.25 * float(out["latent"]), rtol=1e-5)
    # the first images against the CPU oracle (eval forward: same codebook, no EMA side effects)
    ref = O.vqvae_forward(st, cfg, img[:sub], training=False)
    same = True
    for key, want in (("t", ref[2]), ("b", ref[3])):
        got = ids[key][:sub]
        badk = (got != want).reshape(-1)
        if bool(badk.any()):
            same = False
            mg, _ = O.quantize_margin_chunked(qin[key][:sub], st[f"quantize_{key}.embed"])
            sc = qin[key][:sub].reshape(badk.numel(), -1).double().pow(2).sum(-1) + 1.0
            assert float((mg / sc)[badk].max()) < 2e-6, f"{key}: oracle index mismatch away from a near-tie"
    if same:
        close(dec[:sub], ref[0], rtol=1e-3, atol=1e-4)
    return m


def test_config4_full_batch_step(amd):
    """BASELINE configs[3] at its bench batch (256x256, n_embed 8192, batch 32): the <64,16,512> kernel."""
    _fullsize_step_checks(amd, O.VQVAEConfig(n_embed=8192), 256, 32, 51)


def test_config5_full_batch_step(amd):
    """BASELINE configs[4] at its per-GPU batch (512x512, batch 8)."""
    _fullsize_step_checks(amd, O.DEFAULT, 512, 8, 52)


def _two_step_state(amd, cfg, size, batch, seed):
    m = amd.VQVAE(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
                  embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    m.load_state_dict(O.make_state(cfg, seed))
    m.to(dev())
    tr = amd.Stage1Trainer(m, lr=3e-4)
    for s in range(2):
        tr.step(O.make_images(batch, size, seed + s).to(dev()))
    torch.cuda.synchronize()
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("case", ["tiny", "default256"])
def test_train_step_is_bit_reproducible(amd, case):
    """No float atomics anywhere on the path (split-K slabs and the EMA sums are reduced in a fixed order): two
    runs from the same state on the same inputs give bit-identical parameters AND codebook buffers."""
    cfg, size, batch = (O.TINY, 32, 4) if case == "tiny" else (O.DEFAULT, 256, 8)
    a = _two_step_state(amd, cfg, size, batch, 77)
    b = _two_step_state(amd, cfg, size, batch, 77)
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k} differs between two identical runs"


@pytest.mark.parametrize("n_res_block", [0, 3])
def test_block_count_variants_step_vs_oracle(amd, n_res_block):
    """Encoder/Decoder without ResBlocks (the trailing ReLU then fuses into a conv and its backward mask into the
    consumers' dgrad launches) and with three of them: one train step vs the oracle, element-wise gradients."""
    cfg = O.VQVAEConfig(channel=32, n_res_block=n_res_block, n_res_channel=8, embed_dim=16, n_embed=64)
    _step_vs_oracle(amd, cfg, 32, 3, 41 + n_res_block)
