"""Generate tests/golden/*.npz by running the REFERENCE's own vqvae.py on CPU.

Runs only in the build container (needs /root/reference, read-only).  The
reference never travels: only inputs-by-seed and its numeric outputs are stored.
All inputs/weights come from oracle/rng.py so the fixtures hold outputs only.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = os.environ.get("VQ2_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import vqvae as ref  # noqa: E402  (the reference module)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rng  # noqa: E402
from oracle import vqvae_oracle as O  # noqa: E402
from oracle.make_golden_cases import (BLOCK_CASES, CONV_FLAVOURS, DEEP_ADAIN_CASES, DEEP_BLOCK_CASES,  # noqa: E402
                                      DEEP_CONV_FLAVOURS,
                                      DEEP_EMBED_SCALE, DEEP_GAIN, DEEP_SEED, SCHED_CASES, SEED, block_state, conv_inputs,
                                      quantize_inputs, thin)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def n(x):
    return x.detach().cpu().numpy().copy()


# ---------------------------------------------------------------- Quantize
def quantize_case(tag, D, K, xshape, training, tie=False, store_full=True):
    q = ref.Quantize(D, K)
    x, embed, cs0, gw = quantize_inputs(tag, D, K, xshape, tie)
    q.embed.copy_(t(embed))
    q.embed_avg.copy_(t(embed) * t(cs0)[None, :])
    q.cluster_size.copy_(t(cs0))
    q.train(training)
    xt = t(x).clone().requires_grad_(True)
    out, diff, idx = q(xt)
    loss = (out * t(gw)).sum() + 0.25 * diff
    loss.backward()
    d = {f"{tag}.idx": n(idx).astype(np.int32), f"{tag}.diff": n(diff)}
    if store_full:
        d.update({f"{tag}.out": n(out), f"{tag}.xgrad": n(xt.grad),
                  f"{tag}.embed_after": n(q.embed), f"{tag}.cluster_size_after": n(q.cluster_size),
                  f"{tag}.embed_avg_after": n(q.embed_avg)})
    else:
        d.update({f"{tag}.out_rows": n(out).reshape(-1, D)[::16],
                  f"{tag}.xgrad_rows": n(xt.grad).reshape(-1, D)[::16],
                  f"{tag}.cluster_size_after": n(q.cluster_size),
                  f"{tag}.embed_after_cols": n(q.embed)[:, ::64]})
    return d


def gen_quantize():
    d = {}
    d.update(quantize_case("q512_train", 64, 512, (2, 8, 8, 64), True))
    d.update(quantize_case("q512_eval", 64, 512, (2, 8, 8, 64), False))
    d.update(quantize_case("q512_tie", 64, 512, (2, 8, 8, 64), True, tie=True))
    d.update(quantize_case("q8192_train", 64, 8192, (2, 16, 16, 64), True, store_full=False))
    d.update(quantize_case("q64_train", 16, 64, (2, 4, 4, 16), True))
    np.savez_compressed(os.path.join(OUT, "quantize.npz"), **d)


# ---------------------------------------------------------------- conv flavours
def gen_convs():
    import torch.nn.functional as F
    d = {}
    for tag, kind, ws, stride, pad, hw in CONV_FLAVOURS:
        x, w, b = (t(a).requires_grad_(True) for a in conv_inputs(tag, kind, ws, hw))
        if kind == "conv":
            y = F.conv2d(x, w, b, stride=stride, padding=pad)
        else:
            y = F.conv_transpose2d(x, w, b, stride=stride, padding=pad)
        gy = t(rng.normal(SEED, f"{tag}.gy", tuple(y.shape)))
        y.backward(gy)
        d[f"{tag}.y"] = n(y)
        d[f"{tag}.gx"] = n(x.grad)
        d[f"{tag}.gw"] = n(w.grad)
        d[f"{tag}.gb"] = n(b.grad)
    np.savez_compressed(os.path.join(OUT, "convs.npz"), **d)


# ---------------------------------------------------------------- blocks
def gen_blocks():
    d = {}
    for tag, kind, args, xs in BLOCK_CASES:
        if kind == "resblock":
            m = ref.ResBlock(*args)
        elif kind == "encoder":
            m = ref.Encoder(*args[:4], stride=args[4])
        else:
            m = ref.Decoder(*args[:5], stride=args[5])
        st = block_state(tag, kind, args)
        assert list(m.state_dict().keys()) == list(st.keys()), tag
        m.load_state_dict(st)
        x = t(rng.normal(SEED, f"{tag}.x", xs)).requires_grad_(True)
        y = m(x)
        gy = t(rng.normal(SEED, f"{tag}.gy", tuple(y.shape)))
        y.backward(gy)
        d[f"{tag}.y"] = n(y)
        d[f"{tag}.gx"] = n(x.grad)
        for k, p in m.named_parameters():
            d[f"{tag}.g.{k}"] = n(p.grad)
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **d)


# ---------------------------------------------------------------- tiny VQVAE, 3 Adam steps
def gen_tiny():
    cfg = O.TINY
    m = ref.VQVAE(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
                  embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    st = O.make_state(cfg, SEED)
    assert list(m.state_dict().keys()) == list(st.keys())
    m.load_state_dict(st)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)
    d = {}
    for step in range(3):
        img = O.make_images(2, 32, SEED + 100 * step)
        opt.zero_grad()
        dec, diff = m(img)
        recon = torch.nn.functional.mse_loss(dec, img)
        latent = diff.mean()
        loss = recon + 0.25 * latent
        loss.backward()
        if step == 0:
            d["s0.dec"] = n(dec)
            d["s0.diff"] = n(diff)
            for k, p in m.named_parameters():
                if p.grad is not None:
                    d[f"s0.g.{k}"] = n(p.grad)
            d["s0.nograd"] = np.array([k for k, p in m.named_parameters() if p.grad is None])
        d[f"s{step}.loss"] = n(loss)
        d[f"s{step}.recon"] = n(recon)
        d[f"s{step}.latent"] = n(latent)
        opt.step()
        if step in (0, 2):
            for k, v in m.state_dict().items():
                if not k.startswith("dec_ir."):
                    d[f"s{step}.after.{k}"] = n(v)
    # ids of step 0: fresh model, same weights, eval mode (indices do not depend on the mode)
    m2 = ref.VQVAE(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
                   embed_dim=cfg.embed_dim, n_embed=cfg.n_embed)
    m2.load_state_dict(O.make_state(cfg, SEED))
    m2.eval()
    with torch.no_grad():
        qt, qb, diff, id_t, id_b = m2.encode(O.make_images(2, 32, SEED))
    d["s0.id_t"] = n(id_t).astype(np.int32)
    d["s0.id_b"] = n(id_b).astype(np.int32)
    d["s0.quant_t"] = n(qt)
    d["s0.quant_b"] = n(qb)
    d["eval.diff"] = n(diff)
    np.savez_compressed(os.path.join(OUT, "tiny_vqvae.npz"), **d)


# ---------------------------------------------------------------- config 1 (single level)
def gen_single_level():
    cfg = O.DEFAULT
    st = O.make_single_level_state(cfg, SEED)
    enc = ref.Encoder(3, 128, 2, 32, stride=4)
    qconv = torch.nn.Conv2d(128, 64, 1)
    quant = ref.Quantize(64, 512)
    dec = ref.Decoder(64, 3, 128, 2, 32, stride=4)
    enc.load_state_dict({k[4:]: v for k, v in st.items() if k.startswith("enc.")})
    qconv.load_state_dict({k[len("quantize_conv."):]: v for k, v in st.items() if k.startswith("quantize_conv.")})
    quant.load_state_dict({k[len("quantize."):]: v for k, v in st.items() if k.startswith("quantize.")})
    dec.load_state_dict({k[4:]: v for k, v in st.items() if k.startswith("dec.")})
    img = O.make_images(16, 32, SEED)
    h = qconv(enc(img)).permute(0, 2, 3, 1)
    q, diff, idx = quant(h)
    out = dec(q.permute(0, 3, 1, 2))
    recon = torch.nn.functional.mse_loss(out, img)
    loss = recon + 0.25 * diff
    loss.backward()
    d = {"dec": n(out)[:2], "diff": n(diff), "idx": n(idx).astype(np.int32), "loss": n(loss),
         "recon": n(recon)}
    for name, mod in (("enc", enc), ("quantize_conv", qconv), ("dec", dec)):
        for k, p in mod.named_parameters():
            d[f"gnorm.{name}.{k}"] = n(p.grad.norm())
    d["cluster_size_after"] = n(quant.cluster_size)
    np.savez_compressed(os.path.join(OUT, "single_level.npz"), **d)


# ---------------------------------------------------------------- full-size 256^2, B=2
def gen_full256():
    cfg = O.DEFAULT
    m = ref.VQVAE()
    m.load_state_dict(O.make_state(cfg, SEED))
    m.train()
    img = O.make_images(2, 256, SEED)
    # ids via hooks on the two Quantize modules
    ids = {}
    m.quantize_t.register_forward_hook(lambda mod, i, o: ids.__setitem__("t", o[2]))
    m.quantize_b.register_forward_hook(lambda mod, i, o: ids.__setitem__("b", o[2]))
    dec, diff = m(img)
    recon = torch.nn.functional.mse_loss(dec, img)
    loss = recon + 0.25 * diff.mean()
    loss.backward()
    d = {"id_t": n(ids["t"]).astype(np.int16), "id_b": n(ids["b"]).astype(np.int16),
         "diff": n(diff), "recon": n(recon), "loss": n(loss),
         "dec_sample": n(dec)[:, :, ::16, ::16]}
    for k, p in m.named_parameters():
        if p.grad is not None:
            d[f"gnorm.{k}"] = n(p.grad.norm())
    d["cluster_size_t_after"] = n(m.quantize_t.cluster_size)
    d["cluster_size_b_after"] = n(m.quantize_b.cluster_size)
    np.savez_compressed(os.path.join(OUT, "full256.npz"), **d)


def gen_deep():
    """VQVAE_Deep (vqvae_deep.py): the tiny configuration end to end (encode -> quantize -> upsample/cat ->
    decode(quant, style) -> MSE + 0.25*latent -> all gradients), AdaIN alone, the extra conv flavours, an
    embed_dim-256 Quantize, and the default model's state_dict keys/shapes."""
    import torch.nn.functional as F
    import vqvae_deep as refd           # /root/reference/vqvae_deep.py
    from oracle import vqvae_deep_oracle as OD
    d = {}
    # --- layout of the default model
    big = refd.VQVAE_Deep()
    d["default.keys"] = np.array(list(big.state_dict().keys()))
    d["default.shapes"] = np.array([list(v.shape) + [0] * (4 - v.dim()) for v in big.state_dict().values()], np.int64)
    d["default.n_params"] = np.array(sum(p.numel() for p in big.parameters()), np.int64)
    del big
    # --- conv flavours
    for tag, kind, ws, stride, pad, hw in DEEP_CONV_FLAVOURS:
        x, w, b = (t(a).requires_grad_(True) for a in conv_inputs(tag, kind, ws, hw))
        y = (F.conv2d(x, w, b, stride=stride, padding=pad) if kind == "conv" else
             F.conv_transpose2d(x, w, b, stride=stride, padding=pad))
        y.backward(t(rng.normal(SEED, f"{tag}.gy", tuple(y.shape))))
        d.update({f"{tag}.y": n(y), f"{tag}.gx": n(x.grad), f"{tag}.gw": thin(n(w.grad)), f"{tag}.gb": n(b.grad)})
    # --- Quantize with embed_dim 256
    d.update(quantize_case("q256_train", 256, 512, (2, 8, 8, 256), True))
    # --- AdaIN (+ the ReLU that follows it in AdainResBlk)
    for tag, sd, c, (nb, h, w) in DEEP_ADAIN_CASES:
        m = refd.AdaIN(sd, c)
        wt = rng.uniform(DEEP_SEED, f"{tag}.fc.w", (2 * c, sd), -1, 1) / np.sqrt(sd)
        m.fc.weight.data.copy_(t(wt.astype(np.float32)))
        m.fc.bias.data.copy_(t(rng.uniform(DEEP_SEED, f"{tag}.fc.b", (2 * c,), -0.5, 0.5)))
        x = t(rng.normal(DEEP_SEED, f"{tag}.x", (nb, c, h, w)) * 1.5 + 0.3).requires_grad_(True)
        s = t(rng.normal(DEEP_SEED, f"{tag}.s", (nb, sd))).requires_grad_(True)
        y = F.relu(m(x, s))
        y.backward(t(rng.normal(DEEP_SEED, f"{tag}.gy", (nb, c, h, w))))
        d.update({f"{tag}.y": n(y), f"{tag}.gx": n(x.grad), f"{tag}.gs": n(s.grad), f"{tag}.gw": thin(n(m.fc.weight.grad)),
                  f"{tag}.gb": n(m.fc.bias.grad)})
    # --- Encoder / Decoder at the strides VQVAE_Deep does not use itself
    for tag, kind, args, xs in DEEP_BLOCK_CASES:
        m = refd.Encoder(*args) if kind == "encoder" else refd.Decoder(*args)
        sd = m.state_dict()
        for k in sd:    # deterministic weights from the counter RNG, same rule as the test side (deep_block_state)
            shape = tuple(sd[k].shape)
            fan = int(np.prod(shape[1:])) if len(shape) > 1 else 16
            sd[k] = t((rng.uniform(DEEP_SEED, f"{tag}.{k}", shape, -1, 1) / np.sqrt(fan)).astype(np.float32))
        m.load_state_dict(sd)
        x = t(rng.normal(DEEP_SEED, f"{tag}.x", xs)).requires_grad_(True)
        styled = kind == "decoder" and args[3] > 1
        sty = t(rng.normal(DEEP_SEED, f"{tag}.s", (xs[0], args[3]))).requires_grad_(True) if styled else None
        y = m(x, sty) if styled else m(x)
        y.backward(t(rng.normal(DEEP_SEED, f"{tag}.gy", tuple(y.shape))))
        d[f"{tag}.keys"] = np.array(list(sd.keys()))
        d[f"{tag}.y"] = n(y)
        d[f"{tag}.gx"] = n(x.grad)
        if styled:
            d[f"{tag}.gs"] = n(sty.grad)
        for k, p_ in m.named_parameters():
            if p_.grad is not None:
                d[f"{tag}.g.{k}"] = n(p_.grad)
    # --- the tiny model
    cfg = OD.DEEP_TINY
    m = refd.VQVAE_Deep(channel=cfg.channel, n_res_block=cfg.n_res_block, n_res_channel=cfg.n_res_channel,
                        embed_dim=cfg.embed_dim, n_embed=cfg.n_embed, style_dim=cfg.style_dim)
    st = OD.make_deep_state(cfg, DEEP_SEED, DEEP_EMBED_SCALE, DEEP_GAIN)
    assert list(st.keys()) == list(m.state_dict().keys())
    m.load_state_dict(st)
    m.train()
    img = O.make_images(2, 32, DEEP_SEED)
    style = OD.make_style(2, cfg, DEEP_SEED).requires_grad_(True)
    enc_b, enc_t = m.encode(img)
    qt, qb, diff, id_t, id_b = m.quantize(enc_b, enc_t)
    quant = torch.cat([m.upsample_t(qt), qb], 1)
    dec = m.decode(quant, style)
    recon = F.mse_loss(dec, img)
    loss = recon + 0.25 * diff.mean()
    loss.backward()
    d.update({"tiny.enc_b": n(enc_b), "tiny.enc_t": n(enc_t), "tiny.quant_t": n(qt), "tiny.quant_b": n(qb),
              "tiny.diff": n(diff), "tiny.id_t": n(id_t).astype(np.int32), "tiny.id_b": n(id_b).astype(np.int32),
              "tiny.quant": n(quant), "tiny.dec": n(dec), "tiny.recon": n(recon), "tiny.loss": n(loss),
              "tiny.g.style": n(style.grad)})
    for k, p in m.named_parameters():
        if p.grad is not None:
            d[f"tiny.g.{k}"] = n(p.grad)
    for k in ("quantize_t.cluster_size", "quantize_b.cluster_size", "quantize_t.embed_avg", "quantize_b.embed",
              "quantize_t.embed"):
        d[f"tiny.after.{k}"] = n(m.state_dict()[k])
    try:        # the fork's forward() calls decode() without its style argument (vqvae_deep.py:277)
        m(img)
        d["forward_raises"] = np.array("")
    except TypeError as e:
        d["forward_raises"] = np.array(str(e))
    np.savez_compressed(os.path.join(OUT, "deep.npz"), **d)


def gen_scheduler():
    """(lr, momentum) trajectories of the reference's CycleScheduler (scheduler.py:251-320) driving a stock Adam."""
    import scheduler as ref_sched   # /root/reference/scheduler.py
    d = {}
    for tag, kw, steps in SCHED_CASES:
        opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        s = ref_sched.CycleScheduler(opt, **kw)
        lrs, moms, group_lr, group_b1 = [], [], [], []
        for _ in range(steps):
            lr, mom = s.step()
            lrs.append(lr)
            moms.append(np.nan if mom is None else mom)
            group_lr.append(opt.param_groups[0]["lr"])
            group_b1.append(opt.param_groups[0]["betas"][0])
        d[f"{tag}.lr"] = np.asarray(lrs, np.float64)
        d[f"{tag}.momentum"] = np.asarray(moms, np.float64)
        d[f"{tag}.group_lr"] = np.asarray(group_lr, np.float64)
        d[f"{tag}.group_beta1"] = np.asarray(group_b1, np.float64)
    np.savez_compressed(os.path.join(OUT, "scheduler.npz"), **d)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    which = sys.argv[1:] or ["quantize", "convs", "blocks", "tiny", "single", "full", "scheduler", "deep"]
    if "quantize" in which:
        gen_quantize()
    if "convs" in which:
        gen_convs()
    if "blocks" in which:
        gen_blocks()
    if "tiny" in which:
        gen_tiny()
    if "single" in which:
        gen_single_level()
    if "full" in which:
        gen_full256()
    if "scheduler" in which:
        gen_scheduler()
    if "deep" in which:
        gen_deep()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
