"""CPU: the oracle (oracle/vqvae_oracle.py) against golden vectors produced by the
reference's own vqvae.py (oracle/make_golden.py).  This is what pins the oracle."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import rng
from oracle import vqvae_oracle as O
from oracle.make_golden_cases import (BLOCK_CASES, CONV_FLAVOURS, SEED, block_state, conv_inputs,
                                      quantize_inputs)

TOL = dict(rtol=1e-5, atol=1e-6)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, **kw):
    kw = {**TOL, **kw}
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), **kw)


def run_quantize(tag, D, K, xshape, training, tie=False):
    x, embed, cs0, gw = quantize_inputs(tag, D, K, xshape, tie)
    e = t(embed).clone()
    cs = t(cs0).clone()
    ea = (t(embed) * t(cs0)[None, :]).clone()
    xt = t(x).clone().requires_grad_(True)
    out, diff, idx = O.quantize_forward(xt, e, cs, ea, training)
    ((out * t(gw)).sum() + 0.25 * diff).backward()
    return out, diff, idx, xt.grad, e, cs, ea


def test_quantize_train_eval_tie(golden):
    g = golden("quantize")
    for tag, D, K, xs, tr, tie in [("q512_train", 64, 512, (2, 8, 8, 64), True, False),
                                   ("q512_eval", 64, 512, (2, 8, 8, 64), False, False),
                                   ("q512_tie", 64, 512, (2, 8, 8, 64), True, True),
                                   ("q64_train", 16, 64, (2, 4, 4, 16), True, False)]:
        out, diff, idx, gx, e, cs, ea = run_quantize(tag, D, K, xs, tr, tie)
        assert np.array_equal(idx.numpy().astype(np.int32), g[f"{tag}.idx"]), tag
        close(out.detach(), g[f"{tag}.out"])
        close(diff.detach(), g[f"{tag}.diff"])
        close(gx, g[f"{tag}.xgrad"])
        close(e, g[f"{tag}.embed_after"])
        close(cs, g[f"{tag}.cluster_size_after"])
        close(ea, g[f"{tag}.embed_avg_after"])
    gi = g["q512_tie.idx"].reshape(-1)
    assert gi[0] == 5 and gi[1] == 64 and gi[2] == 5  # first index wins on exact ties


def test_quantize_8192(golden):
    g = golden("quantize")
    out, diff, idx, gx, e, cs, ea = run_quantize("q8192_train", 64, 8192, (2, 16, 16, 64), True)
    assert np.array_equal(idx.numpy().astype(np.int32), g["q8192_train.idx"])
    close(out.detach().reshape(-1, 64)[::16], g["q8192_train.out_rows"])
    close(gx.reshape(-1, 64)[::16], g["q8192_train.xgrad_rows"])
    close(cs, g["q8192_train.cluster_size_after"])
    close(e[:, ::64], g["q8192_train.embed_after_cols"], rtol=1e-4)


def test_quantize_stats_match_onehot_gemm():
    x = t(rng.normal(1, "s.x", (3, 5, 7, 16)))
    e = t(rng.normal(1, "s.e", (16, 32)))
    _, _, idx = O.quantize_forward(x, e.clone(), torch.zeros(32), e.clone(), False)
    counts, sums = O.quantize_stats(x, idx, 32)
    oh = F.one_hot(idx.reshape(-1), 32).float()
    close(counts, oh.sum(0))
    close(sums, x.reshape(-1, 16).t() @ oh, rtol=1e-5, atol=1e-5)


def test_conv_flavours(golden):
    g = golden("convs")
    for tag, kind, ws, stride, pad, hw in CONV_FLAVOURS:
        x, w, b = (t(a).requires_grad_(True) for a in conv_inputs(tag, kind, ws, hw))
        fn = F.conv2d if kind == "conv" else F.conv_transpose2d
        y = fn(x, w, b, stride=stride, padding=pad)
        y.backward(t(rng.normal(SEED, f"{tag}.gy", tuple(y.shape))))
        close(y.detach(), g[f"{tag}.y"])
        close(x.grad, g[f"{tag}.gx"])
        close(w.grad, g[f"{tag}.gw"], atol=1e-5)
        close(b.grad, g[f"{tag}.gb"], atol=1e-5)


def test_blocks(golden):
    g = golden("blocks")
    for tag, kind, args, xs in BLOCK_CASES:
        st = block_state(tag, kind, args)
        leaves = {k: v.clone().requires_grad_(True) for k, v in st.items()}
        x = t(rng.normal(SEED, f"{tag}.x", xs)).requires_grad_(True)
        if kind == "resblock":
            y = O.resblock({f"rb.{k}": v for k, v in leaves.items()}, "rb", x)
        elif kind == "encoder":
            y = O.encoder({f"m.{k}": v for k, v in leaves.items()}, "m", x, args[2], args[4])
        else:
            y = O.decoder({f"m.{k}": v for k, v in leaves.items()}, "m", x, args[3], args[5])
        y.backward(t(rng.normal(SEED, f"{tag}.gy", tuple(y.shape))))
        close(y.detach(), g[f"{tag}.y"])
        close(x.grad, g[f"{tag}.gx"])
        for k, v in leaves.items():
            close(v.grad, g[f"{tag}.g.{k}"], atol=1e-5)


def test_state_spec_matches_reference_layout(golden):
    g = golden("tiny_vqvae")
    keys = [k[len("s0.after."):] for k in g.files if k.startswith("s0.after.")]
    spec = O.state_spec(O.TINY)
    live = [k for k in spec if not k.startswith("dec_ir.")]
    assert sorted(keys) == sorted(live)
    for k in live:
        assert tuple(g[f"s0.after.{k}"].shape) == tuple(spec[k]), k
    assert len(O.state_spec(O.DEFAULT)) == 86
    n_all = sum(int(np.prod(s)) for k, s in O.state_spec(O.DEFAULT).items() if not O.is_buffer(k))
    n_live = sum(int(np.prod(s)) for k, s in O.state_spec(O.DEFAULT).items() if O.is_live_param(k))
    assert n_all == 1833092 and n_live == 1388867


def test_tiny_vqvae_three_adam_steps(golden):
    g = golden("tiny_vqvae")
    cfg = O.TINY
    st = O.make_state(cfg, SEED)
    adam = O.AdamState({k: v for k, v in st.items() if not O.is_buffer(k)})
    for step in range(3):
        img = O.make_images(2, 32, SEED + 100 * step)
        r = O.train_step(st, cfg, img, adam)
        close(r["loss"], g[f"s{step}.loss"])
        close(r["recon"], g[f"s{step}.recon"])
        close(r["latent"], g[f"s{step}.latent"])
        if step == 0:
            close(r["dec"], g["s0.dec"])
            close(r["diff"], g["s0.diff"])
            assert np.array_equal(r["ids"][0].numpy().astype(np.int32), g["s0.id_t"])
            assert np.array_equal(r["ids"][1].numpy().astype(np.int32), g["s0.id_b"])
            for k, gr in r["grads"].items():
                close(gr, g[f"s0.g.{k}"], atol=1e-6)
            assert all(k.startswith("dec_ir.") for k in g["s0.nograd"])
        if step in (0, 2):
            for k, v in st.items():
                if not k.startswith("dec_ir."):
                    close(v, g[f"s{step}.after.{k}"], rtol=2e-5, atol=2e-6)


def test_single_level_config1(golden):
    g = golden("single_level")
    cfg = O.DEFAULT
    st = O.make_single_level_state(cfg, SEED)
    leaves = {k: (v.clone().requires_grad_(True) if not O.is_buffer(k) else v.clone()) for k, v in st.items()}
    img = O.make_images(16, 32, SEED)
    dec, diff, idx = O.single_level_forward(leaves, cfg, img, True)
    loss, recon, latent = O.stage1_loss(dec, diff, img)
    loss.backward()
    assert np.array_equal(idx.numpy().astype(np.int32), g["idx"])
    close(dec.detach()[:2], g["dec"])
    close(loss.detach(), g["loss"])
    close(recon.detach(), g["recon"])
    for k, v in leaves.items():
        if not O.is_buffer(k):
            close(v.grad.norm(), g[f"gnorm.{k}"], rtol=1e-4)
    close(leaves["quantize.cluster_size"], g["cluster_size_after"])


def test_full256_default(golden):
    g = golden("full256")
    cfg = O.DEFAULT
    st = O.make_state(cfg, SEED)
    adam = O.AdamState({k: v for k, v in st.items() if not O.is_buffer(k)})
    img = O.make_images(2, 256, SEED)
    r = O.train_step(st, cfg, img, adam)
    assert np.array_equal(r["ids"][0].numpy().astype(np.int16), g["id_t"])
    assert np.array_equal(r["ids"][1].numpy().astype(np.int16), g["id_b"])
    close(r["diff"], g["diff"])
    close(r["recon"], g["recon"])
    close(r["loss"], g["loss"])
    close(r["dec"][:, :, ::16, ::16], g["dec_sample"])
    for k, gr in r["grads"].items():
        close(gr.norm(), g[f"gnorm.{k}"], rtol=1e-4)
    close(st["quantize_t.cluster_size"], g["cluster_size_t_after"])
    close(st["quantize_b.cluster_size"], g["cluster_size_b_after"])


def test_cycle_schedule_shape():
    s = O.CycleSchedule(3e-4, 1000, warmup_proportion=0.05)
    lrs = [s.step() for _ in range(1000)]
    assert abs(lrs[49] - 3e-4) < 1e-12            # end of linear warm-up (scheduler.py:231-248)
    assert abs(lrs[0] - (3e-4 / 25 + (1 / 50) * (3e-4 - 3e-4 / 25))) < 1e-12
    assert abs(lrs[-1] - 3e-4 / 25 / 1e4) < 1e-12  # cosine anneal floor (scheduler.py:274)
    assert max(lrs) <= 3e-4 + 1e-15
