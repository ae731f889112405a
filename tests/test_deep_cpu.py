"""CPU: the VQVAE_Deep oracle (oracle/vqvae_deep_oracle.py) against the outputs captured from the reference's own
vqvae_deep.py (tests/golden/deep.npz), and the product module's state_dict layout against the reference's."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import vqvae_deep_oracle as OD
from oracle import vqvae_oracle as O
from oracle.make_golden_cases import DEEP_EMBED_SCALE, DEEP_GAIN, DEEP_SEED


def close(a, b, rtol=1e-5, atol=1e-6, what=""):
    np.testing.assert_allclose(a.detach().numpy() if isinstance(a, torch.Tensor) else a, b, rtol=rtol, atol=atol, err_msg=what)


def test_deep_oracle_matches_reference_golden(golden):
    g = golden("deep")
    cfg = OD.DEEP_TINY
    st = OD.make_deep_state(cfg, DEEP_SEED, DEEP_EMBED_SCALE, DEEP_GAIN)
    leaves = {k: (v.clone().requires_grad_(True) if not O.is_buffer(k) else v.clone()) for k, v in st.items()}
    img = O.make_images(2, 32, DEEP_SEED)
    style = OD.make_style(2, cfg, DEEP_SEED).requires_grad_(True)
    dec, diff, quant, id_t, id_b = OD.deep_forward(leaves, cfg, img, style, training=True)
    assert np.array_equal(id_t.numpy().astype(np.int32), g["tiny.id_t"]) and np.array_equal(id_b.numpy().astype(np.int32), g["tiny.id_b"])
    close(dec, g["tiny.dec"], what="dec")
    close(diff, g["tiny.diff"])
    close(quant, g["tiny.quant"])
    loss = F.mse_loss(dec, img) + 0.25 * diff.mean()
    close(loss, g["tiny.loss"])
    loss.backward()
    close(style.grad, g["tiny.g.style"], rtol=1e-4, atol=1e-7)
    n = 0
    for k, v in leaves.items():
        if O.is_buffer(k):
            continue
        if OD.is_dead_key(k):
            assert v.grad is None and f"tiny.g.{k}" not in g.files, k
        else:
            close(v.grad, g[f"tiny.g.{k}"], rtol=1e-4, atol=1e-6, what=k)
            n += 1
    assert n == sum(1 for f in g.files if f.startswith("tiny.g.") and f != "tiny.g.style")
    for k in ("quantize_t.cluster_size", "quantize_b.cluster_size", "quantize_t.embed_avg", "quantize_b.embed"):
        close(leaves[k], g[f"tiny.after.{k}"], rtol=1e-5, atol=1e-6, what=k)


def test_deep_state_dict_layout_matches_reference(golden):
    """Keys, order, shapes and parameter count of VQVAE_Deep() as captured from the reference (26.6 M parameters,
    the dead AdainResBlk.conv included)."""
    import vqvae2_amd
    g = golden("deep")
    m = vqvae2_amd.VQVAE_Deep()
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["default.keys"]]
    assert [list(v.shape) + [0] * (4 - v.dim()) for v in sd.values()] == g["default.shapes"].tolist()
    assert sum(p.numel() for p in m.parameters()) == int(g["default.n_params"]) == 26639254
    assert list(OD.deep_state_spec(OD.DEEP_DEFAULT).keys()) == list(sd.keys())
    assert m.embed_dim == 512
    t = vqvae2_amd.VQVAE_Deep(channel=32, n_res_block=1, n_res_channel=16, embed_dim=16, n_embed=64, style_dim=24)
    t.load_state_dict(OD.make_deep_state(OD.DEEP_TINY, DEEP_SEED))       # a reference-layout checkpoint round-trips
    assert "decode() missing 1 required positional argument: 'style'" in str(g["forward_raises"])
    try:
        t(torch.zeros(1, 3, 32, 32))
    except TypeError as e:       # the fork's forward() cannot run (vqvae_deep.py:277): same failure here
        assert "missing 1 required positional argument: 'style'" in str(e)
    else:
        raise AssertionError("forward() without a style must raise like the reference")
