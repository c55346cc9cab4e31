"""Host-side mirror of the reference interface: names, constructor arguments, error behaviour, state_dict keys
(checked against the key list the reference module itself produced, stored in the golden fixture)."""
import os

import numpy as np
import pytest
import torch

from cases import ENCODER_CASES


def _build(d=64, L=2, h=4, final_norm="ln"):
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    pe = PositionalEmbeddingLayer("sin1d", 8192, d)
    return CrossTransformerModuleBox(no_patches=8192, pos_embedding_layer=pe, lang_pos_embedding=None, num_layers=L,
                                     patch_dropout=0.1, num_heads=h, fforward_multiplier=2, token_dropout=0.15,
                                     back_to_img_fn="regroup", activ_f="gelu", final_norm=final_norm, input_f_size=d)


def test_state_dict_keys_and_shapes_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "enc_small.npz"))
    enc = _build()
    assert sorted(enc.state_dict().keys()) == list(g["state_dict_keys"])
    sd = enc.state_dict()
    for k in g.files:
        if k.startswith("param/"):
            assert tuple(sd[k[6:]].shape) == g[k].shape, k
    assert sd["padding_mask"].dtype == torch.bool and sd["padding_mask"].shape == (1,)
    assert sd["pos_embedding_layer.pos_embedding"].shape == (1, 8192, 64)
    np.testing.assert_allclose(sd["pos_embedding_layer.pos_embedding"][:, :64].numpy(), g["pos_embedding_head"], atol=1e-6)


def test_layers_start_identical_like_nn_transformer_encoder():
    enc = _build(L=3)
    a, b = enc.t_encoder.layers[0].state_dict(), enc.t_encoder.layers[2].state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    # init families of torch's MultiheadAttention / Linear
    assert enc.t_encoder.layers[0].self_attn.in_proj_bias.abs().max() == 0
    assert enc.t_encoder.layers[0].self_attn.out_proj.bias.abs().max() == 0
    w = enc.t_encoder.layers[0].self_attn.in_proj_weight
    assert abs(w.abs().max().item() - (6 / (64 + 192)) ** 0.5) < 0.01


def test_constructor_errors_follow_reference():
    with pytest.raises(ValueError, match="not implemented"):
        _build(final_norm="bn")
    with pytest.raises(ValueError, match="not implemented"):
        _build(final_norm="other")
    enc = _build(final_norm=False)
    assert isinstance(enc.final_norm_layer, torch.nn.Identity)


def test_registry_and_pos_embedding_errors():
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_wrapper import get_cross_box_encoder
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    assert get_cross_box_encoder("cross_transformer", False) is CrossTransformerModuleBox
    with pytest.raises(ValueError, match="not implemented"):
        get_cross_box_encoder("nope", False)
    with pytest.raises(ValueError):
        PositionalEmbeddingLayer("bogus", 16, 8)


def test_local_mask_matches_reference(golden_dir):
    from transfusion_amd.modeling.cross_fusion.utils import get_visual_token_mask
    m = np.load(os.path.join(golden_dir, "local_mask_3x4_k1.npz"))["mask"]
    assert get_visual_token_mask((3, 4), "global") is None
    assert np.array_equal(get_visual_token_mask((3, 4), "local_1").numpy(), m)
    with pytest.raises(NotImplementedError):
        get_visual_token_mask((3, 4), "weird")


def test_radam_schedule_matches_reference_formula():
    from transfusion_amd.optim import radam_schedule
    n, s, mode = radam_schedule(1, 0.9, 0.999)
    assert mode == 0 and s == -1.0 and n < 5
    n, s, mode = radam_schedule(6, 0.9, 0.999)
    assert mode == 1 and n >= 5 and s > 0
    assert radam_schedule(1, 0.9, 0.999, degenerated_to_sgd=True)[2] == 2


def test_cpu_tensors_fail_loudly():
    from transfusion_amd import _lib
    enc = _build()
    with pytest.raises(_lib.TfError):
        enc(torch.randn(1, 4, 64), torch.randn(1, 3, 64), None)


def test_block_bits_layout_matches_the_reference_mask():
    """vis_tokens_mask [Nv,Nv] -> [S, ceil(S/64)] little-endian u64 words over the JOINT sequence: only visual-visual pairs can be
    blocked (the reference pads the mask with zeros for language rows / columns, cross_f_box_layers.py:87-95), and the cache is
    keyed on tensor identity + version (an in-place edit or a new tensor must not hit it)."""
    import numpy as np
    from oracle import fusion_oracle as O
    enc = _build(d=32, L=1, h=2)
    m = O.local_visual_mask(6, 13, 1)                       # Nv = 78 > 64: the visual block spans two words per row
    Nv, Nl = m.shape[0], 70
    bits = enc._pack_block_bits(m, Nv, Nl, torch.device("cpu"))
    S, SW = Nv + Nl, (Nv + Nl + 63) // 64
    assert bits.shape == (S, SW) and bits.dtype == torch.int64
    words = bits.numpy().view(np.uint64)
    unpacked = ((words[:, :, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).reshape(S, SW * 64)[:, :S].astype(bool)
    expect = np.zeros((S, S), bool)
    expect[:Nv, :Nv] = m.numpy() != 0
    assert np.array_equal(unpacked, expect)
    assert enc._pack_block_bits(m, Nv, Nl, torch.device("cpu")) is bits            # same tensor object, same version: cached
    m[0, 5] = 1 - m[0, 5]                                                          # in-place edit bumps the version
    edited = enc._pack_block_bits(m, Nv, Nl, torch.device("cpu"))
    assert edited is not bits and not torch.equal(edited, bits)
    assert enc._pack_block_bits(m.clone(), Nv, Nl, torch.device("cpu")) is not edited     # a different tensor object never hits the cache
