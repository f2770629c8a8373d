"""CPU: host-side logic of the product package — parameter mirror, BatchNorm folding/packing."""
import numpy as np
import torch

from ml3d import _abi
from ml3d.torch.models import _randla_pack
from ml3d.torch.models.randlanet import RandLANet
from oracle import randlanet_ref as R

CFG = dict(num_neighbors=16, num_layers=4, num_classes=19, sub_sampling_ratio=[4, 4, 4, 4],
           in_channels=3, dim_features=8, dim_output=[16, 64, 128, 256])


def test_model_state_dict_matches_reference_layout():
    m = RandLANet(**CFG, device="cpu")
    sd = m.state_dict()
    ref = R.param_shapes(CFG)
    assert list(sd.keys()) == list(ref.keys())
    for k, s in ref.items():
        assert tuple(sd[k].shape) == tuple(s), k
    m.load_state_dict(R.make_state_dict(CFG, 3))       # strict load of a reference-named state_dict


def test_five_layer_config_layout():
    cfg = dict(CFG, num_layers=5, sub_sampling_ratio=[4, 4, 4, 4, 2], dim_output=[16, 64, 128, 256, 512], in_channels=6)
    m = RandLANet(**cfg, device="cpu")
    assert list(m.state_dict().keys()) == list(R.param_shapes(cfg).keys())


def test_bn_fold_is_exact_affine():
    sd = R.make_state_dict(CFG, 9)
    wt, b = _randla_pack._conv(sd, "encoder.1.mlp1")
    x = np.random.default_rng(0).normal(size=(5, wt.shape[0]))
    w = sd["encoder.1.mlp1.conv.weight"].numpy()[:, :, 0, 0].astype(np.float64)
    y = x @ w.T + sd["encoder.1.mlp1.conv.bias"].numpy()
    p = "encoder.1.mlp1.batch_norm."
    y = (y - sd[p + "running_mean"].numpy()) / np.sqrt(sd[p + "running_var"].numpy().astype(np.float64) + 1e-6) \
        * sd[p + "weight"].numpy() + sd[p + "bias"].numpy()
    assert np.allclose(x @ wt + b, y, rtol=1e-12, atol=1e-12)
    # ConvTranspose2d weights are already [C_in, C_out]
    wt, _ = _randla_pack._conv(sd, "decoder.0", transpose=True)
    assert wt.shape == (768, 256)


def test_pack_matches_layout_slots():
    import __graft_entry__ as ge
    ge.build()
    lib = _abi.get()
    desc = _abi.make_desc(CFG, 1, 1024)
    off = _abi.randla_param_offsets(lib, desc)
    sd = R.make_state_dict(CFG, 9)
    buf = _randla_pack.pack(sd, CFG, off)
    assert buf.dtype == np.float32 and buf.size == off[-1]
    tensors = _randla_pack.slot_tensors(sd, CFG)
    for t, o in zip(tensors, off[:-1]):
        assert np.array_equal(buf[o:o + t.size], t.reshape(-1).astype(np.float32))
    assert (off % 4 == 0).all()      # 16-byte aligned slots
