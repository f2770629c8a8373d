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


def test_default_batcher_collates_like_the_reference_and_a_batch_of_one_is_a_view():
    """``DefaultBatcher.collate_fn`` (default_batcher.py:34-85) on what ``RandLANet.transform`` returns: per-level lists of
    tensors / arrays stacked along a new leading axis, numbers to tensors, strings passed through.  A batch of ONE tensor comes
    back as a view (no copy kernel per entry), with the values and shape torch.stack would give."""
    from ml3d.torch.dataloaders import DefaultBatcher
    rng = np.random.default_rng(0)

    def item(seed):
        r = np.random.default_rng(seed)
        return {"data": {"coords": [torch.from_numpy(r.random((n, 3), dtype=np.float32)) for n in (16, 4)],
                         "neighbor_indices": [torch.from_numpy(r.integers(0, n, (n, 5)).astype(np.int32)) for n in (16, 4)],
                         "features": r.random((16, 3), dtype=np.float32), "labels": r.integers(0, 5, 16), "point_inds": np.arange(16)},
                "attr": {"split": "test", "idx": seed}}
    collate = DefaultBatcher().collate_fn
    for items in ([item(1)], [item(1), item(2), item(3)]):
        out = collate(items)
        B = len(items)
        assert out["attr"]["split"] == ["test"] * B and out["attr"]["idx"].tolist() == [it["attr"]["idx"] for it in items]
        for key in ("coords", "neighbor_indices"):
            for l in range(2):
                want = torch.stack([it["data"][key][l] for it in items], 0)
                got = out["data"][key][l]
                assert got.shape == want.shape and got.dtype == want.dtype and torch.equal(got, want) and got.is_contiguous()
        assert torch.equal(out["data"]["features"], torch.stack([torch.as_tensor(it["data"]["features"]) for it in items], 0))
        assert out["data"]["labels"].shape == (B, 16)
    one = item(7)
    out = collate([one])
    assert out["data"]["coords"][0].data_ptr() == one["data"]["coords"][0].data_ptr()          # the view, not a copy
