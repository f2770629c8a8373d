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


def test_rank_placement_follows_the_gpus_numa_nodes(tmp_path):
    """ml3d.dist.plan_rank_cpus / gpu_numa_node on a fake sysfs: an 8-GPU, 2-socket node (GPUs 0-3 on node 0, 4-7 on node 1, 32
    CPUs per node of which the job may use all): every rank gets a contiguous quarter of ITS node's CPUs, no overlap, 8 host
    threads each (the slice is smaller than the 16-thread cap); unknown nodes fall back to an even split of the allowed set."""
    from ml3d import dist as mdist
    sysfs = tmp_path / "sys"
    ids = ["0000:%02x:00.0" % (0x10 + i) for i in range(8)]
    for i, p in enumerate(ids):
        d = sysfs / "bus" / "pci" / "devices" / p
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % (i // 4))
    for n, cl in ((0, "0-15,64-79"), (1, "16-31,80-95")):
        d = sysfs / "devices" / "system" / "node" / ("node%d" % n)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cl + "\n")
    nodes = [mdist.gpu_numa_node(p, str(sysfs)) for p in ids]
    assert nodes == [0, 0, 0, 0, 1, 1, 1, 1]
    assert mdist.gpu_numa_node("0000:ff:00.0", str(sysfs)) == -1
    allowed = set(range(96))
    plans = [mdist.plan_rank_cpus(r, nodes, allowed, str(sysfs)) for r in range(8)]
    node_cpus = {0: set(mdist._parse_cpulist("0-15,64-79")), 1: set(mdist._parse_cpulist("16-31,80-95"))}
    seen = set()
    for r, (cpus, node, threads) in enumerate(plans):
        assert node == r // 4 and len(cpus) == 8 and threads == 8 and set(cpus) <= node_cpus[node]
        assert not (seen & set(cpus))
        seen |= set(cpus)
    assert plans[0][0] == list(range(0, 8)) and plans[3][0] == list(range(72, 80))
    # the job is confined to 16 CPUs of node 0 only (a cgroup): ranks of node 1 have no CPU there -> even split of what is allowed
    few = set(range(16))
    cpus, node, threads = mdist.plan_rank_cpus(5, nodes, few, str(sysfs))
    assert node == -1 and cpus == [10, 11] and threads == 2
    # no NUMA information at all (containers, the CPU dry run): even split, 16-thread cap
    cpus, node, threads = mdist.plan_rank_cpus(1, [-1, -1], set(range(64)), str(sysfs))
    assert node == -1 and cpus == list(range(32, 64)) and threads == 16
