"""SECOND's convolutions on the two matrix pipes, one MI355X: accuracy of both kernels against a float64 product and their times.

  python tools/bf16x3_check.py [sweeps]        # -> stdout (committed as profiles/r05_bf16x3_conv.log)

Every layer shape of pointpillars_kitti.yml's backbone (point_pillars.py:619-682) at `sweeps` frames (default 16 = one lane of
bench.py's step): ml3d_conv2d_nhwc (f32 MFMA) and ml3d_conv2d_nhwc_bf16x3 (three-way bf16 split, six bf16 MFMAs per product block) on
the same seeded input; max |out - float64| of each, max |bf16x3 - f32|, and the average of 20 launches timed with HIP events.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "open3d-ml_amd"))
from ml3d import ops      # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    shapes = [(64, 64, 2, 496, 432), (64, 64, 1, 248, 216), (64, 128, 2, 248, 216), (128, 128, 1, 124, 108),
              (128, 256, 2, 124, 108), (256, 256, 1, 62, 54)]
    tot = [0.0, 0.0]
    counts = [1, 3, 1, 5, 1, 5]
    print("sweeps %d | cin cout stride HxW | err_f32 err_bf16x3 |bf16x3-f32| out_scale | ms_f32 ms_bf16x3 speedup | TF_f32 TF_bf16x3(f32-equivalent)" % B)
    for (cin, cout, stride, H, W), cnt in zip(shapes, counts):
        x = torch.randn((B, H, W, cin), generator=g).relu_().to(dev)            # post-ReLU activations, as in the network
        w = (torch.randn((9 * cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
        b = (torch.randn((cout,), generator=g) * 0.1).to(dev)
        pk = ops.pack_bf16x3(w)
        o32 = ops.conv2d_nhwc(x, w, b, 3, 3, stride, 1, act=2)
        obf = ops.conv2d_nhwc(x, w, b, 3, 3, stride, 1, act=2, packed=pk)
        # float64 reference on a slice of the batch (torch's conv in double)
        nb = min(B, 2)
        w4 = w.double().view(3, 3, cin, cout).permute(3, 2, 0, 1).contiguous()
        ref = torch.relu(torch.nn.functional.conv2d(x[:nb].double().permute(0, 3, 1, 2), w4, b.double(), stride=stride, padding=1)).permute(0, 2, 3, 1)
        e32 = float((o32[:nb].double() - ref).abs().max())
        ebf = float((obf[:nb].double() - ref).abs().max())
        dd = float((obf - o32).abs().max())
        sc = float(ref.abs().max())
        ms = []
        for packed in (None, pk):
            for _ in range(3):
                ops.conv2d_nhwc(x, w, b, 3, 3, stride, 1, act=2, packed=packed, out=o32)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv2d_nhwc(x, w, b, 3, 3, stride, 1, act=2, packed=packed, out=o32)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 20)
        OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        fl = 2.0 * B * OH * OW * cout * 9 * cin
        tot[0] += ms[0] * cnt
        tot[1] += ms[1] * cnt
        print("%3d %3d s%d %dx%d | %.3g %.3g %.3g %.3g | %.4f %.4f %.2fx | %.1f %.1f" %
              (cin, cout, stride, H, W, e32, ebf, dd, sc, ms[0], ms[1], ms[0] / ms[1], fl / ms[0] * 1e-9, fl / ms[1] * 1e-9))
    print("backbone convolutions of one forward (16 layers): f32 %.3f ms, bf16x3 %.3f ms" % (tot[0], tot[1]))


if __name__ == "__main__":
    main()
