"""Host-side weight preparation for the HIP RandLA-Net forward.

Folds every eval-mode BatchNorm (eps 1e-6, reference ml3d/torch/models/randlanet.py:77,499)
into the preceding 1x1 conv / Linear and lays the result out as [C_in][C_out] row-major
slabs in the slot order documented in include/ml3d_hip.h.  The fold is done in float64 and
rounded once to float32.  Input is a state_dict with the REFERENCE's parameter names
(SURVEY.md Appendix C), so published checkpoints load unchanged.
"""
import numpy as np

BN_EPS = 1e-6


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64) if hasattr(t, "detach") else np.asarray(t, np.float64)


def _fold(w_oi, b_o, sd, bn_prefix):
    """w_oi [C_out, C_in], b_o [C_out] + BatchNorm(bn_prefix) -> (WT [C_in, C_out], b [C_out])."""
    if bn_prefix is not None:
        g, beta = _np(sd[bn_prefix + ".weight"]), _np(sd[bn_prefix + ".bias"])
        mu, var = _np(sd[bn_prefix + ".running_mean"]), _np(sd[bn_prefix + ".running_var"])
        s = g / np.sqrt(var + BN_EPS)
        w_oi = w_oi * s[:, None]
        b_o = (b_o - mu) * s + beta
    return np.ascontiguousarray(w_oi.T), b_o


def _conv(sd, name, bn=True, transpose=False):
    w = _np(sd[name + ".conv.weight"])[:, :, 0, 0]
    if transpose:            # ConvTranspose2d stores [C_in, C_out, 1, 1]  (randlanet.py:486-491)
        w = w.T
    return _fold(w, _np(sd[name + ".conv.bias"]), sd, name + ".batch_norm" if bn else None)


def slot_tensors(sd, cfg):
    """List of float64 arrays in ABI slot order."""
    out = []
    out += list(_fold(_np(sd["fc0.weight"]), _np(sd["fc0.bias"]), sd, "bn0"))
    for l in range(cfg["num_layers"]):
        p = "encoder.%d." % l
        out += list(_conv(sd, p + "mlp1"))
        out += list(_conv(sd, p + "lse1.mlp"))
        out += list(_fold(_np(sd[p + "pool1.score_fn.0.weight"]), _np(sd[p + "pool1.score_fn.0.bias"]), sd, None))
        out += list(_conv(sd, p + "pool1.mlp"))
        out += list(_conv(sd, p + "lse2.mlp"))
        out += list(_fold(_np(sd[p + "pool2.score_fn.0.weight"]), _np(sd[p + "pool2.score_fn.0.bias"]), sd, None))
        out += list(_conv(sd, p + "pool2.mlp"))
        m2w, m2b = _conv(sd, p + "mlp2")
        scw, scb = _conv(sd, p + "shortcut")
        out += [m2w, scw, m2b, scb]        # weights adjacent: one stacked [d + d_in][2d] matrix
    out += list(_conv(sd, "mlp"))
    for i in range(cfg["num_layers"]):
        out += list(_conv(sd, "decoder.%d" % i, transpose=True))
    out += list(_conv(sd, "fc1.0"))
    out += list(_conv(sd, "fc1.1"))
    out += list(_conv(sd, "fc1.3", bn=False))
    return out


def pack(sd, cfg, offsets):
    """-> flat float32 buffer of offsets[-1] floats."""
    tensors = slot_tensors(sd, cfg)
    if len(tensors) != len(offsets) - 1:
        raise RuntimeError("RandLA-Net parameter layout mismatch: %d tensors vs %d slots"
                           % (len(tensors), len(offsets) - 1))
    buf = np.zeros(int(offsets[-1]), np.float32)
    for t, o, nxt in zip(tensors, offsets[:-1], offsets[1:]):
        flat = t.reshape(-1)
        if flat.size > nxt - o:
            raise RuntimeError("slot overflow while packing RandLA-Net parameters")
        buf[o:o + flat.size] = flat.astype(np.float32)
    return buf
