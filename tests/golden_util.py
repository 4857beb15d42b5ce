"""Helpers shared by the parity tests: load the committed golden fixtures (generated from the
reference by oracle/gen_goldens.py) and rebuild the seeded weights / inputs they were made from."""
import ast
import os

import numpy as np
import torch

from oracle import msau_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

NET_CASES = ["net_f8_c13_33x26", "net_f4_c13_b2_64x48", "net_1stage_c32_b2_40x40",
             "net_2stage_c24_dense_24x40", "net_r3_s3_c8_21x35", "net_cfg2_336x256x64", "net_defaults_s6_r3_c8_70x96",
             "net_elu_f8_c13_33x26"]


def load_ops():
    return np.load(os.path.join(GOLDEN, "ops.npz"), allow_pickle=True)


def load_net_case(name):
    """-> (golden npz, cfg, state_dict restricted to cfg's num_blocks, x, label)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=True)
    cfg = ast.literal_eval(str(g["cfg"]))
    seed = int(g["seed"])
    sd3 = O.init_params(dict(cfg, num_blocks=3), seed)
    # the fixture's seeds must regenerate exactly what the reference was fed
    cs = float(sum(float(v.double().abs().sum()) for v in sd3.values()))
    assert abs(cs - float(g["weights_checksum"])) <= 1e-9 * abs(cs), "seeded weights differ from the fixture's"
    keep = set(O.param_shapes(cfg).keys())
    sd = {k: v for k, v in sd3.items() if k in keep}
    x, label = O.synthetic_batch(int(g["B"]), cfg["channels"], int(g["H"]), int(g["W"]), cfg["n_class"],
                                 seed + 1, dense=bool(g["dense"]))
    assert abs(float(x.double().abs().sum()) - float(g["input_checksum"])) < 1e-6 * max(1.0, float(g["input_checksum"]))
    assert float(label.sum()) == float(g["label_checksum"])
    return g, cfg, sd, x, label


def summarize(t, n=64):
    f = t.detach().reshape(-1).double().cpu()
    stride = max(1, f.numel() // n)
    return np.array([float(f.norm()), float(f.sum())]), f[::stride][:n].float().numpy()


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def err(a, b, bf16=False):
    """fp32 storage: max-norm error relative to the tensor's max.  bf16 storage: relative L2 error --
    a ReLU whose pre-activation is within bf16 rounding of 0 legitimately flips, which moves single
    elements of a gradient by O(1) while leaving the tensor as a whole within rounding."""
    return rel_l2(a, b) if bf16 else rel_err(a, b)
