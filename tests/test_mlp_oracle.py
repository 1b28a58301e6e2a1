"""CPU: the residual-MLP restatement (config 5's oracle)."""
import os

import numpy as np
import pytest

from oracle import mppi_oracle as mo

CKPT = "/root/reference/saved_models/mlp_diff_300x100_3l.pth"


@pytest.mark.skipif(not os.path.exists(CKPT), reason="reference checkpoint only exists in the build container")
def test_mlp_forward_matches_torch_on_the_reference_checkpoint():
    import torch
    sd = torch.load(CKPT, map_location="cpu", weights_only=True)  # loader that executes nothing from the file
    w = {k: v.numpy() for k, v in sd.items()}
    assert w["input_layer.weight"].shape == (512, 5) and w["out_layer.weight"].shape == (3, 512)
    z = np.random.default_rng(0).normal(size=(33, 5))
    x = torch.tensor(z)
    lin = lambda t, W, b: t @ torch.tensor(W, dtype=torch.float64).T + torch.tensor(b, dtype=torch.float64)
    h = lin(x, w["input_layer.weight"], w["input_layer.bias"])  # no activation (train/train_diff_mlp.py:32)
    for i in range(3):
        h = torch.tanh(lin(h, w[f"hidden_layer.{i}.weight"], w[f"hidden_layer.{i}.bias"]))
    y = lin(h, w["out_layer.weight"], w["out_layer.bias"])
    np.testing.assert_allclose(mo.mlp_forward(w, z), y.numpy(), rtol=1e-12, atol=1e-13)


def test_zero_residual_reduces_to_the_reference_dynamics():
    """With out_layer = 0 (the state train_diff_mlp.py:27-29 initialises) the MLP oracle IS the diff-drive oracle."""
    import golden_util as gu
    fx = gu.load("dd_c1_moderate")
    w = mo.random_mlp_weights(3)
    w["out_layer.weight"][:] = 0
    w["out_layer.bias"][:] = 0
    o = mo.DiffDriveMlpOracle(**fx["meta"], mlp_weights=w)
    out = o.iteration(fx["x0"], fx["eps"].astype(np.float64))
    np.testing.assert_allclose(out["S"], fx["S"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out["u_returned"], fx["u_returned"], rtol=1e-9, atol=1e-12)
