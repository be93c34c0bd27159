"""Fixtures for tcdiff_amd/io.py from the REAL reference classes (this container only):
  normalizer.npz     : dataset.preprocess.Normalizer fitted on seeded data; normalize / unnormalize outputs
  ref_ckpt_small.pt  : a checkpoint dictionary in the layout of TCDiff.py:266-273 holding a pickled REFERENCE Normalizer
                       instance and `module.`-prefixed weights of two small tensors (data, no source)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from dataset.preprocess import Normalizer  # noqa: E402

g = torch.Generator().manual_seed(123)
data = torch.randn(7, 30, 151, generator=g) * 3 + 0.5
data[..., 9] = 2.0                                     # a constant channel: zero range -> scale 1
norm = Normalizer(data.clone())
x = torch.randn(2, 30, 151, generator=g) * 3
xn = norm.normalize(x.clone())
y = torch.randn(2, 30, 151, generator=g) * 0.8
yu = norm.unnormalize(y.clone())
y147 = torch.randn(2, 30, 147, generator=g) * 0.8      # narrower input: the LAST 147 fitted columns are used
yu147 = norm.unnormalize(y147.clone())
np.savez_compressed(os.path.join(HERE, "normalizer.npz"), data=data.numpy(), x=x.numpy(), xn=xn.numpy(), y=y.numpy(),
                    yu=yu.numpy(), y147=y147.numpy(), yu147=yu147.numpy())
sd = {"module.final_layer.bias": torch.arange(151.0), "module.null_cond_hidden": torch.ones(1, 512) * 0.5}
torch.save({"ema_state_dict": sd, "model_state_dict": sd, "optimizer_state_dict": {"state": {}, "param_groups": []},
            "normalizer": norm}, os.path.join(HERE, "ref_ckpt_small.pt"))
print("ok")
