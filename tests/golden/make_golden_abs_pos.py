"""Golden vectors of the REAL reference built with use_rotary=False (model/model.py:441-448: no rotary embedding anywhere,
PositionalEncoding added to the motion tokens :564 and the music tokens :580; TCDiff.py:76-87 uses the rotary default): this
container only; needs /root/reference.

    python tests/golden/make_golden_abs_pos.py

  c1_abs_pos.npz : config-1 shape (1 clip, 2 dancers x 60 frames, T = 100): the guided evaluation at t = 50 (w = 2), the
     conditional forward at t = 3, and the full 100-step p_sample_loop with injected noise (x after 1 / 50 / 90 / 99 steps and the
     final sample); the state_dict keys of that model.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import refload  # noqa: E402
from oracle import tcdiff_oracle as O  # noqa: E402

torch.set_num_threads(8)


def main():
    dn, S, T = 2, 60, 100
    L = dn * S
    sd = O.synth_state_dict(dn=dn, seq_len=S, use_rotary=False)
    cond = torch.stack([O.synth_cond(0, S)])
    xT = torch.stack([O.synth_xT(0, L)])
    model, diff = refload.build_reference(sd, dn=dn, seq_len=S, n_timestep=T, use_rotary=False)
    assert float((model.abs_pos_encoding.pe - sd["abs_pos_encoding.pe"]).abs().max()) == 0.0     # the synthetic buffer IS the module's
    out = {"state_dict_keys": np.array(sorted(model.state_dict().keys()))}
    with torch.no_grad():
        out["guided_w2_t50"] = model.guided_forward(xT, cond, torch.tensor([50]), 2).numpy()
        out["fwd_cond_t3"] = model(xT, cond, torch.tensor([3]), cond_drop_prob=0.0).numpy()
        eps_fn = O.batch_step_noise([0], L)
        counter = {"i": T}

        def like(tensor):
            counter["i"] -= 1
            return eps_fn(counter["i"], tensor.shape)

        with refload.patched_randn(like_fn=like):
            x, chain = diff.p_sample_loop((1, L, 151), cond, noise=xT.clone(), return_diffusion=True)
        assert counter["i"] == 0
        out.update(final=x.numpy(), after_step_99=chain[1].numpy(), after_step_50=chain[50].numpy(),
                   after_step_10=chain[90].numpy(), after_step_1=chain[99].numpy())
    # the oracle's restatement of the same option
    with torch.no_grad():
        og = O.guided_forward(sd, xT, cond, torch.tensor([50]), 2)
        oc = O.decoder_forward(sd, xT, cond, torch.tensor([3]), cond_drop_prob=0.0)
    print("oracle vs reference: guided", float((og - torch.from_numpy(out["guided_w2_t50"])).abs().max()),
          "forward", float((oc - torch.from_numpy(out["fwd_cond_t3"])).abs().max()))
    np.savez_compressed(os.path.join(HERE, "c1_abs_pos.npz"), **out)


if __name__ == "__main__":
    main()
