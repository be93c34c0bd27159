"""Generate golden vectors from the REAL reference (build container only; needs /root/reference).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Inputs are regenerated on any box from the name/seed-keyed recipes in oracle/tcdiff_oracle.py
(synth_state_dict / synth_cond / synth_xT / synth_step_eps / synth_traj), so only OUTPUTS of the
reference are stored.  Every output below is produced by the reference's own classes
(model/model.py DanceDecoder, model/diffusion.py GaussianDiffusion) imported from /root/reference.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refload  # noqa: E402
from oracle import tcdiff_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def save(name, **arrs):
    np.savez(os.path.join(OUT, name + ".npz"), **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", name, {k: np.asarray(v).shape for k, v in arrs.items()})


def cfg_inputs(dn, S, clip_ids):
    L = dn * S
    cond = torch.stack([O.synth_cond(c, S) for c in clip_ids])
    xT = torch.stack([O.synth_xT(c, L) for c in clip_ids])
    return L, cond, xT


def main():
    t0 = time.time()
    # ---- schedule tables ---------------------------------------------------------------------
    for T in (100, 1000):
        sd = O.synth_state_dict(dn=2, seq_len=60)
        _, diff = refload.build_reference(sd, dn=2, seq_len=60, n_timestep=T)
        names = list(O.make_tables(T).keys())
        save(f"tables_T{T}", **{n: getattr(diff, n).numpy() for n in names})

    # ---- key set / shapes of the reference state_dict ------------------------------------------
    for dn, S in ((2, 60), (3, 150)):
        sd = O.synth_state_dict(dn=dn, seq_len=S)
        model, _ = refload.build_reference(sd, dn=dn, seq_len=S, n_timestep=100)
        ref_sd = model.state_dict()
        assert list(sorted(ref_sd)) == list(sorted(sd)), "key mismatch"
        for k in ref_sd:
            assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    save("state_dict_keys_dn3", keys=np.array(sorted(ref_sd.keys())),
         shapes=np.array([str(tuple(ref_sd[k].shape)) for k in sorted(ref_sd)]),
         param_order=np.array([n for n, _ in model.named_parameters()]))

    # ---- C1: 2 dancers x 60 frames, T=100 ------------------------------------------------------
    dn, S, T = 2, 60, 100
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    model, diff = refload.build_reference(sd, dn=dn, seq_len=S, n_timestep=T)
    L, cond, xT = cfg_inputs(dn, S, [0])
    with torch.no_grad():
        outs = {}
        for t in (99, 3):
            tt = torch.full((1,), t, dtype=torch.long)
            outs[f"fwd_cond_t{t}"] = model(xT, cond, tt, cond_drop_prob=0.0).numpy()
            outs[f"fwd_unc_t{t}"] = model(xT, cond, tt, cond_drop_prob=1.0).numpy()
            outs[f"guided_w2_t{t}"] = model.guided_forward(xT, cond, tt, 2).numpy()
        save("c1_forward", **outs)

        # block-level goldens on the real modules (layer 0 / encoder 0)
        g = torch.Generator().manual_seed(77)
        xb = torch.randn(1, L, 512, generator=g)
        mem = torch.randn(1, S + 2, 512, generator=g)
        tb = torch.randn(1, 512, generator=g)
        xe = torch.randn(1, S, 512, generator=g)
        layer = model.seqTransDecoder.stack[0]
        traj_emb = torch.zeros(1, L - 1, 512)
        blk = {
            "rotary_x": model.rotary.rotate_queries_or_keys(xb).numpy(),
            "sinusoidal": model.time_mlp[0](torch.tensor([0, 1, 37, 99, 999])).numpy(),
            "decoder_layer0": layer(xb, mem, tb, traj_emb, model.embeddings_table.weight, None).numpy(),
            "encoder_layer0": model.cond_encoder[0](xe).numpy(),
            "sbi_self": layer.self_attn(xb, xb, xb, model.embeddings_table.weight).numpy(),
            "film1_scale": layer.film1(tb)[0].numpy(),
            "film1_shift": layer.film1(tb)[1].numpy(),
        }
        save("c1_blocks", **blk)

        # full p_sample_loop with injected noise
        eps_fn = O.batch_step_noise([0], L)
        counter = {"i": T}

        def like(tensor):
            counter["i"] -= 1
            return eps_fn(counter["i"], tensor.shape)

        with refload.patched_randn(like_fn=like):
            x, chain = diff.p_sample_loop((1, L, 151), cond, noise=xT.clone(), return_diffusion=True)
        assert counter["i"] == 0
        # chain[k] is x after k steps (chain[0] = x_T); step index i = T-k was just executed
        save("c1_p_sample_loop", final=x.numpy(), after_step_99=chain[1].numpy(), after_step_50=chain[50].numpy(),
             after_step_10=chain[90].numpy(), after_step_1=chain[99].numpy())
        print("C1 loop done", time.time() - t0)

    # ---- C2 shape: 3 dancers x 150 frames, T=1000 ----------------------------------------------
    dn, S, T = 3, 150, 1000
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    model, diff = refload.build_reference(sd, dn=dn, seq_len=S, n_timestep=T)
    L, cond, xT = cfg_inputs(dn, S, [0, 1])
    with torch.no_grad():
        outs = {}
        for t in (999, 37):
            tt = torch.full((1,), t, dtype=torch.long)
            outs[f"guided_w2_t{t}"] = model.guided_forward(xT[:1], cond[:1], tt, 2).numpy()
        tt = torch.tensor([500, 20])
        outs["fwd_cond_b2_t500_20"] = model(xT, cond, tt, cond_drop_prob=0.0).numpy()
        save("c2_forward", **outs)

        # three consecutive DDPM steps from x_T (steps 999, 998, 997) and three low-t steps (2,1,0) from x_T
        eps_fn = O.batch_step_noise([0], L)
        steps = {}
        for start in (1000, 3):
            counter = {"i": start}

            def like(tensor):
                counter["i"] -= 1
                return eps_fn(counter["i"], tensor.shape)

            x = xT[:1].clone()
            with refload.patched_randn(like_fn=like):
                for i in reversed(range(start - 3, start)):
                    x, _ = diff.p_sample(x, cond[:1], torch.full((1,), i, dtype=torch.long))
                    steps[f"after_step_{i}"] = x.numpy()
        save("c2_ddpm_steps", **steps)

        # ddim_sample with trajectory in-painting, B=1
        x0 = torch.stack([O.synth_traj(0, L)])
        calls = {"n": 0}
        times = [a for a, _ in O.ddim_time_pairs(T)]

        def like(tensor):
            t = times[calls["n"]]
            calls["n"] += 1
            return eps_fn(t, tensor.shape)

        with refload.patched_randn(like_fn=like, randn_fn=lambda *a, **k: xT[:1].clone()):
            xd = diff.ddim_sample((1, L, 151), cond[:1], x_0=x0.clone())
        save("c2_ddim", final=xd.numpy())
        print("C2 ddim done", time.time() - t0)

        # long_ddim_sample, B=2 (window coupling + weight ramp)
        eps2 = O.batch_step_noise([0, 1], L)
        x0l = torch.stack([O.synth_traj(c, L) for c in (0, 1)]).reshape(2, S, dn, 3)
        calls = {"n": 0}

        def like2(tensor):
            t = times[calls["n"]]
            calls["n"] += 1
            return eps2(t, tensor.shape)

        with refload.patched_randn(like_fn=like2, randn_fn=lambda *a, **k: xT.clone()):
            xl = diff.long_ddim_sample((2, L, 151), cond, x_0=x0l.clone())
        save("c2_long_ddim", final=xl.numpy())
    print("all goldens written in %.1f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
