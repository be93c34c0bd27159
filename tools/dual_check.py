"""Two-stream vs single-stream sampling: bitwise comparison and timing (GPU box)."""
import sys, time, torch
sys.path.insert(0, ".")
from tcdiff_amd import DanceDecoder, GaussianDiffusion
from tcdiff_amd import weights as W

def main():
    dn, frames, B, T = 3, 150, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 1000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    L = dn * frames
    model = DanceDecoder(nfeats=151, seq_len=frames, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8,
                         cond_feature_dim=438, required_dancer_num=dn, compute_dtype="bf16")
    model.load_state_dict(W.synth_state_dict_like(model))
    model = model.cuda().eval()
    diff = GaussianDiffusion(model, frames, 151, None, schedule="cosine", n_timestep=T, predict_epsilon=False,
                             guidance_weight=2, seq_len=frames).cuda().eval()
    cond = torch.stack([W.synth_cond(c, frames) for c in range(B)]).cuda()
    xT = torch.stack([W.synth_xT(c, L) for c in range(B)]).cuda()
    res = {}
    skews = [float(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else []
    for dual in [False, False] + skews:
        diff.dual_stream = dual is not False
        diff.dual_skew_us = -1.0 if isinstance(dual, bool) else abs(dual) % 1000
        diff.dual_parts = 2 if isinstance(dual, bool) else int(abs(dual) // 1000) or 2
        tseq = list(range(T - 1, T - 1 - steps, -1))
        torch.cuda.synchronize(); t0 = time.time()
        x = diff._run(0, (B, L, 151), cond, xT.clone().float(), tseq, diff._ddpm_params(tseq), seed=1234)
        t_enq = time.time() - t0
        torch.cuda.synchronize(); dt = time.time() - t0
        print(f"dual={dual}: {dt / steps * 1e3:.3f} ms/step (host enqueue {t_enq / steps * 1e3:.3f} ms/step)", flush=True)
        res.setdefault(dual, x)
    # CPU cost of one graph launch on an idle queue
    graphs = [g for k, g in diff.__dict__.get("_graphs", {}).items() if not (isinstance(k, tuple) and k and k[0] == "warm")]
    for g in graphs[:3]:
        gl = g if isinstance(g, list) else [g]
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.time(); gl[0].replay(); ts.append(time.time() - t0); torch.cuda.synchronize()
        print("graph.replay() host time (ms):", [round(t * 1e3, 3) for t in ts], flush=True)
    for k, v in res.items():
        print(k, "bitwise equal to single-stream:", torch.equal(res[False], v))

main()
