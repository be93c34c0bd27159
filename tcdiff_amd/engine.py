"""Host-side sequencing of the TCDiff denoiser on MI355X.

The engine owns (a) the packed weights (T-typed, K padded, fused [Wq|Wk|Wv] / FiLM stacks), (b) the HBM
workspaces for one batch plan, (c) the launch sequence of one network evaluation.  Every arithmetic op is a
launcher of libtcdiff_gfx950.so (tcdiff_amd/kernels.py); torch supplies device memory and the stream only.

Data layout in HBM (T = bf16 or f32 operand type, always fp32 for the residual stream):
  x          fp32 [B*L, 151]           motion tensor, token index = frame*dn + dancer (model/diffusion.py:640,651)
  xa         fp32 [R, 512]             residual stream, R = n_branch*B*L rows; branch-major: [uncond clips | cond clips]
  h, rot     T    [R, 512]             LayerNorm output and its rotary image (GEMM A operands)
  Q, K, V    T    [n_seq][8][Lp][64]   head-major attention images, Lp = L rounded up to 128 (zero padded)
  Kc, Vc     T    [8 layers][2B][8][Lpc][64] cross-attention K/V caches: slot 0 = null conditioning,
                                       slot 1+i = clip i; rows 0..S-1 step-invariant, rows S,S+1 = time tokens
  film       fp32 [2B][24*1024]        (scale|shift) of the 24 DenseFiLM blocks for this step; on the chain path the rows are
                                       pre-folded with the LayerNorm weights / linear2 bias around them (load_weights)
  tables     t_base fp32 [n_t,512], kv_tab T [8][n_t][2][1024]  time path evaluated once per timestep set
Step-invariant work (music encoder, cross K/V of the 150 music rows, time tables, rotary table) is hoisted
out of the DDPM loop; results are identical to recomputing it every step.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch

from . import _lib as L
from . import kernels as K

MERGE12_MAX_L = 512      # measured: profiles/r06_small_batch_split.txt
NL_FILM = 3


class DenoiserEngine:
    def __init__(self, cfg: dict, device: torch.device, compute: str = "bf16"):
        L.load()  # fail loudly when the HIP library is missing
        if device.type != "cuda":
            raise L.TcdiffError("DenoiserEngine needs a HIP device (cuda:N); there is no CPU fallback")
        self.cfg = dict(cfg)
        self.dev = device
        self.dt = K.dtype_id(compute)
        self.T = K.TORCH_DT[self.dt]
        self.kt = K.k_tile(self.dt)
        c = self.cfg
        if c["latent"] != 512 or c["n_head"] * 64 != 512:
            raise L.TcdiffError(f"latent_dim={c['latent']}, num_heads={c['n_head']}: the gfx950 kernels are built for the width the "
                                "reference instantiates, latent_dim=512 as 8 heads x 64 (TCDiff.py:76-87); the constructor's own "
                                "defaults (256 / 4, model/model.py:417-431) and any other width are not built")
        self.D, self.H, self.NL = 512, c["n_head"], c["n_layers"]
        self.S, self.dn, self.nf, self.ff = c["seq_len"], c["dn"], c["nfeats"], c["ff"]
        self.Lseq = self.S * self.dn
        self.Lp = K.round_up(self.Lseq, 128)
        self.Lps = K.round_up(self.S, 128)       # music encoder self-attention
        self.Lpc = K.round_up(self.S + 2, 128)   # cross-attention memory
        self.w: Dict[str, torch.Tensor] = {}
        self.weights_version = None
        self.plan_B = 0
        self.tables_key = None
        self.film_tab = None
        self._sampler_state = None
        self._side, self._forked = None, False     # forked stream of the step prologue's conditioning part (step_prologue)
        self._xcur = 0                             # small-job layers (chain_split): which of b["xa"], b["xb"] holds the residual stream
        # row-block chain kernels (csrc/chain.hip): bf16 only; the f32 parity mode keeps the op-by-op kernels
        self.act = int(cfg.get("act", L.ACT_GELU))    # feed-forward activation (TC_ACT_*); the chain kernels are GELU only
        # use_rotary=False (model/model.py:441-448,564,580): identity rotary table + PositionalEncoding rows added to the motion and
        # music tokens -- on the op-by-op kernels (the chain launches fuse the last fusion linear with layer 0's norm1)
        self.abs_pos = bool(cfg.get("abs_pos", False))
        self.use_chain = self.dt == L.DT_BF16 and os.environ.get("TCDIFF_CHAIN", "1") != "0" and self.ff == 1024 \
            and self.H == 8 and self.act == L.ACT_GELU and not self.abs_pos
        # TCDIFF_CHAIN=2 (default): cross-attention inside the chain too: per layer self-attention + ONE chain launch
        self.use_full = self.use_chain and os.environ.get("TCDIFF_CHAIN", "2") == "2"
        self.nkt = (self.S + 2 + 31) // 32          # 32-key tiles of the cross-attention memory
        # throughput mode: input_projection folded into the first fusion linear (one GEMM, K = dn * nfeats padded);
        # the f32 parity mode keeps the reference's two GEMMs (and its summation order)
        self.fold_in = self.dt == L.DT_BF16 and os.environ.get("TCDIFF_FOLD_IN", "1") != "0"
        self.kin = K.round_up(self.dn * self.nf, 64)
        # ... and final_layer folded into the last decoder layer's linear3 (the last chain launch writes the output)
        self.fold_out = self.use_chain and os.environ.get("TCDIFF_FOLD_OUT", "1") != "0"
        # ... and the last fusion linear + layer 0's norm1 / rotary / Q, K, V as one chain launch per (frame block, dancer)
        self.front = self.use_full and os.environ.get("TCDIFF_FRONT", "1") != "0" and self.S >= 8
        # the self-attention of layers 1.. inside the chain launch (row blocks cut per sequence, Q / K / V handed from launch to
        # launch in MFMA-fragment order; csrc/chain.hip): every job size (16- / 32- / 64-row blocks), bf16, 8-wave form
        self.fuse_sa = self.use_full and self.dt == L.DT_BF16 and os.environ.get("TCDIFF_FUSE_SA", "1") == "1"
        self.chain_nw = int(os.environ.get("TCDIFF_CHAIN_NW", "8"))      # waves per workgroup of the chain launches (8 or 4)
        if self.chain_nw not in (4, 8):
            raise L.TcdiffError("TCDIFF_CHAIN_NW must be 8 or 4")
        self.reset_graphs()

    def reset_graphs(self):
        """Captured step graphs hold raw device pointers: drop them whenever any buffer may have moved."""
        self.graphs = {}
        self.graph_warm = set()
        self.generation = getattr(self, "generation", 0) + 1

    def sampler_state(self, n_steps: int, rows: int, nfeat: int):
        """Persistent device-side sampler state (addresses are baked into captured graphs)."""
        st = self._sampler_state
        if st is None or st["x"].shape != (rows, nfeat) or st["cap"] < n_steps:
            cap = max(4096, n_steps)
            dev = self.dev
            st = dict(cap=cap,
                      x=torch.zeros(rows, nfeat, device=dev), eps=torch.zeros(rows, nfeat, device=dev),
                      cval=torch.zeros(rows, nfeat, device=dev), qeps=torch.zeros(rows, nfeat, device=dev),
                      cmask=torch.zeros(rows, nfeat, device=dev),
                      traj=torch.zeros(rows, 3, device=dev), counter=torch.zeros(8, device=dev, dtype=torch.int32),
                      rows=torch.zeros(cap, device=dev, dtype=torch.int32),
                      tseq=torch.zeros(cap, device=dev, dtype=torch.int32),
                      params=torch.zeros(cap, 8, device=dev), cparams=torch.zeros(cap, 8, device=dev))
            self._sampler_state = st
            self.reset_graphs()
        return st

    # ------------------------------------------------------------------------------------------
    # weights
    # ------------------------------------------------------------------------------------------
    def _pack(self, w: torch.Tensor, kpad: Optional[int] = None) -> torch.Tensor:
        n, k = w.shape
        kp = K.round_up(k, 64) if kpad is None else kpad
        out = torch.zeros(n, kp, device=self.dev, dtype=self.T)
        out[:, :k] = w.to(device=self.dev, dtype=self.T)
        if self.dt == L.DT_BF16X3:         # split-bf16 storage: [hi x4 | lo x4] per 16-byte chunk (csrc/common.h MmaBF16x3)
            return K.to_x3(out)
        return out.contiguous()

    # ---- weight streams of the chain kernels (include/tcdiff_hip.h, tcdiff_chain_args.wstream) -------------------
    # A stage (4 KB) is the image of the four 1-KB loads a wave makes for one 32-deep k-step: [n-tile][lane = 16 g + c][8 k]
    # with lane group g holding the 16-byte chunk PI[g] of the k-step (csrc/chain.hip: the activation fragments are read
    # from LDS in that chunk order because it is bank-conflict-free under the tile swizzle).
    _PI = (0, 3, 1, 2)

    @staticmethod
    def _stages_n512(W: torch.Tensor, nw: int = 8) -> torch.Tensor:
        """[512, K] -> [nw waves][K/32 stages][512 NT]: stage = [n-tile NT][lane group 4][16 weight rows][8 k] of one 32-deep
        k-step of the wave's 16 NT rows (NT = 32 / nw; wave w: rows 16 NT w ..; row 16 nt + c, k = 32 ks + 8 PI[g] + j)."""
        K_, NT = W.shape[1], 32 // nw
        w6 = W.reshape(nw, NT, 16, K_ // 32, 4, 8)[:, :, :, :, list(DenoiserEngine._PI), :]     # [w, nt, c, ks, g, j]
        return w6.permute(0, 3, 1, 4, 2, 5).reshape(nw, K_ // 32, 512 * NT)

    @staticmethod
    def _stages_ff1(W1: torch.Tensor, nw: int = 8) -> torch.Tensor:
        """[1024, 512] -> [4 chunks][nw waves][8 stages][512 NT]: wave w owns rows 256 c + 8 NT w .. of chunk c; stage =
        [k-step 2][n-tile NT / 2][lane group 4][16 rows][8 k]."""
        NH = 16 // nw
        w8 = W1.reshape(4, nw, NH, 16, 8, 2, 4, 8)[:, :, :, :, :, :, list(DenoiserEngine._PI), :]   # [ch, w, nt, c, st, k2, g, j]
        return w8.permute(0, 1, 4, 5, 2, 6, 3, 7).reshape(4, nw, 8, 1024 * NH)

    @staticmethod
    def _stages_ff2(W2: torch.Tensor, nw: int = 8) -> torch.Tensor:
        """[512, 1024] -> [4 chunks][nw waves][8 stages][512 NT]: the k-slice [256 c, 256 c + 256) of every row."""
        NT = 32 // nw
        w7 = W2.reshape(nw, NT, 16, 4, 8, 4, 8)[:, :, :, :, :, list(DenoiserEngine._PI), :]       # [w, nt, c, ch, ks, g, j]
        return w7.permute(3, 0, 4, 1, 5, 2, 6).reshape(4, nw, 8, 512 * NT)

    @staticmethod
    def _ffn_order(f1, f2):
        """the feed-forward stages in the order the kernel consumes them: linear1 chunk c, linear2 k-slice c, c = 0..3"""
        return [f for c in range(4) for f in (f1[c], f2[c])]

    def _build_chain_streams(self):
        w = self.w
        # waves per workgroup of the production launches (TC_CHAIN_FRONT / _FULL / _FULL_LAST): 8 x 64 columns or 4 x 128 columns
        # (tcdiff_chain_args.nw); chain A / B alone -- the reference points of tests -- keep the 8-wave form
        nw = self.chain_nw
        n512 = self._stages_n512
        if self.front:
            # TC_CHAIN_FRONT: per dancer, the 512 rows of the last fusion linear it owns (K = 1024: 32 stages), then layer
            # 0's w_qs / w_ks / w_vs (16 stages each)
            qkv = w["l0.qkv.w"]
            tail = [n512(qkv[0:512], nw), n512(qkv[512:1024], nw), n512(qkv[1024:1536], nw)]
            w["front"] = torch.stack([torch.cat([n512(w["f3.w"][512 * d:512 * d + 512], nw)] + tail, 1)
                                      for d in range(self.dn)]).contiguous()          # [dn][nw waves][80][512 NT]
        for l in range(self.NL):
            p = f"l{l}."
            if l + 1 < self.NL or not self.fold_out:
                W3 = w[p + "l3.w"]
            else:
                # the last layer's linear3 and final_layer are two linear maps with nothing in between
                # (model/model.py:344,623): W_f (W_3 h + b_3) + b_f.  The chain's last GEMM takes W_f W_3 (151 of its 512
                # output rows, the rest zero) and writes the network output itself; no separate final projection launch.
                W3f = self.sd_f32(f"seqTransDecoder.stack.{l}.linear3.weight")
                b3 = self.sd_f32(f"seqTransDecoder.stack.{l}.linear3.bias")
                Wf, bf = self.sd_f32("final_layer.weight"), self.sd_f32("final_layer.bias")
                Wo = torch.zeros(512, 512, device=self.dev, dtype=torch.float32)
                Wo[:self.nf] = Wf @ W3f
                bo = torch.zeros(512, device=self.dev, dtype=torch.float32)
                bo[:self.nf] = Wf @ b3 + bf
                w[p + "l3out.b"] = bo.contiguous()
                W3 = Wo.to(self.T).contiguous()

            def chain_a(n):
                return torch.cat([n512(w[p + "sfc.w"], n), n512(w[p + "cq.w"], n)], 1)

            def chain_b(n):
                f1, f2 = self._stages_ff1(w[p + "ff1.w"], n), self._stages_ff2(w[p + "ff2.w"], n)
                parts = [n512(w[p + "cfc.w"], n)] + self._ffn_order(f1, f2) + [n512(W3, n)]
                if l + 1 < self.NL:
                    qkv = w[f"l{l + 1}.qkv.w"]
                    parts += [n512(qkv[0:512], n), n512(qkv[512:1024], n), n512(qkv[1024:1536], n)]
                return torch.cat(parts, 1)
            w[p + "chainA"], w[p + "chainB"] = chain_a(8).contiguous(), chain_b(8).contiguous()
            if self.use_full:
                w[p + "chainF"] = torch.cat([chain_a(nw), chain_b(nw)], 1).contiguous()

    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.dev, dtype=torch.float32).contiguous()

    def load_weights(self, sd: Dict[str, torch.Tensor], version=None):
        """Repack a reference-keyed state_dict (model/model.py:440-540 names) for the kernels."""
        w, f, p = {}, self._f32, self._pack
        g = lambda k: sd[k].detach()
        self.sd_f32 = lambda k: sd[k].detach().to(self.dev, torch.float32)
        w["in.w"], w["in.b"] = p(g("input_projection.weight"), 192), f(g("input_projection.bias"))
        for i, j in ((0, "f1"), (2, "f2"), (4, "f3")):
            w[j + ".w"] = p(g(f"relative_projection_layer.{i}.weight"))
            w[j + ".b"] = f(g(f"relative_projection_layer.{i}.bias"))
        if self.fold_in:
            # input_projection and the first fusion linear are two linear maps with nothing in between
            # (model/model.py:560-561): W3 [x_0 W_in^T + b | x_1 .. | ..] + b3 = x_frame [W3_d W_in]_d^T + (sum_d W3_d b_in + b3)
            # with x_frame the dn * nfeats motion values of one frame, which are contiguous in x.  One GEMM with
            # K = dn * nfeats (453 -> 512) replaces K = 192 + K = 512 dn; products in fp32, rounded to the model dtype once.
            W3 = g("relative_projection_layer.0.weight").to(self.dev, torch.float32).reshape(1024, self.dn, 512)
            Win = g("input_projection.weight").to(self.dev, torch.float32)             # [512, nfeats]
            bin_ = g("input_projection.bias").to(self.dev, torch.float32)
            w["f1in.w"] = p(torch.einsum("odk,kn->odn", W3, Win).reshape(1024, self.dn * self.nf), self.kin)
            w["f1in.b"] = (torch.einsum("odk,k->o", W3, bin_) + w["f1.b"]).contiguous()
        w["t1.w"], w["t1.b"] = p(g("time_mlp.1.weight")), f(g("time_mlp.1.bias"))
        w["tc.w"], w["tc.b"] = p(g("to_time_cond.0.weight")), f(g("to_time_cond.0.bias"))
        w["tt.w"], w["tt.b"] = p(g("to_time_tokens.0.weight")), f(g("to_time_tokens.0.bias"))
        w["null_embed"] = f(g("null_cond_embed")).reshape(self.S, 512)
        w["null_hidden"] = f(g("null_cond_hidden")).reshape(1, 512)
        w["nc.g"], w["nc.b"] = f(g("norm_cond.weight")), f(g("norm_cond.bias"))
        w["c0.w"], w["c0.b"] = p(g("cond_projection.0.weight")), f(g("cond_projection.0.bias"))
        w["c2.w"], w["c2.b"] = p(g("cond_projection.2.weight")), f(g("cond_projection.2.bias"))
        for i in range(2):
            q = f"cond_encoder.{i}."
            w[f"e{i}.qkv.w"], w[f"e{i}.qkv.b"] = p(g(q + "self_attn.in_proj_weight")), f(g(q + "self_attn.in_proj_bias"))
            w[f"e{i}.o.w"], w[f"e{i}.o.b"] = p(g(q + "self_attn.out_proj.weight")), f(g(q + "self_attn.out_proj.bias"))
            w[f"e{i}.l1.w"], w[f"e{i}.l1.b"] = p(g(q + "linear1.weight")), f(g(q + "linear1.bias"))
            w[f"e{i}.l2.w"], w[f"e{i}.l2.b"] = p(g(q + "linear2.weight")), f(g(q + "linear2.bias"))
            for n in ("norm1", "norm2"):
                w[f"e{i}.{n}.g"], w[f"e{i}.{n}.b"] = f(g(q + n + ".weight")), f(g(q + n + ".bias"))
        w["na.g"], w["na.b"] = f(g("non_attn_cond_projection.0.weight")), f(g("non_attn_cond_projection.0.bias"))
        w["na1.w"], w["na1.b"] = p(g("non_attn_cond_projection.1.weight")), f(g("non_attn_cond_projection.1.bias"))
        w["na3.w"], w["na3.b"] = p(g("non_attn_cond_projection.3.weight")), f(g("non_attn_cond_projection.3.bias"))
        film_w, film_b = [], []
        for l in range(self.NL):
            q = f"seqTransDecoder.stack.{l}."
            sa, ca = q + "self_attn.", q + "multihead_attn."
            w[f"l{l}.qkv.w"] = p(torch.cat([g(sa + "w_qs.weight"), g(sa + "w_ks.weight"), g(sa + "w_vs.weight")], 0))
            w[f"l{l}.sfc.w"] = p(g(sa + "fc.weight"))
            w[f"l{l}.sln.g"], w[f"l{l}.sln.b"] = f(g(sa + "layer_norm.weight")), f(g(sa + "layer_norm.bias"))
            w[f"l{l}.cq.w"] = p(g(ca + "w_qs.weight"))
            w[f"l{l}.ckv.w"] = p(torch.cat([g(ca + "w_ks.weight"), g(ca + "w_vs.weight")], 0))
            w[f"l{l}.cfc.w"] = p(g(ca + "fc.weight"))
            w[f"l{l}.cln.g"], w[f"l{l}.cln.b"] = f(g(ca + "layer_norm.weight")), f(g(ca + "layer_norm.bias"))
            w[f"l{l}.ff1.w"], w[f"l{l}.ff1.b"] = p(g(q + "linear1.weight")), f(g(q + "linear1.bias"))
            w[f"l{l}.ff2.w"], w[f"l{l}.ff2.b"] = p(g(q + "linear2.weight")), f(g(q + "linear2.bias"))
            w[f"l{l}.l3.w"], w[f"l{l}.l3.b"] = p(g(q + "linear3.weight")), f(g(q + "linear3.bias"))
            for n in ("norm1", "norm2", "norm3", "norm4"):
                w[f"l{l}.{n}.g"], w[f"l{l}.{n}.b"] = f(g(q + n + ".weight")), f(g(q + n + ".bias"))
            for i in (1, 2, 3):
                Wf = g(q + f"film{i}.block.1.weight").to(self.dev, torch.float32)       # [1024, 512]: scale rows, shift rows
                bf_ = g(q + f"film{i}.block.1.bias").to(self.dev, torch.float32)
                if self.use_chain:
                    # The chain kernels take the FiLM rows PRE-FOLDED with what surrounds them (csrc/chain.hip fc epilogue):
                    # (scale + 1) * (LN(z) * g + b) + shift = LN(z) * G + Bv with G = g (scale + 1), Bv = b (scale + 1) + shift,
                    # and for the feed-forward block (scale + 1) * (z + b2) + shift = z * G + Bv with g = 1, b = b2
                    # (model/model.py:103-106,171-173,327,334,339).  scale and shift are LINEAR in the generator's input
                    # (DenseFiLM = Linear(Mish(t)), model/model.py:154-168), so G and Bv are too: the fold goes into the
                    # generator's weights and bias once per checkpoint, in fp32, and costs nothing per step.
                    gvec = {1: w[f"l{l}.sln.g"], 2: w[f"l{l}.cln.g"], 3: torch.ones(512, device=self.dev)}[i]
                    bvec = {1: w[f"l{l}.sln.b"], 2: w[f"l{l}.cln.b"], 3: w[f"l{l}.ff2.b"]}[i]
                    Ws, Wh, bs, bh = Wf[:512], Wf[512:], bf_[:512], bf_[512:]
                    Wf = torch.cat([gvec[:, None] * Ws, bvec[:, None] * Ws + Wh], 0)
                    bf_ = torch.cat([gvec * (bs + 1.0), bvec * (bs + 1.0) + bh], 0)
                film_w.append(Wf)
                film_b.append(bf_)
        w["film.w"] = p(torch.cat(film_w, 0))          # [NL*3*1024, 512]
        w["film.b"] = f(torch.cat(film_b, 0))
        w["fin.w"], w["fin.b"] = p(g("final_layer.weight")), f(g("final_layer.bias"))
        # constants of the architecture
        half = 256
        w["sin_freq"] = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(self.dev)  # model/utils.py:43-44
        n_pos = max(self.Lseq, self.S + 2)
        w["rope"] = torch.empty(n_pos, 512, device=self.dev, dtype=torch.float32)
        if self.abs_pos:      # no rotation: angle 0 everywhere (cos 1, sin 0); the positional rows are added in network / encode_music
            K.rope_table(torch.zeros(256, device=self.dev, dtype=torch.float32), w["rope"], n_pos)
            w["pe"] = f(g("abs_pos_encoding.pe")[:, 0, :])
            if max(self.Lseq, self.S) > w["pe"].shape[0]:
                raise L.TcdiffError(f"PositionalEncoding holds {w['pe'].shape[0]} positions, the sequence has {self.Lseq} tokens "
                                    "(model/utils.py:12,29)")
        else:
            K.rope_table(f(g("rotary.freqs")), w["rope"], n_pos)
        w["rope_cb"] = K.to_cb(w["rope"])                  # the chain kernels read it column-blocked [64][n_pos][8]
        self.w = w
        if self.use_chain:
            self._build_chain_streams()
        self.sd_f32 = None
        self.weights_version = version
        self.tables_key = None
        self.reset_graphs()

    # ------------------------------------------------------------------------------------------
    # workspaces
    # ------------------------------------------------------------------------------------------
    def plan(self, B: int):
        if B == self.plan_B:
            return
        dev, T = self.dev, self.T
        z = lambda *s, dtype=T: torch.zeros(*s, device=dev, dtype=dtype)
        Lq, S, H, NL = self.Lseq, self.S, self.H, self.NL
        R = 2 * B * Lq
        b = {}
        b["xin"] = z(B * S, self.kin) if self.fold_in else z(B * Lq, 192)    # [frames, dn * nfeats] or [tokens, nfeats]
        b["xp"] = z(B * Lq, 512)
        b["f1"], b["f2"] = z(B * S, 1024), z(B * S, 1024)
        b["xs"] = z(B * Lq, 512, dtype=torch.float32)
        b["h"], b["rot"], b["O"] = z(R, 512), z(R, 512), z(R, 512)
        b["Q"], b["K"] = z(2 * B, H, self.Lp, 64), z(2 * B, H, self.Lp, 64)
        b["V"] = z(2 * B, H, self.Lp, 64)
        b["xa"] = z(R, 512, dtype=torch.float32)
        b["h1"] = z(R, 1024)
        b["out"] = z(R, 152, dtype=torch.float32)
        b["film"] = z(2 * B, NL * NL_FILM * 1024, dtype=torch.float32)
        b["film_in"] = z(2 * B, 512)
        b["Kc"] = z(NL, 2 * B, H, self.Lpc, 64)      # slots: sampler uses 0..B, generic forward up to 2B
        b["Vc"] = z(NL, 2 * B, H, self.Lpc, 64)
        if self.use_full:                            # fragment-ordered images of the same caches (csrc/chain.hip)
            b["Kf"] = z(NL, 2 * B, H, self.nkt * 2048)
            b["Vf"] = z(NL, 2 * B, H, self.nkt * 2048)
        if self.fuse_sa:                             # fragment-order Q (per 64-row block of a sequence), K, V of the next layer
            self.skt = (Lq + 31) // 32
            self.n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
            b["Qf"] = z(2 * B * ((Lq + 15) // 16), 8, 4, 2, 64, 8)       # (block, wave)-private; enough for 16-row blocks
            # two of each: a launch reads layer l's keys in every block's prologue while its early blocks already write layer l + 1's.
            # INVARIANT the in-launch attention relies on: every V^T slot of a sequence's last 32-key tile is FINITE (keys >= L are
            # masked, but P = 0 times NaN is NaN).  The chain launch writes every slot it owns (clamped copies of the last row) and
            # zeros the one half-tile nobody owns (store_vfrag, MT = 1); the zero-fill here covers images no launch has written yet.
            b["sKf"], b["sVf"] = z(2, 2 * B, H, self.skt * 2048), z(2, 2 * B, H, self.skt * 2048)
            if self._split_rows(2 * B, Lq):          # small jobs: second residual buffer, two partial-sum slabs (chain_split.hip)
                nblk = 2 * B * ((Lq + 15) // 16)
                b["xb"] = z(R, 512, dtype=torch.float32)
                b["P0"], b["P1"] = z(nblk, 4, 16, 512, dtype=torch.float32), z(nblk, 4, 16, 512, dtype=torch.float32)
        b["hidden_all"] = z(2 * B, 512, dtype=torch.float32)
        b["tidx"] = torch.zeros(2 * B, device=dev, dtype=torch.int32)
        # music encoder (setup only)
        b["cin"] = z(B * S, K.round_up(2 * self.cfg["cond_dim"], 64))
        b["c1"] = z(B * S, K.round_up(self.cfg["cond_dim"], 64))
        b["tok"] = z(B * S, 512, dtype=torch.float32)
        b["mh"], b["mrot"], b["mO"] = z(B * S, 512), z(B * S, 512), z(B * S, 512)
        b["mQ"], b["mK"] = z(B, H, self.Lps, 64), z(B, H, self.Lps, 64)
        b["mV"] = z(B, H, self.Lps, 64)
        b["mh1"] = z(B * S, 1024)
        b["pool"] = z(B, 512, dtype=torch.float32)
        b["pool_h"], b["pool_h2"] = z(B, 512), z(B, 512)
        b["hidden"] = z(B, 512, dtype=torch.float32)
        if self.abs_pos:                              # the positional rows of every token of the batch (fp32, added with tcdiff_add_rows)
            b["pe_x"] = self.w["pe"][:Lq].repeat(B, 1).contiguous()
            b["pe_c"] = self.w["pe"][:S].repeat(B, 1).contiguous()
        self.b = b
        self.plan_B = B
        self.reset_graphs()

    # ------------------------------------------------------------------------------------------
    # step-invariant conditioning
    # ------------------------------------------------------------------------------------------
    def encode_music(self, cond: torch.Tensor):
        """cond (B, 2S or 2S+1, C) fp32 on device -> tokens fp32 [B*S,512] (b['tok']), hidden fp32 [B,512].
        model/model.py:572-583 (pairing, cond_projection, cond_encoder) and :593-597 (pooled projection)."""
        dt, w, b, S, H = self.dt, self.w, self.b, self.S, self.H
        B, clen, Cd = cond.shape
        if clen // 2 != S:
            raise L.TcdiffError(f"cond length {clen} does not pair into seq_len={S} tokens (model/model.py:572-589)")
        cond = cond.contiguous().float()
        M = B * S
        kc0, kc1 = b["cin"].shape[1], b["c1"].shape[1]
        K.convert_pad(dt, cond, b["cin"], M, 2 * Cd, kc0, rows_per_batch=S, batch_stride=clen * Cd, row_stride=2 * Cd)
        K.gemm_tile(dt, b["cin"], w["c0.w"], M, Cd, kc0, bias=w["c0.b"], act=L.ACT_RELU, out=b["c1"], ldc=kc1)
        K.gemm_tile(dt, b["c1"], w["c2.w"], M, 512, kc1, bias=w["c2.b"], mode=L.EPI_STORE_F32, out=b["tok"], ldc=512)
        if self.abs_pos:                              # cond_tokens = abs_pos_encoding(cond_tokens) (model/model.py:580)
            K.add_rows(b["tok"], 512, b["pe_c"], 512, b["tok"], 512, M, 512)
        for i in range(2):
            e = f"e{i}."
            K.ln_rot(dt, b["tok"], M, w[e + "norm1.g"], w[e + "norm1.b"], 1e-5, h=b["mh"], rot=b["mrot"],
                     rope=w["rope"], pos_mod=S)
            K.gemm_tile(dt, b["mrot"], w[e + "qkv.w"], M, 1536, 512, A2=b["mh"], split_n=1024, bias=w[e + "qkv.b"],
                        mode=L.EPI_QKV_HEADS, out=b["mQ"], out_k=b["mK"], out_v=b["mV"], scale_q=0.125, Lseq=S,
                        Lp=self.Lps, H=H, n_q=512, n_k=512)
            K.attention(dt, b["mQ"], b["mK"], b["mV"], b["mO"], B, H, S, S, self.Lps, self.Lps, 512)
            K.gemm_rowln(dt, b["mO"], w[e + "o.w"], M, 512, bias=w[e + "o.b"], xres=b["tok"], xout=b["tok"], Lseq=S,
                         flags=L.ROW_BIAS | L.ROW_RES | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_H,
                         nln_g=w[e + "norm2.g"], nln_b=w[e + "norm2.b"], nln_eps=1e-5, hout=b["mh"])
            K.gemm_tile(dt, b["mh"], w[e + "l1.w"], M, 1024, 512, bias=w[e + "l1.b"], act=self.act, out=b["mh1"],
                        ldc=1024)
            K.gemm_rowln(dt, b["mh1"], w[e + "l2.w"], M, 1024, bias=w[e + "l2.b"], xres=b["tok"], xout=b["tok"],
                         Lseq=S, flags=L.ROW_BIAS | L.ROW_RES | L.ROW_STORE_X)
        self._hidden_of(b["tok"], B, b["hidden"])
        return b["tok"], b["hidden"]

    def _hidden_of(self, tokens, n, out):
        """non_attn_cond_projection(mean over tokens) (model/model.py:593-597)."""
        dt, w, b = self.dt, self.w, self.b
        K.mean_pool(tokens, b["pool"], n, self.S, 512)
        K.ln_rot(dt, b["pool"], n, w["na.g"], w["na.b"], 1e-5, h=b["pool_h"])
        K.gemm_tile(dt, b["pool_h"], w["na1.w"], n, 512, 512, bias=w["na1.b"], act=L.ACT_SILU, out=b["pool_h2"], ldc=512)
        K.gemm_tile(dt, b["pool_h2"], w["na3.w"], n, 512, 512, bias=w["na3.b"], mode=L.EPI_STORE_F32, out=out, ldc=512)

    def fill_kv_slots(self, tokens: torch.Tensor, n: int, slot0: int):
        """mem rows 0..S-1 = norm_cond(tokens) (model/model.py:615-616); K = rot(mem) Wk, V = mem Wv for every layer
        (model/model.py:386-396,78-80) into cache slots slot0..slot0+n-1."""
        dt, w, b, S = self.dt, self.w, self.b, self.S
        M = n * S
        K.ln_rot(dt, tokens, M, w["nc.g"], w["nc.b"], 1e-5, h=b["mh"], rot=b["mrot"], rope=w["rope"], pos_mod=S)
        for l in range(self.NL):
            K.gemm_tile(dt, b["mrot"], w[f"l{l}.ckv.w"], M, 1024, 512, A2=b["mh"], split_n=512, mode=L.EPI_QKV_HEADS,
                        out=None, out_k=b["Kc"][l], out_v=b["Vc"][l], Lseq=S, Lp=self.Lpc, H=self.H, n_q=0, n_k=512,
                        seq_off=slot0)
        if self.use_full:
            K.pack_kv_frags(b["Kc"], b["Vc"], b["Kf"], b["Vf"], self.NL * b["Kc"].shape[1], self.H, self.Lpc, self.nkt, 0, S)

    def build_time_tables(self, times_i32: torch.Tensor):
        """time path for a set of timesteps (model/model.py:601-605,615-616 rows S,S+1, and their K/V rows):
        t_base = to_time_cond(time_mlp(t)) fp32 [n_t,512]; kv_tab T [NL][n_t][2][1024]."""
        dt, w = self.dt, self.w
        n = times_i32.numel()
        dev, T = self.dev, self.T
        emb = torch.empty(n, 512, device=dev, dtype=T)
        th = torch.empty(n, 2048, device=dev, dtype=T)
        t_base = torch.empty(n, 512, device=dev, dtype=torch.float32)
        ttok = torch.empty(n, 1024, device=dev, dtype=torch.float32)
        th_h = torch.empty(2 * n, 512, device=dev, dtype=T)
        th_r = torch.empty(2 * n, 512, device=dev, dtype=T)
        tab = torch.empty(self.NL, n, 2, 1024, device=dev, dtype=T)
        K.sinusoidal(dt, times_i32, n, w["sin_freq"], emb)
        K.gemm_tile(dt, emb, w["t1.w"], n, 2048, 512, bias=w["t1.b"], act=L.ACT_MISH, out=th, ldc=2048)
        K.gemm_tile(dt, th, w["tc.w"], n, 512, 2048, bias=w["tc.b"], mode=L.EPI_STORE_F32, out=t_base, ldc=512)
        K.gemm_tile(dt, th, w["tt.w"], n, 1024, 2048, bias=w["tt.b"], mode=L.EPI_STORE_F32, out=ttok, ldc=1024)
        K.ln_rot(dt, ttok, 2 * n, w["nc.g"], w["nc.b"], 1e-5, h=th_h, rot=th_r, rope=w["rope"], pos_mod=2,
                 pos_base=self.S)
        for l in range(self.NL):
            K.gemm_tile(dt, th_r, w[f"l{l}.ckv.w"], 2 * n, 1024, 512, A2=th_h, split_n=512, out=tab[l], ldc=1024)
        self.t_base, self.kv_tab, self.n_t = t_base, tab, n
        self.film_tab = None        # belongs to the previous tables
        self.tables_key = None      # callers that cache tables set the key after this returns
        self.reset_graphs()         # the tables moved: captured graphs point at the old ones
        return t_base, tab

    # ------------------------------------------------------------------------------------------
    # one network evaluation
    # ------------------------------------------------------------------------------------------
    def per_step_conditioning(self, n_rows_seq: int):
        """FiLM (scale, shift) of all 24 blocks for every sequence row (model/model.py:154-168,612) and the two
        time-token K/V rows of every layer; b['tidx'][i] selects the timestep row of sequence / cache slot i.
        The cache tensor has 2B slots; slots unused by the current mode receive rows that are never read."""
        dt, w, b = self.dt, self.w, self.b
        K.scatter_time_kv(dt, self.kv_tab, self.n_t, b["tidx"], b["Kc"], b["Vc"], self.NL, b["Kc"].shape[1], self.H,
                          self.Lpc, self.S)
        if self.use_full:
            K.pack_kv_frags(b["Kc"], b["Vc"], b["Kf"], b["Vf"], self.NL * b["Kc"].shape[1], self.H, self.Lpc, self.nkt,
                            self.S, self.S + 2)
        K.add_act(dt, self.t_base, b["tidx"], b["hidden_all"], n_rows_seq, L.ACT_MISH, out=b["film_in"])
        nfilm = self.NL * NL_FILM * 1024
        K.gemm_tile(dt, b["film_in"], w["film.w"], n_rows_seq, nfilm, 512, bias=w["film.b"], mode=L.EPI_STORE_F32,
                    out=b["film"], ldc=nfilm)

    def build_film_table(self, B: int, max_bytes: int = 8 << 30):
        """FiLM (scale | shift) rows of all 24 DenseFiLM blocks for EVERY timestep row of the current time tables and every
        distinct conditioning row of the job -- row 0: the null conditioning (all unconditional rows share it), row 1 + i:
        clip i -- as ONE GEMM per job, [n_t (B + 1), 512] x [24 576, 512]^T (model/model.py:154-168,612).  The sampler's step
        prologue then gathers 2B rows per step instead of a per-step GEMM that re-reads the 25 MB of FiLM weights for 32
        rows.  b['hidden_all'][:B] must hold the null hidden row, [B:2B] the clips' (as _prepare leaves them).  1.67 GB at
        T = 1000, B = 16; beyond `max_bytes` the table is not built and the per-step GEMM stays."""
        nfilm = self.NL * NL_FILM * 1024
        n_t, rows = self.n_t, B + 1
        if n_t * rows * nfilm * 4 > max_bytes:
            self.film_tab = None
            return None
        dev, b, w = self.dev, self.b, self.w
        hid = torch.cat([b["hidden_all"][:1], b["hidden_all"][B:2 * B]], 0)                 # [B + 1, 512]: copies only
        hb = hid.repeat(n_t, 1).contiguous()                                               # row (t, j) -> hidden j
        ia = torch.arange(n_t, device=dev, dtype=torch.int32).repeat_interleave(rows).contiguous()
        fin = torch.empty(n_t * rows, 512, device=dev, dtype=self.T)
        K.add_act(self.dt, self.t_base, ia, hb, n_t * rows, L.ACT_MISH, out=fin)
        tab = getattr(self, "_film_tab_buf", None)        # one buffer per shape, refilled per job: captured graphs keep its address
        if tab is None or tab.shape != (n_t, rows, nfilm):
            tab = self._film_tab_buf = torch.empty(n_t, rows, nfilm, device=dev, dtype=torch.float32)
        K.gemm_tile(self.dt, fin, w["film.w"], n_t * rows, nfilm, 512, bias=w["film.b"], mode=L.EPI_STORE_F32, out=tab,
                    ldc=nfilm)
        self.film_tab = tab
        return tab

    def _xin_shape(self, token_rows: int):
        """(rows, columns, padded pitch) of the model-dtype copy of x_t: one row per token, or -- with the input
        projection folded into the first fusion linear -- one row per frame (dn tokens, contiguous in x)"""
        if self.fold_in:
            return token_rows // self.dn, self.dn * self.nf, self.kin
        return token_rows, self.nf, 192

    def step_prologue(self, st: dict, n_rows_seq: int, x: torch.Tensor, rows: int):
        """One sampler step's prologue in a single launch (timestep lookup, FiLM input, time-token K/V rows, model
        dtype copy of x_t, step counter bump), then the FiLM generator GEMM.  Replaces step_begin +
        per_step_conditioning + network's convert_pad + step_end."""
        dt, w, b = self.dt, self.w, self.b
        full = self.use_full
        nfilm = self.NL * NL_FILM * 1024
        tab = getattr(self, "film_tab", None)
        if tab is not None and tab.shape[1] * 2 - 2 != n_rows_seq:
            tab = None                                # a table of another batch plan: fall back to the per-step GEMM
        extra = {} if tab is None else dict(film_tab=tab, film_out=b["film"], film_rows=tab.shape[1], nfilm=nfilm,
                                            n_unc=n_rows_seq // 2)
        args = (dt, st["counter"], st["rows"], b["tidx"], self.t_base, b["hidden_all"], b["film_in"],
                n_rows_seq, self.kv_tab, self.n_t, None if full else b["Kc"], None if full else b["Vc"],
                b["Kf"] if full else None, b["Vf"] if full else None, self.NL, b["Kc"].shape[1], self.H,
                self.Lpc, self.nkt if full else 0, self.S, x, b["xin"], *self._xin_shape(rows))
        if tab is not None and self.use_chain and self.front and os.environ.get("TCDIFF_FORK_PROLOGUE", "0") == "1":
            # Two parts (tcdiff_step_prologue_args.parts): the x_t copy the first GEMM waits for, and -- on a forked stream, beside
            # the input / fusion GEMMs and the front launch, which leave a third of the CUs idle -- the FiLM rows and time-token rows
            # that nothing reads before layer 0's chain launch ~110 us later (network() joins there).  The fork is part of the
            # captured step: a dependency edge of the graph, no host work per step.
            # MEASURED AND OFF BY DEFAULT (round 6, profiles/r06_prologue_fork_ab.txt): four interleaved pairs of 12-job runs on one
            # box, 14.74 clips/s in line against 14.48 forked (-1.8 %, every pair): the 10-us launch it hides costs more as a
            # concurrent neighbour of the DMA-latency-bound fusion GEMMs and as a fork / join pair in every replayed graph.
            K.step_prologue(*args, parts=L.PROLOGUE_X, **extra)
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.dev)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                K.step_prologue(*args, parts=L.PROLOGUE_COND, **extra)
            self._forked = True
            return
        K.step_prologue(*args, **extra)
        if tab is None:
            K.gemm_tile(dt, b["film_in"], w["film.w"], n_rows_seq, nfilm, 512, bias=w["film.b"], mode=L.EPI_STORE_F32,
                        out=b["film"], ldc=nfilm)

    def network(self, x: torch.Tensor, B: int, branches: int, kv_slot0: int, n_shared: int, film_row0: int,
                x_ready: bool = False):
        """DanceDecoder.forward body after the conditioning prologue (model/model.py:553-561,621-623) for
        `branches` stacked copies of the B clips in x (fp32 [B*L, nfeats]).  Returns b['out'] fp32 [branches*B*L, 152].

        branches=2: rows [uncond | cond]; layer-0 self-attention (which depends on x only) is evaluated once and
        shared.  kv_slot0 / n_shared map sequences to cross-attention cache slots; film_row0 is the first FiLM row."""
        dt, w, b = self.dt, self.w, self.b
        Lq, S, H, dn, NL = self.Lseq, self.S, self.H, self.dn, self.NL
        Rs, R = B * Lq, branches * B * Lq
        nseq = branches * B
        fld = NL * NL_FILM * 1024
        film0 = b["film"][film_row0:]
        rope = w["rope"]
        # input projection + fusion projection over per-frame concatenated dancers (model/model.py:560-561)
        if not x_ready:                      # the sampler's step_prologue already wrote b["xin"]
            K.convert_pad(dt, x, b["xin"], *self._xin_shape(Rs))
        # small jobs (the layers in their four-workgroups-per-block form): these products through the launcher's small-M kernel too
        # -- another summation order, so only together with that kernel family (tcdiff_tile_epi.small_m)
        sm = self._split_job(nseq)
        if self.fold_in:
            K.gemm_tile(dt, b["xin"], w["f1in.w"], B * S, 1024, self.kin, bias=w["f1in.b"], act=L.ACT_RELU, out=b["f1"],
                        ldc=1024, small_m=sm)
        else:
            K.gemm_tile(dt, b["xin"], w["in.w"], Rs, 512, 192, bias=w["in.b"], out=b["xp"], ldc=512, small_m=sm)
            K.gemm_tile(dt, b["xp"], w["f1.w"], B * S, 1024, 512 * dn, bias=w["f1.b"], act=L.ACT_RELU, out=b["f1"],
                        ldc=1024, small_m=sm)
        K.gemm_tile(dt, b["f1"], w["f2.w"], B * S, 1024, 1024, bias=w["f2.b"], act=L.ACT_RELU, out=b["f2"], ldc=1024, small_m=sm)
        # last fusion linear, one group per dancer in ONE launch: group d writes token rows m*dn + d (de-interleave:
        # frame row m, dancer d -> token m*dn + d); fused with layer-0 norm1 + rotary
        self._frag_front = self.front and self._split_job(nseq) and os.environ.get("TCDIFF_SPLIT_FRONT", "1") != "0"
        if self._frag_front:
            # SMALL jobs: the last fusion linear as one plain product -- [frames][512 dn] fp32 IS the token rows [frames dn][512] --
            # then layer 0's norm1 / rotary / Q, K, V as fragment images (tcdiff_chain_split part 0): layer 0's self-attention runs
            # inside its first launch like every other layer's.  (TC_CHAIN_FRONT has 9 workgroups for one 3 x 150 clip.)
            K.gemm_tile(dt, b["f2"], w["f3.w"], B * S, 512 * dn, 1024, bias=w["f3.b"], mode=L.EPI_STORE_F32, out=b["xs"],
                        ldc=512 * dn, small_m=True)
            K.chain(L.CHAIN_FRONT, Rs, Lq, None, w["front"][0], split_part=0, xres=b["xs"], nn_g=w["l0.norm1.g"],
                    nn_b=w["l0.norm1.b"], nn_eps=1e-5, rope=w["rope_cb"], qf_out=b["Qf"], kf_out=b["sKf"][0], vf_out=b["sVf"][0],
                    out_nkt=self.skt, scale_q=0.125, H=H)
        elif self.front:
            # ... as ONE chain launch per (64-frame block, dancer) that also runs layer 0's norm1, rotary and Q / K / V
            # projections; b["xs"] (layer 0's residual input) is then column-blocked like the rest of the stream
            K.chain(L.CHAIN_FRONT, B * S, Lq, b["f2"], w["front"], b3=w["f3.b"], nn_g=w["l0.norm1.g"],
                    nn_b=w["l0.norm1.b"], nn_eps=1e-5, rope=w["rope_cb"], xout=b["xs"], q_out=b["Q"], k_out=b["K"],
                    v_out=b["V"], scale_q=0.125, Lp=self.Lp, H=H, dn=dn)
        elif self.abs_pos:
            # x = abs_pos_encoding(x) sits between the fusion projection and layer 0's norm1 (model/model.py:564): three launches
            K.gemm_rowln(dt, b["f2"], w["f3.w"], B * S, 1024, bias=w["f3.b"], xout=b["xs"], Lseq=Lq,
                         flags=L.ROW_BIAS | L.ROW_STORE_X, out_mul=dn, out_add=0, groups=dn)
            K.add_rows(b["xs"], 512, b["pe_x"], 512, b["xs"], 512, Rs, 512)
            K.ln_rot(dt, b["xs"], Rs, w["l0.norm1.g"], w["l0.norm1.b"], 1e-5, h=b["h"], rot=b["rot"], rope=rope, pos_mod=Lq)
        else:
            K.gemm_rowln(dt, b["f2"], w["f3.w"], B * S, 1024, bias=w["f3.b"], xout=b["xs"], Lseq=Lq,
                         flags=L.ROW_BIAS | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_H | L.ROW_STORE_ROT,
                         nln_g=w["l0.norm1.g"], nln_b=w["l0.norm1.b"], nln_eps=1e-5, hout=b["h"], rout=b["rot"],
                         rope=rope, out_mul=dn, out_add=0, groups=dn)
        Kc0 = b["Kc"][:, kv_slot0:]
        Vc0 = b["Vc"][:, kv_slot0:]
        if self._forked:                           # the conditioning part of the step prologue (step_prologue): first needed here
            torch.cuda.current_stream().wait_stream(self._side)
            self._forked = False
        for l in range(NL):
            p = f"l{l}."
            rows_sa = Rs if l == 0 else R          # layer-0 self-attention is branch-independent
            nseq_sa = B if l == 0 else nseq
            if self.use_chain:
                self._layer_chained(l, B, branches, Kc0, Vc0, film0, fld, n_shared, kv_slot0)
                continue
            # ---- self-attention block (model/model.py:326-327,374-383,71-107)
            K.gemm_tile(dt, b["rot"], w[p + "qkv.w"], rows_sa, 1536, 512, A2=b["h"], split_n=1024, mode=L.EPI_QKV_HEADS,
                        out=b["Q"], out_k=b["K"], out_v=b["V"], scale_q=0.125, Lseq=Lq, Lp=self.Lp, H=H, n_q=512,
                        n_k=512)
            K.attention(dt, b["Q"], b["K"], b["V"], b["O"], nseq_sa, H, Lq, Lq, self.Lp, self.Lp, 512)
            K.gemm_rowln(dt, b["O"], w[p + "sfc.w"], R, 512, a_mod=Rs if l == 0 else 0,
                         flags=L.ROW_LN_POST | L.ROW_FILM | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_ROT,
                         ln_g=w[p + "sln.g"], ln_b=w[p + "sln.b"], ln_eps=1e-6, film=film0[:, (l * 3 + 0) * 1024:],
                         film_ld=fld, xres=b["xs"] if l == 0 else b["xa"], xres_mod=Rs if l == 0 else 0, xout=b["xa"],
                         Lseq=Lq, nln_g=w[p + "norm2.g"], nln_b=w[p + "norm2.b"], nln_eps=1e-5, rout=b["rot"], rope=rope)
            # ---- cross-attention block (model/model.py:331-334,386-396)
            K.gemm_tile(dt, b["rot"], w[p + "cq.w"], R, 512, 512, mode=L.EPI_QKV_HEADS, out=b["Q"], out_k=None,
                        out_v=None, scale_q=0.125, Lseq=Lq, Lp=self.Lp, H=H, n_q=512, n_k=0)
            K.attention(dt, b["Q"], Kc0[l], Vc0[l], b["O"], nseq, H, Lq, S + 2, self.Lp, self.Lpc, 512,
                        n_shared=n_shared)
            K.gemm_rowln(dt, b["O"], w[p + "cfc.w"], R, 512,
                         flags=L.ROW_LN_POST | L.ROW_FILM | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_H,
                         ln_g=w[p + "cln.g"], ln_b=w[p + "cln.b"], ln_eps=1e-6, film=film0[:, (l * 3 + 1) * 1024:],
                         film_ld=fld, xres=b["xa"], xout=b["xa"], Lseq=Lq, nln_g=w[p + "norm3.g"],
                         nln_b=w[p + "norm3.b"], nln_eps=1e-5, hout=b["h"])
            # ---- feed-forward block (model/model.py:338-339,399-401)
            K.gemm_tile(dt, b["h"], w[p + "ff1.w"], R, 1024, 512, bias=w[p + "ff1.b"], act=self.act, out=b["h1"],
                        ldc=1024)
            K.gemm_rowln(dt, b["h1"], w[p + "ff2.w"], R, 1024, flags=L.ROW_BIAS | L.ROW_FILM | L.ROW_NEXT_LN | L.ROW_STORE_H,
                         bias=w[p + "ff2.b"], film=film0[:, (l * 3 + 2) * 1024:], film_ld=fld, xres=b["xa"], Lseq=Lq,
                         nln_g=w[p + "norm4.g"], nln_b=w[p + "norm4.b"], nln_eps=1e-5, hout=b["h"])
            # ---- x = linear3(norm4(x)), no residual (model/model.py:344); fused with the next layer's norm1+rotary
            if l + 1 < NL:
                n1 = f"l{l + 1}.norm1."
                K.gemm_rowln(dt, b["h"], w[p + "l3.w"], R, 512, bias=w[p + "l3.b"], xout=b["xa"], Lseq=Lq,
                             flags=L.ROW_BIAS | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_H | L.ROW_STORE_ROT,
                             nln_g=w[n1 + "g"], nln_b=w[n1 + "b"], nln_eps=1e-5, hout=b["h"], rout=b["rot"], rope=rope)
            else:
                K.gemm_rowln(dt, b["h"], w[p + "l3.w"], R, 512, bias=w[p + "l3.b"], Lseq=Lq,
                             flags=L.ROW_BIAS | L.ROW_STORE_H, hout=b["h"])
        if self.fold_out:
            return b["out"]      # written by the last chain launch (final_layer folded into its linear3)
        # final layer (model/model.py:623)
        K.gemm_tile(dt, b["h"], w["fin.w"], R, self.nf, 512, bias=w["fin.b"], mode=L.EPI_STORE_F32, out=b["out"], ldc=152)
        return b["out"]

    def _split_job(self, nseq: int) -> bool:
        """whether this forward runs the decoder layers in their small-job form (csrc/chain_split.hip)"""
        return bool(self.use_full and self.fuse_sa and self.chain_nw == 8 and "xb" in self.b and self._split_rows(nseq, self.Lseq))

    @staticmethod
    def _merge12(Lq: int) -> bool:
        """parts 1 + 2 of the small-job layer as one launch (tcdiff_chain_split part 12): TCDIFF_SPLIT_MERGE=0 never, =1 always,
        default: sequences of at most MERGE12_MAX_L tokens (every member then streams the sequence's K / V for all eight heads)"""
        v = os.environ.get("TCDIFF_SPLIT_MERGE", "")
        return v == "1" or (v != "0" and Lq <= MERGE12_MAX_L)

    def _split_rows(self, nseq: int, Lq: int) -> bool:
        """the small-job form of the layer (four workgroups per 16-row block): when all of them fit the chip at once"""
        if os.environ.get("TCDIFF_SPLIT", "1") == "0" or self.chain_nw != 8 or Lq < 16:
            return False
        return 4 * nseq * ((Lq + 15) // 16) <= self.n_cu

    def _layer_chained(self, l: int, B: int, branches: int, Kc0, Vc0, film0, fld: int, n_shared: int, kv_slot0: int):
        """One decoder layer as attention / chain A / attention / chain B (csrc/chain.hip): the Q, K, V images of
        this layer's self-attention were written by the previous layer's chain B (layer 0: by the QKV GEMM below)."""
        dt, w, b = self.dt, self.w, self.b
        Lq, S, H, NL = self.Lseq, self.S, self.H, self.NL
        Rs, R = B * Lq, branches * B * Lq
        nseq = branches * B
        p = f"l{l}."
        rope = w["rope_cb"]
        if l == 0 and not self.front:
            K.gemm_tile(dt, b["rot"], w[p + "qkv.w"], Rs, 1536, 512, A2=b["h"], split_n=1024, mode=L.EPI_QKV_HEADS,
                        out=b["Q"], out_k=b["K"], out_v=b["V"], scale_q=0.125, Lseq=Lq, Lp=self.Lp, H=H, n_q=512,
                        n_k=512)
        # fused: layers 1.. compute their self-attention inside the chain launch from the fragments the previous launch wrote
        fused = self.fuse_sa and self.chain_nw == 8
        frag0 = l == 0 and getattr(self, "_frag_front", False)      # layer 0's Q / K / V came as fragment images (network())
        if not (fused and l > 0) and not frag0:
            K.attention(dt, b["Q"], b["K"], b["V"], b["O"], B if l == 0 else nseq, H, Lq, Lq, self.Lp, self.Lp, 512)
        last = l + 1 == NL
        nn = f"l{l + 1}.norm1." if not last else None
        tail = dict(b1=w[p + "ff1.b"], film3=film0[:, (l * 3 + 2) * 1024:], n4_g=w[p + "norm4.g"],
                    n4_b=w[p + "norm4.b"], n4_eps=1e-5, b3=w[p + "l3out.b"] if last and self.fold_out else w[p + "l3.b"],
                    nn_g=None if last else w[nn + "g"], nn_b=None if last else w[nn + "b"], nn_eps=1e-5,
                    q_out=None if last else b["Q"], k_out=None if last else b["K"], v_out=None if last else b["V"],
                    h_out=(b["out"] if self.fold_out else b["h"]) if last else None,
                    out_ld=152 if last and self.fold_out else 0, scale_q=0.125, Lp=self.Lp, H=H)
        if fused:
            # the smallest row blocks that still give every block its own CU (a block streams the layer's weights whatever its rows)
            tail.update(seq_blocks=True, mt=next((m for m in (1, 2) if nseq * ((Lq + 16 * m - 1) // (16 * m)) <= self.n_cu), 4))
            if l > 0 or frag0:
                tail.update(sa_q=b["Qf"], sa_kf=b["sKf"][l & 1], sa_vf=b["sVf"][l & 1], sa_nkt=self.skt)
            if not last:
                tail.update(q_out=None, k_out=None, v_out=None, qf_out=b["Qf"], kf_out=b["sKf"][(l + 1) & 1],
                            vf_out=b["sVf"][(l + 1) & 1], out_nkt=self.skt)
        head = dict(a_mod=Rs if l == 0 else 0, ln_eps=1e-6,     # (film rows: pre-folded with sln / cln / ff2.b, load_weights)
                    film=film0[:, (l * 3 + 0) * 1024:], film_ld=fld, xres=b["xs"] if l == 0 else b["xa"],
                    xres_mod=Rs if l == 0 else 0, xres_rowmajor=l == 0 and (not self.front or frag0), xout=b["xa"],
                    n2_g=w[p + "norm2.g"],
                    n2_b=w[p + "norm2.b"], n2_eps=1e-5, rope=rope)     # b["xa"] is COLUMN-BLOCKED on this path
        if self._split_job(nseq):
            # SMALL jobs (csrc/chain_split.hip): four workgroups per 16-row block, four launches per layer; the residual stream
            # alternates between b["xa"] and b["xb"] (a part reads whole rows and stores quarters), the partial sums between two slabs
            full = dict(filmb=film0[:, (l * 3 + 1) * 1024:], n3_g=w[p + "norm3.g"], n3_b=w[p + "norm3.b"], kf=b["Kf"][l, kv_slot0:],
                        vf=b["Vf"][l, kv_slot0:], n_shared=n_shared, nkt=self.nkt, Lk=S + 2)
            args = {**head, **tail, **full}
            args["mt"] = 1
            mode = L.CHAIN_FULL_LAST if last else L.CHAIN_FULL
            X, P = (b["xa"], b["xb"]), (b["P0"], b["P1"])
            cur = self._xcur                           # X[cur] holds this layer's input x (layer 0 reads b["xs"] instead)
            o1 = 0 if l == 0 else cur ^ 1
            launch = lambda part, **kw: K.chain(mode, R, Lq, b["O"], w[p + "chainF"], split_part=part, **{**args, **kw})
            x_in = {} if l == 0 else dict(xres=X[cur])                                         # layer 0: xres = b["xs"] (head)
            if self._merge12(Lq):
                # short sequences: self-attention of all eight heads and the whole fc in every member, one exchange less
                launch(12, p_out=P[1], xout=X[o1], **x_in)
            else:
                launch(1, p_out=P[0])
                launch(2, p_in=P[0], p_out=P[1], xout=X[o1], **x_in)
            flat = dict(xres_mod=0, xres_rowmajor=False)
            launch(3, p_in=P[1], p_out=P[0], xres=X[o1], xout=X[o1 ^ 1], **flat)
            launch(4, p_in=P[0], xres=X[o1 ^ 1], xout=X[o1], **flat)
            self._xcur = o1
            return
        if self.use_full:
            # self-attention tail, cross-attention (K / V from the fragment-ordered caches) and feed-forward in ONE launch
            K.chain(L.CHAIN_FULL_LAST if last else L.CHAIN_FULL, R, Lq, b["O"], w[p + "chainF"],
                    filmb=film0[:, (l * 3 + 1) * 1024:], n3_g=w[p + "norm3.g"],
                    n3_b=w[p + "norm3.b"], kf=b["Kf"][l, kv_slot0:], vf=b["Vf"][l, kv_slot0:], n_shared=n_shared,
                    nkt=self.nkt, Lk=S + 2, **head, **tail)
            return
        K.chain(L.CHAIN_A, R, Lq, b["O"], w[p + "chainA"], **head, q_out=b["Q"], scale_q=0.125, Lp=self.Lp, H=H)
        K.attention(dt, b["Q"], Kc0[l], Vc0[l], b["O"], nseq, H, Lq, S + 2, self.Lp, self.Lpc, 512, n_shared=n_shared)
        K.chain(L.CHAIN_B_LAST if last else L.CHAIN_B, R, Lq, b["O"], w[p + "chainB"],
                ln_eps=1e-6, film=film0[:, (l * 3 + 1) * 1024:], film_ld=fld,
                xres=b["xa"], xout=b["xa"], n2_g=w[p + "norm3.g"], n2_b=w[p + "norm3.b"], n2_eps=1e-5, rope=rope, **tail)
