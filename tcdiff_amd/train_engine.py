"""Training step of the TCDiff denoiser on MI355X: train-mode forward and the backward pass as an explicit schedule of
HIP launches (libtcdiff_gfx950.so), behind ONE ``torch.autograd.Function``.

The reference trains with torch autograd over ``DanceDecoder.forward`` (model/model.py:548-624, train mode: nn.Dropout at
:98,103,240,244-245,383,396,400-401 and inside nn.MultiheadAttention) and ``accelerator.backward(total_loss)``
(TCDiff.py:227-234).  Here ``DanceDecoder.forward`` with gradients enabled calls :func:`denoiser_train`, whose forward runs
the kernels below and keeps the activations the reverse pass needs, and whose backward walks the same graph in reverse:

  nn.Linear            forward  tcdiff_gemm_tile;  dgrad = tcdiff_gemm_tile against the transposed weight pack;
                       wgrad = dY^T X straight from the token-major dY and X: queued per decoder layer and flushed as ONE
                       evenly split launch (tcdiff_gemm_tn_grouped); shapes that form does not take are repacked by
                       tcdiff_cast_transpose for tcdiff_gemm_splitk
  attention            tcdiff_attention_train / tcdiff_attention_bwd (weights recomputed, dropout bits regenerated)
  everything between   tcdiff_row_fwd / tcdiff_row_bwd (bias, dropout, LayerNorms, FiLM, residual, rotary; the backward adds the
                       LayerNorm / FiLM / bias gradients itself); activations (+ dropout) ride in the neighbouring GEMMs'
                       epilogues; the conditioning path's selects / pool / adds have their own adjoints

From the third step with the same shapes on, forward and backward are replayed hipGraphs captured from this very schedule
(``TrainEngine.use_graphs``): the ~550 launches cost more host time through Python than device time on a slow host.

Gradients land in ONE flat fp32 buffer (fused linears contiguous) of which every parameter's ``.grad`` is set to a view: the
data-parallel all-reduce (tcdiff_amd/dist.py) runs on slices of that buffer without a copy.  torch contributes device
memory, the stream and the autograd hook -- no arithmetic.  There is no CPU path: off-GPU this module raises.

One forward may be outstanding per model (the engine keeps its activations until the matching backward), which is what
TCDiff.train_loop does (TCDiff.py:227-234).
"""
from __future__ import annotations

import math
import re
import warnings
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import kernels as K

DEAD = ("traj_Modulation", "traj_embedding", "embeddings_table")      # parameters the forward never uses (model/model.py:346-355,371,557)
import os as _os
# row_bwd blocks per sequence.  The kernel holds 196 VGPRs (7 column accumulators + a row in flight), i.e. two waves per SIMD =
# ONE 8-wave block per CU: at 32 sequences 8 blocks each are exactly one resident round, and every further block only adds
# adders to the gradient atomics (batch 32, ms per step in row_bwd: 4 blocks 2.06, 8: 1.46, 16: 1.80, 32: 2.25).
_ROWB_CHUNKS = int(_os.environ.get("TCDIFF_ROWB_CHUNKS", "8"))
_ALLOW_CPU = False          # tools/dryrun_train.py only: host-side dry run of the schedule against a stub library
_ROWS_KEY = re.compile(r"^l\d+\.")      # linears evaluated by tcdiff_gemm_rows: the decoder layers' (see _Lin)


def _is_dead(name: str) -> bool:
    return any(d in name for d in DEAD)


# dropout sites of PositionalEncoding (use_rotary=False); 0-7: the music encoder's, 16 + 8 l ..: decoder layer l's (oracle DropPlan)
SITE_PE_X, SITE_PE_COND = 8, 9

class _Lin:
    """One nn.Linear, or several stacked along the output dimension and evaluated as one GEMM."""

    def __init__(self, eng: "TrainEngine", key: str, wnames: Sequence[str], bnames: Optional[Sequence[str]] = None,
                 split: int = 0):
        self.eng, self.key, self.wnames, self.bnames, self.split = eng, key, list(wnames), list(bnames or []), split
        ps = [eng.params[n] for n in self.wnames]
        self.K = ps[0].shape[1]
        self.rows = [p.shape[0] for p in ps]
        self.N = sum(self.rows)
        self.Kp = K.round_up(self.K, eng.kt)
        self.groups = [(0, split), (split, self.N)] if split else [(0, self.N)]
        # The decoder layers' linears (tall products: every token row against K = 512 / 1024) go through the row-block GEMM
        # (tcdiff_gemm_rows): their operand packs are per-wave fragment streams, forward order and input-gradient order.
        # (head-major scatters cross at most one sequence boundary per 64-row block: sequences of >= 64 tokens)
        self.use_rows = bool(eng.use_rows and _ROWS_KEY.match(key) and eng.Lq >= 64 and K.gemm_rows_ok(eng.dt, self.N, self.K) and
                             all(r % 512 == 0 for r in self.rows) and all(hi - lo in (512, 1024) for lo, hi in self.groups))
        if self.use_rows:
            self.Wf, self.WbT = None, None
            self.Ws = torch.zeros(8, (self.N // 512) * (self.K // 32), 2048, device=eng.dev, dtype=eng.T)
            self.WsT = [torch.zeros(8, (self.K // 512) * ((hi - lo) // 32), 2048, device=eng.dev, dtype=eng.T)
                        for lo, hi in self.groups]
        else:
            self.Wf = torch.zeros(self.N, self.Kp, device=eng.dev, dtype=eng.T)
            self.WbT = [torch.zeros(self.K, K.round_up(hi - lo, eng.kt), device=eng.dev, dtype=eng.T) for lo, hi in self.groups]
        # stacked biases need one contiguous copy; a lone bias is the parameter itself
        self.bias = torch.zeros(self.N, device=eng.dev, dtype=torch.float32) if len(self.bnames) > 1 else None

    def pack_entries(self):
        """Descriptors of tcdiff_cast_transpose_multi that (re)build this linear's operand packs from the fp32 master
        parameters: W as [N, Kp] and, per operand group, W^T."""
        eng, r0, out = self.eng, 0, []
        if self.use_rows:
            return out
        for name, rows in zip(self.wnames, self.rows):
            w = eng.params[name].detach()
            # a parameter that spans the operand split (nn.MultiheadAttention's packed in_proj_weight) is packed in two pieces
            cuts = [0, self.split - r0, rows] if (self.split and r0 < self.split < r0 + rows) else [0, rows]
            for a, b in zip(cuts[:-1], cuts[1:]):
                gi = 1 if (self.split and r0 + a >= self.split) else 0
                lo = self.groups[gi][0]
                wt = self.WbT[gi]
                out.append(dict(src=w[a:b], rows=b - a, cols=self.K, ld_src=self.K, dst=self.Wf[r0 + a:], ld_dst=self.Kp,
                                cols_pad=self.Kp, dstT=wt.view(-1)[r0 + a - lo:], ld_dstT=wt.shape[1], rows_pad=b - a))
            r0 += rows
        return out

    def stream_entries(self):
        """Descriptors of tcdiff_pack_row_streams for this linear's two stream packs: Wn = W going forward (a stacked
        parameter is a range of 512-column phases), Wn = W[lo:hi]^T per operand group for the input gradient (a stacked
        parameter is a range of k-steps)."""
        eng, r0, out = self.eng, 0, []
        if not self.use_rows:
            return out
        Kd = self.K
        for name, rows in zip(self.wnames, self.rows):
            w = eng.params[name].detach()
            out.append(dict(src=w, sn=Kd, sk=1, N=rows, K=Kd, dst=self.Ws, np_dst=self.N // 512, p0=r0 // 512))
            for gi, (lo, hi) in enumerate(self.groups):
                a, b = max(lo, r0) - r0, min(hi, r0 + rows) - r0          # this parameter's rows inside the group
                if a < b:
                    out.append(dict(src=w[a:b], sn=1, sk=Kd, N=Kd, K=b - a, dst=self.WsT[gi], kst_dst=(hi - lo) // 32,
                                    ks0=(r0 + a - lo) // 32))
            r0 += rows
        return out

    def pack_bias(self):
        eng = self.eng
        if len(self.bnames) > 1:
            torch.cat([eng.params[n].detach().reshape(-1) for n in self.bnames], out=self.bias)
        elif self.bnames:
            self.bias = eng.params[self.bnames[0]].detach()

    # ---- forward ---------------------------------------------------------------------------------------------------
    def fwd(self, A, M, *, A2=None, out=None, f32=False, ldc=None, heads=None, rows=None, act=None):
        """out[M, N] = A W^T + b.  f32: fp32 output, else T.  heads = dict(out, out_k, out_v, scale_q, Lseq, Lp, n_q, n_k):
        scatter to head-major images.  rows = (lo, hi): only that slice of the stacked outputs.
        act = (TC_ACT_*, dropout site or None): the activation (+ nn.Dropout) that follows, in the GEMM's epilogue -- returns
        (out, act_out); shapes the epilogue does not take run tcdiff_act_drop as a second launch."""
        eng = self.eng
        if act is not None:
            kind, site = act
            thr, sc = (eng.thr, eng.dscale) if site is not None else (0, 1.0)
            y = eng.e(M, out.shape[1])
            ld = ldc if ldc else self.N
            if self.use_rows:
                K.gemm_rows(A, self.Ws, M, self.N, self.K, lda=A.shape[1], bias=self.bias, out=out, ldc=ld, out2=y, ldc2=ld, act2=kind,
                            seed=eng.seed, site=site or 0, thr=thr, drop_scale=sc)
            elif self.N % (16 // out.element_size()) == 0 and ld == out.shape[1] and not eng.no_fuse:
                K.gemm_tile(eng.dt, A, self.Wf, M, self.N, self.Kp, bias=self.bias, mode=L.EPI_STORE_T, out=out, ldc=ld, out2=y,
                            ldc2=ld, act2=kind, seed=eng.seed, site=site or 0, thr=thr, drop_scale=sc)
            else:
                K.gemm_tile(eng.dt, A, self.Wf, M, self.N, self.Kp, bias=self.bias, mode=L.EPI_STORE_T, out=out, ldc=ld)
                K.act_drop(eng.dt, out, out.shape[1], y, y.shape[1], M, self.N, kind, eng.seed, site or 0, thr, sc)
            return out, y
        if self.use_rows:
            if rows:
                raise L.TcdiffError("a row-streamed linear is evaluated whole")
            if heads:
                K.gemm_rows(A, self.Ws, M, self.N, self.K, lda=A.shape[1], A2=A2, split_n=self.split if A2 is not None else 0,
                            bias=self.bias, mode=L.EPI_QKV_HEADS, H=eng.H, **heads)
                return None
            K.gemm_rows(A, self.Ws, M, self.N, self.K, lda=A.shape[1], A2=A2, split_n=self.split if A2 is not None else 0,
                        bias=self.bias, mode=L.EPI_STORE_F32 if f32 else L.EPI_STORE_T, out=out, ldc=ldc if ldc else self.N)
            return out
        lo, hi = rows if rows else (0, self.N)
        W = self.Wf[lo:hi]
        bias = self.bias[lo:hi] if self.bias is not None else None
        if heads:
            K.gemm_tile(eng.dt, A, W, M, hi - lo, self.Kp, A2=A2, split_n=self.split if A2 is not None else 0, bias=bias,
                        mode=L.EPI_QKV_HEADS, H=eng.H, **heads)
            return None
        K.gemm_tile(eng.dt, A, W, M, hi - lo, self.Kp, A2=A2, split_n=self.split if A2 is not None else 0, bias=bias,
                    mode=L.EPI_STORE_F32 if f32 else L.EPI_STORE_T, out=out, ldc=ldc if ldc else hi - lo)
        return out

    # ---- backward --------------------------------------------------------------------------------------------------
    def bwd(self, dY, ld_dy, M, Xs, want, bias_done=False):
        """dY [M, N] (fp32 or T, leading dimension ld_dy).  Xs[g]: T operand [M, Kp] of group g.  want[g]: None, or
        ("T" | "F32", out tensor, ldc) or ("HEADS", dict) -- where the input gradient of group g goes.
        Weight gradients accumulate into the flat gradient buffer; the bias gradient is the column sum of dY."""
        eng = self.eng
        dt, kt = eng.dt, eng.kt
        N, Mp = self.N, K.round_up(M, kt)
        Np = K.round_up(N, kt)
        gW = eng.gW[self.key]
        gb = None if bias_done else eng.gB.get(self.key)      # bias_done: the producer of dY summed its columns already
        direct = dY.dtype == eng.T and ld_dy == Np and N == Np     # dY itself is a valid K-contiguous operand
        # weight gradient straight from the token-major dY and X (tcdiff_gemm_tn) where the shapes allow; otherwise dY^T / X^T
        # are repacked by cast_transpose for tcdiff_gemm_splitk
        tn = direct and M == Mp and all(K.gemm_tn_ok(dt, hi - lo, self.K, M) and Xs[gi].dtype == eng.T
                                        for gi, (lo, hi) in enumerate(self.groups))
        dYt = None
        if tn:
            dYT = dY
            if gb is not None:
                K.cast_transpose(dt, dY, M, N, ld_dy, colsum=gb)
        elif direct:
            dYT, dYt = dY, eng.e(N, Mp)
            K.cast_transpose(dt, dY, M, N, ld_dy, dstT=dYt, ld_dstT=Mp, rows_pad=Mp, colsum=gb)
        else:
            dYT, dYt = eng.e(M, Np), eng.e(N, Mp)
            K.cast_transpose(dt, dY, M, N, ld_dy, dst=dYT, ld_dst=Np, cols_pad=Np, dstT=dYt, ld_dstT=Mp, rows_pad=Mp,
                             colsum=gb)
        outs = []
        for gi, (lo, hi) in enumerate(self.groups):
            ng = hi - lo
            ngp = self.WbT[gi].shape[1] if not self.use_rows else ng
            w = want[gi]
            if w is not None and self.use_rows:                 # dX_g = dY[:, lo:hi] W[lo:hi], W read as its transposed stream
                A = dYT.view(-1)[lo:]
                if w[0] == "HEADS":
                    K.gemm_rows(A, self.WsT[gi], M, self.K, ng, lda=Np, mode=L.EPI_QKV_HEADS, H=eng.H, **w[1])
                elif w[0] == "ACT":
                    _, da, ld, a_src, kind, site = w
                    thr, sc = (eng.thr, eng.dscale) if site is not None else (0, 1.0)
                    if a_src.dtype != eng.T or da.dtype != eng.T:
                        raise L.TcdiffError("the fused activation backward takes T-typed operands")
                    K.gemm_rows(A, self.WsT[gi], M, self.K, ng, lda=Np, out=da, ldc=ld, act_src=a_src, ld_src=a_src.shape[1],
                                act2=kind, seed=eng.seed, site=site or 0, thr=thr, drop_scale=sc)
                else:
                    K.gemm_rows(A, self.WsT[gi], M, self.K, ng, lda=Np, mode=L.EPI_STORE_F32 if w[0] == "F32" else L.EPI_STORE_T,
                                out=w[1], ldc=w[2])
            elif w is not None:                                 # dX_g = dY[:, lo:hi] W[lo:hi]
                A = dYT.view(-1)[lo:]
                if w[0] == "HEADS":
                    K.gemm_tile(dt, A, self.WbT[gi], M, self.K, ngp, lda=Np, mode=L.EPI_QKV_HEADS, H=eng.H, **w[1])
                elif w[0] == "ACT":                             # ("ACT", da, ld, a, TC_ACT_*, site): through the activation's backward
                    _, da, ld, a_src, kind, site = w
                    thr, sc = (eng.thr, eng.dscale) if site is not None else (0, 1.0)
                    if self.K % (16 // da.element_size()) == 0 and a_src.dtype == eng.T and da.dtype == eng.T and not eng.no_fuse:
                        K.gemm_tile(dt, A, self.WbT[gi], M, self.K, ngp, lda=Np, mode=L.EPI_STORE_T, out=da, ldc=ld,
                                    act_src=a_src, ld_src=a_src.shape[1], act2=kind, seed=eng.seed, site=site or 0, thr=thr,
                                    drop_scale=sc)
                    else:
                        dy = eng.e(M, ld)
                        K.gemm_tile(dt, A, self.WbT[gi], M, self.K, ngp, lda=Np, mode=L.EPI_STORE_T, out=dy, ldc=ld)
                        K.act_drop_bwd(dt, a_src, a_src.shape[1], dy, ld, da, M, self.K, kind, eng.seed, site or 0, thr, sc)
                elif M <= 128 and ngp >= 4096 and w[2] == self.K:
                    # one row panel over a long contraction (the 24 FiLM generators' input gradient: 32 rows x 24 576): as a plain
                    # tile GEMM four workgroups walk 384 k-tiles each (242 us); split over the contraction with fp32 atomics
                    # every CU takes six (then one small cast for a T-typed consumer)
                    acc = w[1] if w[0] == "F32" else eng.e(M, self.K, dtype=torch.float32)
                    eng.tt(lambda acc=acc: acc.zero_())
                    K.gemm_splitk(dt, A, self.WbT[gi], M, self.K, ngp, Np, ngp, acc, self.K, max(1, min(64, ngp // kt)))
                    if w[0] != "F32":
                        K.cast_transpose(dt, acc, M, self.K, self.K, dst=w[1], ld_dst=w[2], cols_pad=self.K)
                else:
                    K.gemm_tile(dt, A, self.WbT[gi], M, self.K, ngp, lda=Np,
                                mode=L.EPI_STORE_F32 if w[0] == "F32" else L.EPI_STORE_T, out=w[1], ldc=w[2])
            X = Xs[gi]
            tiles = ((ng + 127) // 128) * ((self.K + 127) // 128)
            # one workgroup per CU: measured (tools/gemm_shapes.py, 14 400 rows) 512x512: 16 splits 29 us, 32 splits 40 us, 1: 149 us
            splits = max(1, min(256 // tiles, Mp // kt, 64))
            if tn:
                if eng.group_wgrad:       # queued: all weight gradients of the layer go out as one evenly split launch
                    eng.queue_wgrad((dY.view(-1)[lo:], X, ng, self.K, M, ld_dy, X.shape[1], gW[lo * self.K:], self.K))
                else:
                    K.gemm_tn(dt, dY.view(-1)[lo:], X, ng, self.K, M, ld_dy, X.shape[1], gW[lo * self.K:], self.K, splits)
                continue
            Xt = eng.e(self.K, Mp)
            K.cast_transpose(dt, X, M, self.K, X.shape[1], dstT=Xt, ld_dstT=Mp, rows_pad=Mp)
            K.gemm_splitk(dt, dYt[lo:], Xt, ng, self.K, Mp, Mp, Mp, gW[lo * self.K:], self.K, splits)
        return outs


class TrainEngine:
    """Packed weights, flat gradient buffer and the forward / backward schedule of one DanceDecoder."""

    def __init__(self, model, compute: str):
        L.load()
        self.model = model
        self.params: Dict[str, torch.nn.Parameter] = dict(model.named_parameters())
        self.param_ids = tuple(id(p) for p in self.params.values())     # model.train_engine() rebuilds when objects are replaced
        p0 = next(iter(self.params.values()))
        if p0.device.type != "cuda" and not _ALLOW_CPU:
            raise L.TcdiffError("the training step runs on MI355X only (no CPU fallback; the CPU oracle is test-only)")
        self.dev = p0.device
        self.dt = K.dtype_id(compute)
        if self.dt == L.DT_BF16X3:
            # the split-bf16 arithmetic exists for the sampler's GEMM / attention launchers only; a training step of a model in that
            # mode runs the exact-fp32 schedule (same storage, same gradients as compute_dtype="f32")
            self.dt = L.DT_F32
        self.mode_dt = K.dtype_id(compute)      # what model.train_engine() compares with the model's mode
        self.act = int(getattr(model, "act_id", L.ACT_GELU))      # feed-forward activation (model/model.py:244,400)
        self.T = K.TORCH_DT[self.dt]
        self.kt = K.k_tile(self.dt)
        c = model.engine_config()
        if c["latent"] != 512 or c["n_head"] * 64 != 512:
            raise L.TcdiffError(f"latent_dim={c['latent']}, num_heads={c['n_head']}: the gfx950 kernels are built for the width the "
                                "reference instantiates, latent_dim=512 as 8 heads x 64 (TCDiff.py:76-87); the constructor's own "
                                "defaults (256 / 4, model/model.py:417-431) and any other width are not built")
        self.H, self.NL, self.S, self.dn, self.nf, self.ff = c["n_head"], c["n_layers"], c["seq_len"], c["dn"], \
            c["nfeats"], c["ff"]
        self.Cd = c["cond_dim"]
        self.Lq = self.S * self.dn
        self.Lp = K.round_up(self.Lq, 128)
        self.Lps = K.round_up(self.S, 128)
        self.Lpc = K.round_up(self.S + 2, 128)
        self.p_drop = float(getattr(model, "dropout_p", 0.1))
        self.seed = torch.zeros(2, device=self.dev, dtype=torch.int32)
        self.packed_version = None
        self._ct = self._ct_ptrs = self._ct_keep = None
        self._ws = self._ws_keep = None
        self._wq = []
        self._graphs, self._pool, self._graph_broken = {}, None, None
        self._gv = self._gv_flat = None
        self.sv = None
        self._gen = 0
        self._pz = {}
        self._define()
        half = 256
        self.sin_freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(self.dev)   # model/utils.py:43-44
        n_pos = max(self.Lq, self.S + 2)
        self.rope = torch.empty(n_pos, 512, device=self.dev, dtype=torch.float32)
        # use_rotary=False (model/model.py:441-448): no rotation anywhere (angle 0: cos 1, sin 0 -- the same launches) and
        # PositionalEncoding, with its own nn.Dropout, on the motion tokens and the music tokens (model/model.py:564,580)
        self.abs_pos = not getattr(model, "use_rotary", True)
        if self.abs_pos:
            K.rope_table(torch.zeros(256, device=self.dev, dtype=torch.float32), self.rope, n_pos)
            self.pe = model.abs_pos_encoding.pe.detach()[:, 0, :].to(device=self.dev, dtype=torch.float32).contiguous()
            if max(self.Lq, self.S) > self.pe.shape[0]:
                raise L.TcdiffError(f"PositionalEncoding holds {self.pe.shape[0]} positions, the sequence has {self.Lq} tokens "
                                    "(model/utils.py:12,29)")
        else:
            K.rope_table(model.rotary.freqs.detach().float().contiguous(), self.rope, n_pos)

    # ------------------------------------------------------------------------------------------------------------------
    # parameter groups and the flat gradient buffer
    # ------------------------------------------------------------------------------------------------------------------
    def _define(self):
        lins: Dict[str, _Lin] = {}

        def lin(key, w, b=None, split=0):
            lins[key] = _Lin(self, key, w, b, split)

        def wb(prefix):
            return [prefix + ".weight"], [prefix + ".bias"]

        lin("in", *wb("input_projection"))
        for i, k in ((0, "f1"), (2, "f2"), (4, "f3")):
            lin(k, *wb(f"relative_projection_layer.{i}"))
        lin("t1", *wb("time_mlp.1"))
        lin("tct", ["to_time_cond.0.weight", "to_time_tokens.0.weight"], ["to_time_cond.0.bias", "to_time_tokens.0.bias"])
        lin("c0", *wb("cond_projection.0"))
        lin("c2", *wb("cond_projection.2"))
        for i in range(2):
            q = f"cond_encoder.{i}."
            lin(f"e{i}.qkv", [q + "self_attn.in_proj_weight"], [q + "self_attn.in_proj_bias"], split=1024)
            lin(f"e{i}.o", *wb(q + "self_attn.out_proj"))
            lin(f"e{i}.l1", *wb(q + "linear1"))
            lin(f"e{i}.l2", *wb(q + "linear2"))
        lin("na1", *wb("non_attn_cond_projection.1"))
        lin("na3", *wb("non_attn_cond_projection.3"))
        st = "seqTransDecoder.stack."
        for l in range(self.NL):
            q = f"{st}{l}."
            lin(f"l{l}.qkv", [q + "self_attn.w_qs.weight", q + "self_attn.w_ks.weight", q + "self_attn.w_vs.weight"],
                split=1024)
            lin(f"l{l}.sfc", [q + "self_attn.fc.weight"])
            lin(f"l{l}.cq", [q + "multihead_attn.w_qs.weight"])
            lin(f"l{l}.cfc", [q + "multihead_attn.fc.weight"])
            lin(f"l{l}.ff1", *wb(q + "linear1"))
            lin(f"l{l}.ff2", *wb(q + "linear2"))
            lin(f"l{l}.l3", *wb(q + "linear3"))
        nk = 512 * self.NL
        lin("ckv", [f"{st}{l}.multihead_attn.w_ks.weight" for l in range(self.NL)] +
            [f"{st}{l}.multihead_attn.w_vs.weight" for l in range(self.NL)], split=nk)
        film = [f"{st}{l}.film{i}.block.1" for l in range(self.NL) for i in (1, 2, 3)]
        lin("film", [f + ".weight" for f in film], [f + ".bias" for f in film])
        lin("fin", *wb("final_layer"))
        self.lins = lins
        # flat gradient buffer: [stacked weights | stacked biases] of every fused linear, then the remaining live parameters
        used, off, self.slot = set(), 0, {}
        for lk in lins.values():
            for names in (lk.wnames, lk.bnames):                 # stacked parts are contiguous; every stack starts 16-byte aligned
                off = K.round_up(off, 4)
                for n in names:
                    self.slot[n] = (off, self.params[n].numel())
                    off += self.params[n].numel()
                    used.add(n)
        for n, p in self.params.items():
            if n not in used and not _is_dead(n):
                off = K.round_up(off, 4)                         # 16-byte aligned rows for the row kernels
                self.slot[n] = (off, p.numel())
                off += p.numel()
        self.n_grad = off
        self.order = list(self.params.keys())
        # flat range of every decoder layer's linears (contiguous by construction): all-reduced as soon as the layer's
        # backward has run, under the backward of the layers before it (tcdiff_amd/dist.py FlatGradientAllReducer)
        self.layer_range = []
        for l in range(self.NL):
            first, last = lins[f"l{l}.qkv"], lins[f"l{l}.l3"]
            lo = self.slot[first.wnames[0]][0]
            o, n = self.slot[(last.bnames or last.wnames)[-1]]
            self.layer_range.append((lo, o + n))
        # Data-parallel gradient averaging (dist.FlatGradientAllReducer) is OPT-IN and bound to a process group the trainer
        # names (enable_grad_sync): a job that initialised torch.distributed for sharded sampling only and then fine-tunes on
        # one rank would otherwise block in an all-reduce nobody else enters.  TCDIFF_GRAD_SYNC=1 restores
        # "whenever a default group with more than one rank exists" for launchers that cannot call it.
        self.grad_sync = None
        if _os.environ.get("TCDIFF_GRAD_SYNC", "0") == "1":
            self.enable_grad_sync()
        self._new_flat()

    def enable_grad_sync(self, group=None, on: bool = True):
        """Average gradients over `group` (None: the default process group) inside every backward from now on.  Every rank
        of the group must run the same number of backward passes per step (use drop_last / equal shards: an uneven last
        batch would pair collectives of different steps)."""
        from .dist import FlatGradientAllReducer
        self.grad_sync = FlatGradientAllReducer(group=group) if on else None
        return self.grad_sync

    def _new_flat(self):
        self.flat = torch.zeros(self.n_grad, device=self.dev, dtype=torch.float32)
        self.gW, self.gB = {}, {}
        for key, lk in self.lins.items():
            o0 = self.slot[lk.wnames[0]][0]
            self.gW[key] = self.flat[o0:o0 + lk.N * lk.K]
            if lk.bnames:
                b0 = self.slot[lk.bnames[0]][0]
                self.gB[key] = self.flat[b0:b0 + lk.N]

    def g(self, name: str) -> torch.Tensor:
        o, n = self.slot[name]
        return self.flat[o:o + n]

    def grad_views(self) -> List[Optional[torch.Tensor]]:
        out = []
        for n in self.order:
            if n in self.slot:
                o, cnt = self.slot[n]
                out.append(self.flat[o:o + cnt].view(self.params[n].shape))
            else:
                out.append(None)
        return out

    # True: hand every gradient back through autograd (AccumulateGrad nodes fire, so grad-accumulator hooks -- torch DDP /
    # FSDP reducers, accelerate-prepared wrappers -- see them) instead of assigning .grad directly; ~1.5 ms of host time per
    # step.  Hooks registered on the AccumulateGrad nodes are not visible from the Parameter, hence a switch and not a probe;
    # such a wrapper does its own gradient averaging, so `enable_grad_sync` must stay off with it.
    grads_through_autograd = False

    def deliver_grads(self) -> List[Optional[torch.Tensor]]:
        """What the autograd node returns for the parameters.  Normally nothing: every live parameter's .grad is SET here to a
        persistent view of the flat buffer (or added to, if the caller kept gradients from an earlier backward) -- the 310
        AccumulateGrad nodes and 310 fresh view tensors per step cost ~1.5 ms of host time between the end of the backward and
        the optimizer's launch, during which the GPU idled.  A parameter with tensor hooks gets its gradient through autograd."""
        if self.grads_through_autograd or \
                any(p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None) for p in self.params.values()):
            return [g if (g is not None and self.params[n].requires_grad) else None for n, g in zip(self.order, self.grad_views())]
        if self._gv is None or self._gv_flat is not self.flat:
            self._gv = [(self.params[n], self.flat[o:o + cnt].view(self.params[n].shape)) for n, (o, cnt) in self.slot.items()]
            self._gv_flat = self.flat
        for p, v in self._gv:
            if not p.requires_grad:           # frozen layer: torch would not give it a gradient either
                continue
            g = p.grad
            if g is None:
                p.grad = v
            else:
                g.add_(v)
        return [None] * len(self.order)

    def repack(self):
        """W / W^T operand packs of every linear from the fp32 masters, when a parameter changed (once per optimizer step):
        ONE tcdiff_cast_transpose_multi launch over a device table of all ~125 matrices (as separate launches 0.7 ms)."""
        ver = tuple(p._version for p in self.params.values())
        # (`p.data = other` moves a parameter to new storage WITHOUT changing its version counter -- load_state_dict(assign=True),
        # model.to(...) on the same device: the addresses are part of the key)
        ptrs = tuple(p.data_ptr() for p in self.params.values())
        if ver == self.packed_version and ptrs == self._ct_ptrs:
            return
        if self._ct is None or self._ct_ptrs != ptrs:              # parameters (re)allocated: the table holds raw pointers
            if self._ct is not None:
                # ... and so do the captured graphs (LayerNorm / null-embedding / bias pointers) and the cached .grad views
                # (stale Parameter objects): drop them, the next steps recapture
                for st in self._graphs.values():
                    if st.get("sv") is not None:
                        st["sv"].pop("graph", None)
                    st.clear()
                self._graphs = {}
                self._gv = self._gv_flat = None
            ents = [d for lk in self.lins.values() for d in lk.pack_entries()]
            self._ct, self._ct_ptrs, self._ct_keep = K.ct_table(self.dt, ents, self.dev), ptrs, ents
            sents = [e for lk in self.lins.values() for e in lk.stream_entries()]
            self._ws, self._ws_keep = (K.ws_table(sents, self.dev), sents) if sents else (None, None)
        K.cast_transpose_multi(self.dt, self._ct)
        if self._ws is not None:
            K.pack_row_streams(self._ws)
        for lk in self.lins.values():
            lk.pack_bias()
        self.packed_version = ver

    def P(self, name):           # fp32 master parameter (LayerNorm weights, null embeddings)
        return self.params[name].detach()

    # ------------------------------------------------------------------------------------------------------------------
    # small launch helpers
    # ------------------------------------------------------------------------------------------------------------------
    def z(self, *shape, dtype=None):
        return torch.zeros(*shape, device=self.dev, dtype=self.T if dtype is None else dtype)

    no_fuse = bool(int(_os.environ.get("TCDIFF_TRAIN_NOFUSE", "0")))     # A/B: activations as separate launches
    use_rows = bool(int(_os.environ.get("TCDIFF_TRAIN_ROWS", "1")))      # A/B: 0 = every linear through gemm_tile
    poison = False      # tests: fill every "empty" workspace with NaN, so that a kernel reading what nothing wrote shows up

    def pz(self, key, *shape, dtype=None):
        """Persistent zero-initialised buffer (head-major images, lse / delta rows): kernels only ever write the valid rows, so
        the zero padding written once stays valid and nothing is re-filled per step (345 fill launches, ~1 ms, at batch 32).
        One buffer per (key, shape): what the backward still needs (Q, K, V, lse of every layer) has its own key."""
        dtype = self.T if dtype is None else dtype
        k = (key, tuple(shape), dtype)
        t = self._pz.get(k)
        if t is None:
            t = torch.zeros(*shape, device=self.dev, dtype=dtype)
            self._pz[k] = t
        return t

    def e(self, *shape, dtype=None):
        if self.poison:
            return torch.full(shape, float("nan"), device=self.dev, dtype=self.T if dtype is None else dtype)
        return torch.empty(*shape, device=self.dev, dtype=self.T if dtype is None else dtype)

    def _row(self, **kw):
        a = K.row_args(seed=self.seed, drop_thr=self.thr, drop_scale=self.dscale, **kw)
        return a

    def row_fwd(self, **kw):
        K.row_fwd(self.dt, self._row(**kw))

    # From the third step with the same shapes on, forward and backward are REPLAYED: the first two steps run the Python schedule,
    # the third runs it once more under a stream capture (its private memory pool makes every workspace address permanent), which
    # yields one hipGraph each for the ~250 launches of the forward and the ~300 of the backward, and -- recorded on the way -- the
    # list of launcher calls (function + ctypes arguments) and torch copies.  Later steps copy the inputs into the captured input
    # tensors and replay.  Why: the Python schedule costs 10-13 ms of host time per step (ctypes, allocator, argument structs)
    # against 12.0 ms of device time at batch 32, so on a slow host the GPU waits for launches (13.6-14.8 ms per step measured on
    # such boxes, 11.9 ms at batch 4 where the device needs 9.5); replayed, the step is device-bound everywhere (12.7 / 9.6 ms).
    #   TCDIFF_TRAIN_GRAPH=2 (default) hipGraph replay; =1 the recorded command list as ordinary launches (same time, ~3 us of
    #   host per launch); =0 never capture.  Data-parallel runs replay the backward as one graph per decoder layer with that layer's
    #   gradient all-reduce launched between two replays (_capture_bwd_segments).
    use_graphs = int(_os.environ.get("TCDIFF_TRAIN_GRAPH", "2"))
    group_wgrad = not bool(int(_os.environ.get("TCDIFF_TRAIN_NOGROUP", "0")))      # A/B: one tcdiff_gemm_tn launch per linear

    def tt(self, fn):
        """a torch op of the schedule (a copy / fill on tensors that stay put): run it and, while recording, remember it"""
        fn()
        if L._rec is not None:
            L._rec.append((None, fn))

    @staticmethod
    def _replay(cmds):
        """the recorded launches again, on the current stream (every launcher's last argument is the stream)"""
        s = K.stream()
        for fn, args in cmds:
            if fn is None:
                args()
            else:
                rc = fn(*args[:-1], s)
                if rc != 0:
                    L.check(rc, "replayed launch")

    def queue_wgrad(self, prob):
        """prob = gemm_tn's arguments; the operand tensors stay referenced (alive) until the flush."""
        self._wq.append(prob)
        if len(self._wq) == L.TN_MAX_PROB:
            self.flush_wgrad()

    def flush_wgrad(self):
        if self._wq:
            K.gemm_tn_grouped(self.dt, self._wq)
            self._wq = []

    def row_bwd(self, *, M, L_, ln=None, nln=None, lin=None, **kw):
        """tcdiff_row_bwd with its parameter gradients added straight into the flat gradient buffer.
        ln / nln: parameter-name prefixes of the post / next LayerNorm; lin: key of the nn.Linear that produced z, whose bias
        gradient is the column sum of d_z (then that linear's bwd is called with bias_done=True)."""
        chunks = max(1, min(_ROWB_CHUNKS, L_ // 16))
        g = {}
        if ln:
            g.update(g_ln_g=self.g(ln + ".weight"), g_ln_b=self.g(ln + ".bias"))
        if nln:
            g.update(g_nln_g=self.g(nln + ".weight"), g_nln_b=self.g(nln + ".bias"))
        if lin:
            g.update(g_bias=self.gB[lin])
        K.row_bwd(self.dt, self._row(M=M, L=L_, chunks=chunks, **g, **kw))

    def act_fwd(self, a, rows, cols, act, site=None):
        y = self.e(rows, a.shape[1])
        thr, sc = (self.thr, self.dscale) if site is not None else (0, 1.0)
        K.act_drop(self.dt, a, a.shape[1], y, y.shape[1], rows, cols, act, self.seed, site or 0, thr, sc)
        return y

    def act_bwd(self, a, dy, rows, cols, act, site=None):
        da = self.e(rows, a.shape[1], dtype=a.dtype)
        thr, sc = (self.thr, self.dscale) if site is not None else (0, 1.0)
        K.act_drop_bwd(self.dt, a, a.shape[1], dy, dy.shape[1], da, rows, cols, act, self.seed, site or 0, thr, sc)
        return da

    # ------------------------------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------------------------------
    def forward(self, x, cond, times, keep, seed: Tuple[int, int], p_drop: float):
        """x (B, Lq, nf) fp32, cond (B, 2S(+1), Cd), times (B,) long, keep (B,) bool -> out (B, Lq, nf) fp32.
        model/model.py:548-624 with train-mode dropout (probability p_drop; 0 = the eval-mode arithmetic).
        Host part (dropout parameters, seed words to device memory, weight repack when a parameter changed, input conversion),
        then the launch schedule `_fwd` -- directly, or from the third call with the same shapes on as a replayed hipGraph
        (the ~250 launches of the forward cost ~5 ms of Python / ctypes time per step, the ~300 of the backward ~8 ms: more than
        their device time on a slow host)."""
        B = x.shape[0]
        self.thr, self.dscale = K.drop_params(p_drop)
        # bf16: the attention forwards also keep what O's 8 bits dropped, for the backward's delta (tcdiff_attention_train; A/B: =0)
        self.olo = self.dt == L.DT_BF16 and _os.environ.get("TCDIFF_TRAIN_OLO", "1") != "0"
        # two scalar fills, not a copy from a host tensor: a pageable host-to-device copy blocks the host until everything queued
        # before it has run -- one full device sync per step, after which the GPU idled while Python queued the next forward
        self.seed[0].fill_(seed[0] & 0x7FFFFFFF)
        self.seed[1].fill_(seed[1] & 0x7FFFFFFF)
        self.repack()
        x = x.reshape(B, self.Lq, self.nf).to(device=self.dev, dtype=torch.float32).contiguous()
        cond = cond.to(device=self.dev, dtype=torch.float32).contiguous()
        if cond.shape[1] // 2 != self.S:
            raise L.TcdiffError(f"cond length {cond.shape[1]} does not pair into seq_len={self.S} tokens (model/model.py:572-589)")
        times = times.to(device=self.dev, dtype=torch.int32).contiguous()
        keep = keep.to(device=self.dev, dtype=torch.uint8).contiguous()
        if not self.use_graphs or self.poison or self._graph_broken:
            return self._fwd(x, cond, times, keep)
        key = (B, cond.shape[1], float(p_drop))
        st = self._graphs.setdefault(key, {"n": 0, "fwd": None, "bwd": None})
        st["n"] += 1
        if st["fwd"] is None:
            if st["n"] < 3:
                return self._fwd(x, cond, times, keep)
            try:
                # a captured shape pins its whole activation set (5.7 GB at batch 32): at most three shapes stay captured
                live = [k for k, v in self._graphs.items() if v["fwd"] is not None]
                for k in live[:max(0, len(live) - 2)]:
                    old = self._graphs.pop(k)
                    if old.get("sv") is not None:
                        old["sv"].pop("graph", None)
                    if self.sv is old.get("sv"):
                        self.sv = None
                    old.clear()
                if self._pool is None:
                    self._pool = torch.cuda.graph_pool_handle()
                st["x"], st["cond"], st["t"], st["keep"] = x.clone(), cond.clone(), times.clone(), keep.clone()
                g = torch.cuda.CUDAGraph()
                L._rec = cmds = []
                try:
                    with torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
                        out = self._fwd(st["x"], st["cond"], st["t"], st["keep"])
                finally:
                    L._rec = None
                st["fwd"], st["out"], st["sv"], st["fwd_cmds"] = g, out, self.sv, cmds
                st["sv"]["graph"] = st
            except Exception as ex:                       # noqa: BLE001  -- never let the optimisation take the step down
                self._graph_broken = f"{type(ex).__name__}: {ex}"
                warnings.warn(f"tcdiff_amd: capturing the training step failed ({self._graph_broken}); continuing with the eager schedule")
                st["fwd"] = None
                return self._fwd(x, cond, times, keep)
        else:
            st["x"].copy_(x)
            st["cond"].copy_(cond)
            st["t"].copy_(times)
            st["keep"].copy_(keep)
        if self.use_graphs == 2:
            st["fwd"].replay()
        else:
            self._replay(st["fwd_cmds"])
        self.sv = st["sv"]
        self.sv["pending"] = True
        self._gen += 1
        self.sv["gen"] = self._gen
        return st["out"].view(B, self.Lq, self.nf)

    def _fwd(self, x, cond, times, keep_u8):
        """the forward's launches; x (B, Lq, nf) fp32, cond (B, clen, Cd) fp32, times int32, keep uint8: device, contiguous"""
        dt, lins = self.dt, self.lins
        B = x.shape[0]
        S, dn, Lq, nf, H, NL, Cd = self.S, self.dn, self.Lq, self.nf, self.H, self.NL, self.Cd
        M, Ms, Mc = B * Lq, B * S, B * (S + 2)
        self._gen += 1
        sv = dict(B=B, pending=True, gen=self._gen)
        P, e, z = self.P, self.e, self.z
        f32 = torch.float32
        x = x.view(M, nf)
        clen = cond.shape[1]
        sv["keep"] = keep_u8

        # ---- music branch: cond_projection, two encoder layers (model/model.py:572-583,211-245) -------------------------
        kc0 = lins["c0"].Kp
        cin = self.pz("cin", Ms, kc0)
        K.convert_pad(dt, cond, cin, Ms, 2 * Cd, kc0, rows_per_batch=S, batch_stride=clen * Cd, row_stride=2 * Cd)
        kc1 = lins["c2"].Kp
        c0a = self.pz("c0a", Ms, kc1)
        lins["c0"].fwd(cin, Ms, out=c0a, ldc=kc1)
        c1 = self.act_fwd(c0a, Ms, Cd, L.ACT_RELU)
        tok = e(Ms, 512, dtype=f32)
        lins["c2"].fwd(c1, Ms, out=tok, f32=True)
        if self.abs_pos:                      # cond_tokens = abs_pos_encoding(cond_tokens) (model/model.py:580): dropout site 9
            K.pos_drop(tok, Ms, 512, self.pe, S, self.seed, SITE_PE_COND, self.thr, self.dscale)
        sv.update(cin=cin, c0a=c0a, c1=c1, tok0=tok)
        mh, mrot = e(Ms, 512), e(Ms, 512)
        self.row_fwd(flags=L.ROWF_NEXT_LN | L.ROWF_STORE_H | L.ROWF_STORE_ROT, M=Ms, L=S, z=tok,
                     nln_g=P("cond_encoder.0.norm1.weight"), nln_b=P("cond_encoder.0.norm1.bias"), nln_eps=1e-5, hout=mh,
                     rout=mrot, rope=self.rope, pos_mod=S)
        enc = []
        for i in range(2):
            q = f"cond_encoder.{i}."
            s = dict(x_in=tok, h=mh, rot=mrot)
            Qi, Ki, Vi = (self.pz(f"e{i}.{n}", B, H, self.Lps, 64) for n in "QKV")
            lins[f"e{i}.qkv"].fwd(mrot, Ms, A2=mh, heads=dict(out=Qi, out_k=Ki, out_v=Vi, scale_q=0.125, Lseq=S,
                                                               Lp=self.Lps, n_q=512, n_k=512))
            O, lse = e(Ms, 512), self.pz(f"e{i}.lse", B, H, self.Lps, dtype=f32)
            Olo = e(Ms, 512) if self.olo else None      # what O's 8 bits dropped: the backward's delta reads O + O_lo (kernels.attention_train)
            K.attention_train(dt, Qi, Ki, Vi, O, lse, B, H, S, S, self.Lps, self.Lps, 512, self.seed, 4 * i + 0, self.thr,
                              self.dscale, O_lo=Olo)
            zo = e(Ms, 512, dtype=f32)
            lins[f"e{i}.o"].fwd(O, Ms, out=zo, f32=True)
            x2, h2 = e(Ms, 512, dtype=f32), e(Ms, 512)
            self.row_fwd(flags=L.ROWF_DROP_PRE | L.ROWF_RES | L.ROWF_STORE_X | L.ROWF_NEXT_LN | L.ROWF_STORE_H, M=Ms, L=S,
                         z=zo, xres=tok, xout=x2, nln_g=P(q + "norm2.weight"), nln_b=P(q + "norm2.bias"), nln_eps=1e-5,
                         hout=h2, site_pre=4 * i + 1)
            a, f = lins[f"e{i}.l1"].fwd(h2, Ms, out=e(Ms, 1024), act=(self.act, 4 * i + 2))
            zf = e(Ms, 512, dtype=f32)
            lins[f"e{i}.l2"].fwd(f, Ms, out=zf, f32=True)
            x3 = e(Ms, 512, dtype=f32)
            fl = L.ROWF_DROP_PRE | L.ROWF_RES | L.ROWF_STORE_X
            if i == 0:
                mh, mrot = e(Ms, 512), e(Ms, 512)
                self.row_fwd(flags=fl | L.ROWF_NEXT_LN | L.ROWF_STORE_H | L.ROWF_STORE_ROT, M=Ms, L=S, z=zf, xres=x2,
                             xout=x3, nln_g=P("cond_encoder.1.norm1.weight"), nln_b=P("cond_encoder.1.norm1.bias"),
                             nln_eps=1e-5, hout=mh, rout=mrot, rope=self.rope, pos_mod=S, site_pre=4 * i + 3)
            else:
                self.row_fwd(flags=fl, M=Ms, L=S, z=zf, xres=x2, xout=x3, site_pre=4 * i + 3)
            s.update(Q=Qi, K=Ki, V=Vi, O=O, Olo=Olo, lse=lse, zo=zo, x2=x2, h2=h2, a=a, f=f, zf=zf)
            enc.append(s)
            tok = x3
        sv["enc"] = enc
        # ---- null-conditioning select, pooled hidden (model/model.py:585-597,609-610) -----------------------------------
        sel_tok = e(B, S * 512, dtype=f32)
        K.select_rows(tok, P("null_cond_embed").reshape(-1), keep_u8, sel_tok, B, S * 512)
        pooled = e(B, 512, dtype=f32)
        K.mean_pool(sel_tok, pooled, B, S, 512)
        ph = e(B, 512)
        self.row_fwd(flags=L.ROWF_NEXT_LN | L.ROWF_STORE_H, M=B, L=1, z=pooled, nln_g=P("non_attn_cond_projection.0.weight"),
                     nln_b=P("non_attn_cond_projection.0.bias"), nln_eps=1e-5, hout=ph)
        pa = e(B, 512)
        lins["na1"].fwd(ph, B, out=pa)
        pb = self.act_fwd(pa, B, 512, L.ACT_SILU)
        hid = e(B, 512, dtype=f32)
        lins["na3"].fwd(pb, B, out=hid, f32=True)
        sel_hid = e(B, 512, dtype=f32)
        K.select_rows(hid, P("null_cond_hidden").reshape(-1), keep_u8, sel_hid, B, 512)
        sv.update(pooled=pooled, ph=ph, pa=pa, pb=pb)
        # ---- time path (model/model.py:601-612) and the FiLM generators (:154-168) ----------------------------------------
        emb = e(B, 512)
        K.sinusoidal(dt, times, B, self.sin_freq, emb)
        ta = e(B, 2048)
        lins["t1"].fwd(emb, B, out=ta)
        th = self.act_fwd(ta, B, 2048, L.ACT_MISH)
        tcat = e(B, 1536, dtype=f32)                      # [to_time_cond | to_time_tokens]
        lins["tct"].fwd(th, B, out=tcat, f32=True)
        pre = e(B, 512, dtype=f32)
        K.add_rows(tcat, 1536, sel_hid, 512, pre, 512, B, 512)            # t += cond_hidden (:612)
        fin = self.act_fwd(pre, B, 512, L.ACT_MISH)
        nfilm = NL * 3 * 1024
        film = e(B, nfilm, dtype=f32)
        lins["film"].fwd(fin, B, out=film, f32=True)
        sv.update(emb=emb, ta=ta, th=th, pre=pre, fin=fin, film=film)
        # ---- memory = norm_cond(cat(tokens, time tokens)) and every layer's cross-attention K / V (:615-616,386-396) --------
        memin = e(B, S + 2, 512, dtype=f32)                # cat(tokens, the two time tokens): copies only
        self.tt(lambda: memin[:, :S].copy_(sel_tok.view(B, S, 512)))
        self.tt(lambda: memin[:, S:].copy_(tcat[:, 512:].unflatten(1, (2, 512))))
        mem_h, mem_rot = e(Mc, 512), e(Mc, 512)
        self.row_fwd(flags=L.ROWF_NEXT_LN | L.ROWF_STORE_H | L.ROWF_STORE_ROT, M=Mc, L=S + 2, z=memin.view(Mc, 512),
                     nln_g=P("norm_cond.weight"), nln_b=P("norm_cond.bias"), nln_eps=1e-5, hout=mem_h, rout=mem_rot,
                     rope=self.rope, pos_mod=S + 2)
        Kc, Vc = self.pz("Kc", NL, B, H, self.Lpc, 64), self.pz("Vc", NL, B, H, self.Lpc, 64)
        nk = 512 * NL
        # all layers' w_ks(rot(memory)) and w_vs(memory) as ONE GEMM: columns [0, nk) take rot(memory), [nk, 2 nk) memory; the
        # epilogue scatters layer l's eight heads into the image Kc[l] / Vc[l]
        lins["ckv"].fwd(mem_rot, Mc, A2=mem_h, heads=dict(out=None, out_k=Kc, out_v=Vc, scale_q=1.0, Lseq=S + 2, Lp=self.Lpc, n_q=0,
                                                          n_k=nk, hgroup=H, hgroup_stride=B * H * self.Lpc * 64))
        sv.update(memin=memin, mem_h=mem_h, mem_rot=mem_rot, Kc=Kc, Vc=Vc)
        # ---- motion: input projection + fusion projection (model/model.py:560-561) --------------------------------------------
        xin = self.pz("xin", M, lins["in"].Kp)
        K.convert_pad(dt, x, xin, M, nf, lins["in"].Kp)
        xp = e(M, 512)
        lins["in"].fwd(xin, M, out=xp)
        xpf = xp.view(Ms, 512 * dn)
        f1a, f1 = lins["f1"].fwd(xpf, Ms, out=e(Ms, 1024), act=(L.ACT_RELU, None))
        f2a, f2 = lins["f2"].fwd(f1, Ms, out=e(Ms, 1024), act=(L.ACT_RELU, None))
        xs = e(Ms, 512 * dn, dtype=f32)
        lins["f3"].fwd(f2, Ms, out=xs, f32=True)
        xs = xs.view(M, 512)
        if self.abs_pos:                      # x = abs_pos_encoding(x) (model/model.py:564): dropout site 8
            K.pos_drop(xs, M, 512, self.pe, Lq, self.seed, SITE_PE_X, self.thr, self.dscale)
        st = "seqTransDecoder.stack."
        h1, r1 = e(M, 512), e(M, 512)
        self.row_fwd(flags=L.ROWF_NEXT_LN | L.ROWF_STORE_H | L.ROWF_STORE_ROT, M=M, L=Lq, z=xs, nln_g=P(st + "0.norm1.weight"),
                     nln_b=P(st + "0.norm1.bias"), nln_eps=1e-5, hout=h1, rout=r1, rope=self.rope, pos_mod=Lq)
        sv.update(xin=xin, xpf=xpf, f1a=f1a, f1=f1, f2a=f2a, f2=f2, xs=xs)
        # ---- decoder layers (model/model.py:308-344) ---------------------------------------------------------------------------
        layers = []
        xcur = xs
        Lp, Lpc = self.Lp, self.Lpc
        for l in range(NL):
            q = f"{st}{l}."
            s = dict(x=xcur, h1=h1, r1=r1)
            sd = 16 + 8 * l
            Q, Kk, V = (self.pz(f"l{l}.{n}", B, H, Lp, 64) for n in "QKV")
            lins[f"l{l}.qkv"].fwd(r1, M, A2=h1, heads=dict(out=Q, out_k=Kk, out_v=V, scale_q=0.125, Lseq=Lq, Lp=Lp, n_q=512,
                                                            n_k=512))
            O, lse = e(M, 512), self.pz(f"l{l}.lse", B, H, Lp, dtype=f32)
            Olo = e(M, 512) if self.olo else None
            K.attention_train(dt, Q, Kk, V, O, lse, B, H, Lq, Lq, Lp, Lp, 512, self.seed, sd + 0, self.thr, self.dscale, O_lo=Olo)
            z1 = e(M, 512, dtype=f32)
            lins[f"l{l}.sfc"].fwd(O, M, out=z1, f32=True)
            x2, r2 = e(M, 512, dtype=f32), e(M, 512)
            blk = L.ROWF_DROP_PRE | L.ROWF_LN_POST | L.ROWF_DROP_POST | L.ROWF_FILM | L.ROWF_RES | L.ROWF_STORE_X | L.ROWF_NEXT_LN
            self.row_fwd(flags=blk | L.ROWF_STORE_ROT, M=M, L=Lq, z=z1, ln_g=P(q + "self_attn.layer_norm.weight"),
                         ln_b=P(q + "self_attn.layer_norm.bias"), ln_eps=1e-6, film=film[:, (3 * l) * 1024:], film_ld=nfilm,
                         xres=xcur, xout=x2, nln_g=P(q + "norm2.weight"), nln_b=P(q + "norm2.bias"), nln_eps=1e-5, rout=r2,
                         rope=self.rope, pos_mod=Lq, site_pre=sd + 1, site_post=sd + 2)
            Qc = self.pz(f"l{l}.Qc", B, H, Lp, 64)
            lins[f"l{l}.cq"].fwd(r2, M, heads=dict(out=Qc, out_k=None, out_v=None, scale_q=0.125, Lseq=Lq, Lp=Lp, n_q=512, n_k=0))
            Oc, lsec = e(M, 512), self.pz(f"l{l}.lsec", B, H, Lp, dtype=f32)
            Oclo = e(M, 512) if self.olo else None
            K.attention_train(dt, Qc, Kc[l], Vc[l], Oc, lsec, B, H, Lq, S + 2, Lp, Lpc, 512, self.seed, sd + 3, self.thr,
                              self.dscale, O_lo=Oclo)
            z2 = e(M, 512, dtype=f32)
            lins[f"l{l}.cfc"].fwd(Oc, M, out=z2, f32=True)
            x3, h3 = e(M, 512, dtype=f32), e(M, 512)
            self.row_fwd(flags=blk | L.ROWF_STORE_H, M=M, L=Lq, z=z2, ln_g=P(q + "multihead_attn.layer_norm.weight"),
                         ln_b=P(q + "multihead_attn.layer_norm.bias"), ln_eps=1e-6, film=film[:, (3 * l + 1) * 1024:],
                         film_ld=nfilm, xres=x2, xout=x3, nln_g=P(q + "norm3.weight"), nln_b=P(q + "norm3.bias"), nln_eps=1e-5,
                         hout=h3, site_pre=sd + 4, site_post=sd + 5)
            a, f = lins[f"l{l}.ff1"].fwd(h3, M, out=e(M, 1024), act=(self.act, sd + 6))
            z3 = e(M, 512, dtype=f32)
            lins[f"l{l}.ff2"].fwd(f, M, out=z3, f32=True)
            h4 = e(M, 512)
            # x4 itself is not kept: the layer output linear3(norm4(x4)) has no residual (model/model.py:344)
            self.row_fwd(flags=L.ROWF_DROP_PRE | L.ROWF_FILM | L.ROWF_RES | L.ROWF_NEXT_LN | L.ROWF_STORE_H, M=M, L=Lq, z=z3,
                         film=film[:, (3 * l + 2) * 1024:], film_ld=nfilm, xres=x3, nln_g=P(q + "norm4.weight"),
                         nln_b=P(q + "norm4.bias"), nln_eps=1e-5, hout=h4, site_pre=sd + 7)
            z4 = e(M, 512, dtype=f32)
            lins[f"l{l}.l3"].fwd(h4, M, out=z4, f32=True)
            s.update(Q=Q, K=Kk, V=V, O=O, Olo=Olo, Oclo=Oclo, lse=lse, z1=z1, x2=x2, r2=r2, Qc=Qc, Oc=Oc, lsec=lsec, z2=z2, x3=x3, h3=h3, a=a, f=f,
                     z3=z3, h4=h4, z4=z4)
            layers.append(s)
            if l + 1 < NL:
                h1, r1 = e(M, 512), e(M, 512)
                self.row_fwd(flags=L.ROWF_NEXT_LN | L.ROWF_STORE_H | L.ROWF_STORE_ROT, M=M, L=Lq, z=z4,
                             nln_g=P(f"{st}{l + 1}.norm1.weight"), nln_b=P(f"{st}{l + 1}.norm1.bias"), nln_eps=1e-5, hout=h1,
                             rout=r1, rope=self.rope, pos_mod=Lq)
                xcur = z4
        sv["layers"] = layers
        hT = e(M, 512)
        K.cast_transpose(dt, z4, M, 512, 512, dst=hT, ld_dst=512, cols_pad=512)
        out = e(M, nf, dtype=f32)
        lins["fin"].fwd(hT, M, out=out, f32=True, ldc=nf)
        sv["hT"] = hT
        self.sv = sv
        return out.view(B, Lq, nf)

    # ------------------------------------------------------------------------------------------------------------------
    # backward
    # ------------------------------------------------------------------------------------------------------------------
    def backward(self, d_out: torch.Tensor, gen: Optional[int] = None) -> List[Optional[torch.Tensor]]:
        sv = self.sv
        if sv is None or not sv.get("pending"):
            raise L.TcdiffError("backward without a matching train-mode forward (one forward may be outstanding per model)")
        if gen is not None and sv.get("gen") != gen:
            # fwd(A), fwd(B), loss_A.backward(): the engine holds B's activations and dropout seed -- refuse instead of
            # silently differentiating A through them
            raise L.TcdiffError("backward of a train-mode forward that is no longer the most recent one: the engine keeps "
                                "the activations of ONE forward per model (run forward -> backward in pairs)")
        sv["pending"] = False
        # a parameter's .grad may still alias the previous step's flat buffer (the caller accumulates across calls
        # instead of zero_grad): never overwrite gradients somebody still holds
        p_first = self.params[next(iter(self.slot))]
        aliased = p_first.grad is not None and \
            p_first.grad.untyped_storage().data_ptr() == self.flat.untyped_storage().data_ptr()
        if aliased:
            self._new_flat()
            for st in self._graphs.values():              # captured backwards write the buffer that was just given away
                st["bwd"] = None
                st["bwd_segs"] = None
        sync = self.grad_sync if (self.grad_sync is not None and self.grad_sync.active()) else None
        if self.grad_sync is None and not getattr(self, "_warned_no_sync", False):
            # the reference trains under DDP (TCDiff.py:51-52,232); here averaging is opt-in -- say so ONCE when a multi-rank job
            # runs a backward without it, instead of training unsynchronised replicas silently
            self._warned_no_sync = True
            import torch.distributed as _td
            if _td.is_available() and _td.is_initialized() and _td.get_world_size() > 1:
                warnings.warn("tcdiff_amd: torch.distributed has %d ranks but gradient averaging is off: call "
                              "model.train_engine().enable_grad_sync(group) or set TCDIFF_GRAD_SYNC=1 (a DDP / accelerate wrapper "
                              "that averages .grad itself needs neither)" % _td.get_world_size(), stacklevel=2)
        B = sv["B"]
        d_out = d_out.reshape(B * self.Lq, self.nf).to(dtype=torch.float32).contiguous()
        st = sv.get("graph")
        if st is not None and sync is not None and self.use_graphs == 2 and not self.poison and not self._graph_broken:
            # data-parallel AND replayed: the backward is captured in SEGMENTS that end where a range of the flat gradient
            # buffer is complete (after every decoder layer); replaying segment k, then launching range k's all-reduce
            # (asynchronously, on the collective's own stream behind an event of this one), then segment k + 1 keeps both the
            # replay speed and the overlap of the eager schedule (VERDICT r3 #4c: config 5's 8-GPU form ran eagerly)
            if st.get("bwd_segs") is None:
                try:
                    st["dout"] = d_out.clone()
                    st["bwd_segs"] = self._capture_bwd_segments(sv, st["dout"])
                except Exception as ex:                   # noqa: BLE001
                    self._graph_broken = f"{type(ex).__name__}: {ex}"
                    warnings.warn(f"tcdiff_amd: capturing the data-parallel backward failed ({self._graph_broken}); continuing "
                                  f"with the eager schedule")
                    st["bwd_segs"] = None
                    self._wq = []
                    self._bwd(sv, d_out, sync, zero=True)
                    return self.deliver_grads()
            else:
                st["dout"].copy_(d_out)
            for g, ranges in st["bwd_segs"]:
                g.replay()
                for lo, hi in ranges:
                    sync.ready(self.flat, lo, hi)
            sync.finish()
            self.sv = None
            return self.deliver_grads()
        if st is None or sync is not None or self.poison or self._graph_broken:
            self._bwd(sv, d_out, sync, zero=not aliased)
            return self.deliver_grads()
        if st["bwd"] is None:
            try:
                st["dout"] = d_out.clone()
                g = torch.cuda.CUDAGraph()
                L._rec = cmds = []
                try:
                    with torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
                        self._bwd(sv, st["dout"], None, zero=True)
                finally:
                    L._rec = None
                st["bwd"], st["bwd_cmds"] = g, cmds
            except Exception as ex:                       # noqa: BLE001
                self._graph_broken = f"{type(ex).__name__}: {ex}"
                warnings.warn(f"tcdiff_amd: capturing the training step failed ({self._graph_broken}); continuing with the eager schedule")
                st["bwd"] = None
                self._wq = []
                self._bwd(sv, d_out, None, zero=True)
                return self.deliver_grads()
        else:
            st["dout"].copy_(d_out)
        if self.use_graphs == 2:
            st["bwd"].replay()
        else:
            self._replay(st["bwd_cmds"])
        self.sv = None
        return self.deliver_grads()

    def _capture_bwd_segments(self, sv, dout):
        """Capture the backward as a list of (hipGraph, [(lo, hi) ranges of the flat gradient buffer complete behind it]): the
        schedule runs ONCE under stream capture with a stand-in for the gradient averager whose ready() ends the current graph
        and begins the next one; nothing executes here, the caller replays."""
        segs, eng = [], self

        class _Cut:
            g, n0 = None, 0

            def begin(c):
                c.g = torch.cuda.CUDAGraph()
                c.g.capture_begin(pool=eng._pool, capture_error_mode="thread_local")
                c.n0 = len(L._rec)

            def end(c):
                c.g.capture_end()
                if len(L._rec) == c.n0:
                    raise L.TcdiffError("empty segment in the captured backward")
                segs.append([c.g, []])
                c.g = None

            def ready(c, flat, lo, hi, more=True):
                if c.g is not None:
                    c.end()
                segs[-1][1].append((lo, hi))
                if more:
                    c.begin()

            def finish(c):
                if c.g is not None:
                    c.end()
        cut = _Cut()
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream())
        L._rec = []
        try:
            with torch.cuda.stream(side):
                cut.begin()
                self._bwd(sv, dout, cut, zero=True)
                cut.finish()
        except BaseException:
            # a failure in the middle of a segment (kernel error, allocation, unsupported op) must not leave `side` in
            # capture: the caller falls back to the eager schedule on this thread, and a live thread_local capture would
            # refuse its first allocation or sync
            if cut.g is not None:
                try:
                    cut.g.capture_end()
                except Exception:
                    pass
                cut.g = None
            segs.clear()
            raise
        finally:
            L._rec = None
        torch.cuda.current_stream().wait_stream(side)
        return segs

    def _bwd(self, sv, d_out, sync, zero=True):
        """the backward's launches; d_out (M, nf) fp32 contiguous; gradients accumulate into self.flat (zeroed first)"""
        dt, lins = self.dt, self.lins
        B = sv["B"]
        S, dn, Lq, nf, H, NL = self.S, self.dn, self.Lq, self.nf, self.H, self.NL
        M, Ms, Mc = B * Lq, B * S, B * (S + 2)
        Lp, Lpc, Lps = self.Lp, self.Lpc, self.Lps
        e, z, P = self.e, self.z, self.P
        f32 = torch.float32
        st = "seqTransDecoder.stack."
        if zero:
            self.tt(lambda: self.flat.zero_())
        nfilm = NL * 3 * 1024
        dfilm = e(B, nfilm, dtype=f32)
        self.tt(lambda: dfilm.zero_())
        nk = 512 * NL
        dKV = e(Mc, 2 * nk)                              # [dK of layer 0..NL-1 | dV of layer 0..NL-1], token-major

        # ---- final layer --------------------------------------------------------------------------------------------------
        dhT = e(M, 512)
        lins["fin"].bwd(d_out, nf, M, [sv["hT"]], [("T", dhT, 512)])
        g_x, g_h, g_r = None, None, None                  # gradients reaching the NEXT layer's inputs (x fp32, h1 / r1 T)
        dz4 = dhT
        for l in reversed(range(NL)):
            s = sv["layers"][l]
            q = f"{st}{l}."
            sd = 16 + 8 * l
            if l + 1 < NL:                                # x' = linear3(.) feeds the next layer: residual + norm1 + rotary
                dz4 = e(M, 512)
                self.row_bwd(M=M, L_=Lq, nln=f"{st}{l + 1}.norm1", lin=f"l{l}.l3",
                             flags=L.ROWF_NEXT_LN | L.ROWF_STORE_H | L.ROWF_STORE_ROT,
                             z=s["z4"], nln_g=P(f"{st}{l + 1}.norm1.weight"), nln_b=P(f"{st}{l + 1}.norm1.bias"), nln_eps=1e-5,
                             rope=self.rope, pos_mod=Lq, d_xn=g_x, d_h=g_h, d_rot=g_r, d_z=dz4)
            dh4 = e(M, 512)
            lins[f"l{l}.l3"].bwd(dz4, 512, M, [s["h4"]], [("T", dh4, 512)], bias_done=l + 1 < NL)
            # feed-forward block
            dz3, gx3 = e(M, 512), e(M, 512, dtype=f32)
            self.row_bwd(M=M, L_=Lq, nln=q + "norm4", lin=f"l{l}.ff2",
                         flags=L.ROWF_DROP_PRE | L.ROWF_FILM | L.ROWF_RES | L.ROWF_NEXT_LN | L.ROWF_STORE_H,
                         z=s["z3"], film=sv["film"][:, (3 * l + 2) * 1024:], film_ld=nfilm, xres=s["x3"],
                         nln_g=P(q + "norm4.weight"), nln_b=P(q + "norm4.bias"), nln_eps=1e-5, site_pre=sd + 7, d_h=dh4, d_z=dz3,
                         d_xres=gx3, d_film=dfilm[:, (3 * l + 2) * 1024:], dfilm_ld=nfilm)
            da = e(M, 1024)
            lins[f"l{l}.ff2"].bwd(dz3, 512, M, [s["f"]], [("ACT", da, 1024, s["a"], self.act, sd + 6)], bias_done=True)
            dh3 = e(M, 512)
            lins[f"l{l}.ff1"].bwd(da, 1024, M, [s["h3"]], [("T", dh3, 512)])
            # cross-attention block
            blk = L.ROWF_DROP_PRE | L.ROWF_LN_POST | L.ROWF_DROP_POST | L.ROWF_FILM | L.ROWF_RES | L.ROWF_NEXT_LN
            dz2, gx2 = e(M, 512), e(M, 512, dtype=f32)
            self.row_bwd(M=M, L_=Lq, ln=q + "multihead_attn.layer_norm", nln=q + "norm3", flags=blk | L.ROWF_STORE_H, z=s["z2"],
                         ln_g=P(q + "multihead_attn.layer_norm.weight"), ln_b=P(q + "multihead_attn.layer_norm.bias"),
                         ln_eps=1e-6, film=sv["film"][:, (3 * l + 1) * 1024:], film_ld=nfilm, xres=s["x2"],
                         nln_g=P(q + "norm3.weight"), nln_b=P(q + "norm3.bias"), nln_eps=1e-5, site_pre=sd + 4, site_post=sd + 5,
                         d_xn=gx3, d_h=dh3, d_z=dz2, d_xres=gx2, d_film=dfilm[:, (3 * l + 1) * 1024:], dfilm_ld=nfilm)
            dOc = self.pz("dO", B, H, Lp, 64)
            lins[f"l{l}.cfc"].bwd(dz2, 512, M, [s["Oc"]], [("HEADS", dict(out=dOc, out_k=None, out_v=None, scale_q=1.0, Lseq=Lq,
                                                                             Lp=Lp, n_q=512, n_k=0))])
            dQc, delta = e(M, 512), self.pz("delta", B, H, Lp, dtype=f32)
            K.attention_bwd(dt, s["Qc"], sv["Kc"][l], sv["Vc"][l], s["Oc"], dOc, s["lsec"], delta, dQc, 512,
                            dKV.view(-1)[512 * l:], dKV.view(-1)[nk + 512 * l:], 2 * nk, B, H, Lq, S + 2, Lp, Lpc, 512, 0.125,
                            self.seed, sd + 3, self.thr, self.dscale, O_lo=s["Oclo"])
            dr2 = e(M, 512)
            lins[f"l{l}.cq"].bwd(dQc, 512, M, [s["r2"]], [("T", dr2, 512)])
            # self-attention block
            dz1, gx1 = e(M, 512), e(M, 512, dtype=f32)
            self.row_bwd(M=M, L_=Lq, ln=q + "self_attn.layer_norm", nln=q + "norm2", flags=blk | L.ROWF_STORE_ROT, z=s["z1"],
                         ln_g=P(q + "self_attn.layer_norm.weight"), ln_b=P(q + "self_attn.layer_norm.bias"), ln_eps=1e-6,
                         film=sv["film"][:, (3 * l) * 1024:], film_ld=nfilm, xres=s["x"], nln_g=P(q + "norm2.weight"),
                         nln_b=P(q + "norm2.bias"), nln_eps=1e-5, rope=self.rope, pos_mod=Lq, site_pre=sd + 1, site_post=sd + 2,
                         d_xn=gx2, d_rot=dr2, d_z=dz1, d_xres=gx1, d_film=dfilm[:, (3 * l) * 1024:], dfilm_ld=nfilm)
            dO = self.pz("dO", B, H, Lp, 64)
            lins[f"l{l}.sfc"].bwd(dz1, 512, M, [s["O"]], [("HEADS", dict(out=dO, out_k=None, out_v=None, scale_q=1.0, Lseq=Lq,
                                                                           Lp=Lp, n_q=512, n_k=0))])
            dQKV, delta = e(M, 1536), self.pz("delta", B, H, Lp, dtype=f32)
            K.attention_bwd(dt, s["Q"], s["K"], s["V"], s["O"], dO, s["lse"], delta, dQKV, 1536, dQKV.view(-1)[512:],
                            dQKV.view(-1)[1024:], 1536, B, H, Lq, Lq, Lp, Lp, 512, 0.125, self.seed, sd + 0, self.thr,
                            self.dscale, O_lo=s["Olo"])
            g_r, g_h = e(M, 512), e(M, 512)
            lins[f"l{l}.qkv"].bwd(dQKV, 1536, M, [s["r1"], s["h1"]], [("T", g_r, 512), ("T", g_h, 512)])
            g_x = gx1
            self.flush_wgrad()                            # the layer's seven weight gradients: one evenly split launch
            if sync is not None:                          # this layer's linears are complete: average them across the ranks now
                sync.ready(self.flat, *self.layer_range[l])
        # ---- front: layer 0's norm1 / rotary on the fusion projection's output, then the fusion MLP -----------------------------
        dxs = e(M, 512)
        self.row_bwd(M=M, L_=Lq, nln=st + "0.norm1", flags=L.ROWF_NEXT_LN | L.ROWF_STORE_H | L.ROWF_STORE_ROT, z=sv["xs"],
                     nln_g=P(st + "0.norm1.weight"), nln_b=P(st + "0.norm1.bias"), nln_eps=1e-5, rope=self.rope, pos_mod=Lq,
                     d_xn=g_x, d_h=g_h, d_rot=g_r, d_z=dxs)
        if self.abs_pos and self.thr:         # through PositionalEncoding's dropout (the mask of site 8 again)
            dxs = self.act_bwd(dxs, dxs, M, 512, L.ACT_NONE, site=SITE_PE_X)
        df2a, df1a = e(Ms, 1024), e(Ms, 1024)
        lins["f3"].bwd(dxs.view(Ms, 512 * dn), 512 * dn, Ms, [sv["f2"]], [("ACT", df2a, 1024, sv["f2a"], L.ACT_RELU, None)])
        lins["f2"].bwd(df2a, 1024, Ms, [sv["f1"]], [("ACT", df1a, 1024, sv["f1a"], L.ACT_RELU, None)])
        dxp = e(Ms, 512 * dn)
        lins["f1"].bwd(df1a, 1024, Ms, [sv["xpf"]], [("T", dxp, 512 * dn)])
        lins["in"].bwd(dxp.view(M, 512), 512, M, [sv["xin"]], [None])
        # ---- cross-attention K / V of all layers -> memory rows ---------------------------------------------------------------------
        d_mrot, d_mh = e(Mc, 512), e(Mc, 512)
        lins["ckv"].bwd(dKV, 2 * nk, Mc, [sv["mem_rot"], sv["mem_h"]], [("T", d_mrot, 512), ("T", d_mh, 512)])
        d_memin = e(Mc, 512, dtype=f32)
        self.row_bwd(M=Mc, L_=S + 2, nln="norm_cond", flags=L.ROWF_NEXT_LN | L.ROWF_STORE_H | L.ROWF_STORE_ROT,
                     z=sv["memin"].view(Mc, 512), nln_g=P("norm_cond.weight"), nln_b=P("norm_cond.bias"), nln_eps=1e-5,
                     rope=self.rope, pos_mod=S + 2, d_h=d_mh, d_rot=d_mrot, d_z=d_memin, dz_f32=1)
        d_memin = d_memin.view(B, S + 2, 512)
        g_tok_mem = e(B, S, 512, dtype=f32)                # copies only
        self.tt(lambda: g_tok_mem.copy_(d_memin[:, :S]))
        # ---- FiLM generators and the time path ------------------------------------------------------------------------------------------
        dfin = e(B, 512)
        lins["film"].bwd(dfilm, nfilm, B, [sv["fin"]], [("T", dfin, 512)])
        d_pre = self.act_bwd(sv["pre"], dfin, B, 512, L.ACT_MISH)          # fp32: = d t_base = d cond_hidden
        d_tcat = e(B, 1536, dtype=f32)
        self.tt(lambda: d_tcat[:, :512].copy_(d_pre))
        self.tt(lambda: d_tcat[:, 512:].unflatten(1, (2, 512)).copy_(d_memin[:, S:]))
        dth = e(B, 2048)
        lins["tct"].bwd(d_tcat, 1536, B, [sv["th"]], [("T", dth, 2048)])
        dta = self.act_bwd(sv["ta"], dth, B, 2048, L.ACT_MISH)
        lins["t1"].bwd(dta, 2048, B, [sv["emb"]], [None])
        # ---- pooled hidden, null-conditioning selects ---------------------------------------------------------------------------------------
        d_hid = e(B, 512, dtype=f32)
        K.select_rows_bwd(d_pre, sv["keep"], d_hid, self.g("null_cond_hidden"), B, 512)
        dpb = e(B, 512)
        lins["na3"].bwd(d_hid, 512, B, [sv["pb"]], [("T", dpb, 512)])
        dpa = self.act_bwd(sv["pa"], dpb, B, 512, L.ACT_SILU)
        dph = e(B, 512)
        lins["na1"].bwd(dpa, 512, B, [sv["ph"]], [("T", dph, 512)])
        d_pool = e(B, 512, dtype=f32)
        self.row_bwd(M=B, L_=1, nln="non_attn_cond_projection.0", flags=L.ROWF_NEXT_LN | L.ROWF_STORE_H, z=sv["pooled"],
                     nln_g=P("non_attn_cond_projection.0.weight"), nln_b=P("non_attn_cond_projection.0.bias"), nln_eps=1e-5,
                     d_h=dph, d_z=d_pool, dz_f32=1)
        d_sel = e(B, S * 512, dtype=f32)
        K.pool_bwd(g_tok_mem, d_pool, d_sel, B, S, 512)
        g_tok = e(Ms, 512, dtype=f32)
        K.select_rows_bwd(d_sel, sv["keep"], g_tok, self.g("null_cond_embed"), B, S * 512)
        # ---- music encoder layers -------------------------------------------------------------------------------------------------------------
        g_h, g_r = None, None
        for i in reversed(range(2)):
            s = sv["enc"][i]
            q = f"cond_encoder.{i}."
            dzf, gx2 = e(Ms, 512), e(Ms, 512, dtype=f32)
            fl = L.ROWF_DROP_PRE | L.ROWF_RES
            if i == 0:
                self.row_bwd(M=Ms, L_=S, nln="cond_encoder.1.norm1", lin=f"e{i}.l2",
                             flags=fl | L.ROWF_NEXT_LN | L.ROWF_STORE_H | L.ROWF_STORE_ROT,
                             z=s["zf"], xres=s["x2"], nln_g=P("cond_encoder.1.norm1.weight"), nln_b=P("cond_encoder.1.norm1.bias"),
                             nln_eps=1e-5, rope=self.rope, pos_mod=S, site_pre=4 * i + 3, d_xn=g_tok, d_h=g_h, d_rot=g_r, d_z=dzf,
                             d_xres=gx2)
            else:
                self.row_bwd(M=Ms, L_=S, lin=f"e{i}.l2", flags=fl, z=s["zf"], xres=s["x2"], site_pre=4 * i + 3, d_xn=g_tok,
                             d_z=dzf, d_xres=gx2)
            da = e(Ms, 1024)
            lins[f"e{i}.l2"].bwd(dzf, 512, Ms, [s["f"]], [("ACT", da, 1024, s["a"], self.act, 4 * i + 2)], bias_done=True)
            dh2 = e(Ms, 512)
            lins[f"e{i}.l1"].bwd(da, 1024, Ms, [s["h2"]], [("T", dh2, 512)])
            dzo, gx1 = e(Ms, 512), e(Ms, 512, dtype=f32)
            self.row_bwd(M=Ms, L_=S, nln=q + "norm2", lin=f"e{i}.o", flags=fl | L.ROWF_NEXT_LN | L.ROWF_STORE_H, z=s["zo"],
                         xres=s["x_in"],
                         nln_g=P(q + "norm2.weight"), nln_b=P(q + "norm2.bias"), nln_eps=1e-5, site_pre=4 * i + 1, d_xn=gx2,
                         d_h=dh2, d_z=dzo, d_xres=gx1)
            dO = self.pz("dOe", B, H, Lps, 64)
            lins[f"e{i}.o"].bwd(dzo, 512, Ms, [s["O"]], [("HEADS", dict(out=dO, out_k=None, out_v=None, scale_q=1.0, Lseq=S,
                                                                        Lp=Lps, n_q=512, n_k=0))], bias_done=True)
            dQKV, delta = e(Ms, 1536), self.pz("deltae", B, H, Lps, dtype=f32)
            K.attention_bwd(dt, s["Q"], s["K"], s["V"], s["O"], dO, s["lse"], delta, dQKV, 1536, dQKV.view(-1)[512:],
                            dQKV.view(-1)[1024:], 1536, B, H, S, S, Lps, Lps, 512, 0.125, self.seed, 4 * i + 0, self.thr,
                            self.dscale, O_lo=s["Olo"])
            g_r, g_h = e(Ms, 512), e(Ms, 512)
            lins[f"e{i}.qkv"].bwd(dQKV, 1536, Ms, [s["rot"], s["h"]], [("T", g_r, 512), ("T", g_h, 512)])
            g_tok = gx1
        dtok0 = e(Ms, 512)
        pe_drop = self.abs_pos and self.thr     # (then c2's bias gradient is taken behind the mask, by the linear's own backward)
        self.row_bwd(M=Ms, L_=S, nln="cond_encoder.0.norm1", lin=None if pe_drop else "c2",
                     flags=L.ROWF_NEXT_LN | L.ROWF_STORE_H | L.ROWF_STORE_ROT, z=sv["tok0"],
                     nln_g=P("cond_encoder.0.norm1.weight"), nln_b=P("cond_encoder.0.norm1.bias"), nln_eps=1e-5, rope=self.rope,
                     pos_mod=S, d_xn=g_tok, d_h=g_h, d_rot=g_r, d_z=dtok0)
        dc1 = self.pz("dc1", Ms, sv["c1"].shape[1])
        if pe_drop:
            dtok0 = self.act_bwd(dtok0, dtok0, Ms, 512, L.ACT_NONE, site=SITE_PE_COND)
        lins["c2"].bwd(dtok0, 512, Ms, [sv["c1"]], [("T", dc1, dc1.shape[1])], bias_done=not pe_drop)
        dc0a = self.act_bwd(sv["c0a"], dc1, Ms, self.Cd, L.ACT_RELU)
        lins["c0"].bwd(dc0a, dc0a.shape[1], Ms, [sv["cin"]], [None])
        self.flush_wgrad()
        self.sv = None
        if sync is not None:                              # everything outside the decoder layers, then wait for all of it
            sync.ready(self.flat, 0, self.layer_range[0][0], more=False)
            sync.ready(self.flat, self.layer_range[-1][1], self.n_grad, more=False)
            sync.finish()


class _DenoiserTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng: TrainEngine, x, cond, times, keep, seed, p_drop, *params):
        ctx.eng = eng
        out = eng.forward(x, cond, times, keep, seed, p_drop)
        ctx.gen = eng.sv["gen"]       # the saved activations (and dropout seed) this node's backward has to find
        return out

    @staticmethod
    def backward(ctx, d_out):
        grads = ctx.eng.backward(d_out, gen=ctx.gen)
        return (None,) * 7 + tuple(grads)


def denoiser_train(model, x, cond, times, keep, seed: Tuple[int, int], p_drop: float) -> torch.Tensor:
    """DanceDecoder.forward with an autograd graph (train-mode dropout of probability p_drop, 0 = eval arithmetic)."""
    eng = model.train_engine()
    return _DenoiserTrainFn.apply(eng, x, cond, times, keep, seed, p_drop, *[eng.params[n] for n in eng.order])
