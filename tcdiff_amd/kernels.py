"""Thin tensor-level wrappers over the C ABI (include/tcdiff_hip.h).  torch is used only for device
memory and the current HIP stream; every computation is a launcher of libtcdiff_gfx950.so."""
import ctypes as C

import torch

from . import _lib as L

TORCH_DT = {L.DT_F32: torch.float32, L.DT_BF16: torch.bfloat16, L.DT_BF16X3: torch.float32}


def dtype_id(name_or_dtype) -> int:
    if isinstance(name_or_dtype, str) and name_or_dtype == "bf16x3":
        return L.DT_BF16X3
    if name_or_dtype in ("bf16", torch.bfloat16, L.DT_BF16):
        return L.DT_BF16
    if name_or_dtype in ("f32", "fp32", torch.float32, L.DT_F32):
        return L.DT_F32
    if name_or_dtype == L.DT_BF16X3 and not isinstance(name_or_dtype, bool):
        return L.DT_BF16X3
    raise ValueError(f"unsupported compute dtype {name_or_dtype!r} (bf16, f32 or bf16x3)")


def k_tile(dt: int) -> int:
    return 64 if dt == L.DT_BF16 else 32


def _sdt(dt: int) -> int:
    """the split-bf16 mode (TC_DTYPE_BF16X3) exists on the sampler's path only (GEMM / attention / elementwise launchers of the
    denoiser); the training-step launchers never see it (TrainEngine maps the mode to f32)"""
    return L.DT_F32 if dt == L.DT_BF16X3 else dt


def to_x3(w: torch.Tensor) -> torch.Tensor:
    """fp32 [..., K] (K % 4 == 0) -> the split-bf16 storage of csrc/common.h MmaBF16x3, returned as a float32-typed tensor of the
    same shape holding the bit patterns: every 16-byte chunk of four elements becomes [hi x4 | lo x4] bf16 with hi = bf16(e),
    lo = bf16(e - hi) (round to nearest even, as v_cvt_pk_bf16_f32)."""
    w = w.to(torch.float32).contiguous()
    if w.shape[-1] % 4:
        raise ValueError("to_x3: the last dimension must be a multiple of 4")
    hi = w.to(torch.bfloat16)
    lo = (w - hi.to(torch.float32)).to(torch.bfloat16)
    q = w.shape[:-1] + (w.shape[-1] // 4, 4)
    out = torch.cat([hi.view(torch.int16).reshape(q), lo.view(torch.int16).reshape(q)], -1).contiguous()   # [..., K/4, 8] int16
    return out.view(torch.float32).reshape(w.shape)


def from_x3(t: torch.Tensor) -> torch.Tensor:
    """inverse of to_x3 (tests, diagnostics): hi + lo as fp32"""
    q = t.contiguous().view(torch.int16).reshape(t.shape[:-1] + (t.shape[-1] // 4, 8))
    hi = q[..., :4].contiguous().view(torch.bfloat16).to(torch.float32)
    lo = q[..., 4:].contiguous().view(torch.bfloat16).to(torch.float32)
    return (hi + lo).reshape(t.shape)


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    """device address for a c_void_p argument / struct field (a plain int converts as well as a c_void_p object and costs a third)"""
    return None if t is None else t.data_ptr()


def gemm_tile(dt, A, W, M, N, K, *, lda=None, ldw=None, A2=None, split_n=0, a_mod=0, mode=L.EPI_STORE_T,
              act=L.ACT_NONE, bias=None, out=None, ldc=0, out_k=None, out_v=None, scale_q=1.0, Lseq=0, Lp=0, H=0,
              n_q=0, n_k=0, tok_off=0, seq_off=0, out2=None, ldc2=0, act_src=None, ld_src=0, act2=L.ACT_NONE, seed=None,
              site=0, thr=0, drop_scale=1.0, hgroup=0, hgroup_stride=0, small_m=False):
    """out2 / act_src: the training step's fused activation epilogues; small_m: the launcher may take its small-M kernel
    (include/tcdiff_hip.h tcdiff_tile_epi)."""
    lib = L.load()
    e = L.TileEpi(mode, act, scale_q, _p(bias), _p(out), _p(out_k), _p(out_v), ldc, Lseq, Lp, H, n_q, n_k, tok_off,
                  seq_off, 0, _p(out2), ldc2, _p(act_src), ld_src, act2, _p(seed), site, thr, drop_scale, hgroup, hgroup_stride,
                  int(bool(small_m)))
    rc = lib.tcdiff_gemm_tile(dt, _p(A), _p(A2), split_n, _p(W), M, N, K, lda if lda else K, ldw if ldw else K,
                              a_mod, C.byref(e), stream())
    L.check(rc, "tcdiff_gemm_tile")


def gemm_rows_ok(dt, N, K) -> bool:
    """shapes tcdiff_gemm_rows takes (bf16, K = 512 or 1024, N a multiple of 512)"""
    return dt == L.DT_BF16 and K in (512, 1024) and N > 0 and N % 512 == 0


def gemm_rows(A, wstream, M, N, K, *, lda=None, A2=None, split_n=0, mode=L.EPI_STORE_T, bias=None, out=None, ldc=0,
              out_k=None, out_v=None, scale_q=1.0, Lseq=0, Lp=0, H=0, n_q=0, n_k=0, out2=None, ldc2=0, act_src=None, ld_src=0,
              act2=L.ACT_NONE, seed=None, site=0, thr=0, drop_scale=1.0, mt=0):
    """C[M, N] = A[M, K] Wn^T with gemm_tile's epilogues; wstream = row_streams() of Wn: [8 waves][N/512 * K/32][2048] bf16"""
    if tuple(wstream.shape) != (8, (N // 512) * (K // 32), 2048) or not wstream.is_contiguous():
        raise L.TcdiffError(f"weight stream of a [{N}, {K}] matrix must be contiguous [8, {(N // 512) * (K // 32)}, 2048], got "
                            f"{tuple(wstream.shape)}")
    e = L.TileEpi(mode, L.ACT_NONE, scale_q, _p(bias), _p(out), _p(out_k), _p(out_v), ldc, Lseq, Lp, H, n_q, n_k, 0, 0, 0,
                  _p(out2), ldc2, _p(act_src), ld_src, act2, _p(seed), site, thr, drop_scale, 0, 0)
    rc = L.load().tcdiff_gemm_rows(_p(A), _p(A2), split_n, _p(wstream), M, N, K, lda if lda else K, C.byref(e), mt, stream())
    L.check(rc, "tcdiff_gemm_rows")


def ws_table(entries, device):
    """Device table for pack_row_streams.  entries: dicts with src (fp32 tensor, element (n, k) at src.view(-1)[n sn + k sk]),
    sn, sk, N, K, dst ([8, np_dst * kst_dst, 2048] bf16) and, for a piece of a stacked matrix, np_dst, p0, kst_dst, ks0 (default:
    the whole matrix).  Returns (table tensor, n_desc, max N * K); the table holds raw pointers (the caller keeps the tensors
    alive and rebuilds it when one is reallocated)."""
    n = len(entries)
    arr = (L.WsDesc * n)()
    mx = 0
    for d, e in zip(arr, entries):
        N, K_ = e["N"], e["K"]
        if e["src"].dtype != torch.float32 or N % 512 or K_ % 32:
            raise L.TcdiffError("pack_row_streams takes fp32 sources with N % 512 == 0 and K % 32 == 0")
        npd, p0, kd, ks0 = e.get("np_dst", N // 512), e.get("p0", 0), e.get("kst_dst", K_ // 32), e.get("ks0", 0)
        if p0 + N // 512 > npd or ks0 + K_ // 32 > kd:
            raise L.TcdiffError("pack_row_streams: the piece does not fit the stream")
        if e["dst"].numel() != 8 * npd * kd * 2048 or e["dst"].element_size() != 2 or not e["dst"].is_contiguous():
            raise L.TcdiffError("pack_row_streams: dst must be a contiguous 2-byte tensor of [8, np_dst * kst_dst, 2048]")
        d.src, d.dst, d.sn, d.sk, d.N, d.K = _p(e["src"]), _p(e["dst"]), e["sn"], e["sk"], N, K_
        d.np_dst, d.p0, d.kst_dst, d.ks0 = npd, p0, kd, ks0
        mx = max(mx, N * K_)
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
    return raw.to(device), n, mx


def pack_row_streams(table):
    tab, n, mx = table
    L.check(L.load().tcdiff_pack_row_streams(_p(tab), n, mx, stream()), "tcdiff_pack_row_streams")


def row_streams(W, transposed=False):
    """the stream pack of one fp32 matrix W [R, C] (a convenience over ws_table + pack_row_streams): Wn = W (N = R, K = C) or,
    transposed, Wn = W^T (N = C, K = R)"""
    R, Cc = W.shape
    N, K_ = (Cc, R) if transposed else (R, Cc)
    dst = torch.empty(8, (N // 512) * (K_ // 32), 2048, device=W.device, dtype=torch.bfloat16)
    ld = W.stride(0)
    pack_row_streams(ws_table([dict(src=W, sn=1 if transposed else ld, sk=ld if transposed else 1, N=N, K=K_, dst=dst)], W.device))
    return dst


def gemm_rowln(dt, A, W, M, K, *, flags, lda=None, ldw=None, a_mod=0, bias=None, ln_g=None, ln_b=None, ln_eps=1e-6,
               film=None, film_ld=0, xres=None, xres_mod=0, xout=None, Lseq=1, nln_g=None, nln_b=None, nln_eps=1e-5,
               hout=None, rout=None, rope=None, out_mul=1, out_add=0, groups=1):
    lib = L.load()
    e = L.RowEpi(flags, _p(bias), _p(ln_g), _p(ln_b), ln_eps, _p(film), film_ld, _p(xres), xres_mod, _p(xout), Lseq,
                 _p(nln_g), _p(nln_b), nln_eps, _p(hout), _p(rout), _p(rope), out_mul, out_add, groups)
    rc = lib.tcdiff_gemm_rowln(dt, _p(A), _p(W), M, K, lda if lda else K, ldw if ldw else K, a_mod, C.byref(e),
                               stream())
    L.check(rc, "tcdiff_gemm_rowln")


def attention(dt, Q, K, V, O, n_seq, H, Lq, Lk, Lp_q, Lp_k, ldo, n_shared=0, ng=0):
    rc = L.load().tcdiff_attention(dt, _p(Q), _p(K), _p(V), _p(O), n_seq, H, Lq, Lk, Lp_q, Lp_k, ldo, n_shared, ng,
                                   stream())
    L.check(rc, "tcdiff_attention")


def chain(mode, M, Lseq, A, wstream, *, ln_eps=1e-6, film=None, film_ld=0, xres=None,
          xout=None, n2_g=None, n2_b=None, n2_eps=1e-5,
          rope=None, q_out=None, scale_q=0.125, Lp=0, H=8, a_mod=0, xres_mod=0, b1=None, film3=None, n4_g=None,
          n4_b=None, n4_eps=1e-5, b3=None, nn_g=None, nn_b=None, nn_eps=1e-5, k_out=None, v_out=None, h_out=None,
          filmb=None, n3_g=None, n3_b=None, kf=None, vf=None, n_shared=0, nkt=0, Lk=0,
          xres_rowmajor=False, out_ld=0, dn=1, mt=0, seq_blocks=False, sa_q=None, sa_kf=None, sa_vf=None, sa_nkt=0,
          qf_out=None, kf_out=None, vf_out=None, out_nkt=0, split_part=None, p_in=None, p_out=None):
    """rope: the COLUMN-BLOCKED rotary table (to_cb(rope_table)); xres / xout: column-blocked fp32 residual stream
    (xres_rowmajor: xres is a plain [*, 512] matrix); wstream: [(dn,) 8 waves, n_stages, 2048] bf16 (or the 4-wave form
    [(dn,) 4, n_stages, 4096]: tcdiff_chain_args.nw), the stage and wave counts are read from it.  film / filmb / film3 rows are PRE-FOLDED with the LayerNorm weights / linear2 bias around them
    (fold_film below; engine.load_weights folds them into the FiLM generator).  seq_blocks / sa_* / *f_out: row blocks cut per
    sequence, the layer's self-attention inside the launch and the fragment-order Q / K / V outputs that feed it.
    split_part = 1 .. 4: the small-job form of the layer, four workgroups per 16-row block and four launches per layer
    (tcdiff_chain_split: xres -> xout not in place, p_in / p_out the partial-sum buffers); split_part = 0 with CHAIN_FRONT: layer 0's
    Q / K / V fragment images from row-major token rows.
    See include/tcdiff_hip.h tcdiff_chain_args."""
    nw = wstream.shape[-3]
    if (nw, wstream.shape[-1]) not in ((8, 2048), (4, 4096)) or not wstream.is_contiguous():
        raise L.TcdiffError("weight stream must be contiguous [.., 8 waves, n_stages, 2048] or [.., 4 waves, n_stages, 4096] bf16, "
                            f"got {tuple(wstream.shape)}")
    n_stages = wstream.shape[-2]
    a = L.ChainArgs(mode, n_stages, M, Lseq, a_mod, xres_mod, H, Lp, _p(A), _p(wstream), _p(film),
                    _p(xres), _p(xout), _p(n2_g), _p(n2_b), _p(rope), _p(q_out), _p(b1), _p(film3), _p(n4_g),
                    _p(n4_b), _p(b3), _p(nn_g), _p(nn_b), _p(k_out), _p(v_out), _p(h_out), film_ld, ln_eps, n2_eps,
                    n4_eps, nn_eps, scale_q, _p(filmb), _p(n3_g), _p(n3_b), _p(kf), _p(vf), n_shared,
                    nkt, Lk, int(bool(xres_rowmajor)), 0 if rope is None else rope.shape[1], dn, mt, out_ld, nw,
                    int(bool(seq_blocks)), _p(sa_q), _p(sa_kf), _p(sa_vf), sa_nkt, _p(qf_out), _p(kf_out), _p(vf_out), out_nkt)
    if split_part is not None:
        L.check(L.load().tcdiff_chain_split(C.byref(a), split_part, _p(p_in), _p(p_out), stream()), "tcdiff_chain_split")
        return
    L.check(L.load().tcdiff_chain(C.byref(a), stream()), "tcdiff_chain")


def fold_film(film: torch.Tensor, g, b) -> torch.Tensor:
    """[rows, 1024] raw DenseFiLM rows (scale | shift) -> the pre-folded rows the chain kernels take:
    [g (scale + 1) | b (scale + 1) + shift]; g = None stands for 1 (the feed-forward block, b = linear2's bias)."""
    sc1 = film[:, :512] + 1.0
    return torch.cat([sc1 if g is None else g * sc1, b * sc1 + film[:, 512:1024]], 1).contiguous()


def to_cb(x: torch.Tensor) -> torch.Tensor:
    """[rows, 512] -> column-blocked [64, rows, 8] (element (row, c) at ((c // 8) * rows + row) * 8 + c % 8)."""
    rows = x.shape[0]
    return x.reshape(rows, 64, 8).permute(1, 0, 2).contiguous()


def from_cb(x: torch.Tensor, rows: int) -> torch.Tensor:
    """column-blocked [64, rows, 8] (any shape with 512 * rows elements) -> [rows, 512]"""
    return x.reshape(64, rows, 8).permute(1, 0, 2).reshape(rows, 512).contiguous()


def pack_kv_frags(Kc, Vc, Kf, Vf, n_slots, H, Lp, nkt, key_lo, key_hi):
    L.check(L.load().tcdiff_pack_kv_frags(_p(Kc), _p(Vc), _p(Kf), _p(Vf), n_slots, H, Lp, nkt, key_lo, key_hi, stream()),
            "tcdiff_pack_kv_frags")


def ln_rot(dt, x, rows, g, b, eps, *, h=None, rot=None, y32=None, rope=None, pos_mod=0, pos_base=0):
    rc = L.load().tcdiff_ln_rot(dt, _p(x), rows, _p(g), _p(b), eps, _p(h), _p(rot), _p(y32), _p(rope), pos_mod,
                                pos_base, stream())
    L.check(rc, "tcdiff_ln_rot")


def rope_table(freqs, rope, n_pos):
    L.check(L.load().tcdiff_rope_table(_p(freqs), _p(rope), n_pos, stream()), "tcdiff_rope_table")


def convert_pad(dt, src, dst, rows, cols, ld_dst, rows_per_batch=None, batch_stride=0, row_stride=None):
    rpb = rows if rows_per_batch is None else rows_per_batch
    rs = cols if row_stride is None else row_stride
    rc = L.load().tcdiff_convert_pad(dt, _p(src), _p(dst), rows, cols, ld_dst, rpb, batch_stride, rs, stream())
    L.check(rc, "tcdiff_convert_pad")


def sinusoidal(dt, times_i32, n, freq, emb):
    L.check(L.load().tcdiff_sinusoidal(dt, _p(times_i32), n, _p(freq), _p(emb), stream()), "tcdiff_sinusoidal")


def mean_pool(x, out, B, S, Cn):
    L.check(L.load().tcdiff_mean_pool(_p(x), _p(out), B, S, Cn, stream()), "tcdiff_mean_pool")


def add_act(dt, a, ia, b, n, act, out=None, out32=None):
    L.check(L.load().tcdiff_add_act(dt, _p(a), _p(ia), _p(b), n, act, _p(out), _p(out32), stream()),
            "tcdiff_add_act")


def scatter_time_kv(dt, tab, n_t, tidx, Kc, Vc, NL, n_kv, H, Lp, tok0):
    rc = L.load().tcdiff_scatter_time_kv(dt, _p(tab), n_t, _p(tidx), _p(Kc), _p(Vc), NL, n_kv, H, Lp, tok0,
                                         stream())
    L.check(rc, "tcdiff_scatter_time_kv")


def step_begin(counter, tseq, tidx, n):
    L.check(L.load().tcdiff_step_begin(_p(counter), _p(tseq), _p(tidx), n, stream()), "tcdiff_step_begin")


def step_end(counter):
    L.check(L.load().tcdiff_step_end(_p(counter), stream()), "tcdiff_step_end")


def step_prologue(dt, counter, tseq, tidx, t_base, hidden, film_in, n_seq, tab, n_t, Kc, Vc, Kf, Vf, NL, n_kv, H, Lp, nkt,
                  tok0, x=None, xin=None, rows=0, nfeat=0, ld_xin=0, film_tab=None, film_out=None, film_rows=0, nfilm=0,
                  n_unc=0, parts=0):
    a = L.StepPrologueArgs(_p(counter), _p(tseq), _p(tidx), _p(t_base), _p(hidden), _p(film_in), n_seq, _p(tab), n_t,
                           _p(Kc), _p(Vc), _p(Kf), _p(Vf), NL, n_kv, H, Lp, nkt, tok0, _p(x), _p(xin), rows, nfeat,
                           ld_xin, _p(film_tab), _p(film_out), film_rows, nfilm, n_unc, parts)
    L.check(L.load().tcdiff_step_prologue(dt, C.byref(a), stream()), "tcdiff_step_prologue")


def sampler_update(mode, out_unc, out_cond, ldo, x, eps, traj, x0_out, n_rows, nfeat, Lseq, counter, params, tseq,
                   seed=0, clip0=0):
    rc = L.load().tcdiff_sampler_update(mode, _p(out_unc), _p(out_cond), ldo, _p(x), _p(eps), _p(traj), _p(x0_out),
                                        n_rows, nfeat, Lseq, _p(counter), _p(params), _p(tseq), seed, clip0,
                                        stream())
    L.check(rc, "tcdiff_sampler_update")


def window_couple(x, b, seq_len, row_elems):
    L.check(L.load().tcdiff_window_couple(_p(x), b, seq_len, row_elems, stream()), "tcdiff_window_couple")


def sampler_constrain(kind, x, mask, mask_rows, value, q_eps, n_rows, nfeat, Lseq, counter, params, tseq, seed=0, clip0=0):
    rc = L.load().tcdiff_sampler_constrain(kind, _p(x), _p(mask), mask_rows, _p(value), _p(q_eps), n_rows, nfeat, Lseq,
                                           _p(counter), _p(params), _p(tseq), seed, clip0, stream())
    L.check(rc, "tcdiff_sampler_constrain")


def window_couple_step(x, b, seq_len, row_elems, counter, params):
    L.check(L.load().tcdiff_window_couple_step(_p(x), b, seq_len, row_elems, _p(counter), _p(params), stream()),
            "tcdiff_window_couple_step")


def cfg_combine(out_unc, out_cond, ldo, w, y, n_rows, nfeat):
    L.check(L.load().tcdiff_cfg_combine(_p(out_unc), _p(out_cond), ldo, float(w), _p(y), n_rows, nfeat, stream()),
            "tcdiff_cfg_combine")


def ema_chunk_table(ma_tensors, cur_tensors, device):
    """Device table of (ma ptr, cur ptr, n) chunks of <= 65536 elements for tcdiff_ema_update (int64 [n_chunks, 3])."""
    rows = []
    for a, c in zip(ma_tensors, cur_tensors):
        n = a.numel()
        for lo in range(0, n, 65536):
            rows.append((a.data_ptr() + 4 * lo, c.data_ptr() + 4 * lo, min(65536, n - lo)))
    return torch.tensor(rows, dtype=torch.int64, device=device).reshape(-1, 3)


def ema_update(table, beta):
    L.check(L.load().tcdiff_ema_update(_p(table), table.shape[0], float(beta), float(1.0 - beta), stream()),
            "tcdiff_ema_update")



# ---- training-side rows ------------------------------------------------------------------------------------------------
def q_sample_traj(x_start, noise, t, sqrt_ac, sqrt_1mac, x_noisy, b, dn, S, Cn):
    L.check(L.load().tcdiff_q_sample_traj(_p(x_start), _p(noise), _p(t), _p(sqrt_ac), _p(sqrt_1mac), _p(x_noisy), b, dn,
                                          S, Cn, stream()), "tcdiff_q_sample_traj")


def ax_from_6v(rot6d, n_rows, per_row, row_stride, out):
    L.check(L.load().tcdiff_ax_from_6v(_p(rot6d), n_rows, per_row, row_stride, _p(out), stream()), "tcdiff_ax_from_6v")


def smpl_fk(axis_angle, root, n, parents, offsets, joints):
    par = (C.c_int * 24)(*[int(p) for p in parents])
    off = (C.c_float * 72)(*[float(v) for row in offsets for v in row])
    L.check(L.load().tcdiff_smpl_fk(_p(axis_angle), _p(root), n, par, off, _p(joints), stream()), "tcdiff_smpl_fk")


def loss_terms(model_out, x_start, joints_model, joints_target, p2_weight, t, out, b, dn, S, Cn, l1):
    L.check(L.load().tcdiff_loss_terms(_p(model_out), _p(x_start), _p(joints_model), _p(joints_target), _p(p2_weight),
                                       _p(t), _p(out), b, dn, S, Cn, int(l1), stream()), "tcdiff_loss_terms")


def adan_chunk_table(params, grads, ms, vs, ns, pgs, device):
    """int64 [n_chunks, 7] device table of (p, g, m, v, n, prev_grad pointers, count) for tcdiff_adan_step."""
    rows = []
    for p, g, m, v, n, pg in zip(params, grads, ms, vs, ns, pgs):
        cnt = p.numel()
        for lo in range(0, cnt, 65536):
            o = 4 * lo
            rows.append((p.data_ptr() + o, g.data_ptr() + o, m.data_ptr() + o, v.data_ptr() + o, n.data_ptr() + o,
                         pg.data_ptr() + o, min(65536, cnt - lo)))
    return torch.tensor(rows, dtype=torch.int64, device=device).reshape(-1, 7)


def adan_step(table, scalars):
    L.check(L.load().tcdiff_adan_step(_p(table), table.shape[0], C.byref(scalars), stream()), "tcdiff_adan_step")


# ---- training step: train-mode forward pieces and the backward pass ---------------------------------------------------
def drop_params(p: float):
    """(threshold, scale) of the counter-hash dropout (csrc/train_common.h): keep iff hash >= floor(p * 2^32)."""
    if p <= 0.0:
        return 0, 1.0
    return int(p * 4294967296.0), 1.0 / (1.0 - p)


def cast_transpose(dt, src, rows, cols, ld_src, *, dst=None, ld_dst=0, cols_pad=0, dstT=None, ld_dstT=0, rows_pad=0,
                   colsum=None):
    src_f32 = int(src.dtype == torch.float32)           # fp32 source (always, in the f32 mode) or a T-typed one
    rc = L.load().tcdiff_cast_transpose(_sdt(dt), src_f32, _p(src), rows, cols, ld_src, _p(dst), ld_dst, cols_pad, _p(dstT),
                                        ld_dstT, rows_pad, _p(colsum), stream())
    L.check(rc, "tcdiff_cast_transpose")


def gemm_splitk(dt, A, W, M, N, K, lda, ldw, out, ldc, splits):
    L.check(L.load().tcdiff_gemm_splitk(_sdt(dt), _p(A), _p(W), M, N, K, lda, ldw, _p(out), ldc, splits, stream()),
            "tcdiff_gemm_splitk")


def gemm_tn_ok(dt, M, N, K) -> bool:
    """shapes tcdiff_gemm_tn takes (everything else is repacked by cast_transpose and goes through gemm_splitk)"""
    return M % 128 == 0 and N % 128 == 0 and K % k_tile(dt) == 0


def gemm_tn(dt, A, B, M, N, K, lda, ldb, out, ldc, splits):
    """out[m][n] += sum_k A[k][m] B[k][n] (both operands row-major over the contraction index)."""
    L.check(L.load().tcdiff_gemm_tn(_sdt(dt), _p(A), _p(B), M, N, K, lda, ldb, _p(out), ldc, splits, stream()), "tcdiff_gemm_tn")


def gemm_tn_grouped(dt, problems):
    """problems: list of (A, B, M, N, K, lda, ldb, out, ldc) as gemm_tn takes them, at most L.TN_MAX_PROB: all of them in ONE
    launch with the work split evenly over the CUs (the weight gradients a decoder layer's backward has queued)."""
    n = len(problems)
    arr = (L.TnProblem * n)()
    for d, (A, B, M, N, K, lda, ldb, out, ldc) in zip(arr, problems):
        d.A, d.B, d.out = _p(A), _p(B), _p(out)
        d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, lda, ldb, ldc
    L.check(L.load().tcdiff_gemm_tn_grouped(_sdt(dt), arr, n, stream()), "tcdiff_gemm_tn_grouped")


def ct_table(dt, entries, device):
    """Device table for cast_transpose_multi.  entries: dicts with src (fp32 tensor view), rows, cols, ld_src and dst /
    ld_dst / cols_pad and / or dstT / ld_dstT / rows_pad.  Returns (table tensor, n_desc, n_tiles); the table holds raw
    pointers, so the caller keeps the tensors alive and rebuilds it when one of them is reallocated."""
    lib = L.load()
    n, tile0 = len(entries), 0
    arr = (L.CtDesc * n)()
    for d, e in zip(arr, entries):
        if e["src"].dtype != torch.float32:
            raise L.TcdiffError("cast_transpose_multi takes fp32 sources")
        d.src, d.dst, d.dstT = _p(e["src"]), _p(e.get("dst")), _p(e.get("dstT"))
        d.rows, d.cols, d.ld_src = e["rows"], e["cols"], e["ld_src"]
        d.ld_dst, d.cols_pad, d.ld_dstT, d.rows_pad = e.get("ld_dst", 0), e.get("cols_pad", 0), e.get("ld_dstT", 0), \
            e.get("rows_pad", 0)
        d.tile0 = tile0
        cnt = lib.tcdiff_ct_desc_init(_sdt(dt), C.byref(d))
        L.check(min(cnt, 0), "tcdiff_ct_desc_init")
        tile0 += cnt
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
    return raw.to(device), n, tile0


def cast_transpose_multi(dt, table):
    tab, n, tiles = table
    L.check(L.load().tcdiff_cast_transpose_multi(_sdt(dt), _p(tab), n, tiles, stream()), "tcdiff_cast_transpose_multi")


def pos_drop(x, rows, cols, pe=None, pos_mod=1, seed=None, site=0, thr=0, scale=1.0):
    """x = dropout(x + pe[row % pos_mod]) in place, fp32 rows (PositionalEncoding in train mode, model/utils.py:27-32)"""
    L.check(L.load().tcdiff_pos_drop(_p(x), rows, cols, _p(pe), pos_mod, _p(seed), site, thr, scale, stream()), "tcdiff_pos_drop")


def act_drop(dt, a, ld_a, y, ld_y, rows, cols, act, seed=None, site=0, thr=0, scale=1.0):
    a_f32 = int(a.dtype == torch.float32)
    L.check(L.load().tcdiff_act_drop(_sdt(dt), a_f32, _p(a), ld_a, _p(y), ld_y, rows, cols, act, _p(seed), site, thr, scale,
                                     stream()), "tcdiff_act_drop")


def act_drop_bwd(dt, a, ld_a, dy, ld_y, da, rows, cols, act, seed=None, site=0, thr=0, scale=1.0):
    a_f32 = int(a.dtype == torch.float32)
    L.check(L.load().tcdiff_act_drop_bwd(_sdt(dt), a_f32, _p(a), ld_a, _p(dy), ld_y, _p(da), rows, cols, act, _p(seed), site,
                                         thr, scale, stream()), "tcdiff_act_drop_bwd")


def row_args(**kw) -> "L.RowArgs":
    return L.RowArgs(**{k: (v.data_ptr() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})


def row_fwd(dt, a):
    L.check(L.load().tcdiff_row_fwd(_sdt(dt), C.byref(a), stream()), "tcdiff_row_fwd")


def row_bwd(dt, a):
    L.check(L.load().tcdiff_row_bwd(_sdt(dt), C.byref(a), stream()), "tcdiff_row_bwd")


def row_param_reduce(partials, n_blocks, d_bias=None, d_ln_g=None, d_ln_b=None, d_nln_g=None, d_nln_b=None):
    L.check(L.load().tcdiff_row_param_reduce(_p(partials), n_blocks, _p(d_bias), _p(d_ln_g), _p(d_ln_b), _p(d_nln_g),
                                             _p(d_nln_b), stream()), "tcdiff_row_param_reduce")


def attention_train(dt, Q, K, V, O, lse, n_seq, H, Lq, Lk, Lp_q, Lp_k, ldo, seed, site, thr, scale, O_lo=None):
    """O_lo (bf16 mode): a second image like O that receives what O's 8 bits dropped; hand it to attention_bwd"""
    L.check(L.load().tcdiff_attention_train(_sdt(dt), _p(Q), _p(K), _p(V), _p(O), _p(O_lo), _p(lse), n_seq, H, Lq, Lk, Lp_q, Lp_k,
                                            ldo, _p(seed), site, thr, scale, stream()), "tcdiff_attention_train")


def attention_bwd(dt, Q, K, V, O, dO, lse, delta, dQ, ld_dq, dK, dV, ld_dkv, n_seq, H, Lq, Lk, Lp_q, Lp_k, ldo, scale_q,
                  seed, site, thr, scale, O_lo=None):
    L.check(L.load().tcdiff_attention_bwd(_sdt(dt), _p(Q), _p(K), _p(V), _p(O), _p(O_lo), _p(dO), _p(lse), _p(delta), _p(dQ),
                                          ld_dq, _p(dK), _p(dV), ld_dkv, n_seq, H, Lq, Lk, Lp_q, Lp_k, ldo, scale_q, _p(seed),
                                          site, thr, scale, stream()), "tcdiff_attention_bwd")


def add_rows(a, ld_a, b, ld_b, out, ld_out, rows, cols):
    L.check(L.load().tcdiff_add_rows(_p(a), ld_a, _p(b), ld_b, _p(out), ld_out, rows, cols, stream()), "tcdiff_add_rows")


def select_rows(x, nul, keep_u8, out, B, n):
    L.check(L.load().tcdiff_select_rows(_p(x), _p(nul), _p(keep_u8), _p(out), B, n, stream()), "tcdiff_select_rows")


def select_rows_bwd(g, keep_u8, dx, dnul, B, n):
    L.check(L.load().tcdiff_select_rows_bwd(_p(g), _p(keep_u8), _p(dx), _p(dnul), B, n, stream()),
            "tcdiff_select_rows_bwd")


def pool_bwd(g_tok, g_pool, dx, B, S, Cn):
    L.check(L.load().tcdiff_pool_bwd(_p(g_tok), _p(g_pool), _p(dx), B, S, Cn, stream()), "tcdiff_pool_bwd")


def loss_total(terms, b, out):
    L.check(L.load().tcdiff_loss_total(_p(terms), b, _p(out), stream()), "tcdiff_loss_total")


def loss_terms_bwd(model_out, x_start, joints_model, joints_target, p2_weight, t, gscale, d_out, d_joints, b, dn, S, Cn,
                   l1):
    L.check(L.load().tcdiff_loss_terms_bwd(_p(model_out), _p(x_start), _p(joints_model), _p(joints_target), _p(p2_weight),
                                           _p(t), _p(gscale), _p(d_out), _p(d_joints), b, dn, S, Cn, int(l1), stream()),
            "tcdiff_loss_terms_bwd")


def fk_bwd(motion, d_joints, n, Cn, parents, offsets, d_out):
    par = (C.c_int * 24)(*[int(p) for p in parents])
    off = (C.c_float * 72)(*[float(v) for row in offsets for v in row])
    L.check(L.load().tcdiff_fk_bwd(_p(motion), _p(d_joints), n, Cn, par, off, _p(d_out), stream()), "tcdiff_fk_bwd")
