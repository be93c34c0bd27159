"""ctypes binding of libtcdiff_gfx950.so (include/tcdiff_hip.h).

The product path has NO fallback: if the shared library is missing or a launcher returns an error, this
module raises.  Nothing here touches ``oracle/``.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TCDIFF_LIB_PATH") or os.path.join(HERE, "libtcdiff_gfx950.so")   # override: diagnostic builds

DT_F32, DT_BF16, DT_BF16X3 = 0, 1, 2      # TC_DTYPE_* of include/tcdiff_hip.h
ACT_NONE, ACT_RELU, ACT_GELU, ACT_MISH, ACT_SILU = 0, 1, 2, 3, 4
EPI_STORE_T, EPI_STORE_F32, EPI_QKV_HEADS = 0, 1, 2
ROW_BIAS, ROW_LN_POST, ROW_FILM, ROW_RES, ROW_STORE_X, ROW_NEXT_LN, ROW_STORE_H, ROW_STORE_ROT = \
    1, 2, 4, 8, 16, 32, 64, 128
SAMPLER_DDPM, SAMPLER_DDIM = 0, 1
SAMPLER_ADVANCE = 0x100          # OR into the mode: sampler_update also advances counter[3] (step_prologue protocol)
CHAIN_A, CHAIN_B, CHAIN_B_LAST, CHAIN_FULL, CHAIN_FULL_LAST, CHAIN_FRONT = 0, 1, 2, 3, 4, 5

_vp, _i, _f, _l = C.c_void_p, C.c_int, C.c_float, C.c_long


class TileEpi(C.Structure):
    _fields_ = [("mode", _i), ("act", _i), ("scale_q", _f), ("bias", _vp), ("out", _vp), ("out_k", _vp),
                ("out_v", _vp), ("ldc", _i), ("L", _i), ("Lp", _i), ("H", _i), ("n_q", _i), ("n_k", _i),
                ("tok_off", _i), ("seq_off", _i), ("k_splits", _i), ("out2", _vp), ("ldc2", _i), ("act_src", _vp),
                ("ld_src", _i), ("act2", _i), ("drop_seed", _vp), ("drop_site", _i), ("drop_thr", C.c_uint32),
                ("drop_scale", _f), ("hgroup", _i), ("hgroup_stride", _l), ("small_m", _i)]


class RowEpi(C.Structure):
    _fields_ = [("flags", _i), ("bias", _vp), ("ln_g", _vp), ("ln_b", _vp), ("ln_eps", _f), ("film", _vp),
                ("film_ld", _i), ("xres", _vp), ("xres_mod", _i), ("xout", _vp), ("L", _i), ("nln_g", _vp),
                ("nln_b", _vp), ("nln_eps", _f), ("hout", _vp), ("rout", _vp), ("rope", _vp), ("out_mul", _i),
                ("out_add", _i), ("groups", _i)]


class ChainArgs(C.Structure):
    _fields_ = [("mode", _i), ("n_stages", _i), ("M", _i), ("L", _i), ("a_mod", _i), ("xres_mod", _i), ("H", _i),
                ("Lp", _i), ("A", _vp), ("wstream", _vp), ("film", _vp), ("xres", _vp),
                ("xout", _vp), ("n2_g", _vp), ("n2_b", _vp), ("rope", _vp), ("q_out", _vp), ("b1", _vp),
                ("film3", _vp), ("n4_g", _vp), ("n4_b", _vp), ("b3", _vp), ("nn_g", _vp), ("nn_b", _vp),
                ("k_out", _vp), ("v_out", _vp), ("h_out", _vp), ("film_ld", _i), ("ln_eps", _f), ("n2_eps", _f),
                ("n4_eps", _f), ("nn_eps", _f), ("scale_q", _f), ("filmb", _vp),
                ("n3_g", _vp), ("n3_b", _vp), ("kf", _vp), ("vf", _vp), ("n_shared", _i), ("nkt", _i), ("Lk", _i),
                ("xres_rowmajor", _i), ("rope_rows", _i), ("dn", _i), ("mt", _i), ("out_ld", _i), ("nw", _i), ("seq_blocks", _i), ("sa_q", _vp), ("sa_kf", _vp), ("sa_vf", _vp),
                ("sa_nkt", _i), ("qf_out", _vp), ("kf_out", _vp), ("vf_out", _vp), ("out_nkt", _i)]


PROLOGUE_X, PROLOGUE_COND = 1, 2      # tcdiff_step_prologue_args.parts


class StepPrologueArgs(C.Structure):
    _fields_ = [("counter", _vp), ("tseq", _vp), ("tidx", _vp), ("t_base", _vp), ("hidden", _vp), ("film_in", _vp),
                ("n_seq", _i), ("tab", _vp), ("n_t", _i), ("Kc", _vp), ("Vc", _vp), ("Kf", _vp), ("Vf", _vp),
                ("NL", _i), ("n_kv", _i), ("H", _i), ("Lp", _i), ("nkt", _i), ("tok0", _i), ("x", _vp), ("xin", _vp),
                ("rows", _i), ("nfeat", _i), ("ld_xin", _i), ("film_tab", _vp), ("film_out", _vp), ("film_rows", _i),
                ("nfilm", _i), ("n_unc", _i), ("parts", _i)]


class RowArgs(C.Structure):
    """tcdiff_row_args (include/tcdiff_hip.h): the row-local block glue of the training step, forward and backward."""
    _fields_ = [("flags", _i), ("M", _i), ("L", _i), ("z", _vp), ("bias", _vp), ("ln_g", _vp), ("ln_b", _vp),
                ("ln_eps", _f), ("film", _vp), ("film_ld", _i), ("xres", _vp), ("xout", _vp), ("nln_g", _vp),
                ("nln_b", _vp), ("nln_eps", _f), ("hout", _vp), ("rout", _vp), ("rope", _vp), ("pos_mod", _i),
                ("pos_base", _i), ("seed", _vp), ("drop_thr", C.c_uint32), ("drop_scale", _f), ("site_pre", _i),
                ("site_post", _i), ("d_xn", _vp), ("d_h", _vp), ("d_rot", _vp), ("d_z", _vp), ("d_xres", _vp),
                ("d_film", _vp), ("dfilm_ld", _i), ("partials", _vp), ("chunks", _i), ("dz_f32", _i),
                ("g_bias", _vp), ("g_ln_g", _vp), ("g_ln_b", _vp), ("g_nln_g", _vp), ("g_nln_b", _vp)]


class CtDesc(C.Structure):
    """tcdiff_ct_desc (include/tcdiff_hip.h): one matrix of tcdiff_cast_transpose_multi's table."""
    _fields_ = [("src", _vp), ("dst", _vp), ("dstT", _vp), ("rows", _i), ("cols", _i), ("ld_src", _i), ("ld_dst", _i),
                ("cols_pad", _i), ("ld_dstT", _i), ("rows_pad", _i), ("tile0", _i), ("tiles_x", _i), ("vec", _i)]


class WsDesc(C.Structure):
    """tcdiff_ws_desc (include/tcdiff_hip.h): one matrix of tcdiff_pack_row_streams' table."""
    _fields_ = [("src", _vp), ("dst", _vp), ("sn", C.c_long), ("sk", C.c_long), ("N", _i), ("K", _i),
                ("np_dst", _i), ("p0", _i), ("kst_dst", _i), ("ks0", _i)]


class TnProblem(C.Structure):
    """tcdiff_tn_problem (include/tcdiff_hip.h): one weight gradient of tcdiff_gemm_tn_grouped."""
    _fields_ = [("A", _vp), ("B", _vp), ("out", _vp), ("M", _i), ("N", _i), ("K", _i), ("lda", _i), ("ldb", _i), ("ldc", _i),
                ("nk", _i), ("unit0", _i)]


TN_MAX_PROB = 16

ROWF_BIAS, ROWF_DROP_PRE, ROWF_LN_POST, ROWF_DROP_POST, ROWF_FILM, ROWF_RES, ROWF_STORE_X, ROWF_NEXT_LN, ROWF_STORE_H, \
    ROWF_STORE_ROT = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512


class AdanScalars(C.Structure):
    _fields_ = [(n, _f) for n in ("b1", "omb1", "b2", "omb2", "b3", "omb3", "cm", "cv", "cn", "eps", "lr", "denom")] + \
        [("first", _i)]


_SIGS = {
    "tcdiff_gemm_tile": [_i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, C.POINTER(TileEpi), _vp],
    "tcdiff_gemm_rowln": [_i, _vp, _vp, _i, _i, _i, _i, _i, C.POINTER(RowEpi), _vp],
    "tcdiff_attention": [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tcdiff_q_sample_traj": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "tcdiff_ax_from_6v": [_vp, _l, _i, _l, _vp, _vp],
    "tcdiff_smpl_fk": [_vp, _vp, _l, C.POINTER(_i), C.POINTER(_f), _vp, _vp],
    "tcdiff_loss_terms": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "tcdiff_adan_step": [_vp, _i, C.POINTER(AdanScalars), _vp],
    "tcdiff_pack_kv_frags": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "tcdiff_chain": [C.POINTER(ChainArgs), _vp],
    "tcdiff_chain_split": [C.POINTER(ChainArgs), _i, _vp, _vp, _vp],
    "tcdiff_ln_rot": [_i, _vp, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "tcdiff_rope_table": [_vp, _vp, _i, _vp],
    "tcdiff_convert_pad": [_i, _vp, _vp, _i, _i, _i, _i, _l, _l, _vp],
    "tcdiff_sinusoidal": [_i, _vp, _i, _vp, _vp, _vp],
    "tcdiff_mean_pool": [_vp, _vp, _i, _i, _i, _vp],
    "tcdiff_add_act": [_i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "tcdiff_scatter_time_kv": [_i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "tcdiff_step_begin": [_vp, _vp, _vp, _i, _vp],
    "tcdiff_step_end": [_vp, _vp],
    "tcdiff_step_prologue": [_i, _vp, _vp],
    "tcdiff_sampler_update": [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, C.c_uint64, _i, _vp],
    "tcdiff_window_couple": [_vp, _i, _i, _i, _vp],
    "tcdiff_window_couple_step": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "tcdiff_sampler_constrain": [_i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, C.c_uint64, _i, _vp],
    "tcdiff_ema_update": [_vp, _i, _f, _f, _vp],
    "tcdiff_cfg_combine": [_vp, _vp, _i, _f, _vp, _i, _i, _vp],
    # training step (csrc/train_ops.hip, attention_train.hip, gemm.hip)
    "tcdiff_cast_transpose": [_i, _i, _vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp],
    "tcdiff_gemm_splitk": [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp],
    "tcdiff_gemm_tn": [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp],
    "tcdiff_gemm_tn_grouped": [_i, C.POINTER(TnProblem), _i, _vp],
    "tcdiff_ct_desc_init": [_i, C.POINTER(CtDesc)],
    "tcdiff_cast_transpose_multi": [_i, _vp, _i, _i, _vp],
    "tcdiff_pack_row_streams": [_vp, _i, _i, _vp],
    "tcdiff_gemm_rows": [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp],
    "tcdiff_pos_drop": [_vp, _i, _i, _vp, _i, _vp, _i, C.c_uint32, _f, _vp],
    "tcdiff_act_drop": [_i, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, C.c_uint32, _f, _vp],
    "tcdiff_act_drop_bwd": [_i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, C.c_uint32, _f, _vp],
    "tcdiff_row_fwd": [_i, C.POINTER(RowArgs), _vp],
    "tcdiff_row_bwd": [_i, C.POINTER(RowArgs), _vp],
    "tcdiff_row_param_reduce": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "tcdiff_attention_train": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, C.c_uint32, _f, _vp],
    "tcdiff_attention_bwd": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i,
                             _f, _vp, _i, C.c_uint32, _f, _vp],
    "tcdiff_add_rows": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp],
    "tcdiff_select_rows": [_vp, _vp, _vp, _vp, _i, _l, _vp],
    "tcdiff_select_rows_bwd": [_vp, _vp, _vp, _vp, _i, _l, _vp],
    "tcdiff_pool_bwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "tcdiff_loss_total": [_vp, _i, _vp, _vp],
    "tcdiff_loss_terms_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "tcdiff_fk_bwd": [_vp, _vp, _l, _i, C.POINTER(_i), C.POINTER(_f), _vp, _vp],
}

EXPORTS = sorted(list(_SIGS) + ["tcdiff_version"])

_lib = None


class TcdiffError(RuntimeError):
    pass


_rec = None          # a list while train_engine records a command list: every launcher call is appended as (function, arguments)


class _Recorder:
    """Stands in for the library while a step is being recorded: calls go through unchanged and are remembered."""

    def __init__(self, lib, out):
        self._lib, self._out = lib, out

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        out = self._out

        def call(*args):
            rc = fn(*args)
            out.append((fn, args))
            return rc
        return call


def load():
    """Load the HIP library; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib if _rec is None else _Recorder(_lib, _rec)
    if not os.path.exists(LIB_PATH):
        raise TcdiffError(f"{LIB_PATH} not found: build it with `python -m tcdiff_amd.build` "
                          "(the MI355X path has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _i
    lib.tcdiff_version.restype = C.c_char_p
    lib.tcdiff_version.argtypes = []
    _lib = lib
    return lib


_ERR = {-1: "invalid argument", -2: "misaligned pointer / leading dimension", -3: "kernel launch failed",
        -4: "unsupported configuration"}


def check(rc: int, what: str):
    if rc != 0:
        raise TcdiffError(f"{what}: {_ERR.get(rc, 'error')} (code {rc})")


def ptr(t):
    """data pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()
