"""The weight-stream and K / V fragment images of the chain kernel (csrc/chain.hip, round 4: v_mfma_f32_16x16x32_bf16),
checked on the CPU against a lane-exact emulation of the instruction's operand maps (cdna_hip_programming.md section 3:
A[row l & 15][k = 8 (l >> 4) + j], B[k = 8 (l >> 4) + j][col l & 15], C/D col = l & 15, row = 4 (l >> 4) + reg): a wrong
permutation in engine._stages_* or in the kf / vf index maps shows up here, without a GPU."""
import numpy as np
import torch

from tcdiff_amd.engine import DenoiserEngine as E

PI = (0, 3, 1, 2)
LANE = np.arange(64)
C, G = LANE & 15, LANE >> 4


def mfma16(a_frag, b_frag, acc):
    """a_frag, b_frag: [64 lanes][8]; acc: [64 lanes][4] -> acc + A B with the hardware's lane maps"""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a_frag[l]
        B[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b_frag[l]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        out[l] += D[4 * (l >> 4):4 * (l >> 4) + 4, l & 15]
    return out


def act_frag(act, ks, mt):
    """what a lane reads from the LDS activation block for k-step ks, row tile mt: row 16 mt + c, chunk PI[g] of the step"""
    f = np.zeros((64, 8))
    for l in range(64):
        k0 = 32 * ks + 8 * PI[l >> 4]
        f[l] = act[16 * mt + (l & 15), k0:k0 + 8]
    return f


def run_gemm(stages, act, wave, nst, ntiles=4, k2=1):
    """stages: [nst][512 ntiles k2] of one wave; returns acc[nt][mt][64][4]"""
    acc = np.zeros((ntiles, 4, 64, 4))
    for st in range(nst):
        fr = stages[st].reshape(k2, ntiles, 64, 8)
        for kk in range(k2):
            ks = k2 * st + kk
            for mt in range(4):
                b = act_frag(act, ks, mt)
                for nt in range(ntiles):
                    acc[nt, mt] = mfma16(fr[kk, nt], b, acc[nt, mt])
    return acc


def acc_to_matrix(acc, ncols):
    """acc[nt][mt][lane][r] -> [64 rows][ncols]: column 16 nt + 4 g + r of row 16 mt + c"""
    out = np.zeros((64, ncols))
    for nt in range(acc.shape[0]):
        for mt in range(4):
            for l in range(64):
                out[16 * mt + (l & 15), 16 * nt + 4 * (l >> 4):16 * nt + 4 * (l >> 4) + 4] = acc[nt, mt, l]
    return out


def test_weight_stream_images_compute_the_gemms():
    rng = np.random.default_rng(0)
    for K_ in (512, 1024):
        W = rng.integers(-3, 4, (512, K_)).astype(np.float64)
        act = rng.integers(-3, 4, (64, K_)).astype(np.float64)
        st = E._stages_n512(torch.from_numpy(W)).numpy()
        assert st.shape == (8, K_ // 32, 2048)
        want = act @ W.T
        for wave in (0, 3, 7):
            got = acc_to_matrix(run_gemm(st[wave], act, wave, K_ // 32), 64)
            assert np.array_equal(got, want[:, 64 * wave:64 * wave + 64])
    # linear1 chunk (wave: 32 columns of a 256-column chunk) and linear2 k-slice
    W1 = rng.integers(-3, 4, (1024, 512)).astype(np.float64)
    act = rng.integers(-3, 4, (64, 512)).astype(np.float64)
    s1 = E._stages_ff1(torch.from_numpy(W1)).numpy()
    assert s1.shape == (4, 8, 8, 2048)
    want = act @ W1.T
    for ch, wave in ((0, 0), (2, 5), (3, 7)):
        got = acc_to_matrix(run_gemm(s1[ch, wave], act, wave, 8, ntiles=2, k2=2), 32)
        assert np.array_equal(got, want[:, 256 * ch + 32 * wave:256 * ch + 32 * wave + 32])
    W2 = rng.integers(-3, 4, (512, 1024)).astype(np.float64)
    h1 = rng.integers(-3, 4, (64, 1024)).astype(np.float64)
    s2 = E._stages_ff2(torch.from_numpy(W2)).numpy()
    assert s2.shape == (4, 8, 8, 2048)
    want = h1 @ W2.T
    for wave in (1, 6):
        tot = np.zeros((64, 64))
        for ch in range(4):
            tot += acc_to_matrix(run_gemm(s2[ch, wave], h1[:, 256 * ch:256 * ch + 256], wave, 8), 64)
        assert np.array_equal(tot, want[:, 64 * wave:64 * wave + 64])


def test_four_wave_stream_images_compute_the_gemms():
    """the 4-wave form (tcdiff_chain_args.nw = 4: wave w owns 128 columns = 8 n-tiles, 8-KB stages)"""
    rng = np.random.default_rng(2)
    for K_ in (512, 1024):
        W = rng.integers(-3, 4, (512, K_)).astype(np.float64)
        act = rng.integers(-3, 4, (64, K_)).astype(np.float64)
        st = E._stages_n512(torch.from_numpy(W), 4).numpy()
        assert st.shape == (4, K_ // 32, 4096)
        want = act @ W.T
        for wave in (0, 3):
            got = acc_to_matrix(run_gemm(st[wave], act, wave, K_ // 32, ntiles=8), 128)
            assert np.array_equal(got, want[:, 128 * wave:128 * wave + 128])
    W1 = rng.integers(-3, 4, (1024, 512)).astype(np.float64)
    act = rng.integers(-3, 4, (64, 512)).astype(np.float64)
    s1 = E._stages_ff1(torch.from_numpy(W1), 4).numpy()
    assert s1.shape == (4, 4, 8, 4096)
    want = act @ W1.T
    for ch, wave in ((0, 0), (2, 1), (3, 3)):
        got = acc_to_matrix(run_gemm(s1[ch, wave], act, wave, 8, ntiles=4, k2=2), 64)
        assert np.array_equal(got, want[:, 256 * ch + 64 * wave:256 * ch + 64 * wave + 64])
    W2 = rng.integers(-3, 4, (512, 1024)).astype(np.float64)
    h1 = rng.integers(-3, 4, (64, 1024)).astype(np.float64)
    s2 = E._stages_ff2(torch.from_numpy(W2), 4).numpy()
    assert s2.shape == (4, 4, 8, 4096)
    want = h1 @ W2.T
    for wave in (1, 2):
        tot = np.zeros((64, 128))
        for ch in range(4):
            tot += acc_to_matrix(run_gemm(s2[ch, wave], h1[:, 256 * ch:256 * ch + 256], wave, 8, ntiles=8), 128)
        assert np.array_equal(tot, want[:, 128 * wave:128 * wave + 128])


def kf_index(key, d):
    kt, k32, d32 = key >> 5, key & 31, d & 31
    g, jj = (d32 & 15) >> 2, 4 * (d32 >> 4) + (d32 & 3)
    return (((kt * 2 + (k32 >> 4)) * 2 + (d >> 5)) * 64 + g * 16 + (k32 & 15)) * 8 + jj


def vf_index(key, d):
    kt, k32 = key >> 5, key & 31
    g, jj = (k32 & 15) >> 2, 4 * (k32 >> 4) + (k32 & 3)
    return ((kt * 4 + (d >> 4)) * 64 + g * 16 + (d & 15)) * 8 + jj


def test_kv_fragment_images_compute_the_cross_attention_products():
    """S^T = K Q^T with Q^T taken from the projection's accumulator tiles, O^T = V^T P^T with P^T from the score tiles: the
    index maps of csrc/ops.hip (restated above) against plain matrix products, one 32-key tile."""
    rng = np.random.default_rng(1)
    Kt = rng.integers(-3, 4, (32, 64)).astype(np.float64)       # [key][d]
    Vt = rng.integers(-3, 4, (32, 64)).astype(np.float64)
    Q = rng.integers(-3, 4, (16, 64)).astype(np.float64)        # one 16-row tile [m][d]
    Kf, Vf = np.zeros(4096 // 2), np.zeros(4096 // 2)
    idx = set()
    for key in range(32):
        for d in range(64):
            Kf[kf_index(key, d)] = Kt[key, d]
            Vf[vf_index(key, d)] = Vt[key, d]
            idx.add(kf_index(key, d))
    assert len(idx) == 2048 and len({vf_index(k, d) for k in range(32) for d in range(64)}) == 2048
    Kf, Vf = Kf.reshape(4, 64, 8), Vf.reshape(4, 64, 8)
    # the w_qs accumulators of this row tile: qacc[nt][lane][r] = Q[m = c][d = 16 nt + 4 g + r]
    qacc = np.zeros((4, 64, 4))
    for nt in range(4):
        for l in range(64):
            qacc[nt, l] = Q[l & 15, 16 * nt + 4 * (l >> 4):16 * nt + 4 * (l >> 4) + 4]
    qf = [np.concatenate([qacc[2 * s], qacc[2 * s + 1]], axis=1) for s in range(2)]      # [lane][8]: lo tile, hi tile
    s0 = mfma16(Kf[1], qf[1], mfma16(Kf[0], qf[0], np.zeros((64, 4))))
    s1 = mfma16(Kf[3], qf[1], mfma16(Kf[2], qf[0], np.zeros((64, 4))))
    S = Q @ Kt.T                                                                       # [m][key]
    for l in range(64):
        c, g = l & 15, l >> 4
        assert np.array_equal(s0[l], S[c, 4 * g:4 * g + 4]) and np.array_equal(s1[l], S[c, 16 + 4 * g:16 + 4 * g + 4])
    pf = np.concatenate([s0, s1], axis=1)                                              # P^T B operand: [lane][8]
    O = S @ Vt                                                                         # [m][d]
    for dt in range(4):
        o = mfma16(Vf[dt], pf, np.zeros((64, 4)))
        for l in range(64):
            c, g = l & 15, l >> 4
            assert np.array_equal(o[l], O[c, 16 * dt + 4 * g:16 * dt + 4 * g + 4])


def test_no_inline_asm_vector_instruction_reads_kernel_registers():
    """Round 6 (profiles/r06_attention_pipeline.txt, section 3): on gfx950 a VALU read of an MFMA result needs software wait states
    that hipcc's hazard pass inserts for instructions it KNOWS -- an `asm("v_max3_f32 ..")` got none and read stale accumulators as
    soon as it stood close to the MFMAs.  The kernels may contain scalar / wait / LDS-DMA asm statements (no VGPR results of matrix
    instructions involved); a vector ALU instruction in an asm string is refused here."""
    import glob
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tcdiff_amd", "csrc")
    bad = []
    for f in sorted(glob.glob(os.path.join(root, "*.h")) + glob.glob(os.path.join(root, "*.hip"))):
        src = open(f).read()
        for m in re.finditer(r'\basm\s*(?:volatile)?\s*\(\s*((?:"[^"]*"\s*)+)', src):
            text = "".join(re.findall(r'"([^"]*)"', m.group(1)))
            if re.search(r"\bv_(?!readfirstlane)", text):           # any VALU / MFMA mnemonic
                line = src.count("\n", 0, m.start()) + 1
                if not src.splitlines()[line - 1].lstrip().startswith("//"):
                    bad.append(f"{os.path.basename(f)}:{line}: {text[:60]}")
    assert not bad, bad
