// The decoder layer for SMALL jobs (bf16, gfx950): four workgroups per 16-row block, four launches per layer.
//
// A row block streams a layer's 5.5 MB of weights through ONE CU's vector-memory port whatever its rows (chain.hip: ~48 us at
// 115 GB/s), so a job of a few hundred rows -- one clip, what TCDiff.py renders (TCDiff.py:292-303, model/diffusion.py:386-442) --
// leaves 200 CUs idle while 58 sixteen-row blocks each take ~75 us per layer.  Here FOUR workgroups ("members" c = 0..3, one CU
// each) share a block and each streams about a third of the weights; the partition ALTERNATES so that only three exchanges per
// layer cross the CUs, and the exchanges are launch boundaries (no in-kernel hand-off, nothing to hang):
//
//   part 1  self-attention of heads 2c, 2c+1 (each head's key tiles over 4 waves, softmax partials merged through LDS)
//           -> fc over the member's OWN 128 contraction columns (K split)                       -> partial sums P[block][c]
//   part 2  z = sum_c P; LayerNorm(1e-6), FiLM, +x -> x; norm2, rotary; w_qs for heads 2c, 2c+1 (N split, K over 4 waves);
//           cross-attention of those heads; fc over the member's 128 columns (K split)          -> P
//   part 3  z = sum_c P; LN, FiLM, +x -> x; norm3; linear1 rows 256c.. (N split), GELU, linear2 over those 256 (K split)  -> P
//   part 4  z = sum_c P; FiLM, +x; norm4; linear3 (every member, all of it: 0.5 MB, no exchange) -> x'; norm1', rotary;
//           w_qs / w_ks / w_vs of heads 2c, 2c+1 -> the next layer's Q / K / V fragment images (chain.hip's formats)
//
//   part 0  (once per forward, TC_CHAIN_FRONT stream) layer 0's norm1, rotary, w_qs / w_ks / w_vs of heads 2c, 2c+1 from the token
//           rows the fusion projection left -> layer 0's fragment images: its self-attention then runs in part 1 like every layer's
//
// Measured (profiles/r06_split_stamps.txt, r06_flag_sync_probe.txt): parts 1-4 = 9 / 16 / 14 / 19 us against 75
// for the fused 16-row launch; in parts 2-4 the first ~6 us are the exchange itself (kernel arguments, then partial slabs and residual
// rows from memory), the rest runs at the CU's ingest rate -- and an exchange through flags inside one launch costs MORE than a
// launch boundary on this machine (7-9 us with scope bits on the accesses, 33 us with fences, against 5.9).
// (model/model.py:97-107,323-344,374-401 as chain.hip.)  A member streams 0.13 + 0.26 + 0.5 + 0.9 MB instead of 5.5; every
// member redoes the row-local LayerNorm / FiLM / residual of the 16 rows (nothing) and stores its quarter of x.  The weights are
// chain.hip's per-wave streams, unchanged: a K split is a stage range of a wave's phase, an N split is the stream of the wave
// that owns those columns.  Row blocks are cut per sequence as in the fused launch (tcdiff_chain_args.seq_blocks), MT = 1.
// Results equal the fused launch's up to fp32 summation order (four partial sums instead of one chain of 16 k-steps).
// The residual stream is NOT updated in place here: every member reads whole rows of x and stores a quarter, so a part reads `xres`
// and writes `xout`, two different buffers (the caller alternates them: parts 2, 3, 4 each flip).
#include "common.h"
#include "tcdiff_hip.h"

#include "chain_core.h"

#define CS_NM 4                        // members per block
// LDS (the fused kernel's map, 64-row geometry: a 16-row block uses rows 0..15 of every 8-KB k-tile)
#define CS_RED(w) (CH_ABUF + (w) * 8192 + 2048)      // 4 KB per wave inside the UNUSED rows 16..47 of k-tile w: cross-wave reductions

#ifdef CS_STAMP   // diagnostic build: s_memrealtime (100 MHz) of every wave of block 0 / member CS_STAMP_MEMBER at the marked points, into h_out
#ifndef CS_STAMP_MEMBER
#define CS_STAMP_MEMBER 0
#endif
#define CS_T(part, i) do { if (g.lblk == 0 && g.member == CS_STAMP_MEMBER && (threadIdx.x & 63) == 0 && a.h_out) \
        reinterpret_cast<unsigned long long*>(a.h_out)[((part) - 1) * 128 + (threadIdx.x >> 6) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CS_T(part, i) do { } while (0)
#endif

struct SplitGeo {
    int lblk, member, bseq, bis, m0, Mend, L, Mtot;
};
DEVINL SplitGeo split_geo(const tcdiff_chain_args& a) {
    SplitGeo g;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);      // consecutive logical ids share an XCD: a block's four members do
    g.lblk = logical / CS_NM;
    g.member = logical % CS_NM;
    const int nbs = (a.L + 15) / 16;
    g.bseq = g.lblk / nbs;
    g.bis = g.lblk - g.bseq * nbs;
    g.L = a.L;
    g.Mtot = a.M;
    g.m0 = g.bseq * a.L + g.bis * 16;
    g.Mend = (g.bseq + 1) * a.L;                                // first row past the block's sequence: rows beyond recompute row Mend - 1
    return g;
}
DEVINL int my_row(const SplitGeo& g, int c) {
    const int m = g.m0 + c;
    return m < g.Mend ? m : g.Mend - 1;
}
// column-blocked fp32 [rows][512]: element (row, col) at ((col / 8) * rows + row) * 8 + col % 8
DEVINL long cb_index(long rows, int row, int col) { return ((long)(col >> 3) * rows + row) * 8 + (col & 7); }

// acc[nt] (columns 64 wave + 16 nt + 4 g .. of row c) <-> the member's partial-sum slab [16 rows][512] fp32
DEVINL void partial_store(const f32x4_t (&acc)[4][1], float* P, const SplitGeo& g, int wave, int lane) {
    const int c = lane & 15, gg = lane >> 4;
    float* dst = P + ((long)(g.lblk * CS_NM + g.member) * 16 + c) * 512 + 64 * wave + 4 * gg;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4_t*>(dst + 16 * nt) = acc[nt][0];
}
DEVINL void partial_sum(f32x4_t (&acc)[4][1], const float* P, const SplitGeo& g, int wave, int lane) {
    const int c = lane & 15, gg = lane >> 4;
    const float* src = P + ((long)(g.lblk * CS_NM) * 16 + c) * 512 + 64 * wave + 4 * gg;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        f32x4_t s = ld4(src + 16 * nt);
#pragma unroll
        for (int m = 1; m < CS_NM; ++m) s += ld4(src + (long)m * 16 * 512 + 16 * nt);      // fixed order: deterministic
        acc[nt][0] = s;
    }
}

// a wave's stream positioned at `stage` with the next CH_D stages loaded
DEVINL void stream_at(WStream& ws, const tcdiff_chain_args& a, int stream_wave, unsigned stage, int lane) {
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.wstream)) + (long)stream_wave * a.n_stages * CH_STAGE, 0,
        a.n_stages * CH_STAGE, 0x00020000);
    ws.voff = (unsigned)lane * 16u;
    ws.pos = stage;
    ws.last = (unsigned)a.n_stages - 1;
#pragma unroll
    for (int i = 0; i < CH_D; ++i) ws_load(ws, i, stage + (unsigned)i);
}

// These kernels are chains of short dependent steps, so every global load whose address does not depend on an earlier step is ISSUED
// AT THE TOP of the kernel (FiLM rows, residual rows, norm weights, rotary rows, biases, the weight-stream rings of the later phases)
// and used where it is needed: a 16-row block has the registers for it, and one exposed L2 round trip is ~1.5 us of a ~10-us kernel.
// Vectors indexed by the column only (FiLM rows -- a block lies in one sequence --, LayerNorm weights, biases) would cost a full
// 1-KB wave load per 16 bytes a lane needs (lanes that differ in the row ask for the same address, and the address unit is paid per
// lane): lanes 0..15 of a wave fetch the wave's 64 columns once (256 B), park them in the wave's own LDS slot and every lane reads
// its chunk back.  No barrier: a wave reads what it wrote.  Slots: the FiLM / vector area the fused launch stages to (CH_FILM, CH_VEC).
DEVINL f32x4_t col_fetch(const float* vec, int wave, int lane) {
    f32x4_t v = {0, 0, 0, 0};
    if (lane < 16) v = ld4(vec + 64 * wave + 4 * lane);
    return v;
}
DEVINL void col_park(char* smem, int slot, f32x4_t v, int wave, int lane) {
    if (lane < 16) *reinterpret_cast<f32x4_t*>(smem + slot * 2048 + wave * 256 + lane * 16) = v;
}
DEVINL f32x4_t col_get(const char* smem, int slot, int wave, int nt, int gg) {
    return *reinterpret_cast<const f32x4_t*>(smem + slot * 2048 + wave * 256 + (4 * nt + gg) * 16);
}

struct RowC { f32x4_t G, Bv, x[4]; };             // the block's FiLM row (pre-folded; column chunks) and the residual rows
DEVINL RowC row_consts(const tcdiff_chain_args& a, const SplitGeo& g, const float* film, const float* xin, long xin_rows, int xin_mod,
                       bool xin_rowmajor, int wave, int lane) {
    RowC r;
    const int c = lane & 15, gg = lane >> 4;
    const int row = my_row(g, c);
    const float* fr = film + (long)g.bseq * a.film_ld;
    const int rin = xin_mod > 0 ? row % xin_mod : row;
    r.G = col_fetch(fr, wave, lane);
    r.Bv = col_fetch(fr + 512, wave, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = 64 * wave + 16 * nt + 4 * gg;
        r.x[nt] = xin_rowmajor ? ld4(xin + (long)rin * 512 + col) : ld4(xin + cb_index(xin_rows, rin, col));
    }
    return r;
}
struct NormC { f32x4_t g, b, r[4]; };             // a LayerNorm's weights (column chunks) and (ROT) the rows' rotary table entries
template <bool ROT>
DEVINL NormC norm_consts(const tcdiff_chain_args& a, const SplitGeo& g, const float* ng, const float* nb, int wave, int lane) {
    NormC n;
    const int c = lane & 15, gg = lane >> 4;
    const int row = my_row(g, c);
    const int pos = row - (row / g.L) * g.L;
    n.g = col_fetch(ng, wave, lane);
    n.b = col_fetch(nb, wave, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
        n.r[nt] = ROT ? ld4(a.rope + cb_index(a.rope_rows, pos, 64 * wave + 16 * nt + 4 * gg)) : f32x4_t{1, 0, 1, 0};       // cos0 sin0 cos1 sin1
    return n;
}

// z (accumulators) -> x_new = x + LN_eps(z) G + Bv (LN = false: x + z G + Bv), stored by the member that owns the columns;
// returns with acc = x_new
template <bool LN>
DEVINL void block_epilogue(f32x4_t (&acc)[4][1], const RowC& rc, const SplitGeo& g, float eps, float* xout, float* scr, char* smem,
                           int wave, int lane, bool store) {
    float nmr[1] = {0.0f}, rstd[1] = {1.0f};
    col_park(smem, 0, rc.G, wave, lane);
    col_park(smem, 1, rc.Bv, wave, lane);
    if (LN) row_stats<1, 4>(acc, scr, wave, lane, eps, nmr, rstd);
    const int c = lane & 15, gg = lane >> 4;
    const int row = my_row(g, c);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = 64 * wave + 16 * nt + 4 * gg;
        const f32x4_t G = col_get(smem, 0, wave, nt, gg), Bv = col_get(smem, 1, wave, nt, gg);
        f32x4_t o;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float u = LN ? fmaf(acc[nt][0][t], rstd[0], nmr[0]) : acc[nt][0][t];
            o[t] = rc.x[nt][t] + fmaf(u, G[t], Bv[t]);
        }
        acc[nt][0] = o;
        if (store && xout) *reinterpret_cast<f32x4_t*>(xout + cb_index(g.Mtot, row, col)) = o;
    }
}

// LayerNorm(acc) (optionally rotated) -> bf16 activation block in LDS (k-tile = wave); `plain`: the un-rotated image too
template <bool ROT>
DEVINL void norm_lds(const f32x4_t (&acc)[4][1], const NormC& nc, float eps, float* scr, char* smem, char* abuf, char* plain, int wave,
                     int lane) {
    float nmr[1], rstd[1];
    col_park(smem, 2, nc.g, wave, lane);
    col_park(smem, 3, nc.b, wave, lane);
    row_stats<1, 4>(acc, scr, wave, lane, eps, nmr, rstd);
    const int gg = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const f32x4_t wg = col_get(smem, 2, wave, nt, gg), wb = col_get(smem, 3, wave, nt, gg);
        float u[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) u[t] = fmaf(fmaf(acc[nt][0][t], rstd[0], nmr[0]), wg[t], wb[t]);
        const int wo = wave * 8192 + act_wr_off(lane, nt);
        if (plain) {
            uint2 pk = {pack_bf2(u[0], u[1]), pack_bf2(u[2], u[3])};
            *reinterpret_cast<uint2*>(plain + wo) = pk;
        }
        if (ROT) {
            const f32x4_t q = nc.r[nt];
            const float y0 = u[0] * q[0] - u[1] * q[1], y1 = u[1] * q[0] + u[0] * q[1];
            const float y2 = u[2] * q[2] - u[3] * q[3], y3 = u[3] * q[2] + u[2] * q[3];
            u[0] = y0; u[1] = y1; u[2] = y2; u[3] = y3;
        }
        uint2 pk = {pack_bf2(u[0], u[1]), pack_bf2(u[2], u[3])};
        *reinterpret_cast<uint2*>(abuf + wo) = pk;
    }
}

// The 512 x 128 slice of a projection that belongs to head (2 member + wave / 4): the head's 64 output columns, its 16-stage phase
// `phase0` cut in four k-quarters over the head's four waves (j = wave % 4: stages phase0 + 4 j .., issued by head_stream -- early),
// partial tiles summed through LDS; afterwards EVERY wave of the head holds the full tile.  SWAP: transposed tiles (store_vfrag).
// Two barriers.
DEVINL void head_stream(WStream& ws, const tcdiff_chain_args& a, const SplitGeo& g, unsigned phase0, int wave, int lane) {
    stream_at(ws, a, 2 * g.member + (wave >> 2), phase0 + 4u * (wave & 3), lane);
}
template <bool SWAP>
DEVINL void head_projection(f32x4_t (&acc)[4][1], WStream& ws, const char* act, char* smem, int wave, int lane) {
    const int i = wave >> 2, j = wave & 3;
    zero(acc);
    phase_n512<4, true, 1, 4, SWAP>(acc, act + 2 * j * 8192, ws, lane);
    f32x4_t* red = reinterpret_cast<f32x4_t*>(smem + CS_RED(wave));
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) red[nt * 64 + lane] = acc[nt][0];
    lds_barrier();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        f32x4_t s = reinterpret_cast<const f32x4_t*>(smem + CS_RED(4 * i))[nt * 64 + lane];
#pragma unroll
        for (int jj = 1; jj < 4; ++jj) s += reinterpret_cast<const f32x4_t*>(smem + CS_RED(4 * i + jj))[nt * 64 + lane];
        acc[nt][0] = s;
    }
    lds_barrier();      // the reduction area is free again
}

// A 16-stage product that EVERY member (and every block) computes in full -- linear3, the merged form's fc -- is the same 512 KB read
// by all workgroups of the launch at the same time: 88 GB/s per CU against 116 for the member-private streams of part 3
// (profiles/r06_split_stamps.txt).  The four k-quarters commute, so a workgroup starts at quarter (block + member) % 4: four
// phases of readers instead of one.  Two rings alternate (a quarter's ring is refilled with the quarter after next while the other is
// consumed); `first` holds quarter q(0) already (issued at the top of the launch).  fp32 summation order differs between members by
// the rotation (each member uses its own copy of the result: ~1e-7, far below the bf16 operands that follow).
DEVINL unsigned quarter_of(int i, int rot) { return (unsigned)((i + rot) & 3); }
template <class AfterThird, class AfterFourth>
DEVINL void rotated_product16(f32x4_t (&acc)[4][1], const char* act, const tcdiff_chain_args& a, int stream_wave, unsigned stage0, int rot,
                              WStream& first, WStream& second, int lane, AfterThird&& after_third, AfterFourth&& after_fourth) {
    phase_n512<4, true, 1>(acc, act + quarter_of(0, rot) * 16384u, first, lane);
    stream_at(first, a, stream_wave, stage0 + 4u * quarter_of(2, rot), lane);
    phase_n512<4, true, 1>(acc, act + quarter_of(1, rot) * 16384u, second, lane);
    stream_at(second, a, stream_wave, stage0 + 4u * quarter_of(3, rot), lane);
    phase_n512<4, true, 1>(acc, act + quarter_of(2, rot) * 16384u, first, lane);
    after_third();                                          // `first` is free
    phase_n512<4, true, 1>(acc, act + quarter_of(3, rot) * 16384u, second, lane);
    after_fourth();                                         // `second` is free
}

// Two projections of the same activation block (the next layer's Q and K) with ONE exchange: the second reduction area sits in the
// unused rows of the other activation buffer's k-tiles.
#define CS_RED2(w) (CH_ABUF2 + (w) * 8192 + 2048)
DEVINL void head_projection2(f32x4_t (&a0)[4][1], f32x4_t (&a1)[4][1], WStream& w0, WStream& w1, const char* act, char* smem, int wave,
                             int lane) {
    const int i = wave >> 2, j = wave & 3;
    zero(a0);
    zero(a1);
    phase_n512<4, true, 1, 4, false>(a0, act + 2 * j * 8192, w0, lane);
    phase_n512<4, true, 1, 4, false>(a1, act + 2 * j * 8192, w1, lane);
    f32x4_t* r0 = reinterpret_cast<f32x4_t*>(smem + CS_RED(wave));
    f32x4_t* r1 = reinterpret_cast<f32x4_t*>(smem + CS_RED2(wave));
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        r0[nt * 64 + lane] = a0[nt][0];
        r1[nt * 64 + lane] = a1[nt][0];
    }
    lds_barrier();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        f32x4_t s0 = reinterpret_cast<const f32x4_t*>(smem + CS_RED(4 * i))[nt * 64 + lane];
        f32x4_t s1 = reinterpret_cast<const f32x4_t*>(smem + CS_RED2(4 * i))[nt * 64 + lane];
#pragma unroll
        for (int jj = 1; jj < 4; ++jj) {
            s0 += reinterpret_cast<const f32x4_t*>(smem + CS_RED(4 * i + jj))[nt * 64 + lane];
            s1 += reinterpret_cast<const f32x4_t*>(smem + CS_RED2(4 * i + jj))[nt * 64 + lane];
        }
        a0[nt][0] = s0;
        a1[nt][0] = s1;
    }
    lds_barrier();      // the reduction areas are free again
}

// softmax(q k^T) v of the block's 16 rows for head (2 member + wave / 4), its 32-key tiles dealt to the head's four waves (tile kt
// on wave kt % 4), each wave an online softmax as chain.hip's cross_attention (exp2 domain, lazy running maximum, row sums by an
// all-ones MFMA), the four partial results merged through LDS (flash-decoding: rescale to the common maximum).  O (bf16) -> the
// member's activation block, k-tile wave / 4.  qf: the head's Q^T fragments (scaled by log2 e / sqrt d_k).  Two barriers.
struct KVTiles { u32x4 k0[4], v0[4], k1[4], v1[4]; };      // a wave's first two key tiles (j, j + 4), issued ahead of the attention
DEVINL void kv_tile(const __amdgpu_buffer_rsrc_t& r, unsigned image_off, int kt, int nkt, int lane, u32x4 (&f)[4]) {
    const unsigned so = image_off + (unsigned)(kt < nkt ? kt : nkt - 1) * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)lane * 16u + 1024u * q, so, 0));
}
// WPH: waves per head -- 4 (a workgroup runs two heads: wave w is head w / 4, key tiles w % 4, w % 4 + 4, ..) or 1 (eight heads, a
// wave runs all key tiles of head w: no merge)
// WPH = 1 walks the tiles in a ROTATED order, start tile `rot0` % nkt (the online softmax does not care): the workgroups of a
// sequence -- all of which read the same K / V -- then do not all ask for the same lines at the same time.
DEVINL int kt_wrap(int kt, int nkt) { return kt >= nkt ? kt - nkt : kt; }
template <int WPH = 4>
DEVINL void kv_issue(KVTiles& t, const void* kf, const void* vf, unsigned image_off, int nkt, int wave, int lane, int rot0 = 0) {
    const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(kf), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(vf), 0, -1, 0x00020000);
    const int j = WPH == 4 ? wave & 3 : rot0 % nkt;
    const int j1 = WPH == 4 ? j + 4 : kt_wrap(j + 1, nkt);
    kv_tile(kr, image_off, j, nkt, lane, t.k0);
    kv_tile(vr, image_off, j, nkt, lane, t.v0);
    kv_tile(kr, image_off, j1, nkt, lane, t.k1);
    kv_tile(vr, image_off, j1, nkt, lane, t.v1);
}
template <int WPH = 4>
DEVINL void head_attention(const u32x4 (&qf)[2], KVTiles& pre, const void* kf, const void* vf, unsigned image_off, int nkt, int Lk,
                           char* smem, int wave, int lane, int rot0 = 0) {
    const int i = WPH == 4 ? wave >> 2 : wave, j = WPH == 4 ? wave & 3 : 0, c = lane & 15, gg = lane >> 4;
    u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    asm volatile("" : "+v"(ones));
    const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(kf), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(vf), 0, -1, 0x00020000);
    auto ld_tile = [&](const __amdgpu_buffer_rsrc_t& r, int kt, u32x4 (&f)[4]) { kv_tile(r, image_off, kt, nkt, lane, f); };
    f32x4_t o[4], lacc = {0, 0, 0, 0};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0, 0, 0, 0};
    float m_run = -INFINITY, nb = 0.0f;
    u32x4 (&kn)[4] = pre.k0, (&vn)[4] = pre.v0, (&kn2)[4] = pre.k1, (&vn2)[4] = pre.v1;      // the wave's tiles run TWO ahead (1-4 tiles of a
                                                                                               // 450-key sequence per wave: latency, not rate)
    const int count = WPH == 4 ? (nkt - j + 3) / 4 : nkt;       // tiles of this wave (none: j >= nkt)
    int kt = WPH == 4 ? j : rot0 % nkt;
#pragma unroll 1
    for (int it = 0; it < count; ++it, kt = WPH == 4 ? kt + 4 : kt_wrap(kt + 1, nkt)) {
        u32x4 kc[4], vc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            kc[q] = kn[q];
            vc[q] = vn[q];
            kn[q] = kn2[q];
            vn[q] = vn2[q];
        }
        const int k2 = WPH == 4 ? kt + 8 : kt_wrap(kt_wrap(kt + 1, nkt) + 1, nkt);     // (WPH = 4 past the end: the last tile again, unused)
        ld_tile(kr, k2, kn2);
        ld_tile(vr, k2, vn2);
        f32x4_t s0 = {nb, nb, nb, nb}, s1 = s0;
        mma16(s0, kc[0], qf[0]);
        mma16(s1, kc[2], qf[0]);
        mma16(s0, kc[1], qf[1]);
        mma16(s1, kc[3], qf[1]);
        if (kt * 32 + 32 > Lk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (kt * 32 + 4 * gg + t >= Lk) s0[t] = -INFINITY;
                if (kt * 32 + 16 + 4 * gg + t >= Lk) s1[t] = -INFINITY;
            }
        }
        const float lm = lane_max8(s0, s1, -INFINITY);
        const bool first = it == 0;
        if (__builtin_amdgcn_ballot_w64(first || lm > CH_ATT_THR) != 0) {
            const float mx = ar4_max(lm);                                   // relative to -nb
            // (a wave's first tile may hold masked keys only for some row?  no: a tile has >= 1 valid key and every row sees every key)
            const float m_new = (first || mx > CH_ATT_THR) ? fmaxf(m_run, mx - nb) : m_run;
            const float shift = m_new + nb;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // first tile: exp2(-inf) = 0, o and lacc are 0
            m_run = m_new;
            nb = -m_new;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                lacc[t] *= alpha;
                s0[t] -= shift;
                s1[t] -= shift;
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int t = 0; t < 4; ++t) o[dt][t] *= alpha;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s0[t] = __builtin_amdgcn_exp2f(s0[t]);
            s1[t] = __builtin_amdgcn_exp2f(s1[t]);
        }
        const u32x4 pf = {pack_bf2(s0[0], s0[1]), pack_bf2(s0[2], s0[3]), pack_bf2(s1[0], s1[1]), pack_bf2(s1[2], s1[3])};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) mma16(o[dt], vc[dt], pf);
        mma16(lacc, ones, pf);
    }
    if (WPH == 1) {      // the wave has seen every key: normalise and write the head's 64 columns (k-tile `wave`) of the activation block
        const float inv1 = __builtin_amdgcn_rcpf(lacc[0]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            uint2 pk = {pack_bf2(o[dt][0] * inv1, o[dt][1] * inv1), pack_bf2(o[dt][2] * inv1, o[dt][3] * inv1)};
            *reinterpret_cast<uint2*>(smem + CH_ABUF + i * 8192 + act_wr_off(lane, dt)) = pk;
        }
        return;
    }
    // ---- merge the head's four partial softmaxes: per query (lane c) the maximum m, the sum l, O^T[feature 16 dt + 4 g + t][query c]
    float* red = reinterpret_cast<float*>(smem + CS_RED(wave));     // [64 features][16 queries] | m[16] at 1024 | l[16] at 1040 (floats)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int t = 0; t < 4; ++t) red[(16 * dt + 4 * gg + t) * 16 + c] = o[dt][t];
    if (gg == 0) {
        red[1024 + c] = m_run;
        red[1040 + c] = lacc[0];
    }
    lds_barrier();
    float mw[4], M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        mw[w] = reinterpret_cast<const float*>(smem + CS_RED(4 * i + w))[1024 + c];
        M = fmaxf(M, mw[w]);
    }
    float lsum = 0.0f, ov[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float* rw = reinterpret_cast<const float*>(smem + CS_RED(4 * i + w));
        const float sc = mw[w] == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(mw[w] - M);      // a wave without tiles: weight 0
        lsum = fmaf(rw[1040 + c], sc, lsum);
#pragma unroll
        for (int t = 0; t < 4; ++t) ov[t] = fmaf(rw[(16 * j + 4 * gg + t) * 16 + c], sc, ov[t]);     // this wave finishes d tile j
    }
    const float inv = __builtin_amdgcn_rcpf(lsum);
    lds_barrier();      // every wave has read the partials: the area (and k-tile i's rows 0..15 below) may be rewritten
    uint2 pk = {pack_bf2(ov[0] * inv, ov[1] * inv), pack_bf2(ov[2] * inv, ov[3] * inv)};
    *reinterpret_cast<uint2*>(smem + CH_ABUF + i * 8192 + act_wr_off(lane, j)) = pk;
}

// accumulator tiles of a head's Q^T -> the score MFMA's B operands (chain.hip cross_attention)
DEVINL void q_fragments(const f32x4_t (&qacc)[4][1], float qs, u32x4 (&qf)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f32x4_t lo = qacc[2 * s][0], hi = qacc[2 * s + 1][0];
        qf[s][0] = pack_bf2(lo[0] * qs, lo[1] * qs);
        qf[s][1] = pack_bf2(lo[2] * qs, lo[3] * qs);
        qf[s][2] = pack_bf2(hi[0] * qs, hi[1] * qs);
        qf[s][3] = pack_bf2(hi[2] * qs, hi[3] * qs);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// part 0 (TC_CHAIN_FRONT): layer 0's norm1 + rotary -> Q, K ; norm1 -> V of the token rows the fusion projection left (row-major fp32:
// model/model.py:561 seen per token), heads 2 member, 2 member + 1, as the fragment images part 1 of layer 0 reads.  The weights: the
// w_qs / w_ks / w_vs phases at the end of the TC_CHAIN_FRONT stream (stages 32 / 48 / 64 of its 80; any dancer's copy).
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void chain_split0_kernel(tcdiff_chain_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeo g = split_geo(a);
    char* abuf = smem + CH_ABUF;
    float* scr = reinterpret_cast<float*>(smem + CH_SCR);
    f32x4_t acc[4][1];
    {
        const int c = lane & 15, gg = lane >> 4;
        const float* src = a.xres + (long)my_row(g, c) * 512 + 64 * wave + 4 * gg;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt][0] = ld4(src + 16 * nt);
    }
    const NormC nn = norm_consts<true>(a, g, a.nn_g, a.nn_b, wave, lane);
    WStream wq, wk;
    head_stream(wq, a, g, 32u, wave, lane);
    head_stream(wk, a, g, 48u, wave, lane);
    norm_lds<true>(acc, nn, a.nn_eps, scr, smem, abuf, smem + CH_ABUF2, wave, lane);
    lds_barrier();
    const int head = 2 * g.member + (wave >> 2);
    const bool writer = (wave & 3) == 0;
    f32x4_t t[4][1], tk[4][1];
    head_projection2(t, tk, wq, wk, abuf, smem, wave, lane);
    WStream wv;
    head_stream(wv, a, g, 64u, wave, lane);
    if (writer) store_qfrag<1>(t, a.qf_out, a.scale_q, g.lblk, head, lane);
    if (writer) store_kfrag<1>(tk, a.kf_out, g.bseq, g.bis * 16, a.out_nkt, head, lane);
    head_projection<true>(t, wv, smem + CH_ABUF2, smem, wave, lane);
    if (writer) store_vfrag<1>(t, a.vf_out, g.bseq, g.bis * 16, a.out_nkt, a.L, head, lane);
}

// ------------------------------------------------------------------------------------------------------------------------------
// part 1: self-attention (or the attention output rows of layer 0) -> fc, K split
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void chain_split1_kernel(tcdiff_chain_args a, float* p_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeo g = split_geo(a);
    char* abuf = smem + CH_ABUF;
    WStream ws;
    CS_T(1, 0);
    stream_at(ws, a, wave, 4u * g.member, lane);          // fc: the member's four k-steps of every wave's 16-stage phase
    if (a.sa_q) {
        const int head = 2 * g.member + (wave >> 2);
        // layer 0 under classifier-free guidance: the stacked branches share x, hence Q / K / V (a_mod rows of them exist)
        const int sseq = a.a_mod > 0 ? g.bseq % (a.a_mod / a.L) : g.bseq;
        const int qblk = sseq * ((a.L + 15) / 16) + g.bis;
        const u32x4* qsrc = reinterpret_cast<const u32x4*>(a.sa_q) + ((long)(qblk * 8 + head) * 8) * 64 + lane;
        const u32x4 qf[2] = {qsrc[0], qsrc[64]};
        const unsigned img = (unsigned)((sseq * a.H + head) * a.sa_nkt) * 4096u;
        KVTiles pre;
        kv_issue(pre, a.sa_kf, a.sa_vf, img, a.sa_nkt, wave, lane);
        head_attention(qf, pre, a.sa_kf, a.sa_vf, img, a.sa_nkt, a.L, smem, wave, lane);
    } else if (wave < 2) {
        // layer 0: O of the stand-alone attention launch, the member's two k-tiles (columns 128 member ..) of the 16 rows
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
            stage_glds<16, 2>(abuf + kt * 8192, reinterpret_cast<const char*>(a.A) + (2 * g.member + kt) * TC_ROWB, 1024, g.m0, g.Mend,
                              a.a_mod, wave, lane);
    }
    CS_T(1, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    CS_T(1, 2);
    f32x4_t acc[4][1];
    zero(acc);
    phase_n512<4, true, 1>(acc, abuf, ws, lane);
    CS_T(1, 3);
    partial_store(acc, p_out, g, wave, lane);
    CS_T(1, 4);
}

// ------------------------------------------------------------------------------------------------------------------------------
// parts 1 + 2 in ONE launch (part = 12): every member runs the self-attention of ALL eight heads of its 16 rows (one wave per head)
// and the whole fc -- 0.7 MB more K / V and 0.4 MB more weights per member than the split form -- and saves the exchange between them
// (a launch boundary: ~6 us of exchange + ~2.5 us of ramp + the gap).  Pays for short sequences (config 1: 120 keys) and about breaks
// even at 450; then as part 2.
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void chain_split12_kernel(tcdiff_chain_args a, float* p_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeo g = split_geo(a);
    char* abuf = smem + CH_ABUF;
    float* scr = reinterpret_cast<float*>(smem + CH_SCR);
    const bool mine = (wave >> 1) == g.member;
    const int rot = (g.lblk + g.member) & 3;               // fc of the self-attention block: all 16 stages, k-quarters rotated
    WStream ws, wq;
    stream_at(ws, a, wave, 4u * quarter_of(0, rot), lane);
    if (a.sa_q) {
        const int sseq = a.a_mod > 0 ? g.bseq % (a.a_mod / a.L) : g.bseq;      // (layer 0 under guidance: see part 1)
        const int qblk = sseq * ((a.L + 15) / 16) + g.bis;
        const u32x4* qsrc = reinterpret_cast<const u32x4*>(a.sa_q) + ((long)(qblk * 8 + wave) * 8) * 64 + lane;
        const u32x4 qf[2] = {qsrc[0], qsrc[64]};
        const unsigned img = (unsigned)((sseq * a.H + wave) * a.sa_nkt) * 4096u;
        KVTiles pre;
        const int rk = 5 * g.bis + 3 * g.member;           // start tile: spread over the sequence's blocks and members
        kv_issue<1>(pre, a.sa_kf, a.sa_vf, img, a.sa_nkt, wave, lane, rk);
        head_attention<1>(qf, pre, a.sa_kf, a.sa_vf, img, a.sa_nkt, a.L, smem, wave, lane, rk);
    } else {
        // layer 0 without the fragment front: O of the stand-alone attention launch, all eight k-tiles of the 16 rows
        stage_glds<16, 1>(abuf + wave * 8192, reinterpret_cast<const char*>(a.A) + wave * TC_ROWB, 1024, g.m0, g.Mend, a.a_mod, 0, lane);
    }
    const RowC rc = row_consts(a, g, a.film, a.xres, a.xres_mod > 0 ? a.xres_mod : a.M, a.xres_mod, a.xres_rowmajor != 0, wave, lane);
    const NormC nc = norm_consts<true>(a, g, a.n2_g, a.n2_b, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    stream_at(wq, a, wave, 4u * quarter_of(1, rot), lane);
    f32x4_t acc[4][1];
    zero(acc);
    rotated_product16(acc, abuf, a, wave, 0u, rot, ws, wq, lane, [] {}, [&] { head_stream(wq, a, g, 16u, wave, lane); });
    lds_barrier();                                         // every wave is out of fc: the activation block is rewritten below
    block_epilogue<true>(acc, rc, g, a.ln_eps, a.xout, scr, smem, wave, lane, mine);
    norm_lds<true>(acc, nc, a.n2_eps, scr + 1024, smem, abuf, nullptr, wave, lane);
    lds_barrier();
    f32x4_t qacc[4][1];
    const int head = 2 * g.member + (wave >> 2);
    const int kv = g.bseq < a.n_shared ? 0 : g.bseq - a.n_shared + (a.n_shared > 0 ? 1 : 0);
    const unsigned img = (unsigned)((kv * a.H + head) * a.nkt) * 4096u;
    KVTiles pre;
    kv_issue(pre, a.kf, a.vf, img, a.nkt, wave, lane);
    stream_at(ws, a, wave, 32u + 4u * g.member, lane);
    head_projection<false>(qacc, wq, abuf, smem, wave, lane);
    u32x4 qf[2];
    q_fragments(qacc, a.scale_q * CH_LOG2E, qf);
    head_attention(qf, pre, a.kf, a.vf, img, a.nkt, a.Lk, smem, wave, lane);
    lds_barrier();
    zero(acc);
    phase_n512<4, true, 1>(acc, abuf, ws, lane);
    partial_store(acc, p_out, g, wave, lane);
}

// ------------------------------------------------------------------------------------------------------------------------------
// part 2: self-attention block tail, cross-attention of the member's heads, fc K split
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void chain_split2_kernel(tcdiff_chain_args a, const float* p_in, float* p_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeo g = split_geo(a);
    char* abuf = smem + CH_ABUF;
    float* scr = reinterpret_cast<float*>(smem + CH_SCR);
    const bool mine = (wave >> 1) == g.member;            // this member stores columns 128 member .. of x
    f32x4_t acc[4][1];
    CS_T(2, 0);
    partial_sum(acc, p_in, g, wave, lane);
    const RowC rc = row_consts(a, g, a.film, a.xres, a.xres_mod > 0 ? a.xres_mod : a.M, a.xres_mod, a.xres_rowmajor != 0, wave, lane);
    const NormC nc = norm_consts<true>(a, g, a.n2_g, a.n2_b, wave, lane);
    WStream wq;
    head_stream(wq, a, g, 16u, wave, lane);               // w_qs (cross) of heads 2 member, 2 member + 1: stages 16.. of those waves' streams
    CS_T(2, 1);
    block_epilogue<true>(acc, rc, g, a.ln_eps, a.xout, scr, smem, wave, lane, mine);
    CS_T(2, 2);
    norm_lds<true>(acc, nc, a.n2_eps, scr + 1024, smem, abuf, nullptr, wave, lane);
    lds_barrier();
    CS_T(2, 3);
    // Q = rot(norm2 x) W_q^T (model/model.py:387,78)
    f32x4_t qacc[4][1];
    WStream ws;
    const int head = 2 * g.member + (wave >> 2);
    const int kv = g.bseq < a.n_shared ? 0 : g.bseq - a.n_shared + (a.n_shared > 0 ? 1 : 0);
    const unsigned img = (unsigned)((kv * a.H + head) * a.nkt) * 4096u;
    KVTiles pre;
    kv_issue(pre, a.kf, a.vf, img, a.nkt, wave, lane);    // the wave's first key tiles and ...
    stream_at(ws, a, wave, 32u + 4u * g.member, lane);    // ... the cross-attention block's fc: in flight under the projection
    head_projection<false>(qacc, wq, abuf, smem, wave, lane);
    CS_T(2, 4);
    u32x4 qf[2];
    q_fragments(qacc, a.scale_q * CH_LOG2E, qf);
    head_attention(qf, pre, a.kf, a.vf, img, a.nkt, a.Lk, smem, wave, lane);
    CS_T(2, 5);
    lds_barrier();
    CS_T(2, 6);
    zero(acc);
    phase_n512<4, true, 1>(acc, abuf, ws, lane);
    CS_T(2, 7);
    partial_store(acc, p_out, g, wave, lane);
    CS_T(2, 8);
}

// ------------------------------------------------------------------------------------------------------------------------------
// part 3: cross-attention block tail, feed-forward chunk `member`
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void chain_split3_kernel(tcdiff_chain_args a, const float* p_in, float* p_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeo g = split_geo(a);
    char* abuf = smem + CH_ABUF;
    char* hb = smem + CH_H1C;
    float* scr = reinterpret_cast<float*>(smem + CH_SCR);
    const bool mine = (wave >> 1) == g.member;
    f32x4_t acc[4][1];
    CS_T(3, 0);
    partial_sum(acc, p_in, g, wave, lane);
    const RowC rc = row_consts(a, g, a.filmb, a.xres, a.M, 0, false, wave, lane);
    const NormC nc = norm_consts<false>(a, g, a.n3_g, a.n3_b, wave, lane);
    const float* b1 = a.b1 + 256 * g.member + 32 * wave + 4 * (lane >> 4);
    const f32x4_t b1v[2] = {ld4(b1), ld4(b1 + 16)};
    WStream ws;
    stream_at(ws, a, wave, 48u + 16u * g.member, lane);   // {linear1 chunk, linear2 chunk} of this member: 16 contiguous stages
    CS_T(3, 1);
    block_epilogue<true>(acc, rc, g, a.ln_eps, a.xout, scr, smem, wave, lane, mine);
    CS_T(3, 2);
    norm_lds<false>(acc, nc, a.n2_eps, scr + 1024, smem, abuf, nullptr, wave, lane);
    lds_barrier();
    CS_T(3, 3);
    f32x4_t a1[2][1];
    a1[0][0] = a1[1][0] = f32x4_t{0, 0, 0, 0};
    phase_ff1<1>(a1, abuf, ws, lane);
    CS_T(3, 4);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int cc = 32 * wave + 16 * nt;               // chunk column: k-tile cc / 64, 16-byte chunk (cc % 64) / 8 + (g >> 1)
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = a1[nt][0][t] + b1v[nt][t];
        act4_ct<ACT_GELU>(v, ACT_GELU);
        uint2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *reinterpret_cast<uint2*>(hb + (cc >> 6) * 8192 + act_wr_off(lane, 0, (cc & 63) >> 3)) = pk;
    }
    lds_barrier();
    CS_T(3, 5);
    zero(acc);
    phase_n512<8, true, 1>(acc, hb, ws, lane);
    CS_T(3, 6);
    partial_store(acc, p_out, g, wave, lane);
    CS_T(3, 7);
}

// ------------------------------------------------------------------------------------------------------------------------------
// part 4: feed-forward block tail, linear3, the next layer's Q / K / V of the member's heads
// ------------------------------------------------------------------------------------------------------------------------------
template <bool LAST>
__global__ __launch_bounds__(512) void chain_split4_kernel(tcdiff_chain_args a, const float* p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeo g = split_geo(a);
    char* abuf = smem + CH_ABUF;
    float* scr = reinterpret_cast<float*>(smem + CH_SCR);
    const bool mine = (wave >> 1) == g.member;
    f32x4_t acc[4][1];
    CS_T(4, 0);
    partial_sum(acc, p_in, g, wave, lane);
    const RowC rc = row_consts(a, g, a.film3, a.xres, a.M, 0, false, wave, lane);
    const NormC n4 = norm_consts<false>(a, g, a.n4_g, a.n4_b, wave, lane);
    const int rot = (g.lblk + g.member) & 3;               // linear3: every member computes all of it, k-quarters rotated (rotated_product16)
    WStream wa, wb;
    stream_at(wa, a, wave, 112u + 4u * quarter_of(0, rot), lane);
    const f32x4_t b3c = col_fetch(a.b3, wave, lane);
    CS_T(4, 1);
    block_epilogue<false>(acc, rc, g, 0.0f, nullptr, scr, smem, wave, lane, false);
    CS_T(4, 2);
    norm_lds<false>(acc, n4, a.n4_eps, scr, smem, abuf, nullptr, wave, lane);
    lds_barrier();
    CS_T(4, 3);
    stream_at(wb, a, wave, 112u + 4u * quarter_of(1, rot), lane);
    NormC nn;
    if (!LAST) nn = norm_consts<true>(a, g, a.nn_g, a.nn_b, wave, lane);      // the next phase's constants: in flight under linear3
    zero(acc);
    rotated_product16(acc, abuf, a, wave, 112u, rot, wa, wb, lane,
                      [&] { if (!LAST) head_stream(wa, a, g, 128u, wave, lane); },       // the next layer's w_qs slice, then its w_ks slice:
                      [&] { if (!LAST) head_stream(wb, a, g, 144u, wave, lane); });      // land under the stores and the norm
    if (!LAST) CS_T(4, 4);
    lds_barrier();                                         // every wave is out of linear3: the activation block is rewritten below
    {
        const int c = lane & 15, gg = lane >> 4;
        const int row = my_row(g, c);
        col_park(smem, 4, b3c, wave, lane);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = 64 * wave + 16 * nt + 4 * gg;
            acc[nt][0] += col_get(smem, 4, wave, nt, gg);
            if (!mine) continue;
            if (LAST && a.out_ld > 0) {
                if (col < a.out_ld) *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(a.h_out) + (long)row * a.out_ld + col) = acc[nt][0];
            } else if (LAST) {
                uint2 pk = {pack_bf2(acc[nt][0][0], acc[nt][0][1]), pack_bf2(acc[nt][0][2], acc[nt][0][3])};
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.h_out) + (long)row * 512 + col) = pk;
            } else {
                *reinterpret_cast<f32x4_t*>(a.xout + cb_index(g.Mtot, row, col)) = acc[nt][0];
            }
        }
    }
    if (LAST) return;
    CS_T(4, 5);
    // next layer: norm1 + rotary -> Q, K ; norm1 -> V (model/model.py:326,374-383,78-80), heads 2 member, 2 member + 1
    norm_lds<true>(acc, nn, a.nn_eps, scr + 1024, smem, abuf, smem + CH_ABUF2, wave, lane);
    lds_barrier();
    CS_T(4, 6);
    const int head = 2 * g.member + (wave >> 2);
    const bool writer = (wave & 3) == 0;                   // one wave per head stores (all four hold the full tiles)
    f32x4_t t[4][1], tk[4][1];
    head_projection2(t, tk, wa, wb, abuf, smem, wave, lane);
    head_stream(wa, a, g, 160u, wave, lane);              // w_vs slice
    if (writer) store_qfrag<1>(t, a.qf_out, a.scale_q, g.lblk, head, lane);
    CS_T(4, 7);
    if (writer) store_kfrag<1>(tk, a.kf_out, g.bseq, g.bis * 16, a.out_nkt, head, lane);
    CS_T(4, 8);
    head_projection<true>(t, wa, smem + CH_ABUF2, smem, wave, lane);
    if (writer) store_vfrag<1>(t, a.vf_out, g.bseq, g.bis * 16, a.out_nkt, a.L, head, lane);
    CS_T(4, 9);
}

static bool cs_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int tcdiff_chain_split(const tcdiff_chain_args* a, int part, const float* p_in, float* p_out, hipStream_t stream) {
    if (!a || a->M <= 0 || a->L < 16 || a->M % a->L || !a->wstream || part < 0 || (part > 4 && part != 12)) return TC_ERR_ARG;
    static tc_dev_state dev_state;
    const int n_cu = tc_device_once(dev_state, [](int) {
        const void* fns[7] = {reinterpret_cast<const void*>(chain_split12_kernel),
                              reinterpret_cast<const void*>(chain_split0_kernel), reinterpret_cast<const void*>(chain_split1_kernel),
                              reinterpret_cast<const void*>(chain_split2_kernel), reinterpret_cast<const void*>(chain_split3_kernel),
                              reinterpret_cast<const void*>(chain_split4_kernel<false>),
                              reinterpret_cast<const void*>(chain_split4_kernel<true>)};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, CH_SMEM);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    });
    if (n_cu < 0) return n_cu;
    const dim3 grid((unsigned)((a->M / a->L) * ((a->L + 15) / 16) * CS_NM)), blk(512);
    if (part == 0) {
        if (a->mode != TC_CHAIN_FRONT) return TC_ERR_UNSUPPORTED;
        if (a->n_stages != 80 || a->H != 8 || a->nw == 4 || !a->xres || !a->nn_g || !a->nn_b || !a->rope || a->rope_rows < a->L ||
            !a->qf_out || !a->kf_out || !a->vf_out || a->out_nkt <= 0 || a->L > 32 * a->out_nkt)
            return TC_ERR_ARG;
        const void* ptrs0[] = {a->wstream, a->xres, a->nn_g, a->nn_b, a->rope, a->qf_out, a->kf_out, a->vf_out};
        for (const void* p : ptrs0)
            if (!cs_al16(p)) return TC_ERR_ALIGN;
        hipLaunchKernelGGL(chain_split0_kernel, grid, blk, CH_SMEM, stream, *a);
        TC_CHECK_LAUNCH();
        return TC_OK;
    }
    if (part > 1 && a->xres == a->xout) return TC_ERR_ARG;          // not in place (see the file header)
    const bool last = a->mode == TC_CHAIN_FULL_LAST;
    if (a->mode != TC_CHAIN_FULL && !last) return TC_ERR_UNSUPPORTED;
    if (a->n_stages != (last ? 128 : 176) || a->H != 8 || a->nw == 4 || !a->seq_blocks) return TC_ERR_ARG;
    if (!a->film || !a->filmb || !a->film3 || a->film_ld % 4 || !a->xres || !a->xout || !a->n2_g || !a->n2_b || !a->n3_g || !a->n3_b ||
        !a->n4_g || !a->n4_b || !a->b1 || !a->b3 || !a->rope || a->rope_rows < a->L || !a->kf || !a->vf || a->nkt <= 0 || a->Lk <= 0 ||
        a->Lk > 32 * a->nkt || a->n_shared < 0)
        return TC_ERR_ARG;
    if (a->sa_q ? (!a->sa_kf || !a->sa_vf || a->sa_nkt <= 0 || a->L > 32 * a->sa_nkt || a->a_mod % a->L) : !a->A) return TC_ERR_ARG;
    if (last ? !a->h_out || a->out_ld < 0 || a->out_ld % 4 || a->out_ld > 512
             : (!a->qf_out || !a->kf_out || !a->vf_out || !a->nn_g || !a->nn_b || a->out_nkt <= 0 || a->L > 32 * a->out_nkt))
        return TC_ERR_ARG;
    if ((part > 1 && part != 12 && !p_in) || (part != 4 && !p_out)) return TC_ERR_ARG;
    const void* ptrs[] = {a->A, a->wstream, a->film, a->filmb, a->film3, a->xres, a->xout, a->n2_g, a->n2_b, a->n3_g, a->n3_b, a->n4_g,
                          a->n4_b, a->b1, a->b3, a->nn_g, a->nn_b, a->rope, a->kf, a->vf, a->sa_q, a->sa_kf, a->sa_vf, a->qf_out,
                          a->kf_out, a->vf_out, a->h_out, p_in, p_out};
    for (const void* p : ptrs)
        if (p && !cs_al16(p)) return TC_ERR_ALIGN;
    if (part == 12) hipLaunchKernelGGL(chain_split12_kernel, grid, blk, CH_SMEM, stream, *a, p_out);
    else if (part == 1) hipLaunchKernelGGL(chain_split1_kernel, grid, blk, CH_SMEM, stream, *a, p_out);
    else if (part == 2) hipLaunchKernelGGL(chain_split2_kernel, grid, blk, CH_SMEM, stream, *a, p_in, p_out);
    else if (part == 3) hipLaunchKernelGGL(chain_split3_kernel, grid, blk, CH_SMEM, stream, *a, p_in, p_out);
    else if (last) hipLaunchKernelGGL(chain_split4_kernel<true>, grid, blk, CH_SMEM, stream, *a, p_in);
    else hipLaunchKernelGGL(chain_split4_kernel<false>, grid, blk, CH_SMEM, stream, *a, p_in);
    TC_CHECK_LAUNCH();
    return TC_OK;
}
