// Row-block chains of the TCDiff decoder layer (bf16 throughput mode), gfx950.
//
// Everything between the two attentions of a FiLMTransformerDecoderLayer is ROW-LOCAL (a token row needs only itself,
// the weights and its sequence's FiLM vectors): model/model.py:103-106,327 (fc + LayerNorm + FiLM + residual),
// :332,387 (norm2 + rotary + w_qs), and :334,338-339,344,399-401 followed by the next layer's :326,374-383
// (fc + LN + FiLM + residual, norm3, linear1 + GELU, linear2 + FiLM + residual, norm4, linear3, norm1 + rotary,
// w_qs/w_ks/w_vs).  The op-by-op kernels (gemm.hip) move every intermediate through HBM/MALL: ~38 KB per row and layer
// and nine launches.  Here one workgroup keeps a block of 64 rows on the CU for a whole chain:
//
//   chain A : O_self --fc--> LN(1e-6), FiLM, +x --> x (fp32, HBM)  --norm2, rotary--> [LDS] --w_qs--> Q image
//   chain B : O_cross --fc--> LN, FiLM, +x --> x (HBM) --norm3--> [LDS] --linear1, GELU--> [LDS, 256-column chunks]
//             --linear2--> FiLM, +x --norm4--> [LDS] --linear3--> x' (fp32, HBM) --norm1', rotary--> [LDS]
//             --w_qs / w_ks--> Q, K images ; norm1' --> [LDS] --w_vs--> V image        (last layer: stops after linear3)
//   FULL    : chain A, then the cross-attention itself (head w on wave w; K / V from fragment-ordered cache images, the
//             Q^T accumulator tile is the B operand of the score MFMA), then chain B: ONE launch per decoder layer.
//             This is the production path; A and B alone are the reference it is tested against.
//
// Only the weights stream.  Structure:
//   * 8 waves, wave w owns output columns [64 w, 64 w + 64) of every 512-wide GEMM (= head w of Q / K / V) for all 64
//     rows; MFMA operand roles are swapped (A operand = weight rows, B operand = activation rows), so a lane holds 4
//     consecutive columns of ONE row per register quad: LayerNorm statistics are in-register sums + one cross-half
//     swap + an 8-wave exchange through LDS, and FiLM / residual / rotary / bf16 packing need no transposition.
//   * the weights of a chain are packed ON THE HOST (engine.py, once per checkpoint) into one linear stream per wave in
//     consumption order, in 2-KB stages that are already the MFMA fragment image ([n-tile][half][row][16 B]).  A wave
//     consumes only fragments of its OWN columns, so weights never touch LDS: a stage is two coalesced 1-KB global
//     loads straight into registers, 8 stages (16 KB per wave, 128 KB per CU) are in flight in a register ring whose
//     slots are compile-time indices, and the GEMM phases (fully unrolled) have NO workgroup barrier and no
//     hand-written waits.  The stream runs ahead across GEMM and epilogue boundaries (the next GEMM's first stages land
//     during the LayerNorm in front of it).
//   * activations live in LDS as [k-tile][64 rows][128 B] with the XOR chunk swizzle of common.h (tile_off).
//   * every global access of an epilogue is a contiguous kilobyte per wave instruction: the fp32 residual stream and the
//     rotary table are column-blocked (RowPipe below), head-major images leave through wave-private LDS staging
//     (store_heads).  With the accumulator layout the natural "my 16 bytes of my row" access is 32 separate line
//     requests per instruction; those passes cost 30 us of a 130-us launch before.
// Barriers: one pair per LayerNorm (statistics exchange) and one per activation hand-off.
#include "common.h"
#include "tcdiff_hip.h"

// LDS map.  The small constant areas come first so that their reads are `base VGPR + 16-bit immediate`.
#define CH_FILM 0            //  8 KB  FiLM (scale | shift) rows of the <= 2 sequences this block touches, for the next epilogue
#define CH_VEC 8192          // 12 KB  six 512-float vectors (LayerNorm weights, biases) of the next epilogue(s)
#define CH_SCR 20480         //  8 KB  LayerNorm statistics exchange: 2 x [2][8 waves][64 rows] floats
#define CH_ABUF 28672        // 64 KB  activation block [8 k-tiles][64][128 B]
#define CH_H1C 94208         // 32 KB  GELU(linear1) chunk [4 k-tiles][64][128 B]
#define CH_ABUF2 94208       // 64 KB  second activation block (un-rotated norm1 image for V); overlays the dead h1 chunk
#define CH_STG7 159744       //  4 KB  eighth staging slot of store_heads (slots 0-6: the first 28 KB)
#define CH_SMEM 163840
#ifndef CH_D
#define CH_D 8               // weight stages in flight per wave (registers): 8 x 2 KB x 8 waves = 128 KB per CU
#endif
#define CH_R 16              // ring slots inside a GEMM phase: a phase tops the CH_D resident stages up to CH_R in flight
#ifndef CH_QKV_R
#define CH_QKV_R CH_D        // ring depth of the Q / K / V GEMMs at the end of the launch
#endif
#ifndef CH_XP
#define CH_XP 0              // K / V register sets of the pipelined in-kernel cross-attention for a 5-tile memory (0: off).
                             // Measured with 2 and 3 sets: the launch takes the same time (115.1 / 116.5 vs 115.7 / 114.3 us,
                             // same box): the cross-attention streams 640 KB of K / V per block and is bound by that, like
                             // the weight stream, not by the round trip per tile.  Kept as a build option (-DCH_XP=3).
#endif
#define CH_STAGE 2048

typedef const float* fptr;

// A wave's weight stream.  The fragments of a stage are private to the wave (it owns the output columns they produce),
// so they never need LDS: a stage is two coalesced 1-KB global loads straight into the registers the MFMAs read, and
// the ring of CH_D stages in flight is a register array indexed at compile time (every loop over it is unrolled).
// The compiler counts vmcnt for these loads itself.  Past the end of the stream the last stage is re-read (never used).
struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;   // this wave's stream as a raw buffer: a stage address is SGPR descriptor + SGPR stage
    unsigned voff;                 // offset + this one VGPR (lane * 16); as 64-bit global addresses every stage load
                                   // cost a v_lshl_add_u64 and a VGPR pair (760 VALU instructions per launch and wave)
    unsigned pos;      // stages consumed so far (wave-uniform)
    unsigned last;     // index of the last stage
    u32x4 a[CH_R], b[CH_R];   // slots 0 .. CH_D-1 live across phases, the rest only inside a GEMM phase
};
DEVINL void ws_load(WStream& ws, int slot, unsigned stage) {
    const unsigned st = stage < ws.last ? stage : ws.last;
    const unsigned so = st * CH_STAGE;                       // scalar
    ws.a[slot] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, so, 0));
    ws.b[slot] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff + 1024u, so, 0));
}
#define CH_MMA(acc, w, a) MmaBF16::mma(acc, w, a)
#define CH_FRAG(at, row, ch) lds_frag(at, row, ch)
// IR-level fence for memory operations + machine-scheduler fence for everything: keeps an unrolled epilogue loop one
// iteration at a time (see the fc epilogue)
#define CH_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
DEVINL void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Fresh copies of the lane / wave index that the compiler cannot relate to earlier ones: every phase derives its LDS
// and global addresses from its own copy, so address arithmetic is recomputed per phase (a few VALU ops) instead of
// being hoisted to the top of the kernel and kept alive across it -- which, with 64 accumulator + 32 ring registers
// resident, spilled ~150 VGPRs, and every scratch reload in an epilogue is a full memory round trip.
DEVINL int fresh_v(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
DEVINL int fresh_s(int x) {
    asm volatile("" : "+s"(x));
    return x;
}

DEVINL f32x4_t ld4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
DEVINL void zero(f32x16_t& v) {
    const f32x16_t z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v = z;
}

// acc[mi][ni] (rows 32 mi + r, columns 64 wave + 32 ni + ..) += act[64 x 16 NST] (k-steps 0.. of `abuf`) * W stages;
// a stage = one 16-deep k-step of the wave's 64 weight rows: fragment ni = weight rows 32 ni + r.  NST % CH_D == 0.
// Fully unrolled (NST <= 32 stage bodies): a rolled loop carries the ring through a phi, and hipcc placed a register
// copy of the most recently loaded slot at the loop header -- i.e. `s_waitcnt vmcnt(0)`, a full drain of the wave's
// weight stream, every CH_D stages.  The activation fragments of stage ks + 1 are read before the MFMAs of stage ks.
// (Tried: the two waves of a SIMD taking turns at issue priority every few stages (s_setprio), because with the
// default oldest-first arbitration waves 0-3 finish every barrier-free stretch ~2 us before waves 4-7.  It balances
// them, and the stretch takes exactly as long: the pair is bound by what the CU gets from L2, not by arbitration.)
// TAIL: the launch's last phase -- its last CH_D stages refill nothing (there is nothing behind them).
template <int NST, bool TAIL = false, int R = CH_D>
DEVINL void phase_n512(f32x16_t (&acc)[2][2], const char* abuf, WStream& ws, int lane) {
    static_assert(NST % R == 0, "a phase starts and ends at ring slot 0");
    lane = fresh_v(lane);
    const int r = lane & 31, h = lane >> 5;
    u32x4 a0 = CH_FRAG(abuf, r, h), a1 = CH_FRAG(abuf, 32 + r, h);
    // Between phases CH_D stages are in flight (64 VGPRs, all the epilogues can spare).  Inside a phase the accumulators
    // and the ring are all that is live, so the phase opens by topping the ring up to CH_R stages: the weight stream of
    // a CU is latency-bound (bytes in flight / ~1.7 us), and the deeper ring is what the register file allows HERE.
    // Relative stage j lives in slot j % CH_R; nothing past the next phase's first CH_D stages is loaded, so the phase
    // ends as it began: stages NST .. NST + CH_D - 1 in slots 0 .. CH_D - 1.
    const unsigned base = ws.pos;
#pragma unroll
    for (int j = CH_D; j < R; ++j)
        if (!(TAIL && j >= NST)) ws_load(ws, j, base + j);
#pragma unroll
    for (int ks = 0; ks < NST; ++ks) {
        const int i = ks % R;
        const u32x4 w0 = ws.a[i], w1 = ws.b[i];
#ifdef CH_NO_TAIL
        if (ks + R < NST + CH_D) ws_load(ws, i, base + ks + R);
#else
        if (ks + R < NST + CH_D && !(TAIL && ks + R >= NST)) ws_load(ws, i, base + ks + R);
#endif
        u32x4 n0 = a0, n1 = a1;
        if (ks + 1 < NST) {
            const char* at = abuf + ((ks + 1) >> 2) * 8192;
            const int ch = 2 * ((ks + 1) & 3) + h;
            n0 = CH_FRAG(at, r, ch);
            n1 = CH_FRAG(at, 32 + r, ch);
        }
        // Stage order: the next stage's two LDS reads and the refill are ISSUED, then this stage's MFMAs run (the
        // ~100 cycles of LDS latency pass under them even when the wave is alone on its SIMD), then the fence.  Left to
        // itself hipcc schedules read, wait, use, and a wave whose SIMD mate is parked at a barrier exposes the whole
        // LDS latency in every stage.
        __builtin_amdgcn_sched_barrier(0);
        CH_MMA(acc[0][0], w0, a0);
        CH_MMA(acc[0][1], w1, a0);
        CH_MMA(acc[1][0], w0, a1);
        CH_MMA(acc[1][1], w1, a1);
        __builtin_amdgcn_sched_barrier(0);   // ... and the MFMAs do not sink below the next stage's reads either
        // Stage fence.  Memory clobber: the refill stays in its own stage, the stream never drains.  The next stage's
        // fragments pass THROUGH it, so the next stage's MFMAs cannot be pulled up to right behind their reads.
        asm volatile("" : "+v"(n0), "+v"(n1) : : "memory");
        a0 = n0;
        a1 = n1;
    }
    ws.pos = base + NST;
}
// The rolled form (one CH_D-stage body, looped): kept for TC_CHAIN_A alone, whose fully unrolled build hipcc spills
// (277 VGPRs); that mode is the reference the fused launch is tested against, not the production path.
DEVINL void phase_n512_rolled(f32x16_t (&acc)[2][2], const char* abuf, int nst, WStream& ws, int lane) {
    lane = fresh_v(lane);
    const int r = lane & 31, h = lane >> 5;
#pragma unroll 1
    for (int s0 = 0; s0 < nst; s0 += CH_D) {
#pragma unroll
        for (int i = 0; i < CH_D; ++i) {
            const u32x4 w0 = ws.a[i], w1 = ws.b[i];
            ws_load(ws, i, ws.pos + CH_D);
            ws.pos++;
            const int ks = s0 + i;
            const char* at = abuf + (ks >> 2) * 8192;
            const int ch = 2 * (ks & 3) + h;
            const u32x4 a0 = CH_FRAG(at, r, ch), a1 = CH_FRAG(at, 32 + r, ch);
            CH_MMA(acc[0][0], w0, a0);
            CH_MMA(acc[0][1], w1, a0);
            CH_MMA(acc[1][0], w0, a1);
            CH_MMA(acc[1][1], w1, a1);
            asm volatile("" ::: "memory");
        }
    }
}
// linear1 chunk: acc[mi] (columns 32 wave + ..) += act[64 x 512] * W1 chunk; a stage = 2 k-steps of the wave's 32 rows
DEVINL void phase_ff1(f32x16_t (&acc)[2], const char* abuf, WStream& ws, int lane) {
    lane = fresh_v(lane);
    const int r = lane & 31, h = lane >> 5;
    u32x4 a0 = CH_FRAG(abuf, r, h), a1 = CH_FRAG(abuf, 32 + r, h);
    const unsigned base = ws.pos;
    // (CH_D deep only: two accumulator sets are live in the feed-forward loop)
#pragma unroll
    for (int st = 0; st < 16; ++st) {
        const int i = st % CH_D;
        const u32x4 wk[2] = {ws.a[i], ws.b[i]};
        ws_load(ws, i, base + st + CH_D);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int ks = 2 * st + k2;
            u32x4 n0 = a0, n1 = a1;
            if (ks + 1 < 32) {
                const char* at = abuf + ((ks + 1) >> 2) * 8192;
                const int ch = 2 * ((ks + 1) & 3) + h;
                n0 = CH_FRAG(at, r, ch);
                n1 = CH_FRAG(at, 32 + r, ch);
            }
            __builtin_amdgcn_sched_barrier(0);   // see phase_n512
            CH_MMA(acc[0], wk[k2], a0);
            CH_MMA(acc[1], wk[k2], a1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(n0), "+v"(n1) : : "memory");
            a0 = n0;
            a1 = n1;
        }
    }
    ws.pos = base + 16;
}

// LayerNorm statistics of the 64 rows over all 512 columns: this lane's rows are 32 mi + r.  One exchange: every wave
// publishes (sum, sum of squares) of its 64 columns, var = E[v^2] - mean^2 in fp32 (|mean| is of the order of the
// standard deviation for these activations: the cancellation costs ~1e-7 relative, far below the bf16 operands).
// Returns rstd and nmr = -mean * rstd: the normalised value is fma(v, rstd, nmr), one op per element instead of two.
// The sums run on float2 accumulators (v_pk_add_f32 / v_pk_fma_f32: 64 instructions for the 64 values, not 128).
DEVINL void row_stats(const f32x16_t (&acc)[2][2], float* scr, int wave, int lane, float eps, float (&nmr)[2],
                      float (&rstd)[2]) {
    lane = fresh_v(lane);
    wave = fresh_s(wave);
    const int r = lane & 31;
    float s[2], s2[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        f32x2_t t = {0.0f, 0.0f}, t2 = {0.0f, 0.0f};
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const f32x2_t v = {acc[mi][ni][q], acc[mi][ni][q + 1]};
                t += v;
                t2 = __builtin_elementwise_fma(v, v, t2);
            }
        const float ts = t[0] + t[1], t2s = t2[0] + t2[1];
        s[mi] = ts + other_half(ts);
        s2[mi] = t2s + other_half(t2s);
    }
    if (lane < 32) {
        scr[wave * 64 + r] = s[0];
        scr[wave * 64 + 32 + r] = s[1];
        scr[512 + wave * 64 + r] = s2[0];
        scr[512 + wave * 64 + 32 + r] = s2[1];
    }
    lds_barrier();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        float t = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            t += scr[w * 64 + 32 * mi + r];
            t2 += scr[512 + w * 64 + 32 * mi + r];
        }
        const float mean = t * (1.0f / 512.0f);
        const float var = fmaxf(t2 * (1.0f / 512.0f) - mean * mean, 0.0f);
        rstd[mi] = rsqrtf(var + eps);
        nmr[mi] = -mean * rstd[mi];
    }
}

// constants of an epilogue come from LDS (staged by stage_consts): no long-latency loads, no long-lived registers
// Byte offset of this lane's first column (64 wave + 4 h) in a 512-float LDS vector, as a value the compiler cannot take
// apart: the per-iteration column offsets then fold into the ds_read immediates instead of being recomputed (under
// register pressure hipcc rematerialised ~2 VALU adds per constant read: 90 per epilogue).
DEVINL int col_base_bytes(int wave, int h) {
    int v = (64 * wave + 4 * h) * 4;
    asm volatile("" : "+v"(v));
    return v;
}
DEVINL f32x4_t lds4b(const char* base, int byte_off) {
    return *reinterpret_cast<const f32x4_t*>(base + byte_off);
}
DEVINL f32x4_t lds4(const char* base, int float_index) {
    return *reinterpret_cast<const f32x4_t*>(base + float_index * 4);
}

// This lane's two rows of a [rows, 512] fp32 matrix (residual stream, rotary table), 8 column groups each: a register
// pipeline 4 groups deep (32 VGPRs; all 16 float4 at once would not fit beside accumulators and weight ring).
// Layouts.  Row-major [row][512]: a load instruction then touches 32 rows x 32 bytes = 32 separate line requests, and
// five such passes plus three store passes cost ~17 us of a 130-us launch (measured by ablation).  COLUMN-BLOCKED
// [64 groups of 8 columns][rows][8 floats]: the 32 rows of a lane half are consecutive, so an instruction reads ONE
// contiguous kilobyte.  The residual stream between chain launches and the rotary table handed to them are
// column-blocked; only layer 0's input (written by gemm_rowln) is row-major (`xres_rowmajor`).
struct RowPipe {
    __amdgpu_buffer_rsrc_t rsrc;   // the matrix as a raw buffer: address = SGPR descriptor + SGPR (column group) + VGPR (row)
    unsigned voff[2];              // byte offset of this lane's 16 bytes of its row inside a column group
    unsigned soff, its;            // byte offset of column group 8 wave, bytes between column groups (wave-uniform)
    f32x4_t q[4][2];               // [group & 3][row tile]
};
DEVINL __amdgpu_buffer_rsrc_t f32_buffer(const float* base, long n_floats) {
    const long bytes = n_floats * 4;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes < 0xFFFFFFFFl ? (int)bytes : -1, 0x00020000);
}
DEVINL void rp_issue(RowPipe& rp, int it) {
#ifdef CH_ABLATE_XLOAD
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) rp.q[it & 3][mi] = f32x4_t{0.5f, 0.25f, 0.5f, 0.25f};
#else
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
        rp.q[it & 3][mi] = __builtin_bit_cast(
            f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rp.rsrc, rp.voff[mi], rp.soff + (unsigned)it * rp.its, 0));
#endif
}
// rows: row count of the column-blocked matrix, or 0 for a row-major one (`total_rows` rows of 512 floats)
DEVINL void rp_start(RowPipe& rp, const float* base, const int (&row)[2], long rows, long total_rows, int wave, int h) {
    rp.rsrc = f32_buffer(base, total_rows * 512);
    rp.its = rows > 0 ? (unsigned)rows * 32u : 32u;
    rp.soff = rows > 0 ? (unsigned)wave * 8u * rp.its : (unsigned)wave * 256u;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
        rp.voff[mi] = rows > 0 ? (unsigned)row[mi] * 32u + 16u * h : (unsigned)row[mi] * 2048u + 16u * h;
#pragma unroll
    for (int it = 0; it < 4; ++it) rp_issue(rp, it);
}
// store this lane's 16 bytes of column group 8 wave + it of a column-blocked matrix of `rows` rows
DEVINL void cb_store(__amdgpu_buffer_rsrc_t rsrc, long rows, int wave, int it, int row, int h, f32x4_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (unsigned)row * 32u + 16u * h,
                                           (unsigned)(wave * 8 + it) * ((unsigned)rows * 32u), 0);
}

// u = LayerNorm(acc) (optionally rotated) -> bf16 -> activation block in LDS (k = column); g, b: LDS vectors;
// rp: the rotary rows (cos0 sin0 cos1 sin1 per column quad), started by the caller before the statistics exchange
template <bool ROT>
DEVINL void norm_to_lds(const f32x16_t (&acc)[2][2], const float (&nmr)[2], const float (&rstd)[2], const char* g,
                        const char* b, RowPipe& rp, char* abuf, int wave, int lane, char* plain) {
    lane = fresh_v(lane);
    wave = fresh_s(wave);
    const int r = lane & 31, h = lane >> 5;
    const int cb0 = col_base_bytes(wave, h);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int ni = it >> 2, gq = it & 3;
        const f32x4_t g4 = lds4b(g + cb0, 32 * it), b4 = lds4b(b + cb0, 32 * it);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            float u[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) u[t] = fmaf(fmaf(acc[mi][ni][4 * gq + t], rstd[mi], nmr[mi]), g4[t], b4[t]);
            if (plain) {   // the un-rotated image too (V = norm1(x) W_v)
                uint2 pk;
                pk.x = pack_bf2(u[0], u[1]);
                pk.y = pack_bf2(u[2], u[3]);
                *reinterpret_cast<uint2*>(plain + wave * 8192 + tile_off(32 * mi + r, 4 * ni + gq) + 8 * h) = pk;
            }
            if (ROT) {
                const f32x4_t q = rp.q[it & 3][mi];   // cos0 sin0 cos1 sin1
                const float y0 = u[0] * q[0] - u[1] * q[1], y1 = u[1] * q[0] + u[0] * q[1];
                const float y2 = u[2] * q[2] - u[3] * q[3], y3 = u[3] * q[2] + u[2] * q[3];
                u[0] = y0; u[1] = y1; u[2] = y2; u[3] = y3;
            }
            uint2 pk;
            pk.x = pack_bf2(u[0], u[1]);
            pk.y = pack_bf2(u[2], u[3]);
            *reinterpret_cast<uint2*>(abuf + wave * 8192 + tile_off(32 * mi + r, 4 * ni + gq) + 8 * h) = pk;
        }
        if (ROT && it + 4 < 8) rp_issue(rp, it + 4);
        CH_FENCE();   // one column group at a time (see the fc epilogue)
    }
}

// head-major scatter of a 512-wide projection (wave = head): model/model.py:78-80,92-95.  The accumulator layout gives a
// lane 8 bytes of a row at a time; written like that every store instruction makes 32 sixteen-byte write requests, and
// the three scatters of the next layer's Q, K, V cost 13 us of a 130-us launch.  Instead each 32-row half of the wave's
// [64 rows][64 columns] tile goes through 4 KB of the (by now idle) constants area -- wave-private, XOR-swizzled by
// (row >> 1) & 7 (the 64 banks hold two 128-byte rows), no barrier -- and leaves as 16 bytes per lane, 8 lanes per 128-byte row: 8 full lines per instruction.
DEVINL char* stage_area(char* smem, int wave) { return smem + (wave < 7 ? wave * 4096 : CH_STG7); }
template <bool SCALE>   // Q carries 1 / sqrt(d_k); K and V are stored as they are
DEVINL void store_heads(const f32x16_t (&acc)[2][2], void* base, float scale, int L, int Lp, int H, int m0, int M,
                        int wave, int lane, char* smem, int dn = 1, int dancer = 0) {
    // rows are FRAMES m0 .. of dancer `dancer` (token = frame dn + dancer; dn = 1: rows are tokens); L tokens per sequence
#ifdef CH_ABLATE_STORES   // timing experiment only: how much of the Q / K / V tail is the head-major scatter?
    if (M > 0) return;
#endif
    lane = fresh_v(lane);
    wave = fresh_s(wave);
    const int r = lane & 31, h = lane >> 5;
    char* stg = stage_area(smem, wave);
    const int row0 = lane >> 3, ch = lane & 7;      // read side: row row0 + 8 k, 16-byte chunk ch
    const int Lf = L / dn;                          // frames per sequence
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 pk;
                if (SCALE) {
                    pk.x = pack_bf2(acc[mi][ni][4 * gq + 0] * scale, acc[mi][ni][4 * gq + 1] * scale);
                    pk.y = pack_bf2(acc[mi][ni][4 * gq + 2] * scale, acc[mi][ni][4 * gq + 3] * scale);
                } else {
                    pk.x = pack_bf2(acc[mi][ni][4 * gq + 0], acc[mi][ni][4 * gq + 1]);
                    pk.y = pack_bf2(acc[mi][ni][4 * gq + 2], acc[mi][ni][4 * gq + 3]);
                }
                *reinterpret_cast<uint2*>(stg + r * 128 + (((4 * ni + gq) ^ ((r >> 1) & 7)) << 4) + 8 * h) = pk;
            }
        // the LDS queue of a wave is in order: its reads below see its writes above
        int m = m0 + 32 * mi + row0;
        const int seq = m / Lf;
        int tokf = m - seq * Lf;
        // destination of (sequence, head = wave, token, chunk): +8 frames = +8 dn tokens of 128 bytes; past the end of a
        // sequence the next one starts (H * Lp - L) rows further
        uint16_t* dst = reinterpret_cast<uint16_t*>(base) + (((long)seq * H + wave) * Lp + tokf * dn + dancer) * 64 + ch * 8;
        const long wrap = ((long)H * Lp - L) * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = row0 + 8 * k;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
            if (m < M) *reinterpret_cast<u32x4*>(dst) = v;
            m += 8;
            tokf += 8;
            dst += 8 * 64 * dn;
            if (tokf >= Lf) {
                tokf -= Lf;
                dst += wrap;
            }
        }
    }
}

// Cross-attention of this wave's head inside the chain (model/model.py:386-396,97-102 with cached K / V): the wave owns
// head `wave` of all 64 rows.  qacc = (rot W_q^T)^T tiles straight from the projection GEMM (lane = row, registers = d):
// scaled and packed they ARE the B operand of S^T = K Q^T, with d in the order the accumulator holds it -- the K / V
// caches are kept in a second, fragment-ordered image (tcdiff_pack_kv_frags) whose 1-KB pieces load straight into the
// A operands.  Online softmax over 32-key tiles exactly as csrc/attention.hip; O^T = V^T P^T with P^T fed from the S^T
// accumulator registers.  A 32-row tile that straddles two sequences runs once per sequence and every lane keeps the
// result of its own row's sequence.  Output: bf16 O rows into the activation block (columns 64 wave ..).
DEVINL void cross_attention(const f32x16_t (&qacc)[2][2], const tcdiff_chain_args& a, int m0, char* abuf, int wave,
                            int lane) {
    lane = fresh_v(lane);
    wave = fresh_s(wave);
    const int r = lane & 31, h = lane >> 5;
    const int M = a.M, L = a.L, nkt = a.nkt;
    u32x4 qf[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int s16 = 0; s16 < 4; ++s16) {
            const int ni = s16 >> 1, o8 = 8 * (s16 & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                qf[mi][s16][j] = pack_bf2(qacc[mi][ni][o8 + 2 * j] * a.scale_q, qacc[mi][ni][o8 + 2 * j + 1] * a.scale_q);
        }
    constexpr float LOG2E = 1.4426950408889634f;
    const long head_bytes = (long)nkt * 4096;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        int ra = m0 + 32 * mi, rb = ra + 31;
        ra = ra < M ? ra : M - 1;
        rb = rb < M ? rb : M - 1;
        const int sa = ra / L, sb = rb / L;                 // wave-uniform
        int mrow = m0 + 32 * mi + r;
        mrow = mrow < M ? mrow : M - 1;
        const int my_seq = mrow / L;
#pragma unroll 1
        for (int seq = sa; seq <= sb; ++seq) {
            const int kv = seq < a.n_shared ? 0 : seq - a.n_shared + (a.n_shared > 0 ? 1 : 0);
            const u32x4* Kp = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.kf) +
                                                              ((long)kv * a.H + wave) * head_bytes) + lane;
            const u32x4* Vp = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.vf) +
                                                              ((long)kv * a.H + wave) * head_bytes) + lane;
            f32x16_t o[2];
            zero(o[0]);
            zero(o[1]);
            float m_run = -INFINITY, l_run = 0.0f;
            u32x4 kn[4];                                           // K fragments run one tile ahead
#pragma unroll
            for (int i = 0; i < 4; ++i) kn[i] = Kp[i * 64];
#pragma unroll 1
            for (int kt = 0; kt < nkt; ++kt) {
                u32x4 kc[4], vc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    kc[i] = kn[i];
                    vc[i] = Vp[(kt * 4 + i) * 64];                 // V of this tile: in flight under QK^T and the softmax
                }
                const int nx = kt + 1 < nkt ? kt + 1 : kt;       // the last iteration re-reads its own tile (unused)
#pragma unroll
                for (int i = 0; i < 4; ++i) kn[i] = Kp[(nx * 4 + i) * 64];
                f32x16_t s;
                zero(s);
#pragma unroll
                for (int s16 = 0; s16 < 4; ++s16) MmaBF16::mma(s, kc[s16], qf[mi][s16]);
                if (kt * 32 + 32 > a.Lk) {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (kt * 32 + acc_row(q, h) >= a.Lk) s[q] = -INFINITY;
                }
                float mx = s[0];
#pragma unroll
                for (int q = 1; q < 16; ++q) mx = fmaxf(mx, s[q]);
                mx = fmaxf(mx, other_half(mx)) * LOG2E;
                const float m_new = fmaxf(m_run, mx);
                // x = s log2(e) - m and the row sum as float2 ops (v_pk_fma_f32 / v_pk_add_f32); exp2 stays scalar
                f32x2_t rs2 = {0.0f, 0.0f};
                const f32x2_t l2 = {LOG2E, LOG2E}, nm = {-m_new, -m_new};
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    const f32x2_t x = __builtin_elementwise_fma(f32x2_t{s[q], s[q + 1]}, l2, nm);
                    const f32x2_t pp = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                    s[q] = pp[0];
                    s[q + 1] = pp[1];
                    rs2 += pp;
                }
                float rs = rs2[0] + rs2[1];
                rs += other_half(rs);
                // the running maximum moves in the first tile or two; afterwards the whole wave skips the rescale
                if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: exp2(-inf) = 0, o is 0
                    l_run *= alpha;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int q = 0; q < 16; ++q) o[dt][q] *= alpha;
                    m_run = m_new;
                }
                l_run += rs;
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    u32x4 pf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pf[j] = pack_bf2(s[8 * sp + 2 * j], s[8 * sp + 2 * j + 1]);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) MmaBF16::mma(o[dt], vc[sp * 2 + dt], pf);
                }
            }
            if (my_seq == seq) {
                const float inv = 1.0f / l_run;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        uint2 pk;
                        pk.x = pack_bf2(o[dt][4 * gq + 0] * inv, o[dt][4 * gq + 1] * inv);
                        pk.y = pack_bf2(o[dt][4 * gq + 2] * inv, o[dt][4 * gq + 3] * inv);
                        *reinterpret_cast<uint2*>(abuf + wave * 8192 + tile_off(32 * mi + r, 4 * dt + gq) + 8 * h) = pk;
                    }
            }
        }
    }
}

// The same cross-attention with the K / V fragment loads of later tiles in flight while a tile is computed, for a memory of
// exactly NKT tiles (the benchmark's 150 + 2 keys: NKT = 5).  The loop above exposes one L2 round trip per tile (a dependent
// chain K -> S -> softmax -> P -> V of ~0.3 us of work per ~1 us of latency, 10 tiles per wave).  Here the (<= 2 NKT) tiles
// of a row tile are straight-line code over NSET register sets (a rolled loop would carry them through phis, and a copy
// of a set in flight is a vmcnt drain), addressed through a raw buffer (SGPR descriptor + SGPR tile offset).  The sets
// need the weight ring's registers: the caller runs the GEMM in front without refilling the ring and re-primes it behind.
template <int NKT, int NSET>
DEVINL void cross_attention_p(const f32x16_t (&qacc)[2][2], const tcdiff_chain_args& a, int m0, char* abuf, int wave,
                              int lane) {
    static_assert(NSET == 2 || NSET == 3, "two or three K / V register sets");
    lane = fresh_v(lane);
    wave = fresh_s(wave);
    const int r = lane & 31, h = lane >> 5;
    const int M = a.M, L = a.L;
    u32x4 qf[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int s16 = 0; s16 < 4; ++s16) {
            const int ni = s16 >> 1, o8 = 8 * (s16 & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                qf[mi][s16][j] = pack_bf2(qacc[mi][ni][o8 + 2 * j] * a.scale_q, qacc[mi][ni][o8 + 2 * j + 1] * a.scale_q);
        }
    constexpr float LOG2E = 1.4426950408889634f;
    const unsigned voff = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.kf), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.vf), 0, -1, 0x00020000);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        int ra = m0 + 32 * mi, rb = ra + 31;
        ra = ra < M ? ra : M - 1;
        rb = rb < M ? rb : M - 1;
        const int sa = ra / L, sb = rb / L;                 // wave-uniform; sb - sa is 0 or 1
        int mrow = m0 + 32 * mi + r;
        mrow = mrow < M ? mrow : M - 1;
        const int my_seq = mrow / L;
        const int J = (sb - sa + 1) * NKT;                  // tiles of this row tile
        auto fetch = [&](int j, u32x4 (&kc)[4], u32x4 (&vc)[4]) {   // tile j = (sequence sa + j / NKT, key tile j % NKT)
            const int seq = sa + (j >= NKT ? 1 : 0), kt = j >= NKT ? j - NKT : j;
            const int kv = seq < a.n_shared ? 0 : seq - a.n_shared + (a.n_shared > 0 ? 1 : 0);
            const unsigned so = (unsigned)(((kv * a.H + wave) * NKT + kt) * 4096);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kc[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(kr, voff + 1024u * i, so, 0));
                vc[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vr, voff + 1024u * i, so, 0));
            }
        };
        f32x16_t o[2];
        float m_run = -INFINITY, l_run = 0.0f;
        auto tile = [&](int j, const u32x4 (&kc)[4], const u32x4 (&vc)[4]) {
            const int kt = j >= NKT ? j - NKT : j;
            if (kt == 0) {
                zero(o[0]);
                zero(o[1]);
                m_run = -INFINITY;
                l_run = 0.0f;
            }
            f32x16_t s;
            zero(s);
#pragma unroll
            for (int s16 = 0; s16 < 4; ++s16) MmaBF16::mma(s, kc[s16], qf[mi][s16]);
            if (kt * 32 + 32 > a.Lk) {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (kt * 32 + acc_row(q, h) >= a.Lk) s[q] = -INFINITY;
            }
            float mx = s[0];
#pragma unroll
            for (int q = 1; q < 16; ++q) mx = fmaxf(mx, s[q]);
            mx = fmaxf(mx, other_half(mx)) * LOG2E;
            const float m_new = fmaxf(m_run, mx);
            f32x2_t rs2 = {0.0f, 0.0f};
            const f32x2_t l2 = {LOG2E, LOG2E}, nm = {-m_new, -m_new};
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const f32x2_t x = __builtin_elementwise_fma(f32x2_t{s[q], s[q + 1]}, l2, nm);
                const f32x2_t pp = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                s[q] = pp[0];
                s[q + 1] = pp[1];
                rs2 += pp;
            }
            float rs = rs2[0] + rs2[1];
            rs += other_half(rs);
            if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int q = 0; q < 16; ++q) o[dt][q] *= alpha;
                m_run = m_new;
            }
            l_run += rs;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                u32x4 pf;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) pf[jj] = pack_bf2(s[8 * sp + 2 * jj], s[8 * sp + 2 * jj + 1]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) MmaBF16::mma(o[dt], vc[sp * 2 + dt], pf);
            }
            if (kt == NKT - 1 && my_seq == sa + (j >= NKT ? 1 : 0)) {   // this sequence is done: rows that belong to it
                const float inv = 1.0f / l_run;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        uint2 pk;
                        pk.x = pack_bf2(o[dt][4 * gq + 0] * inv, o[dt][4 * gq + 1] * inv);
                        pk.y = pack_bf2(o[dt][4 * gq + 2] * inv, o[dt][4 * gq + 3] * inv);
                        *reinterpret_cast<uint2*>(abuf + wave * 8192 + tile_off(32 * mi + r, 4 * dt + gq) + 8 * h) = pk;
                    }
            }
        };
        u32x4 kk[NSET][4], vv[NSET][4];
#pragma unroll
        for (int p = 0; p < NSET - 1; ++p) fetch(p, kk[p], vv[p]);     // NKT >= NSET - 1
#pragma unroll
        for (int j = 0; j < 2 * NKT; ++j) {                  // straight-line: tiles beyond J are skipped (wave-uniform)
            if (j < J) {
                if (j + NSET - 1 < J) fetch(j + NSET - 1, kk[(j + NSET - 1) % NSET], vv[(j + NSET - 1) % NSET]);
                tile(j, kk[j % NSET], vv[j % NSET]);
            }
        }
    }
}

#ifdef CH_STAMP   // diagnostic build: per-phase timestamps of block 0, every wave, into the (otherwise unused) h_out buffer
#ifndef CH_STAMP_BLOCK
#define CH_STAMP_BLOCK 0      // the LOGICAL 64-row block that writes the stamps (7 = rows 448..511: straddles two 450-row sequences)
#endif
#define CH_T(i) do { if (m0 == 64 * CH_STAMP_BLOCK && (threadIdx.x & 63) == 0 && (MODE == TC_CHAIN_B || MODE == TC_CHAIN_FULL)) \
        reinterpret_cast<unsigned long long*>(a.h_out)[(threadIdx.x >> 6) * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// the shader-clock counter beside the 100 MHz one: slots 60 / 61 = s_memtime at the first / last stamp
#define CH_TC(i) do { if (m0 == 64 * CH_STAMP_BLOCK && (threadIdx.x & 63) == 0 && (MODE == TC_CHAIN_B || MODE == TC_CHAIN_FULL)) \
        reinterpret_cast<unsigned long long*>(a.h_out)[(threadIdx.x >> 6) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CH_TC(i) do { } while (0)
#define CH_T(i) do { } while (0)
#endif

// XP > 0: the pipelined in-kernel cross-attention with XP register sets (memory of exactly 5 key tiles), see cross_attention_p
template <int MODE, int XP = 0>
__global__ __launch_bounds__(512) void chain_kernel(tcdiff_chain_args a) {
    constexpr bool HAS_A = MODE == TC_CHAIN_A || MODE == TC_CHAIN_FULL || MODE == TC_CHAIN_FULL_LAST;   // fc + norm2 + w_qs
    constexpr bool FULL = MODE == TC_CHAIN_FULL || MODE == TC_CHAIN_FULL_LAST;                          // + cross-attention
    constexpr bool LAST = MODE == TC_CHAIN_B_LAST || MODE == TC_CHAIN_FULL_LAST;
    // TC_CHAIN_FRONT: the last fusion linear of ONE dancer for a block of 64 FRAMES (A = 64 rows of 1024), which is layer
    // 0's residual input, then layer 0's norm1 + rotary and Q / K / V -- the tail of chain B with a K = 1024 GEMM in
    // front.  Frame F, dancer d <-> token row F dn + d (model/model.py:561; model/diffusion.py:640,651).
    constexpr bool FRONT = MODE == TC_CHAIN_FRONT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int dn = FRONT ? a.dn : 1;
    const int dancer = FRONT ? (int)(blockIdx.x % (unsigned)dn) : 0;
    const int m0 = FRONT ? (int)(blockIdx.x / (unsigned)dn) * 64 : xcd_remap(blockIdx.x, gridDim.x) * 64;
    CH_T(0);
    CH_TC(60);
    const int M = a.M, L = a.L;        // FRONT: M = frames, L = TOKENS per sequence
    char* abuf = smem + CH_ABUF;
    char* h1c = smem + CH_H1C;
    float* scr = reinterpret_cast<float*>(smem + CH_SCR);
    char* cfilm = smem + CH_FILM;      // [2 sequences][scale 512 | shift 512] floats
    char* cvec = smem + CH_VEC;        // six vectors of 512 floats

    // rows of this lane (both row tiles), clamped: rows past M recompute row M - 1 (their inputs are clamped to it)
    int mc[2], sidx[2];
    const int seq0 = (m0 < M ? m0 : M - 1) / L;
    const int seq_last = (M - 1) / L;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + 32 * mi + r;
        mc[mi] = m < M ? m : M - 1;
        sidx[mi] = mc[mi] / L - seq0;          // 0 or 1: L >= 64 rows per sequence (checked by the launcher)
    }
    // Epilogue constants go through LDS.  Thread t carries one float4 of the FiLM rows and up to two of the vectors
    // from global memory to LDS; they are fetched early (latency hidden behind a GEMM) and stored once the previous
    // epilogue no longer reads the area.
    struct Consts { f32x4_t f, v0, v1; };
    auto fetch_consts = [&](const float* film, const float* const (&vec)[6]) {
        Consts k;
        int sq = seq0 + (tid >> 8);
        sq = sq < seq_last ? sq : seq_last;
        k.f = film ? ld4(film + (long)sq * a.film_ld + (tid & 255) * 4) : f32x4_t{0, 0, 0, 0};
        const float* p0 = vec[tid >> 7];
        const float* p1 = tid < 256 ? vec[4 + (tid >> 7)] : nullptr;
        k.v0 = p0 ? ld4(p0 + (tid & 127) * 4) : f32x4_t{0, 0, 0, 0};
        k.v1 = p1 ? ld4(p1 + (tid & 127) * 4) : f32x4_t{0, 0, 0, 0};
        return k;
    };
    auto store_consts = [&](const Consts& k) {
        *reinterpret_cast<f32x4_t*>(cfilm + tid * 16) = k.f;
        *reinterpret_cast<f32x4_t*>(cvec + tid * 16) = k.v0;
        if (tid < 256) *reinterpret_cast<f32x4_t*>(cvec + 8192 + tid * 16) = k.v1;
    };
    auto vecp = [&](int slot) { return cvec + slot * 2048; };
    // constants of the fc block that opens chain B: its own set when chain A ran in front of it in this launch
    const float* fcb_g = FULL ? a.lnb_g : a.ln_g;
    const float* fcb_b = FULL ? a.lnb_b : a.ln_b;
    const float* fcb_film = FULL ? a.filmb : a.film;
    const float* n3_g = FULL ? a.n3_g : a.n2_g;
    const float* n3_b = FULL ? a.n3_b : a.n2_b;

    // ---- the block's input rows (attention output) -> LDS, the first CH_D weight stages -> registers, constants -> LDS
#pragma unroll
    for (int kt = 0; kt < (FRONT ? 16 : 8); ++kt)      // FRONT: 64 rows of 1024 = the activation block and its twin, contiguous
        stage_glds<64, 8>(abuf + kt * 8192, reinterpret_cast<const char*>(a.A) + kt * TC_ROWB, FRONT ? 2048 : 1024, m0, M,
                          a.a_mod, wave, lane);
    WStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.wstream)) + ((long)dancer * 8 + wave) * a.n_stages * CH_STAGE, 0,
        a.n_stages * CH_STAGE, 0x00020000);   // raw buffer (stride 0), bounds = the wave's stream, 32-bit data format
    ws.voff = (unsigned)lane * 16u;
    ws.pos = 0;
    ws.last = (unsigned)a.n_stages - 1;
#pragma unroll
    for (int i = 0; i < CH_D; ++i) ws_load(ws, i, (unsigned)i);
    if (FRONT) {
        const float* const v[6] = {a.b3 + 512 * dancer, a.nn_g, a.nn_b, nullptr, nullptr, nullptr};
        store_consts(fetch_consts(nullptr, v));
    } else {
        const float* const v[6] = {a.ln_g, a.ln_b, a.n2_g, a.n2_b, nullptr, nullptr};
        store_consts(fetch_consts(a.film, v));
    }
    // a wait the compiler can SEE (an asm s_waitcnt is invisible to its vmcnt bookkeeping: it would then treat the
    // prologue loads as still pending at the loop header and wait vmcnt(0) on every trip): vmcnt(0), others untouched
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    CH_T(1);

    f32x16_t acc[2][2];
    auto clear = [&]() {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) zero(acc[mi][ni]);
    };
    float nmr[2], rstd[2];             // LayerNorm of the current rows: u = fma(v, rstd, nmr)
    RowPipe rp;                        // residual rows, later rotary rows, of this lane
    // token row of this lane's rows in the residual stream, and its position in its sequence
    const int trow[2] = {mc[0] * dn + dancer, mc[1] * dn + dancer};
    const long xrows = (long)M * dn;
    const __amdgpu_buffer_rsrc_t xo = f32_buffer(a.xout, xrows * 512);     // the residual stream out
    int pos[2] = {trow[0] % L, trow[1] % L};
    Consts nxt;

    // fc epilogue: LayerNorm(eps), FiLM, residual -> x in the accumulators and in xout (model/model.py:103-106,171-173,
    // 327 / 334); constants in vector slots 0, 1 and the FiLM area; the residual rows were started by the caller
    auto fc_epilogue = [&](float eps, int stamp) {
        lds_barrier();                 // every wave is out of the GEMM: the activation block may be overwritten
        CH_T(stamp);
        row_stats(acc, scr, wave, lane, eps, nmr, rstd);
        CH_T(stamp + 1);
        // fresh copies: the two inlined instances of this epilogue must not share (and keep alive) their addresses
        const int hh = fresh_v(h), wv = fresh_s(wave);
        const int mcl[2] = {fresh_v(mc[0]), fresh_v(mc[1])};
        const int cb0 = col_base_bytes(wv, hh);
        // FiLM rows of this lane's two rows: sequence 0 or 1 of the block (4 KB apart), as opaque byte offsets too
        int fb[2] = {sidx[0] * 4096 + cb0, sidx[1] * 4096 + cb0};
        asm volatile("" : "+v"(fb[0]), "+v"(fb[1]));
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int ni = it >> 2, gq = it & 3;
            const f32x4_t g4 = lds4b(vecp(0) + cb0, 32 * it), b4 = lds4b(vecp(1) + cb0, 32 * it);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const f32x4_t sc = lds4b(cfilm + fb[mi], 32 * it), sh = lds4b(cfilm + fb[mi], 2048 + 32 * it);
                const f32x4_t x4 = rp.q[it & 3][mi];
                f32x4_t o;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float v = fmaf(fmaf(acc[mi][ni][4 * gq + t], rstd[mi], nmr[mi]), g4[t], b4[t]);
                    v = (sc[t] + 1.0f) * v + sh[t];
                    v = x4[t] + v;
                    acc[mi][ni][4 * gq + t] = v;
                    o[t] = v;
                }
                // unguarded: rows past M rewrite row M - 1 with the SAME values (same inputs; every load of this column
                // group, in place or not, was issued before this store)
                cb_store(xo, M, wv, it, mcl[mi], hh, o);
            }
            if (it + 4 < 8) rp_issue(rp, it + 4);
            // one column group at a time: without a fence hipcc hoists the loads of ALL eight groups (row pipeline
            // refills and LDS constants) above the arithmetic, needs ~100 more registers and spills them
            CH_FENCE();
        }
    };

    if (HAS_A) {
        // ================= self-attention block tail: fc + LayerNorm(1e-6) + FiLM + residual, norm2 + rotary, w_qs
        clear();
        if (FULL)
            phase_n512<32>(acc, abuf, ws, lane);
        else
            phase_n512_rolled(acc, abuf, 32, ws, lane);
        CH_T(2);
        {
            int rr[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) rr[mi] = a.xres_mod > 0 ? mc[mi] % a.xres_mod : mc[mi];
            rp_start(rp, a.xres, rr, a.xres_rowmajor ? 0 : (a.xres_mod > 0 ? a.xres_mod : M), a.xres_mod > 0 ? a.xres_mod : M, wave, h);   // in flight during the statistics exchange
        }
        fc_epilogue(a.ln_eps, 40);
        CH_T(3);
        if (FULL) {
            const float* const v[6] = {fcb_g, fcb_b, n3_g, n3_b, nullptr, nullptr};
            nxt = fetch_consts(fcb_film, v);
        }
        rp_start(rp, a.rope, pos, a.rope_rows, a.rope_rows, wave, h);
        row_stats(acc, scr + 1024, wave, lane, a.n2_eps, nmr, rstd);
        CH_T(4);
        // norm2 + rotary (model/model.py:332,387) -> LDS -> Q = rot W_q^T / 8 (model/model.py:78,97)
        norm_to_lds<true>(acc, nmr, rstd, vecp(2), vecp(3), rp, abuf, wave, lane, nullptr);
        lds_barrier();
        CH_T(34);
        if (FULL) store_consts(nxt);   // the cross-attention fc block's constants: read two barriers from here
        clear();
        if (!FULL) {
            phase_n512_rolled(acc, abuf, 32, ws, lane);
            store_heads<true>(acc, a.q_out, a.scale_q, L, a.Lp, a.H, m0, M, wave, lane, smem);
            return;
        }
        // ================= cross-attention in place (the Q image never leaves the registers)
        if constexpr (XP > 0) {
            // pipelined form: the weight ring is left empty behind the w_qs GEMM (its registers hold K / V tiles in the
            // cross-attention) and re-primed with the next GEMM's first stages afterwards
            phase_n512<32, true>(acc, abuf, ws, lane);
            CH_T(35);
            lds_barrier();             // every wave is out of the w_qs GEMM: the activation block becomes O
            cross_attention_p<5, XP>(acc, a, m0, abuf, wave, lane);
#pragma unroll
            for (int i = 0; i < CH_D; ++i) ws_load(ws, i, ws.pos + i);
        } else {
            phase_n512<32>(acc, abuf, ws, lane);
            CH_T(35);
            lds_barrier();             // every wave is out of the w_qs GEMM: the activation block becomes O
            cross_attention(acc, a, m0, abuf, wave, lane);
        }
        CH_T(36);
        lds_barrier();
    }
    if constexpr (!FRONT) {
    // ================= cross-attention block tail: fc + LayerNorm(1e-6) + FiLM + residual (model/model.py:334)
    clear();
    phase_n512<32>(acc, abuf, ws, lane);
    if (FULL) {
        rp_start(rp, a.xout, mc, M, M, wave, h);    // the x this lane stored in the first fc epilogue
    } else {
        int rr[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) rr[mi] = a.xres_mod > 0 ? mc[mi] % a.xres_mod : mc[mi];
        rp_start(rp, a.xres, rr, a.xres_rowmajor ? 0 : (a.xres_mod > 0 ? a.xres_mod : M), a.xres_mod > 0 ? a.xres_mod : M, wave, h);
    }
    CH_T(37);
    fc_epilogue(a.ln_eps, 42);
    CH_T(38);
    row_stats(acc, scr + 1024, wave, lane, a.n2_eps, nmr, rstd);
    CH_T(39);
    // ================= feed-forward (model/model.py:338-339,399-401): norm3 -> LDS
    {
        const float* const v[6] = {a.b1, a.b1 + 512, a.b2, a.n4_g, a.n4_b, nullptr};
        nxt = fetch_consts(a.film3, v);
    }
    norm_to_lds<false>(acc, nmr, rstd, vecp(2), vecp(3), rp, abuf, wave, lane, nullptr);
    lds_barrier();                     // nobody reads the fc constants any more
    store_consts(nxt);
    lds_barrier();                     // ... and everybody sees the feed-forward constants
    CH_T(5);
    clear();   // acc = linear2 accumulator
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        f32x16_t a1[2];
        zero(a1[0]);
        zero(a1[1]);
        phase_ff1(a1, abuf, ws, lane);
        CH_T(6 + 4 * c);
        // two h1 buffers: chunk c - 2's linear2 reads of this one finished before the barrier of chunk c - 1
        char* hb = h1c + (c & 1) * 32768;
        CH_T(7 + 4 * c);
        {
            const int nb = 256 * c + 32 * wave;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4_t b4 = lds4(cvec, nb + 8 * gq + 4 * h);      // b1 occupies vector slots 0 and 1
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    float v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = a1[mi][4 * gq + t] + b4[t];
                    act4_ct<ACT_GELU>(v, ACT_GELU);
                    uint2 pk;
                    pk.x = pack_bf2(v[0], v[1]);
                    pk.y = pack_bf2(v[2], v[3]);
                    // chunk column 32 wave + 8 gq + 4 h: k-tile wave / 2, 16-byte chunk 4 (wave & 1) + gq
                    *reinterpret_cast<uint2*>(hb + (wave >> 1) * 8192 + tile_off(32 * mi + r, 4 * (wave & 1) + gq) +
                                              8 * h) = pk;
                }
            }
        }
        lds_barrier();
        CH_T(8 + 4 * c);
        phase_n512<16>(acc, hb, ws, lane);
        CH_T(9 + 4 * c);
    }
    // linear2 bias, FiLM, residual (the x this lane stored above), norm4 -> LDS
    rp_start(rp, a.xout, mc, M, M, wave, h);
    const int cb2 = col_base_bytes(fresh_s(wave), fresh_v(h));
    int fb2[2] = {sidx[0] * 4096 + cb2, sidx[1] * 4096 + cb2};
    asm volatile("" : "+v"(fb2[0]), "+v"(fb2[1]));
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int ni = it >> 2, gq = it & 3;
        const f32x4_t b4 = lds4b(vecp(2) + cb2, 32 * it);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const f32x4_t sc = lds4b(cfilm + fb2[mi], 32 * it), sh = lds4b(cfilm + fb2[mi], 2048 + 32 * it);
            const f32x4_t x4 = rp.q[it & 3][mi];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = acc[mi][ni][4 * gq + t] + b4[t];
                v = (sc[t] + 1.0f) * v + sh[t];
                acc[mi][ni][4 * gq + t] = x4[t] + v;
            }
            // nothing in this iteration touches memory after its loads, so the arithmetic is free to sink below the
            // loads of all later iterations (whose operands then all have to be kept): pin it to this iteration
            asm volatile("" ::"v"(acc[mi][ni][4 * gq + 0]), "v"(acc[mi][ni][4 * gq + 1]), "v"(acc[mi][ni][4 * gq + 2]),
                         "v"(acc[mi][ni][4 * gq + 3]));
        }
        if (it + 4 < 8) rp_issue(rp, it + 4);
        CH_FENCE();
    }
    CH_T(22);
    {
        const float* const v[6] = {a.b3, a.nn_g, a.nn_b, nullptr, nullptr, nullptr};
        nxt = fetch_consts(nullptr, v);
    }
    lds_barrier();                     // every wave is out of the last linear2 chunk (and of its constants' first use)
    row_stats(acc, scr, wave, lane, a.n4_eps, nmr, rstd);
    CH_T(23);
    norm_to_lds<false>(acc, nmr, rstd, vecp(3), vecp(4), rp, abuf, wave, lane, nullptr);
    lds_barrier();
    store_consts(nxt);                 // b3, norm1': read after the barrier that follows linear3
    }   // !FRONT
    // ================= x' = linear3(norm4(x)) + b3, no residual (model/model.py:344); FRONT: the last fusion linear of this
    // block's dancer over K = 1024 (model/model.py:526-528), whose output is layer 0's residual input
    CH_T(24);
    clear();
    if (FRONT)
        phase_n512<64>(acc, abuf, ws, lane);
    else if (LAST)
        phase_n512<32, true>(acc, abuf, ws, lane);
    else
        phase_n512<32>(acc, abuf, ws, lane);
    CH_T(25);
    lds_barrier();
    const int cb3 = col_base_bytes(fresh_s(wave), fresh_v(h));
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int n = 64 * wave + 32 * ni + 8 * gq + 4 * h;
            const f32x4_t b4 = lds4b(vecp(0) + cb3, 32 * (4 * ni + gq));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                f32x4_t o;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[mi][ni][4 * gq + t] += b4[t];
                    o[t] = acc[mi][ni][4 * gq + t];
                }
                if (LAST && a.out_ld > 0) {
                    // linear3 carries the final projection folded into it (engine.py: W_final W_3, two linear maps
                    // with nothing in between, model/model.py:344,623): columns [0, out_ld) ARE the network output
                    if (n < a.out_ld)
                        *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(a.h_out) + (long)mc[mi] * a.out_ld + n) = o;
                } else if (LAST) {   // a separate final projection reads bf16 rows (model/model.py:623)
                    uint2 pk;
                    pk.x = pack_bf2(o[0], o[1]);
                    pk.y = pack_bf2(o[2], o[3]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.h_out) + (long)mc[mi] * 512 + n) = pk;
                } else {
                    cb_store(xo, xrows, wave, 4 * ni + gq, trow[mi], h, o);
                }
            }
            CH_FENCE();
        }
    if (LAST) return;
    // ================= next layer: norm1 + rotary -> Q, K ; norm1 -> V (model/model.py:326,374-383,78-80)
    CH_T(26);
    rp_start(rp, a.rope, pos, a.rope_rows, a.rope_rows, wave, h);
    row_stats(acc, scr, wave, lane, a.nn_eps, nmr, rstd);
    CH_T(27);
    norm_to_lds<true>(acc, nmr, rstd, vecp(1), vecp(2), rp, abuf, wave, lane, smem + CH_ABUF2);
    lds_barrier();
    CH_T(28);
    clear();
    phase_n512<32, false, CH_QKV_R>(acc, abuf, ws, lane);
    CH_T(29);
    store_heads<true>(acc, a.q_out, a.scale_q, L, a.Lp, a.H, m0, M, wave, lane, smem, dn, dancer);
    clear();
    phase_n512<32, false, CH_QKV_R>(acc, abuf, ws, lane);
    CH_T(30);
    store_heads<false>(acc, a.k_out, 1.0f, L, a.Lp, a.H, m0, M, wave, lane, smem, dn, dancer);
    CH_T(31);
    clear();
    phase_n512<32, true, CH_QKV_R>(acc, smem + CH_ABUF2, ws, lane);
    CH_T(32);
    store_heads<false>(acc, a.v_out, 1.0f, L, a.Lp, a.H, m0, M, wave, lane, smem, dn, dancer);
    CH_T(33);
    CH_TC(61);
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int tcdiff_chain(const tcdiff_chain_args* a, hipStream_t stream) {
    if (!a || a->M <= 0 || a->L <= 0 || !a->A || !a->wstream) return TC_ERR_ARG;
    const bool front = a->mode == TC_CHAIN_FRONT;
    if (!front && a->L < 64) return TC_ERR_UNSUPPORTED;   // a 64-row block must touch at most two sequences
    const int want = a->mode == TC_CHAIN_A ? 64 : a->mode == TC_CHAIN_B ? 288 : a->mode == TC_CHAIN_B_LAST ? 192 :
                     a->mode == TC_CHAIN_FULL ? 352 : a->mode == TC_CHAIN_FULL_LAST ? 256 : front ? 160 : -1;
    if (want < 0 || a->n_stages != want) return TC_ERR_ARG;
    if (a->out_ld < 0 || a->out_ld % 4 || a->out_ld > 512) return TC_ERR_ARG;
    const void* ptrs[] = {a->A, a->wstream, a->ln_g, a->ln_b, a->film, a->xres, a->xout, a->n2_g, a->n2_b, a->rope,
                          a->q_out, a->b1, a->b2, a->film3, a->n4_g, a->n4_b, a->b3, a->nn_g, a->nn_b, a->k_out,
                          a->v_out, a->h_out, a->lnb_g, a->lnb_b, a->filmb, a->n3_g, a->n3_b, a->kf, a->vf};
    for (const void* p : ptrs)
        if (p && !al16(p)) return TC_ERR_ALIGN;
    if (front) {
        // A = bf16 [M frames][1024]; b3 = the 512 dn biases of the last fusion linear; xout = layer 0's residual input
        if (a->dn <= 0 || a->L % a->dn || a->L / a->dn < 8 || !a->b3 || !a->nn_g || !a->nn_b || !a->rope || !a->xout ||
            !a->q_out || !a->k_out || !a->v_out || a->H != 8 || a->Lp <= 0 || a->a_mod != 0)
            return TC_ERR_ARG;
        if ((long)a->M * 2048 >= (1L << 32)) return TC_ERR_ARG;
    } else {
        if (!a->ln_g || !a->ln_b || !a->film || a->film_ld % 4 || !a->xres || !a->xout || !a->n2_g || !a->n2_b) return TC_ERR_ARG;
        if ((long)(a->a_mod > 0 ? a->a_mod : a->M) * 1024 >= (1L << 32)) return TC_ERR_ARG;
    }
    const bool has_a = a->mode == TC_CHAIN_A || a->mode == TC_CHAIN_FULL || a->mode == TC_CHAIN_FULL_LAST;
    const bool has_b = a->mode != TC_CHAIN_A && !front;
    const bool full = a->mode == TC_CHAIN_FULL || a->mode == TC_CHAIN_FULL_LAST;
    const bool last = a->mode == TC_CHAIN_B_LAST || a->mode == TC_CHAIN_FULL_LAST;
    if (has_a && (!a->rope || a->H != 8)) return TC_ERR_ARG;
    if (a->mode == TC_CHAIN_A && (!a->q_out || a->Lp <= 0)) return TC_ERR_ARG;
    if (has_b && (!a->b1 || !a->b2 || !a->film3 || !a->n4_g || !a->n4_b || !a->b3)) return TC_ERR_ARG;
    if (has_b && !last &&
        (!a->rope || !a->nn_g || !a->nn_b || !a->q_out || !a->k_out || !a->v_out || a->H != 8 || a->Lp <= 0))
        return TC_ERR_ARG;
    if (last && !a->h_out) return TC_ERR_ARG;
    if (full && (!a->lnb_g || !a->lnb_b || !a->filmb || !a->n3_g || !a->n3_b || !a->kf || !a->vf || a->nkt <= 0 ||
                 a->Lk <= 0 || a->Lk > 32 * a->nkt || a->n_shared < 0))
        return TC_ERR_ARG;
    static tc_dev_state dev_state;
    const int n_cu = tc_device_once(dev_state, [](int) {
#if CH_XP > 0
        {
            const void* xf[2] = {reinterpret_cast<const void*>(chain_kernel<TC_CHAIN_FULL, CH_XP>),
                                 reinterpret_cast<const void*>(chain_kernel<TC_CHAIN_FULL_LAST, CH_XP>)};
            for (const void* f : xf)
                if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, CH_SMEM) != hipSuccess)
                    return hipErrorInvalidValue;
        }
#endif
        const void* fns[6] = {reinterpret_cast<const void*>(chain_kernel<TC_CHAIN_A>),
                              reinterpret_cast<const void*>(chain_kernel<TC_CHAIN_B>),
                              reinterpret_cast<const void*>(chain_kernel<TC_CHAIN_B_LAST>),
                              reinterpret_cast<const void*>(chain_kernel<TC_CHAIN_FULL>),
                              reinterpret_cast<const void*>(chain_kernel<TC_CHAIN_FULL_LAST>),
                              reinterpret_cast<const void*>(chain_kernel<TC_CHAIN_FRONT>)};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, CH_SMEM);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    });
    if (n_cu < 0) return n_cu;
    dim3 grid(((a->M + 63) / 64) * (front ? a->dn : 1));
    switch (a->mode) {
        case TC_CHAIN_FRONT: hipLaunchKernelGGL(chain_kernel<TC_CHAIN_FRONT>, grid, dim3(512), CH_SMEM, stream, *a); break;
        case TC_CHAIN_A: hipLaunchKernelGGL(chain_kernel<TC_CHAIN_A>, grid, dim3(512), CH_SMEM, stream, *a); break;
        case TC_CHAIN_B: hipLaunchKernelGGL(chain_kernel<TC_CHAIN_B>, grid, dim3(512), CH_SMEM, stream, *a); break;
        case TC_CHAIN_B_LAST: hipLaunchKernelGGL(chain_kernel<TC_CHAIN_B_LAST>, grid, dim3(512), CH_SMEM, stream, *a); break;
#if CH_XP > 0
        case TC_CHAIN_FULL:
            if (a->nkt == 5) hipLaunchKernelGGL((chain_kernel<TC_CHAIN_FULL, CH_XP>), grid, dim3(512), CH_SMEM, stream, *a);
            else hipLaunchKernelGGL(chain_kernel<TC_CHAIN_FULL>, grid, dim3(512), CH_SMEM, stream, *a);
            break;
        default:
            if (a->nkt == 5) hipLaunchKernelGGL((chain_kernel<TC_CHAIN_FULL_LAST, CH_XP>), grid, dim3(512), CH_SMEM, stream, *a);
            else hipLaunchKernelGGL(chain_kernel<TC_CHAIN_FULL_LAST>, grid, dim3(512), CH_SMEM, stream, *a);
            break;
#else
        case TC_CHAIN_FULL: hipLaunchKernelGGL(chain_kernel<TC_CHAIN_FULL>, grid, dim3(512), CH_SMEM, stream, *a); break;
        default: hipLaunchKernelGGL(chain_kernel<TC_CHAIN_FULL_LAST>, grid, dim3(512), CH_SMEM, stream, *a); break;
#endif
    }
    TC_CHECK_LAUNCH();
    return TC_OK;
}
