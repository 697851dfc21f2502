// ttnet_split.hip -- the N-env forward of the reference's actor / critic MLPs (DDPG/networks.py:55-68, 138-147) on the
// 16-bit MFMA of gfx950 at f32 accuracy ("split-f16").
//
// Why: the f32-input MFMA runs at 1/16 of the f16/bf16 rate (155 TFLOP/s vs ~2.5 PFLOP/s dense), and fc2 (400 -> 300)
// is 93 % of the network's arithmetic.  An f32 number x (24 significand bits) is, to within its own rounding error, the
// sum of TWO f16 numbers when both conversions round to nearest: h = rn16(x) leaves |x - h| <= 2^-12 |x| and
// m = rn16(x - h) leaves |x - h - m| <= 2^-24 |x| (11 + 1 + 11 + 1 bits; the remainder x - h is exact in f32).  A
// product of two f16 numbers is exact in f32 (22 bits), so
//     x*w = h*h' + (h*m' + m*h') + [m*m' + (x - h - m)*w + x*(w - h' - m')]
// where every term of the bracket is below 2^-24 |x*w|: three f16 MFMAs with f32 accumulation reproduce the f32 product
// to within ~3 of its own rounding errors, at 3/16 of the f32 MFMA's cycles (the round-1 kernel used three bf16 pieces
// and six MFMAs per product block).  f16 has 5 exponent bits, so both operands are scaled by powers of two (exact) to
// keep their `m` pieces normal: activations by 2^4, weights by 2^6; the accumulator is scaled back by 2^-10 (exact)
// where the bias is added.  Ranges: |activation| < 4094, |weight| < 1023 -- beyond them a piece becomes inf and the row's
// output NaN (loud, not silently wrong); pieces below the normal range carry an absolute error <= 2^-25 / scale.
//
// Orientation: everything is computed TRANSPOSED (neurons on the MFMA's M axis, envs on N), with
// v_mfma_f32_32x32x16_f16.  Its 32x32 result has the env on the lane (lane & 31) and 16 neurons in the lane's
// registers, which is exactly the B-operand layout of the next product that sums over neurons: the layer-1 result
// goes through LayerNorm + ReLU in registers and feeds layer 2 with no LDS round trip and no lane movement
// (the k order inside a step is permuted; the packed weights are laid out to match).  Both LayerNorms and the
// 300 -> 1 head reduce over registers plus ONE cross-lane exchange (lane ^ 32).
//
//   workgroup = 4 waves = 128 envs (32 per wave); per wave: h1 = 13 tiles x 16 registers (f32), layer-2
//   accumulators 10 tiles x 16 registers; fc2 is streamed once per workgroup, as pre-split f16 planes in MFMA
//   fragment order, through a 7-slot LDS ring shared by the four waves (20 KB per k16 step), filled by LDS-DMA five
//   steps ahead of its use.
//   Layer 1 (K = 23 + the bias as a 24th input that is always 1) runs on the same three-product scheme from pre-split
//   fc1 fragments, also brought into LDS by DMA.
#include "ttnet_common.h"
#include "ttnet_pack.h"
#include "ttstamps.h"

namespace ttnet {
namespace {

#ifdef TT_STAMPS   // diagnostic build only: clocks of thread 0 of every workgroup at the phase boundaries
__device__ unsigned long long g_nstamps[16];
__device__ unsigned long long g_bstamps[4096 * 8];      // [block][phase] wall clock (100 MHz); [block][7] = XCC id
#define NSTAMP(i) do { if (threadIdx.x == 0) { const unsigned long long w_ = wall_clock64();                                \
        if (blockIdx.x < 4096) { g_bstamps[blockIdx.x * 8 + (i)] = w_;                                                      \
            if ((i) == 0) g_bstamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); }     \
        if (blockIdx.x == 0) { g_nstamps[i] = w_; g_nstamps[8 + i] = clock64(); } } } while (0)
// begin / end of every tile a workgroup takes: [block][round of the tile loop][2] (tools/step_timeline.py)
__device__ unsigned long long g_tile[512][4][2];
__device__ TTLog g_log_tile;
#define TSTAMP(round, e) do { if (threadIdx.x == 0 && blockIdx.x < 512 && (round) < 4) g_tile[blockIdx.x][round][e] = wall_clock64(); } while (0)
#else
#define NSTAMP(i) do { } while (0)
#define TSTAMP(round, e) do { } while (0)
#endif

__global__ __launch_bounds__(256) void k_split_pack(const Weights W, const bool critic, unsigned char *__restrict__ ws,
                                                    unsigned char *__restrict__ ws_alt, long long *__restrict__ bump,
                                                    const RingCursor cur) {
    split_pack_body(W, critic, ws, ws_alt, bump, cur, blockIdx.x * 256 + threadIdx.x);
    publish_image(cur, PACK_BLOCKS);
}

// What opens a pipelined vector step, in ONE launch (two small kernels would each cost their ~4 us of launch and a
// dependency gap on the loop's critical path): the policy's image from the actor's current weights, and the first batch of
// this step's learn() -- or all of its batches, one per update (RingSample.draws) -- from the replay ring (four sampled
// transitions per 256-thread workgroup).
__global__ __launch_bounds__(256) void k_pack_and_sample(const Weights W, const bool critic, unsigned char *__restrict__ ws,
                                                         unsigned char *__restrict__ ws_alt, const RingSample R,
                                                         const RingCursor cur) {
    if ((int)blockIdx.x < PACK_BLOCKS) {
        split_pack_body(W, critic, ws, ws_alt, nullptr, cur, blockIdx.x * 256 + threadIdx.x);
        publish_image(cur, PACK_BLOCKS);
        return;
    }
    const int b = ((int)blockIdx.x - PACK_BLOCKS) * 4 + (threadIdx.x >> 6);
    if (b < R.batch * R.draws) ring_sample_row(R, b, threadIdx.x & 63);
}

// LDS reads through laundered address-space-3 bases.  The ring + vectors span 152 KB and a ds_read's immediate offset reaches
// 64 KB, so three byte bases (0, 64 KB, 128 KB; + this lane's 16 bytes) cover every fragment with an immediate; the bases pass
// through an empty asm at the top of every tile so that the compiler can neither hoist the ~300 fragment addresses of a tile out
// of the tile loop nor keep them live across it (the loop form spilled ~400 registers that way).
using lds_b = const __attribute__((address_space(3))) unsigned char;
using v4u = __attribute__((ext_vector_type(4))) unsigned;
struct LdsBases {
    lds_b *b0, *b1, *b2;
};
__device__ __forceinline__ uint4 lds_frag(const LdsBases &L, const int off) {      // off: compile-time byte offset (+ lane * 16)
    lds_b *p = (off >> 16) == 0 ? L.b0 : ((off >> 16) == 1 ? L.b1 : L.b2);
    const v4u t = *(const __attribute__((address_space(3))) v4u *)(p + (off & 0xFFFF));
    return make_uint4(t[0], t[1], t[2], t[3]);
}

// One workgroup = 4 waves = 128 envs per TILE; a workgroup takes the tiles tile0 + blockIdx.x, + gridDim.x, ... < tile_end in
// turn.  From the second tile on, the tile's observations, packed fc1 and the first two k16 steps of fc2 are requested before
// the PREVIOUS tile's epilogue and land behind it; the per-neuron vectors stay in LDS.
template <bool CRITIC>
__global__ __launch_bounds__(256, 1) void k_mlp_split(const int n, const int tile0, const int tile_end,
                                                      int *__restrict__ cursor_e, const long long *__restrict__ step_e,
                                                      const float *__restrict__ obs,
                                                      const unsigned char *__restrict__ ws,
                                                      const unsigned char *__restrict__ ws_alt,
                                                      const float *__restrict__ action, float *__restrict__ out,
                                                      const ActArgs act) {
    // The first 16 dwords of the arguments reach the wave in SGPRs (kernarg preload, build.py): n .. action.  cursor_e / step_e
    // repeat act.cursor / act.step_dev there, so that the launch's first loads -- the step counter and the image epochs --
    // leave before the rest of the argument segment (the struct) has been fetched.
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const float *p1_s = reinterpret_cast<const float *>(lds_raw + RING_BYTES);              // g1' | be1'       [2][416]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // this lane's view of the per-neuron vectors, its address held in a register: the region lies beyond the 64 KB a
    // ds_read's immediate offset reaches from address 0, and left to itself the compiler forms every address with an add
    using lds_f = const __attribute__((address_space(3))) float;
    using v4 = __attribute__((ext_vector_type(4))) float;
    auto lds4 = [](lds_f *q) -> float4 {
        const v4 t = *(const __attribute__((address_space(3))) v4 *)q;
        return make_float4(t[0], t[1], t[2], t[3]);
    };
    lds_f *pv1_0 = (lds_f *)(p1_s + 4 * h);
    LdsBases LB0;
    LB0.b0 = (lds_b *)lds_raw + lane * 16;
    LB0.b1 = LB0.b0 + 65536;
    LB0.b2 = LB0.b0 + 131072;
    // ring addressing: the running step's slots follow from the ring's step counter (the same numbers the opening pack launch
    // leaves in the cursor buffer); the first workgroup leaves them where the env step reads them.  The IMAGE of this step
    // (that of the step's parity: the other one may be being written for the next step) comes from the opening pack launch,
    // which need not be ordered before this launch: await_image (ttnet_common.h) -- behind the first tile's observation loads,
    // which do not depend on it.
    const EpochPair epochs = await_image_early(cursor_e);        // (requested beside the step counter, not behind it)
    const long long kstep = cursor_e ? *step_e : 0;
    const bool odd = act.cursor && (kstep & 1);
    const bool local = act.cursor && act.ring_slots > 0;
    const int slot_t = local ? (int)(kstep % act.ring_slots) : 0;
    const float *obs_base = local ? obs + (size_t)slot_t * act.ring_n * IN : obs;      // (re-read from the cursor below when !local)
    // LDS is filled by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = one 1 KB piece per wave-instruction, no
    // staging registers).  The statements are inline asm, so hipcc neither counts them nor drains them at a barrier:
    // each wave retires its own pieces with an s_waitcnt vmcnt before the barrier that precedes their first read.
    // A piece = global bytes [base + voff .. + 1 KB) -> LDS [lds_wave + dst_const .. + 1 KB).  The global side is ONE SGPR base
    // (the image) plus a per-lane VGPR offset that carries everything else (lane * 16, the wave's share, the piece), the LDS side
    // a per-wave SGPR plus a constant added INSIDE the asm: no scalar address arithmetic is left to the compiler, which forms
    // 64-bit scalar adds on the vector ALU when SCC is live and then cannot feed them to an "s" operand.
    const unsigned lds_base0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds_raw;
    auto dma_piece = [&](const unsigned char *base, const unsigned voff, const unsigned lds_wave, const unsigned dst_const)
                         __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_wave), "s"(dst_const) : "memory", "scc");
    };
    const unsigned voff_w2 = lane * 16 + wave * 5120 + WS_W2;                              // this wave's 5 pieces of a k16 step
    const unsigned voff_w1 = lane * 16 + wave * (W1_PIECES / 4) * 1024 + WS_W1;            // ... 13 pieces of packed fc1
    const unsigned voff_vec = lane * 16 + wave * (VEC_PIECES / 4) * 1024 + WS_VEC;         // ... 3 pieces of the vectors
    const unsigned ldsw_w2_0 = lds_base0 + wave * 5120, ldsw_w1_0 = lds_base0 + W1_OFF + wave * (W1_PIECES / 4) * 1024;
    // this wave's share of packed fc1 (13 pieces) / of k16 step s of fc2 (5 pieces), from image `w`
    auto issue_fc1 = [&](const unsigned char *w, const unsigned ldsw_w1) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < W1_PIECES / 4; ++i) dma_piece(w, voff_w1 + i * 1024, ldsw_w1, i * 1024);
    };
    auto issue_step_piece = [&](const unsigned char *w, const unsigned ldsw_w2, const int s, const int i) __attribute__((always_inline)) {
        dma_piece(w, voff_w2 + s * CHUNK_BYTES + i * 1024, ldsw_w2, (s % RING) * CHUNK_BYTES + i * 1024);
    };
    // unconditional loads from clamped (always valid) addresses, selected afterwards: 16 loads in flight at once
    // instead of 16 exec-masked round trips
    float xo[S1][8];                                                          // obs^T as the B operand: input 16 s + 8 h + j
    // choose_action's noise inputs of the tile's rows -- the OU state and the previous step's done flag (DDPG_agent.py:36-49,
    // noise.py:13-17, trainv2.py:492) -- are requested WITH the observations: read where the head's output is known (finish_row)
    // they were two dependent round trips to memory at the very end of every tile, with nothing left to hide them behind
    float ou_pre = 0.f;
    unsigned dp_pre = 0u;
    const bool with_noise = !CRITIC && act.ou != nullptr;                     // (uniform over the launch)
    const uint8_t *dprev = act.done_prev;                                    // the done flags that restart a row's noise, or nullptr
    float *araw = act.act_raw;
    auto load_obs = [&](const int tile) __attribute__((always_inline)) {
        const int row = tile * ROWS + wave * WROWS + r;
        const int rowc = row < n ? row : n - 1;
        const float *orow = obs_base + (size_t)rowc * IN;
#pragma unroll
        for (int s = 0; s < S1; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * s + 8 * h + j;
                xo[s][j] = orow[k < IN ? k : IN - 1];
            }
        if (with_noise) {
            ou_pre = act.ou[rowc];
            dp_pre = dprev ? (unsigned)dprev[rowc] : 0u;
        }
    };

    NSTAMP(0);
    // ---- prologue of the FIRST tile: this lane's observation features first (ordinary loads: the compiler waits for them on
    // its own count), then packed fc1 + the per-neuron vectors, then k16 steps 0 and 1 of fc2 by DMA
    int tile = tile0 + (int)blockIdx.x;
    if (tile >= tile_end) return;
    if (local) {                          // the ring's slots of this step follow from the step counter: no look at the cursor
        const int sl = act.ring_slots;
        const int slot_prev = slot_t == 0 ? sl - 1 : slot_t - 1;          // (one 64-bit modulo per launch, not three)
        dprev = kstep > 0 ? act.done_prev + (size_t)slot_prev * act.ring_n : nullptr;
        araw = act.act_raw + (size_t)slot_t * act.ring_n;
    }
    if (!act.cursor || local) load_obs(tile);
    await_image(act.cursor, act.step_dev, epochs);
    const unsigned char *wsl = (odd && ws_alt) ? ws_alt : ws;
    if (act.cursor && !local) {          // (a caller that gave no slot count: the cursor the pack launch wrote)
        obs_base = resolve_obs(act, obs);
        const int *c = cursor_of(act);
        dprev = c[3] ? act.done_prev + (size_t)c[2] * act.ring_n : nullptr;
        araw = act.act_raw + (size_t)c[0] * act.ring_n;
        load_obs(tile);
    }
    const unsigned long long noise_step = act.step + (act.step_dev ? (unsigned long long)*act.step_dev : 0ull);
    if (act.cursor && tile0 == 0 && blockIdx.x == 0 && tid == 0)      // this launch has begun: the step chain is at step kstep
        __hip_atomic_store(act.cursor + CUR_PROGRESS, (int)(kstep + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (act.cursor && tile0 == 0 && blockIdx.x == 0 && tid < 4) {
        const int sl = act.ring_slots;
        act.cursor[tid] = !local ? cursor_of(act)[tid]
                                 : (tid == 0 ? slot_t : tid == 1 ? (slot_t + 1 == sl ? 0 : slot_t + 1)
                                                      : tid == 2 ? (slot_t == 0 ? sl - 1 : slot_t - 1) : (kstep > 0 ? 1 : 0));
    }
    issue_fc1(wsl, ldsw_w1_0);
#pragma unroll
    for (int i = 0; i < VEC_PIECES / 4; ++i)
        dma_piece(wsl, voff_vec + i * 1024, lds_base0 + RING_BYTES + wave * (VEC_PIECES / 4) * 1024, i * 1024);
#pragma unroll
    for (int i = 0; i < 10; ++i) issue_step_piece(wsl, ldsw_w2_0, i / 5, i % 5);
    bool first = true;
#ifdef TT_STAMPS
    int round = 0;
    if (threadIdx.x == 0 && blockIdx.x < 512) for (int q = 0; q < 4; ++q) g_tile[blockIdx.x][q][0] = g_tile[blockIdx.x][q][1] = 0ull;
    if (threadIdx.x == 0 && blockIdx.x < 512) g_tile[blockIdx.x][0][0] = wall_clock64();
#endif

#pragma unroll 1
    for (;;) {
        // tile-invariant addresses, opaque per tile (see LdsBases)
        const unsigned char *wsl_cur = wsl;
        unsigned ldsw_w2 = ldsw_w2_0, ldsw_w1 = ldsw_w1_0;
        LdsBases LB = LB0;
        lds_f *pv1 = pv1_0;
        asm volatile("" : "+s"(wsl_cur), "+s"(ldsw_w2), "+s"(ldsw_w1), "+v"(LB.b0), "+v"(LB.b1), "+v"(LB.b2), "+v"(pv1));
        lds_f *pv2 = pv1 + 2 * H1P;
        const int row = tile * ROWS + wave * WROWS + r;                       // this lane's env (lanes r and r+32 share it)
#pragma unroll
        for (int s = 0; s < S1; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * s + 8 * h + j;
                xo[s][j] = k < IN ? (row < n ? xo[s][j] : 0.f) : (k == IN ? 1.f : 0.f);
            }
        // this row's noise for the step (independent of the network: see load_obs), from the values requested a tile ago
        float ou_next = 0.f;
        if (with_noise) ou_next = ou_advance(row < n ? row : n - 1, ou_pre, dp_pre != 0u, act, noise_step);
        uint4 xb[S1][2];                                                      // the observation's h and m pieces
#pragma unroll
        for (int s = 0; s < S1; ++s) {
            uint32_t ph[4], pm[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) split2(xo[s][2 * p] * SX, xo[s][2 * p + 1] * SX, ph[p], pm[p]);
            xb[s][0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
            xb[s][1] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        }
        // first tile: fc1 + vectors landed, fc2 steps 0, 1 may still fly.  Later tiles: everything of this tile was requested
        // before the previous tile's epilogue; that epilogue's stores are among the outstanding operations and stores and
        // loads retire in no fixed order, so nothing short of 0 tells that the DMA pieces have landed
        if (first) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (first) NSTAMP(1);

        // ---- layer 1: acc1[t][v] = SX*SW * (pre-activation + bias) of neuron 32t + 8(v>>2) + 4h + (v&3), env `row`
        f32x16 acc1[T1];
#pragma unroll
        for (int t = 0; t < T1; ++t) {
#pragma unroll
            for (int v = 0; v < 16; ++v) acc1[t][v] = 0.f;
#pragma unroll
            for (int s = 0; s < S1; ++s) {
                const uint4 ah = lds_frag(LB, W1_OFF + ((s * T1 + t) * 2) * 1024), am = lds_frag(LB, W1_OFF + ((s * T1 + t) * 2 + 1) * 1024);
                acc1[t] = mfma_f16(am, xb[s][0], acc1[t]);                    // small terms first
                acc1[t] = mfma_f16(ah, xb[s][1], acc1[t]);
                acc1[t] = mfma_f16(ah, xb[s][0], acc1[t]);
            }
        }
        if (first) NSTAMP(2);
        // every wave is done with packed fc1: its slots now take k16 steps 2..5 of fc2 (they land during LayerNorm 1)
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 10; i < 5 * (AHEAD + 1); ++i) issue_step_piece(wsl_cur, ldsw_w2, i / 5, i % 5);
        uint32_t hb[STEPS_FULL][4], mb[STEPS_FULL][4];
        // LayerNorm(400) (biased variance, eps 1e-5) on the SCALED pre-activations: mean and deviations scale with them,
        // 1/sigma absorbs the scale; then gamma*SX, beta*SX and ReLU give the layer-2 operand already scaled by SX.
        // Tile 12 holds neurons 384..399 in v < 8.
        // Only the two statistics passes stand between the layers.  The third pass -- normalise, ReLU, split into the two f16
        // pieces layer 2 multiplies with -- needs nothing but mean and 1/sigma and produces the B operand of ONE k16 step from
        // 16 accumulator registers, so it runs INSIDE layer 2 (round 4), two steps ahead of the step that multiplies with it, in
        // the issue slots the matrix pipe leaves free (ln_chunk below).  hb[s] / mb[s] are the B fragments (h, m planes) of k16
        // step s -- element j of lane half h is neuron 16 s + 8 (j >> 2) + 4 h + (j & 3), i.e. registers 8 (s & 1) .. + 7 of tile
        // s >> 1, the order the packed fc2 fragments are laid out in.
        f32x2 nmean, rstdp;
        {
            f32x2 sp = pk(0.f, 0.f);
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int v = 0; v < (t == T1 - 1 ? 8 : 16); v += 4)
                    sp += pk(acc1[t][v], acc1[t][v + 1]) + pk(acc1[t][v + 2], acc1[t][v + 3]);
            const float s = sp[0] + sp[1];
            const float mean = (s + __shfl_xor(s, 32)) * (1.f / H1);
            nmean = pk(-mean, -mean);
            f32x2 ssp = pk(0.f, 0.f);
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int v = 0; v < (t == T1 - 1 ? 8 : 16); v += 2) {
                    const f32x2 d = pk(acc1[t][v], acc1[t][v + 1]) + nmean;   // (formed again where it is used: same operation, same bits)
                    ssp = pk_fma(d, d, ssp);
                }
            const float ss = ssp[0] + ssp[1];
            const float var = (ss + __shfl_xor(ss, 32)) * (1.f / H1) * (UNSCALE * UNSCALE);
            const float rstd = rsqrtf(var + 1e-5f) * UNSCALE;
            rstdp = pk(rstd, rstd);
        }
        // One GROUP = four neurons of this lane (registers 4i .. 4i+3 of tile t; group number G = 4t + i = 2 * step + half), in
        // five chunks of 3-4 vector instructions; a chunk fits in the shadow of one MFMA.  gamma/beta of group G + 1 are read
        // from LDS in chunk 2 of group G (three tiles of MFMAs before their use).
        constexpr int GROUPS = 2 * STEPS;                                     // 50: tile 12 has groups 48, 49 only
        float4 gq[2], bq[2];
        f32x2 ya, yb;
        float y0, y1, y2, y3;
        auto ln_read = [&](const int G) __attribute__((always_inline)) {
            gq[G & 1] = lds4(pv1 + 8 * G);                                    // 32 t + 8 i
            bq[G & 1] = lds4(pv1 + H1P + 8 * G);
        };
        auto ln_chunk = [&](const int G, const int k) __attribute__((always_inline)) {
            const int t = G >> 2, i = G & 3, st = G >> 1, e = 2 * (G & 1);
            const float4 g = gq[G & 1], be = bq[G & 1];
            if (k == 0) ya = pk_fma((pk(acc1[t][4 * i], acc1[t][4 * i + 1]) + nmean) * rstdp, pk(g.x, g.y), pk(be.x, be.y));
            if (k == 1) yb = pk_fma((pk(acc1[t][4 * i + 2], acc1[t][4 * i + 3]) + nmean) * rstdp, pk(g.z, g.w), pk(be.z, be.w));
            if (k == 2) {
                y0 = fmaxf(ya[0], 0.f); y1 = fmaxf(ya[1], 0.f); y2 = fmaxf(yb[0], 0.f); y3 = fmaxf(yb[1], 0.f);
                if (G + 1 < GROUPS) ln_read(G + 1);
            }
            if (k == 3) split2(y0, y1, hb[st][e], mb[st][e]);
            if (k == 4) split2(y2, y3, hb[st][e + 1], mb[st][e + 1]);
        };
        // steps 0 and 1 (groups 0..3) before the first MFMA
        ln_read(0);
#pragma unroll
        for (int G = 0; G < 4; ++G)
#pragma unroll
            for (int k = 0; k < 5; ++k) ln_chunk(G, k);

        if (first) NSTAMP(3);
        // ---- layer 2: 25 k16 steps x 10 neuron tiles x 3 f16 MFMAs; the h1 registers are the B operand
        f32x16 acc2[T2];
#pragma unroll
        for (int u = 0; u < T2; ++u)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc2[u][v] = 0.f;
        // Hand-pipelined issue order, pinned with sched_barrier(0) between every pair of instruction groups: each of the
        // three MFMAs of a tile is followed by one small job that issues while the matrix pipe is busy (an MFMA holds the
        // wave's issue port for 8 of its 32 cycles) -- the two fragment reads of the tile two ahead, and in tiles 5..9 one
        // LDS-DMA piece of step s + 6 -- and, behind the first MFMA of every tile, one chunk of LayerNorm 1's last pass for the B
        // operand of step s + 2 (a tenth of a step's worth: 3-4 vector instructions).
        // One barrier per step, in the middle (before tile 5): it publishes step s + 1 (every wave has waited for its own
        // pieces of it with a COUNTED vmcnt that leaves the four younger steps in flight) and tells that every wave is past
        // step s - 1, whose slot the DMA of step s + 6 overwrites.  Fragments are read TWO tiles ahead
        // (tiles 0, 1 of step s + 1 during tiles 8, 9 of step s, after that barrier), so a read has six MFMAs to land.
#define SB __builtin_amdgcn_sched_barrier(0)
        asm volatile("s_waitcnt vmcnt(20)" ::: "memory");      // steps 0 and 1 (this wave's pieces); 2..5 may still fly
        __builtin_amdgcn_s_barrier();
        // A fragments (h, m) of three consecutive tiles: the current one, the next, and the one being read (two ahead);
        // rotated by renaming at the end of every tile (the loops are fully unrolled: no moves)
        uint4 c0h = lds_frag(LB, 0), c0m = lds_frag(LB, 1024), c1h = lds_frag(LB, 2048), c1m = lds_frag(LB, 3072), c2h = c0h, c2m = c0m;
        SB;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const uint4 vh = make_uint4(hb[s][0], hb[s][1], hb[s][2], hb[s][3]), vm = make_uint4(mb[s][0], mb[s][1], mb[s][2], mb[s][3]);
            const int slot = (s % RING) * CHUNK_BYTES, nslot = ((s + 1) % RING) * CHUNK_BYTES;
            // (per step, so that the step's five DMA addresses are formed here and not all 125 at the top of the tile)
            const unsigned char *wsl_s = wsl_cur;
            unsigned lds_s = ldsw_w2;
            asm volatile("" : "+s"(wsl_s), "+s"(lds_s));
#pragma unroll
            for (int u = 0; u < T2; ++u) {
                if (u == T2 / 2) {
                    // step s + 1 must have landed; the steps issued after it (up to four) may still be in flight
                    switch (STEPS - 2 - s < 4 ? STEPS - 2 - s : 4) {
                        case 4: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
                        case 3: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
                        case 2: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                        case 1: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    }
#ifndef TT_DBG_NOBAR
                    __builtin_amdgcn_s_barrier();
#endif
                    SB;
                }
                const bool more = u + 2 < T2 || s + 1 < STEPS;                // is there a tile two ahead to prefetch
                const int nx = u + 2 < T2 ? slot + (u + 2) * 2048 : nslot + (u + 2 - T2) * 2048;
                acc2[u] = mfma_f16(c0h, vm, acc2[u]); SB;      // small terms first
                if (more) c2h = lds_frag(LB, nx);
                if (s + 2 < STEPS) ln_chunk(2 * (s + 2) + u / 5, u % 5);      // LayerNorm 1's last pass for step s + 2
                SB;
                acc2[u] = mfma_f16(c0m, vh, acc2[u]); SB;
                if (more) c2m = lds_frag(LB, nx + 1024);
                SB;
                acc2[u] = mfma_f16(c0h, vh, acc2[u]); SB;
#ifndef TT_DBG_NODMA
                if (u >= 5 && s + RING - 1 < STEPS) issue_step_piece(wsl_s, lds_s, s + RING - 1, u - 5);
#endif
                SB;
                c0h = c1h; c0m = c1m; c1h = c2h; c1m = c2m;
            }
        }
#undef SB

        if (first) NSTAMP(4);
        // ---- the next tile of this workgroup: its observations, packed fc1 and fc2 steps 0, 1 are requested NOW (every wave is
        // past its last fragment read: the ring is free) and land behind the epilogue below
        const int next = tile + (int)gridDim.x;
        const bool more_tiles = next < tile_end;
        if (more_tiles) {
            __builtin_amdgcn_s_barrier();
            load_obs(next);
            issue_fc1(wsl_cur, ldsw_w1);
#pragma unroll
            for (int i = 0; i < 10; ++i) issue_step_piece(wsl_cur, ldsw_w2, i / 5, i % 5);
        }
        // ---- epilogue: scale back + bias, LayerNorm(300), (critic: + action_value(a)), ReLU, head; acc2[u][v] is neuron
        // 32u + 8(v>>2) + 4h + (v&3); 300 = 9*32 + 12, so groups of four are all real or all padding
        // (the per-neuron vectors are read from LDS one tile ahead of their use, as in LayerNorm 1)
        f32x2 s2p = pk(0.f, 0.f);
        {
            float4 bq[2][4];
            auto ldb = [&](const int u, float4 (&b)[4]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = lds4(pv2 + 32 * u + 8 * i);   // zero beyond 300
            };
            ldb(0, bq[0]);
#pragma unroll
            for (int u = 0; u < T2; ++u) {
                if (u + 1 < T2) ldb(u + 1, bq[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 b = bq[u & 1][i];
                    const f32x2 xa = pk_fma(pk(acc2[u][4 * i], acc2[u][4 * i + 1]), pk(UNSCALE, UNSCALE), pk(b.x, b.y));
                    const f32x2 xb2 = pk_fma(pk(acc2[u][4 * i + 2], acc2[u][4 * i + 3]), pk(UNSCALE, UNSCALE), pk(b.z, b.w));
                    acc2[u][4 * i] = xa[0]; acc2[u][4 * i + 1] = xa[1]; acc2[u][4 * i + 2] = xb2[0]; acc2[u][4 * i + 3] = xb2[1];
                    s2p += xa + xb2;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const float s2 = s2p[0] + s2p[1];
        const float mean2 = (s2 + __shfl_xor(s2, 32)) * (1.f / H2);
        const f32x2 nmean2 = pk(-mean2, -mean2);
        // deviations in place (the last pass needs nothing else of the pre-activations); padding groups do not count
        f32x2 ss2p = pk(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < T2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool real = 32 * u + 8 * i + 4 * h < H2;
                const f32x2 keep = real ? pk(1.f, 1.f) : pk(0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f32x2 d = pk(acc2[u][4 * i + j], acc2[u][4 * i + j + 1]) + nmean2;
                    acc2[u][4 * i + j] = d[0]; acc2[u][4 * i + j + 1] = d[1];
                    const f32x2 dk = (32 * u + 8 * i + 4 < H2) ? d : d * keep;      // only the last tiles can hold padding
                    ss2p = pk_fma(dk, dk, ss2p);
                }
            }
        const float ss2 = ss2p[0] + ss2p[1];
        const float rstd2 = rsqrtf((ss2 + __shfl_xor(ss2, 32)) * (1.f / H2) + 1e-5f);
        const f32x2 rstd2p = pk(rstd2, rstd2);
        const float av = (CRITIC && row < n) ? action[row] : 0.f;
        f32x2 dotp = pk(0.f, 0.f);
        {
            constexpr int NV = CRITIC ? 5 : 3;                  // gamma2, beta2, w3 (, wa, ba)
            float4 vq[2][4][NV];
            auto ldv = [&](const int u, float4 (&v)[4][NV]) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < NV; ++q) v[i][q] = lds4(pv2 + (q + 1) * H2P + 32 * u + 8 * i);
            };
            ldv(0, vq[0]);
#pragma unroll
            for (int u = 0; u < T2; ++u) {
                if (u + 1 < T2) ldv(u + 1, vq[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 g = vq[u & 1][i][0], be = vq[u & 1][i][1], w3 = vq[u & 1][i][2];
                    f32x2 ya = pk_fma(pk(acc2[u][4 * i], acc2[u][4 * i + 1]) * rstd2p, pk(g.x, g.y), pk(be.x, be.y));
                    f32x2 yb = pk_fma(pk(acc2[u][4 * i + 2], acc2[u][4 * i + 3]) * rstd2p, pk(g.z, g.w), pk(be.z, be.w));
                    if (CRITIC) {
                        const float4 wa = vq[u & 1][i][NV - 2], ba = vq[u & 1][i][NV - 1];
                        ya += pk_fma(pk(av, av), pk(wa.x, wa.y), pk(ba.x, ba.y));
                        yb += pk_fma(pk(av, av), pk(wa.z, wa.w), pk(ba.z, ba.w));
                    }
                    dotp = pk_fma(pk(fmaxf(ya[0], 0.f), fmaxf(ya[1], 0.f)), pk(w3.x, w3.y), dotp);      // w3 = 0 on padding
                    dotp = pk_fma(pk(fmaxf(yb[0], 0.f), fmaxf(yb[1], 0.f)), pk(w3.z, w3.w), dotp);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const float dot = dotp[0] + dotp[1];
        const float v = dot + __shfl_xor(dot, 32) + lds4(pv1 - 4 * h + VEC_FLOATS).x;
        if (h == 0 && row < n) {
            if (with_noise) finish_row_noise(row, v, out, act, ou_next, araw);
            else finish_row<CRITIC>(row, v, out, act);
        }
        if (first) NSTAMP(5);
#ifdef TT_STAMPS
        TSTAMP(round, 1);
        if (threadIdx.x == 0 && blockIdx.x < 512 && round < 4) tt_log_add(g_log_tile, g_tile[blockIdx.x][round][0], g_tile[blockIdx.x][round][1]);
        ++round;
        if (more_tiles) TSTAMP(round, 0);
#endif
        if (!more_tiles) break;
        tile = next;
        first = false;
    }
}

}  // namespace

#ifdef TT_STAMPS
int split_debug_stamps(unsigned long long *out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_nstamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -3;
}
int split_debug_tiles(unsigned long long *out4096) {
    return hipMemcpyFromSymbol(out4096, HIP_SYMBOL(g_tile), sizeof(unsigned long long) * 512 * 4 * 2) == hipSuccess ? 0 : -3;
}
int split_debug_log(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_log_tile), sizeof(TTLog)) != hipSuccess) return -3;
    if (reset) { const unsigned long long z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_log_tile), &z, sizeof(z)) != hipSuccess) return -3; }
    return 0;
}
int split_debug_poll(unsigned long long *out4) {
    return hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_poll), sizeof(unsigned long long) * 4) == hipSuccess ? 0 : -3;
}
int split_debug_block_stamps(unsigned long long *out, int nblocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bstamps), sizeof(unsigned long long) * 8 * (nblocks < 4096 ? nblocks : 4096)) == hipSuccess ? 0 : -3;
}
#endif

size_t split_ws_bytes() { return (size_t)WS_BYTES; }

int split_pack(const tt_mlp_weights *w, bool critic, void *ws, long long *bump, const RingCursor &cur, hipStream_t stream) {
    // (the second image takes part only when the caller packs into the struct's own workspace)
    unsigned char *alt = ws == w->split_ws ? reinterpret_cast<unsigned char *>(w->split_ws_alt) : nullptr;
    hipLaunchKernelGGL(k_split_pack, dim3(PACK_BLOCKS), dim3(256), 0, stream, to_weights(w), critic,
                       reinterpret_cast<unsigned char *>(ws), alt, bump, cur);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int split_pack_and_sample(const tt_mlp_weights *w, bool critic, void *ws, const RingSample &R, const RingCursor &cur,
                          hipStream_t stream) {
    unsigned char *alt = ws == w->split_ws ? reinterpret_cast<unsigned char *>(w->split_ws_alt) : nullptr;
    hipLaunchKernelGGL(k_pack_and_sample, dim3(PACK_BLOCKS + (R.batch * R.draws + 3) / 4), dim3(256), 0, stream, to_weights(w), critic,
                       reinterpret_cast<unsigned char *>(ws), alt, R, cur);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

template <bool CRITIC>
static int launch_split(int n, const float *obs, const float *action, const tt_mlp_weights *w, float *out,
                        const ActArgs &act, hipStream_t stream) {
    static bool attr[64] = {};      // per device: a function attribute set on one device says nothing about another
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TT_EHIP;
    if (!attr[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_split<CRITIC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return TT_EHIP;
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cus[dev] = c;
        attr[dev] = true;
    }
    // One workgroup per CU is resident (LDS), and a workgroup takes its tiles in turn (tile, tile + grid, ...).
    // max_workgroups > 0: never more than that many CUs are busy with this forward (the rest stay free for launches on other
    // streams): the tiles are shared out in rounds of at most that many, ONE launch of the equal share (512 tiles, limit 192 ->
    // 171 workgroups x 3 tiles).  capped_grids > 0: only that many rounds are capped -- what the cap makes room for is over by
    // then -- and the rest of the tiles go out in a second launch over the whole chip.  0: one launch over the whole chip.
    const int ntiles = (n + ROWS - 1) / ROWS;
    const int chip = cus[dev];
    int capped = 0;
    if (w->max_workgroups > 0) {
        const int limit = w->max_workgroups;
        capped = ntiles;
        if (w->capped_grids > 0 && (long long)w->capped_grids * limit < ntiles) capped = w->capped_grids * limit;
        const int rounds = (capped + limit - 1) / limit, g = (capped + rounds - 1) / rounds;
        hipLaunchKernelGGL((k_mlp_split<CRITIC>), dim3(g), dim3(256), LDS_BYTES, stream, n, 0, capped, act.cursor, act.step_dev, obs,
                           reinterpret_cast<const unsigned char *>(w->split_ws),
                           reinterpret_cast<const unsigned char *>(w->split_ws_alt), action, out, act);
    }
    if (capped < ntiles) {
        const int rest = ntiles - capped, rounds = (rest + chip - 1) / chip, g = (rest + rounds - 1) / rounds;
        hipLaunchKernelGGL((k_mlp_split<CRITIC>), dim3(g), dim3(256), LDS_BYTES, stream, n, capped, ntiles, act.cursor, act.step_dev, obs,
                           reinterpret_cast<const unsigned char *>(w->split_ws),
                           reinterpret_cast<const unsigned char *>(w->split_ws_alt), action, out, act);
    }
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int split_forward(bool critic, int n, const float *obs, const float *action, const tt_mlp_weights *w, float *out,
                  const ActArgs &act, hipStream_t stream) {
    return critic ? launch_split<true>(n, obs, action, w, out, act, stream)
                  : launch_split<false>(n, obs, action, w, out, act, stream);
}

}  // namespace ttnet
