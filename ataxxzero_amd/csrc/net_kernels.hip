// Fused policy/value tower for gfx950 (MI355X): the whole residual conv net of
// model.py:38-79 in ONE kernel, activations resident in LDS from the first
// convolution to the heads, every contraction on MFMA.
//
// Math (reference model.py:116-142, TF semantics stated in SURVEY.md §8 a7):
//   h0 = relu(bn(conv3x3(features)))                       model.py:56-57
//   per block: t = relu(bn(conv3x3(h))); h = relu(bn(conv3x3(t)) + h)   :132-142
//   policy = conv1x1(h)  (17 raw logits per cell)          model.py:66-69
//   value  = tanh(reshape(conv1x1(h), 49) . fc_w + fc_b)   model.py:71-79
// conv = NHWC/HWIO cross-correlation with SAME zero padding; bn is the loaded
// inference form (x - mean) / sqrt(var + eps), gamma = 1, beta = 0.
//
// Mapping.  Every convolution is the implicit GEMM
//     out^T[oc][cell] = sum_k W^T[oc][k] * im2col^T[k][cell],   k = (tap, channel)
// with the WEIGHTS as the MFMA A operand (rows = 32 output channels per wave) and
// the ACTIVATIONS as the B operand (columns = 32 board cells).  The 32x32 f32
// accumulator then holds, per lane, 16 output channels of ONE cell in groups of
// four consecutive channels, so the epilogue (bn, skip, relu, convert) writes
// 8-byte runs straight back into the [cell][channel] LDS image that the next
// layer reads as its B operand: no transpose, no global round trip.
//
// A workgroup = 4 waves = BOARDS boards; wave w owns output channels [32w, 32w+32).
// LDS holds two activation images laid out [channel unit][cell slot]: a unit is 8
// channels = 16 B (16-bit types) or one f32 channel, and the slots of a unit are the
// workgroup's cells followed by a few always-zero slots.  The 32 cells of one B-fragment
// read are therefore 512 contiguous bytes (no bank conflicts, no swizzle) and moving to
// the next k-step is an immediate offset on the ds_read: the k-loop spends no VALU on
// addresses.  Taps that fall off the board read a zero slot with the same residue mod
// 8/16 as the cell they replace, so they stay off the other lanes' banks.
// Weights are pre-packed on the host in exact A-fragment order (1 KiB per
// wave-instruction, batch-norm scale folded in) and streamed from L2 through a
// four-deep register ring that stays warm across layers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <mutex>

#include "azh_host.h"
#include "engine_device.h"   // advance_worker: the search loop's queued moves ride in the tower launch's first workgroups

namespace azh {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

constexpr int F = 128;        // filters (model.py:16)
constexpr int OCT = F / 32;   // 32-channel output tiles = waves per workgroup
constexpr int NTHREADS = 64 * OCT;
#ifndef AZH_RING
#define AZH_RING 4
#endif
constexpr int RING = AZH_RING;   // A-fragment look-ahead in k-steps (a power of two dividing 8)
#ifndef AZH_BDIST
#define AZH_BDIST 2
#endif

template <int DT> struct Traits;

template <> struct Traits<AZH_DTYPE_BF16> {
    typedef __bf16 elem;
    typedef bf16x8 afrag;     // 8 consecutive k of one output channel
    typedef bf16x4 quad;
    typedef bf16x2 pair;
    static constexpr int KSTEP = 16;   // k per MFMA
    static constexpr int ESIZE = 2;
    __device__ static f32x16 mfma(afrag a, afrag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Traits<AZH_DTYPE_F16> {
    typedef _Float16 elem;
    typedef f16x8 afrag;
    typedef f16x4 quad;
    typedef f16x2 pair;
    static constexpr int KSTEP = 16;
    static constexpr int ESIZE = 2;
    __device__ static f32x16 mfma(afrag a, afrag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Traits<AZH_DTYPE_F32> {
    typedef float elem;
    typedef float afrag;      // one k of one output channel
    typedef f32x4 quad;
    typedef f32x2 pair;
    static constexpr int KSTEP = 2;
    static constexpr int ESIZE = 4;
    __device__ static f32x16 mfma(afrag a, afrag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
};

// BOARDS boards per workgroup.  16-bit types: 6 boards fill one CU's LDS (one workgroup per
// CU, one wave per SIMD); 3 boards let two workgroups share a CU (two waves per SIMD hide
// each other's LDS / L2 latency and smooth the tail of the grid).  f32: 3 boards.
template <int DT, int NB, int FT = F> struct Geo {
    typedef Traits<DT> Tr;
    static constexpr int FILTERS = FT;                 // channels of the tower (model.py:16; 64 / 128 / 256 are built)
    static constexpr int OCTF = FT / 32;               // waves per workgroup: one 32-channel output tile each
    static constexpr int NTHR = 64 * OCTF;
    static constexpr int BOARDS = NB;
    static constexpr int NC = 49 * BOARDS;             // real cells per workgroup
    static constexpr int NT = (NC + 31) / 32;          // 32-cell MFMA column tiles
    static constexpr int Z = 16;                       // zero slots per unit, shared by both images
    static constexpr int NSLOT = 2 * NC + Z;           // slots of one unit: [image 0][zeros][image 1]
    static constexpr int UB = DT == AZH_DTYPE_F32 ? 4 : 16;               // bytes of one unit in one slot
    static constexpr int NUNIT = FT * Tr::ESIZE / UB;                       // units per image
    static constexpr int CS = NSLOT * UB;              // byte stride between units
    static constexpr int VCELL_OFF = NUNIT * CS;       // f32 value-cell scratch behind the images
    static constexpr int LDS_BYTES = VCELL_OFF + NC * 4;
    static constexpr int KSTEPS_FULL = FT / Tr::KSTEP; // k-steps per tap, FT-channel input
    static constexpr int KSTEPS_IN = (Tr::KSTEP >= 4) ? 1 : 4 / Tr::KSTEP;  // 4 input planes
    // slot (within a unit) of cell c of image img
    __device__ static int real_slot(int img, int c) { return img * (NC + Z) + c; }
    // zero slot standing in for the off-board / pad cell index c of image img: the one with the
    // same residue mod 16 as the slot c would have had, so a B-fragment read (16 lanes x 16 B per
    // LDS pass) keeps every lane on its own banks
    __device__ static int zero_slot(int img, int c) { return NC + ((img * NC + c - NC) & (Z - 1)); }
    // byte offset of (slot, unit, byte)
    __device__ static int off(int slot, int unit, int byte = 0) { return unit * CS + slot * UB + byte; }
    // 16-bit types: channel ch lives in unit ch/8 at byte 2*(ch%8); f32: unit ch
    __device__ static int ch_off(int slot, int ch)
    {
        if constexpr (DT == AZH_DTYPE_F32) return off(slot, ch);
        else return off(slot, ch >> 3, (ch & 7) << 1);
    }
};

struct TowerArgs {
    const void *conv_w;      // packed A fragments, all tower layers back to back
    const void *head_w;      // packed A fragments of the fused policy+value 1x1 conv
    const void *conv_w2;     // variant 2 packing (16x16x32 fragments)
    const void *head_w2;
    const float *shift;      // [2B+1][128]  -mean/sqrt(var+eps); the scale 1/sqrt(var+eps) is folded into the packed weights
    const float *fc_w;       // [49]
    float fc_b;
    int blocks;
    const unsigned long long *boards;  // [..][2] (mover, opponent), indexed by game
    const int *list;         // dense list of games to evaluate (or nullptr = identity)
    const int *count;        // device count (or nullptr -> n)
    int n;
    unsigned long long blockers;
    float *logits;           // [..][833], indexed by game
    float *values;           // [..], indexed by game
    unsigned long long *stamps;  // diagnostic build only: [workgroup][wave][128] s_memtime stamps
    int sym;                 // 1: test-time symmetry averaging (nn_evals.py:48-62): virtual board v is list entry
                             // v >> 3 under dihedral symmetry v & 7; outputs are indexed by v, not by game
};

// The device-resident search loop plays its queued moves (sample, record, re-root: advance_game, engine_device.h) in the FIRST
// H.workers workgroups of the tower launch: they are dispatched before any tile's workgroup, so the re-roots start with the
// launch instead of waiting beside it for a slot the tower never leaves, and the loop needs no side stream and no cross-stream
// events.  A worker runs one wave (the others return), uses the workgroup's LDS as its scratch, and is gone after ~0.1 ms; the
// tiles are numbered from workgroup H.workers on (H.workers is a multiple of 8: a tile keeps its XCD).  true: this
// workgroup was a worker and is done.  H.workers = 0 (every launch outside that loop): nothing happens.
__device__ inline bool tower_prologue(const AdvanceHook &H, unsigned char *smem, int &wg)
{
    wg = (int)blockIdx.x;
    if (H.workers == 0)
        return false;
    // (a launch that fits the chip at once keeps its tiles in front: H.at_head = 0, the workers are its last workgroups)
    const int tiles = (int)gridDim.x - H.workers;
    const int worker = H.at_head ? wg : wg - tiles;
    if (worker >= 0 && worker < H.workers) {
        if (threadIdx.x < WAVE)
            advance_worker(H.P, smem, worker, H.workers);
        return true;
    }
    if (H.at_head)
        wg -= H.workers;
    return false;
}

// apply_symmetry (nn_evals.py:8-16): cell (x, y) of the transformed tensor shows cell (ox, oy) of the original
__device__ __host__ inline void sym_cell(int s, int x, int y, int &ox, int &oy)
{
    const int a = (s & 4) ? y : x, b = (s & 4) ? x : y;
    ox = (s & 1) ? 6 - a : a;
    oy = (s & 2) ? 6 - b : b;
}

// source game and source square of cell (x, y) of (virtual) board v
__device__ inline int tower_src(const TowerArgs &A, int v, int x, int y, int &sq)
{
    int ox = x, oy = y;
    if (A.sym)
        sym_cell(v & 7, x, y, ox, oy);
    sq = ox + 7 * (6 - oy);
    const int gi = A.sym ? (v >> 3) : v;
    return A.list ? A.list[gi] : gi;
}

// row of the output arrays for (virtual) board v
__device__ inline int tower_dst(const TowerArgs &A, int v)
{
    if (A.sym)
        return v;
    return A.list ? A.list[v] : v;
}

template <int V> struct IC { static constexpr int value = V; };

template <int I, int N, typename Fn> __device__ inline void static_for(Fn &&fn)
{
    if constexpr (I < N) {
        fn(IC<I>());
        static_for<I + 1, N>(fn);
    }
}

// One convolution layer for the workgroup's boards, KS k-steps per tap (compile time).
// `in_img` / `out_img` pick the LDS images (0 / 1); with `skip` the layer adds the residual
// input, i.e. the current content of the output image.
__device__ inline unsigned long long stamp_now()
{
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <int DT, int NB, int KS, bool STAMP = false, int FT = F>
__device__ inline void conv_layer(unsigned char *lds, int in_img, int out_img, bool skip,
                                  const typename Traits<DT>::afrag *__restrict__ wp,
                                  typename Traits<DT>::afrag (&a)[RING], f32x16 &sh, const float *__restrict__ shift_next,
                                  const int (&vmask)[Geo<DT, NB>::NT], int wave, int lane,
                                  unsigned long long *st = nullptr)
{
    typedef Traits<DT> Tr;
    typedef Geo<DT, NB, FT> G;
    typedef typename Tr::afrag afrag;
    constexpr int NT = G::NT;
    constexpr int TOTAL = 9 * KS;
    const int r = lane & 31, h = lane >> 5;
    // Accumulators start at the batch-norm shift of their channel (rows 8q + 4h + i of the
    // wave's 32-channel tile) plus, for the second conv of a block, the residual input read
    // from the output image (still intact: this layer only writes it in its epilogue).  The
    // epilogue then has no affine step and no read-modify-write left.  `sh` was loaded
    // during the previous layer; the next layer's shift is fetched behind the k-loop.
    f32x16 acc[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ct++)
        acc[ct] = sh;
    if (skip) {
        // branch-free: pad lanes read a zero slot, so all reads of the wave issue back to back
        typename Tr::quad sk[NT][4];
#pragma unroll
        for (int ct = 0; ct < NT; ct++) {
            const int cell = ct * 32 + r;
            const int slot = cell < G::NC ? G::real_slot(out_img, cell) : G::zero_slot(out_img, cell);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int ch = 32 * wave + 8 * q + 4 * h;
                if constexpr (DT == AZH_DTYPE_F32) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        sk[ct][q][i] = *reinterpret_cast<const float *>(lds + G::ch_off(slot, ch + i));
                } else {
                    sk[ct][q] = *reinterpret_cast<const typename Tr::quad *>(lds + G::ch_off(slot, ch));
                }
            }
        }
#pragma unroll
        for (int ct = 0; ct < NT; ct++)
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc[ct][4 * q + i] += (float)sk[ct][q][i];
    }
    auto fetch_shift = [&]() {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 t4 = *reinterpret_cast<const f32x4 *>(shift_next + 32 * wave + 8 * q + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; i++)
                sh[4 * q + i] = t4[i];
        }
    };

    // A fragments: wave-uniform base + one 32-bit lane offset; the fragment of step s sits
    // s * (waves per workgroup) KiB further on.
    const char *wbase = reinterpret_cast<const char *>(wp);
    const unsigned lane_off = (unsigned)((wave * 64 + lane) * sizeof(afrag));
    constexpr size_t STEP_BYTES = (size_t)G::OCTF * 64 * sizeof(afrag);
    // Steps >= TOTAL run on into the next layer's fragments (layers are contiguous in the
    // packed buffer, which is padded at the end): the ring is already warm when the next
    // layer starts.
    auto load_a = [&](int step) {
        const char *p = wbase + (size_t)step * STEP_BYTES;
        return *reinterpret_cast<const afrag *>(p + lane_off);
    };

    // B fragments: rows[ct] = LDS byte address of the tap's source slot for this lane's k
    // half; k-step ks is an immediate 2*CS*ks further on.  Off-board taps (and pad lanes)
    // read a zero slot of the same residue.
    auto rows_for = [&](int tap, int (&dst)[NT]) {
        const int drow = (tap / 3 - 1) * 7 + (tap % 3 - 1);
#pragma unroll
        for (int ct = 0; ct < NT; ct++) {
            const int c = ct * 32 + r + drow;
            const int slot = ((vmask[ct] >> tap) & 1) ? G::real_slot(in_img, c) : G::zero_slot(in_img, c);
            dst[ct] = h * G::CS + slot * G::UB;
        }
    };
    auto load_b = [&](afrag (&bf)[NT], const int (&rows)[NT], int ks) {
#pragma unroll
        for (int ct = 0; ct < NT; ct++)
            bf[ct] = *reinterpret_cast<const afrag *>(lds + rows[ct] + ks * (2 * G::CS));
    };

    // Software pipeline.  Step s = (tap, ks): while its MFMAs run, the B fragments of step
    // s+BD are on their way from LDS and the A fragment of step s+RING from L2.  The body is
    // straight-line code (compile-time phases, no branches) so that the waits the compiler
    // inserts are exact counted waits and never drain the prefetch.  BD = 2 for the 16-bit
    // towers (a lone wave's step is only 160 cycles, less than a loaded LDS round trip).
    constexpr int BD = (DT == AZH_DTYPE_F32) ? 1 : AZH_BDIST;
    constexpr int NBUF = BD + 1;
    afrag b[NBUF][NT];
    int cur[NT], nxt[NT];
    auto one_step = [&](auto ph_tag, auto buf_tag, const int (&src)[NT], int ks_target, int s) {
        constexpr int ph = decltype(ph_tag)::value, bu = decltype(buf_tag)::value;
        load_b(b[(bu + BD) % NBUF], src, ks_target);
        __builtin_amdgcn_sched_barrier(0);  // the prefetch stays ahead of this step's MFMAs
#pragma unroll
        for (int ct = 0; ct < NT; ct++)
            acc[ct] = Tr::mfma(a[ph], b[bu][ct], acc[ct]);
        a[ph] = load_a(s + RING);
    };
    if constexpr (STAMP) st[0] = stamp_now();

    if constexpr (TOTAL <= 18) {
        // few steps (the 4-plane input layer): everything unrolled, rows recomputed per step
        static_for<0, BD>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            rows_for(i / KS, cur);
            load_b(b[i % NBUF], cur, i % KS);
        });
        static_for<0, TOTAL>([&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value, t = s + BD;
            constexpr int tap_t = t / KS < 9 ? t / KS : 8;
            rows_for(tap_t, cur);
            one_step(IC<(s % RING)>(), IC<(s % NBUF)>(), cur, t % KS, s);
        });
    } else if constexpr (BD == 1) {
        static_assert(KS % RING == 0, "k-steps per tap must be a multiple of the A ring");
        constexpr int CH = KS <= 8 ? KS : RING;   // unrolled steps per chunk
        rows_for(0, cur);
        load_b(b[0], cur, 0);
        for (int tap = 0; tap < 9; tap++) {
            rows_for(tap < 8 ? tap + 1 : 8, nxt);
            for (int k0 = 0; k0 < KS; k0 += CH) {
                const bool last_chunk = k0 + CH == KS;
                int tail[NT];  // source rows of the chunk's last prefetch
#pragma unroll
                for (int ct = 0; ct < NT; ct++)
                    tail[ct] = last_chunk ? nxt[ct] : cur[ct];
                const int tail_ks = last_chunk ? 0 : k0 + CH;
                const int s0 = tap * KS + k0;
                static_for<0, CH>([&](auto j_tag) {
                    constexpr int j = decltype(j_tag)::value;
                    if constexpr (j + 1 < CH)
                        one_step(IC<(j % RING)>(), IC<(j & 1)>(), cur, k0 + j + 1, s0 + j);
                    else
                        one_step(IC<(j % RING)>(), IC<(j & 1)>(), tail, tail_ks, s0 + j);
                });
            }
#pragma unroll
            for (int ct = 0; ct < NT; ct++)
                cur[ct] = nxt[ct];
        }
    } else {
        // KS steps per tap, B fragments two steps ahead in three buffers: the unrolled body
        // covers three taps so that every buffer and ring slot keeps a compile-time name.
        static_assert(BD == 2 && KS >= BD && (3 * KS) % RING == 0, "pipeline shape");
        rows_for(0, cur);
        load_b(b[0], cur, 0);
        load_b(b[1], cur, 1);
        for (int t0 = 0; t0 < 9; t0 += 3) {
            static_for<0, 3>([&](auto tt_tag) {
                constexpr int tt = decltype(tt_tag)::value;
                const int tap = t0 + tt;
                rows_for(tap < 8 ? tap + 1 : 8, nxt);
                static_for<0, KS>([&](auto j_tag) {
                    constexpr int j = decltype(j_tag)::value, sl = tt * KS + j;
                    if constexpr (j + BD < KS)
                        one_step(IC<(sl % RING)>(), IC<(sl % NBUF)>(), cur, j + BD, tap * KS + j);
                    else
                        one_step(IC<(sl % RING)>(), IC<(sl % NBUF)>(), nxt, j + BD - KS, tap * KS + j);
                });
#pragma unroll
                for (int ct = 0; ct < NT; ct++)
                    cur[ct] = nxt[ct];
            });
        }
    }

    if constexpr (STAMP) st[1] = stamp_now();
    fetch_shift();  // next layer's shift: in flight behind the epilogue and the barrier
    // the ring now holds steps TOTAL .. TOTAL+RING-1 at slots (TOTAL + i) % RING: rotate so
    // that slot i is the next layer's step i
    if constexpr (TOTAL % RING != 0) {
        afrag t[RING];
#pragma unroll
        for (int i = 0; i < RING; i++)
            t[i] = a[(TOTAL + i) % RING];
#pragma unroll
        for (int i = 0; i < RING; i++)
            a[i] = t[i];
    }

    // epilogue: relu, convert, write [cell][channel]
#pragma unroll
    for (int ct = 0; ct < NT; ct++) {
        const int cell = ct * 32 + r;
        if (cell < G::NC) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int ch = 32 * wave + 8 * q + 4 * h;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    v[i] = acc[ct][4 * q + i];
                if constexpr (DT == AZH_DTYPE_F32) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        *reinterpret_cast<float *>(lds + G::ch_off(G::real_slot(out_img, cell), ch + i)) = v[i] > 0.0f ? v[i] : 0.0f;
                } else {
                    typedef typename Tr::pair pair;
                    const int off = G::ch_off(G::real_slot(out_img, cell), ch);
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        v[i] = v[i] > 0.0f ? v[i] : 0.0f;
                    f32x2 lo, hi;
                    lo[0] = v[0]; lo[1] = v[1]; hi[0] = v[2]; hi[1] = v[3];
                    const pair plo = __builtin_convertvector(lo, pair), phi = __builtin_convertvector(hi, pair);
                    uint2 packed;
                    packed.x = __builtin_bit_cast(unsigned, plo);
                    packed.y = __builtin_bit_cast(unsigned, phi);
                    *reinterpret_cast<uint2 *>(lds + off) = packed;
                }
            }
        }
    }
    if constexpr (STAMP) st[2] = stamp_now();
}

template <int DT, int NB, int WPS, bool STAMP = false, int FT = F>
__global__ __launch_bounds__(64 * (FT / 32), WPS) void k_tower(TowerArgs A, AdvanceHook H)
{
    typedef Traits<DT> Tr;
    typedef Geo<DT, NB, FT> G;
    constexpr int OCT = G::OCTF, NTHREADS = G::NTHR, F = FT;  // shadow the 128-filter constants of the fast path
    typedef typename Tr::afrag afrag;
    extern __shared__ __align__(16) unsigned char smem[];
    int wg;
    if (tower_prologue(H, smem, wg))
        return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = (A.count ? *A.count : A.n) * (A.sym ? 8 : 1);
    const int tile0 = wg * G::BOARDS;
    if (tile0 >= n)
        return;
    const int nb = (n - tile0) < G::BOARDS ? (n - tile0) : G::BOARDS;

    float *vcell = reinterpret_cast<float *>(smem + G::VCELL_OFF);
    // diagnostic stamps (STAMP builds only): 4 per layer {loop start, loop end, epilogue end, barrier passed}
    unsigned long long *st = nullptr;
    if constexpr (STAMP) {
        st = A.stamps + ((size_t)wg * OCT + wave) * 128;
        st[0] = stamp_now();
    }

    // zero both images (zero slots, pad channels of the input planes, unused boards)
    for (int i = tid * 16; i < G::VCELL_OFF; i += NTHREADS * 16)
        *reinterpret_cast<uint4 *>(smem + i) = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // input planes (cpp/self_play_client.cpp:174-202): ones, mover, opponent, blockers
    for (int cell = tid; cell < nb * 49; cell += NTHREADS) {
        const int bl = cell / 49, c = cell % 49;
        const int x = c / 7, y = c % 7;
        int sq;
        const int game = tower_src(A, tile0 + bl, x, y, sq);
        const unsigned long long mover = A.boards[2 * (size_t)game + 0];
        const unsigned long long opp = A.boards[2 * (size_t)game + 1];
        const float f1 = (float)((mover >> sq) & 1ULL), f2 = (float)((opp >> sq) & 1ULL);
        const float f3 = (float)((A.blockers >> sq) & 1ULL);
        if constexpr (DT == AZH_DTYPE_F32) {
            *reinterpret_cast<float *>(smem + G::ch_off(G::real_slot(0, cell), 0)) = 1.0f;
            *reinterpret_cast<float *>(smem + G::ch_off(G::real_slot(0, cell), 1)) = f1;
            *reinterpret_cast<float *>(smem + G::ch_off(G::real_slot(0, cell), 2)) = f2;
            *reinterpret_cast<float *>(smem + G::ch_off(G::real_slot(0, cell), 3)) = f3;
        } else {
            typename Tr::quad o;
            o[0] = (typename Tr::elem)1.0f;
            o[1] = (typename Tr::elem)f1;
            o[2] = (typename Tr::elem)f2;
            o[3] = (typename Tr::elem)f3;
            *reinterpret_cast<typename Tr::quad *>(smem + G::ch_off(G::real_slot(0, cell), 0)) = o;
        }
    }

    // per column tile: which of the 9 taps stay on the board for this lane's cell
    int vmask[G::NT];
    {
        const int r = lane & 31;
#pragma unroll
        for (int ct = 0; ct < G::NT; ct++) {
            const int cell = ct * 32 + r;
            int m = 0;
            if (cell < G::NC) {
                const int c = cell % 49, x = c / 7, y = c % 7;
                for (int tap = 0; tap < 9; tap++) {
                    const int xx = x + tap / 3 - 1, yy = y + tap % 3 - 1;
                    if (xx >= 0 && xx < 7 && yy >= 0 && yy < 7)
                        m |= 1 << tap;
                }
            }
            vmask[ct] = m;
        }
    }
    __syncthreads();

    // tower
    const afrag *wp = reinterpret_cast<const afrag *>(A.conv_w);
    const size_t l0 = (size_t)9 * G::KSTEPS_IN * OCT * 64;
    const size_t lf = (size_t)9 * G::KSTEPS_FULL * OCT * 64;
    afrag aring[RING];  // A-fragment ring, kept warm across layers
#pragma unroll
    for (int i = 0; i < RING; i++)
        aring[i] = wp[(size_t)i * OCT * 64 + wave * 64 + lane];
    f32x16 sh;       // batch-norm shift of the coming layer, fetched one layer ahead
    {
        const int h = lane >> 5;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 t4 = *reinterpret_cast<const f32x4 *>(A.shift + 32 * wave + 8 * q + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; i++)
                sh[4 * q + i] = t4[i];
        }
    }
    if constexpr (STAMP) st[1] = stamp_now();
    // (the shift table has one spare row so the last layer's look-ahead stays in bounds)
    conv_layer<DT, NB, G::KSTEPS_IN, STAMP, FT>(smem, 0, 1, false, wp, aring, sh, A.shift + F, vmask, wave, lane, st + 4);
    __syncthreads();
    if constexpr (STAMP) st[7] = stamp_now();
    wp += l0;
    for (int b = 0; b < A.blocks; b++) {
        const float *t1 = A.shift + (size_t)(1 + 2 * b) * F;
        conv_layer<DT, NB, G::KSTEPS_FULL, STAMP, FT>(smem, 1, 0, false, wp, aring, sh, t1 + F, vmask, wave, lane,
                                                  st + 8 + 8 * b);
        __syncthreads();
        if constexpr (STAMP) st[8 + 8 * b + 3] = stamp_now();
        wp += lf;
        conv_layer<DT, NB, G::KSTEPS_FULL, STAMP, FT>(smem, 0, 1, true, wp, aring, sh, t1 + 2 * F, vmask, wave, lane,
                                                  st + 12 + 8 * b);
        __syncthreads();
        if constexpr (STAMP) st[12 + 8 * b + 3] = stamp_now();
        wp += lf;
    }

    // heads: one 32-row A tile = 17 policy channels + the value conv channel (row 17);
    // column tiles are dealt round-robin to the waves; logits go from the accumulators to HBM.
    {
        const afrag *hp = reinterpret_cast<const afrag *>(A.head_w) + lane;
        const int r = lane & 31, h = lane >> 5;
        for (int ct = wave; ct < G::NT; ct += OCT) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; i++)
                acc[i] = 0.0f;
            const int cell = ct * 32 + r;
            const int slot = cell < G::NC ? G::real_slot(1, cell) : G::zero_slot(1, cell);
            for (int ks = 0; ks < G::KSTEPS_FULL; ks++) {
                const afrag a = hp[(size_t)ks * 64];
                const afrag bfrag = *reinterpret_cast<const afrag *>(smem + G::off(slot, 2 * ks + h));
                acc = Tr::mfma(a, bfrag, acc);
            }
            if (cell < nb * 49) {
                const int bl = cell / 49, c = cell % 49;
                float *dst = A.logits + (size_t)tower_dst(A, tile0 + bl) * 833 + 17 * c;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int oc = (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (oc < 17)
                        dst[oc] = acc[i];
                    else if (oc == 17)
                        vcell[cell] = acc[i];
                }
            }
        }
    }
    __syncthreads();
    if (tid < nb) {
        const int game = tower_dst(A, tile0 + tid);
        float s = 0.0f;
        for (int c = 0; c < 49; c++)
            s = __builtin_fmaf(vcell[tid * 49 + c], A.fc_w[c], s);
        A.values[game] = tanhf(s + A.fc_b);
    }
    if constexpr (STAMP) st[2] = stamp_now();
}

// ------------------------------------------------------------------ variant 2 (16-bit types)
// v_mfma_f32_16x16x32: each wave owns 64 output channels (four 16-channel A tiles) x five 16-cell
// B tiles, so one B fragment feeds four MFMAs: half the LDS read bytes per FLOP of variant 1,
// the same accumulator count (4 x 5 x 4 = 80 registers) and the same HBM/L2 traffic.  Waves
// 0/1 take channels 0-63, waves 2/3 channels 64-127; even waves take cell tiles 0-4, odd 5-9.
// A fragment: lane l holds W[oc = 16T + (l & 15)][k = 8 (l >> 4) + j]; B fragment: lane l holds
// act[cell = 16t + (l & 15)][k = 8 (l >> 4) + j]; D: lane l holds oc = 4 (l >> 4) + reg of cell l & 15.
#ifndef AZH_RING2
#define AZH_RING2 2  // A-fragment ring depth of variant 2 (steps of prefetch distance)
#endif
constexpr int RING2 = AZH_RING2;
// The one behavioural build switch left in the tower: off-board taps as out-of-range LDS reads (1, default; the device
// is probed for it in azh_net_create) or steered to zero slots (0; tests/test_gpu_net.py builds and compares both).
// The other A/B switches of round 2 (packed relu, buffer loads for the weights, first MFMA opens the accumulator,
// late row addresses: profiles/round2_tower_variants.txt) are settled and their losing sides removed.
#ifndef AZH_OOBZERO
#define AZH_OOBZERO 1   /* +1.5 % at 16 K boards, +1.3 % at 3.6 K (profiles/round2_tower_variants.txt, call 9) */
#endif

// The first layer (4 input planes) as a 1x1 convolution over an im2col image of the planes — K = 9 taps x 4 planes = 36,
// two k-steps of 32 — instead of nine k-steps (one per tap) that multiply 28 zero channels each: 7 of the tower's 873 k-steps
// less (model.py:56-57 is the same sum in another order).  0: the first layer walks its nine taps like every other layer.
#ifndef AZH_FIRST_IM2COL
#define AZH_FIRST_IM2COL 1
#endif

template <int DT> struct Mfma16;
template <> struct Mfma16<AZH_DTYPE_BF16> {
    __device__ static f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<AZH_DTYPE_F16> {
    __device__ static f32x4 mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

struct Geo2 {
    static constexpr int BOARDS = 3;
    static constexpr int NC = 147;
    static constexpr int NT = 10;                 // 16-cell tiles (160 >= 147)
    static constexpr int TPW = 5;                 // cell tiles per wave
    static constexpr int Z = 16;
    static constexpr int NSLOT = 320;             // 2*147 + 16 = 310 slots, padded so the unit stride is a
                                                  // multiple of the 256-B bank row (the four k-groups of a
                                                  // B-fragment read sit in four consecutive units)
    static constexpr int UB = 16;
    static constexpr int NUNIT = 16;
    static constexpr int CS = NSLOT * UB;         // 5120
    static constexpr int LDS_BYTES = NUNIT * CS;  // 81,920: two workgroups per CU
    static constexpr int KS_FULL = F / 32, KS_IN = 1;
    __device__ static int real_slot(int img, int c) { return img * (NC + Z) + c; }
    __device__ static int zero_slot(int img, int c) { return NC + ((img * NC + c - NC) & (Z - 1)); }
    __device__ static int ch_off(int slot, int ch) { return (ch >> 3) * CS + slot * UB + ((ch & 7) << 1); }
    // value-cell scratch lives in the 10 pad slots of units 0..3
    __device__ static int vcell_off(int c) { return (c / 40) * CS + 310 * UB + (c % 40) * 4; }
    static constexpr int RING = RING2;            // A-fragment ring depth (k-steps of prefetch distance)
    static constexpr bool SKIP = true;            // the tile table has edge tiles whose off-board taps are dropped (skip_pair)
    __device__ static int tile_cell(int tile, int r);  // cell of lane r of cell tile `tile` (TILE_CELL below)
    static constexpr int NA = 4;                  // 16-channel A tiles per wave: waves 0/1 channels 0-63, waves 2/3 64-127 ...
    __device__ static int first_channel(int wave) { return 64 * (wave >> 1); }
    __device__ static int first_a_tile(int wave) { return 4 * (wave >> 1); }
    __device__ static int first_tile(int wave) { return TPW * (wave & 1); }  // ... even waves cell tiles 0-4, odd 5-9
};

// Cells of a workgroup are numbered c = 21 y + 7 board + x (the rows of the three boards interleaved), and LDS slot =
// cell, so a neighbour is one uniform offset away (dx + 21 dy).  WHICH 16 cells form an MFMA cell tile is free, as long
// as the 16 lanes of a tile keep distinct slots mod 16 (the bank rule of ds_read_b128's 16-lane groups): lane r of a
// tile always holds a cell with c % 16 == r.  Four tiles are filled with cells of one board edge each — y = 0, y = 6
// (wave half 0) and x = 0, x = 6 (wave half 1), 16 cells apiece — so the three taps that look past that edge are
// all-zero for the whole tile and are dropped at compile time (B loads and MFMAs).  The other six tiles take the k-th
// remaining cell of every residue class; the sixth of them holds only the three cells that are left (144-146, all on
// y = 6) and sits in wave half 0, whose unrolled inner index is dy, so its dy = +1 taps are dropped as well: 15 of the
// 90 (tile, tap) pairs per layer, 9 in wave half 0 and 6 in wave half 1.  Generated by tools/tower_tile_table.py;
// entries >= 0x100 are empty lanes (pseudo cell = lane, never valid).
// The FLOP count reported for the kernel stays the padded-tap figure.
__device__ const unsigned short TILE_CELL[10][16] = {
    {0x010, 0x001, 0x002, 0x003, 0x004, 0x005, 0x006, 0x007, 0x008, 0x009, 0x00a, 0x00b, 0x00c, 0x00d, 0x00e, 0x00f},
    {0x080, 0x081, 0x082, 0x083, 0x084, 0x085, 0x086, 0x087, 0x088, 0x089, 0x08a, 0x08b, 0x08c, 0x08d, 0x08e, 0x07f},
    {0x000, 0x011, 0x012, 0x013, 0x024, 0x025, 0x016, 0x017, 0x018, 0x019, 0x01a, 0x02b, 0x02c, 0x01d, 0x01e, 0x01f},
    {0x020, 0x021, 0x032, 0x033, 0x034, 0x035, 0x026, 0x027, 0x028, 0x039, 0x03a, 0x03b, 0x03c, 0x02d, 0x02e, 0x02f},
    {0x090, 0x091, 0x092, 0x103, 0x104, 0x105, 0x106, 0x107, 0x108, 0x109, 0x10a, 0x10b, 0x10c, 0x10d, 0x10e, 0x10f},
    {0x070, 0x031, 0x062, 0x023, 0x054, 0x015, 0x046, 0x077, 0x038, 0x069, 0x02a, 0x05b, 0x01c, 0x04d, 0x07e, 0x03f},
    {0x030, 0x061, 0x022, 0x053, 0x014, 0x045, 0x076, 0x037, 0x068, 0x029, 0x05a, 0x01b, 0x04c, 0x07d, 0x03e, 0x06f},
    {0x040, 0x041, 0x042, 0x043, 0x044, 0x055, 0x036, 0x047, 0x048, 0x049, 0x04a, 0x04b, 0x05c, 0x03d, 0x04e, 0x04f},
    {0x050, 0x051, 0x052, 0x063, 0x064, 0x065, 0x056, 0x057, 0x058, 0x059, 0x06a, 0x06b, 0x06c, 0x05d, 0x05e, 0x05f},
    {0x060, 0x071, 0x072, 0x073, 0x074, 0x075, 0x066, 0x067, 0x078, 0x079, 0x07a, 0x07b, 0x07c, 0x06d, 0x06e, 0x08f},
};

__device__ inline int Geo2::tile_cell(int tile, int r) { return TILE_CELL[tile][r]; }

// THIN batches — a match's last games, a UAI engine's single position: a launch of a handful of boards lasts as long as ONE
// workgroup needs for the 25 layers, and a 3-board workgroup spends that time on 720 MFMAs per wave and layer whether its
// boards are real or not.  Geo2Thin gives every board a workgroup of its own: the same LDS image (board 0's cells at slots
// 21 y + x, the other boards' slots unused), four cell tiles instead of ten (49 cells: 16 + 16 + 13 + 4, lane r of a tile
// again a cell with slot % 16 == r), two per wave, no edge tiles — 288 MFMAs per wave and layer — and a deeper A ring,
// because with 8 MFMAs per k-step instead of 20 the weight stream from L2 is what a step waits for.  A board's result does
// not depend on anything but the board (and differs from the 3-board kernel's in the last bits: another summation order).
#ifndef AZH_RING_THIN
#define AZH_RING_THIN 6
#endif
__device__ const unsigned short TILE_CELL1[4][16] = {
    {0x000, 0x001, 0x002, 0x003, 0x004, 0x005, 0x006, 0x017, 0x018, 0x019, 0x01a, 0x01b, 0x02c, 0x02d, 0x02e, 0x02f},
    {0x030, 0x041, 0x042, 0x043, 0x044, 0x015, 0x016, 0x057, 0x058, 0x059, 0x02a, 0x02b, 0x06c, 0x06d, 0x06e, 0x03f},
    {0x040, 0x081, 0x082, 0x083, 0x054, 0x045, 0x056, 0x107, 0x108, 0x069, 0x05a, 0x06b, 0x10c, 0x10d, 0x07e, 0x06f},
    {0x080, 0x101, 0x102, 0x103, 0x084, 0x055, 0x106, 0x107, 0x108, 0x109, 0x06a, 0x10b, 0x10c, 0x10d, 0x10e, 0x07f},
};
struct Geo2Thin : Geo2 {
    static constexpr int BOARDS = 1;
    // every wave takes ALL four cell tiles and 32 of the channels (two A tiles): the same 8 MFMAs per k-step as 4 A tiles x 2
    // cell tiles, but every A fragment is loaded by one wave only — 8 KB of weights per step and workgroup instead of 16,
    // which at 64 B per clock from L2 was what a step waited for (0.118 ms per launch with 4 x 2, profiles/round5_thin_tower.txt)
    static constexpr int NA = 2;
    static constexpr int TPW = 4;
    __device__ static int first_channel(int wave) { return 32 * wave; }
    __device__ static int first_a_tile(int wave) { return 2 * wave; }
    __device__ static int first_tile(int) { return 0; }
    static constexpr int RING = AZH_RING_THIN;
    static constexpr bool SKIP = false;
    __device__ static int tile_cell(int tile, int r) { return TILE_CELL1[tile][r]; }
};
static_assert((3 * Geo2::KS_FULL) % Geo2Thin::RING == 0, "ring phase must be compile-time inside a row");

__device__ inline void cell_xy(int c, int &bl, int &x, int &y)
{
    y = c / 21;
    const int r = c - 21 * y;
    bl = r / 7;
    x = r - 7 * bl;
}

// The taps of a layer are walked as three rows of three: wave half 0 walks dx in the (run-time) row loop and dy inside
// the unrolled row, wave half 1 the other way round, so that for both halves "tile 0 skips inner index 0, tile 1 skips
// inner index 2" is a compile-time fact; in wave half 0 the partial tile (4) skips inner index 2 like the y = 6 tile.
// (the partial tile's three pairs: +1.1 % at 3.6 K and 16 K boards, profiles/round3_partial_tile_skip.txt)
__device__ constexpr bool skip_pair(int chf, int ct, int inner)
{
    return (ct == 0 && inner == 0) || (ct == 1 && inner == 2) || (chf == 0 && ct == 4 && inner == 2);
}

// Epilogue of a variant-2 layer — relu, convert, write 4 channels (8 bytes) per (A tile, cell tile) into image out_img —
// as TEXT shared by conv_layer2 and first_layer2 (as a function taking the 80 accumulator registers by reference it
// made the kernel spill 240 of them).  relu after the conversion, on the packed pair: a negative bf16 / f16 is a negative
// int16, so one packed integer max with 0 clears it (conversion and relu commute: both are monotone and keep the sign).
// Uses lds, out_img, acc, vmask, cellv, cbase, kg, NA, Tr, G, TPW of the enclosing function.
#if AZH_OOBZERO
#define AZH_LAYER_EPILOGUE2 \
    _Pragma("unroll") \
    for (int ct = 0; ct < TPW; ct++) { \
        if (!(vmask[ct] & 0x200)) { \
            const int cellq = cellv[ct] - kg * G::CS + out_img * ((G::NC + G::Z) * G::UB); \
    _Pragma("unroll") \
            for (int t = 0; t < NA; t++) { \
                float v[4]; \
    _Pragma("unroll") \
                for (int i = 0; i < 4; i++) \
                    v[i] = acc[t][ct][i]; \
                f32x2 lo, hi; \
                lo[0] = v[0]; lo[1] = v[1]; hi[0] = v[2]; hi[1] = v[3]; \
                typedef typename Tr::pair pair; \
                const pair plo = __builtin_convertvector(lo, pair), phi = __builtin_convertvector(hi, pair); \
                uint2 packed; \
                typedef short short2v __attribute__((ext_vector_type(2))); \
                const short2v zero2 = {0, 0}; \
                packed.x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(short2v, plo), zero2)); \
                packed.y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(short2v, phi), zero2)); \
                *reinterpret_cast<uint2 *>(lds + cellq + G::ch_off(0, cbase + 16 * t + 4 * kg)) = packed; \
            } \
        } \
    }
#else
#define AZH_LAYER_EPILOGUE2 \
    _Pragma("unroll") \
    for (int ct = 0; ct < TPW; ct++) { \
        if (!(cellv[ct] & 0x100)) { \
            const int cell = cellv[ct]; \
    _Pragma("unroll") \
            for (int t = 0; t < NA; t++) { \
                float v[4]; \
    _Pragma("unroll") \
                for (int i = 0; i < 4; i++) \
                    v[i] = acc[t][ct][i]; \
                f32x2 lo, hi; \
                lo[0] = v[0]; lo[1] = v[1]; hi[0] = v[2]; hi[1] = v[3]; \
                typedef typename Tr::pair pair; \
                const pair plo = __builtin_convertvector(lo, pair), phi = __builtin_convertvector(hi, pair); \
                uint2 packed; \
                typedef short short2v __attribute__((ext_vector_type(2))); \
                const short2v zero2 = {0, 0}; \
                packed.x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(short2v, plo), zero2)); \
                packed.y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(short2v, phi), zero2)); \
                *reinterpret_cast<uint2 *>(lds + G::ch_off(G::real_slot(out_img, cell), cbase + 16 * t + 4 * kg)) = packed; \
            } \
        } \
    }
#endif

template <int DT, int KS, int CHF, bool STAMP, typename G = Geo2>
__device__ inline void conv_layer2(unsigned char *lds, int in_img, int out_img, bool skip,
                                   __amdgpu_buffer_rsrc_t wrsrc, const void *wbase0,
                                   const typename Traits<DT>::afrag *__restrict__ wp,
                                   typename Traits<DT>::afrag (&a)[G::RING][G::NA], f32x16 &sh, const float *__restrict__ shift_next,
                                   const int (&vmask)[G::TPW], const int (&cellv)[G::TPW], int wave, int lane,
                                   unsigned long long *st)
{
    typedef Traits<DT> Tr;
    constexpr int RING2 = G::RING;                 // (shadows the global: this geometry's ring depth)
    auto skip_pair = [](int chf, int ct, int inner) constexpr { return G::SKIP && azh::skip_pair(chf, ct, inner); };
    typedef typename Tr::afrag afrag;
    constexpr int TPW = G::TPW;
    constexpr int TOTAL = 9 * KS;
    constexpr int ROW = 3 * KS;  // steps per row of three taps
    static_assert(ROW % RING2 == 0 || KS == 1, "ring phase must be compile-time inside a row");
    static_assert(RING2 <= ROW, "the ring prefetch reaches at most one row ahead");
    const int kg = lane >> 4;
    constexpr int NA = G::NA;                      // 16-channel A tiles per wave
    const int cbase = G::first_channel(wave);      // the wave's first output channel
    const int atile0 = G::first_a_tile(wave);      // ... = its first 16-channel A tile
    // accumulators [A tile][cell tile], started at the batch-norm shift (+ residual input)
    f32x4 acc[NA][TPW];
    f32x4 shq[NA];  // the shift of this wave's four channel quads
#pragma unroll
    for (int t = 0; t < NA; t++)
#pragma unroll
        for (int i = 0; i < 4; i++)
            shq[t][i] = sh[4 * t + i];
    // Without a residual input the accumulators are not initialised at all: the first MFMA of every (A tile, cell tile)
    // takes the shift as its C operand and writes the accumulator (80 register copies per layer saved); with one they
    // start at shift + residual input.
    const bool first_c = !skip;
    if (!first_c) {
#pragma unroll
        for (int t = 0; t < NA; t++)
#pragma unroll
            for (int ct = 0; ct < TPW; ct++)
                acc[t][ct] = shq[t];
    }
    if (skip) {
        typename Tr::quad sk[NA][TPW];
#pragma unroll
        for (int ct = 0; ct < TPW; ct++) {
#if AZH_OOBZERO
            // cellv = kg * CS + cell * UB, the empty-lane flag is bit 9 of vmask: an empty lane reads past the end of LDS
            const int base = (cellv[ct] - kg * G::CS + out_img * ((G::NC + G::Z) * G::UB)) +
                             (int)(__builtin_amdgcn_ubfe((unsigned)vmask[ct], 9, 1) << 28);
#pragma unroll
            for (int t = 0; t < NA; t++)
                sk[t][ct] = *reinterpret_cast<const typename Tr::quad *>(lds + base + G::ch_off(0, cbase + 16 * t + 4 * kg));
#else
            const int cell = cellv[ct] & 0xFF;
            const int slot = (cellv[ct] & 0x100) ? G::zero_slot(out_img, cell) : G::real_slot(out_img, cell);
#pragma unroll
            for (int t = 0; t < NA; t++)
                sk[t][ct] = *reinterpret_cast<const typename Tr::quad *>(lds + G::ch_off(slot, cbase + 16 * t + 4 * kg));
#endif
        }
#pragma unroll
        for (int t = 0; t < NA; t++)
#pragma unroll
            for (int ct = 0; ct < TPW; ct++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc[t][ct][i] += (float)sk[t][ct][i];
    }
    auto fetch_shift = [&]() {
#pragma unroll
        for (int t = 0; t < NA; t++) {
            const f32x4 t4 = *reinterpret_cast<const f32x4 *>(shift_next + cbase + 16 * t + 4 * kg);
#pragma unroll
            for (int i = 0; i < 4; i++)
                sh[4 * t + i] = t4[i];
        }
    };

    // A fragments of packed step s = tap * KS + ks: [s][oc tile 8][lane][8 elements]; this wave reads tiles atile0 .. atile0 + NA - 1
    const unsigned lane_off = (unsigned)(lane * sizeof(afrag));
    // buffer loads: resource (base of the packed weights) and the step's offset in scalar registers, the lane's 16-B
    // slot in one VGPR, the tile in the immediate — no 64-bit vector address arithmetic in the k-loop
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const unsigned wbyte = (unsigned)(reinterpret_cast<const char *>(wp) - reinterpret_cast<const char *>(wbase0)) +
                           (unsigned)__builtin_amdgcn_readfirstlane(atile0 * 1024);
    auto load_a = [&](afrag (&dst)[NA], int step) {
        const unsigned soff = wbyte + (unsigned)step * 8192u;
#pragma unroll
        for (int t = 0; t < NA; t++)
            dst[t] = __builtin_bit_cast(afrag, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_off + t * 1024, soff, 0));
    };
    // (row o, inner i) -> tap = 3 dxi + dyi
    auto tap_of = [&](int o, int i) { return CHF == 0 ? 3 * o + i : 3 * i + o; };
    // packed step of walk position (row o, step j of the row); positions past the layer continue into the next
    // layer's stream, whose first KS steps are tap 0 in either walk
    auto step_of = [&](int o, int j) {
        if (j >= ROW) {
            j -= ROW;
            o += 1;
        }
        return o < 3 ? tap_of(o, j / KS) * KS + j % KS : TOTAL + j;
    };
    // byte offsets of this lane's B fragments for the tap at (row o, inner i); skipped pairs are left alone.
    auto rows_for = [&](int o, auto i_tag, int (&dst)[TPW]) {
        constexpr int i = decltype(i_tag)::value;
        const int tap = tap_of(o, i);
        const int drow = CHF == 0 ? (o - 1) + 21 * (i - 1) : (i - 1) + 21 * (o - 1);
        static_for<0, TPW>([&](auto ct_tag) {
            constexpr int ct = decltype(ct_tag)::value;
            if constexpr (!skip_pair(CHF, ct, i)) {
#if AZH_OOBZERO
                // An LDS read past the workgroup's allocation returns zeros (tools/microbench/lds_oob.hip): a tap that
                // looks off the board sets bit 28 of its byte address instead of being steered to a zero slot — three
                // instructions per (tile, tap): neighbour offset, the tap's bit of the lane's off-board mask, merge.
                const unsigned c16 = (unsigned)(cellv[ct] + (drow + in_img * (G::NC + G::Z)) * G::UB);
                // (added, not or-ed: for an off-board tap c16 may be a small negative number, and 2^28 + c16 stays far out
                // of range with the k-step's immediate offset on top, where 0xFFFFFxxx would wrap back into the image)
                dst[ct] = (int)((__builtin_amdgcn_ubfe((unsigned)vmask[ct], (unsigned)tap, 1) << 28) + c16);
#else
                const int c = (cellv[ct] & 0xFF) + drow;
                const int slot = ((vmask[ct] >> tap) & 1) ? G::real_slot(in_img, c) : G::zero_slot(in_img, c);
                dst[ct] = kg * G::CS + slot * G::UB;
#endif
            }
        });
    };
    auto load_b = [&](afrag (&bf)[TPW], const int (&rows)[TPW], int ks, auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        static_for<0, TPW>([&](auto ct_tag) {
            constexpr int ct = decltype(ct_tag)::value;
            if constexpr (!skip_pair(CHF, ct, i))
                bf[ct] = *reinterpret_cast<const afrag *>(lds + rows[ct] + ks * (4 * G::CS));
        });
    };
    afrag b[2][TPW];
    int cur[TPW], nxt[TPW];
    // one row of three taps = 3 KS steps, straight-line: step j consumes inner tap j / KS, k-step j % KS
    auto tap_row = [&](int o, auto par0_tag, auto ring0_tag, auto first_tag) {
        constexpr int par0 = decltype(par0_tag)::value;    // B buffer parity of the row's first step
        constexpr int ring0 = decltype(ring0_tag)::value;  // A ring slot of the row's first step
        constexpr bool FIRST = decltype(first_tag)::value != 0;  // the layer's first row, accumulators not initialised
        static_for<0, ROW>([&](auto j_tag) {
            constexpr int j = decltype(j_tag)::value;
            constexpr int i = j / KS, ks = j % KS;
            constexpr int par = (par0 + j) & 1;
            constexpr int rs = (ring0 + j) % RING2;
            constexpr int j1 = j + 1;
            constexpr int i1 = (j1 / KS) % 3, ks1 = j1 % KS;
            constexpr bool rows_late = KS > 1;  // (with one k-step per tap the offsets are needed at once)
            if constexpr (ks == 0 && !rows_late) {  // offsets of the tap after this one (it may belong to the next row)
                const int o1 = i == 2 ? (o < 2 ? o + 1 : 2) : o;
                rows_for(o1, IC<(i + 1) % 3>(), nxt);
            }
            // prefetch the B fragments of the next step
            if constexpr (ks1 == 0)
                load_b(b[par ^ 1], nxt, 0, IC<i1>());
            else
                load_b(b[par ^ 1], cur, ks1, IC<i1>());
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ks == 0 && rows_late) {
                // the next tap's offsets are first used KS - 1 steps from here: their address arithmetic goes behind the
                // barrier, where it can issue between this step's MFMAs instead of ahead of them
                const int o1 = i == 2 ? (o < 2 ? o + 1 : 2) : o;
                rows_for(o1, IC<(i + 1) % 3>(), nxt);
            }
#pragma unroll
            for (int t = 0; t < NA; t++)
                static_for<0, TPW>([&](auto ct_tag) {
                    constexpr int ct = decltype(ct_tag)::value;
                    if constexpr (!skip_pair(CHF, ct, i)) {
                        // the first step this cell tile takes part in (tile 0 sits out inner tap 0)
                        constexpr bool opens = FIRST && j == (skip_pair(CHF, ct, 0) ? KS : 0);
                        acc[t][ct] = Mfma16<DT>::mfma(a[rs][t], b[par][ct], opens ? shq[t] : acc[t][ct]);
                    }
                });
            load_a(a[rs], step_of(o, j + RING2));
            if constexpr (ks1 == 0) {
#pragma unroll
                for (int ct = 0; ct < TPW; ct++)
                    cur[ct] = nxt[ct];
            }
        });
    };
    rows_for(0, IC<0>(), cur);
#pragma unroll
    for (int ct = 0; ct < TPW; ct++)
        nxt[ct] = cur[ct];
    load_b(b[0], cur, 0, IC<0>());
    if constexpr (STAMP) st[0] = stamp_now();
    if constexpr (ROW % 2 != 0 || ROW % RING2 != 0) {
        static_for<0, 3>([&](auto d_tag) {
            constexpr int d = decltype(d_tag)::value;
            if (d == 0 && first_c)
                tap_row(d, IC<(ROW * d) & 1>(), IC<(ROW * d) % RING2>(), IC<1>());
            else
                tap_row(d, IC<(ROW * d) & 1>(), IC<(ROW * d) % RING2>(), IC<0>());
        });
    } else {
        if (first_c)
            tap_row(0, IC<0>(), IC<0>(), IC<1>());
#pragma nounroll
        for (int o = first_c ? 1 : 0; o < 3; o++)
            tap_row(o, IC<0>(), IC<0>(), IC<0>());
    }
    if constexpr (STAMP) st[1] = stamp_now();
    fetch_shift();
    if constexpr (TOTAL % RING2 != 0) {  // rotate so that slot i holds walk position TOTAL + i (next layer's step i)
        constexpr int sh_ = TOTAL % RING2;
        afrag tmp[RING2][NA];
#pragma unroll
        for (int i = 0; i < RING2; i++)
#pragma unroll
            for (int t = 0; t < NA; t++)
                tmp[i][t] = a[(i + sh_) % RING2][t];
#pragma unroll
        for (int i = 0; i < RING2; i++)
#pragma unroll
            for (int t = 0; t < NA; t++)
                a[i][t] = tmp[i][t];
    }
    AZH_LAYER_EPILOGUE2
    if constexpr (STAMP) st[2] = stamp_now();
}

#if AZH_FIRST_IM2COL
// The first layer as a 1x1 convolution over the im2col image of the input planes (built by tower2_run in units 0-7 of image
// 0: k = 4 tap + plane, zeros for taps that look off the board and for k >= 36): two k-steps, no tap walk, no off-board
// logic.  Packed A stream of the layer: [step 2][oc tile 8][lane 64][8] (net_pack); the ring holds its two steps on
// entry and the next layer's first two on exit, like every layer.
template <int DT, bool STAMP, typename G = Geo2>
__device__ inline void first_layer2(unsigned char *lds, __amdgpu_buffer_rsrc_t wrsrc, const void *wbase0,
                                    const typename Traits<DT>::afrag *__restrict__ wp,
                                    typename Traits<DT>::afrag (&a)[G::RING][G::NA], f32x16 &sh, const float *__restrict__ shift_next,
                                    const int (&vmask)[G::TPW], const int (&cellv)[G::TPW], int wave, int lane,
                                    unsigned long long *st)
{
    typedef Traits<DT> Tr;
    typedef typename Tr::afrag afrag;
    constexpr int TPW = G::TPW;
    constexpr int RING2 = G::RING;   // slots 0, 1: this layer's two steps; slots 2 ..: the next layer's first RING - 2
    static_assert(RING2 >= 2, "the two steps of the im2col layer come out of the ring");
    const int kg = lane >> 4;
    constexpr int NA = G::NA;                      // 16-channel A tiles per wave
    const int cbase = G::first_channel(wave);      // the wave's first output channel
    const int atile0 = G::first_a_tile(wave);      // ... = its first 16-channel A tile
    constexpr int out_img = 1;
    f32x4 acc[NA][TPW];
    f32x4 shq[NA];
#pragma unroll
    for (int t = 0; t < NA; t++)
#pragma unroll
        for (int i = 0; i < 4; i++)
            shq[t][i] = sh[4 * t + i];
    // B fragments of both steps: unit 4 s + kg of the lane's cell (an empty lane reads zeros)
    afrag b[2][TPW];
#pragma unroll
    for (int ct = 0; ct < TPW; ct++) {
#if AZH_OOBZERO
        const int base = cellv[ct] + (int)(__builtin_amdgcn_ubfe((unsigned)vmask[ct], 9, 1) << 28);
#else
        const int cell = cellv[ct] & 0xFF;
        const int base = kg * G::CS + ((cellv[ct] & 0x100) ? G::zero_slot(0, cell) : G::real_slot(0, cell)) * G::UB;
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
            b[ks][ct] = *reinterpret_cast<const afrag *>(lds + base + ks * (4 * G::CS));
    }
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const unsigned lane_off = (unsigned)(lane * sizeof(afrag));
    const unsigned wbyte = (unsigned)(reinterpret_cast<const char *>(wp) - reinterpret_cast<const char *>(wbase0)) +
                           (unsigned)__builtin_amdgcn_readfirstlane(atile0 * 1024);
    if constexpr (STAMP) st[0] = stamp_now();
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
        for (int t = 0; t < NA; t++)
#pragma unroll
            for (int ct = 0; ct < TPW; ct++)
                acc[t][ct] = Mfma16<DT>::mfma(a[ks][t], b[ks][ct], ks == 0 ? shq[t] : acc[t][ct]);
        // the next layer's step RING - 2 + ks (its stream follows this layer's two steps)
#pragma unroll
        for (int t = 0; t < NA; t++)
            a[ks][t] = __builtin_bit_cast(afrag, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_off + t * 1024,
                                                                                              wbyte + (unsigned)(RING2 + ks) * 8192u, 0));
    }
    if constexpr (RING2 > 2) {  // rotate so that slot i holds the next layer's step i
        afrag tmp[RING2][NA];
#pragma unroll
        for (int i = 0; i < RING2; i++)
#pragma unroll
            for (int t = 0; t < NA; t++)
                tmp[i][t] = a[(i + 2) % RING2][t];
#pragma unroll
        for (int i = 0; i < RING2; i++)
#pragma unroll
            for (int t = 0; t < NA; t++)
                a[i][t] = tmp[i][t];
    }
    if constexpr (STAMP) st[1] = stamp_now();
#pragma unroll
    for (int t = 0; t < NA; t++) {
        const f32x4 t4 = *reinterpret_cast<const f32x4 *>(shift_next + cbase + 16 * t + 4 * kg);
#pragma unroll
        for (int i = 0; i < 4; i++)
            sh[4 * t + i] = t4[i];
    }
    AZH_LAYER_EPILOGUE2
    if constexpr (STAMP) st[2] = stamp_now();
}
#endif

// `tbase`: the wave's first cell tile in the geometry's tile table (G::TPW * CHF where the two wave halves are separate
// instantiations; a run-time value where one instantiation serves both).
template <int DT, int CHF, bool STAMP, typename G = Geo2>
__device__ inline void tower2_body(const TowerArgs &A, unsigned char *smem, int tile0, int nb, int wave, int lane,
                                   unsigned long long *st, int tbase)
{
    typedef Traits<DT> Tr;
    typedef typename Tr::afrag afrag;
    constexpr int RING2 = G::RING;
    int vmask[G::TPW], cellv[G::TPW];
    {
        const int r = lane & 15;
#pragma unroll
        for (int ct = 0; ct < G::TPW; ct++) {
            const int cv = G::tile_cell(tbase + ct, r);
            cellv[ct] = cv;
            int m = 0;
            if (!(cv & 0x100)) {
                int bl, x, y;
                cell_xy(cv, bl, x, y);
                for (int tap = 0; tap < 9; tap++) {
                    const int xx = x + tap / 3 - 1, yy = y + tap % 3 - 1;
                    if (xx >= 0 && xx < 7 && yy >= 0 && yy < 7)
                        m |= 1 << tap;
                }
            }
            vmask[ct] = m;
#if AZH_OOBZERO
            // bit set = the tap looks off the board; bit 9 = empty lane (all taps off); cellv = byte offset of the lane's
            // cell in a unit plus the lane's k-group row
            vmask[ct] = (cv & 0x100) ? 0x3FF : (~m & 0x1FF);
            cellv[ct] = (lane >> 4) * G::CS + (cv & 0xFF) * G::UB;
#endif
        }
    }
    const afrag *wp = reinterpret_cast<const afrag *>(A.conv_w2);
    // buffer resource over the packed weights (raw 32-bit data, no bounds smaller than the stream; gfx94x/gfx950 word 3)
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(A.conv_w2), 0, 0x7FFFFFFF, 0x00020000);
#if AZH_FIRST_IM2COL
    const size_t l0 = (size_t)2 * 8 * 64, lf = (size_t)9 * G::KS_FULL * 8 * 64;
#else
    const size_t l0 = (size_t)9 * G::KS_IN * 8 * 64, lf = (size_t)9 * G::KS_FULL * 8 * 64;
#endif
    constexpr int NA = G::NA;
    const int cbase = G::first_channel(wave), atile0 = G::first_a_tile(wave), kg = lane >> 4;
    afrag aring[RING2][G::NA];
#pragma unroll
    for (int i = 0; i < RING2; i++) {
#if AZH_FIRST_IM2COL
        const int step = i;  // the im2col layer's two steps
#else
        // walk position i of the first layer (KS_IN = 1: position = inner tap of row 0)
        const int step = CHF == 0 ? i : 3 * i;
#endif
#pragma unroll
        for (int t = 0; t < NA; t++)
            aring[i][t] = wp[((size_t)step * 8 + atile0 + t) * 64 + lane];
    }
    f32x16 sh;
#pragma unroll
    for (int t = 0; t < NA; t++) {
        const f32x4 t4 = *reinterpret_cast<const f32x4 *>(A.shift + cbase + 16 * t + 4 * kg);
#pragma unroll
        for (int i = 0; i < 4; i++)
            sh[4 * t + i] = t4[i];
    }
    if constexpr (STAMP) st[1] = stamp_now();
#if AZH_FIRST_IM2COL
    first_layer2<DT, STAMP, G>(smem, wrsrc, A.conv_w2, wp, aring, sh, A.shift + F, vmask, cellv, wave, lane, st + 4);
#else
    conv_layer2<DT, G::KS_IN, CHF, STAMP, G>(smem, 0, 1, false, wrsrc, A.conv_w2, wp, aring, sh, A.shift + F, vmask, cellv, wave, lane, st + 4);
#endif
    __syncthreads();
    if constexpr (STAMP) st[7] = stamp_now();
    wp += l0;
    for (int b = 0; b < A.blocks; b++) {
        const float *t1 = A.shift + (size_t)(1 + 2 * b) * F;
        conv_layer2<DT, G::KS_FULL, CHF, STAMP, G>(smem, 1, 0, false, wrsrc, A.conv_w2, wp, aring, sh, t1 + F, vmask, cellv, wave, lane,
                                                st + 8 + 8 * b);
        __syncthreads();
        if constexpr (STAMP) st[8 + 8 * b + 3] = stamp_now();
        wp += lf;
        conv_layer2<DT, G::KS_FULL, CHF, STAMP, G>(smem, 0, 1, true, wrsrc, A.conv_w2, wp, aring, sh, t1 + 2 * F, vmask, cellv, wave, lane,
                                                st + 12 + 8 * b);
        __syncthreads();
        if constexpr (STAMP) {
            st[12 + 8 * b + 3] = stamp_now();
            if (b < 16)
                st[105 + b] = __builtin_amdgcn_s_memrealtime();  // 100 MHz, one base for the chip: the shader clock block by block
        }
        wp += lf;
    }
}

// The whole net for the workgroup's boards tile0 .. tile0 + 2 of the n (virtual) boards A lists.
template <int DT, bool STAMP = false, typename G = Geo2>
__device__ __forceinline__ void tower2_run(const TowerArgs &A, unsigned char *smem, int tile0, int n)
{
    typedef Traits<DT> Tr;
    typedef typename Tr::afrag afrag;
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index is the same in all 64 lanes: say so, and everything derived from it (the weight stream's base
    // address above all) lives in scalar registers instead of being recomputed per lane
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = (n - tile0) < G::BOARDS ? (n - tile0) : G::BOARDS;
    unsigned long long *st = nullptr;
    if constexpr (STAMP) {
        st = A.stamps + ((size_t)blockIdx.x * OCT + wave) * 128;
        st[0] = stamp_now();
        st[3] = __builtin_amdgcn_s_memrealtime();
    }
    for (int i = tid * 16; i < G::LDS_BYTES; i += NTHREADS * 16)
        *reinterpret_cast<uint4 *>(smem + i) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // input planes (cpp/self_play_client.cpp:174-202): ones, mover, opponent, blockers
#if AZH_FIRST_IM2COL
    // ... as the im2col image the first layer multiplies: for cell c and tap (dx, dy) the four planes of cell (x + dx, y + dy),
    // zeros off the board; k = 4 tap + plane, i.e. unit tap / 2, bytes 8 (tap & 1) ..; one 16-byte store per (cell, tap pair)
    for (int item = tid; item < G::NC * 5; item += NTHREADS) {
        const int u = item / G::NC, cell = item - u * G::NC;
        int bl, x, y;
        cell_xy(cell, bl, x, y);
        if (bl < nb) {
            int sq0;
            const int game = tower_src(A, tile0 + bl, x, y, sq0);
            const unsigned long long mover = A.boards[2 * (size_t)game + 0];
            const unsigned long long opp = A.boards[2 * (size_t)game + 1];
            typename Tr::afrag o;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int tap = 2 * u + h;
                const int xx = x + tap / 3 - 1, yy = y + tap % 3 - 1;
                const bool on = tap < 9 && xx >= 0 && xx < 7 && yy >= 0 && yy < 7;
                int sq = 0;
                if (on)
                    (void)tower_src(A, tile0 + bl, xx, yy, sq);
                o[4 * h + 0] = (typename Tr::elem)(on ? 1.0f : 0.0f);
                o[4 * h + 1] = (typename Tr::elem)(float)(on ? (mover >> sq) & 1ULL : 0ULL);
                o[4 * h + 2] = (typename Tr::elem)(float)(on ? (opp >> sq) & 1ULL : 0ULL);
                o[4 * h + 3] = (typename Tr::elem)(float)(on ? (A.blockers >> sq) & 1ULL : 0ULL);
            }
            *reinterpret_cast<typename Tr::afrag *>(smem + u * G::CS + G::real_slot(0, cell) * G::UB) = o;
        }
    }
#else
    for (int cell = tid; cell < G::NC; cell += NTHREADS) {
        int bl, x, y;
        cell_xy(cell, bl, x, y);
        if (bl < nb) {
            int sq;
            const int game = tower_src(A, tile0 + bl, x, y, sq);
            const unsigned long long mover = A.boards[2 * (size_t)game + 0];
            const unsigned long long opp = A.boards[2 * (size_t)game + 1];
            typename Tr::quad o;
            o[0] = (typename Tr::elem)1.0f;
            o[1] = (typename Tr::elem)(float)((mover >> sq) & 1ULL);
            o[2] = (typename Tr::elem)(float)((opp >> sq) & 1ULL);
            o[3] = (typename Tr::elem)(float)((A.blockers >> sq) & 1ULL);
            *reinterpret_cast<typename Tr::quad *>(smem + G::ch_off(G::real_slot(0, cell), 0)) = o;
        }
    }
#endif
    __syncthreads();
    // the two cell halves run separate instantiations: which (tile, tap) pairs are skipped is compile-time
    if constexpr (G::SKIP) {
        if (wave & 1)
            tower2_body<DT, 1, STAMP, G>(A, smem, tile0, nb, wave, lane, st, G::TPW);
        else
            tower2_body<DT, 0, STAMP, G>(A, smem, tile0, nb, wave, lane, st, 0);
    } else {
        // no compile-time fact depends on the wave half: one instantiation, the half picks the tiles
        tower2_body<DT, 0, STAMP, G>(A, smem, tile0, nb, wave, lane, st, G::first_tile(wave));
    }
    // heads: two 16-row A tiles (policy 0-15 | policy 16, value conv, pad); cell tiles dealt to the waves
    {
        const afrag *hp = reinterpret_cast<const afrag *>(A.head_w2) + lane;
        const int r = lane & 15, kg = lane >> 4;
        for (int ct = wave; ct < G::NT; ct += OCT) {
            f32x4 acc2[2];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc2[t][i] = 0.0f;
            const int cell = 16 * ct + r;
            const int slot = cell < G::NC ? G::real_slot(1, cell) : G::zero_slot(1, cell);
#pragma unroll
            for (int ks = 0; ks < G::KS_FULL; ks++) {
                const afrag bfrag = *reinterpret_cast<const afrag *>(smem + (4 * ks + kg) * G::CS + slot * G::UB);
#pragma unroll
                for (int t = 0; t < 2; t++)
                    acc2[t] = Mfma16<DT>::mfma(hp[(size_t)(ks * 2 + t) * 64], bfrag, acc2[t]);
            }
            int bl = 0, x = 0, y = 0;
            if (cell < G::NC)
                cell_xy(cell, bl, x, y);
            if (cell < G::NC && bl < nb) {
                float *dst = A.logits + (size_t)tower_dst(A, tile0 + bl) * 833 + 17 * (7 * x + y);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    dst[4 * kg + i] = acc2[0][i];
                if (kg == 0) {
                    dst[16] = acc2[1][0];
                    *reinterpret_cast<float *>(smem + G::vcell_off(cell)) = acc2[1][1];
                }
            }
        }
    }
    __syncthreads();
    if (tid < nb) {
        const int game = tower_dst(A, tile0 + tid);
        float s = 0.0f;
        for (int c = 0; c < 49; c++)  // c = 7 x + y, model.py:75's reshape order
            s = __builtin_fmaf(*reinterpret_cast<const float *>(smem + G::vcell_off(21 * (c % 7) + 7 * tid + c / 7)), A.fc_w[c], s);
        A.values[game] = tanhf(s + A.fc_b);
    }
    if constexpr (STAMP) {
        st[2] = stamp_now();
        st[104] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int DT, bool STAMP = false>
__global__ __launch_bounds__(NTHREADS, 2) void k_tower2(TowerArgs A, AdvanceHook H)
{
    extern __shared__ __align__(16) unsigned char smem[];
    int wg;
    if (tower_prologue(H, smem, wg))
        return;
    const int n = (A.count ? *A.count : A.n) * (A.sym ? 8 : 1);
    const int tile0 = wg * Geo2::BOARDS;
    if (tile0 >= n)
        return;
    tower2_run<DT, STAMP>(A, smem, tile0, n);
}

template <int DT>
__global__ __launch_bounds__(NTHREADS, 2) void k_tower2_thin(TowerArgs A, AdvanceHook H)
{
    extern __shared__ __align__(16) unsigned char smem[];
    int wg;
    if (tower_prologue(H, smem, wg))
        return;
    const int n = (A.count ? *A.count : A.n) * (A.sym ? 8 : 1);
    const int tile0 = wg;
    if (tile0 >= n)
        return;
    tower2_run<DT, false, Geo2Thin>(A, smem, tile0, n);
}

// Two nets in ONE launch (arena, uai_ringmaster.py:221-265: every position is searched by the net whose move it is): the
// first ceil(nA / 3) workgroups evaluate list A with net A's weights, the following ceil(nB / 3) list B with net B's.
// Everything that differs between the two — weights, shifts, value head, list, count, block count — is picked once per
// workgroup from the kernel arguments with scalar selects (blockIdx is uniform); the body is the one k_tower2 runs, so a
// board's result is bit for bit that of the single-net launch.  Two third-full launches back to back become one launch
// whose workgroups all start together: at 2 x 500 leaves the iteration's evaluator time halves.
template <int DT, typename G = Geo2>
__global__ __launch_bounds__(NTHREADS, 2) void k_tower2_pair(TowerArgs A, TowerArgs B, AdvanceHook H)
{
    extern __shared__ __align__(16) unsigned char smem[];
    int wg;
    if (tower_prologue(H, smem, wg))
        return;
    const int na = *A.count;
    const int wga = (na + G::BOARDS - 1) / G::BOARDS;
    const bool second = wg >= wga;
    TowerArgs X = A;
    if (second) {
        X.conv_w2 = B.conv_w2;
        X.head_w2 = B.head_w2;
        X.shift = B.shift;
        X.fc_w = B.fc_w;
        X.fc_b = B.fc_b;
        X.blocks = B.blocks;
        X.list = B.list;
    }
    const int n = second ? *B.count : na;
    const int tile0 = (wg - (second ? wga : 0)) * G::BOARDS;
    if (tile0 >= n)
        return;
    tower2_run<DT, false, G>(X, smem, tile0, n);
}

// ------------------------------------------------------------------ host side

static inline uint16_t f32_to_bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u)
        return (uint16_t)((u >> 16) | 0x40);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static inline uint16_t f32_to_f16(float f)
{
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

template <typename T> static inline T cvt_elem(float f, int dt);
template <> inline uint16_t cvt_elem<uint16_t>(float f, int dt) { return dt == AZH_DTYPE_BF16 ? f32_to_bf16(f) : f32_to_f16(f); }
template <> inline float cvt_elem<float>(float f, int) { return f; }

// Pack W[tap][cin][128] (HWIO, tap = 3*i + j) into A-fragment order.
// `scale` (per output column, may be null) is multiplied in before the conversion.
template <typename T>
static void pack_conv(const float *w, const float *scale, int cin, int taps, int ksteps, int kstep, int octiles,
                      int ocols, int dt, std::vector<T> &out)
{
    const int per_lane = kstep / 2;  // k elements per lane per MFMA
    for (int tap = 0; tap < taps; tap++)
        for (int ks = 0; ks < ksteps; ks++)
            for (int ot = 0; ot < octiles; ot++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < per_lane; e++) {
                        const int c = kstep * ks + per_lane * (lane >> 5) + e;
                        const int oc = 32 * ot + (lane & 31);
                        float v = 0.0f;
                        if (c < cin && oc < ocols)
                            v = w[((size_t)tap * cin + c) * ocols + oc] * (scale ? scale[oc] : 1.0f);
                        out.push_back(cvt_elem<T>(v, dt));
                    }
}

// Variant 2 packing: [tap][ks (32 channels)][oc tile 16 x ntiles][lane][8]: lane l holds
// W[c = 32 ks + 8 (l >> 4) + e][oc = 16 T + (l & 15)].
static void pack_conv16(const float *w, const float *scale, int cin, int taps, int ksteps, int ntiles, int ocols, int dt,
                        std::vector<uint16_t> &out)
{
    for (int tap = 0; tap < taps; tap++)
        for (int ks = 0; ks < ksteps; ks++)
            for (int T = 0; T < ntiles; T++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 8; e++) {
                        const int c = 32 * ks + 8 * (lane >> 4) + e, oc = 16 * T + (lane & 15);
                        float v = 0.0f;
                        if (c < cin && oc < ocols)
                            v = w[((size_t)tap * cin + c) * ocols + oc] * (scale ? scale[oc] : 1.0f);
                        out.push_back(cvt_elem<uint16_t>(v, dt));
                    }
}

struct NetDtypeBuffers {
    void *conv_w = nullptr;
    void *head_w = nullptr;
    void *conv_w2 = nullptr;
    void *head_w2 = nullptr;
};

}  // namespace azh

using namespace azh;

struct azh_net {
    int blocks = 0, filters = 0;
    std::vector<float> conv_flat;  // host copy, reference order
    std::vector<float> scale;      // [2B+1][128] 1/sqrt(var+eps), folded into the packed weights
    float *d_shift = nullptr, *d_fcw = nullptr;
    float fc_b = 0.0f;
    NetDtypeBuffers bufs[3];
};

static int net_pack(azh_net *net, int dt)
{
    if (net->bufs[dt].conv_w)
        return 0;
    const int B = net->blocks;
    const int F = net->filters, OCT = F / 32;  // run-time width (the 128-filter constants are shadowed on purpose)
    const float *p = net->conv_flat.data();
    const int kstep = dt == AZH_DTYPE_F32 ? 2 : 16;
    const int ks_in = dt == AZH_DTYPE_F32 ? 2 : 1;
    const int ks_full = F / kstep;
    const size_t head_off = (size_t)9 * 4 * F + (size_t)2 * B * 9 * F * F;
    // fused head matrix [128][32]: columns 0..16 policy, 17 value conv
    std::vector<float> head((size_t)F * 32, 0.0f);
    for (int c = 0; c < F; c++) {
        for (int t = 0; t < 17; t++)
            head[(size_t)c * 32 + t] = p[head_off + (size_t)c * 17 + t];
        head[(size_t)c * 32 + 17] = p[head_off + (size_t)F * 17 + c];
    }
    auto upload = [&](const void *src, size_t bytes, void **dst) -> int {
        AZH_HIP(hipMalloc(dst, bytes));
        AZH_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return 0;
    };
    if (dt == AZH_DTYPE_F32) {
        std::vector<float> cw, hw;
        const float *sc = net->scale.data();
        pack_conv<float>(p, sc, 4, 9, ks_in, kstep, OCT, F, dt, cw);
        for (int l = 0; l < 2 * B; l++)
            pack_conv<float>(p + (size_t)9 * 4 * F + (size_t)l * 9 * F * F, sc + (size_t)(l + 1) * F, F, 9, ks_full, kstep,
                             OCT, F, dt, cw);
        cw.resize(cw.size() + (size_t)RING * OCT * 64, 0.0f);  // the A ring reads RING steps past the last layer
        pack_conv<float>(head.data(), nullptr, F, 1, ks_full, kstep, 1, 32, dt, hw);
        if (upload(cw.data(), cw.size() * 4, &net->bufs[dt].conv_w)) return -1;
        if (upload(hw.data(), hw.size() * 4, &net->bufs[dt].head_w)) return -1;
    } else {
        std::vector<uint16_t> cw, hw;
        const float *sc = net->scale.data();
        pack_conv<uint16_t>(p, sc, 4, 9, ks_in, kstep, OCT, F, dt, cw);
        for (int l = 0; l < 2 * B; l++)
            pack_conv<uint16_t>(p + (size_t)9 * 4 * F + (size_t)l * 9 * F * F, sc + (size_t)(l + 1) * F, F, 9, ks_full,
                                kstep, OCT, F, dt, cw);
        cw.resize(cw.size() + (size_t)RING * OCT * 64 * 8, 0);  // the A ring reads RING steps past the last layer
        pack_conv<uint16_t>(head.data(), nullptr, F, 1, ks_full, kstep, 1, 32, dt, hw);
        if (upload(cw.data(), cw.size() * 2, &net->bufs[dt].conv_w)) return -1;
        if (upload(hw.data(), hw.size() * 2, &net->bufs[dt].head_w)) return -1;
        if (F != 128)
            return 0;  // variant 2 (the tuned 16x16x32 tower) is built for 128 filters
        // variant 2
        std::vector<uint16_t> cw2, hw2;
#if AZH_FIRST_IM2COL
        // the first layer as a 1x1 convolution over k = 4 tap + plane (36 of 64): [step 2][oc tile 8][lane 64][8]
        for (int ks = 0; ks < 2; ks++)
            for (int T = 0; T < 8; T++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 8; e++) {
                        const int k = 32 * ks + 8 * (lane >> 4) + e, tap = k / 4, c = k % 4, oc = 16 * T + (lane & 15);
                        const float v = tap < 9 ? p[((size_t)tap * 4 + c) * F + oc] * sc[oc] : 0.0f;
                        cw2.push_back(cvt_elem<uint16_t>(v, dt));
                    }
#else
        pack_conv16(p, sc, 4, 9, 1, 8, F, dt, cw2);
#endif
        for (int l = 0; l < 2 * B; l++)
            pack_conv16(p + (size_t)9 * 4 * F + (size_t)l * 9 * F * F, sc + (size_t)(l + 1) * F, F, 9, 4, 8, F, dt, cw2);
        // the A ring reads its depth in steps past the last layer (the thin geometry's ring is the deeper one)
        cw2.resize(cw2.size() + (size_t)(RING2 > AZH_RING_THIN ? RING2 : AZH_RING_THIN) * 8 * 64 * 8, 0);
        pack_conv16(head.data(), nullptr, F, 1, 4, 2, 32, dt, hw2);
        if (upload(cw2.data(), cw2.size() * 2, &net->bufs[dt].conv_w2)) return -1;
        if (upload(hw2.data(), hw2.size() * 2, &net->bufs[dt].head_w2)) return -1;
    }
    return 0;
}

constexpr int MAX_DEVICES = 64;
static int current_device()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES)
        return -1;
    return dev;
}

#if AZH_OOBZERO
// Variant 2 relies on the LDS range check: a ds_read beyond the workgroup's allocation must return zeros.  Checked once
// per device and process before the first net is handed out (reads at 80 KiB, 160 KiB and 2^28 past a 4 KiB allocation
// filled with ones, with and without an instruction offset): a device that answered anything else would make every
// board edge wrong, so there the 16-bit towers run variant 1 instead (lds_range_check).
__global__ void k_lds_range_probe(unsigned *out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x)
        reinterpret_cast<unsigned *>(smem)[i] = 0x01010101u;
    __syncthreads();
    const unsigned addrs[3] = {81920u + 16u * threadIdx.x, 163840u + 16u * threadIdx.x, (1u << 28) + 16u * threadIdx.x - 352u};
    unsigned any = 0;
    for (int k = 0; k < 3; k++) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v, w;
        asm volatile("ds_read_b128 %0, %1 offset:61440\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addrs[k]) : "memory");
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addrs[k]) : "memory");
        any |= v.x | v.y | v.z | v.w | w.x | w.y | w.z | w.w;
    }
    const unsigned inside = reinterpret_cast<const unsigned *>(smem)[threadIdx.x];
    if (any)
        atomicOr(out, 1u);
    if (inside != 0x01010101u)
        atomicOr(out, 2u);
}

// 1: out-of-range LDS reads return zeros on the current device (variant 2 may run), 0: they do not (the 16-bit towers
// fall back to variant 1, whose off-board taps read zero slots inside the allocation), < 0: the probe itself failed.
static int lds_range_check()
{
    static std::mutex mu;
    static int state[MAX_DEVICES] = {};  // 0 unknown, 1 ok, -1 failed; guarded by mu
    const int dev = current_device();
    if (dev < 0)
        return azh_fail(-4, "lds_range_check: hipGetDevice failed");
    std::lock_guard<std::mutex> lock(mu);
    if (state[dev] == 0) {
        unsigned *d = nullptr, h = 0xFFu;
        AZH_HIP(hipMalloc((void **)&d, 4));
        hipError_t rc = hipMemset(d, 0, 4);
        if (rc == hipSuccess) {
            hipLaunchKernelGGL(k_lds_range_probe, dim3(1), dim3(64), 4096, 0, d);
            rc = hipGetLastError();
        }
        if (rc == hipSuccess) rc = hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        (void)hipFree(d);
        AZH_HIP(rc);
        state[dev] = h == 0 ? 1 : -1;
        if (h != 0)
            fprintf(stderr, "ataxxzero_hip: device %d does not return zeros for LDS reads beyond the workgroup's allocation; "
                            "the 16-bit towers use the 32x32 variant here (or rebuild with -DAZH_OOBZERO=0)\n", dev);
    }
    return state[dev] > 0 ? 1 : 0;
}
#endif

extern "C" int azh_net_create(int blocks, int filters, const float *conv_flat, const float *bn_flat,
                              float bn_eps, azh_net **out)
{
    if (!out || !conv_flat || !bn_flat)
        return azh_fail(-1, "azh_net_create: null argument");
    if (filters != 64 && filters != 128 && filters != 256)
        return azh_fail(-2, "azh_net_create: towers are built for 64, 128 (model.py:16) and 256 filters, got %d", filters);
    const int F = filters;
    if (blocks < 0 || blocks > 64)
        return azh_fail(-2, "azh_net_create: bad block count %d", blocks);
    if (azh_require_device())
        return -3;
#if AZH_OOBZERO
    if (lds_range_check() < 0)  // probes the device once; the answer picks the tower variant in net_launch
        return -5;
#endif
    azh_net *net = new azh_net();
    net->blocks = blocks;
    net->filters = filters;
    const size_t n_conv = (size_t)9 * 4 * F + (size_t)2 * blocks * 9 * F * F + (size_t)F * 17 + F + 49 + 1;
    net->conv_flat.assign(conv_flat, conv_flat + n_conv);
    const int nbn = 2 * blocks + 1;
    std::vector<float> scale((size_t)nbn * F), shift((size_t)(nbn + 1) * F, 0.0f);  // + one look-ahead row
    for (int l = 0; l < nbn; l++)
        for (int c = 0; c < F; c++) {
            const float mean = bn_flat[((size_t)2 * l) * F + c], var = bn_flat[((size_t)2 * l + 1) * F + c];
            const double inv = 1.0 / sqrt((double)var + (double)bn_eps);
            scale[(size_t)l * F + c] = (float)inv;
            shift[(size_t)l * F + c] = (float)(-(double)mean * inv);
        }
    const float *fc = conv_flat + n_conv - 50;
    net->fc_b = fc[49];
    net->scale = scale;
    hipError_t rc = hipMalloc((void **)&net->d_shift, shift.size() * 4);
    if (rc == hipSuccess) rc = hipMalloc((void **)&net->d_fcw, 49 * 4);
    if (rc == hipSuccess) rc = hipMemcpy(net->d_shift, shift.data(), shift.size() * 4, hipMemcpyHostToDevice);
    if (rc == hipSuccess) rc = hipMemcpy(net->d_fcw, fc, 49 * 4, hipMemcpyHostToDevice);
    if (rc != hipSuccess) {
        azh_net_destroy(net);  // frees whatever was allocated
        AZH_HIP(rc);
    }
    *out = net;
    return 0;
}

extern "C" void azh_net_destroy(azh_net *net)
{
    if (!net)
        return;
    for (auto &b : net->bufs) {
        if (b.conv_w) (void)hipFree(b.conv_w);
        if (b.head_w) (void)hipFree(b.head_w);
        if (b.conv_w2) (void)hipFree(b.conv_w2);
        if (b.head_w2) (void)hipFree(b.head_w2);
    }
    if (net->d_shift) (void)hipFree(net->d_shift);
    if (net->d_fcw) (void)hipFree(net->d_fcw);
    delete net;
}


static const AdvanceHook &no_hook()
{
    static const AdvanceHook h = {};   // workers = 0
    return h;
}

// Where a launch of `tiles` tile workgroups puts its move-playing workgroups: in front when the launch has more workgroups than
// the chip has slots for them (`per_cu` workgroups of this kernel per CU), behind the tiles when everything starts at once.
static AdvanceHook place_workers(const AdvanceHook &H, int tiles, int per_cu)
{
    static int cus[MAX_DEVICES] = {};
    AdvanceHook out = H;
    if (H.workers == 0)
        return out;
    const int dev = current_device();
    if (dev >= 0 && cus[dev] == 0) {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    out.at_head = tiles + H.workers > per_cu * (dev >= 0 ? cus[dev] : 256) ? 1 : 0;
    return out;
}

template <int DT, int NB, int WPS, bool STAMP = false, int FT = F>
static int launch_tower(const TowerArgs &args, int max_n, hipStream_t stream, const AdvanceHook &H = no_hook())
{
    typedef Geo<DT, NB, FT> G;
    static_assert(G::LDS_BYTES <= 160 * 1024, "tower image does not fit one CU's LDS");
    static_assert(G::LDS_BYTES >= (int)sizeof(TreeLds), "the launch's move-playing workgroups use the tower's LDS as their scratch");
    static bool attr_set[MAX_DEVICES] = {};  // the attribute is per device: a process may drive several GPUs
    const int dev = current_device();
    if (dev < 0)
        return azh_fail(-4, "launch_tower: hipGetDevice failed");
    if (!attr_set[dev]) {
        AZH_HIP(hipFuncSetAttribute((const void *)k_tower<DT, NB, WPS, STAMP, FT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    G::LDS_BYTES));
        attr_set[dev] = true;
    }
    const int tiles = (max_n + G::BOARDS - 1) / G::BOARDS, grid = tiles + H.workers;
    if (grid <= 0)
        return 0;
    hipLaunchKernelGGL((k_tower<DT, NB, WPS, STAMP, FT>), dim3(grid), dim3(G::NTHR), G::LDS_BYTES, stream, args,
                       place_workers(H, tiles, WPS * 256 / G::NTHR > 0 ? WPS * 256 / G::NTHR : 1));
    AZH_HIP(hipGetLastError());
    return 0;
}

// Boards per workgroup for the 16-bit towers: 3 (two workgroups per CU) unless
// AZH_TOWER_BOARDS=6 asks for one 6-board workgroup per CU.
template <int DT, bool STAMP = false> static int launch_tower2(const TowerArgs &args, int max_n, hipStream_t stream,
                                                               const AdvanceHook &H = no_hook())
{
    static_assert(Geo2::LDS_BYTES >= (int)sizeof(TreeLds), "the launch's move-playing workgroups use the tower's LDS as their scratch");
    static bool attr_set[MAX_DEVICES] = {};
    const int dev = current_device();
    if (dev < 0)
        return azh_fail(-4, "launch_tower2: hipGetDevice failed");
    if (!attr_set[dev]) {
        AZH_HIP(hipFuncSetAttribute((const void *)k_tower2<DT, STAMP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    Geo2::LDS_BYTES));
        attr_set[dev] = true;
    }
    const int tiles = (max_n + Geo2::BOARDS - 1) / Geo2::BOARDS, grid = tiles + H.workers;
    if (grid <= 0)
        return 0;
    hipLaunchKernelGGL((k_tower2<DT, STAMP>), dim3(grid), dim3(NTHREADS), Geo2::LDS_BYTES, stream, args, place_workers(H, tiles, 2));
    AZH_HIP(hipGetLastError());
    return 0;
}

// one board per workgroup (Geo2Thin): the launch for a handful of boards
template <int DT> static int launch_tower2_thin(const TowerArgs &args, int max_n, hipStream_t stream, const AdvanceHook &H = no_hook())
{
    static bool attr_set[MAX_DEVICES] = {};
    const int dev = current_device();
    if (dev < 0)
        return azh_fail(-4, "launch_tower2_thin: hipGetDevice failed");
    if (!attr_set[dev]) {
        AZH_HIP(hipFuncSetAttribute((const void *)k_tower2_thin<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo2::LDS_BYTES));
        attr_set[dev] = true;
    }
    if (max_n + H.workers <= 0)
        return 0;
    hipLaunchKernelGGL((k_tower2_thin<DT>), dim3(max_n + H.workers), dim3(NTHREADS), Geo2::LDS_BYTES, stream, args,
                       place_workers(H, max_n, 2));
    AZH_HIP(hipGetLastError());
    return 0;
}

// The 16-bit towers run variant 2 (16x16x32 MFMA, 64 channels per wave); AZH_TOWER=1 selects
// variant 1 (32x32x16, 32 channels per wave), which is also the f32 tower's shape.
static int tower_variant()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("AZH_TOWER");
        v = (e && atoi(e) == 1) ? 1 : 2;
    }
    return v;
}

static int tower_boards()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("AZH_TOWER_BOARDS");
        v = (e && atoi(e) == 6) ? 6 : 3;
    }
    return v;
}

// evaluate (nn_evals.py:48-62): logits of symmetry s are brought back with the spatial inverse
// symmetry (the 17 move-type layers are NOT permuted: the reference averages them as they are) and
// averaged in the fixed order s = 0..7; the value is the mean of the eight tanh outputs.
__global__ __launch_bounds__(256) void k_sym_reduce(const float *__restrict__ sym_logits, const float *__restrict__ sym_values,
                                                    const int *__restrict__ list, const int *__restrict__ count, int n_max,
                                                    float *__restrict__ logits, float *__restrict__ values)
{
    const int n = count ? *count : n_max;
    const int i = blockIdx.x;
    if (i >= n)
        return;
    const int game = list ? list[i] : i;
    const int inv[8] = {0, 1, 2, 3, 4, 6, 5, 7};  // nn_evals.py:27
    for (int idx = threadIdx.x; idx < 833; idx += blockDim.x) {
        const int c = idx / 17, l = idx % 17;
        const int x = c / 7, y = c % 7;
        float acc = 0.0f;
        for (int s = 0; s < 8; s++) {
            int ox, oy;
            sym_cell(inv[s], x, y, ox, oy);
            acc += sym_logits[((size_t)i * 8 + s) * 833 + 17 * (7 * ox + oy) + l];
        }
        logits[(size_t)game * 833 + idx] = acc * 0.125f;
    }
    if (threadIdx.x == 0) {
        float acc = 0.0f;
        for (int s = 0; s < 8; s++)
            acc += sym_values[(size_t)i * 8 + s];
        values[game] = acc * 0.125f;
    }
}

// Evaluate up to max_n boards (dense list `list`/`count` on the device, or the
// first n boards when both are null).  Everything is indexed by game.
static int net_launch(azh_net *net, int dtype, const unsigned long long *d_boards, const int *d_list,
                      const int *d_count, int max_n, unsigned long long blockers, float *d_logits,
                      float *d_values, hipStream_t stream, unsigned long long *d_stamps, int sym, int thin = 0,
                      const void *hook = nullptr);

// thin: the caller expects a handful of boards (<= 512): one board per workgroup (Geo2Thin) where that kernel applies
int azh_net_launch(azh_net *net, int dtype, const unsigned long long *d_boards, const int *d_list,
                   const int *d_count, int max_n, unsigned long long blockers, float *d_logits,
                   float *d_values, hipStream_t stream, unsigned long long *d_stamps, int thin, const void *advance_hook)
{
    return net_launch(net, dtype, d_boards, d_list, d_count, max_n, blockers, d_logits, d_values, stream, d_stamps, 0, thin,
                      advance_hook);
}

// Symmetry-averaged evaluation: the tower runs 8 * n virtual boards into the scratch arrays
// (d_tmp_logits [8 max_n][833], d_tmp_values [8 max_n]), k_sym_reduce writes the averages by game.
int azh_net_launch_sym(azh_net *net, int dtype, const unsigned long long *d_boards, const int *d_list,
                       const int *d_count, int max_n, unsigned long long blockers, float *d_tmp_logits,
                       float *d_tmp_values, float *d_logits, float *d_values, hipStream_t stream, int thin, const void *advance_hook)
{
    if (max_n <= 0)
        return 0;
    int rc = net_launch(net, dtype, d_boards, d_list, d_count, max_n, blockers, d_tmp_logits, d_tmp_values, stream,
                        nullptr, 1, thin, advance_hook);
    if (rc)
        return rc;
    hipLaunchKernelGGL(k_sym_reduce, dim3(max_n), dim3(256), 0, stream, (const float *)d_tmp_logits,
                       (const float *)d_tmp_values, d_list, d_count, max_n, d_logits, d_values);
    AZH_HIP(hipGetLastError());
    return 0;
}

static TowerArgs tower_args(const azh_net *net, int dtype, const unsigned long long *d_boards, const int *d_list,
                            const int *d_count, int max_n, unsigned long long blockers, float *d_logits, float *d_values,
                            unsigned long long *d_stamps, int sym)
{
    TowerArgs a;
    a.conv_w = net->bufs[dtype].conv_w;
    a.head_w = net->bufs[dtype].head_w;
    a.conv_w2 = net->bufs[dtype].conv_w2;
    a.head_w2 = net->bufs[dtype].head_w2;
    a.shift = net->d_shift;
    a.fc_w = net->d_fcw;
    a.fc_b = net->fc_b;
    a.blocks = net->blocks;
    a.boards = d_boards;
    a.list = d_list;
    a.count = d_count;
    a.n = max_n;
    a.blockers = blockers;
    a.logits = d_logits;
    a.values = d_values;
    a.stamps = d_stamps;
    a.sym = sym;
    return a;
}

// Two nets, two dense leaf lists, ONE launch (k_tower2_pair): the arena's evaluator.  Returns 1 — nothing launched — where
// the fused pair kernel does not apply (f32, another width, AZH_TOWER=1, a device without the LDS range check): the caller
// then launches the two nets one after the other.  max_n bounds count_a + count_b (a game has one leaf).
int azh_net_launch_pair(azh_net *net_a, azh_net *net_b, int dtype, const unsigned long long *d_boards, const int *d_list_a,
                        const int *d_count_a, const int *d_list_b, const int *d_count_b, int max_n,
                        unsigned long long blockers, float *d_logits, float *d_values, hipStream_t stream, int thin,
                        const void *advance_hook)
{
    if (dtype != AZH_DTYPE_BF16 && dtype != AZH_DTYPE_F16)
        return 1;
    if (net_a->filters != 128 || net_b->filters != 128 || tower_variant() != 2 || !d_count_a || !d_count_b)
        return 1;
#if AZH_OOBZERO
    if (lds_range_check() != 1)
        return 1;
#endif
    if (net_pack(net_a, dtype) || net_pack(net_b, dtype))
        return -1;
    const TowerArgs a = tower_args(net_a, dtype, d_boards, d_list_a, d_count_a, max_n, blockers, d_logits, d_values, nullptr, 0);
    const TowerArgs b = tower_args(net_b, dtype, d_boards, d_list_b, d_count_b, max_n, blockers, d_logits, d_values, nullptr, 0);
    static bool attr_set[MAX_DEVICES][4] = {};
    const int dev = current_device();
    if (dev < 0)
        return azh_fail(-4, "azh_net_launch_pair: hipGetDevice failed");
    const int k = (dtype == AZH_DTYPE_BF16 ? 0 : 1) + (thin ? 2 : 0);
    const void *fn = k == 0 ? (const void *)k_tower2_pair<AZH_DTYPE_BF16> : k == 1 ? (const void *)k_tower2_pair<AZH_DTYPE_F16>
                   : k == 2 ? (const void *)k_tower2_pair<AZH_DTYPE_BF16, Geo2Thin> : (const void *)k_tower2_pair<AZH_DTYPE_F16, Geo2Thin>;
    if (!attr_set[dev][k]) {
        AZH_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, Geo2::LDS_BYTES));
        attr_set[dev][k] = true;
    }
    // ceil(nA / B) + ceil(nB / B) <= (nA + nB) / B + 2 workgroups of B boards (B = 3, or 1 for thin batches)
    const int tiles = max_n / (thin ? Geo2Thin::BOARDS : Geo2::BOARDS) + 2;
    const AdvanceHook H = place_workers(advance_hook ? *static_cast<const AdvanceHook *>(advance_hook) : no_hook(), tiles, 2);
    const int grid = tiles + H.workers;
    switch (k) {
    case 0: hipLaunchKernelGGL((k_tower2_pair<AZH_DTYPE_BF16>), dim3(grid), dim3(NTHREADS), Geo2::LDS_BYTES, stream, a, b, H); break;
    case 1: hipLaunchKernelGGL((k_tower2_pair<AZH_DTYPE_F16>), dim3(grid), dim3(NTHREADS), Geo2::LDS_BYTES, stream, a, b, H); break;
    case 2: hipLaunchKernelGGL((k_tower2_pair<AZH_DTYPE_BF16, Geo2Thin>), dim3(grid), dim3(NTHREADS), Geo2::LDS_BYTES, stream, a, b, H); break;
    default: hipLaunchKernelGGL((k_tower2_pair<AZH_DTYPE_F16, Geo2Thin>), dim3(grid), dim3(NTHREADS), Geo2::LDS_BYTES, stream, a, b, H); break;
    }
    AZH_HIP(hipGetLastError());
    return 0;
}

static int net_launch(azh_net *net, int dtype, const unsigned long long *d_boards, const int *d_list,
                      const int *d_count, int max_n, unsigned long long blockers, float *d_logits,
                      float *d_values, hipStream_t stream, unsigned long long *d_stamps, int sym, int thin, const void *hook)
{
    if (dtype < 0 || dtype > 2)
        return azh_fail(-2, "bad dtype %d", dtype);
    const AdvanceHook &H = hook ? *static_cast<const AdvanceHook *>(hook) : no_hook();
    if (net_pack(net, dtype))
        return -1;
    TowerArgs a = tower_args(net, dtype, d_boards, d_list, d_count, max_n, blockers, d_logits, d_values, d_stamps, sym);
    if (sym)
        max_n *= 8;  // grid size: virtual boards
    if (net->filters != 128) {
        // Other widths (the reference's FILTERS is a class attribute, uai_interface.py:92-93 patches it): the 32x32 tower
        // templated on the width — one wave per 32 output channels, 3 boards per workgroup (1 for f32 at 256 filters: LDS).
        if (d_stamps)
            return azh_fail(-2, "stamps are built for the 128-filter bf16 tower only");
        if (net->filters == 64) {
            switch (dtype) {
            case AZH_DTYPE_F32: return launch_tower<AZH_DTYPE_F32, 3, 1, false, 64>(a, max_n, stream, H);
            case AZH_DTYPE_BF16: return launch_tower<AZH_DTYPE_BF16, 3, 1, false, 64>(a, max_n, stream, H);
            default: return launch_tower<AZH_DTYPE_F16, 3, 1, false, 64>(a, max_n, stream, H);
            }
        }
        switch (dtype) {
        case AZH_DTYPE_F32: return launch_tower<AZH_DTYPE_F32, 1, 2, false, 256>(a, max_n, stream, H);
        case AZH_DTYPE_BF16: return launch_tower<AZH_DTYPE_BF16, 3, 2, false, 256>(a, max_n, stream, H);
        default: return launch_tower<AZH_DTYPE_F16, 3, 2, false, 256>(a, max_n, stream, H);
        }
    }
    const bool six = tower_boards() == 6;
    bool v2 = tower_variant() == 2 && dtype != AZH_DTYPE_F32;
#if AZH_OOBZERO
    if (v2 && lds_range_check() != 1)
        v2 = false;
#endif
    if (v2) {
        if (d_stamps)
            return dtype == AZH_DTYPE_BF16 ? launch_tower2<AZH_DTYPE_BF16, true>(a, max_n, stream, H)
                                           : azh_fail(-2, "stamps are built for bf16 only");
        if (thin)
            return dtype == AZH_DTYPE_BF16 ? launch_tower2_thin<AZH_DTYPE_BF16>(a, max_n, stream, H)
                                           : launch_tower2_thin<AZH_DTYPE_F16>(a, max_n, stream, H);
        return dtype == AZH_DTYPE_BF16 ? launch_tower2<AZH_DTYPE_BF16>(a, max_n, stream, H)
                                       : launch_tower2<AZH_DTYPE_F16>(a, max_n, stream, H);
    }
    if (d_stamps) {  // diagnostic instantiations (bf16 only)
        if (dtype != AZH_DTYPE_BF16)
            return azh_fail(-2, "stamps are built for bf16 only");
        return six ? launch_tower<AZH_DTYPE_BF16, 6, 1, true>(a, max_n, stream, H)
                   : launch_tower<AZH_DTYPE_BF16, 3, 2, true>(a, max_n, stream, H);
    }
    switch (dtype) {
    case AZH_DTYPE_F32: return launch_tower<AZH_DTYPE_F32, 3, 1>(a, max_n, stream, H);
    case AZH_DTYPE_BF16:
        return six ? launch_tower<AZH_DTYPE_BF16, 6, 1>(a, max_n, stream, H) : launch_tower<AZH_DTYPE_BF16, 3, 2>(a, max_n, stream, H);
    default:
        return six ? launch_tower<AZH_DTYPE_F16, 6, 1>(a, max_n, stream, H) : launch_tower<AZH_DTYPE_F16, 3, 2>(a, max_n, stream, H);
    }
}

// device temporary owned by a host function: released when the function returns, error paths included
struct DevBuf {
    void *p = nullptr;
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    template <typename T> T *as() const { return (T *)p; }
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

static int net_forward(azh_net *net, int dtype, int n, const uint64_t *leaf_boards, uint64_t blockers, float *logits_out,
                       float *values_out, int thin);

extern "C" int azh_net_forward(azh_net *net, int dtype, int n, const uint64_t *leaf_boards,
                               uint64_t blockers, float *logits_out, float *values_out)
{
    return net_forward(net, dtype, n, leaf_boards, blockers, logits_out, values_out, 0);
}

extern "C" int azh_net_forward_thin(azh_net *net, int dtype, int n, const uint64_t *leaf_boards,
                                    uint64_t blockers, float *logits_out, float *values_out)
{
    return net_forward(net, dtype, n, leaf_boards, blockers, logits_out, values_out, 1);
}

static int net_forward(azh_net *net, int dtype, int n, const uint64_t *leaf_boards, uint64_t blockers, float *logits_out,
                       float *values_out, int thin)
{
    if (!net || !leaf_boards || !logits_out || !values_out || n < 0)
        return azh_fail(-1, "azh_net_forward: bad argument");
    if (n == 0)
        return 0;
    DevBuf d_b, d_l, d_v;  // freed on every path out of this function
    AZH_HIP(d_b.alloc((size_t)n * 16));
    AZH_HIP(d_l.alloc((size_t)n * 833 * 4));
    AZH_HIP(d_v.alloc((size_t)n * 4));
    AZH_HIP(hipMemcpy(d_b.p, leaf_boards, (size_t)n * 16, hipMemcpyHostToDevice));
    const int rc = azh_net_launch(net, dtype, d_b.as<unsigned long long>(), nullptr, nullptr, n, blockers, d_l.as<float>(),
                                  d_v.as<float>(), 0, nullptr, thin);
    if (rc)
        return rc;
    AZH_HIP(hipDeviceSynchronize());
    AZH_HIP(hipMemcpy(logits_out, d_l.p, (size_t)n * 833 * 4, hipMemcpyDeviceToHost));
    AZH_HIP(hipMemcpy(values_out, d_v.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int azh_net_forward_sym(azh_net *net, int dtype, int n, const uint64_t *leaf_boards,
                                   uint64_t blockers, float *logits_out, float *values_out)
{
    if (!net || !leaf_boards || !logits_out || !values_out || n < 0)
        return azh_fail(-1, "azh_net_forward_sym: bad argument");
    if (n == 0)
        return 0;
    DevBuf d_b, d_l, d_v, d_tl, d_tv;
    AZH_HIP(d_b.alloc((size_t)n * 16));
    AZH_HIP(d_l.alloc((size_t)n * 833 * 4));
    AZH_HIP(d_v.alloc((size_t)n * 4));
    AZH_HIP(d_tl.alloc((size_t)n * 8 * 833 * 4));
    AZH_HIP(d_tv.alloc((size_t)n * 8 * 4));
    AZH_HIP(hipMemcpy(d_b.p, leaf_boards, (size_t)n * 16, hipMemcpyHostToDevice));
    const int rc = azh_net_launch_sym(net, dtype, d_b.as<unsigned long long>(), nullptr, nullptr, n, blockers,
                                      d_tl.as<float>(), d_tv.as<float>(), d_l.as<float>(), d_v.as<float>(), 0);
    if (rc)
        return rc;
    AZH_HIP(hipDeviceSynchronize());
    AZH_HIP(hipMemcpy(logits_out, d_l.p, (size_t)n * 833 * 4, hipMemcpyDeviceToHost));
    AZH_HIP(hipMemcpy(values_out, d_v.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

// Net-only timing hook: `iters` launches of the tower over n synthetic boards, HIP-event
// timed on one stream; *ms_out = average milliseconds per launch.
static int net_bench(azh_net *net, int dtype, int n, int iters, float *ms_out, int thin);
extern "C" int azh_net_bench(azh_net *net, int dtype, int n, int iters, float *ms_out) { return net_bench(net, dtype, n, iters, ms_out, 0); }
extern "C" int azh_net_bench_thin(azh_net *net, int dtype, int n, int iters, float *ms_out) { return net_bench(net, dtype, n, iters, ms_out, 1); }

static int net_bench(azh_net *net, int dtype, int n, int iters, float *ms_out, int thin)
{
    if (!net || n <= 0 || iters <= 0 || !ms_out)
        return azh_fail(-1, "azh_net_bench: bad argument");
    std::vector<unsigned long long> boards((size_t)n * 2);
    unsigned long long x = 0x9E3779B97F4A7C15ULL;
    for (int i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const unsigned long long m = x & 0x1FFFFFFFFFFFFULL;
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        boards[2 * (size_t)i] = m;
        boards[2 * (size_t)i + 1] = x & 0x1FFFFFFFFFFFFULL & ~m;
    }
    unsigned long long *d_b = nullptr;
    float *d_l = nullptr, *d_v = nullptr;
    AZH_HIP(hipMalloc((void **)&d_b, (size_t)n * 16));
    AZH_HIP(hipMalloc((void **)&d_l, (size_t)n * 833 * 4));
    AZH_HIP(hipMalloc((void **)&d_v, (size_t)n * 4));
    AZH_HIP(hipMemcpy(d_b, boards.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    hipStream_t st;
    hipEvent_t e0, e1;
    AZH_HIP(hipStreamCreate(&st));
    AZH_HIP(hipEventCreate(&e0));
    AZH_HIP(hipEventCreate(&e1));
    int rc = 0;
    for (int i = 0; i < 3 && rc == 0; i++)
        rc = azh_net_launch(net, dtype, d_b, nullptr, nullptr, n, 0, d_l, d_v, st, nullptr, thin);
    AZH_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters && rc == 0; i++)
        rc = azh_net_launch(net, dtype, d_b, nullptr, nullptr, n, 0, d_l, d_v, st, nullptr, thin);
    AZH_HIP(hipEventRecord(e1, st));
    AZH_HIP(hipStreamSynchronize(st));
    float ms = 0.0f;
    AZH_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / (float)iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(st);
    (void)hipFree(d_b);
    (void)hipFree(d_l);
    (void)hipFree(d_v);
    return rc;
}

// Diagnostic: one launch of the stamped bf16 tower over n synthetic boards; copies the
// s_memtime stamps of the first `wgs` workgroups ([wg][wave][128] u64) to `out`.
extern "C" int azh_net_stamps(azh_net *net, int n, int wgs, uint64_t *out)
{
    if (!net || n <= 0 || wgs <= 0 || !out)
        return azh_fail(-1, "azh_net_stamps: bad argument");
    if (net->filters != 128)
        return azh_fail(-2, "azh_net_stamps: built for the 128-filter bf16 tower only");
    const int grid_max = (n + 2) / 3;
    std::vector<unsigned long long> boards((size_t)n * 2);
    unsigned long long x = 0x9E3779B97F4A7C15ULL;
    for (int i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const unsigned long long m = x & 0x1FFFFFFFFFFFFULL;
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        boards[2 * (size_t)i] = m;
        boards[2 * (size_t)i + 1] = x & 0x1FFFFFFFFFFFFULL & ~m;
    }
    unsigned long long *d_b = nullptr, *d_s = nullptr;
    float *d_l = nullptr, *d_v = nullptr;
    const size_t sbytes = (size_t)grid_max * OCT * 128 * 8;
    AZH_HIP(hipMalloc((void **)&d_b, (size_t)n * 16));
    AZH_HIP(hipMalloc((void **)&d_l, (size_t)n * 833 * 4));
    AZH_HIP(hipMalloc((void **)&d_v, (size_t)n * 4));
    AZH_HIP(hipMalloc((void **)&d_s, sbytes));
    AZH_HIP(hipMemset(d_s, 0, sbytes));
    AZH_HIP(hipMemcpy(d_b, boards.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    int rc = 0;
    for (int i = 0; i < 3 && rc == 0; i++)
        rc = azh_net_launch(net, AZH_DTYPE_BF16, d_b, nullptr, nullptr, n, 0, d_l, d_v, 0, d_s);
    AZH_HIP(hipDeviceSynchronize());
    const int have = wgs < grid_max ? wgs : grid_max;
    AZH_HIP(hipMemcpy(out, d_s, (size_t)have * OCT * 128 * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_b); (void)hipFree(d_l); (void)hipFree(d_v); (void)hipFree(d_s);
    return rc;
}
