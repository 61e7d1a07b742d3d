// fp32-accurate nn.Linear on the bf16 matrix pipe ("split-bf16"): C = act(A W^T + b) with every fp32 operand taken as the
// exact sum of three bf16 numbers, a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1), a3 = a - a1 - a2: 3 x 8 significant
// bits), and the six products a_i w_j with i + j <= 4 evaluated on v_mfma_f32_16x16x32_bf16 (16x the rate of the fp32 MFMA;
// every single product is exact in fp32) into fp32 accumulators that are flushed into f64 running sums every SECOND 32-deep K
// stage.  What is dropped (a2 w3 + a3 w2 + a3 w3) is below 2^-24 of |a||w| per product.  Measured against exactly evaluated
// dot products on MLP-shaped operands (tools/sb16_numerics.hip, profiles/r04_sb16_numerics.txt): rms 0.25-0.26 ulp of the
// result's scale at K = 416 ... 3072 -- the same as the fp32 MFMA chain with a flush per stage that the MLP launches ran on
// (0.26), with half as many flushes and 6 x 16-cycle instead of 8 x 32-cycle matrix instructions per 16 x 16 x 32 block.
//
// Replaces the same reference code as gemm.hip (nn.Linear + LeakyReLU of utils/mlp.py:8-28, gat2.py:53-55) for the MLP launches
// (MLP mode 3) and the GAT launches of layers >= 1 (GAT mode 4).  Activations stay fp32 in memory and are split in registers by
// the MFMA waves (88 vector instructions per stage and wave); the weights are split once (k_split_planes) into three bf16 planes
// [3][rows][ldw].
//
// Three kernels, one arithmetic -- per 16 x 16 output tile and K stage the six MFMAs in the order (a1,w3) (a2,w2) (a1,w2)
// (a3,w1) (a2,w1) (a1,w1), stages ascending; with f64 sums (MLP) one fp32 chain per stage PAIR (started from a zero C operand,
// added to the f64 running sum after the odd stage and after the last one; per STAGE in the maximum-accuracy mode), without
// (GAT) the even and the odd stages in two fp32 chains added at the end -- so a row has the same bits in a batch of one and of a
// thousand:
//   k_linear_sb         tile kernel: ONE twelve-wave workgroup per CU, persistent; 256 x 64 | 80 tiles, eight MFMA waves, four loader
//                       waves (LDS-DMA of the fp32 activation tile + three weight planes per stage, `saddr + voffset` loads: no
//                       vector instruction per piece), a ring of three stages, one barrier per stage, the second MFMA wave of every
//                       SIMD half a stage behind the first
//   k_linear_sb_skinny  one wave per 16 x 16 tile, operands streamed from global memory (small batches, narrow outputs)
//   k_linear_sb_ks      the same with the stage pairs of a tile dealt to eight waves and an ordered f64 reduction through LDS
// What bounds the tile kernel: POWER.  In-kernel clock of its MLP launches on real operands 2.0 GHz (2.4 GHz
// with zero weights, same cycle count); the 32 x 32 x 16 instruction in three schedules saved 6-12 % of the cycles and ran at
// 1.68-1.76 GHz -- the same wall time (profiles/r05_sb_clock.txt, r05_sb32_forms.txt; those kernels: git show 80a9985).
#include <cstdlib>
#include <type_traits>

#include "mpe_internal.h"
#include "sb_common.h"

#ifndef SB_PRIO_MFMA
#define SB_PRIO_MFMA 2
#endif
#ifndef SB_PRIO_SPLIT
#define SB_PRIO_SPLIT 0        // priority of an MFMA wave while it splits its fragments (its MFMA phases: 2, its flush: 0, loaders: 3)
#endif

namespace mpe {

// (kernels in namespace mpe, not an anonymous one: profilers print "(anonymous namespace)::" in front of such names and the
// tools that cut a kernel name at its first parenthesis then see nothing)
namespace sb {

// In-kernel clock of a launch (MI355X_MICROARCH.md, DVFS give-back item 6), DIAGNOSTIC builds only (make exp EXPFLAGS=-DMPE_SB_CLOCK,
// tools/sb_clock_probe.py): the first MFMA wave of every workgroup stamps the shader clock (s_memtime) and the constant 100 MHz
// clock (s_memrealtime) around its tile loop into a buffer nothing else reads.  The product build has no stamp.
#ifdef MPE_SB_CLOCK
// (one bucket per launch shape -- (n ^ k_pad / 32) & 15 -- so that the launches of a whole GAT or MLP pass keep their stamps apart)
__device__ unsigned long long g_sb_stamp[16][256][4];
#define SB_STAMP(SLOT)                                                                              \
    do {                                                                                            \
        if (threadIdx.x == 0) {                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            unsigned long long *st_ = g_sb_stamp[(n ^ (k_pad >> 5)) & 15][blockIdx.x & 255];        \
            st_[SLOT] = __builtin_amdgcn_s_memtime();                                               \
            st_[(SLOT) + 1] = __builtin_amdgcn_s_memrealtime();                                     \
            __builtin_amdgcn_sched_barrier(0);                                                      \
        }                                                                                           \
    } while (0)
#else
#define SB_STAMP(SLOT) do { } while (0)
#endif

__device__ __forceinline__ int a_swz(int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 2); }   // gemm.hip: dma_swz
// 64-byte rows (weight planes): chunk c of row r sits at position c ^ w_swz(r).  ds_read_b128 serves its 64 lanes in four
// NON-contiguous groups of 16 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, and the same + 32: MI355X_MICROARCH.md, LDS table);
// in the fragment read (lane = 16 fq + fr reads chunk fq of row fr) a group therefore holds all sixteen rows, rows 0-3 and
// 12-15 with one chunk index and rows 4-11 with the other of a pair (0 | 1 or 2 | 3).  The four rows that share (row & 3) --
// the same 64-byte quarter of the 256-byte bank line -- must land on four different positions: the table {0, 2, 3, 1} over
// (row >> 2) & 3 does that for all four groups (with the plain (row >> 2) & 3 of the first version rows r and r + 4 of a
// group collided: SQ_LDS_BANK_CONFLICT was 43 % of the LDS-active cycles, 7.0 instead of 4 cycles per ds_read_b128).
__device__ __forceinline__ int w_swz(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

// One LDS-DMA piece (1 KiB per wave-instruction) in the `saddr + voffset` form: wave-uniform 64-bit base in an SGPR pair, per-lane
// 32-bit byte offset, LDS destination (wave-uniform byte address) through M0, written in the statement that reads it.  hipcc does
// not count asm loads: the caller waits with an explicit `s_waitcnt vmcnt(0)` before the stage barrier.
__device__ __forceinline__ void glds16(unsigned voff, const void *sbase, unsigned lds_dst) {
    unsigned keep;
#ifndef SB_GLDS_MOD
#define SB_GLDS_MOD ""            // cache-policy bits of the staging loads (experiments: " sc0", " sc1", " nt")
#endif
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" SB_GLDS_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// One K stage in LDS: the activation tile, fp32, the image of k_linear_dma (256 rows x 128 B = 32 KiB), and the three weight planes
// (16 NTT rows x 64 B each).  Three stages are resident: the second MFMA wave of every SIMD runs half a stage behind the first and
// still reads stage kt - 1 while stage kt + 1 lands.
constexpr int SB_MW = 8;             // MFMA waves per workgroup (32 rows each): 256-row tiles
constexpr int SB_NL = 4;             // loader waves
constexpr int SB_RING = 3;
__host__ __device__ constexpr int sb_a_bytes() { return SB_MW * 32 * GEMM_BK * 4; }
__host__ __device__ constexpr int sb_stage_bytes(int ntt) { return sb_a_bytes() + 3 * ntt * 16 * GEMM_BK * 2; }

// The tile kernel: ONE twelve-wave workgroup per CU -- eight MFMA waves of 32 rows x 16 NTT features (256 x 64 tiles with f64 sums,
// 256 x 80 without) and four loader waves -- PERSISTENT: the workgroup walks the tiles bid, bid + gridDim.x, ... (gridDim.x a
// multiple of eight: a workgroup's tiles keep its XCD and the tile order of a plain launch), the loader waves run on across tile
// borders (stage 0 of the next tile is issued behind the barrier of the last stage of this one) and the result stores of a tile
// drain while the next one is multiplied.  (Forms that were measured and are gone: six-wave workgroups, which never ran two per
// CU -- the dispatcher deals a workgroup's waves to the SIMDs 2-2-1-1 from the same SIMD each time --; four MFMA waves and two
// stages at two workgroups per CU; one workgroup per tile; the 32 x 32 x 16 instruction, round 5: DESIGN.md 7.1.)
// A12 (fc2 of a graph-attention layer with 40-wide attention heads, 80-wide tiles): the epilogue also emits a1 | a2 =
// <ft2[row, head, :], attn_l / attn_r[head]> (gat2.py:57-58) from the values the lanes hold -- the code of k_linear_dma<.., A12>,
// same lane layout, same canonical order (coef40() in gat.hip mirrors it for the paths that compute the coefficients elsewhere).
// FL (F64 only) = K stages per f64 flush: 2 = the default cadence (an fp32 chain per stage PAIR), 1 = a chain and a flush per stage
// (the maximum-accuracy mode of the MLP, mpe_set_precision MLP 4: rms error 0.13-0.18 instead of 0.24-0.26 ulp of the output scale,
// tools/sb16_numerics.hip).
template <bool LEAKY, int NTT, bool F64, bool A12 = false, int FL = 2>
__global__ __launch_bounds__(64 * (SB_MW + SB_NL), 3) void k_linear_sb(const float *__restrict__ A, int lda,
                                                                   const unsigned short *__restrict__ W3, size_t w_plane, int ldw,
                                                                   const float *__restrict__ bias, float *__restrict__ C, int ldc,
                                                                   int m_cap, const int32_t *__restrict__ d_m, int n, int k_pad,
                                                                   float slope, int ntn, int n_major,
                                                                   const float *__restrict__ attn_l = nullptr,
                                                                   const float *__restrict__ attn_r = nullptr,
                                                                   float *__restrict__ a12 = nullptr, int out_half = 0) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    // Stage barrier: this wave's LDS operations done + s_barrier, and NOT __syncthreads(): its fence also waits for the wave's global
    // stores, i.e. for the result rows of the previous tile (what must have landed in LDS is waited for by the loader waves
    // themselves: their explicit vmcnt(0) in front of this barrier).
    auto stage_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    constexpr int MW = SB_MW, MT = 2, RING = SB_RING;
    static_assert(FL == 2 || (FL == 1 && F64), "flush cadence: every stage pair, or every stage of an f64-sum launch");
    constexpr int STAGE = sb_stage_bytes(NTT);
    constexpr int SB_A_BYTES = sb_a_bytes(), BM = 16 * MT * MW;
    constexpr int WPL = NTT * 16 * GEMM_BK * 2;       // bytes of one weight plane of a stage
    int M = m_cap;
    if (d_m) {
        const int dm = *d_m;
        M = dm < m_cap ? dm : m_cap;
    }
    const int ntm = (M + BM - 1) / BM;
    const int bid = blockIdx.x, nwg = ntm * ntn;
    if (bid >= nwg) return;
    const int vstep = (int)gridDim.x;                   // tiles of this workgroup: bid, bid + vstep, ...
    // tile of (virtual) workgroup v: XCD-aware order (v & 7 = the XCD the hardware dispatcher gives workgroup v, and v + 8 k stays there)
    auto tile_of = [&](int v, int &m0o, int &n0o) {
        const int xcd = v & 7, q = nwg >> 3, r = nwg & 7;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        int tm, tn;
        if (n_major) {
            constexpr int RB = 8;
            const int band = swz / (RB * ntn), rem = swz - band * (RB * ntn);
            const int rows = ntm - band * RB < RB ? ntm - band * RB : RB;
            tn = rem / rows;
            tm = band * RB + (rem - tn * rows);
        } else {
            tm = swz / ntn;
            tn = swz - tm * ntn;
        }
        m0o = tm * BM;
        n0o = tn * NTT * 16;
    };
    int m0, n0;
    tile_of(bid, m0, n0);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nk = k_pad / GEMM_BK;

    if (wave >= MW) {
        // ---- loader waves: 32 activation groups (8 rows x 128 B each) + 3 NTT weight groups (16 rows x 64 B) per stage ----
        const int li = wave - MW;
        const int dr = lane >> 3, dp = lane & 7, wr = lane >> 2, wc = lane & 3;
        __builtin_amdgcn_s_setprio(3);
        constexpr int NA = (BM / 8) / SB_NL;
        constexpr int NW = 3 * NTT, NWL = (NW + SB_NL - 1) / SB_NL;
        // Addresses as a UNIFORM base (SGPR pair, advanced by scalar adds per stage) + a loop-invariant 32-bit byte offset per lane:
        // the loads then take the `saddr + voffset` form and a stage's pieces cost the loader wave no vector instruction at all.
        // (With per-lane 64-bit pointers every piece carried a v_lshl_add_u64 that had to find a slot on a SIMD whose vector issue
        // the two MFMA waves keep busy -- the loader waves, not the memory path, paced the stage.)
        const unsigned char *abase, *wbase;
        unsigned la[NA], lw[NWL];
        int lw_dst[NWL];
        auto setup = [&]() {                                // the tile at (m0, n0)
            abase = reinterpret_cast<const unsigned char *>(A + (size_t)m0 * lda);
            wbase = reinterpret_cast<const unsigned char *>(W3 + (size_t)n0 * ldw);
#pragma unroll
            for (int g = 0; g < NA; ++g) {
                const int row = (li * NA + g) * 8 + dr;
                int grow = m0 + row;
                grow = grow < M ? grow : M - 1;             // (M - 1 >= m0: the tile exists)
                la[g] = (unsigned)(((grow - m0) * lda + ((dp ^ a_swz(row)) << 2)) * 4);
            }
        };
        setup();
#pragma unroll
        for (int g = 0; g < NWL; ++g) {
            int idx = li + SB_NL * g;
            idx = idx < NW ? idx : NW - 1;
            const int p = idx / NTT, grp = idx - p * NTT;
            const int row = grp * 16 + wr;
            lw[g] = (unsigned)((p * w_plane + (size_t)row * ldw + ((wc ^ w_swz(row)) << 3)) * 2);
            lw_dst[g] = SB_A_BYTES + p * WPL + grp * 1024;
        }
        auto fill = [&](int kt, int buf) {
            const unsigned base = (unsigned)(size_t)(lds_void *)lds + (unsigned)(buf * STAGE);      // buf = kt mod RING
            const unsigned char *ab = abase + (size_t)kt * (GEMM_BK * 4), *wb = wbase + (size_t)kt * (GEMM_BK * 2);
#pragma unroll
            for (int g = 0; g < NA; ++g) glds16(la[g], ab, base + (unsigned)((li * NA + g) * 8 * 128));
#pragma unroll
            for (int g = 0; g < NWL; ++g)
                if ((g + 1) * SB_NL <= NW || li + SB_NL * g < NW) glds16(lw[g], wb, base + (unsigned)lw_dst[g]);
        };
        // One stage ahead: stage kt + 1 is issued after barrier kt and has landed before barrier kt + 1.  The buffer it goes to is
        // the one of stage kt - 2: stage kt - 1 is still being read by the second wave of every SIMD between barriers kt and kt + 1.
        fill(0, 0);
        int nb = 1;
        for (int v = bid; v < nwg; v += vstep) {
            const bool more = v + vstep < nwg;
            for (int kt = 0; kt < nk; ++kt) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stage kt has landed ...
                stage_barrier();                                        // ... and nobody reads the buffer stage kt + 1 goes to
                if (kt + 1 < nk) {
                    fill(kt + 1, nb);
                } else if (more) {                                      // behind the last stage: stage 0 of this workgroup's next tile
                    tile_of(v + vstep, m0, n0);
                    setup();
                    fill(0, nb);
                }
                nb = nb + 1 == RING ? 0 : nb + 1;
            }
        }
        return;
    }

    // ---- MFMA waves: 32 rows x 16 NTT features each ----
    const int fq = lane >> 4, fr = lane & 15;
    int a_rd[MT], w_rd[NTT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_rd[mt] = (wave * 16 * MT + mt * 16 + fr) * 128;
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) {
        const int row = nt * 16 + fr;
        w_rd[nt] = SB_A_BYTES + row * 64 + ((fq ^ w_swz(row)) << 4);
    }
    const int fsw = a_swz(fr);
    const int c0 = ((fq * 2 + 0) ^ fsw) << 4, c1 = ((fq * 2 + 1) ^ fsw) << 4;      // k = 8 fq .. 8 fq + 7 of the row

    // Without f64 sums (GAT launches) the even and the odd K stages accumulate into separate fp32 chains that are added at the
    // end: half the chain length of a single accumulator (rms error 0.5-0.6 ulp of the output scale at K = 416 against 0.75 for one
    // chain and 0.91 for the fp32 MFMA chain it replaces).
    int b = 0;                                         // ring buffer of the stage in hand (runs on across the tiles of the workgroup)
    SB_STAMP(0);
    for (int v = bid; v < nwg; v += vstep) {
    if (v != bid) tile_of(v, m0, n0);
    f32x4 acc[NTT][MT];
    f32x4 acc_odd[NTT][MT];               // (unused with F64: the compiler drops it)
    double run[F64 ? NTT : 1][MT][4];
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc_odd[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (F64) {
#pragma unroll
                for (int i = 0; i < 4; ++i) run[F64 ? nt : 0][mt][i] = 0.0;
            }
        }
    {
        // Eight MFMA waves, ring of three stages, the two MFMA waves of a SIMD HALF A STAGE APART.  A stage of a wave is: split the
        // activation fragments (88 vector instructions, nothing to multiply yet), then the six-product chains of the first and of the
        // second half of the column tiles.  Waves 0-3 run a stage between two barriers.  Waves 4-7 (the second wave of every SIMD)
        // run the loop rotated by half a stage: after barrier kt they finish stage kt - 1 (second half of its column tiles: MFMAs
        // only) and then start stage kt (split + first half) -- so one wave's split arithmetic falls into the other's MFMA-only half,
        // where two waves in step would both sit in front of their first MFMA.  Stage kt - 1 is still read after barrier kt: that is
        // what the third buffer is for (the loaders refill it after barrier kt + 1).  Every 16 x 16 tile still sees its six products
        // per stage in the canonical order, stages ascending: the bits of the wave-per-tile kernels.
        bf16x8 ap[MT][3];
        auto split_stage = [&](const unsigned char *buf) {
            // f64-sum launches: a wave in its vector-heavy phases (split arithmetic, flush) yields the issue port to the other wave
            // of its SIMD while that one multiplies (MLP launches 159.3 -> 153.8 us; the plain GAT launches lost 0.8 % with it)
            if (F64) __builtin_amdgcn_s_setprio(SB_PRIO_SPLIT);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(buf + a_rd[mt] + c0);
                const f32x4 x1 = *reinterpret_cast<const f32x4 *>(buf + a_rd[mt] + c1);
                split8(x0, x1, ap[mt][0], ap[mt][1], ap[mt][2]);
            }
        };
        constexpr int NH = (NTT + 1) / 2;              // column tiles of the first half
        auto half = [&](const unsigned char *buf, const int h, f32x4 (&ACC)[NTT][MT], auto from_zero) {
            if (F64) __builtin_amdgcn_s_setprio(SB_PRIO_MFMA);
#pragma unroll
            for (int nt = h ? NH : 0; nt < (h ? NTT : NH); ++nt) {
                bf16x8 wp[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) wp[p] = *reinterpret_cast<const bf16x8 *>(buf + p * WPL + w_rd[nt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (decltype(from_zero)::value) SB_STAGE0(ACC[nt][mt], ap[mt], wp);
                    else SB_STAGE(ACC[nt][mt], ap[mt], wp);
                }
            }
        };
        // half h of an even / odd stage (f64 sums: one fp32 chain per flush interval, started from a zero C operand; without:
        // even / odd accumulators)
        auto even_half = [&](const unsigned char *buf, const int h) {
            if (F64) half(buf, h, acc, std::true_type());
            else half(buf, h, acc, std::false_type());
        };
        auto odd_half = [&](const unsigned char *buf, const int h) {
            if (F64) half(buf, h, acc, std::false_type());
            else half(buf, h, acc_odd, std::false_type());
        };
        auto flush = [&]() {                           // the chains' sums into the f64 running sums
            if (!F64) return;
            __builtin_amdgcn_s_setprio(0);
            if (FL == 1) __builtin_amdgcn_sched_barrier(0);      // (flush per stage: left to mix the flush with the next stage's fragment reads the compiler spills)
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) run[F64 ? nt : 0][mt][i] += (double)acc[nt][mt][i];
            if (FL == 1) __builtin_amdgcn_sched_barrier(0);
        };
        auto buf_at = [&](int i) { return lds + i * STAGE; };
        auto next_b = [&](int i) { return i + 1 == RING ? 0 : i + 1; };
        if constexpr (FL == 1) {
            // a chain and a flush per stage: every stage is an "even" one (its chains start from a zero C operand)
            if (wave < MW / 2) {
#pragma unroll 1
                for (int kt = 0; kt < nk; ++kt) {
                    stage_barrier();                   // barrier kt: stage kt has landed
                    split_stage(buf_at(b));
                    even_half(buf_at(b), 0);
                    even_half(buf_at(b), 1);
                    b = next_b(b);
                    flush();
                }
            } else {
                stage_barrier();                       // barrier 0 (of this tile)
                split_stage(buf_at(b));
                even_half(buf_at(b), 0);
#pragma unroll 1
                for (int kt = 1; kt < nk; ++kt) {
                    stage_barrier();                   // barrier kt
                    even_half(buf_at(b), 1);           // second half of stage kt - 1
                    flush();
                    b = next_b(b);
                    split_stage(buf_at(b));
                    even_half(buf_at(b), 0);
                }
                even_half(buf_at(b), 1);
                flush();
                b = next_b(b);                         // the first stage of the next tile
            }
        } else
        if (wave < MW / 2) {
#pragma unroll 1
            for (int kt = 0; kt < nk; kt += 2) {
                stage_barrier();                       // barrier kt: stage kt has landed
                split_stage(buf_at(b));
                even_half(buf_at(b), 0);
                even_half(buf_at(b), 1);
                b = next_b(b);
                if (kt + 1 >= nk) flush();
                if (kt + 1 < nk) {
                    stage_barrier();
                    split_stage(buf_at(b));
                    odd_half(buf_at(b), 0);
                    odd_half(buf_at(b), 1);
                    b = next_b(b);
                    flush();
                }
            }
        } else {
            stage_barrier();                           // barrier 0 (of this tile)
            split_stage(buf_at(b));
            even_half(buf_at(b), 0);
            int kt = 1;
#pragma unroll 1
            for (; kt + 1 < nk; kt += 2) {             // kt odd
                stage_barrier();                       // barrier kt
                even_half(buf_at(b), 1);               // second half of stage kt - 1
                b = next_b(b);
                split_stage(buf_at(b));
                odd_half(buf_at(b), 0);
                stage_barrier();                       // barrier kt + 1
                odd_half(buf_at(b), 1);
                flush();
                b = next_b(b);
                split_stage(buf_at(b));
                even_half(buf_at(b), 0);
            }
            if (kt < nk) {                             // nk even: one odd stage left
                stage_barrier();
                even_half(buf_at(b), 1);
                b = next_b(b);
                split_stage(buf_at(b));
                odd_half(buf_at(b), 0);
                odd_half(buf_at(b), 1);
            } else {                                   // nk odd: the last (even) stage has its first half done
                even_half(buf_at(b), 1);
            }
            flush();
            b = next_b(b);                             // the first stage of the next tile
        }
    }
    float pl[2][2], pr[2][2];            // A12: per (head of the tile, row tile) partial dot products
    if (A12) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) pl[h][mt] = pr[h][mt] = 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) {
        const int nb = n0 + nt * 16 + fq * 4;
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + nb);
        f32x4 al = {0.f, 0.f, 0.f, 0.f}, ar = {0.f, 0.f, 0.f, 0.f};
        if (A12 && nb + 3 < n) {
            al = *reinterpret_cast<const f32x4 *>(attn_l + nb);
            ar = *reinterpret_cast<const f32x4 *>(attn_r + nb);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + wave * 16 * MT + mt * 16 + fr;
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = F64 ? (float)(run[F64 ? nt : 0][mt][i] + (double)bv[i]) : (acc[nt][mt][i] + acc_odd[nt][mt][i]) + bv[i];
                if (LEAKY) v[i] = v[i] > 0.f ? v[i] : v[i] * slope;
            }
            if (A12) {
                // a 4-feature group never straddles the 40-feature head boundary of the tile
                const int hp = (nt * 16 + fq * 4) >= 40 ? 1 : 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float l1 = __builtin_fmaf(v[i], al[i], hp ? pl[1][mt] : pl[0][mt]);
                    const float r1 = __builtin_fmaf(v[i], ar[i], hp ? pr[1][mt] : pr[0][mt]);
                    if (hp) { pl[1][mt] = l1; pr[1][mt] = r1; } else { pl[0][mt] = l1; pr[0][mt] = r1; }
                }
            }
            if (m >= M) continue;
            if (out_half) {                    // fp16 rows for the attention stage (configs[4]); ldc counts halves
                _Float16 *dh = reinterpret_cast<_Float16 *>(C) + (size_t)m * ldc + nb;
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                if (nb + 3 < n) {
                    *reinterpret_cast<h4 *>(dh) = (h4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (nb + i < n) dh[i] = (_Float16)v[i];
                }
                continue;
            }
            float *dst = C + (size_t)m * ldc + nb;
            if (nb + 3 < n) {
                *reinterpret_cast<f32x4 *>(dst) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (nb + i < n) dst[i] = v[i];
            }
        }
    }
    if (A12) {
        // (s0 + s1) + (s2 + s3) over the four lane groups: after the two exchanges all four lanes of a row hold the same bits, so
        // lane group 0 stores the tile's a1 pair and lane group 1 its a2 pair, eight bytes each (heads 2 tn and 2 tn + 1 are
        // neighbours in a row of a1 | a2) -- one store instruction per row tile where four scattered 4-byte ones stood
        const int head0 = n0 / 40;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float x[2], y[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                x[h] = pl[h][mt];
                y[h] = pr[h][mt];
                x[h] = x[h] + __shfl_xor(x[h], 16);
                y[h] = y[h] + __shfl_xor(y[h], 16);
                x[h] = x[h] + __shfl_xor(x[h], 32);
                y[h] = y[h] + __shfl_xor(y[h], 32);
            }
            const int m = m0 + wave * 16 * MT + mt * 16 + fr;
            if (m >= M || head0 * 40 >= n || fq > 1) continue;
            float *dst = a12 + (size_t)m * 32 + (fq ? 16 : 0) + head0;
            const float v0 = fq ? y[0] : x[0], v1 = fq ? y[1] : x[1];
            if ((head0 + 1) * 40 < n) *reinterpret_cast<float2 *>(dst) = make_float2(v0, v1);
            else dst[0] = v0;
        }
    }
    }      // tiles of this workgroup
    SB_STAMP(2);
}

// ---- one wave per 16 x 16 tile (small batches, narrow outputs): operands streamed from global memory --------------------------
constexpr int SBS_DEPTH = 4;

struct SbFrag {
    f32x4 a0, a1;
    bf16x8 w[3];
};

__device__ __forceinline__ void sb_load(SbFrag &f, const float *pa, const unsigned short *pw, size_t w_plane, int ko) {
    f.a0 = *reinterpret_cast<const f32x4 *>(pa + ko);
    f.a1 = *reinterpret_cast<const f32x4 *>(pa + ko + 4);
#pragma unroll
    for (int p = 0; p < 3; ++p) f.w[p] = *reinterpret_cast<const bf16x8 *>(pw + p * w_plane + ko);
}

template <bool LEAKY, bool F64>
__global__ __launch_bounds__(256) void k_linear_sb_skinny(const float *__restrict__ A, int lda, const unsigned short *__restrict__ W3,
                                                           size_t w_plane, int ldw, const float *__restrict__ bias,
                                                           float *__restrict__ C, int ldc, int m_cap,
                                                           const int32_t *__restrict__ d_m, int n, int k_pad, float slope, int nt16, int fl) {
    int M = m_cap;
    if (d_m) {
        const int dm = *d_m;
        M = dm < m_cap ? dm : m_cap;
    }
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tm = gw / nt16, tn = gw - tm * nt16;
    if (tm * 16 >= M) return;                                // no barriers below: waves may leave early
    const int fq = lane >> 4, fr = lane & 15;
    int grow = tm * 16 + fr;
    grow = grow < M ? grow : M - 1;
    const float *pa = A + (size_t)grow * lda + 8 * fq;       // stage kt: k = 32 kt + 8 fq + {0..7}
    const unsigned short *pw = W3 + (size_t)(tn * 16 + fr) * ldw + 8 * fq;
    const int nk = k_pad / GEMM_BK;
    SbFrag fr_[SBS_DEPTH];
#pragma unroll
    for (int d = 0; d < SBS_DEPTH; ++d) sb_load(fr_[d], pa, pw, w_plane, (d < nk ? d : nk - 1) * GEMM_BK);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_odd = {0.f, 0.f, 0.f, 0.f};
    double run[4] = {0.0, 0.0, 0.0, 0.0};
    auto stage = [&](const SbFrag &f, int kt) {
        bf16x8 ap[3];
        split8(f.a0, f.a1, ap[0], ap[1], ap[2]);
        if (F64 || !(kt & 1)) SB_STAGE(acc, ap, f.w);
        else SB_STAGE(acc_odd, ap, f.w);             // without f64 sums: even / odd stages in separate chains, as k_linear_sb
        if (F64 && ((kt & 1) || kt == nk - 1 || fl == 1)) {       // fl = K stages per f64 flush (k_linear_sb: FL)
#pragma unroll
            for (int i = 0; i < 4; ++i) run[i] += (double)acc[i];
            acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    int kt0 = 0;
    for (; kt0 + SBS_DEPTH <= nk; kt0 += SBS_DEPTH) {
#pragma unroll
        for (int d = 0; d < SBS_DEPTH; ++d) {
            stage(fr_[d], kt0 + d);
            int kn = kt0 + d + SBS_DEPTH;                    // refill the slot in place (clamped: no branch)
            kn = (kn < nk ? kn : nk - 1) * GEMM_BK;
            sb_load(fr_[d], pa, pw, w_plane, kn);
        }
    }
#pragma unroll
    for (int d = 0; d < SBS_DEPTH; ++d)
        if (kt0 + d < nk) stage(fr_[d], kt0 + d);
    const int m = tm * 16 + fr;
    if (m >= M) return;
    const int nb = tn * 16 + fq * 4;
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + nb);
    float *dst = C + (size_t)m * ldc + nb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = F64 ? (float)(run[i] + (double)bv[i]) : (acc[i] + acc_odd[i]) + bv[i];
        if (LEAKY) v = v > 0.f ? v : v * slope;
        if (nb + i < n) dst[i] = v;
    }
}

// The stage PAIRS of a tile (= the units between two f64 flushes) dealt round-robin to the KS waves of a workgroup; fp32 pair
// results parked in LDS, then every element summed over the pairs in pair order in f64: the additions of k_linear_sb in the
// same order (and exact anyway: fp32 terms in an f64 sum).
// RT = 2: a workgroup takes TWO 16-row tiles with one fetch of the weight fragments (17 ... 128 rows: eight frames per call, the MLP of
// a few dozen persons): every row tile used to be a workgroup of its own that streamed the layer's planes again.
template <bool LEAKY, int KS, int FL, int RT = 1>
__global__ __launch_bounds__(64 * KS) void k_linear_sb_ks(const float *__restrict__ A, int lda, const unsigned short *__restrict__ W3,
                                                          size_t w_plane, int ldw, const float *__restrict__ bias,
                                                          float *__restrict__ C, int ldc, int m_cap,
                                                          const int32_t *__restrict__ d_m, int n, int k_pad, float slope, int nt16,
                                                          DecodeEpi dec) {
    constexpr int fl = FL;                                   // K stages per f64 flush (a template argument: with one stage per unit the second fragment is not even requested)
    extern __shared__ __attribute__((aligned(16))) float s_part[];       // [pairs][RT][64 lanes][4]
    int M = m_cap;
    if (d_m) {
        const int dm = *d_m;
        M = dm < m_cap ? dm : m_cap;
    }
    const int tg = blockIdx.x / nt16, tn = blockIdx.x - tg * nt16;
    const int tm = tg * RT;                                  // first row tile of the workgroup
    if (tm * 16 >= M) return;                                // whole workgroup leaves: no barrier reached
    const bool two = RT == 2 && (tm + 1) * 16 < M;           // (uniform) the second row tile holds rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fq = lane >> 4, fr = lane & 15;
    int grow = tm * 16 + fr;
    grow = grow < M ? grow : M - 1;
    const float *pa = A + (size_t)grow * lda + 8 * fq;
    int grow2 = (tm + 1) * 16 + fr;
    grow2 = grow2 < M ? grow2 : M - 1;
    const float *pa2 = A + (size_t)grow2 * lda + 8 * fq;
    const unsigned short *pw = W3 + (size_t)(tn * 16 + fr) * ldw + 8 * fq;
    // (fl = K stages per f64 flush: the units are stage pairs, or single stages in the flush-per-stage mode)
    const int nk = k_pad / GEMM_BK, npair = (nk + fl - 1) / fl;
    for (int j = wave; j < npair; j += KS) {
        SbFrag f0, f1;
        f32x4 b00 = {0.f, 0.f, 0.f, 0.f}, b01 = b00, b10 = b00, b11 = b00;         // the second row tile's activation fragments
        const int k0 = fl * j, k1 = fl * j + 1 < nk ? fl * j + 1 : nk - 1;
        sb_load(f0, pa, pw, w_plane, k0 * GEMM_BK);
        if (FL == 2) sb_load(f1, pa, pw, w_plane, k1 * GEMM_BK);
        if (RT == 2 && two) {
            b00 = *reinterpret_cast<const f32x4 *>(pa2 + k0 * GEMM_BK);
            b01 = *reinterpret_cast<const f32x4 *>(pa2 + k0 * GEMM_BK + 4);
            if (FL == 2) {
                b10 = *reinterpret_cast<const f32x4 *>(pa2 + k1 * GEMM_BK);
                b11 = *reinterpret_cast<const f32x4 *>(pa2 + k1 * GEMM_BK + 4);
            }
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bf16x8 ap[3];
        split8(f0.a0, f0.a1, ap[0], ap[1], ap[2]);
        SB_STAGE(acc, ap, f0.w);
        if (fl == 2 && 2 * j + 1 < nk) {
            split8(f1.a0, f1.a1, ap[0], ap[1], ap[2]);
            SB_STAGE(acc, ap, f1.w);
        }
        *reinterpret_cast<f32x4 *>(&s_part[((j * RT) * 64 + lane) * 4]) = acc;
        if (RT == 2 && two) {
            f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
            split8(b00, b01, ap[0], ap[1], ap[2]);
            SB_STAGE(acc2, ap, f0.w);
            if (fl == 2 && 2 * j + 1 < nk) {
                split8(b10, b11, ap[0], ap[1], ap[2]);
                SB_STAGE(acc2, ap, f1.w);
            }
            *reinterpret_cast<f32x4 *>(&s_part[((j * RT + 1) * 64 + lane) * 4]) = acc2;
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    const int rt = t >> 8, tt = t & 255;                     // row tile of the workgroup, element of its 16 x 16 tile
    if (rt >= RT || (rt == 1 && !two)) return;
    double run = 0.0;
    for (int j = 0; j < npair; ++j) run += (double)s_part[(j * RT + rt) * 256 + tt];  // pair order, as the tile kernel
    const int el = tt >> 2, i = tt & 3;                      // element (lane el, component i) of the MFMA tile
    const int m = (tm + rt) * 16 + (el & 15), nb = tn * 16 + (el >> 4) * 4 + i;
    if (m >= M || nb >= n) return;
    float v = (float)(run + (double)bias[nb]);
    if (LEAKY) v = v > 0.f ? v : v * slope;
    C[(size_t)m * ldc + nb] = v;
    if (dec.poses) {
        // k_decode from here: the frame of row m by bisection of the persons' prefix, then y * scale into the (frame, person) slot
        int lo = 0, hi = dec.n_frames;                       // person_off[lo] <= m < person_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (dec.person_off[mid] <= m) lo = mid;
            else hi = mid;
        }
        const int pp = m - dec.person_off[lo];
        if (pp < dec.pcap && nb < dec.n_out) dec.poses[((size_t)lo * dec.pcap + pp) * dec.n_out + nb] = v * dec.scale;
    }
}

// fp32 weights [rows][ld] -> three bf16 planes [3][rows][ld]
__global__ void k_split_planes(const float *__restrict__ w, size_t count, unsigned short *__restrict__ planes) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= count) return;
    const float a = w[i], b = i + 1 < count ? w[i + 1] : 0.f;
    const unsigned u0 = pack2_bf16(a, b);
    const float ra = a - __uint_as_float(u0 << 16), rb = b - __uint_as_float(u0 & 0xFFFF0000u);
    const unsigned u1 = pack2_bf16(ra, rb);
    const float sa = ra - __uint_as_float(u1 << 16), sb = rb - __uint_as_float(u1 & 0xFFFF0000u);
    const unsigned u2 = pack2_bf16(sa, sb);
    const unsigned us[3] = {u0, u1, u2};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        planes[p * count + i] = (unsigned short)(us[p] & 0xFFFFu);
        if (i + 1 < count) planes[p * count + i + 1] = (unsigned short)(us[p] >> 16);
    }
}

}  // namespace sb
using namespace sb;

#ifdef MPE_SB_CLOCK
}  // namespace mpe
extern "C" int mpe_debug_sb_stamps(unsigned long long *out, int clear) {      // diagnostic builds only; not part of include/mpe.h: [16][256][4]
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mpe::sb::g_sb_stamp), sizeof(mpe::sb::g_sb_stamp)) != hipSuccess) return -1;
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(mpe::sb::g_sb_stamp)) != hipSuccess || hipMemset(p, 0, sizeof(mpe::sb::g_sb_stamp)) != hipSuccess) return -1;
    }
    return 0;
}
namespace mpe {
using namespace sb;
#endif

hipError_t launch_split_planes(hipStream_t s, const float *w, size_t count, unsigned short *planes) {
    if (count == 0) return hipSuccess;
    const size_t pairs = (count + 1) / 2;
    hipLaunchKernelGGL(k_split_planes, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, w, count, planes);
    return hipGetLastError();
}

int device_cu_count() {
    static int cu_of_device[64];                    // (asked once per device: zero-initialised, benign if two threads ask at once)
    int dev = 0;
    (void)hipGetDevice(&dev);
    int &cu = cu_of_device[dev & 63];
    if (cu == 0) {
        int c = 256;
        (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        cu = c > 0 ? c : 256;
    }
    return cu;
}

// Same dispatch rules as launch_linear's f64-sum branch (gemm.hip): K split over eight waves / one wave per 16 x 16 tile for
// small batches and for outputs of at most four 16-wide tiles at any batch size, the tile kernel otherwise.
static int sb_skinny_waves() {
    static const int v = getenv("MPE_SKINNY_WAVES") ? atoi(getenv("MPE_SKINNY_WAVES")) : 1024;
    return v;
}
static int sb_narrow_on() {
    static const int v = getenv("MPE_GEMM_NARROW") ? atoi(getenv("MPE_GEMM_NARROW")) : 1;
    return v;
}

// Row tiles (of CAPACITY: the rows a batch really holds are a device-side count, typically a third of it) up to which the K-split
// kernel takes an f64-sum launch instead of the tile kernel.  Measured per row tile that holds rows against the tile kernel's time
// for up to 256 rows: 7 against 102 us on the 3072 x 3072 layer, 4.8 against 70 on 2048 x 2048, 1.5 against 37 on 1024 x 1024 -- the
// tile kernel puts N / 64 workgroups on the chip, each streaming its weights alone.  Thresholds from the sweep of frames per call
// (5 x 4 frames, capacity 10 persons per frame; us per call before / after): 33 frames 835 / 626, 40: ~850 / 653, 48: 858 / 708,
// 64: 892 / 795, 96: 986 / 981 (profiles/r06_frames_per_call.txt).
static int few_rows_tiles(int nt16) { return nt16 >= 128 ? 32 : 48; }

// Does launch_linear_sb16 take the tile kernel for this shape?  (Only the tile kernel stores fp16 rows: callers that want
// `out_half` ask here first and keep the launch on the fp32 MFMA otherwise.)
bool linear_sb16_uses_tile_kernel(int m_cap, int n, bool f64) {
    const int nt16 = (n + 15) / 16;
    const long waves16 = (long)((m_cap + 15) / 16) * nt16;
    const bool narrow = sb_narrow_on() && nt16 <= (f64 ? 4 : 1);
    const bool few_rows = f64 && sb_skinny_waves() > 0 && (m_cap + 15) / 16 <= few_rows_tiles(nt16) && nt16 > 4;      // (launch_linear_sb16)
    return !(waves16 <= sb_skinny_waves() || narrow || few_rows);
}

hipError_t launch_linear_sb16(hipStream_t s, const float *A, int lda, const unsigned short *W3, size_t w_plane, int ldw,
                              const float *bias, float *C, int ldc, int m_cap, const int32_t *d_m, int n, int k_pad, bool leaky,
                              float slope, bool f64, const AttnCoef *coef, bool *coef_done, bool out_half, int flush_stages,
                              const DecodeEpi *dec, bool *dec_done) {
    if (coef_done) *coef_done = false;
    if (dec_done) *dec_done = false;
    if (m_cap <= 0 || n <= 0) return hipSuccess;
    if (flush_stages != 1 && flush_stages != 2) return hipErrorInvalidValue;
    const int skinny_waves = sb_skinny_waves();
    const int narrow_on = sb_narrow_on();
    const int nt16 = (n + 15) / 16, nk = k_pad / GEMM_BK;
    const long waves16 = (long)((m_cap + 15) / 16) * nt16;
    const bool narrow = narrow_on && nt16 <= (f64 ? 4 : 1);
    // fp16 result rows (fc2 launches of the fp16-attention mode) exist in the tile kernel's fp32-chain forms only: refuse the
    // launch rather than store fp32 rows into a buffer the caller strides in halves
    if (out_half && (f64 || leaky || waves16 <= skinny_waves || narrow)) return hipErrorInvalidValue;
    // A few row tiles of a WIDE layer (the MLP of 9 ... 32 frames: capacity 90 ... 320 rows, a fraction of them persons): the tile kernel
    // would put 48 workgroups on the chip, each streaming its 64 features' weights alone (100 us for the 3072 x 3072 layer whatever the
    // rows: 16 frames spent 315 of their 673 us there); the K-split kernel streams the layer once per two row tiles THAT HOLD ROWS
    // (workgroups of empty tiles leave at once), 25 us at 64 rows.  At capacity it can be up to 2x behind the tile kernel, at the
    // usual third of it 1.3-2x ahead.  MPE_SKINNY_WAVES=0 still means the tile kernel everywhere.  (few_rows_tiles: the break-even per width.)
    const bool few_rows = f64 && skinny_waves > 0 && (m_cap + 15) / 16 <= few_rows_tiles(nt16) && nt16 > 4;
    if (f64 && (waves16 <= skinny_waves || narrow || few_rows) && nk <= 128 * flush_stages && nk >= 8) {
        // two row tiles per workgroup, the weight fragments fetched once for both (LDS: 2 KB per unit then) -- where two row tiles as
        // workgroups of their own would be more workgroups than the chip has CUs (eight frames: 32 rows x 3072 features = 384), so
        // that the second tile would wait for a CU anyway; below that the tiles run side by side and share the weights through the
        // L2, and a workgroup that takes both only serialises them (measured at eight 5 x 4 frames: 14.9 / 30.0 us as one tile per
        // workgroup, 12.5 / 23.4 as two on the 3072-wide layers; 17.3 against 22.6 on the 2048-wide ones).  The rows a batch really
        // holds are a device-side count: the rule looks at the layer's width only.
        const int row_tiles = (m_cap + 15) / 16;
        const bool rt2 = row_tiles > 1 && 2 * nt16 > device_cu_count() && (size_t)((nk + flush_stages - 1) / flush_stages) * 2048 <= 128 * 1024;
        const size_t shm = (size_t)((nk + flush_stages - 1) / flush_stages) * (rt2 ? 2048 : 1024);
        static PerDeviceFlag attr_done;
        if (!attr_done.test()) {
            hipError_t e = hipSuccess;
            const void *fns[] = {reinterpret_cast<const void *>(k_linear_sb_ks<true, 8, 2>), reinterpret_cast<const void *>(k_linear_sb_ks<false, 8, 2>),
                                 reinterpret_cast<const void *>(k_linear_sb_ks<true, 8, 1>), reinterpret_cast<const void *>(k_linear_sb_ks<false, 8, 1>),
                                 reinterpret_cast<const void *>(k_linear_sb_ks<true, 8, 2, 2>), reinterpret_cast<const void *>(k_linear_sb_ks<false, 8, 2, 2>),
                                 reinterpret_cast<const void *>(k_linear_sb_ks<true, 8, 1, 2>), reinterpret_cast<const void *>(k_linear_sb_ks<false, 8, 1, 2>)};
            for (const void *fn : fns)
                if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return e;
            attr_done.set();
        }
        DecodeEpi de{};
        if (dec) de = *dec;
        const unsigned grid = (unsigned)((rt2 ? (row_tiles + 1) / 2 : row_tiles) * nt16);
#define MPE_KS(L_, F_)                                                                                                                          \
    do {                                                                                                                                        \
        if (rt2) hipLaunchKernelGGL((k_linear_sb_ks<L_, 8, F_, 2>), dim3(grid), dim3(512), shm, s, A, lda, W3, w_plane, ldw, bias, C, ldc, m_cap, d_m, n, k_pad, slope, nt16, de); \
        else hipLaunchKernelGGL((k_linear_sb_ks<L_, 8, F_, 1>), dim3(grid), dim3(512), shm, s, A, lda, W3, w_plane, ldw, bias, C, ldc, m_cap, d_m, n, k_pad, slope, nt16, de);     \
    } while (0)
        if (leaky && flush_stages == 2) MPE_KS(true, 2);
        else if (leaky) MPE_KS(true, 1);
        else if (flush_stages == 2) MPE_KS(false, 2);
        else MPE_KS(false, 1);
#undef MPE_KS
        if (dec && dec_done) *dec_done = true;
        return hipGetLastError();
    }
    if (waves16 <= skinny_waves || narrow) {
        const dim3 grid((unsigned)((waves16 + 3) / 4)), block(256);
#define MPE_SBS(L_, F_)                                                                                                   \
    hipLaunchKernelGGL((k_linear_sb_skinny<L_, F_>), grid, block, 0, s, A, lda, W3, w_plane, ldw, bias, C, ldc, m_cap, d_m, n, k_pad, \
                       slope, nt16, flush_stages)
        if (leaky && f64) MPE_SBS(true, true);
        else if (leaky) MPE_SBS(true, false);
        else if (f64) MPE_SBS(false, true);
        else MPE_SBS(false, false);
#undef MPE_SBS
        return hipGetLastError();
    }
    static PerDeviceFlag lds_attr;
    if (!lds_attr.test()) {
        hipError_t e = hipSuccess;
        const void *fns[] = {reinterpret_cast<const void *>(k_linear_sb<true, 4, true>), reinterpret_cast<const void *>(k_linear_sb<false, 4, true>),
                             reinterpret_cast<const void *>(k_linear_sb<true, 4, true, false, 1>), reinterpret_cast<const void *>(k_linear_sb<false, 4, true, false, 1>),
                             reinterpret_cast<const void *>(k_linear_sb<true, 5, false>), reinterpret_cast<const void *>(k_linear_sb<false, 5, false>),
                             reinterpret_cast<const void *>(k_linear_sb<false, 5, false, true>)};
        for (const void *fn : fns)
            if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        lds_attr.set();
    }
    const int n_major = (size_t)n * k_pad * sizeof(float) > (size_t)(2u << 20) ? 1 : 0;
    // persistent workgroups: one per CU walking its tiles (grid = the CU count rounded down to a multiple of eight, or the tile count)
    const int cu = device_cu_count();
    const int n_cu = cu >= 8 ? cu / 8 * 8 : 8;
    const bool with_coef = coef && coef->out_dim == 40 && n == coef->heads * 40 && !leaky && !f64;
    // 64-wide feature tiles with f64 sums (with 80 the running sums of the wider wave tile do not fit the 168 registers that three
    // waves per SIMD leave; the MLP's layers balance with 64 anyway), 80-wide without (two 40-wide attention heads per tile)
    const int ntt = f64 ? 4 : 5;
    const int ntn = (n + ntt * 16 - 1) / (ntt * 16);
    const int tiles = ((m_cap + 255) / 256) * ntn;
    const dim3 grid((unsigned)(tiles < n_cu ? tiles : n_cu)), block(64 * (SB_MW + SB_NL));
    const size_t shm = (size_t)SB_RING * sb_stage_bytes(ntt);
#define MPE_SB(L_, N_, F_, FL_) \
    hipLaunchKernelGGL((k_linear_sb<L_, N_, F_, false, FL_>), grid, block, shm, s, A, lda, W3, w_plane, ldw, bias, C, ldc, m_cap, d_m, n, k_pad, slope, ntn, n_major, \
                       nullptr, nullptr, nullptr, out_half ? 1 : 0)
    if (f64 && flush_stages == 1) {
        if (leaky) MPE_SB(true, 4, true, 1);
        else MPE_SB(false, 4, true, 1);
    } else if (f64) {
        if (leaky) MPE_SB(true, 4, true, 2);
        else MPE_SB(false, 4, true, 2);
    } else if (with_coef) {
        hipLaunchKernelGGL((k_linear_sb<false, 5, false, true>), grid, block, shm, s, A, lda, W3, w_plane, ldw, bias, C, ldc, m_cap, d_m, n, k_pad,
                           slope, ntn, n_major, coef->attn_l, coef->attn_r, coef->a12, out_half ? 1 : 0);
        if (coef_done) *coef_done = true;
    } else {
        if (leaky) MPE_SB(true, 5, false, 2);
        else MPE_SB(false, 5, false, 2);
    }
#undef MPE_SB
    return hipGetLastError();
}

}  // namespace mpe
