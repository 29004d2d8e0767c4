// admm_solve.hip -- batched horizon-QP build + OSQP-style ADMM solve, one wavefront per MPC instance.
//
// Replaces, per instance, the reference chain
//   _buildMatEqConst (CTRL:477-529 / PLAN:434-486), _buildMatCost (CTRL:382-473 / PLAN:145-169),
//   _buildMatIneqConst (CTRL:329-378 / PLAN:173-181) and the OSQP setup()+solve() call
//   (CTRL:273-325 / PLAN:200-215; algorithm = OSQP 0.6.x defaults, see oracle/osqp_ref.c header).
//
// Design (CDNA4): the whole solve of one instance -- Ruiz equilibration, KKT factorisation, every ADMM
// iteration, adaptive-rho refactorisations, termination / infeasibility tests and the polish step --
// runs inside ONE 64-lane wavefront with all state resident in LDS, so HBM sees only the LPV blocks
// going in and the trajectory coming out.  The QP is never materialised as sparse matrices: the
// decision vector is kept stage-interleaved, w_k = (x_k, u_k), which makes
//     K = P + sigma I + A' diag(rho) A
// block tridiagonal with 8x8 blocks.  K is factored as a block L D L' with explicit inverses of the
// pivots (S_k^-1) so that a solve is 3 tile mat-vecs per stage.  Tile element [i][j] lives in lane
// 8*i+j; reductions across i or j are wave shuffles.
//
// Two instantiations per model: Solver<NX, NT> with a compile-time horizon NT keeps the factor tiles
// (S_k^-1, L_k: 2 doubles per stage per lane) in REGISTERS and fully unrolls the two KKT sweeps;
// Solver<NX, 0> takes the horizon at run time and keeps those tiles in LDS.
// LDS per instance (doubles): scaled [A|B] tiles [NS][72] and 19 vectors [NS][8] (NS = N+1, everything
// padded to 8 per stage so that index = 8*stage + component): 37.7 KB at N = 20, i.e. 4 instances per
// CU (one wavefront per SIMD), which is exactly BASELINE configs[1] (1024 instances) in one residency.
#include <atomic>
#include <type_traits>

#include "lpvmpc_device.hpp"
#include "wave_ops.hpp"

namespace lpvmpc {

#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS != 5
#define STAMP(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp[slot] += t_ - tlast; tlast = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif
#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS == 5     // the parts of a whole solve: set-up, equilibration, first factorisation, iterations, checks + rho updates, polish factorisation, polish, output (tools/gpu_stamps_solve.py)
#define STAMP5(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp[slot] += t_ - tlast; tlast = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP5(slot) do { } while (0)
#endif
#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS == 4
#define STAMP4(slot) do { asm volatile("s_waitcnt lgkmcnt(0)"); STAMP(slot); } while (0)      // (tail kernel, inside the phases: everything issued so far has arrived)
#else
#define STAMP4(slot) do { } while (0)
#endif

template <int NX, int NT, int NW, bool MF = false, bool GS = false, bool TAIL = false>
struct Solver {
#ifdef LPVMPC_STAMPS
    unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#endif
    static constexpr int NB = NX + 2;
    static constexpr bool kCtrl = (NX == 6);
    // TAIL: the whole-CU kernel that finishes parked stragglers ("tail kernel").  NW wavefronts share ONE
    // instance: the KKT solve of an ADMM iteration is a dense product with K^-1, which is held in the registers of all of them
    // (dense_apply); everything else -- the factorisation K^-1 is built from, the polish solves -- runs on wavefront 0 with the
    // LDS-tile code of the run-time-horizon kernel.  It only ever resumes pool entries: its LDS begins with the image of the
    // kernel that parked them (the compile-time-horizon layout).
    static constexpr bool kTail = TAIL;
#ifndef LPVMPC_BLOCK_LAUNDER
#define LPVMPC_BLOCK_LAUNDER 2     // 0: off; 2: the lane indices of the factorisation; 1: the thread index too (un-hoists the iteration's own addresses: loses)
#endif
#ifndef LPVMPC_UNI_SCALARS
#define LPVMPC_UNI_SCALARS 1
#endif
    static constexpr bool kUniScalars = TAIL || (LPVMPC_UNI_SCALARS != 0);   // c, rho and their kin are the same in every lane: keep them in scalar registers
    static constexpr bool kFixN = (NT > 0);         // horizon known at compile time
    static constexpr bool kReg = kFixN && !TAIL;    // factor tiles in registers
    static constexpr bool kTwo = (NW == 2);         // two wavefronts per instance: two-sided ("twisted") elimination
    static constexpr bool kMf = MF;                 // the two sweeps run on the matrix cores (v_mfma_f64_4x4x4_4b_f64), see mf_forward
    // FOUR wavefronts per instance (round 4; planner N = 40 / 30): the same two chains, each RELAYED over two wavefronts -- the
    // "outer" wavefront of a chain (0: top, 1: bottom) holds the operand tiles of chain positions 0 .. kHO-1, the "inner" one
    // (2 / 3) those of kHO .. kMid-1 -- so a wavefront keeps half of the factor (<= 256 registers: two wavefronts per SIMD), the
    // element phases run on 256 threads, and the half of them that does not touch the stages a running sweep works on is done BESIDE
    // that sweep by the two wavefronts that wait for the relay (iterate4).  Same arithmetic, step for step, as the two-wavefront kernel.
    static constexpr bool kFour = (NW == 4) && !TAIL;
    static_assert(!MF || ((NW == 2 || NW == 4) && NT > 0), "the MFMA sweeps are written for the two- / four-wavefront compile-time-horizon kernels");
    static_assert(NW != 4 || TAIL || (MF && !GS && NT >= 20 && NT % 2 == 0), "four wavefronts: MFMA sweeps, even compile-time horizon >= 20");
    static_assert(TAIL || NW == 1 || NW == 4 || (NW == 2 && NT >= 16 && NT % 2 == 0), "two wavefronts need an even compile-time horizon >= 16");
    static_assert(!TAIL || (NT > 0 && NW == 8 && !MF && !GS), "the tail kernel: compile-time horizon, eight wavefronts, no MFMA sweeps");
    static constexpr bool kLastOdd = ((NT / 2 - 1) & 1) != 0;   // parity of the last chain position (both chains have NT/2 stages)
    static constexpr int kMid = NT / 2;             // NW == 2: wave 0 eliminates stages 0..kMid-1 upwards, wave 1 stages
    static constexpr int kP0 = kMid, kP1 = NT - kMid;   //         NT..kMid+1 downwards; stage kMid joins the two chains
    static constexpr int kRS = (kReg && !MF) ? (kTwo ? (kP1 > kP0 ? kP1 : kP0) : NT + 1) : 1;
    // Even chain length: the pivot products v_p = S_p^-1 y_p never leave the registers.  The forward step that forms v_p and the backward
    // step that takes it as its C operand are then of the same type (same pairing of the four blocks), so the RAW MFMA result -- every
    // block its own partial sum, not closed -- is the C operand of every block and the backward step's closing add completes both
    // sums at once: no closing move / add, no store and no load per step; 2 registers per chain position.
    // Odd chain length (N = 30): the types would alternate out of step, so the MIDDLE stage is solved as a type B step there (kFlip: its
    // right-hand side taken in layout B, x_m delivered in layout A) and every backward step changes type with it.
    static constexpr bool kFlip = MF && (NW == 2 || NW == 4) && (kMid % 2 != 0);
    static constexpr bool kRawV = MF && (NW == 2 || NW == 4);
    static constexpr int kVQ = kRawV ? kMid : 3;
    static constexpr int kHO = (kMid + 1) / 2;      // NW == 4: operand tiles [0, kHO) on the outer wavefront of a chain, [kHO, kMid) on the inner one
    static constexpr int kMP = MF ? (kFour ? kHO : kMid) : 1;       // chain positions per wavefront in the MFMA sweeps
    // NW == 4: the element phases of an ADMM iteration (right-hand side, update) are cut into set 2 -- kS2n stages of each chain's inner
    // half, from kS2a upwards (top chain, wavefront 2) and from kS2b upwards (bottom chain, wavefront 3), one element per lane: nobody but
    // the owning wavefront's forward steps reads their right-hand side, and their x~ and [A|B] x~ are that wavefront's own backward
    // results (chain positions kHO + 2 .. kMid - 1) -- and set 1, the rest, one element per thread of the workgroup
    static constexpr int kS2n = kMid - kHO - 2 < 8 ? kMid - kHO - 2 : 8;
    static constexpr int kS2a = kHO + 2, kS2b = NT - kHO - 1 - kS2n;
    static_assert(!kFour || (kS2n >= 1 && (NT + 1) * 8 - 16 * kS2n <= 256 && kS2a + kS2n <= kMid && kS2b > kMid && kS2b + kS2n - 1 == NT - kHO - 2), "four wavefronts: element sets do not fit");
    // RED (doubles): [0, 80) as in the two-wavefront kernels; NW == 4 adds [80, 112) residual maxima of four wavefronts, [112, 144) four
    // reduction slots {sum x 4, max x 4}, [144, 160) the relay of the forward sweeps (y at chain position kHO, top / bottom)
    static constexpr int kRedSize = kFour ? 160 : 80;
    static constexpr bool kFourSmall = kFour && 2 * (NT + 1) * 8 < 448;     // N = 20 on four wavefronts
    static constexpr int kStride = 64 * NW;

    const DevCfg &cfg;
    const int N, NS;
    int tid, wv, lane, ti, tj;      // (not const: the tail kernel launders them behind its loop, see launder_ids)
    const int delay;      // controller: stages 0 .. delay-1 carry a pinned-steering equality row in box slot 6 (CTRL:518-527)
    // tiles: scaled [A|B] always in LDS; S^-1 and L in LDS (run-time horizon) or registers (kReg)
    double *tS, *tL, *tA;
    double rS[kRS], rL[kRS];
    double rSm, rLt, rLb;    // NW == 2: pivot inverse of the middle stage and its two link tiles (wave 0; rLb also wave 1)
    // MFMA sweeps: one A-operand register per 8x8 tile and chain step (all four 4x4 blocks of the instruction carry the
    // tile, see mf_forward): fC / bC = the chain tiles (-L_p on the way up, -L_{p+1}' on the way down), fV = the pivot
    // inverse S_{p-1}^-1 and bA = [A|B] of the operand stage (second, independent product of the same step); mS = S_m^-1,
    // tT = [A|B]_0 (wave 0's last product).
    double fC[kMP], fV[kMP], bC[kMP], bA[kMP], mS, tT;
    int gA, gB;                             // gather offsets into a row-major 8x8 tile for the two step types (see mf_forward)
    int li, lj, tlane;                      // "D form" of an 8x8 matrix (see mm8): row / column held by this lane, lane of the transposed element
    // variable-space vectors
    double *X, *Qv, *D, *XT, *DX, *VT, *AT;
    // dynamics rows / box rows (ZT*: scratch rows; DY*: delta_y, or the active-set flags during polish)
    double *Zd, *Yd, *Ed, *ZTd, *DYd;
    double *Zb, *Yb, *Eb, *ZTb, *DYb, *Lo, *Hi;
    double *beq;   // [8] scaled x0 (bounds of the stage-0 dynamics rows)
    double *FX;    // [448] kFourSmall only
    double *SINK;  // [64 + 8 NS] all zero in the MFMA kernels (their sweeps load the C operand of non-owner blocks from it); the DPP two-wavefront
                   // factorisation hands a tile from wave 0 to wave 1 through it; part of the parked image
    double *Pm;    // [64] unscaled stage Hessian block 2*[Q 0; 0 R + 2 diag(dR)] (LDS copy of the weights)
    double *dRl;   // [8]  dR[0..1]
    double *RED;   // [80] per-wave partial results of block-wide reductions / chain hand-over (NW == 2)
    // tail kernel only (behind the parked image): partial results of the block-wide reductions [96], two scratch tiles per
    // wavefront for the block sweeps that build K^-1 [NW][128], and the staging area those sweeps deliver a round of kDenseRound
    // block columns of K^-1 through [kDenseRound][NS][64]
    double *RT, *WS, *STG;
    // K^-1 in registers, columns 8 c + s (c = 0 .. NS-1) of a few rows per thread (g, s = tid & 7), UNEVENLY (round 5): the threads of
    // the first kActW wavefronts, which also own the variable elements, hold two rows (2g, 2g+1: g = tid >> 3), those of the next four
    // wavefronts four rows (kRowsA + 4g ..: g = (tid - 64 kActW) >> 3); the last wavefront holds none (it runs the termination
    // checks).  Every thread has the same 4 NS doubles `dm`: four rows of K^-1 on the heavy wavefronts; on the light ones two rows, and
    // behind them the element's row of T (24), its box rows' state and constants (7 each), x, G, S, c0 (see tail_fused) -- one
    // register array, so that what the light wavefronts keep for the element phase costs the heavy ones nothing
    static constexpr int kDC = TAIL ? NT + 1 : 1;
    static constexpr int kRhsPitch = ((NT + 1) + 1) & ~1;            // tail kernel: doubles per slot of the slot-major right-hand side (even: 16-byte loads)
    static_assert(!TAIL || 8 * (((NT + 1) + 1) & ~1) <= 2 * (NT + 1) * 8, "the slot-major right-hand side must fit VT + AT");
    static constexpr int kDM = TAIL ? 4 * (NT + 1) : 1;
    static constexpr int kRowsA = 2 * 8 * (((NT + 1) * 8 + 63) / 64);   // rows held by the light wavefronts (two per 8-lane group)
    static constexpr int kGroupsB = ((NT + 1) * 8 - kRowsA + 3) / 4;   // 8-lane groups of the heavy wavefronts that hold rows
    static constexpr int kDenseRound = 7;                             // block columns per staging round
    static constexpr int kActW = ((NT + 1) * 8 + 63) / 64;            // wavefronts that own elements in the element loops
    static_assert(!TAIL || (kActW + (kGroupsB + 7) / 8 <= NW - 1 && kActW <= 4 && 2 * kActW < NW), "tail kernel: K^-1 rows / element loops do not fit the workgroup (the last wavefront holds no K^-1 rows: it runs the termination checks)");
    static_assert(!TAIL || ((NT + 1) * 8 * 24 + 64 * (NT + 1) + 8 * 64 <= kDenseRound * (NT + 1) * 64), "tail kernel: T rows and the checker's scratch do not fit the staging area");
    double dm[kDM];
    static constexpr int kTR = 2 * (NT + 1), kBX = kTR + 24, kXS = kBX + 14;      // light wavefronts: T row, box rows (z, y, lo, hi, sb, w, 1/w) x 2, (x, G, S, c0)
    static_assert(!TAIL || kXS + 4 <= kDM, "tail kernel: the element registers do not fit behind the light wavefronts' rows of K^-1");
    // ---- tail kernel, round 5: the ADMM iteration in TWO phases (dense product, fused element phase: tail_fused) ----
    // One variable element e = tid per thread of the first kActW wavefronts, its state in registers: x, G = A_d'(rho_eq z_d - y_d) - q
    // (the dynamics rows' part of the right-hand side, carried in variable space), S = sum of x~ since the last snapshot (y_d is
    // rebuilt from it where it is needed: at the termination checks), c0 = (A_d' b)[e]; and the one or two box rows that act on the
    // variable (controller: vx, delta, a have two each; planner: one per variable): z, y and the row's constants.
    static constexpr int kBR = kCtrl ? 2 : 1;
    double *TT;    // [8 NS][24] rows of T = A_d' A_d: the blocks that multiply x~ of stage k-1, k, k+1 (in STG, rebuilt behind every K^-1 build)
    double *CK;    // scratch of the off-chain termination check (behind TT): A x [2][8 NS], P x, A'y, per-lane maxima [8][64], 1/Ed, 1/Eb, 1/D
    // off-chain termination check (checker_tick): state of wavefront NW-1's coroutine -- wave-uniform, scalar registers
    int cs, cr, ckn, ckk;
    bool ckc, cka;
    double ck_alpha, ck_eps_abs, ck_eps_rel, ck_eps_p, ck_eps_d, ck_rho_tol;      // the settings the checker reads, fetched once (the configuration block is in global memory)
    double c, cinv;
    // row weights: ADMM rho classes (OSQP set_rho_vec) or, while polishing, |flag| = 1/delta on active rows
    bool pol;
    double rho, rho_eq, rinv, rinv_eq;
    // per-lane constants of the compact layout (component = lane & 7 in every 64-element round): recomputed from the lane
    // number where they are used (a handful of integer instructions) instead of living in registers for the whole solve
    struct LaneC {
        int r0, r1, r2, bvar; // box rows acting on variable tj (7 = none, its coefficient is 0; r2 = the delay row on delta); variable of box row tj
        double rmask;         // 1 if tj is a real dynamics row
    };
    // ADMM weight of this thread's box rows (one per element round) and its reciprocal: the class of a row (loose / equality /
    // inequality, OSQP set_rho_vec) follows from its bounds alone, so the pair is fixed between two rho updates -- kept in
    // registers instead of three f64 compares and eight selects per element and iteration in update()
    static constexpr bool kGs = GS;                 // the equilibration vectors D / Ed / Eb live in global memory (SolveArgs::scal)
    static_assert(!GS || (NW == 2 && NT > 20 && !MF), "global scalings: planner two-wavefront kernels only");
    static constexpr bool kCacheW = kReg && !GS && (MF || (NW == 2 && NT > 20));   // (the N <= 20 DPP / one-wave instantiations have no registers to spare; the tail kernel keeps its own copies: tail_begin)
    static constexpr int kRnd = kCacheW ? (kFour ? 2 : ((NT + 1) * 8 + kStride - 1) / kStride) : 1;
    double wbx[kRnd], wbxi[kRnd];
    int wcls;       // (kRegState kernels: the rows' classes instead -- 2 bits per round: 0 inequality, 1 equality, 2 loose; the weights are the uniform rho, rho_eq, rho_min and
                    // their reciprocals, selected where used: seven registers less in a loop that keeps the element state)
    // NW == 4: element of set 1 / set 2 (see kS2n) that thread t owns in the right-hand side and update phases (>= 8 NS: none)
    __device__ __forceinline__ static int elem_set1(int t) {
        int e = t;
        if (e >= 8 * kS2a) e += 8 * kS2n;
        if (e >= 8 * kS2b) e += 8 * kS2n;
        return e;
    }
    __device__ __forceinline__ static int elem_set2(int t) {
        const int l = t & 63, w = t >> 6;
        return (w >= 2 && l < 8 * kS2n) ? 8 * (w == 2 ? kS2a : kS2b) + l : 8 * (NT + 1);
    }
    __device__ __forceinline__ void cache_box_weights() {
        if constexpr (kCacheW) {
#pragma unroll
            for (int r = 0; r < kRnd; ++r) {
                const int e = kFour ? (r == 0 ? elem_set1(tid) : elem_set2(tid))
                                    : tid + r * kStride;
                double w = rho, wi = rinv;
                if (e >= 0 && e < NS * 8) {
                    const double lo = Lo[e], hi = Hi[e];
                    const bool loose = lo < -kInfty * kMinScaling && hi > kInfty * kMinScaling, eq = hi - lo < kRhoTol;
                    w = loose ? kRhoMin : (eq ? rho_eq : rho); wi = loose ? 1.0 / kRhoMin : (eq ? rinv_eq : rinv);
                }
                wbx[r] = w; wbxi[r] = wi;
            }
            if constexpr (kRegState) {
                wcls = 0;
#pragma unroll
                for (int r = 0; r < kRnd; ++r) {
                    const int e = tid + r * kStride;
                    int c_ = 0;
                    if (e >= 0 && e < NS * 8) {
                        const double lo = Lo[e], hi = Hi[e];
                        const bool loose = lo < -kInfty * kMinScaling && hi > kInfty * kMinScaling, eq = hi - lo < kRhoTol;
                        c_ = loose ? 2 : (eq ? 1 : 0);
                    } else c_ = wcls & 3;       // (no second element: the first one's class)
                    wcls |= c_ << (2 * r);
                }
            }
            // (kFuse2: a lane without an element in the second round repeats its first one there -- with the first one's weights)
            if constexpr (MF && NW == 2 && kRnd == 2) { if (!(tid + kStride < NS * 8)) { wbx[1] = wbx[0]; wbxi[1] = wbxi[0]; } }
        }
    }
    int lpack;                // r0 | r1 << 4 | r2 << 8 | bvar << 12 | (tj < NX) << 16: one register for the whole solve, unpacked where used
    __device__ __forceinline__ LaneC lane_consts() const {
        LaneC l;
        const int lp = opaque(lpack);          // unpacked here, not hoisted into five live registers
        l.r0 = lp & 15; l.r1 = (lp >> 4) & 15; l.r2 = (lp >> 8) & 15; l.bvar = (lp >> 12) & 15;
        l.rmask = ((lp >> 16) & 1) ? 1.0 : 0.0;
        return l;
    }

    __device__ __forceinline__ Solver(const DevCfg &cf, double *smem)
        : cfg(cf), N(kFixN ? NT : cf.N), NS((kFixN ? NT : cf.N) + 1), tid(threadIdx.x), wv(threadIdx.x >> 6), lane(threadIdx.x & 63),
          ti((threadIdx.x & 63) >> 3), tj(threadIdx.x & 7), delay(kCtrl ? cf.steering_delay : 0) {
        double *p = smem;
        tA = p; p += NS * kTS;
        tS = tL = nullptr;
        if (!kFixN) { tS = p; p += NS * kTS; tL = p; p += NS * kTS; }
        const int V = NS * 8;
        X = p; p += V; Qv = p; p += V; D = p; p += V; XT = p; p += V; DX = p; p += V; VT = p; p += V; AT = p; p += V;
        Zd = p; p += V; Yd = p; p += V; Ed = p; p += V; ZTd = p; p += V; DYd = p; p += V;
        Zb = p; p += V; Yb = p; p += V; Eb = p; p += V; ZTb = p; p += V; DYb = p; p += V;
        if constexpr (GS) {     // D, Ed, Eb are bound to global memory by run(): the same walk without their three slots
            p = smem + NS * kTS;
            X = p; p += V; Qv = p; p += V; XT = p; p += V; DX = p; p += V; VT = p; p += V; AT = p; p += V;
            Zd = p; p += V; Yd = p; p += V; ZTd = p; p += V; DYd = p; p += V;
            Zb = p; p += V; Yb = p; p += V; ZTb = p; p += V; DYb = p; p += V;
            D = Ed = Eb = nullptr;
        }
        Lo = p; p += V; Hi = p; p += V;
        beq = p; p += 16; Pm = p; p += 64; dRl = p; p += 8; RED = p; p += kRedSize; SINK = p; p += 64 + 8 * NS;
        FX = nullptr;
        if constexpr (kFourSmall) { FX = p; p += 448; }     // (behind the image: the factorisation's hand-over area where VT + AT are too short for it)
        RT = WS = STG = TT = CK = nullptr;
        cs = cr = ckn = ckk = 0; ckc = cka = false; ck_alpha = ck_eps_abs = ck_eps_rel = ck_eps_p = ck_eps_d = ck_rho_tol = 0.0;
        if constexpr (TAIL) {   // behind the image: factor tiles, reduction slots, sweep scratch, staging
            tS = p; p += NS * kTS; tL = p; p += NS * kTS; RT = p; p += 96; WS = p; p += NW * 128; STG = p; p += kDenseRound * NS * 64;
            TT = STG; CK = STG + NS * 8 * 24;
        }
        {
            int first, cnt; rows_on(tj, first, cnt);
            const int r0 = cnt >= 1 ? first : 7, r1 = cnt >= 2 ? first + 1 : 7, r2 = (kCtrl && tj == 6) ? 6 : 7;
            lpack = r0 | (r1 << 4) | (r2 << 8) | (box_var(tj) << 12) | ((tj < NX ? 1 : 0) << 16);
        }
        c = 1.0; cinv = 1.0; pol = false; rho = rho_eq = rinv = rinv_eq = 0.0; rSm = rLt = rLb = 0.0;
        mS = tT = 0.0;
        set_fac_ids(lane);
    }
    // the part of the LDS block that a pool entry carries (try_park / restore): everything up to and including SINK
    static constexpr __host__ __device__ size_t image_doubles(int N) {
        return (size_t)(N + 1) * ((kFixN ? 1 : 3) * kTS + (GS ? 16 : 19) * 8 + 8) + 16 + 64 + 8 + kRedSize + 64;
    }
    static constexpr __host__ __device__ size_t lds_doubles(int N) {
        return image_doubles(N) + (TAIL ? (size_t)(N + 1) * (2 * kTS + kDenseRound * 64) + 96 + NW * 128 : 0) + (kFourSmall ? 448 : 0);
    }

    // ---- problem structure ---------------------------------------------------------------------
    __device__ __forceinline__ int nvar(int k) const { return k < N ? NB : NX; }
    __device__ __forceinline__ int nbox(int k) const { return kCtrl ? (k < N ? (k < delay ? 7 : 6) : 0) : (k < N ? 7 : 5); }
    __device__ __forceinline__ static int box_var(int r) { return kCtrl ? (r < 2 ? 0 : (r < 4 ? 6 : (r < 6 ? 7 : 6))) : r; }
    __device__ __forceinline__ static double box_sign(int r) { return kCtrl ? ((r == 0 || r == 3 || r == 5) ? -1.0 : 1.0) : 1.0; }
    __device__ __forceinline__ static void rows_on(int a, int &first, int &cnt) {
        if (kCtrl) { first = (a == 0) ? 0 : (a == 6 ? 2 : 4); cnt = (a == 0 || a == 6 || a == 7) ? 2 : 0; }
        else { first = a; cnt = 1; }
    }
    // unscaled Hessian entry inside stage block k (P = 2*M0, CTRL:398-432,464 / PLAN:145-169); Pm holds the
    // block of the stages k < N-1, the last input block has one dR less on its diagonal
    __device__ __forceinline__ double Pc(int k, int a, int b) const {
        double v = Pm[a * 8 + b];
        if (a >= NX) { if (k >= N) v = 0.0; else if (k == N - 1 && a == b) v -= 2.0 * dRl[a - NX]; }
        return v;
    }
    // OSQP constraint classes on scaled bounds (set_rho_vec)
    __device__ __forceinline__ static double rho_of(double lo, double hi, double rho) {
        if (lo < -kInfty * kMinScaling && hi > kInfty * kMinScaling) return kRhoMin;
        if (hi - lo < kRhoTol) return kRhoEqOverIneq * rho;
        return rho;
    }
    __device__ __forceinline__ void set_rho(double r) {
        rho = r; rho_eq = kRhoEqOverIneq * r; rinv = 1.0 / rho; rinv_eq = 1.0 / rho_eq;
        if constexpr (kUniScalars) { rho = unid(rho); rho_eq = unid(rho_eq); rinv = unid(rinv); rinv_eq = unid(rinv_eq); }
    }
    // weight of a box row / dynamics row in K = P + sig I + A' diag(W) A
    __device__ __forceinline__ double w_box(int e) const { return pol ? fabs(DYb[e]) : rho_of(Lo[e], Hi[e], rho); }
    __device__ __forceinline__ double w_dyn(int e) const { return pol ? fabs(DYd[e]) : rho_eq; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    // The thread indices again, from a value the optimiser cannot trace back: address arithmetic of the code BEHIND this point is formed
    // behind it, instead of in front of the ADMM loop with one live register per address all the way through the loop (the tail
    // kernel's registers hold K^-1: its loop has none to spare for the set-up and polish code's addresses)
    __device__ __forceinline__ void launder_ids() {
        tid = opaque(tid); wv = tid >> 6; lane = tid & 63; ti = lane >> 3; tj = tid & 7;
        lpack = opaque(lpack);
        set_fac_ids(lane);
    }
    // the lane indices of the factorisation (MFMA operand gathers, D form), from the lane number.  Formed again wherever code behind the
    // ADMM loop's entry needs them (the check block, the code behind the loop) instead of living in five registers through every
    // iteration: the loop keeps the element state there (kRegState)
    __device__ __forceinline__ void set_fac_ids(int l) {
        const int r_ = l >> 4, I_ = (l >> 3) & 1, J_ = (l >> 2) & 1, c_ = l & 3;
        gA = (4 * J_ + c_) * 8 + 4 * I_ + r_;       // lane (r, b = 2I + J, c) of an A operand: type A step T[4J + c][4I + r], type B step T[4I + c][4J + r]
        gB = (4 * I_ + c_) * 8 + 4 * J_ + r_;
        li = 4 * I_ + r_; lj = l & 7;
        tlane = 8 * dgroup(lj) + li;
    }
    // The termination checks, the rho update and the re-factorisation sit INSIDE the iteration loop and run once in 25 iterations or
    // less; what they compute from the lane indices (addresses, lane masks -- the factorisation's above all) is loop invariant, so the
    // optimiser forms it in front of the loop and keeps it in registers through every iteration: scalar register pairs for the masks,
    // which the loop does not have -- they are spilled to lanes of a vector register and cost the per-iteration code two v_readlane
    // per use of ITS OWN masks.  That block therefore works on laundered copies of the factorisation's lane indices (its arithmetic
    // stays inside it) and puts the loop's values back when it ends.  (Laundering the thread index as well un-hoists the iteration's
    // own addresses -- the loop header's phi is no longer folded: 639 -> 709 instructions per iteration in the headline kernel.)
    struct Ids { int tid, wv, lane, ti, tj, lpack, gA, gB, li, lj, tlane; };
    __device__ __forceinline__ Ids ids_save() const { return Ids{tid, wv, lane, ti, tj, lpack, gA, gB, li, lj, tlane}; }
    __device__ __forceinline__ void ids_restore(const Ids &k) {
        tid = k.tid; wv = k.wv; lane = k.lane; ti = k.ti; tj = k.tj; lpack = k.lpack; gA = k.gA; gB = k.gB; li = k.li; lj = k.lj; tlane = k.tlane;
    }
    // The same value, opaque to the optimiser.  The element loops of the ADMM iteration start from it so that their LDS
    // addresses are formed inside the loop as (one per-lane base) + (immediate offset of the array) instead of being hoisted
    // out of the iteration loop as one live address register per (array, access pattern) pair -- those registers, not the
    // arithmetic, are what pushed the kernel into scratch spills.
    __device__ __forceinline__ static int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
    __device__ __forceinline__ static int opaque_s(int v) { asm volatile("" : "+s"(v)); return v; }      // the same for a uniform value (scalar register)
    // wave-local ordering of LDS traffic (one wavefront executes its DS instructions in order; this only stops
    // the compiler from moving them across)
    __device__ __forceinline__ void wsync() const {
        if constexpr (kTwo || kFour || TAIL) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    }
    // all-reduce over the whole instance (one or two wavefronts); both waves combine in the same order, so
    // every lane of the block sees bit-identical results and takes identical branches
    // RED[64 .. 79] holds four slots of {sum wave 0, sum wave 1, max wave 0, max wave 1}.  A reduction needs a single
    // barrier when the previous user of its slot is separated from it by at least one block barrier; the slot is a
    // compile-time argument chosen per call site so that two reductions that can follow each other without an
    // intervening sync() never share one (the last reduction of primal_infeasible and the first of dual_infeasible).
    template <int SLOT>
    __device__ __forceinline__ double bsum(double v) const {
        v = wave_sum(v);
        if constexpr (kTwo) { double *r = RED + 64 + 4 * SLOT; if (lane == 0) r[wv] = v; __syncthreads(); v = r[0] + r[1]; }
        if constexpr (kFour) { double *r = RED + 112 + 8 * SLOT; if (lane == 0) r[wv] = v; __syncthreads(); v = (r[0] + r[1]) + (r[2] + r[3]); }
        if constexpr (TAIL) { double *r = RT + 8 * SLOT; if (lane == 0 && wv < kActW) r[wv] = v; __syncthreads(); v = tail_sum(r); }
        return v;
    }
    // tail kernel: the wavefronts beyond kActW own no element (their partial results are the neutral element and are not stored)
    __device__ __forceinline__ static double tail_sum(const double *r) {
        double v = r[0];
#pragma unroll
        for (int w = 1; w < kActW; ++w) v += r[w];
        return v;
    }
    __device__ __forceinline__ static double tail_max(const double *r) {
        double v = r[0];
#pragma unroll
        for (int w = 1; w < kActW; ++w) v = fmax(v, r[w]);
        return v;
    }
    template <int SLOT>
    __device__ __forceinline__ double bmax(double v) const {
        v = wave_max(v);
        if constexpr (kTwo) { double *r = RED + 64 + 4 * SLOT; if (lane == 0) r[2 + wv] = v; __syncthreads(); v = fmax(r[2], r[3]); }
        if constexpr (kFour) { double *r = RED + 112 + 8 * SLOT + 4; if (lane == 0) r[wv] = v; __syncthreads(); v = fmax(fmax(r[0], r[1]), fmax(r[2], r[3])); }
        if constexpr (TAIL) { double *r = RT + 8 * SLOT + 4; if (lane == 0 && wv < kActW) r[wv] = v; __syncthreads(); v = tail_max(r); }
        return v;
    }
    // sum of s and max of m over the instance with one barrier
    struct SumMax { double s, m; };
    template <int SLOT>
    __device__ __forceinline__ SumMax bsum_bmax(double s_, double m_) const {
        s_ = wave_sum(s_); m_ = wave_max(m_);
        if constexpr (kTwo) {
            double *r = RED + 64 + 4 * SLOT;
            if (lane == 0) { r[wv] = s_; r[2 + wv] = m_; }
            __syncthreads();
            s_ = r[0] + r[1]; m_ = fmax(r[2], r[3]);
        }
        if constexpr (kFour) {
            double *r = RED + 112 + 8 * SLOT;
            if (lane == 0) { r[wv] = s_; r[4 + wv] = m_; }
            __syncthreads();
            s_ = (r[0] + r[1]) + (r[2] + r[3]); m_ = fmax(fmax(r[4], r[5]), fmax(r[6], r[7]));
        }
        if constexpr (TAIL) {
            double *r = RT + 8 * SLOT;
            if (lane == 0 && wv < kActW) { r[wv] = s_; r[4 + wv] = m_; }
            __syncthreads();
            s_ = tail_sum(r); m_ = tail_max(r + 4);
        }
        return SumMax{s_, m_};
    }

    // The planner at N = 30 / 40 has a two- and a four-wavefront form of ONE arithmetic (launch_solve picks by the shape of the launch,
    // lpvmpc.h promises that every output word is the same either way).  A block-wide SUM associates by the number of wavefronts
    // (rounds of kStride elements per thread, then wave sums, then the wavefronts' partial results), so in these kernels every sum
    // whose value reaches a decision -- the cost normalisation of the equilibration, the two infeasibility certificates, the objective
    // value -- is added up in one order: the elements' terms go through AT (free wherever such a sum is taken), wavefront 0 adds them
    // lane by lane and then across the lanes (round 5: the objective; round 6: the other three).  Maxima are exact and stay as they are.
    static constexpr bool kFixedSums = MF && NT >= 30;
    // the caller has stored the terms to AT[0 .. 8 NS)
    __device__ __forceinline__ double ordered_total() const {
        sync();
        if (wv == 0) {
            double v = 0.0;
            for (int e = opaque(lane); e < NS * 8; e += 64) v += AT[e];
            v = wave_sum(v);
            if (lane == 0) RED[0] = v;
        }
        sync();
        const double v = RED[0];
        sync();
        return v;
    }

    // ---- operators on the scaled problem ---------------------------------------------------------
    // After equilibration the products that every A / A' application needs are cached in the two unused
    // rows of each stage tile:  Eid[k][r] = Ed D (identity part of dynamics row r; 0 for padding) at
    // tA[k][6][r] and Sb[k][r] = sign Eb D[var] (box row r; 0 when the row does not exist) at tA[k][7][r].
    // Everything below is branch-free: padding rows / columns carry exact zeros.
    __device__ __forceinline__ double &Eid(int k, int r) const { return tA[k * kTS + 48 + r]; }
    __device__ __forceinline__ double &Sb(int k, int r) const { return tA[k * kTS + 56 + r]; }
    __device__ __forceinline__ void cache_row_coefficients() {
        for (int e = opaque(tid); e < NS * 8; e += kStride) {
            const int k = e >> 3, r = e & 7;
            Eid(k, r) = r < NX ? Ed[e] * D[e] : 0.0;
            Sb(k, r) = r < nbox(k) ? box_sign(r) * Eb[e] * D[k * 8 + box_var(r)] : 0.0;
        }
        sync();
    }
    // dynamics-row product of the previous stage: sum_a [A|B]_{k-1}[r][a] v_{k-1}[a]   (0 for k = 0)
    __device__ __forceinline__ double prev_stage_dot(int k, const double *v) const {
        const int kp = k > 0 ? k - 1 : 0;
        const double *row = tA + kp * kTS + tj * 8, *sv = v + kp * 8;
        const double r0_ = row[0], r1_ = row[1], r2_ = row[2], r3_ = row[3], r4_ = row[4], r5_ = row[5], r6_ = row[6], r7_ = row[7];
        const double s0_ = sv[0], s1_ = sv[1], s2_ = sv[2], s3_ = sv[3], s4_ = sv[4], s5_ = sv[5], s6_ = sv[6], s7_ = sv[7];
        const double acc0 = (r0_ * s0_ + r1_ * s1_) + (r2_ * s2_ + r3_ * s3_);
        const double acc1 = (r4_ * s4_ + r5_ * s5_) + (r6_ * s6_ + r7_ * s7_);
        return k > 0 ? acc0 + acc1 : 0.0;
    }
    // an element loop on the wavefronts w0 .. w0 + nw - 1 only
    template <typename F>
    __device__ __forceinline__ void eloop(int w0, int nw, F f) const {
        if (wv >= w0 && wv < w0 + nw) for (int e = opaque(tid) - 64 * w0; e < NS * 8; e += 64 * nw) f(e);
    }
    // (dstD, dstB) = A * src
    __device__ __forceinline__ void A_mul(const double *src, double *dstD, double *dstB) const {
        const LaneC lc = lane_consts();
        for (int e = opaque(tid); e < NS * 8; e += kStride) {
            const int k = e >> 3;
            dstD[e] = lc.rmask * (Eid(k, tj) * src[e] - prev_stage_dot(k, src));
            dstB[e] = Sb(k, tj) * src[k * 8 + lc.bvar];
        }
    }
    // value of (A' (srcD, srcB))[e]
    template <bool DELAY_ROW = true>
    __device__ __forceinline__ double At_elem(int e, const double *srcD, const double *srcB, const LaneC &lc) const {
        const int k = e >> 3, kn = k < N ? k + 1 : N;         // the stage-N tile is all zero
        const double *col = tA + k * kTS + tj, *sd = srcD + kn * 8;
        double c_[NX], d_[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) { c_[r] = col[r * 8]; d_[r] = sd[r]; }
        const double sb0 = Sb(k, lc.r0), sb1 = Sb(k, lc.r1), w0 = srcB[k * 8 + lc.r0], w1 = srcB[k * 8 + lc.r1], ei = Eid(k, tj), wd = srcD[e];
        double acc0 = c_[0] * d_[0] + c_[1] * d_[1], acc1 = c_[2] * d_[2] + c_[3] * d_[3];
#pragma unroll
        for (int r = 4; r < NX; ++r) acc0 += c_[r] * d_[r];
        double boxes = sb0 * w0 + sb1 * w1;
        if constexpr (DELAY_ROW) { if (kCtrl && delay > 0) boxes += Sb(k, lc.r2) * srcB[k * 8 + lc.r2]; }       // pinned-steering row (uniform branch; DELAY_ROW false: the caller has excluded it)
        return boxes + (ei * wd - (acc0 + acc1));
    }
    // dst = A' * (srcD, srcB)
    __device__ __forceinline__ void At_mul(const double *srcD, const double *srcB, double *dst) const {
        const LaneC lc = lane_consts();
        for (int e = opaque(tid); e < NS * 8; e += kStride) dst[e] = At_elem(e, srcD, srcB, lc);
    }
    // row a of the stage-k Hessian block times (D_k .* v_k), plus the slew-rate coupling to the neighbouring
    // stages; ABS = true gives the infinity norm of the (scaled) column instead.  Branch-free: the Pm row is
    // zero outside its diagonal block, the stage-N input rows are switched off explicitly.
    template <bool ABS, bool RAW = false>
    __device__ __forceinline__ double P_row(int k, const double *v) const { return P_row_d<ABS, RAW>(k, v, D); }
    // the same with the variable scaling taken from Dv (the equilibration alternates between two copies of D)
    template <bool ABS, bool RAW = false>
    __device__ __forceinline__ double P_row_d(int k, const double *v, const double *D) const {
        const int a = tj, e = k * 8 + a;
        const double2 *pr = reinterpret_cast<const double2 *>(Pm + a * 8);
        const double2 *dr = reinterpret_cast<const double2 *>(D + k * 8);
        const double2 *vr = reinterpret_cast<const double2 *>(v + k * 8);
        const bool inp = a >= NX;
        const double dadj = (inp && k == N - 1) ? 2.0 * dRl[inp ? a - NX : 0] : 0.0;     // last input block: one dR less
        double acc = 0.0;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            double2 p = pr[h];
            const double2 d = dr[h];
            if (2 * h == a) p.x -= dadj;
            if (2 * h + 1 == a) p.y -= dadj;
            if (ABS) { acc = fmax(acc, fabs(p.x) * d.x); acc = fmax(acc, fabs(p.y) * d.y); }
            else { const double2 x = vr[h]; acc += p.x * d.x * x.x; acc += p.y * d.y * x.y; }
        }
        // coupling -2 dR between u_k and u_{k+1} / u_{k-1}
        const double cpl = inp ? 2.0 * dRl[inp ? a - NX : 0] : 0.0;
        const int en = (k + 1 < N) ? e + 8 : e, ep = (k >= 1) ? e - 8 : e;
        const double dn = (k + 1 < N) ? D[en] : 0.0, dp = (k >= 1) ? D[ep] : 0.0;
        if (ABS) { acc = fmax(acc, fabs(cpl) * dn); acc = fmax(acc, fabs(cpl) * dp); }
        else acc -= cpl * (dn * v[en] + dp * v[ep]);
        if (RAW) return (inp && k >= N) ? 0.0 : acc;           // the caller applies (acc * c) * D[e]
        return (inp && k >= N) ? 0.0 : acc * c * D[e];
    }
    // dst = P * src   (P = c D P0 D)
    __device__ __forceinline__ void P_mul(const double *src, double *dst) const {
        for (int e = opaque(tid); e < NS * 8; e += kStride) dst[e] = P_row<false>(e >> 3, src);
    }
    // infinity norm of column (k, tj) of the scaled Hessian
    __device__ __forceinline__ double P_colnorm(int k, int) const { return P_row<true>(k, D); }
    // the same without the final scaling by c D[e] (it depends on D only: reusable until D changes)
    __device__ __forceinline__ double P_colacc(int k) const { return P_row<true, true>(k, D); }

    // ---- Ruiz equilibration (OSQP scale_data) ----------------------------------------------------
    // D, E and c are accumulated while the [A|B] tiles stay UNSCALED (entries are multiplied by the current
    // E and D on the fly when the norms are taken); the tiles are scaled once at the end.
    // ---- one equilibration pass with both element rounds of a wavefront in one basic block (see pass() in scale_data) ----
    static constexpr bool kFuseS = kReg && NW == 2 && ((NT + 1) * 8 + kStride - 1) / kStride == 2 && !GS;
    struct NormIn { double de, ede, ebe, col[NX], es[NX], row[NB], ds[NB], b0, b1, b2, dbv; };
    struct NormOut { double d, e, b; };
    __device__ __forceinline__ NormIn norm_load(int e, const double *Ds, const double *Es, const double *Bs, int r0, int r1, int r2, int bvar) const {
        NormIn q;
        const int k = e >> 3, a = tj, kn = k < N ? k + 1 : N, kp = k > 0 ? k - 1 : 0;
        q.de = Ds[e]; q.ede = Es[e]; q.ebe = Bs[e];
        const double *col = tA + k * kTS + a, *row = tA + kp * kTS + a * 8;
#pragma unroll
        for (int r = 0; r < NX; ++r) { q.col[r] = col[r * 8]; q.es[r] = Es[kn * 8 + r]; }
#pragma unroll
        for (int b = 0; b < NB; ++b) { q.row[b] = row[b]; q.ds[b] = Ds[kp * 8 + b]; }
        q.b0 = Bs[k * 8 + r0]; q.b1 = Bs[k * 8 + r1]; q.b2 = Bs[k * 8 + r2]; q.dbv = Ds[k * 8 + bvar];
        return q;
    }
    __device__ __forceinline__ NormOut norm_form(int e, const NormIn &q, double pcol, int r0, int r1, int r2) const {
        const int k = e >> 3, a = tj, nb = nbox(k);
        double cmax = 0.0, rmax = 0.0;
#pragma unroll
        for (int r = 0; r < NX; ++r) cmax = fmax(cmax, fabs(q.col[r]) * q.es[r]);
#pragma unroll
        for (int b = 0; b < NB; ++b) rmax = fmax(rmax, fabs(q.row[b]) * q.ds[b]);
        const double de = q.de, ede = q.ede, ebe = q.ebe;
        const double t0 = q.b0 * de, t1 = q.b1 * de, t2 = q.b2 * de, td = ede * de, tr = rmax * ede, tb = ebe * q.dbv;
        double dn = fmax(pcol, cmax * de);
        dn = fmax(dn, r0 < nb ? t0 : 0.0);
        dn = fmax(dn, r1 < nb ? t1 : 0.0);
        dn = fmax(dn, r2 < nb ? t2 : 0.0);
        dn = fmax(dn, a < NX ? td : 0.0);
        dn = a < nvar(k) ? dn : 0.0;
        double en = fmax(td, k >= 1 ? tr : 0.0);
        en = a < NX ? en : 0.0;
        const double bn = a < nb ? tb : 0.0;
        NormOut o;
        o.d = de * inv_sqrt(limit_scaling(dn));
        o.e = ede * inv_sqrt(limit_scaling(en));
        o.b = ebe * inv_sqrt(limit_scaling(bn));
        return o;
    }
    // P_row_d<true, true>(k, Dn, Dn) -- the un-scaled column norm of the Hessian under the scaling Dn -- with the loads apart
    struct CostIn { double pr[8], dr[8], dnx, dpv, de, qv; };
    __device__ __forceinline__ CostIn cost_load(int e, const double *Dn) const {
        CostIn q;
        const int k = e >> 3, a = tj;
        const double2 *pr = reinterpret_cast<const double2 *>(Pm + a * 8);
        const double2 *dr = reinterpret_cast<const double2 *>(Dn + k * 8);
#pragma unroll
        for (int h = 0; h < 4; ++h) { const double2 p = pr[h], d = dr[h]; q.pr[2 * h] = p.x; q.pr[2 * h + 1] = p.y; q.dr[2 * h] = d.x; q.dr[2 * h + 1] = d.y; }
        const int en = (k + 1 < N) ? e + 8 : e, ep = (k >= 1) ? e - 8 : e;
        q.dnx = Dn[en]; q.dpv = Dn[ep]; q.de = Dn[e]; q.qv = Qv[e];
        return q;
    }
    __device__ __forceinline__ double cost_form(int e, const CostIn &q) const {
        const int k = e >> 3, a = tj;
        const bool inp = a >= NX;
        const double dr2 = 2.0 * dRl[inp ? a - NX : 0];
        const double dadj = (inp & (k == N - 1)) ? dr2 : 0.0;     // last input block: one dR less
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double pa = q.pr[i] - dadj;
            const double pv = i == a ? pa : q.pr[i];
            acc = fmax(acc, fabs(pv) * q.dr[i]);
        }
        const double cpl = inp ? dr2 : 0.0;
        const double dn = (k + 1 < N) ? q.dnx : 0.0, dp = (k >= 1) ? q.dpv : 0.0;
        acc = fmax(acc, fabs(cpl) * dn); acc = fmax(acc, fabs(cpl) * dp);
        return (inp & (k >= N)) ? 0.0 : acc;
    }
    __device__ __forceinline__ void scale_data() {
        const int ntrue = NS * NX + N * 2;
        // rounds of the element loops; with a compile-time horizon the un-scaled Hessian column norms of the cost
        // normalisation are kept for the next iteration's column norms (D does not change in between): one pass less
        constexpr int kRounds = kReg ? ((NT + 1) * 8 + kStride - 1) / kStride : 1;
        double pacc[kRounds];
        const LaneC lc = lane_consts();
        const int r0 = lc.r0, r1 = lc.r1, r2 = lc.r2, bvar = lc.bvar;
        // One equilibration pass reads the scalings (Ds, Es, Bs) and writes the updated ones to (Dn, En, Bn): the passes
        // alternate between the arrays D / Ed / Eb and XT / ZTd / ZTb (free during set-up), so the norms and the update
        // D <- D / sqrt(norm) are one element loop and one barrier (the values are those of the two-loop form, bit for bit).
        auto pass = [&](bool first, const double *Ds, const double *Es, const double *Bs, double *Dn, double *En, double *Bn) {
            // infinity norms of the columns of [P A'; A 0] -> new scalings
            auto norms = [&](int e, double pcol) {
                const int k = e >> 3, a = tj, kn = k < N ? k + 1 : N, kp = k > 0 ? k - 1 : 0;
                const double de = Ds[e], ede = Es[e], ebe = Bs[e];
                const double *col = tA + k * kTS + a, *row = tA + kp * kTS + a * 8;
                double cmax = 0.0, rmax = 0.0;
#pragma unroll
                for (int r = 0; r < NX; ++r) cmax = fmax(cmax, fabs(col[r * 8]) * Es[kn * 8 + r]);      // stage-N tile is zero
#pragma unroll
                for (int b = 0; b < NB; ++b) rmax = fmax(rmax, fabs(row[b]) * Ds[kp * 8 + b]);
                double dn = 0.0;
                if (a < nvar(k)) {
                    dn = fmax(pcol, cmax * de);
                    if (r0 < nbox(k)) dn = fmax(dn, Bs[k * 8 + r0] * de);
                    if (r1 < nbox(k)) dn = fmax(dn, Bs[k * 8 + r1] * de);
                    if (r2 < nbox(k)) dn = fmax(dn, Bs[k * 8 + r2] * de);
                    if (a < NX) dn = fmax(dn, ede * de);
                }
                double en = 0.0;          // dynamics row (k, a)
                if (a < NX) { en = ede * de; if (k >= 1) en = fmax(en, rmax * ede); }
                double bn = 0.0;          // box row (k, a)
                if (a < nbox(k)) bn = ebe * Ds[k * 8 + bvar];
                Dn[e] = de * inv_sqrt(limit_scaling(dn));
                En[e] = ede * inv_sqrt(limit_scaling(en));
                Bn[e] = ebe * inv_sqrt(limit_scaling(bn));
            };
            // Two-wavefront kernels with two element rounds (N = 20), passes behind the first: a wavefront that owns elements in both
            // rounds runs them as ONE basic block -- every load of both rounds first, the arithmetic branch-free (a maximum with a term
            // that does not apply is a maximum with 0: all terms are >= 0), the stores unconditional (a lane without a second element
            // repeats its first one).  The phases of the equilibration are latency bound: ten passes cost a ninth of a default solve.
            // Values and operation order are those of norms() / the loops below.
            bool fused = false;
            if constexpr (kFuseS) fused = !first && fuse2_wave();
            if constexpr (kFuseS) if (fused) {
                const int e0 = tid, e1 = tid + kStride, e1c = e1 < NS * 8 ? e1 : e0;
                const NormIn q0 = norm_load(e0, Ds, Es, Bs, r0, r1, r2, bvar), q1 = norm_load(e1c, Ds, Es, Bs, r0, r1, r2, bvar);
                const double pa1 = e1 < NS * 8 ? pacc[1] : pacc[0];
                const NormOut o0 = norm_form(e0, q0, pacc[0] * c * q0.de, r0, r1, r2), o1 = norm_form(e1c, q1, pa1 * c * q1.de, r0, r1, r2);
                Dn[e0] = o0.d; En[e0] = o0.e; Bn[e0] = o0.b;
                Dn[e1c] = o1.d; En[e1c] = o1.e; Bn[e1c] = o1.b;
            }
            if (!fused) {
            if constexpr (kReg) {
#pragma unroll
                for (int r = 0; r < kRounds; ++r) {
                    const int e = tid + r * kStride;
                    if (e < NS * 8) norms(e, first ? P_row_d<true>(e >> 3, Ds, Ds) : pacc[r] * c * Ds[e]);
                }
            } else {
                for (int e = opaque(tid); e < NS * 8; e += kStride) norms(e, P_row_d<true>(e >> 3, Ds, Ds));
            }
            }
            sync();
            // cost normalisation (with the new D)
            double psum = 0.0, qmax = 0.0;
            if constexpr (kFuseS) { if (fuse2_wave()) {
                const int e0 = tid, e1 = tid + kStride;
                const bool on1 = e1 < NS * 8;
                const int e1c = on1 ? e1 : e0;
                const CostIn q0 = cost_load(e0, Dn), q1 = cost_load(e1c, Dn);
                const double p0 = cost_form(e0, q0), p1 = cost_form(e1c, q1);
                pacc[0] = p0; pacc[1] = p1;
                const double m1 = fabs(c * q1.de * q1.qv);
                if constexpr (kFixedSums) { AT[e0] = p0 * c * q0.de; AT[e1c] = p1 * c * q1.de; }      // (no second element: the first one's term once more)
                else {
                    psum += p0 * c * q0.de;
                    const double ps1 = psum + p1 * c * q1.de;     // (the sum in the shape of the plain loop's: the same contraction)
                    psum = on1 ? ps1 : psum;
                }
                qmax = fmax(qmax, fabs(c * q0.de * q0.qv));
                qmax = fmax(qmax, on1 ? m1 : 0.0);
                fused = true;
            } else fused = false; }
            if (fused) { }
            else if constexpr (kReg) {
#pragma unroll
                for (int r = 0; r < kRounds; ++r) {
                    const int e = tid + r * kStride;
                    if (e < NS * 8) {
                        pacc[r] = P_row_d<true, true>(e >> 3, Dn, Dn);
                        if constexpr (kFixedSums) AT[e] = pacc[r] * c * Dn[e]; else psum += pacc[r] * c * Dn[e];
                        qmax = fmax(qmax, fabs(c * Dn[e] * Qv[e]));
                    }
                }
            } else {
                for (int e = opaque(tid); e < NS * 8; e += kStride) {
                    psum += P_row_d<true>(e >> 3, Dn, Dn);
                    qmax = fmax(qmax, fabs(c * Dn[e] * Qv[e]));
                }
            }
            SumMax pq;
            if constexpr (kFixedSums) { pq.s = ordered_total(); pq.m = bmax<0>(qmax); }     // (one summation order for the two forms of these kernels: see kFixedSums)
            else pq = bsum_bmax<0>(psum, qmax);
            psum = pq.s / (double)ntrue;
            qmax = limit_scaling(pq.m);
            const double ct = limit_scaling(fmax(psum, qmax));
            c *= 1.0 / ct;
        };
        const int n_it = cfg.scaling;
        int it = 0;
        if constexpr (kFuseS) {
            // Round 6: the passes behind the first keep what does not change between passes in REGISTERS -- the element's column of [A|B]_k,
            // its row of [A|B]_{k-1}, q, the thread's row of the Hessian block (38 doubles for two elements) -- and the element's own
            // scalings, which are the previous pass's results.  A pass then loads the neighbours' scalings only (24 instead of 47 values
            // per element): the equilibration is bound by the LDS pipe at four instances per CU (all of them start with it at once).
            // norm_form / cost_form and their operands are those of the fused form in pass(): every scaling is unchanged, bit for bit.
            if (n_it >= 1) { pass(true, D, Ed, Eb, XT, ZTd, ZTb); it = 1; }
            if (n_it >= 2) {
                const bool two = fuse2_wave();
                const int e0 = tid, e1 = tid + kStride;
                const bool on1 = two && e1 < NS * 8;
                const int e1c = on1 ? e1 : e0;
                struct EC { double col[NX], row[NB], qv; };
                auto consts = [&](int e) {
                    EC c_;
                    const int k = e >> 3, a = tj, kp = k > 0 ? k - 1 : 0;
                    const double *col = tA + k * kTS + a, *row = tA + kp * kTS + a * 8;
#pragma unroll
                    for (int r = 0; r < NX; ++r) c_.col[r] = col[r * 8];
#pragma unroll
                    for (int b = 0; b < NB; ++b) c_.row[b] = row[b];
                    c_.qv = Qv[e];
                    return c_;
                };
                const EC c0 = consts(e0), c1 = consts(e1c);
                double prow[8];
                {
                    const double2 *pr = reinterpret_cast<const double2 *>(Pm + tj * 8);
#pragma unroll
                    for (int h = 0; h < 4; ++h) { const double2 p = pr[h]; prow[2 * h] = p.x; prow[2 * h + 1] = p.y; }
                }
                // the thread's own scalings (the first pass left them in XT / ZTd / ZTb) and un-scaled Hessian column norms
                double sd[2] = {XT[e0], XT[e1c]}, se[2] = {ZTd[e0], ZTd[e1c]}, sb[2] = {ZTb[e0], ZTb[e1c]};
                if (!on1) pacc[1] = pacc[0];
                auto nload = [&](int e, const EC &ec, int i, const double *Ds, const double *Es, const double *Bs) {
                    NormIn q;
                    const int k = e >> 3, kn = k < N ? k + 1 : N, kp = k > 0 ? k - 1 : 0;
                    q.de = sd[i]; q.ede = se[i]; q.ebe = sb[i];
#pragma unroll
                    for (int r = 0; r < NX; ++r) { q.col[r] = ec.col[r]; q.es[r] = Es[kn * 8 + r]; }
#pragma unroll
                    for (int b = 0; b < NB; ++b) { q.row[b] = ec.row[b]; q.ds[b] = Ds[kp * 8 + b]; }
                    q.b0 = Bs[k * 8 + r0]; q.b1 = Bs[k * 8 + r1]; q.b2 = Bs[k * 8 + r2]; q.dbv = Ds[k * 8 + bvar];
                    return q;
                };
                auto cload = [&](int e, const EC &ec, int i, const double *Dn) {
                    CostIn q;
                    const int k = e >> 3;
                    const double2 *dr = reinterpret_cast<const double2 *>(Dn + k * 8);
#pragma unroll
                    for (int h = 0; h < 4; ++h) { const double2 d = dr[h]; q.pr[2 * h] = prow[2 * h]; q.pr[2 * h + 1] = prow[2 * h + 1]; q.dr[2 * h] = d.x; q.dr[2 * h + 1] = d.y; }
                    const int en = (k + 1 < N) ? e + 8 : e, ep = (k >= 1) ? e - 8 : e;
                    q.dnx = Dn[en]; q.dpv = Dn[ep]; q.de = sd[i]; q.qv = ec.qv;
                    return q;
                };
                auto rpass = [&](const double *Ds, const double *Es, const double *Bs, double *Dn, double *En, double *Bn) {
                    if (two) {
                        const NormIn q0 = nload(e0, c0, 0, Ds, Es, Bs), q1 = nload(e1c, c1, 1, Ds, Es, Bs);
                        const NormOut o0 = norm_form(e0, q0, pacc[0] * c * q0.de, r0, r1, r2), o1 = norm_form(e1c, q1, pacc[1] * c * q1.de, r0, r1, r2);
                        Dn[e0] = o0.d; En[e0] = o0.e; Bn[e0] = o0.b;
                        Dn[e1c] = o1.d; En[e1c] = o1.e; Bn[e1c] = o1.b;
                        sd[0] = o0.d; se[0] = o0.e; sb[0] = o0.b; sd[1] = o1.d; se[1] = o1.e; sb[1] = o1.b;
                    } else {
                        const NormIn q0 = nload(e0, c0, 0, Ds, Es, Bs);
                        const NormOut o0 = norm_form(e0, q0, pacc[0] * c * q0.de, r0, r1, r2);
                        Dn[e0] = o0.d; En[e0] = o0.e; Bn[e0] = o0.b;
                        sd[0] = o0.d; se[0] = o0.e; sb[0] = o0.b;
                    }
                    sync();
                    double psum = 0.0, qmax = 0.0;
                    if (two) {
                        const CostIn q0 = cload(e0, c0, 0, Dn), q1 = cload(e1c, c1, 1, Dn);
                        const double p0 = cost_form(e0, q0), p1 = cost_form(e1c, q1);
                        pacc[0] = p0; pacc[1] = p1;
                        const double m1 = fabs(c * q1.de * q1.qv);
                        if constexpr (kFixedSums) { AT[e0] = p0 * c * q0.de; AT[e1c] = p1 * c * q1.de; }
                        else {
                            psum += p0 * c * q0.de;
                            const double ps1 = psum + p1 * c * q1.de;
                            psum = on1 ? ps1 : psum;
                        }
                        qmax = fmax(qmax, fabs(c * q0.de * q0.qv));
                        qmax = fmax(qmax, on1 ? m1 : 0.0);
                    } else {
                        const CostIn q0 = cload(e0, c0, 0, Dn);
                        const double p0 = cost_form(e0, q0);
                        pacc[0] = p0;
                        if constexpr (kFixedSums) AT[e0] = p0 * c * q0.de; else psum += p0 * c * q0.de;
                        qmax = fmax(qmax, fabs(c * q0.de * q0.qv));
                    }
                    SumMax pq;
                    if constexpr (kFixedSums) { pq.s = ordered_total(); pq.m = bmax<0>(qmax); }
                    else pq = bsum_bmax<0>(psum, qmax);
                    psum = pq.s / (double)ntrue;
                    qmax = limit_scaling(pq.m);
                    const double ct = limit_scaling(fmax(psum, qmax));
                    c *= 1.0 / ct;
                };
                for (; it + 1 < n_it; it += 2) {
                    rpass(XT, ZTd, ZTb, D, Ed, Eb);
                    rpass(D, Ed, Eb, XT, ZTd, ZTb);
                }
                if (it < n_it) { rpass(XT, ZTd, ZTb, D, Ed, Eb); ++it; }
            }
            if (n_it & 1) {         // odd count: the result sits in the alternate arrays
                for (int e = opaque(tid); e < NS * 8; e += kStride) { D[e] = XT[e]; Ed[e] = ZTd[e]; Eb[e] = ZTb[e]; }
                sync();
            }
        } else {
        for (; it + 1 < n_it; it += 2) {
            pass(it == 0, D, Ed, Eb, XT, ZTd, ZTb);
            pass(false, XT, ZTd, ZTb, D, Ed, Eb);
        }
        if (it < n_it) {        // odd count: one more pass, then bring the result home
            pass(it == 0, D, Ed, Eb, XT, ZTd, ZTb);
            for (int e = opaque(tid); e < NS * 8; e += kStride) { D[e] = XT[e]; Ed[e] = ZTd[e]; Eb[e] = ZTb[e]; }
            sync();
        }
        }
        cinv = 1.0 / c;
        if constexpr (kUniScalars) { c = unid(c); cinv = unid(cinv); }
        // A <- E A D on the stored [A|B] tiles, four stages per trip (all loads before the stores)
        for (int k0 = 4 * wv; k0 < N; k0 += 4 * NW) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u < N ? k0 + u : N - 1;
                v[u] = tA[k * kTS + lane] * (Ed[(k + 1) * 8 + ti] * D[k * 8 + tj]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (k0 + u < N && ti < NX && tj < NB) tA[(k0 + u) * kTS + lane] = v[u];
        }
        sync();
    }

    // ---- block tridiagonal factorisation of K = P + sig I + A' diag(W) A --------------------------
    // K = L S L' (L unit block-lower bidiagonal, S block diagonal).  Each pivot block is factored
    // S_k = C C' (Cholesky) and only the TRIANGULAR factor is inverted; with G = K_{k,k-1} C_{k-1}^-T
    // the next pivot is S_k = K_kk - G G' (a symmetric product of accurately known factors), which
    // keeps the factorisation accurate for the 1/delta-weighted polish system (cond ~ 1e9), where a
    // Schur complement formed through an explicit S^-1 loses ~7 digits.  Stored for the solve:
    // S_k^-1 = C^-T C^-1 (tS / rS) and L_k = G C_{k-1}^-1 (tL / rL; transposed for odd k, see kkt_solve).
    // diagonal block (stage k) of K, element [ti][tj]; identity on the padding
    __device__ __forceinline__ double kd_block(int k, double sig) const {
        const int nv = nvar(k);
        if (!(ti < nv && tj < nv)) return (ti == tj) ? 1.0 : 0.0;
        double kd = Pc(k, ti, tj) * c * D[k * 8 + ti] * D[k * 8 + tj];
        if (ti == tj) {
            kd += sig;
            int first, cnt; rows_on(ti, first, cnt);
            for (int t = 0; t < cnt; ++t) { const int r = first + t; const double sb = Sb(k, r); kd += w_box(k * 8 + r) * sb * sb; }
            if (kCtrl && ti == 6 && delay > 0) { const double sb = Sb(k, 6); kd += w_box(k * 8 + 6) * sb * sb; }   // 0 where the row does not exist
            { const double ei = Eid(k, ti); kd += w_dyn(k * 8 + ti) * ei * ei; }
        }
        if (k < N) {
            const double *ca = tA + k * kTS + ti, *cb = tA + k * kTS + tj;
#pragma unroll
            for (int r = 0; r < NX; ++r) kd += w_dyn((k + 1) * 8 + r) * ca[r * 8] * cb[r * 8];
        }
        return kd;
    }
    // off-diagonal block K_{k,k-1}: rows = stage k, columns = stage k-1 (k >= 1), element [ti][tj]
    __device__ __forceinline__ double ko_down(int k) const {
        double ko = 0.0;
        if (tj < NB) {
            if (ti < NX) ko = -w_dyn(k * 8 + ti) * Eid(k, ti) * tA[(k - 1) * kTS + ti * 8 + tj];
            else if (ti == tj && ti < nvar(k)) ko = c * D[k * 8 + ti] * (-2.0 * dRl[ti - NX]) * D[(k - 1) * 8 + ti];
        }
        return ko;
    }
    // off-diagonal block K_{k,k+1} = K_{k+1,k}': rows = stage k, columns = stage k+1 (k < N), element [ti][tj]
    __device__ __forceinline__ double ko_up(int k) const {
        double ko = 0.0;
        if (ti < NB) {
            if (tj < NX) ko = -w_dyn((k + 1) * 8 + tj) * Eid(k + 1, tj) * tA[k * kTS + tj * 8 + ti];
            else if (ti == tj && tj < nvar(k + 1)) ko = c * D[(k + 1) * 8 + tj] * (-2.0 * dRl[tj - NX]) * D[k * 8 + tj];
        }
        return ko;
    }
    // Cholesky S = C C' merged with the forward substitution C W = I (8 pivots); returns W = C^-1 (lower triangular)
    __device__ __forceinline__ double chol_inverse(double s) const {
        double w = (ti == tj) ? 1.0 : 0.0;
#define LPVMPC_CHOL_STEP(T)                                                                                   \
        {                                                                                                         \
            const double d = bcast_lane<(T) * 9>(s);                                                              \
            const double rs = inv_sqrt(d);                                                                        \
            const double cit = bcast_row<(T)>(s) * rs;                                                            \
            const double cjt = __shfl(s, (T) * 8 + tj) * rs;                                                      \
            const double wtj = __shfl(w, (T) * 8 + tj) * rs;                                                      \
            if (ti > (T) && tj > (T)) s -= cit * cjt;                                                             \
            if (ti == (T)) w = wtj; else if (ti > (T)) w -= cit * wtj;                                            \
        }
        LPVMPC_CHOL_STEP(0) LPVMPC_CHOL_STEP(1) LPVMPC_CHOL_STEP(2) LPVMPC_CHOL_STEP(3)
        LPVMPC_CHOL_STEP(4) LPVMPC_CHOL_STEP(5) LPVMPC_CHOL_STEP(6) LPVMPC_CHOL_STEP(7)
#undef LPVMPC_CHOL_STEP
        return w;
    }
    // 8x8 products go through four 64-double scratch tiles per wavefront in LDS (the x~ / delta_x / scratch
    // vectors are free whenever a factorisation runs): an operand is written once and each lane then reads one
    // row of each factor with 16-byte loads.  T0 = K_{k,pred}, T1 = C_pred^-1, T2 = G, T3 = C_pred^-T (row-major).
    // One elimination step: given the coupling block ko to the previously eliminated stage (whose C^-1 sits in
    // T1/T3), returns s = kd - G G' and l = G C_pred^-1.
    struct SL { double s, l; };
    __device__ __forceinline__ SL schur_step(double *T, double ko, double kd) const {
        double *const T0 = T, *const T2 = T + 128;
        const double2 *const rowi0 = reinterpret_cast<const double2 *>(T0 + ti * 8);
        const double2 *const rowj1 = reinterpret_cast<const double2 *>(T + 64 + tj * 8);
        const double2 *const rowi2 = reinterpret_cast<const double2 *>(T2 + ti * 8);
        const double2 *const rowj2 = reinterpret_cast<const double2 *>(T2 + tj * 8);
        const double2 *const rowj3 = reinterpret_cast<const double2 *>(T + 192 + tj * 8);
        T0[lane] = ko;
        wsync();
        double g = 0.0;                                     // G = Koff C^-T : sum_t Koff[i][t] Cinv[j][t]
#pragma unroll
        for (int t = 0; t < 4; ++t) { const double2 a = rowi0[t], b = rowj1[t]; g += a.x * b.x; g += a.y * b.y; }
        T2[lane] = g;
        wsync();
        double s = kd, l = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {                       // S = Kd - G G' ;  L = G C^-1 = sum_t G[i][t] Cinv^T[j][t]
            const double2 a = rowi2[t], b = rowj2[t], cT = rowj3[t];
            s -= a.x * b.x; s -= a.y * b.y;
            l += a.x * cT.x; l += a.y * cT.y;
        }
        return SL{s, l};
    }
    // publish W = C^-1 (row-major in T1, transposed in T3) and return S^-1 = C^-T C^-1
    __device__ __forceinline__ double publish_and_invert(double *T, double w) const {
        const double2 *const rowi3 = reinterpret_cast<const double2 *>(T + 192 + ti * 8);
        const double2 *const rowj3 = reinterpret_cast<const double2 *>(T + 192 + tj * 8);
        wsync();
        T[64 + lane] = w; T[192 + tj * 8 + ti] = w;
        wsync();
        double sinv = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) { const double2 a = rowi3[t], b = rowj3[t]; sinv += a.x * b.x; sinv += a.y * b.y; }
        return sinv;
    }
    // ---- 8x8 matrices on the matrix cores ---------------------------------------------------------------------------
    // "D form" of an 8x8 matrix M, seen as 2x2 blocks of 4x4: lane (r, b = 2I + J, c) -- r = lane >> 4, b = (lane >> 2) & 3,
    // c = lane & 3 -- holds M[4I + r][4J + c].  It is what four-block MFMAs deliver, it is a B operand as it stands, and used
    // as an A operand it acts as M' (A[i][k] sits in lane (r = k, c = i)).  Row i of M occupies the 8 consecutive lanes of
    // group dgroup(i), column index = lane & 7: D form is the [ti][tj] layout of the other kernels with the rows permuted, so
    // the Cholesky / inversion below is the same DPP code with the row index renamed.
    __device__ __forceinline__ static constexpr int dgroup(int i) { return 2 * (i & 3) + (i >> 2); }
    // c + A B, given at = D form of A' and b = D form of B: two MFMAs (k = 0..3, 4..7); the operand quads are replicated by
    // DPP moves -- A operand of block (I, J): block (K, I) of at, B operand: block (K, J) of b.  The replicas are values of their
    // own (rep_a / rep_b): a matrix that enters several products of an elimination step is replicated once (G' four times as
    // three replicas, W of the previous stage comes replicated from the step that formed it), and a product that is to be
    // subtracted negates its A operand inside the MFMA (mm8r<true>) instead of replicating a negated copy.
    struct Rep2 { double x0, x1; };
    __device__ __forceinline__ static Rep2 rep_a(double at) {
        return Rep2{dpp_into<0x118, 0x8>(dpp_into<0x114, 0x6>(at, at), at),       // quads (0, 0, 1, 1) of at
                    dpp_into<0x108, 0x1>(dpp_into<0x104, 0x6>(at, at), at)};      // quads (2, 2, 3, 3)
    }
    __device__ __forceinline__ static Rep2 rep_b(double b) {
        return Rep2{dpp_into<0x128, 0xC>(b, b), dpp_into<0x128, 0x3>(b, b)};      // quads (0, 1, 0, 1) / (2, 3, 2, 3) of b
    }
    template <bool NEG = false>
    __device__ __forceinline__ static double mm8r(const Rep2 &a, const Rep2 &b, double c) {
        if constexpr (NEG) return mfma4_nega(a.x1, b.x1, mfma4_nega(a.x0, b.x0, c));
        else return mfma4(a.x1, b.x1, mfma4(a.x0, b.x0, c));
    }
    __device__ __forceinline__ static double mm8(double at, double b, double c) { return mm8r(rep_a(at), rep_b(b), c); }
    __device__ __forceinline__ double transpose_d(double v) const { return __shfl(v, tlane); }
    // Cholesky S = C C' merged with the forward substitution C W = I, in D form (see chol_inverse)
    __device__ __forceinline__ double chol_inverse_d(double s) const {
        double w = (li == lj) ? 1.0 : 0.0;
#define LPVMPC_CHOL_STEP_D(T)                                                                                 \
        {                                                                                                         \
            constexpr int G = dgroup(T);                                                                          \
            const double d = bcast_lane<G * 8 + (T)>(s);                                                          \
            const double rs = inv_sqrt(d);                                                                        \
            const double cit = bcast_row<(T)>(s) * rs;                                                            \
            const double cjt = __shfl(s, G * 8 + lj) * rs;                                                        \
            const double wtj = __shfl(w, G * 8 + lj) * rs;                                                        \
            s -= cit * cjt;     /* (unconditional: rows / columns <= T are never read again -- pivot, column and row of a later */ \
                                /* step T' lie in the trailing block, and s itself is not returned) */                    \
            if (li == (T)) w = wtj; else if (li > (T)) w -= cit * wtj;                                            \
        }
        LPVMPC_CHOL_STEP_D(0) LPVMPC_CHOL_STEP_D(1) LPVMPC_CHOL_STEP_D(2) LPVMPC_CHOL_STEP_D(3)
        LPVMPC_CHOL_STEP_D(4) LPVMPC_CHOL_STEP_D(5) LPVMPC_CHOL_STEP_D(6)
        // planner: variable 7 is padding at every stage (identity row and column, untouched by the Schur complements): its pivot step
        // changes nothing (d = rs = 1, w[7][.] *= 1)
        if constexpr (NB == 8) { LPVMPC_CHOL_STEP_D(7) }
#undef LPVMPC_CHOL_STEP_D
        return w;
    }
    // D form with the two off-diagonal 4x4 blocks exchanged (quads 1 <-> 2 of every 16-lane row): turns the A operand of a
    // type A sweep step into the one of a type B step (see mf_forward)
    __device__ __forceinline__ static double qswap(double v) {
#ifdef LPVMPC_USE_SHFL
        const int l = threadIdx.x & 63, q = (l >> 2) & 3;
        return __shfl(v, (q == 1 || q == 2) ? (l ^ 12) : l);
#else
        return dpp_into<0x114, 0x4>(dpp_into<0x104, 0x2>(v, v), v);     // row_shl:4 into quad 1, row_shr:4 into quad 2
#endif
    }
    // ---- the inputs of an elimination step that do not depend on the chain: K_kk and the coupling block, in D form ----------------
    // An elimination step is a chain of dependent products and an 8-pivot Cholesky (latency bound: a stage took ~2 700 cycles whether one
    // or four instances shared the CU); the diagonal block K_kk and the coupling block Ko' of a stage depend on the problem data alone.
    // fac_in() forms them for the NEXT stage -- branch-free (conditional loads become loads from a safe address + a select; the
    // Hessian entry, the slew-rate entry and the lane's index pair are per-lane constants of the factorisation: fac_lane) so that the
    // scheduler may place its loads and arithmetic into the stalls of the current stage's chain.  Values and operation order are those
    // of the former kd_d / ko_down_at / ko_up_at (bitwise: tools/ab_equal.py).
    struct FacLane {
        double pmc, pmlc;      // c P[li][lj] of a stage block / of the last input block (one dR less on its diagonal)
        double m2dr;           // -2 dR of the slew-rate entry of a coupling block (li == lj >= NX), else 0
        int idx, oth;          // coupling block: variable index of the dynamics row / of the other stage's variable (li, lj by the chain's direction)
        int r0, r1, r2;        // box rows acting on variable lj
        bool up;               // the chain runs from stage N downwards: its coupling block is K_{k,k+1}
    };
    struct FacIn { double kd, kot; };
    __device__ __forceinline__ FacLane fac_lane(bool up) const {
        FacLane f;
        const double pm = Pm[li * 8 + lj];
        const bool din = li == lj && li >= NX;
        const double dr = dRl[din ? li - NX : 0];
        const double pml = din ? pm - 2.0 * dr : pm;
        f.pmc = pm * c; f.pmlc = pml * c;
        f.m2dr = din ? -2.0 * dr : 0.0;
        f.idx = up ? li : lj; f.oth = up ? lj : li;
        const LaneC lc = lane_consts();
        f.r0 = lc.r0; f.r1 = lc.r1; f.r2 = lc.r2; f.up = up;
        return f;
    }
    // k: the stage (uniform); KO: with the coupling block to the stage eliminated before it (k - 1, or k + 1 on the chain that runs downwards).
    // Two halves: fac_fetch issues the LDS loads (at the top of the step that runs one stage earlier: the values arrive while that step's
    // products run), fac_form does the arithmetic (beside the Cholesky).
    struct FacRaw { double di, dj, sb0, sb1, sb2, ei, wb0, wb1, wb2, wdj, abl, wkn, wdh, eih, tal, dh, dl; };
    template <bool KO>
    __device__ __forceinline__ FacRaw fac_fetch(const FacLane &f, int k, const double *WDv, const double *WBv) const {
        FacRaw q;
        q.di = D[k * 8 + li]; q.dj = D[k * 8 + lj];
        q.sb0 = Sb(k, f.r0); q.sb1 = Sb(k, f.r1); q.sb2 = kCtrl ? Sb(k, f.r2) : 0.0; q.ei = Eid(k, lj);
        q.wb0 = WBv[k * 8 + f.r0]; q.wb1 = WBv[k * 8 + f.r1]; q.wb2 = kCtrl ? WBv[k * 8 + f.r2] : 0.0; q.wdj = WDv[k * 8 + lj];
        const int kn = k < N ? k + 1 : N, rl = li < NX ? li : 0;
        q.abl = tA[k * kTS + rl * 8 + lj]; q.wkn = WDv[kn * 8 + rl];
        q.wdh = q.eih = q.tal = q.dh = q.dl = 0.0;
        if constexpr (KO) {
            const int hi = f.up ? k + 1 : k, lo = hi - 1;       // the dynamics rows of stage hi couple the two stages
            const int ia = f.idx < NX ? f.idx : 0;
            q.wdh = WDv[hi * 8 + ia]; q.eih = Eid(hi, ia); q.tal = tA[lo * kTS + ia * 8 + f.oth];
            q.dh = D[hi * 8 + f.idx]; q.dl = D[lo * 8 + f.idx];
        }
        return q;
    }
    template <bool KO>
    __device__ __forceinline__ FacIn fac_form(const FacLane &f, int k, const FacRaw &q, double sig) const {
        FacIn r;
        // (every select below chooses between values that are already formed: written with an expression as an operand, or with &&, the
        // front end emits a branch and the optimiser sinks the operand's loads into it -- the basic block ends there)
        const int nv = nvar(k);
        const bool inside = (li < nv) & (lj < nv), dg = li == lj;
        const double pc = k == N - 1 ? f.pmlc : f.pmc;
        const double kdp = pc * q.di * q.dj;
        // diagonal: sigma + the box rows acting on the variable + the identity part of its dynamics row
        const double sb0 = q.sb0, sb1 = q.sb1, sb2 = q.sb2, ei = q.ei;
        double dd = sig + q.wb0 * sb0 * sb0 + q.wb1 * sb1 * sb1 + q.wdj * ei * ei;
        if constexpr (kCtrl) { const double w2l = q.wb2; const double w2 = delay > 0 ? w2l : 0.0; dd += w2 * sb2 * sb2; }    // pinned-steering row
        const double kdd = kdp + dd, ident = dg ? 1.0 : 0.0;
        const double kin = dg ? kdd : kdp;
        double kd = inside ? kin : ident;
        {   // + [A|B]_k' diag(w_{k+1}) [A|B]_k   (stage N has no successor: a zero operand adds nothing)
            const bool abm = (li < NX) & (k < N);
            const double ab = abm ? q.abl : 0.0;
            const double wab = q.wkn * ab;
            kd = mm8(ab, wab, kd);
        }
        r.kd = kd;
        r.kot = 0.0;
        if constexpr (KO) {     // D form of Ko': element [lj][li] of the block whose rows are the variables of stage k
            const int hi = f.up ? k + 1 : k;
            const bool rowd = f.idx < NX;
            const double t1 = -q.wdh * q.eih * q.tal;
            const double t2 = c * q.dh * f.m2dr * q.dl;
            const bool m1 = (f.oth < NB) & rowd, m2 = (f.oth < NB) & !rowd & (li == lj) & (f.idx < nvar(hi));
            const double kz = m2 ? t2 : 0.0;
            r.kot = m1 ? t1 : kz;
        }
        return r;
    }
    template <bool KO>
    __device__ __forceinline__ FacIn fac_in(const FacLane &f, int k, double sig, const double *WDv, const double *WBv) const {
        return fac_form<KO>(f, k, fac_fetch<KO>(f, k, WDv, WBv), sig);
    }
    // entry g (row-major offset) of the scaled [A|B] tile of stage k as an MFMA operand: dynamics rows only (rows 6, 7 of the
    // LDS tile cache row coefficients)
    __device__ __forceinline__ double ab_entry(int k, int g) const {
        const bool in = (g >> 3) < NX;
        const double v = tA[k * kTS + (in ? g : 0)];        // (loaded by every lane, then selected: no branch around the load)
        return in ? v : 0.0;
    }
    __device__ __forceinline__ void factor(double sig) {
        sync();
        if constexpr (kFour) {
            // The two-sided MFMA elimination below, each chain relayed over two wavefronts: the outer one (wv < 2) eliminates chain
            // positions 0 .. kHO and keeps the operand tiles 0 .. kHO-1 that these steps leave, hands W and S^-1 of position kHO to the
            // inner one (wv >= 2) through LDS, which goes on to position kMid-1 (operand tiles kHO .. kMid-2) and takes the link tile
            // kMid-1 from the middle stage (wavefront 2).  Every value is the one the two-wavefront kernel computes.
            double *const WDv = XT, *const WBv = DX, *const PUB = kFourSmall ? FX : VT;       // free vectors: row weights, hand-over area (VT + AT; N = 20: its own 448 words)
            for (int e = opaque(tid); e < NS * 8; e += kStride) { WDv[e] = w_dyn(e); WBv[e] = w_box(e); }
            sync();
            constexpr int P = kMid;
            const int chain = wv & 1;
            const bool inner = wv >= 2;
            double wd = 0.0, wtd = 0.0, sinv = 0.0;       // D forms of W, W' and S^-1 of the stage eliminated last
            Rep2 Awd{0.0, 0.0}, Bwd{0.0, 0.0};           // W replicated as an A / a B operand (formed with S^-1 = W' W, used again by the next step)
            const bool up = __builtin_amdgcn_readfirstlane(chain) != 0;
            const FacLane fl = fac_lane(up);
            auto stage_of = [&](int p) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(up ? N - p : p); };
            FacIn cur{0.0, 0.0};                          // K_kk and Ko' of the step about to run (formed one step ahead: fac_in)
            // first position of a chain: no predecessor
            auto head = [&]() __attribute__((always_inline)) {
                const FacIn nxt = fac_in<true>(fl, stage_of(1), sig, WDv, WBv);
                wd = chol_inverse_d(cur.kd);
                wtd = transpose_d(wd);
                Awd = rep_a(wd); Bwd = rep_b(wd);
                sinv = mm8r(Awd, Bwd, 0.0);
                cur = nxt;
            };
            // (the parity of p is a compile-time argument -- the loops below run two steps per trip: which operand forms a step leaves
            // for the sweeps, and so which tiles take a quad swap, depends on it alone; FETCH: form the next step's inputs)
            auto step = [&](const int p, const int j0, auto odd_c, auto fetch_c) __attribute__((always_inline)) {
                constexpr bool odd = decltype(odd_c)::value, fetch = decltype(fetch_c)::value;
                const int k = stage_of(p);
                const double gt = mm8r(rep_a(wtd), rep_b(cur.kot), 0.0);    // G' = W Ko'
                const Rep2 Agt = rep_a(gt), Bgt = rep_b(gt);
                const double sk = mm8r<true>(Agt, Bgt, cur.kd);             // S = Kd - G G'
                constexpr bool fa = odd, bb = (((P - (odd ? 1 : 0)) & 1) == 0) != kFlip;      // (operand forms: see the two-wavefront branch)
                const double fc = fa ? mm8r<true>(Awd, Bgt, 0.0) : mm8r<true>(Bwd, Agt, 0.0);      // -L' = -W' G'  (as it is / quads 1, 2 exchanged)
                const double bc = bb ? mm8r<true>(Bgt, Awd, 0.0) : mm8r<true>(Agt, Bwd, 0.0);      // -L  = -G W    (exchanged / as it is)
                const double fv = sinv, ba = ab_entry(k, bb ? gB : gA);                            // (S^-1 of the previous stage: formed in the form this step takes)
                // (two levels of comparisons, each against a value the optimiser cannot trace: compared against ONE traceable index -- as a
                // chain or as a switch -- the assignment becomes an indexed store and the tile arrays leave the registers for scratch memory)
                {
                    const int ix = p - 1 - j0;
#pragma unroll
                    for (int ph = 0; ph < (kMP + 3) / 4; ++ph) if (ph == opaque_s(ix >> 2)) {
#pragma unroll
                        for (int pl = 0; pl < 4; ++pl) if (4 * ph + pl < kMP && pl == opaque_s(ix & 3)) {
                            fC[4 * ph + pl] = fc; fV[4 * ph + pl] = fv; bC[4 * ph + pl] = bc; bA[4 * ph + pl] = ba;
                        }
                    }
                }
                FacIn nxt{0.0, 0.0};
                if constexpr (fetch) nxt = fac_in<true>(fl, stage_of(p + 1), sig, WDv, WBv);     // (loads and arithmetic here: fetched at the top of the step, as the two-wavefront kernels do, the 17 values cost these kernels registers their iteration loop pays for)
                wd = chol_inverse_d(sk);
                wtd = transpose_d(wd);
                Awd = rep_a(wd); Bwd = rep_b(wd);
                sinv = odd ? mm8r(Bwd, Awd, 0.0) : mm8r(Awd, Bwd, 0.0);     // S^-1 = W' W, in the form the NEXT step takes (an even one: exchanged)
                if constexpr (fetch) cur = nxt;
            };
            // steps PA .. PB, two per trip; the last one forms the inputs of step PB + 1 only with FETCH_LAST
            auto run_steps = [&](auto pa_c, auto pb_c, const int j0, auto fetch_last_c) __attribute__((always_inline)) {
                constexpr int PA = decltype(pa_c)::value, PB = decltype(pb_c)::value;
                constexpr bool fetch_last = decltype(fetch_last_c)::value;
                constexpr int PE = fetch_last ? PB : PB - 1;
                constexpr std::integral_constant<bool, (PA & 1) != 0> first_c{};
                constexpr std::integral_constant<bool, (PA & 1) == 0> second_c{};
                int p = PA;
#pragma unroll 1
                for (; p + 1 <= PE; p += 2) { step(p, j0, first_c, std::true_type{}); step(p + 1, j0, second_c, std::true_type{}); }
                if constexpr (PE >= PA && ((PE - PA) & 1) == 0) step(PE, j0, std::integral_constant<bool, (PE & 1) != 0>{}, std::true_type{});
                if constexpr (!fetch_last) step(PB, j0, std::integral_constant<bool, (PB & 1) != 0>{}, std::false_type{});
            };
            if (!inner) {
                cur = fac_in<false>(fl, stage_of(0), sig, WDv, WBv);
                head();
                run_steps(std::integral_constant<int, 1>{}, std::integral_constant<int, kHO>{}, 0, std::false_type{});
                PUB[chain * 128 + lane] = wd; PUB[chain * 128 + 64 + lane] = sinv;
            } else {
                cur = fac_in<true>(fl, stage_of(kHO + 1), sig, WDv, WBv);     // (while the outer wavefront eliminates its positions)
            }
            sync();
            if (inner) {
                wd = PUB[chain * 128 + lane]; sinv = PUB[chain * 128 + 64 + lane];
                wtd = transpose_d(wd);
                Awd = rep_a(wd); Bwd = rep_b(wd);
                run_steps(std::integral_constant<int, kHO + 1>{}, std::integral_constant<int, P - 1>{}, kHO, std::true_type{});
                // (cur: K_mm and the coupling block of this chain to the middle stage)
                if (chain == 1) { PUB[256 + lane] = wd; PUB[320 + lane] = wtd; PUB[384 + lane] = cur.kot; }
            }
            sync();
            double ltn = 0.0, ln = 0.0, sm = 0.0;                  // -L_link', -L_link of this wave's chain; S_m^-1
            if (wv == 2) {      // middle stage: S_m = K_mm - G_t G_t' - G_b G_b' with both neighbours' W
                const double gtt = mm8(wtd, cur.kot, 0.0);
                const Rep2 Agtt = rep_a(gtt), Bgtt = rep_b(gtt);
                double sk = mm8r<true>(Agtt, Bgtt, cur.kd);
                ltn = mm8r<true>(Awd, Bgtt, 0.0); ln = mm8r<true>(Agtt, Bwd, 0.0);
                const double wd1 = PUB[256 + lane], wtd1 = PUB[320 + lane], kot1 = PUB[384 + lane];
                const double gtb = mm8(wtd1, kot1, 0.0);
                const Rep2 Agtb = rep_a(gtb), Bgtb = rep_b(gtb);
                sk = mm8r<true>(Agtb, Bgtb, sk);
                PUB[lane] = mm8r<true>(rep_a(wd1), Bgtb, 0.0); PUB[64 + lane] = mm8r<true>(Agtb, rep_b(wd1), 0.0);
                const double wm = chol_inverse_d(sk);
                sm = mm8(wm, wm, 0.0);
                PUB[128 + lane] = sm;
            }
            sync();
            if (wv == 3) { ltn = PUB[lane]; ln = PUB[64 + lane]; sm = PUB[128 + lane]; }
            if (inner) {    // link to the middle stage, the middle pivot (see the two-wavefront branch)
                constexpr bool fa = (P & 1) != 0;
                fC[P - 1 - kHO] = fa ? ltn : qswap(ltn); fV[P - 1 - kHO] = sinv;      // (sinv: already in the form of step P, see step())
                bC[P - 1 - kHO] = kFlip ? ln : qswap(ln); bA[P - 1 - kHO] = ab_entry(kMid, kFlip ? gA : gB);     // (step P - 1 is of type B, of type A with kFlip)
                mS = kFlip ? qswap(sm) : sm;
            } else {        // wave 0's closing product [A|B]_0 x_0 (x_0 leaves the last backward step in layout B iff P is even); an outer
                constexpr bool tb = ((P & 1) == 0) != kFlip;      // wavefront has no use for the middle pivot: the tile takes its register
                mS = ab_entry(0, tb ? gB : gA);
            }
            sync();
            return;
        } else if constexpr (kMf) {
            // Same two-sided elimination as the DPP kernel below (wave 0: stages 0 .. kMid-1 upwards, wave 1: N .. kMid+1
            // downwards), with every 8x8 matrix in D form and every 8x8x8 product on the matrix cores (mm8), nothing staged
            // through LDS.  Per stage: G' = W Ko' (W = C_pred^-1), S = Kd - G G', L' = W' G', L = G W, C C' = S, W <- C^-1,
            // S^-1 = W' W; the finished step leaves its tiles in the operand registers of the MFMA sweeps.
            double *const WDv = XT, *const WBv = DX, *const PUB = VT;       // free vectors: row weights, hand-over area (VT + AT)
            for (int e = opaque(tid); e < NS * 8; e += kStride) { WDv[e] = w_dyn(e); WBv[e] = w_box(e); }
            sync();
            constexpr int P = kMid;
            double wd = 0.0, wtd = 0.0, sinv = 0.0;       // D forms of W, W' and S^-1 of the stage eliminated last
            Rep2 Awd{0.0, 0.0}, Bwd{0.0, 0.0};           // W replicated as an A / a B operand (formed with S^-1 = W' W, used again by the next step)
            const bool up = __builtin_amdgcn_readfirstlane(wv) != 0;
            const FacLane fl = fac_lane(up);
            auto stage_of = [&](int p) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(up ? N - p : p); };
            FacIn cur = fac_in<false>(fl, stage_of(0), sig, WDv, WBv);      // K_kk and Ko' of the step about to run (formed one step ahead: fac_in)
#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS != 5
            tlast = __builtin_amdgcn_s_memtime();
#endif
            {   // first position of the chain: no predecessor
                const FacIn nxt = fac_in<true>(fl, stage_of(1), sig, WDv, WBv);
                STAMP(4);
                wd = chol_inverse_d(cur.kd);
                wtd = transpose_d(wd);
                STAMP(6);
                Awd = rep_a(wd); Bwd = rep_b(wd);
                sinv = mm8r(Awd, Bwd, 0.0);
                STAMP(7);
                cur = nxt;
            }
            // (the parity of p is a compile-time argument -- the loop runs two steps per trip: which operand forms a step leaves for the
            // sweeps, and so which tiles take a quad swap, depends on it alone)
            auto step = [&](const int p, auto odd_c) __attribute__((always_inline)) {
                constexpr bool odd = decltype(odd_c)::value;
                const int k = stage_of(p), kx = stage_of(p + 1);
                const FacRaw raw = fac_fetch<true>(fl, kx, WDv, WBv);       // the next step's data (behind the last step: the middle stage's): in flight during the products
                const double gt = mm8r(rep_a(wtd), rep_b(cur.kot), 0.0);    // G' = W Ko'
                const Rep2 Agt = rep_a(gt), Bgt = rep_b(gt);
                const double sk = mm8r<true>(Agt, Bgt, cur.kd);             // S = Kd - G G'
                // forward step p consumes y_{p-1} (layout A for odd p): tiles -L_p and S_{p-1}^-1; backward step p - 1
                // consumes x at chain position p (layout B when P - p is even): tiles -L_p' and [A|B] of stage k.
                // A type A operand is the D form of the tile's transpose, a type B operand the same with quads 1, 2 exchanged -- and
                // a product with its two off-diagonal 4x4 blocks exchanged is the same product with the replicas' roles exchanged (block
                // b of the result then holds (I, J) = (b & 1, b >> 1): the A operand takes the B pattern of its matrix and vice versa),
                // so a tile comes out of its MFMAs in the form its step takes: no quad swap (round 5; the same MFMAs, the same sums).
                constexpr bool fa = odd, bb = (((P - (odd ? 1 : 0)) & 1) == 0) != kFlip;
                const double fc = fa ? mm8r<true>(Awd, Bgt, 0.0) : mm8r<true>(Bwd, Agt, 0.0);      // -L' = -W' G'  (as it is / exchanged)
                const double bc = bb ? mm8r<true>(Bgt, Awd, 0.0) : mm8r<true>(Agt, Bwd, 0.0);      // -L  = -G W    (exchanged / as it is)
                const double fv = sinv, ba = ab_entry(k, bb ? gB : gA);                            // (S^-1 of the previous stage: formed in the form this step takes)
#pragma unroll
                for (int pp = 0; pp < kMP; ++pp) if (pp == p - 1) { fC[pp] = fc; fV[pp] = fv; bC[pp] = bc; bA[pp] = ba; }
                STAMP(5);
                // the next step's inputs beside this step's Cholesky, in one basic block
                const FacIn nxt = fac_form<true>(fl, kx, raw, sig);
                STAMP(4);
                wd = chol_inverse_d(sk);
                wtd = transpose_d(wd);
                STAMP(6);
                Awd = rep_a(wd); Bwd = rep_b(wd);
                sinv = odd ? mm8r(Bwd, Awd, 0.0) : mm8r(Awd, Bwd, 0.0);     // S^-1 = W' W, in the form the NEXT step takes (an even one: exchanged)
                STAMP(7);
                cur = nxt;
            };
            {
                int p = 1;
#pragma unroll 1
                for (; p + 1 <= P - 1; p += 2) { step(p, std::true_type{}); step(p + 1, std::false_type{}); }
                if constexpr (((P - 1) & 1) != 0) step(P - 1, std::true_type{});
            }
            // (cur: K_mm and this chain's coupling block to the middle stage)
            if (wv == 1) { PUB[lane] = wd; PUB[64 + lane] = wtd; PUB[128 + lane] = cur.kot; }
            sync();
            double ltn = 0.0, ln = 0.0, sm = 0.0;                  // -L_link', -L_link of this wave's chain; S_m^-1
            if (wv == 0) {      // middle stage: S_m = K_mm - G_t G_t' - G_b G_b' with both neighbours' W
                const double gtt = mm8(wtd, cur.kot, 0.0);
                const Rep2 Agtt = rep_a(gtt), Bgtt = rep_b(gtt);
                double sk = mm8r<true>(Agtt, Bgtt, cur.kd);
                ltn = mm8r<true>(Awd, Bgtt, 0.0); ln = mm8r<true>(Agtt, Bwd, 0.0);
                const double wd1 = PUB[lane], wtd1 = PUB[64 + lane], kot1 = PUB[128 + lane];
                const double gtb = mm8(wtd1, kot1, 0.0);
                const Rep2 Agtb = rep_a(gtb), Bgtb = rep_b(gtb);
                sk = mm8r<true>(Agtb, Bgtb, sk);
                PUB[128 + lane] = mm8r<true>(rep_a(wd1), Bgtb, 0.0); PUB[192 + lane] = mm8r<true>(Agtb, rep_b(wd1), 0.0);
                const double wm = chol_inverse_d(sk);
                sm = mm8(wm, wm, 0.0);
                PUB[256 + lane] = sm;
            }
            sync();
            if (wv == 1) { ltn = PUB[128 + lane]; ln = PUB[192 + lane]; sm = PUB[256 + lane]; }
            {   // link to the middle stage (forward step P, backward step P - 1: its operand x_m is in layout B), the middle
                // pivot (type A) and wave 0's closing product [A|B]_0 x_0 (x_0 leaves the last backward step in layout B iff P is even)
                constexpr bool fa = (P & 1) != 0, tb = ((P & 1) == 0) != kFlip;
                fC[P - 1] = fa ? ltn : qswap(ltn); fV[P - 1] = sinv;      // (sinv: already in the form of step P, see step())
                bC[P - 1] = kFlip ? ln : qswap(ln); bA[P - 1] = ab_entry(kMid, kFlip ? gA : gB);     // (step P - 1 is of type B, of type A with kFlip)
                mS = kFlip ? qswap(sm) : sm;
                tT = ab_entry(0, tb ? gB : gA);
            }
            sync();
            return;
        } else if constexpr (kTwo) {
            // ---- two-sided elimination: wave 0 runs stages 0 .. kMid-1 upwards, wave 1 stages N .. kMid+1 downwards ----
            double *const T = XT + wv * 256;
            const int P = wv ? kP1 : kP0;
            for (int p = 0; p < P; ++p) {
                const int k = wv ? N - p : p;
                const double kd = kd_block(k, sig);
                double s = kd, l = 0.0;
                if (p >= 1) { const SL r = schur_step(T, wv ? ko_up(k) : ko_down(k), kd); s = r.s; l = r.l; }
                const double w = chol_inverse(s);
                const double sinv = publish_and_invert(T, w);
                const double lt = (p & 1) ? __shfl(l, tj * 8 + ti) : l;      // odd chain positions keep L transposed
#pragma unroll
                for (int pp = 0; pp < kRS; ++pp) if (pp == p) { rL[pp] = lt; rS[pp] = sinv; }
            }
            sync();
            // ---- middle stage: S_m = K_mm - G_t G_t' - G_b G_b' with both neighbours' C^-1 (wave 0) ----
            if (wv == 0) {
                const double kd = kd_block(kMid, sig);
                double s = kd, lt = 0.0, lb = 0.0;
                if (kP0 >= 1) { const SL r = schur_step(T, ko_down(kMid), kd); s = r.s; lt = r.l; }
                if (kP1 >= 1) {
                    // second neighbour: reuse the step with wave 1's published C^-1 (its T1/T3), on wave 0's T0/T2
                    double *const Tb = XT + 256;
                    wsync();
                    // copy wave 1's C^-1 tiles next to our T0/T2 so that schur_step finds them at T+64 / T+192
                    T[64 + lane] = Tb[64 + lane]; T[192 + lane] = Tb[192 + lane];
                    wsync();
                    const SL r = schur_step(T, ko_up(kMid), s);
                    s = r.s; lb = r.l;
                }
                const double w = chol_inverse(s);
                rSm = publish_and_invert(T, w);
                // link tiles are used with the form of the last chain stage: position P-1 odd -> stored as is
                rLt = ((kP0 - 1) & 1) ? lt : __shfl(lt, tj * 8 + ti);
                rLb = ((kP1 - 1) & 1) ? lb : __shfl(lb, tj * 8 + ti);
                RED[lane] = rLb;                                   // hand the lower link tile (already oriented) and the
                SINK[lane] = rSm;                                  // middle pivot inverse to wave 1
            }
            sync();
            if (wv == 1) { rLb = RED[lane]; rSm = SINK[lane]; }
            sync();
            return;
        }
        double *const T = XT;
#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS != 5
        if constexpr (TAIL) tlast = __builtin_amdgcn_s_memtime();
#endif
        if (!TAIL || wv == 0) {     // (tail kernel: wavefront 0 factors, the tiles go to LDS)
            for (int k = 0; k <= N; ++k) {
                const double kd = kd_block(k, sig);
                double s = kd, l = 0.0;
                if (k >= 1) { const SL r = schur_step(T, ko_down(k), kd); s = r.s; l = r.l; }
                const double w = chol_inverse(s);
                const double sinv = publish_and_invert(T, w);
                if constexpr (kReg) {
                    const double lt = (k & 1) ? __shfl(l, tj * 8 + ti) : l;      // odd stages keep L_k transposed
#pragma unroll
                    for (int kk = 0; kk <= NT; ++kk) if (kk == k) { rL[kk] = lt; rS[kk] = sinv; }
                } else {
                    tL[k * kTS + ((k & 1) ? (tj * 8 + ti) : lane)] = l;
                    tS[k * kTS + lane] = sinv;
                }
            }
        }
        sync();
        if constexpr (TAIL) { STAMP(4); if (!pol) { dense_build(); STAMP(5); } }      // the ADMM system: K^-1 into the registers of the workgroup
    }

    // ---- tail kernel: K^-1 as a dense matrix in registers ---------------------------------------------------------------
    // Block column c of K^-1 (the eight unit vectors of stage c at once) by block sweeps over the LDS tiles of the
    // factorisation K = L S L' above:   Y_c = I,  Y_k = -L_k Y_{k-1} (k > c);   V_k = S_k^-1 Y_k (k >= c), 0 (k < c);
    // X_N = V_N,  X_k = V_k - L_{k+1}' X_{k+1}.   One wavefront per block column, kDenseRound block columns per round; every
    // step is an 8x8x8 product with both operands read row-wise from LDS (the running matrix is kept transposed in the
    // wavefront's scratch tile).  A round's result [column block][stage][i][j] = K^-1[8 stage + i][8 block + j] goes through STG to
    // the threads that own those rows in dense_apply.  Element [a][b] of the tiles: S^-1 row-major, L_k row-major for even k,
    // transposed for odd k (as the sweeps of kkt_solve want them).
    // sum_t A[i][t] * B'[j][t]:  arow = &A[i][0] with element stride sa (1 or 8), brow = &B'[j][0] contiguous
    template <int SA>
    __device__ __forceinline__ static double tile_dot(const double *arow, const double *brow) {
        double a_[8], b_[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { a_[t] = arow[t * SA]; b_[t] = brow[t]; }
        const double p0 = (a_[0] * b_[0] + a_[1] * b_[1]) + (a_[2] * b_[2] + a_[3] * b_[3]);
        const double p1 = (a_[4] * b_[4] + a_[5] * b_[5]) + (a_[6] * b_[6] + a_[7] * b_[7]);
        return p0 + p1;
    }
    // row i of L_k (TRANSPOSED = false) or of L_k' (true) times the matrix whose transpose sits in bt
    template <bool TRANSPOSED>
    __device__ __forceinline__ double l_tile_dot(int k, const double *bt) const {
        const double *t = tL + k * kTS;
        const bool rowmajor = ((k & 1) == 0) != TRANSPOSED;     // the wanted row is contiguous in memory
        return rowmajor ? tile_dot<1>(t + ti * 8, bt + tj * 8) : tile_dot<8>(t + ti, bt + tj * 8);
    }
    __device__ __forceinline__ void dense_column_block(int c, double *out) {
        double *const Yt = WS + wv * 128, *const Xt = Yt + 64;
        // forward sweep and pivot products
        for (int k = 0; k < c; ++k) out[k * 64 + lane] = 0.0;
        Yt[tj * 8 + ti] = (ti == tj) ? 1.0 : 0.0;
        out[c * 64 + lane] = tS[c * kTS + lane];
        wsync();
        for (int k = c + 1; k <= N; ++k) {
            const double y = -l_tile_dot<false>(k, Yt);
            wsync();
            Yt[tj * 8 + ti] = y;
            wsync();
            out[k * 64 + lane] = tile_dot<1>(tS + k * kTS + ti * 8, Yt + tj * 8);
        }
        wsync();
        // backward sweep (in place: out[k] holds V_k until X_k replaces it)
        Xt[tj * 8 + ti] = out[N * 64 + lane];
        wsync();
        for (int k = N - 1; k >= 0; --k) {
            const double x = out[k * 64 + lane] - l_tile_dot<true>(k + 1, Xt);
            wsync();
            out[k * 64 + lane] = x;
            Xt[tj * 8 + ti] = x;
            wsync();
        }
    }
    __device__ __forceinline__ void dense_build() {
        constexpr int kRoundsD = (NT + 1 + kDenseRound - 1) / kDenseRound;
        const int s_ = tid & 7;
        // rows of this thread (light wavefronts: two rows, the other two slots are overwritten by tail_begin; clamped for the threads
        // beyond the last group: they read valid addresses and never use the values)
        const bool light = tid < 64 * kActW;
        const int r0 = light ? 2 * (tid >> 3) : kRowsA + 4 * ((tid - 64 * kActW) >> 3);
        int off[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int r = (r0 + q < NS * 8) ? r0 + q : NS * 8 - 1; off[q] = (r >> 3) * 64 + (r & 7) * 8 + s_; }
#pragma unroll
        for (int rd = 0; rd < kRoundsD; ++rd) {
            const int c = rd * kDenseRound + wv;
            if (wv < kDenseRound && c <= N) dense_column_block(c, STG + wv * (NS * 64));
            sync();
#pragma unroll
            for (int q = 0; q < kDenseRound; ++q) {
                const int cc = rd * kDenseRound + q;
                if (cc <= NT) {
                    const double *src = STG + q * (NS * 64);
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) dm[(q2 * kDC + cc) < kDM ? q2 * kDC + cc : 0] = src[off[q2]];
                }
            }
            sync();
        }
        // the staging area is free until the next build: it now holds T = A_d' A_d row by row -- for variable (k, j) the 24 coefficients of
        // x~ of the stages k-1, k, k+1 -- which the fused element phase needs (tail_fused).  A_d row (k, r): Eid(k, r) on variable (k, r),
        // -[A|B]_{k-1}[r][a] on the variables of stage k-1.  T does not depend on rho; it is rebuilt because the build above used its room.
        for (int i = tid; i < NS * 8 * 24; i += kStride) {
            const int e = i / 24, q = i - 24 * e, blk = q >> 3, a_ = q & 7, k = e >> 3, j = e & 7;
            double v = 0.0;
            if (blk == 0) { if (j < NX && k >= 1) v = -Eid(k, j) * tA[(k - 1) * kTS + j * 8 + a_]; }
            else if (blk == 1) {
                if (k < N) {
#pragma unroll
                    for (int r = 0; r < NX; ++r) v += tA[k * kTS + r * 8 + j] * tA[k * kTS + r * 8 + a_];
                }
                if (a_ == j && j < NX) { const double ei = Eid(k, j); v += ei * ei; }
            } else if (k < N && a_ < NX) v = -tA[k * kTS + a_ * 8 + j] * Eid(k + 1, a_);
            TT[i] = v;
        }
        // the reciprocals of the equilibration vectors that the checker's residual evaluation needs (as residuals() forms them)
        for (int i = tid; i < NS * 8; i += kStride) { CK[40 * NS + 512 + i] = 1.0 / Ed[i]; CK[48 * NS + 512 + i] = 1.0 / Eb[i]; CK[56 * NS + 512 + i] = 1.0 / D[i]; }
#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS == 4
        if (tid < 40) CK[64 * NS + 512 + tid] = 0.0;
#endif
        sync();
    }
    // XT <- K^-1 VT (the KKT solve of an ADMM iteration in the tail kernel; the caller places the barrier)
    __device__ __forceinline__ void dense_apply() {
        const int t0 = tid, s_ = t0 & 7;
        if (t0 < 64 * (NW - 1)) {
            // The right-hand side slice of this thread's column slot (kDC values, 16-byte loads) in three batches: the first two are
            // issued up front, the third behind the first batch's products -- all of it in flight at once would be 2 kDC registers on
            // top of the 8 NS that hold K^-1, and the kernel has 256
            static_assert(kDC == 21, "dense_apply: the batches are written for 21 stages");
            const double2 *rv = reinterpret_cast<const double2 *>(VT + s_ * kRhsPitch);
            double2 ra[4], rb[4], rc[3];
#pragma unroll
            for (int h = 0; h < 4; ++h) { ra[h] = rv[h]; rb[h] = rv[4 + h]; }
            __builtin_amdgcn_sched_barrier(0);
            STAMP4(2);
            const bool light = t0 < 64 * kActW;
            if (light) {                    // light wavefronts: rows 2g, 2g + 1, two partial sums each
                double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    a0 += dm[2 * h] * ra[h].x; a1 += dm[kDC + 2 * h] * ra[h].x;
                    b0 += dm[2 * h + 1] * ra[h].y; b1 += dm[kDC + 2 * h + 1] * ra[h].y;
                }
#pragma unroll
                for (int h = 0; h < 3; ++h) rc[h] = rv[8 + h];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    a0 += dm[8 + 2 * h] * rb[h].x; a1 += dm[kDC + 8 + 2 * h] * rb[h].x;
                    b0 += dm[8 + 2 * h + 1] * rb[h].y; b1 += dm[kDC + 8 + 2 * h + 1] * rb[h].y;
                }
                a0 += dm[16] * rc[0].x; a1 += dm[kDC + 16] * rc[0].x; b0 += dm[17] * rc[0].y; b1 += dm[kDC + 17] * rc[0].y;
                a0 += dm[18] * rc[1].x; a1 += dm[kDC + 18] * rc[1].x; b0 += dm[19] * rc[1].y; b1 += dm[kDC + 19] * rc[1].y;
                a0 += dm[20] * rc[2].x; a1 += dm[kDC + 20] * rc[2].x;
                a0 += b0; a1 += b1;
                red_j2(a0, a1);
                if (s_ == 0) { const int r = 2 * (t0 >> 3); XT[r] = a0; XT[r + 1] = a1; }
                STAMP4(3);
            } else {                        // heavy wavefronts: rows kRowsA + 4g .. + 3, one sum per row (four independent chains)
                double a_[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int h = 0; h < 4; ++h) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { a_[q] += dm[q * kDC + 2 * h] * ra[h].x; a_[q] += dm[q * kDC + 2 * h + 1] * ra[h].y; }
                }
#pragma unroll
                for (int h = 0; h < 3; ++h) rc[h] = rv[8 + h];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { a_[q] += dm[q * kDC + 8 + 2 * h] * rb[h].x; a_[q] += dm[q * kDC + 8 + 2 * h + 1] * rb[h].y; }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a_[q] += dm[q * kDC + 16] * rc[0].x; a_[q] += dm[q * kDC + 17] * rc[0].y;
                    a_[q] += dm[q * kDC + 18] * rc[1].x; a_[q] += dm[q * kDC + 19] * rc[1].y;
                    a_[q] += dm[q * kDC + 20] * rc[2].x;
                }
                red_j2(a_[0], a_[1]); red_j2(a_[2], a_[3]);
                const int g = (t0 - 64 * kActW) >> 3, r = kRowsA + 4 * g;
                if (s_ == 0 && g < kGroupsB) {
                    if constexpr (kRowsA + 4 * kGroupsB == (NT + 1) * 8) {      // (the heavy groups' rows end exactly at the last row: no guards)
                        XT[r] = a_[0]; XT[r + 1] = a_[1]; XT[r + 2] = a_[2]; XT[r + 3] = a_[3];
                    } else {
                        XT[r] = a_[0];
                        if (r + 1 < NS * 8) XT[r + 1] = a_[1];
                        if (r + 2 < NS * 8) XT[r + 2] = a_[2];
                        if (r + 3 < NS * 8) XT[r + 3] = a_[3];
                    }
                }
            }
        }
    }

    // ---- XT <- K^-1 XT ---------------------------------------------------------------------------
    // Stage vectors alternate between "column form" (component ti, even stages) and "row form"
    // (component tj, odd stages); L_k is stored transposed for odd k, so every step of the two sweeps is
    // one tile product followed by one all-reduce, with no lane transposition on the dependent chain.
    // ---- two-sided sweeps (NW == 2) -------------------------------------------------------------------
    // Chain position p of wave w is stage p (w = 0) or N - p (w = 1); even positions use column form, odd ones
    // row form, L of odd positions is stored transposed -- exactly the single-chain scheme below, run on two
    // half-length chains at once.  The chains meet at stage kMid (wave 0), then both sweep outwards again.
    template <bool BOT, bool STASHED>
    __device__ __forceinline__ void twisted_forward() {
        constexpr int P = BOT ? kP1 : kP0;
        // (an all-reduce leaves its result in every lane of the group: all of them store it, to one address -- no select, no dump area; see mf_forward)
        double *const vrow = VT + tj;
        double *const vcol = VT + ti;
        auto stage = [](int p) constexpr { return BOT ? NT - p : p; };
        // The right-hand side of the middle stage is read by BOTH waves in the backward sweep while wave 0 stores x_m
        // over it: keep a copy where nobody writes (RED[48..55]) so that a late wave 1 cannot pick up x_m instead.
        double yc = XT[stage(0) * 8 + ti], yr = 0.0;
        double bq[3];
        bq[1] = XT[stage(1) * 8 + tj];
        bq[2] = (P > 2) ? XT[stage(2) * 8 + ti] : 0.0;
#pragma unroll
        for (int p = 1; p < P; ++p) {
            const double bk = bq[p % 3];
            if (p + 2 < P) bq[(p + 2) % 3] = ((p + 2) & 1) ? XT[stage(p + 2) * 8 + tj] : XT[stage(p + 2) * 8 + ti];
            if (p & 1) {
                double pv = rS[p - 1] * yc, ch = rL[p] * yc;
                red_i2(ch, pv);
                vrow[stage(p - 1) * 8] = pv;
                yr = bk - ch;
            } else {
                double pv = rS[p - 1] * yr, ch = rL[p] * yr;
                red_j2(ch, pv);
                vcol[stage(p - 1) * 8] = pv;
                yc = bk - ch;
            }
        }
        // pivot product of the last chain stage, and this chain's contribution L_link y_last to the middle
        // right-hand side (published for the other wave; both waves then finish the middle stage redundantly)
        const double link = BOT ? rLb : rLt;
        if (kLastOdd) {     // y_last in row form -> contribution in column form
            double pv = rS[P - 1] * yr, cb = link * yr;
            red_j2(cb, pv);
            vcol[stage(P - 1) * 8] = pv;
            if (tj == 0) RED[32 + (BOT ? 8 : 0) + ti] = cb;
        } else {
            double pv = rS[P - 1] * yc, cb = link * yc;
            red_i2(cb, pv);
            vrow[stage(P - 1) * 8] = pv;
            if (ti == 0) RED[32 + (BOT ? 8 : 0) + tj] = cb;
        }
        // (behind the chain, not in front of it: see mf_forward)
        if (!STASHED && !BOT && lane < 8) RED[48 + lane] = XT[kMid * 8 + lane];
    }
    template <bool BOT>
    __device__ __forceinline__ void twisted_backward() {
        constexpr int P = BOT ? kP1 : kP0;
        double *const xrow = XT + tj;
        double *const xcol = XT + ti;
        auto stage = [](int p) constexpr { return BOT ? NT - p : p; };
        const double link = BOT ? rLb : rLt;
        double xc = 0.0, xr = 0.0;
        // middle stage, computed by both waves: y_m = b_m - (L_t y_{m-1}) - (L_b y_{m+1}),  x_m = S_m^-1 y_m
        // first chain step: x_{P-1} = v_{P-1} - L_link' x_m
        if (kLastOdd) {     // contributions / y_m in column form -> x_m in row form -> (via the register transpose) column form
            const double ym = RED[48 + ti] - (RED[32 + ti] + RED[40 + ti]);
            const double xm_row = red_i(rSm * ym);                          // x_m[tj]
            if (!BOT) xrow[kMid * 8] = xm_row;
            const double xm_col = __shfl(xm_row, ti);                       // lane (0, ti) holds x_m[ti]
            xr = VT[stage(P - 1) * 8 + tj] - red_i(link * xm_col); xrow[stage(P - 1) * 8] = xr;
        } else {
            const double ym = RED[48 + tj] - (RED[32 + tj] + RED[40 + tj]);
            const double xm_col = red_j(rSm * ym);                          // x_m[ti]
            if (!BOT) xcol[kMid * 8] = xm_col;
            const double xm_row = __shfl(xm_col, tj * 8);                   // lane (tj, 0) holds x_m[tj]
            xc = VT[stage(P - 1) * 8 + ti] - red_j(link * xm_row); xcol[stage(P - 1) * 8] = xc;
        }
        double vq[3];
        if (P >= 2) vq[(P - 2) % 3] = ((P - 2) & 1) ? VT[stage(P - 2) * 8 + tj] : VT[stage(P - 2) * 8 + ti];
        if (P >= 3) vq[(P - 3) % 3] = ((P - 3) & 1) ? VT[stage(P - 3) * 8 + tj] : VT[stage(P - 3) * 8 + ti];
#pragma unroll
        for (int p = P - 2; p >= 0; --p) {
            const double vk = vq[p % 3];
            if (p - 2 >= 0) vq[(p - 2) % 3] = ((p - 2) & 1) ? VT[stage(p - 2) * 8 + tj] : VT[stage(p - 2) * 8 + ti];
            if (((p + 1) & 1) == 0) { xr = vk - red_i(rL[p + 1] * xc); xrow[stage(p) * 8] = xr; }
            else                    { xc = vk - red_j(rL[p + 1] * xr); xcol[stage(p) * 8] = xc; }
        }
    }

    // ---- the two sweeps on the matrix cores -------------------------------------------------------------------------
    // v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4x4 blocks; lane = 16 r + 4 b + c.  A stage vector v[0..7] is kept
    // replicated over c in one of two layouts:  layout A: lane (r, b, c) holds v[4 (b >> 1) + r],  layout B: v[4 (b & 1) + r].
    // With the tile T cut into 4x4 blocks T_IJ, a "type A" step (operand in layout A) gives the blocks (T00, T10, T01, T11) to
    // b = 0..3: b and b ^ 2 then hold the two halves of one dot product, one row_ror:8 move and one add complete it, and the
    // result stands in layout B.  A "type B" step (operand in layout B) uses (T00, T01, T10, T11), completes with
    // row_half_mirror (b ^ 1; the values are uniform over c) and delivers layout A.  So one chain step
    //     y_p = b_p - L_p y_{p-1}      or      x_p = v_p - L_{p+1}' x_{p+1}
    // is ONE MFMA, two DPP moves and one add on the dependent path (~50 cycles measured, against ~150 for the two-MFMA step
    // with half swaps it replaces); the right-hand side enters as the C operand of the block that owns the result.  A second,
    // independent MFMA per step multiplies the same operand by another tile: the pivot product v_{p-1} = S_{p-1}^-1 y_{p-1}
    // on the way up, the dynamics-row product [A|B]_k x_k (z~ = A x~ of the ADMM update) on the way down.
    struct MfLane {
        int ln, eA, eB;         // lane (opaque to the optimiser), element held in layout A / B
        bool stA, stB;          // this lane stores / carries the C operand of a result in layout A / B (one lane per element)
    };
    __device__ __forceinline__ MfLane mf_lane() const {
        MfLane m;
        m.ln = lane;           // (round 5: not opaque any more: the sweeps' LDS addresses may be hoisted out of the ADMM loop now that the kernels have the registers -- see launder_ids; -9 % instructions per iteration)
        const int r = m.ln >> 4, b = (m.ln >> 2) & 3;
        m.eA = 4 * (b >> 1) + r; m.eB = 4 * (b & 1) + r;
        m.stA = (b & 1) == 0; m.stB = b < 2;
        return m;
    }
    template <bool TYPE_A>
    __device__ __forceinline__ static double mf_close(double d) { return d + dpp_mov<TYPE_A ? 0x128 : 0x141>(d); }
    // (vq: the pivot products the backward sweep's first two steps take -- this wavefront's own stores, fetched here, in front of the
    // barrier between the sweeps, so that behind it only the other chain's contribution to the middle stage is one LDS round trip away)
    template <bool BOT, bool STASHED>
    __device__ __forceinline__ void mf_forward(const MfLane &m, double (&vq)[kVQ]) {
        constexpr int P = kMid;
        auto stage = [](int p) constexpr { return BOT ? NT - p : p; };
        // a type A step delivers layout B: right-hand side element eB in the blocks stB, results stored by the lanes c = 0 of them
        // the right-hand side enters as the C operand of the owner blocks only: the other blocks load their "right-hand side" from the
        // all-zero area (SINK, same immediate offsets) instead of selecting 0 in front of every MFMA
        const double *const csA = m.stB ? XT + m.eB : SINK, *const csB = m.stA ? XT + m.eA : SINK;
        // Every lane holds a finished replica of its element after mf_close (both blocks of a pair end with the whole sum, the
        // columns c are identical): all 64 lanes store it to the element's own address -- eight addresses, eight lanes each, no
        // select and no dump area.  (Lanes that hit one address cost nothing; the earlier form, owner lanes to the vector and
        // the other 56 to a per-lane dump, took the same 6-8 cycles per store but collided with the dump's banks:
        // profiles/r04_microbench_lds_store.txt, r04_phase_pmc.txt.)
        double *const vsA = VT + m.eB;
        double *const vsB = VT + m.eA;
        double Y = XT[stage(0) * 8 + m.eA];
        double bq[3];                                                       // right-hand sides are fetched two steps ahead
        bq[1] = csA[stage(1) * 8];
        bq[2] = (P > 2) ? csB[stage(2) * 8] : 0.0;
        double dvl = 0.0;                                                   // pivot product of the previous step, not yet closed
#pragma unroll
        for (int p = 1; p < P; ++p) {
            const double bl = bq[p % 3];
            if (p + 2 < P) bq[(p + 2) % 3] = ((p + 2) & 1) ? csA[stage(p + 2) * 8] : csB[stage(p + 2) * 8];
            __builtin_amdgcn_sched_barrier(0);          // keep the fetch up here: sunk to its use, every step waits for LDS (placed behind the step's MFMAs in the source: no change)
            // the previous step's pivot product is closed and stored in the shadow of this step's MFMAs
            // (the scheduler closes it in FRONT of them, behind the chain's closing add; forcing it behind the MFMAs -- both back to back, or
            // chain MFMA / close / pivot MFMA as the backward sweep runs -- through register dependencies costs 1.1 % / 1.2 % of the default run)
            if constexpr (kRawV) {
                const double dc = mfma4(fC[p - 1], Y, bl);
                vq[p - 1] = mfma4(fV[p - 1], Y, 0.0);      // v_{p-1}, raw
                Y = (p & 1) ? mf_close<true>(dc) : mf_close<false>(dc);
            } else if (p & 1) {
                const double dc = mfma4(fC[p - 1], Y, bl), dv = mfma4(fV[p - 1], Y, 0.0);
                if (p >= 2) vsB[stage(p - 2) * 8] = mf_close<false>(dvl);
                Y = mf_close<true>(dc);
                dvl = dv;
            } else {
                const double dc = mfma4(fC[p - 1], Y, bl), dv = mfma4(fV[p - 1], Y, 0.0);
                vsA[stage(p - 2) * 8] = mf_close<true>(dvl);
                Y = mf_close<false>(dc);
                dvl = dv;
            }
        }
        if constexpr (!kRawV) { if (P >= 2) { if ((P - 1) & 1) vsA[stage(P - 2) * 8] = mf_close<true>(dvl); else vsB[stage(P - 2) * 8] = mf_close<false>(dvl); } }
        // last chain stage: its pivot product and this chain's contribution -L_link y_last to the middle right-hand side
        constexpr bool la = (P & 1) != 0;
        const double dl = mfma4(fC[P - 1], Y, 0.0), dv = mfma4(fV[P - 1], Y, 0.0);
        // the middle right-hand side survives x_m (see twisted_forward): copied here, behind the chain -- in front of it, the load, the
        // wait and the store were an LDS round trip before wave 0's first step
        double mid = 0.0;
        if (!STASHED && !BOT) mid = XT[kMid * 8 + (m.ln & 7)];
        RED[32 + (BOT ? 8 : 0) + (la ? m.eB : m.eA)] = mf_close<la>(dl);      // (first: the other chain waits for it)
        if constexpr (kRawV) vq[P - 1] = dv; else (la ? vsA : vsB)[stage(P - 1) * 8] = mf_close<la>(dv);
        if (!STASHED && !BOT) RED[48 + (m.ln & 7)] = mid;                      // (every lane stores a replica)
        wsync();
        if constexpr (!kRawV) {
            const double *const vlA = m.stB ? VT + m.eB : SINK, *const vlB = m.stA ? VT + m.eA : SINK;     // (the backward sweep's load pointers)
            vq[0] = vq[1] = vq[2] = 0.0;
            vq[(P - 1) % 3] = vlB[stage(P - 1) * 8];
            if (P >= 2) vq[(P - 2) % 3] = vlA[stage(P - 2) * 8];
        }
    }
    template <bool BOT>
    __device__ __forceinline__ void mf_backward(const MfLane &m, double (&vq)[kVQ]) {
        constexpr int P = kMid;
        auto stage = [](int p) constexpr { return BOT ? NT - p : p; };
        // backward step p consumes x at chain position p + 1; x_m arrives in layout B, so step p is of type B iff P - 1 - p is even
        const double *const vsA = m.stB ? VT + m.eB : SINK, *const vsB = m.stA ? VT + m.eA : SINK;         // v_p enters as C of the owner blocks of the OUTPUT layout (the others read zeros: see mf_forward)
        double *const xsA = XT + m.eB;         // results of a type A step (layout B); every lane stores its replica (see mf_forward)
        double *const xsB = XT + m.eA;
        // [A|B]_k x_k, filed under the dynamics rows of stage k + 1 -- RAW (kRawV): the two blocks of a pair file their partial sums in AT and
        // VT (free since the pivot products stay in registers), update() adds them: the closing move / add leaves the sweep
        double *const asA = (kRawV && !m.stB ? VT : AT) + (BOT ? 0 : 16) + m.eB;
        double *const asB = (kRawV && !m.stA ? VT : AT) + (BOT ? 0 : 16) + m.eA;
        auto typeB = [](int p) constexpr { return (((P - 1 - p) & 1) == 0) != kFlip; };
        // (pivot products are fetched two steps ahead; those of the first two steps came with vq)
        // middle stage, on both waves: x_m = S_m^-1 (b_m - L_t y_{m-1} - L_b y_{m+1}); type A step, x_m in layout B
        const int em = kFlip ? m.eB : m.eA;         // (kFlip: a type B step -- right-hand side in layout B, x_m in layout A)
        const double ym = RED[48 + em] + (RED[32 + em] + RED[40 + em]);
        double X = kFlip ? mf_close<false>(mfma4(mS, ym, 0.0)) : mf_close<true>(mfma4(mS, ym, 0.0));
        if constexpr (!BOT) XT[kMid * 8 + (kFlip ? m.eA : m.eB)] = X;
        double dal = 0.0;                                                   // dynamics-row product of the previous step, not yet closed
#pragma unroll
        for (int p = P - 1; p >= 0; --p) {
            const double vl = vq[kRawV ? p : p % 3];
            if constexpr (!kRawV) { if (p - 2 >= 0) vq[(p - 2) % 3] = typeB(p - 2) ? vsB[stage(p - 2) * 8] : vsA[stage(p - 2) * 8]; }
            __builtin_amdgcn_sched_barrier(0);
            // the previous step's [A|B] product is closed and filed in the shadow of this step's MFMAs; both waves' first
            // step multiplies [A|B]_kMid x_m: wave 1 files it, wave 0 drops it
            const bool keep = BOT || p + 1 != P - 1;
            // (round 4: x of the previous step is stored behind this step's MFMAs, not between its closing add and them -- the store's
            // issue is off the dependent chain)
            if (typeB(p)) {
                const double dc = mfma4(bC[p], X, vl), da = mfma4(bA[p], X, 0.0);
                if (p <= P - 2) xsA[stage(p + 1) * 8] = X;
                if (p <= P - 2 && keep) asA[stage(p + 1) * 8] = (kRawV ? dal : mf_close<true>(dal));
                X = mf_close<false>(dc);
                dal = da;
            } else {
                const double dc = mfma4(bC[p], X, vl), da = mfma4(bA[p], X, 0.0);
                if (p <= P - 2) xsB[stage(p + 1) * 8] = X;
                if (p <= P - 2 && keep) asB[stage(p + 1) * 8] = (kRawV ? dal : mf_close<false>(dal));
                X = mf_close<true>(dc);
                dal = da;
            }
        }
        if (typeB(0)) xsB[stage(0) * 8] = X; else xsA[stage(0) * 8] = X;      // x_0
        {   // product of step 0 (type B iff P - 1 is even)
            constexpr bool keep = BOT || 0 != P - 1;
            if constexpr (keep) { if (typeB(0)) asB[stage(0) * 8] = (kRawV ? dal : mf_close<false>(dal)); else asA[stage(0) * 8] = (kRawV ? dal : mf_close<true>(dal)); }
        }
        if (!BOT) {     // [A|B]_0 x_0 -> dynamics rows of stage 1; x_0 is in layout B iff P is even
            constexpr bool tb = ((P & 1) == 0) != kFlip;
            const double dt = mfma4(tT, X, 0.0);
            ((kRawV && !(tb ? m.stA : m.stB)) ? VT : AT)[8 + (tb ? m.eA : m.eB)] = kRawV ? dt : mf_close<!tb>(dt);
        }
    }

    // ---- the same sweeps relayed over two wavefronts per chain (NW == 4) -----------------------------------------------
    // Forward step p (1 .. P; P = the link step) uses operand tile p - 1; the outer wavefront of a chain runs the steps 1 .. kHO and
    // leaves y at chain position kHO in RED[144 ..] (layout B iff kHO is odd), the inner one picks it up and runs kHO + 1 .. P.
    // Backward step p (P - 1 .. 0) uses tile p: inner wavefront middle stage and P - 1 .. kHO, outer wavefront kHO - 1 .. 0 starting
    // from x at position kHO, which the inner one stored to XT like every other x.  The instructions of a step are those of
    // mf_forward / mf_backward; only the loop bounds and the register index of a tile (p - 1 - J0, p - J0) differ.
    // A wavefront's part of one KKT solve is ONE function with the workgroup's barriers inside (the role is uniform over the
    // wavefront and read into a scalar register: real branches, every wavefront passes the same five barriers B1 .. B4 + the
    // caller's): what a part needs that does not come from the other wavefronts -- addresses, the right-hand sides and pivot
    // products it fetches two steps ahead, which are this wavefront's own stores or older than the last barrier -- is issued in
    // front of the barrier it waits at, so that behind the barrier only the relayed vector is one LDS round trip away (a sweep
    // phase of the two-wavefront kernel costs ~750 cycles beside its ten steps of ~65: section 4 of DESIGN.md).
    // ELEM: the ADMM iteration -- the inner wavefront also builds the right-hand side of its set-2 elements (before its forward
    // steps: only this wavefront reads them) and updates them (after its backward steps: x~ and [A|B] x~ of these stages are its own).
    template <bool BOT, bool ELEM>
    __device__ __forceinline__ void outer4(const MfLane &m, double sigma, double alpha, bool want_delta) {
        // (tools/check_kernel_resources.py counts the s_barrier instructions between the ROLE markers of a group: the four role paths of
        // relay4 sit in wavefront-divergent control flow and must pass the same number of workgroup barriers)
        asm volatile("; LPVMPC_ROLE_BEGIN relay4");
        constexpr int P = kMid;
        auto stage = [](int p) constexpr { return BOT ? NT - p : p; };
        auto typeB = [](int p) constexpr { return (((P - 1 - p) & 1) == 0) != kFlip; };
        double vr[kRawV ? kHO : 1];     // kRawV: the pivot products of this wavefront's chain positions, raw
        {   // ---- forward steps 1 .. kHO
            constexpr int PA = 1, PB = kHO;
            const double *const csA = m.stB ? XT + m.eB : SINK, *const csB = m.stA ? XT + m.eA : SINK;
            double *const vsA = VT + m.eB;
            double *const vsB = VT + m.eA;
            double Y = XT[stage(0) * 8 + m.eA];
            double bq[3] = {0.0, 0.0, 0.0};                                     // right-hand sides are fetched two steps ahead
            bq[PA % 3] = (PA & 1) ? csA[stage(PA) * 8] : csB[stage(PA) * 8];
            if (PA + 1 <= PB) bq[(PA + 1) % 3] = ((PA + 1) & 1) ? csA[stage(PA + 1) * 8] : csB[stage(PA + 1) * 8];
            double dvl = 0.0;                                                   // pivot product of the previous step, not yet closed
#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS == 3
            STAMP(6);
#endif
#pragma unroll
            for (int p = PA; p <= PB; ++p) {
                const double bl = bq[p % 3];
                if (p + 2 <= PB) bq[(p + 2) % 3] = ((p + 2) & 1) ? csA[stage(p + 2) * 8] : csB[stage(p + 2) * 8];
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (kRawV) {      // (the pivot product stays in its register, raw: see kRawV)
                    const double dc = mfma4(fC[p - 1], Y, bl);
                    vr[p - 1] = mfma4(fV[p - 1], Y, 0.0);
                    Y = (p & 1) ? mf_close<true>(dc) : mf_close<false>(dc);
                } else if (p & 1) {
                    const double dc = mfma4(fC[p - 1], Y, bl), dv = mfma4(fV[p - 1], Y, 0.0);
                    if (p > PA) vsB[stage(p - 2) * 8] = mf_close<false>(dvl);
                    Y = mf_close<true>(dc);
                    dvl = dv;
                } else {
                    const double dc = mfma4(fC[p - 1], Y, bl), dv = mfma4(fV[p - 1], Y, 0.0);
                    if (p > PA) vsA[stage(p - 2) * 8] = mf_close<true>(dvl);
                    Y = mf_close<false>(dc);
                    dvl = dv;
                }
            }
#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS == 3
            STAMP(7);
#endif
            RED[144 + (BOT ? 8 : 0) + ((kHO & 1) ? m.eB : m.eA)] = Y;           // y at position kHO: every lane stores its replica
            if constexpr (!kRawV) { if (PB & 1) vsA[stage(PB - 1) * 8] = mf_close<true>(dvl); else vsB[stage(PB - 1) * 8] = mf_close<false>(dvl); }
        }
        // ---- backward steps kHO - 1 .. 0: the pivot products of the first two are this wavefront's own stores
        constexpr int PA = kHO - 1, PB = 0;
        const double *const vsA = m.stB ? VT + m.eB : SINK, *const vsB = m.stA ? VT + m.eA : SINK;
        double *const xsA = XT + m.eB;
        double *const xsB = XT + m.eA;
        // [A|B]_k x_k, filed under the dynamics rows of stage k + 1 -- RAW (kRawV): the two blocks of a pair file their partial sums in AT and
        // VT (free since the pivot products stay in registers), update() adds them: the closing move / add leaves the sweep
        double *const asA = (kRawV && !m.stB ? VT : AT) + (BOT ? 0 : 16) + m.eB;
        double *const asB = (kRawV && !m.stA ? VT : AT) + (BOT ? 0 : 16) + m.eA;
        wsync();
        double vq[3] = {0.0, 0.0, 0.0};
        if constexpr (!kRawV) {
            vq[PA % 3] = typeB(PA) ? vsB[stage(PA) * 8] : vsA[stage(PA) * 8];
            if (PA - 1 >= PB) vq[(PA - 1) % 3] = typeB(PA - 1) ? vsB[stage(PA - 1) * 8] : vsA[stage(PA - 1) * 8];
        }
        sync();             // B1: y is handed over; the inner wavefronts sweep up ...
        STAMP(1);
        sync();             // B2: ... and, behind the middle stage, ...
        STAMP(2);
        sync();             // B3: ... back
        STAMP(3);
        double X = XT[stage(kHO) * 8 + (typeB(kHO) ? m.eA : m.eB)];            // x at position kHO, left by the inner wavefront's last step
        double dal = 0.0;                                                       // dynamics-row product of the previous step, not yet closed
#pragma unroll
        for (int p = PA; p >= PB; --p) {
            const double vl = kRawV ? vr[kRawV ? p : 0] : vq[p % 3];
            if constexpr (!kRawV) { if (p - 2 >= PB) vq[(p - 2) % 3] = typeB(p - 2) ? vsB[stage(p - 2) * 8] : vsA[stage(p - 2) * 8]; }
            __builtin_amdgcn_sched_barrier(0);
            // (x of the previous step is stored behind this step's MFMAs, not between its closing add and them: the store's issue is off the chain)
            if (typeB(p)) {
                const double dc = mfma4(bC[p], X, vl), da = mfma4(bA[p], X, 0.0);
                if (p < PA) { xsA[stage(p + 1) * 8] = X; asA[stage(p + 1) * 8] = (kRawV ? dal : mf_close<true>(dal)); }
                X = mf_close<false>(dc);
                dal = da;
            } else {
                const double dc = mfma4(bC[p], X, vl), da = mfma4(bA[p], X, 0.0);
                if (p < PA) { xsB[stage(p + 1) * 8] = X; asB[stage(p + 1) * 8] = (kRawV ? dal : mf_close<false>(dal)); }
                X = mf_close<true>(dc);
                dal = da;
            }
        }
        if (typeB(0)) { xsB[stage(0) * 8] = X; asB[stage(0) * 8] = (kRawV ? dal : mf_close<false>(dal)); } else { xsA[stage(0) * 8] = X; asA[stage(0) * 8] = (kRawV ? dal : mf_close<true>(dal)); }        // x_0, product of step 0
        if (!BOT) {     // [A|B]_0 x_0 -> dynamics rows of stage 1; x_0 is in layout B iff P is even
            constexpr bool tb = ((P & 1) == 0) != kFlip;
            const double dt = mfma4(mS, X, 0.0);                                // (mS: [A|B]_0 on the outer wavefronts, see factor)
            ((kRawV && !(tb ? m.stA : m.stB)) ? VT : AT)[8 + (tb ? m.eA : m.eB)] = kRawV ? dt : mf_close<!tb>(dt);
        }
        sync();             // B4
        STAMP(4);
        asm volatile("; LPVMPC_ROLE_END relay4");
    }
    template <bool BOT, bool ELEM>
    __device__ __forceinline__ void inner4(const MfLane &m, double sigma, double alpha, bool want_delta) {
        asm volatile("; LPVMPC_ROLE_BEGIN relay4");
        constexpr int P = kMid, J0 = kHO;
        auto stage = [](int p) constexpr { return BOT ? NT - p : p; };
        auto typeB = [](int p) constexpr { return (((P - 1 - p) & 1) == 0) != kFlip; };
        if constexpr (ELEM) { build_rhs_set<2>(sigma); wsync(); }
        double vr[kRawV ? P - J0 : 1];  // kRawV: the pivot products of this wavefront's chain positions, raw
        double *const vsAw = VT + m.eB;
        double *const vsBw = VT + m.eA;
        const double *const vsA = m.stB ? VT + m.eB : SINK, *const vsB = m.stA ? VT + m.eA : SINK;
        double *const xsA = XT + m.eB;
        double *const xsB = XT + m.eA;
        {   // ---- forward steps kHO + 1 .. P (P: the link step)
            constexpr int PA = J0 + 1, PB = P - 1;                              // the steps that take a right-hand side
            const double *const csA = m.stB ? XT + m.eB : SINK, *const csB = m.stA ? XT + m.eA : SINK;
            if (!BOT && m.ln < 8) RED[48 + m.ln] = XT[kMid * 8 + m.ln];        // the middle right-hand side survives x_m (see twisted_forward)
            double bq[3] = {0.0, 0.0, 0.0};
            if (PA <= PB) bq[PA % 3] = (PA & 1) ? csA[stage(PA) * 8] : csB[stage(PA) * 8];
            if (PA + 1 <= PB) bq[(PA + 1) % 3] = ((PA + 1) & 1) ? csA[stage(PA + 1) * 8] : csB[stage(PA + 1) * 8];
            STAMP(0);       // (inner wavefronts: own clock -- right-hand side set 2 + prologue)
            sync();         // B1: the outer wavefront has left y at position kHO
            STAMP(1);       // waiting at B1
            double Y = RED[144 + (BOT ? 8 : 0) + ((J0 & 1) ? m.eB : m.eA)];
            double dvl = 0.0;
#pragma unroll
            for (int p = PA; p <= PB; ++p) {
                const double bl = bq[p % 3];
                if (p + 2 <= PB) bq[(p + 2) % 3] = ((p + 2) & 1) ? csA[stage(p + 2) * 8] : csB[stage(p + 2) * 8];
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (kRawV) {
                    const double dc = mfma4(fC[p - 1 - J0], Y, bl);
                    vr[p - 1 - J0] = mfma4(fV[p - 1 - J0], Y, 0.0);
                    Y = (p & 1) ? mf_close<true>(dc) : mf_close<false>(dc);
                } else if (p & 1) {
                    const double dc = mfma4(fC[p - 1 - J0], Y, bl), dv = mfma4(fV[p - 1 - J0], Y, 0.0);
                    if (p > PA) vsBw[stage(p - 2) * 8] = mf_close<false>(dvl);
                    Y = mf_close<true>(dc);
                    dvl = dv;
                } else {
                    const double dc = mfma4(fC[p - 1 - J0], Y, bl), dv = mfma4(fV[p - 1 - J0], Y, 0.0);
                    if (p > PA) vsAw[stage(p - 2) * 8] = mf_close<true>(dvl);
                    Y = mf_close<false>(dc);
                    dvl = dv;
                }
            }
            // last chain stage: this chain's contribution -L_link y_last to the middle right-hand side first (the other chain waits for it), then the pivot products
            constexpr bool la = (P & 1) != 0;
            const double dl = mfma4(fC[P - 1 - J0], Y, 0.0), dv = mfma4(fV[P - 1 - J0], Y, 0.0);
            RED[32 + (BOT ? 8 : 0) + (la ? m.eB : m.eA)] = mf_close<la>(dl);
            if constexpr (kRawV) vr[P - 1 - J0] = dv;
            else {
                if (PB >= PA) { if (PB & 1) vsAw[stage(PB - 1) * 8] = mf_close<true>(dvl); else vsBw[stage(PB - 1) * 8] = mf_close<false>(dvl); }
                (la ? vsAw : vsBw)[stage(P - 1) * 8] = mf_close<la>(dv);
            }
        }
        // ---- backward: middle stage, steps P - 1 .. kHO; the pivot products of the first two steps are this wavefront's own stores
        constexpr int PA = P - 1, PB = J0;
        // [A|B]_k x_k, filed under the dynamics rows of stage k + 1 -- RAW (kRawV): the two blocks of a pair file their partial sums in AT and
        // VT (free since the pivot products stay in registers), update() adds them: the closing move / add leaves the sweep
        double *const asA = (kRawV && !m.stB ? VT : AT) + (BOT ? 0 : 16) + m.eB;
        double *const asB = (kRawV && !m.stA ? VT : AT) + (BOT ? 0 : 16) + m.eA;
        wsync();
        double vq[3] = {0.0, 0.0, 0.0};
        if constexpr (!kRawV) {
            vq[PA % 3] = typeB(PA) ? vsB[stage(PA) * 8] : vsA[stage(PA) * 8];
            if (PA - 1 >= PB) vq[(PA - 1) % 3] = typeB(PA - 1) ? vsB[stage(PA - 1) * 8] : vsA[stage(PA - 1) * 8];
        }
        STAMP(2);           // forward steps
        sync();             // B2: both chains' contributions to the middle right-hand side are there
        STAMP(3);           // waiting at B2
        // middle stage, on both inner wavefronts: x_m = S_m^-1 (b_m - L_t y_{m-1} - L_b y_{m+1}); type A step, x_m in layout B
        const int em = kFlip ? m.eB : m.eA;         // (kFlip: a type B step -- right-hand side in layout B, x_m in layout A)
        const double ym = RED[48 + em] + (RED[32 + em] + RED[40 + em]);
        double X = kFlip ? mf_close<false>(mfma4(mS, ym, 0.0)) : mf_close<true>(mfma4(mS, ym, 0.0));
        if constexpr (!BOT) XT[kMid * 8 + (kFlip ? m.eA : m.eB)] = X;
        double dal = 0.0;
#pragma unroll
        for (int p = PA; p >= PB; --p) {
            const double vl = kRawV ? vr[kRawV ? p - J0 : 0] : vq[p % 3];
            if constexpr (!kRawV) { if (p - 2 >= PB) vq[(p - 2) % 3] = typeB(p - 2) ? vsB[stage(p - 2) * 8] : vsA[stage(p - 2) * 8]; }
            __builtin_amdgcn_sched_barrier(0);
            const bool keep = BOT || p + 1 != P - 1;      // both chains' first step multiplies [A|B]_kMid x_m: the bottom chain files it
            if (typeB(p)) {     // (x of the previous step is stored behind this step's MFMAs: see outer4)
                const double dc = mfma4(bC[p - J0], X, vl), da = mfma4(bA[p - J0], X, 0.0);
                if (p < PA) xsA[stage(p + 1) * 8] = X;
                if (p < PA && keep) asA[stage(p + 1) * 8] = (kRawV ? dal : mf_close<true>(dal));
                X = mf_close<false>(dc);
                dal = da;
            } else {
                const double dc = mfma4(bC[p - J0], X, vl), da = mfma4(bA[p - J0], X, 0.0);
                if (p < PA) xsB[stage(p + 1) * 8] = X;
                if (p < PA && keep) asB[stage(p + 1) * 8] = (kRawV ? dal : mf_close<false>(dal));
                X = mf_close<true>(dc);
                dal = da;
            }
        }
        if (typeB(PB)) xsB[stage(PB) * 8] = X; else xsA[stage(PB) * 8] = X;      // x at position kHO first: the outer wavefront waits for it
        {   // product of this wavefront's last step
            constexpr bool keep = BOT || PB != P - 1;
            if constexpr (keep) { if (typeB(PB)) asB[stage(PB) * 8] = (kRawV ? dal : mf_close<false>(dal)); else asA[stage(PB) * 8] = (kRawV ? dal : mf_close<true>(dal)); }
        }
        STAMP(4);           // backward steps
        sync();             // B3: x at position kHO is in XT; the outer wavefronts sweep back
        STAMP(5);           // waiting at B3
        if constexpr (ELEM) { wsync(); update<2>(alpha, want_delta); }
        STAMP(6);           // update set 2
        sync();             // B4
        STAMP(7);           // waiting at B4
        asm volatile("; LPVMPC_ROLE_END relay4");
    }
    // the four parts of one KKT solve of the four-wavefront kernel (the right-hand side is complete in XT, a barrier behind it)
    template <bool ELEM>
    __device__ __forceinline__ void relay4(const MfLane &m, double sigma, double alpha, bool want_delta) {
        const int role = __builtin_amdgcn_readfirstlane(wv);
        if (role == 0) outer4<false, ELEM>(m, sigma, alpha, want_delta);
        else if (role == 1) outer4<true, ELEM>(m, sigma, alpha, want_delta);
        else if (role == 2) inner4<false, ELEM>(m, sigma, alpha, want_delta);
        else inner4<true, ELEM>(m, sigma, alpha, want_delta);
    }

    // the two MFMA sweeps of the two-wavefront kernels (the lane constants come from the caller: the ADMM iteration forms them in front
    // of the barrier that completes the right-hand side)
    template <bool STASHED = false>
    __device__ __forceinline__ void kkt_solve_mf(const MfLane &m) {
        double vq[kVQ];
        if (wv == 0) mf_forward<false, STASHED>(m, vq); else mf_forward<true, STASHED>(m, vq);
        sync();
        STAMP(1);
        if (wv == 0) mf_backward<false>(m, vq); else mf_backward<true>(m, vq);
        sync();
        STAMP(2);
    }
    // STASHED: the caller has already copied the middle stage's right-hand side to RED[48..55]
    template <bool STASHED = false>
    __device__ __forceinline__ void kkt_solve() {
        if constexpr (kFour) {      // (the polish solves; the ADMM iteration adds its element phases: iterate4)
            relay4<false>(mf_lane(), 0.0, 0.0, false);
            return;
        } else if constexpr (kMf) {
            kkt_solve_mf<STASHED>(mf_lane());
            return;
        } else if constexpr (kTwo) {
            if (wv == 0) twisted_forward<false, STASHED>(); else twisted_forward<true, STASHED>();
            sync();
            STAMP(1);
            if (wv == 0) twisted_backward<false>(); else twisted_backward<true>();
            sync();
            STAMP(2);
            return;
        }
        if constexpr (kReg) {
            // Results of a reduction are replicated over 8 lanes: all eight store it, to the one address (no exec masking, no
            // select).  Offsets 8*k are immediates.
            double *const vrow = VT + tj;      // row-form results   (component tj)
            double *const vcol = VT + ti;      // column-form results (component ti)
            double *const xrow = XT + tj;
            double *const xcol = XT + ti;
            // forward sweep y_k = b_k - L_k y_{k-1}, with the pivot products v_k = S_k^-1 y_k (off the
            // dependent chain) issued alongside; only v goes to LDS, y stays in registers.
            double yc = XT[ti], yr = 0.0;
            // right-hand sides are fetched two stages ahead of their use (the loads do not depend on the chain)
            double bq[3];
            bq[1] = XT[1 * 8 + tj];
            bq[2] = (NT >= 2) ? XT[2 * 8 + ti] : 0.0;
#pragma unroll
            for (int k = 1; k <= NT; ++k) {
                const double bk = bq[k % 3];
                if (k + 2 <= NT) bq[(k + 2) % 3] = ((k + 2) & 1) ? XT[(k + 2) * 8 + tj] : XT[(k + 2) * 8 + ti];
                if (k & 1) {
                    double pv = rS[k - 1] * yc, ch = rL[k] * yc;            // S symmetric: sum over ti -> row form
                    red_i2(ch, pv);
                    vrow[(k - 1) * 8] = pv;
                    yr = bk - ch;
                } else {
                    double pv = rS[k - 1] * yr, ch = rL[k] * yr;
                    red_j2(ch, pv);
                    vcol[(k - 1) * 8] = pv;
                    yc = bk - ch;
                }
            }
            if (NT & 1) vcol[NT * 8] = red_j(rS[NT] * yr);
            else        vrow[NT * 8] = red_i(rS[NT] * yc);
            sync();
            STAMP(1);
            // backward sweep x_k = v_k - L_{k+1}' x_{k+1}
            double xc = VT[NT * 8 + ti], xr = VT[NT * 8 + tj];
            if (NT & 1) xrow[NT * 8] = xr; else xcol[NT * 8] = xc;
            double vq[3];
            vq[(NT - 1) % 3] = ((NT - 1) & 1) ? VT[(NT - 1) * 8 + tj] : VT[(NT - 1) * 8 + ti];
            if (NT >= 2) vq[(NT - 2) % 3] = ((NT - 2) & 1) ? VT[(NT - 2) * 8 + tj] : VT[(NT - 2) * 8 + ti];
#pragma unroll
            for (int k = NT - 1; k >= 0; --k) {
                const double vk = vq[k % 3];
                if (k - 2 >= 0) vq[(k - 2) % 3] = ((k - 2) & 1) ? VT[(k - 2) * 8 + tj] : VT[(k - 2) * 8 + ti];
                if (((k + 1) & 1) == 0) { xr = vk - red_i(rL[k + 1] * xc); xrow[k * 8] = xr; }
                else                    { xc = vk - red_j(rL[k + 1] * xr); xcol[k * 8] = xc; }
            }
            sync();
            STAMP(2);
            return;
        }
        // (tail kernel: the polish solves -- wavefront 0 alone, its LDS traffic ordered wave-locally, the others wait at the end)
        auto lsync = [&]() { if constexpr (TAIL) wsync(); else sync(); };
        if (!TAIL || wv == 0) {
        // forward: y_k = b_k - L_k y_{k-1}
        double yc = XT[ti], yr = 0.0;
        for (int k = 1; k <= N; k += 2) {
            yr = XT[k * 8 + tj] - red_i(tL[k * kTS + lane] * yc);                    // odd k: tile holds L_k[tj][ti]
            if (ti == 0) XT[k * 8 + tj] = yr;
            if (k + 1 <= N) {
                yc = XT[(k + 1) * 8 + ti] - red_j(tL[(k + 1) * kTS + lane] * yr);   // even k: tile holds L_k[ti][tj]
                if (tj == 0) XT[(k + 1) * 8 + ti] = yc;
            }
        }
        lsync();
        // v_k = Sinv_k y_k
        for (int k = 0; k <= N; ++k) {
            const double t = red_j(tS[k * kTS + lane] * XT[k * 8 + tj]);
            if (tj == 0) VT[k * 8 + ti] = t;
        }
        lsync();
        // backward: x_k = v_k - L_{k+1}' x_{k+1}
        double xc = VT[N * 8 + ti], xr = VT[N * 8 + tj];
        if (lane < 8) XT[N * 8 + lane] = VT[N * 8 + lane];
        for (int k = N - 1; k >= 0; --k) {
            if (((k + 1) & 1) == 0) {       // L_{k+1}[ti][tj] * x_{k+1}[ti], summed over ti -> row form (k is odd)
                xr = VT[k * 8 + tj] - red_i(tL[(k + 1) * kTS + lane] * xc);
                if (ti == 0) XT[k * 8 + tj] = xr;
            } else {                        // tile holds L_{k+1}[tj][ti]; times x_{k+1}[tj], summed over tj -> column form
                xc = VT[k * 8 + ti] - red_j(tL[(k + 1) * kTS + lane] * xr);
                if (tj == 0) XT[k * 8 + ti] = xc;
            }
        }
        }
        sync();
    }

    // ---- residuals (OSQP update_info) -----------------------------------------------------------
    // pri / dua: residual norms; nAxz = max(|Ax|, |z|), nPAq = max(|Px|, |A'y|, |q|) (all that OSQP's tolerances need);
    // s_*: the same in the scaled problem (adaptive rho).  Maxima are exact, so merging them before the cross-lane
    // reductions changes nothing but the number of reductions (8 instead of 14).
    struct Res { double pri, dua, nAxz, nPAq, s_pri, s_dua, s_Axz, s_PAq; };

    // (xv, zd/zb, yd/yb) -> residual norms (the tail kernel leaves A x in ZT*, P x in VT, A'y in AT; the others form them per element)
    __device__ __forceinline__ Res residuals(const double *xv, const double *zd, const double *zb, const double *yd, const double *yb) {
        if constexpr (TAIL) {       // the three products side by side on different wavefronts
            const LaneC lc = lane_consts();
            eloop(0, kActW, [&](int e) { const int k = e >> 3; ZTd[e] = lc.rmask * (Eid(k, tj) * xv[e] - prev_stage_dot(k, xv)); ZTb[e] = Sb(k, tj) * xv[k * 8 + lc.bvar]; });
            eloop(kActW, kActW, [&](int e) { VT[e] = P_row<false>(e >> 3, xv); });
            eloop(2 * kActW, NW - 2 * kActW, [&](int e) { AT[e] = At_elem(e, yd, yb, lc); });
        }
        Res r = {0, 0, 0, 0, 0, 0, 0, 0};
        auto norms_of = [&](int e, double axd, double axb, double px, double aty) {
            const double eid = 1.0 / Ed[e], eib = 1.0 / Eb[e], di = 1.0 / D[e];
            const double rd = axd - zd[e], rb = axb - zb[e];
            r.s_pri = fmax(r.s_pri, fmax(fabs(rd), fabs(rb)));
            r.pri = fmax(r.pri, fmax(fabs(eid * rd), fabs(eib * rb)));
            r.s_Axz = fmax(r.s_Axz, fmax(fmax(fabs(axd), fabs(axb)), fmax(fabs(zd[e]), fabs(zb[e]))));
            r.nAxz = fmax(r.nAxz, fmax(fmax(fabs(eid * axd), fabs(eib * axb)), fmax(fabs(eid * zd[e]), fabs(eib * zb[e]))));
            const double dr = Qv[e] + px + aty;
            r.s_dua = fmax(r.s_dua, fabs(dr));       r.dua = fmax(r.dua, fabs(di * dr));
            r.s_PAq = fmax(r.s_PAq, fmax(fmax(fabs(px), fabs(aty)), fabs(Qv[e])));
            r.nPAq = fmax(r.nPAq, fmax(fmax(fabs(di * px), fabs(di * aty)), fabs(di * Qv[e])));
        };
        if constexpr (TAIL) {
            sync();
            for (int e = opaque(tid); e < NS * 8; e += kStride) norms_of(e, ZTd[e], ZTb[e], VT[e], AT[e]);
        } else {
            // (round 5: the three products of an element and its share of the norms in one loop -- they used to pass through the
            // element's slots of ZT*, VT and AT and a barrier; fenced, so that no product's last operation contracts with the norms'
            // first.  The products are no longer left in those arrays: nothing read them.)
            const LaneC lc = lane_consts();
            auto fence = [](double v) { asm volatile("" : "+v"(v)); return v; };
            for (int e = opaque(tid); e < NS * 8; e += kStride) {
                const int k = e >> 3;
                const double axd = fence(lc.rmask * (Eid(k, tj) * xv[e] - prev_stage_dot(k, xv)));
                const double axb = fence(Sb(k, tj) * xv[k * 8 + lc.bvar]);
                const double px = fence(P_row<false>(k, xv));
                const double aty = fence(At_elem(e, yd, yb, lc));
                norms_of(e, axd, axb, px, aty);
            }
        }
        double m_[8] = {r.pri, r.dua, r.nAxz, r.nPAq, r.s_pri, r.s_dua, r.s_Axz, r.s_PAq};
#pragma unroll
        for (int i = 0; i < 8; ++i) m_[i] = wave_max(m_[i]);
        if constexpr (kTwo) {
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) RED[wv * 16 + i] = m_[i];
            }
            sync();
#pragma unroll
            for (int i = 0; i < 8; ++i) m_[i] = fmax(RED[i], RED[16 + i]);
        }
        if constexpr (kFour) {
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) RED[80 + wv * 8 + i] = m_[i];
            }
            sync();
#pragma unroll
            for (int i = 0; i < 8; ++i) m_[i] = fmax(fmax(RED[80 + i], RED[88 + i]), fmax(RED[96 + i], RED[104 + i]));
        }
        if constexpr (TAIL) {
            if (lane == 0 && wv < kActW) {
#pragma unroll
                for (int i = 0; i < 8; ++i) RT[32 + wv * 8 + i] = m_[i];
            }
            sync();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                double v = RT[32 + i];
#pragma unroll
                for (int w = 1; w < kActW; ++w) v = fmax(v, RT[32 + w * 8 + i]);
                m_[i] = v;
            }
        }
        r.pri = m_[0]; r.dua = cinv * m_[1]; r.nAxz = m_[2]; r.nPAq = cinv * m_[3];
        r.s_pri = m_[4]; r.s_dua = m_[5]; r.s_Axz = m_[6]; r.s_PAq = m_[7];
        sync();
        return r;
    }
    __device__ __forceinline__ double objective(const double *xv) {
        P_mul(xv, VT);
        sync();
        if constexpr (kFixedSums) {     // (one summation order for the two forms of these kernels: see kFixedSums)
            for (int e = opaque(tid); e < NS * 8; e += kStride) AT[e] = xv[e] * (0.5 * VT[e] + Qv[e]);
            return ordered_total() * cinv;
        }
        double v = 0.0;
        for (int e = opaque(tid); e < NS * 8; e += kStride) v += xv[e] * (0.5 * VT[e] + Qv[e]);
        v = bsum<1>(v) * cinv;
        sync();
        return v;
    }

    // bounds of a dynamics row (equalities: l = u)
    __device__ __forceinline__ double dyn_bound(int e) const { return beq[e < 8 ? e : 8]; }

    // ---- infeasibility certificates (OSQP is_primal_infeasible / is_dual_infeasible) ----------------
    __device__ __forceinline__ bool primal_infeasible(double eps) {
        double nd = 0.0;
        for (int e = opaque(tid); e < NS * 8; e += kStride) {
            // dynamics rows have finite bounds: no projection.  box rows: project on the polar recession cone
            double dy = DYb[e];
            const double lo = Lo[e], hi = Hi[e];
            if (hi > kInfty * kMinScaling) { if (lo < -kInfty * kMinScaling) dy = 0.0; else dy = fmin(dy, 0.0); }
            else if (lo < -kInfty * kMinScaling) dy = fmax(dy, 0.0);
            DYb[e] = dy;
            nd = fmax(nd, fmax(fabs(Ed[e] * DYd[e]), fabs(Eb[e] * dy)));
        }
        nd = bmax<0>(nd);
        sync();
        bool res = false;
        if (nd > eps) {
            double lhs = 0.0;
            if constexpr (kFixedSums) {         // (the element's term with explicit fused operations: the same bits in both forms of the kernel)
                for (int e = opaque(tid); e < NS * 8; e += kStride) {
                    const double b = dyn_bound(e), dyd = DYd[e], dyb = DYb[e];
                    const double td = __builtin_fma(b, fmax(dyd, 0.0), b * fmin(dyd, 0.0)), tb = __builtin_fma(Hi[e], fmax(dyb, 0.0), Lo[e] * fmin(dyb, 0.0));
                    AT[e] = td + tb;
                }
                lhs = ordered_total();
            } else {
            for (int e = opaque(tid); e < NS * 8; e += kStride) {
                const double b = dyn_bound(e), dyd = DYd[e], dyb = DYb[e];
                lhs += b * fmax(dyd, 0.0) + b * fmin(dyd, 0.0);
                lhs += Hi[e] * fmax(dyb, 0.0) + Lo[e] * fmin(dyb, 0.0);
            }
            lhs = bsum<1>(lhs);
            }
            if (lhs < -eps * nd) {
                At_mul(DYd, DYb, AT);
                sync();
                double na = 0.0;
                for (int e = opaque(tid); e < NS * 8; e += kStride) na = fmax(na, fabs(AT[e] / D[e]));
                na = bmax<2>(na);
                res = na < eps * nd;
                sync();
            }
        }
        return res;
    }
    __device__ __forceinline__ bool dual_infeasible(double eps) {
        double nd = 0.0, qdx = 0.0;
        if constexpr (kFixedSums) {
            for (int e = opaque(tid); e < NS * 8; e += kStride) { nd = fmax(nd, fabs(D[e] * DX[e])); AT[e] = Qv[e] * DX[e]; }
            qdx = ordered_total(); nd = bmax<3>(nd);
        } else {
        for (int e = opaque(tid); e < NS * 8; e += kStride) { nd = fmax(nd, fabs(D[e] * DX[e])); qdx += Qv[e] * DX[e]; }
        { const SumMax qn = bsum_bmax<3>(qdx, nd); qdx = qn.s; nd = qn.m; }
        }
        bool res = false;
        if (nd > eps && qdx < -c * eps * nd) {
            P_mul(DX, VT);
            sync();
            double np = 0.0;
            for (int e = opaque(tid); e < NS * 8; e += kStride) np = fmax(np, fabs(VT[e] / D[e]));
            np = bmax<0>(np);
            sync();
            if (np < c * eps * nd) {
                A_mul(DX, ZTd, ZTb);
                sync();
                double bad = 0.0;
                for (int e = opaque(tid); e < NS * 8; e += kStride) {
                    const int k = e >> 3, r = e & 7;
                    if (r < NX) { const double v = ZTd[e] / Ed[e]; if (v > eps * nd || v < -eps * nd) bad = 1.0; }
                    if (r < nbox(k)) {
                        const double v = ZTb[e] / Eb[e];
                        if ((Hi[e] < kInfty * kMinScaling && v > eps * nd) || (Lo[e] > -kInfty * kMinScaling && v < -eps * nd)) bad = 1.0;
                    }
                }
                res = bmax<1>(bad) == 0.0;
                sync();
            }
        }
        return res;
    }

    // w rows for the next right-hand side: ZT = rho z - y   (kept in ZT* between iterations)
    __device__ __forceinline__ void recompute_w() {
        for (int e = opaque(tid); e < NS * 8; e += kStride) {
            ZTd[e] = rho_eq * Zd[e] - Yd[e];
            ZTb[e] = rho_of(Lo[e], Hi[e], rho) * Zb[e] - Yb[e];
        }
        sync();
    }
    // XT = sigma x - q + A' (rho z - y)        (OSQP compute_rhs, x part, reduced form)
    // (the tail kernel writes it to VT: dense_apply reads the whole right-hand side while it stores x~ into XT)
    template <bool SYNC = true>
    __device__ __forceinline__ void build_rhs(double sigma) {
        const LaneC lc = lane_consts();
        if constexpr (TAIL) {
            // slot-major, kRhsPitch doubles per slot: the thread (g, s) of dense_apply reads its 21 values rhs[8 c + s] as one contiguous
            // run (16-byte loads, half the LDS instructions of the broadcast reads at stride 8); the last slots spill a few words into AT,
            // which only the termination checks use
            for (int e = opaque(tid); e < NS * 8; e += kStride) VT[(e & 7) * kRhsPitch + (e >> 3)] = At_elem(e, ZTd, ZTb, lc) + (sigma * X[e] - Qv[e]);
        } else if constexpr (kUniScalars && kMf && kTwo && !kRegState) {
            // (round 5: the rounds unrolled from the plain thread index, like update(): their LDS addresses are loop invariants the
            // compiler may keep in registers now.  The kernels with the element state in registers -- kRegState -- run this form once in
            // 25 iterations: they take the rolled loop below, whose addresses are formed where they are used)
            constexpr int kRounds = ((NT + 1) * 8 + kStride - 1) / kStride;
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                const int e = tid + r * kStride;
                if (e < NS * 8) XT[e] = At_elem(e, ZTd, ZTb, lc) + (sigma * X[e] - Qv[e]);
            }
        } else {
            for (int e = opaque(tid); e < NS * 8; e += kStride) XT[e] = At_elem(e, ZTd, ZTb, lc) + (sigma * X[e] - Qv[e]);
        }
        if constexpr (SYNC) sync();
    }
    // NW == 4: the right-hand side of ONE element per thread (SET 1: every thread of the workgroup, SET 2: the inner wavefronts; see kS2a)
    template <int SET>
    __device__ __forceinline__ void build_rhs_set(double sigma) {
        const LaneC lc = lane_consts();
        const int t0 = tid;        // (round 5: not opaque any more -- with the uniform scalars in SGPRs and the post-loop addresses laundered the
                                   // four-wavefront kernels have the registers to let the element phases' addresses be hoisted: -10 % instructions per iteration)
        const int e = SET == 1 ? elem_set1(t0) : elem_set2(t0);
        if (e < NS * 8) XT[e] = At_elem(e, ZTd, ZTb, lc) + (sigma * X[e] - Qv[e]);
    }
    // One ADMM iteration of the four-wavefront kernel: set 1 of the right-hand side on all four wavefronts, the relayed KKT solve
    // with set 2 of the right-hand side and of the update on the inner wavefronts beside it (inner4), set 1 of the update.
    // (Measured and not taken: the right-hand side of a chain's first eight stages on its outer wavefront, fetched in front of B0 --
    // the fetch then waits behind the element phase's LDS traffic in front of the barrier instead of behind it: no gain.  The pivot
    // products v_p kept in registers from the forward to the backward steps of the same wavefront instead of going through VT (80 of
    // 404 LDS instructions per iteration-instance less): the two selects per step that zero the non-owner blocks cost more than the
    // store and the load did -- 2.72 -> 2.90 us per iteration alone on a CU, 3.65 -> 3.74 at two per CU; with the quad swap an odd
    // chain length needs, N = 30: 3.09 -> 3.45.  Pinning the two MFMAs of a step back to back, or the pivot product's closing add
    // between them: +8 % / +10 % per iteration.)
    __device__ __forceinline__ void iterate4(double sigma, double alpha, bool want_delta) {
        build_rhs_set<1>(sigma);
        const MfLane m = mf_lane();         // (in front of the barrier: the sweeps' lane constants form while the stores drain)
        sync();             // B0
        STAMP(0);
        relay4<true>(m, sigma, alpha, want_delta);
        update<1>(alpha, want_delta);
        sync();             // B5
        STAMP(5);
    }
    // ---- the element phases of the two-wavefront MFMA kernels with both rounds in one basic block (kFuse2) ----
    static constexpr bool kFuse2 = kUniScalars && MF && NW == 2 && kCacheW && kRnd == 2;
    // this wavefront owns elements in the second round (uniform; N = 20: wavefront 0 only)
    __device__ __forceinline__ bool fuse2_wave() const { return __builtin_amdgcn_readfirstlane(kStride + 64 * wv) < NS * 8; }
    // right-hand side element: At_elem (without the pinned-steering row) + sigma x - q, loads and arithmetic apart
    struct RhsIn { double c_[NX], d_[NX], sb0, sb1, w0, w1, ei, wd, x, q; };
    // (the fused operations written out: what the compiler contracts `a b + c d` into depends on the code around it -- with implicit
    // contraction the register form of the element phases differed from the LDS form in the last bits.  These are the operations of
    // round 5's build, read off its assembly; the LDS forms -- At_elem, update()'s element() -- compile to the same ones: tools/ab_equal.py)
    // the coupling sum of an element: sum_r [A|B]_k[r][j] (rho z_d - y_d)_{k+1}[r]
    __device__ __forceinline__ double rhs_acc(const RhsIn &r) const {
        double acc0 = __builtin_fma(r.c_[0], r.d_[0], r.c_[1] * r.d_[1]);
        const double acc1 = __builtin_fma(r.c_[2], r.d_[2], r.c_[3] * r.d_[3]);
#pragma unroll
        for (int i = 4; i < NX; ++i) acc0 = __builtin_fma(r.c_[i], r.d_[i], acc0);
        return acc1 + acc0;
    }
    __device__ __forceinline__ static double rhs_boxes(const RhsIn &r) { return __builtin_fma(r.sb0, r.w0, r.sb1 * r.w1); }
    __device__ __forceinline__ static double rhs_sxq(const RhsIn &r, double sigma) { return __builtin_fma(sigma, r.x, -r.q); }
    // (boxes + (Eid w_d - acc)) + (sigma x - q)
    __device__ __forceinline__ static double rhs_finish(double boxes, double ei, double wd, double acc, double sxq) {
        return (boxes + __builtin_fma(ei, wd, -acc)) + sxq;
    }
    __device__ __forceinline__ double rhs_form(const RhsIn &r, double sigma) const {
        return rhs_finish(rhs_boxes(r), r.ei, r.wd, rhs_acc(r), rhs_sxq(r, sigma));
    }
    // ---- element state in registers (round 6) ----------------------------------------------------------------------------------
    // The two element phases of an ADMM iteration pass five vectors through LDS that nobody but the element's own thread touches
    // between two termination checks: x, z and y of the element's dynamics row and box row (update() stores them, the next update()
    // loads them; the right-hand side loads x and the thread's own rho z_d - y_d once more) -- 12 of the 44 LDS accesses of an element
    // and iteration.  In the two-wavefront MFMA kernels with two element rounds (kFuse2) the thread keeps them in registers for the
    // whole ADMM loop: loaded once in front of it, read and written in place by the fused blocks below; ZT* still go to LDS every
    // iteration (the neighbours' right-hand sides gather them), and an iteration that is followed by a termination check or a rho
    // update also files the state and the deltas in the arrays, where the checks, the re-factorisation, parking and the polish read
    // them (update_s, `want`).  The operations and their order are those of update()'s element() and of At_elem, written with explicit
    // fused operations (rhs_form): every output word is unchanged (tools/ab_equal.py).
    // Handles with steeringDelay > 0 (a third box row in the right-hand side: a uniform branch the fused block does not carry) stay on
    // the LDS form of the phases.
    static constexpr bool kRegState = kFuse2;
    static_assert(!kFuse2 || kStride == 128, "element state in registers: the second round / second wavefront own elements 64 and up (el_load_s, BZ)");
    // the thread's elements in the fused element phases: e0, e1 (-1: none), e1c (a lane without a second element repeats its first one)
    struct Own { int e0, e1, e1c; };
    __device__ __forceinline__ Own own() const {
        Own o;
        o.e0 = tid; o.e1 = tid + kStride < NS * 8 ? tid + kStride : -1;
        o.e1c = o.e1 >= 0 ? o.e1 : o.e0;
        return o;
    }
    __device__ __forceinline__ bool two_elems() const { return fuse2_wave(); }
    // (z_d is not part of it: the projection of a dynamics row is its bound, z_d == b after any update; the first iteration of a launch
    // takes it from the array -- cold start z = 0, warm start z = A x)
    struct ElState { double x, yd, zb, yb, wtd; };
    ElState es_[kRegState ? 2 : 1];
    __device__ __forceinline__ void state_load() {
        const Own o = own();
        es_[0] = ElState{X[o.e0], Yd[o.e0], Zb[o.e0], Yb[o.e0], ZTd[o.e0]};
        if (two_elems()) es_[1] = ElState{X[o.e1c], Yd[o.e1c], Zb[o.e1c], Yb[o.e1c], ZTd[o.e1c]};     // (no second element: a copy of the first one, kept equal by equal arithmetic)
    }
    // rho z_d - y_d again from the array (behind recompute_w: the check block may have changed rho)
    __device__ __forceinline__ void state_reload_w() {
        const Own o = own();
        es_[0].wtd = ZTd[o.e0];
        if (two_elems()) es_[1].wtd = ZTd[o.e1c];
    }
    // the state to the arrays (a loop that ends on a plain iteration; z_d = b behind any update)
    __device__ __forceinline__ void state_flush() {
        const Own o = own();
        X[o.e0] = es_[0].x; Zd[o.e0] = dyn_bound(o.e0); Yd[o.e0] = es_[0].yd; Zb[o.e0] = es_[0].zb; Yb[o.e0] = es_[0].yb;
        if (two_elems()) { X[o.e1c] = es_[1].x; Zd[o.e1c] = dyn_bound(o.e1c); Yd[o.e1c] = es_[1].yd; Zb[o.e1c] = es_[1].zb; Yb[o.e1c] = es_[1].yb; }
    }
    // right-hand side of the thread's elements from the state (rhs_load without the loads of X[e] and ZTd[e])
    __device__ __forceinline__ RhsIn rhs_load_s(int e, const LaneC &lc, const ElState &st) const {
        RhsIn r;
        const int k = e >> 3, kn = k < N ? k + 1 : N;         // the stage-N tile is all zero
        const double *col = tA + k * kTS + tj, *sd = ZTd + kn * 8;
#pragma unroll
        for (int i = 0; i < NX; ++i) { r.c_[i] = col[i * 8]; r.d_[i] = sd[i]; }
        r.sb0 = Sb(k, lc.r0); r.sb1 = Sb(k, lc.r1); r.w0 = ZTb[k * 8 + lc.r0]; r.w1 = ZTb[k * 8 + lc.r1]; r.ei = Eid(k, tj); r.wd = st.wtd;
        r.x = st.x; r.q = Qv[e];
        return r;
    }
    __device__ __forceinline__ void rhs_s(double sigma) {
        const LaneC lc = lane_consts();
        const Own o = own();
        if (two_elems()) {
            const RhsIn a0 = rhs_load_s(o.e0, lc, es_[0]), a1 = rhs_load_s(o.e1c, lc, es_[1]);
            const double v0 = rhs_form(a0, sigma), v1 = rhs_form(a1, sigma);
            XT[o.e0] = v0;
            XT[o.e1c] = v1;
        } else {
            const RhsIn a0 = rhs_load_s(o.e0, lc, es_[0]);
            XT[o.e0] = rhs_form(a0, sigma);
        }
    }
    // update of the thread's elements on the state (update()'s element() without the state's loads and stores)
    struct ElInS { double xt, zd, lo, hi, b, sb, xv, ei, dot; };
    struct ElOutS { double wtb, znd, dyd, dyb, dx; };
    // FIRST: the first iteration of a launch -- z_d of the dynamics row comes from the array (see ElState).  BZ: the element's stage is
    // known to be >= 1, whose dynamics rows have the bound 0 (beq[8], an exact +0: only stage 0 carries x0) -- no load
    template <bool FIRST, bool BZ>
    __device__ __forceinline__ ElInS el_load_s(int e, int bvar) const {
        const int k = e >> 3;
        ElInS q;
        q.xt = XT[e]; q.zd = FIRST ? Zd[e] : 0.0; q.lo = Lo[e]; q.hi = Hi[e];
        q.b = BZ ? 0.0 : dyn_bound(e); q.sb = Sb(k, tj); q.xv = XT[k * 8 + bvar]; q.ei = Eid(k, tj);
        const double d0 = AT[e], d1 = kRawV ? VT[e] : 0.0;
        const double dsum = kRawV ? d0 + d1 : d0;
        q.dot = k > 0 ? dsum : 0.0;
        return q;
    }
    // the state is advanced in place; first: z_d of the dynamics row comes from the array (else it is the row's bound: see ElState)
    __device__ __forceinline__ ElOutS el_form_s(const ElInS &q, ElState &st, bool first, double alpha, double oma, double rmask, double w, double winv) const {
        // (explicit fused operations: see rhs_form)
        ElOutS o;
        const double zd = first ? q.zd : q.b;
        const double ztd = rmask * __builtin_fma(q.ei, q.xt, -q.dot);
        const double zrd = __builtin_fma(oma, zd, alpha * ztd);
        const double znd = q.b;
        const double dd = zrd - znd;
        o.dyd = rho_eq * dd;
        const double ynd = __builtin_fma(rho_eq, dd, st.yd);
        const double zrb = __builtin_fma(oma, st.zb, alpha * (q.sb * q.xv));
        const double znb = clipd(__builtin_fma(winv, st.yb, zrb), q.lo, q.hi);
        const double db = zrb - znb;
        o.dyb = w * db;
        const double ynb = __builtin_fma(w, db, st.yb);
        const double xn = __builtin_fma(alpha, q.xt, oma * st.x);
        o.dx = xn - st.x;
        st.x = xn;
        st.yd = ynd; st.wtd = __builtin_fma(rho_eq, znd, -ynd);
        st.yb = ynb; st.zb = znb;
        o.znd = znd;
        o.wtb = __builtin_fma(w, znb, -ynb);
        return o;
    }
    struct WW { double w, wi; };
    __device__ __forceinline__ WW weights_of(int r) const {
        const int c_ = (opaque(wcls) >> (2 * r)) & 3;
        WW q;
        q.w = c_ == 0 ? rho : (c_ == 1 ? rho_eq : kRhoMin);
        q.wi = c_ == 0 ? rinv : (c_ == 1 ? rinv_eq : 1.0 / kRhoMin);
        return q;
    }
    // want: a termination check or a rho update follows -- the state and the deltas go to the arrays as well
    template <bool FIRST>
    __device__ __forceinline__ void update_s(double alpha, double oma, bool want) {
        const LaneC lc = lane_consts();
        const int bvar = lc.bvar;
        const double rmask = lc.rmask;
        const Own o = own();
        auto file = [&](int e, const ElState &st, const ElOutS &q) {
            X[e] = st.x; Zd[e] = q.znd; Yd[e] = st.yd; Zb[e] = st.zb; Yb[e] = st.yb;
            DYd[e] = q.dyd; DYb[e] = q.dyb; DX[e] = q.dx;
        };
        if (two_elems()) {
            // (the second element sits in a stage >= 16: bound 0 -- a lane without one repeats its first element, bound included)
            ElInS a0 = el_load_s<FIRST, false>(o.e0, bvar), a1 = el_load_s<FIRST, true>(o.e1c, bvar);
            a1.b = o.e1 >= 0 ? 0.0 : a0.b;
            const WW w0 = weights_of(0), w1 = weights_of(1);
            const ElOutS o0 = el_form_s(a0, es_[0], FIRST, alpha, oma, rmask, w0.w, w0.wi), o1 = el_form_s(a1, es_[1], FIRST, alpha, oma, rmask, w1.w, w1.wi);
            ZTd[o.e0] = es_[0].wtd; ZTb[o.e0] = o0.wtb;
            ZTd[o.e1c] = es_[1].wtd; ZTb[o.e1c] = o1.wtb;
            if (want) { file(o.e0, es_[0], o0); file(o.e1c, es_[1], o1); }
            asm volatile("; LPVMPC_ROLE_BEGIN regstate_update");
            sync();
            asm volatile("; LPVMPC_ROLE_END regstate_update");
        } else {
            const ElInS a0 = el_load_s<FIRST, true>(o.e0, bvar);        // (a wavefront without a second round owns elements 64 and up: stages >= 8)
            const WW w0 = weights_of(0);
            const ElOutS o0 = el_form_s(a0, es_[0], FIRST, alpha, oma, rmask, w0.w, w0.wi);
            ZTd[o.e0] = es_[0].wtd; ZTb[o.e0] = o0.wtb;
            if (want) file(o.e0, es_[0], o0);
            asm volatile("; LPVMPC_ROLE_BEGIN regstate_update");
            sync();
            asm volatile("; LPVMPC_ROLE_END regstate_update");
        }
    }

    // z~ = A x~ fused with OSQP update_x / update_z / update_y; leaves rho z - y in ZT*.
    // All LDS reads of a round are issued before any of its writes.
    // (SET: four-wavefront kernel only -- 1 / 2 = that element set, no barrier; see iterate4)
    template <int SET = 0>
    __device__ __forceinline__ void update(double alpha, bool want_delta) {
        const double oma = 1.0 - alpha;
        const LaneC lc = lane_consts();
        const int bvar = lc.bvar;
        const double rmask = lc.rmask;
        auto element = [&](int e, bool cached, double wc, double wic) {
            const int k = e >> 3;
            const double xt = XT[e], xo = X[e];
            const double zd = Zd[e], yd = Yd[e], zb = Zb[e], yb = Yb[e], lo = Lo[e], hi = Hi[e];
            const double b = dyn_bound(e), sb = Sb(k, tj), xv = XT[k * 8 + bvar], ei = Eid(k, tj);
            // previous stage's dynamics-row product: left in AT by the backward MFMA sweep, or formed here
            const double dot = kMf ? (k > 0 ? (kRawV ? AT[e] + VT[e] : AT[e]) : 0.0) : prev_stage_dot(k, XT);     // (kRawV: the two partial sums of the backward sweep)
            // dynamics row (k, tj): bounds l = u = b
            const double ztd = rmask * (ei * xt - dot);
            const double zrd = alpha * ztd + oma * zd;
            const double znd = b;           // the projection of zrd + yd / rho on the one-point set [b, b]
            const double dyd = rho_eq * (zrd - znd), ynd = yd + dyd;
            // box row (k, tj)
            double w = wc, winv = wic;
            if (!cached) {
                const bool loose = lo < -kInfty * kMinScaling && hi > kInfty * kMinScaling, eq = hi - lo < kRhoTol;
                w = loose ? kRhoMin : (eq ? rho_eq : rho); winv = loose ? 1.0 / kRhoMin : (eq ? rinv_eq : rinv);
            }
            const double zrb = alpha * (sb * xv) + oma * zb;
            const double znb = clipd(zrb + winv * yb, lo, hi);
            const double dyb = w * (zrb - znb), ynb = yb + dyb;
            const double xn = alpha * xt + oma * xo;
            Yd[e] = ynd; Zd[e] = znd; ZTd[e] = rho_eq * znd - ynd;
            Yb[e] = ynb; Zb[e] = znb; ZTb[e] = w * znb - ynb;
            X[e] = xn;
            if (want_delta) { DYd[e] = dyd; DYb[e] = dyb; DX[e] = xn - xo; }
        };
        if constexpr (kFour) {
            const int t0 = tid;
            if constexpr (SET != 2) { const int e = elem_set1(t0); if (e < NS * 8) element(e, true, wbx[0], wbxi[0]); }
            if constexpr (SET != 1) { const int e = elem_set2(t0); if (e < NS * 8) element(e, true, wbx[1], wbxi[1]); }
            if constexpr (SET != 0) return;
        } else if constexpr (kCacheW) {
            // (round 5: with the uniform scalars in SGPRs and the post-loop addresses laundered, the two-wavefront MFMA kernels have the
            // registers to let this loop's addresses be hoisted: -24 instructions per iteration in the headline kernel)
            const int t0 = (kUniScalars && kMf && kTwo && !kRegState) ? tid : opaque(tid);      // (kRegState: this form runs once in 25 iterations -- no hoisted addresses for it)
#pragma unroll
            for (int r = 0; r < kRnd; ++r) {
                const int e = t0 + r * kStride;
                if (e < NS * 8) element(e, true, wbx[r], wbxi[r]);
            }
        } else {
            for (int e = opaque(tid); e < NS * 8; e += kStride) element(e, false, 0.0, 0.0);
        }
        sync();
    }

    // ================================================================================================================
    // Tail kernel, round 5: the ADMM iteration in two phases, termination checks off the chain
    // ================================================================================================================
    // An iteration of the tail kernel was three phases -- right-hand side, dense product, update -- of which the two element phases
    // (a few dozen instructions each) cost 1700 of 2900 cycles: barrier, address arithmetic, an LDS round trip, a short dependent
    // chain, a store, and the same again.  What forces two element phases is the dynamics rows: the update forms z~_d = A_d x~ (needs
    // x~ of the previous stage), the next right-hand side A_d' (rho z_d - y_d) (needs the rows of the next stage).  The equality rows
    // are LINEAR -- z_d = b after the first iteration, y_d+ = y_d + rho_eq alpha (A_d x~ - b) -- so their share of the right-hand side,
    //     G = A_d' (rho_eq b - y_d) - q,       G+ = G - rho_eq alpha (T x~ - c0),    T = A_d' A_d (block tridiagonal),  c0 = A_d' b,
    // can be carried in VARIABLE space: T x~ for variable (k, j) needs x~ of the stages k-1, k, k+1, which the dense product has just
    // delivered, and nothing another thread computes in this phase.  The box rows of a variable are handled by the variable's own
    // thread (every box row acts on exactly one variable), so the next right-hand side
    //     rhs+ = sigma x+ + G+ + sum_r Sb_r (w_r z_r+ - y_r+)
    // is complete in the thread: ONE element phase, two barriers per iteration.  x, G and the box rows' (z, y) live in registers; y_d is
    // not tracked per iteration (nothing reads it between checks): S = sum of x~ since the last snapshot gives
    // y_d(now) = y_d(snapshot) + rho_eq alpha (A_d S - n b) where it is needed.
    //
    // Termination checks: the last wavefront holds no rows of K^-1 and owns no element.  At a check iteration the element threads file a
    // SNAPSHOT (x, z_b, y_b -> the image's own X, Zb, Yb arrays, which nothing else writes in this loop; x~ and S -> ZTd, ZTb; the
    // deltas -> DX, DYb) and go on iterating; the last wavefront evaluates OSQP's update_info / check_termination / adaptive-rho rule
    // on the snapshot, one 64-element round of one operator per phase slot (checker_tick), wave-locally, with the same formulas and
    // the same summation order as residuals() / check_termination(), and publishes a verdict a few iterations later:
    //   continue           -> nothing happens (the iterations that ran meanwhile are the ones that would have run anyway),
    //   stop (any status)  -> the arrays still hold iteration k: the solve ends there, iteration count k,
    //   new rho            -> back to the snapshot: registers from the arrays, re-factorisation, on from iteration k + 1.
    // Every result is what the check-then-iterate order gives; only the clock differs.
    static constexpr int CK_IDLE = 0, CK_YD = 1, CK_YDX = 17, CK_RESD = 18, CK_AX = 2, CK_PX = 3, CK_AT = 4, CK_RES = 5, CK_RR0 = 6, CK_RR1 = 7, CK_DEC = 8,
                         CK_PI1 = 9, CK_PI2 = 10, CK_PI3 = 11, CK_DI0 = 12, CK_DI1 = 13, CK_DI2 = 14, CK_DI3 = 15, CK_FIN = 16;
    __device__ __forceinline__ static bool uni(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }
    __device__ __forceinline__ static double unid(double v) {
        const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
        return __hiloint2double(hi, lo);
    }
    // registers of the element threads from the image's arrays (X, Zb, Yb, Yd, Zd = b), the first right-hand side
    __device__ __forceinline__ void tail_begin(double sigma) {
        recompute_w();                          // ZT* = rho z - y
        build_rhs(sigma);                       // slot-major, into VT (with its barrier)
        const LaneC lc = lane_consts();
        const int e = opaque(tid);
        if (e < NS * 8) {
            const double2 *tr = reinterpret_cast<const double2 *>(TT + e * 24);
#pragma unroll
            for (int h = 0; h < 12; ++h) { const double2 v = tr[h]; dm[kTR + 2 * h] = v.x; dm[kTR + 2 * h + 1] = v.y; }
#pragma unroll
            for (int i = kBX; i < kXS + 4; ++i) dm[i] = 0.0;
            const int k = e >> 3, kn = k < N ? k + 1 : N;
            const double *col = tA + k * kTS + tj, *sd = ZTd + kn * 8;
            double c_[NX], d_[NX];
#pragma unroll
            for (int r = 0; r < NX; ++r) { c_[r] = col[r * 8]; d_[r] = sd[r]; }
            double acc0 = c_[0] * d_[0] + c_[1] * d_[1], acc1 = c_[2] * d_[2] + c_[3] * d_[3];
#pragma unroll
            for (int r = 4; r < NX; ++r) acc0 += c_[r] * d_[r];
            dm[kXS + 1] = (Eid(k, tj) * ZTd[e] - (acc0 + acc1)) - Qv[e];
            dm[kXS] = X[e];
            dm[kXS + 2] = 0.0;
            dm[kXS + 3] = (k == 0 && tj < NX) ? Eid(0, tj) * beq[tj] : 0.0;
#pragma unroll
            for (int q = 0; q < kBR; ++q) {
                const int eb = k * 8 + (q == 0 ? lc.r0 : lc.r1);
                const double lo = Lo[eb], hi = Hi[eb];
                const bool loose = lo < -kInfty * kMinScaling && hi > kInfty * kMinScaling, eq = hi - lo < kRhoTol;
                double *b_ = dm + kBX + 7 * q;
                b_[0] = Zb[eb]; b_[1] = Yb[eb]; b_[2] = lo; b_[3] = hi; b_[4] = Sb(k, q == 0 ? lc.r0 : lc.r1);
                b_[5] = loose ? kRhoMin : (eq ? rho_eq : rho); b_[6] = loose ? 1.0 / kRhoMin : (eq ? rinv_eq : rinv);
            }
        }
        sync();                                 // ZTd / ZTb are the snapshot's x~ / S from here on
    }
    // the fused element phase: x+, G+, the variable's box rows, the next right-hand side; want: also file the snapshot and the deltas
    __device__ __forceinline__ void tail_fused(double sigma, double alpha, bool want) {
        const int e = tid;
        if (e < NS * 8) {
            const int k = e >> 3, km = k > 0 ? k - 1 : 0, kp = k < N ? k + 1 : N;
            const double2 *xa = reinterpret_cast<const double2 *>(XT + km * 8), *xb = reinterpret_cast<const double2 *>(XT + k * 8),
                          *xc = reinterpret_cast<const double2 *>(XT + kp * 8);
            const double xt = XT[e];
            // (x~ of the three stages in two batches -- all 24 values in flight at once would not fit beside the 8 NS registers of `dm`)
            double2 ua[4], ub[4], uc[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) { ua[h] = xa[h]; ub[h] = xb[h]; }
            __builtin_amdgcn_sched_barrier(0);
            STAMP4(4);
            double sblk[3];
            auto blockdot = [&](const double *tt, const double2 *uu) {      // (the row of T lives in the registers dm[kTR ..]: down, diagonal, up block)
                double p = tt[0] * uu[0].x;        // one chain of fused multiply-adds per block (three independent chains in flight)
                p = fma(tt[1], uu[0].y, p); p = fma(tt[2], uu[1].x, p); p = fma(tt[3], uu[1].y, p);
                p = fma(tt[4], uu[2].x, p); p = fma(tt[5], uu[2].y, p); p = fma(tt[6], uu[3].x, p); p = fma(tt[7], uu[3].y, p);
                return p;
            };
            sblk[0] = blockdot(dm + kTR, ua);
#pragma unroll
            for (int h = 0; h < 4; ++h) uc[h] = xc[h];
            __builtin_amdgcn_sched_barrier(0);
            sblk[1] = blockdot(dm + kTR + 8, ub);
            sblk[2] = blockdot(dm + kTR + 16, uc);
            const double tx = (sblk[0] + sblk[2]) + sblk[1];
            const double oma = 1.0 - alpha;
            const double xr = dm[kXS], gr = dm[kXS + 1], c0r = dm[kXS + 3];
            const double xn = alpha * xt + oma * xr;
            const double gn = gr - (rho_eq * alpha) * (tx - c0r);
            double boxes = 0.0;
            const LaneC lc = lane_consts();
#pragma unroll
            for (int q = 0; q < kBR; ++q) {
                double *b_ = dm + kBX + 7 * q;    // z, y, lo, hi, sb, w, 1/w
                const double zrb = alpha * (b_[4] * xt) + oma * b_[0];
                const double znb = clipd(zrb + b_[6] * b_[1], b_[2], b_[3]);
                const double dyb = b_[5] * (zrb - znb), ynb = b_[1] + dyb;
                boxes += b_[4] * (b_[5] * znb - ynb);
                b_[0] = znb; b_[1] = ynb;
                if (want) { const int eb = k * 8 + (q == 0 ? lc.r0 : lc.r1); Zb[eb] = znb; Yb[eb] = ynb; DYb[eb] = dyb; }
            }
            // (handles with steeringDelay > 0 -- a third box row on delta in the first stages, CTRL:518-527 -- do not take this kernel: launch_solve)
            VT[(e & 7) * kRhsPitch + (e >> 3)] = boxes + (sigma * xn + gn);
            STAMP4(5);
            double sr = dm[kXS + 2] + xt;
            if (want) { X[e] = xn; DX[e] = xn - xr; ZTd[e] = xt; ZTb[e] = sr; sr = 0.0; }
            dm[kXS] = xn; dm[kXS + 1] = gn; dm[kXS + 2] = sr;
        }
    }
    // ---- the loop (both sides) ---------------------------------------------------------------------------------------------
    struct TailCtl { int iter, to_chk, to_adp, status; double rn; };
    // One TRIP = [dense-product slot] barrier [fused-phase slot] barrier.  The element side (CHK = false) runs an ADMM iteration in a
    // trip, or nothing while a pending verdict is waited for; the checker side (CHK = true) runs one checker step per slot.  The verdict
    // word is written in a dense-product slot and read by every thread in the following fused-phase slot.
    // Returns 0: the solve's ADMM phase has ended (t.iter, t.status; the residual norms stay in RT[84..91]); 2: back to the snapshot of iteration t.iter with the new
    // rho t.rn (the caller re-factors and re-enters); 3: as 0, and the closing check asked for the new rho t.rn (reported, not used).
    template <bool CHK>
    __device__ __forceinline__ int tail_loop(TailCtl &t, double sigma, double alpha, int max_iter, int chk_every, int adp_every) {
        int inflight = 0, k_to_chk = 0, k_to_adp = 0, n_acc = 0;      // iteration of the check in flight (0: none), the loop counters behind it, iterations summed in S
        int vk = -1;
        if constexpr (CHK) launder_ids();       // (the checker's addresses are formed on its side of the role branch, not in front of it)
        bool have_it = false, chk = false, adapt = false, want = false;
        for (;;) {
            if (inflight != 0 && vk == inflight) {          // the verdict of the check in flight has arrived
                const int code = __builtin_amdgcn_readfirstlane((int)RT[80]), kk = inflight;
                inflight = 0;
                if (code == 2) { t.status = __builtin_amdgcn_readfirstlane((int)RT[81]); t.iter = kk; return 0; }
                if (code == 3) {
                    t.rn = unid(RT[82]);
                    if (kk >= max_iter) { t.iter = kk; return 3; }
                    t.iter = kk; t.to_chk = k_to_chk; t.to_adp = k_to_adp;
                    return 2;
                }
            }
            if (!have_it) {
                if (t.iter >= max_iter) { if (inflight == 0) return 0; }        // (else: trips without work until the closing verdict)
                else {
                    ++t.iter;
                    chk = chk_every > 0 && --t.to_chk == 0;
                    adapt = adp_every > 0 && --t.to_adp == 0;
                    if (chk) t.to_chk = chk_every;
                    if (adapt) t.to_adp = adp_every;
                    if (t.iter == max_iter) chk = true;     // OSQP's closing update_info / check_termination behind the loop
                    want = chk || adapt;
                    have_it = true;
                }
            }
            // (a check is due while another is in flight -- a check interval shorter than a check's flight: the pending one first)
            const bool work = have_it && !(want && inflight != 0);
            // plain iterations that follow this one before the next check / rho update / closing iteration: they run in the tight loop
            // below, which only a verdict's arrival leaves early
            int more = 0;
            if (work && !want) {
                int m = max_iter - t.iter;
                if (chk_every > 0 && t.to_chk < m) m = t.to_chk;
                if (adp_every > 0 && t.to_adp < m) m = t.to_adp;
                more = m - 1;
            }
            for (;;) {
#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS != 5
                tlast = __builtin_amdgcn_s_memtime();
#endif
                // (the markers of tools/check_kernel_resources.py bracket the element side's trip; the checker's side is off the chain)
                if constexpr (!CHK) asm volatile("; LPVMPC_HOT_BEGIN");
                if constexpr (CHK) {
                    asm volatile("; LPVMPC_ROLE_BEGIN tail_trip x2");      // (this side's loop body runs twice per trip)
#pragma unroll 1
                    for (int sl = 0; sl < 2; ++sl) {
#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS == 4
                        const int st0_ = cs;
                        const unsigned long long tk0_ = __builtin_amdgcn_s_memtime();
                        checker_tick(sl == 0);
                        asm volatile("s_waitcnt lgkmcnt(0)");
                        if (st0_ != CK_IDLE && lane == 0) { CK[64 * NS + 512 + 2 * st0_] += (double)(__builtin_amdgcn_s_memtime() - tk0_); CK[64 * NS + 512 + 2 * st0_ + 1] += 1.0; }
#else
                        checker_tick(sl == 0);
#endif
                        sync();
                        if (sl == 0) vk = __builtin_amdgcn_readfirstlane((int)RT[83]);
                    }
                    asm volatile("; LPVMPC_ROLE_END tail_trip");
                } else {
                    asm volatile("; LPVMPC_ROLE_BEGIN tail_trip");
                    if (work) dense_apply();
                    sync();
                    STAMP(0);
                    const double vkd = RT[83];      // (consumed behind the element phase: its round trip is not on the chain)
                    if (work) tail_fused(sigma, alpha, want);
                    vk = __builtin_amdgcn_readfirstlane((int)vkd);
                    sync();
                    STAMP(1);
                    asm volatile("; LPVMPC_ROLE_END tail_trip");
                }
                if constexpr (!CHK) asm volatile("; LPVMPC_HOT_END");
                if (work) ++n_acc;
                if (more <= 0 || (inflight != 0 && vk == inflight)) break;
                --more; ++t.iter; --t.to_chk; --t.to_adp;
            }
            if (work) {
                if (want) {
                    inflight = t.iter; k_to_chk = t.to_chk; k_to_adp = t.to_adp;
                    if constexpr (CHK) { ckk = t.iter; ckn = n_acc; ckc = chk; cka = adapt; cs = CK_YD; cr = 0; }
                    n_acc = 0;
                }
                have_it = false;
            }
        }
    }
    // ---- the checker: one step per phase slot on the last wavefront ----------------------------------------------------
    // scratch: CK[0 .. 8NS) A x (dynamics rows), [8NS ..) A x (box rows), [16NS ..) P x, [24NS ..) A'y, [32NS ..) per-lane maxima [8][64], behind them
    // 1 / Ed, 1 / Eb, 1 / D (dense_build);
    // RT[56..58] the three rounds' partial sums, RT[59] nd, RT[60] flags (prc | drc << 1 | pic << 2 | dic << 3), RT[64..71] the eight
    // norms of Res; the verdict: RT[80] code (1 continue, 2 stop, 3 new rho), RT[81] status, RT[82] new rho, RT[83] iteration of the
    // check it belongs to (written last; the other wavefronts compare it with the check they are waiting for), RT[84..91] Res
    __device__ __forceinline__ void ck_next_round(int next_state) { if (cr == 2) { cr = 0; cs = next_state; } else ++cr; }
    __device__ __forceinline__ void ck_max_acc(int slot, int ln, double v) {
        double *m = CK + 32 * NS + slot * 64 + ln;
        *m = cr == 0 ? v : fmax(*m, v);
    }
    __device__ __forceinline__ void checker_tick(bool slot_d) {
        if (cs == CK_IDLE) return;
        const int ln = opaque(lane);
        const int e_ = ln + 64 * cr;
        const bool act = e_ < NS * 8;
        const int e = act ? e_ : NS * 8 - 1;             // (clamped: valid addresses, results masked)
        const int k = e >> 3;
        const double eps_p = ck_eps_p, eps_d = ck_eps_d;
        if (cs == CK_YD) {
            // y_d at the snapshot from the sum of x~ since the snapshot before (ZTb): y_d += rho_eq alpha (A_d S - n b)
            const LaneC lc = lane_consts();
            const double b = dyn_bound(e), ei = Eid(k, tj);
            const double us = lc.rmask * (ei * ZTb[e] - prev_stage_dot(k, ZTb));
            const double yd = Yd[e] + (rho_eq * ck_alpha) * (us - (double)ckn * b);
            if (act) Yd[e] = yd;
            ck_next_round(CK_YDX);
        } else if (cs == CK_YDX) {
            // delta y_d of the check iteration from its x~ (ZTd), as update() forms it (z_d = b)
            const LaneC lc = lane_consts();
            const double b = dyn_bound(e), ei = Eid(k, tj);
            const double ux = lc.rmask * (ei * ZTd[e] - prev_stage_dot(k, ZTd));
            const double zrd = ck_alpha * ux + (1.0 - ck_alpha) * b;
            if (act) DYd[e] = rho_eq * (zrd - b);
            ck_next_round(CK_AX);
        } else if (cs == CK_AX) {
            const LaneC lc = lane_consts();
            const double zd = lc.rmask * (Eid(k, tj) * X[e] - prev_stage_dot(k, X)), zb = Sb(k, tj) * X[k * 8 + lc.bvar];
            if (act) { CK[e] = zd; CK[8 * NS + e] = zb; }
            ck_next_round(CK_PX);
        } else if (cs == CK_PX) {
            const double pv = P_row<false>(k, X);
            if (act) CK[16 * NS + e] = pv;
            ck_next_round(CK_AT);
        } else if (cs == CK_AT) {
            const LaneC lc = lane_consts();
            const double av = At_elem(e, Yd, Yb, lc);
            if (act) CK[24 * NS + e] = av;
            ck_next_round(CK_RES);
        } else if (cs == CK_RES) {
            // the element loop of residuals(), maxima per lane: the four primal norms ...
            const double eid = CK[40 * NS + 512 + e], eib = CK[48 * NS + 512 + e];      // 1 / Ed, 1 / Eb (dense_build)
            const double axd = CK[e], axb = CK[8 * NS + e];
            const double zd = Zd[e], zb = Zb[e];
            const double rd = axd - zd, rb = axb - zb;
            ck_max_acc(4, ln, act ? fmax(fabs(rd), fabs(rb)) : 0.0);
            ck_max_acc(0, ln, act ? fmax(fabs(eid * rd), fabs(eib * rb)) : 0.0);
            ck_max_acc(6, ln, act ? fmax(fmax(fabs(axd), fabs(axb)), fmax(fabs(zd), fabs(zb))) : 0.0);
            ck_max_acc(2, ln, act ? fmax(fmax(fabs(eid * axd), fabs(eib * axb)), fmax(fabs(eid * zd), fabs(eib * zb))) : 0.0);
            ck_next_round(CK_RESD);
        } else if (cs == CK_RESD) {
            // ... and the four dual ones
            const double di = CK[56 * NS + 512 + e], px = CK[16 * NS + e], aty = CK[24 * NS + e], q = Qv[e];
            const double dr = q + px + aty;
            ck_max_acc(5, ln, act ? fabs(dr) : 0.0);
            ck_max_acc(1, ln, act ? fabs(di * dr) : 0.0);
            ck_max_acc(7, ln, act ? fmax(fmax(fabs(px), fabs(aty)), fabs(q)) : 0.0);
            ck_max_acc(3, ln, act ? fmax(fmax(fabs(di * px), fabs(di * aty)), fabs(di * q)) : 0.0);
            ck_next_round(CK_RR0);
        } else if (cs == CK_RR0) {
            // the eight cross-lane maxima, two per step (cr = 0 .. 3)
            const double v0 = wave_max(CK[32 * NS + (2 * cr) * 64 + ln]), v1 = wave_max(CK[32 * NS + (2 * cr + 1) * 64 + ln]);
            if (ln == 0) { RT[64 + 2 * cr] = v0; RT[64 + 2 * cr + 1] = v1; }
            if (cr == 3) { cr = 0; cs = CK_DEC; } else ++cr;
        } else if (cs == CK_DEC) {
            const double pri = unid(RT[64]), dua = cinv * unid(RT[65]), nAxz = unid(RT[66]), nPAq = cinv * unid(RT[67]);
            int fl = 0, nxt = CK_FIN;
            if (ckc && !(pri > kInfty || dua > kInfty)) {
                const bool prc = pri < ck_eps_abs + ck_eps_rel * nAxz, drc = dua < ck_eps_abs + ck_eps_rel * nPAq;
                fl = (prc ? 1 : 0) | (drc ? 2 : 0);
                nxt = !prc ? CK_PI1 : (!drc ? CK_DI1 : CK_FIN);
            }
            if (ln == 0) RT[60] = (double)fl;
            cs = __builtin_amdgcn_readfirstlane(nxt);
        } else if (cs == CK_PI1) {
            // is_primal_infeasible, first loop: delta_y projected on the polar of the recession cone, its norm
            double dy = DYb[e];
            const double lo = Lo[e], hi = Hi[e];
            if (hi > kInfty * kMinScaling) { if (lo < -kInfty * kMinScaling) dy = 0.0; else dy = fmin(dy, 0.0); }
            else if (lo < -kInfty * kMinScaling) dy = fmax(dy, 0.0);
            if (act) DYb[e] = dy;
            ck_max_acc(0, ln, act ? fmax(fabs(Ed[e] * DYd[e]), fabs(Eb[e] * dy)) : 0.0);
            if (cr == 2) {
                const double nd = unid(wave_max(CK[32 * NS + ln]));
                if (ln == 0) RT[59] = nd;
                cr = 0;
                cs = uni(nd > eps_p) ? CK_PI2 : CK_DI0;
            } else ++cr;
        } else if (cs == CK_PI2) {
            const double b = dyn_bound(e), dyd = DYd[e], dyb = DYb[e];
            double lhs = 0.0;
            lhs += b * fmax(dyd, 0.0) + b * fmin(dyd, 0.0);
            lhs += Hi[e] * fmax(dyb, 0.0) + Lo[e] * fmin(dyb, 0.0);
            lhs = wave_sum(act ? lhs : 0.0);
            if (ln == 0) RT[56 + cr] = lhs;
            if (cr == 2) {
                wsync();
                const double tot = unid((RT[56] + RT[57]) + RT[58]), nd = unid(RT[59]);
                cr = 0;
                cs = uni(tot < -eps_p * nd) ? CK_PI3 : CK_DI0;
            } else ++cr;
        } else if (cs == CK_PI3) {
            const LaneC lc = lane_consts();
            const double av = At_elem(e, DYd, DYb, lc);
            ck_max_acc(0, ln, act ? fabs(av / D[e]) : 0.0);
            if (cr == 2) {
                const double na = unid(wave_max(CK[32 * NS + ln])), nd = unid(RT[59]);
                if (uni(na < eps_p * nd) && ln == 0) RT[60] = (double)((int)RT[60] | 4);
                cr = 0;
                cs = CK_DI0;
            } else ++cr;
        } else if (cs == CK_DI0) {
            // check_termination: the dual certificate is evaluated when the dual residual has not converged
            const int fl = __builtin_amdgcn_readfirstlane((int)RT[60]);
            cs = (fl & 2) ? CK_FIN : CK_DI1;
        } else if (cs == CK_DI1) {
            const double dx = DX[e];
            ck_max_acc(0, ln, act ? fabs(D[e] * dx) : 0.0);
            const double qs = wave_sum(act ? Qv[e] * dx : 0.0);
            if (ln == 0) RT[56 + cr] = qs;
            if (cr == 2) {
                wsync();
                const double nd = unid(wave_max(CK[32 * NS + ln])), qdx = unid((RT[56] + RT[57]) + RT[58]);
                if (ln == 0) RT[59] = nd;
                cr = 0;
                cs = uni(nd > eps_d && qdx < -c * eps_d * nd) ? CK_DI2 : CK_FIN;
            } else ++cr;
        } else if (cs == CK_DI2) {
            const double pv = P_row<false>(k, DX);
            ck_max_acc(0, ln, act ? fabs(pv / D[e]) : 0.0);
            if (cr == 2) {
                const double np = unid(wave_max(CK[32 * NS + ln])), nd = unid(RT[59]);
                cr = 0;
                cs = uni(np < c * eps_d * nd) ? CK_DI3 : CK_FIN;
            } else ++cr;
        } else if (cs == CK_DI3) {
            const LaneC lc = lane_consts();
            const double nd = unid(RT[59]);
            const double zd = lc.rmask * (Eid(k, tj) * DX[e] - prev_stage_dot(k, DX)), zb = Sb(k, tj) * DX[k * 8 + lc.bvar];
            double bad = 0.0;
            if (tj < NX) { const double v = zd / Ed[e]; if (v > eps_d * nd || v < -eps_d * nd) bad = 1.0; }
            if (tj < nbox(k)) {
                const double v = zb / Eb[e];
                if ((Hi[e] < kInfty * kMinScaling && v > eps_d * nd) || (Lo[e] > -kInfty * kMinScaling && v < -eps_d * nd)) bad = 1.0;
            }
            ck_max_acc(0, ln, act ? bad : 0.0);
            if (cr == 2) {
                const double worst = unid(wave_max(CK[32 * NS + ln]));
                if (uni(worst == 0.0) && ln == 0) RT[60] = (double)((int)RT[60] | 8);
                cr = 0;
                cs = CK_FIN;
            } else ++cr;
        } else if (cs == CK_FIN) {
            if (!slot_d) return;                 // the verdict is written in a dense-product slot only: the others read it behind that slot's barrier
            const int fl = __builtin_amdgcn_readfirstlane((int)RT[60]);
            Res r;
            r.pri = unid(RT[64]); r.dua = cinv * unid(RT[65]); r.nAxz = unid(RT[66]); r.nPAq = cinv * unid(RT[67]);
            r.s_pri = unid(RT[68]); r.s_dua = unid(RT[69]); r.s_Axz = unid(RT[70]); r.s_PAq = unid(RT[71]);
            int st = LPVMPC_UNSOLVED_;
            if (ckc) {
                if (r.pri > kInfty || r.dua > kInfty) st = LPVMPC_NON_CVX_;
                else if ((fl & 3) == 3) st = LPVMPC_SOLVED_;
                else if (fl & 4) st = LPVMPC_PRIMAL_INFEASIBLE_;
                else if (fl & 8) st = LPVMPC_DUAL_INFEASIBLE_;
            }
            int code = 1;
            double rn = rho;
            if (st != LPVMPC_UNSOLVED_) code = 2;
            else if (cka) {
                rn = rho_estimate(r, rho);
                if (rn > rho * ck_rho_tol || rn < rho / ck_rho_tol) code = 3;
            }
            if (ln == 0) {
                RT[80] = (double)code; RT[81] = (double)st; RT[82] = rn;
                RT[84] = r.pri; RT[85] = r.dua; RT[86] = r.nAxz; RT[87] = r.nPAq; RT[88] = r.s_pri; RT[89] = r.s_dua; RT[90] = r.s_Axz; RT[91] = r.s_PAq;
                RT[83] = (double)ckk;
            }
            cs = CK_IDLE;
        }
    }

    // ---- opt-in warm start (the reference always cold-starts, quirk Q8) ---------------
    // As osqp_warm_start: x <- D^-1 x_prev, y <- c E^-1 y_prev, z <- A x; with shift, stage k takes the
    // previous solution's stage k+1 (the last stage that exists for that row / variable is kept).
    __device__ __forceinline__ void warm_start(const double *st, bool shift) {
        const double *sx = st, *syd = st + NS * 8, *syb = st + 2 * NS * 8;
        for (int e = opaque(tid); e < NS * 8; e += kStride) {
            const int k = e >> 3, r = e & 7;
            const int kx = shift ? (r < NX ? (k < N ? k + 1 : N) : (k < N - 1 ? k + 1 : (k < N ? N - 1 : k))) : k;
            const int kd = shift ? (k < N ? k + 1 : N) : k;
            const bool brow_to_N = !kCtrl && r < NX;                       // planner state-box rows exist up to stage N
            const int kb = shift ? (brow_to_N ? (k < N ? k + 1 : N) : (k < N - 1 ? k + 1 : (k < N ? N - 1 : k))) : k;
            X[e] = (r < nvar(k)) ? sx[kx * 8 + r] / D[e] : 0.0;
            Yd[e] = (r < NX) ? c * syd[kd * 8 + r] / Ed[e] : 0.0;
            Yb[e] = (r < nbox(k)) ? c * syb[kb * 8 + r] / Eb[e] : 0.0;
        }
        sync();
        A_mul(X, Zd, Zb);
        sync();
    }
    __device__ __forceinline__ void save_duals(double *st) const {
        double *syd = st + NS * 8, *syb = st + 2 * NS * 8;
        for (int e = opaque(tid); e < NS * 8; e += kStride) { syd[e] = cinv * Ed[e] * Yd[e]; syb[e] = cinv * Eb[e] * Yb[e]; }
    }

    // ---- the whole solve --------------------------------------------------------------------------
    // ---- straggler deferral: park / restore (SolveArgs::defer_after, ::resume) ------------------------------------------
    // A pool entry = the instance's LDS image followed by kParkScalars doubles {c, cinv, rho, iter, to_chk, to_adp}.  Everything
    // the factorisation reads (scaled tiles, D / E, bounds, weights) is part of the image and does not change between set-up
    // and the end of the ADMM loop, so factor() after a restore reproduces the registers of the uninterrupted run.
    // where an instance's results go: the launch's arguments, or -- for a resumed instance -- the pointers its pool entry carries
    // (a resume pass continues instances of several earlier calls).  Read where needed (uniform: scalar registers), not kept
    // live through the ADMM loop.
    struct Outs { double *xPred, *uPred; int32_t *status, *iters, *polish; double *resid, *state; };
    __device__ __forceinline__ Outs outs_of(const SolveArgs &a, int entry) const {
        if (entry < 0) return Outs{a.xPred, a.uPred, a.status, a.iters, a.polish, a.resid, a.state};
        auto uni = [](unsigned long long v) {        // (uniform: scalar registers)
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            return ((unsigned long long)hi << 32) | lo;
        };
        const unsigned long long *pw = reinterpret_cast<const unsigned long long *>(a.pool_in + (size_t)entry * a.pool_stride + image_doubles(N) + 8);
        return Outs{(double *)uni(pw[0]), (double *)uni(pw[1]), (int32_t *)uni(pw[2]), (int32_t *)uni(pw[3]), (int32_t *)uni(pw[4]), (double *)uni(pw[5]), (double *)uni(pw[6])};
    }
    // (admission of try_park below, one thread: the slot taken, or -1.  pool_count: [0] slots handed out, [1] readers of the resume pass,
    // [2] requests of the young class, [3] requests of the middle class; plain atomic adds, no compare-and-swap loop -- the loop's
    // registers cost the ADMM iteration 30 instructions through the allocator, these cost it 7)
    __device__ __forceinline__ static int park_slot(const SolveArgs &a, int iter) {
        const int K = a.resume ? 0 : a.defer_after, R = a.pool_cap >> 2;
        const int cls = (K > 0 && iter < 2 * K) ? 2 : ((K > 0 && iter < 4 * K) ? 3 : 1);
        const int lim = cls == 2 ? a.pool_cap - R : (cls == 3 ? (R >> 1) : 0x7fffffff);
        const int q = atomicAdd(a.pool_count + cls, cls == 1 ? 0 : 1);
        int s = -1;
        if (q < lim) { s = atomicAdd(a.pool_count, 1); if (s >= a.pool_cap) s = -1; }
        atomicAdd(a.defer_stats + (s >= 0 ? 0 : 1), 1ull);
        return s;
    }
    // AGE-ORDERED ADMISSION (round 6).  A main launch parks into a pool of pool_cap entries that is emptied only by the resume pass
    // behind it.  First come, first served let the many nearly-done instances of a small defer_after = K fill the pool at their first
    // check beyond K, and the one 4000-iteration instance of the batch then stayed resident in the main launch for all of its
    // iterations (defer_after 50: 0.99 M solves/s against 2.0 M at 75 / 100 / 125, profiles/r05_burst_sweep_driver.txt).  Now the
    // instances are admitted by age class: the YOUNG ones (fewer than 2 K iterations) may take three quarters of the pool, the MIDDLE
    // ones (2 K .. 4 K) an eighth of their own, and whoever is beyond 4 K takes any free entry -- at least the last eighth.  An instance
    // that is refused goes on iterating and asks again at its next check, older by then: the long runners always find room, however
    // many young ones wanted in.  Resume passes park into an empty pool of the size of the one they consume: nothing is refused there.
    // lpvmpc_defer_stats reports the instances parked and the requests refused.
    __device__ __forceinline__ bool try_park(const SolveArgs &a, int entry, int inst, int iter, int to_chk, int to_adp) {
        if (tid == 0) RED[79] = (double)park_slot(a, iter);
        sync();
        const int slot = (int)RED[79];
        sync();
        if (slot < 0) return false;                              // no room for this age class: the instance simply goes on here and asks again at its next check
        const int n = (int)image_doubles(N);
        double *dst = a.pool + (size_t)slot * a.pool_stride;
        for (int i = tid; i < n; i += kStride) dst[i] = tA[i];          // tA is the base of the LDS block
        if (tid == 0) {
            double *sc = dst + n;
            sc[0] = c; sc[1] = cinv; sc[2] = rho; sc[3] = (double)iter; sc[4] = (double)to_chk; sc[5] = (double)to_adp; sc[6] = (double)inst;
            const Outs o = outs_of(a, entry);
            unsigned long long *pw = reinterpret_cast<unsigned long long *>(sc + 8);
            pw[0] = (unsigned long long)o.xPred; pw[1] = (unsigned long long)o.uPred; pw[2] = (unsigned long long)o.status;
            pw[3] = (unsigned long long)o.iters; pw[4] = (unsigned long long)o.polish; pw[5] = (unsigned long long)o.resid;
            pw[6] = (unsigned long long)o.state;
            if (o.status) o.status[inst] = LPVMPC_PENDING_;
            if (o.iters) o.iters[inst] = iter;
        }
        return true;
    }
    // restores entry `entry` of a.pool_in; returns the instance index
    __device__ __forceinline__ int restore(const SolveArgs &a, int entry, int &iter, int &to_chk, int &to_adp) {
        const int n = (int)image_doubles(N);
        const double *src = a.pool_in + (size_t)entry * a.pool_stride;
        {   // the image, eight loads of a thread in flight at a time (the rolled loop waited for every one of its 40 global loads in turn)
            int i = tid;
            for (; i + 7 * kStride < n; i += 8 * kStride) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[i + u * kStride];
#pragma unroll
                for (int u = 0; u < 8; ++u) tA[i + u * kStride] = v[u];
            }
            for (; i < n; i += kStride) tA[i] = src[i];
        }
        // the MFMA sweeps read their zeros from SINK: an image parked by a kernel that uses the area otherwise (the DPP two-wavefront
        // factorisation hands a tile through it) must not bring its contents along -- lpvmpc_set_option joins parked work before a
        // kernel_variant change, this makes the kernel itself safe
        if constexpr (kMf) { sync(); for (int i = tid; i < 64 + 8 * NS; i += kStride) SINK[i] = 0.0; }
        const double *sc = src + n;
        if constexpr (kUniScalars) {       // (uniform values: scalar registers -- the tail kernel's vector registers hold K^-1)
            c = unid(sc[0]); cinv = unid(sc[1]);
            set_rho(unid(sc[2]));
        } else {
            c = sc[0]; cinv = sc[1];
            set_rho(sc[2]);
        }
        iter = (int)sc[3]; to_chk = (int)sc[4]; to_adp = (int)sc[5];
        const int inst = __builtin_amdgcn_readfirstlane((int)sc[6]);
        pol = false;
        sync();
        return inst;
    }

    // inst: instance index; entry: pool entry to continue (a.resume) or -1
    __device__ __forceinline__ void run(const SolveArgs &a, int inst, int entry = -1) {
#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS == 5
        tlast = __builtin_amdgcn_s_memtime();
#endif
        bool bad = false;              // this thread saw a non-finite input word
        const bool resuming = TAIL || entry >= 0;     // (the tail kernel only ever resumes)
        if constexpr (GS) { D = a.scal + (size_t)inst * 3 * (NS * 8); Ed = D + NS * 8; Eb = Ed + NS * 8; }
        int iter0 = 1, to_chk0 = 0, to_adp0 = 0;
        if (resuming) {
            inst = restore(a, entry, iter0, to_chk0, to_adp0);
            ++iter0;
        } else if constexpr (!TAIL) {
        // ---------- load + build the unscaled problem ----------
        {   // weights -> LDS (the configuration block itself stays in global memory)
            double v = 0.0;
            if (ti < NX && tj < NX) v = 2.0 * cfg.Q[ti * NX + tj];
            else if (ti >= NX && tj >= NX && ti < NB && tj < NB) {
                v = 2.0 * cfg.R[(ti - NX) * 2 + (tj - NX)];
                if (ti == tj) v += 4.0 * cfg.dR[ti - NX];
            }
            Pm[lane] = v;
            if (tid < 8) dRl[tid] = tid < 2 ? cfg.dR[tid] : 0.0;
        }
        for (int e = tid; e < NS * kTS; e += kStride) { tA[e] = 0.0; }
        for (int e = tid; e < 64 + 8 * NS; e += kStride) SINK[e] = 0.0;        // the MFMA sweeps read their zeros here (nothing writes it in those kernels)
        for (int e = opaque(tid); e < NS * 8; e += kStride) {
            X[e] = 0; D[e] = 1.0; DX[e] = 0; Zd[e] = 0; Yd[e] = 0; Ed[e] = 1.0; DYd[e] = 0; ZTd[e] = 0;
            Zb[e] = 0; Yb[e] = 0; Eb[e] = 1.0; DYb[e] = 0; Lo[e] = 0; Hi[e] = 0; Qv[e] = 0; ZTb[e] = 0;
        }
        sync();
        {
            const double *src = a.AB + (size_t)inst * N * NX * NB;
            if constexpr (kFixN) {
                // every load of the thread in flight at once (round 5: the rolled loop waited for each of its eight global loads in turn --
                // 6 us of a 250 us solve)
                constexpr int kTotal = NT * NX * NB, kTrips = (kTotal + kStride - 1) / kStride;
                double v[kTrips];
#pragma unroll
                for (int i = 0; i < kTrips; ++i) { const int e = tid + i * kStride; v[i] = src[e < kTotal ? e : kTotal - 1]; }
#pragma unroll
                for (int i = 0; i < kTrips; ++i) {
                    const int e = tid + i * kStride;
                    if (e < kTotal) {
                        const int k = e / (NX * NB), rem = e - k * (NX * NB), r = rem / NB, col = rem - r * NB;
                        bad |= !__builtin_isfinite(v[i]);
                        tA[k * kTS + r * 8 + col] = v[i];
                    }
                }
            } else
            for (int e = tid; e < N * NX * NB; e += kStride) {
                const int k = e / (NX * NB), rem = e - k * (NX * NB), r = rem / NB, col = rem - r * NB;
                const double v = src[e];
                bad |= !__builtin_isfinite(v);
                tA[k * kTS + r * 8 + col] = v;
            }
        }
        const size_t uos = 2 + delay;       // u_old row: [OldSteering[0], OldAccelera[0], OldSteering[1 .. delay]]
        const double uo0 = a.u_old ? a.u_old[(size_t)inst * uos + 0] : 0.0, uo1 = a.u_old ? a.u_old[(size_t)inst * uos + 1] : 0.0;
        const double mey = (!kCtrl && a.max_ey) ? a.max_ey[inst] : 0.0;
        bad |= !__builtin_isfinite(uo0) || !__builtin_isfinite(uo1) || !__builtin_isfinite(mey);
        // (the reference velocities of this thread's elements, every round's load in flight before the first use)
        constexpr int kVr = kFixN ? ((NT + 1) * 8 + kStride - 1) / kStride : 1;
        double vref[kVr];
        if constexpr (kCtrl && kFixN) {
#pragma unroll
            for (int i = 0; i < kVr; ++i) { const int k_ = (tid + i * kStride) >> 3; vref[i] = a.vel_ref[(size_t)inst * (N + 1) + (k_ < N ? k_ : N)]; }
        }
        int round_ = 0;
        for (int e = opaque(tid); e < NS * 8; e += kStride, ++round_) {
            const int k = e >> 3, r = e & 7;
            // linear cost: controller q = -2 xtrack' M0 (CTRL:434-447); planner q = L_cf (PLAN:163)
            double q = 0.0;
            if (r < NX) {
                if (kCtrl) {
                    double vr_;
                    if constexpr (kFixN) {
                        vr_ = vref[0];
#pragma unroll
                        for (int i = 1; i < kVr; ++i) if (round_ == i) vr_ = vref[i];
                    } else vr_ = a.vel_ref[(size_t)inst * (N + 1) + k];
                    q = -cfg.Q[0 * NX + r] * 2.0 * vr_;
                }
                else q = cfg.Lcf[r];
            } else if (r < NB && k == 0) q = -2.0 * (r == NX ? uo0 : uo1) * cfg.dR[r - NX];   // CTRL:462 / PLAN:167
            bad |= !__builtin_isfinite(q);
            Qv[e] = q;
            // box rows (CTRL:334-348 / PLAN:173-181); infinities clipped to +-1e30 like OSQP's front end
            if (r < nbox(k)) {
                double lo = cfg.box_lo[r], hi = cfg.box_hi[r];
                if (!kCtrl && r == 3) { lo = -mey; hi = mey; }
                if (kCtrl && r == 6) { lo = hi = a.u_old ? a.u_old[(size_t)inst * (2 + delay) + 2 + k] : 0.0; bad |= !__builtin_isfinite(lo); }   // delta_k = OldSteering[k+1]
                Lo[e] = fmax(lo, -kInfty); Hi[e] = fmin(hi, kInfty);
            }
        }
        if (tid < 16) { const double v = (tid < NX) ? a.x0[(size_t)inst * a.x0_stride + tid] : 0.0; bad |= !__builtin_isfinite(v); beq[tid] = v; }
        sync();
        // Non-finite input data (a NaN measurement, or the NaN "solution" of an infeasible previous tick fed back by the
        // caller): no iteration is run, the outputs are NaN and the status is UNSOLVED.  (OSQP's NaN-blind max-norms
        // would declare such a problem solved at its first check and return NaN as well.)
        if (bad) beq[15] = 1.0;
        sync();
        if (beq[15] != 0.0) {
            const double nanv = __builtin_nan("");
            for (int e = opaque(tid); e < NS * 8; e += kStride) {
                const int k = e >> 3, r = e & 7;
                if (r < NX) a.xPred[((size_t)inst * NS + k) * NX + r] = nanv;
                else if (r < NB && k < N) a.uPred[((size_t)inst * N + k) * 2 + (r - NX)] = nanv;
            }
            if (a.state) { double *so = a.state + (size_t)inst * 3 * NS * 8; for (int e = tid; e < 3 * NS * 8; e += kStride) so[e] = 0.0; }
            if (tid == 0) {
                if (a.status) a.status[inst] = LPVMPC_UNSOLVED_;
                if (a.iters) a.iters[inst] = 0;
                if (a.polish) a.polish[inst] = 0;
                if (a.resid) { double *o = a.resid + (size_t)inst * 4; o[0] = o[1] = o[2] = o[3] = nanv; }
            }
            return;
        }

        // ---------- setup: scaling, rho, factorisation ----------
        STAMP5(0);
        if (cfg.scaling > 0) scale_data();
        for (int e = opaque(tid); e < NS * 8; e += kStride) { Qv[e] *= c * D[e]; Lo[e] *= Eb[e]; Hi[e] *= Eb[e]; }
        if (tid < 8) beq[tid] *= Ed[tid];
        sync();
        cache_row_coefficients();
        set_rho(fmin(fmax(cfg.rho, kRhoMin), kRhoMax));
        }
        cache_box_weights();
        STAMP5(1);
        factor(cfg.sigma);                      // first factorisation, or the one that puts a restored instance back into its registers
        STAMP5(2);
        if (!resuming) {
            if (a.warm && a.state) warm_start(a.state + (size_t)inst * 3 * NS * 8, a.warm == 2);
            recompute_w();                      // cold start: x = z = y = 0
        }

        // ---------- ADMM ----------
        double alpha = cfg.alpha, sigma = cfg.sigma, oma_u = 1.0 - cfg.alpha;
        if constexpr (kRegState) { alpha = unid(alpha); sigma = unid(sigma); oma_u = unid(oma_u); }      // (uniform: scalar registers -- the element state takes the vector ones)
        int status = LPVMPC_UNSOLVED_, iter = 0, status_polish = 0;
        double pri_res = 0, dua_res = 0, obj = __builtin_nan("");
        Res R = {0, 0, 0, 0, 0, 0, 0, 0};
        bool checked = false;
        // loop control on local copies and down-counters: the configuration block lives in global memory and stores to the
        // outputs may alias it as far as the compiler knows, so every use inside the loop is a scalar load plus a wait, and
        // iter % interval is a dozen scalar instructions per iteration
        const int max_iter = cfg.max_iter, chk_every = cfg.check_termination > 0 ? cfg.check_termination : 0;
        const int adp_every = (cfg.adaptive_rho && cfg.adaptive_rho_interval > 0) ? cfg.adaptive_rho_interval : 0;
        const double rho_tol = cfg.rho_tol;
        int to_chk = resuming ? to_chk0 : chk_every, to_adp = resuming ? to_adp0 : adp_every;
        const int defer_after = a.defer_after > 0 ? (resuming ? iter0 - 1 + a.defer_after : a.defer_after) : 0;     // park at the first check at or beyond this iteration
        constexpr bool reg_state = kRegState;
        bool filed = true, ran = false;                           // (kRegState) the last iteration filed the state in the arrays; an iteration has run
        if constexpr (kRegState && !TAIL) { if (reg_state) state_load(); }
        if constexpr (TAIL) {
            // ---- the tail kernel's loop: dense product, fused element phase; checks evaluated beside the iterations (see tail_fused).
            // The last wavefront runs the checker's side of the same loop (tail_loop<true>): the same scalar control flow on the same
            // values, hence the same barriers, but none of the other wavefronts' registers or code.  A new rho sends everybody back to
            // the factorisation (which exists once in the kernel, above this loop's entry).
            TailCtl t;
            t.iter = iter0 - 1; t.to_chk = to_chk; t.to_adp = to_adp; t.status = LPVMPC_UNSOLVED_; t.rn = rho;
            ck_alpha = alpha; ck_eps_abs = unid(cfg.eps_abs); ck_eps_rel = unid(cfg.eps_rel); ck_eps_p = unid(cfg.eps_prim_inf); ck_eps_d = unid(cfg.eps_dual_inf); ck_rho_tol = unid(cfg.rho_tol);
            const int role = __builtin_amdgcn_readfirstlane(wv);
            for (;;) {
                if (tid == 0) RT[83] = -1.0;
                tail_begin(sigma);
                const int v = role == NW - 1 ? tail_loop<true>(t, sigma, alpha, max_iter, chk_every, adp_every)
                                             : tail_loop<false>(t, sigma, alpha, max_iter, chk_every, adp_every);
                if (v >= 2) set_rho(t.rn);
                if (v != 2) break;
                factor(sigma);
            }
            launder_ids();
            // the residual norms of the verdict that ended the loop (the last one written)
            R.pri = RT[84]; R.dua = RT[85]; R.nAxz = RT[86]; R.nPAq = RT[87]; R.s_pri = RT[88]; R.s_dua = RT[89]; R.s_Axz = RT[90]; R.s_PAq = RT[91];
            iter = t.iter; status = t.status; pri_res = R.pri; dua_res = R.dua;
            checked = true;
        } else
        for (iter = iter0; iter <= max_iter; ++iter) {
            checked = chk_every > 0 && --to_chk == 0;
            const bool adapt = adp_every > 0 && --to_adp == 0;
            if (checked) to_chk = chk_every;
            if (adapt) to_adp = adp_every;
#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS != 5
            tlast = __builtin_amdgcn_s_memtime();
#endif
            // (the two markers bracket the per-iteration code in the kept assembly: tools/check_kernel_resources.py counts the
            // scratch accesses and SGPR-spill lane moves between them)
            asm volatile("; LPVMPC_HOT_BEGIN");
#ifdef LPVMPC_PHASE_ONLY
            // diagnostic builds (tools/phase_pmc.sh): the loop runs ONE phase of the iteration (1 right-hand side, 2 KKT solve, 3 update;
            // 0 none), so that the hardware counters of a launch can be attributed to it.  The iterates are meaningless.
            if (LPVMPC_PHASE_ONLY == 1) build_rhs(sigma);
            if (LPVMPC_PHASE_ONLY == 2) kkt_solve();
            if (LPVMPC_PHASE_ONLY == 3) update(alpha, checked);
#else
            if constexpr (kFour) iterate4(sigma, alpha, checked);
            else if constexpr (kMf && kRegState) {
                // the element state lives in registers (see kRegState); handles with steeringDelay > 0 run the LDS form
                // (launch_solve sends controller handles with steeringDelay > 0 -- a third box row in the right-hand side, which the fused
                // blocks do not carry -- to the DPP kernel: these kernels have no LDS form of the element phases)
                rhs_s(sigma);
                const MfLane m = mf_lane();     // (in front of the barrier: the sweeps' lane constants form while the stores drain)
                sync();
                STAMP(0);
                kkt_solve_mf(m);
                if (iter == iter0) update_s<true>(alpha, oma_u, checked || adapt); else update_s<false>(alpha, oma_u, checked || adapt);
                filed = checked || adapt; ran = true;
            } else if constexpr (kMf) {
                build_rhs<false>(sigma);
                const MfLane m = mf_lane();     // (in front of the barrier: the sweeps' lane constants form while the stores drain)
                sync();
                STAMP(0);
                kkt_solve_mf(m);
                update(alpha, checked);
            } else {
            build_rhs(sigma);
            STAMP(0);
            kkt_solve();
            update(alpha, checked);         // delta_x / delta_y are only read by the infeasibility tests
            }
#endif
            asm volatile("; LPVMPC_HOT_END");
            STAMP(3);
            STAMP5(3);
            if (checked || adapt) {
                const Ids loop_ids = ids_save();
                if constexpr (kUniScalars && LPVMPC_BLOCK_LAUNDER == 1) launder_ids();       // (see ids_save)
                if constexpr (kUniScalars && LPVMPC_BLOCK_LAUNDER == 2) set_fac_ids(opaque(lane));
                R = residuals(X, Zd, Zb, Yd, Yb);
                pri_res = R.pri; dua_res = R.dua;
                if (checked) { status = check_termination(R, false); if (status != LPVMPC_UNSOLVED_) break; }
                if (adapt) {
                    const double rn = rho_estimate(R, rho);
                    if (rn > rho * rho_tol || rn < rho / rho_tol) { set_rho(rn); cache_box_weights(); factor(sigma); }
                }
                recompute_w();              // the residual evaluation used ZT* as scratch (and rho may have changed)
                if constexpr (kRegState) { if (reg_state) state_reload_w(); }
                // straggler deferral: unsolved at this check and past the budget -> park and end the workgroup (block-uniform)
                if (checked && defer_after > 0 && iter >= defer_after && iter < max_iter && try_park(a, entry, inst, iter, to_chk, to_adp)) return;
                if constexpr (kUniScalars) ids_restore(loop_ids);
                STAMP5(4);
            }
        }
        STAMP5(4);
        if constexpr (kRegState) { if (reg_state && ran && !filed) { state_flush(); sync(); } }      // (a loop that ends on a plain iteration: max_iter off the check grid)
        if constexpr (kUniScalars && !TAIL) launder_ids();      // (the post-loop code's addresses are formed behind the loop: see launder_ids)
        if (iter > max_iter) iter = max_iter;
        if (!checked) {
            R = residuals(X, Zd, Zb, Yd, Yb); pri_res = R.pri; dua_res = R.dua;
            status = check_termination(R, false);
        }
        const bool has_sol = !(status == LPVMPC_PRIMAL_INFEASIBLE_ || status == LPVMPC_PRIMAL_INFEASIBLE_INACC_ ||
                               status == LPVMPC_DUAL_INFEASIBLE_ || status == LPVMPC_DUAL_INFEASIBLE_INACC_ ||
                               status == LPVMPC_NON_CVX_);
        if (has_sol) obj = objective(X);
        if (status == LPVMPC_UNSOLVED_) { status = check_termination(R, true); if (status == LPVMPC_UNSOLVED_) status = LPVMPC_MAX_ITER_; }
        if (status == LPVMPC_PRIMAL_INFEASIBLE_ || status == LPVMPC_PRIMAL_INFEASIBLE_INACC_) obj = kInfty;
        if (status == LPVMPC_DUAL_INFEASIBLE_ || status == LPVMPC_DUAL_INFEASIBLE_INACC_) obj = -kInfty;

        const double rho_admm = rho;
        const Outs o_ = outs_of(a, entry);
        double *const st_out = o_.state ? o_.state + (size_t)inst * 3 * NS * 8 : nullptr;
        if (st_out) save_duals(st_out);        // ADMM duals (the polish below reuses the y arrays)
        // ---------- polish ----------
        if (cfg.polish && status == LPVMPC_SOLVED_) {
            STAMP5(4);
            const PolishOut po = polish(pri_res, dua_res, obj);
            STAMP5(6);
            status_polish = po.flag; pri_res = po.pri; dua_res = po.dua; obj = po.obj;
            if (st_out && status_polish == 1) save_duals(st_out);
        }

        // ---------- write back (store_solution) ----------
        // store_solution: decided on the FINAL status (the approximate check above may have turned an
        // unsolved run into "primal/dual infeasible inaccurate", which carries no solution)
        const bool sol = !(status == LPVMPC_PRIMAL_INFEASIBLE_ || status == LPVMPC_PRIMAL_INFEASIBLE_INACC_ ||
                           status == LPVMPC_DUAL_INFEASIBLE_ || status == LPVMPC_DUAL_INFEASIBLE_INACC_ ||
                           status == LPVMPC_NON_CVX_);
        const double nan = __builtin_nan("");
        for (int e = opaque(tid); e < NS * 8; e += kStride) {
            const int k = e >> 3, r = e & 7;
            const double v = sol ? D[e] * X[e] : nan;
            if (r < NX) o_.xPred[((size_t)inst * NS + k) * NX + r] = v;
            else if (r < NB && k < N) o_.uPred[((size_t)inst * N + k) * 2 + (r - NX)] = v;
            if (st_out) st_out[e] = sol ? v : 0.0;
        }
        if (st_out && !sol) {      // no solution: do not seed the next solve with garbage
            for (int e = opaque(tid); e < NS * 8; e += kStride) { st_out[NS * 8 + e] = 0.0; st_out[2 * NS * 8 + e] = 0.0; }
        }
#if defined(LPVMPC_STAMPS) && LPVMPC_STAMPS == 5
        STAMP5(7);
        if (tid == 0) {
            for (int i = 0; i < 8; ++i) o_.xPred[(size_t)inst * NS * NX + i] = (double)stamp[i];
            if (o_.status) o_.status[inst] = status;
            if (o_.iters) o_.iters[inst] = iter;
            if (o_.polish) o_.polish[inst] = status_polish;
        }
        return;
#endif
#ifdef LPVMPC_STAMPS
        if constexpr (kFour) {          // the inner top wavefront's own eight segments (inner4) behind wavefront 0's six
            if (tid == 128) for (int i = 0; i < 8; ++i) o_.xPred[(size_t)inst * NS * NX + 8 + i] = (double)stamp[i] / iter;
        }
#if LPVMPC_STAMPS == 4
        if constexpr (TAIL) {
            if (tid == 0) for (int i = 0; i < 8; ++i) o_.xPred[(size_t)inst * NS * NX + i] = (double)stamp[i] / iter;
            if (tid == 0) for (int i = 0; i < 40; ++i) o_.xPred[(size_t)inst * NS * NX + 8 + i] = CK[64 * NS + 512 + i];      // checker: cycles and steps per state
        }
#endif
        if (tid == 0 && o_.resid) {     // diagnostic build: resid carries per-iteration cycle counts instead
            double *o = o_.resid + (size_t)inst * 4;
#if LPVMPC_STAMPS == 2
            o[0] = (double)stamp[4]; o[1] = (double)stamp[5]; o[2] = (double)stamp[6]; o[3] = (double)stamp[7];      // factor phases, summed over the solve
#else
            o[0] = (double)stamp[0] / iter; o[1] = (double)stamp[1] / iter; o[2] = (double)stamp[2] / iter; o[3] = (double)stamp[3] / iter;
#endif
            if constexpr (kFour) {      // six segments of iterate4: the first words of the instance's xPred carry them
                for (int i = 0; i < 6; ++i) o_.xPred[(size_t)inst * NS * NX + i] = (double)stamp[i] / iter;
                o_.xPred[(size_t)inst * NS * NX + 6] = (double)stamp[6] / iter; o_.xPred[(size_t)inst * NS * NX + 7] = (double)stamp[7] / iter;      // (LPVMPC_STAMPS == 3: outer forward, B0 .. loop / the loop)
            }
            if (o_.status) o_.status[inst] = status;
            if (o_.iters) o_.iters[inst] = iter;
            return;
        }
#endif
        if (tid == 0) {
            if (o_.status) o_.status[inst] = status;
            if (o_.iters) o_.iters[inst] = iter;
            if (o_.polish) o_.polish[inst] = status_polish;
            if (o_.resid) { double *o = o_.resid + (size_t)inst * 4; o[0] = pri_res; o[1] = dua_res; o[2] = obj; o[3] = rho_admm; }
        }
    }

    // status codes (values of lpvmpc.h / OSQP)
    static constexpr int LPVMPC_SOLVED_ = 1, LPVMPC_SOLVED_INACC_ = 2, LPVMPC_PRIMAL_INFEASIBLE_INACC_ = 3,
                         LPVMPC_DUAL_INFEASIBLE_INACC_ = 4, LPVMPC_MAX_ITER_ = -2, LPVMPC_PRIMAL_INFEASIBLE_ = -3,
                         LPVMPC_DUAL_INFEASIBLE_ = -4, LPVMPC_NON_CVX_ = -7, LPVMPC_UNSOLVED_ = -10;

    // OSQP check_termination: returns the status reached, or LPVMPC_UNSOLVED_ when the iteration must go on
    __device__ __forceinline__ int check_termination(const Res R, bool approx) {
        double ea = cfg.eps_abs, er = cfg.eps_rel, epi = cfg.eps_prim_inf, edi = cfg.eps_dual_inf;
        if (R.pri > kInfty || R.dua > kInfty) return LPVMPC_NON_CVX_;
        if (approx) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
        bool prc = false, drc = false, pic = false, dic = false;
        const double ep = ea + er * R.nAxz;
        if (R.pri < ep) prc = true; else pic = primal_infeasible(epi);
        const double ed = ea + er * R.nPAq;
        if (R.dua < ed) drc = true; else dic = dual_infeasible(edi);
        if (prc && drc) return approx ? LPVMPC_SOLVED_INACC_ : LPVMPC_SOLVED_;
        if (pic) return approx ? LPVMPC_PRIMAL_INFEASIBLE_INACC_ : LPVMPC_PRIMAL_INFEASIBLE_;
        if (dic) return approx ? LPVMPC_DUAL_INFEASIBLE_INACC_ : LPVMPC_DUAL_INFEASIBLE_;
        return LPVMPC_UNSOLVED_;
    }
    // OSQP compute_rho_estimate (scaled-space norms)
    __device__ __forceinline__ double rho_estimate(const Res &R, double rho) const {
        const double pr = R.s_pri / (R.s_Axz + 1e-10);
        const double dr = R.s_dua / (R.s_PAq + 1e-10);
        const double rn = rho * sqrt(pr / (dr + 1e-10));
        return fmin(fmax(rn, kRhoMin), kRhoMax);
    }

    // ---- polish (OSQP polish.c) on the reduced form ----------------------------------------------
    // active rows carry weight 1/delta in K_pol = P + delta I + A_act' A_act / delta; W > 0 marks
    // upper-active, W < 0 lower-active rows (|W| = 1/delta), W = 0 inactive.
    // Results come back by value.  (An earlier version used reference parameters and returned wrong values in the N = 40
    // instantiation; the cause turned out to be the machine-sink placement described in the Makefile, not the references.)
    struct PolishOut { int flag; double pri, dua, obj; };
    __device__ __forceinline__ PolishOut polish(double pri_res, double dua_res, double obj) {
        const double delta = cfg.delta, dinv = 1.0 / cfg.delta;
        for (int e = opaque(tid); e < NS * 8; e += kStride) {
            const int k = e >> 3, r = e & 7;
            double wd = 0.0, wb = 0.0;
            if (r < NX) {
                const double b = dyn_bound(e);
                if (Zd[e] - b < -Yd[e]) wd = -dinv; else if (b - Zd[e] < Yd[e]) wd = dinv;
            }
            if (r < nbox(k)) {
                if (Zb[e] - Lo[e] < -Yb[e]) wb = -dinv; else if (Hi[e] - Zb[e] < Yb[e]) wb = dinv;
            }
            DYd[e] = wd; DYb[e] = wb;          // signed flags: w_dyn / w_box read |flag| while pol is set
        }
        sync();
        pol = true;
        STAMP5(6);
        factor(delta);
        STAMP5(5);
        // px (DX), py (Yd/Yb are overwritten: ADMM duals are no longer needed)
        // Element loops that only hand a value from one loop to the next THROUGH THE ELEMENT'S OWN SLOT are one loop here (round 5:
        // 13 of the polish's ~45 barriers and as many exposed LDS round trips less -- the phases are latency bound).  The operations and
        // their order are those of the separate loops; a value that used to pass through LDS between two of them is fenced, so that the
        // compiler cannot contract its last multiplication with the next loop's first addition.
        const LaneC lc = lane_consts();
        auto fence = [](double v) { asm volatile("" : "+v"(v)); return v; };
        auto A_elem = [&](int e, const double *src, double &vd, double &vb) {       // row e of A * src (see A_mul)
            const int k = e >> 3;
            vd = fence(lc.rmask * (Eid(k, tj) * src[e] - prev_stage_dot(k, src)));
            vb = fence(Sb(k, tj) * src[k * 8 + lc.bvar]);
        };
        auto bound_of = [&](int e) { const double f = DYb[e]; return f > 0 ? Hi[e] : (f < 0 ? Lo[e] : 0.0); };
        // initial solve: rhs = -q + A'(W b)
        for (int e = opaque(tid); e < NS * 8; e += kStride) {
            ZTd[e] = fabs(DYd[e]) * dyn_bound(e);
            ZTb[e] = fabs(DYb[e]) * bound_of(e);
        }
        sync();
        for (int e = opaque(tid); e < NS * 8; e += kStride) XT[e] = At_elem(e, ZTd, ZTb, lc) - Qv[e];
        sync();
        kkt_solve();
        for (int e = opaque(tid); e < NS * 8; e += kStride) {
            double vd, vb;
            A_elem(e, XT, vd, vb);
            DX[e] = XT[e];
            Yd[e] = fabs(DYd[e]) * (vd - dyn_bound(e));
            Yb[e] = fabs(DYb[e]) * (vb - bound_of(e));
        }
        sync();
        for (int it = 0; it < cfg.polish_refine_iter; ++it) {
            // r1 = -q - P px - A' py ; r2 = b - A px (active rows)
            for (int e = opaque(tid); e < NS * 8; e += kStride) {
                VT[e] = P_row<false>(e >> 3, DX);
                AT[e] = At_elem(e, Yd, Yb, lc);
                double vd, vb;
                A_elem(e, DX, vd, vb);
                const double fd = DYd[e], fb = DYb[e];
                const double r2d = (fd != 0.0) ? dyn_bound(e) - vd : 0.0;
                const double r2b = (fb != 0.0) ? (fb > 0 ? Hi[e] : Lo[e]) - vb : 0.0;
                ZTd[e] = r2d; ZTb[e] = r2b;
                Zd[e] = fabs(fd) * r2d; Zb[e] = fabs(fb) * r2b;
            }
            sync();
            for (int e = opaque(tid); e < NS * 8; e += kStride) XT[e] = At_elem(e, Zd, Zb, lc) + (-Qv[e] - VT[e] - AT[e]);
            sync();
            kkt_solve();
            for (int e = opaque(tid); e < NS * 8; e += kStride) {
                double vd, vb;
                A_elem(e, XT, vd, vb);
                DX[e] += XT[e];
                Yd[e] += fabs(DYd[e]) * (vd - ZTd[e]);
                Yb[e] += fabs(DYb[e]) * (vb - ZTb[e]);
            }
            sync();
        }
        // pol_z = A px, project (z, y) on the normal cone
        for (int e = opaque(tid); e < NS * 8; e += kStride) {
            double vd, vb;
            A_elem(e, DX, vd, vb);
            { const double b = dyn_bound(e), t = vd + Yd[e]; Zd[e] = b; Yd[e] = t - b; }         // projection on [b, b]
            { const double t = vb + Yb[e], z = clipd(t, Lo[e], Hi[e]); Zb[e] = z; Yb[e] = t - z; }
        }
        sync();
        const double pobj = objective(DX);
        const Res R = residuals(DX, Zd, Zb, Yd, Yb);
        const bool good = (R.pri < pri_res && R.dua < dua_res) || (R.pri < pri_res && dua_res < 1e-10) ||
                          (R.dua < dua_res && pri_res < 1e-10);
        if (good) {
            for (int e = opaque(tid); e < NS * 8; e += kStride) X[e] = DX[e];
            sync();
            return PolishOut{1, R.pri, R.dua, pobj};
        }
        return PolishOut{-1, pri_res, dua_res, obj};
    }
};

// waves per SIMD the kernel must allow: workgroups per CU are LDS-limited (4 at N = 20, 2 at N = 30 / 40), so only
// the N <= 20 two-wavefront kernels need two waves per SIMD (<= 256 registers); the others may use up to 512
// -- and the planner N = 30 kernel with its equilibration vectors in global memory (three instances per CU: six waves on four SIMDs)
template <int NT, int NW, bool GS>
#ifdef LPVMPC_FORCE_TWO_WAVES_PER_SIMD
constexpr int min_waves_per_simd() { return NW >= 8 ? NW / 4 : NW; }      // diagnostic: provoke register spilling in the big-N kernels
#else
constexpr int min_waves_per_simd() { return ((NW == 2 && NT <= 20) || (NW == 1 && NT > 0 && NT <= 10) || GS || NW == 8 || NW == 4) ? 2 : 1; }     // N = 8 / 10 (one wavefront): 256 registers -> eight / seven instances per CU (N = 10 sat at 256 without the bound in round 5 and at 257 -- four per CU -- in round 6: the bound holds it); NW = 8: the tail kernel, one workgroup per CU
#endif

template <int NX, int NT, int NW, bool MF = false, bool GS = false, bool TAIL = false>
__global__ void __launch_bounds__(64 * NW, (min_waves_per_simd<NT, NW, GS>())) admm_solve_kernel(const DevCfg *__restrict__ cfgp, SolveArgs a) {
    extern __shared__ __align__(16) double smem[];
    int inst = blockIdx.x, entry = -1;
    if (a.resume) {
        // every workgroup of a resume pass reads the number of parked entries; the last one to have read it clears the counters
        // (this launch consumes the pool; launches behind it in the stream park into it again)
        int &n_parked = *reinterpret_cast<int *>(smem);          // (the instance's LDS image is restored over it afterwards)
        if (threadIdx.x == 0) {
            n_parked = atomicAdd(a.pool_in_count, 0);
            __threadfence();
            if (atomicAdd(a.pool_in_count + 1, 1) == (int)gridDim.x - 1) { a.pool_in_count[0] = 0; a.pool_in_count[1] = 0; a.pool_in_count[2] = 0; a.pool_in_count[3] = 0; }
        }
        __syncthreads();
        const int n = n_parked;
        entry = blockIdx.x;
        if (entry >= (n < a.pool_cap ? n : a.pool_cap)) return;
        __syncthreads();          // every wavefront has read the count before restore() stores the image over it
    } else if (TAIL || inst >= a.B) return;
    Solver<NX, NT, NW, MF, GS, TAIL> s(*cfgp, smem);
    s.run(a, inst, entry);
}

template <int NX, int NT, int NW, bool MF = false, bool GS = false, bool TAIL = false>
static hipError_t launch_one(const DevCfg &cfg, const DevCfg *dcfg, const SolveArgs &a, hipStream_t stream) {
    const size_t lds = Solver<NX, NT, NW, MF, GS, TAIL>::lds_doubles(cfg.N) * sizeof(double);
    // The LDS opt-in is a per-function AND per-device attribute: remember it per (instantiation, device ordinal).  Handles on
    // different devices may launch from different threads (lpvmpc.h: thread-safe across handles), hence the atomic mask.
    static std::atomic<uint64_t> attr_mask[4];           // 256 device ordinals
    int dev = 0;
    { hipError_t err = hipGetDevice(&dev); if (err != hipSuccess) return err; }
    const uint64_t bit = 1ull << (dev & 63);
    std::atomic<uint64_t> &word = attr_mask[(dev >> 6) & 3];
    if (!(word.load(std::memory_order_acquire) & bit)) {
        hipError_t err = hipFuncSetAttribute((const void *)admm_solve_kernel<NX, NT, NW, MF, GS, TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (err != hipSuccess) return err;
        word.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((admm_solve_kernel<NX, NT, NW, MF, GS, TAIL>), dim3(a.resume ? a.pool_cap : a.B), dim3(64 * NW), lds, stream, dcfg, a);
    return hipGetLastError();
}

// 1 if (kind, N) has a register-tile instantiation
int solve_has_fast_path(int kind, int N) {
    return (kind == 0 && (N == 20 || N == 10 || N == 8)) || (kind == 1 && (N == 30 || N == 40 || N == 20));
}

size_t solve_lds_bytes(int kind, int N) {
    const bool fast = solve_has_fast_path(kind, N) != 0;
    return ((size_t)(N + 1) * ((fast ? 1 : 3) * kTS + 19 * 8 + 8) + 16 + 64 + 8 + 80 + 64) * sizeof(double);
}

// kernel_variant: 0 = best available, 1 = run-time-horizon kernel (factor tiles in LDS), 2 = compile-time horizon
// with one wavefront per instance (where instantiated)
hipError_t launch_solve(const DevCfg &cfg, const DevCfg *dcfg, const SolveArgs &a, hipStream_t stream, int kernel_variant) {
    const bool generic = kernel_variant == 1, one_wave = kernel_variant == 2, dpp = kernel_variant == 3;
    // a resume pass that runs its entries to completion may use the whole-CU tail kernel (SolveArgs::tail); the entries must
    // have been parked by a kernel with the compile-time-horizon LDS image
    // (controller and planner at N = 20: 168 unknowns either way, K^-1 fits the register file of a CU; 248 / 328 at N = 30 / 40 do not.
    // steeringDelay > 0: the tail kernel's element threads keep two box rows per variable in registers and delta would have three --
    // the reference runs delay 0 (CMAIN:49); such handles finish their parked instances with the two-wavefront kernel)
    if (a.resume && a.tail && a.defer_after == 0 && !generic && cfg.N == 20 && cfg.steering_delay == 0)
        return cfg.kind == 0 ? launch_one<6, 20, 8, false, false, true>(cfg, dcfg, a, stream) : launch_one<5, 20, 8, false, false, true>(cfg, dcfg, a, stream);
#if defined(LPVMPC_DEV_TAIL_ONLY)
    return hipErrorInvalidValue;       // development builds (seconds instead of minutes): the tail kernels only, for looking at their assembly
#elif defined(LPVMPC_DEV_MAIN_ONLY)
    return launch_one<6, 20, 2, true>(cfg, dcfg, a, stream);      // ... or the headline kernel only (no steeringDelay > 0 in this build)
#elif defined(LPVMPC_DEV_C10)
    return launch_one<6, 10, 1>(cfg, dcfg, a, stream);      // ... or the one-wavefront controller kernel at N = 10
#elif defined(LPVMPC_DEV_C20_4W)
    return launch_one<6, 20, 4, true>(cfg, dcfg, a, stream);      // ... or a four-wavefront controller kernel (experiment)
#elif defined(LPVMPC_DEV_P30_2W)
    return launch_one<5, 30, 2, true>(cfg, dcfg, a, stream);      // ... or the two-wavefront MFMA planner kernel at N = 30
#elif defined(LPVMPC_DEV_P30_ONLY)
    return cfg.N == 30 ? launch_one<5, 30, 4, true>(cfg, dcfg, a, stream) : launch_one<5, 40, 4, true>(cfg, dcfg, a, stream);      // ... or the four-wavefront planner kernels
#else
    if (cfg.kind == 0) {
        // (steeringDelay > 0, CTRL:518-527: a third box row on delta in the first stages.  The MFMA kernel's element phases keep their state
        // in registers and carry two box rows per variable: such handles take the DPP kernel -- the reference runs delay 0, CMAIN:49)
        // kernel_variant 9: the LATENCY form of the headline kernel -- its two chains relayed over four wavefronts (Solver::kFour, the planner
        // kernels' relay at kMid = 10; the factorisation's hand-over area is its own 448 words, kFourSmall).  The arithmetic of the
        // two-wavefront kernel step for step; an instance alone on its CU finishes 6 % sooner (B <= 256), a full chip gains nothing
        // (two instances per CU instead of four).  Opt-in, not chosen by launch shape as the planner's forms are: the block sums of this
        // kernel family at N = 20 (the cost normalisation's mean, the certificates' sums, the objective) associate by the number of
        // wavefronts, so where such a sum decides a word may differ in its last bits between the forms.  Launches of a handle with
        // straggler deferral keep the two-wavefront form (the parked image and the tail kernel are that form's).
        if (!generic && cfg.N == 20 && kernel_variant == 9 && cfg.steering_delay == 0 && a.defer_after == 0 && !a.resume)
            return launch_one<6, 20, 4, true>(cfg, dcfg, a, stream);
        if (!generic && cfg.N == 20) return one_wave ? launch_one<6, 20, 1>(cfg, dcfg, a, stream)
                                          : ((dpp || cfg.steering_delay > 0) ? launch_one<6, 20, 2>(cfg, dcfg, a, stream) : launch_one<6, 20, 2, true>(cfg, dcfg, a, stream));
        if (!generic && cfg.N == 10) return launch_one<6, 10, 1>(cfg, dcfg, a, stream);
        if (!generic && cfg.N == 8) return launch_one<6, 8, 1>(cfg, dcfg, a, stream);      // the launch file's controller horizon (MAIN_LAUNCH.launch:117)
        return launch_one<6, 0, 1>(cfg, dcfg, a, stream);
    }
    const bool mf = kernel_variant == 4;      // the planner N = 30 kernel with two wavefronts and MFMA sweeps (two instances per CU)
    // N = 30 (and N = 40 through kernel_variant 6): two forms of ONE arithmetic (every output word equal: test_four_wavefront_planner_kernels_are_bit_identical...),
    // chosen by what the launch is short of.  FOUR wavefronts per instance (round 4: the two chains relayed, two wavefronts per SIMD,
    // Solver::kFour) iterate a quarter faster when an instance has its CU to itself: the form for batches that do not fill the chip, and
    // for everything a handle with straggler deferral launches (the parked image is the four-wavefront one).  TWO wavefronts per
    // instance (kernel_variant 4 / 6) issue fewer instructions per iteration and, since round 5, run both element rounds of a
    // wavefront as one block: on a chip that is full either way -- a plain launch of 512 instances or more, two per CU -- they finish
    // a batch sooner: N = 30, 2.11 against 2.52 us per iteration at B = 512, configs[2] 0.197 -> 0.215 M solves/s.  kernel_variant 8 = four
    // wavefronts whatever the batch.
    // N = 30 kernel_variant 7 = round 3's default: the DPP two-wavefront kernel with its equilibration vectors in global memory (three
    // instances per CU) whenever the caller provides the room, nothing is parked or resumed and the batch exceeds 512 instances, the
    // LDS form (kernel_variant 5) otherwise; N = 40 kernel_variant 3 = DPP sweeps.
    const bool full_chip = kernel_variant != 8 && a.defer_after == 0 && !a.resume && a.B >= 512;
    const bool gs = kernel_variant == 7 && a.scal != nullptr && a.defer_after == 0 && !a.resume && a.B > 512;
    if (!generic && cfg.N == 30) {
        if (one_wave) return launch_one<5, 30, 1>(cfg, dcfg, a, stream);
        if (mf) return launch_one<5, 30, 2, true>(cfg, dcfg, a, stream);
        if (gs) return launch_one<5, 30, 2, false, true>(cfg, dcfg, a, stream);
        if (kernel_variant == 5 || kernel_variant == 7 || dpp) return launch_one<5, 30, 2>(cfg, dcfg, a, stream);
        return full_chip ? launch_one<5, 30, 2, true>(cfg, dcfg, a, stream) : launch_one<5, 30, 4, true>(cfg, dcfg, a, stream);
    }
    if (!generic && cfg.N == 40) {
        if (dpp) return launch_one<5, 40, 2>(cfg, dcfg, a, stream);
        // (N = 40 stays on four wavefronts at every size: its two-wavefront form has three element rounds, not two, and although a 512-batch
        // of it iterates 5 % faster the cascade of configs[4] -- a thinning fleet, controller launches in between -- loses 7 % with it)
        return kernel_variant == 6 ? launch_one<5, 40, 2, true>(cfg, dcfg, a, stream) : launch_one<5, 40, 4, true>(cfg, dcfg, a, stream);
    }
    // (kernel_variant 9 at N = 20: the latency form, as the controller's -- see there)
    if (!generic && cfg.N == 20 && kernel_variant == 9 && a.defer_after == 0 && !a.resume) return launch_one<5, 20, 4, true>(cfg, dcfg, a, stream);
    if (!generic && cfg.N == 20) return dpp ? launch_one<5, 20, 2>(cfg, dcfg, a, stream) : launch_one<5, 20, 2, true>(cfg, dcfg, a, stream);     // the planner half of configs[3]
    return launch_one<5, 0, 1>(cfg, dcfg, a, stream);
#endif
}

}  // namespace lpvmpc
