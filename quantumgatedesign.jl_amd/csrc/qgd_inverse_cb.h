// qgd_inverse_cb.h -- Np = 64: inverse and step propagator by COLUMN-BLOCK Gauss-Jordan elimination of [L_n | R_{n-1}]
// (included by qgd_k_inverse.hip; replaces the solve of forward_evolution.jl:209-220 for all right-hand sides at once).
//
// k_inverse_mfma (qgd_k_inverse.hip, now the fallback) eliminates 4 pivots per panel with every wave of the workgroup
// taking part in every panel: 16 panels x 3 barriers, each wave running the per-panel bookkeeping 16 times, and the product
// P = L^-1 R as a second phase that re-stages the inverse through LDS.  Here:
//
//   * wave v OWNS column block v (16 complex columns) of L and of R over all 64 rows: 16 accumulator tiles.  The right
//     operand of every rank-16 update -- row block k of the wave's own columns -- is already in its registers in B-operand
//     order (accumulator register s of a 16x16x4 f64 tile holds rows 4s..4s+3 = k-step s): no publishing of pivot rows.
//   * block step k: the owner of column block k eliminates its 64 x 16 panel ALONE (four 4-pivot sub-panels: lane = row
//     Gauss-Jordan chain on the 64 x 4 sub-panel, rank-4 MFMA update of the panel; exchanges through LDS inside the
//     wave, no workgroup barrier).  It publishes the multiplier block G (rows of block k: D^-1, other rows: -F D^-1) in
//     natural order; then every wave applies M[:, J] += (G - I_K) M[K, J] to its other tiles: 128 MFMAs per wave and block
//     step, left operands read from LDS (one ds_read_b128 per (re, im) pair, constant offsets).  Three barriers per block
//     step instead of twelve.
//   * R rides through the elimination as four more column blocks, so P = L^-1 R needs no product phase, no staging of the
//     inverse and no loads inside a loop; the MFMA count is the same (2048 per matrix).
//   * the block steps overlap: the NEXT owner applies step k to its L tiles only and goes straight into its panel; its R
//     tiles take steps k and k+1 together after it has published (two multiplier blocks live in LDS, by step parity).
//     While it eliminates, the other three waves do both halves of step k.
//   * pivots: first attempt on the diagonal (no search, pivot rows straight from the registers); a multiplier beyond
//     CB_GROWTH_STATIC restarts the matrix with partial pivoting confined to the 16 rows of the diagonal block (implicit:
//     rows never move, the permutation is resolved when the multiplier block is published).  A pivot confined to a
//     16 x 16 tile can still be small although the matrix is well conditioned: beyond CB_GROWTH (or on a zero pivot) the
//     matrix is reported to the caller, which runs the fully pivoted elimination of k_inverse_mfma on it.
//   * two instantiations: ONE = every workgroup of the launch is resident at once (256 < matrices <= 768) and all of them
//     run in step -- there the first owner loads its R block after its panel, the other waves hold their loads back and L^-1
//     leaves before the last R update (cb_wave); in longer launches the same measures cost time and are compiled out.
//   * registers: 128 accumulators of 168 (three workgroups per CU).  While a wave eliminates its panel half of its R tiles
//     wait in LDS, and every phase derives its lane constants from an opaque copy of the lane id, so that nothing but the
//     accumulators is alive across phases -- left to itself the compiler spilled several hundred registers into the
//     pivot chain and the updates (134 us instead of 98 for 550 matrices).
#pragma once

// wave priorities: the panel of the current owner, and the next owner from its L update to the end of its publish (the critical
// path of a matrix).  550 matrices: 86.0 us without, 83.4 with 1 / 0, 82.2 with 2 / 1, 82.5-83.4 with 3 / 2.
#define CB_PRIO_PANEL 2
#define CB_PRIO_CRIT 1
#define CB_GROWTH_STATIC 8.0
#define CB_GROWTH 256.0
// waves 1-3 hold their loads back by this many 64-cycle units (550 matrices, kernel alone: 80.8 us with 0, 79.9 with 8, 81.5 with 32,
// 82.1 with 64; 83.5 with wave 0 loading its R block up front and L^-1 stored with the rest at the end)
#define CB_FRONT_SLEEP 8
#define CB_ONE_ROUND 768       // 256 CUs x 3 workgroups
#define CB_ONE_ALONE 256       // (one workgroup per CU: nothing to queue behind -- 46.6 us without the measures, 47.3 with)

// LDS map (doubles).  G: a multiplier block, [column pair][row][column & 1][re, im] = 2048 (A-operand order: the 32 lanes
// (row c16, k = 0..1) of a 16-byte read are 32 consecutive slots); two of them, by block-step parity.
#define CB_G      2048
#define CB_O_F    (2 * CB_G)          // the current sub-panel, [column pair][row][column & 1][re, im]: its columns, then its multipliers
#define CB_O_PROW (CB_O_F + 512)      // row block k of the panel, [16 rows][32] (pivoted attempt)
#define CB_WORK   (CB_O_PROW + 512)
#define CB_LDP    65

#ifdef CB_PROFILE       // scripts/ubench/inverse_cb_bench.hip -DCB_PROFILE: s_memtime stamps of workgroup 0 (timing only, same results)
__device__ unsigned long long g_cb_prof[4][8][8];
#define CB_STAMP(K, i) do { if (blockIdx.x == 0 && (lane0 & 63) == 0) g_cb_prof[w][K][i] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_cb_place[4096][4][4];      // per workgroup and wave: HW_ID, XCC_ID, first and last stamp
#ifndef CB_PK
#define CB_PK 0
#endif
#define CB_PSTAMP(i) do { if (K == CB_PK && sp < 2 && blockIdx.x == 0 && lane == 0) g_cb_prof[0][5 + sp][i] = __builtin_amdgcn_s_memtime(); } while (0)
#define CB_PLACE(i) do { if (blockIdx.x < 4096 && (lane0 & 63) == 0) { unsigned long long *q_ = g_cb_place[blockIdx.x][lane0 >> 6]; \
        if ((i) == 0) { q_[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4); q_[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20); } \
        q_[2 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define CB_PLACE(i) do { } while (0)
#define CB_PSTAMP(i) do { } while (0)
#define CB_STAMP(K, i) do { } while (0)
#endif

// The lane id behind an opaque copy: what a phase derives from it (LDS offsets, masks) is then computed where it is used
// instead of once for the whole kernel.
__device__ __forceinline__ int cb_opaque(int v) { asm volatile("" : "+v"(v)); return v; }

__device__ __forceinline__ unsigned row16_max_u32(unsigned key)      // lane 15 of each row of 16 lanes gets the row maximum
{
    // v_max_u32 with the DPP source modifier: one instruction per step (the compiler emits mov + mov_dpp + max); a lane
    // whose source lies outside its row keeps its value.  A DPP read of a VGPR written by the previous vector instruction
    // needs two wait states.
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 0"
                 : "+v"(key));
    return key;
}

// [Bre | Bim] -> [-Bim | Bre]: the value of lane c16 ^ 8 (row_ror:8), sign flipped in the lanes c16 < 8 (smask = 0x80000000 there)
__device__ __forceinline__ double cb_swapneg(double v, int smask)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x128, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x128, 0xF, 0xF, false);
    return __hiloint2double(hi ^ smask, lo);
}

typedef double d2 __attribute__((ext_vector_type(2)));

// The owner's block step: in-place Gauss-Jordan on its 64 x 16 panel M (4 row blocks x 2 tiles), pivots among the rows of
// block K (STATIC: on the diagonal).  Leaves the multiplier block in M, rows and columns of block K in pivot order (rhoL,
// rinvL; cb_publish resolves it).  Returns nonzero (wave-uniform) when the attempt has to be abandoned.
template <int K, bool STATIC>
__device__ __forceinline__ int cb_panel(d4 (&M)[4][2], double *smem, int *rhoL, int *rinvL, const int lane0)
{
    const int lane = cb_opaque(lane0) & 63, c16 = lane & 15, kk = lane >> 4, smask = (c16 < 8) ? (int)0x80000000 : 0;
    double *Fm = smem + CB_O_F, *Prow = smem + CB_O_PROW;
    bool used = (lane >> 4) != K;
    double gmax = 0.0;
    int sing = 0;
    #pragma unroll
    for (int sp = 0; sp < 4; sp++) {
        const int ct = sp >> 1, q0 = (sp & 1) * 4;
        const int s_ = (c16 & 7) - q0;
        const bool mine = s_ >= 0 && s_ < 4;
        const int cofs = ((s_ >> 1) & 1) * 256 + kk * 4 + (s_ & 1) * 2 + (c16 >> 3);
        CB_PSTAMP(0);
        // ---- the sub-panel's columns (and, pivoted attempt, row block K of the panel) go to LDS
        if (mine) {
            #pragma unroll
            for (int i = 0; i < 4; i++)
                #pragma unroll
                for (int r = 0; r < 4; r++) Fm[cofs + (16 * i + 4 * r) * 4] = M[i][ct][r];
        }
        if (!STATIC) {
            #pragma unroll
            for (int c2 = 0; c2 < 2; c2++)
                #pragma unroll
                for (int r = 0; r < 4; r++) Prow[(kk + 4 * r) * 32 + 16 * c2 + c16] = M[K][c2][r];
        }
        wave_lds_fence();
        // ---- Gauss-Jordan on the 64 x 4 sub-panel, lane = row
        double xr[4], xi[4];
        #pragma unroll
        for (int h = 0; h < 2; h++) {
            const d4 v = *(const d4 *)(Fm + (h * 64 + lane) * 4);
            xr[2 * h] = v[0]; xi[2 * h] = v[1]; xr[2 * h + 1] = v[2]; xi[2 * h + 1] = v[3];
        }
        CB_PSTAMP(1);
        int pr[4];
        #pragma unroll
        for (int s = 0; s < 4; s++) {
            int p;
            if (STATIC) {
                p = 16 * K + 4 * sp + s;
            } else {
                const double a1 = __builtin_fabs(xr[s]) + __builtin_fabs(xi[s]);
                const unsigned mag = (unsigned)__double2hiint(a1);
                unsigned key = used ? 0u : ((mag & ~63u) | (unsigned)(63 - lane));
                key = row16_max_u32(key);
                const unsigned uk = (unsigned)__builtin_amdgcn_readlane((int)key, 16 * K + 15);
                p = 63 - (int)(uk & 63u);
                sing |= ((uk >> 6) == 0u) ? 1 : 0;
            }
            pr[s] = p;
            const bool isp = lane == p;
            used = used || isp;
            double yr[4], yi[4];
            #pragma unroll
            for (int q = 0; q < 4; q++) { yr[q] = lane_read(xr[q], p); yi[q] = lane_read(xi[q], p); }
            const double den = fast_rcp(yr[s] * yr[s] + yi[s] * yi[s]);
            // A pivot that is exactly zero (or whose |p|^2 leaves the range of a double) gives den = inf and NaN multipliers,
            // which the growth check below cannot see (fmax drops a NaN): the matrix is regular, only this pivot ORDER fails.
            // Finite pivot, non-finite reciprocal => give the attempt up.  (A pivot that is itself non-finite -- coefficients
            // that overflowed -- goes through: the results are then non-finite, as the reference's would be.)
            if (STATIC) sing |= (!(den * 0.0 == 0.0) && (yr[s] * 0.0 + yi[s] * 0.0 == 0.0)) ? 1 : 0;
            const double ir = yr[s] * den, ii = -yi[s] * den;
            const double delta = isp ? 1.0 : 0.0;
            const double fr = xr[s] - delta, fi = xi[s];
            const double gr = fi * ii - fr * ir, gi = -(fr * ii + fi * ir);       // -(f - delta) / pivot
            #pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q == s) continue;
                xr[q] = __builtin_fma(-gi, yi[q], __builtin_fma(gr, yr[q], xr[q]));
                xi[q] = __builtin_fma(gi, yr[q], __builtin_fma(gr, yi[q], xi[q]));
            }
            xr[s] = delta + gr; xi[s] = gi;
            // growth: the multipliers of THIS pivot (a pair of tiny pivots can undo each other's blow-up within a sub-panel)
            gmax = __builtin_fmax(gmax, __builtin_fmax(__builtin_fabs(gr), __builtin_fabs(gi)));
        }
        CB_PSTAMP(2);
        #pragma unroll
        for (int h = 0; h < 2; h++) {
            d4 v; v[0] = xr[2 * h]; v[1] = xi[2 * h]; v[2] = xr[2 * h + 1]; v[3] = xi[2 * h + 1];
            *(d4 *)(Fm + (h * 64 + lane) * 4) = v;
        }
        if (!STATIC && lane == 0) {
            #pragma unroll
            for (int s = 0; s < 4; s++) { rhoL[4 * sp + s] = pr[s] - 16 * K; rinvL[pr[s] - 16 * K] = 4 * sp + s; }
        }
        wave_lds_fence();
        CB_PSTAMP(3);
        // ---- rank-4 update of the panel: M += A M[P, :], A = multipliers (minus the identity on the pivot rows)
        {
            const int prm = STATIC ? 16 * K + 4 * sp + kk : (kk == 0) ? pr[0] : (kk == 1) ? pr[1] : (kk == 2) ? pr[2] : pr[3];
            double b1[2], b2[2];
            #pragma unroll
            for (int c2 = 0; c2 < 2; c2++) {
                if (STATIC) {       // pivot rows 4 sp .. 4 sp + 3 of block K: accumulator register sp, k index = kk
                    b1[c2] = M[K][c2][sp];
                    b2[c2] = cb_swapneg(b1[c2], smask);
                } else {
                    const double *prow = Prow + (prm - 16 * K) * 32 + 16 * c2;
                    b1[c2] = prow[c16];
                    const double tt = prow[c16 ^ 8];
                    b2[c2] = __hiloint2double(__double2hiint(tt) ^ smask, __double2loint(tt));
                }
            }
            const int aofs = (kk >> 1) * 256 + c16 * 4 + (kk & 1) * 2;
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                const d2 a = *(const d2 *)(Fm + aofs + i * 64);
                double are = a[0];
                if (i == K) are -= (16 * K + c16 == prm) ? 1.0 : 0.0;
                #pragma unroll
                for (int c2 = 0; c2 < 2; c2++) {
                    M[i][c2] = MFMA(are, b1[c2], M[i][c2]);
                    M[i][c2] = MFMA(a[1], b2[c2], M[i][c2]);
                }
            }
            CB_PSTAMP(4);
            if (mine) {
                #pragma unroll
                for (int i = 0; i < 4; i++)
                    #pragma unroll
                    for (int r = 0; r < 4; r++) M[i][ct][r] = Fm[cofs + (16 * i + 4 * r) * 4];
            }
        }
        wave_lds_fence();       // (the next sub-panel rewrites Fm and Prow)
        CB_PSTAMP(5);
    }
    const int grow = !(gmax <= (STATIC ? CB_GROWTH_STATIC : CB_GROWTH));      // (non-finite multipliers do not count: fmax drops a NaN, the matrix goes through with non-finite results)
    return sing | (__builtin_amdgcn_ballot_w64(grow) != 0 ? 2 : 0);
}

// The owner's panel M after cb_panel, written to the multiplier block Gout in natural order -- panel[x][j] belongs at row
// rinv(x) (rows of block K) and column rho(j) -- minus the identity on block K's diagonal.  After a pivoted attempt the wave
// takes its own copy back in natural order.
template <int K, bool STATIC>
__device__ __forceinline__ void cb_publish(d4 (&M)[4][2], double *Gout, const int *rhoL, const int *rinvL, const int lane0)
{
    const int lane = cb_opaque(lane0) & 63, c16 = lane & 15, kk = lane >> 4, plane = c16 >> 3;
    #pragma unroll
    for (int ct = 0; ct < 2; ct++) {
        const int col = STATIC ? 8 * ct + (c16 & 7) : rhoL[8 * ct + (c16 & 7)];
        double *dst = Gout + (col >> 1) * 256 + (col & 1) * 2 + plane;
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                if (i == K) {
                    const int ri = STATIC ? kk + 4 * r : rinvL[kk + 4 * r];
                    dst[(16 * K + ri) * 4] = M[i][ct][r] - ((c16 < 8 && ri == col) ? 1.0 : 0.0);
                } else {
                    dst[(16 * i + kk + 4 * r) * 4] = M[i][ct][r];
                }
            }
        }
    }
    if (!STATIC) {
        wave_lds_fence();
        #pragma unroll
        for (int ct = 0; ct < 2; ct++) {
            const int col = 8 * ct + (c16 & 7);
            const double *src = Gout + (col >> 1) * 256 + (col & 1) * 2 + plane + kk * 4;
            #pragma unroll
            for (int i = 0; i < 4; i++)
                #pragma unroll
                for (int r = 0; r < 4; r++) {
                    double v = src[(16 * i + 4 * r) * 4];
                    if (i == K) v += (c16 < 8 && kk + 4 * r == col) ? 1.0 : 0.0;
                    M[i][ct][r] = v;
                }
        }
    }
}

// The owner's R block to a parking slab and back (lane-major: conflict-free, no address arithmetic).  The slab is the
// multiplier slot the owner will publish into: nobody reads it between barrier C of the previous step and barrier B.
template <bool STORE>
__device__ __forceinline__ void cb_park(d4 (&T)[4][2], double *slab, const int lane0)
{
    const int lane = cb_opaque(lane0) & 63;
    #pragma unroll
    for (int i = 0; i < 4; i++)
        #pragma unroll
        for (int ct = 0; ct < 2; ct++)
            #pragma unroll
            for (int r = 0; r < 4; r++) {
                double *q = slab + ((i * 2 + ct) * 4 + r) * 64 + lane;
                if (STORE) *q = T[i][ct][r]; else T[i][ct][r] = *q;
            }
}

// The rank-16 update of one column block (two tiles over four row blocks) with a published multiplier block:
// T[:, :] += (G - I_K) T[K, :].
template <int K>
__device__ __forceinline__ void cb_update(d4 (&T)[4][2], const double *G, const int lane0)
{
    // One tile column at a time and the left operands of two k-steps at a time: 128 accumulator registers leave room for
    // little else, and a wave that waits for its LDS reads here leaves the SIMD to the other two workgroups.  The
    // scheduling barriers keep the compiler from hoisting all the reads.  Row block K goes last: until then its
    // accumulator registers ARE the right operand.
    const int lane = cb_opaque(lane0) & 63, c16 = lane & 15, kk = lane >> 4, smask = (c16 < 8) ? (int)0x80000000 : 0;
    const int abase = (kk >> 1) * 256 + c16 * 4 + (kk & 1) * 2;
    #pragma unroll
    for (int ct = 0; ct < 2; ct++) {
        double b2[4];
        #pragma unroll
        for (int s = 0; s < 4; s++) b2[s] = cb_swapneg(T[K][ct][s], smask);
        const d4 b1 = T[K][ct];
        // eight groups of four MFMAs (row block, k-step pair); the left operands of group g + 1 are read before the MFMAs of group g
        d2 a[2], an[2];
        #pragma unroll
        for (int u = 0; u < 2; u++) a[u] = *(const d2 *)(G + abase + u * 512 + ((K + 1) & 3) * 64);
        #pragma unroll
        for (int g = 0; g < 8; g++) {
            const int i = (K + 1 + (g >> 1)) & 3, sh = g & 1;
            if (g < 7) {
                const int i2 = (K + 1 + ((g + 1) >> 1)) & 3, sh2 = (g + 1) & 1;
                #pragma unroll
                for (int u = 0; u < 2; u++) an[u] = *(const d2 *)(G + abase + (2 * sh2 + u) * 512 + i2 * 64);
            }
            #pragma unroll
            for (int u = 0; u < 2; u++) {
                T[i][ct] = MFMA(a[u][0], b1[2 * sh + u], T[i][ct]);
                T[i][ct] = MFMA(a[u][1], b2[2 * sh + u], T[i][ct]);
            }
            __builtin_amdgcn_sched_barrier(0);
            a[0] = an[0]; a[1] = an[1];
        }
    }
}

// Block step K of wave W (both compile-time: each wave runs its own straight-line program -- with the roles decided at run
// time the conditional updates of 64-register tile blocks cost several hundred spilled registers).  The next owner (W == K + 1) applies step K to its L tiles only and goes on to its panel; its R
// tiles take steps K and K + 1 after it has published.  Barrier A: the panel of step K is done and every wave has finished
// with the multiplier block of step K - 2 (same slot); B: block K is published; C: the owner of step K has caught up with
// step K - 1, whose slot the next owner now takes as its parking slab.  The first owner has nothing to park (its R block is
// loaded after its panel); L^-1 is final after a wave's update of step 3 and leaves before the wave's last R update.
#define CB_STEP(K) if (!failed) { \
        double *Gk_ = smem + ((K) & 1) * CB_G; \
        CB_STAMP(K, 0); \
        if (W == (K)) { \
            if ((K) > 0 || !ONE) cb_park<true>(MR, Gk_, lane0); \
            __builtin_amdgcn_s_setprio(CB_PRIO_PANEL); \
            const int f_ = cb_panel<K, STATIC>(ML, smem, rhoL, rinvL, lane0); \
            __builtin_amdgcn_s_setprio(CB_PRIO_CRIT); \
            if (f_ && (lane0 & 63) == 0) bad = f_; \
            if ((K) > 0 || !ONE) cb_park<false>(MR, Gk_, lane0); \
        } \
        CB_STAMP(K, 1); \
        lds_barrier(); \
        if (W == (K)) { cb_publish<K, STATIC>(ML, Gk_, rhoL, rinvL, lane0); __builtin_amdgcn_s_setprio(0); if ((K) == 0 && ONE) load_R(); } \
        lds_barrier(); \
        CB_STAMP(K, 2); \
        failed = __builtin_amdgcn_readfirstlane(bad); \
        if (!failed) { \
            if (W == (K) + 1) __builtin_amdgcn_s_setprio(CB_PRIO_CRIT);      /* (the next owner: its L update and its panel are the critical path) */ \
            if (W != (K)) cb_update<K>(ML, Gk_, lane0); \
            if ((K) > 0 && W == (K)) cb_update<((K) > 0 ? (K) - 1 : 0)>(MR, smem + (((K) + 1) & 1) * CB_G, lane0); \
            if ((K) == 3 && ONE) store_Linv(); \
            CB_STAMP(K, 3); \
            lds_barrier(); \
            if (W != (K) + 1) cb_update<K>(MR, Gk_, lane0); \
        } \
        CB_STAMP(K, 4); \
    }

// What a wave program gets: everything wave-uniform (the fields come back through v_readfirstlane in the callee: a function's
// arguments arrive in vector registers).
struct CbArgs {
    const double *L, *R;
    double *LinvT, *Pr, *Pc;
    int *status, *fallbacks;
    double *smem;
    int *rho, *rinv, *bad;
    int n;
};
// A pointer into LDS, said to be one: cb_retry is a real function with two callers (the two kernels), so nothing tells the compiler
// where its pointer arguments point -- left generic, every LDS access in it became a flat_load / flat_store (2800 of them, and the
// pivoted attempt went from 124 to 151 us for 550 matrices).  The low word of a generic LDS address is the LDS offset (what the
// compiler's own generic -> local cast computes, behind a null check that brought the flat instructions back when it was used here).
template <typename T> __device__ __forceinline__ T *cb_lds(T *p)
{
    typedef __attribute__((address_space(3))) T lds_T;
    const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)p);
    return (T *)(lds_T *)(unsigned long long)off;
}
template <typename T> __device__ __forceinline__ T *cb_uniform(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __attribute__((noinline, noreturn)) void cb_fallback(const double *L, const double *R, double *LinvT, double *Pr, double *Pc, int n,
                                                                int *status, int *fallbacks, double *smem, int *rho, int *rinv, int front);
__device__ __attribute__((noinline)) void front_relayout(const double *L, const double *R, double *Pr, double *Pc, int n);

// The program of wave W: a real function, one per wave and attempt, that never returns -- it ends the wave when its outputs
// are written and hands a matrix it has to give up to the next attempt (diagonal pivots -> pivots inside the diagonal tile
// -> k_inverse_mfma's elimination).  Inlined into one kernel the three bodies cost the first one several hundred spilled
// registers; as functions that return they would save and restore 64 callee-saved registers per call.
template <int W, bool STATIC, bool ONE, bool FRONT>
__device__ __forceinline__ int cb_wave(const double *L_, const double *R_, double *LinvT_, double *Pr_, double *Pc_, int n_,
                                       int *status_, int *fallbacks_, double *smem_, int *rhoL_, int *rinvL_, int *bad_)
{
    const double *__restrict__ L = cb_uniform(L_), *__restrict__ R = cb_uniform(R_);
    double *__restrict__ LinvT = cb_uniform(LinvT_), *__restrict__ Pr = cb_uniform(Pr_), *__restrict__ Pc = cb_uniform(Pc_);
    double *smem = cb_lds(smem_);
    int *rhoL = cb_lds(rhoL_), *rinvL = cb_lds(rinvL_);
    (void)status_; (void)fallbacks_;
    int &bad = *cb_lds(bad_);
    const int n = __builtin_amdgcn_readfirstlane(n_);
    constexpr int NP = 64, PW = 128, w = W;
    const int lane0 = threadIdx.x;
    const size_t panel = (size_t)NP * PW, pl = (size_t)NP * NP;
    d4 ML[4][2], MR[4][2];
    CB_PLACE(0);
    // ONE: a launch whose workgroups are all resident at once (three per CU, CB_ONE_ROUND) runs them in step -- the loads of all of
    // them stand in one queue at the start, their stores at the end.  There the measures below pay (83.5 -> 80.5 us for the 550
    // matrices of the headline); in a longer launch, where workgroups start as others finish, they cost 4-6 % (1100 matrices: 148
    // -> 157 us, 2200: 248 -> 258) and are compiled out.  (As a run-time switch: 200 spilled registers, 86 us.)
    // Loads: every wave its L block first.  The first owner (wave 0) starts its panel as soon as that has arrived -- its R block
    // is loaded after the panel (no parking in step 0) -- and the other waves hold their loads back a little: all workgroups of
    // the launch start together, and the panels of step 0 wait for whatever stands in the queue in front of their 16 KB.
    auto load_R = [&]() {
        const int lane = cb_opaque(lane0) & 63, c16 = lane & 15, kk = lane >> 4;
        const __amdgpu_buffer_rsrc_t rr = buffer_of(R + (size_t)(n - 1) * panel + 32 * w);
        const int voff = (kk * PW + c16) * 8;
        #pragma unroll
        for (int i = 0; i < 4; i++)
            #pragma unroll
            for (int ct = 0; ct < 2; ct++)
                #pragma unroll
                for (int r = 0; r < 4; r++) MR[i][ct][r] = buffer_load_f64(rr, voff, ((16 * i + 4 * r) * PW + 16 * ct) * 8);
    };
    {
        const int lane = cb_opaque(lane0) & 63, c16 = lane & 15, kk = lane >> 4;
        const __amdgpu_buffer_rsrc_t rl = buffer_of(L + (size_t)n * panel + 32 * w);
        const int voff = (kk * PW + c16) * 8;
        if (W != 0 && ONE) __builtin_amdgcn_s_sleep(CB_FRONT_SLEEP);
        #pragma unroll
        for (int i = 0; i < 4; i++)
            #pragma unroll
            for (int ct = 0; ct < 2; ct++)
                #pragma unroll
                for (int r = 0; r < 4; r++) ML[i][ct][r] = buffer_load_f64(rl, voff, ((16 * i + 4 * r) * PW + 16 * ct) * 8);
    }
    if (W != 0 || !ONE) load_R();
    // L^-1, row-major planes (left operand of lambda = L^-H y): columns 16w..16w+15 of every row, straight from the registers
    // (issued after the wave's last L update, in front of its last R update)
    auto store_Linv = [&]() {
        const int lane = cb_opaque(lane0) & 63, c16 = lane & 15, kk = lane >> 4;
        const __amdgpu_buffer_rsrc_t rT = buffer_of(LinvT + (size_t)n * 2 * pl + 16 * w);
        const int voff = ((c16 >> 3) * (int)pl + kk * NP + (c16 & 7)) * 8;
        #pragma unroll
        for (int i = 0; i < 4; i++)
            #pragma unroll
            for (int ct = 0; ct < 2; ct++)
                #pragma unroll
                for (int r = 0; r < 4; r++) buffer_store_f64(ML[i][ct][r], rT, voff, ((16 * i + 4 * r) * NP + 8 * ct) * 8);
    };
    CB_STAMP(4, 0);
    if (lane0 == 0) bad = 0;
    __syncthreads();
    CB_STAMP(4, 1);
    int failed = 0;
    CB_STEP(0)
    CB_STEP(1)
    CB_STEP(2)
    CB_STEP(3)
    if (failed) return failed;
    CB_STAMP(4, 2);
    lds_barrier();              // the multiplier blocks are dead: their space stages the column-major planes of P
    const int lane = cb_opaque(lane0) & 63, c16 = lane & 15, kk = lane >> 4;
    if (!ONE) store_Linv();
    if (!FRONT) {   // P, row-major panel (left operand of the adjoint sweep as P^H)
        const __amdgpu_buffer_rsrc_t rP = buffer_of(Pr + (size_t)(n - 1) * panel + 32 * w);
        const int voff = (kk * PW + c16) * 8;
        #pragma unroll
        for (int i = 0; i < 4; i++)
            #pragma unroll
            for (int ct = 0; ct < 2; ct++)
                #pragma unroll
                for (int r = 0; r < 4; r++) buffer_store_f64(MR[i][ct][r], rP, voff, ((16 * i + 4 * r) * PW + 16 * ct) * 8);
    } else {
        // the fused front (qgd_front.h): the elimination ran on [L^H | R^H], MR = Y = S^H.  The forward sweep's left operand, S in
        // column-major planes, is conj(Y) in ROW-major planes -- straight from the registers like L^-1 above, the sign of the
        // imaginary plane flipped in the lanes that hold it
        const __amdgpu_buffer_rsrc_t rS = buffer_of(Pc + (size_t)(n - 1) * 2 * pl + 16 * w);
        const int voff = ((c16 >> 3) * (int)pl + kk * NP + (c16 & 7)) * 8, nmask = (c16 >= 8) ? (int)0x80000000 : 0;
        #pragma unroll
        for (int i = 0; i < 4; i++)
            #pragma unroll
            for (int ct = 0; ct < 2; ct++)
                #pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double v = MR[i][ct][r];
                    buffer_store_f64(__hiloint2double(__double2hiint(v) ^ nmask, __double2loint(v)), rS, voff, ((16 * i + 4 * r) * NP + 8 * ct) * 8);
                }
    }
    {   // P, column-major planes (left operand of the forward sweep): the wave's 16 columns are contiguous there; one plane
        // at a time through the wave's own LDS slab.  FRONT: the adjoint sweep's left operand, S as a row-major panel, is
        // conj(Y) as a COLUMN-major panel -- the same staging, the rows of 64 written as eight runs of (8 re | 8 im)
        double *stage = smem + w * (16 * CB_LDP);
        #pragma unroll
        for (int p = 0; p < 2; p++) {
            if ((c16 >> 3) == p) {
                #pragma unroll
                for (int i = 0; i < 4; i++)
                    #pragma unroll
                    for (int ct = 0; ct < 2; ct++)
                        #pragma unroll
                        for (int r = 0; r < 4; r++) stage[(8 * ct + (c16 & 7)) * CB_LDP + 16 * i + kk + 4 * r] = MR[i][ct][r];
            }
            wave_lds_fence();
            if (!FRONT) {
                const __amdgpu_buffer_rsrc_t rC = buffer_of(Pc + (size_t)(n - 1) * 2 * pl + p * pl + (size_t)16 * NP * w);
                #pragma unroll
                for (int q = 0; q < 16; q++) buffer_store_f64(stage[q * CB_LDP + lane], rC, lane * 8, q * 64 * 8);
            } else {
                const __amdgpu_buffer_rsrc_t rC = buffer_of(Pr + (size_t)(n - 1) * panel + (size_t)16 * PW * w + 8 * p);
                const int voff = ((lane >> 3) * 16 + (lane & 7)) * 8;
                #pragma unroll
                for (int q = 0; q < 16; q++) { const double v = stage[q * CB_LDP + lane]; buffer_store_f64(p ? -v : v, rC, voff, q * PW * 8); }
            }
            wave_lds_fence();
        }
    }
    CB_STAMP(4, 3);
    CB_PLACE(1);
    return 0;
}

template <bool STATIC, bool ONE, bool FRONT>
__device__ __forceinline__ int cb_attempt(const double *L, const double *R, double *LinvT, double *Pr, double *Pc, int n,
                                          int *status, int *fallbacks, double *smem, int *rhoL, int *rinvL, int *bad)
{
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: return cb_wave<0, STATIC, ONE, FRONT>(L, R, LinvT, Pr, Pc, n, status, fallbacks, smem, rhoL, rinvL, bad);
    case 1: return cb_wave<1, STATIC, ONE, FRONT>(L, R, LinvT, Pr, Pc, n, status, fallbacks, smem, rhoL, rinvL, bad);
    case 2: return cb_wave<2, STATIC, ONE, FRONT>(L, R, LinvT, Pr, Pc, n, status, fallbacks, smem, rhoL, rinvL, bad);
    default: return cb_wave<3, STATIC, ONE, FRONT>(L, R, LinvT, Pr, Pc, n, status, fallbacks, smem, rhoL, rinvL, bad);
    }
}

// Second attempt and last resort, as ONE real function that ends the wave (a call in the kernel's cold tail; nothing is alive
// across it).
__device__ __attribute__((noinline, noreturn)) void cb_retry(const double *L, const double *R, double *LinvT, double *Pr, double *Pc, int n,
                                                             int *status, int *fallbacks, double *smem, int *rhoL, int *rinvL, int *bad, int front)
{
    if (fallbacks && threadIdx.x == 0) atomicAdd(fallbacks, 1);                 // (matrices not done by the diagonal attempt)
    if (cb_attempt<false, true, false>(L, R, LinvT, Pr, Pc, n, status, fallbacks, smem, rhoL, rinvL, bad)) {
        __syncthreads();
        cb_fallback(L, R, LinvT, Pr, Pc, n, status, fallbacks, smem, rhoL, rinvL, front);
    }
    // (the pivoted stages write P in the layouts of the two-point form; the fused front's consumers read conj(P^T): re-laid out here,
    //  through the panels of L^H and R^H the elimination is done with -- a cold path)
    if (front) front_relayout(L, R, Pr, Pc, n);
    __builtin_amdgcn_endpgm();
}

// The workgroup's entry.  fallbacks (or null): [0] += matrices not done by the diagonal attempt, [1] += of these, by the last
// resort, [2] != 0: skip the diagonal attempt (k_tables: for the 32 evaluations after one that gave up more than a quarter of
// its matrices there -- a problem whose step matrices are not diagonally dominant pays for the failed attempt once in 33).
template <bool ONE, bool FRONT = false>
__device__ __forceinline__ void inverse_cb_body(const double *L, const double *R, double *LinvT, double *Pr, double *Pc, const int n,
                                                int *status, int *fallbacks, double *smem, int *rhoL, int *rinvL, int *bad)
{
    const bool pivot_first = fallbacks && __builtin_amdgcn_readfirstlane(fallbacks[2]) != 0;
    if (pivot_first || cb_attempt<true, ONE, FRONT>(L, R, LinvT, Pr, Pc, n, status, fallbacks, smem, rhoL, rinvL, bad)) {
        __syncthreads();
        cb_retry(L, R, LinvT, Pr, Pc, n, status, fallbacks, smem, rhoL, rinvL, bad, FRONT ? 1 : 0);
    }
}
