// jacobi.inl — the l x l symmetric eigen-solve of the Rayleigh–Ritz steps (included by pca_solve.hip through iterate.inl, inside namespace srx).
//
// k_jacobi_eig2: two-sided cyclic Jacobi on the 64 x 64 projected matrix, round-robin ordering (32 disjoint rotations per
// round, 63 rounds per sweep), arranged so that a round costs ONE barrier, no index arithmetic and ~half the LDS traffic of
// round 2's 1024-thread solver (2300 clocks per round there: two barriers, a serial "make the 32 rotations" phase, B and U through LDS):
//
//  * SLOT SPACE.  The matrix lives in LDS in tournament positions: pair m is always slots (2m, 2m + 1), and every round ends
//    by moving the data one step round the circle (slot 0 fixed, c_0 = slot 1, c_i = slot 2i, c_{63-i} = slot 2i + 1;
//    c_i -> c_{i+1}).  Thread (I, J), I < J, reads its 2 x 2 block (pair I) x (pair J) from FIXED addresses (two 16-byte
//    reads) and writes the rotated entries to FIXED destinations (the permuted positions, [min][max] of the symmetric
//    storage): every address is a per-thread constant from a compile-time table (kJ2Map).  496 blocks on 256 threads.
//  * NO ROTATION PHASE.  The pivot of next round's pair (2m', 2m' + 1) is one of the four entries a fixed thread has just
//    computed, and the two diagonal entries it needs are the post-rotation diagonals of two pairs of THIS round — carried in
//    the rotation records (t, c, dp, dq), never in the matrix.  That thread makes the next rotation on the spot.  The 32
//    such threads are lanes 0..31 of wave 0 (the table puts their blocks first), so only one wave runs the rotation code.
//  * U IN REGISTERS.  The eigenvector matrix is not in LDS: wave 8 keeps U row-per-lane in registers (64 doubles per lane,
//    columns in PLAYER order) and applies the 32 rotations of a round as scaled rotations — column x is stored as
//    u_x / scale_x, scale_x the product of the c's of its rotations, so new_p = c (p - t q), new_q = c (q + t p) costs two
//    FMAs per rotation and row with the tangents pre-multiplied by the scale ratios (made by the thread that makes the
//    rotation).  Which players sit in pair m at round r is a compile-time constant in the fully unrolled 63-round body:
//    every register index is static; the wave's only LDS traffic are 32 broadcast 16-byte reads per round.
//
// Outputs: U (L x L row-major) receives eigenvector c in COLUMN c, eigenvalues
// descending in theta; rows / columns >= n are 0, theta[c >= n] = 0; kStatEig on non-convergence after 30 sweeps.
// `off_tol2`: the sweep loop stops when sum_{i<j} a_ij^2 <= off_tol2 * sum_i a_ii^2, measured before a sweep's first round;
// the solve leaves after that round (the state is one round better than measured; players are back in their own slots).
// `n_want` (round 5): the sum on the left runs over the pairs with at least one WANTED index — the n_want largest diagonal
// entries, ranked one sweep earlier (the ranks of the leading entries do not move once the first sweep is through).  A Ritz
// step asks for n_pc pairs of a block with 14 guard columns whose Ritz values crowd together (theta_64 / theta_50 = 0.97 on the
// bench matrix): the guard x guard block is a dense perturbation of a multiple of the identity and takes the cyclic Jacobi its
// full ~5 sweeps, while everything a wanted pair's residual sees — its row — is at rounding level two sweeps earlier.
// Rotations between two guard indices leave the norm of every wanted row unchanged, so stopping early costs the wanted pairs
// nothing; U stays orthogonal whatever is left in the guard block, whose diagonal entries are then Rayleigh quotients, not
// eigenvalues: theta_l, the filter's bound, errs upward (a valid bound).  n_want >= n: the plain criterion.

constexpr int kJ2BWaves = 5;                              // wave 0: one block per lane, the 32 rotation makers among them; waves 1-4: two
constexpr int kJ2BThreads = kJ2BWaves * kWave;            // blocks per thread.  6 waves in all: at most two per SIMD, i.e. 256 registers
constexpr int kJ2Threads = (kJ2BWaves + 1) * kWave;      // a thread — the U wave wants 128 for its row and 64 for tangents in flight
constexpr int kJ2Blocks = 496;                            // 2 x 2 blocks above the block diagonal: 32 * 31 / 2
constexpr int kJ2Rounds = L - 1;
constexpr int kJ2Ld = L + 2;                              // row stride of the matrix in LDS (even: 16-byte aligned pairs)

// circle position <-> slot, and the move of one round
constexpr int j2_slot_of_c(int i) { return i == 0 ? 1 : (i <= L / 2 - 1 ? 2 * i : 2 * ((L - 1) - i) + 1); }
constexpr int j2_c_of_slot(int s) { return s == 1 ? 0 : ((s & 1) ? (L - 1) - (s - 1) / 2 : s / 2); }
constexpr int j2_pi(int s) { return s == 0 ? 0 : j2_slot_of_c((j2_c_of_slot(s) + 1) % (L - 1)); }
constexpr int j2_mod(int a) { return ((a % (L - 1)) + (L - 1)) % (L - 1); }
// the players (= initial slots) in pair m at round r
constexpr int j2_player_p(int m, int r) { return m == 0 ? 0 : j2_slot_of_c(j2_mod(m - r)); }
constexpr int j2_player_q(int m, int r) { return m == 0 ? j2_slot_of_c(j2_mod(-r)) : j2_slot_of_c(j2_mod((L - 1) - m - r)); }

struct J2Map {
    unsigned char I[kJ2Blocks], J[kJ2Blocks];
    unsigned short dst[kJ2Blocks][4];      // where the entry (2I + h, 2J + g) goes: index row * kJ2Ld + col, row < col
    signed char fm[kJ2Blocks];             // the pair of the next round whose pivot this block holds (-1: none); blocks 0..31 have fm = 0..31
    unsigned char fsel[kJ2Blocks];         // which entry (2 h + g) is that pivot
    unsigned char fswap[kJ2Blocks];        // 1: the column-side player (from pair J) takes the even slot (p') of the next pair
    unsigned short zdst[L / 2];            // where this round's pivot of pair m (zero after the rotation) goes
};
constexpr J2Map j2_make_map() {
    J2Map mp{};
    int n_fold = 0, n_rest = L / 2;
    for (int I = 0; I < L / 2; ++I)
        for (int J = I + 1; J < L / 2; ++J) {
            int fm = -1, fsel = 0, fswap = 0;
            unsigned short dst[4] = {0, 0, 0, 0};
            for (int h = 0; h < 2; ++h)
                for (int g = 0; g < 2; ++g) {
                    const int a = j2_pi(2 * I + h), b = j2_pi(2 * J + g);
                    const int lo = a < b ? a : b, hi = a < b ? b : a;
                    dst[2 * h + g] = (unsigned short)(lo * kJ2Ld + hi);
                    if ((lo & 1) == 0 && hi == lo + 1) {           // lands on a diagonal block: next round's pivot
                        fm = lo / 2;
                        fsel = 2 * h + g;
                        fswap = b == lo ? 1 : 0;
                    }
                }
            const int at = fm >= 0 ? fm : n_rest++;
            if (fm >= 0) ++n_fold;
            mp.I[at] = (unsigned char)I;
            mp.J[at] = (unsigned char)J;
            for (int e = 0; e < 4; ++e) mp.dst[at][e] = dst[e];
            mp.fm[at] = (signed char)fm;
            mp.fsel[at] = (unsigned char)fsel;
            mp.fswap[at] = (unsigned char)fswap;
        }
    for (int m = 0; m < L / 2; ++m) {
        const int a = j2_pi(2 * m), b = j2_pi(2 * m + 1);
        mp.zdst[m] = (unsigned short)((a < b ? a : b) * kJ2Ld + (a < b ? b : a));
    }
    if (n_fold != L / 2 || n_rest != kJ2Blocks) mp.I[0] = 255;      // (checked by the static_assert below)
    return mp;
}
__constant__ const J2Map kJ2Map = j2_make_map();
static_assert(j2_make_map().I[0] != 255 && j2_make_map().fm[0] == 0 && j2_make_map().fm[L / 2 - 1] == L / 2 - 1 &&
                  j2_make_map().fm[L / 2] == -1,
              "every pair of the next round takes its pivot from exactly one block");

struct alignas(16) J2Lds {
    double A[2][L][kJ2Ld];       // off-diagonal entries at [min][max], slot space; a round reads one copy and writes the other
                                 // (the destinations of one thread are the sources of others)
    double rec[2][L / 2][4];     // per pair of the round of that parity: t, c and the post-rotation diagonals dp, dq
    double tu[2][L / 2][2];      // the same rotations for the SCALED columns of U: (t scale_q / scale_p, t scale_p / scale_q)
    double scale[2][L];          // per slot: product of the c's of the rotations applied to that player's column of U
    double pend[2][L / 2];       // squared pivots the round of that parity annihilates
    double red[kJ2BWaves];
    int rank[L];
    int want[L];                 // per slot (= player, at a sweep's first round): 1 when its diagonal entry ranks among the wanted
    int flag;
};

__device__ __forceinline__ void j2_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// 1 / sqrt(x) and 1 / x for normal positive x, from the hardware seeds (v_rsq_f64 / v_rcp_f64) by one third-order
// correction each — the library's sqrt / rsqrt / division are 3-4x as many dependent operations, and this chain is the
// critical path of a Jacobi round
__device__ __forceinline__ double j2_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);                        // 2^-24 (bench_micro/rsq_precision.hip)
    const double e = __builtin_fma(-x * y, y, 1.0);                  // 1 - x y^2
    return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);   // y (1 + e/2 + 3 e^2/8): 2^-24 -> 5/16 e^3 = 2^-73
}
__device__ __forceinline__ double j2_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);                        // 2^-24
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);             // r (1 + e + e^2): e^3 = 2^-72
}

// rotation annihilating a_pq of the 2 x 2 block (app, apq; apq, aqq) with rows / columns combined as
// new_p = c p - s q, new_q = s p + c q:  t = s / c = sgn(b d) |b| / (|d| + hypot(b, d)), b = 2 apq, d = aqq - app.
__device__ __forceinline__ void j2_rotation(double app, double aqq, double apq, double& t, double& c, double& dp, double& dq) {
    const double b = 2.0 * apq, d = aqq - app;
    const double x = __builtin_fma(b, b, d * d);
    // branch-free: a square below the normal range (nothing to rotate at f64 resolution) goes through the same arithmetic
    // on a harmless argument and is replaced at the end, so that the scheduler can interleave independent work
    const bool live = x > 1e-290 && b != 0.0;
    const double xs = live ? x : 1.0;
    const double h = xs * j2_rsqrt(xs);
    const double u = fabs(d) + h;
    const double ab = ((d >= 0.0) == (b >= 0.0)) ? fabs(b) : -fabs(b);
    const double tt = ab * j2_rcp(u);
    const double cc = u * j2_rsqrt(__builtin_fma(u, u, b * b));
    t = live ? tt : 0.0;
    c = live ? cc : 1.0;
    dp = __builtin_fma(-t, apq, app);
    dq = __builtin_fma(t, apq, aqq);
}

typedef double j2_d2 __attribute__((ext_vector_type(2)));

template <int R>
struct J2Round {
    // the 32 rotations of round R applied to the row a lane holds; (p, q) = the players in pair m at round R: static
    // indices.  The tangent pairs are wave-uniform 16-byte reads (broadcast), fetched in groups of 8 with the next group in
    // flight while this one is applied (64 registers beside the row's 128: the kernel runs at most two waves per SIMD);
    // the compiler barriers keep it from hoisting all 32 loads (128 registers: spills)
    static __device__ __forceinline__ void apply(double (&u)[L], const double (*tup)[2]) {
        constexpr int kG = 8;
        j2_d2 tcur[kG], tnext[kG];
#pragma unroll
        for (int j = 0; j < kG; ++j) tcur[j] = *reinterpret_cast<const j2_d2*>(tup[j]);
#pragma unroll
        for (int g = 0; g < L / 2 / kG; ++g) {
            if (g + 1 < L / 2 / kG) {
#pragma unroll
                for (int j = 0; j < kG; ++j) tnext[j] = *reinterpret_cast<const j2_d2*>(tup[(g + 1) * kG + j]);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < kG; ++j) {
                const int m = g * kG + j;
                const int p = j2_player_p(m, R), q = j2_player_q(m, R);
                const double vp = u[p], vq = u[q];
                u[p] = __builtin_fma(-tcur[j].x, vq, vp);
                u[q] = __builtin_fma(tcur[j].y, vp, vq);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < kG; ++j) tcur[j] = tnext[j];
        }
    }
};

// rounds 1 .. 62 of a sweep for the U wave: each one = the round's rotations, then the round's barrier
template <int... Rs>
__device__ __forceinline__ void j2_u_rounds(double (&u)[L], const J2Lds& S, int& par, std::integer_sequence<int, Rs...>) {
    ((J2Round<Rs + 1>::apply(u, S.tu[par]), j2_barrier(), par ^= 1), ...);
}

__global__ __launch_bounds__(kJ2Threads) void k_jacobi_eig2(double* __restrict__ H /* k_gram1_part's sum: read once, left zeroed */, int n, double* __restrict__ U,
                                                            double* __restrict__ theta, int* __restrict__ status, double off_tol2,
                                                            int n_want) {
    static_assert(L == 64, "the block mapping is written for l = 64");
    extern __shared__ double j2_lds_raw[];
    J2Lds& S = *reinterpret_cast<J2Lds*>(j2_lds_raw);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int lane = tid & (kWave - 1);
    // H goes into the second copy of the matrix (free until round 0 writes it), all of a thread's loads in flight at once; then it is
    // symmetrised, the diagonal parked in scale[0]
    {
        double* const T = &S.A[1][0][0];
        constexpr int kPer = (L * L + kJ2Threads - 1) / kJ2Threads;
        double hv[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int e = tid + u * kJ2Threads;
            hv[u] = e < L * L ? H[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int e = tid + u * kJ2Threads;
            if (e < L * L) {
                T[e] = hv[u];
                H[e] = 0.0;
            }
        }
        j2_barrier();
        for (int e = tid; e < L * L; e += kJ2Threads) {
            const int a = e >> 6, b = e & 63;
            const bool in = a < n && b < n;
            if (a < b) S.A[0][a][b] = in ? 0.5 * (T[a * L + b] + T[b * L + a]) : 0.0;
            if (a == b) S.scale[0][a] = in ? T[e] : 0.0;
        }
    }
    if (tid < L) {
        S.scale[1][tid] = 1.0;
        S.want[tid] = 1;               // the first sweep's measurement: every pair
    }
    if (tid == 0) S.flag = 0;
    j2_barrier();
    // prologue: the rotations of round 0 from the matrix as loaded (pair m = slots 2m, 2m + 1 = players 2m, 2m + 1)
    if (tid < L / 2) {
        const int p = 2 * tid, q = p + 1;
        const double apq = S.A[0][p][q];
        const double app = S.scale[0][p], aqq = S.scale[0][q];
        double t, c, dp, dq;
        j2_rotation(app, aqq, apq, t, c, dp, dq);
        S.rec[0][tid][0] = t;
        S.rec[0][tid][1] = c;
        S.rec[0][tid][2] = dp;
        S.rec[0][tid][3] = dq;
        S.tu[0][tid][0] = t;             // all scales are 1 before the first round
        S.tu[0][tid][1] = t;
        S.scale[0][p] = c;
        S.scale[0][q] = c;
        S.pend[0][tid] = apq * apq;
    }
    j2_barrier();

    bool converged = false;
    int par = 0;                          // parity of the running round counter g = 63 * sweep + r
    // (s_setprio 3 for wave 0, the round's critical path: 184.1 against 183.9 us — the waves sit on different SIMDs)
    if (wave < kJ2BWaves) {
        // ---- B: wave 0 takes blocks 0 .. 63 of the table (one per lane), waves 1 .. 4 the other 432 (two per thread);
        // everything below is fixed for the life of the kernel ----
        constexpr int kCopy = L * kJ2Ld;                  // doubles per copy of the matrix
        double* const Ab = &S.A[0][0][0];
        int bI[2], bJ[2], s0[2], o00[2], o01[2], o10[2], o11[2];
        bool live[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int blk = wave == 0 ? (v == 0 ? tid : kJ2Blocks) : (tid + v * (kJ2BThreads - kWave));
            live[v] = blk < kJ2Blocks;
            const int bb = live[v] ? blk : 0;
            bI[v] = kJ2Map.I[bb];
            bJ[v] = kJ2Map.J[bb];
            s0[v] = (2 * bI[v]) * kJ2Ld + 2 * bJ[v];
            o00[v] = kJ2Map.dst[bb][0];
            o01[v] = kJ2Map.dst[bb][1];
            o10[v] = kJ2Map.dst[bb][2];
            o11[v] = kJ2Map.dst[bb][3];
        }
        // blocks 0 .. 31 (lanes 0 .. 31 of wave 0) hold the pivots of the next round's pairs 0 .. 31
        const bool fold = tid < L / 2;
        const int fm = tid & (L / 2 - 1), fsel = kJ2Map.fsel[fm], fswap = kJ2Map.fswap[fm];
        const int oz = kJ2Map.zdst[fm];
        // the next pair's players as seen from the block: x on the row side (slot 2I + h), y on the column side (slot 2J + g)
        const int fh = fsel >> 1, fg = fsel & 1;
        const int xs = 2 * bI[0] + fh, ys = 2 * bJ[0] + fg;
        int want_next = 1, sweep_count = 0;
        for (int sweep = 0; sweep < 31 && !converged; ++sweep) {
            sweep_count = sweep;
            double wq[2][4];                  // 1 where the entry's pair has a wanted index (ranks of the sweep before), else 0
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int wI0 = S.want[2 * bI[v]], wI1 = S.want[2 * bI[v] + 1], wJ0 = S.want[2 * bJ[v]], wJ1 = S.want[2 * bJ[v] + 1];
                wq[v][0] = (wI0 | wJ0) ? 1.0 : 0.0;
                wq[v][1] = (wI0 | wJ1) ? 1.0 : 0.0;
                wq[v][2] = (wI1 | wJ0) ? 1.0 : 0.0;
                wq[v][3] = (wI1 | wJ1) ? 1.0 : 0.0;
            }
            // one round: the blocks' rotation, next round's rotations made by the 32 fold threads; `measure` (round 0 only, a
            // compile-time copy of the body: the other 62 rounds carry none of it): this thread's share of the off-diagonal norm
            // BEFORE the round, over the pairs with a wanted index
            auto round_body = [&](auto measure) -> double {
                const int nxt = par ^ 1;
                double offsq = 0.0;
                const double* const Ar = Ab + par * kCopy;
                double* const Aw = Ab + nxt * kCopy;
                double val = 0.0;
                // (the rotation makers' extra operands, fetched with the block: nothing below waits for a second LDS round trip)
                double dx = 0.0, dy = 0.0, sx = 1.0, sy = 1.0;
                if (fold) {
                    dx = S.rec[par][bI[0]][2 + fh];       // post-rotation diagonals of x, y
                    dy = S.rec[par][bJ[0]][2 + fg];
                    sx = S.scale[par][xs];
                    sy = S.scale[par][ys];
                }
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    if (!live[v]) continue;
                    const j2_d2 rI = *reinterpret_cast<const j2_d2*>(S.rec[par][bI[v]]);
                    const j2_d2 rJ = *reinterpret_cast<const j2_d2*>(S.rec[par][bJ[v]]);
                    const j2_d2 r0 = *reinterpret_cast<const j2_d2*>(Ar + s0[v]), r1 = *reinterpret_cast<const j2_d2*>(Ar + s0[v] + kJ2Ld);
                    const double b00 = r0.x, b01 = r0.y, b10 = r1.x, b11 = r1.y;
                    const double tI = rI.x, tJ = rJ.x, cc = rI.y * rJ.y;
                    if constexpr (decltype(measure)::value)
                        offsq = __builtin_fma(wq[v][0] * b00, b00, __builtin_fma(wq[v][1] * b01, b01,
                                __builtin_fma(wq[v][2] * b10, b10, __builtin_fma(wq[v][3] * b11, b11, offsq))));
                    // rows: new_p = c (p - t q), new_q = c (q + t p); then the same on the columns; one common factor
                    const double t00 = __builtin_fma(-tI, b10, b00), t01 = __builtin_fma(-tI, b11, b01);
                    const double t10 = __builtin_fma(tI, b00, b10), t11 = __builtin_fma(tI, b01, b11);
                    const double n00 = cc * __builtin_fma(-tJ, t01, t00), n01 = cc * __builtin_fma(tJ, t00, t01);
                    const double n10 = cc * __builtin_fma(-tJ, t11, t10), n11 = cc * __builtin_fma(tJ, t10, t11);
                    Aw[o00[v]] = n00;
                    Aw[o01[v]] = n01;
                    Aw[o10[v]] = n10;
                    Aw[o11[v]] = n11;
                    if (v == 0) val = fsel == 0 ? n00 : fsel == 1 ? n01 : fsel == 2 ? n10 : n11;
                }
                if (fold) {
                    // the block holds the pivot of pair fm of the next round: its rotation, made here
                    Aw[oz] = 0.0;                         // this round's pivot of pair `tid`, annihilated: no block writes it
                    const double app = fswap ? dy : dx, aqq = fswap ? dx : dy;
                    const double scp = fswap ? sy : sx, scq = fswap ? sx : sy;
                    const double rqp = scq * j2_rcp(scp), rpq = scp * j2_rcp(scq);      // (independent of the rotation: beside it)
                    double t, c, dp, dq;
                    j2_rotation(app, aqq, val, t, c, dp, dq);
                    S.rec[nxt][fm][0] = t;
                    S.rec[nxt][fm][1] = c;
                    S.rec[nxt][fm][2] = dp;
                    S.rec[nxt][fm][3] = dq;
                    S.tu[nxt][fm][0] = t * rqp;
                    S.tu[nxt][fm][1] = t * rpq;
                    S.scale[nxt][2 * fm] = scp * c;
                    S.scale[nxt][2 * fm + 1] = scq * c;
                    S.pend[nxt][fm] = val * val;
                }
                return offsq;
            };
            // round 0: measured
            {
                const double off_mine = wave_sum(round_body(std::true_type{}));      // (all 64 lanes of the wave take part in the reduction)
                if (lane == 0) S.red[wave] = off_mine;
                j2_barrier();
                if (wave == 0) {
                    double off = (lane < kJ2BWaves ? S.red[lane] : 0.0) +
                                 (lane < L / 2 && (S.want[2 * lane] | S.want[2 * lane + 1]) ? S.pend[par][lane] : 0.0);
                    const double dv = S.rec[par][lane >> 1][2 + (lane & 1)];
                    double dg = dv * dv;
                    off = wave_sum(off);
                    dg = wave_sum(dg);
                    if (lane == 0) S.flag = !(off > off_tol2 * dg) ? 1 : (sweep >= 30 ? 2 : 0);
                    // (diagnostics, SRX_PCA_TRACE: the measured off-diagonal norm over the diagonal's, per sweep, as float bits)
                    if (lane == 0 && sweep < 32) status[16 + sweep] = __float_as_int((float)sqrt(off / dg));
                    // who is wanted at the NEXT sweep's measurement: rank of this slot's diagonal entry among the 64
                    int rk = 0;
                    for (int j = 0; j < L; ++j) {
                        const double other = S.rec[par][j >> 1][2 + (j & 1)];
                        rk += (other > dv || (other == dv && j < lane)) ? 1 : 0;
                    }
                    want_next = rk < n_want ? 1 : 0;
                }
                j2_barrier();
                if (wave == 0) S.want[lane] = want_next;       // (read again at the next sweep's start: barriers between)
                if (S.flag) {
                    converged = true;
                    break;
                }
                par ^= 1;
            }
            for (int r = 1; r < kJ2Rounds; ++r) {
                (void)round_body(std::false_type{});
                j2_barrier();
                par ^= 1;
            }
        }
        if (tid == 0 && S.flag == 2) atomicOr(status, kStatEig);
        if (tid == 0) status[1] = sweep_count;             // (diagnostics: SRX_PCA_TRACE prints the sweeps of the last eigen-solve)
        // eigenvalues: the diagonal after the last round applied (round 0 of a sweep: every player sits in its own slot),
        // descending; padded indices (>= n) go last
        if (tid < L) {
            const double mine = S.rec[par][tid >> 1][2 + (tid & 1)];
            int rk = 0;
            for (int j = 0; j < L; ++j) {
                if (j == tid) continue;
                const double other = S.rec[par][j >> 1][2 + (j & 1)];
                bool before;
                if (tid >= n) before = (j < n) || j < tid;
                else before = (j < n) && (other > mine || (other == mine && j < tid));
                rk += before ? 1 : 0;
            }
            S.rank[tid] = rk;
            theta[rk] = tid < n ? mine : 0.0;
        }
        j2_barrier();
    } else {
        // ---- U: one row per lane, in registers, columns in player order -------------------------------------------------
        double u[L];
#pragma unroll
        for (int c = 0; c < L; ++c) u[c] = c == lane ? 1.0 : 0.0;
        for (int sweep = 0; sweep < 31 && !converged; ++sweep) {
            // round 0 apart: the convergence flag is read after it
            J2Round<0>::apply(u, S.tu[par]);
            j2_barrier();
            j2_barrier();
            if (S.flag) {
                converged = true;
                break;
            }
            par ^= 1;
            j2_u_rounds(u, S, par, std::make_integer_sequence<int, kJ2Rounds - 1>{});
        }
        j2_barrier();                                      // the ranks are ready
#pragma unroll
        for (int c = 0; c < L; ++c)          // (par: the parity of the last round applied, as on the B side)
            U[(size_t)lane * L + S.rank[c]] = (lane < n && c < n) ? u[c] * S.scale[par][c] : 0.0;
    }
}
