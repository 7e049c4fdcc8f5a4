// log1p64.hpp — ln(1 + x) in f64 for the fused normalise + log1p transform, ~25 VALU operations per value.
//
// The reference computes log1p_transform on f64 values with libm's ln_1p (src/memory/processing/transform/mod.rs:38-42,
// <= 1 ulp).  ocml's f64 log1p costs ~150 instructions and made the f64 row pass VALU-bound (6.1 ms against 3.8 ms of
// memory time at c3); the per-gene moments of HighlyVariable(n) need the transformed values to f64 accuracy whatever the
// storage type, for every non-zero of the matrix.  Table-driven form for x >= 0 (counts scaled by a positive factor):
//   u = fl(1 + x) = 2^e m,  i = top 7 mantissa bits,  r = m * T[i].inv - 1  (|r| < 2^-7, fma),  T[i].inv = fl(1 / (1 + i/128))
//   ln(1 + x) = e ln2 + T[i].log + (r - r^2/2 + ... - r^8/8) + (x - (u - 1)) / u
// with T[i].log = -ln(T[i].inv) to 50 digits (so that the table pair is consistent), ln2 split in two, and the last term
// the rounding of 1 + x (a float reciprocal is enough for it).  Every term is >= 0 for x >= 0: no cancellation.  Measured
// against log1pl over 2^-30 .. 2^15: <= 2.7e-16 relative (2.5 ulp); tests/test_stats_gpu.py::test_log1p_f64_fast_accuracy.
// Negative, NaN and infinite arguments take ocml's log1p (never on count data).
#pragma once
#include <hip/hip_runtime.h>

namespace srx {

struct alignas(16) Log1pTabEntry { double inv, lg; };
static __device__ const Log1pTabEntry kLog1pTab[128] = {
{0x1.0000000000000p+0, 0x0.0p+0},
{0x1.fc07f01fc07f0p-1, 0x1.fe02a6b106799p-8},
{0x1.f81f81f81f820p-1, 0x1.fc0a8b0fc03c4p-7},
{0x1.f44659e4a4271p-1, 0x1.7b91b07d5b126p-6},
{0x1.f07c1f07c1f08p-1, 0x1.f829b0e7832f8p-6},
{0x1.ecc07b301ecc0p-1, 0x1.39e87b9febd68p-5},
{0x1.e9131abf0b767p-1, 0x1.77458f632dcffp-5},
{0x1.e573ac901e574p-1, 0x1.b42dd711971b9p-5},
{0x1.e1e1e1e1e1e1ep-1, 0x1.f0a30c01162a8p-5},
{0x1.de5d6e3f8868ap-1, 0x1.16536eea37ae3p-4},
{0x1.dae6076b981dbp-1, 0x1.341d7961bd1d0p-4},
{0x1.d77b654b82c34p-1, 0x1.51b073f06183cp-4},
{0x1.d41d41d41d41dp-1, 0x1.6f0d28ae56b4ep-4},
{0x1.d0cb58f6ec074p-1, 0x1.8c345d6319b23p-4},
{0x1.cd85689039b0bp-1, 0x1.a926d3a4ad562p-4},
{0x1.ca4b3055ee191p-1, 0x1.c5e548f5bc743p-4},
{0x1.c71c71c71c71cp-1, 0x1.e27076e2af2eap-4},
{0x1.c3f8f01c3f8f0p-1, 0x1.fec9131dbeabcp-4},
{0x1.c0e070381c0e0p-1, 0x1.0d77e7cd08e5bp-3},
{0x1.bdd2b899406f7p-1, 0x1.1b72ad52f67a2p-3},
{0x1.bacf914c1bad0p-1, 0x1.29552f81ff521p-3},
{0x1.b7d6c3dda338bp-1, 0x1.371fc201e8f75p-3},
{0x1.b4e81b4e81b4fp-1, 0x1.44d2b6ccb7d1cp-3},
{0x1.b2036406c80d9p-1, 0x1.526e5e3a1b438p-3},
{0x1.af286bca1af28p-1, 0x1.5ff3070a793d6p-3},
{0x1.ac5701ac5701bp-1, 0x1.6d60fe719d21bp-3},
{0x1.a98ef606a63bep-1, 0x1.7ab890210d907p-3},
{0x1.a6d01a6d01a6dp-1, 0x1.87fa06520c911p-3},
{0x1.a41a41a41a41ap-1, 0x1.9525a9cf456b6p-3},
{0x1.a16d3f97a4b02p-1, 0x1.a23bc1fe2b561p-3},
{0x1.9ec8e951033d9p-1, 0x1.af3c94e80bff3p-3},
{0x1.9c2d14ee4a102p-1, 0x1.bc286742d8cd4p-3},
{0x1.999999999999ap-1, 0x1.c8ff7c79a9a20p-3},
{0x1.970e4f80cb872p-1, 0x1.d5c216b4fbb94p-3},
{0x1.948b0fcd6e9e0p-1, 0x1.e27076e2af2e8p-3},
{0x1.920fb49d0e229p-1, 0x1.ef0adcbdc5935p-3},
{0x1.8f9c18f9c18fap-1, 0x1.fb9186d5e3e29p-3},
{0x1.8d3018d3018d3p-1, 0x1.0402594b4d041p-2},
{0x1.8acb90f6bf3aap-1, 0x1.0a324e27390e2p-2},
{0x1.886e5f0abb04ap-1, 0x1.1058bf9ae4ad4p-2},
{0x1.8618618618618p-1, 0x1.1675cababa60fp-2},
{0x1.83c977ab2beddp-1, 0x1.1c898c16999fbp-2},
{0x1.8181818181818p-1, 0x1.22941fbcf7966p-2},
{0x1.7f405fd017f40p-1, 0x1.2895a13de86a4p-2},
{0x1.7d05f417d05f4p-1, 0x1.2e8e2bae11d31p-2},
{0x1.7ad2208e0ecc3p-1, 0x1.347dd9a987d56p-2},
{0x1.78a4c8178a4c8p-1, 0x1.3a64c556945eap-2},
{0x1.767dce434a9b1p-1, 0x1.404308686a7e4p-2},
{0x1.745d1745d1746p-1, 0x1.4618bc21c5ec2p-2},
{0x1.724287f46debcp-1, 0x1.4be5f957778a1p-2},
{0x1.702e05c0b8170p-1, 0x1.51aad872df82ep-2},
{0x1.6e1f76b4337c7p-1, 0x1.5767717455a6cp-2},
{0x1.6c16c16c16c17p-1, 0x1.5d1bdbf5809cap-2},
{0x1.6a13cd1537290p-1, 0x1.62c82f2b9c796p-2},
{0x1.6816816816817p-1, 0x1.686c81e9b14adp-2},
{0x1.661ec6a5122f9p-1, 0x1.6e08eaa2ba1e4p-2},
{0x1.642c8590b2164p-1, 0x1.739d7f6bbd007p-2},
{0x1.623fa77016240p-1, 0x1.792a55fdd47a1p-2},
{0x1.6058160581606p-1, 0x1.7eaf83b82afc2p-2},
{0x1.5e75bb8d015e7p-1, 0x1.842d1da1e8b18p-2},
{0x1.5c9882b931057p-1, 0x1.89a3386c1425bp-2},
{0x1.5ac056b015ac0p-1, 0x1.8f11e873662c8p-2},
{0x1.58ed2308158edp-1, 0x1.947941c2116fbp-2},
{0x1.571ed3c506b3ap-1, 0x1.99d958117e08ap-2},
{0x1.5555555555555p-1, 0x1.9f323ecbf984dp-2},
{0x1.5390948f40febp-1, 0x1.a484090e5bb09p-2},
{0x1.51d07eae2f815p-1, 0x1.a9cec9a9a084ap-2},
{0x1.5015015015015p-1, 0x1.af1293247786bp-2},
{0x1.4e5e0a72f0539p-1, 0x1.b44f77bcc8f64p-2},
{0x1.4cab88725af6ep-1, 0x1.b9858969310fdp-2},
{0x1.4afd6a052bf5bp-1, 0x1.beb4d9da71b7ap-2},
{0x1.49539e3b2d067p-1, 0x1.c3dd7a7cdad4dp-2},
{0x1.47ae147ae147bp-1, 0x1.c8ff7c79a9a21p-2},
{0x1.460cbc7f5cf9ap-1, 0x1.ce1af0b85f3ecp-2},
{0x1.446f86562d9fbp-1, 0x1.d32fe7e00ebd5p-2},
{0x1.42d6625d51f87p-1, 0x1.d83e7258a2f3ep-2},
{0x1.4141414141414p-1, 0x1.dd46a04c1c4a1p-2},
{0x1.3fb013fb013fbp-1, 0x1.e24881a7c6c26p-2},
{0x1.3e22cbce4a902p-1, 0x1.e744261d68789p-2},
{0x1.3c995a47babe7p-1, 0x1.ec399d2468cc1p-2},
{0x1.3b13b13b13b14p-1, 0x1.f128f5faf06ecp-2},
{0x1.3991c2c187f63p-1, 0x1.f6123fa7028adp-2},
{0x1.3813813813814p-1, 0x1.faf588f78f31dp-2},
{0x1.3698df3de0748p-1, 0x1.ffd2e0857f497p-2},
{0x1.3521cfb2b78c1p-1, 0x1.02552a5a5d0ffp-1},
{0x1.33ae45b57bcb2p-1, 0x1.04bdf9da926d2p-1},
{0x1.323e34a2b10bfp-1, 0x1.0723e5c1cdf41p-1},
{0x1.30d190130d190p-1, 0x1.0986f4f573521p-1},
{0x1.2f684bda12f68p-1, 0x1.0be72e4252a83p-1},
{0x1.2e025c04b8097p-1, 0x1.0e44985d1cc8cp-1},
{0x1.2c9fb4d812ca0p-1, 0x1.109f39e2d4c96p-1},
{0x1.2b404ad012b40p-1, 0x1.12f719593efbdp-1},
{0x1.29e4129e4129ep-1, 0x1.154c3d2f4d5eap-1},
{0x1.288b01288b013p-1, 0x1.179eabbd899a0p-1},
{0x1.27350b8812735p-1, 0x1.19ee6b467c96fp-1},
{0x1.25e22708092f1p-1, 0x1.1c3b81f713c25p-1},
{0x1.2492492492492p-1, 0x1.1e85f5e7040d1p-1},
{0x1.23456789abcdfp-1, 0x1.20cdcd192ab6ep-1},
{0x1.21fb78121fb78p-1, 0x1.23130d7bebf43p-1},
{0x1.20b470c67c0d9p-1, 0x1.2555bce98f7cap-1},
{0x1.1f7047dc11f70p-1, 0x1.2795e1289b11bp-1},
{0x1.1e2ef3b3fb874p-1, 0x1.29d37fec2b08bp-1},
{0x1.1cf06ada2811dp-1, 0x1.2c0e9ed448e8cp-1},
{0x1.1bb4a4046ed29p-1, 0x1.2e47436e40268p-1},
{0x1.1a7b9611a7b96p-1, 0x1.307d7334f10bep-1},
{0x1.19453808ca29cp-1, 0x1.32b1339121d71p-1},
{0x1.1811811811812p-1, 0x1.34e289d9ce1d2p-1},
{0x1.16e0689427379p-1, 0x1.37117b54747b6p-1},
{0x1.15b1e5f75270dp-1, 0x1.393e0d3562a1ap-1},
{0x1.1485f0e0acd3bp-1, 0x1.3b68449fffc23p-1},
{0x1.135c81135c811p-1, 0x1.3d9026a7156fbp-1},
{0x1.12358e75d3033p-1, 0x1.3fb5b84d16f43p-1},
{0x1.1111111111111p-1, 0x1.41d8fe84672afp-1},
{0x1.0fef010fef011p-1, 0x1.43f9fe2f9ce67p-1},
{0x1.0ecf56be69c90p-1, 0x1.4618bc21c5ec2p-1},
{0x1.0db20a88f4696p-1, 0x1.48353d1ea88dfp-1},
{0x1.0c9714fbcda3bp-1, 0x1.4a4f85db03ebbp-1},
{0x1.0b7e6ec259dc8p-1, 0x1.4c679afccee39p-1},
{0x1.0a6810a6810a7p-1, 0x1.4e7d811b75bb0p-1},
{0x1.0953f39010954p-1, 0x1.50913cc01686bp-1},
{0x1.0842108421084p-1, 0x1.52a2d265bc5abp-1},
{0x1.073260a47f7c6p-1, 0x1.54b2467999498p-1},
{0x1.0624dd2f1a9fcp-1, 0x1.56bf9d5b3f399p-1},
{0x1.05197f7d73404p-1, 0x1.58cadb5cd7989p-1},
{0x1.0410410410410p-1, 0x1.5ad404c359f2dp-1},
{0x1.03091b51f5e1ap-1, 0x1.5cdb1dc6c1765p-1},
{0x1.0204081020408p-1, 0x1.5ee02a9241676p-1},
{0x1.0101010101010p-1, 0x1.60e32f44788d9p-1},
};
constexpr int kLog1pTabBytes = 128 * (int)sizeof(Log1pTabEntry);

// stage the table into LDS (every thread of the workgroup calls it; a barrier must follow)
__device__ __forceinline__ void stage_log1p_table(Log1pTabEntry* lds_tab) {
    for (int e = threadIdx.x; e < 128; e += blockDim.x) lds_tab[e] = kLog1pTab[e];
}

__device__ __forceinline__ double log1p_f64_fast(double x, const Log1pTabEntry* __restrict__ tab) {
    if (!(x >= 0.0) || !(x < INFINITY)) return log1p(x);          // negative / NaN / inf: the library routine
    const double u = 1.0 + x;
    const unsigned long long b = (unsigned long long)__double_as_longlong(u);
    const int e = (int)(b >> 52) - 1023;
    const int i = (int)((b >> 45) & 127ull);
    const double m = __longlong_as_double((long long)((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    const Log1pTabEntry t = tab[i];
    const double r = __builtin_fma(m, t.inv, -1.0);
    double p = __builtin_fma(r, -1.0 / 8, 1.0 / 7);
    p = __builtin_fma(r, p, -1.0 / 6);
    p = __builtin_fma(r, p, 1.0 / 5);
    p = __builtin_fma(r, p, -1.0 / 4);
    p = __builtin_fma(r, p, 1.0 / 3);
    p = __builtin_fma(r, p, -0.5);
    const double lr = __builtin_fma(r * r, p, r);
    const double c = (x - (u - 1.0)) * (double)__builtin_amdgcn_rcpf((float)u);
    const double de = (double)e;
    const double y = __builtin_fma(de, 0x1.62e42fee00000p-1, t.lg);
    return y + (lr + __builtin_fma(de, 0x1.a39ef35793c76p-33, c));
}

// The same with a degree-5 polynomial and without the (1 + x)-rounding term: <= 2e-13 relative for x >= 2^-10 (smaller arguments take the full routine; the
// dropped terms are r^6/6 < 4e-14 and the rounding of 1 + x, 1.1e-16 absolute), ~17 operations.  Used per value where
// the result is rounded to f32 or summed into per-gene moments that only have to rank the genes (f32 storage).
__device__ __forceinline__ double log1p_f64_lite(double x, const Log1pTabEntry* __restrict__ tab) {
    if (!(x >= 0x1p-10) || !(x < INFINITY)) return x == 0.0 ? 0.0 : log1p_f64_fast(x, tab);
    const double u = 1.0 + x;
    const unsigned long long b = (unsigned long long)__double_as_longlong(u);
    const int e = (int)(b >> 52) - 1023;
    const int i = (int)((b >> 45) & 127ull);
    const double m = __longlong_as_double((long long)((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    const Log1pTabEntry t = tab[i];
    const double r = __builtin_fma(m, t.inv, -1.0);
    double p = __builtin_fma(r, 1.0 / 5, -1.0 / 4);
    p = __builtin_fma(r, p, 1.0 / 3);
    p = __builtin_fma(r, p, -0.5);
    const double lr = __builtin_fma(r * r, p, r);
    return __builtin_fma((double)e, 0x1.62e42fefa39efp-1, t.lg) + lr;
}

// ---- the storage-precision ln_1p of log1p_transform_inplace (transform/mod.rs:43-47 for F32) ----------------------
template <typename T>
__device__ __forceinline__ T apply_log1p(T x);
// f32 ln(1+x) in ~10 instructions (ocml's log1pf is ~100 and made the fused pass VALU-bound):
// u = fl(1 + x); ln(1+x) = ln(u) * x / (u - 1) compensates the rounding of u (Goldberg / Kahan),
// with ln(u) = v_log_f32(u) * ln 2 (1 ulp hardware log2).  Measured <= 3e-7 relative against f64
// log1p over 1e-30 .. 1e30 (tests/test_stats_gpu.py::test_log1p_f32_accuracy).
template <>
__device__ __forceinline__ float apply_log1p<float>(float x) {
    const float u = 1.0f + x;
    const float d = u - 1.0f;
    const float lg = __builtin_amdgcn_logf(u) * 0.693147180559945309f;
    const float q = x >= 16777216.0f ? 1.0f : x * __builtin_amdgcn_rcpf(d);   // u == x there; rcp would flush
    return (d == 0.0f || !(u < INFINITY)) ? (d == 0.0f ? x : u) : lg * q;
}
template <>
__device__ __forceinline__ double apply_log1p<double>(double x) { return log1p(x); }

// out-of-line: the rare arguments of the routine below (keeps the hot loop's code and register footprint small)
static __device__ __attribute__((noinline)) double log1p_f64_rare(double x, const Log1pTabEntry* tab) {
    return x == 0.0 ? 0.0 : log1p_f64_fast(x, tab);
}

// ln_1p for the per-gene moments of the pipeline, written for the instruction count: the argument's class is read off the
// high word of x (one unsigned compare: 2^-10 <= x < inf), exponent / table index / mantissa come from the high word of
// u = 1 + x with 32-bit operations, degree-5 polynomial (same arithmetic as log1p_f64_lite: <= 2e-13 for those arguments);
// everything else (x < 2^-10, negative, NaN, inf) takes the full routine.
// the class test and the straight-line part on their own: a caller with several independent arguments evaluates the
// straight-line part for all of them (garbage, but no trap, for the rare classes) and patches the rare ones afterwards,
// which leaves the compiler free to interleave the chains
__device__ __forceinline__ bool log1p_f64_moment_is_rare(double x) {
    const unsigned xh = (unsigned)__double2hiint(x);
    return xh - 0x3f500000u >= 0x7ff00000u - 0x3f500000u;
}
__device__ __forceinline__ double log1p_f64_moment_common(double x, const Log1pTabEntry* __restrict__ tab) {
    const double u = 1.0 + x;
    const int uh = __double2hiint(u);
    const int e = (uh >> 20) - 1023;
    const int i = (uh >> 13) & 127;
    const double m = __hiloint2double((uh & 0x000fffff) | 0x3ff00000, __double2loint(u));
    const Log1pTabEntry t = tab[i];
    const double r = __builtin_fma(m, t.inv, -1.0);
    double p = __builtin_fma(r, 1.0 / 5, -1.0 / 4);
    p = __builtin_fma(r, p, 1.0 / 3);
    p = __builtin_fma(r, p, -0.5);
    const double lr = __builtin_fma(r * r, p, r);
    return __builtin_fma((double)e, 0x1.62e42fefa39efp-1, t.lg) + lr;
}
__device__ __forceinline__ double log1p_f64_moment(double x, const Log1pTabEntry* __restrict__ tab) {
    const unsigned xh = (unsigned)__double2hiint(x);
    if (__builtin_expect(xh - 0x3f500000u >= 0x7ff00000u - 0x3f500000u, 0)) return log1p_f64_rare(x, tab);
    const double u = 1.0 + x;
    const int uh = __double2hiint(u);
    const int e = (uh >> 20) - 1023;
    const int i = (uh >> 13) & 127;
    const double m = __hiloint2double((uh & 0x000fffff) | 0x3ff00000, __double2loint(u));
    const Log1pTabEntry t = tab[i];
    const double r = __builtin_fma(m, t.inv, -1.0);
    double p = __builtin_fma(r, 1.0 / 5, -1.0 / 4);
    p = __builtin_fma(r, p, 1.0 / 3);
    p = __builtin_fma(r, p, -0.5);
    const double lr = __builtin_fma(r * r, p, r);
    return __builtin_fma((double)e, 0x1.62e42fefa39efp-1, t.lg) + lr;
}

// The transform of one value of a row whose scale is `scale`: y = ln_1p(f64(v) * scale).  Count data takes few distinct
// values per cell, so every wave that works on a row first builds the row's table y(c) = ln_1p(c * scale) for c = 0 .. 63
// (ONE evaluation per lane: `xf_row_table`), and a value that is a small non-negative integer fetches its result from
// lane c with a shuffle; anything else is evaluated directly.  Both routes go through the same function: the result does
// not depend on which one ran.  MUST be called with every lane of the wave active (the shuffle reads lane c).
__device__ __forceinline__ double xf_row_table(double scale, const Log1pTabEntry* __restrict__ tab) {
    return log1p_f64_fast((double)(threadIdx.x & 63) * scale, tab);
}
template <typename T>
__device__ __forceinline__ double xf_apply(T v, double scale, double row_table, const Log1pTabEntry* __restrict__ tab) {
    const int c = (int)v;
    const bool ok = (T)c == v && (unsigned)c < 64u;
    double y = __shfl(row_table, ok ? c : 0, 64);
    if (!ok) y = log1p_f64_fast((double)v * scale, tab);
    return y;
}

// What the pipeline leaves in X for a raw value v of a row with this scale — so that a pass which reads the raw matrix (the
// compaction of a backed tile) produces exactly the stored values: the f64 logarithm the moments pass forms (degree-5 variant
// for f32 storage, rounded once to f32: correctly rounded but for ~3 values in a million; the full-accuracy one for f64).
// (The three separate calls on an f32 matrix round v * scale to f32 first and take the f32 logarithm: k_row_pass; <= 3e-7 apart.)
__device__ __forceinline__ float xf_stored(float v, double scale, const Log1pTabEntry* __restrict__ tab) {
    return (float)log1p_f64_moment((double)v * scale, tab);
}
__device__ __forceinline__ double xf_stored(double v, double scale, const Log1pTabEntry* __restrict__ tab) {
    return log1p_f64_fast(v * scale, tab);
}

}  // namespace srx
