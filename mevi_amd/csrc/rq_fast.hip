// Residual-quantisation encode on the matrix cores with an exact re-check (round 3).
//
// Replaces pq.get_rq_document_cluster / forward_rq with dist_mode 'l2' (MEVI/pq.py:124-131, 281-305, 337-369) -- the same
// codes, bit for bit, as rq_encode.hip / oracle/mevi_oracle.c (sequential f32 fmaf chains of (r - c)^2, ties to the lowest
// centroid), at the HBM rate instead of the f32 VALU rate.
//
// Idea.  ||r_j - c||^2 = ||r_j||^2 - 2 r_j.c + ||c||^2 and r_j = x - sum_{i<j} C_i[code_i], so
//     r_j.c = x.c - sum_{i<j} C_i[code_i].C_j[c]:
// ONE product x.C^T against ALL M*K centroids (f16 MFMA, f32 accumulate) plus table look-ups
// G[i][a][j][c] = C_i[a].C_j[c] gives an approximation F_j(c) of  dist_j(c) - ||r_j||^2  for every level; the residual is
// never formed and x crosses HBM once (K = 256: once per level -- the accumulators of 768 columns do not fit a wave).
// Only x and C are rounded to f16; x is centred first (x' = x - mu, mu = mean of the level-0 centroids; level-0 centroids
// are centred the same way, which changes no distance) so the rounding error scales with ||x - mu|| ||c - mu||, not with
// the common component of dense-retriever embeddings.
//
// Exactness.  Per row and level, with m = min_c F(c) at c*:  every c with F(c) - m > Delta provably has a LARGER chain
// distance than c* (Delta: rigorous bound, see rf_delta), so the oracle's argmin is among the candidates {F(c) <= m + Delta}.
//   * one candidate  -> code = c* (no exact distance needed);
//   * several        -> the kernel speculates c*, appends a RECORD (row, level, premise codes, candidates); the fix-up
//                       kernel evaluates the candidates' exact chains (the reference arithmetic: residual in f32 in the
//                       reference's operation order, d = fma(r - c, r - c, d) over k) and takes (distance, index) min;
//   * speculation wrong, > 8 candidates, f16 overflow, record buffer full -> the row is flagged and re-encoded from scratch
//     by the exact VALU kernel (rq_encode.hip) through an index list.
// No host synchronisation: record / flag counts stay on the device.
//
// Kernel.  Persistent 512-thread workgroups, tile = 256 x rows x TA*32 centroid columns (TA = 4 | 8 MFMA tiles: all levels of
// (4, 32) in one tile, one level of K = 256 per tile).  Wave w owns x rows 32w..32w+31 against ALL columns: acc[TA] of
// v_mfma_f32_32x32x16_f16 with A = centroids, B = x, so a lane holds 16 TA/2.. distances of ONE row and the per-level argmin
// is in-lane + one cross-half shuffle.  Operands are staged by LDS-DMA (buffer_load ... lds) in units of 32 k: the f16
// centroid image (unit-major, L2 resident) and the f32 x rows straight from the caller's matrix -- converted to f16 between
// the LDS read and the MFMA (fma with the scale and -mu, v_cvt_pk), so there is no f16 copy of the corpus.  Three unit
// buffers (48 KiB each at TA = 8), two units = 64 KiB of x in flight per CU.
// Bound: HBM for (4, 32) (MFMA ~25 % busy); MFMA / LDS for K = 256.

#include "mfma_pp_f16.h"

#include <float.h>
#include <type_traits>
#include <math.h>
#include <vector>

namespace mevi {
namespace {

constexpr int RF_ROWS = 256;  // x rows per tile
constexpr int RF_NBUF = 3;
constexpr int RF_MAXC = 8;    // candidates a record can hold

struct __attribute__((aligned(16))) RfRecord {
  long long row;
  unsigned long long prev;  // premise: codes of levels < level, one byte each
  unsigned char level, ncand, spec, pad;
  unsigned char cand[RF_MAXC];
  unsigned int pad2;
};
static_assert(sizeof(RfRecord) == 32, "record layout");

// per-level constants written by the prep kernels (device memory; the main kernel reads them with scalar loads)
struct RfLevel {
  float inv2;    // -2 / (S_x S_c,j)
  float cnmax;   // max_c ||c'_j||  (rounded up)
  float qpre;    // sum_{i<j} cnmax_i
  float und_c;   // 2 * 2^-25 sqrt(dim) / S_c,j   (x ||x'||: products lost to f16 underflow of the centroids)
  float und_x;   // 2 * 2^-25 sqrt(dim) cnmax_j / S_x
  float sc;      // S_c,j
  float dc2;     // 2 (1 + 2u) max_c ||c'_j - image row / S_c,j||: what the centroids' ONE rounding to f16 really moved them (x ||x'||);
                 // written by rf_image_kernel (atomic max of the bits); 0 with the split image, whose term stays in e16
  float pad;
};

struct RfParams {
  const float *X;
  long long n;
  int dim;
  const _Float16 *img;  // [ngroups][U][TA*32][32] halves: unit-major centroid image, centred (level 0) and scaled
  const float *mus;     // [dim]  mu * S_x
  const float *A;       // [M][Kp]  ||c'||^2 (+inf for padding columns)
  const float *G2;      // [(i*K + a)][M][Kp]  2 * c'_i[a] . c'_j[c]  (i < j)
  const RfLevel *lev;   // [M]
  const float *scal;    // [0] S_x  [1] 1 / S_x^2
  int M, K, Kp, LPG, ngroups;
  int *codes;
  unsigned char *row_flag;
  RfRecord *rec;           // [n_regions][region_cap]: one region per wave of the persistent grid (no atomics on the hot path)
  unsigned int *counters;  // [0] bad rows (bad-list kernel)  [1] rows flagged by the main kernel  [2 ...] unused
  unsigned int *region_n;  // [n_regions] records wanted by the region's wave (may exceed region_cap: the excess rows are flagged)
  unsigned int region_cap;
  float e16, gam;
  int x_row_stride, x_unit_stride;  // bytes between rows / between 32-k units of a row (dim * 4, 128; an experiment sets others)
  long long n_tiles;
  char *xs;   // XDIR: per-workgroup scratch, [grid][dim / 16 k-steps][512 lanes] x 16 B: the f16 x fragments of the row tile in hand
};

// SPLIT: both operands as (hi, lo) f16 pairs, three MFMAs per product (hi.hi + hi.lo + lo.hi; lo scaled by 2^11 so that it
// stays in the f16 normal range) -- the centroid image then holds TA*32 hi rows followed by TA*32 lo rows per unit
constexpr int RF_XFL = RF_ROWS * 32;  // floats of one x unit (256 rows x 128 B)
template <int TA, bool SPLIT>
constexpr int rf_a_floats() { return TA * 32 * 16 * (SPLIT ? 2 : 1); }  // floats of one centroid unit (64 B per image row)
// Rings: three x buffers (two units of x in flight; a fourth, where it fits, measured no faster: the loop alone streams at
// 5.2 TB/s, the per-tile epilogue is what costs -- DESIGN 4.3b), three centroid buffers (two with the split image, which
// leaves room for the tables below; the centroid unit comes from L2 and lands within a cycle).
// TA = 4 shapes (<= 4 levels x 32 centroids) keep the epilogue's tables in LDS: A [M][32] and the i < j blocks of G
// [pair][32][32] (24.5 KiB at M = 4) -- the level chain is four DEPENDENT look-ups per tile, each an L2 round trip otherwise.
constexpr int RF_XB = 3;
template <int TA, bool SPLIT>
constexpr int rf_abuf() { return SPLIT ? 2 : 3; }
template <int TA>
constexpr int rf_table_floats() { return TA == 4 ? 4 * 32 + 6 * 32 * 32 : 0; }
template <int TA, bool SPLIT>
constexpr size_t rf_lds_bytes(int dim) {
  return (size_t)(rf_abuf<TA, SPLIT>() * rf_a_floats<TA, SPLIT>() + RF_XB * RF_XFL + rf_table_floats<TA>()) * 4 + (size_t)dim * 4;
}

// Delta of the header: candidates are the c with F(c) - m <= Delta.
//   F*(c)  = D*(c) - rho  (real arithmetic; D* the distance to the real-number residual, rho = ||r*_j||^2),  |F - F*| <= E1
//   chain  = the oracle's f32 value:  |chain(c) - ||r_f32 - c||^2| <= gam * itself,   ||r_f32 - r*|| <= dl
// c is excluded when D*(c) > D*(c*) (1 + 2.1 gam) + 2.1 e2, with e2 = 2 sqrt(Dh) dl + dl^2 covering the residual's own
// rounding for distances up to Dh.  Returns Delta (+inf when the premises of the bound do not hold -> everything is a candidate).
__device__ __forceinline__ float rf_delta(float m, float rho_hat, float E1, float gam, float dl, float &dpos_out) {
  const float dpos = fmaxf(rho_hat + m + E1, 0.f);  // upper bound of D*(c*)
  const float dh = 2.f * dpos + 8.f * E1 + 1e-30f;
  const float e2 = (2.f * sqrtf(dh) * dl + dl * dl) * 1.01f;
  dpos_out = dpos;
  if (!(2.1f * e2 <= 6.f * E1 + dpos)) return INFINITY;
  return (2.f * E1 + 2.1f * gam * dpos + 2.1f * e2) * 1.01f;
}

// XSPLIT (the TA = 8 shapes, where a second accumulator set does not fit): x alone as a (hi, lo) pair, the un-scaled lo part
// accumulated into the SAME accumulator by a second MFMA -- the bound loses x's rounding (e16 x0.58), K = 256 stays bound by
// its three passes over x.
// XDIR (several groups per row tile, i.e. K = 128 / 256: one level per pass): x crosses HBM, the LDS and the f32 -> f16 conversion
// ONCE per row tile.  The pass of group 0 stores every lane's converted fragment (16 B per k-step) to the workgroup's scratch
// (393 KB at dim 768: it stays in L2 / the Infinity Cache); the passes of the later groups load the fragments straight back into
// registers -- the lane that wrote a fragment is the lane that reads it -- two units ahead, in place of the x LDS-DMA (same
// look-ahead, same counted waits: stores, loads and LDS-DMA retire in issue order).  Needs dim / 32 divisible by 3 (a ring of
// three units in registers, indexed at compile time).
//
// XDEEP (round 5; XDIR shapes with dim / 32 divisible by 12): the passes of groups >= 1 run a DEEPER look-ahead.  PMC of the XDIR
// kernel (profiles/r04_rq_xdir.txt) showed its waves parked 63 % of their life at the per-unit wait: a unit's centroid slab and x
// fragments were requested about one cycle (~0.6 us of MFMA) before their use, an L2 round trip of 1-2 us.  In those passes the
// f32 x ring (96 of the 160 KiB) is dead, so its space is re-cut into 16-KiB slots (slot s = floats [s AFL, (s + 1) AFL)):
//   pass 0           slots 0 1 2 = centroid ring (unit u in slot u mod 3)      3+4 | 5+6 | 7+8 = x ring X0 X1 X2 (f32, 32 KiB each)
//   passes >= 1      deep unit v = (g - 1) U + u:  centroid slab in slot {0, 1, 2, 8}[v mod 4],
//                    its f16 x fragments (LDS-DMA from the scratch: [k-step][thread] x 16 B) in slot {3, 4, 5, 7}[v mod 4]
// Deep unit v + 3 is requested in cycle v (three cycles ahead, both operands through LDS: no register ring), and the counted
// wait lets the requests of the last TWO windows fly.  The hand-overs reuse the XDIR kernel's request times: pass 0's last two
// cycles request deep units 0 and 1 (slots 0 / 3 and 1 / 4 are free by then), deep cycle 0 requests units 2 AND 3, the last two
// deep cycles request the next row tile's centroid units 0, 1 (slots 0, 1) and x units 0, 1 (X0, X1) -- every target slot's
// previous tenant has been read by then (checked case by case in DESIGN 4.3b).  Same MFMA sequence per row: same codes.
template <int TA, int KT, bool SPLIT, bool XSPLIT, bool XDIR = false, bool XDEEP = false>
__device__ __forceinline__ void rq_fast_body(const RfParams &p, float *lds) {
  static_assert(!(SPLIT && XSPLIT), "one or the other");
  static_assert(!XDIR || (!SPLIT && !XSPLIT), "XDIR: plain f16 fragments");
  static_assert(!XDEEP || (XDIR && TA == 8), "XDEEP: a refinement of the XDIR kernels (16-KiB centroid units)");
  constexpr int NB = (SPLIT || XSPLIT) ? 2 : 1;   // B fragments per k-step: hi (, lo)
  constexpr int AROWS = TA * 32 * (SPLIT ? 2 : 1);   // image rows per unit (hi rows, then lo rows)
  constexpr int AFL = rf_a_floats<TA, SPLIT>();      // floats of a centroid unit
  constexpr int XB = RF_XB;                          // x ring: XB buffers, XB - 1 units in flight
  constexpr int DX = XB - 1;
  constexpr int AB = rf_abuf<TA, SPLIT>();           // centroid ring: AB buffers, AB - 1 units ahead
  constexpr int DA = AB - 1;
  constexpr int XBASE = AB * AFL;                    // floats: the x ring starts behind the centroid ring
  constexpr int TBASE = XBASE + XB * RF_XFL;         // epilogue tables (TA = 4), then mu
  constexpr bool TLDS = TA == 4;
  constexpr int PA_PER_WAVE = AROWS / 128;           // A pieces (16 rows x 64 B) per wave and unit
  static_assert((TBASE + rf_table_floats<TA>()) * 4 + 4096 <= 160 * 1024, "LDS budget");
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lrow = lane & 31, half = lane >> 5;
  const int U = p.dim >> 5;
  float *mus_l = lds + TBASE + rf_table_floats<TA>();
  for (int i = t; i < p.dim; i += 512) mus_l[i] = p.mus[i];
  float *tabA = lds + TBASE, *tabG = lds + TBASE + 4 * 32;   // TLDS: A [M][32]; G pair (i < j) at j (j - 1) / 2 + i: [K][32]
  if constexpr (TLDS) {
    for (int i = t; i < p.M * 32; i += 512) tabA[i] = p.A[i];              // Kp = 32 on this shape
    for (int j = 1; j < p.M; ++j)
      for (int i = 0; i < j; ++i)
        for (int e = t; e < p.K * 32; e += 512)
          tabG[(j * (j - 1) / 2 + i) * 1024 + e] = p.G2[((size_t)(i * p.K + (e >> 5)) * p.M + j) * 32 + (e & 31)];
  }
  __syncthreads();

  // ---- DMA duties ---------------------------------------------------------------------------------------------------
  // x: piece q (8 rows x 128 B) of this wave's own 32 rows; lane (r8 = lane >> 3, slot = lane & 7) fetches logical piece
  // slot ^ g(row), g(row) = (row >> 1) & 7, so that the b128 fragment reads below are bank-conflict free
  // (two lane patterns serve the four pieces -- the swizzle key of piece q is (4 q + (r8 >> 1)) & 7: even | odd q -- and one serves
  // every centroid piece; the rest of a piece's address is wave-uniform and rides in the instruction's scalar offset: per-lane
  // invariants are what pass 0 has no registers for)
  int voff_x[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r8 = lane >> 3, row = 32 * w8 + 8 * q + r8;
    voff_x[q] = row * p.x_row_stride + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  // centroid image: piece (16 rows x 64 B); lane (r16 = lane >> 2, slot = lane & 3) fetches piece slot ^ ((row >> 2) & 3)
  int voff_a0;
  {
    const int row = 16 * (w8 * PA_PER_WAVE) + (lane >> 2);
    voff_a0 = row * 64 + (((lane & 3) ^ ((row >> 2) & 3)) << 4);
  }
  const unsigned int a_block_bytes = (unsigned int)(AROWS * 64);  // one unit of one group's image

  // work list of this workgroup: row tiles blockIdx.x, + gridDim.x, ...; inside a row tile the groups in order
  long long rt = blockIdx.x;
  int grp = 0;
  struct Src {
    const void *x;
    unsigned int xbytes;
    const void *a;
  };
  auto src_of = [&](long long rt_, int g_, Src &s) {
    const long long r0 = rt_ * RF_ROWS;
    long long rows = p.n - r0;
    rows = rows > RF_ROWS ? RF_ROWS : (rows < 0 ? 0 : rows);
    s.x = p.X + (size_t)r0 * p.dim;
    s.xbytes = (unsigned int)(rows * p.dim * 4);
    s.a = reinterpret_cast<const char *>(p.img) + (size_t)g_ * U * a_block_bytes;
  };
  auto advance = [&](long long &rt_, int &g_) -> bool {  // next (row tile, group) of this workgroup
    if (++g_ < p.ngroups) return true;
    g_ = 0;
    rt_ += gridDim.x;
    return rt_ < p.n_tiles;
  };
  if (rt >= p.n_tiles) {
    if (lane == 0) p.region_n[blockIdx.x * 8 + w8] = 0u;
    return;
  }
  Src cur, nxt;
  src_of(rt, grp, cur);
  long long rt_n = rt;
  int grp_n = grp;
  bool have_nxt = advance(rt_n, grp_n);
  if (have_nxt) src_of(rt_n, grp_n, nxt);
  else nxt = cur, nxt.xbytes = 0u;

  // the wave's pieces of centroid unit u -> centroid buffer ab (waves 0-7 share the unit), of x unit u -> x buffer xb (its own rows)
  auto dma_a = [&](const Src &s, bool live, int u, int ab) {
    const __amdgpu_buffer_rsrc_t ra =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(s.a), 0, live ? (int)(U * a_block_bytes) : 0, 0x00020000);
#pragma unroll
    for (int ia = 0; ia < PA_PER_WAVE; ++ia)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void *)(lds + ab * AFL + (16 * (w8 * PA_PER_WAVE + ia)) * 16),
                                               16, voff_a0, u * (int)a_block_bytes + ia * 1024, 0, 0);
  };
  auto dma_x = [&](const Src &s, int u, int xb, int first, int count) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(s.x), 0, (int)s.xbytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < first || i >= first + count) continue;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void *)(lds + XBASE + xb * RF_XFL + (32 * w8 + 8 * i) * 32), 16,
                                               voff_x[i & 1], u * p.x_unit_stride + (i >> 1) * 16 * p.x_row_stride, 0, 0);
    }
  };

  // ---- fragment addresses ---------------------------------------------------------------------------------------------
  const int a_sw = (lrow >> 2) & 3;
  const int offa = lrow * 16;                                  // floats; + 32 t * 16 per MFMA tile
  const int offb = XBASE + (32 * w8 + lrow) * 32;              // floats
  const int b_sw = (lrow >> 1) & 7;
  f32x16 acc[TA];
  // Fragments.  A k-step (16 k) is issued as two HALF-STEPS of H = TA/2 MFMAs (centroid tiles [0, H) then [H, TA)); the
  // A fragments of a half-step are read while the previous half-step's MFMAs run, so 2 x H fragments are live instead of the
  // 2 x TA of a whole-k-step double buffer (at TA = 8 that version spilled 124 registers, some inside the loop).
  constexpr int H = TA / 2;
  constexpr int HS = SPLIT ? 2 * H : H;   // fragments of a half-step: [0, H) hi parts, [H, 2H) lo parts
  f16x8 alo[HS], ahi[HS], bq0[NB], bq1[NB];
  f32x16 accx[SPLIT ? TA : 1];            // SPLIT: the two cross terms (x 2^11)
  const float sx = p.scal[0];
  float rho_s = 0.f;  // sum of (x' S_x)^2 over this lane's k (the other half holds the rest)
  auto read_a = [&](int gb, int j, int h0, f16x8 (&dst)[HS]) {
    const float *ub = lds + gb * AFL + offa + (((2 * j + half) ^ a_sw) << 2);
#pragma unroll
    for (int ti = 0; ti < H; ++ti) {
      dst[ti] = *reinterpret_cast<const f16x8 *>(ub + 32 * (h0 + ti) * 16);
      if constexpr (SPLIT) dst[H + ti] = *reinterpret_cast<const f16x8 *>(ub + (TA * 32 + 32 * (h0 + ti)) * 16);
    }
  };
  // this lane's 8 x values of k-step (u, j): centred, scaled, rounded to f16 (and added to the row norm)
  auto read_x = [&](int gb, int u, int j, f16x8 (&b)[NB]) {
    const float *ub = lds + gb * RF_XFL;
    const int c0 = 4 * j + 2 * half;
    int h8 = 8 * half;
    if constexpr (XDEEP) asm volatile("" : "+v"(h8));   // (not an invariant to keep: the unrolled cycles each held their own copy of this address)
    // The four LDS reads of a conversion as inline assembly (round 5, found in the ISA).  Written as C++ loads they made the
    // compiler put `s_waitcnt vmcnt(0)` in front of them in every cycle of pass 0: it sees LDS-DMA requests in flight (the counted
    // waits of this kernel are inline assembly, invisible to its bookkeeping) and cannot tell their targets from the x rows and mu
    // read here, so the whole look-ahead -- the unit requested a moment ago included -- was drained twice per cycle.  The data
    // read here landed before the barrier of the previous cycle (wait_window / the counted vmcnt).  (Measured: nothing at K = 256,
    // whose cycle is bound elsewhere -- profiles/r05_rq_ablation.txt --, -2 % at (4, 32).  Reads kept in flight under the next MFMA
    // block -- a second statement for the wait -- need 16 registers pass 0 does not have at TA = 8: 8 -> 700 B of scratch.)
    float4 x0, x1, m0, m1;
    {
      typedef __attribute__((address_space(3))) const float *lp_t;
      const unsigned int ax0 = (unsigned int)(unsigned long long)(lp_t)(ub + offb + ((c0 ^ b_sw) << 2));
      const unsigned int ax1 = (unsigned int)(unsigned long long)(lp_t)(ub + offb + (((c0 + 1) ^ b_sw) << 2));
      const unsigned int am0 = (unsigned int)(unsigned long long)(lp_t)(mus_l + 32 * u + 16 * j + h8);
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %6 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(x0), "=&v"(x1), "=&v"(m0), "=&v"(m1)
                   : "v"(ax0), "v"(ax1), "v"(am0)
                   : "memory");
    }
    float v[8];
    v[0] = fmaf(x0.x, sx, -m0.x); v[1] = fmaf(x0.y, sx, -m0.y); v[2] = fmaf(x0.z, sx, -m0.z); v[3] = fmaf(x0.w, sx, -m0.w);
    v[4] = fmaf(x1.x, sx, -m1.x); v[5] = fmaf(x1.y, sx, -m1.y); v[6] = fmaf(x1.z, sx, -m1.z); v[7] = fmaf(x1.w, sx, -m1.w);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      rho_s = fmaf(v[i], v[i], rho_s);
      const _Float16 h = (_Float16)v[i];
      b[0][i] = h;
      if constexpr (SPLIT) b[1][i] = (_Float16)((v[i] - (float)h) * 2048.f);
      if constexpr (XSPLIT) b[1][i] = (_Float16)(v[i] - (float)h);
    }
  };
  auto mma_lo = [&](const f16x8 (&b)[NB]) {
#pragma unroll
    for (int ti = 0; ti < H; ++ti) {
      acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ti], b[0], acc[ti], 0, 0, 0);
      if constexpr (XSPLIT) acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ti], b[1], acc[ti], 0, 0, 0);
      if constexpr (SPLIT) {
        accx[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ti], b[1], accx[ti], 0, 0, 0);
        accx[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[H + ti], b[0], accx[ti], 0, 0, 0);
      }
    }
  };
  auto mma_hi = [&](const f16x8 (&b)[NB]) {
#pragma unroll
    for (int ti = 0; ti < H; ++ti) {
      acc[H + ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ti], b[0], acc[H + ti], 0, 0, 0);
      if constexpr (XSPLIT) acc[H + ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ti], b[1], acc[H + ti], 0, 0, 0);
      if constexpr (SPLIT) {
        accx[H + ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ti], b[1], accx[H + ti], 0, 0, 0);
        accx[H + ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[H + ti], b[0], accx[H + ti], 0, 0, 0);
      }
    }
  };

  int ra_ = 0, rx_ = 0;  // ring slots of the unit being computed (centroid ring of 3, x ring of XB)
  // ---- XDIR: the fragment ring in registers, the scratch, the window count -------------------------------------------------------
  f16x8 xr[XDIR && !XDEEP ? 3 : 1][2];   // units u, u + 1, u + 2 of the stream, k-steps 0 | 1; slot = unit mod 3 (XDEEP: through LDS)
  char *xs_t = XDIR ? p.xs + ((size_t)blockIdx.x * (size_t)(U * 2) * 512 + (size_t)t) * 16 : nullptr;   // + k-step * 8192
  int win = 0;   // vector-memory operations issued since the last counted wait (wave-uniform)
  // everything issued BEFORE the window has landed; the window's own operations may stay in flight (they retire in issue order)
  int win_prev = 0;   // XDEEP: operations of the previous window (they too may stay in flight in the deep cycles)
  auto wait_count = [&](int keep) {
    switch (keep) {
#define MEVI_RF_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      MEVI_RF_W(1) MEVI_RF_W(2) MEVI_RF_W(3) MEVI_RF_W(4) MEVI_RF_W(5) MEVI_RF_W(6) MEVI_RF_W(7) MEVI_RF_W(8) MEVI_RF_W(9)
      MEVI_RF_W(10) MEVI_RF_W(11) MEVI_RF_W(12) MEVI_RF_W(13) MEVI_RF_W(14) MEVI_RF_W(15) MEVI_RF_W(16)
#undef MEVI_RF_W
      default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
  };
  auto wait_window = [&]() {
    wait_count(win);
    win_prev = win;
    win = 0;
  };
  // XDEEP: k-step `part` of x16 unit uq (this wave's 64 fragments = 1 KiB of the scratch, [k-step][thread] x 16 B) -> LDS slot
  // (the lane's thread index goes through an opaque copy at every use: left visible, the compiler hoists these addresses out of
  // the tile loop, and pass 0 -- 128 accumulators + the conversion -- then spills into its cycles: scratch loads there are
  // vector-memory operations whose `s_waitcnt vmcnt(0)` drains the whole DMA look-ahead; measured 26 -> 47 ms per encode)
  auto dma_x16 = [&](int uq, int part, int slot) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        p.xs + (size_t)blockIdx.x * (size_t)(U * 2) * 8192, 0, (int)(U * 2 * 8192), 0x00020000);
    int tt = threadIdx.x;
    asm volatile("" : "+v"(tt));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + slot * AFL + (part * 512 + 64 * w8) * 4), 16,
                                             tt * 16, (2 * uq + part) * 8192, 0, 0);
  };
  auto read_x16 = [&](int slot, int j, f16x8 (&b)[NB]) {
    int tt = threadIdx.x;
    asm volatile("" : "+v"(tt));
    b[0] = *reinterpret_cast<const f16x8 *>(lds + slot * AFL + j * 2048 + tt * 4);
  };
  int grp_now = 0;   // group of the tile being computed (set with `grp` below; read by the lambdas)
  // k-step (u, j) of the tile in hand: group 0 converts it from the x ring (and, XDIR, stores the fragment); later groups (XDIR)
  // take it from the register ring
  auto get_x = [&](int u, int j, auto slot_c, f16x8 (&b)[NB]) {
    if constexpr (XDIR) {
      if (grp_now != 0) {
        if constexpr (XDEEP) read_x16(3, j, b);          // only the top-of-pass read of deep unit (g - 1) U: slot {3,4,5,7}[0]
        else b[0] = xr[decltype(slot_c)::value][j];
        return;
      }
      read_x(rx_, u, j, b);
      if constexpr (XDEEP) {   // a buffer store (scalar base + 32-bit lane offset): the 64-bit store addresses of the unrolled cycles
                               // were loop invariants the compiler kept -- and spilled into pass 0's cycles (see dma_x16)
        typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            p.xs + (size_t)blockIdx.x * (size_t)(U * 2) * 8192, 0, (int)(U * 2 * 8192), 0x00020000);
        int tt = threadIdx.x;
        asm volatile("" : "+v"(tt));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, b[0]), rs, tt * 16, (2 * u + j) * 8192, 0);
      } else {
        *reinterpret_cast<f16x8 *>(xs_t + (size_t)(2 * u + j) * 8192) = b[0];
      }
      ++win;
    } else {
      read_x(rx_, u, j, b);
    }
  };
  // half `part` of the x request of a cycle: two LDS-DMA pieces of unit tux of stream tx, or (XDIR, target group > 0) k-step
  // `part` of that unit straight into the register ring
  auto issue_x = [&](const Src &tx, int tgrp, int tux, int wbx, int part, auto slot_c) {
    if constexpr (XDIR) {
      if (tgrp != 0) {
        if constexpr (XDEEP) {     // the first deep units (0, 1), requested by pass 0's last two cycles: slots 3, 4
          dma_x16(tux, part, 3 + tux);
        } else {
          const char *src = xs_t + (size_t)(2 * tux + part) * 8192;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xr[decltype(slot_c)::value][part]) : "v"(src) : "memory");
        }
        ++win;
        return;
      }
      win += 2;
    }
    dma_x(tx, tux, wbx, 2 * part, 2);
  };

  RfRecord *rec_base = p.rec + (size_t)(blockIdx.x * 8 + w8) * p.region_cap;
  unsigned int rec_n = 0u;  // records of this wave so far (wave-uniform)
  // per-row state carried across the groups of a row tile (this lane's row = 32 w8 + lrow of the tile; both halves agree)
  unsigned long long prev = 0ull;
  float rho_hat = 0.f, xn = 0.f;
  bool row_bad = false;

  // One unit (32 k) of the current tile.  On entry alo / bq0 hold k-step (u, 0).  Centroid unit u+2 and x unit u+DX of the
  // stream are requested at the start (their buffers were released by the previous cycle's barrier) -- the centroid pieces
  // FIRST: vmcnt retires in order, and at the barrier everything issued after centroid unit u+1 may stay in flight.  The
  // barrier sits before the LAST half-step: by then every read of unit u has been issued (and, with lgkmcnt(0), completed)
  // and unit u+1 must have landed for the reads of k-step (u+1, 0) that the last half-step issues.
  auto cycle = [&](int u, auto slot_c, bool last) {
    constexpr int S0 = decltype(slot_c)::value;          // XDIR: ring slot of unit u (= u mod 3); u + 1, u + 2 follow
    using S1 = std::integral_constant<int, (S0 + 1) % 3>;
    using S2 = std::integral_constant<int, (S0 + 2) % 3>;
    {
      const bool sp = u + DA >= U;
      dma_a(sp ? nxt : cur, sp ? have_nxt : true, sp ? u + DA - U : u + DA, ra_ == 0 ? AB - 1 : ra_ - 1);
      if constexpr (XDIR) win += PA_PER_WAVE;
    }
    const bool spx = u + DX >= U;
    const Src &tx = spx ? nxt : cur;
    const int tux = spx ? u + DX - U : u + DX;
    const int tgrp = spx ? (have_nxt ? grp_n : 0) : grp_now;     // group of the pass unit tux belongs to
    const int wbx = rx_ == 0 ? XB - 1 : rx_ - 1;
    read_a(ra_, 0, H, ahi);
    mma_lo(bq0);
    issue_x(tx, tgrp, tux, wbx, 0, S2());
    __builtin_amdgcn_sched_barrier(0);
    read_a(ra_, 1, 0, alo);
    get_x(u, 1, slot_c, bq1);
    mma_hi(bq0);
    issue_x(tx, tgrp, tux, wbx, 1, S2());
    __builtin_amdgcn_sched_barrier(0);
    read_a(ra_, 1, H, ahi);
    mma_lo(bq1);
    __builtin_amdgcn_sched_barrier(0);
    // may stay outstanding: what was issued after centroid unit u+1 -- with DA = 2 that unit was requested a cycle ago, ahead of
    // x unit u+2: this cycle's centroid unit u+2 and x unit u+2 stay; with DA = 1 it was requested THIS cycle: only x unit u+2
    constexpr int KEEP = DA == 2 ? PA_PER_WAVE + 4 : 4;
    static_assert(DX == 2 && (KEEP == 4 || KEEP == 5 || KEEP == 6), "vmcnt immediates below");
    static_assert(!XDIR || DA == 2, "XDIR counts its windows for the two-ahead centroid ring");
    if constexpr (XDIR) wait_window();   // the window: this cycle's requests and fragment stores (the end-of-cycle store of the previous one)
    else if constexpr (KEEP == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (KEEP == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ra_ = ra_ == AB - 1 ? 0 : ra_ + 1;
    rx_ = rx_ == XB - 1 ? 0 : rx_ + 1;
    if (!last) {
      read_a(ra_, 0, 0, alo);
      get_x(u + 1, 0, S1(), bq0);
    }
    mma_hi(bq1);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- XDEEP: one unit of a pass of group >= 1 (deep unit v, v mod 4 = I), look-ahead of three units through LDS ------------------
  auto dma_a_grp = [&](int gq, int uq, int slot) {
    Src s;
    s.x = nullptr, s.xbytes = 0u;
    s.a = reinterpret_cast<const char *>(p.img) + (size_t)gq * U * a_block_bytes;
    dma_a(s, true, uq, slot);
  };
  auto cycle_deep = [&](int v, auto i4, bool last) {
    constexpr int I = decltype(i4)::value;
    constexpr int CS[4] = {0, 1, 2, 8}, XS[4] = {3, 4, 5, 7};
    const int V = (p.ngroups - 1) * U;        // deep units of a row tile
    const int w = v + 3;
    int x_uq = -1, nx = -1;                   // this cycle's x16 request (deep unit w) | the next row tile's f32 x unit (stream end)
    if (w < V) {
      if (v == 0) {                           // the ramp: units 0, 1 came from pass 0's last cycles, unit 2 is requested here too
        dma_a_grp(1, 2, CS[2]);
        win += PA_PER_WAVE;
      }
      int gq = 1, uq = w;
      while (uq >= U) uq -= U, ++gq;
      dma_a_grp(gq, uq, CS[(I + 3) & 3]);
      win += PA_PER_WAVE;
      x_uq = uq;
    } else if (v >= V - 2) {                  // next row tile's pass 0 (its prologue state), at the XDIR kernel's request times
      nx = v - (V - 2);
      dma_a(nxt, have_nxt, nx, nx);
      win += PA_PER_WAVE;
    }
    auto x_half = [&](int part) {
      if (x_uq >= 0) {
        if (v == 0) dma_x16(2, part, XS[2]), ++win;
        dma_x16(x_uq, part, XS[(I + 3) & 3]), ++win;
      } else if (nx >= 0) {
        dma_x(nxt, nx, nx, 2 * part, 2), win += 2;
      }
    };
    read_a(CS[I], 0, H, ahi);
    mma_lo(bq0);
    x_half(0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(CS[I], 1, 0, alo);
    read_x16(XS[I], 1, bq1);
    mma_hi(bq0);
    x_half(1);
    __builtin_amdgcn_sched_barrier(0);
    read_a(CS[I], 1, H, ahi);
    mma_lo(bq1);
    __builtin_amdgcn_sched_barrier(0);
    // unit v + 1 must have landed: it was requested three windows ago, so the requests of the last TWO windows may stay in flight --
    // except on the ramp (units 1, 2 were requested in the previous window) and in the stream's last cycle (the next tile's unit 0 was)
    const bool one = v < 2 || v == V - 1;
    wait_count(one ? win : win_prev + win);
    win_prev = win, win = 0;
    __builtin_amdgcn_sched_barrier(0);
    if (!last) {
      read_a(CS[(I + 1) & 3], 0, 0, alo);
      read_x16(XS[(I + 1) & 3], 0, bq0);
    }
    mma_hi(bq1);
    __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: unit 0 (waited for); centroid units up to DA - 1 and x unit 1 stay in flight
  dma_a(cur, true, 0, 0);
  dma_x(cur, 0, 0, 0, 4);
  if constexpr (DA == 2) dma_a(cur, true, 1, 1);
  dma_x(cur, 1, 1, 0, 4);
  if constexpr (DA == 2 && PA_PER_WAVE == 1) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
  else if constexpr (DA == 2) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  win = 0;

  while (true) {
    grp_now = grp;
#pragma unroll
    for (int ti = 0; ti < TA; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
    if constexpr (SPLIT) {
#pragma unroll
      for (int ti = 0; ti < TA; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[ti][r] = 0.f;
    }
    rho_s = 0.f;
    read_a(ra_, 0, 0, alo);     // k-step (0, 0) of the tile: its unit has landed (prologue / the previous tile's last barrier)
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    get_x(0, 0, C0(), bq0);
    using C3 = std::integral_constant<int, 3>;
    if (XDEEP && grp != 0) {    // U is a multiple of 12 (host): deep unit (grp - 1) U + u sits in the slots of u mod 4
      const int v0 = (grp - 1) * U;
      for (int u = 0; u < U; u += 4) {
        cycle_deep(v0 + u, C0(), false);
        cycle_deep(v0 + u + 1, C1(), false);
        cycle_deep(v0 + u + 2, C2(), false);
        cycle_deep(v0 + u + 3, C3(), u + 4 >= U);
      }
    } else if constexpr (XDIR) {       // U is a multiple of 3 (checked by the host): the ring slots are compile-time constants
      for (int u = 0; u < U - 3; u += 3) {
        cycle(u, C0(), false);
        cycle(u + 1, C1(), false);
        cycle(u + 2, C2(), false);
      }
      cycle(U - 3, C0(), false);
      cycle(U - 2, C1(), false);
      cycle(U - 1, C2(), true);
    } else {
      for (int u = 0; u < U - 1; ++u) cycle(u, C0(), false);
      cycle(U - 1, C0(), true);
    }

    // ---- epilogue of (row tile rt, group grp): per level argmin of F, candidate test, record ---------------------------------
    if constexpr (SPLIT) {
#pragma unroll
      for (int ti = 0; ti < TA; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ti][r] = fmaf(accx[ti][r], 1.f / 2048.f, acc[ti][r]);
    }
    const long long row = rt * RF_ROWS + 32 * w8 + lrow;
    if (grp == 0) {
      const float tot = rho_s + __shfl_xor(rho_s, 32);
      rho_hat = tot * p.scal[1] * 1.0001f;  // ||x'||^2, rounded up
      xn = sqrtf(rho_hat) * 1.0001f;
      prev = 0ull;
      row_bad = false;
    }
    const int lev0 = grp * p.LPG;
#pragma unroll
    for (int l = 0; l < TA / KT; ++l) {
      const int j = lev0 + l;
      if (l >= p.LPG || j >= p.M) break;  // uniform
      const RfLevel L = p.lev[j];
      // F in place: acc <- acc * (-2 / (S_x S_c)) + A[c], then += 2 G[i][code_i][j][c] level by level (no second copy of the
      // tile: 128 accumulators + one tile's table values is what fits the 256-register budget at K = 256).  The table values of
      // MFMA tile tk+1 are requested before tile tk is computed (one L2 round trip per pass instead of one per tile: the first
      // version's per-tile fences cost 20 us per K = 256 tile, a quarter of the kernel).
      auto load4 = [&](const float *src, float4 (&v)[4]) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) v[g4] = *reinterpret_cast<const float4 *>(src + 8 * g4);
      };
      {
        const float *Aj = (TLDS ? tabA + j * 32 : p.A + (size_t)j * p.Kp) + 4 * half;
        float4 nv[4];
        load4(Aj, nv);
#pragma unroll
        for (int tk = 0; tk < KT; ++tk) {
          f32x16 &a = acc[l * KT + tk];
          float4 cv[4];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) cv[g4] = nv[g4];
          if (tk + 1 < KT) load4(Aj + 32 * (tk + 1), nv);
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            a[4 * g4] = fmaf(a[4 * g4], L.inv2, cv[g4].x), a[4 * g4 + 1] = fmaf(a[4 * g4 + 1], L.inv2, cv[g4].y);
            a[4 * g4 + 2] = fmaf(a[4 * g4 + 2], L.inv2, cv[g4].z), a[4 * g4 + 3] = fmaf(a[4 * g4 + 3], L.inv2, cv[g4].w);
          }
          if constexpr (KT > 2) __builtin_amdgcn_sched_barrier(0);
        }
      }
      for (int i = 0; i < j; ++i) {
        const unsigned int ci = (unsigned int)(prev >> (8 * i)) & 255u;
        const float *Gi = (TLDS ? tabG + ((j * (j - 1) / 2 + i) * 32 + (int)ci) * 32
                                : p.G2 + ((size_t)(i * p.K + (int)ci) * p.M + j) * p.Kp) + 4 * half;
        float4 nv[4];
        load4(Gi, nv);
#pragma unroll
        for (int tk = 0; tk < KT; ++tk) {
          f32x16 &a = acc[l * KT + tk];
          float4 cv[4];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) cv[g4] = nv[g4];
          if (tk + 1 < KT) load4(Gi + 32 * (tk + 1), nv);
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
            a[4 * g4] += cv[g4].x, a[4 * g4 + 1] += cv[g4].y, a[4 * g4 + 2] += cv[g4].z, a[4 * g4 + 3] += cv[g4].w;
          if constexpr (KT > 2) __builtin_amdgcn_sched_barrier(0);
        }
      }
      float m = INFINITY;
      int mi = 0;
#pragma unroll
      for (int tk = 0; tk < KT; ++tk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float f = acc[l * KT + tk][r];
          if (f < m) {  // ascending c inside a lane: strict < keeps the lowest index
            m = f;
            mi = 32 * tk + (r & 3) + 8 * (r >> 2) + 4 * half;
          }
        }
      }
      {
        const float om = __shfl_xor(m, 32);
        const int oi = __shfl_xor(mi, 32);
        if (om < m || (om == m && oi < mi)) m = om, mi = oi;
      }
      // bound (rf_delta): E1 = everything between F and F*
      const float q = xn + L.qpre;
      // roundings between F and F*: the table entries (A, j of G), the j adds and the fma, each relative to <= cnmax^2 + 2 q cnmax
      const float e24 = (2.f * (float)j + 3.f) * 5.97e-8f;
      const float E1 = (p.e16 * xn * L.cnmax + L.dc2 * xn + e24 * (L.cnmax * L.cnmax + 2.f * q * L.cnmax) + L.und_c * xn + L.und_x) * 1.01f;
      const float dl = (float)j * 5.97e-8f * q * 1.01f;
      float dpos;
      float delta = rf_delta(m, rho_hat, E1, p.gam, dl, dpos);
      const bool finite = (m > -3.0e38f) && (m < 3.0e38f);  // f16 overflow / NaN inputs end here
      if (!finite) delta = INFINITY;
      const float thr = m + delta;
      int cnt = 0;
#pragma unroll
      for (int tk = 0; tk < KT; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) cnt += (acc[l * KT + tk][r] <= thr) ? 1 : 0;
      const int ocnt = __shfl_xor(cnt, 32);
      const int total = cnt + ocnt;
      // ambiguous (or nothing finite): a record in this wave's own region -- slots handed out by ballot, no atomics (a single
      // global counter cost 25 ms: 4 M returning atomics on one word) -- or the row goes to the exact kernel
      const bool amb = total != 1 && !row_bad && row < p.n;
      bool overflow = amb && (!finite || total > RF_MAXC || total < 1);
      const bool want = amb && !overflow;
      const unsigned long long wmask = __ballot(want && half == 0);
      unsigned int slot = rec_n + (unsigned int)__builtin_amdgcn_mbcnt_lo((unsigned int)wmask, 0u);   // rows are lanes 0..31
      slot = __shfl(slot, lrow);
      rec_n += (unsigned int)__popcll(wmask);
      if (want && slot >= p.region_cap) overflow = true;
      if (amb) {
        if (overflow) {
          row_bad = true;
        } else {
          RfRecord *rec = rec_base + slot;
          if (half == 0) {
            rec->row = row;
            rec->prev = prev;
            rec->level = (unsigned char)j;
            rec->ncand = (unsigned char)total;
            rec->spec = (unsigned char)mi;
          }
          int pos = half == 0 ? 0 : ocnt;  // lower half's candidates first
#pragma unroll
          for (int tk = 0; tk < KT; ++tk)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (acc[l * KT + tk][r] <= thr) rec->cand[pos++] = (unsigned char)(32 * tk + (r & 3) + 8 * (r >> 2) + 4 * half);
        }
      }
      prev |= (unsigned long long)(unsigned int)mi << (8 * j);
      rho_hat = (dpos + (delta < INFINITY ? delta : 0.f)) * 1.0001f;  // upper bound of ||r*_{j+1}||^2 for any candidate
      if (half == 0 && row < p.n) p.codes[(size_t)row * p.M + j] = mi;
    }
    if (grp == p.ngroups - 1 && row_bad && half == 0 && row < p.n) p.row_flag[row] = 1;

    if (!have_nxt) break;
    cur = nxt;
    rt = rt_n, grp = grp_n;
    have_nxt = advance(rt_n, grp_n);
    if (have_nxt) src_of(rt_n, grp_n, nxt);
    else nxt.xbytes = 0u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) p.region_n[blockIdx.x * 8 + w8] = rec_n;
}

// (the body lives in a __device__ function: the host pass instantiates a kernel template's own body, and the buffer / LDS
// builtins above do not exist there)
template <int TA, int KT, bool SPLIT, bool XSPLIT, bool XDIR = false, bool XDEEP = false>
__global__ __launch_bounds__(512, 2) void rq_fast_kernel(const RfParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  rq_fast_body<TA, KT, SPLIT, XSPLIT, XDIR, XDEEP>(p, lds);
}

// ---- prep kernels (codebook only: a few hundred KB) -------------------------------------------------------------------------
// mu = mean of the level-0 centroids (f32); one thread per column
__global__ __launch_bounds__(256) void rf_mu_kernel(const float *__restrict__ C, int K, int dim, float *__restrict__ mu) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= dim) return;
  double s = 0.0;
  for (int c = 0; c < K; ++c) s += (double)C[(size_t)c * dim + k];
  mu[k] = (float)(s / (double)K);
}

// per level: max |c'_k| and max ||c'|| over its centroids (c' = c - mu on level 0); one workgroup per level
__global__ __launch_bounds__(256) void rf_level_stats_kernel(const float *__restrict__ C, int M, int K, int dim,
                                                            const float *__restrict__ mu, float *__restrict__ stat) {
  __shared__ float smax[256], snorm[256];
  const int j = blockIdx.x, t = threadIdx.x;
  float mx = 0.f, mn = 0.f;
  for (int c = t; c < K; c += 256) {
    const float *row = C + ((size_t)j * K + c) * dim;
    double ss = 0.0;
    for (int k = 0; k < dim; ++k) {
      const double v = (double)row[k] - (j == 0 ? (double)mu[k] : 0.0);
      ss += v * v;
      mx = fmaxf(mx, (float)fabs(v));
    }
    mn = fmaxf(mn, (float)sqrt(ss) * 1.00001f);
  }
  smax[t] = mx, snorm[t] = mn;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) smax[t] = fmaxf(smax[t], smax[t + s]), snorm[t] = fmaxf(snorm[t], snorm[t + s]);
    __syncthreads();
  }
  if (t == 0) stat[2 * j] = smax[0] * 1.00001f, stat[2 * j + 1] = snorm[0];
}

__device__ __forceinline__ float rf_pow2_scale(float m, int target_exp) {  // power of two S with m S in [2^(t-1), 2^t)
  if (!(m > 0.f)) return 1.f;
  int e;
  (void)frexpf(m, &e);
  int s = target_exp - e;
  s = s > 100 ? 100 : (s < -100 ? -100 : s);
  return ldexpf(1.f, s);
}

// scales and the per-level constants; one thread
__global__ void rf_levels_kernel(const float *__restrict__ stat, int M, int dim, RfLevel *__restrict__ lev, float *__restrict__ scal) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float sx = rf_pow2_scale(stat[0], 11);  // x' = c'_0 + residual: 32x head-room to the f16 maximum
  scal[0] = sx;
  scal[1] = (1.f / sx) * (1.f / sx);
  float qpre = 0.f;
  const float sq = sqrtf((float)dim) * 1.001f;
  for (int j = 0; j < M; ++j) {
    const float sc = rf_pow2_scale(stat[2 * j], 13);
    RfLevel L;
    L.sc = sc;
    L.inv2 = -2.f * (1.f / sx) * (1.f / sc);
    L.cnmax = stat[2 * j + 1];
    L.qpre = qpre;
    L.und_c = 2.f * 2.98e-8f * 1.01f * sq / sc;
    L.und_x = 2.f * 2.98e-8f * 1.01f * sq * L.cnmax / sx;
    L.dc2 = 0.f, L.pad = 0.f;
    lev[j] = L;
    qpre += L.cnmax * 1.00001f;
  }
}

__global__ __launch_bounds__(256) void rf_mus_kernel(const float *__restrict__ mu, const float *__restrict__ scal, int dim,
                                                    float *__restrict__ mus) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < dim) mus[k] = mu[k] * scal[0];
}

// image + A: one wave per (level, centroid incl. padding)
__global__ __launch_bounds__(256) void rf_image_kernel(const float *__restrict__ C, int M, int K, int Kp, int dim, int LPG, int TA,
                                                      int split, const float *__restrict__ mu, RfLevel *__restrict__ lev,
                                                      _Float16 *__restrict__ img, float *__restrict__ A) {
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wv >= M * Kp) return;
  const int j = wv / Kp, c = wv - j * Kp;
  const int g = j / LPG, rowi = (j - g * LPG) * Kp + c;  // row inside the group's TA*32-row block
  const int U = dim >> 5;
  const bool real = c < K;
  const float sc = lev[j].sc;
  double ss = 0.0, dd = 0.0;
  for (int k = lane; k < dim; k += 64) {
    float v = 0.f;
    if (real) {
      v = C[((size_t)j * K + c) * dim + k];
      if (j == 0) v = (float)((double)v - (double)mu[k]);
      ss += (double)v * (double)v;
    }
    const int arows = TA * 32 * (split ? 2 : 1);
    const _Float16 h = (_Float16)(v * sc);
    {
      const double d = (double)v - (double)(float)h / (double)sc;   // S_c is a power of two: exact
      dd += d * d;
    }
    img[(((size_t)g * U + (k >> 5)) * arows + rowi) * 32 + (k & 31)] = h;
    if (split) img[(((size_t)g * U + (k >> 5)) * arows + TA * 32 + rowi) * 32 + (k & 31)] = (_Float16)((v * sc - (float)h) * 2048.f);
  }
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off), dd += __shfl_xor(dd, off);
  if (lane == 0) A[(size_t)j * Kp + c] = real ? (float)ss : INFINITY;
  if (lane == 0 && real && !split) {   // non-negative floats order like their bits
    const float v = (float)(2.0 * (1.0 + 2.0 / 2048.0) * sqrt(dd) * 1.0001);
    atomicMax(reinterpret_cast<unsigned int *>(&lev[j].dc2), __float_as_uint(v));
  }
}

// G2[(i K + a)][j][c] = 2 c'_i[a] . c'_j[c] for i < j (f64 accumulation, rounded once), 0 elsewhere.  One 16 x 16 output tile
// per workgroup, 16-wide k tiles of both centroid blocks staged in LDS (a thread per entry walking its two rows took 0.56 ms
// at (3, 256): its second operand was a 4-byte load per lane 3 KB apart).
__global__ __launch_bounds__(256) void rf_g_kernel(const float *__restrict__ C, int M, int K, int Kp, int dim,
                                                  const float *__restrict__ mu, float *__restrict__ G2) {
  __shared__ double sa[16][17], sb[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int tiles_c = Kp / 16, tiles_a = (K + 15) / 16;
  int b = blockIdx.x;
  const int tc = b % tiles_c;
  b /= tiles_c;
  const int ta = b % tiles_a;
  b /= tiles_a;
  const int j = b % M, i = b / M;
  const int a = 16 * ta + ty, c = 16 * tc + tx;
  double s = 0.0;
  if (i < j) {  // uniform per workgroup
    for (int k0 = 0; k0 < dim; k0 += 16) {
      const int ra = 16 * ta + ty, rc = 16 * tc + ty;  // thread (ty, tx) stages element k0 + tx of rows ra (level i) and rc (level j)
      sa[ty][tx] = ra < K ? (double)C[((size_t)i * K + ra) * dim + k0 + tx] - (i == 0 ? (double)mu[k0 + tx] : 0.0) : 0.0;
      sb[ty][tx] = rc < K ? (double)C[((size_t)j * K + rc) * dim + k0 + tx] : 0.0;
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) s += sa[ty][kk] * sb[tx][kk];
      __syncthreads();
    }
  }
  if (a < K) G2[((size_t)(i * K + a) * M + j) * Kp + c] = (i < j && c < K) ? (float)(2.0 * s) : 0.f;
}

// ---- fix-up: exact chains of a record's candidates -----------------------------------------------------------------------------
// SLOTS lanes per record (one per candidate), 64 / SLOTS records per wave; the <4> instance takes the records with <= 4
// candidates (nearly all), the <8> instance the rest.  The chain is the oracle's: r = ((x - c_0) - c_1) - ... in f32 in that
// order, d = fma(r - c, r - c, d) over k = 0..dim-1.  Per 32-wide k slab the wave first STAGES the records' residual slabs in
// LDS -- lane (record, 16-byte piece) loads its piece of x (whole 128-byte lines per record: the 12 GB of re-read rows come
// at the gather rate) and of the premise centroids, subtracts in the reference's order -- then every (record, candidate) lane
// runs its 32 chain steps on the staged residual and its own centroid slab (8 x 16-byte loads from L2).
template <int SLOTS>
__global__ __launch_bounds__(256) void rf_fixup_kernel(const float *__restrict__ X, int dim, const float *__restrict__ C, int M, int K,
                                                      const RfRecord *__restrict__ rec, const unsigned int *__restrict__ region_n,
                                                      unsigned int region_cap, unsigned int waves_per_region,
                                                      unsigned char *__restrict__ row_flag) {
  constexpr int RPW = 64 / SLOTS;           // records per wave
  constexpr int PPL = RPW * 8 / 64;         // staged 16-byte pieces per lane and slab (2 at SLOTS = 4, 1 at SLOTS = 8)
  constexpr int LD = 36;                    // floats per staged row (32 + pad)
  __shared__ __attribute__((aligned(16))) float stage[4][RPW * LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned int gw = blockIdx.x * 4 + wave;
  const unsigned int region = gw / waves_per_region, chunk = gw - region * waves_per_region;
  unsigned int nrec = region_n[region];
  nrec = nrec < region_cap ? nrec : region_cap;
  if (chunk * RPW >= nrec) return;  // wave-uniform
  const RfRecord *base = rec + (size_t)region * region_cap + (size_t)chunk * RPW;
  const int nhere = (int)(nrec - chunk * RPW < (unsigned)RPW ? nrec - chunk * RPW : RPW);
  float *st = stage[wave];
  // staging duty: piece (lane & 7) of records (lane >> 3) + 8 i
  const float *sx[PPL];
  unsigned long long sprev[PPL];
  int slevel[PPL];
#pragma unroll
  for (int i = 0; i < PPL; ++i) {
    const int r = (lane >> 3) + 8 * i;
    const bool live = r < nhere;
    const RfRecord R = base[live ? r : 0];
    const bool mine = live && (SLOTS == 4 ? R.ncand <= 4 : R.ncand > 4);
    sx[i] = X + (size_t)R.row * dim + (lane & 7) * 4;
    sprev[i] = R.prev;
    slevel[i] = mine ? (int)R.level : -1;  // -1: not this instance's record, nothing staged
  }
  // compute duty: candidate `slot` of record `r`
  const int r = lane / SLOTS, slot = lane % SLOTS;
  const bool rlive = r < nhere;
  const RfRecord R = base[rlive ? r : 0];
  const bool mine = rlive && (SLOTS == 4 ? R.ncand <= 4 : R.ncand > 4);
  if (!__any(mine)) return;  // nothing of this instance's in the wave's records (nearly every wave of the <8> instance)
  const bool has = mine && slot < (int)R.ncand;
  const int cand = has ? (int)R.cand[slot] : 0;
  const float *cc = C + ((size_t)R.level * K + cand) * dim;
  float d = has ? 0.f : INFINITY;
  const int nslab = dim >> 5;  // dim % 32 == 0 on this path
  // Measured on 4.1 M records (C2 corpus, (4, 32); profiles/r03_rq_fixup_variants.txt): this form 5.8 ms (36 waves per CU,
  // no prefetch); + the next slab's x / centroid pieces prefetched in registers (16 waves per CU) 6.8 ms; candidate
  // centroids staged through LDS by whole 128-byte lines (12 waves per CU) 9.3 - 15 ms.  Occupancy beats both refinements.
  for (int s = 0; s < nslab; ++s) {
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
      if (slevel[i] < 0) continue;
      float4 v = *reinterpret_cast<const float4 *>(sx[i] + s * 32);
      for (int j = 0; j < slevel[i]; ++j) {
        const float4 c = *reinterpret_cast<const float4 *>(C + ((size_t)j * K + (int)((sprev[i] >> (8 * j)) & 255ull)) * dim + s * 32 + (lane & 7) * 4);
        v.x -= c.x, v.y -= c.y, v.z -= c.z, v.w -= c.w;
      }
      *reinterpret_cast<float4 *>(st + ((lane >> 3) + 8 * i) * LD + (lane & 7) * 4) = v;
    }
    __builtin_amdgcn_wave_barrier();  // LDS is in order within a wave
    if (has) {
      float4 c[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) c[q] = *reinterpret_cast<const float4 *>(cc + s * 32 + 4 * q);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 rr = *reinterpret_cast<const float4 *>(st + r * LD + 4 * q);
        float e;
        e = rr.x - c[q].x; d = fmaf(e, e, d);
        e = rr.y - c[q].y; d = fmaf(e, e, d);
        e = rr.z - c[q].z; d = fmaf(e, e, d);
        e = rr.w - c[q].w; d = fmaf(e, e, d);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  int best = has ? cand : 0x7fffffff;
  bool nan = has && !(d == d);              // a NaN distance in ANY live slot forces the exact re-encode (comparisons below would skip it)
#pragma unroll
  for (int off = 1; off < SLOTS; off <<= 1) {
    const float od = __shfl_xor(d, off);
    const int oc = __shfl_xor(best, off);
    nan |= (bool)__shfl_xor((int)nan, off);
    if (od < d || (od == d && oc < best)) d = od, best = oc;
  }
  if (mine && slot == 0 && (best != (int)R.spec || nan)) row_flag[R.row] = 1;  // speculation wrong or NaN distances: exact re-encode
}

// flagged rows -> index list (order irrelevant); one atomic per wave
__global__ __launch_bounds__(256) void rf_badlist_kernel(const unsigned char *__restrict__ row_flag, long long n,
                                                        long long *__restrict__ list, unsigned int *__restrict__ counters) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool bad = r < n && row_flag[r];
  const unsigned long long m = __ballot(bad);
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63;
  unsigned int base = 0u;
  if (lane == 0) base = atomicAdd(&counters[0], (unsigned int)__popcll(m));
  base = __shfl(base, 0);
  if (bad) list[base + __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u))] = r;
}

}  // namespace

// exact VALU encoder over an index list (rq_encode.hip)
int rq_encode_exact_rows(const float *x, int64_t dim, const float *codebook, int64_t M, int64_t K, int32_t *codes,
                         const long long *rows, const unsigned int *nrows, int64_t max_rows, hipStream_t stream);

}  // namespace mevi

using namespace mevi;

namespace {
struct RfPlan {
  int Kp, KT, LPG, ngroups, TA;
  bool ok, split, xsplit, xdir, xdeep;
};
RfPlan rf_plan(int64_t dim, int64_t M, int64_t K) {
  RfPlan pl = {0, 0, 0, 0, 0, false, false, false, false, false};
  if (dim % 32 != 0 || dim < 96 || dim > 8192 || M < 1 || M > 8 || K < 1 || K > 256) return pl;
  int Kp = 32;
  while (Kp < K) Kp <<= 1;
  pl.Kp = Kp;
  pl.KT = Kp / 32;
  pl.LPG = 256 / Kp;
  pl.ngroups = (int)((M + pl.LPG - 1) / pl.LPG);
  const int tiles = (int)(M < pl.LPG ? M : pl.LPG) * pl.KT;  // MFMA tiles of the fullest group
  pl.TA = (tiles <= 4 && pl.KT == 1) ? 4 : 8;
  // up to four levels of 32 centroids (the scripts' (4, 32)): the kernel waits for HBM with the matrix cores a quarter busy,
  // so the product is taken in split precision (3 MFMAs) -- an error bound 6x tighter, 6x fewer ambiguous row-levels
  pl.split = pl.TA == 4 && !getenv("MEVI_RQ_NO_SPLIT");
  // the rings, the tables and mu (dim floats) must fit the 160 KiB of LDS: wide rows give up the split image first
  const size_t cap = 160 * 1024;
  if (pl.split && rf_lds_bytes<4, true>((int)dim) > cap) pl.split = false;
  const size_t need = pl.TA == 4 ? (pl.split ? rf_lds_bytes<4, true>((int)dim) : rf_lds_bytes<4, false>((int)dim)) : rf_lds_bytes<8, false>((int)dim);
  pl.ok = need <= cap;
  // the other shapes with few columns per level (K <= 64: the x stream, not the matrix cores, paces them): x alone in split
  // precision (two MFMAs, one accumulator).  At K = 256 it was measured and dropped: main kernel 20.6 -> 26.6 ms for 4.27 ->
  // 2.68 M records (fix-up 5.6 -> 3.6 ms): 28.2 -> 32.9 ms per encode; at K = 128 the variant spills.
  pl.xsplit = !pl.split && pl.KT <= 2 && !getenv("MEVI_RQ_NO_SPLIT");
  // one level per pass (K = 128 / 256 with several levels): x is converted once per row tile and the later passes take the f16
  // fragments back from a per-workgroup scratch (rq_fast_body, XDIR); MEVI_RQ_XDIRECT=0: every pass streams x again (A/B)
  const char *xd = getenv("MEVI_RQ_XDIRECT");
  pl.xdir = pl.ok && pl.ngroups > 1 && pl.TA == 8 && pl.KT >= 4 && !pl.split && !pl.xsplit && (dim / 32) % 3 == 0 && dim / 32 >= 6 &&
            !(xd && atoi(xd) == 0);
  // ... and, where the unit count allows (a ring of three in pass 0, of four in the later passes), those later passes three units
  // ahead through the dead x ring (rq_fast_body, XDEEP); MEVI_RQ_XDEEP=0: the XDIR kernel of round 4 (A/B)
  const char *xe = getenv("MEVI_RQ_XDEEP");
  pl.xdeep = pl.xdir && (dim / 32) % 12 == 0 && !(xe && atoi(xe) == 0);
  return pl;
}
constexpr int RF_GRID = 256;  // persistent workgroups (one per CU of the MI355X); 8 record regions each
struct RfWs {
  _Float16 *img;
  float *mu, *mus, *A, *G2, *stat, *scal;
  RfLevel *lev;
  unsigned int *counters, *region_n;
  unsigned char *row_flag;
  long long *badlist;
  RfRecord *rec;
  char *xs;
  unsigned int region_cap, n_regions, grid;
};
size_t rf_carve(char *base, int64_t n, int64_t dim, int64_t M, int64_t K, const RfPlan &pl, RfWs *ws) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char *q = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return q;
  };
  const int64_t n_tiles = (n + RF_ROWS - 1) / RF_ROWS;
  const unsigned int grid = (unsigned int)(n_tiles < RF_GRID ? n_tiles : RF_GRID);
  const unsigned int n_regions = grid * 8;
  // a wave sees ceil(tiles / grid) x 32 rows x M levels.  3 - 12 % of the row-levels are ambiguous on gaussian data with a
  // random codebook, but ~50 % with a k-means codebook of 256 centroids trained on structureless residuals (256 nearly
  // equidistant centroids: the best two are closer than the f16 bound; measured) -- so there is room for EVERY
  // row-level (32 bytes each: 0.85 - 1.1 GB for the MS MARCO corpus); only rows with > 8 candidates or a wrong speculation
  // fall back to the exact kernel
  const size_t rows_per_wave = (size_t)((n_tiles + grid - 1) / grid) * 32;
  const unsigned int region_cap = (unsigned int)(rows_per_wave * (size_t)M + 64);
  char *img = take((size_t)pl.ngroups * (dim / 32) * pl.TA * 32 * 32 * 2 * (pl.split ? 2 : 1));
  char *mu = take((size_t)dim * 4), *mus = take((size_t)dim * 4);
  char *A = take((size_t)M * pl.Kp * 4);
  char *G2 = take((size_t)M * K * M * pl.Kp * 4);
  char *stat = take(64 * 4), *scal = take(256), *lev = take(sizeof(RfLevel) * 8), *counters = take(256);
  char *region_n = take((size_t)n_regions * 4);
  char *row_flag = take((size_t)n);
  char *badlist = take((size_t)n * 8);
  char *rec = take((size_t)n_regions * region_cap * sizeof(RfRecord));
  char *xs = take(pl.xdir ? (size_t)grid * (size_t)(dim / 16) * 512 * 16 : 0);    // XDIR: a row tile's f16 fragments per workgroup
  if (ws) {
    ws->xs = xs;
    ws->img = reinterpret_cast<_Float16 *>(img);
    ws->mu = reinterpret_cast<float *>(mu), ws->mus = reinterpret_cast<float *>(mus);
    ws->A = reinterpret_cast<float *>(A), ws->G2 = reinterpret_cast<float *>(G2);
    ws->stat = reinterpret_cast<float *>(stat), ws->scal = reinterpret_cast<float *>(scal);
    ws->lev = reinterpret_cast<RfLevel *>(lev);
    ws->counters = reinterpret_cast<unsigned int *>(counters);
    ws->region_n = reinterpret_cast<unsigned int *>(region_n);
    ws->row_flag = reinterpret_cast<unsigned char *>(row_flag);
    ws->badlist = reinterpret_cast<long long *>(badlist);
    ws->rec = reinterpret_cast<RfRecord *>(rec);
    ws->region_cap = region_cap, ws->n_regions = n_regions, ws->grid = grid;
  }
  return off;
}
}  // namespace

extern "C" size_t mevi_rq_encode_fast_workspace_bytes(int64_t n, int64_t dim, int64_t M, int64_t K) {
  const RfPlan pl = rf_plan(dim, M, K);
  if (!pl.ok || n <= 0) return 0;
  return rf_carve(nullptr, n, dim, M, K, pl, nullptr) + 256;
}

extern "C" int mevi_rq_encode_fast_f32(const float *x, int64_t n, int64_t dim, const float *codebook, int64_t M, int64_t K,
                                       int32_t *codes, void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(n >= 0 && dim > 0 && M > 0 && K > 0, MEVI_ERR_INVALID_ARG, "rq_encode_fast: bad shape");
  if (n == 0) return MEVI_OK;
  const RfPlan pl = rf_plan(dim, M, K);
  MEVI_REQUIRE(pl.ok, MEVI_ERR_UNSUPPORTED, "rq_encode_fast: needs dim %% 32 == 0, 96 <= dim <= ~4000 (LDS), M <= 8, K <= 256 (use mevi_rq_encode_f32)");
  MEVI_REQUIRE(x && codebook && codes && workspace, MEVI_ERR_INVALID_ARG, "rq_encode_fast: null pointer");
  MEVI_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)codebook % 16) == 0 && ((uintptr_t)workspace % 256) == 0,
               MEVI_ERR_INVALID_ARG, "rq_encode_fast: x / codebook must be 16-byte, the workspace 256-byte aligned");
  MEVI_REQUIRE(n < (1LL << 40), MEVI_ERR_UNSUPPORTED, "rq_encode_fast: too many rows");
  RfWs ws;
  MEVI_REQUIRE(workspace_bytes >= rf_carve(nullptr, n, dim, M, K, pl, nullptr), MEVI_ERR_WORKSPACE,
               "rq_encode_fast: workspace %zu bytes too small", workspace_bytes);
  (void)rf_carve(reinterpret_cast<char *>(workspace), n, dim, M, K, pl, &ws);

  const int d = (int)dim, Mi = (int)M, Ki = (int)K;
  MEVI_HIP_CHECK(hipMemsetAsync(ws.counters, 0, 256, stream));
  MEVI_HIP_CHECK(hipMemsetAsync(ws.row_flag, 0, (size_t)n, stream));
  MEVI_HIP_CHECK(hipMemsetAsync(ws.img, 0, (size_t)pl.ngroups * (dim / 32) * pl.TA * 32 * 32 * 2 * (pl.split ? 2 : 1), stream));
  hipLaunchKernelGGL(rf_mu_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, stream, codebook, Ki, d, ws.mu);
  hipLaunchKernelGGL(rf_level_stats_kernel, dim3((unsigned)Mi), dim3(256), 0, stream, codebook, Mi, Ki, d, ws.mu, ws.stat);
  hipLaunchKernelGGL(rf_levels_kernel, dim3(1), dim3(64), 0, stream, ws.stat, Mi, d, ws.lev, ws.scal);
  hipLaunchKernelGGL(rf_mus_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, stream, ws.mu, ws.scal, d, ws.mus);
  hipLaunchKernelGGL(rf_image_kernel, dim3((unsigned)((Mi * pl.Kp + 3) / 4)), dim3(256), 0, stream, codebook, Mi, Ki, pl.Kp, d, pl.LPG,
                     pl.TA, pl.split ? 1 : 0, ws.mu, ws.lev, ws.img, ws.A);
  {
    const long long blocks = (long long)Mi * Mi * ((Ki + 15) / 16) * (pl.Kp / 16);
    hipLaunchKernelGGL(rf_g_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, codebook, Mi, Ki, pl.Kp, d, ws.mu, ws.G2);
  }

  RfParams p;
  p.X = x, p.n = n, p.dim = d;
  p.img = ws.img, p.mus = ws.mus, p.A = ws.A, p.G2 = ws.G2, p.lev = ws.lev, p.scal = ws.scal;
  p.M = Mi, p.K = Ki, p.Kp = pl.Kp, p.LPG = pl.LPG, p.ngroups = pl.ngroups;
  p.codes = codes, p.row_flag = ws.row_flag, p.rec = ws.rec, p.counters = ws.counters, p.region_n = ws.region_n;
  p.region_cap = ws.region_cap;
  // The product's error, both operands rounded once to f16 (u = 2^-11), f16 x f16 products exact in f32:
  //   |x~.c~ / (S_x S_c) - x'.c'| <= ||dx|| ||c'|| + ||x'|| ||dc|| + ||dx|| ||dc|| + acc ||x~|| ||c~|| / (S_x S_c)
  // with ||dx|| <= (u + 2^-22) ||x'|| (worst case) and ||dc|| the MEASURED distance of a centroid to its image row
  // (rf_image_kernel -> RfLevel::dc2, the level's maximum; round 5: a vector's roundings do not all go the same way -- ||dc||
  // comes out at ~0.4 u ||c'||: a fifth off the whole bound, 4.29 -> 3.41 M ambiguous row-levels at (3, 256), same codes);
  // acc = 4 x 2^-24 per accumulation step (lets the matrix core truncate); 2^-22 covers the f32 roundings of x S_x - mu S_x and
  // c' S_c.  Everything enters F twice (the factor -2).  e16 multiplies ||x'|| cnmax, dc2 multiplies ||x'||.
  // (The row's own ||dx||, summed beside ||x'||^2 while pass 0 converts, takes another 28 % off the records -- and was measured
  // slower at K = 256 either way: a second running sum in a register spills (K = 128: 19.2 -> 24.1 ms), the same sum by
  // ds_add_f32 into LDS costs pass 0 1.8 ms for 1.9 ms less fix-up.  Not kept.)
  // split: (hi, lo) pairs carry 22 bits (3 x 2^-22 for the two roundings of lo and the dropped lo.lo term); the cross terms'
  // own accumulation is 2^-10 of the main chain's; no dc2.  x-split: x carries 22 bits (2^-22 + the f32 rounding), two MFMAs per step.
  const double acc_step = 4.0 * (double)dim / 16777216.0;
  p.e16 = pl.split ? (float)(2.0 * (3.0 / 4194304.0 + acc_step * (1.0 + 1.0 / 512.0) + 1.0 / 4194304.0) * 1.001)
          : pl.xsplit ? (float)(2.0 * (2.0 / 4194304.0 + acc_step * (1.0 + 1.0 / 1024.0) + 1.0 / 4194304.0) * 1.001)
                      : (float)(2.0 * ((1.0 / 2048.0 + 1.0 / 4194304.0) + acc_step * (1.0 + 1.0 / 512.0) + 1.0 / 4194304.0) * 1.001);
  p.gam = (float)((double)(dim + 2) / 16777216.0 * 1.01);   // the oracle's chain: dim fma + the subtraction, relative
  p.n_tiles = (n + RF_ROWS - 1) / RF_ROWS;
  // (experiment, round 3: reading every 256-row tile as if it were stored unit-major -- row stride 128, unit stride 32 KiB,
  // i.e. 1 KiB-contiguous DMA pieces instead of 128-byte pieces 3 KB apart -- ran 3 % faster: the piece granularity is not what
  // holds the x stream at 4 TB/s)
  p.x_row_stride = d * 4, p.x_unit_stride = 128;
  p.xs = ws.xs;
  const unsigned grid = ws.grid;
  const void *fn = nullptr;
  size_t lds_bytes = 0;
#define MEVI_RF_PICK(TA_, KT_, SP_, XS_)                                        \
  if (pl.TA == TA_ && pl.KT == KT_ && pl.split == SP_ && pl.xsplit == XS_) {    \
    fn = reinterpret_cast<const void *>(rq_fast_kernel<TA_, KT_, SP_, XS_>);    \
    lds_bytes = rf_lds_bytes<TA_, SP_>(d);                                      \
  }
  MEVI_RF_PICK(4, 1, true, false) MEVI_RF_PICK(4, 1, false, true) MEVI_RF_PICK(4, 1, false, false)
  MEVI_RF_PICK(8, 1, false, true) MEVI_RF_PICK(8, 2, false, true)
  MEVI_RF_PICK(8, 1, false, false) MEVI_RF_PICK(8, 2, false, false) MEVI_RF_PICK(8, 4, false, false) MEVI_RF_PICK(8, 8, false, false)
#undef MEVI_RF_PICK
  if (pl.xdir && pl.KT == 4) fn = reinterpret_cast<const void *>(rq_fast_kernel<8, 4, false, false, true>);
  if (pl.xdir && pl.KT == 8) fn = reinterpret_cast<const void *>(rq_fast_kernel<8, 8, false, false, true>);
  if (pl.xdeep && pl.KT == 4) fn = reinterpret_cast<const void *>(rq_fast_kernel<8, 4, false, false, true, true>);
  if (pl.xdeep && pl.KT == 8) fn = reinterpret_cast<const void *>(rq_fast_kernel<8, 8, false, false, true, true>);
  MEVI_REQUIRE(fn != nullptr, MEVI_ERR_UNSUPPORTED, "rq_encode_fast: no kernel for TA=%d KT=%d", pl.TA, pl.KT);
  MEVI_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  {
    void *args[] = {(void *)&p};
    MEVI_HIP_CHECK(hipLaunchKernel(fn, dim3(grid), dim3(512), args, lds_bytes, stream));
  }
  {
    const unsigned int wpr4 = (ws.region_cap + 15) / 16, wpr8 = (ws.region_cap + 7) / 8;
    hipLaunchKernelGGL(rf_fixup_kernel<4>, dim3((unsigned)(((size_t)ws.n_regions * wpr4 + 3) / 4)), dim3(256), 0, stream, x, d, codebook, Mi,
                       Ki, ws.rec, ws.region_n, ws.region_cap, wpr4, ws.row_flag);
    hipLaunchKernelGGL(rf_fixup_kernel<8>, dim3((unsigned)(((size_t)ws.n_regions * wpr8 + 3) / 4)), dim3(256), 0, stream, x, d, codebook, Mi,
                       Ki, ws.rec, ws.region_n, ws.region_cap, wpr8, ws.row_flag);
  }
  hipLaunchKernelGGL(rf_badlist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ws.row_flag, (long long)n, ws.badlist,
                     ws.counters);
  MEVI_HIP_CHECK(hipGetLastError());
  return rq_encode_exact_rows(x, dim, codebook, M, K, codes, ws.badlist, ws.counters, n, stream);
}

// {records appended, rows re-encoded exactly, ambiguous row-levels} of the last fast encode that used `workspace`
// (synchronises the stream: for tests and the bench's report, not for the product path)
extern "C" int mevi_rq_encode_fast_stats(const void *workspace, int64_t n, int64_t dim, int64_t M, int64_t K, int64_t *out3,
                                         void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const RfPlan pl = rf_plan(dim, M, K);
  MEVI_REQUIRE(pl.ok && workspace && out3, MEVI_ERR_INVALID_ARG, "rq_encode_fast_stats: bad arguments");
  RfWs ws;
  (void)rf_carve(reinterpret_cast<char *>(const_cast<void *>(workspace)), n, dim, M, K, pl, &ws);
  unsigned int nbad = 0;
  std::vector<unsigned int> reg(ws.n_regions);
  MEVI_HIP_CHECK(hipMemcpyAsync(&nbad, ws.counters, 4, hipMemcpyDeviceToHost, stream));
  MEVI_HIP_CHECK(hipMemcpyAsync(reg.data(), ws.region_n, (size_t)ws.n_regions * 4, hipMemcpyDeviceToHost, stream));
  MEVI_HIP_CHECK(hipStreamSynchronize(stream));
  int64_t kept = 0, wanted = 0;
  for (unsigned int v : reg) wanted += v, kept += v < ws.region_cap ? v : ws.region_cap;
  out3[0] = kept, out3[1] = nbad, out3[2] = wanted;
  return MEVI_OK;
}
