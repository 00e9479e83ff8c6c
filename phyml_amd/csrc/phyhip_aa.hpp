// phyhip_aa.hpp -- amino-acid (20-state) traversal on the FP64 matrix cores of gfx950, second generation.
//
// Why MFMA here and only here (BASELINE north star): per site-update the 20-state path does
// 2 x 4 x (20x20) matrix-vector products = 6480 flop against 1289 B of traffic; batched over patterns it is
// a dense [20 x 20] x [20 x n] contraction per (child, rate class).  The 4-state path has a 4x4 contraction per lane
// and stays on the VALU.
//
// Shape: everything runs on the four-block form  v_mfma_f64_4x4x4_4b_f64  (4 blocks of D[4x4] += A[4x4] * B[4x4],
// 16 cycles, the same 16 FMA per cycle as the 16x16x4 form), with
//     block b      = one RATE CATEGORY (C = 4; C = 2: two categories x two pattern groups; C = 1: four pattern groups)
//     columns j    = 4 patterns
//     rows i       = 4 output states of row group r (states 4r + i, r = 0..4)
//     k            = 4 input states of k-chunk t (states 4t + k, t = 0..4, ascending: the accumulation order over
//                    input states is the reference's FMA chain, src/avx.c:593-616 -- bit-identical results)
//   lane maps (measured, tools/probes/mfma4b.hip):  A lane = 16k + 4b + i,  B lane = 16k + 4b + j,  D lane = 16i + 4b + j.
// One wave therefore owns a "wave-tile" of 4 patterns x ALL categories (16 patterns x 1 category for C = 1): 25 MFMAs per
// child matrix, 20 rows = 5 row groups exactly (no padding), and
//   * the D fragments of an update ARE the B fragments its parent needs (lane 16i + 4b + j holds states i + 4r of
//     pattern j, category b, on input and on output): results are forwarded in registers, the product of the two
//     children is lane-local;
//   * the maximum over the categories of a pattern (the rescaling rule, src/avx.c:498-510) and the category mixture
//     of Lk_Core (src/lk.c:816-818) are cross-LANE operations of one wave (DPP row rotations by 4 and 8 lanes): no LDS
//     exchange, no barrier -- the first generation (one wave per (16-pattern tile, category), 16x16x4 + 4x4x4 MFMAs)
//     spent a workgroup barrier per operation on them and ran its four waves in lock-step;
//   * waves are fully independent and small: 10 000 patterns are 2500 wave-tiles, 9.8 per CU (the first generation's
//     625 four-wave tiles filled 256 CUs as "2 or 3");
//   * the A operand (the transition matrix: 12.8 KB per matrix with all four categories, natural size, nothing
//     replicated) is the same for EVERY wave of the launch.  The first generation pulled 10 KB per wave-operation
//     through the per-CU vector-memory path (5 GB per launch at cfg3, 57 % of that path's bytes); here ONE loader
//     wave per workgroup stages the two matrices of each operation into an LDS ring (kAaRing operations deep) and
//     every other wave of the workgroup feeds its MFMAs straight from LDS (ds_read_b128: 256 B/clk/CU): the
//     vector-memory path carries only children and results.
// Synchronisation inside a workgroup is two kinds of LDS words and no barrier in the loop: `ready` (operations whose
// matrices are in the ring; written by the loader) and `done[w]` (operations wave w has finished reading; read by the
// loader before it overwrites a slot).  LDS executes a wave's instructions in order, so "data, then flag" needs no fence.
//
// Device layout of an amino-acid partials buffer ("fragment-major"):
//   [wave-tile][320 doubles: aa_slot(k-chunk t, lane)],  lane = 16 (state & 3) + 4 block + (pattern & 3)
// i.e. chunk pairs (0|1), (2|3) as 16 bytes per lane and chunk 4 as 8 bytes per lane: a lane's five values move as
// 2 x dwordx4 + 1 x dwordx2, each instruction one contiguous 1 KiB / 512 B access.  The host-facing layout
// ([pattern][category][state], t_edge::p_lk_*) is restored by phyhip_get_partials / accepted by phyhip_set_partials
// (aa_off below is the only place that knows the mapping).
#pragma once

#include "phyhip_kernels.hpp"
#include "phyhip_nt2.hpp"

namespace phyhip
{

typedef double v2d __attribute__((ext_vector_type(2)));
#ifndef AA_RES_ABL
#define AA_RES_ABL 0 // (timing-only cuts of the resident form, tools/build_big_variant.sh: 1 no wait for the stores, 2 no matrix rebuild, 4 no
                     // exponentials, 8 no global copies, 64 the rebuild without its matrix-core products; 16: every
                     // division the plain way -- valid results)
#endif
constexpr int kAaT      = 5;    // k-chunks of 4 input states = row groups of 4 output states
constexpr int kAaBlock  = 320;  // doubles per wave-tile block of a partials buffer
#ifndef AA_RING
#define AA_RING 4
#endif
constexpr int kAaRing   = AA_RING;    // operations whose matrices the LDS ring holds (2 x 12.8 KB each)
#ifndef AA_MAXCONS
#define AA_MAXCONS 15
#endif
constexpr int kAaMaxCons = AA_MAXCONS;  // consumer waves per workgroup (+ 1 loader = 16 waves = 4 per SIMD at <= 128 VGPRs)
constexpr int kAaMaxConsD2 = 11; // ... of the form that loads two operations ahead (+ 1 loader = 12 waves = 3 per SIMD at <= 168 VGPRs)
constexpr int kAaMaxCons2 = 7;  // ... with two wave-tiles per consumer wave (+ 1 loader = 8 waves = 2 per SIMD at <= 256 VGPRs)


// Write matrix `mat`'s A-operand table (kAaMat doubles) from its natural [c][i][j] entries: `get(c, i, j)`.
// Lane (k, b, i) of the A operand holds P[category of block b][4r + i][4t + k]; blocks beyond the category count
// (C = 3: block 3) get zeros, so their lanes produce zeros.
template <typename F> __device__ __forceinline__ void aa_fill_atable(double *dst, int C, F get)
{
  const int cb = aa_cb(C);
  for (int e = threadIdx.x; e < kAaMat; e += blockDim.x)
  {
    const int lane = e & 63, rt = e >> 6, r = rt % kAaT, t = rt / kAaT;
    const int k = lane >> 4, b = (lane >> 2) & 3, i = lane & 3, c = b % cb;
    dst[aa_a_slot(t, r, lane)] = (c < C) ? get(c, 4 * r + i, 4 * t + k) : 0.0;
  }
}

struct FragParams
{
  const int    *indices; // nullptr: use small_idx
  int           small_idx[kSmallPm];
  int           count;
  int           C;
  const double *pmats; // natural [m][c][i][j]
  double       *afrag;
};

static __global__ __launch_bounds__(256) void aa_frag_kernel(const FragParams q)
{
  int m;
  if (q.indices) m = q.indices[blockIdx.x];
  else
  {
    m = q.small_idx[0];
#pragma unroll
    for (int k = 1; k < kSmallPm; ++k)
      if ((int)blockIdx.x == k) m = q.small_idx[k];
  }
  const double *src = q.pmats + (size_t)m * q.C * 400;
  aa_fill_atable(q.afrag + (size_t)m * kAaMat, q.C, [&](int c, int i, int j) { return src[(size_t)c * 400 + i * 20 + j]; });
}

// Host-computed transition matrices (phyhip_set_transition_matrix, the bit-exact route of src/lk.c:2360): up to
// kUploadBatch matrices per launch are read straight from the pinned staging memory (a few KB over the host link),
// written to their slots and -- for 20 states -- once more in MFMA A-operand order.  One launch replaces one copy
// command (plus one aa_frag_kernel launch) per matrix.
constexpr int kUploadBatch = 16;
struct MatUploadParams
{
  int           count, S, C;
  int           idx[kUploadBatch];
  int           shadow[kUploadBatch]; // slot that receives the matrix's OLD value first, or -1 (virtual buffers, phyhip_host.hpp)
  const double *src[kUploadBatch]; // host-pinned, device-accessible
  double       *pmats;
  double       *afrag;             // nullptr unless 20 states
};

static __global__ __launch_bounds__(256) void upload_matrices_kernel(const MatUploadParams q)
{
  extern __shared__ __attribute__((aligned(16))) double mat[]; // [C][S][S]
  int           m   = q.idx[0], shd = q.shadow[0];
  const double *src = q.src[0];
#pragma unroll
  for (int k = 1; k < kUploadBatch; ++k)
    if ((int)blockIdx.x == k) { m = q.idx[k]; src = q.src[k]; shd = q.shadow[k]; }
  const int n = q.C * q.S * q.S;
  if (shd >= 0)
  { // (as in pmat_kernel: the old value moves to the snapshot slot before anything of the matrix is overwritten)
    for (int e = threadIdx.x; e < n; e += blockDim.x) q.pmats[(size_t)shd * n + e] = q.pmats[(size_t)m * n + e];
    if (q.afrag)
      for (int e = threadIdx.x; e < kAaMat; e += blockDim.x) q.afrag[(size_t)shd * kAaMat + e] = q.afrag[(size_t)m * kAaMat + e];
  }
  double *out = q.pmats + (size_t)m * n;
  if (!q.afrag)
  { // (no A-operand table to derive: straight from the staging memory to the slot, element e by the thread that moved its old
    // value -- no LDS, so any category count fits)
    for (int e = threadIdx.x; e < n; e += blockDim.x) out[e] = src[e];
    return;
  }
  for (int e = threadIdx.x; e < n; e += blockDim.x) mat[e] = src[e];
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += blockDim.x) out[e] = mat[e];
  if (q.afrag)
    aa_fill_atable(q.afrag + (size_t)m * kAaMat, q.C, [&](int c, int i, int j) { return mat[(size_t)c * 400 + i * 20 + j]; });
}

// ---------------------------------------------------------------------------------------------
// K1 + K2 for 20 states.  Workgroup = 1 loader wave (wave 0) + up to kAaMaxCons consumer waves, one wave-tile each.
// Pipeline of a consumer: the host hands every operation over as ready-made buffer descriptors (size 0 = load disabled)
// plus forwarding flags (tips: one 32-bit allowed-state mask per pattern); within step k the children and the auxiliary words of operation k+1 are issued before the
// matrix-core phase into the second raw register set; the previous result is forwarded in registers.  One auxiliary
// dword per child: the scale word of an internal child or the aligned four tip codes of a tip child (the host points the
// descriptor at whichever row exists).
// ---------------------------------------------------------------------------------------------
// ABL (diag build only, results INVALID): 1 no matrix phase, 2 no cross-lane maximum, 4 children loads and result stores
// zero-sized, 8 no ring hand-shake (nothing is loaded), 16 no all-ones test
// ARGS: one- and two-operation launches (an SPR regraft candidate) carry their records in the kernel arguments
// (TreeParams::arg_ir / arg_xr, as the nucleotide kernel's short launches do): no staged copy in front of the launch.  A
// separate instantiation: the records of the list form must stay provably uniform (scalar loads through the noalias
// parameters) -- selecting between two pointers at run time turned every record load into a vector load (+35 % kernel time).
// INL (list form only): an operation may carry ONE child that is a virtual tip x tip result (kOpCh1 / kOpCh2, phyhip_host.hpp):
// its two matrix tables ride in the operation's ring item (four tables instead of two; the host hands out the table slots of
// the 8-table ring and says what must have been released before an item is written), its tips' mask words come with the
// operation's children, and the consumer computes it inside the step -- exactly the step its defining operation would have
// been (matrix columns for one-state tips, products through the matrix cores otherwise, the all-ones rule, the rescaling
// rule) -- instead of that operation occupying a pipeline step of its own.
// NT (list form only): wave-tiles per consumer wave, 1 or 2 (see the consumers below): at most kAaMaxCons2 consumer waves then -- eight
// waves per workgroup, two per SIMD, 256 registers each.
// D2 (list form only): the children of operation k + 2 are requested during step k, as soon as step k has taken its own operands
// out of the registers they arrived in (the default requests operation k + 1's at the top of step k): a load has two steps to
// arrive instead of one -- for alignments whose workgroups hold at most kAaMaxConsD2 consumer waves (three waves per SIMD: the
// operands' second copy needs ~150 registers), where a step is too short to cover a trip to memory under load.
// RES (argument form only): the RESIDENT evaluator of small 20-state alignments -- the workgroups stay on the device and take one
// short evaluation after the other (an SPR regraft candidate: up to four matrices rebuilt, one or two partial updates, the edge
// likelihood; src/spr.c:640-646) from the command record of phyhip_kernels.hpp's resident protocol, in resident_nt2_kernel's word
// format (+ word 34: the model's epoch).  Per command every workgroup rebuilds the queued matrices itself -- pmat20_kernel's
// arithmetic, four waves per matrix, straight into the ring slots its consumers read them from (workgroup 0 also stores them in
// the two global tables) -- so a candidate is no launch at all instead of two; tables that were not rebuilt come from the global
// table by LDS-DMA as usual.  Geometry: up to kAaMaxCons2 wave-tiles per workgroup (eight waves: a command's records and the
// staged state live in registers for the whole loop -- 256 of them per lane); instances whose LAUNCHED form has one tile per
// workgroup (aa_nw = 1, up to 256 tiles), so that one record per tile is one record per workgroup of the launched form and
// the host's sum is the same double whoever served the evaluation.
// One (matrix, row group r) unit of a transition-matrix rebuild on the matrix cores, by ONE wave: the entries P_c[4r + i][.] of
// every category c at once (block = category), written in MFMA A-operand order into the tables `tabs` whose bit is set in `mask`
// (LDS) and, natural layout, to `out` (global; nullptr: nowhere).  P_c = (U diag(e_c)) V (src/models.c:275-298) is computed as its
// transpose, D = V^T (U e_c)^T: with A = V^T's block (k-chunk kc, column chunk t) and B = the (U e_c) block of row group r, the D
// fragment of lane 16 i' + 4 b + j' is P_c[4r + j'][4t + i'] -- exactly what lane (k = i', b, i = j') of the A-operand table holds for
// (row group r, k-chunk t): no transposition, no exchange.  Arithmetic as pmat20_kernel's, bit for bit: the products U[i][k] e_c[k]
// rounded first, one fused multiply-add per eigen index in ascending order from zero (five k-chunks of the 4x4x4 MFMA, whose four
// terms accumulate in ascending k), the 1e-100 floor, the row sum over ascending columns (four lanes 16 apart per column chunk),
// the division.
// NU: units one wave works through side by side (their chains of dependent LDS reads, matrix-core accumulations, shuffles and
// divisions interleave); unit u of the NU is (expt[u], r[u], out[u], mask[u]); a negative r[u] means "none".
template <int C_, int NU>
__device__ __forceinline__ void aa_build_units(const double *const (&expt)[NU] /* each [C][20] */, const double *Us, const double *Vs, const int (&r)[NU],
                                               const int lane, double *const (&out)[NU], double *const tab0, const int (&ta)[NU], const int (&tb)[NU],
                                               const unsigned (&more)[NU], unsigned long long *stamps = nullptr)
{
  // (PHYHIP_RESIDENT_STATS: [16 + k] the wave's wall-clock ticks up to -- 0 operands read, 1 products, 2 row sums, 3 divisions, 4 stored)
  const unsigned long long t_in = stamps ? wall_clock64() : 0ull;
  auto mark = [&](const int k, const double dep) {
    if (stamps)
    {
      unsigned long long z;
      __builtin_memcpy(&z, &dep, 8);
      z = __builtin_amdgcn_readfirstlane((unsigned)z) & 0u; // (the stamp is taken behind what `dep` depends on)
      if (lane == 0) (void)__hip_atomic_fetch_add(&stamps[16 + k], wall_clock64() - t_in + z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  constexpr int CB = C_ == 1 ? 1 : (C_ == 2 ? 2 : 4);
  const int k = lane >> 4, b = (lane >> 2) & 3, x = lane & 3, c = b % CB;
  const bool idle = c >= C_; // (C = 3: block 3 carries no category -- its table entries are zeros)
  // every operand out of LDS first, side by side (one wait), then the arithmetic
  double uu[NU][kAaT], ee[NU][kAaT], vv[kAaT][kAaT];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int kc = 0; kc < kAaT; ++kc)
    {
      uu[u][kc] = Us[(4 * (r[u] < 0 ? 0 : r[u]) + x) * 20 + 4 * kc + k];
      ee[u][kc] = expt[u][(idle ? 0 : c) * 20 + 4 * kc + k];
    }
#pragma unroll
  for (int kc = 0; kc < kAaT; ++kc)
#pragma unroll
    for (int t = 0; t < kAaT; ++t) vv[kc][t] = Vs[(4 * kc + k) * 20 + 4 * t + x]; // (the same A operands for every unit)
  double w[NU][kAaT], D[NU][kAaT], s[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int kc = 0; kc < kAaT; ++kc) w[u][kc] = uu[u][kc] * ee[u][kc];
  mark(0, w[NU - 1][kAaT - 1] + vv[kAaT - 1][kAaT - 1]);
#pragma unroll
  for (int t = 0; t < kAaT; ++t)
  {
    double acc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = 0.0;
#pragma unroll
    for (int kc = 0; kc < kAaT; ++kc)
#pragma unroll
      for (int u = 0; u < NU; ++u)
        acc[u] = (AA_RES_ABL & 64) ? __builtin_fma(vv[kc][t], w[u][kc], acc[u]) : __builtin_amdgcn_mfma_f64_4x4x4f64(vv[kc][t], w[u][kc], acc[u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < NU; ++u) D[u][t] = (acc[u] < kSmallPij) ? kSmallPij : acc[u]; // src/models.c:293
  }
  mark(1, D[NU - 1][kAaT - 1] + D[0][0]);
  // row sums in ascending column order (src/models.c:296-297), on the matrix cores as well: this lane's D fragment of column chunk t
  // IS the B operand B[k = i'][j = j'] = P[4r + j'][4t + i'] of a product with an all-ones A -- D[i][j'] = C + 1 P[..][4t] + 1 P[..][4t + 1]
  // + 1 P[..][4t + 2] + 1 P[..][4t + 3], one fused multiply-add per term in ascending k (the order the products above rest on), and a
  // fused multiply-add by one is the addition: the chain over the five chunks is the reference's running sum, and it arrives in all
  // four lanes (i) of row j' at once.  (Five matrix-core instructions per unit; gathering the four lanes' values by half-exchanges
  // and adding them cost 50 vector instructions per unit -- a third of the rebuild, which is bound by what one CU's SIMDs can issue.)
#pragma unroll
  for (int u = 0; u < NU; ++u) s[u] = 0.0;
#pragma unroll
  for (int t = 0; t < kAaT; ++t)
#pragma unroll
    for (int u = 0; u < NU; ++u) s[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, D[u][t], s[u], 0, 0, 0);
  mark(2, s[0] + s[NU - 1]);
  bool plain = false;
#pragma unroll
  for (int u = 0; u < NU; ++u) plain = plain || !(s[u] > 0x1p-500 && s[u] < 0x1p500);
  plain = (AA_RES_ABL & 16) || __builtin_amdgcn_ballot_w64(plain) != 0;
  double P[NU][kAaT];
  if (plain)
  {
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int t = 0; t < kAaT; ++t) P[u][t] = D[u][t] / s[u];
  }
  else
  {
#pragma unroll
    for (int u = 0; u < NU; ++u)
    {
      const double r0 = __builtin_amdgcn_rcp(s[u]);
      const double e0 = __builtin_fma(-s[u], r0, 1.0), r1 = __builtin_fma(r0, e0, r0);
      const double e1 = __builtin_fma(-s[u], r1, 1.0), rc = __builtin_fma(r1, e1, r1);
#pragma unroll
      for (int t = 0; t < kAaT; ++t)
      {
        const double q0 = D[u][t] * rc, rem = __builtin_fma(-s[u], q0, D[u][t]);
        P[u][t] = __builtin_fma(rem, rc, q0);
      }
    }
  }
  mark(3, P[0][0] + P[NU - 1][kAaT - 1]);
  // where the entries go: the unit's first two tables (numbers of kAaMat-sized tables from tab0; a matrix that is read in one place
  // only has the same number twice -- the same value to the same address) without a test per entry; further tables (a matrix
  // read in three places or more: bits of `more`) the slow way
#pragma unroll
  for (int u = 0; u < NU; ++u)
    if (r[u] >= 0)
    {
      double *const d0 = tab0 + (size_t)ta[u] * kAaMat, *const d1 = tab0 + (size_t)tb[u] * kAaMat;
#pragma unroll
      for (int t = 0; t < kAaT; ++t)
      {
        const double p = idle ? 0.0 : P[u][t];
        const int    sl = aa_a_slot(t, r[u], lane);
        d0[sl] = p;
        if (tb[u] != ta[u]) d1[sl] = p;
        for (unsigned m = more[u]; m; m &= m - 1) tab0[(size_t)__builtin_ctz(m) * kAaMat + sl] = p;
      }
      // natural layout [c][row][col]: the lanes of the first block of each category (C < 4: the other blocks repeat them);
      // written through (device scope): see the resident form's stores below
      if (out[u] && !idle && b < CB)
#pragma unroll
        for (int t = 0; t < kAaT; ++t)
          __hip_atomic_store(&out[u][(size_t)(c * 20 + 4 * r[u] + x) * 20 + 4 * t + k], P[u][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  mark(4, 0.0);
}

// Resident form: one A-operand table (12.5 pieces of 1 KiB) from the global table into a ring slot by LDS-DMA, at device scope (what
// another workgroup's loader wave wrote through a command ago is what arrives); by one wave, nothing waited for.
__device__ __forceinline__ void aa_resident_table(const __amdgpu_buffer_rsrc_t af_rsrc, double *to_d, const unsigned off, const int lane)
{
  typedef __attribute__((address_space(3))) void *lds_ptr;
  char *to = reinterpret_cast<char *>(to_d);
#pragma unroll
  for (int g = 0; g < 12; ++g)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(af_rsrc, (lds_ptr)(to + g * 1024), 16, (unsigned)g * 1024u + (unsigned)lane * 16u, off, 0, 16);
  if (lane < 32)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(af_rsrc, (lds_ptr)(to + 12 * 1024), 16, 12u * 1024u + (unsigned)lane * 16u, off, 0, 16);
}

struct AaResident
{
  ResidentCtl   ctl;
  const double *evec, *ivec, *eval, *rates; // U, U^-1, eigenvalues, category rates (device memory: re-read when the epoch changes)
  double       *pmats_rw, *afrag_rw;        // the two global matrix tables
  // PHYHIP_RESIDENT_STATS: where workgroup 0's time goes per command, wall-clock ticks summed over the commands -- [0] command
  // seen -> parsed, [1] -> eigen system / exponentials staged, [2] -> matrices built, [3] -> global copies issued + records taken out,
  // [4] -> evaluation done and posted; [7] commands
  unsigned long long *stamps;
  int                 stamp_wg; // (the workgroup whose time is taken: PHYHIP_RESIDENT_STATS - 1)
};
constexpr int kResidentAaWords = 10 + 2 * 12 + 1;

template <int C_, bool DBG = false, int ABL = 0, bool ARGS = false, bool INL = false, int NT = 1, bool D2 = false, bool RES = false>
__global__ __launch_bounds__((NT == 2 || RES) ? 64 * (kAaMaxCons2 + 1) : (D2 ? 64 * (kAaMaxConsD2 + 1) : 64 * (kAaMaxCons + 1))) void traverse_aa_kernel(const TreeParams q_in, const IssueRec *__restrict__ irec,
                                                                            const ExecRec *__restrict__ xrec,
                                                                            const double *__restrict__ afrag, int n_frag_mats,
                                                                            const uint32_t *__restrict__ tip_masks,
                                                                            unsigned long long *dbg = nullptr, const AaResident rs = AaResident())
{
  static_assert(!RES || (ARGS && !DBG && ABL == 0 && !INL && NT == 1 && !D2), "resident form: argument form only");
  __shared__ __attribute__((aligned(16))) double ring[kAaRing][2][kAaMat];
  // (read and written with relaxed workgroup-scope atomics: those stay plain ds_read / ds_write instructions, whereas a
  // volatile access to LDS is compiled as a flat access behind a vmcnt(0) wait)
  __shared__ int      s_ready;             // operations (+ the evaluation edge) whose matrices are in the ring
  __shared__ int      s_done[16];          // per consumer wave: operations whose matrices it has finished reading
  __shared__ double   s_wsum[16];          // per wave-tile of the workgroup: its share of the workgroup's sum
  __shared__ unsigned long long stamps[DBG ? 64 * 8 : 1];
  __shared__ int      s_spare[2];          // RES: matrices rebuilt into the ring's spare item (read by nothing in this command), or -1
  __shared__ unsigned long long s_t4;      // RES + PHYHIP_RESIDENT_STATS: when this command's evaluation began
  __shared__ unsigned s_built;             // RES: bit 2 j + w: table w of item j was rebuilt into its ring slot by this workgroup
  // Resident form: the first operation's children, requested by the command loop (end of the kernel) when the command is parsed --
  // in front of the matrix rebuild, ~3 us in which nothing else needs them -- instead of by the evaluation when it starts: a trip
  // to memory (1.3 us, the longest single wait of a resident evaluation) less on its chain.  The evaluation takes them from here.
  u32x4    ext_a01 = {0u, 0u, 0u, 0u}, ext_a23 = {0u, 0u, 0u, 0u}, ext_b01 = {0u, 0u, 0u, 0u}, ext_b23 = {0u, 0u, 0u, 0u};
  u32x2    ext_a4 = {0u, 0u}, ext_b4 = {0u, 0u};
  unsigned ext_xa = 0, ext_xb = 0;
  // One evaluation: everything a launched kernel does (the whole kernel in the launched forms; once per command in the resident one)
  auto run = [&](const TreeParams &q) __attribute__((always_inline)) {
  constexpr int T   = kAaT;
  auto IR = [&](int i) -> IssueRec {
    if constexpr (ARGS) return i ? q.arg_ir[1] : q.arg_ir[0];
    else return irec[i];
  };
  auto XR = [&](int i) -> ExecRec {
    if constexpr (ARGS) return i ? q.arg_xr[1] : q.arg_xr[0];
    else return xrec[i];
  };
  constexpr int CB  = C_ == 1 ? 1 : (C_ == 2 ? 2 : 4); // blocks (categories) per pattern
  constexpr int NPW = 16 / CB;                          // patterns per wave-tile
  // Cache policy of what an evaluation stores.  Resident form: WRITTEN THROUGH at device scope (sc1, = a relaxed device-scope atomic
  // store in the memory model) -- the workgroups never end, and what they store must be in memory, for kernels on other XCDs and
  // for the host's copies, when the sums are posted.  Written through, a plain wait for the stores' acknowledgement is all the
  // posting needs; the alternative -- a release fence per command -- writes the XCD's whole L2 back (buffer_wbl2: ~2 us each, two
  // per command, of a 15 us command).  What the resident workgroups READ of each other's stores (the global A-operand tables
  // workgroup 0 rewrites) they read at device scope likewise; what launched kernels or copies wrote in between is re-read behind the
  // acquire fence the host asks for (command word 1, bit 2).
  constexpr int kStAux = RES ? 16 : PHYHIP_STORE_AUX, kStAuxW = RES ? 16 : 0, kTabAux = RES ? 16 : 0;
  auto gst = [](auto *ptr, const auto v) {
    if constexpr (RES) __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *ptr = v;
  };
  static_assert(C_ >= 1 && C_ <= 4, "one MFMA block per category: at most four");
  static_assert(!INL || (!ARGS && !DBG && ABL == 0), "in-step tip x tip children: list form only");
  static_assert(NT == 1 || (NT == 2 && !ARGS && !DBG && ABL == 0), "two wave-tiles per wave: list form only");
  static_assert(!D2 || (!ARGS && !DBG && ABL == 0), "loads two operations ahead: list form only");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw   = (int)(blockDim.x >> 6) - 1; // consumer waves
  const int tpw  = NT == 1 ? nw : q.aa_tpw;     // wave-tiles per workgroup (the same in every form of an instance: the block sums are)
  const int C    = q.C;
  const size_t ntiles = (size_t)(q.Ppad / NPW);
  // (the list form is padded to an even length by the host -- the nucleotide kernel's convention; the argument form is not)
  const int n_ops   = ARGS ? q.n_real_ops : q.n_ops;
  const int n_items = n_ops + (q.edge_eval ? 1 : 0);

  if (threadIdx.x < 16)
  {
    const size_t tl = (size_t)blockIdx.x * tpw + threadIdx.x * NT; // (the wave's first tile)
    s_done[threadIdx.x] = ((int)threadIdx.x < nw && (int)threadIdx.x * NT < tpw && tl < ntiles) ? 0 : 0x7fffffff; // absent waves never hold a slot
  }
  if (threadIdx.x == 0) s_ready = 0;
  __syncthreads();
  const bool stamper = DBG && blockIdx.x == gridDim.x / 2 && threadIdx.x == 64;
#define PHY_STAMP(k_, i_)                                                                                              \
  if (DBG)                                                                                                             \
  {                                                                                                                    \
    const unsigned long long t_ = __builtin_readcyclecounter();                                                        \
    if (stamper && (k_) < 64) stamps[(k_) * 8 + (i_)] = t_;                                                            \
  }

  if (wave == 0)
  {
    __builtin_amdgcn_s_setprio(3); // a late matrix stalls every consumer of the workgroup: the loader issues first
    // ---- loader: the two A tables of item j (operation j, or the evaluation edge's matrix) -> ring[j % kAaRing] -------
    const __amdgpu_buffer_rsrc_t af_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(afrag), 0, (int)((size_t)n_frag_mats * kAaMat * 8), 0x00020000);
    constexpr int      kPieces = 2 * kAaMat * 8 / 1024; // 25 x 1 KiB per item
    constexpr unsigned kMatB   = kAaMat * 8;            // 12 800 B: piece 12 straddles the two tables
    static_assert(2 * kAaMat * 8 == kPieces * 1024 && kPieces == 25, "an item is 25 pieces of 1 KiB (the vmcnt(25) below)");
    // LDS-DMA (buffer_load_dwordx4 ... lds): the pieces go from L2 straight into the ring, no registers, no ds_write; a
    // wave's pieces land in issue order, so "item j has landed" is vmcnt <= kPieces once item j+1 has been issued -- two
    // items in flight (the 6-bit vmcnt counter allows 63 pieces).
    typedef __attribute__((address_space(3))) void *lds_ptr;
    int  done_seen = 0, flagged = 0; // items known to be released by every consumer / items published in s_ready
    // The loader's own flag accesses are written as instructions: the compiler orders every LDS access it can see behind
    // ALL pending LDS-DMA of the wave (s_waitcnt vmcnt(0)), which would allow only one item in flight.
    const unsigned a_ready = (unsigned)(uintptr_t)(lds_ptr)&s_ready, a_done = (unsigned)(uintptr_t)(lds_ptr)&s_done[lane & 15];
    auto publish   = [&](int n) {
      if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(a_ready), "v"(n) : "memory");
      flagged = n;
    };
    auto slot_free = [&](int need) { // every consumer has finished item need - 1
      if (done_seen >= need) return true;
      int d;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a_done) : "memory");
      if (__builtin_amdgcn_ballot_w64(d < need) != 0) return false;
      done_seen = need;
      return true;
    };
    if constexpr (RES)
    { // what the workgroup rebuilt itself sits in its slot already; every other table comes from the global table, one by one
      const unsigned built = __builtin_amdgcn_readfirstlane(s_built);
      for (int j = 0; j < n_items; ++j)
      {
        double *slot = &ring[j % kAaRing][0][0];
        if (j < n_ops)
        {
          const IssueRec rj = IR(j);
          if (!((built >> (2 * j)) & 1u)) aa_resident_table(af_rsrc, slot, rj.c1_data.x, lane);
          if (!((built >> (2 * j + 1)) & 1u)) aa_resident_table(af_rsrc, slot + kAaMat, rj.c2_data.x, lane);
        }
        else if (!((built >> (2 * j)) & 1u)) aa_resident_table(af_rsrc, slot, (unsigned)q.e_pm * kMatB, lane);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
      asm volatile("" ::: "memory");
      publish(n_items);
      if (rs.stamps && (int)blockIdx.x == rs.stamp_wg && lane == 0) (void)__hip_atomic_fetch_add(&rs.stamps[8 + 6], wall_clock64() - s_t4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (blockIdx.x == 0 && !(AA_RES_ABL & 8))
      { // while the consumers are at work: workgroup 0's copies of the rebuilt tables go to the global A-operand table (the
        // launches that follow read them there), 16 bytes per lane and store
        auto copy = [&](const double *src, const unsigned off) {
          const u32x4 *s4 = reinterpret_cast<const u32x4 *>(src);
          for (int e = lane; e < kAaMat / 2; e += 64) __builtin_amdgcn_raw_buffer_store_b128(s4[e], af_rsrc, (unsigned)e * 16u, off, kStAux);
        };
        for (int j = 0; j < n_items; ++j)
        {
          const double *slot = &ring[j % kAaRing][0][0];
          if (j < n_ops)
          {
            const IssueRec rj = IR(j);
            if ((built >> (2 * j)) & 1u) copy(slot, rj.c1_data.x);
            if ((built >> (2 * j + 1)) & 1u) copy(slot + kAaMat, rj.c2_data.x);
          }
          else if ((built >> (2 * j)) & 1u) copy(slot, (unsigned)q.e_pm * kMatB);
        }
        for (int u = 0; u < 2; ++u)
        {
          const int m = __builtin_amdgcn_readfirstlane(s_spare[u]);
          if (m >= 0) copy(&ring[kAaRing - 1][u][0], (unsigned)m * kMatB);
        }
      }
    }
    else
    for (int j = 0; j < ((ABL & 8) ? 0 : n_items); ++j)
    {
      unsigned off1, off2, off3 = 0, off4 = 0;
      int      need = j - kAaRing + 1, tslot = (j % kAaRing) * 2; // (two tables per item: the slot is free once every consumer has finished item j - kAaRing)
      bool     four = false;
      if (j < n_ops)
      {
        const IssueRec rj = IR(j);
        off1 = rj.c1_data.x; off2 = rj.c2_data.x;
        if constexpr (INL)
        { // the host's plan for this item (flush_impl): table slot, what must be released first, and -- with an in-step child --
          // the two more tables
          four = rj.c1_tip.x != 0;
          off3 = (unsigned)rj.c1_tip.base; off4 = (unsigned)(rj.c1_tip.base >> 32);
          need = (int)rj.c1_tip.bytes; tslot = (int)rj.c2_tip.x;
        }
      }
      else
      {
        off1 = off2 = (unsigned)q.e_pm * kMatB;
        if constexpr (INL) { need = q.aa_e_need; tslot = q.aa_e_slot; }
      }
      if (!slot_free(need))
      { // the consumers may be waiting for what is still in flight: publish it before waiting for them
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
        asm volatile("" ::: "memory");
        publish(j);
        while (!slot_free(need)) __builtin_amdgcn_s_sleep(2);
      }
      asm volatile("" ::: "memory");
      char *slot = reinterpret_cast<char *>(&ring[0][0][0]) + (size_t)tslot * kMatB;
      auto pair = [&](char *to, const unsigned oa, const unsigned ob) { // two tables = 25 pieces of 1 KiB
#pragma unroll
        for (int g = 0; g < kPieces; ++g)
        {
          const unsigned b = (unsigned)g * 1024u + (unsigned)lane * 16u; // byte inside the pair
          lds_ptr dst = (lds_ptr)(to + g * 1024);
          if (g * 1024 + 1024 <= (int)kMatB) __builtin_amdgcn_raw_ptr_buffer_load_lds(af_rsrc, dst, 16, b, oa, 0, 0);
          else if (g * 1024 >= (int)kMatB) __builtin_amdgcn_raw_ptr_buffer_load_lds(af_rsrc, dst, 16, b - kMatB, ob, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(af_rsrc, dst, 16, b < kMatB ? b + oa : b - kMatB + ob, 0, 0, 0);
        }
      };
      pair(slot, off1, off2);
      if (INL && four) pair(slot + 2 * kMatB, off3, off4);
      if (flagged < j)
      { // everything but the item just issued has landed
        if (INL && four) __builtin_amdgcn_s_waitcnt(0xCF72); // vmcnt(50)
        else __builtin_amdgcn_s_waitcnt(0x4F79);             // vmcnt(25)
        asm volatile("" ::: "memory");
        publish(j);
      }
    }
    if constexpr (!RES)
    {
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
      asm volatile("" ::: "memory");
      publish(n_items);
    }
  }
  else
  {
    // ---- consumers -------------------------------------------------------------------------------------------------
    // A consumer wave owns NT wave-tiles (NT = 2: the list form of alignments with enough tiles, flush_impl): every phase of a
    // step runs for both tiles back to back -- two independent chains of loads, matrix-core accumulations and cross-lane scans
    // for the scheduler to interleave -- and the A operand (the same for every tile) is read from the ring ONCE per k-chunk for
    // both tiles' MFMAs; the operation records, flag tests, descriptor set-up and the ring hand-shake are per STEP, not per tile.
    const int w = wave - 1;
    long long tile[NT];
    bool      tact[NT]; // the tile exists (the last wave of a workgroup with an odd number of tiles, the end of the alignment)
#pragma unroll
    for (int i = 0; i < NT; ++i)
    {
      tile[i] = (long long)blockIdx.x * tpw + w * NT + i;
      tact[i] = (w * NT + i < tpw) && tile[i] < (long long)ntiles;
    }
    if (tact[0])
    {
      if constexpr (RES) if (rs.stamps && (int)blockIdx.x == rs.stamp_wg && w == 0 && lane == 0) (void)__hip_atomic_fetch_add(&rs.stamps[8 + 0], wall_clock64() - s_t4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int kk = lane >> 4, b = (lane >> 2) & 3, jj = lane & 3; // state residue / MFMA block / pattern within the group
      const int c  = b % CB;                                        // this lane's rate category
      const bool idle = c >= C_;                                    // C = 3: block 3 carries nothing
      const int       tips = q.tip_count;
      const bool      cls  = q.class_axis != 0;
      constexpr unsigned kNowhere = 0x7fff0000u; // an offset beyond every buffer: loads through it return zeros, stores are dropped
      long long p0[NT];
      bool      pact[NT];
      unsigned  voff_d16[NT], voff_d8[NT], voff_s[NT], voff_t[NT], voff_sst[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i)
      {
        p0[i]   = tile[i] * NPW + (b / CB) * 4 + jj; // < Ppad when the tile exists
        pact[i] = tact[i] && p0[i] < q.P && !idle;
        const unsigned blk_bytes = (unsigned)((size_t)tile[i] * kAaBlock * 8);
        voff_d16[i] = tact[i] ? blk_bytes + lane * 16 : kNowhere;        // chunk pairs
        voff_d8[i]  = tact[i] ? blk_bytes + 2048 + lane * 8 : kNowhere;  // chunk 4
        // class axis (mixture classes as categories): every class has its own scale vector, [buffer][class][pattern]
        voff_s[i] = tact[i] ? ((cls ? (unsigned)(idle ? 0 : c) * (unsigned)q.Ppad : 0u) + (unsigned)p0[i]) * 4u : kNowhere;
        voff_t[i] = tact[i] ? (unsigned)p0[i] * 4u : kNowhere; // tip rows: one allowed-state mask (32 bits) per pattern
        // the scale word of a pattern (of a (class, pattern)) is stored by ONE lane; the others aim past the end of the buffer
        voff_sst[i] = (tact[i] && kk == 0 && !idle && (cls || c == 0)) ? voff_s[i] : kNowhere;
      }

      struct Frag
      { // a lane's five k-chunk values as they come from memory
        u32x4 p01, p23;
        u32x2 p4;
      };
      struct Raw
      {
        Frag     a, b;
        unsigned xa, xb; // scale word, or the dword holding the tip code
        unsigned xt;     // INL: the mask word of the in-step child's SECOND tip
      };
      auto rsrc = [](const Desc &d) {
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(d.base), 0, (ABL & 4) ? 0 : (int)d.bytes, 0x00020000);
      };
      auto load_frag = [](Frag &f, const __amdgpu_buffer_rsrc_t r, unsigned v16, unsigned v8) {
        f.p01 = __builtin_amdgcn_raw_buffer_load_b128(r, v16, 0, PHYHIP_LOAD_AUX);
        f.p23 = __builtin_amdgcn_raw_buffer_load_b128(r, v16 + 1024, 0, PHYHIP_LOAD_AUX);
        f.p4  = __builtin_amdgcn_raw_buffer_load_b64(r, v8, 0, PHYHIP_LOAD_AUX);
      };
      auto unpack = [](const Frag &f, double (&x)[T]) {
        __builtin_memcpy(&x[0], &f.p01, 16);
        __builtin_memcpy(&x[2], &f.p23, 16);
        __builtin_memcpy(&x[4], &f.p4, 8);
      };
      auto pack = [](const double (&x)[T], Frag &f) {
        __builtin_memcpy(&f.p01, &x[0], 16);
        __builtin_memcpy(&f.p23, &x[2], 16);
        __builtin_memcpy(&f.p4, &x[4], 8);
      };
      auto issue_children = [&](const IssueRec &o, Raw (&r)[NT]) {
        const __amdgpu_buffer_rsrc_t d1 = rsrc(o.c1_data), d2 = rsrc(o.c2_data), g1 = rsrc(o.c1_scale), g2 = rsrc(o.c2_scale);
#pragma unroll
        for (int i = 0; i < NT; ++i)
        {
          load_frag(r[i].a, d1, voff_d16[i], voff_d8[i]);
          load_frag(r[i].b, d2, voff_d16[i], voff_d8[i]);
          r[i].xa = __builtin_amdgcn_raw_buffer_load_b32(g1, o.c1_scale.x ? voff_t[i] : voff_s[i], 0, 0);
          r[i].xb = __builtin_amdgcn_raw_buffer_load_b32(g2, o.c2_scale.x ? voff_t[i] : voff_s[i], 0, 0);
          if constexpr (INL) r[i].xt = __builtin_amdgcn_raw_buffer_load_b32(rsrc(o.c2_tip), voff_t[i], 0, 0); // (size 0 without such a child)
        }
      };
      // cross-lane helpers.  Lanes that share a (pattern, category) differ in bits 4-5 (the state residue): the two
      // half-exchange instructions of gfx950 put both partners' values side by side without an LDS round trip.
      auto xor16_pair = [](unsigned v, unsigned &a, unsigned &bb) {
        auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        a = r[0]; bb = r[1];
      };
      auto xor32_pair = [](unsigned v, unsigned &a, unsigned &bb) {
        auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        a = r[0]; bb = r[1];
      };
      auto and_states = [&](unsigned v) { // AND over the four lanes (kk = 0..3) of a (pattern, category)
        unsigned a, bb;
        xor16_pair(v, a, bb); v = a & bb;
        xor32_pair(v, a, bb); return a & bb;
      };
      auto max_states = [&](unsigned v) {
        unsigned a, bb;
        xor16_pair(v, a, bb); v = max(a, bb);
        xor32_pair(v, a, bb); return max(a, bb);
      };
      auto max_cats = [&](unsigned v) { // maximum over the categories of a pattern: lanes that differ in the block bits
        if (CB == 4)
        {
          v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false)); // row_ror:4
          v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false)); // row_ror:8
        }
        else if (CB == 2) v = max(v, (unsigned)__shfl_xor((int)v, 4, 64));
        return v;
      };
      // The rescaling test (src/avx.c:498-506) only needs the high words' order (phyhip_kernels.hpp: kHiInvTwoToLarge).  Two
      // tiles: both maxima travel through the cross-lane scans in ONE register -- the upper 16 bits of each high word (sign,
      // exponent, four mantissa bits: 2^-256 = 0x2FF0 there, and v < 2^-256 <=> upper16(v) < 0x2FF0 for v >= 0), compared as
      // packed halves
      auto scan_max = [&](unsigned (&mx)[NT]) {
        if constexpr (NT == 2)
        {
          typedef unsigned short us2 __attribute__((ext_vector_type(2)));
          auto pmax = [](unsigned x, unsigned y) {
            us2 a, bb;
            __builtin_memcpy(&a, &x, 4); __builtin_memcpy(&bb, &y, 4);
            const us2 m = __builtin_elementwise_max(a, bb);
            unsigned  r;
            __builtin_memcpy(&r, &m, 4);
            return r;
          };
          unsigned v = (mx[0] >> 16) | (mx[1] & 0xFFFF0000u), a, bb;
          xor16_pair(v, a, bb); v = pmax(a, bb);
          xor32_pair(v, a, bb); v = pmax(a, bb);
          if (!cls)
          { // (a mixture class rescales alone)
            if (CB == 4)
            {
              v = pmax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false)); // row_ror:4
              v = pmax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false)); // row_ror:8
            }
            else if (CB == 2) v = pmax(v, (unsigned)__shfl_xor((int)v, 4, 64));
          }
          mx[0] = v << 16; mx[1] = v & 0xFFFF0000u; // (back as high words: the low bits play no part in the test)
        }
        else
        {
          mx[0] = max_states(mx[0]);
          if (!cls) mx[0] = max_cats(mx[0]);
        }
      };
      auto sum_states = [&](double v) { // (kk0 + kk1) + (kk2 + kk3), whichever lane asks: the four lanes' values by the half-exchange
        unsigned w2[2], o[4][2];      // instructions (rows of 16 lanes [r0 r1 r2 r3] -> [r0 r0 r2 r2], [r1 r1 r3 r3] -> r0, r2 and r1, r3 everywhere)
        __builtin_memcpy(w2, &v, 8);
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
          const auto pp = __builtin_amdgcn_permlane16_swap(w2[h], w2[h], false, false);
          const auto q0 = __builtin_amdgcn_permlane32_swap(pp[0], pp[0], false, false);
          const auto q1 = __builtin_amdgcn_permlane32_swap(pp[1], pp[1], false, false);
          o[0][h] = q0[0]; o[2][h] = q0[1]; o[1][h] = q1[0]; o[3][h] = q1[1];
        }
        double r4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_memcpy(&r4[i], o[i], 8);
        return (r4[0] + r4[1]) + (r4[2] + r4[3]);
      };
      auto tip_vec = [&](unsigned word, double (&x)[T]) {
        const unsigned m = (word ? word : 1u) >> kk;
#pragma unroll
        for (int t = 0; t < T; ++t) x[t] = ((m >> (4 * t)) & 1u) ? 1.0 : 0.0;
      };
      // wait until the ring holds item k; returns the item's first A table
      int ready_seen = 0;
      auto wait_item = [&](int k, int tslot) -> const double * { // tslot (INL): the item's first table slot
        while (!(ABL & 8) && ready_seen <= k)
        {
          ready_seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          if (ready_seen <= k) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        if constexpr (INL) return &ring[0][0][0] + (size_t)tslot * kAaMat;
        else return &ring[k % kAaRing][0][0];
      };
      auto release_item = [&](int k) {
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store(&s_done[w], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      // u[r] = sum over input states of P[c][4r + i][.] * x[.]: five row groups x five k-chunks, k ascending
      auto mfma_chunk = [&](const double *A, int t, const double xt, double (&u)[T]) {
        const v2d   *pr = reinterpret_cast<const v2d *>(A + t * kAaBlock) + lane;
        const v2d    a01 = pr[0], a23 = pr[64];
        const double a4  = A[t * kAaBlock + 256 + lane];
        u[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a01.x, xt, u[0], 0, 0, 0);
        u[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a01.y, xt, u[1], 0, 0, 0);
        u[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a23.x, xt, u[2], 0, 0, 0);
        u[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a23.y, xt, u[3], 0, 0, 0);
        u[4] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4, xt, u[4], 0, 0, 0);
      };
      // the same for the tiles of this wave that take the product (all of them: one read of the A operand feeds every tile)
      auto mfma_tiles = [&](const double *A, const double (&x)[NT][T], double (&u)[NT][T], const bool (&take)[NT]) {
        bool all = true;
#pragma unroll
        for (int i = 0; i < NT; ++i) all = all && take[i];
        if (NT > 1 && all)
        {
#pragma unroll
          for (int t = 0; t < T; ++t)
          {
            const v2d   *pr = reinterpret_cast<const v2d *>(A + t * kAaBlock) + lane;
            const v2d    a01 = pr[0], a23 = pr[64];
            const double a4  = A[t * kAaBlock + 256 + lane];
#pragma unroll
            for (int i = 0; i < NT; ++i)
            {
              u[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a01.x, x[i][t], u[i][0], 0, 0, 0);
              u[i][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a01.y, x[i][t], u[i][1], 0, 0, 0);
              u[i][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a23.x, x[i][t], u[i][2], 0, 0, 0);
              u[i][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a23.y, x[i][t], u[i][3], 0, 0, 0);
              u[i][4] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4, x[i][t], u[i][4], 0, 0, 0);
            }
          }
          return;
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
          if (take[i])
          {
#ifdef AA_DEEP
            // (experiment: every A value of the child requested before the first product)
            v2d    a01[T], a23[T];
            double a4[T];
#pragma unroll
            for (int t = 0; t < T; ++t)
            {
              const v2d *pr = reinterpret_cast<const v2d *>(A + t * kAaBlock) + lane;
              a01[t] = pr[0]; a23[t] = pr[64]; a4[t] = A[t * kAaBlock + 256 + lane];
            }
#pragma unroll
            for (int t = 0; t < T; ++t)
            {
              u[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a01[t].x, x[i][t], u[i][0], 0, 0, 0);
              u[i][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a01[t].y, x[i][t], u[i][1], 0, 0, 0);
              u[i][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a23[t].x, x[i][t], u[i][2], 0, 0, 0);
              u[i][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a23[t].y, x[i][t], u[i][3], 0, 0, 0);
              u[i][4] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[t], x[i][t], u[i][4], 0, 0, 0);
            }
#else
#pragma unroll
            for (int t = 0; t < T; ++t) mfma_chunk(A, t, x[i][t], u[i]);
#endif
          }
      };

      // column `state` of a table, rows 4r + kk of this lane's category: the A lane that holds P[c][4r + i][4t + k] is
      // 16k + 4b + i, so this (D) lane kk reads lane 16 (state & 3) + 4b + kk of k-chunk state >> 2
      auto tip_column = [&](const double *A, unsigned mask, double (&u)[T]) {
        const int st = __builtin_ctz(mask), la = 16 * (st & 3) + 4 * b + kk;
        const double *At = A + (st >> 2) * kAaBlock;
        const v2d    *pr = reinterpret_cast<const v2d *>(At) + la;
        const v2d     a01 = pr[0], a23 = pr[64];
        u[0] = a01.x; u[1] = a01.y; u[2] = a23.x; u[3] = a23.y; u[4] = At[256 + la];
      };
      auto one_state = [](unsigned m) { return __builtin_amdgcn_ballot_w64((m & (m - 1u)) != 0u) == 0; }; // every lane's mask is one state

      // results of the last two operations (this lane's D fragments), alternating: step k finds the result of k-1 in one set and
      // writes its own over the result of k-2 in the other (a result that stays virtual -- phyhip_host.hpp -- exists only here)
      double   FA[NT][T], FB[NT][T];
      unsigned scA[NT], scB[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i)
      {
#pragma unroll
        for (int t = 0; t < T; ++t) FA[i][t] = FB[i][t] = 0.0;
        scA[i] = scB[i] = 0;
      }

      // argument form: a side of the evaluation edge that no operation of this launch writes (TreeParams::e_prefetch) is requested
      // now, in front of the operations' own children -- one trip to memory less on the evaluation's chain
      double   epx[NT][T], epy[NT][T];
      unsigned eps[NT][2];
      auto eval_fetch = [&](int i, int idx, double (&v)[T], unsigned &sc) {
        const double *src = q.partials + (size_t)(idx - tips) * (ntiles * kAaBlock) + (size_t)tile[i] * kAaBlock;
#pragma unroll
        for (int t = 0; t < T; ++t) v[t] = src[aa_slot(t, lane)];
        sc = (unsigned)q.scales[(size_t)(idx - tips) * (cls ? C : 1) * q.Ppad + voff_s[i] / 4];
      };
      // ... and so is everything else the site likelihoods read from memory (stationary frequencies, pattern weight, invariant-site
      // state): four dependent trips on the evaluation's chain otherwise (2 us of a 7 us evaluation).  Same values, same arithmetic.
      double pre_pi[T], pre_wt = 0.0, pre_cw[C_];
      int    pre_iv = -1;
#pragma unroll
      for (int cc = 0; cc < C_; ++cc) pre_cw[cc] = 0.0;
#pragma unroll
      for (int t = 0; t < T; ++t) pre_pi[t] = 0.0;
      // (resident form with operations: requested inside the first step, behind its operands' wait -- the operands were requested
      // when the command was parsed, and a wave's loads complete in order: in front of that wait these would be waited for too)
      auto request_eval_inputs = [&]() {
      if constexpr (ARGS)
        if (q.edge_eval)
        {
          const double *pi_e = q.pi + ((cls && !idle) ? c * 20 : 0);
#pragma unroll
          for (int t = 0; t < T; ++t) pre_pi[t] = pi_e[4 * t + kk];
#pragma unroll
          for (int cc = 0; cc < C_; ++cc) pre_cw[cc] = q.cat_w[cc];
          if (!cls && pact[0])
          {
            if (kk == 0 && c == 0) pre_wt = q.wght[p0[0]];
            if (q.invar_model) pre_iv = q.invar[p0[0]];
          }
        }
      if constexpr (ARGS)
        if (q.edge_eval)
        {
#pragma unroll
          for (int i = 0; i < NT; ++i)
          {
            if ((q.e_prefetch & 1) && tact[i]) eval_fetch(i, q.e_parent, epx[i], eps[i][0]);
            if ((q.e_prefetch & 2) && tact[i]) eval_fetch(i, q.e_child, epy[i], eps[i][1]);
          }
        }
      };
      if (!RES || n_ops == 0) request_eval_inputs();

      if (n_ops > 0)
      {
        const int last = n_ops - 1;
        Raw       RA[NT], RB[NT];
        ExecRec   cur = XR(0);
        IssueRec  nx1 = IR((1 < last) ? 1 : last);
        if constexpr (RES)
        { // (requested when the command was parsed: see ext_a01)
          RA[0].a.p01 = ext_a01; RA[0].a.p23 = ext_a23; RA[0].a.p4 = ext_a4;
          RA[0].b.p01 = ext_b01; RA[0].b.p23 = ext_b23; RA[0].b.p4 = ext_b4;
          RA[0].xa = ext_xa; RA[0].xb = ext_xb;
        }
        else issue_children(IR(0), RA);
        if constexpr (D2) issue_children(nx1, RB);
        {
          // The loop body sees [children loads of k+1][4 result stores of k] in flight (per tile) when step k+1 starts.  As many
          // stores through a zero-sized descriptor (dropped by the hardware, but counted) give the loop entry the same shape, so
          // the compiler's merged s_waitcnt counts never make a step wait for the previous step's stores.
          const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, 0x00020000);
          const u32x4                  z4   = {0u, 0u, 0u, 0u};
          const u32x2                  z2   = {0u, 0u};
#pragma unroll
          for (int i = 0; i < NT; ++i)
          {
            __builtin_amdgcn_raw_buffer_store_b128(z4, none, 8192 * i, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(z4, none, 8192 * i + 1024, 0, 0); // distinct offsets: identical stores would be merged
            __builtin_amdgcn_raw_buffer_store_b64(z2, none, 8192 * i + 2048, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(0u, none, 8192 * i + 4096, 0, 0);
          }
        }

        auto step = [&](const int k, Raw (&R)[NT], Raw (&Rn)[NT], double (&Fout)[NT][T], unsigned (&scout)[NT], const double (&Fprev)[NT][T],
                        const unsigned (&scprev)[NT]) {
          const unsigned fl = cur.dst_data.x;
          double         x1[NT][T], x2[NT][T], o[NT][T];
          unsigned       s1[NT], s2[NT];
          PHY_STAMP(k, 0)
          // The loads of operation k+1 go out first (their registers were consumed by step k-1), then the records of
          // k+2 / k+1 are requested into the scalar registers the issue just freed: both have the whole step to arrive.
          // (D2: the records first; the loads of operation k + 2 follow below, into the registers this step's operands arrived in)
          if constexpr (!D2)
            if (!ARGS || k < last) issue_children(nx1, Rn);
          const IssueRec nx2 = IR((k + 2 < last) ? k + 2 : last);
          const ExecRec  nxe = XR((k + 1 < last) ? k + 1 : last);
          unsigned xa[NT], xb[NT], xt[NT]; // this step's auxiliary words (the register set is handed on below)
#pragma unroll
          for (int i = 0; i < NT; ++i) { xa[i] = R[i].xa; xb[i] = R[i].xb; xt[i] = INL ? R[i].xt : 0u; }
          // A tip child whose patterns all carry ONE state contributes a column of its matrix (the reference's Exex / Exin
          // kernels, src/avx.c:527-564): five values per lane straight from the ring, no product.  (Through the matrix
          // cores the result would be the same doubles -- the other 19 products are exact zeros -- at 25 MFMAs.)
          unsigned m1[NT], m2[NT];
          bool     hot1[NT], hot2[NT];
          const bool in1 = INL && (fl & kOpCh1), in2 = INL && (fl & kOpCh2); // (this child is computed below, from the ring)
#pragma unroll
          for (int i = 0; i < NT; ++i)
          {
            m1[i] = m2[i] = 0; hot1[i] = hot2[i] = false;
            if (fl & kOpTip1)
            {
              m1[i]   = xa[i] ? xa[i] : 1u; // (padding patterns carry no state: any column will do)
              hot1[i] = one_state(m1[i]);
            }
            if (fl & kOpTip2)
            {
              m2[i]   = xb[i] ? xb[i] : 1u;
              hot2[i] = one_state(m2[i]);
            }
            if (in1) s1[i] = 0;
            else if (fl & kOpTip1) { if (!hot1[i]) tip_vec(m1[i], x1[i]); s1[i] = 0; }
            else if (fl & kOpF11)
            {
#pragma unroll
              for (int t = 0; t < T; ++t) x1[i][t] = Fprev[i][t];
              s1[i] = scprev[i];
            }
            else if (fl & kOpF12)
            {
#pragma unroll
              for (int t = 0; t < T; ++t) x1[i][t] = Fout[i][t];
              s1[i] = scout[i];
            }
            else { unpack(R[i].a, x1[i]); s1[i] = xa[i]; }
            if (in2) s2[i] = 0;
            else if (fl & kOpTip2) { if (!hot2[i]) tip_vec(m2[i], x2[i]); s2[i] = 0; }
            else if (fl & kOpF21)
            {
#pragma unroll
              for (int t = 0; t < T; ++t) x2[i][t] = Fprev[i][t];
              s2[i] = scprev[i];
            }
            else if (fl & kOpF22)
            {
#pragma unroll
              for (int t = 0; t < T; ++t) x2[i][t] = Fout[i][t];
              s2[i] = scout[i];
            }
            else { unpack(R[i].b, x2[i]); s2[i] = xb[i]; }
          }
          if constexpr (D2) issue_children(nx2, R); // (clamped to the last operation near the end: harmless repeats)
          if constexpr (RES) if (k == 0) request_eval_inputs();
          PHY_STAMP(k, 1)
          // all-ones shortcut of the Inin kernel, per (pattern, category): src/avx.c:575-587 (never with a one-state tip)
          unsigned ones[NT];
          auto all_ones = [&](int i) {
            ones[i] = 1;
#pragma unroll
            for (int t = 0; t < T; ++t) ones[i] &= (unsigned)((x1[i][t] == 1.0) & (x2[i][t] == 1.0));
            if (!(ABL & 16)) ones[i] = and_states(ones[i]);
          };
#pragma unroll
          for (int i = 0; i < NT; ++i)
          {
            ones[i] = 0;
            if (!hot1[i] && !hot2[i] && !in1 && !in2) all_ones(i);
          }
          PHY_STAMP(k, 2)
          double u1[NT][T], u2[NT][T];
#pragma unroll
          for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int t = 0; t < T; ++t) u1[i][t] = u2[i][t] = 0.0;
          {
            const double *A = wait_item(k, (int)cur.dst_scale.x);
            PHY_STAMP(k, 3)
            if constexpr (RES) if (k == 0 && rs.stamps && (int)blockIdx.x == rs.stamp_wg && w == 0 && lane == 0) (void)__hip_atomic_fetch_add(&rs.stamps[8 + 5], wall_clock64() - s_t4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if constexpr (INL)
            {
              if (in1 || in2)
              { // the in-step child: the step of its defining operation (tips a, b through tables 2 and 3 of this item)
                // (one child after the other: its tip vector is dead once its products are issued; what the all-ones rule needs of
                // it is one bit per lane)
                bool     hota[NT], hotb[NT], tk[NT];
                unsigned onesab[NT];
                double   ua[NT][T], ub[NT][T];
                {
                  double va[NT][T];
#pragma unroll
                  for (int i = 0; i < NT; ++i)
                  {
                    const unsigned ma = (in1 ? xa[i] : xb[i]) ? (in1 ? xa[i] : xb[i]) : 1u;
                    hota[i] = one_state(ma); tk[i] = !hota[i]; onesab[i] = 1;
#pragma unroll
                    for (int t = 0; t < T; ++t) ua[i][t] = 0.0;
                    if (hota[i]) tip_column(A + 2 * kAaMat, ma, ua[i]);
                    else
                    {
                      tip_vec(ma, va[i]);
#pragma unroll
                      for (int t = 0; t < T; ++t) onesab[i] &= (unsigned)(va[i][t] == 1.0);
                    }
                  }
                  mfma_tiles(A + 2 * kAaMat, va, ua, tk);
                }
                {
                  double vb[NT][T];
#pragma unroll
                  for (int i = 0; i < NT; ++i)
                  {
                    const unsigned mb = xt[i] ? xt[i] : 1u;
                    hotb[i] = one_state(mb); tk[i] = !hotb[i];
#pragma unroll
                    for (int t = 0; t < T; ++t) ub[i][t] = 0.0;
                    if (hotb[i]) tip_column(A + 3 * kAaMat, mb, ub[i]);
                    else
                    {
                      tip_vec(mb, vb[i]);
#pragma unroll
                      for (int t = 0; t < T; ++t) onesab[i] &= (unsigned)(vb[i][t] == 1.0);
                    }
                  }
                  mfma_tiles(A + 3 * kAaMat, vb, ub, tk);
                }
                unsigned cm[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i)
                {
                  const unsigned cones = (!hota[i] && !hotb[i]) ? and_states(onesab[i]) : 0u;
                  double (&xc)[T] = in1 ? x1[i] : x2[i];
                  cm[i] = 0;
#pragma unroll
                  for (int t = 0; t < T; ++t)
                  {
                    xc[t] = cones ? 1.0 : ua[i][t] * ub[i][t];
                    cm[i] = max(cm[i], hi32(xc[t]));
                  }
                  if (C_ == 3 && idle) cm[i] = 0;
                }
                scan_max(cm);
#pragma unroll
                for (int i = 0; i < NT; ++i)
                {
                  double (&xc)[T] = in1 ? x1[i] : x2[i];
                  unsigned csc = 0;
                  if (cm[i] < kHiInvTwoToLarge && q.apply_scaling)
                  {
#pragma unroll
                    for (int t = 0; t < T; ++t) xc[t] *= kTwoToLarge;
                    csc = kLarge;
                  }
                  if (in1) s1[i] = csc; else s2[i] = csc;
                  if (!hot1[i] && !hot2[i]) all_ones(i); // (the operation's own all-ones test, now that both children are there)
                }
              }
            }
            if (ABL & 1)
            {
#pragma unroll
              for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int t = 0; t < T; ++t) { u1[i][t] = x1[i][t]; u2[i][t] = x2[i][t]; }
            }
            else
            {
              bool tk[NT];
#pragma unroll
              for (int i = 0; i < NT; ++i)
              {
                tk[i] = !hot1[i];
                if (hot1[i]) tip_column(A, m1[i], u1[i]);
              }
              mfma_tiles(A, x1, u1, tk);
#pragma unroll
              for (int i = 0; i < NT; ++i)
              {
                tk[i] = !hot2[i];
                if (hot2[i]) tip_column(A + kAaMat, m2[i], u2[i]);
              }
              mfma_tiles(A + kAaMat, x2, u2, tk);
            }
            release_item(k);
          }
          PHY_STAMP(k, 4)
          unsigned mxh[NT];
#pragma unroll
          for (int i = 0; i < NT; ++i)
          {
            mxh[i] = 0;
#pragma unroll
            for (int t = 0; t < T; ++t)
            {
              o[i][t] = ones[i] ? 1.0 : u1[i][t] * u2[i][t];
              mxh[i]  = max(mxh[i], hi32(o[i][t]));
            }
            if (C_ == 3 && idle) mxh[i] = 0;
          }
          if (!(ABL & 2)) scan_max(mxh);
          PHY_STAMP(k, 5)
          const __amdgpu_buffer_rsrc_t dr = rsrc(cur.dst_data), gr = rsrc(cur.dst_scale);
#pragma unroll
          for (int i = 0; i < NT; ++i)
          {
            unsigned sc = s1[i] + s2[i]; // src/avx.c:462-464
            if (mxh[i] < kHiInvTwoToLarge && q.apply_scaling)
            { // src/avx.c:504-510
#pragma unroll
              for (int t = 0; t < T; ++t) o[i][t] *= kTwoToLarge;
              sc += kLarge;
            }
            Frag wv;
            pack(o[i], wv);
            __builtin_amdgcn_raw_buffer_store_b128(wv.p01, dr, voff_d16[i], 0, kStAux);
            __builtin_amdgcn_raw_buffer_store_b128(wv.p23, dr, voff_d16[i] + 1024, 0, kStAux);
            __builtin_amdgcn_raw_buffer_store_b64(wv.p4, dr, voff_d8[i], 0, kStAux);
            __builtin_amdgcn_raw_buffer_store_b32(sc, gr, voff_sst[i], 0, kStAuxW);
#pragma unroll
            for (int t = 0; t < T; ++t) Fout[i][t] = o[i][t];
            scout[i] = sc;
          }
          PHY_STAMP(k, 6)
          cur   = nxe;
          nx1   = nx2;
        };
        if constexpr (ARGS)
        {
          step(0, RA, RB, FA, scA, FB, scB);
          if (n_ops > 1) step(1, RB, RA, FB, scB, FA, scA);
          else
          { // the evaluation below expects the last result in the second set
#pragma unroll
            for (int i = 0; i < NT; ++i)
            {
#pragma unroll
              for (int t = 0; t < T; ++t) FB[i][t] = FA[i][t];
              scB[i] = scA[i];
            }
          }
        }
        else
        { // (the list form is padded to an even length: the last result ends up in the second set)
          for (int k = 0; k < n_ops; k += 2)
          {
            step(k, RA, RB, FA, scA, FB, scB);
            step(k + 1, RB, RA, FB, scB, FA, scA);
          }
        }
      }

      if (q.edge_eval)
      {
        if constexpr (RES) if (rs.stamps && (int)blockIdx.x == rs.stamp_wg && w == 0 && lane == 0) (void)__hip_atomic_fetch_add(&rs.stamps[8 + 1], wall_clock64() - s_t4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- K2: site likelihood at the evaluation edge (src/lk.c:608-645, 767-861) -------------------------------
        double   x[NT][T], y[NT][T], u[NT][T];
        unsigned sl[NT], sr[NT];
        bool     tk[NT];
        auto side = [&](int i, int idx, double (&v)[T], unsigned &sc, const int which) {
          if (ARGS && tact[i] && ((q.e_prefetch >> which) & 1))
          { // (requested at the start)
#pragma unroll
            for (int t = 0; t < T; ++t) v[t] = which ? epy[i][t] : epx[i][t];
            sc = eps[i][which];
          }
          else if (!tact[i])
          {
#pragma unroll
            for (int t = 0; t < T; ++t) v[t] = 0.0;
            sc = 0;
          }
          else if (idx < tips)
          {
            tip_vec(tip_masks[(size_t)idx * q.Ppad + p0[i]], v);
            sc = 0;
          }
          else if (idx == q.last_dest)
          {
#pragma unroll
            for (int t = 0; t < T; ++t) v[t] = FB[i][t];
            sc = scB[i];
          }
          else eval_fetch(i, idx, v, sc);
        };
#pragma unroll
        for (int i = 0; i < NT; ++i)
        {
          side(i, q.e_parent, x[i], sl[i], 0);
          side(i, q.e_child, y[i], sr[i], 1);
          tk[i] = true;
#pragma unroll
          for (int t = 0; t < T; ++t) u[i][t] = 0.0;
        }
        {
          const double *A = wait_item(n_ops, q.aa_e_slot);
          mfma_tiles(A, x, u, tk); // rows: right-side state
          release_item(n_ops);
        }
        if constexpr (RES) if (rs.stamps && (int)blockIdx.x == rs.stamp_wg && w == 0 && lane == 0) (void)__hip_atomic_fetch_add(&rs.stamps[8 + 2], wall_clock64() - s_t4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double *pi_c = q.pi + ((cls && !idle) ? c * 20 : 0);
#pragma unroll
        for (int i = 0; i < NT; ++i)
        {
          double contrib = 0.0;
          double part = 0.0;
#pragma unroll
          for (int t = 0; t < T; ++t) part += u[i][t] * (y[i][t] * (ARGS ? pre_pi[t] : pi_c[4 * t + kk]));
          // (argument form: pi[invariant state] out of the frequencies requested up front -- lane 16 (iv & 3) + ... holds pi[iv] in
          // chunk iv >> 2)
          double inv_pre = 0.0;
          if constexpr (ARGS)
            if (!cls && q.invar_model)
            {
              const int src = (lane & 15) | ((pre_iv < 0 ? kk : (pre_iv & 3)) << 4);
#pragma unroll
              for (int t = 0; t < T; ++t)
              {
                const double v = __shfl(pre_pi[t], src, 64);
                if ((pre_iv >> 2) == t) inv_pre = v;
              }
            }
          const double lkc = sum_states(part);
          if (pact[i] && kk == 0 && q.site_cat) gst(&q.site_cat[(size_t)p0[i] * C + c], lkc);
          if (cls)
          { // per class: its likelihood (above) and its scale exponent; the mixture is combined by class_combine_kernel
            if (pact[i] && kk == 0) gst(&q.fact[(size_t)c * q.P + p0[i]], q.apply_scaling ? (int)(sl[i] + sr[i]) : 0);
          }
          else
          {
            // the categories of this lane's pattern, in category order (src/lk.c:816-818)
            double site = 0.0;
#pragma unroll
            for (int cc = 0; cc < C_; ++cc) site += __shfl(lkc, (lane & ~(3 << 2)) | (((b / CB) * CB + cc) << 2), 64) * (ARGS ? pre_cw[cc] : q.cat_w[cc]);
            if (pact[i] && kk == 0 && c == 0)
            {
              const double wt = ARGS ? pre_wt : q.wght[p0[i]];
              int          f  = q.apply_scaling ? (int)(sl[i] + sr[i]) : 0;
              if (wt > kSmall)
              {
                if (q.invar_model)
                { // src/lk.c:820-842, 1226-1273
                  const int iv  = ARGS ? pre_iv : q.invar[p0[i]];
                  double    inv = 0.0;
                  bool      issue_ = false;
                  if (iv >= 0)
                  {
                    inv = ARGS ? inv_pre : q.pi[iv];
                    if (q.apply_scaling)
                    {
                      int e = f;
                      do
                      {
                        const int piece = e < 63 ? e : 63;
                        inv *= (double)(1ull << piece);
                        e -= piece;
                      } while (e != 0);
                    }
                    issue_ = isinf(inv);
                  }
                  if (issue_) { f = 0; site = q.pi[iv] * q.pinvar; }
                  else site = site * (1. - q.pinvar) + inv * q.pinvar;
                }
                if (site < kSmall) { site = kSmall; raise_warn(q); }
                const double lsl = log(site) - kLog2 * (double)f;
                if (q.site_lnl) gst(&q.site_lnl[p0[i]], lsl);
                if (q.site_lk) gst(&q.site_lk[p0[i]], dev_exp(lsl));
                contrib = wt * lsl;
              }
              gst(&q.fact[p0[i]], f);
            }
            // this TILE's share: fixed shuffle tree -> deterministic, and the same double whichever wave shape computed it
            // (the tree contrib += shfl_down(contrib, 32 / 16 / 8 / 4 / 2 / 1) as lane 0 sees it, without the steps that can only add
            // zeros -- contributions sit in the lanes with kk = 0, c = 0: lanes 0-15, of them block 0 (four categories), blocks 0 and
            // 2 (two), all four (one) -- and with the last two steps, inside a quad, as DPP moves instead of trips through LDS)
            if (CB <= 2) contrib += __shfl_down(contrib, 8, 64);
            if (CB == 1) contrib += __shfl_down(contrib, 4, 64);
            auto quad = [](const double v, const auto ctrl) { // lane i of a quad <- lane perm[i] of it
              unsigned w2[2];
              __builtin_memcpy(w2, &v, 8);
              w2[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w2[0], decltype(ctrl)::value, 0xF, 0xF, false);
              w2[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w2[1], decltype(ctrl)::value, 0xF, 0xF, false);
              double r;
              __builtin_memcpy(&r, w2, 8);
              return r;
            };
            contrib += quad(contrib, std::integral_constant<int, 0xEE>()); // quad_perm [2, 3, 2, 3]: lanes 0, 1 <- lanes 2, 3
            contrib += quad(contrib, std::integral_constant<int, 0xF5>()); // quad_perm [1, 1, 3, 3]: lane 0 <- lane 1
            if (lane == 0 && w * NT + i < 16) s_wsum[w * NT + i] = contrib;
          }
        }
        if constexpr (RES) if (rs.stamps && (int)blockIdx.x == rs.stamp_wg && w == 0 && lane == 0) (void)__hip_atomic_fetch_add(&rs.stamps[8 + 3], wall_clock64() - s_t4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (q.fence_post && !RES) __threadfence(); // every wave's stores are in memory before the workgroup's sum is posted (resident form: below)
        if constexpr (RES) if (rs.stamps && (int)blockIdx.x == rs.stamp_wg && w == 0 && lane == 0) (void)__hip_atomic_fetch_add(&rs.stamps[8 + 4], wall_clock64() - s_t4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }

  if (DBG && dbg)
  {
    __syncthreads();
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
      for (int i = 0; i < 64 * 8; ++i) dbg[i] = stamps[i];
  }
#undef PHY_STAMP
  if (!q.edge_eval || q.class_axis) return;
  // (resident form: what this wave stored -- results, its share of the rebuilt matrices -- is in memory before the records go out:
  // written through, see kStAux above, so acknowledged = there)
  if constexpr (RES && !(AA_RES_ABL & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wave == 0)
  {
    if constexpr (RES)
    { // one record per wave-tile: the launched form of such an instance has one tile per workgroup, the host adds the same terms
      if (lane < tpw && (size_t)blockIdx.x * tpw + lane < ntiles)
        post_host_block(q.host_blocks + (size_t)blockIdx.x * tpw + lane, s_wsum[lane], q.host_tag);
    }
    else
    { // the consumers' shares in wave order
      double tot = 0.0;
      for (int i = 0; i < tpw; ++i)
        if ((size_t)blockIdx.x * tpw + i < ntiles) tot += s_wsum[i];
      publish_block_sum(q, tot, lane);
    }
  }
  }; // (run)

  if constexpr (!RES) run(q_in);
  else
  {
    __shared__ unsigned long long sh_raw[64];
    __shared__ __attribute__((aligned(16))) double sh_expt[4][4 * 20]; // [rebuilt matrix][category][eigenvalue]
    __shared__ __attribute__((aligned(16))) double sh_U[400], sh_V[400], sh_R[20], sh_rates[4];
    __shared__ int sh_act;
    __shared__ unsigned long long sh_exp[256]; // the exp table (dev_exp), copied once per launch
    exp_tab_to_lds(sh_exp, (int)threadIdx.x, (int)blockDim.x);
    const int          lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = (int)(blockDim.x >> 6);
    unsigned long long last = rs.ctl.start_seq, t_last = wall_clock64(), model_seen = ~0ull;
    bool               mail_open = false;
    constexpr unsigned kMatB = kAaMat * 8;
    for (;;)
    {
      if (wave == 0)
      {
        const int a = resident_poll_wave(rs.ctl, last, t_last, mail_open, sh_raw, 1, lane, blockIdx.x == 0);
        if (lane == 0) sh_act = a;
      }
      __syncthreads();
      const int act = __builtin_amdgcn_readfirstlane(sh_act);
      if (act == 2) return;
      if (act == 0)
      {
        __syncthreads(); // (sh_act / sh_raw are rewritten by the next poll)
        continue;
      }
      // the command's words: lane k of every wave reads word k (ONE trip to LDS for the whole record -- a read per word, each
      // waited for, cost a microsecond per command), and a word is taken out of lane k's register: the same for all lanes and known
      // to be (descriptors and loop bounds belong in scalar registers)
      static_assert(kResidentAaWords <= 64, "one command word per lane");
      const unsigned long long my_word = sh_raw[resident_slot(lane < kResidentAaWords ? lane : 0)];
      auto word = [&](int k) {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)my_word, k), hi = __builtin_amdgcn_readlane((unsigned)(my_word >> 32), k);
        return ((unsigned long long)hi << 32) | lo;
      };
      auto desc = [&](int k) {
        Desc               d;
        const unsigned long long b = word(k + 1);
        d.base = word(k); d.bytes = (unsigned)b; d.x = (unsigned)(b >> 32);
        return d;
      };
      unsigned long long tk[6];
      tk[0] = rs.stamps ? wall_clock64() : 0ull;
      const unsigned long long fl = word(1), ed = word(2), pm = word(3), epoch = word(34);
      TreeParams               q = q_in;
      q.host_tag = word(0);
      q.n_fresh = (int)((fl >> 4) & 15); q.e_prefetch = (int)((fl >> 8) & 3);
      q.edge_eval = 1; q.e_parent = (int)(unsigned)ed; q.e_child = (int)(unsigned)(ed >> 32);
      q.e_pm = (int)(unsigned)pm; q.last_dest = (int)(unsigned)(pm >> 32);
      q.n_real_ops = (int)(fl & 3);
      // what kernels on the stream wrote since the last command (the host says whether any did) is re-read from memory
      if (fl & 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      tk[1] = rs.stamps ? wall_clock64() : 0ull;
      if (wave != 0 && q.n_real_ops > 0)
      { // the consumers' first loads (issue_children's, for operation 0 of the command): see ext_a01
        constexpr int CBr = C_ == 1 ? 1 : (C_ == 2 ? 2 : 4), NPWr = 16 / CBr;
        const int       nwr = (int)(blockDim.x >> 6) - 1;
        const long long tl = (long long)blockIdx.x * nwr + (wave - 1);
        const bool      ta = (wave - 1 < nwr) && tl < (long long)(q_in.Ppad / NPWr);
        const unsigned  nowhere = 0x7fff0000u, blk = (unsigned)((size_t)tl * kAaBlock * 8);
        const unsigned  v16 = ta ? blk + lane * 16 : nowhere, v8 = ta ? blk + 2048 + lane * 8 : nowhere;
        const unsigned  vp = ta ? (unsigned)(tl * NPWr + ((lane >> 2) & 3) / CBr * 4 + (lane & 3)) * 4u : nowhere; // (scale word / tip mask of the lane's pattern)
        const Desc d1 = desc(10), d2 = desc(12), g1 = desc(14), g2 = desc(16);
        auto rs4 = [](const Desc &d) { return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(d.base), 0, (int)d.bytes, 0x00020000); };
        const __amdgpu_buffer_rsrc_t r1 = rs4(d1), r2 = rs4(d2), r3 = rs4(g1), r4 = rs4(g2);
        ext_a01 = __builtin_amdgcn_raw_buffer_load_b128(r1, v16, 0, PHYHIP_LOAD_AUX);
        ext_a23 = __builtin_amdgcn_raw_buffer_load_b128(r1, v16 + 1024, 0, PHYHIP_LOAD_AUX);
        ext_a4  = __builtin_amdgcn_raw_buffer_load_b64(r1, v8, 0, PHYHIP_LOAD_AUX);
        ext_b01 = __builtin_amdgcn_raw_buffer_load_b128(r2, v16, 0, PHYHIP_LOAD_AUX);
        ext_b23 = __builtin_amdgcn_raw_buffer_load_b128(r2, v16 + 1024, 0, PHYHIP_LOAD_AUX);
        ext_b4  = __builtin_amdgcn_raw_buffer_load_b64(r2, v8, 0, PHYHIP_LOAD_AUX);
        ext_xa  = __builtin_amdgcn_raw_buffer_load_b32(r3, vp, 0, 0);
        ext_xb  = __builtin_amdgcn_raw_buffer_load_b32(r4, vp, 0, 0);
      }
      // ---- the queued matrices, by this workgroup, into the ring slots its consumers will read them from ----------------------
      const int nf = q.n_fresh;
      if (nf > 0)
      {
        if (epoch != model_seen)
        { // (the eigen system changes with the model only: staged once per epoch)
          for (int t = threadIdx.x; t < 400; t += blockDim.x) { sh_U[t] = rs.evec[t]; sh_V[t] = rs.ivec[t]; }
          if (threadIdx.x < 20) sh_R[threadIdx.x] = rs.eval[threadIdx.x];
          if (threadIdx.x < C_) sh_rates[threadIdx.x] = rs.rates[threadIdx.x];
          model_seen = epoch;
          __syncthreads();
        }
        if (!(AA_RES_ABL & 4))
        { // (pmat20_exponentials' arithmetic, the matrices side by side: thread t takes (matrix t / 80, category, eigenvalue))
          const int t = (int)threadIdx.x, f = t / (4 * 20), u = t % (4 * 20);
          if (f < nf && u < C_ * 20)
          {
            double len;
            const unsigned long long lb = sh_raw[resident_slot(6 + f)];
            __builtin_memcpy(&len, &lb, 8);
            pmat20_exponentials(sh_expt[f], len, C_, false, sh_R, sh_rates, q.br_len_mult, q.l_min, q.l_max, u, 4 * 20, sh_exp);
          }
        }
      }
      // where each rebuilt matrix is read in this command: the five places a command has -- table 0 / 1 of operation 0, of
      // operation 1, table 0 of the evaluation edge's item -- or, read nowhere (the host sends at most two such), a table of the
      // ring's spare item
      const int n_ops_c = q.n_real_ops;
      auto fidx = [&](int f) { return (int)(unsigned)(word(4 + f / 2) >> (32 * (f & 1))); };
      // (the matrix offsets of the operations' children: the spare words of their data descriptors -- the descriptors themselves are
      // taken out of the command behind the rebuild, where they are needed)
      const unsigned used[5] = {n_ops_c > 0 ? (unsigned)(word(11) >> 32) : ~0u, n_ops_c > 0 ? (unsigned)(word(13) >> 32) : ~0u,
                                n_ops_c > 1 ? (unsigned)(word(23) >> 32) : ~0u, n_ops_c > 1 ? (unsigned)(word(25) >> 32) : ~0u, (unsigned)q.e_pm * kMatB};
      double *const eslot = n_ops_c == 0 ? &ring[0][0][0] : (n_ops_c == 1 ? &ring[1][0][0] : &ring[2][0][0]);
      unsigned built = 0;
      if (nf > 0)
      {
        __syncthreads(); // (exponentials and eigen system staged)
        tk[2] = rs.stamps ? wall_clock64() : 0ull;
        // where each rebuilt matrix goes: table numbers of the ring (table w of item j = 2 j + w; the evaluation edge is item n_ops,
        // table 0; the spare item's two tables take what nothing reads in this command); bits of `built` likewise
        int      fi[4], fta[4], ftb[4];
        unsigned fmore[4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
        {
          fi[f] = 0; fta[f] = ftb[f] = 2 * (kAaRing - 1) + (f & 1); fmore[f] = 0;
          if (f < nf)
          {
            fi[f] = fidx(f);
            const unsigned offb = (unsigned)fi[f] * kMatB;
            unsigned       m = 0; // (bit = table number)
#pragma unroll
            for (int y = 0; y < 4; ++y) m |= (used[y] == offb) ? (1u << y) : 0u;
            if (used[4] == offb) m |= 1u << (2 * n_ops_c);
            built |= m;
            if (m)
            {
              fta[f] = __builtin_ctz(m); m &= m - 1;
              ftb[f] = m ? __builtin_ctz(m) : fta[f];
              fmore[f] = m ? (m & (m - 1)) : 0u;
            }
          }
        }
        auto pick = [](const int f, const auto (&v)[4]) { return f == 0 ? v[0] : (f == 1 ? v[1] : (f == 2 ? v[2] : v[3])); };
        // one wave per (matrix, row group) unit, the units dealt out over the workgroup's waves; a wave with two units (three
        // matrices x five row groups on eight waves) works through them side by side
        if (!(AA_RES_ABL & 2))
          for (int u0 = wave; u0 < nf * kAaT; u0 += 2 * nwaves)
          {
            const int u1 = u0 + nwaves, f0 = u0 / kAaT, f1 = u1 < nf * kAaT ? u1 / kAaT : f0;
            const int           rr[2] = {u0 % kAaT, u1 < nf * kAaT ? u1 % kAaT : -1};
            const double *const ex[2] = {sh_expt[f0], sh_expt[f1]};
            // (the natural-layout copy of matrix f in the global table: by workgroup (f + 1) mod the grid -- workgroup 0's loader wave
            // writes the A-operand copies)
            double *const       oo[2] = {(int)blockIdx.x == (f0 + 1) % (int)gridDim.x ? rs.pmats_rw + (size_t)pick(f0, fi) * C_ * 400 : nullptr,
                                         (int)blockIdx.x == (f1 + 1) % (int)gridDim.x ? rs.pmats_rw + (size_t)pick(f1, fi) * C_ * 400 : nullptr};
            const int           ta[2] = {pick(f0, fta), pick(f1, fta)}, tb[2] = {pick(f0, ftb), pick(f1, ftb)};
            const unsigned      mo[2] = {pick(f0, fmore), pick(f1, fmore)};
            aa_build_units<C_, 2>(ex, sh_U, sh_V, rr, lane, oo, &ring[0][0][0], ta, tb, mo, (rs.stamps && (int)blockIdx.x == rs.stamp_wg && wave == 1) ? rs.stamps : nullptr);
          }
        __syncthreads();
        tk[3] = rs.stamps ? wall_clock64() : 0ull;
      }
      if (threadIdx.x == 0)
      {
        s_built = built;
        s_spare[0] = s_spare[1] = -1;
        for (int f = 0; f < nf; ++f)
        {
          const unsigned offb = (unsigned)fidx(f) * kMatB;
          bool           rd = false;
#pragma unroll
          for (int y = 0; y < 5; ++y) rd = rd || used[y] == offb;
          if (!rd) s_spare[f & 1] = fidx(f);
        }
      }
#pragma unroll
      for (int o = 0; o < 2; ++o)
      {
        q.arg_ir[o].c1_data = desc(10 + o * 12); q.arg_ir[o].c2_data = desc(12 + o * 12);
        q.arg_ir[o].c1_scale = desc(14 + o * 12); q.arg_ir[o].c2_scale = desc(16 + o * 12);
        q.arg_xr[o].dst_data = desc(18 + o * 12); q.arg_xr[o].dst_scale = desc(20 + o * 12);
      }
      tk[4] = rs.stamps ? wall_clock64() : 0ull;
      if (rs.stamps && threadIdx.x == 0) s_t4 = tk[4];
      run(q); // (starts with a workgroup barrier: s_built and the tables are in place for everybody)
      __syncthreads();
      if (rs.stamps && (int)blockIdx.x == rs.stamp_wg && threadIdx.x == 0)
      {
        tk[5] = wall_clock64();
        if (nf == 0) tk[2] = tk[3] = tk[1];
#pragma unroll
        for (int i = 0; i < 5; ++i) (void)__hip_atomic_fetch_add(&rs.stamps[i], tk[i + 1] - tk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_fetch_add(&rs.stamps[7], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      last = last + 1; t_last = wall_clock64();
    }
  }
}

// (defined in phyhip_big.hip, the translation unit compiled for kernels that are one loop around launched kernels' bodies)
int launch_resident_aa(int C, int workgroups, int consumers, hipStream_t stream, const TreeParams &sq, const double *afrag, int n_frag_mats,
                       const uint32_t *tip_masks, const AaResident &rs);

} // namespace phyhip
