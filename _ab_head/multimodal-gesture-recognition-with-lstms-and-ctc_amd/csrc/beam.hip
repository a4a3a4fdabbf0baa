// K9: CTC prefix beam search (K.ctc_decode(greedy=False, beam_width=W, top_paths=1) semantics; BASELINE.json
// config 5 - not present in the reference).  One wave per sequence, fp64 scores.
//
// Per time step (strictly sequential over t, parallel over the <= W*(C+1) candidates across the 64 lanes):
//   1. log y(c) = log(P+eps) - log(sum_c (P+eps))                               (lanes over classes)
//   2. every live beam r yields a "stay" candidate (blank, or repeat of its last label)        idx r*(C+1)
//      and an "extend by c" candidate for every non-blank c                                      idx r*(C+1)+1+c
//      an extension that reproduces the prefix of another live beam r2 (its trie parent is beam r and its last
//      label is c) is merged into r2's stay candidate instead (stay term first, then the extension)
//   3. the W best candidates by lse(p_blank, p_nonblank) - ties to the smaller idx, -inf dropped - are picked by W
//      rounds of a wave-wide (score, idx) arg-max with lane shuffles; extensions take their trie node (parent, label)
//      from a per-sequence hash table, so that a prefix which fell out of the beam and is found again keeps its node
//      id: "same node" is then exactly "same prefix", which the merge rule of step 2 relies on (a prefix p can die
//      while p+c lives; when p comes back, p+c must still be recognised as its child).
// The winner's prefix is read back through the parent links.  Scores are fp64 because fp32 scores near -5000 have
// an ulp of 5e-4 and the beam cut would then depend on libm rounding; the CPU oracle uses the same formulas.
#include "common.h"

namespace {

constexpr int MAXW = 32;   // beam width limit
constexpr int MAXC = 64;   // classes limit
constexpr int KMAX = 34;   // candidates per lane at the limits: ceil(MAXW*(MAXC+1)/64); the kernel is instantiated for 4 / 12 / 34
constexpr double kNegInfD = -__builtin_huge_val();

__device__ __forceinline__ double lse64(double a, double b) {
  if (a == kNegInfD) return b;
  if (b == kNegInfD) return a;
  double m = a > b ? a : b;
  return m + log1p(exp(-fabs(a - b)));
}

__device__ __forceinline__ double shfl_xor_d(double v, int o) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, o);
  hi = __shfl_xor(hi, o);
  return __hiloint2double(hi, lo);
}

// KM = candidate scores a lane keeps in registers (64 * KM >= W * (C + 1)).  One size for every shape (34) used 256 VGPRs and
// spilled 92 bytes of them to scratch memory - also for the reference's own shape (beam 10, 22 classes: 230 candidates, 4 per lane).
template <int KM>
__global__ __launch_bounds__(64) void k_beam(const float* __restrict__ P, const int32_t* __restrict__ input_len, int B, int T,
                                             int C, int skip, int blank, int W, float eps, int merge_repeated,
                                             int32_t* __restrict__ out, int32_t* __restrict__ out_len,
                                             double* __restrict__ logp, int32_t* __restrict__ node_parent,
                                             int32_t* __restrict__ node_label, int nodes_per_seq,
                                             unsigned long long* __restrict__ table, int table_bits) {
  __shared__ double s_logy[MAXC];
  __shared__ double s_pb[MAXW], s_pnb[MAXW], s_tot[MAXW];
  __shared__ double s_npb[MAXW], s_npnb[MAXW];   // stay candidates (after merging)
  __shared__ int s_node[MAXW], s_pnode[MAXW], s_last[MAXW], s_len[MAXW];
  __shared__ unsigned long long s_mmask[MAXW];   // classes whose extension of beam r was merged into another beam
  __shared__ double s_selb[MAXW], s_selnb[MAXW];
  __shared__ int s_selnode[MAXW], s_selpnode[MAXW], s_sellast[MAXW], s_sellen[MAXW];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int To = T - skip;
  int Tp = input_len[b];
  Tp = Tp < 0 ? 0 : (Tp > To ? To : Tp);
  int32_t* par = node_parent + (size_t)b * nodes_per_seq;
  int32_t* lab = node_label + (size_t)b * nodes_per_seq;
  // (parent, label) -> node: open addressing, one 64-bit word per entry = (key + 1) << 32 | node, 0 = empty
  unsigned long long* tab = table + ((size_t)b << table_bits);
  const unsigned tmask = (1u << table_bits) - 1u;
  for (unsigned i = lane; i <= tmask; i += 64) tab[i] = 0ull;
  __threadfence();
  int nb = 1;          // live beams
  int nnodes = 1;      // node 0 = empty prefix
  if (lane == 0) {
    par[0] = -1;
    lab[0] = -1;
    s_pb[0] = 0.0;
    s_pnb[0] = kNegInfD;
    s_node[0] = 0;
    s_pnode[0] = -1;
    s_last[0] = -1;
    s_len[0] = 0;
  }
  __syncthreads();
  const int CP1 = C + 1;
  for (int t = 0; t < Tp; ++t) {
    // ---- 1. frame log-probabilities
    const float* row = P + ((size_t)b * T + skip + t) * C;
    double u = (lane < C) ? (double)row[lane] + (double)eps : 0.0;
    double s = u;
    for (int o = 32; o > 0; o >>= 1) s += shfl_xor_d(s, o);
    if (lane < C) s_logy[lane] = log(u) - log(s);
    __syncthreads();
    // ---- 2a. stay candidates
    if (lane < nb) {
      double pb = s_pb[lane], pnb = s_pnb[lane];
      double tot = lse64(pb, pnb);
      s_tot[lane] = tot;
      s_npb[lane] = tot + s_logy[blank];
      s_npnb[lane] = (s_len[lane] > 0) ? pnb + s_logy[s_last[lane]] : kNegInfD;
      s_mmask[lane] = 0ull;
    }
    __syncthreads();
    // ---- 2b. merge extensions that land on a live beam (lane = r2; at most one (r, c) per r2)
    if (lane < nb && s_len[lane] > 0) {
      int pnode = s_pnode[lane];
      int c = s_last[lane];
      for (int r = 0; r < nb; ++r) {
        if (s_node[r] == pnode) {
          double val = ((s_len[r] > 0 && c == s_last[r]) ? s_pb[r] : s_tot[r]) + s_logy[c];
          s_npnb[lane] = lse64(s_npnb[lane], val);
          atomicOr(&s_mmask[r], 1ull << c);
          break;
        }
      }
    }
    __syncthreads();
    // ---- 3. candidate scores (this lane's slice) and W rounds of arg-max
    double cs[KM];
    const int ncand = nb * CP1;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      int idx = lane + 64 * k;
      double sc = kNegInfD;
      if (idx < ncand) {
        int r = idx / CP1, slot = idx - r * CP1;
        if (slot == 0) {
          sc = lse64(s_npb[r], s_npnb[r]);
        } else {
          int c = slot - 1;
          if (c != blank && !((s_mmask[r] >> c) & 1ull))
            sc = ((s_len[r] > 0 && c == s_last[r]) ? s_pb[r] : s_tot[r]) + s_logy[c];
        }
      }
      cs[k] = sc;
    }
    int nsel = 0;
    const int kused = (ncand + 63) / 64;
    for (int w = 0; w < W; ++w) {
      double best = kNegInfD;
      int bidx = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < KM; ++k) {
        if (k < kused) {
          int idx = lane + 64 * k;
          if (cs[k] > best || (cs[k] == best && cs[k] != kNegInfD && idx < bidx)) {
            best = cs[k];
            bidx = idx;
          }
        }
      }
      for (int o = 32; o > 0; o >>= 1) {
        double ob = shfl_xor_d(best, o);
        int oi = __shfl_xor(bidx, o);
        if (ob > best || (ob == best && oi < bidx)) {
          best = ob;
          bidx = oi;
        }
      }
      if (best == kNegInfD) break;  // wave-uniform
      // the owning lane retires the candidate
#pragma unroll
      for (int k = 0; k < KM; ++k)
        if (lane + 64 * k == bidx) cs[k] = kNegInfD;
      if (lane == 0) {
        int r = bidx / CP1, slot = bidx - r * CP1;
        if (slot == 0) {
          s_selb[nsel] = s_npb[r];
          s_selnb[nsel] = s_npnb[r];
          s_selnode[nsel] = s_node[r];
          s_selpnode[nsel] = s_pnode[r];
          s_sellast[nsel] = s_last[r];
          s_sellen[nsel] = s_len[r];
        } else {
          int c = slot - 1;
          s_selb[nsel] = kNegInfD;
          s_selnb[nsel] = ((s_len[r] > 0 && c == s_last[r]) ? s_pb[r] : s_tot[r]) + s_logy[c];
          s_selnode[nsel] = -1;  // resolved below, all selections in parallel
          s_selpnode[nsel] = s_node[r];
          s_sellast[nsel] = c;
          s_sellen[nsel] = s_len[r] + 1;
        }
      }
      ++nsel;
    }
    __syncthreads();
    if (lane < nsel) {
      int node = s_selnode[lane];
      if (node < 0) {
        // find-or-insert (parent, label); the selected extensions are distinct prefixes, hence distinct keys
        const int pn = s_selpnode[lane], c = s_sellast[lane];
        const unsigned key = (unsigned)pn * 64u + (unsigned)c + 1u;
        const int fresh = nnodes + lane;  // only used if the prefix is new; gaps in the pool are harmless
        unsigned h = (key * 0x9E3779B1u) >> (32 - table_bits);
        for (;;) {
          unsigned long long e = __hip_atomic_load(tab + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (e == 0ull) {
            unsigned long long mine = ((unsigned long long)key << 32) | (unsigned)fresh;
            if (atomicCAS(tab + h, 0ull, mine) == 0ull) {
              par[fresh] = pn;
              lab[fresh] = c;
              node = fresh;
              break;
            }
            continue;  // another lane took the slot: look at it again
          }
          if ((unsigned)(e >> 32) == key) {
            node = (int)(unsigned)e;
            break;
          }
          h = (h + 1u) & tmask;
        }
      }
      s_pb[lane] = s_selb[lane];
      s_pnb[lane] = s_selnb[lane];
      s_node[lane] = node;
      s_pnode[lane] = s_selpnode[lane];
      s_last[lane] = s_sellast[lane];
      s_len[lane] = s_sellen[lane];
    }
    nb = nsel;
    nnodes += W;
    __syncthreads();
  }
  // ---- read the winner back
  __threadfence();
  if (lane == 0) {
    int32_t* o = out + (size_t)b * To;
    int n = 0;
    if (nb > 0) {
      int len = s_len[0];
      int node = s_node[0];
      // write reversed into the tail, then compact forward
      for (int i = len - 1; i >= 0; --i) {  // (nodes were written by other lanes: read them past the L1)
        o[i] = __hip_atomic_load(lab + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        node = __hip_atomic_load(par + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (merge_repeated) {
        for (int i = 0; i < len; ++i)
          if (i == 0 || o[i] != o[i - 1]) o[n++] = o[i];
      } else {
        n = len;
      }
      logp[b] = lse64(s_pb[0], s_pnb[0]);
    } else {
      logp[b] = kNegInfD;
    }
    for (int i = n; i < To; ++i) o[i] = -1;
    out_len[b] = n;
  }
}

}  // namespace

extern "C" {

static int beam_table_bits(size_t nodes) {
  int bits = 6;
  while (((size_t)1 << bits) < 2 * nodes) ++bits;
  return bits;
}

size_t mgr_ctc_beam_ws_bytes(int B, int T, int C, int beam) {
  (void)C;
  size_t nodes = (size_t)T * (beam > 0 ? beam : 1) + 2;
  return mgr_align_up((size_t)B * nodes * sizeof(int32_t), 256) * 2 + ((size_t)B << beam_table_bits(nodes)) * sizeof(unsigned long long);
}

int mgr_ctc_beam_search(mgr_ctx* c, const float* P, const int32_t* input_len, int B, int T, int C, int skip, int blank,
                        int beam, float eps, int merge_repeated, int32_t* out, int32_t* out_len, double* logp, void* ws,
                        size_t ws_bytes) {
  MGR_REQUIRE(c && P && input_len && out && out_len && logp, "null argument");
  MGR_REQUIRE(B > 0 && T > skip && skip >= 0 && C > 1 && C <= MAXC, "bad shape (C <= %d)", MAXC);
  MGR_REQUIRE(beam >= 1 && beam <= MAXW, "beam width %d out of [1,%d]", beam, MAXW);
  MGR_REQUIRE(beam * (C + 1) <= 64 * KMAX, "beam*(C+1) too large");
  MGR_REQUIRE(blank >= 0 && blank < C, "blank out of range");
  MGR_REQUIRE(ws && ws_bytes >= mgr_ctc_beam_ws_bytes(B, T, C, beam), "workspace too small");
  int nodes = T * beam + 2;
  MGR_REQUIRE((size_t)nodes < ((size_t)1 << 25), "T*beam too large for the prefix table");
  int32_t* parent = reinterpret_cast<int32_t*>(ws);
  int32_t* label = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + mgr_align_up((size_t)B * nodes * sizeof(int32_t), 256));
  unsigned long long* table =
      reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + 2 * mgr_align_up((size_t)B * nodes * sizeof(int32_t), 256));
  mgr_prof_begin(c, MGR_K_MISC);
  const int per_lane = (beam * (C + 1) + 63) / 64;
#define MGR_BEAM_LAUNCH(KM)                                                                                        \
  hipLaunchKernelGGL(k_beam<KM>, dim3(B), dim3(64), 0, mgr_stream(c), P, input_len, B, T, C, skip, blank, beam, eps, \
                     merge_repeated, out, out_len, logp, parent, label, nodes, table, beam_table_bits(nodes))
  if (per_lane <= 4)
    MGR_BEAM_LAUNCH(4);
  else if (per_lane <= 12)
    MGR_BEAM_LAUNCH(12);
  else
    MGR_BEAM_LAUNCH(KMAX);
#undef MGR_BEAM_LAUNCH
  MGR_LAUNCH_CHECK();
  mgr_prof_end(c, MGR_K_MISC);
  return 0;
}

}  // extern "C"
