// radix.hip -- least-significant-digit radix sort of (u32 key, u32 value) pairs, hand-written for CDNA4 wavefronts.
//
// Two users, both in the surfel pass (capi.cpp): the 16-bit Morton keys that put the surfel pool in position order before
// it is traced (k_surfel_keys), and the 26-bit hash locations that group the pass's insert requests into independent
// probe-window clusters for the deterministic parallel apply (k_surfel_apply_keys / k_surfel_apply_clusters).
// A few hundred thousand items: the sort is launch- and latency-bound, so it is two small launches per digit:
//   k_radix_histogram per-tile histogram of the digit (LDS atomics), written as a [tile][digit] table
//   k_radix_scatter   every workgroup (a) works out where its tile's items of each digit go from that table itself -- digit
//                     totals and the counts of the tiles before it, ~n_tiles coalesced loads per thread, instead of a separate
//                     scan launch --, (b) ranks its items among the equal digits of the tile with wave ballots (64 items match
//                     their digit bit by bit, popcount of the lower matching lanes) plus per-wave running counters in LDS --
//                     every wave owns a contiguous run of the tile, so (wave, round, lane) order is index order and the sort
//                     is stable with no global atomics anywhere --, (c) scatters.
// (Measured and dropped: counting every scattered item into the NEXT pass's histogram at its destination tile with global
// atomics, which saves the histogram launch: 19 us when the keys are spread, 150 us when a third of them are equal -- the
// "no insert" requests of the apply order -- and the adds pile up on one counter.)
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace dust {
namespace {

constexpr uint32_t kThreads = 512, kWaves = kThreads / 64, kRounds = 8, kTile = kThreads * kRounds;  // 4096 items per workgroup: the
// pool's 345 600 items are 85 tiles (8192-item tiles, 43 workgroups, left most CUs idle: surfel pass +2 %; 2048-item tiles make the
// [tile][digit] table every workgroup sums over four times as long: +2 %)

struct PassArgs {
  const uint32_t* keys_in;
  const uint32_t* vals_in;
  uint32_t* keys_out;
  uint32_t* vals_out;
  uint32_t* hist;       // [tile][digit]
  uint32_t n, n_tiles, shift;
};

template <int BITS>
__global__ void __launch_bounds__(kThreads) k_radix_histogram(PassArgs a) {
  __shared__ uint32_t bins[1 << BITS];
  for (uint32_t d = threadIdx.x; d < (1u << BITS); d += kThreads) bins[d] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kTile;
  uint32_t key[kRounds];
#pragma unroll
  for (uint32_t r = 0; r < kRounds; ++r) {  // the loads in one batch, then the LDS atomics
    const uint32_t i = base + r * kThreads + threadIdx.x;
    key[r] = i < a.n ? a.keys_in[i] : 0u;
  }
#pragma unroll
  for (uint32_t r = 0; r < kRounds; ++r) {
    const uint32_t i = base + r * kThreads + threadIdx.x;
    if (i < a.n) atomicAdd(&bins[(key[r] >> a.shift) & ((1u << BITS) - 1u)], 1u);
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < (1u << BITS); d += kThreads) a.hist[(size_t)blockIdx.x * (1u << BITS) + d] = bins[d];
}

template <int BITS>
__global__ void __launch_bounds__(kThreads) k_radix_scatter(PassArgs a) {
  constexpr uint32_t NB = 1u << BITS;
  __shared__ uint32_t cnt[kWaves][NB];  // per wave: items of each digit seen so far; later: where the wave's items of the digit go
  __shared__ uint32_t start[NB];        // where this tile's items of each digit go
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  for (uint32_t d = threadIdx.x; d < kWaves * NB; d += kThreads) (&cnt[0][0])[d] = 0;
  // (a) digit totals and the share of the tiles before this one
  for (uint32_t d = threadIdx.x; d < NB; d += kThreads) {
    uint32_t total = 0, before = 0;
#pragma unroll 16  // sixteen independent loads per round trip (the rolled loop made one trip per two tiles: most of the kernel's time)
    for (uint32_t t = 0; t < a.n_tiles; ++t) {
      const uint32_t c = a.hist[(size_t)t * NB + d];
      total += c;
      before += t < blockIdx.x ? c : 0u;
    }
    start[d] = before;
    cnt[0][d] = total;  // parked here until the scan below has read it
  }
  __syncthreads();
  if (wave == 0) {  // exclusive scan of the digit totals: each lane a run of NB / 64 digits, the run sums through shuffles
    constexpr uint32_t kRun = NB / 64;
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < kRun; ++k) sum += cnt[0][lane * kRun + k];
    uint32_t inc = sum;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(inc, d);
      if (lane >= d) inc += up;
    }
    uint32_t run = inc - sum;
#pragma unroll
    for (uint32_t k = 0; k < kRun; ++k) {
      const uint32_t c = cnt[0][lane * kRun + k];
      cnt[0][lane * kRun + k] = 0;
      start[lane * kRun + k] += run;
      run += c;
    }
  }
  __syncthreads();
  // (b) ranks
  const uint32_t base = blockIdx.x * kTile + wave * (kRounds * 64u);  // a wave owns kRounds * 64 consecutive items
  const uint64_t lower = (1ull << lane) - 1ull;
  uint32_t key[kRounds], val[kRounds], rank[kRounds];
  // every round's loads first, in one batch: the wave-level fences of the ranking loop keep the compiler from hoisting a round's loads over
  // the round before it, and eight dependent round trips to memory were most of this kernel's 20 us
#pragma unroll
  for (uint32_t r = 0; r < kRounds; ++r) {
    const uint32_t i = base + r * 64u + lane;
    const bool valid = i < a.n;
    key[r] = valid ? a.keys_in[i] : 0u;
    val[r] = valid ? a.vals_in[i] : 0u;
  }
#pragma unroll
  for (uint32_t r = 0; r < kRounds; ++r) {
    const uint32_t i = base + r * 64u + lane;
    const bool valid = i < a.n;
    const uint32_t digit = (key[r] >> a.shift) & (NB - 1u);
    uint64_t peers = __ballot(valid);  // lanes holding the same digit as this one
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      const bool bit = (digit >> b) & 1u;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t before = cnt[wave][digit];
    rank[r] = before + (uint32_t)__popcll(peers & lower);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every peer has read the counter before its first lane advances it
    if (valid && (peers & lower) == 0) cnt[wave][digit] = before + (uint32_t)__popcll(peers);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < NB; d += kThreads) {  // where each wave's items of digit d start
    uint32_t run = start[d];
#pragma unroll
    for (uint32_t w = 0; w < kWaves; ++w) {
      const uint32_t c = cnt[w][d];
      cnt[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
  // (c) scatter
#pragma unroll
  for (uint32_t r = 0; r < kRounds; ++r) {
    const uint32_t i = base + r * 64u + lane;
    if (i < a.n) {
      const uint32_t pos = cnt[wave][(key[r] >> a.shift) & (NB - 1u)] + rank[r];
      a.keys_out[pos] = key[r];
      a.vals_out[pos] = val[r];
    }
  }
}

template <int BITS>
void run_passes(void* scratch, uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n, uint32_t passes,
                hipStream_t s) {
  PassArgs p;
  p.n = n;
  p.n_tiles = (n + kTile - 1) / kTile;
  p.hist = static_cast<uint32_t*>(scratch);
  bool from_a = true;
  for (uint32_t k = 0; k < passes; ++k) {
    p.keys_in = from_a ? keys_a : keys_b; p.vals_in = from_a ? vals_a : vals_b;
    p.keys_out = from_a ? keys_b : keys_a; p.vals_out = from_a ? vals_b : vals_a;
    p.shift = k * BITS;
    hipLaunchKernelGGL(k_radix_histogram<BITS>, dim3(p.n_tiles), dim3(kThreads), 0, s, p);
    hipLaunchKernelGGL(k_radix_scatter<BITS>, dim3(p.n_tiles), dim3(kThreads), 0, s, p);
    from_a = !from_a;
  }
}

}  // namespace

// bytes of scratch radix_sort_pairs needs for n items
size_t radix_sort_scratch_bytes(uint32_t n) { return (size_t)((n + kTile - 1) / kTile) * (1u << 9) * sizeof(uint32_t); }

// Stable sort of n pairs by the low `key_bits` bits of the key. Both buffer pairs are clobbered; *in_b says which one
// holds the result (a: keys_a/vals_a, b: keys_b/vals_b). 8-bit digits for keys of up to 16 bits (two passes), 9-bit
// digits above (26 bits: three passes, 32: four).
hipError_t radix_sort_pairs(void* scratch, uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n,
                            uint32_t key_bits, bool* in_b, hipStream_t s) {
  *in_b = false;
  if (n == 0 || key_bits == 0) return hipSuccess;
  if (key_bits <= 16) {
    const uint32_t passes = (key_bits + 7) / 8;
    run_passes<8>(scratch, keys_a, vals_a, keys_b, vals_b, n, passes, s);
    *in_b = passes & 1u;
  } else {
    const uint32_t passes = (key_bits + 8) / 9;
    run_passes<9>(scratch, keys_a, vals_a, keys_b, vals_b, n, passes, s);
    *in_b = passes & 1u;
  }
  return hipGetLastError();
}

}  // namespace dust
