// kmm_partition.hpp — part of libkmm (MI355X / gfx950); included by kmm.hip inside its anonymous namespace.
// Radix-partitioned (L2-local) path: hist / scan / scatter / probe kernels.
#pragma once

// ------------------------------------------------------------------------------------------------
// Partitioned path.  Random probes that miss L2 are capped at ~55 G requests/s on MI355X while
// L2-resident probes run ~4.6x faster, so large batches are first grouped by hash range
// (partition = h >> shift, each partition's bucket-table slice ~1 MiB, i.e. L2-resident) and then
// probed partition by partition, each XCD working on its own partitions so that the slice stays in
// that XCD's 4 MiB L2.  Per sub-batch, all streaming except the L2-local gathers:
//   k_part_hist     reads -> k-mers per (partition, workgroup): every workgroup owns a fixed set
//                   of tiles (grid-stride), so its histogram row is private — no global atomics
//   k_part_scan1/2  exclusive scan over (partition, workgroup) -> a private, exactly sized output
//                   range per workgroup inside every partition
//   k_part_scatter  same tiles again: k-mers are counting-sorted by partition inside LDS so that
//                   each partition's run leaves the workgroup as contiguous 8-byte stores at the
//                   workgroup's private cursor (kept in LDS)
//   k_part_probe    workgroup b takes chunks of the k-mers of XCD (b mod 8) — partitions are laid
//                   out XCD-major, p mod 8 = XCD — in lock step with the other workgroups of that
//                   XCD.  blockIdx mod 8 is where the dispatcher has been observed to place a
//                   workgroup; it is used for L2 affinity only — every chunk is processed exactly
//                   once whatever the placement.
// ------------------------------------------------------------------------------------------------
constexpr int KMM_MAX_PARTS = 1024;
constexpr int KMM_N_XCD = 8;
constexpr int KMM_CHUNK = 2048;    // k-mers per probe work item (256 lanes x 8)
constexpr int KMM_PART_GRID = 2048; // workgroups of the hist / scatter kernels (fixed: rows of wg_hist)

struct PartView {
    int shift; // partition = hash >> shift
    int P;     // number of partitions, <= KMM_MAX_PARTS
    int PX;    // partition slots per XCD = ceil(P / 8); slot(p) = (p % 8) * PX + p / 8
    uint32_t *wg_hist;  // [8*PX][KMM_PART_GRID] k-mers per (slot, workgroup); after scan1: exclusive
                        //                       prefix over the workgroups of the slot
    uint32_t *slot_tot; // [8*PX]     k-mers per slot
    uint32_t *slot_off; // [8*PX + 1] exclusive prefix of slot_tot (XCD x owns [x*PX, (x+1)*PX))
    uint64_t *kmers;    // grouped k-mers of the sub-batch
};

__device__ __forceinline__ uint32_t slot_of(const IndexView &iv, const PartView &pv, uint64_t q)
{
    const uint32_t p = (uint32_t)(fastmod(q, iv.modulo, iv.magic) >> pv.shift);
    return (p & (KMM_N_XCD - 1)) * pv.PX + (p >> 3);
}

template <int S, int MODE>
__global__ void __launch_bounds__(256) k_part_hist(ReadsView rv, IndexView iv, int k, int also_rc,
                                                   PartView pv, int64_t tile_begin, int64_t tile_end)
{
    __shared__ TileSmem<S> sm;
    __shared__ uint32_t s_hist[KMM_MAX_PARTS + KMM_N_XCD];
    const int n_slots = KMM_N_XCD * pv.PX;
    sm.lut[threadIdx.x] = rv.lut[threadIdx.x];
    for (int i = threadIdx.x; i < n_slots; i += 256)
        s_hist[i] = 0;
    const TileConst tc = tile_const(rv, k);
    for (int64_t tile = tile_begin + blockIdx.x; tile < tile_end; tile += gridDim.x) {
        uint64_t q[S];
        const uint32_t valid = tile_kmers<S, MODE>(rv, tc, tile, k, sm, q);
#pragma unroll
        for (int j = 0; j < S; ++j)
            if ((valid >> j) & 1u) {
                atomicAdd(&s_hist[slot_of(iv, pv, q[j])], 1u);
                if (also_rc)
                    atomicAdd(&s_hist[slot_of(iv, pv, revcomp(q[j], k))], 1u);
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_slots; i += 256)
        pv.wg_hist[(size_t)i * KMM_PART_GRID + blockIdx.x] = s_hist[i];
}

// One workgroup per slot: exclusive scan of the slot's KMM_PART_GRID per-workgroup counts, in place.
__global__ void __launch_bounds__(256) k_part_scan1(PartView pv)
{
    __shared__ uint32_t s_wave[4];
    constexpr int PER = KMM_PART_GRID / 256;
    uint32_t *row = pv.wg_hist + (size_t)blockIdx.x * KMM_PART_GRID;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        v[i] = row[tid * PER + i];
        sum += v[i];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d);
        if (lane >= d)
            inc += o;
    }
    if (lane == 63)
        s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w)
        base += s_wave[w];
    uint32_t run = base + inc - sum;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        row[tid * PER + i] = run;
        run += v[i];
    }
    if (tid == 255)
        pv.slot_tot[blockIdx.x] = run;
}

// One workgroup of 1024 threads: exclusive scan of the (<= 1032) slot totals.
__global__ void __launch_bounds__(1024) k_part_scan2(PartView pv)
{
    __shared__ uint32_t s_a[2048];
    const int n_slots = KMM_N_XCD * pv.PX;
    const int t = threadIdx.x;
    const uint32_t c0 = t < n_slots ? pv.slot_tot[t] : 0u;
    const uint32_t c1 = t + 1024 < n_slots ? pv.slot_tot[t + 1024] : 0u;
    s_a[t] = c0;
    s_a[t + 1024] = c1;
    __syncthreads();
    for (int d = 1; d < 2048; d <<= 1) { // Hillis-Steele inclusive scan over 2048 slots
        uint32_t v0 = t >= d ? s_a[t - d] : 0u;
        uint32_t v1 = s_a[t + 1024 - d];
        __syncthreads();
        s_a[t] += v0;
        s_a[t + 1024] += v1;
        __syncthreads();
    }
    if (t < n_slots)
        pv.slot_off[t] = s_a[t] - c0;
    if (t + 1024 < n_slots)
        pv.slot_off[t + 1024] = s_a[t + 1024] - c1;
    if (t == 0)
        pv.slot_off[n_slots] = s_a[2047];
}

// Exclusive scan of s_cnt[0..n) into s_loc[0..n) (n <= 1280) by one 256-thread workgroup; returns
// the total.
__device__ __forceinline__ uint32_t block_excl_scan(const uint32_t *s_cnt, uint32_t *s_loc, int n,
                                                    uint32_t *s_wave)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = 5; // 256 x 5 = 1280 >= KMM_MAX_PARTS + 8
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int p = tid * PER + i;
        v[i] = p < n ? s_cnt[p] : 0u;
        sum += v[i];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d);
        if (lane >= d)
            inc += o;
    }
    __syncthreads(); // s_wave may still be read by the previous call
    if (lane == 63)
        s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w)
        base += s_wave[w];
    const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    uint32_t run = base + inc - sum;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int p = tid * PER + i;
        if (p < n)
            s_loc[p] = run;
        run += v[i];
    }
    __syncthreads();
    return total;
}

template <int S, int MODE>
__global__ void __launch_bounds__(256) k_part_scatter(ReadsView rv, IndexView iv, int k, int also_rc,
                                                      PartView pv, int64_t tile_begin,
                                                      int64_t tile_end)
{
    constexpr int T = 256 * S;
    constexpr int NS = KMM_MAX_PARTS + KMM_N_XCD;
    __shared__ TileSmem<S> sm;
    __shared__ uint32_t s_cur[NS]; // this workgroup's next free slot per partition (private range)
    __shared__ uint32_t s_cnt[NS]; // k-mers of this tile per partition
    __shared__ uint32_t s_loc[NS]; // where the partition's run starts in s_km
    __shared__ uint32_t s_wave[4];
    __shared__ uint64_t s_km[T];
    __shared__ uint16_t s_pd[T];
    const int tid = threadIdx.x;
    const int n_slots = KMM_N_XCD * pv.PX;
    sm.lut[tid] = rv.lut[tid];
    for (int i = tid; i < n_slots; i += 256)
        s_cur[i] = pv.slot_off[i] + pv.wg_hist[(size_t)i * KMM_PART_GRID + blockIdx.x];
    const TileConst tc = tile_const(rv, k);
    for (int64_t tile = tile_begin + blockIdx.x; tile < tile_end; tile += gridDim.x) {
        uint64_t q[S];
        const uint32_t valid = tile_kmers<S, MODE>(rv, tc, tile, k, sm, q);
        for (int round = 0; round < (also_rc ? 2 : 1); ++round) {
            if (round == 1) {
#pragma unroll
                for (int j = 0; j < S; ++j)
                    q[j] = revcomp(q[j], k);
            }
            for (int i = tid; i < n_slots; i += 256)
                s_cnt[i] = 0;
            __syncthreads(); // also: the previous round's readers of s_km / s_pd / s_loc are done
            uint16_t pid[S], rk[S];
#pragma unroll
            for (int j = 0; j < S; ++j) {
                pid[j] = 0;
                rk[j] = 0;
                if ((valid >> j) & 1u) {
                    pid[j] = (uint16_t)slot_of(iv, pv, q[j]);
                    rk[j] = (uint16_t)atomicAdd(&s_cnt[pid[j]], 1u);
                }
            }
            __syncthreads();
            const uint32_t n_tile = block_excl_scan(s_cnt, s_loc, n_slots, s_wave);
#pragma unroll
            for (int j = 0; j < S; ++j)
                if ((valid >> j) & 1u) {
                    const uint32_t pos = s_loc[pid[j]] + rk[j];
                    s_km[pos] = q[j];
                    s_pd[pos] = pid[j];
                }
            __syncthreads();
            for (uint32_t i = tid; i < n_tile; i += 256) {
                const uint32_t sl = s_pd[i];
                pv.kmers[s_cur[sl] + (i - s_loc[sl])] = s_km[i];
            }
            __syncthreads();
            for (int i = tid; i < n_slots; i += 256)
                s_cur[i] += s_cnt[i];
        }
    }
}

template <int U>
__global__ void __launch_bounds__(256) k_part_probe(IndexView iv, PartView pv, int max_freq)
{
    static_assert(256 * U == KMM_CHUNK, "chunk = one k-mer per lane per unroll slot");
    __shared__ NodeAgg agg;
    LaneStats st;
    agg_init(agg);
    __syncthreads();
    const int tid = threadIdx.x;
    const int x = blockIdx.x & (KMM_N_XCD - 1); // expected XCD of this workgroup (speed only)
    const uint32_t j = blockIdx.x >> 3, nj = gridDim.x >> 3;
    const uint32_t begin_x = pv.slot_off[x * pv.PX], end_x = pv.slot_off[(x + 1) * pv.PX];
    const uint32_t n_chunks = (end_x - begin_x + KMM_CHUNK - 1) / KMM_CHUNK;
    for (uint32_t c = j; c < n_chunks; c += nj) {
        const uint32_t begin = begin_x + c * KMM_CHUNK;
        uint64_t q[U];
        uint32_t valid = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = begin + u * 256 + tid;
            q[u] = 0;
            if (i < end_x) {
                q[u] = __builtin_nontemporal_load(&pv.kmers[i]);
                valid |= 1u << u;
            }
        }
        probe_batch<U, PROBE_NARROW>(iv, agg, st, q, valid, max_freq);
    }
    stats_reduce(agg, st);
    __syncthreads();
    agg_flush(iv, agg);
}
