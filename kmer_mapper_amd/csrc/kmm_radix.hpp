// kmm_radix.hpp — part of libkmm (MI355X / gfx950); included by kmm.hip inside its anonymous namespace.
// Radix path: two block-local partition passes by hash range, then a probe of LDS-resident index slices.
#pragma once

// ------------------------------------------------------------------------------------------------
// Why.  A probe that goes to HBM costs one 64-byte fabric request per k-mer and MI355X serves ~55 G
// such requests/s whatever their width (profiles/r01/gather_bench_mi355x.txt); random global atomics
// ~27 G/s.  Large batches are therefore turned into STREAMING work: the k-mers are grouped by hash
// range until the slice of the index that a group can hit fits into LDS, and a workgroup then answers
// the whole group from LDS and counts the hits there.  Every byte moved is a coalesced stream; there is
// no random HBM access and no global atomic per hit.
//
//   fine partition  f = (kmer % modulo) >> w          (2^w buckets, <= 4096: slice = 16 KB bucket directory +
//                                                      keys + hit counters of its entries in LDS)
//   coarse partition c = f >> f2                       (F2 = 2^f2 fine partitions each, F1 of them)
//
//   pass 1  k_rx_p1   reads (or a k-mer array) -> k-mers; every workgroup sorts a block of 8192 positions
//                     by COARSE partition inside LDS and writes the sorted block contiguously into its own
//                     block area, plus a directory row start1[block][0..F1] (where each run starts)
//           k_rx_colsum / k_rx_mid / k_rx_colscan   column prefix of the directory: P1T[c][block] = k-mers of
//                     coarse partition c before that block, so that c's runs form one virtual array; it is
//                     cut into items of 8192 k-mers (item_desc = first block, c)
//   pass 2  k_rx_p2   item (c, j): gathers its 8192 k-mers from the runs (each ~B/F1 k-mers, contiguous),
//                     sorts them by FINE partition inside LDS, writes the sorted item + directory row start2
//   pass 3  k_rx_p3   work item (fine partition f, up to 1024 items of its coarse partition): loads f's slice
//                     (directory, keys) into LDS, streams f's runs from the items, probes LDS, counts hits per
//                     entry in LDS, then adds the counters to the per-entry count vector `ecnt` (contiguous
//                     atomics), applying the frequency filter (mapper.pyx:64-66) there
//   flush   k_rx_flush  at the next synchronising call: counts[node[e]] += ecnt[e]  (mapper.pyx:68 summed per
//                     entry first — the reference's GpuCounter does exactly this, gpu_counter.py:26-37)
//
// Nothing depends on partition sizes being balanced: block areas and items are exact, a run that is longer
// than expected is just a longer contiguous copy, a partition with many items is probed by several work
// items.  Buckets whose entries lie beyond the slice's LDS key capacity (RX_ECAP entries per fine partition)
// are walked in HBM instead, so results never depend on the layout.
// ------------------------------------------------------------------------------------------------
constexpr int RX_NT = 512;            // threads per workgroup of the three passes
constexpr int RX_R = 4;               // rounds of two 1024-position tiles per pass-1 block
constexpr int RX_B = 2 * 1024 * RX_R; // 8192: positions per pass-1 block = k-mer capacity of a block area / item
constexpr int RX_KPT = RX_B / RX_NT;  // 16 k-mers per thread
constexpr int RX_MAXF = 256;          // largest fan-out of one pass
constexpr int RX_CH = 1024;           // blocks per chunk of the directory scan
constexpr int RX_IC = 1024;           // pass-2 items per pass-3 work item
constexpr int RX_LPR = 32;            // lanes that copy one run
constexpr int RX_NG = RX_NT / RX_LPR; // run copiers per workgroup
constexpr int RX_WMAX = 4096;         // buckets per fine partition (LDS directory)
constexpr int RX_ECAP = 4096;         // entries of a fine partition kept in LDS (keys + counters)
enum { MODE_KMERS = 3 };              // pass-1 source: a uint64 k-mer array instead of read bytes

struct RxView {
    // index side (built once at kmm_index_create)
    const uint32_t *pstart; // [modulo + 1] first entry of every bucket in bucket order (exclusive prefix of the
                            //              bucket sizes; pstart[modulo] = S): any 2^w-bucket slice is a directory
    const uint64_t *pkeys;  // [S] entry k-mers in bucket order
    const uint16_t *pfreq;  // [S]
    uint32_t *ecnt;         // [S] per-entry hit counts not yet added to the node counts
    int w, f2;
    uint32_t PF, F1, F2;
    // batch side
    uint32_t NB;           // pass-1 output blocks of this sub-batch
    uint32_t max_items;
    uint64_t *buf1, *buf2;
    uint16_t *start1;      // [NB][F1 + 1]
    uint32_t *P1T;         // [F1][NB + 1]
    uint16_t *S1T;         // [F1][NB]
    uint32_t *csum;        // [chunks][F1]
    uint32_t *T1;          // [F1]
    uint32_t *item_base;   // [F1 + 1]
    uint32_t *work_base;   // [F1 + 1]
    uint2 *item_desc;      // [max_items] {first block, coarse partition}
    uint16_t *start2;      // [max_items][F2 + 1]
    uint32_t *ctrl;        // [0] items, [1] pass-3 work items
    unsigned long long *queue; // [0] pass-2 item queue, [16] pass-3 work queue
};

__device__ __forceinline__ uint32_t rx_fine(const IndexView &iv, const RxView &rx, uint64_t q)
{
    return (uint32_t)(fastmod(q, iv.modulo, iv.magic) >> rx.w);
}

// Exclusive scan of s_in[0..n) (n <= 256) into s_out[0..n], s_out[n] = total, by a RX_NT-thread workgroup.
// Call after a barrier that completes s_in; ends with a barrier.
__device__ __forceinline__ uint32_t rx_scan256(const uint32_t *s_in, uint32_t *s_out, int n, uint32_t *s_wave)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t v = tid < n ? s_in[tid] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d)
            inc += o;
    }
    if (wave < 4 && lane == 63)
        s_wave[wave] = inc;
    __syncthreads();
    const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (tid < n) {
        uint32_t base = 0;
        for (int x = 0; x < wave; ++x)
            base += s_wave[x];
        s_out[tid] = base + inc - v;
    }
    if (tid == 0)
        s_out[n] = total;
    __syncthreads();
    return total;
}

// Counting sort of the workgroup's k-mers (RX_KPT per thread, `valid` = which are real) by key(q) < F inside
// LDS, then: the sorted run array goes to `out` as one contiguous coalesced copy, where each key's run starts
// (and the total) to dir_row[0..F].  sbuf may hold the inputs: they are in registers before anything is written.
template <typename KeyFn>
__device__ __forceinline__ void rx_sort_emit(const uint64_t (&q)[RX_KPT], uint32_t valid, KeyFn key, int F,
                                             uint64_t *sbuf, uint32_t *s_cnt, uint32_t *s_base, uint32_t *s_wave,
                                             uint64_t *__restrict__ out, uint16_t *__restrict__ dir_row)
{
    const int tid = threadIdx.x;
    if (tid < F)
        s_cnt[tid] = 0;
    __syncthreads();
    uint32_t cr[RX_KPT]; // key << 16 | rank inside the key's run
#pragma unroll
    for (int i = 0; i < RX_KPT; ++i) {
        cr[i] = 0;
        if ((valid >> i) & 1u) {
            const uint32_t c = key(q[i]);
            cr[i] = (c << 16) | atomicAdd(&s_cnt[c], 1u);
        }
    }
    __syncthreads();
    const uint32_t total = rx_scan256(s_cnt, s_base, F, s_wave);
    if (tid <= F)
        dir_row[tid] = (uint16_t)s_base[tid];
#pragma unroll
    for (int i = 0; i < RX_KPT; ++i)
        if ((valid >> i) & 1u)
            sbuf[s_base[cr[i] >> 16] + (cr[i] & 0xFFFFu)] = q[i];
    __syncthreads();
    const uint4 *s4 = reinterpret_cast<const uint4 *>(sbuf);
    uint4 *o4 = reinterpret_cast<uint4 *>(out);
    for (uint32_t i = tid; i < (total + 1) / 2; i += RX_NT)
        o4[i] = s4[i];
    __syncthreads();
}

__device__ __forceinline__ void rx_stat_add(const IndexView &iv, int which, uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0 && v)
        atomicAdd(&iv.stats[(size_t)((blockIdx.x * 8 + (threadIdx.x >> 6)) % KMM_STAT_SHARDS) * KMM_STAT_STRIDE + which],
                  (unsigned long long)v);
}

// ------------------------------------------------------------------------------------------------
// pass 1
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(RX_NT, 4) k_rx_p1(ReadsView rv, const uint64_t *__restrict__ kmers_in, int64_t n_in,
                                                 IndexView iv, RxView rx, int k, int also_rc, int64_t tile_begin,
                                                 uint32_t n_src)
{
    __shared__ TileSmem<4> sm[2];
    __shared__ uint64_t sbuf[RX_B];
    __shared__ uint32_t s_cnt[RX_MAXF], s_base[RX_MAXF + 1], s_wave[4];
    const int tid = threadIdx.x, half = tid >> 8, ltid = tid & 255;
    TileConst tc;
    tc.kmask = 0; tc.bmask = 0; tc.aligned = false;
    if (MODE != MODE_KMERS) {
        sm[half].lut[ltid] = rv.lut[ltid];
        tc = tile_const(rv, k);
    }
    const uint32_t X = also_rc ? 2u : 1u;
    const int F1 = (int)rx.F1;
    uint32_t lookups = 0;
    auto key = [&](uint64_t x) { return rx_fine(iv, rx, x) >> rx.f2; };
    for (uint32_t sb = blockIdx.x; sb < n_src; sb += gridDim.x) {
        uint64_t q[RX_KPT];
        uint32_t valid = 0;
        if (MODE == MODE_KMERS) {
#pragma unroll
            for (int i = 0; i < RX_KPT; ++i) {
                const int64_t idx = (int64_t)sb * RX_B + i * RX_NT + tid;
                q[i] = 0;
                if (idx < n_in) {
                    q[i] = __builtin_nontemporal_load(&kmers_in[idx]);
                    valid |= 1u << i;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < RX_R; ++r) {
                uint64_t qq[4];
                const int64_t tile = tile_begin + ((int64_t)sb * RX_R + r) * 2 + half;
                const uint32_t v = tile_kmers<4, MODE == MODE_KMERS ? MODE_UNIFORM : MODE>(rv, tc, tile, k, sm[half], qq, ltid);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    q[r * 4 + j] = qq[j];
                valid |= v << (r * 4);
            }
        }
        lookups += (uint32_t)__popc(valid) * X;
        for (uint32_t pass = 0; pass < X; ++pass) {
            if (pass) {
#pragma unroll
                for (int i = 0; i < RX_KPT; ++i)
                    q[i] = revcomp(q[i], k);
            }
            const size_t ob = (size_t)sb * X + pass;
            rx_sort_emit(q, valid, key, F1, sbuf, s_cnt, s_base, s_wave, rx.buf1 + ob * RX_B,
                         rx.start1 + ob * (size_t)(F1 + 1));
        }
    }
    rx_stat_add(iv, 0, lookups);
}

// ------------------------------------------------------------------------------------------------
// directory scan between pass 1 and pass 2
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rx_colsum(RxView rx)
{
    const uint32_t c = threadIdx.x, F1 = rx.F1;
    if (c >= F1)
        return;
    const uint32_t b0 = blockIdx.x * RX_CH;
    const uint32_t b1 = b0 + RX_CH < rx.NB ? b0 + RX_CH : rx.NB;
    const uint16_t *p = rx.start1 + (size_t)b0 * (F1 + 1) + c;
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; ++b, p += F1 + 1)
        sum += (uint32_t)p[1] - (uint32_t)p[0];
    rx.csum[(size_t)blockIdx.x * F1 + c] = sum;
}

// exclusive prefix of one value per thread over a 256-thread workgroup; *total gets the sum
__device__ __forceinline__ uint32_t scan256_excl(uint32_t v, uint32_t *s_wave4, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d)
            inc += o;
    }
    __syncthreads(); // s_wave4 may still be read by the previous call
    if (lane == 63)
        s_wave4[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int x = 0; x < wave; ++x)
        base += s_wave4[x];
    *total = s_wave4[0] + s_wave4[1] + s_wave4[2] + s_wave4[3];
    return base + inc - v;
}

// One workgroup: chunk sums -> exclusive chunk offsets per coarse partition, the partition totals, the item
// table (items of RX_B k-mers per coarse partition) and the pass-3 work table.
__global__ void __launch_bounds__(256) k_rx_mid(RxView rx, uint32_t n_chunks)
{
    __shared__ uint32_t s_wave4[4];
    const uint32_t c = threadIdx.x, F1 = rx.F1;
    uint32_t run = 0;
    if (c < F1) {
        uint32_t ch = 0;
        for (; ch + 8 <= n_chunks; ch += 8) { // independent loads first: this loop is latency-bound
            uint32_t t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                t[u] = rx.csum[(size_t)(ch + u) * F1 + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                rx.csum[(size_t)(ch + u) * F1 + c] = run;
                run += t[u];
            }
        }
        for (; ch < n_chunks; ++ch) {
            const uint32_t t = rx.csum[(size_t)ch * F1 + c];
            rx.csum[(size_t)ch * F1 + c] = run;
            run += t;
        }
        rx.T1[c] = run;
        rx.P1T[(size_t)c * (rx.NB + 1) + rx.NB] = run;
    }
    const uint32_t n_items = c < F1 ? (run + RX_B - 1) / RX_B : 0u;
    uint32_t f2c = 0;
    if (c < F1)
        f2c = rx.PF - c * rx.F2 < rx.F2 ? rx.PF - c * rx.F2 : rx.F2;
    const uint32_t works = ((n_items + RX_IC - 1) / RX_IC) * f2c;
    uint32_t tot_items, tot_works;
    const uint32_t ib = scan256_excl(n_items, s_wave4, &tot_items);
    const uint32_t wb = scan256_excl(works, s_wave4, &tot_works);
    if (c < F1) {
        rx.item_base[c] = ib;
        rx.work_base[c] = wb;
    }
    if (c == 0) {
        rx.item_base[F1] = tot_items;
        rx.work_base[F1] = tot_works;
        rx.ctrl[0] = tot_items;
        rx.ctrl[1] = tot_works;
    }
}

__global__ void __launch_bounds__(256) k_rx_colscan(RxView rx)
{
    const uint32_t c = threadIdx.x, F1 = rx.F1;
    if (c >= F1)
        return;
    const uint32_t b0 = blockIdx.x * RX_CH;
    const uint32_t b1 = b0 + RX_CH < rx.NB ? b0 + RX_CH : rx.NB;
    uint32_t run = rx.csum[(size_t)blockIdx.x * F1 + c];
    const uint32_t ib = rx.item_base[c];
    const uint16_t *p = rx.start1 + (size_t)b0 * (F1 + 1) + c;
    uint32_t *P = rx.P1T + (size_t)c * (rx.NB + 1);
    uint16_t *S = rx.S1T + (size_t)c * rx.NB;
    for (uint32_t b = b0; b < b1; ++b, p += F1 + 1) {
        const uint32_t s0 = p[0], cnt = (uint32_t)p[1] - s0;
        P[b] = run;
        S[b] = (uint16_t)s0;
        if (cnt) { // items whose first k-mer lies in this run
            uint32_t m = (run + RX_B - 1) / RX_B;
            while ((uint64_t)m * RX_B < (uint64_t)run + cnt) {
                rx.item_desc[ib + m] = make_uint2(b, c);
                ++m;
            }
        }
        run += cnt;
    }
}

// ------------------------------------------------------------------------------------------------
// pass 2
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RX_NT, 4) k_rx_p2(IndexView iv, RxView rx)
{
    __shared__ uint64_t sbuf[RX_B];
    __shared__ uint32_t s_cnt[RX_MAXF], s_base[RX_MAXF + 1], s_wave[4];
    __shared__ unsigned long long s_next;
    const int tid = threadIdx.x, grp = tid / RX_LPR, lg = tid % RX_LPR;
    const uint32_t n_items = rx.ctrl[0], NB = rx.NB;
    const int F2 = (int)rx.F2;
    auto key = [&](uint64_t x) { return rx_fine(iv, rx, x) & (uint32_t)(F2 - 1); };
    for (;;) {
        if (tid == 0)
            s_next = atomicAdd(&rx.queue[0], 1ull);
        __syncthreads();
        const unsigned long long nx = s_next;
        const uint32_t item = (uint32_t)nx;
        __syncthreads(); // s_next may be rewritten only after everyone has read it
        if (nx >= (unsigned long long)n_items)
            break;
        const uint2 d = rx.item_desc[item];
        const uint32_t b0 = d.x, c = d.y;
        const uint32_t lo = (item - rx.item_base[c]) * RX_B;
        const uint32_t Tc = rx.T1[c];
        const uint32_t hi = Tc - lo < (uint32_t)RX_B ? Tc : lo + RX_B;
        const uint32_t n = hi - lo;
        const uint32_t *P = rx.P1T + (size_t)c * (NB + 1);
        const uint16_t *S = rx.S1T + (size_t)c * NB;
        // gather: copier g takes runs b0 + g, b0 + g + RX_NG, ...; four runs in flight per copier
        for (uint32_t bb = b0 + grp; bb < NB; bb += RX_NG * 4) {
            uint32_t vs[4], ve[4], from[4], to[4];
            const uint64_t *src[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t b = bb + u * RX_NG;
                vs[u] = b < NB ? P[b] : hi;
            }
            if (vs[0] >= hi)
                break; // P is non-decreasing: every later run lies beyond the item as well
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t b = bb + u * RX_NG;
                const bool in = b < NB && vs[u] < hi;
                ve[u] = in ? P[b + 1] : vs[u];
                const uint32_t s = in ? S[b] : 0u;
                from[u] = vs[u] > lo ? vs[u] : lo;
                to[u] = ve[u] < hi ? ve[u] : hi;
                if (!in)
                    to[u] = from[u] = 0;
                src[u] = rx.buf1 + (size_t)b * RX_B + s - vs[u]; // src[u][v] for virtual index v
            }
            uint64_t x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t v = from[u] + lg;
                x[u] = v < to[u] ? __builtin_nontemporal_load(&src[u][v]) : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t v = from[u] + lg;
                if (v < to[u])
                    sbuf[v - lo] = x[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                for (uint32_t v = from[u] + lg + RX_LPR; v < to[u]; v += RX_LPR)
                    sbuf[v - lo] = __builtin_nontemporal_load(&src[u][v]);
        }
        __syncthreads();
        uint64_t q[RX_KPT];
        uint32_t valid = 0;
#pragma unroll
        for (int i = 0; i < RX_KPT; ++i) {
            const uint32_t idx = i * RX_NT + tid;
            q[i] = 0;
            if (idx < n) {
                q[i] = sbuf[idx];
                valid |= 1u << i;
            }
        }
        rx_sort_emit(q, valid, key, F2, sbuf, s_cnt, s_base, s_wave, rx.buf2 + (size_t)item * RX_B,
                     rx.start2 + (size_t)item * (F2 + 1));
    }
}

// ------------------------------------------------------------------------------------------------
// pass 3
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RX_NT, 4) k_rx_p3(IndexView iv, RxView rx, int max_freq)
{
    __shared__ uint32_t sdir[RX_WMAX + 1]; // bucket b of the slice holds entries [sdir[b], sdir[b + 1]) - e0
    __shared__ uint64_t skeys[RX_ECAP];
    __shared__ uint32_t scnt[RX_ECAP];
    __shared__ uint32_t s_wb[RX_MAXF + 1];
    __shared__ unsigned long long s_next;
    const int tid = threadIdx.x, grp = tid / RX_LPR, lg = tid % RX_LPR;
    const uint32_t n_work = rx.ctrl[1], F1 = rx.F1, F2 = rx.F2;
    const uint32_t W = 1u << rx.w;
    for (uint32_t i = tid; i <= F1; i += RX_NT)
        s_wb[i] = rx.work_base[i];
    uint32_t hits = 0;
    for (;;) {
        if (tid == 0)
            s_next = atomicAdd(&rx.queue[16], 1ull);
        __syncthreads(); // also: s_wb is loaded; the previous work item's flush is done
        const uint32_t wi = (uint32_t)s_next;
        const bool done = s_next >= (unsigned long long)n_work;
        __syncthreads();
        if (done)
            break;
        uint32_t c_lo = 0, c_hi = F1; // largest c with s_wb[c] <= wi
        while (c_hi - c_lo > 1) {
            const uint32_t mid = (c_lo + c_hi) >> 1;
            if (s_wb[mid] <= wi)
                c_lo = mid;
            else
                c_hi = mid;
        }
        const uint32_t c = c_lo;
        const uint32_t f2c = rx.PF - c * F2 < F2 ? rx.PF - c * F2 : F2;
        const uint32_t rem = wi - s_wb[c];
        const uint32_t chunk = rem / f2c, g = rem % f2c;
        const uint32_t f = c * F2 + g;
        const uint64_t h0 = (uint64_t)f << rx.w, M = iv.modulo;
        const uint32_t e0 = rx.pstart[h0], e1 = rx.pstart[h0 + W < M ? h0 + W : M];
        const uint32_t ne = e1 - e0 < (uint32_t)RX_ECAP ? e1 - e0 : (uint32_t)RX_ECAP;
        for (uint32_t i = tid; i <= W; i += RX_NT)
            sdir[i] = rx.pstart[h0 + i < M ? h0 + i : M] - e0;
        for (uint32_t i = tid; i < ne; i += RX_NT) {
            skeys[i] = rx.pkeys[(size_t)e0 + i];
            scnt[i] = 0;
        }
        __syncthreads();
        const uint32_t it0 = rx.item_base[c] + chunk * RX_IC;
        const uint32_t it_end = rx.item_base[c + 1];
        const uint32_t it1 = it_end - it0 < (uint32_t)RX_IC ? it_end : it0 + RX_IC;
        auto probe = [&](uint64_t q) {
            const uint32_t hb = (uint32_t)(fastmod(q, iv.modulo, iv.magic) - h0) & (W - 1u);
            const uint32_t st = sdir[hb], cn = sdir[hb + 1] - st;
            if (cn == 0u)
                return;
            if (st + cn > ne) { // the bucket's entries lie beyond the LDS copy: same walk over the HBM arrays
                for (uint32_t j = 0; j < cn; ++j) {
                    const size_t e = (size_t)e0 + st + j;
                    if (rx.pkeys[e] == q && (int)rx.pfreq[e] <= max_freq) {
                        atomicAdd(&rx.ecnt[e], 1u);
                        ++hits;
                    }
                }
                return;
            }
            for (uint32_t j = 0; j < cn; ++j)
                if (skeys[st + j] == q)
                    atomicAdd(&scnt[st + j], 1u);
        };
        for (uint32_t ib = it0 + grp; ib < it1; ib += RX_NG * 4) {
            uint32_t from[4], to[4];
            const uint64_t *src[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t it = ib + u * RX_NG;
                from[u] = to[u] = 0;
                src[u] = rx.buf2;
                if (it < it1) {
                    const uint16_t *row = rx.start2 + (size_t)it * (F2 + 1) + g;
                    from[u] = row[0];
                    to[u] = row[1];
                    src[u] = rx.buf2 + (size_t)it * RX_B;
                }
            }
            uint64_t x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t v = from[u] + lg;
                x[u] = v < to[u] ? __builtin_nontemporal_load(&src[u][v]) : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (from[u] + lg < to[u])
                    probe(x[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                for (uint32_t v = from[u] + lg + RX_LPR; v < to[u]; v += RX_LPR)
                    probe(__builtin_nontemporal_load(&src[u][v]));
        }
        __syncthreads();
        // LDS counters -> per-entry count vector; the frequency filter of mapper.pyx:64-66 is applied here
        for (uint32_t i = tid; i < ne; i += RX_NT) {
            const uint32_t cn = scnt[i];
            if (cn && (int)rx.pfreq[(size_t)e0 + i] <= max_freq) {
                atomicAdd(&rx.ecnt[(size_t)e0 + i], cn);
                hits += cn;
            }
        }
    }
    rx_stat_add(iv, 1, hits);
}

// ------------------------------------------------------------------------------------------------
// flush: per-entry counts -> node counts (and the per-entry vector is cleared)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rx_flush(IndexView iv, uint32_t *__restrict__ ecnt,
                                                  const uint32_t *__restrict__ pnodes, uint64_t n,
                                                  uint32_t *__restrict__ ecnt_acc)
{
    __shared__ NodeAgg agg;
    agg_init(agg);
    __syncthreads();
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (uint64_t)gridDim.x * 256) {
        const uint32_t cn = ecnt[e];
        if (cn) {
            ecnt[e] = 0;
            if (ecnt_acc)
                ecnt_acc[e] += cn; // per-k-mer counting mode: what GpuCounter's table holds (gpu_counter.py:29-34)
            agg_add_n(iv, agg, pnodes[e], cn);
        }
    }
    __syncthreads();
    agg_flush_counts(iv, agg);
}

// per-entry counts in the index's own entry order (GpuCounter semantics, gpu_counter.py:29-34)
__global__ void k_rx_entry_counts(const uint32_t *__restrict__ ecnt, const uint32_t *__restrict__ porig, uint64_t n,
                                  uint32_t *__restrict__ out)
{
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x)
        out[porig[e]] = ecnt[e];
}

// ------------------------------------------------------------------------------------------------
// index side: bucket-ordered entry arrays + the 4-byte bucket directory, built once at index creation
// ------------------------------------------------------------------------------------------------
__global__ void k_rx_bucket_sizes(const int32_t *__restrict__ h2i, const int32_t *__restrict__ nk, uint64_t modulo,
                                  int64_t n_entries, uint32_t *__restrict__ out)
{
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h <= modulo; h += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t c = 0;
        if (h < modulo) { // invalid buckets: KMM_ERR_INDEX from k_pack_buckets
            const int32_t n = nk[h], s = h2i[h];
            if (n > 0 && s >= 0 && (int64_t)s + n <= n_entries)
                c = (uint32_t)n;
        }
        out[h] = c;
    }
}

__global__ void k_rx_pack(const int32_t *__restrict__ h2i, const uint64_t *__restrict__ kmers,
                          const int32_t *__restrict__ nodes, const uint16_t *__restrict__ freqs, uint64_t modulo,
                          int64_t max_node_id, const uint32_t *__restrict__ pstart, uint64_t *__restrict__ pkeys,
                          uint16_t *__restrict__ pfreq, uint32_t *__restrict__ pnodes, uint32_t *__restrict__ porig)
{
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < modulo; h += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t st = pstart[h], c = pstart[h + 1] - st;
        const int32_t s = h2i[h];
        for (uint32_t j = 0; j < c; ++j) {
            int32_t nd = nodes[s + j];
            if (nd < 0 || (int64_t)nd > max_node_id)
                nd = 0; // reported by k_pack_entries
            pkeys[(size_t)st + j] = kmers[s + j];
            pfreq[(size_t)st + j] = freqs[s + j];
            pnodes[(size_t)st + j] = (uint32_t)nd;
            porig[(size_t)st + j] = (uint32_t)(s + j);
        }
    }
}
