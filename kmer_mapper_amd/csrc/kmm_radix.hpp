// kmm_radix.hpp — part of libkmm (MI355X / gfx950); included by kmm.hip inside its anonymous namespace.
// Radix path: two block-local partition passes by hash range, then a probe of LDS-resident index slices.
#pragma once
#include <type_traits>

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
//           k_rx_colsum / k_rx_chunkscan / k_rx_tables / k_rx_colscan   column prefix of the directory: P1T[c][block] = k-mers of
//                     coarse partition c before that block, so that c's runs form one virtual array; it is
//                     cut into items of 8192 k-mers (item_desc = first block, c)
//   pass 2  k_rx_p2f  item (c, j): gathers its 8192 k-mers from the runs (each ~B/F1 k-mers, contiguous), drops those
//                     whose bucket is empty (LDS bitmap of the coarse partition, where one fits), sorts the rest by
//                     FINE partition inside LDS, writes the sorted item + directory row start2
//                     (k_rx_p2: the round-2 form — piece lists, no filter; behind "radix_filter" = 0)
//   pass 3  k_rx_p3   work item (fine partition f, up to 1024 items of its coarse partition): loads f's slice
//                     (directory, keys) into LDS, streams f's runs from the items, probes LDS, counts hits per
//                     entry in LDS (an entry the frequency filter of mapper.pyx:64-66 excludes carries a flag in
//                     its counter), then adds the counters to the per-entry count vector `ecnt` (contiguous atomics)
//   flush   k_rx_flush_sorted / k_rx_flush  at the next synchronising call: counts[node[e]] += ecnt[e]  (mapper.pyx:68
//                     summed per entry first — the reference's GpuCounter does exactly this, gpu_counter.py:26-37)
//
// What the passes are tuned against (profiles/r03/README.md): no pass sits on one resource (vector ALUs 45-70 % busy,
// 4-5 TB/s of HBM traffic, LDS a third of the time); what moves them is fewer instructions per k-mer and fewer
// workgroup barriers — wavefronts drift apart between barriers and overlap each other's LDS, VALU and memory phases.
// Work between two barriers is kept short: wavefront prefix sums use DPP additions, not ds_bpermute; counters are
// cleared off the critical path; the next block / item is requested before the current one is sorted; nothing in a
// block / item loop may spill (a scratch reload waits for vmcnt(0): the prefetch and the copy-out stores).
//
// Between the passes a k-mer q travels as x = (q / modulo) << (w + f2) | (q % modulo) & (2^(w+f2) - 1): the
// quotient and the hash bits BELOW the coarse partition number.  Inside a coarse partition (and so inside a
// fine one) x determines q, the fine partition is bits [w, w + f2) of x and the bucket inside the slice bits
// [0, w): only pass 1 divides by the modulo (mapper.pyx:54), passes 2 and 3 shift and compare.  The index keys
// are kept in the same form.  rx_configure guarantees that the quotient of ANY 64-bit value fits the upper bits.
//
// Nothing depends on partition sizes being balanced: block areas and items are exact, a run that is longer
// than expected is just a longer contiguous copy, a partition with many items is probed by several work
// items.  Buckets whose entries lie beyond the slice's LDS key capacity (RX_ECAP entries per fine partition)
// are walked in HBM instead, so results never depend on the layout.
// ------------------------------------------------------------------------------------------------
constexpr int RX_NT = 512;            // threads per workgroup of the three passes
constexpr int RX_B = 8192;            // positions per pass-1 block = k-mer capacity of a block area / item: each
                                      // 256-thread half of the workgroup owns 4096 of them (flat reads: one tile of
                                      // 16 windows per lane; records mode: four tiles of 4 windows per lane)
constexpr int RX_KPT = RX_B / RX_NT;  // 16 k-mers per thread
constexpr int RX_MAXF = 512;          // largest fan-out of one pass (512 x 512 slices of 8192 buckets = every modulo < 2^32)
#ifndef RX_CHV
#define RX_CHV 256
#endif
constexpr int RX_CH = RX_CHV;            // blocks per chunk of the directory scan
constexpr int RX_IC = 1024;           // pass-2 items per pass-3 work item
#ifndef RX_LPR2V
#define RX_LPR2V 16
#endif
constexpr int RX_LPR = RX_LPR2V;      // lanes of a run copier (pass 2)
constexpr int RX_NG = RX_NT / RX_LPR; // run copiers per workgroup
#ifndef RX_RB1
#define RX_RB1 4
#endif
#ifndef RX_RB2
#define RX_RB2 8
#endif
constexpr int RX_SUBCAP = 1024;       // sub-runs (<= RX_LPR k-mers each) listed in LDS per window (pass 2)
#ifndef RX_SUBCAP3V
#define RX_SUBCAP3V 3072 // (1024 runs of up to 48 k-mers in one window: configs[1] pass 3 2.40 -> 1.92 ms)
#endif
constexpr int RX_SUBCAP3 = RX_SUBCAP3V; // ... pass 3 (one word each)
#ifndef RX_LPR3
#define RX_LPR3 16
#endif
constexpr bool RX_P3_LINECUT = false;
// (Two variants of pass 3 were measured in round 3 and removed again, both bit-exact — profiles/r03: gathering by k-mer
// with a run table and broadcast run starts as k_rx_p2f does, 2.05 vs 1.94 ms: runs of ~32 k-mers fill the 16-lane pieces
// well; entries 0 and 1 of every bucket unrolled with longer buckets queued per wavefront, 1.96 vs 1.96 ms; the next
// batch of pieces requested before the current one is probed (two register sets), 2.02 vs 1.91 ms; three workgroups per
// CU for slices of at most 2560 entries (48 KB of LDS, 80 VGPRs), 1.94 vs 1.94 ms.)
constexpr int RX_LPR_P3 = RX_LPR3;    // ... pass 3 (its runs are shorter: ~32 k-mers)
constexpr int RX_NG3 = RX_NT / RX_LPR_P3;
#ifndef RX_U3
#define RX_U3 8
#endif
constexpr int RX_U = RX_U3;           // sub-runs in flight per copier (pass 3)
#ifndef RX_P3_GROUP
#define RX_P3_GROUP 4
#endif
constexpr int RX_G3 = RX_P3_GROUP;    // k-mers probed side by side (pass 3)
#ifndef RX_U2V
#define RX_U2V 8
#endif
constexpr int RX_U2 = RX_U2V;         // ... pass 2
// The runs passes 2 and 3 copy are read with PLAIN loads: neighbouring runs share their first and last 128-byte
// line, the XCD-aware work distribution (rx_pop) makes one XCD read them at about the same time, and a non-temporal
// load would drop the shared line from L2 in between (measured: pass 2 4.42 -> 4.21 ms, pass 3 4.01 -> 3.88 ms).
#ifdef RX_NT_LOADS
#define RX_LOAD2(p) __builtin_nontemporal_load(p)
#define RX_LOAD3(p) __builtin_nontemporal_load(p)
#else
#define RX_LOAD2(p) (*(p))
#define RX_LOAD3(p) (*(p))
#endif
constexpr int RX_WMAX = 4096;         // buckets per fine partition (LDS directory): two workgroups of pass 3 per CU
constexpr int RX_ECAP = 4096;         // entries of a fine partition kept in LDS (keys + counters)
constexpr int RX_WMAX_BIG = 8192;     // slices of indexes with more than 256 x 256 x 4096 buckets (e.g. the customary
constexpr int RX_ECAP_BIG = 8192;     // modulo 452 930 477): 140 KB of LDS, one workgroup of pass 3 per CU
constexpr int RX_ECAP_MID = 4608;     // 8192-bucket slices at load factor 0.5 (4096 +- 64 entries): 16-bit directory, 1024-piece
                                      // list, 77 KB of LDS: two workgroups per CU (the 1 B-k-mer index)
constexpr uint32_t RX_FILTERED = 0x80000000u; // pass 3: top bit of an LDS hit counter = entry excluded by max_freq
enum { MODE_KMERS = 3 };              // pass-1 source: a uint64 k-mer array instead of read bytes

struct RxView {
    // index side (built once at kmm_index_create)
    const uint16_t *pstart16; // [PF << w] the same relative to the first entry of the bucket's slice (pass 3 loads half
    const uint32_t *slice_e0; // the bytes); [PF + 1] first entry of every slice.  Null: slices beyond 65535 entries
    const uint16_t *slice_fmax; // [PF] largest frequency among a slice's entries: pass 3 loads a slice's frequencies only when
                                // the call's max_index_lookup_frequency lies below it (mapper.pyx:64-66 filters nothing else)
    const uint32_t *pstart; // [modulo + 1] first entry of every bucket in bucket order (exclusive prefix of the
                            //              bucket sizes; pstart[modulo] = S): any 2^w-bucket slice is a directory
    const uint64_t *pkeys;  // [S] entry k-mers in bucket order, in the packed form the passes carry (rx_pack)
    const uint16_t *pfreq;  // [S]
    uint32_t *ecnt;         // [S] per-entry hit counts not yet added to the node counts
    const uint32_t *occ;    // bit h = bucket h holds an entry (padded by one coarse partition's worth of words), or null
    uint32_t p2f_k;         // k_rx_p2f: items per work unit (one bitmap load, one pipeline fill: 16 / 32 / 64 items measured
                            // 4.47 / 4.40 / 4.36 ms at configs[2]); fewer for small batches, so that every CU gets units
    int w, f2;              // sh = w + f2: hash bits below the coarse partition number
    int occ_shift;          // k_rx_p2f: one bit of its LDS bitmap covers 2^occ_shift buckets (0 .. 2)
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
    uint32_t *work_base;   // [F1 + 1] pass-3 rows (coarse partition, chunk of RX_IC items) before coarse partition c
    uint2 *item_desc;      // [max_items] {first block, coarse partition}
    uint16_t *start2;      // [max_items][F2 + 1]
    uint16_t *start2T;     // [F2 + 1][max_items] (transposed for pass 3: one fine partition's run starts are contiguous)
    uint32_t *ctrl;        // [0] items, [1] pass-3 rows, [2] most items of one coarse partition
    unsigned long long *queue; // 2 x 8 work counters, 128 bytes apart: [16 x] pass 2, [128 + 16 x] pass 3, x = XCD
    uint64_t *probe;       // experiments (RX_PROBE_*: what one more byte costs a pass); null in product builds
};

// ------------------------------------------------------------------------------------------------
// Marginal-byte probes (profiles/r05/marginal_bytes.txt).  "Are the passes bound by their bytes?" decides whether narrower
// records between the passes (7 bytes 1 -> 2, 6 bytes 2 -> 3) can buy time.  Measured directly: a build with
// -DRX_PROBE_P1W=D makes pass 1 write 8 / D MORE bytes per k-mer (every D-th 16-byte piece of the sorted block a second
// time, into a scratch buffer of its own), RX_PROBE_P2R / RX_PROBE_P3R make pass 2 / pass 3 read 8 / D more bytes per
// k-mer (a second, never consumed load per D-th k-mer, half of pass 1's output away: no cache holds it), RX_PROBE_P2W as
// P1W for pass 2's output.  The time such a build adds per added byte is what a saved byte would return.  Product
// builds define none of them (the code below compiles to nothing).
// ------------------------------------------------------------------------------------------------
#ifndef RX_PROBE_P1W
#define RX_PROBE_P1W 0
#endif
#ifndef RX_PROBE_P2R
#define RX_PROBE_P2R 0
#endif
#ifndef RX_PROBE_P2W
#define RX_PROBE_P2W 0
#endif
#ifndef RX_PROBE_P3R
#define RX_PROBE_P3R 0
#endif
#define RX_PROBE_ANY (RX_PROBE_P1W || RX_PROBE_P2R || RX_PROBE_P2W || RX_PROBE_P3R)

// q -> packed form; *coarse gets the coarse partition (= hash >> (w + f2)).
__device__ __forceinline__ uint64_t rx_pack(const IndexView &iv, int sh, uint64_t q, uint32_t *coarse)
{
    uint32_t h; // modulo < 2^31 on this path (rx_configure)
    const uint64_t quo = fastdiv_m31(q, (uint32_t)iv.modulo, iv.magic, &h);
    *coarse = h >> sh;
    return (quo << sh) | (uint64_t)(h & ((1u << sh) - 1u));
}

// Exclusive scan of s_in[0..n) (n <= 512) into s_out[0..n], s_out[n] = total, by a RX_NT-thread workgroup.  Lane l of
// wavefront v < 4 owns the counters 128 v + 2 l and 128 v + 2 l + 1 (one 8-byte LDS read; s_in is 8-byte aligned).
// Call after a barrier that completes s_in; ends with a barrier.  ONEBAR: the wavefront that scans counters
// [128 v, 128 v + 128) sums the counters before them itself (DPP reductions) instead of waiting at a second barrier for
// the other wavefronts' totals: pass 1 2.91 -> 2.80 ms, pass 2 3.89 -> 3.96 ms (so pass 2 keeps the exchange).
// (Letting EVERY wavefront scan all the counters for itself — no barrier at all — measured slower: pass 1 3.58 vs
// 3.20 ms; so did the one-barrier form while the reductions still went through ds_bpermute: 3.78.)
__device__ __forceinline__ uint2 rx_pair(const uint32_t *s_in, int c0, int n)
{
    // counters at and beyond n are not part of the scan (the spare counters behind them are never cleared)
    const uint2 p = *reinterpret_cast<const uint2 *>(s_in + c0);
    return make_uint2(c0 < n ? p.x : 0u, c0 + 1 < n ? p.y : 0u);
}

// n <= 256: one counter per lane of wavefronts 0..3 (the form the passes were tuned with)
template <bool ONEBAR>
__device__ __forceinline__ uint32_t rx_scan256(const uint32_t *s_in, uint32_t *s_out, int n, uint32_t *s_wave, const int tid)
{
    if (ONEBAR) {
    (void)s_wave;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave < 4) {
        const uint32_t v = tid < n ? s_in[tid] : 0u;
        const uint32_t inc = wave_scan_incl(v);
        uint32_t before = 0;
        for (int x = 0; x < wave; ++x) {
            const uint32_t t = x * 64 + lane < n ? s_in[x * 64 + lane] : 0u;
            before += (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(t), 63);
        }
        if (tid < n)
            s_out[tid] = before + inc - v;
        if (tid == 255)
            s_out[n] = before + inc; // wavefront 3 has seen every counter
    }
    __syncthreads();
    return s_out[n];
    }
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t v = tid < n ? s_in[tid] : 0u;
    const uint32_t inc = wave_scan_incl(v);
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

template <bool ONEBAR>
__device__ __forceinline__ uint32_t rx_scan512(const uint32_t *s_in, uint32_t *s_out, int n, uint32_t *s_wave, const int tid)
{
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (ONEBAR) {
    (void)s_wave;
    if (wave < 4) {
        const int c0 = wave * 128 + lane * 2;
        const uint2 p = rx_pair(s_in, c0, n);
        const uint32_t v = p.x + p.y;
        const uint32_t inc = wave_scan_incl(v);
        uint32_t before = 0;
        for (int x = 0; x < wave; ++x) {
            const uint2 t = rx_pair(s_in, x * 128 + lane * 2, n);
            before += (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(t.x + t.y), 63);
        }
        const uint32_t ex = before + inc - v;
        if (c0 < n)
            s_out[c0] = ex;
        if (c0 + 1 < n)
            s_out[c0 + 1] = ex + p.x;
        if (tid == 255)
            s_out[n] = before + inc; // wavefront 3 has seen every counter
    }
    __syncthreads();
    return s_out[n];
    }
    uint2 p = make_uint2(0u, 0u);
    uint32_t v = 0, inc = 0;
    const int c0 = wave * 128 + lane * 2;
    if (wave < 4) {
        p = rx_pair(s_in, c0, n);
        v = p.x + p.y;
        inc = wave_scan_incl(v);
        if (lane == 63)
            s_wave[wave] = inc;
    }
    __syncthreads();
    const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (wave < 4) {
        uint32_t base = 0;
        for (int x = 0; x < wave; ++x)
            base += s_wave[x];
        const uint32_t ex = base + inc - v;
        if (c0 < n)
            s_out[c0] = ex;
        if (c0 + 1 < n)
            s_out[c0 + 1] = ex + p.x;
    }
    if (tid == 0)
        s_out[n] = total;
    __syncthreads();
    return total;
}

// Diagnostic builds (-DRX_PHASE_TIMERS): thread 0 of every workgroup sums the shader-clock cycles it spends in
// each phase of a pass into statistics slots 4.. (kmm_get_param "stats_slot_<n>"); product builds compile to nothing.
#ifdef RX_PHASE_TIMERS
#define RX_PT_DECL unsigned long long pt_acc[7] = {0, 0, 0, 0, 0, 0, (unsigned long long)clock64()}
#define RX_PT_ARG , pt_acc
#define RX_PT_ARG2 , pt_acc
#define RX_PT(slot)                                                                                                   \
    do {                                                                                                              \
        if (threadIdx.x == 0) {                                                                                       \
            const unsigned long long pt_now = (unsigned long long)clock64();                                                            \
            pt_acc[slot] += pt_now - pt_acc[6];                                                                       \
            pt_acc[6] = pt_now;                                                                                       \
        }                                                                                                             \
    } while (0)
#define RX_PT_END(iv, base)                                                                                           \
    do {                                                                                                              \
        if (threadIdx.x == 0)                                                                                         \
            for (int pt_i = 0; pt_i < 6; ++pt_i)                                                                      \
                atomicAdd(&(iv).stats[(size_t)(blockIdx.x % KMM_STAT_SHARDS) * KMM_STAT_STRIDE + (base) + pt_i],     \
                          pt_acc[pt_i]);                                                                              \
    } while (0)
#else
#define RX_PT_DECL
#define RX_PT_ARG
#define RX_PT_ARG2 , nullptr
#define RX_PT(slot)
#define RX_PT_END(iv, base)
#endif

// The thread index rebuilt from the wavefront's number (a scalar the kernel derives once) and the lane number the
// hardware counts, in volatile asm: never hoisted out of the block loop, hence never kept alive across it — pass 1 had
// its thread index spilled to scratch, and each reload behind a barrier waited for vmcnt(0): for the prefetched tile
// and for the previous block's copy-out stores.
__device__ __forceinline__ int rx_tid_now(int wave)
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return (wave << 6) | l;
}

// Key of a slot that holds no k-mer: any value >= F; the lane's own spare counter behind the real ones.
__device__ __forceinline__ uint32_t rx_spare_key()
{
    return (uint32_t)RX_MAXF + 1u + (threadIdx.x & 63);
}

// Counting sort of the workgroup's k-mers (RX_KPT per thread) inside LDS.  prep(i) finalises slot i of q (it may
// rewrite q[i]) and returns its key < F, or rx_spare_key() for a slot that holds no k-mer.  Then: the sorted run array goes to
// `out` as one contiguous coalesced copy, where each key's run starts (and the total) to dir_row[0..F].  sbuf may
// hold the inputs: they are in registers before anything is written.
struct RxNoHook {
    __device__ __forceinline__ void operator()() const {}
};

// `mid` runs between the ranking and the scan (pass 2 issues the next item's descriptor loads there).
// Barriers: one after the ranking, one or two in the scan (ONEBAR = pass 1's flavour: one-barrier scan and
// non-temporal copy-out), one after the placement.  ENDBAR = false leaves out the one after the copy-out:
// the caller then guarantees a barrier of its own before sbuf is written again and before the next call's ranking
// (which needs the counters this call clears during its copy-out).
template <int RB, bool ENDBAR, bool ONEBAR, int NT = RX_NT, bool NTSTORE = ONEBAR, typename PrepFn,
          typename MidFn = RxNoHook, int WAVESCAN = 0, bool NOWTID = false, typename PostFn = RxNoHook>
__device__ __forceinline__ void rx_sort_emit(uint64_t (&q)[RX_B / NT], PrepFn prep, int F, uint64_t *sbuf,
                                             uint32_t *s_cnt, uint32_t *s_base, uint32_t *s_wave,
                                             uint64_t *__restrict__ out, uint16_t *__restrict__ dir_row,
                                             unsigned long long *pt_acc = nullptr, MidFn mid = MidFn(),
                                             int n_slots = RX_B / NT, int wave_s = -1, PostFn post = PostFn(),
                                             uint64_t *probe_out = nullptr, int probe_den = 0)
{
    // `post` runs after the placement, before its barrier (pass 1 stages the NEXT tile's codes there: that barrier then
    // also publishes them and the tile's own barrier goes)
    // NOWTID: the thread index is rebuilt from the wavefront's number wave_s (uniform, a scalar register) behind every
    // barrier (rx_tid_now) instead of living in a register across the whole sort
    auto tid_now = [&]() { return NOWTID ? rx_tid_now(wave_s) : (int)threadIdx.x; };
    // n_slots (uniform): only the slots [0, n_slots) of q can hold a k-mer (a caller whose k-mers are packed towards
    // the low slots skips the ranking and placement of the rest: k_rx_p2f after its filter)
    constexpr int KPT = RX_B / NT; // k-mers per thread
    int tid = tid_now();
    (void)pt_acc;
    // s_cnt[0..F] is zero on entry: cleared by the caller before its first call (followed by a barrier) and by every
    // call for the next one, right after the scan has consumed the counts
    // rank inside the key's run: one returning LDS atomic per k-mer, RB in flight before the first result is
    // consumed; RB at a time also bounds the registers
    // the key computation (pass 1: a 64-bit division by the modulo) holds at once
    uint32_t cr[KPT]; // key << 16 | rank
#pragma unroll
    for (int h = 0; h < KPT; h += RB) {
        if (h >= n_slots) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
                cr[h + i] = 0xFFFF0000u; // key beyond every fan-out: not placed
            continue;
        }
        uint32_t ck[RB], rk[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i)
            ck[i] = prep(h + i);
        // (prep returns rx_spare_key() for a slot without a k-mer: a spare counter of the slot's own lane — with
        // one shared spare counter the ~20 % of windows that cross a read boundary serialise every atomic
        // instruction on one LDS address)
#pragma unroll
        for (int i = 0; i < RB; ++i)
            rk[i] = atomicAdd(&s_cnt[ck[i]], 1u); // (unconditional: skipping the spare keys' atomics measured no gain)
#pragma unroll
        for (int i = 0; i < RB; ++i)
            cr[h + i] = (ck[i] << 16) | (rk[i] & 0xFFFFu); // (a spare counter is never cleared: its rank means nothing)
    }
    __syncthreads();
    RX_PT(2); // keys + ranks
    mid();
    uint32_t total;
    if (WAVESCAN == 2 || (WAVESCAN == 1 && F <= 128)) { // (2: the caller guarantees F <= 128 — no other form is compiled)
        // Fan-outs up to 128: EVERY wavefront scans the counters for itself (two per lane, six DPP additions) and keeps
        // the bases in registers — a k-mer's base comes from lane key / 2 through the LDS crossbar (ds_bpermute) instead
        // of an LDS table the whole workgroup would have to wait for at a barrier.
        const int lane = tid & 63;
        const uint2 p = rx_pair(s_cnt, lane * 2, F);
        const uint32_t inc = wave_scan_incl(p.x + p.y);
        const uint32_t ex0 = inc - p.x - p.y;
        total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (tid < 64) {
            if (2 * lane <= F)
                dir_row[2 * lane] = (uint16_t)ex0;
            if (2 * lane + 1 <= F)
                dir_row[2 * lane + 1] = (uint16_t)(ex0 + p.x);
            if (lane == 63 && F == 128)
                dir_row[128] = (uint16_t)inc;
        }
        const uint32_t packed = ex0 | (p.x << 16); // (both <= 8192)
#pragma unroll
        for (int h = 0; h < KPT; ++h) {
            if (h >= n_slots)
                break;
            const uint32_t key = cr[h] >> 16;
            const uint32_t pk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((key >> 1) << 2), (int)packed);
            const uint32_t pos = (pk & 0xFFFFu) + ((key & 1u) ? pk >> 16 : 0u) + (cr[h] & 0xFFFFu);
            if (key < (uint32_t)F)
                sbuf[pos] = q[h];
        }
    } else {
    tid = tid_now();
    total = F <= 256 ? rx_scan256<ONEBAR>(s_cnt, s_base, F, s_wave, tid) : rx_scan512<ONEBAR>(s_cnt, s_base, F, s_wave, tid);
    tid = tid_now();
    // The counters are cleared for the next call HERE — every read of them lies before the scan's (last) barrier, and the
    // barrier behind the placement then orders the clear before the next call's ranking.  (Until round 4 they were
    // cleared behind that barrier, during the copy-out; with ENDBAR = false nothing separated the clear from the next
    // block's first ranking atomic in pass 1's packed-tile loop, whose own tile barrier had gone in round 3: a wavefront
    // without clearing duty could rank into a counter a slower wavefront zeroed afterwards.  It took foreign wavefronts
    // on the same SIMDs — the records compaction on a second stream — to make the slow wavefront slow enough: 100-600
    // of 1.2e9 k-mers lost per batch, caught by the conservation self-check; profiles/r04/records_overlap_fault.txt.)
    if (tid <= F)
        s_cnt[tid] = 0;
    if (NT == RX_MAXF && F == NT && tid == 0)
        s_cnt[NT] = 0;
    if (tid <= F)
        dir_row[tid] = (uint16_t)s_base[tid];
    if (NT == RX_MAXF && F == NT && tid == 0) // (fan-out 512 on 512 threads: one more entry than threads)
        dir_row[NT] = (uint16_t)s_base[NT];
    // (one slot after the other: reading all 16 run starts first and then writing — 16 overlapping LDS round trips —
    // measured SLOWER, pass 1 4.41 vs 3.62 ms, pass 2 4.75 vs 4.30 ms; a slot without a k-mer is skipped under
    // predication: writing it to a dummy element instead measured slower in k_rx_p2f, 3.70 vs 3.60 ms)
#pragma unroll
    for (int h = 0; h < KPT; ++h) {
        if (h >= n_slots)
            break;
        const uint32_t pos = s_base[cr[h] >> 16] + (cr[h] & 0xFFFFu);
        if ((cr[h] >> 16) < (uint32_t)F)
            sbuf[pos] = q[h];
    }
    }
    post();
    __syncthreads();
    RX_PT(3); // scan + placement
    tid = tid_now();
    if (WAVESCAN == 2 || (WAVESCAN == 1 && F <= 128)) {
        // (per-wavefront scans read the counters up to here; the callers of this form alternate between two counter
        // arrays, so the next ranking never touches the array cleared now)
        if (tid <= F)
            s_cnt[tid] = 0;
        if (NT == RX_MAXF && F == NT && tid == 0)
            s_cnt[NT] = 0;
    }
    const uint4 *s4 = reinterpret_cast<const uint4 *>(sbuf);
    uint4 *o4 = reinterpret_cast<uint4 *>(out);
    for (uint32_t i = tid; i < (total + 1) / 2; i += NT) {
        if (NTSTORE) { // (pass 1) streamed out past L2: the directory rows the scan kernels read next stay there
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            const uint4 v = s4[i];
            const u32x4_t w = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t *>(o4) + i);
        } else {      // (pass 2: measured slower with non-temporal stores, 4.20 vs 4.15 ms)
            o4[i] = s4[i];
        }
        if (RX_PROBE_ANY && probe_den > 0 && i % (uint32_t)probe_den == 0u) { // (marginal-byte probe: the piece once more, elsewhere)
            if (NTSTORE) {
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                const uint4 v = s4[i];
                const u32x4_t w = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t *>(probe_out) + i / (uint32_t)probe_den);
            } else {
                reinterpret_cast<uint4 *>(probe_out)[i / (uint32_t)probe_den] = s4[i];
            }
        }
    }
    if (ENDBAR)
        __syncthreads();
    RX_PT(4); // copy-out
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

// Exclusive prefix of one value per thread over the RX_NT-thread workgroup; *total = sum.  One barrier: the
// wavefront totals go to one of two LDS rows in turn (`flip`, uniform, toggled here), so that a wavefront still
// reading the previous call's totals is never overtaken — the call in between has its own barrier.
__device__ __forceinline__ uint32_t rx_scan_threads(uint32_t v, uint32_t (*s_wave8)[RX_NT / 64], uint32_t &flip,
                                                    uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_scan_incl(v);
    uint32_t *row = s_wave8[flip];
    flip ^= 1u;
    if (lane == 63)
        row[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int x = 0; x < RX_NT / 64; ++x) {
        const uint32_t t = row[x];
        base += x < wave ? t : 0u;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

// Runs of k-mers that lie contiguous in HBM are copied by LPR-lane copiers, one 8-byte load per lane.  So that no
// load waits for another, every run is first cut into pieces listed in LDS, and the cuts are made at LPR-element
// boundaries of the BUFFER (LPR = 16: 128-byte lines), not of the run: every load instruction of a copier then
// touches one line (tools/chunk_read_bench.hip: 4.2 vs 3.85 TB/s useful on 256-byte runs, 4.6 vs 3.8 on 344-byte
// runs).  A piece = aligned start `sub_src`, first lane a, n lanes: lane l of the copier holds element sub_src + l
// of the buffer, for a <= l < a + n; sub_meta = dst << 12 | a << 6 | n (dst = where the piece's first k-mer goes,
// if the caller needs that; < 2^20).  This thread's run (len k-mers from element offset src) owns the piece
// indices [pre, pre + rx_n_pieces(len, src)) (pre from rx_scan_threads) and writes those that fall into the window
// [win, win + RX_SUBCAP).
// LINECUT = false cuts every LPR k-mers from the run's start instead (fewer, fuller pieces): pass 3, whose probes
// cost as much as its loads, is faster that way (3.19 vs 3.56 ms), pass 2 with the line cuts (4.16 vs 4.28 ms).
template <int LPR, bool LINECUT>
__device__ __forceinline__ uint32_t rx_n_pieces(uint32_t len, uint64_t src)
{
    const uint32_t off = LINECUT ? (uint32_t)src & (uint32_t)(LPR - 1) : 0u;
    return len ? (off + len + LPR - 1) / LPR : 0u;
}

template <int LPR, bool LINECUT>
__device__ __forceinline__ void rx_list_subruns(uint32_t pre, uint32_t len, uint64_t src, uint32_t dst, uint32_t win,
                                                uint32_t *sub_src, uint32_t *sub_meta)
{
    const uint32_t off = LINECUT ? (uint32_t)src & (uint32_t)(LPR - 1) : 0u;
    const uint32_t nsub = rx_n_pieces<LPR, LINECUT>(len, src);
    const uint32_t j0 = pre > win ? pre : win;
    const uint32_t j1 = pre + nsub < win + (uint32_t)RX_SUBCAP ? pre + nsub : win + (uint32_t)RX_SUBCAP;
    for (uint32_t j = j0; j < j1; ++j) {
        const uint32_t p = j - pre;
        const uint32_t a = p ? 0u : off;
        const uint32_t done = p ? p * LPR - off : 0u; // k-mers of the run before this piece
        const uint32_t n = len - done < (uint32_t)LPR - a ? len - done : (uint32_t)LPR - a;
        sub_src[j - win] = (uint32_t)(src - off) + p * LPR; // element offsets of a sub-batch's buffers fit 32 bits (launch_rx)
        sub_meta[j - win] = ((dst + done) << 12) | (a << 6) | n;
    }
}

// The copiers read the list in batches of STEP sub-runs without checking for its end: entries [n, n rounded up to
// STEP) are cleared (0 k-mers from offset 0) before the barrier that publishes the list.  STEP divides RX_SUBCAP.
template <int STEP>
__device__ __forceinline__ void rx_pad_list(uint32_t n, uint32_t *sub_src, uint32_t *sub_meta)
{
    static_assert(RX_SUBCAP % STEP == 0, "a padded list must fit the window");
    const uint32_t n_pad = (n + STEP - 1) / STEP * STEP;
    for (uint32_t i = n + threadIdx.x; i < n_pad; i += RX_NT) {
        sub_src[i] = 0u;
        sub_meta[i] = 0u;
    }
}

// Work distribution of passes 2 and 3.  Work items that read NEIGHBOURING bytes of the previous pass's output
// (runs of adjacent partitions inside one block / item share their first and last 128-byte line) are handed to
// workgroups of ONE XCD at about the same time, so that the shared lines are fetched from HBM once and hit in
// that XCD's L2: the index space is cut into 8 sub-spaces, one counter each; a workgroup works through the
// sub-space of the XCD it really runs on (HW_REG_XCC_ID) and then through the others in turn.  Every index is
// popped exactly once whatever the placement (placement only decides what hits in L2); a workgroup ends after it
// has seen all 8 counters exhausted.  The next index is popped while the current one is being worked on.
__device__ __forceinline__ uint32_t rx_xcc_id()
{
    return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; // HW_REG_XCC_ID, bits [3:0]
}

// next index of sub-space `sub`, or `limit` when it is exhausted (thread 0 only)
__device__ __forceinline__ uint32_t rx_pop(unsigned long long *counters, uint32_t sub, uint32_t limit)
{
    const unsigned long long u = atomicAdd(&counters[sub * 16], 1ull);
    return u < (unsigned long long)limit ? (uint32_t)u : limit;
}

// ------------------------------------------------------------------------------------------------
// pass 1
// ------------------------------------------------------------------------------------------------
template <int MODE, bool RC, bool C2 = false>
__global__ void __launch_bounds__(RX_NT, 4) k_rx_p1(ReadsView rv, const uint64_t *__restrict__ kmers_in, int64_t n_in,
                                                 IndexView iv, RxView rx, int k, int64_t tile_begin, uint32_t n_src)
{
    constexpr bool PACKED = MODE == MODE_PACKED;   // reads of one length, tiles of whole reads (kmm_tile.hpp)
#ifndef RX_P1_STAGE_EARLY
#define RX_P1_STAGE_EARLY 1
#endif
    constexpr bool STAGE_EARLY = PACKED && !RC && RX_P1_STAGE_EARLY != 0; // three barriers per block instead of four
    constexpr int TM = (MODE == MODE_KMERS || PACKED) ? MODE_UNIFORM : MODE;
    constexpr int S = TM == MODE_RECORDS ? 4 : 16; // windows per lane per tile
    constexpr int R = RX_KPT / S;                  // tiles per half-workgroup per block
    using Smem = typename std::conditional<PACKED, TilePackedSmem, TileSmem<S>>::type;
    using Raw = typename std::conditional<PACKED, TilePackedRaw, TileRaw>::type;
    __shared__ Smem sm[2];
    __shared__ uint64_t sbuf[RX_B];
    __shared__ __attribute__((aligned(8))) uint32_t s_cnt[RX_MAXF + 2 + 64];
    __shared__ uint32_t s_base[RX_MAXF + 1], s_wave[4];
#ifndef RX_P1_INCDIV
#define RX_P1_INCDIV 1
#endif
    // Incremental division (r04).  A lane's 16 windows are consecutive: q' = (q - b) / 4 + c 4^(k-1) (b = the base that
    // leaves, c = the one that enters).  With q = quot M + rem and t = quot mod 4:  q - b = (quot - t) M + (rem + t M - b),
    // 4 | quot - t and 4 | q - b, hence 4 | x = rem + t M - b; b = q mod 4 = (rem + t M) mod 4, so x / 4 = (rem + t M) >> 2,
    // and 0 <= x / 4 < M (x <= rem + 3 M < 4 M).  So (q - b) / 4 = (quot >> 2) M + ((rem + t M) >> 2) is a proper
    // (quotient, remainder) pair; adding c 4^(k-1) = Q_c M + R_c gives rem' = x / 4 + R_c < 2 M < 2^32 (one conditional
    // subtraction, carry into the quotient) and quot' = (quot >> 2) + Q_c + carry.  Exact for every modulo < 2^31 and
    // every k: ONE full division (five quarter-rate multiplies) per lane and block instead of 16.  rem + t M needs 33
    // bits: with M = Mh 2^16 + Ml, (rem + t M) >> 2 = ((rem + t Ml) >> 2) + (t Mh << 14) — 24-bit multiplies, full rate.
    // (Q_c, R_c), c = 0..3, sit in LDS; the entering bases come from the lane's register window (bits [2 k, 2 k + 32)), so
    // the table reads do not depend on the chain of (quot, rem).
    // Measured (profiles/r04/ab_pass1_incremental_division.txt): packed tiles 2.268 -> 2.258 ms at configs[2], 2.297 -> 2.261 ms
    // at configs[1] — a fifth fewer VALU issue slots buy nothing: pass 1 is not bound by its vector ALUs.  Position-based
    // tiles (ragged reads) keep the full division: with the chain their 16 ranks spill (2.53 -> 2.86 ms).
    constexpr bool INCDIV = RX_P1_INCDIV != 0 && !RC && MODE == MODE_PACKED;
    __shared__ uint4 s_qr[4]; // {R_c, Q_c low, Q_c high, -}
    const int tid = threadIdx.x, half = tid >> 8, ltid = tid & 255;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6); // (rx_sort_emit rebuilds the thread index from it)
    if (INCDIV && tid < 4) {
        uint32_t r;
        const uint64_t qc = fastdiv_m31((uint64_t)tid << (2 * (k - 1)), (uint32_t)iv.modulo, iv.magic, &r);
        s_qr[tid] = make_uint4(r, (uint32_t)qc, (uint32_t)(qc >> 32), 0u);
    }
    TileConst tc;
    tc.kmask = 0; tc.bmask = 0; tc.aligned = false;
    if (MODE != MODE_KMERS) {
        sm[half].lut[ltid] = rv.lut[ltid];
        tc = tile_const(rv, k);
    }
    constexpr uint32_t X = RC ? 2u : 1u;
    const int F1 = (int)rx.F1;
    uint32_t lookups = 0;
    const int sh = rx.w + rx.f2;
    const uint32_t spare = rx_spare_key();
    RX_PT_DECL;
#ifndef RX_P1_NO_PREFETCH
    constexpr bool PREFETCH = R == 1 && MODE != MODE_KMERS;
#else
    constexpr bool PREFETCH = false;
#endif
    Raw pw[R];
    if (tid <= F1)
        s_cnt[tid] = 0; // (rx_sort_emit)
    if (tid == 0)
        s_cnt[RX_NT] = 0;
    __syncthreads(); // also: the code table is in LDS
    auto load_tile = [&](int64_t tile, Raw &raw) {
        if constexpr (PACKED)
            tile_packed_load<C2>(rv, tile, ltid, raw);
        else
            tile_load_vec<S, TM, C2>(rv, tc, tile, ltid, raw);
    };
    // Blocks of a workgroup: every gridDim.x-th block of the source, or (RX_P1_CHUNKED) a contiguous range of it.  A plain
    // write stream sustains ~5.8 TB/s with one sequential region per workgroup against 4.6-5.6 for 64 KB tiles handed
    // out round-robin (tools/stream_bench.hip, profiles/r04/stream_bench.txt), but pass 1 does not notice: 2.251 vs
    // 2.269 ms at configs[2], whatever the grid (profiles/r04/ab_pass1_chunked_blocks.txt) — its stores are not what it
    // waits for.  Round-robin stays.
#ifndef RX_P1_CHUNKED
#define RX_P1_CHUNKED 0
#endif
    const uint32_t sb_per = (n_src + gridDim.x - 1) / gridDim.x;
    const uint32_t sb_first = RX_P1_CHUNKED ? blockIdx.x * sb_per : blockIdx.x;
    const uint32_t sb_end = RX_P1_CHUNKED ? (sb_first + sb_per < n_src ? sb_first + sb_per : n_src) : n_src;
    const uint32_t sb_step = RX_P1_CHUNKED ? 1u : gridDim.x;
    for (uint32_t sb = sb_first; sb < sb_end; sb += sb_step) {
        uint64_t q[RX_KPT];
        uint32_t valid = 0;
        TileWin win;
        win.lo = win.hi = 0;
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
            if constexpr (STAGE_EARLY) {
                // packed tiles: this block's codes were staged in LDS behind the previous block's placement (its barrier
                // published them); the next block's bytes are requested now and staged behind this block's placement
                const int64_t tile = tile_begin + ((int64_t)sb * 2 + half);
                if (sb == sb_first) {
                    load_tile(tile, pw[0]);
                    tile_packed_stage<C2>(rv, tile, sm[half], ltid, pw[0]);
                    __syncthreads();
                }
                valid = tile_packed_fetch(rv, tc, tile, sm[half], q, ltid, &win);
                if (sb + sb_step < sb_end)
                    load_tile(tile_begin + ((int64_t)(sb + sb_step) * 2 + half), pw[0]);
            } else {
            // the block's staged bytes: all tiles' loads are in flight together; flat reads (one tile per half):
            // the NEXT block's bytes are requested before this block is sorted, so their latency hides behind it
            if (!PREFETCH || sb == sb_first) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    load_tile(tile_begin + ((int64_t)sb * 2 + half) * R + r, pw[r]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                uint64_t qq[S];
                const int64_t tile = tile_begin + ((int64_t)sb * 2 + half) * R + r;
                uint32_t v;
                // (the first tile of a block needs no opening barrier: the previous block's sort lies in between)
                if constexpr (PACKED)
                    v = tile_packed_kmers<false, C2>(rv, tc, tile, k, sm[half], qq, ltid, pw[r], &win);
                else
                    v = r == 0 ? tile_kmers<S, TM, false, C2>(rv, tc, tile, k, sm[half], qq, ltid, pw[r], &win)
                               : tile_kmers<S, TM, true, C2>(rv, tc, tile, k, sm[half], qq, ltid, pw[r]);
#pragma unroll
                for (int j = 0; j < S; ++j)
                    q[r * S + j] = qq[j];
                valid |= v << (r * S);
            }
            if (PREFETCH && sb + sb_step < sb_end) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    load_tile(tile_begin + ((int64_t)(sb + sb_step) * 2 + half) * R + r, pw[r]);
            }
            }
        }
        lookups += (uint32_t)__popc(valid) * X;
        RX_PT(0); // front end: loads + k-mer windows
        // the only division by the modulo of the whole path (mapper.pyx:54) happens here
        if (RC) {
            uint64_t x[RX_KPT];
            auto fwd = [&](int i) {
                uint32_t c;
                x[i] = rx_pack(iv, sh, q[i], &c);
                return ((valid >> i) & 1u) ? c : spare;
            };
            rx_sort_emit<RX_RB1, true, true, RX_NT, true, decltype(fwd), RxNoHook, 0, true>(
                x, fwd, F1, sbuf, s_cnt, s_base, s_wave, rx.buf1 + (size_t)sb * 2 * RX_B,
                rx.start1 + (size_t)sb * 2 * (size_t)(F1 + 1) RX_PT_ARG2, RxNoHook(), RX_KPT, wave_s);
            auto rev = [&](int i) {
                uint32_t c;
                x[i] = rx_pack(iv, sh, revcomp(q[i], k), &c);
                return ((valid >> i) & 1u) ? c : spare;
            };
            rx_sort_emit<RX_RB1, MODE == MODE_KMERS, true, RX_NT, true, decltype(rev), RxNoHook, 0, true>(
                x, rev, F1, sbuf, s_cnt, s_base, s_wave, rx.buf1 + ((size_t)sb * 2 + 1) * RX_B,
                rx.start1 + ((size_t)sb * 2 + 1) * (size_t)(F1 + 1) RX_PT_ARG2, RxNoHook(), RX_KPT, wave_s);
        } else {
            // (quot, rem) of the lane's current window; INCDIV: slot i follows from slot i - 1 (prep is called in slot order)
            uint32_t qlo = 0, qhi = 0, rem = 0;
            const uint32_t M32 = (uint32_t)iv.modulo, Ml = M32 & 0xFFFFu, Mh = M32 >> 16;
            const uint32_t nb = (uint32_t)((win.lo >> (2 * k)) | (win.hi << (64 - 2 * k))); // the bases entering windows 1, 2, ...
            uint4 e_nx = make_uint4(0u, 0u, 0u, 0u);
            auto fwd = [&](int i) {
                if constexpr (INCDIV && R == 1) {
                    const uint4 e = e_nx;
                    if (i + 1 < RX_KPT) {
                        // (Q_c, R_c) of the NEXT step, requested one step ahead.  The empty asm ties the request to this
                        // step's input: without it the scheduler hoists all 15 table reads to the top (45 registers) and
                        // the ranks of the sort spill — a scratch reload in the block loop waits for vmcnt(0).
                        uint32_t c = (nb >> (2 * i)) & 3u;
                        asm("" : "+v"(c) : "v"(rem));
                        e_nx = s_qr[c];
                    }
                    if (i == 0) {
                        const uint64_t quot = fastdiv_m31(win.lo & tc.kmask, M32, iv.magic, &rem);
                        qlo = (uint32_t)quot;
                        qhi = (uint32_t)(quot >> 32);
                    } else {
                        const uint32_t t = qlo & 3u;
                        uint32_t th = __umul24(t, Mh);
                        asm("" : "+v"(th)); // (keeps t Mh a 24-bit product: folded into t (Mh << 14) it is a quarter-rate 32-bit multiply)
                        const uint32_t y = ((__umul24(t, Ml) + rem) >> 2) + (th << 14); // (rem + t M) >> 2 < M
                        const uint32_t r2 = y + e.x;                                      // < 2 M < 2^32
                        // quot = (quot >> 2) + Q_c + (r2 >= M): the carry chain by hand (the compiler builds 64-bit
                        // register pairs for each of the two additions)
                        uint32_t nlo, nhi; // (fresh registers: the compiler keeps every window's quotient until it packs)
                        asm("v_alignbit_b32 %0, %3, %2, 2\n\t"
                            "v_lshrrev_b32 %1, 2, %3\n\t"
                            "v_cmp_le_u32 vcc, %6, %7\n\t"
                            "v_addc_co_u32 %0, vcc, %0, %4, vcc\n\t"
                            "v_addc_co_u32 %1, vcc, %1, %5, vcc"
                            : "=&v"(nlo), "=&v"(nhi)
                            : "v"(qlo), "v"(qhi), "v"(e.y), "v"(e.z), "s"(M32), "v"(r2)
                            : "vcc");
                        qlo = nlo;
                        qhi = nhi;
                        rem = r2 - M32 < r2 ? r2 - M32 : r2; // (unsigned: r2 - M32 wraps above r2 when r2 < M32)
                    }
                    q[i] = ((((uint64_t)qhi << 32) | qlo) << sh) | (uint64_t)(rem & ((1u << sh) - 1u));
                    return ((valid >> i) & 1u) ? rem >> sh : spare;
                } else {
                    uint32_t c;
                    q[i] = rx_pack(iv, sh, q[i], &c);
                    return ((valid >> i) & 1u) ? c : spare;
                }
            };
            auto post = [&]() { // behind the placement: the next block's codes into LDS (the sort's barrier publishes them)
                if constexpr (STAGE_EARLY) {
                    if (sb + sb_step < sb_end)
                        tile_packed_stage<C2>(rv, tile_begin + ((int64_t)(sb + sb_step) * 2 + half), sm[half], ltid, pw[0]);
                }
            };
            rx_sort_emit<RX_RB1, MODE == MODE_KMERS, true, RX_NT, true, decltype(fwd), RxNoHook, 0, true, decltype(post)>(
                q, fwd, F1, sbuf, s_cnt, s_base, s_wave, rx.buf1 + (size_t)sb * RX_B,
                rx.start1 + (size_t)sb * (size_t)(F1 + 1) RX_PT_ARG2, RxNoHook(), RX_KPT, wave_s, post,
                RX_PROBE_P1W ? rx.probe + (size_t)sb * RX_B : nullptr, RX_PROBE_P1W);
        }
    }
#ifdef RX_PT_P1
    RX_PT_END(iv, 10);
#endif
    rx_stat_add(iv, 0, lookups);
    rx_stat_add(iv, KMM_STAT_RX_P1, lookups); // conservation check (drain): = gathered by pass 2 = probed by pass 3 (+ dropped)
}

// ------------------------------------------------------------------------------------------------
// directory scan between pass 1 and pass 2
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rx_colsum(RxView rx)
{
    const uint32_t F1 = rx.F1;
    const uint32_t b0 = blockIdx.x * RX_CH;
    const uint32_t b1 = b0 + RX_CH < rx.NB ? b0 + RX_CH : rx.NB;
    const size_t ld = F1 + 1;
    for (uint32_t c = threadIdx.x; c < F1; c += 256) {
        const uint16_t *p = rx.start1 + (size_t)b0 * ld + c;
        uint32_t sum = 0, b = b0;
        for (; b + 8 <= b1; b += 8, p += 8 * ld) { // independent loads first: the loop is latency-bound
            uint32_t lo[8], hi[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                lo[u] = p[u * ld];
                hi[u] = p[u * ld + 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                sum += hi[u] - lo[u];
        }
        for (; b < b1; ++b, p += ld)
            sum += (uint32_t)p[1] - (uint32_t)p[0];
        rx.csum[(size_t)blockIdx.x * F1 + c] = sum;
    }
}

// exclusive prefix of one value per thread over a 256-thread workgroup; *total gets the sum
__device__ __forceinline__ uint32_t scan256_excl(uint32_t v, uint32_t *s_wave4, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_scan_incl(v);
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

// One workgroup per coarse partition: its chunk sums -> exclusive chunk offsets (in place), the partition total.
__global__ void __launch_bounds__(256) k_rx_chunkscan(RxView rx, uint32_t n_chunks)
{
    __shared__ uint32_t s_wave4[4];
    const uint32_t c = blockIdx.x, F1 = rx.F1;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_chunks; base += 256) {
        const uint32_t ch = base + threadIdx.x;
        const uint32_t v = ch < n_chunks ? rx.csum[(size_t)ch * F1 + c] : 0u;
        uint32_t tot;
        const uint32_t ex = scan256_excl(v, s_wave4, &tot);
        if (ch < n_chunks)
            rx.csum[(size_t)ch * F1 + c] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        rx.T1[c] = carry;
        rx.P1T[(size_t)c * (rx.NB + 1) + rx.NB] = carry;
    }
}

// exclusive prefix of one value per thread over a 512-thread workgroup; *total gets the sum
__device__ __forceinline__ uint32_t scan512_excl(uint32_t v, uint32_t *s_wave8, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_scan_incl(v);
    __syncthreads(); // s_wave8 may still be read by the previous call
    if (lane == 63)
        s_wave8[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int x = 0; x < 8; ++x) {
        base += x < wave ? s_wave8[x] : 0u;
        tot += s_wave8[x];
    }
    *total = tot;
    return base + inc - v;
}

// One workgroup: the item table (items of RX_B k-mers per coarse partition) and the pass-3 row table.
__global__ void __launch_bounds__(512) k_rx_tables(RxView rx)
{
    static_assert(RX_MAXF <= 512, "one thread per coarse partition");
    __shared__ uint32_t s_wave8[8];
    const uint32_t c = threadIdx.x, F1 = rx.F1;
    const uint32_t run = c < F1 ? rx.T1[c] : 0u;
    const uint32_t n_items = c < F1 ? (run + RX_B - 1) / RX_B : 0u;
    const uint32_t rows = (n_items + RX_IC - 1) / RX_IC;
    uint32_t tot_items, tot_rows;
    const uint32_t ib = scan512_excl(n_items, s_wave8, &tot_items);
    const uint32_t wb = scan512_excl(rows, s_wave8, &tot_rows);
    uint32_t jmax = n_items;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t o = __shfl_xor(jmax, d);
        jmax = o > jmax ? o : jmax;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        s_wave8[threadIdx.x >> 6] = jmax;
    __syncthreads();
    if (c < F1) {
        rx.item_base[c] = ib;
        rx.work_base[c] = wb;
    }
    if (c == 0) {
        uint32_t m = s_wave8[0];
        for (int x = 1; x < 8; ++x)
            m = s_wave8[x] > m ? s_wave8[x] : m;
        rx.item_base[F1] = tot_items;
        rx.work_base[F1] = tot_rows;
        rx.ctrl[0] = tot_items;
        rx.ctrl[1] = tot_rows;
        rx.ctrl[2] = m;
    }
}

// Prefix of every coarse partition's runs over the blocks of one chunk.  Thread c walks column c of the directory
// rows (coalesced across c); the results leave through an LDS tile of 32 blocks so that each partition's 32
// consecutive values are written as one 128-byte (P1T) / 64-byte (S1T) piece.  (Round 4 tried the stores straight from
// registers instead — a thread holds 32 consecutive entries of its row: 16-byte stores, no LDS, no barrier — and it was
// slower, the more so the shorter the pieces: directory scans 0.624 -> 0.675 ms at 32 blocks per step, 0.825 at 16,
// 0.976 at 8 (profiles/r04/ab_colscan_direct_stores.txt): the kernel is bound by the granularity of its writes.)
__global__ void __launch_bounds__(256) k_rx_colscan(RxView rx)
{
    constexpr int TB = 32, CW = 256; // blocks per tile; coarse partitions per sweep (F1 > 256: two sweeps = grid.y)
    __shared__ uint32_t tP[CW][TB + 1];
    __shared__ uint16_t tS[CW][TB + 2];
    const uint32_t F1 = rx.F1, NB = rx.NB;
    const uint32_t b0 = blockIdx.x * RX_CH;
    const uint32_t b1 = b0 + RX_CH < NB ? b0 + RX_CH : NB;
    const size_t ld = F1 + 1;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    { // (grid.y = sweeps of 256 coarse partitions: the walk is latency-bound, so the sweeps run side by side)
    const uint32_t cb = blockIdx.y * CW;
    const uint32_t c = cb + threadIdx.x;
    const uint32_t nc = F1 - cb < (uint32_t)CW ? F1 - cb : (uint32_t)CW;
    uint32_t run = c < F1 ? rx.csum[(size_t)blockIdx.x * F1 + c] : 0u;
    const uint32_t ib = c < F1 ? rx.item_base[c] : 0u;
    for (uint32_t t0 = b0; t0 < b1; t0 += TB) {
        const uint32_t nb = b1 - t0 < (uint32_t)TB ? b1 - t0 : (uint32_t)TB;
        if (c < F1) {
            const uint16_t *p = rx.start1 + (size_t)t0 * ld + c;
            uint32_t lo[TB], hi[TB];
#pragma unroll
            for (int u = 0; u < TB; ++u) { // independent loads first: the walk is latency-bound
                lo[u] = (uint32_t)u < nb ? p[u * ld] : 0u;
                hi[u] = (uint32_t)u < nb ? p[u * ld + 1] : 0u;
            }
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                const uint32_t cnt = hi[u] - lo[u];
                tP[threadIdx.x][u] = run;
                tS[threadIdx.x][u] = (uint16_t)lo[u];
                if (cnt) { // items whose first k-mer lies in this run
                    uint32_t m = (run + RX_B - 1) / RX_B;
                    while ((uint64_t)m * RX_B < (uint64_t)run + cnt) {
                        rx.item_desc[ib + m] = make_uint2(t0 + u, c);
                        ++m;
                    }
                }
                run += cnt;
            }
        }
        __syncthreads();
        // two partitions per wavefront instruction, 32 consecutive blocks each
        const uint32_t bi = lane & 31;
        for (uint32_t cc = wave * 2 + (lane >> 5); cc < nc; cc += 8) {
            if (bi < nb) {
                rx.P1T[(size_t)(cb + cc) * (NB + 1) + t0 + bi] = tP[cc][bi];
                rx.S1T[(size_t)(cb + cc) * NB + t0 + bi] = tS[cc][bi];
            }
        }
        __syncthreads();
    }
    }
}

// start2 [item][F2 + 1] -> start2T [F2 + 1][max_items], RX_TR2 items per workgroup through an LDS tile
constexpr int RX_TR2 = 32;
__global__ void __launch_bounds__(256) k_rx_tr2(RxView rx)
{
    __shared__ uint16_t tile[RX_TR2][RX_MAXF + 2];
    const uint32_t n_items = rx.ctrl[0], cols = rx.F2 + 1;
    const uint32_t i0 = blockIdx.x * RX_TR2;
    if (i0 >= n_items)
        return;
    const uint32_t ni = n_items - i0 < (uint32_t)RX_TR2 ? n_items - i0 : (uint32_t)RX_TR2;
    for (uint32_t e = threadIdx.x; e < ni * cols; e += 256)
        tile[e / cols][e % cols] = rx.start2[(size_t)i0 * cols + e];
    __syncthreads();
    const uint32_t it = threadIdx.x % RX_TR2, c0 = threadIdx.x / RX_TR2;
    if (it < ni)
        for (uint32_t col = c0; col < cols; col += 256 / RX_TR2)
            rx.start2T[(size_t)col * rx.max_items + i0 + it] = tile[it][col];
}

// ------------------------------------------------------------------------------------------------
// pass 2
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RX_NT, 4) k_rx_p2(IndexView iv, RxView rx)
{
    __shared__ uint64_t sbuf[RX_B];
    __shared__ uint32_t sub_src[RX_SUBCAP];
    __shared__ uint32_t sub_meta[RX_SUBCAP], s_wave8[2][RX_NT / 64];
    __shared__ __attribute__((aligned(8))) uint32_t s_cnt[RX_MAXF + 2 + 64];
    __shared__ uint32_t s_base[RX_MAXF + 1], s_wave[4];
    __shared__ uint32_t s_idx;
    const int tid = threadIdx.x, grp = tid / RX_LPR, lg = tid % RX_LPR;
    const uint32_t NB = rx.NB;
    uint32_t gathered = 0; // conservation check (kmm_get_param "radix_p2_kmers"): must equal pass 1's lookups
    uint32_t scan_flip = 0;
    const uint32_t spare = rx_spare_key();
    const int F2 = (int)rx.F2;
    // XCD x takes the coarse partitions [x cs, (x+1) cs), items j-major: the workgroups of one XCD work on item j
    // of ~cs adjacent coarse partitions together, whose runs are neighbours inside every pass-1 block
    const uint32_t cs = (rx.F1 + 7u) / 8u, limit = rx.ctrl[2] * cs;
    const uint32_t home = rx_xcc_id();
    if (tid <= F2)
        s_cnt[tid] = 0; // (rx_sort_emit; the barriers of the first index broadcast lie before the first use)
    if (tid == 0)
        s_cnt[RX_NT] = 0;
    // An item's description (uniform) and this thread's run descriptor of its first round.  Both are loaded while
    // the PREVIOUS item is being sorted: the two dependent round trips (item table, then run descriptors) would
    // otherwise stand at the head of every item (measured: 27 % of the pass).
    struct Item {
        uint32_t valid, c, item, b0, lo, hi;
    };
    struct RunDesc {
        uint32_t vs, ve, st;
    };
    auto describe = [&](uint32_t sub, uint32_t idx) {
        Item d;
        d.valid = 0; d.c = 0; d.item = 0; d.b0 = 0; d.lo = 0; d.hi = 0;
        if (idx >= limit)
            return d;
        const uint32_t cj = idx / cs, cc = sub * cs + idx % cs;
        if (cc >= rx.F1)
            return d; // F1 is no multiple of 8
        const uint32_t ib = rx.item_base[cc];
        if (cj >= rx.item_base[cc + 1] - ib)
            return d; // partition sizes differ
        d.valid = 1;
        d.c = cc;
        d.item = ib + cj;
        d.b0 = rx.item_desc[d.item].x;
        d.lo = cj * RX_B;
        const uint32_t Tc = rx.T1[cc];
        d.hi = Tc - d.lo < (uint32_t)RX_B ? Tc : d.lo + RX_B;
        return d;
    };
    auto run_desc = [&](const Item &d, uint32_t bb) {
        RunDesc r;
        r.vs = 0xFFFFFFFFu; r.ve = 0xFFFFFFFFu; r.st = 0;
        const uint32_t b = bb + tid;
        if (d.valid && b < NB) {
            const uint32_t *P = rx.P1T + (size_t)d.c * (NB + 1);
            r.vs = P[b];
            r.ve = P[b + 1];
            r.st = rx.S1T[(size_t)d.c * NB + b];
        }
        return r;
    };
    RX_PT_DECL;
    for (uint32_t turn = 0; turn < 8u; ++turn) {
    const uint32_t sub = (home + turn) & 7u;
    uint32_t nxt = limit;
    if (tid == 0)
        nxt = rx_pop(rx.queue, sub, limit);
    if (tid == 0)
        s_idx = nxt;
    __syncthreads();
    uint32_t idx = s_idx;
    __syncthreads();
    if (idx >= limit)
        continue;
    if (tid == 0)
        nxt = rx_pop(rx.queue, sub, limit); // the next index is popped while this one is being worked on
    Item it = describe(sub, idx);
    RunDesc rd = run_desc(it, it.b0);
    for (;;) {
        RX_PT(0); // waiting for the work item
        const uint32_t lo = it.lo, hi = it.hi, n = hi - lo;
        if (it.valid) {
            // gather, in rounds of RX_NT runs: (A) one thread per run has its descriptor (contiguous in b), the
            // runs are cut into sub-runs of <= RX_LPR k-mers listed in LDS; (B) copier g copies sub-runs g,
            // g + RX_NG, ... with RX_U2 loads in flight per lane, none depending on another
            for (uint32_t bb = it.b0;; bb += RX_NT) {
                if (bb != it.b0)
                    rd = run_desc(it, bb); // (an item of more than RX_NT runs: tiny runs only)
                const uint32_t b = bb + tid;
                const uint32_t vs = rd.vs < hi ? rd.vs : hi;
                const bool live = vs < hi;
                const uint32_t ve = live ? rd.ve : vs;
                const uint32_t st = live ? rd.st : 0u;
                const uint32_t from = vs > lo ? vs : lo, to = ve < hi ? ve : hi;
                const uint32_t len = live && to > from ? to - from : 0u;
                const uint64_t src = (uint64_t)b * RX_B + st + (from - vs); // element offset of the run's first wanted k-mer
                // (the last thread's "my run does not end the item" rides on the scan's total as bit 31)
                uint32_t n_sub;
                const uint32_t last_more = (tid == RX_NT - 1 && live && ve < hi) ? 0x80000000u : 0u;
                const uint32_t pre = rx_scan_threads(rx_n_pieces<RX_LPR, true>(len, src) | last_more, s_wave8, scan_flip, &n_sub);
                const bool more = n_sub >> 31;
                n_sub &= 0x7FFFFFFFu;
                RX_PT(1); // run descriptors, scan
                for (uint32_t win = 0; win < n_sub; win += RX_SUBCAP) {
                    rx_list_subruns<RX_LPR, true>(pre, len, src, from - lo, win, sub_src, sub_meta);
                    const uint32_t nw = n_sub - win < (uint32_t)RX_SUBCAP ? n_sub - win : (uint32_t)RX_SUBCAP;
                    rx_pad_list<RX_NG * RX_U2>(nw, sub_src, sub_meta);
                    __syncthreads();
                    for (uint32_t j0 = grp; j0 < nw; j0 += RX_NG * RX_U2) {
                        uint64_t x[RX_U2];
                        uint32_t meta[RX_U2], so[RX_U2];
#pragma unroll
                        for (int u = 0; u < RX_U2; ++u) {
                            meta[u] = sub_meta[j0 + u * RX_NG];
                            so[u] = sub_src[j0 + u * RX_NG];
                        }
                        // (lanes past the sub-run's end stay masked: letting them re-read its first k-mer, as pass
                        // 3 does, measured 4 % slower here: 4.65 vs 4.47 ms)
#pragma unroll
                        for (int u = 0; u < RX_U2; ++u) {
                            const uint32_t rel = (uint32_t)lg - ((meta[u] >> 6) & 63u);
                            x[u] = rel < (meta[u] & 63u) ? RX_LOAD2(rx.buf1 + ((size_t)so[u] + lg)) : 0ull;
                        }
#pragma unroll
                        for (int u = 0; u < RX_U2; ++u) {
                            const uint32_t rel = (uint32_t)lg - ((meta[u] >> 6) & 63u);
                            if (rel < (meta[u] & 63u))
                                sbuf[(meta[u] >> 12) + rel] = x[u];
                        }
                    }
                    if (more || win + RX_SUBCAP < n_sub)
                        __syncthreads(); // the list is rewritten by the next window / round (after the last one the
                                         // barrier of the index exchange below follows anyway)
                }
                if (!more)
                    break; // the round's last run ends the item (or lies beyond it)
            }
        }
        // the next item: index now (popped during the gather), description while the k-mers are ranked, run
        // descriptors while they are placed and copied out
        if (tid == 0)
            s_idx = nxt;
        __syncthreads(); // also: the gather's LDS writes are complete
        const uint32_t idx_n = s_idx;
        if (tid == 0 && idx_n < limit)
            nxt = rx_pop(rx.queue, sub, limit);
        const Item it_n = describe(sub, idx_n);
        RunDesc rd_n;
        rd_n.vs = 0xFFFFFFFFu; rd_n.ve = 0xFFFFFFFFu; rd_n.st = 0;
        if (it.valid) {
            uint64_t q[RX_KPT];
            uint32_t valid = 0;
#pragma unroll
            for (int i = 0; i < RX_KPT; ++i) {
                const uint32_t e = i * RX_NT + tid;
                q[i] = 0;
                if (e < n) {
                    q[i] = sbuf[e];
                    valid |= 1u << i;
                }
            }
            gathered += (uint32_t)__popc(valid);
            RX_PT(5); // sub-run list, gather into LDS and back into registers
            auto fine = [&](int i) {
                return ((valid >> i) & 1u) ? ((uint32_t)(q[i] >> rx.w) & (uint32_t)(F2 - 1)) : spare;
            };
            auto mid = [&]() { rd_n = run_desc(it_n, it_n.b0); };
            rx_sort_emit<RX_RB2, false, false>(q, fine, F2, sbuf, s_cnt, s_base, s_wave, rx.buf2 + (size_t)it.item * RX_B,
                            rx.start2 + (size_t)it.item * (F2 + 1) RX_PT_ARG2, mid);
        } else {
            rd_n = run_desc(it_n, it_n.b0);
            __syncthreads(); // s_idx is read by everyone before it is written again
        }
        if (idx_n >= limit)
            break;
        idx = idx_n;
        it = it_n;
        rd = rd_n;
    }
    }
#ifndef RX_PT_P1
    RX_PT_END(iv, 10);
#endif
    rx_stat_add(iv, 2, gathered);
}

// ------------------------------------------------------------------------------------------------
// pass 2 with the empty-bucket filter (k_rx_p2f)
// ------------------------------------------------------------------------------------------------
// 80 % of the k-mers of a read set are not in the index, and at load factor 0.5 three fifths of those fall into an
// EMPTY bucket.  Pass 2 is the first place where a k-mer's bucket range is narrow enough to test that from LDS: a
// coarse partition of 2^19 buckets has a 64 KB occupancy bitmap.  A k-mer whose bucket is empty cannot match
// anything (mapper.pyx:55-58: n_local_hits == 0), so it is dropped here instead of being written by pass 2 and read
// again by pass 3 (16 B of HBM traffic each).  The bitmap leaves no room for a second workgroup per CU (measured:
// the plain pass 2 with one 512-thread workgroup per CU takes 5.96 ms instead of 4.05), so this kernel is built
// differently from k_rx_p2:
//   * ONE workgroup of 1024 threads per CU; the memory phase of item j + 1 overlaps the LDS phase of item j inside
//     the workgroup: the k-mers of the next item are requested (8 per thread, into registers) BEFORE the current
//     item is sorted, and are consumed (tested against the bitmap, the survivors ranked and placed) after; three
//     barriers per item (after ranking, scan, placement): the table of item j + 2 is built next to the scan;
//   * gather by k-mer, not by run: wavefront y takes the item's k-mers [512 y, 512 y + 512); the run a k-mer lies in
//     comes from a bit mask of the item's run starts (mbcnt), or a walk / binary search over the LDS table of run
//     starts; consecutive lanes read consecutive k-mers, every lane of every load instruction is used whatever the
//     run length (piece lists: 52 % at 17-k-mer runs), no list is built, scanned or padded;
//   * a work unit = P2F_K consecutive items of one coarse partition: one bitmap load per unit.
// Output: item slot as in k_rx_p2 (one output item per input item), holding the survivors only.
// bit i of the result = bit 2 i | bit 2 i + 1 of x (16 bits) / = the OR of bits 4 i .. 4 i + 3 (8 bits)
__device__ __forceinline__ uint32_t rx_fold2(uint32_t x)
{
    x = (x | (x >> 1)) & 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0x0000FFFFu;
}
__device__ __forceinline__ uint32_t rx_fold4(uint32_t x)
{
    x = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x11111111u;
    x = (x | (x >> 3)) & 0x03030303u;
    x = (x | (x >> 6)) & 0x000F000Fu;
    return (x | (x >> 12)) & 0xFFu;
}

constexpr int P2F_NT = 1024;
constexpr int P2F_KPT = RX_B / P2F_NT;   // 8 k-mers per thread and item
constexpr int P2F_KMAX = 64;             // most items per work unit (rx.p2f_k: chosen per batch, launch_rx)
constexpr int P2F_LOGBITS = 19;          // buckets per coarse partition the LDS bitmap covers: 2^19 (64 KB)

// FILTER false: coarse partitions beyond 2^21 buckets (no bitmap fits) — the kernel is still the faster pass 2
// FSMALL: at most 128 fine partitions (the launch knows): only the per-wavefront scan is compiled into the sort
template <bool FILTER, bool FSMALL>
__global__ void __launch_bounds__(P2F_NT) k_rx_p2f(IndexView iv, RxView rx)
{
    __shared__ uint32_t s_bits[(1 << P2F_LOGBITS) / 32];
    __shared__ uint64_t sbuf[RX_B];
    __shared__ uint32_t t_vs[P2F_NT + 64]; // run table of the item being requested: where run t starts in the
    __shared__ uint32_t t_off[P2F_NT];    // coarse partition's virtual array; t_off[t] + v = where k-mer v of run t lies
                                          // in pass 1's output, relative to the table's first block
    __shared__ uint16_t t_aux[RX_B / 64 + 1]; // run (table index) of the item's k-mers 0, 64, 128, ...: a wavefront's 64
    __shared__ uint32_t t_last;               // consecutive k-mers lie between two of them; run of the last covered k-mer
    __shared__ uint32_t t_sbits[2][RX_B / 32]; // bit e = a run starts at the item's k-mer e (items alternate between the
    __shared__ uint32_t t_empty[2];            // two masks); != 0: an EMPTY run starts inside the item (mask unusable)
    __shared__ __attribute__((aligned(8))) uint32_t s_cnt2[2][RX_MAXF + 2 + 64]; // the sort's counters: items alternate
    __shared__ uint32_t s_base[RX_MAXF + 1], s_wave[4];
    __shared__ uint32_t s_b0[P2F_KMAX];
    __shared__ uint32_t s_idx;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t NB = rx.NB;
    const int F2 = (int)rx.F2, w = rx.w;
    const uint32_t bmask = (1u << (rx.w + rx.f2)) - 1u;          // bucket inside the coarse partition
    const uint32_t gs = (uint32_t)rx.occ_shift;                  // buckets per bitmap bit: 2^gs (coarse partitions of
                                                                 // up to 2^21 buckets: sparse tables such as modulo 452 930 477)
    constexpr bool nofilt = !FILTER;                             // (a template parameter: as a run-time branch around the
                                                                 // bitmap test it cost 13 % — the test's LDS reads no longer
                                                                 // moved ahead of the requests)
    const uint32_t nwords = nofilt ? 0u : (((uint32_t)F2 << w) >> gs) / 32u; // bitmap words of one coarse partition in LDS (<= 16384)
    const uint32_t spare = rx_spare_key();
    uint32_t gathered = 0, dropped = 0; // conservation check: gathered = pass 1's lookups = pass 3's probes + dropped
    const uint32_t cs = (rx.F1 + 7u) / 8u;
    const uint32_t P2F_K = rx.p2f_k;
    const uint32_t limit = ((rx.ctrl[2] + P2F_K - 1u) / P2F_K) * cs; // units of the largest coarse partition x cs
    const uint32_t home = rx_xcc_id();
    if (tid <= F2)
        s_cnt2[0][tid] = s_cnt2[1][tid] = 0; // (rx_sort_emit)
    RX_PT_DECL;

    struct RunDesc {
        uint32_t vs, ve, st;
    };
    for (uint32_t turn = 0; turn < 8u; ++turn) {
    const uint32_t sub = (home + turn) & 7u;
    for (;;) {
        if (tid == 0)
            s_idx = rx_pop(rx.queue, sub, limit);
        __syncthreads();
        const uint32_t idx = s_idx;
        __syncthreads(); // every wavefront has read s_idx before it is written again (the round-2 race)
        if (idx >= limit)
            break;
        const uint32_t cu = idx / cs, cc = sub * cs + idx % cs;
        if (cc >= rx.F1)
            continue; // F1 is no multiple of 8
        const uint32_t ib = rx.item_base[cc], n_items_c = rx.item_base[cc + 1] - ib;
        if (cu * P2F_K >= n_items_c)
            continue; // partition sizes differ
        const uint32_t j0 = cu * P2F_K;
        const uint32_t n_it = n_items_c - j0 < P2F_K ? n_items_c - j0 : P2F_K;
        const uint32_t Tc = rx.T1[cc];
        const uint32_t *P = rx.P1T + (size_t)cc * (NB + 1);
        const uint16_t *S = rx.S1T + (size_t)cc * NB;
        // unit prologue: first blocks of the unit's items, the coarse partition's bitmap
        if ((uint32_t)tid < n_it)
            s_b0[tid] = rx.item_desc[ib + j0 + tid].x;
        if (nofilt) {
        } else if (gs == 0) {
            const uint4 *src = reinterpret_cast<const uint4 *>(rx.occ + (size_t)cc * nwords);
            uint4 *dst = reinterpret_cast<uint4 *>(s_bits);
            for (uint32_t i = tid; i < nwords / 4u; i += P2F_NT)
                dst[i] = src[i];
            for (uint32_t i = (nwords & ~3u) + tid; i < nwords; i += P2F_NT) // (tiny tables)
                s_bits[i] = rx.occ[(size_t)cc * nwords + i];
        } else {
            // the index keeps one bit per bucket; 2 or 4 neighbouring buckets are folded into one LDS bit here
            // (set = one of them holds an entry): 64 or 128 buckets -> one word
            const uint32_t *src = rx.occ + (((size_t)cc * nwords) << gs);
            for (uint32_t i = tid; i < nwords; i += P2F_NT) {
                uint32_t o = 0;
                if (gs == 1) {
                    const uint2 v = *reinterpret_cast<const uint2 *>(src + 2 * (size_t)i);
                    o = rx_fold2(v.x) | (rx_fold2(v.y) << 16);
                } else {
                    const uint4 v = *reinterpret_cast<const uint4 *>(src + 4 * (size_t)i);
                    o = rx_fold4(v.x) | (rx_fold4(v.y) << 8) | (rx_fold4(v.z) << 16) | (rx_fold4(v.w) << 24);
                }
                s_bits[i] = o;
            }
        }
        if (tid < 2 * (RX_B / 32))
            (&t_sbits[0][0])[tid] = 0u;
        if (tid < 2)
            t_empty[tid] = 0u;
        __syncthreads();

        auto item_lo = [&](uint32_t j) { return (j0 + j) * (uint32_t)RX_B; };
        auto item_n = [&](uint32_t j) {
            const uint32_t lo = (j0 + j) * (uint32_t)RX_B;
            return Tc - lo < (uint32_t)RX_B ? Tc - lo : (uint32_t)RX_B;
        };
        auto load_desc = [&](uint32_t bb) {
            RunDesc r;
            const uint32_t b = bb + tid;
            r.vs = 0xFFFFFFFFu; r.ve = 0xFFFFFFFFu; r.st = 0;
            if (b < NB) {
                r.vs = P[b];
                r.ve = P[b + 1];
                r.st = S[b];
            }
            return r;
        };
        // Table of the runs [bb, bb + 1024) for the item [lo, hi).  Besides its own row every thread marks the run of
        // each k-mer lo + 64 a that lies in its run (t_aux), and of the item's last covered k-mer (t_last): the search of
        // a wavefront's 64 consecutive k-mers then starts from two table indices a few runs apart instead of 0 .. 1023.
        auto put_table = [&](const RunDesc &r, uint32_t lo, uint32_t hi, int mb = -1) {
            // (first table of an item, mb = the item's mask: cleared two items ago, behind that item's barriers)
            if (mb >= 0 && r.vs != 0xFFFFFFFFu && r.vs > lo && r.vs < hi) {
                if (r.ve > r.vs)
                    atomicOr(&t_sbits[mb][(r.vs - lo) >> 5], 1u << ((r.vs - lo) & 31u));
                else
                    t_empty[mb] = 1u; // two runs would share the bit
            }
            t_vs[tid] = r.vs;
            t_off[tid] = (uint32_t)tid * RX_B + r.st - r.vs; // (modulo 2^32: the sum with v is < 1025 x 8192)
            if (tid == P2F_NT - 1) {
                t_vs[P2F_NT] = r.ve; // virtual position the table covers up to (all ones: to the partition's end)
                if (r.ve < hi)
                    t_last = P2F_NT - 1;
            }
            const uint32_t from = r.vs > lo ? r.vs : lo, to = r.ve < hi ? r.ve : hi;
            if (r.vs != 0xFFFFFFFFu && from < to) {
                for (uint32_t a = (from - lo + 63u) >> 6; lo + (a << 6) < to; ++a)
                    t_aux[a] = (uint16_t)tid;
                if (to == hi)
                    t_last = (uint32_t)tid;
            }
        };
        // Request the k-mers lo + e of the item whose position lies in [c_lo, c_hi) (the part of the item the table
        // [bb, bb + 1024) covers).  Wavefront y holds the positions e = (8 y + u) 64 + lane, u = 0 .. 7: eight runs of 64
        // consecutive k-mers.  The run of each k-mer by binary search between the runs of its 64-block's first k-mer
        // and of the next block's (t_aux; the nine values a wavefront needs come from ONE LDS read and readlane).
        // Every load is issued unconditionally, with its element index clamped into pass 1's output: no exec-mask
        // bookkeeping (a slot outside [c_lo, c_hi) holds garbage and no bit of the returned mask).  MERGE: slots
        // outside the range keep what an earlier table left in them.
        auto fill = [&](auto merge_tag, uint32_t bb, uint32_t lo, uint32_t n, uint32_t c_lo, uint32_t c_hi,
                        uint64_t (&x)[P2F_KPT], int mb = -1) {
            constexpr bool MERGE = decltype(merge_tag)::value;
            const uint32_t c_end = lo + n < c_hi ? lo + n : c_hi; // positions this table serves: [c_lo, c_end)
            const uint32_t a0 = wave * 8u;
            uint32_t av = 0;
            {
                const uint32_t pa = lo + ((a0 + (uint32_t)lane) << 6);
                const uint32_t ia = a0 + ((uint32_t)lane <= 8u ? (uint32_t)lane : 8u);
                const uint32_t aux = t_aux[ia];
                av = pa < c_lo ? 0u : (pa < c_end ? aux : t_last);
            }
            uint32_t A[P2F_KPT + 1];
#pragma unroll
            for (int i = 0; i <= P2F_KPT; ++i)
                A[i] = (uint32_t)__builtin_amdgcn_readlane((int)av, i);
            uint32_t v[P2F_KPT], pos[P2F_KPT], len_max = 0;
            // the lane's positions lo + e0 + 64 u rise with u: the ones inside [c_lo, c_end) are a window of u
            const uint32_t e0 = (a0 << 6) + (uint32_t)lane;
            auto n_below = [&](uint32_t bound) { // how many u in 0 .. 7 have lo + e0 + 64 u < bound
                const int32_t c = ((int32_t)(bound - lo) - (int32_t)e0 + 63) >> 6;
                return (uint32_t)(c < 0 ? 0 : (c > P2F_KPT ? P2F_KPT : c));
            };
            const uint32_t vmask = ((1u << n_below(c_end)) - 1u) & ~((1u << n_below(c_lo)) - 1u);
#pragma unroll
            for (int u = 0; u < P2F_KPT; ++u) {
                v[u] = lo + e0 + ((uint32_t)u << 6);
                pos[u] = A[u];
                const uint32_t len = A[u + 1] > A[u] ? A[u + 1] - A[u] : 0u;
                len_max = len > len_max ? len : len_max;
            }
            // The item's run-start bits (first table of an item without empty runs): the run of k-mer e = the run of its
            // 64-block's first k-mer (A[u]) + the number of runs that start at the k-mers after it up to e.  The wavefront's
            // 512 bits are 16 words: one LDS read (lane i reads word i), two readlanes per block, mbcnt per k-mer.
            bool by_bits = false;
            if (!MERGE && mb >= 0 && t_empty[mb] == 0u) { // (uniform)
                by_bits = true;
                const uint32_t mw = t_sbits[mb][a0 * 2u + ((uint32_t)lane & 15u)];
#pragma unroll
                for (int u = 0; u < P2F_KPT; ++u) {
                    // starts at the block's positions 1 .. lane (position 0's run is A[u] itself) = the bits of the mask
                    // shifted down by one that lie below the lane
                    const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)mw, 2 * u);
                    const uint32_t mhi = (uint32_t)__builtin_amdgcn_readlane((int)mw, 2 * u + 1);
                    const uint64_t m1 = (((uint64_t)mhi << 32) | mlo) >> 1;
                    pos[u] += __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
                }
                len_max = 0;
            }
            if (!by_bits) {
            // The wavefront's 512 k-mers span the runs A[0] .. A[8] (about 30 of them at 17 k-mers per run).  With fewer
            // than 64: lane t holds the start of run A[0] + t (ONE LDS read per wavefront and item); every run start a
            // 64-block crosses is broadcast (readlane -> scalar register) and compared against the block's 64 positions:
            // a k-mer's run = A[u] + the number of starts at or below it — two vector instructions per crossed start
            // instead of a binary search per k-mer (seven instructions and an LDS round trip per step).  Run starts do
            // not decrease, so empty runs count themselves in.
            const uint32_t span = A[P2F_KPT] > A[0] ? A[P2F_KPT] - A[0] : 0u;
            if (span < 64u) {
                const uint32_t sv = t_vs[A[0] + (uint32_t)lane]; // (the table is padded by 64 entries)
#pragma unroll
                for (int u = 0; u < P2F_KPT; ++u) {
                    for (uint32_t i = A[u] + 1u; i <= A[u + 1]; ++i) { // (uniform trip count; empty when A[u + 1] <= A[u])
                        const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)sv, (int)(i - A[0]));
                        pos[u] += v[u] >= st ? 1u : 0u;
                    }
                }
                len_max = 0; // (the search below has nothing left to do)
            }
            }
            for (uint32_t stp = len_max ? 1u << (31 - __builtin_clz(len_max)) : 0u; stp >= 1u; stp >>= 1) {
#pragma unroll
                for (int u = 0; u < P2F_KPT; ++u) {
                    const uint32_t cand = pos[u] + stp;
                    const uint32_t t = t_vs[cand <= A[u + 1] ? cand : A[u + 1]]; // (stays inside the table)
                    pos[u] = (cand <= A[u + 1] && t <= v[u]) ? cand : pos[u];
                }
            }
            // (uniform base in scalar registers + a 32-bit byte offset per lane: no 64-bit address arithmetic per k-mer;
            // a table's k-mers lie within 1025 blocks = 64 MB of its first block)
            const uint32_t bbs = (uint32_t)__builtin_amdgcn_readfirstlane((int)bb);
            const char *base = reinterpret_cast<const char *>(rx.buf1 + (size_t)bbs * RX_B);
            const size_t left = (size_t)NB * RX_B - (size_t)bbs * RX_B; // elements of pass 1's output from block bb on
            const uint32_t lim = left > 0x1FFFFFFFull ? 0x1FFFFFFFu : (uint32_t)left - 1u;
#pragma unroll
            for (int u = 0; u < P2F_KPT; ++u) {
                uint32_t el = t_off[pos[u]] + v[u];
                el = el < lim ? el : lim; // never leave pass 1's output, whatever the table says
                const uint64_t y = RX_LOAD2(reinterpret_cast<const uint64_t *>(base + (el << 3)));
                x[u] = (!MERGE || ((vmask >> u) & 1u)) ? y : x[u];
#if RX_PROBE_P2R
                if (u % RX_PROBE_P2R == 0) { // (marginal-byte probe: 8 more bytes, half of pass 1's output away, never consumed)
                    const size_t all = (size_t)NB * RX_B, far = ((size_t)bbs * RX_B + el + all / 2) % all;
                    const uint64_t z = RX_LOAD2(rx.buf1 + far);
                    asm volatile("" ::"v"(z));
                }
#endif
            }
            return vmask;
        };

        // One item: xa holds its k-mers (mask vma); the next item's are requested into xb while this one is sorted.
        uint32_t cover = 0, b0 = 0;
        RunDesc rd;
        rd.vs = rd.ve = 0xFFFFFFFFu; rd.st = 0;
        // rounds beyond the first 1024 runs of item j (rare: tiny runs): table by table, into the same slots
        auto more_rounds = [&](uint32_t j, uint64_t (&x)[P2F_KPT], uint32_t &vm) {
            const uint32_t lo = item_lo(j), n = item_n(j), hi = lo + n;
            uint32_t bb = b0 + P2F_NT;
            while (cover < hi) { // (uniform)
                __syncthreads(); // the table's last readers are done
                const RunDesc r2 = load_desc(bb);
                put_table(r2, lo, hi);
                __syncthreads();
                const uint32_t cover2 = t_vs[P2F_NT];
                vm |= fill(std::true_type(), bb, lo, n, cover, cover2, x);
                cover = cover2;
                bb += P2F_NT;
            }
            __syncthreads(); // ... and of this table, before the next item's is written
        };
        // Three barriers per item — after the ranking, after the scan, after the placement:
        //   top      item j's k-mers (requested one item ago) against the bitmap; item j + 1's requests from the table the
        //            previous item's second barrier published
        //   ranking, barrier 1
        //   scan by wavefronts 0 .. 3 | everyone: table of item j + 2 (its descriptors were loaded one item ago), then the
        //            descriptors of item j + 3 are requested; barrier 2 publishes counters' bases AND the table
        //   placement, barrier 3, copy-out (+ the run-start mask item j + 1 has used is cleared for item j + 3)
        // No closing barrier: the counters alternate between two arrays (item j + 1 ranks into the other one, this one is
        // cleared during the copy-out and next used two barriers later), and the sort buffer is next written behind the
        // next item's second barrier, which no wavefront passes before every wavefront has finished this copy-out.
        auto process = [&](uint32_t j, uint64_t (&xa)[P2F_KPT], uint32_t vma, uint64_t (&xb)[P2F_KPT], uint32_t &vmb) {
            const uint32_t item = ib + j0 + j;
            RX_PT(0);
            // item j's k-mers against the bitmap (waits for them): key = fine partition, or the lane's spare counter
            uint32_t keys[P2F_KPT], n_ok = 0;
            if (nofilt) {
#pragma unroll
                for (int u = 0; u < P2F_KPT; ++u) {
                    const uint32_t ok = (vma >> u) & 1u;
                    n_ok += ok;
                    keys[u] = ok ? __builtin_amdgcn_ubfe((uint32_t)xa[u], (uint32_t)w, (uint32_t)rx.f2) : spare;
                }
            } else {
#pragma unroll
                for (int u = 0; u < P2F_KPT; ++u) {
                    const uint32_t xl = (uint32_t)xa[u]; // packed form: the hash bits below the coarse partition number
                    const uint32_t bit = (xl & bmask) >> gs; // are the low w + f2 <= 22 bits
                    const uint32_t ok = (s_bits[bit >> 5] >> (bit & 31u)) & (vma >> u) & 1u;
                    n_ok += ok;
                    keys[u] = ok ? __builtin_amdgcn_ubfe(xl, (uint32_t)w, (uint32_t)rx.f2) : spare;
                }
            }
            gathered += (uint32_t)__popc(vma);
            dropped += (uint32_t)__popc(vma) - n_ok;
            RX_PT(5); // waiting for the requests + filter
            if (j + 1u < n_it) {
                const uint32_t lo1 = item_lo(j + 1u), n1 = item_n(j + 1u);
                b0 = s_b0[j + 1u];
                cover = t_vs[P2F_NT];
                vmb = fill(std::false_type(), b0, lo1, n1, lo1, cover, xb, (int)((j + 1u) & 1u));
                if (cover < lo1 + n1)
                    more_rounds(j + 1u, xb, vmb); // (rare; own barriers; leaves the table free)
            }
            RX_PT(1); // search + requests
            auto fine = [&](int i) { return keys[i]; };
            auto mid = [&]() { // between barrier 1 and the scan: every request of item j + 1 has read the table
                if (j + 2u < n_it) {
                    put_table(rd, item_lo(j + 2u), item_lo(j + 2u) + item_n(j + 2u), (int)(j & 1u));
                    if (j + 3u < n_it)
                        rd = load_desc(s_b0[j + 3u]);
                }
            };
            rx_sort_emit<P2F_KPT, false, true, P2F_NT, false, decltype(fine), decltype(mid), (FSMALL ? 2 : 1)>(
                xa, fine, F2, sbuf, s_cnt2[j & 1u], s_base, s_wave, rx.buf2 + (size_t)item * RX_B,
                rx.start2 + (size_t)item * (F2 + 1) RX_PT_ARG2, mid, P2F_KPT, -1, RxNoHook(),
                RX_PROBE_P2W ? rx.probe + (size_t)item * RX_B : nullptr, RX_PROBE_P2W);
            if (tid < RX_B / 32)
                t_sbits[(j + 1u) & 1u][tid] = 0u; // item j + 1's mask has served (its requests lie before barrier 1); item
            if (tid == 0)                         // j + 3's bits are set behind the next item's barrier 1
                t_empty[(j + 1u) & 1u] = 0u;
        };

        // prime the pipeline: item 0's table and requests, item 1's table, item 2's run descriptors
        uint64_t x0[P2F_KPT], x1[P2F_KPT];
        uint32_t vm0 = 0, vm1 = 0;
        b0 = s_b0[0];
        rd = load_desc(b0);
        put_table(rd, item_lo(0), item_lo(0) + item_n(0), 0);
        if (n_it > 1u)
            rd = load_desc(s_b0[1]);
        __syncthreads();
        cover = t_vs[P2F_NT];
        vm0 = fill(std::false_type(), b0, item_lo(0), item_n(0), item_lo(0), cover, x0, 0);
        __syncthreads(); // every wavefront has finished its searches before the next table is written (without this one a
                         // stale table entry above a k-mer's position made the offset wrap and the request left the
                         // buffer: memory fault, first GPU run)
        if (cover < item_lo(0) + item_n(0))
            more_rounds(0, x0, vm0);
        if (tid < RX_B / 32)
            t_sbits[0][tid] = 0u; // item 0's mask has served: item 2's bits go there
        if (tid == 0)
            t_empty[0] = 0u;
        if (n_it > 1u) {
            put_table(rd, item_lo(1), item_lo(1) + item_n(1), 1);
            if (n_it > 2u)
                rd = load_desc(s_b0[2]);
        }
        __syncthreads(); // item 1's table is published, mask 0 is clear
        for (uint32_t j = 0; j < n_it; j += 2u) {
            process(j, x0, vm0, x1, vm1);
            if (j + 1u < n_it)
                process(j + 1u, x1, vm1, x0, vm0);
        }
    }
    }
#ifndef RX_PT_P1
    RX_PT_END(iv, 10);
#endif
    rx_stat_add(iv, 2, gathered);
    rx_stat_add(iv, KMM_STAT_RX_DROPPED, dropped);
}

// ------------------------------------------------------------------------------------------------
// pass 3
// ------------------------------------------------------------------------------------------------
// DirT: type of the LDS directory; uint16_t (half the LDS: 8192-bucket slices with two workgroups per CU) only when
// no slice of the index holds more than 65535 entries (launch_rx checks).
// P16: the slice's directory comes from rx.pstart16 (two buckets per 32-bit load, already relative to the slice).
template <int WMAX, int ECAP, int WPS, typename DirT, int SUBCAP = RX_SUBCAP3, bool P16 = false>
__global__ void __launch_bounds__(RX_NT, WPS) k_rx_p3(IndexView iv, RxView rx, int max_freq)
{
    static_assert(!P16 || sizeof(DirT) == 2, "the 16-bit directory is copied as it is");
    __shared__ __attribute__((aligned(4))) DirT sdir[WMAX + 2]; // bucket b of the slice holds entries [sdir[b], sdir[b + 1]) - e0
    __shared__ uint64_t skeys[ECAP];
    __shared__ uint32_t scnt[ECAP];
    __shared__ uint32_t sub_list[SUBCAP];
    __shared__ uint32_t s_wb[RX_MAXF + 1], s_wave8[2][RX_NT / 64];
    __shared__ uint32_t s_idx;
    const int tid = threadIdx.x, grp = tid / RX_LPR_P3, lg = tid % RX_LPR_P3;
    const uint32_t n_rows = rx.ctrl[1], F1 = rx.F1, F2 = rx.F2;
    const uint32_t W = 1u << rx.w;
    const uint64_t M = iv.modulo;
    for (uint32_t i = tid; i <= F1; i += RX_NT)
        s_wb[i] = rx.work_base[i];
    // XCD x takes the fine partitions g in [x gs, (x+1) gs) of every row (coarse partition, chunk of items), g
    // fastest: its workgroups read neighbouring runs of the same items together
    const uint32_t gs = (F2 + 7u) / 8u, limit = n_rows * gs;
    unsigned long long *counters = rx.queue + 128;
    const uint32_t home = rx_xcc_id();
    uint32_t hits = 0, probed = 0, scan_flip = 0;
    __syncthreads(); // s_wb is loaded
    // A work item's description (uniform): its fine partition's place in the index and its items.  It is worked
    // out — two dependent loads — while the PREVIOUS work item streams its k-mers.
    struct Slice {
        uint32_t valid, g, e0, ne, it0, n_it, over, tot, fmax;
        uint64_t h0;
    };
    auto describe = [&](uint32_t sub, uint32_t idx) {
        Slice d;
        d.valid = 0; d.g = 0; d.e0 = 0; d.ne = 0; d.it0 = 0; d.n_it = 0; d.h0 = 0; d.over = 0; d.tot = 0; d.fmax = 0xFFFFu;
        if (idx >= limit)
            return d;
        const uint32_t row = idx / gs, g = sub * gs + idx % gs;
        uint32_t c_lo = 0, c_hi = F1; // largest c with s_wb[c] <= row
        while (c_hi - c_lo > 1) {
            const uint32_t mid = (c_lo + c_hi) >> 1;
            if (s_wb[mid] <= row)
                c_lo = mid;
            else
                c_hi = mid;
        }
        const uint32_t c = c_lo;
        const uint32_t f2c = rx.PF - c * F2 < F2 ? rx.PF - c * F2 : F2;
        if (g >= f2c)
            return d; // the last coarse partition holds fewer fine partitions, or F2 < 8
        const uint32_t chunk = row - s_wb[c];
        d.valid = 1;
        d.g = g;
        d.h0 = (uint64_t)(c * F2 + g) << rx.w;
        uint32_t e1;
        if (P16) {
            d.e0 = rx.slice_e0[c * F2 + g];
            e1 = rx.slice_e0[c * F2 + g + 1u];
            d.fmax = rx.slice_fmax ? (uint32_t)rx.slice_fmax[c * F2 + g] : 0xFFFFu;
        } else {
            d.e0 = rx.pstart[d.h0];
            e1 = rx.pstart[d.h0 + W < M ? d.h0 + W : M];
        }
        d.tot = e1 - d.e0;
        d.ne = e1 - d.e0 < (uint32_t)ECAP ? e1 - d.e0 : (uint32_t)ECAP;
        d.over = e1 - d.e0 > (uint32_t)ECAP ? 1u : 0u; // entries beyond the LDS copy: their buckets are walked in HBM
        d.it0 = rx.item_base[c] + chunk * RX_IC;
        const uint32_t it_end = rx.item_base[c + 1];
        d.n_it = it_end - d.it0 < (uint32_t)RX_IC ? it_end - d.it0 : (uint32_t)RX_IC;
        return d;
    };
    RX_PT_DECL;
    for (uint32_t turn = 0; turn < 8u; ++turn) {
    const uint32_t sub = (home + turn) & 7u;
    uint32_t nxt = limit;
    if (tid == 0)
        nxt = rx_pop(counters, sub, limit);
    if (tid == 0)
        s_idx = nxt;
    __syncthreads();
    const uint32_t idx_first = s_idx;
    __syncthreads();
    if (idx_first >= limit)
        continue;
    if (tid == 0)
        nxt = rx_pop(counters, sub, limit); // the next index is popped while this one is being worked on
    Slice sl = describe(sub, idx_first);
    for (;;) {
        RX_PT(0);
        const uint32_t g = sl.g, e0 = sl.e0, ne = sl.ne, it0 = sl.it0, n_it = sl.n_it;
        const uint64_t h0 = sl.h0;
        uint32_t rf[RX_IC / RX_NT], rt[RX_IC / RX_NT];
#pragma unroll
        for (int j = 0; j < RX_IC / RX_NT; ++j)
            rf[j] = rt[j] = 0;
        if (sl.valid) {
        // the slice (directory + keys) and this thread's run descriptors: every load is issued before the first
        // one is consumed
        uint32_t dv[WMAX / RX_NT + 1];
        uint64_t kv[ECAP / RX_NT];
        uint32_t fv[ECAP / RX_NT];
        // (the frequency filter of mapper.pyx:64-66 can only exclude an entry of a slice whose largest frequency lies above
        // the call's threshold: every other slice's frequencies — 2 of its 12 bytes per entry — stay in HBM)
#ifndef RX_P3_FMAX
#define RX_P3_FMAX 1
#endif
        const bool need_f = RX_P3_FMAX == 0 || (int)sl.fmax > max_freq;
        if (P16) { // two buckets per word (W is even from w = 1 on; w = 0: one 16-bit load by thread 0)
            const uint32_t *p32 = reinterpret_cast<const uint32_t *>(rx.pstart16 + h0);
#pragma unroll
            for (int j = 0; j < WMAX / 2 / RX_NT; ++j) {
                const uint32_t i = tid + j * RX_NT;
                dv[j] = i < W / 2u ? p32[i] : 0u;
            }
            if (W < 2u)
                dv[0] = tid == 0 ? (uint32_t)rx.pstart16[h0] : 0u;
        } else {
#pragma unroll
            for (int j = 0; j <= WMAX / RX_NT; ++j) {
                const uint32_t i = tid + j * RX_NT;
                dv[j] = i <= W ? rx.pstart[h0 + i < M ? h0 + i : M] : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < ECAP / RX_NT; ++j) {
            const uint32_t i = tid + j * RX_NT;
            kv[j] = i < ne ? rx.pkeys[(size_t)e0 + i] : 0ull;
            fv[j] = i < ne && need_f ? rx.pfreq[(size_t)e0 + i] : 0u;
        }
        const uint16_t *rfp = rx.start2T + (size_t)g * rx.max_items + it0;
        const uint16_t *rtp = rfp + rx.max_items;
#pragma unroll
        for (int j = 0; j < RX_IC / RX_NT; ++j) {
            const uint32_t i = tid + j * RX_NT;
            rf[j] = i < n_it ? rfp[i] : 0u;
            rt[j] = i < n_it ? rtp[i] : 0u;
        }
        if (P16) {
            uint32_t *sd32 = reinterpret_cast<uint32_t *>(sdir);
#pragma unroll
            for (int j = 0; j < WMAX / 2 / RX_NT; ++j) {
                const uint32_t i = tid + j * RX_NT;
                if (i < W / 2u)
                    sd32[i] = dv[j];
            }
            if (W < 2u && tid == 0)
                sdir[0] = (DirT)dv[0];
            if (tid == 0)
                sdir[W] = (DirT)sl.tot; // (one past the last bucket: the slice's entries)
        } else {
#pragma unroll
            for (int j = 0; j <= WMAX / RX_NT; ++j) {
                const uint32_t i = tid + j * RX_NT;
                if (i <= W)
                    sdir[i] = (DirT)(dv[j] - e0);
            }
        }
#pragma unroll
        for (int j = 0; j < ECAP / RX_NT; ++j) {
            const uint32_t i = tid + j * RX_NT;
            if (i < ne) {
                skeys[i] = kv[j];
                scnt[i] = (int)fv[j] <= max_freq ? 0u : RX_FILTERED; // the frequency filter of mapper.pyx:64-66
            }
        }
        }
        RX_PT(1); // slice + run descriptors loaded, slice written to LDS
        // the next work item: its index (popped at the start of this one) now, its description during the stream
        if (tid == 0)
            s_idx = nxt;
        __syncthreads(); // also orders the slice's LDS writes before the probes
        const uint32_t idx_n = s_idx;
        if (tid == 0 && idx_n < limit)
            nxt = rx_pop(counters, sub, limit);
        const Slice sl_n = describe(sub, idx_n);
        // the entries of bucket [st, st + cn) against q: every matching entry counts (mapper.pyx:57-68; a k-mer
        // present under several nodes is several entries).  Entries beyond the LDS copy are walked in HBM.
        auto probe_bucket_hbm = [&](uint64_t q, uint32_t st, uint32_t cn) {
            for (uint32_t j = 0; j < cn; ++j) {
                const size_t e = (size_t)e0 + st + j;
                if (rx.pkeys[e] == q && (int)rx.pfreq[e] <= max_freq) {
                    atomicAdd(&rx.ecnt[e], 1u);
                    ++hits;
                }
            }
        };
        if (sl.valid) {
        {
            // Piece list of the work item: a run of the partition inside one item is cut every RX_LPR_P3 k-mers; one
            // word per piece = (first element, relative to the work item's first item: < 1024 x 8192) << 5 | k-mers.
            constexpr int NR = RX_IC / RX_NT;
            uint32_t len[NR], np[NR], np_sum = 0, src[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                len[j] = rt[j] - rf[j];
                src[j] = (uint32_t)(tid + j * RX_NT) * (uint32_t)RX_B + rf[j];
                np[j] = (len[j] + (uint32_t)RX_LPR_P3 - 1u) / (uint32_t)RX_LPR_P3;
                np_sum += np[j];
            }
            // (uniform base in scalar registers + a 32-bit byte offset per lane: no 64-bit address arithmetic per k-mer)
            const char *wbase = reinterpret_cast<const char *>(rx.buf2 + (size_t)__builtin_amdgcn_readfirstlane((int)it0) * RX_B);
            uint32_t n_sub;
            const uint32_t pre = rx_scan_threads(np_sum, s_wave8, scan_flip, &n_sub);
            for (uint32_t win = 0; win < n_sub; win += SUBCAP) {
                uint32_t first = pre;
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const uint32_t j0 = first > win ? first : win;
                    const uint32_t j1 = first + np[j] < win + (uint32_t)SUBCAP ? first + np[j] : win + (uint32_t)SUBCAP;
                    for (uint32_t q = j0; q < j1; ++q) {
                        const uint32_t done = (q - first) * (uint32_t)RX_LPR_P3;
                        const uint32_t n = len[j] - done < (uint32_t)RX_LPR_P3 ? len[j] - done : (uint32_t)RX_LPR_P3;
                        sub_list[q - win] = ((src[j] + done) << 5) | n;
                    }
                    first += np[j];
                }
                const uint32_t nw = n_sub - win < (uint32_t)SUBCAP ? n_sub - win : (uint32_t)SUBCAP;
                constexpr uint32_t STEP = RX_NG3 * RX_U;
                static_assert(SUBCAP % STEP == 0, "a padded list must fit the window");
                const uint32_t n_bat = (nw + STEP - 1u) / STEP; // (uniform)
                for (uint32_t i = nw + tid; i < n_bat * STEP; i += RX_NT)
                    sub_list[i] = 0u; // (padding to whole batches: pieces without k-mers)
                __syncthreads(); // (first window: also orders the slice's LDS writes before the probes)
                RX_PT(2); // scan + sub-run list
                // list entries first, then the loads, nothing conditional in between: the RX_U loads of a lane leave
                // back to back (a lane beyond its piece re-reads the piece's first k-mer); returns the mask of the
                // slots that hold a k-mer.  (The list is padded to whole batches: every lane takes every batch.)
                auto request = [&](uint32_t j0, uint64_t (&x)[RX_U]) {
                    uint32_t e[RX_U], act = 0;
#pragma unroll
                    for (int u = 0; u < RX_U; ++u)
                        e[u] = sub_list[j0 + u * RX_NG3];
#pragma unroll
                    for (int u = 0; u < RX_U; ++u) {
                        const bool in = (uint32_t)lg < (e[u] & 31u);
                        x[u] = RX_LOAD3(reinterpret_cast<const uint64_t *>(wbase + (((e[u] >> 5) + (in ? (uint32_t)lg : 0u)) << 3)));
                        act |= (in ? 1u : 0u) << u;
#if RX_PROBE_P3R
                        if (u % RX_PROBE_P3R == 0) { // (marginal-byte probe: pass 1's output is idle by now — 8 bytes of it per slot)
                            const size_t far = ((size_t)it0 * RX_B + (e[u] >> 5) + (uint32_t)lg) % ((size_t)rx.NB * RX_B);
                            const uint64_t z = RX_LOAD3(rx.buf1 + far);
                            asm volatile("" ::"v"(z));
                        }
#endif
                    }
                    return act;
                };
                // probe (mapper.pyx:53-69 on the LDS slice), RX_G3 k-mers side by side so that their LDS round trips
                // overlap: all bucket bounds; then entry j of every bucket, j = 0, 1, ... (a lane whose bucket has no
                // entry j reads key 0 and ignores it) — nothing conditional between the reads
                auto probe = [&](auto over_tag, const uint64_t (&x)[RX_U], uint32_t actm) {
                    constexpr bool OVER = decltype(over_tag)::value; // (uniform per work item: the check leaves the common loop)
#pragma unroll
                    for (int g0 = 0; g0 < RX_U; g0 += RX_G3) {
                        uint32_t st[RX_G3], cn[RX_G3];
#pragma unroll
                        for (int i = 0; i < RX_G3; ++i) {
                            const uint32_t hb = (uint32_t)x[g0 + i] & (W - 1u); // packed form: bucket = low w bits
                            st[i] = sdir[hb];
                            cn[i] = sdir[hb + 1];
                        }
                        uint32_t mx = 0;
#pragma unroll
                        for (int i = 0; i < RX_G3; ++i) {
                            const bool act = (actm >> (g0 + i)) & 1u;
                            probed += act ? 1u : 0u;
                            cn[i] = act ? cn[i] - st[i] : 0u;
                            if constexpr (OVER) {
                                if (cn[i] && st[i] + cn[i] > ne) { // (rare) entries beyond the LDS copy
                                    probe_bucket_hbm(x[g0 + i], st[i], cn[i]);
                                    cn[i] = 0;
                                }
                            }
                            mx = cn[i] > mx ? cn[i] : mx;
                        }
                        // One loop over j < the longest of the group's buckets: ~4.2 trips per wavefront (the longest of its
                        // 64 x RX_G3 buckets) for 1.3 entries per bucket.  Round 4 measured what the trips behind the second
                        // cost — capping the loop (wrong counts) takes pass 3 from 3.00 ms to 2.37 / 2.43 / 2.73 / 2.91 at 1 /
                        // 2 / 3 / 4 trips (configs[2], 20 M reads) — and then every way of doing that work differently, all
                        // bit-exact, none faster: entries 0-1 (0-2, 0-3) unrolled and the rest walked by the owning lane per
                        // group 2.95 (2.93, 2.97); the same with the walk once per batch of RX_U k-mers 3.29; the long
                        // buckets' entries queued per wavefront in LDS and probed 64 at a time 3.01; idle lanes issuing
                        // no read from the second trip on 3.01.  The 5 % of probes that meet a bucket of three or more entries
                        // cost ~15 % of the pass whichever lanes do them: it is their SIMD slots, not the loop's form
                        // (profiles/r04/ab_pass3_entry_loop.txt).
                        for (uint32_t j = 0; j < mx; ++j) {
                            uint64_t key[RX_G3];
#pragma unroll
                            for (int i = 0; i < RX_G3; ++i)
                                key[i] = skeys[j < cn[i] ? st[i] + j : 0u];
#pragma unroll
                            for (int i = 0; i < RX_G3; ++i)
                                if (j < cn[i] && key[i] == x[g0 + i])
                                    atomicAdd(&scnt[st[i] + j], 1u);
                        }
                    }
                };
                auto stream = [&](auto over_tag) {
                    for (uint32_t bi = 0; bi < n_bat; ++bi) {
                        uint64_t x[RX_U];
                        const uint32_t m = request(grp + bi * STEP, x);
                        probe(over_tag, x, m);
                    }
                };
                if (sl.over)
                    stream(std::true_type());
                else
                    stream(std::false_type());
                __syncthreads();
                RX_PT(3); // streaming + probing
            }
        }
        // LDS counters -> per-entry count vector (entries the frequency filter excludes carry RX_FILTERED); the last
        // window's closing barrier (or, without any window, the scan's) has completed the counters
        for (uint32_t i = tid; i < ne; i += RX_NT) {
            const uint32_t cn = scnt[i];
            if (cn - 1u < RX_FILTERED - 1u) {
                atomicAdd(&rx.ecnt[(size_t)e0 + i], cn);
                hits += cn;
            }
        }
        }
        RX_PT(4); // LDS counters -> ecnt
        __syncthreads(); // the flush has read the counters; s_idx has been read by everyone
        if (idx_n >= limit)
            break;
        sl = sl_n;
    }
    }
    RX_PT_END(iv, 4);
    rx_stat_add(iv, 1, hits);
    rx_stat_add(iv, 3, probed); // conservation check ("radix_p3_kmers")
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
    auto one = [&](uint64_t e, uint32_t cn, uint32_t node) {
        if (cn) {
            if (ecnt_acc)
                ecnt_acc[e] += cn; // per-k-mer counting mode: what GpuCounter's table holds (gpu_counter.py:29-34)
            agg_add_n(iv, agg, node, cn);
        }
    };
    // four entries per lane and step: counts and nodes are requested together (one round trip per step; the nodes of
    // entries without hits are read for nothing, 4 bytes each), the counts are cleared with whole 16-byte stores
    const uint64_t n4 = n / 4, gid = (uint64_t)blockIdx.x * 256 + threadIdx.x, stride = (uint64_t)gridDim.x * 256;
    uint4 *c4 = reinterpret_cast<uint4 *>(ecnt);
    const uint4 *n4p = reinterpret_cast<const uint4 *>(pnodes);
    for (uint64_t i = gid; i < n4; i += stride) {
        const uint4 c = c4[i];
        const uint4 nd = n4p[i];
        if (c.x | c.y | c.z | c.w) {
            c4[i] = make_uint4(0u, 0u, 0u, 0u);
            one(4 * i, c.x, nd.x);
            one(4 * i + 1, c.y, nd.y);
            one(4 * i + 2, c.z, nd.z);
            one(4 * i + 3, c.w, nd.w);
        }
    }
    for (uint64_t e = n4 * 4 + gid; e < n; e += stride) {
        const uint32_t cn = ecnt[e];
        if (cn) {
            ecnt[e] = 0;
            one(e, cn, pnodes[e]);
        }
    }
    __syncthreads();
    agg_flush_counts(iv, agg);
}

// ------------------------------------------------------------------------------------------------
// flush in node order.  counts[node[e]] += ecnt[e] over entries in bucket order is ~S scattered atomics (24 G/s on
// this chip: 4 ms for 10^8 entries).  With the entries listed in NODE order once per index (norder[j] = entry,
// nnode[j] = its node, non-decreasing) the scattered side becomes a gather of 4-byte counts (55 G requests/s) and
// the atomics walk the count vector front to back; entries of one node are neighbours and are summed inside the
// wavefront first (node skew costs one atomic per wavefront and node).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rx_node_hist(const uint32_t *__restrict__ pnodes, uint64_t n, uint32_t *__restrict__ hist)
{
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (uint64_t)gridDim.x * 256)
        atomicAdd(&hist[pnodes[e]], 1u);
}

__global__ void __launch_bounds__(256) k_rx_node_scatter(const uint32_t *__restrict__ pnodes, uint64_t n,
                                                         uint32_t *__restrict__ cursor, uint32_t *__restrict__ norder,
                                                         uint32_t *__restrict__ nnode)
{
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (uint64_t)gridDim.x * 256) {
        const uint32_t nd = pnodes[e];
        const uint32_t j = atomicAdd(&cursor[nd], 1u);
        norder[j] = (uint32_t)e;
        nnode[j] = nd;
    }
}

__global__ void __launch_bounds__(256) k_rx_flush_sorted(IndexView iv, const uint32_t *__restrict__ ecnt,
                                                         const uint32_t *__restrict__ norder,
                                                         const uint32_t *__restrict__ nnode, uint64_t n)
{
    constexpr int U = 4; // independent gathers in flight per lane
    const int lane = threadIdx.x & 63;
    const uint64_t n_round = (n + 255) / 256 * 256; // whole workgroups stay in the loop: shuffles need all lanes
    for (uint64_t j0 = ((uint64_t)blockIdx.x * U) * 256 + threadIdx.x; j0 < n_round; j0 += (uint64_t)gridDim.x * U * 256) {
        uint32_t nd[U], c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t j = j0 + (uint64_t)u * 256;
            nd[u] = j < n ? nnode[j] : 0xFFFFFFFFu;
            c[u] = j < n ? ecnt[norder[j]] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // segmented inclusive sum over the wavefront's lanes (segments = equal nodes, which are contiguous)
            uint32_t sum = c[u];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t os = __shfl_up(sum, d), on = __shfl_up(nd[u], d);
                if (lane >= d && on == nd[u])
                    sum += os;
            }
            const uint32_t next = __shfl_down(nd[u], 1);
            if ((lane == 63 || next != nd[u]) && sum) // last lane of its segment holds the segment's sum
                atomicAdd(&iv.counts[nd[u]], sum);
        }
    }
}

// cuts[t] = first position j of the node-ordered entry list with nnode[j] >= bounds[t] (lane t; binary search)
__global__ void k_rx_node_cuts(const uint32_t *__restrict__ nnode, uint64_t n, const uint32_t *__restrict__ bounds, int n_bounds,
                               unsigned long long *__restrict__ cuts)
{
    for (int t = threadIdx.x; t < n_bounds; t += blockDim.x) {
        const uint32_t b = bounds[t];
        uint64_t lo = 0, hi = n;
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if (nnode[mid] < b)
                lo = mid + 1;
            else
                hi = mid;
        }
        cuts[t] = lo;
    }
}

// most entries in one slice of 2^w buckets
__global__ void __launch_bounds__(256) k_rx_max_slice(const uint32_t *__restrict__ pstart, uint64_t modulo, int w,
                                                      uint32_t PF, unsigned long long *out)
{
    // out[0] = most entries of one slice, out[1] / out[2] = slices with more than RX_ECAP / RX_ECAP_MID entries
    uint32_t m = 0, over = 0, over_mid = 0;
    for (uint64_t f = (uint64_t)blockIdx.x * 256 + threadIdx.x; f < PF; f += (uint64_t)gridDim.x * 256) {
        const uint64_t h0 = f << w, h1 = h0 + (1ull << w) < modulo ? h0 + (1ull << w) : modulo;
        const uint32_t c = pstart[h1] - pstart[h0];
        m = c > m ? c : m;
        over += c > (uint32_t)RX_ECAP ? 1u : 0u;
        over_mid += c > (uint32_t)RX_ECAP_MID ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t o = __shfl_xor(m, d);
        m = o > m ? o : m;
        over += __shfl_xor(over, d);
        over_mid += __shfl_xor(over_mid, d);
    }
    if ((threadIdx.x & 63) == 0) {
        if (m)
            atomicMax(out, (unsigned long long)m);
        if (over)
            atomicAdd(out + 1, (unsigned long long)over);
        if (over_mid)
            atomicAdd(out + 2, (unsigned long long)over_mid);
    }
}

// raw entry k-mers (bucket order) -> packed form for the current (w, f2)
__global__ void k_rx_pack_keys(const uint64_t *__restrict__ raw, uint64_t n, IndexView iv, int sh,
                               uint64_t *__restrict__ packed)
{
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t c;
        packed[e] = rx_pack(iv, sh, raw[e], &c);
    }
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

// largest frequency among the entries of every slice: one wavefront per slice
__global__ void __launch_bounds__(256) k_rx_slice_fmax(const uint32_t *__restrict__ slice_e0, const uint16_t *__restrict__ pfreq,
                                                       uint32_t PF, uint16_t *__restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t f = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); f < PF; f += (uint64_t)gridDim.x * 4) {
        uint32_t m = 0;
        for (uint64_t e = (uint64_t)slice_e0[f] + lane; e < slice_e0[f + 1]; e += 64) {
            const uint32_t v = pfreq[e];
            m = v > m ? v : m;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const uint32_t o = __shfl_xor(m, d);
            m = o > m ? o : m;
        }
        if (lane == 0)
            out[f] = (uint16_t)m;
    }
}

// occupancy bitmap of the bucket directory: bit h = bucket h holds an entry (the array is zeroed first)
// pstart16[h] = pstart[h] - pstart[first bucket of h's slice] for h < PF << w (buckets beyond the modulo: the slice's
// entry count), slice_e0[f] = pstart[f << w] for f <= PF.  Only called when no slice holds more than 65535 entries.
__global__ void __launch_bounds__(256) k_rx_pstart16(const uint32_t *__restrict__ pstart, uint64_t modulo, int w, uint32_t PF,
                                                     uint16_t *__restrict__ out, uint32_t *__restrict__ slice_e0)
{
    const uint64_t n = (uint64_t)PF << w;
    for (uint64_t h = (uint64_t)blockIdx.x * 256 + threadIdx.x; h < n; h += (uint64_t)gridDim.x * 256) {
        const uint64_t f = h >> w, hs = f << w;
        const uint32_t base = pstart[hs < modulo ? hs : modulo];
        out[h] = (uint16_t)(pstart[h < modulo ? h : modulo] - base);
        if (h == hs)
            slice_e0[f] = base;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        slice_e0[PF] = pstart[modulo];
}

__global__ void __launch_bounds__(256) k_rx_build_occ(const uint32_t *__restrict__ pstart, uint64_t modulo, uint32_t *__restrict__ occ)
{
    for (uint64_t wd = (uint64_t)blockIdx.x * 256 + threadIdx.x; wd * 32 < modulo; wd += (uint64_t)gridDim.x * 256) {
        uint32_t bits = 0;
        const uint64_t h0 = wd * 32;
        uint32_t prev = pstart[h0];
        for (uint32_t i = 0; i < 32 && h0 + i < modulo; ++i) {
            const uint32_t nx = pstart[h0 + i + 1];
            bits |= (nx > prev ? 1u : 0u) << i;
            prev = nx;
        }
        occ[wd] = bits;
    }
}

// 64-bit sum of n uint32 values (one atomic per wavefront)
__global__ void __launch_bounds__(256) k_sum_u32(const uint32_t *__restrict__ v, uint64_t n, unsigned long long *out)
{
    unsigned long long s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        s += v[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0 && s)
        atomicAdd(out, s);
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
