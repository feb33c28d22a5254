// kmm.hip — MI355X (gfx950 / CDNA4) kernels and C-ABI host code for kmer_mapper's hot path:
// reads -> 2-bit codes -> rolling k-mer pack -> modulo hash -> bucket gather -> compare/filter ->
// per-node atomic counts.  See include/kmm.h for the boundary and DESIGN.md for the layout.
//
// Reference semantics restated (never copied): kmer_mapper/mapper.pyx:53-69 (lookup),
// kmer_mapper/util.py:71-75 (extraction), kmer_mapper/command_line_interface.py:41 (N->A).
//
// Integer / gather work: no MFMA.  The bound is random HBM accesses, so the kernels are built for
// memory-level parallelism (U independent bucket gathers in flight per lane, then U independent
// first-entry gathers) at high occupancy; reads are staged through LDS as packed 2-bit codes so
// each byte is fetched from HBM exactly once with 16-byte coalesced loads.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kmm.h"

namespace {

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// Nothing C++ throws may cross the C ABI (std::vector / std::function allocate; a thread may fail to start): the entry points
// that run host-side machinery go through this.
template <typename F>
static int guarded(const char *what, F body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(KMM_ERR_NOMEM, "%s: out of host memory", what);
    } catch (const std::exception &e) {
        return fail(KMM_ERR_INTERNAL, "%s: %s", what, e.what());
    } catch (...) {
        return fail(KMM_ERR_INTERNAL, "%s: unexpected exception", what);
    }
}

#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(KMM_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                 \
                        hipGetErrorString(e_));                                                    \
    } while (0)

#define KMMCHK(expr)                                                                               \
    do {                                                                                           \
        int r_ = (expr);                                                                           \
        if (r_ != KMM_OK)                                                                          \
            return r_;                                                                             \
    } while (0)

#include "kmm_probe.hpp"
#include "kmm_tile.hpp"
#include "kmm_records.hpp"
#include "kmm_kernels.hpp"
#include "kmm_radix.hpp"
#include "kmm_build.hpp"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

int ensure(DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p)
        return KMM_OK;
    if (b.p) {
        HIPCHK(hipFree(b.p)); // blocks until the device is idle: safe w.r.t. in-flight kernels
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(e == hipErrorOutOfMemory ? KMM_ERR_NOMEM : KMM_ERR_HIP,
                    "hipMalloc(%zu bytes) -> %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return KMM_OK;
}

void release(DevBuf &b)
{
    if (b.p)
        (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

bool is_device_ptr(const void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError(); // unregistered host memory: clear the sticky error
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

uint64_t magic_for(uint64_t m)
{
    if (m <= 1)
        return ~0ull;
    return (uint64_t)(((unsigned __int128)1 << 64) / m);
}

void default_lut(uint8_t lut[256])
{
    memset(lut, 0xFF, 256);
    lut['A'] = lut['a'] = 0;
    lut['C'] = lut['c'] = 1;
    lut['G'] = lut['g'] = 2;
    lut['T'] = lut['t'] = 3;
    lut['N'] = lut['n'] = 0; // command_line_interface.py:41
}

struct Stage {
    DevBuf bases, offsets, tile_first, start_bits, kmers, lut, aux;
    hipEvent_t done = nullptr; // the last kernel that read this stage has finished
    bool used = false;
};

constexpr unsigned long long NO_BAD = ~0ull;

} // namespace

#include "kmm_comm.hpp"
#include "kmm_hostpack.hpp"
#include "kmm_gpu_inflate.hpp"

// Page-locked buffers are expensive to make (hipHostMalloc: ~50 ms per GB) and cheap to keep: the ones a handle gives up go
// to a process-wide shelf (at most 8 GiB), and kmm_host_reserve puts buffers there ahead of time — from another thread,
// while the index is uploaded — so that a one-shot `kmer_mapper map` does not pay for them inside its map phase.
struct PinnedShelf {
    std::mutex m;
    std::vector<std::pair<uint8_t *, size_t>> free_list;
    size_t bytes = 0;
    uint8_t *take(size_t want, size_t *got)
    {
        std::lock_guard<std::mutex> g(m);
        size_t best = free_list.size();
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].second >= want && (best == free_list.size() || free_list[i].second < free_list[best].second))
                best = i;
        if (best == free_list.size())
            return nullptr;
        uint8_t *p = free_list[best].first;
        *got = free_list[best].second;
        bytes -= *got;
        free_list.erase(free_list.begin() + (long)best);
        return p;
    }
    void give(uint8_t *p, size_t n)
    {
        if (!p)
            return;
        {
            std::lock_guard<std::mutex> g(m);
            if (bytes + n <= ((size_t)8 << 30)) {
                free_list.emplace_back(p, n);
                bytes += n;
                return;
            }
        }
        (void)hipHostFree(p);
    }
};
static PinnedShelf g_shelf;
constexpr size_t RING_SLOT = (size_t)16 << 20;
#ifndef KMM_RING_SLOTS
#define KMM_RING_SLOTS 8 // (A/B builds: a ring the size of a batch never makes the packing threads wait)
#endif
constexpr int RING_SLOTS = KMM_RING_SLOTS;


struct TimedEvent {
    hipEvent_t start, stop;
    int kernel_id;
};

// What kmm_map_bgzf knows about a chunk of compressed bytes once they are on their way to HBM: the member chain.
struct BgzfStaged {
    std::vector<unsigned long long> m_off, o_rel; // member starts in the chunk; inflated offsets from the chunk's first member (o_rel[0] = 0)
    uint64_t p = 0;                               // bytes of whole members
    int chain_err = 0;                            // 1: no member at p, 2: implausible ISIZE
    uint32_t bad_isize = 0, bad_ms = 0;
    bool hit_cap = false;                         // the chain was cut at the size limit of a call
    bool staged = false;                          // the bytes went through the page-locked ring (else: not copied at all yet)
    double ms_scan_inside = 0;
};

struct kmm_index {
    int device = 0;
    hipStream_t stream = nullptr;      // kernels
    hipStream_t copy_stream = nullptr; // host -> HBM staging, overlaps the previous kernel
    hipEvent_t copied = nullptr;
    uint4 *buckets = nullptr;
    uint4 *entries = nullptr;
    uint32_t *occ = nullptr;           // occupancy bitmap, only for indexes small enough (see occ_max_bytes)
    bool use_occ = true;
    int occ_shift = 0;                 // log2(bitmap bits per bucket)
    uint32_t bloom_words = 0;          // != 0: occ is a word-blocked Bloom filter
    bool wide = false;                 // 32-byte buckets (chosen when the index is too large for the bitmap)
    bool direct_deferred = false;      // the direct view (buckets, entries) is packed from the radix view on first use
    size_t direct_bytes = 0;           // HBM bytes of the direct view (resident or not yet)
    size_t occ_bytes_plan = 0;         // size of the direct view's pre-filter (0: none)
    int rx_why_not = 0;                // why the radix path is unavailable: 0 it is available, 1 modulo >= 2^31, 2 more
                                       // than 512 x 512 slices / slices too dense for LDS, 3 out of memory, 4 the
                                       // buckets of the index overlap (sum of bucket sizes > n_entries)
    uint32_t *counts = nullptr;
    uint32_t *own_counts_buf = nullptr;
    uint8_t *lut_default = nullptr;
    uint8_t *lut_codes = nullptr;        // codes 0..3 map to themselves: flat reads compacted from raw records (k_rec_scatter)
    unsigned long long *first_bad = nullptr;
    unsigned long long *stats = nullptr;
    unsigned long long *queue = nullptr; // tile counter of the dynamic schedule
    bool dynamic_schedule = true;
    int dyn_chunk = 16;                  // tiles per grab
    uint64_t modulo = 0, magic = 0;
    int64_t n_entries = 0, max_node_id = 0;
    Stage stage[2];
    int cur = 0;
    int n_cu = 256;
    // path selection / radix path state (kmm_radix.hpp)
    int path = 0;         // 0 auto, 1 direct, 2 radix
    int grid_per_cu = 64; // upper bound on workgroups per CU of the grid-stride fused kernel
    bool rx_ok = false;   // the index fits the radix path's fan-out (<= 512 x 512 fine partitions)
    int rx_w = 12, rx_f2 = 0; // log2 buckets per fine partition, log2 fine partitions per coarse one
    uint32_t rx_PF = 1, rx_F1 = 1, rx_F2 = 1;
    bool rx_flush_sorted = true; // "radix_sorted_flush": use the node-ordered entry list for the flush
    uint32_t rx_max_slice = 0; // most entries in one fine partition's slice (for the current part_shift)
    bool rx_fits_small = false, rx_fits_mid = false; // all but one slice in 1000 hold at most RX_ECAP / RX_ECAP_MID entries
    int rx_grid_per_cu = 2;   // persistent workgroups of passes 2 and 3 per CU (1: leave room for another stream's kernels)
    int64_t rx_min_units = 0; // auto: batches of at least this many positions / k-mers take the radix path
    int64_t rx_sub_cap = ((int64_t)1 << 32) - 2 * RX_B; // k-mer slots per sub-batch of the radix path ("radix_sub_batch_kmers")
    int64_t rx_sub_cap_eff = 0;  // > 0: the smaller size an out-of-memory call settled on, tried first by the next calls
    int rx_sub_cap_eff_age = 0;  // calls since then (at 16 the caller's cap is tried again)
    int64_t rx_sub_cap_last = 0; // the size the last radix call ran with ("radix_sub_batch_kmers_effective")
    const uint32_t *dbg_T1 = nullptr, *dbg_item_base = nullptr; // the latest sub-batch's tables (debug_rx_* parameters)
    const uint16_t *dbg_start1 = nullptr;
    uint32_t dbg_F1 = 0, dbg_NB = 0;
    int host_pack_threads = 0;     // "host_pack_threads": reads / raw records in host memory are packed to 2 bits per base by that
                                   // many host threads before they cross PCIe (default: min(16, the process's CPU budget))
    std::unique_ptr<kmm_hostpack::Workers> pack_pool; // the packing threads, asleep between calls
    uint8_t *pack_pinned = nullptr; // page-locked home of a packed records batch (kmm_hostpack.hpp)
    size_t pack_pinned_bytes = 0;
    uint8_t *ring[RING_SLOTS] = {}; // the page-locked staging ring (ensure_ring)
    uint8_t *pack_bits_pinned = nullptr; // ... and of the read-start bitset of packed raw records
    size_t pack_bits_pinned_bytes = 0;
    int64_t host_packed_calls = 0, host_packed_record_calls = 0;
    int64_t host_pack_slice_kb = 0; // "host_pack_slice_kb": raw bytes per slice of the records packer (0: its default)
    // kmm_map_bgzf: BGZF members inflated on the GPU (kmm_gpu_inflate.hpp).  Two sets of buffers in turn (the copy of call
    // i + 1 runs under the kernels of call i); the uncompressed bytes behind a call's last complete record wait in `carry`
    // for the next call.
    DevBuf bgzf_comp[2], bgzf_raw[2], bgzf_meta[2], bgzf_tabs, bgzf_err, bgzf_carry, bgzf_crc, bgzf_status;
    hipEvent_t bgzf_done[2] = {nullptr, nullptr};
    hipEvent_t bgzf_slot_ev[RING_SLOTS] = {}; // copies out of / into the slots of the staging ring
    bool bgzf_used[2] = {false, false};
    int bgzf_cur = 0;
    int64_t bgzf_carry_len = 0;
    BgzfStaged bgzf_pre;         // the NEXT chunk, staged and walked under this chunk's inflate kernel (kmm_map_bgzf_hint_next)
    bool bgzf_pre_valid = false;
    const uint8_t *bgzf_pre_from = nullptr, *bgzf_hint_ptr = nullptr;
    int64_t bgzf_pre_n = 0, bgzf_hint_n = 0;
    int bgzf_pre_buf = 0;
    int64_t bgzf_prestaged_calls = 0;
    int64_t bgzf_head_skip = 0;  // "bgzf_head_skip": inflated bytes of the next NEW_STREAM call's first member that belong to someone else
    int64_t bgzf_tail_stop = -1; // "bgzf_tail_stop": >= 0: of the next LAST_CHUNK call's last member only this many inflated bytes are taken
    int64_t bgzf_calls = 0, bgzf_members = 0;
    int dbg_bgzf_slot_kb = 0;     // test hook ("debug_bgzf_ring_slot_kb"): slot size of kmm_map_bgzf's staging ring (a power of two, >= 4)
    int64_t dbg_rx_buf_limit = 0; // test hook ("debug_rx_buffer_limit"): a pass-1 buffer beyond this many bytes counts as out of memory
    int dbg_rec_copy_stream = 0; // experiments (tools/records_overlap_bisect.py): compaction kernels on the copy stream again,
    int dbg_rec_skip = 0;        // and which of them to leave out (1 count2, 2 scans, 4 scatter, 8 uniform, 16 the large memsets)
    uint64_t rx_S = 0;        // entries in bucket order
    uint32_t *rx_pstart = nullptr;
    uint64_t *rx_pkeys = nullptr;     // packed form for the current (w, f2)
    uint64_t *rx_pkeys_raw = nullptr; // the k-mers themselves (re-packed when part_shift changes)
    uint16_t *rx_pstart16 = nullptr;  // slice-relative 16-bit directory + first entry of every slice (for the current
    uint32_t *rx_slice_e0 = nullptr;  // part_shift; null when a slice holds more than 65535 entries or HBM is short)
    uint16_t *rx_slice_fmax = nullptr; // largest frequency of every slice (with the two above)
    uint16_t *rx_pfreq = nullptr;
    uint32_t *rx_pnodes = nullptr, *rx_porig = nullptr, *rx_ecnt = nullptr, *rx_ecnt_acc = nullptr;
    uint32_t *rx_norder = nullptr, *rx_nnode = nullptr; // entries in node order (k_rx_flush_sorted); absent if memory is short
    int rx_occ_shift = 0;         // k_rx_p2f folds 2^rx_occ_shift buckets into one bit of its LDS bitmap
    uint32_t *rx_occ = nullptr;   // bit h = bucket h holds an entry: pass 2's empty-bucket filter (k_rx_p2f); optional
    bool rx_filter = true;        // "radix_filter": use the filtering pass 2 whenever a coarse partition's bitmap fits LDS
    bool rx_packed = true;        // "radix_packed_tiles": pass 1 on reads of one length takes tiles of whole reads
    bool ecnt_dirty = false;  // rx_ecnt holds hits that are not in `counts` yet
    bool rx_unchecked = false; // radix passes have run since the conservation counters were last compared (drain)
    DevBuf rx_buf1, rx_buf2, rx_meta, rx_probe; // (rx_probe: scratch of the marginal-byte probe builds, kmm_radix.hpp)
    // deferred device-side error, sticky until kmm_reset_counts
    int sticky_rc = KMM_OK;
    std::string sticky_msg;
    uint64_t map_calls = 0;   // sequence number of map calls on this handle (error reports name the call)
    uint64_t n_radix_batches = 0, n_direct_batches = 0; // which path the batches took ("radix_batches" / "direct_batches")
    ncclComm_t comm = nullptr; // multi-process communicator of this handle (kmm_comm_init_rank)
    int comm_rank = -1, comm_size = 0;
    // flush of node range s under the reduce of node range s - 1 (kmm_comm_reduce_counts): "comm_overlap_slices"
    int comm_slices = 8;
    int64_t comm_sliced_reduces = 0; // kmm_comm_reduce_counts calls that issued one reduce per node range ("comm_sliced_reduces")
    hipStream_t comm_stream = nullptr;
    std::vector<hipEvent_t> comm_events;
    std::vector<uint64_t> flush_cuts; // entry (node order) where node range s begins, for comm_slices ranges; [slices + 1]
    // timing
    bool timing = false;
    std::vector<TimedEvent> ev_used;
    std::vector<TimedEvent> ev_free;
    double ms_total[KMM_N_KERNELS] = {0};
    int64_t launches[KMM_N_KERNELS] = {0};
};

namespace {

IndexView view_of(const kmm_index *ix)
{
    IndexView v;
    v.buckets = ix->buckets;
    v.entries = ix->entries;
    v.occ = ix->use_occ ? ix->occ : nullptr;
    v.occ_shift = ix->occ_shift;
    v.bloom_words = ix->bloom_words;
    v.wide = ix->wide ? 1 : 0;
    v.counts = ix->counts;
    v.stats = ix->stats;
    v.modulo = ix->modulo;
    v.magic = ix->magic;
    return v;
}

struct ScopedTimer {
    kmm_index *ix;
    TimedEvent ev{};
    bool active = false;
    int begin(kmm_index *ix_, int kernel_id)
    {
        ix = ix_;
        if (!ix->timing)
            return KMM_OK;
        if (!ix->ev_free.empty()) {
            ev = ix->ev_free.back();
            ix->ev_free.pop_back();
        } else {
            HIPCHK(hipEventCreate(&ev.start));
            HIPCHK(hipEventCreate(&ev.stop));
        }
        ev.kernel_id = kernel_id;
        HIPCHK(hipEventRecord(ev.start, ix->stream));
        active = true;
        return KMM_OK;
    }
    int end()
    {
        if (!active)
            return KMM_OK;
        HIPCHK(hipEventRecord(ev.stop, ix->stream));
        ix->ev_used.push_back(ev);
        active = false;
        return KMM_OK;
    }
};

int rx_flush(kmm_index *ix);
int ensure_direct(kmm_index *ix);
constexpr size_t KMM_STAT_BYTES = (size_t)KMM_STAT_SHARDS * KMM_STAT_STRIDE * 8;

// Self-check of the radix path at every synchronising call: every k-mer pass 1 emitted must have been gathered by
// pass 2 and probed by pass 3 (or dropped by pass 2's empty-bucket filter).  The three passes count independently
// (per-lane registers -> one sharded atomic per wavefront at kernel end), so a work item that is handed out twice, skipped, or
// seen differently by the wavefronts of one workgroup (the round-2 race) shows up here instead of as
// silently wrong counts.  Called with both streams drained.
int rx_check_conservation(kmm_index *ix)
{
    if (!ix->rx_unchecked)
        return KMM_OK;
    ix->rx_unchecked = false;
    static thread_local std::vector<unsigned long long> st;
    st.resize(KMM_STAT_BYTES / 8);
    HIPCHK(hipMemcpy(st.data(), ix->stats, KMM_STAT_BYTES, hipMemcpyDeviceToHost));
    unsigned long long p1 = 0, p2 = 0, p3 = 0, dropped = 0;
    for (int i = 0; i < KMM_STAT_SHARDS; ++i) {
        const unsigned long long *sh = st.data() + (size_t)i * KMM_STAT_STRIDE;
        p1 += sh[KMM_STAT_RX_P1];
        p2 += sh[2];
        p3 += sh[3];
        dropped += sh[KMM_STAT_RX_DROPPED];
    }
    if (p1 == p2 && p2 == p3 + dropped)
        return KMM_OK;
    const int rc = fail(KMM_ERR_INTERNAL, "radix path self-check failed: pass 1 emitted %llu k-mers, pass 2 gathered %llu, "
                        "pass 3 probed %llu (+ %llu dropped as absent) since the counters were last reset: the node counts "
                        "are invalid until kmm_reset_counts [latest map call #%llu on this handle]",
                        p1, p2, p3, dropped, (unsigned long long)ix->map_calls);
    ix->sticky_rc = rc;
    ix->sticky_msg = g_err;
    return rc;
}

// Drain the streams and surface deferred device-side errors (invalid bases, malformed records, bad offsets).
// The reference raises before any count of the offending chunk is added (bionumpy's encoder, util.py:72); here
// the chunk's valid windows have already been counted when the error is seen, so the error stays on the handle:
// every later synchronising call fails with the same code until kmm_reset_counts clears counts and error together.
int drain(kmm_index *ix)
{
    KMMCHK(rx_flush(ix));
    HIPCHK(hipStreamSynchronize(ix->copy_stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    if (ix->sticky_rc != KMM_OK)
        return fail(ix->sticky_rc, "%s", ix->sticky_msg.c_str());
    KMMCHK(rx_check_conservation(ix));
    unsigned long long bad[3] = {NO_BAD, NO_BAD, NO_BAD};
    HIPCHK(hipMemcpy(bad, ix->first_bad, sizeof bad, hipMemcpyDeviceToHost));
    if (bad[0] != NO_BAD || bad[1] != NO_BAD || bad[2] != NO_BAD) {
        unsigned long long reset[3] = {NO_BAD, NO_BAD, NO_BAD};
        HIPCHK(hipMemcpy(ix->first_bad, reset, sizeof reset, hipMemcpyHostToDevice));
        char where[160];
        snprintf(where, sizeof where, " [one of the map calls since the last synchronising call; the latest was call "
                 "#%llu on this handle; counts are invalid until kmm_reset_counts]", (unsigned long long)ix->map_calls);
        int rc;
        if (bad[2] != NO_BAD)
            rc = fail(KMM_ERR_INVALID_ARG, "read_offsets of a mapped chunk is not non-decreasing at read %llu%s",
                      bad[2], where);
        else if (bad[1] != NO_BAD)
            rc = fail(KMM_ERR_MALFORMED,
                      "record structure violated at byte offset %llu of a mapped chunk (a record line "
                      "does not start with '@' / '+' / '>'): multi-line FASTA/FASTQ is not supported by "
                      "the GPU reader%s", bad[1], where);
        else
            rc = fail(KMM_ERR_INVALID_BASE,
                      "read byte at offset %llu of a mapped chunk is not a nucleotide under the "
                      "lookup table (the reference's DNA encoder raises here)%s", bad[0], where);
        ix->sticky_rc = rc;
        ix->sticky_msg = g_err;
        return rc;
    }
    return KMM_OK;
}

int grid_for(const kmm_index *ix, int64_t work_items, int per_cu)
{
    int64_t cap = (int64_t)ix->n_cu * per_cu;
    int64_t g = work_items < cap ? work_items : cap;
    return (int)(g < 1 ? 1 : g);
}

// Grid of the fused kernel.  Every workgroup pays a LUT load, an aggregation-table init and a flush, so
// small batches want several tiles per workgroup (>= 4 once all CU slots are taken), while large batches
// run ~15 % faster with many more workgroups than CU slots (measured: 8 / 16 / 64 per CU = 23.1 / 21.0 /
// 20.2 ms per 1.2e9 k-mers).
int grid_for_tiles(const kmm_index *ix, int64_t n_tiles)
{
    const int64_t slots = (int64_t)ix->n_cu * 8;
    const int64_t cap = (int64_t)ix->n_cu * ix->grid_per_cu;
    int64_t g = n_tiles <= slots ? n_tiles : n_tiles / 4;
    if (n_tiles > slots && g < slots)
        g = slots;
    if (g > cap)
        g = cap;
    return (int)(g < 1 ? 1 : g);
}

// Stage a host array into the given device buffer on the copy stream; device arrays pass through.
template <typename TT>
int stage_in(kmm_index *ix, DevBuf &buf, const TT *src, size_t count, const TT **dev, bool *staged)
{
    if (count == 0) {
        KMMCHK(ensure(buf, 256));
        *dev = static_cast<const TT *>(buf.p);
        return KMM_OK;
    }
    if (is_device_ptr(src)) {
        *dev = src;
        return KMM_OK;
    }
    KMMCHK(ensure(buf, count * sizeof(TT)));
    HIPCHK(hipMemcpyAsync(buf.p, src, count * sizeof(TT), hipMemcpyHostToDevice, ix->copy_stream));
    *dev = static_cast<const TT *>(buf.p);
    *staged = true;
    return KMM_OK;
}

Stage &next_stage(kmm_index *ix)
{
    Stage &s = ix->stage[ix->cur];
    ix->cur ^= 1;
    return s;
}

// Every map call: (1) the stage's buffers may be overwritten once the kernels of the call that
// last used them are done; (2) kernels may start once the copies are in; (3) borrowed host buffers
// are free again once the copies are done.
int stage_acquire(kmm_index *ix, Stage &s)
{
    if (s.used)
        HIPCHK(hipStreamWaitEvent(ix->copy_stream, s.done, 0));
    return KMM_OK;
}

int stage_copies_done(kmm_index *ix)
{
    HIPCHK(hipEventRecord(ix->copied, ix->copy_stream));
    HIPCHK(hipStreamWaitEvent(ix->stream, ix->copied, 0));
    return KMM_OK;
}

int stage_release(kmm_index *ix, Stage &s, bool staged)
{
    HIPCHK(hipEventRecord(s.done, ix->stream));
    s.used = true;
    if (staged)
        HIPCHK(hipEventSynchronize(ix->copied));
    return KMM_OK;
}

int resolve_lut(kmm_index *ix, Stage &s, const uint8_t *lut, const uint8_t **dev, bool *staged)
{
    if (!lut) {
        *dev = ix->lut_default;
        return KMM_OK;
    }
    return stage_in<uint8_t>(ix, s.lut, lut, 256, dev, staged);
}

// Layout choice by index size (profiles/r01/partitioned_path_ablation.md, ms per 1.2e9 k-mers, same box):
//   16-byte buckets + L2 bitmap vs 32-byte buckets without: 10 M entries 20.2 / 24.4, 15 M 21.4 / 26.3,
//   20 M 23.6 / 27.9, 40 M (10 MB bitmap) 28.7 / 30.3, 100 M (25 MB bitmap) 34.0 / 31.7.
// (with fingerprints: 55 M entries 28.1 bitmap / 30.9 wide, 70 M 29.2 / 31.2, 100 M 30.6 / 31.7, 200 M 32.4 / 27.9)
constexpr size_t KMM_OCC_MAX_BYTES = (size_t)20 << 20;  // 168 M buckets at one bit per bucket
constexpr size_t KMM_BLOOM_MAX_BYTES = (size_t)4 << 20;  // Bloom filter size cap (L2 of one XCD)
constexpr int64_t KMM_BLOOM_MAX_ENTRIES = 20000000;     // beyond: fewer than ~2.5 bits per key, no better than the bitmap
constexpr size_t KMM_OCC_SWEET_BYTES = (size_t)5 << 20; // bitmap size that still lives in the 4 MiB L2s + MALL
constexpr int TILE_S = 4;
constexpr int TILE_T = 256 * TILE_S;

// Fan-out of the radix path for 2^w buckets per fine partition: F1 coarse x F2 fine partitions, each at most `maxf`
// (<= RX_MAXF = 512: 512 x 512 slices of 8192 buckets cover every modulo the index format's int32 tables allow,
// mapper.pyx:22-23,31-32).  f2_force >= 0 (experiments, KMM_RX_F2): that many fine-partition bits.
bool rx_configure(kmm_index *ix, int w, int maxf = RX_MAXF, int f2_force = -1)
{
    if (w < 0 || w > 13 || ix->modulo >= (1ull << 31)) // (pass 1 divides with a 32-bit remainder)
        return false;
    const uint64_t PF = (ix->modulo + (1ull << w) - 1) >> w;
    if (PF > (uint64_t)maxf * maxf)
        return false;
    int lg = 0;
    while ((1ull << lg) < PF)
        ++lg;
    int f2 = (lg + 1) / 2;
    bool for_filter = false;
    int occ_shift = 0;
    // pass 2's empty-bucket filter has 2^19 bits of LDS per coarse partition: take fewer fine-partition bits — more,
    // smaller coarse partitions — when pass 1's fan-out stays within 512; tables too large for that at one bit per
    // bucket get one bit per 2 or 4 buckets (sparse tables such as modulo 452 930 477 with 1e8 entries still lose
    // half of their k-mers there; at load factor 0.5 a bit per 4 buckets would pass 86 %: not taken)
    if (ix->rx_filter && w + f2 > P2F_LOGBITS) {
        const double load = ix->rx_S ? (double)ix->rx_S / (double)ix->modulo : 0.5;
        for (int gs = 0; gs <= 2 && !for_filter; ++gs) {
            const int fb = P2F_LOGBITS + gs - w;
            if (fb < 0 || load * (double)(1 << gs) > 0.75)
                continue;
            if (((PF + (1ull << fb) - 1) >> fb) <= (uint64_t)RX_MAXF && fb <= 9) {
                f2 = fb;
                occ_shift = gs;
                for_filter = true;
            }
        }
    }
    if (f2_force >= 0) {
        f2 = f2_force;
        occ_shift = w + f2 > P2F_LOGBITS ? w + f2 - P2F_LOGBITS : 0; // (rx_filter_active refuses more than 2)
    }
    // the packed form (kmm_radix.hpp) keeps floor(q / modulo) above w + f2 hash bits: it must fit for EVERY
    // 64-bit q (callers may hand over arbitrary uint64 values), else give the quotient more room
    const uint64_t max_quo = ~0ull / ix->modulo;
    auto fits = [&](int sh) { return sh == 0 || (sh < 64 && (max_quo >> (64 - sh)) == 0); };
    while (f2 > 0 && !fits(w + f2))
        --f2;
    if (!fits(w + f2))
        return false;
    const uint64_t F2 = 1ull << f2, F1 = (PF + F2 - 1) / F2;
    if (F2 > (uint64_t)maxf || F1 > (uint64_t)(for_filter ? RX_MAXF : maxf))
        return false;
    ix->rx_w = w;
    ix->rx_f2 = f2;
    ix->rx_occ_shift = w + f2 > P2F_LOGBITS ? w + f2 - P2F_LOGBITS : 0; // (the filter's bits never outnumber its LDS)
    (void)occ_shift;
    ix->rx_PF = (uint32_t)PF;
    ix->rx_F1 = (uint32_t)F1;
    ix->rx_F2 = (uint32_t)F2;
    return true;
}

// Pass 2 filters k-mers of empty buckets (k_rx_p2f) when a coarse partition's bitmap fits its 64 KB of LDS and its
// first bucket starts a bitmap word.
bool rx_filter_active(const kmm_index *ix)
{
    const int sh = ix->rx_w + ix->rx_f2;
    return ix->rx_filter && ix->rx_occ && sh - ix->rx_occ_shift >= 5 && ix->rx_occ_shift <= 2;
}

bool use_radix(const kmm_index *ix, int64_t units)
{
    if (!ix->rx_ok || ix->path == 1)
        return false;
    return ix->path == 2 || ix->rx_ecnt_acc || units >= ix->rx_min_units;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// One sub-batch of the radix path: pass 1 (reads or k-mers -> blocks sorted by coarse partition), the
// directory scan, pass 2 (items sorted by fine partition), pass 3 (LDS probe).  Hits land in rx_ecnt.
template <int MODE>
int launch_rx(kmm_index *ix, const ReadsView &rv, const uint64_t *kmers_in, int64_t n_in, int k, int max_freq,
              int also_rc)
{
    const IndexView iv = view_of(ix);
    const int64_t units = MODE == MODE_KMERS ? n_in : rv.total;
    // blocks of pass 1: 8192 positions, or (packed tiles) the reads of two tiles
    const int64_t n_src_total = MODE == MODE_PACKED ? (rv.n_reads + 2 * (int64_t)rv.pk_rpt - 1) / (2 * (int64_t)rv.pk_rpt)
                                                    : (units + RX_B - 1) / RX_B;
    const uint32_t X = also_rc ? 2u : 1u;
    const uint32_t F1 = ix->rx_F1, F2 = ix->rx_F2;
    // Sub-batches.  Every sub-batch streams the index slices once (pass 3) and pays the per-work-item costs once, so they
    // are as large as the 32-bit prefixes allow: a coarse partition's k-mers are numbered with 32 bits and in the worst
    // case (one k-mer repeated) ALL of a sub-batch's k-mers fall into one coarse partition, hence fewer than 2^32 k-mer
    // slots per sub-batch (r03: 2^31; the 1 B-k-mer index streamed its 14 GB of slices twice per 28 M-read batch).  Every
    // other offset is 64-bit or relative (to a table's first block, to a work item's first item).  Without the HBM for
    // the buffers of that size the call takes one sub-batch more, and again, down to 2^28 slots per sub-batch; the handle
    // remembers the size that fitted ("radix_sub_batch_kmers_effective") for its next 15 calls and then tries the caller's
    // cap again — an allocation that failed because something else held the memory for a moment does not shrink the handle's
    // sub-batches for good (round 4 halved "radix_sub_batch_kmers" itself: four failing rounds for a 2^29-slot batch, for ever).
    // Sub-batches of equal size: no small last one.
    int64_t n_sub = 1, max_src = 0;
    static const bool verbose = getenv("KMM_VERBOSE") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    if (ix->rx_sub_cap_eff > 0 && ++ix->rx_sub_cap_eff_age >= 16)
        ix->rx_sub_cap_eff = 0;
    int64_t cap = ix->rx_sub_cap_eff > 0 && ix->rx_sub_cap_eff < ix->rx_sub_cap ? ix->rx_sub_cap_eff : ix->rx_sub_cap;
    for (;;) {
        const int64_t cap_src = (cap / RX_B) / X > 0 ? (cap / RX_B) / X : 1;
        n_sub = (n_src_total + cap_src - 1) / cap_src;
        max_src = n_sub ? (n_src_total + n_sub - 1) / n_sub : cap_src;
        const size_t NBm = (size_t)max_src * X, chunks_m = (NBm + RX_CH - 1) / RX_CH, items_m = NBm + F1 + 1;
        const size_t meta = align256(NBm * (F1 + 1) * 2) + align256((size_t)F1 * (NBm + 1) * 4) + align256((size_t)F1 * NBm * 2) +
                            align256(chunks_m * F1 * 4) + 3 * align256((size_t)(F1 + 1) * 4) + align256(items_m * 8) +
                            align256(items_m * (F2 + 1) * 2) + align256(items_m * (F2 + 1) * 2 + 256) + align256(64) + align256(2048);
        int rc = ix->dbg_rx_buf_limit && NBm * RX_B * 8 > (size_t)ix->dbg_rx_buf_limit ? (int)KMM_ERR_NOMEM // (test hook)
                                                                                         : ensure(ix->rx_meta, meta);
        if (rc == KMM_OK)
            rc = ensure(ix->rx_buf1, NBm * RX_B * 8);
        if (rc == KMM_OK)
            rc = ensure(ix->rx_buf2, items_m * RX_B * 8);
        if (rc == KMM_OK)
            break;
        if (rc != KMM_ERR_NOMEM || cap <= ((int64_t)1 << 28) || n_src_total <= 1)
            return rc;
        (void)hipGetLastError();
        release(ix->rx_buf1);
        release(ix->rx_buf2);
        // the next size that really is smaller: one sub-batch more than this attempt had
        const int64_t next_src = (n_src_total + n_sub) / (n_sub + 1);
        cap = next_src * RX_B * X;
        if (cap < ((int64_t)1 << 28))
            cap = (int64_t)1 << 28;
        ix->rx_sub_cap_eff = cap;
        ix->rx_sub_cap_eff_age = 0;
    }
    ix->rx_sub_cap_last = cap;
    const double ms_buffers = ms_since(t_0);
    for (int64_t s0 = 0; s0 < n_src_total; s0 += max_src) {
        const uint32_t n_src = (uint32_t)(n_src_total - s0 < max_src ? n_src_total - s0 : max_src);
        const uint32_t NB = n_src * X;
        const uint32_t chunks = (NB + RX_CH - 1) / RX_CH;
        const size_t max_items = (size_t)NB + F1 + 1;
        RxView rx;
        memset(&rx, 0, sizeof rx);
        rx.pstart = ix->rx_pstart; rx.pkeys = ix->rx_pkeys; rx.pfreq = ix->rx_pfreq; rx.ecnt = ix->rx_ecnt;
        rx.pstart16 = ix->rx_pstart16; rx.slice_e0 = ix->rx_slice_e0; rx.slice_fmax = ix->rx_slice_fmax;
        rx.occ = ix->rx_occ;
        rx.occ_shift = rx_filter_active(ix) ? ix->rx_occ_shift : 3; // (3: k_rx_p2f without its filter)
        rx.p2f_k = NB / 2048u < 4u ? 4u : (NB / 2048u > (uint32_t)P2F_KMAX ? (uint32_t)P2F_KMAX : NB / 2048u);
        rx.w = ix->rx_w; rx.f2 = ix->rx_f2; rx.PF = ix->rx_PF; rx.F1 = F1; rx.F2 = F2;
        rx.NB = NB; rx.max_items = (uint32_t)max_items;
        size_t off = 0;
        auto carve = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
        const size_t o_start1 = carve((size_t)NB * (F1 + 1) * 2), o_P1T = carve((size_t)F1 * (NB + 1) * 4),
                     o_S1T = carve((size_t)F1 * NB * 2), o_csum = carve((size_t)chunks * F1 * 4),
                     o_T1 = carve((size_t)F1 * 4), o_ib = carve((size_t)(F1 + 1) * 4), o_wb = carve((size_t)(F1 + 1) * 4),
                     o_desc = carve(max_items * 8), o_start2 = carve(max_items * (F2 + 1) * 2), o_start2T = carve(max_items * (F2 + 1) * 2 + 256),
                     o_ctrl = carve(64),
                     o_queue = carve(2048);
        KMMCHK(ensure(ix->rx_meta, off));
        KMMCHK(ensure(ix->rx_buf1, (size_t)NB * RX_B * 8));
        KMMCHK(ensure(ix->rx_buf2, max_items * RX_B * 8));
        uint8_t *m = (uint8_t *)ix->rx_meta.p;
        rx.start1 = (uint16_t *)(m + o_start1); rx.P1T = (uint32_t *)(m + o_P1T); rx.S1T = (uint16_t *)(m + o_S1T);
        rx.csum = (uint32_t *)(m + o_csum); rx.T1 = (uint32_t *)(m + o_T1); rx.item_base = (uint32_t *)(m + o_ib);
        rx.work_base = (uint32_t *)(m + o_wb); rx.item_desc = (uint2 *)(m + o_desc);
        rx.start2 = (uint16_t *)(m + o_start2); rx.start2T = (uint16_t *)(m + o_start2T);
        rx.ctrl = (uint32_t *)(m + o_ctrl);
        rx.queue = (unsigned long long *)(m + o_queue);
        rx.buf1 = (uint64_t *)ix->rx_buf1.p;
        rx.buf2 = (uint64_t *)ix->rx_buf2.p;
#if RX_PROBE_ANY
        KMMCHK(ensure(ix->rx_probe, max_items * RX_B * 8));
        rx.probe = (uint64_t *)ix->rx_probe.p;
#endif
        ix->dbg_T1 = rx.T1; ix->dbg_item_base = rx.item_base; ix->dbg_start1 = rx.start1; ix->dbg_F1 = F1; ix->dbg_NB = NB;
        HIPCHK(hipMemsetAsync(m + o_ctrl, 0, align256(64) + 2048, ix->stream));
        ScopedTimer tm;
        KMMCHK(tm.begin(ix, KMM_KERNEL_RX_P1));
        static const int p1_per_cu = getenv("KMM_RX_P1_GRID_PER_CU") ? atoi(getenv("KMM_RX_P1_GRID_PER_CU")) : 8; // (experiments)
        const int64_t g1cap = (int64_t)ix->n_cu * (p1_per_cu > 0 ? p1_per_cu : 8);
        const dim3 g1((unsigned)(n_src < g1cap ? n_src : g1cap));
        const int64_t tile0 = s0 * (RX_B / (MODE == MODE_RECORDS ? 1024 : 4096));
        const uint64_t *src_kmers = kmers_in ? kmers_in + s0 * RX_B : nullptr;
        constexpr bool CAN_C2 = MODE == MODE_PACKED || MODE == MODE_UNIFORM || MODE == MODE_GENERAL;
        if (rv.codes2 && !CAN_C2)
            return fail(KMM_ERR_INTERNAL, "2-bit code input reaches pass 1 through the flat-read modes only");
        if constexpr (CAN_C2) {
            if (rv.codes2 && also_rc)
                hipLaunchKernelGGL((k_rx_p1<MODE, true, true>), g1, dim3(RX_NT), 0, ix->stream, rv, src_kmers, n_in - s0 * RX_B,
                                   iv, rx, k, tile0, n_src);
            else if (rv.codes2)
                hipLaunchKernelGGL((k_rx_p1<MODE, false, true>), g1, dim3(RX_NT), 0, ix->stream, rv, src_kmers, n_in - s0 * RX_B,
                                   iv, rx, k, tile0, n_src);
        }
        if (rv.codes2) {
        } else if (also_rc)
            hipLaunchKernelGGL((k_rx_p1<MODE, true>), g1, dim3(RX_NT), 0, ix->stream, rv, src_kmers, n_in - s0 * RX_B,
                               iv, rx, k, tile0, n_src);
        else
            hipLaunchKernelGGL((k_rx_p1<MODE, false>), g1, dim3(RX_NT), 0, ix->stream, rv, src_kmers, n_in - s0 * RX_B,
                               iv, rx, k, tile0, n_src);
        HIPCHK(hipGetLastError());
        KMMCHK(tm.end());
        KMMCHK(tm.begin(ix, KMM_KERNEL_RX_SCAN));
        hipLaunchKernelGGL(k_rx_colsum, dim3(chunks), dim3(256), 0, ix->stream, rx);
        hipLaunchKernelGGL(k_rx_chunkscan, dim3(F1), dim3(256), 0, ix->stream, rx, chunks);
        hipLaunchKernelGGL(k_rx_tables, dim3(1), dim3(512), 0, ix->stream, rx);
        hipLaunchKernelGGL(k_rx_colscan, dim3(chunks, (F1 + 255) / 256), dim3(256), 0, ix->stream, rx);
        HIPCHK(hipGetLastError());
        KMMCHK(tm.end());
        KMMCHK(tm.begin(ix, KMM_KERNEL_RX_P2));
        if (ix->rx_filter) { // gather by k-mer; where a coarse partition's occupancy bitmap fits LDS (one bit per 1, 2 or 4
                           // buckets) the k-mers of empty buckets are dropped here
            const bool flt = rx_filter_active(ix), small = F2 <= 128;
            auto kern = flt ? (small ? k_rx_p2f<true, true> : k_rx_p2f<true, false>)
                            : (small ? k_rx_p2f<false, true> : k_rx_p2f<false, false>);
            hipLaunchKernelGGL(kern, dim3(ix->n_cu), dim3(P2F_NT), 0, ix->stream, iv, rx);
        } else {
            hipLaunchKernelGGL(k_rx_p2, dim3(ix->n_cu * ix->rx_grid_per_cu), dim3(RX_NT), 0, ix->stream, iv, rx);
        }
        HIPCHK(hipGetLastError());
        KMMCHK(tm.end());
        KMMCHK(tm.begin(ix, KMM_KERNEL_RX_SCAN));
        hipLaunchKernelGGL(k_rx_tr2, dim3((unsigned)((max_items + RX_TR2 - 1) / RX_TR2)), dim3(256), 0, ix->stream, rx);
        HIPCHK(hipGetLastError());
        KMMCHK(tm.end());
        KMMCHK(tm.begin(ix, KMM_KERNEL_RX_P3));
        const bool p16 = ix->rx_pstart16 != nullptr; // (implies: no slice beyond 65535 entries)
        if (ix->rx_w > 12 && ix->rx_fits_small && p16)
            // 8192-bucket slices whose entries all fit 4096 keys: 16-bit directory, two workgroups per CU
            hipLaunchKernelGGL((k_rx_p3<RX_WMAX_BIG, RX_ECAP, 4, uint16_t, RX_SUBCAP3, true>), dim3(ix->n_cu * ix->rx_grid_per_cu),
                               dim3(RX_NT), 0, ix->stream, iv, rx, max_freq);
        else if (ix->rx_w > 12 && ix->rx_fits_small && ix->rx_max_slice <= 65535u)
            hipLaunchKernelGGL((k_rx_p3<RX_WMAX_BIG, RX_ECAP, 4, uint16_t>), dim3(ix->n_cu * ix->rx_grid_per_cu), dim3(RX_NT),
                               0, ix->stream, iv, rx, max_freq);
        else if (ix->rx_w > 12 && ix->rx_fits_mid && p16 && !getenv("KMM_RX_NO_MID"))
            // ... at most 4608 keys (load factor 0.5): the same with a shorter piece list
            hipLaunchKernelGGL((k_rx_p3<RX_WMAX_BIG, RX_ECAP_MID, 4, uint16_t, 1024, true>), dim3(ix->n_cu * ix->rx_grid_per_cu),
                               dim3(RX_NT), 0, ix->stream, iv, rx, max_freq);
        else if (ix->rx_w > 12 && ix->rx_fits_mid && ix->rx_max_slice <= 65535u && !getenv("KMM_RX_NO_MID"))
            hipLaunchKernelGGL((k_rx_p3<RX_WMAX_BIG, RX_ECAP_MID, 4, uint16_t, 1024>), dim3(ix->n_cu * ix->rx_grid_per_cu),
                               dim3(RX_NT), 0, ix->stream, iv, rx, max_freq);
        else if (ix->rx_w > 12)
            hipLaunchKernelGGL((k_rx_p3<RX_WMAX_BIG, RX_ECAP_BIG, 2, uint32_t>), dim3(ix->n_cu), dim3(RX_NT), 0, ix->stream,
                               iv, rx, max_freq);
        else if (ix->rx_w <= 12 && p16)
            // slices of up to 4096 buckets (the usual case): 16-bit directory loaded as it is
            hipLaunchKernelGGL((k_rx_p3<RX_WMAX, RX_ECAP, 4, uint16_t, RX_SUBCAP3, true>), dim3(ix->n_cu * ix->rx_grid_per_cu),
                               dim3(RX_NT), 0, ix->stream, iv, rx, max_freq);
        else
            hipLaunchKernelGGL((k_rx_p3<RX_WMAX, RX_ECAP, 4, uint32_t>), dim3(ix->n_cu * ix->rx_grid_per_cu), dim3(RX_NT), 0,
                               ix->stream, iv, rx, max_freq);
        HIPCHK(hipGetLastError());
        KMMCHK(tm.end());
        ix->ecnt_dirty = true;
        ix->rx_unchecked = true;
    }
    ix->n_radix_batches++;
    if (verbose && ms_since(t_0) > 20.0)
        fprintf(stderr, "libkmm: radix passes of %lld positions: %.2f ms for the batch buffers (%zu + %zu + %zu bytes), %.2f ms to issue %lld "
                "sub-batch(es)\n", (long long)units, ms_buffers, ix->rx_meta.cap, ix->rx_buf1.cap, ix->rx_buf2.cap, ms_since(t_0) - ms_buffers,
                (long long)n_sub);
    return KMM_OK;
}

// Per-entry hits of the radix path -> node counts (mapper.pyx:68 summed per entry first).  Asynchronous.
int rx_flush(kmm_index *ix)
{
    if (!ix->ecnt_dirty)
        return KMM_OK;
    ScopedTimer tm;
    KMMCHK(tm.begin(ix, KMM_KERNEL_RX_FLUSH));
    if (ix->rx_norder && !ix->rx_ecnt_acc && ix->rx_flush_sorted) {
        // entries in node order: a gather of the counts + atomics that walk the count vector front to back
        hipLaunchKernelGGL(k_rx_flush_sorted, dim3(ix->n_cu * 8), dim3(256), 0, ix->stream, view_of(ix), ix->rx_ecnt,
                           ix->rx_norder, ix->rx_nnode, ix->rx_S);
        HIPCHK(hipMemsetAsync(ix->rx_ecnt, 0, (size_t)ix->rx_S * 4, ix->stream));
    } else {
        hipLaunchKernelGGL(k_rx_flush, dim3(ix->n_cu * 8), dim3(256), 0, ix->stream, view_of(ix), ix->rx_ecnt,
                           ix->rx_pnodes, ix->rx_S, ix->rx_ecnt_acc);
    }
    HIPCHK(hipGetLastError());
    KMMCHK(tm.end());
    ix->ecnt_dirty = false;
    return KMM_OK;
}

template <int MODE>
int launch_map_reads(kmm_index *ix, const ReadsView &rv, int k, int max_freq, int also_rc)
{
    const int64_t n_tiles = (rv.total + TILE_T - 1) / TILE_T;
    // raw records reach this function only for the direct path: on the radix path they are compacted into flat reads
    // first (map_records_piece_radix)
    const bool radix = MODE == MODE_RECORDS ? false : use_radix(ix, rv.total);
    if (!radix) {
        KMMCHK(ensure_direct(ix));
        ix->n_direct_batches++;
        const IndexView iv = view_of(ix); // (the direct view may have been packed just now)
        ScopedTimer tm;
        KMMCHK(tm.begin(ix, KMM_KERNEL_MAP_READS));
        // large launches: persistent workgroups + dynamic tile queue; small ones: static schedule
        const int64_t slots = (int64_t)ix->n_cu * 8;
        const bool dynamic = ix->dynamic_schedule && n_tiles >= slots * 4 * ix->dyn_chunk;
        if (dynamic)
            HIPCHK(hipMemsetAsync(ix->queue, 0, 3 * sizeof(unsigned long long), ix->stream));
        const dim3 grid(dynamic ? (unsigned)slots : (unsigned)grid_for_tiles(ix, n_tiles));
        unsigned long long *queue = dynamic ? ix->queue : nullptr;
        if (iv.occ && iv.wide)
            hipLaunchKernelGGL((k_map_reads<TILE_S, MODE, PROBE_WIDE_FILTER>), grid, dim3(256), 0, ix->stream, rv,
                               iv, k, max_freq, also_rc, (int64_t)0, n_tiles, queue, ix->dyn_chunk);
        else if (iv.occ)
            hipLaunchKernelGGL((k_map_reads<TILE_S, MODE, PROBE_BITMAP>), grid, dim3(256), 0, ix->stream, rv, iv, k,
                               max_freq, also_rc, (int64_t)0, n_tiles, queue, ix->dyn_chunk);
        else if (iv.wide)
            hipLaunchKernelGGL((k_map_reads<TILE_S, MODE, PROBE_WIDE>), grid, dim3(256), 0, ix->stream, rv, iv, k,
                               max_freq, also_rc, (int64_t)0, n_tiles, queue, ix->dyn_chunk);
        else
            hipLaunchKernelGGL((k_map_reads<TILE_S, MODE, PROBE_NARROW>), grid, dim3(256), 0, ix->stream, rv, iv, k,
                               max_freq, also_rc, (int64_t)0, n_tiles, queue, ix->dyn_chunk);
        HIPCHK(hipGetLastError());
        return tm.end();
    }
    if constexpr (MODE == MODE_RECORDS) {
        return fail(KMM_ERR_INTERNAL, "raw records reach the radix path as flat reads only");
    } else {
        if (MODE == MODE_UNIFORM && rv.pk_rpt) // reads of one length: tiles of whole reads, every computed window a real k-mer
            return launch_rx<MODE_PACKED>(ix, rv, nullptr, 0, k, max_freq, also_rc);
        return launch_rx<MODE>(ix, rv, nullptr, 0, k, max_freq, also_rc);
    }
}

// Exclusive scan of n uint32 values on the device (in -> out): 1024-wide block scans, recursing on the
// block totals (n < 2^32 needs at most 4 levels).  All launches go to `stream`; ensure() may hipFree (device-wide sync).  `scratch` holds one DevBuf pair per level and is sized
// by the caller ONCE — it must not reallocate while outer levels hold references into it.
constexpr size_t SCAN_MAX_LEVELS = 8;
int scan_exclusive(const uint32_t *in, uint32_t *out, uint64_t n, std::vector<DevBuf> &scratch,
                   size_t level, hipStream_t stream)
{
    const uint64_t n_blocks = (n + 1023) / 1024;
    if (level >= SCAN_MAX_LEVELS || scratch.size() < 2 * SCAN_MAX_LEVELS)
        return fail(KMM_ERR_INVALID_ARG, "scan_exclusive: level %zu out of range", level);
    DevBuf &sums = scratch[2 * level], &pre = scratch[2 * level + 1];
    KMMCHK(ensure(sums, (size_t)n_blocks * 4));
    hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)n_blocks), dim3(1024), 0, stream, in, out, (uint32_t *)sums.p, n);
    HIPCHK(hipGetLastError());
    if (n_blocks == 1)
        return KMM_OK;
    KMMCHK(ensure(pre, (size_t)n_blocks * 4));
    KMMCHK(scan_exclusive((const uint32_t *)sums.p, (uint32_t *)pre.p, n_blocks, scratch, level + 1, stream));
    uint64_t g = (n + 255) / 256;
    if (g > 65536)
        g = 65536;
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)g), dim3(256), 0, stream, out, (const uint32_t *)pre.p, n);
    HIPCHK(hipGetLastError());
    return KMM_OK;
}

// The direct view (kmm_probe.hpp): bucket records + 16-byte entries (+ the L2 pre-filter for small indexes), packed
// from the caller's arrays at creation or from the radix view's bucket-ordered copy later.  Synchronous.
template <typename Src>
int direct_build(kmm_index *ix, const Src &src, int64_t n_entries, size_t occ_bytes, uint32_t *d_err)
{
    const uint64_t M = ix->modulo;
    const size_t nb = sizeof(uint4) * (size_t)M * (ix->wide ? 2 : 1), ne = sizeof(uint4) * (size_t)(n_entries > 0 ? n_entries : 1);
    hipError_t e = hipMalloc(&ix->buckets, nb);
    if (e == hipSuccess) e = hipMalloc(&ix->entries, ne);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (ix->buckets) { (void)hipFree(ix->buckets); ix->buckets = nullptr; }
        return fail(e == hipErrorOutOfMemory ? KMM_ERR_NOMEM : KMM_ERR_HIP, "direct view of the index (%zu + %zu bytes): %s",
                    nb, ne, hipGetErrorString(e));
    }
    if (ix->wide)
        hipLaunchKernelGGL((k_pack_buckets_wide<Src>), dim3(grid_for(ix, (int64_t)((M + 255) / 256), 16)), dim3(256), 0,
                           ix->stream, src, M, n_entries, ix->max_node_id, ix->buckets, d_err);
    else
        hipLaunchKernelGGL((k_pack_buckets<Src>), dim3(grid_for(ix, (int64_t)((M + 255) / 256), 16)), dim3(256), 0,
                           ix->stream, src, M, n_entries, ix->max_node_id, ix->buckets, d_err);
    if (n_entries > 0)
        hipLaunchKernelGGL((k_pack_entries<Src>), dim3(grid_for(ix, (n_entries + 255) / 256, 16)), dim3(256), 0, ix->stream,
                           src, n_entries, ix->max_node_id, ix->entries, d_err);
    e = hipGetLastError();
    // occupancy bitmap / Bloom filter (16-byte layout only), built from the k-mers while they are here
    if (e == hipSuccess && occ_bytes) {
        e = hipMalloc(&ix->occ, occ_bytes);
        if (e == hipSuccess) e = hipMemsetAsync(ix->occ, 0, occ_bytes, ix->stream);
        if (e == hipSuccess && n_entries > 0) {
            if (ix->bloom_words)
                hipLaunchKernelGGL(k_build_bloom, dim3(grid_for(ix, (n_entries + 255) / 256, 16)), dim3(256), 0, ix->stream,
                                   src.kmers, n_entries, ix->bloom_words, ix->occ);
            else
                hipLaunchKernelGGL(k_build_occ, dim3(grid_for(ix, (n_entries + 255) / 256, 16)), dim3(256), 0, ix->stream,
                                   src.kmers, n_entries, M, ix->magic, ix->occ_shift, ix->occ);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e != hipSuccess)
        return fail(KMM_ERR_HIP, "index repack: %s", hipGetErrorString(e));
    return KMM_OK;
}

// Called by every entry point that needs the direct view (small batches, kmm_in_index).
int ensure_direct(kmm_index *ix)
{
    if (ix->buckets || !ix->direct_deferred)
        return KMM_OK;
    DevBuf d_err;
    KMMCHK(ensure(d_err, 4));
    HIPCHK(hipMemsetAsync(d_err.p, 0, 4, ix->stream));
    IdxRx src;
    src.pstart = ix->rx_pstart; src.kmers = ix->rx_pkeys_raw; src.nodes = ix->rx_pnodes; src.freqs = ix->rx_pfreq;
    const int rc = direct_build(ix, src, (int64_t)ix->rx_S, ix->occ_bytes_plan, (uint32_t *)d_err.p);
    release(d_err);
    if (rc == KMM_OK)
        ix->direct_deferred = false;
    return rc;
}

int check_k(int k)
{
    if (k < 1 || k > KMM_MAX_K)
        return fail(KMM_ERR_INVALID_ARG, "k=%d outside [1, %d]", k, KMM_MAX_K);
    return KMM_OK;
}

} // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char *kmm_version(void) { return "kmm 0.3.0 (gfx950)"; }

const char *kmm_last_error(void) { return g_err.c_str(); }

int kmm_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0)
        return fail(KMM_ERR_INVALID_ARG, "out is NULL or bytes is 0");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return fail(e == hipErrorOutOfMemory ? KMM_ERR_NOMEM : KMM_ERR_HIP, "hipHostMalloc(%zu bytes) -> %s", bytes,
                    hipGetErrorString(e));
    }
    return KMM_OK;
}

int kmm_host_reserve(int64_t raw_batch_bytes)
{
    if (raw_batch_bytes < 0)
        return fail(KMM_ERR_INVALID_ARG, "raw_batch_bytes negative");
    // the two page-locked buffers a host-packed batch of that many raw bytes needs (map_records_host_packed / map_reads_host_packed)
    const size_t n = (size_t)raw_batch_bytes;
    const size_t want[2] = {(n / 4 + 1024 + 63) & ~(size_t)63, (n / 8 + 256 + 63) & ~(size_t)63};
    for (size_t w : want) {
        uint8_t *p = nullptr;
        const size_t take = w + w / 8;
        if (hipHostMalloc(reinterpret_cast<void **>(&p), take, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return fail(KMM_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed", take);
        }
        g_shelf.give(p, take);
    }
    return KMM_OK;
}

int kmm_host_reserve_buffer(int64_t bytes)
{
    if (bytes < 0)
        return fail(KMM_ERR_INVALID_ARG, "bytes negative");
    if (bytes == 0)
        return KMM_OK;
    uint8_t *p = nullptr;
    const size_t take = ((size_t)bytes + 4095) & ~(size_t)4095;
    if (hipHostMalloc(reinterpret_cast<void **>(&p), take, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(KMM_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed", take);
    }
    g_shelf.give(p, take);
    return KMM_OK;
}

int kmm_host_free(void *p)
{
    if (p)
        HIPCHK(hipHostFree(p));
    return KMM_OK;
}

int kmm_device_count(int *n_devices)
{
    if (!n_devices)
        return fail(KMM_ERR_INVALID_ARG, "n_devices is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *n_devices = n;
    return KMM_OK;
}

int kmm_device_pci_bus_id(int device, char *out, int out_bytes)
{
    if (!out || out_bytes < 16)
        return fail(KMM_ERR_INVALID_ARG, "out is NULL or shorter than 16 bytes");
    out[0] = 0;
    HIPCHK(hipDeviceGetPCIBusId(out, out_bytes, device));
    return KMM_OK;
}

void kmm_index_destroy(kmm_index_t *ix)
{
    if (!ix)
        return;
    (void)hipSetDevice(ix->device);
    if (ix->stream)
        (void)hipStreamSynchronize(ix->stream);
    if (ix->copy_stream)
        (void)hipStreamSynchronize(ix->copy_stream);
    for (Stage &s : ix->stage) {
        release(s.bases);
        release(s.offsets);
        release(s.tile_first);
        release(s.start_bits);
        release(s.kmers);
        release(s.lut);
        release(s.aux);
        if (s.done)
            (void)hipEventDestroy(s.done);
    }
    release(ix->rx_meta);
    release(ix->rx_buf1);
    release(ix->rx_buf2);
    release(ix->rx_probe);
    for (int i = 0; i < 2; ++i) {
        release(ix->bgzf_comp[i]);
        release(ix->bgzf_raw[i]);
        release(ix->bgzf_meta[i]);
        if (ix->bgzf_done[i])
            (void)hipEventDestroy(ix->bgzf_done[i]);
    }
    for (hipEvent_t &ev : ix->bgzf_slot_ev) {
        if (ev)
            (void)hipEventDestroy(ev);
        ev = nullptr;
    }
    release(ix->bgzf_status);
    release(ix->bgzf_tabs);
    release(ix->bgzf_crc);
    release(ix->bgzf_err);
    release(ix->bgzf_carry);
    ix->pack_pool.reset();
    g_shelf.give(ix->pack_pinned, ix->pack_pinned_bytes);
    for (uint8_t *&slot : ix->ring) {
        g_shelf.give(slot, RING_SLOT);
        slot = nullptr;
    }
    g_shelf.give(ix->pack_bits_pinned, ix->pack_bits_pinned_bytes);
    for (hipEvent_t e : ix->comm_events)
        (void)hipEventDestroy(e);
    if (ix->comm_stream)
        (void)hipStreamDestroy(ix->comm_stream);
    for (void *q : {(void *)ix->rx_pstart16, (void *)ix->rx_slice_e0, (void *)ix->rx_slice_fmax, (void *)ix->rx_pstart, (void *)ix->rx_pkeys, (void *)ix->rx_pkeys_raw, (void *)ix->rx_pfreq, (void *)ix->rx_pnodes,
                    (void *)ix->rx_porig, (void *)ix->rx_ecnt, (void *)ix->rx_ecnt_acc, (void *)ix->rx_norder, (void *)ix->rx_nnode,
                    (void *)ix->rx_occ})
        if (q)
            (void)hipFree(q);
    for (auto &ev : ix->ev_used) {
        (void)hipEventDestroy(ev.start);
        (void)hipEventDestroy(ev.stop);
    }
    for (auto &ev : ix->ev_free) {
        (void)hipEventDestroy(ev.start);
        (void)hipEventDestroy(ev.stop);
    }
    if (ix->copied)
        (void)hipEventDestroy(ix->copied);
    if (ix->buckets)
        (void)hipFree(ix->buckets);
    if (ix->entries)
        (void)hipFree(ix->entries);
    if (ix->occ)
        (void)hipFree(ix->occ);
    if (ix->own_counts_buf)
        (void)hipFree(ix->own_counts_buf);
    if (ix->lut_default)
        (void)hipFree(ix->lut_default);
    if (ix->lut_codes)
        (void)hipFree(ix->lut_codes);
    if (ix->first_bad)
        (void)hipFree(ix->first_bad);
    if (ix->stats)
        (void)hipFree(ix->stats);
    if (ix->queue)
        (void)hipFree(ix->queue);
    if (ix->comm && g_rccl.lib)
        (void)g_rccl.CommDestroy(ix->comm);
    if (ix->copy_stream)
        (void)hipStreamDestroy(ix->copy_stream);
    if (ix->stream)
        (void)hipStreamDestroy(ix->stream);
    delete ix;
}

// The entry k-mers in the packed form of the current (w, f2) (kmm_radix.hpp); synchronous.
static int rx_repack_keys(kmm_index *ix)
{
    if (ix->rx_S)
        hipLaunchKernelGGL(k_rx_pack_keys, dim3(grid_for(ix, (int64_t)((ix->rx_S + 255) / 256), 16)), dim3(256), 0,
                           ix->stream, ix->rx_pkeys_raw, ix->rx_S, view_of(ix), ix->rx_w + ix->rx_f2, ix->rx_pkeys);
    HIPCHK(hipGetLastError());
    // most entries of one slice: decides whether pass 3 may keep a 16-bit directory (ix->queue serves as the cell)
    HIPCHK(hipMemsetAsync(ix->queue, 0, 3 * sizeof(unsigned long long), ix->stream));
    hipLaunchKernelGGL(k_rx_max_slice, dim3(grid_for(ix, (int64_t)((ix->rx_PF + 255) / 256), 16)), dim3(256), 0, ix->stream,
                       ix->rx_pstart, ix->modulo, ix->rx_w, ix->rx_PF, ix->queue);
    HIPCHK(hipGetLastError());
    unsigned long long mx[3] = {0, 0, 0};
    HIPCHK(hipMemcpyAsync(mx, ix->queue, sizeof mx, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    ix->rx_max_slice = (uint32_t)mx[0];
    // a slice with more entries than pass 3 keeps in LDS has the buckets behind them walked in HBM: fine for the odd
    // slice (a k-mer stored under 1500 nodes), not as the rule — at most one slice in a thousand (or one slice)
    const unsigned long long odd = ix->rx_PF / 1000 > 1 ? ix->rx_PF / 1000 : 1;
    ix->rx_fits_small = mx[1] <= odd;
    ix->rx_fits_mid = mx[2] <= odd;
    // 16-bit slice-relative directory: pass 3 loads 2 B per bucket instead of 4 (optional: 2 B x modulo of HBM)
    for (void **q : {(void **)&ix->rx_pstart16, (void **)&ix->rx_slice_e0, (void **)&ix->rx_slice_fmax}) {
        if (*q)
            (void)hipFree(*q);
        *q = nullptr;
    }
    if (ix->rx_max_slice <= 65535u && !getenv("KMM_RX_NO_P16")) {
        const size_t n16 = ((size_t)ix->rx_PF << ix->rx_w) + 8;
        if (hipMalloc(&ix->rx_pstart16, n16 * 2) == hipSuccess &&
            hipMalloc(&ix->rx_slice_e0, ((size_t)ix->rx_PF + 2) * 4) == hipSuccess &&
            hipMalloc(&ix->rx_slice_fmax, ((size_t)ix->rx_PF + 2) * 2) == hipSuccess) {
            hipLaunchKernelGGL(k_rx_pstart16, dim3(grid_for(ix, (int64_t)((n16 + 255) / 256), 16)), dim3(256), 0, ix->stream,
                               ix->rx_pstart, ix->modulo, ix->rx_w, ix->rx_PF, ix->rx_pstart16, ix->rx_slice_e0);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(k_rx_slice_fmax, dim3(grid_for(ix, (int64_t)((ix->rx_PF + 3) / 4), 16)), dim3(256), 0, ix->stream,
                               ix->rx_slice_e0, ix->rx_pfreq, ix->rx_PF, ix->rx_slice_fmax);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(ix->stream));
        } else {
            (void)hipGetLastError();
            for (void **q : {(void **)&ix->rx_pstart16, (void **)&ix->rx_slice_e0, (void **)&ix->rx_slice_fmax}) {
                if (*q)
                    (void)hipFree(*q);
                *q = nullptr;
            }
        }
    }
    return KMM_OK;
}

// Radix-path view of the index (kmm_radix.hpp): entries regrouped in bucket order (whatever order the caller's
// hashes_to_index uses) + the exclusive prefix of the bucket sizes, which serves as the bucket directory of any
// 2^w-bucket slice.  Built from the raw arrays while they are still in HBM.
static int rx_build(kmm_index *ix, const int32_t *h2i, const int32_t *nk, const uint64_t *kmers,
                    const int32_t *nodes, const uint16_t *freqs)
{
    const uint64_t M = ix->modulo;
    if (const char *env = getenv("KMM_RX_FILTER")) // experiments: 0 = plain pass 2 and the fan-out chosen without the filter
        ix->rx_filter = atoi(env) != 0;
    ix->rx_why_not = 1;
    if (M >= (1ull << 31))
        return KMM_OK; // beyond the index format's int32 tables: no radix path; the direct path serves every batch
    DevBuf sizes;
    std::vector<DevBuf> scratch(2 * SCAN_MAX_LEVELS);
    int rc = KMM_OK;
    hipError_t e = hipSuccess;
    bool overlap = false;
    do {
        if ((rc = ensure(sizes, (size_t)(M + 1) * 4))) break;
        if ((e = hipMalloc(&ix->rx_pstart, (size_t)(M + 1) * 4))) break;
        hipLaunchKernelGGL(k_rx_bucket_sizes, dim3(grid_for(ix, (int64_t)((M + 256) / 256), 16)), dim3(256), 0,
                           ix->stream, h2i, nk, M, ix->n_entries, (uint32_t *)sizes.p);
        // an index whose buckets overlap (sum of the bucket sizes > n_entries: legal for the reference's loop, which only
        // follows (start, count) per bucket) has no bucket-ordered copy of bounded size, and a 32-bit prefix could
        // wrap: such an index is served by the direct path alone
        unsigned long long sum64 = 0;
        if ((e = hipMemsetAsync(ix->queue, 0, sizeof(unsigned long long), ix->stream))) break;
        hipLaunchKernelGGL(k_sum_u32, dim3(grid_for(ix, (int64_t)((M + 256) / 256), 8)), dim3(256), 0, ix->stream,
                           (const uint32_t *)sizes.p, M + 1, ix->queue);
        if ((e = hipMemcpyAsync(&sum64, ix->queue, 8, hipMemcpyDeviceToHost, ix->stream))) break;
        if ((e = hipStreamSynchronize(ix->stream))) break;
        if (sum64 > (unsigned long long)ix->n_entries) {
            overlap = true;
            break;
        }
        if ((rc = scan_exclusive((const uint32_t *)sizes.p, ix->rx_pstart, M + 1, scratch, 0, ix->stream))) break;
        uint32_t total = 0;
        if ((e = hipMemcpyAsync(&total, ix->rx_pstart + M, 4, hipMemcpyDeviceToHost, ix->stream))) break;
        if ((e = hipStreamSynchronize(ix->stream))) break;
        ix->rx_S = total; // = sum64 <= n_entries < 2^31
        {   // occupancy bitmap for pass 2's empty-bucket filter, padded by one coarse partition's worth of words
            // (k_rx_p2f loads whole partitions); optional: without the memory for it the plain pass 2 runs
            const size_t occ_words = (size_t)((M + 31) / 32) + ((size_t)1 << (P2F_LOGBITS + 2 - 5)); // (up to 4 buckets per LDS bit)
            if (hipMalloc(&ix->rx_occ, occ_words * 4) == hipSuccess) {
                if ((e = hipMemsetAsync(ix->rx_occ, 0, occ_words * 4, ix->stream))) break;
                hipLaunchKernelGGL(k_rx_build_occ, dim3(grid_for(ix, (int64_t)((M / 32 + 256) / 256), 16)), dim3(256), 0, ix->stream,
                                   ix->rx_pstart, M, ix->rx_occ);
            } else {
                (void)hipGetLastError();
                ix->rx_occ = nullptr;
            }
        }
        const size_t S = total ? total : 1;
        if ((e = hipMalloc(&ix->rx_pkeys, S * 8))) break;
        if ((e = hipMalloc(&ix->rx_pkeys_raw, S * 8))) break;
        if ((e = hipMalloc(&ix->rx_pfreq, S * 2))) break;
        if ((e = hipMalloc(&ix->rx_pnodes, S * 4))) break;
        if ((e = hipMalloc(&ix->rx_porig, S * 4))) break;
        if ((e = hipMalloc(&ix->rx_ecnt, S * 4))) break;
        if ((e = hipMemsetAsync(ix->rx_ecnt, 0, S * 4, ix->stream))) break;
        hipLaunchKernelGGL(k_rx_pack, dim3(grid_for(ix, (int64_t)((M + 255) / 256), 16)), dim3(256), 0, ix->stream, h2i,
                           kmers, nodes, freqs, M, ix->max_node_id, ix->rx_pstart, ix->rx_pkeys_raw, ix->rx_pfreq,
                           ix->rx_pnodes, ix->rx_porig);
        if ((e = hipGetLastError())) break;
        if ((e = hipStreamSynchronize(ix->stream))) break;
        // the entries once more in NODE order, for the flush (k_rx_flush_sorted): counting sort by node.  Optional:
        // without the memory for it the flush walks the entries in bucket order (k_rx_flush).
        // (few nodes with many entries each — a graph with 1000 hot nodes — aggregate in the bucket-order kernel's
        // LDS table instead: 1.1 ms against 2.3 ms per flush at 10^8 entries / 1000 nodes; 4.0 against 2.0 ms when
        // every entry has its own node)
        if (total) {
            const uint64_t n_nodes = (uint64_t)ix->max_node_id + 1;
            DevBuf hist, cursor;
            if (n_nodes + 1 < 0xFFFFFFFFull && (uint64_t)total / n_nodes < 8 && ensure(hist, (size_t)(n_nodes + 1) * 4) == KMM_OK &&
                ensure(cursor, (size_t)(n_nodes + 1) * 4) == KMM_OK &&
                hipMalloc(&ix->rx_norder, S * 4) == hipSuccess && hipMalloc(&ix->rx_nnode, S * 4) == hipSuccess) {
                bool ok = hipMemsetAsync(hist.p, 0, (size_t)(n_nodes + 1) * 4, ix->stream) == hipSuccess;
                hipLaunchKernelGGL(k_rx_node_hist, dim3(grid_for(ix, (int64_t)((S + 255) / 256), 16)), dim3(256), 0,
                                   ix->stream, ix->rx_pnodes, (uint64_t)total, (uint32_t *)hist.p);
                ok = ok && scan_exclusive((const uint32_t *)hist.p, (uint32_t *)cursor.p, n_nodes + 1, scratch, 0,
                                          ix->stream) == KMM_OK;
                hipLaunchKernelGGL(k_rx_node_scatter, dim3(grid_for(ix, (int64_t)((S + 255) / 256), 16)), dim3(256), 0,
                                   ix->stream, ix->rx_pnodes, (uint64_t)total, (uint32_t *)cursor.p, ix->rx_norder,
                                   ix->rx_nnode);
                ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(ix->stream) == hipSuccess;
                if (!ok) {
                    (void)hipFree(ix->rx_norder);
                    (void)hipFree(ix->rx_nnode);
                    ix->rx_norder = ix->rx_nnode = nullptr;
                }
            } else {
                (void)hipGetLastError();
                if (ix->rx_norder)
                    (void)hipFree(ix->rx_norder);
                if (ix->rx_nnode)
                    (void)hipFree(ix->rx_nnode);
                ix->rx_norder = ix->rx_nnode = nullptr;
            }
            release(hist);
            release(cursor);
        }
    } while (0);
    release(sizes);
    for (DevBuf &b : scratch)
        release(b);
    if (rc != KMM_OK || e != hipSuccess || overlap) {
        // the radix view is optional: without the memory for it (or for an index with overlapping buckets) the
        // direct path serves every batch; any other failure is an error
        const bool nomem = rc == KMM_ERR_NOMEM || e == hipErrorOutOfMemory;
        (void)hipGetLastError();
        for (void **q : {(void **)&ix->rx_pstart16, (void **)&ix->rx_slice_e0, (void **)&ix->rx_slice_fmax, (void **)&ix->rx_pstart, (void **)&ix->rx_pkeys, (void **)&ix->rx_pkeys_raw, (void **)&ix->rx_pfreq,
                         (void **)&ix->rx_pnodes, (void **)&ix->rx_porig, (void **)&ix->rx_ecnt, (void **)&ix->rx_norder,
                         (void **)&ix->rx_nnode, (void **)&ix->rx_occ}) {
            if (*q)
                (void)hipFree(*q);
            *q = nullptr;
        }
        ix->rx_S = 0;
        ix->rx_ok = false;
        if (overlap || nomem) {
            ix->rx_why_not = overlap ? 4 : 3;
            if (getenv("KMM_VERBOSE"))
                fprintf(stderr, "libkmm: radix view not built (%s): every batch takes the direct path\n",
                        overlap ? "the buckets of the index overlap" : "out of HBM");
            return KMM_OK;
        }
        if (rc != KMM_OK)
            return rc;
        return fail(KMM_ERR_HIP, "radix index build: %s", hipGetErrorString(e));
    }
    // 2^w buckets per fine partition: as many as keep a slice's entries (load factor x 2^w) well inside the LDS key
    // capacity; fewer when the table is dense.  The fan-out must fit 256 x 256 fine partitions.
    int w = 12;
    if (const char *env = getenv("KMM_RX_W")) // experiments / tests: force the slice width
        w = atoi(env);
    else
        while (w > 0 && ((double)ix->rx_S / (double)M * (double)(1u << w) * 1.3 + 64.0 > (double)RX_ECAP ||
                         (1ull << w) > M))
            --w;
    // Fan-out, in order of preference (runs between the passes get shorter, then pass 3 loses its second workgroup
    // per CU): up to 256 x 256 slices of 2^w buckets; 256 x 256 slices of 8192 buckets whose entries fit 4096 keys
    // (16-bit LDS directory, two workgroups of pass 3 per CU); up to 512 x 512 slices of 4096 buckets; slices of 8192
    // buckets with up to 8192 keys (one workgroup of pass 3 per CU), 256 x 256, then 512 x 512: every modulo below
    // 2^31 is covered as long as the load factor lets a slice's entries fit LDS.
    const int f2_force = getenv("KMM_RX_F2") ? atoi(getenv("KMM_RX_F2")) : -1; // experiments: fine-partition bits
    const double load = (double)ix->rx_S / (double)M;
    const bool fits13_small = load * 8192.0 * 1.3 + 64.0 <= (double)RX_ECAP;
    const bool fits13 = load * 8192.0 * 1.3 + 64.0 <= (double)RX_ECAP_BIG;
    ix->rx_ok = rx_configure(ix, w, f2_force >= 0 ? RX_MAXF : 256, f2_force);
    if (!ix->rx_ok && w == 12 && !getenv("KMM_RX_W")) {
        ix->rx_ok = (fits13_small && rx_configure(ix, 13, 256)) || rx_configure(ix, 12, RX_MAXF) ||
                    (fits13 && (rx_configure(ix, 13, 256) || rx_configure(ix, 13, RX_MAXF)));
    } else if (!ix->rx_ok) {
        ix->rx_ok = rx_configure(ix, w, RX_MAXF);
    }
    ix->rx_why_not = ix->rx_ok ? 0 : 2;
    if (getenv("KMM_VERBOSE"))
        fprintf(stderr, "libkmm: modulo %llu, %llu entries: radix path %s (2^%d buckets per slice, %u x %u partitions)\n",
                (unsigned long long)M, (unsigned long long)ix->rx_S, ix->rx_ok ? "available" : "NOT available: slices too "
                "dense for LDS or more than 512 x 512 of them", ix->rx_w, ix->rx_F1, ix->rx_F2);
    if (ix->rx_ok)
        KMMCHK(rx_repack_keys(ix));
    // auto: the radix path streams the whole directory + key arrays once per batch (4 B x modulo + 14 B x entries) and
    // then costs ~6.5 ps per k-mer (10 ps at the 1 B-k-mer index: shorter runs); the direct kernel has no fixed cost and
    // runs at ~60 G k-mers/s below ~1 GB of index, ~38 G above.  The batch size where the two meet
    // (profiles/r03/path_crossover.txt: 16-20 M positions at the 10 M index, ~52 M at the 100 M index; units are
    // base positions, 1.25 per k-mer at 150 bp):
    {
        const double bytes = (double)M * 4.0 + (double)ix->rx_S * 14.0;
        const double fixed = 50e-6 + bytes / 2.6e12;
        const double per_kmer_radix = 6.5e-12 * (1.0 + bytes / 40e9);
        const double per_kmer_direct = 1.0 / (bytes < 1e9 ? 60e9 : 38e9);
        ix->rx_min_units = (int64_t)(1.25 * fixed / (per_kmer_direct - per_kmer_radix));
    }
    if (ix->rx_min_units < ((int64_t)1 << 22))
        ix->rx_min_units = (int64_t)1 << 22;
    if (const char *env = getenv("KMM_RX_MIN_UNITS"))
        ix->rx_min_units = strtoll(env, nullptr, 10);
    return KMM_OK;
}

static bool ensure_ring(kmm_index_t *ix);

static int index_create_impl(kmm_index *ix, const int32_t *h2i, const int32_t *nk,
                             const uint64_t *kmers, const int32_t *nodes, const uint16_t *freqs)
{
    const uint64_t M = ix->modulo;
    const int64_t N = ix->n_entries;
    HIPCHK(hipSetDevice(ix->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, ix->device));
    ix->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ix->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ix->copied, hipEventDisableTiming));
    for (Stage &s : ix->stage)
        HIPCHK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));

    // layout: small indexes get 16-byte buckets + the L2 occupancy bitmap, larger ones 32-byte buckets
    size_t occ_max = KMM_OCC_MAX_BYTES;
    if (const char *env = getenv("KMM_OCC_MAX_BYTES")) // experiments: threshold of the bitmap prefilter
        occ_max = (size_t)strtoull(env, nullptr, 10);
    // bits per bucket: as many (1, 2, 4 or 8) as keep the bitmap within KMM_OCC_SWEET_BYTES — an entry sets
    // the bit chosen by its k-mer's fingerprint, so every doubling halves the false-positive passes of
    // single-entry buckets (10 M-k-mer index: 1 bit 19.7 ms, 2 bits 18.1 ms, 4 bits (10 MB) 21.5 ms per step)
    // (2.5 M: 4 bits/2.5 MB 14.5 ms, 8 bits/5 MB 15.3 ms; 5 M: 2 bits 16.1, 4 bits/5 MB 16.3, 8 bits/10 MB 20.6;
    //  15 M: 1 bit/3.75 MB 20.4, 2 bits/7.5 MB 21.3): the second bit is worth up to 5 MiB, further bits 3 MiB
    int occ_shift = 0;
    while (occ_shift < 3 &&
           ((M << (occ_shift + 1)) + 7) / 8 <= (occ_shift == 0 ? KMM_OCC_SWEET_BYTES : KMM_OCC_SWEET_BYTES * 3 / 5))
        occ_shift++;
    if (const char *env = getenv("KMM_OCC_SHIFT")) // experiments: force 2^shift bitmap bits per bucket
        occ_shift = atoi(env) < 0 ? 0 : (atoi(env) > 3 ? 3 : atoi(env));
    size_t occ_bytes = (size_t)(((M << occ_shift) + 31) / 32) * 4;
    const bool with_occ = (size_t)((M + 31) / 32) * 4 <= occ_max || occ_bytes <= occ_max;
    // Up to ~13 M entries a word-blocked Bloom filter (two bits per key inside one 32-bit word, chosen by a
    // hash of the k-mer alone) of at most 4 MiB beats the per-bucket bitmap: 10 M entries, ms per step:
    // bitmap 2 bits/bucket (5 MB) 18.0; Bloom 2.5 MB 19.6, 4 MB 16.7, 5 MB 17.0, 6.5 MB 18.1.  13 M: Bloom 4 MiB 18.3,
    // bitmap 19.5; 16 M: Bloom 6 MiB 19.9, bitmap 20.5; 20 M and 30 M: equal.
    int64_t bloom_max_entries = KMM_BLOOM_MAX_ENTRIES;
    if (const char *env = getenv("KMM_BLOOM_MAX_ENTRIES")) // experiments
        bloom_max_entries = strtoll(env, nullptr, 10);
    if (with_occ && N <= bloom_max_entries) {
        size_t bb = (size_t)N * 2;             // 16 bits per key is plenty
        const size_t cap = N <= 13000000 ? KMM_BLOOM_MAX_BYTES : KMM_BLOOM_MAX_BYTES * 3 / 2; // 13-20 M: 6 MiB
        if (bb > cap) bb = cap;
        if (bb < 64) bb = 64;
        if (const char *env = getenv("KMM_BLOOM_BYTES")) // experiments: filter size; 0 = per-bucket bitmap
            bb = (size_t)strtoull(env, nullptr, 10);
        if (bb >= 4) {
            ix->bloom_words = (uint32_t)(bb / 4);
            occ_bytes = (size_t)ix->bloom_words * 4;
        }
    }
    ix->occ_shift = occ_shift;
    ix->wide = !with_occ;
    if (const char *env = getenv("KMM_WIDE_BUCKETS")) // experiments: force the bucket layout (0 / 1)
        ix->wide = atoi(env) != 0;
    HIPCHK(hipMalloc(&ix->own_counts_buf, sizeof(uint32_t) * (size_t)(ix->max_node_id + 1)));
    ix->counts = ix->own_counts_buf;
    HIPCHK(hipMemsetAsync(ix->counts, 0, sizeof(uint32_t) * (size_t)(ix->max_node_id + 1), ix->stream));
    HIPCHK(hipMalloc(&ix->lut_default, 256));
    HIPCHK(hipMalloc(&ix->first_bad, 3 * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&ix->queue, 4 * sizeof(unsigned long long))); // (a tile queue head; also the result cells of small reductions)
    HIPCHK(hipMalloc(&ix->stats, KMM_STAT_BYTES));
    HIPCHK(hipMemset(ix->stats, 0, KMM_STAT_BYTES));
    uint8_t lut[256];
    default_lut(lut);
    HIPCHK(hipMemcpy(ix->lut_default, lut, 256, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&ix->lut_codes, 256));
    memset(lut, 0xFF, sizeof lut);
    lut[0] = 0; lut[1] = 1; lut[2] = 2; lut[3] = 3;
    HIPCHK(hipMemcpy(ix->lut_codes, lut, 256, hipMemcpyHostToDevice));
    unsigned long long nb[3] = {NO_BAD, NO_BAD, NO_BAD};
    HIPCHK(hipMemcpy(ix->first_bad, nb, sizeof nb, hipMemcpyHostToDevice));
    {   // host packing of reads that arrive in host memory: on by default with the reference CLI's worker count
        // (-t 16, command_line_interface.py:168) where the process has the cores for it — with fewer than 8 the plain
        // copy over PCIe is the faster way
        const int budget = kmm_hostpack::cpu_budget();
        ix->host_pack_threads = budget >= 8 ? (budget < 16 ? budget : 16) : 0;
        if (const char *env = getenv("KMM_HOST_PACK_THREADS")) {
            const int v = atoi(env);
            ix->host_pack_threads = v < 0 ? 0 : (v > 256 ? 256 : v);
        }
    }

    // The page-locked staging ring (8 x 16 MiB: 7-10 ms to make) is made HERE for an index whose count vector will leave
    // through it (kmm_get_node_counts -> pageable memory; flat reads and BGZF windows come in through it): a one-shot
    // `kmer_mapper map` otherwise makes it inside its first fetch — 18 ms instead of 8 for configs[2]'s 400 MB vector
    // (profiles/r05/cli_populate_ab.txt).  Best effort: without it the first user makes it.
    if (ix->host_pack_threads > 0 && (size_t)(ix->max_node_id + 1) * 4 >= ((size_t)64 << 20))
        (void)ensure_ring(ix);

    // raw arrays -> HBM (temporary); validate, build the radix view, and the direct view now or on first use
    DevBuf d_h2i, d_nk, d_km, d_nd, d_fr, d_err;
    bool staged = false;
    const int32_t *p_h2i = nullptr, *p_nk = nullptr, *p_nd = nullptr;
    const uint64_t *p_km = nullptr;
    const uint16_t *p_fr = nullptr;
    int rc = KMM_OK;
    do {
        if ((rc = stage_in<int32_t>(ix, d_h2i, h2i, (size_t)M, &p_h2i, &staged))) break;
        if ((rc = stage_in<int32_t>(ix, d_nk, nk, (size_t)M, &p_nk, &staged))) break;
        if ((rc = stage_in<uint64_t>(ix, d_km, kmers, (size_t)N, &p_km, &staged))) break;
        if ((rc = stage_in<int32_t>(ix, d_nd, nodes, (size_t)N, &p_nd, &staged))) break;
        if ((rc = stage_in<uint16_t>(ix, d_fr, freqs, (size_t)N, &p_fr, &staged))) break;
        if ((rc = ensure(d_err, 4))) break;
    } while (0);
    if (rc == KMM_OK) {
        hipError_t e = hipMemsetAsync(d_err.p, 0, 4, ix->copy_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->copy_stream);
        if (e != hipSuccess)
            rc = fail(KMM_ERR_HIP, "index upload: %s", hipGetErrorString(e));
    }
    uint32_t err = 0;
    if (rc == KMM_OK) { // what the reference never checks (mapper.pyx:17 disables bounds checks)
        hipLaunchKernelGGL(k_validate_index, dim3(grid_for(ix, (int64_t)((M + 255) / 256), 16)), dim3(256), 0, ix->stream,
                           p_h2i, p_nk, p_nd, M, N, ix->max_node_id, (uint32_t *)d_err.p);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        if (e == hipSuccess) e = hipMemcpy(&err, d_err.p, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            rc = fail(KMM_ERR_HIP, "index validation: %s", hipGetErrorString(e));
    }
    if (rc == KMM_OK && !err)
        rc = rx_build(ix, p_h2i, p_nk, p_km, p_nd, p_fr);
    // HBM budget: an index whose direct view is large (a 10^9-k-mer index: 64 GB of wide buckets + 16 GB of entries
    // beside the 46 GB radix view) keeps only the radix view resident; the direct view is packed from it when the
    // first small batch (or kmm_in_index) needs it.  Small indexes pack both now.
    size_t eager_max = (size_t)16 << 30;
    if (const char *env = getenv("KMM_DIRECT_EAGER_BYTES")) // tests / experiments
        eager_max = (size_t)strtoull(env, nullptr, 10);
    ix->direct_bytes = sizeof(uint4) * ((size_t)M * (ix->wide ? 2 : 1) + (size_t)(N > 0 ? N : 1));
    if (rc == KMM_OK && !err) {
        ix->occ_bytes_plan = with_occ ? occ_bytes : 0;
        if (ix->rx_ok && ix->direct_bytes > eager_max) {
            ix->direct_deferred = true;
        } else {
            IdxRaw src;
            src.h2i = p_h2i; src.nk = p_nk; src.kmers = p_km; src.nodes = p_nd; src.freqs = p_fr;
            rc = direct_build(ix, src, N, with_occ ? occ_bytes : 0, (uint32_t *)d_err.p);
        }
    }
    release(d_h2i); release(d_nk); release(d_km); release(d_nd); release(d_fr); release(d_err);
    if (rc != KMM_OK)
        return rc;
    if (err & 1u)
        return fail(KMM_ERR_INDEX, "index inconsistent: a non-empty bucket "
                    "(hashes_to_index[h], n_kmers[h]) reaches outside [0, n_entries=%lld)",
                    (long long)N);
    if (err & 2u)
        return fail(KMM_ERR_INDEX, "index inconsistent: a node id lies outside [0, max_node_id=%lld]",
                    (long long)ix->max_node_id);
    return KMM_OK;
}

int kmm_index_create(const int32_t *hashes_to_index, const int32_t *n_kmers, uint64_t modulo,
                     const uint64_t *kmers, const int32_t *nodes, const uint16_t *frequencies,
                     int64_t n_entries, int64_t max_node_id, int device, kmm_index_t **out)
{
    if (!out)
        return fail(KMM_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!hashes_to_index || !n_kmers)
        return fail(KMM_ERR_INVALID_ARG, "hashes_to_index / n_kmers is NULL");
    if (modulo < 1)
        return fail(KMM_ERR_INVALID_ARG, "modulo must be >= 1");
    if (n_entries < 0 || max_node_id < 0)
        return fail(KMM_ERR_INVALID_ARG, "n_entries=%lld / max_node_id=%lld negative",
                    (long long)n_entries, (long long)max_node_id);
    if (n_entries > 0 && (!kmers || !nodes || !frequencies))
        return fail(KMM_ERR_INVALID_ARG, "kmers / nodes / frequencies is NULL");
    if (n_entries > 0x7FFFFFFFll)
        return fail(KMM_ERR_INVALID_ARG, "n_entries exceeds the int32 bucket offsets of the index format");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        return fail(KMM_ERR_HIP, "no HIP device available (%s): libkmm has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "0 devices");
    }
    if (device < 0 || device >= ndev)
        return fail(KMM_ERR_INVALID_ARG, "device %d outside [0, %d)", device, ndev);
    kmm_index *ix = new kmm_index();
    ix->device = device;
    ix->modulo = modulo;
    ix->magic = magic_for(modulo);
    ix->n_entries = n_entries;
    ix->max_node_id = max_node_id;
    int rc = index_create_impl(ix, hashes_to_index, n_kmers, kmers, nodes, frequencies);
    if (rc != KMM_OK) {
        std::string keep = g_err;
        kmm_index_destroy(ix);
        g_err = keep;
        return rc;
    }
    *out = ix;
    return KMM_OK;
}

int kmm_reset_counts(kmm_index_t *ix)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipMemsetAsync(ix->counts, 0, sizeof(uint32_t) * (size_t)(ix->max_node_id + 1), ix->stream));
    if (ix->rx_ecnt && ix->ecnt_dirty)
        HIPCHK(hipMemsetAsync(ix->rx_ecnt, 0, sizeof(uint32_t) * (size_t)(ix->rx_S ? ix->rx_S : 1), ix->stream));
    if (ix->rx_ecnt_acc)
        HIPCHK(hipMemsetAsync(ix->rx_ecnt_acc, 0, sizeof(uint32_t) * (size_t)(ix->rx_S ? ix->rx_S : 1), ix->stream));
    ix->ecnt_dirty = false;
    if (ix->sticky_rc != KMM_OK) { // the error of a mapped chunk goes away together with its partial counts
        unsigned long long nb[3] = {NO_BAD, NO_BAD, NO_BAD};
        HIPCHK(hipStreamSynchronize(ix->stream));
        HIPCHK(hipMemcpy(ix->first_bad, nb, sizeof nb, hipMemcpyHostToDevice));
        if (ix->sticky_rc == KMM_ERR_INTERNAL) // the conservation counters restart with the counts
            HIPCHK(hipMemset(ix->stats, 0, KMM_STAT_BYTES));
        ix->sticky_rc = KMM_OK;
        ix->sticky_msg.clear();
    }
    return KMM_OK;
}

int kmm_bind_counts(kmm_index_t *ix, uint32_t *device_counts)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    HIPCHK(hipSetDevice(ix->device));
    KMMCHK(rx_flush(ix)); // hits mapped so far belong to the buffer that was bound when they were mapped
    HIPCHK(hipStreamSynchronize(ix->stream));
    if (!device_counts) {
        ix->counts = ix->own_counts_buf;
        return KMM_OK;
    }
    if (!is_device_ptr(device_counts))
        return fail(KMM_ERR_INVALID_ARG, "kmm_bind_counts needs a device pointer");
    ix->counts = device_counts;
    return KMM_OK;
}

int kmm_counts_device_ptr(kmm_index_t *ix, uint32_t **out)
{
    if (!ix || !out)
        return fail(KMM_ERR_INVALID_ARG, "NULL argument");
    *out = ix->counts;
    return KMM_OK;
}

int kmm_synchronize(kmm_index_t *ix)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    HIPCHK(hipSetDevice(ix->device));
    return drain(ix);
}

static bool ensure_pack_pool(kmm_index_t *ix);
static bool ensure_pinned(uint8_t *&p, size_t &have, size_t want);

// The staging ring: RING_SLOTS page-locked buffers of RING_SLOT bytes, each an allocation of its own.  Host threads fill a
// slot, a copy engine empties it (or the other way round), and a slot is touched by one side at a time.  Why a ring and not
// one buffer the size of the batch: 128 MB are made in 7 ms, a 750 MB batch buffer in 40 — as long as the map phase of a
// whole file.  (It does not make the threads faster: 16 threads pack 300 GB/s into page-locked memory with no copy in
// flight and 200 GB/s with one, whether the copies read the buffer being filled, slots of a ring of 8 or of a ring of 64
// that never makes anyone wait — profiles/r05/pack_without_copies.txt, host_membw_dma.txt; at 200 GB/s the packing of a
// batch takes as long as its 2-bit stream needs to cross PCIe, which is what bounds the leg.)
static bool ensure_ring(kmm_index_t *ix)
{
    for (int i = 0; i < RING_SLOTS; ++i) {
        if (ix->ring[i])
            continue;
        size_t got = 0;
        uint8_t *p = g_shelf.take(RING_SLOT, &got);
        if (p && got != RING_SLOT) { // (a larger buffer someone reserved: not for a slot)
            g_shelf.give(p, got);
            p = nullptr;
        }
        if (!p && hipHostMalloc(reinterpret_cast<void **>(&p), RING_SLOT, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        ix->ring[i] = p;
    }
    for (int i = 0; i < RING_SLOTS; ++i)
        if (!ix->bgzf_slot_ev[i] && hipEventCreateWithFlags(&ix->bgzf_slot_ev[i], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
    return true;
}

// HBM -> PAGEABLE host memory (a fresh numpy array: what get_node_counts hands to the reference's caller).  The runtime's own
// path for pageable memory moves ~10 GB/s and a page-locked landing buffer costs ~50 ms per GB to make — either way tens of
// milliseconds for configs[2]'s 400 MB vector, as long as the whole map phase of a 3 GB FASTQ.  So: through the handle's
// page-locked ring (8 slots of 16 MiB, the one kmm_map_bgzf stages through), the packing threads copying a slot out —
// and taking the destination's first-touch page faults — while the next slots are in flight: PCIe rate.
static int fetch_to_pageable(kmm_index_t *ix, uint8_t *dst, const uint8_t *src, size_t bytes, bool *done)
{
    *done = false;
    constexpr size_t SLOT = RING_SLOT;
    constexpr int SLOTS = RING_SLOTS;
    if (bytes < 4 * SLOT || !ensure_pack_pool(ix) || !ensure_ring(ix))
        return KMM_OK;
    const size_t n_slots = (bytes + SLOT - 1) / SLOT;
    auto issue = [&](size_t c) -> int {
        const size_t b0 = c * SLOT, len = bytes - b0 < SLOT ? bytes - b0 : SLOT;
        HIPCHK(hipMemcpyAsync(ix->ring[c % SLOTS], src + b0, len, hipMemcpyDeviceToHost, ix->copy_stream));
        HIPCHK(hipEventRecord(ix->bgzf_slot_ev[c % SLOTS], ix->copy_stream));
        return KMM_OK;
    };
    for (size_t c = 0; c < n_slots && c < (size_t)SLOTS; ++c)
        KMMCHK(issue(c));
    const int T = ix->pack_pool->size();
    for (size_t c = 0; c < n_slots; ++c) {
        HIPCHK(hipEventSynchronize(ix->bgzf_slot_ev[c % SLOTS]));
        const size_t b0 = c * SLOT, len = bytes - b0 < SLOT ? bytes - b0 : SLOT;
        const uint8_t *from = ix->ring[c % SLOTS];
        const size_t per = ((len + (size_t)T - 1) / (size_t)T + 63) & ~(size_t)63;
        ix->pack_pool->start([=](int w) {
            const size_t a = (size_t)w * per;
            if (a < len)
                memcpy(dst + b0 + a, from + a, len - a < per ? len - a : per);
        });
        ix->pack_pool->wait();
        if (c + SLOTS < n_slots)
            KMMCHK(issue(c + SLOTS));
    }
    *done = true;
    return KMM_OK;
}

int kmm_get_node_counts(kmm_index_t *ix, uint32_t *out)
{
    if (!ix || !out)
        return fail(KMM_ERR_INVALID_ARG, "NULL argument");
    HIPCHK(hipSetDevice(ix->device));
    KMMCHK(drain(ix));
    const size_t bytes = sizeof(uint32_t) * (size_t)(ix->max_node_id + 1);
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    const bool known = hipPointerGetAttributes(&a, out) == hipSuccess;
    if (!known)
        (void)hipGetLastError(); // (ordinary host memory the runtime has never seen)
    const bool on_device = known && (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged);
    const bool page_locked = known && a.type == hipMemoryTypeHost;
    if (!on_device && !page_locked) {
        bool done = false;
        KMMCHK(guarded("kmm_get_node_counts", [&] {
            return fetch_to_pageable(ix, reinterpret_cast<uint8_t *>(out), reinterpret_cast<const uint8_t *>(ix->counts), bytes, &done);
        }));
        if (done)
            return KMM_OK;
    }
    HIPCHK(hipMemcpy(out, ix->counts, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return KMM_OK;
}

int kmm_comm_get_unique_id(uint8_t id[KMM_COMM_ID_BYTES])
{
    if (!id)
        return fail(KMM_ERR_INVALID_ARG, "id is NULL");
    static_assert(KMM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "kmm.h and rccl.h disagree on the id size");
    KMMCHK(rccl_load());
    ncclUniqueId u;
    RCCLCHK(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, KMM_COMM_ID_BYTES);
    return KMM_OK;
}

int kmm_comm_init_rank(kmm_index_t *ix, const uint8_t id[KMM_COMM_ID_BYTES], int n_ranks, int rank)
{
    if (!ix || !id)
        return fail(KMM_ERR_INVALID_ARG, "NULL argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(KMM_ERR_INVALID_ARG, "rank %d outside [0, %d)", rank, n_ranks);
    KMMCHK(rccl_load());
    HIPCHK(hipSetDevice(ix->device));
    if (ix->comm) {
        RCCLCHK(g_rccl.CommDestroy(ix->comm));
        ix->comm = nullptr;
    }
    ncclUniqueId u;
    memcpy(u.internal, id, KMM_COMM_ID_BYTES);
    RCCLCHK(g_rccl.CommInitRank(&ix->comm, n_ranks, u, rank));
    ix->comm_rank = rank;
    ix->comm_size = n_ranks;
    return KMM_OK;
}

int kmm_comm_reduce_counts(kmm_index_t *ix, int root)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    if (!ix->comm)
        return fail(KMM_ERR_INVALID_ARG, "kmm_comm_init_rank has not been called on this handle");
    if (root < -1 || root >= ix->comm_size)
        return fail(KMM_ERR_INVALID_ARG, "root %d outside [-1, %d)", root, ix->comm_size);
    HIPCHK(hipSetDevice(ix->device));
    const size_t n = (size_t)ix->max_node_id + 1;
    auto reduce = [&](size_t first, size_t count, hipStream_t st) {
        // uint32 addition wraps modulo 2^32 like mapper.pyx:37,68, whatever the order of the reduction
        return root < 0 ? g_rccl.AllReduce(ix->counts + first, ix->counts + first, count, ncclUint32, ncclSum, ix->comm, st)
                        : g_rccl.Reduce(ix->counts + first, ix->counts + first, count, ncclUint32, ncclSum, root, ix->comm, st);
    };
    // The tail of a multi-GPU job is flush (per-entry hits -> node counts: 2 ms at the 100 M index, 25 ms at 10^9 entries)
    // + the reduce of 4 (max_node_id + 1) bytes.  With the entries listed in node order the flush of one node RANGE only
    // writes that range of the count vector, so range s is flushed while range s - 1 travels: the ranges' reduces go to
    // a second stream, each behind the event of its flush.  Every rank issues the same sequence of reduces.
    const int S = ix->comm_slices;
    // How many collectives a rank issues must not depend on anything rank-local — what THIS rank has mapped, whether ITS
    // node-ordered entry list could be allocated ("absent if memory is short", rx_build), per-handle modes: a rank that
    // issued one reduce of n elements against its peers' S reduces of n / S would hang the job or corrupt the counts.
    // Only the vector's length and "comm_overlap_slices" (the same on every rank: a parameter of the job) decide; a rank
    // that cannot flush by node range flushes everything first and then issues the same S range reduces.
    if (S > 1 && n >= (size_t)S * 1024) {
        const bool by_range = ix->rx_norder && !ix->rx_ecnt_acc && ix->rx_flush_sorted;
        ix->comm_sliced_reduces++;
        if (!ix->comm_stream)
            HIPCHK(hipStreamCreateWithFlags(&ix->comm_stream, hipStreamNonBlocking));
        while ((int)ix->comm_events.size() < S + 1) {
            hipEvent_t e;
            HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ix->comm_events.push_back(e);
        }
        if (!by_range)
            KMMCHK(rx_flush(ix)); // (every hit is in `counts`; the ranges below then only travel)
        if (by_range && (int)ix->flush_cuts.size() != S + 1) { // once per handle: where the node ranges begin in the node-ordered list
            std::vector<uint32_t> bounds(S + 1);
            for (int t = 0; t <= S; ++t)
                bounds[t] = (uint32_t)(n * (size_t)t / (size_t)S);
            DevBuf d_b, d_c;
            KMMCHK(ensure(d_b, (S + 1) * 4));
            KMMCHK(ensure(d_c, (S + 1) * 8));
            HIPCHK(hipMemcpyAsync(d_b.p, bounds.data(), (S + 1) * 4, hipMemcpyHostToDevice, ix->stream));
            hipLaunchKernelGGL(k_rx_node_cuts, dim3(1), dim3(64), 0, ix->stream, ix->rx_nnode, ix->rx_S, (const uint32_t *)d_b.p, S + 1,
                               (unsigned long long *)d_c.p);
            HIPCHK(hipGetLastError());
            ix->flush_cuts.assign(S + 1, 0);
            HIPCHK(hipMemcpyAsync(ix->flush_cuts.data(), d_c.p, (S + 1) * 8, hipMemcpyDeviceToHost, ix->stream));
            HIPCHK(hipStreamSynchronize(ix->stream));
            release(d_b);
            release(d_c);
            ix->flush_cuts[0] = 0;
            ix->flush_cuts[S] = ix->rx_S;
        }
        ScopedTimer tm;
        KMMCHK(tm.begin(ix, KMM_KERNEL_RX_FLUSH));
        for (int t = 0; t < S; ++t) {
            if (by_range && ix->ecnt_dirty) {
                const uint64_t j0 = ix->flush_cuts[t], j1 = ix->flush_cuts[t + 1];
                if (j1 > j0)
                    hipLaunchKernelGGL(k_rx_flush_sorted, dim3(grid_for(ix, (int64_t)((j1 - j0 + 1023) / 1024), 8)), dim3(256), 0, ix->stream,
                                       view_of(ix), ix->rx_ecnt, ix->rx_norder + j0, ix->rx_nnode + j0, j1 - j0);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipEventRecord(ix->comm_events[t], ix->stream));
            HIPCHK(hipStreamWaitEvent(ix->comm_stream, ix->comm_events[t], 0));
            const size_t first = n * (size_t)t / (size_t)S, end = n * (size_t)(t + 1) / (size_t)S;
            RCCLCHK(reduce(first, end - first, ix->comm_stream));
        }
        if (by_range && ix->ecnt_dirty)
            HIPCHK(hipMemsetAsync(ix->rx_ecnt, 0, (size_t)ix->rx_S * 4, ix->stream));
        KMMCHK(tm.end());
        ix->ecnt_dirty = false;
        HIPCHK(hipEventRecord(ix->comm_events[S], ix->comm_stream));
        HIPCHK(hipStreamWaitEvent(ix->stream, ix->comm_events[S], 0));
        return drain(ix);
    }
    KMMCHK(rx_flush(ix)); // every hit mapped so far is in `counts` before it travels
    RCCLCHK(reduce(0, n, ix->stream));
    return drain(ix);
}

int kmm_comm_destroy(kmm_index_t *ix)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    if (ix->comm) {
        HIPCHK(hipSetDevice(ix->device));
        HIPCHK(hipStreamSynchronize(ix->stream));
        RCCLCHK(g_rccl.CommDestroy(ix->comm));
        ix->comm = nullptr;
        ix->comm_rank = -1;
        ix->comm_size = 0;
    }
    return KMM_OK;
}

int kmm_reduce_counts(kmm_index_t **per_gpu, int n_gpus, int root)
{
    if (!per_gpu || n_gpus < 1)
        return fail(KMM_ERR_INVALID_ARG, "per_gpu is NULL or n_gpus < 1");
    if (root < -1 || root >= n_gpus)
        return fail(KMM_ERR_INVALID_ARG, "root %d outside [-1, %d)", root, n_gpus);
    std::vector<int> devs(n_gpus);
    for (int i = 0; i < n_gpus; ++i) {
        if (!per_gpu[i])
            return fail(KMM_ERR_INVALID_ARG, "per_gpu[%d] is NULL", i);
        if (per_gpu[i]->max_node_id != per_gpu[0]->max_node_id)
            return fail(KMM_ERR_INVALID_ARG, "handles disagree on max_node_id");
        devs[i] = per_gpu[i]->device;
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i])
                return fail(KMM_ERR_INVALID_ARG, "handles %d and %d share device %d: one handle per GPU", j, i, devs[i]);
    }
    if (n_gpus == 1) {
        HIPCHK(hipSetDevice(per_gpu[0]->device));
        return drain(per_gpu[0]);
    }
    KMMCHK(rccl_load());
    std::vector<ncclComm_t> comms(n_gpus);
    RCCLCHK(g_rccl.CommInitAll(comms.data(), n_gpus, devs.data()));
    int rc = KMM_OK;
    const size_t n = (size_t)per_gpu[0]->max_node_id + 1;
    for (int i = 0; i < n_gpus && rc == KMM_OK; ++i) {
        if (hipSetDevice(devs[i]) != hipSuccess)
            rc = fail(KMM_ERR_HIP, "hipSetDevice(%d) failed", devs[i]);
        else
            rc = rx_flush(per_gpu[i]);
    }
    if (rc == KMM_OK) {
        ncclResult_t r = g_rccl.GroupStart();
        for (int i = 0; i < n_gpus && r == ncclSuccess; ++i) {
            kmm_index *ix = per_gpu[i];
            r = root < 0 ? g_rccl.AllReduce(ix->counts, ix->counts, n, ncclUint32, ncclSum, comms[i], ix->stream)
                         : g_rccl.Reduce(ix->counts, ix->counts, n, ncclUint32, ncclSum, root, comms[i], ix->stream);
        }
        const ncclResult_t r2 = g_rccl.GroupEnd();
        if (r == ncclSuccess)
            r = r2;
        if (r != ncclSuccess)
            rc = fail(KMM_ERR_HIP, "RCCL reduce of the node counts: %s", g_rccl.GetErrorString(r));
    }
    for (int i = 0; i < n_gpus; ++i) {
        (void)hipSetDevice(devs[i]);
        const int d = drain(per_gpu[i]);
        if (rc == KMM_OK)
            rc = d;
        (void)g_rccl.CommDestroy(comms[i]);
    }
    return rc;
}

int kmm_get_kmer_counts(kmm_index_t *ix, uint32_t *out)
{
    if (!ix || !out)
        return fail(KMM_ERR_INVALID_ARG, "NULL argument");
    if (!ix->rx_ecnt_acc)
        return fail(KMM_ERR_INVALID_ARG, "per-k-mer counts are only kept in count_kmers mode "
                    "(kmm_set_param(idx, \"count_kmers\", 1) before mapping)");
    HIPCHK(hipSetDevice(ix->device));
    KMMCHK(drain(ix));
    const size_t n = (size_t)ix->n_entries;
    if (n == 0)
        return KMM_OK;
    const bool out_dev = is_device_ptr(out);
    DevBuf tmp;
    uint32_t *d_out = out;
    if (!out_dev) {
        KMMCHK(ensure(tmp, n * 4));
        d_out = (uint32_t *)tmp.p;
    }
    hipError_t e = hipMemsetAsync(d_out, 0, n * 4, ix->stream); // entries that no bucket references stay 0
    if (e == hipSuccess && ix->rx_S) {
        hipLaunchKernelGGL(k_rx_entry_counts, dim3(grid_for(ix, (int64_t)((ix->rx_S + 255) / 256), 16)), dim3(256), 0,
                           ix->stream, ix->rx_ecnt_acc, ix->rx_porig, ix->rx_S, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !out_dev)
        e = hipMemcpyAsync(out, d_out, n * 4, hipMemcpyDeviceToHost, ix->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ix->stream);
    release(tmp);
    if (e != hipSuccess)
        return fail(KMM_ERR_HIP, "kmm_get_kmer_counts: %s", hipGetErrorString(e));
    return KMM_OK;
}

int kmm_map_kmers(kmm_index_t *ix, const uint64_t *kmers, int64_t n, int max_freq, int also_revcomp,
                  int k)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    if (n < 0 || (n > 0 && !kmers))
        return fail(KMM_ERR_INVALID_ARG, "kmers NULL or n negative");
    if (also_revcomp)
        KMMCHK(check_k(k));
    if (n == 0)
        return KMM_OK;
    HIPCHK(hipSetDevice(ix->device));
    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    bool staged = false;
    const uint64_t *d_kmers = nullptr;
    KMMCHK(stage_in<uint64_t>(ix, s.kmers, kmers, (size_t)n, &d_kmers, &staged));
    KMMCHK(stage_copies_done(ix));
    ix->map_calls++;
    if (use_radix(ix, n)) {
        ReadsView rv;
        memset(&rv, 0, sizeof rv);
        rv.first_bad = ix->first_bad;
        KMMCHK(launch_rx<MODE_KMERS>(ix, rv, d_kmers, n, k, max_freq, also_revcomp ? 1 : 0));
        return stage_release(ix, s, staged);
    }
    constexpr int U = 8;
    KMMCHK(ensure_direct(ix));
    ix->n_direct_batches++;
    ScopedTimer tm;
    KMMCHK(tm.begin(ix, KMM_KERNEL_MAP_KMERS));
    {
        const IndexView iv = view_of(ix);
        const int64_t n_spans = (n + 256 * U - 1) / (256 * U);
        const int64_t slots = (int64_t)ix->n_cu * 8;
        const bool dynamic = ix->dynamic_schedule && n_spans >= slots * 4 * ix->dyn_chunk;
        if (dynamic)
            HIPCHK(hipMemsetAsync(ix->queue, 0, 3 * sizeof(unsigned long long), ix->stream));
        const dim3 grid(dynamic ? (unsigned)slots : (unsigned)grid_for_tiles(ix, n_spans));
        unsigned long long *queue = dynamic ? ix->queue : nullptr;
        if (iv.occ && iv.wide)
            hipLaunchKernelGGL((k_map_kmers<U, PROBE_WIDE_FILTER>), grid, dim3(256), 0, ix->stream, d_kmers, n, iv,
                               max_freq, also_revcomp ? 1 : 0, k, queue, ix->dyn_chunk);
        else if (iv.occ)
            hipLaunchKernelGGL((k_map_kmers<U, PROBE_BITMAP>), grid, dim3(256), 0, ix->stream, d_kmers, n, iv,
                               max_freq, also_revcomp ? 1 : 0, k, queue, ix->dyn_chunk);
        else if (iv.wide)
            hipLaunchKernelGGL((k_map_kmers<U, PROBE_WIDE>), grid, dim3(256), 0, ix->stream, d_kmers, n, iv,
                               max_freq, also_revcomp ? 1 : 0, k, queue, ix->dyn_chunk);
        else
            hipLaunchKernelGGL((k_map_kmers<U, PROBE_NARROW>), grid, dim3(256), 0, ix->stream, d_kmers, n, iv,
                               max_freq, also_revcomp ? 1 : 0, k, queue, ix->dyn_chunk);
    }
    HIPCHK(hipGetLastError());
    KMMCHK(tm.end());
    return stage_release(ix, s, staged);
}

// Reads of one length: the uniform front end's constants and, where they apply, the packed tiles of the radix path's
// pass 1 (kmm_tile.hpp): pk_lpr lanes per read, pk_S windows each.
static void set_uniform_geometry(const kmm_index_t *ix, ReadsView &rv, int64_t read_len, int k)
{
    rv.read_len = (uint64_t)read_len;
    rv.read_len_magic = magic_for((uint64_t)read_len);
    if (ix->rx_packed && read_len >= k && read_len - k + 1 <= 4096) {
        const uint32_t W = (uint32_t)(read_len - k + 1), lpr = (W + 15u) / 16u, rpt = 256u / lpr;
        if (rpt >= 1u && (uint64_t)rpt * (uint64_t)read_len <= 8176u) {
            rv.pk_rpt = rpt;
            rv.pk_lpr = lpr;
            rv.pk_S = (W + lpr - 1u) / lpr;
            rv.pk_W = W;
            rv.pk_inv = (65536u + lpr - 1u) / lpr;
        }
    }
}

static int rec_launch_flat(kmm_index_t *ix, const uint32_t *flat, int64_t total, int64_t n_reads, const uint32_t *start_bits,
                           int64_t n_words, int64_t uniform_len, int k, int max_freq, int also_revcomp);

// The handle's packing threads ("host_pack_threads"), created at first use and kept asleep between calls.  false: no
// threads to be had (the caller takes the ordinary route) — nothing thrown by the thread library crosses the C ABI.
static bool ensure_pack_pool(kmm_index_t *ix)
{
    if (ix->host_pack_threads < 1)
        return false;
    if (ix->pack_pool && ix->pack_pool->size() == ix->host_pack_threads)
        return true;
    try {
        ix->pack_pool.reset();
        ix->pack_pool.reset(new kmm_hostpack::Workers(ix->host_pack_threads));
    } catch (...) {
        ix->pack_pool.reset();
        return false;
    }
    return true;
}

// page-locked home of a packed batch: grown, never shrunk; false = none to be had
static bool ensure_pinned(uint8_t *&p, size_t &have, size_t want)
{
    if (have >= want)
        return true;
    g_shelf.give(p, have);
    p = nullptr;
    have = 0;
    if ((p = g_shelf.take(want, &have)))
        return true;
    const size_t take = want + want / 8;
    if (hipHostMalloc(reinterpret_cast<void **>(&p), take, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        p = nullptr;
        return false;
    }
    have = take;
    return true;
}

// Flat reads in HOST memory, default lookup table, a batch of radix size, "host_pack_threads" > 0: packed to 2 bits
// per base by that many host threads into a page-locked buffer, chunk by chunk, each chunk copied to HBM as soon as it is
// packed (copy stream: under the previous call's kernels and under the packing of the next chunks), then mapped like the
// flat reads the records compaction makes (pass 1 on 2-bit codes).  *done = false: a byte outside the table — the caller
// takes the ordinary route, whose kernels report the byte's offset (mapper semantics unchanged: nothing was mapped here).
static int map_reads_host_packed(kmm_index_t *ix, const uint8_t *bases, const int64_t *read_offsets, int64_t total_bases,
                                 int64_t n_reads, int64_t read_len, int k, int max_freq, int also_revcomp, bool *done)
{
    *done = false;
    const size_t total = (size_t)total_bases;
    const size_t packed_total = (total + 3) / 4, code_bytes = (packed_total + 256 + 63) & ~(size_t)63;
    if (!ensure_pack_pool(ix) || !ensure_ring(ix))
        return KMM_OK; // (no threads / no page-locked memory to be had: the ordinary route)
    static const bool verbose = getenv("KMM_VERBOSE") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    KMMCHK(ensure(s.kmers, code_bytes));
    const double ms_stage = ms_since(t_0);
    const auto t_1 = std::chrono::steady_clock::now();
    // Tasks of 4 Mi bases (1 MiB packed) handed out in order; 16 of them fill a slot of the staging ring, a full slot leaves
    // for HBM (copy stream: under the previous call's kernels and under the packing of the next slots) and is written again
    // when its copy has landed.
    // ("debug_ring_slot_kb": tests wrap the ring many times with a small batch)
    const size_t SLOT_BYTES = ix->dbg_bgzf_slot_kb > 0 && ((size_t)ix->dbg_bgzf_slot_kb << 10) < RING_SLOT ? (size_t)ix->dbg_bgzf_slot_kb << 10 : RING_SLOT;
    const size_t CHUNK = 4 * (SLOT_BYTES < ((size_t)1 << 20) ? SLOT_BYTES : (size_t)1 << 20), PER_SLOT = SLOT_BYTES / (CHUNK / 4);
    constexpr int SLOTS = RING_SLOTS;
    const size_t n_chunks = (total + CHUNK - 1) / CHUNK, n_slots = (n_chunks + PER_SLOT - 1) / PER_SLOT;
    std::vector<std::atomic<uint32_t>> filled(n_slots);
    for (auto &f : filled)
        f.store(0, std::memory_order_relaxed);
    std::atomic<size_t> next{0}, slots_free{(size_t)SLOTS};
    std::atomic<bool> stop{false}, bad{false};
    uint8_t *const *ring = ix->ring;
    ix->pack_pool->start([&](int) {
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= n_chunks)
                return;
            const size_t piece = c / PER_SLOT;
            int spins = 0;
            while (piece >= slots_free.load(std::memory_order_acquire) && !stop.load(std::memory_order_relaxed)) {
                if (++spins < 4000) {
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                } else {
                    std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
            }
            if (stop.load(std::memory_order_relaxed))
                return;
            const size_t b0 = c * CHUNK, len = total - b0 < CHUNK ? total - b0 : CHUNK;
            if (!bad.load(std::memory_order_relaxed) && !kmm_hostpack::pack2(bases + b0, len, ring[piece % SLOTS] + (c % PER_SLOT) * (CHUNK / 4)))
                bad.store(true);
            filled[piece].fetch_add(1, std::memory_order_release);
        }
    });
    int rc = KMM_OK;
    size_t landed = 0; // slot-sized pieces whose copy to HBM is known to have finished
    double ms_in_copy_calls = 0;
    // (every second slot on a second copy stream: 141 against 146 G k-mers/s, profiles/r05/h2d_two_copy_streams.txt — one stream)
    for (size_t c = 0; c < n_slots && rc == KMM_OK && !bad.load(); ++c) {
        const size_t first = c * PER_SLOT, want = n_chunks - first < PER_SLOT ? n_chunks - first : PER_SLOT;
        int idle = 0;
        while (filled[c].load(std::memory_order_acquire) < (uint32_t)want) {
            bool did = false;
            while (landed + SLOTS < n_slots && landed < c) { // slots whose copies have landed go back to the threads
                if (hipEventQuery(ix->bgzf_slot_ev[landed % SLOTS]) != hipSuccess) {
                    (void)hipGetLastError(); // ("not ready" is no error to keep)
                    break;
                }
                ++landed;
                slots_free.store(landed + SLOTS, std::memory_order_release);
                did = true;
            }
            if (!did) {
                if (++idle < 2000) {
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                } else {
                    std::this_thread::sleep_for(std::chrono::microseconds(20));
                }
            }
        }
        if (bad.load())
            break;
        const size_t b0 = c * SLOT_BYTES, len = packed_total - b0 < SLOT_BYTES ? packed_total - b0 : SLOT_BYTES;
        const auto t_c = std::chrono::steady_clock::now();
#ifdef KMM_EXPERIMENT_PACK_WITHOUT_COPIES // (tools/ab_build.sh only: how fast do the threads pack when nothing is copied? results are garbage)
        (void)b0; (void)len;
        if (hipEventRecord(ix->bgzf_slot_ev[c % SLOTS], ix->copy_stream) != hipSuccess)
            rc = fail(KMM_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(hipGetLastError()));
#else
        hipStream_t cs = ix->copy_stream;
        if (hipMemcpyAsync((uint8_t *)s.kmers.p + b0, ring[c % SLOTS], len, hipMemcpyHostToDevice, cs) != hipSuccess ||
            hipEventRecord(ix->bgzf_slot_ev[c % SLOTS], cs) != hipSuccess)
            rc = fail(KMM_ERR_HIP, "copy of packed reads: %s", hipGetErrorString(hipGetLastError()));
#endif
        if (verbose)
            ms_in_copy_calls += ms_since(t_c);
        while (rc == KMM_OK && landed + SLOTS < n_slots && landed + SLOTS <= c + 1) { // the ring is full: the oldest copy is waited for
            if (hipEventSynchronize(ix->bgzf_slot_ev[landed % SLOTS]) != hipSuccess) {
                rc = fail(KMM_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(hipGetLastError()));
                break;
            }
            ++landed;
            slots_free.store(landed + SLOTS, std::memory_order_release);
        }
    }
    if (rc != KMM_OK || bad.load())
        stop.store(true);
    ix->pack_pool->wait();
    if (verbose)
        fprintf(stderr, "libkmm: host flat packer: %zu bases, waited %.2f ms for the stage, pack + copies issued %.2f ms (%.1f GB/s; %.2f ms of it "
                "inside the %zu copy calls), %d threads\n", total, ms_stage, ms_since(t_1), (double)total / 1e6 / ms_since(t_1),
                ms_in_copy_calls, n_slots, ix->host_pack_threads);
    if (rc != KMM_OK || bad.load()) {
        // nothing was launched on the handle's stream; the copies issued so far only touched this stage's own buffer
        HIPCHK(hipEventRecord(ix->copied, ix->copy_stream));
        HIPCHK(hipEventSynchronize(ix->copied));
        ix->cur ^= 1; // (hand the stage back: the ordinary route takes it again)
        return rc;
    }
    // (the halo words pass 1 loads behind the last read)
    HIPCHK(hipMemsetAsync((uint8_t *)s.kmers.p + packed_total, 0, code_bytes - packed_total, ix->copy_stream));
    ix->map_calls++;
    ix->host_packed_calls++;
    const uint32_t *start_bits = nullptr;
    int64_t n_words = 0;
    if (read_offsets) { // ragged reads: where they start, as a bitset over the flat bases (what pass 1's flat front end reads)
        bool staged = false;
        const int64_t *d_offs = nullptr;
        KMMCHK(stage_in<int64_t>(ix, s.offsets, read_offsets, (size_t)(n_reads + 1), &d_offs, &staged));
        n_words = (int64_t)total / 32 + 2;
        KMMCHK(ensure(s.start_bits, (size_t)n_words * 4));
        KMMCHK(stage_copies_done(ix));
        HIPCHK(hipMemsetAsync(s.start_bits.p, 0, (size_t)n_words * 4, ix->stream));
        hipLaunchKernelGGL(k_mark_starts, dim3(grid_for(ix, (n_reads + 256) / 256, 8)), dim3(256), 0, ix->stream, d_offs, n_reads,
                           (int64_t)total, (uint32_t *)s.start_bits.p);
        hipLaunchKernelGGL(k_check_offsets, dim3(grid_for(ix, (n_reads + 255) / 256, 8)), dim3(256), 0, ix->stream, d_offs, n_reads,
                           ix->first_bad);
        HIPCHK(hipGetLastError());
        start_bits = (const uint32_t *)s.start_bits.p;
    }
    KMMCHK(rec_launch_flat(ix, (const uint32_t *)s.kmers.p, (int64_t)total, n_reads, start_bits, n_words, read_offsets ? 0 : read_len, k,
                           max_freq, also_revcomp));
    *done = true;
    return stage_release(ix, s, true);
}

// Raw FASTQ / two-line FASTA records in HOST memory (the file mapping, the inflater's output: no page-locked copy of the raw
// bytes is made), default lookup table, a chunk of radix size, "host_pack_threads" > 0: the host threads put the bases
// of the sequence lines straight into the 2-bit stream and the read starts into the bitset (kmm_hostpack::RecordsJob: the
// rules of the device-side compaction, kmm_records.hpp), the stream crosses PCIe while the rest is still being packed,
// and pass 1 runs on it as on the compaction's output.  This is what the reference's `-t` workers do with a chunk
// (bnp parser + encoder, command_line_interface.py:102-111,124-130), minus the k-mer hashing.  *done = false: a byte without
// a code or a malformed record line — the ordinary route maps the chunk and reports the byte's offset.
static int map_records_host_packed(kmm_index_t *ix, const uint8_t *raw, int64_t n_bytes, int format, int k, int max_freq,
                                   int also_revcomp, int64_t *consumed, int64_t *n_records, bool *done)
{
    *done = false;
    const size_t n = (size_t)n_bytes;
    const size_t code_bytes = (n / 4 + 1024 + 63) & ~(size_t)63, bits_bytes = (n / 8 + 256 + 63) & ~(size_t)63;
    static const bool verbose = getenv("KMM_VERBOSE") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    if (!ensure_pack_pool(ix) || !ensure_pinned(ix->pack_pinned, ix->pack_pinned_bytes, code_bytes) ||
        !ensure_pinned(ix->pack_bits_pinned, ix->pack_bits_pinned_bytes, bits_bytes))
        return KMM_OK;
    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    KMMCHK(ensure(s.kmers, code_bytes));
    const double ms_alloc = ms_since(t_0);
    const auto t_1 = std::chrono::steady_clock::now();
    kmm_hostpack::RecordsJob job;
    const size_t slice = ix->host_pack_slice_kb > 0 ? (size_t)ix->host_pack_slice_kb << 10 : kmm_hostpack::RecordsJob::slice_bytes();
    job.prepare(raw, n, format == KMM_FORMAT_FASTQ ? 4 : 2, reinterpret_cast<uint64_t *>(ix->pack_pinned),
                reinterpret_cast<uint32_t *>(ix->pack_bits_pinned), slice);
    ix->pack_pool->start([&job](int) { job.run(); });
    // groups of slices (32 MiB of raw bytes): the words of the stream that lie wholly below the group's end are final
    const size_t GROUP = std::max<size_t>(1, ((size_t)32 << 20) / slice);
    int rc = KMM_OK;
    uint64_t copied_w = 0;
    for (size_t g1 = GROUP; g1 < job.n_slices() && rc == KMM_OK; g1 += GROUP) {
        const uint64_t upto_w = job.wait_packed_prefix(g1) >> 5;
        if (upto_w > copied_w) {
            if (hipMemcpyAsync((uint8_t *)s.kmers.p + copied_w * 8, ix->pack_pinned + copied_w * 8, (size_t)(upto_w - copied_w) * 8,
                               hipMemcpyHostToDevice, ix->copy_stream) != hipSuccess)
                rc = fail(KMM_ERR_HIP, "hipMemcpyAsync of packed records: %s", hipGetErrorString(hipGetLastError()));
            copied_w = upto_w;
        }
    }
    ix->pack_pool->wait();
    const kmm_hostpack::RecordsResult r = job.finish();
    if (verbose)
        fprintf(stderr, "libkmm: host records packer: %zu bytes, buffers %.2f ms, pack %.2f ms (%.1f GB/s), %d threads, %s\n", n, ms_alloc,
                ms_since(t_1), (double)n / 1e6 / ms_since(t_1), ix->host_pack_threads, r.ok ? "ok" : "refused");
    if (rc != KMM_OK || !r.ok) {
        HIPCHK(hipEventRecord(ix->copied, ix->copy_stream));
        HIPCHK(hipEventSynchronize(ix->copied));
        ix->cur ^= 1; // (hand the stage back: the ordinary route takes it again)
        return rc;
    }
    ix->map_calls++;
    ix->host_packed_record_calls++;
    if (consumed)
        *consumed = r.consumed;
    if (n_records)
        *n_records = r.n_records;
    *done = true;
    if (r.n_bases <= 0)
        return stage_release(ix, s, copied_w != 0);
    {   // the rest of the stream: from the word that holds the last base's neighbourhood (finish() cleaned it) to the zero
        // words behind it
        const uint64_t end_w = ((uint64_t)r.n_bases >> 5) + 40, from_w = copied_w < ((uint64_t)r.n_bases >> 5) ? copied_w : ((uint64_t)r.n_bases >> 5);
        HIPCHK(hipMemcpyAsync((uint8_t *)s.kmers.p + from_w * 8, ix->pack_pinned + from_w * 8, (size_t)(end_w - from_w) * 8,
                              hipMemcpyHostToDevice, ix->copy_stream));
    }
    const bool uniform = r.uniform_len >= 16;
    const uint32_t *start_bits = nullptr;
    int64_t n_words = 0;
    if (!uniform) { // ragged reads: the read-start bitset crosses too (1 bit per base)
        n_words = r.n_bases / 32 + 2;
        KMMCHK(ensure(s.start_bits, (size_t)n_words * 4));
        HIPCHK(hipMemcpyAsync(s.start_bits.p, ix->pack_bits_pinned, (size_t)n_words * 4, hipMemcpyHostToDevice, ix->copy_stream));
        start_bits = (const uint32_t *)s.start_bits.p;
    }
    KMMCHK(rec_launch_flat(ix, (const uint32_t *)s.kmers.p, r.n_bases, r.n_records, start_bits, n_words, uniform ? r.uniform_len : 0, k,
                           max_freq, also_revcomp));
    return stage_release(ix, s, true);
}

static int map_reads_common(kmm_index_t *ix, const uint8_t *bases, const int64_t *read_offsets,
                            int64_t n_reads, int64_t read_len, int k, int max_freq,
                            int also_revcomp, const uint8_t *lut)
{
    const bool uniform = (read_offsets == nullptr);
    HIPCHK(hipSetDevice(ix->device));
    int64_t total = 0;
    const bool offs_on_device = !uniform && is_device_ptr(read_offsets);
    if (uniform) {
        total = n_reads * read_len;
    } else {
        int64_t ends[2] = {0, 0};
        if (offs_on_device) {
            HIPCHK(hipMemcpy(&ends[0], read_offsets, 8, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(&ends[1], read_offsets + n_reads, 8, hipMemcpyDeviceToHost));
        } else {
            ends[0] = read_offsets[0];
            ends[1] = read_offsets[n_reads];
        }
        if (ends[0] != 0)
            return fail(KMM_ERR_INVALID_ARG, "read_offsets[0] must be 0 (got %lld)", (long long)ends[0]);
        if (ends[1] < 0)
            return fail(KMM_ERR_INVALID_ARG, "read_offsets[n_reads] negative");
        total = ends[1];
    }
    if (total == 0)
        return KMM_OK;
    if (!bases)
        return fail(KMM_ERR_INVALID_ARG, "bases is NULL");
    if (ix->host_pack_threads > 0 && (uniform ? read_len >= 16 : !offs_on_device) && !lut && use_radix(ix, total) &&
        !is_device_ptr(bases)) {
        bool done = false;
        KMMCHK(map_reads_host_packed(ix, bases, uniform ? nullptr : read_offsets, total, n_reads, read_len, k, max_freq, also_revcomp,
                                     &done));
        if (done)
            return KMM_OK;
    }

    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    ix->map_calls++;
    bool staged = false;
    ReadsView rv;
    memset(&rv, 0, sizeof rv);
    KMMCHK(stage_in<uint8_t>(ix, s.bases, bases, (size_t)total, &rv.bases, &staged));
    KMMCHK(resolve_lut(ix, s, lut, &rv.lut, &staged));
    rv.total = total;
    rv.n_reads = n_reads;
    rv.first_bad = ix->first_bad;
    const int64_t n_tiles = (total + TILE_T - 1) / TILE_T;
    // the uniform kernel's wrap-around handles one read boundary per lane: needs read_len >= S
    const bool uniform_kernel = uniform && read_len >= 16;
    if (uniform_kernel) {
        set_uniform_geometry(ix, rv, read_len, k);
        KMMCHK(stage_copies_done(ix));
        KMMCHK(launch_map_reads<MODE_UNIFORM>(ix, rv, k, max_freq, also_revcomp ? 1 : 0));
    } else {
        if (uniform) {
            KMMCHK(ensure(s.offsets, (size_t)(n_reads + 1) * 8));
            hipLaunchKernelGGL(k_iota_offsets, dim3((unsigned)((n_reads + 1 + 255) / 256)), dim3(256),
                               0, ix->stream, (int64_t *)s.offsets.p, n_reads, read_len);
            HIPCHK(hipGetLastError());
            rv.offsets = (const int64_t *)s.offsets.p;
        } else {
            KMMCHK(stage_in<int64_t>(ix, s.offsets, read_offsets, (size_t)(n_reads + 1), &rv.offsets,
                                     &staged));
        }
        const int64_t n_words = total / 32 + 2;
        KMMCHK(ensure(s.start_bits, (size_t)n_words * 4));
        rv.start_bits = (const uint32_t *)s.start_bits.p;
        rv.n_start_words = n_words;
        KMMCHK(stage_copies_done(ix));
        HIPCHK(hipMemsetAsync(s.start_bits.p, 0, (size_t)n_words * 4, ix->stream));
        hipLaunchKernelGGL(k_mark_starts, dim3(grid_for(ix, (n_reads + 256) / 256, 8)), dim3(256), 0, ix->stream,
                           rv.offsets, n_reads, total, (uint32_t *)s.start_bits.p);
        if (!uniform)
            hipLaunchKernelGGL(k_check_offsets, dim3(grid_for(ix, (n_reads + 255) / 256, 8)), dim3(256), 0,
                               ix->stream, rv.offsets, n_reads, ix->first_bad);
        HIPCHK(hipGetLastError());
        KMMCHK(launch_map_reads<MODE_GENERAL>(ix, rv, k, max_freq, also_revcomp ? 1 : 0));
    }
    return stage_release(ix, s, staged);
}

int kmm_map_reads(kmm_index_t *ix, const uint8_t *bases, const int64_t *read_offsets, int64_t n_reads,
                  int k, int max_freq, int also_revcomp, const uint8_t *lut)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    KMMCHK(check_k(k));
    if (n_reads < 0)
        return fail(KMM_ERR_INVALID_ARG, "n_reads negative");
    if (n_reads == 0)
        return KMM_OK;
    if (!read_offsets)
        return fail(KMM_ERR_INVALID_ARG, "read_offsets is NULL");
    return guarded("kmm_map_reads", [&] { return map_reads_common(ix, bases, read_offsets, n_reads, 0, k, max_freq, also_revcomp, lut); });
}

int kmm_map_reads_uniform(kmm_index_t *ix, const uint8_t *bases, int64_t n_reads, int64_t read_len,
                          int k, int max_freq, int also_revcomp, const uint8_t *lut)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    KMMCHK(check_k(k));
    if (n_reads < 0 || read_len < 0)
        return fail(KMM_ERR_INVALID_ARG, "n_reads / read_len negative");
    if (n_reads == 0 || read_len == 0)
        return KMM_OK;
    return guarded("kmm_map_reads_uniform", [&] { return map_reads_common(ix, bases, nullptr, n_reads, read_len, k, max_freq, also_revcomp, lut); });
}

// Raw records on the radix path: should a piece of n_bytes raw bytes take it?  (Same rule as for flat reads, on the
// bases the piece holds at most: half of a FASTQ piece's bytes, all of a FASTA piece's.)
static bool records_take_radix(const kmm_index_t *ix, int64_t n_bytes, int format)
{
    return use_radix(ix, format == KMM_FORMAT_FASTQ ? n_bytes / 2 : n_bytes);
}

// Raw records on the radix path (r04): census -> the sequence bytes compacted into flat reads of 2-bit codes (one per
// byte) + the read-start bitset (kmm_records.hpp, k_rec_count2 .. k_rec_uniform), all on the copy stream, i.e. under
// the previous call's map kernels; then pass 1 runs on flat reads — on packed tiles when the reads have one length —
// instead of pushing every raw byte through the records front end (22.3 ms per 10 M reads in round 3).
//
// rec_compact_piece: one piece of at most 2^30 raw bytes (the census is a two-level scan over 1024 x 1024 tiles of 1024
// bytes), appended to the flat reads at flat position `flat_base`.  Synchronises the copy stream (the caller's host
// buffer is free afterwards) and returns where the piece's last complete record ends, its records, the flat length
// after it, and whether its reads have one length.
static int rec_compact_piece(kmm_index_t *ix, Stage &s, const uint8_t *d_raw, int64_t n_bytes, int format, const uint8_t *d_lut,
                             int64_t flat_base, uint32_t *flat, uint32_t *start_bits, int64_t *consumed, int64_t *n_records,
                             int64_t *flat_end, int64_t *uniform_len)
{
    const int64_t n_tiles = (n_bytes + REC_TB - 1) / REC_TB; // 4 KiB tiles: one wavefront, 64 bytes per lane
    const int n_super = (int)((n_tiles + 1023) / 1024);
    const size_t n_pad = (size_t)n_super * 1024;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_nl = carve(n_pad * 4), o_seq = carve(n_pad * 8), o_pre = carve(n_pad * 4), o_snl = carve((size_t)n_super * 4 + 64),
                 o_sseq = carve((size_t)n_super * 4 + 64), o_info = carve(256), o_out = carve(256);
    KMMCHK(ensure(s.aux, off));
    uint8_t *a = (uint8_t *)s.aux.p;
    uint32_t *tile_nl = (uint32_t *)(a + o_nl), *tile_pre = (uint32_t *)(a + o_pre), *super_nl = (uint32_t *)(a + o_snl),
             *super_seq = (uint32_t *)(a + o_sseq);
    unsigned long long *tile_seq = (unsigned long long *)(a + o_seq), *d_out = (unsigned long long *)(a + o_out);
    int64_t *d_info = (int64_t *)(a + o_info);
    // The compaction kernels run on the handle's OWN stream, behind the previous call's passes — not on the copy stream
    // beside them: beside them they gain nothing (both fill the CUs: 64.7 against 65.2 ms per two 10 M-read calls).
    // (Running them there is how round 4 found a race in pass 1 that round 3 had left behind — foreign wavefronts on
    // pass 1's SIMDs delayed a counter clear past another wavefront's next ranking atomic, rx_sort_emit; fixed there,
    // profiles/r04/records_overlap_fault.txt — and tools/records_overlap_bisect.py still can: debug_records_copy_stream.)
    // Host -> HBM copies stay on the copy stream; every kernel of a handle runs on the handle's stream.
    hipStream_t cs = ix->dbg_rec_copy_stream ? ix->copy_stream : ix->stream;
    const int skip = ix->dbg_rec_skip;
    const uint32_t pm = (uint32_t)format - 1u, hc = format == KMM_FORMAT_FASTQ ? (uint32_t)'@' : (uint32_t)'>';
    const dim3 g4((unsigned)grid_for(ix, (n_tiles + 3) / 4, 8)); // persistent wavefronts: tiles t, t + waves, ...
    HIPCHK(hipMemsetAsync(tile_nl, 0, n_pad * 4, cs));
    HIPCHK(hipMemsetAsync(tile_seq, 0, n_pad * 8, cs));
    HIPCHK(hipMemsetAsync(d_info, 0, 512, cs)); // (info and out_info: neighbours)
    if (!(skip & 1))
        hipLaunchKernelGGL(k_rec_count2, g4, dim3(256), 0, cs, d_raw, n_bytes, n_tiles, tile_nl, tile_seq);
    if (!(skip & 2)) {
        hipLaunchKernelGGL(k_rec_scan1, dim3(n_super), dim3(1024), 0, cs, tile_nl, super_nl);
        hipLaunchKernelGGL(k_rec_scan2, dim3(1), dim3(1024), 0, cs, d_raw, n_bytes, n_super, tile_nl, super_nl, (uint32_t)format, d_info,
                           (int)REC_TB);
        hipLaunchKernelGGL(k_rec_seq_scan, dim3(n_super), dim3(1024), 0, cs, tile_seq, tile_nl, super_nl, n_tiles, pm, tile_pre, super_seq);
        hipLaunchKernelGGL(k_super_scan, dim3(1), dim3(1024), 0, cs, super_seq, n_super, (uint32_t *)(d_out + 8));
    }
    if (!(skip & 4))
        hipLaunchKernelGGL(k_rec_scatter, g4, dim3(256), 0, cs, d_raw, n_bytes, n_tiles, tile_nl, super_nl, tile_pre, super_seq, d_info,
                           d_lut, pm, hc, flat, (uint64_t)flat_base, start_bits, ix->first_bad, d_out);
    if (!(skip & 8))
        hipLaunchKernelGGL(k_rec_uniform, dim3(grid_for(ix, (n_bytes / 32 + 256) / 256, 4)), dim3(256), 0, cs, start_bits,
                           (uint64_t)flat_base, d_info, d_out);
    HIPCHK(hipGetLastError());
    struct { int64_t info[8]; unsigned long long out[8]; } h;
    memset(&h, 0, sizeof h);
    HIPCHK(hipMemcpyAsync(h.info, d_info, 64, hipMemcpyDeviceToHost, cs));
    HIPCHK(hipMemcpyAsync(h.out, d_out, 64, hipMemcpyDeviceToHost, cs));
    HIPCHK(hipStreamSynchronize(cs)); // (it waited for the copy: the borrowed host buffer is free from here on)
    *consumed = h.info[0];
    *n_records = h.info[1];
    *flat_end = h.info[0] > 0 ? (int64_t)h.out[0] : flat_base;
    const int64_t piece = *flat_end - flat_base, recs = h.info[1];
    const int64_t L = recs > 0 ? piece / recs : 0;
    *uniform_len = (recs > 0 && L * recs == piece && h.out[1] == (unsigned long long)recs && h.out[2] == 0) ? L : 0;
    return KMM_OK;
}

// The flat reads (codes, one per byte) through the radix path.
static int rec_launch_flat(kmm_index_t *ix, const uint32_t *flat, int64_t total, int64_t n_reads, const uint32_t *start_bits,
                           int64_t n_words, int64_t uniform_len, int k, int max_freq, int also_revcomp)
{
    ReadsView rv;
    memset(&rv, 0, sizeof rv);
    rv.bases = reinterpret_cast<const uint8_t *>(flat);
    rv.codes2 = 1;           // 16 two-bit codes per word: pass 1 stages the words as they are
    rv.total = total;
    rv.n_reads = n_reads;
    rv.lut = ix->lut_codes;
    rv.first_bad = ix->first_bad;
    KMMCHK(stage_copies_done(ix));
    if (uniform_len >= 16 && uniform_len * n_reads == total) {
        set_uniform_geometry(ix, rv, uniform_len, k);
        if (rv.pk_rpt)
            return launch_rx<MODE_PACKED>(ix, rv, nullptr, 0, k, max_freq, also_revcomp ? 1 : 0);
        return launch_rx<MODE_UNIFORM>(ix, rv, nullptr, 0, k, max_freq, also_revcomp ? 1 : 0);
    }
    rv.start_bits = start_bits;
    rv.n_start_words = n_words;
    return launch_rx<MODE_GENERAL>(ix, rv, nullptr, 0, k, max_freq, also_revcomp ? 1 : 0);
}

// One piece, compacted and mapped by itself (the unwrapped pieces of multi-line FASTA come this way).
static int map_records_piece_radix(kmm_index_t *ix, Stage &s, const uint8_t *d_raw, int64_t n_bytes, int format, int k,
                                   int max_freq, int also_revcomp, const uint8_t *d_lut, int64_t *consumed, int64_t *n_records)
{
    const size_t n_words = (size_t)n_bytes / 32 + 2, code_bytes = ((size_t)n_bytes / 4 + 256) & ~(size_t)15;
    KMMCHK(ensure(s.start_bits, n_words * 4));
    KMMCHK(ensure(s.kmers, code_bytes));
    KMMCHK(stage_copies_done(ix)); // (a host buffer staged by the caller: the kernels below wait for the copy)
    HIPCHK(hipMemsetAsync(s.start_bits.p, 0, n_words * 4, ix->stream));
    HIPCHK(hipMemsetAsync(s.kmers.p, 0, code_bytes, ix->stream));
    int64_t flat_end = 0, L = 0;
    KMMCHK(rec_compact_piece(ix, s, d_raw, n_bytes, format, d_lut, 0, (uint32_t *)s.kmers.p, (uint32_t *)s.start_bits.p, consumed,
                             n_records, &flat_end, &L));
    if (*consumed <= 0 || flat_end <= 0)
        return KMM_OK;
    return rec_launch_flat(ix, (const uint32_t *)s.kmers.p, flat_end, *n_records, (const uint32_t *)s.start_bits.p, (int64_t)n_words,
                           L, k, max_freq, also_revcomp);
}

// A whole kmm_map_records call on the radix path: the pieces (at most 2^30 raw bytes each, every one starting where the
// previous one's last complete record ended) are compacted one after the other into ONE array of flat reads, which then
// takes the radix path as one batch — the passes' fixed costs per batch (0.85 ms at the 100 M index) are paid once per
// call, not once per GiB of FASTQ.
static int map_records_radix_call(kmm_index_t *ix, const uint8_t *raw, int64_t n_bytes, int format, int k, int max_freq,
                                  int also_revcomp, const uint8_t *lut, int64_t *consumed, int64_t *n_records)
{
    static const bool verbose = getenv("KMM_VERBOSE") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    const double ms_acquire = ms_since(t_0);
    ix->map_calls++;
    bool staged = false;
    const uint8_t *d_lut = nullptr;
    KMMCHK(resolve_lut(ix, s, lut, &d_lut, &staged));
    if (staged)
        KMMCHK(stage_copies_done(ix)); // (a caller's lookup table staged from the host)
    const bool on_device = is_device_ptr(raw);
    const size_t n_words = (size_t)n_bytes / 32 + 2, code_bytes = ((size_t)n_bytes / 4 + 256) & ~(size_t)15;
    KMMCHK(ensure(s.start_bits, n_words * 4));
    KMMCHK(ensure(s.kmers, code_bytes));
    if (!(ix->dbg_rec_skip & 16)) {
        hipStream_t ms = ix->dbg_rec_copy_stream ? ix->copy_stream : ix->stream;
        HIPCHK(hipMemsetAsync(s.start_bits.p, 0, n_words * 4, ms));
        HIPCHK(hipMemsetAsync(s.kmers.p, 0, code_bytes, ms));
    }
    const int64_t piece_max = (int64_t)1 << 30;
    int64_t off = 0, recs = 0, flat = 0, L = -1;
    while (off < n_bytes) {
        const int64_t len = n_bytes - off < piece_max ? n_bytes - off : piece_max;
        const uint8_t *d_raw = raw + off;
        if (!on_device) {
            KMMCHK(ensure(s.bases, (size_t)len));
            HIPCHK(hipMemcpyAsync(s.bases.p, raw + off, (size_t)len, hipMemcpyHostToDevice, ix->copy_stream));
            KMMCHK(stage_copies_done(ix)); // the compaction kernels (handle's stream) wait for the copy
            d_raw = (const uint8_t *)s.bases.p;
        }
        int64_t used = 0, nr = 0, flat_end = flat, Lp = 0;
        KMMCHK(rec_compact_piece(ix, s, d_raw, len, format, d_lut, flat, (uint32_t *)s.kmers.p, (uint32_t *)s.start_bits.p, &used, &nr,
                                 &flat_end, &Lp));
        if (used > 0) {
            L = (L == -1 || L == Lp) ? Lp : 0; // one length over all pieces, or none
            flat = flat_end;
        }
        off += used;
        recs += nr;
        if (used == 0 || len < piece_max)
            break; // no complete record left in reach / the last piece
    }
    if (consumed)
        *consumed = off;
    if (n_records)
        *n_records = recs;
    const double ms_compact = ms_since(t_0) - ms_acquire;
    if (flat > 0 && !ix->dbg_rec_skip && !ix->dbg_rec_copy_stream)
        KMMCHK(rec_launch_flat(ix, (const uint32_t *)s.kmers.p, flat, recs, (const uint32_t *)s.start_bits.p, (int64_t)n_words,
                               L > 0 ? L : 0, k, max_freq, also_revcomp));
    if (verbose)
        fprintf(stderr, "libkmm: records on the radix path: %lld bytes %s: waited %.2f ms for the stage, census + compaction issued %.2f ms, "
                "passes issued %.2f ms\n", (long long)n_bytes, on_device ? "in HBM" : "in host memory", ms_acquire, ms_compact,
                ms_since(t_0) - ms_acquire - ms_compact);
    return stage_release(ix, s, false);
}

// One piece of at most 2^30 bytes (the newline census is a two-level scan over 1024 x 1024 tiles of 1024 bytes).
static int map_records_piece(kmm_index_t *ix, const uint8_t *raw, int64_t n_bytes, int format, int k,
                             int max_freq, int also_revcomp, const uint8_t *lut, int64_t *consumed,
                             int64_t *n_records)
{
    static_assert(TILE_T == 1024, "records mode counts newlines per 1024-byte tile");
    *consumed = 0;
    *n_records = 0;
    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    ix->map_calls++;
    bool staged = false;
    ReadsView rv;
    memset(&rv, 0, sizeof rv);
    KMMCHK(stage_in<uint8_t>(ix, s.bases, raw, (size_t)n_bytes, &rv.bases, &staged));
    KMMCHK(resolve_lut(ix, s, lut, &rv.lut, &staged));
    if (records_take_radix(ix, n_bytes, format)) {
        KMMCHK(map_records_piece_radix(ix, s, rv.bases, n_bytes, format, k, max_freq, also_revcomp, rv.lut, consumed, n_records));
        return stage_release(ix, s, false);
    }
    const int64_t n_tiles = (n_bytes + TILE_T - 1) / TILE_T;
    const int n_super = (int)((n_tiles + 1023) / 1024);
    KMMCHK(ensure(s.tile_first, (size_t)n_super * 1024 * 4));
    KMMCHK(ensure(s.offsets, (size_t)n_super * 4 + 64));
    uint32_t *tile_cnt = (uint32_t *)s.tile_first.p;
    uint32_t *super_tot = (uint32_t *)s.offsets.p;
    int64_t *d_out = (int64_t *)((uint8_t *)s.offsets.p + (((size_t)n_super * 4 + 15) & ~(size_t)15));
    // (kernels run on the handle's stream only — the copy stream carries copies: see rec_compact_piece)
    KMMCHK(stage_copies_done(ix));
    HIPCHK(hipMemsetAsync(tile_cnt, 0, (size_t)n_super * 1024 * 4, ix->stream));
    hipLaunchKernelGGL(k_rec_count, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, ix->stream,
                       rv.bases, n_bytes, n_tiles, tile_cnt);
    hipLaunchKernelGGL(k_rec_scan1, dim3(n_super), dim3(1024), 0, ix->stream, tile_cnt, super_tot);
    hipLaunchKernelGGL(k_rec_scan2, dim3(1), dim3(1024), 0, ix->stream, rv.bases, n_bytes, n_super,
                       tile_cnt, super_tot, (uint32_t)format, d_out, 1024);
    HIPCHK(hipGetLastError());
    int64_t out[3] = {0, 0, 0};
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof out, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream)); // (it waited for the copy: the borrowed host buffer is free from here on)
    *consumed = out[0];
    *n_records = out[1];
    if (out[0] > 0) {
        rv.total = out[0];
        rv.first_bad = ix->first_bad;
        rv.tile_nl = tile_cnt;
        rv.super_nl = super_tot;
        rv.period_mask = (uint32_t)format - 1u;
        rv.header_char = format == KMM_FORMAT_FASTQ ? (uint32_t)'@' : (uint32_t)'>';
        KMMCHK(stage_copies_done(ix));
        KMMCHK(launch_map_reads<MODE_RECORDS>(ix, rv, k, max_freq, also_revcomp ? 1 : 0));
    }
    return stage_release(ix, s, false);
}

// One piece of a multi-line FASTA chunk: unwrapped on the device into two-line FASTA (kmm_records.hpp k_ml_*), then
// mapped like one.  Only whole records are taken: up to the start of the chunk's last header line, or all of it when
// the caller says the chunk ends the file.
static int map_multiline_piece(kmm_index_t *ix, const uint8_t *raw, int64_t n_bytes, bool last, int k, int max_freq,
                               int also_revcomp, const uint8_t *lut, int64_t *consumed, int64_t *n_records)
{
    *consumed = 0;
    *n_records = 0;
    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    bool staged = false;
    const uint8_t *d_raw = nullptr;
    KMMCHK(stage_in<uint8_t>(ix, s.bases, raw, (size_t)n_bytes, &d_raw, &staged));
    const int64_t n_tiles = (n_bytes + 1023) / 1024;
    const int n_super = (int)((n_tiles + 1023) / 1024);
    KMMCHK(ensure(s.tile_first, (size_t)n_super * 1024 * 4));
    KMMCHK(ensure(s.offsets, (size_t)n_super * 4 + 64));
    KMMCHK(ensure(s.start_bits, (size_t)n_tiles * 8 + 64));
    KMMCHK(ensure(s.kmers, (size_t)n_bytes + 16));
    uint32_t *tile_cnt = (uint32_t *)s.tile_first.p;
    uint32_t *super_tot = (uint32_t *)s.offsets.p;
    uint8_t *cells = (uint8_t *)s.offsets.p + (((size_t)n_super * 4 + 15) & ~(size_t)15);
    uint32_t *d_total = (uint32_t *)cells;                       // kept bytes of the whole chunk
    int *d_last_header = (int *)(cells + 8);                     // start of the last header line (-1: none)
    unsigned long long *d_out_len = (unsigned long long *)(cells + 16);
    int32_t *tile_last = (int32_t *)s.start_bits.p, *tile_prev = tile_last + n_tiles;
    uint8_t *unwrapped = (uint8_t *)s.kmers.p;
    KMMCHK(stage_copies_done(ix));
    hipStream_t cs = ix->stream; // (kernels run on the handle's stream only: see rec_compact_piece)
    HIPCHK(hipMemsetAsync(tile_cnt, 0, (size_t)n_super * 1024 * 4, cs));
    HIPCHK(hipMemsetAsync(d_last_header, 0xFF, 4, cs));
    const dim3 g4((unsigned)((n_tiles + 3) / 4));
    hipLaunchKernelGGL(k_ml_tile_last, g4, dim3(256), 0, cs, d_raw, n_bytes, n_tiles, tile_last);
    hipLaunchKernelGGL(k_ml_scan, dim3(1), dim3(1024), 0, cs, tile_last, n_tiles, tile_prev);
    hipLaunchKernelGGL(k_ml_flags, g4, dim3(256), 0, cs, d_raw, n_bytes, n_tiles, tile_prev, tile_cnt, d_last_header);
    hipLaunchKernelGGL(k_rec_scan1, dim3(n_super), dim3(1024), 0, cs, tile_cnt, super_tot);
    hipLaunchKernelGGL(k_super_scan, dim3(1), dim3(1024), 0, cs, super_tot, n_super, d_total);
    HIPCHK(hipGetLastError());
    struct { uint32_t total; uint32_t pad; int last_header; } h = {0, 0, -1};
    HIPCHK(hipMemcpyAsync(&h, cells, 12, hipMemcpyDeviceToHost, cs));
    HIPCHK(hipStreamSynchronize(cs)); // (the borrowed host buffer is free from here on)
    const int64_t limit = last ? n_bytes : (h.last_header > 0 ? (int64_t)h.last_header : 0);
    int64_t out_len = 0;
    if (limit > 0) {
        unsigned long long ol = h.total;
        if (limit < n_bytes) {
            HIPCHK(hipMemsetAsync(d_out_len, 0, 8, cs));
        }
        hipLaunchKernelGGL(k_ml_scatter, g4, dim3(256), 0, cs, d_raw, n_bytes, limit, n_tiles, tile_prev, tile_cnt, super_tot,
                           unwrapped, d_out_len);
        HIPCHK(hipGetLastError());
        if (limit < n_bytes) {
            HIPCHK(hipMemcpyAsync(&ol, d_out_len, 8, hipMemcpyDeviceToHost, cs));
            HIPCHK(hipStreamSynchronize(cs));
        }
        out_len = (int64_t)ol;
    }
    int rc = KMM_OK;
    if (out_len > 0) {
        // the unwrapped records, from HBM, through the two-line parser (its kernels wait for the copy stream)
        int64_t used = 0;
        rc = map_records_piece(ix, unwrapped, out_len, KMM_FORMAT_FASTA2, k, max_freq, also_revcomp, lut, &used, n_records);
        if (rc == KMM_OK && used != out_len)
            rc = fail(KMM_ERR_MALFORMED, "multi-line FASTA chunk: %lld of %lld unwrapped bytes form whole records (a record "
                      "without a sequence line, or bytes before the first '>')", (long long)used, (long long)out_len);
        if (rc == KMM_OK)
            *consumed = limit;
    }
    const int rel = stage_release(ix, s, false); // (after the inner call's kernels: they read this stage's buffer)
    return rc != KMM_OK ? rc : rel;
}

static int map_records_entry(kmm_index_t *ix, const uint8_t *raw, int64_t n_bytes, int format, int k, int max_freq, int also_revcomp,
                             const uint8_t *lut, int64_t *consumed, int64_t *n_records);

int kmm_map_records(kmm_index_t *ix, const uint8_t *raw, int64_t n_bytes, int format, int k,
                    int max_freq, int also_revcomp, const uint8_t *lut, int64_t *consumed,
                    int64_t *n_records)
{
    return guarded("kmm_map_records", [&] { return map_records_entry(ix, raw, n_bytes, format, k, max_freq, also_revcomp, lut, consumed, n_records); });
}

static int map_records_entry(kmm_index_t *ix, const uint8_t *raw, int64_t n_bytes, int format, int k, int max_freq, int also_revcomp,
                             const uint8_t *lut, int64_t *consumed, int64_t *n_records)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    KMMCHK(check_k(k));
    const bool last_chunk = (format & KMM_FORMAT_LAST_CHUNK) != 0;
    format &= ~KMM_FORMAT_LAST_CHUNK;
    if (format != KMM_FORMAT_FASTQ && format != KMM_FORMAT_FASTA2 && format != KMM_FORMAT_FASTA)
        return fail(KMM_ERR_INVALID_ARG, "format must be KMM_FORMAT_FASTQ (4), KMM_FORMAT_FASTA2 (2) or KMM_FORMAT_FASTA (1)");
    if (n_bytes < 0)
        return fail(KMM_ERR_INVALID_ARG, "n_bytes negative");
    if (consumed)
        *consumed = 0;
    if (n_records)
        *n_records = 0;
    if (n_bytes == 0)
        return KMM_OK;
    if (!raw)
        return fail(KMM_ERR_INVALID_ARG, "raw is NULL");
    HIPCHK(hipSetDevice(ix->device));
    // chunks beyond 2^30 bytes are mapped piece by piece: every piece starts where the previous one's last complete
    // record ended, so the pieces cut the chunk exactly as one census over all of it would
    if (format != KMM_FORMAT_FASTA && records_take_radix(ix, n_bytes, format)) {
        // raw bytes in host memory: the host threads pack the sequence lines to 2 bits per base before they cross PCIe
        if (ix->host_pack_threads > 0 && !lut && !ix->dbg_rec_skip && !ix->dbg_rec_copy_stream && !is_device_ptr(raw)) {
            bool done = false;
            KMMCHK(map_records_host_packed(ix, raw, n_bytes, format, k, max_freq, also_revcomp, consumed, n_records, &done));
            if (done)
                return KMM_OK;
        }
        return map_records_radix_call(ix, raw, n_bytes, format, k, max_freq, also_revcomp, lut, consumed, n_records);
    }
    const int64_t piece_max = (int64_t)1 << 30;
    int64_t off = 0, recs = 0;
    while (off < n_bytes) {
        const int64_t len = n_bytes - off < piece_max ? n_bytes - off : piece_max;
        int64_t used = 0, nr = 0;
        if (format == KMM_FORMAT_FASTA)
            KMMCHK(map_multiline_piece(ix, raw + off, len, last_chunk && off + len == n_bytes, k, max_freq, also_revcomp, lut,
                                       &used, &nr));
        else
            KMMCHK(map_records_piece(ix, raw + off, len, format, k, max_freq, also_revcomp, lut, &used, &nr));
        off += used;
        recs += nr;
        if (used == 0 || len < piece_max)
            break; // no complete record left in reach / the last piece
    }
    if (consumed)
        *consumed = off;
    if (n_records)
        *n_records = recs;
    return KMM_OK;
}

int kmm_map_packed(kmm_index_t *ix, const uint32_t *codes, int64_t n_bases, int64_t n_reads, int64_t read_len,
                   const uint32_t *read_starts, int k, int max_freq, int also_revcomp)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    KMMCHK(check_k(k));
    if (n_bases < 0 || n_reads < 0 || read_len < 0)
        return fail(KMM_ERR_INVALID_ARG, "n_bases / n_reads / read_len negative");
    if (n_bases == 0)
        return KMM_OK;
    if (!codes)
        return fail(KMM_ERR_INVALID_ARG, "codes is NULL");
    if (read_len > 0 ? n_reads * read_len != n_bases : !read_starts)
        return fail(KMM_ERR_INVALID_ARG, read_len > 0 ? "n_reads * read_len != n_bases" : "read_starts is NULL and read_len is 0");
    if (!ix->rx_ok)
        return fail(KMM_ERR_INVALID_ARG, "packed reads are mapped by the radix path, which is not available for this index "
                    "(kmm_get_param \"radix_unavailable_reason\")");
    HIPCHK(hipSetDevice(ix->device));
    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    ix->map_calls++;
    bool staged = false;
    const uint32_t *d_codes = nullptr, *d_starts = nullptr;
    KMMCHK(stage_in<uint32_t>(ix, s.kmers, codes, (size_t)((n_bases + 15) / 16), &d_codes, &staged));
    int64_t n_words = 0;
    if (read_len == 0) {
        n_words = n_bases / 32 + 1;
        KMMCHK(stage_in<uint32_t>(ix, s.start_bits, read_starts, (size_t)n_words, &d_starts, &staged));
    } else if (read_len < 16) { // (the uniform front ends need reads of at least 16 bases: shorter ones go as ragged reads)
        n_words = n_bases / 32 + 2;
        KMMCHK(ensure(s.offsets, (size_t)(n_reads + 1) * 8));
        KMMCHK(ensure(s.start_bits, (size_t)n_words * 4));
        HIPCHK(hipMemsetAsync(s.start_bits.p, 0, (size_t)n_words * 4, ix->stream));
        hipLaunchKernelGGL(k_iota_offsets, dim3((unsigned)((n_reads + 1 + 255) / 256)), dim3(256), 0, ix->stream, (int64_t *)s.offsets.p,
                           n_reads, read_len);
        hipLaunchKernelGGL(k_mark_starts, dim3(grid_for(ix, (n_reads + 256) / 256, 8)), dim3(256), 0, ix->stream,
                           (const int64_t *)s.offsets.p, n_reads, n_bases, (uint32_t *)s.start_bits.p);
        HIPCHK(hipGetLastError());
        d_starts = (const uint32_t *)s.start_bits.p;
    }
    KMMCHK(rec_launch_flat(ix, d_codes, n_bases, n_reads, d_starts, n_words, read_len >= 16 ? read_len : 0, k, max_freq, also_revcomp));
    return stage_release(ix, s, staged);
}


// Stages comp[pre, n_comp) to d_comp + pre through the page-locked ring (comp[0, pre) is there already: a member's head left
// over from the chunk before) and walks the member chain of comp[0, n_comp) behind the copying threads.  cap: inflated bytes
// the chain may hold.  all: n_comp is where the chunk ends (an incomplete last member ends the chain).
static int bgzf_stage_and_scan(kmm_index_t *ix, const uint8_t *comp, int64_t pre, int64_t n_comp, uint8_t *d_comp, unsigned long long out_cap,
                               BgzfStaged &st)
{
    auto ms_since = [](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    std::vector<unsigned long long> &m_off = st.m_off, &o_off = st.o_rel;
    m_off.assign(1, 0ull);
    o_off.assign(1, 0ull);
    const int64_t carry = 0; // (offsets relative to the chunk: the caller adds what it carries over)
    uint64_t &p = st.p;
    p = 0;
    bool chain_end = false, last_chunk = false;
    int &chain_err = st.chain_err;
    uint32_t &bad_isize = st.bad_isize, &bad_ms = st.bad_ms;
    chain_err = 0;
    // walks the chain through comp[0, limit); all = the limit is the end of the chunk
    auto scan_upto = [&](uint64_t limit, bool all) {
        while (!chain_end && p + 18 <= limit) {
            const uint32_t ms = kmm_gz::bgzf_member_size(comp + p, limit - p);
            if (!ms) {
                // (a header that needs more bytes than are in reach: wait for them, or — at the end of the chunk — an
                // incomplete member that the caller brings again)
                const uint32_t xlen = (uint32_t)comp[p + 10] | ((uint32_t)comp[p + 11] << 8);
                if (comp[p] == 0x1f && comp[p + 1] == 0x8b && comp[p + 2] == 8 && (comp[p + 3] & 4) && p + 12 + xlen + 8 > limit) {
                    chain_end = all;
                    return;
                }
                chain_err = 1;
                chain_end = true;
                return;
            }
            if (p + ms > limit) {
                chain_end = all; // (an incomplete member at the end of the chunk)
                return;
            }
            const uint32_t isize = kmm_gz::rd32(comp + p + ms - 4);
            if ((uint64_t)isize > (uint64_t)ms * 1032ull + 64ull) {
                chain_err = 2;
                bad_isize = isize;
                bad_ms = ms;
                chain_end = true;
                return;
            }
            if (o_off.back() - (unsigned long long)carry + isize > out_cap && m_off.size() > 1) {
                st.hit_cap = true; // (the call stops at its own size limit: the caller continues with the same flags)
                chain_end = true;
                return;
            }
            p += ms;
            m_off.push_back(p);
            o_off.push_back(o_off.back() + isize);
        }
        if (all)
            chain_end = true;
    };
    (void)last_chunk;
    bool &staged = st.staged;
    double &ms_scan_inside = st.ms_scan_inside;
    staged = false;
    ms_scan_inside = 0;
    // ("debug_bgzf_ring_slot_kb": tests wrap the ring many times with a small input)
    const size_t SLOT = ix->dbg_bgzf_slot_kb > 0 && ((size_t)ix->dbg_bgzf_slot_kb << 10) < RING_SLOT ? (size_t)ix->dbg_bgzf_slot_kb << 10 : RING_SLOT;
    const size_t SUB = SLOT < ((size_t)1 << 20) ? SLOT : (size_t)1 << 20;
    constexpr int SLOTS = RING_SLOTS;
    if (n_comp > pre && ensure_pack_pool(ix) && ensure_ring(ix)) {
        const size_t n_stage = (size_t)(n_comp - pre); // bytes to copy: [pre, n_comp)
        const size_t n_slots = (n_stage + SLOT - 1) / SLOT, n_sub = (n_stage + SUB - 1) / SUB;
        std::vector<std::atomic<uint32_t>> filled(n_slots); // 1 MiB pieces copied, per slot-sized piece
        for (auto &f : filled)
            f.store(0, std::memory_order_relaxed);
        std::atomic<size_t> next{0}, slots_free{(size_t)SLOTS}; // slot-sized pieces [0, slots_free) may be written
        std::atomic<bool> stop{false};
        uint8_t *const *ring = ix->ring;
        ix->pack_pool->start([&](int) {
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= n_sub)
                    return;
                const size_t piece = c / (SLOT / SUB);
                int spins = 0;
                while (piece >= slots_free.load(std::memory_order_acquire) && !stop.load(std::memory_order_relaxed)) {
                    if (++spins < 2000) {
#if defined(__x86_64__)
                        __builtin_ia32_pause();
#endif
                    } else
                        std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
                if (stop.load(std::memory_order_relaxed))
                    return;
                const size_t b0 = c * SUB, len = n_stage - b0 < SUB ? n_stage - b0 : SUB;
                memcpy(ring[piece % SLOTS] + (b0 - piece * SLOT), comp + pre + b0, len);
                filled[piece].fetch_add(1, std::memory_order_release);
            }
        });
        int rc = KMM_OK;
        size_t landed = 0; // slot-sized pieces whose copy to HBM is known to have finished
        for (size_t c = 0; c < n_slots && rc == KMM_OK; ++c) {
            const size_t b0 = c * SLOT, len = n_stage - b0 < SLOT ? n_stage - b0 : SLOT;
            const uint32_t want = (uint32_t)((len + SUB - 1) / SUB);
            while (filled[c].load(std::memory_order_acquire) < want) {
                // meanwhile: slots whose copies have landed are handed back, the chain is walked through what is there
                bool did = false;
                while (landed + SLOTS < n_slots && landed < c) {
                    if (hipEventQuery(ix->bgzf_slot_ev[landed % SLOTS]) != hipSuccess) {
                        (void)hipGetLastError(); // ("not ready" is no error to keep)
                        break;
                    }
                    ++landed;
                    slots_free.store(landed + SLOTS, std::memory_order_release);
                    did = true;
                }
                if (!chain_end && !chain_err && c > 0) {
                    const auto t_s = std::chrono::steady_clock::now();
                    const uint64_t before = p;
                    scan_upto((uint64_t)pre + (uint64_t)b0, false);
                    ms_scan_inside += ms_since(t_s);
                    did = did || p != before;
                }
                if (!did)
                    std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            if (hipMemcpyAsync(d_comp + pre + b0, ring[c % SLOTS], len, hipMemcpyHostToDevice, ix->copy_stream) != hipSuccess ||
                hipEventRecord(ix->bgzf_slot_ev[c % SLOTS], ix->copy_stream) != hipSuccess)
                rc = fail(KMM_ERR_HIP, "copy of compressed bytes: %s", hipGetErrorString(hipGetLastError()));
            // a slot is written again only when its copy has landed: the oldest one is waited for when the ring is full
            while (rc == KMM_OK && landed + SLOTS < n_slots && landed + SLOTS <= c + 1) {
                if (hipEventSynchronize(ix->bgzf_slot_ev[landed % SLOTS]) != hipSuccess) {
                    rc = fail(KMM_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(hipGetLastError()));
                    break;
                }
                ++landed;
                slots_free.store(landed + SLOTS, std::memory_order_release);
            }
        }
        if (rc != KMM_OK)
            stop.store(true);
        ix->pack_pool->wait();
        if (rc != KMM_OK) {
            (void)hipStreamSynchronize(ix->copy_stream);
            return rc;
        }
        staged = true;
    }
    scan_upto((uint64_t)n_comp, true);
    return KMM_OK;
}

int kmm_map_bgzf_hint_next(kmm_index_t *ix, const uint8_t *comp_next, int64_t n_next)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    ix->bgzf_hint_ptr = n_next > 0 ? comp_next : nullptr;
    ix->bgzf_hint_n = n_next > 0 ? n_next : 0;
    return KMM_OK;
}

static int map_bgzf_entry(kmm_index_t *ix, const uint8_t *comp, int64_t n_comp, int format, int k, int max_freq, int also_revcomp,
                          const uint8_t *lut, int64_t *consumed_comp, int64_t *n_records);

int kmm_map_bgzf(kmm_index_t *ix, const uint8_t *comp, int64_t n_comp, int format, int k, int max_freq, int also_revcomp,
                 const uint8_t *lut, int64_t *consumed_comp, int64_t *n_records)
{
    return guarded("kmm_map_bgzf", [&] { return map_bgzf_entry(ix, comp, n_comp, format, k, max_freq, also_revcomp, lut, consumed_comp, n_records); });
}

static int map_bgzf_entry(kmm_index_t *ix, const uint8_t *comp, int64_t n_comp, int format, int k, int max_freq, int also_revcomp,
                          const uint8_t *lut, int64_t *consumed_comp, int64_t *n_records)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    KMMCHK(check_k(k));
    bool last_chunk = (format & KMM_FORMAT_LAST_CHUNK) != 0;
    const bool new_stream = (format & KMM_FORMAT_NEW_STREAM) != 0;
    const int fmt = format & ~(KMM_FORMAT_LAST_CHUNK | KMM_FORMAT_NEW_STREAM);
    if (fmt != KMM_FORMAT_FASTQ && fmt != KMM_FORMAT_FASTA2)
        return fail(KMM_ERR_INVALID_ARG, "kmm_map_bgzf: format must be KMM_FORMAT_FASTQ (4) or KMM_FORMAT_FASTA2 (2)");
    if (n_comp < 0 || (n_comp > 0 && !comp))
        return fail(KMM_ERR_INVALID_ARG, "comp NULL or n_comp negative");
    if (consumed_comp)
        *consumed_comp = 0;
    if (n_records)
        *n_records = 0;
    if (n_comp > 0 && is_device_ptr(comp))
        return fail(KMM_ERR_INVALID_ARG, "kmm_map_bgzf takes the compressed bytes from host memory (the member chain is read there)");
    HIPCHK(hipSetDevice(ix->device));
    if (new_stream)
        ix->bgzf_carry_len = 0;
    static const bool verbose = getenv("KMM_VERBOSE") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    // The compressed bytes go through a RING of page-locked memory (8 slots of 16 MiB; from a file mapping — pageable memory —
    // the runtime's own staging is slow, and a page-locked buffer the size of the window costs ~50 ms per GB to make, more
    // than the whole call): the packing threads copy 1 MiB pieces into the slots, a slot leaves for HBM as soon as it is full
    // and is refilled when its copy has landed.  The member chain is read from the caller's bytes BEHIND the threads — what
    // they have copied is mapped into the process, so the walk (two cache lines per member) pays no page fault — and at the
    // same time: the calling thread has nothing else to do while the threads copy.
    const int cur = ix->bgzf_cur;
    ix->bgzf_cur ^= 1;
    if (!ix->bgzf_done[cur])
        HIPCHK(hipEventCreateWithFlags(&ix->bgzf_done[cur], hipEventDisableTiming));
    const int64_t carry = ix->bgzf_carry_len;
    constexpr unsigned long long CALL_CAP = 7ull << 29, PRE_CARRY = 256ull << 20; // 3.5 GiB per call; what a prestaged chain leaves for a carry
    BgzfStaged st;
    bool from_pre = false;
    if (ix->bgzf_pre_valid && ix->bgzf_pre_from == comp && ix->bgzf_pre_n == n_comp && ix->bgzf_pre_buf == cur &&
        (unsigned long long)carry <= PRE_CARRY) {
        st = std::move(ix->bgzf_pre); // staged and walked while the chunk before this one was being inflated
        st.ms_scan_inside = 0;
        from_pre = true;
    }
    ix->bgzf_pre_valid = false;
    if (!from_pre) {
        if (ix->bgzf_used[cur])
            HIPCHK(hipStreamWaitEvent(ix->copy_stream, ix->bgzf_done[cur], 0)); // the kernels that last read these buffers are done
        KMMCHK(ensure(ix->bgzf_comp[cur], (size_t)n_comp + 64));
        KMMCHK(bgzf_stage_and_scan(ix, comp, 0, n_comp, (uint8_t *)ix->bgzf_comp[cur].p, CALL_CAP - (unsigned long long)carry, st));
    }
    uint8_t *d_comp = (uint8_t *)ix->bgzf_comp[cur].p;
    const bool staged = st.staged;
    if (st.hit_cap)
        last_chunk = false; // (the call stops at its own size limit: the caller continues with the same flags)
    const double ms_stage = ms_since(t_0) - st.ms_scan_inside;
    const uint64_t p = st.p;
    if (st.chain_err) {
        (void)hipStreamSynchronize(ix->copy_stream); // (the page-locked ring is free again)
        if (st.chain_err == 1)
            return fail(KMM_ERR_MALFORMED, "kmm_map_bgzf: no BGZF member at compressed byte %llu of the chunk (a gzip file that bgzip did "
                        "not write has no member sizes in its headers: inflate it on the host)", (unsigned long long)p);
        return fail(KMM_ERR_MALFORMED, "kmm_map_bgzf: member at compressed byte %llu claims %u inflated bytes for %u compressed ones",
                    (unsigned long long)p, st.bad_isize, st.bad_ms);
    }
    std::vector<unsigned long long> &m_off = st.m_off, &o_off = st.o_rel;
    for (unsigned long long &o : o_off) // (the inflated bytes carried over from the call before lie in front)
        o += (unsigned long long)carry;
    const uint32_t n_members = (uint32_t)(m_off.size() - 1);
    const int64_t n_used = (int64_t)p, n_total = (int64_t)o_off.back();
    if (consumed_comp)
        *consumed_comp = n_used;
    if (last_chunk && n_used != n_comp) {
        (void)hipStreamSynchronize(ix->copy_stream);
        return fail(KMM_ERR_MALFORMED, "kmm_map_bgzf: the file ends inside a BGZF member (%lld bytes behind the last whole member)",
                    (long long)(n_comp - n_used));
    }
    if (n_members == 0 && !(last_chunk && carry > 0)) {
        HIPCHK(hipStreamSynchronize(ix->copy_stream)); // (the page-locked buffer is free again)
        return KMM_OK;
    }
    const double ms_scan = ms_since(t_0) - ms_stage;
    KMMCHK(ensure(ix->bgzf_raw[cur], (size_t)n_total + 4096));
    KMMCHK(ensure(ix->bgzf_meta[cur], (size_t)(n_members + 1) * 16 + 64));
    KMMCHK(ensure(ix->bgzf_err, 64));
    if (!ix->bgzf_crc.p) { // the CRC32 tables (slicing by 8), once per handle
        std::vector<uint32_t> t(kmm_gz::CRC_TABLE_WORDS);
        for (int kk = 0; kk < 8; ++kk)
            for (uint32_t bb = 0; bb < 256u; ++bb)
                t[(size_t)kk * 256 + bb] = kmm_gz::crc_table_entry(kk, bb);
        for (int kk = 0; kk < kmm_gz::CRC_SHIFT_WORDS; ++kk) // (x^(8 * 2^kk): a part's register moved forward by the bytes behind it)
            t[8 * 256 + kk] = kmm_gz::crc_shift_table_entry(kk);
        KMMCHK(ensure(ix->bgzf_crc, t.size() * 4));
        HIPCHK(hipMemcpy(ix->bgzf_crc.p, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    const uint32_t grid_threads = ((n_members < 65536u ? n_members : 65536u) + 63u) / 64u * 64u;
    if (n_members) {
        KMMCHK(ensure(ix->bgzf_tabs, (size_t)grid_threads * kmm_gz::SCRATCH_BYTES));
        KMMCHK(ensure(ix->bgzf_status, (size_t)n_members + 64));
    }
    uint8_t *d_raw = (uint8_t *)ix->bgzf_raw[cur].p;
    unsigned long long *d_moff = (unsigned long long *)ix->bgzf_meta[cur].p, *d_ooff = d_moff + (n_members + 1);
    if (n_members) {
        if (!staged)
            HIPCHK(hipMemcpyAsync(d_comp, comp, (size_t)n_used, hipMemcpyHostToDevice, ix->copy_stream));
        HIPCHK(hipMemcpyAsync(d_moff, m_off.data(), (size_t)(n_members + 1) * 8, hipMemcpyHostToDevice, ix->copy_stream));
        HIPCHK(hipMemcpyAsync(d_ooff, o_off.data(), (size_t)(n_members + 1) * 8, hipMemcpyHostToDevice, ix->copy_stream));
    }
    const unsigned int err0[4] = {0u, 0xFFFFFFFFu, 0u, 0u};
    HIPCHK(hipMemcpyAsync(ix->bgzf_err.p, err0, sizeof err0, hipMemcpyHostToDevice, ix->copy_stream));
    KMMCHK(stage_copies_done(ix)); // (the handle's stream waits for the copies; the page-locked buffer is free after the sync below)
    if (carry > 0)
        HIPCHK(hipMemcpyAsync(d_raw, ix->bgzf_carry.p, (size_t)carry, hipMemcpyDeviceToDevice, ix->stream));
    if (n_members) {
        // members refused by the inflater are marked; the CRC32s are checked by a kernel of its own behind it (there nothing holds
        // the occupancy down: 2-3 ms of the inflater's 24 per wavefront become a fraction of a millisecond)
        HIPCHK(hipMemsetAsync(ix->bgzf_status.p, 0, n_members, ix->stream));
        hipLaunchKernelGGL(kmm_gz::k_inflate_bgzf, dim3(grid_threads / 64u), dim3(64), 0, ix->stream, d_comp, d_moff, d_ooff, d_raw, n_members,
                           (uint8_t *)ix->bgzf_tabs.p, (const uint32_t *)nullptr, (unsigned int *)ix->bgzf_err.p,
                           (unsigned long long *)nullptr, (uint8_t *)ix->bgzf_status.p);
        hipLaunchKernelGGL(kmm_gz::k_crc_bgzf, dim3((n_members + 63u) / 64u), dim3(256), 0, ix->stream, d_comp, d_moff, d_ooff,
                           (const uint8_t *)d_raw, n_members, (const uint32_t *)ix->bgzf_crc.p, (unsigned int *)ix->bgzf_err.p,
                           (const uint8_t *)ix->bgzf_status.p);
        HIPCHK(hipGetLastError());
    }
    // The chunk BEHIND this one (kmm_map_bgzf_hint_next: the caller's bytes go on where this chunk ends) is staged and walked NOW,
    // under this chunk's inflate kernel: its 20 ms over PCIe would otherwise stand in front of its own 46 ms of kernel.  The next
    // call starts where this call's whole members end — the head of a member that this chunk cut off is copied in front.
    if (ix->bgzf_hint_ptr && ix->bgzf_hint_ptr == comp + n_comp && ix->bgzf_hint_n > 0 && n_members > 0 && n_used <= n_comp) {
        const int nxt = cur ^ 1;
        const int64_t head_len = n_comp - n_used, total = head_len + ix->bgzf_hint_n;
        bool ok = hipStreamSynchronize(ix->copy_stream) == hipSuccess; // (this chunk's copies have landed: the ring is free)
        if (ok && ix->bgzf_used[nxt])
            ok = hipStreamWaitEvent(ix->copy_stream, ix->bgzf_done[nxt], 0) == hipSuccess;
        if (ok && ix->bgzf_comp[nxt].cap < (size_t)total + 64) // (growing it frees it first: a device-wide wait, once)
            ok = ensure(ix->bgzf_comp[nxt], (size_t)total + 64) == KMM_OK;
        if (ok && head_len > 0)
            ok = hipMemcpyAsync(ix->bgzf_comp[nxt].p, d_comp + n_used, (size_t)head_len, hipMemcpyDeviceToDevice, ix->copy_stream) == hipSuccess;
        if (ok && bgzf_stage_and_scan(ix, comp + n_used, head_len, total, (uint8_t *)ix->bgzf_comp[nxt].p, CALL_CAP - PRE_CARRY, ix->bgzf_pre) == KMM_OK &&
            ix->bgzf_pre.staged && !ix->bgzf_pre.chain_err) {
            ix->bgzf_pre_valid = true;
            ix->bgzf_pre_from = comp + n_used;
            ix->bgzf_pre_n = total;
            ix->bgzf_pre_buf = nxt;
            ix->bgzf_prestaged_calls++;
        } else {
            (void)hipGetLastError();
        }
        (void)hipStreamSynchronize(ix->copy_stream); // (between calls the ring is free: its last slots have landed, ~2 ms)
    }
    ix->bgzf_hint_ptr = nullptr;
    ix->bgzf_hint_n = 0;
    // A RANK'S SHARE of a file (kmer_mapper map with several ranks, bgzf_ranges.py): its first member starts inside a record
    // that belongs to the rank before it — "bgzf_head_skip" bytes of the stream's first member are passed over — and its last
    // member holds the start of the next rank's first record — only "bgzf_tail_stop" bytes of the last member are taken.
    int64_t head = 0, n_raw = n_total;
    if (new_stream) {
        head = ix->bgzf_head_skip;
        ix->bgzf_head_skip = 0;
    }
    if (last_chunk && ix->bgzf_tail_stop >= 0) {
        if (n_members == 0 || (unsigned long long)ix->bgzf_tail_stop > o_off[n_members] - o_off[n_members - 1]) {
            (void)hipStreamSynchronize(ix->stream);
            return fail(KMM_ERR_INVALID_ARG, "kmm_map_bgzf: bgzf_tail_stop %lld lies beyond the last member's %llu bytes",
                        (long long)ix->bgzf_tail_stop, n_members ? o_off[n_members] - o_off[n_members - 1] : 0ull);
        }
        n_raw = (int64_t)o_off[n_members - 1] + ix->bgzf_tail_stop;
        ix->bgzf_tail_stop = -1;
    }
    if (head > n_raw) {
        (void)hipStreamSynchronize(ix->stream);
        return fail(KMM_ERR_INVALID_ARG, "kmm_map_bgzf: bgzf_head_skip %lld lies beyond the %lld bytes of the call", (long long)head,
                    (long long)n_raw);
    }
    unsigned int err[4] = {0, 0, 0, 0};
    uint8_t last_byte = 10;
    HIPCHK(hipMemcpyAsync(err, ix->bgzf_err.p, sizeof err, hipMemcpyDeviceToHost, ix->stream));
    if (last_chunk && n_raw > head)
        HIPCHK(hipMemcpyAsync(&last_byte, d_raw + n_raw - 1, 1, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream)); // (CRC32 / ISIZE of every member are checked before a byte is mapped)
    const double ms_inflate = ms_since(t_0) - ms_scan - ms_stage;
    ix->bgzf_calls++;
    ix->bgzf_members += n_members;
    if (err[0]) {
        static const char *why[] = {"", "header", "reserved block type", "stored block", "code lengths", "Huffman code", "invalid symbol",
                                    "distance too far back", "more data than ISIZE", "compressed data ended early", "less data than ISIZE",
                                    "CRC32 mismatch"};
        ix->bgzf_carry_len = 0;
        return fail(KMM_ERR_MALFORMED, "kmm_map_bgzf: %u corrupt BGZF member(s); the first starts at compressed byte %llu of the chunk: %s",
                    err[0], err[1] < n_members ? m_off[err[1]] : 0ull, err[2] < 12 ? why[err[2]] : "?");
    }
    if (last_chunk && n_raw > head && last_byte != 10) { // a last line without its newline gets one (as the file readers do)
        HIPCHK(hipMemsetAsync(d_raw + n_raw, 10, 1, ix->stream));
        ++n_raw;
    }
    int64_t used = 0, recs = 0;
    if (n_raw > head)
        KMMCHK(kmm_map_records(ix, d_raw + head, n_raw - head, fmt, k, max_freq, also_revcomp, lut, &used, &recs));
    used += head;
    if (n_records)
        *n_records = recs;
    if (verbose)
        fprintf(stderr, "libkmm: kmm_map_bgzf: %u members, %lld -> %lld bytes: member scan %.2f ms, buffers + staging + copy %.2f ms%s, "
                "copy tail + inflate kernel%s %.2f ms, records %.2f ms\n", n_members, (long long)n_used, (long long)n_total, ms_scan, ms_stage,
                from_pre ? " (staged and walked under the call before)" : "", ix->bgzf_pre_valid ? " + the next chunk's staging" : "",
                ms_inflate, ms_since(t_0) - ms_scan - ms_stage - ms_inflate);
    const int64_t tail = n_raw - used;
    if (last_chunk && tail > 0) {
        ix->bgzf_carry_len = 0;
        return fail(KMM_ERR_MALFORMED, "kmm_map_bgzf: the stream ends with %lld bytes that form no complete record", (long long)tail);
    }
    if (tail > 0) {
        KMMCHK(ensure(ix->bgzf_carry, (size_t)tail + 64)); // (may free and reallocate: a device-wide sync, rare)
        HIPCHK(hipMemcpyAsync(ix->bgzf_carry.p, d_raw + used, (size_t)tail, hipMemcpyDeviceToDevice, ix->stream));
    }
    ix->bgzf_carry_len = tail;
    HIPCHK(hipEventRecord(ix->bgzf_done[cur], ix->stream));
    ix->bgzf_used[cur] = true;
    return KMM_OK;
}

int kmm_in_index(kmm_index_t *ix, const uint64_t *kmers, int64_t n, uint8_t *out)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    if (n < 0 || (n > 0 && (!kmers || !out)))
        return fail(KMM_ERR_INVALID_ARG, "NULL argument or n negative");
    if (n == 0)
        return KMM_OK;
    HIPCHK(hipSetDevice(ix->device));
    KMMCHK(ensure_direct(ix));
    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    bool staged = false;
    const uint64_t *d_kmers = nullptr;
    KMMCHK(stage_in<uint64_t>(ix, s.kmers, kmers, (size_t)n, &d_kmers, &staged));
    KMMCHK(stage_copies_done(ix));
    const bool out_dev = is_device_ptr(out);
    DevBuf tmp;
    uint8_t *d_out = out;
    if (!out_dev) {
        KMMCHK(ensure(tmp, (size_t)n));
        d_out = (uint8_t *)tmp.p;
    }
    hipLaunchKernelGGL(k_in_index, dim3(grid_for(ix, (n + 255) / 256, 32)), dim3(256), 0, ix->stream,
                       d_kmers, n, view_of(ix), d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !out_dev)
        e = hipMemcpyAsync(out, d_out, (size_t)n, hipMemcpyDeviceToHost, ix->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ix->stream);
    release(tmp);
    if (e != hipSuccess)
        return fail(KMM_ERR_HIP, "kmm_in_index: %s", hipGetErrorString(e));
    return stage_release(ix, s, staged);
}

int kmm_extract_kmers(int device, const uint8_t *bases, const int64_t *read_offsets, int64_t n_reads,
                      int k, const uint8_t *lut, uint64_t *out, int64_t n_out)
{
    KMMCHK(check_k(k));
    if (n_reads < 0 || n_out < 0)
        return fail(KMM_ERR_INVALID_ARG, "negative size");
    if (n_reads == 0)
        return n_out == 0 ? KMM_OK : fail(KMM_ERR_INVALID_ARG, "n_out != 0 for zero reads");
    if (!read_offsets)
        return fail(KMM_ERR_INVALID_ARG, "read_offsets is NULL");
    int ndev = 0;
    hipError_t e0 = hipGetDeviceCount(&ndev);
    if (e0 != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        return fail(KMM_ERR_HIP, "no HIP device available: libkmm has no CPU fallback");
    }
    HIPCHK(hipSetDevice(device));
    const bool offs_dev = is_device_ptr(read_offsets);
    int64_t ends[2];
    if (offs_dev) {
        HIPCHK(hipMemcpy(&ends[0], read_offsets, 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&ends[1], read_offsets + n_reads, 8, hipMemcpyDeviceToHost));
    } else {
        ends[0] = read_offsets[0];
        ends[1] = read_offsets[n_reads];
        for (int64_t r = 0; r < n_reads; ++r)
            if (read_offsets[r + 1] < read_offsets[r])
                return fail(KMM_ERR_INVALID_ARG, "read_offsets not non-decreasing at read %lld", (long long)r);
    }
    if (ends[0] != 0)
        return fail(KMM_ERR_INVALID_ARG, "read_offsets[0] must be 0");
    const int64_t total = ends[1];
    if (total < 0)
        return fail(KMM_ERR_INVALID_ARG, "read_offsets[n_reads] negative");
    if (total == 0)
        return n_out == 0 ? KMM_OK : fail(KMM_ERR_INVALID_ARG, "n_out != 0 but the reads are empty");
    if (!bases || (n_out > 0 && !out))
        return fail(KMM_ERR_INVALID_ARG, "bases / out is NULL");

    DevBuf d_bases, d_offs, d_lut, d_out, d_bad, d_tf, d_cnt, d_sup;
    int rc = KMM_OK;
    hipError_t e = hipSuccess;
    hipStream_t st = nullptr; // a stream of its own: other handles' work on this device is not stalled
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto copy = [&](void *dst, const void *src, size_t bytes, hipMemcpyKind kind) -> hipError_t {
        hipError_t x = hipMemcpyAsync(dst, src, bytes, kind, st);
        return x == hipSuccess ? hipStreamSynchronize(st) : x;
    };
    const bool out_dev = n_out == 0 || is_device_ptr(out);
    uint8_t lutbuf[256];
    if (lut) {
        if (is_device_ptr(lut))
            e = copy(lutbuf, lut, 256, hipMemcpyDeviceToHost);
        else
            memcpy(lutbuf, lut, 256);
    } else {
        default_lut(lutbuf);
    }
    unsigned long long bad[2] = {NO_BAD, NO_BAD};
    int64_t produced = 0;
    const int64_t n_tiles = (total + TILE_T - 1) / TILE_T;
    const int64_t sub_tiles = (int64_t)1 << 20; // 1024 super-tiles of 1024 tiles per round
    do {
        if (e != hipSuccess) break;
        ReadsView rv;
        memset(&rv, 0, sizeof rv);
        rv.total = total;
        rv.n_reads = n_reads;
        if (is_device_ptr(bases)) {
            rv.bases = bases;
        } else {
            if ((rc = ensure(d_bases, (size_t)total))) break;
            if ((e = copy(d_bases.p, bases, (size_t)total, hipMemcpyHostToDevice))) break;
            rv.bases = (const uint8_t *)d_bases.p;
        }
        if (offs_dev) {
            rv.offsets = read_offsets;
        } else {
            if ((rc = ensure(d_offs, (size_t)(n_reads + 1) * 8))) break;
            if ((e = copy(d_offs.p, read_offsets, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice))) break;
            rv.offsets = (const int64_t *)d_offs.p;
        }
        uint64_t *p_out = out;
        if (!out_dev) {
            if ((rc = ensure(d_out, (size_t)n_out * 8))) break;
            p_out = (uint64_t *)d_out.p;
        }
        const int64_t max_tiles = n_tiles < sub_tiles ? n_tiles : sub_tiles;
        const int max_super = (int)((max_tiles + 1023) / 1024);
        if ((rc = ensure(d_lut, 256))) break;
        if ((rc = ensure(d_bad, 16))) break;
        if ((rc = ensure(d_tf, (size_t)(total / 32 + 2) * 4))) break; // read-start bitset
        if ((rc = ensure(d_cnt, (size_t)max_super * 1024 * 4))) break;
        if ((rc = ensure(d_sup, (size_t)max_super * 4 + 16))) break;
        if ((e = copy(d_lut.p, lutbuf, 256, hipMemcpyHostToDevice))) break;
        if ((e = copy(d_bad.p, bad, 16, hipMemcpyHostToDevice))) break;
        rv.lut = (const uint8_t *)d_lut.p;
        rv.first_bad = (unsigned long long *)d_bad.p;
        rv.start_bits = (const uint32_t *)d_tf.p;
        rv.n_start_words = total / 32 + 2;
        if ((e = hipMemsetAsync(d_tf.p, 0, (size_t)rv.n_start_words * 4, st))) break;
        hipLaunchKernelGGL(k_mark_starts, dim3((unsigned)((n_reads + 256) / 256 < 65536 ? (n_reads + 256) / 256 : 65536)),
                           dim3(256), 0, st, rv.offsets, n_reads, total, (uint32_t *)d_tf.p);
        uint32_t *tile_cnt = (uint32_t *)d_cnt.p;
        uint32_t *super_tot = (uint32_t *)d_sup.p;
        uint32_t *d_total = super_tot + max_super;
        for (int64_t t0 = 0; t0 < n_tiles && rc == KMM_OK; t0 += sub_tiles) {
            const int64_t t1 = t0 + sub_tiles < n_tiles ? t0 + sub_tiles : n_tiles;
            const int n_super = (int)((t1 - t0 + 1023) / 1024);
            int64_t g = t1 - t0;
            if (g > 65536) g = 65536;
            if ((e = hipMemsetAsync(tile_cnt, 0, (size_t)n_super * 1024 * 4, st))) break;
            hipLaunchKernelGGL((k_extract_count<TILE_S, MODE_GENERAL>), dim3((unsigned)g), dim3(256), 0, st, rv,
                               k, t0, t1, tile_cnt);
            hipLaunchKernelGGL(k_rec_scan1, dim3(n_super), dim3(1024), 0, st, tile_cnt, super_tot);
            hipLaunchKernelGGL(k_super_scan, dim3(1), dim3(1024), 0, st, super_tot, n_super, d_total);
            uint32_t sub_total = 0;
            if ((e = hipGetLastError())) break;
            if ((e = copy(&sub_total, d_total, 4, hipMemcpyDeviceToHost))) break;
            if (produced + (int64_t)sub_total > n_out) { // never write past the caller's buffer
                rc = fail(KMM_ERR_INVALID_ARG, "n_out=%lld but the reads hold more k-mers", (long long)n_out);
                break;
            }
            if (sub_total)
                hipLaunchKernelGGL((k_extract_write<TILE_S, MODE_GENERAL>), dim3((unsigned)g), dim3(256), 0, st,
                                   rv, k, t0, t1, tile_cnt, super_tot, p_out + produced);
            produced += sub_total;
        }
        if (rc != KMM_OK || e != hipSuccess) break;
        if ((e = hipGetLastError())) break;
        if ((e = hipStreamSynchronize(st))) break;
        if ((e = copy(bad, d_bad.p, 16, hipMemcpyDeviceToHost))) break;
        if (produced != n_out) {
            rc = fail(KMM_ERR_INVALID_ARG, "n_out=%lld but the reads hold %lld k-mers", (long long)n_out,
                      (long long)produced);
            break;
        }
        if (!out_dev && n_out)
            if ((e = copy(out, d_out.p, (size_t)n_out * 8, hipMemcpyDeviceToHost))) break;
    } while (0);
    release(d_bases); release(d_offs); release(d_lut); release(d_out); release(d_bad); release(d_tf);
    release(d_cnt); release(d_sup);
    (void)hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    if (rc != KMM_OK)
        return rc;
    if (e != hipSuccess)
        return fail(KMM_ERR_HIP, "kmm_extract_kmers: %s", hipGetErrorString(e));
    if (bad[0] != NO_BAD)
        return fail(KMM_ERR_INVALID_BASE, "read byte at offset %llu is not a nucleotide under the "
                    "lookup table (the reference's DNA encoder raises here)", bad[0]);
    return KMM_OK;
}

int kmm_build_index(int device, const uint64_t *kmers, const int32_t *nodes, int64_t n, uint64_t modulo,
                    int32_t *hashes_to_index, int32_t *n_kmers, uint64_t *kmers_out, int32_t *nodes_out,
                    uint16_t *frequencies_out)
{
    if (n < 0 || modulo < 1)
        return fail(KMM_ERR_INVALID_ARG, "n negative or modulo < 1");
    if (n > 0x7FFFFFFFll || modulo > 0x7FFFFFFFull)
        return fail(KMM_ERR_INVALID_ARG, "n / modulo exceed the int32 arrays of the index format");
    if (!hashes_to_index || !n_kmers || (n > 0 && (!kmers || !nodes || !kmers_out || !nodes_out || !frequencies_out)))
        return fail(KMM_ERR_INVALID_ARG, "NULL argument");
    int ndev = 0;
    hipError_t e0 = hipGetDeviceCount(&ndev);
    if (e0 != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        return fail(KMM_ERR_HIP, "no HIP device available: libkmm has no CPU fallback");
    }
    HIPCHK(hipSetDevice(device));
    const uint64_t M = modulo;
    const uint64_t magic = magic_for(M);
    DevBuf d_km, d_nd, d_nk, d_h2i, d_cur, d_src, d_ko, d_no, d_fo, d_list, d_ksort;
    std::vector<DevBuf> scratch(2 * SCAN_MAX_LEVELS);
    int rc = KMM_OK;
    hipError_t e = hipSuccess;
    hipStream_t st = nullptr; // a stream of its own: other handles' work on this device is not stalled
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto up = [&](DevBuf &b, const void *src, size_t bytes, const void **dev) -> bool {
        if (bytes == 0 || is_device_ptr(src)) {
            *dev = src;
            return true;
        }
        if ((rc = ensure(b, bytes)))
            return false;
        if ((e = hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st)))
            return false;
        *dev = b.p;
        return true;
    };
    auto down = [&](void *dst, const void *dev, size_t bytes) -> bool {
        if (bytes == 0 || dst == dev)
            return true;
        e = hipMemcpyAsync(dst, dev, bytes, is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
        return e == hipSuccess;
    };
    do {
        const void *p_km = nullptr, *p_nd = nullptr;
        if (!up(d_km, kmers, (size_t)n * 8, &p_km)) break;
        if (!up(d_nd, nodes, (size_t)n * 4, &p_nd)) break;
        // outputs: work in place when the caller's arrays are already on the device
        uint32_t *w_nk = is_device_ptr(n_kmers) ? (uint32_t *)n_kmers : nullptr;
        uint32_t *w_h2i = is_device_ptr(hashes_to_index) ? (uint32_t *)hashes_to_index : nullptr;
        if (!w_nk) { if ((rc = ensure(d_nk, (size_t)M * 4))) break; w_nk = (uint32_t *)d_nk.p; }
        if (!w_h2i) { if ((rc = ensure(d_h2i, (size_t)M * 4))) break; w_h2i = (uint32_t *)d_h2i.p; }
        if ((rc = ensure(d_cur, (size_t)M * 4))) break;
        if ((e = hipMemsetAsync(w_nk, 0, (size_t)M * 4, st))) break;
        if ((e = hipMemsetAsync(d_cur.p, 0, (size_t)M * 4, st))) break;
        int64_t g = (n + 255) / 256;
        if (g > 65536) g = 65536;
        if (g < 1) g = 1;
        if (n > 0)
            hipLaunchKernelGGL(k_bi_hist, dim3((unsigned)g), dim3(256), 0, st, (const uint64_t *)p_km, n, M, magic, w_nk);
        if ((rc = scan_exclusive(w_nk, w_h2i, M, scratch, 0, st))) break;
        if (n > 0) {
            uint64_t *w_ko = is_device_ptr(kmers_out) ? kmers_out : nullptr;
            int32_t *w_no = is_device_ptr(nodes_out) ? nodes_out : nullptr;
            uint16_t *w_fo = is_device_ptr(frequencies_out) ? frequencies_out : nullptr;
            if (!w_ko) { if ((rc = ensure(d_ko, (size_t)n * 8))) break; w_ko = (uint64_t *)d_ko.p; }
            if (!w_no) { if ((rc = ensure(d_no, (size_t)n * 4))) break; w_no = (int32_t *)d_no.p; }
            if (!w_fo) { if ((rc = ensure(d_fo, (size_t)n * 2))) break; w_fo = (uint16_t *)d_fo.p; }
            if ((rc = ensure(d_src, (size_t)n * 4))) break;
            hipLaunchKernelGGL(k_bi_scatter, dim3((unsigned)g), dim3(256), 0, st, (const uint64_t *)p_km, n, M, magic,
                               w_h2i, (uint32_t *)d_cur.p, (uint32_t *)d_src.p);
            hipLaunchKernelGGL(k_bi_place, dim3((unsigned)g), dim3(256), 0, st, (const uint64_t *)p_km,
                               (const int32_t *)p_nd, n, M, magic, w_h2i, w_nk, (const uint32_t *)d_src.p, w_ko,
                               w_no, w_fo);
            // buckets with more than BI_BIG entries: listed, then ordered by one workgroup each
            const size_t list_cap = (size_t)n / BI_BIG + 2;
            if ((rc = ensure(d_list, (list_cap + 1) * 4))) break;
            if ((rc = ensure(d_ksort, (size_t)n * 8))) break;
            uint32_t *d_count = (uint32_t *)d_list.p + list_cap;
            if ((e = hipMemsetAsync(d_count, 0, 4, st))) break;
            {
                uint64_t gm = (M + 255) / 256;
                if (gm > 65536) gm = 65536;
                hipLaunchKernelGGL(k_bi_list_big, dim3((unsigned)gm), dim3(256), 0, st, w_nk, M, (uint32_t *)d_list.p, d_count);
            }
            hipLaunchKernelGGL(k_bi_big, dim3(1024), dim3(1024), 0, st, (const uint64_t *)p_km, (const int32_t *)p_nd, w_h2i,
                               w_nk, (const uint32_t *)d_list.p, d_count, (uint32_t *)d_src.p, (uint64_t *)d_ksort.p, w_ko,
                               w_no, w_fo);
            if ((e = hipGetLastError())) break;
            if ((e = hipStreamSynchronize(st))) break;
            if (!down(kmers_out, w_ko, (size_t)n * 8)) break;
            if (!down(nodes_out, w_no, (size_t)n * 4)) break;
            if (!down(frequencies_out, w_fo, (size_t)n * 2)) break;
        }
        {   // the format leaves hashes_to_index = 0 for empty buckets (upstream fills only the used ones)
            uint64_t gm = (M + 255) / 256;
            if (gm > 65536) gm = 65536;
            hipLaunchKernelGGL(k_bi_zero_empty, dim3((unsigned)gm), dim3(256), 0, st, w_h2i, w_nk, M);
        }
        if ((e = hipGetLastError())) break;
        if ((e = hipStreamSynchronize(st))) break;
        if (!down(n_kmers, w_nk, (size_t)M * 4)) break;
        if (!down(hashes_to_index, w_h2i, (size_t)M * 4)) break;
    } while (0);
    release(d_km); release(d_nd); release(d_nk); release(d_h2i); release(d_cur); release(d_src);
    release(d_ko); release(d_no); release(d_fo); release(d_list); release(d_ksort);
    for (DevBuf &b : scratch)
        release(b);
    if (st) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
    if (rc != KMM_OK)
        return rc;
    if (e != hipSuccess)
        return fail(KMM_ERR_HIP, "kmm_build_index: %s", hipGetErrorString(e));
    return KMM_OK;
}

int kmm_get_stats(kmm_index_t *ix, int reset, uint64_t *n_lookups, uint64_t *n_hits)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    HIPCHK(hipSetDevice(ix->device));
    KMMCHK(drain(ix));
    std::vector<unsigned long long> st(KMM_STAT_BYTES / 8);
    HIPCHK(hipMemcpy(st.data(), ix->stats, KMM_STAT_BYTES, hipMemcpyDeviceToHost));
    unsigned long long tot[2] = {0, 0};
    for (int i = 0; i < KMM_STAT_SHARDS; ++i) {
        tot[0] += st[(size_t)i * KMM_STAT_STRIDE];
        tot[1] += st[(size_t)i * KMM_STAT_STRIDE + 1];
    }
    if (n_lookups)
        *n_lookups = tot[0];
    if (n_hits)
        *n_hits = tot[1];
    if (reset)
        HIPCHK(hipMemset(ix->stats, 0, KMM_STAT_BYTES));
    return KMM_OK;
}

int kmm_set_timing(kmm_index_t *ix, int enabled)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    ix->timing = enabled != 0;
    return KMM_OK;
}

int kmm_get_timing(kmm_index_t *ix, int kernel_id, double *kernel_ms, int64_t *n_launches)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    if (kernel_id < 0 || kernel_id >= KMM_N_KERNELS)
        return fail(KMM_ERR_INVALID_ARG, "kernel_id %d outside [0, %d)", kernel_id, KMM_N_KERNELS);
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipStreamSynchronize(ix->stream));
    for (auto &ev : ix->ev_used) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev.start, ev.stop));
        ix->ms_total[ev.kernel_id] += ms;
        ix->launches[ev.kernel_id] += 1;
        ix->ev_free.push_back(ev);
    }
    ix->ev_used.clear();
    if (kernel_ms)
        *kernel_ms = ix->ms_total[kernel_id];
    if (n_launches)
        *n_launches = ix->launches[kernel_id];
    ix->ms_total[kernel_id] = 0.0;
    ix->launches[kernel_id] = 0;
    return KMM_OK;
}

int kmm_set_param(kmm_index_t *ix, const char *name, int64_t value)
{
    if (!ix || !name)
        return fail(KMM_ERR_INVALID_ARG, "NULL argument");
    HIPCHK(hipSetDevice(ix->device));
    if (strncmp(name, "debug_records_", 14) != 0) // (the overlap experiment sets its switches between two calls in flight)
        HIPCHK(hipStreamSynchronize(ix->stream)); // scratch layouts depend on the knobs
    if (!strcmp(name, "path")) {
        if (value < 0 || value > 2)
            return fail(KMM_ERR_INVALID_ARG, "path must be 0 (auto), 1 (direct) or 2 (radix)");
        if (value == 2 && !ix->rx_ok)
            return fail(KMM_ERR_INVALID_ARG, "the radix path is not available for this index (modulo %llu: needs modulo < 2^31 "
                        "and slices whose entries fit LDS)", (unsigned long long)ix->modulo);
        ix->path = (int)value;
    } else if (!strcmp(name, "part_shift")) {
        if (!rx_configure(ix, (int)value))
            return fail(KMM_ERR_INVALID_ARG, "part_shift %lld: needs 0 <= shift <= 13, at most 512 x 512 fine partitions "
                        "and 2^shift small enough for the quotient of a 64-bit k-mer by the modulo to fit beside "
                        "the hash bits", (long long)value);
        ix->rx_ok = ix->rx_pstart != nullptr;
        if (ix->rx_ok) {
            KMMCHK(rx_flush(ix)); // nothing of the old layout may be pending
            KMMCHK(rx_repack_keys(ix));
        }
    } else if (!strcmp(name, "radix_packed_tiles")) {
        ix->rx_packed = value != 0;
    } else if (!strcmp(name, "radix_filter")) {
        // 1 (default): pass 2 drops the k-mers of empty buckets where a coarse partition's bitmap fits LDS (the
        // fan-out is chosen for it); 0: the plain pass 2
        ix->rx_filter = value != 0;
        if (ix->rx_pstart && rx_configure(ix, ix->rx_w)) {
            KMMCHK(rx_flush(ix));
            KMMCHK(rx_repack_keys(ix));
        }
    } else if (!strcmp(name, "fine_bits")) {
        // experiments: split the current slice width's fan-out as F2 = 2^value fine partitions per coarse partition
        if (value < 0 || value > 9 || !rx_configure(ix, ix->rx_w, RX_MAXF, (int)value))
            return fail(KMM_ERR_INVALID_ARG, "fine_bits %lld: both fan-outs must stay within 512", (long long)value);
        ix->rx_ok = ix->rx_pstart != nullptr;
        if (ix->rx_ok) {
            KMMCHK(rx_flush(ix));
            KMMCHK(rx_repack_keys(ix));
        }
    } else if (!strcmp(name, "radix_min_units")) {
        ix->rx_min_units = value;
    } else if (!strcmp(name, "radix_sub_batch_kmers")) {
        // k-mer slots per sub-batch of the radix path: at most 2^32 - 2 blocks (launch_rx), at least a few blocks
        if (value < 4 * RX_B || value > ((int64_t)1 << 32) - 2 * RX_B)
            return fail(KMM_ERR_INVALID_ARG, "radix_sub_batch_kmers outside [%d, 2^32 - %d]", 4 * RX_B, 2 * RX_B);
        ix->rx_sub_cap = value;
        ix->rx_sub_cap_eff = 0;
    } else if (!strcmp(name, "comm_overlap_slices")) {
        // kmm_comm_reduce_counts: node ranges whose flush runs under the previous range's reduce (1: flush, then one reduce)
        if (value < 1 || value > 64)
            return fail(KMM_ERR_INVALID_ARG, "comm_overlap_slices outside [1, 64]");
        ix->comm_slices = (int)value;
        ix->flush_cuts.clear();
    } else if (!strcmp(name, "radix_sorted_flush")) {
        ix->rx_flush_sorted = value != 0;
    } else if (!strcmp(name, "radix_grid_per_cu")) {
        if (value < 1 || value > 2)
            return fail(KMM_ERR_INVALID_ARG, "radix_grid_per_cu must be 1 or 2");
        ix->rx_grid_per_cu = (int)value;
    } else if (!strcmp(name, "count_kmers")) {
        // per-k-mer counting mode (GpuCounter semantics, gpu_counter.py:23-37): every batch takes the radix path
        // and the per-entry hit counts are kept (kmm_get_kmer_counts) besides being summed into the node counts
        if (value && !ix->rx_ok)
            return fail(KMM_ERR_INVALID_ARG, "count_kmers needs the radix path, which is not available for this index");
        KMMCHK(rx_flush(ix));
        HIPCHK(hipStreamSynchronize(ix->stream));
        if (value && !ix->rx_ecnt_acc) {
            const size_t S = ix->rx_S ? ix->rx_S : 1;
            HIPCHK(hipMalloc(&ix->rx_ecnt_acc, S * 4));
            HIPCHK(hipMemset(ix->rx_ecnt_acc, 0, S * 4));
        } else if (!value && ix->rx_ecnt_acc) {
            HIPCHK(hipFree(ix->rx_ecnt_acc));
            ix->rx_ecnt_acc = nullptr;
        }
    } else if (!strcmp(name, "grid_per_cu")) {
        if (value < 1 || value > 1024)
            return fail(KMM_ERR_INVALID_ARG, "grid_per_cu outside [1, 1024]");
        ix->grid_per_cu = (int)value;
    } else if (!strcmp(name, "dynamic_schedule")) {
        ix->dynamic_schedule = value != 0;
    } else if (!strcmp(name, "dyn_chunk")) {
        if (value < 1 || value > 4096)
            return fail(KMM_ERR_INVALID_ARG, "dyn_chunk outside [1, 4096]");
        ix->dyn_chunk = (int)value;
    } else if (!strcmp(name, "occupancy_filter")) {
        ix->use_occ = value != 0;
    } else if (!strcmp(name, "debug_records_copy_stream")) {
        // experiments only (tools/records_overlap_bisect.py): the compaction kernels of kmm_map_records on the copy stream,
        // beside the previous call's passes, WITHOUT mapping the call's reads
        ix->dbg_rec_copy_stream = value != 0;
    } else if (!strcmp(name, "host_pack_threads")) {
        // > 0: reads / raw records that arrive in host memory (default lookup table, radix-sized batch) are packed to 2 bits per
        // base by that many host threads before they cross PCIe (kmm_hostpack.hpp); 0: they cross as they are
        if (value < 0 || value > 256)
            return fail(KMM_ERR_INVALID_ARG, "host_pack_threads outside [0, 256]");
        ix->host_pack_threads = (int)value;
    } else if (!strcmp(name, "host_pack_slice_kb")) {
        // raw bytes per slice of the host records packer (kmm_hostpack.hpp RecordsJob), in KiB; 0 = its default (1024)
        if (value < 0 || value > 65536)
            return fail(KMM_ERR_INVALID_ARG, "host_pack_slice_kb outside [0, 65536]");
        ix->host_pack_slice_kb = value;
    } else if (!strcmp(name, "bgzf_head_skip")) {
        if (value < 0)
            return fail(KMM_ERR_INVALID_ARG, "bgzf_head_skip negative");
        ix->bgzf_head_skip = value;
    } else if (!strcmp(name, "bgzf_tail_stop")) {
        ix->bgzf_tail_stop = value < 0 ? -1 : value;
    } else if (!strcmp(name, "debug_bgzf_ring_slot_kb") || !strcmp(name, "debug_ring_slot_kb")) {
        if (value != 0 && (value < 4 || value > (1 << 20) || (value & (value - 1))))
            return fail(KMM_ERR_INVALID_ARG, "debug_ring_slot_kb: 0 or a power of two in [4, 2^20]");
        ix->dbg_bgzf_slot_kb = (int)value;
    } else if (!strcmp(name, "debug_rx_buffer_limit")) {
        // test hook of the out-of-memory route of launch_rx (the call takes more sub-batches until the buffers fit)
        ix->dbg_rx_buf_limit = value;
    } else if (!strcmp(name, "debug_records_skip")) {
        ix->dbg_rec_skip = (int)value;
    } else if (!strcmp(name, "debug_skew_p2_counter")) {
        // test hook of the conservation self-check: adds `value` to the device-side "gathered by pass 2" counter, as a
        // doubly processed work item would; the next synchronising call must fail with KMM_ERR_INTERNAL
        unsigned long long v = 0;
        HIPCHK(hipMemcpy(&v, ix->stats + 2, 8, hipMemcpyDeviceToHost));
        v += (unsigned long long)value;
        HIPCHK(hipMemcpy(ix->stats + 2, &v, 8, hipMemcpyHostToDevice));
        ix->rx_unchecked = true;

    } else {
        return fail(KMM_ERR_INVALID_ARG, "unknown parameter '%s'", name);
    }
    return KMM_OK;
}

int kmm_get_param(kmm_index_t *ix, const char *name, int64_t *value)
{
    if (!ix || !name || !value)
        return fail(KMM_ERR_INVALID_ARG, "NULL argument");
    if (!strcmp(name, "path"))
        *value = ix->path;
    else if (!strcmp(name, "part_shift"))
        *value = ix->rx_w;
    else if (!strcmp(name, "radix_min_units"))
        *value = ix->rx_min_units;
    else if (!strcmp(name, "radix_grid_per_cu"))
        *value = ix->rx_grid_per_cu;
    else if (!strcmp(name, "host_pack_threads"))
        *value = ix->host_pack_threads;
    else if (!strcmp(name, "host_pack_slice_kb"))
        *value = ix->host_pack_slice_kb;
    else if (!strcmp(name, "host_packed_calls")) // map calls whose flat reads crossed PCIe as 2-bit codes
        *value = ix->host_packed_calls;
    else if (!strcmp(name, "host_packed_record_calls")) // kmm_map_records calls whose sequence lines were packed on the host
        *value = ix->host_packed_record_calls;
    else if (!strcmp(name, "bgzf_prestaged_calls")) // kmm_map_bgzf calls that staged the chunk behind them under their own kernel
        *value = ix->bgzf_prestaged_calls;
    else if (!strcmp(name, "bgzf_members")) // BGZF members inflated on the GPU by kmm_map_bgzf
        *value = ix->bgzf_members;
    else if (!strcmp(name, "bgzf_carry_bytes")) // inflated bytes behind the last complete record, waiting for the next call
        *value = ix->bgzf_carry_len;
    else if (!strcmp(name, "host_cpu_budget")) // cores the process may keep busy (affinity mask, cgroup quota)
        *value = kmm_hostpack::cpu_budget();
    else if (!strcmp(name, "radix_sub_batch_kmers"))
        *value = ix->rx_sub_cap;
    else if (!strcmp(name, "radix_sub_batch_kmers_effective"))
        *value = ix->rx_sub_cap_last;
    else if (!strcmp(name, "comm_overlap_slices"))
        *value = ix->comm_slices;
    else if (!strcmp(name, "comm_sliced_reduces")) // kmm_comm_reduce_counts calls that issued one reduce per node range
        *value = ix->comm_sliced_reduces;
    else if (!strcmp(name, "debug_rx_t1_sum") || !strcmp(name, "debug_rx_start1_sum") || !strcmp(name, "debug_rx_items")) {
        // diagnostics of the latest radix sub-batch (both streams drained first): the k-mers its directory accounts for —
        // by coarse partition (T1), by pass-1 block (last entry of every start1 row) — and its items
        HIPCHK(hipSetDevice(ix->device));
        HIPCHK(hipStreamSynchronize(ix->copy_stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        if (!ix->dbg_T1)
            return fail(KMM_ERR_INVALID_ARG, "no radix batch has run on this handle");
        uint64_t sum = 0;
        if (name[9] == 't') {
            std::vector<uint32_t> t(ix->dbg_F1);
            HIPCHK(hipMemcpy(t.data(), ix->dbg_T1, t.size() * 4, hipMemcpyDeviceToHost));
            for (uint32_t v : t)
                sum += v;
        } else if (name[9] == 'i') {
            uint32_t v = 0;
            HIPCHK(hipMemcpy(&v, ix->dbg_item_base + ix->dbg_F1, 4, hipMemcpyDeviceToHost));
            sum = v;
        } else {
            const size_t ld = (size_t)ix->dbg_F1 + 1;
            std::vector<uint16_t> r((size_t)ix->dbg_NB * ld);
            HIPCHK(hipMemcpy(r.data(), ix->dbg_start1, r.size() * 2, hipMemcpyDeviceToHost));
            for (size_t b = 0; b < ix->dbg_NB; ++b)
                sum += r[b * ld + ld - 1];
        }
        *value = (int64_t)sum;
    }
    else if (!strcmp(name, "radix_sorted_flush"))
        *value = (ix->rx_flush_sorted && ix->rx_norder) ? 1 : 0;
    else if (!strcmp(name, "radix_available"))
        *value = ix->rx_ok ? 1 : 0;
    else if (!strcmp(name, "radix_batches"))
        *value = (int64_t)ix->n_radix_batches;
    else if (!strcmp(name, "direct_batches"))
        *value = (int64_t)ix->n_direct_batches;
    else if (!strcmp(name, "radix_unavailable_reason")) // 0 available, 1 modulo >= 2^31, 2 slices, 3 memory, 4 overlapping buckets
        *value = ix->rx_ok ? 0 : ix->rx_why_not;
    else if (!strcmp(name, "direct_view_resident"))
        *value = ix->buckets ? 1 : 0;
    else if (!strcmp(name, "direct_view_bytes"))
        *value = (int64_t)ix->direct_bytes;
    else if (!strcmp(name, "radix_view_bytes"))
        *value = ix->rx_pstart ? (int64_t)((ix->modulo + 1) * 4 + (ix->rx_S ? ix->rx_S : 1) * (8 + 8 + 2 + 4 + 4 + 4 + (ix->rx_norder ? 8 : 0)))
                               : 0;
    else if (!strcmp(name, "radix_filter"))
        *value = (ix->rx_ok && rx_filter_active(ix)) ? 1 : 0;
    else if (!strcmp(name, "radix_p3_keys_in_lds")) // entries of a slice pass 3 keeps in LDS (the rest is walked in HBM)
        *value = !ix->rx_ok ? 0
                 : ix->rx_w <= 12 ? RX_ECAP
                 : (ix->rx_fits_small && ix->rx_max_slice <= 65535u) ? RX_ECAP
                 : (ix->rx_fits_mid && ix->rx_max_slice <= 65535u && !getenv("KMM_RX_NO_MID")) ? RX_ECAP_MID : RX_ECAP_BIG;
    else if (!strcmp(name, "radix_filter_buckets_per_bit")) // 1, 2 or 4 (0: no filter)
        *value = (ix->rx_ok && rx_filter_active(ix)) ? (1 << ix->rx_occ_shift) : 0;
    else if (!strcmp(name, "radix_packed_tiles"))
        *value = ix->rx_packed ? 1 : 0;
    else if (!strcmp(name, "n_fine_per_coarse"))
        *value = ix->rx_ok ? ix->rx_F2 : 0;
    else if (!strcmp(name, "count_kmers"))
        *value = ix->rx_ecnt_acc ? 1 : 0;
    else if (!strcmp(name, "n_coarse_partitions"))
        *value = ix->rx_ok ? ix->rx_F1 : 0;
    else if (!strcmp(name, "radix_p2_kmers") || !strcmp(name, "radix_p3_kmers") || !strcmp(name, "radix_p2_dropped") ||
             !strncmp(name, "stats_slot_", 11)) {
        // conservation check of the radix path: k-mers gathered by pass 2 / probed by pass 3 since the last
        // kmm_get_stats(reset): both must equal the lookups pass 1 emitted
        HIPCHK(hipSetDevice(ix->device));
        KMMCHK(drain(ix));
        std::vector<unsigned long long> st(KMM_STAT_BYTES / 8);
        HIPCHK(hipMemcpy(st.data(), ix->stats, KMM_STAT_BYTES, hipMemcpyDeviceToHost));
        // ("stats_slot_<n>": raw counter n of the statistics block; slots 4.. are only written by diagnostic builds)
        const int slot = name[0] == 's' ? atoi(name + 11) : !strcmp(name, "radix_p2_dropped") ? KMM_STAT_RX_DROPPED
                                                          : name[7] == '2' ? 2 : 3;
        if (slot < 0 || slot >= KMM_STAT_STRIDE)
            return fail(KMM_ERR_INVALID_ARG, "unknown parameter '%s'", name);
        unsigned long long t = 0;
        for (int i = 0; i < KMM_STAT_SHARDS; ++i)
            t += st[(size_t)i * KMM_STAT_STRIDE + slot];
        *value = (int64_t)t;
    }
    else if (!strcmp(name, "grid_per_cu"))
        *value = ix->grid_per_cu;
    else if (!strcmp(name, "dynamic_schedule"))
        *value = ix->dynamic_schedule ? 1 : 0;
    else if (!strcmp(name, "occupancy_filter"))
        *value = (ix->use_occ && ix->occ) ? 1 : 0;
    else if (!strcmp(name, "bloom_filter_bytes"))
        *value = (ix->use_occ && ix->occ) ? (int64_t)ix->bloom_words * 4 : 0;
    else if (!strcmp(name, "occupancy_bits_per_bucket"))
        *value = (ix->use_occ && ix->occ && !ix->bloom_words) ? (1 << ix->occ_shift) : 0;
    else if (!strcmp(name, "wide_buckets"))
        *value = ix->wide ? 1 : 0;
    else if (!strcmp(name, "n_partitions"))
        *value = ix->rx_ok ? ix->rx_PF : 0;
    else
        return fail(KMM_ERR_INVALID_ARG, "unknown parameter '%s'", name);
    return KMM_OK;
}

} // extern "C"
