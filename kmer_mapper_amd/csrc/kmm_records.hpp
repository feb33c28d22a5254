// kmm_records.hpp — part of libkmm (MI355X / gfx950); included by kmm.hip inside its anonymous namespace.
// Records mode pre-pass: newline census of raw FASTQ / two-line FASTA chunks.
#pragma once

// ------------------------------------------------------------------------------------------------
// Records mode pre-pass: newline census of a raw FASTQ / two-line FASTA chunk (tile = 1024 bytes,
// super-tile = 1024 tiles), so that every tile knows the line number of its first byte, and the
// position where the last complete record ends.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rec_count(const uint8_t *__restrict__ raw, int64_t n,
                                                   int64_t n_tiles, uint32_t *__restrict__ tile_cnt)
{
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); // one wavefront per tile
    if (tile >= n_tiles)
        return;
    const int64_t p = tile * 1024 + lane * 16;
    uint32_t c = 0;
    if ((((uintptr_t)raw) & 15u) == 0 && p + 16 <= n) {
        u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(raw + p));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            c += (uint32_t)__popc(bytes_equal(x[i], 10u));
    } else {
        for (int i = 0; i < 16; ++i)
            if (p + i < n && raw[p + i] == 10u)
                ++c;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        c += __shfl_xor(c, d);
    if (lane == 0)
        tile_cnt[tile] = c;
}

// Inclusive Hillis-Steele scan of s_a[0..1024) by a 1024-thread workgroup.
__device__ __forceinline__ void block_scan_1024(uint32_t *s_a)
{
    const int t = threadIdx.x;
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t v = t >= d ? s_a[t - d] : 0u;
        __syncthreads();
        s_a[t] += v;
        __syncthreads();
    }
}

// One workgroup per super-tile: counts -> exclusive prefix inside the super-tile (in place) + total.
__global__ void __launch_bounds__(1024) k_rec_scan1(uint32_t *tile_cnt, uint32_t *super_tot)
{
    __shared__ uint32_t s_a[1024];
    const int t = threadIdx.x;
    const size_t idx = (size_t)blockIdx.x * 1024 + t;
    const uint32_t c = tile_cnt[idx];
    s_a[t] = c;
    __syncthreads();
    block_scan_1024(s_a);
    tile_cnt[idx] = s_a[t] - c;
    if (t == 1023)
        super_tot[blockIdx.x] = s_a[t];
}

// One workgroup: exclusive prefix over the super-tiles, then the byte position just after the last
// newline that completes a record (records have `period` lines).  out = {consumed, n_records, n_lines}.
__global__ void __launch_bounds__(1024) k_rec_scan2(const uint8_t *__restrict__ raw, int64_t n,
                                                    int n_super, const uint32_t *__restrict__ tile_pre,
                                                    uint32_t *super_tot, uint32_t period, int64_t *out)
{
    __shared__ uint32_t s_a[1024];
    __shared__ uint32_t s_super, s_rem, s_super_cnt;
    __shared__ int64_t s_tile;
    const int t = threadIdx.x;
    const uint32_t c = t < n_super ? super_tot[t] : 0u;
    s_a[t] = c;
    __syncthreads();
    block_scan_1024(s_a);
    const uint32_t excl = s_a[t] - c;
    const uint32_t total = s_a[1023];
    __syncthreads();
    if (t < n_super)
        super_tot[t] = excl;
    const uint32_t target = total - total % period;
    if (target == 0) { // same for every thread
        if (t == 0) {
            out[0] = 0;
            out[1] = 0;
            out[2] = total;
        }
        return;
    }
    if (t < n_super && excl < target && target <= excl + c) {
        s_super = (uint32_t)t;
        s_rem = target - excl;
        s_super_cnt = c;
    }
    __syncthreads();
    const uint32_t sup = s_super, rem = s_rem;
    const uint32_t pre = tile_pre[(size_t)sup * 1024 + t];
    const uint32_t nxt = t < 1023 ? tile_pre[(size_t)sup * 1024 + t + 1] : s_super_cnt;
    if (pre < rem && rem <= nxt)
        s_tile = (int64_t)sup * 1024 + t;
    __syncthreads();
    const int64_t tile = s_tile;
    const uint32_t r = rem - tile_pre[tile];
    const int64_t pos = tile * 1024 + t;
    const uint32_t is_nl = (pos < n && raw[pos] == 10u) ? 1u : 0u;
    s_a[t] = is_nl;
    __syncthreads();
    block_scan_1024(s_a);
    if (is_nl && s_a[t] == r)
        out[0] = pos + 1;
    if (t == 0) {
        out[1] = target / period;
        out[2] = total;
    }
}
