// Where does a lane of k_inflate_bgzf spend its time?  Built by tools/gz_phase.py:
//   hipcc -O3 --offload-arch=gfx950 -DKMM_GZ_TIMERS -shared -fPIC -I kmer_mapper_amd/csrc tools/gz_phase.hip -o /tmp/gz_phase.so
#include <hip/hip_runtime.h>

#include "kmm_gpu_inflate.hpp"

#include <cstdio>
#include <vector>

#define CK(x)                                                                                                         \
    do {                                                                                                              \
        hipError_t e_ = (x);                                                                                          \
        if (e_ != hipSuccess) {                                                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                   \
            return -1;                                                                                                \
        }                                                                                                             \
    } while (0)

// timers_out: grid_threads x 8 ticks of 10 ns (see KMM_GZ_T in the header); err3: members in error, first of them, its code
extern "C" int gz_phase(const uint8_t *comp, uint64_t n_comp, const unsigned long long *m_off, const unsigned long long *o_off,
                        uint32_t n_members, uint64_t n_out, uint8_t *out_host, uint32_t grid_threads, int reps, double *kernel_ms,
                        unsigned long long *timers_out, unsigned int *err3)
{
    uint8_t *d_comp, *d_out, *d_tabs, *d_status;
    unsigned long long *d_m, *d_o, *d_t;
    uint32_t *d_crc;
    unsigned int *d_err;
    CK(hipMalloc(&d_comp, n_comp + 64));
    CK(hipMalloc(&d_out, n_out + 64));
    CK(hipMalloc(&d_tabs, (size_t)grid_threads * kmm_gz::SCRATCH_BYTES));
    CK(hipMalloc(&d_m, (n_members + 1) * 8));
    CK(hipMalloc(&d_o, (n_members + 1) * 8));
    CK(hipMalloc(&d_t, (size_t)grid_threads * 64));
    CK(hipMalloc(&d_crc, kmm_gz::CRC_TABLE_WORDS * 4));
    CK(hipMalloc(&d_err, 16));
    CK(hipMalloc(&d_status, n_members + 64));
    std::vector<uint32_t> t(kmm_gz::CRC_TABLE_WORDS);
    for (int k = 0; k < 8; ++k)
        for (uint32_t b = 0; b < 256; ++b)
            t[(size_t)k * 256 + b] = kmm_gz::crc_table_entry(k, b);
    for (int k = 0; k < kmm_gz::CRC_SHIFT_WORDS; ++k)
        t[8 * 256 + k] = kmm_gz::crc_shift_table_entry(k);
    CK(hipMemcpy(d_crc, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_comp, comp, n_comp, hipMemcpyHostToDevice));
    CK(hipMemset(d_comp + n_comp, 0, 64));
    CK(hipMemcpy(d_m, m_off, (n_members + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_o, o_off, (n_members + 1) * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        const unsigned int init[4] = {0u, 0xFFFFFFFFu, 0u, 0u};
        CK(hipMemcpy(d_err, init, 16, hipMemcpyHostToDevice));
        CK(hipMemset(d_t, 0, (size_t)grid_threads * 64));
        CK(hipMemset(d_status, 0, n_members + 64));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kmm_gz::k_inflate_bgzf, dim3(grid_threads / 64u), dim3(64), 0, 0, d_comp, d_m, d_o, d_out, n_members, d_tabs,
                           (const uint32_t *)nullptr, d_err, d_t, d_status);
        hipLaunchKernelGGL(kmm_gz::k_crc_bgzf, dim3((n_members + 63u) / 64u), dim3(256), 0, 0, d_comp, d_m, d_o, (const uint8_t *)d_out,
                           n_members, (const uint32_t *)d_crc, d_err, (const uint8_t *)d_status);
        CK(hipGetLastError());
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best)
            best = ms;
    }
    *kernel_ms = best;
    CK(hipMemcpy(timers_out, d_t, (size_t)grid_threads * 64, hipMemcpyDeviceToHost));
    unsigned int err[4];
    CK(hipMemcpy(err, d_err, 16, hipMemcpyDeviceToHost));
    err3[0] = err[0];
    err3[1] = err[1];
    err3[2] = err[2];
    if (out_host)
        CK(hipMemcpy(out_host, d_out, n_out, hipMemcpyDeviceToHost));
    for (void *p : {(void *)d_comp, (void *)d_out, (void *)d_tabs, (void *)d_m, (void *)d_o, (void *)d_t, (void *)d_crc, (void *)d_err, (void *)d_status})
        (void)hipFree(p);
    return 0;
}
