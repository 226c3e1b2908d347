// Microbenchmark of LS-fast step 2 (row transforms + closed form) on gfx950: the generic 32 x 16 kernel against the
// 16 x 32 kernel fft_rows512_power_kernel, on a synthetic 85-target chunk (2.1 GB of intermediate),
// each timed on "cold" data (written long ago) and right after a kernel that rewrites the intermediate (what the pipeline
// does: the column kernel has just written it).  Also times the column kernel.  Every variant's spectra are compared with
// the generic kernel's.
// Build (from lightkurve_amd/csrc after `make`):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ../../tools/microbench/lsfast_rows ../../tools/microbench/lsfast_rows.hip \
//         capi.o ls.o bls.o regress.o flatten.o pld.o pgsmooth.o fold.o ingest.o
#include "../../lightkurve_amd/csrc/lsfast.hip"

#include <cstdio>
#include <vector>
using namespace lk;

__global__ void fill_kernel(double2 *g, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = (i + seed) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        g[i] = make_double2((double)(h & 0xFFFFF) / 1048576.0 - 0.5, (double)((h >> 20) & 0xFFFFF) / 1048576.0 - 0.5);
    }
}
__global__ void rewrite_kernel(double2 *g, size_t n) {  // rewrites every element with its own value (dirty lines, same data)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double2 v = g[i];
        v.x += 0.0;
        g[i] = v;
    }
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                    \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 85, m1 = 10, m2 = 9, N1 = 1 << m1, N2 = 1 << m2;
    const int64_t M = 100000;
    const size_t nfft = (size_t)N1 * N2, ng = (size_t)B * 3 * nfft;
    double2 *d_in, *d_mid;
    double *d_pref, *d_p;
    FastStats *d_stats;
    int64_t *d_off;
    int *d_rows;
    PeakPart *d_peaks;
    CK(hipMalloc(&d_in, ng * 16));
    CK(hipMalloc(&d_mid, ng * 16));
    CK(hipMalloc(&d_pref, (size_t)B * M * 8));
    CK(hipMalloc(&d_p, (size_t)B * M * 8));
    CK(hipMalloc(&d_stats, B * sizeof(FastStats)));
    CK(hipMalloc(&d_off, (B + 1) * 8));
    CK(hipMalloc(&d_rows, B * 16));
    CK(hipMalloc(&d_peaks, (size_t)B * 512 * sizeof(PeakPart)));
    std::vector<FastStats> st(B);
    std::vector<int64_t> off(B + 1);
    std::vector<int> rows(B * 4);
    for (int b = 0; b < B; ++b) {
        st[b] = FastStats{1.0, 0.0, 1.0, 1234.5 + b, 0.0, 1.0};
        off[b] = (int64_t)b * 20000;
        rows[b * 4] = rows[b * 4 + 1] = 103;
        rows[b * 4 + 2] = 205;
        rows[b * 4 + 3] = 1;
    }
    off[B] = (int64_t)B * 20000;
    CK(hipMemcpy(d_stats, st.data(), B * sizeof(FastStats), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_off, off.data(), (B + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rows, rows.data(), B * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, d_in, ng, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, d_mid, ng, 77u);
    CK(hipDeviceSynchronize());
    lk_handle h;
    const double f0 = 0.01, df = 0.0036;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double mid_bytes = (double)ng * 16 + (double)B * M * 8;
    auto timeit = [&](const char *name, auto launch, double *pout, double bytes) -> int {
        for (int mode = 0; mode < 2; ++mode) {  // 0: cold data, 1: right after a rewrite of the intermediate
            float total = 0.f;
            const int reps = 6;
            for (int r = 0; r < reps + 1; ++r) {
                if (mode) hipLaunchKernelGGL(rewrite_kernel, dim3(8192), dim3(256), 0, 0, d_mid, ng);
                CK(hipEventRecord(e0, 0));
                launch(pout);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r) total += ms;
            }
            CK(hipGetLastError());
            const double us = total / reps * 1e3;
            printf("%-44s %-14s %8.1f us  %6.2f TB/s\n", name, mode ? "after-rewrite" : "cold", us, bytes / us * 1e-6);
        }
        return 0;
    };
    auto compare = [&](const char *name) -> int {
        std::vector<double> a((size_t)B * M), b((size_t)B * M);
        CK(hipMemcpy(a.data(), d_pref, a.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), d_p, b.size() * 8, hipMemcpyDeviceToHost));
        double mx = 0.0, md = 0.0;
        size_t nan_mismatch = 0;
        for (size_t i = 0; i < a.size(); ++i) {
            if ((a[i] != a[i]) != (b[i] != b[i])) ++nan_mismatch;
            if (a[i] == a[i] && b[i] == b[i]) {
                mx = std::max(mx, std::fabs(a[i]));
                md = std::max(md, std::fabs(a[i] - b[i]));
            }
        }
        printf("    %-40s max|diff| / max|ref| = %.3e   (NaN mismatches %zu)\n", name, md / mx, nan_mismatch);
        CK(hipMemset(d_p, 0, (size_t)B * M * 8));
        return 0;
    };
    // ---- generic kernel (the round-3 default)
    const FusedArgs fa{d_off, d_stats, 0, f0, df, M, 1, LK_NORM_LK_AMPLITUDE, nullptr, d_pref, d_peaks};
    timeit("generic 32x16 <5,4,4> (2 WG/CU)", [&](double *p) {
        FusedArgs a = fa;
        a.power = p;
        launch_rows_power_t<5, 4, 4>(&h, m1, B, d_mid, a, PRUNED_CT, 0);
    }, d_pref, mid_bytes);
    // ---- the 16 x 32 kernel (prefetch of the next grid, non-temporal loads)
    (void)want_lds(&h, reinterpret_cast<const void *>(fft_rows512_power_kernel), 160 * 1024);
    timeit("rows512 16x32 (2 WG/CU)", [&](double *p) {
        FusedArgs a = fa;
        a.power = p;
        launch_rows512(&h, m1, B, d_mid, a, 0);
    }, d_p, mid_bytes);
    compare("rows512");
    // ---- column kernel (reads the spread rows, writes the intermediate)
    {
        const double cbytes = (double)ng * 16 + (double)B * (103 + 103 + 205) * N2 * 16;
        float total = 0.f;
        for (int r = 0; r < 7; ++r) {
            CK(hipEventRecord(e0, 0));
            launch_cols_pruned_t<8>(&h, m1, m2, B * 3, d_in, d_rows, d_mid, SpreadArgs{nullptr, nullptr, nullptr, d_off, d_stats, 0, f0, df, 1, nullptr, 0}, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) total += ms;
        }
        CK(hipGetLastError());
        printf("%-44s %-14s %8.1f us  %6.2f TB/s\n", "fft_cols_pruned_kernel<8>", "", total / 6 * 1e3, cbytes / (total / 6 * 1e3) * 1e-6);
    }
    return 0;
}
