// ingest.hip — the steps BEFORE the hot path for a ragged batch of light curves on gfx950: what a
// pipeline over a LightCurveCollection does per target in Python/astropy today, done for all targets in a few launches so
// that batches reach the periodogram / flatten / regression kernels without a host round trip per light curve.
//
//   lk_ingest_batch         LightCurve.remove_nans (src/lightkurve/lightcurve.py:1300-1327) + .normalize (:1216-1292):
//                           drop the cadences whose flux is NaN, order preserved, repack the batch contiguously, divide
//                           flux and flux_err by nanmedian(flux)
//   lk_transit_mask_batch   LightCurve.create_transit_mask (:2967-3037): |((t - t0 + P/2) % P) - P/2| < duration / 2 for
//                           any of the planets (numpy `%`: result carries the divisor's sign)
//   lk_bin_batch            LightCurve.bin (:1558-1763) over astropy aggregate_downsample (astropy@4.3.1
//                           timeseries/downsample.py:12-125): equal-width bins from time_bin_start, nanmean of the flux,
//                           root-mean-square of flux_err (or nanstd of the flux when there are no errors)
#include <cmath>
#include <vector>

#include "block_select.hpp"
#include "lk_common.hpp"

namespace lk {

__device__ __forceinline__ double np_mod_ingest(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

// ------------------------------------------------------------------------------------------------ remove_nans + normalize
__global__ __launch_bounds__(256) void ingest_count_kernel(const double *__restrict__ flux, const int64_t *__restrict__ n_off,
                                                            int64_t *__restrict__ kept) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t lo = n_off[b], n = n_off[b + 1] - lo;
    long long c = 0;
    for (int64_t i = tid; i < n; i += 256) c += isnan(flux[lo + i]) ? 0 : 1;
    __shared__ long long sh[8];
    const long long tot = block_count_fast(c, sh);
    if (tid == 0) kept[b] = tot;
}

__global__ __launch_bounds__(1024) void ingest_scan_kernel(const int64_t *__restrict__ kept, int B, int64_t *__restrict__ new_off) {
    // exclusive prefix sum of B counts by one workgroup: per-thread chunks + wave scans
    __shared__ long long sh[32];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int chunk = (B + nt - 1) / nt;
    const int lo = min(tid * chunk, B), hi = min(lo + chunk, B);
    long long c = 0;
    for (int i = lo; i < hi; ++i) c += kept[i];
    long long inc = c;
    const int lane = tid & 63, nw = nt >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const long long v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) sh[tid >> 6] = inc;
    __syncthreads();
    long long base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += sh[w];
    long long run = base + inc - c;
    for (int i = lo; i < hi; ++i) {
        new_off[i] = run;
        run += kept[i];
    }
    if (tid == nt - 1) {
        long long tot = 0;
        for (int w = 0; w < nw; ++w) tot += sh[w];
        new_off[B] = tot;
    }
}

__global__ __launch_bounds__(512) void ingest_pack_kernel(const double *__restrict__ t, const double *__restrict__ flux,
                                                           const double *__restrict__ err, const int64_t *__restrict__ n_off,
                                                           const int64_t *__restrict__ new_off, int normalize,
                                                           double *__restrict__ t_out, double *__restrict__ f_out,
                                                           double *__restrict__ e_out, double *__restrict__ median_out,
                                                           int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long ing_lds[];
    unsigned long long *sh = ing_lds;                               // 512 words
    double *cand = reinterpret_cast<double *>(ing_lds + 512);       // cap doubles
    int *shi = reinterpret_cast<int *>(cand + cap);
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int64_t lo = n_off[b];
    const int n = (int)(n_off[b + 1] - lo);
    const int64_t olo = new_off[b];
    const int nk = (int)(new_off[b + 1] - olo);
    t += lo;
    flux += lo;
    if (err) err += lo;
    t_out += olo;
    f_out += olo;
    if (e_out) e_out += olo;
    // order-preserving compaction: every wave owns a contiguous strip, positions from ballot prefixes
    const int nw = nt >> 6, wv = tid >> 6, lane = tid & 63;
    const int strip = ((n + nw - 1) / nw + 63) & ~63;
    const int k_lo = min(wv * strip, n), k_hi = min(k_lo + strip, n);
    int c = 0;
    for (int k = k_lo + lane; k < k_hi; k += 64) c += isnan(flux[k]) ? 0 : 1;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) shi[wv] = c;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += shi[w];
    for (int k0 = k_lo; k0 < k_hi; k0 += 64) {
        const int k = k0 + lane;
        const bool in = k < k_hi;
        const double f = in ? flux[k] : 0.0;
        const bool m = in && !isnan(f);
        const unsigned long long bal = __ballot(m);
        if (m) {
            const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            t_out[pos] = t[k];
            f_out[pos] = f;
            if (e_out) e_out[pos] = err ? err[k] : __longlong_as_double(0x7ff8000000000000ll);
        }
        base += __popcll(bal);
    }
    __syncthreads();
    // nanmedian of the flux = median of the kept values; then flux /= median, flux_err /= median (:1283-1284)
    auto val = [&](int i) { return f_out[i]; };
    auto keep = [&](int) { return true; };
    const double med = block_median_sampled(nk, (long long)nk, val, keep, sh, cand, cap);
    if (tid == 0 && median_out) median_out[b] = med;
    if (normalize) {
        for (int i = tid; i < nk; i += nt) {
            f_out[i] = f_out[i] / med;
            if (e_out) e_out[i] = e_out[i] / med;
        }
    }
}

int ingest_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux, const double *err,
                  int normalize, double *t_out, double *f_out, double *e_out, int64_t *new_off_host, double *median_out,
                  hipStream_t stream) {
    LK_REQUIRE(B >= 0 && n_off_host != nullptr && new_off_host != nullptr, "bad batch description");
    if (B == 0) {
        new_off_host[0] = 0;
        return LK_OK;
    }
    LK_REQUIRE(t && flux && t_out && f_out, "NULL buffer");
    LK_REQUIRE(n_off_host[0] == 0, "n_off[0] must be 0");
    for (int b = 0; b < B; ++b) {
        const int64_t n = n_off_host[b + 1] - n_off_host[b];
        LK_REQUIRE(n >= 0 && n < ((int64_t)1 << 30), "target %d has %lld cadences", b, (long long)n);
    }
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 8 * 3 + 4096);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    int64_t *d_kept = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    int64_t *d_new = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    rc = h->stage.copy(d_off, n_off_host, (size_t)(B + 1) * 8, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(ingest_count_kernel, dim3(B), dim3(256), 0, stream, flux, d_off, d_kept);
    hipLaunchKernelGGL(ingest_scan_kernel, dim3(1), dim3(1024), 0, stream, d_kept, B, d_new);
    constexpr int cap = 4096;
    const size_t lds = 512 * 8 + (size_t)cap * 8 + 64 * 4;
    {  // per handle = per device (a process may drive several GPUs)
        const int rc_ = want_lds(h, reinterpret_cast<const void *>(ingest_pack_kernel), 160 * 1024);
        if (rc_) return rc_;
    }
    hipLaunchKernelGGL(ingest_pack_kernel, dim3(B), dim3(512), lds, stream, t, flux, err, d_off, d_new, normalize, t_out,
                       f_out, e_out, median_out, cap);
    LK_HIP_CHECK(hipMemcpyAsync(new_off_host, d_new, (size_t)(B + 1) * 8, hipMemcpyDeviceToHost, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));  // the caller needs the new offsets to address the packed batch
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ FITS table -> arrays
// lk_fits_unpack_batch: what lightkurve's readers do per file through astropy (src/lightkurve/io/generic.py:21-207,
// io/kepler.py:49-53, io/tess.py:45-48), for B files in three launches.  Input: the raw big-endian bytes of each file's
// BINTABLE (row-major records of row_bytes bytes) and, per file, where TIME / flux / flux_err / quality sit in a record
// and their TFORM type; output: (time, flux, flux_err) as float64, quality as int32, for the rows the readers keep —
// time not NaN (generic.py:98-101) and (quality & bitmask) == 0 (utils.py:115) — packed contiguously, order preserved.
// A workgroup stages 256 records at a time in LDS with coalesced 4-byte loads (a record is ~100 unaligned bytes, so
// per-field global loads would touch every cache line several times), then one thread decodes one record from LDS.
struct FitsDesc {
    int row_bytes, n_rows, off_t, code_t, off_f, code_f, off_e, code_e, off_q, code_q;  // TFORM codes: 0 D 1 E 2 J 3 K 4 I 5 B
};
constexpr int FITS_ROWS = 256;

__device__ __forceinline__ unsigned long long fits_be(const uint8_t *p, int nbytes) {
    unsigned long long v = 0;
    for (int i = 0; i < nbytes; ++i) v = (v << 8) | (unsigned long long)p[i];
    return v;
}
__device__ __forceinline__ double fits_real(const uint8_t *rec, int off, int code) {
    if (code == 0) return __longlong_as_double((long long)fits_be(rec + off, 8));
    return (double)__uint_as_float((unsigned)fits_be(rec + off, 4));
}
__device__ __forceinline__ long long fits_int(const uint8_t *rec, int off, int code) {
    if (code == 2) return (long long)(int)(unsigned)fits_be(rec + off, 4);
    if (code == 3) return (long long)fits_be(rec + off, 8);
    if (code == 4) return (long long)(short)(unsigned short)fits_be(rec + off, 2);
    return (long long)rec[off];
}

// stage records [r0, r0 + nr) of one file into LDS (word loads: the file's table starts 4-byte aligned in `raw`)
__device__ __forceinline__ void fits_stage(const uint8_t *__restrict__ tab, long long r0, int nr, int row_bytes,
                                           uint8_t *lds_bytes) {
    const long long b0 = r0 * row_bytes, b1 = b0 + (long long)nr * row_bytes;
    const long long w0 = b0 >> 2, w1 = (b1 + 3) >> 2;  // words covering the byte range (reads <= 3 bytes past: in bounds, see launcher)
    const unsigned int *src = reinterpret_cast<const unsigned int *>(tab);
    unsigned int *dst = reinterpret_cast<unsigned int *>(lds_bytes);
    for (long long w = w0 + threadIdx.x; w < w1; w += blockDim.x) dst[w - w0] = src[w];
}

// STAGE = false: records too long for the LDS stage (target-pixel files: kilobytes per cadence) — the few scalar fields
// are read straight from global memory.  keep_nan_time: Kepler target-pixel files keep cadences whose TIME is NaN
// (targetpixelfile.py:2120-2122), every other reader drops them.  row_out (nullable): source row of every kept record.
template <bool PACK, bool STAGE>
__global__ __launch_bounds__(256) void fits_unpack_kernel(const uint8_t *__restrict__ raw, const int64_t *__restrict__ raw_off,
                                                           const FitsDesc *__restrict__ desc,
                                                           const int64_t *__restrict__ bitmask, int64_t *__restrict__ kept,
                                                           const int64_t *__restrict__ new_off, double *__restrict__ t_out,
                                                           double *__restrict__ f_out, double *__restrict__ e_out,
                                                           int *__restrict__ q_out, int keep_nan_time,
                                                           int *__restrict__ row_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fits_lds[];
    __shared__ int s_cnt[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const FitsDesc d = desc[b];
    const uint8_t *tab = raw + raw_off[b];
    const long long mask = bitmask[b];
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    long long base = PACK ? new_off[b] : 0, total = 0;
    for (long long r0 = 0; r0 < d.n_rows; r0 += FITS_ROWS) {
        const int nr = (int)min((long long)FITS_ROWS, (long long)d.n_rows - r0);
        __syncthreads();
        const uint8_t *rec;
        if (STAGE) {
            fits_stage(tab, r0, nr, d.row_bytes, fits_lds);
            __syncthreads();
            const int skew = (int)((r0 * d.row_bytes) & 3);  // the staged words start at a 4-byte boundary
            rec = fits_lds + skew + (size_t)tid * d.row_bytes;
        } else {
            rec = tab + (size_t)(r0 + min(tid, nr - 1)) * d.row_bytes;
        }
        double tv = qnan, fv = qnan, ev = qnan;
        long long qv = 0;
        bool keep = false;
        if (tid < nr) {
            tv = fits_real(rec, d.off_t, d.code_t);
            if (d.off_q >= 0) qv = fits_int(rec, d.off_q, d.code_q);
            keep = (keep_nan_time || !isnan(tv)) && (qv & mask) == 0;
            if (PACK && keep) {
                fv = fits_real(rec, d.off_f, d.code_f);
                if (d.off_e >= 0) ev = fits_real(rec, d.off_e, d.code_e);
            }
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_cnt[wv] = __popcll(bal);
        __syncthreads();
        int before = 0, all = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wv) before += s_cnt[w];
            all += s_cnt[w];
        }
        if (PACK && keep) {
            const long long pos = base + before + __popcll(bal & ((1ull << lane) - 1ull));
            // a kept cadence without a time reads 0, as TargetPixelFile.time makes it (targetpixelfile.py:333-335)
            t_out[pos] = (keep_nan_time && !isfinite(tv)) ? 0.0 : tv;
            f_out[pos] = fv;
            if (e_out) e_out[pos] = ev;
            if (q_out) q_out[pos] = (int)qv;
            if (row_out) row_out[pos] = (int)(r0 + tid);
        }
        base += all;
        total += all;
    }
    if (!PACK && tid == 0) kept[b] = total;
}

int fits_unpack_launch(lk_handle *h, int B, const uint8_t *raw, const int64_t *raw_off_host, const int32_t *desc_host,
                       const int64_t *bitmask_host, double *t_out, double *f_out, double *e_out, int32_t *q_out,
                       int64_t *new_off_host, hipStream_t stream) {
    LK_REQUIRE(B >= 0 && raw_off_host != nullptr && desc_host != nullptr && new_off_host != nullptr, "bad batch description");
    if (B == 0) {
        new_off_host[0] = 0;
        return LK_OK;
    }
    LK_REQUIRE(raw && bitmask_host && t_out && f_out, "NULL buffer");
    int max_row = 0;
    for (int b = 0; b < B; ++b) {
        const int32_t *d = desc_host + (size_t)b * 10;
        LK_REQUIRE(d[0] >= 1 && d[0] <= 512 && d[1] >= 0, "file %d: record length %d outside 1..512 bytes", b, d[0]);
        LK_REQUIRE((raw_off_host[b] & 3) == 0, "file %d: the table must start at a multiple of 4 bytes in `raw`", b);
        LK_REQUIRE(raw_off_host[b + 1] - raw_off_host[b] >= (int64_t)d[0] * d[1] + 3,
                   "file %d: `raw` holds fewer bytes than rows x record length (+3 bytes of padding)", b);
        const int widths[6] = {8, 4, 4, 8, 2, 1};
        const int offs[4] = {d[2], d[4], d[6], d[8]}, codes[4] = {d[3], d[5], d[7], d[9]};
        for (int c = 0; c < 4; ++c) {
            if (c >= 2 && offs[c] < 0) continue;  // flux_err / quality may be absent
            LK_REQUIRE(codes[c] >= 0 && codes[c] <= 5 && offs[c] >= 0 && offs[c] + widths[codes[c]] <= d[0],
                       "file %d: column %d (offset %d, type %d) does not fit the %d-byte record", b, c, offs[c], codes[c], d[0]);
            LK_REQUIRE(c == 3 ? codes[c] >= 2 : codes[c] <= 1, "file %d: column %d has the wrong kind of TFORM", b, c);
        }
        max_row = std::max(max_row, d[0]);
    }
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 8 * 4 + (size_t)B * sizeof(FitsDesc) + 4096);
    if (rc) return rc;
    int64_t *d_roff = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    int64_t *d_mask = (int64_t *)h->ws.alloc((size_t)B * 8);
    int64_t *d_kept = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    int64_t *d_new = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    FitsDesc *d_desc = (FitsDesc *)h->ws.alloc((size_t)B * sizeof(FitsDesc));
    LK_HIP_CHECK(hipMemcpyAsync(d_roff, raw_off_host, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_mask, bitmask_host, (size_t)B * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_desc, desc_host, (size_t)B * sizeof(FitsDesc), hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));  // pageable sources may be reused by the caller
    const size_t lds = (size_t)FITS_ROWS * max_row + 16;
    {
        // (the kernel also has 16 bytes of static LDS: the dynamic part may not claim all 160 KB)
        int rc_ = want_lds(h, reinterpret_cast<const void *>(fits_unpack_kernel<false, true>), 144 * 1024);
        if (!rc_) rc_ = want_lds(h, reinterpret_cast<const void *>(fits_unpack_kernel<true, true>), 144 * 1024);
        if (rc_) return rc_;
    }
    hipLaunchKernelGGL((fits_unpack_kernel<false, true>), dim3(B), dim3(256), lds, stream, raw, d_roff, d_desc, d_mask, d_kept,
                       (const int64_t *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr, (int *)nullptr, 0,
                       (int *)nullptr);
    hipLaunchKernelGGL(ingest_scan_kernel, dim3(1), dim3(1024), 0, stream, d_kept, B, d_new);
    hipLaunchKernelGGL((fits_unpack_kernel<true, true>), dim3(B), dim3(256), lds, stream, raw, d_roff, d_desc, d_mask,
                       (int64_t *)nullptr, d_new, t_out, f_out, e_out, q_out, 0, (int *)nullptr);
    LK_HIP_CHECK(hipMemcpyAsync(new_off_host, d_new, (size_t)(B + 1) * 8, hipMemcpyDeviceToHost, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// Target-pixel files (the input of PLDCorrector): every kept cadence's pixel vectors (FLUX, FLUX_ERR, FLUX_BKG ...: `npix`
// big-endian float32 per column and record) -> float32 cubes [column][kept cadence][pixel].  One workgroup per kept
// cadence; the pixels of a record are contiguous, so the loads are coalesced 4-byte words when the column is aligned.
__global__ __launch_bounds__(128) void fits_cube_kernel(const uint8_t *__restrict__ tab, int row_bytes, const int *__restrict__ rows,
                                                         int ncols, int off0, int off1, int off2, int off3, int npix,
                                                         size_t col_stride, float *__restrict__ out) {
    const int i = blockIdx.x, tid = threadIdx.x;
    const uint8_t *rec = tab + (size_t)rows[i] * row_bytes;
    const int offs[4] = {off0, off1, off2, off3};
    for (int c = 0; c < ncols; ++c) {
        const uint8_t *src = rec + offs[c];
        float *dst = out + (size_t)c * col_stride + (size_t)i * npix;
        if ((reinterpret_cast<uintptr_t>(src) & 3) == 0) {
            const unsigned int *w = reinterpret_cast<const unsigned int *>(src);
            for (int p = tid; p < npix; p += 128) dst[p] = __uint_as_float(__builtin_bswap32(w[p]));
        } else {
            for (int p = tid; p < npix; p += 128) dst[p] = __uint_as_float((unsigned)fits_be(src + 4 * p, 4));
        }
    }
}

int fits_cube_launch(lk_handle *h, const uint8_t *raw, int row_bytes, int n_rows, int off_time, int code_time, int off_qual,
                     int code_qual, int64_t bitmask, int keep_nan_time, int ncols, const int32_t *col_off_host, int npix,
                     double *t_out, int32_t *q_out, float *cubes_out, int64_t *kept_host, hipStream_t stream) {
    LK_REQUIRE(raw && t_out && cubes_out && kept_host && col_off_host, "NULL buffer");
    LK_REQUIRE(row_bytes >= 1 && n_rows >= 0 && ncols >= 1 && ncols <= 4 && npix >= 1, "bad table description");
    LK_REQUIRE(code_time >= 0 && code_time <= 1 && off_time >= 0 && off_time + (code_time ? 4 : 8) <= row_bytes,
               "TIME does not fit the record");
    {
        const int widths[6] = {8, 4, 4, 8, 2, 1};  // D, E, J, K, I, B (as in fits_unpack_launch)
        LK_REQUIRE(off_qual < 0 || (code_qual >= 2 && code_qual <= 5 && off_qual + widths[code_qual] <= row_bytes),
                   "QUALITY does not fit the record");
    }
    for (int c = 0; c < ncols; ++c)
        LK_REQUIRE(col_off_host[c] >= 0 && (int64_t)col_off_host[c] + 4ll * npix <= row_bytes,
                   "pixel column %d (offset %d, %d pixels) does not fit the %d-byte record", c, col_off_host[c], npix, row_bytes);
    if (n_rows == 0) {
        *kept_host = 0;
        return LK_OK;
    }
    h->ws.reset();
    int rc = h->ws.reserve(64 + sizeof(FitsDesc) + (size_t)n_rows * 4 + (size_t)n_rows * 16 + 4096);
    if (rc) return rc;
    int64_t *d_roff = (int64_t *)h->ws.alloc(16), *d_mask = (int64_t *)h->ws.alloc(8);
    int64_t *d_kept = (int64_t *)h->ws.alloc(16), *d_new = (int64_t *)h->ws.alloc(16);
    FitsDesc *d_desc = (FitsDesc *)h->ws.alloc(sizeof(FitsDesc));
    int *d_rows = (int *)h->ws.alloc((size_t)n_rows * 4);
    double *d_dummy = (double *)h->ws.alloc((size_t)n_rows * 8);  // the scalar kernel's flux slot (TIME again), unused
    const int64_t roff[2] = {0, (int64_t)row_bytes * n_rows};
    const FitsDesc desc{row_bytes, n_rows, off_time, code_time, off_time, code_time, -1, 0, off_qual, code_qual};
    LK_HIP_CHECK(hipMemcpyAsync(d_roff, roff, 16, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_mask, &bitmask, 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_desc, &desc, sizeof(FitsDesc), hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));  // the sources are on this function's stack
    hipLaunchKernelGGL((fits_unpack_kernel<false, false>), dim3(1), dim3(256), 16, stream, raw, d_roff, d_desc, d_mask, d_kept,
                       (const int64_t *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr, (int *)nullptr,
                       keep_nan_time, (int *)nullptr);
    hipLaunchKernelGGL(ingest_scan_kernel, dim3(1), dim3(1024), 0, stream, d_kept, 1, d_new);
    hipLaunchKernelGGL((fits_unpack_kernel<true, false>), dim3(1), dim3(256), 16, stream, raw, d_roff, d_desc, d_mask,
                       (int64_t *)nullptr, d_new, t_out, d_dummy, (double *)nullptr, q_out, keep_nan_time, d_rows);
    int64_t newoff[2];
    LK_HIP_CHECK(hipMemcpyAsync(newoff, d_new, 16, hipMemcpyDeviceToHost, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));
    const int64_t kept = newoff[1];
    *kept_host = kept;
    if (kept > 0) {
        const int o[4] = {col_off_host[0], ncols > 1 ? col_off_host[1] : 0, ncols > 2 ? col_off_host[2] : 0,
                          ncols > 3 ? col_off_host[3] : 0};
        hipLaunchKernelGGL(fits_cube_kernel, dim3((unsigned)kept), dim3(128), 0, stream, raw, row_bytes, d_rows, ncols, o[0], o[1],
                           o[2], o[3], npix, (size_t)n_rows * npix, cubes_out);
    }
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ create_transit_mask
__global__ __launch_bounds__(256) void transit_mask_kernel(const double *__restrict__ t, int64_t ntot,
                                                            const int64_t *__restrict__ n_off, int B,
                                                            const double *__restrict__ period,
                                                            const double *__restrict__ duration,
                                                            const double *__restrict__ transit_time,
                                                            const int *__restrict__ p_off, uint8_t *__restrict__ mask) {
    // one workgroup per (target, 256-cadence tile): blockIdx.y = target
    const int b = blockIdx.y;
    const int64_t lo = n_off[b], n = n_off[b + 1] - lo;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double tv = t[lo + i];
    bool in_transit = false;
    for (int p = p_off[b]; p < p_off[b + 1]; ++p) {
        const double per = period[p], hp = per / 2.0;
        in_transit |= fabs(np_mod_ingest(tv - transit_time[p] + hp, per) - hp) < 0.5 * duration[p];
    }
    mask[lo + i] = in_transit ? 1 : 0;
}

int transit_mask_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const int *p_off_host,
                        const double *period_host, const double *duration_host, const double *transit_time_host,
                        uint8_t *mask, hipStream_t stream) {
    LK_REQUIRE(B >= 0 && n_off_host && p_off_host, "bad batch description");
    if (B == 0) return LK_OK;
    LK_REQUIRE(t && mask, "NULL buffer");
    LK_REQUIRE(B <= 65535, "at most 65535 targets per call");
    const int np_tot = p_off_host[B];
    LK_REQUIRE(p_off_host[0] == 0 && np_tot >= 0, "p_off must be prefix offsets starting at 0");
    LK_REQUIRE(np_tot == 0 || (period_host && duration_host && transit_time_host), "NULL planet parameters");
    int64_t nmax = 0;
    for (int b = 0; b < B; ++b) {
        LK_REQUIRE(p_off_host[b + 1] >= p_off_host[b], "p_off must be non-decreasing");
        nmax = std::max(nmax, n_off_host[b + 1] - n_off_host[b]);
    }
    for (int p = 0; p < np_tot; ++p) LK_REQUIRE(period_host[p] != 0.0, "period must be non-zero");
    if (nmax == 0) return LK_OK;
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 12 + (size_t)np_tot * 24 + 8 * 256 + 4096);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    int *d_poff = (int *)h->ws.alloc((size_t)(B + 1) * 4);
    double *d_par = (double *)h->ws.alloc((size_t)std::max(np_tot, 1) * 24);
    std::vector<double> par((size_t)std::max(np_tot, 1) * 3, 1.0);
    for (int p = 0; p < np_tot; ++p) {
        par[p] = period_host[p];
        par[(size_t)np_tot + p] = duration_host[p];
        par[2 * (size_t)np_tot + p] = transit_time_host[p];
    }
    LK_HIP_CHECK(hipMemcpyAsync(d_off, n_off_host, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_poff, p_off_host, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_par, par.data(), par.size() * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));  // host staging buffers go out of scope
    hipLaunchKernelGGL(transit_mask_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)B), dim3(256), 0, stream, t,
                       n_off_host[B], d_off, B, d_par, d_par + np_tot, d_par + 2 * (size_t)np_tot, d_poff, mask);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ bin
// One thread per (target, bin).  edges[k] = the running sum 0, s, s + s, ... in seconds exactly as numpy's cumsum forms it
// (host-built, shared by all targets); a cadence with relative time r [s] belongs to bin k when edges[k] < r <= edges[k+1]
// (r == 0 -> bin 0) and r < edges[n_bins] (downsample.py:85-97).  Times are sorted, so a bin is a contiguous cadence range,
// found by two binary searches.
__global__ __launch_bounds__(256) void bin_kernel(const double *__restrict__ t, const double *__restrict__ flux,
                                                   const double *__restrict__ err, const int64_t *__restrict__ n_off,
                                                   const int64_t *__restrict__ bin_off, int B,
                                                   const double *__restrict__ start, const double *__restrict__ edges,
                                                   double bin_size_sec, const uint8_t *__restrict__ has_err,
                                                   double *__restrict__ t_out, double *__restrict__ f_out,
                                                   double *__restrict__ e_out) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= bin_off[B]) return;
    int b_lo = 0, b_hi = B;  // the target that owns global bin g: last b with bin_off[b] <= g
    while (b_hi - b_lo > 1) {
        const int mid = (b_lo + b_hi) >> 1;
        if (bin_off[mid] <= g)
            b_lo = mid;
        else
            b_hi = mid;
    }
    const int b = b_lo;
    const int k = (int)(g - bin_off[b]), nb = (int)(bin_off[b + 1] - bin_off[b]);
    const int64_t lo = n_off[b];
    const int n = (int)(n_off[b + 1] - lo);
    const double t0 = start[b];
    auto rel = [&](int i) { return (t[lo + i] - t0) * 86400.0; };
    // first cadence with rel > x (strict) / rel >= x
    auto first_gt = [&](double x) {
        int a = 0, c = n;
        while (a < c) {
            const int mid = (a + c) >> 1;
            if (rel(mid) > x)
                c = mid;
            else
                a = mid + 1;
        }
        return a;
    };
    auto first_ge = [&](double x) {
        int a = 0, c = n;
        while (a < c) {
            const int mid = (a + c) >> 1;
            if (rel(mid) >= x)
                c = mid;
            else
                a = mid + 1;
        }
        return a;
    };
    const int i0 = k == 0 ? first_ge(edges[0]) : first_gt(edges[k]);
    int i1 = first_gt(edges[k + 1]);
    i1 = min(i1, first_ge(edges[nb]));  // keep: rel < last edge
    double s = 0.0, s2 = 0.0;
    int cf = 0, ce = 0;
    for (int i = i0; i < i1; ++i) {
        const double f = flux[lo + i];
        if (!isnan(f)) {
            s += f;
            ++cf;
        }
        if (err && has_err[b]) {
            const double e = err[lo + i];
            if (isfinite(e)) {  // rmse: sqrt(nansum(x^2) / count(isfinite(x)))  (lightcurve.py:167-172)
                ++ce;
            }
            if (!isnan(e)) s2 = fma(e, e, s2);
        }
    }
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double mean = cf ? s / (double)cf : qnan;
    double eo;
    if (err && has_err[b]) {
        eo = ce ? sqrt(s2 / (double)ce) : qnan;
    } else {  // no usable errors anywhere in this light curve: nanstd of the flux in the bin
        double v = 0.0;
        for (int i = i0; i < i1; ++i) {
            const double f = flux[lo + i];
            if (!isnan(f)) {
                const double d = f - mean;
                v = fma(d, d, v);
            }
        }
        eo = cf ? sqrt(v / (double)cf) : qnan;
    }
    if (i1 <= i0) eo = qnan;
    t_out[g] = t0 + (edges[k] + bin_size_sec / 2.0) / 86400.0;
    f_out[g] = (i1 > i0) ? mean : qnan;
    e_out[g] = eo;
}

int bin_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux, const double *err,
               const int64_t *bin_off_host, const double *start_host, const double *edges_host, int64_t n_edges,
               double bin_size_sec, const uint8_t *has_err_host, double *t_out, double *f_out, double *e_out,
               hipStream_t stream) {
    LK_REQUIRE(B >= 0 && n_off_host && bin_off_host, "bad batch description");
    if (B == 0) return LK_OK;
    LK_REQUIRE(t && flux && start_host && edges_host && has_err_host && t_out && f_out && e_out, "NULL buffer");
    LK_REQUIRE(bin_size_sec > 0.0, "time_bin_size must be positive");
    const int64_t nbins = bin_off_host[B];
    LK_REQUIRE(bin_off_host[0] == 0 && nbins >= 0, "bin_off must be prefix offsets starting at 0");
    for (int b = 0; b < B; ++b)
        LK_REQUIRE(bin_off_host[b + 1] >= bin_off_host[b] && bin_off_host[b + 1] - bin_off_host[b] < n_edges,
                   "target %d needs more bin edges than were passed", b);
    if (nbins == 0) return LK_OK;
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 16 + (size_t)B * 9 + (size_t)n_edges * 8 + 6 * 256 + 4096);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8), *d_boff = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    double *d_start = (double *)h->ws.alloc((size_t)B * 8), *d_edges = (double *)h->ws.alloc((size_t)n_edges * 8);
    uint8_t *d_he = (uint8_t *)h->ws.alloc((size_t)B);
    LK_HIP_CHECK(hipMemcpyAsync(d_off, n_off_host, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_boff, bin_off_host, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_start, start_host, (size_t)B * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_edges, edges_host, (size_t)n_edges * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_he, has_err_host, (size_t)B, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));
    hipLaunchKernelGGL(bin_kernel, dim3((unsigned)((nbins + 255) / 256)), dim3(256), 0, stream, t, flux, err, d_off, d_boff,
                       B, d_start, d_edges, bin_size_sec, d_he, t_out, f_out, e_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
