// Would a different row order of the LS-fast intermediate make step 1's stores (and step 2's loads) faster?  (gfx950)
// The intermediate is [grid][column tile c / 16][row k1 < 1024][c % 16] complex doubles: a row is 256 B.  Step 1
// (fft_cols_pruned_kernel<8>: Q = 4 passes) writes rows k1 = 4 q + s in pass s — 256 pieces of 256 B at a 1-KB stride per
// workgroup and pass; step 2 (fft_rows512_power_kernel) reads 8 consecutive rows (2 KB runs) of each of 32 column tiles.
//   order 0 (product):  position(k1) = k1
//   order 1 (by pass):  position(k1) = (k1 % 4) * 256 + k1 / 4   -> step 1 writes 64 KB contiguous per pass,
//                                                                   step 2 reads 4 runs of 512 B per column tile
// Bare access patterns only (no arithmetic): `store` = step 1's stores, `load` = step 2's loads, same workgroup shapes
// (256 threads, 2 per CU by LDS), 84 targets x 3 grids x 8 MiB.
// Build: hipcc --offload-arch=gfx950 -O3 -o ls_layout ls_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef double d2v __attribute__((ext_vector_type(2)));
constexpr int N1 = 1024, N2 = 512, CT = 16;

__device__ __forceinline__ int pos_of(int k1, int order) { return order ? ((k1 & 3) << 8) + (k1 >> 2) : k1; }

// step 1's stores: workgroup (column tile, grid); thread (f = column of the tile, jk < 16); pass s: rows 4 (jk + 16 kb) + s
__global__ __launch_bounds__(256) void store_kernel(double2 *__restrict__ out, int order, int spin, int pad = 0) {
    extern __shared__ double lds_pad[];  // 70 KB: two workgroups per CU, as in the product
    const int tid = threadIdx.x, f = tid & 15, jk = tid >> 4;
    double2 *O = out + ((size_t)blockIdx.y * (N2 / CT) + blockIdx.x) * ((size_t)N1 * CT + pad);
    double2 v = make_double2((double)tid, (double)blockIdx.x);
    for (int s = 0; s < 4; ++s) {
        for (int i = 0; i < spin; ++i) v.x = fma(v.x, 1.0000001, 1e-9);  // stands in for the pass's arithmetic
        if (spin < 0) lds_pad[tid] = v.x;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const int k1 = 4 * (jk + 16 * kb) + s;
            O[(size_t)pos_of(k1, order) * CT + f] = v;
            v.y += 1.0;
        }
    }
}

// step 2's loads: workgroup = 8 rows of one target's three grids; thread (jl, f1 = row, jh): 16 loads per grid
__global__ __launch_bounds__(256) void load_kernel(const double2 *__restrict__ in, double *__restrict__ sink, int order, int pad = 0) {
    extern __shared__ double lds_pad[];
    const int tid = threadIdx.x, jl = tid & 15, f1 = (tid >> 4) & 7, jh = tid >> 7;
    const int tiles = N1 / 8, T = blockIdx.x, tgt = T / tiles, r0 = (T % tiles) * 8;
    double acc = 0.0;
    for (int g = 0; g < 3; ++g) {
        const size_t ts = (size_t)N1 * CT + pad;  // column-tile stride
        const double2 *G = in + ((size_t)tgt * 3 + g) * ((N2 / CT) * ts) + (size_t)jh * ts + (size_t)pos_of(r0 + f1, order) * CT + jl;
        d2v v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(G + (size_t)i * 2 * ts));
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += v[i].x + v[i].y;
    }
    if (acc == 12345.678) sink[0] = acc + lds_pad[0];
}

int main() {
    const int ntgt = 84;
    const size_t n = (size_t)ntgt * 3 * N1 * N2;
    double2 *buf;
    double *sink;
    CK(hipMalloc(&buf, n * 16 + (size_t)ntgt * 3 * (N2 / CT) * 4096 * 16));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, n * 16 + (size_t)ntgt * 3 * (N2 / CT) * 4096 * 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(store_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(load_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double gb = (double)n * 16 / 1e9;
    for (int spin : {0, 200, 600})
        for (int order = 0; order < 2; ++order) {
            float best = 1e30f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                store_kernel<<<dim3(N2 / CT, ntgt * 3), 256, 70 * 1024>>>(buf, order, spin);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("store  order %d  arithmetic %3d fma per pass: %7.1f us  %6.0f GB/s\n", order, spin, best * 1e3, gb / (best * 1e-3));
        }
    for (int order = 0; order < 2; ++order) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            load_kernel<<<dim3(ntgt * (N1 / 8)), 256, 36 * 1024>>>(buf, sink, order);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("load   order %d: %7.1f us  %6.0f GB/s\n", order, best * 1e3, gb / (best * 1e-3));
    }
    // padded column-tile stride (the 32 tiles a row workgroup reads sit 256 KB apart: a power of two)
    for (int pad : {0, 16, 64, 256, 1024, 4096}) {
        float bl = 1e30f, bs = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            store_kernel<<<dim3(N2 / CT, ntgt * 3), 256, 70 * 1024>>>(buf, 0, 200, pad);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < bs) bs = ms;
            CK(hipEventRecord(e0));
            load_kernel<<<dim3(ntgt * (N1 / 8)), 256, 36 * 1024>>>(buf, sink, 0, pad);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < bl) bl = ms;
        }
        printf("tile stride 256 KB + %6d B: store %7.1f us %6.0f GB/s   load %7.1f us %6.0f GB/s\n", pad * 16, bs * 1e3, gb / (bs * 1e-3),
               bl * 1e3, gb / (bl * 1e-3));
    }
    // store then load back to back (what the chunk loop does): does the order change what the Infinity Cache keeps?
    for (int order = 0; order < 2; ++order) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            store_kernel<<<dim3(N2 / CT, ntgt * 3), 256, 70 * 1024>>>(buf, order, 200);
            load_kernel<<<dim3(ntgt * (N1 / 8)), 256, 36 * 1024>>>(buf, sink, order);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("store + load  order %d: %7.1f us\n", order, best * 1e3);
    }
    return 0;
}
