// Would an L2-resident intermediate beat the HBM round trip of the LS-fast four-step FFT?  
// One "team" per XCD = the workgroups that report the same HW_REG_XCC_ID (grid = one workgroup per CU; team size checked,
// not assumed).  Per iteration a team moves one S-byte sub-transform intermediate the way the two FFT steps would:
//   step 1: every workgroup writes its 1/32 slice (contiguous, column-tiled layout [c/16][q][c%16]) with plain 16-B stores
//   team barrier (monotonic counter in the XCD's L2: vmcnt(0) -> __syncthreads -> one relaxed agent atomic add; waiters
//                 poll with relaxed agent loads and then invalidate their L1 — no L2 write-back: same XCD, same L2)
//   step 2: every workgroup reads 8 rows across all 32 column tiles (32 runs of S/1024 bytes), verifies every value
//   team barrier (the buffer is overwritten by the next iteration)
// Reported: us per iteration for S = 2 MB / 1 MB / 0.5 MB (in L2) against the same traffic streamed through buffers far
// larger than L2 + Infinity Cache (the HBM round trip of today's kernels), the bare barrier cost, and the number of stale
// values seen (must be 0 for the scheme to be usable).
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_l2_subfft xcd_l2_subfft.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}

struct Ctl {
    unsigned team_count[8];     // workgroups registered per XCD
    unsigned bar[8][16];        // per-XCD monotonic barrier counters (one 64-B line each)
    unsigned all_in;            // global arrival counter (once, at kernel start)
    unsigned long long stale;   // values that did not match
    unsigned long long cyc[8];  // per-XCD clock64 span of the timed loop (team rank 0)
};

__device__ __forceinline__ void team_barrier(unsigned *ctr, unsigned team, unsigned &epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    epoch += team;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int guard = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch && ++guard < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1: this CU's L1
    }
    __syncthreads();
}

// mode 0: buffer per XCD reused every iteration (L2-resident if it fits); mode 1: a fresh slice of a huge buffer per iteration
__global__ __launch_bounds__(256) void team_kernel(Ctl *ctl, double2 *buf, size_t s_elems, int iters, int mode, size_t big_elems, int do_io) {
    extern __shared__ double2 pad_lds[];  // sized so that one workgroup fits a CU
    const unsigned xcc = xcc_id();
    __shared__ unsigned s_rank, s_team;
    if (threadIdx.x == 0) {
        s_rank = __hip_atomic_fetch_add(&ctl->team_count[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctl->all_in, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int guard = 0;
        while (__hip_atomic_load(&ctl->all_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++guard < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        s_team = __hip_atomic_load(&ctl->team_count[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned rank = s_rank, team = s_team;
    unsigned epoch = 0;
    unsigned *ctr = &ctl->bar[xcc][0];
    const size_t slice = s_elems / team;            // step-1 slice of this workgroup (contiguous)
    const size_t run = s_elems / team / team;       // step-2: `team` runs of this many elements, one per column tile
    unsigned long long bad = 0;
    team_barrier(ctr, team, epoch);
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        double2 *B = mode == 0 ? buf + (size_t)xcc * s_elems
                               : buf + (((size_t)it * 8 + xcc) * s_elems) % big_elems;
        if (do_io) {
#pragma unroll 8
            for (size_t i = threadIdx.x; i < slice; i += 256) {
                const size_t a = (size_t)rank * slice + i;
                B[a] = make_double2((double)(a + (size_t)it * 7919), (double)it);
            }
        }
        team_barrier(ctr, team, epoch);
        if (do_io) {
#pragma unroll 8
            for (size_t idx = threadIdx.x; idx < slice; idx += 256) {  // (column tile, element of the run) flattened
                const size_t tl = idx / run, i = idx - tl * run;
                const size_t a = tl * slice + (size_t)rank * run + i;
                const double2 v = B[a];
                if (v.x != (double)(a + (size_t)it * 7919) || v.y != (double)it) ++bad;
            }
        }
        team_barrier(ctr, team, epoch);
    }
    const unsigned long long t1 = clock64();
    if (bad) atomicAdd(&ctl->stale, bad);
    if (threadIdx.x == 0 && rank == 0) ctl->cyc[xcc] = t1 - t0;
}

int main() {
    Ctl *ctl;
    hipMalloc(&ctl, sizeof(Ctl));
    const size_t big_bytes = (size_t)3 << 30;
    double2 *buf;
    hipMalloc(&buf, big_bytes);
    hipMemset(buf, 0, big_bytes);
    hipFuncSetAttribute(reinterpret_cast<const void *>(team_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 400;
    printf("# %d iterations per launch, 256 workgroups x 256 threads (one per CU)\n", iters);
    for (int do_io = 0; do_io < 2; ++do_io)
        for (int mode = 0; mode < 2; ++mode)
            for (size_t kb : {2048, 1024, 512}) {
                if (!do_io && (mode || kb != 2048)) continue;
                const size_t s_elems = kb * 1024 / 16;
                float best = 1e30f;
                Ctl h;
                for (int rep = 0; rep < 3; ++rep) {
                    hipMemset(ctl, 0, sizeof(Ctl));
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(team_kernel, dim3(256), dim3(256), 90 * 1024, 0, ctl, buf, s_elems, iters, mode, big_bytes / 16, do_io);
                    hipEventRecord(e1);
                    if (hipEventSynchronize(e1) != hipSuccess) {
                        printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
                        return 1;
                    }
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
                const double us = best * 1e3 / iters;
                printf("%-10s %-22s S = %4zu KB per XCD: %7.2f us per (write, barrier, read, barrier)", do_io ? "write+read" : "barriers",
                       mode ? "streamed (3 GB ring)" : "reused buffer (L2)", kb, us);
                if (do_io) printf("  = %5.2f TB/s chip-wide (2 S x 8 / t)", 2.0 * kb * 1024 * 8 / us * 1e-6);
                printf("  teams:");
                for (int x = 0; x < 8; ++x) printf(" %u", h.team_count[x]);
                printf("  stale values: %llu\n", h.stale);
            }
    return 0;
}
