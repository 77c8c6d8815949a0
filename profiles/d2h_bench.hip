// Device-to-host ceiling of the box: how fast 274 MB (one C4 output volume) leave the GPU, by method.
// Build: hipcc --offload-arch=gfx950 -O3 -o scratch/d2h_bench scratch/d2h_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

__global__ void k_copy_nt(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float4 v = src[i];
        __builtin_nontemporal_store(v.x, &dst[i].x);
        __builtin_nontemporal_store(v.y, &dst[i].y);
        __builtin_nontemporal_store(v.z, &dst[i].z);
        __builtin_nontemporal_store(v.w, &dst[i].w);
    }
}

// something that keeps the CUs and HBM busy on another stream
__global__ void k_busy(float *p, size_t n, int iters)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int it = 0; it < iters; it++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = p[i] * 1.0001f + 1.f;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t bytes = (size_t)274 << 20;
    char *d; CK(hipMalloc((void **)&d, bytes));
    CK(hipMemset(d, 1, bytes));
    char *h; CK(hipHostMalloc((void **)&h, bytes, hipHostMallocDefault));
    char *hn; CK(hipHostMalloc((void **)&hn, bytes, hipHostMallocNonCoherent));
    float *busy; const size_t nbusy = (size_t)256 << 20; CK(hipMalloc((void **)&busy, nbusy * 4));
    CK(hipMemset(busy, 0, nbusy * 4));
    hipStream_t st[8];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipStream_t sb; CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    char *hd = nullptr, *hnd = nullptr;
    CK(hipHostGetDevicePointer((void **)&hd, h, 0));
    CK(hipHostGetDevicePointer((void **)&hnd, hn, 0));
    for (int busy_on = 0; busy_on < 2; busy_on++) {
        printf("---- %s ----\n", busy_on ? "beside a kernel that keeps HBM busy" : "idle GPU");
        auto start_busy = [&]() { if (busy_on) hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, sb, busy, nbusy, 40); };
        auto end_busy = [&]() { CK(hipStreamSynchronize(sb)); };
        // 1. hipMemcpyAsync split over ns streams
        for (int ns : {1, 2, 4, 8}) {
            for (int rep = 0; rep < 3; rep++) {
                CK(hipDeviceSynchronize());
                start_busy();
                const double t0 = now();
                const size_t part = bytes / ns;
                for (int i = 0; i < ns; i++) CK(hipMemcpyAsync(h + i * part, d + i * part, part, hipMemcpyDeviceToHost, st[i]));
                for (int i = 0; i < ns; i++) CK(hipStreamSynchronize(st[i]));
                const double dt = now() - t0;
                end_busy();
                if (rep == 2) printf("hipMemcpyAsync, %d stream(s): %.2f ms  %.1f GB/s\n", ns, dt * 1e3, bytes / dt / 1e9);
            }
        }
        // 2. many small copies on one stream (what a batch of rays looks like: 64 pieces of 4.3 MB)
        for (int ns : {1, 2}) {
            for (int rep = 0; rep < 3; rep++) {
                CK(hipDeviceSynchronize());
                start_busy();
                const double t0 = now();
                const int np = 64; const size_t part = bytes / np;
                for (int i = 0; i < np; i++) CK(hipMemcpyAsync(h + i * part, d + i * part, part, hipMemcpyDeviceToHost, st[i % ns]));
                for (int i = 0; i < ns; i++) CK(hipStreamSynchronize(st[i]));
                const double dt = now() - t0;
                end_busy();
                if (rep == 2) printf("64 copies of %.1f MB over %d stream(s): %.2f ms  %.1f GB/s\n", part / 1048576., ns, dt * 1e3, bytes / dt / 1e9);
            }
        }
        // 3. a kernel storing into mapped host memory
        for (int coh = 0; coh < 2; coh++)
            for (int nt = 0; nt < 2; nt++)
                for (int nwg : {4, 8, 16, 32, 64, 256}) {
                    for (int rep = 0; rep < 3; rep++) {
                        CK(hipDeviceSynchronize());
                        start_busy();
                        const double t0 = now();
                        if (nt) hipLaunchKernelGGL(k_copy_nt, dim3(nwg), dim3(256), 0, st[0], (const float4 *)d, (float4 *)(coh ? hd : hnd), bytes / 16);
                        else hipLaunchKernelGGL(k_copy, dim3(nwg), dim3(256), 0, st[0], (const float4 *)d, (float4 *)(coh ? hd : hnd), bytes / 16);
                        CK(hipStreamSynchronize(st[0]));
                        const double dt = now() - t0;
                        end_busy();
                        if (rep == 2) printf("copy kernel%s, %s host memory, %3d workgroups: %.2f ms  %.1f GB/s\n", nt ? " (nontemporal)" : "",
                                             coh ? "coherent" : "non-coherent", nwg, dt * 1e3, bytes / dt / 1e9);
                    }
                }
    }
    // sanity
    for (size_t i = 0; i < bytes; i += 4097) if (h[i] != 1 || hn[i] != 1) { printf("MISMATCH at %zu\n", i); return 1; }
    printf("contents ok\n");
    return 0;
}
