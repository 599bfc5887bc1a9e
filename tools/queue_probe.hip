// tools/queue_probe.hip -- what delays a kernel dispatched on one hardware queue while a long, latency-bound kernel is resident on
// another?  (The decode pipeline's search kernel: 512 workgroups x 256 work-items x 63 KB of LDS, ~90 ms; r03 overlap traces
// showed small kernels of a second stream taking 75-110 ms beside it.)  A spin kernel stands for the search; victims of several
// sizes are dispatched on a second stream some milliseconds later and timed with events; variations: LDS of the spin kernel,
// its grid, what follows it in its own queue, how the streams were made.
//   hipcc --offload-arch=gfx950 -O2 tools/queue_probe.hip -o /tmp/queue_probe && /tmp/queue_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(long long cycles, int *sink)
{
    extern __shared__ int lds[];                      // [0]: "go on", decided by work-item 0 (every work-item leaves in the same round)
    const long long t0 = wall_clock64();
    int v = 0;
    for (;;) {
        if (threadIdx.x == 0) lds[0] = (wall_clock64() - t0 < cycles) ? 1 : 0;
        __syncthreads();
        const int go = lds[0];
        v += lds[1 + (threadIdx.x & 63)];
        __syncthreads();
        if (!go) break;
        lds[1 + (threadIdx.x & 63)] = v;
        __builtin_amdgcn_s_sleep(8);
    }
    if (v == 0x7fffffff) *sink = v;
}
__global__ void victim_kernel(int *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

static hipStream_t mk_stream(int how)
{
    hipStream_t s;
    if (how == 0) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else if (how == 1) { uint32_t m[8]; for (int i = 0; i < 8; ++i) m[i] = 0xffffffffu; CK(hipExtStreamCreateWithCUMask(&s, 8, m)); }
    else CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, how == 2 ? -1 : 0));
    return s;
}

int main()
{
    int *sink, *out;
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&out, (size_t)65536 * 256 * 4 + 4096));
    CK(hipFuncSetAttribute((const void *)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const long long ticks_per_ms = 100000;          // wall_clock64: 100 MHz
    struct Var { const char *name; int lds_kb, grid, follow, how_a, how_b; };
    const Var vars[] = {
        { "spin 512 WG x 63 KB, nothing after it, plain streams", 63, 512, 0, 0, 0 },
        { "spin 512 WG x 63 KB, event record after it, plain streams", 63, 512, 1, 0, 0 },
        { "spin 512 WG x 63 KB, a kernel queued after it, plain streams", 63, 512, 2, 0, 0 },
        { "spin 512 WG x 1 KB, nothing after it, plain streams", 1, 512, 0, 0, 0 },
        { "spin 512 WG x 1 KB, a kernel queued after it, plain streams", 1, 512, 2, 0, 0 },
        { "spin 256 WG x 63 KB, a kernel queued after it, plain streams", 63, 256, 2, 0, 0 },
        { "spin 512 WG x 63 KB, nothing after it, CU-mask streams", 63, 512, 0, 1, 1 },
        { "spin 512 WG x 63 KB, a kernel queued after it, CU-mask streams", 63, 512, 2, 1, 1 },
        { "spin 512 WG x 63 KB, a kernel queued after it, victim stream high priority", 63, 512, 2, 3, 2 },
        { "spin 512 WG x 40 KB (three fit), a kernel queued after it, plain streams", 40, 512, 2, 0, 0 },
    };
    for (const Var &v : vars) {
        hipStream_t a = mk_stream(v.how_a), b = mk_stream(v.how_b);
        hipEvent_t e0, e1, ea0, ea1, ef;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ea0)); CK(hipEventCreate(&ea1)); CK(hipEventCreate(&ef));
        printf("== %s\n", v.name);
        const int sizes[] = { 1, 8, 512, 65536 };            // victim workgroups of 256
        for (int sz : sizes) {
            for (int delay_ms : { 1, 10 }) {
                // warm both
                hipLaunchKernelGGL(victim_kernel, dim3(1), dim3(256), 0, b, out, 256);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(ea0, a));
                hipLaunchKernelGGL(spin_kernel, dim3(v.grid), dim3(256), v.lds_kb * 1024, a, 50 * ticks_per_ms, sink);
                CK(hipEventRecord(ea1, a));
                if (v.follow == 1) CK(hipEventRecord(ef, a));
                if (v.follow == 2) hipLaunchKernelGGL(victim_kernel, dim3(1), dim3(256), 0, a, out, 256);
                std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
                const auto h0 = std::chrono::steady_clock::now();
                CK(hipEventRecord(e0, b));
                hipLaunchKernelGGL(victim_kernel, dim3(sz), dim3(256), 0, b, out, sz * 256);
                CK(hipEventRecord(e1, b));
                CK(hipStreamSynchronize(b));
                const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
                CK(hipDeviceSynchronize());
                float dv = 0, da = 0;
                CK(hipEventElapsedTime(&dv, e0, e1)); CK(hipEventElapsedTime(&da, ea0, ea1));
                printf("   victim %6d WG dispatched %2d ms after the spin kernel: events %8.3f ms, host launch->done %8.3f ms (spin kernel %6.1f ms)\n",
                       sz, delay_ms, dv, host_ms, da);
            }
        }
        CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b));
    }
    return 0;
}
