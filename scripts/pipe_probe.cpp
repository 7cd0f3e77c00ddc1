// Do two hardware queues of one process run side by side, or do they take turns?  (development probe, round 4)
// queue_probe.cpp showed that CU-masked streams each get a hardware queue; the frame sweeps then showed that distinct
// queues are not enough (1 frame x 3 in flight: 76 or 35 frames/s depending on how many OTHER queues the process had made).
// Hypothesis: the command processor has 4 pipes, a queue is bound to pipe (creation index mod 4), and two busy queues on one
// pipe are served one after the other in long turns.  Test: a CHAIN of dependent short kernels (what a frame is) on stream 0
// and on stream j, for every j; side by side = T, taking turns = 2T.
//   hipcc -O2 --offload-arch=gfx950 scripts/pipe_probe.cpp -o scripts/pipe_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void spin(unsigned long long ticks) {
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
}
__global__ void tiny(int* p) { if (p) p[0] = 0; }

static const int CHAIN = 100;                      // kernels per chain
static const unsigned long long TICKS = 1000;      // 10 us each (100 MHz)

static hipGraphExec_t chain_graph(hipStream_t s, int wgs) {
    hipGraph_t g;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < CHAIN; ++k) hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), 0, s, TICKS);
    CK(hipStreamEndCapture(s, &g));
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    return ge;
}

static double run_us(const std::vector<hipGraphExec_t>& ge, const std::vector<hipStream_t>& on) {
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        for (auto s : on) CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < ge.size(); ++i) CK(hipGraphLaunch(ge[i], on[i]));
        for (auto s : on) CK(hipStreamSynchronize(s));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < best) best = us;
    }
    return best;
}

int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 12;
    int wgs = argc > 2 ? atoi(argv[2]) : 256;
    int plain_first = argc > 3 ? atoi(argv[3]) : 0;  // plain streams created (and used) before the CU-masked ones
    CK(hipSetDevice(0));
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, nullptr);
    CK(hipDeviceSynchronize());
    std::vector<hipStream_t> plain(plain_first);
    for (auto& x : plain) { CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking)); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, x, nullptr); CK(hipStreamSynchronize(x)); }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xffffffffu);
    std::vector<hipStream_t> m(n);
    std::vector<hipGraphExec_t> g(n);
    for (int i = 0; i < n; ++i) {
        CK(hipExtStreamCreateWithCUMask(&m[i], (uint32_t)mask.size(), mask.data()));
        g[i] = chain_graph(m[i], wgs);
        CK(hipGraphLaunch(g[i], m[i]));
        CK(hipStreamSynchronize(m[i]));
    }
    double one = run_us({g[0]}, {m[0]});
    printf("%d CU-masked streams after %d plain ones; chain = %d kernels x %d workgroups x 10 us; one chain alone: %.0f us\n", n, plain_first, CHAIN, wgs, one);
    printf("chain on stream 0 + chain on stream j (relative to one chain):\n ");
    for (int j = 1; j < n; ++j) printf(" j=%d:%.2f", j, run_us({g[0], g[j]}, {m[0], m[j]}) / one);
    printf("\nk chains on streams 0..k-1:\n ");
    for (int k = 2; k <= n && k <= 8; ++k) {
        std::vector<hipGraphExec_t> ge(g.begin(), g.begin() + k);
        std::vector<hipStream_t> on(m.begin(), m.begin() + k);
        printf(" k=%d:%.2f", k, run_us(ge, on) / one);
    }
    if (n >= 8) {
        printf("\nchains on {0,1,2,3}: %.2f   {0,1,2,4}: %.2f   {0,4}: %.2f   {1,5}: %.2f   {0,2,5,7}: %.2f   {4,5,6,7}: %.2f\n",
               run_us({g[0], g[1], g[2], g[3]}, {m[0], m[1], m[2], m[3]}) / one, run_us({g[0], g[1], g[2], g[4]}, {m[0], m[1], m[2], m[4]}) / one,
               run_us({g[0], g[4]}, {m[0], m[4]}) / one, run_us({g[1], g[5]}, {m[1], m[5]}) / one,
               run_us({g[0], g[2], g[5], g[7]}, {m[0], m[2], m[5], m[7]}) / one, run_us({g[4], g[5], g[6], g[7]}, {m[4], m[5], m[6], m[7]}) / one);
    }
    // cross-stream dependencies inside the chains (a frame's fork / join): two "frames", each = main chain + side chain joined by events
    if (n >= 8) {
        hipEvent_t ef[2], ej[2];
        for (int k = 0; k < 2; ++k) { CK(hipEventCreateWithFlags(&ef[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej[k], hipEventDisableTiming)); }
        auto frames = [&](int a0, int b0, int a1, int b1) {
            int A[2] = {a0, a1}, B[2] = {b0, b1};
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipDeviceSynchronize());
                auto t0 = std::chrono::steady_clock::now();
                for (int step = 0; step < 4; ++step)
                    for (int f = 0; f < 2; ++f) {
                        CK(hipEventRecord(ef[f], m[A[f]]));
                        CK(hipStreamWaitEvent(m[B[f]], ef[f], 0));
                        CK(hipGraphLaunch(g[B[f]], m[B[f]]));
                        CK(hipEventRecord(ej[f], m[B[f]]));
                        CK(hipGraphLaunch(g[A[f]], m[A[f]]));
                        CK(hipStreamWaitEvent(m[A[f]], ej[f], 0));
                        CK(hipGraphLaunch(g[A[f]], m[A[f]]));
                    }
                CK(hipDeviceSynchronize());
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (us < best) best = us;
            }
            return best / one;
        };
        printf("2 frames x 4 steps of (fork: side chain || main chain; join; main chain)  [ideal 8.0]:\n");
        printf("  main/side on (0,1),(2,3): %.2f   (0,2),(1,3): %.2f   (0,4),(1,5): %.2f   (0,1),(4,5): %.2f   (0,5),(2,7): %.2f\n",
               frames(0, 1, 2, 3), frames(0, 2, 1, 3), frames(0, 4, 1, 5), frames(0, 1, 4, 5), frames(0, 5, 2, 7));
    }
    return 0;
}
