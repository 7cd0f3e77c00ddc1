// Which HIP streams share a hardware queue?  (development probe, round 4; DESIGN.md "launches in flight")
// Two spin kernels (one workgroup each, SPIN_US long) on streams i and j: concurrent = different hardware queues,
// back to back = the same queue.  Prints the alias classes for plain streams in creation / first-use order, for
// priority streams, for CU-masked streams, and what the parallel branches of captured graphs do when two graph
// executables are in flight.
//   hipcc -O2 --offload-arch=gfx950 scripts/queue_probe.cpp -o scripts/queue_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void spin(unsigned long long ticks, int* sink) {
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (sink) sink[0] = 1;
}
__global__ void tiny(int* p) { if (p) p[0] = 0; }

static const unsigned long long SPIN_TICKS = 30000;  // 100 MHz -> 300 us

static double pair_us(hipStream_t a, hipStream_t b) {
    CK(hipStreamSynchronize(a));
    CK(hipStreamSynchronize(b));
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, SPIN_TICKS, nullptr);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, SPIN_TICKS, nullptr);
    CK(hipStreamSynchronize(a));
    CK(hipStreamSynchronize(b));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

static void classes(const char* title, std::vector<hipStream_t>& s) {
    int n = (int)s.size();
    std::vector<int> cls(n, -1);
    int nc = 0;
    printf("%s: %d streams\n", title, n);
    for (int i = 0; i < n; ++i) {
        if (cls[i] >= 0) continue;
        cls[i] = nc;
        for (int j = i + 1; j < n; ++j) {
            if (cls[j] >= 0) continue;
            double us = 1e30;
            for (int r = 0; r < 2; ++r) { double u = pair_us(s[i], s[j]); if (u < us) us = u; }
            if (us > 1.6 * 300.0) cls[j] = nc;  // serialised: same queue
        }
        ++nc;
    }
    printf("  queue class per stream:");
    for (int i = 0; i < n; ++i) printf(" %d", cls[i]);
    printf("   (%d distinct)\n", nc);
    fflush(stdout);
}

static hipGraphExec_t capture_forked(hipStream_t s0, hipStream_t s1, int branches, int nodes_per_branch) {
    hipEvent_t ef, ej;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    hipGraph_t g;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s0, nullptr);
    if (branches == 2) {
        CK(hipEventRecord(ef, s0));
        CK(hipStreamWaitEvent(s1, ef, 0));
        for (int k = 0; k < nodes_per_branch; ++k) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, SPIN_TICKS / nodes_per_branch, nullptr);
        CK(hipEventRecord(ej, s1));
    }
    for (int k = 0; k < nodes_per_branch; ++k) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, SPIN_TICKS / nodes_per_branch, nullptr);
    if (branches == 2) CK(hipStreamWaitEvent(s0, ej, 0));
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s0, nullptr);
    CK(hipStreamEndCapture(s0, &g));
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    return ge;
}

static double graphs_us(std::vector<hipGraphExec_t>& ge, std::vector<hipStream_t>& on, int reps) {
    for (auto s : on) CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r)
        for (size_t i = 0; i < ge.size(); ++i) CK(hipGraphLaunch(ge[i], on[i]));
    for (auto s : on) CK(hipStreamSynchronize(s));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main(int argc, char** argv) {
    int nplain = argc > 1 ? atoi(argv[1]) : 10;
    CK(hipSetDevice(0));
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, nullptr);  // the null stream takes its queue first, as in a torch process
    CK(hipDeviceSynchronize());
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("priority range: least %d greatest %d\n", lo, hi);

    // 1. plain streams, first use in creation order
    std::vector<hipStream_t> s(nplain);
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (auto& x : s) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, x, nullptr); CK(hipStreamSynchronize(x)); }
    classes("plain, first use in creation order", s);
    {   std::vector<hipStream_t> withnull = {nullptr, s[0], s[1], s[2], s[3]};
        classes("null stream + first four", withnull); }

    // 2. plain streams created first, first USE in reverse order: is the queue bound at creation or at first use?
    std::vector<hipStream_t> r(6);
    for (auto& x : r) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (int i = 5; i >= 0; --i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, r[i], nullptr); CK(hipStreamSynchronize(r[i])); }
    classes("plain, first use in REVERSE order", r);
    {   std::vector<hipStream_t> mix = {s[0], s[1], s[2], s[3], r[0], r[1], r[2], r[3], r[4], r[5]};
        classes("first batch [0:4] + second batch", mix); }

    // 3. priority streams
    std::vector<hipStream_t> p(4);
    for (auto& x : p) CK(hipStreamCreateWithPriority(&x, hipStreamNonBlocking, hi));
    for (auto& x : p) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, x, nullptr); CK(hipStreamSynchronize(x)); }
    {   std::vector<hipStream_t> mix = {s[0], s[1], s[2], s[3], p[0], p[1], p[2], p[3]};
        classes("plain [0:4] + four high-priority", mix); }

    // 4. CU-masked streams (full mask): a queue of their own?
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0xffffffffu);
    std::vector<hipStream_t> m(6);
    for (auto& x : m) CK(hipExtStreamCreateWithCUMask(&x, (uint32_t)mask.size(), mask.data()));
    for (auto& x : m) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, x, nullptr); CK(hipStreamSynchronize(x)); }
    {   std::vector<hipStream_t> mix = {s[0], s[1], s[2], s[3], m[0], m[1], m[2], m[3], m[4], m[5]};
        classes("plain [0:4] + six CU-masked (all CUs)", mix); }

    // 5. graphs.  (a) two unforked graphs on two streams of different queues; (b) forked graphs, one in flight; (c) two forked
    //    graphs in flight on different queues; (d) the same on CU-masked streams
    auto report = [&](const char* what, std::vector<hipGraphExec_t> ge, std::vector<hipStream_t> on) {
        graphs_us(ge, on, 2);
        double us = graphs_us(ge, on, 6);
        printf("  %-70s %.0f us per round (one spin chain = 300 us)\n", what, us);
        fflush(stdout);
    };
    printf("graphs (each chain = 10 spin nodes of 30 us):\n");
    hipGraphExec_t g1a = capture_forked(s[0], s[2], 1, 10), g1b = capture_forked(s[1], s[3], 1, 10);
    report("2 unforked graphs on plain s0, s1", {g1a, g1b}, {s[0], s[1]});
    hipGraphExec_t g2a = capture_forked(s[0], s[2], 2, 10);
    report("1 forked graph on s0 (capture partner s2)", {g2a}, {s[0]});
    hipGraphExec_t g2b = capture_forked(s[1], s[3], 2, 10);
    report("2 forked graphs on s0, s1", {g2a, g2b}, {s[0], s[1]});
    hipGraphExec_t g2c = capture_forked(s[4], s[5], 2, 10);
    report("3 forked graphs on s0, s1, s4", {g2a, g2b, g2c}, {s[0], s[1], s[4]});
    report("2 forked graphs on CU-masked m0, m1", {g2a, g2b}, {m[0], m[1]});
    report("3 unforked graphs on s0, s1, s2", {g1a, g1b, capture_forked(s[2], s[3], 1, 10)}, {s[0], s[1], s[2]});
    report("3 unforked graphs on m0, m1, m2", {g1a, g1b, capture_forked(m[2], m[3], 1, 10)}, {m[0], m[1], m[2]});
    report("4 unforked graphs on m0..m3", {g1a, g1b, capture_forked(m[2], m[3], 1, 10), capture_forked(m[3], m[4], 1, 10)}, {m[0], m[1], m[2], m[3]});
    // (e) the fork written by hand: two unforked graphs per "frame" on two own streams, joined by events outside any graph
    {
        hipEvent_t ef[2], ej[2];
        for (int k = 0; k < 2; ++k) { CK(hipEventCreateWithFlags(&ef[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej[k], hipEventDisableTiming)); }
        hipGraphExec_t ga[2] = {capture_forked(m[0], m[1], 1, 10), capture_forked(m[2], m[3], 1, 10)};
        hipGraphExec_t gb[2] = {capture_forked(m[1], m[0], 1, 10), capture_forked(m[3], m[2], 1, 10)};
        auto round = [&](int frames) {
            for (int f = 0; f < frames; ++f) {
                hipStream_t a = m[2 * f], b = m[2 * f + 1];
                CK(hipEventRecord(ef[f], a));
                CK(hipStreamWaitEvent(b, ef[f], 0));
                CK(hipGraphLaunch(gb[f], b));
                CK(hipEventRecord(ej[f], b));
                CK(hipGraphLaunch(ga[f], a));
                CK(hipStreamWaitEvent(a, ej[f], 0));
            }
        };
        for (int frames = 1; frames <= 2; ++frames) {
            round(frames);
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 6; ++k) round(frames);
            CK(hipDeviceSynchronize());
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 6;
            printf("  hand-written fork (2 graphs + events on 2 own CU-masked streams) x %d frame(s) in flight: %.0f us per round\n", frames, us);
        }
    }
    return 0;
}
