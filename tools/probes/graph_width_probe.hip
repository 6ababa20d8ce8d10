// How wide may a captured graph be before hipGraphLaunch of the HIP runtime PyTorch bundles (libamdhip64 7.0.2) reads past
// its exec's parallel-stream pool?  (diagnosis tool, not product code.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/graph_width_probe tools/probes/graph_width_probe.hip
//   graph_width_probe WIDTH [rounds=200] [destroy: 0 never, 1 random older exec each round] [launch on: 0 null stream, 1 a created stream]
//                     [pad: streams created (and kept) per destroyed exec, -1 = WIDTH + 1] [which: 0 launch any of the newest three, 1 newest only, 2 older only]
// WIDTH parallel branches (forked at the head of the capture, joined at its end) next to the origin chain.
#include <hip/hip_runtime.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

__global__ void add_one(float* x, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += 1.0f;
}

static int g_round = -1, g_newest = -1;
static const char* g_where = "start";
static void on_segv(int sig) {
    char msg[160];
    int n = snprintf(msg, sizeof msg, "signal %d in round %d during %s (launching the %s exec)\n", sig, g_round, g_where, g_newest == 1 ? "newest" : g_newest == 0 ? "an older" : "-");
    write(1, msg, n);
    _exit(139);
}
static unsigned rng_state = 777;
static unsigned rnd(unsigned m) {
    rng_state = rng_state * 1664525u + 1013904223u;
    return (rng_state >> 8) % m;
}
static void edge(hipStream_t from, hipStream_t to) {
    hipEvent_t e;
    CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventRecord(e, from));
    CK(hipStreamWaitEvent(to, e, 0));
    CK(hipEventDestroy(e));
}

int main(int argc, char** argv) {
    signal(SIGSEGV, on_segv);
    int width = argc > 1 ? atoi(argv[1]) : 4, rounds = argc > 2 ? atoi(argv[2]) : 200;
    int destroy = argc > 3 ? atoi(argv[3]) : 1, created = argc > 4 ? atoi(argv[4]) : 0;
    int pad = argc > 5 ? atoi(argv[5]) : 0, which = argc > 6 ? atoi(argv[6]) : 0;
    if (pad < 0) pad = width + 1;
    std::vector<hipStream_t> padding;
    const int n = 1 << 12;
    hipStream_t origin, launch = nullptr;
    CK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));
    if (created) CK(hipStreamCreateWithFlags(&launch, hipStreamNonBlocking));
    std::vector<hipStream_t> side(width);
    std::vector<float*> buf(width + 1);
    for (auto& s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto& p : buf) {
        CK(hipMalloc(&p, n * sizeof(float)));
        CK(hipMemset(p, 0, n * sizeof(float)));
    }
    std::vector<hipGraphExec_t> alive;
    dim3 g(n / 256), b(256);
    for (int r = 0; r < rounds; ++r) {
        g_round = r;
        g_where = "capture";
        hipGraph_t graph;
        CK(hipStreamBeginCapture(origin, hipStreamCaptureModeGlobal));
        for (int k = 0; k < width; ++k) {
            edge(origin, side[k]);
            for (int i = 0; i < 3; ++i) add_one<<<g, b, 0, side[k]>>>(buf[k], n);
        }
        for (int i = 0; i < 6; ++i) add_one<<<g, b, 0, origin>>>(buf[width], n);
        for (int k = 0; k < width; ++k) edge(side[k], origin);
        CK(hipStreamEndCapture(origin, &graph));
        hipGraphExec_t ex;
        g_where = "hipGraphInstantiate";
        CK(hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0));
        CK(hipGraphDestroy(graph));
        alive.push_back(ex);
        g_where = "hipGraphLaunch";
        for (int k = 0; k < 2; ++k) {
            size_t back = rnd((unsigned)(alive.size() < 3 ? alive.size() : 3));
            if (which == 1) back = 0;
            if (which == 2 && alive.size() > 1) back = 1 + rnd((unsigned)(alive.size() < 3 ? alive.size() - 1 : 2));
            g_newest = back == 0;
            CK(hipGraphLaunch(alive[alive.size() - 1 - back], launch));
        }
        g_newest = -1;
        if (destroy && alive.size() > 3) {
            g_where = "hipGraphExecDestroy";
            int cnt = 1 + rnd(3);  // uneven: one to three older execs go at once
            CK(hipStreamSynchronize(launch));
            for (int c = 0; c < cnt && alive.size() > 2; ++c) {
                size_t k = rnd((unsigned)alive.size() - 1);
                CK(hipGraphExecDestroy(alive[k]));
                alive.erase(alive.begin() + k);
                for (int q = 0; q < pad; ++q) {  // fill the holes the exec's own streams left in the runtime's queue bookkeeping
                    hipStream_t d;
                    CK(hipStreamCreateWithFlags(&d, hipStreamNonBlocking));
                    padding.push_back(d);
                }
            }
        }
    }
    CK(hipDeviceSynchronize());
    printf("ok: width %d, %d rounds, destroy %d, launch stream %s, pad %d (%zu streams), which %d\n", width, rounds, destroy, created ? "created" : "null", pad, padding.size(), which);
    return 0;
}
