// Stand-alone reproduction attempt for the round-3 host crash inside hipGraphLaunch -- diagnosis
// tool, not product code.   hipcc --offload-arch=gfx950 -O2 -pthread -o tools/probes/capture_probe tools/probes/capture_probe.hip
//
// The engine's capture topology in miniature, many times over with pseudo-random shapes:
//   * an origin stream carrying a chain of small kernels ("the step"),
//   * branch A forked at the head and joined mid-chain ("target pass"), branch B forked and joined mid-chain ("gaze"),
//   * branch T forked at the head and joined at the very end ("look-ahead trunk"),
//   * branch W forked MID-chain -- optionally from a SECOND HOST THREAD, as autograd's device thread does -- carrying one
//     kernel with a 3-KB by-value argument ("grouped weight gradients") and joined at the very end,
//   * optional round-3 habits: a join that also waits on a stream that took no part in this capture (event recorded on a
//     non-capturing stream, waited on by the capturing one), a nested fork (W forked from branch A instead of the origin),
//   * several graphs alive at once over the same streams, replayed interleaved, older ones destroyed in between.
// The branch accumulators are checked at the end (a dropped node would show; an omitted edge cannot: the adds commute).
//   [NO_DESTROY=1] capture_probe [rounds=200] [mode bits: 1 second-thread fork, 2 foreign-stream join, 4 nested fork] [seed]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

static const char* g_where = "start";
static int g_round = -1;
static void on_segv(int sig) {
    char msg[160];
    int n = snprintf(msg, sizeof msg, "\n==== signal %d in round %d during: %s ====\n", sig, g_round, g_where);
    write(2, msg, n);
    void* fr[64];
    int d = backtrace(fr, 64);
    backtrace_symbols_fd(fr, d, 2);
    _exit(139);
}
#define AT(s) (g_where = (s))

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

struct Big {
    float v[768];  // 3 KB of kernel arguments by value (rf_wgrad_grouped passes its 48-entry table this way)
};

__global__ void add_one(float* x, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += 1.0f;
}
__global__ void add_big(float* x, int n, Big b) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += b.v[i % 768];
}
__global__ void add_from(float* x, const float* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += y[i];
}

static unsigned rng_state = 12345;
static unsigned rnd(unsigned m) {
    rng_state = rng_state * 1664525u + 1013904223u;
    return (rng_state >> 8) % m;
}

static void fork(hipStream_t from, hipStream_t to) {
    hipEvent_t e;
    CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventRecord(e, from));
    CK(hipStreamWaitEvent(to, e, 0));
    CK(hipEventDestroy(e));  // torch's Stream.wait_stream does exactly this: the event dies right after the wait
}

struct Built {
    hipGraphExec_t exec;
    float inc_a, inc_t;  // what one replay adds to the A / T accumulators (B: 1, W: 0.5)
};

int main(int argc, char** argv) {
    signal(SIGSEGV, on_segv);
    int rounds = argc > 1 ? atoi(argv[1]) : 200;
    int mode = argc > 2 ? atoi(argv[2]) : 1;
    if (argc > 3) rng_state = (unsigned)atoi(argv[3]);
    const int n = 1 << 14;
    hipStream_t origin, sa, sb, st, sw, foreign;
    CK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sw, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&foreign, hipStreamNonBlocking));
    float *x, *xa, *xb, *xt, *xw;
    for (float** p : {&x, &xa, &xb, &xt, &xw}) {
        CK(hipMalloc(p, n * sizeof(float)));
        CK(hipMemset(*p, 0, n * sizeof(float)));
    }
    Big big;
    for (int i = 0; i < 768; ++i) big.v[i] = 0.5f;
    std::vector<Built> alive;
    double want_a = 0, want_b = 0, want_t = 0, want_w = 0;
    dim3 g((n + 255) / 256), b(256);
    for (int r = 0; r < rounds; ++r) {
        int head = 2 + rnd(6), mid = 3 + rnd(20), tail = 2 + rnd(20), la = 1 + rnd(8);
        hipGraph_t graph;
        g_round = r;
        AT("capture");
        CK(hipStreamBeginCapture(origin, hipStreamCaptureModeGlobal));
        fork(origin, st);
        for (int i = 0; i < la; ++i) add_one<<<g, b, 0, st>>>(xt, n);
        fork(origin, sa);
        for (int i = 0; i < head; ++i) add_one<<<g, b, 0, sa>>>(xa, n);
        for (int i = 0; i < head; ++i) add_one<<<g, b, 0, origin>>>(x, n);
        fork(origin, sb);
        add_one<<<g, b, 0, sb>>>(xb, n);
        for (int i = 0; i < mid; ++i) add_one<<<g, b, 0, origin>>>(x, n);
        fork(sb, origin);  // join B
        add_from<<<g, b, 0, origin>>>(x, xb, n);
        hipStream_t wfrom = (mode & 4) ? sa : origin;
        auto side = [&]() {
            fork(wfrom, sw);
            add_big<<<g, b, 0, sw>>>(xw, n, big);
        };
        if (mode & 1) {
            std::thread t(side);  // autograd's device thread makes this fork while the capturing thread waits for it
            t.join();
        } else {
            side();
        }
        fork(sa, origin);  // join A
        add_from<<<g, b, 0, origin>>>(x, xa, n);
        for (int i = 0; i < tail; ++i) add_one<<<g, b, 0, origin>>>(x, n);
        // final join: every side stream again (round-3 habit: all of them, used or not)
        fork(sa, origin);
        fork(sb, origin);
        fork(sw, origin);
        if (mode & 2) fork(foreign, origin);
        fork(st, origin);
        AT("hipStreamEndCapture");
        hipError_t e = hipStreamEndCapture(origin, &graph);
        if (e != hipSuccess) {
            fprintf(stderr, "round %d: hipStreamEndCapture -> %s\n", r, hipGetErrorString(e));
            return 3;
        }
        Built bt;
        bt.inc_a = (float)head;
        bt.inc_t = (float)la;
        AT("hipGraphInstantiate");
        CK(hipGraphInstantiate(&bt.exec, graph, nullptr, nullptr, 0));
        AT("hipGraphDestroy");
        CK(hipGraphDestroy(graph));  // torch destroys the hipGraph_t right after instantiation
        alive.push_back(bt);
        // replay a few of the graphs alive (the engine alternates between decision variants), newest first
        int reps = 1 + rnd(3);
        for (int k = 0; k < reps; ++k) {
            Built& pick = alive[alive.size() - 1 - rnd((unsigned)std::min<size_t>(alive.size(), 4))];
            AT("hipGraphLaunch");
            CK(hipGraphLaunch(pick.exec, origin));
            want_a += pick.inc_a, want_t += pick.inc_t, want_b += 1, want_w += 0.5;
        }
        if (!getenv("NO_DESTROY") && alive.size() > 6 && rnd(2)) {  // an old engine is garbage-collected
            size_t k = rnd((unsigned)alive.size() - 1);
            CK(hipStreamSynchronize(origin));
            AT("hipGraphExecDestroy");
            CK(hipGraphExecDestroy(alive[k].exec));
            alive.erase(alive.begin() + k);
        }
        if ((r & 31) == 31) {
            CK(hipStreamSynchronize(origin));
            printf("round %d ok (%zu graphs alive)\n", r + 1, alive.size());
            fflush(stdout);
        }
    }
    CK(hipDeviceSynchronize());
    float ha, hb, ht, hw;
    CK(hipMemcpy(&ha, xa + 7, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hb, xb + 7, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&ht, xt + 7, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hw, xw + 7, 4, hipMemcpyDeviceToHost));
    if (ha != (float)want_a || hb != (float)want_b || ht != (float)want_t || hw != (float)want_w) {
        fprintf(stderr, "accumulators: A %g (want %g) B %g (%g) T %g (%g) W %g (%g)\n", ha, want_a, hb, want_b, ht, want_t, hw, want_w);
        return 4;
    }
    printf("capture_probe: %d rounds, mode %d: no crash, no capture error\n", rounds, mode);
    return 0;
}
