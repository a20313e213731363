// sesrq C ABI, part 3 of 4: sesrq_forward_many and its submission pool (one persistent host thread per extra stream).  See include/sesrq.h.
#include <pthread.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include "sesrq_common.h"

using namespace sesrq;

namespace {
// Submission pool of sesrq_forward_many: one persistent host thread per extra stream.  A HIP kernel launch costs the calling thread
// ~3.5 us (round 4, 540p workloads: 10.8-11.2 us per frame of three launches from one thread, where the device needs ~11 us per frame
// when fed): the launches of DIFFERENT streams are independent, so each stream's frames are enqueued by a thread of its own, in order.
struct SubmitJob {
    const sesrq_net *net; const sesrq_frame_io *frames; int count, first, stride, in_dtype, N, H, W;
    void *ws; size_t ws_bytes; void *stream;
    int group;               // frames of this stream per launch sequence (1 = one sesrq_forward per frame)
    int next;                // the stream's next frame
    int rc = 0, bad = -1; std::string err;
    std::atomic<int> *selftest = nullptr;      // sesrq_submit_selftest: the job only counts itself (no net, no device)
};
// one launch sequence of the job: the next frame of its stream, or the next `group` frames as the images of one launch sequence (same
// kernels, N = g, pointer table).  false = nothing left (or an error: j.rc)
static bool job_step(SubmitJob &j) {
    const int k = j.next;
    if (k >= j.count || j.rc) return false;
    int g = 1;
    FrameTable ft;
    ft.n = 0;
    if (j.group > 1) {
        const sesrq_frame_io &f0 = j.frames[k];
        for (g = 0; g < j.group && k + g * j.stride < j.count; ++g) {
            const sesrq_frame_io &f = j.frames[k + g * j.stride];
            if (!f.in || (f.out_q != nullptr) != (f0.out_q != nullptr) || (f.out_f != nullptr) != (f0.out_f != nullptr)) break;
            // the images of one launch sequence are written concurrently: a shared output buffer would be a silent write race
            // (with one launch sequence per frame the same sharing is legal -- stream order)
            for (int e = 0; e < g; ++e)
                if ((f.out_q && f.out_q == ft.out_q[e]) || (f.out_f && (float *)f.out_f == ft.out_f[e])) {
                    j.rc = 1; j.bad = k + g * j.stride;
                    j.err = "frames " + std::to_string(k + e * j.stride) + " and " + std::to_string(k + g * j.stride) +
                            " would share a launch sequence (group = " + std::to_string(j.group) + ") and an output buffer";
                    return false;
                }
            ft.in[g] = f.in; ft.out_q[g] = f.out_q; ft.out_f[g] = (float *)f.out_f;
        }
        if (g < 1) g = 1;
        ft.n = g > 1 ? g : 0;
    }
    const sesrq_frame_io &f = j.frames[k];
    if (forward_impl(j.net, f.in, j.in_dtype, f.out_q, f.out_f, ft.n ? g : j.N, j.H, j.W, j.ws, j.ws_bytes, j.stream, nullptr, nullptr,
                     ft.n ? &ft : nullptr)) {
        j.rc = 1; j.bad = k; j.err = sesrq_last_error();
        return false;
    }
    j.next = k + g * j.stride;
    return true;
}
static int run_job(SubmitJob &j) {
    if (j.selftest) { j.selftest->fetch_add(1, std::memory_order_relaxed); return 0; }
    while (job_step(j)) {}
    return j.rc;
}

using clk = std::chrono::steady_clock;

class SubmitPool {
    // A worker spins on its job slot for spin_us after finishing a job (the bench hands over a block of frames every ~1 ms: a condition
    // variable's wake-up would add 20-50 us to every block), then sleeps on the condition variable until the next call.
    // SESRQ_SPIN_US = 0: no spinning at all (hosts where the ranks outnumber a quarter of the cores).
    const int spin_us = env_knob("SESRQ_SPIN_US", 2000, 0, 1000000);
    static constexpr int TAKE_BACK_MS = 200;      // a job no worker has picked up by then is run by the caller
    static constexpr int STALL_S = 60;            // a worker that holds a job for longer is reported as stalled
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::atomic<SubmitJob *> job{nullptr};      // published by the caller, TAKEN (exchanged to null) by the worker -- or taken back by the caller
        std::atomic<bool> done{true}, asleep{false};
        bool quit = false;
        int spin_us = 0;
    };
    std::vector<std::unique_ptr<Worker>> workers;
    std::mutex call_mu;      // one sesrq_forward_many at a time uses the pool (a second concurrent caller enqueues on its own thread)
    static void loop(Worker *w) {
        int dev = -1;
        for (;;) {
            SubmitJob *j = nullptr;
            const auto t0 = clk::now();
            // (a plain load while the slot is empty: an exchange per spin would keep the cache line bouncing between the cores)
            while (!(w->job.load(std::memory_order_acquire) && (j = w->job.exchange(nullptr, std::memory_order_acq_rel)))) {
                if (w->spin_us == 0 || clk::now() - t0 > std::chrono::microseconds(w->spin_us)) {
                    std::unique_lock<std::mutex> lk(w->mu);
                    w->asleep.store(true, std::memory_order_seq_cst);
                    w->cv.wait(lk, [&] { return w->job.load(std::memory_order_seq_cst) || w->quit; });
                    w->asleep.store(false);
                    if (w->quit) return;
                } else {
                    __builtin_ia32_pause();
                }
            }
            if (!j->selftest && dev != j->net->device) {      // a fresh thread starts on device 0
                const hipError_t e = hipSetDevice(j->net->device);
                if (e != hipSuccess) {
                    j->rc = 1; j->bad = j->next;
                    j->err = std::string("submission thread: hipSetDevice(") + std::to_string(j->net->device) + ") failed: " + hipGetErrorString(e);
                    dev = -1;
                } else {
                    dev = j->net->device;
                }
            }
            if (!j->rc) run_job(*j);
            w->done.store(true, std::memory_order_release);
        }
    }
    void grow(size_t n) {
        while (workers.size() < n) {
            workers.emplace_back(new Worker());
            Worker *w = workers.back().get();
            w->spin_us = spin_us;
            w->th = std::thread(loop, w);
        }
    }
public:
    const pid_t owner = getpid();
    ~SubmitPool() {
        for (auto &w : workers) {
            { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
        }
    }
    // the threads exist (and spin) before the first batch that needs them: creating a thread costs more than the batch
    void ensure(size_t n) {
        std::unique_lock<std::mutex> call(call_mu, std::try_to_lock);
        if (call.owns_lock()) grow(n);
    }
    // jobs[0] runs on the caller's thread, jobs[1..] on the workers; false = the pool is busy (caller falls back to one thread).
    // fatal (set on a stalled worker): the jobs have been moved to a leaked vector -- the worker still holds a pointer into it -- and the
    // caller must return the error without looking at `jobs` again
    bool run(std::vector<SubmitJob> &jobs, std::string *fatal = nullptr) {
        std::unique_lock<std::mutex> call(call_mu, std::try_to_lock);
        if (!call.owns_lock()) return false;
        grow(jobs.size() - 1);
        for (size_t i = 1; i < jobs.size(); ++i) {
            Worker *w = workers[i - 1].get();
            w->done.store(false, std::memory_order_relaxed);
            // seq_cst on purpose: "publish the job, then look whether the worker sleeps" against the worker's "say asleep, then look for a
            // job" is Dekker's pattern -- with a release store the load below may pass it (store buffer), both sides read the old value
            // and the worker sleeps on a pending job while this thread spins on `done`
            w->job.store(&jobs[i], std::memory_order_seq_cst);
            if (w->asleep.load(std::memory_order_seq_cst)) { std::lock_guard<std::mutex> lk(w->mu); w->cv.notify_all(); }
        }
        run_job(jobs[0]);
        // Bounded wait (round 5).  Spin briefly (the workers finish within microseconds of the caller's own job), then yield.  A job that
        // is still in its slot after TAKE_BACK_MS has no worker behind it (thread creation failed, the thread died): the caller takes it
        // back -- the exchange decides who owns it -- and runs it itself.  A job a worker HAS taken cannot be run twice: after STALL_S the
        // call returns an error instead of spinning forever.
        const auto t0 = clk::now();
        for (size_t i = 1; i < jobs.size(); ++i) {
            Worker *w = workers[i - 1].get();
            bool reclaimed = false;
            while (!w->done.load(std::memory_order_acquire)) {
                const auto waited = clk::now() - t0;
                if (waited < std::chrono::microseconds(std::max(spin_us, 50))) { __builtin_ia32_pause(); continue; }
                if (!reclaimed && waited > std::chrono::milliseconds(TAKE_BACK_MS)) {
                    SubmitJob *mine = w->job.exchange(nullptr, std::memory_order_acq_rel);
                    reclaimed = true;
                    if (mine) { run_job(*mine); w->done.store(true, std::memory_order_release); break; }
                }
                if (waited > std::chrono::seconds(STALL_S)) {
                    // the worker owns &jobs[i] and may still write to it: the vector must outlive this call -> moved (element addresses are
                    // kept by a vector move) into an object that is never freed.  `done` stays false: the worker gets no further job.
                    const std::string msg = "submission thread of stream " + std::to_string(i) + " stalled for " + std::to_string(STALL_S) + " s with frames in its hands";
                    if (fatal) *fatal = msg;
                    new std::vector<SubmitJob>(std::move(jobs));
                    jobs.clear();
                    return true;
                }
                std::this_thread::yield();
            }
        }
        return true;
    }
};

// The pool is per PROCESS.  fork() copies the Worker objects but not their threads: a child that used the parent's pool would publish jobs
// nobody takes (round 4: it spun on `done` forever) and its exit-time destructor would join threads that do not exist.  The child handler
// of pthread_atfork drops the pointer (the parent's objects are leaked in the child, never touched), the child's first call builds its own.
std::atomic<SubmitPool *> g_pool{nullptr};
std::mutex g_pool_mu;
void pool_atfork_child() {
    g_pool.store(nullptr, std::memory_order_relaxed);
    new (&g_pool_mu) std::mutex();      // the parent may have held it at the moment of the fork
}
void pool_shutdown() {
    SubmitPool *p = g_pool.exchange(nullptr);
    if (p && p->owner == getpid()) delete p;
}
// joins the threads when the library is unloaded or the process exits (a static object of the library rather than an atexit handler: the
// handler would dangle after a dlclose)
struct PoolReaper { ~PoolReaper() { pool_shutdown(); } } g_pool_reaper;
SubmitPool &submit_pool() {
    SubmitPool *p = g_pool.load(std::memory_order_acquire);
    if (!p) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        p = g_pool.load(std::memory_order_acquire);
        if (!p) {
            static std::once_flag once;
            std::call_once(once, [] { pthread_atfork(nullptr, nullptr, pool_atfork_child); });
            p = new SubmitPool();
            g_pool.store(p, std::memory_order_release);
        }
    }
    return *p;
}
}  // namespace

extern "C" {

int sesrq_forward_many(const sesrq_net *net, const sesrq_frame_io *frames, int count, int in_dtype, int N, int H, int W,
                       void *const *workspaces, size_t workspace_bytes, void *const *streams, int n_streams, int group) {
    if (!net || !frames || !workspaces || !streams) { set_error("sesrq_forward_many: null argument"); return 1; }
    if (count < 0 || n_streams < 1 || n_streams > 64) { set_error("sesrq_forward_many: count must be >= 0 and n_streams in 1..64"); return 1; }
    for (int s = 0; s < std::min(n_streams, count); ++s)
        if (!workspaces[s]) { set_error("sesrq_forward_many: null workspace"); return 1; }
    // Frames per launch sequence: the caller says so (ABI v4; v3 inferred it from the workspace size).  G > 1: up to G consecutive frames of a
    // stream share one launch sequence (pointer table in the kernel arguments, ConvArgs::ft): the launches' fixed cost is paid once per group.
    if (group < 1 || group > SESRQ_GROUP_MAX) { set_error("sesrq_forward_many: group must be in 1.." + std::to_string(SESRQ_GROUP_MAX)); return 1; }
    if (group > 1) {
        if (N != 1) { set_error("sesrq_forward_many: group > 1 needs single-image frames (N == 1)"); return 1; }
        if (!groupable(net)) { set_error("sesrq_forward_many: group > 1 needs the MFMA first- and last-layer kernels (this net / engine option runs them on dot4)"); return 1; }
        if (ws_layout(net, group, H, W).total > workspace_bytes) { set_error("sesrq_forward_many: workspace too small for this group (see sesrq_workspace_bytes(net, group, H, W))"); return 1; }
    }
    // SESRQ_SUBMIT_THREADS=0: everything from the calling thread; default: one thread per stream when a stream gets at least two
    // launch sequences (fewer: waking a thread costs more than the launches it takes over)
    static const int threads_knob = env_knob("SESRQ_SUBMIT_THREADS", 1, 0, 1);
    if (threads_knob && n_streams > 1) submit_pool().ensure((size_t)n_streams - 1);
    std::vector<SubmitJob> jobs;
    const bool pooled = threads_knob && n_streams > 1 && count >= 2 * n_streams * group;
    const int nj = std::min(n_streams, std::max(count, 1));
    for (int s = 0; s < nj; ++s)
        jobs.push_back(SubmitJob{net, frames, count, s, n_streams, in_dtype, N, H, W, workspaces[s], workspace_bytes, streams[s], group, s});
    std::string fatal;
    const bool ran = pooled && submit_pool().run(jobs, &fatal);
    if (!fatal.empty()) { set_error("sesrq_forward_many: " + fatal); return 1; }
    if (!ran) {      // one thread: the streams take turns, one launch sequence each
        for (bool any = true; any;) {
            any = false;
            for (auto &j : jobs) any |= job_step(j);
        }
    }
    int bad = -1;
    for (int j = 0; j < nj; ++j)
        if (jobs[j].rc && (bad < 0 || jobs[j].bad < jobs[bad].bad)) bad = j;
    if (bad >= 0) { set_error("sesrq_forward_many: frame " + std::to_string(jobs[bad].bad) + ": " + jobs[bad].err); return 1; }
    return 0;
}

int sesrq_submit_selftest(int n_streams, int rounds) {
    if (n_streams < 2 || n_streams > 64 || rounds < 1) { set_error("sesrq_submit_selftest: n_streams in 2..64, rounds >= 1"); return -1; }
    std::atomic<int> ran{0};
    for (int r = 0; r < rounds; ++r) {
        std::vector<SubmitJob> jobs((size_t)n_streams);
        for (auto &j : jobs) j.selftest = &ran;
        std::string fatal;
        if (!submit_pool().run(jobs, &fatal)) { set_error("sesrq_submit_selftest: the pool is busy"); return -1; }
        if (!fatal.empty()) { set_error("sesrq_submit_selftest: " + fatal); return -1; }
        for (auto &j : jobs)
            if (j.rc) { set_error("sesrq_submit_selftest: " + j.err); return -1; }
    }
    return ran.load();
}

}  // extern "C"
