// shard_replay.cpp — a batch of independent RGB-D frames sharded over the GPUs of one node by a C++ host
// (BASELINE config 4): ONE process, one host thread per visible device.  Each thread
//   * binds its device (kde_set_device) and owns its stream, its kde_jbf handle and its shard's buffers,
//   * takes the filter parameter block {window, sigmas, pre-smoothing, spatial table} from rank 0 through ONE
//     ncclBroadcast (RCCL over xGMI) and checks its own host-computed table against rank 0's,
//   * filters its contiguous block of ceil(N / G) frames with kde_jbf_process_batch; there is no data-path
//     collective (frames are independent units) and outputs stay on the owning GPU.
// The reference uploads to a single device (main.cpp:160-163); this is the multi-device host a maintainer would
// write on the same C ABI (include/kde_hip.h).
//
// usage: shard_replay [--frames N] [--width W] [--height H] [--window 11] [--steps K] [--warmup W] [--wakeup-ms 150]
//                     [--devices G] [--share-device G] [--frames-file F] [--verify] [--force-rccl-failure] [--rccl-timeout S]
//   Timing follows bench.py: an untimed wake-up load (an idle MI355X sits at its lowest clock level), W warm-up steps,
//   then K timed steps that all device threads start together; every step is K0 (kde_jbf_presmooth_batch) + K1
//   (kde_jbf_filter_batch) -- exactly what kde_jbf_process_batch launches -- bracketed by HIP events on the device's
//   stream, so the per-kernel times are reported per device as bench.py reports them.
//   --frames-file: raw frames written by `bench.py --dump-frames F` (all colour frames, then all depth frames), so that
//   the two hosts can be compared on identical input; without it the frames come from the built-in generator.
//   --verify: the same N frames are also filtered on device 0 alone, as ONE block and as TWO half blocks, and the
//             per-frame checksums of all three runs must be identical (partition independence, bit for bit).
//   --force-rccl-failure: behave as if ncclCommInitAll had failed (test of the fallback below).
//   --share-device G: G host threads that all use device 0, each with its own stream, kde_jbf handle and shard buffers --
//             the code path eight GPUs take, on a one-GPU box, and the test of kde_hip.h's threading contract ("handles are
//             independent and may be used from different threads").  RCCL wants one device per rank, so the parameter
//             block goes the "replicas only" way.
//   --rccl-timeout S (default 60): ncclCommInitAll runs in a helper thread; if it has not returned after S seconds the
//             run goes on without RCCL ("replicas only", flagged) and the process leaves through _Exit at the end.
// Fallback: if the RCCL communicators cannot be created, nothing is restarted -- every device thread forms the
// parameter block itself ("replicas only"), compares it with rank 0's copy in host memory, and the line says so with RCCL's error.
// prints one JSON line: per-device times and PCI addresses, aggregate Mpixels/s, checksum,
// "params_broadcast": "rccl ncclBroadcast" | "replicas only (...)".
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/kde_hip.h"

#define HIP_OK(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            std::exit(2);                                                                          \
        }                                                                                          \
    } while (0)
#define KDE_OK_OR_DIE(x)                                                                           \
    do {                                                                                           \
        int rc_ = (x);                                                                             \
        if (rc_ != KDE_OK) {                                                                       \
            std::fprintf(stderr, "%s: %s (%s:%d)\n", #x, kde_last_error_string(), __FILE__, __LINE__); \
            std::exit(2);                                                                          \
        }                                                                                          \
    } while (0)
#define NCCL_OK(x)                                                                                 \
    do {                                                                                           \
        ncclResult_t r_ = (x);                                                                     \
        if (r_ != ncclSuccess) {                                                                   \
            std::fprintf(stderr, "%s: %s (%s:%d)\n", #x, ncclGetErrorString(r_), __FILE__, __LINE__); \
            std::exit(2);                                                                          \
        }                                                                                          \
    } while (0)

namespace {

struct Options {
    int frames = 64, width = 640, height = 480, window = 11, steps = 20, warmup = 5, devices = 0, share_device = 0;
    float spatial_sigma = 3.0f, color_sigma = 7.65f, depth_sigma = 20.0f, wakeup_ms = 150.0f, rccl_timeout_s = 60.0f;
    bool verify = false, force_rccl_failure = false;
    std::string frames_file;
};

// all device threads enter the timed region together (and leave the wake-up together)
struct SpinBarrier {
    std::atomic<int> count{0}, generation{0};
    int parties = 1;
    void wait()
    {
        const int gen = generation.load();
        if (count.fetch_add(1) + 1 == parties) {
            count.store(0);
            generation.fetch_add(1);
        } else {
            while (generation.load() == gen) std::this_thread::yield();
        }
    }
};

// deterministic synthetic frame (seed = global frame index): a smooth ramp with rectangles, noise and holes.
// Not the Python generator of the tests -- this program only needs frames that differ and exercise every path.
uint32_t lcg(uint32_t& s) { return s = s * 1664525u + 1013904223u; }

void make_frame(int seed, int W, int H, std::vector<uint8_t>& bgr, std::vector<float>& depth)
{
    bgr.resize((size_t)W * H * 3);
    depth.resize((size_t)W * H);
    uint32_t s = 0x9E3779B9u * (uint32_t)(seed + 1);
    int rx[6], ry[6], rw[6], rh[6], rc[6];
    float rz[6];
    for (int k = 0; k < 6; k++) {
        rw[k] = W / 8 + (int)(lcg(s) % (unsigned)(W / 3));
        rh[k] = H / 8 + (int)(lcg(s) % (unsigned)(H / 3));
        rx[k] = (int)(lcg(s) % (unsigned)(W - rw[k]));
        ry[k] = (int)(lcg(s) % (unsigned)(H - rh[k]));
        rc[k] = (int)(lcg(s) & 0xffffff);
        rz[k] = 800.0f + (float)(lcg(s) % 3200u);
    }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float z = 3000.0f + 0.4f * x - 0.2f * y;
            int c = ((x * 255 / W) << 16) | ((y * 255 / H) << 8) | 0x40;
            for (int k = 0; k < 6; k++)
                if (x >= rx[k] && x < rx[k] + rw[k] && y >= ry[k] && y < ry[k] + rh[k]) {
                    z = rz[k] + 0.1f * (x - rx[k]);
                    c = rc[k];
                }
            const uint32_t r = lcg(s);
            z += ((float)(r & 0xff) - 127.5f) * 0.02f;
            if ((r >> 8) % 100u == 0) z = 0.0f;       // 1 % holes
            const size_t q = (size_t)y * W + x;
            depth[q] = z;
            for (int ch = 0; ch < 3; ch++) {
                int v = ((c >> (8 * ch)) & 0xff) + (int)((r >> (16 + 3 * ch)) & 7) - 4;
                bgr[q * 3 + ch] = (uint8_t)std::min(255, std::max(0, v));
            }
        }
}

// FNV-1a over the raw bytes of one frame: equal iff the frames are bit-identical (up to hash collisions)
uint64_t fnv(const float* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n * 4; i++) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

constexpr int kBlockLen = 8 + 31 * 31;     // params + the largest spatial table

void pack_params(const kde_jbf_params& p, const std::vector<float>& table, float* blk)
{
    std::memset(blk, 0, sizeof(float) * kBlockLen);
    blk[0] = (float)p.window_size; blk[1] = p.spatial_sigma; blk[2] = p.color_sigma; blk[3] = p.depth_sigma;
    blk[4] = (float)p.presmooth; blk[5] = (float)p.presmooth_kernel_size; blk[6] = p.presmooth_sigma_color;
    blk[7] = p.presmooth_sigma_spatial;
    std::memcpy(blk + 8, table.data(), table.size() * sizeof(float));
}

kde_jbf_params unpack_params(const float* blk)
{
    kde_jbf_params p;
    p.window_size = (int)blk[0]; p.spatial_sigma = blk[1]; p.color_sigma = blk[2]; p.depth_sigma = blk[3];
    p.presmooth = (int)blk[4]; p.presmooth_kernel_size = (int)blk[5]; p.presmooth_sigma_color = blk[6];
    p.presmooth_sigma_spatial = blk[7];
    return p;
}

struct ShardResult {
    double ms_per_step = 0.0;               // wall time of the K timed steps on this device / K (host clock around the sync)
    double k0_ms = 0.0, k1_ms = 0.0;        // mean launch time of the two kernels over the timed steps (HIP events)
    double k1_ms_median = 0.0, k1_ms_min = 0.0;
    int wakeup_steps = 0;
    std::vector<uint64_t> frame_hash;       // one per frame of the shard
    bool table_matches_rank0 = true;
    std::string pci_bus_id;
};

// filters frames [first, first + count) on `device`; if comm != nullptr the parameter block comes from rank 0's broadcast
ShardResult run_shard(const Options& o, int device, int rank, int first, int count, ncclComm_t comm, const kde_jbf_params& root_params,
                      SpinBarrier* barrier = nullptr, const std::vector<float>* root_block = nullptr)
{
    ShardResult res;
    KDE_OK_OR_DIE(kde_set_device(device));
    {
        char bus[64] = "";
        KDE_OK_OR_DIE(kde_device_pci_bus_id(bus, sizeof(bus)));
        res.pci_bus_id = bus;
    }
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    const size_t px = (size_t)o.width * o.height;

    // ---- parameter block: rank 0 decides, everyone receives the same bytes over RCCL ----
    float* blk_dev = nullptr;
    HIP_OK(hipMalloc(&blk_dev, sizeof(float) * kBlockLen));
    std::vector<float> blk(kBlockLen, 0.0f);
    if (rank == 0) {
        kde_jbf* probe = nullptr;
        KDE_OK_OR_DIE(kde_jbf_create(&probe, 8, 8, 1, &root_params));
        std::vector<float> table((size_t)root_params.window_size * root_params.window_size);
        KDE_OK_OR_DIE(kde_jbf_spatial_table(probe, table.data(), (int)table.size()));
        KDE_OK_OR_DIE(kde_jbf_destroy(probe));
        pack_params(root_params, table, blk.data());
        HIP_OK(hipMemcpyAsync(blk_dev, blk.data(), sizeof(float) * kBlockLen, hipMemcpyHostToDevice, stream));
    }
    if (comm) NCCL_OK(ncclBroadcast(blk_dev, blk_dev, kBlockLen, ncclFloat, 0, comm, stream));
    HIP_OK(hipMemcpyAsync(blk.data(), blk_dev, sizeof(float) * kBlockLen, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    const kde_jbf_params p = (comm || rank == 0) ? unpack_params(blk.data()) : root_params;

    kde_jbf* jbf = nullptr;
    KDE_OK_OR_DIE(kde_jbf_create(&jbf, o.width, o.height, std::max(count, 1), &p));
    {   // every rank provably filters with rank 0's table
        std::vector<float> mine((size_t)p.window_size * p.window_size);
        KDE_OK_OR_DIE(kde_jbf_spatial_table(jbf, mine.data(), (int)mine.size()));
        if (comm || rank == 0) res.table_matches_rank0 = std::memcmp(mine.data(), blk.data() + 8, mine.size() * sizeof(float)) == 0;
        // replicas only (no communicator): this rank formed the block itself; rank 0's copy is compared in host memory
        else if (root_block) res.table_matches_rank0 = std::memcmp(mine.data(), root_block->data() + 8, mine.size() * sizeof(float)) == 0;
    }
    if (count > 0) {
        // ---- this shard's frames, resident in this device's HBM ----
        float* depth_dev = nullptr;
        uint8_t* bgr_dev = nullptr;
        float* out_dev = nullptr;
        HIP_OK(hipMalloc(&depth_dev, px * count * sizeof(float)));
        HIP_OK(hipMalloc(&bgr_dev, px * count * 3));
        HIP_OK(hipMalloc(&out_dev, px * count * sizeof(float)));
        std::vector<uint8_t> bgr;
        std::vector<float> depth;
        if (!o.frames_file.empty()) {
            // bench.py --dump-frames: o.frames colour frames, then o.frames depth frames
            FILE* fp = std::fopen(o.frames_file.c_str(), "rb");
            if (!fp) {
                std::fprintf(stderr, "cannot open %s\n", o.frames_file.c_str());
                std::exit(2);
            }
            bgr.resize(px * 3 * count);
            depth.resize(px * count);
            bool ok = std::fseek(fp, (long)(px * 3 * first), SEEK_SET) == 0 && std::fread(bgr.data(), 1, bgr.size(), fp) == bgr.size();
            ok = ok && std::fseek(fp, (long)(px * 3 * o.frames + px * 4 * first), SEEK_SET) == 0 &&
                 std::fread(depth.data(), 4, depth.size(), fp) == depth.size();
            std::fclose(fp);
            if (!ok) {
                std::fprintf(stderr, "%s is shorter than %d frames of %dx%d\n", o.frames_file.c_str(), o.frames, o.width, o.height);
                std::exit(2);
            }
            HIP_OK(hipMemcpy(depth_dev, depth.data(), px * count * sizeof(float), hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(bgr_dev, bgr.data(), px * 3 * count, hipMemcpyHostToDevice));
        } else {
            for (int f = 0; f < count; f++) {
                make_frame(first + f, o.width, o.height, bgr, depth);
                HIP_OK(hipMemcpy(depth_dev + px * f, depth.data(), px * sizeof(float), hipMemcpyHostToDevice));
                HIP_OK(hipMemcpy(bgr_dev + px * 3 * f, bgr.data(), px * 3, hipMemcpyHostToDevice));
            }
        }
        uint8_t* smooth_dev = nullptr;
        HIP_OK(hipMalloc(&smooth_dev, px * count * 3));
        auto step = [&](hipEvent_t* ev) {
            if (ev) HIP_OK(hipEventRecord(ev[0], stream));
            KDE_OK_OR_DIE(kde_jbf_presmooth_batch(jbf, count, bgr_dev, smooth_dev, stream));             // K0
            if (ev) HIP_OK(hipEventRecord(ev[1], stream));
            KDE_OK_OR_DIE(kde_jbf_filter_batch(jbf, count, depth_dev, smooth_dev, out_dev, stream));     // K1
            if (ev) HIP_OK(hipEventRecord(ev[2], stream));
        };
        // wake-up: untimed load until the device has left its idle clock level (bench.py does the same)
        const auto tw = std::chrono::steady_clock::now();
        while (o.wakeup_ms > 0 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count() < o.wakeup_ms) {
            for (int k = 0; k < 10; k++) step(nullptr);
            HIP_OK(hipStreamSynchronize(stream));
            res.wakeup_steps += 10;
        }
        for (int k = 0; k < o.warmup; k++) step(nullptr);
        std::vector<hipEvent_t> ev((size_t)o.steps * 3);
        for (auto& e : ev) HIP_OK(hipEventCreate(&e));
        HIP_OK(hipStreamSynchronize(stream));
        if (barrier) barrier->wait();                         // all devices start their K timed steps together
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < o.steps; k++) step(&ev[(size_t)k * 3]);
        HIP_OK(hipStreamSynchronize(stream));
        res.ms_per_step = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / o.steps;
        std::vector<float> k1(o.steps);
        for (int k = 0; k < o.steps; k++) {
            float a = 0.0f, b = 0.0f;
            HIP_OK(hipEventElapsedTime(&a, ev[(size_t)k * 3], ev[(size_t)k * 3 + 1]));
            HIP_OK(hipEventElapsedTime(&b, ev[(size_t)k * 3 + 1], ev[(size_t)k * 3 + 2]));
            res.k0_ms += a / o.steps;
            res.k1_ms += b / o.steps;
            k1[k] = b;
        }
        std::sort(k1.begin(), k1.end());
        res.k1_ms_median = k1[k1.size() / 2];
        res.k1_ms_min = k1.front();
        for (auto& e : ev) HIP_OK(hipEventDestroy(e));
        HIP_OK(hipFree(smooth_dev));
        std::vector<float> out(px);
        for (int f = 0; f < count; f++) {
            HIP_OK(hipMemcpy(out.data(), out_dev + px * f, px * sizeof(float), hipMemcpyDeviceToHost));
            res.frame_hash.push_back(fnv(out.data(), px));
        }
        HIP_OK(hipFree(depth_dev));
        HIP_OK(hipFree(bgr_dev));
        HIP_OK(hipFree(out_dev));
    }
    else if (barrier) barrier->wait();        // an empty shard still takes part in the common start
    KDE_OK_OR_DIE(kde_jbf_destroy(jbf));
    HIP_OK(hipFree(blk_dev));
    HIP_OK(hipStreamDestroy(stream));
    return res;
}

// ncclCommInitAll under a deadline: a communicator that cannot be built over xGMI usually hangs instead of failing.
// The call runs in a helper thread on state of its own; past the deadline the caller goes on without it.
struct CommInit {
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
    ncclResult_t rc = ncclSystemError;
    std::vector<ncclComm_t> comms;
    std::vector<int> devs;
};

// -> ncclSuccess and the communicators, an RCCL error, or timed_out = true (the helper thread is then still inside RCCL)
ncclResult_t comm_init_all(int G, float timeout_s, std::vector<ncclComm_t>& comms, bool& timed_out, bool test_hang)
{
    auto st = std::make_shared<CommInit>();
    st->comms.assign(G, nullptr);
    st->devs.resize(G);
    for (int d = 0; d < G; d++) st->devs[d] = d;
    std::thread([st, G, test_hang]() {
        if (test_hang)
            for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));
        const ncclResult_t rc = ncclCommInitAll(st->comms.data(), G, st->devs.data());
        std::lock_guard<std::mutex> lk(st->m);
        st->rc = rc;
        st->done = true;
        st->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(st->m);
    timed_out = !st->cv.wait_for(lk, std::chrono::duration<float>(timeout_s), [&] { return st->done; });
    if (timed_out) return ncclSystemError;
    if (st->rc == ncclSuccess) comms = st->comms;
    return st->rc;
}

}  // namespace

int main(int argc, char** argv)
{
    Options o;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
        if (a == "--frames") o.frames = std::atoi(next());
        else if (a == "--width") o.width = std::atoi(next());
        else if (a == "--height") o.height = std::atoi(next());
        else if (a == "--window") o.window = std::atoi(next());
        else if (a == "--steps") o.steps = std::atoi(next());
        else if (a == "--warmup") o.warmup = std::atoi(next());
        else if (a == "--wakeup-ms") o.wakeup_ms = (float)std::atof(next());
        else if (a == "--frames-file") o.frames_file = next();
        else if (a == "--devices") o.devices = std::atoi(next());
        else if (a == "--share-device") o.share_device = std::atoi(next());
        else if (a == "--rccl-timeout") o.rccl_timeout_s = (float)std::atof(next());
        else if (a == "--verify") o.verify = true;
        else if (a == "--force-rccl-failure") o.force_rccl_failure = true;
        else {
            std::fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    int visible = 0;
    KDE_OK_OR_DIE(kde_device_count(&visible));
    const bool shared = o.share_device > 0;
    const int G = shared ? o.share_device : (o.devices > 0 ? std::min(o.devices, visible) : visible);
    if (G < 1 || visible < 1 || o.frames < 1 || o.steps < 1) {
        std::fprintf(stderr, "need at least one device, frame and step\n");
        return 2;
    }
    kde_jbf_params p;
    KDE_OK_OR_DIE(kde_jbf_default_params(&p));
    p.window_size = o.window;
    p.spatial_sigma = o.spatial_sigma;
    p.color_sigma = o.color_sigma;
    p.depth_sigma = o.depth_sigma;

    // one RCCL communicator per device (single process, one thread per device).  If RCCL does not come up the run goes on
    // in this process without it: every thread forms the parameter block itself and compares it with rank 0's ("replicas only")
    std::vector<ncclComm_t> comms(G, nullptr);
    std::string broadcast_how = "rccl ncclBroadcast";
    bool rccl_stuck = false;
    const bool test_hang = std::getenv("KDE_SHARD_REPLAY_TEST_HANG") != nullptr;      // test of the deadline
    const ncclResult_t init_rc = (o.force_rccl_failure || shared) ? ncclSystemError
                                                                  : comm_init_all(G, o.rccl_timeout_s, comms, rccl_stuck, test_hang);
    std::vector<float> root_block(kBlockLen, 0.0f);
    if (init_rc != ncclSuccess) {
        std::fill(comms.begin(), comms.end(), nullptr);
        char why[160];
        std::snprintf(why, sizeof(why), "not finished after %g s, abandoned", o.rccl_timeout_s);
        broadcast_how = shared ? std::string("replicas only (--share-device: ") + std::to_string(G) + " host threads on device 0, RCCL wants one device per rank)"
                               : std::string("replicas only (ncclCommInitAll: ") +
                                     (o.force_rccl_failure ? "failure forced by --force-rccl-failure" : rccl_stuck ? why : ncclGetErrorString(init_rc)) + ")";
        std::fprintf(stderr, "shard_replay: RCCL unavailable, %s\n", broadcast_how.c_str());
        KDE_OK_OR_DIE(kde_set_device(0));
        kde_jbf* probe = nullptr;
        KDE_OK_OR_DIE(kde_jbf_create(&probe, 8, 8, 1, &p));
        std::vector<float> table((size_t)p.window_size * p.window_size);
        KDE_OK_OR_DIE(kde_jbf_spatial_table(probe, table.data(), (int)table.size()));
        KDE_OK_OR_DIE(kde_jbf_destroy(probe));
        pack_params(p, table, root_block.data());
    }

    // contiguous blocks of ceil(N / G) frames
    const int per = (o.frames + G - 1) / G;
    std::vector<ShardResult> results(G);
    std::vector<std::thread> threads;
    SpinBarrier barrier;
    barrier.parties = G;
    const auto t0 = std::chrono::steady_clock::now();
    for (int d = 0; d < G; d++)
        threads.emplace_back([&, d]() {
            const int first = std::min(d * per, o.frames), count = std::min(per, o.frames - first);
            results[d] = run_shard(o, shared ? 0 : d, d, first, count, comms[d], p, &barrier, &root_block);
        });
    for (auto& t : threads) t.join();
    const double wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int d = 0; d < G; d++)
        if (comms[d]) NCCL_OK(ncclCommDestroy(comms[d]));

    double slowest = 0.0;
    bool tables_ok = true;
    std::vector<uint64_t> hashes;
    std::string per_dev = "[";
    for (int d = 0; d < G; d++) {
        slowest = std::max(slowest, results[d].ms_per_step);
        char buf[384];
        std::snprintf(buf, sizeof(buf), "%s{\"device\": %d, \"pci_bus_id\": \"%s\", \"ms_per_step\": %.4f, \"k0_ms\": %.4f, \"k1_ms\": %.4f, "
                      "\"k1_ms_median\": %.4f, \"k1_ms_min\": %.4f}", d ? ", " : "", d, results[d].pci_bus_id.c_str(), results[d].ms_per_step,
                      results[d].k0_ms, results[d].k1_ms, results[d].k1_ms_median, results[d].k1_ms_min);
        per_dev += buf;
        tables_ok = tables_ok && results[d].table_matches_rank0;
        hashes.insert(hashes.end(), results[d].frame_hash.begin(), results[d].frame_hash.end());
    }
    per_dev += "]";
    uint64_t all = 1469598103934665603ull;
    for (uint64_t h : hashes) all = (all ^ h) * 1099511628211ull;

    bool verified = true;
    if (o.verify) {
        // the same frames on device 0 alone: as one block, and as two half blocks -> per-frame hashes must coincide
        ShardResult one = run_shard(o, 0, 0, 0, o.frames, nullptr, p);
        const int half = o.frames / 2;
        ShardResult a = run_shard(o, 0, 0, 0, half, nullptr, p), b = run_shard(o, 0, 0, half, o.frames - half, nullptr, p);
        std::vector<uint64_t> two(a.frame_hash);
        two.insert(two.end(), b.frame_hash.begin(), b.frame_hash.end());
        verified = one.frame_hash == hashes && two == hashes;
    }
    const double mpix = (double)o.frames * o.width * o.height / (slowest * 1e-3) / 1e6;
    std::printf("{\"devices\": %d, \"host_threads\": %d, \"frames\": %d, \"frames_per_device\": %d, \"width\": %d, \"height\": %d, \"window\": %d, "
                "\"steps\": %d, \"warmup\": %d, \"wakeup_steps_before_warmup\": %d, \"input\": \"%s\", "
                "\"ms_per_step_slowest_device\": %.4f, \"mpixels_per_s\": %.1f, \"per_device\": %s, \"params_broadcast\": \"%s\", "
                "\"tables_match_rank0\": %s, \"checksum\": \"%016llx\", \"verified\": %s, \"wall_s\": %.2f}\n",
                shared ? 1 : G, G, o.frames, per, o.width, o.height, o.window, o.steps, o.warmup, results[0].wakeup_steps,
                o.frames_file.empty() ? "built-in generator" : "frames file (bench.py --dump-frames)", slowest, mpix, per_dev.c_str(),
                broadcast_how.c_str(), tables_ok ? "true" : "false",
                (unsigned long long)all, o.verify ? (verified ? "true" : "false") : "null", wall_s);
    const int rc = (tables_ok && verified) ? 0 : 1;
    if (rccl_stuck) {           // the helper thread is still inside ncclCommInitAll: no static destructor may wait for it
        std::fflush(stdout);
        std::fflush(stderr);
        std::_Exit(rc);
    }
    return rc;
}
