// efx_scale.cpp -- C++ scaling harness over the multi-device C-ABI (include/efx.h, efx_multi_*): the north star's
// "host code stays C/C++" for the node-level job that bench.py runs under torch.distributed.
//
//   tools/efx_scale [--devices N] [--streams S] [--steps K] [--warmup W] [--golden tests/golden/bench_gop12.u64]
//   tools/efx_scale --dry-run 1 --devices N [--streams S]      no device is touched: prints which streams each of N devices
//                                                              would decode (efx_partition_first) and where its host thread
//                                                              would be bound (efx_numa_*, given --pci id0,id1,...)
//
// N devices (default: all visible), S synthetic GOP(12) streams per device (ids 0 .. N*S-1, SURVEY.md 8d config 3 / 5:
// stream k on device floor(k*N/(N*S))), one efx_multi (a context and a host thread per device), no collective on the
// data path.  1. every picture of every stream is decoded with all pictures kept and the per-stream frame-chain hashes
//  are ALL-GATHERED THROUGH RCCL (ncclAllGather over xGMI, one communicator per device in this process) and
// compared on device 0 with the golden table of the unmodified reference decoder; 2. K timed steps with the double
// buffer; prints one line per device and the aggregate.  Runs with one device too (the development lease has one).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "efx.h"

extern "C" {  // espflix_amd/gen/libefx_gen.so: the synthetic stream generator (workload tooling)
void* efxgen_batch_create(uint32_t first_id, int n_streams, int n_pictures, int gop, uint32_t flags, int threads);
void efxgen_batch_destroy(void* h);
uint64_t efxgen_batch_es_size(void* h, int i);
uint64_t efxgen_batch_es_copy(void* h, int i, uint8_t* out, uint64_t cap);
uint64_t efxgen_fnv1a64(const uint8_t* p, uint64_t n, uint64_t h);
}

#define CHECK(x)                                                                              \
    do {                                                                                      \
        int rc_ = (x);                                                                        \
        if (rc_ != 0) {                                                                       \
            fprintf(stderr, "efx_scale: %s failed (%d) at line %d\n", #x, rc_, __LINE__);     \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

int main(int argc, char** argv)
{
    int n_dev = 0, S = 1024, steps = 20, warmup = 3, dry_run = 0;
    const int P = 12;
    std::string golden, pci;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--dry-run")) dry_run = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--pci")) pci = argv[i + 1];
        if (!strcmp(argv[i], "--devices")) n_dev = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--streams")) S = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--steps")) steps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--warmup")) warmup = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--golden")) golden = argv[i + 1];
    }
    if (dry_run) {
        // the job's partition and placement, without a device: what an 8-GPU node would be asked to do
        if (n_dev <= 0 || S <= 0) {
            fprintf(stderr, "efx_scale --dry-run: give --devices N\n");
            return 1;
        }
        const int total = n_dev * S;
        std::vector<std::string> ids;
        for (size_t a = 0; a < pci.size();) {
            const size_t b = pci.find(',', a);
            ids.push_back(pci.substr(a, b == std::string::npos ? std::string::npos : b - a));
            a = b == std::string::npos ? pci.size() : b + 1;
        }
        int covered = 0;
        for (int r = 0; r < n_dev; r++) {
            const int first = efx_partition_first(total, n_dev, r), end = efx_partition_first(total, n_dev, r + 1);
            const int node = r < (int)ids.size() ? efx_numa_node_of_pci(ids[r].c_str()) : -1;
            printf("device %d: streams [%d, %d) = %d, first id on it %d (floor(k*%d/%d) = %d), numa node %d, %d cpus\n", r, first, end,
                   end - first, first, n_dev, total, (int)((long long)first * n_dev / total), node, efx_numa_cpus_of_node(node, nullptr, 0));
            if ((long long)first * n_dev / total != r || end - first > S)
                return 1;
            covered += end - first;
        }
        printf("DRY_RUN devices=%d streams_total=%d covered=%d\n", n_dev, total, covered);
        return covered == total ? 0 : 1;
    }
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        fprintf(stderr, "efx_scale: no HIP device (libefx has no CPU path)\n");
        return 1;
    }
    if (n_dev <= 0 || n_dev > visible)
        n_dev = visible;
    std::vector<int> devices(n_dev);
    for (int r = 0; r < n_dev; r++)
        devices[r] = r;
    const int total = n_dev * S;

    // ---- the batch: ids 0 .. total-1, generated on the host cores -----------------------------------------------------
    void* batch = efxgen_batch_create(0, total, P, 12, 0, 16);
    if (!batch)
        return 1;
    std::vector<std::vector<uint8_t>> es(total);
    std::vector<const uint8_t*> ptr(total);
    std::vector<size_t> len(total);
    size_t per_dev_bytes = 0, bytes = 0;
    for (int k = 0; k < total; k++) {
        es[k].resize(efxgen_batch_es_size(batch, k));
        efxgen_batch_es_copy(batch, k, es[k].data(), es[k].size());
        ptr[k] = es[k].data();
        len[k] = es[k].size();
        bytes += len[k];
        if ((k + 1) % S == 0) {
            per_dev_bytes = bytes > per_dev_bytes ? bytes : per_dev_bytes;
            bytes = 0;
        }
    }
    efxgen_batch_destroy(batch);

    // ---- 1. throughput: the reference's double buffer, K timed steps (before RCCL is initialised in this process) -----------------------------------------------------
    efx_config cfg{};
    cfg.max_streams = S;
    cfg.max_pictures = P;
    cfg.ring_depth = 2;
    cfg.max_stream_bytes = per_dev_bytes + 64 * (size_t)S;
    efx_multi* m = nullptr;
    CHECK(efx_multi_create(&cfg, devices.data(), n_dev, &m));
    CHECK(efx_multi_upload_streams(m, total, ptr.data(), len.data(), EFX_FORMAT_ES));
    for (int i = 0; i < warmup; i++)
        CHECK(efx_multi_decode(m));
    CHECK(efx_multi_sync(m));
    for (int r = 0; r < n_dev; r++) {
        CHECK(hipSetDevice(devices[r]));
        CHECK(efx_set_timing(efx_multi_context(m, r), 1));
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; i++)
        CHECK(efx_multi_decode(m));
    CHECK(efx_multi_sync(m));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int r = 0; r < n_dev; r++) {
        CHECK(hipSetDevice(devices[r]));
        efx_timing t{};
        CHECK(efx_get_timing(efx_multi_context(m, r), &t));
        printf("device %d: %d streams, k_index %.3f k_parse %.3f k_recon %.3f ms per step (stage sums), %llu pictures per step\n", devices[r],
               efx_partition_first(total, n_dev, r + 1) - efx_partition_first(total, n_dev, r), t.index_ms, t.parse_ms, t.recon_ms,
               (unsigned long long)t.pictures);
    }
    char result[512];
    snprintf(result, sizeof(result), "{\"metric\": \"MPEG-1 352x192 frames/s\", \"value\": %.1f, \"n_gpus\": %d, \"steps\": %d, \"ms_per_step\": %.4f, "
           "\"streams_total\": %d, \"harness\": \"tools/efx_scale.cpp (efx_multi C-ABI, RCCL all-gather)\"}\n",
           (double)total * P * steps / dt, n_dev, steps, dt / steps * 1e3, total);
    efx_multi_destroy(m);
    m = nullptr;

    // ---- 2. parity: every picture kept, chain hashes gathered through RCCL ----------------------------------------------
    cfg.ring_depth = P + 1;
    CHECK(efx_multi_create(&cfg, devices.data(), n_dev, &m));
    CHECK(efx_multi_upload_streams(m, total, ptr.data(), len.data(), EFX_FORMAT_ES));
    CHECK(efx_multi_decode(m));
    CHECK(efx_multi_sync(m));
    std::vector<int> npic(total);
    std::vector<uint32_t> status(total);
    CHECK(efx_multi_results(m, npic.data(), status.data()));
    std::vector<uint64_t> hashes((size_t)total * (P + 1));
    CHECK(efx_multi_frame_hashes(m, hashes.data()));
    std::vector<uint64_t> chain(total);
    for (int k = 0; k < total; k++) {
        if (npic[k] != P || status[k]) {
            fprintf(stderr, "efx_scale: stream %d decoded %d pictures, status %u\n", k, npic[k], status[k]);
            return 1;
        }
        int dev, local;
        CHECK(efx_multi_locate(m, k, &dev, &local));
        CHECK(hipSetDevice(devices[dev]));
        uint64_t h = 0xcbf29ce484222325ull;
        for (int p = 0; p < P; p++) {
            int slot;
            CHECK(efx_stream_picture_slot(efx_multi_context(m, dev), local, p, &slot));
            const uint64_t fh = hashes[(size_t)k * (P + 1) + slot];
            h = efxgen_fnv1a64(reinterpret_cast<const uint8_t*>(&fh), 8, h);  // (little-endian host)
        }
        chain[k] = h;
    }
    efx_multi_destroy(m);

    // all-gather over RCCL: device r contributes the hashes of ITS block, every device ends up with all of them
    std::vector<ncclComm_t> comms(n_dev);
    CHECK(ncclCommInitAll(comms.data(), n_dev, devices.data()));
    std::vector<uint64_t*> d_send(n_dev), d_recv(n_dev);
    std::vector<hipStream_t> st(n_dev);
    for (int r = 0; r < n_dev; r++) {
        CHECK(hipSetDevice(devices[r]));
        CHECK(hipStreamCreate(&st[r]));
        CHECK(hipMalloc(reinterpret_cast<void**>(&d_send[r]), (size_t)S * 8));
        CHECK(hipMalloc(reinterpret_cast<void**>(&d_recv[r]), (size_t)total * 8));
        CHECK(hipMemcpy(d_send[r], chain.data() + (size_t)efx_partition_first(total, n_dev, r), (size_t)S * 8, hipMemcpyHostToDevice));
    }
    CHECK(ncclGroupStart());
    for (int r = 0; r < n_dev; r++)
        CHECK(ncclAllGather(d_send[r], d_recv[r], (size_t)S, ncclUint64, comms[r], st[r]));
    CHECK(ncclGroupEnd());
    std::vector<uint64_t> gathered(total);
    for (int r = 0; r < n_dev; r++) {
        CHECK(hipSetDevice(devices[r]));
        CHECK(hipStreamSynchronize(st[r]));
        CHECK(hipMemcpy(gathered.data(), d_recv[r], (size_t)total * 8, hipMemcpyDeviceToHost));
        if (memcmp(gathered.data(), chain.data(), (size_t)total * 8)) {
            fprintf(stderr, "efx_scale: device %d holds a different gathered table\n", r);
            return 1;
        }
        (void)hipFree(d_send[r]);
        (void)hipFree(d_recv[r]);
        (void)hipStreamDestroy(st[r]);
        ncclCommDestroy(comms[r]);
    }
    const char* parity = "unchecked (no --golden)";
    if (!golden.empty()) {
        FILE* f = fopen(golden.c_str(), "rb");
        if (!f) {
            fprintf(stderr, "efx_scale: cannot open %s\n", golden.c_str());
            return 1;
        }
        std::vector<uint64_t> row(P);
        for (int k = 0; k < total; k++) {
            if (fread(row.data(), 8, P, f) != (size_t)P) {
                fprintf(stderr, "efx_scale: golden table has fewer than %d rows\n", total);
                return 1;
            }
            uint64_t h = 0xcbf29ce484222325ull;
            for (int p = 0; p < P; p++)
                h = efxgen_fnv1a64(reinterpret_cast<const uint8_t*>(&row[p]), 8, h);
            if (h != gathered[k]) {
                fprintf(stderr, "efx_scale: stream %d differs from the reference decoder\n", k);
                return 1;
            }
        }
        fclose(f);
        parity = "every stream equals the reference decoder";
    }
    printf("RCCL all-gather of %d chain hashes over %d device(s): ok; parity: %s\n", total, n_dev, parity);
    fputs(result, stdout);  // (the number is printed only for a decoder that agrees with the reference)

    return 0;
}
