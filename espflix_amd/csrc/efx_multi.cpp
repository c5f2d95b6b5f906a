// efx_multi.cpp -- the multi-device entry points of include/efx.h: one efx_ctx and one host thread per device,
// streams dealt in contiguous blocks.  Host code only: every device does exactly what a
// single-device caller would make it do, through the same C-ABI.
#include <hip/hip_runtime.h>

#include <sched.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "efx.h"

struct efx_multi {
    struct Worker {
        int device = 0;
        efx_ctx* ctx = nullptr;
        std::thread thread;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int()> job;  // pending work (empty: idle)
        bool quit = false, done = true;
        int result = EFX_OK;
        int first = 0, count = 0;  // its block of the last upload
    };
    std::vector<Worker*> w;
    efx_config cfg{};
    int n_streams = 0;
    std::string err;
};

namespace {

// ---- NUMA placement of a device's host side ---------------------------------------------------------------------------------
// At eight devices per node the ingest path is a host-memory problem: every context stages (or reads in place) tens of
// megabytes per step, and a staging buffer or a worker thread on the other socket pays the inter-socket link twice.  The
// kernel publishes a PCI device's node in sysfs; the thread that drives a device (and therefore first-touches its pinned
// staging memory) is bound to that node's CPUs before it creates the context.  EFX_SYSFS_ROOT relocates the tree
// (tests/test_numa.py runs this against a fake one), EFX_NUMA=0 turns the binding off.
std::string sysfs_root()
{
    const char* r = getenv("EFX_SYSFS_ROOT");
    return r ? std::string(r) : std::string();
}

bool read_line(const std::string& path, std::string* out)
{
    FILE* f = fopen(path.c_str(), "r");
    if (!f)
        return false;
    char buf[4096];
    const bool ok = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!ok)
        return false;
    size_t n = strlen(buf);
    while (n && (buf[n - 1] == '\n' || buf[n - 1] == ' '))
        buf[--n] = 0;
    *out = buf;
    return true;
}

// "0-15,32-47" -> cpu numbers
std::vector<int> parse_cpulist(const std::string& s)
{
    std::vector<int> out;
    const char* p = s.c_str();
    while (*p) {
        char* e = nullptr;
        long a = strtol(p, &e, 10);
        if (e == p)
            break;
        long b = a;
        p = e;
        if (*p == '-') {
            b = strtol(p + 1, &e, 10);
            p = e;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            out.push_back((int)c);
        if (*p == ',')
            p++;
    }
    return out;
}

void worker_main(efx_multi::Worker* w)
{
    if (!(getenv("EFX_NUMA") && atoi(getenv("EFX_NUMA")) == 0)) {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, w->device) == hipSuccess)
            (void)efx_numa_bind_thread(efx_numa_node_of_pci(bus));
    }
    (void)hipSetDevice(w->device);  // the HIP context of this thread: every efx_* call on w->ctx runs here
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        w->cv.wait(lk, [&] { return w->quit || w->job; });
        if (w->quit)
            return;
        std::function<int()> job;
        job.swap(w->job);
        lk.unlock();
        const int r = job();
        lk.lock();
        w->result = r;
        w->done = true;
        w->cv.notify_all();
    }
}

// run f(r, worker) on every device's thread, wait for all, return the first failure
int fan_out(efx_multi* m, const std::function<int(int, efx_multi::Worker*)>& f)
{
    for (size_t r = 0; r < m->w.size(); r++) {
        efx_multi::Worker* w = m->w[r];
        std::lock_guard<std::mutex> lk(w->mu);
        w->done = false;
        w->job = [=] { return f((int)r, w); };
        w->cv.notify_all();
    }
    int rc = EFX_OK;
    for (size_t r = 0; r < m->w.size(); r++) {
        efx_multi::Worker* w = m->w[r];
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->done; });
        if (w->result != EFX_OK && rc == EFX_OK) {
            rc = w->result;
            m->err = "device " + std::to_string(w->device) + ": " + (w->ctx ? efx_last_error(w->ctx) : efx_status_string(rc));
        }
    }
    return rc;
}

}  // namespace

extern "C" {

int efx_numa_node_of_pci(const char* pci_bus_id)
{
    if (!pci_bus_id || !*pci_bus_id)
        return -1;
    std::string id(pci_bus_id);
    for (char& c : id)
        c = (char)tolower((unsigned char)c);
    if (id.size() == 7)  // "c1:00.0" -> "0000:c1:00.0"
        id = "0000:" + id;
    std::string line;
    if (!read_line(sysfs_root() + "/sys/bus/pci/devices/" + id + "/numa_node", &line))
        return -1;
    return atoi(line.c_str());  // (-1 on single-node machines)
}

int efx_numa_cpus_of_node(int node, int* cpus, int cap)
{
    if (node < 0)
        return 0;
    std::string line;
    if (!read_line(sysfs_root() + "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", &line))
        return 0;
    const std::vector<int> v = parse_cpulist(line);
    for (size_t i = 0; i < v.size() && cpus && (int)i < cap; i++)
        cpus[i] = v[i];
    return (int)v.size();
}

int efx_numa_bind_thread(int node)
{
    int cpus[CPU_SETSIZE];
    const int n = efx_numa_cpus_of_node(node, cpus, CPU_SETSIZE);
    if (n <= 0)
        return 0;
    // only CPUs this thread may use already (a container's cpuset): an empty intersection leaves the mask alone
    cpu_set_t now, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof now, &now) != 0)
        return 0;
    int kept = 0;
    for (int i = 0; i < n && i < CPU_SETSIZE; i++)
        if (CPU_ISSET(cpus[i], &now)) {
            CPU_SET(cpus[i], &want);
            kept++;
        }
    if (!kept || sched_setaffinity(0, sizeof want, &want) != 0)
        return 0;
    return kept;
}

int efx_partition_first(int total, int parts, int part)
{
    if (total < 0 || parts <= 0 || part < 0)
        return -1;
    if (part >= parts)
        return total;
    return (int)(((int64_t)part * total + parts - 1) / parts);  // stream k lives on floor(k * parts / total)
}

int efx_multi_create(const efx_config* cfg, const int* devices, int n_devices, efx_multi** out)
{
    if (!cfg || !devices || !out || n_devices <= 0 || cfg->hip_stream)
        return EFX_ERR_ARG;
    efx_multi* m = new efx_multi;
    m->cfg = *cfg;
    for (int r = 0; r < n_devices; r++) {
        efx_multi::Worker* w = new efx_multi::Worker;
        w->device = devices[r];
        m->w.push_back(w);
        w->thread = std::thread(worker_main, w);
    }
    const int rc = fan_out(m, [&](int, efx_multi::Worker* w) {
        efx_config c = m->cfg;
        c.device = w->device;
        return efx_create(&c, &w->ctx);
    });
    if (rc != EFX_OK) {
        efx_multi_destroy(m);
        return rc;
    }
    *out = m;
    return EFX_OK;
}

void efx_multi_destroy(efx_multi* m)
{
    if (!m)
        return;
    (void)fan_out(m, [](int, efx_multi::Worker* w) {
        efx_destroy(w->ctx);
        w->ctx = nullptr;
        return (int)EFX_OK;
    });
    for (efx_multi::Worker* w : m->w) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->quit = true;
            w->cv.notify_all();
        }
        w->thread.join();
        delete w;
    }
    delete m;
}

int efx_multi_device_count(const efx_multi* m) { return m ? (int)m->w.size() : 0; }
efx_ctx* efx_multi_context(efx_multi* m, int r) { return m && r >= 0 && r < (int)m->w.size() ? m->w[r]->ctx : nullptr; }
const char* efx_multi_last_error(const efx_multi* m) { return m ? m->err.c_str() : "null efx_multi"; }

int efx_multi_upload_streams(efx_multi* m, int n_streams, const uint8_t* const* data, const size_t* len, int format)
{
    if (!m || !data || !len || n_streams <= 0)
        return EFX_ERR_ARG;
    const int R = (int)m->w.size();
    for (int r = 0; r < R; r++)
        if (efx_partition_first(n_streams, R, r + 1) - efx_partition_first(n_streams, R, r) > m->cfg.max_streams) {
            m->err = "efx_multi_upload_streams: more streams per device than max_streams";
            return EFX_ERR_CAPACITY;
        }
    const int rc = fan_out(m, [&](int r, efx_multi::Worker* w) {
        w->first = efx_partition_first(n_streams, R, r);
        w->count = efx_partition_first(n_streams, R, r + 1) - w->first;
        return w->count ? efx_upload_streams(w->ctx, w->count, data + w->first, len + w->first, format) : (int)EFX_OK;
    });
    if (rc == EFX_OK)
        m->n_streams = n_streams;
    return rc;
}

int efx_multi_locate(const efx_multi* m, int stream, int* device_index, int* local_stream)
{
    if (!m || stream < 0 || stream >= m->n_streams)
        return EFX_ERR_ARG;
    for (size_t r = 0; r < m->w.size(); r++)
        if (stream < m->w[r]->first + m->w[r]->count) {
            if (device_index)
                *device_index = (int)r;
            if (local_stream)
                *local_stream = stream - m->w[r]->first;
            return EFX_OK;
        }
    return EFX_ERR_ARG;
}

int efx_multi_decode(efx_multi* m)
{
    if (!m)
        return EFX_ERR_ARG;
    if (!m->n_streams)
        return EFX_ERR_STATE;
    return fan_out(m, [](int, efx_multi::Worker* w) { return w->count ? efx_decode(w->ctx) : (int)EFX_OK; });
}

int efx_multi_sync(efx_multi* m)
{
    return m ? fan_out(m, [](int, efx_multi::Worker* w) { return efx_sync(w->ctx); }) : (int)EFX_ERR_ARG;
}

int efx_multi_reset(efx_multi* m)
{
    return m ? fan_out(m, [](int, efx_multi::Worker* w) { return efx_reset(w->ctx); }) : (int)EFX_ERR_ARG;
}

int efx_multi_results(efx_multi* m, int* n_pictures, uint32_t* status)
{
    if (!m)
        return EFX_ERR_ARG;
    return fan_out(m, [=](int, efx_multi::Worker* w) {
        for (int i = 0; i < w->count; i++) {
            int rc = EFX_OK;
            if (n_pictures && (rc = efx_picture_count(w->ctx, i, n_pictures + w->first + i)) != EFX_OK)
                return rc;
            if (status && (rc = efx_stream_status(w->ctx, i, status + w->first + i)) != EFX_OK)
                return rc;
        }
        return (int)EFX_OK;
    });
}

int efx_multi_frame_hashes(efx_multi* m, uint64_t* out)
{
    if (!m || !out)
        return EFX_ERR_ARG;
    const int D = m->cfg.ring_depth < 2 ? 2 : m->cfg.ring_depth;
    return fan_out(m, [=](int, efx_multi::Worker* w) {
        return w->count ? efx_frame_hashes(w->ctx, 0, w->count, out + (size_t)w->first * D) : (int)EFX_OK;
    });
}

}  // extern "C"
