// stub_rccl.cpp — TEST INFRASTRUCTURE, not product: a stand-in for the six RCCL entry points libzgpt2_hip binds by dlopen
// (csrc/dist.hip), so that the library's multi-rank path — the id shipped to the ranks, zg_dist_init on every rank, the
// weight-region broadcast with a RECEIVING rank, the all-gather, finalize — can run with world_size > 1 on a box with ONE GPU
// (RCCL itself refuses two ranks on one device: "Duplicate GPU detected").  Transport: files in a per-communicator directory under
// /tmp (the 128-byte id names it); device buffers are staged through the host with hipMemcpy.  Same C ABI as rccl.h for the calls
// used: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclBroadcast, ncclAllGather, ncclGetErrorString.
// Selected with ZGPT2_RCCL_LIB=<path to this .so> (a test hook of dist.hip).  RCCL proper is exercised with one rank
// (tests/test_dist_gpu.py) and by bench.py / torch.distributed on multi-GPU nodes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

typedef struct { char internal[128]; } ncclUniqueId;
struct StubComm {
    std::string dir;
    int nranks, rank;
    unsigned seq;
};
typedef StubComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
static const int kOk = 0, kErr = 5 /* ncclInvalidUsage */, kSys = 2 /* ncclSystemError */;

static bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
static bool wait_for(const std::string& p, double seconds = 60.0) {
    for (int i = 0; i < (int)(seconds * 1000); ++i) {
        if (exists(p)) return true;
        usleep(1000);
    }
    return false;
}
static bool write_file(const std::string& p, const void* data, size_t n) {  // atomic: readers never see a partial file
    const std::string tmp = p + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = n == 0 || fwrite(data, 1, n, f) == n;
    fclose(f);
    return ok && rename(tmp.c_str(), p.c_str()) == 0;
}
static bool read_file(const std::string& p, void* data, size_t n) {
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) return false;
    const bool ok = n == 0 || fread(data, 1, n, f) == n;
    fclose(f);
    return ok;
}

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "zgstub_%d_%ld_%ld", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec);
    return kOk;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || strncmp(id.internal, "zgstub_", 7) != 0) return kErr;
    id.internal[127] = 0;
    StubComm* c = new StubComm{std::string("/tmp/") + id.internal, nranks, rank, 0};
    mkdir(c->dir.c_str(), 0700);  // (every rank tries; one wins)
    if (!write_file(c->dir + "/ready." + std::to_string(rank), "", 0)) { delete c; return kSys; }
    for (int r = 0; r < nranks; ++r)  // the rendezvous ncclCommInitRank is
        if (!wait_for(c->dir + "/ready." + std::to_string(r))) { delete c; return kSys; }
    *comm = c;
    return kOk;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return kErr;
    // leave the directory to the last rank out: every rank marks itself gone, the one that sees all marks removes the files
    write_file(c->dir + "/gone." + std::to_string(c->rank), "", 0);
    bool all = true;
    for (int r = 0; r < c->nranks; ++r) all = all && exists(c->dir + "/gone." + std::to_string(r));
    if (all) {
        const std::string cmd = "rm -rf '" + c->dir + "'";
        if (system(cmd.c_str()) != 0) { /* best effort */ }
    }
    delete c;
    return kOk;
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t, int root, ncclComm_t c, hipStream_t s) {
    if (!c || root < 0 || root >= c->nranks) return kErr;
    if (hipStreamSynchronize(s) != hipSuccess) return kSys;
    const std::string f = c->dir + "/bcast." + std::to_string(++c->seq);
    std::vector<char> h(count);
    if (c->rank == root) {
        if (hipMemcpy(h.data(), send, count, hipMemcpyDeviceToHost) != hipSuccess) return kSys;
        if (!write_file(f, h.data(), count)) return kSys;
        if (recv != send && hipMemcpy(recv, send, count, hipMemcpyDeviceToDevice) != hipSuccess) return kSys;
    } else {
        if (!wait_for(f) || !read_file(f, h.data(), count)) return kSys;
        if (hipMemcpy(recv, h.data(), count, hipMemcpyHostToDevice) != hipSuccess) return kSys;
    }
    return kOk;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t c, hipStream_t s) {
    if (!c) return kErr;
    if (hipStreamSynchronize(s) != hipSuccess) return kSys;
    const std::string base = c->dir + "/gather." + std::to_string(++c->seq) + ".";
    std::vector<char> h(count);
    if (hipMemcpy(h.data(), send, count, hipMemcpyDeviceToHost) != hipSuccess) return kSys;
    if (!write_file(base + std::to_string(c->rank), h.data(), count)) return kSys;
    for (int r = 0; r < c->nranks; ++r) {
        if (!wait_for(base + std::to_string(r)) || !read_file(base + std::to_string(r), h.data(), count)) return kSys;
        if (hipMemcpy(static_cast<char*>(recv) + (size_t)r * count, h.data(), count, hipMemcpyHostToDevice) != hipSuccess) return kSys;
    }
    return kOk;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == kOk ? "no error" : r == kErr ? "stub: invalid usage" : "stub: system error (file transport)"; }

}  // extern "C"
