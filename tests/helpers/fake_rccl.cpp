// fake_rccl.cpp — TEST DOUBLE for the ten RCCL entry points liblgmi's comm.cpp binds (csrc/comm.cpp: load_rccl).
//
// Why: a test box has ONE GPU and RCCL refuses two ranks on one device ("invalid usage"), so the N > 1 wire logic of
// lgmi_comm_gather — who sends what to whom, in which order, at which offsets; the same-batch reductions — could
// only run with world = 1.  This library carries the same calls between PROCESSES THAT SHARE ONE GPU through files
// in /dev/shm (device -> host -> file -> host -> device).  It is slow and synchronous by design; it is loaded only
// when LGMI_RCCL_LIB points at it (tests/test_gpu_gather2.py) and is not part of the product.
//
// Semantics kept from NCCL: sends and receives between a pair of ranks match in issue order; operations between
// ncclGroupStart and ncclGroupEnd are issued together (all sends are posted before any receive blocks, so a
// grouped exchange cannot deadlock); every call is ordered after the work already on its stream (the stream is
// drained first) and its effect is visible to work enqueued after it.
//
//   g++ -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.cpp -o librccl_fake.so -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {
struct Comm {
    int rank = 0, world = 1;
    std::string dir;
    std::map<int, unsigned long long> send_seq, recv_seq;   // per peer
    unsigned long long coll_seq = 0;
};
struct Op { bool send; void* buf; size_t bytes; int peer; Comm* c; hipStream_t st; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

bool write_file(const std::string& path, const void* data, size_t n) {
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = n == 0 || fwrite(data, 1, n, f) == n;
    fclose(f);
    return ok && rename(tmp.c_str(), path.c_str()) == 0;
}

bool read_file(const std::string& path, void* data, size_t n) {   // waits for the peer (two minutes at most)
    const auto t0 = std::chrono::steady_clock::now();
    struct stat sb;
    while (stat(path.c_str(), &sb) != 0) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if ((size_t)sb.st_size != n) { fprintf(stderr, "[fake rccl] %s: %zu bytes, expected %zu\n", path.c_str(), (size_t)sb.st_size, n); return false; }
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool ok = n == 0 || fread(data, 1, n, f) == n;
    fclose(f);
    unlink(path.c_str());
    return ok;
}

std::string msg(const Comm* c, const char* kind, int src, int dst, unsigned long long seq) {
    char b[96];
    snprintf(b, sizeof b, "/%s_%d_%d_%llu", kind, src, dst, seq);
    return c->dir + b;
}

ncclResult_t do_send(const Op& o) {
    std::vector<char> h(o.bytes);
    if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
    if (o.bytes && hipMemcpy(h.data(), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    return write_file(msg(o.c, "p2p", o.c->rank, o.peer, o.c->send_seq[o.peer]++), h.data(), o.bytes) ? ncclSuccess : ncclSystemError;
}
ncclResult_t do_recv(const Op& o) {
    std::vector<char> h(o.bytes);
    if (!read_file(msg(o.c, "p2p", o.peer, o.c->rank, o.c->recv_seq[o.peer]++), h.data(), o.bytes)) return ncclSystemError;
    if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
    if (o.bytes && hipMemcpy(o.buf, h.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
ncclResult_t flush() {
    ncclResult_t r = ncclSuccess;
    for (const Op& o : g_ops) if (o.send && r == ncclSuccess) r = do_send(o);
    for (const Op& o : g_ops) if (!o.send && r == ncclSuccess) r = do_recv(o);
    g_ops.clear();
    return r;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "frccl_%d_%lld", (int)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
    Comm* c = new Comm;
    c->rank = rank; c->world = world;
    c->dir = std::string("/dev/shm/") + id.internal;
    mkdir(c->dir.c_str(), 0700);                             // every rank tries; the first one wins
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (c) { rmdir(c->dir.c_str()); delete c; }              // succeeds for the last rank out (the directory is empty then)
    return ncclSuccess;
}

// what lgmi_comm_info() asks (optional entry points of the real library)
ncclResult_t ncclGetVersion(int* v) { *v = 0; return ncclSuccess; }                       // 0.0.0: not a release of RCCL
ncclResult_t ncclCommCount(const ncclComm_t comm, int* n) { *n = reinterpret_cast<Comm*>(comm)->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* r) { *r = reinterpret_cast<Comm*>(comm)->rank; return ncclSuccess; }

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return (--g_depth == 0) ? flush() : ncclSuccess; }

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    g_ops.push_back(Op{true, const_cast<void*>(buf), count * type_size(t), peer, reinterpret_cast<Comm*>(comm), st});
    return g_depth ? ncclSuccess : flush();
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    g_ops.push_back(Op{false, buf, count * type_size(t), peer, reinterpret_cast<Comm*>(comm), st});
    return g_depth ? ncclSuccess : flush();
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t st) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t n = count * type_size(t);
    const unsigned long long seq = c->coll_seq++;
    std::vector<char> h(n);
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    if (n && hipMemcpy(h.data(), send, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank && !write_file(msg(c, "ag", c->rank, r, seq), h.data(), n)) return ncclSystemError;
    for (int r = 0; r < c->world; ++r) {
        if (r != c->rank && !read_file(msg(c, "ag", r, c->rank, seq), h.data(), n)) return ncclSystemError;
        if (r == c->rank && n && hipMemcpy(h.data(), send, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (n && hipMemcpy((char*)recv + (size_t)r * n, h.data(), n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

ncclResult_t ncclReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, int root,
                        ncclComm_t comm, hipStream_t st) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (op != ncclSum || (t != ncclUint64 && t != ncclUint32)) return ncclInvalidArgument;   // what comm.cpp uses
    const size_t n = count * type_size(t);
    const unsigned long long seq = c->coll_seq++;
    std::vector<char> h(n), acc(n);
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    if (n && hipMemcpy(acc.data(), send, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (c->rank != root) return write_file(msg(c, "rd", c->rank, root, seq), acc.data(), n) ? ncclSuccess : ncclSystemError;
    for (int r = 0; r < c->world; ++r) {
        if (r == root) continue;
        if (!read_file(msg(c, "rd", r, root, seq), h.data(), n)) return ncclSystemError;
        if (t == ncclUint64) for (size_t k = 0; k < count; ++k) ((unsigned long long*)acc.data())[k] += ((unsigned long long*)h.data())[k];
        else                 for (size_t k = 0; k < count; ++k) ((unsigned int*)acc.data())[k] += ((unsigned int*)h.data())[k];
    }
    if (n && hipMemcpy(recv, acc.data(), n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake rccl: HIP error";
        case ncclSystemError: return "fake rccl: peer message missing, short or unwritable";
        case ncclInvalidArgument: return "fake rccl: unsupported reduction";
        default: return "fake rccl: error";
    }
}

}  // extern "C"
