// tests/simt/fake_rccl.cpp — TEST INFRASTRUCTURE: a stand-in for librccl.so.1 for the SIMT-emulated build.
//
// libsublinear_hip resolves RCCL at run time (dlopen "librccl.so.1"); with tests/simt/_build first in LD_LIBRARY_PATH it finds this file
// instead.  It implements the twelve entry points sl_comm.hip uses — ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy / Abort,
// ncclCommGetAsyncError, ncclAllGather, ncclAllReduce(sum, double), ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd,
// ncclGetErrorString — between the PROCESSES of a multi-rank test, as messages in /dev/shm (one file per message, written under a
// temporary name and renamed, so a sender never waits; a receiver polls for its file, reads it and removes it).  Synchronous: an
// operation is complete when the call returns (the emulator's streams are synchronous too).
// What it checks is the LIBRARY's use of the collectives — counts, offsets, in-place buffers, the compact halo buffer of the all-reduce
// form, the send / receive lists — not RCCL: the results must still equal the one-GPU solve bit for bit.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5,
               ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef int ncclRedOp_t;
typedef int ncclDataType_t;
struct fake_comm {
    std::string id;
    int rank = 0, world = 0;
    std::vector<uint64_t> sent, received;        // message counters per peer
    struct op { bool send; void *buf; size_t bytes; int peer; };
    std::vector<op> group;
    int group_depth = 0;
};
typedef fake_comm *ncclComm_t;
}

namespace {
size_t type_bytes(ncclDataType_t t) { return t == 8 ? 8 : t == 7 ? 4 : t == 4 || t == 5 ? 8 : t == 2 || t == 3 ? 4 : 1; }   // ncclDouble = 8 is what the library sends
long timeout_ms() { const char *e = getenv("SL_COMM_TIMEOUT_MS"); const long v = e ? atol(e) : 60000; return v > 0 ? 3 * v : 60000; }
std::string msg_path(const fake_comm *c, int from, int to, uint64_t seq)
{
    char b[256];
    snprintf(b, sizeof(b), "/dev/shm/simt_rccl_%s_%d_%d_%llu", c->id.c_str(), from, to, (unsigned long long)seq);
    return b;
}
ncclResult_t put(fake_comm *c, int peer, const void *buf, size_t bytes)
{
    const std::string path = msg_path(c, c->rank, peer, c->sent[peer]++), tmp = path + ".tmp";
    const int fd = open(tmp.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
    if (fd < 0) return ncclSystemError;
    const char *p = static_cast<const char *>(buf);
    for (size_t off = 0; off < bytes;) { const ssize_t w = write(fd, p + off, bytes - off); if (w <= 0) { close(fd); return ncclSystemError; } off += (size_t)w; }
    close(fd);
    return rename(tmp.c_str(), path.c_str()) == 0 ? ncclSuccess : ncclSystemError;
}
ncclResult_t get(fake_comm *c, int peer, void *buf, size_t bytes)
{
    const std::string path = msg_path(c, peer, c->rank, c->received[peer]++);
    const auto t0 = std::chrono::steady_clock::now();
    int fd = -1;
    for (unsigned spin = 0; (fd = open(path.c_str(), O_RDONLY)) < 0; ++spin) {
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > (double)timeout_ms()) return ncclSystemError;
        std::this_thread::sleep_for(std::chrono::microseconds(spin < 100 ? 20 : 500));
    }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (size_t)sb.st_size != bytes) { close(fd); unlink(path.c_str()); return ncclInvalidArgument; }   // the two sides disagree about a count
    char *p = static_cast<char *>(buf);
    for (size_t off = 0; off < bytes;) { const ssize_t r = read(fd, p + off, bytes - off); if (r <= 0) { close(fd); return ncclSystemError; } off += (size_t)r; }
    close(fd);
    unlink(path.c_str());
    return ncclSuccess;
}
}   // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    const int fd = open("/dev/urandom", O_RDONLY);
    unsigned char r[8] = {0};
    if (fd >= 0) { (void)!read(fd, r, sizeof(r)); close(fd); }
    snprintf(id->internal, sizeof(id->internal), "%02x%02x%02x%02x%02x%02x%02x%02x_%d", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], (int)getpid());
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    fake_comm *c = new fake_comm();
    id.internal[127] = 0;
    c->id = id.internal; c->rank = rank; c->world = nranks;
    c->sent.assign((size_t)nranks, 0); c->received.assign((size_t)nranks, 0);
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
ncclResult_t ncclCommAbort(ncclComm_t c) { delete c; return ncclSuccess; }
ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t *e) { *e = ncclSuccess; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) { case ncclSuccess: return "no error"; case ncclSystemError: return "system error (simt stand-in: message file / time limit)";
                 case ncclInvalidArgument: return "invalid argument (simt stand-in: the two sides of a transfer disagree about its size)";
                 case ncclInvalidUsage: return "invalid usage"; default: return "error (simt stand-in for RCCL)"; }
}
ncclResult_t ncclGroupStart() { return ncclSuccess; }      // (operations are queued per communicator: see ncclSend / ncclRecv)
static thread_local std::vector<fake_comm *> g_open;
static ncclResult_t flush(fake_comm *c)
{
    ncclResult_t r = ncclSuccess;
    for (const auto &o : c->group) if (o.send && r == ncclSuccess) r = put(c, o.peer, o.buf, o.bytes);       // every send first: nobody waits for a receiver
    for (const auto &o : c->group) if (!o.send && r == ncclSuccess) r = get(c, o.peer, o.buf, o.bytes);
    c->group.clear();
    return r;
}
ncclResult_t ncclGroupEnd()
{
    ncclResult_t r = ncclSuccess;
    for (fake_comm *c : g_open) { const ncclResult_t e = flush(c); if (r == ncclSuccess) r = e; }
    g_open.clear();
    return r;
}
static ncclResult_t enqueue(fake_comm *c, bool send, void *buf, size_t bytes, int peer)
{
    if (!c || peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    c->group.push_back({send, buf, bytes, peer});
    bool known = false;
    for (fake_comm *o : g_open) known |= o == c;
    if (!known) g_open.push_back(c);
    return ncclSuccess;
}
// (the library always brackets its sends and receives with ncclGroupStart / ncclGroupEnd; outside a group the real RCCL would block pairwise)
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, void *) { return enqueue(c, true, const_cast<void *>(buf), count * type_bytes(t), peer); }
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, void *) { return enqueue(c, false, buf, count * type_bytes(t), peer); }

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, void *)
{
    const size_t bytes = count * type_bytes(t);
    std::vector<char> mine(static_cast<const char *>(send), static_cast<const char *>(send) + bytes);     // in place is allowed: sendbuff = recvbuff + rank * count
    for (int p = 0; p < c->world; ++p) if (p != c->rank) { const ncclResult_t r = put(c, p, mine.data(), bytes); if (r != ncclSuccess) return r; }
    memcpy(static_cast<char *>(recv) + (size_t)c->rank * bytes, mine.data(), bytes);
    for (int p = 0; p < c->world; ++p) if (p != c->rank) { const ncclResult_t r = get(c, p, static_cast<char *>(recv) + (size_t)p * bytes, bytes); if (r != ncclSuccess) return r; }
    return ncclSuccess;
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, void *)
{
    if (t != 8 || op != 0) return ncclInvalidArgument;                    // double, sum: the only form the library uses
    const size_t bytes = count * 8;
    std::vector<std::vector<double>> all((size_t)c->world, std::vector<double>(count));
    memcpy(all[(size_t)c->rank].data(), send, bytes);
    for (int p = 0; p < c->world; ++p) if (p != c->rank) { const ncclResult_t r = put(c, p, all[(size_t)c->rank].data(), bytes); if (r != ncclSuccess) return r; }
    for (int p = 0; p < c->world; ++p) if (p != c->rank) { const ncclResult_t r = get(c, p, all[(size_t)p].data(), bytes); if (r != ncclSuccess) return r; }
    double *out = static_cast<double *>(recv);
    for (size_t i = 0; i < count; ++i) {                                   // rank order; RCCL promises no order — the library's halo form needs none (one
        double acc = all[0][i];                                            // contribution per element, -0.0 everywhere else)
        for (int p = 1; p < c->world; ++p) acc = acc + all[(size_t)p][i];
        out[i] = acc;
    }
    return ncclSuccess;
}
}
