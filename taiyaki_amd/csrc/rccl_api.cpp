// rccl_api.cpp -- the data-parallel gradient all-reduce straight on RCCL, behind the C ABI
// (include/taiyaki_amd_flipflop.h, "multi-GPU").
//
// Replaces the collective of the reference's DistributedDataParallel wrap
// (bin/train_flipflop.py:384-397: the gradient all-reduce inside backward) for callers that do
// not go through torch.distributed: one communicator per process (one process per GPU), one
// ncclAllReduce(SUM) of the flat fp32 gradient arena per optimiser step, enqueued on the HIP
// stream the caller names -- a side stream, so that it overlaps the RNN backward of the earlier
// layers; xGMI is point-to-point, RCCL picks ring / tree per message size.
//
// Built as its OWN shared library (libtaiyaki_amd_rccl.so): a process that already carries an
// RCCL (PyTorch bundles one) must not get a second copy through the kernel library.  The Python
// trainers use torch.distributed's ProcessGroupNCCL -- the same RCCL calls; this library is what
// a C / C++ host (or a Taiyaki build without torch.distributed) would bind.
#include <errno.h>
#include <hip/hip_runtime.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <fcntl.h>
#include <poll.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/socket.h>
#include <time.h>
#include <unistd.h>

#include <vector>

#include "../../include/taiyaki_amd_flipflop.h"

extern "C" {

size_t tk_rccl_unique_id_bytes(void) { return sizeof(ncclUniqueId); }

// rank 0 creates the id and hands its bytes to the other ranks (file, socket, MPI: the caller's
// business -- the reference uses a TCP store at MASTER_ADDR:MASTER_PORT, train_flipflop.py:255-268)
int tk_rccl_unique_id(void *id_out, size_t bytes) {
    if (id_out == nullptr || bytes < sizeof(ncclUniqueId)) return TK_ERR_BAD_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return TK_ERR_LAUNCH;
    memcpy(id_out, &id, sizeof(id));
    return TK_OK;
}

// ---------------------------------------------------------------------------
// rendezvous on plain sockets (the reference: torch's TCP store at MASTER_ADDR:MASTER_PORT)
// ---------------------------------------------------------------------------
namespace {
constexpr uint32_t RV_MAGIC = 0x56524b54u;      // "TKRV"
struct RvHello { uint32_t magic, version, rank, nranks; };
struct RvHead { uint32_t magic, bytes; };

long long now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
}
// all `n` bytes out / in before `deadline`, or false
bool io_all(int fd, void *p, size_t n, bool writing, long long deadline) {
    char *c = static_cast<char *>(p);
    while (n > 0) {
        const long long left = deadline - now_ms();
        if (left <= 0) return false;
        pollfd pf{fd, (short)(writing ? POLLOUT : POLLIN), 0};
        const int pr = poll(&pf, 1, (int)(left > 1000 ? 1000 : left));
        if (pr < 0 && errno != EINTR) return false;
        if (pr <= 0) continue;
        const ssize_t k = writing ? send(fd, c, n, MSG_NOSIGNAL) : recv(fd, c, n, 0);
        if (k == 0 && !writing) return false;       // peer closed
        if (k < 0) {
            if (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK) continue;
            return false;
        }
        c += k;
        n -= (size_t)k;
    }
    return true;
}
struct Fds {
    std::vector<int> v;
    ~Fds() {
        for (int fd : v)
            if (fd >= 0) close(fd);
    }
};

// listening socket on `addr` (null / unbindable: the wildcard address -- torch's store binds it too, and MASTER_ADDR may be a
// name that resolves to an address this host does not own behind a NAT or in a container)
static int rv_listen(const char *addr, const char *ports, int backlog) {
    for (int pass = 0; pass < 2; ++pass) {
        addrinfo hints{}, *res = nullptr;
        hints.ai_family = AF_UNSPEC;
        hints.ai_socktype = SOCK_STREAM;
        hints.ai_flags = AI_PASSIVE;
        if (getaddrinfo(pass == 0 ? addr : nullptr, ports, &hints, &res) != 0 || res == nullptr) continue;
        int ls = -1;
        for (addrinfo *ai = res; ai != nullptr && ls < 0; ai = ai->ai_next) {
            ls = socket(ai->ai_family, ai->ai_socktype, ai->ai_protocol);
            if (ls < 0) continue;
            const int one = 1;
            (void)setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
            if (bind(ls, ai->ai_addr, ai->ai_addrlen) != 0 || listen(ls, backlog) != 0) {
                close(ls);
                ls = -1;
            }
        }
        freeaddrinfo(res);
        if (ls >= 0) return ls;
    }
    return -1;
}

int rv_serve(const char *addr, int port, int nranks, const void *buf, size_t bytes, long long deadline) {
    char ports[16];
    snprintf(ports, sizeof(ports), "%d", port);
    Fds fds;
    const int ls = rv_listen(addr, ports, nranks + 8);
    if (ls < 0) return TK_ERR_LAUNCH;
    fds.v.push_back(ls);
    (void)fcntl(ls, F_SETFL, fcntl(ls, F_GETFL, 0) | O_NONBLOCK);
    // connections that have not said hello yet: read side by side (a silent stray one -- a port scanner, a health check --
    // costs nobody anything and is dropped after 2 s), not one after the other
    struct Pending {
        int fd;
        RvHello h;
        size_t got;
        long long until;
    };
    std::vector<Pending> pend;
    std::vector<int> peer(nranks, -1);
    int have = 0;
    auto drop = [&](int fd) {
        for (int &f : fds.v)
            if (f == fd) f = -1;
        close(fd);
    };
    while (have < nranks - 1) {
        const long long now = now_ms(), left = deadline - now;
        if (left <= 0) return TK_ERR_LAUNCH;
        std::vector<pollfd> pfs;
        pfs.push_back(pollfd{ls, POLLIN, 0});
        for (const Pending &q : pend) pfs.push_back(pollfd{q.fd, POLLIN, 0});
        const int pr = poll(pfs.data(), (nfds_t)pfs.size(), (int)(left > 250 ? 250 : left));
        if (pr < 0) {
            if (errno == EINTR) continue;
            return TK_ERR_LAUNCH;                   // (not a spin until the deadline)
        }
        const bool incoming = (pfs[0].revents & POLLIN) != 0;
        for (size_t i = 0; i < pend.size();) {
            Pending &q = pend[i];
            bool gone = now_ms() > q.until;
            if (!gone && (pfs[i + 1].revents & (POLLIN | POLLHUP | POLLERR))) {
                const ssize_t k = recv(q.fd, reinterpret_cast<char *>(&q.h) + q.got, sizeof(q.h) - q.got, 0);
                if (k > 0) q.got += (size_t)k;
                else if (k == 0 || (errno != EINTR && errno != EAGAIN && errno != EWOULDBLOCK)) gone = true;
            }
            bool taken = false;
            if (!gone && q.got == sizeof(q.h)) {
                // a well-formed hello of THIS job from a rank not seen yet; anything else -- another job on the port, a rank
                // started twice, a scanner -- is closed and ignored: it must not take the other ranks down with it
                const bool ok = q.h.magic == RV_MAGIC && q.h.version == 1 && (int)q.h.nranks == nranks && q.h.rank != 0 &&
                                (int)q.h.rank < nranks && peer[q.h.rank] < 0;
                if (ok) {
                    peer[q.h.rank] = q.fd;
                    ++have;
                    taken = true;
                } else {
                    gone = true;
                }
            }
            if (gone) drop(q.fd);
            if (gone || taken) {
                // (pfs indices follow pend: keep them aligned by erasing from both)
                pend.erase(pend.begin() + (long)i);
                pfs.erase(pfs.begin() + (long)i + 1);
            } else {
                ++i;
            }
        }
        if (incoming) {                             // (after the pass over `pend`: its entries and `pfs` stay aligned)
            const int fd = accept(ls, nullptr, nullptr);
            if (fd >= 0) {
                fds.v.push_back(fd);
                const int one = 1;
                (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
                (void)fcntl(fd, F_SETFL, fcntl(fd, F_GETFL, 0) | O_NONBLOCK);
                pend.push_back(Pending{fd, RvHello{}, 0, now_ms() + 2000});
            }
        }
    }
    for (const Pending &q : pend) drop(q.fd);
    RvHead head{RV_MAGIC, (uint32_t)bytes};
    for (int r = 1; r < nranks; ++r)
        if (!io_all(peer[r], &head, sizeof(head), true, deadline) ||
            !io_all(peer[r], const_cast<void *>(buf), bytes, true, deadline))
            return TK_ERR_LAUNCH;
    for (int r = 1; r < nranks; ++r) {
        uint32_t ack = 0;
        if (!io_all(peer[r], &ack, sizeof(ack), false, deadline) || ack != RV_MAGIC) return TK_ERR_LAUNCH;
    }
    return TK_OK;
}

// connect within the deadline: non-blocking connect + poll (a blocking connect to an unroutable address sits in the kernel's
// SYN timeout for about two minutes whatever the caller's deadline says)
static int rv_connect(const addrinfo *ai, long long deadline) {
    const int fd = socket(ai->ai_family, ai->ai_socktype, ai->ai_protocol);
    if (fd < 0) return -1;
    (void)fcntl(fd, F_SETFL, fcntl(fd, F_GETFL, 0) | O_NONBLOCK);
    if (connect(fd, ai->ai_addr, ai->ai_addrlen) == 0) return fd;
    if (errno != EINPROGRESS && errno != EINTR) {
        close(fd);
        return -1;
    }
    for (;;) {
        const long long left = deadline - now_ms();
        if (left <= 0) break;
        pollfd pf{fd, POLLOUT, 0};
        const int pr = poll(&pf, 1, (int)(left > 1000 ? 1000 : left));
        if (pr < 0 && errno != EINTR) break;
        if (pr <= 0) continue;
        int err = 0;
        socklen_t len = sizeof(err);
        if (getsockopt(fd, SOL_SOCKET, SO_ERROR, &err, &len) == 0 && err == 0) return fd;
        break;
    }
    close(fd);
    return -1;
}

int rv_join(const char *addr, int port, int rank, int nranks, void *buf, size_t bytes, long long deadline) {
    char ports[16];
    snprintf(ports, sizeof(ports), "%d", port);
    for (;;) {                                      // rank 0 may not be listening yet, or dropped us: retry until the deadline
        if (now_ms() >= deadline) return TK_ERR_LAUNCH;
        addrinfo hints{}, *res = nullptr;
        hints.ai_family = AF_UNSPEC;
        hints.ai_socktype = SOCK_STREAM;
        if (getaddrinfo(addr, ports, &hints, &res) != 0 || res == nullptr) return TK_ERR_BAD_ARG;
        int fd = -1;
        for (addrinfo *ai = res; ai != nullptr && fd < 0; ai = ai->ai_next) fd = rv_connect(ai, deadline);
        freeaddrinfo(res);
        if (fd < 0) {
            usleep(50 * 1000);
            continue;
        }
        Fds fds;
        fds.v.push_back(fd);
        const int one = 1;
        (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        RvHello h{RV_MAGIC, 1, (uint32_t)rank, (uint32_t)nranks};
        RvHead head{};
        // (a connection rank 0 accepted and then lost -- it restarted, or a predecessor of ours still held the rank's slot --
        // is tried again, not an error, while the deadline lasts; a WRONG answer is an error at once)
        if (!io_all(fd, &h, sizeof(h), true, deadline) || !io_all(fd, &head, sizeof(head), false, deadline)) {
            usleep(50 * 1000);
            continue;
        }
        if (head.magic != RV_MAGIC || head.bytes != (uint32_t)bytes) return TK_ERR_LAUNCH;
        if (!io_all(fd, buf, bytes, false, deadline)) return TK_ERR_LAUNCH;
        uint32_t ack = RV_MAGIC;
        return io_all(fd, &ack, sizeof(ack), true, deadline) ? TK_OK : TK_ERR_LAUNCH;
    }
}
}  // namespace

int tk_rendezvous_bytes(const char *addr, int port, int rank, int nranks, void *buf, size_t bytes, int timeout_ms) {
    if (addr == nullptr || buf == nullptr || port <= 0 || port > 65535 || nranks < 1 || rank < 0 || rank >= nranks ||
        bytes == 0 || bytes > (1u << 20) || timeout_ms <= 0)
        return TK_ERR_BAD_ARG;
    if (nranks == 1) return TK_OK;
    const long long deadline = now_ms() + timeout_ms;
    return rank == 0 ? rv_serve(addr, port, nranks, buf, bytes, deadline) : rv_join(addr, port, rank, nranks, buf, bytes, deadline);
}

int tk_rccl_comm_init_rendezvous(void **comm_out, const char *addr, int port, int rank, int nranks, int timeout_ms) {
    if (comm_out == nullptr) return TK_ERR_BAD_ARG;
    ncclUniqueId id;
    memset(&id, 0, sizeof(id));
    if (rank == 0 && ncclGetUniqueId(&id) != ncclSuccess) return TK_ERR_LAUNCH;
    const int rc = tk_rendezvous_bytes(addr, port, rank, nranks, &id, sizeof(id), timeout_ms);
    if (rc != TK_OK) return rc;
    return tk_rccl_comm_init(comm_out, nranks, &id, rank);
}

int tk_rccl_comm_init(void **comm_out, int nranks, const void *id_bytes, int rank) {
    if (comm_out == nullptr || id_bytes == nullptr || nranks < 1 || rank < 0 || rank >= nranks)
        return TK_ERR_BAD_ARG;
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t comm;
    if (ncclCommInitRank(&comm, nranks, id, rank) != ncclSuccess) return TK_ERR_LAUNCH;
    *comm_out = comm;
    return TK_OK;
}

// SUM over the ranks, in place, on `stream`; the 1 / nranks factor is the caller's (the trainers
// fold it into the clipping pass).  Returns when the collective is ENQUEUED.
int tk_allreduce_f32_dev(void *comm, float *buf, size_t n, void *stream) {
    if (comm == nullptr || buf == nullptr) return TK_ERR_BAD_ARG;
    if (n == 0) return TK_OK;
    const ncclResult_t rc = ncclAllReduce(buf, buf, n, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm),
                                          static_cast<hipStream_t>(stream));
    return rc == ncclSuccess ? TK_OK : TK_ERR_LAUNCH;
}

// rank `root`'s buffer to every rank (the parameter broadcast that replaces the reference's
// checkpoint-file + barrier handshake, train_flipflop.py:380-392)
int tk_broadcast_f32_dev(void *comm, float *buf, size_t n, int root, void *stream) {
    if (comm == nullptr || buf == nullptr) return TK_ERR_BAD_ARG;
    if (n == 0) return TK_OK;
    const ncclResult_t rc = ncclBroadcast(buf, buf, n, ncclFloat32, root, static_cast<ncclComm_t>(comm),
                                          static_cast<hipStream_t>(stream));
    return rc == ncclSuccess ? TK_OK : TK_ERR_LAUNCH;
}

int tk_rccl_comm_destroy(void *comm) {
    if (comm == nullptr) return TK_OK;
    return ncclCommDestroy(static_cast<ncclComm_t>(comm)) == ncclSuccess ? TK_OK : TK_ERR_LAUNCH;
}

}  // extern "C"
