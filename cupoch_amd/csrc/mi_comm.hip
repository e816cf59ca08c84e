// mi_comm.hip -- how the ranks of a sharded registration exchange their 32 sums: the node's mailbox in shared host
// memory, device inboxes over HIP IPC, in-library RCCL; set-up, self-test, choice, failure handling (mailbox.h)
// (one translation unit of libmi_icp.so; csrc/ctx.h lists them)
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include "ctx.h"
#include "mailbox_kernels.h"

using namespace mi;
using namespace mi::eng;
using host::Mat4;

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;  // (optional)
};

bool load_rccl(Rccl& r) {
    if (r.handle) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) return false;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))dlsym(r.handle, "ncclCommCount");
    return r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
}

Rccl g_rccl;

}  // namespace

namespace mi {
namespace eng {

MailArgs mail_args(const mi_icp_ctx* c) {
    MailArgs m;
    m.box = c->mail_dev;
    m.seq_dev = (uint32_t*)c->mail_state.p;
    const bool direct = c->inbox != nullptr && c->xchg == 2;
    m.inbox = direct ? c->inbox : nullptr;
    m.peers = direct ? (unsigned long long* const*)c->inbox_table.p : nullptr;
    m.rank = c->rank;
    m.nranks = c->nranks;
    static const uint32_t limit = [] { const char* e = std::getenv("MI_ICP_MAIL_SPIN_LIMIT"); const long v = e ? std::atol(e) : 0; return v > 0 ? (uint32_t)v : kMailSpinLimit; }();
    m.spin_limit = limit;
    return m;
}

// Device inboxes: nobody may free an inbox a peer's kernel could still write to.  Every rank closes what it opened
// and says so in the box; an inbox is freed once every peer has (or after 2 s: a peer that died holds no kernel).
void inbox_close(mi_icp_ctx* c) {
    if (!c->inbox) return;
    (void)hipStreamSynchronize(c->stream);
    for (int r = 0; r < c->nranks && r < kMailRanks; ++r)
        if (r != c->rank && c->inbox_peer[r]) (void)hipIpcCloseMemHandle(c->inbox_peer[r]);
    for (auto& p : c->inbox_peer) p = nullptr;
    if (c->mail_host) {
        MailBox* box = c->mail_host;
        __atomic_store_n(&box->inbox_closed[c->rank], 1u, __ATOMIC_RELEASE);
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            bool all = true;
            for (int r = 0; r < c->nranks && r < kMailRanks; ++r) all = all && __atomic_load_n(&box->inbox_closed[r], __ATOMIC_ACQUIRE) != 0u;
            if (all || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    (void)hipFree(c->inbox);
    c->inbox = nullptr;
    (void)hipGetLastError();
}

// ... set up after the box itself (every rank is attached): inbox, handle into the box, wait for the peers', open
// them.  All ranks end in the same mode: a rank that fails says so in the box before the others look.
bool inbox_open(mi_icp_ctx* c, MailBox* box, int nranks, int rank, const std::function<bool()>& late) {
    auto wait_all = [&](uint32_t state) {
        for (;;) {
            bool all = true;
            for (int r = 0; r < nranks; ++r) all = all && __atomic_load_n(&box->inbox_state[r], __ATOMIC_ACQUIRE) >= state;
            if (all) return true;
            if (late()) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    };
    auto give_up = [&] { __atomic_store_n(&box->device_failed, 1u, __ATOMIC_RELEASE); };
    const size_t bytes = kMailInboxWords * sizeof(unsigned long long);
    void* mine = nullptr;
    if (hipExtMallocWithFlags(&mine, bytes, hipDeviceMallocFinegrained) != hipSuccess || hipMemset(mine, 0, bytes) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess || hipIpcGetMemHandle(&box->inbox[rank], mine) != hipSuccess) {
        (void)hipGetLastError();
        give_up();
    }
    __atomic_store_n(&box->inbox_state[rank], 1u, __ATOMIC_RELEASE);
    if (!wait_all(1u)) give_up();
    bool ok = __atomic_load_n(&box->device_failed, __ATOMIC_ACQUIRE) == 0u;
    if (ok) {
        for (int r = 0; r < nranks && ok; ++r) {
            if (r == rank) {
                c->inbox_peer[r] = (unsigned long long*)mine;
            } else {
                void* p = nullptr;
                if (hipIpcOpenMemHandle(&p, box->inbox[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                    (void)hipGetLastError();
                    give_up();
                    ok = false;
                } else {
                    c->inbox_peer[r] = (unsigned long long*)p;
                }
            }
        }
    }
    __atomic_store_n(&box->inbox_state[rank], 2u, __ATOMIC_RELEASE);
    if (!wait_all(2u)) give_up();
    ok = __atomic_load_n(&box->device_failed, __ATOMIC_ACQUIRE) == 0u;
    // The table of everybody's inboxes goes up BEFORE the mode is final: a rank whose upload fails -- or that ran out of
    // time in the wait above while a peer had already passed it -- says so, and nobody decides before every rank has
    // reached state 3 (table uploaded or given up).  Until round 6 the decision was read after state 2 and a late failure
    // left the ranks posting into different media, every exchange running into its 10-s timeout.
    unsigned long long** table = nullptr;
    if (ok && (ensure(c, c->inbox_table, kMailRanks, &table) != MI_ICP_OK ||
               hipMemcpy(table, c->inbox_peer, sizeof(c->inbox_peer), hipMemcpyHostToDevice) != hipSuccess)) {
        (void)hipGetLastError();
        give_up();
    }
    __atomic_store_n(&box->inbox_state[rank], 3u, __ATOMIC_RELEASE);
    if (!wait_all(3u)) give_up();  // (a peer that never arrives: this rank falls back; so do the others, once they see the flag or run out of time themselves)
    ok = __atomic_load_n(&box->device_failed, __ATOMIC_ACQUIRE) == 0u;
    if (!ok) {
        for (int r = 0; r < nranks; ++r)
            if (r != rank && c->inbox_peer[r]) (void)hipIpcCloseMemHandle(c->inbox_peer[r]);
        for (auto& p : c->inbox_peer) p = nullptr;
        __atomic_store_n(&box->inbox_closed[rank], 1u, __ATOMIC_RELEASE);
        if (mine) (void)hipFree(mine);
        (void)hipGetLastError();
        return false;
    }
    c->inbox = (unsigned long long*)mine;
    return true;
}

void mailbox_close(mi_icp_ctx* c) {
    inbox_close(c);
    if (c->mail_host) {
        (void)hipHostUnregister(c->mail_host);
        (void)munmap(c->mail_host, c->mail_bytes);
    }
    // (the name is rank 0's to remove, and only while it still refers to this box: once every rank has
    // attached rank 0 unlinks it at once, so that no later job -- or crash -- finds it)
    if (c->mail_linked && !c->mail_name.empty()) (void)shm_unlink(c->mail_name.c_str());
    c->mail_linked = false;
    c->mail_host = c->mail_dev = nullptr;
    c->mail_name.clear();
    c->xchg = c->comm ? 3 : 0;
    c->tune_epoch = 0;
}

static long mail_attach_timeout_ms() {
    static const long v = [] { const char* e = std::getenv("MI_ICP_MAIL_ATTACH_MS"); const long t = e ? std::atol(e) : 0; return t > 0 ? t : 30000L; }();
    return v;
}

// Rank 0 creates and zeroes the box and waits until every other rank has mapped AND registered it with
// HIP (`attached`), then declares it in use (`go`) and removes the name.  The others open the name, wait for
// `ready`, refuse a box that is in use already (a leftover of another job under the same name: its `go` is
// set -- they retry until rank 0 has replaced it), register, attach and wait for `go`.  All ranks of a job
// pass the same name.  MI_ICP_MAIL_ATTACH_MS: how long anybody waits (default 30 s).
int mailbox_open(mi_icp_ctx* c, const std::string& name, int nranks, int rank) {
    mailbox_close(c);
    if (nranks > kMailRanks) return fail(c, MI_ICP_ERR_COMM, "mailbox: %d ranks (at most %d)", nranks, kMailRanks);
    const size_t bytes = (sizeof(MailBox) + 4095) / 4096 * 4096;
    const auto t0 = std::chrono::steady_clock::now();
    const auto late = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(mail_attach_timeout_ms()); };
    MailBox* box = nullptr;
    void* dev = nullptr;
    auto drop = [&](void* p) {
        if (dev) (void)hipHostUnregister(p);
        dev = nullptr;
        (void)munmap(p, bytes);
    };
    auto map_fd = [&](int fd) -> void* {
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        return p == MAP_FAILED ? nullptr : p;
    };
    auto reg = [&](void* p) {
        if (hipHostRegister(p, bytes, hipHostRegisterMapped) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostUnregister(p);
            dev = nullptr;
            return false;
        }
        return true;
    };
    if (rank == 0) {
        (void)shm_unlink(name.c_str());  // a stale box of a crashed job
        int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) {
            if (fd >= 0) close(fd);
            (void)shm_unlink(name.c_str());
            return fail(c, MI_ICP_ERR_COMM, "mailbox: cannot create shared memory %s", name.c_str());
        }
        void* p = map_fd(fd);
        if (!p || !reg(p)) {
            if (p) (void)munmap(p, bytes);
            (void)shm_unlink(name.c_str());
            return fail(c, MI_ICP_ERR_COMM, "mailbox: cannot map / register shared memory %s", name.c_str());
        }
        box = (MailBox*)p;
        std::memset(p, 0, bytes);
        box->nranks = (uint32_t)nranks;
        // device inboxes (mailbox.h) are set up next to the box unless MI_ICP_MAILBOX=host ON RANK 0 says not to; they
        // are USED when rank 0's MI_ICP_MAILBOX=device or mi_icp_comm_autotune finds them faster
        const char* mode = std::getenv("MI_ICP_MAILBOX");
        box->device_mode = (mode && std::strcmp(mode, "host") == 0) ? 0u : 1u;
        box->use_device = (mode && std::strcmp(mode, "device") == 0) ? 1u : 0u;
        __atomic_store_n(&box->ready, 1u, __ATOMIC_RELEASE);
        while (__atomic_load_n(&box->attached, __ATOMIC_ACQUIRE) != (uint32_t)(nranks - 1)) {
            if (late()) {
                drop(p);
                (void)shm_unlink(name.c_str());
                return fail(c, MI_ICP_ERR_COMM, "mailbox: not all of the %d other ranks attached to %s in time", nranks - 1, name.c_str());
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        __atomic_store_n(&box->go, 1u, __ATOMIC_RELEASE);
        (void)shm_unlink(name.c_str());  // every rank holds its mapping: the name has done its job
    } else {
        for (;;) {
            if (late()) return fail(c, MI_ICP_ERR_COMM, "mailbox: no usable shared memory %s appeared in time", name.c_str());
            int fd = shm_open(name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) {
                if (fd >= 0) close(fd);
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
                continue;
            }
            void* p = map_fd(fd);
            if (!p) return fail(c, MI_ICP_ERR_COMM, "mailbox: mmap failed");
            box = (MailBox*)p;
            bool usable = false;
            while (!late()) {
                if (__atomic_load_n(&box->go, __ATOMIC_ACQUIRE) != 0u) break;            // in use: not ours
                if (__atomic_load_n(&box->ready, __ATOMIC_ACQUIRE) == 1u) {
                    usable = true;
                    break;
                }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            if (!usable || __atomic_load_n(&box->go, __ATOMIC_ACQUIRE) != 0u) {
                (void)munmap(p, bytes);
                box = nullptr;
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
                continue;
            }
            if (box->nranks != (uint32_t)nranks) {
                const uint32_t made_for = box->nranks;
                (void)munmap(p, bytes);
                return fail(c, MI_ICP_ERR_COMM, "mailbox: %s was made for %u ranks, not %d", name.c_str(), made_for, nranks);
            }
            if (!reg(p)) {
                (void)munmap(p, bytes);
                return fail(c, MI_ICP_ERR_COMM, "mailbox: hipHostRegister failed");
            }
            (void)__atomic_fetch_add(&box->attached, 1u, __ATOMIC_ACQ_REL);
            // While waiting for `go`: is the NAME still this box?  A crashed job's leftover (ready, never started) under a
            // reused name looks like ours; rank 0 replaces it (unlink + create), after which the name leads to another
            // inode -- this mapping is then dropped and the name opened again (the wait used to run into the
            // attach time-out, and rank 0's with it).
            bool replaced = false;
            for (int polls = 0; __atomic_load_n(&box->go, __ATOMIC_ACQUIRE) != 1u; ++polls) {
                if (late()) {  // (e.g. the box was a crashed job's and rank 0 never came)
                    drop(p);
                    return fail(c, MI_ICP_ERR_COMM, "mailbox: rank 0 did not start %s in time", name.c_str());
                }
                if (polls % 100 == 99) {
                    struct stat now;
                    const int fd2 = shm_open(name.c_str(), O_RDWR, 0600);
                    const bool other = fd2 >= 0 && fstat(fd2, &now) == 0 && (now.st_ino != st.st_ino || now.st_dev != st.st_dev);
                    if (fd2 >= 0) close(fd2);
                    if (other && __atomic_load_n(&box->go, __ATOMIC_ACQUIRE) != 1u) {
                        replaced = true;
                        break;
                    }
                }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            if (replaced) {
                drop(p);
                box = nullptr;
                continue;
            }
            break;
        }
    }
    uint32_t* state;
    TRY(ensure(c, c->mail_state, 64, &state));
    HIPCHK(c, hipMemsetAsync(state, 0, 64 * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->mail_host = box;
    c->mail_dev = (MailBox*)dev;
    c->mail_bytes = bytes;
    c->mail_name = name;
    c->mail_linked = false;  // (rank 0 has removed the name already)
    c->comm_broken = false;
    c->nranks = nranks;
    c->rank = rank;
    if (box->device_mode) (void)inbox_open(c, box, nranks, rank, late);  // (failing that, on every rank alike: the box's own words)
    // Which path is USED comes from the box (rank 0's environment, published before `ready`), never from this rank's
    // own: ranks that disagreed would post into inboxes nobody polls.  inbox_open ends alike on every rank.
    c->xchg = (c->inbox && box->use_device != 0u) ? 2 : 1;
    return MI_ICP_OK;
}

// A failed exchange leaves the ranks' exchange counters apart: whatever they post from now on could be taken
// for another exchange's.  The mailbox is given up and every call that would exchange fails until the
// communicator has been destroyed / initialised again.
int comm_failed(mi_icp_ctx* c, const char* what) {
    mailbox_close(c);
    c->comm_broken = true;
    c->loop_active = false;
    return fail(c, MI_ICP_ERR_COMM, "%s; the communicator is void: destroy it and initialise a new one", what);
}

int comm_usable(mi_icp_ctx* c) {
    if (c->comm_broken)
        return fail(c, MI_ICP_ERR_COMM, "the communicator is void after a failed exchange (timed out): destroy it and initialise a new one");
    return MI_ICP_OK;
}

int allreduce_system(mi_icp_ctx* c) {
    TRY(comm_usable(c));
    if (mail_on(c)) {  // one-shot exchange through the mailbox
        int32_t* state = (int32_t*)c->mail_state.p;
        mail_allreduce_kernel<<<1, 64, 0, c->stream>>>(mail_args(c), (double*)c->sys_dev.p, state + 1);
        KCHK(c);
        return MI_ICP_OK;
    }
    if (!c->comm) return MI_ICP_OK;
    double* sys = (double*)c->sys_dev.p;
    ncclResult_t r = g_rccl.AllReduce(sys, sys, kSysSize, ncclDouble, ncclSum, c->comm, c->stream);
    if (r != ncclSuccess) return fail(c, MI_ICP_ERR_COMM, "ncclAllReduce failed (%d)", (int)r);
    return MI_ICP_OK;
}


void comm_release(mi_icp_ctx* c) {
    mailbox_close(c);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
}

}  // namespace eng
}  // namespace mi

extern "C" {

// ---------------------------------------------------------------------------
int mi_icp_comm_unique_id(char* id128) {
    if (!id128) return MI_ICP_ERR_INVALID;
    if (!load_rccl(g_rccl)) return MI_ICP_ERR_COMM;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return MI_ICP_ERR_COMM;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, 128);
    return MI_ICP_OK;
}

int mi_icp_comm_init(mi_icp_ctx* c, const char* id128, int nranks, int rank) {
    TRY(check_ctx(c));
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, MI_ICP_ERR_INVALID, "comm_init: bad arguments");
    if (!load_rccl(g_rccl)) return fail(c, MI_ICP_ERR_COMM, "librccl could not be loaded: %s", dlerror());
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    if (c->comm) {
        g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    // ncclCommInitRank blocks until EVERY rank has joined; one that never does (a crashed peer, a bootstrap socket the
    // container's network does not route) would hold the caller forever -- and a scaling run with it, although the
    // node's mailbox needs no RCCL at all.  So the call runs on a helper thread that owns nothing but its result, and
    // the caller waits MI_ICP_COMM_INIT_MS (default 120 s; <= 0: for ever) for it: past that the communicator is given
    // up (the thread is left behind, blocked; it touches nothing of this context, and destroys the communicator itself
    // should it still form), the call fails with MI_ICP_ERR_COMM and the caller may go on with mi_icp_comm_init_local.
    // The deadline covers the SYMMETRIC failure -- no rank gets a communicator.  Should one rank give up a moment before
    // its peers' calls return, those peers go on to the agreement all-reduce below without it and wait there: callers
    // that cannot rule that out run this call beside their main path under their own watchdog, as bench.py does
    // (rccl_beside), and use mi_icp_comm_init_local for the exchange itself.
    struct InitResult {
        std::mutex m;
        std::condition_variable cv;
        bool done = false;
        bool abandoned = false;  // the caller has given up: a communicator that still forms is the thread's to destroy
        ncclResult_t r = ncclSuccess;
        ncclComm_t comm = nullptr;
    };
    static const long init_ms = [] { const char* e = std::getenv("MI_ICP_COMM_INIT_MS"); return e ? std::atol(e) : 120000L; }();
    auto res = std::make_shared<InitResult>();
    {
        const int device = c->device;
        std::thread([res, device, nranks, id, rank] {
            ncclComm_t comm = nullptr;
            ncclResult_t r = (hipSetDevice(device) == hipSuccess) ? g_rccl.CommInitRank(&comm, nranks, id, rank) : ncclUnhandledCudaError;
            std::lock_guard<std::mutex> g(res->m);
            if (res->abandoned) {  // (nobody will ever look at the result: do not leak the communicator)
                if (r == ncclSuccess && comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm);
                return;
            }
            res->r = r;
            res->comm = comm;
            res->done = true;
            res->cv.notify_all();
        }).detach();
    }
    {
        std::unique_lock<std::mutex> g(res->m);
        if (init_ms > 0) {
            if (!res->cv.wait_for(g, std::chrono::milliseconds(init_ms), [&] { return res->done; })) {
                res->abandoned = true;
                return fail(c, MI_ICP_ERR_COMM, "ncclCommInitRank did not return within %ld ms (MI_ICP_COMM_INIT_MS): given up", init_ms);
            }
        } else {
            res->cv.wait(g, [&] { return res->done; });
        }
    }
    const ncclResult_t r = res->r;
    c->comm = res->comm;
    if (r != ncclSuccess) {
        c->comm = nullptr;
        return fail(c, MI_ICP_ERR_COMM, "ncclCommInitRank failed (%d)", (int)r);
    }
    c->nranks = nranks;
    c->rank = rank;
    c->xchg = 3;
    // One node: the per-iteration exchange goes through the mailbox (mailbox.h) instead of an
    // ncclAllReduce launch; the communicator stays for whatever the mailbox cannot do.  The box is named
    // after the job's unique id.  MI_ICP_NO_MAILBOX=1, more than 16 ranks or a failed set-up: RCCL only.
    const bool no_mailbox = std::getenv("MI_ICP_NO_MAILBOX") != nullptr;  // (read at every call: a caller may fall back)
    c->comm_broken = false;
    if (!no_mailbox && nranks > 1 && nranks <= kMailRanks) {
        unsigned long long h = 1469598103934665603ull;  // FNV-1a of the id
        for (int i = 0; i < 128; ++i) h = (h ^ (unsigned char)id128[i]) * 1099511628211ull;
        char name[64];
        std::snprintf(name, sizeof(name), "/mi_icp_%016llx", h);
        const int opened = mailbox_open(c, name, nranks, rank) == MI_ICP_OK ? 1 : 0;  // (c->err says why not; not fatal)
        // The ranks must AGREE on how they exchange: one that could not open the box while its peers did would
        // wait in an ncclAllReduce nobody joins, and they for a post that never comes.  So: a min over the
        // communicator that exists by now, and the mailbox only if every rank has it.
        int32_t* flag;
        TRY(ensure(c, c->mail_state, 64, (uint32_t**)&flag));
        int32_t* agree = flag + 32;  // (behind the exchange counter and its error word)
        HIPCHK(c, hipMemcpyAsync(agree, &opened, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        ncclResult_t ar = g_rccl.AllReduce(agree, agree, 1, ncclInt32, ncclMin, c->comm, c->stream);
        int32_t all = 0;
        if (ar == ncclSuccess) {
            HIPCHK(c, hipMemcpyAsync(&all, agree, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (ar != ncclSuccess || all != 1) mailbox_close(c);
    }
    if (!c->mail_dev) c->xchg = 3;
    return MI_ICP_OK;
}

int mi_icp_comm_init_local(mi_icp_ctx* c, const char* job_name, int nranks, int rank) {
    TRY(check_ctx(c));
    if (!job_name || !job_name[0] || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(c, MI_ICP_ERR_INVALID, "comm_init_local: bad arguments");
    std::string name = "/mi_icp_";
    for (const char* p = job_name; *p && name.size() < 60; ++p)
        name += (std::isalnum((unsigned char)*p) || *p == '_' || *p == '-') ? *p : '_';
    c->nranks = nranks;
    c->rank = rank;
    c->comm_broken = false;
    // (MI_ICP_MAILBOX_SOLO: a one-rank box, to time the exchange's fixed cost on a single GPU)
    if (nranks > 1 || std::getenv("MI_ICP_MAILBOX_SOLO")) {
        const int rc = mailbox_open(c, name, nranks, rank);
        if (rc != MI_ICP_OK) {
            c->nranks = 1;
            c->rank = 0;
            return rc;
        }
    }
    return MI_ICP_OK;
}

int mi_icp_comm_kind(const mi_icp_ctx* c) {
    if (!c) return 0;
    if (mail_on(c)) return c->xchg == 2 ? 3 : 2;
    return c->comm ? 1 : 0;
}

// ---- the exchange's self-test and choice --------------------------------------------------------------------
namespace {
// Every rank's CPU writes four doubles into the box and reads everybody's: a barrier and an all-gather in one, through
// the shared mapping alone (no GPU, no RCCL).  False: a rank did not show up within the attach time-out.
bool box_gather(mi_icp_ctx* c, const double v[4], double out[kMailRanks][4]) {
    MailBox* box = c->mail_host;
    const uint32_t epoch = ++c->tune_epoch;
    const int slot = (int)(epoch & 1u);
    for (int k = 0; k < 4; ++k) box->tune_val[slot][c->rank][k] = v[k];
    __atomic_store_n(&box->tune_epoch[c->rank], epoch, __ATOMIC_RELEASE);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        bool all = true;
        for (int r = 0; r < c->nranks; ++r) all = all && __atomic_load_n(&box->tune_epoch[r], __ATOMIC_ACQUIRE) >= epoch;
        if (all) break;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(mail_attach_timeout_ms())) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    for (int r = 0; r < c->nranks; ++r)
        for (int k = 0; k < 4; ++k) out[r][k] = box->tune_val[slot][r][k];
    return true;
}
}  // namespace

int mi_icp_comm_autotune(mi_icp_ctx* c, int exchanges, double* lat_us3, int* info4) {
    TRY(check_ctx(c));
    if (!lat_us3 || !info4) return fail(c, MI_ICP_ERR_INVALID, "comm_autotune: null argument");
    TRY(comm_usable(c));
    for (int k = 0; k < 3; ++k) lat_us3[k] = -1.0;  // -1: path not available, -2: failed its self-test
    info4[0] = info4[1] = info4[2] = info4[3] = 0;
    const int n = std::min(std::max(exchanges > 0 ? exchanges : 200, 8), 10000);
    info4[2] = n;
    if (c->comm && g_rccl.CommCount) {
        int cnt = 0;
        if (g_rccl.CommCount(c->comm, &cnt) == ncclSuccess) info4[1] = cnt;
    }
    if (!c->mail_dev && !c->comm) return MI_ICP_OK;  // a single rank: nothing to choose
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double* buf;
    TRY(ensure(c, c->sys_dev, kSysSize, &buf));
    uint32_t* state;
    TRY(ensure(c, c->mail_state, 64, &state));
    int32_t* status = (int32_t*)state + 8;  // [8]: timed out, [9]: wrong totals
    int32_t* st_host = reinterpret_cast<int32_t*>(c->sys_host + 40);  // (spare words of the pinned buffer)
    const bool box = c->mail_dev != nullptr && c->mail_host != nullptr;
    bool verified = true;
    // the mailbox paths: n exchanges inside one launch
    constexpr uint32_t kSelfTestSpin = 1u << 20;  // ~2 s of polling: a path that does not deliver fails fast
    const int before = c->xchg;
    // test hook: MI_ICP_SELFTEST_BREAK="wrong:<path>" / "mute:<path>" makes the LAST rank post a wrong vector / nothing
    // on that path (tests/test_gpu_distributed.py: a path that fails is skipped on every rank alike, never fatal)
    int break_path = 0;
    bool break_mute = false;
    if (const char* e = std::getenv("MI_ICP_SELFTEST_BREAK")) {
        if (c->rank == c->nranks - 1 && (std::strncmp(e, "wrong:", 6) == 0 || std::strncmp(e, "mute:", 5) == 0)) {
            break_mute = e[0] == 'm';
            break_path = std::atoi(std::strchr(e, ':') + 1);
        }
    }
    for (int path = 1; path <= 2 && box; ++path) {
        if (path == 2 && !c->inbox) continue;
        double mine[4] = {0, 0, 0, 0}, all[kMailRanks][4];
        if (!box_gather(c, mine, all)) return comm_failed(c, "comm_autotune: the ranks did not meet at the self-test");
        c->xchg = path;
        MailArgs m = mail_args(c);
        m.spin_limit = kSelfTestSpin;
        float ms = 0.0f;
        bool ok = true;
        for (int round = 0; round < 2 && ok; ++round) {  // (a short round first: first touch of the mappings, launch skew)
            const int cnt = round == 0 ? 4 : n;
            ok = hipMemsetAsync(status, 0, 2 * sizeof(int32_t), c->stream) == hipSuccess &&
                 hipEventRecord(c->ev[0], c->stream) == hipSuccess;
            if (!ok) break;
            if (break_path == path && break_mute) {
                ok = false;  // (says nothing; its peers' kernels time out)
                break;
            }
            mail_selftest_kernel<<<1, 256, 0, c->stream>>>(m, cnt, (break_path == path) ? 0.5 : 0.0, buf, status);
            ok = hipGetLastError() == hipSuccess && hipEventRecord(c->ev[1], c->stream) == hipSuccess &&
                 hipMemcpyAsync(st_host, status, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                 hipStreamSynchronize(c->stream) == hipSuccess && hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess;
            ok = ok && st_host[0] == 0 && st_host[1] == 0;
        }
        (void)hipGetLastError();
        // this rank's figure, and its exchange counter (should a path have failed, the ranks' counters are apart)
        uint32_t seq = 0;
        (void)hipMemcpy(&seq, state, sizeof(uint32_t), hipMemcpyDeviceToHost);
        mine[0] = ok ? (double)ms * 1e3 / (double)n : 1e30;
        mine[1] = (double)seq;
        if (!box_gather(c, mine, all)) return comm_failed(c, "comm_autotune: the ranks did not meet after a self-test");
        double worst = 0.0, top = 0.0;
        for (int r = 0; r < c->nranks; ++r) {
            worst = std::max(worst, all[r][0]);
            top = std::max(top, all[r][1]);
        }
        if (worst < 1e29) {
            lat_us3[path - 1] = worst;
        } else {
            lat_us3[path - 1] = -2.0;
            // re-align: every rank continues from the same exchange number, beyond anything posted so far
            const uint32_t fresh = (uint32_t)top + 4096u;
            HIPCHK(c, hipMemcpy(state, &fresh, sizeof(uint32_t), hipMemcpyHostToDevice));
            if (!box_gather(c, mine, all)) return comm_failed(c, "comm_autotune: the ranks did not meet after a failed self-test");
        }
    }
    c->xchg = before;
    // the in-library RCCL all-reduce: n collectives, each behind a one-block kernel (the loop's step kernel stands
    // behind every all-reduce like that)
    if (c->comm) {
        bool ok = hipMemsetAsync(status, 0, 2 * sizeof(int32_t), c->stream) == hipSuccess;
        float ms = 0.0f;
        for (int round = 0; round < 2 && ok; ++round) {
            const int cnt = round == 0 ? 4 : n;
            ok = hipEventRecord(c->ev[0], c->stream) == hipSuccess;
            for (int it = 0; it < cnt && ok; ++it) {
                rccl_selftest_fill<<<1, 64, 0, c->stream>>>(buf, c->rank, c->nranks, it, status);
                ok = g_rccl.AllReduce(buf, buf, kSysSize, ncclDouble, ncclSum, c->comm, c->stream) == ncclSuccess;
            }
            rccl_selftest_fill<<<1, 64, 0, c->stream>>>(buf, c->rank, c->nranks, cnt, status);  // (checks the last one)
            ok = ok && hipEventRecord(c->ev[1], c->stream) == hipSuccess &&
                 hipMemcpyAsync(st_host, status, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                 hipStreamSynchronize(c->stream) == hipSuccess && hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess;
            ok = ok && st_host[1] == 0;
        }
        (void)hipGetLastError();
        double lat = ok ? (double)ms * 1e3 / (double)n : 1e30;
        if (box) {
            double mine[4] = {lat, 0, 0, 0}, all[kMailRanks][4];
            if (!box_gather(c, mine, all)) return comm_failed(c, "comm_autotune: the ranks did not meet after the RCCL self-test");
            for (int r = 0; r < c->nranks; ++r) lat = std::max(lat, all[r][0]);
        } else if (ok) {  // no box: the communicator itself carries the maximum
            double* d = buf;
            HIPCHK(c, hipMemcpy(d, &lat, sizeof(double), hipMemcpyHostToDevice));
            if (g_rccl.AllReduce(d, d, 1, ncclDouble, ncclMax, c->comm, c->stream) == ncclSuccess) {
                HIPCHK(c, hipMemcpyAsync(c->sys_host, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                lat = c->sys_host[0];
            }
        }
        lat_us3[2] = lat < 1e29 ? lat : -2.0;
        verified = verified && lat < 1e29;
    }
    // the fastest path that passed on EVERY rank (the figures are the maxima over the ranks: identical everywhere)
    int best = 0;
    for (int p = 1; p <= 3; ++p)
        if (lat_us3[p - 1] >= 0.0 && (best == 0 || lat_us3[p - 1] < lat_us3[best - 1])) best = p;
    if (best == 0) return comm_failed(c, "comm_autotune: no exchange path passed its self-test on every rank");
    for (int p = 1; p <= 3; ++p) verified = verified && lat_us3[p - 1] != -2.0;
    c->xchg = best;
    info4[0] = best;
    info4[3] = verified ? 1 : 0;
    return MI_ICP_OK;
}

int mi_icp_comm_destroy(mi_icp_ctx* c) {
    TRY(check_ctx(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    mailbox_close(c);
    if (c->comm) {
        g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->nranks = 1;
    c->rank = 0;
    c->comm_broken = false;
    c->xchg = 0;
    return MI_ICP_OK;
}

}  // extern "C"
