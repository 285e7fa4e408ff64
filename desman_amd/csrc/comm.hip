// comm.hip -- RCCL (xGMI) inside the library, without torch: the fit-record gather of the chain scheduler and the
// per-iteration exchange of a chain sharded by positions ("plain RCCL C API via the same .so", sec. 8(e)).
//
// librccl is NOT a link-time dependency: it is dlopen()ed the first time a communicator is asked for (single-GPU users never
// load it), and the handful of entry points used are declared here from the public rccl.h ABI (ncclResult_t = int,
// ncclUniqueId = 128 opaque bytes, ncclComm_t = opaque pointer, ncclSum = 0, ncclUint32 = 3, ncclFloat64 = 8).
// The reference has no communication layer at all: its fan-out is background shell jobs and its gather `cat */fit.txt`
// (scripts/runDesman.sh:15-21, complete_example/README.md:626-627).
#include "dsm_host.h"

#include <dlfcn.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace {

typedef void *nccl_comm_t;
struct nccl_uid { char internal[128]; };
enum { NCCL_SUM = 0, NCCL_MAX = 2, NCCL_UINT32 = 3, NCCL_FLOAT64 = 8 };

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(nccl_uid *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return DSM_OK;
    // DESMAN_HIP_RCCL: an explicit library (e.g. the one a host application already loaded); else the loader's search path, then /opt/rocm
    const char *env = getenv("DESMAN_HIP_RCCL");
    const char *cands[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    std::string tried;
    for (const char *p : cands) {
        if (!p || !*p) continue;
        h = dlopen(p, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        tried += std::string(tried.empty() ? "" : "; ") + dlerror();
    }
    if (!h) { dsm_set_error("RCCL not available: %s", tried.c_str()); return DSM_ERR_UNSUPPORTED; }
#define SYM(field, name)                                                                                    \
    do {                                                                                                    \
        *(void **)(&g_rccl.field) = dlsym(h, name);                                                         \
        if (!g_rccl.field) { dsm_set_error("RCCL: symbol %s missing", name); dlclose(h); return DSM_ERR_UNSUPPORTED; } \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce"); SYM(AllGather, "ncclAllGather"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString"); SYM(GetVersion, "ncclGetVersion");
#undef SYM
    g_rccl.h = h;
    return DSM_OK;
}

}  // namespace

#define NCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        int _r = (expr);                                                                                      \
        if (_r != 0) { dsm_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); return DSM_ERR_COMM; } \
    } while (0)

struct dsm_comm {
    nccl_comm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;       // the communicator's own stream (gathers, barriers); the sharded chain uses the context's
    double *buf = nullptr;              // device staging of the host-side collectives
    size_t buf_cap = 0;
};

static int comm_buf(dsm_comm *m, size_t n)
{
    if (n <= m->buf_cap) return DSM_OK;
    if (m->buf) { (void)hipFree(m->buf); m->buf = nullptr; m->buf_cap = 0; }
    hipError_t e = hipMalloc((void **)&m->buf, n * sizeof(double));
    if (e != hipSuccess) { dsm_set_error("hipMalloc(%zu B) failed: %s", n * sizeof(double), hipGetErrorString(e)); return DSM_ERR_NOMEM; }
    m->buf_cap = n;
    return DSM_OK;
}

extern "C" int dsm_comm_unique_id(void *id128)
{
    if (!id128) { dsm_set_error("dsm_comm_unique_id: null buffer"); return DSM_ERR_ARG; }
    int r = load_rccl();
    if (r != DSM_OK) return r;
    nccl_uid id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return DSM_OK;
}

extern "C" int dsm_comm_create(dsm_comm **out, const void *id128, int rank, int world, int device)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) { dsm_set_error("dsm_comm_create: bad arguments (rank %d of %d)", rank, world); return DSM_ERR_ARG; }
    int r = load_rccl();
    if (r != DSM_OK) return r;
    HIP_TRY(hipSetDevice(device));
    dsm_comm *m = new dsm_comm;
    m->rank = rank; m->world = world; m->device = device;
    nccl_uid id;
    memcpy(&id, id128, sizeof id);
    int rc = g_rccl.CommInitRank(&m->comm, world, id, rank);
    if (rc != 0) { dsm_set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, g_rccl.GetErrorString(rc)); delete m; return DSM_ERR_COMM; }
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { dsm_set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); (void)g_rccl.CommDestroy(m->comm); delete m; return DSM_ERR_HIP; }
    *out = m;
    return DSM_OK;
}

extern "C" int dsm_comm_destroy(dsm_comm *m)
{
    if (!m) return DSM_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) { (void)hipStreamSynchronize(m->stream); }
    if (m->comm) (void)g_rccl.CommDestroy(m->comm);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    if (m->buf) (void)hipFree(m->buf);
    delete m;
    return DSM_OK;
}

extern "C" int dsm_comm_rank(const dsm_comm *m) { return m ? m->rank : -1; }
extern "C" int dsm_comm_world(const dsm_comm *m) { return m ? m->world : -1; }

// recv[world][n] <- every rank's send[n] (host buffers): the one exchange of the chain scheduler (fit records)
extern "C" int dsm_comm_allgather_f64(dsm_comm *m, const double *send, double *recv, size_t n)
{
    if (!m || !send || !recv) { dsm_set_error("dsm_comm_allgather_f64: bad arguments"); return DSM_ERR_ARG; }
    if (n == 0) return DSM_OK;
    HIP_TRY(hipSetDevice(m->device));
    int r = comm_buf(m, n * (size_t)(m->world + 1));
    if (r != DSM_OK) return r;
    double *d_send = m->buf, *d_recv = m->buf + n;
    HIP_TRY(hipMemcpyAsync(d_send, send, n * sizeof(double), hipMemcpyHostToDevice, m->stream));
    NCCL_TRY(g_rccl.AllGather(d_send, d_recv, n, NCCL_FLOAT64, m->comm, m->stream));
    HIP_TRY(hipMemcpyAsync(recv, d_recv, n * (size_t)m->world * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return DSM_OK;
}

// data[n] <- element-wise sum (op 0) or maximum (op 1) over the ranks (host buffer, in place); n = 0: a barrier
extern "C" int dsm_comm_allreduce_f64(dsm_comm *m, double *data, size_t n, int op)
{
    if (!m || (n && !data) || (op != 0 && op != 1)) { dsm_set_error("dsm_comm_allreduce_f64: bad arguments"); return DSM_ERR_ARG; }
    HIP_TRY(hipSetDevice(m->device));
    double zero = 0.0;
    const size_t nn = n ? n : 1;
    int r = comm_buf(m, nn);
    if (r != DSM_OK) return r;
    HIP_TRY(hipMemcpyAsync(m->buf, n ? data : &zero, nn * sizeof(double), hipMemcpyHostToDevice, m->stream));
    NCCL_TRY(g_rccl.AllReduce(m->buf, m->buf, nn, NCCL_FLOAT64, op == 0 ? NCCL_SUM : NCCL_MAX, m->comm, m->stream));
    if (n) HIP_TRY(hipMemcpyAsync(data, m->buf, n * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return DSM_OK;
}

extern "C" int dsm_comm_barrier(dsm_comm *m) { return dsm_comm_allreduce_f64(m, nullptr, 0, 0); }

// the exchange of a sharded chain's iteration, ENQUEUED on the chain's stream: sum of the subset table (uint32, n_tab words; 0: skip)
// and of the 18-double vector over the ranks, in place, as one grouped RCCL call -- no host synchronisation (api.hip:
// dsm_ctx_gibbs_update_sharded_comm)
int comm_enqueue_exchange(dsm_comm *m, uint32_t *tab, size_t n_tab, double *vec, size_t n_vec, hipStream_t stream)
{
    NCCL_TRY(g_rccl.GroupStart());
    int r1 = n_tab ? g_rccl.AllReduce(tab, tab, n_tab, NCCL_UINT32, NCCL_SUM, m->comm, stream) : 0;
    int r2 = g_rccl.AllReduce(vec, vec, n_vec, NCCL_FLOAT64, NCCL_SUM, m->comm, stream);
    int r3 = g_rccl.GroupEnd();
    if (r1 || r2 || r3) { dsm_set_error("RCCL all-reduce of the shard exchange failed: %s", g_rccl.GetErrorString(r1 ? r1 : r2 ? r2 : r3)); return DSM_ERR_COMM; }
    return DSM_OK;
}

int comm_world(const dsm_comm *m) { return m->world; }
int comm_device(const dsm_comm *m) { return m->device; }
