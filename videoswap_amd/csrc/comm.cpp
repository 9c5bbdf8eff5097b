// RCCL collectives of the frame-sharded long-clip mode behind the C ABI: one communicator per
// process (= per GPU), created from a 128-byte unique id that rank 0 generates and the host distributes out of band
// (torch.distributed's store, MPI, a file ... — the library does not care).
//
//   vsx_allgather_kv      K|V rows of the LOCAL frames [B, f_local*hw, 2C] -> all frames [B, P, f_local*hw, 2C]: one
//                         ncclAllGather per batch item inside a group (the frame axis is not outermost when B > 1),
//                         written straight into the buffer vsx_temporal_attention_f16 reads (no list gather, no cat);
//   vsx_allgather_f32     fp32 GroupNorm partial sums [nimg, nchunks, groups, 2] -> [P][...]; vsx_groupnorm_apply
//                         reduces them in rank order, so every rank computes bit-identical statistics;
//   vsx_allreduce_gnstats in-place fp32 sum over the ranks (the §8b form; the host uses the all-gather form above
//                         because its result does not depend on the ring order).
//
// librccl is opened with dlopen at vsx_comm_init (SONAME librccl.so.1: inside a PyTorch process this resolves to the
// copy PyTorch already loaded), so libvsx.so itself has no link-time dependency on it and single-GPU users never touch
// it.  Collectives are asynchronous on the stream passed in; the caller orders them against compute with events.
//
// A RECORDING communicator (vsx_comm_init_recording, ABI 8) stands in for librccl where no second GPU exists: rank r of P
// runs the very same entry points — argument checks, stride arithmetic, group structure, peer loop — but every
// ncclSend / ncclRecv / ncclAllGather / ncclAllReduce / local copy is appended to a log (vsx_comm_recorded) instead of
// being executed.  A test replays the P logs against each other with NCCL's matching rule (the k-th send of rank a to
// rank b meets the k-th receive of rank b from rank a) and compares the result with the layouts FrameShard promises:
// the marshalling of the C-ABI collectives is then checked for P = 2, 4, 8 on a box with one GPU or none.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <vector>

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    ncclComm_t comm = nullptr;
    int rank = -1, nranks = 0;
};

Rccl g;

// ---- recording communicator -----------------------------------------------------------------------------------------
// one record = 6 int64: op (VSX_COMM_OP_*), peer (-1: collective / local), source offset, destination offset (ELEMENTS
// from the base pointers of the entry point's two buffers; -1: not applicable), element count, element size in bytes
struct Recording {
    bool on = false;
    std::vector<int64_t> log;
    const char* src_base = nullptr;
    const char* dst_base = nullptr;
} rec;
int rec_comm_token;                 // address = the recording communicator's handle (never dereferenced)

size_t dtype_size(ncclDataType_t t) { return t == ncclFloat16 ? 2 : (t == ncclFloat32 ? 4 : 1); }

void record(int64_t op, int64_t peer, const void* src, const void* dst, size_t count, ncclDataType_t t) {
    const int64_t es = (int64_t)dtype_size(t);
    const int64_t so = src ? (static_cast<const char*>(src) - rec.src_base) / es : -1;
    const int64_t dof = dst ? (static_cast<const char*>(dst) - rec.dst_base) / es : -1;
    const int64_t r[6] = {op, peer, so, dof, (int64_t)count, es};
    rec.log.insert(rec.log.end(), r, r + 6);
}

ncclResult_t rec_send(const void* buf, size_t n, ncclDataType_t t, int peer, ncclComm_t, hipStream_t) {
    record(1 /* SEND */, peer, buf, nullptr, n, t);
    return ncclSuccess;
}
ncclResult_t rec_recv(void* buf, size_t n, ncclDataType_t t, int peer, ncclComm_t, hipStream_t) {
    record(2 /* RECV */, peer, nullptr, buf, n, t);
    return ncclSuccess;
}
ncclResult_t rec_allgather(const void* s, void* d, size_t n, ncclDataType_t t, ncclComm_t, hipStream_t) {
    record(3 /* ALLGATHER: dst block of rank q at dst + q * n */, -1, s, d, n, t);
    return ncclSuccess;
}
ncclResult_t rec_allreduce(const void* s, void* d, size_t n, ncclDataType_t t, ncclRedOp_t, ncclComm_t, hipStream_t) {
    record(4 /* ALLREDUCE (sum) */, -1, s, d, n, t);
    return ncclSuccess;
}
ncclResult_t rec_group_start() {
    const int64_t r[6] = {6, -1, -1, -1, 0, 0};
    rec.log.insert(rec.log.end(), r, r + 6);
    return ncclSuccess;
}
ncclResult_t rec_group_end() {
    const int64_t r[6] = {7, -1, -1, -1, 0, 0};
    rec.log.insert(rec.log.end(), r, r + 6);
    return ncclSuccess;
}
const char* rec_error_string(ncclResult_t) { return "recording communicator"; }

struct RecScope {                   // base pointers of the running entry point's buffers (offsets in the log)
    RecScope(const void* s, const void* d) {
        rec.src_base = static_cast<const char*>(s);
        rec.dst_base = static_cast<const char*>(d);
    }
};

int open_rccl() {
    if (g.handle) return VSX_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g.handle) break;
    }
    if (!g.handle) return vsx_fail(VSX_E_UNSUPPORTED, "comm: cannot open librccl (%s)", dlerror());
#define VSX_SYM(field, name)                                                            \
    g.field = reinterpret_cast<decltype(g.field)>(dlsym(g.handle, name));               \
    if (!g.field) return vsx_fail(VSX_E_UNSUPPORTED, "comm: librccl has no symbol %s", name)
    VSX_SYM(GetUniqueId, "ncclGetUniqueId");
    VSX_SYM(CommInitRank, "ncclCommInitRank");
    VSX_SYM(CommDestroy, "ncclCommDestroy");
    VSX_SYM(AllGather, "ncclAllGather");
    VSX_SYM(AllReduce, "ncclAllReduce");
    VSX_SYM(Send, "ncclSend");
    VSX_SYM(Recv, "ncclRecv");
    VSX_SYM(GroupStart, "ncclGroupStart");
    VSX_SYM(GroupEnd, "ncclGroupEnd");
    VSX_SYM(GetErrorString, "ncclGetErrorString");
#undef VSX_SYM
    return VSX_OK;
}

int check(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return VSX_OK;
    return vsx_fail(VSX_E_LAUNCH, "%s: %s", what, g.GetErrorString ? g.GetErrorString(r) : "rccl error");
}

}  // namespace

extern "C" int vsx_comm_unique_id(void* id128) {
    VSX_REQUIRE(id128 != nullptr, VSX_E_BADSHAPE, "comm_unique_id: null buffer");
    int rc = open_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    rc = check(g.GetUniqueId(&id), "ncclGetUniqueId");
    if (rc) return rc;
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return VSX_OK;
}

extern "C" int vsx_comm_init(int64_t rank, int64_t nranks, const void* id128) {
    VSX_REQUIRE(id128 != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, VSX_E_BADSHAPE,
                "comm_init: rank %ld of %ld", (long)rank, (long)nranks);
    VSX_REQUIRE(g.comm == nullptr, VSX_E_UNSUPPORTED, "comm_init: a communicator already exists (vsx_comm_destroy first)");
    int rc = open_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    rc = check(g.CommInitRank(&g.comm, (int)nranks, id, (int)rank), "ncclCommInitRank");
    if (rc) { g.comm = nullptr; return rc; }
    g.rank = (int)rank;
    g.nranks = (int)nranks;
    return VSX_OK;
}

extern "C" int vsx_comm_init_recording(int64_t rank, int64_t nranks) {
    VSX_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, VSX_E_BADSHAPE, "comm_init_recording: rank %ld of %ld",
                (long)rank, (long)nranks);
    VSX_REQUIRE(g.comm == nullptr, VSX_E_UNSUPPORTED,
                "comm_init_recording: a communicator already exists (vsx_comm_destroy first)");
    g.Send = rec_send;
    g.Recv = rec_recv;
    g.AllGather = rec_allgather;
    g.AllReduce = rec_allreduce;
    g.GroupStart = rec_group_start;
    g.GroupEnd = rec_group_end;
    g.GetErrorString = rec_error_string;
    g.comm = reinterpret_cast<ncclComm_t>(&rec_comm_token);
    g.rank = (int)rank;
    g.nranks = (int)nranks;
    rec.on = true;
    rec.log.clear();
    return VSX_OK;
}

extern "C" int64_t vsx_comm_recorded(int64_t* out, int64_t capacity) {
    const int64_t n = (int64_t)(rec.log.size() / 6);
    if (out == nullptr) return n;
    const int64_t take = n < capacity ? n : capacity;
    memcpy(out, rec.log.data(), (size_t)take * 6 * sizeof(int64_t));
    rec.log.erase(rec.log.begin(), rec.log.begin() + take * 6);
    return take;
}

extern "C" int64_t vsx_comm_size(void) { return g.comm ? g.nranks : 0; }
extern "C" int64_t vsx_comm_rank(void) { return g.comm ? g.rank : -1; }

extern "C" int vsx_comm_destroy(void) {
    if (!g.comm) return VSX_OK;
    int rc = VSX_OK;
    if (rec.on) {                   // recording communicator: the librccl bindings are looked up again by the next init
        rec.on = false;
        rec.log.clear();
        g = Rccl();                 // (a librccl handle opened earlier stays loaded; dlopen counts references)
    } else {
        rc = check(g.CommDestroy(g.comm), "ncclCommDestroy");
    }
    g.comm = nullptr;
    g.rank = -1;
    g.nranks = 0;
    return rc;
}

extern "C" int vsx_allgather_kv(const void* kv_local, void* kv_all, int64_t batch, int64_t elems_per_batch,
                                vsx_stream_t stream) {
    VSX_REQUIRE(g.comm != nullptr, VSX_E_UNSUPPORTED, "allgather_kv: no communicator (vsx_comm_init)");
    VSX_REQUIRE(kv_local && kv_all && batch > 0 && elems_per_batch > 0, VSX_E_BADSHAPE, "allgather_kv: bad arguments");
    const half_t* src = static_cast<const half_t*>(kv_local);
    half_t* dst = static_cast<half_t*>(kv_all);
    const RecScope scope(kv_local, kv_all);
    int rc = check(g.GroupStart(), "ncclGroupStart");
    if (rc) return rc;
    for (int64_t b = 0; b < batch; ++b) {
        rc = check(g.AllGather(src + b * elems_per_batch, dst + b * g.nranks * elems_per_batch, (size_t)elems_per_batch,
                               ncclFloat16, g.comm, (hipStream_t)stream), "ncclAllGather");
        if (rc) { (void)g.GroupEnd(); return rc; }
    }
    return check(g.GroupEnd(), "ncclGroupEnd");
}

extern "C" int vsx_allgather_f32(const float* local, float* all, int64_t count, vsx_stream_t stream) {
    VSX_REQUIRE(g.comm != nullptr, VSX_E_UNSUPPORTED, "allgather_f32: no communicator (vsx_comm_init)");
    VSX_REQUIRE(local && all && count > 0, VSX_E_BADSHAPE, "allgather_f32: bad arguments");
    const RecScope scope(local, all);
    return check(g.AllGather(local, all, (size_t)count, ncclFloat32, g.comm, (hipStream_t)stream), "ncclAllGather");
}

extern "C" int vsx_allreduce_gnstats(float* partial, int64_t count, vsx_stream_t stream) {
    VSX_REQUIRE(g.comm != nullptr, VSX_E_UNSUPPORTED, "allreduce_gnstats: no communicator (vsx_comm_init)");
    VSX_REQUIRE(partial && count > 0, VSX_E_BADSHAPE, "allreduce_gnstats: bad arguments");
    const RecScope scope(partial, partial);
    return check(g.AllReduce(partial, partial, (size_t)count, ncclFloat32, ncclSum, g.comm, (hipStream_t)stream),
                 "ncclAllReduce");
}

/* All-to-all of fp16 blocks with two-level strides (elements) on both sides: for every peer p and block (o, i),
 * o < nouter, i < ninner, `block_elems` contiguous elements travel from
 *     send + p * send_strides[0] + o * send_strides[1] + i * send_strides[2]     on this rank   to
 *     recv + r * recv_strides[0] + o * recv_strides[1] + i * recv_strides[2]     on rank p      (r = this rank).
 * frames -> sites ([B, f, P, hw/P, C] -> [B, P, f, hw/P, C]): nouter = B, ninner = f, block = hw/P * C,
 *     send strides (block, f*P*block, P*block), recv strides (f*block, P*f*block, block); sites -> frames is the
 *     same call with the two stride triples exchanged.  One group = one RCCL launch; the block to this rank itself is
 *     a device copy on the same stream. */
extern "C" int vsx_alltoall_f16(const void* send, void* recv, int64_t nouter, int64_t ninner, int64_t block_elems,
                                const int64_t* send_strides, const int64_t* recv_strides, vsx_stream_t stream) {
    VSX_REQUIRE(g.comm != nullptr, VSX_E_UNSUPPORTED, "alltoall_f16: no communicator (vsx_comm_init)");
    VSX_REQUIRE(send && recv && send_strides && recv_strides && nouter > 0 && ninner > 0 && block_elems > 0,
                VSX_E_BADSHAPE, "alltoall_f16: bad arguments");
    const half_t* src = static_cast<const half_t*>(send);
    half_t* dst = static_cast<half_t*>(recv);
    const RecScope scope(send, recv);
    for (int64_t o = 0; o < nouter; ++o)
        for (int64_t i = 0; i < ninner; ++i) {
            const half_t* s_ = src + g.rank * send_strides[0] + o * send_strides[1] + i * send_strides[2];
            half_t* d_ = dst + g.rank * recv_strides[0] + o * recv_strides[1] + i * recv_strides[2];
            if (rec.on) {
                record(5 /* LOCAL COPY */, g.rank, s_, d_, (size_t)block_elems, ncclFloat16);
                continue;
            }
            const hipError_t e = hipMemcpyAsync(d_, s_, (size_t)block_elems * sizeof(half_t), hipMemcpyDeviceToDevice,
                                                (hipStream_t)stream);
            if (e != hipSuccess) return vsx_fail(VSX_E_LAUNCH, "alltoall_f16: local copy: %s", hipGetErrorString(e));
        }
    if (g.nranks == 1) return VSX_OK;
    int rc = check(g.GroupStart(), "ncclGroupStart");
    if (rc) return rc;
    for (int p = 0; p < g.nranks && !rc; ++p) {
        if (p == g.rank) continue;
        for (int64_t o = 0; o < nouter && !rc; ++o)
            for (int64_t i = 0; i < ninner && !rc; ++i) {
                rc = check(g.Send(src + p * send_strides[0] + o * send_strides[1] + i * send_strides[2],
                                  (size_t)block_elems, ncclFloat16, p, g.comm, (hipStream_t)stream), "ncclSend");
                if (!rc)
                    rc = check(g.Recv(dst + p * recv_strides[0] + o * recv_strides[1] + i * recv_strides[2],
                                      (size_t)block_elems, ncclFloat16, p, g.comm, (hipStream_t)stream), "ncclRecv");
            }
    }
    const int rc_end = check(g.GroupEnd(), "ncclGroupEnd");
    return rc ? rc : rc_end;
}
