// Multi-GPU entry points: row-shard halo exchange and zonal partial reduction over RCCL (xGMI).
//
// The reference has no communication layer at all; its distributed semantics are dask's
// map_overlap(depth=k//2, boundary=nan) for stencils (e.g. xrspatial/slope.py:94-97) and the
// per-block-partials + combine of _stats_dask_numpy (xrspatial/zonal.py:198-259).  Here:
//   * one process per GPU, a raster sharded on the row axis, ONE exchange of k//2 rows with
//     each neighbour (ncclSend/ncclRecv grouped so the four transfers progress together over
//     the direct xGMI links), after which every stencil kernel runs with halo_top/halo_bot set;
//   * zonal partials are KB-sized: three sum all-reduces + min + max, grouped in one launch.
#include "xrs_common.h"

#include <rccl/rccl.h>

using namespace xrs;

#define XRS_NCCL(call)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return ::xrs::fail("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

namespace {
struct Comm {
    ncclComm_t nccl;
    int nranks, rank;
};
}  // namespace

extern "C" {

int xrs_comm_unique_id(void *id128) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!id128) return fail("xrs_comm_unique_id: null pointer");
    ncclUniqueId id;
    XRS_NCCL(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int xrs_comm_init_rank(void **comm, const void *id128, int nranks, int rank) {
    if (!comm || !id128) return fail("xrs_comm_init_rank: null pointer");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("xrs_comm_init_rank: bad rank %d of %d", rank, nranks);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm *c = new Comm{nullptr, nranks, rank};
    ncclResult_t r = ncclCommInitRank(&c->nccl, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail("ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    *comm = c;
    return 0;
}

int xrs_comm_info(void *comm, int *info4) {
    // what RCCL itself reports for this communicator: {ncclGetVersion code, ncclCommCount, ncclCommUserRank,
    // ncclCommCuDevice} -- bench.py --dry-rccl prints them so that a 30-second run tells an initialisation problem
    // (wrong rank count, wrong device) from a performance problem
    if (!comm || !info4) return fail("xrs_comm_info: null pointer");
    Comm *c = static_cast<Comm *>(comm);
    int version = 0, count = 0, rank = -1, dev = -1;
    XRS_NCCL(ncclGetVersion(&version));
    XRS_NCCL(ncclCommCount(c->nccl, &count));
    XRS_NCCL(ncclCommUserRank(c->nccl, &rank));
    XRS_NCCL(ncclCommCuDevice(c->nccl, &dev));
    info4[0] = version; info4[1] = count; info4[2] = rank; info4[3] = dev;
    return 0;
}

int xrs_comm_destroy(void *comm) {
    if (!comm) return 0;
    Comm *c = static_cast<Comm *>(comm);
    ncclResult_t r = ncclCommDestroy(c->nccl);
    delete c;
    if (r != ncclSuccess) return fail("ncclCommDestroy failed: %s", ncclGetErrorString(r));
    return 0;
}

int xrs_halo_exchange_f32(void *comm, float *shard_dev, int64_t rows, int64_t cols, int64_t ld, int halo,
                          void *stream) {
    if (!comm || !shard_dev) return fail("xrs_halo_exchange_f32: null pointer");
    if (halo < 0 || rows < halo || cols <= 0 || ld < cols) return fail("xrs_halo_exchange_f32: bad shape");
    Comm *c = static_cast<Comm *>(comm);
    if (halo == 0 || c->nranks == 1) return 0;
    hipStream_t s = as_stream(stream);
    const int up = c->rank - 1, down = c->rank + 1;
    // rows are `ld` apart: a contiguous block of `halo` rows is halo*ld elements (the last row's
    // padding is included only when ld > cols, which stays inside the allocation by contract).
    const size_t count = (size_t)(halo - 1) * ld + cols;
    float *top_halo = shard_dev - (int64_t)halo * ld;      // rows [-halo, 0)
    float *first_rows = shard_dev;                          // rows [0, halo)
    float *last_rows = shard_dev + (rows - halo) * ld;      // rows [rows-halo, rows)
    float *bot_halo = shard_dev + rows * ld;                // rows [rows, rows+halo)
    XRS_NCCL(ncclGroupStart());
    if (up >= 0) {
        XRS_NCCL(ncclSend(first_rows, count, ncclFloat32, up, c->nccl, s));
        XRS_NCCL(ncclRecv(top_halo, count, ncclFloat32, up, c->nccl, s));
    }
    if (down < c->nranks) {
        XRS_NCCL(ncclSend(last_rows, count, ncclFloat32, down, c->nccl, s));
        XRS_NCCL(ncclRecv(bot_halo, count, ncclFloat32, down, c->nccl, s));
    }
    XRS_NCCL(ncclGroupEnd());
    return 0;
}

/* Loop-back check of the RCCL point-to-point plumbing on ONE GPU: a grouped ncclSend/ncclRecv of `count`
 * floats from src_dev to dst_dev addressed to this rank itself.  Used by the single-GPU test-suite (the
 * real halo exchange needs >= 2 GPUs); not part of the data path. */
int xrs_comm_selftest_f32(void *comm, const float *src_dev, float *dst_dev, int64_t count, void *stream) {
    if (!comm || !src_dev || !dst_dev || count < 0) return fail("xrs_comm_selftest_f32: bad argument");
    Comm *c = static_cast<Comm *>(comm);
    hipStream_t s = as_stream(stream);
    XRS_NCCL(ncclGroupStart());
    XRS_NCCL(ncclSend(src_dev, (size_t)count, ncclFloat32, c->rank, c->nccl, s));
    XRS_NCCL(ncclRecv(dst_dev, (size_t)count, ncclFloat32, c->rank, c->nccl, s));
    XRS_NCCL(ncclGroupEnd());
    return 0;
}

int xrs_zonal_allreduce(void *comm, uint64_t *count_dev, double *sum_dev, double *sumsq_dev, void *min_dev,
                        void *max_dev, int minmax_f64, int n_zones, void *stream) {
    if (!comm) return fail("xrs_zonal_allreduce: null communicator");
    if (n_zones <= 0) return 0;
    Comm *c = static_cast<Comm *>(comm);
    hipStream_t s = as_stream(stream);     // (a 1-rank communicator still goes through RCCL: in-place no-op reduce)
    XRS_NCCL(ncclGroupStart());
    XRS_NCCL(ncclAllReduce(count_dev, count_dev, n_zones, ncclUint64, ncclSum, c->nccl, s));
    XRS_NCCL(ncclAllReduce(sum_dev, sum_dev, n_zones, ncclFloat64, ncclSum, c->nccl, s));
    XRS_NCCL(ncclAllReduce(sumsq_dev, sumsq_dev, n_zones, ncclFloat64, ncclSum, c->nccl, s));
    const ncclDataType_t mt = minmax_f64 ? ncclFloat64 : ncclFloat32;
    XRS_NCCL(ncclAllReduce(min_dev, min_dev, n_zones, mt, ncclMin, c->nccl, s));
    XRS_NCCL(ncclAllReduce(max_dev, max_dev, n_zones, mt, ncclMax, c->nccl, s));
    XRS_NCCL(ncclGroupEnd());
    return 0;
}

/* Plain typed all-reduce of a device buffer, in place (control-plane values of the sharded operators: zone-id ranges,
 * presence maps, moment triples, the benchmark's barrier and max-over-ranks).  op: 0 = sum, 1 = min, 2 = max. */
static int allreduce_typed(const char *who, void *comm, void *buf_dev, int64_t count, ncclDataType_t dt, int op, void *stream) {
    if (!comm) return fail("%s: null communicator", who);
    if (count < 0 || (count > 0 && !buf_dev)) return fail("%s: bad buffer", who);
    if (op < 0 || op > 2) return fail("%s: op must be 0 (sum), 1 (min) or 2 (max)", who);
    if (count == 0) return 0;
    Comm *c = static_cast<Comm *>(comm);
    const ncclRedOp_t ops[3] = {ncclSum, ncclMin, ncclMax};
    XRS_NCCL(ncclAllReduce(buf_dev, buf_dev, (size_t)count, dt, ops[op], c->nccl, as_stream(stream)));
    return 0;
}

int xrs_allreduce_f64(void *comm, double *buf_dev, int64_t count, int op, void *stream) {
    return allreduce_typed("xrs_allreduce_f64", comm, buf_dev, count, ncclFloat64, op, stream);
}

int xrs_allreduce_u8(void *comm, uint8_t *buf_dev, int64_t count, int op, void *stream) {
    return allreduce_typed("xrs_allreduce_u8", comm, buf_dev, count, ncclUint8, op, stream);
}

int xrs_allreduce_u64(void *comm, uint64_t *buf_dev, int64_t count, int op, void *stream) {
    return allreduce_typed("xrs_allreduce_u64", comm, buf_dev, count, ncclUint64, op, stream);
}

}  // extern "C"
