// dist.hip — multi-GPU behind the C ABI (include/shimmer_hip.h "multi-GPU"; north star: "image tiles shard
// embarrassingly across the 8 GPUs of one node with an RCCL gather of per-tile spectral film buffers over xGMI").
//
// The reference has ONE parallel region — the rayon par_iter over 8x8 tiles of ImageTileIntegrator::render (integrator.rs:242-304) —
// and its film writes are unsynchronised because a tile owns its pixels (integrator.rs:277-295). The same ownership shards the frame
// over GPUs: the scene is replicated, each rank renders all spp-waves of its tiles with no collective, and ONE exchange at the end moves
// film rows to the root. Tiles are sharded in blocks of whole tile rows, so what a rank owns is a set of full-width row ranges of the
// row-major film: each block is one contiguous byte range, sent from the owner's device film straight into the same range of the
// root's device film. No staging buffer, no sum (ownership is exclusive; the rows a rank does not own are never written by it).
//   one process per GPU  (torchrun / MPI): RCCL — ncclGroupStart; ncclRecv per foreign block on the root / ncclSend per own block on a
//                        peer; ncclGroupEnd. Every peer -> root transfer rides that peer's own xGMI link.
//   one process, n GPUs  (shm_render_multi): one host thread + scene replica per device, hipMemcpyPeerAsync per block over xGMI.
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <chrono>
#include <memory>
#include <thread>

#include "wavefront.h"

struct DistState {
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    std::vector<ShmTile> tiles;       // Tile::tile(pixel_bounds, 8, 8) of the whole frame
    std::vector<ShmTile> my_tiles;    // this rank's shard
    int rows_per_block = 0;           // tile rows per block (rank 0's value, broadcast at shm_dist_init: the gather plan must be the same on every rank)
    ShmFilmPixel* d_scratch = nullptr;  // shm_dist_selftest only
    double* d_ctl = nullptr;          // CTL_DOUBLES doubles of device scratch for the small control collectives (status word, barrier, reductions)
    bool broken = false;              // the communicator was aborted after an error: every later collective fails fast
};
constexpr uint32_t CTL_DOUBLES = 4096;

namespace {

#define NCCL_TRY(expr)                                                                           \
    do {                                                                                         \
        ncclResult_t _r = (expr);                                                                \
        if (_r != ncclSuccess) {                                                                 \
            shm_err() = std::string(#expr) + ": " + ncclGetErrorString(_r);                      \
            return SHM_ERR_DEVICE;                                                               \
        }                                                                                        \
    } while (0)

int default_rows_per_block(uint32_t tile_rows, int world) {
    // 16 blocks per rank: measured on the C5 frame (3840x2160, S3) by rendering every rank's tile set on one GPU (tools/shard_balance.py,
    // profiles/r02_shard_balance.txt): mean / max of the per-rank times is 0.93 with 8 blocks per rank, 0.99 with 16 and with 33
    int blocks_per_rank = 16;
    if (const char* e = getenv("SHM_SHARD_BLOCKS")) { int v = atoi(e); if (v >= 1) blocks_per_rank = v; }
    return std::max(1, (int)tile_rows / (world * blocks_per_rank));
}

struct Block { int owner; size_t offset_px; size_t n_px; };  // a contiguous range of film pixels

// After an error on a communicator nothing on it can be trusted (a half-issued group, a peer that is gone): abort it so that the
// peers' pending operations fail instead of waiting forever, and make every later collective on this scene fail fast.
int comm_fail(DistState* d, const std::string& what) {
    if (d->comm) { ncclCommAbort(d->comm); d->comm = nullptr; }
    d->broken = true;
    shm_err() = what;
    return SHM_ERR_DEVICE;
}

// Waits for the render stream WITHOUT blocking inside the runtime: polls the stream and the communicator's asynchronous error state, so
// that a peer that died (its process was killed, its GPU faulted) or never arrives turns into an error code here instead of a hang.
// SHM_DIST_TIMEOUT_S (default 900) bounds the wait.
int wait_collective(ShmScene* s, DistState* d, const char* what) {
    static const double timeout_s = [] { const char* e = getenv("SHM_DIST_TIMEOUT_S"); double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 900.0; }();
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; ++spin) {
        hipError_t q = hipStreamQuery(s->stream);
        if (q == hipSuccess) return SHM_OK;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); return comm_fail(d, std::string(what) + ": " + hipGetErrorString(q)); }
        if (d->comm) {
            ncclResult_t async = ncclSuccess;
            ncclResult_t r = ncclCommGetAsyncError(d->comm, &async);
            if (r != ncclSuccess || (async != ncclSuccess && async != ncclInProgress))
                return comm_fail(d, std::string(what) + ": RCCL reports " + ncclGetErrorString(r != ncclSuccess ? r : async));
        }
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));  // the first ~2000 polls spin: small collectives finish in microseconds
        if ((spin & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
            return comm_fail(d, std::string(what) + ": no completion after " + std::to_string((int)timeout_s) + " s (a peer rank is missing or stuck); communicator aborted");
    }
}

// One small all-reduce over the communicator through the device scratch (host values in, host values out). world 1 without a communicator: identity.
int ctl_allreduce(ShmScene* s, DistState* d, double* values, uint32_t n, ncclRedOp_t op, const char* what) {
    if (d->broken) { shm_err() = std::string(what) + ": the communicator was aborted by an earlier error"; return SHM_ERR_DEVICE; }
    if (!d->comm) return SHM_OK;
    if (n > CTL_DOUBLES) { shm_err() = std::string(what) + ": more than 4096 values"; return SHM_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipMemcpyAsync(d->d_ctl, values, n * sizeof(double), hipMemcpyHostToDevice, s->stream));
    ncclResult_t r = ncclAllReduce(d->d_ctl, d->d_ctl, n, ncclDouble, op, d->comm, s->stream);
    if (r != ncclSuccess) return comm_fail(d, std::string(what) + ": ncclAllReduce: " + ncclGetErrorString(r));
    int rc = wait_collective(s, d, what);
    if (rc != SHM_OK) return rc;
    HIP_TRY(hipMemcpy(values, d->d_ctl, n * sizeof(double), hipMemcpyDeviceToHost));
    return SHM_OK;
}

// The gather plan: which film rows belong to which rank (the pixel rows of each block of tile rows)
std::vector<Block> film_blocks(const int32_t pb[4], int rows_per_block, int world) {
    std::vector<Block> out;
    const int width = pb[2] - pb[0], height = pb[3] - pb[1];
    const int block_px_rows = rows_per_block * 8;
    int b = 0;
    for (int y = 0; y < height; y += block_px_rows, ++b) {
        const int rows = std::min(block_px_rows, height - y);
        out.push_back(Block{b % world, (size_t)y * (size_t)width, (size_t)rows * (size_t)width});
    }
    return out;
}

// this rank's tiles under d->rows_per_block
int reshard(ShmScene* s, DistState* d) {
    const int32_t* pb = s->flat.film.pixel_bounds;
    const uint32_t tiles_per_row = (uint32_t)((pb[2] - pb[0] + 7) / 8), n = (uint32_t)d->tiles.size();
    std::vector<uint32_t> idx(std::max<uint32_t>(n, 1u));
    uint32_t n_mine = 0;
    int rc = shm_shard_tiles(n, tiles_per_row, d->rank, d->world, d->rows_per_block, idx.data(), &n_mine);
    if (rc != SHM_OK) return rc;
    d->my_tiles.resize(n_mine);
    for (uint32_t k = 0; k < n_mine; ++k) d->my_tiles[k] = d->tiles[idx[k]];
    return SHM_OK;
}

int prepare_shard(ShmScene* s, DistState* d) {
    const int32_t* pb = s->flat.film.pixel_bounds;
    const int width = pb[2] - pb[0], height = pb[3] - pb[1];
    const uint32_t tiles_per_row = (uint32_t)((width + 7) / 8), tile_rows = (uint32_t)((height + 7) / 8);
    d->tiles.resize((size_t)tiles_per_row * tile_rows);
    uint32_t n = 0;
    int rc = shm_tile_bounds(pb, 8, 8, d->tiles.data(), &n);
    if (rc != SHM_OK) return rc;
    d->tiles.resize(n);
    d->rows_per_block = default_rows_per_block(tile_rows, d->world);
    return reshard(s, d);
}

// RCCL gather of the film rows into rank 0 (`dst` on the root is its own film, or a scratch film for the self test, where `self` makes
// the root send its own rows to itself through the same group).
int gather_rccl(ShmScene* s, DistState* d, ShmFilmPixel* dst, bool self, ShmStats* stats) {
    if (d->broken || !d->comm) { shm_err() = "film gather: the communicator was aborted by an earlier error"; return SHM_ERR_DEVICE; }
    const std::vector<Block> blocks = film_blocks(s->flat.film.pixel_bounds, d->rows_per_block, d->world);
    EventPool ev{s};
    hipEvent_t e0 = ev.get(), e1 = ev.get();
    if (ev.failed) { shm_err() = "hipEventCreate failed"; return SHM_ERR_DEVICE; }
    HIP_TRY(hipEventRecord(e0, s->stream));
    uint64_t bytes = 0;
    // no early return between ncclGroupStart and ncclGroupEnd: a group left open poisons every later call on this thread
    ncclResult_t first = ncclGroupStart();
    const char* failed = first != ncclSuccess ? "ncclGroupStart" : nullptr;
    if (!failed) {
        for (const Block& b : blocks) {
            const size_t count = b.n_px * 4;  // doubles: {rgb_sum[3], weight_sum}
            double* mine = reinterpret_cast<double*>(s->d_film + b.offset_px);
            double* into = reinterpret_cast<double*>(dst + b.offset_px);
            if (b.owner == 0 && !self) continue;
            if (d->rank == b.owner) {
                ncclResult_t r = ncclSend(mine, count, ncclDouble, 0, d->comm, s->stream);
                if (r != ncclSuccess) { first = r; failed = "ncclSend"; break; }
                if (d->rank != 0) bytes += count * 8;
            }
            if (d->rank == 0) {
                ncclResult_t r = ncclRecv(into, count, ncclDouble, b.owner, d->comm, s->stream);
                if (r != ncclSuccess) { first = r; failed = "ncclRecv"; break; }
                bytes += count * 8;
            }
        }
        ncclResult_t end = ncclGroupEnd();
        if (!failed && end != ncclSuccess) { first = end; failed = "ncclGroupEnd"; }
    }
    if (failed) return comm_fail(d, std::string("film gather: ") + failed + ": " + ncclGetErrorString(first));
    HIP_TRY(hipEventRecord(e1, s->stream));
    int rc = wait_collective(s, d, "film gather");
    if (rc != SHM_OK) return rc;
    if (stats) {
        float ms = 0.0f;
        hipEventElapsedTime(&ms, e0, e1);
        stats->ms_gather += ms;
        stats->gather_bytes += bytes;
    }
    return SHM_OK;
}

}  // namespace

void wf_dist_release(ShmScene* s) {
    if (!s || !s->dist) return;
    if (s->dist->comm) ncclCommDestroy(s->dist->comm);
    if (s->dist->d_scratch) hipFree(s->dist->d_scratch);
    if (s->dist->d_ctl) hipFree(s->dist->d_ctl);
    delete s->dist;
    s->dist = nullptr;
}

extern "C" {

int shm_shard_tiles(uint32_t n_tiles, uint32_t tiles_per_row, int32_t rank, int32_t world, int32_t rows_per_block, uint32_t* idx_out, uint32_t* n_out) {
    if (!idx_out || !n_out || tiles_per_row == 0 || world < 1 || rank < 0 || rank >= world || rows_per_block < 0) {
        shm_err() = "invalid shard arguments";
        return SHM_ERR_INVALID_ARGUMENT;
    }
    const uint32_t tile_rows = (n_tiles + tiles_per_row - 1) / tiles_per_row;
    const uint32_t rpb = rows_per_block > 0 ? (uint32_t)rows_per_block : (uint32_t)default_rows_per_block(tile_rows, world);
    uint32_t n = 0;
    for (uint32_t t = 0; t < n_tiles; ++t) {
        const uint32_t block = (t / tiles_per_row) / rpb;
        if ((int32_t)(block % (uint32_t)world) == rank) idx_out[n++] = t;
    }
    *n_out = n;
    return SHM_OK;
}

int shm_dist_unique_id(uint8_t id_out[SHM_DIST_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == SHM_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id_out) { shm_err() = "id_out is null"; return SHM_ERR_INVALID_ARGUMENT; }
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return SHM_OK;
}

int shm_dist_init(ShmScene* s, int32_t rank, int32_t world, const uint8_t id_bytes[SHM_DIST_ID_BYTES]) {
    if (!s || !id_bytes || world < 1 || rank < 0 || rank >= world) { shm_err() = "invalid dist arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(s->device));
    wf_dist_release(s);
    std::unique_ptr<DistState> d(new DistState());
    d->rank = rank;
    d->world = world;
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    NCCL_TRY(ncclCommInitRank(&d->comm, world, id, rank));
    if (hipMalloc((void**)&d->d_ctl, CTL_DOUBLES * sizeof(double)) != hipSuccess) { ncclCommAbort(d->comm); shm_err() = "hipMalloc of the control scratch failed"; return SHM_ERR_OUT_OF_MEMORY; }
    int rc = prepare_shard(s, d.get());
    if (rc != SHM_OK) { ncclCommAbort(d->comm); hipFree(d->d_ctl); return rc; }
    s->dist = d.release();
    // The gather plan depends on rows_per_block, which depends on the process environment (SHM_SHARD_BLOCKS): every rank takes rank 0's
    // value, so a launcher that exports different environments to its ranks cannot make block ownership differ between them.
    double v[2] = {rank == 0 ? (double)s->dist->rows_per_block : 0.0, 1.0};
    rc = ctl_allreduce(s, s->dist, v, 2, ncclSum, "shm_dist_init: broadcast of the shard plan");
    if (rc == SHM_OK && (int)v[1] != world) { rc = SHM_ERR_DEVICE; shm_err() = "shm_dist_init: the communicator counts " + std::to_string((int)v[1]) + " ranks, expected " + std::to_string(world); }
    if (rc == SHM_OK && (int)v[0] != s->dist->rows_per_block) {
        s->dist->rows_per_block = (int)v[0];
        rc = reshard(s, s->dist);
    }
    if (rc != SHM_OK) { std::string keep = shm_err(); wf_dist_release(s); shm_err() = keep; return rc; }
    return SHM_OK;
}

int shm_dist_finalize(ShmScene* s) {
    if (!s) return SHM_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(s->device));
    wf_dist_release(s);
    return SHM_OK;
}

int shm_render_sharded(ShmScene* s, const ShmRenderParams* params, ShmStats* stats) {
    if (!s || !params) { shm_err() = "invalid render arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(s->device));
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!s->dist) {  // world = 1 without a communicator
        std::unique_ptr<DistState> d(new DistState());
        int rc = prepare_shard(s, d.get());
        if (rc != SHM_OK) return rc;
        s->dist = d.release();
    }
    DistState* d = s->dist;
    if (d->broken) { shm_err() = "shm_render_sharded: the communicator was aborted by an earlier error"; return SHM_ERR_DEVICE; }
    int rc = shm_film_clear(s);
    if (rc == SHM_OK && !d->my_tiles.empty()) rc = shm_render_device(s, params, d->my_tiles.data(), (uint32_t)d->my_tiles.size(), stats);
    if (d->world == 1) return rc;
    // A rank whose render failed (out of memory, a faulted kernel) must not leave the others waiting in ncclRecv / ncclSend: every rank
    // contributes its status to one small all-reduce first, and the gather only happens when all of them rendered.
    const std::string mine = rc != SHM_OK ? shm_err() : std::string();
    double failed = rc != SHM_OK ? 1.0 : 0.0;
    int rc2 = ctl_allreduce(s, d, &failed, 1, ncclSum, "shm_render_sharded: status exchange");
    if (rc != SHM_OK) { shm_err() = mine; return rc; }
    if (rc2 != SHM_OK) return rc2;
    if (failed != 0.0) { shm_err() = "shm_render_sharded: " + std::to_string((int)failed) + " peer rank(s) failed to render their tiles; no gather"; return SHM_ERR_DEVICE; }
    return gather_rccl(s, d, s->d_film, false, stats);
}

int shm_dist_barrier(ShmScene* s) {
    if (!s) { shm_err() = "scene is null"; return SHM_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));  // this rank's own work first: a barrier is "everything every rank enqueued is done"
    if (s->stream2) HIP_TRY(hipStreamSynchronize(s->stream2));
    if (!s->dist || !s->dist->comm && !s->dist->broken) return SHM_OK;
    double one = 1.0;
    int rc = ctl_allreduce(s, s->dist, &one, 1, ncclSum, "shm_dist_barrier");
    if (rc == SHM_OK && (int)one != s->dist->world) { shm_err() = "shm_dist_barrier: counted " + std::to_string((int)one) + " ranks"; return SHM_ERR_INTERNAL; }
    return rc;
}

int shm_dist_allreduce_f64(ShmScene* s, double* values, uint32_t n, int32_t op) {
    if (!s || (!values && n) || op < SHM_REDUCE_SUM || op > SHM_REDUCE_MIN) { shm_err() = "invalid all-reduce arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(s->device));
    if (!s->dist || n == 0) return SHM_OK;
    return ctl_allreduce(s, s->dist, values, n, op == SHM_REDUCE_SUM ? ncclSum : (op == SHM_REDUCE_MAX ? ncclMax : ncclMin), "shm_dist_allreduce_f64");
}

int shm_dist_allgather_f64(ShmScene* s, const double* mine, uint32_t n, double* all_out) {
    if (!s || !mine || !all_out || n == 0) { shm_err() = "invalid all-gather arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(s->device));
    DistState* d = s->dist;
    if (!d || !d->comm) {
        if (d && d->broken) { shm_err() = "shm_dist_allgather_f64: the communicator was aborted by an earlier error"; return SHM_ERR_DEVICE; }
        memcpy(all_out, mine, n * sizeof(double));
        return SHM_OK;
    }
    if ((uint64_t)n * (uint64_t)d->world > CTL_DOUBLES) { shm_err() = "shm_dist_allgather_f64: more than 4096 values in total"; return SHM_ERR_INVALID_ARGUMENT; }
    // in place: this rank's values sit at their final position of the receive buffer
    HIP_TRY(hipMemcpyAsync(d->d_ctl + (size_t)d->rank * n, mine, n * sizeof(double), hipMemcpyHostToDevice, s->stream));
    ncclResult_t r = ncclAllGather(d->d_ctl + (size_t)d->rank * n, d->d_ctl, n, ncclDouble, d->comm, s->stream);
    if (r != ncclSuccess) return comm_fail(d, std::string("shm_dist_allgather_f64: ncclAllGather: ") + ncclGetErrorString(r));
    int rc = wait_collective(s, d, "shm_dist_allgather_f64");
    if (rc != SHM_OK) return rc;
    HIP_TRY(hipMemcpy(all_out, d->d_ctl, (size_t)n * d->world * sizeof(double), hipMemcpyDeviceToHost));
    return SHM_OK;
}

int shm_dist_info(ShmScene* s, ShmDistInfo* out) {
    if (!out) { shm_err() = "out is null"; return SHM_ERR_INVALID_ARGUMENT; }
    memset(out, 0, sizeof(*out));
    out->world = 1;
    int v = 0;
    if (ncclGetVersion(&v) == ncclSuccess) out->rccl_version = v;
    int hv = 0;
    if (hipRuntimeGetVersion(&hv) == hipSuccess) out->hip_runtime_version = hv;
    // which shared objects this process actually mapped for the two runtimes (a host that imported another ROCm stack first — a Python
    // package with bundled libraries, say — would show up here)
    Dl_info di;
    if (dladdr((void*)&ncclGetVersion, &di) && di.dli_fname) snprintf(out->librccl_path, sizeof(out->librccl_path), "%s", di.dli_fname);
    if (dladdr((void*)&hipRuntimeGetVersion, &di) && di.dli_fname) snprintf(out->libamdhip_path, sizeof(out->libamdhip_path), "%s", di.dli_fname);
    if (s && s->dist) {
        out->rank = s->dist->rank;
        out->world = s->dist->world;
        out->rows_per_block = s->dist->rows_per_block;
        out->n_my_tiles = (uint32_t)s->dist->my_tiles.size();
        if (s->dist->comm) {
            int c = 0, dev = -1;
            if (ncclCommCount(s->dist->comm, &c) == ncclSuccess) out->rccl_ranks = c;
            if (ncclCommCuDevice(s->dist->comm, &dev) == ncclSuccess) out->rccl_device = dev;
        }
    }
    return SHM_OK;
}

int shm_device_synchronize(int32_t device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return SHM_OK;
}

int shm_dist_selftest(ShmScene* s) {
    if (!s || !s->dist || !s->dist->comm) { shm_err() = "shm_dist_selftest needs shm_dist_init"; return SHM_ERR_INVALID_ARGUMENT; }
    DistState* d = s->dist;
    if (d->rank != 0) return SHM_OK;  // only the root has a receiving side
    HIP_TRY(hipSetDevice(s->device));
    const size_t bytes = s->n_film_pixels * sizeof(ShmFilmPixel);
    if (!d->d_scratch) HIP_TRY(hipMalloc((void**)&d->d_scratch, bytes));
    HIP_TRY(hipMemsetAsync(d->d_scratch, 0xff, bytes, s->stream));
    // the root's own rows only (a world of 1 owns every block): send to self, receive into the scratch film
    const std::vector<Block> blocks = film_blocks(s->flat.film.pixel_bounds, d->rows_per_block, d->world);
    ncclResult_t first = ncclGroupStart();
    const char* failed = first != ncclSuccess ? "ncclGroupStart" : nullptr;
    if (!failed) {
        for (const Block& b : blocks) {
            if (b.owner != 0) continue;
            ncclResult_t r = ncclSend(reinterpret_cast<double*>(s->d_film + b.offset_px), b.n_px * 4, ncclDouble, 0, d->comm, s->stream);
            if (r == ncclSuccess) r = ncclRecv(reinterpret_cast<double*>(d->d_scratch + b.offset_px), b.n_px * 4, ncclDouble, 0, d->comm, s->stream);
            if (r != ncclSuccess) { first = r; failed = "ncclSend / ncclRecv"; break; }
        }
        ncclResult_t end = ncclGroupEnd();
        if (!failed && end != ncclSuccess) { first = end; failed = "ncclGroupEnd"; }
    }
    if (failed) return comm_fail(d, std::string("shm_dist_selftest: ") + failed + ": " + ncclGetErrorString(first));
    int rcw = wait_collective(s, d, "shm_dist_selftest");
    if (rcw != SHM_OK) return rcw;
    std::vector<ShmFilmPixel> a(s->n_film_pixels), c(s->n_film_pixels);
    HIP_TRY(hipMemcpy(a.data(), s->d_film, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c.data(), d->d_scratch, bytes, hipMemcpyDeviceToHost));
    for (const Block& b : blocks) {
        if (b.owner != 0) continue;
        if (memcmp(a.data() + b.offset_px, c.data() + b.offset_px, b.n_px * sizeof(ShmFilmPixel)) != 0) {
            shm_err() = "RCCL loopback of the film rows differs from the film";
            return SHM_ERR_INTERNAL;
        }
    }
    return SHM_OK;
}

int shm_render_multi(const ShmSceneDesc* desc, const int32_t* devices, int32_t n_devices, const ShmRenderParams* params, ShmFilmPixel* film_out,
                     ShmStats* stats_per_device) {
    if (!desc || !devices || n_devices < 1 || n_devices > 64 || !params || !film_out) { shm_err() = "invalid multi-device arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    const int n = n_devices;
    std::vector<ShmScene*> scenes((size_t)n, nullptr);
    std::vector<int> rcs((size_t)n, SHM_OK);
    std::vector<std::string> errs((size_t)n);
    std::vector<ShmStats> stats((size_t)n);
    for (auto& st : stats) memset(&st, 0, sizeof(st));
    // phase 1: one replica per device, each on its own host thread (scene upload and rendering run concurrently on all devices)
    auto worker = [&](int r) {
        try {
            int rc = shm_scene_create(desc, devices[r], &scenes[(size_t)r]);
            if (rc == SHM_OK) {
                ShmScene* s = scenes[(size_t)r];
                DistState d;
                d.rank = r;
                d.world = n;
                rc = prepare_shard(s, &d);
                if (rc == SHM_OK) rc = shm_film_clear(s);
                if (rc == SHM_OK && !d.my_tiles.empty()) rc = shm_render_device(s, params, d.my_tiles.data(), (uint32_t)d.my_tiles.size(), &stats[(size_t)r]);
            }
            rcs[(size_t)r] = rc;
            if (rc != SHM_OK) errs[(size_t)r] = shm_err();
        } catch (const std::exception& e) {
            rcs[(size_t)r] = SHM_ERR_INTERNAL;
            errs[(size_t)r] = e.what();
        }
    };
    {
        std::vector<std::thread> threads;
        for (int r = 0; r < n; ++r) threads.emplace_back(worker, r);
        for (auto& t : threads) t.join();
    }
    int rc = SHM_OK;
    for (int r = 0; r < n && rc == SHM_OK; ++r)
        if (rcs[(size_t)r] != SHM_OK) { rc = rcs[(size_t)r]; shm_err() = "device " + std::to_string(devices[r]) + ": " + errs[(size_t)r]; }
    // phase 2: film rows of every peer -> the first replica's film, one hipMemcpyPeerAsync per block on the owner's stream
    if (rc == SHM_OK && n > 1) {
        ShmScene* root = scenes[0];
        const int32_t* pb = root->flat.film.pixel_bounds;
        const uint32_t tile_rows = (uint32_t)((pb[3] - pb[1] + 7) / 8);
        const std::vector<Block> blocks = film_blocks(pb, default_rows_per_block(tile_rows, n), n);
        for (int r = 1; r < n; ++r) {  // peer access lets the copy engine go over xGMI directly (an error just means staged copies)
            int can = 0;
            if (devices[r] != devices[0] && hipDeviceCanAccessPeer(&can, devices[r], devices[0]) == hipSuccess && can) {
                hipSetDevice(devices[r]);
                hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
                if (e != hipSuccess) (void)hipGetLastError();  // already enabled / not supported
            }
        }
        std::vector<hipEvent_t> e0((size_t)n, nullptr), e1((size_t)n, nullptr);
        for (int r = 1; r < n && rc == SHM_OK; ++r) {
            ShmScene* s = scenes[(size_t)r];
            if (hipSetDevice(s->device) != hipSuccess) { shm_err() = "hipSetDevice failed"; rc = SHM_ERR_DEVICE; break; }
            hipEventCreate(&e0[(size_t)r]);
            hipEventCreate(&e1[(size_t)r]);
            hipEventRecord(e0[(size_t)r], s->stream);
            for (const Block& b : blocks) {
                if (b.owner != r) continue;
                const size_t bytes = b.n_px * sizeof(ShmFilmPixel);
                if (hipMemcpyPeerAsync(root->d_film + b.offset_px, root->device, s->d_film + b.offset_px, s->device, bytes, s->stream) != hipSuccess) {
                    shm_err() = "hipMemcpyPeerAsync of a film block failed";
                    rc = SHM_ERR_DEVICE;
                    break;
                }
                stats[(size_t)r].gather_bytes += bytes;
            }
            hipEventRecord(e1[(size_t)r], s->stream);
        }
        for (int r = 1; r < n; ++r) {
            ShmScene* s = scenes[(size_t)r];
            hipSetDevice(s->device);
            if (hipStreamSynchronize(s->stream) != hipSuccess && rc == SHM_OK) { shm_err() = "film gather failed"; rc = SHM_ERR_DEVICE; }
            if (e0[(size_t)r] && e1[(size_t)r]) {
                float ms = 0.0f;
                if (hipEventElapsedTime(&ms, e0[(size_t)r], e1[(size_t)r]) == hipSuccess) stats[(size_t)r].ms_gather = ms;
            }
            if (e0[(size_t)r]) hipEventDestroy(e0[(size_t)r]);
            if (e1[(size_t)r]) hipEventDestroy(e1[(size_t)r]);
        }
    }
    if (rc == SHM_OK) rc = shm_film_read(scenes[0], film_out);
    for (ShmScene* s : scenes) shm_scene_destroy(s);
    if (rc == SHM_OK && stats_per_device) memcpy(stats_per_device, stats.data(), sizeof(ShmStats) * (size_t)n);
    return rc;
}

}  // extern "C"
