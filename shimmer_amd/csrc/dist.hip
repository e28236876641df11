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

#include <memory>
#include <thread>

#include "wavefront.h"

struct DistState {
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    std::vector<ShmTile> tiles;       // Tile::tile(pixel_bounds, 8, 8) of the whole frame
    std::vector<ShmTile> my_tiles;    // this rank's shard
    int rows_per_block = 0;           // tile rows per block
    ShmFilmPixel* d_scratch = nullptr;  // shm_dist_selftest only
};

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
    std::vector<uint32_t> idx(n);
    uint32_t n_mine = 0;
    rc = shm_shard_tiles(n, tiles_per_row, d->rank, d->world, d->rows_per_block, idx.data(), &n_mine);
    if (rc != SHM_OK) return rc;
    d->my_tiles.resize(n_mine);
    for (uint32_t k = 0; k < n_mine; ++k) d->my_tiles[k] = d->tiles[idx[k]];
    return SHM_OK;
}

// RCCL gather of the film rows into rank 0 (`dst` on the root is its own film, or a scratch film for the self test, where `self` makes
// the root send its own rows to itself through the same group).
int gather_rccl(ShmScene* s, DistState* d, ShmFilmPixel* dst, bool self, ShmStats* stats) {
    const std::vector<Block> blocks = film_blocks(s->flat.film.pixel_bounds, d->rows_per_block, d->world);
    EventPool ev{s};
    hipEvent_t e0 = ev.get(), e1 = ev.get();
    if (ev.failed) { shm_err() = "hipEventCreate failed"; return SHM_ERR_DEVICE; }
    HIP_TRY(hipEventRecord(e0, s->stream));
    uint64_t bytes = 0;
    NCCL_TRY(ncclGroupStart());
    for (const Block& b : blocks) {
        const size_t count = b.n_px * 4;  // doubles: {rgb_sum[3], weight_sum}
        double* mine = reinterpret_cast<double*>(s->d_film + b.offset_px);
        double* into = reinterpret_cast<double*>(dst + b.offset_px);
        if (b.owner != 0 || self) {
            if (d->rank == b.owner) { NCCL_TRY(ncclSend(mine, count, ncclDouble, 0, d->comm, s->stream)); if (d->rank != 0) bytes += count * 8; }
            if (d->rank == 0) { NCCL_TRY(ncclRecv(into, count, ncclDouble, b.owner, d->comm, s->stream)); bytes += count * 8; }
        }
    }
    NCCL_TRY(ncclGroupEnd());
    HIP_TRY(hipEventRecord(e1, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
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
    int rc = prepare_shard(s, d.get());
    if (rc != SHM_OK) { ncclCommDestroy(d->comm); return rc; }
    s->dist = d.release();
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
    int rc = shm_film_clear(s);
    if (rc != SHM_OK) return rc;
    if (!d->my_tiles.empty()) {
        rc = shm_render_device(s, params, d->my_tiles.data(), (uint32_t)d->my_tiles.size(), stats);
        if (rc != SHM_OK) return rc;
    }
    if (d->world > 1) return gather_rccl(s, d, s->d_film, false, stats);
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
    NCCL_TRY(ncclGroupStart());
    for (const Block& b : blocks) {
        if (b.owner != 0) continue;
        NCCL_TRY(ncclSend(reinterpret_cast<double*>(s->d_film + b.offset_px), b.n_px * 4, ncclDouble, 0, d->comm, s->stream));
        NCCL_TRY(ncclRecv(reinterpret_cast<double*>(d->d_scratch + b.offset_px), b.n_px * 4, ncclDouble, 0, d->comm, s->stream));
    }
    NCCL_TRY(ncclGroupEnd());
    HIP_TRY(hipStreamSynchronize(s->stream));
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
