/* pfd.h — C-ABI of libpfd_hip.so: MI355X-native D8 flow-accumulation hot path.
 *
 * This is the drop-in boundary for the path
 *     D8 raster -> downstream graph -> cell ordering -> accuflux / upstream_area,
 *     Strahler order, basins, HAND
 * of Deltares/pyflwdir (reference v0.5.12).  The reference has no FFI of its own: its seam
 * is "method on FlwdirRaster validates & flattens, then calls a free function on flat
 * C-contiguous numpy arrays" (reference pyflwdir/flwdir.py:567-602, pyflwdir/pyflwdir.py:
 * 770-801, :564-599, :1485-1511).  Each entry point below replaces one of those free
 * functions and cites it.  Plain C: opaque handle, plain pointers and sizes, int status.
 *
 * Conventions
 *   - Every function returns PFD_OK (0) or a negative PFD_E* code; pfd_last_error() gives
 *     the message of the last failure on the calling thread.
 *   - Rasters are row-major (C order), linear cell index i = r*ncol + c.
 *   - `memspace` says where the caller's buffers live: PFD_HOST (the library stages them
 *     through HBM) or PFD_DEVICE (pointers into the HBM of the handle's GPU; nothing is
 *     copied).  Inputs are never modified; outputs are caller-allocated and fully written.
 *   - A handle owns all device state (normalised D8 codes, level structure, scratch) of one
 *     raster (or one row block of a raster in a multi-GPU job) on one GPU.  Handles are not
 *     thread-safe; use one per host thread / process.
 */
#ifndef PFD_H_
#define PFD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFD_ABI_VERSION 1

/* status codes */
#define PFD_OK 0
#define PFD_EINVAL (-1)      /* bad argument (shape, dtype code, NULL pointer, ...) */
#define PFD_ENODEVICE (-2)   /* no usable HIP device / device index out of range */
#define PFD_EHIP (-3)        /* a HIP runtime call failed (message has the HIP error) */
#define PFD_ENOMEM (-4)      /* device or host allocation failed */
#define PFD_EBADCODE (-5)    /* raster holds a value that is not a D8 code (core_d8._all) */
#define PFD_ENOPITS (-6)     /* raster has no pit: reference raises "no pits found" */
#define PFD_EUNSUPPORTED (-7)/* valid request outside what the HIP path implements */
#define PFD_ECOMM (-8)       /* RCCL call failed (multi-GPU) */

/* memory spaces */
#define PFD_HOST 0
#define PFD_DEVICE 1

/* element dtype codes (payloads and index exports) */
#define PFD_I32 1
#define PFD_U32 2
#define PFD_I64 3
#define PFD_F32 4
#define PFD_F64 5

/* accumulation direction: streams.accuflux (up) / streams.accuflux_ds (down) */
#define PFD_UP 0
#define PFD_DOWN 1

typedef struct pfd_raster pfd_raster;

/* ---- library / device --------------------------------------------------------------- */
int pfd_abi_version(void);
const char *pfd_last_error(void);
int pfd_device_count(int *count);
/* device memory helpers so that a ctypes caller can keep rasters resident in HBM without
 * any other GPU library (bench.py uses these). */
int pfd_malloc(int device, size_t bytes, void **ptr);
int pfd_free(int device, void *ptr);
int pfd_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes);
int pfd_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes);
int pfd_device_synchronize(int device);
/* the library caches freed HBM blocks for reuse; pfd_trim returns them to the driver
 * (device < 0: all devices) */
int pfd_trim(int device);
/* Reserve `bytes` of HBM on `device` as ONE arena that the library's working buffers (>= 1 MiB) are carved from before
 * anything else is tried: after this call a steady state never calls hipMalloc, whose latency for multi-GiB blocks is
 * unpredictable on this hardware (0.2 ms or seconds; DESIGN.md "allocation").  May be called more than once (more
 * arenas); requests that do not fit fall back to the class cache / hipMalloc.  bytes == 0 releases the device's arenas
 * that hold no live block.  Callers that time first calls (bench.py, a serving process) reserve before they start. */
int pfd_reserve(int device, size_t bytes);
/* allocator counters since the process started: [0] hipMalloc calls, [1] exact-class cache hits, [2] near-fit cache hits,
 * [3] blocks carved from reserved arenas, [4] idle cached bytes, [5] reserved bytes, [6] of them free, [7] live blocks */
int pfd_alloc_stats(int64_t out[8]);
/* free and total HBM of `device` in bytes (hipMemGetInfo): out[0] free, out[1] total.  The Python front end sizes its
 * default arena with it (pyflwdir_amd/_hip.py ensure_reserved). */
int pfd_mem_info(int device, int64_t out[2]);
/* Host <-> device traffic of the calling thread's API calls since the last call with reset != 0 (§8d "report H2D/D2H
 * separately"): out[0] bytes uploaded from PFD_HOST arguments, out[1] ms spent in those uploads (wall clock of the
 * staging calls), out[2] bytes downloaded into PFD_HOST results, out[3] ms of the downloads (from the moment the stream
 * is idle, i.e. compute excluded), out[4] ms the host spent pre-faulting result pages (runs beside the kernels),
 * out[5] number of host results.  A result buffer of >= 64 MiB is touched by a few host threads while the kernels run
 * (first-touch page faults bound a copy into fresh pages at ~15 GB/s on this host; into touched pages it runs at
 * ~50 GB/s), one byte per page, read and written back unchanged. */
int pfd_transfer_stats(double out[6], int reset);

/* ---- raster handle -------------------------------------------------------------------
 * pfd_raster_create: replaces core_d8.from_array (reference pyflwdir/core_d8.py:42-67) as the
 * constructor of the device-side graph.  `d8` holds nrow*ncol uint8 D8 codes (1 E, 2 SE,
 * 4 S, 8 SW, 16 W, 32 NW, 64 N, 128 NE, 0/255 pit, 247 nodata; core_d8.py:14-19).  Cells
 * whose target lies outside the raster or on a nodata cell become pits (core_d8.py:57-63).
 * Any other value -> PFD_EBADCODE (the reference only accepts such rasters with
 * check_ftype=False and then decodes them with log2 arithmetic the device path does not
 * imitate).  Fails with PFD_ENOPITS when no pit exists (reference pyflwdir/flwdir.py:126).
 * Rasters of more than 4294967294 cells (32-bit device indices) are accepted for
 * pfd_upstream_area_cell only (LDS-tiled path, e.g. 90000 x 90000 on one 288 GB GPU). */
int pfd_raster_create(const uint8_t *d8, int64_t nrow, int64_t ncol, int memspace, int device,
                      pfd_raster **out);
int pfd_raster_destroy(pfd_raster *h);
/* A flow graph given by its downstream indices (idx_dtype PFD_I32 / PFD_U32 / PFD_I64; -1 cast = nodata,
 * own index = pit) whose links are NOT restricted to the 8 neighbours: NEXTXY rasters (reference
 * pyflwdir/core_nextxy.py:41-68) and upscaled networks (FlwdirRaster(idxs_ds=...), pyflwdir.py:1079-1085).
 * Served by the general level engine (upstream CSR + one launch per level): ordering / idxs_seq / rank /
 * upstream_count / upstream_area / accuflux / Strahler and classic order / basins / HAND / main_upstream /
 * stream_distance (real_length: `step_lengths` = n HOST float32, one per cell); the tile engines, the
 * multi-GPU entry points, ucat_area, floodplains and snap return PFD_EUNSUPPORTED on such a handle. */
int pfd_raster_create_general(const void *idxs_ds, int idx_dtype, int64_t nrow, int64_t ncol, int memspace, int device,
                              pfd_raster **out);
/* One row block of a raster that is tiled over several GPUs (Multi-GPU): `d8` holds
 * halo_top + own_rows + halo_bot rows; the halo rows (0 or 1 each) are copies of the adjacent
 * rows of the neighbouring blocks and are only used to decide where flow leaves the block.
 * Block handles support pfd_upstream_area_cell_blocks / pfd_upstream_area_cell_dist only. */
int pfd_raster_create_block(const uint8_t *d8, int64_t own_rows, int64_t ncol, int halo_top, int halo_bot,
                            int memspace, int device, pfd_raster **out);
/* Same raster / row block (halo_top = halo_bot = 0: a whole raster), but decoding, the pit rule,
 * validation and the counts are DEFERRED to the first operation on the handle: fused into the
 * first tile pass of pfd_upstream_area_cell (and its _blocks/_dist/_begin forms), done by a
 * separate pass for every other entry point.  PFD_EBADCODE / PFD_ENOPITS are then returned by
 * that operation instead of by this call, and pfd_raster_info reports n_valid = n_pits = -1
 * until then.  With PFD_DEVICE the buffer is referenced, not copied: it must stay valid and
 * unmodified until the first operation on the handle has returned.
 * Mirrors from_array(..., check_ftype=False) (reference pyflwdir/pyflwdir.py:93-100) in that the
 * constructor does no validation pass; unlike it, invalid codes are still rejected later. */
int pfd_raster_create_deferred(const uint8_t *d8, int64_t own_rows, int64_t ncol, int halo_top, int halo_bot,
                               int memspace, int device, pfd_raster **out);
/* Normalises and validates a deferred handle now (no-op on any other handle). */
int pfd_raster_validate(pfd_raster *h);

/* info[0]=nrow [1]=ncol [2]=n_valid [3]=n_pits [4]=n_seq (cells draining to a pit; -1 until
 * the cells are ordered) [5]=n_levels (max rank + 1; -1 until ordered) [6]=device
 * [7]=bytes of HBM held by the handle */
int pfd_raster_info(pfd_raster *h, int64_t info[8]);

/* FlwdirRaster.add_pits (reference pyflwdir/flwdir.py:261-279): turn the given cells into
 * pits and invalidate the ordering.  Indices must address valid cells. */
int pfd_add_pits(pfd_raster *h, const int64_t *idxs, int64_t k);

/* ---- index exports (public attributes of FlwdirRaster) ------------------------------- */
/* idxs_ds in the reference's index dtype (PFD_I32/PFD_U32/PFD_I64; pyflwdir.py:105-127):
 * -1 (cast) on nodata, own index on pits.  core_d8.from_array, core_d8.py:42-67. */
int pfd_idxs_ds(pfd_raster *h, int idx_dtype, void *out, int memspace);
/* pit indices ascending; out has n_pits entries.  core_d8.py:62 / core.pit_indices */
int pfd_idxs_pit(pfd_raster *h, int idx_dtype, void *out, int memspace);
/* core.upstream_count (reference pyflwdir/core.py:50-61): int8 in-degree, -9 on nodata;
 * `mask` (uint8, may be NULL) restricts the contributing cells. */
int pfd_upstream_count(pfd_raster *h, const uint8_t *mask, int8_t *out, int memspace);
/* Flwdir.order_cells("walk") (reference pyflwdir/flwdir.py:231-250): build the level
 * structure (cells grouped by rank) that every sweep below uses.  Called implicitly. */
int pfd_order_cells(pfd_raster *h);
/* core.idxs_seq (reference pyflwdir/core.py:87-117): the exact breadth-first order of the
 * reference (pits ascending, then each dequeued cell's upstream cells ascending); out has
 * n_seq entries.  A raster beyond 2^32 - 2 cells (the int64 rung of pyflwdir.py:105-127) takes idx_dtype PFD_I64
 * only: built without the level structure from the tiled rank query and a level-by-level expansion with 64-bit queue
 * entries (csrc/order64.hip); `out` must hold n_valid entries, n_seq of them are written (pfd_raster_info afterwards) —
 * cells that never reach a pit are left out, found by a walk from the pits with one host look per level. */
int pfd_idxs_seq(pfd_raster *h, int idx_dtype, void *out, int memspace);
/* General idxs_ds graphs only: install the cell sequence the sweeps follow.  Flwdir.order_cells("sort")
 * (reference pyflwdir/flwdir.py:231-245; the only ordering of NEXTXY rasters, pyflwdir.py:292-297) sorts the
 * cells by rank with numpy's argsort, and the serial loops add upstream cells in the reverse order of that
 * sequence — a float accumulation depends on it.  `seq`: HOST array of n_seq indices, ordered by rank, every
 * cell that drains to a pit exactly once (anything else: PFD_EINVAL).  seq == NULL with n_seq == 0 forgets an
 * installed sequence: the sweeps (and pfd_idxs_seq) are back on the breadth-first order of core.idxs_seq.
 * pfd_add_pits forgets it as well (it describes the graph before the edit). */
int pfd_set_idxs_seq(pfd_raster *h, int idx_dtype, const void *seq, int64_t n_seq);
/* core.rank (reference pyflwdir/core.py:17-47): int32 distance to the pit, -1 for cells that
 * do not drain to a pit, -9999 on nodata.  Beyond 2^32 - 2 cells: the tiled rank query; a raster with cells that never
 * reach a pit is walked from its pits instead (slower: a host look per level), so that those cells read -1. */
int pfd_rank(pfd_raster *h, int32_t *out, int memspace);

/* ---- sweeps ---------------------------------------------------------------------------- */
/* FlwdirRaster.upstream_area(unit="cell") (reference pyflwdir/pyflwdir.py:770-801): int32
 * upstream cell count incl. the cell itself, -9999 on nodata cells, cells that do not drain
 * to a pit keep 1.  Fused fast path (no weight array is read). */
int pfd_upstream_area_cell(pfd_raster *h, int32_t *out, int memspace);
/* same result through the generic level-by-level engine (kept for cross-checking) */
int pfd_upstream_area_cell_levels(pfd_raster *h, int32_t *out, int memspace);
/* streams.accuflux / streams.accuflux_ds (reference pyflwdir/streams.py:15-41, :44-70).
 * dtype in {PFD_I32, PFD_I64, PFD_F32, PFD_F64}; `nodata_i` is used for the integer types,
 * `nodata_f` for the float types; has_nodata=0 disables the nodata test (the Python
 * comparison can never match, e.g. NaN).  Children are added in the reference's order
 * (descending linear index), so float results are bit-identical to the serial loop.
 * If mask_invalid != 0 nodata cells of the raster are set to `nodata` afterwards
 * (FlwdirRaster.upstream_area, pyflwdir.py:800). */
int pfd_accuflux(pfd_raster *h, int dtype, const void *data, int64_t nodata_i, double nodata_f,
                 int has_nodata, int direction, int mask_invalid, void *out, int memspace);
/* The same accumulation for a payload that is constant along raster rows: `row_values` is a HOST
 * pointer to nrow values of `dtype` (cell areas of a regular grid depend on the row only —
 * FlwdirRaster.upstream_area(unit != "cell"), reference pyflwdir/pyflwdir.py:770-801 with
 * gis_utils.area_grid, gis_utils.py:388-402 — so no n-element input has to exist or travel). */
int pfd_accuflux_rows(pfd_raster *h, int dtype, const void *row_values, int64_t nodata_i, double nodata_f,
                      int has_nodata, int direction, int mask_invalid, void *out, int memspace);
/* OPT-IN, tolerance mode of FlwdirRaster.upstream_area(unit != "cell") on lat/lon grids (reference
 * pyflwdir/pyflwdir.py:770-801: a float64 accumulation of cell areas): `row_values` is a HOST pointer to nrow float64
 * areas, `out` receives n float64 sums, -9999 on nodata cells.  The areas are quantised to 64-bit fixed point (the
 * largest power-of-two scale that keeps the raster's total below 2^64; *quantum, if given, receives one unit of it) and
 * accumulated as integers on the LDS-tiled engine of pfd_upstream_area_cell (a cell gets the integer part of its row's
 * scaled area plus its Bresenham share of the fraction, so that any run of consecutive cells of a row is off by less than
 * one quantum): the result does not depend on any execution order, every value is the exact sum of the quantised areas,
 * |error| < upstream cells x quantum in the worst case (relative: <= n_cells / 2^63 x mean / min area, 1.3e-10 at
 * 30000^2, reached at cells with a handful of upstream cells) plus one float64 rounding — NOT bit-identical to the reference's
 * serial float64 sum (pfd_accuflux_rows is).  *used = 0: not taken (cycles, row block, general graph, a value that is not
 * finite and positive, a supertile the 64-bit LDS form cannot hold); `out` is then undefined, call pfd_accuflux_rows. */
int pfd_upstream_area_rows_fixed(pfd_raster *h, const double *row_values, double *out, int memspace, int *used,
                                 double *quantum);
/* streams.strahler_order (reference pyflwdir/streams.py:228-269); mask uint8 or NULL. */
int pfd_strahler(pfd_raster *h, const uint8_t *mask, uint8_t *out, int memspace);
/* basins.basins + core.fillnodata_upstream (reference pyflwdir/basins.py:12-18,
 * pyflwdir/core.py:120-146): `outlets` are k linear indices (host memory, always), `ids`
 * their k labels of `id_size` bytes each (1, 2, 4 or 8; no zeros).  out: n labels. */
int pfd_basins(pfd_raster *h, const int64_t *outlets, const void *ids, int64_t k, int id_size,
               void *out, int memspace);
/* dem.height_above_nearest_drain (reference pyflwdir/dem.py:299-330): drain uint8 (==1 is
 * drain), elevtn PFD_F32 or PFD_F64, out float64 (-9999 off the sequence). */
int pfd_hand(pfd_raster *h, const uint8_t *drain, int elev_dtype, const void *elevtn, double *out,
             int memspace);

/* ---- SURVEY 8(f)-1 -------------------------------------------------------------------------
 * core.main_upstream (reference pyflwdir/core.py:191-219; Flwdir.main_upstream flwdir.py:252-258):
 * per cell the linear index of the upstream cell with the largest `uparea` (payload dtype
 * PFD_I32 / PFD_I64 / PFD_F32 / PFD_F64; strictly larger than upa_min, cast to that dtype, and
 * than every upstream cell of lower index), the index dtype's missing value (-1 / 0xFFFFFFFF)
 * for headwaters and nodata cells.  out: n indices of idx_dtype. */
int pfd_main_upstream(pfd_raster *h, int dtype, const void *uparea, double upa_min, int idx_dtype, void *out,
                      int memspace);
/* arithmetics.upstream_sum (reference pyflwdir/arithmetics.py:147-169; Flwdir.upstream_sum flwdir.py:412-433):
 * per cell the sum of `data` over the cells directly upstream, in the payload dtype (int wrap-around, float
 * addition in ascending cell index like the reference's loop); cells whose own or downstream value is `nodata`
 * get nodata in the place the serial loop would write it; 0 elsewhere (headwaters, nodata cells).  has_nodata = 0:
 * no value compares equal to the missing value (NaN, or not representable in the payload dtype). */
int pfd_upstream_sum(pfd_raster *h, int dtype, const void *data, int64_t nodata_i, double nodata_f, int has_nodata,
                     void *out, int memspace);
/* streams.stream_order, the classic "bottom up" order (reference pyflwdir/streams.py:191-225;
 * Flwdir.stream_order(type="classic") flwdir.py:540-543): uint8; pits 1, tributaries (cells that
 * are not the main upstream cell of a downstream cell with more than one upstream cell inside
 * `mask`) one more than the cell they drain into, 0 outside `mask` (nullable: all cells). */
int pfd_stream_order_classic(pfd_raster *h, int idx_dtype, const void *idxs_us_main, const uint8_t *mask,
                             uint8_t *out, int memspace);
/* streams.stream_distance (reference pyflwdir/streams.py:272-315; FlwdirRaster.stream_distance
 * pyflwdir.py:837-863): distance to the outlet or to the next downstream cell with mask != 0
 * (nullable), -9999 for cells that do not drain to a pit.  real_length == 0: int32 cell counts.
 * real_length != 0: float32; `step_lengths` (HOST pointer, 3 * (2*nrow - 1) floats) holds the
 * length of one step by row sum r0 + r1 and kind {vertical, horizontal, diagonal} — the host
 * evaluates gis_utils.distance (gis_utils.py:452-486), the device only adds in float32. */
int pfd_stream_distance(pfd_raster *h, const uint8_t *mask, int real_length, const float *step_lengths, void *out,
                        int memspace);

/* ---- multi-GPU: upstream_area(unit="cell") on a raster row-tiled over several GPUs -----------
 * In-process form: `hs` are the nblocks row-block handles (top to bottom) of ONE process,
 * outs[b] receives own_rows(b)*ncol int32. */
int pfd_upstream_area_cell_blocks(pfd_raster **hs, int nblocks, int32_t **outs, int memspace);
/* One-process-per-GPU form over RCCL.  Rank 0 obtains a unique id (128 bytes) and ships it to
 * the other ranks by any host-side means; every rank then creates its communicator and calls
 * pfd_upstream_area_cell_dist collectively with its own block (rank r holds block r). */
typedef struct pfd_comm pfd_comm;
int pfd_comm_unique_id(void *id_out, size_t len);
int pfd_comm_create(const void *id, size_t len, int rank, int world, int device, pfd_comm **out);
int pfd_comm_destroy(pfd_comm *c);
/* what RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) */
int pfd_comm_info(pfd_comm *c, int *nranks, int *rank, int *device);
int pfd_upstream_area_cell_dist(pfd_raster *h, pfd_comm *comm, int32_t *out, int memspace);

/* The exchange step of the iterated row-block collectives (sharded HAND, float accuflux, area-unit upstream_area,
 * Strahler, stream_distance: pyflwdir_amd/dist.py DistributedRaster), device to device over RCCL — SURVEY 8e's
 * "ncclSend/ncclRecv pairs in one ncclGroup": rank k sends the first OWN row of `result_dev` to rank k-1 and the
 * last OWN row to rank k+1 and receives their boundary rows as its new halo values.
 *   result_dev  the block's result on the device (own + halo rows, elem_bytes per cell: 1, 4, 8 or 16)
 *   seed_dev    DEVICE, 2 * ncol elements: the halo values of the next sweep (top halo row, bottom halo row), updated
 *               in place; sides without a neighbour keep their content
 *   counters    in: two local counts; out: [0], [1] their sums over the ranks (one ncclAllReduce), [2] 1 if a
 *               received row differs (bitwise) from the halo values it replaces, [3] the sum of [2] over the ranks
 * No row travels through the host: one stream synchronisation and a 32-byte read-back per exchange.  Every rank
 * must call it (collective).  A rank on which something fails locally (a handle whose halo rows do not fit its rank, a
 * HIP call) still takes part in the send / recv group and in the all-reduce — nobody is left waiting — adds 1 to
 * counters[1] (callers use [1] as "ranks that failed") and returns its own error afterwards.  The block sweeps read `halo_seed_host` from the DEVICE after pfd_set_block_io(h,
 * PFD_DEVICE). */
int pfd_comm_exchange_rows(pfd_comm *comm, pfd_raster *h, const void *result_dev, int elem_bytes, void *seed_dev,
                           int64_t counters[4]);
/* all-gather of `nbytes` host bytes per rank through the device and RCCL (the one-shot boundary records of the sharded
 * basins): out_host receives world * nbytes bytes in rank order.  Collective. */
int pfd_comm_allgather_host(pfd_comm *comm, pfd_raster *h, const void *in_host, size_t nbytes, void *out_host);
/* where the pfd_*_block entry points read their `halo_seed_host` argument from: PFD_HOST (default) or PFD_DEVICE */
int pfd_set_block_io(pfd_raster *h, int seed_memspace);
/* Split-phase form for callers that move the boundary records themselves (MPI, gloo, shared memory ...):
 * begin() runs the local phase and returns this block's record (4*ncol uint32, host memory); finish()
 * takes the records of all blocks in block order (nblocks*4*ncol uint32, host) and completes the pass.
 * *complete = 0 reports cells that never reach a pit (cycles). */
int pfd_upstream_area_cell_begin(pfd_raster *h, int32_t *out, int memspace, uint32_t *record_host);
int pfd_upstream_area_cell_finish(pfd_raster *h, const uint32_t *all_records_host, int nblocks, int block,
                                  int *complete);

/* HAND (reference pyflwdir/dem.py:299-330) of a ROW BLOCK: like pfd_hand on the block's device raster (own rows +
 * halo rows; drain / elevtn / out cover all of them), except that a valid halo cell — the neighbouring block's
 * boundary cell — takes its height from `halo_seed_host` (HOST, 2 * ncol doubles: top halo row, bottom halo row)
 * instead of being computed.  A path that leaves the block therefore continues the neighbour's sum with the same
 * operands in the same order: bit-identical to the whole raster once the seeds are the neighbour's final values.
 * The caller iterates (pyflwdir_amd/dist.py DistributedRaster.hand): seeds start as -inf ("not known yet", which
 * every sum that depends on it inherits), the blocks exchange their boundary rows, until no owned cell is -inf.
 * update != 0: `out` holds the result of an earlier call (other seeds); only its -inf cells are recomputed (collected
 * and relaxed to a fixpoint; a full sweep again when they are more than 1/16 of the block).
 * boundary_rows_host (nullable): receives the first and the last OWN row (2 * ncol doubles); n_unknown (nullable):
 * the number of own cells that are still -inf. */
int pfd_hand_block(pfd_raster *h, const uint8_t *drain, int elev_dtype, const void *elevtn,
                   const double *halo_seed_host, int update, double *out, int memspace, double *boundary_rows_host,
                   int64_t *n_unknown);

/* Up-sweeps of a ROW BLOCK — accuflux (reference pyflwdir/streams.py:15-41, direction "up") and Strahler order
 * (streams.py:228-269) — for rasters cut into row blocks (several GPUs, or one GPU and more than 2^32 - 2 cells).
 * Like pfd_accuflux / pfd_strahler on the block's device raster (own rows + halo rows; data / mask / out cover all of
 * them), except that a valid halo cell — the neighbouring block's boundary cell — has the value given in
 * `halo_seed_host` (HOST, 2 * ncol elements of the result type: top halo row, bottom halo row).  A halo cell that drains
 * into the block is added by its downstream cell in the serial loop's position (upstream cells in descending linear
 * index), so floats come out bit-identical to the whole raster once the seeds are the neighbours' final rows.  The caller
 * iterates (pyflwdir_amd/dist.py up_blocks): exchange the boundary rows and sweep again until no row changes.
 *   direction:   PFD_UP, or PFD_DOWN (streams.accuflux_ds, streams.py:44-70: values travel upstream; a halo cell the
 *                block drains into then holds the neighbour's value, and the same exchange-until-stable applies).
 *   by_row != 0: `data` holds one value per ROW of the device raster (HOST; upstream_area(unit="km2") on lat/lon grids).
 *   verify != 0: nothing is computed; `out` holds a result, *n_bad receives the number of own cells whose value is not
 *                the one their upstream cells (halo seeds included) give — the all-cell check of a blocked result.
 *   boundary_rows_host (nullable): receives the first and the last OWN row (2 * ncol elements). */
/* Repeated up-sweeps of one block (the exchange-until-stable iteration above) — what the next pfd_accuflux_block
 * (direction PFD_UP) / pfd_strahler_block call on `h` keeps or reuses, with result and payload in DEVICE memory:
 *   0 (default)  nothing is kept (a kept sweep is released);
 *   1            a full sweep whose per-chain arrays and seeds stay on the handle (8-16 bytes per trunk slot);
 *   2            `out` still holds the result of the kept sweep of the same operation, payload and mask: only the
 *                chains below a halo seed that CHANGED are folded again (no tile pass, no pass over the raster: the
 *                cost follows the cells the changed seeds reach).  Falls back to mode 1 when there is no kept sweep
 *                of this operation into this buffer.  Results are the full sweep's, bit for bit.
 * Handles whose block has no exact-order plan (PFD_BLOCK_LEVELS) ignore the mode.  With mode 1 or 2 a call whose
 * memspace is not PFD_DEVICE returns PFD_EINVAL (the kept sweep refers to `out` by address). */
int pfd_set_block_update(pfd_raster *h, int mode);
int pfd_accuflux_block(pfd_raster *h, int dtype, const void *data, int by_row, int64_t nodata_i, double nodata_f,
                       int has_nodata, int direction, const void *halo_seed_host, int verify, void *out, int memspace,
                       void *boundary_rows_host, int64_t *n_bad);
int pfd_strahler_block(pfd_raster *h, const uint8_t *mask, const uint8_t *halo_seed_host, int verify, uint8_t *out,
                       int memspace, uint8_t *boundary_rows_host, int64_t *n_bad);
/* stream_distance (reference pyflwdir/streams.py:272-315) of a row block, like pfd_accuflux_block in direction "down":
 * the halo cells the block drains into hold the neighbour's distances (`halo_seed_host`: 2 * ncol int32, or float32
 * when real_length != 0); `step_lengths` covers the rows of the block's device raster (3 * (2 * nrow - 1) floats: the
 * slice of the whole raster's table that starts at the block's first device row). */
int pfd_stream_distance_block(pfd_raster *h, const uint8_t *mask, int real_length, const float *step_lengths,
                              const void *halo_seed_host, int verify, void *out, int memspace, void *boundary_rows_host,
                              int64_t *n_bad);
/* dem.floodplains (reference pyflwdir/dem.py:333-379) of a row block, like pfd_stream_distance_block: the halo cells the
 * block drains into hold the neighbouring block's floodplain STATE, 16 bytes per cell {float z, float h, int32 flag, int32
 * pad} (z / h: elevation and height threshold of the stream cell that started the floodplain; flag 1 floodplain, 0 not,
 * -1 nodata); `state` (own + halo rows of the block's device raster) is the result, `halo_seed_host` / boundary rows hold
 * 2 * ncol such records.  `is_stream` / `stream_h` as in pfd_floodplains.  Iterated by pyflwdir_amd/dist.py
 * floodplains_blocks until no boundary row changes: floodplains on rasters beyond 2^32 - 2 cells. */
int pfd_floodplains_block(pfd_raster *h, int elev_dtype, const void *elevtn, const uint8_t *is_stream, const float *stream_h,
                          const void *halo_seed_host, int verify, void *state, int memspace, void *boundary_rows_host,
                          int64_t *n_bad);
/* The int8 result of dem.floodplains (1 floodplain, 0 not, -1 off the sequence; dem.py:333-379) for the block's OWN rows,
 * from the DEVICE-resident `state` of pfd_floodplains_block: own_rows * ncol bytes into `out` (host or device) — a caller
 * that assembles the raster on the host moves 1 byte per cell instead of the 16-byte state records. */
int pfd_floodplains_block_flags(pfd_raster *h, const void *state_dev, int8_t *out, int memspace);
/* Classic stream order (reference pyflwdir/streams.py:191-225 with core.main_upstream, core.py:191-219) over row blocks.
 * pfd_trib_info_block: one byte per cell of the block's device raster — low 4 bits the slot (0-7; 15 none) of the cell's
 * main upstream cell (largest `uparea` > upa_min, first in ascending index), bit 4 set when more than one upstream cell
 * lies inside `mask` (nullable) — from the upstream masks that include the halo cells.  The byte of a HALO cell is
 * incomplete (its upstream cells lie beyond the block): the caller replaces the halo rows of `tinfo` with the
 * neighbouring blocks' boundary rows, then pfd_stream_order_classic_block sweeps down- to upstream with the halo cells
 * the block drains into holding the neighbour's orders (`halo_seed_host`: 2 * ncol uint8), like
 * pfd_stream_distance_block; iterated until no boundary row changes (pyflwdir_amd/dist.py classic_blocks). */
int pfd_trib_info_block(pfd_raster *h, int dtype, const void *uparea, double upa_min, const uint8_t *mask, uint8_t *tinfo,
                        int memspace);
int pfd_stream_order_classic_block(pfd_raster *h, const uint8_t *tinfo, const uint8_t *mask, const uint8_t *halo_seed_host,
                                   int verify, uint8_t *out, int memspace, uint8_t *boundary_rows_host, int64_t *n_bad);

/* basins (reference pyflwdir/basins.py:12-18, core.py:120-146) on a raster row-tiled over several GPUs /
 * processes, split-phase like pfd_upstream_area_cell_begin/_finish (Multi-GPU): `outlets` are k
 * linear indices INTO THE BLOCK'S OWN ROWS (r * ncol + c, r counted from the block's first own row), `ids`
 * their labels (id_size bytes each, no zeros).  begin() runs the local label query and returns the block's
 * boundary record (6 * ncol uint32, host); finish() takes the records of all blocks in block order
 * (nblocks * 6 * ncol uint32, host), resolves the paths that leave the block and writes the own_rows * ncol
 * labels to the `out` given to begin().  *complete = 0 reports a cycle through several blocks. */
int pfd_basins_begin(pfd_raster *h, const int64_t *outlets, const void *ids, int64_t k, int id_size, void *out,
                     int memspace, uint32_t *record_host);
int pfd_basins_finish(pfd_raster *h, const uint32_t *all_records_host, int nblocks, int block, int *complete);

/* ---- the step before the path (DEM -> D8) -------------------------------------------------
 * dem.fill_depressions (reference pyflwdir/dem.py:17-143; from_dem pyflwdir/pyflwdir.py:51-102): priority flood
 * (Wang & Liu 2006) with the reference's heap order (float32 elevation, boundary flag, row, column).  HOST
 * function (inherently sequential), all pointers are host pointers.  dtype PFD_F32 / PFD_F64 / PFD_I32;
 * nodata NaN selects isnan(); max_depth < 0: fill everything; outlets_min != 0: one outlet at the lowest edge
 * cell; has_elv_max: edge outlets only up to elv_max; idxs_pit (nullable): user outlets instead of the edge;
 * connectivity 4 or 8.  elev_out: filled elevation (dtype of the input), d8_out: uint8 D8 codes (247 nodata). */
int pfd_fill_depressions(int dtype, const void *elevtn, int64_t nrow, int64_t ncol, double nodata, double max_depth,
                         int outlets_min, int has_elv_max, double elv_max, const int64_t *idxs_pit, int64_t npit,
                         int connectivity, void *elev_out, uint8_t *d8_out);

/* ---- SURVEY 8(f)-4 --------------------------------------------------------------------------------
 * subgrid.ucat_area (reference pyflwdir/subgrid.py:51-93; FlwdirRaster.ucat_area pyflwdir.py:1159-1191): unit
 * catchment map and area.  `idxs_out`: k outlet cells (HOST; < 0 = missing).  map_out: n labels of map_dtype
 * (PFD_I32/U32/I64; label i+1 for outlet i, 0 elsewhere).  area_dtype PFD_I32: cell counts (area_rows NULL);
 * PFD_F32/PFD_F64: `area_rows` = nrow HOST values, the area of a cell of that row; area_out: k HOST values of
 * area_dtype (-9999 for missing outlets), float sums accumulated in the reference's (idxs_seq) order.
 * Any raster size (round 6): beyond 2^32 - 2 cells the label query runs on the whole raster (tiles and slots, 32-bit
 * values) and the float sums walk the 64-bit sequence (csrc/order64.hip) in sorted pieces, bit-identical to the
 * reference's loop; a raster of that size WITH cycles returns PFD_EUNSUPPORTED (its label query needs the level
 * engine's 32-bit order) — as does pfd_basins, which otherwise runs at any size too. */
int pfd_ucat_area(pfd_raster *h, const int64_t *idxs_out, int64_t k, int map_dtype, void *map_out, int memspace,
                  int area_dtype, const void *area_rows, void *area_out);
/* dem.floodplains (reference pyflwdir/dem.py:333-379; FlwdirRaster.floodplains pyflwdir.py:1513-1545):
 * `is_stream` uint8 (1 where uparea >= upa_min), `stream_h` float32 (uparea ** b on those cells — evaluated by
 * the caller in the reference's dtype), elevtn PFD_F32 / PFD_F64; out int8: 1 floodplain, 0 not, -1 off the sequence. */
int pfd_floodplains(pfd_raster *h, int elev_dtype, const void *elevtn, const uint8_t *is_stream, const float *stream_h,
                    int8_t *out, int memspace);

/* core.snap in downstream direction, cell units (reference pyflwdir/core.py:440-480, Flwdir.snap
 * flwdir.py:404-463; used by basins(streams=...) / add_pits(streams=...), flwdir.py:805-811): per start cell
 * (k HOST indices) the first cell downstream, itself included, where mask != 0, or the pit its path ends in;
 * dist_out = cells walked (float32 like the reference); max_hops < 0: unlimited.  64-bit cell indices: both snap
 * entries serve whole rasters of any size. */
int pfd_snap_downstream(pfd_raster *h, const int64_t *idxs, int64_t k, const uint8_t *mask, int memspace,
                        int64_t max_hops, int64_t *idxs_out, float *dist_out);
/* The general form of core.snap (reference pyflwdir/core.py:440-480, core._trace :308-366, Flwdir.snap
 * flwdir.py:500-560): `idxs_us_main` == NULL walks downstream, else upstream along the caller's main upstream
 * cells (HOST, n int64, negative = none); `step_lengths` == NULL counts cells, else adds metres from the HOST
 * table [2*nrow - 1][3] float64 (row + next row; vertical, horizontal, diagonal step: gis_utils.distance
 * evaluated by the host); `mask` (HOST, nullable) ends a walk on the first cell where it is set; with
 * has_max_length a walk also ends before the step that would exceed max_length.  Distances accumulate in float64
 * (the reference's Python float) and are stored as float32. */
int pfd_snap(pfd_raster *h, const int64_t *idxs, int64_t k, const uint8_t *mask, const int64_t *idxs_us_main,
             const double *step_lengths, int has_max_length, double max_length, int64_t *idxs_out, float *dist_out);

/* ---- instrumentation ---------------------------------------------------------------------
 * HIP-event timing (events recorded on the handle's own stream) of the phases of the LAST
 * sweep/ordering call on the handle.  Enable with pfd_set_profiling(h, 1).  Up to max_seg
 * segments are returned: ms[i] GPU milliseconds, launches[i] kernel launches in the segment,
 * names = segment names joined by ';' (e.g. "order_cells;init;sweep_count_up").  enable = 2 additionally makes
 * the tile passes of upstream_area("cell") count their pointer-doubling rounds (pfd_graph_stats; costs two
 * atomics per tile, so not for timed runs). */
int pfd_set_profiling(pfd_raster *h, int enable);
int pfd_last_timing(pfd_raster *h, int max_seg, double *ms, int64_t *launches, char *names,
                    size_t names_len, int *nseg);

/* Graph statistics a benchmark reports next to every number: stats[0] = n_valid,
 * [1] = n_pits, [2] = max rank = longest flow path in cells (reference core.rank, pyflwdir/core.py:17-47;
 * -1 if the raster holds cycles or is a row block, -2 if it is beyond the slot ids of the tiled rank query: unknown), [3..11] = number of valid cells with 0..8
 * upstream cells (reference core.upstream_count, pyflwdir/core.py:50-61); [12..15] = pointer-doubling rounds
 * of the last upstream_area("cell") pass this handle ran under pfd_set_profiling(h, 2): max and sum over the tiles of the
 * local pass, max and sum of the final pass (0 if there was none).  Works on rasters beyond 2^32 cells. */
int pfd_graph_stats(pfd_raster *h, int64_t stats[16]);

/* Verifies an upstream_area("cell") result of this raster by its local equations, at any size
 * (whole rasters, incl. beyond 2^32 cells): res[0] = valid cells with upa != 1 + sum over the cells
 * draining into them, [1] = nodata cells != -9999, [2] = sum of upa over the pits, [3] = pits,
 * [4] = sum of all n values (two's complement), [5] = valid cells.  On a raster without cycles
 * res[0] == res[1] == 0 holds for exactly one array: the reference's result (streams.accuflux over
 * ones, pyflwdir/streams.py:15-41; invariant [2] == [5]: tests/test_streams_basins.py:24-27). */
int pfd_verify_upstream_area_cell(pfd_raster *h, const int32_t *upa, int memspace, int64_t res[8]);

/* The same for a basins() result with uint32 ids (reference pyflwdir/basins.py:12-18, core.py:120-146) and for a HAND
 * result (dem.py:299-330), each cell against its downstream cell only: res[0] = valid cells violating their equation
 * (labels: seeded cell == its id, other cells == the label of their downstream cell, 0 for a pit; HAND: 0 on drain
 * cells, else hand[ds] + (double)(elevtn[x] - elevtn[ds]) with the difference in the elevation dtype, bit for bit),
 * res[1] = nodata cells != 0 / != -9999, res[2] = sum of all values (HAND: of their bit patterns), res[3] = labelled
 * cells / drain cells.  `outlets` (k distinct linear indices, HOST) and `ids` (HOST) are the seeds of the labels. */
int pfd_verify_basins(pfd_raster *h, const int64_t *outlets, const uint32_t *ids, int64_t k, const uint32_t *labels,
                      int memspace, int64_t res[4]);
int pfd_verify_hand(pfd_raster *h, const uint8_t *drain, int elev_dtype, const void *elevtn, const double *hand,
                    int memspace, int64_t res[4]);

/* sum of n int32 values in HBM (two's complement, 64 bit): the checksum multi-block runs compare with a
 * single-GPU run of the same raster */
int pfd_checksum_i32(int device, const int32_t *dev_ptr, int64_t n, int64_t *sum);
/* number of NaN / +-inf values of a DEVICE array (dtype PFD_F32 or PFD_F64): callers with device-resident elevations check
 * them with it before the row-block HAND (pfd_hand_block), whose "-inf = height not known yet" a non-finite difference imitates */
int pfd_count_nonfinite(int device, int dtype, const void *dev_ptr, int64_t n, int64_t *count);

/* ---- synthetic rasters (bench / tests; device twin of oracle/pfd_oracle.c orc_synth_*) ---- */
/* writes rows [row0, row0+nrows) of the nrow x ncol synthetic raster to device memory */
int pfd_synth_d8(int device, uint64_t seed, int64_t nrow, int64_t ncol, int64_t tilt, int64_t white,
                 int32_t nodata_pct, int64_t row0, int64_t nrows, uint8_t *out_dev);
int pfd_synth_elev_f32(int device, uint64_t seed, int64_t nrow, int64_t ncol, int64_t tilt,
                       int64_t white, int32_t nodata_pct, int64_t row0, int64_t nrows, float *out_dev);
/* a host base raster (brow x bcol D8 codes) tiled over nrow x ncol cells in HBM, every copy inside a one-cell nodata
 * frame: bench.py's realistic regime (the reference's Rhine sub-basin) at sizes that are never shipped over PCIe */
int pfd_synth_mosaic(int device, const uint8_t *base_host, int64_t brow, int64_t bcol, int64_t nrow, int64_t ncol,
                     uint8_t *out_dev);
/* measurement aid: one streaming pass over nbytes of buf_dev with `width`-byte accesses per lane (1, 4, 8, 16), reading or
 * writing — a kernel with a known byte count, for calibrating the rocprofv3 FETCH_SIZE / WRITE_SIZE counters per access
 * width (tools/prof_calib.sh) */
int pfd_calib_traffic(int device, void *buf_dev, size_t nbytes, int width, int write);
int pfd_synth_weights_f32(int device, uint64_t seed, int64_t i0, int64_t n, float *out_dev);

#ifdef __cplusplus
}
#endif
#endif /* PFD_H_ */
