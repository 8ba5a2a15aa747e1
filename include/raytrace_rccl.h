/*
 * raytrace_rccl.h -- C-ABI of libraytrace_rccl.so: the tile exchange of the one-process-per-GPU render path over RCCL / xGMI.
 *
 * BASELINE.json's north star: "The image is row-tile partitioned across the 8 GPUs of one node with a final RCCL gather over
 * xGMI".  The reference renders on the implicit device 0 and has one device-to-host copy of the whole image
 * (optimized.cu:849-856); with one process per GPU each rank renders its interleaved 8-row tiles (rt_render_device with an
 * rt_rows{rank * tile_rows, n, tile_rows, world}, raytrace_hip.h) and this library moves them into the root's frame:
 *
 *     rank 0:  rt_comm_id_create(id)  -> hand the 128 bytes to every rank (file, environment, MPI, a socket)
 *     every rank:  rt_comm_create(&c, device, rank, world, id)
 *                  rt_render_device(ctx, &p, &rows, tiles_dev, rt_comm_stream(c));        // (+ rt_tonemap_device for the 8-bit image)
 *                  rt_comm_gather_tiles(c, tiles_dev, W, H, bytes_per_pixel, tile_rows, 0, frame_dev_on_root, NULL);
 *                  rt_comm_sync(c);
 *
 * A tile is contiguous in the frame, so the root can receive every peer tile straight into place (one grouped ncclSend / ncclRecv
 * per tile; no staging buffer, no de-interleave pass) and copy its own tiles device to device; small tiles travel coalesced instead,
 * one message per peer plus one placement kernel (rt_comm_set_plan).  On MI355X's fully connected
 * xGMI every peer has its own link into the root, so the seven transfers run side by side (not a ring).
 * Kept out of libraytrace_hip.so so that the single-GPU path does not load librccl.
 *
 * Plain pointers and sizes only.  Every function returns RT_COMM_OK (0) or a negative code with a message in
 * rt_comm_last_error; nothing exits or throws.  There is no fallback transport in this library.
 */
#ifndef RAYTRACE_RCCL_H
#define RAYTRACE_RCCL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_COMM_ABI_VERSION 2
#define RT_COMM_ID_BYTES 128     /* NCCL_UNIQUE_ID_BYTES */
#define RT_COMM_MAX_TILES 8192   /* tiles of one frame (one send / receive each) */
#define RT_COMM_MAX_WORLD 64     /* ranks of one communicator */
#define RT_COMM_COALESCE_BELOW_DEFAULT (256u * 1024u)   /* AUTO plan: tiles smaller than this travel coalesced */

typedef enum rt_comm_status {
    RT_COMM_OK = 0,
    RT_COMM_ERR_INVALID = -1,    /* bad argument                                   */
    RT_COMM_ERR_HIP = -2,        /* a HIP call failed                              */
    RT_COMM_ERR_RCCL = -3        /* an RCCL call failed (message names the call)   */
} rt_comm_status;

typedef struct rt_comm rt_comm;

int rt_comm_abi_version(void);
/* rank 0, once per communicator: ncclGetUniqueId */
int rt_comm_id_create(unsigned char id[RT_COMM_ID_BYTES]);
/* every rank (collective: returns when all `world` ranks have called it): ncclCommInitRank on `device`, plus a stream of the
 * communicator's own.  Two ranks of one communicator must not share a device (RCCL refuses duplicate devices). */
int rt_comm_create(rt_comm **comm, int device, int rank, int world, const unsigned char id[RT_COMM_ID_BYTES]);
int rt_comm_destroy(rt_comm *comm);
const char *rt_comm_last_error(const rt_comm *comm);    /* comm may be NULL: last error of a call without a communicator */
int rt_comm_rank(const rt_comm *comm);
int rt_comm_world(const rt_comm *comm);
/* the communicator's stream as a hipStream_t: pass it to rt_render_device / rt_tonemap_device so that render, tone mapping and
 * exchange are ordered without a host-side wait */
void *rt_comm_stream(const rt_comm *comm);
/*
 * The gather (collective).  tiles_dev: this rank's dense tile buffer -- the rows of its tiles (tile t belongs to rank t % world;
 * rows [t * tile_rows, min(H, (t + 1) * tile_rows))) in tile order, bytes_per_pixel * W bytes per row (16 = the float4 frame,
 * 3 = the tone-mapped RGB8 image).  frame_dev: on `root` the H x W frame in its device memory; ignored elsewhere (may be NULL).
 * stream: a hipStream_t, or NULL for the communicator's own.  Asynchronous: rt_comm_sync waits for it (on whichever stream it went).
 */
int rt_comm_gather_tiles(rt_comm *comm, const void *tiles_dev, int W, int H, int bytes_per_pixel, int tile_rows, int root,
                         void *frame_dev, void *stream);
/* The plan rt_comm_gather_tiles follows for tile t of a frame (host arithmetic, no device, no communicator: tests replay the whole
 * exchange with it): the rank that renders the tile, where the tile starts in that rank's dense buffer and in the frame, its size. */
typedef struct rt_comm_tile {
    int32_t  owner;          /* t % world                                                        */
    int32_t  rows;           /* tile_rows, fewer for the last tile of a frame whose height is not a multiple */
    uint64_t local_offset;   /* bytes from the start of the owner's tile buffer                 */
    uint64_t frame_offset;   /* bytes from the start of the frame                               */
    uint64_t bytes;
} rt_comm_tile;
int rt_comm_tile_plan(int W, int H, int bytes_per_pixel, int tile_rows, int world, int t, rt_comm_tile *out);
/*
 * The exchange plan.  PER_TILE: one send / receive per tile, every tile received straight into its place in the frame (no staging, no
 * second pass) -- right for big tiles (7680x4320 float4: 983 KB each).  COALESCED: every peer sends its whole dense buffer as ONE
 * message into a staging area on the root, and one kernel on the root puts all tiles into place -- right for small ones (1920x1080
 * RGB8 at world 8: 17 messages of 46 KB per peer become one of 783 KB).  AUTO (default) picks COALESCED when a full tile
 * (W * bytes_per_pixel * tile_rows) is smaller than coalesce_below_bytes (0 = RT_COMM_COALESCE_BELOW_DEFAULT).  Every rank of a
 * communicator must set the same plan and threshold: both sides of the exchange derive their message sizes from it.
 * The staging area is ONE per communicator: a coalesced gather waits (stream-side, through an event) until the previous gather's placement
 * kernel has read it, on whatever stream that one ran -- gathers of several frames in flight on different streams are ordered by the library.
 */
typedef enum rt_comm_plan { RT_COMM_PLAN_AUTO = 0, RT_COMM_PLAN_PER_TILE = 1, RT_COMM_PLAN_COALESCED = 2 } rt_comm_plan;
int rt_comm_set_plan(rt_comm *comm, int plan, uint64_t coalesce_below_bytes);
int rt_comm_last_plan(const rt_comm *comm);              /* the plan the last rt_comm_gather_tiles followed (PER_TILE for world 1) */
/* what AUTO resolves to for a frame (host arithmetic) */
int rt_comm_choose_plan(int plan, uint64_t coalesce_below_bytes, int W, int bytes_per_pixel, int tile_rows);
/* The coalesced plan for rank `peer` (host arithmetic; tests replay the exchange with it and rt_comm_tile_plan): the one message it
 * sends -- its whole dense buffer -- and where the root stages it.  The placement kernel then copies tile t from
 * stage + stage_offset(owner) + tile_plan(t).local_offset (the root's own tiles from its own buffer) to tile_plan(t).frame_offset. */
typedef struct rt_comm_peer {
    uint64_t stage_offset;   /* bytes from the start of the root's staging area (256-byte aligned; 0 for the root itself) */
    uint64_t bytes;          /* size of the rank's dense tile buffer = of its one message                                */
    int32_t  n_tiles;        /* tiles it holds (0 when there are more ranks than tiles)                                  */
    int32_t  reserved;
} rt_comm_peer;
int rt_comm_peer_plan(int W, int H, int bytes_per_pixel, int tile_rows, int world, int root, int peer, rt_comm_peer *out);
/* bytes the last rt_comm_gather_tiles of this rank sent (peers) or received (root) over the fabric */
uint64_t rt_comm_last_bytes(const rt_comm *comm);
int rt_comm_sync(rt_comm *comm);                         /* waits for the last gather: hipStreamSynchronize of the stream it was issued on (the
                                                            caller's, if one was passed) and of the communicator's own */

#ifdef __cplusplus
}
#endif
#endif
