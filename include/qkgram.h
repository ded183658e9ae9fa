/*
 * qkgram.h -- C ABI of the MI355X (gfx950) quantum-kernel Gram engine.
 *
 * Drop-in boundary for ONE hot path of mmetcalf14/qml-cutensornet: filling
 *     K[j, i] = |<psi(x_i)|psi(y_j)>|^2
 * from matrix-product states.  Each entry point below names the reference
 * interface it replaces.  Short names:
 *     G = gpu_backend/kernel_state_ansatz.py     (reference, GPU backend)
 *     J = KernelPkg/src/KernelPkg.jl             (reference, CPU engine)
 *
 * Conventions
 *   - every function returns 0 on success, a negative QK_E* code otherwise;
 *     qk_last_error() gives a thread-local message.  No exception, callback or
 *     C++/torch type crosses this ABI: plain pointers and sizes only.
 *   - the library is batch-first: nothing here is called per Gram entry.
 *   - the caller owns host buffers; the library owns device memory behind
 *     opaque handles.  Calls on one context are serialised by the caller.
 *   - "device pointer" arguments are plain HIP device addresses (for example
 *     torch.Tensor.data_ptr()); work is enqueued on the context's stream.
 *   - there is NO CPU fallback: without a usable gfx950 device
 *     qk_ctx_create() fails and nothing else can be called.
 */
#ifndef QKGRAM_H
#define QKGRAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QK_OK 0
#define QK_EINVAL (-1)   /* bad argument / inconsistent shapes            */
#define QK_EDEVICE (-2)  /* HIP error (no device, launch/alloc failure)   */
#define QK_ENOMEM (-3)   /* host allocation failure                       */

/* host layout of one site tensor handed to qk_mps_set_create() */
#define QK_LAYOUT_LPR 0 /* [left bond][physical][right bond], C order */
#define QK_LAYOUT_LRP 1 /* [left bond][right bond][physical], C order */

/* flags of qk_plan_create() */
#define QK_PLAN_SYMMETRIC 1u /* Y is X: compute i <= j only, mirror on scatter (G:390-395) */
#define QK_PLAN_QUADS 2u     /* pairs in 2x2 blocks {i1,i2} x {j1,j2} of consecutive states: one workgroup sweeps a block in
                                lockstep (experimental: measured no faster than pairs on cfg4; such plans can be CREATED here but are
                                swept only by the lab library libqklab.so -- qk_gram_values of libqkgram.so returns QK_EINVAL).  The pair list then holds 4 entries per
                                block (a symmetric plan's diagonal blocks include one mirrored pair i > j, an odd set's last
                                block repeats its state): scatter handles both. */

#define QK_PLAN_ORIENT 4u    /* symmetric plans only: a pair {i, j} is listed as (i, j) or as (j, i), whichever order of contraction is
                                cheaper on the matrix cores (the sweep contracts the environment with the Y tensor first; which state
                                plays Y decides the padded tile counts).  |<x_i|x_j>| = |<x_j|x_i>|, so K is unchanged and the optional z
                                output is the overlap of the pair AS LISTED.  This is the host-side greedy choice of the contraction
                                order (north star; reference call site G:380). */

typedef struct qk_ctx qk_ctx;         /* one per device; replaces CuTensorNetHandle(device_id), G:213,255,366 */
typedef struct qk_mps_set qk_mps_set; /* a device-resident list of MPS; replaces the per-rank lists of
                                         pytket-cutensornet MPS objects mps_x_chunk / mps_y_chunk, G:210,252,290 */
typedef struct qk_plan qk_plan;       /* ordered list of (x, y) pairs assigned to one rank; replaces the chunk /
                                         round-robin bookkeeping of G:154,184,331-334,384-385 */

/* Algorithmic work of the last gram launch: complex128,
 * 8 real flops per complex multiply-add, each site tensor of both operands
 * read once per pair.  kernel_ms is the HIP-event time of the sweep kernel
 * on the context's stream (0 until the events have completed). */
typedef struct qk_stats {
  int64_t pairs;
  double flops;        /* sum over pairs and sites of 8*min(a*b*2*b' + 2*a*a'*b', a*b*2*a' + 2*b*a'*b'): the cheaper association per site */
  double padded_flops; /* the same with every bond rounded up to the MFMA tile (16)   */
  double bytes;        /* sum over pairs of 16*2*sum_k(a_k a_k+1 + b_k b_k+1) + 8     */
  double kernel_ms;    /* device time of the last sweep launch                        */
  int32_t grid;        /* workgroups launched                                         */
  int32_t max_bond;    /* largest padded bond among the two sets                      */
  int32_t kernel;      /* which sweep kernel ran: QK_KERNEL_* (see qk_kernel_name)    */
  int32_t precision;   /* 64 or 32: bits of a real of the sets it ran on              */
  /* A split sweep (sets of very different entanglement: the pairs whose sites all fit the site-fused kernel's smaller LDS
   * buffer are listed last and swept by its two-workgroups-per-CU shape, right after the launch named by `kernel`):
   * the second launch's share of the numbers above; all zero when the sweep was one launch. */
  int64_t second_pairs;
  double second_flops, second_padded_flops, second_bytes;
  double second_ms;     /* device time of the second launch (kernel_ms covers both)    */
  int32_t second_kernel; /* QK_KERNEL_FUSED2; QK_KERNEL_WAVE2 for a mixed set (its pairs of two small states); or QK_KERNEL_NONE */
  int32_t queues;        /* device work queues of the launch: 8 per class of pairs (one per XCD), 1 = one list */
  /* Tail accounting from device clocks (first workgroup start, first and last workgroup exit per launch): the share of a
   * launch's duration during which the chip was draining -- some workgroups had run out of work, the last one had not.
   * 0 for kernels that do not record it.  At a 1/8 share of the Gram the tails weigh eight times more than on one GPU.   */
  double tail_frac, second_tail_frac;
  /* Device time between the start of the call and the start of the sweep: the kernels that make a set's derived images
   * (interleaved image, edge blocks, merged steps) on the FIRST Gram of a set -- part of a cold Gram (the reference's
   * kernel_mat_time, G:322, 432-434, brackets set-up and tiles alike), ~0 afterwards.                                   */
  double derive_ms;
} qk_stats;

/* sweep kernels of qk_gram_values (qk_stats.kernel) */
#define QK_KERNEL_NONE 0
#define QK_KERNEL_WAVE 1    /* qk_sweep_wave_kernel: one pair per wavefront, bonds <= 16, fp64           */
#define QK_KERNEL_SMALL 2   /* qk_sweep_small_kernel: X, T in LDS, bonds <= 32                            */
#define QK_KERNEL_FUSED1 3  /* qk_sweep_fused_kernel<12, 2, 8192, 3, false>: site-fused sweep, one workgroup per CU, single tiles (QK_FUSED_DUAL=0) */
#define QK_KERNEL_FUSED2 4  /* qk_sweep_fused_kernel<8, 1, 4608, 4, false>: site-fused sweep, two workgroups per CU */
#define QK_KERNEL_RING 5    /* qk_sweep_ring_kernel: X, T in an L2-resident scratch                       */
#define QK_KERNEL_LAB 6     /* an experimental kernel (libqklab.so only)                                  */
#define QK_KERNEL_WAVE2 7   /* qk_sweep_wave2_kernel<3, double | float>: one pair per wavefront, bonds <= 32 (2 x 2 tiles), fp64 arithmetic on a complex128 or complex64 image */
#define QK_KERNEL_WAVE2_PLAIN 9 /* qk_sweep_wave2_kernel<0, double>: the same with plain loads instead of the LDS-DMA ring (QK_WAVE2=2) */
#define QK_KERNEL_FUSED_DUAL 8 /* qk_sweep_fused_dual_kernel<12, 8192, 3, false>: site-fused sweep, one workgroup per CU, pairs of tiles per wave (the default form of that shape) */
/* the DET forms of the three site-fused shapes (QK_DETERMINISTIC=1: contributions to X' added in a fixed order -- bit-reproducible Grams) */
#define QK_KERNEL_FUSED_DUAL_DET 10
#define QK_KERNEL_FUSED2_DET 11
#define QK_KERNEL_FUSED1_DET 12
/* the kernel's name as rocprofv3 prints it (without the "void " and the argument list) */
const char* qk_kernel_name(int32_t kernel, int32_t precision);

const char* qk_last_error(void);

/* number of gfx950 devices visible to the process (0 if none); replaces
 * cupy.cuda.runtime.getDeviceCount(), G:5,152 */
int qk_device_count(void);

/* ---- context --------------------------------------------------------------- */
int qk_ctx_create(int device_id, qk_ctx** out);
int qk_ctx_destroy(qk_ctx* ctx);
/* Run subsequent work on exactly this hipStream_t.  NULL is HIP's null (default) stream --
 * which is what torch.cuda.current_stream().cuda_stream returns for torch's default stream.
 * A new context starts on a private non-blocking stream; qk_ctx_use_own_stream() goes back to it. */
int qk_ctx_set_stream(qk_ctx* ctx, void* hip_stream);
int qk_ctx_use_own_stream(qk_ctx* ctx);
int qk_ctx_synchronize(qk_ctx* ctx);
/* Release the device memory a context keeps between calls: the sweep's per-workgroup scratch and the device builder's arena and
 * workspace (tens of GB at large bond caps; kept because allocating them costs seconds).  They come back on demand.          */
int qk_ctx_trim(qk_ctx* ctx);

/* ---- MPS sets ---------------------------------------------------------------
 * Upload n_states MPS of n_sites sites each.
 *   bond_dims    [n_states][n_sites+1] int32, bond_dims[s][0] = bond_dims[s][n_sites] = 1
 *   site_tensors [n_states][n_sites]   host pointers to complex128 (re,im interleaved)
 *                arrays of shape (chi_k, 2, chi_k+1) in `layout`
 * Replaces keeping pytket-cutensornet MPS objects (cupy tensors) alive on the
 * device between simulate() and vdot(), G:221-226, 290, 370-374.
 * Device layout: per site, split re/im planes [chi_k^][2][chi_k+1^] with every
 * bond zero-padded to a multiple of 16 (the f64 MFMA tile).                    */
int qk_mps_set_create(qk_ctx* ctx, int32_t n_states, int32_t n_sites, const int32_t* bond_dims,
                      const double* const* site_tensors, int32_t layout, qk_mps_set** out);
int qk_mps_set_destroy(qk_mps_set* set);
int qk_mps_set_info(const qk_mps_set* set, int32_t* n_states, int32_t* n_sites, int32_t* max_padded_bond,
                    int64_t* device_bytes);

/* ---- packed set images: the exchange format between ranks --------------------------------------------------
 * Replaces the pickled per-MPS sendrecv / send / recv of the reference's ring (G:341-352, 415-419): a rank packs
 * only ITS share of the states (qk_mps_set_create from host tensors, or qk_mps_set_from_built on the device), the
 * images are exchanged as flat buffers (one RCCL all-gather of the planes, the small tables beside it) and every rank
 * assembles the whole set with qk_mps_set_from_packed -- nothing is re-packed element by element, and with RCCL
 * nothing crosses PCIe.
 * qk_mps_set_image: the fp64 image of a set: *n_doubles doubles at *planes_dev (valid while the set lives); tables
 *   copied to dims_true[n_states][n_sites+1] and offsets[n_states][n_sites] (re-plane offsets in doubles); each may be NULL.
 * qk_mps_set_from_packed: a set of n_states states whose site tensors lie in the device buffer `planes_dev`
 *   (n_doubles doubles, copied device-to-device into memory the new set owns) at the given offsets, each in the
 *   padded split-plane layout of qk_mps_set_create (what qk_mps_set_image hands out).
 * qk_mps_set_copy_image: the planes copied into `dst` (n_doubles doubles; a device buffer -- for example the send
 *   buffer of the all-gather -- or a host buffer: the direction is inferred from the pointer).
 * In qk_mps_set_from_packed `planes` may likewise be a device or a host buffer.                                     */
int qk_mps_set_image(const qk_mps_set* set, int64_t* n_doubles, const double** planes_dev, int32_t* dims_true, int64_t* offsets);
int qk_mps_set_copy_image(const qk_mps_set* set, double* dst, int64_t n_doubles);
int qk_mps_set_from_packed(qk_ctx* ctx, int32_t n_states, int32_t n_sites, const int32_t* dims_true, const int64_t* offsets,
                           const double* planes_dev, int64_t n_doubles, qk_mps_set** out);

/* Precision of a set's device image: 64 (complex128 planes, what qk_mps_set_create builds) or 32.
 * qk_mps_set_to_f32 makes a complex64 copy on the device (same layout, same element offsets).  A sweep whose two
 * sets are fp32 runs on v_mfma_f32_16x16x4_f32 with fp32 X/T scratch; outputs stay double.  The reference never
 * leaves fp64 (G:141-144 does not set float_precision); this is SURVEY.md section 8f row N4, the fp32-vs-fp64
 * tolerance sweep its cfg5 asks for.  Mixing an fp32 and an fp64 set in one call is QK_EINVAL.                  */
int qk_mps_set_precision(const qk_mps_set* set);
int qk_mps_set_to_f32(qk_ctx* ctx, const qk_mps_set* src, qk_mps_set** out);

/* Host-only packing helper used by qk_mps_set_create (exported so that the
 * packing can be unit-tested without a GPU).  Writes one state's padded planar
 * image; `out` must hold qk_pack_state_size() doubles.                         */
int64_t qk_pack_state_size(int32_t n_sites, const int32_t* bond_dims);
int qk_pack_state(int32_t n_sites, const int32_t* bond_dims, const double* const* site_tensors, int32_t layout,
                  double* out, int64_t* site_offsets /* [n_sites] offsets of each re-plane, in doubles */);

/* ---- plans (host side; no GPU needed) ---------------------------------------
 * Enumerate the pairs of the Gram in tiles of `block` x `block` states (block <= 0:
 * one tile = the whole Gram), sort each tile by decreasing estimated cost, deal the
 * pairs to `world_size` ranks in serpentine order (0..W-1, W-1..0, ...: equal counts
 * +-1 and flops within a fraction of a percent) and keep rank `rank`'s share, itself
 * ordered heaviest first for the device-side work queue.  x_dims / y_dims are the bond_dims tables of the two sets
 * (y_dims = NULL with QK_PLAN_SYMMETRIC).  Pairs are (x index i, y index j);
 * the Gram entry is K[j][i]  (rows = Y, cols = X: G:387, J:106).               */
int qk_plan_create(int32_t n_sites, int32_t nx, const int32_t* x_dims, int32_t ny, const int32_t* y_dims,
                   uint32_t flags, int32_t world_size, int32_t rank, int32_t block, qk_plan** out);
int qk_plan_destroy(qk_plan* plan);
int64_t qk_plan_num_pairs(const qk_plan* plan);       /* this rank's pairs              */
int64_t qk_plan_total_pairs(const qk_plan* plan);     /* all ranks                      */
int64_t qk_plan_max_pairs_per_rank(const qk_plan* plan);
const int32_t* qk_plan_pairs(const qk_plan* plan);    /* [num_pairs][2] = (i, j), host  */
int qk_plan_stats(const qk_plan* plan, qk_stats* out); /* algorithmic flops/bytes of this rank's share */
/* Pairs [qk_plan_first_run, num_pairs) are the SECOND RUN of the list, swept by its own launch right behind the first:
 *   - a mixed set (states with every bond <= 32 next to larger ones, at least 64 pairs of two small states): those small-small
 *     pairs, for the one-pair-per-wavefront sweep -- kernel choice is per PAIR, one large state does not drag the rest along;
 *   - otherwise the pairs with >= QK_PLAN_SPLIT (environment, default 0.75) of their padded work in sites that fit the site-fused
 *     kernel's smaller LDS buffer, for its two-workgroups-per-CU shape.
 * == num_pairs when the plan holds (nearly) one class only: the launch then takes the two-workgroups-per-CU shape when >= 75 % of
 * the padded work sits in sites that fit its buffer AND >= 50 % in sites of at most the narrow size (QK_PLAN_FIT, 3072 elements of
 * X), the 12-wave dual shape otherwise (e.g. a set whose bonds were cut at 64: every site 4 x 4 tiles).                   */
int64_t qk_plan_first_run(const qk_plan* plan);
/* XCD-aware work queues (default plans; QK_PLAN_XCD=0 in the environment or an explicit `block` gives the flat cost-ordered
 * list).  The states are sorted by weight, the Gram is cut into tiles of QK_PLAN_TILE x QK_PLAN_TILE (default 8 x 8) pairs in
 * that order -- the pairs of a tile share their x and y states and cost about the same --, the tiles are dealt heaviest
 * first to the least loaded rank and, per run of this rank's list, to 8 queues: queue s of the first run = pairs
 * [qstart[s], qstart[s+1]), s = 0..7, of the second run s = 8..15 (qstart[8] = qk_plan_first_run, qstart[16] = num_pairs).
 * On the device a workgroup reads the id of the XCD it runs on (8 XCDs with a private 4 MiB L2 each) and drains that queue
 * first, then steals from the others: the workgroups that share an L2 stream the same few states.  Replaces the chunk
 * bookkeeping of G:154, 331-334, 384-385 like the rest of the plan.  Returns the number of queues (16, or 1 for a flat
 * list: then qstart[0] = 0 and qstart[1..16] = num_pairs); qstart may be NULL.                                            */
int qk_plan_queues(const qk_plan* plan, int64_t* qstart /* [17] */);
/* EDGE BLOCKS: how many sites at either end of the chain the site-fused sweep takes from per-state blocks instead of walking
 * them (0: none).  While the bonds still grow like 2^k, contracting the first k sites of each state across their physical legs
 * into one matrix L[s][a] and starting a pair's environment as X = Ly^T conj(Lx) -- one product with K = 2^k -- is cheaper than k
 * sites of the chain and saves their per-site costs; likewise at the right end, where the overlap is sum X . (Ry^T conj(Rx)).
 * The planner picks the k (4..9) that minimises a cost model over a sample of the pairs (qkgram.hip: choose_edge_k; QK_EDGE=0
 * disables, QK_EDGE=k forces); the blocks are made once per set on first use (2^k x padded bond complex numbers per state and
 * end; counted by qk_mps_set_info).  This is the host-side choice of the contraction order at the ends of the chain (north star;
 * reference call site G:380); the algorithmic flop count of qk_stats does not change.                                        */
int32_t qk_plan_edge_sites(const qk_plan* plan);
/* The plans of ALL world_size ranks from one cost pass (a one-process communicator, qk_gram_sharded: pricing the Gram's pairs and
 * dealing its tiles is the same work for every rank and is done once; each rank then prices only its own share).  out[world_size];
 * plan r equals what qk_plan_create(..., world_size, r, 0, ...) returns.  Replaces G:154, 331-334 like qk_plan_create.      */
int qk_plan_create_all(int32_t n_sites, int32_t nx, const int32_t* x_dims, int32_t ny, const int32_t* y_dims,
                       uint32_t flags, int32_t world_size, qk_plan** out);
/* What the plan cost and what it could save.  plan_ms = host wall time of qk_plan_create for this plan (the reference's
 * kernel_mat_time, G:322, 432-434, brackets the set-up of the tiling phase too: the planner is on the cold path of every Gram),
 * threads = host threads it used, tile_reuse_bytes = the bytes of this rank's share if every state were read once per plan
 * tile (8 x 8 pairs) it takes part in -- the tile-reuse lower bound of SURVEY 8d, beside qk_stats.bytes (every state read
 * once per PAIR).  Any out pointer may be NULL.                                                                          */
int qk_plan_cost(const qk_plan* plan, double* plan_ms, int32_t* threads, double* tile_reuse_bytes);
/* MERGED STEPS (no entry point: part of qk_gram_values; QK_MERGE=0 disables).  Between the edge blocks the site-fused sweep may walk
 * two neighbouring sites as ONE step: the set holds, beside its plain image, the chain's sites contracted in twos over the bond between
 * them (tensors [l][4][r], made once per set on first use and counted by qk_mps_set_info), and a workgroup decides per pair and step:
 * the merged tensor where that keeps an LDS-resident step in the LDS and costs no more padded work than the two sites, the two plain
 * sites otherwise (a dip of the bond).  Same matrix work where the bonds are level, half the barriers, set-ups and stream turn-arounds;
 * again a host-prepared choice of the contraction order (reference call site G:380).  qk_stats' algorithmic flops and bytes stay those
 * of the plain chain.                                                                                                          */

/* ---- the hot path -----------------------------------------------------------
 * qk_gram_values: for every pair p of the plan compute z_p = <x_i|y_j> and write
 *     values_dev[p] = |z_p|^2                 (G:380-383, J:106)
 *     z_dev[2p], z_dev[2p+1] = re, im of z_p  (optional, may be NULL)
 * One persistent launch; returns after enqueueing (asynchronous).
 * yset = NULL means Y is X.
 * The sweep kernel is chosen from the two sets' largest padded bond and precision:
 *   16, fp64            one pair per wavefront, entirely in registers (qk_sweep_wave_kernel);
 *   <= 32, fp64         one pair per wavefront with 2 x 2 register tiles, fed through a per-wave LDS-DMA ring
 *                       (qk_sweep_wave2_kernel<3, double>);
 *   <= 32, complex64    the SAME kernel on the complex64 image: single-precision storage, fp64 arithmetic
 *                       (qk_sweep_wave2_kernel<3, float>); with QK_WAVE2=0 | 2 the LDS-resident small-bond sweep in
 *                       complex64 arithmetic on the fp32 matrix cores (qk_sweep_small_kernel<float>);
 *   larger, fp64        the site-fused sweep (X in LDS, T in registers, qk_fused.h), bonds up to 512;
 *   larger, complex64   and fp64 bonds > 512: the ring sweep (X / T in an L2-resident scratch; complex64 arithmetic on
 *                       the fp32 matrix cores for complex64 sets).
 * All compute the same chain of complex GEMMs on the matrix cores and agree to rounding (tests/test_gpu_parity.py);
 * QK_WAVE=0 / QK_WAVE2=0 / QK_SMALL=0 / QK_FUSED=0 in the environment fall back to the next more general one, QK_FUSED=2
 * uses the fused sweep from bond 17, QK_FUSED_WGS=1|2 fixes its workgroups per CU.                                   */
int qk_gram_values(qk_ctx* ctx, const qk_mps_set* xset, const qk_mps_set* yset, const qk_plan* plan,
                   double* values_dev, double* z_dev);

/* Synchronous form for host communicators (mpi4py, gloo): the same sweep, values copied back
 * into values_host[num_pairs] (and z_host[2*num_pairs] unless NULL) before returning.        */
int qk_gram_values_host(qk_ctx* ctx, const qk_mps_set* xset, const qk_mps_set* yset, const qk_plan* plan,
                        double* values_host, double* z_host);

/* Scatter packed values into the dense matrix K (device, row-major, leading
 * dimension ld): K[j][i] = v, and K[i][j] = v as well when `mirror` != 0
 * (G:387, 390-395).  pairs_dev: [n][2] int32 on the device.                    */
int qk_scatter(qk_ctx* ctx, const int32_t* pairs_dev, const double* values_dev, int64_t n, double* k_dev,
               int64_t ld, int32_t mirror);

/* Convenience, synchronous: whole Gram of xset (cols) vs yset (rows, NULL =
 * symmetric) into a host matrix out[ny][ld].  Replaces the double loop
 * G:372-400 plus the final reduce G:428 for a single process.                  */
int qk_gram_host(qk_ctx* ctx, const qk_mps_set* xset, const qk_mps_set* yset, double* out, int64_t ld);

/* Convenience, synchronous: complex overlaps z[j][i] = <x_i|y_j> (re,im
 * interleaved, out[ny][nx][2]); the batched form of MPS.vdot, G:380.           */
int qk_overlaps_host(qk_ctx* ctx, const qk_mps_set* xset, const qk_mps_set* yset, double* out);

int qk_get_stats(qk_ctx* ctx, qk_stats* out);

/* ---- profiler ranges -----------------------------------------------------------------------------------------
 * roctx ranges (rocprofv3 --marker-trace) named "qk:build", "qk:upload", "qk:sweep", "qk:scatter", "qk:allgather_values",
 * "qk:allgather_sets" are opened by the library around its own phases -- the reference's MPI.Wtime() sites G:209-231
 * (circuit simulation), G:379-381 (vdot) and G:341-352 (round robin) --; callers use the same two entry points for theirs.
 * No-ops unless a profiler is attached (or QK_ROCTX=1): the marker library is resolved at first use.                  */
int qk_range_push(const char* name);
int qk_range_pop(void);

/* ---- multi-GPU: one process, k MI355X of one node, RCCL over xGMI ----------------------------------------------
 * Replaces the reference's communicator plumbing: rank / chunk bookkeeping G:149-199 (one MPI process per GPU, device =
 * rank % n_devices, G:152), the ring of pickled MPS G:341-352, 415-419 and the final comm.reduce(kernel_mat, SUM) G:428
 * (which only ever gathers: every rank's matrix is zero outside its own tiles).
 *   qk_comm_init_all      a context per device (device_ids = NULL: devices 0..n-1) and an RCCL communicator clique
 *                         (ncclCommInitAll); RCCL is loaded at run time (librccl.so.1), the single-GPU entry points do not
 *                         need it.  qk_comm_ctx(comm, r) is rank r's context: sets for rank r are created on it.
 *   qk_mps_set_allgather  local[r] = the states [lo[r], lo[r] + n_r) that rank r built or uploaded (NULL = an empty share);
 *                         ONE ncclAllGather of the packed images and full_out[r] = the whole set of `total` states on
 *                         every device (the caller destroys them).
 *   qk_gram_sharded       xsets[r] / ysets[r] = the whole set(s) on device r (ysets = NULL: symmetric Gram).  Every device
 *                         sweeps rank r's share of the plan (qk_plan_create(world = n, rank = r)) in one launch, the packed
 *                         values meet in ONE ncclAllGather, every device scatters (and mirrors) its own dense K; rank 0's K
 *                         is copied to out_host[ny][ld] (may be NULL) and the call returns when every device is done.
 *                         Plans and buffers are kept while the same sets come back.
 *   qk_comm_device_gram   device r's dense K of the last call (valid until the next one);  qk_comm_stats: rank r's sweep
 *                         statistics and the device time between enqueueing the all-gather on rank 0's stream and its
 *                         completion (includes waiting for the slowest rank; the reference's r0_RR_recv key).           */
typedef struct qk_comm qk_comm;
int qk_comm_init_all(int32_t n_devices, const int32_t* device_ids, qk_comm** out);
int qk_comm_destroy(qk_comm* comm);
int32_t qk_comm_size(const qk_comm* comm);
qk_ctx* qk_comm_ctx(qk_comm* comm, int32_t rank);
int qk_mps_set_allgather(qk_comm* comm, qk_mps_set* const* local, const int32_t* lo, int32_t total, qk_mps_set** full_out);
int qk_gram_sharded(qk_comm* comm, qk_mps_set* const* xsets, qk_mps_set* const* ysets, double* out_host, int64_t ld);
int qk_comm_device_gram(qk_comm* comm, int32_t rank, const double** k_dev);
int qk_comm_stats(const qk_comm* comm, int32_t rank, qk_stats* out, double* allgather_ms);

/* Device self-test of the f64 MFMA fragment maps the kernels rely on (returns
 * 0 if the 16x16x4 product of two known matrices matches the host result). */
int qk_selftest_mfma(qk_ctx* ctx);

/* ---- device MPS builder ------------------------------------------------------------
 * Replaces simulate(libhandle, circ, SimulationAlgorithm.MPSxGate, config) (G:141-144, 221, 263) for the
 * ansatz gate program: all data points share the gate structure (op, q0: n_ops entries; 0 = H, 1 = Rz,
 * 2 = XXPhase on (q0, q0+1), 3 = SWAP on (q0, q0+1)) and differ in the half-turn angles alpha[n_states][n_ops].
 * One persistent launch; per two-qubit gate a one-sided Jacobi SVD and the truncation rule of the host builder:
 * drop the trailing singular values whose squared weight stays <= trunc_budget (= 1 - truncation_fidelity,
 * G:141-144; criterion as KernelPkg.jl:68), values <= value_of_zero never count.  max_bond bounds every bond
 * (QK_EINVAL if a state outgrows it).  The result stays on the device until downloaded: per state and site a
 * complex128 tensor [chi_l][2][chi_r] row-major, sites back to back, state s at offsets[s] (complex elements).    */
typedef struct qk_built qk_built;
#define QK_BUILD_PARTIAL 1u /* a state that outgrows max_bond is dropped (its fidelity reads -1, it has no tensors: build it
                             * elsewhere) instead of failing the call; its workgroup stops at the offending gate        */
#define QK_BUILD_TRUNCATE 2u /* max_bond is a bond CAP: at most max_bond singular values survive a gate (the `chi` of pytket-cutensornet's
                             * Config, reference G:141-144, which the reference leaves unset); the weight it costs goes into the
                             * state's fidelity like any other truncation.  Without it a state that needs more is an error (or, with
                             * QK_BUILD_PARTIAL, dropped).                                                                          */
int qk_build_mps(qk_ctx* ctx, int32_t n_states, int32_t n_qubits, int32_t n_ops, const int8_t* op, const int32_t* q0,
                 const double* alpha, double trunc_budget, double value_of_zero, int32_t max_bond, uint32_t flags,
                 qk_built** out);
/* dims[n_states][n_qubits+1], fidelity[n_states], offsets[n_states], total complex elements, kernel time; any may be NULL */
int qk_built_info(const qk_built* built, int32_t* dims, double* fidelity, int64_t* offsets, int64_t* total_complex,
                  double* kernel_ms);
int qk_built_download(const qk_built* built, double* host /* 2 * total_complex doubles (re, im interleaved) */);
/* The built states as a set of the Gram engine (the image qk_mps_set_create makes from host tensors), packed on the
 * device: nothing crosses PCIe between the builder and the sweep.  The set is independent of `built` afterwards.  */
int qk_mps_set_from_built(qk_ctx* ctx, const qk_built* built, qk_mps_set** out);
int qk_built_destroy(qk_built* built);
/* Diagnostic: the builder's Jacobi primitive on one host matrix a[p][q] (complex128 row-major, overwritten by A V);
 * v_out[q][q], sig_out[q] = column norms of A V, ord_out[q] = columns by decreasing norm.                          */
int qk_debug_jacobi(qk_ctx* ctx, int32_t p, int32_t q, double* a_inout, double* v_out, double* sig_out, int32_t* ord_out);
/* Diagnostic: the builder's factorisation for matrices beyond its LDS working set -- columns sorted, R by Gram-Schmidt, block
 * Jacobi of R^H on the f64 matrix cores, W = A V -- on one host matrix a[p][q] (16 <= q <= 1024, p <= 1024): a <- W = A V
 * (columns beyond the numerical rank are zero), v_out[q][q], sig_out[q] (0 beyond the rank), ord_out[q] as above;
 * stats_out[6] (may be NULL) = sweeps, then device time in 100 MHz ticks: all, sort + copy, Gram-Schmidt, sweeps, V and W. */
int qk_debug_jacobi_precond(qk_ctx* ctx, int32_t p, int32_t q, double* a_inout, double* v_out, double* sig_out, int32_t* ord_out, int32_t* stats_out);

#ifdef __cplusplus
}
#endif
#endif /* QKGRAM_H */
