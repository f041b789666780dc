/*
 * pyqsm_hip.h — C-ABI of libpyqsm_hip.so, the MI355X (gfx950) implementation of
 * pyQSM's point-cloud geometry hot path.
 *
 * The reference (wischmcj/pyQSM) is pure Python and has no FFI of its own; the
 * seam it offers is a handful of Python call sites into third-party engines.
 * Every entry point below names the reference call it stands in for
 * (file:line relative to the reference root).  The Python wrappers in
 * pyqsm_amd/ bind these symbols with ctypes and expose the reference's own
 * function names and signatures on top.
 *
 * Conventions
 *   - plain pointers and sizes only; caller owns every buffer it passes in;
 *     arrays are C-contiguous; no exceptions cross the ABI
 *   - return 0 on success, a negative PYQSM_E* code on failure; the message is
 *     available (per thread) from pyqsm_last_error()
 *   - "host" entry points take host pointers and stage through HBM themselves;
 *     "_dev" entry points take device pointers (HBM-resident input/output,
 *     obtained from pyqsm_dev_malloc or any hipMalloc'd allocation) and are
 *     asynchronous on the library's per-device stream until pyqsm_sync()
 *   - the library owns one HIP stream and one scratch arena per device and
 *     calling thread: threads may call concurrently, each is ordered on its own
 *     stream (pyqsm_stream / pyqsm_sync / pyqsm_prof_* act on the caller's)
 */
#ifndef PYQSM_HIP_H
#define PYQSM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYQSM_OK 0
#define PYQSM_EINVAL (-1)   /* bad argument (null pointer, negative size, ...)   */
#define PYQSM_EHIP (-2)     /* a HIP runtime call failed                         */
#define PYQSM_ENODEV (-3)   /* no usable GPU / device index out of range         */
#define PYQSM_ERANGE (-4)   /* input outside what the kernels are built for      */
#define PYQSM_ENOMEM (-5)   /* device or host allocation failed                  */
#define PYQSM_ENOCONV (-6)  /* CG stopped (max_it / stagnation) above rtol; best
                               iterate is still returned                         */

#define PYQSM_MISS_PRIM 0xFFFFFFFFu /* primitive id reported for a ray that hits nothing */

/* ---- library / device management ------------------------------------- */

/* Number of visible GPUs (0 when none); never fails. */
int pyqsm_device_count(void);
/* Create the calling thread's context for `device` (stream + arena). Idempotent. */
int pyqsm_init(int device);
/* Destroy every context created by pyqsm_init. */
int pyqsm_shutdown(void);
/* Message of the last failure on the calling thread ("" when none). */
const char* pyqsm_last_error(void);
/* "pyqsm_hip <version> gfx950" */
const char* pyqsm_version(void);
/* Block until the library stream of `device` has drained. */
int pyqsm_sync(int device);
/* The library's hipStream_t for `device`, as an opaque pointer (for callers
 * that order their own work against it). */
void* pyqsm_stream(int device);

/* HBM buffers for the _dev entry points. */
int pyqsm_dev_malloc(int device, size_t bytes, void** out);
int pyqsm_dev_free(int device, void* p);
int pyqsm_h2d(int device, void* dst_dev, const void* src_host, size_t bytes);
int pyqsm_d2h(int device, void* dst_host, const void* src_dev, size_t bytes);
/* Release memory returned through an `**` out-parameter of a host entry point. */
void pyqsm_free(void* p);
/* A host buffer from the same pool those out-parameters come from: page-locked when it is a
 * megabyte or more (plain malloc otherwise or when page-locking fails), released with pyqsm_free.
 * Results written into such a buffer leave the device at link speed and without blocking the
 * caller's thread; into pageable memory every 24 MB of pyqsm_extract_skeleton's per-step shifts
 * cost 12 ms of staging (0.26 s of a 2.8 s loop). NULL when out of memory. */
void* pyqsm_host_alloc(size_t bytes);

/* HIP-event timers around named kernel groups on the library stream.
 * Disabled by default (zero overhead); bench.py switches them on to measure
 * the dominant kernel's average launch duration live. on = 1: phase timers and the
 * dominant kernels of each path; on = 2: also single kernels inside the solver's
 * iteration loops (k_bspmv_f, k_down_l0, k_up_l0: adds two event records per launch)
 * and the pair-test counter of the DBSCAN core pass ("core_pair_tests": launches =
 * lane-tests executed, ms = 0). */
int pyqsm_prof_enable(int device, int on);
int pyqsm_prof_reset(int device);
/* Sum of elapsed ms and number of launches recorded under `name`
 * since the last reset. Synchronises the stream. */
int pyqsm_prof_get(int device, const char* name, double* total_ms, int64_t* launches);

/* ---- ray x triangle sweep -------------------------------------------- */
/*
 * Closest-hit ray casting against a triangle soup, brute force (no BVH).
 * Stands in for open3d.t.geometry.RaycastingScene.add_triangles + cast_rays as
 * called at pyQSM/viz/ray_casting.py:275-279 (also :218-225, :316-319).
 *   verts  f32 [V,3]     tris  i32 [T,3] (indices into verts)
 *   rays   f32 [R,6] = (ox,oy,oz,dx,dy,dz); d need not be unit, t is in units of |d|
 *   t_hit  f32 [R]  (+inf on miss)      prim_id u32 [R] (PYQSM_MISS_PRIM on miss)
 *   uv     f32 [R,2] or NULL; hit = (1-u-v)*v0 + u*v1 + v*v2 (ray_casting.py:172-180)
 *   Ties in t go to the lowest triangle index. A hit needs t > 0.
 */
int pyqsm_cast_rays(const float* verts, int64_t V, const int32_t* tris, int64_t T,
                    const float* rays, int64_t R,
                    float* t_hit, uint32_t* prim_id, float* uv, int32_t device);

/* Device-resident form. `tri9` is the expanded mesh produced by
 * pyqsm_expand_tris_dev: f32 [T,12] = (v0.xyz, e1.xyz, e2.xyz, 0,0,0). */
int pyqsm_expand_tris_dev(const float* verts_dev, int64_t V, const int32_t* tris_dev,
                          int64_t T, float* tri12_dev, int32_t device);
int pyqsm_cast_rays_dev(const float* tri12_dev, int64_t T, const float* rays_dev, int64_t R,
                        float* t_hit_dev, uint32_t* prim_id_dev, float* uv_dev,
                        int32_t device);

/* All intersections (RaycastingScene.list_intersections, ray_casting.py:168) and
 * crossing counts (compute_occupancy, ray_casting.py:65-69 = odd count).
 *   counts i32 [R] = number of triangles each ray crosses with t > 0.
 *   If hits_cap > 0: up to hits_cap records (ray_id u32, prim_id u32, t f32, u f32, v f32)
 *   are written, ordered by ray id then triangle id; *n_hits = total found. */
int pyqsm_list_intersections(const float* verts, int64_t V, const int32_t* tris, int64_t T,
                             const float* rays, int64_t R, int32_t* counts,
                             uint32_t* ray_ids, uint32_t* prim_ids, float* t, float* uv,
                             int64_t hits_cap, int64_t* n_hits, int32_t device);

/*
 * Unsigned distance from query points to the mesh and the index of the closest
 * triangle — open3d RaycastingScene.compute_distance, which `mri`
 * (pyQSM/viz/ray_casting.py:237-260) reaches through compute_signed_distance; the sign
 * is the occupancy of pyqsm_list_intersections (inside = negative), applied by the
 * wrapper. Brute force, fp32; lowest triangle index on ties; T = 0 gives +inf.
 *   qry f32 [Q,3]; dist f32 [Q]; prim u32 [Q].
 */
int pyqsm_point_mesh_distance(const float* verts, int64_t V, const int32_t* tris, int64_t T,
                              const float* qry, int64_t Q, float* dist, uint32_t* prim,
                              int32_t device);

/* ---- the ray sweep over several GPUs (RCCL over xGMI) ------------------ */
/*
 * One process driving n_devices GPUs of the node (stands in for
 * scene.cast_rays(rays) at pyQSM/viz/ray_casting.py:275-279 on more than one GPU):
 * the mesh is expanded on device 0 and replicated with ncclBroadcast, rays are split
 * into contiguous shards (sizes differ by at most one, in device order), every device
 * sweeps its shard, and ncclAllGather leaves the full (t, prim[, uv]) on every device;
 * device 0's copy goes to the host arrays. Results are identical to pyqsm_cast_rays
 * (rays are independent). Same argument meaning as pyqsm_cast_rays; n_devices <= 0
 * means every visible GPU; n_devices = 1 still goes through RCCL.
 */
int pyqsm_cast_rays_multi(const float* verts, int64_t V, const int32_t* tris, int64_t T,
                          const float* rays, int64_t R,
                          float* t_hit, uint32_t* prim_id, float* uv, int32_t n_devices);

/* The contiguous shard [*begin, *end) of `rank` among `world` ranks over n items, as
 * pyqsm_cast_rays_multi splits the rays (sizes differ by at most one, in rank order; needs no GPU). */
int pyqsm_shard_bounds(int64_t n, int32_t world, int32_t rank, int64_t* begin, int64_t* end);

/*
 * One process PER GPU (torchrun-style launches): a communicator per process and the
 * data-path collectives on the library stream of the rank's device. Rank 0 creates the
 * id and ships its PYQSM_COMM_ID_BYTES bytes to the other ranks by any means (a file, a
 * socket, a CPU rendezvous); pyqsm_comm_init_rank is collective. The _dev calls take
 * device pointers and are asynchronous until pyqsm_sync(device).
 *   broadcast: in place, `bytes` from rank `root`;  all_gather: recv holds world *
 *   bytes_per_rank, rank r's block at offset r * bytes_per_rank (send may alias its
 *   own block);  all_reduce_max: host scalar in/out, returns when every rank has the
 *   maximum (doubles as a barrier).
 */
#define PYQSM_COMM_ID_BYTES 128
int pyqsm_comm_unique_id(uint8_t* id);
int pyqsm_comm_init_rank(const uint8_t* id, int32_t world, int32_t rank, int32_t device);
int pyqsm_comm_finalize(void);
/* world = 0 when no communicator exists */
int pyqsm_comm_info(int32_t* world, int32_t* rank, int32_t* device);
int pyqsm_comm_broadcast_dev(void* buf_dev, int64_t bytes, int32_t root);
int pyqsm_comm_all_gather_dev(const void* send_dev, void* recv_dev, int64_t bytes_per_rank);
int pyqsm_comm_all_reduce_max(double* value);

/* ---- eps-neighbourhood clustering (DBSCAN) ---------------------------- */
/*
 * Stands in for sklearn.cluster.DBSCAN(eps, min_samples).fit(points) at
 * pyQSM/math_utils/fit.py:223 and open3d PointCloud.cluster_dbscan at
 * pyQSM/geometry/point_cloud_processing.py:185,209.
 *   xyz f64 [n,3]; labels i64 [n] (-1 = noise); is_core u8 [n] (may be NULL)
 * Semantics (bit-exact with scikit-learn): core <=> #{j : d2(i,j) <= eps*eps} >=
 * min_pts counting i itself, d2 = ((dx*dx + dy*dy) + dz*dz) in fp64 without
 * contraction; clusters = connected components of the core-core eps graph,
 * numbered by ascending smallest core index; a border point takes the smallest
 * cluster label among its core neighbours.
 */
int pyqsm_dbscan(const double* xyz, int64_t n, double eps, int32_t min_pts,
                 int64_t* labels, uint8_t* is_core, int32_t device);
int pyqsm_dbscan_dev(const double* xyz_dev, int64_t n, double eps, int32_t min_pts,
                     int64_t* labels_dev, uint8_t* is_core_dev, int64_t* n_clusters,
                     int32_t device);
/*
 * The same with the neighbourhood's boundary as a parameter. radius_inclusive != 0:
 * d2 <= eps*eps, scikit-learn's (the two calls above). radius_inclusive == 0: d2 < eps*eps —
 * what open3d cluster_dbscan (pyQSM/geometry/point_cloud_processing.py:185,209) computes if
 * nanoflann's radius search compares strictly (Open3D is not installable
 * here, so which of the two it is stays unpinned: this is the switch for whoever can check).
 * Everything else — self counted, numbering, border rule — is unchanged.
 */
int pyqsm_dbscan_ex(const double* xyz, int64_t n, double eps, int32_t min_pts,
                    int32_t radius_inclusive, int64_t* labels, uint8_t* is_core, int32_t device);
int pyqsm_dbscan_dev_ex(const double* xyz_dev, int64_t n, double eps, int32_t min_pts,
                        int32_t radius_inclusive, int64_t* labels_dev, uint8_t* is_core_dev,
                        int64_t* n_clusters, int32_t device);

/* ---- k nearest neighbours --------------------------------------------- */
/*
 * Exact kNN over one cloud (query set = data set). Stands in for the neighbour
 * search inside robust_laplacian.point_cloud_laplacian (pyQSM/geometry/
 * skeletonize.py:253-255) and scipy cKDTree.query (pyQSM/geometry/
 * reconstruction.py:238-240).
 *   idx i32 [n,k], d2 f64 [n,k] (squared distance), ascending by (d2, index);
 *   exclude_self != 0 drops the query point itself. Rows are padded with
 *   idx = n, d2 = +inf when the cloud has fewer than k (other) points.
 */
int pyqsm_knn(const double* xyz, int64_t n, int32_t k, int32_t exclude_self,
              int32_t* idx, double* d2, int32_t device);
int pyqsm_knn_dev(const double* xyz_dev, int64_t n, int32_t k, int32_t exclude_self,
                  int32_t* idx_dev, double* d2_dev, int32_t device);

/* ---- RANSAC circle / cylinder ------------------------------------------ */
/*
 * Stands in for pyransac3d.Circle().fit / Cylinder().fit as called at
 * pyQSM/math_utils/fit.py:277-283. The caller supplies the 3-point samples
 * (the reference draws them from Python's unseeded `random`), one row per
 * hypothesis, so that runs are reproducible.
 *   pts f64 [n,3]; triples i64 [H,3]; shape 0 = circle, 1 = cylinder
 *   center[3], axis[3], *radius: model of the winning hypothesis (first
 *   hypothesis with a strictly larger inlier count wins)
 *   inliers i64 [n] capacity, ascending; *n_inliers = count; *best = winning row
 *   (-1 and n_inliers = 0 when no hypothesis has an inlier).
 */
int pyqsm_ransac(const double* pts, int64_t n, const int64_t* triples, int64_t H,
                 int32_t shape, double thresh, double center[3], double axis[3],
                 double* radius, int64_t* inliers, int64_t* n_inliers, int64_t* best,
                 int32_t device);
/* The two halves separately: models f64 [H,8] = (cx,cy,cz, ax,ay,az, r, valid). */
int pyqsm_ransac_models(const double* pts, int64_t n, const int64_t* triples, int64_t H,
                        double* models, int32_t device);
int pyqsm_ransac_count(const double* pts, int64_t n, const double* models, int64_t H,
                       int32_t shape, double thresh, int32_t* counts, int32_t device);
/*
 * Many independent fits in one call — the z-slices of a stem, each fitted like
 * pyQSM/math_utils/fit.py:277-283 fits one cluster (a call per slice is 0.6 ms of
 * launches and round trips for 0.05 ms of work). S point sets stacked in pts
 * (seg_start i64 [S+1], from 0 to n), H hypotheses per set: triples i64 [S,H,3]
 * with indices LOCAL to the set (a set with fewer than three points gets rows of -1).
 * Per set what pyqsm_ransac returns: centers f64 [S,3], axes f64 [S,3], radii f64 [S],
 * best i64 [S] (-1: no hypothesis with an inlier), n_inliers i64 [S], and the inliers
 * as ascending local indices, set after set, in inliers i64 [n] (capacity).
 */
int pyqsm_ransac_batch(const double* pts, int64_t n, const int64_t* seg_start, int64_t n_seg,
                       const int64_t* triples, int64_t H, int32_t shape, double thresh,
                       double* centers, double* axes, double* radii, int64_t* inliers,
                       int64_t* n_inliers, int64_t* best, int32_t device);

/* ---- Laplacian-contraction solve --------------------------------------- */
/*
 * One contraction solve of pyQSM/geometry/skeletonize.py:148-180
 * (least_squares_sparse): the reference stacks A = [L W_L ; W_H] (:164, the weights
 * scale the COLUMNS of L), so this minimises |L W_L x|^2 + |W_H (x - p)|^2 per
 * coordinate, i.e. (W_L L' L W_L + W_H^2) x = W_H^2 p, by a preconditioned conjugate
 * gradient over 3 right-hand sides that never forms L'L.
 *   L as CSR (indptr i32 [n+1], indices i32 [nnz], vals f64 [nnz]);
 *   wl, wh, f64 [n]; pts f64 [n,3] (also the start vector); out f64 [n,3]
 *   positive wl that is constant along every edge of L (uniform, as extract_skeleton
 *   produces it, or one value per connected block of a block-diagonal L): flexible
 *   CG preconditioned by B^-2, B = W_L L + W_H; stops when the preconditioned
 *   residual |B^-2 r| / |x|, an estimate of the relative error of x (B^-2 A has
 *   its spectrum in [1/2, 1] for uniform W_H), is <= rtol for every coordinate;
 *   any other wl: Jacobi-CG, stops when |r|/|b| <= rtol. Also stops after max_it sparse passes or when
 *   the estimate has stopped improving (attainable accuracy); in those two cases
 *   returns PYQSM_ENOCONV with the best iterate in `out`. `resid` always receives
 *   the true relative residuals |r|/|b| of the returned iterate.
 */
int pyqsm_lbc_solve(const int32_t* indptr, const int32_t* indices, const double* vals,
                    int64_t n, const double* wl, const double* wh, const double* pts,
                    double rtol, int32_t max_it, double* out, int32_t* iters,
                    double* resid, int32_t device);
/* y = L x for 3 columns at once (x, y f64 [n,3]); the SpMV the solve is built on. */
int pyqsm_spmv3(const int32_t* indptr, const int32_t* indices, const double* vals,
                int64_t n, const double* x, double* y, int32_t device);
/* In-place clamp of every coordinate into [lo, hi] (skeletonize.py:291-296). */
int pyqsm_clamp(double* pts, int64_t n, const double lo[3], const double hi[3],
                int32_t device);

/* ---- fixed-radius queries -------------------------------------------------- */
/*
 * All points within `radius` of one centre (inclusive, d <= radius), ascending
 * indices: scipy KDTree(points).query_ball_point(center, r) as called at
 * pyQSM/utils/lib_integration.py:114-115 (find_neighbors_in_ball).
 *   out_idx i64 [n] capacity; *count = number written.
 */
int pyqsm_ball_query(const double* xyz, int64_t n, const double center[3], double radius,
                     int64_t* out_idx, int64_t* count, int32_t device);
/*
 * Union of the (at most k_cap nearest) neighbours within `radius` (strict,
 * d < radius) of every query point: the index set produced by
 * scipy KDTree(src).query(qry, k, distance_upper_bound=radius) at
 * pyQSM/geometry/reconstruction.py:238-244 and pyQSM/tree_isolation.py:126-131.
 *   mark u8 [n]: 1 for every source point some query selects;
 *   counts i32 [m]: neighbours each query selects (<= k_cap).
 */
int pyqsm_radius_mark(const double* src, int64_t n, const double* qry, int64_t m, double radius,
                      int32_t k_cap, uint8_t* mark, int32_t* counts, int32_t device);
/*
 * The padded tables themselves — what scipy KDTree(src).query(qry, k,
 * distance_upper_bound=radius) returns at pyQSM/geometry/reconstruction.py:238-240
 * and get_neighbors_kdtree(return_pcd=False) hands to pyQSM/canopy_metrics.py:238:
 * per query the (up to) k nearest source points with d < radius, ascending by
 * (distance, index); missing entries are padded with distance +inf and index n.
 *   idx i64 [m,k], dist f64 [m,k] (distances, not squared); 1 <= k <= 2048.
 */
int pyqsm_radius_knn(const double* src, int64_t n, const double* qry, int64_t m, double radius,
                     int32_t k, int64_t* idx, double* dist, int32_t device);
/*
 * pyqsm_radius_mark with a label per query point: label[j] = the smallest label among
 * the query points that select source point j (-1: none). One call replaces one cycle
 * of the region growing of pyQSM/tree_isolation.py:207-256 (extend_seed_clusters), where
 * clusters are visited in index order and the first one to reach a free point keeps it.
 *   qry_label i32 [m] (>= 0); label i32 [n]; counts i32 [m] as in pyqsm_radius_mark.
 */
int pyqsm_radius_label(const double* src, int64_t n, const double* qry, int64_t m,
                       const int32_t* qry_label, double radius, int32_t k_cap, int32_t* label,
                       int32_t* counts, int32_t device);

/* ---- farthest-point down-sampling ---------------------------------------- */
/*
 * Stands in for open3d PointCloud.farthest_point_down_sample(num_samples) as
 * called by extract_topology at pyQSM/geometry/skeletonize.py:127-132.
 *   out_idx i32 [num_samples]: indices in selection order, the first being
 *   start_index (Open3D starts at 0); each next one is the point farthest from
 *   everything selected so far (squared distance in fp64, lowest index on ties).
 */
int pyqsm_fps(const double* xyz, int64_t n, int64_t num_samples, int64_t start_index,
              int32_t* out_idx, int32_t device);

/* ---- point-cloud Laplacian ---------------------------------------------- */
/*
 * Stands in for robust_laplacian.point_cloud_laplacian(pts, mollify_factor,
 * n_neighbors) at pyQSM/geometry/skeletonize.py:253-255,341-343: kNN -> PCA
 * normal -> tangent-plane projection -> local Delaunay fan per point -> union
 * of fan triangles -> mollified cotangent Laplacian and lumped mass, both / 3.
 *   On success the indptr / indices / vals out-parameters hold a CSR matrix
 *   (symmetric, zero row sums) owned by the library: release each with
 *   pyqsm_free. mass f64 [n].
 */
int pyqsm_pc_laplacian(const double* xyz, int64_t n, int32_t k, double moll,
                       int64_t* nnz, int32_t** indptr, int32_t** indices, double** vals,
                       double* mass, int32_t device);
/*
 * The same for several clouds stacked into one array (the per-cluster calls of
 * pyQSM/qsm_generation.py:182-316 batched into one build): points [seg_start[s],
 * seg_start[s+1]) are cloud s, n_seg clouds, seg_start i64 [n_seg + 1] from 0 to n.
 * The mollification length (max(0, largest triangle slack + moll x mean edge length))
 * is taken per cloud, as n_seg separate calls would. The clouds must lie apart (further
 * than any k-neighbourhood reaches) for the result to be block diagonal; the caller
 * arranges that (extract_skeleton_batch moves every cloud to its own lattice cell).
 */
int pyqsm_pc_laplacian_seg(const double* xyz, int64_t n, const int64_t* seg_start, int64_t n_seg,
                           int32_t k, double moll, int64_t* nnz, int32_t** indptr,
                           int32_t** indices, double** vals, double* mass, int32_t device);

/* ---- the whole contraction loop, resident in HBM ------------------------------ */
/*
 * extract_skeleton of pyQSM/geometry/skeletonize.py:240-373 as ONE call: Laplacian ->
 * contraction solve -> clamp (:291-296) -> weights (:329-335) -> next Laplacian, up to
 * max_iter times, with the points, the matrix and the weights staying on the device
 * (the Python loop moves ~200 MB over PCIe per step at one million points). The loop's
 * bookkeeping is the reference's (W_H updated with the mass of the Laplacian just used,
 * volume ratio lagging one step, stop when a solve changes nothing or returns only NaN).
 *   xyz f64 [n,3]; seg_start i64 [n_seg+1] (NULL with n_seg = 1): several clouds stacked
 *   into one array are contracted together — one build and one block-diagonal solve per
 *   step — while weights, clamp box, volume ratio and termination stay per cloud;
 *   lo, hi f64 [n_seg,3]: the clamp box of every cloud (the reference takes the min / max
 *   bound of the oriented bounding box, :240-241); rtol, solver_max_it as pyqsm_lbc_solve.
 *   out_pts, total_shift f64 [n,3]; steps f64 [max(max_iter,1), n, 3] or NULL: the shift of
 *   every step (zero for clouds that had stopped); n_steps i32 [n_seg]: steps each cloud
 *   took; solve_iters i32 / solve_resid f64 / solve_ok u8 [max(max_iter,1)] (each may be
 *   NULL): per solve, as pyqsm_lbc_solve reports them; *n_solves: solves run.
 */
int pyqsm_extract_skeleton(const double* xyz, int64_t n, const int64_t* seg_start, int64_t n_seg,
                           int32_t k, double moll, int32_t max_iter, double termination_ratio,
                           double contraction_factor, double attraction_factor,
                           double max_contraction, double max_attraction, const double* lo,
                           const double* hi, double rtol, int32_t solver_max_it, double* out_pts,
                           double* total_shift, double* steps, int32_t* n_steps,
                           int32_t* solve_iters, double* solve_resid, uint8_t* solve_ok,
                           int32_t* n_solves, int32_t device);

/* ---- filters in front of the oriented bounding box's convex hull ----------------- */
/*
 * pyQSM/geometry/skeletonize.py:240-241 clamps the contracted points into
 * pcd.get_oriented_bounding_box().get_min_bound() / get_max_bound(); Open3D builds that box from
 * a PCA of the convex hull's vertices. Qhull over every point of a scan is 0.3 s per million
 * points; these two calls cut its input down to ~1 % without changing the hull:
 *   pyqsm_extreme_points: idx i64 [n_dirs] = the point with the largest x . dirs[d] (dirs f64
 *   [n_dirs,3]; lowest index on ties);
 *   pyqsm_outside_halfspaces: eq f64 [n_planes,4] = (a, o) of half-spaces a . x + o <= 0 (the
 *   facets of a small polytope whose corners are points of the cloud, scipy.spatial.ConvexHull
 *   .equations); idx i64 [capacity n] receives, ascending, every i with a . x_i + o >= -margin for
 *   some plane — the points that are not strictly inside — and *count their number. A point
 *   strictly inside such a polytope is strictly inside the cloud's hull, so the hull of the
 *   listed points has the same vertices. At most 256 planes.
 */
int pyqsm_extreme_points(const double* xyz, int64_t n, const double* dirs, int32_t n_dirs,
                         int64_t* idx, int32_t device);
int pyqsm_outside_halfspaces(const double* xyz, int64_t n, const double* eq, int32_t n_planes,
                             double margin, int64_t* idx, int64_t* count, int32_t device);

/*
 * Host-side helper of that loop, exported so that it can be checked without a GPU: the mean of
 * v[0..n) with NumPy's summation order (pairwise in blocks of 128 with eight accumulators, one
 * 8192-element buffer after the other, as np.add.reduce runs over a contiguous array with the
 * default buffer size), i.e. bit for bit what np.mean(M.diagonal())
 * of pyQSM/geometry/skeletonize.py:265,349 returns. The initial Laplacian weight is
 * 10^3 c sqrt(mean M): one ulp of difference there is amplified by the loop to millimetres
 * after twenty steps, so the native loop takes the mean the way the Python loop gets it.
 * n <= 0: *out = NaN.
 */
int pyqsm_mean_f64(const double* v, int64_t n, double* out);

#ifdef __cplusplus
}
#endif
#endif /* PYQSM_HIP_H */
