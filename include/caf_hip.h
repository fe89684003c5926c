/*
 * caf_hip.h -- C ABI of the MI355X (gfx950) Cross Ambiguity Function engine.
 *
 * This is the drop-in boundary for the caf_rust hot path.  The reference has no
 * FFI today; its extension point is "one more `impl CafSurface for X`"
 * (caf_rust/src/caf/mod.rs:23-66, pattern at :67-68,:118-119,:388-389).  An
 * eighth backend `CafHip` binds exactly the entry points below (the Rust stub
 * is in INTEGRATION.md).  Each entry point cites the reference interface it
 * replaces.
 *
 * Conventions
 *   - complex slices are interleaved {re, im}: `const double*` of 2*n doubles is
 *     a `&[Complex64]` (num_complex::Complex64 is #[repr(C)] {re:f64, im:f64});
 *     the `_c64` twins take interleaved floats (numpy complex64, the on-disk
 *     ".c64" format of caf_rust/src/utils.rs:10-35).
 *   - every function returns an int status (CAF_OK == 0); nothing unwinds across
 *     the boundary.  The reference panics instead (assert!/unwrap,
 *     xcor_rustfft.rs:54-55); a host shim turns non-zero into a panic.
 *   - outputs are caller-allocated; inputs are borrowed for the call.
 *   - a context is bound to one GPU and is NOT thread-safe; use one per thread.
 *   - lengths: n must be a power of two >= 1 (xcor_rustfft.rs:2 "Assumes
 *     equal-length, power of 2"); caf_surface pads to 2n itself (mod.rs:130-131).
 *     The reference's FFTplanner would take any length: INTEGRATION.md "Where the
 *     ABI refuses what the reference accepts".
 *   - fs == 0 is accepted and gives what the reference gives (mod.rs:54-56: dt = inf,
 *     NaN phasors): an all-NaN surface, rows (idx 0, val 0.0), peak (0.0, 0), row -1;
 *     apply_freq_shift leaves sample 0 and turns every later sample into NaN.
 *   - there is NO CPU fallback: without a usable HIP device every call fails
 *     with CAF_ERR_NO_DEVICE / CAF_ERR_HIP.
 */
#ifndef CAF_HIP_H
#define CAF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAF_ABI_VERSION 5

enum caf_status {
    CAF_OK = 0,
    CAF_ERR_BAD_ARG = 1,   /* null pointer / bad enum */
    CAF_ERR_LENGTH = 2,    /* n == 0 or not a power of two (xcor_rustfft.rs:54-55 assert) */
    CAF_ERR_HIP = 3,       /* a HIP runtime call failed; see caf_last_error_string() */
    CAF_ERR_NOMEM = 4,
    CAF_ERR_NO_DEVICE = 5, /* no HIP device visible / device id out of range / the device is not gfx950 (MI355X) */
    CAF_ERR_STATE = 6,     /* call order violated (e.g. stream used after destroy) */
    CAF_ERR_RCCL = 7,      /* librccl could not be loaded or an RCCL call failed (CAF_MULTI_REDUCE_RCCL only) */
    CAF_ERR_TIMEOUT = 8    /* a multi-device call ran into the deadline of caf_multi_surface_set_timeout (ABI 5) */
};

enum caf_dtype {
    CAF_C128 = 0, /* complex128 in, f64 |.|^2 out  (the reference's arithmetic) */
    CAF_C64 = 1   /* complex64 in,  f32 |.|^2 out  (phase still computed in f64) */
};

typedef struct caf_ctx caf_ctx;
typedef struct caf_plan caf_plan;

/* Global peak record, mod.rs:31-42 (find_peak).  row == -1 means "no row had a
 * peak > 0": freq = 0.0, idx = 0, val = 0.0 exactly like the reference's
 * initial `max` (mod.rs:32-35). */
typedef struct caf_peak {
    double val;   /* xcor_peak_val of the winning row (f32 paths: widened) */
    double freq;  /* CafSurfaceRow.freq of the winning row */
    uint64_t idx; /* xcor_peak_idx: sample index in [0, 2n) */
    int64_t row;  /* position in the freq list (global position for shards) */
} caf_peak;

int caf_abi_version(void);
/* Thread-local description of the last failure on this thread ("" if none). */
const char *caf_last_error_string(void);
/* Number of HIP devices visible; 0 without a GPU (never touches the GPU state
 * beyond counting). */
int caf_device_count(void);

/* ---- context ----------------------------------------------------------- */
int caf_ctx_create(int device_id, caf_ctx **out);
/* Destroys the context AND every plan created from it that is still alive (their handles
 * become invalid); CAF_ERR_STATE if one of those plans still has a live caf_stream. */
int caf_ctx_destroy(caf_ctx *ctx);
/* Run all work of this context on a caller-owned hipStream_t, e.g. torch's current
 * stream.  The handle is used as given: NULL is HIP's null (legacy default) stream, which
 * is what torch.cuda.current_stream().cuda_stream returns for torch's default stream.
 * caf_ctx_reset_stream goes back to the context's private non-blocking stream. */
int caf_ctx_set_stream(caf_ctx *ctx, void *hip_stream);
int caf_ctx_reset_stream(caf_ctx *ctx);
int caf_ctx_synchronize(caf_ctx *ctx);
/* Device facts for reports: CU count and name (buf may be NULL). */
int caf_ctx_device_info(caf_ctx *ctx, int *cu_count, char *name_buf, size_t name_cap);

/* ---- a1: CafSurface::apply_freq_shift, mod.rs:46-65 -------------------------
 * out[i] = in[i] * e^{j*ph*i}, ph = ((2*PI)*freq_shift)*(1/fs).  The reference
 * builds the phasor by a sequential recurrence; here every phasor is evaluated
 * directly from the f64 phase (difference <= ~2e-14 absolute over 8192 samples). */
int caf_apply_freq_shift_c128(caf_ctx *ctx, const double *in, size_t n,
                              double freq_shift, uint32_t fs, double *out);
int caf_apply_freq_shift_c64(caf_ctx *ctx, const float *in, size_t n,
                             double freq_shift, uint32_t fs, float *out);

/* ---- a2-a4: xcor_rustfft::Xcor::{new,run}, xcor_rustfft.rs:29-78 ------------
 * out[k] = sum_m a[(m+k) mod n] * conj(b[m]) = IFFT(FFT(a)*conj(FFT(b))/n)
 * (unnormalised inverse).  Tables are cached inside the context per n, which is
 * what Xcor::new/clone provide.  One kernel launch for n = 2 ... 16384 (complex128:
 * 8192) reading / writing pinned staging copies of a, b, out; radix-16 Stockham passes over HBM beyond. */
int caf_xcor_c128(caf_ctx *ctx, const double *a, const double *b, size_t n, double *out);
int caf_xcor_c64(caf_ctx *ctx, const float *a, const float *b, size_t n, float *out);

/* ---- a5-a7: CafSurface::caf_surface + find_peak, mod.rs:26-42,121-166 -------
 * needle, haystack: n complex each (equal length, mod.rs / xcor assert).
 * surface : nfreq * 2n values row-major (row r = freqs_hz[r]) or NULL to skip
 *           the device->host copy of the 26 MB surface.
 * row_idx / row_val : nfreq entries (CafSurfaceRow.xcor_peak_idx/_val), may be NULL.
 * peak    : find_peak() of the rows in list order (first strictly-greater wins).
 * Host pointers; blocks until the results are in host memory. */
int caf_surface_c128(caf_ctx *ctx, const double *needle, const double *haystack,
                     size_t n, const double *freqs_hz, size_t nfreq, uint32_t fs,
                     double *surface, uint64_t *row_idx, double *row_val,
                     caf_peak *peak);
int caf_surface_c64(caf_ctx *ctx, const float *needle, const float *haystack,
                    size_t n, const double *freqs_hz, size_t nfreq, uint32_t fs,
                    float *surface, uint64_t *row_idx, float *row_val,
                    caf_peak *peak);

/* How the host-pointer calls run: the context keeps the plans (tables, staging, device buffers) of the four most
 * recently used (n, freq list, fs, dtype) combinations -- what Xcor::new / clone buy in the reference
 * (xcor_rustfft.rs:29-46,82-93) -- so alternating between a few shapes costs no set-up.  Inputs are copied into
 * pinned staging by the CPU and read from there by the kernels; row peaks and the caf_peak record are written
 * to pinned memory by the kernels.  n = 4096 is one kernel launch per call whose completion is a polled pinned
 * word (peaks only: ~45 us per call on MI355X).  A requested surface (26 MB at 400 x 8192 complex128) is
 * stored by the row kernel straight into the caller's buffer while the other rows compute IF that buffer is
 * memory of caf_host_alloc / caf_host_register below; any other (pageable) buffer costs one device-to-host
 * copy after the kernel. */

/* Host memory the kernels of this context may address directly.  caf_host_alloc returns pinned memory
 * (free it with caf_host_free, or let caf_ctx_destroy do it); caf_host_register pins a range the caller owns
 * (page-locking 26 MB costs ~1.2 ms the first time: do it once for a long-lived buffer, never per call) until
 * caf_host_unregister / caf_ctx_destroy.  A Rust host keeps one such arena inside its CafHip backend and builds
 * the Vec<CafSurfaceRow> rows (mod.rs:156-161) from it. */
/* caf_host_register takes WHOLE PAGES only (CAF_ERR_BAD_ARG unless ptr and bytes are multiples of sysconf(_SC_PAGESIZE)):
 * page-locking works on pages, so a range that starts or ends inside a page would pin, and map for the GPU, whatever else the
 * allocator keeps in that page -- and leave it so after the block has gone back to the heap.  Register memory whose lifetime
 * the caller controls (mmap, aligned_alloc of a page multiple, an arena kept for the life of the context) and unregister it
 * BEFORE it is unmapped or freed.  Rows are callee-owned in the reference (mod.rs:156-161), which is why the arena exists. */
int caf_host_alloc(caf_ctx *ctx, size_t bytes, void **out);
int caf_host_free(caf_ctx *ctx, void *ptr);
int caf_host_register(caf_ctx *ctx, void *ptr, size_t bytes);
int caf_host_unregister(caf_ctx *ctx, void *ptr);

/* find_peak over caller-held rows (mod.rs:31-42), evaluated on the device with
 * the same kernel the fused path uses. */
int caf_find_peak(caf_ctx *ctx, const double *freqs_hz, const uint64_t *row_idx,
                  const double *row_val, size_t nfreq, caf_peak *peak);

/* ---- device-resident path (bench, streaming, multi-GPU shards) --------------
 * A plan fixes (n, freq list, fs, dtype) like Xcor::new fixes n, and owns the
 * per-row phasor tables.  [row_begin,row_end) selects this GPU's contiguous
 * shard of the freq list; peak.row is reported in GLOBAL list
 * positions so a min over equal peaks preserves "first row wins". */
int caf_plan_create(caf_ctx *ctx, size_t n, const double *freqs_hz, size_t nfreq,
                    uint32_t fs, int dtype, size_t row_begin, size_t row_end,
                    caf_plan **out);
int caf_plan_destroy(caf_plan *plan);
/* Name of the kernel path the plan selected: "fused4096" (n = 4096), "chain" (every other n
 * from 1024 to 131072 in complex64 / 65536 in complex128: LDS-resident single-pass rows), "small"
 * (n = 1 ... 512: lane-group rows, one launch) or "generic" (anything larger: radix-16 Stockham passes over HBM).  The measurement build can
 * also report "tiled65536" (round 1's three-pass n = 32768 form). */
const char *caf_plan_path(const caf_plan *plan);
size_t caf_plan_rows(const caf_plan *plan);
/* Name of the dominant (row) kernel the plan launches, as rocprofv3 prints it without
 * arguments, e.g. "caf::k_seq_rows<double, 15, caf::SeqIo<double> >" -- for matching bench.py's roofline
 * object against profiles/. */
const char *caf_plan_kernel_name(const caf_plan *plan);

/* Enqueue `batch` surfaces on the context's stream and return immediately.
 * ALL pointers are device pointers (hipMalloc / torch tensors):
 *   d_needle, d_haystack : [batch][n] complex (dtype of the plan)
 *   d_surface            : [batch][rows][2n] real or NULL (skip the store)
 *   d_row_idx            : [batch][rows] uint64
 *   d_row_val            : [batch][rows] real
 *   d_peak               : [batch] caf_peak
 * rows = row_end - row_begin of the plan. */
int caf_surface_dev(caf_plan *plan, const void *d_needle, const void *d_haystack,
                    size_t batch, void *d_surface, uint64_t *d_row_idx,
                    void *d_row_val, caf_peak *d_peak);

/* HIP-event timing of the dominant kernel for bench.py's roofline object:
 * start/stop events are recorded on the plan's stream around the row-kernel
 * launches of every caf_surface_dev call between begin and end; end
 * synchronises and returns the accumulated kernel milliseconds and the number
 * of launches. */
int caf_plan_timing_begin(caf_plan *plan);
int caf_plan_timing_end(caf_plan *plan, double *kernel_ms_total, uint64_t *launches);

/* ---- views of a surface in the other cook-off implementations' conventions ----------
 * (cheap epilogues over a |.|^2 surface already in memory)
 *   CAF_VIEW_GO     : caf_go/caf.go:95-116 + main.go:35 -- 2n lags, magnitude (not squared),
 *                     needle padded at the end / haystack at the front:
 *                     out[r][k] = sqrt(surface[r][(n - k) mod 2n]),  lag = n - k
 *   CAF_VIEW_PYTHON : caf_python/caf.py:15-18,145 -- scipy correlate(mode='same'): n lags,
 *                     magnitude, reversed lag axis: out[r][i] = sqrt(surface[r][(n/2 - i) mod 2n]),
 *                     tau = n/2 - i
 * Host pointers; `surface` is rows x 2n of the dtype's real type, `out` rows x (2n | n). */
enum caf_view { CAF_VIEW_GO = 1, CAF_VIEW_PYTHON = 2 };
int caf_surface_view(caf_ctx *ctx, int dtype, const void *surface, size_t rows, size_t n, int view, void *out);

/* ---- streaming (BASELINE configs[4]) ----------------------------------------------
 * Back-to-back surfaces from host memory: `nslots` (>= 2) independent slots, each with
 * pinned host staging for `batch` (needle, haystack) pairs, its own device buffers, its
 * own HIP stream (from a per-context pool) and ONE captured hipGraph of kernel nodes
 * {haystack spectrum, row kernel, find_peak}: the kernels read the pinned inputs and write the
 * pinned results through their device mappings (the haystack is read in place, the needle is
 * staged into device memory by the spectrum launch, find_peak writes the row peaks and the
 * caf_peak records out) -- no copy-engine nodes.  While slot k computes, the caller fills slot
 * k+1's pinned buffers and submits it: its input transfer overlaps slot k's kernels.  The fastest form measured on MI355X, and
 * the one that keeps its rate from creation to creation, is batch = 20 ... 32 on two slots (n = 4096, 400 rows: 16 to 25 rounds of the
 * persistent row workgroups per replay instead of 6.25 with eight surfaces; the row launch of a replay of 16 surfaces or more leaves 32 workgroup slots free so
 * that the other slot's staging + spectrum launch runs beside it: profiles/r05_stream/form_stability.txt);
 * single-surface replays depend on how the runtime arbitrates the slot streams' hardware queues (the slot streams are probed at
 * creation so that they sit on separate ones; profiles/r03_stream/form_stability.txt).  Surfaces stay on the device (d_surface of the slot, NULL if want_surface == 0);
 * only (tau, f) + per-row peaks come back, as SURVEY.md section 8d prescribes.
 * Plans whose row kernel keeps its intermediate data on-chip ("fused4096", "chain", "small") give
 * every slot private device state, so slots execute concurrently.  Plans on the
 * "generic" path share the plan's pass workspaces: their slots run on ONE stream, in
 * submit order (slot k+1's H2D waits for slot k's kernels).
 * Lifetime: the captured graphs hold pointers into the plan's tables and workspaces;
 * caf_plan_destroy and caf_ctx_destroy return CAF_ERR_STATE while a caf_stream of the plan
 * is alive -- destroy streams first, then plans, then the context (caf_ctx_destroy itself
 * destroys every plan that is still alive). */
typedef struct caf_stream caf_stream;
int caf_stream_create(caf_plan *plan, size_t batch, int nslots, int want_surface, caf_stream **out);
/* Same with flags.  CAF_STREAM_SPLIT: the slot's ONE graph holds `batch` (<= 16) independent
 * SINGLE-SURFACE node chains (each its own stage-in, haystack spectrum, row kernel, find_peak,
 * stage-out) instead of one batched chain -- as parallel branches when the plan's per-slot state
 * is private ("fused4096", "chain"), in series otherwise.  One replay then retires `batch`
 * surfaces at single-surface granularity; host buffers and results are laid out as in the
 * batched mode.
 * Single-surface chains (batch == 1, or CAF_STREAM_SPLIT) of "fused4096" plans run as roles of ONE
 * launch (csrc/kernels_surf4096.hpp: needle staging, haystack spectrum, Doppler rows, find_peak) when
 * at most two surfaces are in flight (nslots * chains per slot <= 2), and as TWO kernel nodes
 * {staging + spectrum | rows + find_peak} otherwise; CAF_STREAM_ONE_KERNEL / CAF_STREAM_TWO_KERNELS
 * force either, CAF_STREAM_THREE_KERNELS keeps the older {spectrum, rows, find_peak} chain (for
 * comparison).  In the first two forms completion is read from a pinned sequence word. */
enum caf_stream_flags {
    CAF_STREAM_SPLIT = 1,
    CAF_STREAM_THREE_KERNELS = 2,
    CAF_STREAM_TWO_KERNELS = 4,
    CAF_STREAM_ONE_KERNEL = 8,
    /* BASELINE configs[4] to the letter: the slot's inputs cross PCIe as hipMemcpyAsync (copy-engine) nodes of the graph into
     * device buffers and the peaks / row records come back as hipMemcpyAsync nodes, instead of kernels reading and writing the
     * mapped pinned buffers in place (the default, faster: bench.py reports both).  Accepted with every batch size and with
     * CAF_STREAM_SPLIT; it always takes the {copy, spectrum, rows, find_peak, copy} chain -- never the single-launch surface,
     * whose kernel reads and writes the pinned buffers itself -- so combining it with CAF_STREAM_ONE_KERNEL or
     * CAF_STREAM_TWO_KERNELS is CAF_ERR_BAD_ARG. */
    CAF_STREAM_MEMCPY_NODES = 16
};
int caf_stream_create_ex(caf_plan *plan, size_t batch, int nslots, int want_surface, unsigned flags,
                         caf_stream **out);
int caf_stream_destroy(caf_stream *st);
/* Pinned host buffers of a slot: [batch][n] complex each (dtype of the plan). */
int caf_stream_host_buffers(caf_stream *st, int slot, void **needle, void **haystack);
/* The whole loop in one call: `count` host-resident pairs (needles / haystacks: [count][n] complex of
 * the plan's dtype, ordinary host memory) go through the slots in order, `batch` per replay (a ragged
 * last replay is padded with zeros); peaks[count] and, if non-NULL, row_idx / row_val [count][rows]
 * come back in input order.  Surfaces stay in the slots (caf_stream_surface).  This is BASELINE
 * configs[4] as a compiled host would run it: one native loop over the pairs (the reference's
 * benches call caf_surface once per iteration from Rust, caf_bench.rs:150-168). */
int caf_stream_run(caf_stream *st, const void *needles, const void *haystacks, size_t count, caf_peak *peaks,
                   uint64_t *row_idx, void *row_val);
/* Where the host thread spent the last caf_stream_run: seconds4 = {filling the pinned slots (memcpy of the pairs),
 * launching the graphs, waiting for slots (polled sequence word / stream), collecting results}; the rest of the
 * call's wall time is loop overhead.  The GPU works under all four. */
int caf_stream_run_stats(caf_stream *st, double *seconds4);
/* Replay the slot's graph on the slot's stream (asynchronous). */
int caf_stream_submit(caf_stream *st, int slot);
/* Block until the slot's last submit finished; copies out `batch` caf_peak records and,
 * if non-NULL, batch*rows row indices / values (value type of the plan's dtype).  When this returns, the
 * slot's surface slab (caf_stream_surface) is complete in device memory too: in the polled forms every wave of a
 * row has waited for its surface stores before the row is counted towards the sequence word. */
int caf_stream_wait(caf_stream *st, int slot, caf_peak *peaks, uint64_t *row_idx, void *row_val);
/* Device address of the slot's surface slab ([batch][rows][2n]) or NULL. */
void *caf_stream_surface(caf_stream *st, int slot);

/* ---- surface-parallel multi-GPU streaming (second decomposition) ---------------
 * Whole surfaces round-robin over devices: the unit of work the reference's callers hand out when many
 * surfaces are wanted (one caf_surface call per iteration, benches/caf_bench.rs:150-168; independent pool tasks,
 * mod.rs:404-457).  One caf_ctx + caf_plan + caf_stream (eight surfaces per replay -- the form whose rate does not depend on
 * how the runtime maps slot streams to hardware queues -- `nslots` slots) per entry of
 * device_ids -- an id may repeat: two contexts on one GPU -- each driven by its own host thread during a run.
 * Pair k of the caller's arrays goes to worker k % ndev; peaks / row_idx / row_val come back in INPUT order.
 * No collective: a surface's (tau, f) is complete on the device that computed it.  If workers fail, the call
 * returns the status of the first failing one (by position) with its message.
 * caf_multi_stream_share is the assignment rule by itself (no GPU needed): worker w of nworkers handles the
 * `items` pairs first, first + stride, ... of `count`; ranks of a one-process-per-GPU job use it to pick their
 * share of a common input set. */
typedef struct caf_multi_stream caf_multi_stream;
int caf_multi_stream_share(size_t count, int nworkers, int worker, size_t *first, size_t *stride, size_t *items);
int caf_multi_stream_create(const int *device_ids, int ndev, size_t n, const double *freqs_hz, size_t nfreq,
                            uint32_t fs, int dtype, int nslots, int want_surface, caf_multi_stream **out);
int caf_multi_stream_devices(const caf_multi_stream *ms);
int caf_multi_stream_run(caf_multi_stream *ms, const void *needles, const void *haystacks, size_t count,
                         caf_peak *peaks, uint64_t *row_idx, void *row_val);
/* A deadline for every later caf_multi_stream_run (ABI 5), with the meaning of caf_multi_surface_set_timeout below: each
 * worker's waits for its slots are polls against `seconds` from the start of the run; on expiry the run returns
 * CAF_ERR_TIMEOUT naming worker and device, later runs return CAF_ERR_STATE, and caf_multi_stream_destroy leaves behind
 * (does not wait for) the workers of a device that has not drained 2 s later, reporting CAF_ERR_TIMEOUT.  0 = none. */
int caf_multi_stream_set_timeout(caf_multi_stream *ms, double seconds);
/* With want_surface != 0 every worker's slots keep their surfaces on the worker's device, as caf_stream_create does
 * (the reference's row record carries the magnitudes, mod.rs:17-22,156-161): caf_multi_stream_surface is the device
 * address of a worker's slot slab ([8][rows][2n] of the dtype's real type; NULL without surfaces), and
 * caf_multi_stream_locate says where pair `pair` of the last run over `count` pairs lives: worker, slot, position in the
 * slab, and whether it is still resident (a later replay of the same run reuses the slot). */
void *caf_multi_stream_surface(caf_multi_stream *ms, int worker, int slot);
int caf_multi_stream_locate(const caf_multi_stream *ms, size_t count, size_t pair, int *worker, int *slot, size_t *index,
                            int *resident);
int caf_multi_stream_destroy(caf_multi_stream *ms);

/* ---- Doppler-row shards of ONE surface over several devices (first decomposition) --------------
 * The reference's fan-out and join live INSIDE the operator: one caf_surface call spreads the rows over pool workers and
 * returns the joined rows (CafRustFFTThreadpool::caf_surface, mod.rs:391-461; one task per row :404-457), and find_peak
 * scans the joined rows (:31-42).  Here the workers are GPUs.  Worker r of ndev owns the contiguous rows
 * [r*nfreq/ndev, (r+1)*nfreq/ndev) of the freq list (caf_multi_surface_shard: the rule by itself, no GPU needed) with its
 * own context, row-shard plan and staging, driven by its own host thread for the length of a run (worker 0 by the caller's
 * thread).  An id may repeat (several contexts on one GPU).  Inputs are replicated, the haystack spectrum is computed on
 * every device, each device writes its rows of `surface` -- in place if that memory came from caf_multi_surface_host_alloc /
 * _host_register, else through a device slab and one copy per device -- and its rows of row_idx / row_val; there is no
 * collective on the data path.  `peak` is find_peak over all rows: the largest value, among equal values the lowest global
 * row (== the reference's first-strictly-greater scan), joined
 *   flags == 0                 on the host from the ndev shard records (caf_multi_surface_reduce: the rule by itself);
 *   CAF_MULTI_REDUCE_RCCL      through RCCL inside this process: ncclAllReduce(max) over the shard values, then
 *                              ncclAllReduce(min) over (global_row << 32 | idx) keys of the shards holding the maximum
 *                              (RCCL has no MAXLOC), on the workers' streams over xGMI.  Needs distinct device ids
 *                              (one RCCL rank per GPU).  librccl is dlopen()ed at the first such create
 *                              (caf_rccl_library names the file; default "librccl.so.1"): the library does not link it.
 * surface: nfreq x 2n of the dtype's real type or NULL; row_idx / row_val: nfreq entries or NULL; needle / haystack: n
 * complex of the dtype (host pointers).  Blocks until every result is in host memory.  If workers fail, the call returns
 * the status of the first failing one (by position) with its message.  Like a context, the object is NOT thread-safe: one
 * caf_multi_surface_run at a time (the object's own worker threads are an implementation detail of that one call). */
typedef struct caf_multi_surface caf_multi_surface;
enum caf_multi_flags {
    CAF_MULTI_REDUCE_RCCL = 1,
    /* every worker keeps its rows of the surface in its own HBM ("the surface stays sharded, no gather on
     * the timed path"): caf_multi_surface_run takes surface = NULL and caf_multi_surface_slab(h, worker) is the device
     * address of rows [row_begin, row_end) x 2n on that worker's GPU */
    CAF_MULTI_SURFACE_ON_DEVICE = 2
};
int caf_multi_surface_shard(size_t nfreq, int nworkers, int worker, size_t *row_begin, size_t *row_end);
int caf_multi_surface_reduce(const caf_peak *shard_peaks, int nshards, caf_peak *out);
int caf_rccl_library(const char *path);
int caf_multi_surface_create(const int *device_ids, int ndev, size_t n, const double *freqs_hz, size_t nfreq, uint32_t fs,
                             int dtype, unsigned flags, caf_multi_surface **out);
int caf_multi_surface_devices(const caf_multi_surface *h);
void *caf_multi_surface_slab(caf_multi_surface *h, int worker);
/* device id, row shard and row-kernel name of one worker (any pointer may be NULL) */
int caf_multi_surface_worker_info(const caf_multi_surface *h, int worker, int *device, size_t *row_begin, size_t *row_end,
                                  const char **kernel_name);
int caf_multi_surface_run(caf_multi_surface *h, const void *needle, const void *haystack, void *surface,
                          uint64_t *row_idx, void *row_val, caf_peak *peak);
/* A deadline for every later caf_multi_surface_run / _run_batch of the object (ABI 5).  The reference's join cannot wait for
 * ever: `rx.recv().unwrap()` per row (mod.rs:452-457) panics when a pool worker died.  A GPU worker can stay silent instead
 * (a device that never finishes its launch, a peer that never enters the RCCL exchange), so with seconds > 0 every wait
 * inside a call -- each worker for its device, the caller for the worker threads, the RCCL join for every device -- is a poll
 * against `seconds` from the start of the call.  On expiry the call returns CAF_ERR_TIMEOUT, caf_last_error_string() names the
 * worker and its device, the object is unusable (every later run: CAF_ERR_STATE) and the work stays wherever it is: the
 * library never resets a device or restarts anything.  caf_multi_surface_destroy then gives the devices 2 s to drain and
 * LEAVES BEHIND (does not free, does not wait for) the context and buffers of a worker that has not; it reports that with
 * CAF_ERR_TIMEOUT (the handle is invalid either way).  seconds == 0 (the default) means no deadline: plain blocking waits.
 * What the deadline cannot bound: a runtime call that blocks while ENQUEUEING work (the caller then gets CAF_ERR_TIMEOUT
 * 2 s after the deadline and the stuck thread is abandoned: it may still read the input arrays of that call when the
 * runtime lets it go, so keep them alive until caf_multi_surface_destroy has returned CAF_OK -- for the rest of the process
 * if it reports threads left behind).  After an ordinary timeout (a device that did not finish) nothing of the library
 * touches the caller's memory again except the device work already queued, which writes only library-owned buffers and
 * memory obtained from caf_multi_surface_host_alloc / _host_register (kept registered while that device is busy). */
int caf_multi_surface_set_timeout(caf_multi_surface *h, double seconds);
/* B surfaces per call -- the loop of benches/caf_bench.rs:150-168 (one caf_surface call per iteration, each fanning its rows out
 * over the pool and joining them, mod.rs:391-461) handed to the operator as ONE call, so that every device runs ONE launch of
 * its row kernel over its rows [row_begin, row_end) of ALL B surfaces instead of B launches with the chip idle in between.
 *   needles, haystacks : [batch][n] complex of the dtype, host pointers; every worker takes its own copy of all B pairs (the
 *                        inputs are replicated, the haystack spectra are computed on every device).  Both NULL: re-run the
 *                        pairs of the previous call, which are still in every worker's HBM (batch must equal that call's) --
 *                        the form bench.py times ("inputs already resident in HBM").
 *   row_idx / row_val  : [batch][nfreq] or NULL (NULL: the row records stay on the devices, caf_multi_surface_batch_results)
 *   peaks              : [batch] find_peak records (mod.rs:31-42; largest value, among equal values the lowest global row),
 *                        joined on the host (flags == 0: caf_multi_surface_reduce per surface) or, with CAF_MULTI_REDUCE_RCCL,
 *                        by ONE grouped ncclAllReduce(max) over the B shard values and ONE ncclAllReduce(min) over the B
 *                        (global_row << 32 | idx) keys per call, on the workers' streams over xGMI.
 * With CAF_MULTI_SURFACE_ON_DEVICE every worker keeps its slab [batch][row_end - row_begin][2n] of the B surfaces in its own
 * HBM until the next batch call (the surface stays sharded); without that flag the batch call computes
 * row records and peaks only (for host-resident surfaces call caf_multi_surface_run per pair).  No collective on the data path.
 * caf_multi_surface_timing_* and caf_multi_surface_run_stats cover batch calls too.  Blocks until the results are in host memory. */
int caf_multi_surface_run_batch(caf_multi_surface *h, const void *needles, const void *haystacks, size_t batch,
                                uint64_t *row_idx, void *row_val, caf_peak *peaks);
/* one worker's share of the last batch call: the number of pairs resident, DEVICE addresses (on that worker's GPU) of its slab
 * [batch][rows][2n] (NULL without CAF_MULTI_SURFACE_ON_DEVICE), row records [batch][rows] and shard find_peak records [batch]
 * (global row positions), and the pinned host copy of the latter; any pointer may be NULL */
int caf_multi_surface_batch_results(caf_multi_surface *h, int worker, size_t *batch, void **d_slab, uint64_t **d_row_idx,
                                    void **d_row_val, caf_peak **d_peaks, const caf_peak **h_peaks);
/* last run: seconds2 = {fan-out + shards + join of the worker threads, peak reduction}; shard_peaks[ndev] = every
 * worker's own find_peak record (global row positions); either may be NULL.  (A batch re-run of resident pairs with
 * CAF_MULTI_REDUCE_RCCL and no row records wanted queues the RCCL join behind the row launches and waits ONCE: the
 * devices' compute time then shows up in the second figure.) */
int caf_multi_surface_run_stats(caf_multi_surface *h, double *seconds2, caf_peak *shard_peaks);
/* HIP-event time of every worker's row kernel between begin and end: kernel_ms_total[ndev], launches[ndev] (n = 4096
 * plans run a surface as ONE launch and report no separate row-kernel time: launches 0) */
int caf_multi_surface_timing_begin(caf_multi_surface *h);
int caf_multi_surface_timing_end(caf_multi_surface *h, double *kernel_ms_total, uint64_t *launches);
/* host memory EVERY worker may write in place (pinned, portable): the multi-device counterpart of caf_host_alloc /
 * caf_host_register (whole pages only, like it); released by the matching call or by caf_multi_surface_destroy */
int caf_multi_surface_host_alloc(caf_multi_surface *h, size_t bytes, void **out);
int caf_multi_surface_host_free(caf_multi_surface *h, void *ptr);
int caf_multi_surface_host_register(caf_multi_surface *h, void *ptr, size_t bytes);
int caf_multi_surface_host_unregister(caf_multi_surface *h, void *ptr);
int caf_multi_surface_destroy(caf_multi_surface *h);

/* ---- the peak exchange for hosts that run the collectives themselves (one process per GPU) -------------
 * find_peak over contiguous row shards (mod.rs:31-42 over the joined rows: largest value, among equal values the lowest
 * global row) = all-reduce(MAX) over the shards' peak values + all-reduce(MIN) over (global_row << 32 | idx) keys of the
 * shards that hold the maximum (RCCL has no MAXLOC).  caf_multi_surface_* runs both inside the library; a host with one
 * process per GPU (torch.distributed, MPI) runs the two collectives itself and takes the three element-sized kernels around
 * them from here.  d_peaks: the `count` shard records caf_surface_dev left on this context's device (global row positions);
 * d_red: 4 * count 8-byte words of device memory, laid out [val | gmax | key | gkey]; asynchronous, on the context's stream.
 *   stage 0   gmax[b] = this shard's peak value (0.0 without a peak)            then all-reduce gmax[0..count) with MAX, IN PLACE (f64)
 *   stage 1   gkey[b] = (row << 32 | idx) if this shard holds gmax[b] > 0, else INT64_MAX   then all-reduce gkey with MIN, IN PLACE (int64)
 *   stage 2   d_out[b] = {gmax, freqs_all[row], idx, row}; {0.0, 0.0, 0, -1} if no shard had a peak (the reference's initial maximum)
 * Keys are SIGNED 64-bit (a row position is < 2^31): torch and MPI reduce int64. */
int caf_peak_exchange_stage(caf_ctx *ctx, int stage, const caf_peak *d_peaks, size_t count, void *d_red,
                            const double *d_freqs_all, size_t nfreq_all, caf_peak *d_out);

/* ---- debug: red zones ---------------------------------------------------------------------------------------------
 * GPU AddressSanitizer is not available for this target, so the library can police its own allocations: after
 * caf_debug_guard_bands(bytes) every device / pinned allocation the library makes (tables, spectra, slabs, staging, stream
 * slots, caf_host_alloc memory, ...) is laid out [guard | allocation | guard] with the guards (rounded up to whole 4 KiB
 * pages) filled with 0xA5; caf_debug_check_guards synchronises every device that holds such an allocation and verifies all
 * guards of all live allocations: CAF_OK, or CAF_ERR_STATE with the first damaged allocation (size, allocation site, byte
 * offset of the stray store) in caf_last_error_string().  Process-wide; bytes == 0 switches it off again for later
 * allocations.  A debugging aid: allocations become slower, kernels do not. */
int caf_debug_guard_bands(size_t bytes);
int caf_debug_check_guards(size_t *allocations_checked, size_t *violations);

#ifdef __cplusplus
}
#endif
#endif /* CAF_HIP_H */
