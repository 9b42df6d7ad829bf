/*
 * mifft.h -- C ABI of libmifft, the MI355X (gfx950) batched N-D radix-N FFT.
 *
 * This is the drop-in boundary for the ONE hot path of martinvuyk/hackathon-fft:
 * the GPU overloads of `plan_fft` / `fft` and everything below them.  Every
 * entry point cites the reference interface it replaces (paths relative to the
 * reference repository root).  Plain pointers and sizes only; no torch / HIP
 * types in the signatures (the stream is passed as an opaque `void*` that must
 * be a `hipStream_t`, or NULL for the default stream).
 *
 * The library has NO CPU execution path: every exec entry point runs
 * hand-written HIP kernels on the plan's device and fails with
 * MIFFT_ERR_NO_DEVICE when no gfx950 device is usable.  The CPU restatement of
 * the reference lives in oracle/ and is test infrastructure only.
 *
 * Data layout (reference: fft/fft/fft.mojo:20-46): row-major
 *     x   : (batch, d0[, d1[, ...]], C_in)   C_in  in {1 (real), 2 (re,im)}     (1..MIFFT_MAX_DIMS dims)
 *     out : (batch, d0[, d1[, ...]], 2)      interleaved (re, im)
 * Forward transform is unnormalised with exp(-2*pi*i*nk/N); the inverse uses
 * the conjugate twiddles and scales each transformed dimension by 1/N_dim
 * (fft/fft/_fft.mojo:292-294), i.e. matches numpy.fft.fftn / ifftn over axes
 * 1..ndim.  A real input produces the FULL N-point spectrum
 * (fft/fft/_fft.mojo:254-257), not numpy's half spectrum.  (For real input with 2..4
 * transformed dimensions the last pass may compute only half of the spectrum and store the
 * other half as its conjugate mirror image -- the same values to rounding, exactly Hermitian;
 * never with MIFFT_FLAG_FAITHFUL_STAGES.)
 *
 * Environment (read ONCE per process, at the first plan; hackathon_fft_amd/csrc/mifft_config.h):
 *     MIFFT_JIT=0           no runtime specialisation (hipRTC): lengths without a precompiled kernel run
 *                           on the literal-stage kernels instead
 *     MIFFT_JIT_CACHE_DIR   directory of the on-disk cache of runtime-specialised code objects (shared
 *                           between processes / ranks; files are written atomically)
 *     MIFFT_JIT_VERBOSE=1   say on stderr why a runtime specialisation failed
 * Nothing else is configurable in libmifft.so: every size threshold of the plan-time policy is a
 * constant derived from the 256-MiB Infinity Cache (kInfinityCacheBytes).  The measurement switches the
 * scripts under tools/ use (MIFFT_ND_CACHE, MIFFT_NTS_*, MIFFT_FOURSTEP_*, MIFFT_FS_*, MIFFT_ROW2D,
 * MIFFT_JIT_NT, MIFFT_JIT_IMAGE, MIFFT_DPP, MIFFT_HERM*, MIFFT_HS, MIFFT_GRID_PER_CU) and the fault injection of the tests
 * (MIFFT_TEST_FAIL_SCRATCH_ALLOC) exist only in the LAB build, libmifft_lab.so
 * (-DMIFFT_EXPERIMENTAL -DMIFFT_TESTING, same ABI), together with the experimental kernels that stayed
 * negative results; the host package loads it only when MIFFT_LIBRARY points at it.
 */
#ifndef MIFFT_H
#define MIFFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIFFT_VERSION_MAJOR 0
#define MIFFT_VERSION_MINOR 1

#define MIFFT_MAX_DIMS 6
#define MIFFT_MAX_STAGES 64

/* element types (in_dtype: any of these; out_dtype: F32 or F64 only,
 * reference: `comptime assert out_dtype.is_floating_point()` fft/fft/_fft.mojo:226) */
typedef enum {
    MIFFT_F32 = 0,
    MIFFT_F64 = 1,
    MIFFT_U8 = 2, /* reference 2-D/3-D tests feed uint8: fft/tests.mojo:467,524 */
    MIFFT_I32 = 3,
    /* the reference casts ANY element type in its first-stage load (`x.load(...).cast[out_dtype]()`,
     * fft/fft/_fft.mojo:243-257); these are widened to the plan's out_dtype in the first pass the same way: */
    MIFFT_I8 = 4,
    MIFFT_I16 = 5,
    MIFFT_U16 = 6,
    MIFFT_F16 = 7,  /* IEEE binary16 */
    MIFFT_BF16 = 8  /* bfloat16: the upper 16 bits of a binary32 */
} mifft_dtype;

/* error codes; the reference raises these as compile-time asserts
 * (fft/fft/fft.mojo:22-46, fft/fft/_utils.mojo:189-220) or DeviceContext errors */
typedef enum {
    MIFFT_OK = 0,
    MIFFT_ERR_BAD_RANK = -1,       /* ndim < 1 or > MIFFT_MAX_DIMS          (fft.mojo:22-26) */
    MIFFT_ERR_BAD_DIM = -2,        /* a transformed dim is < 2              (fft.mojo:43-46) */
    MIFFT_ERR_BAD_COMPONENTS = -3, /* C_in not in {1,2}                     (fft.mojo:30-32) */
    MIFFT_ERR_BAD_DTYPE = -4,      /* out dtype not floating                (_fft.mojo:226)  */
    MIFFT_ERR_BAD_BASES = -5,      /* powers of bases do not multiply to N  (_utils.mojo:206-219) */
    MIFFT_ERR_BASE_ONE = -6,       /* a base equals 1 (or 0)                (_utils.mojo:220) */
    MIFFT_ERR_NO_BASES = -7,       /* empty bases list for a dim            (_utils.mojo:189-191) */
    MIFFT_ERR_BAD_BATCH = -8,      /* batch < 0                                               */
    MIFFT_ERR_TOO_LARGE = -9,      /* a dim does not fit one workgroup's LDS (reference: dims >
                                      max_thread_block_size are unsupported off NVIDIA,
                                      _ndim_fft_gpu.mojo:100-108,514-519)                     */
    MIFFT_ERR_NO_DEVICE = -10,     /* no usable HIP device / device < 0                       */
    MIFFT_ERR_HIP = -11,           /* HIP runtime error; text via mifft_last_error()          */
    MIFFT_ERR_NULL = -12,          /* NULL plan / buffer                                      */
    MIFFT_ERR_ALIAS = -13,         /* x and out overlap (reference is out-of-place)           */
    MIFFT_ERR_BUFFER_TOO_SMALL = -14
} mifft_status;

/* plan flags */
#define MIFFT_FLAG_NONE 0u
/* Run every user radix stage literally (one LDS pass per stage, one thread per
 * output element, sequential complex-FMA accumulation) exactly as
 * _radix_n_fft_kernel_stockham does (fft/fft/_fft.mojo:189-296).  Default (0)
 * lets the planner fuse consecutive stages into register butterflies when a
 * specialised kernel exists for the dimension.  Mirrors the reference's
 * `_test=` code-path forcing (fft/fft/_ndim_fft_gpu.mojo:453-459). */
#define MIFFT_FLAG_FAITHFUL_STAGES 1u

typedef struct mifft_plan mifft_plan;

/*
 * mifft_plan_create -- replaces the GPU overload of
 *   plan_fft[in_dtype, out_dtype, in_layout, out_layout, *, bases, inverse,
 *            runtime_twfs, max_cluster_size, _test](*, ctx) -> _GPUPlan
 *   (fft/fft/fft.mojo:161-210; _GPUPlan.__init__ fft/fft/_ndim_fft_gpu.mojo:179-207).
 *
 *  device         HIP device ordinal (>= 0).
 *  ndim, dims     transformed dims d0..d{ndim-1} (the layout minus batch and C).
 *  batch          leading dimension (independent transforms).
 *  in_components  1 = real input, 2 = interleaved complex input.
 *  inverse        0 forward, 1 inverse.
 *  bases_flat / bases_len
 *                 per-dim user radix lists, concatenated; bases_len[d] entries
 *                 belong to dim d.  NULL/NULL selects the reference's GPU default
 *                 (_estimate_best_bases target "gpu", fft/fft/fft.mojo:49-104).
 *                 Lists are normalised exactly like _build_ordered_bases
 *                 (fft/fft/_utils.mojo:162-183).
 *  flags          MIFFT_FLAG_*.
 * The plan owns its device twiddle tables (a few KiB per dimension).  The reference's
 * plan always owns a scratch tensor of the output size (_ndim_fft_gpu.mojo:185); here
 * only three routes do -- query it with mifft_plan_scratch_bytes():
 *   - a STRIDED dimension longer than 4096 points: the two-pass four-step through the scratch
 *     (column tiles of N1 points out -> scratch with a row-granular transposition, column tiles of
 *     N2 points scratch -> out), the default for such dimensions,
 *   - its fallback when a factor has no fused column tile: transpose -> rows -> transpose, and
 *   - a contiguous dimension beyond one LDS row (> 16384 points) whose factorisation
 *     has no transposed-store kernel (the three-launch four-step).
 * Every other plan (all BASELINE configs) works without scratch: the contiguous
 * dimension goes x -> out, the others run in place on out.  Running out of device
 * memory for the scratch is MIFFT_ERR_HIP, never a silent fallback.
 * A strided dimension must span fewer than 2^32 elements (N * stride < 2^32, i.e. one
 * transform below 32 GB): MIFFT_ERR_TOO_LARGE otherwise.
 * mifft_plan_stages() reports the user's literal stages; the executed kernel fuses
 * consecutive stages into 2-4 register-butterfly passes unless
 * MIFFT_FLAG_FAITHFUL_STAGES is set (mifft_plan_kernel_name() shows the fused radices).
 * `runtime_twfs`, `max_cluster_size` have no MI355X meaning and do not exist here.
 */
int mifft_plan_create(mifft_plan** out_plan, int device, int in_dtype, int out_dtype,
                      int ndim, const int64_t* dims, int64_t batch, int in_components,
                      int inverse, const uint32_t* bases_flat, const int32_t* bases_len,
                      uint32_t flags);

/*
 * mifft_plan_create_slab -- mifft_plan_create for ONE SLAB of a batch that is split over several plans (one per GPU,
 * SURVEY.md 8e; the reference has no counterpart: its plan always covers the whole batch, fft/fft/fft.mojo:161-210).
 * `batch` transforms are planned and executed; every choice that depends on the tensor size (streaming / non-temporal
 * twins, cache policy, four-step threshold) is made as for `whole_batch` transforms, so that the slab's results equal
 * the same rows of a single plan over the whole batch BIT FOR BIT.  whole_batch = 0 is mifft_plan_create.
 */
int mifft_plan_create_slab(mifft_plan** out_plan, int device, int in_dtype, int out_dtype,
                           int ndim, const int64_t* dims, int64_t batch, int in_components,
                           int inverse, const uint32_t* bases_flat, const int32_t* bases_len,
                           uint32_t flags, int64_t whole_batch);

/*
 * mifft_exec -- replaces the GPU overload of
 *   fft(output, x, ctx, *, plan)  (fft/fft/fft.mojo:262-323 ->
 *   _run_gpu_nd_fft fft/fft/_ndim_fft_gpu.mojo:462-642).
 * `x` and `out` are DEVICE pointers on the plan's device.  Asynchronous on
 * `stream`; no allocation, no hidden synchronisation (the reference's caller
 * synchronises: fft/bench.mojo:51-52).  `x` is never written; every element of
 * `out` is written; `x` and `out` must not overlap.  One in-flight exec per plan
 * per stream order (same contract as the reference's shared scratch).
 */
int mifft_exec(const mifft_plan* plan, const void* x, void* out, void* stream);

/*
 * mifft_exec_batch -- same as mifft_exec on a sub-range of the leading
 * dimension: transforms `count` batch entries starting at `first` (pointers
 * still address entry 0).  This is what the batch-sharded multi-GPU host uses:
 * every rank runs the same plan on its own contiguous slab.  No reference
 * counterpart (the reference is single-device); the grid-over-batch it
 * generalises is fft/fft/_ndim_fft_gpu.mojo:428-450.
 */
int mifft_exec_batch(const mifft_plan* plan, const void* x, void* out, int64_t first,
                     int64_t count, void* stream);

/* frees device tables; replaces _GPUPlan's destructor (ArcPointer / DeviceBuffer drop) */
void mifft_plan_destroy(mifft_plan* plan);

/* --- plan introspection (the reference exposes these as compile-time values) --- */

/* ordered (descending) per-stage radices of dim `dim` -- _get_ordered_bases_processed_list
 * (fft/fft/_utils.mojo:186-221).  Returns the stage count, or a negative status. */
int mifft_plan_stages(const mifft_plan* plan, int dim, uint32_t* radices_out, int capacity);

/* name of the kernel family chosen for dim `dim` ("generic", "wave1024", ...) */
const char* mifft_plan_kernel_name(const mifft_plan* plan, int dim);

/* number of kernel launches one exec enqueues (reference: d + 2(d-1),
 * fft/fft/_ndim_fft_gpu.mojo:634-642) */
int mifft_plan_num_launches(const mifft_plan* plan);

/* required sizes in bytes of x and out for the full batch */
size_t mifft_plan_in_bytes(const mifft_plan* plan);
size_t mifft_plan_out_bytes(const mifft_plan* plan);
/* bytes of the plan-owned scratch tensor (0 for plans that need none, see mifft_plan_create);
 * reference: _GPUPlan.calc_buf, fft/fft/_ndim_fft_gpu.mojo:176-185, which exists for every plan */
size_t mifft_plan_scratch_bytes(const mifft_plan* plan);

/*
 * mifft_plan_device_status -- device-side error flags of the execs enqueued so far with this plan on `stream`;
 * reads and clears them.  Only the experimental L2-resident image kernel of the LAB build can raise one: its
 * XCD-local barrier spins are bounded, and an exec whose spin expired (bit 0) or that was dispatched with a surplus
 * workgroup on one XCD (bit 1) has produced output that must not be trusted.  Every other kernel family has no
 * inter-workgroup dependency and reports 0 without touching the device.  SYNCHRONISES `stream` when the plan
 * contains such a pass -- this is a check to run after a batch of execs, not part of the exec path.
 * No reference counterpart (the reference's only device-wide synchronisation is the kernel boundary,
 * fft/fft/_ndim_fft_gpu.mojo:634-642).
 */
int mifft_plan_device_status(const mifft_plan* plan, void* stream, uint32_t* flags_out);

/* --- planner helpers, usable without a device (pure host logic) --- */

/* _build_ordered_bases (fft/fft/_utils.mojo:162-183) + the product / base==1 checks of
 * _get_ordered_bases_processed_list (:186-221).  Returns stage count or negative status. */
int mifft_ordered_bases(uint32_t length, const uint32_t* bases, int nbases,
                        uint32_t* ordered_out, int capacity);

/* _estimate_best_bases (fft/fft/fft.mojo:49-104); target_gpu != 0 selects the "gpu"
 * branch (:61-80), else the prime list (:83-104).  Returns count or negative status. */
int mifft_estimate_bases(uint32_t length, int target_gpu, uint32_t* bases_out, int capacity);

/* --- diagnostics --- */
const char* mifft_last_error(void);       /* thread-local text of the last failure */
const char* mifft_status_string(int status);
int mifft_version(void);                  /* major*100 + minor */
int mifft_device_count(void);             /* usable HIP devices, 0 if none */

/*
 * mifft_time_exec -- enqueue `iters` back-to-back mifft_exec calls on `stream`
 * bracketed by HIP events recorded on that same stream and return the average
 * milliseconds per exec in *ms_out (synchronises the events, not the device).
 * Measurement helper for bench.py's roofline line; mirrors the timed loop of
 * fft/bench.mojo:48-54.
 */
int mifft_time_exec(const mifft_plan* plan, const void* x, void* out, void* stream, int iters,
                    float* ms_out);

/*
 * mifft_jit_precompile -- build (do not load) the runtime-specialised fused kernel that plan creation would use for
 * a dimension of `length` points without a precompiled table entry, and report the size of its gfx950 code object.
 * The reference compiles every shape at build time (shapes are Mojo comptime parameters, fft/fft/fft.mojo:123-135);
 * libmifft specialises its tile kernel with hipRTC on first use and caches the result per process.  Needs no
 * device: usable to warm the cache ahead of plan creation or to check a length on a build machine.
 * in_dtype: element type of the tensor the pass reads (any mifft_dtype; widened to out_dtype in the load);
 * out_dtype: MIFFT_F32 / MIFFT_F64; strided: 0 contiguous rows, 1 the in-place column-tile form, 2 a fused
 * length x length plane (two innermost dimensions in one LDS tile), k >= 8 a fused k x length plane (length = the
 * contiguous side); real_input != 0: the C_in = 1 twin.
 * Returns MIFFT_OK, or MIFFT_ERR_TOO_LARGE when the length has no fused configuration (a prime factor above 4093,
 * three prime factors above 31, or a tile beyond the 160 KiB of LDS) -- such lengths run on the literal-stage kernels.
 */
int mifft_jit_precompile(int in_dtype, int out_dtype, int64_t length, int strided, int real_input,
                         size_t* code_bytes_out);

#ifdef __cplusplus
}
#endif
#endif /* MIFFT_H */
